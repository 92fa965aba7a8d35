#!/usr/bin/env python3
"""Probe: fixed cost of one ldm_gemm launch vs its K-dependent part (graph-replay timing of 10
back-to-back launches), and the back-to-back period of a trivial kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_tf2_amd import ops  # noqa: E402
from tools.gemm_bench import time_fn  # noqa: E402

dev, dt = torch.device("cuda:0"), torch.bfloat16
tiny_in = torch.zeros(8, 8, device=dev, dtype=dt)
tiny_out = torch.zeros(8, 8, device=dev, dtype=dt)
print(f"trivial kernel (cast 8x8) back-to-back period: {time_fn(lambda: ops.cast(tiny_in, tiny_out)) * 1e3:.2f} us")
for M, N in ((32768, 320), (8192, 640), (2048, 1280), (32768, 2560)):
  for K in (64, 128, 320, 640, 1280):
    x = torch.randn(M, K, device=dev).to(dt)
    w = torch.randn(N, K, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    ms = time_fn(lambda: ops.linear(x, w, out))
    print(f"M={M:6d} N={N:5d} K={K:5d}: {ms * 1e3:7.1f} us   ({2 * M * N * K / ms / 1e9:7.1f} TFLOP/s; out {M * N * 2 / 1e6:.0f} MB, A {M * K * 2 / 1e6:.0f} MB)")
