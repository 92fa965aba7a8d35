#!/usr/bin/env python3
"""Sweep (tile, split_k) for the small-M convolutions (4x4 / 8x8 maps at R rows): checks the
cost model's choice against the measured best."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_tf2_amd import ops  # noqa: E402
from tools.gemm_bench import time_fn  # noqa: E402

dev, dt, R = torch.device("cuda:0"), torch.bfloat16, 32
for hw, cin, cout in ((4, 1280, 1280), (4, 2560, 1280), (8, 1280, 1280), (8, 2560, 1280), (8, 640, 1280), (16, 640, 640)):
  x = torch.randn(R, hw, hw, cin, device=dev).to(dt)
  w = torch.randn(cout, 9 * cin, device=dev).to(dt)
  b = torch.zeros(cout, device=dev)
  out = torch.empty(R, hw, hw, cout, device=dev, dtype=dt)
  gf = 2.0 * R * hw * hw * 9 * cin * cout / 1e9
  auto = time_fn(lambda: ops.conv3x3(x, w, out, bias=b)) * 1e3
  res = []
  for tile in (1, 2, 3, 4, 6):
    for split in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32):
      try:
        us = time_fn(lambda: ops.conv3x3(x, w, out, bias=b, tile=tile, split_k=split), rounds=3) * 1e3
      except Exception:
        continue
      res.append((us, tile, split))
  res.sort()
  best = ", ".join(f"t{t}/s{s}: {u:.1f}us" for u, t, s in res[:5])
  print(f"conv {hw}x{hw} {cin}->{cout} ({gf:.1f} GF): auto {auto:.1f} us ({gf / auto / 1e-3 / 1e3:.0f} TF/s) | best {best}")
