#!/usr/bin/env python3
"""Timing of ldm_attention on the U-Net's attention shapes (R rows = 2 x batch):
self- and cross-attention of the three transformer levels, graph-replayed, best of 5.

    python tools/attn_bench.py [--rows 32] [--dtype bf16]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("LDM_ATTN_WAVES"):            # A/B of the workgroup shape: tools build only (make tools)
  import tools.toolslib  # noqa: F401,E402
from ldm_tf2_amd import ops  # noqa: E402
from tools.gemm_bench import time_fn  # noqa: E402

# (tokens, heads, head dim, padded head dim, launches per U-Net evaluation)
LEVELS = [(1024, 8, 40, 48, 5), (256, 8, 80, 80, 5), (64, 8, 160, 160, 6)]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--rows", type=int, default=32)
  ap.add_argument("--dtype", default="bf16")
  ap.add_argument("--ms", action="store_true", help="d = 40 level through ldm_attention_ms (matrix-side softmax)")
  args = ap.parse_args()
  dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
  dev = torch.device("cuda:0")
  R = args.rows
  total = 0.0
  for T, H, d, sp, n in LEVELS:
    for Tk, kind in ((T, "self"), (77, "cross")):
      ld = (Tk + 7) // 8 * 8
      q = torch.randn(R, T, H * sp, device=dev).to(dt)
      k = torch.randn(R, Tk, H * sp, device=dev).to(dt)
      vt = torch.randn(R, H * sp, ld, device=dev).to(dt)
      o = torch.empty_like(q)
      use_ms = args.ms and sp == 48 and dt == torch.bfloat16
      if use_ms:                                    # what the projections deliver (include/ldm_hip.h)
        q = (q.float() * (d ** -0.5 * 1.4426950408889634)).to(dt)
        for t_, pad0 in ((q, True), (k, False)):
          v4 = t_.view(R, -1, H, sp)
          v4[..., d:] = 0
          if not pad0:
            v4[..., d] = 1
        v3 = vt.view(R, H, sp, ld)
        v3[:, :, d:, :] = 0
        v3[:, :, d, :] = 1
      ms = time_fn(lambda: ops.attention(q, k, vt, o, H, sp, d ** -0.5, matrix_softmax=use_ms), 5)
      kind = kind + ("*" if use_ms else "")
      gf = 4.0 * R * H * T * Tk * d * 1e-9
      total += ms * n
      print(f"{kind:5s} T={T:5d} Tk={Tk:5d} d={d:3d} (Sp {sp:3d}): {ms * 1e3:7.1f} us  {gf / ms:6.0f} TFLOP/s (unpadded)  x{n}")
  print(f"attention per U-Net evaluation: {total:.3f} ms")


if __name__ == "__main__":
  main()
