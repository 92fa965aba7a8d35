#!/usr/bin/env python3
"""Halo-staged conv tiles (15 / 16) against the ping-pong tiles they derive from (9 / 11) on the U-Net's
stride-1 convolutions at the 32x32 and 16x16 levels (R rows), isolated, graph-replayed.

    python tools/conv_ring_probe.py [--rows 32]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_tf2_amd import ops  # noqa: E402
from tools.gemm_bench import time_fn  # noqa: E402

SHAPES = [(32, 320, 320, 7), (32, 640, 320, 2), (32, 960, 320, 1), (32, 640, 640, 1),
          (16, 640, 640, 6), (16, 1280, 640, 1), (16, 960, 640, 1), (16, 1920, 640, 1), (16, 320, 640, 1),
          (16, 1280, 1280, 1)]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--rows", type=int, default=32)
  args = ap.parse_args()
  dev = torch.device("cuda:0")
  R = args.rows
  ws = ops.new_workspace(dev) if hasattr(ops, "new_workspace") else None
  tot = {}
  for hw, cin, cout, n in SHAPES:
    x = torch.randn(R, hw, hw, cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty(R, hw, hw, cout, device=dev, dtype=torch.bfloat16)
    gf = 2.0 * R * hw * hw * cout * 9 * cin * 1e-9
    line = f"conv {hw:2d}^2 {cin:4d}->{cout:4d} x{n}: "
    for tile in (9, 15, 11, 16):
      bn = 160 if tile in (9, 15) else 128
      if cout % bn:
        continue
      best = None
      for split in (1, 2):
        if split == 2 and R * hw * hw // 256 * (cout // bn) >= 256:
          continue
        try:
          ms = time_fn(lambda: ops.conv3x3(x, w, out, tile=tile, split_k=split), 3)
        except Exception as e:
          line += f" t{tile}/s{split}: {str(e)[:40]}"
          continue
        if best is None or ms < best[0]:
          best = (ms, split)
      if best:
        line += f" t{tile}: {best[0] * 1e3:6.1f} us (s{best[1]}) {gf / best[0]:5.0f} TF |"
        tot[tile] = tot.get(tile, 0.0) + best[0] * n
    print(line, flush=True)
  print("ms per U-Net evaluation over these shapes:", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
  main()
