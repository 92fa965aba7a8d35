#!/usr/bin/env python3
"""The C x C Dense launch of the 16x16 level (M = 8192 rows, K = N = 640; unet.py:248-338 at C = 640), the class
VERDICT r3 puts at 0.18 of the MFMA roof.

  python tools/dense_probe.py time        every tile / epilogue, HOT (one buffer set, 10 launches per graph) and
                                          ROTATING (16 buffer sets = 350 MB, so operands come from HBM / MALL as
                                          they do in the step)
  python tools/dense_probe.py pmc [tile]  20 launches on rotating buffers: workload for rocprofv3 --pmc passes
                                          (tools/pmc_kernel_avg.py <counter_collection.csv> gemm3_kernel)
  [--m 8192 --k 640 --n 640]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_tf2_amd import ops  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("mode", choices=["time", "pmc"])
  ap.add_argument("tile", nargs="?", type=int, default=14)
  ap.add_argument("--m", type=int, default=8192)
  ap.add_argument("--k", type=int, default=640)
  ap.add_argument("--n", type=int, default=640)
  ap.add_argument("--sets", type=int, default=16)
  args = ap.parse_args()
  M, K, N = args.m, args.k, args.n
  dev, bf = torch.device("cuda:0"), torch.bfloat16
  sets = []
  for i in range(args.sets):
    x = torch.randn(M, K, device=dev).to(bf)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
    r = torch.randn(M, N, device=dev).to(bf)
    o = torch.empty(M, N, device=dev, dtype=bf)
    sets.append((x, w, r, o))
  b = torch.randn(N, device=dev)

  def launch(s, tile, res, split=0):
    x, w, r, o = sets[s]
    ops.linear(x, w, o, bias=b, residual=r if res else None, tile=tile, split_k=split)

  if args.mode == "pmc":
    for i in range(20):
      launch(i % args.sets, args.tile, True)
    torch.cuda.synchronize()
    print("done")
    return

  def timed(fn_list, inner):
    for f in fn_list:
      f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
      for f in fn_list:
        f()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      g.replay()
      e1.record()
      e1.synchronize()
      best = min(best, e0.elapsed_time(e1) / inner)
    return best * 1e3

  gf = 2.0 * M * N * K / 1e9
  print(f"# M={M} K={K} N={N}: {gf:.2f} GFLOP, {(M * K + N * K + M * N) * 2 / 1e6:.1f} MB (+{M * N * 2 / 1e6:.1f} residual)")
  print(f"# {'tile':>4s} {'res':>3s}  {'hot us':>8s} {'TF/s':>6s}   {'rotating us':>11s} {'TF/s':>6s}")
  for tile in (14, 13, 11, 12, 10, 2, 3, 4, 17, 18, 19, 0):
    if tile in (13, 10) and N % 160:
      continue
    for res in (False, True):
      try:
        hot = timed([lambda: launch(0, tile, res)] * 10, 10)
        rot = timed([(lambda s=s: launch(s, tile, res)) for s in range(args.sets)], args.sets)
      except RuntimeError as e:
        print(f"  {tile:4d} {int(res):3d}  -- {str(e)[:80]}")
        continue
      print(f"  {tile:4d} {int(res):3d}  {hot:8.1f} {gf / hot * 1e3:6.0f}   {rot:11.1f} {gf / rot * 1e3:6.0f}")


if __name__ == "__main__":
  main()
