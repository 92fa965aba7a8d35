#!/usr/bin/env python3
"""Per-shape timing of ldm_gemm (MFMA GEMM / implicit-GEMM conv) on the U-Net's layer
shapes for every tile configuration, interleaved rounds in one
process, random data.  Prints TFLOP/s per (shape, tile) and the best tile.

    python tools/gemm_bench.py [--rows 32] [--dtype bf16] [--rounds 5] [--filter conv]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_tf2_amd import ops  # noqa: E402

# (kind, hw, Cin/K, Cout/N, count per U-Net eval, extra)
CONVS = [
    (32, 320, 320, 7), (8, 1280, 1280, 7), (16, 640, 640, 6), (8, 2560, 1280, 2), (16, 1280, 1280, 1),
    (32, 640, 640, 1), (32, 640, 320, 2), (16, 1920, 640, 1), (32, 960, 320, 1), (4, 1280, 1280, 11),
    (16, 1280, 640, 1), (4, 2560, 1280, 3), (8, 1920, 1280, 1), (16, 960, 640, 1), (16, 320, 640, 1),
    (8, 640, 1280, 1),
]
GEMMS = [  # (T per row, K, N, count)
    (1024, 320, 320, 30), (256, 640, 640, 30), (64, 1280, 1280, 30),
    (1024, 320, 1024, 5), (256, 640, 1536, 5), (64, 1280, 2560, 5),   # fused q|k projections (padded heads)
    (1024, 320, 2560, 5), (256, 640, 5120, 5), (64, 1280, 10240, 5),   # GEGLU
    (1024, 1280, 320, 5), (256, 2560, 640, 5), (64, 5120, 1280, 5),    # FF out
    (16, 1280, 1280, 8), (16, 1280, 10240, 1), (16, 5120, 1280, 1),
    (1024, 640, 320, 2), (1024, 960, 320, 1), (256, 1920, 640, 1), (64, 2560, 1280, 2), (16, 2560, 1280, 3),
]


def time_fn(fn, rounds=5, inner=10):
  """GPU time per call: `inner` calls captured in a HIP graph and replayed (no host
  launch latency in the measurement); best of `rounds`."""
  fn()
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(inner):
      fn()
  g.replay()
  torch.cuda.synchronize()
  best = 1e9
  for _ in range(rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    e1.synchronize()
    best = min(best, e0.elapsed_time(e1) / inner)
  return best


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--rows", type=int, default=32)
  ap.add_argument("--dtype", default="bf16")
  ap.add_argument("--rounds", type=int, default=5)
  ap.add_argument("--filter", default="")
  ap.add_argument("--tiles", default="0,1,2,3,4")
  ap.add_argument("--match", default="", help="only shapes whose label contains this (e.g. '32:640:640' or 'M=8192')")
  args = ap.parse_args()
  global CONVS, GEMMS
  if args.match:
    CONVS = [c for c in CONVS if args.match in f"{c[0]}:{c[1]}:{c[2]}"]
    GEMMS = [g for g in GEMMS if args.match in f"T={g[0]},K={g[1]},N={g[2]}"]
  dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
  dev = torch.device("cuda:0")
  R = args.rows
  tiles = [int(t) for t in args.tiles.split(",")]
  tot = {t: 0.0 for t in tiles}
  tot_best = 0.0
  tot_gf = 0.0
  print(f"{'shape':38s} {'GF':>8s} " + " ".join(f"{'t' + str(t):>9s}" for t in tiles) + "   (TFLOP/s; ms of best)")
  if "gemm" not in args.filter:
    for hw, cin, cout, cnt in CONVS:
      x = torch.randn(R, hw, hw, cin, device=dev).to(dt)
      w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).to(dt)
      b = torch.randn(cout, device=dev)
      out = torch.empty(R, hw, hw, cout, device=dev, dtype=dt)
      gf = 2.0 * R * hw * hw * cout * 9 * cin / 1e9
      res = []
      for t in tiles:
        if t in (13, 14) and cout % (160 if t == 13 else 128):
          res.append(float("inf"))
          continue
        ms = time_fn(lambda: ops.conv3x3(x, w, out, bias=b, tile=t), args.rounds)
        res.append(ms)
        tot[t] += ms * cnt
      tot_best += min(res) * cnt
      tot_gf += gf * cnt
      print(f"conv {hw:3d}x{hw:<3d} {cin:5d}->{cout:<5d} x{cnt:<3d}        {gf:8.1f} " +
            " ".join(f"{gf / ms:9.1f}" for ms in res) + f"   {min(res):.3f}")
  if "conv" not in args.filter:
    for T, K, N, cnt in GEMMS:
      x = torch.randn(R * T, K, device=dev).to(dt)
      w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
      b = torch.randn(N, device=dev)
      out = torch.empty(R * T, N, device=dev, dtype=dt)
      gf = 2.0 * R * T * N * K / 1e9
      res = []
      for t in tiles:
        if t in (13, 14) and N % (160 if t == 13 else 128):
          res.append(float("inf"))
          continue
        ms = time_fn(lambda: ops.linear(x, w, out, bias=b, tile=t), args.rounds)
        res.append(ms)
        tot[t] += ms * cnt
      tot_best += min(res) * cnt
      tot_gf += gf * cnt
      print(f"gemm M={R * T:6d} K={K:5d} N={N:5d} x{cnt:<3d}    {gf:8.1f} " +
            " ".join(f"{gf / ms:9.1f}" for ms in res) + f"   {min(res):.3f}")
  if "conv" not in args.filter:
    # GEGLU epilogue (value * exact-erf gelu(gate)): the three FF-in GEMMs with their real epilogue
    for T, K, N in [(1024, 320, 2560), (256, 640, 5120), (64, 1280, 10240)]:
      x = torch.randn(R * T, K, device=dev).to(dt)
      w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
      b = torch.randn(N, device=dev)
      out = torch.empty(R * T, N // 2, device=dev, dtype=dt)
      for gt in (0, 1, 2, 11, 14):
        ms = time_fn(lambda: ops.linear(x, w, out, bias=b, act=ops.ACT_GEGLU, tile=gt), args.rounds)
        print(f"geglu M={R * T:6d} K={K:5d} N={N:5d} tile {gt:2d}: {ms * 1e3:7.1f} us  {2.0 * R * T * N * K / 1e9 / ms:7.1f} TFLOP/s")
  print("total ms per U-Net step by tile:", {t: round(v, 2) for t, v in tot.items()},
        "best-per-shape:", round(tot_best, 2), f"=> {tot_gf / tot_best:.0f} TFLOP/s")


if __name__ == "__main__":
  main()
