#!/usr/bin/env python3
"""GB/s of the HBM-bound kernels (GroupNorm partial/apply, LayerNorm, attention) on the
U-Net's shapes at R rows.  Algorithmic bytes = read once + write once."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldm_tf2_amd import ops  # noqa: E402


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
  args = ap.parse_args()
  R, dt, dev = args.rows, torch.bfloat16, torch.device("cuda:0")
  tot = 0.0
  print("LayerNorm")
  for T, C, cnt in [(1024, 320, 15), (256, 640, 15), (64, 1280, 15), (16, 1280, 3)]:
    x = torch.randn(R * T, C, device=dev).to(dt)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    out = torch.empty_like(x)
    ms = time_fn(lambda: ops.layernorm(x, g, b, out))
    tot += ms * cnt
    print(f"  [{R * T:6d} x {C:4d}] x{cnt:2d}  {ms * 1e3:7.1f} us  {2 * x.numel() * 2 / ms / 1e6:7.0f} GB/s")
  print("GroupNorm (partial + apply)")
  for hw, C, cnt in [(32, 320, 13), (16, 640, 11), (8, 1280, 11), (4, 1280, 12), (32, 640, 2), (32, 960, 1),
                     (16, 1920, 1), (16, 1280, 1), (16, 960, 1), (16, 320, 1), (8, 2560, 2), (8, 1920, 1),
                     (8, 640, 1), (4, 2560, 3)]:
    x = torch.randn(R, hw, hw, C, device=dev).to(dt)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    out = torch.empty_like(x)
    part = torch.empty(R * 128 * 64, device=dev)
    ms = time_fn(lambda: ops.groupnorm(x, g, b, out, 1e-5, silu=True, partial=part))
    tot += ms * cnt
    print(f"  [{R},{hw:2d},{hw:2d},{C:4d}] x{cnt:2d}  {ms * 1e3:7.1f} us  {3 * x.numel() * 2 / ms / 1e6:7.0f} GB/s (2 reads + 1 write)")
  print("attention (self, cross)")
  for T, Tk, S, sp, cnt in [(1024, 1024, 40, 48, 5), (1024, 77, 40, 48, 5), (256, 256, 80, 80, 5), (256, 77, 80, 80, 5),
                            (1024, 1024, 40, 64, 0), (256, 256, 80, 96, 0),
                            (64, 64, 160, 160, 5), (64, 77, 160, 160, 5), (16, 16, 160, 160, 1), (16, 77, 160, 160, 1)]:
    H = 8
    q = torch.randn(R, T, H * sp, device=dev).to(dt)
    k = torch.randn(R, Tk, H * sp, device=dev).to(dt)
    tkp = (Tk + 7) // 8 * 8
    vt = torch.randn(R, H * sp, tkp, device=dev).to(dt)
    out = torch.empty_like(q)
    ms = time_fn(lambda: ops.attention(q, k, vt, out, H, sp, S ** -0.5))
    tot += ms * cnt
    gf = 4.0 * R * H * T * Tk * S / 1e9
    print(f"  T={T:4d} Tk={Tk:4d} S={S:3d} x{cnt}  {ms * 1e3:7.1f} us  {gf / ms:7.1f} TFLOP/s (unpadded flops)")
  print(f"total per U-Net step: {tot:.2f} ms")


if __name__ == "__main__":
  main()
