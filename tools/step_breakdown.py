#!/usr/bin/env python3
"""Where one U-Net step spends its time, per ldm_gemm problem (in situ).

Runs the full-size U-Net step eagerly with every ldm_gemm launch bracketed by HIP events
(tools.gemm_hooks.time_gemms) and groups the brackets by problem key: launches per step, total
microseconds, algorithmic TFLOP/s and the (tile, split) the launch used.  Brackets add
~1-2 us per launch, so small launches read a little slow; the ranking is what matters.

    python tools/step_breakdown.py [--batch 16] [--latent 32] [--dtype bf16] [--reps 5]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ldm_tf2_amd import weights as Wt  # noqa: E402
from tools.gemm_hooks import time_gemms  # noqa: E402
from ldm_tf2_amd.unet import UNet  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--batch", type=int, default=16)
  ap.add_argument("--latent", type=int, default=32)
  ap.add_argument("--dtype", default="bf16")
  ap.add_argument("--reps", type=int, default=5)
  ap.add_argument("--out", default="")
  args = ap.parse_args()
  dev = torch.device("cuda:0")
  dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
  cfg = bench.FULL
  w = Wt.init_weights(Wt.unet_manifest(**cfg["unet"]), seed=2, scope="unet")
  unet = UNet(**cfg["unet"], weights=w, dtype=dt, device=dev)
  R = 2 * args.batch
  ctx = (torch.randn(R, 77, 1280, device=dev) * 0.5).to(dt)
  unet.set_context(ctx)
  x = torch.randn(R, args.latent, args.latent, 4, device=dev)
  t = torch.full((R,), 981, dtype=torch.int32, device=dev)
  for _ in range(2):
    unet.forward(x, t_rows=t, shared_t=True)
  torch.cuda.synchronize()
  agg = {}
  tot_step = 0.0
  for _ in range(args.reps):
    sink = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with time_gemms(sink):
      e0.record()
      unet.forward(x, t_rows=t, shared_t=True)
      e1.record()
      torch.cuda.synchronize()
    tot_step += e0.elapsed_time(e1)
    for rec in sink:
      a, b, key, info = rec
      d = agg.setdefault(key, {"n": 0, "us": 0.0, "info": info})
      d["n"] += 1
      d["us"] += a.elapsed_time(b) * 1e3
  rows = []
  for key, d in agg.items():
    M, N, K, batch, act, dtype, tile, split = d["info"]
    n = d["n"] / args.reps
    us = d["us"] / args.reps
    gf = 2.0 * M * N * K * batch * n * 1e-9
    rows.append(dict(key=key, launches=n, us=us, us_each=us / n, gflop=gf, tflops=gf / us * 1e3 if us else 0,
                     tile=tile, split=split))
  rows.sort(key=lambda r: -r["us"])
  tot_us = sum(r["us"] for r in rows)
  tot_gf = sum(r["gflop"] for r in rows)
  print(f"eager step (bracketed): {tot_step / args.reps:.2f} ms; ldm_gemm launches sum {tot_us / 1e3:.2f} ms, "
        f"{tot_gf:.0f} GFLOP -> {tot_gf / tot_us * 1e3:.0f} TFLOP/s")
  print(f"{'problem':86s} {'n':>4s} {'us/step':>8s} {'us each':>8s} {'TF/s':>6s} {'tile':>4s} {'spl':>3s} {'cum%':>5s}")
  cum = 0.0
  for r in rows:
    cum += r["us"]
    print(f"{r['key']:86s} {r['launches']:4.0f} {r['us']:8.1f} {r['us_each']:8.1f} {r['tflops']:6.0f} "
          f"{r['tile']:4d} {r['split']:3d} {100 * cum / tot_us:5.1f}")
  if args.out:
    with open(args.out, "w") as f:
      json.dump(dict(step_ms=tot_step / args.reps, family_ms=tot_us / 1e3, rows=rows), f, indent=1)


if __name__ == "__main__":
  main()
