#!/usr/bin/env python3
"""Cold vs warm instruction cache: per-dispatch durations of X in the replayed sequence  F1 F2 F3 X X X  (F: launches
with large, different code on all CUs), from a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d /tmp/ic -- python3 tools/icache_probe.py
    python3 tools/icache_probe.py --parse /tmp/ic

The first X after the F's starts with the code of other kernels in the CUs' instruction caches (what every launch
of a U-Net step sees); the second and third find their own.
"""
import argparse
import csv
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(d):
  tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
  rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
  names = [r["Kernel_Name"] for r in rows]
  dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in rows]
  # runs of three identical kernels
  by = {}
  i = 0
  while i + 2 < len(rows):
    if names[i] == names[i + 1] == names[i + 2] and (i == 0 or names[i - 1] != names[i]):
      by.setdefault(names[i], []).append(dur[i:i + 3])
      i += 3
    else:
      i += 1
  for k, v in by.items():
    a = np.array(v[len(v) // 2:])                     # second half of the replays
    print(f"{k[:70]:70s} first {np.median(a[:, 0]):7.1f} us   second {np.median(a[:, 1]):7.1f}   third {np.median(a[:, 2]):7.1f}   ({len(a)} runs)")


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--parse")
  args = ap.parse_args()
  if args.parse:
    return parse(args.parse)
  import torch
  from ldm_tf2_amd import layout as L, ops
  dev, bf = torch.device("cuda:0"), torch.bfloat16
  R = 32
  g = torch.Generator().manual_seed(0)
  rn = lambda *s: torch.randn(*s, generator=g).to(dev)
  x32 = rn(R, 32, 32, 320).to(bf)
  y32 = torch.empty_like(x32)
  gam, bet = torch.ones(320, device=dev), torch.zeros(320, device=dev)
  w3 = L.conv_kernel(np.random.default_rng(0).standard_normal((3, 3, 320, 320)).astype(np.float32) * 0.02, bf, dev)
  wo = (rn(320, 384) * 0.05).to(bf)
  att = rn(R * 1024, 384).to(bf)
  h = torch.empty(R * 1024, 320, dtype=bf, device=dev)
  b320 = torch.zeros(320, device=dev)
  # flushers: other shapes -> other kernels
  x16 = rn(R, 16, 16, 640).to(bf); y16 = torch.empty_like(x16)
  g16, b16 = torch.ones(640, device=dev), torch.zeros(640, device=dev)
  w16 = L.conv_kernel(np.random.default_rng(1).standard_normal((3, 3, 640, 640)).astype(np.float32) * 0.02, bf, dev)
  q = rn(R, 256, 640).to(bf); k = rn(R, 256, 640).to(bf); vt = rn(R, 640, 256).to(bf); ao = torch.empty_like(q)
  w640 = (rn(640, 640) * 0.05).to(bf); h16 = torch.empty(R * 256, 640, dtype=bf, device=dev); b640 = torch.zeros(640, device=dev)

  def flush():
    ops.conv3x3(x16, w16, y16, bias=b640)
    ops.groupnorm(x16, g16, b16, y16, 1e-5, silu=True)
    ops.attention(q, k, vt, ao, 8, 80, 80 ** -0.5)
    ops.linear(q.view(-1, 640), w640, h16, bias=b640, residual=h16)

  xs = {
      "groupnorm 32x32x320": lambda: ops.groupnorm(x32, gam, bet, y32, 1e-5, silu=True),
      "o-projection [32768,384]x[384,320]": lambda: ops.linear(att, wo, h, bias=b320, residual=h),
      "conv3x3 320->320 @32": lambda: ops.conv3x3(x32, w3, y32, bias=b320),
  }
  for fn in list(xs.values()) + [flush]:
    fn()
  torch.cuda.synchronize()
  gr = torch.cuda.CUDAGraph()
  with torch.cuda.graph(gr):
    for fn in xs.values():
      flush()
      fn(); fn(); fn()
  for _ in range(12):
    gr.replay()
  torch.cuda.synchronize()


if __name__ == "__main__":
  main()
