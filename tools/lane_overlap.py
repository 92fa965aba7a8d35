#!/usr/bin/env python3
"""Did the branches of a multi-branch step (UNet(lanes=N)) really run side by side?

Reads a `rocprofv3 --kernel-trace --output-format csv` trace, takes ONE U-Net evaluation (between two
`time_embedding_kernel` dispatches) and reports
  * span, summed kernel time, the time during which >= 2 kernels were in flight, idle time;
  * per hardware queue: dispatches, summed kernel time;
  * per kernel name: dispatches, average duration (compare with the single-branch trace of the same rows
    to see how much a kernel stretches when it shares the chip);
  * optionally the first K dispatches as a two-column timeline.

    python tools/lane_overlap.py <kernel_trace.csv> [--eval K] [--show 60]
"""
import argparse
import collections
import csv
import re
import sys


def short(name):
  name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("ldm_gemm_detail::", "")
  name = re.sub(r"\(.*$", "", name)
  return name.replace("unsigned short", "bf16")[:60]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("trace_csv")
  ap.add_argument("--eval", type=int, default=-1)
  ap.add_argument("--show", type=int, default=0)
  args = ap.parse_args()
  rows = list(csv.DictReader(open(args.trace_csv)))
  key = lambda *names: next((n for n in names if n in rows[0]), None)
  kn, ks, ke = key("Kernel_Name", "Name"), key("Start_Timestamp", "Start"), key("End_Timestamp", "End")
  kq = key("Queue_Id", "Queue", "Stream_Id")
  rows.sort(key=lambda r: int(r[ks]))
  marks = [i for i, r in enumerate(rows) if "time_embedding_kernel" in r[kn] or "select_row_kernel" in r[kn]]
  if len(marks) < 2:
    sys.exit("fewer than two U-Net evaluations in the trace")
  k = args.eval if args.eval >= 0 else len(marks) // 2
  k = min(k, len(marks) - 2)
  seg = rows[marks[k]:marks[k + 1]]
  t0 = int(seg[0][ks])
  ev = []
  for r in seg:
    ev.append((int(r[ks]), 1))
    ev.append((int(r[ke]), -1))
  ev.sort()
  depth, last, busy1, busy2, idle = 0, t0, 0, 0, 0
  for t, d in ev:
    dt = t - last
    if depth == 0:
      idle += dt
    elif depth == 1:
      busy1 += dt
    else:
      busy2 += dt
    depth += d
    last = t
  span = last - t0
  tot = sum(int(r[ke]) - int(r[ks]) for r in seg)
  print(f"# evaluation {k} of {len(marks)}: {len(seg)} dispatches, span {span / 1e3:.1f} us, summed kernel time {tot / 1e3:.1f} us")
  print(f"# one kernel in flight {busy1 / 1e3:.1f} us, two or more {busy2 / 1e3:.1f} us ({100.0 * busy2 / span:.1f} % of the span), idle {idle / 1e3:.1f} us")
  if kq:
    q = collections.defaultdict(lambda: [0, 0])
    for r in seg:
      q[r[kq]][0] += 1
      q[r[kq]][1] += int(r[ke]) - int(r[ks])
    for name, (n, t) in sorted(q.items()):
      print(f"# queue {name}: {n} dispatches, {t / 1e3:.1f} us of kernel time")
  by = collections.defaultdict(lambda: [0, 0])
  for r in seg:
    by[short(r[kn])][0] += 1
    by[short(r[kn])][1] += int(r[ke]) - int(r[ks])
  print(f"# {'kernel':60s} {'n':>4s} {'avg_us':>8s} {'sum_us':>9s}")
  for name, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {name:60s} {n:4d} {t / n / 1e3:8.1f} {t / 1e3:9.1f}")
  if args.show:
    print("# first dispatches: t_us dur_us queue kernel")
    for r in seg[:args.show]:
      print(f"  {(int(r[ks]) - t0) / 1e3:8.1f} {(int(r[ke]) - int(r[ks])) / 1e3:7.1f}  {r[kq] if kq else '-':>4s}  {short(r[kn])}")


if __name__ == "__main__":
  main()
