#!/usr/bin/env python3
"""Timeline of ONE U-Net evaluation out of a `rocprofv3 --kernel-trace --output-format csv` run:
every dispatch between two consecutive `time_embedding_kernel` launches, in start order, with its
duration and the idle gap in front of it.  The host walk is deterministic, so the n-th line is
always the same layer: this is the table that says which launch of the step costs what IN SITU
(the per-kernel averages of --stats mix levels).

    python tools/step_timeline.py <..._kernel_trace.csv> [--eval K] [--csv out.csv] > timeline.txt
"""
import argparse
import csv
import re
import sys


def short(name):
  name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("ldm_gemm_detail::", "")
  name = re.sub(r"\(.*$", "", name)
  return name.replace("unsigned short", "bf16")[:64]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("trace_csv")
  ap.add_argument("--eval", type=int, default=-1, help="which U-Net evaluation (default: the middle one)")
  ap.add_argument("--csv", default=None)
  args = ap.parse_args()
  rows = list(csv.DictReader(open(args.trace_csv)))
  key = lambda *names: next(n for n in names if n in rows[0])
  kn, ks, ke = key("Kernel_Name", "Name"), key("Start_Timestamp", "Start"), key("End_Timestamp", "End")
  gx = next((n for n in ("Grid_Size_X", "Grid_Size") if n in rows[0]), None)
  wx = next((n for n in ("Workgroup_Size_X", "Workgroup_Size") if n in rows[0]), None)
  rows.sort(key=lambda r: int(r[ks]))
  marks = [i for i, r in enumerate(rows) if "time_embedding_kernel" in r[kn] or "select_row_kernel" in r[kn]]
  if len(marks) < 2:
    sys.exit("fewer than two U-Net evaluations in the trace")
  k = args.eval if args.eval >= 0 else len(marks) // 2
  k = min(k, len(marks) - 2)
  seg = rows[marks[k]:marks[k + 1]]
  t0 = int(seg[0][ks])
  prev_end = t0
  tot = gap_tot = 0.0
  out = []
  for i, r in enumerate(seg):
    s, e = int(r[ks]), int(r[ke])
    dur, gap = (e - s) / 1e3, (s - prev_end) / 1e3
    prev_end = max(prev_end, e)
    tot += dur
    gap_tot += max(gap, 0.0)
    wg = int(r[wx]) if wx else 0
    grid = int(r[gx]) // max(wg, 1) if gx else 0
    out.append((i, short(r[kn]), grid, wg, dur, gap, (s - t0) / 1e3))
  span = (prev_end - t0) / 1e3
  print(f"# U-Net evaluation {k} of {len(marks)}: {len(seg)} dispatches, kernel time {tot:.1f} us, "
        f"gaps {gap_tot:.1f} us, span {span:.1f} us")
  print(f"# {'idx':>3} {'t_us':>8} {'dur_us':>8} {'gap_us':>7} {'wgs':>6} {'wgsz':>5}  kernel")
  for i, name, grid, wg, dur, gap, t in out:
    print(f"{i:5d} {t:8.1f} {dur:8.1f} {gap:7.2f} {grid:6d} {wg:5d}  {name}")
  if args.csv:
    with open(args.csv, "w", newline="") as f:
      w = csv.writer(f)
      w.writerow(["idx", "kernel", "workgroups", "wg_size", "dur_us", "gap_us", "t_us"])
      for rec in out:
        w.writerow([rec[0], rec[1], rec[2], rec[3], f"{rec[4]:.2f}", f"{rec[5]:.2f}", f"{rec[6]:.1f}"])


if __name__ == "__main__":
  main()
