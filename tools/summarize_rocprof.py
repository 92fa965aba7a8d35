#!/usr/bin/env python3
"""Condenses a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py into
a small markdown summary for profiles/: per-kernel totals, and the MFMA GEMM/conv
family's time per U-Net evaluation (the number bench.py's `roofline` object reports).

    python tools/summarize_rocprof.py gpurun_out/prof/<host>/<pid>_kernel_stats.csv \
        [--bench-json gpurun_out/bench.log] > profiles/rNN_bench_kernel_stats.md
"""
import argparse
import csv
import json
import re


def short(name):
  name = name.replace("(anonymous namespace)::", "").replace("void ", "")
  name = re.sub(r"\(.*$", "", name)
  return name[:72]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("stats_csv")
  ap.add_argument("--bench-json", default=None)
  ap.add_argument("--top", type=int, default=24)
  args = ap.parse_args()
  rows = list(csv.DictReader(open(args.stats_csv)))
  tot = sum(float(r["TotalDurationNs"]) for r in rows)
  # one cfg_ddim_kernel per U-Net evaluation (the step's first launch is select_row_kernel when the per-loop temb table
  # is in use, time_embedding_kernel otherwise; the CFG + DDIM update ends every evaluation of bench.py)
  evals = sum(int(r["Calls"]) for r in rows if "cfg_ddim_kernel" in r["Name"])
  gemm = [r for r in rows if "gemm_kernel<" in r["Name"] or "gemm3_kernel<" in r["Name"] or "st_tail_kernel<" in r["Name"]]
  # split-K reduces: the plain reduce launches and the GroupNorm launches that complete a deferred product
  red = [r for r in rows if "splitk_epilogue" in r["Name"] or ("gn_fused_kernel<" in r["Name"] and ", true>" in r["Name"])]
  red_ns = sum(float(r["TotalDurationNs"]) for r in red)
  gemm_ns = sum(float(r["TotalDurationNs"]) for r in gemm)
  gemm_calls = sum(int(r["Calls"]) for r in gemm)
  print("# rocprofv3 --kernel-trace --stats summary\n")
  print(f"* total kernel time: {tot / 1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} dispatches")
  print(f"* U-Net evaluations in the run (cfg_ddim_kernel dispatches): {evals}")
  if evals:
    print(f"* MFMA GEMM/conv family (`gemm_kernel<...>` / `gemm3_kernel<...>` / `st_tail_kernel<...>`, all tile shapes): {gemm_ns / 1e6:.1f} ms, "
          f"{gemm_calls} launches = **{gemm_ns / 1e6 / evals:.3f} ms per U-Net evaluation** "
          f"(incl. the text encoder's and decoder's launches, which add a few %), "
          f"average launch {gemm_ns / 1e3 / max(gemm_calls, 1):.1f} us")
    print(f"* the family as bench.py's `roofline` defines it (every launch made by `ldm_gemm`: `gemm_kernel<...>` / `gemm3_kernel<...>`, the row-panel chains `st_tail_kernel<...>` "
          f"+ the split-K reduces `splitk_epilogue*` / `gn_fused_kernel<..., true>` = reduce fused with the next GroupNorm): **{(gemm_ns + red_ns) / 1e6 / evals:.3f} ms per U-Net evaluation**, "
          f"{(gemm_calls + sum(int(r['Calls']) for r in red)) / evals:.0f} kernel launches per evaluation")
    print(f"* all kernels: {tot / 1e6 / evals:.3f} ms per U-Net evaluation (upper bound: includes text encoder + decoder)")
  if args.bench_json:
    for line in open(args.bench_json):
      line = line.strip()
      if line.startswith("{") and '"metric"' in line:
        b = json.loads(line)
        r = b.get("roofline") or {}
        print(f"* bench.py line of the same command: value {b['value']:.3f} {b['unit']}, "
              f"{b.get('ms_per_unet_step', 0):.3f} ms per U-Net step (HIP events around the graph replays), "
              f"ldm_gemm family {r.get('ms_per_unet_step_in_kernel', 0):.3f} ms per step by HIP events (graph-replay difference) "
              f"-> {r.get('achieved', 0):.0f} TFLOP/s = {100 * r.get('frac', 0):.1f}% of {r.get('peak')} TFLOP/s")
  print("\n| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|")
  for r in rows[:args.top]:
    print(f"| `{short(r['Name'])}` | {int(r['Calls'])} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
          f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")


if __name__ == "__main__":
  main()
