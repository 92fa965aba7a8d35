#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel family over a bench.py run and divides by the
number of U-Net evaluations (time_embedding_kernel dispatches).
usage: tools/pmc_family.py <counter_collection.csv> [...]"""
import collections
import csv
import sys

for path in sys.argv[1:]:
  rows = list(csv.DictReader(open(path)))
  evals = len({r["Dispatch_Id"] for r in rows if "time_embedding_kernel" in r["Kernel_Name"]})
  fam = collections.defaultdict(lambda: collections.defaultdict(float))
  dur = collections.defaultdict(float)
  seen = set()
  for r in rows:
    n = r["Kernel_Name"]
    f = ("gemm_kernel" if "gemm_kernel<" in n else "attn_kernel" if "attn_kernel" in n else
         "groupnorm" if "gn_" in n else "layernorm" if "layernorm" in n else "other")
    fam[f][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
      seen.add(r["Dispatch_Id"])
      dur[f] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
  print(f"{path}: {evals} U-Net evaluations")
  for f, d in fam.items():
    print(f"  {f:12s} {dur[f] / max(evals, 1):8.3f} ms/eval  " +
          "  ".join(f"{c}={v / max(evals, 1):.4e}/eval" for c, v in sorted(d.items())))
