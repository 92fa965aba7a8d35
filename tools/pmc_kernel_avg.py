#!/usr/bin/env python3
"""Averages every counter of a rocprofv3 counter_collection.csv per kernel name (substring filter)."""
import collections
import csv
import sys

path, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(path)):
  k = r["Kernel_Name"]
  if filt not in k:
    continue
  acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
  n[k].add(r["Dispatch_Id"])
for k, d in acc.items():
  c = len(n[k])
  print(k[:90], f"({c} dispatches)")
  for name, v in sorted(d.items()):
    print(f"   {name:32s} {v / c:16.4e}")
