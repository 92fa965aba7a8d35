#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel family over a bench.py run and divides by the
number of U-Net evaluations (time_embedding_kernel dispatches).

usage: tools/pmc_family.py <counter_collection.csv> [...] [--json profiles/rNN_pmc_traffic.json]

With --json, FETCH_SIZE / WRITE_SIZE (KB, from separate passes) of the MFMA GEMM/conv family
are converted to HBM-side bytes per U-Net evaluation as MI355X_MICROARCH.md prescribes
(gfx950: FETCH_SIZE tallies 128-byte requests at 64 B -> x2; WRITE_SIZE is exact) and written
for bench.py's `roofline.traffic`.
"""
import collections
import csv
import json
import sys

args = sys.argv[1:]
out_json = None
if "--json" in args:
  i = args.index("--json")
  out_json = args[i + 1]
  del args[i:i + 2]

totals = {}
for path in args:
  rows = list(csv.DictReader(open(path)))
  fam = collections.defaultdict(lambda: collections.defaultdict(float))
  dur = collections.defaultdict(float)
  cnt = collections.defaultdict(int)
  # only dispatches INSIDE a complete U-Net evaluation count: from its time_embedding_kernel to
  # its cfg_ddim_kernel, and only evaluations that contain the GEMM family (bench.py also
  # replays a step captured WITHOUT it); the text encoder's and decoder's launches are left out
  rows.sort(key=lambda r: int(r["Dispatch_Id"]))
  evals = 0
  win = None
  for r in rows:
    n = r["Kernel_Name"]
    if "time_embedding_kernel" in n or "select_row_kernel" in n:   # first launch of an evaluation (per-step MLP / row of the per-loop temb table)
      win = {"fam": collections.defaultdict(lambda: collections.defaultdict(float)),
             "dur": collections.defaultdict(float), "cnt": collections.defaultdict(int), "seen": set()}
    if win is None:
      continue
    # gn_fused_kernel<..., true> = ldm_groupnorm_splitk: the split-K reduce of a conv fused into the
    # GroupNorm that consumes it -- counted with the GEMM family (it reads that family's f32 slabs)
    f = ("gemm_kernel" if ("gemm_kernel<" in n or "gemm3_kernel<" in n or "st_tail_kernel<" in n) else "attn_kernel" if "attn_kernel" in n else
         "groupnorm_splitk" if ("gn_fused_kernel<" in n and ", true>" in n) else
         "groupnorm" if "gn_" in n else "layernorm" if "layernorm" in n else
         "splitk_reduce" if "splitk" in n else "other")
    win["fam"][f][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in win["seen"]:
      win["seen"].add(r["Dispatch_Id"])
      win["dur"][f] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
      win["cnt"][f] += 1
    if "cfg_ddim_kernel" in n:
      if win["cnt"].get("gemm_kernel", 0) > 0:
        evals += 1
        for f2, d in win["fam"].items():
          for c, v in d.items():
            fam[f2][c] += v
        for f2, v in win["dur"].items():
          dur[f2] += v
        for f2, v in win["cnt"].items():
          cnt[f2] += v
      win = None
  print(f"{path}: {evals} U-Net evaluations")
  for f, d in fam.items():
    print(f"  {f:14s} {dur[f] / max(evals, 1):8.3f} ms/eval {cnt[f] / max(evals, 1):7.1f} launches/eval  " +
          "  ".join(f"{c}={v / max(evals, 1):.4e}/eval" for c, v in sorted(d.items())))
    for c, v in d.items():
      totals.setdefault(f, {})[c] = v / max(evals, 1)
  totals.setdefault("_launches_per_eval", {}).update({f: cnt[f] / max(evals, 1) for f in cnt})

if out_json:
  # the family bench.py's roofline names = every launch ldm_gemm makes: the GEMM / conv kernels AND
  # their split-K reduce launches
  g = dict(totals.get("gemm_kernel", {}))
  for extra in ("splitk_reduce", "groupnorm_splitk"):
    for c, v in totals.get(extra, {}).items():
      g[c] = g.get(c, 0.0) + v
  res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py; "
                   "family = gemm_kernel<...> + gemm3_kernel<...> + st_tail_kernel<...> (row-panel chains) + splitk_epilogue* + the GroupNorm launches that complete a deferred split-K product (gn_fused_kernel<..., true>: their own GroupNorm read/write is included, an over-count); per U-Net evaluation (dispatches between time_embedding_kernel and cfg_ddim_kernel only)",
         "fetch_kb_raw_per_eval": g.get("FETCH_SIZE"), "write_kb_raw_per_eval": g.get("WRITE_SIZE"),
         "launches_per_eval": (totals.get("_launches_per_eval", {}).get("gemm_kernel", 0) +
                               totals.get("_launches_per_eval", {}).get("splitk_reduce", 0) +
                               totals.get("_launches_per_eval", {}).get("groupnorm_splitk", 0))}
  if g.get("FETCH_SIZE") is not None and g.get("WRITE_SIZE") is not None:
    res["hbm_read_bytes_per_eval"] = g["FETCH_SIZE"] * 1024.0 * 2.0     # gfx950 correction
    res["hbm_write_bytes_per_eval"] = g["WRITE_SIZE"] * 1024.0
    res["hbm_bytes_per_eval"] = res["hbm_read_bytes_per_eval"] + res["hbm_write_bytes_per_eval"]
  for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_LDS_BANK_CONFLICT",
            "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"):
    if k in g:
      res[k + "_per_eval"] = g[k]
  json.dump(res, open(out_json, "w"), indent=1)
  print("wrote", out_json)
