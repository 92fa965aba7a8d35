#!/bin/bash
# Profiles of one round, run ON THE GPU BOX from the repo root:
#     bash tools/profile_round.sh r03 [c4]
# 1. rocprofv3 --kernel-trace --stats of the bench command -> <out>/rNN_bench_kernel_stats.{md,csv},
#    rNN_bench_under_rocprof.json (the bench line of that same run)
# 2. three separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ counters; never combined with a
#    trace domain other than --kernel-trace) of a 10-DDIM-step pass -> rNN_pmc_families.txt,
#    rNN_pmc_traffic.json (bench.py's roofline.traffic reads the latter from profiles/)
# Raw rocprof output stays in /tmp (it exceeds what gpurun merges back); copy <out>/rNN_* to profiles/.
set -e
R=${1:-r00}
CFG=${2:-c3}          # bench.py --config; for a configuration other than c3 only the kernel trace is taken
[ "$CFG" != "c3" ] && R=${R}_${CFG}
OUT=$PWD/gpurun_out/summary_$R
RAW=/tmp/prof_$R
rm -rf "$RAW"; mkdir -p "$OUT" "$RAW"
export TMPDIR=/tmp
BENCH="bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-variant"
echo "[profile] kernel trace"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/kt" -- python3 $BENCH > "$OUT/bench_under_rocprof.log" 2>&1
grep '^{"metric"' "$OUT/bench_under_rocprof.log" > "$OUT/${R}_bench_under_rocprof.json"
STATS=$(find "$RAW/kt" -name '*kernel_stats.csv' | head -1)
python3 tools/summarize_rocprof.py "$STATS" --bench-json "$OUT/${R}_bench_under_rocprof.json" --top 40 > "$OUT/${R}_bench_kernel_stats.md"
cp "$STATS" "$OUT/${R}_bench_kernel_stats.csv"
TRACE=$(find "$RAW/kt" -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py "$TRACE" --csv "$OUT/${R}_step_timeline.csv" > "$OUT/${R}_step_timeline.txt"
if [ "$CFG" != "c3" ]; then echo "[profile] done: $(ls $OUT)"; exit 0; fi
PMCB="bench.py --config $CFG --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-variant"
CSVS=""
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  TAG=$(echo $C | cut -d' ' -f1)
  echo "[profile] pmc $C"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$RAW/pmc_$TAG" -- python3 $PMCB > "$OUT/pmc_$TAG.log" 2>&1
  CSVS="$CSVS $(find "$RAW/pmc_$TAG" -name '*counter_collection.csv' | head -1)"
done
python3 tools/pmc_family.py $CSVS --json "$OUT/${R}_pmc_traffic.json" > "$OUT/${R}_pmc_families.txt"
echo "[profile] done: $(ls $OUT)"
