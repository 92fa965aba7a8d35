#!/bin/bash
# Kernel traces of the two-branch step and of its single-branch references (run on the GPU box):
#     bash tools/lane_trace.sh <out-dir>
set -e
OUT=${1:-gpurun_out/lanes}
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {   # tag, batch, variant spec
  rm -rf /tmp/lt_$1
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt_$1 -- python3 tools/ab_step.py --batch $2 --rounds 1 --reps 3 $3 > "$OUT/$1.log" 2>&1
  python3 tools/lane_overlap.py "$(find /tmp/lt_$1 -name '*kernel_trace.csv' | head -1)" --show ${4:-0} > "$OUT/$1_overlap.txt"
}
run lanes2_b16 16 lanes2=lanes:2 80
run single_b16 16 base=
run single_b8 8 base=
ls "$OUT"
