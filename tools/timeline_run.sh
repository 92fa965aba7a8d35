#!/bin/bash
# Per-launch in-situ timeline of one U-Net evaluation, run ON THE GPU BOX from the repo root:
#     bash tools/timeline_run.sh <tag> [bench config, default c3]
# -> gpurun_out/timeline_<tag>_<cfg>.txt / .csv (tools/step_timeline.py)
set -e
TAG=${1:-t}
CFG=${2:-c3}
RAW=/tmp/tl_${TAG}_${CFG}
rm -rf "$RAW"; mkdir -p "$RAW" gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$RAW" -- python3 bench.py --config $CFG --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-variant > gpurun_out/timeline_${TAG}_${CFG}.log 2>&1
TR=$(find "$RAW" -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py "$TR" --csv gpurun_out/timeline_${TAG}_${CFG}.csv > gpurun_out/timeline_${TAG}_${CFG}.txt
head -1 gpurun_out/timeline_${TAG}_${CFG}.txt
