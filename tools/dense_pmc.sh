#!/bin/bash
# PMC anatomy of the C x C Dense launch (run on the GPU box): bash tools/dense_pmc.sh <out-file> [tile] [probe args]
set -e
OUT=${1:-gpurun_out/dense_pmc.txt}
TILE=${2:-14}
shift 2 || true
export TMPDIR=/tmp
: > "$OUT"
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  TAG=$(echo $C | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/dp_$TAG
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/dp_$TAG -- python3 tools/dense_probe.py pmc $TILE "$@" > /tmp/dp_$TAG.log 2>&1 || { echo "pass $C failed" >> "$OUT"; tail -3 /tmp/dp_$TAG.log >> "$OUT"; continue; }
  python3 tools/pmc_kernel_avg.py "$(find /tmp/dp_$TAG -name '*counter_collection.csv' | head -1)" gemm >> "$OUT"
done
cat "$OUT"
