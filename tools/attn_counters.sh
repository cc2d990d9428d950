#!/bin/bash
# Attention kernels in isolation (tools/bench_attn.py: every Swin stage geometry + the text encoder's packed rows): per-dispatch kernel trace
# and SQ counters, grouped afterwards by (kernel, grid) = (pass, stage, mode) with tools/attn_counters.py -> profiles/rNN_attn_counters.csv.
#   bash tools/attn_counters.sh [tag]     (raw output under gpurun_out/prof/attn_<tag>/)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-base}
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/prof/attn_$TAG
rm -rf $O
mkdir -p $O
export IT=3
timeout -k 10 200 python3 $R/tools/bench_attn.py > $O/timings.log 2>&1
echo "timings done"
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace -d $O/kt -- python3 $R/tools/bench_attn.py > $O/kt.log 2>&1
echo "kt done"
timeout -k 10 200 rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $O/pmc -- python3 $R/tools/bench_attn.py > $O/pmc.log 2>&1
echo "pmc done"
find $O -name "*.db" -delete 2>/dev/null || true
du -sh $O
