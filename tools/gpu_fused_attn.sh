#!/bin/bash
# development loop of the fused window backward: parity tests, then per-stage timings fused vs three-pass
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/fused
mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests/test_gpu_kernels.py -q -k "fused_window_backward or swin_window_attention or vertical_mask or droppath_dropped" > $O/tests.txt 2>&1
echo "tests rc=$?"; tail -5 $O/tests.txt
CASES=swin IT=10 timeout -k 10 150 python3 tools/bench_attn.py auto > $O/attn_fused.txt 2>&1; echo "bench fused rc=$?"
CASES=swin IT=10 MVULD_ATTN_BWD_FUSED=0 timeout -k 10 150 python3 tools/bench_attn.py auto > $O/attn_3pass.txt 2>&1; echo "bench 3pass rc=$?"
cut -c1-80 $O/attn_fused.txt $O/attn_3pass.txt
