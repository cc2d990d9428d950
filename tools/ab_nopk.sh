#!/bin/bash
# A/B of packed-f32 VALU (v_pk_*_f32) vs single-issue scalar f32 beside MFMAs (MI355X_MICROARCH.md, per-instruction constants:
# "packed f32 VALU: ... an anti-lever beside MFMAs").  Variant libraries: -Xclang -target-feature -Xclang -packed-fp32-ops.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_nopk
mkdir -p $O
cd $R
tools/microbench/dpp_wave_shift > $O/dpp.txt 2>&1 || echo "dpp probe failed"
echo "dpp done"
for rep in 1 2; do
  IT=10 timeout -k 10 150 python3 tools/bench_attn.py auto > $O/attn_base_$rep.txt 2>&1
  IT=10 MVULD_HIP_LIB=$R/build_variants/libmvuld_attn_nopk.so timeout -k 10 150 python3 tools/bench_attn.py auto > $O/attn_nopk_$rep.txt 2>&1
  echo "attn rep $rep done"
done
MVULD_HIP_LIB=$R/build_variants/libmvuld_all_nopk.so timeout -k 10 400 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "attention or gemm_p256 or gelu or mlp_panel" > $O/parity_nopk.txt 2>&1
echo "parity done"
for rep in 1 2; do
  timeout -k 10 200 python3 tools/gemm_shapes.py --reps 5 --only nt --csv $O/gs_base_$rep.csv > $O/gs_base_$rep.log 2>&1
  MVULD_HIP_LIB=$R/build_variants/libmvuld_all_nopk.so timeout -k 10 200 python3 tools/gemm_shapes.py --reps 5 --only nt --csv $O/gs_nopk_$rep.csv > $O/gs_nopk_$rep.log 2>&1
  echo "gs rep $rep done"
done
