#!/bin/bash
# timing experiments of the fused window backward (-DAF_X=n builds under build_variants/: wrong results, right timings)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/fused
cd $R
export MVULD_ATTN_BWD_FUSED=1 IT=10 CASES="swin s"
for v in base ${AF_VARIANTS:-1 2 3 4}; do
  lib=$R/mvuld_amd/libmvuld_hip.so; [ $v != base ] && lib=$R/build_variants/libmvuld_afx$v.so
  echo "== variant $v"
  MVULD_HIP_LIB=$lib timeout -k 10 120 python3 tools/bench_attn.py auto 2>&1 | grep -v amdgpu.ids | cut -c1-64
done
