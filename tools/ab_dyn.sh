#!/bin/bash
# A/B of the dynamic tile walk in the whole step (same box, alternating): MVULD_GEMM_DYNAMIC_TILES = 0 | 1 (| 2 where built)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/dyn
mkdir -p $O
cd $R
Q="--no-cpu-baseline --no-kernel-timing --no-varlen --no-infer --no-fp8 --steps 40 --warmup 8"
for rep in 1 2; do
  for d in ${MODES:-0 1}; do
    MVULD_GEMM_DYNAMIC_TILES=$d timeout -k 10 200 python3 bench.py $Q > $O/bench_d${d}_$rep.json 2> $O/bench_d${d}_$rep.err || { echo "bench d=$d failed"; tail -5 $O/bench_d${d}_$rep.err; exit 1; }
    python3 -c "import json,sys; d=json.load(open('$O/bench_d${d}_$rep.json')); print('dyn=$d rep=$rep ms_per_step', d['ms_per_step'], 'value', d['value'])"
  done
done
