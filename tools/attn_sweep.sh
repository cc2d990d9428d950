#!/bin/bash
# Sweep an environment knob of the attention launchers over values on chosen bench_attn cases: bash tools/attn_sweep.sh VAR "v1 v2 .." ["case filter"]
VAR=$1; VALS=$2; export CASES=${3:-swin s2}; export IT=${IT:-10}
for v in $VALS; do
  echo "== $VAR=$v"
  env $VAR=$v python3 tools/bench_attn.py 2>&1 | grep -v amdgpu.ids
done
