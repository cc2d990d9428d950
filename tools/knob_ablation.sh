#!/bin/bash
# One box, one call: the step with each round-3 (second session) saving switched off in turn, 60 timed steps each -> gpurun_out/knob_ablation.csv
# (copy to profiles/rNN_knob_ablation.csv).  Boxes differ by +-2 %, runs on one box by +-0.1 ms.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/knob_ablation.csv
run() { python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-varlen --no-infer --no-fp8 --no-kernel-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"; }
echo "setting,ms_per_step,functions_per_s" > $O
emit() { echo "$1,$(echo $2 | tr ' ' ',')" >> $O; echo "$1 $2"; }
emit "default (all on)" "$(run)"
emit "MVULD_GELU_DG=0" "$(MVULD_GELU_DG=0 run)"
emit "MVULD_DROPPATH_SKIP=0" "$(MVULD_DROPPATH_SKIP=0 run)"
emit "MVULD_ATTN_YSKIP=0" "$(MVULD_ATTN_YSKIP=0 run)"
emit "MVULD_SPLIT3_FUSED=0" "$(MVULD_SPLIT3_FUSED=0 run)"
emit "MVULD_LN_DEFER=0 MVULD_LN_DROP=0" "$(MVULD_LN_DEFER=0 MVULD_LN_DROP=0 run)"
emit "all six off" "$(MVULD_GELU_DG=0 MVULD_DROPPATH_SKIP=0 MVULD_ATTN_YSKIP=0 MVULD_SPLIT3_FUSED=0 MVULD_LN_DEFER=0 MVULD_LN_DROP=0 run)"
emit "default (all on), again" "$(run)"
