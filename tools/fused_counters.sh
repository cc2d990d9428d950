#!/bin/bash
# SQ counters of the fused window backward on the stage-2 geometry (tools/bench_attn.py CASES="swin s2")
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/prof/fusedc
rm -rf $O; mkdir -p $O
export IT=3 CASES="swin s2" MVULD_ATTN_BWD_FUSED=1
timeout -k 10 150 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace -d $O/p1 -- python3 $R/tools/bench_attn.py > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout -k 10 150 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace -d $O/p2 -- python3 $R/tools/bench_attn.py > $O/p2.log 2>&1; echo "p2 rc=$?"
find $O -name "*.db" -delete 2>/dev/null
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/prof/fusedc"
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for p in glob.glob(O+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"][:48]
        if "attn" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k in acc:
    print(k)
    for c in sorted(acc[k]): print(f"   {c:28s} {acc[k][c]/max(1,n[k][c]):16.0f}  (n={n[k][c]})")
PY
