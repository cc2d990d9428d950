#!/bin/bash
# One gpurun call's worth of profiles for profiles/rNN_*: kernel-trace stats of the bench step (one stream / default streams), HBM traffic
# counters (separate --pmc passes, as MI355X_MICROARCH.md prescribes), per-shape GEMM timings + counters.  Raw output under
# gpurun_out/prof/; condense on the host with tools/summarize_profile.py, tools/kernel_stats.py and tools/summarize_gemm_shapes.py.
#   bash tools/profile_round.sh [bench|shapes|all]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
WHAT=${1:-all}
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/prof
mkdir -p $O
B="$R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-varlen --no-infer --no-fp8"
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
    rm -rf $O/kt $O/kt2 $O/fetch $O/write
    MVULD_CONCURRENT=0 timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -- python3 $B > $O/kt.log 2>&1
    echo "kt done"
    timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt2 -- python3 $B > $O/kt2.log 2>&1
    echo "kt2 done"
    MVULD_CONCURRENT=0 timeout -k 10 300 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $O/fetch -- python3 $B > $O/fetch.log 2>&1
    echo "fetch done"
fi
if [ "$WHAT" = shapes ] || [ "$WHAT" = all ]; then
    rm -rf $O/gs_pmc $O/gs_traf
    G="$R/tools/gemm_shapes.py --reps 5"
    timeout -k 10 200 python3 $G --csv $O/gemm_shapes_timings.csv > $O/gs.log 2>&1
    echo "gs timings done"
    timeout -k 10 150 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $O/gs_pmc/p1 -- python3 $G > $O/gs_pmc1.log 2>&1
    echo "gs pmc1 done"
    timeout -k 10 150 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY --kernel-trace -d $O/gs_pmc/p2 -- python3 $G > $O/gs_pmc2.log 2>&1
    echo "gs pmc2 done"
    timeout -k 10 150 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $O/gs_traf/p1 -- python3 $G > $O/gs_traf1.log 2>&1
    echo "gs fetch done"
    timeout -k 10 150 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $O/gs_traf/p2 -- python3 $G > $O/gs_traf2.log 2>&1
    echo "gs write done"
fi
if [ "$WHAT" = write ] || [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
    # (a WRITE_SIZE pass of this command once sat idle after start-up until its timeout; it runs last and on a short leash)
    rm -rf $O/write
    MVULD_CONCURRENT=0 timeout -k 10 150 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $O/write -- python3 $B > $O/write.log 2>&1
    echo "write done"
fi
# keep what travels back small: stats and counter CSVs only (the per-dispatch kernel traces of the bench runs are tens of MB)
find $O -name "*.db" -delete 2>/dev/null || true
find $O/kt $O/kt2 -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null || true
du -sh $O
