#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's dominant kernel: one kernel-trace/stats run, two separate PMC passes
# for the memory-side traffic (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md PMC slots) and one
# SQ pass (wave cycles, waits, VALU / LDS activity, bank conflicts) and one SQ_INSTS pass (wave-level instruction counts: the
# VALU floor of bench.py's roofline).  PMC passes carry --kernel-trace only.
# Usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag> [bench args...]
set -e
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 200 --warmup 20 --no-cpu-baseline --no-extras $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $REPO/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $REPO/bench.py $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq -o sq -- python3 $REPO/bench.py $ARGS > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $OUT/insts -o insts -- python3 $REPO/bench.py $ARGS > $OUT/insts.log 2>&1
cd $REPO
python3 tools/summarize_profile.py $OUT $OUT/pmc.json > $OUT/summary.txt
cat $OUT/summary.txt
