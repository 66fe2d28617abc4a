#!/bin/bash
# SQ stall breakdown of the bench's kernels (one PMC pass, 8 SQ counters).
set -e
TAG=${1:-sq}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT -o sq -- python3 $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-max-sources "$@" > $OUT/run.log 2>&1
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    a = agg[r["Kernel_Name"][:60]][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in agg.items():
    if "gas" not in k and "k_" not in k: continue
    print(k)
    for c, (s, n) in sorted(d.items()):
        print(f"   {c:24s} {s/n:16.1f}")
PY
