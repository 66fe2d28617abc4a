#!/bin/bash
# Dev aid: rocprofv3 duration of k_dir_order for library builds (-DGAS_DIRSORT_ABL=1|2 variants; any ablation switch
# added to a kernel must keep every address in range).  GPU box, repo root.  Measured: full 24.1 us, synthetic keys
# 14.7 us (the single-CU gather of 8192 parameter lines costs ~9 us), no rank loop 9.4 us (the ballot match ~14 us).
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset GAS_AMD_LIB; else export GAS_AMD_LIB=$GRAFT_REPO_ROOT/$lib; fi
  rm -rf /tmp/ds_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ds_prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-max-sources --direction-order --no-pipelined-mix > /tmp/ds.log 2>&1
  python3 - "$lib" <<PY
import csv,glob,sys
f=glob.glob("/tmp/ds_prof/**/t_kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_dir_order" in r["Name"] or "k_hrtf_ols" in r["Name"]:
        print("%-28s %-12s calls %4s avg %8.2f us" % (sys.argv[1], r["Name"].split("::")[-1][:12], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
