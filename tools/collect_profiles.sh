#!/bin/bash
# Copies what tools/profile_bench.sh left under gpurun_out/prof_<tag>/ into profiles/ (the committed, judged copies):
# <tag>_summary.txt, <tag>_kernel_stats.csv, <tag>_pmc.json, <tag>_bench_line_under_rocprof.json
set -e
for d in gpurun_out/prof_*; do
  tag=${d#gpurun_out/prof_}
  [ -f $d/summary.txt ] || continue
  cp $d/summary.txt profiles/${tag}_summary.txt
  s=$(find $d/trace -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp $s profiles/${tag}_kernel_stats.csv
  [ -f $d/pmc.json ] && cp $d/pmc.json profiles/${tag}_pmc.json
  grep '^{' $d/trace.log | tail -1 > profiles/${tag}_bench_line_under_rocprof.json
done
ls profiles | grep -c r02
