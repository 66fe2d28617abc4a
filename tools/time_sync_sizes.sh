#!/bin/bash
# Dev aid (GPU box): the synchronous [HRTF] callback over a ladder of sizes, with the history rows non-temporal from
# GAS_NT_HIST_MIN sources on (0 = always, 4294967295 = never).  Usage: tools/time_sync_sizes.sh "<sizes>" "<mins>"
for n in $1; do
  for mn in $2; do
    GAS_NT_HIST_MIN=$mn python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-pipelined-mix --sources-per-gpu $n --marked-callbacks 32 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('n %8d  nt_hist_min %10s  %9.2f us/step  kernel %9.2f us  frac %.3f' % (int(sys.argv[1]), sys.argv[2], 1e3*d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['frac']))" $n $mn
  done
done
