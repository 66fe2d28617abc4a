#!/bin/bash
# Dev aid: A/B kernel builds on the GPU box.  Usage: tools/ab_libs.sh "<bench args>" lib1.so lib2.so ...
# ("-" = the in-tree library).  Prints us/step and the event-timed dominant kernel for each.
ARGS=$1; shift
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset GAS_AMD_LIB; else export GAS_AMD_LIB=$PWD/$lib; fi
  python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras $ARGS 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-32s %8.2f us/step  kernel %7.2f us' % (sys.argv[1], 1e3*d['ms_per_step'], d['roofline']['kernel_us']))" "$lib"
done
