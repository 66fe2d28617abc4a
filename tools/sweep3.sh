#!/bin/bash
# dev aid: bench with different extra args
for a in "$@"; do
  echo "== $a"
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-max-sources $a 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('ms/step %.4f  kernel_us %.2f  frac %.3f  value %.3e'%(r['ms_per_step'], r['roofline']['kernel_us'], r['roofline']['frac'], r['value']))"
done
