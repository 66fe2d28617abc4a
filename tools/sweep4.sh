#!/bin/bash
# dev aid: compare prebuilt library variants on a given workload: sweep4.sh "<bench args>" tag...
D=godot-audio-spatializer_amd
ARGS="$1"; shift
cp $D/libgas_amd.so /tmp/libgas_amd_keep.so
for tag in "$@"; do
  cp $D/libgas_amd_$tag.so $D/libgas_amd.so
  echo "== variant $tag"
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-max-sources $ARGS 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('ms/step %.4f  kernel_us %.2f  frac %.3f  value %.3e'%(r['ms_per_step'], r['roofline']['kernel_us'], r['roofline']['frac'], r['value']))"
done
cp /tmp/libgas_amd_keep.so $D/libgas_amd.so
