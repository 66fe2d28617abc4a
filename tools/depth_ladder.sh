for d in 8 10 12 16; do
  for rep in 1 2; do
  python3 bench.py --steps 320 --warmup 32 --no-cpu-baseline --no-extras --batch-depth $d --reduce-bucket 32 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('depth %2d  %7.2f us/step  kernel %7.2f us/launch  frac %.3f' % (int(sys.argv[1]), 1e3*d['ms_per_step'], r['kernel_us'], r['frac']))" $d
  done
done
