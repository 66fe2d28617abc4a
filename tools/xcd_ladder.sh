for n in 65536 262144 1048576; do
  for flag in "" "--xcd-order"; do
    python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-pipelined-mix --sources-per-gpu $n --marked-callbacks 32 $flag 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('n %8d %-12s %9.2f us/step  kernel %9.2f us' % (int(sys.argv[1]), sys.argv[2] if len(sys.argv)>2 else '-', 1e3*d['ms_per_step'], d['roofline']['kernel_us']))" $n $flag
  done
done
