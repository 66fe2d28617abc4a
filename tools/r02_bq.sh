set -e
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/gputests.log 2>&1 || { tail -40 gpurun_out/r02/gputests.log; exit 1; }
tail -3 gpurun_out/r02/gputests.log
for pipe in 0 1; do
GAS_BIQUAD_PIPE=$pipe python bench.py --workload biquad --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipe=$pipe  %.2f us/step  kernel %s %.2f us' % (1e3*d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_us']))"
GAS_BIQUAD_PIPE=$pipe python bench.py --workload biquad --sources-per-gpu 4096 --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipe=$pipe n=4096 %.2f us/step  kernel %s %.2f us' % (1e3*d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_us']))"
GAS_BIQUAD_PIPE=$pipe python bench.py --workload biquad --sources-per-gpu 8192 --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipe=$pipe n=8192 %.2f us/step  kernel %s %.2f us' % (1e3*d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_us']))"
done
