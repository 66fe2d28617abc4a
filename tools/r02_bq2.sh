set -e
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "biquad or mix_channel or process_frames" 2>&1 | tail -2
GAS_AMD_LIB=$PWD/build/variants/libgas_stamps.so python tools/pipe_busy_probe.py 2>&1 | grep -v amdgpu
for pipe in 0 1; do
GAS_BIQUAD_PIPE=$pipe python bench.py --workload biquad --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipe=$pipe  %.2f us/step  kernel %s %.2f us' % (1e3*d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_us']))"
done
