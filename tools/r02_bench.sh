set -e
mkdir -p gpurun_out/r02
python bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_20.json 2> gpurun_out/r02/bench_20.err
python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > gpurun_out/r02/bench_200.json 2>> gpurun_out/r02/bench_20.err
python bench.py --steps 2000 --warmup 20 --no-extras --no-cpu-baseline > gpurun_out/r02/bench_2000.json 2>> gpurun_out/r02/bench_20.err
python3 - <<'PY'
import json
for f in ("20","200","2000"):
    d=json.loads(open(f"gpurun_out/r02/bench_{f}.json").read().strip().splitlines()[-1])
    rf=d["roofline"]
    print(f, "ms/step %.5f wall %.5f value %.4e kernel_us %.2f frac %.3f" % (d["ms_per_step"], d["wall_ms_per_step"], d["value"], rf["kernel_us"], rf["frac"]), {k:rf.get(k) for k in ("copy_us","copy_peak","frac_of_copy")})
    for k in ("ordered","exact_peaks","latency","max_sources_under_10ms","extras_error"):
        if k in d: print("   ",k,d[k])
    if "cpu_baseline" in d: print("    cpu", d["cpu_baseline"]["value"])
PY
