"""Condenses a tools/profile_bench.sh output directory into the text summary committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def pmc_avg(out, name, counter, kernel_substr):
    cc = find(os.path.join(out, name), "*counter_collection.csv")
    if not cc:
        return None
    tot, n = 0.0, 0
    with open(cc) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]:
                tot += float(r["Counter_Value"])
                n += 1
    return tot / n if n else None


def write_pmc_json(out, path):
    """HBM traffic per launch of the bench's dominant kernel, corrected as MI355X_MICROARCH.md prescribes:
    FETCH_SIZE reads exactly 1/2 of streamed bytes on gfx950 (re-verified for 4/8/16 B loads with
    tools/micro/fetch_calib.hip), WRITE_SIZE is exact; both counters are in KiB; separate passes."""
    import json

    line = None
    for l in open(os.path.join(out, "trace.log")):
        if l.startswith("{"):
            line = json.loads(l)
    if not line:
        return
    kern = line["roofline"]["kernel"].split("<")[0]
    fetch = pmc_avg(out, "fetch", "FETCH_SIZE", kern)
    write = pmc_avg(out, "write", "WRITE_SIZE", kern)
    if fetch is None or write is None:
        return
    rec = {
        "kernel": line["roofline"]["kernel"],
        "workload": line["config"]["workload"],
        "sources_per_gpu": line["config"]["sources_per_gpu"],
        "peaks": line["config"].get("peaks"),
        "pipelined_mix": bool(line["config"].get("pipelined_mix", False)),
        "experiment": line["config"].get("experiment", ""),
        "lib_sha16": line["config"].get("lib_sha16"),
        "callbacks_per_launch": line["roofline"].get("callbacks_per_launch", 1),
        "fetch_size_kib_raw": fetch,
        "write_size_kib_raw": write,
        "fetch_correction": 2.0,
        "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
        "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"],
    }
    with open(path, "w") as f:
        json.dump(rec, f, indent=1)


def main(out):
    stats = find(os.path.join(out, "trace"), "*kernel_stats.csv")
    print("== rocprofv3 --kernel-trace --stats (kernel_stats.csv) ==")
    if stats:
        with open(stats) as f:
            rows = list(csv.DictReader(f))
        print(f"{'kernel':70s} {'calls':>7s} {'avg_ns':>12s} {'min_ns':>10s} {'max_ns':>10s} {'pct':>7s}")
        for r in rows[:12]:
            print(f"{r['Name'][:70]:70s} {r['Calls']:>7s} {float(r['AverageNs']):12.1f} {r['MinNs']:>10s} {r['MaxNs']:>10s} {float(r['Percentage']):7.2f}")
    for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        cc = find(os.path.join(out, name), "*counter_collection.csv")
        print(f"== rocprofv3 --pmc {counter} (per-dispatch average, raw counter units = KiB) ==")
        if not cc:
            print("missing")
            continue
        agg = defaultdict(lambda: [0.0, 0])
        with open(cc) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != counter:
                    continue
                a = agg[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
        for k, (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
            print(f"{k[:70]:70s} dispatches={n:6d} avg={s / n:14.1f} KiB")


def sq_summary(out, sub="sq", title="SQ_* (per-dispatch average)"):
    """Per-dispatch averages of an SQ pass for the library's kernels (quad-cycle units, MI355X_MICROARCH.md)."""
    cc = find(os.path.join(out, sub), "*counter_collection.csv")
    print(f"== rocprofv3 --pmc {title} ==")
    if not cc:
        print("missing")
        return
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(cc) as f:
        for r in csv.DictReader(f):
            a = agg[r["Kernel_Name"][:70]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    for k, d in agg.items():
        if "k_" not in k:
            continue
        print(k)
        for c, (s, n) in sorted(d.items()):
            print(f"   {c:24s} {s / n:16.1f}   ({n} dispatches)")


def bench_line(out):
    """The JSON line bench.py printed under the tracer (its event-timed kernel_us next to the trace's average)."""
    import json

    for l in open(os.path.join(out, "trace.log")):
        if l.startswith("{"):
            d = json.loads(l)
            rf = d["roofline"]
            print("== bench.py line of the traced run ==")
            print(f"ms_per_step {d['ms_per_step']:.5f}  value {d['value']:.4e}  kernel {rf['kernel']} kernel_us(events) {rf['kernel_us']:.2f}  algorithmic bytes {rf['algorithmic_bytes_per_launch']}  frac {rf['frac']:.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
    sq_summary(sys.argv[1])
    sq_summary(sys.argv[1], "insts", "SQ_INSTS_* (wave-level instructions per dispatch, average)")
    bench_line(sys.argv[1])
    if len(sys.argv) > 2:
        write_pmc_json(sys.argv[1], sys.argv[2])
