"""Condenses a tools/profile_bench.sh output directory into the text summary committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


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


if __name__ == "__main__":
    main(sys.argv[1])
