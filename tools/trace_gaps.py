"""Dev aid: idle gaps between consecutive kernels in a rocprofv3 --kernel-trace CSV (where does a step's time go?)."""
import csv
import re
import sys
from collections import defaultdict

def short(name):
    m = re.search(r"(k_\w+|\w+)\s*[<(]", name.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else name[:24]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 3  # ignore warm-up
rows = rows[skip:]
dur = defaultdict(list)
gap_after = defaultdict(list)
for a, b in zip(rows, rows[1:]):
    na = short(a["Kernel_Name"])
    dur[na].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap_after[na + " -> " + short(b["Kernel_Name"])].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"kernels {len(rows)}  span {span / 1e3:.1f} us  busy {busy / 1e3:.1f} us ({100 * busy / span:.1f} %)")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  dur  {k:50s} n={len(v):5d} avg {sum(v) / len(v) / 1e3:7.2f} us")
for k, v in sorted(gap_after.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"  gap  {k:50s} n={len(v):5d} avg {sum(v) / len(v) / 1e3:7.2f} us")
