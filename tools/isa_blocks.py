#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -S listing, per basic block.

Usage: tools/isa_blocks.py <listing.s> <kernel-substring> [--min N]
Prints for every basic block of the first kernel whose mangled name contains the substring: label, line range, and the
number of VALU (v_*, excluding v_readlane/v_readfirstlane which issue on the VALU but are counted apart), cross-lane
(v_readlane, v_readfirstlane, v_writelane, DPP forms), LDS (ds_*), vector-memory (global_/buffer_/flat_/scratch_),
scalar (s_*) and wait (s_waitcnt / s_nop / s_sleep / s_barrier) instructions, plus the blocks' branch targets.
The committed summaries under profiles/ (rNN_isa_*.txt) are this tool's output; the VALU floor quoted in DESIGN.md is
the per-source VALU count (the blocks the source loop executes once per source) x 4 cycles.
"""
import re
import sys


def classify(op):
    if op in ("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio", "s_waitcnt_vscnt", "s_waitcnt_depctr"):
        return "wait"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")) or "_dpp" in op:
        return "xlane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, sub = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 0
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*:", l) and sub in l.split(":")[0]:
            start = i
            break
    if start is None:
        raise SystemExit(f"no kernel matching {sub!r}")
    print("kernel", lines[start].split(":")[0])
    blocks = []
    cur = {"label": "entry", "first": start + 1, "n": {}, "targets": [], "dpp": 0}
    for i in range(start + 1, len(lines)):
        l = lines[i]
        if l.startswith(".Lfunc_end") or l.strip().startswith(".section"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur["last"] = i
            blocks.append(cur)
            cur = {"label": m.group(1), "first": i + 1, "n": {}, "targets": [], "dpp": 0}
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = classify(op)
        if c == "valu" and ("row_shr" in s or "row_bcast" in s or "quad_perm" in s or "row_shl" in s):
            c = "xlane"
        cur["n"][c] = cur["n"].get(c, 0) + 1
        if op.startswith(("s_cbranch", "s_branch")):
            cur["targets"].append(s.split()[-1])
    cur["last"] = i
    blocks.append(cur)
    keys = ("valu", "xlane", "lds", "vmem", "salu", "wait", "other")
    print(f"{'block':12s} {'lines':>13s} " + " ".join(f"{k:>6s}" for k in keys) + "  -> targets")
    tot = dict.fromkeys(keys, 0)
    for b in blocks:
        n = sum(b["n"].values())
        for k in keys:
            tot[k] += b["n"].get(k, 0)
        if n < min_n:
            continue
        print(f"{b['label']:12s} {b['first']:6d}-{b['last']:<6d} " + " ".join(f"{b['n'].get(k, 0):6d}" for k in keys) + "  -> " + ",".join(b["targets"]))
    print(f"{'TOTAL':12s} {'':13s} " + " ".join(f"{tot[k]:6d}" for k in keys))


if __name__ == "__main__":
    main()
