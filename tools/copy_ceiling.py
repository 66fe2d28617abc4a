#!/usr/bin/env python3
"""Development aid: the copy-bandwidth ceiling of a launch the size of k_hrtf_ols' (gas_bandwidth_probe), swept
over launch geometry.  bench.py reports the best geometry's figure as roofline.copy_peak."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    import godot_audio_spatializer_amd as gas

    ctx = gas.SpatializerContext(max_sources=64, frames=512)
    # (label, read bytes, written bytes): source rows + history read + HRIR taps | history written + partial mixes
    cases = [
        ("hrtf 8192 src", 8192 * (4096 + 1024) + (2 << 20), 8192 * 1024 + (1 << 20)),
        ("hrtf 4096 src", 4096 * (4096 + 1024) + (2 << 20), 4096 * 1024 + (1 << 19)),
        ("hrtf 65536 src", 65536 * (4096 + 1024) + (2 << 20), 65536 * 1024 + (1 << 20)),
        ("read only 54 MB", 54 << 20, 0),
    ]
    for label, rd, wr in cases:
        best = None
        for wgs in (256, 512, 1024, 2048, 4096, 8192):
            for unroll in (2, 4, 8):
                us = ctx.bandwidth_probe(rd, wr, wgs, unroll, args.iters)
                tb = (rd + wr) / us / 1e6
                print(f"{label:16s} rd {rd / 1e6:7.1f} MB wr {wr / 1e6:6.1f} MB  wgs {wgs:5d} unroll {unroll}  {us:8.2f} us  {tb:5.2f} TB/s")
                if best is None or us < best[0]:
                    best = (us, wgs, unroll)
        print(f"== {label}: best {best[0]:.2f} us at {best[1]} workgroups, unroll {best[2]} -> {(rd + wr) / best[0] / 1e6:.2f} TB/s = {(rd + wr) / best[0] / 1e6 / 8.0:.3f} of 8 TB/s")
    ctx.close()


if __name__ == "__main__":
    main()
