#!/usr/bin/env python3
"""Synchronous-form timing of the plain-[HRTF] callback (k_hrtf_uni + k_mix_reduce, one launch per callback) at several
source counts, for the eight-wave and the twelve-wave form of the kernel (gas_tune_uni12_min).

Per size and form: GPU-timeline time per callback over `--steps` ordered callbacks queued back to back (device-resident
rotating sources, device-published parameters every second callback, 1 source in 64 draining), the dominant kernel's
span from the library's own event bracket (gas_profile_*), and both as a fraction of the 8 TB/s HBM roof over the
algorithmic bytes the library reports for that launch.  Usage: tools/time_uni.py [--sizes 8192,10240,65536] [--steps 200]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="8192,10240,16384,65536,1048576")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--dirs", type=int, default=1024)
    ap.add_argument("--forms", default="8,12")
    ap.add_argument("--flags", default="ordered", choices=["ordered", "pipelined"])
    args = ap.parse_args()
    import torch

    import godot_audio_spatializer_amd as gas
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    lib = gas.load_library()
    F = 512
    hrir = synth.synthetic_hrir(np.random.default_rng(1234), dirs=args.dirs)
    rows = []
    for n in [int(x) for x in args.sizes.split(",")]:
        steps = max(20, min(args.steps, (1 << 24) // n))
        n_bufs = max(2, min(16, -(-(320 << 20) // (n * F * 8))))
        gen = torch.Generator(device="cuda")
        gen.manual_seed(1)
        srcs = [torch.rand(n, F, 2, device="cuda", generator=gen) - 0.5 for _ in range(n_bufs)]
        prng = np.random.default_rng(5)
        psets = []
        for _ in range(2):
            chunks = [synth.draw_params(prng, min(1 << 18, n - a), dirs=args.dirs, frames=F) for a in range(0, n, 1 << 18)]
            p = np.concatenate(chunks)
            psets.append(torch.from_numpy(p.view(np.uint8).reshape(n, 128).copy()).cuda())
        out = torch.zeros(2, 1, F, 2, device="cuda")
        peaks = torch.zeros(n, 2, device="cuda")
        for form in [int(x) for x in args.forms.split(",")]:
            lib.gas_tune_uni12_min(1 if form == 12 else 0)
            flags = K.FLAG_PEAKS_DRAINING_ONLY | (K.FLAG_PIPELINED_MIX if args.flags == "pipelined" else 0)
            with gas.SpatializerContext(max_sources=n, frames=F, flags=flags) as ctx:
                ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                ctx.hrtf_load(hrir)
                slots = ctx.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
                for s_ in slots[::64]:
                    ctx.source_set_draining(int(s_), True)
                ctx.params_publish_device(psets[0].data_ptr(), n, slots)

                def step(k, first=False):
                    if k % 2 == 0 and not first:
                        ctx.params_publish_device(psets[(k // 2) % 2].data_ptr(), n)
                    rc = ctx.process_block_raw(srcs[k % n_bufs].data_ptr(), slots if first else None, n, F, out[k % 2].data_ptr(), peaks.data_ptr(), K.MEM_DEVICE)
                    assert rc == 0, rc

                step(0, True)
                for k in range(1, 24):
                    step(k)
                ctx.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for k in range(steps):
                    step(k)
                ctx.join_outputs()
                e1.record()
                torch.cuda.synchronize()
                us_step = e0.elapsed_time(e1) / steps * 1e3
                ctx.profile_enable(1)
                ctx.profile_read(reset=True)
                for k in range(min(steps, 48)):
                    step(k)
                ctx.join_outputs()
                torch.cuda.synchronize()
                prof = ctx.profile_read(reset=True)
                ctx.profile_enable(0)
                k_us = prof["kernel_ms"] / max(prof["launches"], 1) * 1e3
                B = prof["bytes_per_launch"]
                rows.append({"sources": n, "waves": form, "mode": args.flags, "us_per_callback": round(us_step, 2), "kernel": prof["kernel"], "kernel_us": round(k_us, 2), "algorithmic_bytes": B, "frac_kernel": round(B / (k_us * 1e-6) / 8e12, 3) if k_us > 0 else None, "frac_callback": round(B / (us_step * 1e-6) / 8e12, 3)})
                print(json.dumps(rows[-1]), flush=True)
        del srcs, psets, out, peaks
        torch.cuda.empty_cache()
    lib.gas_tune_uni12_min(0)


if __name__ == "__main__":
    main()
