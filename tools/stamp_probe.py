#!/usr/bin/env python3
"""Development aid: where a k_hrtf_ols launch spends its time, from per-wave s_memrealtime stamps.

Needs a diagnostic library built with -DGAS_STAMPS (godot-audio-spatializer_amd/build.py --variant stamps -DGAS_STAMPS)
and loaded through GAS_AMD_LIB.  Runs the bench's default workload (8192 HRTF sources, 1 in 64 draining, device-side
parameter publish every second callback), then reads the stamps of the LAST launch:
  0 body entry, 1 twiddles ready / prologue loads issued, 2 first source's data consumed, 3 source loop done,
  4 first epilogue barrier passed, 5 partial mix stored.
Prints min / median / max over the waves, in us relative to the earliest stamp 0 (100 MHz ticks).
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sources", type=int, default=8192)
    ap.add_argument("--callbacks", type=int, default=41)
    ap.add_argument("--draining-every", type=int, default=64)
    ap.add_argument("--no-pipelined-mix", action="store_true")
    args = ap.parse_args()
    import torch

    import godot_audio_spatializer_amd as gas
    from godot_audio_spatializer_amd import synth

    lib = gas.load_library()
    if not hasattr(lib, "gas_debug_read_stamps"):
        raise SystemExit("library lacks gas_debug_read_stamps: build with -DGAS_STAMPS and set GAS_AMD_LIB")
    n, F = args.sources, 512
    rng = np.random.default_rng(1234)
    hrir = synth.synthetic_hrir(rng, dirs=1024)
    flags = gas.capi.FLAG_PEAKS_DRAINING_ONLY | (0 if args.no_pipelined_mix else gas.capi.FLAG_PIPELINED_MIX)
    ctx = gas.SpatializerContext(max_sources=n, frames=F, flags=flags)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.hrtf_load(hrir)
    slots = ctx.source_alloc_many(n, 2, (3,))
    if args.draining_every > 0:
        for s in slots[:: args.draining_every]:
            ctx.source_set_draining(int(s), True)
    psets = [torch.from_numpy(synth.draw_params(rng, n, dirs=1024, frames=F).view(np.uint8).reshape(n, 128).copy()).cuda() for _ in range(2)]
    ctx.params_publish_batch(slots, synth.draw_params(rng, n, dirs=1024, frames=F))
    n_bufs = 12
    srcs = [torch.rand(n, F, 2, device="cuda") - 0.5 for _ in range(n_bufs)]
    out = torch.zeros(2, 1, F, 2, device="cuda")
    peaks = torch.zeros(n, 2, device="cuda")
    rc = ctx.process_block_raw(srcs[0].data_ptr(), slots, n, F, out[0].data_ptr(), peaks.data_ptr(), 1)
    assert rc == 0, rc
    for k in range(args.callbacks):
        if k % 2 == 0:
            ctx.params_publish_device(psets[(k // 2) % 2].data_ptr(), n)
        rc = ctx.process_block_raw(srcs[k % n_bufs].data_ptr(), None, n, F, out[k % 2].data_ptr(), peaks.data_ptr(), 1)
        assert rc == 0, rc
    ctx.synchronize()
    W = 8192
    buf = np.zeros((W, 8), np.uint64)
    lib.gas_debug_read_stamps.argtypes = [C.c_void_p, C.c_uint64]
    assert lib.gas_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
    live = buf[:, 0] > 0
    st = buf[live].astype(np.int64)
    t0 = st[:, 0].min()
    names = ["entry", "prologue issued", "first data", "loop done", "epilogue barrier 1", "partial stored", "kernargs arrived", "slot/row lists arrived"]
    print(f"{live.sum()} waves stamped; last callback {'published fresh parameters' if (args.callbacks - 1) % 2 == 0 else 'reused the table'}")
    for i, nm in enumerate(names[:6]):
        col = st[:, i]
        ok = col > 0
        if not ok.any():
            continue
        rel = (col[ok] - t0) * 0.01
        print(f"  {i} {nm:20s} min {rel.min():7.2f}  p50 {np.median(rel):7.2f}  p90 {np.percentile(rel, 90):7.2f}  max {rel.max():7.2f} us   ({ok.sum()} waves)")
    if (st[:, 7] > 0).any():  # shader clock held during the launch: s_memtime cycles per s_memrealtime tick (100 MHz)
        ok = (st[:, 7] > 0) & (st[:, 5] > 0)
        mhz = (st[ok, 7] - st[ok, 6]) / np.maximum(st[ok, 5] - st[ok, 0], 1) * 100.0
        print(f"  shader clock inside the launch: p10 {np.percentile(mhz, 10):.0f}  p50 {np.median(mhz):.0f}  p90 {np.percentile(mhz, 90):.0f} MHz")
    names = names[:6]
    # per-workgroup spread of the loop end (who the first epilogue barrier waits for)
    wg = np.nonzero(live)[0] // 8
    loop = (st[:, 3] - t0) * 0.01
    spread = [loop[wg == w].max() - loop[wg == w].min() for w in np.unique(wg)]
    print(f"  loop-done spread inside a workgroup: p50 {np.median(spread):.2f}  max {np.max(spread):.2f} us")
    # who is late: by workgroup index mod 8 (workgroups are dealt round-robin over the 8 XCDs) and by body kind
    end = np.where(st[:, 5] > 0, st[:, 5], st[:, 3])
    endrel = (end - t0) * 0.01
    wg_ids = np.unique(wg)
    wg_end = np.array([endrel[wg == w].max() for w in wg_ids])
    wg_first = np.array([((st[:, 2] - t0) * 0.01)[wg == w].min() for w in wg_ids])
    is_pk = np.array([(st[wg == w, 5] == 0).all() for w in wg_ids])
    for x in range(8):
        m = (wg_ids % 8 == x) & ~is_pk
        print(f"  wg%8={x}: first data {wg_first[m].mean():6.2f}  end mean {wg_end[m].mean():6.2f}  max {wg_end[m].max():6.2f}  ({m.sum()} wgs)")
    if is_pk.any():
        print(f"  exact-peak workgroups ({is_pk.sum()}): loop end mean {wg_end[is_pk].mean():.2f} max {wg_end[is_pk].max():.2f}")
    order = np.argsort(wg_end)
    print("  latest workgroups:", [(int(wg_ids[i]), round(float(wg_end[i]), 2)) for i in order[-8:]])
    print("  earliest workgroups:", [(int(wg_ids[i]), round(float(wg_end[i]), 2)) for i in order[:8]])
    ctx.close()


if __name__ == "__main__":
    main()
