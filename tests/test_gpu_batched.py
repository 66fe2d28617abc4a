"""GAS_FLAG_BATCHED_LAUNCH: consecutive device-memory callbacks of an unchanged plain-[HRTF] list run as one
k_hrtf_multi launch per `depth` callbacks.  Same operations in the same order as k_hrtf_uni launches, so every mix and
every peak must be BITWISE what the ordered mode produces -- whatever falls between the calls of a batch
(device-published rows for any block, host publishes, a host-memory call, a new list, a join, callbacks left over at
the end)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(gas, flags, n, F, T, events, dirs=48, depth=None):
    """events: {t: [names]} applied before callback t: 'dev_publish', 'host_publish', 'host_call', 'relist', 'join'."""
    import torch

    from godot_audio_spatializer_amd import synth

    K = gas.capi
    rng = np.random.default_rng(33)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=dirs)
    ctx = gas.SpatializerContext(max_sources=n + 4, frames=F, flags=flags)
    ctx.hrtf_load(hrir)
    if depth is not None:
        ctx.set_batch_depth(depth)
    slots = ctx.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
    for s in slots[::11]:
        ctx.source_set_draining(s, True)
    ctx.params_publish_batch(slots, synth.draw_params(rng, n, dirs=dirs, frames=F))
    outs = torch.full((T, 1, F, 2), float("nan"), device="cuda")
    peaks = torch.zeros(T, n, 2, device="cuda")
    keep = []  # sources and device parameter rows stay alive until the end (the paired mode's contract)
    torch.cuda.synchronize()
    first = True
    for t in range(T):
        ev = events.get(t, ())
        src = synth.draw_sources(rng, n, F)
        if "host_publish" in ev:
            ctx.params_publish_batch(slots[: n // 2], synth.draw_params(rng, n // 2, dirs=dirs, frames=F))
        if "dev_publish" in ev and t == 0:  # the device form addresses the previous callback's list: none yet
            ctx.params_publish_batch(slots, synth.draw_params(rng, n, dirs=dirs, frames=F))
        elif "dev_publish" in ev:
            for _ in range(list(ev).count("dev_publish")):  # twice in a row: the first one is scattered, not consumed
                p = synth.draw_params(rng, n, dirs=dirs, frames=F)
                d_p = torch.from_numpy(p.view(np.uint8).reshape(n, -1).copy()).cuda()
                keep.append(d_p)
                torch.cuda.synchronize()
                ctx.params_publish_device(d_p.data_ptr(), n)
        if "join" in ev:
            ctx.join_outputs()
        if "host_call" in ev:
            mix, pk = ctx.process_block(src, slots)
            outs[t] = torch.from_numpy(mix).cuda()
            peaks[t] = torch.from_numpy(pk).cuda()
            first = True  # the host call passed the list; pass it again to the device calls (as a caller would not need to)
            continue
        d_src = torch.from_numpy(src).cuda()
        keep.append(d_src)
        torch.cuda.synchronize()
        relist = first or "relist" in ev
        rc = ctx.process_block_raw(d_src.data_ptr(), slots if relist else None, n, F, outs[t].data_ptr(), peaks[t].data_ptr(), K.MEM_DEVICE)
        assert rc == 0
        first = False
    ctx.synchronize()
    res, pk = outs.cpu().numpy(), peaks.cpu().numpy()
    ctx.close()
    return res, pk


CASES = {
    "plain_even": dict(n=2048, F=512, T=8, events={t: ["dev_publish"] for t in range(0, 8, 2)}),
    "plain_odd_tail": dict(n=2048, F=512, T=9, events={t: ["dev_publish"] for t in range(0, 9, 2)}),
    "publish_inside_a_pair": dict(n=2500, F=512, T=8, events={1: ["dev_publish"], 3: ["dev_publish"], 4: ["dev_publish"], 5: ["dev_publish"]}),
    "no_publish_at_all": dict(n=2048, F=512, T=6, events={}),
    "host_publish_breaks_a_pair": dict(n=2300, F=512, T=9, events={3: ["host_publish"], 4: ["host_publish"], 6: ["dev_publish"]}),
    "host_call_and_relist": dict(n=2048, F=512, T=10, events={2: ["dev_publish"], 3: ["host_call"], 6: ["relist"], 7: ["dev_publish"]}),
    "join_between": dict(n=2048, F=512, T=7, events={1: ["join"], 4: ["join", "dev_publish"]}),
    "f256": dict(n=4100, F=256, T=6, events={0: ["dev_publish"], 3: ["dev_publish"]}),
    "f128": dict(n=2048, F=128, T=5, events={2: ["dev_publish"]}),
    "many_per_wave": dict(n=9000, F=512, T=6, events={0: ["dev_publish"], 2: ["dev_publish"], 4: ["dev_publish"]}),
    "too_small_to_pair": dict(n=700, F=512, T=6, events={0: ["dev_publish"], 2: ["dev_publish"]}),
    "hist_through_memory": dict(n=14500, F=512, T=5, events={0: ["dev_publish"], 2: ["dev_publish"], 3: ["dev_publish"]}),  # 8 sources per wave: history rows do not fit the LDS
    "f256_hist_through_memory": dict(n=10000, F=256, T=5, events={1: ["dev_publish"]}),
    "two_device_publishes_in_a_row": dict(n=2048, F=512, T=7, events={2: ["dev_publish", "dev_publish"], 3: ["dev_publish"], 5: ["dev_publish", "dev_publish"]}),
    "long_run": dict(n=2048, F=512, T=21, events={t: ["dev_publish"] for t in range(0, 21, 2)}),
    "long_run_with_breaks": dict(n=2200, F=512, T=23, events={3: ["dev_publish"], 5: ["host_publish"], 9: ["join"], 10: ["dev_publish"], 13: ["relist"], 14: ["dev_publish"], 19: ["host_call"]}),
}


_BASE = {}


@pytest.mark.parametrize("depth", [2, 3, 8, 16])
@pytest.mark.parametrize("case", list(CASES))
def test_batched_launch_is_bitwise_the_ordered_mode(gas, case, depth):
    K = gas.capi
    args = CASES[case]
    if case not in _BASE:
        _BASE[case] = _render(gas, K.FLAG_PEAKS_DRAINING_ONLY, **args)
    base, pk0 = _BASE[case]
    pair, pk1 = _render(gas, K.FLAG_PEAKS_DRAINING_ONLY | K.FLAG_PIPELINED_MIX | K.FLAG_BATCHED_LAUNCH, depth=depth, **args)
    assert not np.isnan(base).any() and np.abs(base).max() > 0
    assert np.array_equal(base, pair)
    assert np.array_equal(pk0, pk1)


def test_batched_launch_exact_peaks_for_every_source(gas):
    K = gas.capi
    args = dict(n=2048, F=512, T=5, events={0: ["dev_publish"], 2: ["dev_publish"]})
    base, pk0 = _render(gas, 0, **args)
    pair, pk1 = _render(gas, K.FLAG_PIPELINED_MIX | K.FLAG_BATCHED_LAUNCH, depth=4, **args)
    assert np.array_equal(base, pair)
    assert np.array_equal(pk0, pk1) and np.isfinite(pk0).all()


def test_batched_launch_really_batches(gas):
    """The profiler names the dominant launch: with the flag a stream of unchanged callbacks runs k_hrtf_multi."""
    import torch

    from godot_audio_spatializer_amd import synth

    K = gas.capi
    n, F = 2048, 512
    rng = np.random.default_rng(1)
    ctx = gas.SpatializerContext(max_sources=n, frames=F, flags=K.FLAG_PEAKS_DRAINING_ONLY | K.FLAG_PIPELINED_MIX | K.FLAG_BATCHED_LAUNCH)
    try:
        ctx.hrtf_load(synth.synthetic_hrir(np.random.default_rng(7), dirs=32))
        slots = ctx.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
        ctx.params_publish_batch(slots, synth.draw_params(rng, n, dirs=32, frames=F))
        src = torch.from_numpy(synth.draw_sources(rng, n, F)).cuda()
        out = torch.zeros(8, 1, F, 2, device="cuda")
        pk = torch.zeros(n, 2, device="cuda")
        torch.cuda.synchronize()
        assert ctx.process_block_raw(src.data_ptr(), slots, n, F, out[0].data_ptr(), pk.data_ptr(), K.MEM_DEVICE) == 0
        ctx.synchronize()
        ctx.profile_enable(1)
        ctx.profile_read(reset=True)
        for t in range(1, 7):
            assert ctx.process_block_raw(src.data_ptr(), None, n, F, out[t].data_ptr(), pk.data_ptr(), K.MEM_DEVICE) == 0
        ctx.synchronize()
        prof = ctx.profile_read()
        assert prof["kernel"].startswith("k_hrtf_multi") and prof["launches"] == 3
        ctx.set_batch_depth(6)
        ctx.profile_read(reset=True)
        for t in range(6):
            assert ctx.process_block_raw(src.data_ptr(), None, n, F, out[t].data_ptr(), pk.data_ptr(), K.MEM_DEVICE) == 0
        ctx.synchronize()
        assert ctx.profile_read()["launches"] == 1
        with pytest.raises(gas.GasError):
            ctx.set_batch_depth(17)
        ctx.profile_enable(0)
    finally:
        ctx.close()


def test_batched_launch_soak_against_the_unbatched_mode(gas):
    """600 callbacks at depth 10 with a publish pattern that does not line up with the batches, against one launch per
    callback: bitwise, every callback.  (The hand-over inside k_hrtf_multi is a pair of LDS counters; this is the long
    run that would show a lost or early hand-over as a wrong mix.)"""
    import torch

    from godot_audio_spatializer_amd import synth

    K = gas.capi
    n, F, T, dirs = 4096, 512, 600, 64
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=dirs)
    rng = np.random.default_rng(5)
    pool = [torch.from_numpy(synth.draw_sources(rng, n, F)).cuda() for _ in range(12)]
    psets = [torch.from_numpy(synth.draw_params(rng, n, dirs=dirs, frames=F).view(np.uint8).reshape(n, -1).copy()).cuda() for _ in range(5)]
    first = synth.draw_params(rng, n, dirs=dirs, frames=F)
    res = {}
    for name, flags in (("one", K.FLAG_PEAKS_DRAINING_ONLY | K.FLAG_PIPELINED_MIX), ("batched", K.FLAG_PEAKS_DRAINING_ONLY | K.FLAG_PIPELINED_MIX | K.FLAG_BATCHED_LAUNCH)):
        ctx = gas.SpatializerContext(max_sources=n, frames=F, flags=flags)
        ctx.hrtf_load(hrir)
        if name == "batched":
            ctx.set_batch_depth(10)
        slots = ctx.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
        for s in slots[::13]:
            ctx.source_set_draining(s, True)
        ctx.params_publish_batch(slots, first)
        outs = torch.full((T, 1, F, 2), float("nan"), device="cuda")
        peaks = torch.zeros(n, 2, device="cuda")
        torch.cuda.synchronize()
        for t in range(T):
            if t > 0 and t % 3 == 0:
                ctx.params_publish_device(psets[(t // 3) % len(psets)].data_ptr(), n)
            assert ctx.process_block_raw(pool[t % len(pool)].data_ptr(), slots if t == 0 else None, n, F, outs[t].data_ptr(), peaks.data_ptr(), K.MEM_DEVICE) == 0
            if t % 97 == 96:
                ctx.join_outputs()
        ctx.synchronize()
        res[name] = (outs.cpu().numpy(), peaks.cpu().numpy())
        ctx.close()
    assert not np.isnan(res["one"][0]).any()
    assert np.array_equal(res["one"][0], res["batched"][0])
    assert np.array_equal(res["one"][1], res["batched"][1])


@pytest.mark.parametrize("seed", range(12))
def test_batched_launch_random_event_sequences(gas, seed):
    """Random sizes, frame counts, depths and event sequences: batched == ordered, bitwise."""
    K = gas.capi
    rng = np.random.default_rng(400 + seed)
    n = int(rng.integers(2048, 7000))
    F = int(rng.choice([128, 256, 384, 512]))
    T = int(rng.integers(6, 30))
    depth = int(rng.integers(2, 17))
    names = ["dev_publish", "host_publish", "join", "relist", "host_call"]
    weights = [0.45, 0.15, 0.15, 0.15, 0.10]
    events = {}
    for t in range(T):
        if rng.random() < 0.4:
            events[t] = [str(rng.choice(names, p=weights))]
    peaks_flag = K.FLAG_PEAKS_DRAINING_ONLY if rng.random() < 0.7 else 0
    args = dict(n=n, F=F, T=T, events=events)
    base, pk0 = _render(gas, peaks_flag, **args)
    got, pk1 = _render(gas, peaks_flag | K.FLAG_PIPELINED_MIX | K.FLAG_BATCHED_LAUNCH, depth=depth, **args)
    assert not np.isnan(base).any()
    assert np.array_equal(base, got), f"seed {seed}: n {n} F {F} T {T} depth {depth} events {events}"
    assert np.array_equal(pk0, pk1)


def test_batched_launch_against_the_oracle(gas, ob):
    """The headline kernel compared with the oracle directly (not through k_hrtf_uni): depth 10, device-published rows
    taking effect at some blocks, draining sources (exact peaks), a join in the middle of a batch, and a list that
    gains a playback mid-run (a second oracle starts there; the mixes add).  Every callback's mix within 1e-5 relative
    RMS, the draining sources' peaks within 2e-5."""
    import torch

    from godot_audio_spatializer_amd import synth
    from helpers import TOL, rel_rms

    K = gas.capi
    n, F, T, dirs = 3000, 512, 26, 48
    rng = np.random.default_rng(101)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=dirs)
    publish_at = {3, 4, 11, 12, 20}  # callbacks in front of which device-resident parameter rows are published
    join_at, grow_at = 13, 17
    with gas.SpatializerContext(max_sources=n + 4, frames=F, flags=K.FLAG_PEAKS_DRAINING_ONLY | K.FLAG_PIPELINED_MIX | K.FLAG_BATCHED_LAUNCH) as ctx:
        ctx.hrtf_load(hrir)
        ctx.set_batch_depth(10)
        slots = ctx.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
        draining = np.arange(0, n, 11)
        for s in draining:
            ctx.source_set_draining(int(slots[s]), True)
        params = synth.draw_params(rng, n, dirs=dirs, frames=F)
        ctx.params_publish_batch(slots, params)
        ora = ob.BatchOracle(ob.KIND_EFFECT, n, F, chain=[K.FX_HRTF], hrir=hrir)
        extra_ora, extra_params = None, None
        outs = torch.full((T, 1, F, 2), float("nan"), device="cuda")
        peaks = torch.zeros(T, n + 1, 2, device="cuda")
        keep, want, want_pk = [], [], []
        torch.cuda.synchronize()
        cur_n, first = n, True
        for t in range(T):
            if t == grow_at:  # one more playback joins the list (newest first in the reference; any order here)
                new = ctx.source_alloc(K.KIND_EFFECT, (K.FX_HRTF,))
                extra_params = synth.draw_params(rng, 1, dirs=dirs, frames=F)
                ctx.params_publish(new, extra_params)
                slots = np.append(slots, np.uint32(new))
                extra_ora = ob.BatchOracle(ob.KIND_EFFECT, 1, F, chain=[K.FX_HRTF], hrir=hrir)
                cur_n, first = n + 1, True
            if t in publish_at:
                p = synth.draw_params(rng, cur_n, dirs=dirs, frames=F)
                d_p = torch.from_numpy(p.view(np.uint8).reshape(cur_n, -1).copy()).cuda()
                keep.append(d_p)
                torch.cuda.synchronize()
                if first:
                    ctx.params_publish_batch(slots, p)  # the device form addresses the previous callback's list
                else:
                    ctx.params_publish_device(d_p.data_ptr(), cur_n)
                params = p[:n]
                if cur_n > n:
                    extra_params = p[n:]
            if t == join_at:
                ctx.join_outputs()
            src = synth.draw_sources(rng, cur_n, F)
            d_src = torch.from_numpy(src).cuda()
            keep.append(d_src)
            torch.cuda.synchronize()
            rc = ctx.process_block_raw(d_src.data_ptr(), slots if first else None, cur_n, F, outs[t].data_ptr(), peaks[t].data_ptr(), K.MEM_DEVICE)
            assert rc == 0
            first = False
            _, rp, r64 = ora.block(params.astype(ob.PARAMS_DTYPE), src[:n], want64=True)
            if extra_ora is not None:
                _, rp1, e64 = extra_ora.block(extra_params.astype(ob.PARAMS_DTYPE), src[n:], want64=True)
                r64 = r64 + e64
            want.append(r64)
            want_pk.append(rp)
        ctx.synchronize()
        got, got_pk = outs.cpu().numpy(), peaks.cpu().numpy()
    for t in range(T):
        assert rel_rms(got[t][0], want[t][0]) <= TOL, f"callback {t}"
        np.testing.assert_allclose(got_pk[t][draining], want_pk[t][draining], rtol=2e-5, atol=1e-7, err_msg=f"callback {t}")
        assert np.all(np.isposinf(np.delete(got_pk[t][:n], draining, axis=0)))
