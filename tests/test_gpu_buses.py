"""several output buses in one launch (gas_process_block_buses) vs AudioServer's per-bus
multiply-accumulate of each playback's frames by get_bus_map's factors (audio_spatializer.cpp:274-324; dry bus /
Area3D override / reverb send: audio_spatializer_3d.cpp:437-461).  The expectation is built source by source from the
oracle's mix_channel outputs and the oracle's get_bus_map."""
import ctypes as C

import numpy as np
import pytest

from helpers import TOL, rel_rms

pytestmark = pytest.mark.gpu


def oracle_bus_map(ob, smc, channel, bus_volume, mix_volumes):
    L = ob.lib()
    pass

    out = np.zeros((4, 2), np.float32)
    bv = np.ascontiguousarray(bus_volume, np.float32)
    mv = np.ascontiguousarray(mix_volumes, np.float32)
    L.gaso_bus_map(smc, channel, bv.ctypes.data, mv.ctypes.data, out.ctypes.data)
    return out


@pytest.mark.parametrize("n,channels,n_buses", [(70, 2, 3), (256, 1, 2), (1000, 4, 6), (9000, 1, 2)])
def test_buses_match_per_source_bus_maps(gas, ob, n, channels, n_buses):
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    F = 512
    rng = np.random.default_rng(n)
    with gas.SpatializerContext(max_sources=n, frames=F, channel_count=channels) as ctx:
        slots = ctx.source_alloc_many(n, K.KIND_3D_MIX)
        oracles = [ob.BatchOracle(ob.KIND_3D_MIX, 1, F, channel_count=channels) for _ in range(n)] if n <= 1000 else None
        whole = ob.BatchOracle(ob.KIND_3D_MIX, n, F, channel_count=channels)
        for cb in range(3):
            p = synth.draw_params(rng, n, channel_count=channels, frames=F)
            p["mix_volumes"][::11] = 0.0  # silent pairs: the send is 0 there (mix volume <= 0, :300-305)
            # reverb send volumes per source and pair (what calculate_spatialization's reverb_vol would give)
            reverb = (p["mix_volumes"] * rng.uniform(0.0, 1.5, (n, 1, 1))).astype(np.float32)
            routes = K.bus_routes(n)
            routes["dry_bus"] = rng.integers(0, n_buses, n)  # the player's bus or an Area3D override
            routes["send_bus"] = np.where(rng.uniform(size=n) < 0.6, rng.integers(0, n_buses, n), K.BUS_NONE)
            for s in range(n):
                for c in range(channels):
                    routes["send"][s, c] = oracle_bus_map(ob, 1, c, reverb[s], p["mix_volumes"][s])[c]
            ctx.params_publish_batch(slots, p)
            ctx.bus_routes_publish(slots, routes)
            src = synth.draw_sources(rng, n, F)
            got, peaks = ctx.process_block_buses(src, slots, n_buses)
            _, wpeaks, w64 = whole.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
            np.testing.assert_allclose(peaks, wpeaks, rtol=2e-5, atol=1e-7)  # the gate sees y, not a bus
            if oracles is None:
                # full size: the buses must add up to what they were split from when every send is dropped
                routes["send_bus"] = K.BUS_NONE
                continue
            want = np.zeros((n_buses, channels, F, 2), np.float64)
            for s in range(n):
                _, _, y = oracles[s].block(p[s:s + 1].astype(ob.PARAMS_DTYPE), src[s:s + 1], want64=True)
                y32 = y[:channels].astype(np.float32)
                want[routes["dry_bus"][s]] += y32
                if routes["send_bus"][s] != K.BUS_NONE:
                    for c in range(channels):
                        want[routes["send_bus"][s], c] += (y32[c] * routes["send"][s, c][None, :]).astype(np.float32)  # [ENGINE] AudioServer: frame * volume, f32
            for b in range(n_buses):
                for c in range(channels):
                    if np.abs(want[b, c]).max() > 0:
                        assert rel_rms(got[b, c], want[b, c]) <= TOL, (cb, b, c)
                    else:
                        assert not got[b, c].any()


def test_buses_without_sends_sum_to_the_single_mix(gas):
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    n, F = 3000, 512
    rng = np.random.default_rng(4)
    p = synth.draw_params(rng, n, frames=F)
    src = synth.draw_sources(rng, n, F)
    routes = K.bus_routes(n)
    routes["dry_bus"] = rng.integers(0, 4, n)
    routes["send_bus"] = K.BUS_NONE
    outs = []
    for buses in (True, False):
        with gas.SpatializerContext(max_sources=n, frames=F) as ctx:
            slots = ctx.source_alloc_many(n, K.KIND_3D_MIX)
            ctx.params_publish_batch(slots, p)
            if buses:
                ctx.bus_routes_publish(slots, routes)
                outs.append(ctx.process_block_buses(src, slots, 4)[0].astype(np.float64).sum(axis=0))
            else:
                outs.append(ctx.process_block(src, slots)[0].astype(np.float64))
    assert rel_rms(outs[0][0], outs[1][0]) <= TOL


def test_buses_reject_other_kinds_and_bad_counts(gas):
    K = gas.capi
    with gas.SpatializerContext(max_sources=2, frames=512) as ctx:
        s = ctx.source_alloc(K.KIND_EFFECT)
        ctx.params_publish(s, np.zeros(1, K.PARAMS_DTYPE))
        with pytest.raises(gas.GasError) as ei:
            ctx.process_block_buses(np.zeros((1, 512, 2), np.float32), [s], 2)
        assert ei.value.status == -6
        with pytest.raises(gas.GasError) as ei:
            ctx.process_block_buses(np.zeros((0, 512, 2), np.float32), [], 7)
        assert ei.value.status == -1
        out, _ = ctx.process_block_buses(np.zeros((0, 512, 2), np.float32), [], 3)
        assert out.shape == (3, 1, 512, 2) and not out.any()


@pytest.mark.parametrize("chain_name,n,n_buses,F", [("hrtf", 150, 2, 512), ("hrtf", 130, 1, 256), ("hrtf", 160, 4, 512), ("er_hrtf", 90, 3, 256), ("shelf", 64, 2, 512), ("shelf_hrtf", 70, 6, 512), ("hrtf", 2100, 2, 512)])
def test_effect_kinds_route_to_buses(gas, ob, chain_name, n, n_buses, F):
    """Effect chains (the HRTF spatializer's dry bus + reverb send): run staged, rows mixed per bus.  Expectation: each
    source's own output from a one-source oracle, times its weight on each bus, summed in f32 products like AudioServer."""
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    HS, ER, HRTF = K.FX_HIGHSHELF, K.FX_EARLY_REFLECTIONS, K.FX_HRTF
    chain = {"hrtf": (HRTF,), "er_hrtf": (ER, HRTF), "shelf": (HS,), "shelf_hrtf": (HS, HRTF)}[chain_name]
    ring = 2048 if ER in chain else 0
    dirs = 24
    rng = np.random.default_rng(n + n_buses)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=dirs)
    with gas.SpatializerContext(max_sources=n, frames=F, er_ring_frames=ring) as ctx:
        ctx.hrtf_load(hrir)
        slots = ctx.source_alloc_many(n, K.KIND_EFFECT, chain)
        per_source = n <= 200
        oracles = [ob.BatchOracle(ob.KIND_EFFECT, 1, F, chain=list(chain), hrir=hrir, er_ring_frames=max(ring, 1)) for _ in range(n)] if per_source else None
        whole = ob.BatchOracle(ob.KIND_EFFECT, n, F, chain=list(chain), hrir=hrir, er_ring_frames=max(ring, 1))
        for cb in range(3):
            p = synth.draw_params(rng, n, dirs=dirs, ring_frames=max(ring, 2 * F), frames=F)
            routes = K.bus_routes(n)
            routes["dry_bus"] = rng.integers(0, n_buses, n)
            routes["send_bus"] = np.where(rng.uniform(size=n) < 0.6, rng.integers(0, n_buses, n), K.BUS_NONE)
            routes["send"][:, 0, :] = rng.uniform(0.0, 1.2, (n, 2)).astype(np.float32)
            ctx.params_publish_batch(slots, p)
            ctx.bus_routes_publish(slots, routes)
            src = synth.draw_sources(rng, n, F)
            got, peaks = ctx.process_block_buses(src, slots, n_buses)
            _, wpeaks, w64 = whole.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
            np.testing.assert_allclose(peaks, wpeaks, rtol=3e-5, atol=1e-7)  # the gate sees y, not a bus
            if not per_source:
                assert np.isfinite(got).all()  # full size: peaks above; the per-bus sums are checked by the split test below
                continue
            want = np.zeros((n_buses, F, 2), np.float64)
            for s_ in range(n):
                _, _, y = oracles[s_].block(p[s_:s_ + 1].astype(ob.PARAMS_DTYPE), src[s_:s_ + 1], want64=True)
                y32 = y[0].astype(np.float32)
                want[routes["dry_bus"][s_]] += y32
                if routes["send_bus"][s_] != K.BUS_NONE:
                    want[routes["send_bus"][s_]] += (y32 * routes["send"][s_, 0][None, :]).astype(np.float32)
            for b in range(n_buses):
                if np.abs(want[b]).max() > 0:
                    assert rel_rms(got[b, 0], want[b]) <= TOL, (cb, b)
                else:
                    assert not got[b, 0].any()


def test_effect_buses_without_sends_sum_to_the_single_mix(gas):
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    n, F, dirs = 2600, 512, 32
    rng = np.random.default_rng(8)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=dirs)
    p = synth.draw_params(rng, n, dirs=dirs, frames=F)
    src = synth.draw_sources(rng, n, F)
    routes = K.bus_routes(n)
    routes["dry_bus"] = rng.integers(0, 3, n)
    routes["send_bus"] = K.BUS_NONE
    outs = []
    for buses in (True, False):
        with gas.SpatializerContext(max_sources=n, frames=F) as ctx:
            ctx.hrtf_load(hrir)
            slots = ctx.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
            ctx.params_publish_batch(slots, p)
            if buses:
                ctx.bus_routes_publish(slots, routes)
                got, pk = ctx.process_block_buses(src, slots, 3)
                outs.append((got.astype(np.float64).sum(axis=0)[0], pk))
                # the list is regrouped for the next ordinary callback (the staged grouping was the bus call's only)
                mix2, _ = ctx.process_block(src, slots)
                assert np.isfinite(mix2).all()
            else:
                mix, pk = ctx.process_block(src, slots)
                outs.append((mix[0].astype(np.float64), pk))
    assert rel_rms(outs[0][0], outs[1][0]) <= 2e-6
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=3e-5, atol=1e-7)


def test_bus_call_reuses_the_list(gas):
    """slots == NULL: the previous bus call's list, for the forms that keep the ordinary grouping."""
    import torch

    from godot_audio_spatializer_amd import synth

    K = gas.capi
    n, F = 2300, 512
    rng = np.random.default_rng(3)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=32)
    with gas.SpatializerContext(max_sources=n, frames=F) as ctx:
        ctx.hrtf_load(hrir)
        slots = ctx.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
        ctx.params_publish_batch(slots, synth.draw_params(rng, n, dirs=32, frames=F))
        routes = K.bus_routes(n)
        routes["send_bus"] = 1
        routes["send"][:, 0, :] = 0.25
        ctx.bus_routes_publish(slots, routes)
        src = torch.from_numpy(synth.draw_sources(rng, n, F)).cuda()
        out = torch.zeros(2, 2, 1, F, 2, device="cuda")
        pk = torch.zeros(n, 2, device="cuda")
        s32 = np.ascontiguousarray(slots, np.uint32)
        torch.cuda.synchronize()
        # same input twice from a fresh state on two contexts would be the strict check; here: call 2 (reused list) must
        # equal call 2 with the list passed, on a twin context
        rc = ctx.lib.gas_process_block_buses(ctx.h, src.data_ptr(), s32.ctypes.data, n, F, out[0, 0].data_ptr(), 2, pk.data_ptr(), K.MEM_DEVICE)
        assert rc == 0
        rc = ctx.lib.gas_process_block_buses(ctx.h, src.data_ptr(), None, n, F, out[0].data_ptr(), 2, pk.data_ptr(), K.MEM_DEVICE)
        assert rc == 0
        ctx.synchronize()
        assert ctx.lib.gas_process_block_buses(ctx.h, src.data_ptr(), None, n + 1, F, out[0].data_ptr(), 2, pk.data_ptr(), K.MEM_DEVICE) != 0
        out3 = torch.zeros(3, 1, F, 2, device="cuda")
        assert ctx.lib.gas_process_block_buses(ctx.h, src.data_ptr(), None, n, F, out3.data_ptr(), 3, pk.data_ptr(), K.MEM_DEVICE) == 0  # three buses: two fused launches over the same list
        ctx.synchronize()
        assert not out3[2].any()  # nobody sends to bus 2
    with gas.SpatializerContext(max_sources=n, frames=F) as ctx2:
        ctx2.hrtf_load(hrir)
        slots2 = ctx2.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
        rng = np.random.default_rng(3)
        ctx2.params_publish_batch(slots2, synth.draw_params(rng, n, dirs=32, frames=F))
        ctx2.bus_routes_publish(slots2, routes)
        s2 = np.ascontiguousarray(slots2, np.uint32)
        for _ in range(2):
            assert ctx2.lib.gas_process_block_buses(ctx2.h, src.data_ptr(), s2.ctypes.data, n, F, out[1].data_ptr(), 2, pk.data_ptr(), K.MEM_DEVICE) == 0
        out3b = torch.zeros(3, 1, F, 2, device="cuda")
        assert ctx2.lib.gas_process_block_buses(ctx2.h, src.data_ptr(), s2.ctypes.data, n, F, out3b.data_ptr(), 3, pk.data_ptr(), K.MEM_DEVICE) == 0
        ctx2.synchronize()
    a, b = out[0].cpu().numpy(), out[1].cpu().numpy()
    assert np.array_equal(a, b) and np.abs(a[1]).max() > 0
    assert np.array_equal(out3.cpu().numpy(), out3b.cpu().numpy()) and np.abs(out3b[1].cpu().numpy()).max() > 0  # the third callback, reused list vs named list


@pytest.mark.parametrize("peaks_mode", ["every_source", "draining_only"])
def test_fused_two_bus_form_bus_sums_against_the_oracle_at_size(gas, ob, peaks_mode):
    """k_hrtf_uni<BUS2> at a size where per-source oracles are too slow: bus b's mix is linear in each source's gain, so
    the expectation for (bus, ear) is ONE batch oracle whose gains are scaled by every source's weight on that bus
    and ear (routes held constant over the callbacks, so the scaled previous gains stay consistent).  Peaks are the
    unscaled oracle's (the gate sees y, not a bus) -- for every source, or for the draining ones only."""
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    n, F, dirs, n_buses = 2304 + 77, 512, 40, 2
    rng = np.random.default_rng(21)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=dirs)
    routes = K.bus_routes(n)
    routes["dry_bus"] = rng.integers(0, n_buses, n)
    routes["send_bus"] = np.where(rng.uniform(size=n) < 0.7, rng.integers(0, n_buses, n), K.BUS_NONE)
    routes["send"][:, 0, :] = rng.uniform(0.0, 1.2, (n, 2)).astype(np.float32)
    w = np.zeros((n_buses, 2, n), np.float32)  # weight of source s on (bus, ear)
    for b in range(n_buses):
        for ear in range(2):
            w[b, ear] = (routes["dry_bus"] == b) + np.where(routes["send_bus"] == b, routes["send"][:, 0, ear], 0.0)
    flags = K.FLAG_PEAKS_DRAINING_ONLY if peaks_mode == "draining_only" else 0
    draining = np.arange(0, n, 29)
    plain = ob.BatchOracle(ob.KIND_EFFECT, n, F, chain=[K.FX_HRTF], hrir=hrir)
    scaled = [[ob.BatchOracle(ob.KIND_EFFECT, n, F, chain=[K.FX_HRTF], hrir=hrir) for _ in range(2)] for _ in range(n_buses)]
    with gas.SpatializerContext(max_sources=n, frames=F, flags=flags) as ctx:
        ctx.hrtf_load(hrir)
        slots = ctx.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
        if peaks_mode == "draining_only":
            for s in draining:
                ctx.source_set_draining(int(slots[s]), True)
        ctx.bus_routes_publish(slots, routes)
        for cb in range(3):
            p = synth.draw_params(rng, n, dirs=dirs, frames=F)
            ctx.params_publish_batch(slots, p)
            src = synth.draw_sources(rng, n, F)
            got, peaks = ctx.process_block_buses(src, slots, n_buses)
            _, wpeaks, _ = plain.block(p.astype(ob.PARAMS_DTYPE), src)
            if peaks_mode == "every_source":
                np.testing.assert_allclose(peaks, wpeaks, rtol=3e-5, atol=1e-7)
            else:
                np.testing.assert_allclose(peaks[draining], wpeaks[draining], rtol=3e-5, atol=1e-7)
                assert np.all(np.isposinf(np.delete(peaks, draining, axis=0)))
            for b in range(n_buses):
                for ear in range(2):
                    q = p.copy()
                    q["hrtf_gain"] = (p["hrtf_gain"] * w[b, ear]).astype(np.float32)
                    _, _, y64 = scaled[b][ear].block(q.astype(ob.PARAMS_DTYPE), src, want64=True)
                    assert rel_rms(got[b, 0, :, ear], y64[0, :, ear]) <= TOL, (cb, b, ear)


@pytest.mark.parametrize("kind_name,chain,channels", [("KIND_3D_MIX", (), 2), ("KIND_EFFECT", (3,), 1), ("KIND_EFFECT", (1, 3), 1)])
def test_up_to_six_buses_per_playback(gas, ob, kind_name, chain, channels):
    """audio_spatializer.cpp:283-287 walks EVERY key of bus_volumes, up to MAX_BUSES_PER_PLAYBACK = 6: a playback's dry
    bus plus up to five sends (gas_bus_route.more_bus / more_send).  Oracle: per-source outputs, multiplied per bus the
    way AudioServer does ([ENGINE] frame * volume, f32) and summed in f64."""
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    kind = getattr(K, kind_name)
    okind = getattr(ob, kind_name)
    F, n, n_buses = 512, 90, 6
    rng = np.random.default_rng(17)
    hrir = synth.synthetic_hrir(rng, dirs=16) if 3 in chain else None
    with gas.SpatializerContext(max_sources=n, frames=F, channel_count=channels) as ctx:
        if hrir is not None:
            ctx.hrtf_load(hrir)
        slots = ctx.source_alloc_many(n, kind, chain)
        oracles = [ob.BatchOracle(okind, 1, F, channel_count=channels, chain=chain, hrir=hrir) for _ in range(n)]
        for cb in range(3):
            p = synth.draw_params(rng, n, dirs=16, channel_count=channels, frames=F)
            routes = K.bus_routes(n)
            routes["dry_bus"] = rng.integers(0, n_buses, n)
            n_sends = rng.integers(0, 6, n)  # 0..5 sends
            for s in range(n):
                order = rng.permutation(n_buses)  # distinct target buses, like dictionary keys; may include the dry bus
                for k in range(n_sends[s]):
                    vol = rng.uniform(0.0, 1.2, (K.MAX_CHANNELS, 2)).astype(np.float32)
                    if k == 0:
                        routes["send_bus"][s], routes["send"][s] = order[k], vol
                    else:
                        routes["more_bus"][s, k - 1], routes["more_send"][s, k - 1] = order[k], vol
            ctx.params_publish_batch(slots, p)
            ctx.bus_routes_publish(slots, routes)
            src = synth.draw_sources(rng, n, F)
            got, peaks = ctx.process_block_buses(src, slots, n_buses)
            want = np.zeros((n_buses, channels, F, 2), np.float64)
            for s in range(n):
                _, pk, y = oracles[s].block(p[s:s + 1].astype(ob.PARAMS_DTYPE), src[s:s + 1], want64=True)
                np.testing.assert_allclose(peaks[s], pk[0], rtol=2e-5, atol=1e-7)
                y32 = y[:channels].astype(np.float32)
                want[routes["dry_bus"][s]] += y32
                sends = [(routes["send_bus"][s], routes["send"][s])] + [(routes["more_bus"][s, k], routes["more_send"][s, k]) for k in range(K.MAX_MORE_SENDS)]
                for bus, vol in sends:
                    if bus != K.BUS_NONE:
                        for c in range(channels):
                            want[bus, c] += (y32[c] * vol[c][None, :]).astype(np.float32)
            for b in range(n_buses):
                for c in range(channels):
                    assert rel_rms(got[b, c], want[b, c]) <= TOL, (cb, b, c)


def test_probe_between_bus_callbacks_leaves_the_bus_buffers_alone(gas, ob):
    """Regression (round-2 advisor, high): gas_bandwidth_probe growing its arena used to free the bus form's route /
    partial / output buffers; buses -> probe -> buses on ONE context must keep working and agree with a twin context
    that never probed."""
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    n, F = 300, 512
    outs = []
    for probe in (True, False):
        rng = np.random.default_rng(8)
        with gas.SpatializerContext(max_sources=n, frames=F, channel_count=2) as ctx:
            slots = ctx.source_alloc_many(n, K.KIND_3D_MIX)
            routes = K.bus_routes(n)
            routes["dry_bus"] = rng.integers(0, 3, n)
            routes["send_bus"] = 2
            routes["send"] = 0.5
            ctx.bus_routes_publish(slots, routes)
            got = []
            for cb in range(3):
                ctx.params_publish_batch(slots, synth.draw_params(rng, n, channel_count=2, frames=F))
                got.append(ctx.process_block_buses(synth.draw_sources(rng, n, F), slots, 3)[0])
                if probe and cb == 0:
                    assert ctx.bandwidth_probe(8 << 20, 1 << 20, 256, 2, 3) > 0  # grows both arenas
                    assert ctx.bandwidth_probe(32 << 20, 4 << 20, 256, 2, 3) > 0  # ... and again
            outs.append(np.stack(got))
    np.testing.assert_array_equal(outs[0], outs[1])
