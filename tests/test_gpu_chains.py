"""effect chains without a fused kernel run staged (one launch per effect through ping-pong row
buffers, then k_rows_accumulate) and must match the oracle's ping-pong chain (audio_spatializer_effect.cpp:52-76)."""
import numpy as np
import pytest

from helpers import TOL, rel_rms

pytestmark = pytest.mark.gpu

HS, ER, HRTF = 1, 2, 3


def _hrir(dirs=32, seed=5):
    from godot_audio_spatializer_amd import synth

    return synth.synthetic_hrir(np.random.default_rng(seed), dirs=dirs)


@pytest.mark.parametrize(
    "chain,frames",
    [
        ((HS, HRTF), 512),
        ((HRTF, HS), 512),
        ((HS, HS), 512),
        ((ER, HS), 256),
        ((HS, ER), 256),
        ((HS, ER, HRTF), 256),
        ((ER, HRTF, HS), 256),
        ((HS, ER, HRTF, HS), 128),
        ((HRTF, ER), 256),
    ],
)
def test_staged_chain_matches_oracle(gas, ob, chain, frames):
    from test_gpu_parity import run_pair

    ring = 4096 if ER in chain else 0
    hrir = _hrir() if HRTF in chain else None
    run_pair(gas, ob, gas.capi.KIND_EFFECT, chain, 70, frames, 10, hrir=hrir, ring=ring, dirs=32, redraw_every=3)


@pytest.mark.parametrize("chain,frames,n", [((HS, HRTF), 512, 700), ((HS, ER, HRTF), 256, 300)])
def test_staged_chain_ending_in_hrtf_reports_draining_peaks_only(gas, ob, chain, frames, n):
    """GAS_FLAG_PEAKS_DRAINING_ONLY on a staged chain whose last stage is the one-launch HRTF kernel: the draining
    sources' exact peaks, +inf for the others, the mix as before; run_pair also flips nothing else."""
    from test_gpu_parity import run_pair

    ring = 4096 if ER in chain else 0
    run_pair(gas, ob, gas.capi.KIND_EFFECT, chain, n, frames, 5, hrir=_hrir(), ring=ring, dirs=32, redraw_every=2, flags=gas.capi.FLAG_PEAKS_DRAINING_ONLY, draining_every=5)
    run_pair(gas, ob, gas.capi.KIND_EFFECT, chain, 40, frames, 3, hrir=_hrir(), ring=ring, dirs=32, flags=gas.capi.FLAG_PEAKS_DRAINING_ONLY, draining_every=0)  # none draining
    run_pair(gas, ob, gas.capi.KIND_EFFECT, chain, 40, frames, 3, hrir=_hrir(), ring=ring, dirs=32, flags=gas.capi.FLAG_PEAKS_DRAINING_ONLY, draining_every=1)  # all draining


def test_staged_chain_crossfade(gas, ob):
    from test_gpu_parity import run_pair

    run_pair(gas, ob, gas.capi.KIND_EFFECT, (HS, HRTF), 45, 512, 8, hrir=_hrir(), dirs=32, redraw_every=2, flags=gas.capi.FLAG_HRTF_CROSSFADE)


def test_mixed_chains_in_one_callback(gas, ob):
    """Fused and staged chains interleaved in one callback: the mix is the sum of the per-chain oracles' mixes and
    every source's peak lands on its own row."""
    from godot_audio_spatializer_amd import synth

    rng = np.random.default_rng(11)
    frames, ring = 256, 4096
    hrir = _hrir()
    chains = [(HRTF,), (HS, HRTF), (ER, HRTF), (HRTF, HS), (HS,), (HS, ER), (HS, HRTF)]
    per = [9, 13, 7, 11, 5, 8, 6]
    n = sum(per)
    ctx = gas.SpatializerContext(max_sources=n, frames=frames, er_ring_frames=ring)
    ctx.hrtf_load(hrir)
    slots, oras, owner = [], [], []
    for k, (ch, m) in enumerate(zip(chains, per)):
        slots.append(ctx.source_alloc_many(m, gas.capi.KIND_EFFECT, ch))
        oras.append(ob.BatchOracle(ob.KIND_EFFECT, m, frames, chain=ch, hrir=hrir, er_ring_frames=ring))
        owner += [k] * m
    slots = np.concatenate(slots)
    owner = np.asarray(owner)
    perm = rng.permutation(n)  # interleave the chains in the callback's order
    for b in range(8):
        if b % 3 == 0:
            p = synth.draw_params(rng, n, dirs=32, ring_frames=ring, frames=frames)
            ctx.params_publish_batch(slots, p)
        src = synth.draw_sources(rng, n, frames)
        mix, peaks = ctx.process_block(src[perm], slots[perm])
        ref = np.zeros((frames, 2))
        rpeaks = np.zeros((n, 2), np.float32)
        for k, o in enumerate(oras):
            sel = owner == k
            _, pk, m64 = o.block(p[sel].astype(ob.PARAMS_DTYPE), src[sel], want64=True)
            ref += m64[0]
            rpeaks[sel] = pk
        assert rel_rms(mix[0], ref) <= TOL, f"block {b}"
        np.testing.assert_allclose(peaks, rpeaks[perm], rtol=2e-5, atol=1e-7)
    ctx.close()


def test_staged_chain_single_instance_entry(gas, ob):
    """gas_process_frames_1 on a staged chain (the per-instance plugin call, audio_spatializer.h:146)."""
    from godot_audio_spatializer_amd import synth

    rng = np.random.default_rng(3)
    frames = 512
    hrir = _hrir()
    ctx = gas.SpatializerContext(max_sources=4, frames=frames)
    ctx.hrtf_load(hrir)
    (slot,) = ctx.source_alloc_many(1, gas.capi.KIND_EFFECT, (HS, HRTF))
    ora = ob.BatchOracle(ob.KIND_EFFECT, 1, frames, chain=(HS, HRTF), hrir=hrir)
    for b in range(4):
        p = synth.draw_params(rng, 1, dirs=32, frames=frames)
        ctx.params_publish_batch(np.asarray([slot], np.uint32), p)
        src = synth.draw_sources(rng, 1, frames)
        out = ctx.process_frames_1(slot, src[0])
        _, _, m64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
        assert rel_rms(out, m64[0]) <= TOL
    ctx.close()


def test_unsupported_chains_still_fail_loudly(gas):
    K = gas.capi
    ctx = gas.SpatializerContext(max_sources=4, frames=256, er_ring_frames=0)
    with pytest.raises(gas.GasError):
        ctx.source_alloc_many(1, K.KIND_EFFECT, (HS, ER))  # needs the ring
    with pytest.raises(gas.GasError):
        ctx.source_alloc_many(1, K.KIND_EFFECT, (HRTF, HS, HRTF))  # one HRTF history per playback
    with pytest.raises(gas.GasError):
        ctx.source_alloc_many(1, K.KIND_EFFECT, (HS, 10))  # no such effect kind
    ctx.close()


LP, HP, BP, NOTCH, LSH, AMP = 4, 5, 6, 7, 8, 9


@pytest.mark.parametrize(
    "chain,frames",
    [
        ((LP,), 512),
        ((HP, HRTF), 512),
        ((AMP,), 256),
        ((BP, AMP, HRTF), 256),
        ((NOTCH, LSH), 512),
        ((AMP, HS, LP, AMP), 128),
        ((ER, LP), 256),
    ],
)
def test_engine_effect_kinds_match_oracle(gas, ob, chain, frames):
    """further AudioEffect kinds behind the chain (audio_spatializer_effect.cpp:79-88 instantiates any
    AudioEffect): the engine's other one-biquad filters and the amplifier, settings per playback and chain position
    (gas_fx_settings), re-published between callbacks as a _process_effects script would."""
    from godot_audio_spatializer_amd import synth

    n, ring = 75, 4096 if ER in chain else 0
    hrir = _hrir() if HRTF in chain else None
    rng = np.random.default_rng(21)
    with gas.SpatializerContext(max_sources=n + 4, frames=frames, er_ring_frames=ring) as ctx:
        if hrir is not None:
            ctx.hrtf_load(hrir)
        slots = ctx.source_alloc_many(n, gas.capi.KIND_EFFECT, chain)
        ora = ob.BatchOracle(ob.KIND_EFFECT, n, frames, chain=chain, hrir=hrir, er_ring_frames=max(ring, 1))
        for b in range(9):
            if b % 3 == 0:
                p = synth.draw_params(rng, n, dirs=32, ring_frames=max(ring, 2 * frames), frames=frames)
                ctx.params_publish_batch(slots, p)
            if b in (1, 2, 5, 7):  # block 0 runs on the resource defaults; then the settings move, for some sources only
                who = np.arange(n) if b == 1 else rng.choice(n, n // 2, replace=False)
                st = ctx.fx_settings_defaults(len(who))
                st["filter_cutoff_hz"] = np.exp(rng.uniform(np.log(80.0), np.log(12000.0), (len(who), 4)))
                st["filter_resonance"] = rng.uniform(0.3, 2.0, (len(who), 4))
                st["filter_gain"] = np.exp(rng.uniform(np.log(0.1), np.log(3.0), (len(who), 4)))
                st["amplify_volume_db"] = rng.uniform(-18.0, 6.0, (len(who), 4))
                ctx.fx_settings_publish(slots[who], st)
                for k, s in enumerate(who):
                    for j in range(len(chain)):
                        ora.set_fx_settings(int(s), j, st["filter_cutoff_hz"][k, j], st["filter_resonance"][k, j], st["filter_gain"][k, j], st["amplify_volume_db"][k, j])
            src = synth.draw_sources(rng, n, frames)
            mix, peaks = ctx.process_block(src, slots)
            _, rpeaks, r64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
            assert rel_rms(mix[0], r64[0]) <= TOL, f"block {b}"
            # a recursive filter BEHIND early reflections / the FFT amplifies their last-bit differences from the oracle by
            # ~1 / (1 - r^2) (r = pole radius: 190x at 80 Hz, Q = 2): a single source's peak is then good to 1e-4, the mix
            # (errors average over sources and frames) still to TOL
            seen_new = False
            loose = False
            for k in chain:
                loose = loose or (seen_new and LP <= k <= LSH)
                seen_new = seen_new or k in (ER, HRTF)
            np.testing.assert_allclose(peaks, rpeaks, rtol=2e-4 if loose else 2e-5, atol=1e-7)


def test_engine_effect_settings_do_not_survive_a_slot(gas, ob):
    """A freed slot handed to a new playback starts from the resource defaults again (effect instances are created per
    playback, audio_spatializer_effect.cpp:79-88)."""
    from godot_audio_spatializer_amd import synth

    F = 256
    rng = np.random.default_rng(2)
    with gas.SpatializerContext(max_sources=2, frames=F) as ctx:
        p = synth.draw_params(rng, 1, dirs=8)
        src = synth.draw_sources(rng, 1, F)
        other = ctx.source_alloc_many(1, gas.capi.KIND_EFFECT, ())
        ctx.params_publish_batch(other, p)
        s0 = ctx.source_alloc_many(1, gas.capi.KIND_EFFECT, (AMP,))
        st = ctx.fx_settings_defaults(1)
        st["amplify_volume_db"] = -20.0
        ctx.fx_settings_publish(s0, st)
        ctx.params_publish_batch(s0, p)
        quiet, _ = ctx.process_block(src, s0)
        np.testing.assert_allclose(quiet[0], src[0] * np.float32(0.1), rtol=2e-6)
        ctx.source_free(int(s0[0]))
        ctx.process_block(src, other)  # a block boundary: the free takes effect
        s1 = ctx.source_alloc_many(1, gas.capi.KIND_EFFECT, (AMP,))
        assert s1[0] == s0[0]
        ctx.params_publish_batch(s1, p)
        loud, _ = ctx.process_block(src, s1)
        np.testing.assert_array_equal(loud[0], src[0])  # 0 dB, no ramp on a fresh instance
    with pytest.raises(gas.GasError):
        with gas.SpatializerContext(max_sources=2, frames=F) as ctx:
            ctx.source_alloc_many(1, gas.capi.KIND_EFFECT, (10,))  # unknown kind


@pytest.mark.parametrize("chain,frames", [((HS, HRTF), 512), ((LP, HRTF), 256), ((HS,), 512), ((NOTCH, HP), 128)])
def test_filter_stage_as_a_scan_matches_oracle(gas, ob, chain, frames, monkeypatch):
    """From 512 sources on (and below 32768) a rows-out filter stage runs k_shelf_scan: one wave per source, the block's
    recurrence split over the lanes and stitched by an affine scan -- another rounding than the engine's serial loop, so
    this is the test that says how far apart they are: the mix to TOL, a single source's peak to 2e-5 where the poles
    are well inside the unit circle (the default 5 kHz shelf: ~2e-6 measured).  Sources whose poles are close to it
    (r^2 > 0.9) take the kernel's serial path; some are mixed in here.  The same callbacks with GAS_SHELF_SCAN=0 (the
    serial stage kernel) are the control."""
    from godot_audio_spatializer_amd import synth

    # [filter, HRTF] has a third form, the default: both effects in one launch (k_hrtf_uni<FLT>, the filter on the mean of
    # the ears -- the HRTF's input -- with the scan's arithmetic); GAS_UNI_FLT=0 gives the two staged forms.  4500
    # sources: a wave of the one-launch kernel takes several in sequence.
    n = 4500 if chain == (HS, HRTF) else 1500
    hrir = _hrir() if HRTF in chain else None
    seen = {}
    for scan, flt in (("1", "1"), ("1", "0"), ("0", "0")) if HRTF in chain else (("1", "0"), ("0", "0")):
        monkeypatch.setenv("GAS_SHELF_SCAN", scan)
        monkeypatch.setenv("GAS_UNI_FLT", flt)
        rng = np.random.default_rng(31)
        with gas.SpatializerContext(max_sources=n, frames=frames) as ctx:
            if hrir is not None:
                ctx.hrtf_load(hrir)
            slots = ctx.source_alloc_many(n, gas.capi.KIND_EFFECT, chain)
            ora = ob.BatchOracle(ob.KIND_EFFECT, n, frames, chain=chain, hrir=hrir)
            worst = 0.0
            for b in range(6):
                if b % 2 == 0:
                    p = synth.draw_params(rng, n, dirs=32, frames=frames)
                    p["fx_shelf_cutoff_hz"] = np.exp(rng.uniform(np.log(800.0), np.log(12000.0), n)).astype(np.float32)
                    p["fx_shelf_cutoff_hz"][::97] = 60.0  # poles next to the unit circle: the serial path
                    ctx.params_publish_batch(slots, p)
                    st = ctx.fx_settings_defaults(n)
                    st["filter_cutoff_hz"] = np.exp(rng.uniform(np.log(500.0), np.log(12000.0), (n, 4)))
                    st["filter_cutoff_hz"][::89] = 90.0
                    st["filter_resonance"] = rng.uniform(0.4, 1.2, (n, 4))
                    ctx.fx_settings_publish(slots, st)
                    for s_ in range(n):
                        for j in range(len(chain)):
                            ora.set_fx_settings(s_, j, st["filter_cutoff_hz"][s_, j], st["filter_resonance"][s_, j], 1.0, 0.0)
                src = synth.draw_sources(rng, n, frames)
                mix, peaks = ctx.process_block(src, slots)
                _, rpeaks, r64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
                worst = max(worst, rel_rms(mix[0], r64[0]))
                assert rel_rms(mix[0], r64[0]) <= TOL, f"scan={scan} flt={flt} block {b}"
                np.testing.assert_allclose(peaks, rpeaks, rtol=1e-4, atol=1e-6, err_msg=f"scan={scan} flt={flt} block {b}")
                seen.setdefault((scan, flt), []).append((mix.copy(), peaks.copy()))
    if HRTF in chain:  # the one-launch form runs k_shelf_scan's operations in k_shelf_scan's order: the two-launch form's bits
        for (m1, p1), (m0, p0) in zip(seen[("1", "1")], seen[("1", "0")]):
            np.testing.assert_array_equal(m1, m0)
            np.testing.assert_array_equal(p1, p0)


@pytest.mark.parametrize("frames", [128, 256, 384])
@pytest.mark.parametrize("chain", [(HS, HRTF), (BP, HRTF), (HS, ER, HRTF)])
def test_one_launch_chain_tails_at_every_block_size(gas, ob, chain, frames):
    """k_hrtf_uni<FLT> ([filter, HRTF]) and k_hrtf_uni<ER> behind a filter stage ([.., ER, HRTF]) at the block sizes the
    other tests leave out, 2600 playbacks (waves with one and with two sources), a sixth of them draining under
    GAS_FLAG_PEAKS_DRAINING_ONLY."""
    from test_gpu_parity import run_pair

    ring = 2048 if ER in chain else 0
    run_pair(gas, ob, gas.capi.KIND_EFFECT, chain, 2600, frames, 4, hrir=_hrir(), ring=ring, dirs=32, redraw_every=2, flags=gas.capi.FLAG_PEAKS_DRAINING_ONLY, draining_every=6)


def test_scan_forms_after_a_switch_to_an_ill_conditioned_filter(gas, ob, monkeypatch):
    """What the scan forms (k_shelf_scan, k_hrtf_uni<FLT>) cost in parity, stated as a test.  Their processor history
    differs from the engine's serial loop in the last bits (~1e-6 relative: other association, powers of M); a filter
    set LATER on the same playback with poles next to the unit circle (here: the 2 kHz default, then a 95 Hz high-pass
    at Q 1.9) amplifies any history difference ~1 / (1 - r), and the callbacks after the switch leave the 1e-5 band
    (measured 3e-5; bounded here by 2e-4) until the difference has rung out.  GAS_SHELF_SCAN=0 selects the engine-order
    kernels for every callback size and stays inside the band throughout."""
    from godot_audio_spatializer_amd import synth

    n, frames, chain = 600, 512, (HP, HRTF)
    hrir = _hrir()
    for scan, bound in (("0", TOL), ("1", 2e-4)):
        monkeypatch.setenv("GAS_SHELF_SCAN", scan)
        rng = np.random.default_rng(41)
        with gas.SpatializerContext(max_sources=n, frames=frames) as ctx:
            ctx.hrtf_load(hrir)
            slots = ctx.source_alloc_many(n, gas.capi.KIND_EFFECT, chain)
            ora = ob.BatchOracle(ob.KIND_EFFECT, n, frames, chain=chain, hrir=hrir)
            p = synth.draw_params(rng, n, dirs=32, frames=frames)
            ctx.params_publish_batch(slots, p)
            for b in range(5):
                if b == 2:
                    st = ctx.fx_settings_defaults(n)
                    st["filter_cutoff_hz"][:] = 95.0
                    st["filter_resonance"][:] = 1.9
                    ctx.fx_settings_publish(slots, st)
                    for s_ in range(n):
                        ora.set_fx_settings(s_, 0, 95.0, 1.9, 1.0, 0.0)
                src = synth.draw_sources(rng, n, frames)
                mix, _ = ctx.process_block(src, slots)
                _, _, r64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
                assert rel_rms(mix[0], r64[0]) <= (TOL if b < 2 else bound), f"scan={scan} block {b}"
