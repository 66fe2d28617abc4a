"""Seeded random configurations through the C ABI against the oracle: source counts from 1 up, every frame count
the HRTF path supports, every fused and a few staged chains, every flag combination, random draining sets, random
callback orders.  Catches the corner a hand-written case forgets (one source, fewer sources than waves, groups that
are all draining, runs of equal directions across wave borders, ...)."""
import numpy as np
import pytest

from helpers import TOL, rel_rms

pytestmark = pytest.mark.gpu

HS, ER, HRTF = 1, 2, 3
CHAINS = [(), (HS,), (ER,), (HRTF,), (ER, HRTF), (HS, HRTF), (HRTF, HS), (HS, ER, HRTF)]


@pytest.mark.parametrize("seed", range(40))
def test_random_configuration(gas, ob, seed):
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    rng = np.random.default_rng(1000 + seed)
    chain = CHAINS[int(rng.integers(len(CHAINS)))]
    frames = int(rng.choice([128, 256, 384, 512]))  # what gas_ctx_create accepts
    n = int(rng.choice([1, 2, 7, 8, 9, 63, 65, 200, 513, 1100]))
    dirs = int(rng.choice([1, 3, 16, 97]))
    ring = 2048 if ER in chain else 0
    flags = 0
    fused_hrtf = chain in ((HRTF,), (ER, HRTF))
    if rng.random() < 0.6:
        flags |= K.FLAG_PEAKS_DRAINING_ONLY
    if HRTF in chain and rng.random() < 0.3:
        flags |= K.FLAG_HRTF_CROSSFADE
    if rng.random() < 0.4:
        flags |= K.FLAG_DIRECTION_RUNS
    if rng.random() < 0.3:
        flags |= K.FLAG_DIRECTION_ORDER
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=dirs) if HRTF in chain else None
    ctx = gas.SpatializerContext(max_sources=n, frames=frames, er_ring_frames=ring, flags=flags)
    if hrir is not None:
        ctx.hrtf_load(hrir)
    slots = ctx.source_alloc_many(n, K.KIND_EFFECT, chain)
    ora = ob.BatchOracle(ob.KIND_EFFECT, n, frames, chain=chain, hrir=hrir, er_ring_frames=max(ring, 1), crossfade=bool(flags & K.FLAG_HRTF_CROSSFADE))
    draining = rng.random(n) < float(rng.choice([0.0, 0.1, 1.0]))
    for s in slots[draining]:
        ctx.source_set_draining(s, True)
    # which chains leave a non-draining source's peak unmeasured under GAS_FLAG_PEAKS_DRAINING_ONLY: the fused HRTF chains,
    # and staged chains whose last stage is the one-launch HRTF kernel (no cross-fade / direction-run flags)
    one_launch = not (flags & (K.FLAG_HRTF_CROSSFADE | K.FLAG_DIRECTION_RUNS | K.FLAG_DIRECTION_ORDER))
    skips_peaks = fused_hrtf or (one_launch and len(chain) >= 2 and chain[-1] == HRTF)
    exact = draining | (not (flags & K.FLAG_PEAKS_DRAINING_ONLY)) | (not skips_peaks)
    p = None
    for b in range(5):
        if b == 0 or rng.random() < 0.6:
            p = synth.draw_params(rng, n, dirs=dirs, ring_frames=max(ring, 2 * frames), frames=frames)
            if rng.random() < 0.5:
                p["hrtf_dir"] = np.sort(p["hrtf_dir"])  # runs of equal directions
            ctx.params_publish_batch(slots, p)
        src = synth.draw_sources(rng, n, frames)
        perm = rng.permutation(n) if rng.random() < 0.5 else np.arange(n)
        mix, peaks = ctx.process_block(src[perm], slots[perm])
        _, rpeaks, r64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
        assert rel_rms(mix[0], r64[0]) <= TOL, f"seed {seed} block {b}: chain {chain} n {n} F {frames} dirs {dirs} flags {flags}"
        ex = exact[perm]
        np.testing.assert_allclose(peaks[ex], rpeaks[perm][ex], rtol=3e-5, atol=1e-7)
        assert np.all(np.isposinf(peaks[~ex]))
    ctx.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_3d_configuration(gas, ob, seed):
    """The two AudioSpatializerInstance3D entry points with random channel-pair counts, source counts, bypassed and
    silent sources (the history-clear and gain-only branches, audio_spatializer_3d.cpp:518-521,599-605) and filter
    settings from 40 Hz to just under Nyquist."""
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    rng = np.random.default_rng(5000 + seed)
    kind = K.KIND_3D_MIX if rng.random() < 0.7 else K.KIND_3D_PROCESS
    C = int(rng.integers(1, 5))
    frames = int(rng.choice([128, 256, 384, 512]))
    n = int(rng.choice([1, 5, 31, 32, 33, 250, 700]))
    ctx = gas.SpatializerContext(max_sources=n, frames=frames, channel_count=C)
    slots = ctx.source_alloc_many(n, kind)
    ora = ob.BatchOracle(kind, n, frames, channel_count=C)
    Cn = C if kind == K.KIND_3D_MIX else 1
    p = None
    for b in range(6):
        if b == 0 or rng.random() < 0.7:
            p = synth.draw_params(rng, n, channel_count=C, frames=frames)
            sel = rng.random(n)
            p["linear_attenuation"][sel < 0.2] = 0.0005  # bypass branch
            p["linear_attenuation"][(sel >= 0.2) & (sel < 0.3)] = 1.0
            p["attenuation_filter_cutoff_hz"] = np.exp(rng.uniform(np.log(40.0), np.log(23000.0), n))
            silent = rng.random(n) < 0.15
            p["mix_volumes"][silent] = 0.0  # next audible block starts from cleared filter history
            ctx.params_publish_batch(slots, p)
        src = synth.draw_sources(rng, n, frames)
        mix, peaks = ctx.process_block(src, slots)
        rmix, rpeaks, r64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
        for c in range(Cn):
            assert rel_rms(mix[c], r64[c]) <= TOL, f"seed {seed} block {b} channel {c}: kind {kind} C {C} n {n} F {frames}"
        np.testing.assert_allclose(peaks, rpeaks, rtol=3e-5, atol=1e-7)
    ctx.close()
