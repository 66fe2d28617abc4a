"""GPU parity: the HIP path through the C ABI vs the CPU oracle on identical seeded inputs.

Tolerance: 1e-5 relative RMS per callback (BASELINE.json north_star), peaks to 1e-5 relative.
PARITY UNPINNED at the engine boundary (oracle/gas_oracle.h).
"""
import numpy as np
import pytest

from helpers import TOL, rel_rms

pytestmark = pytest.mark.gpu

ORACLE_KIND = {0: 0, 1: 1, 2: 2}


def run_pair(gas, ob, kind, chain, n, frames, blocks, channel_count=1, seed=0, dirs=64, redraw_every=2, ring=0, hrir=None, params_hook=None, flags=0, draining_every=0):
    from godot_audio_spatializer_amd import synth

    rng = np.random.default_rng(seed)
    C = channel_count if kind == gas.capi.KIND_3D_MIX else 1
    ctx = gas.SpatializerContext(max_sources=n + 8, frames=frames, channel_count=channel_count, er_ring_frames=ring, flags=flags)
    if hrir is not None:
        ctx.hrtf_load(hrir)
    slots = ctx.source_alloc_many(n, kind, chain)
    ora = ob.BatchOracle(kind, n, frames, channel_count=channel_count, chain=chain, hrir=hrir, er_ring_frames=max(ring, 1), crossfade=bool(flags & gas.capi.FLAG_HRTF_CROSSFADE))
    exact = np.ones(n, bool)
    if flags & gas.capi.FLAG_PEAKS_DRAINING_ONLY:
        exact[:] = False
        if draining_every:
            exact[::draining_every] = True
            for s_ in slots[exact]:
                ctx.source_set_draining(s_, True)
    worst = 0.0
    for b in range(blocks):
        if b % redraw_every == 0:
            p = synth.draw_params(rng, n, dirs=dirs, channel_count=channel_count, ring_frames=max(ring, 2 * frames), frames=frames)
            if params_hook:
                params_hook(b, p)
            ctx.params_publish_batch(slots, p)
        src = synth.draw_sources(rng, n, frames)
        mix, peaks = ctx.process_block(src, slots)
        rmix, rpeaks, rmix64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
        assert mix.shape == (channel_count, frames, 2)
        for c in range(C):
            e = rel_rms(mix[c], rmix64[c])
            worst = max(worst, e)
            assert e <= TOL, f"block {b} channel {c}: rel rms {e}"
        for c in range(C, channel_count):
            assert not mix[c].any()
        np.testing.assert_allclose(peaks[exact], rpeaks[exact], rtol=2e-5, atol=1e-7)
        assert np.all(np.isposinf(peaks[~exact]))
    ctx.close()
    return worst


@pytest.mark.parametrize("n", [1, 31, 256])
def test_mix_channel_cfg2(gas, ob, n):
    """cfg2: pan ramp + distance high-shelf (mix_channel), F=512, stereo."""
    run_pair(gas, ob, gas.capi.KIND_3D_MIX, (), n, 512, 6)


def test_mix_channel_four_pairs(gas, ob):
    run_pair(gas, ob, gas.capi.KIND_3D_MIX, (), 96, 512, 5, channel_count=4)


def test_mix_channel_bypass_branch(gas, ob):
    """linear_attenuation < 0.001 takes the gain-only branch (audio_spatializer_3d.cpp:599-605) for some sources."""

    def hook(b, p):
        p["linear_attenuation"][::3] = 0.0005
        if b >= 2:
            p["linear_attenuation"][1::3] = 0.0

    run_pair(gas, ob, gas.capi.KIND_3D_MIX, (), 100, 512, 6, params_hook=hook)


def test_process_frames_mode(gas, ob):
    run_pair(gas, ob, gas.capi.KIND_3D_PROCESS, (), 130, 512, 6)


def test_process_frames_256(gas, ob):
    run_pair(gas, ob, gas.capi.KIND_3D_PROCESS, (), 70, 256, 5)


def test_effect_empty_chain(gas, ob):
    run_pair(gas, ob, gas.capi.KIND_EFFECT, (), 40, 512, 2)


def test_effect_highshelf(gas, ob):
    run_pair(gas, ob, gas.capi.KIND_EFFECT, (gas.capi.FX_HIGHSHELF,), 77, 512, 5)


@pytest.mark.parametrize("frames,n", [(512, 1), (512, 67), (256, 130), (512, 300), (128, 90), (384, 75)])
def test_hrtf(gas, ob, frames, n):
    """cfg3 shape at oracle-friendly size: per-source 256-tap HRTF, direction redrawn every 2 callbacks."""
    from godot_audio_spatializer_amd import synth

    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=64)
    run_pair(gas, ob, gas.capi.KIND_EFFECT, (gas.capi.FX_HRTF,), n, frames, 5, hrir=hrir)


def test_er_hrtf_cfg5(gas, ob):
    """cfg5 shape: 8-tap early reflections -> HRTF, F=256, ring 4096."""
    from godot_audio_spatializer_amd import synth

    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=64)
    run_pair(gas, ob, gas.capi.KIND_EFFECT, (gas.capi.FX_EARLY_REFLECTIONS, gas.capi.FX_HRTF), 90, 256, 20, hrir=hrir, ring=4096, redraw_every=3)


@pytest.mark.parametrize("uni_er", ["2", "0"])
def test_er_hrtf_both_kernels_many_sources_per_wave(gas, ob, monkeypatch, uni_er):
    """[ER, HRTF] through k_hrtf_uni<ER> (GAS_UNI_ER=2: always; by default when at most a quarter of the playbacks
    want their exact peak) and through k_hrtf_ols<ER> (GAS_UNI_ER=0; read when the context is made), at a size where a wave handles several sources in sequence; exact peaks only for the draining sixth, so the
    launch carries the frequency-domain group and the exact-peak group behind it (peak_from)."""
    from godot_audio_spatializer_amd import synth

    monkeypatch.setenv("GAS_UNI_ER", uni_er)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=64)
    run_pair(gas, ob, gas.capi.KIND_EFFECT, (gas.capi.FX_EARLY_REFLECTIONS, gas.capi.FX_HRTF), 5000, 256, 4, hrir=hrir, ring=4096, redraw_every=2, flags=gas.capi.FLAG_PEAKS_DRAINING_ONLY, draining_every=6)
    run_pair(gas, ob, gas.capi.KIND_EFFECT, (gas.capi.FX_EARLY_REFLECTIONS, gas.capi.FX_HRTF), 2100, 512, 3, hrir=hrir, ring=2048, redraw_every=2)


def test_er_only(gas, ob):
    run_pair(gas, ob, gas.capi.KIND_EFFECT, (gas.capi.FX_EARLY_REFLECTIONS,), 50, 256, 20, ring=4096, redraw_every=3)


@pytest.mark.parametrize("frames", [128, 384])
def test_other_block_sizes(gas, ob, frames):
    """The context accepts any multiple of 128 up to 512 frames per callback; 512 is the engine's mix step."""
    run_pair(gas, ob, gas.capi.KIND_3D_MIX, (), 40, frames, 4)
    run_pair(gas, ob, gas.capi.KIND_EFFECT, (gas.capi.FX_EARLY_REFLECTIONS, gas.capi.FX_HRTF), 33, frames, 40 if frames == 128 else 14, hrir=__import__("godot_audio_spatializer_amd").synth.synthetic_hrir(np.random.default_rng(7), dirs=64), ring=4096, redraw_every=3)


@pytest.mark.parametrize("frames,chain,ring", [(512, (3,), 0), (256, (3,), 0), (256, (2, 3), 4096)])
def test_hrtf_frequency_domain_path(gas, ob, frames, chain, ring):
    """GAS_FLAG_PEAKS_DRAINING_ONLY: non-draining sources are summed in the frequency domain.  The mix must
    match the oracle exactly as before; draining sources report exact peaks, the others +inf."""
    from godot_audio_spatializer_amd import synth

    rng = np.random.default_rng(3)
    n = 150
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=64)
    ctx = gas.SpatializerContext(max_sources=n, frames=frames, er_ring_frames=ring, flags=gas.capi.FLAG_PEAKS_DRAINING_ONLY)
    ctx.hrtf_load(hrir)
    slots = ctx.source_alloc_many(n, gas.capi.KIND_EFFECT, chain)
    draining = np.zeros(n, bool)
    ora = ob.BatchOracle(ob.KIND_EFFECT, n, frames, chain=chain, hrir=hrir, er_ring_frames=max(ring, 1))
    for b in range(6):
        if b == 2:  # some streams end: their peaks become observable
            draining[::7] = True
            for s in slots[draining]:
                ctx.source_set_draining(s, True)
        if b == 4:
            ctx.source_set_draining(slots[7], False)
            draining[7] = False
        if b % 2 == 0:
            p = synth.draw_params(rng, n, dirs=64, ring_frames=max(ring, 2 * frames), frames=frames)
            ctx.params_publish_batch(slots, p)
        src = synth.draw_sources(rng, n, frames)
        mix, peaks = ctx.process_block(src, slots)
        _, rpeaks, r64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
        assert rel_rms(mix[0], r64[0]) <= TOL
        np.testing.assert_allclose(peaks[draining], rpeaks[draining], rtol=2e-5, atol=1e-7)
        assert np.all(np.isposinf(peaks[~draining]))
    ctx.close()


@pytest.mark.parametrize("mode", ["exact_peaks", "frequency_domain", "er_chain"])
@pytest.mark.parametrize("frames", [512, 256])
def test_hrtf_direction_crossfade(gas, ob, mode, frames):
    """SURVEY.md 8f#4 (GAS_FLAG_HRTF_CROSSFADE): directions are redrawn every 2 callbacks, so half the callbacks
    cross-fade every source's HRIR pair and the other half take the unchanged-direction path."""
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=64)
    flags = K.FLAG_HRTF_CROSSFADE | (K.FLAG_PEAKS_DRAINING_ONLY if mode == "frequency_domain" else 0)
    chain = (K.FX_EARLY_REFLECTIONS, K.FX_HRTF) if mode == "er_chain" else (K.FX_HRTF,)
    run_pair(gas, ob, K.KIND_EFFECT, chain, 150, frames, 8, hrir=hrir, flags=flags, draining_every=9, ring=4096 if mode == "er_chain" else 0)


@pytest.mark.parametrize("cutoff", [50.0, 500.0, 20500.0])
@pytest.mark.parametrize("gain", [0.0011, 0.05, 1.0])
def test_biquad_extreme_filter_settings(gas, ob, cutoff, gain):
    """Low cutoffs / small shelf gains put the poles almost on the unit circle, where the f32 recurrence amplifies any
    rounding difference (an FMA in place of mul + add: up to 9e-4).  The kernel is compiled without FMA contraction so
    it rounds like the reference arithmetic and stays inside the 1e-5 bar at every legal setting."""

    def hook(b, p):
        p["linear_attenuation"] = gain
        p["fx_shelf_gain"] = gain
        p["attenuation_filter_cutoff_hz"] = cutoff
        p["fx_shelf_cutoff_hz"] = cutoff

    run_pair(gas, ob, gas.capi.KIND_3D_MIX, (), 48, 512, 10, params_hook=hook)
    run_pair(gas, ob, gas.capi.KIND_EFFECT, (gas.capi.FX_HIGHSHELF,), 40, 512, 6, params_hook=hook)


@pytest.mark.parametrize("n,dirs,chain,frames,ring", [(1100, 32, (3,), 512, 0), (9000, 64, (3,), 256, 0), (700, 16, (2, 3), 256, 4096), (7000, 3000, (3,), 256, 0)])
def test_hrtf_direction_ordered_groups(gas, ob, n, dirs, chain, frames, ring):
    """Frequency-domain groups large enough to be direction-ordered (k_dir_order): sources sharing an HRIR
    direction are summed before ONE forward FFT.  Callbacks alternate between fresh parameters (order rebuilt),
    unchanged parameters (order reused) and a changed slot list; draining sources keep the exact-peak path."""
    from godot_audio_spatializer_amd import synth

    rng = np.random.default_rng(n)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=dirs)
    ctx = gas.SpatializerContext(max_sources=n, frames=frames, er_ring_frames=ring, flags=gas.capi.FLAG_PEAKS_DRAINING_ONLY | gas.capi.FLAG_DIRECTION_ORDER)
    ctx.hrtf_load(hrir)
    slots = ctx.source_alloc_many(n, gas.capi.KIND_EFFECT, chain)
    draining = np.zeros(n, bool)
    draining[5::37] = True
    for s in slots[draining]:
        ctx.source_set_draining(s, True)
    ora = ob.BatchOracle(ob.KIND_EFFECT, n, frames, chain=chain, hrir=hrir, er_ring_frames=max(ring, 1))
    perm = np.arange(n)
    for b in range(5):
        if b in (0, 1, 3):
            p = synth.draw_params(rng, n, dirs=dirs, ring_frames=max(ring, 2 * frames), frames=frames)
            ctx.params_publish_batch(slots, p)
        if b == 4:
            perm = rng.permutation(n)  # same playbacks, another callback order
        src = synth.draw_sources(rng, n, frames)
        mix, peaks = ctx.process_block(src[perm], slots[perm])
        _, rpeaks, r64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
        assert rel_rms(mix[0], r64[0]) <= TOL, f"block {b}"
        np.testing.assert_allclose(peaks[draining[perm]], rpeaks[perm][draining[perm]], rtol=2e-5, atol=1e-7)
        assert np.all(np.isposinf(peaks[~draining[perm]]))
    ctx.close()


@pytest.mark.parametrize("chain,frames,ring", [((3,), 512, 0), ((2, 3), 256, 4096)])
def test_hrtf_direction_runs_in_the_callers_list(gas, ob, chain, frames, ring):
    """GAS_FLAG_DIRECTION_RUNS: the caller's list has runs of equal directions (here: long runs, runs of one, and a
    run that crosses wave and workgroup boundaries); each run is summed before one forward FFT."""
    from godot_audio_spatializer_amd import synth

    rng = np.random.default_rng(5)
    n, dirs = 333, 24
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=dirs)
    K = gas.capi
    ctx = gas.SpatializerContext(max_sources=n, frames=frames, er_ring_frames=ring, flags=K.FLAG_PEAKS_DRAINING_ONLY | K.FLAG_DIRECTION_RUNS)
    ctx.hrtf_load(hrir)
    slots = ctx.source_alloc_many(n, K.KIND_EFFECT, chain)
    ctx.source_set_draining(slots[100], True)
    ora = ob.BatchOracle(ob.KIND_EFFECT, n, frames, chain=chain, hrir=hrir, er_ring_frames=max(ring, 1))
    for b in range(5):
        p = synth.draw_params(rng, n, dirs=dirs, ring_frames=max(ring, 2 * frames), frames=frames)
        d = np.sort(p["hrtf_dir"])
        d[:90] = 3  # one run across many waves
        if b % 2:
            d[200:] = rng.integers(0, dirs, n - 200)  # runs of one behind it
        p["hrtf_dir"] = d
        ctx.params_publish_batch(slots, p)
        src = synth.draw_sources(rng, n, frames)
        mix, peaks = ctx.process_block(src, slots)
        _, rpeaks, r64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
        assert rel_rms(mix[0], r64[0]) <= TOL, f"block {b}"
        np.testing.assert_allclose(peaks[100], rpeaks[100], rtol=2e-5, atol=1e-7)
    ctx.close()


@pytest.mark.parametrize("kind_name", ["mix_channel", "process_frames", "fx_highshelf"])
@pytest.mark.parametrize("n,frames,channels", [(256, 512, 1), (37, 384, 2), (1000, 256, 1)])
def test_biquad_pipeline_is_bitwise_the_single_wave_kernel(gas, kind_name, n, frames, channels):
    """k_biquad_pipe (eight-wave software pipeline, small callbacks) against k_biquad_mix (one wave per 32 sources):
    same operations, same rounding points -> the mixes, the peaks and the carried state (three callbacks, parameters
    changing, some sources on the bypass branch) must agree to the bit.  GAS_BIQUAD_PIPE=0 selects the old kernel."""
    import os

    from godot_audio_spatializer_amd import synth

    K = gas.capi
    kind, chain = {"mix_channel": (K.KIND_3D_MIX, ()), "process_frames": (K.KIND_3D_PROCESS, ()), "fx_highshelf": (K.KIND_EFFECT, (K.FX_HIGHSHELF,))}[kind_name]
    C = channels if kind == K.KIND_3D_MIX else 1

    def render(pipe):
        os.environ["GAS_BIQUAD_PIPE"] = "1" if pipe else "0"
        rng = np.random.default_rng(11)
        outs = []
        with gas.SpatializerContext(max_sources=n, frames=frames, channel_count=channels) as ctx:
            slots = ctx.source_alloc_many(n, kind, chain)
            for cb in range(3):
                p = synth.draw_params(rng, n, channel_count=channels, frames=frames)
                p["linear_attenuation"][::5] = 0.0005  # bypass branch (:599-605) for every fifth source
                ctx.params_publish_batch(slots, p)
                mix, peaks = ctx.process_block(synth.draw_sources(rng, n, frames), slots)
                outs.append((mix[:C].copy(), peaks.copy()))
        return outs

    try:
        a, b = render(True), render(False)
    finally:
        os.environ.pop("GAS_BIQUAD_PIPE", None)
    for (ma, pa), (mb, pb) in zip(a, b):
        assert np.isfinite(ma).all() and np.abs(ma).max() > 0
        np.testing.assert_array_equal(ma, mb)
        np.testing.assert_array_equal(pa, pb)
