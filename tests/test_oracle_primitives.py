"""The CPU oracle against checks that share no code with it (SURVEY.md section 8c ii).

PARITY UNPINNED: the reference has no tests/fixtures; these pin the restatement to the published
algorithms (RBJ high-shelf identities, scipy.signal.lfilter, closed forms, float64 numpy convolution).
"""
import ctypes as C

import numpy as np
import pytest
from scipy.signal import lfilter

from helpers import rel_rms


def coeffs(ob, sr, cutoff, gain, q=1.0, stages=1):
    c = ob.Coeffs()
    ob.lib().gaso_highshelf_coeffs(sr, cutoff, q, gain, stages, C.byref(c))
    b = np.array([c.b0, c.b1, c.b2], dtype=np.float64)
    a = np.array([1.0, -c.a1, -c.a2], dtype=np.float64)  # a1, a2 are stored negated (Appendix B)
    return b, a


def test_db_to_linear(ob):
    L = ob.lib()
    assert L.gaso_db_to_linear(-80.0) == pytest.approx(1e-4, rel=1e-6)  # audio_spatializer.cpp:465 threshold
    assert L.gaso_db_to_linear(-24.0) == pytest.approx(10 ** (-24 / 20), rel=1e-6)
    assert L.gaso_db_to_linear(0.0) == 1.0
    assert L.gaso_linear_to_db(L.gaso_db_to_linear(-12.5)) == pytest.approx(-12.5, abs=1e-4)


@pytest.mark.parametrize("gain", [0.001, 0.0631, 0.25, 0.7, 1.0])
def test_highshelf_identities(ob, gain):
    b, a = coeffs(ob, 48000.0, 5000.0, gain)
    assert b.sum() / a.sum() == pytest.approx(1.0, abs=3e-4)  # DC gain 1 (f32-rounded coefficients; poles near z=1 at small A)
    nyq = (b[0] - b[1] + b[2]) / (a[0] - a[1] + a[2])
    assert nyq == pytest.approx(gain * gain, rel=2e-3)  # Nyquist gain A^2
    assert np.all(np.abs(np.roots(a)) < 1.0)  # stable
    if gain == 1.0:
        np.testing.assert_allclose(b, a, atol=1e-7)  # A = 1: passthrough


def test_highshelf_matches_rbj_formula(ob):
    sr, fc, A = 48000.0, 5000.0, 0.3
    w = 2 * np.pi * fc / sr
    beta = np.sqrt(A)  # Q = 1
    a0 = (A + 1) - (A - 1) * np.cos(w) + beta * np.sin(w)
    rb = np.array([A * ((A + 1) + (A - 1) * np.cos(w) + beta * np.sin(w)), -2 * A * ((A - 1) + (A + 1) * np.cos(w)), A * ((A + 1) + (A - 1) * np.cos(w) - beta * np.sin(w))]) / a0
    ra = np.array([a0, 2 * ((A - 1) - (A + 1) * np.cos(w)), (A + 1) - (A - 1) * np.cos(w) - beta * np.sin(w)]) / a0
    b, a = coeffs(ob, sr, fc, A)
    np.testing.assert_allclose(b, rb, rtol=1e-6)
    np.testing.assert_allclose(a, ra, rtol=1e-6)


def test_cutoff_and_gain_clamps(ob):
    hi, _ = coeffs(ob, 48000.0, 1e9, 0.5)
    lim, _ = coeffs(ob, 48000.0, 24000.0 + 512.0, 0.5)  # sr/2 + 512
    np.testing.assert_array_equal(hi, lim)
    lo, _ = coeffs(ob, 48000.0, 0.0, 0.5)
    one, _ = coeffs(ob, 48000.0, 1.0, 0.5)
    np.testing.assert_array_equal(lo, one)
    g0, _ = coeffs(ob, 48000.0, 5000.0, 0.0)
    g1, _ = coeffs(ob, 48000.0, 5000.0, 0.001)
    np.testing.assert_array_equal(g0, g1)


def params(ob, n, **kw):
    p = np.zeros(n, ob.PARAMS_DTYPE)
    p["pitch_scale"] = 1.0
    p["attenuation_filter_cutoff_hz"] = 5000.0
    for k, v in kw.items():
        p[k] = v
    return p


def test_steady_state_biquad_vs_lfilter(ob):
    """After the coefficient ramp has settled the per-sample-interpolated filter is a plain biquad."""
    F, rng = 512, np.random.default_rng(0)
    ora = ob.BatchOracle(ob.KIND_3D_PROCESS, 1, F)
    p = params(ob, 1, linear_attenuation=0.2)
    p["mix_volumes"][:, 0] = [0.6, 0.4]
    x = rng.uniform(-0.5, 0.5, (40, 1, F, 2)).astype(np.float32)
    outs = [ora.block(p, x[i])[0][0] for i in range(40)]
    st = ora.states[0].pd3d.filter_processors[0]
    b, a = coeffs(ob, 48000.0, 5000.0, 0.2)
    # coefficients converge geometrically towards the target (each block closes the remaining gap by ~F/F)
    assert abs(st.coeffs.b0 - b[0]) < 1e-5
    y = np.concatenate(outs)[:, 0]
    ref = lfilter(b, a, np.concatenate([x[i, 0, :, 0] for i in range(40)]).astype(np.float64))
    # compare the last blocks, where the interpolated coefficients equal the target to f32 precision
    assert rel_rms(y[-4 * F:], ref[-4 * F:]) < 5e-4


def test_first_block_ramps_from_zero_coefficients(ob):
    """Processor coefficients start at zero: the first output sample is exactly 0 (x * b0 with b0 = 0)."""
    F = 512
    ora = ob.BatchOracle(ob.KIND_3D_MIX, 1, F)
    p = params(ob, 1, linear_attenuation=0.5)
    p["mix_volumes"][:, 0] = [1.0, 1.0]
    src = np.ones((1, F, 2), np.float32)
    mix, peaks, _ = ora.block(p, src)
    assert mix[0, 0, 0] == 0.0 and mix[0, 0, 1] == 0.0
    assert mix[0, 1, 0] != 0.0 or mix[0, 2, 0] != 0.0
    inc = ora.states[0].pd3d.filter_processors[0]
    b, a = coeffs(ob, 48000.0, 5000.0, 0.5)
    assert inc.incr.b0 == pytest.approx(b[0] / F, rel=1e-6)


def test_mix_channel_lerp_never_reaches_target(ob):
    """Bypass branch (gain < 0.001): out = (v1 * t + (1 - t) * v0) * x with t = i / F in [0, (F-1)/F]."""
    F = 512
    ora = ob.BatchOracle(ob.KIND_3D_MIX, 1, F)
    p = params(ob, 1, linear_attenuation=0.0)
    p["mix_volumes"][:, 0] = [0.8, 0.2]
    src = np.ones((1, F, 2), np.float32)
    m1, _, _ = ora.block(p, src)
    t = (np.arange(F, dtype=np.float32) / np.float32(F)).astype(np.float32)
    np.testing.assert_array_equal(m1[0, :, 0], np.float32(0.8) * t + (1 - t) * np.float32(0.0))
    assert m1[0, -1, 0] < 0.8
    p["mix_volumes"][:, 0] = [0.4, 0.6]
    m2, _, _ = ora.block(p, src)
    np.testing.assert_array_equal(m2[0, :, 0], np.float32(0.4) * t + (1 - t) * np.float32(0.8))  # starts from the previous target
    # the processors were never touched in the bypass branch
    assert ora.states[0].pd3d.filter_processors[0].ha1 == 0.0


def test_history_cleared_when_previous_volume_is_zero(ob):
    F = 512
    rng = np.random.default_rng(1)
    ora = ob.BatchOracle(ob.KIND_3D_MIX, 1, F)
    p = params(ob, 1, linear_attenuation=0.3)
    p["mix_volumes"][:, 0] = [0.5, 0.5]
    ora.block(p, rng.uniform(-0.5, 0.5, (1, F, 2)).astype(np.float32))
    assert ora.states[0].pd3d.filter_processors[0].ha1 != 0.0
    p["mix_volumes"][:, 0] = [0.0, 0.0]
    ora.block(p, rng.uniform(-0.5, 0.5, (1, F, 2)).astype(np.float32))  # prev becomes (0, 0)
    p["mix_volumes"][:, 0] = [0.5, 0.5]
    ora.block(p, np.zeros((1, F, 2), np.float32))  # is_just_started -> history cleared, zero input
    st = ora.states[0].pd3d.filter_processors[0]
    assert st.ha1 == 0.0 and st.hb1 == 0.0


def test_process_frames_prev_volume_is_max_pair(ob):
    F = 512
    ora = ob.BatchOracle(ob.KIND_3D_PROCESS, 1, F)
    p = params(ob, 1, linear_attenuation=0.5)
    p["mix_volumes"][0] = [[0.1, 0.2], [0.05, 0.9], [0.9, 0.3], [0.0, 0.0]]  # strict '>' : pair 1 wins the tie at 0.9
    ora.block(p, np.zeros((1, F, 2), np.float32))
    pd = ora.states[0].pd3d
    assert pd.prev_count == 1
    assert tuple(pd.prev_mix_volumes[0]) == pytest.approx((0.05, 0.9))
    # no volume is applied by process_frames: gain 1 filter settles to identity on a constant
    p2 = params(ob, 1, linear_attenuation=1.0)
    ora2 = ob.BatchOracle(ob.KIND_3D_PROCESS, 1, F)
    for _ in range(30):
        out, _, _ = ora2.block(p2, np.full((1, F, 2), 0.25, np.float32))
    np.testing.assert_allclose(out[0], 0.25, rtol=1e-4)


def test_effect_chain_ping_pong_truth_table(ob):
    """audio_spatializer_effect.cpp:52-76: the last effect always writes p_output_buf."""
    L = ob.lib()
    F = 64
    src = np.zeros((F, 2), np.float32)
    out = np.zeros((F, 2), np.float32)
    tmp = np.zeros((F, 2), np.float32)
    p = params(ob, 1, fx_shelf_gain=1.0, fx_shelf_cutoff_hz=5000.0)
    pp = p.ctypes.data_as(C.POINTER(ob.Params))
    for E in range(0, 5):
        pd = ob.PDataEffect()
        pd.n_effects = E
        for j in range(E):
            pd.kinds[j] = ob.FX_HIGHSHELF
        trace = L.gaso_process_frames_effect(pp, C.byref(pd), None, out.ctypes.data, src.ctypes.data, F, tmp.ctypes.data, 48000.0)
        for j in range(E):
            is_even = (j + E) % 2 == 0
            dst_is_temp = bool(trace >> (2 * j) & 1)
            src_is_temp = bool(trace >> (2 * j + 1) & 1)
            assert dst_is_temp == is_even
            assert src_is_temp == (j > 0 and not is_even)
        if E:
            assert not (trace >> (2 * (E - 1)) & 1)  # last destination is the output buffer


def test_highshelf_effect_snaps_coefficients(ob):
    """[ENGINE] AudioEffectHighShelfFilter: no interpolation, so it IS lfilter from the first sample."""
    F, rng = 256, np.random.default_rng(3)
    ora = ob.BatchOracle(ob.KIND_EFFECT, 1, F, chain=[ob.FX_HIGHSHELF])
    p = params(ob, 1, fx_shelf_gain=0.35, fx_shelf_cutoff_hz=3000.0)
    x = rng.uniform(-0.5, 0.5, (6, 1, F, 2)).astype(np.float32)
    y = np.concatenate([ora.block(p, x[i])[0][0] for i in range(6)])
    b, a = coeffs(ob, 48000.0, 3000.0, 0.35)
    for ear in range(2):
        ref = lfilter(b, a, x[:, 0, :, ear].reshape(-1).astype(np.float64))
        assert rel_rms(y[:, ear], ref) < 2e-6


def test_hrtf_direct_vs_float64_convolution_across_blocks(ob):
    F, n, rng = 512, 3, np.random.default_rng(5)
    hrir = (rng.standard_normal((4, 2, 256)) * np.exp(-np.arange(256) / 32)).astype(np.float32)
    for impl in (0, 1):
        ora = ob.BatchOracle(ob.KIND_EFFECT, n, F, chain=[ob.FX_HRTF], hrir=hrir, hrtf_impl=impl)
        p = params(ob, n, hrtf_gain=1.0)
        p["hrtf_dir"] = [0, 1, 3]
        x = rng.uniform(-0.5, 0.5, (5, n, F, 2)).astype(np.float32)
        ys = []
        for i in range(5):
            ys.append(ora.block(p, x[i])[0][0] if n == 1 else None)
        # per-source check through the peaks-free path: run sources one at a time
        for s in range(n):
            o1 = ob.BatchOracle(ob.KIND_EFFECT, 1, F, chain=[ob.FX_HRTF], hrir=hrir, hrtf_impl=impl)
            out = np.concatenate([o1.block(p[s:s + 1], x[i, s:s + 1])[0][0] for i in range(5)])
            mono = ((x[:, s, :, 0] + x[:, s, :, 1]) * np.float32(0.5)).reshape(-1)
            t = np.tile(np.arange(F, dtype=np.float32) / np.float32(F), 5)
            g = np.ones_like(t)
            g[:F] = t[:F]  # first block ramps 0 -> 1, afterwards 1 -> 1
            xs = (mono * g).astype(np.float64)
            for ear in range(2):
                ref = np.convolve(xs, hrir[p["hrtf_dir"][s], ear].astype(np.float64))[: 5 * F]
                assert rel_rms(out[:, ear], ref) < 5e-7 if impl == 0 else rel_rms(out[:, ear], ref) < 5e-6


def test_early_reflections_closed_form(ob):
    F, R = 256, 4096
    rng = np.random.default_rng(8)
    ora = ob.BatchOracle(ob.KIND_EFFECT, 1, F, chain=[ob.FX_EARLY_REFLECTIONS], er_ring_frames=R)
    p = params(ob, 1)
    p["er_gain"] = 0.7 ** np.arange(1, 9)
    p["er_delay"] = [[48, 100, 255, 256, 700, 1500, 3000, 3840]]
    x = rng.uniform(-0.5, 0.5, (20, 1, F, 2)).astype(np.float32)
    y = np.concatenate([ora.block(p, x[i])[0][0] for i in range(20)])
    xs = x[:, 0].reshape(-1, 2).astype(np.float64)
    ref = xs.copy()
    for g, d in zip(p["er_gain"][0], (int(v) for v in p["er_delay"][0])):
        ref[d:] += np.float64(g) * xs[:-d]
    assert rel_rms(y, ref) < 3e-7


def test_hrtf_crossfade_closed_form(ob):
    """8f#4: when the direction changes, out = t * (x * h_new) + (1 - t) * (x * h_old), t = i / F, over hist ++ x."""
    F, rng = 256, np.random.default_rng(11)
    hrir = (rng.standard_normal((3, 2, 256)) * np.exp(-np.arange(256) / 32)).astype(np.float32)
    ora = ob.BatchOracle(ob.KIND_EFFECT, 1, F, chain=[ob.FX_HRTF], hrir=hrir, crossfade=True)
    p = params(ob, 1, hrtf_gain=1.0)
    dirs = [0, 0, 2, 1, 1]
    x = rng.uniform(-0.5, 0.5, (5, 1, F, 2)).astype(np.float32)
    outs = []
    for b, d in enumerate(dirs):
        p["hrtf_dir"] = d
        outs.append(ora.block(p, x[b])[0][0])
    mono = ((x[:, 0, :, 0] + x[:, 0, :, 1]) * np.float32(0.5)).reshape(-1)
    g = np.ones(5 * F, np.float32)
    g[:F] = np.arange(F, dtype=np.float32) / np.float32(F)
    xs = (mono * g).astype(np.float64)
    full = {d: [np.convolve(xs, hrir[d, ear].astype(np.float64))[: 5 * F] for ear in range(2)] for d in set(dirs)}
    t = (np.arange(F, dtype=np.float32) / np.float32(F)).astype(np.float64)
    for b, d in enumerate(dirs):
        prev = dirs[b - 1] if b else d
        sl = slice(b * F, (b + 1) * F)
        for ear in range(2):
            want = full[d][ear][sl] if prev == d else t * full[d][ear][sl] + (1 - t) * full[prev][ear][sl]
            assert rel_rms(outs[b][:, ear], want) < 5e-7, (b, ear)


# ---- the other AudioFilterSW modes and AudioEffectAmplify (further AudioEffect kinds) ------------------------
# [ENGINE] recollection, unpinned like the rest of Appendix B; what is checked here shares no code with the oracle: the
# RBJ cookbook forms written out in numpy, the transfer-function identities of each mode, scipy's lfilter.


def _filter_coeffs(ob, kind, sr, fc, q, gain):
    out = ob.Coeffs()
    L = ob.lib()
    L.gaso_filter_coeffs.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(ob.Coeffs)]
    L.gaso_filter_coeffs.restype = None
    L.gaso_filter_coeffs(kind, sr, fc, q, gain, C.byref(out))
    return np.array([out.b0, out.b1, out.b2], np.float64), np.array([1.0, -out.a1, -out.a2], np.float64)  # feedback terms are stored negated


def _rbj(kind, ob, sr, fc, q, A):
    w = 2 * np.pi * fc / sr
    cs, sn = np.cos(w), np.sin(w)
    if kind == ob.FX_BANDPASS:
        q = 2 * q
    al = sn / (2 * q)
    if kind == ob.FX_LOWPASS:
        b, a = [(1 - cs) / 2, 1 - cs, (1 - cs) / 2], [1 + al, -2 * cs, 1 - al]
    elif kind == ob.FX_HIGHPASS:
        b, a = [(1 + cs) / 2, -(1 + cs), (1 + cs) / 2], [1 + al, -2 * cs, 1 - al]
    elif kind == ob.FX_BANDPASS:  # the engine's variant: peak gain sqrt(Q + 1)
        b, a = [al * np.sqrt(q + 1), 0.0, -al * np.sqrt(q + 1)], [1 + al, -2 * cs, 1 - al]
    elif kind == ob.FX_NOTCH:
        b, a = [1.0, -2 * cs, 1.0], [1 + al, -2 * cs, 1 - al]
    else:  # low shelf, A passed as the linear gain itself, beta = sqrt(A) / sqrt(Q)
        be = np.sqrt(A) / np.sqrt(q)
        b = [A * ((A + 1) - (A - 1) * cs + be * sn), 2 * A * ((A - 1) - (A + 1) * cs), A * ((A + 1) - (A - 1) * cs - be * sn)]
        a = [(A + 1) + (A - 1) * cs + be * sn, -2 * ((A - 1) + (A + 1) * cs), (A + 1) + (A - 1) * cs - be * sn]
    return np.array(b) / a[0], np.array(a) / a[0]


def _gain_at(b, a, w):
    z = np.exp(-1j * w * np.arange(3))
    return abs(np.dot(b, z) / np.dot(a, z))


@pytest.mark.parametrize("kind_name", ["FX_LOWPASS", "FX_HIGHPASS", "FX_BANDPASS", "FX_NOTCH", "FX_LOWSHELF"])
def test_filter_modes_match_rbj_and_their_identities(ob, kind_name):
    kind = getattr(ob, kind_name)
    sr, fc, q, A = 48000.0, 2000.0, 0.5, 0.4
    b, a = _filter_coeffs(ob, kind, sr, fc, q, A)
    rb, ra = _rbj(kind, ob, sr, fc, q, A)
    np.testing.assert_allclose(b, rb, rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(a, ra, rtol=2e-6, atol=1e-9)
    assert np.all(np.abs(np.roots(a)) < 1.0)
    w0 = 2 * np.pi * fc / sr
    dc, nyq, at_fc = _gain_at(b, a, 0.0), _gain_at(b, a, np.pi), _gain_at(b, a, w0)
    if kind == ob.FX_LOWPASS:
        assert dc == pytest.approx(1.0, abs=1e-5) and nyq < 1e-6 and at_fc == pytest.approx(q, rel=1e-4)  # |H(w0)| = Q
    elif kind == ob.FX_HIGHPASS:
        assert dc < 1e-6 and nyq == pytest.approx(1.0, abs=1e-5) and at_fc == pytest.approx(q, rel=1e-4)
    elif kind == ob.FX_BANDPASS:
        assert dc < 1e-6 and nyq < 1e-6 and at_fc == pytest.approx(np.sqrt(2 * q + 1), rel=1e-4)
    elif kind == ob.FX_NOTCH:
        assert dc == pytest.approx(1.0, abs=1e-5) and nyq == pytest.approx(1.0, abs=1e-5) and at_fc < 1e-5
    else:
        assert dc == pytest.approx(A * A, rel=1e-3) and nyq == pytest.approx(1.0, abs=1e-4)  # the shelf lifts DC by A^2


@pytest.mark.parametrize("kind_name", ["FX_LOWPASS", "FX_HIGHPASS", "FX_BANDPASS", "FX_NOTCH", "FX_LOWSHELF"])
def test_filter_effects_are_lfilter_from_the_first_sample(ob, kind_name):
    """[ENGINE] AudioEffectFilterInstance at FILTER_6DB: coefficients snapped every block, one processor per ear."""
    kind = getattr(ob, kind_name)
    F, rng = 256, np.random.default_rng(4)
    ora = ob.BatchOracle(ob.KIND_EFFECT, 1, F, chain=[kind])
    ora.set_fx_settings(0, 0, cutoff_hz=1500.0, resonance=0.8, gain=0.5)
    p = params(ob, 1)
    x = rng.uniform(-0.5, 0.5, (5, 1, F, 2)).astype(np.float32)
    y = np.concatenate([ora.block(p, x[i])[0][0] for i in range(5)])
    b, a = _filter_coeffs(ob, kind, 48000.0, 1500.0, 0.8, 0.5)
    for ear in range(2):
        ref = lfilter(b, a, x[:, 0, :, ear].reshape(-1).astype(np.float64))
        assert rel_rms(y[:, ear], ref) < 5e-6


def test_amplify_ramps_from_the_previous_volume(ob):
    """[ENGINE] AudioEffectAmplify: vol ramps linearly from db_to_linear(previous volume_db) towards the new one and,
    like the mix_channel lerp, stops one increment short of it; the first block has no ramp."""
    F = 128
    ora = ob.BatchOracle(ob.KIND_EFFECT, 1, F, chain=[ob.FX_AMPLIFY])
    p = params(ob, 1)
    x = np.ones((1, F, 2), np.float32)
    ora.set_fx_settings(0, 0, volume_db=-6.0)
    y0 = ora.block(p, x)[0][0]
    v0 = 10.0 ** (-6.0 / 20.0)
    np.testing.assert_allclose(y0, v0, rtol=1e-6)
    ora.set_fx_settings(0, 0, volume_db=3.0)
    y1 = ora.block(p, x)[0][0]
    v1 = 10.0 ** (3.0 / 20.0)
    ref = v0 + (v1 - v0) * np.arange(F) / F
    np.testing.assert_allclose(y1[:, 0], ref, rtol=2e-5)
    np.testing.assert_array_equal(y1[:, 0], y1[:, 1])
    y2 = ora.block(p, x)[0][0]
    np.testing.assert_allclose(y2, v1, rtol=1e-6)
