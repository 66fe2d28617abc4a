"""AudioSpatializerHRTF from a MEASURED set -- M irregular (azimuth, elevation) positions x 2 ears x taps,
regridded on the device onto the library's azimuth x elevation grid (gas_hrtf_load_positions).  NEW, no reference
counterpart: parity unpinned.  The checker is a float64 numpy restatement of the rule written down in include/gas_amd.h
(nearest on the sphere / three nearest weighted by 1 / angle); cells whose choice hangs on a near-tie in float32 are
excluded from the element-wise comparison."""
import numpy as np
import pytest

from helpers import TOL, rel_rms

pytestmark = pytest.mark.gpu


def unit(az, el):
    return np.stack([np.cos(el) * np.sin(az), np.sin(el), -np.cos(el) * np.cos(az)], -1)


def regrid_ref(pos, hrir, n_az, n_el, interpolation):
    m, _, taps = hrir.shape
    p = unit(pos[:, 0].astype(np.float64), pos[:, 1].astype(np.float64))
    out = np.zeros((n_az * n_el, 2, 256))
    safe = np.ones(n_az * n_el, bool)
    for ei in range(n_el):
        for ai in range(n_az):
            az = ai * 2 * np.pi / n_az
            el = -np.pi / 2 + ei * np.pi / (n_el - 1) if n_el > 1 else 0.0
            d = np.sqrt(((p - unit(np.float64(az), np.float64(el))) ** 2).sum(-1))  # chord on the unit sphere
            order = np.lexsort((np.arange(m), d))
            k = 1 if interpolation == 0 else min(3, m)
            if m > k and d[order[k]] - d[order[k - 1]] < 1e-5:
                safe[ei * n_az + ai] = False  # the k-th and (k+1)-th nearest are a near-tie
            if interpolation == 0 or m == 1:
                out[ei * n_az + ai, :, :taps] = hrir[order[0]]
            else:
                w = 1.0 / (2 * np.arcsin(np.minimum(1.0, d[order[:k]] / 2)) + 1e-4)
                w /= w.sum()
                out[ei * n_az + ai, :, :taps] = np.tensordot(w, hrir[order[:k]].astype(np.float64), 1)
    return out, safe


@pytest.mark.parametrize("interpolation,m,taps", [(0, 300, 256), (1, 300, 200), (1, 2, 64), (0, 1, 256)])
def test_regridded_set_matches_the_rule(gas, interpolation, m, taps):
    rng = np.random.default_rng(m + taps)
    n_az, n_el = 32, 9
    pos = np.stack([rng.uniform(-np.pi, np.pi, m), np.arcsin(rng.uniform(-1, 1, m))], -1).astype(np.float32)
    hrir = (rng.standard_normal((m, 2, taps)) * np.exp(-np.arange(taps) / 32.0)).astype(np.float32)
    with gas.SpatializerContext(max_sources=4, frames=512) as ctx:
        got = ctx.hrtf_load_positions(pos, hrir, n_az, n_el, interpolation)
    want, safe = regrid_ref(pos, hrir, n_az, n_el, interpolation)
    assert safe.sum() > 0.9 * len(safe)
    np.testing.assert_allclose(got[safe], want[safe], rtol=2e-3, atol=2e-5)  # weights are f32 functions of an f32 acos
    if interpolation == 0:
        np.testing.assert_array_equal(got[safe], want[safe].astype(np.float32))  # nearest: a copy
    assert not got[:, :, taps:].any()  # shorter sets are zero-padded


def test_a_set_measured_on_the_grid_is_the_pregridded_load(gas, ob):
    """Positions that ARE the grid cells (in scrambled order): the loader must reproduce gas_hrtf_load of the gridded
    array, and the HRTF path on top of it must match the oracle run on that array."""
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    rng = np.random.default_rng(9)
    n_az, n_el, F, n = 16, 5, 512, 40
    dirs = n_az * n_el
    hrir = synth.synthetic_hrir(rng, dirs=dirs)
    cells = np.arange(dirs)
    az = (cells % n_az) * 2 * np.pi / n_az
    el = -np.pi / 2 + (cells // n_az) * np.pi / (n_el - 1)
    # the poles hold n_az coincident cells each: keep one measurement per pole so that "nearest" is unambiguous
    keep = np.array([c for c in cells if (c // n_az not in (0, n_el - 1)) or c % n_az == 0])
    perm = rng.permutation(len(keep))
    pos = np.stack([az[keep][perm], el[keep][perm]], -1).astype(np.float32)
    with gas.SpatializerContext(max_sources=n, frames=F) as ctx:
        grid = ctx.hrtf_load_positions(pos, hrir[keep][perm], n_az, n_el, 0)
        mid = np.array([c for c in cells if c // n_az not in (0, n_el - 1)])
        np.testing.assert_array_equal(grid[mid], hrir[mid])
        for pole in (0, n_el - 1):
            np.testing.assert_array_equal(grid[pole * n_az: (pole + 1) * n_az], np.broadcast_to(hrir[pole * n_az], (n_az, 2, 256)))
        slots = ctx.source_alloc_many(n, K.KIND_EFFECT, (K.FX_HRTF,))
        ora = ob.BatchOracle(ob.KIND_EFFECT, n, F, chain=(ob.FX_HRTF,), hrir=grid)
        for b in range(4):
            p = synth.draw_params(rng, n, dirs=dirs)
            ctx.params_publish_batch(slots, p)
            src = synth.draw_sources(rng, n, F)
            mix, peaks = ctx.process_block(src, slots)
            _, rpeaks, r64 = ora.block(p.astype(ob.PARAMS_DTYPE), src, want64=True)
            assert rel_rms(mix[0], r64[0]) <= TOL
            np.testing.assert_allclose(peaks, rpeaks, rtol=2e-5, atol=1e-7)


def test_bad_arguments(gas):
    with gas.SpatializerContext(max_sources=2, frames=512) as ctx:
        pos = np.zeros((3, 2), np.float32)
        with pytest.raises(gas.GasError):
            ctx.hrtf_load_positions(pos, np.zeros((3, 2, 300), np.float32), 8, 3)  # more than 256 taps
        with pytest.raises(gas.GasError):
            ctx.hrtf_load_positions(pos, np.zeros((3, 2, 16), np.float32), 8, 3, interpolation=2)
