"""BASELINE.json's full sizes, checked through size-independent properties (the oracle is too slow there):
additivity over a split of the sources, invariance under row permutation, gain linearity, bitwise run-to-run
reproducibility (no float atomics), and a spot check of a random subset against the oracle."""
import numpy as np
import pytest

from helpers import TOL, rel_rms

pytestmark = pytest.mark.gpu

CASES = {
    "cfg2_biquad_256": dict(kind=0, chain=(), n=256, frames=512, ring=0),
    "cfg3_hrtf_4096": dict(kind=2, chain=(3,), n=4096, frames=512, ring=0),
    "cfg4_shard_hrtf_8192": dict(kind=2, chain=(3,), n=8192, frames=512, ring=0),
    "cfg5_erhrtf_4096": dict(kind=2, chain=(2, 3), n=4096, frames=256, ring=4096),
    "biquad_65536": dict(kind=0, chain=(), n=65536, frames=512, ring=0),
}


def run(gas, case, order=None, subset=None, gain_scale=1.0, blocks=3, flags=0, seed=11):
    """Runs `blocks` callbacks over the (sub)set of sources in `order`; returns the last mix and peaks."""
    from godot_audio_spatializer_amd import synth

    c = CASES[case]
    n, F = c["n"], c["frames"]
    rng = np.random.default_rng(seed)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=256)
    params = [synth.draw_params(rng, n, dirs=256, ring_frames=max(c["ring"], 2 * F), frames=F) for _ in range(2)]
    srcs = [synth.draw_sources(rng, n, F) for _ in range(blocks)]
    idx = np.arange(n) if subset is None else np.asarray(subset)
    if order is not None:
        idx = idx[order]
    with gas.SpatializerContext(max_sources=n, frames=F, er_ring_frames=c["ring"], flags=flags) as ctx:
        if 3 in c["chain"]:
            ctx.hrtf_load(hrir)
        slots = ctx.source_alloc_many(len(idx), c["kind"], c["chain"])
        for b in range(blocks):
            p = params[b % 2][idx].copy()
            p["hrtf_gain"] *= gain_scale
            p["mix_volumes"] *= gain_scale
            ctx.params_publish_batch(slots, p)
            mix, peaks = ctx.process_block(srcs[b][idx], slots)
    return mix, peaks, (params, srcs, hrir)


@pytest.mark.parametrize("case", list(CASES))
def test_additive_over_a_split_and_permutation_invariant(gas, case):
    n = CASES[case]["n"]
    full, pk_full, _ = run(gas, case)
    rng = np.random.default_rng(0)
    perm = rng.permutation(n)
    a, _, _ = run(gas, case, subset=perm[: n // 3])
    b, _, _ = run(gas, case, subset=perm[n // 3:])
    assert rel_rms(a.astype(np.float64) + b, full) <= TOL  # sources are independent; the mix is their sum
    shuffled, pk_sh, _ = run(gas, case, order=perm)
    assert rel_rms(shuffled, full) <= TOL
    np.testing.assert_array_equal(pk_sh, pk_full[perm])  # per-source results do not depend on the row


@pytest.mark.parametrize("case", ["cfg3_hrtf_4096", "cfg5_erhrtf_4096", "cfg2_biquad_256"])
def test_gain_linearity(gas, case):
    base, pk, _ = run(gas, case)
    half, pk2, _ = run(gas, case, gain_scale=0.5)
    np.testing.assert_allclose(half, base * 0.5, rtol=0, atol=1e-6 * np.abs(base).max())  # exact power-of-two scaling up to the lerp's rounding
    np.testing.assert_allclose(pk2, pk * 0.5, rtol=1e-5)


@pytest.mark.parametrize("case", ["cfg4_shard_hrtf_8192", "biquad_65536"])
def test_bitwise_reproducible(gas, case):
    a, pa, _ = run(gas, case)
    b, pb, _ = run(gas, case)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(pa, pb)


def test_frequency_domain_mode_equals_exact_mode_at_full_size(gas):
    exact, _, _ = run(gas, "cfg4_shard_hrtf_8192")
    fd, pk, _ = run(gas, "cfg4_shard_hrtf_8192", flags=gas.capi.FLAG_PEAKS_DRAINING_ONLY)
    assert rel_rms(fd, exact) <= TOL
    assert np.all(np.isposinf(pk))


@pytest.mark.parametrize("case", ["cfg4_shard_hrtf_8192", "cfg5_erhrtf_4096"])
def test_direction_order_leaves_the_mix_unchanged_at_full_size(gas, case):
    """GAS_FLAG_DIRECTION_ORDER only regroups the frequency-domain sum (f32 summation order); it is also
    bitwise reproducible run to run (the device sort is stable and atomic-free)."""
    K = gas.capi
    plain, _, _ = run(gas, case, flags=K.FLAG_PEAKS_DRAINING_ONLY)
    ordered, pk, _ = run(gas, case, flags=K.FLAG_PEAKS_DRAINING_ONLY | K.FLAG_DIRECTION_ORDER)
    again, _, _ = run(gas, case, flags=K.FLAG_PEAKS_DRAINING_ONLY | K.FLAG_DIRECTION_ORDER)
    assert rel_rms(ordered, plain) <= TOL
    assert np.array_equal(ordered, again)
    assert np.all(np.isposinf(pk))


@pytest.mark.parametrize("case", ["cfg3_hrtf_4096", "cfg5_erhrtf_4096"])
def test_random_subset_against_oracle(gas, ob, case):
    """128 sources drawn from the full-size problem: GPU on the subset == oracle on the subset (and by the
    additivity test above the full mix is the sum of such subsets)."""
    c = CASES[case]
    n, F = c["n"], c["frames"]
    sub = np.sort(np.random.default_rng(1).choice(n, 128, replace=False))
    blocks = 3 if c["ring"] == 0 else 18
    mix, peaks, (params, srcs, hrir) = run(gas, case, subset=sub, blocks=blocks, seed=11)
    ora = ob.BatchOracle(ob.KIND_EFFECT, len(sub), F, chain=list(c["chain"]), hrir=hrir, er_ring_frames=max(c["ring"], 1))
    for b in range(blocks):
        _, rp, r64 = ora.block(params[b % 2][sub].astype(ob.PARAMS_DTYPE), srcs[b][sub], want64=True)
    assert rel_rms(mix[0], r64[0]) <= TOL
    np.testing.assert_allclose(peaks, rp, rtol=2e-5, atol=1e-7)
