"""k_hrtf_uni's twelve-wave form (three waves per SIMD, HRIR rows staged through LDS by global_load_lds) against the
oracle: the callbacks that take it are large (every one of 256 x 12 waves needs a source), so these cases are too.
The form is switched on for the test through gas_tune_uni12_min and restored afterwards; the eight-wave form runs the
same inputs as a cross-check (same oracle tolerance, not bitwise: the sources are split over the waves differently)."""
import numpy as np
import pytest

from helpers import TOL, rel_rms

pytestmark = pytest.mark.gpu


@pytest.fixture()
def twelve(gas):
    lib = gas.load_library()
    was = lib.gas_tune_uni12_min(1)
    yield lib
    lib.gas_tune_uni12_min(was)


def _run(gas, n, F, blocks, flags, draining, seed, dirs=64, device_rows=False, holes=False):
    from godot_audio_spatializer_amd import synth

    rng = np.random.default_rng(seed)
    hrir = synth.synthetic_hrir(np.random.default_rng(3), dirs=dirs)
    K = gas.capi
    params = [synth.draw_params(rng, n, dirs=dirs, frames=F) for _ in range(blocks)]
    srcs = [synth.draw_sources(rng, n, F) for _ in range(blocks)]
    mixes, peaks = [], []
    with gas.SpatializerContext(max_sources=n + 8, frames=F, flags=flags) as ctx:
        ctx.hrtf_load(hrir)
        slots = ctx.source_alloc_many(n + (8 if holes else 0), K.KIND_EFFECT, (K.FX_HRTF,))
        if holes:  # a slot list with gaps: the kernel reads the list instead of a contiguous range
            slots = np.delete(slots, [1, 5, 77, 300, 301, 1000, 2000, 2500])
        for s in draining:
            ctx.source_set_draining(int(slots[s]), True)
        if device_rows:
            import torch

            d_src = [torch.from_numpy(s).cuda() for s in srcs]
            d_par = [torch.from_numpy(p.view(np.uint8).reshape(n, 128).copy()).cuda() for p in params]
            d_out = torch.zeros(blocks, 1, F, 2, device="cuda")
            d_pk = torch.zeros(blocks, n, 2, device="cuda")
            for b in range(blocks):
                ctx.params_publish_device(d_par[b].data_ptr(), n, slots if b == 0 else None)
                rc = ctx.process_block_raw(d_src[b].data_ptr(), slots if b == 0 else None, n, F, d_out[b].data_ptr(), d_pk[b].data_ptr(), K.MEM_DEVICE)
                assert rc == 0
            ctx.synchronize()
            mixes, peaks = list(d_out.cpu().numpy()), list(d_pk.cpu().numpy())
        else:
            for b in range(blocks):
                ctx.params_publish_batch(slots, params[b])
                m, p = ctx.process_block(srcs[b], slots)
                mixes.append(m)
                peaks.append(p)
    return mixes, peaks, params, srcs, hrir


@pytest.mark.parametrize("F", [512, 256, 128, 384])
@pytest.mark.parametrize("mode", ["all_peaks", "draining_only"])
def test_twelve_wave_form_matches_oracle(gas, ob, twelve, F, mode):
    K = gas.capi
    n, blocks = 3072 + 517, 3  # uneven split: some waves get one source, some two
    flags = K.FLAG_PEAKS_DRAINING_ONLY if mode == "draining_only" else 0
    draining = list(range(0, n, 37)) if mode == "draining_only" else []
    mixes, peaks, params, srcs, hrir = _run(gas, n, F, blocks, flags, draining, seed=5 + F)
    ora = ob.BatchOracle(ob.KIND_EFFECT, n, F, chain=[K.FX_HRTF], hrir=hrir)
    for b in range(blocks):
        _, rp, r64 = ora.block(params[b].astype(ob.PARAMS_DTYPE), srcs[b], want64=True)
        assert rel_rms(mixes[b][0], r64[0]) <= TOL, f"block {b}"
        if mode == "all_peaks":
            np.testing.assert_allclose(peaks[b], rp, rtol=2e-5, atol=1e-7)
        else:
            np.testing.assert_allclose(peaks[b][draining], rp[draining], rtol=2e-5, atol=1e-7)
            rest = np.setdiff1d(np.arange(n), draining)
            assert np.all(np.isposinf(peaks[b][rest]))


def test_twelve_wave_form_device_rows_and_slot_list(gas, ob, twelve):
    """Device-published parameter rows read by the kernel itself (and written through), a slot list with holes, the
    carried partial-mix sum of GAS_FLAG_PIPELINED_MIX, five callbacks so that histories and gains carry over."""
    K = gas.capi
    n, F, blocks = 4100, 512, 5
    flags = K.FLAG_PEAKS_DRAINING_ONLY | K.FLAG_PIPELINED_MIX
    draining = [0, 63, 64, 4099]
    mixes, peaks, params, srcs, hrir = _run(gas, n, F, blocks, flags, draining, seed=77, device_rows=True, holes=True)
    ora = ob.BatchOracle(ob.KIND_EFFECT, n, F, chain=[K.FX_HRTF], hrir=hrir)
    for b in range(blocks):
        _, rp, r64 = ora.block(params[b].astype(ob.PARAMS_DTYPE), srcs[b], want64=True)
        assert rel_rms(mixes[b][0], r64[0]) <= TOL, f"block {b}"
        np.testing.assert_allclose(peaks[b][draining], rp[draining], rtol=2e-5, atol=1e-7)


def test_twelve_and_eight_wave_forms_agree(gas, twelve):
    K = gas.capi
    n, F = 8192, 512
    a, pa, *_ = _run(gas, n, F, 2, K.FLAG_PEAKS_DRAINING_ONLY, [5, 4000], seed=9)
    twelve.gas_tune_uni12_min(0)
    b, pb, *_ = _run(gas, n, F, 2, K.FLAG_PEAKS_DRAINING_ONLY, [5, 4000], seed=9)
    for x, y in zip(a, b):
        assert rel_rms(x, y) <= TOL
    for x, y in zip(pa, pb):
        np.testing.assert_allclose(x[[5, 4000]], y[[5, 4000]], rtol=2e-5, atol=1e-7)
