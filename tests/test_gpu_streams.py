"""device-resident source sampling (gas_stream_*, gas_process_block_streams) vs the oracle's
restatement of the source window, fade-out and gate (audio_spatializer.cpp:367-408,464-469)."""
import os

import numpy as np
import pytest

from helpers import mix_matches
from test_oracle_mixer import Rig

pytestmark = pytest.mark.gpu

SPEECH = os.path.join(os.path.dirname(__file__), "golden", "speech_excerpt_s16.npy")


def to_float_stereo(pcm):
    f = pcm.astype(np.float32) / np.float32(32768.0)
    return np.stack([f, f], axis=1) if pcm.ndim == 1 else f


@pytest.mark.parametrize("kind_name", ["effect_copy", "mix_channel", "hrtf", "hrtf_crossfade", "hrtf_throughput_mode"])
def test_streams_match_oracle_mixer(gas, ob, kind_name):
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    rng = np.random.default_rng(9)
    F = 512
    speech = np.load(SPEECH)
    # int16 mono speech (realistic), int16 stereo noise, float32 mono; lengths ending mid-callback / near the block end
    pcms = [speech[:5000], speech[2000:2000 + 1024 + 500], (rng.uniform(-0.5, 0.5, (3000, 2)) * 32767).astype(np.int16), rng.uniform(-0.5, 0.5, 700).astype(np.float32), speech]
    floats = [to_float_stereo(p) if p.dtype == np.int16 else np.stack([p, p], axis=1) for p in pcms]
    n = len(pcms)
    kind, okind, chain, ochain, hrir = {
        "effect_copy": (K.KIND_EFFECT, ob.KIND_EFFECT, (), (), None),
        "mix_channel": (K.KIND_3D_MIX, ob.KIND_3D_MIX, (), (), None),
        "hrtf": (K.KIND_EFFECT, ob.KIND_EFFECT, (K.FX_HRTF,), (ob.FX_HRTF,), synth.synthetic_hrir(np.random.default_rng(7), dirs=8)),
        "hrtf_crossfade": (K.KIND_EFFECT, ob.KIND_EFFECT, (K.FX_HRTF,), (ob.FX_HRTF,), synth.synthetic_hrir(np.random.default_rng(7), dirs=8)),
        "hrtf_throughput_mode": (K.KIND_EFFECT, ob.KIND_EFFECT, (K.FX_HRTF,), (ob.FX_HRTF,), synth.synthetic_hrir(np.random.default_rng(7), dirs=8)),
    }[kind_name]
    xf = kind_name == "hrtf_crossfade"
    params = synth.draw_params(rng, n, dirs=8)
    with gas.SpatializerContext(max_sources=n, frames=F, flags=K.FLAG_PEAKS_DRAINING_ONLY | (K.FLAG_HRTF_CROSSFADE if xf else 0) | (K.FLAG_PIPELINED_MIX if kind_name.endswith("throughput_mode") else 0)) as ctx:
        if hrir is not None:
            ctx.hrtf_load(hrir)
        slots = ctx.source_alloc_many(n, kind, chain)
        ctx.params_publish_batch(slots, params)
        for s, p in zip(slots, pcms):
            ctx.source_bind_stream(s, ctx.stream_create(p))
        rig = Rig(ob, okind, floats, F, chain=ochain, hrir=hrir)
        if xf:
            rig.hrtf.crossfade = 1
        rig.params[:] = params.astype(ob.PARAMS_DTYPE)
        active = np.ones(n, bool)
        thr = 1e-4  # db_to_linear(-80 dB), audio_spatializer.cpp:465
        for cb in range(30):
            if xf and cb % 2 == 1:  # move every source: the next callback cross-fades
                params["hrtf_dir"] = (params["hrtf_dir"] + 1 + cb) % 8
                ctx.params_publish_batch(slots, params)
                rig.params[:] = params.astype(ob.PARAMS_DTYPE)
            live = slots[active]
            got, peaks, hf = ctx.process_block_streams(live)
            rc, want = rig.get_mixed_frames(0)
            assert rc == 0
            assert mix_matches(got[0], want), f"callback {cb}"
            idx = np.flatnonzero(active)
            for j, i in enumerate(idx):
                assert bool(rig.pbs[i].has_frames) == bool(hf[j]), (cb, i)
                if not hf[j]:  # host gate on the returned peak (:464-469)
                    np.testing.assert_allclose(peaks[j], tuple(rig.pbs[i].last_peak), rtol=2e-5, atol=1e-7)
                    if peaks[j].max() <= thr:
                        active[i] = False
                else:
                    assert np.all(np.isposinf(peaks[j])) or not kind_name.startswith("hrtf")
            for i in range(n):
                assert active[i] == bool(rig.pbs[i].active), (cb, i)
            if not active.any():
                break
        assert not active.any()  # every playback drained and was gated off


def test_stream_start_offset_and_rebind(gas):
    K = gas.capi
    F = 512
    ramp = np.arange(1, 4001, dtype=np.float32) / 8192.0
    with gas.SpatializerContext(max_sources=2, frames=F) as ctx:
        sid = ctx.stream_create(ramp)
        slot = ctx.source_alloc(K.KIND_EFFECT)  # empty chain = copy
        ctx.params_publish(slot, np.zeros(1, K.PARAMS_DTYPE))
        ctx.source_bind_stream(slot, sid, start_frame=1000)
        mix, _, hf = ctx.process_block_streams([slot])
        assert hf[0] and not mix[0, :64].any()  # zeroed lookahead (audio_spatializer.cpp:61-63)
        np.testing.assert_array_equal(mix[0, 64:, 0], ramp[1000:1000 + F - 64])
        np.testing.assert_array_equal(mix[0, 64:, 1], ramp[1000:1000 + F - 64])  # mono feeds both ears
        mix, _, _ = ctx.process_block_streams([slot])
        np.testing.assert_array_equal(mix[0, :, 0], ramp[1000 + F - 64:1000 + 2 * F - 64])
        with pytest.raises(gas.GasError):
            ctx.stream_destroy(sid)  # still bound
        ctx.source_bind_stream(slot, sid, start_frame=0)  # restart
        mix, _, _ = ctx.process_block_streams([slot])
        np.testing.assert_array_equal(mix[0, 64:, 0], ramp[: F - 64])


def test_device_sampler_rejects_pitch_shift(gas):
    K = gas.capi
    with gas.SpatializerContext(max_sources=1, frames=512) as ctx:
        slot = ctx.source_alloc(K.KIND_EFFECT)
        p = np.zeros(1, K.PARAMS_DTYPE)
        p["pitch_scale"] = 1.5
        ctx.params_publish(slot, p)
        ctx.source_bind_stream(slot, ctx.stream_create(np.zeros(2048, np.int16)))
        with pytest.raises(gas.GasError) as ei:
            ctx.process_block_streams([slot])
        assert ei.value.status == -6


def test_freed_playback_releases_its_stream(gas):
    """A playback bound to a device stream that ended and was freed must not keep the stream "bound" for ever
    (gas_stream_destroy), and the re-allocated slot must not inherit its cursor."""
    K = gas.capi
    F = 512
    ramp = np.arange(1, 1201, dtype=np.float32) / 4096.0
    with gas.SpatializerContext(max_sources=1, frames=F) as ctx:
        sid = ctx.stream_create(ramp)
        slot = ctx.source_alloc(K.KIND_EFFECT)
        ctx.params_publish(slot, np.zeros(1, K.PARAMS_DTYPE))
        ctx.source_bind_stream(slot, sid)
        hf = [True]
        for _ in range(6):  # play to the end and past it
            _, _, hf = ctx.process_block_streams([slot])
        assert not hf[0]
        ctx.source_free(slot)  # deferred: takes effect at the next block boundary
        mix, _, _ = ctx.process_block_streams([])
        assert not mix.any()
        ctx.stream_destroy(sid)  # no longer bound
        again = ctx.source_alloc(K.KIND_EFFECT)
        assert again == slot
        ctx.params_publish(again, np.zeros(1, K.PARAMS_DTYPE))
        mix, _, hf = ctx.process_block_streams([again])  # fresh slot, nothing bound: silence, no stale cursor
        assert not hf[0] and not mix.any()


def test_failed_streams_callback_consumes_no_frames(gas, ob):
    """The reference consumes a playback's frames only when it mixes them (audio_spatializer.cpp:378): a streams
    callback that fails (here: HRTF playbacks before gas_hrtf_load) must leave every cursor where it was, so that the
    first successful callback starts at frame 0."""
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    F = 512
    rng = np.random.default_rng(5)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=8)
    pcms = [(rng.uniform(-0.5, 0.5, 2500) * 32767).astype(np.int16), (rng.uniform(-0.5, 0.5, (1800, 2)) * 32767).astype(np.int16)]
    floats = [to_float_stereo(p) for p in pcms]
    params = synth.draw_params(rng, 2, dirs=8)
    with gas.SpatializerContext(max_sources=2, frames=F) as ctx:
        slots = ctx.source_alloc_many(2, K.KIND_EFFECT, (K.FX_HRTF,))
        ctx.params_publish_batch(slots, params)
        for s, p in zip(slots, pcms):
            ctx.source_bind_stream(s, ctx.stream_create(p))
        for _ in range(2):
            with pytest.raises(gas.GasError) as ei:
                ctx.process_block_streams(slots)
            assert ei.value.status == -5  # GAS_ERR_NO_HRTF
        ctx.hrtf_load(hrir)
        rig = Rig(ob, ob.KIND_EFFECT, floats, F, chain=(ob.FX_HRTF,), hrir=hrir)
        rig.params[:] = params.astype(ob.PARAMS_DTYPE)
        for cb in range(4):
            got, _, hf = ctx.process_block_streams(slots)
            rc, want = rig.get_mixed_frames(0)
            assert rc == 0 and mix_matches(got[0], want), f"callback {cb}"
            assert [bool(x) for x in hf] == [bool(p.has_frames) for p in rig.pbs]


@pytest.mark.parametrize("pitch", [0.5, 0.97, 1.0, 1.06, 2.0])
@pytest.mark.parametrize("kind_name", ["effect_copy", "hrtf"])
def test_resampled_streams_match_oracle_mixer(gas, ob, kind_name, pitch):
    """Pitch-scaled device sampling (audio_spatializer.cpp:375-378 hands pitch_scale to the sampler; doppler sets it,
    audio_spatializer_3d.cpp:405-434): playbacks of a gas_stream_set_resampled stream follow the oracle's restatement
    of [ENGINE] AudioStreamPlaybackResampled::mix (16.16 position, 4-point cubic; engine recollection, parity
    unpinned) through the lookahead window, the end-of-stream fade-out and the silence gate.  One playback changes
    its pitch every callback (a moving source)."""
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    rng = np.random.default_rng(13)
    F = 512
    speech = np.load(SPEECH)
    pcms = [speech[:4000], (rng.uniform(-0.5, 0.5, (2600, 2)) * 32767).astype(np.int16), rng.uniform(-0.5, 0.5, 1500).astype(np.float32), speech[1000:1000 + 3333]]
    floats = [to_float_stereo(p) if p.dtype == np.int16 else np.stack([p, p], axis=1) for p in pcms]
    n = len(pcms)
    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=8) if kind_name == "hrtf" else None
    chain, ochain = ((K.FX_HRTF,), (ob.FX_HRTF,)) if kind_name == "hrtf" else ((), ())
    params = synth.draw_params(rng, n, dirs=8)
    params["pitch_scale"] = pitch
    with gas.SpatializerContext(max_sources=n, frames=F, flags=K.FLAG_PEAKS_DRAINING_ONLY) as ctx:
        if hrir is not None:
            ctx.hrtf_load(hrir)
        slots = ctx.source_alloc_many(n, K.KIND_EFFECT, chain)
        for s, p in zip(slots, pcms):
            sid = ctx.stream_create(p)
            ctx.stream_set_resampled(sid, True)
            ctx.source_bind_stream(s, sid)
        rig = Rig(ob, ob.KIND_EFFECT, floats, F, chain=ochain, hrir=hrir)
        for i in range(n):
            rig.pbs[i].resampled = 1
        active = np.ones(n, bool)
        ended = 0
        for cb in range(40):
            params["pitch_scale"][3] = pitch * (1.0 + 0.03 * np.sin(cb))  # a moving source: doppler changes every tick
            ctx.params_publish_batch(slots, params)
            rig.params[:] = params.astype(ob.PARAMS_DTYPE)
            live = slots[active]
            got, peaks, hf = ctx.process_block_streams(live)
            rc, want = rig.get_mixed_frames(0)
            assert rc == 0
            assert mix_matches(got[0], want), f"callback {cb}"
            idx = np.flatnonzero(active)
            for j, i in enumerate(idx):
                assert bool(rig.pbs[i].has_frames) == bool(hf[j]), (cb, i)
                if not hf[j]:
                    np.testing.assert_allclose(peaks[j], tuple(rig.pbs[i].last_peak), rtol=2e-5, atol=1e-7)
                    if peaks[j].max() <= 1e-4:
                        active[i] = False
                        ended += 1
            for i in range(n):
                assert active[i] == bool(rig.pbs[i].active), (cb, i)
            if not active.any():
                break
        assert ended == n  # every playback ran out, faded, rang out and was gated off


def test_resampled_class_is_chosen_before_binding(gas):
    K = gas.capi
    with gas.SpatializerContext(max_sources=1, frames=512) as ctx:
        sid = ctx.stream_create(np.zeros(2048, np.int16))
        slot = ctx.source_alloc(K.KIND_EFFECT)
        ctx.source_bind_stream(slot, sid)
        with pytest.raises(gas.GasError):
            ctx.stream_set_resampled(sid, True)  # already bound
