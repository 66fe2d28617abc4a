"""The C++ host layer (include/gas_amd_host.h) on the GPU vs the oracle's restatement of the whole mixer
(audio_spatializer.cpp:326-527): playbacks of different lengths ending mid-callback, fade-out, silence gate,
channel latch, stop, list GC."""
import numpy as np
import pytest

from helpers import mix_matches
from test_oracle_mixer import Rig

pytestmark = pytest.mark.gpu


def drive(gas, ob, kind, okind, chain, ochain, channel_count, lengths, F, callbacks, hrir=None, stop_at=None, seed=0, device_streams=False):
    from godot_audio_spatializer_amd import synth

    rng = np.random.default_rng(seed)
    streams = [rng.uniform(-0.5, 0.5, (n, 2)).astype(np.float32) for n in lengths]
    if device_streams:  # int16 PCM in HBM; the oracle sees the same samples as floats
        pcm = [(s * 32767).astype(np.int16) for s in streams]
        streams = [p.astype(np.float32) / np.float32(32768.0) for p in pcm]
    params = synth.draw_params(rng, len(lengths), dirs=8, channel_count=channel_count)
    with gas.SpatializerContext(max_sources=len(lengths) + 4, frames=F, channel_count=channel_count) as ctx:
        if hrir is not None:
            ctx.hrtf_load(hrir)
        host = gas.capi.BatchedSpatializerHost(ctx, kind, chain)
        # the host inserts at the head (newest first): start in reverse so list order == oracle array order
        ids = [None] * len(lengths)
        for i in reversed(range(len(lengths))):
            ids[i] = host.start_playback_device_stream(ctx.stream_create(pcm[i])) if device_streams else host.start_playback_array(streams[i])
            host.set_spatializer_parameters(ids[i], params[i])
        rig = Rig(ob, okind, streams, F, channel_count=channel_count, chain=ochain, hrir=hrir)
        rig.params[:] = params.astype(ob.PARAMS_DTYPE)
        C = channel_count if kind == gas.capi.KIND_3D_MIX else 1
        for cb in range(callbacks):
            if stop_at and cb == stop_at[0]:
                host.stop_playback(ids[stop_at[1]])
                rig.pbs[stop_at[1]].active = 0
            for c in range(C):
                rc, got = host.get_mixed_frames(c, F)
                orc, want = rig.get_mixed_frames(c)
                assert rc == 0 and orc == 0
                assert mix_matches(got, want), f"callback {cb} channel {c}"
            for i in range(len(lengths)):
                assert host.is_playback_active(ids[i]) == bool(rig.pbs[i].active), f"callback {cb} playback {i}"
        n_alive = sum(int(rig.pbs[i].active) for i in range(len(lengths)))
        assert host.playback_count() == n_alive  # inactive nodes were reaped (audio_spatializer.cpp:473-482)
        # errors mirror :521-522
        assert host.get_mixed_frames(C, F)[0] == -11 if C < 4 else True
        host.close()


def test_host_effect_copy_streams_end_mid_block(gas, ob):
    drive(gas, ob, gas.capi.KIND_EFFECT, ob.KIND_EFFECT, (), (), 1, [700, 1300, 2048, 5000, 1024 + 500], 512, 8)


def test_host_mix_channel_two_pairs_with_latch(gas, ob):
    drive(gas, ob, gas.capi.KIND_3D_MIX, ob.KIND_3D_MIX, (), (), 2, [900, 3000, 2000], 512, 7)


def test_host_hrtf_tail_rings_out_then_gates(gas, ob):
    """After the stream ends the HRIR tail keeps the playback active until its peak drops under -80 dB."""
    from godot_audio_spatializer_amd import synth

    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=8)
    drive(gas, ob, gas.capi.KIND_EFFECT, ob.KIND_EFFECT, (gas.capi.FX_HRTF,), (ob.FX_HRTF,), 1, [600, 1500, 4000], 512, 9, hrir=hrir)


def test_host_stop_playback(gas, ob):
    drive(gas, ob, gas.capi.KIND_3D_PROCESS, ob.KIND_3D_PROCESS, (), (), 1, [4000, 4000, 4000], 512, 5, stop_at=(2, 1))


def test_host_device_stream_mode(gas, ob):
    """The host layer over HBM-resident PCM: no source frames cross PCIe; same results as the CPU-sampled host."""
    from godot_audio_spatializer_amd import synth

    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=8)
    drive(gas, ob, gas.capi.KIND_EFFECT, ob.KIND_EFFECT, (gas.capi.FX_HRTF,), (ob.FX_HRTF,), 1, [600, 1500, 4000, 2048], 512, 10, hrir=hrir, device_streams=True)
    drive(gas, ob, gas.capi.KIND_3D_MIX, ob.KIND_3D_MIX, (), (), 2, [900, 3000, 2000], 512, 7, device_streams=True, stop_at=(3, 1))


def test_host_modes_do_not_mix(gas):
    with gas.SpatializerContext(max_sources=4, frames=512) as ctx:
        host = gas.capi.BatchedSpatializerHost(ctx, gas.capi.KIND_EFFECT)
        host.start_playback_array(np.zeros((1024, 2), np.float32))
        with pytest.raises(gas.GasError) as ei:
            host.start_playback_device_stream(ctx.stream_create(np.zeros(1024, np.int16)))
        assert ei.value.status == -10
        host.close()
