"""The C++ host layer (include/gas_amd_host.h) on the GPU vs the oracle's restatement of the whole mixer
(audio_spatializer.cpp:326-527): playbacks of different lengths ending mid-callback, fade-out, silence gate,
channel latch, stop, list GC."""
import numpy as np
import pytest

from helpers import mix_matches
from test_oracle_mixer import Rig

pytestmark = pytest.mark.gpu


def drive(gas, ob, kind, okind, chain, ochain, channel_count, lengths, F, callbacks, hrir=None, stop_at=None, seed=0, device_streams=False, pause_plan=None):
    """pause_plan: {callback: [(playback index, paused), ...]} applied before that callback on both sides."""
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
            for i, paused in (pause_plan or {}).get(cb, ()):
                assert host.set_playback_paused(ids[i], paused) == 0
                rig.pbs[i].paused = int(paused)
            for c in range(C):
                rc, got = host.get_mixed_frames(c, F)
                orc, want = rig.get_mixed_frames(c)
                assert rc == 0 and orc == 0
                assert mix_matches(got, want), f"callback {cb} channel {c}"
            for i in range(len(lengths)):
                assert host.is_playback_active(ids[i]) == bool(rig.pbs[i].active), f"callback {cb} playback {i}"
                if rig.pbs[i].active:  # get_playback_position (audio_spatializer.cpp:144-157): frames consumed, lookahead included
                    assert host.is_playback_paused(ids[i]) == bool(rig.pbs[i].paused)
                    assert host.get_playback_position(ids[i]) == rig.pbs[i].stream_pos, f"callback {cb} playback {i}"
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


@pytest.mark.parametrize("device_streams", [False, True])
def test_host_pause_resume_per_playback(gas, ob, device_streams):
    """set_playback_paused per playback (audio_spatializer.cpp:115-122 moved from the proxies to the playback): pause
    mid-stream, resume, pause during the HRIR ring-out; a paused playback keeps its state, lookahead and position, is
    neither gated nor reaped, and the others play on."""
    from godot_audio_spatializer_amd import synth

    hrir = synth.synthetic_hrir(np.random.default_rng(7), dirs=8)
    plan = {
        1: [(1, True)],  # mid-stream
        2: [(0, True)],  # ringing out: its 600-frame stream ended in callback 1, the gate would take it after callback 2
        3: [(1, False)],
        5: [(0, False)],  # rings out now
        6: [(2, True)],
        8: [(2, False)],
    }
    drive(gas, ob, gas.capi.KIND_EFFECT, ob.KIND_EFFECT, (gas.capi.FX_HRTF,), (ob.FX_HRTF,), 1, [600, 1500, 4000, 2048], 512, 14, hrir=hrir, pause_plan=plan, device_streams=device_streams)
    drive(gas, ob, gas.capi.KIND_3D_MIX, ob.KIND_3D_MIX, (), (), 2, [900, 3000, 2000], 512, 11, pause_plan={1: [(0, True), (2, True)], 4: [(0, False)], 5: [(2, False)]}, device_streams=device_streams)


def test_host_pause_of_everything_gives_silence_and_keeps_the_list(gas, ob):
    F = 512
    with gas.SpatializerContext(max_sources=4, frames=F) as ctx:
        host = gas.capi.BatchedSpatializerHost(ctx, gas.capi.KIND_EFFECT)
        ids = [host.start_playback_array(np.full((4 * F, 2), 0.25, np.float32)) for _ in range(2)]
        p = np.zeros(1, gas.capi.PARAMS_DTYPE)
        for i in ids:
            host.set_spatializer_parameters(i, p[0])
        rc, a = host.get_mixed_frames(0, F)
        assert rc == 0 and a[64:].any()
        for i in ids:
            host.set_playback_paused(i, True)
        pos = [host.get_playback_position(i) for i in ids]
        for _ in range(3):
            rc, m = host.get_mixed_frames(0, F)
            assert rc == 0 and not m.any()
        assert host.playback_count() == 2 and all(host.is_playback_active(i) and host.is_playback_paused(i) for i in ids)
        assert [host.get_playback_position(i) for i in ids] == pos
        assert host.set_playback_paused(12345, True) == -3 and not host.is_playback_paused(12345)  # unknown id (audio_spatializer.cpp:161-170: false)
        assert host.get_playback_position(12345) == 0  # :152-155
        host.set_playback_paused(ids[0], False)
        rc, m = host.get_mixed_frames(0, F)
        np.testing.assert_array_equal(m[:, 0], np.full(F, 0.25, np.float32))  # resumes where it stopped: no new start silence
        host.close()


def test_host_process_effects_hook_and_release(gas, ob):
    """_process_effects (audio_spatializer_effect.cpp:39,90-92): the example sets its high-shelf's gain from the
    spatializer parameters on the audio thread (gd_spatializer_instance.gd:125-127).  Here the hook edits the playback's
    parameter row; the result must be what the oracle gives with the edited rows.  Plus the release callback that stands
    in for the list node's Ref<AudioStreamPlayback> (audio_spatializer.cpp:538-547)."""
    from godot_audio_spatializer_amd import synth

    F, n = 512, 3
    rng = np.random.default_rng(5)
    streams = [rng.uniform(-0.5, 0.5, (m, 2)).astype(np.float32) for m in (3000, 3000, 1200)]
    params = synth.draw_params(rng, n, dirs=8)
    params["fx_shelf_gain"] = 1.0
    params["fx_shelf_cutoff_hz"] = 3000.0
    cursors = [0] * n

    with gas.SpatializerContext(max_sources=8, frames=F) as ctx:
        host = gas.capi.BatchedSpatializerHost(ctx, gas.capi.KIND_EFFECT, (gas.capi.FX_HIGHSHELF,))
        released, calls = [], []
        host.set_release_fn(lambda pid, user: released.append((pid, user)))

        def feed(i):
            def mix(buf, rate, frames):
                got = min(frames, len(streams[i]) - cursors[i])
                buf[:got] = streams[i][cursors[i]: cursors[i] + got]
                buf[got:] = 0.0
                cursors[i] += got
                return got

            return mix

        ids = [None] * n
        for i in reversed(range(n)):
            ids[i] = host.start_playback(feed(i), user=1000 + i)
            host.set_spatializer_parameters(ids[i], params[i])

        def hook(pid, row):
            calls.append(pid)
            if pid == ids[1]:
                return False  # untouched: keeps gain 1
            row["fx_shelf_gain"] = 0.25 + 0.5 * row["linear_attenuation"]
            return True

        host.set_process_effects_fn(hook)
        rig = Rig(ob, ob.KIND_EFFECT, streams, F, chain=(ob.FX_HIGHSHELF,))
        rig.params[:] = params.astype(ob.PARAMS_DTYPE)
        for i in (0, 2):
            rig.params["fx_shelf_gain"][i] = np.float32(0.25) + np.float32(0.5) * params["linear_attenuation"][i]
        for cb in range(6):
            if cb == 3:
                host.set_process_effects_fn(None)  # the edited rows stay until the next set_spatializer_parameters
                n_calls = len(calls)
            rc, got = host.get_mixed_frames(0, F)
            orc, want = rig.get_mixed_frames(0)
            assert rc == 0 and orc == 0 and mix_matches(got, want), f"callback {cb}"
        assert len(calls) == n_calls and set(calls) == set(ids)
        assert not host.is_playback_active(ids[2]) and released == []  # reaped by the audio thread, not yet released
        assert host.collect_released() == 1 and released == [(ids[2], 1002)]
        host.stop_playback(ids[0])
        host.get_mixed_frames(0, F)
        host.close()  # destroy releases the rest
        assert sorted(released) == sorted((ids[i], 1000 + i) for i in range(n))


def test_host_engine_effect_chain_with_settings(gas, ob):
    """The batched host over a chain of the engine-effect kinds: gas_host_set_effect_settings queues a playback's
    settings like its parameters (what a script writes to the AudioEffect resources); ends, fade-outs and the gate as
    for every other chain."""
    from godot_audio_spatializer_amd import synth

    K = gas.capi
    F, lengths = 256, [700, 1900, 1300, 2600]
    chain, ochain = (K.FX_LOWPASS, K.FX_AMPLIFY), (ob.FX_LOWPASS, ob.FX_AMPLIFY)
    rng = np.random.default_rng(12)
    streams = [rng.uniform(-0.5, 0.5, (m, 2)).astype(np.float32) for m in lengths]
    params = synth.draw_params(rng, len(lengths), dirs=8, frames=F)
    with gas.SpatializerContext(max_sources=8, frames=F) as ctx:
        host = K.BatchedSpatializerHost(ctx, K.KIND_EFFECT, chain)
        ids = [None] * len(lengths)
        for i in reversed(range(len(lengths))):
            ids[i] = host.start_playback_array(streams[i])
            host.set_spatializer_parameters(ids[i], params[i])
        rig = Rig(ob, ob.KIND_EFFECT, streams, F, chain=ochain)
        rig.params[:] = params.astype(ob.PARAMS_DTYPE)
        for i in range(len(lengths)):
            for j in range(2):
                fx = rig.pbs[i].pdfx.fx[j]
                fx.cutoff_hz, fx.resonance, fx.gain, fx.volume_db = 2000.0, 0.5, 1.0, 0.0
        for cb in range(14):
            if cb in (1, 4, 7):
                for i in range(len(lengths)):
                    st = ctx.fx_settings_defaults(1)
                    st["filter_cutoff_hz"][0, 0] = 400.0 * (i + 1) * (cb + 1)
                    st["filter_resonance"][0, 0] = 0.5 + 0.1 * i
                    st["amplify_volume_db"][0, 1] = -3.0 * i - cb
                    assert host.set_effect_settings(ids[i], st) in (0, -3)  # -3: ended and reaped meanwhile
                    fx0, fx1 = rig.pbs[i].pdfx.fx[0], rig.pbs[i].pdfx.fx[1]
                    fx0.cutoff_hz, fx0.resonance = float(st["filter_cutoff_hz"][0, 0]), float(st["filter_resonance"][0, 0])
                    fx1.volume_db = float(st["amplify_volume_db"][0, 1])
            rc, got = host.get_mixed_frames(0, F)
            orc, want = rig.get_mixed_frames(0)
            assert rc == 0 and orc == 0 and mix_matches(got, want), f"callback {cb}"
            for i in range(len(lengths)):
                assert host.is_playback_active(ids[i]) == bool(rig.pbs[i].active), f"callback {cb} playback {i}"
        host.close()


def test_host_modes_do_not_mix(gas):
    with gas.SpatializerContext(max_sources=4, frames=512) as ctx:
        host = gas.capi.BatchedSpatializerHost(ctx, gas.capi.KIND_EFFECT)
        host.start_playback_array(np.zeros((1024, 2), np.float32))
        with pytest.raises(gas.GasError) as ei:
            host.start_playback_device_stream(ctx.stream_create(np.zeros(1024, np.int16)))
        assert ei.value.status == -10
        host.close()
