"""The oracle's restatement of the mixer (audio_spatializer.cpp:326-527): source window, fade-out,
accumulate order, peak, silence gate, once-per-callback latch, frame-count rule, bus map."""
import ctypes as C

import numpy as np
import pytest


class Rig:
    """One AudioSpatializerInstance with n playbacks over synthetic streams."""

    def __init__(self, ob, kind, streams, F, channel_count=1, chain=(), hrir=None):
        self.ob, self.F, self.n = ob, F, len(streams)
        self.C = channel_count
        self.inst = ob.Instance()
        self.inst.kind = kind
        self.inst.channel_count = channel_count if kind == ob.KIND_3D_MIX else 1
        self.inst.mix_rate = 48000.0
        self.inst.disable_threshold_db = -80.0
        for c in range(ob.MAX_CHANNELS):
            self.inst.channel_mixed[c] = 1  # audio_spatializer.cpp:78-80
        self.bufs = [np.zeros((F + 64, 2), np.float32)] + [np.zeros((F, 2), np.float32) for _ in range(3)]
        self.inst.playback_buffer, self.inst.process_buffer, self.inst.temp_buffer, self.inst.fx_temp = [b.ctypes.data_as(C.POINTER(ob.Frame)) for b in self.bufs]
        self.mix = [np.zeros((F, 2), np.float32) for _ in range(ob.MAX_CHANNELS)]
        for c in range(ob.MAX_CHANNELS):
            self.inst.mix_buffer[c] = self.mix[c].ctypes.data_as(C.POINTER(ob.Frame))
        self.hrtf = ob.make_hrtf(hrir) if hrir is not None else None
        if self.hrtf is not None:
            self.inst.hrtf = C.pointer(self.hrtf)
        self.streams = [np.ascontiguousarray(s, np.float32) for s in streams]
        self.pbs = (ob.Playback * self.n)()
        for i, s in enumerate(self.streams):
            self.pbs[i].stream = s.ctypes.data_as(C.POINTER(ob.Frame))
            self.pbs[i].stream_frames = len(s)
            self.pbs[i].active = 1
            self.pbs[i].has_frames = 1
            self.pbs[i].pdfx.n_effects = len(chain)
            for j, k in enumerate(chain):
                self.pbs[i].pdfx.kinds[j] = k
        self.pb_ptrs = (C.POINTER(ob.Playback) * self.n)(*[C.pointer(self.pbs[i]) for i in range(self.n)])
        self.params = np.zeros(self.n, ob.PARAMS_DTYPE)
        self.params["attenuation_filter_cutoff_hz"] = 5000.0
        self.pp = (C.POINTER(ob.Params) * self.n)()

    def _bind(self):
        base = self.params.ctypes.data
        for i in range(self.n):
            self.pp[i] = C.cast(base + i * 128, C.POINTER(self.ob.Params))

    def get_mixed_frames(self, channel, F=None):
        self._bind()
        F = self.F if F is None else F
        out = np.full((F, 2), np.nan, np.float32)
        rc = self.ob.lib().gaso_get_mixed_frames(C.byref(self.inst), self.pp, self.pb_ptrs, self.n, channel, out.ctypes.data, F)
        return rc, out


def test_lookahead_delays_by_64_frames(ob):
    F = 512
    s = np.zeros((4 * F, 2), np.float32)
    s[:, 0] = np.arange(1, 4 * F + 1)
    s[:, 1] = -s[:, 0]
    rig = Rig(ob, ob.KIND_EFFECT, [s], F)  # empty chain = copy
    rc, o = rig.get_mixed_frames(0)
    assert rc == 0
    assert not o[:64].any()  # lookahead starts zeroed (audio_spatializer.cpp:61-63)
    np.testing.assert_array_equal(o[64:, 0], s[: F - 64, 0])
    rc, o2 = rig.get_mixed_frames(0)
    np.testing.assert_array_equal(o2[:, 0], s[F - 64: 2 * F - 64, 0])


def test_fade_out_envelope_and_silence_gate(ob):
    F = 512
    n_stream = F + 200  # second callback is short: 200 real frames
    s = np.ones((n_stream, 2), np.float32)
    rig = Rig(ob, ob.KIND_EFFECT, [s], F)
    rig.get_mixed_frames(0)
    rc, o = rig.get_mixed_frames(0)
    # working buffer = lookahead[64] ++ 200 real ++ zeros; fade applies to indices [200, 264), zero after
    np.testing.assert_array_equal(o[:200, 0], 1.0)
    k = np.arange(64, dtype=np.float32)
    coef = np.cumprod(np.full(64, np.float32(0.96), np.float32)).astype(np.float32)
    env = coef * (np.float32(64.0) - k) / np.float32(64.0)  # 0.96^(k+1) * (64 - k) / 64
    np.testing.assert_allclose(o[200:264, 0], env, rtol=1e-6)
    assert not o[264:].any()
    pb = rig.pbs[0]
    assert pb.has_frames == 0 and pb.active == 1  # peak 1.0 > 1e-4: stays active this callback
    rc, o3 = rig.get_mixed_frames(0)
    assert not o3.any()
    assert rig.pbs[0].active == 0  # zero output <= threshold -> deactivated (audio_spatializer.cpp:464-469)
    rc, o4 = rig.get_mixed_frames(0)
    assert rc == 0 and not o4.any()


def test_fade_truncated_when_stream_ends_near_block_end(ob):
    F = 512
    s = np.ones((F + 500, 2), np.float32)  # second callback: 500 real frames, fade limit 564 > 512
    rig = Rig(ob, ob.KIND_EFFECT, [s], F)
    rig.get_mixed_frames(0)
    rc, o = rig.get_mixed_frames(0)
    np.testing.assert_array_equal(o[:500, 0], 1.0)
    assert np.all(o[500:, 0] < 1.0) and np.all(o[500:, 0] > 0.0)  # only 12 fade steps fit
    assert rig.pbs[0].has_frames == 0


def test_accumulates_in_list_order_and_peaks(ob):
    F, rng = 512, np.random.default_rng(0)
    streams = [rng.uniform(-0.5, 0.5, (2 * F, 2)).astype(np.float32) for _ in range(5)]
    rig = Rig(ob, ob.KIND_EFFECT, streams, F)
    rig.get_mixed_frames(0)
    rc, o = rig.get_mixed_frames(0)
    acc = np.zeros((F, 2), np.float32)
    for s in streams:  # serial f32 +=, array order
        acc += s[F - 64: 2 * F - 64]
    np.testing.assert_array_equal(o, acc)
    for i, s in enumerate(streams):
        w = np.abs(s[F - 64: 2 * F - 64])
        assert tuple(rig.pbs[i].last_peak) == (w[:, 0].max(), w[:, 1].max())


def test_inactive_playbacks_are_skipped(ob):
    F = 512
    streams = [np.full((2 * F, 2), v, np.float32) for v in (0.25, 0.5)]
    rig = Rig(ob, ob.KIND_EFFECT, streams, F)
    rig.pbs[1].active = 0
    rig.get_mixed_frames(0)
    rc, o = rig.get_mixed_frames(0)
    np.testing.assert_array_equal(o, 0.25)
    assert rig.pbs[1].stream_pos == 0  # not even sampled (audio_spatializer.cpp:354-357)


def test_channel_mixed_latch(ob):
    """audio_spatializer.cpp:494-508: with C = 2 each callback asks channel 0 then 1; only the first remixes."""
    F = 512
    s = np.ones((8 * F, 2), np.float32)
    rig = Rig(ob, ob.KIND_3D_MIX, [s], F, channel_count=2)
    rig.params["mix_volumes"][0, 0] = [1.0, 1.0]
    rig.params["mix_volumes"][0, 1] = [0.5, 0.5]
    rig.get_mixed_frames(0)
    pos = rig.pbs[0].stream_pos
    assert pos == F
    rig.get_mixed_frames(1)
    assert rig.pbs[0].stream_pos == F  # channel 1 served from the same mix
    rig.get_mixed_frames(0)
    assert rig.pbs[0].stream_pos == 2 * F
    # channel_count == 1: every call remixes
    rig1 = Rig(ob, ob.KIND_EFFECT, [s], F)
    rig1.get_mixed_frames(0)
    rig1.get_mixed_frames(0)
    assert rig1.pbs[0].stream_pos == 2 * F


def test_frame_count_and_channel_rules(ob):
    F = 512
    rig = Rig(ob, ob.KIND_EFFECT, [np.ones((4 * F, 2), np.float32)], F)
    assert rig.get_mixed_frames(0)[0] == 0
    rig.inst.channel_mixed[0] = 0  # served from the existing mix: size must match (audio_spatializer.cpp:522)
    assert rig.get_mixed_frames(0, F=256)[0] == -1
    assert rig.get_mixed_frames(1)[0] == -1  # "Unexpected channel" (:521)


def test_bus_map(ob):
    L = ob.lib()
    bus = np.array([[0.5, 0.25], [0.2, 0.1], [0, 0], [0.3, 0.3]], np.float32)
    mixv = np.array([[0.5, 0.5], [0.0, 0.4], [0, 0], [0.6, 0.3]], np.float32)
    out = np.zeros((4, 2), np.float32)
    L.gaso_bus_map(1, 1, bus.ctypes.data, mixv.ctypes.data, out.ctypes.data)
    np.testing.assert_allclose(out, [[0, 0], [0.0, 0.25], [0, 0], [0, 0]])  # only channel 1, 0 where mix <= 0
    L.gaso_bus_map(0, 1, bus.ctypes.data, mixv.ctypes.data, out.ctypes.data)
    np.testing.assert_array_equal(out, mixv)  # not mixing channels: raw mix volumes
