"""ctypes binding of oracle/libgas_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY (see oracle/gas_oracle.h): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product
package.  PARITY UNPINNED: the reference has no golden vectors and cannot be
built here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgas_oracle.so")

MAX_CHANNELS = 4
LOOKAHEAD = 64
HRTF_TAPS = 256
ER_TAPS = 8
MAX_EFFECTS = 4

FX_HIGHSHELF = 1
FX_EARLY_REFLECTIONS = 2
FX_HRTF = 3
FX_LOWPASS, FX_HIGHPASS, FX_BANDPASS, FX_NOTCH, FX_LOWSHELF, FX_AMPLIFY = 4, 5, 6, 7, 8, 9

KIND_3D_MIX = 0
KIND_3D_PROCESS = 1
KIND_EFFECT = 2


class Frame(C.Structure):
    _fields_ = [("l", C.c_float), ("r", C.c_float)]


class Coeffs(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("a1", "a2", "b0", "b1", "b2")]


class Processor(C.Structure):
    _fields_ = [("coeffs", Coeffs), ("incr", Coeffs)] + [(k, C.c_float) for k in ("ha1", "ha2", "hb1", "hb2")]


class Params(C.Structure):
    _fields_ = [
        ("mix_volumes", (C.c_float * 2) * MAX_CHANNELS),
        ("pitch_scale", C.c_float),
        ("linear_attenuation", C.c_float),
        ("attenuation_filter_cutoff_hz", C.c_float),
        ("update_parameters", C.c_uint32),
        ("hrtf_gain", C.c_float),
        ("hrtf_dir", C.c_uint32),
        ("fx_shelf_gain", C.c_float),
        ("fx_shelf_cutoff_hz", C.c_float),
        ("er_gain", C.c_float * ER_TAPS),
        ("er_delay", C.c_uint32 * ER_TAPS),
    ]


# numpy view of the same 128-byte layout (shared by the product's gas_params).
PARAMS_DTYPE = np.dtype(
    [
        ("mix_volumes", np.float32, (MAX_CHANNELS, 2)),
        ("pitch_scale", np.float32),
        ("linear_attenuation", np.float32),
        ("attenuation_filter_cutoff_hz", np.float32),
        ("update_parameters", np.uint32),
        ("hrtf_gain", np.float32),
        ("hrtf_dir", np.uint32),
        ("fx_shelf_gain", np.float32),
        ("fx_shelf_cutoff_hz", np.float32),
        ("er_gain", np.float32, (ER_TAPS,)),
        ("er_delay", np.uint32, (ER_TAPS,)),
    ]
)
assert PARAMS_DTYPE.itemsize == C.sizeof(Params) == 128


class PData3D(C.Structure):
    _fields_ = [
        ("prev_mix_volumes", (C.c_float * 2) * MAX_CHANNELS),
        ("prev_count", C.c_int32),
        ("filter_processors", Processor * 8),
    ]


class FxState(C.Structure):
    _fields_ = [
        ("shelf", Processor * 2),
        ("ring", C.POINTER(Frame)),
        ("ring_frames", C.c_uint32),
        ("ring_pos", C.c_uint32),
        ("hist", C.c_float * (HRTF_TAPS - 1)),
        ("prev_gain", C.c_float),
        ("prev_dir_plus1", C.c_int32),
        ("cutoff_hz", C.c_float),
        ("resonance", C.c_float),
        ("gain", C.c_float),
        ("volume_db", C.c_float),
        ("amp_mix_volume_db", C.c_float),
        ("amp_started", C.c_int32),
    ]


class PDataEffect(C.Structure):
    _fields_ = [("n_effects", C.c_int32), ("kinds", C.c_int32 * MAX_EFFECTS), ("fx", FxState * MAX_EFFECTS)]


class Hrtf(C.Structure):
    _fields_ = [("hrir", C.POINTER(C.c_float)), ("dirs", C.c_uint32), ("impl", C.c_int32), ("spec", C.POINTER(C.c_float)), ("spec_len", C.c_int32), ("crossfade", C.c_int32)]


class Playback(C.Structure):
    _fields_ = [
        ("stream", C.POINTER(Frame)),
        ("stream_frames", C.c_int64),
        ("stream_pos", C.c_int64),
        ("active", C.c_int32),
        ("has_frames", C.c_int32),
        ("lookahead", Frame * LOOKAHEAD),
        ("pd3d", PData3D),
        ("pdfx", PDataEffect),
        ("last_peak", C.c_float * 2),
        ("resampled", C.c_int32),
        ("paused", C.c_int32),
        ("mix_offset", C.c_uint64),
    ]


class Instance(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("channel_count", C.c_int32),
        ("mix_rate", C.c_float),
        ("disable_threshold_db", C.c_float),
        ("channel_mixed", C.c_int32 * MAX_CHANNELS),
        ("hrtf", C.POINTER(Hrtf)),
        ("playback_buffer", C.POINTER(Frame)),
        ("process_buffer", C.POINTER(Frame)),
        ("temp_buffer", C.POINTER(Frame)),
        ("fx_temp", C.POINTER(Frame)),
        ("mix_buffer", C.POINTER(Frame) * MAX_CHANNELS),
        ("mix_buffer_size", C.c_int32),
    ]


class Spat3DConfig(C.Structure):
    _fields_ = [
        ("attenuation_model", C.c_int32),
        ("unit_size", C.c_float),
        ("max_distance", C.c_float),
        ("panning_strength", C.c_float),
        ("emission_angle_enabled", C.c_int32),
        ("emission_angle", C.c_float),
        ("emission_angle_filter_attenuation_db", C.c_float),
        ("attenuation_filter_cutoff_hz", C.c_float),
        ("attenuation_filter_db", C.c_float),
        ("doppler_tracking", C.c_int32),
        ("doppler_speed_of_sound", C.c_float),
        ("global_panning_strength", C.c_float),
        ("speaker_mode", C.c_int32),
        ("hrtf_n_az", C.c_uint32),
        ("hrtf_n_el", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


assert C.sizeof(Spat3DConfig) == 64


class BatchState(C.Structure):
    _fields_ = [("pd3d", PData3D), ("pdfx", PDataEffect)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile). Building the checker is not using it."""
    src = os.path.join(_HERE, "gas_oracle.c")
    hdr = os.path.join(_HERE, "gas_oracle.h")
    if (
        not force
        and os.path.exists(_LIB_PATH)
        and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))
    ):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libgas_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    FP = C.POINTER(Frame)
    L.gaso_db_to_linear.restype = C.c_float
    L.gaso_db_to_linear.argtypes = [C.c_float]
    L.gaso_linear_to_db.restype = C.c_float
    L.gaso_linear_to_db.argtypes = [C.c_float]
    L.gaso_highshelf_coeffs.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(Coeffs)]
    L.gaso_processor_update_coeffs.argtypes = [C.POINTER(Processor), C.POINTER(Coeffs), C.c_int]
    L.gaso_processor_process_one.restype = C.c_float
    L.gaso_processor_process_one.argtypes = [C.POINTER(Processor), C.c_float]
    L.gaso_processor_process_one_interp.restype = C.c_float
    L.gaso_processor_process_one_interp.argtypes = [C.POINTER(Processor), C.c_float]
    L.gaso_process_frames_3d.argtypes = [C.POINTER(Params), C.POINTER(PData3D), C.c_void_p, C.c_void_p, C.c_int, C.c_float]
    L.gaso_mix_channel_3d.argtypes = [C.POINTER(Params), C.POINTER(PData3D), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float]
    L.gaso_fx_process.argtypes = [C.c_int, C.POINTER(Params), C.POINTER(FxState), C.POINTER(Hrtf), C.c_void_p, C.c_void_p, C.c_int, C.c_float]
    L.gaso_process_frames_effect.restype = C.c_uint32
    L.gaso_process_frames_effect.argtypes = [C.POINTER(Params), C.POINTER(PDataEffect), C.POINTER(Hrtf), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float]
    L.gaso_fetch_source.argtypes = [C.POINTER(Playback), C.c_void_p, C.c_int, C.c_float, C.c_float]
    L.gaso_mix_from_playback_list.argtypes = [C.POINTER(Instance), C.POINTER(C.POINTER(Params)), C.POINTER(C.POINTER(Playback)), C.c_int, C.c_int]
    L.gaso_check_channel_mixed.restype = C.c_int
    L.gaso_check_channel_mixed.argtypes = [C.POINTER(Instance), C.c_int]
    L.gaso_get_mixed_frames.restype = C.c_int
    L.gaso_get_mixed_frames.argtypes = [C.POINTER(Instance), C.POINTER(C.POINTER(Params)), C.POINTER(C.POINTER(Playback)), C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.gaso_bus_map.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gaso_batch_block.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(BatchState), C.POINTER(Hrtf), C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gaso_calc_spatialization.restype = C.c_int
    L.gaso_calc_spatialization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_void_p]
    L.gaso_calc_spatialization_area.restype = C.c_int
    L.gaso_calc_spatialization_area.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gaso_hrtf_prepare.argtypes = [C.POINTER(Hrtf), C.c_int]
    L.gaso_hrtf_release.argtypes = [C.POINTER(Hrtf)]
    L.gaso_hrtf_ols_radix2.argtypes = [C.POINTER(Params), C.POINTER(FxState), C.POINTER(Hrtf), C.c_void_p, C.c_void_p, C.c_int]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def make_hrtf(hrir, impl=0, crossfade=False):
    """hrir: float32 [dirs][2][256]. Returns (Hrtf struct, keepalive)."""
    hrir = np.ascontiguousarray(hrir, dtype=np.float32)
    assert hrir.ndim == 3 and hrir.shape[1] == 2 and hrir.shape[2] == HRTF_TAPS
    h = Hrtf(hrir.ctypes.data_as(C.POINTER(C.c_float)), hrir.shape[0], impl, None, 0, int(crossfade))
    h._keep = hrir
    return h


class BatchOracle:
    """State for n_src independent sources of one kind, advanced one callback at a time.

    kind: KIND_3D_MIX / KIND_3D_PROCESS / KIND_EFFECT; chain: list of FX_* for KIND_EFFECT.
    """

    def __init__(self, kind, n_src, frames, channel_count=1, chain=(), hrir=None, mix_rate=48000.0, er_ring_frames=4096, hrtf_impl=0, crossfade=False):
        self.kind = kind
        self.n_src = n_src
        self.frames = frames
        self.channel_count = channel_count if kind == KIND_3D_MIX else 1
        self.mix_rate = float(mix_rate)
        self.states = (BatchState * n_src)()
        self.hrtf = make_hrtf(hrir, hrtf_impl, crossfade) if hrir is not None else None
        if self.hrtf is not None and hrtf_impl == 1:
            fft_len = 1
            while fft_len < frames + HRTF_TAPS - 1:
                fft_len *= 2
            lib().gaso_hrtf_prepare(C.byref(self.hrtf), fft_len)
        self._rings = []
        for s in range(n_src):
            pd = self.states[s].pdfx
            pd.n_effects = len(chain)
            for j, k in enumerate(chain):
                pd.kinds[j] = k
                pd.fx[j].cutoff_hz, pd.fx[j].resonance, pd.fx[j].gain, pd.fx[j].volume_db = 2000.0, 0.5, 1.0, 0.0  # [ENGINE] resource defaults
                if k == FX_EARLY_REFLECTIONS:
                    ring = np.zeros((er_ring_frames, 2), dtype=np.float32)
                    self._rings.append(ring)
                    pd.fx[j].ring = ring.ctypes.data_as(C.POINTER(Frame))
                    pd.fx[j].ring_frames = er_ring_frames

    def set_fx_settings(self, s, j, cutoff_hz=None, resonance=None, gain=None, volume_db=None):
        """What a script sets on effect j's resource for source s (the GAS_FX_LOWPASS .. GAS_FX_AMPLIFY kinds)."""
        fx = self.states[s].pdfx.fx[j]
        for name, v in (("cutoff_hz", cutoff_hz), ("resonance", resonance), ("gain", gain), ("volume_db", volume_db)):
            if v is not None:
                setattr(fx, name, float(v))

    def block(self, params, src, want64=False):
        """params: PARAMS_DTYPE[n_src]; src: float32 [n_src][F][2]. Returns (mix [C][F][2], peaks [n_src][2], mix64|None)."""
        params = np.ascontiguousarray(params, dtype=PARAMS_DTYPE)
        src = np.ascontiguousarray(src, dtype=np.float32)
        assert params.shape == (self.n_src,) and src.shape == (self.n_src, self.frames, 2)
        mix = np.zeros((self.channel_count, self.frames, 2), dtype=np.float32)
        mix64 = np.zeros((self.channel_count, self.frames, 2), dtype=np.float64) if want64 else None
        peaks = np.zeros((self.n_src, 2), dtype=np.float32)
        lib().gaso_batch_block(
            self.kind,
            self.channel_count,
            _ptr(params),
            self.states,
            C.byref(self.hrtf) if self.hrtf is not None else None,
            _ptr(src),
            self.n_src,
            self.frames,
            self.mix_rate,
            _ptr(mix),
            _ptr(mix64) if want64 else None,
            _ptr(peaks),
        )
        return mix, peaks, mix64


def calc_spatialization(cfgs, cfg_index, poses, listeners, was_further, out_params):
    """Row-by-row gaso_calc_spatialization over numpy structured arrays laid out like the product's PODs
    (64-byte config, 48-byte pose, 64-byte listener, 128-byte params).  was_further: int32[n], updated in place.
    Returns in_range[n]."""
    n = len(poses)
    L = lib()
    in_range = np.zeros(n, np.int32)
    for i in range(n):
        ci = int(cfg_index[i]) if cfg_index is not None else 0
        wf = C.c_int32(int(was_further[i]))
        in_range[i] = L.gaso_calc_spatialization(cfgs.ctypes.data + 64 * ci, poses.ctypes.data + 48 * i, listeners.ctypes.data, len(listeners), C.byref(wf), out_params.ctypes.data + 128 * i)
        was_further[i] = wf.value
    return in_range


AREA_SEND_DTYPE = np.dtype([("using_reverb_bus", np.uint32), ("reverb_uniformity", np.float32), ("reverb_amount", np.float32), ("present", np.uint32)])


def calc_spatialization_areas(cfgs, cfg_index, poses, listeners, was_further, areas, listener_area_pos, out_params):
    """Row-by-row gaso_calc_spatialization_area.  areas: AREA_SEND_DTYPE[n]; listener_area_pos: float32 [n][L][3] or None.
    Returns (in_range[n], reverb[n][4][2])."""
    n = len(poses)
    L = lib()
    in_range = np.zeros(n, np.int32)
    areas = np.ascontiguousarray(areas, dtype=AREA_SEND_DTYPE)
    lap = np.ascontiguousarray(listener_area_pos, dtype=np.float32) if listener_area_pos is not None else None
    reverb = np.zeros((n, 4, 2), np.float32)
    nl = len(listeners)
    for i in range(n):
        ci = int(cfg_index[i]) if cfg_index is not None else 0
        wf = C.c_int32(int(was_further[i]))
        in_range[i] = L.gaso_calc_spatialization_area(
            C.c_void_p(cfgs.ctypes.data + 64 * ci), C.c_void_p(poses.ctypes.data + 48 * i), C.c_void_p(listeners.ctypes.data), nl, C.byref(wf),
            C.c_void_p(areas.ctypes.data + 16 * i), C.c_void_p(lap.ctypes.data + 12 * nl * i) if lap is not None else None,
            C.c_void_p(out_params.ctypes.data + 128 * i), C.c_void_p(reverb.ctypes.data + 32 * i))
        was_further[i] = wf.value
    return in_range, reverb
