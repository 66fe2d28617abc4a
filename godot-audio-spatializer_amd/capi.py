"""ctypes binding of libgas_amd.so (include/gas_amd.h).

Fails loudly when the HIP library is missing or cannot be loaded: there is no CPU path.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

MAX_CHANNELS = 4
LOOKAHEAD = 64
HRTF_TAPS = 256
ER_TAPS = 8

KIND_3D_MIX = 0
KIND_3D_PROCESS = 1
KIND_EFFECT = 2
FX_HIGHSHELF = 1
FX_EARLY_REFLECTIONS = 2
FX_HRTF = 3
FX_LOWPASS, FX_HIGHPASS, FX_BANDPASS, FX_NOTCH, FX_LOWSHELF, FX_AMPLIFY = 4, 5, 6, 7, 8, 9
MAX_EFFECTS = 4
# gas_fx_settings: settings of the engine-effect kinds by chain position (64 bytes)
FX_SETTINGS_DTYPE = np.dtype([("filter_cutoff_hz", np.float32, (MAX_EFFECTS,)), ("filter_resonance", np.float32, (MAX_EFFECTS,)), ("filter_gain", np.float32, (MAX_EFFECTS,)), ("amplify_volume_db", np.float32, (MAX_EFFECTS,))])
MEM_HOST = 0
MEM_DEVICE = 1
FLAG_PEAKS_DRAINING_ONLY = 1
FLAG_HRTF_CROSSFADE = 2
FLAG_DIRECTION_ORDER = 4
FLAG_PIPELINED_MIX = 8
FLAG_DIRECTION_RUNS = 16
FLAG_XCD_ORDER = 32
FLAG_BATCHED_LAUNCH = 64

STATUS = {
    0: "GAS_OK",
    -1: "GAS_ERR_INVALID_ARGUMENT",
    -2: "GAS_ERR_OUT_OF_SLOTS",
    -3: "GAS_ERR_BAD_SLOT",
    -4: "GAS_ERR_FRAME_COUNT",
    -5: "GAS_ERR_NO_HRTF",
    -6: "GAS_ERR_UNSUPPORTED_CHAIN",
    -7: "GAS_ERR_DEVICE",
    -8: "GAS_ERR_NO_DEVICE",
    -9: "GAS_ERR_OUT_OF_MEMORY",
    -10: "GAS_ERR_KIND_MISMATCH",
    -11: "GAS_ERR_BAD_CHANNEL",
    -12: "GAS_ERR_NO_PARAMS",
}

# gas_params, 128 bytes (include/gas_amd.h)
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
assert PARAMS_DTYPE.itemsize == 128


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("max_sources", C.c_uint32),
        ("frames", C.c_uint32),
        ("channel_count", C.c_uint32),
        ("mix_rate", C.c_float),
        ("er_ring_frames", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class Profile(C.Structure):
    _fields_ = [
        ("launches", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("bytes_per_launch", C.c_uint64),
        ("kernel_name", C.c_char * 64),
        ("callbacks_per_launch", C.c_uint32),
        ("reserved", C.c_uint32),
        ("bytes_per_callback_formula", C.c_uint64),
    ]


SPAT3D_CONFIG_DTYPE = np.dtype(
    [
        ("attenuation_model", np.int32),
        ("unit_size", np.float32),
        ("max_distance", np.float32),
        ("panning_strength", np.float32),
        ("emission_angle_enabled", np.int32),
        ("emission_angle", np.float32),
        ("emission_angle_filter_attenuation_db", np.float32),
        ("attenuation_filter_cutoff_hz", np.float32),
        ("attenuation_filter_db", np.float32),
        ("doppler_tracking", np.int32),
        ("doppler_speed_of_sound", np.float32),
        ("global_panning_strength", np.float32),
        ("speaker_mode", np.int32),
        ("hrtf_n_az", np.uint32),
        ("hrtf_n_el", np.uint32),
        ("reserved", np.uint32),
    ]
)
POSE_DTYPE = np.dtype([("position", np.float32, (3,)), ("volume_db", np.float32), ("velocity", np.float32, (3,)), ("max_db", np.float32), ("forward", np.float32, (3,)), ("pitch_scale", np.float32)])
AREA_SEND_DTYPE = np.dtype([("using_reverb_bus", np.uint32), ("reverb_uniformity", np.float32), ("reverb_amount", np.float32), ("present", np.uint32)])
MAX_MORE_SENDS = 4
BUS_ROUTE_DTYPE = np.dtype([("dry_bus", np.uint32), ("send_bus", np.uint32), ("send", np.float32, (MAX_CHANNELS, 2)), ("more_bus", np.uint32, (MAX_MORE_SENDS,)), ("more_send", np.float32, (MAX_MORE_SENDS, MAX_CHANNELS, 2))])
BUS_NONE = 0xFFFFFFFF


def bus_routes(n):
    """n default routes: dry bus 0, no sends (gas_bus_route with every send slot GAS_BUS_NONE)."""
    r = np.zeros(n, BUS_ROUTE_DTYPE)
    r["send_bus"] = BUS_NONE
    r["more_bus"] = BUS_NONE
    return r
LISTENER_DTYPE = np.dtype([("basis", np.float32, (3, 3)), ("origin", np.float32, (3,)), ("velocity", np.float32, (3,)), ("pad", np.float32)])
assert SPAT3D_CONFIG_DTYPE.itemsize == 64 and POSE_DTYPE.itemsize == 48 and LISTENER_DTYPE.itemsize == 64


def default_spat3d_config(n=1, **kw):
    """AudioSpatializer3D defaults (audio_spatializer_3d.h:171-187), stereo, global panning strength 0.5."""
    c = np.zeros(n, SPAT3D_CONFIG_DTYPE)
    c["unit_size"] = 10.0
    c["panning_strength"] = 1.0
    c["emission_angle"] = 45.0
    c["emission_angle_filter_attenuation_db"] = -12.0
    c["attenuation_filter_cutoff_hz"] = 5000.0
    c["attenuation_filter_db"] = -24.0
    c["doppler_speed_of_sound"] = 343.0
    c["global_panning_strength"] = 0.5
    for k, v in kw.items():
        c[k] = v
    return c


# every symbol include/gas_amd.h declares
EXPORTS = [
    "gas_abi_version",
    "gas_ctx_create",
    "gas_ctx_destroy",
    "gas_ctx_set_stream",
    "gas_ctx_synchronize",
    "gas_ctx_join_outputs",
    "gas_ctx_get_config",
    "gas_strerror",
    "gas_last_device_error",
    "gas_source_alloc",
    "gas_source_free",
    "gas_source_reset",
    "gas_source_set_draining",
    "gas_params_publish",
    "gas_fx_settings_publish",
    "gas_params_publish_batch",
    "gas_hrtf_load",
    "gas_hrtf_load_positions",
    "gas_calc_spatialization",
    "gas_calc_spatialization_areas",
    "gas_stream_create",
    "gas_stream_destroy",
    "gas_stream_get_info",
    "gas_stream_positions",
    "gas_stream_set_resampled",
    "gas_source_bind_stream",
    "gas_process_block_streams",
    "gas_process_block",
    "gas_bus_routes_publish",
    "gas_process_block_buses",
    "gas_process_frames_1",
    "gas_mix_channel_1",
    "gas_profile_enable",
    "gas_profile_read",
    "gas_bandwidth_probe",
    "gas_ctx_set_batch_depth",
    "gas_ctx_read_hrtf_order",
    "gas_tune_uni12_min",
    "gas_tune_nt_hist_min",
]


class GasError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        name = STATUS.get(status, str(status))
        super().__init__(f"{where}: {name}" + (f" ({detail})" if detail else ""))


_lib = None


def library_path():
    # GAS_AMD_LIB: development override for A/B-ing kernel builds (tools/README.md); never a fallback
    return os.environ.get("GAS_AMD_LIB") or _build.LIB


def load_library():
    """Load libgas_amd.so; raises if it was never built (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The spatializer has no CPU fallback."
        )
    # PyTorch-ROCm wheels bundle their own libamdhip64; two HIP runtimes in one process leave the
    # second one without devices.  Import torch first (when present) so libgas_amd.so binds to the
    # runtime torch already loaded (same soname) and streams/pointers can be shared with it.
    if os.environ.get("GAS_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(path)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    L.gas_abi_version.restype = i32
    L.gas_ctx_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.gas_ctx_destroy.argtypes = [vp]
    L.gas_ctx_destroy.restype = None
    L.gas_ctx_set_stream.argtypes = [vp, vp]
    L.gas_ctx_synchronize.argtypes = [vp]
    L.gas_ctx_join_outputs.argtypes = [vp]
    L.gas_ctx_get_config.argtypes = [vp, C.POINTER(Config)]
    L.gas_strerror.argtypes = [i32]
    L.gas_strerror.restype = C.c_char_p
    L.gas_last_device_error.argtypes = [vp]
    L.gas_last_device_error.restype = C.c_char_p
    L.gas_source_alloc.argtypes = [vp, i32, C.POINTER(C.c_int32), u32, C.POINTER(u32)]
    L.gas_source_free.argtypes = [vp, u32]
    L.gas_source_reset.argtypes = [vp, u32]
    L.gas_source_set_draining.argtypes = [vp, u32, i32]
    L.gas_params_publish.argtypes = [vp, u32, vp]
    L.gas_fx_settings_publish.argtypes = [vp, vp, vp, u32]
    L.gas_params_publish_batch.argtypes = [vp, vp, vp, u32, i32]
    L.gas_hrtf_load.argtypes = [vp, vp, u32, u32]
    L.gas_hrtf_load_positions.argtypes = [vp, vp, vp, u32, u32, u32, u32, i32, vp]
    L.gas_stream_create.argtypes = [vp, vp, i32, u32, C.c_uint64, C.POINTER(u32)]
    L.gas_stream_destroy.argtypes = [vp, u32]
    L.gas_stream_positions.argtypes = [vp, u32, vp]
    L.gas_stream_get_info.argtypes = [vp, u32, C.POINTER(C.c_uint64), C.POINTER(u32), C.POINTER(i32)]
    L.gas_stream_set_resampled.argtypes = [vp, u32, i32]
    L.gas_source_bind_stream.argtypes = [vp, u32, u32, C.c_uint64]
    L.gas_process_block_streams.argtypes = [vp, vp, u32, u32, vp, vp, vp, i32]
    L.gas_calc_spatialization.argtypes = [vp, vp, u32, vp, vp, vp, u32, vp, u32, vp, i32]
    L.gas_calc_spatialization_areas.argtypes = [vp, vp, u32, vp, vp, vp, u32, vp, u32, vp, vp, vp, vp, i32]
    L.gas_process_block.argtypes = [vp, vp, vp, u32, u32, vp, vp, i32]
    L.gas_bus_routes_publish.argtypes = [vp, vp, vp, u32]
    L.gas_process_block_buses.argtypes = [vp, vp, vp, u32, u32, vp, u32, vp, i32]
    L.gas_process_frames_1.argtypes = [vp, u32, vp, vp, i32]
    L.gas_mix_channel_1.argtypes = [vp, u32, i32, vp, vp, i32]
    L.gas_profile_enable.argtypes = [vp, i32]
    L.gas_profile_read.argtypes = [vp, C.POINTER(Profile), i32]
    L.gas_bandwidth_probe.argtypes = [vp, C.c_uint64, C.c_uint64, u32, u32, u32, C.POINTER(C.c_double)]
    L.gas_ctx_read_hrtf_order.argtypes = [vp, vp, u32]
    L.gas_tune_uni12_min.argtypes = [u32]
    L.gas_tune_uni12_min.restype = u32
    L.gas_tune_nt_hist_min.argtypes = [u32]
    L.gas_tune_nt_hist_min.restype = u32
    L.gas_ctx_set_batch_depth.argtypes = [vp, u32]
    _lib = L
    return L


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class SpatializerContext:
    """Thin object wrapper over one gas_ctx (one GPU)."""

    def __init__(self, max_sources, frames=512, channel_count=1, mix_rate=48000.0, er_ring_frames=0, device=0, flags=0):
        self.lib = load_library()
        self.frames = int(frames)
        self.channel_count = int(channel_count)
        self.max_sources = int(max_sources)
        cfg = Config(C.sizeof(Config), device, max_sources, frames, channel_count, mix_rate, er_ring_frames, flags)
        h = C.c_void_p()
        rc = self.lib.gas_ctx_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise GasError(rc, "gas_ctx_create", self.lib.gas_strerror(rc).decode())
        self.h = h

    def _check(self, rc, where):
        if rc != 0:
            detail = self.lib.gas_strerror(rc).decode()
            if rc == -7:
                detail += ": " + self.lib.gas_last_device_error(self.h).decode()
            raise GasError(rc, where, detail)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gas_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- slots ----
    def source_alloc(self, kind, effects=()):
        fx = (C.c_int32 * max(1, len(effects)))(*effects)
        slot = C.c_uint32()
        self._check(self.lib.gas_source_alloc(self.h, kind, fx, len(effects), C.byref(slot)), "gas_source_alloc")
        return slot.value

    def source_alloc_many(self, n, kind, effects=()):
        return np.array([self.source_alloc(kind, effects) for _ in range(n)], dtype=np.uint32)

    def source_free(self, slot):
        self._check(self.lib.gas_source_free(self.h, int(slot)), "gas_source_free")

    def source_set_draining(self, slot, draining=True):
        self._check(self.lib.gas_source_set_draining(self.h, int(slot), int(draining)), "gas_source_set_draining")

    def source_reset(self, slot):
        self._check(self.lib.gas_source_reset(self.h, int(slot)), "gas_source_reset")

    # ---- parameters ----
    def params_publish(self, slot, params):
        p = np.ascontiguousarray(params, dtype=PARAMS_DTYPE).reshape(1)
        self._check(self.lib.gas_params_publish(self.h, int(slot), _np_ptr(p)), "gas_params_publish")

    @staticmethod
    def fx_settings_defaults(n):
        d = np.zeros(n, FX_SETTINGS_DTYPE)
        d["filter_cutoff_hz"], d["filter_resonance"], d["filter_gain"] = 2000.0, 0.5, 1.0
        return d

    def fx_settings_publish(self, slots, settings):
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        f = np.ascontiguousarray(settings, dtype=FX_SETTINGS_DTYPE)
        assert s.shape == f.shape
        self._check(self.lib.gas_fx_settings_publish(self.h, _np_ptr(s), _np_ptr(f), len(s)), "gas_fx_settings_publish")

    def params_publish_batch(self, slots, params):
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        p = np.ascontiguousarray(params, dtype=PARAMS_DTYPE)
        assert s.shape == p.shape
        self._check(self.lib.gas_params_publish_batch(self.h, _np_ptr(s), _np_ptr(p), len(s), MEM_HOST), "gas_params_publish_batch")

    def params_publish_device(self, params_dev_ptr, n, slots=None):
        sp = None
        if slots is not None:
            s = np.ascontiguousarray(slots, dtype=np.uint32)
            sp = _np_ptr(s)
        self._check(self.lib.gas_params_publish_batch(self.h, sp, C.c_void_p(params_dev_ptr), n, MEM_DEVICE), "gas_params_publish_batch")

    def calc_spatialization(self, cfgs, poses, listeners, slots, cfg_index=None, want_params=True):
        """Host-array form of gas_calc_spatialization; returns the generated gas_params rows (or None)."""
        cfgs = np.ascontiguousarray(cfgs, dtype=SPAT3D_CONFIG_DTYPE)
        poses = np.ascontiguousarray(poses, dtype=POSE_DTYPE)
        listeners = np.ascontiguousarray(listeners, dtype=LISTENER_DTYPE)
        slots = np.ascontiguousarray(slots, dtype=np.uint32)
        ci = np.ascontiguousarray(cfg_index, dtype=np.uint32) if cfg_index is not None else None
        out = np.zeros(len(slots), PARAMS_DTYPE) if want_params else None
        rc = self.lib.gas_calc_spatialization(self.h, _np_ptr(cfgs), len(cfgs), _np_ptr(ci) if ci is not None else None, _np_ptr(poses), _np_ptr(listeners), len(listeners), _np_ptr(slots), len(slots), _np_ptr(out) if want_params else None, MEM_HOST)
        self._check(rc, "gas_calc_spatialization")
        return out

    def calc_spatialization_areas(self, cfgs, poses, listeners, slots, areas, listener_area_pos, cfg_index=None):
        """Host-array form of gas_calc_spatialization_areas; returns (gas_params rows, reverb volumes [n][4][2])."""
        cfgs = np.ascontiguousarray(cfgs, dtype=SPAT3D_CONFIG_DTYPE)
        poses = np.ascontiguousarray(poses, dtype=POSE_DTYPE)
        listeners = np.ascontiguousarray(listeners, dtype=LISTENER_DTYPE)
        slots = np.ascontiguousarray(slots, dtype=np.uint32)
        areas = np.ascontiguousarray(areas, dtype=AREA_SEND_DTYPE)
        lap = np.ascontiguousarray(listener_area_pos, dtype=np.float32) if listener_area_pos is not None else None
        ci = np.ascontiguousarray(cfg_index, dtype=np.uint32) if cfg_index is not None else None
        n = len(slots)
        assert len(areas) == n and (lap is None or lap.shape == (n, len(listeners), 3))
        out = np.zeros(n, PARAMS_DTYPE)
        reverb = np.full((n, 4, 2), np.nan, np.float32)
        rc = self.lib.gas_calc_spatialization_areas(self.h, _np_ptr(cfgs), len(cfgs), _np_ptr(ci) if ci is not None else None, _np_ptr(poses), _np_ptr(listeners), len(listeners), _np_ptr(slots), n, _np_ptr(areas), _np_ptr(lap) if lap is not None else None, _np_ptr(out), _np_ptr(reverb), MEM_HOST)
        self._check(rc, "gas_calc_spatialization_areas")
        return out, reverb

    # ---- device-resident streams ----
    def stream_create(self, pcm):
        """pcm: int16 or float32 array [frames] (mono) or [frames][2] (stereo)."""
        a = np.ascontiguousarray(pcm)
        fmt = 0 if a.dtype == np.int16 else 1
        if fmt == 1:
            a = a.astype(np.float32)
        ch = 1 if a.ndim == 1 else a.shape[1]
        sid = C.c_uint32()
        self._check(self.lib.gas_stream_create(self.h, _np_ptr(a), fmt, ch, a.shape[0], C.byref(sid)), "gas_stream_create")
        return sid.value

    def stream_set_resampled(self, sid, on=True):
        self._check(self.lib.gas_stream_set_resampled(self.h, sid, int(on)), "gas_stream_set_resampled")

    def stream_destroy(self, sid):
        self._check(self.lib.gas_stream_destroy(self.h, sid), "gas_stream_destroy")

    def source_bind_stream(self, slot, sid, start_frame=0):
        self._check(self.lib.gas_source_bind_stream(self.h, int(slot), int(sid), int(start_frame)), "gas_source_bind_stream")

    def process_block_streams(self, slots):
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        n = len(s)
        out = np.full((self.channel_count, self.frames, 2), np.nan, dtype=np.float32)
        peaks = np.zeros((max(n, 1), 2), dtype=np.float32)
        hf = np.zeros(max(n, 1), dtype=np.uint8)
        rc = self.lib.gas_process_block_streams(self.h, _np_ptr(s) if n else None, n, self.frames, _np_ptr(out), _np_ptr(peaks), _np_ptr(hf), MEM_HOST)
        self._check(rc, "gas_process_block_streams")
        return out, peaks[:n], hf[:n].astype(bool)

    def hrtf_load(self, hrir):
        h = np.ascontiguousarray(hrir, dtype=np.float32)
        assert h.ndim == 3 and h.shape[1] == 2
        self._check(self.lib.gas_hrtf_load(self.h, _np_ptr(h), h.shape[0], h.shape[2]), "gas_hrtf_load")

    def hrtf_load_positions(self, positions, hrir, az_steps, el_steps, interpolation=0):
        """positions [m][2] (azimuth, elevation) radians, hrir [m][2][taps]; returns the gridded set [dirs][2][256]."""
        pos = np.ascontiguousarray(positions, dtype=np.float32)
        h = np.ascontiguousarray(hrir, dtype=np.float32)
        assert pos.ndim == 2 and pos.shape[1] == 2 and h.ndim == 3 and h.shape[:2] == (pos.shape[0], 2)
        out = np.zeros((az_steps * el_steps, 2, HRTF_TAPS), np.float32)
        self._check(self.lib.gas_hrtf_load_positions(self.h, _np_ptr(pos), _np_ptr(h), pos.shape[0], h.shape[2], az_steps, el_steps, int(interpolation), _np_ptr(out)), "gas_hrtf_load_positions")
        return out

    # ---- hot path ----
    def process_block(self, src, slots):
        """Host-memory callback: src float32 [n][F][2], slots uint32 [n] -> (mix [C][F][2], peaks [n][2])."""
        src = np.ascontiguousarray(src, dtype=np.float32)
        n = src.shape[0] if src.ndim == 3 else 0
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        assert len(s) == n
        out = np.full((self.channel_count, self.frames, 2), np.nan, dtype=np.float32)
        peaks = np.zeros((max(n, 1), 2), dtype=np.float32)
        frames = src.shape[1] if src.ndim == 3 else self.frames
        rc = self.lib.gas_process_block(self.h, _np_ptr(src) if n else None, _np_ptr(s) if n else None, n, frames, _np_ptr(out), _np_ptr(peaks), MEM_HOST)
        self._check(rc, "gas_process_block")
        return out, peaks[:n]

    def bus_routes_publish(self, slots, routes):
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        r = np.ascontiguousarray(routes, dtype=BUS_ROUTE_DTYPE)
        assert s.shape == r.shape
        self._check(self.lib.gas_bus_routes_publish(self.h, _np_ptr(s), _np_ptr(r), len(s)), "gas_bus_routes_publish")

    def process_block_buses(self, src, slots, n_buses):
        """Host-memory callback over several buses: -> (mix [n_buses][C][F][2], peaks [n][2])."""
        src = np.ascontiguousarray(src, dtype=np.float32)
        n = src.shape[0] if src.ndim == 3 else 0
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        out = np.full((n_buses, self.channel_count, self.frames, 2), np.nan, dtype=np.float32)
        peaks = np.zeros((max(n, 1), 2), dtype=np.float32)
        rc = self.lib.gas_process_block_buses(self.h, _np_ptr(src) if n else None, _np_ptr(s) if n else None, n, self.frames, _np_ptr(out), n_buses, _np_ptr(peaks), MEM_HOST)
        self._check(rc, "gas_process_block_buses")
        return out, peaks[:n]

    def process_block_raw(self, src_ptr, slots, n, frames, out_ptr, peaks_ptr, mem):
        """Raw pointers (device or host). slots: numpy uint32 array or None (reuse the previous list)."""
        sp = None
        if slots is not None:
            s = np.ascontiguousarray(slots, dtype=np.uint32)
            sp = _np_ptr(s)
        return self.lib.gas_process_block(self.h, C.c_void_p(src_ptr), sp, n, frames, C.c_void_p(out_ptr), C.c_void_p(peaks_ptr) if peaks_ptr else None, mem)

    def process_frames_1(self, slot, src):
        src = np.ascontiguousarray(src, dtype=np.float32)
        out = np.full_like(src, np.nan)
        self._check(self.lib.gas_process_frames_1(self.h, int(slot), _np_ptr(out), _np_ptr(src), src.shape[0]), "gas_process_frames_1")
        return out

    def mix_channel_1(self, slot, channel, src):
        src = np.ascontiguousarray(src, dtype=np.float32)
        out = np.full_like(src, np.nan)
        self._check(self.lib.gas_mix_channel_1(self.h, int(slot), int(channel), _np_ptr(out), _np_ptr(src), src.shape[0]), "gas_mix_channel_1")
        return out

    def set_stream(self, hip_stream):
        self._check(self.lib.gas_ctx_set_stream(self.h, C.c_void_p(hip_stream)), "gas_ctx_set_stream")

    def synchronize(self):
        self._check(self.lib.gas_ctx_synchronize(self.h), "gas_ctx_synchronize")

    def join_outputs(self):
        """FLAG_PIPELINED_MIX: order the context's stream behind every `out` written so far (non-blocking)."""
        self._check(self.lib.gas_ctx_join_outputs(self.h), "gas_ctx_join_outputs")

    def profile_enable(self, on=1):
        self._check(self.lib.gas_profile_enable(self.h, int(on)), "gas_profile_enable")

    def bandwidth_probe(self, read_bytes, write_bytes, workgroups=2048, unroll=4, iters=50):
        """Span (us) of one pure streaming launch over the given byte counts, timed like the dominant kernel."""
        us = C.c_double()
        self._check(self.lib.gas_bandwidth_probe(self.h, int(read_bytes), int(write_bytes), int(workgroups), int(unroll), int(iters), C.byref(us)), "gas_bandwidth_probe")
        return us.value

    def set_batch_depth(self, depth):
        """FLAG_BATCHED_LAUNCH: callbacks per k_hrtf_multi launch (1 .. 16)."""
        self._check(self.lib.gas_ctx_set_batch_depth(self.h, int(depth)), "gas_ctx_set_batch_depth")

    def read_hrtf_order(self, n):
        """Diagnostic: the XCD-affine processing order of the last callback's plain [HRTF] sources."""
        out = np.zeros(int(n), np.uint32)
        self._check(self.lib.gas_ctx_read_hrtf_order(self.h, out.ctypes.data, int(n)), "gas_ctx_read_hrtf_order")
        return out

    def profile_read(self, reset=True):
        p = Profile()
        self._check(self.lib.gas_profile_read(self.h, C.byref(p), int(reset)), "gas_profile_read")
        return {"launches": p.launches, "kernel_ms": p.kernel_ms, "bytes_per_launch": p.bytes_per_launch, "kernel": p.kernel_name.decode(), "callbacks_per_launch": p.callbacks_per_launch, "bytes_per_callback_formula": p.bytes_per_callback_formula}


class BatchedSpatializerHost:
    """ctypes view of include/gas_amd_host.h: the CPU-side half of AudioSpatializerInstance
    (source window, fade-out, latch, silence gate, list GC) over one gas_process_block per callback."""

    def __init__(self, ctx, kind, effects=()):
        self.ctx = ctx
        L = self.lib = ctx.lib
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
        L.gas_host_create.argtypes = [vp, i32, C.POINTER(C.c_int32), u32, C.POINTER(vp)]
        L.gas_host_destroy.argtypes = [vp]
        L.gas_host_destroy.restype = None
        L.gas_host_start_playback_array.argtypes = [vp, vp, C.c_int64, C.POINTER(u32)]
        L.gas_host_stop_playback.argtypes = [vp, u32]
        L.gas_host_start_playback_device_stream.argtypes = [vp, u32, C.c_uint64, C.POINTER(u32)]
        L.gas_host_set_spatializer_parameters.argtypes = [vp, u32, vp]
        L.gas_host_set_playback_disable_threshold_db.argtypes = [vp, C.c_float]
        L.gas_host_set_playback_disable_threshold_db.restype = None
        L.gas_host_is_playback_active.argtypes = [vp, u32]
        L.gas_host_playback_count.argtypes = [vp]
        L.gas_host_set_playback_paused.argtypes = [vp, u32, i32]
        L.gas_host_is_playback_paused.argtypes = [vp, u32]
        L.gas_host_get_playback_position.argtypes = [vp, u32, C.POINTER(C.c_uint64)]
        L.gas_host_get_mixed_frames.argtypes = [vp, i32, vp, i32]
        L.gas_host_set_effect_settings.argtypes = [vp, u32, vp]
        L.gas_host_set_release_fn.argtypes = [vp, vp, vp]
        L.gas_host_collect_released.argtypes = [vp]
        L.gas_host_set_process_effects_fn.argtypes = [vp, vp, vp]
        L.gas_host_start_playback.argtypes = [vp, vp, vp, C.POINTER(u32)]
        self._callbacks = []  # ctypes trampolines must outlive their registration
        fx = (C.c_int32 * max(1, len(effects)))(*effects)
        h = C.c_void_p()
        ctx._check(L.gas_host_create(ctx.h, kind, fx, len(effects), C.byref(h)), "gas_host_create")
        self.h = h
        self._streams = []

    def close(self):
        if self.h:
            self.lib.gas_host_destroy(self.h)
            self.h = None

    def start_playback_array(self, stream):
        s = np.ascontiguousarray(stream, dtype=np.float32)
        self._streams.append(s)  # the host reads it on every callback: keep it alive
        pid = C.c_uint32()
        self.ctx._check(self.lib.gas_host_start_playback_array(self.h, _np_ptr(s), s.shape[0], C.byref(pid)), "gas_host_start_playback_array")
        return pid.value

    def start_playback_device_stream(self, stream_id, start_frame=0):
        pid = C.c_uint32()
        self.ctx._check(self.lib.gas_host_start_playback_device_stream(self.h, int(stream_id), int(start_frame), C.byref(pid)), "gas_host_start_playback_device_stream")
        return pid.value

    STREAM_MIX_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int)
    RELEASE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_void_p)
    PROCESS_EFFECTS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p)

    def start_playback(self, mix, user=0):
        """mix(buffer: float32 [frames, 2] view, rate_scale, frames) -> frames mixed: the engine's
        AudioStreamPlayback::mix as a Python callable (tests only; it runs on whichever thread mixes)."""

        def tramp(u, buf, rate, frames):
            view = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_float)), shape=(frames, 2))
            return int(mix(view, rate, frames))

        cb = self.STREAM_MIX_FN(tramp)
        self._callbacks.append(cb)
        pid = C.c_uint32()
        self.ctx._check(self.lib.gas_host_start_playback(self.h, C.cast(cb, C.c_void_p), C.c_void_p(user), C.byref(pid)), "gas_host_start_playback")
        return pid.value

    def set_release_fn(self, fn):
        """fn(id, user) for every playback the host has finished with (control thread)."""
        cb = self.RELEASE_FN(lambda _u, pid, user: fn(pid, user or 0)) if fn else None
        self._callbacks.append(cb)
        return self.lib.gas_host_set_release_fn(self.h, C.cast(cb, C.c_void_p) if cb else None, None)

    def collect_released(self):
        return self.lib.gas_host_collect_released(self.h)

    def set_process_effects_fn(self, fn):
        """fn(id, params: one-element PARAMS_DTYPE view) -> truthy when it edited the row (audio thread)."""

        def tramp(_u, pid, p):
            row = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(PARAMS_DTYPE.itemsize,)).view(PARAMS_DTYPE)
            return int(bool(fn(pid, row)))

        cb = self.PROCESS_EFFECTS_FN(tramp) if fn else None
        self._callbacks.append(cb)
        return self.lib.gas_host_set_process_effects_fn(self.h, C.cast(cb, C.c_void_p) if cb else None, None)

    def stop_playback(self, pid):
        return self.lib.gas_host_stop_playback(self.h, pid)

    def set_spatializer_parameters(self, pid, params):
        """Returns GAS_OK, or GAS_ERR_BAD_SLOT when the playback has ended and been reaped meanwhile (not an error for a
        control thread: the reference's update_spatializer_parameters has nothing to address then either)."""
        p = np.ascontiguousarray(params, dtype=PARAMS_DTYPE).reshape(1)
        rc = self.lib.gas_host_set_spatializer_parameters(self.h, pid, _np_ptr(p))
        if rc != -3:
            self.ctx._check(rc, "gas_host_set_spatializer_parameters")
        return rc

    def set_effect_settings(self, pid, settings):
        f = np.ascontiguousarray(settings, dtype=FX_SETTINGS_DTYPE).reshape(1)
        return self.lib.gas_host_set_effect_settings(self.h, pid, _np_ptr(f))

    def is_playback_active(self, pid):
        return bool(self.lib.gas_host_is_playback_active(self.h, pid))

    def playback_count(self):
        return self.lib.gas_host_playback_count(self.h)

    def set_playback_paused(self, pid, paused=True):
        return self.lib.gas_host_set_playback_paused(self.h, pid, int(paused))

    def is_playback_paused(self, pid):
        return bool(self.lib.gas_host_is_playback_paused(self.h, pid))

    def get_playback_position(self, pid):
        """Frames of the stream consumed so far (0 for an unknown id)."""
        f = C.c_uint64()
        self.lib.gas_host_get_playback_position(self.h, pid, C.byref(f))
        return f.value

    def get_mixed_frames(self, channel, frame_count):
        out = np.full((frame_count, 2), np.nan, dtype=np.float32)
        rc = self.lib.gas_host_get_mixed_frames(self.h, channel, _np_ptr(out), frame_count)
        return rc, out


class MultiContext:
    """ctypes view of gas_multi_* (include/gas_amd_host.h): several per-device contexts in one process."""

    def __init__(self, devices, max_sources, frames=512, channel_count=1, mix_rate=48000.0, er_ring_frames=0, flags=0):
        L = self.lib = load_library()
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
        L.gas_multi_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_int32), u32, C.POINTER(vp)]
        L.gas_multi_destroy.argtypes = [vp]
        L.gas_multi_destroy.restype = None
        L.gas_multi_shards.argtypes = [vp]
        L.gas_multi_shards.restype = u32
        L.gas_multi_shard.argtypes = [vp, u32]
        L.gas_multi_shard.restype = vp
        L.gas_multi_least_loaded.argtypes = [vp]
        L.gas_multi_least_loaded.restype = u32
        L.gas_multi_note_alloc.argtypes = [vp, u32, i32]
        L.gas_multi_note_alloc.restype = None
        L.gas_multi_process_block.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u32), u32, vp, C.POINTER(vp)]
        L.gas_multi_process_block_mem.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u32), u32, vp, C.POINTER(vp), i32]
        L.gas_multi_synchronize.argtypes = [vp]
        L.gas_multi_root_stream.argtypes = [vp]
        L.gas_multi_root_stream.restype = vp
        self.frames, self.channel_count = frames, channel_count
        cfg = Config(C.sizeof(Config), 0, max_sources, frames, channel_count, mix_rate, er_ring_frames, flags)
        dev = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        rc = L.gas_multi_create(C.byref(cfg), dev, len(devices), C.byref(h))
        if rc != 0:
            raise GasError(rc, "gas_multi_create", L.gas_strerror(rc).decode())
        self.h = h
        self.shards = []
        for g in range(len(devices)):
            ctx = SpatializerContext.__new__(SpatializerContext)  # a view: the multi-context owns the handle
            ctx.lib, ctx.frames, ctx.channel_count, ctx.max_sources = L, frames, channel_count, max_sources
            ctx.h = C.c_void_p(L.gas_multi_shard(h, g))
            ctx.close = lambda: None
            self.shards.append(ctx)

    def close(self):
        if self.h:
            for s in self.shards:
                s.h = None
            self.lib.gas_multi_destroy(self.h)
            self.h = None

    def process_block_device(self, src_ptrs, slots_per_shard, out_ptr, peaks_ptrs=None):
        """Device-memory callback: src_ptrs[g] / peaks_ptrs[g] are device addresses on shard g's device, out_ptr on the
        root device; only enqueues (gas_multi_synchronize / the root stream order the result)."""
        G = len(self.shards)
        sls = [np.ascontiguousarray(s, dtype=np.uint32) for s in slots_per_shard]
        vp = C.c_void_p
        a_src = (vp * G)(*src_ptrs)
        a_sl = (vp * G)(*[s.ctypes.data for s in sls])
        a_pk = (vp * G)(*(peaks_ptrs if peaks_ptrs is not None else [None] * G))
        a_n = (C.c_uint32 * G)(*[len(s) for s in sls])
        rc = self.lib.gas_multi_process_block_mem(self.h, a_src, a_sl, a_n, self.frames, C.c_void_p(out_ptr), a_pk, MEM_DEVICE)
        if rc != 0:
            raise GasError(rc, "gas_multi_process_block_mem", self.lib.gas_strerror(rc).decode())

    def synchronize(self):
        rc = self.lib.gas_multi_synchronize(self.h)
        if rc != 0:
            raise GasError(rc, "gas_multi_synchronize", self.lib.gas_strerror(rc).decode())

    def process_block(self, src_per_shard, slots_per_shard):
        G = len(self.shards)
        srcs = [np.ascontiguousarray(s, dtype=np.float32) for s in src_per_shard]
        sls = [np.ascontiguousarray(s, dtype=np.uint32) for s in slots_per_shard]
        pks = [np.zeros((max(len(s), 1), 2), np.float32) for s in sls]
        vp = C.c_void_p
        a_src = (vp * G)(*[s.ctypes.data for s in srcs])
        a_sl = (vp * G)(*[s.ctypes.data for s in sls])
        a_pk = (vp * G)(*[p.ctypes.data for p in pks])
        a_n = (C.c_uint32 * G)(*[len(s) for s in sls])
        out = np.full((self.channel_count, self.frames, 2), np.nan, np.float32)
        rc = self.lib.gas_multi_process_block(self.h, a_src, a_sl, a_n, self.frames, _np_ptr(out), a_pk)
        if rc != 0:
            raise GasError(rc, "gas_multi_process_block", self.lib.gas_strerror(rc).decode())
        return out, [p[: len(s)] for p, s in zip(pks, sls)]
