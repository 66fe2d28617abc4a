"""MI355X-native many-source spatializer: the one hot path of godot-audio-spatializer
(AudioSpatializerInstance::_mix_from_playback_list and the plugin DSP it dispatches to),
as hand-written HIP kernels behind the C ABI of include/gas_amd.h.

Python here is harness-side plumbing only (ctypes binding, host-side mirror of the plugin
interface, multi-GPU sharding glue); the product is libgas_amd.so.
"""
from . import build  # noqa: F401
from . import capi  # noqa: F401
from .capi import (  # noqa: F401
    PARAMS_DTYPE,
    GasError,
    SpatializerContext,
    load_library,
)
