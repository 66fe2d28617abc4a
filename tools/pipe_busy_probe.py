#!/usr/bin/env python3
"""Development aid: busy cycles per role of k_biquad_pipe's software pipeline (needs a -DGAS_STAMPS build via GAS_AMD_LIB)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import godot_audio_spatializer_amd as gas  # noqa: E402
from godot_audio_spatializer_amd import synth  # noqa: E402

lib = gas.load_library()
K = gas.capi
n, F = 256, 512
rng = np.random.default_rng(0)
with gas.SpatializerContext(max_sources=n, frames=F) as ctx:
    slots = ctx.source_alloc_many(n, K.KIND_3D_MIX)
    for _ in range(4):
        ctx.params_publish_batch(slots, synth.draw_params(rng, n))
        ctx.process_block(synth.draw_sources(rng, n, F), slots)
    buf = np.zeros(16, np.uint64)
    assert lib.gas_debug_read_pipe_busy(buf.ctypes.data_as(C.c_void_p)) == 0
    names = ["REC", "COEF", "FIR0", "FIR1", "FIR2", "FIR3", "LOAD", "POST"]
    for i, nm in enumerate(names):
        print(f"{nm:5s} busy {int(buf[2 * i]):8d} cycles of {int(buf[2 * i + 1]):8d} in the loop ({int(buf[2*i])/19/32:.1f} per frame-step)")
