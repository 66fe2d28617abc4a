#!/usr/bin/env python3
"""bench.py -- headline benchmark: mixed AudioFrames/s of the batched spatializer hot path.

One "step" = one audio callback: every active source's F-frame AudioFrame buffer (already resident
in HBM) -> per-source spatialization -> N-source stereo mix, through the C ABI (gas_process_block).
Default workload = BASELINE.json configs[3]'s per-GPU shard (8192 HRTF sources per GPU, 256-tap
overlap-save, F = 512 @ 48 kHz); at N GPUs the job is 8192*N sources (configs[3] itself at N = 8),
weak scaling, one process per GPU, partial mixes sum-reduced to rank 0 over RCCL.  `--workload`
selects the other configs for ad-hoc runs.

Prints ONE JSON line on rank 0 (contract in the task statement).  How the numbers are taken:

* `value` / `ms_per_step`: EXACTLY --steps callbacks, queued back to back after --warmup untimed ones,
  bracketed by barrier + torch.cuda.synchronize() on both sides; the time is the span between ONE pair of
  HIP events recorded on the launch stream around those callbacks (GPU timeline: no host sync latency
  inside, so 20 steps read the same as 200), MAX over ranks.  The host wall clock around the same
  region is reported beside it (`wall_ms_per_step`).  No event markers sit inside this pass.
* This is the library's THROUGHPUT mode (callbacks queued back to back: GAS_FLAG_PIPELINED_MIX sums
  callback t's partial mixes inside callback t+1's launch; GAS_FLAG_PEAKS_DRAINING_ONLY measures peaks
  only where the reference reads them).  The contract-faithful figures are in the same line:
  `ordered` (synchronous two-dispatch callback, what an audio thread runs), `exact_peaks` (ordered,
  every source's peak), `latency` (one synchronous callback at a time, host-timed p50 / p99).
* `roofline`: a SEPARATE short pass brackets the dominant launch of every callback with HIP events
  inside the library (gas_profile_*); `copy_peak` is a pure streaming launch over the same byte count,
  timed the same way in the same run (gas_bandwidth_probe), i.e. what the memory system gives a
  launch of this size.
* `cpu_baseline`: the oracle's reference-equivalent scalar path, 1 core, bounded sample.

HRTF / early-reflection workloads have no reference counterpart: parity for them is against this
repository's own oracle ("parity unpinned").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (kind, chain, frames, default sources per GPU, er ring, description)
    "hrtf": (2, (3,), 512, 8192, 0, "cfg4 shard: 8192 sources/GPU, 256-tap HRTF overlap-save FFT -> stereo, 512-frame @48kHz"),
    "hrtf4096": (2, (3,), 512, 4096, 0, "cfg3: 4096 sources, 256-tap HRTF overlap-save FFT -> stereo, 512-frame @48kHz"),
    "biquad": (0, (), 512, 256, 0, "cfg2: 256 sources, pan + distance high-shelf biquad (mix_channel), 512-frame @48kHz"),
    "erhrtf": (2, (2, 3), 256, 4096, 4096, "cfg5: 4096 sources, 8-tap early reflections + HRTF chain, 256-frame @48kHz"),
    "hrtf4096_f256": (2, (3,), 256, 4096, 0, "comparison for cfg5 (not a BASELINE config): 4096 sources, HRTF alone, 256-frame @48kHz"),
}
HBM_PEAK = 8.0e12  # MI355X_MICROARCH.md: 8 TB/s spec
N_SRC_BUFFERS_BYTES = int(os.environ.get("GAS_BENCH_SRC_BYTES", 320 << 20))  # rotate source buffers over > 256 MiB so the Infinity Cache cannot hold them (the override is a cache experiment, never the headline)
CONDITION_STEPS = 64  # untimed callbacks in front of the --warmup ones: clocks and caches settle independently of --warmup


def lib_sha16():
    """First 16 hex digits of the sha256 of the library this run loads: PMC records carry it, so a record taken with
    another build of the kernels is refused instead of silently going stale."""
    import hashlib

    from godot_audio_spatializer_amd import capi

    try:
        return hashlib.sha256(open(capi.library_path(), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def pmc_traffic(kernel, workload, n_local, peaks, pipelined, experiment=""):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/*_pmc.json,
    written by tools/profile_bench.sh; FETCH_SIZE x2 + WRITE_SIZE), or None.  The newest
    matching record wins (files sort by round); a record taken with a different build of the library (lib_sha16) only
    matches when GAS_BENCH_ACCEPT_STALE_PMC is set, and is then labelled stale."""
    import glob

    best = None
    sha = lib_sha16()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("kernel") == kernel and rec.get("workload") == workload and rec.get("sources_per_gpu") == n_local and rec.get("peaks") == peaks and bool(rec.get("pipelined_mix", False)) == pipelined and rec.get("experiment", "") == experiment:
            fresh = rec.get("lib_sha16") is not None and rec.get("lib_sha16") == sha
            if fresh or os.environ.get("GAS_BENCH_ACCEPT_STALE_PMC"):
                best = (rec["traffic_bytes_per_launch"], os.path.basename(path) + ("" if fresh else " [STALE: taken with another build of the library]"), rec)
    return best


# Vector-ALU floor of the HRTF kernels: wave-level VALU instructions one source costs (static census of the code object,
# tools/isa_blocks.py -> profiles/r03_isa_*.txt; cross-checked with SQ_INSTS_VALU per launch, profiles/r03_notes.md) x
# the issue rate one SIMD sustains for them with two waves resident (tools/micro/valurate.hip: one wave64 f32 VALU
# instruction per ~4 shader cycles per SIMD at 2 waves/SIMD -- not the 2 cycles of the SIMD-32 data path) over the
# chip's 1024 SIMDs at the 2.4 GHz peak clock.  A floor, like the HBM floor beside it: nothing overlaps perfectly.
VALU_PER_SOURCE = {"k_hrtf_multi": 384, "k_hrtf_uni": 388}
VALU_CYCLES_PER_INST = 4.0
SIMDS = 1024
CLOCK_HZ = 2.4e9


def roofline_entry(prof, n_sources, desc=None, peaks=None, pipelined=False, experiment=""):
    """The `roofline` object of one timed pass (contract: achieved = bytes the launch must move / its average span)."""
    launches = max(prof["launches"], 1)
    k_us = prof["kernel_ms"] / launches * 1e3
    B = prof["bytes_per_launch"]
    K = max(1, prof.get("callbacks_per_launch", 1))
    achieved = B / (k_us * 1e-6) if k_us > 0 else 0.0
    r = {
        "kernel": prof["kernel"],
        "kernel_us": k_us,
        "launches_timed": prof["launches"],
        "callbacks_per_launch": K,
        "algorithmic_bytes_per_launch": B,
        "achieved": achieved / 1e9,
        "peak": HBM_PEAK / 1e9,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK,
        "hbm_floor_us": B / HBM_PEAK * 1e6,
        "traffic": None,
        "traffic_source": None,
    }
    if K > 1:
        Bf = prof.get("bytes_per_callback_formula", B)
        r["per_callback_formula"] = {"bytes_per_launch": Bf, "frac": (Bf / (k_us * 1e-6) / HBM_PEAK) if k_us > 0 else 0.0, "what": "SURVEY.md 8d's per-callback bytes x callbacks per launch (round 2's figure): counts the history rows and the HRIR table once per callback although the batched kernel moves them once per launch -- kept for comparison, not a roofline"}
    vps = VALU_PER_SOURCE.get(prof["kernel"])
    if vps:
        r["valu_floor_us"] = vps * VALU_CYCLES_PER_INST * n_sources * K / SIMDS / CLOCK_HZ * 1e6
        r["valu_per_source"] = vps
        r["bound"] = "valu" if r["valu_floor_us"] > r["hbm_floor_us"] else "hbm"
        r["bound_evidence"] = "the higher of the two floors (VALU wave-instructions per source x 4 cycles / (1024 SIMDs x 2.4 GHz) vs bytes / 8 TB/s); SQ counters of the same launch in profiles/ (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, SQ_WAIT_ANY)"
    else:
        r["bound"] = "hbm" if n_sources >= 4096 else "latency"
    if desc is not None:
        t = pmc_traffic(prof["kernel"], desc, n_sources, peaks, pipelined, experiment)
        if t:
            r["traffic"] = float(t[0])
            r["traffic_source"] = "profiles/" + t[1] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; FETCH x2 on gfx950)"
            r["traffic_over_algorithmic"] = float(t[0]) / B if B else None
    return r


def _cpu_worker(args):
    """One worker of the all-cores baseline: its shard of the sources, `blocks` callbacks; returns seconds."""
    kind, chain, frames, dirs, hrir, ring, n, blocks, seed = args
    from oracle import binding as ob
    from godot_audio_spatializer_amd import synth

    rng = np.random.default_rng(seed)
    ora = ob.BatchOracle(kind, n, frames, chain=chain, hrir=hrir, er_ring_frames=max(ring, 1), hrtf_impl=1)
    p = synth.draw_params(rng, n, dirs=dirs, ring_frames=max(ring, 2 * frames), frames=frames).astype(ob.PARAMS_DTYPE)
    src = synth.draw_sources(rng, n, frames)
    ora.block(p, src)  # warm-up (state, caches)
    t0 = time.perf_counter()
    for _ in range(blocks):
        ora.block(p, src)
    return time.perf_counter() - t0


def cpu_baseline(kind, chain, frames, dirs, hrir, ring, budget_s=10.0):
    """Reference-equivalent CPU path (oracle, scalar f32) on a bounded sample of the workload: 1 core (the faithful
    figure: Godot mixes on one audio thread), plus the same sample sharded over every host core."""
    n = 512 if 3 in chain else 2048
    # size the sample to ~budget_s of single-core work
    t_probe = _cpu_worker((kind, chain, frames, dirs, hrir, ring, n, 2, 1234)) / 2
    blocks = int(max(4, min(256, budget_s / max(t_probe, 1e-6))))
    dt = _cpu_worker((kind, chain, frames, dirs, hrir, ring, n, blocks, 1234))
    out = {
        "value": n * frames * blocks / dt,
        "unit": "AudioFrames/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n} sources x {blocks} callbacks of the same workload, oracle scalar f32 path (HRTF by radix-2 overlap-save), 1 thread",
    }
    try:
        # all host cores: independent child processes of this script in --cpu-worker mode (they never touch the GPU),
        # each on its shard of the sources, hard timeout so the headline line can never hang on them
        import subprocess
        import tempfile

        cores = min(len(os.sched_getaffinity(0)), 16)  # a one-GPU box's CPU share, not every visible CPU
        if cores > 1:
            per = max(1, n // cores)
            with tempfile.TemporaryDirectory() as td:
                hp = os.path.join(td, "hrir.npy")
                np.save(hp, hrir if hrir is not None else np.zeros((1, 2, 256), np.float32))
                cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", json.dumps([kind, list(chain), frames, dirs, hp if hrir is not None else "", ring, per, blocks])]
                procs = [subprocess.Popen(cmd + [str(1234 + i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(cores)]
                times = []
                deadline = time.time() + 6 * budget_s + 60
                for pr in procs:
                    try:
                        o, _ = pr.communicate(timeout=max(1.0, deadline - time.time()))
                        times.append(float(o.strip().splitlines()[-1]))
                    except Exception:
                        pr.kill()
                        raise
            out["all_cores"] = {"value": per * cores * frames * blocks / max(times), "cores": cores, "sample": f"{per} sources per process x {cores} processes x {blocks} callbacks"}
    except Exception as e:  # the extra figure must never cost the headline line
        out["all_cores"] = {"error": repr(e)}
    return out


class Runner:
    """One context + the bench's callback loop over it: parameters republished every second callback (an emulated
    60 Hz physics tick), source buffers rotated over more than the Infinity Cache, mixes landing in buckets that are
    sum-reduced to rank 0 when more than one GPU takes part."""

    def __init__(self, env, flags, n_local, bucket):
        gas, torch, synth, sharding = env["gas"], env["torch"], env["synth"], env["sharding"]
        args, kind, chain, frames, ring = env["args"], env["kind"], env["chain"], env["frames"], env["ring"]
        self.env, self.gas, self.torch = env, gas, torch
        self.n_local, self.frames, self.flags = n_local, frames, flags
        self.pipelined = bool(flags & gas.capi.FLAG_PIPELINED_MIX)
        self.paired = self.pipelined and bool(flags & gas.capi.FLAG_BATCHED_LAUNCH)
        self.depth = max(1, min(16, args.batch_depth)) if self.paired else 1
        self.ctx = gas.SpatializerContext(max_sources=n_local, frames=frames, channel_count=1, er_ring_frames=ring, device=env["local_rank"], flags=flags)
        if self.paired:
            self.ctx.set_batch_depth(self.depth)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        if env["hrir"] is not None:
            self.ctx.hrtf_load(env["hrir"])
        self.slots = self.ctx.source_alloc_many(n_local, kind, chain)
        self.n_draining = 0
        if kind == 2 and args.draining_every > 0:
            for s_ in self.slots[:: args.draining_every]:
                self.ctx.source_set_draining(int(s_), True)
                self.n_draining += 1
        # two physics ticks of parameters, device-resident, alternated every 2 callbacks
        prng = np.random.default_rng(1234 + 7919 * env["rank"])
        self.psets = []
        for _ in range(2):
            p = synth.draw_params(prng, n_local, dirs=args.dirs, ring_frames=max(ring, 2 * frames), frames=frames)
            if args.presorted_directions:
                p["hrtf_dir"] = np.sort(p["hrtf_dir"])
            if args.xcd_directions:
                per_wg = max(1, n_local // 256)
                p["hrtf_dir"] = (prng.integers(0, args.dirs // 8, n_local) * 8 + (np.arange(n_local) // per_wg) % 8).astype(np.uint32)
            self.psets.append(torch.from_numpy(p.view(np.uint8).reshape(n_local, 128).copy()).cuda())
        self.ctx.params_publish_batch(self.slots, synth.draw_params(prng, n_local, dirs=args.dirs, ring_frames=max(ring, 2 * frames), frames=frames))
        # rotating source buffers: synthetic uniform(-0.5, 0.5) AudioFrames, footprint > Infinity Cache; shared by the
        # runners of one process (the same frames feed every mode)
        key = (n_local, frames)
        if key not in env["srcs"]:
            buf_bytes = n_local * frames * 8
            n_bufs = max(2, min(16, -(-N_SRC_BUFFERS_BYTES // buf_bytes)))
            gen = torch.Generator(device="cuda")
            gen.manual_seed(1234 + env["rank"])
            env["srcs"][key] = [torch.rand(n_local, frames, 2, device="cuda", generator=gen) - 0.5 for _ in range(n_bufs)]
        self.srcs = env["srcs"][key]
        self.n_bufs = len(self.srcs)
        # Partial mixes land in buckets of B callbacks; on N > 1 GPUs each full bucket is sum-reduced to rank 0 in ONE
        # collective (B x 4 KiB) on a side stream while the next bucket is being computed: the 4 KiB per-callback
        # message is latency-bound over xGMI.  That is a THROUGHPUT arrangement (callbacks queued back to back, the
        # mix of a callback reaches rank 0 up to B callbacks later); a real-time host uses --reduce-bucket 1.
        if self.depth > 1 and bucket >= self.depth:
            bucket = bucket // self.depth * self.depth  # whole batches per bucket: the carried sums line up with the reduces
        self.B = B = max(1, bucket)
        self.buckets = [torch.zeros(B, 1, frames, 2, device="cuda") for _ in range(2)]
        self.peaks = torch.zeros(n_local, 2, device="cuda")
        # buckets: the collective beside the compute stream (its ~50 us of host time are spread over the bucket); one or two
        # callbacks per reduce: a plain stream-ordered call (~10 us of host time)
        self.comm_stream = torch.cuda.Stream() if env["world"] > 1 and B > 2 else None
        self.reducer = sharding.PartialMixReducer(env["dist"] if env["world"] > 1 else None, root=0, comm_stream=self.comm_stream)
        self.pending = [None, None]
        self.owed = [False, False]
        rc = self.ctx.process_block_raw(self.srcs[0].data_ptr(), self.slots, n_local, frames, self.buckets[0][0].data_ptr(), self.peaks.data_ptr(), gas.capi.MEM_DEVICE)
        if rc != 0:
            raise SystemExit(f"gas_process_block failed: {rc}")
        torch.cuda.synchronize()
        # raw device addresses, looked up once: tensor indexing costs microseconds per call and this loop is the host
        # side of a ~20 us callback
        self.pset_ptr = [t.data_ptr() for t in self.psets]
        self.src_ptr = [t.data_ptr() for t in self.srcs]
        self.out_ptr = [[self.buckets[b][i].data_ptr() for i in range(B)] for b in range(2)]
        self.peaks_ptr = self.peaks.data_ptr()

    def step(self, k):
        ctx, B, n, F = self.ctx, self.B, self.n_local, self.frames
        if k % 2 == 0:
            ctx.params_publish_device(self.pset_ptr[(k // 2) % 2], n)
        b, i = (k // B) % 2, k % B
        if i == 0:
            self.reducer.wait(self.pending[b])  # the bucket's previous reduce must be done before it is rewritten
            self.pending[b] = None
            self.owed[b] = True  # this bucket is being filled: it owes rank 0 one reduce
        rc = ctx.process_block_raw(self.src_ptr[k % self.n_bufs], None, n, F, self.out_ptr[b][i], self.peaks_ptr, 1)
        if rc != 0:
            raise SystemExit(f"gas_process_block failed: {rc}")
        if self.pipelined:
            # the last mix of the previous bucket rode in the launch above: that bucket is complete in stream order now
            # (batched launches: the first batch of this bucket carries the sums of the previous bucket's last batch)
            aligned = self.depth > 1 and B % self.depth == 0
            if i == (self.depth - 1 if aligned else 0) and k > i:
                if self.depth > 1 and not aligned and self.env["world"] > 1:
                    ctx.join_outputs()  # buckets and batches do not line up: run what waits, sum what is pending
                self.pending[1 - b] = self.reducer.reduce(self.buckets[1 - b])
                self.owed[1 - b] = False
        elif i == B - 1:
            self.pending[b] = self.reducer.reduce(self.buckets[b])
            self.owed[b] = False

    def drain(self, k_end):
        # the bucket holding the last callback still has to reach rank 0: always in pipelined mode (its reduce is
        # issued one callback late), else only when it is partly filled
        B = self.B
        if k_end > 0 and any(self.owed):
            self.ctx.join_outputs()  # run what waits for its batch, enqueue the pending sums
            last = ((k_end - 1) // B) % 2
            for b in (1 - last, last):  # the older bucket first (batched launches can leave both owing)
                if self.owed[b]:
                    self.reducer.wait(self.pending[b])
                    self.pending[b] = self.reducer.reduce(self.buckets[b])
                    self.owed[b] = False
        for i in range(2):
            self.reducer.wait(self.pending[i])
            self.pending[i] = None
        if self.comm_stream is not None:
            self.torch.cuda.current_stream().wait_stream(self.comm_stream)

    def fence(self):
        torch, env = self.torch, self.env
        torch.cuda.synchronize()
        if env["world"] > 1:
            env["dist"].barrier()
            torch.cuda.synchronize()

    def timed(self, steps, warmup):
        """(GPU-timeline ms, wall ms, host enqueue ms) of exactly `steps` callbacks, every output complete inside."""
        torch = self.torch
        n_pre = max(0, CONDITION_STEPS - warmup) + warmup
        for k in range(n_pre):
            self.step(k)
        self.drain(n_pre)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.fence()
        t0 = time.perf_counter()
        ev0.record()
        for k in range(steps):
            self.step(k)
        t_enq = time.perf_counter() - t0
        self.drain(steps)
        ev1.record()
        self.fence()
        dt_wall = time.perf_counter() - t0
        return ev0.elapsed_time(ev1), dt_wall * 1e3, t_enq * 1e3

    def marked(self, callbacks):
        """Dominant-kernel time from HIP events recorded inside the library around every callback's main launch."""
        callbacks = max(self.depth, callbacks // self.depth * self.depth)  # whole batches: every timed launch has the same size
        self.ctx.profile_enable(1)
        self.ctx.profile_read(reset=True)
        for k in range(callbacks):
            self.step(k)
        self.drain(callbacks)
        self.torch.cuda.synchronize()
        prof = self.ctx.profile_read(reset=True)
        self.ctx.profile_enable(False)
        return prof

    def latency(self, callbacks):
        """Synchronous callbacks one at a time (publish every second one, process_block, wait for the mix): host-timed."""
        ctx, n, F = self.ctx, self.n_local, self.frames
        out = self.out_ptr[0][0]
        ts = []
        for k in range(callbacks + 16):
            t0 = time.perf_counter()
            if k % 2 == 0:
                ctx.params_publish_device(self.pset_ptr[(k // 2) % 2], n)
            rc = ctx.process_block_raw(self.src_ptr[k % self.n_bufs], None, n, F, out, self.peaks_ptr, 1)
            ctx.synchronize()
            ts.append(time.perf_counter() - t0)
            if rc != 0:
                raise SystemExit(f"gas_process_block failed: {rc}")
        a = np.sort(np.array(ts[16:])) * 1e3
        return {"callbacks": callbacks, "p50_ms": float(a[len(a) // 2]), "p99_ms": float(a[min(len(a) - 1, int(np.ceil(0.99 * len(a))) - 1)]), "max_ms": float(a[-1])}

    def close(self):
        self.ctx.close()


def in_process_multi(env, devices, n_per_shard, callbacks=200, warmup=30):
    """SURVEY 8e's single-process form (gas_multi_*): one context per entry of `devices`, every shard's partial mix
    written all-to-one into the root's gather buffer, one ordered sum on the root -- per callback, i.e. the REAL-TIME
    arrangement (no reduce bucket).  Device-resident shard inputs (gas_multi_process_block_mem(GAS_MEM_DEVICE)): the call
    only enqueues.  Returns per-callback times: GPU timeline of the root stream and host clock."""
    gas, torch, synth = env["gas"], env["torch"], env["synth"]
    args, kind, chain, frames, ring, hrir = env["args"], env["kind"], env["chain"], env["frames"], env["ring"], env["hrir"]
    K = gas.capi
    G = len(devices)
    multi = K.MultiContext(devices, max_sources=n_per_shard, frames=frames, er_ring_frames=ring, flags=K.FLAG_PEAKS_DRAINING_ONLY)
    try:
        prng = np.random.default_rng(4321)
        slots, srcs, peaks, psets = [], [], [], []
        for g, ctx in enumerate(multi.shards):
            if hrir is not None:
                ctx.hrtf_load(hrir)
            sl = ctx.source_alloc_many(n_per_shard, kind, chain)
            slots.append(np.ascontiguousarray(sl, np.uint32))
            ctx.params_publish_batch(sl, synth.draw_params(prng, n_per_shard, dirs=args.dirs, ring_frames=max(ring, 2 * frames), frames=frames))
            dev = f"cuda:{devices[g]}"
            n_bufs = max(2, min(8, -(-(N_SRC_BUFFERS_BYTES // G) // (n_per_shard * frames * 8))))
            srcs.append([torch.rand(n_per_shard, frames, 2, device=dev) - 0.5 for _ in range(n_bufs)])
            peaks.append(torch.zeros(n_per_shard, 2, device=dev))
            p = synth.draw_params(prng, n_per_shard, dirs=args.dirs, ring_frames=max(ring, 2 * frames), frames=frames)
            psets.append(torch.from_numpy(p.view(np.uint8).reshape(n_per_shard, 128).copy()).to(dev))
        out = torch.zeros(1, frames, 2, device=f"cuda:{devices[0]}")
        torch.cuda.synchronize()
        root = torch.cuda.ExternalStream(multi.lib.gas_multi_root_stream(multi.h), device=f"cuda:{devices[0]}")
        # raw ctypes argument arrays built once: the loop below is the host side of a ~20 us callback
        import ctypes as C

        vp = C.c_void_p
        a_sl = (vp * G)(*[s.ctypes.data for s in slots])
        a_same = (vp * G)(*([None] * G))  # slots[g] == NULL: the shard's previous list (no re-grouping per callback)
        a_pk = (vp * G)(*[t.data_ptr() for t in peaks])
        a_n = (C.c_uint32 * G)(*([n_per_shard] * G))
        a_src = [(vp * G)(*[srcs[g][b % len(srcs[g])].data_ptr() for g in range(G)]) for b in range(max(len(x) for x in srcs))]

        def step(k):
            if k % 2 == 0:  # the emulated physics tick: device-resident rows, per shard
                for g, ctx in enumerate(multi.shards):
                    ctx.params_publish_device(psets[g].data_ptr(), n_per_shard)
            rc = multi.lib.gas_multi_process_block_mem(multi.h, a_src[k % len(a_src)], a_same, a_n, frames, vp(out.data_ptr()), a_pk, K.MEM_DEVICE)
            if rc != 0:
                raise RuntimeError(f"gas_multi_process_block_mem: {rc}")

        rc = multi.lib.gas_multi_process_block_mem(multi.h, a_src[0], a_sl, a_n, frames, vp(out.data_ptr()), a_pk, K.MEM_DEVICE)  # names the lists
        if rc != 0:
            raise RuntimeError(f"gas_multi_process_block_mem: {rc}")
        for k in range(warmup):
            step(k)
        multi.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(root)
        for k in range(callbacks):
            step(k)
        t_enq = time.perf_counter() - t0
        e1.record(root)
        multi.synchronize()
        wall = time.perf_counter() - t0
        gpu_ms = e0.elapsed_time(e1)
        return {"shards": G, "devices": list(devices), "sources_per_shard": n_per_shard, "callbacks": callbacks, "ms_per_step": gpu_ms / callbacks, "wall_ms_per_step": wall * 1e3 / callbacks, "host_enqueue_us_per_step": t_enq * 1e6 / callbacks, "value": G * n_per_shard * frames * callbacks / (wall if len(set(devices)) > 1 else gpu_ms * 1e-3), "direct_write": os.environ.get("GAS_MULTI_DIRECT", "1") != "0"}
    finally:
        multi.close()


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-worker":  # child of cpu_baseline(): CPU only, prints seconds
        kind, chain, frames, dirs, hp, ring, n, blocks = json.loads(sys.argv[2])
        hrir = np.load(hp) if hp else None
        print(_cpu_worker((kind, tuple(chain), frames, dirs, hrir, ring, n, blocks, int(sys.argv[3]))))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="hrtf", choices=sorted(WORKLOADS))
    ap.add_argument("--sources-per-gpu", type=int, default=0)
    ap.add_argument("--dirs", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-max-sources", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the ordered / exact-peaks / latency / copy-ceiling / max-sources passes (profiling runs)")
    ap.add_argument("--marked-callbacks", type=int, default=640, help="callbacks of the separate pass that brackets the dominant launch with HIP events (0 = none: roofline fields empty)")
    ap.add_argument("--reduce-bucket", type=int, default=32, help="callbacks per cross-GPU reduce (N > 1); 1 = every callback's mix is reduced on its own (real-time arrangement)")
    ap.add_argument("--crossfade", action="store_true", help="GAS_FLAG_HRTF_CROSSFADE: blend old/new HRIRs when a source's direction changes (SURVEY 8f#4)")
    ap.add_argument("--no-pipelined-mix", action="store_true", help="headline without GAS_FLAG_PIPELINED_MIX: the partial-mix sum of callback t runs before callback t+1's DSP kernel instead of under it")
    ap.add_argument("--no-batched-launch", action="store_true", help="throughput mode without GAS_FLAG_BATCHED_LAUNCH: one k_hrtf_uni launch per callback instead of one k_hrtf_multi launch per --batch-depth callbacks")
    ap.add_argument("--batch-depth", type=int, default=10, help="GAS_FLAG_BATCHED_LAUNCH: callbacks per k_hrtf_multi launch (2 .. 16)")
    ap.add_argument("--direction-order", action="store_true", help="GAS_FLAG_DIRECTION_ORDER: let the library group sources by HRIR direction (device sort per publish)")
    ap.add_argument("--presorted-directions", action="store_true", help="GAS_FLAG_DIRECTION_RUNS with parameters whose HRIR directions are grouped in callback order (what a caller that sorts its list gets)")
    ap.add_argument("--xcd-order", action="store_true", help="GAS_FLAG_XCD_ORDER: XCD-affine processing order rebuilt on the device after every publish (experiment: fewer L2 fills, no net gain)")
    ap.add_argument("--xcd-directions", action="store_true", help="experiment: draw each source's HRIR direction from the eighth of the table that belongs to its workgroup's XCD (upper bound of an XCD-aware source partition)")
    ap.add_argument("--draining-every", type=int, default=64, help="1 source in N has ended its stream (exact peak needed, audio_spatializer.cpp:464-469); 0 = none")
    ap.add_argument("--exact-peaks", action="store_true", help="headline with the exact peak of every source")
    ap.add_argument("--in-process", action="store_true", help="ONE process drives --gpus N devices through gas_multi_* (all-to-one peer writes of the 4 KiB partial mixes, one ordered sum on device 0, every callback: the real-time arrangement); launch WITHOUT torchrun.  With fewer visible GPUs than N the shards share the devices there are (one GPU: the per-callback cost of the gather + root sum)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import godot_audio_spatializer_amd as gas
    from godot_audio_spatializer_amd import sharding, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.in_process:
        if world != 1:
            raise SystemExit("--in-process is a single process: launch it without torch.distributed.run")
        return main_in_process(args, gas, torch, sharding, synth)
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the spatializer has no CPU path")
    backend = os.environ.get("GAS_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N > 1 path on fewer GPUs than ranks
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    kind, chain, frames, n_default, ring, desc = WORKLOADS[args.workload]
    if args.sources_per_gpu and args.sources_per_gpu != n_default:
        desc += f" -- run with {args.sources_per_gpu} sources/GPU instead"
    # anything that makes the run something other than the workload as described (PMC records only match like with like)
    experiment = " ".join(f for f, on in (("--no-batched-launch", args.no_batched_launch and not args.no_pipelined_mix), ("--xcd-directions", args.xcd_directions), ("--xcd-order", args.xcd_order), ("--direction-order", args.direction_order), ("--presorted-directions", args.presorted_directions), ("--crossfade", args.crossfade)) if on)
    n_local = args.sources_per_gpu or n_default
    n_total = n_local * world
    begin, end = sharding.shard_range(n_total, rank, world)
    assert end - begin == n_local

    rng = np.random.default_rng(1234)
    hrir = synth.synthetic_hrir(rng, dirs=args.dirs) if 3 in chain else None
    env = {"gas": gas, "torch": torch, "synth": synth, "sharding": sharding, "dist": dist, "args": args, "kind": kind, "chain": chain, "frames": frames, "ring": ring, "hrir": hrir, "rank": rank, "local_rank": local_rank, "world": world, "srcs": {}}

    K = gas.capi
    # Peaks are produced where the reference consumes them: playbacks whose stream has ended
    # (audio_spatializer.cpp:464-469).  1 source in 64 is in that state here.  --exact-peaks measures every source's.
    base_flags = 0 if args.exact_peaks else K.FLAG_PEAKS_DRAINING_ONLY
    if args.crossfade:
        base_flags |= K.FLAG_HRTF_CROSSFADE
    if args.direction_order:
        base_flags |= K.FLAG_DIRECTION_ORDER
    if args.presorted_directions:
        base_flags |= K.FLAG_DIRECTION_RUNS
    if args.xcd_order:
        base_flags |= K.FLAG_XCD_ORDER
    head_flags = base_flags | (0 if args.no_pipelined_mix else K.FLAG_PIPELINED_MIX | (0 if args.no_batched_launch else K.FLAG_BATCHED_LAUNCH))

    # ---- headline pass: exactly --steps callbacks, no markers inside -----------------------------------------------
    run = Runner(env, head_flags, n_local, args.reduce_bucket)
    gpu_ms, wall_ms, enq_ms = run.timed(args.steps, args.warmup)
    if world > 1:
        tmax = torch.tensor([gpu_ms, wall_ms], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        gpu_ms, wall_ms = float(tmax[0].item()), float(tmax[1].item())
    ms_per_step = gpu_ms / args.steps
    value = n_total * frames * args.steps / (gpu_ms * 1e-3)
    # ---- marked pass: the dominant kernel's span, events inside the library ----------------------------------------
    prof = run.marked(args.marked_callbacks) if args.marked_callbacks > 0 else {"launches": 0, "kernel_ms": 0.0, "bytes_per_launch": 0, "kernel": ""}
    n_draining = run.n_draining

    result = None
    achieved = 0.0
    peaks_desc = "every source" if args.exact_peaks else f"draining sources only ({n_draining} of {n_local} per GPU)"
    if rank == 0:
        rccl_ranks = dist.get_world_size() if world > 1 else 1
        result = {
            "metric": "mixed AudioFrames/s",
            "value": value,
            "unit": "AudioFrames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "wall_ms_per_step": wall_ms / args.steps,
            "config": {
                "workload": desc,
                "mode": ("ordered (synchronous two-dispatch callback)" if args.no_pipelined_mix else "throughput: callbacks queued back to back, GAS_FLAG_PIPELINED_MIX (the sum of callback t's partial mixes rides in a later launch)" + ("" if args.no_batched_launch else f" + GAS_FLAG_BATCHED_LAUNCH ({args.batch_depth} consecutive callbacks per launch; outputs complete at gas_ctx_join_outputs)")) + ("" if args.exact_peaks else " + GAS_FLAG_PEAKS_DRAINING_ONLY"),
                "timing": f"one HIP event pair on the launch stream around the {args.steps} callbacks (GPU timeline), {CONDITION_STEPS} untimed conditioning callbacks in front, max over ranks; wall_ms_per_step = host clock around the same region incl. the closing synchronize",
                "parity": "unpinned: HRTF / early reflections have no reference counterpart, outputs are checked against this repository's oracle (DESIGN.md section 0)" if (3 in chain or 2 in chain) else "oracle restates audio_spatializer_3d.cpp:554-609; engine primitives unpinned (DESIGN.md section 0)",
                "sources_total": n_total,
                "sources_per_gpu": n_local,
                "frames_per_callback": frames,
                "sample_rate_hz": 48000,
                "hrir_directions": args.dirs if hrir is not None else 0,
                "hrir_crossfade": bool(args.crossfade),
                "pipelined_mix": not args.no_pipelined_mix,
                "callbacks_per_launch": run.depth,
                "host_enqueue_us_per_step": enq_ms / args.steps * 1e3,
                "peaks": peaks_desc,
                "experiment": experiment,
                "parallelism": (f"source-sharded x{world} (world {world}, this rank on cuda:{local_rank}, {backend} reports {rccl_ranks} ranks), sum-reduce to rank 0 of {run.B} callbacks' partial mixes ({run.B * frames * 8} B) per collective" + (" on a side stream" if run.comm_stream is not None else " (stream-ordered on the compute stream)") + (" -- throughput arrangement: a callback's mix reaches rank 0 up to that many callbacks later; --reduce-bucket 1 is the real-time arrangement" if run.B > 1 else "")) if world > 1 else "single GPU (world 1, cuda:%d)" % local_rank,
                "reduce_bucket": run.B,
                "realtime_budget_ms": frames / 48000.0 * 1e3,
            },
            "roofline": roofline_entry(prof, n_local, desc, peaks_desc, not args.no_pipelined_mix, experiment),
        }
        result["config"]["lib_sha16"] = lib_sha16()
        achieved = result["roofline"]["achieved"] * 1e9

    # ---- extras on one GPU ------------------------------------------------------------------------------------------
    extras = rank == 0 and world == 1 and not args.no_extras
    if extras:
        try:
            # same-run copy-bandwidth ceiling over the dominant launch's algorithmic bytes (reads: frames + history +
            # taps; writes: history + partial mixes), best of a few launch geometries
            B_alg = int(result["roofline"]["algorithmic_bytes_per_launch"]) or n_local * frames * 8
            wr = (n_local * (512 - frames // 2) * 4 if 3 in chain else 0) + (1 << 20)
            wr = min(wr, B_alg // 2) // 16 * 16
            rd = (B_alg - wr) // 16 * 16
            best = None
            for wgs, unroll in ((2048, 8), (4096, 8), (4096, 4), (1024, 2), (512, 4)):
                us = run.ctx.bandwidth_probe(rd, wr, wgs, unroll, 24)
                if best is None or us < best[0]:
                    best = (us, wgs, unroll)
            copy_bps = (rd + wr) / (best[0] * 1e-6)
            result["roofline"].update({"copy_us": best[0], "copy_peak": copy_bps / 1e9, "frac_of_copy": (achieved / copy_bps) if copy_bps > 0 else None, "copy_probe": f"gas_bandwidth_probe: {rd} B read from a rotating 320 MiB arena + {wr} B written, {best[1]} workgroups x 256 threads, {best[2]} loads in flight per thread, same event bracket as kernel_us"})
        except Exception as e:
            result["roofline"]["copy_error"] = repr(e)
    run.close()
    del run
    if extras:
        try:
            ord_flags = base_flags & ~K.FLAG_PIPELINED_MIX
            r2 = Runner(env, ord_flags, n_local, 1)
            g2, w2, _ = r2.timed(200, 20)
            p2 = r2.marked(32)
            result["ordered"] = {"ms_per_step": g2 / 200, "value": n_local * frames * 200 / (g2 * 1e-3), "wall_ms_per_step": w2 / 200, "kernel_us": p2["kernel_ms"] / max(p2["launches"], 1) * 1e3, "steps": 200, "what": "no GAS_FLAG_PIPELINED_MIX: one DSP launch per callback and its sum complete in stream order -- what the plugin's synchronous mix() runs; peaks: " + peaks_desc, "roofline": roofline_entry(p2, n_local, desc, peaks_desc, False, "")}
            lat = {"sources": n_local, **r2.latency(256)}
            r2.close()
            del r2
            if head_flags & K.FLAG_BATCHED_LAUNCH:
                r5 = Runner(env, head_flags & ~K.FLAG_BATCHED_LAUNCH, n_local, args.reduce_bucket)
                g5, w5, _ = r5.timed(200, 20)
                p5 = r5.marked(32)
                result["unbatched"] = {"ms_per_step": g5 / 200, "value": n_local * frames * 200 / (g5 * 1e-3), "wall_ms_per_step": w5 / 200, "kernel_us": p5["kernel_ms"] / max(p5["launches"], 1) * 1e3, "kernel": p5["kernel"], "steps": 200, "what": "throughput mode without GAS_FLAG_BATCHED_LAUNCH: GAS_FLAG_PIPELINED_MIX only, one k_hrtf_uni launch per callback (round 1's headline arrangement)", "roofline": roofline_entry(p5, n_local, desc, peaks_desc, True, "--no-batched-launch")}
                r5.close()
                del r5
            if kind == 2 and not args.exact_peaks:
                r3 = Runner(env, 0, n_local, 1)
                g3, w3, _ = r3.timed(200, 20)
                p3 = r3.marked(32)
                result["exact_peaks"] = {"ms_per_step": g3 / 200, "value": n_local * frames * 200 / (g3 * 1e-3), "kernel_us": p3["kernel_ms"] / max(p3["launches"], 1) * 1e3, "steps": 200, "what": "ordered mode, exact output peak of every source (no GAS_FLAG_PEAKS_DRAINING_ONLY): the reference's per-playback peak computed for all", "roofline": roofline_entry(p3, n_local, desc, "every source", False, "")}
                r3.close()
                del r3
            result["latency"] = {"what": "one synchronous callback at a time (device-resident parameter publish every second callback + gas_process_block + gas_ctx_synchronize), ordered mode, host clock", "budget_ms": frames / 48000.0 * 1e3, "runs": [lat]}
            if args.workload == "hrtf" and n_local == 8192:
                r4 = Runner(env, ord_flags, 10240, 1)  # the north star's ">= 10^4 sources inside one callback", stated directly
                result["latency"]["runs"].append({"sources": 10240, **r4.latency(256)})
                g4, _, _ = r4.timed(200, 20)
                p4 = r4.marked(32)
                desc4 = WORKLOADS[args.workload][5] + " -- run with 10240 sources/GPU instead"
                result["roofline_sync"] = [
                    {"sources": n_local, "ms_per_step": result["ordered"]["ms_per_step"], **result["ordered"]["roofline"]},
                    {"sources": 10240, "ms_per_step": g4 / 200, **roofline_entry(p4, 10240, desc4, f"draining sources only ({r4.n_draining} of 10240 per GPU)", False, "")},
                ]
                r4.close()
                del r4
            if args.workload == "hrtf" and n_local == 8192:
                # the sources as 16-bit PCM streams resident in HBM -- the library samples the 64-frame-delayed
                # windows itself (fused into the HRTF launch), so a callback moves no audio over PCIe and needs no caller rows
                n_cb, n_warm = 60, 10
                per = (n_cb + n_warm + 2) * frames
                block = (np.random.default_rng(9).uniform(-0.5, 0.5, 1 << 24) * 32767).astype(np.int16)
                pcm = np.tile(block, -(-n_local * per // block.size))[: n_local * per]
                sctx = gas.SpatializerContext(max_sources=n_local, frames=frames, flags=K.FLAG_PEAKS_DRAINING_ONLY, device=local_rank)
                sctx.set_stream(torch.cuda.current_stream().cuda_stream)
                sctx.hrtf_load(hrir)
                sl = sctx.source_alloc_many(n_local, kind, chain)
                sctx.params_publish_batch(sl, synth.draw_params(np.random.default_rng(10), n_local, dirs=args.dirs, frames=frames))
                sid = sctx.stream_create(pcm)
                for i, s_ in enumerate(sl):
                    sctx.source_bind_stream(s_, sid, i * per)
                s32 = np.ascontiguousarray(sl, np.uint32)
                s_out = torch.zeros(1, frames, 2, device="cuda")
                s_pk = torch.zeros(n_local, 2, device="cuda")

                def s_step():
                    rc_ = sctx.lib.gas_process_block_streams(sctx.h, s32.ctypes.data, n_local, frames, s_out.data_ptr(), s_pk.data_ptr(), None, K.MEM_DEVICE)
                    if rc_ != 0:
                        raise RuntimeError(f"gas_process_block_streams: {rc_}")

                for _ in range(n_warm):
                    s_step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n_cb):
                    s_step()
                e1.record()
                torch.cuda.synchronize()
                s_ms = e0.elapsed_time(e1) / n_cb
                result["streams"] = {"ms_per_step": s_ms, "value": n_local * frames / (s_ms * 1e-3), "callbacks": n_cb, "what": "gas_process_block_streams: the same sources as 16-bit mono PCM streams resident in HBM (sampling fused into the HRTF launch; ordered mode, one k_mix_reduce per callback; GPU timeline)", "stream_bytes_per_callback": n_local * frames * 2}
                sctx.close()
                del sctx, pcm
            if args.workload == "hrtf" and n_local == 8192:
                # SURVEY 8e, real-time arrangement inside one process: the callback's sources split over G contexts of THIS
                # device (what there is to measure on one GPU: the cost of the all-to-one gather + root sum per callback,
                # beside the unsharded ordered callback above)
                result["in_process_multi"] = {"what": "gas_multi_process_block_mem(GAS_MEM_DEVICE), G contexts sharing cuda:0, 8192 sources in total, a device-resident publish every 2nd callback, every callback gathered and summed on the root (no bucket); compare with ordered.ms_per_step (one context, same sources)", "runs": [in_process_multi(env, [local_rank] * G, n_local // G, callbacks=200) for G in (2, 8)]}
                for r_ in result["in_process_multi"]["runs"]:
                    r_["added_us_vs_one_context"] = (r_["ms_per_step"] - result["ordered"]["ms_per_step"]) * 1e3
            if args.workload == "hrtf" and n_local == 8192:
                # the chains either side of the headline's -- [HIGHSHELF, HRTF] (the reference example's shelf in
                # front of the HRTF, examples/godot-gd-spatializer/gd_spatializer.gd:11-20) and cfg5's [ER, HRTF] -- as
                # synchronous callbacks, one-launch form vs the form it replaced (environment switches read at context creation)
                def chain_us(chain_, n_, frames_, ring_, env_key, env_val):
                    old = os.environ.get(env_key)
                    os.environ[env_key] = env_val
                    try:
                        cctx = gas.SpatializerContext(max_sources=n_, frames=frames_, er_ring_frames=ring_, flags=K.FLAG_PEAKS_DRAINING_ONLY, device=local_rank)
                    finally:
                        if old is None:
                            del os.environ[env_key]
                        else:
                            os.environ[env_key] = old
                    cctx.set_stream(torch.cuda.current_stream().cuda_stream)
                    cctx.hrtf_load(hrir)
                    sl_ = cctx.source_alloc_many(n_, K.KIND_EFFECT, chain_)
                    cctx.params_publish_batch(sl_, synth.draw_params(np.random.default_rng(12), n_, dirs=args.dirs, ring_frames=max(ring_, 2 * frames_), frames=frames_))
                    c_src = torch.rand(n_, frames_, 2, device="cuda") - 0.5
                    c_out = torch.zeros(1, frames_, 2, device="cuda")
                    c_pk = torch.zeros(n_, 2, device="cuda")
                    for i_ in range(10):
                        cctx.process_block_raw(c_src.data_ptr(), sl_ if i_ == 0 else None, n_, frames_, c_out.data_ptr(), c_pk.data_ptr(), 1)
                    torch.cuda.synchronize()
                    e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0_.record()
                    for _ in range(200):
                        cctx.process_block_raw(c_src.data_ptr(), None, n_, frames_, c_out.data_ptr(), c_pk.data_ptr(), 1)
                    e1_.record()
                    torch.cuda.synchronize()
                    cctx.close()
                    return 1e3 * e0_.elapsed_time(e1_) / 200

                result["chains"] = {
                    "what": "ordered callbacks (one DSP launch + k_mix_reduce, GPU timeline, 200 callbacks), GAS_FLAG_PEAKS_DRAINING_ONLY, no playback draining; us per callback",
                    "runs": [
                        {"chain": "[HIGHSHELF, HRTF]", "sources": 8192, "frames": 512, "one_launch_us": chain_us((1, 3), 8192, 512, 0, "GAS_UNI_FLT", "1"), "two_launch_us": chain_us((1, 3), 8192, 512, 0, "GAS_UNI_FLT", "0"), "kernels": "k_hrtf_uni<FLT> vs k_shelf_scan + k_hrtf_uni"},
                        {"chain": "[HIGHSHELF, ER, HRTF]", "sources": 8192, "frames": 256, "one_launch_us": chain_us((1, 2, 3), 8192, 256, 4096, "GAS_UNI_ER", "1"), "two_launch_us": chain_us((1, 2, 3), 8192, 256, 4096, "GAS_UNI_ER", "0"), "kernels": "k_shelf_scan + k_hrtf_uni<ER> vs k_shelf_scan + k_er_only + k_hrtf_uni (the chain's last two effects in one launch or two)"},
                    ],
                }
        except Exception as e:  # the extras must never cost the headline line
            result["extras_error"] = repr(e)
    env["srcs"].clear()
    torch.cuda.empty_cache()
    if rank == 0 and world == 1:
        if not args.no_max_sources and args.workload.startswith("hrtf") and not args.no_extras:
            try:
                result["max_sources_under_10ms"] = probe_max_sources(gas, synth, torch, kind, chain, frames, hrir, args.dirs)
                if result["max_sources_under_10ms"] and "roofline" in result["max_sources_under_10ms"] and "roofline_sync" in result:
                    m = result["max_sources_under_10ms"]
                    result["roofline_sync"].append({"sources": m["sources"], "ms_per_step": m["p50_callback_ms"], **m["roofline"]})
            except Exception as e:  # the probe must never cost the headline line
                result["max_sources_under_10ms"] = {"error": str(e)}
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(kind, chain, frames, args.dirs, hrir, ring)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_in_process(args, gas, torch, sharding, synth):
    """bench.py --gpus N --in-process: the same workload, source-sharded over N contexts of ONE process (gas_multi_*)."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the spatializer has no CPU path")
    kind, chain, frames, n_default, ring, desc = WORKLOADS[args.workload]
    n_local = args.sources_per_gpu or n_default
    visible = torch.cuda.device_count()
    devices = [g % visible for g in range(args.gpus)]
    rng = np.random.default_rng(1234)
    hrir = synth.synthetic_hrir(rng, dirs=args.dirs) if 3 in chain else None
    env = {"gas": gas, "torch": torch, "synth": synth, "sharding": sharding, "args": args, "kind": kind, "chain": chain, "frames": frames, "ring": ring, "hrir": hrir}
    shared = len(set(devices)) < len(devices)
    r = in_process_multi(env, devices, n_local, callbacks=args.steps, warmup=max(args.warmup, 30))
    ms = r["wall_ms_per_step"] if not shared and len(devices) > 1 else r["ms_per_step"]
    result = {
        "metric": "mixed AudioFrames/s",
        "value": args.gpus * n_local * frames / (ms * 1e-3),
        "unit": "AudioFrames/s",
        "n_gpus": len(set(devices)),
        "steps": args.steps,
        "warmup": max(args.warmup, 30),
        "ms_per_step": ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": desc,
            "mode": "in-process (gas_multi_*): one context per shard, ordered callbacks, every shard's partial mix written all-to-one into the root's gather buffer and summed there EVERY callback (real-time arrangement, no reduce bucket)",
            "timing": "host clock around the callbacks incl. the closing gas_multi_synchronize (several devices), or the root stream's GPU timeline (shards sharing one device)",
            "shards": args.gpus,
            "devices": devices,
            "shards_share_devices": shared,
            "sources_total": args.gpus * n_local,
            "sources_per_shard": n_local,
            "frames_per_callback": frames,
            "parallelism": f"source-sharded x{args.gpus} in one process over devices {devices}",
            "lib_sha16": lib_sha16(),
        },
        "in_process_multi": r,
    }
    print(json.dumps(result), flush=True)


def probe_max_sources(gas, synth, torch, kind, chain, frames, hrir, dirs, samples=100):
    """Largest N (from a fixed ladder) whose p99 callback time stays under 10 ms on one GPU: `samples` synchronous
    callbacks per rung, host clock around gas_process_block + synchronize.  One context sized for the top rung; each
    rung runs the first N slots."""
    ladder = [1 << 20, 1 << 21, 3 << 20, 1 << 22, 5 << 20, 6 << 20, 7 << 20, 15 << 19, 31 << 18]
    top = ladder[-1]
    best = None
    ctx = gas.SpatializerContext(max_sources=top, frames=frames, channel_count=1, flags=gas.capi.FLAG_PEAKS_DRAINING_ONLY)
    try:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.hrtf_load(hrir)
        slots = ctx.source_alloc_many(top, kind, chain)
        rng = np.random.default_rng(99)
        chunk = 1 << 20
        for a in range(0, top, chunk):  # publish in chunks: the host staging copy is 128 B per source
            ctx.params_publish_batch(slots[a:a + chunk], synth.draw_params(rng, min(chunk, top - a), dirs=dirs, frames=frames))
        src = torch.rand(top, frames, 2, device="cuda") - 0.5
        out = torch.zeros(1, frames, 2, device="cuda")
        peaks = torch.zeros(top, 2, device="cuda")
        for n in ladder:
            rc = ctx.process_block_raw(src.data_ptr(), slots[:n], n, frames, out.data_ptr(), peaks.data_ptr(), 1)
            if rc != 0:
                break
            torch.cuda.synchronize()
            times = []
            for _ in range(samples + 2):
                t0 = time.perf_counter()
                ctx.process_block_raw(src.data_ptr(), None, n, frames, out.data_ptr(), peaks.data_ptr(), 1)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            a = np.sort(np.array(times[2:])) * 1e3
            p99 = float(a[min(len(a) - 1, int(np.ceil(0.99 * len(a))) - 1)])
            if p99 < 10.0:
                best = {"sources": n, "p50_callback_ms": float(a[len(a) // 2]), "p99_callback_ms": p99, "max_callback_ms": float(a[-1]), "samples": samples}
                # the synchronous launch of this rung, event-timed inside the library, as a roofline line of its own
                ctx.profile_enable(1)
                ctx.profile_read(reset=True)
                for _ in range(8):
                    ctx.process_block_raw(src.data_ptr(), None, n, frames, out.data_ptr(), peaks.data_ptr(), 1)
                torch.cuda.synchronize()
                prof = ctx.profile_read(reset=True)
                ctx.profile_enable(0)
                best["roofline"] = roofline_entry(prof, n, WORKLOADS["hrtf"][5] + f" -- run with {n} sources/GPU instead", f"draining sources only (0 of {n} per GPU)", False, "")
            else:
                break
        del src, out, peaks
    finally:
        ctx.close()
        torch.cuda.empty_cache()
    return best


if __name__ == "__main__":
    main()
