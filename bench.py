#!/usr/bin/env python
"""bench.py — STFT frames/s (fp32, N=1024 hop=256) on MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 60 s mono 48 kHz fp32 streams, N=1024, hop=256, periodic
Hann, :valid padding, no scaling -> 11 247 frames per stream.  One 60 s stream is 103.7 MB of traffic
(~18 us at the roofline) and fits the 256 MiB Infinity Cache, so a STEP is one launch over a batch of
`--streams` (default 32) independent 60 s streams per GPU (3.3 GB working set >> L3): steady-state HBM
numbers.  The single-stream figure (config 2 exactly as written) is reported beside it in `single_stream`.
Inputs and outputs are device-resident (HBM) when the timed region starts; N GPUs each process their own
batch (weak scaling, no data-path collective — frames are independent).

No torch: under a launcher (RANK / WORLD_SIZE / LOCAL_RANK in the environment, e.g. torch.distributed.run) every
rank joins an RCCL communicator created by libnxsig.so itself (nxsig_group_create_rank: ncclCommInitRank, the unique id
travels through a file on the node); barrier, max-over-ranks and the optional assembly all-gather go through the C ABI.

Clock pre-conditioning: a GPU that has been idle starts a run of launches at boost clocks, overshoots its power budget
about 5 launches in and takes ~20 launches to settle (round-1 trace: 576 -> 725 -> 571 us).  Before the W warm-up steps
the bench therefore launches the same step until the series has settled (`settled_tail`: the mean of the last 10 launches within
2 % of the mean of the 10 before them and no outlier among them; at most 300 launches / ~0.2 s), untimed, so that a short timed window (--steps 20) measures the steady state.  The per-launch
distribution of the TIMED steps (min / median / p90 / max) is printed so that a transient stays visible.

Prints ONE JSON line on rank 0.  `roofline.achieved` = algorithmic bytes per launch
(hop*4 + K*8 = 9216 B/frame x frames) / mean kernel time measured with HIP events on the library's stream.
`cpu_baseline` = the oracle's C restatement of the BinaryBackend path (oracle/bb_baseline.c) timed on this
box's host cores on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR = 48000
SECONDS = 60
N_FFT = 1024
HOP = 256
L = SR * SECONDS                      # 2 880 000 samples
M = (L - N_FFT) // HOP + 1            # 11 247 frames
BYTES_PER_FRAME = HOP * 4 + N_FFT * 8  # 9 216 algorithmic bytes / frame
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md chip table (spec); measured copy ceiling 6290


def synth(seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal(L, dtype=np.float32)  # N(0,1), never zero-filled (DVFS)


def effective_cores():
    """CPU time this process may actually burn, in cores: min(affinity mask, cgroup CPU quota).  A container on a 256-thread host
    reports 256 from nproc / sched_getaffinity while its cgroup grants a handful (the measured all-thread speed-up says which)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota, src = None, None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota, src = float(q) / float(per), "cgroup v2 cpu.max"
    except Exception:
        pass
    if quota is None:
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota, src = q / per, "cgroup v1 cfs_quota"
        except Exception:
            pass
    eff = aff if quota is None else min(float(aff), quota)
    return {"affinity": aff, "cgroup_quota": quota, "effective": eff, "source": src or "sched_getaffinity (no cgroup CPU quota set)"}


def cpu_baseline(max_seconds: float):
    """oracle C port (single thread, like one BEAM scheduler) on a bounded sample of the same workload."""
    from oracle import bb_baseline, nx_oracle as O

    w = O.hann(N_FFT)
    x = synth(1234)
    nfr = 1024  # calibrate
    t0 = time.perf_counter()
    bb_baseline.stft(x[: (nfr - 1) * HOP + N_FFT], w, HOP, N_FFT, threads=1)
    dt = time.perf_counter() - t0
    frames = int(min(M, max(nfr, nfr * (max_seconds * 0.6) / max(dt, 1e-6))))
    seg = x[: (frames - 1) * HOP + N_FFT]
    t0 = time.perf_counter()
    bb_baseline.stft(seg, w, HOP, N_FFT, threads=1)
    dt1 = time.perf_counter() - t0
    # all-cores leg: ~5 s of wall on every host core the process may use (OpenMP over frames, persistent per-thread
    # scratch).  os.cpu_count() can exceed what a container is allowed to burn, so the pass count comes from a MEASURED
    # all-thread pass (which also brings the thread pool up and pre-faults the result buffer), not from cores x 1-thread rate.
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    zbuf = np.zeros((M, N_FFT), np.complex64)
    bb_baseline.stft_repeat(x, w, HOP, N_FFT, 1, cores, out=zbuf)
    t0 = time.perf_counter()
    bb_baseline.stft_repeat(x, w, HOP, N_FFT, 2, cores, out=zbuf)
    per_pass = (time.perf_counter() - t0) / 2
    reps = max(1, min(4096, int(min(max_seconds * 0.3, 5.0) / max(per_pass, 1e-4))))
    t0 = time.perf_counter()
    done, _ = bb_baseline.stft_repeat(x, w, HOP, N_FFT, reps, cores, out=zbuf)
    dtn = time.perf_counter() - t0
    return {
        "value": frames / dt1,
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {frames} frames of one 60 s mono 48 kHz stream (N=1024 hop=256 Hann), oracle/bb_baseline.c "
                  f"recursive radix-2 in f64, 1 thread; Nx.BinaryBackend itself cannot run here (no BEAM: "
                  f"elixir={'found' if _which('elixir') else 'not found'})",
        "all_cores": {"value": done / dtn, "cores": cores, "cores_effective": effective_cores(), "seconds": dtn,
                      "speedup_over_1_thread": (done / dtn) / (frames / dt1),
                      "sample": f"{reps} passes over the 60 s stream ({done} frames), OpenMP static over frames"},
    }


class MemWatch:
    """device-memory high-water mark of this rank (nxsig_mem_info = hipMemGetInfo, sampled after every allocation phase)"""

    def __init__(self, ctx, lib, C):
        self.ctx, self.lib, self.C = ctx, lib, C
        self.total, self.min_free, self.marks = None, None, []

    def sample(self, label):
        f, t = self.C.c_size_t(), self.C.c_size_t()
        try:
            if self.lib.nxsig_mem_info(self.ctx.handle, self.C.byref(f), self.C.byref(t)) != 0:
                return
        except AttributeError:
            return
        self.total = t.value
        used = t.value - f.value
        if self.min_free is None or f.value < self.min_free:
            self.min_free = f.value
        self.marks.append((label, round(used / 2**30, 2)))

    def report(self):
        if self.total is None:
            return None
        return {"high_water_GiB": round((self.total - self.min_free) / 2**30, 2), "device_total_GiB": round(self.total / 2**30, 2),
                "after_GiB_in_use": dict(self.marks)}


def rccl_info(lib, C):
    """the RCCL library libnxsig.so actually dlopen()ed: version code as ncclGetVersion reports it and the shared object's path"""
    try:
        v, buf = C.c_int32(0), C.create_string_buffer(512)
        rc = lib.nxsig_rccl_info(C.byref(v), buf, 512)
        code = int(v.value)
        return {"loaded": rc == 0, "version_code": code, "version": f"{code // 10000}.{code // 100 % 100}.{code % 100}" if code else None,
                "path": buf.value.decode(errors="replace") or None, "headers_built_against": "2.27.7 (/opt/rocm/include/rccl/rccl.h)"}
    except Exception as e:  # noqa: BLE001
        return {"loaded": False, "error": repr(e)[:120]}


def load_diag():
    """tools/libnxsig_diag.so (tools/diag_mix.hip, built by __graft_entry__.build()): no-math traffic models of the iSTFT / FIR
    kernels in their shipped geometry.  Measurement infrastructure only; absent -> the mix ceilings are reported as null."""
    import ctypes as C

    path = os.path.join(ROOT, "tools", "libnxsig_diag.so")
    if not os.path.exists(path):
        return None
    try:
        d = C.CDLL(path)
        d.nxdiag_istft_mix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int]
        d.nxdiag_fir_mix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int]
        d.nxdiag_stft_mix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int]
        d.nxdiag_stft2048_mix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int]
        d.nxdiag_pcie_pinned.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        return d
    except (OSError, AttributeError):   # missing, or an older build without one of the models
        return None


def lap_mean(laps):
    """mean launch duration of a series of event-to-event laps -> (trimmed mean, laps left out).  A lap several times the median is not a
    launch: the host thread was descheduled and the queue ran dry (seen once in ten boxes: one 6.5 ms lap among twenty of 0.4 ms).
    Such laps are left out of the TRIMMED mean and counted (`stalled_laps`); every block reports the all-laps mean beside it
    (`kernel_ms_all_laps` / `frac_all_laps`) and min / median / max over ALL laps — one policy for the headline and the secondaries."""
    med = float(np.median(laps))
    kept = [v for v in laps if v <= 3.0 * med]
    return float(np.mean(kept)), len(laps) - len(kept)


def lap_us(laps):
    return {"min": round(min(laps) * 1e3, 1), "median": round(float(np.median(laps)) * 1e3, 1),
            "p90": round(float(np.percentile(laps, 90)) * 1e3, 1), "max": round(max(laps) * 1e3, 1), "n": len(laps)}


SETTLE_RULE = "mean of the last 10 laps within 2 % of the mean of the 10 before them, and no lap of the last 10 above 1.10 x their minimum"


def settled_tail(hist):
    """the clock transient of a GPU that was idle (boost -> power excursion -> steady state: 576 -> 725 -> 571 us on the headline) is over when
    two consecutive windows of 10 launches agree in the mean and the last one holds no outlier.  (Round 4's rule — 10 laps within 3 % of
    the running MINIMUM — never fired on a box whose steady-state laps scatter by 6 %: 300 launches, settled: false, on a settled GPU.)"""
    if len(hist) < 20:
        return False
    a, b = hist[-20:-10], hist[-10:]
    return abs(float(np.mean(b)) / float(np.mean(a)) - 1.0) <= 0.02 and max(b) <= 1.10 * min(b)


def settle(ctx, fn, cap, nbytes=None):
    """The headline's clock pre-conditioning as a helper every block uses: launch `fn` from wherever the clocks are (after an upload: idle)
    until the series has settled (`settled_tail`; at least 30, at most `cap` launches), one HIP-event interval per launch.  Returns what a cold caller saw (the first 20 laps) and whether the series settled."""
    ctx.sync()
    hist = []
    settled = False
    while len(hist) < max(cap, 20):
        ctx.timer_lap()
        for _ in range(10):
            fn()
            ctx.timer_lap()
        hist += ctx.timer_laps()
        if len(hist) >= 30 and settled_tail(hist):
            settled = True
            break
    d = {"launches": len(hist), "settled": settled, "rule": SETTLE_RULE, "first10_us": [round(v * 1e3, 1) for v in hist[:10]],
         "last10_us": [round(v * 1e3, 1) for v in hist[-10:]], "min_us": round(min(hist) * 1e3, 1)}
    cold = float(np.mean(hist[:20]))
    d["cold20"] = {"mean_us": round(cold * 1e3, 1)}
    if nbytes:
        d["cold20"]["frac"] = nbytes / (cold * 1e-3) / 1e9 / HBM_PEAK_GBS
    return d


def secondary_rooflines(ctx, lib, S, _lib, C, barrier=None, seconds3=SECONDS, seconds45=600, streams3=16, channels45=8, keep_z4=False,
                        settle_cap=300, yard=None, dry=False, mem=None, check=True):
    """Rooflines of the other BASELINE configs at their FULL per-GPU shard, measured like the headline (HIP events on the
    library's stream, one interval per launch, device-resident data; algorithmic bytes per SURVEY 8d):
      config 3  istft N=1024 hop=256, 16 x 60 s            10 240 B/frame  (K*8 read + hop*8 written, c64 out)
      config 4  stft  N=2048 hop=512, 8 ch x 600 s          18 432 B/frame  (one GPU's share of 64 channels)
      config 5  fir   257 taps :same, 8 ch x 600 s          8 B/sample      (4 in + 4 out)
    Every block is on the headline's footing: after the upload has left the GPU idle, the block's own kernel is
    launched until the clocks have settled (`settle` / `settled_tail`, the headline's rule; at most `settle_cap`
    launches; the first 20 laps are reported as `cold20`), then kernel and no-math traffic model (`mix_ceiling`, tools/diag_mix.hip: the
    same traffic in the kernel's launch geometry) are timed INTERLEAVED — A B A B, 10 laps each after 2 untimed launches — so that
    `kernel_over_ceiling` compares two series taken under the same clocks.  The block's own figure (`achieved`, `frac`) is the series of
    40 laps taken right after the settle loop, before any model launch: the model draws less power than the kernel, and kernel laps that
    follow it pay a clock excursion of their own (`mix_ceiling.kernel_interleaved` shows them).
    Under a launcher EVERY rank runs this on its own shard (`barrier` lines the ranks up before each of the two blocks so the GPUs
    of the node work at the same time); main() reduces the per-rank kernel times with max-over-ranks."""
    out = {}
    rng = np.random.Generator(np.random.PCG64(99))
    diag = load_diag()
    stream = C.c_void_p(lib.nxsig_get_stream(ctx.handle)) if diag is not None else None
    sync = barrier if barrier is not None else (lambda: None)
    yard = yard or {}

    def fill(buf, rows, n):
        chunk = rng.standard_normal(n, dtype=np.float32)
        for r in range(rows):  # one full-entropy stream, rolled per row
            xr = np.roll(chunk, 977 * r)
            _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(buf.ptr + r * n * 4), xr.ctypes.data_as(C.c_void_p), xr.nbytes))
        return chunk   # row r of the buffer is np.roll(chunk, 977 * r): the in-run checks rebuild the slices they need from it

    def measure(fn, reps, warm):
        import gc
        for _ in range(warm):
            fn()
        ctx.sync()
        gc.collect()
        gc.disable()   # a collector pause while the launches are being queued lets the queue run dry (see lap_mean)
        try:
            ctx.timer_lap()
            for _ in range(reps):
                fn()
                ctx.timer_lap()
            return ctx.timer_laps()
        finally:
            gc.enable()

    def interleaved(kernel_fn, mix_fn, rounds=2, reps=10):
        """A B A B: `rounds` x (`reps` kernel laps, `reps` model laps), 2 untimed launches in front of each series"""
        if dry:   # preflight: ONE launch of the kernel and of its model
            return measure(kernel_fn, 1, 0), (measure(mix_fn, 1, 0) if mix_fn is not None else [])
        kl, ml = [], []
        for _ in range(rounds):
            kl += measure(kernel_fn, reps, 2)
            if mix_fn is not None:
                ml += measure(mix_fn, reps, 2)
        return kl, ml

    try:
        traffic_tab = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        traffic_tab = {}

    def timed_after_settle(kernel_fn):
        """the block's number of record: 40 laps right after its own settle loop, nothing else launched in between.  Forty, because the
        power management of a loaded MI355X hunts with a period of about 20 launches of these kernels (config 4's laps of one run:
        1425 .. 1539 .. 1423 us over 20 launches, profiles/r05/README.md): a 20-lap window catches an arbitrary part of one cycle."""
        return measure(kernel_fn, 1 if dry else 40, 0)

    def block(workload, kernel, nbytes, laps, extra, pre, traffic_key=None):
        ms, stalled = lap_mean(laps)
        ms_all = float(np.mean(laps))
        ach = nbytes / (ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes of the same launch shape (profiles/traffic.json), checked against the shape
        traffic = traffic_tab.get(traffic_key + "_bytes_per_launch") if traffic_key and traffic_tab.get(traffic_key + "_algorithmic_bytes") == nbytes else None
        d = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
             "traffic_source": "profiles/traffic.json (PMC passes of tools/profile_bench.sh, not measured in this run)",
             "kernel_ms": ms, "kernel_ms_all_laps": ms_all, "frac_all_laps": nbytes / (ms_all * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "stalled_laps": stalled, "kernel_us": lap_us(laps), "laps_us": [round(v * 1e3, 1) for v in laps], "settled": bool(pre.get("settled")), "precondition": pre,
             "cold20": pre.get("cold20"), "workload": workload, "kernel": kernel, "algorithmic_bytes": nbytes}
        d.update(extra)
        return d

    def ceiling(d, nbytes, klaps, mlaps, what, plain_key=None):
        """the no-math traffic model, timed INTERLEAVED with the kernel (A B A B: same buffers, same stream, same clocks): the ratio
        `kernel_over_ceiling` compares the two interleaved series; `frac` above stays the kernel's own settled series"""
        if not mlaps:
            d["mix_ceiling"] = None
        else:
            ms, stalled = lap_mean(mlaps)
            kms, kst = lap_mean(klaps)
            gbs = nbytes / (ms * 1e-3) / 1e9
            kgbs = nbytes / (kms * 1e-3) / 1e9
            d["mix_ceiling"] = {"GBps": gbs, "frac_of_peak": gbs / HBM_PEAK_GBS, "kernel_over_ceiling": kgbs / gbs, "what": what,
                                "laps_us": lap_us(mlaps), "stalled_laps": stalled, "interleaved": "A B A B, 10 laps each",
                                "kernel_interleaved": {"frac_of_peak": kgbs / HBM_PEAK_GBS, "laps_us": lap_us(klaps), "stalled_laps": kst}}
        if plain_key and yard.get(plain_key):
            # the geometry-free yardstick of the same read : write ratio, timed in this process by yardsticks()
            d["plain_" + plain_key] = {k: yard[plain_key][k] for k in ("GBps", "frac_of_peak", "what") if k in yard[plain_key]}
            d["plain_" + plain_key]["kernel_over_plain"] = d["achieved"] / yard[plain_key]["GBps"]

    # N > 1: the ranks are lined up ONCE per block, outside the try blocks — a rank whose block fails still meets the others at the
    # next line-up (a barrier inside the timed helpers would be skipped by a failing rank and hang the rest)
    sync()
    # ---- config 3: istft of 16 x 60 s
    L3 = SR * seconds3
    M3 = (L3 - N_FFT) // HOP + 1
    try:
        B3 = streams3
        w = S.windows.hann(N_FFT)
        x3 = ctx.empty((B3, L3), np.float32)
        fill(x3, B3, L3)
        z3, _, _ = S.stft(x3, w, ctx=ctx, overlap_length=N_FFT - HOP, fft_length=N_FFT, sampling_rate=SR)
        y3 = ctx.empty((B3, M3 * HOP + N_FFT - HOP), np.complex64)
        p3 = _lib.StftParams(N_FFT, HOP, N_FFT, 0, 0, 0, _lib.SCALE_NONE, 0, float(SR))
        wp = w.ctypes.data_as(C.c_void_p)
        nb3 = B3 * M3 * (N_FFT * 8 + HOP * 8)
        k3 = lambda: _lib.check(lib.nxsig_istft_c64(ctx.handle, C.c_void_p(z3.ptr), M3, B3, wp, C.byref(p3), C.c_void_p(y3.ptr), _lib.DEVICE))  # noqa: E731
        if mem is not None:
            mem.sample("config 3 buffers")
        pre = settle(ctx, k3, settle_cap, nb3) if not dry else {"launches": 0, "settled": False, "dry": True}
        # round trip of config 3 on interior samples (size-independent property): y ~ x — read BEFORE the model overwrites y
        chk = np.empty(4096, np.complex64)
        _lib.check(lib.nxsig_download(ctx.handle, chk.ctypes.data_as(C.c_void_p), C.c_void_p(y3.ptr + 8 * 100000), chk.nbytes))
        ref = np.empty(4096, np.float32)
        _lib.check(lib.nxsig_download(ctx.handle, ref.ctypes.data_as(C.c_void_p), C.c_void_p(x3.ptr + 4 * 100000), ref.nbytes))
        m3 = (lambda: diag.nxdiag_istft_mix(stream, C.c_void_p(z3.ptr), C.c_void_p(y3.ptr), B3 * M3, 8, 3)) if diag is not None else None
        laps = timed_after_settle(k3)
        klaps, mlaps = interleaved(k3, m3)
        out["roofline_istft"] = block(f"config 3: istft N=1024 hop=256, {B3} x {seconds3} s mono 48 kHz, c64 out", "k_istft_wave<1024> (+ k_istft_edge_fix)",
                                      nb3, laps, {"bytes_per_frame": N_FFT * 8 + HOP * 8, "frames": B3 * M3, "frames_per_s": B3 * M3 / (lap_mean(laps)[0] * 1e-3)}, pre, "istft")
        out["roofline_istft"]["roundtrip_max_err"] = float(np.max(np.abs(chk.real - ref)) / np.max(np.abs(ref)))
        ceiling(out["roofline_istft"], nb3, klaps, mlaps,
                "tools/diag_mix.hip k_istft_mix: 8 KiB nt-read + 2 KiB nt-written per frame, no math, 8 runs per CU, two frames ahead, 3 halo frames per run", "mix_4to1")
        for b in (x3, z3, y3):
            b.free()
    except Exception as e:  # noqa: BLE001
        out["roofline_istft"] = {"error": repr(e)[:200]}
    sync()
    # ---- configs 4 / 5: one GPU's 8 channels x 10 min
    z4 = None
    try:
        B4, L4, N4, H4 = channels45, SR * seconds45, 2048, 512
        M4 = (L4 - N4) // H4 + 1
        x4 = ctx.empty((B4, L4), np.float32)
        chunk4 = fill(x4, B4, L4)
        w4 = S.windows.hann(N4)
        z4 = ctx.empty((B4, M4, N4), np.complex64)
        p4 = _lib.StftParams(N4, H4, N4, 0, 0, 0, _lib.SCALE_NONE, 0, float(SR))
        wp4 = w4.ctypes.data_as(C.c_void_p)
        nb4 = B4 * M4 * (H4 * 4 + N4 * 8)
        k4 = lambda: _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(x4.ptr), L4, B4, L4, wp4, C.byref(p4), C.c_void_p(z4.ptr), None, _lib.DEVICE))  # noqa: E731
        if mem is not None:
            mem.sample("config 4 buffers")
        pre = settle(ctx, k4, settle_cap, nb4) if not dry else {"launches": 0, "settled": False, "dry": True}
        tab4 = ctx.to_device(np.zeros(3072, np.float32)) if diag is not None else None
        m4 = (lambda: diag.nxdiag_stft2048_mix(stream, C.c_void_p(x4.ptr), C.c_void_p(z4.ptr), C.c_void_p(tab4.ptr), B4, L4, H4, 8)) if diag is not None else None
        laps = timed_after_settle(k4)
        # in-run check of config 4 (before the model overwrites the spectrum): first frame of the first channel, a mid frame, the last
        # frame of the last channel against the oracle on the matching input slices
        err4 = None
        if check:
            from oracle import nx_oracle as O
            worst, ref = 0.0, 0.0
            for (r, m) in ((0, 0), (B4 // 2, M4 // 2), (B4 - 1, M4 - 1)):
                zg = np.empty((1, N4), np.complex64)
                _lib.check(lib.nxsig_download(ctx.handle, zg.ctypes.data_as(C.c_void_p), C.c_void_p(z4.ptr + (r * M4 + m) * N4 * 8), zg.nbytes))
                xs = np.roll(chunk4, 977 * r)[m * H4: m * H4 + N4]
                zo = O.stft(xs, w4, overlap_length=N4 - H4, fft_length=N4, sampling_rate=SR)[0]
                worst = max(worst, float(np.max(np.abs(zg - zo))))
                ref = max(ref, float(np.max(np.abs(zo))))
            err4 = worst / ref
        klaps, mlaps = interleaved(k4, m4)
        out["roofline_stft2048"] = block(f"config 4 (one GPU's shard of 64 channels): stft N=2048 hop=512, {B4} ch x {seconds45} s @48 kHz", "k_stft_wave<1024, real-2x>",
                                         nb4, laps, {"bytes_per_frame": H4 * 4 + N4 * 8, "frames": B4 * M4, "frames_per_s": B4 * M4 / (lap_mean(laps)[0] * 1e-3)}, pre, "stft2048")
        out["roofline_stft2048"]["max_norm_err_vs_oracle"] = err4
        ceiling(out["roofline_stft2048"], nb4, klaps, mlaps,
                "tools/diag_mix.hip k_stft2048_mix: the real-2x kernel's loads and stores in its launch geometry (8 frames per wave), no math", "mix_1to8")
        if tab4 is not None:
            tab4.free()
        if keep_z4:   # the assembly gathers THIS spectrum: the model overwrote it
            k4()
        else:
            z4.free()
            z4 = None
        h = S.filters.firwin(257, [4000.0], sampling_rate=float(SR))
        y5 = ctx.empty((B4, L4), np.float32)
        hp = h.ctypes.data_as(C.c_void_p)
        k5 = lambda: _lib.check(lib.nxsig_fir_f32(ctx.handle, C.c_void_p(x4.ptr), L4, B4, L4, hp, 257, _lib.CONV_SAME, C.c_void_p(y5.ptr), _lib.DEVICE))  # noqa: E731
        if mem is not None:
            mem.sample("config 5 buffers")
        pre = settle(ctx, k5, settle_cap, B4 * L4 * 8) if not dry else {"launches": 0, "settled": False, "dry": True}
        m5 = (lambda: diag.nxdiag_fir_mix(stream, C.c_void_p(x4.ptr), C.c_void_p(y5.ptr), B4, L4, 8)) if diag is not None else None
        laps = timed_after_settle(k5)
        # in-run check of config 5: three 2 048-sample slices (row start, mid stream, row end) against the direct f64 convolution of
        # the matching input samples (`:same`: y[n] = sum_k h[k] x[n + 128 - k], zero beyond the stream)
        err5 = None
        if check:
            worst, ref = 0.0, 0.0
            hh = np.asarray(h, np.float64)
            for (r, n0) in ((0, 0), (B4 // 2, L4 // 2), (B4 - 1, L4 - 2048)):
                yg = np.empty(2048, np.float32)
                _lib.check(lib.nxsig_download(ctx.handle, yg.ctypes.data_as(C.c_void_p), C.c_void_p(y5.ptr + (r * L4 + n0) * 4), yg.nbytes))
                xr = np.roll(chunk4, 977 * r).astype(np.float64)
                lo, hi = n0 - 128, n0 + 2048 + 128
                seg = np.zeros(hi - lo)
                a, b = max(lo, 0), min(hi, L4)
                seg[a - lo: b - lo] = xr[a:b]
                yo = np.convolve(seg, hh, mode="valid")     # yo[i] = sum_k h[k] seg[i + 256 - k] = y[n0 + i]
                worst = max(worst, float(np.max(np.abs(yg - yo[:2048]))))
                ref = max(ref, float(np.max(np.abs(yo))))
            err5 = worst / ref
        klaps, mlaps = interleaved(k5, m5)
        out["roofline_fir"] = block(f"config 5 (one GPU's shard): fir 257 taps :same, {B4} ch x {seconds45} s @48 kHz", "nxsig_fir_f32 (stream + edge + poison pass)",
                                    B4 * L4 * 8, laps, {"bytes_per_sample": 8, "samples": B4 * L4, "samples_per_s": B4 * L4 / (lap_mean(laps)[0] * 1e-3)}, pre, "fir")
        out["roofline_fir"]["max_norm_err_vs_direct_f64"] = err5
        ceiling(out["roofline_fir"], B4 * L4 * 8, klaps, mlaps,
                "tools/diag_mix.hip k_fir_mix: the overlap-save stream of k_fir_wave<1024> (two 1024-sample blocks read per 1536 outputs, 8-byte accesses, sc1 nt stores), no math",
                "mix_1to1")
        x4.free()
        y5.free()
    except Exception as e:  # noqa: BLE001
        out["roofline_fir"] = out.get("roofline_fir") or {"error": repr(e)[:200]}
    out["_z4"] = z4
    return out


def yardsticks(ctx, lib, _lib, C, diag, xd, x_bytes, zd, z_bytes):
    """Independent bandwidth yardsticks of THIS box, in this process, on the bench's own buffers: the runtime's
    hipMemcpyDtoDAsync, a float4 grid-stride copy / read / fill, and plain 4 : 1, 1 : 8 and 1 : 1 read / write streams over 1 KiB wave
    rows with no kernel-specific geometry (tools/diag_mix.hip k_y_*).  Each: 5 untimed + 20 timed launches (HIP events on the library's
    stream) at two grid sizes, the better one reported.  `zd` (the spectrum buffer) and `xd` (the input) are scratch from here on."""
    if diag is None or not hasattr(diag, "nxdiag_y_copy"):
        return None
    vp, sz = C.c_void_p, C.c_size_t
    diag.nxdiag_y_copy.argtypes = [vp, vp, vp, sz, C.c_int]
    diag.nxdiag_y_fill.argtypes = [vp, vp, sz, C.c_int]
    diag.nxdiag_y_read.argtypes = [vp, vp, vp, sz, C.c_int]
    diag.nxdiag_y_mix.argtypes = [vp, vp, vp, sz, C.c_int, C.c_int, C.c_int]
    diag.nxdiag_y_memcpy.argtypes = [vp, vp, vp, sz]
    stream = vp(lib.nxsig_get_stream(ctx.handle))
    kib = 1024
    half = (z_bytes // 2) // kib * kib
    fifth = (z_bytes // 5) // kib * kib
    steps41 = fifth // kib
    steps18 = min(x_bytes // kib, z_bytes // (8 * kib))
    Z, X = zd.ptr, xd.ptr
    q4 = (z_bytes // (4 * kib))          # 4 KiB wave steps in the spectrum buffer
    h4 = (half // (4 * kib))
    # geometry g: > 0 = grid-stride over g long-lived workgroups, < 0 = short-lived workgroups handing out -g consecutive steps per wave
    geos = (2048, 8192, -2, -8)
    # the iSTFT kernel's own geometry on the plain stream: 8 resident waves per CU (2048 on the chip), each walking ONE contiguous run —
    # 2048 far-apart address streams instead of one contiguous front
    persist41 = steps41 // 2048 + 1
    cases = [
        ("read", z_bytes, "16-byte loads of the spectrum buffer, no stores", geos,
         lambda g: diag.nxdiag_y_read(stream, vp(Z), vp(X), z_bytes, g) if g > 0 else diag.nxdiag_y_mix(stream, vp(Z), vp(X), q4, 4, 0, g)),
        ("copy", 2 * half, "16-byte copy, first half of the spectrum buffer -> second half (read + written bytes)", geos,
         lambda g: diag.nxdiag_y_copy(stream, vp(Z), vp(Z + half), half, g) if g > 0 else diag.nxdiag_y_mix(stream, vp(Z), vp(Z + half), h4, 4, 4, g)),
        ("memcpy_dtod", 2 * half, "hipMemcpyDtoDAsync of the same halves (read + written bytes)", (0,), lambda g: diag.nxdiag_y_memcpy(stream, vp(Z), vp(Z + half), half)),
        ("mix_4to1", steps41 * 5 * kib, "plain 4 : 1 stream (the iSTFT's ratio): a wave reads 4 KiB (4 x 16 B per lane) and writes 1 KiB per step", geos + (-persist41,),
         lambda g: diag.nxdiag_y_mix(stream, vp(Z), vp(Z + 4 * fifth), steps41, 4, 1, g)),
        ("mix_1to8", steps18 * 9 * kib, "plain 1 : 8 stream (the STFT's ratio): a wave reads 1 KiB and writes 8 KiB per step", geos,
         lambda g: diag.nxdiag_y_mix(stream, vp(X), vp(Z), steps18, 1, 8, g)),
        ("mix_1to1", 2 * half, "plain 1 : 1 stream (the FIR's ratio) over 1 KiB wave rows", geos, lambda g: diag.nxdiag_y_mix(stream, vp(Z), vp(Z + half), half // kib, 1, 1, g)),
        ("fill", z_bytes, "16-byte stores over the spectrum buffer, no loads (constant data: last, it overwrites the buffer)", geos,
         lambda g: diag.nxdiag_y_fill(stream, vp(Z), z_bytes, g) if g > 0 else diag.nxdiag_y_mix(stream, vp(X), vp(Z), q4, 0, 4, g)),
    ]
    res = {}
    for name, nbytes, what, grids, fn in cases:
        best = None
        for grid in grids:
            rc = 0
            for _ in range(5):
                rc |= fn(grid)
            ctx.sync()
            ctx.timer_lap()
            for _ in range(20):
                rc |= fn(grid)
                ctx.timer_lap()
            laps = ctx.timer_laps()
            if rc:
                continue
            ms, stalled = lap_mean(laps)
            gbs = nbytes / (ms * 1e-3) / 1e9
            if best is None or gbs > best["GBps"]:
                geo = "runtime" if grid == 0 else (f"grid-stride, {grid} workgroups" if grid > 0 else (
                    f"short-lived workgroups, {-grid} steps per wave" if -grid < 64 else f"persistent: 2048 waves, one contiguous run of {-grid} steps each"))
                best = {"GBps": gbs, "frac_of_peak": gbs / HBM_PEAK_GBS, "bytes": int(nbytes), "geometry": geo, "laps_us": lap_us(laps), "stalled_laps": stalled, "what": what,
                        "all_geometries_GBps": {}}
            res.setdefault("_all", {}).setdefault(name, {})["persistent_2048_runs" if grid < -64 else str(grid)] = round(gbs)
        if best is not None:
            best["all_geometries_GBps"] = res.get("_all", {}).get(name, {})
        res[name] = best
    res.pop("_all", None)
    return res


def assembly_config4(ctx, group, lib, S, _lib, C, world, rank, channels, seconds, share_gpu, mem=None, reps=3):
    """the config-4-sized final assembly, timed SEPARATELY from frames/s.  Every rank computes its shard of the 64-channel
    spectrogram (stft N=2048 hop=512, `channels` x `seconds` s; 8 x 600 s = 7.37 GB) straight into its slot of a full-size buffer and
    `nxsig_group_allgather` (RCCL ncclAllGather, in place) leaves all `world` shards on every GPU — 51.6 GB received per GPU at
    world 8, priced against the 7 x 153 GB/s of xGMI links a GPU has.  If the buffers do not fit, the channel count is halved."""
    N4, H4 = 2048, 512
    L4 = SR * seconds
    M4 = (L4 - N4) // H4 + 1
    ch = channels
    zt = x4 = None
    while True:
        # the shard size must be the SAME on every rank (it is the all-gather's count): a rank that cannot allocate makes every
        # rank halve the channel count (one all-reduce per attempt), never just itself
        ok, err = True, None
        try:
            zt = ctx.empty((world, ch, M4, N4), np.complex64)
            x4 = ctx.empty((ch, L4), np.float32)
        except Exception as e:  # noqa: BLE001  (out of memory: a smaller shard)
            ok, err = False, repr(e)[:120]
        if group.allreduce([0.0 if ok else 1.0], "max")[0] == 0.0:
            break
        for b in (zt, x4):
            if b is not None:
                b.free()
        zt = x4 = None
        if ch == 1:
            return {"error": "config-4-sized assembly buffers do not fit on some rank" + (": " + err if err else "")}
        ch //= 2
    shard = ch * M4 * N4 * 8
    if mem is not None:
        mem.sample("config-4 assembly buffers")
    rng = np.random.Generator(np.random.PCG64(4242 + rank))
    chunk = rng.standard_normal(L4, dtype=np.float32)
    for r in range(ch):
        xr = np.roll(chunk, 977 * r)
        _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(x4.ptr + r * L4 * 4), xr.ctypes.data_as(C.c_void_p), xr.nbytes))
    own = zt.ptr + rank * shard
    w4 = S.windows.hann(N4)
    p4 = _lib.StftParams(N4, H4, N4, 0, 0, 0, _lib.SCALE_NONE, 0, float(SR))
    _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(x4.ptr), L4, ch, L4, w4.ctypes.data_as(C.c_void_p), C.byref(p4), C.c_void_p(own), None, _lib.DEVICE))
    ctx.sync()
    before = np.empty((64, N4), np.complex64)
    _lib.check(lib.nxsig_download(ctx.handle, before.ctypes.data_as(C.c_void_p), C.c_void_p(own), before.nbytes))
    counts = [shard] * world
    group.allgather([own], counts, [zt.ptr])  # warm-up (connection set-up, first-touch of the peers' buffers)
    group.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        group.allgather([own], counts, [zt.ptr])
    ctx.sync()
    group.barrier()
    dt = (time.perf_counter() - t0) / reps
    dt = group.allreduce([dt], "max")[0]
    after = np.empty((64, N4), np.complex64)
    _lib.check(lib.nxsig_download(ctx.handle, after.ctypes.data_as(C.c_void_p), C.c_void_p(own), after.nbytes))
    # a peer's shard really arrived: rank r's slot is non-zero and differs from the own one (different seeds)
    peer = (rank + 1) % world
    got = np.empty((64, N4), np.complex64)
    _lib.check(lib.nxsig_download(ctx.handle, got.ctypes.data_as(C.c_void_p), C.c_void_p(zt.ptr + peer * shard), got.nbytes))
    recv = (world - 1) * shard / dt / 1e9
    res = {"collective": "nxsig_group_allgather (RCCL ncclAllGather through the C ABI, in place), every rank's config-4 shard",
           "workload": f"stft N=2048 hop=512, {ch} ch x {seconds} s per rank", "bytes_per_rank": int(shard), "world": world,
           "bytes_received_per_gpu": int((world - 1) * shard), "ms": dt * 1e3, "recv_GBps_per_rank": recv,
           "frac_of_xgmi_7x153": recv / (7 * 153.0), "own_shard_intact": bool(np.array_equal(before.view(np.uint32), after.view(np.uint32))),
           "peer_shard_arrived": bool(world == 1 or (np.any(got != 0) and not np.array_equal(got.view(np.uint32), after.view(np.uint32))))}
    if share_gpu:
        res["note"] = "--share-gpu: the ranks share one device and talk over sockets, the rate says nothing about xGMI"
    zt.free()
    x4.free()
    return res


def reduce_secondary(sec, world, allreduce):
    """N > 1: BASELINE configs 3 / 4 / 5 as the node runs them — every rank its own shard at the same time — reduced with
    max-over-ranks of the per-rank mean kernel time (one all-reduce of three doubles).  A rank whose block failed contributes +inf."""
    keys = [("roofline_istft", "config3", "frames"), ("roofline_stft2048", "config4", "frames"), ("roofline_fir", "config5", "samples")]
    mine = [float(sec.get(k, {}).get("kernel_ms", 1e30)) for k, _, _ in keys]
    worst = allreduce(mine, "max")
    res = {}
    for (k, name, unit), ms in zip(keys, worst):
        d = sec.get(k, {})
        if ms >= 1e29 or "algorithmic_bytes" not in d:
            res[name] = {"error": d.get("error", "a rank failed this block"), "ranks": world}
            continue
        per_gpu = d["algorithmic_bytes"] / (ms * 1e-3) / 1e9
        res[name] = {"workload": d["workload"] + f" — on each of {world} ranks at the same time", "ranks": world,
                     "kernel_ms_max_over_ranks": ms, "kernel_ms_rank0": d["kernel_ms"],
                     f"{unit}_per_s_total": world * d[unit] / (ms * 1e-3), "per_gpu_GBps": per_gpu, "per_gpu_frac": per_gpu / HBM_PEAK_GBS,
                     "rank0": {kk: d[kk] for kk in ("frac", "kernel_us", "mix_ceiling") if kk in d}}
    return res


class FileControl:
    """Control plane of last resort for N > 1 when the RCCL communicator cannot be created: barrier and max-over-ranks
    through files on the node (nxsig_rendezvous_publish / _fetch).  No data moves through it; the measurement itself
    (per-rank kernels, HIP events) is unchanged, only the cross-rank synchronisation is coarser (~ms)."""

    def __init__(self, lib, base, world, rank):
        self.lib, self.base, self.world, self.rank, self.k = lib, base, world, rank, 0

    def _exchange(self, values):
        import ctypes as C
        import struct

        self.k += 1
        blob = struct.pack("<4d", *(list(values) + [0.0] * 4)[:4])
        me = f"{self.base}.c{self.k}.{self.rank}".encode()
        if self.lib.nxsig_rendezvous_publish(me, blob, len(blob)) != 0:
            raise RuntimeError("file control plane: publish failed")
        out = []
        for r in range(self.world):
            buf = (C.c_ubyte * 32)()
            if self.lib.nxsig_rendezvous_fetch(f"{self.base}.c{self.k}.{r}".encode(), buf, 32, 300000, 0) != 0:
                raise RuntimeError("file control plane: fetch timed out")
            out.append(struct.unpack("<4d", bytes(buf)))
        return out

    def barrier(self):
        self._exchange([0.0])

    def allreduce(self, values, op="max"):
        rows = self._exchange(values)
        return [max(r[i] for r in rows) for i in range(len(values))]

    def cleanup(self):
        """every rank removes its own files once nobody can still need them: ranks != 0 announce `fin`, wait for rank 0's
        `fin`, remove; rank 0 waits for every announcement, publishes its own and removes it when the others' are gone"""
        fin = lambda r: f"{self.base}.fin.{r}"  # noqa: E731
        deadline = time.time() + 120.0

        def wait(cond):
            while not cond():
                if time.time() > deadline:
                    return False
                time.sleep(0.002)
            return True

        if self.rank != 0:
            self.lib.nxsig_rendezvous_publish(fin(self.rank).encode(), b"x", 1)
            wait(lambda: os.path.exists(fin(0)))
        else:
            wait(lambda: all(os.path.exists(fin(r)) for r in range(1, self.world)))
            self.lib.nxsig_rendezvous_publish(fin(0).encode(), b"x", 1)
            wait(lambda: not any(os.path.exists(fin(r)) for r in range(1, self.world)))
        for name in [f"{self.base}.c{k}.{self.rank}" for k in range(1, self.k + 1)] + [fin(self.rank)]:
            try:
                os.remove(name)
            except OSError:
                pass


def self_spawn(args) -> int:
    """`python bench.py --gpus N` with no RANK / WORLD_SIZE in the environment: start N copies of this script, one per GPU, with the
    variables a launcher sets (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT).  Rank 0 inherits stdout
    (the ONE JSON line), every rank inherits stderr.  Refuses — loudly, exit 2 — when the node has fewer than N GPUs (unless
    --share-gpu, the one-GPU test mode)."""
    import ctypes as C
    import socket
    import subprocess

    from nx_signal_amd import _lib

    n = args.gpus
    ndev = C.c_int(0)
    try:
        _lib.check(_lib.load().nxsig_device_count(C.byref(ndev)))
    except Exception as e:  # noqa: BLE001
        print(f"bench.py: cannot count GPUs ({e!r})", file=sys.stderr)
        return 2
    if ndev.value < n and not args.share_gpu:
        print(f"bench.py: --gpus {n} requested but this node has {ndev.value} GPU(s); refusing to report a {n}-GPU line from fewer devices "
              f"(--share-gpu is the one-GPU test mode)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    nonce = "%x" % int.from_bytes(os.urandom(8), "little")  # one per launch: part of the rendezvous file's name
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NXSIG_BENCH_SELF_SPAWNED="1", NXSIG_RDZV_NONCE=nonce, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr.fileno()))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


def _which(exe):
    import shutil

    return shutil.which(exe)


def main():
    if os.environ.get("NXSIG_BENCH_HANG_DUMP"):   # diagnostics: every thread's Python stack on stderr after that many seconds (each rank)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["NXSIG_BENCH_HANG_DUMP"]), exit=False, file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=32, help="independent 60 s streams per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the roofline blocks of configs 3 / 4 / 5")
    ap.add_argument("--no-yardsticks", action="store_true", help="skip the plain copy / read / fill / mix bandwidth yardsticks")
    ap.add_argument("--dry", action="store_true",
                    help="PREFLIGHT (first contact with an N-GPU node): allocate exactly what the run allocates, create the group, run ONE launch of every "
                         "block and of both assemblies, and print one JSON line with `dry: true`, the per-rank memory high-water marks and the RCCL "
                         "library actually loaded — no settling, no timing claims, no CPU baseline")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST MODE for one-GPU boxes: every rank uses device LOCAL_RANK %% device_count and claims its own "
                         "NCCL_HOSTID, so several ranks can form an RCCL communicator on ONE GPU (socket transport over lo); "
                         "exercises the launcher / rendezvous / collective path, the rate is NOT a scaling figure")
    ap.add_argument("--secondary-seconds", type=int, default=600, help="length of the config 4 / 5 channels (BASELINE: 600 s; tests shrink it)")
    ap.add_argument("--secondary-channels", type=int, default=8, help="channels per GPU of configs 4 / 5 (BASELINE: 64 channels / 8 GPUs)")
    ap.add_argument("--istft-seconds", type=int, default=SECONDS, help="length of the config 3 streams (BASELINE: 60 s)")
    ap.add_argument("--assembly-channels", type=int, default=None,
                    help="N > 1: channels of the config-4-sized all-gather per rank (default: the whole shard = --secondary-channels; "
                         "--share-gpu: 1, because N ranks x N shards must fit ONE device there)")
    ap.add_argument("--precondition", type=int, default=300,
                    help="max untimed launches spent settling the clocks before the warm-up steps (0 = none)")
    args = ap.parse_args()
    if args.dry:
        args.steps, args.warmup, args.precondition, args.cpu_seconds, args.no_yardsticks = 1, 0, 0, 0.0, True

    # The contract is ONE JSON line on stdout.  RCCL prints a banner ("Hostname : ...", "Librccl path : ...") to the
    # process' stdout when the communicator comes up, so file descriptor 1 points at stderr until the result is ready.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one child per GPU, the environment a launcher would
        # set), never a silent 1-GPU run that reports n_gpus: 1 for a --gpus N request
        os.dup2(saved_stdout, 1)
        sys.exit(self_spawn(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" in os.environ and world >= 1 and args.gpus > 1:
            print(f"[bench rank {rank}] --gpus {args.gpus} but the launcher set WORLD_SIZE={world}: measuring {world} rank(s)", file=sys.stderr)
        args.gpus = world

    import nx_signal_amd as S
    from nx_signal_amd import _lib, sharding
    import ctypes as C

    if args.share_gpu:
        os.environ["NCCL_HOSTID"] = f"nxsig-bench-rank{rank}"
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
        ndev = C.c_int()
        _lib.check(_lib.load().nxsig_device_count(C.byref(ndev)))
        local_rank = local_rank % max(ndev.value, 1)

    group = None
    filectl = None
    comm_error = None
    hung_thread = False
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:  # under a launcher (any world size): RCCL through the C ABI
        # the communicator comes up in a watchdog thread: a rendezvous that never completes (a rank that died, a launcher
        # that does not give the ranks a common parent ...) must not hang the measurement
        import threading

        box = {}

        def _init():
            try:
                box["group"] = sharding.Group.ranked(world, rank, local_rank, timeout_ms=90000)
            except Exception as e:  # noqa: BLE001  (never lose the measurement to the communicator)
                box["error"] = repr(e)[:300]

        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(timeout=150.0 if world > 1 else 60.0)
        if "group" in box:
            group = box["group"]
        else:
            comm_error = box.get("error", "RCCL communicator creation did not finish in time")
            hung_thread = th.is_alive()
            print(f"[bench rank {rank}] RCCL group creation failed: {comm_error}; control plane falls back to files", file=sys.stderr)
            if args.dry and not hung_thread:
                # the preflight exists to find exactly this BEFORE the measured run: a librccl that cannot be loaded or is older than the
                # library accepts (group.cpp: kMinRccl) fails here, with the reason, instead of at the first collective of the real run
                print(json.dumps({"dry": True, "preflight_failed": "RCCL", "error": comm_error, "rccl": rccl_info(_lib.load(), C)}), flush=True)
                sys.exit(3)
            if world > 1:
                filectl = FileControl(_lib.load(), sharding.rendezvous_path() + ".ctl", world, rank)
    ctx = group.contexts[0] if group is not None else S.Context(local_rank)
    lib = _lib.load()
    mem = MemWatch(ctx, lib, C)
    mem.sample("start")
    w = S.windows.hann(N_FFT)
    B = args.streams

    # inputs resident in HBM before the timed region: B independent streams, seed = 1234 + global stream index
    xd = ctx.empty((B, L), np.float32)
    for b in range(B):
        xb = synth(1234 + rank * B + b)   # reproducible: the in-run check below regenerates the streams it samples from these seeds
        _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(xd.ptr + b * L * 4), xb.ctypes.data_as(C.c_void_p), xb.nbytes))
    zd = ctx.empty((B, M, N_FFT), np.complex64)
    p = _lib.StftParams(N_FFT, HOP, N_FFT, _lib.PAD_VALID, 0, 0, _lib.SCALE_NONE, 0, float(SR))
    wp = w.ctypes.data_as(C.c_void_p)
    mem.sample("headline buffers")

    def step(batch=B):
        _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, batch, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, _lib.DEVICE))

    def barrier():
        if group is not None:
            group.barrier()  # waits for the stream, all-reduces one word over RCCL, waits again
        else:
            ctx.sync()
            if filectl is not None:
                filectl.barrier()

    # ---- what a cold caller sees: the first 20 launches from idle, one HIP-event interval each
    step()  # the first call of a shape builds the context's tables (window, twiddles: a one-time 4 ms): not a rate
    ctx.sync()
    ctx.timer_lap()
    for _ in range(1 if args.dry else 20):
        step()
        ctx.timer_lap()
    cold20 = ctx.timer_laps()

    # ---- clock pre-conditioning (untimed, see the module docstring)
    precondition = {"launches": 0, "settled": False, "rule": SETTLE_RULE}
    if args.precondition > 0:
        hist = list(cold20)
        while len(hist) < args.precondition:
            ctx.timer_lap()
            for _ in range(10):
                step()
                ctx.timer_lap()
            hist += ctx.timer_laps()
            if len(hist) >= 30 and settled_tail(hist):
                precondition["settled"] = True
                break
        precondition.update(launches=len(hist), first10_us=[round(v * 1e3, 1) for v in hist[:10]],
                            last10_us=[round(v * 1e3, 1) for v in hist[-10:]], min_us=round(min(hist) * 1e3, 1))
        # what a COLD caller sees: launches 4 .. 10 of the first series sit on the power excursion of a GPU that was idle
        # (boost clocks overshoot the power budget about 5 launches in); reported beside the settled figure, never instead of it
        if len(hist) >= 10:
            burst_ms = float(np.mean(hist[3:10]))
            precondition["cold_burst"] = {"launches": "4..10 of the first series", "mean_us": round(burst_ms * 1e3, 1),
                                          "frac": (B * M * BYTES_PER_FRAME) / (burst_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    for _ in range(args.warmup):
        step()
    import gc
    gc.collect()
    gc.disable()   # no collector pause between two launches of the timed window
    barrier()
    t0 = time.perf_counter()
    ctx.timer_lap()
    for _ in range(args.steps):
        step()
        ctx.timer_lap()
    laps = ctx.timer_laps()  # HIP events on the stream the kernels run on, one interval per step (synchronises)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    kernel_ms_total = float(sum(laps))
    if group is not None:
        elapsed, kernel_ms_total = group.allreduce([elapsed, kernel_ms_total], "max")
    elif filectl is not None:
        elapsed, kernel_ms_total = filectl.allreduce([elapsed, kernel_ms_total], "max")

    frames_per_step = B * M * world
    ms_per_step = elapsed * 1e3 / args.steps
    value = frames_per_step / (elapsed / args.steps)
    kernel_ms = kernel_ms_total / args.steps
    achieved = (B * M * BYTES_PER_FRAME) / (kernel_ms * 1e-3) / 1e9  # GB/s per GPU, algorithmic bytes

    # config 2 exactly as written: ONE 60 s stream per launch, back-to-back launches (L3-resident; one round of workgroups: start-up latencies dominate)
    for _ in range(1 if args.dry else 20):
        step(1)
    ctx.sync()
    ctx.timer_start()
    reps = 1 if args.dry else 200
    for _ in range(reps):
        step(1)
    single_ms = ctx.timer_stop() / reps
    single = {
        "workload": "1 x 60 s mono (config 2 as written, 103.7 MB: Infinity-Cache resident; the one-round geometry of round 6: 176 twelve-wave workgroups, one per CU; start-up and the XCDs' dispatch stagger are ~3.5 of its ~18.5 us, profiles/r06/one_round_phase_trace.txt)",
        "ms_per_launch": single_ms,
        "frames_per_s": M / (single_ms * 1e-3),
        "algorithmic_GBps": M * BYTES_PER_FRAME / (single_ms * 1e-3) / 1e9,
    }

    # ---- the path a drop-in caller takes by default: HOST tensors in, HOST tensors out (NXSIG_HOST: what NxSignalAMD.stft(%Nx.Tensor{}) and
    # the Python mirror on numpy arrays call).  8 x config 2 per call = 92 MB up, 737 MB down over PCIe: chunked pageable copies with the
    # next chunk's pages pre-faulted by host threads (api.cpp: Staged).  Beside it: the link's own rate for the same bytes (one
    # hipMemcpy into / out of pinned memory: nothing faster can cross it) and the pinned-slot pipeline built in round 6 (NXSIG_HOST_PIPE=1,
    # not faster: profiles/r06/host_path.txt).  Never part of `value`.
    host_path = None
    if rank == 0 and not args.dry:
        try:
            HB = 8
            xh = np.stack([synth(1234 + b) for b in range(HB)])   # (stream 0 = the headline's stream 0 on rank 0)
            zh = np.empty((HB, M, N_FFT), np.complex64)

            def host_call():
                _lib.check(lib.nxsig_stft_f32(ctx.handle, xh.ctypes.data_as(C.c_void_p), L, HB, L, wp, C.byref(p), zh.ctypes.data_as(C.c_void_p), None, _lib.HOST))

            host_call()   # first call: pinned slots, scratch, page faults of the result buffer
            t_h = []
            for _ in range(3):
                t0h = time.perf_counter()
                host_call()
                t_h.append(time.perf_counter() - t0h)
            best = min(t_h)
            zfresh = np.empty((HB, M, N_FFT), np.complex64)   # a caller that allocates its result per call (never touched pages)
            t0h = time.perf_counter()
            _lib.check(lib.nxsig_stft_f32(ctx.handle, xh.ctypes.data_as(C.c_void_p), L, HB, L, wp, C.byref(p), zfresh.ctypes.data_as(C.c_void_p), None, _lib.HOST))
            t_fresh = time.perf_counter() - t0h
            same = bool(np.array_equal(zfresh.view(np.uint32), zh.view(np.uint32)))
            ctx.set_tuning("HOST_PIPE", 1)
            host_call()
            t0h = time.perf_counter()
            host_call()
            t_old = time.perf_counter() - t0h
            ctx.clear_tuning("HOST_PIPE")
            zdev = np.empty((M, N_FFT), np.complex64)     # device-resident result of the first stream: the bytes must be the same
            _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, 1, L, wp, C.byref(p), C.c_void_p(zd.ptr), None, _lib.DEVICE))
            _lib.check(lib.nxsig_download(ctx.handle, zdev.ctypes.data_as(C.c_void_p), C.c_void_p(zd.ptr), zdev.nbytes))
            link = None
            dg = load_diag()
            if dg is not None:
                d2h, h2d = C.c_double(0.0), C.c_double(0.0)
                if dg.nxdiag_pcie_pinned(C.c_void_p(zd.ptr), zh.nbytes, C.byref(d2h), C.byref(h2d)) == 0:
                    link = {"d2h_pinned_GBps": d2h.value, "h2d_pinned_GBps": h2d.value,
                            "floor_ms": (zh.nbytes / d2h.value + xh.nbytes / h2d.value) / 1e6}
            host_path = {
                "workload": f"{HB} x config 2 per call through NXSIG_HOST: {xh.nbytes / 1e6:.0f} MB up, {zh.nbytes / 1e6:.0f} MB down",
                "ms_per_call": best * 1e3, "frames_per_s": HB * M / best, "result_GBps": zh.nbytes / best / 1e9,
                "ms_per_call_fresh_result_buffer": t_fresh * 1e3, "frames_per_s_fresh_result_buffer": HB * M / t_fresh,
                "ms_per_call_pinned_pipeline": t_old * 1e3, "frames_per_s_pinned_pipeline": HB * M / t_old,
                "bit_identical_fresh_vs_reused": same,
                "bit_identical_to_device_path": bool(np.array_equal(zh[0].view(np.uint32), zdev.view(np.uint32))),
                "link": link,
                "frac_of_link_d2h": (zh.nbytes / best / 1e9) / link["d2h_pinned_GBps"] if link else None,
            }
            del xh, zh, zfresh
        except Exception as e:  # noqa: BLE001
            host_path = {"error": repr(e)[:300]}

    # optional final assembly, timed SEPARATELY from frames/s: RCCL all-gather of one 60 s stream's
    # spectrum per rank (92 MB each).  Outputs otherwise stay sharded and device-resident.
    assembly = None
    if group is not None:
        zt, local_err = None, None
        try:  # this rank's part: buffer + own shard
            if os.environ.get("NXSIG_BENCH_FAIL_ASSEMBLY_RANK") == str(rank):   # failure injection (tests/test_bench_multi.py)
                raise MemoryError("injected: this rank cannot prepare its shard of the config-2 assembly")
            zt = ctx.empty((world, M, N_FFT), np.complex64)  # full-size buffer: the own shard is computed in place
            mem.sample("config-2 assembly buffer")
            own = zt.ptr + rank * M * N_FFT * 8
            _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xd.ptr), L, 1, L, wp, C.byref(p), C.c_void_p(own), None, _lib.DEVICE))
            ctx.sync()
        except Exception as e:  # noqa: BLE001
            local_err = repr(e)[:200]
        # the ranks AGREE before the first collective is queued: a rank whose part failed must not leave its peers inside an all-gather
        # (round 6: a stale HIP error on rank 0 did exactly that to the one-device test mode)
        any_bad = group.allreduce([1.0 if local_err else 0.0], "max")[0] != 0.0
    if group is not None and any_bad:
        assembly = {"error": local_err or "another rank could not prepare its shard; the collective was skipped on every rank"}
        if zt is not None:
            zt.free()
    elif group is not None:
        try:  # the optional assembly must never take the measurement down with it
            counts = [M * N_FFT * 8] * world
            for _ in range(0 if args.dry else 2):
                group.allgather([own], counts, [zt.ptr])
            group.barrier()
            tg = time.perf_counter()
            reps_g = 1 if args.dry else 5
            for _ in range(reps_g):
                group.allgather([own], counts, [zt.ptr])
            ctx.sync()
            tg = (time.perf_counter() - tg) / reps_g
            chk = np.empty((64, N_FFT), np.complex64)
            ref = np.empty((64, N_FFT), np.complex64)
            _lib.check(lib.nxsig_download(ctx.handle, chk.ctypes.data_as(C.c_void_p), C.c_void_p(own), chk.nbytes))
            _lib.check(lib.nxsig_download(ctx.handle, ref.ctypes.data_as(C.c_void_p), C.c_void_p(zd.ptr), ref.nbytes))
            assembly = {"collective": "nxsig_group_allgather (RCCL ncclAllGather through the C ABI, in place)",
                        "bytes_per_rank": int(M * N_FFT * 8), "ms": tg * 1e3,
                        "recv_GBps_per_rank": (world - 1) * M * N_FFT * 8 / tg / 1e9 if world > 1 else 0.0,
                        "own_shard_intact": bool(np.array_equal(chk.view(np.uint32), ref.view(np.uint32)))}
            zt.free()
        except Exception as e:  # noqa: BLE001
            assembly = {"error": repr(e)[:200]}

    verify = None
    if not args.no_verify and rank == 0:
        from oracle import nx_oracle as O  # checker only, outside the timed region

        # 64 (stream, frame) pairs over the WHOLE launch shape of the timed step: first / last stream, first / last
        # frames of a row (the last pair of a row is ragged: M is odd), both sides of workgroup seams (a workgroup takes 8 frame pairs
        # = 16 frames, a wave 2 pairs), both sides of the row seam, and seeded random picks in every remaining stream
        picks = []
        for b in sorted({0, B - 1}):
            picks += [(b, m) for m in (0, 1, 3, 4, 15, 16, 17, M // 2 - 1, M // 2, M - 17, M - 16, M - 3, M - 2, M - 1) if 0 <= m < M]
        for b in sorted({1, B // 2, B - 2} & set(range(B)) - {0, B - 1}):
            picks += [(b, m) for m in (0, 16, M // 2, M - 16, M - 2, M - 1) if 0 <= m < M]
        prng = np.random.Generator(np.random.PCG64(77))
        rest = [b for b in range(B) if b not in {p[0] for p in picks}]
        while len(picks) < 64 and rest:
            for b in rest:
                if len(picks) < 64:
                    picks.append((b, int(prng.integers(0, M))))
        picks = picks[:64]
        worst, scale_ref, by_stream = 0.0, 0.0, {}
        for b, m in picks:
            by_stream.setdefault(b, []).append(m)
        for b, ms_ in by_stream.items():
            xb = synth(1234 + rank * B + b)
            for m in ms_:
                zg = np.empty((1, N_FFT), np.complex64)
                _lib.check(lib.nxsig_download(ctx.handle, zg.ctypes.data_as(C.c_void_p), C.c_void_p(zd.ptr + (b * M + m) * N_FFT * 8), zg.nbytes))
                zo, _, _ = O.stft(xb[m * HOP: m * HOP + N_FFT], w, overlap_length=N_FFT - HOP, fft_length=N_FFT, sampling_rate=SR)
                worst = max(worst, float(np.max(np.abs(zg - zo))))
                scale_ref = max(scale_ref, float(np.max(np.abs(zo))))
        verify = {"frames": len(picks), "streams": len(by_stream), "max_norm_err": worst / scale_ref, "tolerance": 1e-5,
                  "last_frame_of_last_stream": (B - 1, M - 1) in picks,
                  "what": "max |z_gpu - z_oracle| / max |z_oracle| over the sampled (stream, frame) pairs of the timed launch shape"}

    # ---- the headline's own no-math ceiling: the kernel's stream (32 four-byte loads and 16 sixteen-byte `sc1 nt` stores per lane and
    # frame pair, 2 pairs per wave, short-lived 4-wave workgroups, 12 KB of tables per workgroup) with no arithmetic, same buffers
    # (the spectrum buffer is scratch from here on: the verification above has read it), same stream, right after the timed steps
    headline_ceiling = None
    diag = load_diag()
    if diag is not None and rank == 0:
        try:
            tabd = ctx.empty((3072,), np.float32)
            tb = np.linspace(0.5, 1.5, 3072, dtype=np.float32)
            _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(tabd.ptr), tb.ctypes.data_as(C.c_void_p), tb.nbytes))
            stream = C.c_void_p(lib.nxsig_get_stream(ctx.handle))
            mix = lambda: diag.nxdiag_stft_mix(stream, C.c_void_p(xd.ptr), C.c_void_p(zd.ptr), C.c_void_p(tabd.ptr), B, L, HOP, 2)  # noqa: E731
            for _ in range(0 if args.dry else 10):
                mix()
            ctx.sync()
            ctx.timer_lap()
            for _ in range(1 if args.dry else 20):
                mix()
                ctx.timer_lap()
            mlaps = ctx.timer_laps()
            # the model walks whole pairs: (M // 2) * 2 frames per stream
            gbs = B * (M // 2) * 2 * BYTES_PER_FRAME / (float(np.mean(mlaps)) * 1e-3) / 1e9
            headline_ceiling = {"GBps": gbs, "frac_of_peak": gbs / HBM_PEAK_GBS, "kernel_over_ceiling": achieved / gbs,
                                "what": "tools/diag_mix.hip k_stft_mix: the headline kernel's loads and stores in its launch geometry (2 frame "
                                        "pairs per wave), no math"}
            # config 2 as written beside ITS no-math model (round 6): ONE 60 s stream per launch, kernel and model interleaved A B A B,
            # 100 back-to-back launches per lap.  What the launch shape allows, whatever the arithmetic costs.
            if not args.dry and isinstance(single, dict):
                mix1 = lambda: diag.nxdiag_stft_mix(stream, C.c_void_p(xd.ptr), C.c_void_p(zd.ptr), C.c_void_p(tabd.ptr), 1, L, HOP, 2)  # noqa: E731
                tk, tm = [], []
                for _ in range(4):
                    for fn, acc in ((lambda: step(1), tk), (mix1, tm)):
                        for _ in range(10):
                            fn()
                        ctx.sync()
                        ctx.timer_start()
                        for _ in range(100):
                            fn()
                        acc.append(ctx.timer_stop() / 100)
                single["interleaved_with_model"] = {
                    "kernel_us": round(float(np.median(tk)) * 1e3, 2), "model_us": round(float(np.median(tm)) * 1e3, 2),
                    "model_frames_per_s": (M // 2) * 2 / (float(np.median(tm)) * 1e-3), "kernel_over_ceiling": float(np.median(tm)) / float(np.median(tk)),
                    "what": "tools/diag_mix.hip k_stft_mix on ONE stream: the loads and stores of the launch, 2 frame pairs per wave, no math"}
            tabd.free()
        except Exception as e:  # noqa: BLE001
            headline_ceiling = {"error": repr(e)[:160]}

    # ---- independent bandwidth yardsticks of this box (same process, same buffers; both are scratch from here on)
    yard = None
    if rank == 0 and not args.no_yardsticks:
        try:
            yard = yardsticks(ctx, lib, _lib, C, diag, xd, B * L * 4, zd, B * M * N_FFT * 8)
        except Exception as e:  # noqa: BLE001
            yard = {"error": repr(e)[:160]}

    # ---- BASELINE configs 3 / 4 / 5 at their full per-GPU shard: on every rank at the same time (N > 1: max-over-ranks)
    for b in (xd, zd):
        b.free()
    sec, sec_multi, assembly4 = {}, None, None
    if not args.no_secondary:
        sec = secondary_rooflines(ctx, lib, S, _lib, C, barrier=barrier if world > 1 else None, seconds3=args.istft_seconds,
                                  seconds45=args.secondary_seconds, channels45=args.secondary_channels, settle_cap=args.precondition,
                                  yard=yard if isinstance(yard, dict) and "error" not in yard else None, dry=args.dry, mem=mem,
                                  check=(not args.no_verify and rank == 0))
        sec.pop("_z4", None)
        if world > 1:
            try:
                sec_multi = reduce_secondary(sec, world, group.allreduce if group is not None else filectl.allreduce)
            except Exception as e:  # noqa: BLE001
                sec_multi = {"error": repr(e)[:200]}
    if group is not None and world > 1 and not args.no_secondary:
        ch = args.assembly_channels if args.assembly_channels is not None else (1 if args.share_gpu else args.secondary_channels)
        secs = min(args.secondary_seconds, 60) if (args.share_gpu and args.assembly_channels is None) else args.secondary_seconds
        try:
            assembly4 = assembly_config4(ctx, group, lib, S, _lib, C, world, rank, ch, secs, args.share_gpu, mem=mem, reps=1 if args.dry else 3)
        except Exception as e:  # noqa: BLE001
            assembly4 = {"error": repr(e)[:200]}

    mem.sample("end")
    memrep = mem.report()
    per_rank_hw = None
    if world > 1 and memrep is not None and world <= 64:
        mine = [0.0] * world
        mine[rank] = memrep["high_water_GiB"]
        try:
            per_rank_hw = [round(v, 2) for v in group.allreduce(mine, "max")] if group is not None else None   # (the file control plane carries 4 doubles)
        except Exception:  # noqa: BLE001
            per_rank_hw = None
    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("stft_batch%d_bytes_per_launch" % B)
            except Exception:
                traffic = None
        out = {
            "metric": "STFT frames/sec (fp32, N=1024 hop=256)",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{B} x (60 s mono 48 kHz f32) per GPU per step, N=1024 hop=256 periodic Hann, :valid, "
                            f"c64 full spectrum out; inputs/outputs device-resident",
                "streams_per_gpu": B, "frames_per_stream": M, "frame_length": N_FFT, "hop": HOP, "fft_length": N_FFT,
                "parallelism": f"streams sharded over {world} GPU(s), no data-path collective"
                               + (" [--share-gpu test mode: the ranks share one device, not a scaling figure]" if args.share_gpu else ""),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/traffic.json: PMC FETCH_SIZE x 2 + WRITE_SIZE per launch of this launch shape, collected "
                                  "by tools/profile_bench.sh in separate rocprofv3 --pmc passes; NOT measured in this run",
                "kernel_ms": kernel_ms, "bytes_per_frame": BYTES_PER_FRAME,
                # two clocks, both printed: `frac` = mean HIP-event interval per step (the kernel's launch duration, what rocprofv3 shows);
                # `frac_wall` = the same bytes over ms_per_step (the host clock `value` uses: barrier + sync on both sides included)
                "frac_wall": (B * M * BYTES_PER_FRAME) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_trimmed": (B * M * BYTES_PER_FRAME) / (lap_mean(laps)[0] * 1e-3) / 1e9 / HBM_PEAK_GBS if world == 1 else None,
                "frac_of_copy_this_box": (achieved / yard["copy"]["GBps"]) if isinstance(yard, dict) and yard.get("copy") else None,
                "kernel_us": lap_us(laps),
                "mix_ceiling": headline_ceiling,
                # laps several times the median are host stalls (the queue ran dry), not launches; `value` and `achieved` keep them —
                # the contract times exactly K steps — and this says how many there were
                "stalled_laps": int(sum(1 for v in laps if v > 3.0 * float(np.median(laps)))),
            },
            "value_cold": (world * B * M / (float(np.mean(cold20)) * 1e-3)) if cold20 else None,
            "value_cold_note": "frames/s of the first 20 launches (after the one table-building call) following the idle period of the input upload, before any pre-conditioning "
                               "(what a caller that issues a handful of calls from idle sees); `value` is the settled rate",
            "precondition": precondition,
            "dry": bool(args.dry),
            "preflight": {"memory_rank0": memrep, "high_water_GiB_per_rank": per_rank_hw, "rccl": rccl_info(lib, C) if ("RANK" in os.environ or args.dry) else None,
                          "note": "dry run: one launch of every block, no settling — the figures above are NOT measurements" if args.dry else None},
            "comm": ({"backend": "RCCL via libnxsig.so (ncclCommInitRank)", "world": world, "torch": False, "rccl": rccl_info(lib, C)} if group is not None
                     else ({"backend": "file control plane (RCCL group creation failed)", "world": world, "torch": False,
                            "error": comm_error} if comm_error else None)),
            "single_stream": single,
            "host_path": host_path,
            "assembly": assembly,
            "max_norm_err_vs_oracle": verify["max_norm_err"] if verify else None,
            "verify": verify,
            "yardsticks": yard,
            "device": ctx.name(),
        }
        out.update(sec)  # roofline_istft / roofline_stft2048 / roofline_fir of THIS rank (N = 1: the whole story)
        if sec_multi is not None:
            out.update(sec_multi)  # config3 / config4 / config5: the node's figures, max-over-ranks
        if assembly4 is not None:
            out["assembly_config4"] = assembly4
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        # the line leads with the contract's keys, then FLAT scalars of every block (a reader of the driver's `parsed` record gets the four
        # fractions, the single-stream rate, the cold rate and the in-run error without digging through lap arrays), then the blocks
        def frac_of(key):
            blk = out.get(key)
            return blk.get("frac") if isinstance(blk, dict) else None
        flat = {
            "frac_stft1024": out["roofline"]["frac"],
            "frac_istft1024": frac_of("roofline_istft"),
            "frac_stft2048": frac_of("roofline_stft2048"),
            "frac_fir257": frac_of("roofline_fir"),
            "single_stream_fps": single["frames_per_s"] if isinstance(single, dict) else None,
            "host_path_fps": host_path.get("frames_per_s") if isinstance(host_path, dict) else None,
            "single_stream_over_ceiling": (single.get("interleaved_with_model") or {}).get("kernel_over_ceiling") if isinstance(single, dict) else None,
            "value_cold": out.get("value_cold"),
            "max_norm_err": out.get("max_norm_err_vs_oracle"),
        }
        lead = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"]
        ordered = {k: out[k] for k in lead}
        ordered.update(flat)
        for k in ("roofline", "cpu_baseline"):
            if k in out:
                ordered[k] = out[k]
        ordered.update({k: v for k, v in out.items() if k not in ordered})
        out = ordered
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)  # whatever the teardown prints must not follow the JSON line
    if group is not None:
        group.barrier()
        group.close()
    if filectl is not None:
        filectl.cleanup()
    if hung_thread:  # a communicator call that never returned would block interpreter shutdown
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
