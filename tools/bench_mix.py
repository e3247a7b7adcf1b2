"""The iSTFT / FIR kernels beside their no-math traffic models (tools/diag_mix.hip) in ONE process, interleaved rounds: which part of
the distance to the roofline is the access pattern's and which the math's.  usage: python tools/bench_mix.py [rounds=3] [stft istft fir]"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S
from nx_signal_amd import _lib
import bench

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = S.Context(0); lib = _lib.load(); diag = bench.load_diag()
diag.nxdiag_istft_mix2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int]
diag.nxdiag_fir_mix2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int]
stream = C.c_void_p(lib.nxsig_get_stream(ctx.handle))
rng = np.random.Generator(np.random.PCG64(7))
N, HOP, SR, L = 1024, 256, 48000, 48000 * 60
M = (L - N) // HOP + 1

def timeit(fn, reps=20, warm=10):
    for _ in range(warm): fn()
    ctx.sync(); ctx.timer_lap()
    for _ in range(reps):
        fn(); ctx.timer_lap()
    return float(np.mean(ctx.timer_laps()))

# ---- stft headline, 32 x 60 s
if "stft" in (sys.argv[2:] or ["stft", "istft", "fir"]):
    diag.nxdiag_stft_mix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int]
    Bs = 32
    xs = ctx.empty((Bs, L), np.float32)
    chunk0 = rng.standard_normal(L, dtype=np.float32)
    for r in range(Bs):
        xr = np.roll(chunk0, 977 * r)
        _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(xs.ptr + r * L * 4), xr.ctypes.data_as(C.c_void_p), xr.nbytes))
    zs = ctx.empty((Bs, M, N), np.complex64)
    tab = ctx.to_device(rng.standard_normal(3072).astype(np.float32))
    ws = S.windows.hann(N); ps = _lib.StftParams(N, HOP, N, 0, 0, 0, _lib.SCALE_NONE, 0, float(SR))
    nbs = Bs * M * 9216
    cases = {"stft kernel (headline)": lambda: _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xs.ptr), L, Bs, L, ws.ctypes.data_as(C.c_void_p), C.byref(ps), C.c_void_p(zs.ptr), None, 1))}
    for ppw in (1, 2, 3, 4):
        cases[f"stft mix {ppw} pairs/wave"] = (lambda ppw=ppw: diag.nxdiag_stft_mix(stream, C.c_void_p(xs.ptr), C.c_void_p(zs.ptr), C.c_void_p(tab.ptr), Bs, L, HOP, ppw))
    res = {k: [] for k in cases}
    for r in range(rounds):
        for k, fn in cases.items():
            res[k].append(nbs / (timeit(fn) * 1e-3) / 1e9)
    for k, v in res.items():
        print(json.dumps({"case": k, "GBps": [round(a, 1) for a in v], "frac_of_8TBps": round(float(np.median(v)) / 8000, 4)}), flush=True)
    for b in (xs, zs, tab): b.free()

# ---- stft N = 2048 hop 512, config 4's shard
if "stft2048" in (sys.argv[2:] or []):
    diag.nxdiag_stft2048_mix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int]
    B4, L4, N4, H4 = 8, SR * 600, 2048, 512
    M4 = (L4 - N4) // H4 + 1
    xs = ctx.empty((B4, L4), np.float32)
    chunk0 = rng.standard_normal(L4, dtype=np.float32)
    for r in range(B4):
        xr = np.roll(chunk0, 977 * r)
        _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(xs.ptr + r * L4 * 4), xr.ctypes.data_as(C.c_void_p), xr.nbytes))
    zs = ctx.empty((B4, M4, N4), np.complex64)
    tab = ctx.to_device(rng.standard_normal(3072).astype(np.float32))
    ws = S.windows.hann(N4); ps = _lib.StftParams(N4, H4, N4, 0, 0, 0, _lib.SCALE_NONE, 0, float(SR))
    nbs = B4 * M4 * (H4 * 4 + N4 * 8)
    cases = {"stft N=2048 kernel (config 4)": lambda: _lib.check(lib.nxsig_stft_f32(ctx.handle, C.c_void_p(xs.ptr), L4, B4, L4, ws.ctypes.data_as(C.c_void_p), C.byref(ps), C.c_void_p(zs.ptr), None, 1))}
    for upw in (1, 2, 4, 8, 12):
        cases[f"stft2048 mix {upw} frames/wave"] = (lambda upw=upw: diag.nxdiag_stft2048_mix(stream, C.c_void_p(xs.ptr), C.c_void_p(zs.ptr), C.c_void_p(tab.ptr), B4, L4, H4, upw))
    res = {k: [] for k in cases}
    for r in range(rounds):
        for k, fn in cases.items():
            res[k].append(nbs / (timeit(fn, 10, 5) * 1e-3) / 1e9)
    for k, v in res.items():
        print(json.dumps({"case": k, "GBps": [round(a, 1) for a in v], "frac_of_8TBps": round(float(np.median(v)) / 8000, 4)}), flush=True)
    for b in (xs, zs, tab): b.free()
    sys.exit(0)

# ---- istft, config 3
B = 16
w = S.windows.hann(N)
x = ctx.to_device(rng.standard_normal((B, L), dtype=np.float32))
z, _, _ = S.stft(x, w, ctx=ctx, overlap_length=N - HOP, fft_length=N, sampling_rate=SR)
y = ctx.empty((B, M * HOP + N - HOP), np.complex64)
p = _lib.StftParams(N, HOP, N, 0, 0, 0, _lib.SCALE_NONE, 0, float(SR))
wp = w.ctypes.data_as(C.c_void_p)
nb = B * M * 10240
cases = {"istft kernel": lambda: _lib.check(lib.nxsig_istft_c64(ctx.handle, C.c_void_p(z.ptr), M, B, wp, C.byref(p), C.c_void_p(y.ptr), 1))}
for wpc in (8, 12, 16):
    for lb in (8, 16, 108, 116):
        for halo in (3, 0):
            cases[f"istft mix {wpc} runs/CU {lb % 100:2d}-B {'default-policy' if lb > 100 else 'nt'} loads halo {halo}"] = (lambda wpc=wpc, lb=lb, halo=halo: diag.nxdiag_istft_mix2(stream, C.c_void_p(z.ptr), C.c_void_p(y.ptr), B * M, wpc, halo, lb))
diag.nxdiag_istft_mix3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int]
for rl in (2, 4, 8, 16, 32, 64):
    for halo in (3, 0):
        for lb in (8, 16):
            cases[f"istft mix short-lived workgroups, {rl} frames/wave {lb:2d}-B nt loads halo {halo}"] = (lambda rl=rl, halo=halo, lb=lb: diag.nxdiag_istft_mix3(stream, C.c_void_p(z.ptr), C.c_void_p(y.ptr), B * M, rl, halo, lb))
diag.nxdiag_istft_mix4.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int]
for wpc in (8, 12):
    for wb in (0, 1, 4, 8):
        cases[f"istft mix {wpc} runs/CU halo 0, stores in bursts of {wb} frames" if wb else f"istft mix {wpc} runs/CU halo 0, NO stores (8 KiB per frame counted)"] = (lambda wpc=wpc, wb=wb: diag.nxdiag_istft_mix4(stream, C.c_void_p(z.ptr), C.c_void_p(y.ptr), B * M, wpc, 0, wb))
diag.nxdiag_istft_mix_pol.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int]
def polname(a): return "+".join(n for b, n in ((1, "sc0"), (16, "sc1"), (2, "nt")) if a & b) or "default"
for la, sa in ((2, 0), (2, 2), (2, 16), (2, 18), (2, 1), (2, 17), (2, 19), (2, 3), (0, 2), (0, 18), (16, 18), (18, 18), (0, 0), (1, 18), (17, 18), (3, 18)):
    cases[f"istft mix 8 runs/CU halo 0 policy: loads {polname(la)}, stores {polname(sa)}"] = (lambda la=la, sa=sa: _lib.check(diag.nxdiag_istft_mix_pol(stream, C.c_void_p(z.ptr), C.c_void_p(y.ptr), B * M, 8, la, sa)))
if "istft-policy" in sys.argv[2:]:
    cases = {k: v for k, v in cases.items() if "policy" in k or " 8 runs/CU  8-B nt" in k or k == "istft kernel"}
if "istft-bursts" in sys.argv[2:]:
    cases = {k: v for k, v in cases.items() if "bursts" in k or "NO stores" in k or " 8 runs/CU  8-B nt" in k or k == "istft kernel"}
if "istft-short" in sys.argv[2:]:
    cases = {k: v for k, v in cases.items() if "short-lived" in k or " 8 runs/CU  8-B nt" in k or k == "istft kernel"}
res = {k: [] for k in cases}
for r in range(rounds):
    for k, fn in cases.items():
        res[k].append((B * M * 8192 if "NO stores" in k else nb) / (timeit(fn) * 1e-3) / 1e9)
for k, v in res.items():
    print(json.dumps({"case": k, "GBps": [round(a, 1) for a in v], "frac_of_8TBps": round(float(np.median(v)) / 8000, 4)}), flush=True)
if "only-istft" in sys.argv[2:]: sys.exit(0)
# the same streams over CONSTANT data (zeros): what part of the ceiling is the data's (bus toggling / power), not the pattern's
zero = np.zeros(1 << 22, np.uint8)
for off in range(0, B * M * N * 8, zero.nbytes):
    nbz = min(zero.nbytes, B * M * N * 8 - off)
    _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(z.ptr + off), zero.ctypes.data_as(C.c_void_p), nbz))
res = {k: [] for k in cases if " 8 runs/CU" in k or k == "istft kernel"}
for r in range(rounds):
    for k in res:
        res[k].append(nb / (timeit(cases[k]) * 1e-3) / 1e9)
for k, v in res.items():
    print(json.dumps({"case": k + "  [zero-filled spectrum]", "GBps": [round(a, 1) for a in v], "frac_of_8TBps": round(float(np.median(v)) / 8000, 4)}), flush=True)
for b in (x, z, y): b.free()

# ---- fir, config 5
B4, L4 = 8, SR * 600
x4 = ctx.empty((B4, L4), np.float32)
chunk = rng.standard_normal(L4, dtype=np.float32)
for r in range(B4):
    xr = np.roll(chunk, 977 * r)
    _lib.check(lib.nxsig_upload(ctx.handle, C.c_void_p(x4.ptr + r * L4 * 4), xr.ctypes.data_as(C.c_void_p), xr.nbytes))
y5 = ctx.empty((B4, L4), np.float32)
h = S.filters.firwin(257, [4000.0], sampling_rate=float(SR)); hp = h.ctypes.data_as(C.c_void_p)
nb = B4 * L4 * 8
cases = {"fir kernel": lambda: _lib.check(lib.nxsig_fir_f32(ctx.handle, C.c_void_p(x4.ptr), L4, B4, L4, hp, 257, _lib.CONV_SAME, C.c_void_p(y5.ptr), 1))}
for ppw in (1, 2, 3, 4, 6, 8, 16):
    for wide in (0, 1):
        cases[f"fir mix {ppw} pairs/wave {'16' if wide else ' 8'}-B accesses"] = (lambda ppw=ppw, wide=wide: diag.nxdiag_fir_mix2(stream, C.c_void_p(x4.ptr), C.c_void_p(y5.ptr), B4, L4, ppw, wide))
res = {k: [] for k in cases}
for r in range(rounds):
    for k, fn in cases.items():
        res[k].append(nb / (timeit(fn, 10, 5) * 1e-3) / 1e9)
for k, v in res.items():
    print(json.dumps({"case": k, "GBps": [round(a, 1) for a in v], "frac_of_8TBps": round(float(np.median(v)) / 8000, 4)}), flush=True)
