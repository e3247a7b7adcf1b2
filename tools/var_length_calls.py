"""Per-call host cost when the length changes from call to call (tools only): istft with 128 distinct frame counts, stft / fir with 128
distinct signal lengths, against repeated calls of one shape.  Measured (round 5): istft 26 us per call at a fixed M, 54 us the first
time an M is seen (edge candidate list + table upload), 32 us afterwards; stft 21 / 20 us; fir 34 / 47 us."""
import time, numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nx_signal_amd as S
ctx = S.Context(0)
N, hop, rows = 1024, 256, 16
w = S.windows.hann(N)
rng = np.random.default_rng(0)
z = (rng.standard_normal((rows, 400, N)) + 1j * rng.standard_normal((rows, 400, N))).astype(np.complex64)
zd = {M: ctx.to_device(np.ascontiguousarray(z[:, :M])) for M in range(200, 328)}
def run(Ms, label):
    ctx.sync(); t0 = time.perf_counter()
    for M in Ms:
        y = S.istft(zd[M], w, overlap_length=N - hop, fft_length=N, sampling_rate=16000, ctx=ctx)
    ctx.sync(); dt = (time.perf_counter() - t0) / len(Ms)
    print(label, round(dt * 1e6, 1), "us per call")
run([200] * 64, "warm same M")
run([200] * 128, "same M")
run(list(range(200, 328)), "128 distinct M, first time")
run(list(range(200, 328)), "128 distinct M, second time")
# forward with varying L
x = rng.standard_normal((rows, 90000)).astype(np.float32)
xd = {L: ctx.to_device(np.ascontiguousarray(x[:, :L])) for L in range(80000, 80128)}
def runf(Ls, label):
    ctx.sync(); t0 = time.perf_counter()
    for L in Ls:
        zz = S.stft(xd[L], w, overlap_length=N - hop, fft_length=N, sampling_rate=16000, ctx=ctx)
    ctx.sync(); dt = (time.perf_counter() - t0) / len(Ls)
    print(label, round(dt * 1e6, 1), "us per call")
runf([80000] * 64, "stft warm")
runf([80000] * 128, "stft same L")
runf(list(range(80000, 80128)), "stft 128 distinct L")
h = S.filters.firwin(257, [0.2])
def runfir(Ls, label):
    ctx.sync(); t0 = time.perf_counter()
    for L in Ls:
        yy = S.filters.fir(xd[L], h, mode="same", ctx=ctx)
    ctx.sync(); dt = (time.perf_counter() - t0) / len(Ls)
    print(label, round(dt * 1e6, 1), "us per call")
runfir([80000] * 64, "fir warm")
runfir([80000] * 128, "fir same L")
runfir(list(range(80000, 80128)), "fir 128 distinct L")
