"""
CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  nx_signal_amd/ (the product) never does; it fails loudly without the HIP library.

What this is
------------
A numpy restatement of the reference's STFT / iSTFT / windows / FIR path
(elixir-nx/nx_signal v0.3.0, /root/reference) *as evaluated by Nx.BinaryBackend*
(hex package nx 0.11.0, mix.lock:10 — not vendored in the reference, so its numeric rules
are restated here from SURVEY.md Appendix A and pinned against the reference's own doctest
and unit-test vectors, see tests/golden/reference_vectors.json and tests/test_oracle_golden.py).

Numeric model:
  * tensors are f32 / c64; every elementwise op is computed in IEEE double on the
    f32-decoded operands and rounded ONCE to the output type.  For + - * / this is
    identical to native f32 arithmetic, so numpy float32 ops reproduce it.
  * transcendental functions: libm double on the f32 input, then round to f32.
  * Elixir-number (op) Elixir-number folds in double at trace time; a number that meets
    an f32 tensor becomes an f32 constant first.
  * Nx.fft / Nx.ifft: per row, in double complex, zero-pad/truncate to `length`,
    (components with |x| <= eps=1e-10 zeroed), rounded to c64.
  * Nx.sum / Nx.dot / Nx.indexed_add: accumulate in double, round once.

Parity status: PINNED at toy sizes by the reference's doctests (bit-exact, see the golden
test); at N=1024/2048 the reference holds no vector, so large-size parity is pinned only
through this model ("parity unpinned by the reference's own tests at N>=1024").
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
f64 = np.float64
c64 = np.complex64
c128 = np.complex128

FFT_EPS = 1.0e-10  # Nx.fft :eps default [Nx, recalled]; unobservable in every pinned vector


# --------------------------------------------------------------------------------------
# helpers implementing the Appendix A rules
# --------------------------------------------------------------------------------------
def _t64(x):
    """transcendental helper input: decode f32 -> double."""
    return np.asarray(x, dtype=f32).astype(f64)


def _cos32(x):
    return np.cos(_t64(x)).astype(f32)


def _sin32(x):
    return np.sin(_t64(x)).astype(f32)


def _sqrt32(x):
    return np.sqrt(_t64(x)).astype(f32)


def _exp32(x):
    return np.exp(_t64(x)).astype(f32)


def _log32(x):
    return np.log(_t64(x)).astype(f32)


def _pow32(x, p):
    """Nx.pow(f32, number): :math.pow in double, one rounding."""
    return np.power(_t64(x), float(p)).astype(f32)


def _sum32(x):
    """Nx.sum of an f32 tensor: accumulate in double, round once (rule 6)."""
    return f32(math.fsum(np.asarray(x, dtype=f32).astype(f64).tolist()))


def _iota32(n):
    return np.arange(n, dtype=f32)


def linspace32(start, stop, n, endpoint=True):
    """Nx.linspace in f32 (rule 5): iota * step + start, step = (stop-start)/(n-1 | n)."""
    start = f32(start)
    stop = f32(stop)
    div = (n - 1) if endpoint else n
    with np.errstate(divide="ignore", invalid="ignore"):
        step = f32(stop - start) / f32(div)
        return (_iota32(n) * step + start).astype(f32)


# --------------------------------------------------------------------------------------
# NxSignal.Windows  (lib/nx_signal/windows.ex)
# --------------------------------------------------------------------------------------
def rectangular(n, dtype=np.int64):
    """windows.ex:33-36 — default type s64."""
    return np.ones(n, dtype=dtype)


def bartlett(n):
    """windows.ex:57-78."""
    n_on_2 = n // 2
    left_size = n_on_2 + n % 2
    left_idx = _iota32(left_size)
    right_idx = _iota32(n_on_2) + f32(left_size)
    nf = f32(n)
    left = left_idx * f32(2) / nf
    right = f32(2) - right_idx * f32(2) / nf
    return np.concatenate([left, right]).astype(f32)


def triangular(n):
    """windows.ex:98-126."""
    n_on_2 = (n + 1) // 2
    idx = _iota32(n_on_2) + f32(1)
    if n % 2 == 1:
        left = idx * f32(2) / f32(n + 1)
        return np.concatenate([left, left[::-1][1:]]).astype(f32)
    left = (f32(2) * idx - f32(1)) / f32(n)
    return np.concatenate([left, left[::-1]]).astype(f32)


def _cos_term(k, mult, lm1):
    """Nx.cos(mult_pi * n / (l - 1)) with `mult * @pi` folded in double -> f32 constant."""
    c = f32(mult * math.pi)
    with np.errstate(divide="ignore", invalid="ignore"):
        ang = (c * k) / f32(lm1)
    return _cos32(ang)


def blackman(n, is_periodic=True):
    """windows.ex:160-202: 0.42 - 0.5*cos(2 pi n/(l-1)) + 0.08*cos(4 pi n/(l-1)), half built then mirrored."""
    l = n + 1 if is_periodic else n
    m = -(-l // 2)
    k = _iota32(m)
    left = (f32(0.42) - f32(0.5) * _cos_term(k, 2, l - 1)) + f32(0.08) * _cos_term(k, 4, l - 1)
    left = left.astype(f32)
    if l % 2 == 0:
        w = np.concatenate([left, left[::-1]])
    else:
        w = np.concatenate([left, left[::-1][1:]])
    if is_periodic:
        w = w[:-1]
    return w.astype(f32)


def hamming(n, is_periodic=True):
    """windows.ex:225-250: 0.54 - 0.46*cos(2 pi n/(l-1))."""
    l = n + 1 if is_periodic else n
    k = _iota32(l)
    w = f32(0.54) - f32(0.46) * _cos_term(k, 2, l - 1)
    return (w[: l - 1] if is_periodic else w).astype(f32)


def hann(n, is_periodic=True):
    """windows.ex:278-305: 0.5*(1 - cos(2 pi n/(l-1)))."""
    l = n + 1 if is_periodic else n
    k = _iota32(l)
    w = f32(0.5) * (f32(1) - _cos_term(k, 2, l - 1))
    return (w[: l - 1] if is_periodic else w).astype(f32)


def _kaiser_i0(x):
    """windows.ex:371-386 (polynomial / asymptotic I0), op-by-op in f32."""
    ax = np.abs(np.asarray(x, dtype=f32))
    small = (
        f32(1)
        + _pow32(ax, 2) / f32(4)
        + _pow32(ax, 4) / f32(64)
        + _pow32(ax, 6) / f32(2304)
        + _pow32(ax, 8) / f32(147456)
    ).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        two_pi = f32(2) * f32(math.pi)  # 2 * Nx.Constants.pi() (f32 tensor)
        # EMPIRICAL PIN: the bracket (1 + 1/(8x) + 9/(128 x^2)) reproduces the reference's three kaiser
        # doctests (windows.ex:322-338) bit-for-bit only when evaluated in double and rounded once;
        # rounding each of its ops to f32 leaves kaiser(4, periodic)[1] two ulp low.  Every other
        # window doctest is reproduced with strict per-op f32 rounding.
        a64 = ax.astype(f64)
        bracket = (1.0 + 1.0 / (8.0 * a64) + 9.0 / (128.0 * a64 * a64)).astype(f32)
        large = (_exp32(ax) / _sqrt32(two_pi * ax) * bracket).astype(f32)
    return np.where(ax < f32(3.75), small, large).astype(f32)


def kaiser(n, beta=12.0, is_periodic=True, eps=1.0e-7):
    """windows.ex:341-369."""
    wl = n + 1 if is_periodic else n
    ratio = linspace32(-1, 1, wl, endpoint=True)
    sqrt_arg = np.maximum(f32(1) - _pow32(ratio, 2), f32(eps)).astype(f32)
    r = f32(beta) * _sqrt32(sqrt_arg)
    w = (_kaiser_i0(r) / _kaiser_i0(f32(beta))).astype(f32)
    return (w[:n] if is_periodic else w).astype(f32)


# --------------------------------------------------------------------------------------
# NxSignal.Waveforms.sinc  (lib/nx_signal/waveforms.ex:451-457)
# --------------------------------------------------------------------------------------
def sinc(t):
    t = np.asarray(t, dtype=f32) * f32(math.pi)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = _sin32(t) / t
    return np.where(t == 0, f32(1), s).astype(f32)


# --------------------------------------------------------------------------------------
# NxSignal.Filters.firwin  (lib/nx_signal/filters.ex:147-279)
# --------------------------------------------------------------------------------------
def firwin(num_taps, cutoff, window="hamming", pass_zero=True, scale=True, sampling_rate=2.0):
    if not isinstance(cutoff, (list, tuple)):
        raise ValueError(f"cutoff must be a list of frequencies, got: {cutoff!r}")  # filters.ex:160-162
    nyq = sampling_rate / 2.0
    cl = sorted(c / nyq for c in cutoff)  # filters.ex:164
    if cl[0] <= 0.0:
        raise ValueError(f"cutoff must be strictly between 0 and Nyquist (exclusive), got: {cl[0] * nyq}")
    if cl[-1] >= 1.0:
        raise ValueError(f"cutoff must be strictly between 0 and Nyquist (exclusive), got: {cl[-1] * nyq}")
    even_n_cuts = len(cl) % 2 == 0
    nyquist_gain = (pass_zero and even_n_cuts) or ((not pass_zero) and (not even_n_cuts))
    if nyquist_gain and num_taps % 2 == 0:
        raise ValueError(
            "a filter with non-zero gain at Nyquist (e.g. highpass) requires "
            f"an odd number of taps, got: {num_taps}"
        )
    m = (num_taps - 1) / 2.0
    alpha = (_iota32(num_taps) - f32(m)).astype(f32)  # filters.ex:195-196
    all_freqs = [0.0] + cl + [1.0]
    pairs = [(all_freqs[i], all_freqs[i + 1]) for i in range(len(all_freqs) - 1)]
    h = np.zeros(num_taps, dtype=f32)
    for i, (a, b) in enumerate(pairs):
        if (i % 2 == 0) if pass_zero else (i % 2 == 1):
            # firwin_contribution, filters.ex:223-227: acc + b*sinc(b*alpha) - a*sinc(a*alpha)
            ca = f32(a) * sinc(f32(a) * alpha)
            cb = f32(b) * sinc(f32(b) * alpha)
            h = ((h + cb) - ca).astype(f32)
    w = _firwin_window(num_taps, window)
    h = (h * w).astype(f32)
    if not scale:
        return h
    if pass_zero:
        sf = 0.0
    elif len(cl) == 1:
        sf = 1.0
    else:
        sf = (cl[0] + cl[1]) / 2.0
    c = _cos32(alpha * f32(math.pi * sf))
    dot = f32(math.fsum((h.astype(f64) * c.astype(f64)).tolist()))  # Nx.dot: double accumulate
    return (h / np.abs(dot)).astype(f32)


def _firwin_window(num_taps, window):
    """filters.ex:254-279."""
    if window == "hamming":
        return hamming(num_taps, is_periodic=False)
    if window == "hann":
        return hann(num_taps, is_periodic=False)
    if window == "blackman":
        return blackman(num_taps, is_periodic=False)
    if window == "bartlett":
        return bartlett(num_taps)
    if window == "rectangular":
        return rectangular(num_taps, dtype=f32)
    if isinstance(window, tuple) and len(window) == 2 and window[0] == "kaiser":
        return kaiser(num_taps, beta=window[1], is_periodic=False)
    raise ValueError(
        f"unknown window {window!r}, supported: :hamming, :hann, :blackman, :bartlett, :rectangular, {{:kaiser, beta}}"
    )


# --------------------------------------------------------------------------------------
# Nx.fft / Nx.ifft (BinaryBackend model, Appendix A rule 7)
# --------------------------------------------------------------------------------------
def next_pow2(n):
    p = 1
    while p < n:
        p *= 2
    return p


def _resolve_len(fft_length, n):
    if fft_length in (None, "power_of_two"):
        return next_pow2(n)
    return int(fft_length)


def _eps_clean(z, eps):
    if eps is None or eps <= 0:
        return z
    re = np.where(np.abs(z.real) <= eps, 0.0, z.real)
    im = np.where(np.abs(z.imag) <= eps, 0.0, z.imag)
    return re + 1j * im


def fft(x, length=None, eps=FFT_EPS):
    """Nx.fft(x, length:) over the last axis.  f32/c64 (or integer) in -> c64 out."""
    x = np.asarray(x)
    xin = x.astype(c64).astype(c128) if np.iscomplexobj(x) else x.astype(f32).astype(f64)
    n = x.shape[-1]
    k = n if length is None else _resolve_len(length, n)
    z = np.fft.fft(xin, n=k, axis=-1)
    return np.ascontiguousarray(_eps_clean(z, eps).astype(c64))


def ifft(x, length=None, eps=FFT_EPS):
    x = np.asarray(x)
    xin = x.astype(c64).astype(c128) if np.iscomplexobj(x) else x.astype(f32).astype(f64)
    n = x.shape[-1]
    k = n if length is None else _resolve_len(length, n)
    z = np.fft.ifft(xin, n=k, axis=-1)
    return np.ascontiguousarray(_eps_clean(z, eps).astype(c64))


# --------------------------------------------------------------------------------------
# NxSignal.as_windowed  (lib/nx_signal.ex:249-364)
# --------------------------------------------------------------------------------------
def _pad_config(L, N, padding):
    """returns ('reflect'|'zeros', lo, hi) following nx_signal.ex:257-331."""
    if padding == "reflect":
        return "reflect", N // 2, N // 2  # :262, :348-349
    if padding == "valid":
        return "zeros", 0, 0
    if padding == "same":
        tot = max(L - 1 + N - L, 0)  # :310
        return "zeros", tot // 2, tot - tot // 2  # floor / ceil :311
    if isinstance(padding, (list, tuple)):
        if len(padding) != 1 or len(padding[0]) != 2 or not all(isinstance(v, (int, np.integer)) for v in padding[0]):
            raise ValueError(
                "padding must be a list of {high, low} tuples, where each element is an integer. " f"Got: {padding!r}"
            )
        return "zeros", int(padding[0][0]), int(padding[0][1])
    raise ValueError(
        "invalid padding mode specified, padding must be one of :valid, :same, or a padding configuration, "
        f"got: {padding!r}"
    )


def reflect_index(i, L):
    """Nx.reflect index map: mirror without repeating the edge sample, periodic 2(L-1)."""
    if L == 1:
        return 0
    p = 2 * (L - 1)
    i = i % p
    return i if i < L else p - i


def num_frames(L, N, stride, padding="valid"):
    _, lo, hi = _pad_config(L, N, padding)
    Lp = L + lo + hi
    if Lp < N:
        raise ValueError("window is larger than the (padded) signal")
    return (Lp - N) // stride + 1


def pad_signal(x, N, padding):
    x = np.asarray(x)
    L = x.shape[-1]
    mode, lo, hi = _pad_config(L, N, padding)
    if mode == "reflect":
        idx = np.array([reflect_index(i - lo, L) for i in range(L + lo + hi)], dtype=np.int64)
        return x[..., idx]
    # Nx.pad with possibly negative lo/hi (crop)
    out = x
    if lo < 0:
        out = out[..., -lo:]
        lo = 0
    if hi < 0:
        out = out[..., :hi]
        hi = 0
    pw = [(0, 0)] * (out.ndim - 1) + [(lo, hi)]
    return np.pad(out, pw)


def as_windowed(x, window_length, stride=1, padding="valid"):
    if not (isinstance(stride, (int, np.integer)) and stride >= 1):
        raise ValueError(f"expected an integer >= 1 or a list of integers, got: {stride!r}")  # :282-284
    x = np.asarray(x)
    xp = pad_signal(x, window_length, padding)
    Lp = xp.shape[-1]
    if Lp < window_length:
        raise ValueError("window is larger than the (padded) signal")
    M = (Lp - window_length) // stride + 1
    idx = np.arange(M)[:, None] * stride + np.arange(window_length)[None, :]
    return xp[..., idx]


# --------------------------------------------------------------------------------------
# NxSignal.stft  (lib/nx_signal.ex:68-130)
# --------------------------------------------------------------------------------------
def _scale_factor(window, scaling, sampling_rate):
    """f32 scalar the spectrum is divided by (stft) / multiplied by (istft); None for nil."""
    w = np.asarray(window, dtype=f32)
    if scaling is None:
        return None
    if scaling == "spectrum":
        return _sum32(w)  # :116, :614
    if scaling == "psd":
        s2 = _sum32((w * w).astype(f32))  # window ** 2 (pow in double == exact product, rounded)
        return f32(np.sqrt(f64(f32(f32(sampling_rate) * s2))))  # :119, :617
    raise ValueError(f"invalid :scaling, expected one of :spectrum, :psd or nil, got: {scaling!r}")


def fft_frequencies(sampling_rate, fft_length, endpoint=False):
    """nx_signal.ex:154-166."""
    step = f32(f32(sampling_rate) / f32(fft_length))
    return linspace32(0, step * f32(fft_length), fft_length, endpoint=endpoint)


def stft_times(frame_length, sampling_rate, M):
    """nx_signal.ex:109-111 (spacing N/(2 fs) regardless of hop — SURVEY B4)."""
    time_step = f32(f64(frame_length) / f64(f32(f32(2) * f32(sampling_rate))))
    last = f32(time_step * f32(M))
    return linspace32(time_step, last, M, endpoint=True)


def stft(
    data,
    window,
    overlap_length=None,
    fft_length="power_of_two",
    window_padding="valid",
    sampling_rate=100,
    scaling=None,
    eps=FFT_EPS,
):
    """Returns (z c64[..., M, K], times f32[M], freqs f32[K])."""
    window = np.asarray(window)
    N = window.shape[0]
    if overlap_length is None:
        overlap_length = N // 2  # :83
    hop = N - overlap_length
    data = np.asarray(data)
    frames = as_windowed(data, N, hop, window_padding)  # :94-100
    if np.iscomplexobj(frames):
        # c64 samples (IQ data): the reference frames, multiplies and transforms whatever tensor it is given (:94-102).  c64 x f32 is
        # componentwise (SURVEY App. A rule 9: each component an exact f32 product), the transform is the same Nx.fft over c64 rows
        fc = frames.astype(c64)
        wf = window.astype(f32)
        fr = ((fc.real.astype(f32) * wf) + 1j * (fc.imag.astype(f32) * wf)).astype(c64)  # :101
        z = fft(fr, length=fft_length, eps=eps)  # :102
    else:
        fr = frames.astype(f32) * window.astype(f32)  # :101 exact f32 product
        z = fft(fr.astype(f32), length=fft_length, eps=eps)  # :102
    K = z.shape[-1]
    M = z.shape[-2]
    sf = _scale_factor(window, scaling, sampling_rate)
    if sf is not None:
        z = (z.real.astype(f32) / sf + 1j * (z.imag.astype(f32) / sf)).astype(c64)  # :116/:119 componentwise
    return z, stft_times(N, sampling_rate, M), fft_frequencies(sampling_rate, K)


# --------------------------------------------------------------------------------------
# NxSignal.overlap_and_add / istft  (lib/nx_signal.ex:582-638, 684-736)
# --------------------------------------------------------------------------------------
def overlap_and_add(t, overlap_length, dtype=None):
    t = np.asarray(t)
    M, N = t.shape[-2], t.shape[-1]
    if overlap_length >= N:
        raise ValueError(
            f"overlap_length must be a number less than the window size {N}, got: {N}"
        )  # :692-695 (message quirk B10)
    stride = N - overlap_length
    out_len = M * stride + overlap_length
    lead = t.shape[:-2]
    acc_t = c128 if np.iscomplexobj(t) else f64
    tt = t.reshape((-1, M, N)).astype(acc_t)
    out = np.zeros((tt.shape[0], out_len), dtype=acc_t)
    for m in range(M):  # indexed_add: contributions summed in double, rounded once (rule 8)
        out[:, m * stride : m * stride + N] += tt[:, m, :]
    out = out.reshape(lead + (out_len,))
    return out.astype(dtype if dtype is not None else t.dtype)


def istft(z, window, overlap_length=None, fft_length=None, sampling_rate=1000, scaling=None, eps=FFT_EPS):
    """Returns c64[..., M*hop + overlap]  (SURVEY B8)."""
    window = np.asarray(window, dtype=f32)
    N = window.shape[0]
    if overlap_length is None:
        overlap_length = N // 2  # :594-601
    if scaling == "psd" and sampling_rate is None:
        raise ValueError(":sampling_rate is mandatory if scaling is :psd")
    z = np.asarray(z).astype(c64)
    K = _resolve_len(fft_length, z.shape[-1])
    frames = ifft(z, length=K, eps=eps)  # :609
    sf = _scale_factor(window, scaling, sampling_rate)
    re, im = frames.real.astype(f32), frames.imag.astype(f32)
    if sf is not None:
        re, im = re * sf, im * sf  # :614/:617
    if K != N:
        raise ValueError("istft requires fft_length == window length (broadcast {M,K} x {N})")
    re, im = (re * window).astype(f32), (im * window).astype(f32)  # :628
    num = overlap_and_add(re.astype(f64) + 1j * im.astype(f64), overlap_length, dtype=c64)
    w2 = _pow32(np.abs(window), 2)  # Nx.abs(window) ** 2
    den = overlap_and_add(np.broadcast_to(w2, z.shape[:-1] + (N,)), overlap_length, dtype=f32)  # :630-633
    den = np.where(den > f32(1.0e-10), den, f32(1.0)).astype(f32)  # :635
    out = (num.real.astype(f32) / den + 1j * (num.imag.astype(f32) / den)).astype(c64)  # :637
    return out


# --------------------------------------------------------------------------------------
# NxSignal.Convolution.fftconvolve (n-D)  (lib/nx_signal/convolution.ex:252-347)
# --------------------------------------------------------------------------------------
def fftconvolve(in1, in2, mode="full", eps=FFT_EPS):
    """n-D: fft_nd of both operands over the axes where neither dimension is 1 with lengths s1 + s2 - 1 (:258-278), broadcast
    product (c64 x c64 in double, rounded once), ifft_nd over the same axes (:284), Nx.real for real operands (:286-291),
    apply_mode / centered (:300-347)."""
    if mode not in ("full", "same", "valid"):
        raise ValueError(f"expected mode to be one of [:full, :same, :valid], got: {mode!r}")
    a = np.asarray(in1)
    b = np.asarray(in2)
    if a.ndim != b.ndim:
        raise ValueError("Rank of in1 and in2 must be equal.")
    s1, s2 = list(a.shape), list(b.shape)
    lengths_all = [x + y - 1 for x, y in zip(s1, s2)]
    axes = [i for i, (x, y) in enumerate(zip(s1, s2)) if x != 1 and y != 1]  # :265-274
    lengths = [lengths_all[i] for i in axes]
    is_c = np.iscomplexobj(a) or np.iscomplexobj(b)
    if axes:
        sp1 = fft_nd(a, axes=axes, lengths=lengths)
        sp2 = fft_nd(b, axes=axes, lengths=lengths)
        c = (sp1.astype(c128) * sp2.astype(c128)).astype(c64)  # c64 * c64 in double, rounded once per component
        out = fft_nd(c, axes=axes, lengths=[None] * len(axes), inverse=True)
    else:
        # no FFT axis: broadcasting product of the inputs
        out = (a.astype(c128) * b.astype(c128)).astype(c64) if is_c else (a.astype(f32) * b.astype(f32)).astype(f32)
    if not is_c and np.iscomplexobj(out):
        out = out.real.astype(f32)  # :286-291
    if mode == "full":
        return np.ascontiguousarray(out)
    if mode == "same":
        new = s1
    else:
        ok1 = all(x >= y for x, y in zip(s1, s2))
        ok2 = all(y >= x for x, y in zip(s1, s2))
        if not (ok1 or ok2):
            raise ValueError("For 'valid' mode, one must be at least as large as the other in every dimension.")
        big, small = (s1, s2) if ok1 else (s2, s1)
        new = [x - y + 1 for x, y in zip(big, small)]
    sl = tuple(slice((cur - n) // 2, (cur - n) // 2 + n) for cur, n in zip(out.shape, new))  # centered, :319-329
    return np.ascontiguousarray(out[sl])


def convolve_direct(in1, in2, mode="full"):
    """NxSignal.Convolution.convolve(method: :direct) — lib/nx_signal/convolution.ex:95-218: Nx.conv of in1 (zero-padded per
    mode, :157-190) with in2 reversed along every axis (:137); :valid makes the larger operand the volume (:120-135).  Each
    output is the sum over the kernel window in row-major order of products formed and accumulated in double (Elixir floats /
    Complex structs in the BinaryBackend), rounded once to f32 / c64.  Scalars (rank 0) multiply (:97-99, :147-150)."""
    if mode not in ("full", "same", "valid"):
        raise ValueError(f"expected mode to be one of [:full, :same, :valid], got: {mode!r}")
    a = np.asarray(in1)
    b = np.asarray(in2)
    if a.ndim != b.ndim:
        if a.ndim == 0 or b.ndim == 0:
            raise ValueError(f"Incompatible ranks: {{{a.ndim}, {b.ndim}}}")
        raise ValueError(f"NxSignal.convolve/3 requires both inputs to have the same rank or one of them to be a scalar, got {a.ndim} and {b.ndim}")
    scalar = a.ndim == 0
    if scalar:
        a, b = a.reshape(1), b.reshape(1)
    if mode == "valid":
        ok1 = all(x >= y for x, y in zip(a.shape, b.shape))
        ok2 = all(x <= y for x, y in zip(a.shape, b.shape))
        if not (ok1 or ok2):
            raise ValueError("For :valid mode, one must be at least as large as the other in every dimension")
        if not ok1:
            a, b = b, a
    is_c = np.iscomplexobj(a) or np.iscomplexobj(b)
    wide = c128 if is_c else f64
    vol = a.astype(c64 if np.iscomplexobj(a) else f32).astype(wide)
    ker = b.astype(c64 if np.iscomplexobj(b) else f32).astype(wide)
    ker = ker[tuple(slice(None, None, -1) for _ in range(ker.ndim))]
    pads = []
    for k in ker.shape:
        if mode == "full":
            pads.append((k - 1, k - 1))
        elif mode == "same":
            pads.append(((k - 1) - (k - 1) // 2, (k - 1) // 2))
        else:
            pads.append((0, 0))
    vp = np.pad(vol, pads)
    oshape = tuple(v - k + 1 for v, k in zip(vp.shape, ker.shape))
    acc = np.zeros(oshape, dtype=wide)
    for j in np.ndindex(*ker.shape):  # row-major over the window, like the backend's weighted sum
        win = vp[tuple(slice(jj, jj + o) for jj, o in zip(j, oshape))]
        kv = ker[j]
        if is_c:  # Complex.multiply then Complex.add: every real operation rounds once (the f32 x f32 products are exact)
            pr = win.real * kv.real - win.imag * kv.imag
            pi = win.real * kv.imag + win.imag * kv.real
            acc = (acc.real + pr) + 1j * (acc.imag + pi)
        else:
            acc = acc + win * kv
    out = acc.astype(c64 if is_c else f32)
    return out.reshape(()) if scalar else np.ascontiguousarray(out)


def direct_convolve_f64(x, h):
    """Independent check (SURVEY §4 direct-vs-FFT pattern): full linear convolution in double."""
    return np.convolve(np.asarray(x, dtype=f64), np.asarray(h, dtype=f64))


# --------------------------------------------------------------------------------------
# NxSignal.mel_filters / stft_to_mel  (lib/nx_signal.ex:397-513) — used here only to pin the
# oracle's stft at K=16 + :reflect through the reference's stft_to_mel doctest (:465-483).
# --------------------------------------------------------------------------------------
def mel_filters(fft_length, mel_bins, sampling_rate, max_mel=3016, f_sp=200 / 3):
    fftfreqs = fft_frequencies(sampling_rate, fft_length)
    mels = linspace32(0, f32(f32(max_mel) / f32(f_sp)), mel_bins + 2, endpoint=True)
    freqs = (f32(f_sp) * mels).astype(f32)
    min_log_hz = 1000
    min_log_mel = f32(f32(min_log_hz) / f32(f_sp))
    logstep = f32(_log32(f32(6.4)) / f32(27))
    log_t = mels >= min_log_mel
    with np.errstate(over="ignore"):
        mel_log = (f32(min_log_hz) * _exp32(logstep * (mels - min_log_mel))).astype(f32)
    mel_f = np.where(log_t, mel_log, freqs).astype(f32)
    fdiff = (mel_f[1:] - mel_f[:-1]).astype(f32)[:, None]
    ramps = (mel_f[:, None] - fftfreqs[None, :]).astype(f32)
    lower = (-ramps[0:mel_bins] / fdiff[0:mel_bins]).astype(f32)
    upper = (ramps[2 : mel_bins + 2] / fdiff[1 : mel_bins + 1]).astype(f32)
    # :erlang.max(0, -0.0) keeps the first operand (+0.0) — the doctest prints 0.0, never -0.0
    weights = (np.maximum(f32(0), np.minimum(lower, upper)) + f32(0)).astype(f32)
    enorm = (f32(2.0) / (mel_f[2 : mel_bins + 2] - mel_f[0:mel_bins])).astype(f32)
    return (weights * enorm[:, None]).astype(f32)


def stft_to_mel(z, sampling_rate, fft_length, mel_bins=128):
    z = np.asarray(z).astype(c64)
    absz = np.hypot(z.real.astype(f64), z.imag.astype(f64)).astype(f32)
    mags = _pow32(absz, 2)
    filt = mel_filters(fft_length, mel_bins, sampling_rate)
    fs = fft_length // 2
    a = mags[..., :fs].astype(f64)
    b = filt[:, :fs].astype(f64)
    mel_spec = np.einsum("mf,bf->mb", a, b).astype(f32)
    clipped = np.maximum(mel_spec, f32(1.0e-10)).astype(f32)
    log_spec = (_log32(clipped) / _log32(f32(10))).astype(f32)
    log_spec = np.maximum(log_spec, f32(log_spec.max() - f32(8))).astype(f32)
    return ((log_spec + f32(4)) / f32(4)).astype(f32)


# --------------------------------------------------------------------------------------
# deterministic synthetic inputs shared by tests / bench: never zero-filled
# --------------------------------------------------------------------------------------
def synth_signal(length, seed=1234, channels=None):
    """N(0,1) fp32 via numpy PCG64 (fixed algorithm, identical bits everywhere numpy 2.x runs)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    shape = (length,) if channels is None else (channels, length)
    return rng.standard_normal(shape, dtype=f32)


def fft_nd(x, axes=(-1,), lengths=None, inverse=False):
    """NxSignal.Transforms.fft_nd / ifft_nd — lib/nx_signal/transforms.ex:5-21: fold Nx.fft / Nx.ifft(axis:, length:)
    over the axes list (each stage rounds to c64 like the BinaryBackend call it stands for)."""
    acc = np.asarray(x)
    lengths = list(lengths) if lengths is not None else [None] * len(axes)
    for ax, ln in zip(axes, lengths):
        moved = np.moveaxis(acc, ax, -1)
        out = ifft(moved, length=ln) if inverse else fft(moved, length=ln)
        acc = np.moveaxis(out, -1, ax)
    return np.ascontiguousarray(acc)


def correlate(a, b, mode="full", method="fft"):
    """NxSignal.Convolution.correlate/3 — lib/nx_signal/convolution.ex:87-93: convolve(in1, conj(reverse(in2)), opts)"""
    k = np.asarray(b)
    k = k[tuple(slice(None, None, -1) for _ in range(k.ndim))]
    if np.iscomplexobj(k):
        k = np.conj(k)
    if method == "direct":
        return convolve_direct(np.asarray(a), np.ascontiguousarray(k), mode=mode)
    return fftconvolve(np.asarray(a), np.ascontiguousarray(k), mode=mode)


# ======================================================================================
# f64 / c128 tier  (the checker of nx_signal_amd's nxsig_*_f64 / _c128 entry points)
# ======================================================================================
# PARITY UNPINNED: the reference holds NO f64 vector for this path (its only f64 tests are of Filters.wiener, outside §8), and
# Nx is absent from /root/reference, so how Nx promotes numbers / f32 tensors that meet an f64 tensor is restated from its
# documented rules, not observed:
#   * a number that meets an f64 tensor becomes an f64 constant; an f32 TENSOR (Nx.Constants.pi(), the float arguments of a
#     defnp, kaiser_bessel_i0(beta) on the number beta) keeps its f32-rounded value and is widened;
#   * Nx.linspace(..., type: f64) is taken as iota * step + start with every op in f64 (SURVEY App. A rule 5);
#   * Nx.fft / Nx.ifft of f64 / c128: double complex per row, the same eps clean-up, no final rounding;
#   * Nx.sum / Nx.indexed_add of f64: sequential double accumulation.
# libm differs between platforms in the last ulp of cos / sin / exp in double, so the generators are compared to 4 ulp.
def _lin64(start, stop, n, endpoint=True):
    div = (n - 1) if endpoint else n
    with np.errstate(divide="ignore", invalid="ignore"):
        step = (f64(stop) - f64(start)) / f64(div)
        return np.arange(n, dtype=f64) * step + f64(start)


def _cos_term64(k, mult, lm1):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.cos((f64(mult * math.pi) * k) / f64(lm1))


def _kaiser_i0_64(x):
    ax = np.abs(np.asarray(x, dtype=f64))
    small = 1.0 + ax ** 2 / 4.0 + ax ** 4 / 64.0 + ax ** 6 / 2304.0 + ax ** 8 / 147456.0
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        two_pi = f64(f32(2) * f32(math.pi))  # 2 * Nx.Constants.pi(): an f32 tensor
        large = np.exp(ax) / np.sqrt(two_pi * ax) * (1.0 + 1.0 / (8.0 * ax) + 9.0 / (128.0 * ax * ax))
    return np.where(ax < 3.75, small, large)


def window_f64(kind, n, is_periodic=True, beta=12.0, eps=1.0e-7):
    """NxSignal.Windows.<kind>(n, type: {:f, 64}) — lib/nx_signal/windows.ex, every op in double"""
    if kind == "rectangular":
        return np.ones(n, dtype=f64)
    if kind == "bartlett":
        n2 = n // 2
        left = n2 + n % 2
        li, ri = np.arange(left, dtype=f64), np.arange(n2, dtype=f64) + f64(left)
        return np.concatenate([li * 2.0 / f64(n), 2.0 - ri * 2.0 / f64(n)])
    if kind == "triangular":
        h = (n + 1) // 2
        idx = np.arange(h, dtype=f64) + 1.0
        if n % 2 == 1:
            left = idx * 2.0 / f64(n + 1)
            return np.concatenate([left, left[::-1][1:]])
        left = (2.0 * idx - 1.0) / f64(n)
        return np.concatenate([left, left[::-1]])
    l = n + 1 if is_periodic else n
    if kind == "blackman":
        m = -(-l // 2)
        k = np.arange(m, dtype=f64)
        left = (0.42 - 0.5 * _cos_term64(k, 2, l - 1)) + 0.08 * _cos_term64(k, 4, l - 1)
        w = np.concatenate([left, left[::-1]]) if l % 2 == 0 else np.concatenate([left, left[::-1][1:]])
        return w[:n]
    k = np.arange(l, dtype=f64)
    if kind == "hamming":
        return (0.54 - 0.46 * _cos_term64(k, 2, l - 1))[:n]
    if kind == "hann":
        return (0.5 * (1.0 - _cos_term64(k, 2, l - 1)))[:n]
    if kind == "kaiser":
        ratio = _lin64(-1, 1, l, endpoint=True)
        arg = np.maximum(1.0 - ratio ** 2, f64(eps))
        den = f64(_kaiser_i0(f32(beta)))  # on the NUMBER beta: an f32 tensor (windows.ex:362)
        return (_kaiser_i0_64(f64(beta) * np.sqrt(arg)) / den)[:n]
    raise ValueError(kind)


def sinc_f64(t):
    t = np.asarray(t, dtype=f64) * f64(f32(math.pi))  # pi() of Nx.Constants: the f32 constant
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.sin(t) / t
    return np.where(t == 0, 1.0, s)


def firwin_f64(num_taps, cutoff, window="hamming", pass_zero=True, scale=True, sampling_rate=2.0):
    """filters.ex:147-252 with type: {:f, 64}; the error cases are those of firwin"""
    nyq = sampling_rate / 2.0
    cl = sorted(c / nyq for c in cutoff)
    alpha = np.arange(num_taps, dtype=f64) - f64((num_taps - 1) / 2.0)
    all_freqs = [0.0] + cl + [1.0]
    h = np.zeros(num_taps, dtype=f64)
    for i in range(len(all_freqs) - 1):
        if (i % 2 == 0) if pass_zero else (i % 2 == 1):
            a, b = f64(f32(all_freqs[i])), f64(f32(all_freqs[i + 1]))  # defnp arguments: f32 tensors
            h = (h + b * sinc_f64(b * alpha)) - a * sinc_f64(a * alpha)
    if isinstance(window, tuple):
        w = window_f64("kaiser", num_taps, is_periodic=False, beta=window[1])
    else:
        w = window_f64(window, num_taps, is_periodic=False)
    h = h * w
    if not scale:
        return h
    sf = 0.0 if pass_zero else (1.0 if len(cl) == 1 else (cl[0] + cl[1]) / 2.0)
    dot = 0.0
    for v in (h * np.cos(alpha * f64(math.pi * sf))).tolist():
        dot += v
    return h / abs(dot)


def fft_frequencies_f64(sampling_rate, fft_length, endpoint=False):
    step = f32(f32(sampling_rate) / f32(fft_length))  # sampling_rate enters the defn as an f32 tensor
    return _lin64(0, f64(f32(step * f32(fft_length))), fft_length, endpoint=endpoint)


def _scale_factor_f64(window, scaling, sampling_rate):
    """the :scaling scalar in the WINDOW's type, widened (Nx.sum(window), Nx.sqrt(fs * Nx.sum(window ** 2)))"""
    window = np.asarray(window)
    if scaling is None:
        return None
    if window.dtype != f64:
        return f64(_scale_factor(window, scaling, sampling_rate))
    acc = 0.0
    if scaling == "spectrum":
        for v in window.tolist():
            acc += v
        return f64(acc)
    if scaling == "psd":
        for v in (window * window).tolist():
            acc += v
        return f64(np.sqrt(f64(f32(sampling_rate)) * acc))
    raise ValueError(f"invalid :scaling, expected one of :spectrum, :psd or nil, got: {scaling!r}")


def stft_f64(data, window, overlap_length=None, fft_length="power_of_two", window_padding="valid", sampling_rate=100, scaling=None,
             eps=FFT_EPS):
    """stft of f64 samples and / or with an f64 window: (z c128[..., M, K], times f32, freqs f32) — lib/nx_signal.ex:88-130"""
    window = np.asarray(window)
    N = window.shape[0]
    if overlap_length is None:
        overlap_length = N // 2
    hop = N - overlap_length
    d = np.asarray(data)
    if np.iscomplexobj(d):  # c128 samples (or c64 under an f64 window): complex x real componentwise, like the c64 path (SURVEY App. A rule 9)
        frames = as_windowed(d.astype(c128), N, hop, window_padding)
        w64 = window.astype(f64)
        fr = frames.real * w64 + 1j * (frames.imag * w64)
    else:
        frames = as_windowed(d.astype(f64), N, hop, window_padding)
        fr = frames * window.astype(f64)  # :101 in f64 (either operand widens exactly)
    K = _resolve_len(fft_length, N)
    z = _eps_clean(np.fft.fft(fr, n=K, axis=-1), eps)
    sf = _scale_factor_f64(window, scaling, sampling_rate)
    if sf is not None:
        z = z.real / sf + 1j * (z.imag / sf)
    return z.astype(c128), stft_times(N, sampling_rate, z.shape[-2]), fft_frequencies(sampling_rate, K)


def overlap_and_add_f64(t, overlap_length):
    """overlap_and_add of an f64 / c128 tensor: sequential double accumulation in frame order"""
    t = np.asarray(t)
    M, N = t.shape[-2], t.shape[-1]
    stride = N - overlap_length
    out_len = M * stride + overlap_length
    lead = t.shape[:-2]
    tt = t.reshape((-1, M, N))
    out = np.zeros((tt.shape[0], out_len), dtype=t.dtype)
    for m in range(M):
        out[:, m * stride: m * stride + N] += tt[:, m, :]
    return out.reshape(lead + (out_len,))


def istft_f64(z, window, overlap_length=None, sampling_rate=1000, scaling=None, eps=FFT_EPS):
    """istft of a c128 spectrum (window f32 or f64): c128[..., M*hop + overlap] — lib/nx_signal.ex:582-638"""
    window = np.asarray(window)
    N = window.shape[0]
    if overlap_length is None:
        overlap_length = N // 2
    z = np.asarray(z).astype(c128)
    if z.shape[-1] != N:
        raise ValueError("istft requires fft_length == window length (broadcast {M,K} x {N})")
    frames = _eps_clean(np.fft.ifft(z, axis=-1), eps)  # :609
    sf = _scale_factor_f64(window, scaling, sampling_rate)
    if sf is not None:
        frames = frames.real * sf + 1j * (frames.imag * sf)
    w64 = window.astype(f64)
    frames = frames.real * w64 + 1j * (frames.imag * w64)  # :628
    num = overlap_and_add_f64(frames, overlap_length)
    if window.dtype == f64:
        w2 = np.abs(w64) ** 2
        den = overlap_and_add_f64(np.broadcast_to(w2, z.shape[:-1] + (N,)).copy(), overlap_length)
        den = np.where(den > 1.0e-10, den, 1.0)
    else:  # the normaliser lives in the window's type: f32 products, double accumulation rounded to f32, f32 comparison
        w2 = _pow32(np.abs(window.astype(f32)), 2)
        den = overlap_and_add(np.broadcast_to(w2, z.shape[:-1] + (N,)), overlap_length, dtype=f32)
        den = np.where(den > f32(1.0e-10), den, f32(1.0)).astype(f64)
    return (num.real / den + 1j * (num.imag / den)).astype(c128)


def fftconvolve_f64(in1, in2, mode="full", eps=FFT_EPS):
    """1-D real fftconvolve of f64 operands: one transform of length n1 + n2 - 1 in c128 — convolution.ex:252-329"""
    a = np.asarray(in1, dtype=f64)
    b = np.asarray(in2, dtype=f64)
    n = a.shape[-1] + b.shape[-1] - 1
    sp = _eps_clean(np.fft.fft(a, n=n, axis=-1), eps) * _eps_clean(np.fft.fft(b, n=n, axis=-1), eps)
    out = _eps_clean(np.fft.ifft(sp, axis=-1), eps).real
    if mode == "full":
        return out
    new = a.shape[-1] if mode == "same" else abs(a.shape[-1] - b.shape[-1]) + 1
    start = (n - new) // 2
    return out[..., start:start + new]
