// Row transforms on the wave-private cores: Nx.fft / Nx.ifft(length: K) over the last axis for K = 1024, 2048 (one pass
// through the 1024- / 2048-point core) and K = 4096 (four passes through the 1024-point core + a lane-local radix-4, the
// decimation-in-time split of kernels_wave_8k.hip).  One wave per row, no workgroup barrier after the table preload.
// These are the rows NxSignal.Transforms.fft_nd / ifft_nd (lib/nx_signal/transforms.ex:5-21), n-D fftconvolve
// (lib/nx_signal/convolution.ex:252-347), the four-step / Bluestein transforms of kernels_nd.hip and the generic istft
// (lib/nx_signal.ex:609) are folded over; the workgroup LDS kernels of kernels_generic.hip keep every other length.
//   in : f32 or c64 rows of n_in elements, zero-padded / truncated to K (Nx.fft(length:))
//   out: c64 rows of K bins, 16-byte stores (two adjacent bins per lane, 1 KiB per wave instruction)
//   inverse: x 1 / K; optional istft epilogue (x scale x window[k], lib/nx_signal.ex:611-628)
#include "wave_stft.hpp"

namespace nxsig {

struct RowsWaveArgs {
  const void* in;
  int64_t rows;
  int32_t n_in, in_is_real;
  const v2f* twB;
  const v2f* twC;
  const v2f* tw4k;          // K = 4096: w_4096^k (forward) or its conjugate (inverse), k < 1024
  const float* post_window;  // f32[K] or nullptr
  float post_scale;
  int32_t has_post_scale;
  int32_t clean;             // 1: eps clean-up of the finished transform (0 when the rows are a sub-step of a longer transform)
  int64_t chunk;            // rows per workgroup
  v2f* out;
  // ---- framed input (NxSignal.stft of COMPLEX samples, lib/nx_signal.ex:94-102; forward only): row r is frame r % fM of signal row
  // r / fM of `in` (c64[..][fL], rows fstride elements apart); element idx of the frame is padded-signal sample (r % fM) fhop + idx
  // (zero / mirror padding by flo, lib/nx_signal.ex:338 / :349) times pre_window[idx] (componentwise exact f32 products, :101); the
  // finished spectrum is divided by `div` (:113-127).  fM == 0: plain rows.
  int64_t fM = 0, fL = 0, flo = 0, fstride = 0;
  int32_t fhop = 0, freflect = 0;
  const float* pre_window = nullptr;   // f32[n_in]
  float div = 1.0f;
  int32_t has_div = 0;
};

template <bool INV>
__device__ __forceinline__ v4f rows_epilogue(const RowsWaveArgs& a, v2f z0, v2f z1, int k, float invK) {
  v4f o = v4f{z0.x, z0.y, z1.x, z1.y};
  if (INV) o = o * invK;  // exact for powers of two
  if (a.clean) o = fft_eps0(o);  // Nx.fft / Nx.ifft clean-up, ahead of the istft epilogue
  if (!INV && a.has_div) o = o / a.div;   // stft :spectrum / :psd: true division like the reference
  if (a.has_post_scale) o = o * a.post_scale;
  if (a.post_window) {
    const v2f w = *reinterpret_cast<const v2f*>(a.post_window + k);
    o = v4f{o.x * w.x, o.y * w.x, o.z * w.y, o.w * w.y};
  }
  return o;
}

// element `idx` of the row (zero beyond n_in)
__device__ __forceinline__ v2f rows_fetch(const RowsWaveArgs& a, const void* row, int idx) {
  if (idx >= a.n_in) return v2f{0.f, 0.f};
  if (a.in_is_real) return v2f{reinterpret_cast<const float*>(row)[idx], 0.f};
  return reinterpret_cast<const v2f*>(row)[idx];
}

// element `idx` of frame m of a complex signal row (framed input): padded-signal index q = m hop + idx
__device__ __forceinline__ v2f rows_fetch_framed(const RowsWaveArgs& a, const v2f* sig, int64_t q0, int idx) {
  if (idx >= a.n_in) return v2f{0.f, 0.f};
  int64_t pos = q0 + idx - a.flo;
  if (q0 - a.flo >= 0 && q0 - a.flo + a.n_in <= a.fL) {   // wave-uniform: the frame lies inside the signal (all but the edge frames)
  } else if (a.freflect) {
    if (a.fL == 1) pos = 0;
    else {
      const int64_t period = 2 * (a.fL - 1);
      pos %= period;
      if (pos < 0) pos += period;
      if (pos >= a.fL) pos = period - pos;
    }
  } else if (pos < 0 || pos >= a.fL) return v2f{0.f, 0.f};   // zero padding (an exact 0 x w = 0 either way)
  const v2f v = sig[pos];
  const float w = a.pre_window[idx];
  return v2f{v.x * w, v.y * w};
}

// K = 1024 / 2048: one core pass per row, the next row's points prefetched during the butterflies
template <int K, bool INV, int W>
__global__ __launch_bounds__(64 * W) void k_fft_rows_wave(RowsWaveArgs a) {
  constexpr int P = K / 64, R3 = K / 256, NQ = K / 128, XCH = K + K / 16 + 16;
  v2f* s_twB = reinterpret_cast<v2f*>(g_wave_smem);
  v2f* s_twC = s_twB + 256;
  v2f* s_x = s_twC + R3 * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 256; i += 64 * W) s_twB[i] = a.twB[i];
  for (int i = tid; i < R3 * 256; i += 64 * W) s_twC[i] = a.twC[i];
  __syncthreads();
  v2f* xb = s_x + wave * XCH;
  const int64_t r_begin = (int64_t)blockIdx.x * a.chunk;
  int64_t r_end = r_begin + a.chunk;
  if (r_end > a.rows) r_end = a.rows;
  const size_t row_bytes = (size_t)a.n_in * (a.in_is_real ? 4 : 8);
  const float invK = 1.0f / (float)K;
  v2f nx[P];
  auto issue = [&](int64_t r) {
    if (!INV && a.fM > 0) {   // wave-uniform
      const int64_t b = r / a.fM, m = r - b * a.fM;
      const v2f* sig = static_cast<const v2f*>(a.in) + (size_t)b * a.fstride;
      const int64_t p0 = m * a.fhop - a.flo;
      if (p0 >= 0 && p0 + a.n_in <= a.fL) {   // wave-uniform: the frame lies inside the signal (all but the edge frames): plain loads
        const v2f* fp = sig + p0 + lane;
#pragma unroll
        for (int s = 0; s < P; ++s) {
          const bool in = lane + 64 * s < a.n_in;
          const v2f v = in ? fp[64 * s] : v2f{0.f, 0.f};
          const float w = in ? a.pre_window[lane + 64 * s] : 0.0f;
          nx[s] = v2f{v.x * w, v.y * w};
        }
        return;
      }
#pragma unroll
      for (int s = 0; s < P; ++s) nx[s] = rows_fetch_framed(a, sig, m * a.fhop, lane + 64 * s);
      return;
    }
    const char* row = static_cast<const char*>(a.in) + (size_t)r * row_bytes;
#pragma unroll
    for (int s = 0; s < P; ++s) nx[s] = rows_fetch(a, row, lane + 64 * s);
  };
  if (r_begin + wave < r_end) issue(r_begin + wave);
  for (int64_t r = r_begin + wave; r < r_end; r += W) {
    v2f d[P];
#pragma unroll
    for (int s = 0; s < P; ++s) d[s] = nx[s];
    issue(r + W < r_end ? r + W : r);  // unconditional prefetch (the last iteration re-reads its own row)
    __builtin_amdgcn_sched_barrier(0);
    v2f zz[2][NQ];
    wave_fft_core<K, INV>(d, zz, xb, s_twB, s_twC, lane);
    v2f* orow = a.out + (size_t)r * K + 2 * lane;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      __builtin_nontemporal_store(rows_epilogue<INV>(a, zz[0][q], zz[1][q], 2 * lane + 128 * q, invK), (gv4f*)(orow + 128 * q));
  }
}

// K = 4096: Y_r = FFT_1024(x[4 n' + r]); X[k + 1024 m] = sum_r w_4096^(r k) w_4^(r m) Y_r[k] (conjugated tables for the inverse)
template <bool INV, int W>
__global__ __launch_bounds__(64 * W) void k_fft_rows_wave_4k(RowsWaveArgs a) {
  constexpr int K = 1024, P = 16, NQ = 8, XCH = K + K / 16 + 16, KOUT = 4096;
  v2f* s_twB = reinterpret_cast<v2f*>(g_wave_smem);
  v2f* s_twC = s_twB + 256;
  v2f* s_t4 = s_twC + 4 * 256;
  v2f* s_x = s_t4 + K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 256; i += 64 * W) s_twB[i] = a.twB[i];
  for (int i = tid; i < 4 * 256; i += 64 * W) s_twC[i] = a.twC[i];
  for (int i = tid; i < K; i += 64 * W) s_t4[i] = a.tw4k[i];
  __syncthreads();
  v2f* xb = s_x + wave * XCH;
  const int64_t r_begin = (int64_t)blockIdx.x * a.chunk;
  int64_t r_end = r_begin + a.chunk;
  if (r_end > a.rows) r_end = a.rows;
  const size_t row_bytes = (size_t)a.n_in * (a.in_is_real ? 4 : 8);
  const float invK = 1.0f / (float)KOUT;
  const bool full_c64 = a.fM == 0 && !a.in_is_real && a.n_in >= KOUT && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 && (row_bytes & 15) == 0;  // uniform
  for (int64_t r = r_begin + wave; r < r_end; r += W) {
    const char* row = static_cast<const char*>(a.in) + (size_t)r * row_bytes;
    const int64_t fb = a.fM > 0 ? r / a.fM : 0;
    const v2f* fsig = static_cast<const v2f*>(a.in) + (size_t)fb * a.fstride;
    const int64_t fq0 = a.fM > 0 ? (r - fb * a.fM) * a.fhop : 0;
    v2f y[4][2][NQ];
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // sub-sequences (0, 1) then (2, 3): each 16-byte load carries two of them
      v2f d0[P], d1[P];
      if (full_c64) {
        const v4f* p4 = reinterpret_cast<const v4f*>(row) + 2 * lane + h;
#pragma unroll
        for (int s = 0; s < P; ++s) { const v4f t = p4[128 * s]; d0[s] = v2f{t.x, t.y}; d1[s] = v2f{t.z, t.w}; }
      } else if (!INV && a.fM > 0 && fq0 - a.flo >= 0 && fq0 - a.flo + a.n_in <= a.fL) {   // wave-uniform: a frame inside the signal
        const v2f* fp = fsig + (fq0 - a.flo);
#pragma unroll
        for (int s = 0; s < P; ++s) {
          const int i0 = 4 * (lane + 64 * s) + 2 * h;
          const bool in0 = i0 < a.n_in, in1 = i0 + 1 < a.n_in;
          const v2f v0 = in0 ? fp[i0] : v2f{0.f, 0.f}, v1 = in1 ? fp[i0 + 1] : v2f{0.f, 0.f};
          const float w0 = in0 ? a.pre_window[i0] : 0.0f, w1 = in1 ? a.pre_window[i0 + 1] : 0.0f;
          d0[s] = v2f{v0.x * w0, v0.y * w0};
          d1[s] = v2f{v1.x * w1, v1.y * w1};
        }
      } else {
#pragma unroll
        for (int s = 0; s < P; ++s) {
          if (!INV && a.fM > 0) {
            d0[s] = rows_fetch_framed(a, fsig, fq0, 4 * (lane + 64 * s) + 2 * h);
            d1[s] = rows_fetch_framed(a, fsig, fq0, 4 * (lane + 64 * s) + 2 * h + 1);
          } else {
            d0[s] = rows_fetch(a, row, 4 * (lane + 64 * s) + 2 * h);
            d1[s] = rows_fetch(a, row, 4 * (lane + 64 * s) + 2 * h + 1);
          }
        }
      }
      wave_fft_core<K, INV>(d0, y[2 * h], xb, s_twB, s_twC, lane);
      __builtin_amdgcn_sched_barrier(0);
      wave_fft_core<K, INV>(d1, y[2 * h + 1], xb, s_twB, s_twC, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
    const StreamRow os(a.out + (size_t)r * KOUT, KOUT * 8);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      v2f o[4][2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const v2f t1 = s_t4[2 * lane + e + 128 * q];
        const v2f t2 = wcmul(t1, t1), t3 = wcmul(t2, t1);
        v2f a0 = y[0][e][q], a1 = wcmul(y[1][e][q], t1), a2 = wcmul(y[2][e][q], t2), a3 = wcmul(y[3][e][q], t3);
        dft4<INV>(a0, a1, a2, a3);
        o[0][e] = a0; o[1][e] = a1; o[2][e] = a2; o[3][e] = a3;
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
        os.st16(rows_epilogue<INV>(a, o[m][0], o[m][1], 2 * lane + 128 * q + 1024 * m, invK), lane * 16 + 1024 * q + 8192 * m);
    }
  }
}

struct RowsFraming {   // framed input of launch_fft_rows_wave (see RowsWaveArgs)
  int64_t M, L, lo, stride;
  int32_t hop, reflect;
  const float* pre_window;
  float div;
  int32_t has_div;
};
static const RowsFraming* g_rows_framing_none = nullptr;
int launch_fft_rows_wave_framed(Ctx* c, const void* in, bool in_is_real, int64_t rows, int32_t n_in, int32_t K, bool inverse,
                                const float* post_window, float post_scale, bool has_post_scale, float2* out, bool* handled, bool clean,
                                const RowsFraming* fr);
// returns handled = false for lengths / shapes the wave kernels do not take
int launch_fft_rows_wave(Ctx* c, const void* in, bool in_is_real, int64_t rows, int32_t n_in, int32_t K, bool inverse,
                         const float* post_window, float post_scale, bool has_post_scale, float2* out, bool* handled, bool clean) {
  return launch_fft_rows_wave_framed(c, in, in_is_real, rows, n_in, K, inverse, post_window, post_scale, has_post_scale, out, handled, clean,
                                     g_rows_framing_none);
}
// NxSignal.stft of complex samples (lib/nx_signal.ex:94-102) for fft_length 1024 / 2048 / 4096 in ONE launch: frame slice x window fused
// into the loads of the row kernels, :spectrum / :psd division into their epilogue.  handled = false: the caller takes the two-step path.
int launch_stft_c64_wave(Ctx* c, const StftLaunch& s, bool* handled) {
  *handled = false;
  if (s.fr.M == 0 || s.batch == 0) return NXSIG_OK;
  if (tune(c, kT_DISABLE_WAVE, 0)) return NXSIG_OK;
  RowsFraming f;
  f.M = s.fr.M; f.L = s.fr.L; f.lo = s.fr.lo; f.stride = s.batch_stride; f.hop = s.fr.hop; f.reflect = s.fr.reflect;
  f.pre_window = s.window; f.div = s.inv_scale_div; f.has_div = s.has_scale;
  const int n_in = s.fr.N < s.K ? s.fr.N : s.K;   // Nx.fft(length: K) truncates longer frames
  return launch_fft_rows_wave_framed(c, s.x, false, (int64_t)s.batch * s.fr.M, n_in, s.K, false, nullptr, 1.0f, false, s.z, handled, true, &f);
}
int launch_fft_rows_wave_framed(Ctx* c, const void* in, bool in_is_real, int64_t rows, int32_t n_in, int32_t K, bool inverse,
                                const float* post_window, float post_scale, bool has_post_scale, float2* out, bool* handled, bool clean,
                                const RowsFraming* fr) {
  *handled = false;
  if ((K != 1024 && K != 2048 && K != 4096) || rows < 1 || tune(c, kT_DISABLE_WAVE_ROWS, 0)) return NXSIG_OK;
  if (post_window && (reinterpret_cast<uintptr_t>(post_window) & 7) != 0) return NXSIG_OK;
  const int C = K == 2048 ? 2048 : 1024;
  int rc = ensure_wave_tables(c, C);
  if (rc) return rc;
  *handled = true;
  Ctx::WaveTables& wt = c->wave_tables[C];
  RowsWaveArgs a;
  a.in = in; a.rows = rows; a.n_in = n_in; a.in_is_real = in_is_real ? 1 : 0;
  a.twB = reinterpret_cast<const v2f*>(inverse ? wt.twBi : wt.twB);
  a.twC = reinterpret_cast<const v2f*>(inverse ? wt.twCi : wt.twC);
  a.tw4k = nullptr;
  a.post_window = post_window; a.post_scale = post_scale; a.has_post_scale = has_post_scale ? 1 : 0; a.clean = clean ? 1 : 0;
  a.out = reinterpret_cast<v2f*>(out);
  if (fr) {
    a.fM = fr->M; a.fL = fr->L; a.flo = fr->lo; a.fstride = fr->stride; a.fhop = fr->hop; a.freflect = fr->reflect;
    a.pre_window = fr->pre_window; a.div = fr->div; a.has_div = fr->has_div;
  }
  constexpr int W = 4;
  const int rpw = fill_units_per_wave(c, rows, 4, K == 4096 ? 2 : 4);
  a.chunk = (int64_t)W * (rpw < 1 ? 1 : rpw);
  const int64_t blocks = (rows + a.chunk - 1) / a.chunk;
  if (blocks > 0x7fffffffLL) return set_error(NXSIG_ERR_UNSUPPORTED, "fft: too many rows for one launch");
  auto go = [&](auto kernel, size_t lds) -> int {
    if (lds > 64 * 1024)
      NXSIG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dispatch_note(fr ? "stft_c64.rows" : "fft.rows_wave");
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, a);
    NXSIG_HIP_TRY(hipGetLastError());
    return NXSIG_OK;
  };
  if (K == 1024) {
    const size_t lds = (size_t)(256 + 4 * 256) * 8 + (size_t)W * (1024 + 64 + 16) * 8;
    return inverse ? go(k_fft_rows_wave<1024, true, W>, lds) : go(k_fft_rows_wave<1024, false, W>, lds);
  }
  if (K == 2048) {
    const size_t lds = (size_t)(256 + 8 * 256) * 8 + (size_t)W * (2048 + 128 + 16) * 8;
    return inverse ? go(k_fft_rows_wave<2048, true, W>, lds) : go(k_fft_rows_wave<2048, false, W>, lds);
  }
  {  // 4096: combine twiddles w_4096^k (conjugated for the inverse)
    const uint64_t key = inverse ? 0x8B15000000000001ull : 0x8B15000000000000ull;
    auto hit = c->memo.find(key);
    if (hit != c->memo.end()) a.tw4k = reinterpret_cast<const v2f*>(hit->second[0]);
    else {
      std::vector<float2> t4(1024);
      for (int k = 0; k < 1024; ++k) {
        const double ang = (inverse ? 1.0 : -1.0) * 6.283185307179586476925286766559 * (double)k / 4096.0;
        t4[k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
      }
      const void* d4 = nullptr;
      if ((rc = ctx_table(c, inverse ? 0x8B16ull : 0x8B17ull, t4.data(), t4.size() * sizeof(float2), &d4))) return rc;
      c->memo[key] = {reinterpret_cast<uint64_t>(d4)};
      a.tw4k = reinterpret_cast<const v2f*>(d4);
    }
    const size_t lds = (size_t)(256 + 4 * 256 + 1024) * 8 + (size_t)W * (1024 + 64 + 16) * 8;
    return inverse ? go(k_fft_rows_wave_4k<true, W>, lds) : go(k_fft_rows_wave_4k<false, W>, lds);
  }
}

}  // namespace nxsig
