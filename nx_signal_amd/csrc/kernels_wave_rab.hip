// Wave kernels, part 5 (round 5): composite fft lengths — list 0 of wave_rab.hpp (320 / 480 / 640 / 960) and the dispatchers over
// all lists.  handled = false: the caller falls through (Bluestein wave kernel, generic kernels / generic inverse path).
#include "wave_rab.hpp"

namespace nxsig {

int launch_stft_rab_p1(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel);   // kernels_wave_rab_p1.hip ...
int launch_stft_rab_p2(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel);
int launch_stft_rab_p3(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel);
int launch_stft_rab_p5(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel);
int launch_stft_rab_p6(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel);
int launch_stft_rab_p7(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel);
int launch_istft_rab_p1(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled);
int launch_istft_rab_p2(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled);
int launch_istft_rab_p3(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled);
int launch_istft_rab_p5(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled);
int launch_istft_rab_p6(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled);
int launch_istft_rab_p7(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled);
int launch_istft_rab_p4(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled);
int launch_stft_rab_c64_p1(Ctx* c, const StftLaunch& s, bool* handled);
int launch_stft_rab_c64_p2(Ctx* c, const StftLaunch& s, bool* handled);
int launch_stft_rab_c64_p3(Ctx* c, const StftLaunch& s, bool* handled);
int launch_stft_rab_c64_p5(Ctx* c, const StftLaunch& s, bool* handled);
int launch_stft_rab_c64_p6(Ctx* c, const StftLaunch& s, bool* handled);
int launch_stft_rab_c64_p7(Ctx* c, const StftLaunch& s, bool* handled);
int launch_stft_rab_c64_p4(Ctx* c, const StftLaunch& s, bool* handled);

// 0..3, 5: the list that holds fft length K, -1: none
int rab_length_part(int K) {
  switch (K) {
#define X(KK, A, B) case KK:
    NXSIG_RAB_PART0(X) return 0;
    NXSIG_RAB_PART1(X) return 1;
    NXSIG_RAB_PART2(X) return 2;
    NXSIG_RAB_PART3(X) return 3;
    NXSIG_RAB_PART5(X) return 5;
    NXSIG_RAB_PART6(X) return 6;
    NXSIG_RAB_PART7(X) return 7;
#undef X
    default: return -1;
  }
}

// frame lengths that have an INVERSE kernel only (the forward direction has the power-of-two front ends)
bool rab_inverse_only(int K) {
  switch (K) {
#define X(KK, A, B) case KK:
    NXSIG_RAB_INVERSE_ONLY(X) return true;
#undef X
    default: return false;
  }
}

int launch_stft_rab(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel) {
  *handled = false;
  if (s.fr.M == 0 || s.batch == 0 || s.window_padK == nullptr) return NXSIG_OK;
  if (tune(c, kT_DISABLE_RAB, 0) || tune(c, kT_DISABLE_WAVE, 0)) return NXSIG_OK;
  switch (rab_length_part(s.K)) {
    case 1: return launch_stft_rab_p1(c, s, handled, mel);
    case 2: return launch_stft_rab_p2(c, s, handled, mel);
    case 3: return launch_stft_rab_p3(c, s, handled, mel);
    case 5: return launch_stft_rab_p5(c, s, handled, mel);
    case 6: return launch_stft_rab_p6(c, s, handled, mel);
    case 7: return launch_stft_rab_p7(c, s, handled, mel);
    case 0: break;
    default: return NXSIG_OK;
  }
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_rab<A, B>(c, s, handled, mel);   // (480 as 16 x 30 measured 0.45 against 0.51: half of its pass-B lanes idle)
    NXSIG_RAB_PART0(X)
#undef X
    default: return NXSIG_OK;
  }
}

// stft of complex samples: every length of the four lists + the power-of-two lengths of the inverse-only list (128 ... 1024)
int launch_stft_rab_c64(Ctx* c, const StftLaunch& s, bool* handled) {
  *handled = false;
  if (s.fr.M == 0 || s.batch == 0 || s.window_padK == nullptr) return NXSIG_OK;
  if (tune(c, kT_DISABLE_RAB, 0) || tune(c, kT_DISABLE_WAVE, 0)) return NXSIG_OK;
  if (rab_inverse_only(s.K)) return launch_stft_rab_c64_p4(c, s, handled);
  switch (rab_length_part(s.K)) {
    case 1: return launch_stft_rab_c64_p1(c, s, handled);
    case 2: return launch_stft_rab_c64_p2(c, s, handled);
    case 3: return launch_stft_rab_c64_p3(c, s, handled);
    case 5: return launch_stft_rab_c64_p5(c, s, handled);
    case 6: return launch_stft_rab_c64_p6(c, s, handled);
    case 7: return launch_stft_rab_c64_p7(c, s, handled);
    case 0: break;
    default: return NXSIG_OK;
  }
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_rab_c64<A, B>(c, s, handled);
    NXSIG_RAB_PART0(X)
#undef X
    default: return NXSIG_OK;
  }
}

int launch_istft_rab(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled) {
  *handled = false;
  if (s.K != s.N || s.M == 0 || s.batch == 0 || window_host == nullptr || s.filt != nullptr) return NXSIG_OK;
  if (tune(c, kT_DISABLE_RAB, 0) || tune(c, kT_DISABLE_WAVE, 0)) return NXSIG_OK;
  if (rab_inverse_only(s.K)) return launch_istft_rab_p4(c, s, window_host, handled);
  switch (rab_length_part(s.K)) {
    case 1: return launch_istft_rab_p1(c, s, window_host, handled);
    case 2: return launch_istft_rab_p2(c, s, window_host, handled);
    case 3: return launch_istft_rab_p3(c, s, window_host, handled);
    case 5: return launch_istft_rab_p5(c, s, window_host, handled);
    case 6: return launch_istft_rab_p6(c, s, window_host, handled);
    case 7: return launch_istft_rab_p7(c, s, window_host, handled);
    case 0: break;
    default: return NXSIG_OK;
  }
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_istft_rab_AB<A, B>(c, s, window_host, handled);
    NXSIG_RAB_PART0(X)
#undef X
    default: return NXSIG_OK;
  }
}

}  // namespace nxsig
