// Wave kernels, part 5 (round 6): composite fft lengths, list 7 of wave_rab.hpp — 2400 = 50 x 48, 2880 = 60 x 48, 3840 = 64 x 60 (the 50 / 60 / 80 ms frames of 48 kHz audio); complex-spectrum sink only (dispatched by kernels_wave_rab.hip)
#include "wave_rab.hpp"

namespace nxsig {

int launch_stft_rab_p7(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel) {
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_rab<A, B, false>(c, s, handled, mel);
    NXSIG_RAB_PART7(X)
#undef X
    default: return NXSIG_OK;
  }
}

int launch_istft_rab_p7(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled) {
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_istft_rab_AB<A, B>(c, s, window_host, handled);
    NXSIG_RAB_PART7(X)
#undef X
    default: return NXSIG_OK;
  }
}

int launch_stft_rab_c64_p7(Ctx* c, const StftLaunch& s, bool* handled) {
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_rab_c64<A, B>(c, s, handled);
    NXSIG_RAB_PART7(X)
#undef X
    default: return NXSIG_OK;
  }
}

}  // namespace nxsig
