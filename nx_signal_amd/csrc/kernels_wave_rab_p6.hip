// Wave kernels, part 5 (round 6): composite fft lengths, list 6 of wave_rab.hpp — radix 7: 882 = 42 x 21, 1764 = 42 x 42 (the 20 / 40 ms frames of 44.1 kHz audio); complex-spectrum sink only (dispatched by kernels_wave_rab.hip)
#include "wave_rab.hpp"

namespace nxsig {

int launch_stft_rab_p6(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel) {
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_rab<A, B, false>(c, s, handled, mel);
    NXSIG_RAB_PART6(X)
#undef X
    default: return NXSIG_OK;
  }
}

int launch_istft_rab_p6(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled) {
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_istft_rab_AB<A, B>(c, s, window_host, handled);
    NXSIG_RAB_PART6(X)
#undef X
    default: return NXSIG_OK;
  }
}

int launch_stft_rab_c64_p6(Ctx* c, const StftLaunch& s, bool* handled) {
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_rab_c64<A, B>(c, s, handled);
    NXSIG_RAB_PART6(X)
#undef X
    default: return NXSIG_OK;
  }
}

}  // namespace nxsig
