// Wave kernels, part 5 (round 5): composite fft lengths, list 5 of wave_rab.hpp (the 12- and 48-point codelets: 192 ... 1920, among them the 40 ms frame of 48 kHz audio) (dispatched by kernels_wave_rab.hip)
#include "wave_rab.hpp"

namespace nxsig {

int launch_stft_rab_p5(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel) {
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_rab<A, B>(c, s, handled, mel);
    NXSIG_RAB_PART5(X)
#undef X
    default: return NXSIG_OK;
  }
}

int launch_istft_rab_p5(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled) {
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_istft_rab_AB<A, B>(c, s, window_host, handled);
    NXSIG_RAB_PART5(X)
#undef X
    default: return NXSIG_OK;
  }
}

int launch_stft_rab_c64_p5(Ctx* c, const StftLaunch& s, bool* handled) {
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_rab_c64<A, B>(c, s, handled);
    NXSIG_RAB_PART5(X)
#undef X
    default: return NXSIG_OK;
  }
}

}  // namespace nxsig
