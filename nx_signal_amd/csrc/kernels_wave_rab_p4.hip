// Wave kernels, part 5 (round 5): the inverse-only list of wave_rab.hpp — istft of power-of-two frame lengths at any hop
// (NxSignal.istft, lib/nx_signal.ex:609-637, e.g. 512-sample frames every 160 samples); dispatched by kernels_wave_rab.hip
#include "wave_rab.hpp"

namespace nxsig {

int launch_istft_rab_p4(Ctx* c, const IstftLaunch& s, const float* window_host, bool* handled) {
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_istft_rab_AB<A, B>(c, s, window_host, handled);
    NXSIG_RAB_INVERSE_ONLY(X)
#undef X
    default: return NXSIG_OK;
  }
}

int launch_stft_rab_c64_p4(Ctx* c, const StftLaunch& s, bool* handled) {
  switch (s.K) {
#define X(KK, A, B) case KK: return launch_rab_c64<A, B>(c, s, handled);
    NXSIG_RAB_INVERSE_ONLY(X)
#undef X
    default: return NXSIG_OK;
  }
}

}  // namespace nxsig
