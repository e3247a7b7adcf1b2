// Wave-private FFT kernels, part 2 of 3: the fused STFT -> log-mel kernels and their launcher.
#include "wave_stft.hpp"

namespace nxsig {

int launch_stft_r20(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel);  // kernels_wave_r20.hip
int launch_stft_rab(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel);  // kernels_wave_rab.hip
int rab_length_part(int K);

// fused stft -> log-mel; *handled = false when the shape is not covered (the caller falls back to stft + stft_to_mel)
int launch_stft_mel_wave(Ctx* c, const StftLaunch& s, int mel_bins, const float* filters_host, float* out, bool* handled) {
  *handled = false;
  if (s.fr.M == 0 || s.batch == 0 || s.window_padK == nullptr) return NXSIG_OK;
  if (tune(c, kT_DISABLE_WAVE, 0)) return NXSIG_OK;
  MelLaunch mel{mel_bins, filters_host, out, handled};
  switch (s.K) {
    case 1024: return launch_wave<1024, kModePair, 4, 2, kSinkMel>(c, s, &mel);
    case 512: return launch_wave<1024, kModeQuad, 4, 2, kSinkMel>(c, s, &mel);
    case 256: return launch_wave<1024, kModeQuad, 4, 4, kSinkMel>(c, s, &mel);
    case 128: return launch_wave<1024, kModeQuad, 4, 8, kSinkMel>(c, s, &mel);
    case 2048: return launch_wave<1024, kModeReal2x, 4, 2, kSinkMel>(c, s, &mel);
    case 4096: return launch_wave<2048, kModeReal2x, 4, 2, kSinkMel>(c, s, &mel);
    default:
      if (s.K == 400) {  // native 20 x 20 kernel
        bool h20 = false;
        int rc20 = launch_stft_r20(c, s, &h20, &mel);
        if (rc20 || h20) return rc20;
      }
      if (rab_length_part(s.K) >= 0) {  // native A x B kernels (round 5)
        bool hab = false;
        int rcab = launch_stft_rab(c, s, &hab, &mel);
        if (rcab || hab) return rcab;
      }
      if ((s.K & (s.K - 1)) != 0 && s.K > 16 && s.K <= 1024 && !tune(c, kT_DISABLE_BLUE_WAVE, 0))
        return s.K <= 512 ? launch_blue_wave<1024, kSinkMel>(c, s, &mel) : launch_blue_wave<2048, kSinkMel>(c, s, &mel);
      return NXSIG_OK;
  }
}

}  // namespace nxsig
