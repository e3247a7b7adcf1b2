// Wave-private FFT kernels, part 3 of 3: the fused STFT -> magnitude / power / dBFS kernels and their launcher.
#include "wave_stft.hpp"

namespace nxsig {

int launch_stft_r20(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel);  // kernels_wave_r20.hip
int launch_stft_rab(Ctx* c, const StftLaunch& s, bool* handled, const MelLaunch* mel);  // kernels_wave_rab.hip
int rab_length_part(int K);

// fused stft -> magnitude / power / dBFS spectrogram of the bins below fft_length / 2
int launch_stft_mag_wave(Ctx* c, const StftLaunch& s, int kind, float* out, bool* handled) {
  *handled = false;
  if (s.fr.M == 0 || s.batch == 0 || s.window_padK == nullptr) return NXSIG_OK;
  if (tune(c, kT_DISABLE_WAVE, 0)) return NXSIG_OK;
  MelLaunch mel{0, nullptr, out, handled, kind};
  if (kind == 4 && s.K != 1024) return NXSIG_OK;   // packed one-sided form: fused for the pair front-end, two-step elsewhere
  switch (s.K) {
    case 1024: return launch_wave<1024, kModePair, 4, 2, kSinkMag>(c, s, &mel);
    case 512: return launch_wave<1024, kModeQuad, 4, 2, kSinkMag>(c, s, &mel);
    case 256: return launch_wave<1024, kModeQuad, 4, 4, kSinkMag>(c, s, &mel);
    case 128: return launch_wave<1024, kModeQuad, 4, 8, kSinkMag>(c, s, &mel);
    case 2048: return launch_wave<1024, kModeReal2x, 4, 2, kSinkMag>(c, s, &mel);
    case 4096:
      if (kind == 3) return NXSIG_OK;  // the 2048-point core rounds a few bins differently when half its outputs are dead code:
                                       // the one-sided form keeps its "same bits as stft" promise through the two-step path
      return launch_wave<2048, kModeReal2x, 4, 2, kSinkMag>(c, s, &mel);
    default:
      const bool ab = rab_length_part(s.K) >= 0;
      if (kind == 3 && s.K != 400 && !ab) return NXSIG_OK;  // one-sided complex output: the power-of-two front-ends above, the 20 x 20
                                                            // and the A x B kernels store it; the rest slice the full spectrum
      if (s.K == 400) {  // native 20 x 20 kernel
        bool h20 = false;
        int rc20 = launch_stft_r20(c, s, &h20, &mel);
        if (rc20 || h20) return rc20;
      }
      if (ab) {  // native A x B kernels (round 5)
        bool hab = false;
        int rcab = launch_stft_rab(c, s, &hab, &mel);
        if (rcab || hab) return rcab;
      }
      if (kind == 3) return NXSIG_OK;
      if ((s.K & (s.K - 1)) != 0 && s.K > 16 && s.K <= 1024 && !tune(c, kT_DISABLE_BLUE_WAVE, 0))
        return s.K <= 512 ? launch_blue_wave<1024, kSinkMag>(c, s, &mel) : launch_blue_wave<2048, kSinkMag>(c, s, &mel);
      return NXSIG_OK;
  }
}

}  // namespace nxsig
