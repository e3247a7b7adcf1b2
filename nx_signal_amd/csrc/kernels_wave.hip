// Tuned wave-per-frame kernels (gfx950).  PLACEHOLDER until the generic path is parity-green on hardware:
// every launcher reports "not handled" so the dispatcher in api.cpp falls through to kernels_generic.hip.
#include "nxsig_internal.h"

namespace nxsig {
int launch_stft_wave(Ctx*, const StftLaunch&, bool* handled) { *handled = false; return NXSIG_OK; }
int launch_istft_wave(Ctx*, const IstftLaunch&, bool* handled) { *handled = false; return NXSIG_OK; }
int launch_fir_wave(Ctx*, const FirLaunch&, bool* handled) { *handled = false; return NXSIG_OK; }
}  // namespace nxsig
