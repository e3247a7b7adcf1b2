#!/bin/bash
# The parity / fuzz / tuned-kernel / n-D suites once per dispatch knob, so that the paths the default dispatch does not take
# stay correct (run through gpurun).  NXSIG_DISABLE_WAVE=1 sends everything to the generic kernels; the two tests that pin
# properties of the tuned kernels only are left out there.
cd "$(dirname "$0")/.."
SUITES="tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_tuned_kernels.py tests/test_gpu_nd.py"
for e in ${MATRIX_LIST:-NXSIG_FIR_R2K=0 NXSIG_FIR_R2K=2 NXSIG_DISABLE_RAB=1 NXSIG_DISABLE_R20=1 NXSIG_DISABLE_BLUE_WAVE=1 NXSIG_DISABLE_WAVE_ROWS=1 NXSIG_ISTFT_DEEP=0 NXSIG_FIR32=2 NXSIG_FIR32=0 NXSIG_FIR_PAD_TAPS=0 NXSIG_FIR_PHASE=0 NXSIG_DISABLE_FUSED_FILTER=1 NXSIG_POOL_MAX_MB=0 NXSIG_MEL_LDS_KB=150 NXSIG_FFT_TILED=0 NXSIG_FFT_TILE_ELEMS=2048 NXSIG_FFT_TILE_NT=256 NXSIG_FFT_COLUMNS=0 NXSIG_CONV_POW2=0 NXSIG_DIRECT_FAST=0 NXSIG_FIR_DLINE=0 NXSIG_FIR_DLINE=2 NXSIG_WAVE_SMALL_W=0 NXSIG_HOST_PIPE=1 NXSIG_WAVE_UNITS_PER_WAVE=1 NXSIG_FIR_UNITS_PER_WAVE=1 NXSIG_ISTFT_MIN_RUN=1 NXSIG_ISTFT_MIN_RUN=8}; do
  echo "== $e"; env $e python -m pytest $SUITES -q -m gpu 2>&1 | tail -1
done
# the first form of the stft_to_mel kernel does not fit fft_length 8192 into the LDS (the tiled form does)
echo "== NXSIG_MEL_TILE=0"; NXSIG_MEL_TILE=0 python -m pytest $SUITES -q -m gpu -k "not 8192-20-48000 and not stft_to_mel_bits" 2>&1 | tail -1
echo "== NXSIG_DISABLE_WAVE=1"; NXSIG_DISABLE_WAVE=1 python -m pytest $SUITES -q -m gpu -k "not leak_samples and not reused" 2>&1 | tail -1
