"""BASELINE configs 4 / 5 the way a multi-GPU host would run them: 64 channels sharded over a group, no data-path collective.

One process drives every GPU of the node through a LOCAL group (what the Elixir host does through the NIF); on a box with a
single GPU the members share it, which exercises the same code (plans, per-member streams, assembly) without the scaling.
Checks the sharded results against the unsharded calls bit for bit.  Run on a GPU box: python examples/sharded_groups.py [members]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import _lib, sharding  # noqa: E402


def main():
    lib = _lib.load()
    import ctypes as C
    n = C.c_int()
    _lib.check(lib.nxsig_device_count(C.byref(n)))
    members = int(sys.argv[1]) if len(sys.argv) > 1 else max(n.value, 2)
    devices = [i % n.value for i in range(members)]
    g = sharding.Group.local(members, devices=devices)
    print(f"group: {g.world} members on devices {devices}, RCCL communicators: {g.has_rccl}")
    rng = np.random.default_rng(0)
    ch, L = 64, 48000 * 20                       # 64 channels x 20 s (the configs use 600 s: same code, 30x the data)
    x = rng.standard_normal((ch, L), dtype=np.float32)
    w = S.windows.hann(2048)
    opts = dict(overlap_length=2048 - 512, fft_length=2048, sampling_rate=48000)
    h = S.filters.firwin(257, [4000.0], sampling_rate=48000)
    print("channel plan:", [sharding.shard_channels(ch, g.world, r) for r in range(g.world)])
    t0 = time.perf_counter(); z = sharding.stft_sharded(g, x, w, axis="channels", **opts); t1 = time.perf_counter()
    zr, _, _ = S.stft(x, w, **opts)
    print(f"config 4 shape: stft of {ch} ch sharded by channels {z.shape} in {t1 - t0:.2f} s (host tensors: PCIe-bound); identical to the unsharded call: {np.array_equal(z.view(np.uint32), zr.view(np.uint32))}")
    y = sharding.fir_sharded(g, x, h, mode="same", axis="channels")
    print(f"config 5 shape: fir 257 taps :same sharded by channels {y.shape}; identical: {np.array_equal(y.view(np.uint32), S.filters.fir(x, h).view(np.uint32))}")
    # one long stream by frame ranges, assembled with the all-gather (RCCL when every member has its own GPU)
    long = rng.standard_normal(48000 * 120, dtype=np.float32)
    w1 = S.windows.hann(1024)
    o1 = dict(overlap_length=768, fft_length=1024, sampling_rate=48000)
    zl = sharding.stft_sharded(g, long, w1, axis="frames", gather=True, **o1)
    yl = sharding.istft_sharded(g, zl, w1, axis="frames", gather=True, overlap_length=768, sampling_rate=48000)
    zf, _, _ = S.stft(long, w1, **o1)
    print(f"120 s stream by frame ranges: stft {zl.shape} normalised err vs unsharded {float(np.max(np.abs(zl - zf)) / np.max(np.abs(zf))):.1e}; istft round trip err on the interior "
          f"{float(np.max(np.abs(yl.real[1024:-1024] - long[1024:len(yl) - 1024]))):.1e}; sharded istft identical to unsharded: "
          f"{np.array_equal(yl.view(np.uint32), S.istft(zl, w1, overlap_length=768, sampling_rate=48000).view(np.uint32))}")
    # the one call with an exchange step: the log-mel's clamp needs the maximum over EVERY shard (an all-reduce between its passes)
    xm = x[:, : 48000 * 5].copy()
    xm[: ch // 2] *= np.float32(1e-3)            # half of the channels are quiet: their floor comes from the loud members' maximum
    mo = dict(overlap_length=2048 - 512, fft_length=2048, sampling_rate=48000, mel_bins=128)
    mel = sharding.mel_spectrogram_sharded(g, xm, w, axis="channels", **mo)
    print(f"log-mel of {ch} ch sharded by channels {mel.shape}, global maximum all-reduced; identical to the unsharded call: "
          f"{np.array_equal(mel.view(np.uint32), S.mel_spectrogram(xm, w, **mo).view(np.uint32))}")
    g.close()


if __name__ == "__main__":
    main()
