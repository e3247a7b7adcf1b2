#!/usr/bin/env python
"""One RANK of a two-process RANKED group (real RCCL) running the sharded log-mel on its channel shard: the clamp floor must come
from the all-reduced maximum of BOTH ranks.  Started twice by tools/two_rank_one_gpu.sh (RANK / WORLD_SIZE / MASTER_* in the env).
Prints one JSON line; the oracle is used as the checker only."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402
from nx_signal_amd import sharding  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
g = sharding.Group.ranked()
N, hop, mb, L = 1024, 256, 128, 1024 + 256 * 399
rng = np.random.default_rng(5)
x = rng.standard_normal((4, L)).astype(np.float32)
x[:2] *= np.float32(1e-3)                      # rank 0's channels are quiet: only rank 1's maximum clamps them correctly
w = S.windows.hann(N)
opts = dict(overlap_length=N - hop, fft_length=N, sampling_rate=16000, mel_bins=mb)
c0, c1 = sharding.shard_channels(4, world, rank)
xd = [g.contexts[0].to_device(x[c0:c1])]
out = sharding.mel_spectrogram_sharded(g, xd, w, axis="channels", length=L, batch=4, **opts)[0].numpy()
full = S.mel_spectrogram(x, w, g.contexts[0], **opts)   # the whole tensor on this rank's GPU, unsharded
same = bool(np.array_equal(out.view(np.uint32), full[c0:c1].view(np.uint32)))
alone = S.mel_spectrogram(x[c0:c1], w, g.contexts[0], **opts)   # what the shard would be WITHOUT the exchange step
print(json.dumps({"rank": rank, "world": g.world, "has_rccl": bool(g.has_rccl), "shard_rows": [c0, c1],
                  "equals_unsharded_bits": same, "differs_from_unreduced": bool(not np.array_equal(out, alone)) if rank == 0 else None,
                  "floor": float(out.min()), "global_floor": float(full.min())}), flush=True)
g.barrier()
sys.exit(0 if same else 1)
