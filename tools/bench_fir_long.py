#!/usr/bin/env python
"""FIR filters beyond 4096 taps (reverb-length impulse responses): one long transform per row (nxsig_fir_f32's long-filter path).
One JSON object per line.  usage: python tools/bench_fir_long.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nx_signal_amd as S  # noqa: E402

ctx = S.Context(0)
rng = np.random.default_rng(0)
for rows, L, taps in [(8, 480000, 8193), (8, 480000, 48001), (2, 4800000, 96001)]:
    x = ctx.to_device(rng.standard_normal((rows, L)).astype(np.float32))
    h = rng.standard_normal(taps).astype(np.float32) / taps
    for _ in range(2):
        y = S.filters.fir(x, h, mode="same", ctx=ctx)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        y = S.filters.fir(x, h, mode="same", ctx=ctx)
    ctx.sync()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(json.dumps({"case": f"fir {taps} taps :same, {rows} rows x {L} samples (device-resident)", "ms": ms,
                      "Msamples_per_s": rows * L / ms / 1e3}), flush=True)
