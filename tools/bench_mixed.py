#!/usr/bin/env python
"""bench_mixed.py -- per-state precision routing on the configs[2] scoring stage: the BASELINE model with a share of its
states holding one Gaussian over the two-term fp16 form's conditioning limits.

    python tools/bench_mixed.py [share ...]        default shares: 0 0.01 0.1 0.5 1.0

Prints, per share, the states on fp16 rows, ms of the routed pass (AASR_PREC_F16X2) and of the whole model on three
bf16 terms (AASR_PREC_BF16X3), and the ratio to the all-f16x2 pass."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from aaltoasr_amd import capi, synth  # noqa: E402

shares = [float(a) for a in sys.argv[1:]] or [0.0, 0.01, 0.1, 0.5, 1.0]
F = int(os.environ.get("AASR_BENCH_FRAMES", "449280"))
capi.check(capi.lib().aasr_set_device(0))
model = synth.make_model(D=bench.DIM, G=bench.G, S=bench.S, comps=bench.COMPS)
dev = torch.device("cuda", 0)
d_fr = torch.randn((F, bench.DIM), device=dev, dtype=torch.float32)
pitch = (bench.S + 31) // 32 * 32
d_out = torch.empty((F, pitch), device=dev, dtype=torch.float32)
stream = torch.cuda.current_stream()


def timed(g, reps=5):
    g.score_dev_pitched(d_fr, d_out, pitch, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        g.score_dev_pitched(d_fr, d_out, pitch, stream)
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rng = np.random.default_rng(synth.SEED + 99)
base = None
for share in shares:
    n_bad = int(round(share * bench.S))
    bad = sorted(rng.choice(bench.S, n_bad, replace=False).tolist()) if n_bad else []
    g = capi.Gmm.from_arrays(*(synth.push_states_over_the_f16_limits(model, bad) if bad else model))
    n16, moved = g.precision_states()
    ms4 = timed(g)
    g.set_precision(3)
    ms3 = timed(g)
    g.close()
    if base is None:
        base = ms4
    print("share %.2f: %4d states over the limits, %4d on fp16 rows (%d moved by the probe): routed %.3f ms (x%.3f, target x%.3f), "
          "whole model bf16x3 %.3f ms" % (share, n_bad, n16, moved, ms4, ms4 / base, 1 + 0.65 * share, ms3), flush=True)
