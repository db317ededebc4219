#!/usr/bin/env python
"""stage_split.py -- ms per stage of the configs[2] chain (features / scoring / LNA), each run on its own, for the
library AASR_LIBDIR selects: the same-box A/B of a kernel change (boxes differ by up to 30 % on the memory-bound stages).

    python tools/stage_split.py [reps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from aaltoasr_amd import capi, pipeline, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
capi.check(capi.lib().aasr_set_device(0))
gmm = capi.Gmm.from_arrays(*synth.make_model(D=bench.DIM, G=bench.G, S=bench.S, comps=bench.COMPS))
r = pipeline.FullChainBench(gmm, n_utts=360, seconds=10.0, rank=0, device=torch.device("cuda:0"))
r.step()
print(os.environ.get("AASR_LIBDIR", "lib"), r.stage_split(reps), flush=True)
