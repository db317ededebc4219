#!/usr/bin/env python
"""exp_overlap.py -- configs[2] with its three stages on three HIP streams.

The chain's stages load different parts of the chip: the feature kernels are bound by vector-instruction issue,
the scoring kernel by the matrix pipe, the LNA pass by HBM.  Run back to back on one stream they add up
(1.05 + 9.05 + 1.73 ms); cut into K groups of utterances -- group c's scoring waits for its features, its LNA pass
for its scores (events) -- features of group c+1 and the LNA pass of group c-1 can run beside the scoring of
group c.  This measures what that buys, for several K, against the one-stream step.

    python tools/exp_overlap.py [K ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from aaltoasr_amd import capi, pipeline, synth  # noqa: E402


def main():
    ks = [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 12]
    dev = torch.device("cuda:0")
    capi.check(capi.lib().aasr_set_device(0))
    gmm = capi.Gmm.from_arrays(*synth.make_model(D=bench.DIM, G=bench.G, S=bench.S, comps=bench.COMPS))
    r = pipeline.FullChainBench(gmm, n_utts=360, seconds=10.0, rank=0, device=dev)
    n_utts = len(r.utts)

    def timed(fn, reps=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    print("one stream, one launch per stage: %.3f ms" % timed(r.step), flush=True)
    want = torch.empty_like(r.d_bytes)
    r.step()
    torch.cuda.synchronize()
    want.copy_(r.d_bytes)
    for K in ks:
        cuts = [n_utts * c // K for c in range(K + 1)]
        feats = [capi.Feat(r.cfg_text) for _ in range(K)]
        s_f, s_g, s_l = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        ev_f = [torch.cuda.Event() for _ in range(K)]
        ev_g = [torch.cuda.Event() for _ in range(K)]
        main_stream = torch.cuda.current_stream()
        ev_in, ev_out = torch.cuda.Event(), torch.cuda.Event()

        def step():
            ev_in.record(main_stream)
            for s in (s_f, s_g, s_l):
                s.wait_event(ev_in)
            for c in range(K):
                u0, u1 = cuts[c], cuts[c + 1]
                p0, p1 = int(r.pcm_off[u0]), int(r.pcm_off[u1])
                f0, f1 = int(r.frame_off[u0]), int(r.frame_off[u1])
                feats[c].run_batch_dev(r.d_pcm[p0:p1], r.pcm_off[u0:u1 + 1] - p0, r.frame_off[u0:u1 + 1] - f0,
                                       r.d_fea[f0:f1], s_f)
                ev_f[c].record(s_f)
                s_g.wait_event(ev_f[c])
                gmm.score_dev_pitched(r.d_fea[f0:f1], r.d_ll[f0:f1], r.pitch, s_g)
                ev_g[c].record(s_g)
                s_l.wait_event(ev_g[c])
                capi.lna_encode_dev(r.d_ll[f0:f1], True, r.lnabytes, None, r.d_bytes[f0:f1], s_l, num_states=r.S)
            ev_out.record(s_l)
            main_stream.wait_event(ev_out)

        r.d_bytes.zero_()
        ms = timed(step)
        torch.cuda.synchronize()
        same = bool(torch.equal(want, r.d_bytes))
        print("K = %2d groups on three streams: %.3f ms   LNA bytes identical to the one-stream step: %s"
              % (K, ms, same), flush=True)


if __name__ == "__main__":
    main()
