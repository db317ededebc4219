"""overlap_probe.py -- does the LNA stage (HBM-bound) hide under the scoring kernel (matrix-pipe bound) of the next
batch when the two run on separate HIP streams?  Prints ms per step of the serial chain and of the two-stream chain.

    python tools/overlap_probe.py [steps]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from aaltoasr_amd import capi, pipeline, synth  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    capi.check(capi.lib().aasr_set_device(0))
    gmm = capi.Gmm.from_arrays(*synth.make_model(D=bench.DIM, G=bench.G, S=bench.S, comps=bench.COMPS))
    gmm.set_precision(4)
    r = pipeline.FullChainBench(gmm, n_utts=360, seconds=10.0, rank=0, device=dev)
    ll = [r.d_ll, torch.empty_like(r.d_ll)]
    by = [r.d_bytes, torch.empty_like(r.d_bytes)]
    fe = [r.d_fea, torch.empty_like(r.d_fea)]

    def serial(K):
        for _ in range(K):
            r.step()

    for prio in (0, -1):
        sA = torch.cuda.Stream()
        sB = torch.cuda.Stream(priority=prio)
        ev_s = [torch.cuda.Event(), torch.cuda.Event()]
        ev_l = [torch.cuda.Event(), torch.cuda.Event()]

        def overlapped(K):
            for k in range(K):
                b = k & 1
                if k >= 2:
                    sA.wait_event(ev_l[b])
                r.feat.run_batch_dev(r.d_pcm, r.pcm_off, r.frame_off, fe[b], sA)
                gmm.score_dev_pitched(fe[b], ll[b], r.pitch, sA)
                ev_s[b].record(sA)
                sB.wait_event(ev_s[b])
                capi.lna_encode_dev(ll[b], True, r.lnabytes, None, by[b], sB, num_states=r.S)
                ev_l[b].record(sB)

        for name, fn in (("serial", serial), ("two streams (lna priority %d)" % prio, overlapped)):
            fn(3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(steps)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / steps
            print("%-34s %.3f ms/step  %.0f frames/s" % (name, ms, r.total_frames / ms * 1e3), flush=True)
    # same codes from both arrangements
    serial(1)
    torch.cuda.synchronize()
    print("codes differing between the two buffers:", int((by[0] != by[1]).sum()))


main()
