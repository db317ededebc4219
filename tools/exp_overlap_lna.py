"""exp_overlap_lna.py -- configs[2] with the LNA pass of batch k (bound by HBM) beside the feature chain of batch k + 1 (bound by
vector issue and latency chains) on two HIP streams; the scoring kernel, which fills every CU by itself, stays alone.
Round 3's three-stream experiment (tools/exp_overlap.py) put work beside the scoring kernel and measured slower.

    python tools/exp_overlap_lna.py [steps]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from aaltoasr_amd import capi, pipeline, synth  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    capi.check(capi.lib().aasr_set_device(0))
    gmm = capi.Gmm.from_arrays(*synth.make_model(D=bench.DIM, G=bench.G, S=bench.S, comps=bench.COMPS))
    gmm.set_precision(4)
    r = pipeline.FullChainBench(gmm, n_utts=360, seconds=10.0, rank=0, device=torch.device("cuda:0"))
    fe = [r.d_fea, torch.empty_like(r.d_fea)]
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    hi = torch.cuda.Stream(priority=-1)

    def serial(K):
        for _ in range(K):
            r.step()

    def overlapped(K):
        # main: score(k), then features(k + 1); side: LNA(k) -- both start when score(k) ends
        ev_scored = torch.cuda.Event()
        ev_lna = torch.cuda.Event()
        r.feat.run_batch_dev(r.d_pcm, r.pcm_off, r.frame_off, fe[0], main_s)
        for k in range(K):
            b = k & 1
            if k:
                main_s.wait_event(ev_lna)          # the scores of batch k - 1 have been read
            gmm.score_dev_pitched(fe[b], r.d_ll, r.pitch, main_s)
            ev_scored.record(main_s)
            side.wait_event(ev_scored)
            capi.lna_encode_dev(r.d_ll, True, r.lnabytes, None, r.d_bytes, side, num_states=r.S)
            ev_lna.record(side)
            if k + 1 < K:
                r.feat.run_batch_dev(r.d_pcm, r.pcm_off, r.frame_off, fe[b ^ 1], main_s)
        main_s.wait_event(ev_lna)

    def overlapped_hi(K):
        # the same with the feature kernels on a high-priority stream: their workgroups go first when slots free up
        ev_scored, ev_lna = torch.cuda.Event(), torch.cuda.Event()
        ev_feat = [torch.cuda.Event(), torch.cuda.Event()]
        r.feat.run_batch_dev(r.d_pcm, r.pcm_off, r.frame_off, fe[0], hi)
        ev_feat[0].record(hi)
        for k in range(K):
            b = k & 1
            main_s.wait_event(ev_feat[b])
            if k:
                main_s.wait_event(ev_lna)
            gmm.score_dev_pitched(fe[b], r.d_ll, r.pitch, main_s)
            ev_scored.record(main_s)
            side.wait_event(ev_scored)
            capi.lna_encode_dev(r.d_ll, True, r.lnabytes, None, r.d_bytes, side, num_states=r.S)
            ev_lna.record(side)
            if k + 1 < K:
                hi.wait_event(ev_scored)
                r.feat.run_batch_dev(r.d_pcm, r.pcm_off, r.frame_off, fe[b ^ 1], hi)
                ev_feat[b ^ 1].record(hi)
        main_s.wait_event(ev_lna)

    for name, fn in (("one stream", serial), ("LNA(k) beside features(k+1)", overlapped),
                     ("... features at high priority", overlapped_hi), ("one stream", serial),
                     ("LNA(k) beside features(k+1)", overlapped), ("... features at high priority", overlapped_hi)):
        fn(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        print("%-30s %.3f ms/step  %.1f M frames/s" % (name, ms, r.total_frames / ms / 1e3), flush=True)


main()
