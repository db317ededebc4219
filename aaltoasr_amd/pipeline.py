"""pipeline.py -- device-resident whole-path runner used by bench.py
(BASELINE configs[2]/[3]): synthetic 16 kHz utterances -> MFCC chain ->
state log-likelihoods -> 2-byte LNA codes, everything resident in HBM.

Plumbing only: every stage is a call through the C ABI (capi.Feat / capi.Gmm /
capi.lna_encode_dev); torch supplies device memory and the stream.
"""
from __future__ import annotations

import os

import numpy as np

from . import capi, synth

class FullChainBench:
    def __init__(self, gmm: capi.Gmm, n_utts: int, seconds: float, rank: int, device,
                 cfg_text: str = None, lnabytes: int = 2, utts=None):
        import torch
        self.torch = torch
        self.gmm = gmm
        self.cfg_text = cfg_text if cfg_text is not None else synth.make_feature_config()
        self.feat = capi.Feat(self.cfg_text)
        self.lnabytes = lnabytes
        sr = self.feat.sample_rate
        n = int(round(seconds * sr))
        if utts is None:
            # a handful of distinct seeded utterances, tiled: the audio content does
            # not change the work, generating 1 h of noise on the host would only
            # slow the bench start-up
            base = [synth.make_audio(n, seed=synth.SEED + 1000 * rank + i, sample_rate=sr) for i in range(8)]
            utts = [base[i % len(base)] for i in range(n_utts)]
        self.utts = utts
        frames = [self.feat.eof_frame(len(u)) for u in utts]
        self.pcm_off = np.concatenate([[0], np.cumsum([len(u) for u in utts])]).astype(np.int64)
        self.frame_off = np.concatenate([[0], np.cumsum(frames)]).astype(np.int64)
        self.total_frames = int(self.frame_off[-1])
        self.d_pcm = torch.from_numpy(np.concatenate(utts)).to(device)
        S = gmm.num_states
        self.d_fea = torch.empty((self.total_frames, self.feat.dim), dtype=torch.float32, device=device)
        # the score matrix stays on the device: rows padded to whole 64-byte lines where the scoring
        # kernel can write them that way (what the C++ recipe driver does too)
        self.S = S
        self.pitch = (S + 31) // 32 * 32 if gmm.score_pitch_ok() else S
        self.d_ll = torch.empty((self.total_frames, self.pitch), dtype=torch.float32, device=device)
        self.d_bytes = torch.empty((self.total_frames, S * lnabytes), dtype=torch.uint8, device=device)
        self.stream = torch.cuda.current_stream()

    def step(self) -> None:
        self.feat.run_batch_dev(self.d_pcm, self.pcm_off, self.frame_off, self.d_fea, self.stream)
        self.gmm.score_dev_pitched(self.d_fea, self.d_ll, self.pitch, self.stream)
        capi.lna_encode_dev(self.d_ll, True, self.lnabytes, None, self.d_bytes, self.stream, num_states=self.S)

    def score_only(self) -> None:
        self.gmm.score_dev_pitched(self.d_fea, self.d_ll, self.pitch, self.stream)

    def features_only(self) -> None:
        self.feat.run_batch_dev(self.d_pcm, self.pcm_off, self.frame_off, self.d_fea, self.stream)

    def lna_only(self) -> None:
        capi.lna_encode_dev(self.d_ll, True, self.lnabytes, None, self.d_bytes, self.stream, num_states=self.S)

    def stage_split(self, reps: int = 3) -> dict:
        """ms per step of each stage run on its own (HIP events on the launch stream)."""
        torch = self.torch
        out = {}
        for name, fn in (("features", self.features_only), ("scoring", self.score_only), ("lna", self.lna_only)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            torch.cuda.synchronize()
            e0.record(self.stream)
            for _ in range(reps):
                fn()
            e1.record(self.stream)
            torch.cuda.synchronize()
            out[name] = round(e0.elapsed_time(e1) / reps, 4)
        return out

    def release(self) -> None:
        """Drops the device buffers (the bench measures another workload afterwards)."""
        self.d_pcm = self.d_fea = self.d_ll = self.d_bytes = None

    def bytes_written_per_frame(self) -> dict:
        """Algorithmic HBM bytes each stage stores per frame in this arrangement."""
        return {"features_f32": self.feat.dim * 4, "state_scores_f32": self.pitch * 4,
                "lna_codes": self.S * self.lnabytes}
