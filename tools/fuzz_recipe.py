"""Randomised end-to-end sweep of the recipe driver (aasr_run_recipe: files -> features -> scoring ->
LNA files) against the oracle's restatement of phone_probs (aku/phone_probs.cc:145-267, Recipe::read):
random 39-d models (ragged / tied mixtures, optionally clustered), 1-7 utterances from 300 samples
(two frames) to 2.5 s, recipe lines with start-time / end-time (which stay in force on later lines),
comments, -B / -I slices, 2- and 4-byte output, -N.  Per iteration: the set of files written, their frame
counts, 4-byte values within 1e-4 outside the reference's float-denormal band, 2-byte codes within one step
(and the fraction equal), and every file equal byte for byte to the same utterance run alone.
`python tools/fuzz_recipe.py SEED N`; exits non-zero on a failure."""
import os
import shutil
import sys
import tempfile
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _wav(path, pcm):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(pcm.astype("<i2").tobytes())


def run(seed=1, N=20, verbose=False):
    from aaltoasr_amd import capi, synth
    from oracle import oracle as O
    O.build()
    rng = np.random.default_rng(seed)
    cfg = open(os.path.join(ROOT, "tests", "golden", "mfcc_cms_norm.feaconf")).read()
    ch, ft = O.FeatureChain(cfg), capi.Feat(cfg)
    fails, worst = [], {"lp": 0.0, "code": 0, "equal": 1.0, "files": 0, "frames": 0}
    for it in range(N):
        d = tempfile.mkdtemp(prefix="aasr_fuzzr_")
        try:
            S = int(rng.integers(3, 50))
            tied = bool(rng.integers(0, 2))
            model = synth.make_model(D=39, G=int(rng.integers(S, 6 * S + 1)) if tied else 9 * S, S=S, comps_range=(1, 9),
                                     tied=tied, seed=int(rng.integers(1, 1 << 30)))
            om, gm = O.DiagModel(*model), capi.Gmm.from_arrays(*model)
            G = model[0].shape[0]
            clustered = bool(rng.integers(0, 3) == 0) and int(0.3 * G) >= 2   # the reader refuses C > 0.3 G
            if clustered:
                C = int(rng.integers(2, min(12, int(0.3 * G)) + 1))
                g2c = synth.make_clustering(model[0], C)
                pairs = [(int(i), int(c)) for i, c in enumerate(g2c)]
                minc, ming = float(rng.choice([0.0, 0.3])), float(rng.choice([0.1, 0.25, 0.6]))
                om.set_clustering(C, pairs, minc, ming)
                gm.set_clustering(C, pairs)
                gm.set_clustering_min_evals(minc, ming)
            nutt = int(rng.integers(1, 8))
            lines, pcms = ["# fuzz recipe %d/%d" % (seed, it)], []
            for u in range(nutt):
                n = int(rng.choice([300, 385, 1000, int(rng.integers(400, 40000))]))
                pcm = synth.make_audio(n, seed=int(rng.integers(1, 1 << 30)))
                pcms.append(pcm)
                _wav(os.path.join(d, "u%d.wav" % u), pcm)
                line = "audio=%s lna=u%d.lna" % (os.path.join(d, "u%d.wav" % u), u)
                r = rng.integers(0, 6)
                if r == 0:
                    line += " start-time=%.3f end-time=%.3f" % (rng.uniform(0, 0.5), rng.uniform(0.6, 3.0))
                elif r == 1:
                    line += " start-time=0 end-time=0"
                elif r == 2:
                    line += " start-time=%.3f" % rng.uniform(0, 0.2)
                lines.append(line)
                if rng.integers(0, 5) == 0:
                    lines.append("")
            recipe = os.path.join(d, "r.recipe")
            open(recipe, "w").write("\n".join(lines) + "\n")
            nb = int(rng.choice([0, 0, 2, 3]))
            bi = int(rng.integers(1, nb + 1)) if nb else 0
            nbytes = int(rng.choice([2, 4]))
            normalize = bool(rng.integers(0, 4) != 0)
            out = os.path.join(d, "out")
            os.makedirs(out)
            ctx = "seed %d it %d S %d utts %d B %d I %d bytes %d norm %d clustered %d" % (
                seed, it, S, nutt, nb, bi, nbytes, normalize, clustered)
            infos = O.recipe_read(open(recipe).read(), nb, bi)
            st = capi.run_recipe(ft, gm, recipe, lnabytes=nbytes, normalize=normalize, num_batches=nb, batch_index=bi,
                                 out_dir=out)
            want_files = sorted(i.lna_path for i in infos)
            got_files = sorted(os.listdir(out))
            if got_files != want_files or st.utterances != len(infos):
                fails.append("%s: files %s, expected %s" % (ctx, got_files, want_files))
                continue
            total = 0
            for info in infos:
                u = int(os.path.basename(info.audio_path)[1:-4])
                pcm = pcms[u]
                start, end = O.recipe_frame_limits(info, ch.frame_rate)
                eof = ch.num_frames(len(pcm))
                stop = min(end, eof)
                data = open(os.path.join(out, info.lna_path), "rb").read()
                nfr = max(0, stop - start)
                total += nfr
                if data[:5] != O.lna_header(S, nbytes) or len(data) != 5 + nfr * S * nbytes:
                    fails.append("%s: %s has %d bytes, expected %d frames" % (ctx, info.lna_path, len(data), nfr))
                    continue
                single, n1 = capi.run_utterance(ft, gm, pcm, start_frame=start, end_frame=end, lnabytes=nbytes,
                                                normalize=normalize)
                if n1 != nfr or single != data:
                    fails.append("%s: %s differs from the utterance run alone" % (ctx, info.lna_path))
                if nfr == 0:
                    continue
                fea = ch.generate(pcm, start, nfr)
                if clustered:
                    ll = om.score_clustered(fea)
                    lik = np.exp(ll)
                else:
                    ll, lik = om.score(fea, want_lik=True)
                lp_ref, by_ref = O.lna_encode(lik, normalize, nbytes)
                body = np.frombuffer(data[5:], np.uint8).reshape(nfr, -1)
                if nbytes == 4:
                    lp = body.view("<f4")
                    smooth = (ll > -87.0) | (ll < -104.5)
                    e = float(np.abs(lp - lp_ref)[smooth].max()) if smooth.any() else 0.0
                    worst["lp"] = max(worst["lp"], e)
                    if e > 1e-4:
                        fails.append("%s: %s 4-byte values off by %.3g" % (ctx, info.lna_path, e))
                else:
                    code = body.reshape(nfr, S, 2).astype(int)
                    code = code[..., 0] * 256 + code[..., 1]
                    cref = by_ref.reshape(nfr, S, 2).astype(int)
                    cref = cref[..., 0] * 256 + cref[..., 1]
                    dmax = int(np.abs(code - cref).max())
                    worst["code"] = max(worst["code"], dmax)
                    worst["equal"] = min(worst["equal"], float((code == cref).mean()))
                    if dmax > 1:
                        fails.append("%s: %s 2-byte codes off by %d" % (ctx, info.lna_path, dmax))
            if st.frames != total:
                fails.append("%s: %d frames reported, %d expected" % (ctx, st.frames, total))
            worst["files"] += len(infos)
            worst["frames"] += total
        except Exception as e:  # noqa: BLE001 -- a crash is a finding too
            fails.append("seed %d it %d: %s: %s" % (seed, it, type(e).__name__, e))
        finally:
            shutil.rmtree(d, ignore_errors=True)
        if verbose and fails and fails[-1].startswith("seed %d it %d" % (seed, it)):
            print("FAIL", fails[-1])
    return worst, fails


if __name__ == "__main__":
    worst, fails = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1,
                       int(sys.argv[2]) if len(sys.argv) > 2 else 20, verbose=True)
    print(worst)
    for f in fails[:20]:
        print("FAIL", f)
    print("failures: %d" % len(fails))
    sys.exit(1 if fails else 0)
