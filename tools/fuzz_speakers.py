"""Randomised sweep of speaker / utterance configurations through the recipe driver (phone_probs -S):
random .spkc files (per speaker any subset of: VTLN warp, normalisation, feature transform + bias, model-side
CMLLR as one global transform or per mixture / Gaussian groups; utterance entries; a default speaker and
utterance), recipes whose speaker= / utterance= keys stay in force on later lines, unknown speakers falling
back to the defaults -- against the oracle's restatement of aku::SpeakerConfig driving its feature chain and
AdaptedGaussian scoring (4-byte LNA without normalisation = the state log-likelihoods, 1e-4).
`python tools/fuzz_speakers.py SEED N`; exits non-zero on a failure."""
import os
import shutil
import sys
import tempfile
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = """module
{
  name audiofile
  type audiofile
  sample_rate 16000
}
module
{
  name fft
  type fft
  magnitude 0
  sources audiofile
}
module
{
  name vtln
  type vtln
  sources fft
}
module
{
  name mel
  type mel
  sources vtln
}
module
{
  name mfcc
  type dct
  dim 12
  sources mel
}
module
{
  name d1
  type delta
  sources mfcc
}
module
{
  name merged
  type merge
  sources mfcc d1
}
module
{
  name norm
  type normalization
  sources merged
}
module
{
  name mllr
  type lin_transform
  sources norm
}
"""
D = 24


def _nums(v):
    return " ".join("%.6g" % x for x in np.ravel(v))


def _w(rng):
    a = np.eye(D) * rng.uniform(0.9, 1.1, D) + 0.02 * rng.standard_normal((D, D))
    return np.hstack([0.3 * rng.standard_normal(D)[:, None], a])


def _speaker_block(rng, S, G):
    t = ""
    if rng.integers(0, 2):
        t += "  %svtln\n  {\n    warp_factor %.4g\n  }\n" % ("feature " if rng.integers(0, 2) else "", rng.uniform(0.88, 1.12))
    else:
        t += "  vtln\n  {\n  }\n"
    if rng.integers(0, 2):
        t += "  feature norm\n  {\n    mean %s\n    scale %s\n  }\n" % (_nums(rng.uniform(-1, 1, D)), _nums(rng.uniform(0.8, 1.2, D)))
    else:
        t += "  feature norm\n  {\n  }\n"
    r = rng.integers(0, 3)
    if r == 0:
        t += "  feature mllr\n  {\n    matrix %s\n    bias %s\n  }\n" % (
            _nums(np.eye(D) + 0.02 * rng.standard_normal((D, D))), _nums(0.2 * rng.standard_normal(D)))
    elif r == 1:
        t += "  feature mllr\n  {\n    matrix %s\n  }\n" % _nums(np.eye(D) + 0.02 * rng.standard_normal((D, D)))
    else:
        t += "  feature mllr\n  {\n  }\n"
    r = rng.integers(0, 4)
    if r == 0:
        t += "  model cmllr\n  {\n    unitmode UNIT_NO\n    w1 %s\n  }\n" % _nums(_w(rng))
    elif r == 1:       # groups of mixtures (states); the rest unadapted
        perm = rng.permutation(S)
        cut = sorted(rng.integers(0, S + 1, 2))
        groups = [perm[:cut[0]], perm[cut[0]:cut[1]]]
        body = "".join("    w%d %s %s\n" % (k + 1, " ".join(str(int(x)) for x in g), _nums(_w(rng)))
                       for k, g in enumerate(groups) if len(g))
        t += "  model cmllr\n  {\n    unitmode UNIT_MIX\n%s  }\n" % body
    elif r == 2:
        perm = rng.permutation(G)[:int(rng.integers(1, G + 1))]
        t += "  model cmllr\n  {\n    unitmode UNIT_GAUSSIAN\n    w1 %s %s\n  }\n" % (" ".join(str(int(x)) for x in perm), _nums(_w(rng)))
    else:
        t += "  model cmllr\n  {\n  }\n"
    return t


def run(seed=1, N=10, verbose=False):
    from aaltoasr_amd import capi, synth
    from oracle import oracle as O
    O.build()
    rng = np.random.default_rng(seed)
    base_fea = O.FeatureChain(CFG).generate(synth.make_audio(24000, seed=90), 0, 180)
    mu, sd = base_fea.mean(0), base_fea.std(0) + 1e-3
    fails, worst = [], {"ll": 0.0, "files": 0, "visible": 0}
    for it in range(N):
        d = tempfile.mkdtemp(prefix="aasr_fuzzs_")
        ctx = "seed %d it %d" % (seed, it)
        try:
            S = int(rng.integers(4, 20))
            tied = bool(rng.integers(0, 2))
            model = list(synth.make_model(D=D, G=int(rng.integers(S, 5 * S)) if tied else 6 * S, S=S, comps_range=(1, 6),
                                          tied=tied, var_lo=1.0, var_hi=6.0, seed=int(rng.integers(1, 1 << 30))))
            model[0] = mu + 0.7 * sd * model[0]
            model[1] = sd * sd * model[1]
            G = model[0].shape[0]
            base = os.path.join(d, "m")
            O.write_gk(base + ".gk", model[0], model[1])
            O.write_mc(base + ".mc", model[2], model[3], model[4])
            O.write_ph(base + ".ph", S)
            mean, var = O.read_gk(base + ".gk")
            names = ["s%d" % k for k in range(int(rng.integers(1, 5)))]
            text = "speaker default\n{\n  vtln\n  {\n  }\n  feature norm\n  {\n  }\n  feature mllr\n  {\n  }\n  model cmllr\n  {\n  }\n}\n"
            for nme in names:
                text += "speaker %s\n{\n%s}\n" % (nme, _speaker_block(rng, S, G))
            utts = []
            if rng.integers(0, 2):
                text += "utterance default\n{\n  vtln\n  {\n  }\n}\n"
                for k in range(int(rng.integers(0, 3))):
                    utts.append("u%d" % k)
                    text += "utterance u%d\n{\n  vtln\n  {\n    warp_factor %.4g\n  }\n}\n" % (k, rng.uniform(0.9, 1.1))
            spkc = os.path.join(d, "s.spkc")
            open(spkc, "w").write(text)
            nutt = int(rng.integers(1, 7))
            pcms, lines = [], []
            for u in range(nutt):
                pcm = synth.make_audio(int(rng.integers(3000, 30000)), seed=int(rng.integers(1, 1 << 30)))
                pcms.append(pcm)
                with wave.open(os.path.join(d, "a%d.wav" % u), "wb") as w:
                    w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
                    w.writeframes(pcm.astype("<i2").tobytes())
                line = "audio=%s lna=a%d.lna" % (os.path.join(d, "a%d.wav" % u), u)
                if u == 0 or rng.integers(0, 2):
                    line += " speaker=%s" % rng.choice(names + ["nobody"])
                if utts and rng.integers(0, 3) == 0:
                    line += " utterance=%s" % rng.choice(utts)
                lines.append(line)
            recipe = os.path.join(d, "r.recipe")
            open(recipe, "w").write("\n".join(lines) + "\n")
            # the oracle's loop (aku/phone_probs.cc:176-198)
            ch = O.FeatureChain(CFG)
            om = O.DiagModel(mean, var, model[2], model[3], model[4])
            osc = O.SpeakerConfig(ch, om)
            osc.read_text(text)
            want = []
            for pcm, info in zip(pcms, O.recipe_read(open(recipe).read())):
                osc.set_speaker(info.speaker_id)
                if info.utterance_id:
                    osc.set_utterance(info.utterance_id)
                fea = ch.generate(pcm, 0, ch.num_frames(len(pcm)))
                want.append(O.score_adapted(om, fea, osc.g2t, osc.W))
            ft = capi.Feat(CFG)
            gm = capi.Gmm.from_files(base + ".gk", base + ".mc", base + ".ph")
            sc = capi.SpeakerConfig(ft, gm)
            sc.read_file(spkc)
            out = os.path.join(d, "out")
            os.makedirs(out)
            st = capi.run_recipe(ft, gm, recipe, lnabytes=4, normalize=False, out_dir=out, speakers=sc)
            if st.utterances != nutt:
                fails.append("%s: %d utterances written, %d expected" % (ctx, st.utterances, nutt))
                continue
            for u, ll in enumerate(want):
                got = O.lna_decode(open(os.path.join(out, "a%d.lna" % u), "rb").read())
                if got.shape != ll.shape:
                    fails.append("%s: a%d.lna has shape %s, expected %s" % (ctx, u, got.shape, ll.shape))
                    continue
                ok = ll > -85
                worst["visible"] += int(ok.sum())
                e = float(np.abs(got - ll)[ok].max()) if ok.any() else 0.0
                worst["ll"] = max(worst["ll"], e)
                if e > 1e-4:
                    fails.append("%s: a%d.lna off by %.3g (%s)" % (ctx, u, e, lines[u].split(" ", 2)[-1]))
            worst["files"] += nutt
        except Exception as e:  # noqa: BLE001
            fails.append("%s: %s: %s" % (ctx, type(e).__name__, e))
        finally:
            shutil.rmtree(d, ignore_errors=True)
        if verbose and fails and fails[-1].startswith(ctx):
            print("FAIL", fails[-1])
    return worst, fails


if __name__ == "__main__":
    worst, fails = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1,
                       int(sys.argv[2]) if len(sys.argv) > 2 else 10, verbose=True)
    print(worst)
    print("failures: %d" % len(fails))
    sys.exit(1 if fails else 0)
