"""What a user gets who links the reference's own phone_probs.cc (text unchanged) against the adapters
(oracle/_ref/phone_probs_refmain): frames/s of the per-frame caller loop served from the engine's block
cache, beside the engine's batched phone_probs on the same files.  configs[2]'s model (D = 39,
50 000 Gaussians, 3 125 states x 16), N 10-s utterances.  Two recipe sizes separate start-up (model text
parse) from the per-frame rate."""
import os, subprocess, sys, tempfile, time, wave
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aaltoasr_amd import synth
from oracle import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
d = tempfile.mkdtemp(prefix="aasr_refmain_")
cfg = os.path.join(ROOT, "tests", "golden", "mfcc_cms_norm.feaconf")
mean, var, off, idx, w = synth.make_model(D=39, G=50000, S=3125, comps=16)
base = os.path.join(d, "m")
t = time.time()
oracle.write_gk(base + ".gk", mean, var)
oracle.write_mc(base + ".mc", off, idx, w)
oracle.write_ph(base + ".ph", 3125)
print("model files written in %.1f s (%.0f MB .gk)" % (time.time() - t, os.path.getsize(base + ".gk") / 1e6))
pcm = synth.make_audio(160000, seed=1)
lines = []
for i in range(N):
    p = os.path.join(d, "u%04d.wav" % i)
    with wave.open(p, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
        f.writeframes(np.roll(pcm, 977 * i).astype("<i2").tobytes())
    lines.append("audio=%s lna=u%04d.lna" % (p, i))
frames_per = 1248


def run(exe, n, tag, extra=()):
    rec = os.path.join(d, "r%d.recipe" % n)
    open(rec, "w").write("\n".join(lines[:n]) + "\n")
    out = os.path.join(d, "out_%s_%d" % (tag, n))
    os.makedirs(out, exist_ok=True)
    t = time.time()
    r = subprocess.run([exe, "-b", base, "-c", cfg, "-r", rec, "-o", out, "--lnabytes=2", *extra],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return time.time() - t


for tag, exe in (("reference main on the adapters", os.path.join(ROOT, "oracle", "_ref", "phone_probs_refmain")),
                 ("engine phone_probs (batched)", os.path.join(ROOT, "aaltoasr_amd", "lib", "bin", "phone_probs"))):
    for env_prec in ("3",):
        os.environ["AASR_PREC"] = env_prec
        small, big = max(1, N // 6), N
        ts, tb = run(exe, small, tag[:3]), run(exe, big, tag[:3])
        if tb - ts > 0.2:
            rate = (big - small) * frames_per / (tb - ts)
            print("%s: %d utt %.2f s, %d utt %.2f s -> start-up %.2f s, %.0f frames/s steady" % (
                tag, small, ts, big, tb, ts - small * frames_per / rate, rate))
        else:   # start-up (model text parse, device set-up) hides the per-frame cost at this size
            print("%s: %d utt %.2f s, %d utt %.2f s (start-up bound; see tools/bench_recipe.py for its rate)" % (
                tag, small, ts, big, tb))
