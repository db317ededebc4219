"""End-to-end recipe run (BASELINE configs[3] shape, scaled down): N synthetic 10-s WAVs on
local disk -> features -> scoring (50k Gaussians) -> 2-byte LNA files.  Prints wall time,
device time and frames/s, i.e. including file reads, PCIe and LNA writes."""
import os, sys, tempfile, time, wave
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (loads the HIP runtime first)
from aaltoasr_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d = tempfile.mkdtemp(prefix="aasr_recipe_")
cfg = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                        "mfcc_cms_norm.feaconf")).read()
lines = []
base = synth.make_audio(160000, seed=1)
for i in range(N):
    p = os.path.join(d, "u%05d.wav" % i)
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(np.roll(base, 977 * i).astype("<i2").tobytes())
    lines.append("audio=%s lna=%s" % (p, os.path.join(d, "u%05d.lna" % i)))
recipe = os.path.join(d, "r.recipe")
open(recipe, "w").write("\n".join(lines) + "\n")
ft = capi.Feat(cfg)
gm = capi.Gmm.from_arrays(*synth.make_model(D=39, G=50000, S=3125, comps=16))
gm.set_precision(3)
for rep in range(2):
    t = time.time()
    st = capi.run_recipe(ft, gm, recipe, lnabytes=2)
    wall = time.time() - t
    print("run %d: %d utterances, %d frames, wall %.3f s, device %.3f s, %.2f M frames/s end to end" % (
        rep, st.utterances, st.frames, wall, st.seconds_device, st.frames / wall / 1e6))
