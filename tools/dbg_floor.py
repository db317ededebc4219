import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from aaltoasr_amd import capi, synth
from oracle import oracle as O
model = synth.make_model(D=39, G=256, S=32, comps=8)
fr = synth.make_frames(64); fr[:16] *= 3.0; fr[16:32] *= 6.0; fr[32:48] += 4.0
ref = O.DiagModel(*model).score(fr.astype(np.float64))
got = capi.Gmm.from_arrays(*model).score(fr)
err = np.abs(got - ref)
idx = np.argsort(err.ravel())[::-1][:12]
for i in idx:
    f, s = divmod(i, 32)
    print(f, s, 'ref %.6f got %.6f err %.3g' % (ref[f, s], got[f, s], err[f, s]))
