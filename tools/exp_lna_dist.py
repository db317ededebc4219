import os, sys
sys.path.insert(0, '/root/repo')
import torch, bench
from aaltoasr_amd import capi, pipeline, synth
capi.check(capi.lib().aasr_set_device(0))
gmm = capi.Gmm.from_arrays(*synth.make_model(D=bench.DIM, G=bench.G, S=bench.S, comps=bench.COMPS))
r = pipeline.FullChainBench(gmm, n_utts=360, seconds=10.0, rank=0, device=torch.device("cuda:0"))
r.step(); torch.cuda.synchronize()
ll = r.d_ll[:, :r.S]
mx = ll.max(dim=1).values
print("frames", mx.numel(), "max ll per frame: min %.1f mean %.1f max %.1f" % (mx.min().item(), mx.mean().item(), mx.max().item()))
for th in (-51, -60, -87.3, -103.9):
    print("  fraction of frames with best state >= %.1f: %.4f" % (th, (mx >= th).float().mean().item()))
print("fraction of values in the float-denormal band [-103.98, -87.34): %.4f" % (((ll < -87.3365) & (ll >= -103.98)).float().mean().item()))
print("fraction of values at the floor: %.4f" % ((ll <= -115.0).float().mean().item()))
