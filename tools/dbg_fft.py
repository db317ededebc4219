import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from aaltoasr_amd import capi, synth
from oracle import oracle as O
pcm, _ = O.read_wav_pcm16('tests/golden/short.wav')
for mag in (0, 1):
    cfg = "module\n{\n name a\n type audiofile\n sample_rate 16000\n}\nmodule\n{\n name f\n type fft\n magnitude %d\n sources a\n}\n" % mag
    ch = O.FeatureChain(cfg); ft = capi.Feat(cfg)
    want = ch.generate(pcm, 0, 60); got = ft.run(pcm, 0, 60, dtype=np.float64)
    bad = (want != got)
    print('mag', mag, 'mismatch frac', bad.mean(), 'max rel', (np.abs(want-got)/np.maximum(np.abs(want),1e-30)).max())
    print(' per-bin mismatch counts (first 12 bins):', bad.sum(0)[:12], ' last:', bad.sum(0)[-5:])
