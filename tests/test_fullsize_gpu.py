"""Parity at BASELINE.json's full model size (configs[1]/[2]: 39-d, 50 000
Gaussians, 3 125 states x 16): the whole block is scored on the GPU, the oracle
re-scores a sample of its frames against all 50 000 Gaussians, and
size-independent properties cover the rest (partition invariance, posterior
rows summing to one, LNA codes consistent with their own log-probabilities)."""
import numpy as np
import pytest

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu

D, G, S, COMPS = 39, 50000, 3125, 16
F = 20000


@pytest.fixture(scope="module")
def big(capi, oracle):
    import torch
    model = synth.make_model(D=D, G=G, S=S, comps=COMPS)
    frames = synth.make_frames(F, D=D, seed=77)
    g = capi.Gmm.from_arrays(*model)
    d_fr = torch.from_numpy(frames).cuda()
    outs = {}
    for name, prec in (("f32", 0), ("bf16x3", 3)):
        g.set_precision(prec)
        d_out = torch.empty((F, S), dtype=torch.float32, device="cuda")
        g.score_dev(d_fr, d_out)
        torch.cuda.synchronize()
        outs[name] = d_out
    g.set_precision(0)
    return dict(model=model, frames=frames, g=g, outs=outs, om=oracle.DiagModel(*model), d_fr=d_fr)


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_sampled_frames_against_oracle(big, prec):
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(F, 48, replace=False))
    pick[0], pick[-1] = 0, F - 1                      # block edges included
    ref = big["om"].score(big["frames"][pick].astype(np.float64))
    got = big["outs"][prec][pick.tolist()].cpu().numpy()
    err = np.abs(got - ref)
    assert err.max() <= 1e-4, "%s: max |dll| %.3g" % (prec, err.max())


def test_f32_and_bf16x3_agree_everywhere(big):
    d = (big["outs"]["f32"] - big["outs"]["bf16x3"]).abs().max().item()
    assert d <= 1e-4


def test_partition_invariance_at_full_model_size(big):
    """Scoring a sub-block alone gives the same bits as inside the big block."""
    import torch
    g = big["g"]
    for lo, hi in ((0, 1), (255, 513), (F - 300, F)):
        d_out = torch.empty((hi - lo, S), dtype=torch.float32, device="cuda")
        g.score_dev(big["d_fr"][lo:hi].contiguous(), d_out)
        torch.cuda.synchronize()
        assert torch.equal(d_out, big["outs"]["f32"][lo:hi])


def test_lna_rows_are_posteriors_and_codes_follow_them(capi, big):
    import torch
    ll = big["outs"]["f32"][:4096].contiguous()
    d_lp = torch.empty_like(ll)
    d_by = torch.empty((ll.shape[0], S * 2), dtype=torch.uint8, device="cuda")
    capi.lna_encode_dev(ll, True, 2, d_lp, d_by)
    torch.cuda.synchronize()
    lp = d_lp.cpu().numpy().astype(np.float64)
    assert np.allclose(np.exp(lp).sum(1), 1.0, atol=2e-5)
    code = d_by.cpu().numpy().reshape(-1, S, 2).astype(np.int64)
    code = code[..., 0] * 256 + code[..., 1]
    want = np.where(lp < -36.008, 0xFFFF, (-1820.0 * lp + .5).astype(np.int64) & 0xFFFF)
    assert np.array_equal(code, want)
