"""Subspace-constrained Gaussians (half of BASELINE configs[4]): PCGMM and SCGMM
entries of a 'variable' .gk file, expanded at load time onto the dense factor-row kernels, against
the oracle's restatement of aku/Subspaces.cc:450-469, 745-768 and aku/Distributions.cc:1638-1648,
1683-1704, 1785-1802, 1851-1859, 1886-1916.  PARITY UNPINNED: the reference does not compile this
code (USE_SUBSPACE_COV is never defined) and holds no goldens for it."""
import os
import subprocess
import sys

import numpy as np
import pytest

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spd(rng, d, scale=1.0):
    a = rng.standard_normal((d, d)) * 0.3
    return scale * (a @ a.T + 0.5 * np.eye(d))


def _sym(rng, d, scale):
    a = rng.standard_normal((d, d))
    return scale * 0.5 * (a + a.T)


def _pcgmm_entries(oracle, rng, d, K, G, n_diag=0):
    basis = np.array([_spd(rng, d)] + [_sym(rng, d, 0.5 / (K * np.sqrt(d))) for _ in range(K - 1)])
    entries = [("precision_subspace", 7, basis)]
    for g in range(G):
        kk = int(rng.integers(1, K + 1))
        lam = np.concatenate([[rng.uniform(0.5, 2.0)], rng.uniform(-0.4, 0.4, kk - 1)])
        entries.append(("pcgmm", 7, rng.standard_normal(d) * 0.8, lam))
    for g in range(n_diag):
        entries.insert(int(rng.integers(1, len(entries) + 1)),
                       ("diag", rng.standard_normal(d), np.exp(rng.uniform(-1, 1, d))))
    return entries


def _scgmm_entries(oracle, rng, d, K, G):
    thetas = []
    for b in range(K):
        P = _spd(rng, d) if b == 0 else _sym(rng, d, 0.5 / (K * np.sqrt(d)))
        thetas.append(np.concatenate([rng.standard_normal(d) * (0.5 if b == 0 else 0.1), oracle.map_m2v(P)]))
    entries = [("exponential_subspace", 3, np.array(thetas))]
    for g in range(G):
        kk = int(rng.integers(1, K + 1))
        lam = np.concatenate([[rng.uniform(0.5, 2.0)], rng.uniform(-0.4, 0.4, kk - 1)])
        entries.append(("scgmm", 3, lam))
    return entries


def _files(oracle, tmp_path, name, d, entries, S, comps, seed):
    G = sum(1 for e in entries if e[0] in ("pcgmm", "scgmm", "diag"))
    _, _, off, idx, w = synth.make_model(D=d, G=G, S=S, comps=comps, seed=seed, tied=True)
    base = str(tmp_path / name)
    oracle.write_gk_subspace(base + ".gk", d, entries)
    oracle.write_mc(base + ".mc", off, idx, w)
    oracle.write_ph(base + ".ph", S)
    return base, off, idx, w


def _check(capi, g, ref, frames, tol=1e-4):
    for prec in (0, 3, 4):
        g.set_precision(prec)
        got = g.score(frames)
        vis = ref > -103.97
        err = np.abs(got - ref)
        assert err[vis].max() <= tol and err.max() <= 2 * tol, (prec, err[vis].max(), err.max())


@pytest.mark.parametrize("d,K,G", [(5, 3, 24), (13, 6, 40), (39, 12, 48)])
def test_pcgmm_pool_matches_oracle(capi, oracle, tmp_path, d, K, G):
    rng = np.random.default_rng(100 + d)
    entries = _pcgmm_entries(oracle, rng, d, K, G, n_diag=4)
    base, off, idx, w = _files(oracle, tmp_path, "pc", d, entries, 6, 5, seed=d)
    frames = synth.make_frames(200, D=d, seed=3) * 0.8
    om = oracle.SubspaceModel(entries, d, off, idx, w)
    ref = om.score(frames.astype(np.float64))
    g = capi.Gmm.from_files(base + ".gk", base + ".mc", base + ".ph")
    assert g.num_gaussians == G + 4
    _check(capi, g, ref, frames)
    # the as-written expression (stray ';', no quadratic term) is a different function
    bug = oracle.SubspaceModel(entries, d, off, idx, w, pcgmm_as_written=True).score(frames.astype(np.float64))
    assert np.abs(bug - ref).max() > 1.0


@pytest.mark.parametrize("d,K,G", [(5, 3, 24), (13, 5, 36), (39, 10, 40)])
def test_scgmm_pool_matches_oracle(capi, oracle, tmp_path, d, K, G):
    """Includes the reference's constant as written: log det P - psi^T P^-1 psi - d log(2*3.1416),
    with P assembled through map_v2m's float 1/sqrt(2)."""
    rng = np.random.default_rng(200 + d)
    entries = _scgmm_entries(oracle, rng, d, K, G)
    base, off, idx, w = _files(oracle, tmp_path, "sc", d, entries, 5, 6, seed=d + 1)
    frames = synth.make_frames(150, D=d, seed=4) * 0.7
    ref = oracle.SubspaceModel(entries, d, off, idx, w).score(frames.astype(np.float64))
    g = capi.Gmm.from_files(base + ".gk", base + ".mc", base + ".ph")
    _check(capi, g, ref, frames)
    # the model cache carries the constant offsets
    g.write_cache(str(tmp_path / "sc.aasr"))
    g2 = capi.Gmm.from_cache(str(tmp_path / "sc.aasr"))
    g.set_precision(0)
    g2.set_precision(0)
    assert np.array_equal(g.score(frames), g2.score(frames))


def test_mixed_subspaces_in_one_pool(capi, oracle, tmp_path):
    rng = np.random.default_rng(9)
    d = 8
    entries = _pcgmm_entries(oracle, rng, d, 4, 10, n_diag=3) + _scgmm_entries(oracle, rng, d, 3, 9)
    base, off, idx, w = _files(oracle, tmp_path, "mix", d, entries, 4, 7, seed=2)
    frames = synth.make_frames(120, D=d, seed=8)
    ref = oracle.SubspaceModel(entries, d, off, idx, w).score(frames.astype(np.float64))
    _check(capi, capi.Gmm.from_files(base + ".gk", base + ".mc", base + ".ph"), ref, frames)


def test_pcgmm_at_a_larger_size_sampled(capi, oracle, tmp_path):
    """D = 39, 1 500 PCGMM Gaussians over a 40-element precision basis, 20 000 frames on the
    device; the oracle re-scores sampled frames."""
    rng = np.random.default_rng(77)
    d, K, G, F = 39, 40, 1500, 20000
    entries = _pcgmm_entries(oracle, rng, d, K, G)
    base, off, idx, w = _files(oracle, tmp_path, "big", d, entries, 100, 15, seed=5)
    frames = synth.make_frames(F, D=d, seed=6) * 0.6
    g = capi.Gmm.from_files(base + ".gk", base + ".mc", base + ".ph")
    om = oracle.SubspaceModel(entries, d, off, idx, w)
    pick = np.sort(rng.choice(F, 12, replace=False))
    ref = om.score(frames[pick].astype(np.float64))
    for prec in (0, 3, 4):
        g.set_precision(prec)
        got = g.score(frames)[pick]
        vis = ref > -103.97
        assert np.abs(got - ref)[vis].max() <= 1e-4


def test_errors_and_the_as_written_switch(capi, oracle, tmp_path):
    rng = np.random.default_rng(5)
    d = 5
    entries = _pcgmm_entries(oracle, rng, d, 3, 6)
    base, off, idx, w = _files(oracle, tmp_path, "e", d, entries, 2, 3, seed=1)
    # a Gaussian naming an undefined subspace
    bad = [("pcgmm", 99, np.zeros(d), np.ones(1))] + entries
    oracle.write_gk_subspace(str(tmp_path / "bad.gk"), d, bad)
    with pytest.raises(capi.AasrError, match="has not been defined"):
        capi.Gmm.from_files(str(tmp_path / "bad.gk"), base + ".mc", base + ".ph")
    # a precision that is not positive definite
    neg = [entries[0], ("pcgmm", 7, np.zeros(d), np.array([-1.0]))] + entries[2:]
    oracle.write_gk_subspace(str(tmp_path / "neg.gk"), d, neg)
    with pytest.raises(capi.AasrError, match="not positive definite"):
        capi.Gmm.from_files(str(tmp_path / "neg.gk"), base + ".mc", base + ".ph")
    # the legacy header forms have no subspace to score with
    open(tmp_path / "legacy.gk", "w").write("2 %d pcgmm\n" % d)
    with pytest.raises(capi.AasrError, match="names no subspace"):
        capi.Gmm.from_files(str(tmp_path / "legacy.gk"), base + ".mc", base + ".ph")
    # AASR_PCGMM_AS_WRITTEN=1: the engine refuses rather than scoring the linear expression
    code = ("import sys; sys.path.insert(0, %r); from aaltoasr_amd import capi\n"
            "try:\n    capi.Gmm.from_files(%r, %r, %r); print('LOADED')\n"
            "except capi.AasrError as e:\n    print('REFUSED', e)\n" % (ROOT, base + ".gk", base + ".mc", base + ".ph"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, AASR_PCGMM_AS_WRITTEN="1"))
    assert "REFUSED" in r.stdout and "stray ';'" in r.stdout, r.stdout + r.stderr
