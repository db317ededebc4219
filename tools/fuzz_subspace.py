"""Randomised sweep of subspace-constrained pools (PCGMM / SCGMM entries of a 'variable' .gk file, with diagonal
entries mixed in) against oracle.SubspaceModel -- PARITY UNPINNED like tests/test_subspace_gpu.py, whose model
builders it reuses: the reference does not compile this code.  `python tools/fuzz_subspace.py SEED N`."""
import importlib.util
import os
import pathlib
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(seed=1, N=20, verbose=False):
    from aaltoasr_amd import capi, synth
    from oracle import oracle as O
    O.build()
    spec = importlib.util.spec_from_file_location("tsub", os.path.join(ROOT, "tests", "test_subspace_gpu.py"))
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    rng = np.random.default_rng(seed)
    worst, fails = {"pcgmm": 0.0, "scgmm": 0.0, "mixed": 0.0, "refused": 0}, []
    for it in range(N):
        d = pathlib.Path(tempfile.mkdtemp(prefix="aasr_fuzzsub_"))
        try:
            dim = int(rng.choice([2, 5, 8, 13, 24, 39]))
            kind = ["pcgmm", "scgmm", "mixed"][int(rng.integers(0, 3))]
            K, G = int(rng.integers(1, 9)), int(rng.integers(4, 40))
            if kind == "pcgmm":
                entries = T._pcgmm_entries(O, rng, dim, K, G, n_diag=int(rng.integers(0, 4)))
            elif kind == "scgmm":
                entries = T._scgmm_entries(O, rng, dim, K, G)
            else:
                entries = T._pcgmm_entries(O, rng, dim, K, G, n_diag=2) + T._scgmm_entries(O, rng, dim, max(1, K // 2), max(4, G // 2))
            S, comps = int(rng.integers(1, 9)), int(rng.integers(1, 8))
            base, off, idx, w = T._files(O, d, "m", dim, entries, S, comps, seed=int(rng.integers(1, 1 << 30)))
            frames = (synth.make_frames(int(rng.integers(1, 200)), D=dim, seed=int(rng.integers(1, 1 << 30))) * rng.uniform(0.4, 1.0)).astype(np.float32)
            ref = O.SubspaceModel(entries, dim, off, idx, w).score(frames.astype(np.float64))
            ctx = "seed %d it %d %s dim %d K %d G %d S %d" % (seed, it, kind, dim, K, G, S)
            try:
                g = capi.Gmm.from_files(base + ".gk", base + ".mc", base + ".ph")
            except capi.AasrError as e:
                worst["refused"] += 1
                if verbose:
                    print("refused:", ctx, str(e)[:100])
                continue
            for prec in (0, 3, 4):
                try:
                    g.set_precision(prec)
                except capi.AasrError:
                    continue
                got = g.score(frames)
                err = np.abs(got - ref)
                vis = ref > -103.97
                evis = float(err[vis].max()) if vis.any() else 0.0
                worst[kind] = max(worst[kind], evis)
                with np.errstate(under="ignore"):
                    flushes = (np.exp(got[~vis].astype(np.float64)).astype(np.float32) <= np.float32(2.0 ** -149)).all()
                if evis > 1e-4 or not flushes:
                    fails.append("%s prec %d: %.3g (visible %.3g)" % (ctx, prec, float(err.max()), evis))
                    if verbose:
                        print("FAIL", fails[-1])
        except Exception as e:  # noqa: BLE001
            fails.append("seed %d it %d: %s: %s" % (seed, it, type(e).__name__, e))
            if verbose:
                print("FAIL", fails[-1])
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return worst, fails


if __name__ == "__main__":
    worst, fails = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1,
                       int(sys.argv[2]) if len(sys.argv) > 2 else 20, verbose=True)
    print(worst)
    print("failures: %d" % len(fails))
    sys.exit(1 if fails else 0)
