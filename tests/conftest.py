import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def capi():
    from aaltoasr_amd import build, capi as A
    build.build()
    A.lib()
    return A


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


TOL_LL = 1e-4                     # north_star: LNA log-likelihoods within 1e-4 of the reference CPU path
LL_FLUSH = -150.0 * 0.6931471805599453   # ln 2^-150: below it the reference's float storage of the
                                         # likelihood (phone_probs.cc:224-236) holds 0.0 and the LNA entry is the floor
CODES_EQUAL_MIN = 0.99            # 2-byte code equality: never more than one step apart, and at least this share
                                  # identical (a different v_exp_f32 / libm rounding moves a fraction of a per cent)


def assert_ll(got, want, what=""):
    """The parity contract on state log-likelihoods: |got - want| <= 1e-4 wherever the reference's
    float storage of the likelihood holds a value (ll > ln 2^-150).  Below that the reference emits
    the floor whatever the digits of ll, so there the assertion is the observable one: the engine's
    value flushes as well -- as a float likelihood it is 0 or at most the one denormal quantum that a
    1e-4 move across the edge produces."""
    import numpy as np
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), what
    err = np.abs(got - want)
    vis = want > LL_FLUSH
    if vis.any():
        worst = err[vis].max()
        assert worst <= TOL_LL, "%s: max |dll| %.3g at %s" % (
            what, worst, np.unravel_index(np.where(vis, err, -1.0).argmax(), err.shape))
    if (~vis).any():
        lik32 = np.exp(got[~vis]).astype(np.float32)
        assert (lik32 <= np.float32(2.0 ** -149)).all(), \
            "%s: a value the reference flushes to the floor is visible here (max ll %.6f)" % (what, got[~vis].max())
    return float(err[vis].max()) if vis.any() else 0.0


def assert_lp_denormal_band(lp_got, lik_ref, what=""):
    """Normalised 4-byte LNA values inside the float-denormal band of the reference's likelihood storage.

    aku/phone_probs.cc:224-236 keeps the LINEAR state likelihood in a float before it normalises: for
    ln 2^-149 < ll < ln 2^-126 that float is a denormal q * 2^-149, q = rint(lik * 2^149), and the value written is
    log(q * 2^-149 / Z).  A 1e-4 difference in ll moves q by up to 1e-4 * q + 1 (one quantum where the rounding
    falls the other way), i.e. up to log(1 + 1/q) in the logarithm -- 0.69 at q = 1 -- so the log-domain bar cannot
    hold there and the contract is stated on q itself: the engine's value, taken back to quanta with the oracle's own
    normaliser, is within one quantum (+ the 2e-4 relative that 1e-4 on ll and on Z allow) of the oracle's.
    `lik_ref`: the oracle's double state likelihoods [frames x states]; `lp_got`: the engine's float log-probs.
    Returns the number of band values checked."""
    import numpy as np
    lik_ref = np.asarray(lik_ref, np.float64)
    obs = lik_ref.astype(np.float32)                      # (float) state_likelihood, denormals kept
    z = obs.astype(np.float64).sum(axis=1, keepdims=True)
    z[z == 0] = 1.0
    quantum = 2.0 ** -149
    band = (obs < np.float32(2.0 ** -126)) & (lik_ref >= 1e-50)   # denormal or flushed by the float, not by the 1e-50 clamp
    if not band.any():
        return 0
    q_ref = obs.astype(np.float64) / quantum
    q_got = np.exp(np.asarray(lp_got, np.float64)) * z / quantum
    bad = band & (np.abs(q_got - q_ref) > 1.01 + 2e-4 * q_ref)
    assert not bad.any(), "%s: %d of %d denormal-band values are more than one quantum off (first at %s: %.3f against %.0f quanta)" % (
        what, int(bad.sum()), int(band.sum()), np.argwhere(bad)[0], q_got[bad][0], q_ref[bad][0])
    return int(band.sum())


def observed(label, value, bound):
    """Assert `value >= bound` and, with AASR_PRINT_OBSERVED=1, print what was observed (used to keep
    the thresholds of the LNA code-equality tests at what the hardware actually delivers)."""
    import os
    if os.environ.get("AASR_PRINT_OBSERVED") == "1":
        print("OBSERVED %s %.6f (bound %.4f)" % (label, value, bound))
    assert value >= bound, (label, value, bound)
