"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, exports
every symbol include/aasr.h declares, and fails loudly without a device."""
import ctypes as C

import numpy as np
import pytest


def test_all_declared_symbols_exported(capi):
    L = capi.lib()
    names = capi.declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_version_and_header(capi):
    assert b"gfx950" in capi.lib().aasr_version()
    assert capi.lna_header(3125, 2) == bytes([0, 0, 0x0c, 0x35, 2])
    assert capi.lna_header(256, 4) == bytes([0, 0, 1, 0, 4])


def test_no_cpu_fallback(capi):
    """Without a HIP device the compute entry points must refuse, not compute."""
    if capi.lib().aasr_device_count() > 0:
        pytest.skip("GPU present")
    from aaltoasr_amd import synth
    mean, var, off, idx, w = synth.make_model(D=4, G=8, S=2, comps=4)
    with pytest.raises(capi.AasrError) as ei:
        capi.Gmm.from_arrays(mean, var, off, idx, w)
    assert ei.value.code == capi.AASR_ERR_NO_DEVICE
    with pytest.raises(capi.AasrError) as ei:
        capi.lna_encode(np.zeros((2, 3), np.float32))
    assert ei.value.code == capi.AASR_ERR_NO_DEVICE


def test_argument_validation(capi):
    L = capi.lib()
    h = C.c_void_p()
    assert L.aasr_gmm_create_diag(0, 1, None, None, 1, None, None, None, C.byref(h)) == capi.AASR_ERR_INVALID
    assert L.aasr_lna_encode(None, 1, 3, 1, 3, None, None) == capi.AASR_ERR_INVALID
    assert b"lnabytes" in L.aasr_last_error()
