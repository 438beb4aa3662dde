"""C-ABI checks that need no GPU: the library loads, exports every symbol include/hypmerge.h
declares, and refuses to create an engine when no HIP device is present (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "hypmerge.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hm_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from hyptokenizer_amd import _lib
    assert _header_functions() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    from hyptokenizer_amd import _lib
    L = _lib.load()
    for name in _header_functions():
        assert hasattr(L, name), name
    assert L.hm_abi_version() == 2


def test_engine_creation_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from hyptokenizer_amd import _lib
    from hyptokenizer_amd.engine import HypMergeUnavailable, MergeEngine
    L = _lib.load()
    h = C.c_void_p(0)
    st = L.hm_engine_create(C.byref(h), 0, 1000, 11, 1, 0)
    assert st != 0 and not h.value
    assert b"no HIP device" in L.hm_last_error(None) or st > 0
    with pytest.raises(HypMergeUnavailable):
        MergeEngine(1000, 11, "lorentz")


def test_argument_errors_are_reported():
    from hyptokenizer_amd import _lib
    L = _lib.load()
    h = C.c_void_p(0)
    assert L.hm_engine_create(None, 0, 1000, 11, 1, 0) == _lib.HM_E_ARG
    assert L.hm_engine_create(C.byref(h), 0, 1000, 1, 1, 0) == _lib.HM_E_ARG          # d1 < 2
    assert L.hm_engine_create(C.byref(h), 0, 1000, 400, 1, 0) == _lib.HM_E_ARG        # d1 > 129
    assert L.hm_engine_create(C.byref(h), 0, 10 ** 7, 11, 1, 0) == _lib.HM_E_ARG      # rows > 131072
    assert L.hm_engine_create(C.byref(h), 0, 1000, 11, 7, 0) == _lib.HM_E_ARG      # sign mode
    assert L.hm_engine_create(C.byref(h), 0, 1000, 11, 1, 9) == _lib.HM_E_ARG      # prefilter form
    assert L.hm_set_table(None, None, 11, 5, None) == _lib.HM_E_ARG
    assert L.hm_rows(None) == -1
    assert L.hm_last_error(None)


def test_device_functions_refuse_cpu_tensors():
    import torch
    from hyptokenizer_amd.embedding import lorentz_model as LM
    from hyptokenizer_amd.engine import HypMergeUnavailable
    x = torch.zeros(3, 4)
    x[:, 0] = 1
    for fn in (lambda: LM.distance(x, x), lambda: LM.batch_distance(x, x), lambda: LM.log_map(x, x),
               lambda: LM.exp_map(x, x), lambda: LM.project_to_hyperboloid(x), lambda: LM.minkowski_dot(x, x)):
        with pytest.raises(HypMergeUnavailable):
            fn()
    with pytest.raises(ValueError):
        LM.set_sign_convention("bogus")
