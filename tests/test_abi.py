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
    assert L.hm_abi_version() == 3


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


def test_tokenize_rule_table_builder_is_host_only():
    """hm_tokenize_build_table needs no GPU: every rule is found by the kernel's probe sequence, a repeated pair keeps
    the LAST result (dict assignment, hyperbolic_merge.py:425-428)."""
    import numpy as np
    from hyptokenizer_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(0)
    n_sym, n_rules = 500, 2000
    left = rng.integers(0, n_sym, n_rules).astype(np.int32)
    right = rng.integers(0, n_sym, n_rules).astype(np.int32)
    merged = rng.integers(0, n_sym, n_rules).astype(np.int32)
    left[-1], right[-1] = left[0], right[0]                    # repeated pair
    cap = int(L.hm_tokenize_table_capacity(n_rules))
    assert cap >= 8 * n_rules and cap & (cap - 1) == 0
    table = np.empty(cap, dtype=np.uint64)
    assert L.hm_tokenize_build_table(left.ctypes.data, right.ctypes.data, merged.ctypes.data, n_rules, table.ctypes.data, cap) == 0
    want = {}
    for a, b, ab in zip(left.tolist(), right.tolist(), merged.tolist()):
        want[(a, b)] = ab
    n_buckets = cap // 2
    shift = 64 - (n_buckets.bit_length() - 1)
    tab = [int(x) for x in table]
    longest = 0
    for (a, b), ab in want.items():
        tag = (a << 21) | b
        s = ((tag * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)) >> shift
        steps = 1
        while True:
            e0, e1 = tab[2 * s], tab[2 * s + 1]
            hit = [e for e in (e0, e1) if e != 0 and e >> 22 == tag]
            if hit:
                assert (hit[0] & ((1 << 22) - 1)) - 1 == ab
                break
            assert e0 != 0 and e1 != 0          # a bucket with a free slot ends the kernel's probe sequence
            s = (s + 1) & (n_buckets - 1)
            steps += 1
        longest = max(longest, steps)
    assert sum(1 for e in tab if e != 0) == len(want) and longest <= 3
    # capacity that is too small or not a power of two is refused, and so are symbols beyond 21 bits
    assert L.hm_tokenize_build_table(left.ctypes.data, right.ctypes.data, merged.ctypes.data, n_rules, table.ctypes.data, 1024) == -1
    big = np.array([1 << 21], dtype=np.int32)
    assert L.hm_tokenize_build_table(big.ctypes.data, big.ctypes.data, big.ctypes.data, 1, table.ctypes.data, cap) == -1
    # no GPU here: the batch call has nothing to run on, but its argument check is reachable
    assert L.hm_tokenize_batch(None, None, None, 5, None, cap, None, None, None, None) == -1
