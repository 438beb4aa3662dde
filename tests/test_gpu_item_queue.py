"""GPU: the pair scan's item queue (a resident grid whose blocks draw items of the launch's list from a device counter,
ScanArgs::dyn in hm_common.h) returns what the one-block-per-item grid returns, bit for bit, and what the oracle returns.

The queue is the default (both prefilter forms) when a launch has more items than the device holds blocks; the `dyn_slots`
knob shrinks the resident grid so that small tables exercise it too (few blocks, each drawing many items, empty items
left of the diagonal included)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hyptokenizer_amd.synthetic import lorentz_table  # noqa: E402


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


FORM = "bf16"


@pytest.fixture(params=["bf16", "f32"], autouse=True)
def prefilter_form(request):
    """both prefilter forms of the scan kernel (two sets of instantiations of the same queue code)"""
    global FORM
    FORM = request.param
    yield request.param


def _engine(X, knobs, max_rows=None, mode="lorentz"):
    from hyptokenizer_amd import _lib
    from hyptokenizer_amd.engine import MergeEngine
    L = _lib.load()
    n, d1 = X.shape
    max_rows = max_rows or n + 64
    table = torch.zeros((max_rows, d1), dtype=torch.float32, device="cuda")
    table[:n] = X.cuda()
    eng = MergeEngine(max_rows, d1, mode, prefilter=FORM)
    for k, v in knobs.items():
        _lib.check(L.hm_debug_set_knob(eng._h, k.encode(), float(v)))
    eng.set_table(table, n)
    return eng, table


QUEUE = {"dyn": 1, "dyn_slots": 8}
STATIC = {"dyn": 0}


@pytest.mark.parametrize("n,d,thr", [(700, 100, 0.45), (3000, 100, 0.42), (2500, 16, 0.3), (6000, 50, 0.40), (20000, 100, 0.41)])
def test_queue_searches_equal_static_grid_and_oracle(oracle, n, d, thr):
    X = lorentz_table(n, d, seed=100 + n, scale=0.05)
    eq, _ = _engine(X, QUEUE if n < 20000 else {"dyn": 1})          # (the largest case: the queue at its real size)
    es, _ = _engine(X, STATIC)
    Xo = X.numpy()
    for t in (thr, thr * 1.15):
        a, b = eq.argmin(1.0, t), es.argmin(1.0, t)
        assert a == b
        gd, gi, gj, gc = eq.topk(1.0, t, 2000)
        sd, si, sj, sc = es.topk(1.0, t, 2000)
        assert gc == sc and np.array_equal(gi, si) and np.array_equal(gj, sj) and np.array_equal(_bits(gd), _bits(sd))
        if n <= 6000:
            od, oi, oj, oc = oracle.pairwise_topk(Xo, n, 1.0, float(np.float32(t)), 1, 2000)
            assert gc == oc and np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(_bits(gd), _bits(od))
            if oc:
                assert a is not None and (a[1], a[2]) == (int(oi[0]), int(oj[0])) and _bits([a[0]])[0] == _bits(od[:1])[0]
    # a row range and the count-free refresh form
    lo, hi = n // 5, n // 2
    assert eq.argmin(1.0, thr * 1.15, row_begin=lo, row_end=hi) == es.argmin(1.0, thr * 1.15, row_begin=lo, row_end=hi)
    q = eq.topk(1.0, thr * 1.15, 500, count=False)
    s = es.topk(1.0, thr * 1.15, 500, count=False)
    assert np.array_equal(q[1], s[1]) and np.array_equal(q[2], s[2]) and np.array_equal(_bits(q[0]), _bits(s[0]))


def test_queue_in_the_device_loops_both_counter_sets():
    """The pipelined standard loop alternates two counter sets (two queue words); the sequential chain uses one.  Same
    merges, same rows as the static grid."""
    n, d, steps = 9000, 48, 30
    X = lorentz_table(n, d, seed=23, scale=0.05)
    runs = []
    for knobs in ({"dyn": 0, "pipeline_pairs": 0}, {"dyn": 1, "dyn_slots": 16, "pipeline_pairs": 0}, {"dyn": 1, "dyn_slots": 16, "pipeline": 0}):
        eng, table = _engine(X, knobs, max_rows=n + steps + 8)
        eng.set_token_lengths(np.arange(1, n + 1, dtype=np.int32) % 5 + 1)
        recs, done = eng.std_merge_steps(1.0, 0.6, table, steps)
        assert done == steps
        runs.append(([r[2:] for r in recs], [_bits([r[1]])[0] for r in recs], table[n:n + steps].cpu().numpy().view(np.uint32).copy()))
    for other in runs[1:]:
        assert other[0] == runs[0][0] and other[1] == runs[0][1] and np.array_equal(other[2], runs[0][2])


def test_queue_word_survives_zeroing_and_many_launches():
    """The queue word carries the launch tag: nothing resets it between launches, and searches of different sizes and
    modes on one engine keep returning the static grid's answers."""
    n, d = 5000, 100
    X = lorentz_table(n, d, seed=5, scale=0.05)
    eq, _ = _engine(X, QUEUE)
    es, _ = _engine(X, STATIC)
    for rep in range(12):
        t = 0.40 + 0.01 * (rep % 4)
        assert eq.argmin(1.0, t) == es.argmin(1.0, t)
        if rep % 3 == 0:
            a, b = eq.topk(1.0, t, 300), es.topk(1.0, t, 300)
            assert a[3] == b[3] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        if rep % 4 == 1:
            r0, r1 = 100 * rep, 100 * rep + 2000
            assert eq.argmin(1.0, t, row_begin=r0, row_end=r1) == es.argmin(1.0, t, row_begin=r0, row_end=r1)
