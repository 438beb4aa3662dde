"""GPU parity: the HIP engine through the C ABI vs the CPU oracle, bit for bit.

Integer results (pair indices, counts) and -- because both sides implement the same canonical
fp32 arithmetic -- distances and midpoint rows are compared for exact equality.
The 1e-5 tolerance of the north star applies between the oracle and the torch reference
(tests/test_oracle_golden.py); here any difference at all is a failure.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hyptokenizer_amd.synthetic import lorentz_table  # noqa: E402

MODES = {"reference": 0, "lorentz": 1}


@pytest.fixture(params=["f32", "bf16", "bf16-512", "bf16-k112"], autouse=True)
def prefilter_form(request, monkeypatch):
    """Every test in this module runs four times: exact fp32 MFMA prefilter, bf16 MFMA prefilter, the bf16
    prefilter in its large-table shape (512-row blocks, forced here at every size), and the bf16 prefilter on image
    rows padded to whole 16-slot k-steps (d = 100: 14 chunks instead of 13 + a half step) -- `hm_debug_set_default_knob`
    applies to every engine created afterwards.  Results must be identical: the prefilter only selects
    survivors, the canonical arithmetic decides."""
    from hyptokenizer_amd import _lib
    L = _lib.load()
    monkeypatch.setenv("HM_SCAN_PRECISION", request.param.split("-")[0])
    _lib.check(L.hm_debug_set_default_knob(None, 0.0, 1))
    if request.param.endswith("-512"):
        _lib.check(L.hm_debug_set_default_knob(b"big_rows", 2.0, 0))
    if request.param.endswith("-k112"):
        _lib.check(L.hm_debug_set_default_knob(b"kc_even", 1.0, 0))
    yield request.param
    _lib.check(L.hm_debug_set_default_knob(None, 0.0, 1))


def _engine(X, mode, max_rows=None):
    from hyptokenizer_amd.engine import MergeEngine
    n, d1 = X.shape
    max_rows = max_rows or n + 512
    table = torch.zeros((max_rows, d1), dtype=torch.float32, device="cuda")
    table[:n] = X.cuda()
    eng = MergeEngine(max_rows, d1, mode)
    eng.set_table(table, n)
    return eng, table


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _quantile_thr(oracle, X, q_pairs, mode):
    """threshold with about q_pairs candidates (lorentz mode)"""
    n = X.shape[0]
    D = oracle.batch_distance(X[:min(n, 600)], X[:min(n, 600)], 1.0, MODES[mode])
    iu = np.triu_indices(D.shape[0], 1)
    dd = np.sort(D[iu])
    frac = q_pairs / (n * (n - 1) / 2)
    return float(dd[min(len(dd) - 1, max(0, int(frac * len(dd))))])


@pytest.mark.parametrize("n,d,scale", [(64, 10, 0.05), (257, 10, 0.05), (1000, 10, 0.05), (700, 50, 0.05),
                                       (1500, 100, 0.05), (300, 5, 0.01), (999, 100, 0.5), (513, 37, 0.05)])
@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_topk_matches_oracle(oracle, n, d, scale, mode):
    X = lorentz_table(n, d, seed=42, scale=scale).numpy()
    eng, _ = _engine(torch.from_numpy(X), mode)
    thrs = [0.1, 1e-5, 10.0] if mode == "reference" else [
        _quantile_thr(oracle, X, 50, mode), _quantile_thr(oracle, X, 5000, mode), _quantile_thr(oracle, X, 60000, mode), 1e-9]
    for thr in thrs:
        for k in (1, 100, 10000):
            gd, gi, gj, gc = eng.topk(1.0, thr, k)
            od, oi, oj, oc = oracle.pairwise_topk(X, n, 1.0, thr, MODES[mode], k)
            assert gc == oc, (thr, k, gc, oc)
            assert np.array_equal(gi, oi) and np.array_equal(gj, oj), (thr, k)
            assert np.array_equal(_bits(gd), _bits(od)), (thr, k)


@pytest.mark.parametrize("n,d,scale", [(257, 10, 0.05), (1000, 10, 0.05), (1500, 100, 0.05), (2100, 50, 0.05), (900, 124, 0.05),
                                       (900, 128, 0.05), (600, 1, 0.3)])
@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_argmin_matches_oracle(oracle, n, d, scale, mode):
    X = lorentz_table(n, d, seed=43, scale=scale).numpy()
    eng, _ = _engine(torch.from_numpy(X), mode)
    for thr in (0.05, 0.3, 5.0, 1e-7):
        got = eng.argmin(1.0, thr)
        od, oi, oj, oc = oracle.pairwise_topk(X, n, 1.0, thr, MODES[mode], 1)
        if oc == 0:
            assert got is None
        else:
            assert got is not None
            assert (got[1], got[2]) == (int(oi[0]), int(oj[0]))
            assert _bits([got[0]])[0] == _bits(od)[0]
    # curvature rescales the distance only
    got = eng.argmin(2.5, 0.3)
    od, oi, oj, oc = oracle.pairwise_topk(X, n, 2.5, 0.3, MODES[mode], 1)
    assert (got is None) == (oc == 0)
    if got:
        assert (got[1], got[2]) == (int(oi[0]), int(oj[0])) and _bits([got[0]])[0] == _bits(od)[0]


@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_row_ranges_partition(oracle, mode):
    """row-sharded scans (multi-GPU decomposition) agree with the oracle on every range"""
    n, d = 1300, 20
    X = lorentz_table(n, d, seed=7, scale=0.05).numpy()
    eng, _ = _engine(torch.from_numpy(X), mode)
    thr = 0.25 if mode == "lorentz" else 0.1
    total = 0
    for (r0, r1) in [(0, 256), (256, 300), (300, 1024), (1024, 1300), (5, 6), (1299, 1300)]:
        gd, gi, gj, gc = eng.topk(1.0, thr, 500, r0, r1)
        od, oi, oj, oc = oracle.pairwise_topk(X, n, 1.0, thr, MODES[mode], 500, r0, r1)
        assert gc == oc and np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(_bits(gd), _bits(od))
        if (r0, r1) != (5, 6) and (r0, r1) != (1299, 1300):
            total += gc
        a = eng.argmin(1.0, thr, r0, r1)
        assert (a is None) == (oc == 0)
        if a:
            assert (a[1], a[2]) == (int(oi[0]), int(oj[0]))
    assert total == oracle.pairwise_count(X, n, 1.0, thr, MODES[mode])


@pytest.mark.parametrize("d", [5, 10, 50, 100])
@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_gathered_kernels_match_oracle(oracle, d, mode):
    n = 400
    X = lorentz_table(n, d, seed=11, scale=0.05).numpy()
    X[17] = X[16]                      # identical rows: NaN tangent
    eng, _ = _engine(torch.from_numpy(X), mode)
    rng = np.random.default_rng(0)
    I = rng.integers(0, n, 300).astype(np.int32)
    J = rng.integers(0, n, 300).astype(np.int32)
    I[:2], J[:2] = [16, 5], [17, 5]
    W = rng.uniform(0.05, 0.95, 300).astype(np.float32)
    for c in (1.0, 0.7):
        assert np.array_equal(_bits(eng.pair_distance(I, J, c)), _bits(oracle.pair_distance(X, I, J, c, MODES[mode])))
        g = eng.midpoint(I, J, W, c).cpu().numpy()
        o = oracle.midpoint_batch(X, I, J, W, c, MODES[mode])
        assert np.array_equal(_bits(g), _bits(o))
        assert np.array_equal(_bits(eng.row_vs_all(3, n, c)), _bits(oracle.row_vs_all(X, n, 3, c, MODES[mode])))


@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_merge_loop_sequence_bit_exact(oracle, mode):
    """argmin -> fused midpoint/append, 60 steps: pair sequence and every new row identical"""
    n0, d = 600, 10
    X = lorentz_table(n0, d, seed=42, scale=0.05).numpy()
    eng, table = _engine(torch.from_numpy(X), mode, max_rows=n0 + 100)
    Xo = np.zeros((n0 + 100, d + 1), np.float32)
    Xo[:n0] = X
    n = n0
    thr = 0.1
    for step in range(60):
        got = eng.argmin(1.0, thr)
        od, oi, oj, oc = oracle.pairwise_topk(Xo, n, 1.0, thr, MODES[mode], 1)
        assert (got is None) == (oc == 0)
        if got is None:
            break
        assert (got[1], got[2]) == (int(oi[0]), int(oj[0])), step
        w = np.float32(0.5 if step % 3 else 1.0 / 3.0)
        eng.merge_append(got[1], got[2], float(w), 1.0, table, n)
        Xo[n] = oracle.midpoint_batch(Xo, [got[1]], [got[2]], [w], 1.0, MODES[mode])[0]
        assert np.array_equal(_bits(table[n].cpu().numpy()), _bits(Xo[n])), step
        n += 1
        assert eng.n == n


@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_banded_decomposition_matches_oracle(oracle, mode):
    """large enough for the XCD-banded / two-phase work decomposition (n >= 4096): counts, ordered
    lists and row-range searches against the oracle"""
    n, d = 6200, 20
    X = lorentz_table(n, d, seed=5, scale=0.05).numpy()
    eng, _ = _engine(torch.from_numpy(X), mode)
    thr = 0.2 if mode == "lorentz" else 0.1
    for (r0, r1, k) in [(0, -1, 5000), (0, -1, 1), (1000, 5000, 3000), (6000, 6200, 50), (0, 130, 400)]:
        rr1 = n if r1 < 0 else r1
        gd, gi, gj, gc = eng.topk(1.0, thr, k, r0, r1)
        od, oi, oj, oc = oracle.pairwise_topk(X, n, 1.0, thr, MODES[mode], k, r0, rr1, fast=True)
        assert gc == oc, (r0, r1, gc, oc)
        assert np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(_bits(gd), _bits(od)), (r0, r1)
        a = eng.argmin(1.0, thr, r0, r1)
        assert (a is None) == (oc == 0)
        if a:
            assert (a[1], a[2]) == (int(oi[0]), int(oj[0]))


@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_argmin_seed_follows_table_edits(oracle, mode):
    """the search seeds its running key with the previous nearest pair (valid while rows are only
    appended); editing or replacing a row must drop the seed, appends and range searches must not be
    misled by it"""
    n, d = 2600, 30
    X = lorentz_table(n, d, seed=21, scale=0.05).numpy()
    eng, table = _engine(torch.from_numpy(X), mode, max_rows=n + 64)
    thr = 0.6 if mode == "lorentz" else 0.1

    def check(live, r0=0, r1=-1):
        got = eng.argmin(1.0, thr, r0, r1)
        od, oi, oj, oc = oracle.pairwise_topk(live, live.shape[0], 1.0, thr, MODES[mode], 1, r0, live.shape[0] if r1 < 0 else r1)
        assert (got is None) == (oc == 0), (r0, r1)
        if got:
            assert (got[1], got[2]) == (int(oi[0]), int(oj[0])) and _bits([got[0]])[0] == _bits(od)[0], (r0, r1)
        return got

    first = check(X)
    assert check(X) == first                                     # seeded repeat
    assert first is not None
    i, j = first[1], first[2]
    # ranges that do / do not contain the seed's row
    check(X, i, i + 1)
    check(X, min(i + 1, n - 2), n)
    check(X, 0, max(i, 1))
    # move one end of the nearest pair far away: the old key is no bound any more
    far = lorentz_table(1, d, seed=99, scale=1.0).numpy()[0]
    X2 = X.copy()
    X2[j] = far
    table[j] = torch.from_numpy(far).cuda()
    eng.update_rows(table, j, j + 1)
    second = check(X2)
    assert mode == "reference" or second != first       # literal mode: every distance is 0, (0, 1) stays first
    # a whole new table
    X3 = lorentz_table(n, d, seed=22, scale=0.05).numpy()
    table[:n] = torch.from_numpy(X3).cuda()
    eng.set_table(table, n)
    check(X3)
    # appends keep the seed: duplicate of a row -> distance 0 must win over the seeded key
    table[n] = table[17]
    eng.update_rows(table, n, n + 1)
    X4 = np.concatenate([X3, X3[17:18]], 0)
    got = check(X4)
    if mode == "lorentz":
        assert got is not None and (got[1], got[2]) == (17, n) and got[0] == 0.0
    check(X4)


@pytest.mark.parametrize("n", [2, 3, 31, 65])
@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_tiny_tables(oracle, n, mode):
    """tables smaller than one MFMA tile / one row block"""
    d = 30
    X = lorentz_table(n, d, seed=n, scale=0.05).numpy()
    eng, _ = _engine(torch.from_numpy(X), mode)
    for thr in (0.2, 50.0):
        od, oi, oj, oc = oracle.pairwise_topk(X, n, 1.0, thr, MODES[mode], 64)
        gd, gi, gj, gc = eng.topk(1.0, thr, 64)
        assert gc == oc and np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(_bits(gd), _bits(od))
        a = eng.argmin(1.0, thr)
        assert (a is None) == (oc == 0)
        if a:
            assert (a[1], a[2]) == (int(oi[0]), int(oj[0])) and _bits([a[0]])[0] == _bits(od)[0]


def test_all_rows_identical_lorentz(oracle):
    """every pair at distance 0 in the non-degenerate sign mode: the zero-class tie flood orders by (i, j)"""
    n, d = 2500, 30
    row = lorentz_table(1, d, seed=3, scale=0.05)
    X = row.repeat(n, 1).numpy()
    eng, _ = _engine(torch.from_numpy(X), "lorentz")
    a = eng.argmin(1.0, 0.5)
    assert a == (0.0, 0, 1)
    assert eng.argmin(1.0, 0.5) == (0.0, 0, 1)                   # seeded repeat
    gd, gi, gj, gc = eng.topk(1.0, 0.5, 300)
    assert gc == n * (n - 1) // 2 and np.all(gd == 0.0)
    want = [(0, j) for j in range(1, 301)]
    assert list(zip(gi.tolist(), gj.tolist())) == want
    b = eng.argmin(1.0, 0.5, 1000, 2000)
    assert b == (0.0, 1000, 1001)


@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_unbounded_threshold(oracle, mode):
    """a threshold beyond every distance (cosh overflows to inf): every pair is a candidate"""
    n, d = 1800, 24
    X = lorentz_table(n, d, seed=8, scale=0.05).numpy()
    eng, _ = _engine(torch.from_numpy(X), mode)
    for thr in (1.0e4, 3.0e38):
        od, oi, oj, oc = oracle.pairwise_topk(X, n, 1.0, thr, MODES[mode], 700, fast=True)
        gd, gi, gj, gc = eng.topk(1.0, thr, 700)
        assert gc == oc == n * (n - 1) // 2
        assert np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(_bits(gd), _bits(od))
        a = eng.argmin(1.0, thr)
        assert (a[1], a[2]) == (int(oi[0]), int(oj[0])) and _bits([a[0]])[0] == _bits(od)[0]


def test_armed_search_survives_interleaved_calls(oracle):
    """an argmin search leaves the counters and the running key armed for the next search of the same range;
    every other entry point that uses them must disarm"""
    n, d = 3000, 30
    X = lorentz_table(n, d, seed=17, scale=0.05).numpy()
    eng, table = _engine(torch.from_numpy(X), "lorentz", max_rows=n + 64)
    thr = 0.6
    od, oi, oj, oc = oracle.pairwise_topk(X, n, 1.0, thr, 1, 1)
    want = (float(od[0]), int(oi[0]), int(oj[0]))
    assert eng.argmin(1.0, thr) == want
    assert eng.argmin(1.0, thr) == want                           # armed
    eng.row_argmin(5, 5, 1.0, thr)
    assert eng.argmin(1.0, thr) == want
    eng.topk(1.0, thr, 50)
    assert eng.argmin(1.0, thr) == want
    eng.candidates(1.0, 0.3)
    assert eng.argmin(1.0, thr) == want
    rec = torch.zeros(4, dtype=torch.int32, device="cuda")
    eng.argmin_into(1.0, thr, 0, -1, rec)
    torch.cuda.synchronize()
    assert rec.tolist()[0] == 1 and tuple(rec.tolist()[2:]) == want[1:]
    assert eng.argmin(1.0, thr) == want
    assert eng.argmin(1.0, thr, 100, 2000) != want or want[1] >= 100   # another range: not armed for it
    assert eng.argmin(1.0, thr) == want
    assert eng.argmin(2.0, thr)[1:] == oracle_best(oracle, X, n, 2.0, thr)   # other curvature, same armed state
    assert eng.argmin(1.0, 1e-6) is None                          # nothing below this threshold
    assert eng.argmin(1.0, thr) == want


def oracle_best(oracle, X, n, c, thr):
    od, oi, oj, oc = oracle.pairwise_topk(X, n, c, thr, 1, 1)
    return (int(oi[0]), int(oj[0]))


def test_threshold_inside_the_bulk_of_the_distances(oracle):
    """half of all pairs are candidates: the bf16 prefilter's undecided shell around the threshold outgrows the
    emission buffer and the search falls back to the fp32 prefilter; counts and lists stay exact"""
    n, d = 9000, 50
    X = lorentz_table(n, d, seed=12, scale=0.05).numpy()
    eng, _ = _engine(torch.from_numpy(X), "lorentz")
    s = eng.pair_distance(np.arange(0, 2000), np.arange(2000, 4000), 1.0)
    thr = float(np.median(s))
    od, oi, oj, oc = oracle.pairwise_topk(X, n, 1.0, thr, 1, 2000, fast=True)
    assert oc > 0.3 * n * (n - 1) / 2
    gd, gi, gj, gc = eng.topk(1.0, thr, 2000)
    assert gc == oc and np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(_bits(gd), _bits(od))
    for _ in range(2):
        a = eng.argmin(1.0, thr)
        assert (a[1], a[2]) == (int(oi[0]), int(oj[0])) and _bits([a[0]])[0] == _bits(od)[0]
    assert eng.scan_stats()["emitted"] < 100000                   # seeded + key read at block start: no first-tile flood


# ---------------------------------------------------------------------------------------------------------
# round 2: batched merges, device-resident loops, uncounted top-k, the acceptance test of the emission cut,
# golden candidate lists and edge fixtures straight on the HIP engine
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_merge_batch_equals_sequential_merges(oracle, mode):
    """hm_merge_append_batch: a chain of merges (later ones read rows written by earlier ones) in one launch"""
    n, d = 900, 33
    X = lorentz_table(n, d, seed=6, scale=0.05)
    engA, tabA = _engine(X, mode, n + 64)
    engB, tabB = _engine(X, mode, n + 64)
    I = np.array([3, 900, 901, 5, 902, 902], np.int32)
    J = np.array([9, 4, 900, 901, 7, 902], np.int32)
    W = np.array([0.5, 0.25, 2 / 3, 0.5, 0.1, 0.5], np.float32)
    engA.merge_append_batch(I, J, W, 1.3, tabA, n)
    for t in range(len(I)):
        engB.merge_append(int(I[t]), int(J[t]), float(W[t]), 1.3, tabB, n + t)
    assert engA.n == engB.n == n + len(I)
    assert np.array_equal(_bits(tabA.cpu().numpy()), _bits(tabB.cpu().numpy()))
    thr = 0.4
    assert engA.argmin(1.3, thr) == engB.argmin(1.3, thr)
    a, b = engA.topk(1.3, thr, 50), engB.topk(1.3, thr, 50)
    assert a[3] == b[3] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(_bits(a[0]), _bits(b[0]))
    engA.truncate(n + 2)
    assert engA.n == n + 2
    # independent form (every operand below first_row): all merges at once, same rows
    engC, tabC = _engine(X, mode, n + 64)
    engD, tabD = _engine(X, mode, n + 64)
    I2 = np.array([3, 5, 7, 3, 800], np.int32)
    J2 = np.array([9, 6, 899, 9, 2], np.int32)
    W2 = np.array([0.5, 0.25, 0.75, 0.5, 0.3], np.float32)
    engC.merge_append_batch(I2, J2, W2, 1.3, tabC, n, independent=True)
    engD.merge_append_batch(I2, J2, W2, 1.3, tabD, n, independent=False)
    assert np.array_equal(_bits(tabC.cpu().numpy()), _bits(tabD.cpu().numpy()))
    assert engC.topk(1.3, thr, 20)[1].tolist() == engD.topk(1.3, thr, 20)[1].tolist()


@pytest.mark.parametrize("mode", ["lorentz", "reference"])
@pytest.mark.parametrize("n,d,thr", [(1200, 50, 0.42), (300, 10, 0.2), (2100, 100, 0.6)])
def test_device_resident_loops_equal_the_oracle_loop(oracle, mode, n, d, thr):
    """hm_std_merge_steps / hm_incr_merge_steps: K steps per call, records and rows bit-equal to the oracle's
    step-by-step loop (lengths-driven weights, duplicates, NaN rows of the literal mode)"""
    steps = 70
    X = lorentz_table(n, d, seed=21, scale=0.05)
    lens = [1 + (k % 3) for k in range(n)]
    # oracle loop
    Xo = np.zeros((n + 128, d + 1), np.float32)
    Xo[:n] = X.numpy()
    ol = list(lens)
    want = []
    cur = n
    for _ in range(steps):
        od, oi, oj, oc = oracle.pairwise_topk(Xo, cur, 1.0, float(np.float32(thr)), MODES[mode], 1)
        if oc == 0:
            break
        i, j = int(oi[0]), int(oj[0])
        w = np.float32(ol[j] / (ol[i] + ol[j]))
        Xo[cur] = oracle.midpoint_batch(Xo, [i], [j], [w], 1.0, MODES[mode])[0]
        ol.append(ol[i] + ol[j])
        want.append((int(_bits(od[:1])[0]), i, j))
        cur += 1
    for kind in ("std", "incr"):
        eng, table = _engine(X, mode, n + 128)
        eng.set_token_lengths(lens)
        got = []
        best = eng.argmin(1.0, thr) if kind == "incr" else None
        left = steps
        while left > 0:
            k = min(left, 64)
            if kind == "std":
                recs, done = eng.std_merge_steps(1.0, thr, table, k)
            else:
                recs, done, best = eng.incr_merge_steps(1.0, thr, table, k, best)
            got += [(int(_bits([r[1]])[0]), r[2], r[3]) for r in recs[:done]]
            left -= done
            if done < k:
                assert recs[done][0] in (0, 2)
                if recs[done][0] == 2:                     # overflow (tie flood): this step through the host path
                    a = eng.argmin(1.0, thr)
                    li, lj = ol[a[1]], ol[a[2]]
                    eng.merge_append(a[1], a[2], lj / (li + lj), 1.0, table, eng.n)
                    eng.set_token_lengths(ol[:eng.n])
                    got.append((int(_bits([a[0]])[0]), a[1], a[2]))
                    left -= 1
                    if kind == "incr":
                        best = eng.argmin(1.0, thr)
                else:
                    break
        assert got == want, kind
        assert eng.n == cur
        assert np.array_equal(_bits(table[:cur].cpu().numpy()), _bits(Xo[:cur])), kind
        # the engine is in a consistent state afterwards: a plain search agrees with the oracle
        od, oi, oj, oc = oracle.pairwise_topk(Xo, cur, 1.0, float(np.float32(thr)), MODES[mode], 5)
        gd, gi, gj, gc = eng.topk(1.0, thr, 5)
        assert gc == oc and np.array_equal(gi, oi) and np.array_equal(gj, oj)


def test_uncounted_topk_and_late_count(oracle):
    """topk(count=False) returns the same list; the exact total can be asked for later -- also after rows were
    appended -- and equals the count of the table as it was (n_limit)"""
    n, d = 4000, 60
    X = lorentz_table(n, d, seed=9, scale=0.05)
    eng, table = _engine(X, "lorentz", n + 64)
    s = eng.pair_distance(np.arange(0, 1500), np.arange(1500, 3000), 1.0)
    for q in (0.05, 2.0, 40.0):
        thr = float(np.percentile(s, q))
        full = eng.topk(1.0, thr, 500)
        lazy = eng.topk(1.0, thr, 500, count=False)
        assert np.array_equal(full[1], lazy[1]) and np.array_equal(full[2], lazy[2]) and np.array_equal(_bits(full[0]), _bits(lazy[0]))
        assert lazy[3] in (-1, full[3]) and (lazy[3] == full[3] or full[3] >= 500)
        assert eng.count_candidates(1.0, thr) == full[3]
        eng.merge_append(int(full[1][0]), int(full[2][0]), 0.5, 1.0, table, eng.n)
        assert eng.count_candidates(1.0, thr, n_limit=n) == full[3] or eng.n != n + 1
        again = eng.topk(1.0, thr, 500, count=False)       # predicted cut of the previous refresh
        oracle_d, oi, oj, oc = oracle.pairwise_topk(table.cpu().numpy(), eng.n, 1.0, float(np.float32(thr)), 1, 500)
        assert np.array_equal(again[1], oi) and np.array_equal(again[2], oj) and np.array_equal(_bits(again[0]), _bits(oracle_d))
        eng.truncate(n)


def test_forced_tight_cut_is_detected(oracle):
    """a cut that is too tight must not be accepted just because the undecided shell around a bulk
    threshold holds >= k valid entries.  The next search is forced to start from a cut far below the k-th entry."""
    n, d = 3000, 40
    X = lorentz_table(n, d, seed=13, scale=0.05)
    eng, table = _engine(X, "lorentz", n + 8)
    s = eng.pair_distance(np.arange(0, 1000), np.arange(1000, 2000), 1.0)
    thr = float(np.percentile(s, 35.0))                    # a threshold in the bulk: large undecided shell under bf16
    k = 2000
    od, oi, oj, oc = oracle.pairwise_topk(X.numpy(), n, 1.0, float(np.float32(thr)), 1, k)
    u_small = np.float32(np.cosh(float(od[5])))            # cut at the 6th entry's u: far fewer than k below it
    eng.debug_force_cut(int(u_small.view(np.uint32)), k, 1.0)
    gd, gi, gj, gc = eng.topk(1.0, thr, k)
    assert gc == oc and np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(_bits(gd), _bits(od))
    eng.debug_force_cut(int(u_small.view(np.uint32)), k, 1.0)
    gd, gi, gj, gc = eng.topk(1.0, thr, k, count=False)
    assert np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(_bits(gd), _bits(od))


@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_golden_candidate_lists_on_the_hip_engine(golden_dir, mode):
    """G2 (reference's own _find_merge_candidates / _find_merge_candidates_fast lists) fed to the HIP engine
    directly: identical (i, j) lists and counts, distances within 1e-5 (hyperbolic_merge.py:247-269,378,
    fast...:91-95)"""
    import os
    z = np.load(os.path.join(golden_dir, f"g2_candidates_{mode}.npz"))
    for (n, d) in [(64, 10), (101, 10), (257, 10), (1000, 10), (300, 50)]:
        X = torch.from_numpy(z[f"n{n}_d{d}_X"])
        thrs = z[f"n{n}_d{d}_thr"]
        eng, _ = _engine(X, mode)
        for ti, thr in enumerate(thrs.tolist()):
            key = f"n{n}_d{d}_t{ti}"
            if f"{key}_std_count" not in z.files:
                continue
            t32 = float(np.float32(thr))
            if n <= 100:                                   # the reference's double-compare branch (:270-289)
                t = np.float32(thr)
                t32 = float(t if float(t) >= thr else np.nextafter(t, np.float32(np.inf)))
            ci, cj, cd, total = eng.candidates(1.0, t32)
            assert total == int(z[f"{key}_std_count"]), key
            m = len(z[f"{key}_std_i"])
            assert np.array_equal(ci[:m], z[f"{key}_std_i"]) and np.array_equal(cj[:m], z[f"{key}_std_j"]), key
            assert np.allclose(cd[:m], z[f"{key}_std_d"], atol=1e-5), key
            fd, fi, fj, fc = eng.topk(1.0, t32, 10000)
            assert fc == int(z[f"{key}_fast_count"]), key
            mf = min(len(fi), len(z[f"{key}_fast_i"]))
            ref_d = z[f"{key}_fast_d"][:mf]
            # the reference's order is by ITS fp32 distances; ours by the canonical ones: identical wherever the
            # reference's neighbours differ by more than its own rounding noise -- compare as sets per distance tie group
            assert np.allclose(fd[:mf], ref_d, atol=1e-5), key
            same = (fi[:mf] == z[f"{key}_fast_i"][:mf]) & (fj[:mf] == z[f"{key}_fast_j"][:mf])
            if not same.all():
                bad = np.nonzero(~same)[0]
                assert np.all(np.abs(ref_d[bad] - fd[bad]) <= 1e-6), key      # only swaps inside the noise
                assert sorted(zip(fi[:mf].tolist(), fj[:mf].tolist())) == sorted(zip(z[f"{key}_fast_i"][:mf].tolist(), z[f"{key}_fast_j"][:mf].tolist())) or mf == 10000, key
            assert min(int(fc), 10000) == int(z[f"{key}_cache_len"]), key


@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_golden_edge_rows_on_the_hip_engine(oracle, golden_dir, mode):
    """G1 edge fixture (identical rows, the origin, a NaN row, a zero row) as an engine table: pair distances,
    candidate list and midpoints against the reference's values"""
    import os
    from helpers import nan_equal_close
    z = np.load(os.path.join(golden_dir, f"g1_primitives_{mode}.npz"))
    E = torch.from_numpy(z["edge_X"])
    eng, table = _engine(E, mode, 64)
    a, b = z["edge_pairs_a"], z["edge_pairs_b"]
    got = eng.pair_distance(a, b, 1.0)
    assert nan_equal_close(got, z["edge_dist"], 1e-5)
    mid = eng.midpoint(a, b, np.full(len(a), 0.5, np.float32), 1.0).cpu().numpy()
    assert nan_equal_close(mid, z["edge_mid"], 1e-5)
    bd = z["edge_bd"]
    n = E.shape[0]
    ci, cj, cd, total = eng.candidates(1.0, 0.5)
    want = [(i, j) for i in range(n) for j in range(i + 1, n) if bd[i, j] < np.float32(0.5)]      # NaN never passes
    assert total == len(want) and list(zip(ci.tolist(), cj.tolist())) == want
    best = eng.argmin(1.0, 0.5)
    if want:
        order = sorted(want, key=lambda p: (float(bd[p]), p))
        assert best is not None and (best[1], best[2]) == order[0]
    else:
        assert best is None


def test_loop_timing_mode_changes_nothing_but_the_statistics(oracle):
    """hm_debug_time_loops (bench.py's instrumented batch): same records as the plain batch; the timing of a complete
    batch is consistent (scans inside the batch, every launch counted)."""
    from hyptokenizer_amd.engine import MergeEngine
    n, d = 3000, 40
    X = lorentz_table(n, d, seed=21, scale=0.05)
    runs = []
    for timed in (False, True):
        table = torch.zeros((n + 64, d + 1), device="cuda")
        table[:n] = X.cuda()
        eng = MergeEngine(n + 64, d + 1, "lorentz")
        eng.set_table(table, n)
        eng.set_token_lengths(np.ones(n, np.int32))
        eng.debug_time_loops(timed)
        eng.scan_totals(reset=True)
        recs, done = eng.std_merge_steps(1.0, 0.4, table, 20)
        runs.append((recs, done, table[n:n + 20].cpu()))
        if timed:
            t = eng.last_loop_timing()
            assert t["steps"] == 20 and 0.0 < t["scan_ms"] < t["batch_ms"]
            assert eng.scan_totals()["launches"] == 20
            eng.debug_time_loops(False)
    assert runs[0][1] == runs[1][1] == 20 and runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][2].view(torch.int32), runs[1][2].view(torch.int32))


def test_pipelined_loop_equals_sequential_loop_and_survives_an_order_fault():
    """The standard loop's software pipeline (a step's tail work under the next step's scan, two buffer sets, the newest
    row's pairs by a row pass) merges the same pairs into the same rows as the strictly sequential chain; a scan whose order
    guard trips (forced here through the `pipe_fault_at` knob) hands the remaining steps to the sequential path."""
    from hyptokenizer_amd import _lib
    from hyptokenizer_amd.engine import MergeEngine
    L = _lib.load()
    n, d, steps = 9000, 48, 40
    X = lorentz_table(n, d, seed=21, scale=0.05)
    runs = []
    for mode in ("seq", "pipe", "fault"):
        table = torch.zeros((n + steps + 8, d + 1), device="cuda")
        table[:n] = X.cuda()
        eng = MergeEngine(n + steps + 8, d + 1, "lorentz")
        _lib.check(L.hm_debug_set_knob(eng._h, b"pipeline", 0.0 if mode == "seq" else 1.0))
        _lib.check(L.hm_debug_set_knob(eng._h, b"pipeline_pairs", 0.0))              # (pipelined at this small size too: tails longer than scans trip the guard by themselves)
        if mode == "fault":
            _lib.check(L.hm_debug_set_knob(eng._h, b"pipe_fault_at", 17.0))
        eng.set_table(table, n)
        eng.set_token_lengths(np.arange(1, n + 1, dtype=np.int32) % 5 + 1)
        recs, done = eng.std_merge_steps(1.0, 0.6, table, steps)
        assert done == steps, (mode, done)
        # a second batch on the same engine (the sets are re-armed from the engine's seed)
        recs2, done2 = eng.std_merge_steps(1.0, 0.6, table, 3) if n + steps + 3 <= table.shape[0] else ([], 0)
        runs.append(([r[2:] for r in recs] + [r[2:] for r in recs2], [_bits([r[1]])[0] for r in recs], table[n:n + steps].cpu().numpy().view(np.uint32).copy(),
                     eng.argmin(1.0, 0.6)))
    for other in runs[1:]:
        assert other[0] == runs[0][0] and other[1] == runs[0][1] and np.array_equal(other[2], runs[0][2]) and other[3] == runs[0][3]
