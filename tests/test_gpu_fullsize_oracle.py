"""HIP engine against the oracle AT THE BASELINE SIZES (BASELINE.json configs 2-4: V = 50 000 / 100 000, d = 50 / 100,
both prefilter forms): nearest pair, ordered top-10 000 and exact candidate count bit for bit, and a 20-step merge loop
(pairs and merged rows).  The oracle's OpenMP form (`fast=True`: exact fp32 prefilter, canonical evaluation of the
survivors -- checked against the plain form in tests/test_oracle_golden.py) does one V = 50 000 search in well under a
second on the GPU box's host cores, so these are ordinary `-m gpu` tests, not bench-only checks."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import bits  # noqa: E402
from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V,d,prefilter", [(50000, 100, "bf16"), (50000, 100, "f32"), (50000, 50, "f32"), (100000, 100, "bf16")])
def test_search_at_benchmark_size_equals_oracle(oracle, V, d, prefilter):
    from hyptokenizer_amd.engine import MergeEngine
    X = lorentz_table(V, d, seed=42, scale=0.05)
    Xn = X.numpy()
    table = torch.zeros((V + 8, d + 1), device="cuda")
    table[:V] = X.cuda()
    eng = MergeEngine(V + 8, d + 1, "lorentz", prefilter=prefilter)
    eng.set_table(table, V)
    s = eng.pair_distance(np.arange(0, 4000), np.arange(4000, 8000), 1.0)
    # the bench's threshold and one inside the lower tail of the distance distribution (hundreds of thousands of candidates)
    for thr in (0.5, float(np.percentile(s, 0.05))):
        od, oi, oj, oc = oracle.pairwise_topk(Xn, V, 1.0, thr, 1, 10000, fast=True)
        a = eng.argmin(1.0, thr)
        if oc == 0:
            assert a is None
            continue
        assert a is not None and (a[1], a[2]) == (int(oi[0]), int(oj[0])) and bits([a[0]])[0] == bits(od)[0], (thr, a)
        dd, ii, jj, cnt = eng.topk(1.0, thr, 10000)
        assert cnt == oc, (thr, cnt, oc)
        assert np.array_equal(ii, oi) and np.array_equal(jj, oj) and np.array_equal(bits(dd), bits(od)), thr
        dd2, ii2, jj2, cnt2 = eng.topk(1.0, thr, 10000, count=False)             # the refresh form (predicted cut, no count)
        assert np.array_equal(ii2, oi) and np.array_equal(jj2, oj) and np.array_equal(bits(dd2), bits(od)), thr
        assert cnt2 in (-1, oc)
        # a row range of the same search (what one rank of a sharded run does)
        r0, r1 = V // 3, V // 3 + 7001
        pd, pi, pj, pc = oracle.pairwise_topk(Xn, V, 1.0, thr, 1, 2000, r0, r1, fast=True)
        gd, gi, gj, gc = eng.topk(1.0, thr, 2000, r0, r1)
        assert gc == pc and np.array_equal(gi, pi) and np.array_equal(gj, pj) and np.array_equal(bits(gd), bits(pd)), thr


@pytest.mark.parametrize("prefilter", ["bf16", "f32"])
def test_twenty_loop_steps_at_benchmark_size_equal_the_oracle_loop(oracle, prefilter):
    """BASELINE configs[1] / [2] as benched: HyperbolicTokenizer.optimize_merges at V = 50 000, d = 100 (device-resident
    batch) against search -> [0] -> midpoint repeated on the oracle: same pairs, same rows bit for bit."""
    from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
    V, d, thr, steps = 50000, 100, 0.5, 20
    X = lorentz_table(V, d, seed=42, scale=0.05)
    tok = HyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X.clone()), merge_threshold=thr, device=torch.device("cuda"),
                              max_vocab_size=V + steps + 8, sign_convention="lorentz", prefilter=prefilter)
    tok.optimize_merges(steps=steps, log_every=10 ** 9)
    assert len(tok.merge_history) == steps
    Xo = np.zeros((V + steps + 8, d + 1), np.float32)
    Xo[:V] = X.numpy()
    cur = V
    thr32 = float(np.float32(thr))
    for (a, b, ab) in tok.merge_history:
        hd, hi, hj, hc = oracle.pairwise_topk(Xo, cur, 1.0, thr32, 1, 1, fast=True)
        assert hc > 0 and (tok.vocab[int(hi[0])], tok.vocab[int(hj[0])]) == (a, b), (cur, a, b)
        Xo[cur] = oracle.midpoint_batch(Xo, hi[:1], hj[:1], [np.float32(len(b) / (len(a) + len(b)))], 1.0, 1)[0]
        cur += 1
    got = tok.embeddings.data[V:cur].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), Xo[V:cur].view(np.uint32))


def test_config5_at_benchmark_size_equals_the_oracle_engine(oracle):
    """BASELINE configs[4] as benched (`legs.config5_enhanced`: EnhancedFastHyperbolicTokenizer, V = 100 000, d = 100,
    frequency-aware scoring + adaptive curvature) against the SAME host class driven by the oracle's engine double
    (tests/helpers.py OracleEngine: top-k search, coherence distances, curvature-loss distances and the whole-table
    re-projection all from oracle/): same candidates with the same scores, same merges into the same rows bit for bit, same
    curvature after the curvature step, same generator states.  The refresh scores every candidate (919 here)."""
    import random
    from helpers import OracleEngine
    from hyptokenizer_amd.engine import MergeEngine
    from hyptokenizer_amd.tokenizer.enhanced_fast_hyperbolic_merge import EnhancedFastHyperbolicTokenizer
    V, d, thr, steps = 100000, 100, 0.4636, 7
    X = lorentz_table(V, d, seed=42, scale=0.05)
    vocab = cjk_vocab(V)
    rs = np.random.RandomState(42)
    a, b = rs.randint(0, V, 200000), rs.randint(0, V, 200000)
    cnt = rs.zipf(1.2, 200000).clip(max=10 ** 6)
    freq = {(vocab[i], vocab[j]): int(c) for i, j, c in zip(a.tolist(), b.tolist(), cnt.tolist())}
    rows = V + steps + 64
    runs = []
    for kind in ("hip", "oracle"):
        random.seed(42)
        torch.manual_seed(42)
        dev = torch.device("cuda" if kind == "hip" else "cpu")
        eng = MergeEngine(rows, d + 1, "lorentz", dev) if kind == "hip" else OracleEngine(rows, d + 1, "lorentz", fast=True)
        tok = EnhancedFastHyperbolicTokenizer(vocab, torch.nn.Parameter(X.clone()), curvature=1.0, merge_threshold=thr, device=dev,
                                              max_vocab_size=rows, sign_convention="lorentz", use_frequency_aware=True,
                                              use_hierarchical=False, use_adaptive_curvature=True, use_compression_aware=False,
                                              optimize_curvature_freq=3, engine=eng)
        tok.pair_frequencies = freq
        scored, inner = [], tok._score_candidates

        def recording(dd, ii, jj, _inner=inner, _scored=scored):
            out = _inner(dd, ii, jj)
            _scored.extend((c.token_i, c.token_j, float(c.distance).hex(), float(c.semantic_score).hex(), float(c.combined_score).hex()) for c in out)
            return out
        tok._score_candidates = recording
        tok.optimize_merges(steps=steps, log_every=10 ** 9, adaptive_threshold=False)
        runs.append(dict(
            scored=scored,
            merges=[list(m) for m in tok.merge_history],
            rows=tok.embeddings.data[V:tok.current_vocab_size].cpu().numpy().view(np.uint32).copy(),
            curvature=float(torch.as_tensor(tok.get_curvature()).detach()),
            cache=[(c.token_i, c.token_j, float(c.distance), float(getattr(c, 'combined_score', 0.0)), float(getattr(c, 'semantic_score', 0.0)))
                   for c in list(tok.cache.candidates)],
            col0=tok.embeddings.data[:V:997, 0].cpu().numpy().view(np.uint32).copy(),       # the re-projected time column, sampled
            states=(repr(random.getstate()), torch.get_rng_state().numpy().tobytes())))
    hip, ora = runs
    assert len(hip["merges"]) == steps and hip["merges"] == ora["merges"]
    assert np.array_equal(hip["rows"], ora["rows"])
    assert hip["curvature"] == ora["curvature"] and hip["curvature"] != 1.0
    assert np.array_equal(hip["col0"], ora["col0"])
    assert len(hip["cache"]) > 100 and repr(hip["cache"]) == repr(ora["cache"])
    # every candidate that was scored (919 at the refresh + 100 per step): same pair, distance, coherence and combined score
    assert len(hip["scored"]) >= 919 and hip["scored"] == ora["scored"]
    assert hip["states"] == ora["states"]


@pytest.mark.parametrize("n,d,scale,seed,dup,q,f,form", [
    (17842, 128, 0.01, 221827902, False, 1.0, 1.5, "bf16"),      # threshold above every distance: 159 M candidates within 1e-2 of u = 1
    (27883, 100, 0.01, 730370198, True, 0.2, 1.5, "bf16"),       # the same with exact duplicates
    (27102, 5, 0.01, 313578432, False, 0.01, 1.5, "bf16"),       # few dimensions, bf16 prefilter forced
    (843, 1, 0.2, 192548318, False, 0.001, 1.0, "bf16"),         # d = 1: the case that exceeded the one-operand bf16 margin
])
def test_very_dense_tables_from_the_fuzz_runs(oracle, n, d, scale, seed, dup, q, f, form):
    """Round-3 fuzz findings kept as tests (tools/fuzz_loops.py, tools/fuzz_parity.py): tables whose distances are so
    concentrated that the nearest-pair search overflows its emission buffer in both prefilter forms and ends in the exact
    top-1 search -- whose emission cut then has to be bisected (a geometric widening overshoots: P(u - 1 < x) ~ x^(d/2)) --
    and the d = 1 table on which the bf16 margin has to cover the rounding of BOTH operands of a product."""
    from hyptokenizer_amd.engine import MergeEngine
    X = lorentz_table(n, d, seed=seed, scale=scale)
    if dup:
        X[[5, 77, 1234]] = X[[9000, 78, 20000]]
    Xn = X.numpy()
    m = min(n, 300)
    D = oracle.batch_distance(Xn[:m], Xn[:m], 1.0, 1)[np.triu_indices(m, 1)]
    thr = float(np.quantile(D[np.isfinite(D)], q)) * f
    table = torch.zeros((n + 4, d + 1), device="cuda")
    table[:n] = X.cuda()
    eng = MergeEngine(n + 4, d + 1, "lorentz", prefilter=form)
    eng.set_table(table, n)
    od, oi, oj, oc = oracle.pairwise_topk(Xn, n, 1.0, thr, 1, 7, fast=True)
    for rep in range(2):
        a = eng.argmin(1.0, thr)
        assert a is not None and (a[1], a[2]) == (int(oi[0]), int(oj[0])) and bits([a[0]])[0] == bits(od)[0], (rep, a)
        dd, ii, jj, cnt = eng.topk(1.0, thr, 7)
        assert cnt == oc and np.array_equal(ii, oi) and np.array_equal(jj, oj) and np.array_equal(bits(dd), bits(od)), rep


@pytest.mark.parametrize("n,d,scale,c,q,f,k,r0,r1,form", [
    (24218, 100, 0.001, 0.3, 0.2, 1.0, 7, 14162, 21884, "bf16"),     # fuzz: every u within ~1000 ulps of 1; a row range
    (19654, 5, 0.001, 1.0, 0.2, 1.5, 7, 6849, 9601, "bf16"),        # fuzz: ~10^6 pairs per ulp of u
    (17453, 128, 0.001, 4.0, 0.01, 1.0, 7, 0, -1, "f32"),
    (14391, 2, 0.001, 1.0, 0.6, 1.0, 1000, 0, -1, "bf16"),
])
def test_tables_no_prefilter_can_separate_take_the_exact_path(oracle, n, d, scale, c, q, f, k, r0, r1, form):
    """Embeddings of scale 1e-3: all pairwise u lie within a few hundred ulps of 1 (up to ~10^6 pairs per ulp), the prefilter's
    margin alone spans tens of ulps, so NO emission cut fits the buffer.  The reference evaluates every pair, and so does the
    engine's last resort (hm_exact.hip): nearest pair, ordered top-k and the exact count equal the oracle's."""
    from hyptokenizer_amd.engine import MergeEngine
    X = lorentz_table(n, d, seed=n + d, scale=scale)
    Xn = X.numpy()
    m = min(n, 300)
    D = oracle.batch_distance(Xn[:m], Xn[:m], c, 1)[np.triu_indices(m, 1)]
    thr = float(np.quantile(D[np.isfinite(D)], q)) * f
    table = torch.zeros((n + 4, d + 1), device="cuda")
    table[:n] = X.cuda()
    eng = MergeEngine(n + 4, d + 1, "lorentz", prefilter=form)
    eng.set_table(table, n)
    rr1 = n if r1 < 0 else r1
    od, oi, oj, oc = oracle.pairwise_topk(Xn, n, c, thr, 1, k, r0, rr1, fast=True)
    assert oc > 100000
    for rep in range(2):                                       # (the second round goes straight to the exact path: remembered per table)
        dd, ii, jj, cnt = eng.topk(c, thr, k, r0, r1)
        assert cnt == oc and np.array_equal(ii, oi) and np.array_equal(jj, oj) and np.array_equal(bits(dd), bits(od)), rep
        a = eng.argmin(c, thr, r0, r1)
        assert a is not None and (a[1], a[2]) == (int(oi[0]), int(oj[0])) and bits([a[0]])[0] == bits(od)[0], rep


@pytest.mark.parametrize("mode,n,d", [("lorentz", 3000, 40), ("lorentz", 700, 5), ("reference", 6500, 16), ("lorentz", 130, 127)])
def test_exact_path_forced_on_ordinary_tables(oracle, mode, n, d):
    """knob exact_search: the prefilter-free path on tables the two-stage search handles too -- same lists, counts, row ranges;
    literal sign mode at n = 6 500: 21 M pairs tied at distance 0 (more than the emission buffer: the per-row count decides
    how many rows of the tie are needed)."""
    from hyptokenizer_amd import _lib
    from hyptokenizer_amd.engine import MergeEngine
    L = _lib.load()
    X = lorentz_table(n, d, seed=11, scale=0.07)
    X[17] = X[5]
    Xn = X.numpy()
    sm = 1 if mode == "lorentz" else 0
    table = torch.zeros((n + 4, d + 1), device="cuda")
    table[:n] = X.cuda()
    eng = MergeEngine(n + 4, d + 1, mode)
    _lib.check(L.hm_debug_set_knob(eng._h, b"exact_search", 1.0))
    eng.set_table(table, n)
    ref = MergeEngine(n + 4, d + 1, mode)
    ref.set_table(table, n)
    s = ref.pair_distance(np.arange(0, n // 2), np.arange(n // 2, 2 * (n // 2)), 1.0)
    thrs = [0.1] if mode == "reference" else [float(np.percentile(s, 5)), float(np.percentile(s, 60)), 1e9]
    for thr in thrs:
        for k, r0, r1 in [(1, 0, -1), (300, 0, -1), (10000, 0, -1), (50, n // 3, n // 3 + 200), (0, 0, -1)]:
            od, oi, oj, oc = oracle.pairwise_topk(Xn, n, 1.0, thr, sm, max(k, 1), r0, n if r1 < 0 else r1, fast=n > 300)
            gd, gi, gj, gc = eng.topk(1.0, thr, k, r0, r1)
            assert gc == oc, (thr, k, r0, gc, oc)
            if k > 0:
                assert np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(bits(gd), bits(od)), (thr, k, r0)
            rd, ri, rj, rc = ref.topk(1.0, thr, k, r0, r1)
            assert rc == gc and np.array_equal(ri, gi) and np.array_equal(rj, gj) and np.array_equal(bits(rd), bits(gd))
        assert eng.count_candidates(1.0, thr) == ref.count_candidates(1.0, thr)
