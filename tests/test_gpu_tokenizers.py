"""GPU: the tokenizer classes, the CLI and the Lorentz function surface on the real engine against
the reference's golden vectors, plus size-independent properties at the benchmark sizes."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import bits, nan_equal_close  # noqa: E402
from test_host_logic import check_cli, check_sequences  # noqa: E402
from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table  # noqa: E402

MODES = {"reference": 0, "lorentz": 1}


@pytest.mark.parametrize("mode", ["reference", "lorentz"])
def test_merge_sequences_match_reference_on_gpu(golden_dir, mode):
    check_sequences(golden_dir, mode, None, "cuda")


@pytest.mark.parametrize("mode", ["reference", "lorentz"])
def test_cli_matches_reference_on_gpu(golden_dir, mode, tmp_path):
    check_cli(golden_dir, mode, tmp_path, None, init_device="cpu")


@pytest.mark.parametrize("mode", ["reference", "lorentz"])
def test_lorentz_functions_match_goldens_and_oracle(oracle, golden_dir, mode):
    from hyptokenizer_amd.embedding import lorentz_model as LM
    z = np.load(os.path.join(golden_dir, f"g1_primitives_{mode}.npz"))
    sm = MODES[mode]
    for d in (10, 50, 100):
        for scale in (0.01, 0.05, 0.5):
            tag = f"d{d}_s{scale}"
            Xn = z[f"{tag}_X"]
            X = torch.from_numpy(Xn).cuda()
            bd = LM.batch_distance(X, X, 1.0, sign_convention=mode).cpu().numpy()
            assert np.array_equal(bits(bd), bits(oracle.batch_distance(Xn, Xn, 1.0, sm)))          # oracle: exact
            off = ~np.eye(64, dtype=bool)
            assert np.allclose(bd[off], z[f"{tag}_bd"][off], atol=1e-5)                              # reference: 1e-5
            assert np.array_equal(bits(LM.batch_distance_optimized(X, X, 2.0, sign_convention=mode).cpu().numpy()),
                                  bits(oracle.batch_distance(Xn, Xn, 2.0, sm)))
            a, b = X[0:63], X[1:64]
            dist = LM.distance(a, b, 1.0, sign_convention=mode).cpu().numpy()
            assert np.array_equal(bits(dist), bits(oracle.distance(Xn[0:63], Xn[1:64], 1.0, sm)))
            assert np.allclose(dist, z[f"{tag}_dist"], atol=1e-5)
            md = LM.minkowski_dot(a, b, sign_convention=mode).cpu().numpy()
            assert np.array_equal(bits(md), bits(z[f"{tag}_mdot"]))                                  # bit-exact vs torch
            lg = LM.log_map(a, b, 1.0, sign_convention=mode)
            assert np.array_equal(bits(lg.cpu().numpy()), bits(oracle.log_map(Xn[0:63], Xn[1:64], sm)))
            assert nan_equal_close(lg.cpu().numpy(), z[f"{tag}_log"], 1e-5)
            for w in (0.5, 1.0 / 3.0, 0.75):
                v = lg * w
                ex = LM.exp_map(a, v, 1.0)
                ref = z[f"{tag}_exp_w{w:.4f}"]
                assert nan_equal_close(ex.cpu().numpy(), ref, 1e-5 * max(1.0, float(np.nanmax(np.abs(ref))) if not np.isnan(ref).all() else 1.0))
                mid = LM.project_to_hyperboloid(ex, 1.0).cpu().numpy()
                refm = z[f"{tag}_mid_w{w:.4f}"]
                assert nan_equal_close(mid, refm, 1e-5 * max(1.0, float(np.nanmax(np.abs(refm))) if not np.isnan(refm).all() else 1.0))
            P = torch.from_numpy(z[f"{tag}_P"]).cuda()
            assert nan_equal_close(LM.project_to_hyperboloid(P, 2.0).cpu().numpy(), z[f"{tag}_proj_c2"], 4e-5)
    # broadcasting form used by _compute_pairwise_distances / the CLI callback
    X = torch.from_numpy(z["d10_s0.05_X"]).cuda()
    dm = LM.distance(X.unsqueeze(1), X.unsqueeze(0), 1.0, sign_convention=mode)
    assert dm.shape == (64, 64)
    assert torch.equal(dm, LM.batch_distance(X, X, 1.0, sign_convention=mode))
    # helpers outside the hot path keep working on top of minkowski_dot
    assert LM.minkowski_norm(X, sign_convention="reference").shape == (64,)
    assert LM.lorentz_to_klein(X).shape == (64, 10)
    assert LM.parallel_transport(X * 0, X, X.roll(1, 0)).shape == X.shape
    assert LM.riemannian_gradient(X, X).shape == X.shape


def test_tokenizer_public_methods_on_gpu(oracle):
    from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
    from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
    X = lorentz_table(300, 10, seed=42, scale=0.05)
    tok = HyperbolicTokenizer(cjk_vocab(300), torch.nn.Parameter(X), merge_threshold=0.15, sign_convention="lorentz")
    assert tok.embeddings.shape == (100000, 11) and tok.embeddings.is_cuda
    cands = tok._find_merge_candidates()
    oi, oj, od, total = oracle.pairwise_candidates(X.numpy(), 300, 1.0, float(np.float32(0.15)), 1)
    assert len(cands) == total and [c[0] for c in cands] == oi.tolist() and [c[1] for c in cands] == oj.tolist()
    assert np.array_equal(bits([c[2] for c in cands]), bits(od))
    D = tok._compute_pairwise_distances()
    assert D.shape == (300, 300) and torch.allclose(D, D.t(), atol=1e-5) and float(D.diagonal().abs().max()) < 2e-3
    assert tok._evaluate_candidates_parallel(cands[:100]) == cands[0]
    # in-place edit of the Parameter is picked up; edits through .data need refresh_engine()
    with torch.no_grad():
        tok.embeddings[5] = tok.embeddings[6]
    best = tok._best_candidate()
    assert best[:2] == (5, 6) and best[2] == 0.0
    # small vocabulary (n <= 100 branch of the reference: double compare)
    small = HyperbolicTokenizer(cjk_vocab(50), torch.nn.Parameter(X[:50]), merge_threshold=0.2, sign_convention="lorentz")
    c2 = small._find_merge_candidates()
    oi, oj, od, total = oracle.pairwise_candidates(X[:50].numpy(), 50, 1.0, 0.2, 1)
    assert len(c2) == total
    ftok = FastHyperbolicTokenizer(cjk_vocab(300), torch.nn.Parameter(X), merge_threshold=0.15, sign_convention="lorentz",
                                   cache_size=50)
    found = ftok._find_merge_candidates_fast()
    assert len(found) == len(cands) and found.stored == 50 and len(ftok.cache.candidates) == 50
    again = ftok._find_merge_candidates_fast()
    assert len(again) == 50 and len(ftok.cache.candidates) == 0     # pop(100) drains a 50-entry cache
    tup = ftok._find_merge_candidates()
    assert len(tup) == 50 and isinstance(tup[0], tuple)


@pytest.mark.parametrize("V,d,prefilter", [(50000, 100, "bf16"), (50000, 50, "bf16"), (100000, 100, "bf16"),
                                           (50000, 50, "f32"), (50000, 100, "f32")])
def test_full_size_properties(V, d, prefilter):
    """BASELINE sizes, properties that need no oracle run:
    (1) the nearest pair's distance equals the gathered-distance kernel on that pair, bit for bit;
    (2) the top-k list is sorted in (d, i, j) order, has i < j, and starts with the argmin;
    (3) row-sharded searches partition the count and their merged lists reproduce the global list;
    (4) appending a duplicate of a row makes (row, new) the nearest pair at distance 0;
    (5) every listed distance is reproduced by the gathered kernel."""
    from hyptokenizer_amd.engine import MergeEngine
    X = lorentz_table(V, d, seed=42, scale=0.05)
    table = torch.zeros((V + 8, d + 1), device="cuda")
    table[:V] = X.cuda()
    eng = MergeEngine(V + 8, d + 1, "lorentz", prefilter=prefilter)     # BASELINE config 2 names the fp32 form, config 3 the bf16 one
    eng.set_table(table, V)
    s = eng.pair_distance(np.arange(0, 4000), np.arange(4000, 8000), 1.0)
    thr = float(np.percentile(s, 0.05))
    a = eng.argmin(1.0, thr)
    dd, ii, jj, cnt = eng.topk(1.0, thr, 10000)
    assert a is not None and cnt >= len(dd) > 0
    assert (a[1], a[2]) == (int(ii[0]), int(jj[0])) and bits([a[0]])[0] == bits(dd)[0]
    assert np.all(ii < jj) and np.all(jj < V) and np.all(dd < np.float32(thr))
    key = list(zip(bits(dd).tolist(), ii.tolist(), jj.tolist()))
    assert key == sorted(key) and len(set(zip(ii.tolist(), jj.tolist()))) == len(ii)
    assert np.array_equal(bits(eng.pair_distance(ii, jj, 1.0)), bits(dd))
    parts = [0, V // 5, V // 2, V - 300, V]
    tot, merged = 0, []
    for r0, r1 in zip(parts[:-1], parts[1:]):
        pd, pi, pj, pc = eng.topk(1.0, thr, 10000, r0, r1)
        tot += pc
        merged += list(zip(bits(pd).tolist(), pi.tolist(), pj.tolist()))
        assert np.all((pi >= r0) & (pi < r1))
    assert tot == cnt
    assert sorted(merged)[: len(key)] == key
    if V >= 100000:
        # ranges large enough for the 512-row-block form of the scan, with ragged ends inside a block
        tot2, merged2 = 0, []
        for r0, r1 in ((0, 777), (777, 90001), (90001, V)):
            pd, pi, pj, pc = eng.topk(1.0, thr, 10000, r0, r1)
            tot2 += pc
            merged2 += list(zip(bits(pd).tolist(), pi.tolist(), pj.tolist()))
            assert np.all((pi >= r0) & (pi < r1))
            a2 = eng.argmin(1.0, thr, r0, r1)
            assert (a2 is None) == (pc == 0)
            if a2:
                assert (int(bits([a2[0]])[0]), a2[1], a2[2]) == (int(bits(pd)[0]), int(pi[0]), int(pj[0]))
        assert tot2 == cnt and sorted(merged2)[: len(key)] == key
    # second refresh uses the predicted cut: same answer
    dd2, ii2, jj2, cnt2 = eng.topk(1.0, thr, 10000)
    assert cnt2 == cnt and np.array_equal(ii2, ii) and np.array_equal(jj2, jj) and np.array_equal(bits(dd2), bits(dd))
    row = 12345
    table[V] = table[row]
    eng.update_rows(table, V, V + 1)
    b = eng.argmin(1.0, thr)
    assert (b[1], b[2]) == (row, V) and b[0] == 0.0
    # midpoint of identical rows: NaN tangent -> NaN row, which never forms a candidate
    eng.merge_append(row, V, 0.5, 1.0, table, V + 1)
    assert torch.isnan(table[V + 1]).all()
    c2 = eng.argmin(1.0, thr)
    assert (c2[1], c2[2]) == (row, V)


@pytest.mark.parametrize("mode", ["reference", "lorentz"])
def test_row_argmin_matches_oracle(oracle, mode):
    """K3 + reduction against the oracle's row-vs-all distances: nearest partner, threshold, ties"""
    from hyptokenizer_amd.engine import MergeEngine
    V, d = 3000, 40
    X = lorentz_table(V, d, seed=5, scale=0.1)
    X[77] = X[2100]                                      # an exact duplicate: distance 0 tie-breaks by index
    Xn = X.numpy()
    eng = MergeEngine(V, d + 1, mode)
    eng.set_table(X.cuda(), V)
    for row, npart in [(2999, 2999), (2100, 2100), (77, 3000), (0, 3000), (1500, 10), (5, 0)]:
        dist = np.asarray(oracle.row_vs_all(Xn, max(npart, row + 1), row, 1.0, MODES[mode]), np.float32)[:npart]
        for thr in (0.3, 1.5, 1e9):
            want = None
            for i in np.nonzero(dist < np.float32(thr))[0].tolist():
                if i != row:
                    key = (int(bits([dist[i]])[0]), min(i, row), max(i, row))
                    want = key if want is None or key < want else want
            got = eng.row_argmin(row, npart, 1.0, thr)
            assert (got is None) == (want is None), (row, npart, thr)
            if got is not None:
                assert (int(bits([got[0]])[0]), got[1], got[2]) == want, (row, npart, thr)


@pytest.mark.parametrize("V,d,thr", [(6000, 64, 0.6), (150, 16, 1.0), (95, 16, 1.0)])
def test_incremental_loop_equals_full_search(V, d, thr):
    """incremental=True merges the same pairs into the same rows as a full search every step
    (95 rows: the loop crosses the reference's n <= 100 double-compare branch)"""
    from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
    X = lorentz_table(V, d, seed=11, scale=0.05)
    runs = []
    for inc in (False, True):
        tok = HyperbolicTokenizer(vocab=cjk_vocab(V), embeddings=torch.nn.Parameter(X.clone()), merge_threshold=thr,
                                  device=torch.device("cuda"), max_vocab_size=V + 400, sign_convention="lorentz", incremental=inc)
        tok.optimize_merges(steps=120, log_every=10 ** 9)
        tok.embeddings.data[3] = tok.embeddings.data[V - 1]       # an edit of the table invalidates the running minimum
        tok.refresh_engine()
        tok.optimize_merges(steps=40, log_every=10 ** 9)
        runs.append(tok)
    a, b = runs
    assert a.merge_history == b.merge_history and len(a.merge_history) == 160
    n = a.current_vocab_size
    assert torch.equal(a.embeddings.data[:n].view(torch.int32), b.embeddings.data[:n].view(torch.int32))


def test_device_record_and_world1_nccl_shard(oracle):
    """asynchronous argmin record (multi-GPU exchange path) equals the host form; a 1-rank RCCL
    process group drives the sharded tokenizer through the same code the N-GPU run uses"""
    import socket
    import torch.distributed as dist
    from hyptokenizer_amd.engine import MergeEngine
    from hyptokenizer_amd.sharding import ShardContext
    from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
    from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
    n, d = 5000, 30
    X = lorentz_table(n, d, seed=3, scale=0.05)
    table = torch.zeros((n + 64, d + 1), device="cuda")
    table[:n] = X.cuda()
    eng = MergeEngine(n + 64, d + 1, "lorentz")
    eng.set_table(table, n)
    rec = torch.empty(4, dtype=torch.int32, device="cuda")
    for thr, r0, r1 in ((0.3, 0, -1), (0.3, 1000, 2500), (1e-9, 0, -1)):
        host = eng.argmin(1.0, thr, r0, r1)
        eng.argmin_into(1.0, thr, r0, r1, rec)
        got = rec.cpu().numpy()
        if host is None:
            assert got[0] == 0
        else:
            assert got[0] == 1 and (int(got[2]), int(got[3])) == (host[1], host[2])
            assert np.uint32(got[1]) == np.float32(host[0]).view(np.uint32)
    assert eng.scan_totals()["launches"] >= 6
    # armed device-record searches back to back, interleaved with host searches, in the tie-flood mode (every pair
    # at distance 0: the first search may overflow its emission buffer and report found = 2)
    n2 = 6000
    X2 = lorentz_table(n2, d, seed=4, scale=0.05)
    eng2 = MergeEngine(n2 + 64, d + 1, "reference")
    t2 = torch.zeros((n2 + 64, d + 1), device="cuda")
    t2[:n2] = X2.cuda()
    eng2.set_table(t2, n2)
    for rnd in range(3):
        for _ in range(2):
            eng2.argmin_into(1.0, 0.1, 0, -1, rec)
            got = rec.cpu().numpy()
            assert got[0] in (1, 2)
            if got[0] == 1:
                assert (int(got[1]), int(got[2]), int(got[3])) == (0, 0, 1)
        assert eng2.argmin(1.0, 0.1) == (0.0, 0, 1)
        eng2.merge_append(0, 1, 0.5, 1.0, t2, n2 + rnd)          # appends a NaN row: never a candidate
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        ctx = ShardContext(device=torch.device("cuda", 0))
        runs = []
        for shard in (None, ctx):
            tok = HyperbolicTokenizer(cjk_vocab(n), torch.nn.Parameter(X), merge_threshold=0.3, max_vocab_size=n + 64,
                                      sign_convention="lorentz", shard=shard)
            tok.optimize_merges(steps=40, log_every=10 ** 9)
            ftok = FastHyperbolicTokenizer(cjk_vocab(n), torch.nn.Parameter(X), merge_threshold=0.3, max_vocab_size=n + 256,
                                           sign_convention="lorentz", shard=shard)
            ftok.optimize_merges(steps=120, log_every=10 ** 9, adaptive_threshold=False)
            runs.append((list(tok.merge_history), tok.embeddings.data[n:n + 40].cpu(), list(ftok.merge_history)))
        assert runs[0][0] == runs[1][0] and runs[0][2] == runs[1][2]
        assert torch.equal(runs[0][1].view(torch.int32), runs[1][1].view(torch.int32))      # bits: NaN rows included
        # the sharded runs went through the library's own exchange step (hm_comm_init + hm_shard_merge_steps / hm_global_topk)
        assert tok._engine.comm_info() == (0, 1) and ftok._engine.comm_info() == (0, 1)
        # ... whose one-shot forms equal the single-process searches, and whose loop equals the single-process loop
        eng.comm_init()
        assert eng.comm_info() == (0, 1)
        for thr in (0.3, 0.25, 1e-9):
            assert eng.global_argmin(1.0, thr) == eng.argmin(1.0, thr)
            gd, gi, gj, gc = eng.global_topk(1.0, thr, 300)
            hd, hi, hj, hc = eng.topk(1.0, thr, 300)
            assert gc == hc and np.array_equal(gi, hi) and np.array_equal(gj, hj) and np.array_equal(bits(gd), bits(hd))
        od, oi, oj, oc = oracle.pairwise_topk(X.numpy(), n, 1.0, 0.3, 1, 300)
        gd, gi, gj, gc = eng.global_topk(1.0, 0.3, 300)
        assert gc == oc and np.array_equal(gi, oi) and np.array_equal(gj, oj) and np.array_equal(bits(gd), bits(od))
        t_a, t_b = table.clone(), table.clone()
        eng_b = MergeEngine(n + 64, d + 1, "lorentz")
        eng_b.set_table(t_b, n)
        eng.set_table(t_a, n)
        lens = np.ones(n, np.int32)
        eng.set_token_lengths(lens)
        eng_b.set_token_lengths(lens)
        eng.debug_time_loops(True)
        ra, da = eng.shard_merge_steps(1.0, 0.3, t_a, 24)
        rb, db = eng_b.std_merge_steps(1.0, 0.3, t_b, 24)
        assert da == db == 24 and [r[2:] for r in ra] == [r[2:] for r in rb] and [bits([r[1]])[0] for r in ra] == [bits([r[1]])[0] for r in rb]
        assert torch.equal(t_a.view(torch.int32), t_b.view(torch.int32))
        tm = eng.last_loop_timing()
        assert tm["steps"] == 24 and 0 < tm["scan_ms"] < tm["batch_ms"]         # every scan of the sharded batch carried its events
        # a loop that runs out of candidates stops on the device in the sharded form too
        t_c = table.clone()
        eng.set_table(t_c, n)
        dmin = eng.argmin(1.0, 10.0)[0]
        ra, da = eng.shard_merge_steps(1.0, dmin, t_c, 5)                       # nothing is strictly below the nearest distance
        assert da == 0 and ra[0][0] == 0 and all(r[0] == 3 for r in ra[1:])
        ra, da = eng.shard_merge_steps(1.0, 0.0, t_c, 3)                        # ... nor below a non-positive threshold
        assert da == 0 and ra[0][0] == 0 and all(r[0] == 3 for r in ra[1:])
        eng.comm_destroy()
        assert eng.comm_info() is None
    finally:
        dist.destroy_process_group()


def test_very_dense_table_argmin_falls_back_to_exact_top1(oracle):
    """131k points in a 2-D hyperboloid: tens of millions of pairs sit inside the running key's slack band, the
    bounded second pass of the argmin search overflows too, and the search finishes through the top-1 path"""
    from hyptokenizer_amd.engine import MergeEngine
    V, d = 131064, 2
    X = lorentz_table(V, d, seed=1, scale=0.05)
    table = torch.zeros((V + 8, d + 1), device="cuda")
    table[:V] = X.cuda()
    eng = MergeEngine(V + 8, d + 1, "lorentz")
    eng.set_table(table, V)
    s = eng.pair_distance(np.arange(0, 3000), np.arange(3000, 6000), 1.0)
    thr = float(np.percentile(s, 0.1))
    od, oi, oj, oc = oracle.pairwise_topk(X.numpy(), V, 1.0, thr, 1, 200, fast=True)
    a = eng.argmin(1.0, thr)
    assert a is not None and (a[1], a[2]) == (int(oi[0]), int(oj[0])) and bits([a[0]])[0] == bits(od)[0]
    assert eng.argmin(1.0, thr) == a
    dd, ii, jj, cnt = eng.topk(1.0, thr, 200)
    assert cnt == oc and np.array_equal(ii, oi) and np.array_equal(jj, oj) and np.array_equal(bits(dd), bits(od))


def test_literal_sign_mode_at_benchmark_size():
    """The classes' DEFAULT mode (arithmetic as shipped) at V = 50 000, d = 100: every pair is a
    candidate at distance 0.0 and the order is row-major -- nearest pair (0, 1), the 10 000 best are
    (0, 1) .. (0, 10000), the count is N(N-1)/2, and the fast loop merges (0,1),(0,1),(0,101),(0,201),(0,301)
    (hyperbolic_merge.py:247-269,378; fast_hyperbolic_merge.py:91-95)"""
    from hyptokenizer_amd.engine import MergeEngine
    from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
    from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
    V, d = 50000, 100
    X = lorentz_table(V, d, seed=42, scale=0.05)
    table = torch.zeros((V + 64, d + 1), device="cuda")
    table[:V] = X.cuda()
    eng = MergeEngine(V + 64, d + 1, "reference")
    eng.set_table(table, V)
    assert eng.argmin(1.0, 0.1) == (0.0, 0, 1)
    assert eng.argmin(1.0, 0.1) == (0.0, 0, 1)             # armed / seeded second search
    dd, ii, jj, cnt = eng.topk(1.0, 0.1, 10000)
    assert cnt == V * (V - 1) // 2
    assert np.all(dd == 0.0) and np.all(ii == 0) and np.array_equal(jj, np.arange(1, 10001, dtype=np.int32))
    assert eng.count_candidates(1.0, 0.1) == cnt
    ftok = FastHyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X), merge_threshold=0.1, max_vocab_size=V + 64)
    pairs = []
    orig = ftok._append_token
    ftok._append_token = lambda i, j: (pairs.append((i, j)), orig(i, j))[1]
    ftok.optimize_merges(steps=5, log_every=10 ** 9)
    assert pairs == [(0, 1), (0, 1), (0, 101), (0, 201), (0, 301)]
    assert ftok.merge_threshold == 1e-5                    # "Maximum distance is near zero" rewrite (fast...:493-499)
    assert torch.isnan(ftok.embeddings.data[V:V + 5]).all()
    tok = HyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X), merge_threshold=0.1, max_vocab_size=V + 64)
    tok.optimize_merges(steps=3, log_every=10 ** 9)
    assert [m[:2] for m in tok.merge_history] == [(cjk_vocab(2)[0], cjk_vocab(2)[1])] * 3


def test_fast_loop_batched_merges_equal_step_by_step():
    """the merges between two refreshes go to the engine as one launch (FastHyperbolicTokenizer._plan_merges);
    same history, same rows, same cache state as with one launch per merge"""
    from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
    V, d = 6000, 48
    X = lorentz_table(V, d, seed=17, scale=0.05)
    runs = []
    for batched in (True, False):
        import random
        random.seed(5)
        tok = FastHyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X.clone()), merge_threshold=0.5, max_vocab_size=V + 400,
                                      sign_convention="lorentz", cache_size=2500)
        tok.batch_merges = batched
        tok.lazy_count = batched
        tok.optimize_merges(steps=333, log_every=50)
        runs.append(tok)
    a, b = runs
    assert a.merge_history == b.merge_history and len(a.merge_history) == 333
    n = a.current_vocab_size
    assert torch.equal(a.embeddings.data[:n + 8].view(torch.int32), b.embeddings.data[:n + 8].view(torch.int32))
    assert a.merge_threshold == b.merge_threshold and len(a.cache) == len(b.cache)
    assert a.last_run_stats["num_candidates"] == b.last_run_stats["num_candidates"]


def test_fast_loop_prefetched_refreshes_equal_synchronous_ones():
    """the next refresh is enqueued behind the planned merges (topk_refresh_begin) and collected when the loop gets there:
    same history, rows, thresholds and candidate counts as with synchronous refreshes -- and the prefetched lists are
    really the ones consumed; log lines, statistics and a threshold rescale in between must not disturb it"""
    from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
    V, d = 6000, 48
    X = lorentz_table(V, d, seed=17, scale=0.05)
    runs, used = [], []
    for prefetch in (True, False):
        import random
        random.seed(5)
        tok = FastHyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X.clone()), merge_threshold=0.5, max_vocab_size=V + 1400,
                                      sign_convention="lorentz", cache_size=2500)
        tok.prefetch_refresh = prefetch
        eng = tok._get_engine()
        ends = [0, 0]
        inner = eng.topk_refresh_end

        def counted(_inner=inner, _ends=ends):
            out = _inner()
            _ends[0] += 1
            _ends[1] += out is not None
            return out
        eng.topk_refresh_end = counted
        tok.optimize_merges(steps=1230, log_every=400)          # crosses step 1000: the adaptive threshold is rescaled there
        runs.append(tok)
        used.append(ends)
    a, b = runs
    assert used[1] == [0, 0] and used[0][0] >= 10 and used[0][1] >= 8        # refreshes in flight were collected and delivered lists
    assert a.merge_history == b.merge_history and len(a.merge_history) == 1230
    n = a.current_vocab_size
    assert torch.equal(a.embeddings.data[:n + 8].view(torch.int32), b.embeddings.data[:n + 8].view(torch.int32))
    assert a.merge_threshold == b.merge_threshold and len(a.cache) == len(b.cache)
    assert a.last_run_stats["num_candidates"] == b.last_run_stats["num_candidates"]
    # the engine refuses other work while a refresh is in flight
    eng = a._get_engine()
    if eng.topk_refresh_begin(1.0, a._search_threshold(), a.cache.max_size):
        with pytest.raises(RuntimeError):
            eng.argmin(1.0, 0.5)
        eng.topk_refresh_end()
