"""world_size-2 (and 3) runs of the row-sharded search over gloo on CPU.

Each rank holds a replica, scans only its row range (oracle-backed engine double), and the ranks
exchange records with torch.distributed; the merged result and the whole merge sequence must
equal the single-process run.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hyptokenizer_amd.sharding import pairs_in_rows, partition_rows


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_rows_balances_pairs():
    for n in (10, 1000, 50000, 100000):
        for w in (1, 2, 3, 4, 8):
            b = partition_rows(n, w)
            assert b[0] == 0 and b[-1] == n and len(b) == w + 1 and all(x <= y for x, y in zip(b, b[1:]))
            tot = sum(pairs_in_rows(n, r0, r1) for r0, r1 in zip(b, b[1:]))
            assert tot == n * (n - 1) // 2
            if n >= 50000:
                shares = [pairs_in_rows(n, r0, r1) / tot for r0, r1 in zip(b, b[1:])]
                assert max(shares) - min(shares) < 0.02, shares
                assert all(x % 256 == 0 for x in b[1:-1])


def _worker(rank, world, port, mode, q):
    os.environ["TQDM_DISABLE"] = "1"
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import random
        from helpers import OracleEngine
        from hyptokenizer_amd.sharding import ShardContext, sharded_argmin, sharded_topk
        from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table
        from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
        from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
        torch.set_num_threads(1)
        ctx = ShardContext()
        n, d = 700, 10
        X = lorentz_table(n, d, seed=42, scale=0.05)
        eng = OracleEngine(2000, d + 1, mode)
        eng.set_table(torch.cat([X, torch.zeros(2000 - n, d + 1)]), n)
        thr = 0.12 if mode == "lorentz" else 0.1
        a = sharded_argmin(eng, ctx, 1.0, thr)
        dd, ii, jj, cnt = sharded_topk(eng, ctx, 1.0, thr, 300)
        out = {"argmin": a, "topk": (dd.view(np.uint32).tolist(), ii.tolist(), jj.tolist(), cnt)}
        # whole loops: std 25 steps, fast 130 steps (two refreshes)
        random.seed(42)
        tok = HyperbolicTokenizer(cjk_vocab(n), torch.nn.Parameter(X.clone()), merge_threshold=thr, device=torch.device("cpu"),
                                  max_vocab_size=2000, sign_convention=mode, engine=OracleEngine(2000, d + 1, mode), shard=ctx)
        tok.optimize_merges(steps=25, log_every=10 ** 9)
        out["std_merges"] = list(tok.merge_history)
        out["std_rows"] = tok.embeddings.data[n:tok.current_vocab_size].numpy().view(np.uint32).tolist()
        random.seed(42)
        ftok = FastHyperbolicTokenizer(cjk_vocab(n), torch.nn.Parameter(X.clone()), merge_threshold=thr,
                                       device=torch.device("cpu"), max_vocab_size=2000, sign_convention=mode,
                                       engine=OracleEngine(2000, d + 1, mode), shard=ctx)
        ftok.optimize_merges(steps=130, log_every=1000)
        out["fast_merges"] = list(ftok.merge_history)
        out["fast_thr"] = ftok.merge_threshold
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def _single(mode):
    import random
    from helpers import OracleEngine
    from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table
    from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
    from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
    n, d = 700, 10
    X = lorentz_table(n, d, seed=42, scale=0.05)
    eng = OracleEngine(2000, d + 1, mode)
    eng.set_table(torch.cat([X, torch.zeros(2000 - n, d + 1)]), n)
    thr = 0.12 if mode == "lorentz" else 0.1
    a = eng.argmin(1.0, thr)
    dd, ii, jj, cnt = eng.topk(1.0, thr, 300)
    out = {"argmin": a, "topk": (dd.view(np.uint32).tolist(), ii.tolist(), jj.tolist(), cnt)}
    random.seed(42)
    tok = HyperbolicTokenizer(cjk_vocab(n), torch.nn.Parameter(X.clone()), merge_threshold=thr, device=torch.device("cpu"),
                              max_vocab_size=2000, sign_convention=mode, engine=OracleEngine(2000, d + 1, mode))
    tok.optimize_merges(steps=25, log_every=10 ** 9)
    out["std_merges"] = list(tok.merge_history)
    out["std_rows"] = tok.embeddings.data[n:tok.current_vocab_size].numpy().view(np.uint32).tolist()
    random.seed(42)
    ftok = FastHyperbolicTokenizer(cjk_vocab(n), torch.nn.Parameter(X.clone()), merge_threshold=thr, device=torch.device("cpu"),
                                   max_vocab_size=2000, sign_convention=mode, engine=OracleEngine(2000, d + 1, mode))
    ftok.optimize_merges(steps=130, log_every=1000)
    out["fast_merges"] = list(ftok.merge_history)
    out["fast_thr"] = ftok.merge_threshold
    return out


@pytest.mark.parametrize("world,mode", [(2, "lorentz"), (2, "reference"), (3, "lorentz")])
def test_sharded_search_equals_single_process(world, mode):
    os.environ["TQDM_DISABLE"] = "1"
    ref = _single(mode)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        got = results[r]
        assert got["argmin"] == ref["argmin"]
        assert got["topk"] == ref["topk"]
        assert got["std_merges"] == ref["std_merges"] and got["std_rows"] == ref["std_rows"]
        assert got["fast_merges"] == ref["fast_merges"] and got["fast_thr"] == ref["fast_thr"]


def _cli_worker(rank, world, port, mode, golden_dir, tmp, q):
    os.environ["TQDM_DISABLE"] = "1"
    import pathlib
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)       # (the CLI joins a group that already exists)
    try:
        from helpers import OracleEngine
        from test_host_logic import check_cli
        torch.set_num_threads(1)
        out = pathlib.Path(tmp) / f"rank{rank}"
        out.mkdir()
        check_cli(golden_dir, mode, out, OracleEngine, writer=(rank == 0))
        q.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_cli_under_a_two_rank_launch_reproduces_the_goldens(golden_dir, tmp_path, mode):
    """scripts/train_hyperbolic_tokenizer.py with WORLD_SIZE = 2 (as `python -m torch.distributed.run --nproc-per-node 2`
    sets it): the search is row-sharded over the ranks, both replicas stay identical, rank 0 alone writes the reference's
    output files -- which equal the CLI goldens captured from the reference, in both tokenizer modes."""
    os.environ["TQDM_DISABLE"] = "1"
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cli_worker, args=(r, world, port, mode, golden_dir, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: "ok", 1: "ok"}


def test_cli_num_gpus_must_match_the_launch(monkeypatch):
    from hyptokenizer_amd.scripts import train_hyperbolic_tokenizer as T
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit):
        T._join_process_group(4)
    assert T._join_process_group(1) == (None, None) and T._join_process_group(None) == (None, None)


def _enhanced_run(golden_dir, shard, min_count=None):
    """a G5-style run of config 5 (frequency-aware + adaptive curvature) on the oracle-backed engine: scores of the first
    refresh, merge history, threshold, merged rows"""
    import hyptokenizer_amd.sharding as S
    from test_enhanced_golden import load_g5, make_tok, oracle_engine, seed_all
    if min_count is not None:
        S.COHERENCE_SHARD_MIN = min_count
    z, meta = load_g5(golden_dir, "lorentz")
    seed_all(123)
    tok = make_tok(z, meta, "lorentz", "freq_only", oracle_engine, use_adaptive_curvature=True, optimize_curvature_freq=7,
                   cache_size=40, shard=shard)       # (a cache smaller than the candidate count: the refresh lists ALL candidates)
    n0 = tok.current_vocab_size
    scored = tok._find_merge_candidates_fast()
    first = [(c.token_i, c.token_j, float(c.combined_score), float(c.semantic_score)) for c in scored[:200]]
    seed_all(123)
    tok2 = make_tok(z, meta, "lorentz", "freq_only", oracle_engine, use_adaptive_curvature=True, optimize_curvature_freq=7,
                    cache_size=40, shard=shard)
    tok2.optimize_merges(steps=24, log_every=10 ** 9, adaptive_threshold=False)
    return {"first": first, "n_scored": len(scored), "merges": list(tok2.merge_history), "thr": tok2.merge_threshold,
            "curv": repr(float(torch.as_tensor(tok2.get_curvature()).detach())),        # (repr: a NaN equals a NaN)
            "rows": tok2.embeddings.data[n0:tok2.current_vocab_size].numpy().view(np.uint32).tolist()}


def _enhanced_worker(rank, world, port, golden_dir, q):
    os.environ["TQDM_DISABLE"] = "1"
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hyptokenizer_amd.sharding import ShardContext
        torch.set_num_threads(1)
        q.put((rank, _enhanced_run(golden_dir, ShardContext(), min_count=8)))    # (8: the coherence batches ARE partitioned)
    finally:
        dist.destroy_process_group()


def test_enhanced_tokenizer_row_sharded_equals_single_process(golden_dir):
    """BASELINE config 5 on more than one rank: EnhancedFastHyperbolicTokenizer(shard=ctx) -- sharded candidate search and
    listing, coherence batches partitioned by candidate over the ranks and all-gathered, the same seeded generator on every
    rank -- reproduces the single-process run: candidate order and scores of a refresh, merges, curvature, merged rows."""
    os.environ["TQDM_DISABLE"] = "1"
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    ref = _enhanced_run(golden_dir, None)
    assert ref["n_scored"] > 40 and len(ref["merges"]) >= 10
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_enhanced_worker, args=(r, world, port, golden_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert results[r] == ref, r
