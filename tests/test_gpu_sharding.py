"""GPU: two ranks sharing the one MI355X of the test box, real engines, records exchanged over gloo.

The N-GPU run uses the same code with RCCL (``backend="nccl"``) and one GPU per rank; what is checked
here is everything except the transport: row ranges of the real scan kernels, per-rank running-key
seeds, replicated deterministic merges -- the sharded loops must reproduce the single-process merge
sequence and rows bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N, D, THR = 6000, 40, 0.55


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(shard):
    import random
    from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table
    from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
    from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
    dev = torch.device("cuda", 0)
    X = lorentz_table(N, D, seed=42, scale=0.05)
    out = {}
    tok = HyperbolicTokenizer(cjk_vocab(N), torch.nn.Parameter(X.clone()), merge_threshold=THR, device=dev, max_vocab_size=N + 512,
                              sign_convention="lorentz", shard=shard)
    eng = tok._get_engine()
    calls = [0]
    inner = eng.shard_merge_step

    def counted(*a, **k):
        calls[0] += 1
        return inner(*a, **k)
    eng.shard_merge_step = counted
    tok.optimize_merges(steps=40, log_every=10 ** 9)
    out["shard_merge_steps"] = calls[0]          # the sharded run goes through the device-resident batch
    out["std_merges"] = list(tok.merge_history)
    out["std_rows"] = tok.embeddings.data[N:tok.current_vocab_size].cpu().numpy().view(np.uint32).tolist()
    # a loop that finds no candidate: the device-side stop at the first step and the skipped steps behind it
    etok = HyperbolicTokenizer(cjk_vocab(N), torch.nn.Parameter(X.clone()), merge_threshold=THR, device=dev, max_vocab_size=N + 512,
                               sign_convention="lorentz", shard=shard)
    near = etok._get_engine().argmin(1.0, THR)                 # nearest pair of the untouched table (whole table: same on every rank)
    etok.merge_threshold = float(np.float32(near[0]) * np.float32(0.98))     # nothing below it: the first step ends the loop
    etok.optimize_merges(steps=60, log_every=10 ** 9)
    out["exhaust_merges"] = list(etok.merge_history)
    random.seed(42)
    ftok = FastHyperbolicTokenizer(cjk_vocab(N), torch.nn.Parameter(X.clone()), merge_threshold=THR, device=dev,
                                   max_vocab_size=N + 512, sign_convention="lorentz", shard=shard, cache_size=500)
    ftok.optimize_merges(steps=230, log_every=1000)
    out["fast_merges"] = list(ftok.merge_history)
    out["fast_thr"] = ftok.merge_threshold
    itok = HyperbolicTokenizer(cjk_vocab(N), torch.nn.Parameter(X.clone()), merge_threshold=THR, device=dev, max_vocab_size=N + 512,
                               sign_convention="lorentz", shard=shard, incremental=True)
    itok.optimize_merges(steps=40, log_every=10 ** 9)
    out["incr_merges"] = list(itok.merge_history)
    return out


def _worker(rank, world, port, q):
    os.environ["TQDM_DISABLE"] = "1"
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hyptokenizer_amd.sharding import ShardContext
        q.put((rank, _run(ShardContext())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_reproduce_the_single_process_run(world):
    os.environ["TQDM_DISABLE"] = "1"
    ref = _run(None)
    assert len(ref["std_merges"]) == 40 and len(ref["fast_merges"]) == 230 and ref["incr_merges"] == ref["std_merges"]
    assert ref["exhaust_merges"] == []
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ref.pop("shard_merge_steps") == 0
    for r in range(world):
        assert results[r].pop("shard_merge_steps") == 40, r      # every step through hm_shard_merge_step
        assert results[r] == ref, r


def _nccl_worker(rank, world, port, q):
    os.environ["TQDM_DISABLE"] = "1"
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from hyptokenizer_amd.sharding import ShardContext
        from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table
        from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
        from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
        dev = torch.device("cuda", rank)
        ctx = ShardContext(device=dev)
        X = lorentz_table(N, D, seed=42, scale=0.05)
        tok = HyperbolicTokenizer(cjk_vocab(N), torch.nn.Parameter(X.clone()), merge_threshold=THR, device=dev, max_vocab_size=N + 512,
                                  sign_convention="lorentz", shard=ctx)
        tok.optimize_merges(steps=40, log_every=10 ** 9)
        import random
        random.seed(42)
        ftok = FastHyperbolicTokenizer(cjk_vocab(N), torch.nn.Parameter(X.clone()), merge_threshold=THR, device=dev,
                                       max_vocab_size=N + 512, sign_convention="lorentz", shard=ctx, cache_size=500)
        ftok.optimize_merges(steps=230, log_every=1000)
        q.put((rank, {"std_merges": list(tok.merge_history), "fast_merges": list(ftok.merge_history),
                      "std_rows": tok.embeddings.data[N:tok.current_vocab_size].cpu().numpy().view(np.uint32).tolist()}))
    finally:
        dist.destroy_process_group()


def test_two_ranks_over_rccl_when_two_gpus_are_visible():
    """N > 1 over RCCL (backend "nccl", one GPU per rank): runs wherever at least two devices are visible (the
    driver's 8-GPU node), skipped on the one-GPU test box.  Same comparisons as the shared-GPU gloo test."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs (RCCL); the one-GPU box covers the same logic over gloo")
    os.environ["TQDM_DISABLE"] = "1"
    ref = _run(None)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        assert results[r]["std_merges"] == ref["std_merges"] and results[r]["fast_merges"] == ref["fast_merges"], r
        assert results[r]["std_rows"] == ref["std_rows"], r


@pytest.mark.gpu
def test_device_merge_of_gathered_topk_lists_matches_lexsort():
    """The N-rank refresh merges the ranks' ordered lists on the device (sharding.merge_topk_lists_device); with one
    GPU here the gathered buffer is built by hand: ragged lists, ties in the distance, a rank with nothing."""
    from hyptokenizer_amd.sharding import merge_topk_lists_device
    rng = np.random.default_rng(3)
    world, k = 5, 700
    buf = np.zeros((world, k + 1, 3), np.int32)
    lists, total = [], 0
    for r in range(world):
        m = [700, 0, 123, 700, 699][r]
        d = np.sort(rng.choice(np.linspace(0.0, 0.4, 300, dtype=np.float32), size=m))       # many equal distances
        i = rng.integers(0, 5000, m).astype(np.int32)
        j = (i + rng.integers(1, 5000, m)).astype(np.int32)
        cnt = m + int(rng.integers(0, 2 ** 33))
        buf[r, 0] = (m, cnt & 0x7FFFFFFF, cnt >> 31)
        buf[r, 1:m + 1, 0] = d.view(np.int32)
        buf[r, 1:m + 1, 1], buf[r, 1:m + 1, 2] = i, j
        lists.append((d, i, j))
        total += cnt
    du = np.concatenate([x[0].view(np.uint32) for x in lists])
    ii = np.concatenate([x[1] for x in lists])
    jj = np.concatenate([x[2] for x in lists])
    order = np.lexsort((jj, ii, du))[:k]
    d, i, j, tot = merge_topk_lists_device(torch.from_numpy(buf).cuda(), k)
    assert tot == total
    assert np.array_equal(d.view(np.uint32), du[order]) and np.array_equal(i, ii[order]) and np.array_equal(j, jj[order])


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_on_one_gpu():
    """``bench.py --gpus 2`` as the driver launches it (torch.distributed.run, one process per rank), rehearsed on ONE GPU:
    HM_BENCH_REHEARSE=1 puts both ranks on cuda:0 and the exchange on gloo.  Checks that the N > 1 path of the bench runs end
    to end -- rendezvous, sharded device loop, sharded fast and incremental loops, the bench's own collectives, one JSON line
    from rank 0 -- and that the sharded incremental loop merges what the sharded full search merges."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HM_BENCH_REHEARSE="1", TQDM_DISABLE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--no-legs", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["incremental"]["same_merges_as_full_search"] is True and d["fast_path"]["merges_per_s"] > 0
