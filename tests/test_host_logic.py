"""Host logic of the tokenizer classes against the reference's golden merge sequences.

The engine is replaced by the oracle-backed test double (tests/helpers.py) so that these run in the
GPU-less container; tests/test_gpu_tokenizers.py runs the same checks on the real engine.
Pinned: merge-pair sequences, merged rows (1e-5, NaNs in place), threshold dynamics, Python RNG
consumption, cache stepping, vocab strings, save/load files, the CLI end to end.
"""
import hashlib
import json
import os
import random

import numpy as np
import pytest
import torch

from helpers import OracleEngine, nan_equal_close, oracle_distance, oracle_exp_map, oracle_project
from hyptokenizer_amd.synthetic import cjk_vocab
from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import (AdaptiveMergeCache, CandidateList, FastHyperbolicTokenizer,
                                                              MergeCandidate)
from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer


def seed_all(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def state_hash():
    return hashlib.sha256(repr(random.getstate()).encode()).hexdigest()


def make(cls, X, mode, thr=0.1, engine_factory=None, device="cpu", **kw):
    V = X.shape[0]
    eng = engine_factory(100000, X.shape[1], mode) if engine_factory else None
    return cls(vocab=cjk_vocab(V), embeddings=torch.nn.Parameter(torch.as_tensor(X).clone()), curvature=1.0,
               merge_threshold=thr, device=torch.device(device), use_approximate_search=False, sign_convention=mode,
               engine=eng, **kw)


def record(tok):
    pairs = []
    inner = tok._merge_tokens

    def wrapped(i, j):
        pairs.append((int(i), int(j)))
        return inner(i, j)

    tok._merge_tokens = wrapped
    return pairs


def check_sequences(golden_dir, mode, engine_factory, device):
    z = np.load(os.path.join(golden_dir, f"g3_sequences_{mode}.npz"))
    meta = json.load(open(os.path.join(golden_dir, f"g3_sequences_{mode}.json")))
    X = z["X"]
    V = X.shape[0]
    # standard tokenizer
    seed_all(42)
    tok = make(HyperbolicTokenizer, X, mode, engine_factory=engine_factory, device=device)
    pairs = record(tok)
    tok.optimize_merges(steps=meta["std_steps"], log_every=10 ** 9, parallel_eval=False)
    assert np.array_equal(np.array(pairs, np.int32).reshape(-1, 2), z["std_pairs"])
    rows = tok.embeddings.data[V:tok.current_vocab_size].cpu().numpy()
    assert nan_equal_close(rows, z["std_rows"], 1e-5)
    assert tok.merge_threshold == meta["std_threshold"]
    assert tok.vocab[V:] == meta["std_vocab_tail"]
    # incremental maintenance of the nearest pair: same pairs, same rows, bit for bit
    seed_all(42)
    itok = make(HyperbolicTokenizer, X, mode, engine_factory=engine_factory, device=device, incremental=True)
    ipairs = record(itok)
    itok.optimize_merges(steps=meta["std_steps"], log_every=10 ** 9, parallel_eval=False)
    assert np.array_equal(np.array(ipairs, np.int32).reshape(-1, 2), z["std_pairs"])
    irows = itok.embeddings.data[V:itok.current_vocab_size]
    assert torch.equal(irows.view(torch.int32).cpu(), tok.embeddings.data[V:tok.current_vocab_size].view(torch.int32).cpu())
    # fast tokenizer, two logging cadences (log steps consume the Python RNG)
    for key in ("fast", "fastlog"):
        seed_all(42)
        ftok = make(FastHyperbolicTokenizer, X, mode, engine_factory=engine_factory, device=device)
        fpairs = record(ftok)
        assert state_hash() == meta[f"{key}_random_state_before"]
        ftok.optimize_merges(steps=meta[f"{key}_steps"], log_every=meta[f"{key}_log_every"])
        assert np.array_equal(np.array(fpairs, np.int32).reshape(-1, 2), z[f"{key}_pairs"]), key
        rows = ftok.embeddings.data[V:ftok.current_vocab_size].cpu().numpy()
        assert nan_equal_close(rows, z[f"{key}_rows"], 1e-5), key
        assert ftok.merge_threshold == meta[f"{key}_threshold"]
        assert state_hash() == meta[f"{key}_random_state_after"], key
        assert len(ftok.cache.candidates) == meta[f"{key}_cache_len"]
        assert [list(m) for m in ftok.merge_history[:5]] == meta[f"{key}_merges"]
    # distance statistics with a pinned RNG (fast_hyperbolic_merge.py:433-465)
    seed_all(123)
    ftok = make(FastHyperbolicTokenizer, X, mode, engine_factory=engine_factory, device=device)
    st = ftok._compute_distance_statistics()
    for k in ("min", "max", "mean", "std"):
        assert abs(float(st[k]) - meta["stats"][k]) <= 1e-6, k
    assert state_hash() == meta["stats_random_state_after"]
    seed_all(123)
    small = make(FastHyperbolicTokenizer, X[:30], mode, engine_factory=engine_factory, device=device)
    st = small._compute_distance_statistics()
    for k in ("min", "max", "mean", "std"):
        assert abs(float(st[k]) - meta["stats_small"][k]) <= 1e-6, k
    # initial threshold rewrite (fast_hyperbolic_merge.py:493-505)
    seed_all(42)
    ftok = make(FastHyperbolicTokenizer, X, mode, thr=5.0, engine_factory=engine_factory, device=device)
    fpairs = record(ftok)
    ftok.optimize_merges(steps=3, log_every=1000)
    assert abs(ftok.merge_threshold - meta["rewrite_threshold"]) <= 1e-7 * max(1.0, meta["rewrite_threshold"])
    assert np.array_equal(np.array(fpairs, np.int32).reshape(-1, 2), z["rewrite_pairs"])


@pytest.mark.parametrize("mode", ["reference", "lorentz"])
def test_merge_sequences_match_reference(golden_dir, mode):
    check_sequences(golden_dir, mode, OracleEngine, "cpu")


def check_cli(golden_dir, mode, tmp_path, engine_factory, init_device="cpu", writer=True):
    """End-to-end CLI run on a copy of the reference's 45-line initial vocabulary (42 tokens after
    load_vocab drops blanks): vocab.json, merges.json, config.json identical, live rows close.
    ``writer=False``: a rank of a multi-process launch that does not write the files (rank 0 does)."""
    from hyptokenizer_amd.scripts import train_hyperbolic_tokenizer as T
    res = json.load(open(os.path.join(golden_dir, f"cli_{mode}.json")))
    arr = np.load(os.path.join(golden_dir, f"cli_{mode}.npz"))
    vocab_file = os.path.join(golden_dir, "vocab_initial.txt")
    assert len(T.load_vocab(vocab_file)) == res["n_vocab_initial"] == 42
    if engine_factory is not None:
        # the test double needs CPU tensors: run the CLI body with patched engine construction
        import hyptokenizer_amd.tokenizer.hyperbolic_merge as HMOD
        orig = HMOD.HyperbolicTokenizer._get_engine

        def patched(self):
            if self._engine is None:
                self._engine = engine_factory(self.max_vocab_size, self.embeddings.size(1), self.sign_convention)
            return orig(self)

        HMOD.HyperbolicTokenizer._get_engine = patched
        saved = (T.exp_map, T.project_to_hyperboloid, T.distance)
        T.exp_map, T.project_to_hyperboloid, T.distance = oracle_exp_map, oracle_project, oracle_distance
    try:
        for fast in (True, False):
            out = str(tmp_path / f"out_{mode}_{int(fast)}")
            T.train_tokenizer(vocab_path=vocab_file, output_dir=out, embedding_dim=5, curvature=1.0, merge_threshold=0.1,
                              merge_steps=8, log_every=4, target_vocab_size=500, seed=42, use_fast_tokenizer=fast,
                              no_faiss=True, sign_convention=mode, init_device=init_device)
            key = "fast" if fast else "std"
            if not writer:
                assert not os.path.exists(os.path.join(out, "vocab.json"))
                continue
            assert json.load(open(os.path.join(out, "vocab.json"))) == res[key]["vocab"]
            assert json.load(open(os.path.join(out, "merges.json"))) == res[key]["merges"]
            cfg = json.load(open(os.path.join(out, "config.json")))
            ref_cfg = res[key]["config"]
            assert set(cfg) == set(ref_cfg)
            for k in ref_cfg:
                if k == "merge_threshold":
                    assert abs(cfg[k] - ref_cfg[k]) <= 1e-7 * max(1.0, abs(ref_cfg[k]))
                else:
                    assert cfg[k] == ref_cfg[k], k
            emb = torch.load(os.path.join(out, "embeddings.pt"), weights_only=True).numpy()
            assert emb.shape == arr[f"{key}_embeddings"].shape
            assert nan_equal_close(emb, arr[f"{key}_embeddings"], 1e-5)
            st = json.load(open(os.path.join(out, "training_stats.json")))
            assert st["step"] == res[key]["training_stats"]["step"]
            assert st["vocab_size"] == res[key]["training_stats"]["vocab_size"]
            assert np.allclose(st["distortion"], res[key]["training_stats"]["distortion"], atol=1e-5, equal_nan=True)
    finally:
        if engine_factory is not None:
            HMOD.HyperbolicTokenizer._get_engine = orig
            T.exp_map, T.project_to_hyperboloid, T.distance = saved


def test_cpu_search_raises_without_engine():
    """no HIP device, no injected engine: the product refuses instead of falling back"""
    from hyptokenizer_amd.engine import HypMergeUnavailable
    X = np.zeros((5, 4), np.float32)
    X[:, 0] = 1
    tok = make(HyperbolicTokenizer, X, "lorentz")
    with pytest.raises(HypMergeUnavailable):
        tok._find_merge_candidates()
    with pytest.raises(HypMergeUnavailable):
        tok.optimize_merges(steps=1)
    ftok = make(FastHyperbolicTokenizer, X, "lorentz")
    with pytest.raises(HypMergeUnavailable):
        ftok.optimize_merges(steps=1)


def test_cache_lazy_batches_keep_reference_semantics():
    """batches served straight from a refresh's arrays behave like the reference's popped lists:
    same entries, same bookkeeping (hit_count per served pair, stats), mixing with materialised
    entries falls back to lists"""
    d = np.array([0.1, 0.2, 0.2, 0.3, 0.4], np.float32)
    i = np.array([0, 1, 2, 3, 4], np.int32)
    j = np.array([5, 6, 7, 8, 9], np.int32)
    cache = AdaptiveMergeCache(max_size=4)
    cache.add_batch(CandidateList(d, i, j, 5))
    assert len(cache) == 4                                   # truncated to max_size
    first = cache.get_best(2)
    assert len(first) == 2 and bool(first) and first[0] == MergeCandidate(float(d[0]), 0, 5)
    assert [(c.token_i, c.token_j) for c in first] == [(0, 5), (1, 6)]
    assert cache.hit_count == {(0, 5): 1, (1, 6): 1} and cache.get_stats()["hit_count"] == 2
    assert [c.token_i for c in cache.candidates] == [2, 3]   # reading the attribute materialises the rest
    rest = cache.get_best(100)                               # ... and later pops are plain lists
    assert isinstance(rest, list) and [c.token_i for c in rest] == [2, 3]
    assert cache.hit_count[(2, 7)] == 1 and cache.get_stats()["hit_count"] == 4 and len(cache) == 0
    assert cache.get_best(1) == [] and cache.miss_count == 1
    # served twice -> counted twice
    cache.add_batch(CandidateList(d[:1], i[:1], j[:1], 1))
    cache.get_best(1)
    assert cache.hit_count[(0, 5)] == 2


def test_incremental_state_invalidation():
    """the running minimum is recomputed when the threshold moves, the table is edited, or the
    search crosses the reference's n <= 100 compare branch; otherwise one row pass per step"""
    from hyptokenizer_amd.synthetic import lorentz_table
    X = lorentz_table(97, 12, seed=3, scale=0.05).numpy()
    calls = {"full": 0, "row": 0}

    class Counting(OracleEngine):
        def argmin(self, *a, **k):
            calls["full"] += 1
            return super().argmin(*a, **k)

        def row_argmin(self, *a, **k):
            calls["row"] += 1
            return super().row_argmin(*a, **k)

    full = make(HyperbolicTokenizer, X, "lorentz", thr=0.7, engine_factory=OracleEngine)
    inc = make(HyperbolicTokenizer, X, "lorentz", thr=0.7, engine_factory=Counting, incremental=True)
    for tok in (full, inc):
        tok.optimize_merges(steps=3, log_every=10 ** 9)          # 97 -> 100 rows
    # (the batched loop folds the new row's nearest partner in right after every merge: one row pass per step)
    assert calls == {"full": 1, "row": 3}
    for tok in (full, inc):
        tok.optimize_merges(steps=3, log_every=10 ** 9)          # crosses n = 100: threshold form changes once
    assert calls == {"full": 2, "row": 6}                       # 0.7 rounds differently in the two compare forms
    before = dict(calls)
    for tok in (full, inc):
        tok.merge_threshold = 0.6
        tok.optimize_merges(steps=2, log_every=10 ** 9)
    assert calls["full"] == before["full"] + 1 and calls["row"] == before["row"] + 2
    for tok in (full, inc):
        tok.embeddings.data[5] = tok.embeddings.data[50]
        tok.refresh_engine()
        tok.optimize_merges(steps=2, log_every=10 ** 9)
    assert calls["full"] == before["full"] + 2
    assert full.merge_history == inc.merge_history and len(inc.merge_history) == 10
    n = inc.current_vocab_size
    assert torch.equal(full.embeddings.data[:n].view(torch.int32), inc.embeddings.data[:n].view(torch.int32))


def test_cache_semantics():
    """AdaptiveMergeCache (fast_hyperbolic_merge.py:63-133) and the refresh quirk of section 3.2"""
    cache = AdaptiveMergeCache(max_size=5)
    assert cache.get_best(3) == [] and cache.miss_count == 1
    cache.add_batch([MergeCandidate(0.3, 1, 2), MergeCandidate(0.1, 3, 4), MergeCandidate(0.1, 0, 9),
                     MergeCandidate(0.2, 5, 6), MergeCandidate(0.5, 7, 8), MergeCandidate(0.4, 1, 3), MergeCandidate(0.9, 2, 3)])
    assert [(c.token_i, c.token_j) for c in cache.candidates] == [(3, 4), (0, 9), (5, 6), (1, 2), (1, 3)]   # stable, truncated
    best = cache.get_best(2)
    assert [(c.token_i, c.token_j) for c in best] == [(3, 4), (0, 9)] and len(cache.candidates) == 3
    st = cache.get_stats()
    assert st["size"] == 3 and st["hit_count"] == 2 and st["miss_count"] == 1
    lst = CandidateList(np.array([0.1, 0.2], np.float32), np.array([1, 2], np.int32), np.array([5, 6], np.int32), 7)
    assert len(lst) == 7 and bool(lst) and lst[0].token_j == 5 and [c.token_i for c in lst] == [1, 2]
    assert not CandidateList(np.empty(0, np.float32), np.empty(0, np.int32), np.empty(0, np.int32), 0)


def test_reference_unit_test_properties(tmp_path):
    """Properties the reference's tests/test_hyperbolic_tokenizer.py pins (vocab of 9, d = 5)."""
    vocab = ["<pad>", "<bos>", "<eos>", "<unk>", "a", "b", "c", "d", "e"]
    from hyptokenizer_amd.synthetic import lorentz_table
    emb = lorentz_table(len(vocab), 5, seed=42, scale=0.01)
    tok = HyperbolicTokenizer(vocab=vocab, embeddings=torch.nn.Parameter(emb), curvature=1.0, merge_threshold=0.5, lr=1e-3,
                              device=torch.device("cpu"), sign_convention="lorentz", engine=OracleEngine(100000, 6, "lorentz"))
    # test_initialization (:73-88)
    assert tok.vocab == vocab and tok.curvature == 1.0 and tok.merge_threshold == 0.5 and tok.lr == 1e-3
    assert all(tok.token2idx[t] == k for k, t in enumerate(vocab))
    assert torch.allclose(tok.embeddings[:9], emb)
    # test_find_merge_candidates (:107-130)
    tok.merge_threshold = 10.0
    cands = tok._find_merge_candidates()
    assert len(cands) == 36
    for i, j, dist in cands:
        assert isinstance(i, int) and isinstance(j, int) and isinstance(dist, float)
        assert 0 <= i < j < 9 and dist <= tok.merge_threshold
    # test_merge_tokens (:132-156)
    tok._merge_tokens(4, 5)
    assert len(tok.vocab) == 10 and tok.vocab[-1] == "ab" and tok.token2idx["ab"] == 9
    assert tok.merge_history == [("a", "b", "ab")] and tok.current_vocab_size == 10
    # test_tokenize_encode_decode (:158-185)
    tok.vocab.append("cd")
    tok.token2idx["cd"] = 10
    tok.merge_history.append(("c", "d", "cd"))
    assert tok.tokenize("abcde") == ["ab", "cd", "e"]
    ids = tok.encode("abcde")
    assert ids == [tok.token2idx["ab"], tok.token2idx["cd"], tok.token2idx["e"]] and tok.decode(ids) == "abcde"
    assert tok.encode("z") == [3]
    # table full -> ValueError (:343-344)
    small = HyperbolicTokenizer(vocab=vocab, embeddings=torch.nn.Parameter(emb), max_vocab_size=9, device=torch.device("cpu"),
                                engine=OracleEngine(9, 6, "reference"))
    with pytest.raises(ValueError):
        small._merge_tokens(0, 1)
    # test_save_load (:187-228)
    tok.vocab.pop()
    del tok.token2idx["cd"]
    tok.merge_history.pop()
    path = str(tmp_path / "tok")
    tok.save(path)
    for f in ("vocab.json", "embeddings.pt", "merges.json", "config.json"):
        assert os.path.exists(os.path.join(path, f))
    cfg = json.load(open(os.path.join(path, "config.json")))
    assert set(cfg) == {"curvature", "merge_threshold", "embedding_dim", "max_vocab_size", "use_approximate_search"}
    back = HyperbolicTokenizer.load(path, device=torch.device("cpu"))
    assert back.vocab == tok.vocab and back.current_vocab_size == 10 and back.curvature == tok.curvature
    assert back.merge_threshold == tok.merge_threshold
    assert torch.equal(back.embeddings[:10], tok.embeddings[:10].detach())
    assert [tuple(m) for m in back.merge_history] == tok.merge_history
    assert all(back.token2idx[t] == k for t, k in tok.token2idx.items())
    assert back.tokenize("abe") == ["ab", "e"]


@pytest.mark.parametrize("mode", ["reference", "lorentz"])
def test_cli_matches_reference(golden_dir, mode, tmp_path):
    check_cli(golden_dir, mode, tmp_path, OracleEngine)


def test_pair_sampler_consumes_the_generator_like_random_sample():
    """_compute_distance_statistics draws its pairs through _sample_pairs: same pairs and same generator state as the
    reference's ``random.sample(range(n), 2)`` calls (fast_hyperbolic_merge.py:448-449), for large and tiny n."""
    import random
    from hyptokenizer_amd.tokenizer import fast_hyperbolic_merge as F
    for n, count in ((50000, 1000), (22, 300), (21, 50), (5, 10), (2, 1)):
        random.seed(1234 + n)
        ii, jj = F._sample_pairs(n, count)
        state = random.getstate()
        random.seed(1234 + n)
        want = [random.sample(range(n), 2) for _ in range(count)]
        assert [w[0] for w in want] == ii and [w[1] for w in want] == jj
        assert random.getstate() == state
    assert F._PAIR_SAMPLER_OK is True          # the shortcut is in use on this interpreter
