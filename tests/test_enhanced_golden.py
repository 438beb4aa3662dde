"""BASELINE config 5 (EnhancedFastHyperbolicTokenizer) against the G5 goldens captured from the REFERENCE
(tests/golden/make_golden.py g5_enhanced: tokenizer/enhanced_fast_hyperbolic_merge.py imported through the
two-name shim of SURVEY.md F8).

CPU tests: the class's host logic (RNG order, scoring, sort, phases, thresholds, save / load) on the
oracle-backed engine double; ``tests/test_gpu_enhanced.py`` runs the same comparisons through the HIP kernels.
Bar: candidate order and merge pairs identical; scores within 1e-5; merged rows within 1e-5.

The adaptive-curvature step is compared with the reference's own loss code run under a ONE-LINE patch (the
reference as shipped raises at loss.backward(), F8): "parity unpinned" with respect to the shipped behaviour.
"""
import json
import os
import random

import numpy as np
import pytest
import torch

from helpers import OracleEngine, nan_equal_close

ATOL = 1e-5

CONFIGS = {
    "freq_hier": dict(use_frequency_aware=True, use_hierarchical=True, use_adaptive_curvature=False,
                      use_compression_aware=False),
    "freq_comp_adapt": dict(use_frequency_aware=True, use_hierarchical=False, use_adaptive_curvature=True,
                            use_compression_aware=True, optimize_curvature_freq=10 ** 6),
    "freq_only": dict(use_frequency_aware=True, use_hierarchical=False, use_adaptive_curvature=False,
                      use_compression_aware=False),
}


def pair_frequencies(vocab):
    """the integer rule of make_golden.synthetic_pair_frequencies"""
    n = len(vocab)
    return {(vocab[a], vocab[b]): 1 + (a * 31 + b * 17) % 997
            for a in range(n) for b in range(n) if (a * 7 + b * 13) % 5 == 0}


def load_g5(golden_dir, mode):
    z = np.load(os.path.join(golden_dir, f"g5_enhanced_{mode}.npz"))
    with open(os.path.join(golden_dir, f"g5_enhanced_{mode}.json")) as f:
        meta = json.load(f)
    return z, meta


def seed_all(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def make_tok(z, meta, mode, name, make_engine, device="cpu", max_vocab_size=None, **extra):
    from hyptokenizer_amd.synthetic import cjk_vocab
    from hyptokenizer_amd.tokenizer.enhanced_fast_hyperbolic_merge import EnhancedFastHyperbolicTokenizer
    X = torch.from_numpy(z["X"])
    n = X.shape[0]
    vocab = cjk_vocab(n)
    flags = dict(CONFIGS[name]) if name in CONFIGS else {}
    flags.update(extra)
    if flags.get("use_compression_aware"):
        flags["corpus_sample"] = list(meta["corpus_sample"])
    rows = max_vocab_size or (n + 64)
    tok = EnhancedFastHyperbolicTokenizer(
        vocab=vocab, embeddings=torch.nn.Parameter(X.clone()), curvature=1.0, merge_threshold=meta["thr"],
        device=torch.device(device), use_approximate_search=False, max_vocab_size=rows, sign_convention=mode,
        engine=make_engine(rows, X.shape[1], mode), **flags)
    if flags.get("use_frequency_aware"):
        tok.pair_frequencies = pair_frequencies(vocab)
    return tok


def oracle_engine(rows, d1, mode):
    return OracleEngine(rows, d1, mode, fast=False)


def check_scores(z, meta, mode, make_engine, device="cpu"):
    from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer, MergeCandidate
    for name in CONFIGS:
        seed_all(42)
        tok = make_tok(z, meta, mode, name, make_engine, device)
        base = FastHyperbolicTokenizer._find_merge_candidates_fast(tok)
        assert len(base) == meta[f"score_{name}_n_candidates"]
        cands = list(base[:64])
        assert [c.token_i for c in cands] == z[f"score_{name}_i"].tolist()
        assert [c.token_j for c in cands] == z[f"score_{name}_j"].tolist()
        torch.manual_seed(123)
        scored = [tok._score_candidate(MergeCandidate(c.distance, c.token_i, c.token_j)) for c in cands]
        for fld in ("frequency_score", "semantic_score", "compression_score", "morphology_score", "combined_score"):
            got = np.array([getattr(s, fld) for s in scored], np.float64)
            assert nan_equal_close(got, z[f"score_{name}_{fld}"], ATOL), (name, fld)
        # one candidate at a time or all at once: the same torch RNG calls in the same order
        torch.manual_seed(123)
        batch = tok._score_candidates([c.distance for c in cands], [c.token_i for c in cands], [c.token_j for c in cands])
        assert [b.combined_score for b in batch] == [s.combined_score for s in scored] or mode == "reference"


def check_order(z, meta, mode, make_engine, device="cpu"):
    for name in ("freq_hier", "freq_only"):
        seed_all(42)
        tok = make_tok(z, meta, mode, name, make_engine, device)
        torch.manual_seed(321)
        first = tok._find_merge_candidates_fast()
        second = tok._find_merge_candidates_fast()
        for tag, lst in (("first", first), ("second", second)):
            assert [c.token_i for c in lst] == z[f"order_{name}_{tag}_i"].tolist(), (name, tag)
            assert [c.token_j for c in lst] == z[f"order_{name}_{tag}_j"].tolist(), (name, tag)
            assert nan_equal_close([c.combined_score for c in lst], z[f"order_{name}_{tag}_score"], ATOL), (name, tag)
        assert len(tok.cache.candidates) == meta[f"order_{name}_cache_len"]


def check_sequences(z, meta, mode, make_engine, device="cpu", tmp_path=None):
    n = z["X"].shape[0]
    for name in CONFIGS:
        info = meta[f"seq_{name}"]
        kw = dict(info["kwargs"])
        if "phase_transition_steps" in kw:
            kw["phase_transition_steps"] = {int(a): b for a, b in kw["phase_transition_steps"].items()}
        seed_all(42)
        tok = make_tok(z, meta, mode, name, make_engine, device)
        torch.manual_seed(777)
        tok.optimize_merges(**kw)
        want = z[f"seq_{name}_pairs"].tolist()
        assert [list(m) for m in tok.merge_history] == info["merges"], name
        m = tok.current_vocab_size
        assert m == n + len(want)
        rows = tok.embeddings.data[n:m].cpu().numpy()
        assert nan_equal_close(rows, z[f"seq_{name}_rows"], ATOL), name
        assert tok.merge_threshold == pytest.approx(info["threshold"], rel=1e-12)
        assert tok.current_phase == info["phase"]
        assert float(tok.get_curvature()) == pytest.approx(info["curvature"], abs=1e-7)
        assert len(tok.cache.candidates) == info["cache_len"]
        if tok.use_adaptive_curvature:
            assert [list(p) for p in tok.merge_pairs] == info["merge_pairs"]
        stats = {str(k): v for k, v in getattr(tok, "training_stats", {}).items()}
        assert set(stats) == set(info["training_stats"])
        for step, rec in info["training_stats"].items():
            for key, val in rec.items():
                if isinstance(val, float):
                    assert (np.isnan(val) and np.isnan(stats[step][key])) or stats[step][key] == pytest.approx(val, abs=ATOL), (name, step, key)
                else:
                    assert stats[step][key] == val
        # the loop consumed both generators exactly as the reference did
        import hashlib
        assert hashlib.sha256(repr(random.getstate()).encode()).hexdigest() == info["random_state_after"], name
        assert hashlib.sha256(torch.get_rng_state().numpy().tobytes()).hexdigest() == info["torch_state_after"], name
        if tmp_path is not None:
            check_save_load(tok, meta, name, make_engine, mode, tmp_path / name, device)


def check_save_load(tok, meta, name, make_engine, mode, path, device):
    """save() writes the reference's files and keys (enhanced...:1211-1298); load() restores the object"""
    from hyptokenizer_amd.tokenizer.enhanced_fast_hyperbolic_merge import EnhancedFastHyperbolicTokenizer
    tok.save(str(path))
    assert sorted(os.listdir(path)) == meta[f"save_{name}_files"]
    ref_cfg = meta[f"save_{name}_enhanced_config.json"]
    with open(path / "enhanced_config.json") as f:
        cfg = json.load(f)
    assert set(cfg) == set(ref_cfg)
    for key, val in ref_cfg.items():
        if isinstance(val, float):
            assert cfg[key] == pytest.approx(val, abs=1e-7), key
        else:
            assert cfg[key] == val, key
    for fn in ("frequencies.json", "hierarchical_data.json", "merges.json", "vocab.json"):
        if f"save_{name}_{fn}" in meta:
            with open(path / fn) as f:
                assert json.load(f) == meta[f"save_{name}_{fn}"], fn
    rows = tok.max_vocab_size
    back = EnhancedFastHyperbolicTokenizer.load(str(path), device=torch.device(device), sign_convention=mode,
                                                engine=make_engine(rows, tok.embeddings.size(1), mode))
    assert back.vocab == tok.vocab and back.merge_history == [list(m) for m in tok.merge_history]
    assert back.current_phase == tok.current_phase and back.current_vocab_size == tok.current_vocab_size
    n = tok.current_vocab_size
    assert nan_equal_close(back.embeddings.data[:n].cpu().numpy(), tok.embeddings.data[:n].cpu().numpy(), 0.0)
    assert back.pair_frequencies == tok.pair_frequencies
    if tok.use_adaptive_curvature:
        assert float(back.curvature) == float(tok.curvature)
        assert [tuple(p) for p in back.merge_pairs] == [tuple(p) for p in tok.merge_pairs]


def check_curvature(z, meta, make_engine, device="cpu"):
    """three Adam steps on c + re-projection, against the patched reference (module docstring)"""
    from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
    n = z["X"].shape[0]
    seed_all(42)
    tok = make_tok(z, meta, "lorentz", None, make_engine, device, max_vocab_size=n + 40, use_frequency_aware=False,
                   use_hierarchical=False, use_adaptive_curvature=True, use_compression_aware=False,
                   optimize_curvature_freq=10 ** 6, curvature_lr=0.01)
    first = FastHyperbolicTokenizer._find_merge_candidates_fast(tok)
    for q in (0, 5, 10, 20, 30, 45):
        tok._merge_tokens(first[q].token_i, first[q].token_j)
    assert [list(p) for p in tok.merge_pairs] == meta["curv_merge_pairs"]
    assert tok.current_vocab_size == meta["curv_n"]
    assert nan_equal_close(tok.embeddings.data.cpu().numpy(), z["curv_rows_before"], ATOL)
    torch.manual_seed(99)
    for rec in meta["curv_steps"]:
        emb = tok.embeddings.detach()
        state = torch.get_rng_state()
        h = float(tok._compute_hierarchy_preservation_loss(emb))
        d = float(tok._compute_distortion_loss(emb))
        torch.set_rng_state(state)
        tok._optimize_curvature(emb)
        tok._project_embeddings()
        assert h == pytest.approx(rec["hierarchy_loss"], abs=2e-6)
        assert d == pytest.approx(rec["distortion_loss"], abs=2e-6)
        assert float(tok.curvature) == pytest.approx(rec["curvature_after"], abs=2e-6)
    import hashlib
    assert hashlib.sha256(torch.get_rng_state().numpy().tobytes()).hexdigest() == meta["curv_torch_state_after"]
    assert nan_equal_close(tok.embeddings.data.cpu().numpy(), z["curv_rows_after"], ATOL)


@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_score_candidate(golden_dir, mode):
    z, meta = load_g5(golden_dir, mode)
    check_scores(z, meta, mode, oracle_engine)


@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_candidate_order(golden_dir, mode):
    z, meta = load_g5(golden_dir, mode)
    check_order(z, meta, mode, oracle_engine)


@pytest.mark.parametrize("mode", ["lorentz", "reference"])
def test_merge_sequences_and_save_load(golden_dir, mode, tmp_path):
    z, meta = load_g5(golden_dir, mode)
    check_sequences(z, meta, mode, oracle_engine, tmp_path=tmp_path)


def test_curvature_step_against_patched_reference(golden_dir):
    z, meta = load_g5(golden_dir, "lorentz")
    check_curvature(z, meta, oracle_engine)


def test_row_means_equal_numpy_mean_of_a_list():
    """the vectorised mean of _semantic_coherence_batch is np.mean(list) of the reference, bit for bit"""
    from hyptokenizer_amd.tokenizer.enhanced_fast_hyperbolic_merge import _row_means
    rs = np.random.RandomState(3)
    for ns in (1, 2, 7, 8, 9, 33, 50):
        D = (rs.rand(400, ns) * 3).astype(np.float32)
        keep = np.ones_like(D, bool)
        keep[5, 0] = False
        if ns > 1:
            keep[9, ns - 1] = False
        got = _row_means(D, keep)
        for r in range(D.shape[0]):
            vals = [float(v) for v, k in zip(D[r].tolist(), keep[r].tolist()) if k]
            want = np.mean(vals) if vals else np.nan
            assert (np.isnan(want) and np.isnan(got[r])) or got[r] == want, (ns, r)


def test_randperm_prefix_helper_equals_torch_randperm():
    """hm_randperm_prefix (host helper of the library): same samples as torch.randperm(n)[:ns] and the same
    generator state afterwards, across block boundaries of the MT19937 recurrence"""
    from hyptokenizer_amd.tokenizer.enhanced_fast_hyperbolic_merge import randperm_prefixes
    for seed, n, ns, count in [(0, 1, 1, 3), (1, 2, 2, 5), (2, 50, 50, 4), (3, 51, 50, 7), (4, 623, 50, 9), (5, 624, 50, 9),
                               (6, 625, 50, 9), (7, 1249, 17, 6), (8, 100000, 50, 5), (9, 3001, 50, 300), (10, 40, 40, 40)]:
        torch.manual_seed(seed)
        torch.rand(seed % 7 + 1)                           # an arbitrary position inside a block
        state = torch.get_rng_state()
        want = np.stack([torch.randperm(n)[:ns].numpy() for _ in range(count)])
        after = torch.get_rng_state()
        torch.set_rng_state(state)
        got = randperm_prefixes(n, ns, count)
        assert np.array_equal(got, want), (seed, n, ns)
        assert torch.equal(torch.get_rng_state(), after), (seed, n, ns)
        assert torch.equal(torch.randperm(11), (torch.set_rng_state(after), torch.randperm(11))[1])
