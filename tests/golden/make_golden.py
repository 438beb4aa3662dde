#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container: it imports the reference's Python modules from
/root/reference (never copied, never shipped) and records inputs + outputs as small .npz / .json
fixtures.  Refuses to run when /root/reference is absent (e.g. on the GPU box).

Two oracle modes:
  reference : the modules exactly as shipped (all distances 0.0, midpoints NaN -- SURVEY F2/F3)
  lorentz   : the same code with the sign of the Minkowski form flipped:
              ``embedding.lorentz_model.minkowski_dot`` negated (fixes distance/log_map) and
              ``batch_distance`` invoked as ``orig(x, -y, c)`` (it inlines its own dot product).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [reference|lorentz|all] [all|g5|g6|g7]
"""
from __future__ import annotations

import hashlib
import json
import os
import random
import sys
import tempfile
import warnings

REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("make_golden.py: /root/reference is not present; golden vectors can only be regenerated "
             "in the build container.")

os.environ.setdefault("TQDM_DISABLE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import embedding.lorentz_model as L  # noqa: E402  (reference)
import tokenizer.hyperbolic_merge as HM  # noqa: E402  (reference)
import tokenizer.fast_hyperbolic_merge as FM  # noqa: E402  (reference)

from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table  # noqa: E402  (ours: inputs only)

import logging  # noqa: E402

logging.disable(logging.CRITICAL)

_ORIG = {
    "minkowski_dot": L.minkowski_dot,
    "batch_distance": L.batch_distance,
}


def set_mode(mode: str) -> None:
    """Install / remove the two sign patches."""
    if mode == "reference":
        L.minkowski_dot = _ORIG["minkowski_dot"]
        bd = _ORIG["batch_distance"]
    elif mode == "lorentz":
        L.minkowski_dot = lambda a, b: -_ORIG["minkowski_dot"](a, b)
        bd = lambda x, y, c=1.0: _ORIG["batch_distance"](x, -y, c)  # noqa: E731
    else:
        raise ValueError(mode)
    L.batch_distance = bd
    HM.batch_distance = bd
    HM.batch_distance_compiled = bd
    FM.batch_distance = bd


def seed_all(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


# ---------------------------------------------------------------------------------------------
# G1: primitives
# ---------------------------------------------------------------------------------------------
def g1_primitives(mode: str) -> None:
    out = {}
    for d in (10, 50, 100):
        for scale in (0.01, 0.05, 0.5):
            tag = f"d{d}_s{scale}"
            X = lorentz_table(64, d, seed=42, scale=scale)
            out[f"{tag}_X"] = X.numpy()
            out[f"{tag}_bd"] = L.batch_distance(X, X, 1.0).numpy()
            out[f"{tag}_bd_c2"] = L.batch_distance(X, X, 2.0).numpy()
            a, b = X[0:63], X[1:64]
            out[f"{tag}_dist"] = L.distance(a, b, 1.0).numpy()
            out[f"{tag}_mdot"] = L.minkowski_dot(a, b).numpy()
            lg = L.log_map(a, b, 1.0)
            out[f"{tag}_log"] = lg.numpy()
            for w in (0.5, 1.0 / 3.0, 0.75):
                v = lg * w
                ex = L.exp_map(a, v, 1.0)
                out[f"{tag}_exp_w{w:.4f}"] = ex.numpy()
                out[f"{tag}_mid_w{w:.4f}"] = L.project_to_hyperboloid(ex, 1.0).numpy()
            # projection of off-manifold points
            g = torch.Generator().manual_seed(7)
            P = torch.randn(16, d + 1, generator=g)
            out[f"{tag}_P"] = P.numpy()
            out[f"{tag}_proj"] = L.project_to_hyperboloid(P, 1.0).numpy()
            out[f"{tag}_proj_c2"] = L.project_to_hyperboloid(P, 2.0).numpy()
    # edge cases: identical rows, origin, a NaN row, a zero (unused) row
    d = 10
    X = lorentz_table(4, d, seed=5, scale=0.05)
    origin = torch.zeros(1, d + 1)
    origin[0, 0] = 1.0
    E = torch.cat([origin, origin, X[0:1], X[0:1], X[1:2],
                   torch.full((1, d + 1), float("nan")), torch.zeros(1, d + 1)], 0)
    out["edge_X"] = E.numpy()
    out["edge_bd"] = L.batch_distance(E, E, 1.0).numpy()
    a = E[[0, 2, 2, 0, 5]]
    b = E[[1, 3, 4, 4, 4]]
    out["edge_pairs_a"] = np.array([0, 2, 2, 0, 5], np.int32)
    out["edge_pairs_b"] = np.array([1, 3, 4, 4, 4], np.int32)
    out["edge_dist"] = L.distance(a, b, 1.0).numpy()
    lg = L.log_map(a, b, 1.0)
    out["edge_log"] = lg.numpy()
    out["edge_mid"] = L.project_to_hyperboloid(L.exp_map(a, lg * 0.5, 1.0), 1.0).numpy()
    np.savez_compressed(os.path.join(HERE, f"g1_primitives_{mode}.npz"), **out)


# ---------------------------------------------------------------------------------------------
# G2: candidate search
# ---------------------------------------------------------------------------------------------
def _mk_tok(cls, X, thr, **kw):
    vocab = cjk_vocab(X.shape[0])
    return cls(vocab=vocab, embeddings=torch.nn.Parameter(X.clone()), curvature=1.0, merge_threshold=thr,
               device=torch.device("cpu"), use_approximate_search=False, **kw)


def g2_candidates(mode: str) -> None:
    out = {}
    keep = 12000
    cfgs = [(64, 10, 0.05), (101, 10, 0.05), (257, 10, 0.05), (1000, 10, 0.05), (300, 50, 0.05)]
    for (n, d, scale) in cfgs:
        X = lorentz_table(n, d, seed=42, scale=scale)
        out[f"n{n}_d{d}_X"] = X.numpy()
        # thresholds: none / ~1e3 / >1e4 candidates (lorentz); literal mode: everything is a candidate
        full = L.batch_distance(X, X, 1.0)
        iu = torch.triu_indices(n, n, 1)
        dd = full[iu[0], iu[1]].sort().values
        thrs = [float(dd[0]) * 0.5 if mode == "lorentz" else 0.0,
                float(dd[min(1000, dd.numel() - 1)]) if mode == "lorentz" else 0.1,
                float(dd[min(15000, dd.numel() - 1)]) if mode == "lorentz" else 0.5]
        if n == 1000 and mode == "lorentz":
            thrs.append(0.1)
        out[f"n{n}_d{d}_thr"] = np.array(thrs, np.float64)
        for ti, thr in enumerate(thrs):
            if mode == "reference" and n == 1000 and ti == 2:
                continue  # 499 500 candidates through the reference's Python loop twice: skip
            tok = _mk_tok(HM.HyperbolicTokenizer, X, thr)
            cand = tok._find_merge_candidates()
            key = f"n{n}_d{d}_t{ti}"
            out[f"{key}_std_count"] = np.int64(len(cand))
            out[f"{key}_std_i"] = np.array([c[0] for c in cand[:keep]], np.int32)
            out[f"{key}_std_j"] = np.array([c[1] for c in cand[:keep]], np.int32)
            out[f"{key}_std_d"] = np.array([c[2] for c in cand[:keep]], np.float32)
            ftok = _mk_tok(FM.FastHyperbolicTokenizer, X, thr)
            fc = ftok._find_merge_candidates_fast()
            out[f"{key}_fast_count"] = np.int64(len(fc))
            out[f"{key}_fast_i"] = np.array([c.token_i for c in fc[:keep]], np.int32)
            out[f"{key}_fast_j"] = np.array([c.token_j for c in fc[:keep]], np.int32)
            out[f"{key}_fast_d"] = np.array([c.distance for c in fc[:keep]], np.float32)
            out[f"{key}_cache_len"] = np.int64(len(ftok.cache.candidates))
    np.savez_compressed(os.path.join(HERE, f"g2_candidates_{mode}.npz"), **out)


# ---------------------------------------------------------------------------------------------
# G3 / G4: merge sequences, threshold dynamics, distance statistics
# ---------------------------------------------------------------------------------------------
def _record(tok):
    pairs = []
    orig = tok._merge_tokens

    def wrapped(i, j):
        pairs.append((int(i), int(j)))
        return orig(i, j)

    tok._merge_tokens = wrapped
    return pairs


def _state_hash() -> str:
    return hashlib.sha256(repr(random.getstate()).encode()).hexdigest()


def g3_sequences(mode: str) -> None:
    out = {}
    meta = {}
    V, d, scale, thr = 1000, 10, 0.05, 0.1
    X = lorentz_table(V, d, seed=42, scale=scale)
    out["X"] = X.numpy()
    std_steps = 200 if mode == "lorentz" else 3
    # standard tokenizer (hyperbolic_merge.py:357-412), parallel_eval has no effect on the result
    seed_all(42)
    tok = _mk_tok(HM.HyperbolicTokenizer, X, thr)
    pairs = _record(tok)
    tok.optimize_merges(steps=std_steps, log_every=10 ** 9, parallel_eval=False)
    n = tok.current_vocab_size
    out["std_pairs"] = np.array(pairs, np.int32).reshape(-1, 2)
    out["std_rows"] = tok.embeddings.data[V:n].numpy().copy()
    meta["std_threshold"] = tok.merge_threshold
    meta["std_vocab_tail"] = tok.vocab[V:n]
    meta["std_steps"] = std_steps
    # fast tokenizer (fast_hyperbolic_merge.py:467-576)
    for steps, log_every, key in ((200, 1000, "fast"), (250, 50, "fastlog")):
        seed_all(42)
        ftok = _mk_tok(FM.FastHyperbolicTokenizer, X, thr)
        fpairs = _record(ftok)
        h0 = _state_hash()
        ftok.optimize_merges(steps=steps, log_every=log_every)
        n = ftok.current_vocab_size
        out[f"{key}_pairs"] = np.array(fpairs, np.int32).reshape(-1, 2)
        out[f"{key}_rows"] = ftok.embeddings.data[V:n].numpy().copy()
        meta[f"{key}_threshold"] = ftok.merge_threshold
        meta[f"{key}_steps"] = steps
        meta[f"{key}_log_every"] = log_every
        meta[f"{key}_random_state_before"] = h0
        meta[f"{key}_random_state_after"] = _state_hash()
        meta[f"{key}_cache_len"] = len(ftok.cache.candidates)
        meta[f"{key}_merges"] = [list(m) for m in ftok.merge_history[:5]]
    # G4: distance statistics with a pinned Python RNG
    seed_all(123)
    ftok = _mk_tok(FM.FastHyperbolicTokenizer, X, thr)
    st = ftok._compute_distance_statistics()
    meta["stats"] = {k: float(v) for k, v in st.items()}
    meta["stats_random_state_after"] = _state_hash()
    seed_all(123)
    small = _mk_tok(FM.FastHyperbolicTokenizer, X[:30], thr)
    st = small._compute_distance_statistics()
    meta["stats_small"] = {k: float(v) for k, v in st.items()}
    # a second scale where the CLI-style "threshold above max" rewrite fires
    # (fast_hyperbolic_merge.py:502-505)
    seed_all(42)
    ftok = _mk_tok(FM.FastHyperbolicTokenizer, X, 5.0)
    fpairs = _record(ftok)
    ftok.optimize_merges(steps=3, log_every=1000)
    meta["rewrite_threshold"] = ftok.merge_threshold
    out["rewrite_pairs"] = np.array(fpairs, np.int32).reshape(-1, 2)
    np.savez_compressed(os.path.join(HERE, f"g3_sequences_{mode}.npz"), **out)
    with open(os.path.join(HERE, f"g3_sequences_{mode}.json"), "w") as f:
        json.dump(meta, f, indent=1, ensure_ascii=False)


# ---------------------------------------------------------------------------------------------
# CLI run on the reference's own vocab_initial.txt
# ---------------------------------------------------------------------------------------------
def g_cli(mode: str) -> None:
    import scripts.train_hyperbolic_tokenizer as T  # reference CLI module

    vocab_path = os.path.join(REF, "data/processed/wiki/vocab_initial.txt")
    res = {}
    arrays = {}
    for fast in (True, False):
        with tempfile.TemporaryDirectory() as td:
            T.train_tokenizer(vocab_path=vocab_path, output_dir=td, embedding_dim=5, curvature=1.0,
                              merge_threshold=0.1, merge_steps=8, log_every=4, target_vocab_size=500, seed=42,
                              use_fast_tokenizer=fast, no_faiss=True)
            key = "fast" if fast else "std"
            res[key] = {
                "vocab": json.load(open(os.path.join(td, "vocab.json"))),
                "merges": json.load(open(os.path.join(td, "merges.json"))),
                "config": json.load(open(os.path.join(td, "config.json"))),
                "training_stats": json.load(open(os.path.join(td, "training_stats.json"))),
            }
            emb = torch.load(os.path.join(td, "embeddings.pt"), weights_only=True)
            arrays[f"{key}_embeddings"] = emb.numpy().copy()
    # the initial table the CLI builds for this seed (pins RNG order of initialize_embeddings)
    T.set_seeds(42)
    vocab = T.load_vocab(vocab_path)
    arrays["init_embeddings"] = T.initialize_embeddings(vocab, 5, 1.0, torch.device("cpu")).numpy().copy()
    res["n_vocab_initial"] = len(vocab)
    np.savez_compressed(os.path.join(HERE, f"cli_{mode}.npz"), **arrays)
    with open(os.path.join(HERE, f"cli_{mode}.json"), "w") as f:
        json.dump(res, f, indent=1, ensure_ascii=False)


# ---------------------------------------------------------------------------------------------
# G5: enhanced tokenizer (BASELINE config 5: frequency-aware scoring + adaptive curvature)
# ---------------------------------------------------------------------------------------------
def _enhanced_module():
    """tokenizer/enhanced_fast_hyperbolic_merge.py does not import as shipped: it takes
    poincare_to_lorentz / lorentz_to_poincare from embedding.lorentz_model, where they do not live.
    The two names are injected from embedding.poincare_ball (neither is ever called on this path)."""
    import embedding.poincare_ball as P
    L.poincare_to_lorentz = P.poincare_to_lorentz
    L.lorentz_to_poincare = P.lorentz_to_poincare
    import tokenizer.enhanced_fast_hyperbolic_merge as EM
    return EM


def synthetic_pair_frequencies(vocab):
    """Deterministic pair-frequency table over a vocabulary (the reference builds it from a corpus,
    enhanced...:266-289): every fifth ordered pair gets a pseudo-random count.  Pure integer rule,
    so that tests rebuild it without the reference."""
    n = len(vocab)
    out = {}
    for a in range(n):
        for b in range(n):
            if (a * 7 + b * 13) % 5 == 0:
                out[(vocab[a], vocab[b])] = 1 + (a * 31 + b * 17) % 997
    return out


def synthetic_corpus_sample(vocab, cands, seed=11, lines=6, chunks=12):
    """Text lines in which the concatenations of some near pairs occur (so that the compression score
    of enhanced...:849-899 is non-zero for them); stored in the fixture as data."""
    rs = np.random.RandomState(seed)
    n = len(vocab)
    top = cands[:40]
    out = []
    for _ in range(lines):
        parts = []
        for _ in range(chunks):
            if top and rs.rand() < 0.5:
                c = top[int(rs.randint(0, len(top)))]
                parts.append(vocab[c.token_i] + vocab[c.token_j])
            else:
                parts.append(vocab[int(rs.randint(0, n))])
        out.append("".join(parts))
    return out


def _torch_state_hash() -> str:
    return hashlib.sha256(torch.get_rng_state().numpy().tobytes()).hexdigest()


G5_CONFIGS = {
    # name: (n, d, thr, flags)
    "freq_hier": dict(use_frequency_aware=True, use_hierarchical=True, use_adaptive_curvature=False,
                      use_compression_aware=False),
    "freq_comp_adapt": dict(use_frequency_aware=True, use_hierarchical=False, use_adaptive_curvature=True,
                            use_compression_aware=True, optimize_curvature_freq=10 ** 6),
    "freq_only": dict(use_frequency_aware=True, use_hierarchical=False, use_adaptive_curvature=False,
                      use_compression_aware=False),
}


def _mk_enh(EM, X, thr, flags, max_vocab_size=None, curvature=1.0, corpus=None):
    vocab = cjk_vocab(X.shape[0])
    kw = dict(flags)
    if kw.get("use_compression_aware"):
        kw["corpus_sample"] = list(corpus or [])
    tok = EM.EnhancedFastHyperbolicTokenizer(
        vocab=vocab, embeddings=torch.nn.Parameter(X.clone()), curvature=curvature, merge_threshold=thr,
        device=torch.device("cpu"), use_approximate_search=False,
        max_vocab_size=max_vocab_size or (X.shape[0] + 64), **kw)
    if kw.get("use_frequency_aware"):
        tok.pair_frequencies = synthetic_pair_frequencies(vocab)
    return tok


def g5_enhanced(mode: str) -> None:
    EM = _enhanced_module()
    out, meta = {}, {}
    lor = mode == "lorentz"
    n, d, scale = (300, 10, 0.05) if lor else (40, 10, 0.05)
    thr = 0.1
    X = lorentz_table(n, d, seed=42, scale=scale)
    out["X"] = X.numpy()
    meta["n"], meta["d"], meta["thr"] = n, d, thr
    seed_all(42)
    probe = _mk_tok(FM.FastHyperbolicTokenizer, X, thr)
    corpus = synthetic_corpus_sample(cjk_vocab(n), list(probe._find_merge_candidates_fast()[:40]))
    meta["corpus_sample"] = corpus

    # (a) _score_candidate on the first 64 candidates of the parent's refresh (enhanced...:903-990)
    for name, flags in G5_CONFIGS.items():
        seed_all(42)
        tok = _mk_enh(EM, X, thr, flags, corpus=corpus)
        base = FM.FastHyperbolicTokenizer._find_merge_candidates_fast(tok)
        cands = list(base[:64])
        torch.manual_seed(123)
        scored = [tok._score_candidate(c) for c in cands]
        out[f"score_{name}_i"] = np.array([c.token_i for c in cands], np.int32)
        out[f"score_{name}_j"] = np.array([c.token_j for c in cands], np.int32)
        out[f"score_{name}_d"] = np.array([c.distance for c in cands], np.float32)
        for fld in ("frequency_score", "semantic_score", "compression_score", "morphology_score", "combined_score"):
            out[f"score_{name}_{fld}"] = np.array([getattr(s, fld) for s in scored], np.float64)
        meta[f"score_{name}_torch_state_after"] = _torch_state_hash()
        meta[f"score_{name}_n_candidates"] = len(base)

    # (b) candidate order of _find_merge_candidates_fast (enhanced...:992-1013): refresh, then a cache pop
    for name in ("freq_hier", "freq_only"):
        seed_all(42)
        tok = _mk_enh(EM, X, thr, G5_CONFIGS[name], corpus=corpus)
        torch.manual_seed(321)
        first = tok._find_merge_candidates_fast()
        second = tok._find_merge_candidates_fast()
        for tag, lst in (("first", first), ("second", second)):
            out[f"order_{name}_{tag}_i"] = np.array([c.token_i for c in lst], np.int32)
            out[f"order_{name}_{tag}_j"] = np.array([c.token_j for c in lst], np.int32)
            out[f"order_{name}_{tag}_score"] = np.array([c.combined_score for c in lst], np.float64)
        meta[f"order_{name}_cache_len"] = len(tok.cache.candidates)
        meta[f"order_{name}_torch_state_after"] = _torch_state_hash()

    # (c) merge sequences with the curvature step never firing (enhanced...:1015-1209)
    seqs = {
        "freq_hier": dict(steps=12, log_every=5, phase_transition_steps={2: 4, 3: 8}),
        "freq_comp_adapt": dict(steps=8, log_every=3),
        "freq_only": dict(steps=12, log_every=5),
    }
    for name, kw in seqs.items():
        seed_all(42)
        tok = _mk_enh(EM, X, thr, G5_CONFIGS[name], corpus=corpus)
        pairs = _record(tok)
        torch.manual_seed(777)
        tok.optimize_merges(**kw)
        m = tok.current_vocab_size
        out[f"seq_{name}_pairs"] = np.array(pairs, np.int32).reshape(-1, 2)
        out[f"seq_{name}_rows"] = tok.embeddings.data[n:m].numpy().copy()
        meta[f"seq_{name}"] = {
            "kwargs": {k: ({str(a): b for a, b in v.items()} if isinstance(v, dict) else v) for k, v in kw.items()},
            "threshold": tok.merge_threshold,
            "phase": tok.current_phase,
            "curvature": float(tok.get_curvature().item() if hasattr(tok.get_curvature(), "item") else tok.get_curvature()),
            "training_stats": {str(k): v for k, v in getattr(tok, "training_stats", {}).items()},
            "random_state_after": _state_hash(),
            "torch_state_after": _torch_state_hash(),
            "merges": [list(t) for t in tok.merge_history],
            "cache_len": len(tok.cache.candidates),
            "merge_pairs": [list(p) for p in getattr(tok, "merge_pairs", [])],
        }
        # (d) what save() writes for this trained tokenizer (enhanced...:1211-1298): the JSON documents
        with tempfile.TemporaryDirectory() as td:
            tok.save(td)
            files = sorted(os.listdir(td))
            meta[f"save_{name}_files"] = files
            for fn in files:
                if fn.endswith(".json"):
                    meta[f"save_{name}_{fn}"] = json.load(open(os.path.join(td, fn)))
            emb = torch.load(os.path.join(td, "embeddings.pt"), weights_only=True)
            meta[f"save_{name}_embeddings_shape"] = list(emb.shape)

    # (e) the adaptive-curvature step.  As shipped it raises at loss.backward() (SURVEY F8:
    # distance() re-wraps c with torch.tensor(), which detaches the Parameter).  Under a ONE-LINE patch of
    # distance() that keeps c attached, the reference's own loss functions, Adam step, clamp and
    # re-projection run; their outputs pin the analytic-gradient form of the build.  This is NOT the
    # behaviour of the reference as shipped (that is an exception) -- recorded as "patched".
    if lor:
        def distance_attached(x, y, c=1.0):
            xy = -L.minkowski_dot(x, y)
            xy = torch.clamp(xy, min=1.0 + 1e-8)
            ct = c if isinstance(c, torch.Tensor) else torch.tensor(c, device=x.device, dtype=x.dtype)
            return torch.acosh(xy) / torch.sqrt(ct)
        orig = EM.distance
        EM.distance = distance_attached
        try:
            seed_all(42)
            flags = dict(use_frequency_aware=False, use_hierarchical=False, use_adaptive_curvature=True,
                         use_compression_aware=False, optimize_curvature_freq=10 ** 6, curvature_lr=0.01)
            tok = _mk_enh(EM, X, thr, flags, max_vocab_size=n + 40, curvature=1.0)
            first = FM.FastHyperbolicTokenizer._find_merge_candidates_fast(tok)
            for q in (0, 5, 10, 20, 30, 45):            # distinct pairs: no duplicated / NaN rows
                tok._merge_tokens(first[q].token_i, first[q].token_j)
            out["curv_rows_before"] = tok.embeddings.data.numpy().copy()
            meta["curv_merge_pairs"] = [list(p) for p in tok.merge_pairs]
            meta["curv_n"] = tok.current_vocab_size
            steps_rec = []
            torch.manual_seed(99)
            for _ in range(3):
                emb = tok.embeddings.detach()
                st0 = torch.get_rng_state()
                h = tok._compute_hierarchy_preservation_loss(emb)
                dl = tok._compute_distortion_loss(emb)
                torch.set_rng_state(st0)
                tok._optimize_curvature(emb)
                tok._project_embeddings()
                steps_rec.append({"hierarchy_loss": float(h.item()), "distortion_loss": float(dl.item()),
                                  "curvature_after": float(tok.curvature.item())})
            meta["curv_steps"] = steps_rec
            meta["curv_torch_state_after"] = _torch_state_hash()
            out["curv_rows_after"] = tok.embeddings.data.numpy().copy()
        finally:
            EM.distance = orig

    np.savez_compressed(os.path.join(HERE, f"g5_enhanced_{mode}.npz"), **out)
    with open(os.path.join(HERE, f"g5_enhanced_{mode}.json"), "w") as f:
        json.dump(meta, f, indent=1, ensure_ascii=False)


# ---------------------------------------------------------------------------------------------
# G6: tokenize / encode / decode (hyperbolic_merge.py:414-471) on lines of the reference's own
# data/processed/wikitext103/test.txt, with a tokenizer trained by the reference's CLI function
# ---------------------------------------------------------------------------------------------
def g6_tokenize(mode: str) -> None:
    import scripts.train_hyperbolic_tokenizer as T
    vocab_path = os.path.join(REF, "data/processed/wiki/vocab_initial.txt")
    with tempfile.TemporaryDirectory() as td:
        T.train_tokenizer(vocab_path=vocab_path, output_dir=td, embedding_dim=5, curvature=1.0, merge_threshold=0.1,
                          merge_steps=60, log_every=20, target_vocab_size=500, seed=42, use_fast_tokenizer=True, no_faiss=True)
        tok = FM.FastHyperbolicTokenizer.load(td, device=torch.device("cpu")) if hasattr(FM.FastHyperbolicTokenizer, "load") else None
        vocab = json.load(open(os.path.join(td, "vocab.json")))
        merges = json.load(open(os.path.join(td, "merges.json")))
    # a few hand-made rules on top, so that multi-level merges occur in ordinary text (th, he, the, in, ing, ...)
    extra = [("t", "h", "th"), ("h", "e", "he"), ("th", "e", "the"), ("i", "n", "in"), ("in", "g", "ing"), ("e", "r", "er"),
             ("a", "n", "an"), ("an", "d", "and"), ("o", "n", "on"), ("r", "e", "re"), ("e", "d", "ed"), (" ", "t", " t"),
             (" t", "he", " the"), ("s", " ", "s "), ("e", "s", "es"), ("t", "i", "ti"), ("ti", "on", "tion")]
    X = lorentz_table(len(vocab), 5, seed=1, scale=0.05)
    tok = HM.HyperbolicTokenizer(vocab=list(vocab), embeddings=torch.nn.Parameter(X), device=torch.device("cpu"),
                                 max_vocab_size=len(vocab) + 8, use_approximate_search=False)
    tok.merge_history = [tuple(m) for m in merges] + extra
    for (_a, _b, ab) in extra:
        if ab not in tok.token2idx:
            tok.vocab.append(ab)
            tok.token2idx[ab] = len(tok.vocab) - 1
    lines = []
    with open(os.path.join(REF, "data/processed/wikitext103/test.txt"), encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if len(line) > 40:
                lines.append(line[:240])
            if len(lines) == 24:
                break
    lines += ["", "a", "the", "ththththe", "in ing inging", "zzz \u4e2d\u6587 the", "tion" * 8]
    out = {
        "vocab": tok.vocab, "merges": [list(m) for m in tok.merge_history], "lines": lines,
        "tokens": [tok.tokenize(t) for t in lines],
        "ids": [tok.encode(t) for t in lines],
    }
    out["decoded"] = [tok.decode(i) for i in out["ids"]]
    with open(os.path.join(HERE, f"g6_tokenize_{mode}.json"), "w") as f:
        json.dump(out, f, ensure_ascii=False)


# ---------------------------------------------------------------------------------------------
# G7: directories written by the reference's own save() (hyperbolic_merge.py:473-499; enhanced...:1211-1298),
# kept as files, plus what the reference's own load() makes of them -- the build's load() must read them
# ---------------------------------------------------------------------------------------------
def g7_saved_dirs(mode: str) -> None:
    import shutil
    EM = _enhanced_module()
    root = os.path.join(HERE, f"g7_saved_{mode}")
    shutil.rmtree(root, ignore_errors=True)
    os.makedirs(root)
    n, d = 40, 5
    X = lorentz_table(n, d, seed=7, scale=0.05)
    thr = 0.1 if mode == "reference" else 0.35
    summary = {}

    def describe(tok):
        k = tok.current_vocab_size
        return {"vocab": list(tok.vocab), "merge_history": [list(m) for m in tok.merge_history], "current_vocab_size": int(k),
                "max_vocab_size": int(tok.max_vocab_size), "curvature": float(tok.curvature), "merge_threshold": float(tok.merge_threshold),
                "embedding_bits": tok.embeddings.data[:k].detach().numpy().view(np.uint32).tolist(),
                "tokenize": [tok.tokenize(t) for t in ("".join(tok.vocab[:6]), tok.vocab[-1] + tok.vocab[0], "")],
                "encode": [tok.encode(t) for t in ("".join(tok.vocab[:6]), tok.vocab[-1] + "z")]}

    seed_all(42)
    std = _mk_tok(HM.HyperbolicTokenizer, X, thr, max_vocab_size=n + 24)
    std.optimize_merges(steps=6, log_every=10 ** 9)
    std.save(os.path.join(root, "std"))
    summary["std"] = describe(HM.HyperbolicTokenizer.load(os.path.join(root, "std"), device=torch.device("cpu")))
    seed_all(42)
    fast = _mk_tok(FM.FastHyperbolicTokenizer, X, thr, max_vocab_size=n + 24)
    fast.optimize_merges(steps=8, log_every=10 ** 9, adaptive_threshold=False)
    fast.save(os.path.join(root, "fast"))
    summary["fast"] = describe(FM.FastHyperbolicTokenizer.load(os.path.join(root, "fast"), device=torch.device("cpu")))
    seed_all(42)
    enh = _mk_enh(EM, X, thr, dict(use_frequency_aware=True, use_hierarchical=False, use_adaptive_curvature=False,
                                   use_compression_aware=False), max_vocab_size=n + 24)
    enh.optimize_merges(steps=5, log_every=10 ** 9, adaptive_threshold=False)
    enh.save(os.path.join(root, "enhanced"))
    # The reference's enhanced save() writes the WHOLE pre-allocated table (max_vocab_size rows) and its load() hands
    # that to a constructor that expects len(vocab) rows: it raises on its own file whenever the table is not full.
    # The expectation is therefore taken from the object that was saved.
    try:
        EM.EnhancedFastHyperbolicTokenizer.load(os.path.join(root, "enhanced"), device=torch.device("cpu"))
        summary["enhanced_reference_load"] = "ok"
    except Exception as exc:
        summary["enhanced_reference_load"] = f"raises {type(exc).__name__}"
    summary["enhanced"] = describe(enh)
    summary["enhanced"]["pair_frequencies"] = sorted([[a, b, int(c)] for (a, b), c in enh.pair_frequencies.items()])[:50]
    summary["enhanced"]["n_pair_frequencies"] = len(enh.pair_frequencies)
    summary["files"] = {k: sorted(os.listdir(os.path.join(root, k))) for k in ("std", "fast", "enhanced")}
    with open(os.path.join(root, "expected.json"), "w") as f:
        json.dump(summary, f, ensure_ascii=False)


def main() -> None:
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    only = sys.argv[2] if len(sys.argv) > 2 else "all"      # e.g. "g5": regenerate one family
    modes = ("reference", "lorentz") if which == "all" else (which,)
    for mode in modes:
        set_mode(mode)
        if only == "g5":
            g5_enhanced(mode)
            print(f"[{mode}] g5 done", flush=True)
            continue
        if only == "g6":
            g6_tokenize(mode)
            print(f"[{mode}] g6 done", flush=True)
            continue
        if only == "g7":
            g7_saved_dirs(mode)
            print(f"[{mode}] g7 done", flush=True)
            continue
        g1_primitives(mode)
        print(f"[{mode}] g1 done", flush=True)
        g2_candidates(mode)
        print(f"[{mode}] g2 done", flush=True)
        g3_sequences(mode)
        print(f"[{mode}] g3 done", flush=True)
        g_cli(mode)
        print(f"[{mode}] cli done", flush=True)
        g5_enhanced(mode)
        print(f"[{mode}] g5 done", flush=True)
        g6_tokenize(mode)
        print(f"[{mode}] g6 done", flush=True)
    set_mode("reference")


if __name__ == "__main__":
    main()
