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

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [reference|lorentz|all]
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


def main() -> None:
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    modes = ("reference", "lorentz") if which == "all" else (which,)
    for mode in modes:
        set_mode(mode)
        g1_primitives(mode)
        print(f"[{mode}] g1 done", flush=True)
        g2_candidates(mode)
        print(f"[{mode}] g2 done", flush=True)
        g3_sequences(mode)
        print(f"[{mode}] g3 done", flush=True)
        g_cli(mode)
        print(f"[{mode}] cli done", flush=True)
    set_mode("reference")


if __name__ == "__main__":
    main()
