"""Randomised class-level sweep (development aid): python tools/fuzz_tokenizers.py [CASES] [SEED]
HyperbolicTokenizer (full search and incremental), FastHyperbolicTokenizer and EnhancedFastHyperbolicTokenizer on the HIP
engine against THE SAME classes driven by the oracle's engine double (tests/helpers.py OracleEngine): merge histories,
appended rows bit for bit, thresholds, generator states.  Random sizes / widths / scales / thresholds / signs / cache sizes /
rebuild frequencies / step counts (several refreshes per run)."""
import os
import random
import sys

import numpy as np
import torch

os.environ.setdefault("TQDM_DISABLE", "1")
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
sys.path.insert(0, "oracle")
from helpers import OracleEngine  # noqa: E402
from hyptokenizer_amd.engine import MergeEngine  # noqa: E402
from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table  # noqa: E402
from hyptokenizer_amd.tokenizer.enhanced_fast_hyperbolic_merge import EnhancedFastHyperbolicTokenizer  # noqa: E402
from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer  # noqa: E402
from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer  # noqa: E402
from oracle import hm_oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for t in range(cases):
    n = int(rng.choice([rng.integers(20, 200), rng.integers(200, 2500)]))
    d = int(rng.choice([3, 8, 16, 32, 50, 100]))
    scale = float(rng.choice([0.01, 0.05, 0.2]))
    mode = str(rng.choice(["lorentz", "lorentz", "lorentz", "reference"]))
    kind = str(rng.choice(["std", "incr", "fast", "fast", "enhanced"]))
    steps = int(rng.choice([3, 17, 60, 150])) if kind != "enhanced" else int(rng.choice([3, 12, 30]))
    seed = int(rng.integers(1 << 30))
    X = lorentz_table(n, d, seed=seed, scale=scale)
    sm = 1 if mode == "lorentz" else 0
    m = min(n, 300)
    D = O.batch_distance(X.numpy()[:m], X.numpy()[:m], 1.0, sm)[np.triu_indices(m, 1)]
    D = D[np.isfinite(D)]
    thr = float(np.quantile(D, rng.choice([0.02, 0.2, 0.7]))) if len(D) else 0.1
    if mode == "reference":
        thr = 0.1
    cache = int(rng.choice([50, 1000, 10000]))
    rebuild = int(rng.choice([7, 100]))
    cfreq = int(rng.choice([2, 5, 1000]))
    rows = n + steps + 8
    vocab = cjk_vocab(n)
    outs = []
    try:
        for which in ("hip", "oracle"):
            random.seed(99)
            torch.manual_seed(99)
            cpu_only = os.environ.get("FUZZ_CPU_ONLY") == "1"          # (dry run without a GPU: the oracle's two search forms against each other)
            dev = torch.device("cuda" if which == "hip" and not cpu_only else "cpu")
            if which == "hip" and not cpu_only:
                eng = MergeEngine(rows, d + 1, mode, dev)
            else:
                eng = OracleEngine(rows, d + 1, mode, fast=(which == "hip") or n > 300)       # (the OpenMP form for the larger tables: checked against the plain form in tests/)
            emb = torch.nn.Parameter(X.clone())
            if kind in ("std", "incr"):
                tok = HyperbolicTokenizer(vocab, emb, merge_threshold=thr, device=dev, max_vocab_size=rows, sign_convention=mode,
                                          engine=eng, incremental=(kind == "incr"))
                tok.optimize_merges(steps=steps, log_every=10 ** 9)
            elif kind == "fast":
                tok = FastHyperbolicTokenizer(vocab, emb, merge_threshold=thr, device=dev, max_vocab_size=rows, cache_size=cache,
                                              rebuild_frequency=rebuild, sign_convention=mode, engine=eng)
                tok.optimize_merges(steps=steps, log_every=10 ** 9, adaptive_threshold=bool(t & 1))
            else:
                tok = EnhancedFastHyperbolicTokenizer(vocab, emb, merge_threshold=thr, device=dev, max_vocab_size=rows, cache_size=cache,
                                                      rebuild_frequency=rebuild, sign_convention=mode, engine=eng, use_frequency_aware=True,
                                                      use_hierarchical=bool(t & 2), use_adaptive_curvature=(mode == "lorentz"),
                                                      use_compression_aware=False, optimize_curvature_freq=cfreq)
                tok.pair_frequencies = {(vocab[a], vocab[a + 1]): a % 7 + 1 for a in range(0, n - 1, 3)}
                tok.optimize_merges(steps=steps, log_every=10 ** 9, adaptive_threshold=bool(t & 1))
            outs.append(dict(merges=[list(mm) for mm in tok.merge_history],
                             rows=tok.embeddings.data[n:tok.current_vocab_size].cpu().numpy().view(np.uint32).copy(),
                             thr=float(tok.merge_threshold), curv=repr(float(torch.as_tensor(tok.get_curvature()).detach())) if hasattr(tok, "get_curvature") else "",
                             states=(repr(random.getstate()), torch.get_rng_state().numpy().tobytes())))
        a, b = outs
        fa, fb = a["rows"].view(np.float32), b["rows"].view(np.float32)
        nan_a, nan_b = np.isnan(fa), np.isnan(fb)              # (NaN payload bits are not compared: they differ between x86 and the GPU)
        rows_ok = fa.shape == fb.shape and np.array_equal(nan_a, nan_b) and np.array_equal(a["rows"][~nan_a], b["rows"][~nan_b])
        ok = (a["merges"] == b["merges"] and rows_ok and a["thr"] == b["thr"] and a["curv"] == b["curv"] and a["states"] == b["states"])
        if not ok:
            bad += 1
            first = next((q for q, (x, y) in enumerate(zip(a["merges"], b["merges"])) if x != y), None)
            print("MISMATCH", dict(case=t, n=n, d=d, scale=scale, mode=mode, kind=kind, steps=steps, seed=seed, thr=thr, cache=cache, rebuild=rebuild, cfreq=cfreq),
                  "merges", len(a["merges"]), len(b["merges"]), "first differing step", first, "rows ok", rows_ok, "states equal", a["states"] == b["states"], "thr", a["thr"], b["thr"], "curv", a["curv"], b["curv"], flush=True)
    except Exception as ex:
        bad += 1
        print("ERROR", dict(case=t, n=n, d=d, scale=scale, mode=mode, kind=kind, steps=steps, seed=seed, thr=thr), repr(ex)[:300], flush=True)
    if t % 10 == 9:
        print(f"{t + 1} cases, {bad} bad", flush=True)
print(f"done: {cases} cases, {bad} bad")
