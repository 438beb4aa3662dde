#!/usr/bin/env python
"""Time the pair scan of ONE build of libhypmerge.so (HYPMERGE_LIB=<path> selects it): event-timed launch
duration of the argmin scan at V = 50 000 and 100 000 (d = 100, bf16 and fp32 prefilter), the top-k refresh,
and the merges/s of the standard / incremental / fast loops.  Prints one JSON line.  tools/scan_variants.sh runs
it once per variant build in a fresh process."""
import json
import os
import sys
import time

os.environ.setdefault("TQDM_DISABLE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from hyptokenizer_amd.engine import MergeEngine  # noqa: E402
from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table  # noqa: E402

THR, C = 0.5, 1.0
out = {"lib": os.environ.get("HYPMERGE_LIB", "default"), "tag": os.environ.get("HM_VARIANT_TAG", "")}
quick = "--quick" in sys.argv
dev = torch.device("cuda", 0)
for V, forms in ((50000, ("bf16", "f32")), (100000, ("bf16",))):
    X = lorentz_table(V, 100, seed=42, scale=0.05)
    table = torch.zeros((V + 600, 101), device=dev)
    table[:V] = X.to(dev)
    for form in forms:
        if quick and form == "f32":
            continue
        eng = MergeEngine(V + 600, 101, "lorentz", dev, prefilter=form)
        eng.set_table(table, V)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.08:
            r = eng.argmin(C, THR)
        eng.scan_totals(reset=True)
        reps = 30 if form == "bf16" else 8
        for _ in range(reps):
            r = eng.argmin(C, THR)
        tt = eng.scan_totals()
        ms = tt["scan_ms"] / tt["launches"]
        fl = 2.0 * 101 * tt["pairs"] / tt["launches"]
        out[f"scan_ms_{V}_{form}"] = round(ms, 4)
        out[f"pflops_{V}_{form}"] = round(fl / (ms * 1e-3) / 1e15, 4)
        out[f"emitted_{V}_{form}"] = eng.scan_stats()["emitted"]
        out[f"pair_{V}_{form}"] = list(r) if r else None
        if form == "bf16":
            # top-k refresh (uncounted and counted), second call = predicted cut
            for cnt in (False, True):
                eng.topk(C, THR, 10000, count=cnt)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    eng.topk(C, THR, 10000, count=cnt)
                torch.cuda.synchronize()
                out[f"topk_ms_{V}_{'count' if cnt else 'nocount'}"] = round((time.perf_counter() - t1) / 5 * 1e3, 4)
                out[f"topk_scan_ms_{V}_{'count' if cnt else 'nocount'}"] = round(eng.scan_stats()["scan_ms"], 4)
        del eng
    if V == 50000 and not quick:
        from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
        from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
        vocab = cjk_vocab(V)
        for name, kw, steps in (("std", {}, 200), ("incr", {"incremental": True}, 640)):
            tok = HyperbolicTokenizer(vocab, torch.nn.Parameter(X), merge_threshold=THR, device=dev, max_vocab_size=V + steps + 100,
                                      sign_convention="lorentz", **kw)
            tok.optimize_merges(steps=16, log_every=10 ** 9)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            tok.optimize_merges(steps=steps, log_every=10 ** 9)
            torch.cuda.synchronize()
            out[f"{name}_merges_per_s"] = round(steps / (time.perf_counter() - t1), 1)
            del tok
        ftok = FastHyperbolicTokenizer(vocab, torch.nn.Parameter(X), merge_threshold=THR, device=dev, max_vocab_size=V + 2400,
                                       sign_convention="lorentz")
        ftok.optimize_merges(steps=202, log_every=10 ** 9, adaptive_threshold=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ftok.optimize_merges(steps=2020, log_every=10 ** 9, adaptive_threshold=False)
        torch.cuda.synchronize()
        out["fast_merges_per_s"] = round(2020 / (time.perf_counter() - t1), 1)
        del ftok
    del table
print(json.dumps(out), flush=True)
