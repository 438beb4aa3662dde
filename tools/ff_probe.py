#!/usr/bin/env python
"""where does the fast loop spend its host time on the GPU box? (section timers)"""
import os, sys, time
os.environ.setdefault("TQDM_DISABLE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table
import hyptokenizer_amd.tokenizer.fast_hyperbolic_merge as M
V = 50000
X = lorentz_table(V, 100, seed=42, scale=0.05)
tok = M.FastHyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X), merge_threshold=0.5, device=torch.device("cuda"),
                                max_vocab_size=V + 4500, sign_convention="lorentz")
tok.optimize_merges(steps=202, log_every=10 ** 9, adaptive_threshold=False)
torch.cuda.synchronize()
T = {"ff": 0.0, "key": 0.0, "find": 0.0, "plan": 0.0}
C = M.FastHyperbolicTokenizer
o = {"ff": C._fast_forward, "key": C._table_key, "find": C._find_merge_candidates_fast, "plan": C._plan_merges}
def w(name, fn):
    def f(self, *a, **k):
        t = time.perf_counter(); r = fn(self, *a, **k); T[name] += time.perf_counter() - t; return r
    return f
C._fast_forward = w("ff", o["ff"]); C._table_key = w("key", o["key"])
C._find_merge_candidates_fast = w("find", o["find"]); C._plan_merges = w("plan", o["plan"])
t0 = time.perf_counter()
tok.optimize_merges(steps=2020, log_every=10 ** 9, adaptive_threshold=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("loop", t1 - t0, "sync after", t2 - t1, T)
