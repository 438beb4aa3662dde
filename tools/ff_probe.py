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

# ---- second pass: the fast-forward body with time marks between its statements
import numpy as np
marks = {}
def mark(name, t):
    marks[name] = marks.get(name, 0.0) + (time.perf_counter() - t)
    return time.perf_counter()
def ff2(self, step, steps, log_every, adaptive_threshold):
    t = time.perf_counter()
    cache, plan = self.cache, self._plan
    if plan is None or cache._list or cache._arr is None:
        return 0
    d_arr, i_arr, j_arr = cache._arr
    pos, n_arr, p = cache._pos, len(d_arr), plan.pos
    n = self.current_vocab_size
    if pos >= n_arr or not plan.matches(int(i_arr[pos]), int(j_arr[pos]), n):
        return 0
    k = min(len(plan.i) - p, (n_arr - pos + 99) // 100, steps - step)
    if step % log_every == 0:
        return 0
    k = min(k, log_every - 1 - step % log_every)
    if k <= 0:
        return 0
    t = mark("prologue", t)
    A, B = plan.i[p:p + k], plan.j[p:p + k]
    vocab = self.vocab
    lefts = [vocab[a] for a in A]
    rights = [vocab[b] for b in B]
    merged = [x + y for x, y in zip(lefts, rights)]
    t = mark("strings", t)
    vocab.extend(merged)
    t = mark("vocab.extend", t)
    self.token2idx.update(zip(merged, range(n, n + k)))
    t = mark("dict.update", t)
    self.merge_history.extend(zip(lefts, rights, merged))
    t = mark("history.extend", t)
    hi = min(pos + 100 * k, n_arr)
    cache._served.append((i_arr, j_arr, pos, hi))
    cache._hits += hi - pos
    if hi >= n_arr:
        cache._arr, cache._pos = None, 0
    else:
        cache._pos = hi
    plan.pos = p + k
    if plan.pos >= len(plan.i):
        self._plan = None
    self.current_vocab_size = n + k
    self.merges_since_rebuild += k
    self._engine_key = o["key"](self)
    t = mark("epilogue", t)
    return k
C._fast_forward = ff2
t0 = time.perf_counter()
tok.optimize_merges(steps=2020, log_every=10 ** 9, adaptive_threshold=False)
print("loop2", time.perf_counter() - t0, {k: round(v * 1e3, 3) for k, v in marks.items()}, "ms")
import gc
print("gc counts", gc.get_count(), gc.get_threshold(), "vocab type", type(tok.vocab))
