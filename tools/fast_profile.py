#!/usr/bin/env python
"""cProfile of the fast loop at V = 50 000, d = 100 (where does the host time of a step go?)"""
import cProfile, os, pstats, sys, time
os.environ.setdefault("TQDM_DISABLE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table
from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
V = 50000
X = lorentz_table(V, 100, seed=42, scale=0.05)
tok = FastHyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X), merge_threshold=0.5, device=torch.device("cuda"),
                              max_vocab_size=V + 4500, sign_convention="lorentz")
tok.optimize_merges(steps=202, log_every=10 ** 9, adaptive_threshold=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
tok.optimize_merges(steps=2020, log_every=10 ** 9, adaptive_threshold=False)
torch.cuda.synchronize()
print("plain:", 2020 / (time.perf_counter() - t0), "merges/s")
pr = cProfile.Profile()
pr.enable()
tok.optimize_merges(steps=2020, log_every=10 ** 9, adaptive_threshold=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
