"""The bench's fast-path leg alone (FastHyperbolicTokenizer, V = 50 000, d = 100, cache 10 000): for rocprofv3 traces
and host-side timing of one refresh cycle (tools/fast_run.py [steps])."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table  # noqa: E402
from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2020
V, D = 50000, 100
dev = torch.device("cuda:0")
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 42
X = lorentz_table(V, D, seed=seed, scale=0.05)
tok = FastHyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X), curvature=1.0, merge_threshold=0.5, device=dev,
                              max_vocab_size=V + steps + 64, sign_convention="lorentz")
tok._get_engine()
tok.optimize_merges(steps=202, log_every=10 ** 9, adaptive_threshold=False)
torch.cuda.synchronize()
import cProfile
import pstats
pr = cProfile.Profile() if os.environ.get("FAST_PROFILE") else None
t0 = time.perf_counter()
if pr:
    pr.enable()
tok.optimize_merges(steps=steps - 202, log_every=10 ** 9, adaptive_threshold=False)
if pr:
    pr.disable()
torch.cuda.synchronize()
el = time.perf_counter() - t0
done = len(tok.merge_history) - 202
print(f"{done} merges in {el * 1e3:.2f} ms = {done / el:.0f} merges/s, {el / done * 1e6:.2f} us/merge")
if pr:
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
