"""HyperbolicTokenizer.optimize_merges at a given size: merges/s on the GPU and (optionally) the oracle's search
rate on the host cores beside it.  usage: python tools/std_loop.py V d STEPS [cpu]"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table
from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
V, d, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
X = lorentz_table(V, d, seed=42, scale=0.05)
tok = HyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X), merge_threshold=0.5, device=torch.device("cuda"),
                          max_vocab_size=V + steps + 80, sign_convention="lorentz")
tok.optimize_merges(steps=10, log_every=10 ** 9)
torch.cuda.synchronize(); t0 = time.perf_counter()
tok.optimize_merges(steps=steps, log_every=10 ** 9)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
tot = tok._get_engine().scan_totals()
print(f"GPU: V={V} d={d}: {steps / dt:.1f} merges/s, {1e3 * dt / steps:.3f} ms/step, scan {tot['scan_ms'] / tot['launches']:.3f} ms")
if len(sys.argv) > 4:
    sys.path.insert(0, "oracle")
    from oracle import hm_oracle as O
    Xn = X.numpy()
    O.pairwise_topk(Xn, V, 1.0, 0.5, 1, 1, fast=True)
    t0 = time.perf_counter()
    for _ in range(3):
        r = O.pairwise_topk(Xn, V, 1.0, 0.5, 1, 1, fast=True)
    dtc = (time.perf_counter() - t0) / 3
    print(f"CPU oracle ({O.num_threads()} threads): {dtc:.3f} s per search = {1 / dtc:.2f} merges/s; GPU/CPU = {steps / dt * dtc:.0f}x; "
          f"same pair: {(int(r[1][0]), int(r[2][0])) == tuple(tok._get_engine().argmin(1.0, 0.5)[1:]) if False else 'n/a'}")
