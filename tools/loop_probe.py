"""Where a 64-step device batch of the standard loop spends its wall time: inside hm_std_merge_steps (enqueue + device +
sync) against the Python bookkeeping around it (tools/loop_probe.py [V] [steps])."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table  # noqa: E402
from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
X = lorentz_table(V, 100, seed=0, scale=0.05)
tok = HyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X), curvature=1.0, merge_threshold=0.5, device=dev,
                          max_vocab_size=V + steps + 200, sign_convention="lorentz")
tok.optimize_merges(steps=70, log_every=10 ** 9)            # warm
eng = tok._get_engine()
inner = [0.0, 0]
orig = eng.std_merge_steps


def timed(*a, **k):
    t0 = time.perf_counter()
    r = orig(*a, **k)
    inner[0] += time.perf_counter() - t0
    inner[1] += 1
    return r


eng.std_merge_steps = timed
torch.cuda.synchronize()
t0 = time.perf_counter()
tok.optimize_merges(steps=steps, log_every=10 ** 9)
torch.cuda.synchronize()
el = time.perf_counter() - t0
tot = eng.scan_totals() if hasattr(eng, "scan_totals") else None
print(f"V={V} steps={steps}: wall {el * 1e3:.2f} ms = {el / steps * 1e6:.1f} us/step; inside hm_std_merge_steps {inner[0] * 1e3:.2f} ms "
      f"({inner[1]} calls, {inner[0] / steps * 1e6:.1f} us/step); Python around it {(el - inner[0]) / steps * 1e6:.1f} us/step; totals {tot}")
