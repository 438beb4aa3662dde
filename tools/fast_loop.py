"""A FastHyperbolicTokenizer run at the bench size (profiling target for the refresh kernels)."""
import sys, torch
sys.path.insert(0, ".")
from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table
from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
V, d = 50000, 100
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 505
X = lorentz_table(V, d, seed=42, scale=0.05)
tok = FastHyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X), merge_threshold=0.5, device=torch.device("cuda"),
                              max_vocab_size=V + steps + 64, sign_convention="lorentz")
tok.optimize_merges(steps=steps, log_every=10 ** 9, adaptive_threshold=False)
torch.cuda.synchronize()
print(len(tok.merge_history), tok.cache.get_stats())
