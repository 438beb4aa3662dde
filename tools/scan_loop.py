"""Runs a few full argmin scans at the bench size (profiling target)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from hyptokenizer_amd.engine import MergeEngine
from hyptokenizer_amd.synthetic import lorentz_table
V, d = 50000, 100
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 5
X = lorentz_table(V, d, seed=42, scale=0.05)
table = torch.zeros((V + 64, d + 1), device="cuda"); table[:V] = X.cuda()
eng = MergeEngine(V + 64, d + 1, "lorentz"); eng.set_table(table, V)
for _ in range(n_iter):
    r = eng.argmin(1.0, 0.5)
print(r, eng.scan_stats())
