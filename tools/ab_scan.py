"""Interleaved A/B timing of pair-scan kernel variants in ONE process (cdna guide rule 24).
usage: python tools/ab_scan.py build_variants/libhm_a.so build_variants/libhm_b.so ..."""
import os, sys, statistics, numpy as np, torch
sys.path.insert(0, ".")
from hyptokenizer_amd import _lib
from hyptokenizer_amd.engine import MergeEngine
from hyptokenizer_amd.synthetic import lorentz_table
V, d = int(os.environ.get('AB_V', 50000)), int(os.environ.get('AB_D', 100))
X = lorentz_table(V, d, seed=42, scale=0.05)
table = torch.zeros((V + 64, d + 1), device="cuda"); table[:V] = X.cuda()
engines = {}
for path in sys.argv[1:]:
    _lib._lib = None
    _lib.LIB_PATH = path
    e = MergeEngine(V + 64, d + 1, "lorentz"); e.set_table(table, V)
    engines[path.split("libhm_")[-1].replace(".so", "")] = e
res = {k: [] for k in engines}
ans = {}
for rnd in range(12):
    for k, e in engines.items():
        r = e.argmin(1.0, 0.5)
        ans.setdefault(k, r)
        if rnd >= 2:
            res[k].append(e.scan_stats()["scan_ms"])
flops = V * (V - 1) * (d + 1)
for k in engines:
    med, mn = statistics.median(res[k]), min(res[k])
    print(f"{k:10s} median {med:.4f} ms  min {mn:.4f} ms  -> {flops/med/1e9:.1f} TF (median)  answer {ans[k]}")
