"""Per-library scan statistics (emitted entries, passes) for argmin and top-k; development aid."""
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from hyptokenizer_amd import _lib
from hyptokenizer_amd.engine import MergeEngine
from hyptokenizer_amd.synthetic import lorentz_table
V, d = int(os.environ.get('AB_V', 50000)), int(os.environ.get('AB_D', 100))
X = lorentz_table(V, d, seed=42, scale=0.05)
table = torch.zeros((V + 64, d + 1), device="cuda"); table[:V] = X.cuda()
for path in sys.argv[1:]:
    _lib._lib = None
    _lib.LIB_PATH = path
    e = MergeEngine(V + 64, d + 1, "lorentz"); e.set_table(table, V)
    for it in range(4):
        r = e.argmin(1.0, 0.5); st = e.scan_stats()
        print(path.split("libhm_")[-1], "argmin", it, r, {k: st[k] for k in ("scan_ms", "emitted", "passes")})
    for it in range(2):
        dd, ii, jj, cnt = e.topk(1.0, 0.5, 10000); st = e.scan_stats()
        print(path.split("libhm_")[-1], "topk", it, len(dd), cnt, {k: st[k] for k in ("scan_ms", "emitted", "passes")})
