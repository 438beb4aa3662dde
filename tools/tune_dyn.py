"""Sweep HM_TUNE_K1/K2/SPLIT of a HM_PERSIST=2 build; development aid: python tools/tune_dyn.py LIB [REFLIB]"""
import itertools, os, statistics, sys, torch
sys.path.insert(0, ".")
from hyptokenizer_amd import _lib
from hyptokenizer_amd.engine import MergeEngine
from hyptokenizer_amd.synthetic import lorentz_table
V, d = int(os.environ.get('AB_V', 50000)), int(os.environ.get('AB_D', 100))
X = lorentz_table(V, d, seed=42, scale=0.05)
table = torch.zeros((V + 64, d + 1), device="cuda"); table[:V] = X.cuda()
engines = {}
if len(sys.argv) > 2:
    _lib._lib = None; _lib.LIB_PATH = sys.argv[2]
    e = MergeEngine(V + 64, d + 1, "lorentz"); e.set_table(table, V); engines[("ref", 0, 0)] = e
_lib._lib = None; _lib.LIB_PATH = sys.argv[1]
for k1, k2, sp in itertools.product([int(x) for x in os.environ.get("K1", "32,64,128").split(",")],
                                    [int(x) for x in os.environ.get("K2", "8,16,32").split(",")],
                                    [float(x) for x in os.environ.get("SPLIT", "0.6,0.8,0.95").split(",")]):
    os.environ.update(HM_TUNE_K1=str(k1), HM_TUNE_K2=str(k2), HM_TUNE_SPLIT=str(sp))
    e = MergeEngine(V + 64, d + 1, "lorentz"); e.set_table(table, V); engines[(k1, k2, sp)] = e
res = {k: [] for k in engines}
for rnd in range(9):
    for k, e in engines.items():
        e.argmin(1.0, 0.5)
        if rnd >= 2:
            res[k].append(e.scan_stats()["scan_ms"])
flops = V * (V - 1) * (d + 1)
for k in sorted(res, key=lambda k: statistics.median(res[k])):
    med = statistics.median(res[k])
    print(f"k1 {k[0]} k2 {k[1]} split {k[2]}: median {med:.4f} ms min {min(res[k]):.4f} -> {flops/med/1e9:.0f} TF")
