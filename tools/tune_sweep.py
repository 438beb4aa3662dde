"""Sweep the work-decomposition knobs (HM_TUNE_*) of the in-tree library; development aid."""
import itertools, os, statistics, sys, torch
sys.path.insert(0, ".")
from hyptokenizer_amd.engine import MergeEngine
from hyptokenizer_amd.synthetic import lorentz_table
V, d = int(os.environ.get('AB_V', 50000)), int(os.environ.get('AB_D', 100))
X = lorentz_table(V, d, seed=42, scale=0.05)
table = torch.zeros((V + 64, d + 1), device="cuda"); table[:V] = X.cuda()
chunks = [int(x) for x in os.environ.get("SWEEP_CHUNK", "48,64,96,128").split(",")]
tails = [float(x) for x in os.environ.get("SWEEP_TAIL", "0.15").split(",")]
divs = [int(x) for x in os.environ.get("SWEEP_DIV", "4").split(",")]
engines = {}
for c, t, dv in itertools.product(chunks, tails, divs):
    os.environ.update(HM_TUNE_CHUNK=str(c), HM_TUNE_TAIL=str(t), HM_TUNE_TAIL_DIV=str(dv))
    e = MergeEngine(V + 64, d + 1, "lorentz"); e.set_table(table, V)
    engines[(c, t, dv)] = e
res = {k: [] for k in engines}
for rnd in range(10):
    for k, e in engines.items():
        e.argmin(1.0, 0.5)
        if rnd >= 2:
            res[k].append(e.scan_stats()["scan_ms"])
flops = V * (V - 1) * (d + 1)
for k in sorted(res, key=lambda k: statistics.median(res[k])):
    med = statistics.median(res[k])
    print(f"chunk {k[0]:4d} tail {k[1]:.2f} div {k[2]}: median {med:.4f} ms min {min(res[k]):.4f} -> {flops/med/1e9:.0f} TF")
