"""Wall time of a search that ends in the prefilter-free exact path (hm_exact.hip): embeddings of scale 1e-3."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from hyptokenizer_amd.engine import MergeEngine
from hyptokenizer_amd.synthetic import lorentz_table
for n, d in ((25000, 100), (50000, 100), (100000, 100)):
    X = lorentz_table(n, d, seed=3, scale=0.001)
    table = torch.zeros((n + 4, d + 1), device="cuda"); table[:n] = X.cuda()
    eng = MergeEngine(n + 4, d + 1, "lorentz"); eng.set_table(table, n)
    s = eng.pair_distance(np.arange(0, 2000), np.arange(2000, 4000), 1.0)
    thr = float(np.percentile(s, 20))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dd, ii, jj, cnt = eng.topk(1.0, thr, 100)
        t1 = time.perf_counter()
        a = eng.argmin(1.0, thr)
        t2 = time.perf_counter()
        print(f"n={n} d={d} rep {rep}: top-100 {1e3*(t1-t0):.1f} ms (count {cnt}), argmin {1e3*(t2-t1):.1f} ms -> {a}", flush=True)
