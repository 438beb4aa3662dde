"""Interleaved A/B timing of the pair scan under different work-decomposition knobs of ONE library build
(`hm_debug_set_default_knob`), in one process: python tools/ab_knobs.py "name:knob=value,knob=value" ...
e.g.  python tools/ab_knobs.py "s256:big_rows=1000000" "s512:big_rows=2" "k112:kc_even=1"
AB_V / AB_D select the table, AB_MODE = argmin | topk."""
import os, sys, statistics, torch
sys.path.insert(0, ".")
from hyptokenizer_amd import _lib
from hyptokenizer_amd.engine import MergeEngine
from hyptokenizer_amd.synthetic import lorentz_table

V, d = int(os.environ.get("AB_V", 50000)), int(os.environ.get("AB_D", 100))
mode = os.environ.get("AB_MODE", "argmin")
L = _lib.load()
X = lorentz_table(V, d, seed=42, scale=0.05)
table = torch.zeros((V + 64, d + 1), device="cuda"); table[:V] = X.cuda()
engines = {}
for spec in sys.argv[1:]:
    name, _, kv = spec.partition(":")
    L.hm_debug_set_default_knob(None, 0.0, 1)
    for item in filter(None, kv.split(",")):
        k, v = item.split("=")
        _lib.check(L.hm_debug_set_default_knob(k.encode(), float(v), 0))
    e = MergeEngine(V + 64, d + 1, "lorentz"); e.set_table(table, V)
    engines[name] = e
L.hm_debug_set_default_knob(None, 0.0, 1)
res = {k: [] for k in engines}
ans = {}
for rnd in range(14):
    for k, e in engines.items():
        if mode == "argmin":
            r = e.argmin(1.0, 0.5)
        else:
            dd, ii, jj, cnt = e.topk(1.0, 0.5, 10000, count=(mode == "topk_count"))
            r = (float(dd[0]), int(ii[0]), int(jj[0]), int(ii.sum()), int(jj.sum()), cnt)
        ans.setdefault(k, r)
        if rnd >= 2:
            res[k].append(e.scan_stats()["scan_ms"])
flops = V * (V - 1) * (d + 1)
for k in engines:
    med, mn = statistics.median(res[k]), min(res[k])
    print(f"{k:12s} V={V} median {med:.4f} ms  min {mn:.4f} ms  -> {flops/med/1e9:.1f} TF = {flops/med/1e9/2500:.3f} of 2.5 PF  answer {ans[k]}", flush=True)
