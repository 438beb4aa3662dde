"""Interleaved A/B of the standard merge loop (full search + merge per step, device-resident batches) under different
work-decomposition knobs and library builds: python tools/ab_loop.py "name[@path/to/lib.so]:knob=value,..." ...
Per configuration one tokenizer; rounds of `AB_STEPS` steps alternate between the configurations; reports the median
wall time per step and the mean event-timed scan of the timed rounds.  AB_V / AB_D select the table."""
import os, sys, statistics, time, torch
os.environ.setdefault("TQDM_DISABLE", "1")
sys.path.insert(0, ".")
from hyptokenizer_amd import _lib
from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table
from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer

V, d = int(os.environ.get("AB_V", 50000)), int(os.environ.get("AB_D", 100))
steps, rounds = int(os.environ.get("AB_STEPS", 64)), int(os.environ.get("AB_ROUNDS", 7))
DEFAULT_LIB = _lib.LIB_PATH
X = lorentz_table(V, d, seed=42, scale=0.05)
vocab = cjk_vocab(V)
toks = {}
for spec in sys.argv[1:]:
    name, _, kv = spec.partition(":")
    name, _, libpath = name.partition("@")
    _lib._lib = None
    _lib.LIB_PATH = libpath or DEFAULT_LIB
    L = _lib.load()
    if hasattr(L, "hm_debug_set_default_knob"):
        L.hm_debug_set_default_knob(None, 0.0, 1)
        for item in filter(None, kv.split(",")):
            k, v = item.split("=")
            _lib.check(L.hm_debug_set_default_knob(k.encode(), float(v), 0))
    t = HyperbolicTokenizer(vocab, torch.nn.Parameter(X.clone()), merge_threshold=0.5, device=torch.device("cuda", 0),
                            max_vocab_size=V + steps * (rounds + 2) + 64, sign_convention="lorentz")
    t.optimize_merges(steps=8, log_every=10 ** 9)
    toks[name] = t
    if hasattr(L, "hm_debug_set_default_knob"): L.hm_debug_set_default_knob(None, 0.0, 1)
for t in toks.values():
    t._engine.debug_time_loops(True)
    t._engine.scan_totals(reset=True)
wall = {k: [] for k in toks}
for rnd in range(rounds):
    for k, t in toks.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t.optimize_merges(steps=steps, log_every=10 ** 9)
        torch.cuda.synchronize()
        if rnd >= 1:
            wall[k].append((time.perf_counter() - t0) / steps * 1e3)
ref = None
for k, t in toks.items():
    tt = t._engine.scan_totals()
    med = statistics.median(wall[k])
    hist = [tuple(m[:2]) for m in t.merge_history]
    same = ref is None or hist == ref
    if ref is None: ref = hist
    fl = V * (V - 1) * (d + 1)
    scan = tt["scan_ms"] / max(tt["launches"], 1)
    print(f"{k:12s} V={V} ms/step median {med:.4f} min {min(wall[k]):.4f} -> {1e3/med:.0f} merges/s; scan mean {scan:.4f} ms over {tt['launches']} "
          f"launches = {fl/scan/1e9/2500:.3f} of 2.5 PF; same merges as first: {same}", flush=True)
