"""Randomised parity sweep of the device-resident loops (development aid): python tools/fuzz_loops.py [CASES] [SEED]
Random table sizes / widths / scales / thresholds / sign modes / prefilter forms / token lengths / step counts: the
software-pipelined standard loop, the sequential standard loop and the incremental loop must merge the same pairs at the
same distance bits into the same rows; small cases are also replayed on the oracle (search -> first pair -> midpoint)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "oracle")
from hyptokenizer_amd import _lib  # noqa: E402
from hyptokenizer_amd.engine import MergeEngine  # noqa: E402
from hyptokenizer_amd.synthetic import lorentz_table  # noqa: E402
from oracle import hm_oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
L = _lib.load()
bad = 0
for t in range(cases):
    n = int(rng.choice([rng.integers(3, 300), rng.integers(300, 6000), rng.integers(6000, 30000)]))
    d = int(rng.choice([2, 5, 8, 16, 24, 31, 50, 64, 100, 124, 128]))
    scale = float(rng.choice([0.01, 0.05, 0.05, 0.2]))
    mode = str(rng.choice(["lorentz", "lorentz", "lorentz", "reference"]))
    form = str(rng.choice(["f32", "bf16", "bf16"]))
    c = float(rng.choice([1.0, 1.0, 0.3, 4.0]))
    steps = int(rng.choice([1, 2, 3, 7, 20, 41, 64]))
    X = lorentz_table(n, d, seed=int(rng.integers(1 << 30)), scale=scale)
    if rng.random() < 0.2 and n > 10:
        X[rng.integers(n, size=3)] = X[rng.integers(n, size=3)]
    Xn = X.numpy()
    sm = 1 if mode == "lorentz" else 0
    m = min(n, 300)
    D = O.batch_distance(Xn[:m], Xn[:m], c, sm)[np.triu_indices(m, 1)]
    D = D[np.isfinite(D)]
    thr = float(np.quantile(D, rng.choice([0.0005, 0.01, 0.2, 1.0]))) * float(rng.choice([1.0, 1.5])) if len(D) else 0.1
    if mode == "reference":
        thr = 0.1
    lens = rng.integers(1, 6, size=n).astype(np.int32)
    runs = {}
    for variant in ("seq", "pipe", "incr"):
        table = torch.zeros((n + steps + 4, d + 1), device="cuda")
        table[:n] = X.cuda()
        eng = MergeEngine(n + steps + 4, d + 1, mode, prefilter=form)
        _lib.check(L.hm_debug_set_knob(eng._h, b"pipeline", 0.0 if variant == "seq" else 1.0))
        _lib.check(L.hm_debug_set_knob(eng._h, b"pipeline_pairs", 0.0))
        for kv in filter(None, os.environ.get("FUZZ_KNOBS", "").split(",")):      # e.g. FUZZ_KNOBS=dyn_slots=6: the scan's item queue on small tables
            _lib.check(L.hm_debug_set_knob(eng._h, kv.split("=")[0].encode(), float(kv.split("=")[1])))
        eng.set_table(table, n)
        eng.set_token_lengths(lens)
        if variant == "incr":
            best = eng.argmin(c, thr)
            recs, done, _b = eng.incr_merge_steps(c, thr, table, steps, best)
        else:
            recs, done = eng.std_merge_steps(c, thr, table, steps)
        if variant != "incr" and done < steps and recs[done][0] == 2:
            runs[variant] = ("overflow", done)
        else:
            runs[variant] = (done, [tuple(int(v) if not isinstance(v, float) else int(np.float32(v).view(np.uint32)) for v in r[1:]) for r in recs[:done]],
                             table[n:n + done].cpu().numpy().view(np.uint32).copy())
        del eng
    ok = True
    ref = runs["seq"]
    for v in ("pipe", "incr"):
        r = runs[v]
        if ref[0] == "overflow" or r[0] == "overflow":
            # an emission overflow stops a batch at that step (the host path takes it); the steps before it must agree
            continue
        if r[0] != ref[0] or r[1] != ref[1] or not np.array_equal(r[2], ref[2]):
            ok = False
            print(f"MISMATCH case {t}: {v} vs seq n={n} d={d} mode={mode} form={form} c={c} thr={thr} steps={steps} done {r[0]} / {ref[0]}", flush=True)
    if ok and ref[0] != "overflow" and n <= 1500 and mode == "lorentz":
        # oracle replay: search -> first pair -> midpoint with the weight of the token lengths
        T = np.zeros((n + steps + 4, d + 1), np.float32)
        T[:n] = Xn
        ln = list(lens)
        for s in range(ref[0]):
            od, oi, oj, oc = O.pairwise_topk(T, n + s, c, thr, sm, 1)
            if oc == 0 or (int(oi[0]), int(oj[0])) != (ref[1][s][1], ref[1][s][2]) or int(np.float32(od[0]).view(np.uint32)) != ref[1][s][0]:
                ok = False
                print(f"ORACLE MISMATCH case {t} step {s}: n={n} d={d} form={form} thr={thr}", flush=True)
                break
            i, j = int(oi[0]), int(oj[0])
            w = np.float32(np.float64(ln[j]) / np.float64(ln[i] + ln[j]))
            T[n + s] = O.midpoint_batch(T, [i], [j], [w], c, sm)[0]
            ln.append(ln[i] + ln[j])
            if not np.array_equal(T[n + s].view(np.uint32), ref[2][s]):
                ok = False
                print(f"ORACLE ROW MISMATCH case {t} step {s}: n={n} d={d}", flush=True)
                break
    bad += 0 if ok else 1
    if t % 10 == 9:
        print(f"{t + 1} cases, {bad} bad", flush=True)
print(f"done: {cases} cases, {bad} bad")
sys.exit(1 if bad else 0)
