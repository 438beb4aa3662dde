"""Randomised GPU-vs-oracle parity sweep (development aid): python tools/fuzz_parity.py [CASES] [SEED]
Random table sizes / widths / scales / thresholds / sign modes / prefilter forms / row ranges / curvatures;
argmin, top-k and count must match the oracle bit for bit.  FUZZ_KNOBS="knob=value,..." adds default knobs to every case."""
import os, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
from hyptokenizer_amd.engine import MergeEngine
from hyptokenizer_amd.synthetic import lorentz_table
from oracle import hm_oracle as O
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for t in range(cases):
    n = int(rng.integers(3500, 30000)) if rng.random() < 0.12 else int(rng.integers(2, 3500)); d = int(rng.choice([1, 2, 3, 5, 8, 10, 16, 23, 24, 31, 50, 64, 77, 100, 112, 124, 125, 128]))
    scale = float(rng.choice([0.001, 0.01, 0.05, 0.05, 0.2, 1.0])); mode = str(rng.choice(["lorentz", "lorentz", "reference"]))
    form = str(rng.choice(["f32", "bf16", "bf16-512", "bf16-k112", "bf16-512-k112"])); c = float(rng.choice([1.0, 1.0, 0.3, 4.0]))
    os.environ["HM_SCAN_PRECISION"] = form.split("-")[0]
    from hyptokenizer_amd import _lib
    _L = _lib.load()
    _L.hm_debug_set_default_knob(None, 0.0, 1)
    if "512" in form: _L.hm_debug_set_default_knob(b"big_rows", 2.0, 0)
    if form.endswith("k112"): _L.hm_debug_set_default_knob(b"kc_even", 1.0, 0)
    for kv in filter(None, os.environ.get("FUZZ_KNOBS", "").split(",")):      # e.g. FUZZ_KNOBS=dyn_slots=6: the scan's item queue on small tables too
        _L.hm_debug_set_default_knob(kv.split("=")[0].encode(), float(kv.split("=")[1]), 0)
    if rng.random() < 0.12:                               # the prefilter-free path (hm_exact.hip) on an ordinary table
        _L.hm_debug_set_default_knob(b"exact_search", 1.0, 0); form += "+exact"
    X = lorentz_table(n, d, seed=int(rng.integers(1 << 30)), scale=scale)
    if rng.random() < 0.2 and n > 10:                     # a few exact duplicates
        X[rng.integers(n, size=3)] = X[rng.integers(n, size=3)]
    table = torch.zeros((n + 4, d + 1), device="cuda"); table[:n] = X.cuda()
    eng = MergeEngine(n + 4, d + 1, mode); eng.set_table(table, n)
    Xn = X.numpy(); sm = 1 if mode == "lorentz" else 0
    m = min(n, 400)
    D = O.batch_distance(Xn[:m], Xn[:m], c, sm)[np.triu_indices(m, 1)]
    D = D[np.isfinite(D)]
    thr = float(np.quantile(D, rng.choice([0.001, 0.01, 0.2, 0.6, 1.0]))) * float(rng.choice([1.0, 1.0, 1.5])) if len(D) else 0.1
    if mode == "reference": thr = 0.1
    k = int(rng.choice([1, 7, 100, 1000, 10000]))
    if rng.random() < 0.4 and n > 4:
        r0 = int(rng.integers(0, n - 1)); r1 = int(rng.integers(r0 + 1, n + 1))
    else:
        r0, r1 = 0, -1
    rr1 = n if r1 < 0 else r1
    try:
        for rep in range(2):
            a = eng.argmin(c, thr, r0, r1)
            dd, ii, jj, cnt = eng.topk(c, thr, k, r0, r1)
            od, oi, oj, oc = O.pairwise_topk(Xn, n, c, thr, sm, k, r0, rr1, fast=bool(n > 300))
            ok = (cnt == oc and np.array_equal(ii, oi) and np.array_equal(jj, oj) and np.array_equal(dd.view(np.uint32), od.view(np.uint32))
                  and ((a is None) == (oc == 0)) and (a is None or ((a[1], a[2]) == (int(oi[0]), int(oj[0])) and np.float32(a[0]).view(np.uint32) == od.view(np.uint32)[0])))
            if not ok:
                bad += 1
                print("MISMATCH", dict(n=n, d=d, scale=scale, mode=mode, form=form, c=c, thr=thr, k=k, r0=r0, r1=r1, rep=rep), "gpu", cnt, a, "oracle", oc,
                      (float(od[0]), int(oi[0]), int(oj[0])) if oc else None, flush=True)
                break
        # merges: appended rows must carry the oracle's bits; the new row's nearest partner likewise
        if n >= 4:
            I = rng.integers(0, n, size=3).astype(np.int32); J = rng.integers(0, n, size=3).astype(np.int32)
            W = rng.random(3).astype(np.float32)
            want = O.midpoint_batch(Xn, I, J, W, c, sm)
            for q in range(3):
                eng.merge_append(int(I[q]), int(J[q]), float(W[q]), c, table, n + q)
            got = table[n:n + 3].cpu().numpy()
            if not np.array_equal(got.view(np.uint32), np.asarray(want, np.float32).view(np.uint32)):
                bad += 1
                print("MERGE MISMATCH", dict(n=n, d=d, scale=scale, mode=mode, c=c), flush=True)
            full = np.concatenate([Xn, np.asarray(want, np.float32)], 0)
            ra = eng.row_argmin(n + 2, n + 2, c, 1e30)
            dist = np.asarray(O.row_vs_all(full, n + 3, n + 2, c, sm), np.float32)[:n + 2]
            fin = np.nonzero(dist < np.float32(1e30))[0]
            best = None
            for q in fin.tolist():
                key = (int(dist[q].view(np.uint32)), q)
                best = key if best is None or key < best else best
            if (ra is None) != (best is None) or (ra is not None and (int(np.float32(ra[0]).view(np.uint32)), ra[1]) != best):
                bad += 1
                print("ROW ARGMIN MISMATCH", dict(n=n, d=d, scale=scale, mode=mode, c=c), ra, best, flush=True)
    except Exception as ex:
        bad += 1
        print("ERROR", dict(n=n, d=d, scale=scale, mode=mode, form=form, c=c, thr=thr, k=k, r0=r0, r1=r1), repr(ex)[:200], flush=True)
    if t % 25 == 24:
        print(f"{t + 1} cases, {bad} bad", flush=True)
print(f"{cases} cases, {bad} bad")
