"""Stateful GPU-vs-oracle fuzz (development aid): random sequences of searches, merges, row edits and range searches on
one engine -- exercises the running-key seed, the armed-next-search state and the top-k cut prediction.
python tools/fuzz_stateful.py [SEQUENCES] [SEED]"""
import os, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
from hyptokenizer_amd.engine import MergeEngine
from hyptokenizer_amd.synthetic import lorentz_table
from oracle import hm_oracle as O
seqs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for t in range(seqs):
    n = int(rng.integers(300, 2500)); d = int(rng.choice([3, 10, 30, 50, 100])); mode = str(rng.choice(["lorentz", "lorentz", "reference"]))
    os.environ["HM_SCAN_PRECISION"] = str(rng.choice(["f32", "bf16"]))
    cap = n + 40
    X = np.zeros((cap, d + 1), np.float32); X[:n] = lorentz_table(n, d, seed=int(rng.integers(1 << 30)), scale=0.05).numpy()
    table = torch.from_numpy(X).cuda()
    eng = MergeEngine(cap, d + 1, mode); eng.set_table(table, n)
    sm = 1 if mode == "lorentz" else 0
    D = O.batch_distance(X[:300], X[:300], 1.0, sm)[np.triu_indices(300, 1)]
    thr = float(np.quantile(D[np.isfinite(D)], 0.02)) if mode == "lorentz" else 0.1
    rec = torch.zeros(4, dtype=torch.int32, device="cuda")
    log = []
    for step in range(14):
        op = str(rng.choice(["argmin", "argmin", "argmin", "topk", "topk_nc", "refresh", "merge", "merge", "merges", "edit", "range", "dev", "rowmin", "thr"]))
        log.append(op)
        try:
            if op == "merge" and n < cap - 1:
                i, j = int(rng.integers(n)), int(rng.integers(n)); w = float(rng.random())
                X[n] = O.midpoint_batch(X[:n], np.array([i], np.int32), np.array([j], np.int32), np.array([w], np.float32), 1.0, sm)[0]
                eng.merge_append(i, j, w, 1.0, table, n); n += 1
                continue
            if op == "merges":                     # a handful of appended rows: what a fast-tokenizer cycle does before its refresh
                for _ in range(int(rng.integers(1, 6))):
                    if n >= cap - 1:
                        break
                    i, j = int(rng.integers(n)), int(rng.integers(n)); w = float(rng.random())
                    X[n] = O.midpoint_batch(X[:n], np.array([i], np.int32), np.array([j], np.int32), np.array([w], np.float32), 1.0, sm)[0]
                    eng.merge_append(i, j, w, 1.0, table, n); n += 1
                continue
            if op == "edit":
                r = int(rng.integers(n)); X[r] = lorentz_table(1, d, seed=int(rng.integers(1 << 30)), scale=0.05).numpy()[0]
                table[r] = torch.from_numpy(X[r]).cuda(); eng.update_rows(table, r, r + 1)
                continue
            if op == "thr":
                thr *= float(rng.choice([0.8, 1.25])); continue
            if op == "rowmin":
                eng.row_argmin(int(rng.integers(n)), n, 1.0, thr); continue
            r0, r1 = 0, -1
            if op == "range":
                r0 = int(rng.integers(0, n - 1)); r1 = int(rng.integers(r0 + 1, n + 1))
            rr1 = n if r1 < 0 else r1
            od, oi, oj, oc = O.pairwise_topk(X[:n], n, 1.0, thr, sm, 50, r0, rr1, fast=True)
            want = None if oc == 0 else (int(od.view(np.uint32)[0]), int(oi[0]), int(oj[0]))
            if op == "topk":
                dd, ii, jj, cnt = eng.topk(1.0, thr, 50, r0, r1)
                ok = cnt == oc and np.array_equal(ii, oi) and np.array_equal(jj, oj) and np.array_equal(dd.view(np.uint32), od.view(np.uint32))
            elif op in ("topk_nc", "refresh"):          # the uncounted refresh (incremental where it applies), whole or in two halves
                out = None
                if op == "refresh" and eng.topk_refresh_begin(1.0, thr, 50):
                    out = eng.topk_refresh_end()
                if out is None:
                    out = eng.topk(1.0, thr, 50, count=False)[:3]
                dd, ii, jj = out
                ok = np.array_equal(ii, oi) and np.array_equal(jj, oj) and np.array_equal(dd.view(np.uint32), od.view(np.uint32))
            elif op == "dev":
                eng.argmin_into(1.0, thr, r0, r1, rec); g = rec.cpu().numpy()
                got = None if g[0] == 0 else (int(np.uint32(g[1])), int(g[2]), int(g[3]))
                ok = g[0] != 2 and got == want
            else:
                a = eng.argmin(1.0, thr, r0, r1)
                got = None if a is None else (int(np.float32(a[0]).view(np.uint32)), a[1], a[2])
                ok = got == want
            if not ok:
                bad += 1
                print("MISMATCH", dict(seq=t, n=n, d=d, mode=mode, form=os.environ["HM_SCAN_PRECISION"], thr=thr, op=op, r0=r0, r1=r1), log, flush=True)
                break
        except Exception as ex:
            bad += 1
            print("ERROR", dict(seq=t, n=n, d=d, mode=mode, op=op), repr(ex)[:200], log, flush=True)
            break
print(f"{seqs} sequences, {bad} bad")
