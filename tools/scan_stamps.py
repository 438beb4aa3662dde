"""Per-phase cycle shares of the pair scan from a diagnostic build (-DHM_DIAG_STAMPS, tools/build_variant.sh):
HYPMERGE_LIB=build_variants/libhm_stamps.so python tools/scan_stamps.py "name:knob=value,..." ...
The stamps (s_memtime) slow the kernel down: only the ratios mean something."""
import ctypes as C, os, sys, torch
sys.path.insert(0, ".")
from hyptokenizer_amd import _lib
from hyptokenizer_amd.engine import MergeEngine
from hyptokenizer_amd.synthetic import lorentz_table
V, d = int(os.environ.get("AB_V", 50000)), int(os.environ.get("AB_D", 100))
L = _lib.load()
X = lorentz_table(V, d, seed=42, scale=0.05)
table = torch.zeros((V + 64, d + 1), device="cuda"); table[:V] = X.cuda()
names = ["kernel", "dequeue", "rows+prologue", "tile loops", "dma wait", "barrier", "slow path"]
for spec in sys.argv[1:]:
    name, _, kv = spec.partition(":")
    L.hm_debug_set_default_knob(None, 0.0, 1)
    for item in filter(None, kv.split(",")):
        k, v = item.split("=")
        _lib.check(L.hm_debug_set_default_knob(k.encode(), float(v), 0))
    e = MergeEngine(V + 64, d + 1, "lorentz"); e.set_table(table, V)
    for _ in range(4): e.argmin(1.0, 0.5)
    buf = (C.c_ulonglong * (16 + 4 * 8192 + 8 * 4096))()
    L.hm_diag_read(e._h, buf, 0)
    for _ in range(3): e.argmin(1.0, 0.5)
    L.hm_diag_read(e._h, buf, 0)
    e.argmin(1.0, 0.5)
    ms = e.scan_stats()["scan_ms"]
    L.hm_diag_read(e._h, buf, 1)
    for q in range(9): buf[q] *= 4
    tot = buf[0]
    print(f"{name} V={V} scan {ms:.4f} ms, waves/scan {buf[7]//4}, units/scan {buf[8]//4}, mean wave lifetime {tot/max(buf[7],1):.0f} cycles:", "  ".join(f"{n} {buf[i]/tot:.3f}" for i, n in enumerate(names) if i), flush=True)
    import numpy as np
    blk = np.array(buf[16:16 + 4 * 8192], dtype=np.uint64).reshape(-1, 4)
    blk = blk[blk[:, 1] > 0]
    st, en = blk[:, 0].astype(np.float64), blk[:, 1].astype(np.float64)
    t0 = st.min()
    st, en = (st - t0) / 100.0, (en - t0) / 100.0          # microseconds (last scan only)
    print(f"   last scan: {len(blk)} blocks; start us p0/p50/p100 {st.min():.1f}/{np.median(st):.1f}/{st.max():.1f}; end us p0/p10/p50/p90/p100 "
          f"{en.min():.1f}/{np.percentile(en, 10):.1f}/{np.median(en):.1f}/{np.percentile(en, 90):.1f}/{en.max():.1f}; "
          f"blocks per XCD {np.bincount(blk[:, 2].astype(int), minlength=8).tolist()}; units per block min/max {int(blk[:, 3].min())}/{int(blk[:, 3].max())}; "
          f"mean block-resident time / kernel span {float(((en - st).sum()) / (len(blk) * en.max())):.3f}", flush=True)
    for x in range(8):
        m = blk[:, 2] == x
        if m.any(): print(f"      XCD {x}: last block ends {en[m].max():.1f} us, median end {np.median(en[m]):.1f}", flush=True)
    un = np.array(buf[16 + 4 * 8192:], dtype=np.uint64).reshape(-1, 8)
    un = un[un[:, 6] > 0]
    if len(un):
        np.save(f"gpurun_out/units_{name}_{V}.npy", un)
        dur = (un[:, 6].astype(np.float64) - un[:, 5].astype(np.float64)) / 100.0
        nt = un[:, 4].astype(np.float64)
        per = dur / np.maximum(nt, 1)
        print(f"      {len(un)} units logged: us per tile p10/p50/p90/p100 {np.percentile(per, 10):.2f}/{np.median(per):.2f}/{np.percentile(per, 90):.2f}/{per.max():.2f}", flush=True)
        for x in range(8):
            m = (un[:, 1] == x) & (nt >= 8)
            if m.any(): print(f"         XCD {x}: {int(m.sum())} units >= 8 tiles, us per tile median {np.median(per[m]):.2f}, tiles total {int(nt[un[:, 1] == x].sum())}", flush=True)
        order = np.argsort(un[:, 5])
        late = order[-12:]
        print("         last units started (start us, dur us, rb, tiles, xcd):", [(round((float(un[i, 5]) - t0) / 100.0, 1), round(dur[i], 1), int(un[i, 2]), int(un[i, 4]), int(un[i, 1])) for i in late], flush=True)
        longest = np.argsort(-(un[:, 6].astype(np.float64)))[:8]
        print("         last units to END (start us, dur us, rb, tiles, xcd):", [(round((float(un[i, 5]) - t0) / 100.0, 1), round(dur[i], 1), int(un[i, 2]), int(un[i, 4]), int(un[i, 1])) for i in longest], flush=True)
