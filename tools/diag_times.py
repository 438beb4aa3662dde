"""Per-block timing of a HM_DIAG_TIMES build (development aid): python tools/diag_times.py LIB"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, ".")
from hyptokenizer_amd import _lib
from hyptokenizer_amd.engine import MergeEngine
from hyptokenizer_amd.synthetic import lorentz_table
_lib.LIB_PATH = sys.argv[1]
V, d = 50000, 100
X = lorentz_table(V, d, seed=42, scale=0.05)
table = torch.zeros((V + 64, d + 1), device="cuda"); table[:V] = X.cuda()
e = MergeEngine(V + 64, d + 1, "lorentz"); e.set_table(table, V)
for _ in range(3):
    r = e.argmin(1.0, 0.5)
print(r, e.scan_stats())
L = e._L
buf = (C.c_uint32 * 4096)()
L.hm_debug_read_hist.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
L.hm_debug_read_hist(e._h, buf, 4096)
a = np.array(buf, np.uint32).reshape(256, 4, 4)     # block, wave, field
ticks, slow, sticks, pt = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
us = ticks / 100.0
print("block time us (wave 0): min %.1f median %.1f max %.1f" % (us[:, 0].min(), np.median(us[:, 0]), us[:, 0].max()))
order = np.argsort(-us[:, 0])[:12]
for b in order:
    print("block", b, "us", us[b].round(1), "slow entries", slow[b], "slow us", (sticks[b] / 100.0).round(1), "passes", pt[b, 0] >> 16, "tiles", pt[b, 0] & 0xffff)
t0 = slow[:, 1].astype(np.int64)
t0 = (t0 - t0.min()) / 100.0
end = t0 + us[:, 1]
print("start offsets us (even blocks): first half median %.1f max %.1f | second half median %.1f max %.1f" % (
    np.median(t0[:128]), t0[:128].max(), np.median(t0[128:]), t0[128:].max()))
print("end us: first half median %.1f max %.1f | second half median %.1f max %.1f" % (np.median(end[:128]), end[:128].max(), np.median(end[128:]), end[128:].max()))
print("duration by even block index (every 8th):", us[::8, 0].round(0))
print("start offsets sample:", t0[::16].round(1))
slow = slow.copy(); slow[:, 1] = 0
print("total slow entries", slow.sum(), "total slow us", sticks.sum() / 100.0, "mean us per entry", sticks.sum() / 100.0 / max(1, slow.sum()))
tiles = (pt[:, 0] & 0xffff).astype(np.float64); passes = (pt[:, 0] >> 16).astype(np.float64)
A = np.stack([tiles, passes, np.ones_like(tiles)], 1)
coef, *_ = np.linalg.lstsq(A, us[:, 0], rcond=None)
print("fit: block us = %.3f * tiles + %.3f * passes + %.1f   (tiles %d..%d, passes %d..%d)" % (coef[0], coef[1], coef[2], tiles.min(), tiles.max(), passes.min(), passes.max()))
print("sum tiles (sampled even blocks)", tiles.sum(), "sum passes", passes.sum())
