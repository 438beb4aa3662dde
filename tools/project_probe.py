"""hm_project_table timing: all rows live (table + both images written) against few rows live (table column only)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyptokenizer_amd import _lib  # noqa: E402
from hyptokenizer_amd.engine import MergeEngine  # noqa: E402
from hyptokenizer_amd.synthetic import lorentz_table  # noqa: E402

L = _lib.load()
n, D = 50000, 100
dev = torch.device("cuda:0")
X = lorentz_table(n, D, seed=42, scale=0.05)
table = torch.zeros((n + 2048, D + 1), device=dev)
table[:n] = X.to(dev)
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for live in (n, 64):
    eng = MergeEngine(n + 2048, D + 1, "lorentz", dev)
    eng.set_table(table, live)
    us = timed(lambda: L.hm_project_table(eng._h, C.c_void_p(table.data_ptr()), table.stride(0), n, C.c_float(1.0), stream))
    print(f"project_table rows={n} live={live}: {us:.2f} us")
    us = timed(lambda: eng.set_table(table, live), 20)
    print(f"set_table live={live}: {us:.2f} us")
    del eng
