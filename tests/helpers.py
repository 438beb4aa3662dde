"""Test-only helpers.

``OracleEngine`` is a TEST DOUBLE with the interface of ``hyptokenizer_amd.engine.MergeEngine``,
backed by the CPU oracle.  It exists so that the host logic of the tokenizer classes (cache
stepping, threshold dynamics, RNG order, save/load) can be checked against the reference's golden
vectors in the GPU-less build container.  Product code never constructs it: without a HIP device
the tokenizers raise ``HypMergeUnavailable``.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import hm_oracle as O

MODES = {"reference": 0, "lorentz": 1}


class OracleEngine:
    def __init__(self, max_rows: int, d1: int, sign_convention="reference", fast: bool = True):
        self.max_rows, self.d1 = max_rows, d1
        self.sign_mode = MODES[sign_convention] if isinstance(sign_convention, str) else int(sign_convention)
        self.X = np.zeros((max_rows, d1), np.float32)
        self._n = 0
        self.fast = fast
        self.calls = {"topk": 0, "argmin": 0, "candidates": 0, "set_table": 0}

    @property
    def n(self) -> int:
        return self._n

    def set_table(self, table: torch.Tensor, n_rows: int) -> None:
        self.calls["set_table"] += 1
        self.X[:n_rows] = table.detach().cpu().numpy()[:n_rows]
        self.X[n_rows:] = 0
        self._n = int(n_rows)

    def update_rows(self, table, r0, r1):
        self.X[r0:r1] = table.detach().cpu().numpy()[r0:r1]
        self._n = max(self._n, r1)

    def _range(self, row_begin, row_end):
        return max(0, row_begin), (self._n if row_end < 0 else min(row_end, self._n))

    def topk(self, c, thr, k, row_begin=0, row_end=-1, count=True):
        self.calls["topk"] += 1
        r0, r1 = self._range(row_begin, row_end)
        thr = float(np.float32(thr))
        if not thr > 0 or self._n < 2:
            e = np.empty(0, np.float32)
            return e, np.empty(0, np.int32), np.empty(0, np.int32), 0
        if not count and k > 0:                      # the uncounted form: -1 when at least k candidates exist
            d, i, j, total = O.pairwise_topk(self.X, self._n, float(c), thr, self.sign_mode, int(k), r0, r1, fast=self.fast)
            return d, i, j, (-1 if total >= k else total)
        return O.pairwise_topk(self.X, self._n, float(c), thr, self.sign_mode, max(int(k), 1), r0, r1, fast=self.fast) \
            if k > 0 else (np.empty(0, np.float32), np.empty(0, np.int32), np.empty(0, np.int32),
                           O.pairwise_count(self.X, self._n, float(c), thr, self.sign_mode, r0, r1))

    def argmin(self, c, thr, row_begin=0, row_end=-1):
        self.calls["argmin"] += 1
        d, i, j, cnt = self.topk(c, thr, 1, row_begin, row_end)
        self.calls["topk"] -= 1
        if cnt == 0:
            return None
        return float(d[0]), int(i[0]), int(j[0])

    def candidates(self, c, thr, row_begin=0, row_end=-1, cap=1 << 24):
        self.calls["candidates"] += 1
        r0, r1 = self._range(row_begin, row_end)
        i, j, d, total = O.pairwise_candidates(self.X, self._n, float(c), float(np.float32(thr)), self.sign_mode,
                                               cap=min(cap, 1 << 22), row_begin=r0, row_end=r1)
        return i, j, d, total

    def pair_distance(self, I, J, c):
        return O.pair_distance(self.X, np.asarray(I, np.int32), np.asarray(J, np.int32), float(c), self.sign_mode)

    def midpoint(self, I, J, W, c):
        return torch.from_numpy(O.midpoint_batch(self.X, I, J, W, float(c), self.sign_mode))

    def merge_append(self, i, j, w, c, table, new_row):
        row = O.midpoint_batch(self.X, [i], [j], [np.float32(w)], float(c), self.sign_mode)[0]
        self.X[new_row] = row
        table.detach()[new_row] = torch.from_numpy(row)
        self._n = max(self._n, new_row + 1)

    def row_vs_all(self, row, n, c):
        return O.row_vs_all(self.X, n, row, float(c), self.sign_mode)

    def row_argmin(self, row, n_partners, c, thr):
        d = np.asarray(O.row_vs_all(self.X, max(n_partners, row + 1), row, float(c), self.sign_mode), np.float32)[:n_partners]
        best = None
        for i in np.nonzero(d < np.float32(thr))[0].tolist():
            if i == row:
                continue
            key = (float(d[i]), min(i, row), max(i, row))
            if best is None or key < best:
                best = key
        return best

    # -- config 5 ---------------------------------------------------------------------------------
    def coherence_distances(self, I, J, W, S, c):
        return O.coherence_distances(self.X, I, J, W, S, float(c), self.sign_mode)

    def project_table(self, table, n_rows, c):
        t = table.detach()
        X = np.ascontiguousarray(t.cpu().numpy()[:n_rows], np.float32)
        O.project_table(X, n_rows, float(c))
        t[:n_rows] = torch.from_numpy(X)
        self.X[:self._n] = X[:self._n]

    def rows_pair_distance(self, table, A, B, c):
        T = table.detach().cpu().numpy()
        return O.distance(T[np.asarray(A, np.int64)], T[np.asarray(B, np.int64)], float(c), self.sign_mode)

    def merge_append_batch(self, I, J, W, c, table, first_row, independent=False):
        for t, (i, j, w) in enumerate(zip(np.asarray(I).tolist(), np.asarray(J).tolist(), np.asarray(W).tolist())):
            self.merge_append(int(i), int(j), np.float32(w), c, table, first_row + t)

    def truncate(self, n_rows):
        self.X[n_rows:] = 0
        self._n = int(n_rows)

    def count_candidates(self, c, thr, n_limit=-1):
        n = self._n if n_limit < 0 else min(n_limit, self._n)
        thr = float(np.float32(thr))
        if not thr > 0 or n < 2:
            return 0
        return O.pairwise_count(self.X, n, float(c), thr, self.sign_mode, 0, n)

    # -- device-resident loops (same record format as MergeEngine) ---------------------------------
    def set_token_lengths(self, lengths):
        self.lens = list(int(v) for v in lengths)

    def std_merge_steps(self, c, thr, table, steps):
        recs, done, alive = [], 0, True
        for _ in range(steps):
            if not alive:
                recs.append((3, 0.0, -1, -1))
                continue
            hit = self.argmin(c, thr)
            if hit is None:
                recs.append((0, 0.0, -1, -1))
                alive = False
                continue
            d, i, j = hit
            li, lj = self.lens[i], self.lens[j]
            self.merge_append(i, j, lj / (li + lj), c, table, self._n)
            self.lens.append(li + lj)
            recs.append((1, d, i, j))
            done += 1
        return recs, done

    def incr_merge_steps(self, c, thr, table, steps, best):
        recs, done = [], 0
        for _ in range(steps):
            if best is None:
                recs.append((0, 0.0, -1, -1) if len([r for r in recs if r[0] != 1]) == 0 else (3, 0.0, -1, -1))
                continue
            d, i, j = best
            li, lj = self.lens[i], self.lens[j]
            row = self._n
            self.merge_append(i, j, lj / (li + lj), c, table, row)
            self.lens.append(li + lj)
            recs.append((1, d, i, j))
            done += 1
            cand = self.row_argmin(row, row, c, thr)
            if cand is not None and cand < best:
                best = cand
        return recs, done, best

    def scan_stats(self):
        return {"scan_ms": 0.0, "pairs": 0, "emitted": 0, "passes": 0}


def bits(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def nan_equal_close(a, b, atol):
    """|a-b| <= atol with NaNs required in the same places."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    return bool(np.all(np.abs(a[~na] - b[~nb]) <= atol))


# ----------------------------------------------------------------------------------------------
# oracle-backed stand-ins for the device functions of hyptokenizer_amd.embedding.lorentz_model,
# used by CPU tests of callers (CLI) only
# ----------------------------------------------------------------------------------------------
def oracle_exp_map(x, v, c=1.0):
    shape = torch.broadcast_shapes(x.shape, v.shape)
    xb = x.expand(shape).reshape(-1, shape[-1]).numpy()
    vb = v.expand(shape).reshape(-1, shape[-1]).numpy()
    return torch.from_numpy(O.exp_map(xb, vb)).reshape(shape)


def oracle_project(x, c=1.0):
    return torch.from_numpy(O.project(x.reshape(-1, x.shape[-1]).numpy(), float(c))).reshape(x.shape)


def oracle_distance(x, y, c=1.0, *, sign_convention="reference"):
    shape = torch.broadcast_shapes(x.shape, y.shape)
    xb = x.expand(shape).reshape(-1, shape[-1]).contiguous().numpy()
    yb = y.expand(shape).reshape(-1, shape[-1]).contiguous().numpy()
    return torch.from_numpy(O.distance(xb, yb, float(c), MODES[sign_convention])).reshape(shape[:-1])
