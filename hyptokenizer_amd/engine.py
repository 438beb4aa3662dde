"""MergeEngine: the host-side handle of the gfx950 merge engine (libhypmerge.so).

One engine per GPU.  The engine keeps a device-resident *scan image* of the live rows of the
token-embedding table (``HyperbolicTokenizer.embeddings[:n]``, hyperbolic_merge.py:145-153) and
serves the candidate searches and the midpoint update of the merge loop through the C ABI of
``include/hypmerge.h``.  PyTorch provides device memory and the stream only.

The product path has no CPU implementation: constructing a ``MergeEngine`` without a HIP device or
without the built library raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import SIGN_LORENTZ, SIGN_REFERENCE, HypMergeError, HypMergeUnavailable  # noqa: F401

SIGN_MODES = {"reference": SIGN_REFERENCE, "lorentz": SIGN_LORENTZ}
PREFILTERS = {"auto": _lib.PREFILTER_AUTO, "f32": _lib.PREFILTER_F32, "bf16": _lib.PREFILTER_BF16}
MAX_ROWS = 131072        # hm_engine_create: largest table
MAX_WIDTH = 129          # ... and widest row (d + 1)


def sign_mode_id(sign_convention) -> int:
    if isinstance(sign_convention, str):
        try:
            return SIGN_MODES[sign_convention]
        except KeyError:
            raise ValueError(f"sign_convention must be 'reference' or 'lorentz', got {sign_convention!r}") from None
    if sign_convention in (0, 1):
        return int(sign_convention)
    raise ValueError(f"bad sign convention {sign_convention!r}")


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _np_ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def _f(x) -> float:
    """Python float of a number or of a (possibly grad-requiring) 0-d tensor such as the enhanced tokenizer's curvature."""
    return float(x.detach()) if isinstance(x, torch.Tensor) else float(x)


def _nan_c(c) -> bool:
    """A NaN curvature -- what the enhanced tokenizer's Adam step leaves behind once a NaN row (merge of identical rows, SURVEY
    F6) is sampled into its loss: the reference then computes on silently, every distance is NaN (``acosh(u) / sqrt(nan)``), no
    pair is below any threshold and ``project_to_hyperboloid`` writes NaN time coordinates.  The C ABI rejects a curvature that
    is not > 0, so the host mirror answers these calls itself, with exactly those values."""
    v = _f(c)
    return v != v


class MergeEngine:
    """Handle of one ``hm_engine`` (include/hypmerge.h)."""

    def __init__(self, max_rows: int, d1: int, sign_convention="reference", device: Optional[torch.device] = None,
                 prefilter: str = "auto"):
        """``prefilter``: form of the pair scan's MFMA prefilter -- "auto" (bf16 from d >= 24), "f32", "bf16".
        Results do not depend on it (every reported distance is re-evaluated in the canonical fp32 arithmetic)."""
        self._L = _lib.load()
        if prefilter not in PREFILTERS:
            raise ValueError(f"prefilter must be one of {sorted(PREFILTERS)}, got {prefilter!r}")
        if not torch.cuda.is_available():
            raise HypMergeUnavailable("MergeEngine needs a HIP device (torch.cuda.is_available() is False); "
                                      "there is no CPU fallback")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != "cuda":
            raise HypMergeUnavailable(f"MergeEngine needs a cuda(HIP) device, got {dev}")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.max_rows = int(max_rows)
        self.d1 = int(d1)
        self.sign_mode = sign_mode_id(sign_convention)
        self._h = C.c_void_p(0)
        _lib.check(self._L.hm_engine_create(C.byref(self._h), self.device.index, self.max_rows, self.d1,
                                            self.sign_mode, PREFILTERS[prefilter]))
        self._rec_buf = np.zeros(4 * _lib.LOOP_MAX_STEPS, np.uint32)
        self._refresh_k = None       # k of a pending topk_refresh_begin

    # ------------------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.hm_engine_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> C.c_void_p:
        if getattr(self, "_refresh_k", None) is not None:
            raise RuntimeError("a top-k refresh is pending on this engine: call topk_refresh_end() first")
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, st: int) -> None:
        _lib.check(st, self._h)

    def _check_table(self, table: torch.Tensor) -> torch.Tensor:
        t = table.detach()
        if t.device != self.device or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 \
                or t.shape[1] != self.d1:
            raise ValueError("table must be a float32 [rows, d1] tensor with unit inner stride on the engine's device")
        return t

    # ------------------------------------------------------------------------------------------
    @property
    def n(self) -> int:
        return int(self._L.hm_rows(self._h))

    def set_table(self, table: torch.Tensor, n_rows: int) -> None:
        t = self._check_table(table)
        self._chk(self._L.hm_set_table(self._h, _ptr(t), t.stride(0), int(n_rows), self._stream()))

    def update_rows(self, table: torch.Tensor, row_begin: int, row_end: int) -> None:
        t = self._check_table(table)
        self._chk(self._L.hm_update_rows(self._h, _ptr(t), t.stride(0), int(row_begin), int(row_end), self._stream()))

    # ------------------------------------------------------------------------------------------
    def argmin(self, c: float, thr: float, row_begin: int = 0, row_end: int = -1) -> Optional[Tuple[float, int, int]]:
        """Nearest pair (d, i, j) with d < thr in the reference's order, or None."""
        if _nan_c(c):
            return None
        d, i, j, f = C.c_float(0), C.c_int32(-1), C.c_int32(-1), C.c_int32(0)
        self._chk(self._L.hm_pairwise_argmin(self._h, _f(c), float(thr), int(row_begin), int(row_end),
                                             C.byref(d), C.byref(i), C.byref(j), C.byref(f), self._stream()))
        if not f.value:
            return None
        return float(d.value), int(i.value), int(j.value)

    def row_argmin(self, row: int, n_partners: int, c: float, thr: float) -> Optional[Tuple[float, int, int]]:
        """Nearest partner of `row` among rows [0, n_partners): (d, i, j) with i < j, or None."""
        if _nan_c(c):
            return None
        d, i, j, f = C.c_float(0), C.c_int32(-1), C.c_int32(-1), C.c_int32(0)
        self._chk(self._L.hm_row_argmin(self._h, int(row), int(n_partners), _f(c), float(thr),
                                        C.byref(d), C.byref(i), C.byref(j), C.byref(f), self._stream()))
        if not f.value:
            return None
        return float(d.value), int(i.value), int(j.value)

    def argmin_into(self, c: float, thr: float, row_begin: int, row_end: int, rec: torch.Tensor) -> None:
        """Asynchronous nearest-pair search: writes int32[4] {found, bits(d), i, j} into the device
        tensor `rec` on the current stream (found = 2: buffer overflow, use ``argmin``)."""
        if rec.device != self.device or rec.dtype != torch.int32 or rec.numel() < 4 or not rec.is_contiguous():
            raise ValueError("rec must be a contiguous int32[4] tensor on the engine's device")
        self._chk(self._L.hm_pairwise_argmin_dev(self._h, _f(c), float(thr), int(row_begin), int(row_end),
                                                 _ptr(rec), self._stream()))

    def topk(self, c: float, thr: float, k: int, row_begin: int = 0, row_end: int = -1, count: bool = True):
        """k smallest candidates (d, i, j) in order and the exact candidate count.  ``count=False``: the
        count is -1 when at least k candidates exist (not counted: the scan then only visits what lies below
        its emission cut; ``count_candidates`` delivers the number later)."""
        if _nan_c(c):
            return np.empty(0, np.float32), np.empty(0, np.int32), np.empty(0, np.int32), 0
        k = int(k)
        d = np.empty(k, np.float32)
        i = np.empty(k, np.int32)
        j = np.empty(k, np.int32)
        n_out, total = C.c_int64(0), C.c_int64(0)
        fn = self._L.hm_pairwise_topk if count else self._L.hm_pairwise_topk_nocount
        self._chk(fn(self._h, _f(c), float(thr), k, int(row_begin), int(row_end),
                     _np_ptr(d), _np_ptr(i), _np_ptr(j), C.byref(n_out), C.byref(total), self._stream()))
        m = int(n_out.value)
        return d[:m], i[:m], j[:m], int(total.value)

    def topk_refresh_begin(self, c: float, thr: float, k: int) -> bool:
        """Enqueue the refresh of a table that only grew since the last whole-table ``topk`` and return at once
        (``False``: not of that kind -- use ``topk``).  Nothing else may be asked of the engine until
        ``topk_refresh_end``."""
        if _nan_c(c):
            return False
        st = self._L.hm_topk_refresh_begin(self._h, _f(c), float(thr), int(k), self._stream())
        if st == _lib.HM_E_NA:                     # the refresh is not of the incremental kind (HM_E_STATE stays an error)
            return False
        self._chk(st)
        self._refresh_k = int(k)
        return True

    def topk_refresh_end(self):
        """Wait for the refresh -> (d, i, j), or ``None`` when it has to be redone through ``topk`` (more new entries
        than the device-side sort takes)."""
        k = self._refresh_k
        if k is None:
            raise RuntimeError("no refresh pending")
        self._refresh_k = None
        d = np.empty(k, np.float32)
        i = np.empty(k, np.int32)
        j = np.empty(k, np.int32)
        n_out = C.c_int64(0)
        st = self._L.hm_topk_refresh_end(self._h, _np_ptr(d), _np_ptr(i), _np_ptr(j), C.byref(n_out))
        if st == -2:                               # HM_E_CAPACITY
            return None
        self._chk(st)
        m = int(n_out.value)
        return d[:m], i[:m], j[:m]

    def count_candidates(self, c: float, thr: float, n_limit: int = -1) -> int:
        """Exact number of candidates among the first ``n_limit`` rows (-1: all live rows)."""
        if _nan_c(c):
            return 0
        total = C.c_int64(0)
        self._chk(self._L.hm_pairwise_count(self._h, _f(c), float(thr), int(n_limit), C.byref(total), self._stream()))
        return int(total.value)

    def set_prefilter(self, prefilter: str) -> None:
        self._chk(self._L.hm_set_prefilter(self._h, PREFILTERS[prefilter]))

    def debug_force_cut(self, cut_bits: int, k: int, c: float) -> None:
        """test hook (include/hypmerge.h hm_debug_force_cut)"""
        self._chk(self._L.hm_debug_force_cut(self._h, int(cut_bits), int(k), _f(c)))

    def candidates(self, c: float, thr: float, row_begin: int = 0, row_end: int = -1, cap: int = 1 << 24):
        """All candidates in row-major order: (i, j, d, total)."""
        if _nan_c(c):
            return np.empty(0, np.int32), np.empty(0, np.int32), np.empty(0, np.float32), 0
        total = C.c_int64(0)
        # first call sizes the arrays
        self._chk(self._L.hm_pairwise_candidates(self._h, _f(c), float(thr), int(row_begin), int(row_end), 0,
                                                 None, None, None, C.byref(total), self._stream()))
        m = min(int(total.value), int(cap))
        i = np.empty(m, np.int32)
        j = np.empty(m, np.int32)
        d = np.empty(m, np.float32)
        if m:
            self._chk(self._L.hm_pairwise_candidates(self._h, _f(c), float(thr), int(row_begin), int(row_end), m,
                                                     _np_ptr(i), _np_ptr(j), _np_ptr(d), C.byref(total),
                                                     self._stream()))
            order = np.lexsort((j, i))
            i, j, d = i[order], j[order], d[order]
        return i, j, d, int(total.value)

    # ------------------------------------------------------------------------------------------
    def _idx(self, a: Sequence[int]) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.int32).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=self.device)

    def pair_distance(self, I, J, c: float) -> np.ndarray:
        if _nan_c(c):
            return np.full(len(I), np.nan, np.float32)
        ti, tj = self._idx(I), self._idx(J)
        out = torch.empty(ti.numel(), dtype=torch.float32, device=self.device)
        self._chk(self._L.hm_pair_distance(self._h, _ptr(ti), _ptr(tj), ti.numel(), _f(c), _ptr(out), self._stream()))
        return out.cpu().numpy()

    def midpoint(self, I, J, W, c: float) -> torch.Tensor:
        ti, tj = self._idx(I), self._idx(J)
        tw = torch.as_tensor(np.ascontiguousarray(W, dtype=np.float32), device=self.device)
        out = torch.empty((ti.numel(), self.d1), dtype=torch.float32, device=self.device)
        self._chk(self._L.hm_midpoint_batch(self._h, _ptr(ti), _ptr(tj), _ptr(tw), ti.numel(), _f(c), _ptr(out),
                                            self._stream()))
        return out

    def merge_append(self, i: int, j: int, w: float, c: float, table: torch.Tensor, new_row: int) -> None:
        t = self._check_table(table)
        if not (0 <= new_row < t.shape[0]):
            raise ValueError("new_row outside the table")
        self._chk(self._L.hm_merge_append(self._h, int(i), int(j), float(w), _f(c), _ptr(t), t.stride(0),
                                          int(new_row), self._stream()))

    def merge_append_batch(self, I, J, W, c: float, table: torch.Tensor, first_row: int, independent: bool = False) -> None:
        """Several merges in one launch: merge t -> row first_row + t.  ``independent``: no merge reads a row the
        batch writes (all operands < first_row): all at once instead of a sequential chain.  Host arrays (numpy /
        lists, at most 4096 merges per call) go through the engine's pinned staging buffer."""
        t = self._check_table(table)
        ii = np.ascontiguousarray(I, dtype=np.int32)
        jj = np.ascontiguousarray(J, dtype=np.int32)
        ww = np.ascontiguousarray(W, dtype=np.float32)
        if not (0 <= first_row and first_row + ii.shape[0] <= t.shape[0]):
            raise ValueError("merge batch outside the table")
        for lo in range(0, ii.shape[0], 4096):
            hi = min(lo + 4096, ii.shape[0])
            self._chk(self._L.hm_merge_append_batch_host(self._h, _np_ptr(ii[lo:hi]), _np_ptr(jj[lo:hi]), _np_ptr(ww[lo:hi]), hi - lo,
                                                         _f(c), _ptr(t), t.stride(0), int(first_row) + lo,
                                                         1 if independent else 0, self._stream()))

    def truncate(self, n_rows: int) -> None:
        self._chk(self._L.hm_truncate(self._h, int(n_rows), self._stream()))

    # -- device-resident loops ----------------------------------------------------------------------
    def set_token_lengths(self, lengths) -> None:
        a = np.ascontiguousarray(lengths, dtype=np.int32)
        self._chk(self._L.hm_set_token_lengths(self._h, _np_ptr(a), a.shape[0], self._stream()))

    def _unpack(self, steps: int):
        r = self._rec_buf[:4 * steps].reshape(steps, 4)
        return [(int(f), float(np.uint32(b).view(np.float32)), int(i), int(j)) for f, b, i, j in r.tolist()]

    def std_merge_steps(self, c: float, thr: float, table: torch.Tensor, steps: int):
        """``steps`` (<= 256) iterations of the standard loop on the device -> (records, done);
        record = (found, d, i, j): found 1 merged, 0 no candidate, 2 overflow at this step, 3 skipped."""
        t = self._check_table(table)
        done = C.c_int64(0)
        self._chk(self._L.hm_std_merge_steps(self._h, _f(c), float(thr), _ptr(t), t.stride(0), int(steps),
                                             _np_ptr(self._rec_buf), C.byref(done), self._stream()))
        return self._unpack(int(steps)), int(done.value)

    def incr_merge_steps(self, c: float, thr: float, table: torch.Tensor, steps: int, best):
        """``steps`` iterations with the nearest pair maintained incrementally; ``best`` = (d, i, j) of the
        current table or None -> (records, done, best after the last executed step)."""
        t = self._check_table(table)
        b = np.zeros(4, np.uint32)
        if best is not None:
            b[0] = 1
            b[1] = np.float32(best[0]).view(np.uint32)
            b[2], b[3] = best[1], best[2]
        done = C.c_int64(0)
        self._chk(self._L.hm_incr_merge_steps(self._h, _f(c), float(thr), _ptr(t), t.stride(0), int(steps),
                                              _np_ptr(b), _np_ptr(self._rec_buf), C.byref(done), self._stream()))
        nb = (float(b[1:2].view(np.float32)[0]), int(b[2]), int(b[3])) if b[0] == 1 else None
        return self._unpack(int(steps)), int(done.value), nb

    # -- row-sharded device-resident loop ----------------------------------------------------------------
    def shard_loop_begin(self) -> None:
        self._chk(self._L.hm_shard_loop_begin(self._h, self._stream()))

    def shard_merge_step(self, recs: torch.Tensor, world: int, c: float, table: torch.Tensor, step: int) -> None:
        """Global nearest pair of the gathered records (int32 [world, 4] on this device) and its merge, on the device."""
        t = self._check_table(table)
        if recs.device != self.device or recs.dtype != torch.int32 or recs.numel() < 4 * world or not recs.is_contiguous():
            raise ValueError("recs must be a contiguous int32[world * 4] tensor on the engine's device")
        self._chk(self._L.hm_shard_merge_step(self._h, _ptr(recs), int(world), _f(c), _ptr(t), t.stride(0), int(step), self._stream()))

    def shard_loop_end(self, steps: int):
        done = C.c_int64(0)
        self._chk(self._L.hm_shard_loop_end(self._h, int(steps), _np_ptr(self._rec_buf), C.byref(done), self._stream()))
        return self._unpack(int(steps)), int(done.value)

    # -- the exchange step inside the library (RCCL communicator bound to the engine) -----------------------
    def comm_init(self, group=None) -> None:
        """Bind an RCCL communicator of the ranks of ``group`` (a ``torch.distributed`` group of any backend: it only
        carries the 128-byte id from rank 0 to the others) to this engine.  Collective: every rank calls it."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ident = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
            _lib.check(self._L.hm_comm_unique_id(buf))
            ident = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
        gdev = self.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        ident = ident.to(gdev)
        dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(ident.cpu().numpy().tobytes())
        with torch.cuda.device(self.device):
            self._chk(self._L.hm_comm_init(self._h, C.c_char_p(raw), int(rank), int(world)))

    def comm_info(self):
        """(rank, world) of the bound communicator, or None."""
        r, w = C.c_int(-1), C.c_int(0)
        self._L.hm_comm_info(self._h, C.byref(r), C.byref(w))
        return (int(r.value), int(w.value)) if w.value > 0 else None

    def comm_destroy(self) -> None:
        self._chk(self._L.hm_comm_destroy(self._h))

    def shard_merge_steps(self, c: float, thr: float, table: torch.Tensor, steps: int):
        """``steps`` (<= 256) iterations of the row-sharded standard loop, enqueued by the library: per step the scan of
        this rank's rows, the all-gather of the records (RCCL) and the merge -> (records, done), identical on every rank."""
        t = self._check_table(table)
        done = C.c_int64(0)
        self._chk(self._L.hm_shard_merge_steps(self._h, _f(c), float(thr), _ptr(t), t.stride(0), int(steps),
                                               _np_ptr(self._rec_buf), C.byref(done), self._stream()))
        return self._unpack(int(steps)), int(done.value)

    def global_argmin(self, c: float, thr: float) -> Optional[Tuple[float, int, int]]:
        """Nearest pair of the whole table through the bound communicator (collective)."""
        d, i, j, f = C.c_float(0), C.c_int32(-1), C.c_int32(-1), C.c_int32(0)
        self._chk(self._L.hm_global_argmin(self._h, _f(c), float(thr), C.byref(d), C.byref(i), C.byref(j), C.byref(f), self._stream()))
        return (float(d.value), int(i.value), int(j.value)) if f.value else None

    def global_topk(self, c: float, thr: float, k: int):
        """k smallest candidates of the whole table in order and their exact number (collective): the ranks' lists are
        gathered and merged on the device."""
        k = int(k)
        d, i, j = np.empty(k, np.float32), np.empty(k, np.int32), np.empty(k, np.int32)
        n_out, total = C.c_int64(0), C.c_int64(0)
        self._chk(self._L.hm_global_topk(self._h, _f(c), float(thr), k, _np_ptr(d), _np_ptr(i), _np_ptr(j), C.byref(n_out),
                                         C.byref(total), self._stream()))
        m = int(n_out.value)
        return d[:m], i[:m], j[:m], int(total.value)

    def debug_time_loops(self, on: bool) -> None:
        """Measurement aid: event pairs around every scan of the following ``std_merge_steps`` batches."""
        self._chk(self._L.hm_debug_time_loops(self._h, 1 if on else 0))

    def last_loop_timing(self) -> dict:
        b, s, k = C.c_float(0), C.c_float(0), C.c_int64(0)
        self._L.hm_last_loop_timing(self._h, C.byref(b), C.byref(s), C.byref(k))
        return {"batch_ms": float(b.value), "scan_ms": float(s.value), "steps": int(k.value)}

    # -- enhanced tokenizer (config 5) ----------------------------------------------------------------
    def coherence_distances(self, I, J, W, S, c: float) -> np.ndarray:
        """distance(exp_map(x_i, w * log_map(x_i, x_j)), x_s) for every candidate t and its samples S[t, :]
        (enhanced_fast_hyperbolic_merge.py:308-333) -> float32 [b, ns]."""
        if _nan_c(c):
            return np.full((len(I), np.asarray(S).reshape(len(I), -1).shape[1] if len(I) else 0), np.nan, np.float32)
        ti, tj = self._idx(I), self._idx(J)
        tw = torch.as_tensor(np.ascontiguousarray(W, dtype=np.float32), device=self.device)
        S = np.ascontiguousarray(S, dtype=np.int32).reshape(ti.numel(), -1)
        ts = torch.as_tensor(S, device=self.device)
        out = torch.empty(S.shape, dtype=torch.float32, device=self.device)
        if S.size:
            self._chk(self._L.hm_coherence_batch(self._h, _ptr(ti), _ptr(tj), _ptr(tw), _ptr(ts), ti.numel(), S.shape[1],
                                                 _f(c), _ptr(out), self._stream()))
        return out.cpu().numpy()

    def coherence_distances_begin(self, I, J, W, S, c: float):
        """The same launch without the wait: kernel and the copy of its result into pinned host memory are enqueued and a
        handle comes back at once -- the caller draws the NEXT batch's samples on the host meanwhile (the enhanced
        tokenizer's scoring: the host RNG is the long pole) and collects with ``coherence_distances_end``."""
        if _nan_c(c):
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            return (torch.from_numpy(self.coherence_distances(I, J, W, S, c)), ev, None)
        ti, tj = self._idx(I), self._idx(J)
        tw = torch.as_tensor(np.ascontiguousarray(W, dtype=np.float32), device=self.device)
        S = np.ascontiguousarray(S, dtype=np.int32).reshape(ti.numel(), -1)
        ts = torch.as_tensor(S, device=self.device)
        out = torch.empty(S.shape, dtype=torch.float32, device=self.device)
        host = torch.empty(S.shape, dtype=torch.float32).pin_memory()
        ev = torch.cuda.Event()
        if S.size:
            self._chk(self._L.hm_coherence_batch(self._h, _ptr(ti), _ptr(tj), _ptr(tw), _ptr(ts), ti.numel(), S.shape[1],
                                                 _f(c), _ptr(out), self._stream()))
            host.copy_(out, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.device))
        return (host, ev, (ti, tj, tw, ts, out))           # (the operands stay referenced until the kernel has run)

    @staticmethod
    def coherence_distances_end(handle) -> np.ndarray:
        host, ev, _keep = handle
        ev.synchronize()
        return host.numpy()

    def project_table(self, table: torch.Tensor, n_rows: int, c: float) -> None:
        """``project_to_hyperboloid`` over rows [0, n_rows) of ``table`` in place + image refresh
        (enhanced_fast_hyperbolic_merge.py:784-792)."""
        t = self._check_table(table)
        if n_rows > t.shape[0]:
            raise ValueError("n_rows outside the table")
        if _nan_c(c):                              # sqrt(1 + nan * r^2): every time coordinate NaN; images rebuilt from the table
            t[: int(n_rows), 0] = float("nan")
            self.set_table(t, self.n)
            return
        self._chk(self._L.hm_project_table(self._h, _ptr(t), t.stride(0), int(n_rows), _f(c), self._stream()))

    def rows_pair_distance(self, table: torch.Tensor, A, B, c: float) -> np.ndarray:
        """distance(table[A[t]], table[B[t]]) on ANY rows of the caller's table (not only live image rows)."""
        if _nan_c(c):
            return np.full(len(A), np.nan, np.float32)
        t = table.detach()
        ia = torch.as_tensor(np.ascontiguousarray(A, dtype=np.int64), device=t.device)
        ib = torch.as_tensor(np.ascontiguousarray(B, dtype=np.int64), device=t.device)
        out = device_rows_op("distance", t.index_select(0, ia), t.index_select(0, ib), _f(c), self.sign_mode)
        return out.cpu().numpy()

    def row_vs_all(self, row: int, n: int, c: float) -> np.ndarray:
        if _nan_c(c):
            return np.full(int(n), np.nan, np.float32)
        out = torch.empty(int(n), dtype=torch.float32, device=self.device)
        self._chk(self._L.hm_row_vs_all(self._h, int(row), int(n), _f(c), _ptr(out), self._stream()))
        return out.cpu().numpy()

    def scan_stats(self) -> dict:
        ms, pairs, emitted, passes = C.c_float(0), C.c_int64(0), C.c_int64(0), C.c_int32(0)
        self._L.hm_last_scan_stats(self._h, C.byref(ms), C.byref(pairs), C.byref(emitted), C.byref(passes))
        return {"scan_ms": float(ms.value), "pairs": int(pairs.value), "emitted": int(emitted.value),
                "passes": int(passes.value)}

    def scan_totals(self, reset: bool = False) -> dict:
        """Summed event-timed duration / pairs / launches of the pair-scan kernel (bench roofline)."""
        ms, pairs, launches = C.c_double(0), C.c_int64(0), C.c_int64(0)
        self._L.hm_scan_totals(self._h, C.byref(ms), C.byref(pairs), C.byref(launches), 1 if reset else 0)
        return {"scan_ms": float(ms.value), "pairs": int(pairs.value), "launches": int(launches.value)}


# ----------------------------------------------------------------------------------------------
# engine-independent device functions (embedding/lorentz_model.py surface)
# ----------------------------------------------------------------------------------------------
def _stream_of(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _require_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if not t.is_cuda:
            raise HypMergeUnavailable("hyptokenizer_amd kernels run on a HIP device only (tensor is on "
                                      f"{t.device}); there is no CPU fallback")


def _rows2d(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    t = t.reshape(-1, t.shape[-1])
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


def device_batch_distance(x: torch.Tensor, y: torch.Tensor, c: float, sign_mode: int) -> torch.Tensor:
    _require_cuda(x, y)
    L = _lib.load()
    xx, yy = _rows2d(x), _rows2d(y)
    out = torch.empty((xx.shape[0], yy.shape[0]), dtype=torch.float32, device=xx.device)
    with torch.cuda.device(xx.device):
        _lib.check(L.hm_batch_distance(_ptr(xx), xx.shape[0], _ptr(yy), yy.shape[0], xx.stride(0) if xx.shape[0] > 1 else xx.shape[1],
                                       yy.stride(0) if yy.shape[0] > 1 else yy.shape[1], xx.shape[1], _f(c), int(sign_mode),
                                       _ptr(out), _stream_of(xx)))
    return out


def _broadcast_rows(x: torch.Tensor, y: torch.Tensor):
    shape = torch.broadcast_shapes(x.shape, y.shape)
    xb = x.expand(shape).reshape(-1, shape[-1]).contiguous().float()
    yb = y.expand(shape).reshape(-1, shape[-1]).contiguous().float()
    return xb, yb, shape


def device_rows_op(op: str, x: torch.Tensor, y: Optional[torch.Tensor], c: float, sign_mode: int) -> torch.Tensor:
    """Row-wise Lorentz primitive on broadcast operands; result shaped like the reference's."""
    L = _lib.load()
    if y is not None:
        _require_cuda(x, y)
        xb, yb, shape = _broadcast_rows(x.detach(), y.detach())
    else:
        _require_cuda(x)
        shape = x.shape
        xb = x.detach().reshape(-1, shape[-1]).contiguous().float()
        yb = None
    b, d1 = xb.shape
    s = _stream_of(xb)
    with torch.cuda.device(xb.device):
        if op in ("minkowski", "distance"):
            out = torch.empty(b, dtype=torch.float32, device=xb.device)
            if op == "minkowski":
                _lib.check(L.hm_rows_minkowski(_ptr(xb), _ptr(yb), b, d1, d1, int(sign_mode), _ptr(out), s))
            else:
                _lib.check(L.hm_rows_distance(_ptr(xb), _ptr(yb), b, d1, d1, _f(c), int(sign_mode), _ptr(out), s))
            return out.reshape(shape[:-1])
        out = torch.empty((b, d1), dtype=torch.float32, device=xb.device)
        if op == "log_map":
            _lib.check(L.hm_rows_log_map(_ptr(xb), _ptr(yb), b, d1, d1, int(sign_mode), _ptr(out), d1, s))
        elif op == "exp_map":
            _lib.check(L.hm_rows_exp_map(_ptr(xb), _ptr(yb), b, d1, d1, _ptr(out), d1, s))
        elif op == "project":
            _lib.check(L.hm_rows_project(_ptr(xb), b, d1, d1, _f(c), _ptr(out), d1, s))
        else:
            raise ValueError(op)
    return out.reshape(shape)
