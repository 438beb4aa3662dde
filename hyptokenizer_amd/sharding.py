"""Row-sharded candidate search across the GPUs of one node.

Every pair (i, j) is independent.  The table is replicated on every rank (<= 40 MB); rank p scans
the strict upper triangle for rows ``i`` in its range ``[r_p, r_{p+1})`` (ranges of equal pair
count), and the only exchange is a tiny all-gather over RCCL (``torch.distributed`` backend
"nccl" on ROCm) of each rank's best record / ordered list.  Every rank then holds the same global
result and applies the merge to its own replica (deterministic kernels => identical replicas), so
no row is ever broadcast.  The reference has nothing to compare with here: it is single-process.

With the ``gloo`` backend the same code runs on CPU tensors (tests use it with the oracle-backed
engine double).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

ROW_ALIGN = 256          # scan kernels work on 256-row blocks: aligned cuts waste no partial block


def partition_rows(n: int, world: int, align: int = ROW_ALIGN) -> List[int]:
    """Boundaries ``b[0]=0 <= ... <= b[world]=n`` such that each range owns ~1/world of the
    ``n(n-1)/2`` pairs (row i has ``n-1-i`` partners): ``b[p] = n (1 - sqrt(1 - p/world))``,
    rounded to a multiple of ``align`` when the table is large enough."""
    if world < 1:
        raise ValueError("world must be >= 1")
    bounds = [0]
    for p in range(1, world):
        x = n * (1.0 - math.sqrt(1.0 - p / world))
        if n >= 4 * align * world:
            x = round(x / align) * align
        b = int(min(max(round(x), bounds[-1]), n))
        bounds.append(b)
    bounds.append(n)
    return bounds


def pairs_in_rows(n: int, r0: int, r1: int) -> int:
    cnt = r1 - r0
    return cnt * (n - 1) - (r0 + r1 - 1) * cnt // 2


def merge_topk_lists_device(allv: torch.Tensor, k: int):
    """k smallest of the ranks' gathered lists by (distance bits, i, j), on the device.  ``allv``: int32
    [world, k + 1, 3]; row 0 of a rank = (entries, count low 31 bits, count high bits), rows 1.. = (bits(d), i, j).
    Three stable sorts (j, then i, then the distance bits -- non-negative floats order like their int32 views); only
    the k winners and the headers travel to the host."""
    world = allv.shape[0]
    hdr = allv[:, 0, :].cpu().numpy()
    total = int(sum(int(h[1]) + (int(h[2]) << 31) for h in hdr))
    ent = allv[:, 1:, :]
    valid = torch.arange(k, device=allv.device)[None, :] < allv[:, 0, 0:1]
    du = torch.where(valid, ent[:, :, 0], torch.full_like(ent[:, :, 0], 0x7FFFFFFF)).reshape(-1)
    ii, jj = ent[:, :, 1].reshape(-1), ent[:, :, 2].reshape(-1)
    order = torch.sort(jj, stable=True).indices
    order = order[torch.sort(ii[order], stable=True).indices]
    order = order[torch.sort(du[order], stable=True).indices]
    keep = order[:min(k, int(hdr[:, 0].sum()))]
    win = torch.stack((du[keep], ii[keep], jj[keep]), dim=1).cpu().numpy()
    assert world >= 1
    return (np.ascontiguousarray(win[:, 0]).view(np.float32), np.ascontiguousarray(win[:, 1]).astype(np.int32),
            np.ascontiguousarray(win[:, 2]).astype(np.int32), total)


class ShardContext:
    """Process-group plumbing for one rank."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, device: Optional[torch.device] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        backend = dist.get_backend(group)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        self.device = torch.device(device)
        # buffers of the per-step record exchange, allocated once (the step is latency-bound)
        self._rec = torch.zeros(4, dtype=torch.int32, device=self.device)
        self._recs = torch.zeros(4 * self.world, dtype=torch.int32, device=self.device)
        self._host = torch.zeros(4 * self.world, dtype=torch.int32)
        if self.device.type == "cuda":
            self._host = self._host.pin_memory()
        self._rec_dev = {}          # engine device -> int32[4] record the engine writes (always device memory)

    def bind_engine(self, engine) -> bool:
        """With an RCCL ("nccl") group and an engine on this rank's GPU, bind a communicator of the group's ranks to the
        engine (once; collective -- every rank reaches this at the same point of the same loop) so that the exchange step
        runs inside the library: ``hm_shard_merge_steps`` / ``hm_global_argmin`` / ``hm_global_topk``.  False: the group is
        a CPU one (tests, ranks sharing a GPU) and the exchange stays with ``torch.distributed`` in Python."""
        if dist.get_backend(self.group) != "nccl" or not hasattr(engine, "comm_init"):
            return False
        edev = getattr(engine, "device", None)
        if edev is None or edev.type != "cuda" or edev != self.device:
            return False
        if engine.comm_info() is None:
            engine.comm_init(self.group)
        return True

    def record_buffer(self, device: torch.device) -> torch.Tensor:
        """The 16-byte record an engine on `device` writes with ``argmin_into`` (allocated once per device)."""
        key = str(device)
        if key not in self._rec_dev:
            self._rec_dev[key] = torch.zeros(4, dtype=torch.int32, device=device)
        return self._rec_dev[key]

    def row_range(self, n: int) -> Tuple[int, int]:
        b = partition_rows(n, self.world)
        return b[self.rank], b[self.rank + 1]

    # -- C1: global nearest pair -------------------------------------------------------------
    def global_argmin(self, local: Optional[Tuple[float, int, int]]) -> Optional[Tuple[float, int, int]]:
        """all-gather of one 16-byte record per rank, lexicographic min over (d bits, i, j)."""
        rec = torch.zeros(4, dtype=torch.int32)
        if local is not None:
            dbits = int(np.float32(local[0]).view(np.uint32))
            rec = torch.tensor([1, dbits if dbits < 2 ** 31 else dbits - 2 ** 32, local[1], local[2]], dtype=torch.int32)
        mine = rec.to(self.device)
        out = torch.empty(4 * self.world, dtype=torch.int32, device=self.device)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        recs = out.cpu().numpy().reshape(self.world, 4)
        best = None
        for found, dbits, i, j in recs.tolist():
            if not found:
                continue
            key = (dbits & 0xFFFFFFFF, i, j)
            if best is None or key < best:
                best = key
        if best is None:
            return None
        return float(np.uint32(best[0]).view(np.float32)), int(best[1]), int(best[2])

    # -- C2: global ordered top-k -------------------------------------------------------------
    def global_topk(self, d: np.ndarray, i: np.ndarray, j: np.ndarray, count: int, k: int):
        """all-gather of the ranks' ordered lists (padded to k) and counts; k-way merge by
        (d bits, i, j).  Returns (d, i, j, total_count), identical on every rank."""
        k = int(k)
        buf = np.zeros((k + 1, 3), np.int32)
        m = len(d)
        buf[0, 0] = m
        buf[0, 1] = count & 0x7FFFFFFF
        buf[0, 2] = count >> 31
        if m:
            buf[1:m + 1, 0] = np.ascontiguousarray(d, np.float32).view(np.int32)
            buf[1:m + 1, 1] = i
            buf[1:m + 1, 2] = j
        mine = torch.from_numpy(buf).to(self.device)
        out = torch.empty((self.world * (k + 1), 3), dtype=torch.int32, device=self.device)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        if self.device.type == "cuda":
            return merge_topk_lists_device(out.view(self.world, k + 1, 3), k)
        allb = out.cpu().numpy().reshape(self.world, k + 1, 3)
        total = 0
        ds, is_, js = [], [], []
        for r in range(self.world):
            mr = int(allb[r, 0, 0])
            total += int(allb[r, 0, 1]) + (int(allb[r, 0, 2]) << 31)
            ds.append(allb[r, 1:mr + 1, 0].view(np.uint32))
            is_.append(allb[r, 1:mr + 1, 1])
            js.append(allb[r, 1:mr + 1, 2])
        du, ii, jj = np.concatenate(ds), np.concatenate(is_), np.concatenate(js)
        order = np.lexsort((jj, ii, du))[:k]
        return du[order].view(np.float32), ii[order].astype(np.int32), jj[order].astype(np.int32), total


def sharded_argmin(engine, ctx: ShardContext, c: float, thr: float):
    """Global nearest pair.  An engine that can leave its record on the device (``argmin_into``) always does: the
    search is asynchronous, armed for the next one of the same range, and costs ONE host synchronisation per step --
    scan -> record in HBM -> all-gather on the same stream -> one 16*world-byte read-back (RCCL, backend "nccl"), or
    record -> host -> all-gather over the CPU backend (gloo: tests, ranks that share a GPU).  The same engine code
    path (armed / seeded searches, overflow reports, row ranges that move as the table grows) either way."""
    if ctx.bind_engine(engine):
        return engine.global_argmin(c, thr)
    r0, r1 = ctx.row_range(engine.n)
    edev = getattr(engine, "device", None)
    if hasattr(engine, "argmin_into") and edev is not None and edev.type == "cuda":
        rec = ctx.record_buffer(edev)
        engine.argmin_into(c, thr, r0, r1, rec)
        if ctx.device.type == "cuda":
            dist.all_gather_into_tensor(ctx._recs, rec if rec.device == ctx.device else rec.to(ctx.device), group=ctx.group)
            ctx._host.copy_(ctx._recs, non_blocking=True)
            torch.cuda.current_stream(ctx.device).synchronize()
        else:
            mine = rec.cpu()                                   # synchronises the engine's stream
            dist.all_gather_into_tensor(ctx._host, mine, group=ctx.group)
        recs = ctx._host.numpy().reshape(ctx.world, 4)
        if not (recs[:, 0] == 2).any():
            return _best_of_records(recs)
        # some rank overflowed its emission buffer (tie flood): every rank takes the bounded host path
    return ctx.global_argmin(engine.argmin(c, thr, r0, r1))


def _best_of_records(recs: np.ndarray):
    best = None
    for found, dbits, i, j in recs.tolist():
        if found != 1:
            continue
        key = (dbits & 0xFFFFFFFF, i, j)
        if best is None or key < best:
            best = key
    if best is None:
        return None
    return float(np.uint32(best[0]).view(np.float32)), int(best[1]), int(best[2])


def sharded_topk(engine, ctx: ShardContext, c: float, thr: float, k: int):
    """Global ordered top-k and exact count.  RCCL group: the ranks' lists never leave the device (``hm_global_topk``);
    CPU group: gathered as numpy arrays through ``torch.distributed``."""
    if ctx.bind_engine(engine):
        return engine.global_topk(c, thr, k)
    r0, r1 = ctx.row_range(engine.n)
    d, i, j, cnt = engine.topk(c, thr, k, r0, r1)
    return ctx.global_topk(d, i, j, cnt, k)


def _all_gather_ragged(ctx: ShardContext, arr: np.ndarray) -> List[np.ndarray]:
    """all-gather of per-rank int32 / float32 arrays of different lengths (first axis): lengths first, then the arrays
    padded to the longest.  Returns the ranks' arrays in rank order."""
    a = np.ascontiguousarray(arr)
    width = int(np.prod(a.shape[1:])) if a.ndim > 1 else 1
    raw = a.reshape(-1).view(np.int32)
    lens = torch.zeros(ctx.world, dtype=torch.int64, device=ctx.device)
    dist.all_gather_into_tensor(lens, torch.tensor([raw.size], dtype=torch.int64, device=ctx.device), group=ctx.group)
    lens = lens.cpu().tolist()
    cap = max(max(lens), 1)
    mine = torch.zeros(cap, dtype=torch.int32, device=ctx.device)
    if raw.size:
        mine[:raw.size] = torch.from_numpy(raw.copy()).to(ctx.device)
    out = torch.empty(ctx.world * cap, dtype=torch.int32, device=ctx.device)
    dist.all_gather_into_tensor(out, mine, group=ctx.group)
    flat = out.cpu().numpy().reshape(ctx.world, cap)
    parts = []
    for r in range(ctx.world):
        p = flat[r, :lens[r]].view(a.dtype)
        parts.append(p.reshape((-1,) + a.shape[1:]) if width > 1 or a.ndim > 1 else p)
    return parts


def sharded_candidates(engine, ctx: ShardContext, c: float, thr: float):
    """Every candidate of the table in row-major order -- (i, j, d, total) as ``engine.candidates`` -- listed by row
    ranges on the ranks and concatenated in rank order (the ranges ascend, each rank's list is row-major)."""
    r0, r1 = ctx.row_range(engine.n)
    i, j, d, _t = engine.candidates(c, thr, r0, r1)
    gi = _all_gather_ragged(ctx, i.astype(np.int32))
    gj = _all_gather_ragged(ctx, j.astype(np.int32))
    gd = _all_gather_ragged(ctx, d.astype(np.float32))
    ii, jj, dd = np.concatenate(gi), np.concatenate(gj), np.concatenate(gd)
    return ii, jj, dd, int(len(ii))


COHERENCE_SHARD_MIN = 2048      # candidates per batch from which the coherence kernel's work is worth splitting over the ranks


def sharded_coherence(engine, ctx: ShardContext, ii: np.ndarray, jj: np.ndarray, w: np.ndarray, samples: np.ndarray, c: float,
                      min_count: Optional[int] = None) -> np.ndarray:
    """``engine.coherence_distances`` with the candidates of a large batch (a refresh scores EVERY candidate) cut into
    ``world`` contiguous slices, one per rank, and the distances all-gathered; small batches are computed by every rank
    for itself (deterministic kernels: identical replicas, and an exchange costs more than the launch)."""
    count = len(ii)
    if count < (COHERENCE_SHARD_MIN if min_count is None else min_count) or ctx.world == 1:
        return engine.coherence_distances(ii, jj, w, samples, c)
    cuts = [count * r // ctx.world for r in range(ctx.world + 1)]
    lo, hi = cuts[ctx.rank], cuts[ctx.rank + 1]
    mine = engine.coherence_distances(ii[lo:hi], jj[lo:hi], w[lo:hi], samples[lo:hi], c) if hi > lo \
        else np.zeros((0, samples.shape[1]), np.float32)
    return np.concatenate(_all_gather_ragged(ctx, np.ascontiguousarray(mine, np.float32)), axis=0)
