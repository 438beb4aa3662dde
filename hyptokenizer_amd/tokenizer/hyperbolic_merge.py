"""HyperbolicTokenizer on the MI355X merge engine.

Same class surface as the reference's ``tokenizer/hyperbolic_merge.py`` (constructor kwargs,
attributes, method names, save/load files); the candidate search and the midpoint update run on the
GPU through ``hyptokenizer_amd.engine.MergeEngine`` instead of ``batch_distance`` + Python loops.
What a step computes is unchanged:

* candidates = pairs ``i < j`` with ``d(i, j) < merge_threshold`` (reference ``:247-269``),
* the merged pair is the first of the list sorted by distance, ties in row-major order (``:378``),
* the new row is ``project(exp_map(x_i, w_j * log_map(x_i, x_j)))`` appended at row ``n``
  (``:326-351``); rows are never removed.

Additive keyword-only arguments: ``sign_convention`` ("reference" = arithmetic as shipped, the
default; "lorentz" = sign-corrected) and ``engine`` (an object with the
``MergeEngine`` interface; tests inject an oracle-backed double, the product never does) and
``shard`` (a ``hyptokenizer_amd.sharding.ShardContext``: the candidate search is row-sharded over
the ranks of a process group and every rank applies the same merge to its replica), and
``incremental`` (maintain the nearest pair across steps instead of re-searching: rows are only ever
appended, so after a merge the global minimum is ``min(previous minimum, nearest partner of the new
row)`` -- one row-vs-all pass per step, same pairs, same distances), and ``prefilter``
("auto" | "f32" | "bf16": the MFMA form of the pair scan; results do not depend on it).

``optimize_merges`` runs its steps in batches ON THE DEVICE (``MergeEngine.std_merge_steps`` /
``incr_merge_steps``: search -> exact re-evaluation -> merge, step after step without a host round
trip; the merge weight only needs token lengths, which live in a device array) and replays the
records on the host afterwards for the token strings -- same pairs, same rows, same log lines.
The FAISS pre-filter of the reference (``:203-244``, ``:593-625``) is replaced by the exact GPU
search and never used: ``FAISS_AVAILABLE`` is always False here.
"""
from __future__ import annotations

import contextlib
import functools
import gc
import json
import logging
import os
from typing import List, Optional, Tuple

import numpy as np
import torch
from tqdm import tqdm

from ..embedding.lorentz_model import batch_distance, distance
from .._lib import LOOP_MAX_STEPS
from ..engine import MAX_ROWS, MAX_WIDTH, HypMergeUnavailable, MergeEngine, sign_mode_id

logger = logging.getLogger(__name__)

# progress bars: off when TQDM_DISABLE is set, and automatically on a non-TTY stderr (disable=None)
TQDM_OFF = True if os.environ.get("TQDM_DISABLE") else None

FAISS_AVAILABLE = False            # the HNSW / Flat index path is replaced by the exact GPU search
TORCH_COMPILE_AVAILABLE = hasattr(torch, "compile")
USING_COMPILED = False             # no tracing compiler on this path: the kernels are hand-written
distance_compiled = distance
batch_distance_compiled = batch_distance


def threshold_for_fp32_compare(thr: float) -> float:
    """``tensor_fp32 < python_float`` compares in fp32 (reference ``:262``): nearest float32."""
    return float(np.float32(thr))


def threshold_for_double_compare(thr: float) -> float:
    """``float(d32) < thr`` in double (reference ``:288``, n <= 100 branch) equals the fp32 test
    ``d32 < t`` with ``t`` the smallest float32 >= thr."""
    t = np.float32(thr)
    if float(t) < thr:
        t = np.nextafter(t, np.float32(np.inf))
    return float(t)


class _MergePlan:
    """Merges issued to the engine ahead of the host loop: merge t = (i[t], j[t]) -> row row0 + t; `pos` = next one."""
    __slots__ = ("i", "j", "row0", "pos")

    def __init__(self, i, j, row0):
        self.i, self.j, self.row0, self.pos = i, j, row0, 0

    def matches(self, i: int, j: int, row: int) -> bool:
        p = self.pos
        return p < len(self.i) and self.i[p] == i and self.j[p] == j and self.row0 + p == row


@contextlib.contextmanager
def _cyclic_gc_paused():
    """The merge loops create strings, tuples of strings and numbers -- nothing that can form a reference cycle --
    but their allocations drive the interpreter's generational collector, and one full sweep of a heap that holds
    torch (about half a million tracked objects after ``import torch``) costs tens of milliseconds: more than a
    hundred loop steps.  The collector is paused for the duration of a loop and restored afterwards."""
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


def _loop_without_cyclic_gc(fn):
    @functools.wraps(fn)
    def run(*args, **kwargs):
        with _cyclic_gc_paused():
            return fn(*args, **kwargs)
    return run


class HyperbolicTokenizer:
    """Tokenizer whose merges are chosen by hyperbolic distance between token embeddings."""

    def __init__(
        self,
        vocab: List[str],
        embeddings: torch.nn.Parameter,
        curvature: float = 1.0,
        merge_threshold: float = 0.1,
        lr: float = 1e-3,
        device: Optional[torch.device] = None,
        max_vocab_size: int = 100000,
        use_approximate_search: bool = True,
        *,
        sign_convention: str = "reference",
        engine=None,
        shard=None,
        incremental: bool = False,
        prefilter: str = "auto",
    ):
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.device = torch.device(device)
        self.vocab = list(vocab)
        self.current_vocab_size = len(vocab)
        self.max_vocab_size = max_vocab_size
        self.curvature = curvature
        self.merge_threshold = merge_threshold
        self.lr = lr
        self.use_approximate_search = bool(use_approximate_search and FAISS_AVAILABLE)
        self.sign_convention = sign_convention
        sign_mode_id(sign_convention)

        width = embeddings.size(1)
        if engine is None and (not 2 <= max_vocab_size <= MAX_ROWS or not 2 <= width <= MAX_WIDTH):
            # the reference accepts any size and fails (or thrashes) later; say it where the object is built
            raise ValueError(f"the merge engine takes tables of 2..{MAX_ROWS} rows and 2..{MAX_WIDTH} columns "
                             f"(max_vocab_size={max_vocab_size}, embedding width={width})")
        table = torch.zeros((max_vocab_size, width), dtype=embeddings.dtype, device=self.device)
        table[: self.current_vocab_size] = embeddings.detach().to(self.device)
        self.embeddings = torch.nn.Parameter(table)

        self.token2idx = {tok: k for k, tok in enumerate(self.vocab)}
        self.merge_history: List[Tuple[str, str, str]] = []
        self.index = None

        self._engine = engine
        self._engine_key = None       # (table identity, version, rows) the engine image was built from
        self.shard = shard            # hyptokenizer_amd.sharding.ShardContext: row-sharded search over ranks
        self.incremental = bool(incremental)
        self.prefilter = prefilter
        self._inc = None              # incremental search state: ((threshold, curvature), rows covered, best (d, i, j) | None)
        self._len_state = None        # (engine id, rows) whose token lengths the engine holds (device-resident loops)
        self._plan = None             # _MergePlan: merges already issued to the engine ahead of the host loop
        self.device_loop = True       # optimize_merges may run its steps in batches on the device

    # ------------------------------------------------------------------------------------------
    # engine plumbing
    # ------------------------------------------------------------------------------------------
    def _get_engine(self):
        """Engine whose image holds rows ``[0, current_vocab_size)`` of ``self.embeddings``."""
        if self._engine is None:
            if self.device.type != "cuda":
                raise HypMergeUnavailable(
                    "HyperbolicTokenizer's candidate search and merge run on a HIP device only "
                    f"(device={self.device}); there is no CPU fallback")
            self._engine = MergeEngine(self.max_vocab_size, self.embeddings.size(1), self.sign_convention,
                                       self.device, prefilter=self.prefilter)
        key = self._table_key()
        if key != self._engine_key:
            self._engine.set_table(self.embeddings.data, self.current_vocab_size)
            self._engine_key = key
            self._inc = None          # the table changed under the maintained minimum
        return self._engine

    def _table_key(self):
        # in-place edits of the Parameter bump _version; edits through ``.data`` do not -- call
        # refresh_engine() after those
        return (self.embeddings.data_ptr(), self.embeddings._version, self.current_vocab_size)

    def refresh_engine(self) -> None:
        """Force the engine to re-read the table (after editing ``embeddings.data`` by hand)."""
        self._engine_key = None

    def _search_threshold(self, n: Optional[int] = None) -> float:
        n = self.current_vocab_size if n is None else n
        return threshold_for_fp32_compare(self.merge_threshold) if n > 100 \
            else threshold_for_double_compare(self.merge_threshold)

    # ------------------------------------------------------------------------------------------
    # reference surface
    # ------------------------------------------------------------------------------------------
    def _compute_pairwise_distances(self) -> torch.Tensor:
        """Full ``[n, n]`` distance matrix (reference ``:166-190``)."""
        n = self.current_vocab_size
        live = self.embeddings.data[:n]
        return batch_distance(live, live, self.curvature, sign_convention=self.sign_convention)

    def _find_merge_candidates(self) -> List[Tuple[int, int, float]]:
        """Every ``(i, j, distance)`` with ``i < j`` and ``distance < merge_threshold``, row-major
        (reference ``:192-291``)."""
        eng = self._get_engine()
        i, j, d, _total = eng.candidates(self.curvature, self._search_threshold())
        return [(int(a), int(b), float(x)) for a, b, x in zip(i.tolist(), j.tolist(), d.tolist())]

    def _best_candidate(self) -> Optional[Tuple[int, int, float]]:
        """``sorted(candidates, key=distance)[0]`` without building the list."""
        if self.incremental:
            hit = self._best_incremental()
        elif self.shard is not None:
            from ..sharding import sharded_argmin
            hit = sharded_argmin(self._get_engine(), self.shard, self.curvature, self._search_threshold())
        else:
            hit = self._get_engine().argmin(self.curvature, self._search_threshold())
        if hit is None:
            return None
        d, i, j = hit
        return i, j, d

    def _best_incremental(self) -> Optional[Tuple[float, int, int]]:
        """The same ``(d, i, j)`` the full search returns, maintained across steps.  The candidate
        set only grows (``_merge_tokens`` appends row ``n`` and removes nothing, reference
        ``:326-351``), so ``min`` over it is a running minimum: one full search when the state is
        new or stale (table edited, threshold changed), then one row-vs-all reduction per appended
        row.  The key order (d, i, j) is the full search's (distance, then row-major)."""
        eng = self._get_engine()                  # drops self._inc when the table was edited
        n = self.current_vocab_size
        thr = (self._search_threshold(), float(self.curvature))     # a distance is acosh(u) / sqrt(c): c is part of the key
        st = self._inc
        if st is None or st[0] != thr or st[1] > n:
            if self.shard is not None:
                from ..sharding import sharded_argmin
                best = sharded_argmin(eng, self.shard, self.curvature, thr[0])
            else:
                best = eng.argmin(self.curvature, thr[0])
            rows = n
        else:
            _, rows, best = st
            for r in range(rows, n):              # every rank does this redundantly: no exchange needed
                cand = eng.row_argmin(r, r, self.curvature, thr[0])
                if cand is not None and (best is None or cand < best):
                    best = cand
            rows = n
        self._inc = (thr, rows, best)
        return best

    def _is_valid_merge(self, token_i: str, token_j: str) -> bool:
        return True               # reference ``:293-307``

    def _merge_weight(self, i: int, j: int) -> float:
        li, lj = len(self.vocab[i]), len(self.vocab[j])
        return lj / (li + lj)

    def _append_token(self, i: int, j: int) -> None:
        """The host half of a merge (reference ``:343-355``): strings, index, history."""
        left, right = self.vocab[i], self.vocab[j]
        merged = left + right
        n = self.current_vocab_size
        self.vocab.append(merged)
        self.token2idx[merged] = n
        self.current_vocab_size = n + 1
        self.merge_history.append((left, right, merged))

    def _merge_tokens(self, i: int, j: int) -> None:
        """Append the merged token and its embedding (reference ``:309-355``)."""
        n = self.current_vocab_size
        if n >= self.max_vocab_size:
            raise ValueError(f"Maximum vocabulary size {self.max_vocab_size} reached. Cannot merge more tokens.")
        plan = self._plan
        if plan is not None:
            if plan.matches(i, j, n):             # issued ahead of time (a batch of merges known in advance): row n is there
                plan.pos += 1
                if plan.pos >= len(plan.i):
                    self._plan = None
                self._append_token(i, j)
                self._engine_key = self._table_key()
                return
            self._cancel_plan()
        eng = self._get_engine()
        eng.merge_append(i, j, self._merge_weight(i, j), self.curvature, self.embeddings.data, n)
        self._append_token(i, j)
        self._engine_key = self._table_key()      # the image already holds row n

    def _cancel_plan(self) -> None:
        """The loop left the path a batch of merges was issued for: drop the rows appended ahead of time."""
        if self._plan is not None:
            self._plan = None
            n = self.current_vocab_size
            self._engine.truncate(n)
            self.embeddings.data[n:].zero_()

    # ------------------------------------------------------------------------------------------
    # device-resident batches of the loop
    # ------------------------------------------------------------------------------------------
    def _device_loop_ok(self) -> bool:
        """Steps may run on the device when nothing the batch would bypass is customised: the search and
        the merge are this class's own (a subclass or an instance attribute that overrides them is
        honoured by falling back to the step-by-step loop), no row-sharding, an engine that has the loops."""
        cls = type(self)
        if not (self.device_loop
                and cls._merge_tokens is HyperbolicTokenizer._merge_tokens
                and cls._best_candidate is HyperbolicTokenizer._best_candidate
                and cls._append_token is HyperbolicTokenizer._append_token
                and not ({"_merge_tokens", "_best_candidate", "_append_token"} & set(self.__dict__))):
            return False
        eng = self._get_engine()
        if self.shard is None:
            return hasattr(eng, "std_merge_steps")
        # row-sharded: per step a search of the rank's rows, an all-gather of the records and the merge from the gathered
        # records on every replica -- all enqueued, one synchronisation per batch (full search every step only)
        return (not self.incremental and hasattr(eng, "shard_merge_step")
                and getattr(eng, "device", None) is not None and eng.device.type == "cuda")

    def _sync_token_lengths(self, eng) -> None:
        n = self.current_vocab_size
        if self._len_state != (id(eng), n):
            eng.set_token_lengths(np.fromiter(map(len, self.vocab[:n]), dtype=np.int32, count=n))
            self._len_state = (id(eng), n)

    def _device_steps(self, count: int):
        """Up to ``count`` loop steps in one engine call -> list of (i, j, distance) merged, and whether the
        loop ran out of candidates.  A step whose search overflowed the engine's emission buffer is run
        through the step-by-step path (bounded rerun) and the batch resumes after it."""
        eng = self._get_engine()
        merged, exhausted = [], False
        while count > 0 and not exhausted:
            room = self.max_vocab_size - self.current_vocab_size
            if room <= 0:
                raise ValueError(f"Maximum vocabulary size {self.max_vocab_size} reached. Cannot merge more tokens.")
            k = min(count, LOOP_MAX_STEPS, room)
            n = self.current_vocab_size
            if n <= 100:
                k = min(k, 101 - n)               # the reference compares differently up to 100 tokens (:270-289): one threshold per batch
            self._sync_token_lengths(eng)
            thr = self._search_threshold()            # of the table as it is now; the same for every step of the batch
            if self.shard is not None:
                recs, done = self._sharded_batch(eng, thr, k)
            elif self.incremental:
                best = self._best_incremental()
                recs, done, best_after = eng.incr_merge_steps(self.curvature, thr, self.embeddings.data, k, best)
            else:
                recs, done = eng.std_merge_steps(self.curvature, thr, self.embeddings.data, k)
            for (_f, d, i, j) in recs[:done]:
                self._append_token(i, j)
                merged.append((i, j, d))
            self._engine_key = self._table_key()
            self._len_state = (id(eng), self.current_vocab_size)
            if self.incremental:
                self._inc = ((thr, float(self.curvature)), self.current_vocab_size, best_after)
            count -= done
            if done < k:
                verdict = recs[done][0]
                if verdict == 2:                      # emission overflow: this one step through the bounded host path
                    best = self._best_candidate()
                    if best is None:
                        exhausted = True
                    else:
                        self._merge_tokens(best[0], best[1])
                        merged.append(best)
                        count -= 1
                else:
                    exhausted = True
        return merged, exhausted

    def _sharded_batch(self, eng, thr: float, k: int):
        """``k`` steps of the row-sharded loop without a host round trip per step: search of this rank's
        rows -> record in device memory -> all-gather of the ranks' records on the same stream (RCCL) -> every rank
        merges the global nearest pair into its replica (``hm_shard_merge_step``).  With a CPU process group (tests,
        ranks sharing one GPU) the gather goes through the host and synchronises per step; the kernels are the same."""
        import torch.distributed as dist
        from ..sharding import partition_rows
        ctx = self.shard
        if ctx.bind_engine(eng):                  # RCCL group: the whole batch is enqueued by the library (hm_shard_merge_steps)
            return eng.shard_merge_steps(self.curvature, thr, self.embeddings.data, k)
        n0 = self.current_vocab_size
        dev = eng.device
        rec = ctx.record_buffer(dev)
        gathered = torch.empty((k, 4 * ctx.world), dtype=torch.int32, device=dev)
        table = self.embeddings.data
        eng.shard_loop_begin()
        try:
            for s in range(k):
                b = partition_rows(n0 + s, ctx.world)
                eng.argmin_into(self.curvature, thr, b[ctx.rank], b[ctx.rank + 1], rec)
                if ctx.device.type == "cuda":
                    dist.all_gather_into_tensor(gathered[s], rec if rec.device == ctx.device else rec.to(ctx.device), group=ctx.group)
                else:
                    host = torch.empty(4 * ctx.world, dtype=torch.int32)
                    dist.all_gather_into_tensor(host, rec.cpu(), group=ctx.group)
                    gathered[s].copy_(host)
                eng.shard_merge_step(gathered[s], ctx.world, self.curvature, table, s)
        finally:
            recs, done = eng.shard_loop_end(k)
        return recs, done

    @_loop_without_cyclic_gc
    def optimize_merges(self, steps: int = 10000, log_every: int = 1000, parallel_eval: bool = True,
                        sample_ratio: float = 1.0) -> None:
        """Greedy merge loop (reference ``:357-412``).  ``parallel_eval`` and ``sample_ratio`` never
        change which pair is merged in the reference (``:381-393``, ``:553-591``) and are accepted
        for compatibility."""
        bar = tqdm(range(steps), desc="Optimizing merges", disable=TQDM_OFF)
        if self._device_loop_ok():
            step = 0
            while step < steps:
                merged, exhausted = self._device_steps(min(LOOP_MAX_STEPS, steps - step))
                base = self.current_vocab_size - len(merged)
                for t, (i, j, dist) in enumerate(merged):
                    if (step + 1) % log_every == 0:
                        logger.info(f"Step {step+1}: merged '{self.vocab[i]}' + '{self.vocab[j]}' -> "
                                    f"'{self.vocab[base + t]}' (dist: {dist:.4f})")
                        logger.info(f"Vocabulary size: {base + t + 1}")
                    step += 1
                if not bar.disable:
                    bar.update(len(merged))
                    if merged:
                        bar.set_postfix({"vocab_size": len(self.vocab), "best_dist": merged[-1][2],
                                         "threshold": self.merge_threshold})
                if exhausted:
                    logger.info("No more merge candidates found. Stopping.")
                    break
            return
        for step in bar:
            best = self._best_candidate()
            if best is None:
                logger.info("No more merge candidates found. Stopping.")
                break
            i, j, dist = best
            self._merge_tokens(i, j)
            if (step + 1) % log_every == 0:
                logger.info(f"Step {step+1}: merged '{self.vocab[i]}' + '{self.vocab[j]}' -> "
                            f"'{self.vocab[-1]}' (dist: {dist:.4f})")
                logger.info(f"Vocabulary size: {len(self.vocab)}")
            if not bar.disable:                   # display only (tqdm formats the postfix even when disabled)
                bar.set_postfix({"vocab_size": len(self.vocab), "best_dist": dist, "threshold": self.merge_threshold})

    def _evaluate_candidates_parallel(self, candidates: List[Tuple[int, int, float]]) -> Tuple[int, int, float]:
        """Reference ``:553-591``: simulate the merges, then return the first candidate unchanged."""
        if candidates:
            eng = self._get_engine()
            ii = [c[0] for c in candidates]
            jj = [c[1] for c in candidates]
            eng.midpoint(ii, jj, [self._merge_weight(a, b) for a, b in zip(ii, jj)], self.curvature)
        return candidates[0]

    def _init_faiss_index(self) -> None:     # reference ``:593-605``; replaced by the GPU search
        self.index = None

    def _update_faiss_index(self) -> None:   # reference ``:607-625``; replaced by the GPU search
        return None

    # ------------------------------------------------------------------------------------------
    # inference (host Python, same semantics as the reference ``:414-471``)
    # ------------------------------------------------------------------------------------------
    def tokenize(self, text: str) -> List[str]:
        if not hasattr(self, "_merge_rules"):
            self._merge_rules = {(a, b): ab for a, b, ab in self.merge_history}
        rules = self._merge_rules
        toks = list(text)
        again = True
        while again:
            again = False
            k = 0
            while k < len(toks) - 1:
                new = rules.get((toks[k], toks[k + 1]))
                if new is None:
                    k += 1
                else:
                    toks[k:k + 2] = [new]
                    again = True
        return toks

    def encode(self, text: str) -> List[int]:
        unk = self.token2idx.get("<unk>", 3)
        return [self.token2idx.get(t, unk) for t in self.tokenize(text)]

    # batch forms on the GPU (hm_tokenize_batch): what scripts/benchmark_efficiency.py:58-94 does line by line
    def _batch_encoder(self):
        if not hasattr(self, "_merge_rules"):       # built once and kept, like the reference's cache (:424-428)
            self._merge_rules = {(a, b): ab for a, b, ab in self.merge_history}
        enc = getattr(self, "_encoder", None)
        if enc is None or enc[0] is not self._merge_rules or enc[1] != len(self.token2idx):
            from .batch_encoder import BatchEncoder
            enc = (self._merge_rules, len(self.token2idx), BatchEncoder(self._merge_rules, self.token2idx, self.device))
            self._encoder = enc
        return enc[2]

    def tokenize_batch(self, texts: List[str]) -> List[List[str]]:
        """``[self.tokenize(t) for t in texts]`` in one kernel launch."""
        return self._batch_encoder().tokenize_batch(texts)

    def encode_batch(self, texts: List[str]) -> List[List[int]]:
        """``[self.encode(t) for t in texts]`` in one kernel launch."""
        return self._batch_encoder().encode_batch(texts)

    def decode(self, indices: List[int]) -> str:
        return "".join(self.vocab[k] for k in indices)

    # ------------------------------------------------------------------------------------------
    # persistence: same four files and keys as the reference (``:473-551``)
    # ------------------------------------------------------------------------------------------
    def save(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "vocab.json"), "w") as f:
            json.dump(self.vocab, f)
        # the reference saves a view of the whole pre-allocated table (2.4 MB for 50 rows);
        # the same live rows are stored compactly here and load identically
        live = self.embeddings[: self.current_vocab_size].detach().cpu().clone()
        torch.save(live, os.path.join(path, "embeddings.pt"))
        with open(os.path.join(path, "merges.json"), "w") as f:
            json.dump(self.merge_history, f)
        config = {
            "curvature": self.curvature if isinstance(self.curvature, (int, float)) else float(self.curvature),
            "merge_threshold": self.merge_threshold,
            "embedding_dim": self.embeddings.size(1) - 1,
            "max_vocab_size": self.max_vocab_size,
            "use_approximate_search": self.use_approximate_search,
        }
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(config, f)

    @classmethod
    def load(cls, path: str, device: Optional[torch.device] = None, **kwargs) -> "HyperbolicTokenizer":
        with open(os.path.join(path, "vocab.json"), "r") as f:
            vocab = json.load(f)
        rows = torch.load(os.path.join(path, "embeddings.pt"), map_location="cpu", weights_only=True)
        with open(os.path.join(path, "config.json"), "r") as f:
            config = json.load(f)
        tok = cls(
            vocab=vocab,
            embeddings=torch.nn.Parameter(rows),
            curvature=config["curvature"],
            merge_threshold=config["merge_threshold"],
            device=device,
            max_vocab_size=config.get("max_vocab_size", 100000),
            use_approximate_search=config.get("use_approximate_search", True),
            **kwargs,
        )
        with open(os.path.join(path, "merges.json"), "r") as f:
            tok.merge_history = json.load(f)
        tok.current_vocab_size = len(tok.vocab)
        return tok
