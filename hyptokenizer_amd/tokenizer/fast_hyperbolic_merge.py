"""FastHyperbolicTokenizer on the MI355X merge engine.

Class surface of the reference's ``tokenizer/fast_hyperbolic_merge.py``: ``MergeCandidate``,
``AdaptiveMergeCache``, ``FastHyperbolicTokenizer`` with the same constructor kwargs, attributes,
cache stepping and threshold dynamics.  The reference's recompute branch -- ``batch_distance`` over
the whole table, a Python loop over ``nonzero()``, a full sort, and the FAISS/HNSW sampled search
above 10 000 tokens (``:274-374``) -- is replaced by ONE exact GPU search that returns the ordered
``cache_size`` best candidates and the exact candidate count (``MergeEngine.topk``).

Cache semantics kept exactly: a refresh stores ``S[:max_size]`` WITHOUT
removing ``S[0]``, returns ``S``; later steps pop 100 entries and merge the first of them; cached
entries are never invalidated.  Only ``candidates[0]``, ``len(candidates)`` and emptiness are ever
consumed by the loop (``:511-549``).
"""
from __future__ import annotations

import logging
import random
import time
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .hyperbolic_merge import TQDM_OFF, HyperbolicTokenizer, _loop_without_cyclic_gc

FAISS_AVAILABLE = False     # replaced entirely by the exact GPU search

logger = logging.getLogger(__name__)


@dataclass
class MergeCandidate:
    """``(distance, token_i, token_j)``; ordering by distance only (reference ``:52-60``)."""
    distance: float
    token_i: int
    token_j: int

    def __lt__(self, other):
        return self.distance < other.distance


class CandidateList(Sequence):
    """The sorted candidate list ``S`` of one refresh.

    ``len()`` is the exact number of candidates; only the first ``cache_size`` entries (everything
    the reference's cache ever keeps, ``:91-95``) are stored.  Indexing past them raises
    ``IndexError``."""

    def __init__(self, d: np.ndarray, i: np.ndarray, j: np.ndarray, total: int, counter=None):
        self._d, self._i, self._j = d, i, j
        self._total = int(total)          # -1: at least len(d) candidates, not counted yet
        self._counter = counter           # () -> exact total (one counting pass of the engine), used on demand

    def __len__(self) -> int:
        if self._total < 0:
            self._total = int(self._counter())
        return self._total

    @property
    def stored(self) -> int:
        return len(self._d)

    def __getitem__(self, k):
        if isinstance(k, slice):
            idx = range(*k.indices(self.stored))
            return [self[q] for q in idx]
        if k < 0:
            k += len(self)
        if not 0 <= k < self.stored:
            raise IndexError("only the first cache_size candidates of a refresh are materialised")
        return MergeCandidate(float(self._d[k]), int(self._i[k]), int(self._j[k]))

    def __iter__(self):
        for k in range(self.stored):
            yield self[k]

    def __bool__(self) -> bool:
        return self._total != 0


class AdaptiveMergeCache:
    """Sorted candidate cache (reference ``:63-133``): ``add_batch`` = concat + stable sort +
    truncate to ``max_size``; ``get_best(n)`` pops the first n.  ``hit_count`` only records pairs
    that were actually served (the reference also stores a zero for every candidate ever seen,
    which has no effect on ``get_stats``).

    A refresh hands over up to ``max_size`` sorted candidates as arrays (``CandidateList``); they
    stay arrays and become ``MergeCandidate`` objects only when popped or when ``candidates`` is
    read, so a refresh does not build 10 000 Python objects that are mostly never looked at."""

    def __init__(self, max_size: int = 10000):
        self.max_size = max_size
        self._list: List[MergeCandidate] = []     # materialised entries (always in front of the arrays)
        self._arr = None                          # (d, i, j) arrays of not-yet-materialised entries
        self._pos = 0
        self._hit_count: Dict[Tuple[int, int], int] = {}
        self._served: List[Tuple[np.ndarray, np.ndarray, int, int]] = []   # (i, j, lo, hi): batches served straight from the arrays
        self.miss_count: int = 0
        self._hits = 0                            # running sum(hit_count.values())

    # -- the reference's public attributes ------------------------------------------------------
    @property
    def hit_count(self) -> Dict[Tuple[int, int], int]:
        """served pairs -> times served; batches popped as arrays are folded in when this is read (or when more than
        ``_SERVED_MAX`` of them have piled up: each entry pins a whole refresh's arrays)"""
        self._fold_served()
        return self._hit_count

    _SERVED_MAX = 64

    def _fold_served(self) -> None:
        for ii, jj, lo, hi in self._served:
            for key in zip(ii[lo:hi].tolist(), jj[lo:hi].tolist()):
                self._hit_count[key] = self._hit_count.get(key, 0) + 1
        self._served = []

    def _note_served(self, ii, jj, lo: int, hi: int) -> None:
        self._served.append((ii, jj, lo, hi))
        self._hits += hi - lo
        if len(self._served) > self._SERVED_MAX:
            self._fold_served()

    @hit_count.setter
    def hit_count(self, value) -> None:
        self._hit_count = dict(value)
        self._served = []

    @property
    def candidates(self) -> List[MergeCandidate]:
        self._materialise_all()
        return self._list

    @candidates.setter
    def candidates(self, value) -> None:
        self._list = list(value)
        self._arr = None
        self._pos = 0

    def _pending(self) -> int:
        return 0 if self._arr is None else len(self._arr[0]) - self._pos

    def __len__(self) -> int:
        return len(self._list) + self._pending()

    def _take(self, n: int) -> List[MergeCandidate]:
        """materialise the next n array entries"""
        if self._arr is None or n <= 0:
            return []
        d, i, j = self._arr
        hi = min(self._pos + n, len(d))
        out = [MergeCandidate(float(a), int(b), int(c))
               for a, b, c in zip(d[self._pos:hi].tolist(), i[self._pos:hi].tolist(), j[self._pos:hi].tolist())]
        self._pos = hi
        if self._pos >= len(d):
            self._arr, self._pos = None, 0
        return out

    def _materialise_all(self) -> None:
        if self._arr is not None:
            self._list = self._list + self._take(self._pending())

    def add_batch(self, new_candidates) -> None:
        if isinstance(new_candidates, CandidateList) and len(self) == 0:
            k = min(self.max_size, new_candidates.stored)
            self._list = []
            self._arr = (new_candidates._d[:k], new_candidates._i[:k], new_candidates._j[:k]) if k else None
            self._pos = 0
            return
        fresh = new_candidates[: self.max_size] if isinstance(new_candidates, CandidateList) else list(new_candidates)
        merged = self.candidates + fresh
        merged.sort()                       # stable, by distance only
        self.candidates = merged[: self.max_size]

    def get_best(self, n: int = 1):
        """Pop the first n entries.  Served straight from the refresh's arrays when nothing has been
        materialised: the batch is then a ``CandidateList`` (same ``len`` / indexing / iteration as the
        reference's list) -- the merge loop only ever looks at entry 0 of the 100 it pops."""
        if len(self) == 0:
            self.miss_count += 1
            return []
        if not self._list and self._arr is not None:
            d, i, j = self._arr
            lo = self._pos
            hi = min(lo + n, len(d))
            self._pos = hi
            if hi >= len(d):
                self._arr, self._pos = None, 0
            self._note_served(i, j, lo, hi)
            return CandidateList(d[lo:hi], i[lo:hi], j[lo:hi], hi - lo)
        best = self._list[:n]
        self._list = self._list[n:]
        if len(best) < n:
            best = best + self._take(n - len(best))
        hc = self.hit_count
        for cand in best:
            key = (cand.token_i, cand.token_j)
            hc[key] = hc.get(key, 0) + 1
        self._hits += len(best)
        return best

    def get_stats(self) -> Dict[str, Any]:
        hits = self._hits
        return {
            "size": len(self),
            "max_size": self.max_size,
            "hit_count": hits,
            "miss_count": self.miss_count,
            "hit_ratio": hits / (hits + self.miss_count + 1e-10),
        }


_PAIR_SAMPLER_OK = None


def _sample_pairs(n: int, count: int):
    """``count`` draws of ``random.sample(range(n), 2)`` (reference ``:448-449``) as two index lists, consuming the
    module-level generator exactly as those calls do.  For n > 21 CPython's ``sample`` is two ``_randbelow(n)`` draws
    with a redraw while the second repeats the first; calling ``_randbelow`` directly skips ``sample``'s per-call
    set-up (3 us -> 1 us per pair).  The shortcut is checked once against ``random.sample`` itself -- values and final
    generator state -- and is not used if the interpreter's ``sample`` behaves differently."""
    global _PAIR_SAMPLER_OK
    inst = getattr(random, "_inst", None)
    fast_possible = n > 21 and inst is not None and hasattr(inst, "_randbelow")
    if fast_possible and _PAIR_SAMPLER_OK is None:
        state = random.getstate()
        want = [tuple(random.sample(range(1000), 2)) for _ in range(64)] + [tuple(random.sample(range(23), 2)) for _ in range(64)]
        after = random.getstate()
        random.setstate(state)
        got = []
        for m in (1000,) * 64 + (23,) * 64:
            a = inst._randbelow(m)
            b = inst._randbelow(m)
            while b == a:
                b = inst._randbelow(m)
            got.append((a, b))
        _PAIR_SAMPLER_OK = (got == want and random.getstate() == after)
        random.setstate(state)
    ii, jj = [], []
    if fast_possible and _PAIR_SAMPLER_OK:
        rb = inst._randbelow
        for _ in range(count):
            a = rb(n)
            b = rb(n)
            while b == a:
                b = rb(n)
            ii.append(a)
            jj.append(b)
    else:
        for _ in range(count):
            a, b = random.sample(range(n), 2)
            ii.append(a)
            jj.append(b)
    return ii, jj


class FastHyperbolicTokenizer(HyperbolicTokenizer):
    """Merge loop with a candidate cache; one exact GPU search per ~101 steps."""

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
        cache_size: int = 10000,
        rebuild_frequency: int = 100,
        hnsw_m: int = 32,
        hnsw_ef_construction: int = 200,
        hnsw_ef_search: int = 100,
        *,
        sign_convention: str = "reference",
        engine=None,
        shard=None,
    ):
        super().__init__(vocab=vocab, embeddings=embeddings, curvature=curvature, merge_threshold=merge_threshold,
                         lr=lr, device=device, max_vocab_size=max_vocab_size,
                         use_approximate_search=use_approximate_search, sign_convention=sign_convention,
                         engine=engine, shard=shard)
        self.index = None
        self.index_outdated = True
        self.cache = AdaptiveMergeCache(max_size=cache_size)
        self.rebuild_frequency = rebuild_frequency
        self.merges_since_rebuild = 0
        # HNSW knobs are kept as attributes for compatibility; no index exists on this path
        self.hnsw_m = hnsw_m
        self.hnsw_ef_construction = hnsw_ef_construction
        self.hnsw_ef_search = hnsw_ef_search
        self.lazy_count = True        # refreshes do not count every candidate; len(candidates) counts on demand
        self.batch_merges = True      # the merges between two refreshes go to the engine as one launch
        self._refreshed = None        # the CandidateList of this step's refresh (None on a cache pop)
        # When a refresh returns, every merge up to the NEXT refresh is known and issued (``_plan_merges``), so the table
        # of the next refresh exists on the device long before the loop's string bookkeeping gets there: its search is
        # enqueued right behind the merges (``topk_refresh_begin``) and collected when the loop arrives -- used only if
        # the loop arrives in exactly the state it was started for.
        self.prefetch_refresh = True
        self._prefetch = None         # key (engine id, rows, threshold, curvature, k) of a refresh in flight
        self._prefetched = None       # (key, (d, i, j) | None) of a finished one

    def _build_faiss_index(self) -> None:
        """Reference ``:195-240``.  The HNSW index is replaced by the exact search: nothing to build."""
        self.use_approximate_search = False
        self.index = None

    def _finish_prefetch(self) -> None:
        """Nothing else may be asked of the engine while a prefetched refresh is in flight: collect it first."""
        key = getattr(self, "_prefetch", None)
        if key is not None:
            self._prefetch = None
            self._prefetched = (key, self._engine.topk_refresh_end())

    def _get_engine(self):
        self._finish_prefetch()
        return super()._get_engine()

    def _cancel_plan(self) -> None:
        if self._plan is not None:        # rows issued ahead of time are about to be dropped: so is a refresh made with them
            self._finish_prefetch()
            self._prefetched = None
        super()._cancel_plan()

    def _find_merge_candidates(self) -> List[Tuple[int, int, float]]:
        """Base-class tuple format (reference ``:242-251``); holds the materialised candidates."""
        found = self._find_merge_candidates_fast()
        return [(c.token_i, c.token_j, c.distance) for c in found]

    def _find_merge_candidates_fast(self):
        """Cache pop, else one exact search + cache refill (reference ``:253-376``).

        The refresh asks the engine for the ordered ``cache_size`` best candidates only; ``len()`` of the
        returned list -- which the reference consumes in two log lines (``:521,526``) -- is the exact
        number of candidates and is counted on demand (rows are only ever appended, so the pairs of the
        table as it was at the refresh can be counted at any later time)."""
        cached = self.cache.get_best(100)
        self._refreshed = None
        if cached:
            return cached
        self._cancel_plan()
        eng = self._get_engine()
        c, thr, n0 = self.curvature, self._search_threshold(), self.current_vocab_size
        if self.shard is not None:
            from ..sharding import sharded_topk
            d, i, j, total = sharded_topk(eng, self.shard, c, thr, self.cache.max_size)
            found = CandidateList(d, i, j, total)
        elif self.lazy_count and hasattr(eng, "count_candidates"):
            got, self._prefetched = self._prefetched, None          # (_get_engine above has collected a refresh in flight)
            if got is not None and got[1] is not None and got[0] == (id(eng), n0, thr, float(c), self.cache.max_size):
                d, i, j = got[1]
                total = -1
            else:
                d, i, j, total = eng.topk(c, thr, self.cache.max_size, count=False)

            def count_all():
                self._finish_prefetch()
                return eng.count_candidates(c, thr, n0)
            found = CandidateList(d, i, j, total, counter=count_all)
        else:
            d, i, j, total = eng.topk(c, thr, self.cache.max_size)
            found = CandidateList(d, i, j, total)
        self.cache.add_batch(found)
        self._refreshed = found
        return found

    def _fast_forward(self, step: int, steps: int, log_every: int, adaptive_threshold: bool) -> int:
        """Replays loop steps ``step, step + 1, ...`` while they are plain cache pops whose merges were issued ahead
        of time (``_plan_merges``) and nothing is logged, sampled or rescaled in them (reference ``:511-576``: a cache
        pop of 100, ``candidates[0]``, ``_merge_tokens``) -- all of them at once, with list operations instead of a
        Python statement per step.  Returns the number of steps done (0: take the long way)."""
        cache, plan = self.cache, self._plan
        if plan is None or cache._list or cache._arr is None:
            return 0
        d_arr, i_arr, j_arr = cache._arr
        pos, n_arr, p = cache._pos, len(d_arr), plan.pos
        n = self.current_vocab_size
        if pos >= n_arr or not plan.matches(int(i_arr[pos]), int(j_arr[pos]), n):
            return 0
        # how many steps until something other than pop + merge happens
        k = min(len(plan.i) - p, (n_arr - pos + 99) // 100, steps - step)
        if step % log_every == 0:
            return 0
        k = min(k, log_every - 1 - step % log_every)            # step s with (s + 1) % log_every == 0 logs a line
        if adaptive_threshold:
            k = min(k, 1000 - step % 1000 if step % 1000 else 0)  # step s > 0 with s % 1000 == 0 rescales the threshold
        if k <= 0:
            return 0
        A, B = plan.i[p:p + k], plan.j[p:p + k]
        vocab = self.vocab
        lefts = [vocab[a] for a in A]
        rights = [vocab[b] for b in B]
        merged = [x + y for x, y in zip(lefts, rights)]
        vocab.extend(merged)
        self.token2idx.update(zip(merged, range(n, n + k)))
        self.merge_history.extend(zip(lefts, rights, merged))
        hi = min(pos + 100 * k, n_arr)
        cache._note_served(i_arr, j_arr, pos, hi)
        if hi >= n_arr:
            cache._arr, cache._pos = None, 0
        else:
            cache._pos = hi
        plan.pos = p + k
        if plan.pos >= len(plan.i):
            self._plan = None
        self.current_vocab_size = n + k
        self.merges_since_rebuild += k
        if self.merges_since_rebuild >= self.rebuild_frequency:
            self.index_outdated = True
        self._engine_key = self._table_key()
        return k

    def _token_lengths(self) -> np.ndarray:
        """len(vocab[r]) for every row, kept as an array and extended as tokens are appended"""
        lens = getattr(self, "_lens", None)
        n = len(self.vocab)
        if lens is None or len(lens) > n:
            lens = np.fromiter(map(len, self.vocab), np.int64, n)
        elif len(lens) < n:
            lens = np.concatenate([lens, np.fromiter(map(len, self.vocab[len(lens):]), np.int64, n - len(lens))])
        self._lens = lens
        return lens

    def _plan_merges(self, found: "CandidateList", steps_left: int) -> None:
        """Every merge up to the next refresh is known when a refresh returns (this
        step merges ``S[0]``, the following ones pop 100 cached entries each and merge the first of them,
        ``S[0], S[100], S[200], ...``; cached entries are never invalidated).  All of them are issued to the
        engine as ONE launch; the loop's ``_merge_tokens`` calls then find their rows already written and
        only do the string bookkeeping.  Capped by the steps the loop has left and by the table size."""
        cls = type(self)
        if (not self.batch_merges or self.shard is not None or self._plan is not None
                or cls._find_merge_candidates_fast is not FastHyperbolicTokenizer._find_merge_candidates_fast
                or cls._merge_tokens is not FastHyperbolicTokenizer._merge_tokens
                or {"_merge_tokens", "_find_merge_candidates_fast", "_append_token"} & set(self.__dict__)):
            return
        eng = self._get_engine()
        stored = min(found.stored, self.cache.max_size)
        if stored == 0 or not hasattr(eng, "merge_append_batch"):
            return
        n = self.current_vocab_size
        count = min(1 + (stored + 99) // 100, steps_left, self.max_vocab_size - n)
        if count < 2:
            return
        picks = np.arange(-100, 100 * (count - 1), 100)
        picks[0] = 0                                              # [0, 0, 100, 200, ...]
        ii = found._i[picks]
        jj = found._j[picks]
        lens = self._token_lengths()
        li, lj = lens[ii], lens[jj]
        w = lj / (li + lj)
        eng.merge_append_batch(ii, jj, w.astype(np.float32), self.curvature, self.embeddings.data, n,
                               independent=bool(max(int(ii.max()), int(jj.max())) < n))
        from .hyperbolic_merge import _MergePlan
        self._plan = _MergePlan(ii.tolist(), jj.tolist(), n)
        if (self.prefetch_refresh and self.lazy_count and self._prefetch is None and hasattr(eng, "topk_refresh_begin")
                and count == 1 + (stored + 99) // 100 and steps_left > count):
            # the loop refreshes next when these `count` merges are done: at n + count rows, same threshold
            c = float(self.curvature)
            thr = self._search_threshold(n + count)
            if eng.topk_refresh_begin(c, thr, self.cache.max_size):
                self._prefetch = (id(eng), n + count, thr, c, self.cache.max_size)

    def _merge_tokens(self, i: int, j: int) -> None:
        super()._merge_tokens(i, j)
        self.merges_since_rebuild += 1
        if self.merges_since_rebuild >= self.rebuild_frequency:
            self.index_outdated = True

    def _evaluate_merge_quality_batch(self, candidates: List[MergeCandidate], text_sample: List[str]) -> List[float]:
        """Reference ``:394-431`` (never called by the loop): frequency x length balance / distance."""
        scores = []
        for cand in candidates[:100]:
            a, b = self.vocab[cand.token_i], self.vocab[cand.token_j]
            freq = sum(1 for text in text_sample if a + b in text)
            balance = 1.0 / (1.0 + abs(len(a) - len(b)))
            scores.append(freq * balance / (cand.distance + 1e-6))
        return scores

    def _compute_distance_statistics(self, sample_size: int = 1000) -> Dict[str, float]:
        """Distances of ``random.sample(range(n), 2)`` pairs (reference ``:433-465``).  The Python
        RNG is consumed in the reference's order; the distances are one batched kernel call."""
        n = self.current_vocab_size
        count = min(sample_size, n * (n - 1) // 2)
        ii, jj = _sample_pairs(n, count)
        if not ii:
            return {"min": 0.0, "max": 0.0, "mean": 0.0, "std": 0.0}
        dists = [float(v) for v in self._get_engine().pair_distance(ii, jj, self.curvature)]
        return {"min": min(dists), "max": max(dists), "mean": np.mean(dists), "std": np.std(dists)}

    @_loop_without_cyclic_gc
    def optimize_merges(self, steps: int = 10000, log_every: int = 1000, text_sample: Optional[List[str]] = None,
                        adaptive_threshold: bool = True) -> None:
        """Reference ``:467-576`` step for step (threshold rewrites, statistics calls and their RNG
        consumption, empty-step handling, x1.1 every 1000 steps)."""
        from tqdm import tqdm

        bar = tqdm(total=steps, desc="Optimizing merges", disable=TQDM_OFF)
        empty_steps = 0
        stats = {"step": [], "vocab_size": [], "min_dist": [], "max_dist": [], "mean_dist": [], "num_candidates": []}

        if adaptive_threshold:
            ds = self._compute_distance_statistics()
            logger.info(f"Initial distance statistics: min={ds['min']:.6f}, max={ds['max']:.6f}, mean={ds['mean']:.6f}")
            if ds["max"] < 1e-6:
                logger.warning("WARNING: Maximum distance is near zero! This will prevent finding merge candidates.")
                logger.warning("Consider reinitializing embeddings with a larger initialization scale.")
                self.merge_threshold = 1e-5
                logger.info(f"Auto-adjusting merge threshold to {self.merge_threshold:.6f}")
            if ds["max"] > 0 and self.merge_threshold > ds["max"]:
                self.merge_threshold = min(self.merge_threshold, ds["mean"] * 1.5)
                logger.info(f"Adjusted initial merge threshold to {self.merge_threshold:.6f}")

        step = -1
        while step + 1 < steps:
            step += 1
            if self._plan:
                # the steps up to the next refresh / log line are fully determined: replay them without the per-step
                # machinery (same pops, same merges, same bookkeeping)
                done = self._fast_forward(step, steps, log_every, adaptive_threshold)
                if done:
                    if not bar.disable:
                        bar.update(done)
                    step += done - 1
                    continue
            t0 = time.time()
            found = self._find_merge_candidates_fast()
            if self._refreshed is not None and found:
                self._plan_merges(self._refreshed, steps - step)

            if step % log_every == 0 or not found:
                ds = self._compute_distance_statistics()
                stats["step"].append(step)
                stats["vocab_size"].append(self.current_vocab_size)
                stats["min_dist"].append(ds["min"])
                stats["max_dist"].append(ds["max"])
                stats["mean_dist"].append(ds["mean"])
                stats["num_candidates"].append(len(found))
                logger.info(f"\nStep {step}: vocab_size={self.current_vocab_size}")
                logger.info(f"  Distance stats: min={ds['min']:.6f}, max={ds['max']:.6f}, mean={ds['mean']:.6f}")
                logger.info(f"  Merge candidates: {len(found)}")
                logger.info(f"  Merge threshold: {self.merge_threshold:.6f}")

            if not found:
                empty_steps += 1
                if empty_steps > 5 and adaptive_threshold:
                    self.merge_threshold *= 1.5
                    logger.info(f"No candidates found. Increasing threshold to {self.merge_threshold:.6f}")
                    empty_steps = 0
                    continue
                elif empty_steps > 10:
                    logger.info(f"No more merge candidates found after {step} steps")
                    break
                continue
            empty_steps = 0

            best = found[0]
            self._merge_tokens(best.token_i, best.token_j)

            if not bar.disable:                   # display only (tqdm formats the postfix even when disabled)
                bar.update(1)
                elapsed = time.time() - t0
                cs = self.cache.get_stats()
                bar.set_postfix({"vocab_size": self.current_vocab_size, "best_dist": best.distance,
                                 "threshold": self.merge_threshold, "time": f"{elapsed:.2f}s",
                                 "hit_ratio": f"{cs['hit_ratio']:.2f}"})
            if (step + 1) % log_every == 0:
                logger.info(f"Step {step+1}: merged '{self.vocab[best.token_i]}' + "
                            f"'{self.vocab[best.token_j]}' -> '{self.vocab[-1]}' (dist: {best.distance:.4f})")
            if adaptive_threshold and step > 0 and step % 1000 == 0:
                self.merge_threshold *= 1.1
        if not bar.disable:
            bar.close()
        self.last_run_stats = stats     # the reference builds this dict and drops it (:484)
