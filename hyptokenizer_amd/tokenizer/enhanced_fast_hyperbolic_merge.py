"""EnhancedFastHyperbolicTokenizer on the MI355X merge engine (BASELINE config 5).

Class surface of the reference's ``tokenizer/enhanced_fast_hyperbolic_merge.py`` (which does not even
import as shipped): the ``FastHyperbolicTokenizer`` loop plus a per-candidate score

    ``alpha / (1 + d)  +  beta * frequency  +  gamma * coherence  (+ compression, morphology)``

(reference ``_score_candidate`` ``:903-990``), hierarchical phases with their own thresholds
(``:514-530``, ``:1056-1066``) and an adaptive curvature that is re-optimised every
``optimize_curvature_freq`` steps, after which the WHOLE table is re-projected (``:753-792``).

What runs where
* **GPU (one fused kernel, ``hm_coherence_batch``)** -- the numeric part of the score: for every
  candidate the simulated merged embedding ``exp_map(x_i, w_j * log_map(x_i, x_j))`` (not projected,
  ``:313-321``) and its distances to <= 50 sampled rows (``:323-333``).  The reference spends
  ~2.5 ms per candidate here (50 ``distance().item()`` calls); a refresh scores EVERY candidate.
* **GPU (``hm_project_table``)** -- ``_project_embeddings`` (``:784-792``) over the whole table in
  place, scan images and norm bounds rebuilt in the same pass.
* **host Python, reference order kept** -- everything that consumes an RNG or touches strings:
  ``torch.randperm`` per candidate (same call order, so the same samples), the frequency table,
  compression and morphology heuristics, phase logic, threshold dynamics, statistics sampling with
  ``random.sample``, the sort by combined score (Python's stable ``list.sort`` on the same keys).

Adaptive curvature: the reference's step raises at ``loss.backward()`` (``:774``; ``distance``
re-wraps ``c`` with ``torch.tensor`` and detaches it, F8), so there is nothing to be identical to.
Here the two losses are evaluated with the reference's sampling (same RNG calls in the same order),
all distances in one batched kernel call, and the gradient is analytic (every distance is
``acosh(u)/sqrt(c)``, so ``dd/dc = -d / (2c)``); Adam, the clamp to [0.1, 10] and the re-projection
follow the reference.  tests/golden/g5_enhanced_lorentz.* pin this against the reference's own loss
code run under a one-line patch that keeps ``c`` attached -- parity with the reference AS SHIPPED is
unpinned (it raises).
"""
from __future__ import annotations

import json
import logging
import os
import random
import re
from collections import Counter
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Set, Tuple, Union

import numpy as np
import torch
from tqdm import tqdm

from .fast_hyperbolic_merge import CandidateList, FastHyperbolicTokenizer, MergeCandidate
from .hyperbolic_merge import TQDM_OFF, _loop_without_cyclic_gc

try:  # the reference consults WordNet when nltk is installed (:39-47); it is optional here too
    import nltk  # noqa: F401
    from nltk.corpus import wordnet
    NLTK_AVAILABLE = True
except ImportError:
    wordnet = None
    NLTK_AVAILABLE = False

logger = logging.getLogger(__name__)

PHASE_THRESHOLDS = {1: 0.05, 2: 0.1}          # reference :525-530; any later phase: 0.2
PREFIXES = {"re", "un", "in", "im", "il", "ir", "dis", "en", "em", "non", "de", "pre", "pro", "mis"}
SUFFIXES = {"ing", "ed", "er", "est", "ly", "ity", "ment", "ness", "able", "ible", "al", "ial"}
COHERENCE_SAMPLES = 50                        # reference :324


@dataclass
class EnhancedMergeCandidate(MergeCandidate):
    """``MergeCandidate`` + the component scores; ordered by ``combined_score`` (the negated
    combined score, so that an ascending sort puts the best first; reference ``:52-63``)."""
    frequency_score: float = 0.0
    semantic_score: float = 0.0
    compression_score: float = 0.0
    morphology_score: float = 0.0
    combined_score: float = 0.0

    def __lt__(self, other):
        return self.combined_score < other.combined_score


_RANDPERM_HELPER_OK = None


def _randperm_helper_ok() -> bool:
    """One-time check of the library's MT19937 helper against THIS torch build's ``torch.randperm`` -- prefix values and
    the generator state left behind, at a small n and at n = 100 003 (the helper follows randperm's 32-bit draw and
    Fisher-Yates order; a build that changes either would otherwise sample other coherence rows than the reference).
    On a mismatch the helper is not used again: plain ``torch.randperm`` calls."""
    global _RANDPERM_HELPER_OK
    if _RANDPERM_HELPER_OK is None:
        _RANDPERM_HELPER_OK = True                    # (so that the calls below go through the helper)
        keep = torch.get_rng_state()
        ok = True
        try:
            for n, ns, count in ((37, 5, 3), (100003, COHERENCE_SAMPLES, 2)):
                torch.manual_seed(1234567 + n)
                want = np.stack([torch.randperm(n)[:ns].numpy() for _ in range(count)]).astype(np.int32)
                st_want = torch.get_rng_state()
                torch.manual_seed(1234567 + n)
                got = randperm_prefixes(n, ns, count)
                ok = ok and np.array_equal(got, want) and torch.equal(torch.get_rng_state(), st_want)
        except Exception:
            ok = False
        finally:
            torch.set_rng_state(keep)
        _RANDPERM_HELPER_OK = bool(ok)
        if not ok:
            logger.warning("hm_randperm_prefix disagrees with this torch build's randperm: using torch.randperm")
    return _RANDPERM_HELPER_OK


def randperm_prefixes(n: int, ns: int, count: int) -> np.ndarray:
    """``[torch.randperm(n)[:ns] for _ in range(count)]`` as an int32 ``[count, ns]`` array, consuming torch's
    CPU generator exactly as those calls do.  Served by the library's host helper ``hm_randperm_prefix`` (the
    generator's MT19937 state is read from and written back to ``torch.get/set_rng_state``); plain
    ``torch.randperm`` calls when that is not possible (n >= 2^32 / 20, an unexpected state layout)."""
    out = np.empty((count, ns), np.int32)
    if count == 0 or ns == 0:
        for _ in range(count):
            torch.randperm(n)
        return out
    st = torch.get_rng_state()
    if st.numel() == 5056 and n < (2 ** 32 - 1) // 20 and ns <= 4096 and _randperm_helper_ok():
        from .. import _lib
        import ctypes as C
        raw = st.numpy().copy()
        # CPUGeneratorImplStateLegacy: u64 seed | i32 left | i32 seeded | u64 next | u64 state[624] | ...
        left = raw[8:12].view(np.int32)
        seeded = raw[12:16].view(np.int32)
        nxt = raw[16:24].view(np.uint64)
        words = raw[24:24 + 624 * 8].view(np.uint64)
        if seeded[0] == 1 and 1 <= left[0] <= 624:
            mt = words.astype(np.uint32)
            c_left, c_next = C.c_int32(int(left[0])), C.c_uint32(int(nxt[0]))
            rc = _lib.load().hm_randperm_prefix(C.c_void_p(mt.ctypes.data), C.byref(c_left), C.byref(c_next), int(n), int(ns),
                                                int(count), C.c_void_p(out.ctypes.data))
            if rc == 0:
                words[:] = mt
                left[0] = c_left.value
                nxt[0] = c_next.value
                torch.set_rng_state(torch.from_numpy(raw))
                return out
    for t in range(count):
        out[t] = torch.randperm(n)[:ns].numpy()
    return out


def _row_means(dist: np.ndarray, keep: np.ndarray) -> np.ndarray:
    """``np.mean`` of the kept entries of every row in float64 -- the value ``np.mean(list_of_floats)``
    gives the reference (``:340``), same summation order: full rows go through one reduction over the
    contiguous axis (numpy's pairwise sum, as for a 1-D array); rows with skipped samples one by one."""
    d64 = dist.astype(np.float64)
    out = np.empty(d64.shape[0], np.float64)
    full = keep.all(axis=1)
    if full.any():
        out[full] = np.add.reduce(np.ascontiguousarray(d64[full]), axis=1) / d64.shape[1]
    for r in np.nonzero(~full)[0].tolist():
        vals = d64[r][keep[r]]
        out[r] = np.mean(vals) if vals.size else np.nan       # empty: the caller returns 0.0 (:335-336)
    return out


class EnhancedFastHyperbolicTokenizer(FastHyperbolicTokenizer):
    """``FastHyperbolicTokenizer`` + frequency / coherence / compression / morphology scoring,
    hierarchical phases and adaptive curvature."""

    def __init__(
        self,
        vocab: List[str],
        embeddings: torch.nn.Parameter,
        curvature: float = 1.0,
        merge_threshold: float = 0.5,
        lr: float = 1e-3,
        device: Optional[torch.device] = None,
        max_vocab_size: int = 100000,
        use_approximate_search: bool = True,
        cache_size: int = 10000,
        rebuild_frequency: int = 100,
        hnsw_m: int = 32,
        hnsw_ef_construction: int = 200,
        hnsw_ef_search: int = 100,
        use_frequency_aware: bool = True,
        use_hierarchical: bool = True,
        use_adaptive_curvature: bool = True,
        use_compression_aware: bool = True,
        corpus_path: Optional[str] = None,
        alpha: float = 0.4,
        beta: float = 0.4,
        gamma: float = 0.2,
        language: str = "english",
        curvature_lr: float = 0.01,
        hierarchy_weight: float = 1.0,
        distortion_weight: float = 0.1,
        optimize_curvature_freq: int = 100,
        corpus_sample: Optional[List[str]] = None,
        compression_weight: float = 0.7,
        distance_weight: float = 0.3,
        sample_size: int = 100,
        *,
        sign_convention: str = "reference",
        engine=None,
        shard=None,
    ):
        super().__init__(vocab=vocab, embeddings=embeddings, curvature=curvature, merge_threshold=merge_threshold,
                         lr=lr, device=device, max_vocab_size=max_vocab_size,
                         use_approximate_search=use_approximate_search, cache_size=cache_size,
                         rebuild_frequency=rebuild_frequency, hnsw_m=hnsw_m,
                         hnsw_ef_construction=hnsw_ef_construction, hnsw_ef_search=hnsw_ef_search,
                         sign_convention=sign_convention, engine=engine, shard=shard)
        self.lazy_count = False       # a refresh scores EVERY candidate: the exact total is needed at once
        self.batch_merges = False     # which pair is merged depends on the scores, not on the cache order
        self.use_frequency_aware = use_frequency_aware
        self.use_hierarchical = use_hierarchical
        self.use_adaptive_curvature = use_adaptive_curvature
        self.use_compression_aware = use_compression_aware
        self.current_phase = 1

        if use_frequency_aware:
            self.alpha, self.beta, self.gamma = alpha, beta, gamma
            self.pair_frequencies: Dict[Tuple[str, str], int] = {}
            if corpus_path and os.path.exists(corpus_path):
                self._compute_pair_frequencies(corpus_path)

        if use_hierarchical:
            self.language = language
            self.token_frequencies: Dict[str, int] = {}
            self.common_morphemes: Set[str] = set()
            self.common_words: Set[str] = set()
            if corpus_path and os.path.exists(corpus_path):
                self._compute_corpus_statistics(corpus_path)

        if use_adaptive_curvature:
            self.static_curvature = curvature
            # a scalar the host optimises: kept on the CPU (the kernels take it by value)
            self.curvature = torch.nn.Parameter(torch.tensor(float(curvature), dtype=torch.float32))
            self.curvature_lr = curvature_lr
            self.curvature_optimizer = torch.optim.Adam([self.curvature], lr=curvature_lr)
            self.hierarchy_weight = hierarchy_weight
            self.distortion_weight = distortion_weight
            self.optimize_curvature_freq = optimize_curvature_freq
            self.merge_pairs: List[Tuple[int, int]] = []
            self._project_embeddings()              # reference :243-244 (the whole table, unused rows too)

        if use_compression_aware:
            self.compression_weight = compression_weight
            self.distance_weight = distance_weight
            self.sample_size = sample_size
            self.corpus_sample = corpus_sample or []
            self.tokenize_cache: Dict[str, Any] = {}

        logger.info(f"Initialized EnhancedFastHyperbolicTokenizer with features: frequency={use_frequency_aware}, "
                    f"hierarchical={use_hierarchical}, adaptive_curvature={use_adaptive_curvature}, "
                    f"compression={use_compression_aware}")

    # ------------------------------------------------------------------------------------------
    # curvature access
    # ------------------------------------------------------------------------------------------
    def get_curvature(self) -> Union[float, torch.Tensor]:
        """Reference ``:374-384``."""
        if self.use_adaptive_curvature:
            return self.curvature
        return getattr(self, "static_curvature", self.curvature)

    def _c(self) -> float:
        c = self.get_curvature()
        return float(c.detach()) if isinstance(c, torch.Tensor) else float(c)

    # ------------------------------------------------------------------------------------------
    # frequency-aware scoring
    # ------------------------------------------------------------------------------------------
    def _compute_pair_frequencies(self, corpus_path: str) -> None:
        """Adjacent-token pair counts over a corpus file (reference ``:266-289``)."""
        if not self.use_frequency_aware:
            return
        logger.info("Computing pair frequencies from corpus...")
        seen = 0
        with open(corpus_path, "r", encoding="utf-8") as f:
            for line in tqdm(f, desc="Computing frequencies", disable=TQDM_OFF):
                toks = self.tokenize(line.strip())
                for pair in zip(toks, toks[1:]):
                    self.pair_frequencies[pair] = self.pair_frequencies.get(pair, 0) + 1
                    seen += 1
        logger.info(f"Computed frequencies for {len(self.pair_frequencies)} unique token pairs "
                    f"from {seen} total pairs")

    def _frequency_scores(self, ii: np.ndarray, jj: np.ndarray) -> np.ndarray:
        """``log1p(freq(ti, tj)) / log1p(max freq)`` per candidate (reference ``:348-372``)."""
        out = np.zeros(len(ii), np.float64)
        if not self.use_frequency_aware or not self.pair_frequencies:
            return out
        pf = self.pair_frequencies
        # max over the table (the reference recomputes it for every candidate, :368): once per table state here --
        # re-derived whenever the dict object or its size changes and at the start of every optimize_merges call
        key = (id(pf), len(pf))
        cached = getattr(self, "_freq_top", None)
        if cached is None or cached[0] != key:
            cached = (key, max(pf.values()))
            self._freq_top = cached
        top = cached[1]
        if not top > 0:
            return out
        counts = np.fromiter((pf.get((self.vocab[a], self.vocab[b]), 0) for a, b in zip(ii.tolist(), jj.tolist())),
                             dtype=np.float64, count=len(ii))
        return np.log1p(counts) / np.log1p(top)

    def _compute_frequency_score(self, i: int, j: int) -> float:
        return float(self._frequency_scores(np.array([i]), np.array([j]))[0])

    def _coherence_samples(self, count: int) -> np.ndarray:
        """``torch.randperm(n)[:50]`` once per candidate, in candidate order (reference ``:324-325``:
        the torch CPU generator is consumed exactly as there)."""
        n = self.current_vocab_size
        return randperm_prefixes(n, min(COHERENCE_SAMPLES, n), count)

    def _semantic_coherence_batch(self, ii: np.ndarray, jj: np.ndarray) -> np.ndarray:
        """Reference ``_compute_semantic_coherence`` (``:291-346``) for a list of candidates: the RNG
        calls happen first, in list order; midpoints and all ``count x 50`` distances are ONE kernel
        launch; mean / sigmoid in float64 as ``np.mean`` / ``np.exp`` give them."""
        count = len(ii)
        if not self.use_frequency_aware or count == 0:
            return np.zeros(count, np.float64)
        lens = self._token_lengths()                      # len(vocab[r]), kept as an array and extended as tokens are appended
        li, lj = lens[ii], lens[jj]
        w = (lj / (li + lj)).astype(np.float64)           # weight_j, a Python double in the reference (:316)
        eng = self._get_engine()
        if self.shard is not None:
            from ..sharding import sharded_coherence
            samples = self._coherence_samples(count)          # (every rank draws the same samples: same seeded generator)
            dist = sharded_coherence(eng, self.shard, ii, jj, w.astype(np.float32), samples, self._c())
        elif count >= 64 and hasattr(eng, "coherence_distances_begin"):
            # two halves, samples drawn in candidate order as ever: while the first half's kernel and result copy run, the
            # host draws the second half's permutations (the long pole: one MT19937 pass over n draws per candidate)
            h = count // 2
            s0 = self._coherence_samples(h)
            pend = eng.coherence_distances_begin(ii[:h], jj[:h], w[:h].astype(np.float32), s0, self._c())
            s1 = self._coherence_samples(count - h)
            d1 = eng.coherence_distances(ii[h:], jj[h:], w[h:].astype(np.float32), s1, self._c())
            samples = np.concatenate([s0, s1])
            dist = np.concatenate([eng.coherence_distances_end(pend), d1])
        else:
            samples = self._coherence_samples(count)
            dist = eng.coherence_distances(ii, jj, w.astype(np.float32), samples, self._c())
        keep = (samples != ii[:, None]) & (samples != jj[:, None])
        avg = _row_means(dist, keep)
        with np.errstate(over="ignore", invalid="ignore"):
            coh = 1.0 / (1.0 + np.exp(avg - self.merge_threshold))
        coh[~keep.any(axis=1)] = 0.0
        return coh

    def _compute_semantic_coherence(self, i: int, j: int) -> float:
        return float(self._semantic_coherence_batch(np.array([i], np.int32), np.array([j], np.int32))[0])

    # ------------------------------------------------------------------------------------------
    # hierarchical strategy (host string logic, reference :388-633)
    # ------------------------------------------------------------------------------------------
    def _compute_corpus_statistics(self, corpus_path: str) -> None:
        """Word counts, 2..5-gram counts, and the frequent ones of each (reference ``:388-437``)."""
        if not self.use_hierarchical:
            return
        logger.info("Computing corpus statistics for hierarchical merging...")
        words, grams = Counter(), Counter()
        with open(corpus_path, "r", encoding="utf-8") as f:
            for line in tqdm(f, desc="Analyzing corpus", disable=TQDM_OFF):
                found = re.findall(r"\b\w+\b", line.lower())
                words.update(found)
                for word in found:
                    for n in range(2, min(6, len(word) + 1)):
                        grams.update(word[k:k + n] for k in range(len(word) - n + 1))
        self.token_frequencies = dict(words)
        gram_cut = np.percentile(list(grams.values()), 80)
        self.common_morphemes = {g for g, cnt in grams.items() if cnt >= gram_cut}
        word_cut = np.percentile(list(words.values()), 70)
        self.common_words = {w for w, cnt in words.items() if cnt >= word_cut}
        logger.info(f"Identified {len(self.common_morphemes)} common morphemes and "
                    f"{len(self.common_words)} common words")

    def _is_potential_morpheme(self, token: str) -> bool:
        """Reference ``:439-483``."""
        if not self.use_hierarchical or token in self.common_morphemes:
            return True
        if NLTK_AVAILABLE:
            if token in PREFIXES or token in SUFFIXES:
                return True
            if len(token) > 2 and any(wordnet.synsets(token, pos=p)
                                      for p in (wordnet.NOUN, wordnet.VERB, wordnet.ADJ, wordnet.ADV)):
                return True
        if 2 <= len(token) <= 5 and sum(1 for w in self.common_words if token in w) >= 5:
            return True
        return False

    def _is_valid_word(self, token: str) -> bool:
        """Reference ``:485-512``."""
        if not self.use_hierarchical or token in self.common_words:
            return True
        if NLTK_AVAILABLE and wordnet.synsets(token):
            return True
        return len(token) >= 3 and re.search(r"[aeiou]", token) is not None

    def _get_merge_phase_threshold(self) -> float:
        if not self.use_hierarchical:
            return self.merge_threshold
        return PHASE_THRESHOLDS.get(self.current_phase, 0.2)

    def _morphology_score(self, i: int, j: int) -> float:
        """Phase-dependent score of ``_score_candidate`` (reference ``:934-943``)."""
        a, b = self.vocab[i], self.vocab[j]
        if self.current_phase == 1:
            return 0.8 if len(a) <= 2 and len(b) <= 2 else 0.2
        if self.current_phase == 2:
            return 0.9 if self._is_potential_morpheme(a + b) else 0.3
        return 1.0 if self._is_valid_word(a + b) else 0.4

    def _filter_by_current_phase(self, candidates: List[MergeCandidate]) -> List[MergeCandidate]:
        """Reference ``:532-633`` (never called by its loop): phase-dependent distance discount."""
        if not self.use_hierarchical or not candidates:
            return candidates
        discount = {1: 0.9, 2: 0.8}.get(self.current_phase, 0.7)
        hit, miss = {1: (0.8, 0.2), 2: (0.9, 0.3)}.get(self.current_phase, (1.0, 0.4))
        out = []
        for cand in candidates:
            score = self._morphology_score(cand.token_i, cand.token_j)
            good = score == hit
            out.append(EnhancedMergeCandidate(distance=cand.distance * discount if good else cand.distance,
                                              token_i=cand.token_i, token_j=cand.token_j,
                                              morphology_score=hit if good else miss))
        return out

    # ------------------------------------------------------------------------------------------
    # compression-aware scoring (host string logic, reference :813-899)
    # ------------------------------------------------------------------------------------------
    def _tokenize_with_vocab(self, text: str, vocab: List[str]) -> List[str]:
        """Greedy longest match over ``vocab``, single characters where nothing matches (reference
        ``:813-847``; the longest matching entry is found by length instead of by scanning a sorted
        copy of the vocabulary at every position -- same tokens)."""
        if not self.use_compression_aware:
            return self.tokenize(text)
        entries = set(vocab)
        longest = max((len(t) for t in entries), default=1)
        out, k = [], 0
        while k < len(text):
            for width in range(min(longest, len(text) - k), 0, -1):
                piece = text[k:k + width]
                if piece in entries:
                    break
            else:
                piece = text[k]
            if not piece:
                piece = text[k]
            out.append(piece)
            k += len(piece)
        return out

    def _compute_compression_score(self, i: int, j: int) -> float:
        """Reference ``:849-899`` including its cache keys (``"original"``, ``merge_{i}_{j}_{text[:20]}``)."""
        if not self.use_compression_aware or not self.corpus_sample:
            return 0.0
        cache = self.tokenize_cache
        if "original" not in cache:
            cache["original"] = sum(len(self.tokenize(text)) for text in self.corpus_sample)
        before = cache["original"]
        trial_vocab = None
        after = 0
        used = min(len(self.corpus_sample), 10)
        for text in self.corpus_sample[:used]:
            key = f"merge_{i}_{j}_{text[:20]}"
            if key not in cache:
                if trial_vocab is None:
                    trial_vocab = self.vocab + [self.vocab[i] + self.vocab[j]]
                cache[key] = len(self._tokenize_with_vocab(text, trial_vocab))
            after += cache[key]
        if used < len(self.corpus_sample):
            after = after * (len(self.corpus_sample) / used)
        ratio = 1.0 if after == 0 else before / after
        return max(0.0, min(1.0, (ratio - 1.0) / 1.0))

    # ------------------------------------------------------------------------------------------
    # combined score
    # ------------------------------------------------------------------------------------------
    def _score_weights(self) -> Tuple[float, float, float, float, float]:
        """(alpha, beta, gamma, compression weight, morphology weight) after the successive scalings of
        reference ``:945-968`` (same multiplications in the same order)."""
        alpha, beta, gamma = 0.7, 0.0, 0.0
        if self.use_frequency_aware:
            alpha, beta, gamma = self.alpha, self.beta, self.gamma
        cw = 0.0
        if self.use_compression_aware:
            cw = self.compression_weight
            alpha *= (1 - cw)
            beta *= (1 - cw)
            gamma *= (1 - cw)
        mw = 0.0
        if self.use_hierarchical:
            mw = 0.3
            alpha *= (1 - mw)
            beta *= (1 - mw)
            gamma *= (1 - mw)
            if self.use_compression_aware:
                cw *= (1 - mw)
        return alpha, beta, gamma, cw, mw

    def _score_candidates(self, dd, ii, jj) -> List[EnhancedMergeCandidate]:
        """``[_score_candidate(c) for c in candidates]`` (reference ``:1008``) with the device work of
        all candidates in one launch.  ``dd`` are the candidates' distances as Python floats of fp32."""
        ii = np.ascontiguousarray(ii, np.int32)
        jj = np.ascontiguousarray(jj, np.int32)
        dist = np.asarray(dd, np.float64)
        count = len(ii)
        freq = sem = None
        if self.use_frequency_aware:
            freq = self._frequency_scores(ii, jj)
            sem = self._semantic_coherence_batch(ii, jj)
        pairs = list(zip(ii.tolist(), jj.tolist()))
        comp = [self._compute_compression_score(a, b) for a, b in pairs] if self.use_compression_aware \
            else [0.0] * count
        morph = [self._morphology_score(a, b) for a, b in pairs] if self.use_hierarchical else [0.0] * count
        alpha, beta, gamma, cw, mw = self._score_weights()
        fz = np.zeros(count) if freq is None else freq
        sz = np.zeros(count) if sem is None else sem
        # left to right as the reference writes it: ((((a*ds + b*f) + g*s) + cw*c) + mw*m)
        total = alpha * (1.0 / (1.0 + dist))
        total = total + beta * fz
        total = total + gamma * sz
        total = total + cw * np.asarray(comp, np.float64)
        total = total + mw * np.asarray(morph, np.float64)
        return [EnhancedMergeCandidate(distance=d, token_i=a, token_j=b, frequency_score=f, semantic_score=s,
                                       compression_score=c, morphology_score=m, combined_score=-t)
                for d, (a, b), f, s, c, m, t in zip(dist.tolist(), pairs, fz.tolist(), sz.tolist(), comp, morph,
                                                    total.tolist())]

    def _score_candidate(self, candidate: MergeCandidate) -> EnhancedMergeCandidate:
        """Reference ``:903-990`` for one candidate."""
        return self._score_candidates([candidate.distance], [candidate.token_i], [candidate.token_j])[0]

    def _find_merge_candidates_fast(self):
        """Parent's candidates (cache pop, else one exact search), every one of them scored, sorted by
        combined score (reference ``:992-1013``).  A refresh scores ALL candidates below the threshold,
        not only the ``cache_size`` the cache keeps -- when there are more, the full list is fetched."""
        basic = super()._find_merge_candidates_fast()
        if not (self.use_frequency_aware or self.use_hierarchical or self.use_compression_aware
                or self.use_adaptive_curvature):
            return basic
        if isinstance(basic, CandidateList):
            if len(basic) > basic.stored:
                if self.shard is not None:
                    from ..sharding import sharded_candidates
                    i, j, d, _total = sharded_candidates(self._get_engine(), self.shard, self._c(), self._search_threshold())
                else:
                    i, j, d, _total = self._get_engine().candidates(self._c(), self._search_threshold())
                order = np.argsort(d, kind="stable")          # row-major list, stable by distance = S
                dd, ii, jj = d[order], i[order], j[order]
            else:
                dd, ii, jj = basic._d, basic._i, basic._j
            dd = [float(x) for x in np.asarray(dd, np.float32).tolist()]
        else:
            dd = [c.distance for c in basic]
            ii = [c.token_i for c in basic]
            jj = [c.token_j for c in basic]
        if len(ii) == 0:
            return []
        scored = self._score_candidates(dd, ii, jj)
        scored.sort()
        return scored

    # ------------------------------------------------------------------------------------------
    # adaptive curvature
    # ------------------------------------------------------------------------------------------
    def _curvature_terms(self, embeddings: torch.Tensor):
        """Both losses of the curvature step and their derivatives in c.

        Sampling exactly as the reference (``:655-688``, ``:720-733``): per used merge pair one
        ``torch.randperm(rows)[:10]`` minus the pair itself; then ``min(500, ...)`` draws of
        ``torch.randint(0, rows, (2,))``.  ``rows = len(embeddings)`` is the whole pre-allocated table,
        as there.  All distances come from one batched kernel call on the table rows."""
        rows = len(embeddings)
        c = self._c()
        groups = []                       # (i, j, sample indices)
        A, B = [], []
        used_pairs = 0
        if self.use_adaptive_curvature and getattr(self, "merge_pairs", None):
            used_pairs = min(len(self.merge_pairs), 100)
            for (i, j) in self.merge_pairs[-used_pairs:]:
                if i >= rows or j >= rows:
                    continue
                take = min(10, rows - 2)
                smp = torch.from_numpy(randperm_prefixes(rows, take, 1)[0].astype(np.int64))      # = torch.randperm(rows)[:take], same generator state after
                smp = smp[~torch.isin(smp, torch.tensor([i, j]))].tolist()
                if not smp:
                    continue
                groups.append((i, j, smp))
                A.append(i); B.append(j)
                for k in smp:
                    A.append(i); B.append(k)
                for k in smp:
                    A.append(j); B.append(k)
        n_h = len(A)
        draws = min(500, rows * (rows - 1) // 2)
        for _ in range(draws):
            a, b = torch.randint(0, rows, (2,)).tolist()
            if a != b:
                A.append(a); B.append(b)
        dist = np.zeros(0, np.float32)
        if A:
            dist = np.asarray(self._get_engine().rows_pair_distance(embeddings, A, B, c), np.float32)
        f32 = np.float32
        inv2c = -1.0 / (2.0 * c)
        # hierarchy preservation (:690-702): relu(pair - other + 0.1).mean() for both ends, / (2 * pairs)
        H, dH = f32(0.0), 0.0
        pos = 0
        for (_i, _j, smp) in groups:
            k = len(smp)
            pd = dist[pos]
            oi = dist[pos + 1:pos + 1 + k]
            oj = dist[pos + 1 + k:pos + 1 + 2 * k]
            pos += 1 + 2 * k
            for other in (oi, oj):
                gap = (pd - other) + f32(0.1)
                act = gap > 0
                H = f32(H + f32(np.where(act, gap, f32(0)).astype(f32).sum(dtype=f32) / f32(k)))
                # d/dc of (pd - other) = -(pd - other) / (2c) on the active entries
                dH += float(np.sum((pd - other)[act].astype(np.float64))) * inv2c / k
        if used_pairs > 0:
            H = f32(H / f32(2 * used_pairs))
            dH = dH / (2 * used_pairs)
        # distortion (:738-751): exp(-10 mean) + 0.1 var (unbiased)
        dd = dist[n_h:]
        if dd.size == 0:
            D, dD = f32(0.0), 0.0
        else:
            mean = dd.mean(dtype=f32)
            var = f32(np.var(dd.astype(np.float64), ddof=1)) if dd.size > 1 else f32(np.nan)
            collapse = f32(np.exp(f32(-10.0) * mean))
            D = f32(collapse + f32(0.1) * var)
            m64, v64 = float(mean), float(var)
            dD = -10.0 * float(collapse) * (m64 * inv2c) + 0.1 * (-v64 / c)
        return H, D, dH, dD

    def _compute_hierarchy_preservation_loss(self, embeddings: torch.Tensor) -> torch.Tensor:
        """Reference ``:637-702`` (value only; consumes the same RNG calls)."""
        if not self.use_adaptive_curvature or not getattr(self, "merge_pairs", None):
            return torch.tensor(0.0)
        state = torch.get_rng_state()
        H, _D, _dH, _dD = self._curvature_terms(embeddings)
        # the distortion draws are not part of this function in the reference: rewind and replay the hierarchy part
        torch.set_rng_state(state)
        rows = len(embeddings)
        used = min(len(self.merge_pairs), 100)
        for (i, j) in self.merge_pairs[-used:]:
            if i < rows and j < rows:
                torch.randperm(rows)
        return torch.tensor(float(H))

    def _compute_distortion_loss(self, embeddings: torch.Tensor) -> torch.Tensor:
        """Reference ``:704-751`` (value only; consumes the same RNG calls)."""
        if not self.use_adaptive_curvature:
            return torch.tensor(0.0)
        saved = getattr(self, "merge_pairs", None)
        self.merge_pairs = []
        try:
            _H, D, _dH, _dD = self._curvature_terms(embeddings)
        finally:
            self.merge_pairs = saved
        return torch.tensor(float(D))

    def _optimize_curvature(self, embeddings: torch.Tensor) -> None:
        """One Adam step on c (reference ``:753-782``), gradient analytic (module docstring)."""
        if not self.use_adaptive_curvature:
            return
        H, D, dH, dD = self._curvature_terms(embeddings)
        loss = float(self.hierarchy_weight) * float(H) + float(self.distortion_weight) * float(D)
        grad = float(self.hierarchy_weight) * dH + float(self.distortion_weight) * dD
        self.curvature_optimizer.zero_grad()
        self.curvature.grad = torch.tensor(grad, dtype=torch.float32)
        self.curvature_optimizer.step()
        with torch.no_grad():
            self.curvature.clamp_(min=0.1, max=10.0)
        self._inc = None
        logger.info(f"Optimized curvature: {self.curvature.item():.4f}, "
                    f"Loss: {loss:.4f} (H: {float(H):.4f}, D: {float(D):.4f})")

    def _project_embeddings(self) -> None:
        """``project_to_hyperboloid`` over the whole table with the current curvature (reference
        ``:784-792``): one in-place pass that also rebuilds the engine's images of the live rows."""
        if not self.use_adaptive_curvature:
            return
        eng = self._get_engine()
        eng.project_table(self.embeddings.data, self.max_vocab_size, self._c())
        self._engine_key = self._table_key()
        self._inc = None

    def _merge_tokens(self, i: int, j: int) -> None:
        """Reference ``:794-809``: remember the pair for the hierarchy loss, then merge."""
        if self.use_adaptive_curvature:
            self.merge_pairs.append((i, j))
        super()._merge_tokens(i, j)

    # ------------------------------------------------------------------------------------------
    # the loop
    # ------------------------------------------------------------------------------------------
    @_loop_without_cyclic_gc
    def optimize_merges(self, steps: int = 10000, log_every: int = 1000, corpus_sample: Optional[List[str]] = None,
                        adaptive_threshold: bool = True,
                        phase_transition_steps: Optional[Dict[int, int]] = None) -> None:
        """Reference ``:1015-1209``."""
        self._freq_top = None            # counts may have been edited since the last call
        if corpus_sample and self.use_compression_aware:
            self.corpus_sample = corpus_sample
            self.tokenize_cache = {}
        if self.use_hierarchical and phase_transition_steps is None:
            phase_transition_steps = {2: 1000, 3: 6000}

        bar = tqdm(range(steps), desc="Optimizing merges", disable=TQDM_OFF)
        misses = 0
        stats: Dict[int, Dict[str, Any]] = {}
        if self.use_hierarchical:
            self.merge_threshold = self._get_merge_phase_threshold()
            logger.info(f"Starting with phase {self.current_phase} threshold: {self.merge_threshold:.4f}")

        for step in bar:
            if self.use_hierarchical and step in phase_transition_steps.values():
                for phase, at in phase_transition_steps.items():
                    if step == at:
                        self.current_phase = phase
                        self.merge_threshold = self._get_merge_phase_threshold()
                        logger.info(f"Transitioning to phase {self.current_phase} with threshold: "
                                    f"{self.merge_threshold:.4f}")
                        if hasattr(self, "tokenize_cache"):
                            self.tokenize_cache = {}

            if self.use_adaptive_curvature and step > 0 and step % self.optimize_curvature_freq == 0:
                self._optimize_curvature(self.embeddings.detach())
                self._project_embeddings()

            if step % log_every == 0 and adaptive_threshold:
                ds = self._sampled_distances()
                if ds:
                    lo, hi, mean = min(ds), max(ds), np.mean(ds)
                    logger.info(f"\nStep {step}: vocab_size={self.current_vocab_size}")
                    logger.info(f"  Distance stats: min={lo:.6f}, max={hi:.6f}, mean={mean:.6f}")
                    logger.info(f"  Merge threshold: {self.merge_threshold:.6f}")
                    stats[step] = {"vocab_size": self.current_vocab_size, "min_dist": lo, "max_dist": hi,
                                   "mean_dist": mean, "phase": self.current_phase if self.use_hierarchical else 0}

            found = self._find_merge_candidates_fast()
            if not found:
                misses += 1
                if misses > 5 and adaptive_threshold:
                    before = self.merge_threshold
                    self.merge_threshold *= 1.5
                    logger.info(f"No candidates found. Increasing threshold from {before:.6f} to "
                                f"{self.merge_threshold:.6f}")
                    misses = 0
                elif misses > 10:
                    logger.info(f"No more merge candidates found after {step} steps")
                    break
                continue
            misses = 0

            best = found[0]
            i, j, dist = best.token_i, best.token_j, best.distance
            if isinstance(best, EnhancedMergeCandidate):
                shown = -best.combined_score
                if step % log_every == 0:
                    logger.info(f"  Best candidate scores: distance={best.distance:.4f}, "
                                f"frequency={best.frequency_score:.4f}, semantic={best.semantic_score:.4f}, "
                                f"compression={best.compression_score:.4f}, morphology={best.morphology_score:.4f}, "
                                f"combined={shown:.4f}")
            else:
                shown = 1.0 / (1.0 + dist)

            self._merge_tokens(i, j)

            if getattr(self, "tokenize_cache", None):
                for key in [k for k in self.tokenize_cache if k.startswith("merge_")]:
                    self.tokenize_cache.pop(key, None)

            if not bar.disable:
                post = {"vocab_size": self.current_vocab_size, "score": f"{shown:.4f}",
                        "threshold": f"{self.merge_threshold:.4f}"}
                if self.use_adaptive_curvature:
                    post["curvature"] = f"{self.curvature.item():.4f}"
                if self.use_hierarchical:
                    post["phase"] = self.current_phase
                bar.set_postfix(post)

            if (step + 1) % log_every == 0:
                logger.info(f"Step {step+1}: merged '{self.vocab[i]}' + '{self.vocab[j]}' -> '{self.vocab[-1]}' "
                            f"(score: {shown:.4f}, phase: {self.current_phase if self.use_hierarchical else 0})")

            if adaptive_threshold and step > 0 and step % 1000 == 0:
                if self.use_hierarchical:
                    self.merge_threshold = self._get_merge_phase_threshold() * (1.1 ** (step // 1000))
                else:
                    self.merge_threshold *= 1.1

        if stats:
            self.training_stats = stats
            logger.info(f"\nCompleted optimization with {len(self.vocab)} tokens")
            logger.info(f"Final merge threshold: {self.merge_threshold:.6f}")
            if self.use_adaptive_curvature:
                logger.info(f"Final curvature: {self.curvature.item():.6f}")
            if self.use_hierarchical:
                logger.info(f"Final phase: {self.current_phase}")

    def _sampled_distances(self) -> List[float]:
        """The statistics sample of the loop (reference ``:1078-1091``): ``random.sample(range(n), 2)`` per
        draw in the reference's order, the distances in one kernel call."""
        n = self.current_vocab_size
        draws = min(1000, n * (n - 1) // 2)
        ii, jj = [], []
        for _ in range(draws):
            a, b = random.sample(range(n), 2)
            ii.append(a)
            jj.append(b)
        if not ii:
            return []
        return [float(v) for v in self._get_engine().pair_distance(ii, jj, self._c())]

    # ------------------------------------------------------------------------------------------
    # persistence (reference :1211-1427): same files and keys
    # ------------------------------------------------------------------------------------------
    def save(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "vocab.json"), "w") as f:
            json.dump(self.vocab, f)
        # the reference stores the whole pre-allocated Parameter here (its own load() then fails on the
        # shape); the live rows are what a loader can use
        torch.save(self.embeddings.data[: self.current_vocab_size].detach().cpu().clone(),
                   os.path.join(path, "embeddings.pt"))
        with open(os.path.join(path, "merges.json"), "w") as f:
            json.dump(self.merge_history, f)
        if self.use_adaptive_curvature:
            torch.save(self.curvature.detach().cpu().clone(), os.path.join(path, "curvature.pt"))
        config = {
            "curvature": self._c(),
            "merge_threshold": self.merge_threshold,
            "max_vocab_size": self.max_vocab_size,
            "use_approximate_search": self.use_approximate_search,
            "use_frequency_aware": self.use_frequency_aware,
            "use_hierarchical": self.use_hierarchical,
            "use_adaptive_curvature": self.use_adaptive_curvature,
            "use_compression_aware": self.use_compression_aware,
            "alpha": getattr(self, "alpha", 0.4),
            "beta": getattr(self, "beta", 0.4),
            "gamma": getattr(self, "gamma", 0.2),
            "language": getattr(self, "language", "english"),
            "hierarchy_weight": getattr(self, "hierarchy_weight", 1.0),
            "distortion_weight": getattr(self, "distortion_weight", 0.1),
            "compression_weight": getattr(self, "compression_weight", 0.7),
            "distance_weight": getattr(self, "distance_weight", 0.3),
            "current_phase": getattr(self, "current_phase", 1),
            "current_vocab_size": self.current_vocab_size,
        }
        with open(os.path.join(path, "enhanced_config.json"), "w") as f:
            json.dump(config, f, indent=2)
        if getattr(self, "training_stats", None):
            with open(os.path.join(path, "training_stats.json"), "w") as f:
                json.dump({str(k): v for k, v in self.training_stats.items()}, f, indent=2)
        if self.use_frequency_aware and getattr(self, "pair_frequencies", None):
            with open(os.path.join(path, "frequencies.json"), "w") as f:
                json.dump({f"{a}|{b}": v for (a, b), v in self.pair_frequencies.items()}, f)
        if self.use_hierarchical:
            with open(os.path.join(path, "hierarchical_data.json"), "w") as f:
                json.dump({"language": getattr(self, "language", "english"),
                           "common_morphemes": list(getattr(self, "common_morphemes", set())),
                           "common_words": list(getattr(self, "common_words", set()))}, f)
        if self.use_adaptive_curvature and hasattr(self, "merge_pairs"):
            torch.save([tuple(p) for p in self.merge_pairs], os.path.join(path, "merge_pairs.pt"))

    @classmethod
    def load(cls, path: str, device: Optional[torch.device] = None, **kwargs) -> "EnhancedFastHyperbolicTokenizer":
        with open(os.path.join(path, "vocab.json"), "r") as f:
            vocab = json.load(f)
        rows = torch.load(os.path.join(path, "embeddings.pt"), map_location="cpu", weights_only=True)
        rows = rows.detach()[: len(vocab)]        # a directory written by the reference holds the whole table
        try:
            with open(os.path.join(path, "enhanced_config.json"), "r") as f:
                config = json.load(f)
        except FileNotFoundError:                 # a base-class directory
            with open(os.path.join(path, "config.json"), "r") as f:
                config = json.load(f)
            config.update({"use_frequency_aware": False, "use_hierarchical": False,
                           "use_adaptive_curvature": False, "use_compression_aware": False})
        tok = cls(
            vocab=vocab, embeddings=torch.nn.Parameter(rows), curvature=config.get("curvature", 1.0),
            merge_threshold=config.get("merge_threshold", 0.1), device=device,
            max_vocab_size=config.get("max_vocab_size", 100000),
            use_approximate_search=config.get("use_approximate_search", True),
            use_frequency_aware=config.get("use_frequency_aware", False),
            use_hierarchical=config.get("use_hierarchical", False),
            use_adaptive_curvature=config.get("use_adaptive_curvature", False),
            use_compression_aware=config.get("use_compression_aware", False),
            alpha=config.get("alpha", 0.4), beta=config.get("beta", 0.4), gamma=config.get("gamma", 0.2),
            language=config.get("language", "english"), hierarchy_weight=config.get("hierarchy_weight", 1.0),
            distortion_weight=config.get("distortion_weight", 0.1),
            compression_weight=config.get("compression_weight", 0.7),
            distance_weight=config.get("distance_weight", 0.3), **kwargs)
        with open(os.path.join(path, "merges.json"), "r") as f:
            tok.merge_history = json.load(f)
        tok.current_phase = config.get("current_phase", 1)
        tok.current_vocab_size = min(config.get("current_vocab_size", len(tok.vocab)), len(tok.vocab))
        if tok.use_adaptive_curvature:
            try:
                value = torch.load(os.path.join(path, "curvature.pt"), map_location="cpu", weights_only=True)
                tok.curvature = torch.nn.Parameter(value.detach().clone().float().reshape(()))
                tok.merge_pairs = [tuple(p) for p in
                                   torch.load(os.path.join(path, "merge_pairs.pt"), map_location="cpu", weights_only=True)]
                tok.curvature_optimizer = torch.optim.Adam([tok.curvature], lr=config.get("curvature_lr", 0.01))
            except FileNotFoundError:
                logger.warning("Could not load adaptive curvature data")
        if tok.use_frequency_aware:
            try:
                with open(os.path.join(path, "frequencies.json"), "r") as f:
                    tok.pair_frequencies = {tuple(k.split("|")): v for k, v in json.load(f).items()}
            except FileNotFoundError:
                logger.warning("Could not load frequency data")
        if tok.use_hierarchical:
            try:
                with open(os.path.join(path, "hierarchical_data.json"), "r") as f:
                    data = json.load(f)
                tok.language = data.get("language", "english")
                tok.common_morphemes = set(data.get("common_morphemes", []))
                tok.common_words = set(data.get("common_words", []))
            except FileNotFoundError:
                logger.warning("Could not load hierarchical data")
        try:
            with open(os.path.join(path, "training_stats.json"), "r") as f:
                tok.training_stats = {int(k): v for k, v in json.load(f).items()}
        except FileNotFoundError:
            pass
        return tok
