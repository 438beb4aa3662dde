#!/usr/bin/env python
"""Headline benchmark of the MI355X merge engine (BASELINE.json metric:
"merge-steps/sec + pairwise Lorentz-dist GB/s, V=50k d=100, 1/2/4/8 GPU").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE merge iteration of ``HyperbolicTokenizer.optimize_merges`` (reference
``tokenizer/hyperbolic_merge.py:357-412``): a full all-pairs Lorentz-distance search for the
nearest pair below the threshold, then the log-map/exp-map midpoint appended as a new row.
Headline workload (BASELINE configs[2], the one the metric is quoted on): V = 50 000 synthetic Lorentz
rows, d = 100, fp32 table, curvature 1, sign-corrected Minkowski form ("lorentz": the mode in which the
search is non-degenerate), threshold 0.5, bf16-MFMA prefilter + canonical fp32
re-evaluation.  Inputs are resident in HBM before the timed region.  With N > 1 ranks the rows of the
pair triangle are sharded over the ranks (equal pair counts), each rank scans its share of the SAME
problem (strong scaling), the best records are all-gathered over RCCL and every rank applies the
merge to its replica.

One JSON line on rank 0.  `roofline` is for the dominant kernel of the timed run (hm_scan_kernel):
achieved = N(N-1)(d+1) algorithmic flops per launch (triangle only; the reference's dense convention
would be 2x) / the launch duration measured with HIP events on the launch stream inside the timed
region.  `legs` holds the other BASELINE configurations measured the same way on this GPU (N = 1 only):
V = 100 000 d = 100 (the north star's target size), V = 50 000 d = 50 with the exact fp32-MFMA prefilter
(config 2), the literal sign mode (the classes' default), and config 5 (enhanced tokenizer: frequency-aware
scoring + adaptive curvature).  `roofline_bw` lists the bandwidth-bound kernels of the path (one-row-vs-all,
merge, gathered distances, coherence, table re-projection): algorithmic bytes per launch / event time / 8 TB/s.
`cpu_baseline` times the oracle's OpenMP restatement of the same search on a bounded row sample on this host's
cores (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

os.environ.setdefault("TQDM_DISABLE", "1")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

V, D, SCALE, SEED, THR, CURV = 50000, 100, 0.05, 42, 0.5, 1.0
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: bf16 MFMA, dense (not the 2:1-sparsity figure)
PEAK_HBM_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E


def cpu_baseline(X: np.ndarray, device, budget_s: float = 12.0) -> dict:
    """Time the oracle (CPU restatement, OpenMP) on rows [0, R) of the same search (any table: the headline's and the
    V = 100 000 leg's) and extrapolate by pair count; check that the GPU returns the identical pair for the same row range."""
    dim = X.shape[1] - 1
    from oracle import hm_oracle as O
    from hyptokenizer_amd.engine import MergeEngine
    O.build()
    n = X.shape[0]
    cores = O.num_threads()
    gpu_engine = MergeEngine(n + 8, X.shape[1], "lorentz", device)
    table = torch.zeros((n + 8, X.shape[1]), dtype=torch.float32, device=device)
    table[:n] = torch.from_numpy(X).to(device)
    gpu_engine.set_table(table, n)

    def pairs(r):
        return r * (n - 1) - (r - 1) * r // 2

    O.pairwise_topk(X, n, CURV, THR, 1, 1, 0, 64, fast=True)              # warm (page in, thread pool)
    r = 1024
    t0 = time.perf_counter()
    O.pairwise_topk(X, n, CURV, THR, 1, 1, 0, r, fast=True)
    t_cal = time.perf_counter() - t0
    rows = int(min(n - 1, max(r, r * budget_s / max(t_cal, 1e-3))))
    rows = max(256, rows // 256 * 256)
    t0 = time.perf_counter()
    od, oi, oj, oc = O.pairwise_topk(X, n, CURV, THR, 1, 1, 0, rows, fast=True)
    t = time.perf_counter() - t0
    frac = pairs(rows) / pairs(n - 1)
    est_scan_s = t / frac
    g = gpu_engine.argmin(CURV, THR, 0, rows)
    same = (g is None and oc == 0) or (g is not None and oc > 0 and (g[1], g[2]) == (int(oi[0]), int(oj[0]))
                                       and np.float32(g[0]).view(np.uint32) == od[:1].view(np.uint32)[0])
    del gpu_engine, table
    return {
        "value": 1.0 / est_scan_s, "unit": "merges/s", "cores": cores, "kind": "port",
        "sample": f"rows [0,{rows}) of the V={n} d={dim} search = {100 * frac:.1f}% of all pairs in {t:.2f} s, "
                  f"extrapolated by pair count (midpoint cost negligible)",
        "gflops": 2.0 * pairs(rows) * (dim + 1) / t / 1e9,
        "same_pair_as_gpu_on_sample": bool(same),
        "note": "the oracle's own OpenMP restatement (the reference cannot run at this size, SURVEY F9): context, not a target",
    }


def traffic_from_profiles(form: str):
    """Fabric-side bytes per scan launch from the committed PMC pass (profiles/*_pmc_scan_kernel_<form>.json)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_scan_kernel_{form}.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            dd = json.load(f)["derived"]
        return dd["fetch_bytes_per_launch_corrected"] + dd["write_bytes_per_launch"], os.path.basename(files[-1])
    except Exception:
        return None, None


def warm_clocks(eng, ms: float = 60.0, thr: float = THR) -> None:
    """>= `ms` of back-to-back scans before a timed region: a 20-step run must measure the steady state"""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        eng.argmin(CURV, thr)
    torch.cuda.synchronize()


def std_loop_leg(vocab_size, dim, prefilter, sign, steps, device, thr=THR, label="", with_cpu=False):
    """merges/s of the standard loop + the scan's roofline for another configuration (N = 1); `with_cpu`: the CPU
    baseline of the same search timed beside it, as for the headline"""
    from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table
    from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
    X = lorentz_table(vocab_size, dim, seed=SEED, scale=SCALE)
    tok = HyperbolicTokenizer(cjk_vocab(vocab_size), torch.nn.Parameter(X), curvature=CURV, merge_threshold=thr, device=device,
                              max_vocab_size=vocab_size + steps + 80, sign_convention=sign, prefilter=prefilter)
    eng = tok._get_engine()
    tok.optimize_merges(steps=8, log_every=10 ** 9)
    gc.collect()                                 # (before the warm-up, not between it and the timed loop)
    warm_clocks(eng, 40.0, thr)
    eng.debug_time_loops(True)                   # every scan of the timed loop carries its event pair (in the dispatch)
    eng.scan_totals(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tok.optimize_merges(steps=steps, log_every=10 ** 9)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    done = len(tok.merge_history) - 8
    tot = eng.scan_totals()
    eng.debug_time_loops(False)
    out = {"workload": label, "merges_per_s": done / el if el > 0 else None, "ms_per_step": 1e3 * el / max(done, 1), "steps": done}
    if tot["launches"] > 0 and tot["scan_ms"] > 0:
        ms = tot["scan_ms"] / tot["launches"]
        fl = 2.0 * (dim + 1) * tot["pairs"] / tot["launches"]
        bf = prefilter != "f32" and dim >= 24
        peak = PEAK_BF16_MFMA_TFLOPS if bf else PEAK_FP32_MFMA_TFLOPS
        out["roofline"] = {"bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                           "frac": fl / (ms * 1e-3) / 1e12 / peak, "avg_launch_ms": ms, "timed_launches": tot["launches"],
                           "form": "bf16" if bf else "f32"}
    del tok, eng
    if with_cpu and out["merges_per_s"]:
        out["cpu_baseline"] = cpu_baseline(X.numpy(), device, budget_s=10.0)
        out["speedup_vs_cpu_baseline"] = out["merges_per_s"] / out["cpu_baseline"]["value"]
    return out


def bandwidth_kernels(device) -> list:
    """Event-timed bandwidth-bound kernels of the path at V = 50 000, d = 100: algorithmic bytes
    per launch / average duration of back-to-back launches on the launch stream / 8 TB/s."""
    from hyptokenizer_amd import _lib
    from hyptokenizer_amd.engine import MergeEngine
    from hyptokenizer_amd.synthetic import lorentz_table
    L = _lib.load()
    n, d1 = V, D + 1
    X = lorentz_table(n, D, seed=SEED, scale=SCALE)
    table = torch.zeros((n + 2048, d1), device=device)
    table[:n] = X.to(device)
    eng = MergeEngine(n + 2048, d1, "lorentz", device)
    eng.set_table(table, n)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    rs_bytes = 4 * (4 * 25 + 8)                      # fp32 image row (NG = 25, odd chunk count): 432 B
    out = []

    def timed(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3     # s per launch

    def rec(name, nbytes, secs, note):
        gbps = nbytes / secs / 1e9
        out.append({"kernel": name, "bytes_per_launch": nbytes, "avg_launch_us": secs * 1e6, "achieved": gbps,
                    "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS, "bound": "hbm", "note": note})

    # gathered pair distances (K5): two image rows read per output
    b = 200000
    I = torch.randint(0, n, (b,), dtype=torch.int32, device=device)
    J = torch.randint(0, n, (b,), dtype=torch.int32, device=device)
    o = torch.empty(b, device=device)
    s = timed(lambda: L.hm_pair_distance(eng._h, C.c_void_p(I.data_ptr()), C.c_void_p(J.data_ptr()), b, C.c_float(CURV),
                                         C.c_void_p(o.data_ptr()), stream), 20)
    rec("hm_pairdist_kernel", b * (2 * rs_bytes + 12), s, f"{b} gathered pairs, 2 x 432 B rows each (random rows: cache-resident image)")
    # one row vs all (K3): the whole image once
    o2 = torch.empty(n, device=device)
    s = timed(lambda: L.hm_row_vs_all(eng._h, n - 1, n, C.c_float(CURV), C.c_void_p(o2.data_ptr()), stream), 50)
    rec("hm_rowvsall_kernel", n * (rs_bytes + 4), s, "row n-1 against all rows: the fp32 image once + n distances")
    # coherence (config 5): per candidate 2 rows + 50 sampled rows read, 50 distances written
    bc, ns = 10000, 50
    Ic, Jc = I[:bc].contiguous(), J[:bc].contiguous()
    W = torch.full((bc,), 0.5, device=device)
    S = torch.randint(0, n, (bc, ns), dtype=torch.int32, device=device)
    oc = torch.empty((bc, ns), device=device)
    s = timed(lambda: L.hm_coherence_batch(eng._h, C.c_void_p(Ic.data_ptr()), C.c_void_p(Jc.data_ptr()), C.c_void_p(W.data_ptr()),
                                           C.c_void_p(S.data_ptr()), bc, ns, C.c_float(CURV), C.c_void_p(oc.data_ptr()), stream), 20)
    rec("hm_coherence_kernel", bc * ((ns + 2) * rs_bytes + ns * 8), s, f"{bc} candidates x {ns} sampled rows (config 5 refresh)")
    # table re-projection (K6): spatial part of every table row read, x0 written (+ image time slots)
    s = timed(lambda: L.hm_project_table(eng._h, C.c_void_p(table.data_ptr()), table.stride(0), n, C.c_float(CURV), stream), 50)
    rec("hm_project_table_kernel", n * (4 * D + 4 + 4 + 12), s, "rows [0, n): d spatial floats read, x0 written to the table and both images")
    # incremental step (merge + new row vs all + fold), from the device-resident loop
    eng.set_table(table, n)
    eng.set_token_lengths(np.ones(n, np.int32))
    best = eng.argmin(CURV, THR)
    if best is not None:
        eng.incr_merge_steps(CURV, THR, table, 8, best)
        best = eng.argmin(CURV, THR)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done_total = 0
        for _ in range(8):
            _r, done, best = eng.incr_merge_steps(CURV, THR, table, 64, best)
            done_total += done
        torch.cuda.synchronize()
        s = (time.perf_counter() - t0) / max(done_total, 1)
        rec("hm_incr_step_kernel", (n + done_total / 2) * rs_bytes, s,
            "one launch per merge: midpoint + new row vs all rows + fold; wall time of 64-step batches / steps (host sync per batch included)")
    del eng
    # the same row-vs-all pass on the largest table an engine takes (131 072 rows): at 50 000 rows a launch moves
    # 21.8 MB, which the whole chip streams in under 3 us -- launch latency and one memory round trip are most of the
    # 7-8 us measured above; the larger table shows the kernel's streaming rate
    nb = 131072
    tb = torch.zeros((nb, d1), device=device)
    tb[:] = lorentz_table(nb, D, seed=SEED + 1, scale=SCALE).to(device)
    engb = MergeEngine(nb, d1, "lorentz", device)
    engb.set_table(tb, nb)
    ob = torch.empty(nb, device=device)
    s = timed(lambda: L.hm_row_vs_all(engb._h, nb - 1, nb, C.c_float(CURV), C.c_void_p(ob.data_ptr()), stream), 50)
    rec("hm_rowvsall_kernel (n = 131072)", nb * (rs_bytes + 4), s, "row n-1 against all rows of the largest table (56.7 MB image)")
    del engb
    return out


def config5_leg(device, steps: int = 24) -> dict:
    """BASELINE config 5 on one GPU: EnhancedFastHyperbolicTokenizer, frequency-aware scoring + adaptive curvature
    (V = 100 000, d = 100).  Per step: 100 cached candidates scored (torch.randperm(n) per candidate on the host, as the
    reference draws them; midpoint + 50 gathered distances per candidate in one fused kernel); one refresh scores every
    candidate; the curvature step fires once (analytic gradient; the reference's raises).  A SHORT run on purpose:
    the reference's loop keeps merged tokens in the table, so on this synthetic table it merges the same nearest pairs again
    and again, the duplicates multiply the candidate list and every refresh scores every candidate (200 steps: 60 ms per step,
    a refresh of 19 s -- the algorithm's own cost, measured in round 3)."""
    import random
    from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table
    from hyptokenizer_amd.tokenizer.enhanced_fast_hyperbolic_merge import EnhancedFastHyperbolicTokenizer
    n5, d5 = 100000, 100
    X = lorentz_table(n5, d5, seed=SEED, scale=SCALE)
    vocab = cjk_vocab(n5)
    random.seed(SEED)
    torch.manual_seed(SEED)
    tok = EnhancedFastHyperbolicTokenizer(vocab, torch.nn.Parameter(X), curvature=CURV, merge_threshold=0.45, device=device,
                                          max_vocab_size=n5 + steps + 64, sign_convention="lorentz", use_frequency_aware=True,
                                          use_hierarchical=False, use_adaptive_curvature=True, use_compression_aware=False,
                                          optimize_curvature_freq=steps // 2)
    rs = np.random.RandomState(SEED)                  # synthetic pair-frequency table: Zipf(1.2) counts over random pairs
    a, b = rs.randint(0, n5, 200000), rs.randint(0, n5, 200000)
    cnt = rs.zipf(1.2, 200000).clip(max=10 ** 6)
    tok.pair_frequencies = {(vocab[i], vocab[j]): int(c) for i, j, c in zip(a.tolist(), b.tolist(), cnt.tolist())}
    # a threshold with a few thousand candidates at the first refresh (every one of them is scored, as in the reference)
    eng = tok._get_engine()
    thr = 0.45
    for _ in range(12):
        cnt0 = eng.count_candidates(tok._c(), thr)
        if cnt0 > 6000:
            thr *= 0.97
        elif cnt0 < 600:
            thr *= 1.015
        else:
            break
    tok.merge_threshold = thr
    # first-use costs out of the timed region (code objects of the coherence / projection kernels, their launch attributes,
    # the helper's one-time self-check): a throw-away tokenizer of the same width runs a few steps and one curvature step
    prime = EnhancedFastHyperbolicTokenizer(cjk_vocab(4000), torch.nn.Parameter(lorentz_table(4000, d5, seed=SEED + 3, scale=SCALE)),
                                            curvature=CURV, merge_threshold=thr, device=device, max_vocab_size=4064,
                                            sign_convention="lorentz", use_frequency_aware=True, use_hierarchical=False,
                                            use_adaptive_curvature=True, use_compression_aware=False, optimize_curvature_freq=2)
    prime.pair_frequencies = {}
    prime.optimize_merges(steps=4, log_every=10 ** 9, adaptive_threshold=False)
    del prime
    gc.collect()
    warm_clocks(eng, 40.0, thr)
    state = (random.getstate(), torch.get_rng_state())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tok.optimize_merges(steps=steps, log_every=10 ** 9, adaptive_threshold=False)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    done = len(tok.merge_history)
    # where the time goes, measured on the same tokenizer after the run (generator states restored afterwards):
    # the host RNG the reference semantics prescribe, the fused kernel, one refresh, one curvature step
    t1 = time.perf_counter()
    for _ in range(20):
        torch.randperm(n5)
    rp = (time.perf_counter() - t1) / 20
    t1 = time.perf_counter()
    smp = tok._coherence_samples(100)
    rng100 = time.perf_counter() - t1
    ii = np.arange(100, dtype=np.int32)
    jj = ii + 1000
    t1 = time.perf_counter()
    for _ in range(5):
        eng.coherence_distances(ii, jj, np.full(100, 0.5, np.float32), smp, tok._c())
    k100 = (time.perf_counter() - t1) / 5
    t1 = time.perf_counter()
    scored = tok._score_candidates([0.4] * 100, ii, jj)
    score100 = time.perf_counter() - t1
    del scored
    tok.cache.candidates = []
    t1 = time.perf_counter()
    tok._find_merge_candidates_fast()
    refresh_s = time.perf_counter() - t1
    t1 = time.perf_counter()
    tok._optimize_curvature(tok.embeddings)
    tok._project_embeddings()
    torch.cuda.synchronize()
    curv_s = time.perf_counter() - t1
    random.setstate(state[0])
    torch.set_rng_state(state[1])
    return {"workload": f"EnhancedFastHyperbolicTokenizer.optimize_merges V={n5} d={d5} lorentz thr={thr:.4f} ({cnt0} candidates at the "
                        f"first refresh) freq-aware + adaptive curvature (one curvature step + whole-table re-projection inside the run)",
            "merges_per_s": done / el, "ms_per_step": 1e3 * el / max(done, 1), "steps": done,
            "curvature_after": float(torch.as_tensor(tok.get_curvature()).detach()),
            "host_randperm_ms": rp * 1e3,
            "breakdown_ms": {"host_rng_per_step_100_candidates": rng100 * 1e3, "coherence_kernel_and_copies_100_candidates": k100 * 1e3,
                             "scoring_100_candidates_total": score100 * 1e3, "one_refresh_all_candidates_scored": refresh_s * 1e3,
                             "one_curvature_step_with_table_reprojection": curv_s * 1e3,
                             "note": "a step pops and scores 100 cached candidates (host RNG + one fused kernel, the second half's "
                                     "permutations drawn while the first half's kernel runs); every ~101st step is a refresh; the "
                                     "curvature step fires once in the timed run"},
            "note": "per scored candidate the host draws torch.randperm(n)[:50] (reference semantics, enhanced_fast_hyperbolic_merge.py:"
                    "324-325) -- through the library's MT19937 helper (~0.1 ms at n = 100 000; torch.randperm itself: host_randperm_ms); "
                    "the reference needs ~2.5 ms of distance() calls per candidate on top and cannot run its all-pairs search at this "
                    "size (SURVEY F9)"}


def tokenize_leg(device, n_lines: int = 400_000) -> dict:
    """HyperbolicTokenizer.tokenize over a batch of lines (hm_tokenize_batch, one lane per line).
    Synthetic English-like text (a 12 000-word lexicon used with Zipf frequencies, ~200 characters per line) and the
    prefix-chain rules that build its 8 000 most frequent words.
    Timed: the kernel with symbols resident in HBM; beside it the end-to-end encode_batch (string -> symbols on the
    host, PCIe both ways, Python lists out) and the oracle's pure-Python loop (what the reference runs) on a sample."""
    import random
    from hyptokenizer_amd.tokenizer.batch_encoder import BatchEncoder
    rng = np.random.default_rng(SEED)
    letters = np.array(list("etaoinshrdlucmfwypvbgkjqxz"))
    p = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.2, 2.4, 2.0, 1.9, 1.0, 1.5, 2.0, 0.8, 0.15, 0.1, 0.15, 0.07])
    p = p / p.sum()
    # a lexicon of 12 000 words (lengths 1..12, English letter frequencies) used with Zipf(1.1) frequencies; the rules
    # build the 8 000 most frequent words left to right (prefix + next letter), as merges learned from such text would
    n_words = 12000
    wlen = np.clip(rng.poisson(4.2, n_words), 1, 12)
    wl = rng.choice(len(letters), size=int(wlen.sum()), p=p)
    wends = np.cumsum(wlen)
    words = ["".join(letters[wl[int(e - k):int(e)]].tolist()) for e, k in zip(wends, wlen)]
    merges, seen = [], set()
    for wd in words[:8000]:
        for k in range(2, len(wd) + 1):
            r = (wd[:k - 1], wd[k - 1])
            if r not in seen:
                seen.add(r)
                merges.append((r[0], r[1], wd[:k]))
    vocab = ["<pad>", "<bos>", "<eos>", "<unk>"] + sorted({x for m in merges for x in m} | set(letters.tolist()) | {" ", ",", "."})
    token2idx = {t: k for k, t in enumerate(vocab)}
    rules = {(a, b): ab for a, b, ab in merges}
    enc = BatchEncoder(rules, token2idx, device)
    zipf = 1.0 / np.arange(1, n_words + 1) ** 1.1
    zipf /= zipf.sum()
    words_per_line = rng.integers(8, 70, size=n_lines)
    pick = rng.choice(n_words, size=int(words_per_line.sum()), p=zipf)
    warr = np.array(words, dtype=object)[pick]
    wend = np.cumsum(words_per_line)
    lines = [" ".join(warr[int(e - k):int(e)].tolist()) for e, k in zip(wend, words_per_line)]
    lens = np.fromiter((len(t) for t in lines), dtype=np.int64, count=n_lines)
    sym_h, off_h = enc.symbols(lines)
    sym, off = torch.from_numpy(sym_h).to(device), torch.from_numpy(off_h).to(device)
    order = torch.argsort(off[1:] - off[:-1], descending=True, stable=True)
    out, out_len, _ = enc.run(sym, off, order)
    torch.cuda.synchronize()
    n_out = int(out_len.sum().item())
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    ev0.record()
    for _ in range(reps):
        enc.run(sym, off, order)
    ev1.record()
    torch.cuda.synchronize()
    k_ms = ev0.elapsed_time(ev1) / reps
    t0 = time.perf_counter()
    ids = enc.encode_batch(lines)
    e2e = time.perf_counter() - t0
    t0 = time.perf_counter()
    enc.encode_arrays(lines)
    e2e_arr = time.perf_counter() - t0
    # the reference's loop (oracle, pure Python) on a bounded sample; the results must agree
    from oracle import hm_oracle as O
    sample = list(range(0, n_lines, max(1, n_lines // 3000)))
    t0 = time.perf_counter()
    want = [O.encode(rules, token2idx, lines[k]) for k in sample]
    cpu_s = time.perf_counter() - t0
    n_chars = int(lens.sum())
    sample_chars = int(sum(lens[k] for k in sample))
    alg_bytes = 4.0 * (n_chars + n_out) + 12.0 * n_lines           # symbols in, tokens out, offsets + lengths
    return {"workload": f"tokenize_batch: {n_lines} lines, {n_chars} characters, {len(rules)} rules (prefix chains of the 8000 most frequent words)",
            "kernel": "hm_tokenize_kernel", "ms": k_ms, "chars_per_s": n_chars / (k_ms * 1e-3), "lines_per_s": n_lines / (k_ms * 1e-3),
            "tokens_out": n_out, "compression_chars_per_token": n_chars / max(n_out, 1),
            "roofline": {"bound": "hbm", "achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, "traffic": None,
                         "algorithmic_bytes": alg_bytes,
                         "note": "one lane per line, a chain of dependent rule-table probes per line: bound by probe latency x the slowest of a wave's 64 lanes, not by HBM bandwidth"},
            "end_to_end_encode_batch_chars_per_s": n_chars / e2e, "end_to_end_encode_arrays_chars_per_s": n_chars / e2e_arr,
            "cpu_baseline": {"value": sample_chars / cpu_s, "unit": "chars/s", "cores": 1, "kind": "port",
                             "sample": f"{len(sample)} of the lines through the oracle's pure-Python tokenize + encode"},
            "sample_matches_cpu": all(ids[k] == w for k, w in zip(sample, want))}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="headline line only")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the merge engine has no CPU fallback")
    # HM_BENCH_REHEARSE=1: the N > 1 code path on a box with ONE GPU -- every rank uses cuda:0 and the exchange goes over
    # gloo (CPU tensors).  Only for checking that the path runs; its numbers mean nothing.
    rehearse = os.environ.get("HM_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    xdev = torch.device("cpu") if rehearse else device          # where the bench's own small collectives live
    shard = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        from hyptokenizer_amd.sharding import ShardContext
        shard = ShardContext(device=xdev)

    from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table
    from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
    from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer

    X = lorentz_table(V, D, seed=SEED, scale=SCALE)
    vocab = cjk_vocab(V)
    steps_total = args.steps + args.warmup
    tok = HyperbolicTokenizer(vocab, torch.nn.Parameter(X), curvature=CURV, merge_threshold=THR, device=device,
                              max_vocab_size=V + steps_total + 192, sign_convention="lorentz", shard=shard)
    # code objects of the loop's kernels loaded and their launch attributes set before anything is timed, whatever
    # --warmup says: three steps of a throw-away table of the same width
    prime = HyperbolicTokenizer(cjk_vocab(3000), torch.nn.Parameter(lorentz_table(3000, D, seed=SEED + 7, scale=SCALE)), curvature=CURV,
                                merge_threshold=THR, device=device, max_vocab_size=3100, sign_convention="lorentz")
    prime.optimize_merges(steps=3, log_every=10 ** 9)
    del prime
    eng = tok._get_engine()                      # builds the scan image: inputs resident before timing
    tok._sync_token_lengths(eng)                 # (the token lengths the device-resident loop reads are inputs too)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def settle():
        # a generation-2 collection over three 50k-token vocabularies is a 50 ms host stall: collect between the legs and
        # park the survivors -- BEFORE the clocks are warmed, never between the warm-up and a timed region (the GPU idles
        # for the duration and the timed region then starts on cold clocks: its first hundred scans ran 5-15 % slower)
        gc.collect()
        gc.freeze()

    tok.optimize_merges(steps=args.warmup, log_every=10 ** 9)
    settle()
    if world == 1:
        warm_clocks(eng)                         # >= 60 ms of scans whatever --warmup says
    else:
        # the same for the sharded run: a FIXED number of global searches (a time-based loop would let the ranks disagree
        # on the number of collectives)
        from hyptokenizer_amd.sharding import sharded_argmin
        for _ in range(200):
            sharded_argmin(eng, shard, CURV, tok._search_threshold())
        torch.cuda.synchronize()
    # Every scan launch of the timed region carries its own HIP event pair IN the dispatch (start / stop timestamps of that
    # kernel: no extra packets on the stream), and each device batch one pair around it: the roofline's mean launch duration
    # and the time a step spends outside the scan both come from the timed region itself.
    timing = world == 1 or not rehearse          # (the in-library sharded loop times every scan too; the gloo rehearsal does not)
    if timing:
        eng.debug_time_loops(True)
    eng.scan_totals(reset=True)
    barrier()
    t0 = time.perf_counter()
    tok.optimize_merges(steps=args.steps, log_every=10 ** 9)
    barrier()
    elapsed = time.perf_counter() - t0
    merges_done = len(tok.merge_history) - args.warmup
    tot = eng.scan_totals()
    loop_t = None
    if timing:
        lt = eng.last_loop_timing()
        eng.debug_time_loops(False)
        if lt["steps"] > 0:
            loop_t = lt
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        agg = torch.tensor([tot["scan_ms"], float(tot["pairs"]), float(tot["launches"])], dtype=torch.float64, device=xdev)
        lst = [torch.zeros_like(agg) for _ in range(world)]
        dist.all_gather(lst, agg)
        per_rank = [x.cpu().tolist() for x in lst]
    else:
        per_rank = [[tot["scan_ms"], float(tot["pairs"]), float(tot["launches"])]]

    # fast-path figure (FastHyperbolicTokenizer semantics: one exact top-10000 search per ~101 steps)
    fast = None
    if rank == 0 or world > 1:
        fsteps = 2020
        ftok = FastHyperbolicTokenizer(vocab, torch.nn.Parameter(X), curvature=CURV, merge_threshold=THR, device=device,
                                       max_vocab_size=V + fsteps + 64, sign_convention="lorentz", shard=shard)
        ftok._get_engine()
        ftok.optimize_merges(steps=202, log_every=10 ** 9, adaptive_threshold=False)
        settle()
        barrier()
        tf0 = time.perf_counter()
        ftok.optimize_merges(steps=fsteps - 202, log_every=10 ** 9, adaptive_threshold=False)
        barrier()
        tf = time.perf_counter() - tf0
        fast = {"merges_per_s": (len(ftok.merge_history) - 202) / tf, "steps": fsteps - 202,
                "note": "FastHyperbolicTokenizer.optimize_merges: cache of 10000, one exact top-k search per ~101 steps; "
                        "the merges between two refreshes are issued as one launch, the next refresh is enqueued behind them"}
        del ftok

    # incremental figure (SURVEY 8(d) variant (ii)): same merges, nearest pair maintained with one
    # row-vs-all pass per step instead of a full search; the merge sequence is checked against the timed run's
    incr = None
    if rank == 0 or world > 1:
        isteps = max(args.steps + args.warmup, 2)
        itok = HyperbolicTokenizer(vocab, torch.nn.Parameter(X), curvature=CURV, merge_threshold=THR, device=device,
                                   max_vocab_size=V + isteps + 64, sign_convention="lorentz", shard=shard, incremental=True)
        itok._get_engine()
        itok.optimize_merges(steps=1, log_every=10 ** 9)           # the one full search
        settle()
        barrier()
        ti0 = time.perf_counter()
        itok.optimize_merges(steps=isteps - 1, log_every=10 ** 9)
        barrier()
        ti = time.perf_counter() - ti0
        incr = {"merges_per_s": (len(itok.merge_history) - 1) / ti, "steps": isteps - 1,
                "same_merges_as_full_search": itok.merge_history == tok.merge_history[:len(itok.merge_history)],
                "note": "HyperbolicTokenizer(incremental=True): one full search, then one launch per merge "
                        "(midpoint + new row vs all + fold), up to 256 steps per host call"}
        del itok

    legs, bw = None, None
    if rank == 0 and world == 1 and not args.no_legs:
        legs = {}
        lsteps = max(20, min(args.steps, 100))
        for key, fn in (
            ("v100k_d100_bf16", lambda: std_loop_leg(100000, 100, "bf16", "lorentz", lsteps, device, with_cpu=not args.no_cpu_baseline,
                                                     label="V=100000 d=100 lorentz thr=0.5, bf16 prefilter (north star target size)")),
            ("v50k_d50_f32", lambda: std_loop_leg(50000, 50, "f32", "lorentz", lsteps, device,
                                                  label="V=50000 d=50 lorentz thr=0.5, fp32-MFMA prefilter (BASELINE config 2)")),
            ("v50k_d100_f32", lambda: std_loop_leg(50000, 100, "f32", "lorentz", 20, device,
                                                   label="V=50000 d=100 lorentz thr=0.5, fp32-MFMA prefilter")),
            ("v50k_d100_literal", lambda: std_loop_leg(50000, 100, "auto", "reference", 20, device, thr=0.1,
                                                       label="V=50000 d=100 literal sign (the classes' default: every pair at distance 0, tie flood) thr=0.1")),
            ("config5_enhanced", lambda: config5_leg(device)),
            ("tokenize_batch", lambda: tokenize_leg(device)),
        ):
            try:
                legs[key] = fn()
            except Exception as exc:             # a leg must not take the headline line down with it
                legs[key] = {"error": f"{type(exc).__name__}: {exc}"}
            settle()
            barrier()
        try:
            bw = bandwidth_kernels(device)
        except Exception as exc:
            bw = [{"error": f"{type(exc).__name__}: {exc}"}]

    if rank == 0:
        launches = sum(p[2] for p in per_rank)
        avg_ms = sum(p[0] for p in per_rank) / max(launches, 1.0)           # mean launch duration
        flops_per_launch = 2.0 * (D + 1) * sum(p[1] for p in per_rank) / max(launches, 1.0)
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        form = os.environ.get("HM_SCAN_PRECISION", "auto")
        bf16 = form != "f32"                     # auto picks the bf16 prefilter at d = 100
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
        kernel_name = "hm_scan_kernel<13 chunks,lorentz,ARGMIN,bf16>" if bf16 else "hm_scan_kernel<NG=25,lorentz,ARGMIN,fp32>"
        traffic, traffic_src = traffic_from_profiles("bf16" if bf16 else "f32") if world == 1 else (None, None)
        n_mid = V + args.warmup + args.steps / 2.0
        scan_ms_per_step = max(p[0] / max(p[2], 1.0) for p in per_rank)     # slowest rank's scan per step
        out = {
            "metric": "merge_steps_per_sec",
            "value": merges_done / elapsed,
            "unit": "merges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "bf16 prefilter + f32 canonical" if bf16 else "f32",
            "data": "synthetic",
            "config": {"workload": f"HyperbolicTokenizer.optimize_merges, full all-pairs search every step, V={V} d={D} "
                                   f"fp32, lorentz sign, thr={THR}, c={CURV}, scale={SCALE}, seed={SEED}",
                       "vocab": V, "dim": D, "merge_threshold": THR, "parallelism": f"rows sharded over {world} rank(s)"},
            "pairwise_dist_GBps_effective": (n_mid * n_mid * 4.0) / (scan_ms_per_step * 1e-3) / 1e9,
            "step_overhead_ms": 1e3 * elapsed / max(args.steps, 1) - avg_ms,
            "step_overhead_note": ("ms_per_step - mean scan launch of the timed region; on the device alone (last batch: wall time "
                                   f"between its first scan and its last tail kernel - its scans) / steps = "
                                   f"{(loop_t['batch_ms'] - loop_t['scan_ms']) / loop_t['steps'] * 1e3:.1f} us over {loop_t['steps']} steps"
                                   if loop_t else "ms_per_step - mean scan launch"),
            "step_overhead_device_ms": ((loop_t["batch_ms"] - loop_t["scan_ms"]) / loop_t["steps"]) if loop_t else None,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "traffic_note": ("bytes per launch past the XCD L2s (FETCH_SIZE corrected x2 + WRITE_SIZE) from "
                                          f"profiles/{traffic_src}; served by the Infinity Cache (the scan image is "
                                          "resident), compulsory bytes = the image once") if traffic else None,
                         "kernel": kernel_name, "avg_launch_ms": avg_ms,
                         "flops_per_launch": flops_per_launch, "timed_launches": launches,
                         "note": "flops = N(N-1)(d+1) algorithmic (triangle); peak = dense MFMA peak of the prefilter's dtype; "
                                 "avg_launch_ms = mean over ALL scan launches of the timed region (HIP events carried in each dispatch, "
                                 "on the launch stream)"},
            "fast_path": fast,
            "incremental": incr,
            "legs": legs,
            "roofline_bw": bw,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(X.numpy(), device)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
