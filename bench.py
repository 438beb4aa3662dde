#!/usr/bin/env python
"""Headline benchmark of the MI355X merge engine (BASELINE.json metric:
"merge-steps/sec + pairwise Lorentz-dist GB/s, V=50k d=100, 1/2/4/8 GPU").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE merge iteration of ``HyperbolicTokenizer.optimize_merges`` (reference
``tokenizer/hyperbolic_merge.py:357-412``): a full all-pairs Lorentz-distance search for the
nearest pair below the threshold, then the log-map/exp-map midpoint appended as a new row.
Workload: V = 50 000 synthetic Lorentz rows, d = 100, fp32, curvature 1, sign-corrected Minkowski
form ("lorentz": the mode in which the search is non-degenerate), threshold 0.5.
Inputs are resident in HBM before the timed region.  With N > 1 ranks the rows of the pair
triangle are sharded over the ranks (equal pair counts), each rank scans its share of the SAME
problem (strong scaling), the best records are all-gathered over RCCL and every rank applies the
merge to its replica.

One JSON line on rank 0; see README/DESIGN.md for the fields.  `roofline` is for the dominant
kernel of the timed run (hm_scan_kernel; by default its bf16-MFMA prefilter form, survivors are
re-evaluated in the canonical fp32 arithmetic): achieved = N(N-1)(d+1) flops per launch (triangle
only; the reference's dense convention would be 2x) / the launch duration measured with HIP events
on the launch stream inside the timed region.  `roofline_fp32_form` is the same search with the
exact fp32-MFMA prefilter (HM_SCAN_PRECISION=f32), timed right after on the same table.  `cpu_baseline` times the oracle's OpenMP restatement of
the same search on a bounded row sample on this host's cores (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
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


def cpu_baseline(X: np.ndarray, device, budget_s: float = 12.0) -> dict:
    """Time the oracle (CPU restatement, OpenMP) on rows [0, R) of the same search and extrapolate
    by pair count; check that the GPU returns the identical pair for the same row range."""
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
    return {
        "value": 1.0 / est_scan_s, "unit": "merges/s", "cores": cores, "kind": "port",
        "sample": f"rows [0,{rows}) of the V={n} d={D} search = {100 * frac:.1f}% of all pairs in {t:.2f} s, "
                  f"extrapolated by pair count (midpoint cost negligible)",
        "gflops": 2.0 * pairs(rows) * (D + 1) / t / 1e9,
        "same_pair_as_gpu_on_sample": bool(same),
    }


def traffic_from_profiles(form: str):
    """Fabric-side bytes per scan launch from the committed PMC pass (profiles/*_pmc_scan_kernel.json:
    FETCH_SIZE x 1024 x 2 on gfx950 + WRITE_SIZE x 1024, MI355X_MICROARCH.md HBM section).  PMC
    counters cannot be collected from inside the timed run; same kernel, same size."""
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


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the merge engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    shard = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
        from hyptokenizer_amd.sharding import ShardContext
        shard = ShardContext(device=device)

    from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table
    from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
    from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer

    X = lorentz_table(V, D, seed=SEED, scale=SCALE)
    vocab = cjk_vocab(V)
    steps_total = args.steps + args.warmup
    tok = HyperbolicTokenizer(vocab, torch.nn.Parameter(X), curvature=CURV, merge_threshold=THR, device=device,
                              max_vocab_size=V + steps_total + 64, sign_convention="lorentz", shard=shard)
    eng = tok._get_engine()                      # builds the scan image: inputs resident before timing
    torch.cuda.synchronize()

    def barrier():
        # a generation-2 collection over three 50k-token vocabularies is a 50 ms host stall: collect between the
        # legs and park the survivors, so that no leg's clock is charged for it
        gc.collect()
        gc.freeze()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    tok.optimize_merges(steps=args.warmup, log_every=10 ** 9)
    eng.scan_totals(reset=True)
    barrier()
    t0 = time.perf_counter()
    tok.optimize_merges(steps=args.steps, log_every=10 ** 9)
    barrier()
    elapsed = time.perf_counter() - t0
    merges_done = len(tok.merge_history) - args.warmup
    tot = eng.scan_totals()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        agg = torch.tensor([tot["scan_ms"], float(tot["pairs"]), float(tot["launches"])], dtype=torch.float64, device=device)
        lst = [torch.zeros_like(agg) for _ in range(world)]
        dist.all_gather(lst, agg)
        per_rank = [x.cpu().tolist() for x in lst]
    else:
        per_rank = [[tot["scan_ms"], float(tot["pairs"]), float(tot["launches"])]]

    # fast-path figure (FastHyperbolicTokenizer semantics: one exact top-10000 search per ~101 steps)
    fast = None
    if rank == 0 or world > 1:
        fsteps = 1010
        ftok = FastHyperbolicTokenizer(vocab, torch.nn.Parameter(X), curvature=CURV, merge_threshold=THR, device=device,
                                       max_vocab_size=V + fsteps + 64, sign_convention="lorentz", shard=shard)
        ftok._get_engine()
        ftok.optimize_merges(steps=101, log_every=10 ** 9, adaptive_threshold=False)
        barrier()
        tf0 = time.perf_counter()
        ftok.optimize_merges(steps=fsteps - 101, log_every=10 ** 9, adaptive_threshold=False)
        barrier()
        tf = time.perf_counter() - tf0
        fast = {"merges_per_s": (len(ftok.merge_history) - 101) / tf, "steps": fsteps - 101,
                "note": "FastHyperbolicTokenizer.optimize_merges: cache of 10000, one exact top-k search per ~101 steps"}

    # incremental figure (SURVEY 8(d) variant (ii)): same merges, nearest pair maintained with one
    # row-vs-all pass per step instead of a full search; the merge sequence is checked against the
    # timed run's
    incr = None
    if rank == 0 or world > 1:
        isteps = max(args.steps + args.warmup, 2)
        itok = HyperbolicTokenizer(vocab, torch.nn.Parameter(X), curvature=CURV, merge_threshold=THR, device=device,
                                   max_vocab_size=V + isteps + 64, sign_convention="lorentz", shard=shard, incremental=True)
        itok._get_engine()
        itok.optimize_merges(steps=1, log_every=10 ** 9)           # the one full search
        barrier()
        ti0 = time.perf_counter()
        itok.optimize_merges(steps=isteps - 1, log_every=10 ** 9)
        barrier()
        ti = time.perf_counter() - ti0
        incr = {"merges_per_s": (len(itok.merge_history) - 1) / ti, "steps": isteps - 1,
                "same_merges_as_full_search": itok.merge_history == tok.merge_history[:len(itok.merge_history)],
                "note": "HyperbolicTokenizer(incremental=True): one full search, then one row-vs-all reduction per merge"}

    fp32_form = None
    if rank == 0 and world == 1 and os.environ.get("HM_SCAN_PRECISION", "auto") != "f32":
        from hyptokenizer_amd.engine import MergeEngine
        os.environ["HM_SCAN_PRECISION"] = "f32"
        try:
            e32 = MergeEngine(V + 8, D + 1, "lorentz", device)
            t32 = torch.zeros((V + 8, D + 1), dtype=torch.float32, device=device)
            t32[:V] = X.to(device)
            e32.set_table(t32, V)
            for _ in range(3):
                e32.argmin(CURV, THR)
            e32.scan_totals(reset=True)
            for _ in range(20):
                r32 = e32.argmin(CURV, THR)
            tt = e32.scan_totals()
            ms32 = tt["scan_ms"] / tt["launches"]
            fl32 = 2.0 * (D + 1) * tt["pairs"] / tt["launches"]
            fp32_form = {"bound": "mfma", "achieved": fl32 / (ms32 * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": fl32 / (ms32 * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "kernel": "hm_scan_kernel<NG=25,lorentz,ARGMIN,fp32>", "avg_launch_ms": ms32, "launches": tt["launches"],
                         "nearest_pair": list(r32) if r32 else None}
        finally:
            os.environ.pop("HM_SCAN_PRECISION", None)

    if rank == 0:
        launches = sum(p[2] for p in per_rank)
        avg_ms = sum(p[0] for p in per_rank) / max(launches, 1.0)           # mean launch duration
        flops_per_launch = 2.0 * (D + 1) * sum(p[1] for p in per_rank) / max(launches, 1.0)
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        form = os.environ.get("HM_SCAN_PRECISION", "auto")
        bf16 = form != "f32"                     # auto picks the bf16 prefilter at d = 100
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
        kernel_name = "hm_scan_kernel<KS=7,lorentz,ARGMIN,bf16>" if bf16 else "hm_scan_kernel<NG=25,lorentz,ARGMIN,fp32>"
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
            "dtype": "bf16 prefilter + f32 canonical" if os.environ.get("HM_SCAN_PRECISION", "auto") != "f32" else "f32",
            "data": "synthetic",
            "config": {"workload": f"HyperbolicTokenizer.optimize_merges, full all-pairs search every step, V={V} d={D} "
                                   f"fp32, lorentz sign, thr={THR}, c={CURV}, scale={SCALE}, seed={SEED}",
                       "vocab": V, "dim": D, "merge_threshold": THR, "parallelism": f"rows sharded over {world} rank(s)"},
            "pairwise_dist_GBps_effective": (n_mid * n_mid * 4.0) / (scan_ms_per_step * 1e-3) / 1e9,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "traffic_note": ("bytes per launch past the XCD L2s (FETCH_SIZE corrected x2 + WRITE_SIZE) from "
                                          f"profiles/{traffic_src}; served by the Infinity Cache (the scan image is "
                                          "resident), compulsory bytes = the image once") if traffic else None,
                         "kernel": kernel_name, "avg_launch_ms": avg_ms,
                         "flops_per_launch": flops_per_launch, "launches": launches,
                         "note": "flops = N(N-1)(d+1) algorithmic (triangle); peak = dense MFMA peak of the prefilter's dtype"},
            "roofline_fp32_form": fp32_form,
            "fast_path": fast,
            "incremental": incr,
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
