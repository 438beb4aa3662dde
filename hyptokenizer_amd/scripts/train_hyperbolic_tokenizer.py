#!/usr/bin/env python
"""Train a hyperbolic tokenizer on the MI355X merge engine.

CLI surface of the reference's ``scripts/train_hyperbolic_tokenizer.py`` (same typer options,
defaults, output files).  Additive options: ``--sign-convention`` (reference | lorentz),
``--init-device`` (where the random initial tangent vectors are drawn; ``cpu`` reproduces a run of
the reference on a CPU-only host bit for bit in the RNG stream) and ``--num-gpus``.

Multi-GPU: launch one process per GPU (``python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 -m hyptokenizer_amd.scripts.train_hyperbolic_tokenizer ...``): with ``WORLD_SIZE > 1`` the
script joins the process group (RCCL, backend "nccl", one GPU per rank by ``LOCAL_RANK``), the candidate search is
row-sharded over the ranks (``hyptokenizer_amd.sharding``), every rank keeps an identical replica of the tokenizer
and rank 0 alone writes the output files.  ``--num-gpus`` is a check that the launch has the intended width.
"""
from __future__ import annotations

import json
import logging
import os
import random
from typing import Any, Dict, List, Optional

import numpy as np
import torch
import typer
from tqdm import tqdm

from hyptokenizer_amd.embedding.lorentz_model import distance, exp_map, project_to_hyperboloid
from hyptokenizer_amd.tokenizer.fast_hyperbolic_merge import FastHyperbolicTokenizer
from hyptokenizer_amd.tokenizer.hyperbolic_merge import TQDM_OFF, HyperbolicTokenizer

logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
logger = logging.getLogger(__name__)


def set_seeds(seed: int = 42) -> None:
    """Reference ``:36-47``."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def load_vocab(vocab_path: str) -> List[str]:
    """One token per line, stripped; empty lines (and the lone space token) are dropped (``:50-61``)."""
    with open(vocab_path, "r", encoding="utf-8") as f:
        return [ln.strip() for ln in f if ln.strip()]


def initialize_embeddings(vocab: List[str], embedding_dim: int, curvature: float = 1.0,
                          device: Optional[torch.device] = None, init_device: Optional[str] = None) -> torch.Tensor:
    """Random points near the origin (reference ``:64-109``): spatial tangent ``randn * 0.01``,
    ``exp_map`` at the origin row by row, then ``project_to_hyperboloid``.  The reference loops over
    rows in Python; here the same per-row arithmetic is one batched kernel."""
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    device = torch.device(device)
    gen_device = device if init_device is None else torch.device(init_device)
    n = len(vocab)
    tangent = torch.zeros((n, embedding_dim + 1), dtype=torch.float32, device=device)
    tangent[:, 1:] = (torch.randn((n, embedding_dim), dtype=torch.float32, device=gen_device) * 0.01).to(device)
    origin = torch.zeros((1, embedding_dim + 1), dtype=torch.float32, device=device)
    origin[0, 0] = 1.0
    points = exp_map(origin.expand(n, -1), tangent, curvature)
    return project_to_hyperboloid(points, curvature)


def _join_process_group(num_gpus: Optional[int]):
    """Under ``torch.distributed.run`` (``WORLD_SIZE > 1``): initialise the process group BEFORE any GPU call of this
    process -- RCCL with one GPU per rank when GPUs are visible, else gloo (CPU tests) -- and return (context, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if num_gpus is not None and num_gpus > 1 and world != num_gpus:
        raise SystemExit(f"--num-gpus {num_gpus} needs a launch with {num_gpus} processes (python -m torch.distributed.run "
                         f"--nproc-per-node {num_gpus} ...); WORLD_SIZE is {world}")
    if world <= 1:
        return None, None
    import torch.distributed as dist
    from hyptokenizer_amd.sharding import ShardContext
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    own = not dist.is_initialized()
    device = None
    if own:
        if torch.cuda.device_count() > 0 and os.environ.get("HM_CLI_BACKEND", "") != "gloo":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            device = torch.device("cuda", local)
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    elif dist.get_backend() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    ctx = ShardContext(device=device)
    ctx.owns_group = own
    return ctx, device


def train_tokenizer(
    vocab_path: str,
    output_dir: str,
    embedding_dim: int = 50,
    curvature: float = 1.0,
    merge_threshold: float = 0.1,
    learning_rate: float = 1e-3,
    merge_steps: int = 100000,
    log_every: int = 1000,
    target_vocab_size: Optional[int] = None,
    seed: int = 42,
    use_fast_tokenizer: bool = True,
    hnsw_m: int = 32,
    hnsw_ef_construction: int = 200,
    hnsw_ef_search: int = 100,
    cache_size: int = 10000,
    rebuild_frequency: int = 100,
    no_faiss: bool = False,
    sign_convention: str = "reference",
    init_device: Optional[str] = None,
    num_gpus: Optional[int] = None,
) -> Dict[str, Any]:
    """Reference ``:112-297``.  Row-sharded over the ranks of the launch when ``WORLD_SIZE > 1`` (every rank runs the
    same seeded program on its replica; rank 0 writes the files)."""
    shard, shard_device = _join_process_group(num_gpus)
    set_seeds(seed)
    device = shard_device if shard_device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    logger.info(f"Using device: {device}" + (f" (rank {shard.rank} of {shard.world}, rows sharded)" if shard else ""))
    vocab = load_vocab(vocab_path)
    logger.info(f"Loaded vocabulary with {len(vocab)} tokens")
    embeddings = initialize_embeddings(vocab, embedding_dim, curvature, device, init_device)
    logger.info(f"Initialized embeddings with shape {embeddings.shape}")

    if use_fast_tokenizer:
        tokenizer = FastHyperbolicTokenizer(
            vocab=vocab, embeddings=torch.nn.Parameter(embeddings), curvature=curvature,
            merge_threshold=merge_threshold, lr=learning_rate, device=device, hnsw_m=hnsw_m,
            hnsw_ef_construction=hnsw_ef_construction, hnsw_ef_search=hnsw_ef_search, cache_size=cache_size,
            rebuild_frequency=rebuild_frequency, use_approximate_search=not no_faiss,
            sign_convention=sign_convention, shard=shard)
        logger.info("Using FastHyperbolicTokenizer with the exact GPU candidate search (no FAISS on this path)")
    else:
        tokenizer = HyperbolicTokenizer(vocab=vocab, embeddings=torch.nn.Parameter(embeddings), curvature=curvature,
                                        merge_threshold=merge_threshold, lr=learning_rate, device=device,
                                        sign_convention=sign_convention, shard=shard)
        logger.info("Using standard HyperbolicTokenizer")
    logger.info("Created hyperbolic tokenizer")

    stats: Dict[str, list] = {"vocab_size": [], "distortion": [], "step": []}

    def log_callback(step: int, tok: HyperbolicTokenizer) -> None:
        """Average distance over <= 1000 sampled tokens (reference ``:201-226``), one kernel call."""
        if step % log_every != 0:
            return
        stats["vocab_size"].append(len(tok.vocab))
        stats["step"].append(step)
        n = min(1000, len(tok.vocab))
        idx = torch.randperm(len(tok.vocab))[:n]
        sample = tok.embeddings.data[idx.to(tok.embeddings.device)]
        dm = distance(sample.unsqueeze(1), sample.unsqueeze(0), tok.curvature, sign_convention=tok.sign_convention)
        iu = torch.triu_indices(n, n, 1, device=dm.device)
        upper = torch.zeros((n, n), device=dm.device)
        upper[iu[0], iu[1]] = dm[iu[0], iu[1]]
        avg = (upper + upper.t()).sum() / (n * (n - 1))
        stats["distortion"].append(avg.item())
        logger.info(f"Step {step}: vocab_size={len(tok.vocab)}, avg_distortion={avg:.4f}")

    logger.info(f"Starting merge optimization for {merge_steps} steps")
    if use_fast_tokenizer:
        tokenizer.optimize_merges(steps=merge_steps, log_every=log_every)
    else:
        # reference ``:236-286``: own loop with callback, target size and x1.05 every 1000 steps
        bar = tqdm(range(merge_steps), desc="Optimizing merges", disable=TQDM_OFF)
        for step in bar:
            log_callback(step, tokenizer)
            if target_vocab_size is not None and len(tokenizer.vocab) >= target_vocab_size:
                logger.info(f"Reached target vocabulary size {target_vocab_size}")
                break
            best = tokenizer._best_candidate()
            if best is None:
                logger.info(f"No more merge candidates found after {step} steps")
                break
            i, j, dist = best
            tokenizer._merge_tokens(i, j)
            bar.set_postfix({"vocab_size": len(tokenizer.vocab), "best_dist": dist,
                             "threshold": tokenizer.merge_threshold})
            if step > 0 and step % 1000 == 0:
                tokenizer.merge_threshold *= 1.05

    if shard is None or shard.rank == 0:          # identical replicas: one writer
        os.makedirs(output_dir, exist_ok=True)
        tokenizer.save(output_dir)
        logger.info(f"Saved tokenizer to {output_dir}")
        with open(os.path.join(output_dir, "training_stats.json"), "w") as f:
            json.dump(stats, f)
    if shard is not None:
        import torch.distributed as dist
        dist.barrier()
        if getattr(shard, "owns_group", False):
            dist.destroy_process_group()
    return stats


def main(
    vocab_path: str = "data/processed/wiki/vocab_initial.txt",
    output_dir: str = "results/hyperbolic/v50000",
    embedding_dim: int = 5,
    curvature: float = 1.0,
    merge_threshold: float = 0.1,
    learning_rate: float = 1e-3,
    merge_steps: int = 100,
    log_every: int = 10,
    target_vocab_size: Optional[int] = 500,
    seed: int = 42,
    use_fast_tokenizer: bool = True,
    hnsw_m: int = 32,
    hnsw_ef_construction: int = 200,
    hnsw_ef_search: int = 100,
    cache_size: int = 10000,
    rebuild_frequency: int = 100,
    no_faiss: bool = False,
    sign_convention: str = "reference",
    init_device: Optional[str] = None,
    num_gpus: Optional[int] = None,
) -> None:
    """Train a hyperbolic tokenizer with the given parameters."""
    train_tokenizer(vocab_path=vocab_path, output_dir=output_dir, embedding_dim=embedding_dim, curvature=curvature,
                    merge_threshold=merge_threshold, learning_rate=learning_rate, merge_steps=merge_steps,
                    log_every=log_every, target_vocab_size=target_vocab_size, seed=seed,
                    use_fast_tokenizer=use_fast_tokenizer, hnsw_m=hnsw_m,
                    hnsw_ef_construction=hnsw_ef_construction, hnsw_ef_search=hnsw_ef_search, cache_size=cache_size,
                    rebuild_frequency=rebuild_frequency, no_faiss=no_faiss, sign_convention=sign_convention,
                    init_device=init_device, num_gpus=num_gpus)


if __name__ == "__main__":
    typer.run(main)
