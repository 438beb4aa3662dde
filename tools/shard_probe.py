"""Per-step wall time of the row-sharded standard loop with ONE rank on this GPU (RCCL process group of size 1): what the
sharded step chain costs beside the scan (tools/shard_probe.py [V] [steps])."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyptokenizer_amd.sharding import ShardContext  # noqa: E402
from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table  # noqa: E402
from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
X = lorentz_table(V, 100, seed=42, scale=0.05)
for label, shard, loop in (("single, device loop", None, True), ("sharded(1), device loop", ShardContext(device=torch.device("cuda", 0)), True),
                           ("sharded(1), host-driven steps", ShardContext(device=torch.device("cuda", 0)), False)):
    tok = HyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X.clone()), merge_threshold=0.5, device=torch.device("cuda", 0),
                              max_vocab_size=V + steps + 100, sign_convention="lorentz", shard=shard)
    tok.device_loop = loop
    tok.optimize_merges(steps=40, log_every=10 ** 9)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tok.optimize_merges(steps=steps, log_every=10 ** 9)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{label}: {el / steps * 1e6:.1f} us/step ({steps / el:.0f} merges/s)", flush=True)
dist.destroy_process_group()
