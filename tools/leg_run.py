"""Run one or more of bench.py's legs by name on cuda:0 and print their JSON (tools/leg_run.py tokenize_batch ...)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
table = {
    "tokenize_batch": lambda: bench.tokenize_leg(dev),
    "config5_enhanced": lambda: bench.config5_leg(dev),
    "bandwidth": lambda: bench.bandwidth_kernels(dev),
}
for name in sys.argv[1:]:
    print(json.dumps({name: table[name]()}), flush=True)
