"""cProfile of the bench's config-5 leg (EnhancedFastHyperbolicTokenizer at V = 100 000)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
pr = cProfile.Profile()
pr.enable()
out = bench.config5_leg(dev)
pr.disable()
print({k: v for k, v in out.items() if k != "note"})
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats("hyptokenizer_amd|randperm|numpy", 30)
st.sort_stats("tottime").print_stats(18)
