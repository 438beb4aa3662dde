"""Randomised GPU-vs-oracle sweep of the batch tokenizer (development aid): python tools/fuzz_tokenize.py [CASES] [SEED]
Random alphabets, rule sets (chains, repeated pairs, results that are not the concatenation, results reachable two
ways), line lengths around the 32-symbol staging rounds, characters outside the vocabulary: token lists, ids and pass
counts must equal the oracle's (= the reference's) Python loop."""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyptokenizer_amd.tokenizer.batch_encoder import BatchEncoder  # noqa: E402
from oracle import hm_oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
bad = 0
for t in range(cases):
    alphabet = rng.sample("abcdefghijklmnopqrstuvwxyz .,-", rng.choice([2, 3, 5, 9, 20]))
    pool, merges = list(alphabet), []
    for _ in range(rng.choice([0, 1, 5, 40, 300, 3000])):
        a, b = rng.choice(pool), rng.choice(pool)
        if len(a) + len(b) > rng.choice([4, 8, 16]):
            continue
        r = rng.random()
        ab = a + b if r < 0.9 else (rng.choice(pool) if r < 0.95 else a + b + "#")
        merges.append((a, b, ab))
        pool.append(ab)
    vocab = ["<pad>", "<bos>", "<eos>", "<unk>"] + sorted(set(pool) - ({rng.choice(alphabet)} if rng.random() < 0.3 else set()))
    t2i = {s: k for k, s in enumerate(vocab)}
    rules = O.merge_rules(merges)
    enc = BatchEncoder(rules, t2i, dev)
    extra = "Zé中\U0001F600" if rng.random() < 0.5 else ""
    lens = [rng.choice([0, 1, 2, 31, 32, 33, 63, 64, 65, 96, 97, 200, 1025]) if rng.random() < 0.7 else rng.randrange(0, 400)
            for _ in range(rng.choice([1, 2, 63, 64, 65, 200]))]
    lines = ["".join(rng.choice(alphabet + list(extra)) for _ in range(n)) for n in lens]
    want = [O.tokenize(rules, s, count_passes=True) for s in lines]
    got = enc.tokenize_batch(lines)
    ids = enc.encode_batch(lines)
    sym, off = enc.symbols(lines)
    _out, _len, passes = enc.run(torch.from_numpy(sym).to(dev), torch.from_numpy(off).to(dev), None, want_passes=True)
    ok = (got == [w[0] for w in want] and ids == [O.encode(rules, t2i, s) for s in lines]
          and passes.cpu().tolist() == [w[1] for w in want])
    if not ok:
        bad += 1
        print(f"case {t}: MISMATCH (alphabet {len(alphabet)}, rules {len(rules)}, lines {len(lines)})", flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
