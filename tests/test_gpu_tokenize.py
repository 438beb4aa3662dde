"""hm_tokenize_batch on the GPU against the reference's captured lists (G6), the oracle on seeded random rule sets,
and size-independent properties at a full-size batch."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import hm_oracle as O

pytestmark = pytest.mark.gpu


def make_encoder(merges, vocab):
    from hyptokenizer_amd.tokenizer.batch_encoder import BatchEncoder
    token2idx = {}
    for k, t in enumerate(vocab):
        token2idx[t] = k
    rules = O.merge_rules(merges)
    return BatchEncoder(rules, token2idx, torch.device("cuda:0")), rules, token2idx


def test_g6_lists_on_the_gpu(golden_dir):
    with open(os.path.join(golden_dir, "g6_tokenize_lorentz.json"), encoding="utf-8") as f:
        g6 = json.load(f)
    enc, _rules, _t2i = make_encoder(g6["merges"], g6["vocab"])
    assert enc.tokenize_batch(g6["lines"]) == g6["tokens"]
    ids = enc.encode_batch(g6["lines"])
    assert ids == g6["ids"]
    assert ["".join(g6["vocab"][k] for k in row) for row in ids] == g6["decoded"]
    # order of the lines in the batch does not matter, duplicates and empty batches are fine
    rev = list(reversed(g6["lines"]))
    assert enc.tokenize_batch(rev) == list(reversed(g6["tokens"]))
    assert enc.tokenize_batch([]) == [] and enc.tokenize_batch(["", ""]) == [[], []]


def test_tokenizer_class_batch_methods(golden_dir):
    from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
    with open(os.path.join(golden_dir, "g6_tokenize_lorentz.json"), encoding="utf-8") as f:
        g6 = json.load(f)
    v = g6["vocab"]
    emb = torch.zeros(len(v), 6)
    emb[:, 0] = 1.0
    tok = HyperbolicTokenizer(vocab=list(v), embeddings=torch.nn.Parameter(emb), max_vocab_size=len(v) + 1, device=torch.device("cuda:0"))
    tok.merge_history = [tuple(m) for m in g6["merges"]]
    assert tok.tokenize_batch(g6["lines"]) == g6["tokens"] == [tok.tokenize(t) for t in g6["lines"]]
    assert tok.encode_batch(g6["lines"]) == g6["ids"]


def random_rules(rng, alphabet, n_rules, max_len=6):
    """Rule sets with chains (results reused as operands), duplicate pairs and results reachable two ways."""
    pool = list(alphabet)
    merges = []
    for _ in range(n_rules):
        a, b = rng.choice(pool), rng.choice(pool)
        if len(a) + len(b) > max_len:
            continue
        merges.append((a, b, a + b))
        pool.append(a + b)
    # a few rules whose result is NOT the concatenation, and one repeated pair with another result
    merges.append(("a", "a", "b"))
    merges.append((merges[0][0], merges[0][1], "zz"))
    return merges


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_rules_against_the_oracle(seed):
    rng = random.Random(seed)
    alphabet = "abcdefgh "[: 4 + seed * 2] if seed < 3 else "ab"
    merges = random_rules(rng, alphabet, 40 + 150 * seed)
    vocab = ["<pad>", "<bos>", "<eos>", "<unk>"] + sorted({s for m in merges for s in m} | set(alphabet))
    enc, rules, t2i = make_encoder(merges, vocab)
    lines = ["".join(rng.choice(alphabet + ("Zé\U0001F600" if k % 7 == 0 else "")) for _ in range(rng.choice([0, 1, 2, 3, 17, 64, 257, 1000])))
             for k in range(300)]
    got = enc.tokenize_batch(lines)
    want = [O.tokenize(rules, t) for t in lines]
    assert got == want
    assert enc.encode_batch(lines) == [O.encode(rules, t2i, t) for t in lines]
    # pass counts equal the reference's while-loop
    sym, off = enc.symbols(lines)
    out, out_len, passes = enc.run(torch.from_numpy(sym).cuda(), torch.from_numpy(off).cuda(), None, want_passes=True)
    assert passes.cpu().tolist() == [O.tokenize(rules, t, count_passes=True)[1] for t in lines]
    assert out_len.cpu().tolist() == [len(w) for w in want]


def test_full_size_batch_properties():
    """200k lines / ~40M characters: decode(encode(x)) == x, idempotence on re-joined tokens is NOT a property of the
    reference's algorithm, so the checks are the round trip, a sampled comparison with the oracle, and order independence."""
    rng = np.random.default_rng(5)
    alphabet = np.array(list("etaoinshrdlu cmfwyp,."))
    merges = random_rules(random.Random(9), "etaoinshrdlu ", 3000, max_len=8)[:-2]      # concatenating rules only: decode inverts encode
    vocab = ["<pad>", "<bos>", "<eos>", "<unk>"] + sorted({s for m in merges for s in m} | set(alphabet.tolist()))
    enc, rules, t2i = make_encoder(merges, vocab)
    n = 200_000
    lens = rng.integers(0, 400, size=n)
    chars = alphabet[rng.integers(0, len(alphabet), size=int(lens.sum()))]
    joined = "".join(chars.tolist())
    ends = np.cumsum(lens)
    lines = [joined[int(e - k):int(e)] for e, k in zip(ends, lens)]
    ids = enc.encode_batch(lines)
    for k in rng.integers(0, n, size=300).tolist():
        assert ids[k] == O.encode(rules, t2i, lines[k])
    assert all("".join(vocab[q] for q in row) == t for row, t in zip(ids[:20000], lines[:20000]))
    perm = rng.permutation(n)[:5000]
    assert enc.encode_batch([lines[k] for k in perm.tolist()]) == [ids[k] for k in perm.tolist()]
