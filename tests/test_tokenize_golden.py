"""G6: the oracle's tokenize / encode against lists captured from the reference's own tokenize / encode / decode
(tests/golden/make_golden.py g6_tokenize: rules trained by the reference's CLI function plus hand-made multi-level
rules, lines of the reference's data/processed/wikitext103/test.txt)."""
import json
import os

import pytest

from oracle import hm_oracle as O


@pytest.fixture(scope="module")
def g6(golden_dir):
    with open(os.path.join(golden_dir, "g6_tokenize_lorentz.json"), encoding="utf-8") as f:
        return json.load(f)


def token_maps(g6):
    token2idx = {}
    for k, t in enumerate(g6["vocab"]):
        token2idx[t] = k
    return O.merge_rules(g6["merges"]), token2idx


def test_oracle_tokenize_matches_reference(g6):
    rules, token2idx = token_maps(g6)
    assert len(g6["lines"]) >= 30
    for text, toks, ids, dec in zip(g6["lines"], g6["tokens"], g6["ids"], g6["decoded"]):
        assert O.tokenize(rules, text) == toks
        assert O.encode(rules, token2idx, text) == ids
        assert "".join(g6["vocab"][k] for k in ids) == dec
    # multi-level rules really fire in the fixture
    flat = {t for toks in g6["tokens"] for t in toks}
    assert {"the", "ing", "tion", " the"} <= flat


def test_host_tokenize_matches_reference(g6):
    import torch
    from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
    from helpers import OracleEngine
    v = g6["vocab"]
    tok = HyperbolicTokenizer(vocab=list(v), embeddings=torch.nn.Parameter(torch.zeros(len(v), 6)),
                              max_vocab_size=len(v) + 1, device=torch.device("cpu"), engine=OracleEngine(len(v) + 1, 6, "lorentz"))
    tok.merge_history = [tuple(m) for m in g6["merges"]]
    for text, toks, ids in zip(g6["lines"], g6["tokens"], g6["ids"]):
        assert tok.tokenize(text) == toks and tok.encode(text) == ids
