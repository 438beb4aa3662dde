"""HIP engine against the oracle AT THE BASELINE SIZES (BASELINE.json configs 2-4: V = 50 000 / 100 000, d = 50 / 100,
both prefilter forms): nearest pair, ordered top-10 000 and exact candidate count bit for bit, and a 20-step merge loop
(pairs and merged rows).  The oracle's OpenMP form (`fast=True`: exact fp32 prefilter, canonical evaluation of the
survivors -- checked against the plain form in tests/test_oracle_golden.py) does one V = 50 000 search in well under a
second on the GPU box's host cores, so these are ordinary `-m gpu` tests, not bench-only checks."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import bits  # noqa: E402
from hyptokenizer_amd.synthetic import cjk_vocab, lorentz_table  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V,d,prefilter", [(50000, 100, "bf16"), (50000, 100, "f32"), (50000, 50, "f32"), (100000, 100, "bf16")])
def test_search_at_benchmark_size_equals_oracle(oracle, V, d, prefilter):
    from hyptokenizer_amd.engine import MergeEngine
    X = lorentz_table(V, d, seed=42, scale=0.05)
    Xn = X.numpy()
    table = torch.zeros((V + 8, d + 1), device="cuda")
    table[:V] = X.cuda()
    eng = MergeEngine(V + 8, d + 1, "lorentz", prefilter=prefilter)
    eng.set_table(table, V)
    s = eng.pair_distance(np.arange(0, 4000), np.arange(4000, 8000), 1.0)
    # the bench's threshold and one inside the lower tail of the distance distribution (hundreds of thousands of candidates)
    for thr in (0.5, float(np.percentile(s, 0.05))):
        od, oi, oj, oc = oracle.pairwise_topk(Xn, V, 1.0, thr, 1, 10000, fast=True)
        a = eng.argmin(1.0, thr)
        if oc == 0:
            assert a is None
            continue
        assert a is not None and (a[1], a[2]) == (int(oi[0]), int(oj[0])) and bits([a[0]])[0] == bits(od)[0], (thr, a)
        dd, ii, jj, cnt = eng.topk(1.0, thr, 10000)
        assert cnt == oc, (thr, cnt, oc)
        assert np.array_equal(ii, oi) and np.array_equal(jj, oj) and np.array_equal(bits(dd), bits(od)), thr
        dd2, ii2, jj2, cnt2 = eng.topk(1.0, thr, 10000, count=False)             # the refresh form (predicted cut, no count)
        assert np.array_equal(ii2, oi) and np.array_equal(jj2, oj) and np.array_equal(bits(dd2), bits(od)), thr
        assert cnt2 in (-1, oc)
        # a row range of the same search (what one rank of a sharded run does)
        r0, r1 = V // 3, V // 3 + 7001
        pd, pi, pj, pc = oracle.pairwise_topk(Xn, V, 1.0, thr, 1, 2000, r0, r1, fast=True)
        gd, gi, gj, gc = eng.topk(1.0, thr, 2000, r0, r1)
        assert gc == pc and np.array_equal(gi, pi) and np.array_equal(gj, pj) and np.array_equal(bits(gd), bits(pd)), thr


@pytest.mark.parametrize("prefilter", ["bf16", "f32"])
def test_twenty_loop_steps_at_benchmark_size_equal_the_oracle_loop(oracle, prefilter):
    """BASELINE configs[1] / [2] as benched: HyperbolicTokenizer.optimize_merges at V = 50 000, d = 100 (device-resident
    batch) against search -> [0] -> midpoint repeated on the oracle: same pairs, same rows bit for bit."""
    from hyptokenizer_amd.tokenizer.hyperbolic_merge import HyperbolicTokenizer
    V, d, thr, steps = 50000, 100, 0.5, 20
    X = lorentz_table(V, d, seed=42, scale=0.05)
    tok = HyperbolicTokenizer(cjk_vocab(V), torch.nn.Parameter(X.clone()), merge_threshold=thr, device=torch.device("cuda"),
                              max_vocab_size=V + steps + 8, sign_convention="lorentz", prefilter=prefilter)
    tok.optimize_merges(steps=steps, log_every=10 ** 9)
    assert len(tok.merge_history) == steps
    Xo = np.zeros((V + steps + 8, d + 1), np.float32)
    Xo[:V] = X.numpy()
    cur = V
    thr32 = float(np.float32(thr))
    for (a, b, ab) in tok.merge_history:
        hd, hi, hj, hc = oracle.pairwise_topk(Xo, cur, 1.0, thr32, 1, 1, fast=True)
        assert hc > 0 and (tok.vocab[int(hi[0])], tok.vocab[int(hj[0])]) == (a, b), (cur, a, b)
        Xo[cur] = oracle.midpoint_batch(Xo, hi[:1], hj[:1], [np.float32(len(b) / (len(a) + len(b)))], 1.0, 1)[0]
        cur += 1
    got = tok.embeddings.data[V:cur].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), Xo[V:cur].view(np.uint32))
