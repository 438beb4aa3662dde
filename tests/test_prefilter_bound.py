"""Host-side check of the error bound the bf16 prefilter relies on (hm_scan_delta in
hyptokenizer_amd/csrc/hm_engine.hip): the MFMA sees bf16-rounded spatial coordinates and the time
coordinate split as hi + lo; the scan keeps every pair with u_f < u_hi + delta, so delta must bound
|u_f - u| for EVERY pair or a candidate could be lost.  The kernel cannot run here; its operand
rounding is emulated in numpy (round-to-nearest-even to 8 significant bits, products and sums in
float64 = what an exact accumulator would give; the fp32 accumulation error is the separate
(K + 8) * 2^-23 * max||x||^2 term of the bound, checked against a float32 fmaf-order emulation)."""
import numpy as np
import pytest

from hyptokenizer_amd.synthetic import lorentz_table


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32"""
    b = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    b = (b + 0x7FFF + ((b >> 16) & 1)) >> 16 << 16
    return b.astype(np.uint32).view(np.float32)


def delta_bf16(X, kchunks):
    """the kernel's bound for the bf16 form (kterms = 8 * KC, KC = chunks of 8 K-slots per image row)"""
    X64 = X.astype(np.float64)
    rmax2 = np.float32((X64 ** 2).sum(1).max())
    rmax2s = np.float32((X64[:, 1:] ** 2).sum(1).max())
    d = np.float32(8 * kchunks + 8) * np.float32(1.1920929e-07) * rmax2 * np.float32(1.0001)
    return float(d + np.float32(0.00782) * rmax2s + np.float32(4.7e-5) * rmax2)


@pytest.mark.parametrize("n,d,scale,seed", [(600, 100, 0.05, 1), (600, 100, 0.5, 2), (400, 50, 1.0, 3), (300, 24, 2.0, 4),
                                            (500, 116, 0.2, 5),
                                            # few terms: the roundings of a product's two operands line up (the fuzz case of round 3
                                            # that exceeded the one-operand bound 0.00392 the kernel used until then)
                                            (843, 1, 0.2, 192548318), (900, 1, 1.0, 6), (700, 2, 0.2, 7), (700, 3, 0.5, 8)])
def test_bf16_operand_rounding_stays_inside_delta(n, d, scale, seed):
    X = lorentz_table(n, d, seed=seed, scale=scale).numpy()
    kc = min(v for v in (2, 4, 8, 13, 14, 16) if v >= (d + 4 + 7) // 8)      # hm_pick_kc (hm_engine.hip)
    delta = delta_bf16(X, kc)
    Xs = bf16_rne(X[:, 1:]).astype(np.float64)
    hi = bf16_rne(X[:, 0])
    lo = bf16_rne(X[:, 0] - hi)                                  # x0 ~ hi + lo
    hi, lo = hi.astype(np.float64), lo.astype(np.float64)
    # streamed side [hi, lo, hi, 0] against stationary side [-hi, -hi, -lo, 0]: -(hi*hi' + hi*lo' + lo*hi')
    time_f = np.outer(hi, hi) + np.outer(hi, lo) + np.outer(lo, hi)
    u_f = time_f - Xs @ Xs.T                                     # lorentz sign: u = x0 y0 - <xs, ys>
    X64 = X.astype(np.float64)
    u = np.outer(X64[:, 0], X64[:, 0]) - X64[:, 1:] @ X64[:, 1:].T
    err = np.abs(u_f - u).max()
    assert err <= delta, (err, delta)
    # the bound is not vacuous either: within two orders of magnitude of the worst observed error
    assert delta <= 300 * max(err, 1e-12)


def test_bf16_bound_holds_for_operands_at_the_worst_rounding_point():
    """one spatial coordinate just below a rounding tie (relative error 2^-8 / (1 + 2^-8) on BOTH operands of the product):
    the bound must cover 2 * 2^-8 of ||x_s|| ||y_s||, not 2^-8"""
    xs = np.float32(1.0 + 2.0 ** -8 - 2.0 ** -20)              # rounds down to 1.0
    X = np.array([[np.sqrt(np.float32(1.0) + xs * xs), xs]] * 2, np.float32)
    delta = delta_bf16(X, 2)
    Xs = bf16_rne(X[:, 1:]).astype(np.float64)
    hi = bf16_rne(X[:, 0])
    lo = bf16_rne(X[:, 0] - hi).astype(np.float64)
    hi = hi.astype(np.float64)
    u_f = hi[0] * hi[1] + hi[0] * lo[1] + lo[0] * hi[1] - Xs[0, 0] * Xs[1, 0]
    u = float(X[0, 0]) * float(X[1, 0]) - float(X[0, 1]) * float(X[1, 1])
    err = abs(u_f - u)
    assert err > 0.00392 * float(xs) ** 2                      # beyond the one-operand bound
    assert err <= delta


def test_bf16_rounding_emulation_is_rne():
    x = np.array([1.0, 1.00390625, 1.005859375, 1.001953125, -3.1415927, 1e-30, 65504.0], np.float32)
    r = bf16_rne(x)
    assert r[0] == 1.0 and r[1] == np.float32(1.0) and r[2] == np.float32(1.0078125)    # ties to even, then up
    assert r[3] == np.float32(1.0)
    assert np.all(np.abs(r - x) <= np.abs(x) * 2.0 ** -8)


@pytest.mark.parametrize("d,scale", [(100, 0.05), (100, 1.0), (10, 0.5)])
def test_fp32_chain_vs_canonical_stays_inside_delta(oracle, d, scale):
    """fp32 form: the MFMA's fmaf chain and the canonical torch-order sum are two roundings of the same
    exact value; both within (K + 8) * 2^-23 * max||x||^2 of each other (K = floats per image row)."""
    n = 300
    X = lorentz_table(n, d, seed=7, scale=scale).numpy()
    ng = (d + 3) // 4
    rs = 4 * ng + 4 + (4 if (ng + 1) % 2 == 0 else 0)
    rmax2 = np.float32((X.astype(np.float64) ** 2).sum(1).max())
    delta = float(np.float32(rs + 8) * np.float32(1.1920929e-07) * rmax2 * np.float32(1.0001))
    # fmaf chain in float32: acc = fma(a_k, b_k, acc) over the spatial coordinates, then the time product
    acc = np.zeros((n, n), np.float64)
    for k in range(1, d + 1):
        acc = (acc + np.outer(X[:, k].astype(np.float64), X[:, k].astype(np.float64))).astype(np.float32).astype(np.float64)
    acc = (acc - np.outer(X[:, 0].astype(np.float64), X[:, 0].astype(np.float64))).astype(np.float32)
    u_f = -acc.astype(np.float64)
    sel = np.arange(0, n, 7)
    ii, jj = np.meshgrid(sel, sel, indexing="ij")
    u_c = oracle.minkowski_u(X[ii.ravel()], X[jj.ravel()], 1).astype(np.float64).reshape(len(sel), len(sel))
    assert np.abs(u_f[np.ix_(sel, sel)] - u_c).max() <= delta
