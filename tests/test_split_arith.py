"""The arithmetic claim behind csrc/ganet_split.h, checked on the CPU with numpy (no GPU): the three-way bf16 split
of an fp32 value is EXACT, and six bf16 x bf16 products accumulated in fp32 reproduce an fp32 GEMM at least as
accurately as a sequential fp32 FMA chain does."""
import numpy as np


def _trunc_bf16(x):
    return (x.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)


def _split3(x):
    a1 = _trunc_bf16(x)
    r = x - a1
    a2 = _trunc_bf16(r)
    a3 = r - a2
    return a1, a2, a3


def test_three_way_split_is_exact_and_every_piece_is_a_bf16():
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.standard_normal(200000).astype(np.float32) * np.exp(rng.uniform(-40, 40, 200000)).astype(np.float32),
        np.array([0.0, -0.0, 1.0, -1.0, 3.0e-39, -1.0e-40, 3.4e38, -3.4e38, np.float32(1) + np.float32(2 ** -23)],
                 np.float32)])
    a1, a2, a3 = _split3(x)
    # the sum is exact (evaluate in float64: no rounding can hide a difference) — for every input
    assert np.array_equal(a1.astype(np.float64) + a2.astype(np.float64) + a3.astype(np.float64), x.astype(np.float64))
    # and every piece of a NORMAL fp32 value is representable in bf16 (its low 16 bits are zero): 8 + 8 + 8
    # significand bits. (A subnormal input keeps only the 7 mantissa bits that lie in the upper half of its fp32 word:
    # the rest, below 2^-132, is dropped by the kernel's pack — irrelevant next to the fp32 rounding of anything
    # it is added to.)
    normal = np.abs(x) >= np.float32(2.0 ** -100)
    for p in (a1, a2, a3):
        assert np.array_equal(_trunc_bf16(p[normal]), p[normal])
    sub = ~normal & (x != 0)
    assert sub.any() and np.all(np.abs(a3[sub] - _trunc_bf16(a3[sub])) < 2.0 ** -132)


def test_six_products_match_an_fp32_gemm():
    rng = np.random.default_rng(1)
    M, K, N = 512, 128, 128
    A = np.log1p(np.exp(rng.standard_normal((M, K)) * 2)).astype(np.float32)       # softplus-like activations
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    # sequential fp32 FMA chain (what v_mfma_f32_32x32x2_f32 computes)
    acc = np.zeros((M, N), np.float32)
    for k in range(K):
        acc = (acc.astype(np.float64) + A[:, k:k + 1].astype(np.float64) * W[None, :, k].astype(np.float64)).astype(np.float32)
    err_f32 = np.abs(acc - ref).max() / np.abs(ref).max()
    # split: per 16-k MFMA step the six products are exact and added into the fp32 accumulator one after the other
    a, w = _split3(A), _split3(W)
    acc = np.zeros((M, N), np.float32)
    for k0 in range(0, K, 16):
        for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):
            p = a[i][:, k0:k0 + 16].astype(np.float64) @ w[j][:, k0:k0 + 16].astype(np.float64).T
            acc = (acc.astype(np.float64) + p).astype(np.float32)
    err_split = np.abs(acc - ref).max() / np.abs(ref).max()
    # a plain bf16 GEMM for scale
    err_bf16 = np.abs(a[0].astype(np.float64) @ w[0].astype(np.float64).T - ref).max() / np.abs(ref).max()
    assert err_split <= 1.25 * err_f32 + 1e-7, (err_split, err_f32)
    assert err_split < 1e-6 < 1e-3 < err_bf16, (err_split, err_bf16)
