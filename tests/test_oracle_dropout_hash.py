"""The build's train-mode dropout mask (oracle/ditto_oracle.py hash_dropout_mask = csrc/common.h drop_stream / drop_keep): not in the
reference (torch draws its mask from Philox, which a backward kernel cannot regenerate); the oracle defines it, the HIP kernels are
checked against the oracle bit for bit on the GPU (tests/test_gpu_train.py).  Here: the vectorised oracle against a scalar
restatement of the same integer arithmetic, and the statistics a dropout mask needs (rate, independence of neighbours)."""
import numpy as np

from oracle import ditto_oracle as O

M32 = 0xFFFFFFFF


def _lowbias32(h):
    h ^= h >> 16; h = (h * 0x7FEB352D) & M32
    h ^= h >> 15; h = (h * 0x846CA68B) & M32
    h ^= h >> 16
    return h


def _mix24(h):
    h ^= h >> 13
    return ((h & 0xFFFFFF) * 0xD2B74F) & M32


def _keep(seed, layer, bh, i, j, p):
    thr = min(int(float(np.float32(p)) * 4294967296.0), M32)
    lo, hi = seed & M32, (seed >> 32) & M32
    stream = _lowbias32(lo ^ _lowbias32((hi + layer * 0x632BE5AB + bh * 0x9E3779B1) & M32))
    return _mix24(stream ^ ((i * 0x9E3779B1 + j * 0x85EBCA6B) & M32)) >= thr


def test_vectorised_mask_equals_the_scalar_restatement():
    seed, layer, B, H, Sq, Skv, p = 0x1234567890ABCDEF, 3, 2, 3, 37, 29, 0.1
    m = O.hash_dropout_mask(seed, layer, B, H, Sq, Skv, p).numpy()
    for bh in range(B * H):
        for i in range(Sq):
            for j in range(Skv):
                assert m[bh // H, bh % H, i, j] == float(_keep(seed, layer, bh, i, j, p)), (bh, i, j)


def test_mask_statistics():
    """1024 x 1024 masks of 12 (batch, head) streams at p = 0.1: keep rate within 4 sigma of 0.9 per stream, correlation along the
    query axis, the key axis and the diagonal at lags 1, 2, 3, 8, 64 within 4 sigma of 0, the variance of 8x8 block sums within
    3 % of binomial, and two streams uncorrelated."""
    n, p = 1024, 0.1
    m = O.hash_dropout_mask(0xDEADBEEF12345678, 7, 1, 12, n, n, p).numpy()[0].astype(np.float64)
    sig_rate = np.sqrt(p * (1 - p)) / n
    for k in m:
        assert abs(k.mean() - (1 - p)) < 4 * sig_rate
        z = k - k.mean()
        v = z.var()
        for lag in (1, 2, 3, 8, 64):
            for a, b in ((z[:, lag:], z[:, :-lag]), (z[lag:], z[:-lag]), (z[lag:, lag:], z[:-lag, :-lag])):
                assert abs((a * b).mean() / v) < 4.0 / n, lag
        blocks = k.reshape(n // 8, 8, n // 8, 8).sum((1, 3))
        assert abs(blocks.var() / (64 * p * (1 - p)) - 1) < 0.03
    z0, z1 = m[0] - m[0].mean(), m[1] - m[1].mean()
    assert abs((z0 * z1).mean() / (z0.std() * z1.std())) < 4.0 / n
