"""The build's train-mode dropout mask (oracle/ditto_oracle.py hash_dropout_mask = csrc/common.h drop_stream / drop_keep): not in the
reference (torch draws its mask from Philox, which a backward kernel cannot regenerate); the oracle defines it, the HIP kernels are
checked against the oracle bit for bit on the GPU (tests/test_gpu_train.py).  Here: the vectorised oracle against a scalar
restatement of the same integer arithmetic, and the statistics a dropout mask needs: rate, independence of neighbours inside a
stream, and independence BETWEEN the (batch, head, layer, seed) streams (round 3's hash failed the last one: its stream word
was xored in front of a GF(2)-linear fold, so every stream saw the same 24-bit image of (query, key))."""
import itertools

import numpy as np

from oracle import ditto_oracle as O

M32 = 0xFFFFFFFF


def _lowbias32(h):
    h ^= h >> 16; h = (h * 0x7FEB352D) & M32
    h ^= h >> 15; h = (h * 0x846CA68B) & M32
    h ^= h >> 16
    return h


def _keep(seed, layer, bh, i, j, p):
    thr = min(int(float(np.float32(p)) * 4294967296.0), M32)
    lo, hi = seed & M32, (seed >> 32) & M32
    a = _lowbias32(lo ^ _lowbias32((hi + layer * 0x632BE5AB + bh * 0x9E3779B1) & M32))
    b = _lowbias32(a ^ 0x5BD1E995)
    x = (a + ((i * 0x9E3779B1 + j * 0x85EBCA6B) & M32)) & M32
    x ^= x >> 13
    x = (x + b) & M32
    x ^= x >> 9
    return (((x & 0xFFFFFF) * 0xD2B74F) & M32) >= thr


def test_vectorised_mask_equals_the_scalar_restatement():
    seed, layer, B, H, Sq, Skv, p = 0x1234567890ABCDEF, 3, 2, 3, 37, 29, 0.1
    m = O.hash_dropout_mask(seed, layer, B, H, Sq, Skv, p).numpy()
    for bh in range(B * H):
        for i in range(Sq):
            for j in range(Skv):
                assert m[bh // H, bh % H, i, j] == float(_keep(seed, layer, bh, i, j, p)), (bh, i, j)


def test_mask_statistics_inside_a_stream():
    """1024 x 1024 masks of 12 (batch, head) streams at p = 0.1: keep rate within 4 sigma of 0.9 per stream, correlation along the
    query axis, the key axis and the diagonal at lags 1, 2, 3, 8, 64 within 4 sigma of 0, the variance of 8x8 block sums within
    4 % of binomial."""
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
        assert abs(blocks.var() / (64 * p * (1 - p)) - 1) < 0.04       # 3.6 sigma of a variance over 16 384 blocks


def test_streams_are_independent_of_each_other():
    """ALL pairs of the 12 (batch, head) streams, over three (seed, layer) choices, plus pairs across layers and across seeds:
    mask correlation within 4.5 sigma (sigma = 1 / n: 66 pairs x 3 + 24 draws) and rms over the pairs within 1.5 sigma."""
    n, p = 1024, 0.1
    cors = []
    sets = []
    for seed, layer in ((0xDEADBEEF12345678, 7), (0x0123456789ABCDEF, 0), (77, 11)):
        m = O.hash_dropout_mask(seed, layer, 1, 12, n, n, p).numpy()[0].astype(np.float32)
        z = m - m.mean(axis=(1, 2), keepdims=True)
        z /= z.std(axis=(1, 2), keepdims=True)
        sets.append(z)
        for a, b in itertools.combinations(range(12), 2):
            cors.append(float((z[a] * z[b]).mean()))
    for a in range(12):                                   # same (batch, head), other layer / other seed
        cors.append(float((sets[0][a] * sets[1][a]).mean()))
        cors.append(float((sets[1][a] * sets[2][a]).mean()))
    cors = np.array(cors)
    assert np.abs(cors).max() < 4.5 / n, np.abs(cors).max()
    assert np.sqrt((cors ** 2).mean()) < 1.5 / n, np.sqrt((cors ** 2).mean())


def test_no_pair_of_positions_decides_identically_in_every_stream():
    """Pairs of (query, key) positions whose 32-bit hash values COLLIDE in one stream (a 24-bit bottleneck makes ~19 000 such
    pairs per 1024 x 1024 tile unavoidable) must agree in the other streams only at the rate independent decisions do,
    p^2 + (1 - p)^2 = 0.82 — round 3's hash gave 1.0 here."""
    n, p, H = 1024, 0.1, 24
    a, b = O.hash_dropout_streams(0xDEADBEEF12345678, 7, H)
    i = np.arange(n, dtype=np.uint64)[:, None]
    j = np.arange(n, dtype=np.uint64)[None, :]
    ctr = (i * np.uint64(0x9E3779B1) + j * np.uint64(0x85EBCA6B)) & np.uint64(M32)
    h0 = O._elem_hash(a[0], b[0], ctr).ravel()
    order = np.argsort(h0, kind="stable")
    hs = h0[order]
    same = np.nonzero(hs[1:] == hs[:-1])[0]
    assert len(same) > 5000                                # the bottleneck is there; the question is what the other streams do
    ia, ib = order[same], order[same + 1]
    masks = O.hash_dropout_mask(0xDEADBEEF12345678, 7, 1, H, n, n, p).numpy()[0].reshape(H, -1)
    agree = np.array([(masks[s][ia] == masks[s][ib]).mean() for s in range(1, H)])
    want = p * p + (1 - p) ** 2
    assert abs(agree.mean() - want) < 0.01, agree.mean()
    every = np.all(masks[1:, ia] == masks[1:, ib], axis=0).mean()     # pairs that agree in ALL 23 other streams
    assert every < 0.05, every                                         # independent: 0.82 ** 23 = 0.01
