"""TEST INFRASTRUCTURE (like everything under oracle/): numpy restatement of the counter-based generator behind
ditto_noise_normal — Philox4x32-10 (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11;
Random123 constants) and the Box-Muller mapping csrc/rowwise.hip applies to its output.  There is no reference
counterpart (the reference draws torch.randn_like from torch's global generator, src/model/SpeechGenerator.py:141,154):
the algorithm is pinned by the published known-answer vectors of Random123 (tests/test_oracle_golden.py)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
TAG = 0x44695454   # counter word 3 of csrc/rowwise.hip normal4(): "DiTT"


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays (counters) with scalar / array keys; returns the four output words."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint32) for v in (c0, c1, c2, c3))
    k0, k1 = np.asarray(k0, dtype=np.uint32), np.asarray(k1, dtype=np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = k0 + W0, k1 + W1
    return c0, c1, c2, c3


def noise_normal(seed: int, step: int, n: int) -> np.ndarray:
    """The n (multiple of 4) N(0,1) values ditto_noise_normal writes for one utterance: fp64 arithmetic on the same
    24-bit uniforms, so the device values (v_log / v_sin / v_cos) agree to a few 1e-6."""
    assert n % 4 == 0
    quad = np.arange(n // 4, dtype=np.uint64)
    w = philox4x32_10((quad & np.uint64(0xFFFFFFFF)).astype(np.uint32), (quad >> np.uint64(32)).astype(np.uint32),
                      np.uint32(step & 0xFFFFFFFF), np.uint32(TAG), np.uint32(seed & 0xFFFFFFFF),
                      np.uint32((seed >> 32) & 0xFFFFFFFF))
    out = np.empty((n // 4, 4), dtype=np.float64)
    for h in range(2):
        u1 = ((w[2 * h] >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0
        u2 = ((w[2 * h + 1] >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0
        r = np.sqrt(-2.0 * np.log(u1))
        out[:, 2 * h] = r * np.cos(2.0 * np.pi * u2)
        out[:, 2 * h + 1] = r * np.sin(2.0 * np.pi * u2)
    return out.reshape(-1)
