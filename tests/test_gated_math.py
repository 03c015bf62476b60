"""The gated-MLP activation of the GEMM epilogue, gelu_erf(a) * sigmoid(g) (reference src/components/DiT.py:152-154), is
evaluated in the kernels by a rational erf (csrc/common.h fast_gelu_sigmoid2).  This CPU test pins its CONSTANTS: it parses
them out of the header, restates the formula in numpy float32 (same operation order) and compares with the fp64 value."""
import math
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "ditto_tts_amd", "csrc", "common.h")
f = np.float32


def _constants():
    s = open(SRC).read()
    body = s[s.index("DITTO_DEV f32x2 fast_gelu_sigmoid2(f32x2 x, f32x2 g) {"):]
    body = body[:body.index("\n}\n")]
    # Horner chains as written: first line "u * c6 + c5", then "pn * u + c"
    first = re.search(r"f32x2 pn = u \* ([-0-9.e+]+)f \+ \((-?[0-9.e+-]+)f\);", body)
    rest = re.findall(r"pn = pn \* u \+ ([-0-9.e+]+)f;", body)
    P = [float(first.group(1)), float(first.group(2))] + [float(v) for v in rest]          # highest degree first
    firstq = re.search(r"f32x2 qd = u \* ([-0-9.e+]+)f \+ ([-0-9.e+]+)f;", body)
    restq = re.findall(r"qd = qd \* u \+ ([-0-9.e+]+)f;", body)
    Q = [float(firstq.group(1)), float(firstq.group(2))] + [float(v) for v in restq]
    return P, Q


def _horner(co, u):
    acc = np.full_like(u, f(co[0]))
    for a in co[1:]:
        acc = (acc * u + f(a)).astype(np.float32)
    return acc


def test_rational_erf_constants_and_product():
    P, Q = _constants()
    assert len(P) == 7 and len(Q) == 5 and Q[-1] == 1.0
    erf64 = np.vectorize(math.erf)
    # erf itself
    z = np.linspace(-6, 6, 400001).astype(np.float32)
    zc = np.clip(z, -4, 4).astype(np.float32)
    u = (zc * zc).astype(np.float32)
    got = (zc * _horner(P, u) / _horner(Q, u)).astype(np.float32)
    assert np.abs(got - erf64(z.astype(np.float64))).max() < 6e-7
    assert _horner(Q, u).min() >= 1.0
    # the fused product, as the kernel evaluates it
    xs = np.linspace(-12, 12, 2401).astype(np.float32)
    gs = np.linspace(-100, 30, 261).astype(np.float32)
    X, G = np.meshgrid(xs, gs)
    zl = np.maximum((X * f(0.70710678118654752440)).astype(np.float32), f(-4)).astype(np.float32)
    zc = np.minimum(zl, f(4)).astype(np.float32)
    u = (zc * zc).astype(np.float32)
    Pv, Qv = _horner(P, u), _horner(Q, u)
    num = ((zl * f(0.70710678118654752440)).astype(np.float32) * (zc * Pv + Qv).astype(np.float32)).astype(np.float32)
    with np.errstate(over="ignore"):
        E = np.exp2((G * f(-1.4426950408889634)).astype(np.float32)).astype(np.float32)
        den = (Qv * E + Qv).astype(np.float32)
        out = (num * (f(1) / den)).astype(np.float32)
    X64, G64 = X.astype(np.float64), G.astype(np.float64)
    want = 0.5 * X64 * (1 + erf64(X64 / math.sqrt(2))) / (1 + np.exp(-G64))
    assert np.isfinite(out).all()
    err = np.abs(out - want)
    assert err.max() < 3e-6                                   # |x| <= 12: 4e-7 of erf times |x| / 2
    big = np.abs(want) > 1e-3
    assert (err[big] / np.abs(want[big])).max() < 5e-4        # an eighth of a bf16 rounding step, in the far negative tail
    # the tails: far negative x must decay (the lower-clamped x / 2 bounds it by 2.8 * |residual of the fit|), far
    # positive x must give x * sigmoid(g)
    xt = np.concatenate([-np.logspace(math.log10(5.7), 4, 400), np.logspace(math.log10(5.7), 4, 400)]).astype(np.float32)
    gt = np.array([-3.0, 0.0, 2.5], dtype=np.float32)
    X, G = np.meshgrid(xt, gt)
    zl = np.maximum((X * f(0.70710678118654752440)).astype(np.float32), f(-4)).astype(np.float32)
    zc = np.minimum(zl, f(4)).astype(np.float32)
    u = (zc * zc).astype(np.float32)
    Pv, Qv = _horner(P, u), _horner(Q, u)
    num = ((zl * f(0.70710678118654752440)).astype(np.float32) * (zc * Pv + Qv).astype(np.float32)).astype(np.float32)
    E = np.exp2((G * f(-1.4426950408889634)).astype(np.float32)).astype(np.float32)
    out = (num * (f(1) / (Qv * E + Qv).astype(np.float32))).astype(np.float32)
    neg = X < 0
    assert np.abs(out[neg]).max() < 2e-6
    wantp = X[~neg].astype(np.float64) / (1 + np.exp(-G[~neg].astype(np.float64)))
    assert (np.abs(out[~neg] - wantp) / wantp).max() < 2e-6

