"""GELU for the bf16 activation epilogues without v_exp / v_rcp (magvit2_pytorch_amd/csrc/common.h Act<BF16>::gelu):
    gelu(v) = max(v, 0) - 0.5 r(|v|),   r(a) = a erfc(a / sqrt 2)
r is a smooth bump that decays like a Gaussian: on [0, A] it is a degree-10 polynomial in t = 2 a / A - 1 to 1.5e-5 (Horner in t is well
conditioned: every coefficient < 1 in magnitude - unlike erf itself, which needs degree 17+ and cancels in float32, profiles/r04_findings.md
section 11), and beyond A = 4.5 it is < 1e-5 and clamped.  This script fits the polynomial (iteratively re-weighted least squares on
Chebyshev nodes) and evaluates the float32 Horner form the kernel runs against the exact function.  python tools/gelu_bump_fit.py"""
import numpy as np
from math import erfc, erf, sqrt
from numpy.polynomial import chebyshev as Ch
verfc = np.vectorize(erfc); verf = np.vectorize(erf)
def fit(A, deg):
    n = 6000
    t = np.cos(np.pi * (np.arange(n) + 0.5) / n)          # [-1, 1]
    a = (t + 1) * 0.5 * A
    y = a * verfc(a / sqrt(2))
    V = Ch.chebvander(t, deg)
    w = np.ones(n)
    for it in range(80):
        c, *_ = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)
        r = np.abs(V @ c - y)
        w = w * (1 + 4 * r / r.max()); w /= w.mean()
    return c            # Chebyshev coefficients in t = 2 a / A - 1
def eval32(c, A, v, basis='mono'):
    v = v.astype(np.float32)
    a = np.minimum(np.abs(v), np.float32(A)).astype(np.float32)
    if basis == 'mono':       # monomial in s = a (Horner): convert cheb(t) -> poly(t) -> poly(a)
        pt = Ch.cheb2poly(c)
        # t = k a - 1
        k = 2.0 / A
        from numpy.polynomial import polynomial as P
        pa = np.zeros(1)
        base = np.array([-1.0, k])
        pw = np.ones(1)
        for coef in pt:
            pa = P.polyadd(pa, coef * pw)
            pw = P.polymul(pw, base)
        co = pa.astype(np.float32)
        acc = np.full_like(a, co[-1])
        for j in range(len(co) - 2, -1, -1):
            acc = (acc * a + co[j]).astype(np.float32)
        return (np.maximum(v, np.float32(0)) - np.float32(0.5) * acc).astype(np.float32), np.abs(pa).max()
    else:                      # Horner in t = k a - 1 (one extra fma, better conditioned)
        pt = Ch.cheb2poly(c).astype(np.float32)
        t = (a * np.float32(2.0 / A) - np.float32(1)).astype(np.float32)
        acc = np.full_like(a, pt[-1])
        for j in range(len(pt) - 2, -1, -1):
            acc = (acc * t + pt[j]).astype(np.float32)
        return (np.maximum(v, np.float32(0)) - np.float32(0.5) * acc).astype(np.float32), np.abs(pt).max()
v = np.linspace(-9, 9, 600001)
exact = 0.5 * v * (1 + verf(v / sqrt(2)))
for A in (4.0, 4.5, 5.0):
    for deg in (8, 9, 10, 11, 12, 14):
        c = fit(A, deg)
        for basis in ('mono', 't'):
            g, mc = eval32(c, A, v, basis)
            err = np.abs(g - exact)
            print(f'A={A} deg={deg} {basis:4s}: max|coef| {mc:9.2f}  gelu abs err {err.max():.2e} at v={v[err.argmax()]:.2f}')
print('----')
A, deg = 4.5, 10
c = fit(A, deg)
pt = Ch.cheb2poly(c) * -0.5
print('A', A, 'coefficients of P\'(t), t = a * (2/A) - 1, low to high:')
print(', '.join(f'{x:.9e}f' for x in pt))
# check
v32 = v.astype(np.float32)
a = np.minimum(np.abs(v32), np.float32(A)); t = (a * np.float32(2.0 / A) - np.float32(1)).astype(np.float32)
co = pt.astype(np.float32)
acc = np.full_like(a, co[-1])
for j in range(len(co) - 2, -1, -1): acc = (acc * t + co[j]).astype(np.float32)
g = (np.maximum(v32, np.float32(0)) + acc).astype(np.float32)
print('max abs err', np.abs(g - exact).max())
