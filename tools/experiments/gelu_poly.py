"""Sizing of a division-free, exp-free GELU for the bf16 epilogues (profiles/r04_findings.md section 11; not built into the library).

erf(x) = x * P(x^2) on |x| <= A, clamped outside, with P fitted on Chebyshev nodes; evaluated here exactly as the kernel would (float32
Horner, one rounding per fma is emulated by float32 multiply-add) and compared with erf in float64 and with the Abramowitz-Stegun 7.1.26
form the library uses now (Act<BF16>::gelu, common.h).  Prints the coefficients as a C initialiser.
usage: python tools/experiments/gelu_poly.py [degree_in_x2=13] [A=3.92]"""
import sys
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf

deg = int(sys.argv[1]) if len(sys.argv) > 1 else 13
A = float(sys.argv[2]) if len(sys.argv) > 2 else 3.92
nodes = np.cos(np.pi * (np.arange(6000) + 0.5) / 6000)
t = (nodes + 1) / 2 * A * A
x = np.sqrt(t)
f = np.where(x > 0, erf(x) / np.maximum(x, 1e-300), 2 / np.sqrt(np.pi))
cheb = C.chebfit(nodes, f, deg)
# to monomials in t: u = 2 t / A^2 - 1
pu = C.cheb2poly(cheb)
pt = np.zeros(deg + 1)
for k, c in enumerate(pu):                       # c * (2 t / A^2 - 1)^k
    term = np.array([1.0])
    for _ in range(k):
        term = np.convolve(term, np.array([-1.0, 2.0 / (A * A)]))
    pt[:len(term)] += c * term
coef = pt.astype(np.float32)                     # ascending powers of t


def erf_poly32(v):
    xx = np.clip(v.astype(np.float32), np.float32(-A), np.float32(A))
    tt = (xx * xx).astype(np.float32)
    p = np.full_like(tt, coef[-1])
    for c in coef[-2::-1]:
        p = (p * tt + c).astype(np.float32)
    return (p * xx).astype(np.float32)


def erf_as32(v):                                 # Act<BF16>::gelu's erf, float32
    xx = np.abs(v).astype(np.float32)
    tt = (np.float32(1) / (np.float32(1) + np.float32(0.3275911) * xx)).astype(np.float32)
    poly = tt * (np.float32(0.254829592) + tt * (np.float32(-0.284496736) + tt * (np.float32(1.421413741) + tt * (np.float32(-1.453152027) + tt * np.float32(1.061405429)))))
    e = (np.float32(1) - poly.astype(np.float32) * np.exp(-(xx * xx)).astype(np.float32)).astype(np.float32)
    return np.where(v < 0, -e, e).astype(np.float32)


g = np.linspace(-9, 9, 2_000_001)
xs = (g / np.sqrt(2.0))
ref_erf = erf(xs)
ref_gelu = 0.5 * g * (1 + ref_erf)
for name, fn in (('polynomial', erf_poly32), ('Abramowitz-Stegun (library)', erf_as32)):
    e = fn(xs.astype(np.float32)).astype(np.float64)
    gel = 0.5 * g * (1 + e)
    print(f'{name:30s} max |erf error| {np.max(np.abs(e - ref_erf)):.2e}   max |gelu error| {np.max(np.abs(gel - ref_gelu)):.2e}   '
          f'max gelu error relative to max(|gelu|, 2^-9) {np.max(np.abs(gel - ref_gelu) / np.maximum(np.abs(ref_gelu), 2.0 ** -9)):.2e}')
print(f'degree {deg} in x^2 (odd polynomial of degree {2 * deg + 1} in x), |x| clamped to {A}: {deg + 1} fmas + 1 mul + 1 med3 per value')
print('static constexpr float kErfPoly[] = {' + ', '.join(f'{c:.9e}f' for c in coef) + '};   // ascending powers of x^2')
