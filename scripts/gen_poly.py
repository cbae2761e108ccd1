#!/usr/bin/env python3
"""Minimax-style polynomial coefficients for the device exp / sin / cos cores.

Chebyshev-node interpolation in 60-digit mpmath (within a small factor of the
true minimax error), converted to the monomial basis and rounded to float64.
Prints C++ constants for mtg_math.h and the measured max relative errors.
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 60


def cheb_fit(f, a, b, deg):
    """Interpolate f on [a, b] at deg+1 Chebyshev nodes; return monomial coeffs (in x)."""
    n = deg + 1
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    A = mp.matrix(n, n)
    rhs = mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for j in range(n):
            A[i, j] = x ** j
        rhs[i] = f(x)
    c = mp.lu_solve(A, rhs)
    return [c[i] for i in range(n)]


def show(name, coeffs):
    print("// %s" % name)
    for i, c in enumerate(coeffs):
        print("    %s, // [%d]" % (float(c).hex(), i))
    print("   ", ", ".join(repr(float(c)) for c in coeffs))


# exp(r) = 1 + r + r^2 * P(r),  |r| <= ln2/2 (+ slack)
R = mp.log(2) / 2 * mp.mpf("1.0001")
for deg in (8, 9):
    P = cheb_fit(lambda r: (mp.exp(r) - 1 - r) / (r * r) if r != 0 else mp.mpf(1) / 2, -R, R, deg)
    Pd = [float(c) for c in P]
    rs = np.linspace(-float(R), float(R), 200001)
    poly = np.zeros_like(rs)
    for c in Pd[::-1]:
        poly = poly * rs + c
    approx = 1.0 + (rs + rs * rs * poly)
    exact = np.array([float(mp.exp(mp.mpf(float(r)))) for r in rs[::50]])
    err = np.max(np.abs(approx[::50] - exact) / exact)
    print("exp: P degree %d (total degree %d): max rel err %.3e" % (deg, deg + 2, err))
    show("exp P deg %d" % deg, P)

# sin(r) = r + r^3 S(r^2), cos(r) = 1 - r^2/2 + r^4 C(r^2), |r| <= pi/4
Q = mp.pi / 4 * mp.mpf("1.0001")
for deg in (4, 5):
    S = cheb_fit(lambda z: (mp.sin(mp.sqrt(z)) - mp.sqrt(z)) / (z * mp.sqrt(z)) if z != 0 else -mp.mpf(1) / 6,
                 mp.mpf(0), Q * Q, deg)
    C = cheb_fit(lambda z: (mp.cos(mp.sqrt(z)) - 1 + z / 2) / (z * z) if z != 0 else mp.mpf(1) / 24,
                 mp.mpf(0), Q * Q, deg)
    Sd, Cd = [float(c) for c in S], [float(c) for c in C]
    rs = np.linspace(-float(Q), float(Q), 200001)
    z = rs * rs
    ps = np.zeros_like(rs); pc = np.zeros_like(rs)
    for c in Sd[::-1]:
        ps = ps * z + c
    for c in Cd[::-1]:
        pc = pc * z + c
    s_ap = rs + rs * z * ps
    c_ap = 1.0 - 0.5 * z + z * z * pc
    s_ex = np.array([float(mp.sin(mp.mpf(float(r)))) for r in rs[::50]])
    c_ex = np.array([float(mp.cos(mp.mpf(float(r)))) for r in rs[::50]])
    print("sin/cos: degree %d in z: max abs err sin %.3e cos %.3e" % (
        deg, np.max(np.abs(s_ap[::50] - s_ex)), np.max(np.abs(c_ap[::50] - c_ex))))
    show("sin S deg %d" % deg, S)
    show("cos C deg %d" % deg, C)

# pi/2 split in three parts: 30 + 30 + 53 bits (k * p1, k * p2 exact for k < 2^23)
pio2 = mp.pi / 2
def trunc_bits(x, bits):
    m, e = mp.frexp(x)
    return mp.ldexp(mp.floor(mp.ldexp(m, bits)), e - bits)
p1 = trunc_bits(pio2, 30)
p2 = trunc_bits(pio2 - p1, 30)
p3 = pio2 - p1 - p2
print("pio2 parts:", float(p1).hex(), float(p2).hex(), float(p3).hex())
print("          :", repr(float(p1)), repr(float(p2)), repr(float(p3)))
print("residual  :", mp.nstr(pio2 - (mp.mpf(float(p1)) + mp.mpf(float(p2)) + mp.mpf(float(p3))), 5))
print("2/pi      :", float(2 / mp.pi).hex(), repr(float(2 / mp.pi)))
ln2 = mp.log(2)
l1 = trunc_bits(ln2, 32)
l2 = ln2 - l1
print("ln2 parts :", float(l1).hex(), float(l2).hex(), repr(float(l1)), repr(float(l2)))
print("log2e     :", float(1 / ln2).hex(), repr(float(1 / ln2)))
