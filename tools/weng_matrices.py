"""Winograd / Toom-Cook matrices A^T, G, B^T of F(4,3) and F(4,2) in exact rational arithmetic (sympy), with a float64 check against the
direct correlation: the constants of csrc/winograd_eng.hip and lgm_hip/weng.py.  usage: python tools/weng_matrices.py"""
# Toom-Cook matrices for F(m, r) with given interpolation points (last point = infinity), exact fractions.
from fractions import Fraction as Fr
import numpy as np

def matrices(m, r, pts):
    n = m + r - 1
    assert len(pts) == n - 1
    # polynomial evaluation: G (n x r) evaluates filter polynomial at points; BT (n x n); AT (m x n)
    # Following wincnn: AT = transposed evaluation of output poly, G = scaled evaluation, BT from Lagrange basis.
    import sympy as sp
    a = [sp.Rational(p.numerator, p.denominator) for p in pts]
    # f_i = prod_{j != i} (a_i - a_j)
    def f(i):
        v = sp.Integer(1)
        for j in range(n - 1):
            if j != i:
                v *= (a[i] - a[j])
        return v
    x = sp.symbols('x')
    # AT: m x n : AT[i][j] = a_j^i, last column = [0..0,1]
    AT = sp.zeros(m, n)
    for i in range(m):
        for j in range(n - 1):
            AT[i, j] = a[j] ** i
    AT[m - 1, n - 1] = 1
    # G: n x r : G[j][k] = a_j^k / f_j ; last row = [0...0,1]
    G = sp.zeros(n, r)
    for j in range(n - 1):
        for k in range(r):
            G[j, k] = a[j] ** k / f(j)
    G[n - 1, r - 1] = 1
    # BT: n x n : rows = coefficients of Lagrange-like polys: row j (j<n-1): prod_{i != j}(x - a_i) ; last row: prod_i (x - a_i)
    BT = sp.zeros(n, n)
    for j in range(n - 1):
        p = sp.Integer(1)
        for i in range(n - 1):
            if i != j:
                p *= (x - a[i])
        c = sp.Poly(p, x).all_coeffs()[::-1]
        for k, v in enumerate(c):
            BT[j, k] = v
    p = sp.Integer(1)
    for i in range(n - 1):
        p *= (x - a[i])
    c = sp.Poly(p, x).all_coeffs()[::-1]
    for k, v in enumerate(c):
        BT[n - 1, k] = v
    return AT, G, BT

def check(m, r, pts):
    AT, G, BT = matrices(m, r, pts)
    A = np.array(AT.tolist(), dtype=np.float64); Gm = np.array(G.tolist(), dtype=np.float64); B = np.array(BT.tolist(), dtype=np.float64)
    rng = np.random.default_rng(0)
    d = rng.standard_normal(m + r - 1); g = rng.standard_normal(r)
    y = A @ ((Gm @ g) * (B @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    print("1D err", np.abs(y - ref).max())
    return AT, G, BT

if __name__ == "__main__":
    import sympy as sp
    AT, G, BT = check(4, 3, [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2)])
    print("F(4,3)"); sp.pprint(AT); sp.pprint(G); sp.pprint(BT)
    AT, G, BT = check(4, 2, [Fr(0), Fr(1), Fr(-1), Fr(2)])
    print("F(4,2)"); sp.pprint(AT); sp.pprint(G); sp.pprint(BT)
