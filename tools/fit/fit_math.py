#!/usr/bin/env python3
"""dev tool: coefficients of the polynomial kernels behind pg_fast_atan2 / pg_fast_sincos2pi (csrc/pg_fastmath.h).

Fits, in float64, an odd polynomial a * Q(a^2) to atan(a) on [0, 1] and even / odd polynomials to cos / sin (2 pi r) on
r in [-1/4, 1/4] by a Remez exchange on the ABSOLUTE error, then evaluates them the way the kernel does -- float32 Horner with
fused multiply-adds -- and prints the worst absolute error against the float64 function.  Run on the CPU; the printed arrays are
pasted into pg_fastmath.h."""
import numpy as np


def remez(f, basis, lo, hi, n, iters=30):
    """min max |f(x) - sum c_i basis_i(x)| on [lo, hi]; returns c."""
    k = np.arange(n + 1)
    x = 0.5 * (lo + hi) + 0.5 * (hi - lo) * np.cos(np.pi * k / n)[::-1]
    grid = np.linspace(lo, hi, 200001)
    for _ in range(iters):
        A = np.column_stack([b(x) for b in basis] + [(-1.0) ** k])
        sol = np.linalg.solve(A, f(x))
        c = sol[:-1]
        err = f(grid) - sum(ci * b(grid) for ci, b in zip(c, basis))
        # new extrema: local maxima of |err| between sign changes
        idx = [0]
        s = np.sign(err)
        cuts = np.where(s[1:] * s[:-1] < 0)[0]
        segs = np.split(np.arange(len(grid)), cuts + 1)
        ext = [seg[np.argmax(np.abs(err[seg]))] for seg in segs]
        if len(ext) < n + 1:
            break
        ext = sorted(ext, key=lambda i: -abs(err[i]))[: n + 1]
        x = np.sort(grid[ext])
    return c, np.max(np.abs(err))


def f32_fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def horner32(c, s):
    acc = np.full_like(s, np.float32(c[-1]))
    for ci in c[-2::-1]:
        acc = f32_fma(acc, s, np.full_like(s, np.float32(ci)))
    return acc


if __name__ == "__main__":
    np.set_printoptions(precision=17)
    for nt in (7, 8, 9):
        basis = [lambda a, j=j: a ** (2 * j + 1) for j in range(nt)]
        c, e = remez(np.arctan, basis, 0.0, 1.0, nt)
        a = np.linspace(0, 1, 2000001).astype(np.float32)
        s = (a * a).astype(np.float32)
        q = horner32(c, s)
        got = (q * a).astype(np.float32)
        print(f"atan, {nt} terms: remez {e:.3e}, float32 evaluation {np.max(np.abs(got.astype(np.float64) - np.arctan(a.astype(np.float64)))):.3e}")
        print("  ", ", ".join(f"{v:.9e}f" for v in c))
    # sin / cos of 2 pi r, r in [-1/4, 1/4]
    for nt in (4, 5, 6):
        bs = [lambda r, j=j: r ** (2 * j + 1) for j in range(nt)]
        bc = [lambda r, j=j: r ** (2 * j) for j in range(nt + 1)]
        cs_, es = remez(lambda r: np.sin(2 * np.pi * r), bs, 1e-9, 0.25, nt)
        cc_, ec = remez(lambda r: np.cos(2 * np.pi * r), bc, 0.0, 0.25, nt + 1)
        r = np.linspace(-0.25, 0.25, 2000001).astype(np.float32)
        s = (r * r).astype(np.float32)
        sn = (horner32(cs_, s) * r).astype(np.float32)
        cn = horner32(cc_, s)
        print(f"sin {nt} terms: remez {es:.3e}, f32 {np.max(np.abs(sn - np.sin(2 * np.pi * r.astype(np.float64)))):.3e};  "
              f"cos {nt + 1} terms: remez {ec:.3e}, f32 {np.max(np.abs(cn - np.cos(2 * np.pi * r.astype(np.float64)))):.3e}")
        print("   sin:", ", ".join(f"{v:.9e}f" for v in cs_))
        print("   cos:", ", ".join(f"{v:.9e}f" for v in cc_))
