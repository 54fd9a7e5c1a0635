#!/usr/bin/env python3
"""Generate the erfcx polynomial of csrc/ceg_math.h (60-digit mpmath): Chebyshev interpolant of
erfcx(2/u - 2) on u in [2/(2+XMAX), 1], converted to monomials in t = (2u - (a+b))/(b-a).
Prints the C++ Horner body and the max relative error of a double-precision Horner evaluation."""
import sys
import mpmath as mp
import numpy as np

mp.mp.dps = 60
XMAX, DEG = mp.mpf(5), 18


def erfcx(x):
    return mp.exp(x * x) * mp.erfc(x)


a, b = 2 / (2 + XMAX), mp.mpf(1)
N = DEG + 1
nodes = [mp.cos(mp.pi * (k + mp.mpf(1) / 2) / N) for k in range(N)]
vals = [erfcx(2 / ((b - a) / 2 * t + (a + b) / 2) - 2) for t in nodes]
c = [2 * mp.fsum(vals[k] * mp.cos(mp.pi * j * (k + mp.mpf(1) / 2) / N) for k in range(N)) / N for j in range(N)]
c[0] /= 2
T = [[mp.mpf(0)] * N for _ in range(N)]
T[0][0] = mp.mpf(1)
T[1][1] = mp.mpf(1)
for k in range(2, N):
    for j in range(N):
        T[k][j] = (2 * T[k - 1][j - 1] if j > 0 else 0) - T[k - 2][j]
mono = [mp.fsum(c[k] * T[k][j] for k in range(N)) for j in range(N)]
md = [float(v) for v in mono]
s1, s0 = float(2 / (b - a)), float(-(a + b) / (b - a))
xs = np.linspace(0, float(XMAX), 4001)
t = (2.0 / (2.0 + xs)) * s1 + s0
p = np.zeros_like(t)
for coef in md[::-1]:
    p = p * t + coef
ref = np.array([float(erfcx(mp.mpf(float(x)))) for x in xs])
print(f"// t = {s1!r}*u + ({s0!r}); max rel err (double Horner, 4001 samples) = {np.max(np.abs(p / ref - 1)):.2e}", file=sys.stderr)
lines = [f"    double p = {md[-1]!r};"]
for coef in md[-2::-1]:
    lines.append(f"    p = __builtin_fma(p, t, {coef!r});")
print("\n".join(lines))
