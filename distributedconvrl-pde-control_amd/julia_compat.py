"""Julia Base semantics the reference's setup code silently relies on (host-side, setup time)."""
import math
from fractions import Fraction

import numpy as np

_M32 = 16777216  # maxintfloat(Float32) -- Base.rat narrows Float64 to Float32 for its bound


def _rat(x):
    y, a, d, b, c = x, 1, 1, 0, 0
    while abs(y) <= _M32:
        f = math.trunc(y)
        y -= f
        a, c = f * a + c, a
        b, d = f * b + d, b
        if max(abs(a), abs(b)) > _M32:
            return c, d
        if (b != 0 and a / b == x) or y == 0:
            break
        y = 1.0 / y
    return a, b


def _between(a, x, b):
    return a <= x <= b or b <= x <= a


def float_range(start, step, stop):
    """`start:step:stop` for Float64 as Julia evaluates it (twiceprecision.jl).  The element
    COUNT matters for prepare_gaussians (scripts/KS/setup/KSSetup.jl:87): rounding in
    `dx - 50dx` usually makes the range one short of nx+100."""
    start, step, stop = float(start), float(step), float(stop)
    sn, sd = _rat(step)
    if sd != 0 and sn / sd == step:
        an, ad = _rat(start)
        bn, bd = _rat(stop)
        if ad != 0 and bd != 0 and an / ad == start and bn / bd == stop:
            den = ad * sd // math.gcd(ad, sd)
            if den != 0 and abs(start * den) <= 2.0 ** 53 and abs(step * den) <= 2.0 ** 53:
                s_n, t_n = round(start * den), round(step * den)
                ln = max(0, (den * bn - bd * s_n + t_n * bd) // (t_n * bd))
                if _between(start, start + (ln - 1) * step, stop + step / 2) and \
                        not _between(start, start + ln * step, stop):
                    return np.array([float(Fraction(s_n + i * t_n, den)) for i in range(ln)])
    lf = (stop - start) / step
    if lf < 0:
        ln = 0
    elif lf == 0:
        ln = 1
    else:
        ln = int(round(lf)) + 1
        over = start + (ln - 1) * step
        ln -= int(start < stop < over) + int(start > stop > over)
    fs, ft = Fraction(start), Fraction(step)
    return np.array([float(fs + i * ft) for i in range(ln)])
