"""Helpers that reproduce Julia Base semantics the reference silently relies on.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
from fractions import Fraction
import math

import numpy as np

_M32 = 16777216  # maxintfloat(Float32): Base.rat narrows Float64 -> Float32 for its bound


def _rat(x):
    """Base.rat (base/twiceprecision.jl): continued-fraction rational guess."""
    y = x
    a = d = 1
    b = c = 0
    while abs(y) <= _M32:
        f = math.trunc(y)
        y -= f
        a, c = f * a + c, a
        b, d = f * b + d, b
        if max(abs(a), abs(b)) > _M32:
            return c, d
        if b != 0 and a / b == x:
            break
        if y == 0:
            break
        y = 1.0 / y
    return a, b


def _isbetween(a, x, b):
    return a <= x <= b or b <= x <= a


def julia_float_range(start, step, stop):
    """Elements of Julia's `start:step:stop` for Float64 (base/twiceprecision.jl `(:)`).

    Used by prepare_gaussians (scripts/KS/setup/KSSetup.jl:87): the LENGTH matters --
    rounding in `dx - 50dx` makes the range one element short of the ideal nx+100, which
    changes the periodic wrap of the right tail (291-point support)."""
    start, step, stop = float(start), float(step), float(stop)
    sn, sd = _rat(step)
    if sd != 0 and sn / sd == step:
        an, ad = _rat(start)
        bn, bd = _rat(stop)
        if ad != 0 and bd != 0 and an / ad == start and bn / bd == stop:
            den = ad * sd // math.gcd(ad, sd)
            m = 2.0 ** 53
            if den != 0 and abs(start * den) <= m and abs(step * den) <= m:
                start_n = round(start * den)
                step_n = round(step * den)
                ln = max(0, (den * bn - bd * start_n + step_n * bd) // (step_n * bd))
                if _isbetween(start, start + (ln - 1) * step, stop + step / 2) and \
                        not _isbetween(start, start + ln * step, stop):
                    return np.array([float(Fraction(start_n + i * step_n, den)) for i in range(ln)])
    lf = (stop - start) / step
    if lf < 0:
        ln = 0
    elif lf == 0:
        ln = 1
    else:
        ln = int(round(lf)) + 1
        stop2 = start + (ln - 1) * step
        ln -= int(start < stop < stop2) + int(start > stop > stop2)
    fs, ft = Fraction(start), Fraction(step)
    return np.array([float(fs + i * ft) for i in range(ln)])


def circshift(a, k, axis=0):
    """Julia circshift(a, k): result[i] = a[i - k] (periodic) == numpy.roll(a, k)."""
    return np.roll(a, k, axis=axis)
