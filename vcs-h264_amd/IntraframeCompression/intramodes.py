"""Drop-in for the reference's IntraframeCompression/intramodes.py (:7-179): the fifteen
intra predictors as importable functions -- upstream's intraframe.py does
`from intramodes import *` (intraframe.py:7), so callers may reach for them by name.

Same names, arguments and results (float64 arrays) as upstream, NumPy-2 uint8 quirks
included (`u + l` wraps element-wise in dc4x4, `3 * x // 4` wraps when x is a uint8
scalar).  They are tiny host-side definitions for 4 / 8 / 16-sample inputs; the plane-level
work functions of intraframe.py (luma4x4, luma16x16, chroma8x8) never call them -- there the
same arithmetic runs inside the HIP kernels (csrc/intra_kernels.hpp).  Checked against the
reference's own functions on recorded inputs (tests/golden/intramodes_small.npz).

Written from the structure of the predictors rather than pixel by pixel: the directional
4x4 modes are 2- and 3-tap filters sliding along one sequence of neighbours,
    f2(a, b) = a // 2 + b // 2,      f3(a, b, c) = a // 4 + b // 2 + c // 4,
each term floored on its own, as upstream does.
"""
import numpy as np


def _f2(a, b):
    return a // 2 + b // 2


def _f3(a, b, c):
    return a // 4 + b // 2 + c // 4


def _fill(n, fn):
    out = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            out[i, j] = fn(i, j)
    return out


def _rows(n, v):      # every row is the vector above the block
    return _fill(n, lambda i, j: v[j])


def _cols(n, v):      # every column is the vector left of the block
    return _fill(n, lambda i, j: v[i])


# ---- 4x4 (intramodes.py:7-143) -------------------------------------------------------
def vertical4x4(u):
    return _rows(4, u)


def horizontal4x4(l):
    return _cols(4, l)


def dc4x4(u, l):
    return np.ones((4, 4)) * (np.sum(u + l) // 8)     # `u + l` keeps the operands' dtype (:21)


def downleft4x4(u, ur):
    t = list(u) + list(ur)                             # above, then above-right
    def px(i, j):
        k = i + j
        return t[6] // 4 + 3 * t[7] // 4 if k == 6 else _f3(t[k], t[k + 1], t[k + 2])
    return _fill(4, px)


def _corner_sequence(ul, u, l):
    """Left column bottom-up, the corner, the row above left-to-right: l3 l2 l1 l0 ul u0 u1 u2 u3."""
    return [l[3], l[2], l[1], l[0], ul, u[0], u[1], u[2], u[3]]


def downright4x4(ul, u, l):
    s = _corner_sequence(ul, u, l)
    def px(i, j):
        d = j - i                                       # constant along a down-right diagonal
        if d == 0:
            return ul // 4 + u[0] // 2 + l[0] // 4      # upstream's own taps on the main diagonal (:54)
        if d == -1:
            return u[0] // 4 + l[0] // 2 + l[1] // 4    # ... and on the one below it (:58)
        return _f3(s[3 + d], s[4 + d], s[5 + d])
    return _fill(4, px)


def verticalright4x4(ul, u, l):
    s = _corner_sequence(ul, u, l)
    def px(i, j):
        if i == 0:
            return _f2(s[4 + j], s[5 + j])
        if i == 1:
            return _f3(s[3 + j], s[4 + j], s[5 + j])
        if j == 0:                                      # first column of rows 2, 3: down the left edge
            return _f3(s[4 - i], s[5 - i], s[6 - i])
        return px(i - 2, j - 1)                         # the rest repeats two rows up, one to the left
    return _fill(4, px)


def horizontaldown4x4(ul, u, l):
    s = _corner_sequence(ul, u, l)
    def px(i, j):
        if j >= 2 and i >= 1:
            return px(i - 1, j - 2)                     # repeats one row up, two to the left
        if i == 0:
            return _f2(ul, l[0]) if j == 0 else _f3(s[2 + j], s[3 + j], s[4 + j])
        if j == 0:
            return _f2(l[i - 1], l[i])
        if i == 1:
            return ul // 4 + l[1] // 2 + l[2] // 4      # upstream's own taps for this pixel (:96)
        return _f3(l[i - 2], l[i - 1], l[i])
    return _fill(4, px)


def verticalleft4x4(u, ur):
    t = list(u) + list(ur)
    def px(i, j):
        k = j + i // 2
        return _f2(t[k], t[k + 1]) if i % 2 == 0 else _f3(t[k], t[k + 1], t[k + 2])
    return _fill(4, px)


def horizontalup4x4(l):
    def px(i, j):
        z = j + 2 * i
        if z >= 6:
            return l[3]
        if z == 5:
            return l[2] // 4 + 3 * l[3] // 4
        k = z // 2
        return _f2(l[k], l[k + 1]) if z % 2 == 0 else _f3(l[k], l[k + 1], l[k + 2])
    return _fill(4, px)


# ---- 16x16 luma and 8x8 chroma (intramodes.py:145-179) --------------------------------
def vertical16x16(u):
    return _rows(16, u)


def horizontal16x16(l):
    return _cols(16, l)


def dc16x16(u, l):
    return np.ones((16, 16)) * ((np.sum(u) + np.sum(l)) // 32)


def vertical8x8(u):
    return _rows(8, u)


def horizontal8x8(l):
    return _cols(8, l)


def dc8x8(u, l):
    return np.ones((8, 8)) * ((np.sum(u) + np.sum(l)) // 16)
