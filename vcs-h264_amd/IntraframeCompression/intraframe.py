"""Drop-in for the three work functions of the reference's
IntraframeCompression/intraframe.py -- luma4x4 (:24-151), luma16x16 (:153-225),
chroma8x8 (:228-317): H.264-style intra prediction with minimum-|error| mode
decision.  Same names, arguments and return values (float64 residual /
prediction planes and mode maps); the per-block Python loops and the
intramodes.py predictors run as HIP kernels (libvcs_hip.so, no CPU fallback).
Results are bit-identical to the reference's under NumPy 2 (its uint8 wrap
quirks included) -- tests/golden/intra_small.npz.

The reference module also runs a matplotlib demo at import (not part of this
path) and keeps the 4x4 integer core-transform matrices together with a
commented-out loop that would apply them to the luma residual (:14-22,
:352-358).  `transform4x4` below is that loop as a kernel; nothing upstream
runs it, so its parity is pinned only by arithmetic (exact for the
integer-valued residuals the predictors produce), not by a reference output.
"""
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_PKG, os.path.join(_PKG, "InterframeCompression")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from _runtime import any_context  # noqa: E402

blockw = 4
blockh = 4
blocksize = 4

DCTTransform = np.array([[1, 1, 1, 1],          # intraframe.py:14-17
                         [2, 1, -1, -2],
                         [1, -1, -1, 1],
                         [1, -2, 2, -1]])
IDCTTransform = np.array([[1, 1, 1, 0.5],       # intraframe.py:19-22 (kept for completeness,
                          [1, 0.5, -1, -1],     # unused upstream as well)
                          [1, -0.5, -1, 1],
                          [1, -1, 1, -0.5]])


def _plane(a, name):
    a = np.asarray(a)
    if a.ndim != 2 or a.dtype != np.uint8:
        raise TypeError("%s must be a 2-D uint8 plane (what cv2.split returns)" % name)
    return a


def luma4x4(Y):
    Y = _plane(Y, "Y")
    Yres, Ypred, modes = any_context(*Y.shape).intra_luma(Y, 4)
    return Yres, Ypred, modes


def transform4x4(Yres):
    """The commented-out block of intraframe.py:352-358: every 4x4 block of the residual
    plane becomes np.round(DCTTransform @ block @ DCTTransform.T).  Returns a new plane."""
    Yres = np.asarray(Yres, dtype=np.float64)
    if Yres.ndim != 2:
        raise TypeError("Yres must be a 2-D plane")
    return any_context(*Yres.shape).intra_transform4x4(Yres)


def luma16x16(Y):
    Y = _plane(Y, "Y")
    Yres, Ypred, modes = any_context(*Y.shape).intra_luma(Y, 16)
    return Yres, Ypred, modes


def chroma8x8(Cr, Cb):
    Cr, Cb = _plane(Cr, "Cr"), _plane(Cb, "Cb")
    if Cr.shape != Cb.shape:
        raise ValueError("Cr and Cb differ in shape")
    Crres, Crpred, Cbres, Cbpred, modes = any_context(*Cr.shape).intra_chroma(Cr, Cb)
    return Crres, Crpred, Cbres, Cbpred, modes
