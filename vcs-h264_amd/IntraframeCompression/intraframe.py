"""Drop-in for the three work functions of the reference's
IntraframeCompression/intraframe.py -- luma4x4 (:24-151), luma16x16 (:153-225),
chroma8x8 (:228-317): H.264-style intra prediction with minimum-|error| mode
decision.  Same names, arguments and return values (float64 residual /
prediction planes and mode maps); the per-block Python loops and the
intramodes.py predictors run as HIP kernels (libvcs_hip.so, no CPU fallback).
Results are bit-identical to the reference's under NumPy 2 (its uint8 wrap
quirks included) -- tests/golden/intra_small.npz.

The reference module also runs a matplotlib demo at import and keeps 4x4
integer-transform matrices that its own code never applies (:14-22, :352-358);
neither is part of this path.
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


def _plane(a, name):
    a = np.asarray(a)
    if a.ndim != 2 or a.dtype != np.uint8:
        raise TypeError("%s must be a 2-D uint8 plane (what cv2.split returns)" % name)
    return a


def luma4x4(Y):
    Y = _plane(Y, "Y")
    Yres, Ypred, modes = any_context(*Y.shape).intra_luma(Y, 4)
    return Yres, Ypred, modes


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
