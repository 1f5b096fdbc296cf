"""4:2:0 chroma subsampling on MI355X -- the computation of the reference's
ChromaSubsampling/chroma.py (:8-21), which upstream is a run-at-import script (it reads an
image, subsamples, rebuilds and shows it): BGR -> YCrCb with OpenCV's 8-bit conversion, a 2x2
box filter over Cr and Cb (cv2.boxFilter, ksize=(2, 2)), every second row and column.

    Y, crSamples, cbSamples = subsample420(img)      # the reference's `subsampledImg` list

runs as one HIP kernel (libvcs_hip.so: vcs_chroma420; no CPU fallback).  OpenCV's box filter
and colour conversion are third-party arithmetic restated from OpenCV 4.x: parity unpinned
(DESIGN.md 3); the kernel is bit-identical to the CPU oracle's restatement.
"""
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_PKG, os.path.join(_PKG, "InterframeCompression")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from _runtime import any_context  # noqa: E402

sampRate = 2       # chroma.py:12
ksize = (2, 2)     # chroma.py:13


def subsample420(img):
    img = np.asarray(img)
    if img.ndim != 3 or img.shape[2] != 3 or img.dtype != np.uint8:
        raise TypeError("subsample420 expects an H x W x 3 uint8 BGR image (what cv2.imread returns)")
    Y, cr, cb = any_context(img.shape[0], img.shape[1]).chroma420(img)
    return [Y, cr, cb]
