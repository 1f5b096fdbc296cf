"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Minimal stand-in for the five `cv2` entry points the reference's inter-frame
modules touch, so that `/root/reference/InterframeCompression/{motion,
DCTcompressor,encoder}.py` can be imported UNMODIFIED in the build container
(opencv-python is not installed there and there is no network) by
`oracle/gen_golden.py`.  It is put on sys.path only by that script.

What each function pins / does not pin:
  subtract   motion.py:111           saturating uint8 subtract -- documented,
                                     unambiguous OpenCV semantics.
  resize     DCTcompressor.py:52-53  identity when the size is unchanged
                                     (true for H, W multiples of the block);
                                     anything else raises.
  split      DCTcompressor.py:56     per-channel views.
  cvtColor   DCTcompressor.py:55,92  BGR<->YCrCb 8-bit.  OpenCV itself is a
                                     third-party dependency absent from
                                     /root/reference (version unpinned, no
                                     requirements file): this is a restatement
                                     of OpenCV 4.x's published 8-bit fixed-point
                                     path (imgproc/src/color_yuv.simd.hpp,
                                     RGB2YCrCb_i<uchar> / YCrCb2RGB_i<uchar>).
                                     The conversion is therefore "parity
                                     unpinned" against a real OpenCV binary.
  VideoWriter_fourcc  decoder.py:16  dummy.
"""
import numpy as np

COLOR_BGR2YCR_CB = 36
COLOR_YCR_CB2BGR = 38
COLOR_BGR2RGB = 4

_SHIFT = 14
_R2Y, _G2Y, _B2Y = 4899, 9617, 1868
_YCRI, _YCBI = 11682, 9241
_CR2RI, _CB2GI, _CR2GI, _CB2BI = 22987, -5636, -11698, 29049


def _descale(x):
    return (x + (1 << (_SHIFT - 1))) >> _SHIFT


def subtract(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.dtype == np.uint8 and b.dtype == np.uint8
    return np.clip(a.astype(np.int16) - b.astype(np.int16), 0, 255).astype(np.uint8)


def resize(img, dsize):
    w, h = dsize
    if (h, w) != tuple(img.shape[:2]):
        raise NotImplementedError("cv2 stand-in: resize only as identity")
    return img.copy()


def split(img):
    return [img[:, :, c] for c in range(img.shape[2])]


def cvtColor(img, code):
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
    a = img.astype(np.int32)
    if code == COLOR_BGR2YCR_CB:
        B, G, R = a[..., 0], a[..., 1], a[..., 2]
        Y = _descale(R * _R2Y + G * _G2Y + B * _B2Y)
        delta = 128 << _SHIFT
        Cr = _descale((R - Y) * _YCRI + delta)
        Cb = _descale((B - Y) * _YCBI + delta)
        out = np.stack([Y, Cr, Cb], axis=-1)
    elif code == COLOR_YCR_CB2BGR:
        Y, Cr, Cb = a[..., 0], a[..., 1] - 128, a[..., 2] - 128
        B = Y + _descale(Cb * _CB2BI)
        G = Y + _descale(Cb * _CB2GI + Cr * _CR2GI)
        R = Y + _descale(Cr * _CR2RI)
        out = np.stack([B, G, R], axis=-1)
    else:
        raise NotImplementedError("cv2 stand-in: cvtColor code %r" % (code,))
    return np.clip(out, 0, 255).astype(np.uint8)


def VideoWriter_fourcc(*args):
    return 0
