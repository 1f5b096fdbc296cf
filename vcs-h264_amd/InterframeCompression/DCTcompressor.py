"""DCTCompressor -- drop-in for the reference's
InterframeCompression/DCTcompressor.py (class :41-93, tables :11-38): BGR ->
YCrCb, level shift, 8x8 float64 DCT, division by the JPEG tables (no
rounding); and the inverse.  Runs as HIP kernels on MI355X via libvcs_hip.so;
float64 planes are bit-identical to the reference's (tests/test_gpu_parity.py).

Extensions (keyword-only, default off): `qf` picks another quality factor for
this instance (the reference has one module-level QF), `coeff` selects
float32 / rounded-int16 planes.
"""
import numpy as np

from _runtime import any_context, ensure_q, vcs_hip

TEST_COMPRESSOR = False
QUANTIZE = True

QF = 50.0  # quality factor, must be between 1 and 99
_q50 = vcs_hip.make_q(QF)
# JPEG luminance / chrominance tables (DCTcompressor.py:11-27): QF = 50 gives
# scale 1, i.e. the raw tables.
QY = _q50[0].astype(np.int64)
QC = _q50[1].astype(np.int64)
Q = [_q50[0], _q50[1], _q50[2]]

_COEFF = {"f64": 0, "float64": 0, "f32": 1, "float32": 1, "i16": 2, "int16": 2, "round": 2}


class DCTCompressor:

    def __init__(self, block_size, *, device=0, qf=None, coeff="f64"):
        self.blocksize = block_size
        self.compressed = []
        self.Q = Q if qf is None else list(vcs_hip.make_q(qf))
        self.device = device
        self.coeff_mode = _COEFF[coeff]

    def _check(self, H, W):
        if self.blocksize != 8:
            # the reference fails inside np.true_divide / np.multiply (:71, :86)
            raise ValueError("operands could not be broadcast together with shapes (%d,%d) (8,8)"
                             % (self.blocksize, self.blocksize))
        if H % 8 or W % 8:
            raise NotImplementedError(
                "image %dx%d is not a multiple of 8: the reference resamples it with cv2.resize "
                "(DCTcompressor.py:52-53); that resample is not reproduced" % (W, H))

    def _ctx(self, H, W):
        ctx = any_context(H, W, self.device)
        ensure_q(ctx, self.Q, self.coeff_mode)
        return ctx

    def compress(self, input_bgrimg):
        img = np.asarray(input_bgrimg)
        if img.ndim != 3 or img.shape[2] != 3 or img.dtype != np.uint8:
            raise TypeError("compress expects an H x W x 3 uint8 BGR image")
        H, W = img.shape[:2]
        self._check(H, W)
        print("begin compression")
        planes = self._ctx(H, W).compress(img)
        return [planes[0], planes[1], planes[2]]

    def decompress(self, compressed, imshape):
        print("begin decompression")
        H, W = np.asarray(compressed[0]).shape
        self._check(H, W)
        planes = np.stack([np.asarray(p) for p in compressed])
        newBGR = self._ctx(H, W).decompress(planes)
        print("decompression finished")
        return newBGR
