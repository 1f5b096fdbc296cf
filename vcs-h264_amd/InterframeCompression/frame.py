"""Encoded-frame record shared by Encoder and Decoder.

Same constructor keywords and attribute names as the reference's
InterframeCompression/frame.py:1-8 (encoder.py:68-69 builds it by keyword,
decoder.py:62-65 reads .mv/.r/.c/.ref_i, main.py never touches it).
"""
import numpy as np


class Frame:
    __slots__ = ("t", "mv", "r", "c", "i", "ref_i")

    def __init__(self, frame_type, motion_vectors, residuals, block_coords, index, ref_idx):
        self.t = frame_type        # "I" or "P"
        self.mv = motion_vectors   # N x [dx, dy]
        self.r = residuals         # [Y, Cr, Cb] coefficient planes, or the uint8 residual
        self.c = block_coords      # N x [x, y]
        self.i = index
        self.ref_i = ref_idx

    def __repr__(self):
        n = 0 if self.mv is None else len(self.mv)
        return "Frame(%s, index=%s, ref=%s, %d blocks)" % (self.t, self.i, self.ref_i, n)


class RowList:
    """An (N, 2) integer array that reads like the reference's list of [a, b] pairs
    (Frame.mv: [dx, dy]; Frame.c: [x, y] -- motion.py:34,98) without building N Python
    lists up front: len, indexing (int -> [a, b] of Python ints, slice -> list of pairs),
    iteration, == against a list, .tolist(); np.asarray(view) is the array itself (what
    the drop-in Decoder and MotionProcessor use).  Used by Encoder(resident=True); the
    default Encoder stores plain lists, exactly as upstream."""

    __slots__ = ("_a",)

    def __init__(self, array):
        a = np.asarray(array)
        if a.ndim != 2 or a.shape[1] != 2:
            raise ValueError("RowList wraps an (N, 2) array")
        self._a = a

    def __len__(self):
        return self._a.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self._a[i].tolist()
        return [int(self._a[i, 0]), int(self._a[i, 1])]

    def __iter__(self):
        return iter(self._a.tolist())

    def __array__(self, dtype=None, copy=None):
        a = self._a if dtype is None else self._a.astype(dtype, copy=False)
        return a.copy() if copy else a

    def __eq__(self, other):
        if isinstance(other, RowList):
            other = other._a
        try:
            o = np.asarray(other)
            return o.shape == self._a.shape and bool((o == self._a).all())
        except Exception:
            return NotImplemented

    __hash__ = None

    def tolist(self):
        return self._a.tolist()

    def __repr__(self):
        return "RowList(%d x 2)" % len(self)


class LazyRowList(RowList):
    """A RowList whose array is still on its way back from the GPU (the pipelined frame
    feeder, Encoder(resident=True)): the first access of any kind waits for it."""

    __slots__ = ("_pending", "_wide")

    def __init__(self, pending):
        self._pending = pending
        self._wide = None

    @property
    def _a(self):
        if self._wide is None:      # int64 like the default Encoder's vectors, converted once
            self._wide = self._pending.complete().astype(np.int64)
        return self._wide

    def __repr__(self):
        return "LazyRowList(%s)" % ("pending" if self._pending.mv is None else "%d x 2" % len(self))
