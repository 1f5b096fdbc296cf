"""Encoded-frame record shared by Encoder and Decoder.

Same constructor keywords and attribute names as the reference's
InterframeCompression/frame.py:1-8 (encoder.py:68-69 builds it by keyword,
decoder.py:62-65 reads .mv/.r/.c/.ref_i, main.py never touches it).
"""


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
