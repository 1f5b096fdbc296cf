"""Decoder -- drop-in for the reference's InterframeCompression/decoder.py
(:11-69).  P frames are rebuilt by one HIP kernel (dequantise + IDCT + colour +
motion-compensated add); writing the mp4 still goes through cv2.VideoWriter,
imported lazily so that this module loads without OpenCV.

Extension (keyword-only): `sink` -- a callable that receives every output
frame instead of the cv2.VideoWriter (used by the tests; cv2 is optional).
"""
import numpy as np

from DCTcompressor import DCTCompressor
from motion import MotionProcessor, _as_frame
from _runtime import ensure_q, ensure_ref, vcs_hip

WRITE_REF_FRAMES = True


class Decoder:
    def __init__(self, encoded_frames, fps, shape, ref_frames, block_size, with_DCT, *,
                 dct_block_size=None, device=0, sink=None):
        self.encoded_frames = encoded_frames
        self.fps = fps
        self.shape = shape
        self.ref_frames = ref_frames
        self.MotionProcessor = MotionProcessor(block_size=block_size, shape=shape, device=device)
        self.DCTCompressor = DCTCompressor(
            block_size=block_size if dct_block_size is None else dct_block_size, device=device)
        self.with_DCT = with_DCT
        self.sink = sink
        self.fourcc = None
        if sink is None:
            import cv2  # only needed to write output.mp4 (decoder.py:16,25)
            self.fourcc = cv2.VideoWriter_fourcc(*'X264')

    def reconstruct_video(self, with_residuals):
        out = None
        write = self.sink
        if write is None:
            import cv2
            out = cv2.VideoWriter('output.mp4', self.fourcc, self.fps,
                                  (self.shape[1], self.shape[0]))
            write = out.write
        print("Set up video writer")
        num_ref_seen = 0
        count = 0
        for cur_frame in self.encoded_frames:  # decoder.py:30-44
            if cur_frame.t == "I" and WRITE_REF_FRAMES:
                write(self.ref_frames[num_ref_seen])
                num_ref_seen += 1
            elif cur_frame.t == "P":
                write(self._reconstruct_P_frame(cur_frame, with_residuals))
            count += 1
        print("Finished writing frames of length", count + 1)
        if out is not None:
            out.release()
        return

    #################################################################
    # PRIVATE METHODS

    def _fully_reconstruct(self, residuals, img):
        if self.with_DCT:
            decompressed_residuals = self.DCTCompressor.decompress(
                compressed=residuals, imshape=img.shape)
            return self.MotionProcessor._ctx().residual(img, decompressed_residuals, +1)
        return self.MotionProcessor._ctx().residual(np.asarray(img), np.asarray(residuals), +1)

    def _reconstruct_P_frame(self, cur_frame, with_residuals):
        mp, dc = self.MotionProcessor, self.DCTCompressor
        ref = self.ref_frames[cur_frame.ref_i]
        std = mp._block_coords()
        co = np.asarray(cur_frame.c).reshape(-1, 2)
        fused = with_residuals and self.with_DCT and co.shape == std.shape and \
            np.array_equal(co, std)
        if not fused:
            reconstruct_img = mp.reconstruct_from_motion_vectors(cur_frame.mv, ref, cur_frame.c)
            if with_residuals:
                return self._fully_reconstruct(residuals=cur_frame.r, img=reconstruct_img)
            return reconstruct_img
        # fused: MC + dequantise + IDCT + colour + wrap-add in one kernel
        dc._check(mp.shape[0], mp.shape[1])
        ctx = mp._ctx()
        ensure_ref(ctx, _as_frame(ref, mp.shape, "ref_frame"))
        ensure_q(ctx, dc.Q, dc.coeff_mode)
        mv = np.asarray(cur_frame.mv, np.int64).reshape(-1, 2)
        num_static = int(np.count_nonzero((mv == 0).all(axis=1)))
        print("There are", num_static, "static blocks out of", len(mv), "blocks")
        print("begin decompression")
        r = cur_frame.r
        if isinstance(r, vcs_hip.ResidentPlanes) and r.resident and r._ctx is ctx:
            frame = ctx.store_decode_p(mv.astype(np.int16), r)  # planes never left HBM
        else:
            planes = np.stack([np.asarray(p) for p in r])
            frame = ctx.decode_p(mv.astype(np.int16), planes)
        print("decompression finished")
        return frame
