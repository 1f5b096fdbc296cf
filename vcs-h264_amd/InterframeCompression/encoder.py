"""Encoder -- drop-in for the reference's InterframeCompression/encoder.py
(:9-70): same constructor, `encode_frame`, `encoded_frames`, `ref_frames`.

P frames go through the fused HIP path (one upload, motion search kernel, one
MC + residual + colour + DCT + quantise kernel, one download) instead of four
separate host round trips; the stored Frame is identical to the reference's.

Extension (keyword-only): `dct_block_size` -- the reference hands ONE
block_size to both MotionProcessor and DCTCompressor (encoder.py:16-18), so
with_DCT only works at 8; pass dct_block_size=8 to compose e.g. 16x16 motion
blocks with the 8x8 DCT (the BASELINE workload).

Where the results live.  By DEFAULT (what an unchanged main.py gets) the
coefficient planes of Frame.r stay in HBM: Frame.r is a list-like view of three
planes that downloads on first access, the drop-in Decoder decodes straight
from HBM -- the 8x-frame-size planes never cross PCIe (50 MB of float64 per
1080p frame: 4.7 ms, against 0.3 ms for everything else) --, Frame.mv and
Frame.c are list-like views of integer arrays (frame.RowList: len, indexing,
iteration, ==, tolist) instead of N Python lists, and encode_frame returns once
the frame is enqueued (Frame.mv waits on first access).  main.py:48-50 and
decoder.py:52-69 use nothing else of them.  The reference's aliasing of the
caller's I frame (encoder.py:42) holds: the whole I frame is hashed per P frame.
VCS_RESIDENT=0 (or resident=False) restores real `list` objects and NumPy planes
per frame, downloaded on the spot.  resident=True (explicit) additionally
replaces the per-frame hash by a generation (this encoder, I frame index):
a caller that edits an I frame in place then calls invalidate_ref().
"""
import itertools
import os
import math

import numpy as np

from DCTcompressor import DCTCompressor
from frame import Frame, LazyRowList, RowList
from motion import MotionProcessor, _as_frame
from _runtime import ensure_q, ensure_ref, flush_async, invalidate_ref, submit_async


_SERIAL = itertools.count(1)   # names an Encoder for good (id() values are recycled)


class Encoder:

    def __init__(self, pattern, shape, block_size, with_DCT, *, dct_block_size=None, device=0,
                 resident=None, async_slots=None):
        self.ref_frames = []
        self.encoded_frames = []
        self._serial = next(_SERIAL)
        self.pattern = pattern
        self.ENCODING_PATTERN_LENGTH = len(pattern)
        self.MotionProcessor = MotionProcessor(block_size=block_size, shape=shape, device=device)
        self.DCTCompressor = DCTCompressor(
            block_size=block_size if dct_block_size is None else dct_block_size, device=device)
        self.with_DCT = with_DCT
        self.resident = (os.environ.get("VCS_RESIDENT", "1") != "0") if resident is None else bool(resident)
        # explicit resident=True: the caller vouches for its I frames (generation instead of a hash per P frame)
        self._trust_ref = resident is True or (resident is None and os.environ.get("VCS_RESIDENT") == "1")
        # HBM budget of the planes kept behind Frame.r (49.8 MB per 1080p frame): past it the
        # oldest frames' planes are spilled to host memory (VCS_RESIDENT_MAX_BYTES, default 64 GiB)
        self.resident_max_bytes = int(os.environ.get("VCS_RESIDENT_MAX_BYTES", str(64 << 30)))
        self._resident_bytes = 0
        self._resident_fifo = []
        # resident mode is pipelined: encode_frame returns once the frame is ENQUEUED (its
        # upload overlaps the previous frame's kernels); Frame.mv waits on first access.
        # async_slots=0 keeps one host synchronisation per frame (VCS_ASYNC_SLOTS overrides).
        if async_slots is None:
            async_slots = int(os.environ.get("VCS_ASYNC_SLOTS", "3"))
        self.async_slots = int(async_slots) if self.resident else 0

    def next_buffer(self):
        """The frame feeder's producer side: a page-locked H x W x 3 uint8 array that is safe to
        overwrite NOW (a ring of async_slots + 2 buffers, so the frames still in flight keep
        theirs).  Write the next captured frame into it -- e.g. `cap.read(image=buf)` in an adapted
        main.py -- and hand it to encode_frame: the upload is then a DMA straight out of it."""
        from _runtime import vcs_hip
        if getattr(self, "_ring", None) is None:
            shape = (self.MotionProcessor.shape[0], self.MotionProcessor.shape[1], 3)
            self._ring = [vcs_hip.pinned_empty(shape) for _ in range(max(self.async_slots, 1) + 2)]
            self._ring_pos = 0
        buf = self._ring[self._ring_pos]
        self._ring_pos = (self._ring_pos + 1) % len(self._ring)
        return buf

    def flush(self):
        """Wait for every frame still in flight (pipelined resident mode)."""
        flush_async(self.MotionProcessor._ctx())

    def invalidate_ref(self):
        """Call after editing an I frame IN PLACE when resident=True was asked for explicitly: the
        next P frame then re-uploads its reference (every other mode hashes the whole I frame per P
        frame and needs no such call)."""
        invalidate_ref(self.MotionProcessor._ctx())

    def encode_frame(self, input_frame, frame_num):
        print("Encoding new frame of index", frame_num)
        if frame_num % self.ENCODING_PATTERN_LENGTH == 0:  # encoder.py:25
            encoded_frame = self._process_I_frame(input_frame, frame_num)
        else:
            encoded_frame = self._process_P_frame(input_frame, frame_num)
        print("Encoded frame of type", encoded_frame.t)
        self.encoded_frames.append(encoded_frame)
        return

    #################################################################
    # PRIVATE METHODS

    def _process_I_frame(self, input_frame, frame_num):
        self.ref_frames.append(input_frame)  # aliases the caller's array (encoder.py:42)
        return Frame("I", None, None, None, frame_num, frame_num % self.ENCODING_PATTERN_LENGTH)

    def _process_B_frame(self, input, frame_num):
        return  # not implemented upstream either (encoder.py:45-47)

    def _process_P_frame(self, input, frame_num):
        print("Processing P frame")
        ref_idx = math.floor(frame_num / self.ENCODING_PATTERN_LENGTH)  # encoder.py:51
        ref = self.ref_frames[ref_idx]
        mp, dc = self.MotionProcessor, self.DCTCompressor
        cur = _as_frame(input, mp.shape, "input_frame")
        if self.with_DCT:
            dc._check(mp.shape[0], mp.shape[1])
        ctx = mp._ctx()
        # explicit resident mode: the I frame is identified by (this encoder's serial, its index) -- no
        # per-frame hashing of 6-25 MB; otherwise: full hash (the caller's array is aliased, encoder.py:42)
        ensure_ref(ctx, _as_frame(ref, mp.shape, "ref_frame"),
                   generation=(self._serial, ref_idx) if self._trust_ref else None)
        if self.with_DCT:
            ensure_q(ctx, dc.Q, dc.coeff_mode)
        keep = self.with_DCT and self.resident
        if keep and self.async_slots >= 1:
            return self._submit_P_frame(ctx, mp, cur, frame_num, ref_idx)
        if keep:
            mv, planes, n_static = ctx.store_encode_p(cur)
            res = None
            self._track_resident(planes)
        else:
            mv, planes, res, n_static = ctx.encode_p(cur, want_planes=bool(self.with_DCT),
                                                     want_res=not self.with_DCT)
        mp.last_static_blocks = n_static
        print("Finished processing motion. Got motion vectors.")
        num_static = int(np.count_nonzero((mv == 0).all(axis=1)))
        print("There are", num_static, "static blocks out of", len(mv), "blocks")
        if self.with_DCT:
            print("begin compression")
            r = planes if keep else [planes[0], planes[1], planes[2]]
        else:
            r = res
        if self.resident:  # list-like views, no per-block Python objects (frame.RowList)
            return Frame("P", motion_vectors=RowList(mv.astype(np.int64)), residuals=r,
                         block_coords=RowList(self._coords(mp)), index=frame_num, ref_idx=ref_idx)
        return Frame("P", motion_vectors=mv.astype(np.int64).tolist(), residuals=r,
                     block_coords=mp._block_coords().tolist(), index=frame_num, ref_idx=ref_idx)

    def _track_resident(self, planes):
        self._resident_fifo.append(planes)
        self._resident_bytes += planes.nbytes
        while self._resident_bytes > self.resident_max_bytes and len(self._resident_fifo) > 1:
            old = self._resident_fifo.pop(0)
            self._resident_bytes -= old.nbytes
            old.spill()

    def _submit_P_frame(self, ctx, mp, cur, frame_num, ref_idx):
        """Pipelined form of the P-frame path: upload + kernels + vector download are only
        ENQUEUED here (vcs_encode_p_async); the progress lines that need the result are
        printed when it arrives.  `cur` must not be rewritten in place before then when it
        lives in page-locked memory (vcs_hip.pinned_empty): call flush() first."""
        def done(p):
            mp.last_static_blocks = p.n_static
            # blocks whose vector is (0, 0): one pass over the pairs viewed as 32-bit words
            num_static = len(p.mv) - int(np.count_nonzero(p.mv.view(np.uint32)))
            print("There are", num_static, "static blocks out of", len(p.mv), "blocks")
        planes, pending = submit_async(ctx, cur, self.async_slots, done)
        self._track_resident(planes)
        print("Finished processing motion. Got motion vectors.")
        print("begin compression")
        return Frame("P", motion_vectors=LazyRowList(pending), residuals=planes,
                     block_coords=RowList(self._coords(mp)), index=frame_num, ref_idx=ref_idx)

    def _coords(self, mp):
        c = getattr(self, "_coords_cache", None)
        if c is None:
            c = self._coords_cache = mp._block_coords()  # the same grid for every frame
        return c
