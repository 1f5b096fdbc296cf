"""MotionProcessor -- drop-in for the reference's InterframeCompression/motion.py
with the per-block Python loops replaced by HIP kernels on MI355X.

Same class, constructor and public methods as motion.py:14-69; results are
bit-identical (tests/test_gpu_parity.py).  The work is done by libvcs_hip.so
through ctypes (vcs_hip.py); there is no CPU fallback.
"""
import numpy as np

from _runtime import ensure_ref, get_context

# motion.py:8-11 -- kept as module constants because callers may read them
SIMILARITY_THRESHOLD = 2000
CHANNELS = 3
WRITE_STATIC_BLOCK = True
# Extensions, NOT in the reference (defaults = the reference's behaviour):
# SEARCH "grid" (motion.py:132-140) or "tss" (three-step search), COST "wrap" (the uint8
# wrap-around sum of motion.py:146) or "sad" (sum |r - c|).  Edit like the constants above.
SEARCH = "grid"
COST = "wrap"


def _as_frame(a, shape, what):
    a = np.asarray(a)
    if a.dtype != np.uint8:
        raise TypeError("%s must be a uint8 BGR frame, got dtype %s" % (what, a.dtype))
    want = (shape[0], shape[1], CHANNELS)
    if a.shape != want:
        raise ValueError("%s has shape %s but the processor was built for %s"
                         % (what, a.shape, want))
    return np.ascontiguousarray(a)


class MotionProcessor:
    def __init__(self, block_size, shape, *, device=0):
        self.block_size = block_size
        self.shape = shape
        self.search_window_size = block_size * 2  # motion.py:18
        self.device = device
        self.last_static_blocks = None  # blocks that took the static early-out

    # -- internals ---------------------------------------------------------
    def _ctx(self):
        ctx = get_context(self.shape[0], self.shape[1], self.block_size, self.device)
        # module constants / attributes may have been edited by hand, as the
        # reference expects (SURVEY.md section 5 "config / flags")
        params = (self.search_window_size, round(self.block_size / 3), SIMILARITY_THRESHOLD, SEARCH, COST)
        if getattr(ctx, "_me_params", None) != params:
            ctx.set_params(win=params[0], step=params[1], sim_thresh=params[2])
            ctx.set_search(SEARCH, COST)
            ctx._me_params = params
        return ctx

    def _block_coords(self):
        """motion.py:74-98: raster grid of full blocks, entries [x, y]."""
        bs = self.block_size
        nby, nbx = self.shape[0] // bs, self.shape[1] // bs
        ys, xs = np.mgrid[0:nby, 0:nbx]
        return np.stack([xs.ravel() * bs, ys.ravel() * bs], axis=1)

    # -- public API (motion.py:20-69) ----------------------------------------
    def process_motion_prediction(self, input_frame, ref_frame):
        ctx = self._ctx()
        cur = _as_frame(input_frame, self.shape, "input_frame")
        ensure_ref(ctx, _as_frame(ref_frame, self.shape, "ref_frame"))
        mv, n_static = ctx.me(cur)
        self.last_static_blocks = n_static
        return [mv.astype(np.int64).tolist(), self._block_coords().tolist()]

    def get_residuals(self, input_frame, reconstructed):
        a = np.asarray(input_frame)
        b = np.asarray(reconstructed)
        if a.dtype != np.uint8 or b.dtype != np.uint8 or a.shape != b.shape or \
                a.size > 3 * self.shape[0] * self.shape[1]:
            return a - b  # not the frame-sized uint8 case the kernels cover
        return self._ctx().residual(a, b, -1)

    def reconstruct_from_motion_vectors(self, motion_vectors, ref_frame, block_coords):
        ctx = self._ctx()
        ref = _as_frame(ref_frame, self.shape, "ref_frame")
        ensure_ref(ctx, ref)
        mv = np.asarray(motion_vectors, np.int64).reshape(-1, 2)
        co = np.asarray(block_coords, np.int64).reshape(-1, 2)
        if len(mv) != len(co):
            raise IndexError("motion_vectors and block_coords differ in length")
        if np.abs(mv).max(initial=0) > 32767:
            raise IndexError("motion vector out of range")
        std = self._block_coords()
        coords = None if (co.shape == std.shape and np.array_equal(co, std)) else co
        img = ctx.mc(mv.astype(np.int16), coords)
        num_static = int(np.count_nonzero((mv == 0).all(axis=1))) if WRITE_STATIC_BLOCK else 0
        print("There are", num_static, "static blocks out of", len(co), "blocks")
        return img
