"""Shared plumbing for the drop-in modules: locates vcs_hip.py (one directory
up), caches one libvcs_hip context per (H, W, motion block size, device)."""
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

import vcs_hip  # noqa: E402

_CONTEXTS = {}


def get_context(H, W, me_bs, device=0):
    key = (int(H), int(W), int(me_bs), int(device))
    ctx = _CONTEXTS.get(key)
    if ctx is None:
        ctx = vcs_hip.Context(H, W, me_bs, 8, device)
        ctx._ref_token = None
        ctx._q_token = None
        ctx._pending = None
        _CONTEXTS[key] = ctx
    return ctx


def any_context(H, W, device=0):
    """A context for work that does not depend on the motion block size."""
    for (h, w, _bs, dev), ctx in _CONTEXTS.items():
        if (h, w, dev) == (int(H), int(W), int(device)):
            return ctx
    return get_context(H, W, 8, device)


def release_contexts():
    for ctx in _CONTEXTS.values():
        ctx.close()
    _CONTEXTS.clear()


try:  # fast non-cryptographic hash (several GB/s); zlib.crc32 otherwise
    from xxhash import xxh3_64_intdigest as _hash_bytes
except ImportError:  # pragma: no cover
    from zlib import crc32 as _hash_bytes


def _token(frame):
    """Identity + a hash of EVERY byte of a frame.  The reference aliases the
    caller's I-frame array (encoder.py:42), so an in-place edit of that buffer must
    reach the P frames encoded afterwards: the resident copy in HBM is only reused
    while the whole buffer still hashes the same (0.3-0.6 ms per 1080p frame; the
    frame feeder, which owns its reference, uses an explicit generation instead)."""
    return (frame.shape, _hash_bytes(memoryview(frame).cast("B")))


def ensure_ref(ctx, ref_frame, generation=None):
    """Make `ref_frame` the context's resident reference.  `generation` (any hashable,
    not None): the caller vouches that the same generation means the same bytes --
    no hashing (Encoder(resident=True) keys it on the I frame's index and exposes
    invalidate_ref() for callers that edit the I frame in place)."""
    ref = np.ascontiguousarray(ref_frame)
    tok = ("gen", generation, ref.shape) if generation is not None else _token(ref)
    if ctx._ref_token != tok:
        ctx.set_ref(ref)
        ctx._ref_token = tok


def invalidate_ref(ctx):
    ctx._ref_token = None


def ensure_q(ctx, Q, coeff_mode=0):
    # per-frame call: the token is the tables' bytes (1.5 KB), taken without building a stacked
    # copy when they are float64 arrays already, as the drop-in modules' own tables are
    try:
        tok = (b"".join(p.tobytes() if p.dtype == np.float64 else np.asarray(p, np.float64).tobytes()
                        for p in Q), coeff_mode)
    except AttributeError:
        tok = (b"".join(np.asarray(p, np.float64).tobytes() for p in Q), coeff_mode)
    if ctx._q_token == tok:
        return
    q = np.ascontiguousarray(np.stack([np.asarray(p, np.float64) for p in Q]))
    if q.shape != (3, 8, 8):
        # same failure class as np.true_divide(d, self.Q[channel]) in the reference
        raise ValueError("operands could not be broadcast together: Q tables must be 8x8, got %s"
                         % (q.shape,))
    # Q and coeff_mode only: the motion-search parameters (window, stride,
    # motion.SIMILARITY_THRESHOLD) keep whatever MotionProcessor pushed
    ctx.set_params(Q=q, coeff_mode=coeff_mode)
    ctx._q_token = tok


# ---- pipelined frame feeder (vcs_encode_p_async / vcs_wait) -------------
class Pending:
    """One P frame in flight through the context's staging slots.  complete() blocks until
    its motion vectors are back (idempotent); the input array is kept alive until then."""

    __slots__ = ("ctx", "slot", "cur", "mv", "n_static", "on_done")

    def __init__(self, ctx, slot, cur, on_done=None):
        self.ctx, self.slot, self.cur, self.on_done = ctx, slot, cur, on_done
        self.mv = self.n_static = None

    def complete(self):
        if self.mv is None:
            mv, ns = self.ctx.wait(self.slot)
            self.mv, self.n_static = mv, ns   # int16 (N, 2), as the kernel wrote it
            if self.ctx._pending[self.slot] is self:
                self.ctx._pending[self.slot] = None
            self.cur = None
            if self.on_done is not None:
                self.on_done(self)
        return self.mv


def submit_async(ctx, cur, slots, on_done=None):
    """Enqueue `cur` in the next staging slot of `ctx` (completing whatever still occupies
    it).  Returns (ResidentPlanes, Pending)."""
    if getattr(ctx, "_pending", None) is None or len(ctx._pending) != slots:
        flush_async(ctx)
        ctx.async_reserve(slots)
        ctx._pending = [None] * slots
        ctx._next_slot = 0
    s = ctx._next_slot
    ctx._next_slot = (s + 1) % slots
    if ctx._pending[s] is not None:
        ctx._pending[s].complete()
    planes = ctx.encode_p_async(s, cur)
    p = Pending(ctx, s, cur, on_done)
    ctx._pending[s] = p
    return planes, p


def flush_async(ctx):
    for p in list(getattr(ctx, "_pending", None) or []):
        if p is not None:
            p.complete()
