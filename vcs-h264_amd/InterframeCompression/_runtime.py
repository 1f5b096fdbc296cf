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


def _token(frame):
    """Identity + sparse checksum of a frame: lets the resident I frame be
    reused across the P frames of a GOP without re-hashing 6-25 MB per call."""
    flat = frame.reshape(-1)
    return (id(frame), frame.ctypes.data, frame.shape, int(flat[::61].sum(dtype=np.uint64)))


def ensure_ref(ctx, ref_frame):
    ref = np.ascontiguousarray(ref_frame)
    tok = _token(ref)
    if ctx._ref_token != tok:
        ctx.set_ref(ref)
        ctx._ref_token = tok


def ensure_q(ctx, Q, coeff_mode=0):
    q = np.ascontiguousarray(np.stack([np.asarray(p, np.float64) for p in Q]))
    if q.shape != (3, 8, 8):
        # same failure class as np.true_divide(d, self.Q[channel]) in the reference
        raise ValueError("operands could not be broadcast together: Q tables must be 8x8, got %s"
                         % (q.shape,))
    tok = (q.tobytes(), coeff_mode)
    if ctx._q_token != tok:
        ctx.set_params(Q=q, coeff_mode=coeff_mode)
        ctx._q_token = tok
