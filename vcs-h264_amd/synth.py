"""Synthetic frame generators for tests and bench.py.

There is no decodable video in the build environment (the reference's inputs
are H.264 mp4 files and neither cv2 nor ffmpeg is present), so every
measurement runs on `synth_v1`: a blocky random texture plus per-pixel noise,
globally shifted between the reference (I) frame and the current (P) frame and
re-noised.  At 16x16 blocks every block fails the static test (100 % searched,
the worst case for motion search); at 8x8 about 89 % are searched.
"""
import numpy as np


def synth_ref(H, W, seed=1234):
    """The I frame of synth_v1."""
    rng = np.random.default_rng(seed)
    lo = rng.integers(0, 256, (H // 8 + 2, W // 8 + 2, 3))
    base = np.kron(lo, np.ones((8, 8, 1)))[:H, :W]
    ref = np.clip(0.7 * base + rng.integers(0, 77, (H, W, 3)), 0, 255).astype(np.uint8)
    return ref, rng


def synth_v1(H, W, seed=1234, shift=(3, -5)):
    """(ref, cur) pair: cur = ref rolled by `shift` (rows, cols) + noise in [-2, 2]."""
    ref, rng = synth_ref(H, W, seed)
    cur = np.roll(ref, shift, axis=(0, 1))
    cur = np.clip(cur.astype(np.int64) + rng.integers(-2, 3, (H, W, 3)), 0, 255).astype(np.uint8)
    return ref, cur


def gop_shift(k):
    """Global shift of P frame k of a GOP (k >= 1)."""
    return (k % 7 - 3, (2 * k) % 11 - 5)


def synth_gop(H, W, n_p, seed=1234, first=1):
    """I frame + n_p P frames (P frame k uses gop_shift(k), noise seed seed+k).

    Returns (ref H x W x 3 uint8, frames n_p x H x W x 3 uint8).  `first` is the
    GOP index of the first P frame, so that ranks of a multi-GPU run can each
    build their own contiguous slice of one long GOP.
    """
    ref, _ = synth_ref(H, W, seed)
    frames = np.empty((n_p, H, W, 3), np.uint8)
    for n in range(n_p):
        k = first + n
        rng = np.random.default_rng(seed + 7919 * k)
        cur = np.roll(ref, gop_shift(k), axis=(0, 1)).astype(np.int16)
        cur += rng.integers(-2, 3, (H, W, 3), dtype=np.int16)
        frames[n] = np.clip(cur, 0, 255).astype(np.uint8)
    return ref, frames


def synth_static(H, W, seed=1234, bs=16, frac=0.5):
    """Pair in which a random `frac` of the bs x bs blocks are (nearly) static."""
    ref, cur = synth_v1(H, W, seed)
    rng = np.random.default_rng(seed + 1)
    out = cur.copy()
    for y in range(0, H - bs + 1, bs):
        for x in range(0, W - bs + 1, bs):
            if rng.random() < frac:
                blk = ref[y:y + bs, x:x + bs].astype(np.int16)
                blk += rng.integers(-1, 2, blk.shape, dtype=np.int16)
                out[y:y + bs, x:x + bs] = np.clip(blk, 0, 255).astype(np.uint8)
    return ref, out
