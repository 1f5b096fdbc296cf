"""TEST / MEASUREMENT INFRASTRUCTURE -- not product code (only tests/ and bench.py's
cpu_baseline leg may import anything under oracle/).

A structure-faithful CPU restatement of the reference's inter-frame encode path: the same
Python-level loop nest and the same NumPy call per step as the reference, so that its run
time is tied to the reference's own (within +-20 % of the
1.23 macroblocks/ms the unmodified reference measured for the 16x16 search in the build
container).  It is the slow twin of oracle/vcs_oracle.c and produces the same numbers.

Reference structure followed (file:line under /root/reference/InterframeCompression/):
  motion.py:20-36    one Python iteration per block: find a match, turn it into a vector
  motion.py:109-116  static test: saturating uint8 subtract of the co-located block, np.abs, np.sum
  motion.py:118-152  two nested range() loops over the clipped window with stride round(bs/3);
                     per candidate: slice, uint8 `-` (wraps), np.abs, np.sum, strict `<`
  DCTcompressor.py:63-72, :111-133  per 8x8 tile: rebuild the DCT matrix with a 64-iteration
                     Python loop (once per matmul pair), two np.matmul, np.true_divide
"""
import math

import numpy as np

SIMILARITY_THRESHOLD = 2000  # motion.py:8


def _sat_sub(a, b):
    """cv2.subtract on uint8 (motion.py:111): saturating."""
    return np.where(a > b, a - b, 0).astype(np.uint8)


def find_match(img, block, x, y, bs, win, H, W):
    colocated = img[y:y + bs, x:x + bs]
    if np.sum(np.abs(_sat_sub(colocated, block))) <= SIMILARITY_THRESHOLD:
        return x, y, True
    best, bx, by = 9999999999, 0, 0
    i_min, j_min = max(y - win, 0), max(x - win, 0)
    i_max, j_max = min(y + win, H), min(x + win, W)
    step = round(bs / 3)
    for i in range(i_min, i_max, step):
        if i + bs >= i_max:
            continue
        for j in range(j_min, j_max, step):
            if j + bs >= j_max:
                continue
            cand = img[i:i + bs, j:j + bs]
            diff = np.sum(np.abs(cand - block))  # uint8 wrap-around, not |a - b| (motion.py:146)
            if diff < best:
                best, bx, by = diff, j, i
    return bx, by, False


def me(cur, ref, bs, block_range=None):
    """Motion vectors of the blocks in `block_range` (default: all), raster order."""
    H, W = cur.shape[:2]
    nbx, nby = W // bs, H // bs
    ks = range(nbx * nby) if block_range is None else block_range
    mv, n_static = [], 0
    for k in ks:
        x, y = (k % nbx) * bs, (k // nbx) * bs
        bx, by, st = find_match(ref, cur[y:y + bs, x:x + bs], x, y, bs, 2 * bs, H, W)
        n_static += st
        mv.append([bx - x, by - y])
    return np.array(mv, np.int16).reshape(-1, 2), n_static


def dct_matrix(n=8):
    D = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            D[i, j] = 1 / math.sqrt(n) if i == 0 else \
                math.sqrt(2 / n) * math.cos((2 * j + 1) * i * math.pi / (2 * n))
    return D


def compress_plane(plane_i16, Q8x8, tile_rows=None):
    """DCTcompressor.py:63-72 over one level-shifted channel (rows of tiles in `tile_rows`)."""
    H, W = plane_i16.shape
    out = np.zeros(plane_i16.shape)
    rows = range(0, H, 8) if tile_rows is None else tile_rows
    for i in rows:
        for j in range(0, W, 8):
            D = dct_matrix(8)                       # rebuilt per tile (DCTcompressor.py:112)
            d = np.matmul(np.matmul(D, plane_i16[i:i + 8, j:j + 8]), D.transpose())
            out[i:i + 8, j:j + 8] = np.true_divide(d, Q8x8)
    return out


def timed_sample(cur, ref, bs, Q, budget_s=8.0):
    """Runs the loop on a growing sample of one P frame until `budget_s` is used: a band of
    motion blocks, then the 8x8 tiles under the same band (3 channels).  Returns
    (macroblocks done, seconds for the search, seconds for the DCT of their tiles)."""
    import time
    H, W = cur.shape[:2]
    nbx = W // bs
    done, t_me = 0, 0.0
    t0 = time.perf_counter()
    row = (H // bs) // 2                                # an interior block row: full windows
    while time.perf_counter() - t0 < budget_s * 0.6 and row < H // bs:
        t1 = time.perf_counter()
        me(cur, ref, bs, range(row * nbx, (row + 1) * nbx))
        t_me += time.perf_counter() - t1
        done += nbx
        row += 1
    # the tiles those blocks cover: bs/8 tile rows per block row, three channels
    rows_done = done // nbx
    y0 = ((H // bs) // 2) * bs
    plane = cur[..., 0].astype(np.int16) - 128          # any channel: the cost is data-independent
    t1 = time.perf_counter()
    for ch in range(3):
        compress_plane(plane, Q[ch], range(y0, y0 + rows_done * bs, 8))
    t_dct = time.perf_counter() - t1
    return done, t_me, t_dct
