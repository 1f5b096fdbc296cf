"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper around oracle/liboracle.so.

The C file (oracle/vcs_oracle.c) is the CPU restatement of the reference's
inter-frame path; every function there cites the reference lines it follows.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product path (vcs-h264_amd/) must never do so.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u8p = ctypes.POINTER(ctypes.c_uint8)
_i16p = ctypes.POINTER(ctypes.c_int16)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "vcs_oracle.c")
    if force or not os.path.exists(so) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def usable_cpus(cap=16):
    """CPUs this process may really use: the affinity mask, clipped by the cgroup's CPU quota
    (a GPU box shows every host core in the mask but grants a share of them) and by `cap`."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        # one OpenMP thread per usable CPU, sleeping (not spinning) between regions: a team as
        # large as the host's core count under a 16-CPU share made every small call cost ~0.4 s
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        if "OMP_NUM_THREADS" not in os.environ:
            os.environ["OMP_NUM_THREADS"] = str(usable_cpus())
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _bgr(*frames):
    for a in frames:
        if a.ndim != 3 or a.shape[2] != 3 or a.shape != frames[0].shape:
            raise ValueError("frames must be H x W x 3 and alike, got %r" % (a.shape,))


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def default_step(bs):
    return round(bs / 3)  # motion.py:132


def num_blocks(H, W, bs):
    return (H // bs) * (W // bs)


def block_coords(H, W, bs):
    out = np.zeros((num_blocks(H, W, bs), 2), np.int32)
    lib().vcso_block_coords(H, W, bs, _p(out, _i32p))
    return out


def me(cur, ref, bs, win=None, step=None, thresh=2000, want_cost=False):
    """motion.py:20-36.  Returns (mv int16 N x 2 [dx,dy], n_static[, cost])."""
    cur = _c(cur, np.uint8)
    ref = _c(ref, np.uint8)
    _bgr(cur, ref)
    H, W = cur.shape[:2]
    win = 2 * bs if win is None else win
    step = default_step(bs) if step is None else step
    N = num_blocks(H, W, bs)
    mv = np.zeros((N, 2), np.int16)
    ns = ctypes.c_uint32(0)
    cost = np.zeros(N, np.uint64) if want_cost else None
    rc = lib().vcso_me(_p(cur, _u8p), _p(ref, _u8p), H, W, bs, win, step,
                       ctypes.c_uint32(thresh), _p(mv, _i16p), ctypes.byref(ns),
                       _p(cost, _u64p))
    if rc:
        raise ValueError("vcso_me rc=%d" % rc)
    return (mv, ns.value, cost) if want_cost else (mv, ns.value)


def me_ext(cur, ref, bs, search="grid", cost="wrap", win=None, step=None, thresh=2000,
           want_cost=False):
    """Extension searches (not in the reference; parity unpinned): vcso_me_ext."""
    cur, ref = _c(cur, np.uint8), _c(ref, np.uint8)
    _bgr(cur, ref)
    H, W = cur.shape[:2]
    win = 2 * bs if win is None else win
    step = default_step(bs) if step is None else step
    N = num_blocks(H, W, bs)
    mv = np.zeros((N, 2), np.int16)
    ns = ctypes.c_uint32(0)
    co = np.zeros(N, np.uint64)
    rc = lib().vcso_me_ext(_p(cur, _u8p), _p(ref, _u8p), H, W, bs, win, step,
                           ctypes.c_uint32(max(0, min(int(thresh), 0xFFFFFFFF))),
                           {"grid": 0, "tss": 1}[search], {"wrap": 0, "sad": 1}[cost],
                           _p(mv, _i16p), ctypes.byref(ns), _p(co, _u64p))
    if rc:
        raise ValueError("vcso_me_ext failed")
    return (mv, ns.value, co) if want_cost else (mv, ns.value)


def mc(ref, mv, bs, coords=None):
    """motion.py:42-69."""
    ref = _c(ref, np.uint8)
    mv = _c(mv, np.int16)
    H, W = ref.shape[:2]
    if coords is not None:
        coords = _c(coords, np.int32)
    pred = np.zeros_like(ref)
    rc = lib().vcso_mc(_p(ref, _u8p), _p(mv, _i16p), _p(coords, _i32p),
                       mv.shape[0], H, W, bs, _p(pred, _u8p))
    if rc:
        raise ValueError("vcso_mc: block out of bounds")
    return pred


def residual(cur, pred):
    """motion.py:38-40."""
    cur = _c(cur, np.uint8)
    pred = _c(pred, np.uint8)
    out = np.zeros_like(cur)
    lib().vcso_residual(_p(cur, _u8p), _p(pred, _u8p), ctypes.c_size_t(cur.size),
                        _p(out, _u8p))
    return out


def wrap_add(a, b):
    a = _c(a, np.uint8)
    b = _c(b, np.uint8)
    out = np.zeros_like(a)
    lib().vcso_wrap_add(_p(a, _u8p), _p(b, _u8p), ctypes.c_size_t(a.size),
                        _p(out, _u8p))
    return out


def bgr2ycrcb(bgr):
    bgr = _c(bgr, np.uint8)
    out = np.zeros_like(bgr)
    lib().vcso_bgr2ycrcb(_p(bgr, _u8p), ctypes.c_size_t(bgr.size // 3), _p(out, _u8p))
    return out


def ycrcb2bgr(ycc):
    ycc = _c(ycc, np.uint8)
    out = np.zeros_like(ycc)
    lib().vcso_ycrcb2bgr(_p(ycc, _u8p), ctypes.c_size_t(ycc.size // 3), _p(out, _u8p))
    return out


def chroma420(bgr):
    """chroma.py:8-21 (boxFilter restated; parity unpinned): (Y, Cr samples, Cb samples)."""
    bgr = _c(bgr, np.uint8)
    H, W = bgr.shape[:2]
    Y = np.zeros((H, W), np.uint8)
    cr = np.zeros(((H + 1) // 2, (W + 1) // 2), np.uint8)
    cb = np.zeros_like(cr)
    lib().vcso_chroma420(_p(bgr, _u8p), H, W, _p(Y, _u8p), _p(cr, _u8p), _p(cb, _u8p))
    return Y, cr, cb


def dct_matrix(n=8):
    D = np.zeros((n, n), np.float64)
    lib().vcso_dct_matrix(n, _p(D, _f64p))
    return D


def make_q(qf=50.0):
    Q = np.zeros((3, 8, 8), np.float64)
    rc = lib().vcso_make_q(ctypes.c_double(qf), _p(Q, _f64p))
    if rc:
        raise ValueError("quality factor must be between 1 and 99")
    return Q


def compress(bgr, Q=None, want_ycc=False):
    """DCTcompressor.py:49-74 -> float64 (3, H, W) [Y, Cr, Cb]."""
    bgr = _c(bgr, np.uint8)
    H, W = bgr.shape[:2]
    Q = make_q() if Q is None else _c(Q, np.float64)
    planes = np.zeros((3, H, W), np.float64)
    ycc = np.zeros_like(bgr) if want_ycc else None
    rc = lib().vcso_compress(_p(bgr, _u8p), H, W, _p(Q, _f64p), _p(planes, _f64p),
                             _p(ycc, _u8p))
    if rc:
        raise ValueError("H and W must be multiples of 8")
    return (planes, ycc) if want_ycc else planes


def decompress(planes, Q=None, want_pre=False):
    """DCTcompressor.py:76-93 -> uint8 BGR H x W x 3."""
    planes = _c(planes, np.float64)
    _, H, W = planes.shape
    Q = make_q() if Q is None else _c(Q, np.float64)
    bgr = np.zeros((H, W, 3), np.uint8)
    pre = np.zeros((3, H, W), np.float64) if want_pre else None
    rc = lib().vcso_decompress(_p(planes, _f64p), H, W, _p(Q, _f64p), _p(bgr, _u8p),
                               _p(pre, _f64p))
    if rc:
        raise ValueError("H and W must be multiples of 8")
    return (bgr, pre) if want_pre else bgr


def encode_p(cur, ref, me_bs, Q=None, win=None, step=None, thresh=2000,
             want_planes=True, want_res=False):
    """encoder.py:49-70 with MotionProcessor(me_bs) -> DCTCompressor(8)."""
    cur = _c(cur, np.uint8)
    ref = _c(ref, np.uint8)
    _bgr(cur, ref)
    H, W = cur.shape[:2]
    win = 2 * me_bs if win is None else win
    step = default_step(me_bs) if step is None else step
    Q = make_q() if Q is None else _c(Q, np.float64)
    N = num_blocks(H, W, me_bs)
    mv = np.zeros((N, 2), np.int16)
    planes = np.zeros((3, H, W), np.float64) if want_planes else None
    res = np.zeros_like(cur) if want_res else None
    ns = ctypes.c_uint32(0)
    rc = lib().vcso_encode_p(_p(cur, _u8p), _p(ref, _u8p), H, W, me_bs, win, step,
                             ctypes.c_uint32(thresh), _p(Q, _f64p), _p(mv, _i16p),
                             _p(planes, _f64p), ctypes.byref(ns), _p(res, _u8p))
    if rc:
        raise ValueError("vcso_encode_p rc=%d" % rc)
    out = [mv, planes, ns.value]
    if want_res:
        out.append(res)
    return tuple(out)


def decode_p(mv, planes, ref, me_bs, Q=None):
    """decoder.py:52-69."""
    mv = _c(mv, np.int16)
    planes = _c(planes, np.float64)
    ref = _c(ref, np.uint8)
    H, W = ref.shape[:2]
    Q = make_q() if Q is None else _c(Q, np.float64)
    out = np.zeros_like(ref)
    rc = lib().vcso_decode_p(_p(mv, _i16p), _p(planes, _f64p), _p(ref, _u8p), H, W,
                             me_bs, _p(Q, _f64p), _p(out, _u8p))
    if rc:
        raise ValueError("vcso_decode_p rc=%d" % rc)
    return out


def round_i16(planes):
    planes = _c(planes, np.float64)
    out = np.zeros(planes.shape, np.int16)
    lib().vcso_round_i16(_p(planes, _f64p), ctypes.c_size_t(planes.size), _p(out, _i16p))
    return out


def intra_luma4x4(Y):
    """intraframe.py:24-151 -> (res, pred, modes) float64."""
    Y = _c(Y, np.uint8)
    H, W = Y.shape
    res, pred = np.zeros((H, W)), np.zeros((H, W))
    modes = np.zeros((H // 4, W // 4))
    rc = lib().vcso_intra_luma4x4(_p(Y, _u8p), H, W, _p(res, _f64p), _p(pred, _f64p),
                                  _p(modes, _f64p))
    if rc:
        raise ValueError("luma4x4 needs H, W multiples of 4 and W >= 8")
    return res, pred, modes


def intra_luma16x16(Y):
    """intraframe.py:153-225."""
    Y = _c(Y, np.uint8)
    H, W = Y.shape
    res, pred = np.zeros((H, W)), np.zeros((H, W))
    modes = np.zeros((H // 16, W // 16))
    rc = lib().vcso_intra_luma16x16(_p(Y, _u8p), H, W, _p(res, _f64p), _p(pred, _f64p),
                                    _p(modes, _f64p))
    if rc:
        raise ValueError("luma16x16 needs H, W multiples of 16")
    return res, pred, modes


def intra_chroma8x8(Cr, Cb):
    """intraframe.py:228-317 -> (Crres, Crpred, Cbres, Cbpred, modes)."""
    Cr = _c(Cr, np.uint8)
    Cb = _c(Cb, np.uint8)
    H, W = Cr.shape
    out = [np.zeros((H, W)) for _ in range(4)]
    modes = np.zeros((H // 8, W // 8))
    rc = lib().vcso_intra_chroma8x8(_p(Cr, _u8p), _p(Cb, _u8p), H, W, _p(out[0], _f64p),
                                    _p(out[1], _f64p), _p(out[2], _f64p), _p(out[3], _f64p),
                                    _p(modes, _f64p))
    if rc:
        raise ValueError("chroma8x8 needs H, W multiples of 8")
    return out[0], out[1], out[2], out[3], modes


def intra_transform4x4(X):
    """intraframe.py:352-358 (commented out upstream): np.round(C . X . C^T) per 4x4 block."""
    X = _c(X, np.float64)
    H, W = X.shape
    out = np.zeros((H, W))
    rc = lib().vcso_intra_transform4x4(_p(X, _f64p), H, W, _p(out, _f64p))
    if rc:
        raise ValueError("transform4x4 needs H, W multiples of 4")
    return out


def set_threads(n):
    """OpenMP thread count for the cpu_baseline timing (bench.py)."""
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(int(n))
    except OSError:
        pass
