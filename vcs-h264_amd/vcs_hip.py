"""ctypes binding of libvcs_hip.so (include/vcs_hip.h) -- the only way the
Python host code reaches the MI355X kernels.  No PyTorch, no CPU fallback:
if the library or a GPU is missing, construction raises.

`Context` is a thin object wrapper over the C handle; array arguments are
NumPy arrays (uint8 frames, int16 motion vectors, float64 planes).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_T_START = __import__("time").time()   # this process's start, for the rendezvous file's freshness test
LIB_PATH = os.environ.get("VCS_HIP_LIB") or os.path.join(_HERE, "lib", "libvcs_hip.so")

VCS_COEFF_F64, VCS_COEFF_F32, VCS_COEFF_I16 = 0, 1, 2
KEEP_THRESH = -(1 << 63)  # VCS_KEEP_THRESH
COEFF_DTYPES = {0: np.float64, 1: np.float32, 2: np.int16}
T_ME, T_ENC_TAIL, T_DEC, T_AUX, T_ME_PREP, T_COUNT = 0, 1, 2, 3, 4, 5

_c_ctx = ctypes.c_void_p
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i16p = ctypes.POINTER(ctypes.c_int16)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_f64p = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p

# name -> (restype, argtypes): every symbol include/vcs_hip.h declares
SIGNATURES = {
    "vcs_abi_version": (ctypes.c_int, []),
    "vcs_build_id": (ctypes.c_char_p, []),
    "vcs_experiment_build": (ctypes.c_int, []),
    "vcs_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "vcs_host_alloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    "vcs_host_free": (ctypes.c_int, [ctypes.c_void_p]),
    "vcs_last_error": (ctypes.c_char_p, [_c_ctx]),
    "vcs_ctx_create": (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.POINTER(_c_ctx)]),
    "vcs_ctx_destroy": (ctypes.c_int, [_c_ctx]),
    "vcs_set_params": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int, ctypes.c_int64, _f64p,
                                      ctypes.c_int]),
    "vcs_set_search": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int]),
    "vcs_make_q": (ctypes.c_int, [ctypes.c_double, _f64p]),
    "vcs_dct_matrix": (ctypes.c_int, [_f64p]),
    "vcs_set_ref": (ctypes.c_int, [_c_ctx, _u8p]),
    "vcs_me": (ctypes.c_int, [_c_ctx, _u8p, _i16p, _u32p, _u64p]),
    "vcs_mc": (ctypes.c_int, [_c_ctx, _i16p, _i32p, ctypes.c_int, _u8p]),
    "vcs_residual": (ctypes.c_int, [_c_ctx, _u8p, _u8p, ctypes.c_int, ctypes.c_size_t, _u8p]),
    "vcs_compress": (ctypes.c_int, [_c_ctx, _u8p, _vp]),
    "vcs_decompress": (ctypes.c_int, [_c_ctx, _vp, _u8p]),
    "vcs_encode_p": (ctypes.c_int, [_c_ctx, _u8p, _i16p, _vp, _u8p, _u32p]),
    "vcs_decode_p": (ctypes.c_int, [_c_ctx, _i16p, _vp, _u8p]),
    "vcs_store_encode_p": (ctypes.c_int, [_c_ctx, _u8p, _i16p, _u32p, ctypes.POINTER(_vp)]),
    "vcs_store_planes": (ctypes.c_int, [_c_ctx, _vp, _vp]),
    "vcs_store_decode_p": (ctypes.c_int, [_c_ctx, _i16p, _vp, _u8p]),
    "vcs_store_free": (ctypes.c_int, [_c_ctx, _vp]),
    "vcs_async_reserve": (ctypes.c_int, [_c_ctx, ctypes.c_int]),
    "vcs_encode_p_async": (ctypes.c_int, [_c_ctx, ctypes.c_int, _u8p, ctypes.POINTER(_vp)]),
    "vcs_wait": (ctypes.c_int, [_c_ctx, ctypes.c_int, _i16p, _u32p]),
    "vcs_shard_frames": (ctypes.c_int, [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int)] * 2),
    "vcs_encode_gop": (ctypes.c_int, [ctypes.POINTER(_c_ctx), ctypes.c_int, _u8p, ctypes.c_int, _i16p,
                                      _vp]),
    "vcs_comm_unique_id": (ctypes.c_int, [_vp, ctypes.c_size_t]),
    "vcs_group_create": (ctypes.c_int, [ctypes.POINTER(_c_ctx), ctypes.c_int, ctypes.POINTER(_vp)]),
    "vcs_group_create_rank": (ctypes.c_int, [_c_ctx, _vp, ctypes.c_int, ctypes.c_int,
                                             ctypes.POINTER(_vp)]),
    "vcs_group_destroy": (ctypes.c_int, [_vp]),
    "vcs_group_last_error": (ctypes.c_char_p, [_vp]),
    "vcs_group_info": (ctypes.c_int, [_vp] + [ctypes.POINTER(ctypes.c_int)] * 4),
    "vcs_group_layout": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "vcs_group_layout_sets": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "vcs_group_use_set": (ctypes.c_int, [_vp, ctypes.c_int]),
    "vcs_group_shard": (ctypes.c_int, [_vp, ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 2),
    "vcs_group_upload": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _u8p]),
    "vcs_group_set_ref": (ctypes.c_int, [_vp, _u8p]),
    "vcs_group_prefetch_ref": (ctypes.c_int, [_vp]),
    "vcs_group_encode": (ctypes.c_int, [_vp]),
    "vcs_group_decode": (ctypes.c_int, [_vp]),
    "vcs_group_sync": (ctypes.c_int, [_vp]),
    "vcs_group_get_mv": (ctypes.c_int, [_vp, _i16p, _u32p]),
    "vcs_group_max_f64": (ctypes.c_int, [_vp, _f64p]),
    "vcs_group_barrier": (ctypes.c_int, [_vp]),
    "vcs_gop_reserve": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int]),
    "vcs_gop_upload": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int, _u8p]),
    "vcs_gop_encode": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int]),
    "vcs_gop_decode": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int]),
    "vcs_gop_get_mv": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int, _i16p, _u32p]),
    "vcs_gop_get_planes": (ctypes.c_int, [_c_ctx, ctypes.c_int, _vp]),
    "vcs_gop_get_decoded": (ctypes.c_int, [_c_ctx, ctypes.c_int, _u8p]),
    "vcs_sync": (ctypes.c_int, [_c_ctx]),
    "vcs_dev_ptr": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.POINTER(_vp),
                                   ctypes.POINTER(ctypes.c_size_t)]),
    "vcs_ref_written": (ctypes.c_int, [_c_ctx]),
    "vcs_intra_luma4x4": (ctypes.c_int, [_c_ctx, _u8p, _f64p, _f64p, _f64p]),
    "vcs_intra_luma16x16": (ctypes.c_int, [_c_ctx, _u8p, _f64p, _f64p, _f64p]),
    "vcs_intra_chroma8x8": (ctypes.c_int, [_c_ctx, _u8p, _u8p, _f64p, _f64p, _f64p, _f64p, _f64p]),
    "vcs_intra_transform4x4": (ctypes.c_int, [_c_ctx, _f64p, _f64p]),
    "vcs_intra_batch_reserve": (ctypes.c_int, [_c_ctx, ctypes.c_int]),
    "vcs_intra_batch_upload": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int, _u8p, _u8p, _u8p]),
    "vcs_intra_batch_run": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "vcs_intra_batch_get": (ctypes.c_int, [_c_ctx, ctypes.c_int, ctypes.c_int, _f64p, _f64p, _f64p, _f64p, _f64p]),
    "vcs_chroma420": (ctypes.c_int, [_c_ctx, _u8p, _u8p, _u8p, _u8p]),
    "vcs_stream_wait": (ctypes.c_int, [_c_ctx, _vp]),
    "vcs_stream_signal": (ctypes.c_int, [_c_ctx, _vp]),
    "vcs_stream_handle": (ctypes.c_int, [_c_ctx, ctypes.POINTER(ctypes.c_void_p)]),
    "vcs_ref_swap": (ctypes.c_int, [_c_ctx]),
    "vcs_set_l3_prefetch": (ctypes.c_int, [_c_ctx, ctypes.c_int]),
    "vcs_ref_next_ready": (ctypes.c_int, [_c_ctx, ctypes.c_void_p]),
    "vcs_set_ref_next": (ctypes.c_int, [_c_ctx, ctypes.c_void_p]),
    "vcs_timing_enable": (ctypes.c_int, [_c_ctx, ctypes.c_int]),
    "vcs_timing_reset": (ctypes.c_int, [_c_ctx]),
    "vcs_timing": (ctypes.c_int, [_c_ctx, _f64p, _u64p, ctypes.c_int]),
    "vcs_me_cycles": (ctypes.c_int, [_c_ctx, _u64p, ctypes.c_int]),
    "vcs_me_clock": (ctypes.c_int, [_c_ctx, _f64p, ctypes.c_int]),
    "vcs_measure_copy_bw": (ctypes.c_int, [_c_ctx, ctypes.c_size_t, ctypes.c_int, _f64p]),
    "vcs_measure_fill_bw": (ctypes.c_int, [_c_ctx, ctypes.c_size_t, ctypes.c_int, _f64p]),
}

_LIB = None


class VcsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libvcs_hip error %d: %s" % (code, msg))
        self.code = code


def load(path=None):
    """Load libvcs_hip.so and bind every declared symbol.  Raises OSError if the
    library is missing -- there is deliberately no CPU fallback."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OSError("%s not found: build it with `python __graft_entry__.py build` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback" % p)
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _LIB = lib
    return lib


def source_build_id(extra=""):
    """What vcs_build_id() of a library built from the sources in the tree NOW would return (csrc/Makefile:
    sha256 over the extra flags line and the sources in the Makefile's order)."""
    import hashlib
    import re
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    names = re.search(r"^SRC\s*:=\s*(.*)$", mk, re.M).group(1).split() + re.search(r"^HDR\s*:=\s*(.*)$", mk, re.M).group(1).split()
    h = hashlib.sha256((extra.strip() + "\n").encode())
    for n in names:
        h.update(open(os.path.join(csrc, n), "rb").read())
    return h.hexdigest()[:16]


def build_id():
    """Identity of the loaded library's sources + flags (vcs_build_id)."""
    return load().vcs_build_id().decode()


def device_count():
    n = ctypes.c_int(0)
    rc = load().vcs_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


def pinned_empty(shape, dtype=np.uint8):
    """An uninitialised NumPy array in page-locked host memory (vcs_host_alloc): frames the
    producer writes here, and planes downloaded here, cross PCIe by DMA instead of through
    the runtime's staging copies.  Freed when the array (and every view of it) is gone."""
    import weakref
    dtype = np.dtype(dtype)
    shape = (shape,) if isinstance(shape, int) else tuple(int(x) for x in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes <= 0:
        raise ValueError("empty shape")
    lib = load()
    p = ctypes.c_void_p()
    rc = lib.vcs_host_alloc(ctypes.c_size_t(nbytes), ctypes.byref(p))
    if rc:
        raise VcsError(rc, (lib.vcs_last_error(None) or b"").decode())
    buf = (ctypes.c_uint8 * nbytes).from_address(p.value)
    weakref.finalize(buf, lib.vcs_host_free, ctypes.c_void_p(p.value))  # the array keeps `buf` alive
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def make_q(qf=50.0):
    """The Q list of DCTcompressor.py:29-38 for quality factor `qf` -> (3, 8, 8)."""
    Q = np.zeros((3, 8, 8), np.float64)
    rc = load().vcs_make_q(float(qf), Q.ctypes.data_as(_f64p))
    if rc:
        raise ValueError("Invalid quality setting, must be between 1 and 99.")
    return Q


def dct_matrix():
    D = np.zeros((8, 8), np.float64)
    load().vcs_dct_matrix(D.ctypes.data_as(_f64p))
    return D


def _frame(a, shape=None, name="frame"):
    a = np.asarray(a)
    if a.dtype != np.uint8:
        raise TypeError("%s must be uint8, got %s" % (name, a.dtype))
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return np.ascontiguousarray(a)


class ResidentPlanes:
    """The three coefficient planes of one P frame, left in HBM (vcs_store_*).

    Behaves like the reference's `[Y, Cr, Cb]` list of float64 arrays (len 3,
    indexing, iteration); the 8x-frame-size download happens on first access
    only.  The decoder consumes the HBM copy directly."""

    def __init__(self, ctx, handle):
        self._ctx, self._h, self._host = ctx, handle, None
        self._mode = ctx.coeff_mode

    def _planes(self):
        if self._host is None:
            c = self._ctx
            if self._h is None:
                raise VcsError(-4, "resident planes were released before they were read")
            out = np.zeros((3, c.H, c.W), COEFF_DTYPES[self._mode])
            c._ck(c._lib.vcs_store_planes(c._h, self._h, out.ctypes.data_as(_vp)))
            self._host = out
        return self._host

    def __len__(self):
        return 3

    def __getitem__(self, i):
        return [self._planes()[0], self._planes()[1], self._planes()[2]][i]

    def __iter__(self):
        return iter([self._planes()[k] for k in range(3)])

    @property
    def resident(self):
        return self._h is not None and bool(self._ctx._h.value)

    @property
    def nbytes(self):
        c = self._ctx
        return 3 * c.H * c.W * np.dtype(COEFF_DTYPES[self._mode]).itemsize

    def spill(self):
        """Download the planes to host memory and free the HBM copy (the view keeps
        working; the decoder then takes the host path)."""
        if self._h is not None:
            self._planes()
            self.release()

    def release(self):
        if self._h is not None:
            alive = getattr(self._ctx, "_h", None) is not None and self._ctx._h.value
            # vcs_store_free accepts a NULL context: the HBM copy is freed even when the
            # owning context was closed first
            self._ctx._lib.vcs_store_free(self._ctx._h if alive else None, self._h)
        self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Context:
    """One HIP device context for frames of H x W (vcs_ctx_create)."""

    def __init__(self, H, W, me_bs, dct_bs=8, device=0):
        self._lib = load()
        self._h = _c_ctx()
        self.H, self.W, self.me_bs = int(H), int(W), int(me_bs)
        rc = self._lib.vcs_ctx_create(int(device), self.H, self.W, self.me_bs, int(dct_bs),
                                      ctypes.byref(self._h))
        if rc:
            self._h = _c_ctx()
            raise VcsError(rc, (self._lib.vcs_last_error(None) or b"").decode())
        self.N = (self.H // self.me_bs) * (self.W // self.me_bs)
        self.F = 3 * self.H * self.W
        self.shape = (self.H, self.W, 3)
        self.coeff_mode = VCS_COEFF_F64
        self.slots = 0
        self.async_slots = 0

    # -- plumbing ----------------------------------------------------------
    def _ck(self, rc):
        if rc:
            raise VcsError(rc, (self._lib.vcs_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            grp = getattr(self, "_group", None)
            grp = grp() if grp is not None else None
            if grp is not None:          # vcs_ctx_destroy takes the group with it: drop the stale handle
                grp._g = _vp()
            self._lib.vcs_ctx_destroy(self._h)
            self._h = _c_ctx()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def coeff_dtype(self):
        return COEFF_DTYPES[self.coeff_mode]

    # -- parameters --------------------------------------------------------
    def set_params(self, win=0, step=0, sim_thresh=None, Q=None, coeff_mode=None):
        """Every argument left at its default keeps the context's current value
        (sim_thresh=None -> VCS_KEEP_THRESH: pushing Q never resets the threshold)."""
        cm = self.coeff_mode if coeff_mode is None else int(coeff_mode)
        sim_thresh = KEEP_THRESH if sim_thresh is None else int(sim_thresh)
        qp = None
        if Q is not None:
            Q = np.ascontiguousarray(np.asarray(Q, np.float64).reshape(3, 8, 8))
            qp = Q.ctypes.data_as(_f64p)
        self._ck(self._lib.vcs_set_params(self._h, int(win), int(step), int(sim_thresh), qp, cm))
        self.coeff_mode = cm

    def set_search(self, search="grid", cost="wrap"):
        """Extensions (not in the reference): search "grid" (reference) / "tss" (three-step),
        cost "wrap" (reference) / "sad"; None keeps a selector."""
        sv = {None: -1, "grid": 0, "tss": 1}[search]
        cv = {None: -1, "wrap": 0, "sad": 1}[cost]
        self._ck(self._lib.vcs_set_search(self._h, sv, cv))

    # -- single-frame entry points ------------------------------------------
    def set_ref(self, ref):
        ref = _frame(ref, self.shape, "reference frame")
        self._ck(self._lib.vcs_set_ref(self._h, ref.ctypes.data_as(_u8p)))

    def me(self, cur, want_cost=False):
        cur = _frame(cur, self.shape, "input frame")
        mv = np.zeros((self.N, 2), np.int16)
        ns = ctypes.c_uint32(0)
        cost = np.zeros(self.N, np.uint64) if want_cost else None
        self._ck(self._lib.vcs_me(self._h, cur.ctypes.data_as(_u8p), mv.ctypes.data_as(_i16p),
                                  ctypes.byref(ns),
                                  cost.ctypes.data_as(_u64p) if want_cost else None))
        return (mv, ns.value, cost) if want_cost else (mv, ns.value)

    def mc(self, mv, coords=None):
        mv = np.ascontiguousarray(np.asarray(mv, np.int16).reshape(-1, 2))
        cp = None
        if coords is not None:
            coords = np.ascontiguousarray(np.asarray(coords, np.int32).reshape(-1, 2))
            if coords.shape[0] != mv.shape[0]:
                raise ValueError("motion_vectors and block_coords differ in length")
            cp = coords.ctypes.data_as(_i32p)
        pred = np.zeros(self.shape, np.uint8)
        self._ck(self._lib.vcs_mc(self._h, mv.ctypes.data_as(_i16p), cp, mv.shape[0],
                                  pred.ctypes.data_as(_u8p)))
        return pred

    def residual(self, a, b, sign=-1):
        a = _frame(a, name="a")
        b = _frame(b, a.shape, "b")
        out = np.zeros(a.shape, np.uint8)
        self._ck(self._lib.vcs_residual(self._h, a.ctypes.data_as(_u8p), b.ctypes.data_as(_u8p),
                                        int(sign), a.size, out.ctypes.data_as(_u8p)))
        return out

    def compress(self, bgr):
        bgr = _frame(bgr, self.shape, "image")
        planes = np.zeros((3, self.H, self.W), self.coeff_dtype)
        self._ck(self._lib.vcs_compress(self._h, bgr.ctypes.data_as(_u8p),
                                        planes.ctypes.data_as(_vp)))
        return planes

    def _planes_in(self, planes):
        planes = np.ascontiguousarray(np.asarray(planes, self.coeff_dtype))
        if planes.shape != (3, self.H, self.W):
            raise ValueError("planes have shape %s, expected %s" %
                             (planes.shape, (3, self.H, self.W)))
        return planes

    def decompress(self, planes):
        planes = self._planes_in(planes)
        out = np.zeros(self.shape, np.uint8)
        self._ck(self._lib.vcs_decompress(self._h, planes.ctypes.data_as(_vp),
                                          out.ctypes.data_as(_u8p)))
        return out

    def encode_p(self, cur, want_planes=True, want_res=False):
        cur = _frame(cur, self.shape, "input frame")
        mv = np.zeros((self.N, 2), np.int16)
        planes = np.zeros((3, self.H, self.W), self.coeff_dtype) if want_planes else None
        res = np.zeros(self.shape, np.uint8) if want_res else None
        ns = ctypes.c_uint32(0)
        self._ck(self._lib.vcs_encode_p(self._h, cur.ctypes.data_as(_u8p),
                                        mv.ctypes.data_as(_i16p),
                                        planes.ctypes.data_as(_vp) if want_planes else None,
                                        res.ctypes.data_as(_u8p) if want_res else None,
                                        ctypes.byref(ns)))
        return mv, planes, res, ns.value

    def decode_p(self, mv, planes=None):
        mv = np.ascontiguousarray(np.asarray(mv, np.int16).reshape(-1, 2))
        if mv.shape[0] != self.N:
            raise ValueError("expected %d motion vectors, got %d" % (self.N, mv.shape[0]))
        pp = None
        if planes is not None:
            planes = self._planes_in(planes)
            pp = planes.ctypes.data_as(_vp)
        out = np.zeros(self.shape, np.uint8)
        self._ck(self._lib.vcs_decode_p(self._h, mv.ctypes.data_as(_i16p), pp,
                                        out.ctypes.data_as(_u8p)))
        return out

    # -- HBM-resident planes (next row N3) ------------------------------------
    def store_encode_p(self, cur):
        cur = _frame(cur, self.shape, "input frame")
        mv = np.zeros((self.N, 2), np.int16)
        ns = ctypes.c_uint32(0)
        h = _vp()
        self._ck(self._lib.vcs_store_encode_p(self._h, cur.ctypes.data_as(_u8p),
                                              mv.ctypes.data_as(_i16p), ctypes.byref(ns),
                                              ctypes.byref(h)))
        return mv, ResidentPlanes(self, h), ns.value

    def store_decode_p(self, mv, planes):
        mv = np.ascontiguousarray(np.asarray(mv, np.int16).reshape(-1, 2))
        if mv.shape[0] != self.N:
            raise ValueError("expected %d motion vectors, got %d" % (self.N, mv.shape[0]))
        out = np.zeros(self.shape, np.uint8)
        self._ck(self._lib.vcs_store_decode_p(self._h, mv.ctypes.data_as(_i16p), planes._h,
                                              out.ctypes.data_as(_u8p)))
        return out

    # -- pipelined feeder (next row N3) ----------------------------------------
    def async_reserve(self, n_slots):
        self._ck(self._lib.vcs_async_reserve(self._h, int(n_slots)))
        self.async_slots = int(n_slots)

    def encode_p_async(self, slot, cur):
        """Enqueue one P frame in `slot`; returns its (not yet filled) ResidentPlanes.  `cur`
        must stay untouched until wait(slot) when it lives in page-locked memory."""
        cur = _frame(cur, self.shape, "input frame")
        h = _vp()
        self._ck(self._lib.vcs_encode_p_async(self._h, int(slot), cur.ctypes.data_as(_u8p),
                                              ctypes.byref(h)))
        # the buffer the upload reads (a fresh contiguous copy when the caller passed a strided view)
        # lives until wait(slot): the copy engine may still be reading it when this call returns
        if not hasattr(self, "_async_inputs"):
            self._async_inputs = {}
        self._async_inputs[int(slot)] = cur
        return ResidentPlanes(self, h)

    def wait(self, slot):
        mv = np.empty((self.N, 2), np.int16)
        ns = ctypes.c_uint32(0)
        try:
            self._ck(self._lib.vcs_wait(self._h, int(slot), mv.ctypes.data_as(_i16p), ctypes.byref(ns)))
        finally:
            getattr(self, "_async_inputs", {}).pop(int(slot), None)
        return mv, ns.value

    # -- intra prediction (next row N1) ---------------------------------------
    def _plane(self, a, name):
        a = np.asarray(a)
        if a.dtype != np.uint8 or a.shape != (self.H, self.W):
            raise ValueError("%s must be a uint8 plane of shape (%d, %d)" % (name, self.H, self.W))
        return np.ascontiguousarray(a)

    def intra_luma(self, Y, bs):
        Y = self._plane(Y, "Y")
        res, pred = np.zeros((self.H, self.W)), np.zeros((self.H, self.W))
        modes = np.zeros((self.H // bs, self.W // bs))
        fn = self._lib.vcs_intra_luma4x4 if bs == 4 else self._lib.vcs_intra_luma16x16
        self._ck(fn(self._h, Y.ctypes.data_as(_u8p), res.ctypes.data_as(_f64p),
                    pred.ctypes.data_as(_f64p), modes.ctypes.data_as(_f64p)))
        return res, pred, modes

    def intra_chroma(self, Cr, Cb):
        Cr, Cb = self._plane(Cr, "Cr"), self._plane(Cb, "Cb")
        outs = [np.zeros((self.H, self.W)) for _ in range(4)]
        modes = np.zeros((self.H // 8, self.W // 8))
        self._ck(self._lib.vcs_intra_chroma8x8(self._h, Cr.ctypes.data_as(_u8p),
                                               Cb.ctypes.data_as(_u8p),
                                               *[o.ctypes.data_as(_f64p) for o in outs],
                                               modes.ctypes.data_as(_f64p)))
        return outs[0], outs[1], outs[2], outs[3], modes

    # -- the same over a batch of I frames resident in HBM (vcs_intra_batch_*) --
    INTRA_LUMA4, INTRA_LUMA16, INTRA_CHROMA8, INTRA_TRANSFORM4 = 1, 2, 4, 8

    def intra_batch_reserve(self, n_frames):
        self._ck(self._lib.vcs_intra_batch_reserve(self._h, int(n_frames)))

    def intra_batch_upload(self, first, Y, Cr, Cb):
        """Y, Cr, Cb: uint8 arrays of shape (n, H, W)."""
        arrs = [np.ascontiguousarray(a, np.uint8) for a in (Y, Cr, Cb)]
        n = arrs[0].shape[0]
        for a in arrs:
            if a.shape != (n, self.H, self.W):
                raise ValueError("planes must be uint8 of shape (n, %d, %d)" % (self.H, self.W))
        self._ck(self._lib.vcs_intra_batch_upload(self._h, int(first), n, *[a.ctypes.data_as(_u8p) for a in arrs]))

    def intra_batch_run(self, first, n, what=15):
        self._ck(self._lib.vcs_intra_batch_run(self._h, int(first), int(n), int(what)))

    def intra_batch_get(self, frame, which):
        """which 0 / 1: (res, pred, modes) of luma 4x4 / 16x16; 2: (crres, crpred, cbres, cbpred, modes) of chroma
        8x8; 3: the 4x4 transform of the luma 4x4 residual."""
        H, W = self.H, self.W
        bs = {0: 4, 1: 16, 2: 8}.get(which)
        planes = [np.zeros((H, W), np.float64) for _ in range({0: 2, 1: 2, 2: 4, 3: 1}[which])]
        modes = np.zeros((H // bs, W // bs), np.float64) if bs else None
        ptrs = [p.ctypes.data_as(_f64p) for p in planes] + [None] * (4 - len(planes))
        self._ck(self._lib.vcs_intra_batch_get(self._h, int(frame), int(which), *ptrs,
                                               modes.ctypes.data_as(_f64p) if bs else None))
        return planes[0] if which == 3 else tuple(planes) + (modes,)

    def intra_transform4x4(self, res):
        """np.round(C . X . C^T) per 4x4 block of a float64 (H, W) plane (intraframe.py:352-358)."""
        res = np.ascontiguousarray(res, dtype=np.float64)
        if res.shape != (self.H, self.W):
            raise ValueError("plane must have shape (%d, %d)" % (self.H, self.W))
        out = np.zeros_like(res)
        self._ck(self._lib.vcs_intra_transform4x4(self._h, res.ctypes.data_as(_f64p),
                                                  out.ctypes.data_as(_f64p)))
        return out

    def chroma420(self, bgr, want_y=True):
        """4:2:0 subsampling of a BGR frame (chroma.py:8-21): (Y or None, Cr samples, Cb samples)."""
        bgr = _frame(bgr, self.shape, "image")
        hc, wc = (self.H + 1) // 2, (self.W + 1) // 2
        Y = np.zeros((self.H, self.W), np.uint8) if want_y else None
        cr, cb = np.zeros((hc, wc), np.uint8), np.zeros((hc, wc), np.uint8)
        self._ck(self._lib.vcs_chroma420(self._h, bgr.ctypes.data_as(_u8p),
                                         Y.ctypes.data_as(_u8p) if want_y else None,
                                         cr.ctypes.data_as(_u8p), cb.ctypes.data_as(_u8p)))
        return Y, cr, cb

    # -- GOP path -----------------------------------------------------------
    def gop_reserve(self, n_slots, with_decode=False):
        self._ck(self._lib.vcs_gop_reserve(self._h, int(n_slots), int(bool(with_decode))))
        self.slots = int(n_slots)

    def gop_upload(self, first, frames):
        frames = np.asarray(frames)
        if frames.dtype != np.uint8 or frames.shape[1:] != self.shape:
            raise ValueError("frames must be uint8 of shape (n, %d, %d, 3)" % (self.H, self.W))
        frames = np.ascontiguousarray(frames)
        self._ck(self._lib.vcs_gop_upload(self._h, int(first), frames.shape[0],
                                          frames.ctypes.data_as(_u8p)))

    def gop_encode(self, first, n):
        self._ck(self._lib.vcs_gop_encode(self._h, int(first), int(n)))

    def gop_decode(self, first, n):
        self._ck(self._lib.vcs_gop_decode(self._h, int(first), int(n)))

    def gop_get_mv(self, first, n):
        mv = np.zeros((n, self.N, 2), np.int16)
        ns = np.zeros(n, np.uint32)
        self._ck(self._lib.vcs_gop_get_mv(self._h, int(first), int(n), mv.ctypes.data_as(_i16p),
                                          ns.ctypes.data_as(_u32p)))
        return mv, ns

    def gop_get_planes(self, slot):
        planes = np.zeros((3, self.H, self.W), self.coeff_dtype)
        self._ck(self._lib.vcs_gop_get_planes(self._h, int(slot), planes.ctypes.data_as(_vp)))
        return planes

    def gop_get_decoded(self, slot):
        out = np.zeros(self.shape, np.uint8)
        self._ck(self._lib.vcs_gop_get_decoded(self._h, int(slot), out.ctypes.data_as(_u8p)))
        return out

    def sync(self):
        self._ck(self._lib.vcs_sync(self._h))

    def dev_ptr(self, which):
        p = _vp()
        b = ctypes.c_size_t(0)
        self._ck(self._lib.vcs_dev_ptr(self._h, int(which), ctypes.byref(p), ctypes.byref(b)))
        return p.value, b.value

    def ref_written(self):
        self._ck(self._lib.vcs_ref_written(self._h))

    def ref_swap(self):
        """Exchange the current and the alternate reference buffer (dev_ptr 0 <-> 3)."""
        self._ck(self._lib.vcs_ref_swap(self._h))

    def set_l3_prefetch(self, mode):
        """-1 default, 0 off, 1 inside the search launch, 2 separate pass (vcs_set_l3_prefetch)."""
        self._ck(self._lib.vcs_set_l3_prefetch(self._h, int(mode)))

    def ref_next_ready(self, written_on_stream=0):
        """The alternate reference buffer (dev_ptr 3) holds the next GOP's I frame: its search tables are cut
        beside the search kernel of the GOP encoded next; `ref_swap` then finds them ready."""
        self._ck(self._lib.vcs_ref_next_ready(self._h, ctypes.c_void_p(int(written_on_stream) or None)))

    def set_ref_next(self, ref):
        """Upload the NEXT GOP's I frame into the alternate buffer and declare it ready (see ref_next_ready)."""
        ref = _frame(ref, self.shape, "reference frame")
        self._ck(self._lib.vcs_set_ref_next(self._h, ref.ctypes.data_as(_u8p)))

    def stream_wait(self, foreign_stream):
        """Our stream waits for `foreign_stream` (a hipStream_t as int; 0 = default)."""
        self._ck(self._lib.vcs_stream_wait(self._h, ctypes.c_void_p(int(foreign_stream) or None)))

    def stream_signal(self, foreign_stream):
        """`foreign_stream` waits for our stream."""
        self._ck(self._lib.vcs_stream_signal(self._h, ctypes.c_void_p(int(foreign_stream) or None)))

    def stream_handle(self):
        """The context's hipStream_t as an int (e.g. for torch.cuda.ExternalStream)."""
        p = ctypes.c_void_p()
        self._ck(self._lib.vcs_stream_handle(self._h, ctypes.byref(p)))
        return int(p.value or 0)

    # -- measurement --------------------------------------------------------
    def timing_enable(self, on=True):
        self._ck(self._lib.vcs_timing_enable(self._h, int(bool(on))))

    def timing_reset(self):
        self._ck(self._lib.vcs_timing_reset(self._h))

    def timing(self):
        ms = np.zeros(T_COUNT, np.float64)
        n = np.zeros(T_COUNT, np.uint64)
        self._ck(self._lib.vcs_timing(self._h, ms.ctypes.data_as(_f64p), n.ctypes.data_as(_u64p),
                                      T_COUNT))
        return ms, n

    def me_cycles(self, reset=True):
        v = ctypes.c_uint64(0)
        self._ck(self._lib.vcs_me_cycles(self._h, ctypes.byref(v), int(bool(reset))))
        return v.value

    def me_clock(self, reset=True):
        """GHz the chip held while k_me16b ran (timing on), 0.0 if it has not run."""
        v = ctypes.c_double(0)
        self._ck(self._lib.vcs_me_clock(self._h, ctypes.byref(v), int(bool(reset))))
        return v.value

    def measure_copy_bw(self, nbytes=1 << 30, iters=10):
        g = ctypes.c_double(0)
        self._ck(self._lib.vcs_measure_copy_bw(self._h, int(nbytes), int(iters), ctypes.byref(g)))
        return g.value

    def measure_fill_bw(self, nbytes=1 << 30, iters=10):
        g = ctypes.c_double(0)
        self._ck(self._lib.vcs_measure_fill_bw(self._h, int(nbytes), int(iters), ctypes.byref(g)))
        return g.value


# ---------------------------------------------------------------------------
# Multi-GPU GOP encode: RCCL inside libvcs_hip.so (vcs_group_*, vcs_encode_gop)
# ---------------------------------------------------------------------------
def shard_frames(n_frames, world, rank):
    """Contiguous chunk [first, first + count) of `n_frames` P frames for shard `rank`
    (vcs_shard_frames: sizes differ by at most one, earlier shards take the larger)."""
    first, count = ctypes.c_int(0), ctypes.c_int(0)
    rc = load().vcs_shard_frames(int(n_frames), int(world), int(rank), ctypes.byref(first),
                                 ctypes.byref(count))
    if rc:
        raise ValueError("bad shard request")
    return first.value, count.value


def comm_unique_id():
    """128 opaque bytes from ncclGetUniqueId: rank 0 makes them, every rank of a
    process-per-GPU job passes the same bytes to Group.from_rank."""
    buf = ctypes.create_string_buffer(128)
    lib = load()
    rc = lib.vcs_comm_unique_id(buf, 128)
    if rc:
        raise VcsError(rc, (lib.vcs_group_last_error(None) or b"").decode())
    return buf.raw


TRANSPORTS = {1: "RCCL, overlapped with the neighbouring GOPs' kernels", 2: "RCCL, in sequence with the kernels",
              3: "same-device copies, overlapped", 4: "same-device copies, in sequence"}


class Group:
    """The P frames of a GOP sharded over several contexts (one per GPU); see
    include/vcs_hip.h.  Group(contexts): single process.  Group.from_rank(ctx, id, world,
    rank): one process per GPU."""

    def __init__(self, contexts, _handle=None, _world=None):
        self._lib = load()
        self.contexts = list(contexts)
        self._g = _vp()
        if _handle is not None:
            self._g = _handle
        else:
            arr = (_c_ctx * len(self.contexts))(*[c._h for c in self.contexts])
            rc = self._lib.vcs_group_create(arr, len(self.contexts), ctypes.byref(self._g))
            if rc:
                self._g = _vp()
                raise VcsError(rc, (self._lib.vcs_group_last_error(None) or b"").decode())
        w, r0, nl, tr = (ctypes.c_int(0) for _ in range(4))
        self._lib.vcs_group_info(self._g, ctypes.byref(w), ctypes.byref(r0), ctypes.byref(nl),
                                 ctypes.byref(tr))
        self.world, self.rank0, self.n_local, self.transport = w.value, r0.value, nl.value, tr.value
        self.N = self.contexts[0].N
        self.n_p = None
        import weakref
        for c in self.contexts:
            c._group = weakref.ref(self)

    @classmethod
    def from_rank(cls, ctx, unique_id, world, rank):
        lib = load()
        g = _vp()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        rc = lib.vcs_group_create_rank(ctx._h, buf, int(world), int(rank), ctypes.byref(g))
        if rc:
            raise VcsError(rc, (lib.vcs_group_last_error(None) or b"").decode())
        return cls([ctx], _handle=g)

    def _ck(self, rc):
        if rc:
            raise VcsError(rc, (self._lib.vcs_group_last_error(self._g if self._g.value else None) or b"").decode())

    def close(self):
        if getattr(self, "_g", None) is not None and self._g.value:
            self._lib.vcs_group_destroy(self._g)   # (Context.close clears _g when it takes the group down)
            self._g = _vp()
        for c in self.contexts:
            c._group = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def layout(self, n_p_frames, with_decode=False, sets=1):
        self._ck(self._lib.vcs_group_layout_sets(self._g, int(n_p_frames), int(bool(with_decode)),
                                                 int(sets)))
        self.n_p = int(n_p_frames)
        self.sets, self.set = int(sets), 0
        for c in self.contexts:
            c.slots = max(c.slots, 1)

    def use_set(self, k):
        self._ck(self._lib.vcs_group_use_set(self._g, int(k)))
        self.set = int(k)

    def slot_of(self, local_idx, k):
        """Context slot of P frame `k` (GOP index) on local shard `local_idx`, set in use."""
        first, count = self.shard(local_idx)
        if not first <= k < first + count:
            raise IndexError("P frame %d is not on local shard %d" % (k, local_idx))
        return self.set * count + k - first

    def shard(self, local_idx=0):
        f, n = ctypes.c_int(0), ctypes.c_int(0)
        self._ck(self._lib.vcs_group_shard(self._g, int(local_idx), ctypes.byref(f), ctypes.byref(n)))
        return f.value, n.value

    def upload(self, first_p, frames):
        frames = np.asarray(frames)
        shape = self.contexts[0].shape
        if frames.dtype != np.uint8 or frames.shape[1:] != shape:
            raise ValueError("frames must be uint8 of shape (n, %d, %d, 3)" % shape[:2])
        frames = np.ascontiguousarray(frames)
        self._ck(self._lib.vcs_group_upload(self._g, int(first_p), frames.shape[0],
                                            frames.ctypes.data_as(_u8p)))

    def set_ref(self, ref):
        """Stage the next GOP's I frame.  EVERY process of a process-per-GPU job calls this (it
        keeps their sequences of collectives in step); only the one holding shard 0 needs `ref`."""
        if self.rank0 != 0:
            self._ck(self._lib.vcs_group_set_ref(self._g, None))
            return
        ref = _frame(ref, self.contexts[0].shape, "reference frame")
        self._ck(self._lib.vcs_group_set_ref(self._g, ref.ctypes.data_as(_u8p)))

    def prefetch_ref(self):
        self._ck(self._lib.vcs_group_prefetch_ref(self._g))

    def encode(self):
        self._ck(self._lib.vcs_group_encode(self._g))

    def decode(self):
        self._ck(self._lib.vcs_group_decode(self._g))

    def sync(self):
        self._ck(self._lib.vcs_group_sync(self._g))

    def get_mv(self):
        mv = np.zeros((self.n_p, self.N, 2), np.int16)
        ns = np.zeros(self.n_p, np.uint32)
        self._ck(self._lib.vcs_group_get_mv(self._g, mv.ctypes.data_as(_i16p), ns.ctypes.data_as(_u32p)))
        return mv, ns

    def max_over_ranks(self, v):
        x = ctypes.c_double(float(v))
        self._ck(self._lib.vcs_group_max_f64(self._g, ctypes.byref(x)))
        return x.value

    def barrier(self):
        self._ck(self._lib.vcs_group_barrier(self._g))


def encode_gop(contexts, frames, want_planes=False):
    """vcs_encode_gop: frames[0] is the I frame, frames[1:] its P frames.  Returns
    (mv (n-1, N, 2) int16, planes (n-1, 3, H, W) or None)."""
    lib = load()
    c0 = contexts[0]
    frames = np.asarray(frames)
    if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[1:] != c0.shape or len(frames) < 1:
        raise ValueError("frames must be uint8 of shape (n >= 1, %d, %d, 3)" % c0.shape[:2])
    frames = np.ascontiguousarray(frames)
    n = frames.shape[0]
    mv = np.zeros((n - 1, c0.N, 2), np.int16)
    planes = np.zeros((n - 1, 3, c0.H, c0.W), c0.coeff_dtype) if want_planes else None
    arr = (_c_ctx * len(contexts))(*[c._h for c in contexts])
    rc = lib.vcs_encode_gop(arr, len(contexts), frames.ctypes.data_as(_u8p), n,
                            mv.ctypes.data_as(_i16p),
                            planes.ctypes.data_as(_vp) if want_planes else None)
    if rc:
        raise VcsError(rc, (lib.vcs_group_last_error(None) or b"").decode())
    return mv, planes


def _job_tag():
    """Names one launch of a process-per-GPU job on this node: the launcher's port and run
    id, plus -- under torch.distributed.run, whose workers are all children of one agent
    process -- that agent's pid, so that a file left behind by an earlier (crashed) launch
    on the same port is never mistaken for this one's.  VCS_RENDEZVOUS_TAG overrides."""
    t = os.environ.get("VCS_RENDEZVOUS_TAG")
    if t:
        return t
    agent = os.getppid() if "TORCHELASTIC_RUN_ID" in os.environ else 0
    return "%s_%s_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                            os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), agent)


def rendezvous_unique_id(rank, world, timeout_s=180.0, tag=None, make_id=None):
    """Hands rank 0's RCCL unique id to the other processes of a one-node, process-per-GPU
    job (torch.distributed.run sets RANK / WORLD_SIZE / MASTER_PORT; nothing of PyTorch is
    imported).  Rank 0 writes the 128 bytes to a file named after the job in /dev/shm (or
    $TMPDIR), atomically; the others poll for it.  Files older than the timeout are stale."""
    import tempfile
    import time
    make_id = make_id or comm_unique_id
    if world == 1:
        return make_id()      # nobody to hand it to
    tag = tag or _job_tag()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = os.path.join(base, "vcs_hip_rccl_id_%d_%s_w%d" % (os.getuid(), tag, world))
    if rank == 0:
        try:
            os.unlink(path)            # whatever an earlier (crashed) launch with this tag left behind
        except OSError:
            pass
        uid = make_id()
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(b"VCSID1" + uid)
        os.replace(tmp, path)

        # The other ranks only take a file touched after their own start (below); a rank the launcher
        # starts late must still find this one fresh, so it is touched for a while from a thread
        # (rank 0 itself is about to block in ncclCommInitRank).  A file nobody touches any more --
        # one left by a crashed launch -- is never accepted.
        def keep_fresh():
            t_end = time.time() + timeout_s   # (until the slowest rank's own time-out: a rank that arrives later has timed out anyway)
            while time.time() < t_end:
                try:
                    os.utime(path, None)
                except OSError:
                    return             # rendezvous_cleanup removed it: everybody has joined
                time.sleep(0.1)
        import threading
        threading.Thread(target=keep_fresh, daemon=True).start()
        return uid
    t0 = time.time()
    while time.time() - t0 < timeout_s:
        try:
            st = os.stat(path)
            if st.st_mtime >= _T_START - 0.5 and time.time() - st.st_mtime < 5.0:
                blob = open(path, "rb").read()
                if len(blob) == 6 + 128 and blob[:6] == b"VCSID1":
                    return blob[6:]
        except OSError:
            pass
        time.sleep(0.05)
    raise TimeoutError("rank %d: no RCCL unique id from rank 0 at %s" % (rank, path))


def rendezvous_cleanup(rank, world, tag=None):
    """Rank 0 removes the id file once every rank has joined (call after a barrier)."""
    import tempfile
    if world == 1 or rank != 0:
        return
    tag = tag or _job_tag()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    try:
        os.unlink(os.path.join(base, "vcs_hip_rccl_id_%d_%s_w%d" % (os.getuid(), tag, world)))
    except OSError:
        pass
