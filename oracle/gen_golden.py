#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF.

TEST INFRASTRUCTURE.  Runs only in the build container, where the reference is
mounted at /root/reference; the GPU box never sees the reference, only the
.npz files this script wrote (data: inputs + expected outputs, no reference
source in any form).

How: puts /root/reference/InterframeCompression on sys.path and imports the
reference's own, unmodified motion.py / DCTcompressor.py / encoder.py /
decoder.py.  Their `import cv2` is satisfied by oracle/cv2_standin/cv2.py
(opencv-python is not installed and there is no network) -- see that file for
exactly which five functions it provides and which of them is a third-party
restatement (cvtColor -> the colour conversion is "parity unpinned"; every
other number in the fixtures is produced by reference code).

Usage:  python oracle/gen_golden.py            (about a minute; 1080p ME is slow)
"""
import contextlib
import hashlib
import io
import os
import sys

os.environ.setdefault("TQDM_DISABLE", "1")
os.environ.setdefault("MPLBACKEND", "Agg")

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "cv2_standin"))
sys.path.insert(0, os.path.join(REF, "InterframeCompression"))
sys.path.insert(0, os.path.join(REPO, "vcs-h264_amd"))

import numpy as np  # noqa: E402

import motion as ref_motion  # noqa: E402  (the reference's)
import DCTcompressor as ref_dct  # noqa: E402
import encoder as ref_encoder  # noqa: E402
import decoder as ref_decoder  # noqa: E402
from synth import synth_v1, synth_gop  # noqa: E402  (ours: input generator only)

assert ref_motion.__file__.startswith(REF), ref_motion.__file__
assert ref_dct.__file__.startswith(REF), ref_dct.__file__

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_me_all(cur, ref, bs):
    """mv, coords, mc image, residual -- all by reference code."""
    H, W = cur.shape[:2]
    mp = ref_motion.MotionProcessor(bs, [H, W])
    mvs, coords = mp.process_motion_prediction(cur, ref)
    mc = quiet(mp.reconstruct_from_motion_vectors, mvs, ref, coords)
    res = mp.get_residuals(input_frame=cur, reconstructed=mc)
    mv = np.array(mvs, np.int32).reshape(-1, 2)
    co = np.array(coords, np.int32).reshape(-1, 2)
    return mv, co, mc, res


def q_for(qf):
    """DCTcompressor.py:29-38 evaluated for another QF (the module computes it
    once at import for QF=50; the class has no QF parameter).  QY/QC are the
    reference's own arrays."""
    if qf == 50.0:
        return [np.array(q) for q in ref_dct.Q]
    scale = 50 / qf if (qf < 50 and qf > 1) else (100 - qf) / 50
    return [np.clip(np.round(ref_dct.QY * scale), 1, 255),
            np.clip(np.round(ref_dct.QC * scale), 1, 255),
            np.clip(np.round(ref_dct.QC * scale), 1, 255)]


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print("wrote %-22s %7.1f KB" % (name, os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------
def g1_me_small():
    ref, cur = synth_v1(144, 176, seed=1234)
    d = dict(ref=ref, cur=cur)
    for bs in (4, 8, 16):
        mv, co, mc, res = ref_me_all(cur, ref, bs)
        d["mv_%d" % bs] = mv.astype(np.int16)
        d["coords_%d" % bs] = co
        d["mc_%d" % bs] = mc
        d["res_%d" % bs] = res
    save("me_small.npz", **d)


def g2_me_edges():
    rng = np.random.default_rng(42)
    cases = []

    def add(name, ref, cur, bs):
        cases.append((name, np.ascontiguousarray(ref), np.ascontiguousarray(cur), bs))

    r = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    add("all_static", r, r.copy(), 8)
    # static threshold: sum(sat(ref-cur)) == 2000 -> static, 2001 -> searched
    for tot, nm in ((2000, "thr_eq"), (2001, "thr_gt")):
        rr = rng.integers(100, 256, (64, 64, 3), dtype=np.uint8)
        cc = rr.copy()
        blk = cc[16:32, 16:32].reshape(-1)  # copy (non-contiguous slice)
        blk = blk.copy()
        blk[:20] -= 100
        if tot == 2001:
            blk[20] -= 1
        cc[16:32, 16:32] = blk.reshape(16, 16, 3)
        add(nm, rr, cc, 16)
    # one-sided static test: the block only got brighter -> score 0 -> static
    rr = rng.integers(0, 100, (48, 48, 3), dtype=np.uint8)
    cc = rr.copy()
    cc[16:32, 16:32] += 150
    add("brighter_is_static", rr, cc, 16)
    # flat reference: every candidate ties, first in raster order wins
    add("ties_flat", np.full((64, 80, 3), 200, np.uint8),
        np.full((64, 80, 3), 10, np.uint8), 16)
    add("ties_flat8", np.full((40, 48, 3), 200, np.uint8),
        np.full((40, 48, 3), 10, np.uint8), 8)
    # wrap extremes
    add("wrap_ref0_cur255", np.zeros((48, 48, 3), np.uint8),
        np.full((48, 48, 3), 255, np.uint8), 8)
    add("wrap_ref255_cur0", np.full((48, 48, 3), 255, np.uint8),
        np.zeros((48, 48, 3), np.uint8), 8)
    # frames so small that no candidate passes the bounds tests -> best stays [0,0]
    a = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
    add("no_candidates_1blk", a, 255 - a, 16)
    a = rng.integers(0, 256, (16, 48, 3), dtype=np.uint8)
    add("no_candidates_row", a, np.roll(a, 7, axis=1), 16)
    # sizes that are not multiples of the block: partial blocks are ignored
    a = rng.integers(0, 256, (50, 70, 3), dtype=np.uint8)
    add("ragged_8", a, np.roll(a, (2, 3), axis=(0, 1)), 8)
    a = rng.integers(0, 256, (70, 90, 3), dtype=np.uint8)
    add("ragged_16", a, np.roll(a, (-4, 6), axis=(0, 1)), 16)
    a = rng.integers(0, 256, (30, 26, 3), dtype=np.uint8)
    add("ragged_4", a, np.roll(a, (1, -2), axis=(0, 1)), 4)
    # odd block sizes exercise STEP_SIZE = round(bs/3)
    a = rng.integers(0, 256, (60, 72, 3), dtype=np.uint8)
    add("bs12", a, np.roll(a, (3, 4), axis=(0, 1)), 12)
    add("bs6", a, np.roll(a, (-2, 2), axis=(0, 1)), 6)
    # random content with a mixture of static and moving blocks
    a = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    c = a.copy()
    c[16:48, 32:64] = np.roll(a, (5, -7), axis=(0, 1))[16:48, 32:64]
    add("mixed_16", a, c, 16)
    add("mixed_8", a, c, 8)

    d = {"names": np.array([c[0] for c in cases])}
    for k, (name, ref, cur, bs) in enumerate(cases):
        mv, co, mc, res = ref_me_all(cur, ref, bs)
        d["c%d_ref" % k] = ref
        d["c%d_cur" % k] = cur
        d["c%d_bs" % k] = np.int32(bs)
        d["c%d_mv" % k] = mv.astype(np.int32)
        d["c%d_mc" % k] = mc
        d["c%d_res" % k] = res
    save("me_edges.npz", **d)


def g3_me_1080p():
    ref, cur = synth_v1(1080, 1920, seed=1234)
    d = dict(ref_sha=np.array(sha(ref)), cur_sha=np.array(sha(cur)))
    for bs in (16, 8):
        mv, co, mc, res = ref_me_all(cur, ref, bs)
        d["mv_%d" % bs] = mv.astype(np.int16)
        d["mc_sha_%d" % bs] = np.array(sha(mc))
        d["res_sha_%d" % bs] = np.array(sha(res))
        print("  1080p bs=%d done (%d blocks, %d zero MVs)" %
              (bs, len(mv), int((mv == 0).all(axis=1).sum())))
    save("me_1080p.npz", **d)


def g4_dct_small():
    rng = np.random.default_rng(7)
    H, W = 64, 96
    yy, xx = np.mgrid[0:H, 0:W]
    smooth = (40 * np.sin(yy / 9.0)[..., None] + 30 * np.cos(xx / 7.0)[..., None] +
              np.array([10, 60, 120])).astype(np.int64)
    img = ((smooth + rng.integers(-20, 21, (H, W, 3))) % 256).astype(np.uint8)
    img[:16, :16] = rng.integers(0, 256, (16, 16, 3))  # full-range corner
    img[16:24, :8] = 0
    img[16:24, 8:16] = 255
    d = dict(img=img)
    import cv2 as standin
    d["ycc_standin"] = standin.cvtColor(img, standin.COLOR_BGR2YCR_CB)
    for qf in (10.0, 50.0, 99.0):
        c = ref_dct.DCTCompressor(8)
        c.Q = q_for(qf)
        planes = quiet(c.compress, img)
        dec = quiet(c.decompress, compressed=planes, imshape=img.shape)
        tag = "q%d" % int(qf)
        d["Q_" + tag] = np.stack(c.Q).astype(np.float64)
        d["planes_" + tag] = np.stack(planes)
        d["dec_" + tag] = dec
    d["dctmat"] = ref_dct.DCTCompressor(8)._dctMatrix()
    save("dct_small.npz", **d)


def g5_codec_small():
    H, W = 64, 96
    ref, frames = synth_gop(H, W, 3, seed=99)
    seq = [ref, frames[0], frames[1], ref[::-1].copy(), frames[2]]  # I P P | I P
    pattern = ["I", "P", "P"]
    d = dict(frames=np.stack(seq))
    for with_dct in (True, False):
        enc = ref_encoder.Encoder(pattern, [H, W], 8, with_dct)
        for n, f in enumerate(seq):
            quiet(enc.encode_frame, f, n)
        dec = ref_decoder.Decoder(encoded_frames=enc.encoded_frames, fps=25.0, shape=[H, W],
                                  ref_frames=enc.ref_frames, block_size=8, with_DCT=with_dct)
        tag = "dct" if with_dct else "raw"
        for n, fr in enumerate(enc.encoded_frames):
            d["%s_type_%d" % (tag, n)] = np.array(fr.t)
            d["%s_refi_%d" % (tag, n)] = np.int32(fr.ref_i)
            if fr.t != "P":
                continue
            d["%s_mv_%d" % (tag, n)] = np.array(fr.mv, np.int32)
            d["%s_r_%d" % (tag, n)] = np.stack(fr.r) if with_dct else fr.r
            d["%s_full_%d" % (tag, n)] = quiet(dec._reconstruct_P_frame, fr, True)
            d["%s_mconly_%d" % (tag, n)] = quiet(dec._reconstruct_P_frame, fr, False)
    save("codec_small.npz", **d)


def g6_real_small():
    from PIL import Image
    d = {}
    p = os.path.join(REF, "images", "sequences", "minor-jump")
    a = np.array(Image.open(os.path.join(p, "0.png")).convert("RGB"))[:160, :160, ::-1]
    b = np.array(Image.open(os.path.join(p, "1.png")).convert("RGB"))[:160, :160, ::-1]
    pairs = {"jump": (np.ascontiguousarray(a), np.ascontiguousarray(b))}
    p = os.path.join(REF, "images", "corgi-underwater")
    names = sorted(os.listdir(p))
    jp = [n for n in names if n.lower().endswith(".jpg")][:2]
    a = np.array(Image.open(os.path.join(p, jp[0])).convert("RGB"))[:256, :256, ::-1]
    b = np.array(Image.open(os.path.join(p, jp[1])).convert("RGB"))[:256, :256, ::-1]
    pairs["corgi"] = (np.ascontiguousarray(a), np.ascontiguousarray(b))
    for nm, (ref, cur) in pairs.items():
        d[nm + "_ref"] = ref
        d[nm + "_cur"] = cur
        for bs in (8, 16):
            mv, co, mc, res = ref_me_all(cur, ref, bs)
            d["%s_mv_%d" % (nm, bs)] = mv.astype(np.int32)
            d["%s_res_sha_%d" % (nm, bs)] = np.array(sha(res))
            if bs == 16:
                c = ref_dct.DCTCompressor(8)
                planes = quiet(c.compress, res)
                dec = quiet(c.decompress, compressed=planes, imshape=res.shape)
                d[nm + "_planes16_sha"] = np.array(sha(np.stack(planes)))
                d[nm + "_planes16_sum"] = np.float64(np.abs(np.stack(planes)).sum())
                d[nm + "_full16"] = mc + dec  # decoder.py:57
    save("real_small.npz", **d)


def g7_intra_small():
    """Intra prediction.  intraframe.py runs its demo at import
    (line 391), so only its definitions -- source lines 1-318, the three
    functions luma4x4 / luma16x16 / chroma8x8 -- are executed, in place, from
    /root/reference; intramodes.py is imported normally."""
    import warnings
    idir = os.path.join(REF, "IntraframeCompression")
    sys.path.insert(0, idir)
    src = open(os.path.join(idir, "intraframe.py")).read().split("\n")
    ns = {"__name__": "ref_intraframe_defs"}
    exec(compile("\n".join(src[:318]), os.path.join(idir, "intraframe.py"), "exec"), ns)
    import cv2 as standin
    rng = np.random.default_rng(11)
    H, W = 64, 96
    yy, xx = np.mgrid[0:H, 0:W]
    img = (np.stack([120 + 60 * np.sin(xx / 11.0) + 30 * np.cos(yy / 5.0),
                     90 + 80 * np.cos((xx + yy) / 9.0), 140 + 70 * np.sin(yy / 7.0)], axis=-1)
           + rng.integers(-25, 26, (H, W, 3))).clip(0, 255).astype(np.uint8)
    img[16:32, 32:64] = rng.integers(0, 256, (16, 32, 3))      # full-range texture
    img[40:48, :16] = 255                                      # exercises the 3*x//4 wrap
    img[48:56, 80:] = 0
    ycc = standin.cvtColor(img, standin.COLOR_BGR2YCR_CB)
    Y, Cr, Cb = [np.ascontiguousarray(ycc[..., k]) for k in range(3)]
    d = dict(Y=Y, Cr=Cr, Cb=Cb)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # the reference's uint8 scalar overflow warnings
        r, p, m = ns["luma4x4"](Y)
        d.update(l4_res=r, l4_pred=p, l4_modes=m)
        r, p, m = ns["luma16x16"](Y)
        d.update(l16_res=r, l16_pred=p, l16_modes=m)
        rr, rp, br, bp, m = ns["chroma8x8"](Cr, Cb)
        d.update(c8_crres=rr, c8_crpred=rp, c8_cbres=br, c8_cbpred=bp, c8_modes=m)
        # a second, fully random plane set: every mode and every edge case gets hit
        Y2 = rng.integers(0, 256, (48, 64), dtype=np.uint8)
        Cr2 = rng.integers(0, 256, (48, 64), dtype=np.uint8)
        Cb2 = rng.integers(0, 256, (48, 64), dtype=np.uint8)
        d.update(Y2=Y2, Cr2=Cr2, Cb2=Cb2)
        r, p, m = ns["luma4x4"](Y2)
        d.update(l4_res2=r, l4_pred2=p, l4_modes2=m)
        r, p, m = ns["luma16x16"](Y2)
        d.update(l16_res2=r, l16_pred2=p, l16_modes2=m)
        rr, rp, br, bp, m = ns["chroma8x8"](Cr2, Cb2)
        d.update(c8_crres2=rr, c8_crpred2=rp, c8_cbres2=br, c8_cbpred2=bp, c8_modes2=m)
    save("intra_small.npz", **d)


def g8_intramodes_small():
    """The fifteen predictors of the reference's intramodes.py (:7-179), called one by one on
    recorded neighbour vectors (uint8, as intraframe.py hands them over): 40 random cases plus
    the extremes that exercise the uint8 wrap of `u + l` and `3 * x // 4`."""
    import warnings
    idir = os.path.join(REF, "IntraframeCompression")
    if idir not in sys.path:
        sys.path.insert(0, idir)
    import intramodes as ref_im
    assert ref_im.__file__.startswith(REF)
    rng = np.random.default_rng(23)
    n = 44
    u = rng.integers(0, 256, (n, 4), dtype=np.uint8)
    ur = rng.integers(0, 256, (n, 4), dtype=np.uint8)
    le = rng.integers(0, 256, (n, 4), dtype=np.uint8)
    ul = rng.integers(0, 256, n, dtype=np.uint8)
    u16 = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    l16 = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    u8 = rng.integers(0, 256, (n, 8), dtype=np.uint8)
    l8 = rng.integers(0, 256, (n, 8), dtype=np.uint8)
    for arr in (u, ur, le, u16, l16, u8, l8):
        arr[40], arr[41], arr[42], arr[43] = 255, 0, 86, 171
    ul[40:44] = (255, 0, 86, 171)
    d = dict(u=u, ur=ur, l=le, ul=ul, u16=u16, l16=l16, u8=u8, l8=l8)
    calls = {"vertical4x4": lambda k: (u[k],), "horizontal4x4": lambda k: (le[k],),
             "dc4x4": lambda k: (u[k], le[k]), "downleft4x4": lambda k: (u[k], ur[k]),
             "downright4x4": lambda k: (ul[k], u[k], le[k]),
             "verticalright4x4": lambda k: (ul[k], u[k], le[k]),
             "horizontaldown4x4": lambda k: (ul[k], u[k], le[k]),
             "verticalleft4x4": lambda k: (u[k], ur[k]), "horizontalup4x4": lambda k: (le[k],),
             "vertical16x16": lambda k: (u16[k],), "horizontal16x16": lambda k: (l16[k],),
             "dc16x16": lambda k: (u16[k], l16[k]), "vertical8x8": lambda k: (u8[k],),
             "horizontal8x8": lambda k: (l8[k],), "dc8x8": lambda k: (u8[k], l8[k])}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, args in calls.items():
            d["out_" + name] = np.stack([getattr(ref_im, name)(*args(k)) for k in range(n)])
    save("intramodes_small.npz", **d)


if __name__ == "__main__":
    only = sys.argv[1:]
    for fn in (g1_me_small, g2_me_edges, g4_dct_small, g5_codec_small, g6_real_small,
               g7_intra_small, g8_intramodes_small, g3_me_1080p):
        if only and fn.__name__.split("_")[0] not in only:
            continue
        print(fn.__name__)
        fn()
