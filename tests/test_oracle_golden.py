"""The CPU oracle (oracle/vcs_oracle.c) against the golden vectors captured by
running the reference itself (oracle/gen_golden.py).  Everything here is
bit-exact; this is what "pins" the oracle.  No GPU needed."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("bs", [4, 8, 16])
def test_me_small(oracle, bs):
    g = load_golden("me_small.npz")
    ref, cur = g["ref"], g["cur"]
    mv, ns = oracle.me(cur, ref, bs)
    assert np.array_equal(mv, g["mv_%d" % bs])
    assert np.array_equal(oracle.block_coords(*ref.shape[:2], bs), g["coords_%d" % bs])
    pred = oracle.mc(ref, mv, bs)
    assert np.array_equal(pred, g["mc_%d" % bs])
    assert np.array_equal(oracle.residual(cur, pred), g["res_%d" % bs])


def test_me_edges(oracle):
    g = load_golden("me_edges.npz")
    names = [str(n) for n in g["names"]]
    assert len(names) >= 15
    for k, name in enumerate(names):
        ref, cur, bs = g["c%d_ref" % k], g["c%d_cur" % k], int(g["c%d_bs" % k])
        mv, ns = oracle.me(cur, ref, bs)
        assert np.array_equal(mv, g["c%d_mv" % k]), name
        pred = oracle.mc(ref, mv, bs)
        assert np.array_equal(pred, g["c%d_mc" % k]), name
        assert np.array_equal(oracle.residual(cur, pred), g["c%d_res" % k]), name
    # spot checks of what the cases are meant to exercise
    i = names.index("thr_eq"); j = names.index("thr_gt")
    assert oracle.me(g["c%d_cur" % i], g["c%d_ref" % i], 16)[1] == 16
    assert oracle.me(g["c%d_cur" % j], g["c%d_ref" % j], 16)[1] == 15
    k = names.index("no_candidates_row")
    assert np.array_equal(g["c%d_mv" % k], [[0, 0], [-16, 0], [-32, 0]])


@pytest.mark.parametrize("bs", [16, 8])
def test_me_1080p(oracle, bs):
    from synth import synth_v1
    g = load_golden("me_1080p.npz")
    ref, cur = synth_v1(1080, 1920, seed=1234)
    if sha(ref) != str(g["ref_sha"]) or sha(cur) != str(g["cur_sha"]):
        pytest.skip("synthetic input drifted (numpy RNG differs from the fixture's)")
    mv, ns = oracle.me(cur, ref, bs)
    assert np.array_equal(mv, g["mv_%d" % bs])
    pred = oracle.mc(ref, mv, bs)
    assert sha(pred) == str(g["mc_sha_%d" % bs])
    assert sha(oracle.residual(cur, pred)) == str(g["res_sha_%d" % bs])


def test_dct_matrix_and_q(oracle):
    g = load_golden("dct_small.npz")
    assert np.array_equal(oracle.dct_matrix(8), g["dctmat"])
    for qf in (10, 50, 99):
        assert np.array_equal(oracle.make_q(float(qf)), g["Q_q%d" % qf])
    with pytest.raises(ValueError):
        oracle.make_q(100.0)


@pytest.mark.parametrize("qf", [10, 50, 99])
def test_dct_small_bit_exact(oracle, qf):
    """float64 coefficient planes and decoded pixels, bit for bit (the FMA-chain
    restatement of np.matmul -- see oracle/vcs_oracle.c header)."""
    g = load_golden("dct_small.npz")
    img, Q = g["img"], g["Q_q%d" % qf]
    planes, ycc = oracle.compress(img, Q, want_ycc=True)
    assert np.array_equal(ycc, g["ycc_standin"])
    assert np.array_equal(planes, g["planes_q%d" % qf])
    assert np.array_equal(oracle.decompress(g["planes_q%d" % qf], Q), g["dec_q%d" % qf])


def test_codec_small(oracle):
    g = load_golden("codec_small.npz")
    frames = g["frames"]
    refs = {0: frames[0], 1: frames[3]}
    for n in (1, 2, 4):
        ref = refs[int(g["dct_refi_%d" % n])]
        mv, planes, ns, res = oracle.encode_p(frames[n], ref, 8, want_res=True)
        assert np.array_equal(mv, g["dct_mv_%d" % n])
        assert np.array_equal(planes, g["dct_r_%d" % n])
        assert np.array_equal(res, g["raw_r_%d" % n])
        assert np.array_equal(oracle.decode_p(mv, planes, ref, 8), g["dct_full_%d" % n])
        pred = oracle.mc(ref, mv, 8)
        assert np.array_equal(pred, g["dct_mconly_%d" % n])
        assert np.array_equal(oracle.wrap_add(pred, res), g["raw_full_%d" % n])


@pytest.mark.parametrize("nm", ["jump", "corgi"])
def test_real_small(oracle, nm):
    g = load_golden("real_small.npz")
    ref, cur = g[nm + "_ref"], g[nm + "_cur"]
    for bs in (8, 16):
        mv, ns = oracle.me(cur, ref, bs)
        assert np.array_equal(mv, g["%s_mv_%d" % (nm, bs)])
        res = oracle.residual(cur, oracle.mc(ref, mv, bs))
        assert sha(res) == str(g["%s_res_sha_%d" % (nm, bs)])
    mv, planes, ns = oracle.encode_p(cur, ref, 16)
    assert sha(planes) == str(g[nm + "_planes16_sha"])
    assert np.array_equal(oracle.decode_p(mv, planes, ref, 16), g[nm + "_full16"])


def test_colour_roundtrip_properties(oracle):
    """The (unpinned) OpenCV restatement: sanity properties only."""
    grey = np.repeat(np.arange(256, dtype=np.uint8)[:, None, None], 3, axis=2)
    ycc = oracle.bgr2ycrcb(grey)
    assert np.array_equal(ycc[..., 0], grey[..., 0])
    assert (ycc[..., 1:] == 128).all()
    assert np.array_equal(oracle.ycrcb2bgr(ycc), grey)


@pytest.mark.parametrize("sfx", ["", "2"])
def test_intra_small(oracle, sfx):
    """Intra prediction: integer restatement vs the reference's
    luma4x4 / luma16x16 / chroma8x8, NumPy-2 uint8 quirks included."""
    g = load_golden("intra_small.npz")
    Y, Cr, Cb = g["Y" + sfx], g["Cr" + sfx], g["Cb" + sfx]
    res, pred, modes = oracle.intra_luma4x4(Y)
    assert np.array_equal(modes, g["l4_modes" + sfx])
    assert np.array_equal(pred, g["l4_pred" + sfx]) and np.array_equal(res, g["l4_res" + sfx])
    assert len(np.unique(modes)) >= 8  # the fixture exercises (nearly) every mode
    res, pred, modes = oracle.intra_luma16x16(Y)
    assert np.array_equal(modes, g["l16_modes" + sfx])
    assert np.array_equal(pred, g["l16_pred" + sfx]) and np.array_equal(res, g["l16_res" + sfx])
    crres, crpred, cbres, cbpred, modes = oracle.intra_chroma8x8(Cr, Cb)
    assert np.array_equal(modes, g["c8_modes" + sfx])
    assert np.array_equal(crpred, g["c8_crpred" + sfx]) and np.array_equal(crres, g["c8_crres" + sfx])
    assert np.array_equal(cbpred, g["c8_cbpred" + sfx]) and np.array_equal(cbres, g["c8_cbres" + sfx])


def test_intra_transform4x4_matches_numpy(oracle):
    """The commented-out 4x4 integer transform of intraframe.py:352-358, restated with the
    same NumPy calls (no running upstream code: parity by arithmetic only)."""
    rng = np.random.default_rng(5)
    C = np.array([[1, 1, 1, 1], [2, 1, -1, -2], [1, -1, -1, 1], [1, -2, 2, -1]])
    X = rng.integers(-255, 256, (32, 48)).astype(np.float64)
    want = np.zeros_like(X)
    for i in range(0, 32, 4):
        for j in range(0, 48, 4):
            t = np.matmul(C, X[i:i + 4, j:j + 4])
            want[i:i + 4, j:j + 4] = np.round(np.matmul(t, C.transpose()))
    assert np.array_equal(oracle.intra_transform4x4(X), want)


def test_chroma420_matches_numpy_restatement(oracle):
    """vcso_chroma420 (chroma.py:8-21) against the same definition written with NumPy: reflect-101
    padding on the top / left, 2x2 sums, sum * 0.25 rounded half to even, [::2, ::2]."""
    rng = np.random.default_rng(5)
    for H, W in ((7, 10), (16, 16), (2, 4), (5, 2), (33, 18)):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        Y, cr, cb = oracle.chroma420(img)
        ycc = oracle.bgr2ycrcb(img)

        def box(pl):
            P = np.pad(pl.astype(np.int64), ((1, 0), (1, 0)), mode="reflect")
            s = P[:-1, :-1] + P[:-1, 1:] + P[1:, :-1] + P[1:, 1:]
            return np.rint(s * 0.25).astype(np.uint8)
        assert np.array_equal(Y, ycc[..., 0])
        assert np.array_equal(cr, box(ycc[..., 1])[::2, ::2]), (H, W)
        assert np.array_equal(cb, box(ycc[..., 2])[::2, ::2]), (H, W)
