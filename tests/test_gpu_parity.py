"""Parity of the HIP path (through the C ABI / the drop-in classes) against
(a) the golden vectors captured from the reference and (b) the CPU oracle on
seeded inputs.  Everything is bit-exact: motion vectors, MC image, residual,
float64 coefficient planes, decoded pixels.  Needs an MI355X: -m gpu."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def vh():
    import vcs_hip
    vcs_hip.load()
    assert vcs_hip.device_count() >= 1, "no GPU visible to libvcs_hip"
    return vcs_hip


def ctx_for(vh, frame, bs):
    H, W = frame.shape[:2]
    return vh.Context(H, W, bs)


# ---------------------------------------------------------------- golden ----
@pytest.mark.parametrize("bs", [4, 8, 16])
def test_golden_me_small(vh, bs):
    g = load_golden("me_small.npz")
    ref, cur = g["ref"], g["cur"]
    with ctx_for(vh, ref, bs) as c:
        c.set_ref(ref)
        mv, ns = c.me(cur)
        assert np.array_equal(mv, g["mv_%d" % bs])
        pred = c.mc(mv)
        assert np.array_equal(pred, g["mc_%d" % bs])
        assert np.array_equal(c.residual(cur, pred), g["res_%d" % bs])
        _, _, res, _ = c.encode_p(cur, want_planes=False, want_res=True)
        assert np.array_equal(res, g["res_%d" % bs])


def test_golden_me_edges(vh, oracle):
    g = load_golden("me_edges.npz")
    names = [str(n) for n in g["names"]]
    for k, name in enumerate(names):
        ref, cur, bs = g["c%d_ref" % k], g["c%d_cur" % k], int(g["c%d_bs" % k])
        with ctx_for(vh, ref, bs) as c:
            c.set_ref(ref)
            mv, ns, cost = c.me(cur, want_cost=True)
            assert np.array_equal(mv, g["c%d_mv" % k]), name
            omv, ons, ocost = oracle.me(cur, ref, bs, want_cost=True)
            assert ns == ons, name
            assert np.array_equal(cost, ocost), name
            pred = c.mc(mv)
            assert np.array_equal(pred, g["c%d_mc" % k]), name
            assert np.array_equal(c.residual(cur, pred), g["c%d_res" % k]), name


@pytest.mark.parametrize("bs", [16, 8])
def test_golden_me_1080p(vh, bs):
    from synth import synth_v1
    g = load_golden("me_1080p.npz")
    ref, cur = synth_v1(1080, 1920, seed=1234)
    if sha(ref) != str(g["ref_sha"]) or sha(cur) != str(g["cur_sha"]):
        pytest.skip("synthetic input drifted (numpy RNG differs from the fixture's)")
    with ctx_for(vh, ref, bs) as c:
        c.set_ref(ref)
        mv, planes, res, ns = c.encode_p(cur, want_planes=False, want_res=True)
        assert np.array_equal(mv, g["mv_%d" % bs])
        assert sha(res) == str(g["res_sha_%d" % bs])
        assert sha(c.mc(mv)) == str(g["mc_sha_%d" % bs])


@pytest.mark.parametrize("qf", [10, 50, 99])
def test_golden_dct_small(vh, qf):
    g = load_golden("dct_small.npz")
    img, Q = g["img"], g["Q_q%d" % qf]
    assert np.array_equal(vh.make_q(float(qf)), Q)
    assert np.array_equal(vh.dct_matrix(), g["dctmat"])
    with ctx_for(vh, img, 8) as c:
        c.set_params(Q=Q)
        planes = c.compress(img)
        assert planes.dtype == np.float64
        assert np.array_equal(planes, g["planes_q%d" % qf])
        assert np.array_equal(c.decompress(g["planes_q%d" % qf]), g["dec_q%d" % qf])


@pytest.mark.parametrize("resident_env", [None, "0"])
def test_golden_codec_dropin(vh, capsys, monkeypatch, resident_env):
    """The reference's own call sequence (main.py:29-50) on the drop-in classes, replaying a run of the
    reference's Encoder / Decoder.  Default (no keyword, no environment variable: what an unchanged main.py
    gets): Frame.mv / Frame.c are list-like views, Frame.r a list-like view of planes kept in HBM -- every
    access main.py:48-50 and decoder.py:52-69 make of them must give the reference's values.
    VCS_RESIDENT=0: real Python lists and NumPy planes, type for type as upstream."""
    import encoder as E
    import decoder as D
    if resident_env is None:
        monkeypatch.delenv("VCS_RESIDENT", raising=False)
    else:
        monkeypatch.setenv("VCS_RESIDENT", resident_env)
    g = load_golden("codec_small.npz")
    frames = g["frames"]
    H, W = frames.shape[1:3]
    for with_dct, tag in ((True, "dct"), (False, "raw")):
        enc = E.Encoder(pattern=["I", "P", "P"], shape=[H, W], block_size=8, with_DCT=with_dct)
        assert enc.resident == (resident_env is None)
        for n in range(len(frames)):
            enc.encode_frame(frames[n], n)
        assert [f.t for f in enc.encoded_frames] == ["I", "P", "P", "I", "P"]
        assert enc.ref_frames[0] is frames[0] or np.shares_memory(enc.ref_frames[0], frames[0])
        sink = []
        dec = D.Decoder(encoded_frames=enc.encoded_frames, fps=25.0, shape=[H, W],
                        ref_frames=enc.ref_frames, block_size=8, with_DCT=with_dct,
                        sink=sink.append)
        for n, fr in enumerate(enc.encoded_frames):
            assert fr.i == n and fr.ref_i == int(g["%s_refi_%d" % (tag, n)])
            if fr.t != "P":
                continue
            want_mv = g["%s_mv_%d" % (tag, n)]
            if resident_env == "0":
                assert isinstance(fr.mv, list) and isinstance(fr.c, list)
            # what a list of [dx, dy] pairs offers: len, indexing down to Python ints, iteration, ==
            assert len(fr.mv) == len(want_mv) == len(fr.c)
            assert isinstance(fr.mv[0][0], int) and isinstance(fr.c[0][1], int)
            assert fr.mv[3] == want_mv[3].tolist() and fr.mv == want_mv.tolist()
            assert [list(p) for p in fr.mv] == want_mv.tolist()
            assert np.array_equal(np.array(fr.mv), want_mv)
            if with_dct:
                if resident_env == "0":
                    assert isinstance(fr.r, list)
                assert len(fr.r) == 3
                assert np.array_equal(np.stack(list(fr.r)), g["dct_r_%d" % n])
                assert np.array_equal(fr.r[1], g["dct_r_%d" % n][1])
            else:
                assert np.array_equal(fr.r, g["raw_r_%d" % n])
            assert np.array_equal(dec._reconstruct_P_frame(fr, True), g["%s_full_%d" % (tag, n)])
            assert np.array_equal(dec._reconstruct_P_frame(fr, False),
                                  g["%s_mconly_%d" % (tag, n)])
        dec.reconstruct_video(with_residuals=True)
        assert len(sink) == 5
        assert np.array_equal(sink[4], g["%s_full_4" % tag])
    capsys.readouterr()


def test_default_mode_honours_the_aliased_i_frame(vh, oracle, capsys, monkeypatch):
    """encoder.py:42 stores the caller's I-frame ARRAY, so an in-place edit between two P frames reaches the
    second one.  The default mode (planes in HBM, pipelined) keeps that without any call: the whole I frame
    is hashed per P frame."""
    import encoder as E
    from synth import synth_v1
    monkeypatch.delenv("VCS_RESIDENT", raising=False)
    ref, cur = synth_v1(144, 176, seed=77)
    ref = ref.copy()
    enc = E.Encoder(pattern=["I", "P", "P"], shape=[144, 176], block_size=16, with_DCT=True, dct_block_size=8)
    enc.encode_frame(ref, 0)
    enc.encode_frame(cur, 1)
    want1 = oracle.me(cur, ref, 16)[0]
    assert np.array_equal(np.asarray(enc.encoded_frames[1].mv), want1)
    ref[40:90, 30:120] = np.roll(ref[40:90, 30:120], 7, axis=1)      # edit the I frame the encoder aliases
    enc.encode_frame(cur, 2)
    want2 = oracle.me(cur, ref, 16)[0]
    assert not np.array_equal(want1, want2)
    assert np.array_equal(np.asarray(enc.encoded_frames[2].mv), want2)
    assert np.array_equal(np.stack(list(enc.encoded_frames[2].r)), oracle.encode_p(cur, ref, 16)[1])
    capsys.readouterr()


def test_golden_motion_dropin_classes(vh, capsys):
    """MotionProcessor / DCTCompressor used stand-alone, as encoder.py does."""
    import motion as M
    import DCTcompressor as C
    g = load_golden("me_small.npz")
    ref, cur = g["ref"], g["cur"]
    mp = M.MotionProcessor(16, [ref.shape[0], ref.shape[1]])
    mvs, coords = mp.process_motion_prediction(cur, ref)
    assert np.array_equal(np.array(mvs), g["mv_16"])
    assert np.array_equal(np.array(coords), g["coords_16"])
    img = mp.reconstruct_from_motion_vectors(mvs, ref, coords)
    assert np.array_equal(img, g["mc_16"])
    res = mp.get_residuals(input_frame=cur, reconstructed=img)
    assert np.array_equal(res, g["res_16"])
    # a permuted coords list takes the explicit-coords kernel path
    perm = np.random.default_rng(0).permutation(len(mvs))
    img2 = mp.reconstruct_from_motion_vectors([mvs[i] for i in perm], ref,
                                              [coords[i] for i in perm])
    assert np.array_equal(img2, g["mc_16"])
    d = load_golden("dct_small.npz")
    c = C.DCTCompressor(8)
    out = c.compress(d["img"])
    assert np.array_equal(np.stack(out), d["planes_q50"])
    assert np.array_equal(c.decompress(compressed=out, imshape=d["img"].shape), d["dec_q50"])
    c.Q = list(d["Q_q10"])  # instances expose .Q; substituting tables is the "q sweep"
    assert np.array_equal(np.stack(c.compress(d["img"])), d["planes_q10"])
    with pytest.raises(ValueError):
        C.DCTCompressor(16).compress(d["img"])
    capsys.readouterr()


@pytest.mark.parametrize("nm", ["jump", "corgi"])
def test_golden_real_small(vh, nm):
    g = load_golden("real_small.npz")
    ref, cur = g[nm + "_ref"], g[nm + "_cur"]
    for bs in (8, 16):
        with ctx_for(vh, ref, bs) as c:
            c.set_ref(ref)
            mv, planes, res, ns = c.encode_p(cur, want_planes=(bs == 16), want_res=True)
            assert np.array_equal(mv, g["%s_mv_%d" % (nm, bs)])
            assert sha(res) == str(g["%s_res_sha_%d" % (nm, bs)])
            if bs == 16:
                assert sha(planes) == str(g[nm + "_planes16_sha"])
                assert np.array_equal(c.decode_p(mv, planes), g[nm + "_full16"])


def test_ref_swap_alternates_two_resident_references(vh, oracle):
    """vcs_ref_swap: two resident reference buffers (the multi-GPU host layer broadcasts the
    next GOP's I frame into the alternate one); each keeps its content and its own sum table."""
    from synth import synth_v1
    refA, cur = synth_v1(96, 128, seed=21)
    refB = np.roll(refA, (3, -5), axis=(0, 1))
    wantA, _ = oracle.me(cur, refA, 16)
    wantB, _ = oracle.me(cur, refB, 16)
    assert not np.array_equal(wantA, wantB)
    with vh.Context(96, 128, 16) as c:
        c.set_ref(refA)
        assert np.array_equal(c.me(cur)[0], wantA)
        p_cur, _ = c.dev_ptr(0)
        p_alt, _ = c.dev_ptr(3)
        c.ref_swap()                      # current: the (zeroed) alternate buffer
        assert c.dev_ptr(0)[0] == p_alt and c.dev_ptr(3)[0] == p_cur
        c.set_ref(refB)
        assert np.array_equal(c.me(cur)[0], wantB)
        for want in (wantA, wantB, wantA):
            c.ref_swap()
            assert np.array_equal(c.me(cur)[0], want)


@pytest.mark.parametrize("bs", [16, 8])
def test_next_reference_tables_cut_beside_the_search(vh, oracle, bs):
    """vcs_set_ref_next / vcs_ref_next_ready: the next GOP's I frame is staged in the alternate buffer and the
    bit-plane search's per-reference tables are cut on a side stream beside this GOP's search kernel.  Three
    GOPs with three DIFFERENT I frames, each staged while the previous GOP is encoded; a fourth whose I
    frame arrives late (plain set_ref after a swap: the shadow table of the stale content must not be used)."""
    from synth import synth_gop
    H, W, n = 144, 176, 7
    refs, gops = [], []
    for k in range(4):
        r, fr = synth_gop(H, W, n, seed=40 + k)
        refs.append(r)
        gops.append(fr)
    with vh.Context(H, W, bs) as c:
        c.gop_reserve(n)
        c.set_ref(refs[0])
        for k in range(3):
            if k + 1 < 3:
                c.set_ref_next(refs[k + 1])          # known early: its tables are cut beside GOP k's search
            c.gop_upload(0, gops[k])
            c.gop_encode(0, n)
            mv, ns = c.gop_get_mv(0, n)
            for f in range(n):
                want, wns = oracle.me(gops[k][f], refs[k], bs)
                assert np.array_equal(mv[f], want) and ns[f] == wns, "GOP %d frame %d" % (k, f)
            assert np.array_equal(c.gop_get_planes(n - 1), oracle.encode_p(gops[k][n - 1], refs[k], bs)[1])
            if k + 1 < 3:
                c.ref_swap()
        # declared ready, then overwritten after the swap: the table cut from the old content is stale
        c.set_ref_next(refs[0])
        c.gop_upload(0, gops[2])
        c.gop_encode(0, n)                           # (cuts refs[0]'s table in the shadow)
        c.ref_swap()
        c.set_ref(refs[3])                           # ... but the GOP's I frame turns out to be another one
        c.gop_upload(0, gops[3])
        c.gop_encode(0, n)
        mv, ns = c.gop_get_mv(0, n)
        for f in range(n):
            want, wns = oracle.me(gops[3][f], refs[3], bs)
            assert np.array_equal(mv[f], want) and ns[f] == wns, "late I frame, frame %d" % f


def test_l3_prefetch_modes_change_nothing(vh, oracle):
    """vcs_set_l3_prefetch: where the batch's frames are brought into the Infinity Cache for the tail (inside the
    search launch, a separate pass, not at all) is scheduling only -- vectors and planes are the same."""
    from synth import synth_gop
    H, W, n = 144, 176, 7
    ref, frames = synth_gop(H, W, n, seed=8)
    want_mv = [oracle.me(frames[k], ref, 16)[0] for k in range(n)]
    want_planes = oracle.encode_p(frames[n - 1], ref, 16)[1]
    with vh.Context(H, W, 16) as c:
        c.set_ref(ref)
        c.gop_reserve(n)
        c.gop_upload(0, frames)
        for mode in (0, 1, 2, -1):
            c.set_l3_prefetch(mode)
            c.gop_encode(0, n)
            mv, _ = c.gop_get_mv(0, n)
            assert all(np.array_equal(mv[k], want_mv[k]) for k in range(n)), mode
            assert np.array_equal(c.gop_get_planes(n - 1), want_planes), mode
        with pytest.raises(vh.VcsError):
            c.set_l3_prefetch(3)


def test_pinned_host_buffers(vh, oracle):
    """vcs_host_alloc (frame feeder): frames produced into page-locked memory go
    through the same entry points with the same results; the memory is released with the array."""
    import gc
    from synth import synth_v1
    ref, cur = synth_v1(96, 128, seed=31)
    pref, pcur = vh.pinned_empty(ref.shape), vh.pinned_empty(cur.shape)
    assert pref.flags["C_CONTIGUOUS"] and pref.dtype == np.uint8 and pref.shape == ref.shape
    pref[...] = ref
    pcur[...] = cur
    planes_out = vh.pinned_empty((3, 96, 128), np.float64)
    with vh.Context(96, 128, 16) as c:
        c.set_ref(pref)
        mv, planes, _, ns = c.encode_p(pcur)
        omv, oplanes, ons = oracle.encode_p(cur, ref, 16)
        assert np.array_equal(mv, omv) and ns == ons and np.array_equal(planes, oplanes)
        planes_out[...] = planes
        assert np.array_equal(c.decode_p(mv, planes_out), oracle.decode_p(omv, oplanes, ref, 16))
    view = pcur[10:20]
    del pref, pcur, planes_out
    gc.collect()
    assert view.sum() == cur[10:20].sum()      # a view keeps the allocation alive
    with pytest.raises(ValueError):
        vh.pinned_empty((0, 4))


def _psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


@pytest.mark.parametrize("nm", ["jump", "corgi"])
def test_psnr_matches_reference(vh, nm):
    """BASELINE north_star: reconstruction within 1e-4 dB PSNR of the reference.  The
    decoded frames are bit-identical (tests above), so the float64-plane path differs by
    exactly 0.  The reduced-precision plane formats are NOT in the reference and are not
    parity cases: its decoder truncates float64 to uint8 with wrap-around
    (DCTcompressor.py:88), so a last-bit change of a coefficient can flip a pixel by 1 or
    by 255 -- float32 planes land within a dB (either side), integer levels are a
    different operating point altogether.  Both are only sanity-bounded here."""
    g = load_golden("real_small.npz")
    ref, cur = g[nm + "_ref"], g[nm + "_cur"]
    want = _psnr(g[nm + "_full16"], cur)          # what the reference's decoder achieves
    with ctx_for(vh, ref, 16) as c:
        c.set_ref(ref)
        mv, planes, _, _ = c.encode_p(cur)
        assert abs(_psnr(c.decode_p(mv, planes), cur) - want) <= 1e-4
        c.set_params(coeff_mode=vh.VCS_COEFF_F32)
        mv, p32, _, _ = c.encode_p(cur)
        assert abs(_psnr(c.decode_p(mv, p32), cur) - want) < 2.0
        c.set_params(coeff_mode=vh.VCS_COEFF_I16)
        mv, p16, _, _ = c.encode_p(cur)
        assert _psnr(c.decode_p(mv, p16), cur) > 4.0


# ---------------------------------------------------------------- oracle ----
@pytest.mark.parametrize("H,W,bs", [
    (64, 64, 16), (72, 88, 8), (50, 70, 8), (70, 90, 16), (30, 26, 4), (33, 47, 5),
    (40, 52, 6), (60, 72, 12), (96, 160, 32), (17, 23, 2), (41, 39, 3), (128, 96, 16),
    (21, 64, 7), (200, 328, 16), (136, 240, 8)])
def test_me_vs_oracle_shapes(vh, oracle, H, W, bs):
    rng = np.random.default_rng(H * 1000 + W + bs)
    for trial in range(3):
        ref = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        if trial == 0:
            cur = np.roll(ref, (rng.integers(-6, 7), rng.integers(-6, 7)), axis=(0, 1))
        elif trial == 1:
            cur = ref.copy()
            m = rng.random((H, W, 1)) < 0.3
            cur = np.where(m, rng.integers(0, 256, (H, W, 3), dtype=np.uint8), cur)
        else:
            cur = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        with vh.Context(H, W, bs) as c:
            c.set_ref(ref)
            mv, ns, cost = c.me(cur, want_cost=True)
            omv, ons, ocost = oracle.me(cur, ref, bs, want_cost=True)
            assert np.array_equal(mv, omv)
            assert ns == ons
            assert np.array_equal(cost, ocost)
            assert np.array_equal(c.mc(mv), oracle.mc(ref, omv, bs))


def test_me_params_vs_oracle(vh, oracle):
    """Non-default window / stride / threshold (the `step=1` dense search of
    SURVEY 8f N4 included): oracle is the only checker here."""
    from synth import synth_v1
    ref, cur = synth_v1(96, 128, seed=5)
    for bs, win, step, thr in [(16, 32, 1, 2000), (16, 16, 2, 0), (8, 16, 1, 500),
                               (16, 40, 5, 10**12), (4, 8, 1, 100), (8, 24, 4, 2000)]:
        with vh.Context(96, 128, bs) as c:
            c.set_params(win=win, step=step, sim_thresh=thr)
            c.set_ref(ref)
            mv, ns, cost = c.me(cur, want_cost=True)
            omv, ons, ocost = oracle.me(cur, ref, bs, win=win, step=step,
                                        thresh=min(thr, 2**32 - 1), want_cost=True)
            assert np.array_equal(mv, omv), (bs, win, step, thr)
            assert ns == ons and np.array_equal(cost, ocost)
    with vh.Context(96, 128, 8) as c:  # negative threshold: nothing can be static
        c.set_params(sim_thresh=-1)
        c.set_ref(ref)
        mv, ns = c.me(ref)
        assert ns == 0


@pytest.mark.parametrize("me_bs", [8, 16])
def test_encode_decode_vs_oracle(vh, oracle, me_bs):
    from synth import synth_v1
    ref, cur = synth_v1(136, 200, seed=77)
    for qf in (10.0, 50.0, 99.0):
        Q = oracle.make_q(qf)
        with vh.Context(136, 200, me_bs) as c:
            c.set_params(Q=Q)
            c.set_ref(ref)
            mv, planes, res, ns = c.encode_p(cur, want_planes=True, want_res=True)
            omv, oplanes, ons, ores = oracle.encode_p(cur, ref, me_bs, Q=Q, want_res=True)
            assert np.array_equal(mv, omv) and ns == ons
            assert np.array_equal(res, ores)
            assert np.array_equal(planes, oplanes)
            assert np.array_equal(c.decode_p(mv, planes), oracle.decode_p(omv, oplanes, ref, me_bs, Q=Q))
            assert np.array_equal(c.decode_p(mv, None), oracle.mc(ref, omv, me_bs))
            assert np.array_equal(c.compress(cur), oracle.compress(cur, Q))
            assert np.array_equal(c.decompress(oplanes), oracle.decompress(oplanes, Q))


def test_dct_extremes_vs_oracle(vh, oracle):
    rng = np.random.default_rng(3)
    H, W = 64, 72
    imgs = [np.zeros((H, W, 3), np.uint8), np.full((H, W, 3), 255, np.uint8),
            rng.integers(0, 256, (H, W, 3), dtype=np.uint8),
            (rng.integers(0, 2, (H, W, 3)) * 255).astype(np.uint8)]
    with vh.Context(H, W, 8) as c:
        for img in imgs:
            planes = c.compress(img)
            assert np.array_equal(planes, oracle.compress(img))
            assert np.array_equal(c.decompress(planes), oracle.decompress(planes))
        # arbitrary (not DCT-produced) planes exercise the truncating store + wraps
        wild = rng.normal(0, 40, (3, H, W))
        assert np.array_equal(c.decompress(wild), oracle.decompress(wild))


def test_planes_identical_to_the_bit(vh, oracle):
    """float64 planes compared as raw 64-bit patterns: np.array_equal calls -0.0 and +0.0
    equal, a stored bitstream would not.  Flat and half-flat frames produce exact zeros
    (np.true_divide keeps the numerator's sign, DCTcompressor.py:71; the exact 3-FMA
    division of k_fdct would turn a -0.0 numerator into +0.0, but fma chains that start
    from +0.0 never produce one)."""
    rng = np.random.default_rng(17)
    H, W = 64, 128
    flat = np.full((H, W, 3), 97, np.uint8)
    half = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    half[:, W // 2:] = (200, 3, 77)
    dark = np.zeros((H, W, 3), np.uint8)
    dark[::2] = 255
    n_zero = 0
    with vh.Context(H, W, 8) as c:
        for img in (flat, half, dark, rng.integers(0, 256, (H, W, 3), dtype=np.uint8)):
            want = oracle.compress(img)
            got = c.compress(img)
            assert np.array_equal(got.view(np.int64), want.view(np.int64))
            n_zero += int(np.count_nonzero(want == 0))
            assert not np.any((want == 0) & np.signbit(want))
    assert n_zero > 0  # exact zeros did occur


def test_coeff_modes_vs_oracle(vh, oracle):
    from synth import synth_v1
    ref, cur = synth_v1(64, 96, seed=11)
    omv, oplanes, ons = oracle.encode_p(cur, ref, 16)
    with vh.Context(64, 96, 16) as c:
        c.set_ref(ref)
        c.set_params(coeff_mode=vh.VCS_COEFF_F32)
        mv, planes, _, _ = c.encode_p(cur)
        assert planes.dtype == np.float32
        assert np.array_equal(planes, oplanes.astype(np.float32))
        c.set_params(coeff_mode=vh.VCS_COEFF_I16)
        mv, planes, _, _ = c.encode_p(cur)
        assert planes.dtype == np.int16
        assert np.array_equal(planes, oracle.round_i16(oplanes))
        # rounded levels decode like float64 planes holding the same integers
        assert np.array_equal(c.decode_p(mv, planes),
                              oracle.decode_p(omv, planes.astype(np.float64), ref, 16))


# ------------------------------------------------------------- GOP / 1080p ----
def test_gop_1080p_vs_oracle(vh, oracle):
    """BASELINE workload shape: 1080p, 16x16 motion blocks (+-32), 8x8 DCT."""
    from synth import synth_gop
    H, W, n = 1080, 1920, 4
    ref, frames = synth_gop(H, W, n, seed=1234)
    with vh.Context(H, W, 16) as c:
        c.set_ref(ref)
        c.gop_reserve(n, with_decode=True)
        c.gop_upload(0, frames)
        c.gop_encode(0, n)
        c.gop_decode(0, n)
        mv, ns = c.gop_get_mv(0, n)
        for k in range(n):
            omv, ons = oracle.me(frames[k], ref, 16)
            assert np.array_equal(mv[k], omv), k
            assert ns[k] == ons
        # property at full size: every vector lies on the reference's stride-5 grid
        assert ((mv[..., 0] + 32) % 5 == 0)[:, 2 * 120 + 2].all()
        for k in (0, n - 1):
            _, oplanes, _ = oracle.encode_p(frames[k], ref, 16)
            planes = c.gop_get_planes(k)
            assert np.array_equal(planes, oplanes), k
            dec = c.gop_get_decoded(k)
            assert np.array_equal(dec, oracle.decode_p(mv[k], oplanes, ref, 16))
            # Round-trip PROPERTY (the parity assertions are the exact ones above).  The
            # quantiser never rounds, so the only losses are the truncating float64 -> uint8
            # store of DCTcompressor.py:88 (up to -1 in each of Y, Cr, Cb) and the two integer
            # colour conversions; through the inverse matrix (B = Y + 1.77 (Cb - 128), ...) that
            # is a few levels per channel.  Measured on this content: max 6, mean 1.03 -- the
            # bounds leave room for other content, they are not a tolerance on parity.
            err = np.abs(dec.astype(np.int16) - frames[k].astype(np.int16))
            err = np.minimum(err, 256 - err)
            assert err.max() <= 6 and err.mean() < 1.5


def test_gop_matches_single_frame_calls(vh):
    from synth import synth_gop
    H, W, n = 144, 176, 5
    ref, frames = synth_gop(H, W, n, seed=3)
    with vh.Context(H, W, 8) as c:
        c.set_ref(ref)
        c.gop_reserve(n)
        c.gop_upload(0, frames[:2])
        c.gop_upload(2, frames[2:])
        c.gop_encode(0, 3)
        c.gop_encode(3, 2)
        mv, ns = c.gop_get_mv(0, n)
        for k in range(n):
            smv, splanes, _, sns = c.encode_p(frames[k])
            assert np.array_equal(mv[k], smv) and ns[k] == sns
            assert np.array_equal(c.gop_get_planes(k), splanes)


def test_residual_roundtrip_property_1080p(vh):
    """pred + (cur - pred) == cur with uint8 wrap, at full size."""
    from synth import synth_v1
    ref, cur = synth_v1(1080, 1920, seed=9)
    with vh.Context(1080, 1920, 16) as c:
        c.set_ref(ref)
        mv, _, res, _ = c.encode_p(cur, want_planes=False, want_res=True)
        pred = c.mc(mv)
        assert np.array_equal(c.residual(pred, res, +1), cur)
        assert (pred[1072:] == 0).all()  # 1080 % 16 rows are never covered (motion.py:83-87)
        assert np.array_equal(res[1072:], cur[1072:])


# ------------------------------------------------------------- errors ----
def test_error_paths(vh):
    with pytest.raises(vh.VcsError):
        vh.Context(64, 64, 1)
    with pytest.raises(vh.VcsError):
        vh.Context(64, 64, 8, dct_bs=4)
    with vh.Context(64, 64, 16) as c:
        with pytest.raises(vh.VcsError) as e:
            c.me(np.zeros((64, 64, 3), np.uint8))
        assert e.value.code == -4  # VCS_E_STATE: no reference
        c.set_ref(np.zeros((64, 64, 3), np.uint8))
        bad = np.zeros((16, 2), np.int16)
        bad[5] = (100, 0)
        with pytest.raises(vh.VcsError) as e:
            c.mc(bad)
        assert e.value.code == -5  # VCS_E_RANGE
        with pytest.raises(vh.VcsError):
            c.decode_p(bad, None)
        with pytest.raises(ValueError):
            c.me(np.zeros((32, 64, 3), np.uint8))
        with pytest.raises(vh.VcsError):
            c.set_params(win=500)
    with vh.Context(50, 70, 8) as c:  # DCT path refuses non-multiples of 8
        with pytest.raises(vh.VcsError):
            c.compress(np.zeros((50, 70, 3), np.uint8))


def test_gop_encoder_single_rank(vh, oracle):
    """gop.DistGopEncoder with the product (HIP) backend, world size 1."""
    from gop import DistGopEncoder
    from synth import synth_gop
    H, W, n = 144, 176, 5
    ref, frames = synth_gop(H, W, n, seed=8)
    enc = DistGopEncoder(H, W, 16, n_frames=n)
    mv, ns = enc.encode_gop(ref, frames)
    for k in range(n):
        omv, oplanes, ons = oracle.encode_p(frames[k], ref, 16)
        assert np.array_equal(mv[k], omv) and ns[k] == ons
        assert np.array_equal(enc.planes(k), oplanes)
    enc.close()


@pytest.mark.parametrize("sfx", ["", "2"])
def test_golden_intra(vh, sfx):
    """Next row N1 / BASELINE config 5: intra prediction vs the reference's outputs."""
    g = load_golden("intra_small.npz")
    Y, Cr, Cb = g["Y" + sfx], g["Cr" + sfx], g["Cb" + sfx]
    sys_path = __import__("sys").path
    idir = __import__("os").path.join(__import__("conftest").PKG, "IntraframeCompression")
    if idir not in sys_path:
        sys_path.insert(0, idir)
    import intraframe as I
    res, pred, modes = I.luma4x4(Y)
    assert np.array_equal(modes, g["l4_modes" + sfx])
    assert np.array_equal(pred, g["l4_pred" + sfx]) and np.array_equal(res, g["l4_res" + sfx])
    res, pred, modes = I.luma16x16(Y)
    assert np.array_equal(modes, g["l16_modes" + sfx])
    assert np.array_equal(pred, g["l16_pred" + sfx]) and np.array_equal(res, g["l16_res" + sfx])
    crres, crpred, cbres, cbpred, modes = I.chroma8x8(Cr, Cb)
    assert np.array_equal(modes, g["c8_modes" + sfx])
    assert np.array_equal(crpred, g["c8_crpred" + sfx]) and np.array_equal(crres, g["c8_crres" + sfx])
    assert np.array_equal(cbpred, g["c8_cbpred" + sfx]) and np.array_equal(cbres, g["c8_cbres" + sfx])


def test_intra_1080p_vs_oracle(vh, oracle):
    """Config 5 shape: a 1080p I frame resized to multiples of 16 (1920 x 1072)."""
    rng = np.random.default_rng(4)
    H, W = 1072, 1920
    yy, xx = np.mgrid[0:H, 0:W]
    base = (128 + 70 * np.sin(xx / 23.0) * np.cos(yy / 17.0)).astype(np.int16)
    planes = [np.clip(base + rng.integers(-30, 31, (H, W)), 0, 255).astype(np.uint8) for _ in range(3)]
    with vh.Context(H, W, 16) as c:
        for bs, fn in ((4, oracle.intra_luma4x4), (16, oracle.intra_luma16x16)):
            got, want = c.intra_luma(planes[0], bs), fn(planes[0])
            for a, b in zip(got, want):
                assert np.array_equal(a, b)
        got, want = c.intra_chroma(planes[1], planes[2]), oracle.intra_chroma8x8(planes[1], planes[2])
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        # "+ DCT" of config 5: the 4x4 integer core transform of the luma residual
        # (commented out upstream, intraframe.py:352-358 -- parity by arithmetic only)
        yres = c.intra_luma(planes[0], 4)[0]
        coef = c.intra_transform4x4(yres)
        assert np.array_equal(coef, oracle.intra_transform4x4(yres))
        C = np.array([[1, 1, 1, 1], [2, 1, -1, -2], [1, -1, -1, 1], [1, -2, 2, -1]])
        blk = yres[8:12, 20:24]
        assert np.array_equal(coef[8:12, 20:24], np.round(np.matmul(np.matmul(C, blk), C.transpose())))


def test_intra_batch_vs_oracle(vh, oracle):
    """vcs_intra_batch_*: the intra kernels over several I frames in one launch each (blockIdx.y = frame) --
    every frame's results against the oracle, a sub-range run, and the error paths."""
    rng = np.random.default_rng(11)
    H, W, n = 96, 160, 5
    Y, Cr, Cb = [rng.integers(0, 256, (n, H, W)).astype(np.uint8) for _ in range(3)]
    Y[2] = 77                                       # a flat frame: ties between the predictors
    Cb[3, :, :] = (np.arange(W) % 256).astype(np.uint8)
    with vh.Context(H, W, 16) as c:
        with pytest.raises(vh.VcsError):
            c.intra_batch_run(0, 1)                 # nothing reserved
        c.intra_batch_reserve(n)
        c.intra_batch_upload(0, Y, Cr, Cb)
        c.intra_batch_run(0, n)                     # all four kernels over all frames
        for f in range(n):
            for which, want in ((0, oracle.intra_luma4x4(Y[f])), (1, oracle.intra_luma16x16(Y[f])),
                                (2, oracle.intra_chroma8x8(Cr[f], Cb[f]))):
                got = c.intra_batch_get(f, which)
                assert len(got) == len(want)
                for a, b in zip(got, want):
                    assert np.array_equal(a, b), (f, which)
            assert np.array_equal(c.intra_batch_get(f, 3), oracle.intra_transform4x4(oracle.intra_luma4x4(Y[f])[0])), f
        # frames 3..4 replaced and re-run alone: the others keep their results
        Y2 = rng.integers(0, 256, (2, H, W)).astype(np.uint8)
        c.intra_batch_upload(3, Y2, Cr[3:], Cb[3:])
        c.intra_batch_run(3, 2, vh.Context.INTRA_LUMA4)
        assert np.array_equal(c.intra_batch_get(4, 0)[0], oracle.intra_luma4x4(Y2[1])[0])
        assert np.array_equal(c.intra_batch_get(1, 0)[0], oracle.intra_luma4x4(Y[1])[0])
        with pytest.raises(vh.VcsError):
            c.intra_batch_run(4, 2)                 # past the reserved frames
        with pytest.raises(vh.VcsError):
            c.intra_batch_run(0, 1, 8)              # the transform needs the luma 4x4 residual of the same call
    with vh.Context(72, 160, 8) as c:
        with pytest.raises(vh.VcsError):
            c.intra_batch_reserve(2)                # 72 is not a multiple of 16


def test_intra_transform_dropin(vh):
    sys_path = __import__("sys").path
    idir = __import__("os").path.join(__import__("conftest").PKG, "IntraframeCompression")
    if idir not in sys_path:
        sys_path.insert(0, idir)
    import intraframe as I
    rng = np.random.default_rng(8)
    Yres = rng.integers(-255, 256, (24, 40)).astype(np.float64)
    want = np.zeros_like(Yres)
    for i in range(0, 24, 4):          # the reference's commented-out loop, verbatim semantics
        for j in range(0, 40, 4):
            t = np.matmul(I.DCTTransform, Yres[i:i + 4, j:j + 4])
            want[i:i + 4, j:j + 4] = np.round(np.matmul(t, I.DCTTransform.transpose()))
    assert np.array_equal(I.transform4x4(Yres), want)
    with pytest.raises(Exception):
        I.transform4x4(np.zeros((6, 8)))


def test_empty_and_tiny_frames(vh, oracle):
    """Frames smaller than one motion block (N = 0) and frames too narrow for the
    specialised kernels."""
    rng = np.random.default_rng(12)
    with vh.Context(8, 8, 16) as c:  # no full block at all
        ref = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
        cur = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
        c.set_ref(ref)
        mv, ns = c.me(cur)
        assert mv.shape == (0, 2) and ns == 0
        mv, planes, res, ns = c.encode_p(cur, want_planes=True, want_res=True)
        assert np.array_equal(res, cur)  # prediction is all zero (motion.py:45-46)
        assert np.array_equal(planes, oracle.compress(cur))
        assert np.array_equal(c.decode_p(mv, planes), oracle.decompress(planes))
    for H, W, bs in [(16, 16, 16), (16, 24, 16), (32, 28, 16), (8, 12, 8), (24, 16, 8)]:
        ref = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        cur = np.roll(ref, (1, 2), axis=(0, 1))
        with vh.Context(H, W, bs) as c:
            c.set_ref(ref)
            mv, ns, cost = c.me(cur, want_cost=True)
            omv, ons, ocost = oracle.me(cur, ref, bs, want_cost=True)
            assert np.array_equal(mv, omv) and ns == ons and np.array_equal(cost, ocost), (H, W, bs)


def test_4k_frame_vs_oracle(vh, oracle):
    """BASELINE config 4 shape (3840 x 2160): largest frame of the configs."""
    from synth import synth_v1
    ref, cur = synth_v1(2160, 3840, seed=4)
    for bs in (16, 8):
        with vh.Context(2160, 3840, bs) as c:
            c.set_ref(ref)
            mv, ns = c.me(cur)
            omv, ons = oracle.me(cur, ref, bs)
            assert np.array_equal(mv, omv) and ns == ons
    with vh.Context(2160, 3840, 16) as c:
        c.set_ref(ref)
        mv, planes, _, _ = c.encode_p(cur)
        _, oplanes, _ = oracle.encode_p(cur, ref, 16)
        assert np.array_equal(planes, oplanes)


def test_static_heavy_content(vh, oracle):
    """Real video is mostly static blocks: the early-out path at 1080p."""
    from synth import synth_static
    ref, cur = synth_static(1080, 1920, seed=2, bs=16, frac=0.7)
    for bs in (16, 8):
        with vh.Context(1080, 1920, bs) as c:
            c.set_ref(ref)
            mv, ns = c.me(cur)
            omv, ons = oracle.me(cur, ref, bs)
            assert np.array_equal(mv, omv) and ns == ons
            assert ns > 0.3 * len(mv)


def test_resident_planes_dropin(vh, capsys):
    """Next row N3: Frame.r left in HBM; Decoder decodes straight from it; the
    lazy view still behaves like the reference's list of three float64 planes."""
    import encoder as E
    import decoder as D
    g = load_golden("codec_small.npz")
    frames = g["frames"]
    H, W = frames.shape[1:3]
    enc = E.Encoder(pattern=["I", "P", "P"], shape=[H, W], block_size=8, with_DCT=True,
                    resident=True)
    for n in range(len(frames)):
        enc.encode_frame(frames[n], n)
    dec = D.Decoder(encoded_frames=enc.encoded_frames, fps=25.0, shape=[H, W],
                    ref_frames=enc.ref_frames, block_size=8, with_DCT=True, sink=[].append)
    for n, fr in enumerate(enc.encoded_frames):
        if fr.t != "P":
            continue
        assert isinstance(fr.r, vh.ResidentPlanes) and fr.r._host is None  # nothing downloaded
        # Frame.mv / Frame.c: list-like views (frame.RowList) that read like upstream's lists
        want_mv = g["dct_mv_%d" % n]
        assert not isinstance(fr.mv, list) and len(fr.mv) == len(want_mv) == len(fr.c)
        assert fr.mv == want_mv.tolist() and fr.mv[3] == want_mv[3].tolist()
        assert isinstance(fr.mv[0][0], int) and fr.mv[1:3] == want_mv[1:3].tolist()
        assert [list(p) for p in fr.mv] == want_mv.tolist()
        assert fr.c[1] == [8, 0] and fr.c.tolist()[W // 8] == [0, 8]
        assert np.array_equal(np.array(fr.mv), g["dct_mv_%d" % n])
        assert np.array_equal(dec._reconstruct_P_frame(fr, True), g["dct_full_%d" % n])
        assert fr.r._host is None                                         # ... still nothing
        assert len(fr.r) == 3 and np.array_equal(np.stack(list(fr.r)), g["dct_r_%d" % n])
        assert np.array_equal(fr.r[1], g["dct_r_%d" % n][1])
    capsys.readouterr()


def test_gop_1080p_rolling_kernel(vh, oracle):
    """13 frames at 1080p: enough column segments for the rolling-window kernel
    (k_me16r) to be selected by default."""
    from synth import synth_gop
    H, W, n = 1080, 1920, 13
    ref, frames = synth_gop(H, W, n, seed=77)
    with vh.Context(H, W, 16) as c:
        c.set_ref(ref)
        c.gop_reserve(n)
        c.gop_upload(0, frames)
        c.gop_encode(0, n)
        mv, ns = c.gop_get_mv(0, n)
        for k in range(n):
            omv, ons = oracle.me(frames[k], ref, 16)
            assert np.array_equal(mv[k], omv), k
            assert ns[k] == ons
        _, oplanes, _ = oracle.encode_p(frames[n - 1], ref, 16)
        assert np.array_equal(c.gop_get_planes(n - 1), oplanes)
        # a second pass over the same slots, with frames the static test accepts: the
        # per-frame static counters restart from zero (k_me16r's primer zeroes them)
        frames2 = frames.copy()
        frames2[3] = ref
        frames2[7] = ref
        c.gop_upload(0, frames2)
        c.gop_encode(0, n)
        mv2, ns2 = c.gop_get_mv(0, n)
        assert ns2[3] == c.N and ns2[7] == c.N and not mv2[3].any()
        keep = [k for k in range(n) if k not in (3, 7)]
        assert np.array_equal(ns2[keep], ns[keep]) and np.array_equal(mv2[keep], mv[keep])


@pytest.mark.parametrize("shape,n,bs", [((360, 640), 7, 16), ((200, 328), 13, 16), ((1080, 1920), 9, 16), ((48, 64), 6, 16),
                                        ((144, 176), 33, 16), ((360, 640), 9, 8), ((200, 328), 17, 8), ((1080, 1920), 8, 8),
                                        ((24, 32), 8, 8), ((144, 176), 35, 8)])
def test_bitplane_search_gop_batches(vh, oracle, shape, n, bs):
    """k_bits_ref / k_bits_cur / k_me16b (the default 16x16 search from 6 frames on) and k_bits_ref8 /
    k_bits_cur8 / k_me8b (8x8, from 8 frames on; more than 16 frames = two frame groups): batches whose
    frames are dealt to the waves unevenly (7; 13 = 7 + 6; 9 = 5 + 4; 33 = 7 + 7 + 7 + 6 + 6),
    pictures with ragged right / bottom edges and fewer than 64 candidates per block, frames whose
    blocks are mostly static, wholly static or not static at all, cost output on a single frame
    batch, and the threshold variants (negative: nothing is static; huge: everything is)."""
    from synth import synth_gop
    H, W = shape
    ref, frames = synth_gop(H, W, n, seed=31)
    frames = np.ascontiguousarray(frames)
    rng = np.random.default_rng(5)
    for k in (1, 2):                                  # about 70 % / 40 % of the blocks (nearly) static
        keep = rng.random((H // bs + 1, W // bs + 1)) < (0.7 if k == 1 else 0.4)
        m = np.kron(keep, np.ones((bs, bs), bool))[:H, :W]
        noisy = np.clip(ref.astype(np.int16) + rng.integers(-1, 2, ref.shape), 0, 255).astype(np.uint8)
        frames[k][m] = noisy[m]
    frames[3] = ref                                   # every block static
    frames[n - 1][: H // 2] = ref[: H // 2]          # the upper half static
    with vh.Context(H, W, bs) as c:
        c.gop_reserve(n)
        c.gop_upload(0, frames)
        for thresh in (2000, 0, 10 ** 9, -1):
            c.set_params(sim_thresh=thresh)
            c.set_ref(ref)
            c.gop_encode(0, n)
            mv, ns = c.gop_get_mv(0, n)
            if thresh < 0:                            # nothing is static; frames without zero-difference
                assert not ns.any()                   # blocks then equal the threshold-0 result
                for k in (0, 4):
                    omv, ons = oracle.me(frames[k], ref, bs, thresh=0)
                    assert ons == 0 and np.array_equal(mv[k], omv), (thresh, k)
                continue
            for k in range(n):
                omv, ons = oracle.me(frames[k], ref, bs, thresh=min(thresh, 2 ** 32 - 1))
                assert np.array_equal(mv[k], omv), (thresh, k)
                assert ns[k] == ons, (thresh, k)
            if thresh == 2000:
                assert ns[3] == c.N and ns[1] > 0.3 * c.N
                _, oplanes, _ = oracle.encode_p(frames[n - 1], ref, bs)
                assert np.array_equal(c.gop_get_planes(n - 1), oplanes)
                # the same slots again: the static counters restart from zero
                c.gop_encode(0, n)
                mv2, ns2 = c.gop_get_mv(0, n)
                assert np.array_equal(mv2, mv) and np.array_equal(ns2, ns)
        # the per-reference sum(r) table has two makers (k_me16_sums / k_me8<sums> for single frames, the
        # batch's own k_bits_ref + k_bits_cur at 16x16): either one's table serves the other's search
        c.set_params(sim_thresh=2000)
        c.set_ref(ref)
        c.gop_encode(0, n)                              # new reference: the batch makes the table
        m1, s1 = c.me(frames[0])                        # single frame on the same reference
        assert np.array_equal(m1, oracle.me(frames[0], ref, bs)[0])
        c.set_ref(frames[4])                            # new reference: the single-frame kernel makes it
        m2, s2 = c.me(frames[0])
        assert np.array_equal(m2, oracle.me(frames[0], frames[4], bs)[0])
        c.gop_encode(0, n)                              # ... and the batch uses it
        mv3, ns3 = c.gop_get_mv(0, n)
        for k in (0, n - 1):
            omv, ons = oracle.me(frames[k], frames[4], bs)
            assert np.array_equal(mv3[k], omv) and ns3[k] == ons, k


def test_forced_kernel_variants_subprocess(oracle):
    """The same parity cases with every specialised search kernel forced on small and
    ragged frames (VCS_ME16_ROLL=2 and friends), with them disabled (=0), with the generic
    kernel forced (VCS_ME_GENERIC=1), and the DCT cases with the IEEE division
    sequence instead of the exact 3-FMA one (VCS_FAST_DIV=0): one child process per setting."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sel = ("golden_me_small or golden_me_edges or me_vs_oracle_shapes or static_heavy or "
           "4k_frame or empty_and_tiny or golden_real_small or gop_1080p_vs_oracle")
    sel_small = ("golden_me_small or golden_me_edges or me_vs_oracle_shapes or empty_and_tiny or golden_real_small or "
                 "gop_1080p_vs_oracle")   # without the 4K frame and the 1080p static-heavy pair (minutes over all variants)
    dct = "golden_dct_small or encode_decode_vs_oracle or dct_extremes or coeff_modes"
    # ROLL=2: the shared-window kernels (k_me16g, k_me8g) on every frame size; ROLL=2 with
    # GROUP=0: the private-window rolling kernel (k_me16r); ROLL=0: k_me16 / k_me8
    # BITS=2: the bit-plane search (k_bits_ref, k_bits_cur, k_me16b) on every 16x16 launch, single
    # frames and ragged sizes included; BITS=0: never (the kernels below it run GOP-sized batches too)
    variants = (({"VCS_ME16_BITS": "2", "VCS_ME8_BITS": "2"}, sel + " or gop_matches_single or bitplane"),
                       ({"VCS_ME8_BITS": "0", "VCS_ME16_ROLL": "2"}, "bitplane or golden_me_small"),
                       ({"VCS_ME16_BITS": "2", "VCS_BITS_PF": "0"}, "gop_1080p_vs_oracle or gop_matches_single"),
                       # the chunk table in many narrow bands of block columns (what 4K does with four)
                       ({"VCS_ME16_BITS": "2", "VCS_BITS_BAND_KB": "1024"}, "bitplane or golden_me_small or gop_1080p_vs_oracle"),
                       ({"VCS_ME16_BITS": "2", "VCS_L3_PREFETCH": "0"}, "gop_1080p_vs_oracle or gop_matches_single"),
                       ({"VCS_ME16_BITS": "0", "VCS_ME16_ROLL": "2"}, sel),
                       ({"VCS_ME16_BITS": "0", "VCS_ME16_ROLL": "2", "VCS_ME16_GROUP": "0", "VCS_ME8_GROUP": "0"}, sel_small),
                       ({"VCS_ME16_BITS": "0", "VCS_ME16_ROLL": "2", "VCS_ME16_GROUP": "2", "VCS_ME16_WPS": "3"}, sel_small),
                       ({"VCS_ME16_BITS": "0", "VCS_ME16_ROLL": "0"}, sel_small),
                       ({"VCS_ME_GENERIC": "1"}, sel_small), ({"VCS_FAST_DIV": "0"}, dct),
                       # the tail kernels' other strip -> workgroup mappings (4 strips per workgroup)
                       ({"VCS_TAIL_MAP": "1"}, dct + " or gop_1080p_vs_oracle"),
                       ({"VCS_TAIL_MAP": "2"}, dct + " or gop_1080p_vs_oracle"),
                       # Infinity Cache prefetch as a separate pass / off (default: inside the search kernel)
                       ({"VCS_ME16_BITS": "0", "VCS_ME16_ROLL": "2", "VCS_L3_PREFETCH": "2"}, "gop_1080p_vs_oracle or gop_matches_single"),
                       ({"VCS_ME16_BITS": "0", "VCS_ME16_ROLL": "2", "VCS_L3_PREFETCH": "0"}, "gop_1080p_vs_oracle or gop_matches_single"),
                       # the bit-plane search's scratch cannot be allocated: the shared-window kernels serve the batch
                       ({"VCS_TEST_BITS_NOMEM": "1"}, "bitplane or gop_1080p_vs_oracle or golden_me_small or next_reference"))

    def child(v):
        env_add, k = v
        env = dict(os.environ, VCS_ORACLE_MEMO=oracle._root, **env_add)   # children reuse this session's oracle results
        return env_add, subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"),
                                        "-m", "gpu", "-x", "-q", "-k", k, "-p", "no:cacheprovider"],
                                       env=env, capture_output=True, text=True, timeout=900)
    # One child at a time: with the oracle's results memoised on disk a child is mostly start-up, and a fault
    # or a hang is then one variant's, by name.  VCS_TEST_PARALLEL=n runs n at a time (a box allows six
    # processes on the card; this one is the first).
    workers = max(1, min(4, int(os.environ.get("VCS_TEST_PARALLEL", "1"))))
    if workers == 1:
        for v in variants:
            env_add, r = child(v)
            assert r.returncode == 0, (env_add, r.stdout[-1500:], r.stderr[-500:])
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as pool:
            for env_add, r in pool.map(child, variants):
                assert r.returncode == 0, (env_add, r.stdout[-1500:], r.stderr[-500:])


@pytest.mark.parametrize("env", [{"VCS_ME16_BITS": "2", "VCS_ME8_BITS": "2"}, {}, {"VCS_ME16_BITS": "0", "VCS_ME8_BITS": "0"}])
def test_randomised_batches_vs_oracle(env):
    """tools/gpu_fuzz.py: 150 random (size, block size, batch size, threshold, content) cases per kernel
    selection -- the bit-plane search on every launch, the default selection, the round-2 kernels --
    vectors and static counts of every frame, one frame's planes in a quarter of the cases."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_fuzz.py"), "--cases", "150", "--seed", "7"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (env, r.stdout[-2000:], r.stderr[-500:])
