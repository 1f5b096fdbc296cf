"""BASELINE.json's configs exercised at their OWN shape on the GPU (through the
drop-in classes / the C ABI), every result compared with the CPU oracle:

  C1'  main.py's own settings (main.py:13-16): 640x360, ENCODING_PATTERN of 4,
       BLOCK_SIZE 8, with_DCT -- driven exactly as main.py:29-50 drives the
       reference (synthetic frames stand in for videos/traffic_cut.mp4: no
       video decoder exists in the build environment);
  C3   1080p 30-frame GOP (29 P frames in one vcs_gop_encode), QF 10 / 50 / 99;
  C4   the per-rank share of a 3840x2160 30-frame GOP sharded over 8 GPUs
       (rank 0 owns 4 P frames), through the same GOP entry points.

Plus the regressions behind ADVICE.md (round 1): parameters pushed by one drop-in
class survive the other classes' pushes; an I frame edited in place is seen.
"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vh():
    import vcs_hip
    vcs_hip.load()
    assert vcs_hip.device_count() >= 1, "no GPU visible to libvcs_hip"
    return vcs_hip


def _moving_sequence(H, W, n, seed):
    """n frames: a textured scene panning a little every frame, with noise; every 4th
    frame (the I frames of main.py's pattern) is a fresh capture, not a copy."""
    from synth import synth_ref
    base, rng = synth_ref(H + 64, W + 64, seed)
    frames = np.empty((n, H, W, 3), np.uint8)
    for k in range(n):
        oy, ox = 32 + (k * 3) % 17 - 8, 32 + (k * 5) % 23 - 11
        f = base[oy:oy + H, ox:ox + W].astype(np.int16)
        f += rng.integers(-2, 3, (H, W, 3), dtype=np.int16)
        frames[k] = np.clip(f, 0, 255).astype(np.uint8)
    return frames


def test_config1_main_py_settings_dropin(vh, oracle, capsys):
    """C1': 640x360, pattern I P P P, block_size 8, with_DCT, 44 frames, the call
    sequence of main.py:29-50 on the drop-in Encoder / Decoder."""
    import encoder as E
    import decoder as D
    H, W, BLOCK_SIZE, n_frames = 360, 640, 8, 44
    ENCODING_PATTERN = ["I", "P", "P", "P"]      # main.py:16
    frames = _moving_sequence(H, W, n_frames, seed=2022)
    enc = E.Encoder(pattern=ENCODING_PATTERN, shape=[H, W], block_size=BLOCK_SIZE, with_DCT=True)
    for n in range(n_frames):                    # main.py:34-43
        enc.encode_frame(frames[n], n)
    assert len(enc.encoded_frames) == n_frames and len(enc.ref_frames) == n_frames // 4
    sink = []
    dec = D.Decoder(encoded_frames=enc.encoded_frames, fps=29.97, shape=[H, W],
                    ref_frames=enc.ref_frames, block_size=BLOCK_SIZE, with_DCT=True,
                    sink=sink.append)
    dec.reconstruct_video(with_residuals=True)   # main.py:50
    assert len(sink) == n_frames
    n_static_total = 0
    for n, fr in enumerate(enc.encoded_frames):
        if n % 4 == 0:
            assert fr.t == "I" and np.array_equal(sink[n], frames[n])
            continue
        ref = frames[n - n % 4]
        assert fr.t == "P" and fr.ref_i == n // 4
        omv, oplanes, ons = oracle.encode_p(frames[n], ref, BLOCK_SIZE)
        assert np.array_equal(np.array(fr.mv), omv), n
        assert np.array_equal(np.array(fr.c), oracle.block_coords(H, W, BLOCK_SIZE)), n
        assert np.array_equal(np.stack(fr.r), oplanes), n
        assert np.array_equal(sink[n], oracle.decode_p(omv, oplanes, ref, BLOCK_SIZE)), n
        n_static_total += ons
    assert n_static_total > 0  # the static early-out was exercised too
    capsys.readouterr()


def test_config3_1080p_full_gop_qf_sweep(vh, oracle):
    """C3: 29 P frames of a 1080p GOP in one vcs_gop_encode at QF 10 / 50 / 99.
    Vectors of every frame, planes + decoded pixels of 3 sampled frames per QF."""
    from synth import synth_gop
    H, W, n = 1080, 1920, 29
    ref, frames = synth_gop(H, W, n, seed=1234)
    omv = [oracle.me(frames[k], ref, 16) for k in range(n)]  # QF does not enter the search
    samples = {10.0: (0, 13, 28), 50.0: (1, 14, 27), 99.0: (2, 15, 26)}
    with vh.Context(H, W, 16) as c:
        c.set_ref(ref)
        c.gop_reserve(n, with_decode=True)
        c.gop_upload(0, frames)
        for qf, ks in samples.items():
            Q = oracle.make_q(qf)
            assert np.array_equal(vh.make_q(qf), Q)
            c.set_params(Q=Q)
            c.gop_encode(0, n)
            c.gop_decode(0, n)
            mv, ns = c.gop_get_mv(0, n)
            for k in range(n):
                assert np.array_equal(mv[k], omv[k][0]) and ns[k] == omv[k][1], (qf, k)
            for k in ks:
                _, oplanes, _ = oracle.encode_p(frames[k], ref, 16, Q=Q)
                planes = c.gop_get_planes(k)
                assert np.array_equal(planes.view(np.int64), oplanes.view(np.int64)), (qf, k)
                assert np.array_equal(c.gop_get_decoded(k),
                                      oracle.decode_p(mv[k], oplanes, ref, 16, Q=Q)), (qf, k)


def test_config4_4k_per_rank_gop_share(vh, oracle):
    """C4: 3840x2160, 29 P frames over 8 GPUs = 4 frames on rank 0 (gop.shard_frames),
    encoded as one GOP batch; vectors, planes and decoded pixels vs the oracle."""
    from gop import shard_frames
    from synth import synth_gop
    H, W = 2160, 3840
    first, count = shard_frames(29, 8, 0)
    assert (first, count) == (0, 4)
    ref, frames = synth_gop(H, W, count, seed=4321, first=1 + first)
    with vh.Context(H, W, 16) as c:
        c.set_ref(ref)
        c.gop_reserve(count, with_decode=True)
        c.gop_upload(0, frames)
        c.gop_encode(0, count)
        c.gop_decode(0, count)
        mv, ns = c.gop_get_mv(0, count)
        for k in range(count):
            omv, oplanes, ons = oracle.encode_p(frames[k], ref, 16)
            assert np.array_equal(mv[k], omv) and ns[k] == ons, k
            assert np.array_equal(c.gop_get_planes(k).view(np.int64), oplanes.view(np.int64)), k
            assert np.array_equal(c.gop_get_decoded(k), oracle.decode_p(omv, oplanes, ref, 16)), k


def test_config4_4k_two_gpu_share_bitplane_search(vh, oracle):
    """C4 on two GPUs: 15 of the 29 P frames of a 3840x2160 GOP on rank 0 -- enough frames for the
    bit-plane search (one 497 MB chunk table at this width: kBitsBandBytes = 1 GiB; the banded form is
    covered by the VCS_BITS_BAND_KB child of test_forced_kernel_variants_subprocess on small frames and by
    test_banded_chunk_table_on_a_real_4k_frame below), while the tail runs in cache-sized sub-batches;
    vectors of every frame, planes of three, vs the oracle."""
    from gop import shard_frames
    from synth import synth_gop
    H, W = 2160, 3840
    first, count = shard_frames(29, 2, 0)
    assert (first, count) == (0, 15)
    ref, frames = synth_gop(H, W, count, seed=99, first=1 + first)
    frames[5][:, : W // 3] = ref[:, : W // 3]          # a static region
    with vh.Context(H, W, 16) as c:
        c.set_ref(ref)
        c.gop_reserve(count)
        c.gop_upload(0, frames)
        c.gop_encode(0, count)
        mv, ns = c.gop_get_mv(0, count)
        for k in range(count):
            omv, ons = oracle.me(frames[k], ref, 16)
            assert np.array_equal(mv[k], omv) and ns[k] == ons, k
        for k in (0, 7, 14):
            _, oplanes, _ = oracle.encode_p(frames[k], ref, 16)
            assert np.array_equal(c.gop_get_planes(k).view(np.int64), oplanes.view(np.int64)), k


# ------------------------------------------------------------ ADVICE (round 1) ----
def test_banded_chunk_table_on_a_real_4k_frame(oracle):
    """The same 4K share with the chunk table forced into four bands of ~60 block columns (VCS_BITS_BAND_KB =
    128 MB; per launch: k_bits_cur once, then k_bits_ref + k_me16b per band, the prefetch on the last band) --
    the path frames beyond 1 GiB of table take.  One child process (the band size is read once per process)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, VCS_ORACLE_MEMO=oracle._root, VCS_ME16_BITS="2", VCS_BITS_BAND_KB=str(128 * 1024))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_configs.py"), "-m", "gpu", "-x", "-q",
                        "-k", "config4_4k_two_gpu_share", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-500:])


def test_similarity_threshold_survives_q_pushes(vh, oracle, capsys):
    """motion.SIMILARITY_THRESHOLD edited by hand (as the reference expects) must
    survive the Q / coeff_mode pushes of Encoder, Decoder and DCTCompressor on the
    shared context: vcs_set_params keeps the threshold unless told otherwise."""
    import DCTcompressor as C
    import encoder as E
    import motion as M
    from _runtime import release_contexts
    from synth import synth_static
    H, W = 144, 176
    ref, cur = synth_static(H, W, seed=5, bs=8, frac=0.6)
    release_contexts()
    saved = M.SIMILARITY_THRESHOLD
    try:
        for thr in (0, 700):
            M.SIMILARITY_THRESHOLD = thr
            enc = E.Encoder(pattern=["I", "P", "P"], shape=[H, W], block_size=8, with_DCT=True)
            enc.encode_frame(ref, 0)
            enc.encode_frame(cur, 1)                      # pushes ME params, then Q
            C.DCTCompressor(8, qf=35).compress(cur)       # another Q push on the shared context
            enc.encode_frame(cur, 2)                      # ME params unchanged on the host side
            mp = M.MotionProcessor(8, [H, W])
            mv, _ = mp.process_motion_prediction(cur, ref)
            omv, ons = oracle.me(cur, ref, 8, thresh=thr)
            assert np.array_equal(np.array(mv), omv) and mp.last_static_blocks == ons
            for fr in enc.encoded_frames[1:]:
                assert np.array_equal(np.array(fr.mv), omv), thr
        base = oracle.me(cur, ref, 8)[1]
        assert oracle.me(cur, ref, 8, thresh=0)[1] < base  # the thresholds really differ
    finally:
        M.SIMILARITY_THRESHOLD = saved
        release_contexts()
    # the C ABI itself: KEEP sentinels
    with vh.Context(H, W, 8) as c:
        c.set_ref(ref)
        c.set_params(sim_thresh=-1)
        c.set_params(Q=oracle.make_q(20.0))               # must not bring 2000 back
        assert c.me(cur)[1] == 0
        c.set_params(sim_thresh=2000)
        assert c.me(cur)[1] == oracle.me(cur, ref, 8)[1]
    capsys.readouterr()


def test_in_place_edit_of_the_i_frame_is_seen(vh, oracle, capsys):
    """The reference aliases the caller's I frame (encoder.py:42): a pixel edited in
    place between two P frames changes the second one's reference.  One byte, at a
    position the old 1-in-61 sampling never looked at."""
    import encoder as E
    from _runtime import release_contexts
    from synth import synth_v1
    H, W = 144, 176
    ref, cur = synth_v1(H, W, seed=31)
    ref = ref.copy()
    release_contexts()
    for resident in (False, True):
        iframe = ref.copy()
        enc = E.Encoder(pattern=["I", "P", "P"], shape=[H, W], block_size=16, with_DCT=True,
                        dct_block_size=8, resident=resident)
        enc.encode_frame(iframe, 0)
        enc.encode_frame(cur, 1)
        iframe.reshape(-1)[1] ^= 0xFF                       # byte 1: 1 % 61 != 0
        iframe[40:56, 60:76] = 255 - iframe[40:56, 60:76]   # and a visibly different block
        if resident:
            enc.invalidate_ref()                            # documented contract of the fast mode
        enc.encode_frame(cur, 2)
        o1 = oracle.encode_p(cur, ref, 16)
        o2 = oracle.encode_p(cur, iframe, 16)
        assert not np.array_equal(o1[1], o2[1])
        f1, f2 = enc.encoded_frames[1], enc.encoded_frames[2]
        assert np.array_equal(np.asarray(f1.mv), o1[0]) and np.array_equal(np.stack(list(f1.r)), o1[1])
        assert np.array_equal(np.asarray(f2.mv), o2[0]) and np.array_equal(np.stack(list(f2.r)), o2[1])
    release_contexts()
    capsys.readouterr()


def test_resident_planes_outlive_their_context(vh, oracle):
    """ResidentPlanes freed after the context was closed: vcs_store_free(NULL, h)."""
    from synth import synth_v1
    ref, cur = synth_v1(64, 96, seed=3)
    c = vh.Context(64, 96, 16)
    c.set_ref(ref)
    mv, planes, ns = c.store_encode_p(cur)
    host = np.stack(list(planes))
    assert np.array_equal(host, oracle.encode_p(cur, ref, 16)[1])
    mv2, planes2, _ = c.store_encode_p(cur)
    planes2.spill()                                         # download + free, view keeps working
    assert not planes2.resident and np.array_equal(np.stack(list(planes2)), host)
    c.close()
    planes.release()                                        # no error, no leak
    assert planes._h is None


# ------------------------------------------------------- extension searches (N4) ----
@pytest.mark.parametrize("search,cost", [("grid", "sad"), ("tss", "wrap"), ("tss", "sad")])
def test_extension_searches_vs_oracle(vh, oracle, search, cost, capsys):
    """vcs_set_search: three-step search and true-SAD cost (NOT in the reference; the oracle is
    oracle/vcs_oracle.c: vcso_me_ext -- parity unpinned).  Vectors, static counts and winning
    costs, over block sizes, ragged frames, frame edges, window overrides; the reference's own
    mode is untouched by a round trip through the extension."""
    from synth import synth_static, synth_v1
    cases = [(144, 176, 16, None), (136, 200, 8, None), (50, 70, 4, None), (64, 64, 16, None),
             (96, 128, 16, 7), (72, 88, 8, 31), (40, 56, 6, None)]
    for H, W, bs, win in cases:
        ref, cur = synth_v1(H, W, seed=H + bs, shift=(5, -9))
        with vh.Context(H, W, bs) as c:
            c.set_ref(ref)
            if win:
                c.set_params(win=win)
            c.set_search(search, cost)
            mv, ns, co = c.me(cur, want_cost=True)
            omv, ons, oco = oracle.me_ext(cur, ref, bs, search, cost, win=win, want_cost=True)
            assert np.array_equal(mv, omv) and ns == ons, (H, W, bs, win)
            assert np.array_equal(co, oco), (H, W, bs, win)
            # the fused encode path uses the same vectors
            if H % 8 == 0 and W % 8 == 0:
                mv2, planes, _, _ = c.encode_p(cur)
                assert np.array_equal(mv2, omv)
            c.set_search("grid", "wrap")       # back to the reference's search
            mv, ns = c.me(cur)
            omv, ons = oracle.me(cur, ref, bs, win=win)
            assert np.array_equal(mv, omv) and ns == ons
    # static-heavy content and a full 1080p frame
    ref, cur = synth_static(1080, 1920, seed=2, bs=16, frac=0.5)
    with vh.Context(1080, 1920, 16) as c:
        c.set_ref(ref)
        c.set_search(search, cost)
        mv, ns = c.me(cur)
        omv, ons = oracle.me_ext(cur, ref, 16, search, cost)
        assert np.array_equal(mv, omv) and ns == ons and ns > 0
    # through the drop-in class: module constants, as the reference's other knobs
    import motion as M
    from _runtime import release_contexts
    release_contexts()
    ref, cur = synth_v1(144, 176, seed=3, shift=(7, -11))
    M.SEARCH, M.COST = search, cost
    try:
        mv, _ = M.MotionProcessor(16, [144, 176]).process_motion_prediction(cur, ref)
        assert np.array_equal(np.array(mv), oracle.me_ext(cur, ref, 16, search, cost)[0])
    finally:
        M.SEARCH, M.COST = "grid", "wrap"
        release_contexts()
    if search == "tss" and cost == "sad":
        # what the extension is for: the true displacement of a globally shifted frame, (11, -7),
        # is not on the reference's stride-5 grid at all; the three-step search lands on it for a
        # good part of the blocks (it is a greedy search: not for all of them)
        hit = float((np.array(mv) == [11, -7]).all(axis=1).mean())
        assert hit > 0.25, hit
    capsys.readouterr()


def test_chroma420_vs_oracle(vh, oracle):
    """4:2:0 chroma subsampling (chroma.py:8-21; next row N4): kernel vs the oracle's
    restatement of cvtColor + cv2.boxFilter(ksize=(2,2)) + [::2, ::2] (parity unpinned: OpenCV
    is not available), even / odd sizes, extremes, 1080p, and the drop-in function."""
    rng = np.random.default_rng(7)
    for H, W in ((8, 8), (7, 10), (33, 17), (64, 96), (1, 5), (6, 1), (1080, 1920)):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        if H >= 8:
            img[:4] = 255
            img[-2:] = 0
        with vh.Context(H, W, 8) as c:
            Y, cr, cb = c.chroma420(img)
            oY, ocr, ocb = oracle.chroma420(img)
            assert np.array_equal(Y, oY) and np.array_equal(cr, ocr) and np.array_equal(cb, ocb), (H, W)
            assert cr.shape == ((H + 1) // 2, (W + 1) // 2)
            _, cr2, _ = c.chroma420(img, want_y=False)
            assert np.array_equal(cr2, ocr)
    sys_path = __import__("sys").path
    cdir = __import__("os").path.join(__import__("conftest").PKG, "ChromaSubsampling")
    if cdir not in sys_path:
        sys_path.insert(0, cdir)
    import chroma as CH
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    out = CH.subsample420(img)
    want = oracle.chroma420(img)
    assert len(out) == 3 and all(np.array_equal(a, b) for a, b in zip(out, want))
    with pytest.raises(TypeError):
        CH.subsample420(img.astype(np.float32))


def test_config2_1080p_pair_reference_and_dense_search(vh, oracle):
    """C2: a synthetic 1080p I+P pair, 8x8 blocks, +-16 px.  The reference's own search at that
    block size is the stride-3 grid (motion.py:132: round(8/3)); BASELINE's wording "full search"
    is the dense stride-1 grid over the same window -- both against the oracle, whole frame:
    vectors, static count, winning costs, then the fused encode and decode on the dense vectors."""
    from synth import synth_v1
    H, W, bs = 1080, 1920, 8
    ref, cur = synth_v1(H, W, seed=1234)
    with vh.Context(H, W, bs) as c:
        c.set_ref(ref)
        for step in (3, 1):
            c.set_params(step=step)
            mv, ns, cost = c.me(cur, want_cost=True)
            omv, ons, ocost = oracle.me(cur, ref, bs, step=step, want_cost=True)
            assert np.array_equal(mv, omv) and ns == ons, step
            assert np.array_equal(cost, ocost), step
        # dense vectors reach every displacement in [-16, 7]: more than the 8 x 8 grid positions
        assert len(np.unique(mv, axis=0)) > 64
        mv2, planes, _, _ = c.encode_p(cur)
        assert np.array_equal(mv2, omv)
        _, oplanes, _ = oracle.encode_p(cur, ref, bs, step=1)
        assert np.array_equal(planes.view(np.int64), oplanes.view(np.int64))
        assert np.array_equal(c.decode_p(mv2, planes), oracle.decode_p(omv, oplanes, ref, bs))


@pytest.mark.gpu
@pytest.mark.parametrize("bs", [16, 8])
def test_gop_in_cache_sized_sub_batches(bs):
    """vcs_gop_encode splits a GOP whose frames do not fit the Infinity Cache into sub-batches
    (search + prefetch + tail each).  Forced here with a tiny cache budget (the library reads
    VCS_L3_PREFETCH_MAX_KB once, hence the child process): 11 frames in batches of 4, 4, 3 must
    give what frame-by-frame calls and the oracle give, for the shared-window search kernels
    (VCS_ME16_ROLL=2) with their in-kernel prefetch and for the separate prefetch pass."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = r'''
import os, sys
sys.path[:0] = [%r, %r]
import numpy as np
import vcs_hip
from oracle import oracle
from synth import synth_gop
bs = %d
H, W, n = 144, 176, 11
ref, frames = synth_gop(H, W, n, seed=21)
with vcs_hip.Context(H, W, bs, 8) as c:
    c.set_ref(ref)
    c.gop_reserve(n)
    c.gop_upload(0, frames)
    c.gop_encode(0, n)
    mv, ns = c.gop_get_mv(0, n)
    for k in range(n):
        omv, oplanes, _ = oracle.encode_p(frames[k], ref, bs)
        assert np.array_equal(mv[k], omv), k
        assert np.array_equal(c.gop_get_planes(k).view(np.int64), oplanes.view(np.int64)), k
print("ok")
''' % (os.path.dirname(here), os.path.join(os.path.dirname(here), "vcs-h264_amd"), bs)
    frame_kb = (144 * 176 * 3 + 64 + 255) // 256 * 256 / 1024.0
    for mode in ("1", "2"):
        env = dict(os.environ, VCS_L3_PREFETCH=mode, VCS_ME16_ROLL="2",
                   VCS_L3_PREFETCH_MAX_KB=str(int(4 * frame_kb) + 1))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, (mode, r.stdout[-500:], r.stderr[-2000:])
