"""Host-side checks that need no GPU: the C ABI library loads and exports every
symbol include/vcs_hip.h declares, host-only entry points agree with the golden
vectors, the drop-in modules mirror the reference's interface, and the product
path fails loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO, load_golden


@pytest.fixture(scope="module")
def vh():
    import vcs_hip
    if not os.path.exists(vcs_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    vcs_hip.load()
    return vcs_hip


def header_symbols():
    src = open(os.path.join(REPO, "include", "vcs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vcs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(vh):
    syms = header_symbols()
    assert len(syms) >= 30
    lib = ctypes.CDLL(vh.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "libvcs_hip.so does not export %s" % s
        assert s in vh.SIGNATURES, "vcs_hip.py has no prototype for %s" % s
    assert sorted(vh.SIGNATURES) == syms
    assert vh.load().vcs_abi_version() == 2


def test_library_is_the_tree_and_no_experiment_build(vh):
    """The shipped library was built from the sources in the tree (vcs_build_id = hash of csrc/ + flags: what
    bench.py ties profiles/hbm_traffic.json to) and without any VCS_EXP_* experiment switch; every such switch
    in csrc/ is fenced by the umbrella macro in vcs_common.hpp."""
    lib = vh.load()
    assert lib.vcs_experiment_build() == 0
    assert vh.build_id() == vh.source_build_id(), "libvcs_hip.so is older than csrc/: rebuild (python __graft_entry__.py build)"
    csrc = os.path.join(REPO, "vcs-h264_amd", "csrc")
    common = open(os.path.join(csrc, "vcs_common.hpp")).read()
    fenced = set(re.findall(r"defined\((VCS_EXP_[A-Z0-9_]+)\)", common))
    used = set()
    for f in os.listdir(csrc):
        if f.endswith((".hpp", ".hip")):
            for m in re.finditer(r"^\s*#\s*(?:ifdef|ifndef|if|elif)\b.*?\b(VCS_EXP_[A-Z0-9_]+)", open(os.path.join(csrc, f)).read(), re.M):
                used.add(m.group(1))
    assert used <= fenced, "experiment switches outside the umbrella: %s" % sorted(used - fenced)


def test_host_only_entry_points_match_golden(vh):
    g = load_golden("dct_small.npz")
    assert np.array_equal(vh.dct_matrix(), g["dctmat"])  # DCTcompressor.py:124-133
    for qf in (10, 50, 99):
        assert np.array_equal(vh.make_q(float(qf)), g["Q_q%d" % qf])  # :29-38
    with pytest.raises(ValueError):
        vh.make_q(100.0)
    with pytest.raises(ValueError):
        vh.make_q(0.0) if False else vh.make_q(150.0)


def test_no_cpu_fallback(vh):
    if vh.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(vh.VcsError) as e:
        vh.Context(64, 64, 16)
    assert "no usable HIP device" in str(e.value) and "no CPU fallback" in str(e.value)
    # nothing in the product tree may import the oracle
    pkg = os.path.join(REPO, "vcs-h264_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "liboracle" not in txt and "from oracle" not in txt and \
                    "import oracle" not in txt, f


def test_dropin_interface_mirrors_reference(vh, capsys):
    import DCTcompressor as C
    import decoder as D
    import encoder as E
    import frame as F
    import motion as M
    # module-level names the reference exposes (motion.py:8-11, DCTcompressor.py:7-38)
    assert M.SIMILARITY_THRESHOLD == 2000 and M.CHANNELS == 3 and M.WRITE_STATIC_BLOCK is True
    assert C.QF == 50.0 and C.QY.shape == (8, 8) and C.QC.shape == (8, 8) and len(C.Q) == 3
    assert C.QY[1, 5] == 48 and C.QC[0, 0] == 17
    mp = M.MotionProcessor(16, [48, 64])
    assert (mp.block_size, mp.search_window_size, mp.shape) == (16, 32, [48, 64])
    for name in ("process_motion_prediction", "get_residuals", "reconstruct_from_motion_vectors"):
        assert callable(getattr(mp, name))
    c = C.DCTCompressor(8)
    assert c.blocksize == 8 and c.compressed == [] and c.Q is C.Q
    f = F.Frame("P", motion_vectors=[[0, 0]], residuals=None, block_coords=[[0, 0]], index=3,
                ref_idx=1)
    assert (f.t, f.mv, f.r, f.c, f.i, f.ref_i) == ("P", [[0, 0]], None, [[0, 0]], 3, 1)
    # block grid: raster order, [x, y] entries, full blocks only (motion.py:74-98)
    co = M.MotionProcessor(16, [50, 70])._block_coords().tolist()
    assert co == [[0, 0], [16, 0], [32, 0], [48, 0], [0, 16], [16, 16], [32, 16], [48, 16],
                  [0, 32], [16, 32], [32, 32], [48, 32]]
    # I frames never touch the GPU: they alias the caller's array (encoder.py:41-43)
    enc = E.Encoder(pattern=["I", "P", "P", "P"], shape=[48, 64], block_size=8, with_DCT=True)
    img = np.zeros((48, 64, 3), np.uint8)
    enc.encode_frame(img, 0)
    enc.encode_frame(img, 4)
    assert [fr.t for fr in enc.encoded_frames] == ["I", "I"] and enc.ref_frames[0] is img
    assert enc.ENCODING_PATTERN_LENGTH == 4
    if vh.device_count() == 0:  # a P frame needs the GPU and must say so
        with pytest.raises(vh.VcsError):
            enc.encode_frame(img, 1)
    dec = D.Decoder(encoded_frames=enc.encoded_frames, fps=25.0, shape=[48, 64],
                    ref_frames=enc.ref_frames, block_size=8, with_DCT=True, sink=[].append)
    dec.reconstruct_video(with_residuals=True)
    with pytest.raises(ValueError):  # block_size != 8 with DCT fails as in the reference
        C.DCTCompressor(16).compress(np.zeros((48, 64, 3), np.uint8))
    with pytest.raises(NotImplementedError):
        C.DCTCompressor(8).compress(np.zeros((50, 70, 3), np.uint8))
    capsys.readouterr()


def test_synth_is_deterministic():
    from synth import synth_gop, synth_static, synth_v1
    a = synth_v1(48, 64, seed=5)
    b = synth_v1(48, 64, seed=5)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    ref, fr = synth_gop(48, 64, 4, seed=5)
    ref2, fr2 = synth_gop(48, 64, 2, seed=5, first=3)
    assert np.array_equal(ref, ref2) and np.array_equal(fr[2:], fr2)
    r, c = synth_static(64, 64, seed=1, bs=16, frac=0.5)
    assert r.shape == c.shape == (64, 64, 3)


def test_rowlist_reads_like_a_list_of_pairs():
    """frame.RowList (Encoder(resident=True) stores Frame.mv / Frame.c in it): the accesses
    the reference makes on those lists -- len, indexing, iteration, np.array -- and a few
    more, without N Python lists behind it."""
    import sys
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vcs-h264_amd",
                     "InterframeCompression")
    if d not in sys.path:
        sys.path.insert(0, d)
    from frame import RowList
    a = np.array([[1, -2], [3, 4], [-5, 6], [7, 8]], np.int64)
    v = RowList(a)
    assert len(v) == 4 and v[2] == [-5, 6] and v[-1] == [7, 8] and v[1:3] == [[3, 4], [-5, 6]]
    assert type(v[0][0]) is int and list(v) == a.tolist() and v.tolist() == a.tolist()
    assert v == a.tolist() and v == RowList(a.copy()) and not (v == a[:3].tolist())
    assert np.asarray(v) is a and np.array_equal(np.array(v, np.int16), a)
    for (dx, dy), (x, y) in zip(v, RowList(a + 1)):   # decoder.py:62-65 style consumption
        assert (x, y) == (dx + 1, dy + 1)
    with pytest.raises(IndexError):
        v[4]
    with pytest.raises(ValueError):
        RowList(np.zeros((3, 3)))


def test_shard_table_of_the_library_matches_the_contract(vh):
    """vcs_shard_frames (the library's own dealing of P frames to GPUs): contiguous,
    sizes differ by at most one, earlier shards take the larger chunks."""
    for n in (0, 1, 5, 29, 30, 64, 232):
        for world in (1, 2, 3, 4, 8):
            pos = 0
            counts = []
            for r in range(world):
                first, count = vh.shard_frames(n, world, r)
                assert first == pos
                pos += count
                counts.append(count)
            assert pos == n and max(counts) - min(counts) <= 1
            assert counts == sorted(counts, reverse=True)
    assert [vh.shard_frames(29, 8, r)[1] for r in range(8)] == [4, 4, 4, 4, 4, 3, 3, 3]   # config 4
    with pytest.raises(ValueError):
        vh.shard_frames(5, 2, 2)


def test_rendezvous_hands_rank0s_id_to_every_rank(vh, tmp_path):
    """The file rendezvous bench.py uses under torch.distributed.run (no PyTorch in the
    ranks): three processes, rank 0's 128 bytes reach the others; a stale file of another
    launch (different tag) is not picked up."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); import vcs_hip\n"
            "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
            "u = vcs_hip.rendezvous_unique_id(r, w, timeout_s=60, make_id=lambda: bytes([7 + r]) * 128)\n"
            "open(os.path.join(%r, 'id%%d' %% r), 'wb').write(u)\n" % (os.path.join(REPO, "vcs-h264_amd"), str(tmp_path)))
    tag = "test_%d" % os.getpid()
    procs = []
    for r in (2, 1, 0):   # rank 0 last: the others must wait for it
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", VCS_RENDEZVOUS_TAG=tag)
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env))
    assert all(p.wait(timeout=120) == 0 for p in procs)
    ids = [open(tmp_path / ("id%d" % r), "rb").read() for r in range(3)]
    assert ids[0] == bytes([7]) * 128 and ids[1] == ids[0] and ids[2] == ids[0]
    os.environ["VCS_RENDEZVOUS_TAG"] = tag
    try:
        vh.rendezvous_cleanup(0, 3)
        with pytest.raises(TimeoutError):
            vh.rendezvous_unique_id(1, 3, timeout_s=0.3)
    finally:
        del os.environ["VCS_RENDEZVOUS_TAG"]


def test_reference_structured_loop_equals_the_oracle():
    """oracle/ref_loop.py (the slow, structure-faithful CPU baseline) computes what the C
    oracle computes."""
    from oracle import oracle, ref_loop
    from synth import synth_static
    ref, cur = synth_static(64, 96, seed=9, bs=16, frac=0.4)
    for bs in (8, 16):
        mv, ns = ref_loop.me(cur, ref, bs)
        omv, ons = oracle.me(cur, ref, bs)
        assert np.array_equal(mv, omv) and ns == ons
    Q = oracle.make_q(50.0)
    planes, ycc = oracle.compress(cur, Q, want_ycc=True)
    for ch in range(3):
        got = ref_loop.compress_plane(ycc[..., ch].astype(np.int16) - 128, Q[ch])
        assert np.max(np.abs(got - planes[ch])) < 1e-9   # NumPy's matmul order, not the oracle's fma chain


def test_intramodes_dropin_matches_the_reference_functions():
    """IntraframeCompression/intramodes.py (the fifteen predictors as importable functions,
    intraframe.py:7 upstream does `from intramodes import *`) against outputs of the
    reference's own functions on recorded inputs (oracle/gen_golden.py: g8)."""
    import importlib.util
    import warnings
    spec = importlib.util.spec_from_file_location(
        "dropin_intramodes", os.path.join(REPO, "vcs-h264_amd", "IntraframeCompression", "intramodes.py"))
    im = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(im)
    g = load_golden("intramodes_small.npz")
    u, ur, le, ul = g["u"], g["ur"], g["l"], g["ul"]
    args = {"vertical4x4": lambda k: (u[k],), "horizontal4x4": lambda k: (le[k],),
            "dc4x4": lambda k: (u[k], le[k]), "downleft4x4": lambda k: (u[k], ur[k]),
            "downright4x4": lambda k: (ul[k], u[k], le[k]), "verticalright4x4": lambda k: (ul[k], u[k], le[k]),
            "horizontaldown4x4": lambda k: (ul[k], u[k], le[k]), "verticalleft4x4": lambda k: (u[k], ur[k]),
            "horizontalup4x4": lambda k: (le[k],), "vertical16x16": lambda k: (g["u16"][k],),
            "horizontal16x16": lambda k: (g["l16"][k],), "dc16x16": lambda k: (g["u16"][k], g["l16"][k]),
            "vertical8x8": lambda k: (g["u8"][k],), "horizontal8x8": lambda k: (g["l8"][k],),
            "dc8x8": lambda k: (g["u8"][k], g["l8"][k])}
    assert len(args) == 15
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")   # NumPy's uint8 scalar overflow warnings, as upstream
        for name, a in args.items():
            want = g["out_" + name]
            for k in range(len(u)):
                got = getattr(im, name)(*a(k))
                assert got.dtype == np.float64 and np.array_equal(got, want[k]), (name, k)


def test_smem_pipeline_checker_on_synthetic_assembly(tmp_path):
    """tools/check_smem_pipeline.py (part of the library's build): a copy of a register whose hand-issued scalar
    load is still in flight is found along every path of the control-flow graph -- also around a loop's back
    edge and past a block that only sits between the two in the text --, a wait clears the set, and a kernel
    without hand-issued loads is an error (the check would be vacuous)."""
    import subprocess
    import sys
    tool = os.path.join(REPO, "tools", "check_smem_pipeline.py")
    head = "_ZN3vcs7k_demoEv:\n"
    issue = "\t;;#ASMSTART\n\ts_load_dwordx8 s[8:15], s[2:3], s4\n\t;;#ASMEND\n"
    wait = "\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n"
    tail = "\ts_endpgm\n.Lfunc_end0:\n"
    cases = {
        # straight line: request, wait, use
        "ok": head + issue + "\ts_add_u32 s20, s21, s22\n" + wait + "\tv_mov_b32_e32 v0, s8\n" + tail,
        # a copy of the in-flight tuple before the wait
        "copy": head + issue + "\ts_mov_b64 s[30:31], s[8:9]\n" + wait + tail,
        # in flight around a loop's back edge: the use at the loop's head comes before the wait
        "loop": head + issue + ".LBB0_1:\n\tv_mov_b32_e32 v1, s9\n" + wait + issue + "\ts_cbranch_scc1 .LBB0_1\n" + wait + tail,
        # a block that is skipped over (s_branch) while the load is in flight may use the registers' numbers
        # for nothing: it is not reachable with them in flight
        "skipped": head + "\ts_cbranch_scc1 .LBB0_2\n" + issue + "\ts_branch .LBB0_3\n.LBB0_2:\n\ts_mov_b32 s8, 0\n"
                   "\ts_branch .LBB0_4\n.LBB0_3:\n" + wait + ".LBB0_4:\n" + tail,
        "none": head + "\ts_nop 0\n" + tail,
    }
    want = {"ok": 0, "copy": 1, "loop": 1, "skipped": 0, "none": 1}
    for name, text in cases.items():
        f = tmp_path / (name + ".s")
        f.write_text(text)
        r = subprocess.run([sys.executable, tool, str(f), "k_demo"], capture_output=True, text=True)
        assert r.returncode == want[name], (name, r.stdout, r.stderr)
