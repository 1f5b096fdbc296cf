"""The multi-GPU GOP encoder inside libvcs_hip.so (vcs_group_*, vcs_encode_gop)
and the pipelined frame feeder (vcs_encode_p_async / vcs_wait), on the one GPU
a test box has:

  * n_ctx = 1: the RCCL transport for real (ncclCommInitAll over one device, ncclBroadcast,
    ncclAllGather on the collectives' stream, overlapped with the kernels);
  * n_ctx = 2 and 3 contexts that SHARE the device: RCCL refuses duplicate devices, so the
    group runs the identical schedule (sharding, double-buffered reference, staged vectors,
    uneven gather layout, events) with device-to-device copies.  N > 1 distinct GPUs is the
    driver's to run; it is unmeasured here.

Everything is compared with the CPU oracle, bit for bit."""
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


def _oracle_gop(oracle, ref, frames, bs, Q=None):
    return [oracle.encode_p(f, ref, bs, Q=Q) for f in frames]


@pytest.mark.parametrize("n_ctx,n_p", [(1, 5), (2, 5), (3, 7), (3, 2)])
def test_encode_gop_one_call(vh, oracle, n_ctx, n_p):
    """vcs_encode_gop(ctxs, n_ctx, frames, n_frames, mv_out, planes_out): frame 0 = I."""
    from synth import synth_gop
    H, W, bs = 144, 176, 16
    ref, frames = synth_gop(H, W, n_p, seed=40 + n_p)
    gop = np.concatenate([ref[None], frames])
    ctxs = [vh.Context(H, W, bs) for _ in range(n_ctx)]
    try:
        for rep in range(2):   # second call reuses the cached group (and the other reference buffer)
            mv, planes = vh.encode_gop(ctxs, gop, want_planes=True)
            want = _oracle_gop(oracle, ref, frames, bs)
            for k in range(n_p):
                assert np.array_equal(mv[k], want[k][0]), (rep, k)
                assert np.array_equal(planes[k].view(np.int64), want[k][1].view(np.int64)), (rep, k)
        # a GOP with a different I frame through the same group
        ref2, frames2 = synth_gop(H, W, n_p, seed=90)
        mv, _ = vh.encode_gop(ctxs, np.concatenate([ref2[None], frames2]))
        for k in range(n_p):
            assert np.array_equal(mv[k], oracle.me(frames2[k], ref2, bs)[0]), k
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("n_ctx", [1, 3])
def test_group_pipelined_gops(vh, oracle, n_ctx, monkeypatch):
    """The pieces, pipelined as bench.py drives them: encode GOP k, prefetch the I frame of GOP
    k+1 beside it, rotate through resident slot sets; uneven shards (7 frames over 3)."""
    from synth import synth_gop
    H, W, bs, n_p, sets = 144, 176, 16, 7, 2
    gops = [synth_gop(H, W, n_p, seed=s) for s in (5, 6, 7)]
    ctxs = [vh.Context(H, W, bs) for _ in range(n_ctx)]
    g = vh.Group(ctxs)
    try:
        assert g.world == n_ctx and g.rank0 == 0 and g.n_local == n_ctx
        # RCCL with one device (in sequence: nothing to overlap with); copies, overlapped, when shared
        assert g.transport == (2 if n_ctx == 1 else 3)
        g.layout(n_p, with_decode=True, sets=sets)
        table = [g.shard(i) for i in range(n_ctx)]
        assert table == [vh.shard_frames(n_p, n_ctx, r) for r in range(n_ctx)]
        # GOP 0 and 1 resident in the two sets; GOP 2 re-uses set 0 later
        for s in range(sets):
            g.use_set(s)
            g.upload(0, gops[s][1])
        g.set_ref(gops[0][0])
        g.use_set(0)
        g.encode()                       # GOP 0 (broadcasts its I frame itself)
        g.set_ref(gops[1][0])
        g.prefetch_ref()                 # GOP 1's I frame beside GOP 0's kernels
        mv0, ns0 = g.get_mv()
        g.use_set(1)
        g.encode()                       # GOP 1
        g.decode()
        mv1, ns1 = g.get_mv()
        for (ref, frames), mv, ns in ((gops[0], mv0, ns0), (gops[1], mv1, ns1)):
            for k in range(n_p):
                omv, ons = oracle.me(frames[k], ref, bs)
                assert np.array_equal(mv[k], omv) and ns[k] == ons, k
        # planes / decoded frames stay on the shard that made them
        for i, (first, count) in enumerate(table):
            for k in range(first, first + count):
                omv, oplanes, _ = oracle.encode_p(gops[1][1][k], gops[1][0], bs)
                slot = g.slot_of(i, k)
                assert np.array_equal(ctxs[i].gop_get_planes(slot), oplanes), k
                assert np.array_equal(ctxs[i].gop_get_decoded(slot),
                                      oracle.decode_p(omv, oplanes, gops[1][0], bs)), k
        # GOP 2 into set 0 again, third I frame: buffer parity wraps around
        g.use_set(0)
        g.upload(2, gops[2][1][2:])      # partial uploads land on the right shards
        g.upload(0, gops[2][1][:2])
        g.set_ref(gops[2][0])
        g.encode()
        mv2, ns2 = g.get_mv()
        for k in range(n_p):
            omv, ons = oracle.me(gops[2][1][k], gops[2][0], bs)
            assert np.array_equal(mv2[k], omv) and ns2[k] == ons, k
        assert g.max_over_ranks(1.5) == 1.5
        g.barrier()
    finally:
        g.close()
        for c in ctxs:
            c.close()


def test_group_config4_share_1080p(vh, oracle):
    """Config 4's dealing at a size where the shared-window search kernel runs: 29 P frames of a
    1080p GOP over 8 shards would give 4/3 frames each; here 3 contexts on the one GPU take
    10/10/9 of them."""
    from synth import synth_gop
    H, W, bs, n_p = 1080, 1920, 16, 29
    ref, frames = synth_gop(H, W, n_p, seed=1234)
    ctxs = [vh.Context(H, W, bs) for _ in range(3)]
    try:
        mv, _ = vh.encode_gop(ctxs, np.concatenate([ref[None], frames]))
        for k in range(n_p):
            assert np.array_equal(mv[k], oracle.me(frames[k], ref, bs)[0]), k
    finally:
        for c in ctxs:
            c.close()


def test_group_errors(vh):
    c1, c2 = vh.Context(64, 96, 16), vh.Context(64, 64, 16)
    try:
        with pytest.raises(vh.VcsError):
            vh.Group([c1, c2])                     # different geometry
        with pytest.raises(vh.VcsError):
            vh.Group([c1, c1])                     # the same context twice
        g = vh.Group([c1])
        with pytest.raises(vh.VcsError):
            vh.Group([c1])                         # already in a group
        with pytest.raises(vh.VcsError):
            g.encode()                             # no layout yet
        g.layout(3)
        with pytest.raises(vh.VcsError):
            g.encode()                             # no I frame staged
        with pytest.raises(vh.VcsError):
            g.upload(2, np.zeros((2, 64, 96, 3), np.uint8))   # beyond the GOP
        g.close()
        g2 = vh.Group([c1])                        # the context is free again
        g2.close()
    finally:
        c1.close()
        c2.close()


# ------------------------------------------------------------------ the feeder ----
def test_async_slots_vs_oracle(vh, oracle):
    """vcs_encode_p_async / vcs_wait: K frames in flight, results identical to the synchronous
    entry points, slots reusable, page-locked and pageable inputs."""
    from synth import synth_gop
    H, W, bs, n = 144, 176, 16, 9
    ref, frames = synth_gop(H, W, n, seed=77)
    pinned = [vh.pinned_empty((H, W, 3)) for _ in range(4)]
    with vh.Context(H, W, bs) as c:
        c.set_ref(ref)
        c.async_reserve(3)
        with pytest.raises(vh.VcsError):
            c.wait(0)                              # nothing in flight
        handles = {}
        for k in range(n):
            slot = k % 3
            if k >= 3:
                mv, ns = c.wait(slot)
                omv, oplanes, ons = oracle.encode_p(frames[k - 3], ref, bs)
                assert np.array_equal(mv, omv) and ns == ons, k
                assert np.array_equal(np.stack(list(handles.pop(k - 3))), oplanes), k
            src = frames[k]
            if k % 2:                              # alternate page-locked / pageable inputs
                pinned[k % 4][...] = frames[k]
                src = pinned[k % 4]
            handles[k] = c.encode_p_async(slot, src)
            if k == 4:
                with pytest.raises(vh.VcsError):
                    c.encode_p_async(slot, src)    # the slot is busy
        for k in range(n - 3, n):
            mv, ns = c.wait(k % 3)
            omv, oplanes, ons = oracle.encode_p(frames[k], ref, bs)
            assert np.array_equal(mv, omv) and ns == ons
            assert np.array_equal(np.stack(list(handles.pop(k))), oplanes)
        # a new reference between frames: set_ref waits for whatever is in flight
        h = c.encode_p_async(0, frames[0])
        c.set_ref(frames[1])
        mv, ns = c.wait(0)
        assert np.array_equal(mv, oracle.me(frames[0], ref, bs)[0])
        h2 = c.encode_p_async(1, frames[2])
        mv, ns = c.wait(1)
        assert np.array_equal(mv, oracle.me(frames[2], frames[1], bs)[0])
        del h, h2
        # strided views: the upload reads a contiguous copy that only the context keeps alive
        # until wait() -- the temporaries are dropped here before the copies have to finish
        import gc
        c.set_ref(ref)
        for k in range(3):
            c.encode_p_async(k, frames[k][:, ::-1])
        gc.collect()
        junk = [np.full((H, W, 3), 250 - 30 * k, np.uint8) for k in range(8)]   # reuses freed memory, if any
        for k in range(3):
            mv, ns = c.wait(k)
            omv, ons = oracle.me(np.ascontiguousarray(frames[k][:, ::-1]), ref, bs)
            assert np.array_equal(mv, omv) and ns == ons, k
        del junk


def test_pipelined_encoder_dropin(vh, oracle, capsys):
    """Encoder(resident=True) feeds frames through the async slots transparently: encode_frame
    returns at once, Frame.mv waits on first access, the Decoder reads planes straight from HBM;
    contents identical to the reference's call sequence (test_golden_codec_dropin)."""
    import decoder as D
    import encoder as E
    from conftest import load_golden
    from frame import LazyRowList
    g = load_golden("codec_small.npz")
    frames = g["frames"]
    H, W = frames.shape[1:3]
    enc = E.Encoder(pattern=["I", "P", "P"], shape=[H, W], block_size=8, with_DCT=True, resident=True)
    assert enc.async_slots >= 2
    for n in range(len(frames)):
        enc.encode_frame(frames[n], n)
    ps = [f for f in enc.encoded_frames if f.t == "P"]
    assert all(isinstance(f.mv, LazyRowList) for f in ps)
    dec = D.Decoder(encoded_frames=enc.encoded_frames, fps=25.0, shape=[H, W], ref_frames=enc.ref_frames,
                    block_size=8, with_DCT=True, sink=[].append)
    for n, fr in enumerate(enc.encoded_frames):
        if fr.t != "P":
            continue
        assert fr.mv == g["dct_mv_%d" % n].tolist() and len(fr.mv) == len(fr.c)
        assert np.array_equal(np.array(fr.mv), g["dct_mv_%d" % n])
        assert np.array_equal(dec._reconstruct_P_frame(fr, True), g["dct_full_%d" % n])
        assert np.array_equal(np.stack(list(fr.r)), g["dct_r_%d" % n])
    # a longer run: more frames than slots, two GOPs, vs the oracle
    from synth import synth_gop
    H, W = 144, 176
    ref, fs = synth_gop(H, W, 9, seed=12)
    seq = [ref] + list(fs[:4]) + [fs[4]] + list(fs[5:])       # frame 5 is the second I frame
    enc = E.Encoder(pattern=["I"] + ["P"] * 4, shape=[H, W], block_size=16, with_DCT=True,
                    dct_block_size=8, resident=True, async_slots=2)
    for n, f in enumerate(seq):
        if n % 2:                       # every other frame through the feeder's page-locked ring
            buf = enc.next_buffer()
            buf[...] = f
            f = buf
        enc.encode_frame(f, n)
    enc.flush()
    for n, fr in enumerate(enc.encoded_frames):
        if fr.t == "P":
            r = seq[(n // 5) * 5]
            omv, oplanes, _ = oracle.encode_p(seq[n], r, 16)
            assert np.array_equal(np.asarray(fr.mv), omv), n
            assert np.array_equal(np.stack(list(fr.r)), oplanes), n
    capsys.readouterr()


@pytest.mark.parametrize("serial", ["0", "1"])
def test_group_schedules_forced(vh, oracle, serial, monkeypatch):
    """Both schedules on both transports: VCS_GROUP_SERIAL=0 puts the one-device RCCL group's
    collectives on their own stream (what N > 1 does by default), =1 puts the copy transport in
    sequence on the kernels' stream."""
    from synth import synth_gop
    monkeypatch.setenv("VCS_GROUP_SERIAL", serial)
    H, W, bs, n_p = 144, 176, 16, 5
    for n_ctx in (1, 2):
        ctxs = [vh.Context(H, W, bs) for _ in range(n_ctx)]
        try:
            g = vh.Group(ctxs)
            assert g.transport == {("0", 1): 1, ("1", 1): 2, ("0", 2): 3, ("1", 2): 4}[(serial, n_ctx)]
            g.layout(n_p)
            for seed in (31, 32, 33):
                ref, frames = synth_gop(H, W, n_p, seed=seed)
                g.set_ref(ref)
                g.upload(0, frames)
                g.encode()
                g.prefetch_ref()
                mv, ns = g.get_mv()
                for k in range(n_p):
                    omv, ons = oracle.me(frames[k], ref, bs)
                    assert np.array_equal(mv[k], omv) and ns[k] == ons, (n_ctx, seed, k)
            g.close()
        finally:
            for c in ctxs:
                c.close()


@pytest.mark.gpu
def test_bench_rank_path_falls_back_without_rccl(tmp_path):
    """bench.py's process-per-GPU path when RCCL cannot be brought up: two ranks are put on
    ONE GPU (RCCL refuses duplicate devices), both must agree on the no-collective fall-back,
    finish, say so in the JSON line (the launcher contract: one line from rank 0) and leave with status 3."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    bench = os.path.join(os.path.dirname(here), "bench.py")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29671", VCS_BENCH_DEVICE="0", VCS_RENDEZVOUS_TAG="t%d" % os.getpid())
        procs.append(subprocess.Popen(
            [sys.executable, bench, "--gpus", "2", "--steps", "2", "--warmup", "1", "--height", "256",
             "--width", "320", "--frames", "3", "--no-extras", "--no-cpu-baseline"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=420) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 3, (p.returncode, e[-2000:])   # the line is printed, the status says "degraded"
    assert outs[1][0].strip() == ""
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert line["config"]["frames_per_step_total"] == 6
    assert "FALL-BACK" in line["config"]["collectives"]
    # ... and where a driver that only reads the top level looks: not a point of a scaling curve
    assert line["degraded"] == "no_collectives" and line["scaling"] is None
