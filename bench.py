#!/usr/bin/env python3
"""bench.py -- macroblocks/s through the inter-frame encode hot path on MI355X.

Workload (BASELINE.json metric: "macroblocks/sec (ME+DCT) at 1080p +-32px
search"): a synthetic 1080p 30-frame GOP (1 I + 29 P frames, synth_gop), 16x16
motion blocks -> +-32 px window on the reference's stride-5 grid, 8x8 DCT with
the QF=50 tables, float64 coefficient planes.  One STEP = one pass of the hot
path over the 29 P frames of a GOP: the motion-search kernel followed by the
fused MC + residual + colour + DCT + quantise kernel, with every input already
resident in HBM and every output left in HBM.

`value` is the COLD stream: the timed loop rotates through three resident copies
of the GOP, so a step's 180 MB of input frames were last touched three steps ago
and nothing of them is left in the 256 MB Infinity Cache -- what a stream of
GOPs costs.  The same run also reports the loop over ONE resident GOP
(`resident_one_gop`), which the cache flatters.

N > 1: weak scaling -- one long GOP with 29*N P frames, GPU r owns P frames
[29r, 29r+29).  The only exchanges of the path are inside the timed step: an RCCL
broadcast of the I frame from GPU 0 and an all-gather of the motion vectors;
coefficient planes stay on the GPU that produced them.  RCCL lives inside
libvcs_hip.so (vcs_group_*); no PyTorch is imported.  Two launch forms:
  python bench.py --gpus N                   one process, one context per GPU
  python -m torch.distributed.run ... bench.py --gpus N
                                             one process per GPU; the launcher only
                                             hands out RANK / WORLD_SIZE / LOCAL_RANK
`--strong` is BASELINE config 4: ONE 4K 30-frame GOP whose 29 P frames are split
over the N GPUs (4 / 3 frames each at N = 8).

Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "vcs-h264_amd")
for p in (REPO, PKG, os.path.join(PKG, "InterframeCompression")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

INIT_TIMEOUT_S = float(os.environ.get("VCS_BENCH_INIT_TIMEOUT_S", "60"))   # RCCL bring-up / file rendezvous
state_stuck = {}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~5500 measured copy
# issue cycles per wave-instruction on one SIMD, measured on this chip by tools/ub_valu.hip
ISSUE_CYCLES = {"v_lerp_u8": 4, "v_and_b32": 2, "v_bcnt_u32_b32": 4, "other": 4}
# k_me16b is made of v_bitop3_b32 / v_bcnt_u32_b32 (VOP3): 3.6-3.7 issue cycles per wave-instruction
# with 8 waves per SIMD (tools/ub_bits.hip: 586 G wave-instructions/s chip-wide), priced at 4
ISSUE_CYCLES_VOP3 = 4


def kernel_ms(ms, steps):
    """Per-step kernel times of one timing() result; me_prep is k_bits_cur (bit-plane search only)."""
    import vcs_hip
    out = {"me": round(ms[vcs_hip.T_ME] / steps, 4), "enc_tail": round(ms[vcs_hip.T_ENC_TAIL] / steps, 4)}
    if ms[vcs_hip.T_ME_PREP] > 0:
        out["me_prep"] = round(ms[vcs_hip.T_ME_PREP] / steps, 4)
    return out


def cpu_baseline(H, W, bs, ref, frames, budget_s=10.0):
    """The CPU checker on the same workload (`kind: port`, our C restatement with OpenMP
    on the box's host cores), on a bounded sample of the GOP's P frames; beside it the
    structure-faithful NumPy loop (oracle/ref_loop.py, one thread), whose run time is tied
    to the reference's own."""
    from oracle import oracle, ref_loop
    cores = oracle.usable_cpus(16)
    oracle.set_threads(cores)
    oracle.encode_p(frames[0][:64, :64].copy(), ref[:64, :64].copy(), bs)  # warm (build/load)
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < budget_s:   # cycle through the GOP until the budget is used
        oracle.encode_p(frames[done % len(frames)], ref, bs)
        done += 1
    dt = time.perf_counter() - t0
    N = (H // bs) * (W // bs)
    out = {"value": round(done * N / dt / 1e3, 3), "unit": "macroblocks/ms", "cores": cores,
           "kind": "port",
           "sample": "%d P-frame encodes cycling through the GOP (%dx%d, ME %dx%d + 8x8 DCT), "
                     "oracle/vcs_oracle.c with OpenMP, %.1f s" % (done, W, H, bs, bs, dt)}
    try:
        mbs, t_me, t_dct = ref_loop.timed_sample(frames[0], ref, bs, oracle.make_q(50.0), budget_s=6.0)
        out["reference_structured_loop"] = {
            "value": round(mbs / (t_me + t_dct) / 1e3, 3), "me_only": round(mbs / t_me / 1e3, 3),
            "unit": "macroblocks/ms", "cores": 1, "kind": "port",
            "sample": "%d interior macroblocks of one P frame + their 8x8 tiles, per-block / per-tile "
                      "NumPy loop with the reference's call structure (oracle/ref_loop.py); the "
                      "unmodified reference ran the same search at 1.58 and this loop at 1.63 "
                      "macroblocks/ms on one core of the build container (DESIGN.md 6)" % mbs}
    except Exception as e:  # the headline never depends on it
        out["reference_structured_loop"] = {"error": repr(e)}
    return out


def parity_columns(ctx, ref, frames, mv, slot0, bs):
    """what was just timed, against the oracle, outside the timed region."""
    from oracle import oracle
    n = len(frames)
    exact = sum(int(np.array_equal(mv[k], oracle.me(frames[k], ref, bs)[0])) for k in range(n))
    k = n // 2
    omv, oplanes, _ = oracle.encode_p(frames[k], ref, bs)
    planes = ctx.gop_get_planes(slot0 + k)
    dmax = float(np.max(np.abs(planes - oplanes)))
    bits = bool(np.array_equal(planes.view(np.int64), oplanes.view(np.int64)))
    dec = ctx.decode_p(mv[k], planes)
    odec = oracle.decode_p(omv, oplanes, ref, bs)

    def psnr(a, b):
        mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
        return 99.0 if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)
    return {"mv_exact_fraction": exact / max(1, n), "frames_compared": n,
            "planes_max_abs_diff": dmax, "planes_bit_identical": bits, "planes_frame": k,
            "psnr_decoded_db": round(psnr(dec, frames[k]), 4),
            "psnr_delta_db": round(psnr(dec, frames[k]) - psnr(odec, frames[k]), 6),
            "decoded_pixels_identical": bool(np.array_equal(dec, odec)),
            "oracle": "oracle/vcs_oracle.c (pinned to the reference's outputs, tests/golden)"}


def host_path(H, W, bs, ref, frames, device, n_frames=48):
    """SURVEY 8d "separately including H2D/D2H": the reference's own driving loop
    (main.py:34-43: one Encoder.encode_frame per captured frame) through the drop-in
    Encoder.  Three ways: `default` = what an unchanged main.py gets (no keyword, no environment
    variable, frames in ordinary pageable arrays, the whole I frame hashed per P frame);
    `pipelined` = the frame feeder (resident=True, frames produced into its page-locked ring);
    `synchronous` = the feeder with one host synchronisation per frame."""
    import encoder as E
    import vcs_hip
    from _runtime import release_contexts
    pattern = ["I"] + ["P"] * 29
    res = {}
    saved_env = os.environ.pop("VCS_RESIDENT", None)
    for label, kw, pinned in (("default", {}, False), ("pipelined", {"resident": True}, True),
                              ("synchronous", {"resident": True, "async_slots": 0}, True)):
        enc = E.Encoder(pattern=pattern, shape=[H, W], block_size=bs, with_DCT=True, dct_block_size=8,
                        device=device, **kw)
        sink = io.StringIO()
        if pinned:
            iframe = vcs_hip.pinned_empty((H, W, 3))
            ring = [iframe] + [enc.next_buffer() for _ in range(5)]   # the feeder's page-locked ring
        else:
            ring = [np.empty((H, W, 3), np.uint8) for _ in range(6)]   # pageable, as cap.read() returns them
        ring[0][...] = ref
        for k in range(1, 6):
            ring[k][...] = frames[k % len(frames)]

        def run(first, n, produce):
            t0 = time.perf_counter()
            done = 0
            for k in range(first, first + n):
                if k % 30 == 0:                      # the next GOP's I frame
                    enc.flush()
                    enc.encode_frame(ring[0], k)
                    continue
                buf = ring[k % 5 + 1]
                if produce:
                    buf[...] = frames[k % len(frames)]   # the producer's write (e.g. a video decoder's)
                enc.encode_frame(buf, k)
                done += 1
            enc.flush()
            return (time.perf_counter() - t0) / done, done

        with contextlib.redirect_stdout(sink):   # encode_frame prints like the reference
            enc.encode_frame(ring[0], 0)
            run(1, 5, False)                     # warm
            # steady state: the better of two passes (a pass that has to grow the device heap for
            # its plane slabs pays tens of milliseconds once)
            per_frame, done = min(run(6, n_frames, False), run(6 + n_frames, n_frames, False))
            per_frame_prod, _ = min(run(6 + 2 * n_frames, n_frames, True), run(6 + 3 * n_frames, n_frames, True))
        N = (H // bs) * (W // bs)
        res[label] = {"encode_frame_ms": round(per_frame * 1e3, 4),
                      "with_producer_copy_ms": round(per_frame_prod * 1e3, 4),
                      "mb_per_ms": round(N / per_frame / 1e3, 1)}
        res[label]["last_frame_blocks"] = len(enc.encoded_frames[-1].mv)
        del enc
        release_contexts()
    if saved_env is not None:
        os.environ["VCS_RESIDENT"] = saved_env
    out = dict(res["pipelined"])
    out["synchronous_encode_frame_ms"] = res["synchronous"]["encode_frame_ms"]
    out["default_mode"] = dict(res["default"], what=(
        "Encoder(pattern, shape, block_size, with_DCT) exactly as main.py:29-43 builds and drives it: no keyword "
        "beyond dct_block_size, no environment variable, pageable frames, the I frame hashed per P frame "
        "(encoder.py:42's aliasing); planes stay in HBM behind Frame.r, vectors arrive lazily"))
    out["what"] = ("drop-in Encoder(resident=True).encode_frame per 1080p frame from page-locked "
                   "buffers, H2D of the frame and D2H of its vectors included, planes stay in HBM; "
                   "%d frames; with_producer_copy_ms adds the producer's own 6 MB write into the "
                   "page-locked buffer before every call; default_mode: see there" % n_frames)
    return out


def coeff_modes(H, W, bs, ref, frames, device, steps=8):
    """the same GOP step with float32 planes and with rounded int16 levels
    (np.round of the reference's quotient): the output stream shrinks 2x / 4x."""
    import vcs_hip
    out = {}
    n = len(frames)
    N = (H // bs) * (W // bs)
    F = 3 * H * W
    for name, mode, per_px in (("f32", vcs_hip.VCS_COEFF_F32, 4), ("i16_rounded", vcs_hip.VCS_COEFF_I16, 2)):
        with vcs_hip.Context(H, W, bs, 8, device=device) as c:
            c.set_params(coeff_mode=mode)
            c.set_ref(ref)
            c.gop_reserve(2 * n)
            c.gop_upload(0, frames)
            c.gop_upload(n, frames)
            for k in range(2):
                c.ref_written()
                c.gop_encode(k * n, n)
            c.sync()
            t_end = time.perf_counter() + float(os.environ.get("VCS_BENCH_CONDITION_MS", "250")) / 1e3
            k = 0
            while time.perf_counter() < t_end:      # the same clock conditioning as the headline
                c.ref_written()
                c.gop_encode((k % 2) * n, n)
                k += 1
                if k % 16 == 0:
                    c.sync()
            c.sync()
            t0 = time.perf_counter()
            for k in range(steps):
                c.ref_written()
                c.gop_encode((k % 2) * n, n)
            c.sync()
            dt = (time.perf_counter() - t0) / steps
            c.timing_enable(True)           # kernel time: a second loop (the events cost wall time)
            c.timing_reset()
            for k in range(steps):
                c.ref_written()
                c.gop_encode((k % 2) * n, n)
            c.sync()
            ms, l = c.timing()
            tail = ms[vcs_hip.T_ENC_TAIL] / max(1, int(l[vcs_hip.T_ENC_TAIL]))
            alg = ((2 + per_px) * F + 4 * N) * n
            out[name] = {"value": round(n * N / dt / 1e3, 1), "unit": "macroblocks/ms",
                         "ms_per_step": round(dt * 1e3, 4), "enc_tail_ms": round(tail, 4),
                         "enc_tail_gbs": round(alg / (tail * 1e-3) / 1e9, 1)}
    out["note"] = "two resident GOP copies in turn; float64 (the reference's type) is the headline"
    return out


def reference_default_point(H, W, ref, frames, device, steps=10, sets=3):
    """The reference's own default operating point (main.py:15: block_size = 8 -> 8x8 motion blocks, +-16 px,
    stride round(8 / 3) = 3, motion.py:18 / :132) on the same GOP: the whole step (k_bits_ref8, k_bits_cur8,
    k_me8b, k_fdct), `sets` resident copies in turn like the headline, vectors of three frames against the
    oracle outside the timed region."""
    import vcs_hip
    from oracle import oracle
    bs, n = 8, len(frames)
    N = (H // bs) * (W // bs)
    with vcs_hip.Context(H, W, bs, 8, device=device) as c:
        c.set_ref(ref)
        c.gop_reserve(sets * n)
        for s in range(sets):
            c.gop_upload(s * n, frames)
        for k in range(sets):
            c.ref_written()
            c.gop_encode(k * n, n)
        c.sync()
        t_end = time.perf_counter() + float(os.environ.get("VCS_BENCH_CONDITION_MS", "250")) / 1e3
        k = 0
        while time.perf_counter() < t_end:      # the same clock conditioning as the headline
            c.ref_written()
            c.gop_encode((k % sets) * n, n)
            k += 1
            if k % 16 == 0:
                c.sync()
        c.sync()
        t0 = time.perf_counter()
        for k in range(steps):
            c.ref_written()
            c.gop_encode((k % sets) * n, n)
        c.sync()
        dt = (time.perf_counter() - t0) / steps
        last = ((steps - 1) % sets) * n
        mv, ns = c.gop_get_mv(last, n)
        c.timing_enable(True)
        c.timing_reset()
        for k in range(steps):
            c.ref_written()
            c.gop_encode((k % sets) * n, n)
        c.sync()
        ms, _ = c.timing()
    cmp_frames = sorted({0, n // 2, n - 1})
    exact = sum(int(np.array_equal(mv[f], oracle.me(frames[f], ref, bs)[0])) for f in cmp_frames)
    return {"what": "main.py:15's operating point: 8x8 motion blocks, +-16 px, stride 3 (motion.py:18, :132), 8x8 DCT QF 50, "
                    "float64 planes; %d P frames per step, %d resident copies in turn" % (n, sets),
            "value": round(n * N / dt / 1e3, 1), "unit": "8x8 blocks/ms", "ms_per_step": round(dt * 1e3, 4),
            "blocks_per_frame": N, "searched_block_fraction": round(1.0 - float(ns.sum()) / (n * N), 4),
            "kernel_ms_per_step": kernel_ms(ms, steps),
            "mv_exact_fraction": exact / len(cmp_frames), "frames_compared": len(cmp_frames)}


class _FileSync:
    """Barrier / all-to-all of small values through files in /dev/shm: ONLY for the bench's
    fall-back below (the ranks of a launcher job agreeing that RCCL could not be brought up, and
    the max-over-ranks of the timing when it could not).  Never part of the encode path."""

    def __init__(self, rank, world, tag, timeout_s=60.0):
        import tempfile
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        self.prefix = os.path.join(base, "vcs_bench_sync_%d_%s_w%d" % (os.getuid(), tag, world))
        self.rank, self.world, self.seq, self.timeout_s, self.mine = rank, world, 0, timeout_s, []

    def exchange(self, value):
        """Every rank's `value` (a short string), in rank order."""
        self.seq += 1
        path = "%s.%d.%d" % (self.prefix, self.seq, self.rank)
        with open(path + ".tmp", "w") as f:
            f.write(value)
        os.replace(path + ".tmp", path)
        self.mine.append(path)
        out, t0 = [], time.time()
        for r in range(self.world):
            q = "%s.%d.%d" % (self.prefix, self.seq, r)
            while True:
                try:
                    out.append(open(q).read())
                    break
                except OSError:
                    if time.time() - t0 > self.timeout_s:
                        raise TimeoutError("rank %d: rank %d never reached sync point %d" % (self.rank, r, self.seq))
                    time.sleep(0.002)
        while len(self.mine) > 2:          # everybody has read a file once they have written two more
            try:
                os.unlink(self.mine.pop(0))
            except OSError:
                pass
        return out

    def close(self):
        """Leaves nothing behind: after a last symmetric round every rank drops a marker and
        goes; rank 0 waits for the markers (so nobody is still reading) and removes the lot."""
        import glob
        self.exchange("bye")
        self.seq += 1
        open("%s.%d.%d" % (self.prefix, self.seq, self.rank), "w").close()
        if self.rank:
            return
        t0 = time.time()
        while (not all(os.path.exists("%s.%d.%d" % (self.prefix, self.seq, r)) for r in range(self.world))
               and time.time() - t0 < self.timeout_s):
            time.sleep(0.002)
        for q in glob.glob(self.prefix + ".*"):
            try:
                os.unlink(q)
            except OSError:
                pass


class _ReplicaShards:
    """The bench's fall-back when the RCCL group of a launcher job cannot be created: every
    rank still encodes its own chunk of the long GOP's P frames with the same kernels, but the
    I frame reaches each GPU by its own upload and the vectors stay on their GPUs -- no
    collective.  The JSON line says so under config.collectives; it is there so that a
    communication set-up problem costs the exchange step and not the whole measurement."""
    transport = 0
    n_local = 1
    partial = True

    def __init__(self, ctx, rank, world, fsync):
        import vcs_hip
        self.ctx, self.rank, self.world, self.fs = ctx, rank, world, fsync
        self._shard = vcs_hip.shard_frames
        self.set = 0

    def layout(self, n_p, with_decode=False, sets=1):
        self.n_p = n_p
        self.first, self.count = self._shard(n_p, self.world, self.rank)
        self.ctx.gop_reserve(max(1, sets * self.count), with_decode=with_decode)

    def shard(self, i=0):
        return self.first, self.count

    def use_set(self, k):
        self.set = k

    def upload(self, first_p, frames):
        self.ctx.gop_upload(self.set * self.count + first_p - self.first, frames)

    def set_ref(self, ref):
        self.ctx.set_ref(ref)

    def prefetch_ref(self):
        pass

    def encode(self):
        self.ctx.ref_written()
        if self.count:
            self.ctx.gop_encode(self.set * self.count, self.count)

    def sync(self):
        self.ctx.sync()

    def barrier(self):
        self.fs.exchange("b")

    def max_over_ranks(self, v):
        return max(float(x) for x in self.fs.exchange(repr(float(v))))

    def get_mv(self):
        N = self.ctx.N
        mv = np.zeros((self.n_p, N, 2), np.int16)
        ns = np.zeros(self.n_p, np.uint32)
        if self.count:
            m, n = self.ctx.gop_get_mv(self.set * self.count, self.count)
            mv[self.first:self.first + self.count], ns[self.first:self.first + self.count] = m, n
        return mv, ns

    def close(self):
        self.fs.close()


def group_enqueue_probe(H, W, bs, ref, frames, device, steps=12):
    """Host time to enqueue one GOP step through the group layer for 1, 2 and 3 contexts on ONE device
    (same-device groups run the multi-GPU schedule with device copies for collectives): what a
    single-process `--gpus N` job pays per step on the host, by number of local shards.  The shards are
    enqueued by one thread each inside vcs_group_encode (VCS_GROUP_THREADS=0: all from the caller)."""
    import vcs_hip
    out = {}
    n_total = len(frames)
    for n_ctx in (1, 2, 3):
        ctxs = [vcs_hip.Context(H, W, bs, 8, device=device) for _ in range(n_ctx)]
        g = vcs_hip.Group(ctxs)
        try:
            g.layout(n_total)
            for i in range(g.n_local):
                first, count = g.shard(i)
                if count:
                    g.upload(first, frames[first:first + count])
            g.set_ref(ref)
            for _ in range(3):
                g.encode()
                g.prefetch_ref()
            g.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                g.encode()
                g.prefetch_ref()
            t_enq = time.perf_counter() - t0
            g.sync()
            dt = time.perf_counter() - t0
            out[str(n_ctx)] = {"host_enqueue_ms_per_step": round(t_enq / steps * 1e3, 4),
                               "ms_per_step": round(dt / steps * 1e3, 4)}
        finally:
            g.close()
            for c in ctxs:
                c.close()
    out["what"] = ("the %d P frames of the GOP split over n same-device contexts (keys: n); enqueue = vcs_group_encode + "
                   "vcs_group_prefetch_ref per step from Python; threads per shard: %s"
                   % (n_total, "off" if os.environ.get("VCS_GROUP_THREADS") == "0" else "on"))
    return out


def intra_mode(args, real_stdout):
    """BASELINE config 5: intramodes.py 4x4 intra prediction + the 4x4 core
    transform on a 1080p I frame (resized to multiples of 16 as intraframe.py:324 does: 1920 x
    1072), one GPU.  Kernel-only times (HIP events), blocks/ms and achieved HBM GB/s per kernel;
    the entry points themselves also move the float64 planes over PCIe (host_ms)."""
    import vcs_hip
    H, W = 1072, 1920
    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[0:H, 0:W]
    base = (128 + 70 * np.sin(xx / 23.0) * np.cos(yy / 17.0)).astype(np.int16)
    Y, Cr, Cb = [np.clip(base + rng.integers(-30, 31, (H, W)), 0, 255).astype(np.uint8) for _ in range(3)]
    hw = H * W
    res = {}
    with vcs_hip.Context(H, W, 16) as c:
        yres = c.intra_luma(Y, 4)[0]
        jobs = (("luma4x4", lambda: c.intra_luma(Y, 4), hw // 16, hw + 16 * hw + 8 * (hw // 16)),
                ("luma16x16", lambda: c.intra_luma(Y, 16), hw // 256, hw + 16 * hw + 8 * (hw // 256)),
                ("chroma8x8", lambda: c.intra_chroma(Cr, Cb), hw // 64, 2 * hw + 32 * hw + 8 * (hw // 64)),
                ("transform4x4", lambda: c.intra_transform4x4(yres), hw // 16, 16 * hw))
        for name, fn, blocks, alg in jobs:
            fn()
            c.timing_enable(True)
            c.timing_reset()
            t0 = time.perf_counter()
            for _ in range(max(1, args.steps // 4)):
                fn()
            host = (time.perf_counter() - t0) / max(1, args.steps // 4)
            ms, l = c.timing()
            c.timing_enable(False)
            k_ms = ms[vcs_hip.T_AUX] / max(1, int(l[vcs_hip.T_AUX]))
            res[name] = {"kernel_ms": round(k_ms, 4), "blocks": blocks,
                         "blocks_per_ms": round(blocks / k_ms, 1),
                         "algorithmic_bytes": alg, "achieved_gbs": round(alg / (k_ms * 1e-3) / 1e9, 1),
                         "hbm_frac": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "host_ms_incl_pcie": round(host * 1e3, 3)}
        # the same kernels over a batch of I frames resident in HBM (vcs_intra_batch_*: the I frames of many GOPs):
        # one launch per kernel, blockIdx.y = frame -- chroma8x8's 240 serial columns per frame become 240 n
        nb = 24
        c.intra_batch_reserve(nb)
        c.intra_batch_upload(0, np.stack([Y] * nb), np.stack([Cr] * nb), np.stack([Cb] * nb))
        batch = {}
        for name, what, blocks, alg in (("luma4x4", 1, hw // 16, hw + 16 * hw + 8 * (hw // 16)),
                                        ("luma16x16", 2, hw // 256, hw + 16 * hw + 8 * (hw // 256)),
                                        ("chroma8x8", 4, hw // 64, 2 * hw + 32 * hw + 8 * (hw // 64))):
            c.intra_batch_run(0, nb, what)
            c.sync()
            c.timing_enable(True)
            c.timing_reset()
            for _ in range(5):
                c.intra_batch_run(0, nb, what)
            c.sync()
            ms, l = c.timing()
            c.timing_enable(False)
            k_ms = ms[vcs_hip.T_AUX] / max(1, int(l[vcs_hip.T_AUX]))
            batch[name] = {"frames": nb, "kernel_ms": round(k_ms, 4), "ms_per_frame": round(k_ms / nb, 5),
                           "blocks_per_ms": round(nb * blocks / k_ms, 1),
                           "achieved_gbs": round(nb * alg / (k_ms * 1e-3) / 1e9, 1),
                           "hbm_frac": round(nb * alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    out = {"metric": "intra prediction (config 5), 4x4 luma blocks/ms at 1080p", "value": res["luma4x4"]["blocks_per_ms"],
           "unit": "4x4 blocks/ms", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": res["luma4x4"]["kernel_ms"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u8 -> f64 (the reference's output type)", "data": "synthetic",
           "config": {"workload": "configs[4]: luma4x4 / luma16x16 / chroma8x8 / transform4x4 on a %dx%d I frame" % (W, H)},
           "kernels": res,
           "batch_of_i_frames": batch,
           "note": "chroma8x8 is serial down each block column (the reference predicts Cb from the residual "
                   "above, intraframe.py:266): %d waves for the frame" % (W // 8)}
    os.write(real_stdout, (json.dumps(out) + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--bs", type=int, default=16, help="motion block size (window = 2*bs)")
    ap.add_argument("--frames", type=int, default=29, help="P frames per GPU per step (weak scaling)")
    ap.add_argument("--strong", action="store_true",
                    help="BASELINE config 4: ONE 3840x2160 30-frame GOP, its 29 P frames split over the GPUs")
    ap.add_argument("--rotate-sets", type=int, default=3,
                    help="resident copies of the GOP the timed loop rotates through (1 = re-encode one "
                         "resident GOP, whose inputs then sit in the Infinity Cache)")
    ap.add_argument("--group", action="store_true",
                    help="take the multi-GPU code path (vcs_group_*, RCCL) even with one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements (one-GOP loop, host path, parity, bandwidth)")
    ap.add_argument("--with-decode", action="store_true",
                    help="also time the decode kernel over the GOP even with --no-extras (it is part of "
                         "the default secondary measurements)")
    ap.add_argument("--intra", action="store_true",
                    help="BASELINE config 5 instead: the intra-prediction kernels on a 1080p I frame")
    ap.add_argument("--same-ref", action="store_true",
                    help="diagnostic: do not declare a new I frame per step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.steps = max(1, args.steps)
    args.warmup = max(0, args.warmup)
    if world > 1:
        args.gpus = world                 # one process per GPU (launcher)
    n_local = 1 if world > 1 else max(1, args.gpus)
    n_gpus = world if world > 1 else n_local
    use_group = args.group or n_gpus > 1
    args.with_decode = args.with_decode or not args.no_extras   # second line of SURVEY 8d
    # VCS_BENCH_REF_PIPELINE=1: the next GOP's I frame is declared while this GOP is encoded (vcs_ref_next_ready)
    # and its search tables are cut beside this GOP's search kernel instead of in front of the next one's.
    # Measured break-even at best (profiles/r04/experiments.txt): off by default.
    pipelined_ref = os.environ.get("VCS_BENCH_REF_PIPELINE", "0") == "1" and not args.same_ref
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a
    # version banner), so file descriptor 1 is pointed at stderr for the whole run and the
    # line goes to a private copy of the original stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import vcs_hip
    from synth import synth_gop

    if args.intra:
        if rank == 0:
            intra_mode(args, real_stdout)
        return
    H = args.height or (2160 if args.strong else 1080)
    W = args.width or (3840 if args.strong else 1920)
    bs = args.bs
    sets = max(1, args.rotate_sets)
    if world == 1 and vcs_hip.device_count() < n_local:
        sys.exit("bench.py --gpus %d: only %d GPU(s) visible" % (n_local, vcs_hip.device_count()))
    # VCS_BENCH_DEVICE: tests put two ranks on one GPU (RCCL refuses that: the fall-back's trigger)
    devices = [int(os.environ.get("VCS_BENCH_DEVICE", local_rank))] if world > 1 else list(range(n_local))
    ctxs = [vcs_hip.Context(H, W, bs, 8, device=d) for d in devices]
    ctx = ctxs[0]
    N, F = ctx.N, ctx.F
    n_total = 29 if args.strong else args.frames * n_gpus   # P frames of the (long) GOP
    group = None
    rccl_failed = None
    launched = "RANK" in os.environ          # under torch.distributed.run (or any launcher)
    if use_group:
        if world > 1 or launched:            # one process per GPU: ncclCommInitRank
            # ncclCommInitRank and the first barrier block until EVERY rank has joined: if only some
            # ranks fail, the others would sit there for ever.  They run in a thread that is given the
            # file rendezvous' time; a rank still inside after that reports a time-out, all ranks fall
            # back together, and the process leaves through os._exit at the end (the thread is stuck).
            import threading
            box = {}

            def bring_up():
                try:
                    uid = vcs_hip.rendezvous_unique_id(rank, world, timeout_s=INIT_TIMEOUT_S)
                    g_ = vcs_hip.Group.from_rank(ctx, uid, world, rank)
                    g_.barrier()
                    box["group"] = g_
                except Exception as e:       # noqa: BLE001 -- anything RCCL / the rendezvous raises
                    box["err"] = repr(e)
            th = threading.Thread(target=bring_up, daemon=True)
            th.start()
            th.join(INIT_TIMEOUT_S + 5.0)
            err, group = "", box.get("group")
            if th.is_alive():
                err, group = "timed out after %.0f s in the RCCL set-up (another rank never joined)" % INIT_TIMEOUT_S, None
                state_stuck["thread"] = True
            elif "err" in box:
                err, group = box["err"], None
            if world > 1:
                # the ranks agree (through files, RCCL being what is in doubt) whether ALL of them
                # have a communicator; if not, all take the no-collective fall-back
                fsync = _FileSync(rank, world, vcs_hip._job_tag(), timeout_s=INIT_TIMEOUT_S + 30.0)
                errs = [e for e in fsync.exchange(err) if e]
                if errs:
                    if group is not None:
                        group.close()
                    group = _ReplicaShards(ctx, rank, world, fsync)
                    rccl_failed = errs[0][:300]
                else:
                    fsync.close()
            elif err:
                raise RuntimeError(err)
            vcs_hip.rendezvous_cleanup(rank, world)
        else:
            group = vcs_hip.Group(ctxs)
        group.layout(n_total, with_decode=args.with_decode, sets=sets)
        shards = [group.shard(i) for i in range(group.n_local)]
    else:
        shards = [(0, n_total)]
        ctx.gop_reserve(sets * n_total, with_decode=args.with_decode)
    # synthetic, deterministic; each process builds only the slices its GPUs own
    ref = None
    local_frames = []
    for first, count in shards:
        r, fr = synth_gop(H, W, count, seed=1234, first=1 + first)
        ref = r
        local_frames.append(fr)
    for k in range(sets):
        if group is not None:
            group.use_set(k)
            for (first, count), fr in zip(shards, local_frames):
                if count:
                    group.upload(first, fr)
        else:
            ctx.gop_upload(k * n_total, local_frames[0])
    if group is not None:
        group.use_set(0)
        group.set_ref(ref)               # the I frame enters HBM on GPU 0 only (no-op elsewhere)
    else:
        ctx.set_ref(ref)
        if pipelined_ref:
            ctx.set_ref_next(ref)
    state = {"k": 0}

    def step(rot=sets):
        k = state["k"] % rot
        state["k"] += 1
        if group is not None:
            group.use_set(k)
            group.encode()                # I-frame broadcast (prefetched) + ME + tail + MV all-gather
            group.prefetch_ref()          # the next GOP's I frame, beside this GOP's kernels
            return
        # every step is a new GOP: the per-reference work (the sum table) is redone inside
        # the timed region, once per GOP as in a real encode
        if not args.same_ref:
            if pipelined_ref:
                # the double-buffered reference of a stream of GOPs: this GOP's I frame was staged in the
                # alternate buffer while the previous GOP was encoded (here: the same picture, staged once),
                # and the library cut its search tables beside the previous GOP's search kernel
                ctx.ref_swap()
                ctx.ref_next_ready()
            else:
                ctx.ref_written()
        ctx.gop_encode(k * n_total, n_total)   # ME kernel + fused tail kernel

    def fence():
        for c in ctxs:
            c.sync()
        if group is not None:
            group.sync()
            group.barrier()

    def timed(rot, steps, warm, events=False):
        state["k"] = 0
        for _ in range(warm):
            step(rot)
        fence()
        for c in ctxs:
            c.timing_enable(events and not os.environ.get("VCS_BENCH_NO_EVENTS"))
            c.timing_reset()
            c.me_cycles(reset=True)
            c.me_clock(reset=True)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(rot)
        t_enq = time.perf_counter() - t0
        fence()
        dt = time.perf_counter() - t0
        per = [c.timing() for c in ctxs]
        cyc = [c.me_cycles(reset=True) for c in ctxs]
        state["clock_ghz"] = ctxs[0].me_clock(reset=True)
        for c in ctxs:
            c.timing_enable(False)
        if group is not None:
            dt = group.max_over_ranks(dt)
        return dt, t_enq, per, cyc

    # Two passes of W warm-up + K timed steps.  The first starts on an idle GPU: its 20-odd
    # milliseconds fall into the clock governor's ramp (measured: 2.06 GHz and 253 k
    # macroblocks/ms against 2.25 GHz and 265 k once the GPU has been busy for 0.1 s).  Between
    # the passes the same step runs untimed for ~0.25 s, so that the second pass -- the
    # headline -- is what a stream of GOPs sees; the first is reported beside it.
    # The timed passes carry NO timing events: HIP events attached to the five kernels of a step
    # cost ~0.01 ms of it (0.678 vs 0.667 ms, measured; the in-kernel clock measurement of k_me16b,
    # once it reported from every workgroup, another 0.03 -- it is sampled now).  Per-kernel durations
    # (kernel_ms_per_step, roofline) come from a third pass of the same K steps with the events on.
    warm = max(args.warmup, sets if sets > 1 else 0)
    dt0, _, _, _ = timed(sets, args.steps, warm)
    first_pass = {"value": round(n_total * N * args.steps / dt0 / 1e3, 2), "unit": "macroblocks/ms",
                  "ms_per_step": round(dt0 / args.steps * 1e3, 4),
                  "note": "the same %d + %d steps started on an idle GPU, inside the clock ramp" % (warm, args.steps)}
    cond_ms = float(os.environ.get("VCS_BENCH_CONDITION_MS", "250"))
    cond_steps = min(5000, int(cond_ms / max(1e-3, dt0 / args.steps * 1e3))) if cond_ms > 0 else 0
    dt, t_enq, _, _ = timed(sets, args.steps, warm + cond_steps)
    dt_ev, _, per, cyc = timed(sets, args.steps, 2, events=True)   # the GPU is warm: straight on
    ms, launches = per[0]
    clock_ghz_events_pass = state.get("clock_ghz")      # (later passes overwrite state)
    # the search kernel WITHOUT the Infinity Cache prefetch workgroups at its end (they stream the frames for
    # the tail while the launch's last search workgroups drain): the part of the launch the vector-issue
    # roof applies to.  One more short pass, prefetch off (the tail of that pass is slow: not reported).
    me_alone_ms = None
    if group is None and not args.no_extras:
        ctx.set_l3_prefetch(0)
        _, _, per_np, _ = timed(sets, max(4, args.steps // 2), 2, events=True)
        ctx.set_l3_prefetch(-1)
        me_alone_ms = per_np[0][0][vcs_hip.T_ME] / max(1, int(per_np[0][1][vcs_hip.T_ME]))
        timed(sets, 2, 1)      # back to the headline's state (frames of the last GOP searched as the headline left them)

    # what was just timed: vectors of the last GOP (all shards'), static counts
    if group is not None:
        mv_all, ns_all = group.get_mv()
        first0, count0 = shards[0]
        mv0, ns0 = mv_all[first0:first0 + count0], ns_all[first0:first0 + count0]
        last_set = (state["k"] - 1) % sets
        slot0 = last_set * count0
    else:
        last_set = (state["k"] - 1) % sets
        slot0 = last_set * n_total
        mv0, ns0 = ctx.gop_get_mv(slot0, n_total)
        mv_all, ns_all, count0 = mv0, ns0, n_total
    if getattr(group, "partial", False):     # fall-back: only this rank's vectors are here
        searched = 1.0 - float(ns0.sum()) / max(1, len(ns0) * N)
    else:
        searched = 1.0 - float(ns_all.sum()) / max(1, len(ns_all) * N)

    if rank == 0:
        units = n_total * N * args.steps                # macroblocks, whole job
        value = units / dt / 1e3                        # macroblocks per millisecond
        n0 = count0                                     # frames per launch on GPU 0
        ms_step = dt / args.steps * 1e3
        # algorithmic HBM bytes per macroblock: ME reads cur + ref and writes the
        # vectors; the tail reads cur + ref + vectors, writes 3 f64 planes
        per_mb = {vcs_hip.T_ME: (2 * F + 4 * N) / N, vcs_hip.T_ENC_TAIL: (10 * F + 4 * N) / N}
        bits = launches[vcs_hip.T_ME_PREP] > 0     # the bit-plane search ran (k_bits_cur + k_me16b)
        me_name = (("k_me16b" if bs == 16 else "k_me8b") if bits else "k_me16g" if bs == 16 and n0 >= 4
                   else "k_me16r/k_me16" if bs == 16 else "k_me8" if bs == 8 else "k_me_generic")
        names = {vcs_hip.T_ME: me_name, vcs_hip.T_ENC_TAIL: "k_fdct<residual>"}
        # roofline.traffic: PMC bytes from profiles/hbm_traffic.json, but only if that file was collected with
        # THIS build of the library (vcs_build_id: a hash of the kernel sources and flags); else null + stale
        tj = {}
        traffic = {}
        traffic_stale = None
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", "hbm_traffic.json")))
            wl = tj["workload"]
            if (wl["H"], wl["W"], wl["frames_per_launch"], wl["me_bs"]) == (H, W, n0, bs):
                if tj.get("build_id") == vcs_hip.build_id():
                    traffic = {k: v["hbm_bytes_corrected"] for k, v in tj["kernels"].items()}
                else:
                    traffic_stale = {"profiles_build_id": tj.get("build_id"), "library_build_id": vcs_hip.build_id()}
                    tj = {}
        except (OSError, KeyError, ValueError):
            pass

        def roof(slot):
            avg_ms = ms[slot] / max(1, int(launches[slot]))
            alg = per_mb[slot] * n0 * N
            ach = alg / (avg_ms * 1e-3) / 1e9
            r = {"bound": "hbm", "kernel": names[slot], "achieved": round(ach, 1),
                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                 "traffic": traffic.get(names[slot]),
                 "algorithmic_bytes_per_launch": int(alg), "avg_launch_ms": round(avg_ms, 4)}
            if traffic_stale:
                r["traffic_stale"] = True
            if slot == vcs_hip.T_ME and bits:
                # the search's algorithmic bytes (2F + 4N per frame) are the whole stage's: k_bits_cur
                # (current blocks -> bit planes, static test) runs in front of k_me16b
                st_ms = (ms[vcs_hip.T_ME] + ms[vcs_hip.T_ME_PREP]) / max(1, int(launches[slot]))
                st = alg / (st_ms * 1e-3) / 1e9
                r["stage"] = {"kernels": "k_bits_cur + " + me_name, "avg_ms": round(st_ms, 4), "achieved": round(st, 1),
                              "frac": round(st / HBM_PEAK_GBS, 4),
                              "traffic": (traffic.get("k_bits_cur", 0) + traffic.get("k_me16b", 0)) or None}
            return r

        dom = max(per_mb, key=lambda k: ms[k])
        other = vcs_hip.T_ENC_TAIL if dom == vcs_hip.T_ME else vcs_hip.T_ME
        transport = vcs_hip.TRANSPORTS.get(group.transport, "?") if group is not None else None
        if args.strong:
            wl = ("configs[3]: ONE synthetic %dx%d 30-frame GOP, its 29 P frames split over %d GPU(s) "
                  "(%s frames each)" % (W, H, n_gpus, "/".join(str(c) for c in sorted(
                      {vcs_hip.shard_frames(29, n_gpus, r)[1] for r in range(n_gpus)}, reverse=True))))
        else:
            wl = ("configs[2]-shaped: synthetic %dx%d 30-frame GOP (1 I + %d P per GPU)" % (W, H, args.frames))
        wl += (", %dx%d motion blocks, +-%d px stride-%d grid search, 8x8 DCT QF=50, float64 planes; "
               "the timed loop rotates through %d resident copies of the GOP (cold inputs)"
               % (bs, bs, 2 * bs, round(bs / 3), sets))
        out = {
            "metric": "macroblocks/ms (%dx%d, ME+DCT encode) at %s +-%dpx search" % (bs, bs, "4K" if H >= 2160 else "1080p", 2 * bs),
            "value": round(value, 2), "unit": "macroblocks/ms",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True,
            # the no-collective fall-back is not a point of a scaling curve: say so where a driver looks
            "scaling": None if rccl_failed else ("strong" if args.strong else "weak"), "vs_baseline": None,
            "dtype": "u8 (motion search) + f64 (DCT)", "data": "synthetic",
            "config": {"workload": wl,
                       "frames_per_step_gpu0": n0, "frames_per_step_total": n_total,
                       "macroblocks_per_frame": N,
                       "searched_block_fraction": round(searched, 4),
                       "resident_gop_sets_encoded_in_turn": sets,
                       "parallelism": ("P frames sharded over %d GPU(s), %s"
                                       % (n_gpus, "one process per GPU" if (world > 1 or (launched and use_group)) else
                                          "one process, one context per GPU" if n_gpus > 1 else "single context")),
                       "collectives": ("none" if group is None else
                                       "NONE -- FALL-BACK: the RCCL group could not be created (%s); every rank "
                                       "uploaded the I frame itself and kept its vectors, ranks met through "
                                       "files" % rccl_failed if rccl_failed else
                                       "in-library RCCL broadcast (I frame) + all-gather (vectors) per GOP: " + transport)},
            **({"degraded": "no_collectives"} if rccl_failed else {}),
            "clock_conditioning": {"untimed_steps_before_the_warmup": cond_steps,
                                   "why": "the timed steps run at the sustained clock; VCS_BENCH_CONDITION_MS=0 turns it off"},
            "idle_start_first_pass": first_pass,
            "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 4),
            "kernel_ms_per_step": kernel_ms(ms, args.steps),
            "kernel_times_from": {"pass": "a third pass of the same %d steps with HIP events attached to the kernels' "
                                          "dispatch packets (hipExtLaunchKernelGGL); the events cost time, so the "
                                          "headline pass runs without them" % args.steps,
                                  "ms_per_step_with_events": round(dt_ev / args.steps * 1e3, 4)},
            "build_id": vcs_hip.build_id(),
            **({"traffic_stale": traffic_stale} if traffic_stale else {}),
            "roofline": roof(dom),
            "roofline_second_kernel": roof(other),
            # the whole step against HBM: 10F + 4N algorithmic bytes per P frame on GPU 0
            "pipeline_hbm": {"bytes_per_step": int((10 * F + 4 * N) * n0),
                             "gbs": round((10 * F + 4 * N) * n0 / (ms_step * 1e-3) / 1e9, 1),
                             "frac": round((10 * F + 4 * N) * n0 / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "note": "north_star asks for >= 0.40; the search is vector-issue bound (below)"},
        }
        if me_name == "k_me16b":
            # The bit-plane search is bound by vector-ALU issue: 8 v_bitop3_b32 + 1 v_bcnt per 32 byte
            # pairs.  How close it is to that roof, priced with the clock the chip held in this run
            # (longest workgroup lifetime in shader cycles / launch duration).
            me_ms = ms[vcs_hip.T_ME] / max(1, int(launches[vcs_hip.T_ME]))
            valu_per_mb = None      # the PMC count belongs to a build: none for this one -> no fraction
            try:
                kj = tj["kernels"]["k_me16b"]
                valu_per_mb = round(kj["valu_wave_insts_per_dispatch"] / (tj["workload"]["frames_per_launch"] * N), 1)
            except Exception:
                pass
            clock_ghz = clock_ghz_events_pass or None
            core = (8 + 4.5) * 27                # 8 + 4.5 wave passes of 27 instructions per macroblock
            alone = me_alone_ms or me_ms

            def frac(cycles_per_inst, t_ms):
                return (round(valu_per_mb * cycles_per_inst * n0 * N / (1024 * clock_ghz * 1e9 * t_ms * 1e-3), 3)
                        if clock_ghz and valu_per_mb else None)
            out["k_me16b_valu_issue"] = {
                "clock_ghz_measured": round(clock_ghz, 3) if clock_ghz else None,
                "valu_wave_insts_per_macroblock": valu_per_mb,
                "valu_count_from": "profiles/hbm_traffic.json (PMC, this build)" if valu_per_mb else "none: profiles/hbm_traffic.json is of another build",
                "of_which_comparison_core": core,
                "launch_ms": round(me_ms, 4),
                "search_alone_ms": round(alone, 4) if me_alone_ms else None,
                # one vector wave-instruction occupies a SIMD for 4 cycles (SQ_ACTIVE_INST_VALU counts exactly
                # one quad-cycle per instruction in profiles/); tools/ub_bits.hip measures 3.65 for the
                # v_bitop3 / v_bcnt mix when nothing else competes
                "frac": frac(4.0, alone),
                "frac_at_3_65_cycles": frac(3.65, alone),
                "frac_of_whole_launch": frac(4.0, me_ms),
                "how": "vector wave-instructions per macroblock x 4 issue cycles over 1,024 SIMDs x measured clock x "
                       "the search's time without the prefetch workgroups at the launch's end (search_alone_ms: a "
                       "pass with vcs_set_l3_prefetch(0)); frac_of_whole_launch prices the launch as timed in the step"}
        if me_name == "k_me16g":
            # The motion search is bound by vector-ALU issue, not by HBM: how close it is to THAT
            # roof, priced with the clock the chip held in this run (longest workgroup lifetime in
            # shader cycles / launch duration) and per-opcode issue cycles (tools/ub_valu.hip).
            me_ms = ms[vcs_hip.T_ME] / max(1, int(launches[vcs_hip.T_ME]))
            core = 26 * 13                       # row steps x dwords per macroblock: lerp, and, bcnt each
            valu_per_mb = 1142.0
            try:
                kj = tj["kernels"]["k_me16g"]
                valu_per_mb = round(kj["valu_wave_insts_per_dispatch"] / (tj["workload"]["frames_per_launch"] * N), 1)
            except Exception:
                pass
            other_insts = max(0.0, valu_per_mb - 3 * core)
            cyc_per_mb = core * (ISSUE_CYCLES["v_lerp_u8"] + ISSUE_CYCLES["v_and_b32"] +
                                 ISSUE_CYCLES["v_bcnt_u32_b32"]) + other_insts * ISSUE_CYCLES["other"]
            clock_ghz = cyc[0] / (me_ms * 1e-3) / 1e9 if cyc[0] else None
            simds = 256 * 4
            out["k_me16g_valu_issue"] = {
                "clock_ghz_measured": round(clock_ghz, 3) if clock_ghz else None,
                "valu_wave_insts_per_macroblock": valu_per_mb,
                "issue_cycles_per_macroblock": round(cyc_per_mb, 1),
                "frac": (round(cyc_per_mb * n0 * N / (simds * clock_ghz * 1e9 * me_ms * 1e-3), 3)
                         if clock_ghz else None),
                "how": "issue cycles of the kernel's vector instructions (PMC count in profiles/, "
                       "v_lerp_u8 4 / v_and 2 / v_bcnt 4 / other 4 cycles) over 1,024 SIMDs x measured "
                       "clock x launch time"}
        if not args.no_extras:
            try:
                out["hbm_measured_gbs"] = {"copy_read_plus_write": round(ctx.measure_copy_bw(1 << 28, 5), 0),
                                           "fill_write_only": round(ctx.measure_fill_bw(1 << 28, 5), 0)}
            except vcs_hip.VcsError:
                pass
    if not args.no_extras and sets > 1:
        # the loop over ONE resident GOP: its 29 input frames (180 MB) are read again every
        # step and stay in the 256 MB Infinity Cache -- every rank takes part (collectives)
        dt1, _, _, _ = timed(1, args.steps, 2 + cond_steps)   # same clock conditioning as the headline
        _, _, per1, _ = timed(1, args.steps, 2, events=True)
        if rank == 0:
            m1, l1 = per1[0]
            out["resident_one_gop"] = {
                "value": round(n_total * N * args.steps / dt1 / 1e3, 2), "unit": "macroblocks/ms",
                "ms_per_step": round(dt1 / args.steps * 1e3, 4),
                "kernel_ms_per_step": kernel_ms(m1, args.steps),
                "note": "inputs re-read from the Infinity Cache every step; not the headline"}
    if rank == 0:
        if args.with_decode:
            # second line of SURVEY 8d: dequantise + IDCT + colour + MC-add over the same GOP
            ctx.timing_enable(True)
            ctx.timing_reset()
            for k in range(6):                      # the sets in turn: planes read cold, like the encode
                ctx.gop_decode((k % sets) * n0, n0)
            ms2, l2 = ctx.timing()
            ctx.timing_enable(False)
            dec_ms = ms2[vcs_hip.T_DEC] / max(1, int(l2[vcs_hip.T_DEC]))
            alg = (9 * F + 4 * N + F) * n0  # planes + ref + vectors in, frame out
            out["decode"] = {"kernel": "k_idct<residual>", "avg_launch_ms": round(dec_ms, 4),
                             "achieved_gbs": round(alg / (dec_ms * 1e-3) / 1e9, 1),
                             "hbm_frac": round(alg / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "encode_plus_decode_macroblocks_per_ms":
                                 round(n0 * N / (dt / args.steps * 1e3 + dec_ms), 1),
                             "what": "dequantise + IDCT + colour + MC add over GPU 0's %d frames, kernel only "
                                     "(SURVEY 8d, second line); every set was encoded before" % n0}
        if not args.no_extras:
            try:
                out["parity"] = parity_columns(ctx, ref, local_frames[0], mv0, slot0, bs)
            except Exception as e:
                out["parity"] = {"error": repr(e)}
        if not args.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline(H, W, bs, ref, local_frames[0])
        else:
            out["cpu_baseline"] = None
    if group is not None:
        group.barrier()
        group.close()
    for c in ctxs:
        c.close()
    if rank == 0:
        if not args.no_extras and n_gpus == 1 and not args.strong and bs == 16:
            try:
                out["host_path"] = host_path(H, W, bs, ref, local_frames[0], devices[0])
            except Exception as e:
                out["host_path"] = {"error": repr(e)}
            try:
                out["coeff_modes"] = coeff_modes(H, W, bs, ref, local_frames[0], devices[0])
            except Exception as e:
                out["coeff_modes"] = {"error": repr(e)}
            try:
                out["reference_default_point_bs8"] = reference_default_point(H, W, ref, local_frames[0], devices[0])
            except Exception as e:
                out["reference_default_point_bs8"] = {"error": repr(e)}
        if args.group and n_gpus == 1 and not args.strong and not launched:
            try:
                out["group_enqueue_by_local_shards"] = group_enqueue_probe(H, W, bs, ref, local_frames[0], devices[0])
            except Exception as e:
                out["group_enqueue_by_local_shards"] = {"error": repr(e)}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    # The no-collective fall-back is not a measurement of the multi-GPU path: the line above says so
    # (`degraded`, `scaling: null`) and the process leaves with status 3, so that a driver that only looks at
    # exit codes sees it too.  A bring-up thread still inside RCCL cannot be joined: os._exit then.
    sys.stdout.flush()
    sys.stderr.flush()
    if state_stuck:
        os._exit(3 if rccl_failed else 0)
    if rccl_failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
