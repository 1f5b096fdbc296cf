#!/usr/bin/env python3
"""bench.py -- macroblocks/s through the inter-frame encode hot path on MI355X.

Workload (BASELINE.json metric: "macroblocks/sec (ME+DCT) at 1080p +-32px
search"): a synthetic 1080p 30-frame GOP (1 I + 29 P frames, synth_gop), 16x16
motion blocks -> +-32 px window on the reference's stride-5 grid, 8x8 DCT with
the QF=50 tables, float64 coefficient planes.  One STEP = one pass of the hot
path over the 29 P frames of the GOP: the motion-search kernel followed by the
fused MC + residual + colour + DCT + quantise kernel, with every input already
resident in HBM and every output left in HBM.

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling --
one long GOP with 29*N P frames, rank r owns P frames [29r, 29r+29).  The only
exchanges of the path are in the timed step: an RCCL broadcast of the I frame
from rank 0 and an all-gather of the motion vectors; coefficient planes stay on
the GPU that produced them.  PyTorch is used for that plumbing only.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "vcs-h264_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 measured copy


class _DevBytes:
    """Zero-copy view of a libvcs_hip HBM buffer for torch (RCCL plumbing)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1",
                                         "data": (ptr, False), "version": 2}


def cpu_baseline(H, W, bs, ref, frames, budget_s=12.0):
    """The CPU oracle (our C restatement, `kind: port`) on the same workload, all
    host cores via OpenMP, on a bounded sample of the GOP's P frames."""
    from oracle import oracle
    # the GPU box grants 16 host cores per GPU; never oversubscribe past the affinity mask
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    oracle.set_threads(cores)
    oracle.encode_p(frames[0][:64, :64].copy(), ref[:64, :64].copy(), bs)  # warm (build/load)
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < budget_s:   # cycle through the GOP until the budget is used
        oracle.encode_p(frames[done % len(frames)], ref, bs)
        done += 1
    dt = time.perf_counter() - t0
    N = (H // bs) * (W // bs)
    return {"value": round(done * N / dt / 1e3, 3), "unit": "macroblocks/ms", "cores": cores,
            "kind": "port",
            "sample": "%d P-frame encodes cycling through the GOP (%dx%d, ME %dx%d + 8x8 DCT), "
                      "oracle/vcs_oracle.c with OpenMP, %.1f s" % (done, W, H, bs, bs, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--bs", type=int, default=16, help="motion block size (window = 2*bs)")
    ap.add_argument("--frames", type=int, default=29, help="P frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-decode", action="store_true",
                    help="also time the decode kernel over the GOP (reported next to the line)")
    ap.add_argument("--same-ref", action="store_true",
                    help="diagnostic: do not declare a new I frame per step (the per-reference sum "
                         "table is then built once, outside the timed region)")
    ap.add_argument("--slot-sets", type=int, default=1,
                    help="diagnostic (single GPU): K resident copies of the GOP in K sets of slots, "
                         "encoded in turn, so that consecutive steps neither read the same input "
                         "frames nor write the same output planes (default 1: one resident GOP)")
    ap.add_argument("--rotate-sets", type=int, default=3,
                    help="single GPU: after the main measurement, repeat it rotating through this many "
                         "resident copies of the GOP (inputs no longer in the 256 MB Infinity Cache "
                         "from the previous step) and report it as `rotating_inputs`; 0 = skip")
    ap.add_argument("--serial-comm", action="store_true",
                    help="multi-GPU: broadcast -> kernels -> all-gather strictly in sequence "
                         "instead of overlapping the collectives with the neighbouring GOPs' kernels")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the torch.distributed (RCCL) code path even with one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.steps = max(1, args.steps)
    args.warmup = max(0, args.warmup)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a
    # version banner when the process group comes up), so file descriptor 1 is pointed at
    # stderr for the whole run and the line goes to a private copy of the original stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    dist = torch = None
    use_dist = world > 1 or (args.force_dist and "RANK" in os.environ)
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import vcs_hip
    from synth import synth_gop

    H, W, bs, n = args.height, args.width, args.bs, args.frames
    ctx = vcs_hip.Context(H, W, bs, 8, device=local_rank)
    N, F = ctx.N, ctx.F
    # this rank's slice of the (long) GOP; synthetic, deterministic
    ref, frames = synth_gop(H, W, n, seed=1234, first=1 + rank * n)
    overlap = use_dist and not args.serial_comm
    sets = max(1, args.slot_sets) if not use_dist else 1
    rot = max(0, args.rotate_sets) if (not use_dist and sets == 1) else 0
    ctx.gop_reserve(max(sets, rot) * n, with_decode=args.with_decode)
    for k in range(max(sets, rot)):
        ctx.gop_upload(k * n, frames)
    set_state = {"k": 0}
    if use_dist:
        if rank == 0:
            ctx.set_ref(ref)        # the I frame enters HBM on rank 0 only
        ctx.ref_written()
        ptr, nb = ctx.dev_ptr(1)
        t_mv = torch.as_tensor(_DevBytes(ptr, nb), device="cuda")
        ptr, nb = ctx.dev_ptr(0)
        t_ref = torch.as_tensor(_DevBytes(ptr, nb), device="cuda")
        if overlap:
            # Collectives on their own stream, kernels on the context's stream, ordered by
            # events only.  GOP k uses reference buffer k & 1: the I frame of GOP k+1 is
            # broadcast into the other one while GOP k is being encoded (vcs_ref_swap then makes
            # it current, no copy).  GOP k's motion vectors are copied aside behind its kernels
            # (1 MB) and gathered from there while GOP k+1 is encoded into the same slots (writing
            # the coefficient planes of alternate GOPs to alternate HBM regions costs the tail
            # kernel 35 %, measured; so the slots stay put).
            # Every rank enqueues the collectives in the same order:
            # bcast(0), bcast(1), gather(0), bcast(2), gather(1), ...
            try:   # plumbing that only exists in this mode; the same on every rank
                ptr, nb = ctx.dev_ptr(3)
                t_refs = [t_ref, torch.as_tensor(_DevBytes(ptr, nb), device="cuda")]
                ctx.sync()
                t_refs[1].copy_(t_refs[0])           # rank 0: both hold the decoder's output
                xs = torch.cuda.ExternalStream(ctx.stream_handle(),
                                               device=torch.device("cuda", local_rank))
                cs = torch.cuda.Stream()
            except Exception as e:                   # fall back to the sequential form
                print("bench.py: overlapped collectives unavailable (%r), using --serial-comm" % (e,),
                      file=sys.stderr)
                overlap = False
        if overlap:
            mv_stage = [torch.empty_like(t_mv), torch.empty_like(t_mv)]
            t_all = [torch.empty(world * t_mv.numel(), dtype=torch.uint8, device="cuda") for _ in range(2)]
            ev_b = [torch.cuda.Event(), torch.cuda.Event()]   # broadcast into reference buffer i done (cs)
            ev_c = [torch.cuda.Event(), torch.cuda.Event()]   # kernels of GOP k (k & 1 = i) + copy done (xs)
            torch.cuda.synchronize()
            state = {"k": 0}

            def issue_bcast(k):
                i = k & 1
                with torch.cuda.stream(cs):
                    cs.wait_event(ev_c[i])           # GOP k-2 no longer reads this reference buffer
                    dist.broadcast(t_refs[i], src=0)
                    ev_b[i].record(cs)

            issue_bcast(0)
        else:
            ctx.sync()
            t_all = torch.empty(world * t_mv.numel(), dtype=torch.uint8, device="cuda")
    else:
        ctx.set_ref(ref)

    def step():
        if overlap:
            k = state["k"]
            i = k & 1
            state["k"] = k + 1
            issue_bcast(k + 1)                       # next GOP's I frame, beside this GOP's kernels
            # One wait per GOP, on an event that was signalled long ago (during the previous GOP's
            # kernels): this GOP's I frame has arrived -- and, since the collectives run in order
            # on their stream and gather(k-2) was enqueued before bcast(k), GOP k-2's vectors
            # have left mv_stage[i].  (A wait on something that has only just finished costs the
            # kernels' stream ~20 us of queue latency: tried with the vector copy on the other
            # stream.)
            xs.wait_event(ev_b[i])
            if k:
                ctx.ref_swap()                       # buffer i becomes the reference (new sum table)
            else:
                ctx.ref_written()
            ctx.gop_encode(0, n)                     # ME kernel + fused tail kernel
            with torch.cuda.stream(xs):
                mv_stage[i].copy_(t_mv, non_blocking=True)   # 1 MB aside: GOP k+1 reuses the slots
                ev_c[i].record(xs)
            with torch.cuda.stream(cs):
                cs.wait_event(ev_c[i])
                dist.all_gather_into_tensor(t_all[i], mv_stage[i])   # motion-vector collection
            return
        if use_dist:
            ts = torch.cuda.current_stream().cuda_stream
            dist.broadcast(t_ref, src=0)             # I frame over xGMI (after the previous
            ctx.stream_wait(ts)                      # step's all-gather; our stream waits for it)
        # every step is a new GOP: the per-reference work (k_me16's sum table) is
        # redone inside the timed region, once per GOP as in a real encode
        if not args.same_ref:
            ctx.ref_written()
        if sets > 1:
            set_state["k"] = (set_state["k"] + 1) % sets
        ctx.gop_encode(set_state["k"] * n, n)        # ME kernel + fused tail kernel
        if use_dist:
            ctx.stream_signal(ts)                    # collectives wait for the kernels, GPU-side
            dist.all_gather_into_tensor(t_all, t_mv)  # motion-vector collection

    def fence():
        ctx.sync()
        if use_dist:
            torch.cuda.synchronize()                 # all torch streams, the collectives' included
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.timing_enable(not os.environ.get("VCS_BENCH_NO_EVENTS"))   # diagnostic: cost of the event brackets
    ctx.timing_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0                 # host time to enqueue the K steps
    fence()
    dt = time.perf_counter() - t0
    ms, launches = ctx.timing()
    ctx.timing_enable(False)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # parity guard on what was just timed: searched fraction + MV checksum
    mv, ns = ctx.gop_get_mv(0, n)
    searched = 1.0 - float(ns.sum()) / max(1, n * N)

    if rank == 0:
        units = world * n * N * args.steps              # macroblocks, whole job
        value = units / dt / 1e3                        # macroblocks per millisecond
        # algorithmic HBM bytes per macroblock: ME reads cur + ref and
        # writes the vectors; the tail reads cur + ref + vectors, writes 3 f64 planes
        per_mb = {vcs_hip.T_ME: (2 * F + 4 * N) / N, vcs_hip.T_ENC_TAIL: (10 * F + 4 * N) / N}
        names = {vcs_hip.T_ME: "k_me16r" if bs == 16 else "k_me8" if bs == 8 else "k_me_generic",
                 vcs_hip.T_ENC_TAIL: "k_fdct<residual>"}
        # PMC-measured HBM bytes per launch of the same workload (profiles/hbm_traffic.json,
        # produced by profiles/collect.sh; FETCH_SIZE doubled per the gfx950 correction)
        traffic = {}
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", "hbm_traffic.json")))
            wl = tj["workload"]
            if (wl["H"], wl["W"], wl["frames_per_launch"], wl["me_bs"]) == (H, W, n, bs):
                traffic = {k: v["hbm_bytes_corrected"] for k, v in tj["kernels"].items()}
        except (OSError, KeyError, ValueError):
            pass

        def roof(slot):
            avg_ms = ms[slot] / max(1, int(launches[slot]))
            alg = per_mb[slot] * n * N
            ach = alg / (avg_ms * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": names[slot], "achieved": round(ach, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic.get(names[slot]),
                    "algorithmic_bytes_per_launch": int(alg), "avg_launch_ms": round(avg_ms, 4)}

        dom = max(per_mb, key=lambda k: ms[k])
        other = vcs_hip.T_ENC_TAIL if dom == vcs_hip.T_ME else vcs_hip.T_ME
        out = {
            "metric": "macroblocks/ms (16x16, ME+DCT encode) at 1080p +-32px search",
            "value": round(value, 2), "unit": "macroblocks/ms",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (motion search) + f64 (DCT)", "data": "synthetic",
            "config": {"workload": "configs[2]-shaped: synthetic %dx%d 30-frame GOP (1 I + %d P per "
                                   "GPU), %dx%d motion blocks, +-%d px stride-%d grid search, 8x8 "
                                   "DCT QF=50, float64 planes" % (W, H, n, bs, bs, 2 * bs,
                                                                  round(bs / 3)),
                       "frames_per_step_per_gpu": n, "macroblocks_per_frame": N,
                       "searched_block_fraction": round(searched, 4),
                       "resident_gop_sets_encoded_in_turn": sets,
                       "parallelism": "P frames sharded over %d GPU(s)" % world,
                       "collectives": ("none" if not use_dist else
                                       "RCCL broadcast (I frame) + all-gather (vectors) per GOP, " +
                                       ("overlapped with the neighbouring GOPs' kernels" if overlap
                                        else "in sequence with the kernels"))},
            "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 4),
            "kernel_ms_per_step": {"me": round(ms[vcs_hip.T_ME] / args.steps, 4),
                                   "enc_tail": round(ms[vcs_hip.T_ENC_TAIL] / args.steps, 4)},
            "roofline": roof(dom),
            "roofline_second_kernel": roof(other),
        }
        if names[vcs_hip.T_ME] == "k_me16r":
            # the motion search is bound by vector-ALU issue, not by HBM: say how close it
            # is to THAT roof (wave-instructions per launch from profiles/r01 PMC, 16-lane
            # SIMDs issue one wave64 instruction per 4 cycles)
            valu_per_mb = 1216.0
            try:  # the committed PMC count for this workload, if present
                kj = tj["kernels"]["k_me16r"]
                valu_per_mb = round(kj["valu_wave_insts_per_dispatch"] /
                                    (tj["workload"]["frames_per_launch"] * N), 1)
            except Exception:
                pass
            peak = 256 * 4 * 2.4e9 / 4
            me_ms = ms[vcs_hip.T_ME] / max(1, int(launches[vcs_hip.T_ME]))
            ach = valu_per_mb * n * N / (me_ms * 1e-3)
            out["k_me16r_valu_issue"] = {"achieved_ginst_s": round(ach / 1e9, 1),
                                        "peak_ginst_s": round(peak / 1e9, 1),
                                        "frac": round(ach / peak, 3),
                                        "valu_wave_insts_per_macroblock": valu_per_mb}
        try:
            out["hbm_measured_gbs"] = {"copy_read_plus_write": round(ctx.measure_copy_bw(1 << 28, 5), 0),
                                       "fill_write_only": round(ctx.measure_fill_bw(1 << 28, 5), 0)}
        except vcs_hip.VcsError:
            pass
        if rot > 1:
            # The main loop re-encodes ONE resident GOP: its 29 input frames (180 MB) are read
            # again every step and stay in the 256 MB Infinity Cache (the planes are written
            # with non-temporal stores and do not evict them).  Same measurement, rotating
            # through `rot` resident copies so that a step's inputs were last touched rot
            # steps ago: what a stream of GOPs costs when nothing comes from that cache.
            def rstep(k):
                ctx.ref_written()
                ctx.gop_encode((k % rot) * n, n)
            for k in range(rot):
                rstep(k)
            ctx.sync()
            ctx.timing_enable(True)
            ctx.timing_reset()
            t1 = time.perf_counter()
            for k in range(args.steps):
                rstep(k)
            ctx.sync()
            dtr = time.perf_counter() - t1
            msr, lr = ctx.timing()
            ctx.timing_enable(False)
            out["rotating_inputs"] = {
                "resident_gop_sets": rot, "value": round(n * N * args.steps / dtr / 1e3, 2),
                "unit": "macroblocks/ms", "ms_per_step": round(dtr / args.steps * 1e3, 4),
                "kernel_ms_per_step": {"me": round(msr[vcs_hip.T_ME] / args.steps, 4),
                                       "enc_tail": round(msr[vcs_hip.T_ENC_TAIL] / args.steps, 4)}}
        if args.with_decode:
            # second line of SURVEY 8d: dequantise + IDCT + colour + MC-add over the same GOP
            ctx.timing_enable(True)
            ctx.timing_reset()
            for _ in range(5):
                ctx.gop_decode(0, n)
            ms2, l2 = ctx.timing()
            ctx.timing_enable(False)
            dec_ms = ms2[vcs_hip.T_DEC] / max(1, int(l2[vcs_hip.T_DEC]))
            alg = (9 * F + 4 * N + F) * n  # planes + ref + vectors in, frame out
            out["decode"] = {"kernel": "k_idct<residual>", "avg_launch_ms": round(dec_ms, 4),
                             "achieved_gbs": round(alg / (dec_ms * 1e-3) / 1e9, 1),
                             "encode_plus_decode_macroblocks_per_ms":
                                 round(n * N / (dt / args.steps * 1e3 + dec_ms), 1)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(H, W, bs, ref, frames)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
