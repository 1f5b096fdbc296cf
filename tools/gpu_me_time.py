"""Kernel times of one GOP step (search prepass / search / tail) for the library named by
VCS_HIP_LIB (default: the product build).  Three resident copies of a synthetic 1080p GOP in
turn (cold inputs), as bench.py's headline does.
usage: python tools/gpu_me_time.py [--frames 29] [--steps 30] [--H 1080 --W 1920] [--static FRAC]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vcs-h264_amd"))
import vcs_hip  # noqa: E402
from synth import synth_gop  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=29)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--sets", type=int, default=3)
ap.add_argument("--H", type=int, default=1080)
ap.add_argument("--W", type=int, default=1920)
ap.add_argument("--static", type=float, default=0.0, help="fraction of the picture (left part) copied from the reference")
ap.add_argument("--label", default="")
ap.add_argument("--bs", type=int, default=16)
ap.add_argument("--no-events", action="store_true", help="wall time only: no timing events on the kernels")
a = ap.parse_args()

H, W, n = a.H, a.W, a.frames
ref, frames = synth_gop(H, W, n, seed=1234, first=1)
frames = np.ascontiguousarray(frames)
if a.static > 0:
    frames[:, :, : int(W * a.static)] = ref[:, : int(W * a.static)]
# VCS_BENCH_REF_PIPELINE=1: the next I frame is declared early and its tables are cut beside this GOP's search
pipe = os.environ.get("VCS_BENCH_REF_PIPELINE", "0") == "1"


def new_gop(c):
    if pipe:
        c.ref_swap()
        c.ref_next_ready()
    else:
        c.ref_written()


with vcs_hip.Context(H, W, a.bs, 8) as c:
    c.set_ref(ref)
    if pipe:
        c.set_ref_next(ref)
    c.gop_reserve(a.sets * n)
    for s in range(a.sets):
        c.gop_upload(s * n, frames)
    for k in range(max(3, a.sets)):
        new_gop(c)
        c.gop_encode((k % a.sets) * n, n)
    c.sync()
    t_end = time.perf_counter() + 0.25   # clock conditioning, as bench.py
    k = 0
    while time.perf_counter() < t_end:
        new_gop(c)
        c.gop_encode((k % a.sets) * n, n)
        k += 1
        if k % 16 == 0:
            c.sync()
    c.sync()
    c.timing_enable(not a.no_events)
    c.timing_reset()
    t0 = time.perf_counter()
    for k in range(a.steps):
        new_gop(c)
        c.gop_encode((k % a.sets) * n, n)
    c.sync()
    dt = (time.perf_counter() - t0) / a.steps
    ms, l = c.timing()
    per = lambda i: ms[i] / max(1, int(l[i]))
    mv, ns = c.gop_get_mv(0, n)
    print("%-28s step %.4f ms  prep %.4f  search %.4f  tail %.4f  (%.1f k MB/ms)  static %.3f  mv-sum %d"
          % (a.label or os.environ.get("VCS_HIP_LIB", "product")[-28:], dt * 1e3, per(vcs_hip.T_ME_PREP), per(vcs_hip.T_ME),
             per(vcs_hip.T_ENC_TAIL), n * c.N / dt / 1e6, float(np.sum(ns)) / (n * c.N), int(np.abs(mv.astype(np.int64)).sum())))
