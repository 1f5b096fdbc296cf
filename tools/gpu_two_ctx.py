"""Two independent encode pipelines (two contexts = two streams) on one GPU against one: does the chip
overlap the vector-bound search of one GOP with the memory-bound kernels of another?
usage: python tools/gpu_two_ctx.py [--frames 29] [--steps 40]"""
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
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--sets", type=int, default=3)
ap.add_argument("--offset", type=int, default=0, help="1: the second pipeline starts half a step late (one extra search first)")
a = ap.parse_args()
H, W, n = 1080, 1920, a.frames
ref, frames = synth_gop(H, W, n, seed=1234, first=1)
frames = np.ascontiguousarray(frames)


def make():
    c = vcs_hip.Context(H, W, 16, 8)
    c.set_ref(ref)
    c.gop_reserve(a.sets * n)
    for s in range(a.sets):
        c.gop_upload(s * n, frames)
    return c


def run(ctxs, steps):
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        for c in ctxs:
            c.ref_written()
            c.gop_encode((k % a.sets) * n, n)
    for c in ctxs:
        c.sync()
    return (time.perf_counter() - t0) / steps


cs = [make(), make()]
for c in cs:
    for k in range(3):
        c.ref_written(); c.gop_encode(k % a.sets * n, n)
    c.sync()
run(cs, 200)            # clock conditioning
one = run(cs[:1], a.steps)
two = run(cs, a.steps)
N = cs[0].N
print("one pipeline : %.4f ms per GOP  (%.1f k MB/ms)" % (one * 1e3, n * N / one / 1e6))
print("two pipelines: %.4f ms per pair of GOPs = %.4f per GOP  (%.1f k MB/ms)" % (two * 1e3, two * 1e3 / 2, 2 * n * N / two / 1e6))
for c in cs:
    c.close()
