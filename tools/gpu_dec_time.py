"""Decode kernel time (k_idct<residual>: dequantise + IDCT + colour + MC add) over a resident 29 x 1080p GOP
for the library named by VCS_HIP_LIB; three resident sets in turn (planes read cold), as bench.py's `decode`.
usage: python tools/gpu_dec_time.py [--frames 29] [--reps 12] [--label x]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vcs-h264_amd"))
import vcs_hip  # noqa: E402
from synth import synth_gop  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=29)
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--sets", type=int, default=3)
ap.add_argument("--label", default="")
a = ap.parse_args()
H, W, n = 1080, 1920, a.frames
ref, frames = synth_gop(H, W, n, seed=1234, first=1)
with vcs_hip.Context(H, W, 16, 8) as c:
    c.set_ref(ref)
    c.gop_reserve(a.sets * n, with_decode=True)
    for s in range(a.sets):
        c.gop_upload(s * n, np.ascontiguousarray(frames))
        c.gop_encode(s * n, n)
    for k in range(2 * a.sets):
        c.gop_decode((k % a.sets) * n, n)
    c.sync()
    import time
    t_end = time.perf_counter() + 0.3      # clock conditioning, as bench.py
    k = 0
    while time.perf_counter() < t_end:
        c.gop_decode((k % a.sets) * n, n)
        k += 1
        if k % 8 == 0:
            c.sync()
    c.sync()
    c.timing_enable(True)
    c.timing_reset()
    for k in range(a.reps):
        c.gop_decode((k % a.sets) * n, n)
    c.sync()
    ms, l = c.timing()
    d = ms[vcs_hip.T_DEC] / max(1, int(l[vcs_hip.T_DEC]))
    dec = c.gop_get_decoded(n - 1)
    alg = 10 * c.F * n + 4 * c.N * n
    print("%-24s k_idct %.4f ms per %d frames  %.0f GB/s  (decoded sum %d)"
          % (a.label or os.environ.get("VCS_HIP_LIB", "product")[-24:], d, n, alg / d / 1e6, int(dec.astype(np.int64).sum())))
