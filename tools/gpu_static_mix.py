#!/usr/bin/env python3
"""k_me16g on content with static regions (the reference's early-out, motion.py:109-116): time of the search
kernel for the static deal and for the unit queue.  usage: gpu_static_mix.py [region|random] [fraction]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "vcs-h264_amd")]
import numpy as np
import vcs_hip
from synth import synth_gop

kind = sys.argv[1] if len(sys.argv) > 1 else "region"
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
H, W, n, bs = 1080, 1920, 29, 16
ref, fr = synth_gop(H, W, n, seed=1234, first=1)
rng = np.random.default_rng(5)
nby, nbx = H // bs, W // bs
if kind == "region":
    mask = np.zeros((nby, nbx), bool)
    mask[:, :int(frac * nbx)] = True            # the left part of the picture does not move
else:
    mask = rng.random((nby, nbx)) < frac
big = np.kron(mask, np.ones((bs, bs), bool))
full = np.zeros((H, W), bool)
full[:nby * bs, :nbx * bs] = big
for k in range(n):
    noise = rng.integers(-1, 2, (H, W, 3), dtype=np.int16)
    st = np.clip(ref.astype(np.int16) + noise, 0, 255).astype(np.uint8)
    fr[k][full] = st[full]
ctx = vcs_hip.Context(H, W, bs, 8, device=0)
ctx.gop_reserve(n)
ctx.gop_upload(0, fr)
ctx.set_ref(ref)
for _ in range(3):
    ctx.gop_encode(0, n)
ctx.sync()
ctx.timing_enable(True)
ctx.timing_reset()
for _ in range(10):
    ctx.gop_encode(0, n)
ctx.sync()
ms, l = ctx.timing()
mv, ns = ctx.gop_get_mv(0, n)
print("%s %.2f: static blocks %.3f; search kernel %.4f ms, tail %.4f ms  (DYN=%s SEG=%s)"
      % (kind, frac, ns.sum() / (n * ctx.N), ms[vcs_hip.T_ME] / l[vcs_hip.T_ME], ms[vcs_hip.T_ENC_TAIL] / l[vcs_hip.T_ENC_TAIL],
         os.environ.get("VCS_ME16G_DYN", "default"), os.environ.get("VCS_ME16G_SEG", "default")))
