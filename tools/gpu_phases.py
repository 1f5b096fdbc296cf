#!/usr/bin/env python3
"""Shader cycles per phase of k_me16g's block loop, from an experiment build (-DVCS_EXP_PHASES,
VCS_HIP_LIB=vcs-h264_amd/lib/exp/phases.so).  Each probe is an s_memtime + s_waitcnt lgkmcnt(0), so the
absolute numbers are inflated; the shares are what is read."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "vcs-h264_amd")]
import numpy as np
import vcs_hip
from synth import synth_gop

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W, n = 1080, 1920, 29
ctx = vcs_hip.Context(H, W, bs, 8, device=0)
ref, fr = synth_gop(H, W, n, seed=1234, first=1)
ctx.gop_reserve(n)
ctx.gop_upload(0, fr)
ctx.set_ref(ref)
ctx.timing_enable(True)
lib = ctx._lib
lib.vcs_me_phases.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
out = np.zeros(10, np.uint64)
for it in range(3):
    ctx.gop_encode(0, n)
    ctx.sync()
    lib.vcs_me_phases(ctx._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
names = ["issue", "barrier", "roll", "static test", "re-pack + grid", "comparisons", "reduce + result", "loop end", "unit prologue"]
tot = float(out[1:].sum())
print("longest workgroup: %d cycles; wave-cycles in the block loop: %.0f M" % (out[0], tot / 1e6))
blocks = n * ctx.N
for k, nm in enumerate(names):
    print("  %-18s %6.1f %%   %7.0f cycles / block" % (nm, 100.0 * float(out[1 + k]) / tot, float(out[1 + k]) / blocks))
