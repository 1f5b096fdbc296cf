#!/usr/bin/env python3
"""Where a k_me16b wave's time goes, from an experiment build (make EXPERIMENT=-DVCS_EXP_PHASES
OUT=../../tools/exp/phases.so; VCS_HIP_LIB=tools/exp/phases.so): shader cycles between five s_memtime probes,
summed over the waves of one workgroup octet in 16, against the vector instructions of each part."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "vcs-h264_amd")]
import numpy as np
import vcs_hip
from synth import synth_gop

H, W, n = 1080, 1920, 29
ctx = vcs_hip.Context(H, W, 16, 8, device=0)
ref, fr = synth_gop(H, W, n, seed=1234, first=1)
ctx.gop_reserve(n)
ctx.gop_upload(0, fr)
ctx.set_ref(ref)
lib = ctx._lib
lib.vcs_me16b_phases.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
out = np.zeros(8, np.uint64)
for it in range(6):
    ctx.ref_written()
    ctx.gop_encode(0, n)
ctx.sync()
lib.vcs_me16b_phases(ctx._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
for it in range(4):
    ctx.ref_written()
    ctx.gop_encode(0, n)
ctx.sync()
lib.vcs_me16b_phases(ctx._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
waves = float(out[5])
names = ["prologue (unit, first requests, records)", "first pass (64 candidates, 8 rounds)", "between (re-derive, zero counts, wait LDS-direct)",
         "second pass (4.5 passes)", "merge + arg-min + result"]
valu = [100, 64 * 27 * 29 / 32.0 + 30, 40, 36 * 27.5 * 29 / 32.0 + 5 * 26, 90]   # vector instructions per wave, estimated from the ISA
tot = float(out[:5].sum())
print("%d waves sampled; mean wave lifetime %.0f shader cycles" % (waves, tot / waves))
for k, nm in enumerate(names):
    c = float(out[k]) / waves
    print("  %-52s %5.1f %% of the lifetime  %7.0f cycles  ~%5.0f vector instr  -> %5.1f cycles per instruction" % (nm, 100.0 * float(out[k]) / tot, c, valu[k], c / valu[k]))
