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

if bs == 16:
    nwg = 1024
    t = np.zeros(2 * nwg, np.uint64)
    lib.vcs_me_wg_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    lib.vcs_me_wg_times(ctx._h, t.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nwg)
    t = t.reshape(nwg, 2).astype(np.int64)
    t0 = t[:, 0].min()
    st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0     # microseconds (100 MHz)
    life = en - st
    print("workgroups: start %.1f .. %.1f us; end %.1f .. %.1f us; lifetime mean %.1f, min %.1f, max %.1f us"
          % (st.min(), st.max(), en.min(), en.max(), life.mean(), life.min(), life.max()))
    print("wave-slot occupancy over the launch: %.3f" % (life.sum() / (nwg * en.max())))
    for x in range(8):
        m = np.arange(nwg) % 8 == x
        wi = np.arange(nwg)[m] // 8
        four = wi < 96
        print("  XCD %d: end of 4-unit workgroups %.1f (mean) %.1f (max), of 3-unit workgroups %.1f (mean) %.1f (max)"
              % (x, en[m][four].mean(), en[m][four].max(), en[m][~four].mean(), en[m][~four].max()))
    h, edges = np.histogram(en, bins=12)
    print("end-time histogram (us):", [int(e) for e in edges], list(h))

    hw = np.zeros(8 * nwg, np.uint64)
    lib.vcs_me_wg_hw.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    lib.vcs_me_wg_hw(ctx._h, hw.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nwg)
    hw = hw.reshape(nwg, 8)[:, :4]
    hid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
    simd = (hid >> 4) & 3
    cu = (hid >> 8) & 15
    sh = (hid >> 12) & 1
    se = (hid >> 13) & 7
    print("XCC_ID of workgroup b vs b % 8 agree for", int((xcc[:, 0] == np.arange(nwg) % 8).sum()), "of", nwg)
    cuid = xcc * 1000 + se * 100 + sh * 50 + cu           # a label per physical CU
    same_cu = (cuid == cuid[:, :1]).all(axis=1)
    print("workgroups whose 4 waves share a CU:", int(same_cu.sum()))
    nd = np.array([len(set(simd[b])) for b in range(nwg)])
    for k in range(1, 5):
        m = nd == k
        if m.any():
            print("  waves on %d distinct SIMDs: %4d workgroups, lifetime mean %.1f us, end mean %.1f us"
                  % (k, int(m.sum()), life[m].mean(), en[m].mean()))
    per_cu = {}
    for b in range(nwg):
        per_cu.setdefault(int(cuid[b, 0]), []).append(b)
    cnt = np.array([len(v) for v in per_cu.values()])
    print("physical CUs seen: %d; workgroups per CU: min %d max %d" % (len(per_cu), cnt.min(), cnt.max()))
    # waves per (CU, SIMD)
    occ = {}
    for b in range(nwg):
        for w in range(4):
            occ[(int(cuid[b, w]), int(simd[b, w]))] = occ.get((int(cuid[b, w]), int(simd[b, w])), 0) + 1
    oc = np.array(list(occ.values()))
    print("waves per SIMD: min %d max %d, histogram" % (oc.min(), oc.max()), np.bincount(oc))
    late = en > np.percentile(en, 75)
    print("latest quarter of workgroups: distinct-SIMD histogram", np.bincount(nd[late], minlength=5)[1:],
          " all:", np.bincount(nd, minlength=5)[1:])
    print("latest quarter by blockIdx quartile:", np.bincount((np.arange(nwg)[late] * 4) // nwg, minlength=4))
