"""Where the pipelined feeder's time per frame goes (1080p, 16x16 blocks, planes resident):
raw vcs_encode_p_async / vcs_wait from Python with K slots, the H2D copy alone, the kernels alone."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vcs-h264_amd"), os.path.join(ROOT, "vcs-h264_amd", "InterframeCompression")]
import numpy as np
import vcs_hip
from synth import synth_gop
H, W, n = 1080, 1920, 6
ref, frames = synth_gop(H, W, n)
pinned = [vcs_hip.pinned_empty(frames[0].shape) for _ in range(n)]
for k in range(n):
    pinned[k][...] = frames[k]
N = (H // 16) * (W // 16)
reps = 60
for K in (1, 2, 3, 4, 6):
    with vcs_hip.Context(H, W, 16) as c:
        c.set_ref(ref)
        c.async_reserve(K)
        keep = []   # handles stay alive: freeing one synchronises the stream
        def run(src):
            busy = [False] * K
            t0 = time.perf_counter()
            for k in range(reps):
                s = k % K
                if busy[s]:
                    c.wait(s)
                keep.append(c.encode_p_async(s, src[k % n]))
                busy[s] = True
            for s in range(K):
                if busy[s]:
                    c.wait(s)
            return (time.perf_counter() - t0) / reps
        run(pinned)
        dp = run(pinned)
        dg = run(list(frames))
        print("K=%d slots: %.3f ms/frame pinned (%.0f macroblocks/ms), %.3f ms/frame pageable" % (K, dp * 1e3, N / dp / 1e3, dg * 1e3))
with vcs_hip.Context(H, W, 16) as c:
    c.set_ref(ref)
    c.gop_reserve(n)
    t0 = time.perf_counter()
    for _ in range(10):
        c.gop_upload(0, np.stack(pinned))          # pageable stack: not the point
    t1 = time.perf_counter()
    c.gop_upload(0, frames)
    c.timing_enable(True); c.timing_reset()
    for k in range(30):
        c.gop_encode(k % n, 1)
    c.sync()
    ms, l = c.timing()
    print("kernels per single frame: ME %.1f us, tail %.1f us" % (ms[0] / l[0] * 1e3, ms[1] / l[1] * 1e3))
    lib = c._lib
    hip = ctypes.CDLL("libamdhip64.so")
    import numpy as np
    # raw H2D of one pinned frame
    p, nb = c.dev_ptr(0)
    t0 = time.perf_counter()
    for k in range(50):
        hip.hipMemcpy(ctypes.c_void_p(p), ctypes.c_void_p(pinned[k % n].ctypes.data), ctypes.c_size_t(c.F), 1)
    dt = (time.perf_counter() - t0) / 50
    print("hipMemcpy H2D of one pinned 1080p frame: %.3f ms = %.1f GB/s" % (dt * 1e3, c.F / dt / 1e9))
