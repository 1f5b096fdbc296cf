"""PCIe-inclusive rate of the single-frame (host buffer) API at 1080p, for DESIGN.md."""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vcs-h264_amd"), os.path.join(ROOT, "vcs-h264_amd", "InterframeCompression")]
import numpy as np
import vcs_hip
from synth import synth_gop
H, W, n = 1080, 1920, 8
ref, frames = synth_gop(H, W, n)
N = (H // 16) * (W // 16)
with vcs_hip.Context(H, W, 16) as c:
    c.set_ref(ref)
    c.encode_p(frames[0])
    for mode, name in ((0, "float64"), (1, "float32"), (2, "int16")):
        c.set_params(coeff_mode=mode)
        c.encode_p(frames[0])
        t0 = time.perf_counter()
        for k in range(n):
            c.encode_p(frames[k])
        dt = (time.perf_counter() - t0) / n
        print("vcs_encode_p host buffers, %s planes: %.2f ms/frame = %.0f macroblocks/ms" % (name, dt * 1e3, N / dt / 1e3))
    # the same with the frames in page-locked memory (vcs_hip.pinned_empty): DMA at PCIe rate
    pinned = [vcs_hip.pinned_empty(frames[0].shape) for _ in range(n)]
    for k in range(n):
        pinned[k][...] = frames[k]
    c.set_params(coeff_mode=2)
    c.encode_p(pinned[0])
    t0 = time.perf_counter()
    for k in range(n):
        c.encode_p(pinned[k])
    dt = (time.perf_counter() - t0) / n
    print("vcs_encode_p, frames in pinned host memory, int16 planes: %.2f ms/frame = %.0f macroblocks/ms" % (dt * 1e3, N / dt / 1e3))
    c.set_params(coeff_mode=0)
import encoder as E
enc = E.Encoder(pattern=["I"] + ["P"] * 29, shape=[H, W], block_size=16, with_DCT=True, dct_block_size=8)
with contextlib.redirect_stdout(io.StringIO()):
    enc.encode_frame(ref, 0); enc.encode_frame(frames[0], 1)
    t0 = time.perf_counter()
    for k in range(n):
        enc.encode_frame(frames[k], k + 1)
    dt = (time.perf_counter() - t0) / n
print("drop-in Encoder.encode_frame (Python lists, float64): %.2f ms/frame = %.0f macroblocks/ms" % (dt * 1e3, N / dt / 1e3))
import decoder as D
enc = E.Encoder(pattern=["I"] + ["P"] * 29, shape=[H, W], block_size=16, with_DCT=True, dct_block_size=8, resident=True)
with contextlib.redirect_stdout(io.StringIO()):
    enc.encode_frame(ref, 0); enc.encode_frame(frames[0], 1)
    t0 = time.perf_counter()
    for k in range(n):
        enc.encode_frame(frames[k], k + 2)
    dt = (time.perf_counter() - t0) / n
    dec = D.Decoder(encoded_frames=enc.encoded_frames, fps=25.0, shape=[H, W], ref_frames=enc.ref_frames,
                    block_size=16, with_DCT=True, dct_block_size=8, sink=[].append)
    dec._reconstruct_P_frame(enc.encoded_frames[1], True)
    t0 = time.perf_counter()
    for fr in enc.encoded_frames[2:]:
        dec._reconstruct_P_frame(fr, True)
    dd = (time.perf_counter() - t0) / n
print("drop-in Encoder(resident=True).encode_frame: %.2f ms/frame = %.0f macroblocks/ms; Decoder from HBM: %.2f ms/frame" % (dt * 1e3, N / dt / 1e3, dd * 1e3))
enc = E.Encoder(pattern=["I"] + ["P"] * 29, shape=[H, W], block_size=16, with_DCT=True, dct_block_size=8, resident=True)
pref = vcs_hip.pinned_empty(ref.shape); pref[...] = ref
with contextlib.redirect_stdout(io.StringIO()):
    enc.encode_frame(pref, 0); enc.encode_frame(pinned[0], 1)
    t0 = time.perf_counter()
    for k in range(n):
        enc.encode_frame(pinned[k], k + 2)
    dt = (time.perf_counter() - t0) / n
print("drop-in Encoder(resident=True).encode_frame, frames in pinned host memory: %.2f ms/frame = %.0f macroblocks/ms" % (dt * 1e3, N / dt / 1e3))
