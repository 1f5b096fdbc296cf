import os, sys, io, contextlib, cProfile, pstats
ROOT = os.getcwd()
sys.path[:0] = [os.path.join(ROOT, "vcs-h264_amd"), os.path.join(ROOT, "vcs-h264_amd", "InterframeCompression")]
import numpy as np, vcs_hip, encoder as E
from synth import synth_gop
H, W = 1080, 1920
ref, frames = synth_gop(H, W, 5)
enc = E.Encoder(pattern=["I"] + ["P"] * 299, shape=[H, W], block_size=16, with_DCT=True, dct_block_size=8, resident=True)
bufs = [enc.next_buffer() for _ in range(5)]
for k in range(5): bufs[k][...] = frames[k]
with contextlib.redirect_stdout(io.StringIO()):
    enc.encode_frame(ref, 0)
    for k in range(1, 20): enc.encode_frame(bufs[k % 5], k)
    enc.flush()
    pr = cProfile.Profile(); pr.enable()
    for k in range(20, 220): enc.encode_frame(bufs[k % 5], k)
    enc.flush()
    pr.disable()
st = pstats.Stats(pr, stream=sys.stderr); st.sort_stats("tottime").print_stats(14)
