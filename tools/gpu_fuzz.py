"""Randomised parity sweep of the batched motion search against the oracle (GPU box): random picture
sizes (ragged edges included), block sizes 8 / 16, batch sizes 1..40, thresholds, and content from
"every block moves" to "nearly everything static" (a quarter of the batches also checks one frame's
coefficient planes, another quarter one decoded frame), with the bit-plane search forced on every launch
(VCS_ME16_BITS=2 VCS_ME8_BITS=2) or left to the default selection.
usage: python tools/gpu_fuzz.py [--cases 120] [--seed 0] [--max-h 400] [--max-w 640]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "vcs-h264_amd"))
sys.path.insert(0, ROOT)
import vcs_hip  # noqa: E402
from oracle import oracle  # noqa: E402  (the checker; this is a test tool)
from synth import synth_gop  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=120)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--max-h", type=int, default=400)
ap.add_argument("--max-w", type=int, default=640)
a = ap.parse_args()
oracle.build()
rng = np.random.default_rng(a.seed)
t0 = time.time()
bad = 0
for case in range(a.cases):
    bs = int(rng.choice([8, 16]))
    H = int(rng.integers(bs, a.max_h + 1))
    W = int(rng.integers(bs, a.max_w + 1))
    if rng.random() < 0.3:                       # whole blocks only, sometimes
        H, W = max(bs, H // bs * bs), max(bs, W // bs * bs)
    batch = rng.random() < 0.8                   # the batch entry point (its tail needs multiples of 8) ...
    if batch:
        H, W = max(bs, H // 8 * 8), max(bs, W // 8 * 8)
    n = int(rng.integers(1, 41)) if batch else int(rng.integers(1, 4))   # ... or single-frame calls, any size
    thresh = int(rng.choice([2000, 2000, 0, 300, 60000]))
    kind = int(rng.integers(0, 4))
    ref, frames = synth_gop(H, W, n, seed=int(rng.integers(1 << 30)))
    frames = np.ascontiguousarray(frames)
    if kind == 1:                                # noise: no structure, many ties impossible, costs large
        ref = rng.integers(0, 256, ref.shape, dtype=np.uint8)
        frames = rng.integers(0, 256, frames.shape, dtype=np.uint8)
    elif kind == 2:                              # near copies of the reference: static blocks, ties
        for k in range(n):
            d = rng.integers(-1, 2, ref.shape) * (rng.random(ref.shape) < rng.random() * 0.2)
            frames[k] = np.clip(ref.astype(np.int16) + d, 0, 255).astype(np.uint8)
    elif kind == 3:                              # flat areas: equal costs over many candidates (first wins)
        ref[: H // 2] = 100
        frames[:, : H // 2] = 100 + (np.arange(n) % 3)[:, None, None, None]
    with vcs_hip.Context(H, W, bs) as c:
        c.set_params(sim_thresh=thresh)
        c.set_ref(ref)
        if batch:
            dec = case % 4 == 1                  # a quarter of the cases also decode the batch
            c.gop_reserve(n, with_decode=dec)
            c.gop_upload(0, frames)
            c.gop_encode(0, n)
            mv, ns = c.gop_get_mv(0, n)
            if dec:                              # k_idct: one decoded frame against the oracle's decoder
                c.gop_decode(0, n)
                k = int(rng.integers(0, n))
                omv, _ = oracle.me(frames[k], ref, bs, thresh=thresh)
                _, oplanes, _ = oracle.encode_p(frames[k], ref, bs, thresh=thresh)
                if not np.array_equal(c.gop_get_decoded(k), oracle.decode_p(omv, oplanes, ref, bs)):
                    print("DECODE MISMATCH case %d frame %d: H %d W %d bs %d n %d thresh %d kind %d" % (case, k, H, W, bs, n, thresh, kind), flush=True)
                    bad += 1
            if case % 4 == 0:                    # the tail behind the search: one frame's coefficient planes
                k = int(rng.integers(0, n))
                _, oplanes, _ = oracle.encode_p(frames[k], ref, bs, thresh=thresh)
                if not np.array_equal(c.gop_get_planes(k), oplanes):
                    print("PLANES MISMATCH case %d frame %d: H %d W %d bs %d n %d thresh %d kind %d" % (case, k, H, W, bs, n, thresh, kind), flush=True)
                    bad += 1
        else:
            got = [c.me(frames[k]) for k in range(n)]
            mv, ns = [g[0] for g in got], [g[1] for g in got]
    ok = True
    for k in range(n):
        omv, ons = oracle.me(frames[k], ref, bs, thresh=thresh)
        if not (np.array_equal(mv[k], omv) and int(ns[k]) == ons):
            ok = False
            print("MISMATCH case %d frame %d: H %d W %d bs %d n %d thresh %d kind %d  (%d vectors differ, static %d vs %d)"
                  % (case, k, H, W, bs, n, thresh, kind, int((mv[k] != omv).any(axis=1).sum()), int(ns[k]), ons), flush=True)
            break
    bad += not ok
    if case % 20 == 19:
        print("%d cases, %d bad, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
print("fuzz: %d cases, %d mismatching (seed %d)" % (a.cases, bad, a.seed))
sys.exit(1 if bad else 0)
