#!/usr/bin/env python3
"""bench.py's host_path three times in one process (is the first measurement paying one-time costs?)."""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from synth import synth_gop
H, W, bs = 1080, 1920, 16
ref, fr = synth_gop(H, W, 29, seed=1234, first=1)
for i in range(3):
    r = bench.host_path(H, W, bs, ref, fr, 0)
    print(i, {k: v for k, v in r.items() if k != "what"}, flush=True)
