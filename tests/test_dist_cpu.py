"""The N > 1 path on CPU: world_size-2 (and 3) gloo jobs launched exactly as the
driver launches bench.py (torch.distributed.run, 127.0.0.1), with the oracle
standing in for the GPU backend.  Checks the sharding, the I-frame broadcast
and the uneven motion-vector all-gather of gop.DistGopEncoder."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_frames_properties():
    from gop import shard_frames, shard_table
    for n in (0, 1, 5, 29, 30, 64):
        for world in (1, 2, 3, 4, 8):
            table = shard_table(n, world)
            assert sum(c for _, c in table) == n
            pos = 0
            for first, count in table:
                assert first == pos
                pos += count
            counts = [c for _, c in table]
            assert max(counts) - min(counts) <= 1
            assert counts == sorted(counts, reverse=True)
    assert shard_frames(29, 8, 0) == (0, 4) and shard_frames(29, 8, 7) == (26, 3)
    with pytest.raises(ValueError):
        shard_frames(5, 2, 2)


@pytest.mark.parametrize("world,n_frames", [(2, 5), (3, 4)])
def test_dist_gop_gloo(tmp_path, oracle, world, n_frames):
    from synth import synth_gop
    port = 29500 + (os.getpid() + 7 * world) % 400
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "dist_worker.py"),
           str(tmp_path), str(n_frames)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    H, W, bs = 64, 96, 16
    ref, frames = synth_gop(H, W, n_frames, seed=21)
    want = [oracle.encode_p(f, ref, bs) for f in frames]
    seen = 0
    for rank in range(world):
        d = np.load(tmp_path / ("rank%d.npz" % rank))
        # every rank ends up with ALL motion vectors, in GOP order
        assert d["mv"].shape == (n_frames, (H // bs) * (W // bs), 2)
        for k in range(n_frames):
            assert np.array_equal(d["mv"][k], want[k][0]), (rank, k)
            assert d["ns"][k] == want[k][2]
        # ... and keeps the coefficient planes of its own chunk only
        first, count = int(d["first"]), int(d["count"])
        assert d["planes"].shape[0] == count
        for j in range(count):
            assert np.array_equal(d["planes"][j], want[first + j][1])
        seen += count
    assert seen == n_frames
