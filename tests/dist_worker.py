"""Worker for tests/test_dist_cpu.py: one rank of a world_size-N gloo job that
runs gop.DistGopEncoder with an ORACLE-backed backend (test-only injection; the
product's default backend is the HIP context)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "vcs-h264_amd")]


class OracleBackend:
    """set_ref / encode / planes, computed by the CPU oracle (stands in for a GPU context)."""

    def __init__(self, H, W, me_bs):
        from oracle import oracle
        self.o, self.me_bs = oracle, me_bs
        self.N = (H // me_bs) * (W // me_bs)
        self.ref = None
        self._planes = []

    def set_ref(self, ref):
        self.ref = np.array(ref, copy=True)

    def encode(self, frames):
        mvs, nss, self._planes = [], [], []
        for f in frames:
            mv, planes, ns = self.o.encode_p(f, self.ref, self.me_bs)
            mvs.append(mv); nss.append(ns); self._planes.append(planes)
        if not mvs:
            return np.zeros((0, self.N, 2), np.int16), np.zeros(0, np.uint32)
        return np.stack(mvs), np.array(nss, np.uint32)

    def planes(self, slot):
        return self._planes[slot]


def main():
    out_dir, n_frames = sys.argv[1], int(sys.argv[2])
    import torch.distributed as dist
    from gop import DistGopEncoder, shard_frames
    from synth import synth_gop
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    H, W, bs = 64, 96, 16
    ref, frames = synth_gop(H, W, n_frames, seed=21)
    first, count = shard_frames(n_frames, world, rank)
    enc = DistGopEncoder(H, W, bs, n_frames=n_frames, backend=OracleBackend(H, W, bs))
    # only the source rank owns the I frame; the others must get it by broadcast
    mv_all, ns_all = enc.encode_gop(ref if rank == 0 else None, frames[first:first + count])
    local = [enc.backend.planes(k) for k in range(count)]
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), mv=mv_all, ns=ns_all, first=first,
             count=count, planes=np.stack(local) if local else np.zeros((0, 3, H, W)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
