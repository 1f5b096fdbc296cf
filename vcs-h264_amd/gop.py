"""GOP-level host layer: batched, HBM-resident encode of the P frames of a GOP
and its multi-GPU form.

Why the path shards: every P frame is predicted from the RAW I frame of its
GOP, never from another P frame or from reconstructed data (reference
encoder.py:51-52), so the P frames of a GOP are mutually independent.  They
are dealt to the shards in contiguous chunks; the only exchanges are
  1. a broadcast of the I frame from the shard that holds it (F bytes), and
  2. an all-gather of the motion vectors (4*N bytes per frame).
Coefficient planes (8*F bytes per frame) never cross xGMI: each GPU keeps /
downloads its own.  There is no reduction anywhere.

The product path is libvcs_hip.so end to end: the RCCL communicators, the
broadcast, the all-gather and their overlap with the kernels live inside the
library (vcs_group_*, include/vcs_hip.h).  Two ways to run it:

  * one process, one context per GPU:   GopEncoder(H, W, devices=[0, 1, ...])
  * one process per GPU (torch.distributed.run or any launcher that sets
    RANK / WORLD_SIZE / LOCAL_RANK):    DistGopEncoder(H, W, ...)
    -- no PyTorch involved: rank 0's RCCL id travels through a file
    (vcs_hip.rendezvous_unique_id), or through torch.distributed's store if
    the caller has already initialised a process group.

For CPU-only tests of the sharding / gather logic a `backend` object can be
injected into DistGopEncoder (tests use the oracle there, over gloo); then --
and only then -- torch.distributed carries the exchanges.
N > 1 on real GPUs is unmeasured so far (the build environment leases one GPU).
"""
import os

import numpy as np

import vcs_hip

shard_frames = vcs_hip.shard_frames


def shard_table(n_frames, world):
    return [shard_frames(n_frames, world, r) for r in range(world)]


class GopEncoder:
    """Single process, one context per device.  encode_gop(ref, frames) returns the
    motion vectors / static counts of all P frames; planes(k) downloads the
    coefficient planes of P frame k from the GPU that holds them."""

    def __init__(self, H, W, me_bs=16, devices=(0,), Q=None, coeff_mode=vcs_hip.VCS_COEFF_F64):
        self.contexts = [vcs_hip.Context(H, W, me_bs, 8, d) for d in devices]
        for c in self.contexts:
            if Q is not None or coeff_mode != vcs_hip.VCS_COEFF_F64:
                c.set_params(Q=Q, coeff_mode=coeff_mode)
        self.group = vcs_hip.Group(self.contexts)
        self.N = self.contexts[0].N
        self.n_p = None

    def encode_gop(self, ref, frames, with_decode=False):
        frames = np.asarray(frames)
        if self.n_p != len(frames):
            self.group.layout(len(frames), with_decode)
            self.n_p = len(frames)
        self.group.set_ref(ref)
        if len(frames):
            self.group.upload(0, frames)
        self.group.encode()
        return self.group.get_mv()

    def _owner(self, k):
        for i in range(self.group.n_local):
            first, count = self.group.shard(i)
            if first <= k < first + count:
                return self.contexts[i], self.group.slot_of(i, k)
        raise IndexError("P frame %d is not on a local shard" % k)

    def planes(self, k):
        c, slot = self._owner(k)
        return c.gop_get_planes(slot)

    def close(self):
        self.group.close()
        for c in self.contexts:
            c.close()


class DistGopEncoder:
    """Encodes the P frames of one GOP, sharded over the processes of a job.

    encode_gop(ref, frames):
        ref     H x W x 3 uint8 on rank 0 (others may pass None),
        frames  this rank's chunk of P frames (shard_frames tells which),
      returns (mv_all, n_static_all) for ALL P frames of the GOP in order, on
      every rank; this rank's coefficient planes stay on its GPU
      (`planes(local_slot)`).
    """

    def __init__(self, H, W, me_bs=16, n_frames=29, backend=None, group=None, src=0,
                 device=None, Q=None, coeff_mode=vcs_hip.VCS_COEFF_F64):
        self.H, self.W, self.me_bs, self.n_frames, self.src = H, W, me_bs, n_frames, src
        self.dist = None
        self.group = group
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.dist = dist
                self.rank = dist.get_rank(group)
                self.world = dist.get_world_size(group)
        except ImportError:
            pass
        self.N = (H // me_bs) * (W // me_bs)
        self.first, self.count = shard_frames(n_frames, self.world, self.rank)
        self.backend = backend
        self._on_gpu = backend is None
        if self._on_gpu:
            if src != 0:
                raise ValueError("the library broadcasts the I frame from shard 0")
            if device is None:
                device = int(os.environ.get("LOCAL_RANK", "0"))   # one GPU per process
            self.ctx = vcs_hip.Context(H, W, me_bs, 8, device)
            if Q is not None or coeff_mode != vcs_hip.VCS_COEFF_F64:
                self.ctx.set_params(Q=Q, coeff_mode=coeff_mode)
            if self.world == 1:
                self.g = vcs_hip.Group([self.ctx])
            else:
                self.g = vcs_hip.Group.from_rank(self.ctx, self._exchange_id(), self.world, self.rank)
                self.g.barrier()
                vcs_hip.rendezvous_cleanup(self.rank, self.world)
            self.g.layout(n_frames)
            assert self.g.shard(0) == (self.first, self.count)

    def _exchange_id(self):
        if self.dist is not None:   # the caller's process group carries the 128 bytes
            box = [vcs_hip.comm_unique_id() if self.rank == 0 else None]
            self.dist.broadcast_object_list(box, src=0, group=self.group)
            return box[0]
        return vcs_hip.rendezvous_unique_id(self.rank, self.world)

    # -- product path -------------------------------------------------------
    def _encode_gpu(self, ref, frames):
        self.g.set_ref(ref if self.rank == 0 else None)   # every rank: keeps the collectives in step
        if self.count:
            self.g.upload(self.first, frames)
        self.g.encode()
        return self.g.get_mv()

    def planes(self, slot):
        return self.ctx.gop_get_planes(slot) if self._on_gpu else self.backend.planes(slot)

    def close(self):
        if self._on_gpu:
            self.g.close()
            self.ctx.close()

    # -- injected backend (CPU tests): torch.distributed carries the exchanges --
    def _broadcast_ref(self, ref):
        if self.world == 1:
            self.backend.set_ref(ref)
            return
        import torch
        buf = torch.empty((self.H, self.W, 3), dtype=torch.uint8)
        if self.rank == self.src:
            buf.copy_(torch.from_numpy(np.ascontiguousarray(ref)))
        self.dist.broadcast(buf, src=self.src, group=self.group)
        self.backend.set_ref(buf.numpy())

    def _gather(self, local, dtype, trailing):
        """All-gather of per-frame arrays with uneven chunk sizes (padded to the
        largest chunk; 29 frames over 8 ranks gives chunks of 4 and 3)."""
        if self.world == 1:
            return local
        import torch
        table = shard_table(self.n_frames, self.world)
        cmax = max(c for _, c in table)
        pad = np.zeros((cmax,) + trailing, dtype)
        pad[:self.count] = local
        send = torch.from_numpy(pad.view(np.uint8).reshape(-1))
        recv = torch.empty(self.world * send.numel(), dtype=torch.uint8)
        self.dist.all_gather_into_tensor(recv, send, group=self.group)
        allb = recv.numpy().reshape(self.world, -1)
        out = np.zeros((self.n_frames,) + trailing, dtype)
        for r, (first, count) in enumerate(table):
            out[first:first + count] = allb[r].view(dtype).reshape((cmax,) + trailing)[:count]
        return out

    def encode_gop(self, ref, frames):
        frames = np.asarray(frames)
        if len(frames) != self.count:
            raise ValueError("rank %d owns %d P frames of this GOP (frames %d..%d), got %d"
                             % (self.rank, self.count, self.first, self.first + self.count - 1,
                                len(frames)))
        if self._on_gpu:
            return self._encode_gpu(ref, frames)
        self._broadcast_ref(ref)
        mv, ns = self.backend.encode(frames)
        mv_all = self._gather(np.asarray(mv, np.int16), np.int16, (self.N, 2))
        ns_all = self._gather(np.asarray(ns, np.uint32), np.uint32, ())
        return mv_all, ns_all
