"""GOP-level host layer: batched, HBM-resident encode of the P frames of a GOP
and its multi-GPU form (one process per GPU, torch.distributed -- backend
"nccl" is RCCL over xGMI on ROCm).

Why the path shards: every P frame is predicted from the RAW I frame of its
GOP, never from another P frame or from reconstructed data (reference
encoder.py:51-52), so the P frames of a GOP are mutually independent.  They
are dealt to the ranks in contiguous chunks; the only exchanges are
  1. a broadcast of the I frame from the rank that holds it (F bytes), and
  2. an all-gather of the motion vectors (4*N bytes per frame).
Coefficient planes (8*F bytes per frame) never cross xGMI: each rank keeps /
downloads its own.  There is no reduction anywhere.

PyTorch is used for process-group plumbing only; the compute goes through
libvcs_hip.so (vcs_hip.Context).  For CPU-only tests of the distributed logic a
`backend` object with the same four methods can be injected (tests use the
oracle there); the default backend is the HIP context and nothing else.
"""
import numpy as np

import vcs_hip


def shard_frames(n_frames, world, rank):
    """Contiguous chunk [first, first + count) of `n_frames` P frames for `rank`;
    chunk sizes differ by at most one, earlier ranks take the larger chunks."""
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_frames, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def shard_table(n_frames, world):
    return [shard_frames(n_frames, world, r) for r in range(world)]


class _DevBytes:
    """Zero-copy torch view of a libvcs_hip HBM buffer (for RCCL collectives)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1",
                                         "data": (ptr, False), "version": 2}


class HipGopBackend:
    """The product backend: one vcs_hip.Context, frames resident in HBM."""

    def __init__(self, H, W, me_bs, device=0, Q=None, coeff_mode=vcs_hip.VCS_COEFF_F64):
        self.ctx = vcs_hip.Context(H, W, me_bs, 8, device)
        if Q is not None or coeff_mode != vcs_hip.VCS_COEFF_F64:
            self.ctx.set_params(Q=Q, coeff_mode=coeff_mode)
        self.N = self.ctx.N
        self.slots = 0

    def reserve(self, n, with_decode=False):
        if n > self.slots or with_decode:
            self.ctx.gop_reserve(max(n, 1), with_decode)
            self.slots = max(n, 1)

    def set_ref(self, ref):
        self.ctx.set_ref(ref)

    def ref_tensor(self):
        import torch
        ptr, nb = self.ctx.dev_ptr(0)
        return torch.as_tensor(_DevBytes(ptr, nb), device="cuda")

    def ref_written(self):
        self.ctx.ref_written()

    def encode(self, frames):
        n = len(frames)
        self.reserve(n)
        if n:
            self.ctx.gop_upload(0, frames)
            self.ctx.gop_encode(0, n)
        self.ctx.sync()
        return self.ctx.gop_get_mv(0, n) if n else (np.zeros((0, self.N, 2), np.int16),
                                                     np.zeros(0, np.uint32))

    def planes(self, slot):
        return self.ctx.gop_get_planes(slot)

    def close(self):
        self.ctx.close()


class DistGopEncoder:
    """Encodes the P frames of one GOP, sharded over the ranks of `group`.

    encode_gop(ref, frames):
        ref     H x W x 3 uint8 on the source rank (others may pass None),
        frames  this rank's chunk of P frames (shard_frames tells which),
      returns (mv_all, n_static_all) for ALL P frames of the GOP in order, on
      every rank; this rank's coefficient planes stay in its backend
      (`backend.planes(local_slot)`).
    """

    def __init__(self, H, W, me_bs=16, n_frames=29, backend=None, group=None, src=0,
                 device=None):
        self.H, self.W, self.me_bs, self.n_frames, self.src = H, W, me_bs, n_frames, src
        self.dist = None
        self.group = group
        self.rank, self.world = 0, 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.dist = dist
                self.rank = dist.get_rank(group)
                self.world = dist.get_world_size(group)
        except ImportError:
            pass
        if backend is None:
            backend = HipGopBackend(H, W, me_bs, device if device is not None else 0)
        self.backend = backend
        self.N = (H // me_bs) * (W // me_bs)
        self.first, self.count = shard_frames(n_frames, self.world, self.rank)
        self._on_gpu = isinstance(backend, HipGopBackend)

    def _broadcast_ref(self, ref):
        if self.world == 1:
            self.backend.set_ref(ref)
            return
        import torch
        if self._on_gpu:
            # rank `src` uploads once; RCCL broadcasts HBM -> HBM straight into
            # every context's resident reference buffer (zero-copy views)
            if self.rank == self.src:
                self.backend.set_ref(ref)
            t = self.backend.ref_tensor()
            self.dist.broadcast(t, src=self.src, group=self.group)
            torch.cuda.current_stream().synchronize()
            self.backend.ref_written()
        else:
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
        dev = "cuda" if self._on_gpu else "cpu"
        send = torch.from_numpy(pad.view(np.uint8).reshape(-1)).to(dev)
        recv = torch.empty(self.world * send.numel(), dtype=torch.uint8, device=dev)
        self.dist.all_gather_into_tensor(recv, send, group=self.group)
        allb = recv.cpu().numpy().reshape(self.world, -1)
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
        self._broadcast_ref(ref)
        mv, ns = self.backend.encode(frames)
        mv_all = self._gather(np.asarray(mv, np.int16), np.int16, (self.N, 2))
        ns_all = self._gather(np.asarray(ns, np.uint32), np.uint32, ())
        return mv_all, ns_all
