"""Relocalisation-style matching against a descriptor database sharded over GPUs.

One process per GPU.  Rank r owns the contiguous rows [r*ceil(N/W), ...) of the database, every rank
holds all queries.  Each rank runs the top-2 kernel on its shard, the per-shard (best index, d1, d2)
triples (12 bytes per query) are exchanged with one all-gather (RCCL over xGMI with the "nccl"
backend, gloo on CPU), and every rank merges them into the result of a single pass over the whole
database (src/ORBmatcher.cc:217-242 of the reference: strict '<', first index wins).  Frame-batch
extraction needs no such step: frames are independent, see frame_block().
"""
import numpy as np

from .matcher import ORBmatcher


def shard_bounds(n_rows, rank, world):
    """Contiguous rows [lo, hi) of the database that rank owns."""
    per = (n_rows + world - 1) // world
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows)


def frame_block(n_frames, rank, world):
    """Contiguous frame indices [lo, hi) that rank extracts (no halo: extraction is per frame).
    Consecutive-frame matching additionally needs frame hi (the next rank's first frame) as the
    partner of frame hi-1; the rank that owns frame t matches the pair (t, t+1)."""
    per = (n_frames + world - 1) // world
    lo = min(rank * per, n_frames)
    return lo, min(lo + per, n_frames)


class ShardedMatcher:
    """local_top2(queries, idx_base) -> (best int32[nq], d1 uint16[nq], d2 uint16[nq]) as numpy arrays or
    torch tensors: the top-2 of the queries over this rank's shard, indices already global."""

    def __init__(self, local_top2, rank=0, world=1, group=None, device="cpu"):
        self.local_top2 = local_top2
        self.rank, self.world, self.group, self.device = rank, world, group, device

    def match(self, queries, idx_base):
        import torch
        best, d1, d2 = self.local_top2(queries, idx_base)
        packed = torch.stack([torch.as_tensor(np.asarray(best) if not torch.is_tensor(best) else best).to(torch.int32),
                              _as_i32(d1), _as_i32(d2)], dim=1).contiguous().to(self.device)
        if self.world > 1:
            import torch.distributed as dist
            parts = [torch.empty_like(packed) for _ in range(self.world)]
            dist.all_gather(parts, packed, group=self.group)
            out = torch.stack(parts)
        else:
            out = packed[None]
        host = out.cpu().numpy()
        return ORBmatcher.merge_top2(host[:, :, 0], host[:, :, 1].astype(np.uint16), host[:, :, 2].astype(np.uint16))


def _as_i32(v):
    import torch
    if torch.is_tensor(v):
        return (v.to(torch.int32) & 0xFFFF)
    return torch.as_tensor(np.asarray(v).astype(np.int32))
