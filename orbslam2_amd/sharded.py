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


def pair_block(n_frames, rank, world):
    """Consecutive-frame matching over frame blocks: rank r extracts frames [lo, hi_ext) -- its block plus,
    when a later frame exists, frame hi as a halo (the next rank's first frame, extracted twice rather than
    exchanged: 2000 keypoints of 60 bytes are cheaper to recompute than to synchronise) -- and matches the
    pairs (t, t+1) for t in [lo, pair_hi): every pair whose first frame it owns.  Returns (lo, hi_ext, pair_hi)."""
    lo, hi = frame_block(n_frames, rank, world)
    if hi <= lo:
        return lo, lo, lo
    return lo, min(hi + 1, n_frames), min(hi, n_frames - 1)


class ShardedMatcher:
    """local_top2(queries, idx_base) -> (best int32[nq], d1 uint16[nq], d2 uint16[nq]) as numpy arrays or
    torch tensors: the top-2 of the queries over this rank's shard, indices already global."""

    def __init__(self, local_top2, rank=0, world=1, group=None, device="cpu", matcher=None):
        self.local_top2 = local_top2
        self.rank, self.world, self.group, self.device = rank, world, group, device
        self.matcher = matcher  # an ORBmatcher on this rank's GPU: needed by match_device only
        self._bufs = {}
        self._stream = None

    def match_device(self, d_queries, nq, d_db, ndb, idx_base):
        """Everything on the GPU: the shard's top-2 goes straight into an 8-byte-per-query block, one
        all-gather (RCCL) lays the blocks of all ranks side by side, one small kernel merges them.  d_queries
        and d_db are CUDA tensors; returns (best int32, d1 int16, d2 int16) CUDA tensors (views of buffers
        that the next call overwrites).  With a CPU backend (gloo rehearsal) the blocks travel through host
        memory, the merge still runs on the GPU."""
        import torch
        m = self.matcher
        if nq not in self._bufs:
            dev = d_queries.device
            self._bufs[nq] = (torch.empty(8 * nq, dtype=torch.uint8, device=dev),
                              torch.empty(self.world * 8 * nq, dtype=torch.uint8, device=dev),
                              torch.empty(nq, dtype=torch.int32, device=dev),
                              torch.empty(nq, dtype=torch.int16, device=dev),
                              torch.empty(nq, dtype=torch.int16, device=dev))
        blk, allb, best, d1, d2 = self._bufs[nq]
        # Kernels of the library and torch's collective must share one real stream: handed the null stream,
        # the matcher would fall back to its own, unordered with torch's.  Work is fenced against the
        # caller's current stream on both sides.
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=d_queries.device)
        caller = torch.cuda.current_stream()
        self._stream.wait_stream(caller)
        with torch.cuda.stream(self._stream):
            m.set_stream(self._stream.cuda_stream)
            p = blk.data_ptr()
            m.match_top2_device(d_queries.data_ptr(), nq, d_db.data_ptr(), ndb, idx_base, p, p + 4 * nq, p + 6 * nq)
            if self.world > 1:
                import torch.distributed as dist
                if self.device == "cpu":
                    host = blk.cpu()
                    parts = torch.empty(self.world * 8 * nq, dtype=torch.uint8)
                    dist.all_gather(list(parts.chunk(self.world)), host, group=self.group)
                    allb.copy_(parts)
                else:
                    dist.all_gather_into_tensor(allb, blk, group=self.group)
            else:
                allb.copy_(blk)
            m.merge_top2_shards_device(self.world, nq, allb.data_ptr(), 8 * nq, best.data_ptr(), d1.data_ptr(),
                                       d2.data_ptr())
        caller.wait_stream(self._stream)
        return best, d1, d2

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


class RcclShardedMatcher:
    """The library's own sharded matcher (orbx_sharded_* in include/orbx.h): per-shard top-2, ncclAllGather
    and merge enqueued by C code on one HIP stream, the path a C++ caller takes.  Python only carries the
    128-byte communicator id from rank 0 to the other ranks (`exchange_id`: a callable that takes rank 0's
    bytes, or None on other ranks, and returns rank 0's bytes everywhere; default: a torch.distributed
    broadcast over the default group).  One rank per GPU: RCCL rejects two ranks on one device."""

    ID_BYTES = 128

    def __init__(self, rank, world, device, max_queries=4096, exchange_id=None):
        import ctypes as C

        from . import _lib as L
        self._L, self._C = L, C
        lib = L.lib()
        buf = (C.c_uint8 * self.ID_BYTES)()
        if rank == 0:
            L.check(lib.orbx_sharded_unique_id(buf), "orbx_sharded_unique_id")
        raw = bytes(buf)
        if world > 1:
            raw = (exchange_id or _broadcast_bytes)(raw if rank == 0 else None)
        assert len(raw) == self.ID_BYTES
        self._h = C.c_void_p()
        idbuf = (C.c_uint8 * self.ID_BYTES).from_buffer_copy(raw)
        L.check(lib.orbx_sharded_matcher_create(device, rank, world, idbuf, max_queries, C.byref(self._h)),
                "orbx_sharded_matcher_create")
        self.rank, self.world = rank, world

    def set_stream(self, hip_stream):
        self._L.check(self._L.lib().orbx_sharded_matcher_set_stream(self._h, self._C.c_void_p(hip_stream)),
                      "orbx_sharded_matcher_set_stream")

    def match_device(self, d_queries, nq, d_shard, shard_rows, idx_base, d_best, d_d1, d_d2):
        """Device addresses (ints); asynchronous on the handle's stream; collective."""
        P = self._C.c_void_p
        self._L.check(self._L.lib().orbx_match_top2_sharded_device(self._h, P(d_queries), nq, P(d_shard), shard_rows,
                                                                  idx_base, P(d_best), P(d_d1), P(d_d2)),
                      "orbx_match_top2_sharded_device")

    def synchronize(self):
        self._L.check(self._L.lib().orbx_sharded_matcher_synchronize(self._h), "orbx_sharded_matcher_synchronize")

    def close(self):
        if self._h:
            self._L.lib().orbx_sharded_matcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _broadcast_bytes(raw):
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.zeros(RcclShardedMatcher.ID_BYTES, dtype=torch.uint8, device=dev)
    if raw is not None:
        t.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    dist.broadcast(t, src=0)
    return bytes(t.cpu().numpy().tobytes())


def _as_i32(v):
    import torch
    if torch.is_tensor(v):
        return (v.to(torch.int32) & 0xFFFF)
    return torch.as_tensor(np.asarray(v).astype(np.int32))
