"""world_size-2 run of the sharded matcher on CPU (gloo): sharding, all-gather and merge are the
production code; the per-shard top-2 comes from the CPU oracle here because there is no GPU."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ndb, nq, q_out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import orbslam2_amd as orbx
    from orbslam2_amd.sharded import ShardedMatcher, shard_bounds
    from oracle import pyoracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    db = orbx.synth_descriptors(21, ndb)
    db[ndb // 2 + 3] = db[5]  # the same row in both shards: the lower global index must win
    q = db[[5, 100, ndb // 2 + 3, ndb - 1, 17]].copy()[:nq]
    q[1, 7] ^= 0x81
    lo, hi = shard_bounds(ndb, rank, world)
    shard = db[lo:hi]
    sm = ShardedMatcher(lambda qq, base: orc.match_top2(qq, shard, idx_base=base), rank, world, device="cpu")
    best, d1, d2 = sm.match(q, lo)
    ob, o1, o2 = orc.match_top2(q, db)
    ok = bool(np.array_equal(best, ob) and np.array_equal(d1, o1) and np.array_equal(d2, o2) and best[2] == 5)
    q_out.put((rank, ok, best.tolist()))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_sharded_match_equals_single_pass():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 4001, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]


def test_pair_ownership_of_consecutive_frame_matching():
    """Every consecutive pair (t, t+1) is matched by exactly one rank, the one whose block holds t, and that
    rank extracts both frames: its block plus at most one halo frame."""
    from orbslam2_amd.sharded import frame_block, pair_block
    for n, w in [(4096, 8), (4096, 1), (10, 4), (9, 4), (3, 8), (1, 2), (0, 2), (2048, 2)]:
        owners = {}
        for r in range(w):
            lo, hi_ext, pair_hi = pair_block(n, r, w)
            blo, bhi = frame_block(n, r, w)
            assert lo == blo and bhi <= hi_ext <= bhi + 1 and hi_ext <= n
            for t in range(lo, pair_hi):
                assert t not in owners, (n, w, t)
                owners[t] = r
                assert lo <= t and t + 1 < hi_ext  # both frames of the pair were extracted by this rank
                assert blo <= t < bhi              # and the first one is its own
        assert sorted(owners) == list(range(max(n - 1, 0))), (n, w)


def _pair_worker(rank, world, port, n, q_out):
    """Two ranks match the consecutive pairs of one sequence with the halo rule; the pair results must equal
    those of one rank matching the whole sequence (oracle matcher on oracle features: no GPU here)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import orbslam2_amd as orbx
    from orbslam2_amd.sharded import pair_block
    from oracle import pyoracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h = 200, 150
    canvas = orbx.synth_image(77, w + 4 * n, h, 0)
    frame = lambda t: np.ascontiguousarray(canvas[:, 4 * t:4 * t + w])
    lo, hi_ext, pair_hi = pair_block(n, rank, world)
    feats = {t: orc.extract(frame(t), 300, 1.2, 4, 20, 7) for t in range(lo, hi_ext)}
    bounds = (0.0, float(w), 0.0, float(h))

    def match(a, b):
        (k1, d1), (k2, d2) = a, b
        prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], axis=1), np.float32)
        return int(orc.search_for_initialization(k1, d1, k2, d2, bounds, prev, 30, 0.9, True)[0])

    mine = torch.full((n - 1,), -1, dtype=torch.int64)
    for t in range(lo, pair_hi):
        mine[t] = match(feats[t], feats[t + 1])
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)  # the test's own exchange: the data path has none
    merged = torch.stack(parts).max(dim=0).values
    owned_once = bool(((torch.stack(parts) >= 0).sum(dim=0) == 1).all())
    ok = owned_once
    if rank == 0:
        allf = [orc.extract(frame(t), 300, 1.2, 4, 20, 7) for t in range(n)]
        want = [match(allf[t], allf[t + 1]) for t in range(n - 1)]
        ok = ok and merged.tolist() == want and min(want) > 20
    q_out.put((rank, ok, merged.tolist()))
    dist.destroy_process_group()


@pytest.mark.timeout(170)
def test_two_rank_consecutive_frame_matching_with_halo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pair_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _ in res), res


def test_shard_and_frame_bounds():
    from orbslam2_amd.sharded import frame_block, shard_bounds
    for n, w in [(1_000_000, 8), (4096, 8), (10, 4), (3, 8), (0, 2)]:
        cover = []
        for r in range(w):
            lo, hi = shard_bounds(n, r, w)
            assert 0 <= lo <= hi <= n
            cover += list(range(lo, hi)) if n <= 4096 else []
            assert frame_block(n, r, w) == (lo, hi)
        if n <= 4096:
            assert cover == list(range(n))
    assert shard_bounds(1_000_000, 3, 8) == (375000, 500000)
