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
