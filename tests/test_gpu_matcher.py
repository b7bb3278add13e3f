"""GPU parity of the matcher through the C ABI: indices and distances bit-exact against the oracle."""
import os

import numpy as np
import pytest

import orbslam2_amd as orbx

pytestmark = pytest.mark.gpu


def make_db(n, nq, seed=1):
    """Config-4 style data: random database, queries = database rows with 0..80 flipped bits, and a
    near-duplicate next to every tenth source row."""
    db = orbx.synth_descriptors(seed, n)
    rng = np.random.default_rng(seed + 1)
    step = max(1, n // nq)
    q = np.zeros((nq, 32), np.uint8)
    for i in range(nq):
        row = (i * step) % n
        bits = np.unpackbits(db[row])
        k = rng.integers(0, 81)
        flip = rng.choice(256, size=k, replace=False)
        bits[flip] ^= 1
        q[i] = np.packbits(bits)
        if i % 10 == 0 and row + 1 < n:
            b2 = np.unpackbits(db[row])
            b2[rng.choice(256, size=4, replace=False)] ^= 1
            db[row + 1] = np.packbits(b2)
    return db, q


@pytest.mark.parametrize("ndb,nq", [(1, 1), (3, 5), (255, 64), (4096, 300), (100003, 2000)])
def test_top2_vs_oracle(oracle, ndb, nq):
    db, q = make_db(ndb, nq)
    m = orbx.ORBmatcher(0.75)
    best, d1, d2 = m.match_top2(q, db)
    ob, o1, o2 = oracle.match_top2(q, db)
    assert np.array_equal(best, ob)
    assert np.array_equal(d1, o1)
    assert np.array_equal(d2, o2)
    acc, n = m.ratio_test(d1, d2)
    want = [oracle.lib().orc_ratio_accept(int(a), int(b), 50, 0.75) for a, b in zip(o1, o2)]
    assert list(acc.astype(int)) == want and n == sum(want)


def test_top2_ties_and_extremes(oracle):
    # duplicates (first index must win), an all-different row (distance 256 never becomes best)
    q = np.zeros((3, 32), np.uint8)
    q[1] = 0xFF
    q[2, :4] = 0xF0
    db = np.zeros((10, 32), np.uint8)
    db[0] = 0xFF
    db[3] = 0x0F
    db[4:] = 0xFF
    m = orbx.ORBmatcher()
    best, d1, d2 = m.match_top2(q, db)
    ob, o1, o2 = oracle.match_top2(q, db)
    assert np.array_equal(best, ob) and np.array_equal(d1, o1) and np.array_equal(d2, o2)
    only = np.full((1, 32), 0xFF, np.uint8)
    best, d1, d2 = m.match_top2(np.zeros((1, 32), np.uint8), only)
    assert best[0] == -1 and d1[0] == 256 and d2[0] == 256
    # empty database
    best, d1, d2 = m.match_top2(q, np.zeros((0, 32), np.uint8))
    assert list(best) == [-1, -1, -1] and list(d1) == [256] * 3


@pytest.mark.parametrize("ndb,nq", [(31, 32), (32, 33), (33, 127), (63, 128), (545, 129), (8159, 40), (8160, 513),
                                    (8161, 64), (16325, 96), (20000, 1100)])
def test_top2_matrix_core_path_ties_at_tile_and_chunk_borders(oracle, ndb, nq):
    """32 queries or more run on the matrix cores: 32-row tiles, 16 rows per lane, chunks of whole tiles.
    Rows drawn from a pool of 40 descriptors tie all the time, so the first index has to win across
    lanes, tiles and chunks, and the second distance often equals the first."""
    rng = np.random.default_rng(ndb * 7 + nq)
    pool = orbx.synth_descriptors(3, 40)
    db = pool[rng.integers(0, 40, size=ndb)]
    q = pool[rng.integers(0, 40, size=nq)].copy()
    for i in range(0, nq, 3):                      # some queries a few bits off their pool row
        bits = np.unpackbits(q[i])
        bits[rng.choice(256, size=int(rng.integers(1, 9)), replace=False)] ^= 1
        q[i] = np.packbits(bits)
    q[-1] = ~db[0]                                   # distance 256 to row 0
    m = orbx.ORBmatcher(0.75)
    best, d1, d2 = m.match_top2(q, db)
    ob, o1, o2 = oracle.match_top2(q, db)
    assert np.array_equal(best, ob)
    assert np.array_equal(d1, o1)
    assert np.array_equal(d2, o2)


def test_top2_matrix_core_longest_chunks(oracle, monkeypatch):
    """Chunks of the greatest length the key layout allows (32736 rows: row codes up to 32767 below a key
    step of 32768), forced through the tuning variable; ties everywhere, so the row codes decide."""
    monkeypatch.setenv("ORBX_TOP2_BLOCKS", "1")
    rng = np.random.default_rng(5)
    ndb, nq = 32736 + 7301, 70
    pool = orbx.synth_descriptors(9, 25)
    db = pool[rng.integers(0, 25, size=ndb)]
    q = pool[rng.integers(0, 25, size=nq)].copy()
    q[::2, 3] ^= 0x10
    m = orbx.ORBmatcher(0.75)
    best, d1, d2 = m.match_top2(q, db)
    ob, o1, o2 = oracle.match_top2(q, db)
    assert np.array_equal(best, ob) and np.array_equal(d1, o1) and np.array_equal(d2, o2)
    # the last rows of the first chunk and the first of the second
    db2 = orbx.synth_descriptors(10, ndb)
    q2 = db2[[32735, 32736, 32734, 0, ndb - 1]].copy()
    q2 = np.concatenate([q2] * 8)
    best, d1, d2 = m.match_top2(q2, db2)
    assert list(best[:5]) == [32735, 32736, 32734, 0, ndb - 1] and (d1 == 0).all()


def test_sharded_merge_equals_single_pass(oracle):
    db, q = make_db(50000, 500, seed=5)
    m = orbx.ORBmatcher(0.75)
    shards = 8
    per = (len(db) + shards - 1) // shards
    bs, d1s, d2s = [], [], []
    for s in range(shards):
        b, a1, a2 = m.match_top2(q, db[s * per:(s + 1) * per], idx_base=s * per)
        bs.append(b), d1s.append(a1), d2s.append(a2)
    mb, m1, m2 = m.merge_top2(np.stack(bs), np.stack(d1s), np.stack(d2s))
    ob, o1, o2 = oracle.match_top2(q, db)
    assert np.array_equal(mb, ob) and np.array_equal(m1, o1) and np.array_equal(m2, o2)


def test_device_shard_merge_equals_single_pass(oracle):
    """The exchange format of the sharded matcher: every shard's top-2 written straight into its 8-byte-per-query
    block, the blocks side by side as an all-gather leaves them, merged by orbx_top2_merge_shards_device.  Must
    equal one pass over the whole database (ties to the lower shard, duplicates across shards)."""
    import torch
    from orbslam2_amd.sharded import ShardedMatcher, shard_bounds
    ndb, nq, world = 30011, 501, 5
    db, q = make_db(ndb, nq, seed=9)
    db[20000] = db[7]      # the same row in two shards
    db[25000:25003] = q[3]  # exact matches in the last shard, best and second tie at distance 0
    m = orbx.ORBmatcher(0.75)
    d_db, d_q = torch.from_numpy(db).cuda(), torch.from_numpy(q).cuda()
    blocks = torch.zeros(world * 8 * nq, dtype=torch.uint8, device="cuda")
    best = torch.zeros(nq, dtype=torch.int32, device="cuda")
    d1 = torch.zeros(nq, dtype=torch.int16, device="cuda")
    d2 = torch.zeros(nq, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()  # the matcher works on its own stream
    for r in range(world):
        lo, hi = shard_bounds(ndb, r, world)
        p = blocks.data_ptr() + r * 8 * nq
        m.match_top2_device(d_q.data_ptr(), nq, d_db.data_ptr() + lo * 32, hi - lo, lo, p, p + 4 * nq, p + 6 * nq)
    m.merge_top2_shards_device(world, nq, blocks.data_ptr(), 8 * nq, best.data_ptr(), d1.data_ptr(), d2.data_ptr())
    torch.cuda.synchronize()
    ob, o1, o2 = oracle.match_top2(q, db)
    assert np.array_equal(best.cpu().numpy(), ob)
    assert np.array_equal(d1.cpu().numpy().view(np.uint16), o1) and np.array_equal(d2.cpu().numpy().view(np.uint16), o2)
    # the host merge on the same blocks
    hb = blocks.cpu().numpy().reshape(world, 8 * nq)
    hbest = np.stack([hb[r, :4 * nq].view(np.int32) for r in range(world)])
    h1 = np.stack([hb[r, 4 * nq:6 * nq].view(np.uint16) for r in range(world)])
    h2 = np.stack([hb[r, 6 * nq:].view(np.uint16) for r in range(world)])
    mb, m1, m2 = orbx.ORBmatcher.merge_top2(hbest, h1, h2)
    assert np.array_equal(mb, ob) and np.array_equal(m1, o1) and np.array_equal(m2, o2)
    # one rank through ShardedMatcher.match_device (no collective): the same path with a world of one
    sm = ShardedMatcher(None, 0, 1, device="cuda", matcher=m)
    b, a1, a2 = sm.match_device(d_q, nq, d_db, ndb, 0)
    torch.cuda.synchronize()
    assert np.array_equal(b.cpu().numpy(), ob) and np.array_equal(a1.cpu().numpy().view(np.uint16), o1)
    assert np.array_equal(a2.cpu().numpy().view(np.uint16), o2)


def test_full_size_properties():
    """Config-4 size (2000 x 1M): checked through properties instead of the CPU loop: every query was
    cut from a known row, so a query with few flipped bits must find it; d1 <= d2; shard sum rule."""
    n, nq = 1_000_000, 2000
    db = orbx.synth_descriptors(1, n)
    rng = np.random.default_rng(2)
    rows = np.arange(nq) * 500
    q = db[rows].copy()
    flips = rng.integers(0, 60, size=nq)
    for i in range(nq):
        bits = np.unpackbits(q[i])
        bits[rng.choice(256, size=flips[i], replace=False)] ^= 1
        q[i] = np.packbits(bits)
    m = orbx.ORBmatcher(0.75)
    best, d1, d2 = m.match_top2(q, db)
    assert np.array_equal(best, rows)          # random rows sit ~128 bits away; the source row wins
    assert np.array_equal(d1, flips.astype(np.uint16))
    assert (d1 <= d2).all() and (d2 > 60).all() and (d2 <= 128).all()
    half = n // 2
    b0 = m.match_top2(q, db[:half])
    b1 = m.match_top2(q, db[half:], idx_base=half)
    mb, m1, m2 = m.merge_top2(np.stack([b0[0], b1[0]]), np.stack([b0[1], b1[1]]), np.stack([b0[2], b1[2]]))
    assert np.array_equal(mb, best) and np.array_equal(m1, d1) and np.array_equal(m2, d2)


def test_search_for_initialization_vs_oracle(oracle):
    w, h = 1241, 376
    ex = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h)
    img1 = orbx.synth_image(0x0B5EED00, w, h, 0)
    # second frame: the first shifted by a few pixels plus fresh noise -> real matches
    img2 = np.roll(img1, (3, 5), axis=(0, 1))
    noise = np.random.default_rng(3).integers(-2, 3, size=img2.shape)
    img2 = np.clip(img2.astype(np.int16) + noise, 0, 255).astype(np.uint8)
    k1, d1 = ex(img1)
    k2, d2 = ex(img2)
    bounds = (0.0, float(w), 0.0, float(h))
    for ratio, ori, win in [(0.9, True, 100), (0.9, False, 100), (0.6, True, 30)]:
        prev_g = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32).copy()
        prev_o = prev_g.copy()
        m = orbx.ORBmatcher(ratio, ori)
        n, m12, pairs = m.SearchForInitialization(k1, d1, k2, d2, bounds, prev_g, win)
        on, om12, opairs = oracle.search_for_initialization(k1, d1, k2, d2, bounds, prev_o, win, ratio, ori)
        assert n == on and pairs == opairs
        assert np.array_equal(m12, om12)
        assert np.array_equal(prev_g, prev_o)
        assert n > 50


def test_golden_top2_and_device_sharded_path():
    import os
    import torch
    from orbslam2_amd.sharded import ShardedMatcher
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "top2_64x5000.npz"))
    db = orbx.synth_descriptors(11, 5000)
    q = db[::78][:64].copy()
    q[:, 3] ^= 0x3C
    m = orbx.ORBmatcher(0.75)
    best, d1, d2 = m.match_top2(q, db)
    assert np.array_equal(best, g["best"]) and np.array_equal(d1, g["d1"]) and np.array_equal(d2, g["d2"])
    # device-resident form through the sharded wrapper (world size 1: no exchange, same merge code)
    ddb, dq = torch.from_numpy(db).cuda(), torch.from_numpy(q).cuda()
    tb = torch.zeros(64, dtype=torch.int32, device="cuda")
    t1 = torch.zeros(64, dtype=torch.int16, device="cuda")
    t2 = torch.zeros(64, dtype=torch.int16, device="cuda")
    m.set_stream(torch.cuda.current_stream().cuda_stream)

    def local(qq, base):
        m.match_top2_device(qq.data_ptr(), 64, ddb.data_ptr(), 5000, base, tb.data_ptr(), t1.data_ptr(),
                            t2.data_ptr())
        return tb, t1, t2

    sb, s1, s2 = ShardedMatcher(local, 0, 1, device="cuda").match(dq, 0)
    assert np.array_equal(sb, g["best"]) and np.array_equal(s1, g["d1"]) and np.array_equal(s2, g["d2"])


@pytest.mark.parametrize("ratio,ori,win", [(0.9, True, 100), (0.9, False, 100), (0.6, True, 30), (0.9, True, 400)])
def test_search_for_initialization_batch_device_vs_oracle(oracle, ratio, ori, win):
    """Consecutive pairs of a device-resident batch, matched on the GPU, against the oracle pair by pair."""
    import torch
    w, h, nf = 1241, 376, 5
    base = orbx.synth_image(0x0B5EED00, w, h, 0)
    rng = np.random.default_rng(7)
    frames = []
    for t in range(nf):
        img = np.roll(base, (2 * t, 3 * t), axis=(0, 1)).astype(np.int16) + rng.integers(-2, 3, size=base.shape)
        frames.append(np.clip(img, 0, 255).astype(np.uint8))
    frames[3] = orbx.synth_image(99, w, h, 1)  # a flat frame: few or no keypoints on one side of two pairs
    ex = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=nf)
    ex.extract_batch_host(frames)
    host = [ex.batch_download(f) for f in range(nf)]
    d_kp, d_desc, d_cnt, cap = ex.device_results()
    m12 = torch.full((nf - 1, cap), -7, dtype=torch.int32, device="cuda")
    nm = torch.zeros(nf - 1, dtype=torch.int32, device="cuda")
    npairs = torch.zeros(nf - 1, dtype=torch.int32, device="cuda")
    prev_out = torch.full((nf - 1, cap, 2), -1.0, dtype=torch.float32, device="cuda")
    m = orbx.ORBmatcher(ratio, ori)
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    bounds = (0.0, float(w), 0.0, float(h))
    m.search_for_initialization_batch_device(d_kp, d_desc, d_cnt, cap, nf, 600, bounds, m12.data_ptr(),
                                             nm.data_ptr(), npairs.data_ptr(), windowSize=win,
                                             d_prev_matched_out=prev_out.data_ptr())
    torch.cuda.synchronize()
    m12, nm, npairs, prev_out = m12.cpu().numpy(), nm.cpu().numpy(), npairs.cpu().numpy(), prev_out.cpu().numpy()
    total = 0
    for t in range(nf - 1):
        (k1, d1), (k2, d2) = host[t], host[t + 1]
        prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32).copy()
        on, om12, opairs = oracle.search_for_initialization(k1, d1, k2, d2, bounds, prev, win, ratio, ori)
        assert nm[t] == on and npairs[t] == opairs
        assert np.array_equal(m12[t, :len(k1)], om12)
        assert (m12[t, len(k1):] == -1).all()
        hit = om12 >= 0
        assert np.array_equal(prev_out[t, :len(k1)][hit], prev[hit])
        assert (prev_out[t, :len(k1)][~hit] == -1.0).all()
        total += on
    assert total > 100
    # a workspace smaller than the level-0 keypoints is reported per pair
    m12b = torch.zeros((nf - 1, cap), dtype=torch.int32, device="cuda")
    nmb = torch.zeros(nf - 1, dtype=torch.int32, device="cuda")
    npb = torch.zeros(nf - 1, dtype=torch.int32, device="cuda")
    m.search_for_initialization_batch_device(d_kp, d_desc, d_cnt, cap, nf, 8, bounds, m12b.data_ptr(),
                                             nmb.data_ptr(), npb.data_ptr(), windowSize=win)
    torch.cuda.synchronize()
    assert nmb.cpu().numpy()[0] == -3  # ORBX_E_CAPACITY


def test_distinctive_descriptors_vs_oracle(oracle):
    """Segments of 0 .. 200 observations: noisy copies of a base descriptor (so medians tie often) and
    unrelated rows; the chosen index and descriptor must equal the oracle's."""
    rng = np.random.default_rng(9)
    segs = []
    for n in [0, 1, 2, 3, 5, 8, 13, 21, 34, 55, 63, 64, 65, 100, 200, 4, 4, 7]:
        base = rng.integers(0, 256, size=32, dtype=np.uint8)
        rows = np.repeat(base[None], n, axis=0)
        for i in range(n):
            bits = np.unpackbits(rows[i])
            bits[rng.choice(256, size=int(rng.integers(0, 12)), replace=False)] ^= 1
            rows[i] = np.packbits(bits)
            if i % 7 == 6:
                rows[i] = rng.integers(0, 256, size=32, dtype=np.uint8)   # an outlier observation
        segs.append(rows)
    segs.append(np.repeat(rng.integers(0, 256, size=(1, 32), dtype=np.uint8), 9, axis=0))  # all equal: index 0
    m = orbx.ORBmatcher()
    best, out = m.ComputeDistinctiveDescriptors(segs)
    for p, rows in enumerate(segs):
        want = oracle.distinctive_descriptor(rows)
        assert best[p] == want, (p, len(rows))
        if want >= 0:
            assert np.array_equal(out[p], rows[want])
    assert best[0] == -1 and best[-1] == 0


def _feature_vector(words):
    """DBoW2::FeatureVector of a list of node ids per feature: ascending ids, CSR index lists."""
    ids = np.unique(words).astype(np.int32)
    order = np.argsort(words, kind="stable").astype(np.int32)
    counts = np.array([(words == i).sum() for i in ids], np.int32)
    start = np.zeros(len(ids) + 1, np.int32)
    start[1:] = np.cumsum(counts)
    return ids, start, order


def _bow_item(rng, src_desc, src_words, n, noise_bits, valid_frac):
    """A key frame whose features are noisy copies of random features of the source (with the source's
    word most of the time), plus unrelated ones."""
    pick = rng.integers(0, len(src_desc), size=n)
    desc = src_desc[pick].copy()
    for i in range(n):
        bits = np.unpackbits(desc[i])
        bits[rng.choice(256, size=int(rng.integers(0, noise_bits)), replace=False)] ^= 1
        desc[i] = np.packbits(bits)
    words = src_words[pick].copy()
    wrong = rng.random(n) < 0.1
    words[wrong] = rng.integers(0, src_words.max() + 1, size=int(wrong.sum()))
    fresh = rng.random(n) < 0.15
    desc[fresh] = rng.integers(0, 256, size=(int(fresh.sum()), 32), dtype=np.uint8)
    ids, start, items = _feature_vector(words)
    return dict(desc=desc, angle=(rng.random(n) * 360).astype(np.float32), valid=rng.random(n) < valid_frac,
                node_ids=ids, node_start=start, node_items=items)


@pytest.mark.parametrize("strict,ratio,ori", [(False, 0.75, True), (False, 0.7, False), (True, 0.8, True)])
def test_search_by_bow_vs_oracle(oracle, strict, ratio, ori):
    rng = np.random.default_rng(21)
    n2 = 1500
    d2 = rng.integers(0, 256, size=(n2, 32), dtype=np.uint8)
    w2 = rng.integers(0, 90, size=n2)
    w2[:300] = 7          # one crowded vocabulary node: several passes of 64 lanes
    ids2, start2, items2 = _feature_vector(w2)
    frame = dict(desc=d2, angle=(rng.random(n2) * 360).astype(np.float32),
                 valid=(rng.random(n2) < 0.8) if strict else None, node_ids=ids2, node_start=start2,
                 node_items=items2)
    kfs = [_bow_item(rng, d2, w2, n, 40, 0.7) for n in (900, 1200, 1, 300)]
    # angles close to the frame's for most features, so that the rotation histogram has a clear maximum
    for k in kfs:
        k["angle"] = ((rng.random(len(k["desc"])) * 20) + 100).astype(np.float32)
    frame["angle"][:] = 95.0
    m = orbx.ORBmatcher(ratio, ori)
    nm, m12, m21 = m.SearchByBoW(kfs, frame, strict_lt=strict)
    total = 0
    for b, k in enumerate(kfs):
        on, o12, o21 = oracle.search_by_bow(k["desc"], k["angle"], k["valid"],
                                            (k["node_ids"], k["node_start"], k["node_items"]), d2, frame["angle"],
                                            frame["valid"], (ids2, start2, items2), strict, ratio, ori)
        assert nm[b] == on
        assert np.array_equal(m12[b], o12) and np.array_equal(m21[b], o21)
        total += on
    assert total > 300
    # no key frames, and a key frame without features
    assert len(m.SearchByBoW([], frame)[0]) == 0
    empty = dict(desc=np.zeros((0, 32), np.uint8), angle=np.zeros(0, np.float32), valid=np.zeros(0, bool),
                 node_ids=np.zeros(0, np.int32), node_start=np.zeros(1, np.int32), node_items=np.zeros(0, np.int32))
    nm, m12, m21 = m.SearchByBoW([empty], frame)
    assert nm[0] == 0 and (m21 == -1).all()
    # index tables that point outside their arrays are refused before anything reaches the GPU
    for side, key, bad in (("kf", "node_items", len(kfs[3]["desc"])), ("kf", "node_items", -1),
                           ("frame", "node_items", n2), ("kf", "node_start", None)):
        k, f = dict(kfs[3]), dict(frame)
        tgt = k if side == "kf" else f
        arr = np.array(tgt[key], np.int32)
        if bad is None:
            arr[1] = arr[-1] + 5      # an offset beyond the item list
        else:
            arr[len(arr) // 2] = bad
        tgt[key] = arr
        with pytest.raises(orbx.OrbxError):
            m.SearchByBoW([k], f)


def test_frame_grid_and_area_queries_vs_oracle(oracle):
    """AssignFeaturesToGrid / GetFeaturesInArea: the CSR grid and the ordered result lists of a batch of
    queries (windows of several sizes, level filters, centres outside the image, keypoints outside the
    grid bounds) must equal the oracle's."""
    rng = np.random.default_rng(13)
    n = 3000
    kps = np.zeros(n, orbx.KEYPOINT_DTYPE)
    kps["x"] = rng.random(n).astype(np.float32) * 1300 - 30     # some outside [0, 1241)
    kps["y"] = rng.random(n).astype(np.float32) * 400 - 12
    kps["x"][:200] = np.round(kps["x"][:200])                   # exact cell borders and repeated positions
    kps["y"][:200] = np.round(kps["y"][:200] / 7.8333) * 7.8333
    kps["octave"] = rng.integers(0, 8, size=n)
    bounds = (0.0, 1241.0, 0.0, 376.0)
    nq = 500
    qx = (rng.random(nq) * 1400 - 80).astype(np.float32)
    qy = (rng.random(nq) * 450 - 40).astype(np.float32)
    qr = rng.choice([5.0, 10.0, 15.0, 40.0, 100.0, 700.0], size=nq).astype(np.float32)
    lo = rng.choice([-1, 0, 0, 2, 5], size=nq).astype(np.int32)
    hi = np.where(rng.random(nq) < 0.3, -1, lo + rng.integers(0, 3, size=nq)).astype(np.int32)
    m = orbx.ORBmatcher()
    got, cs, ci = m.GetFeaturesInArea(kps, bounds, qx, qy, qr, lo, hi, cap=3000)
    want, ocs, oci = oracle.grid_queries(kps, bounds, qx, qy, qr, lo, hi, cap=3000)
    assert np.array_equal(cs, ocs) and np.array_equal(ci, oci)
    assert cs[-1] < n            # keypoints outside the grid were dropped
    for q in range(nq):
        assert np.array_equal(got[q], want[q]), q
    assert max(len(g) for g in got) > 1000 and min(len(g) for g in got) == 0
    # no keypoints at all
    got, cs, _ = m.GetFeaturesInArea(kps[:0], bounds, qx[:3], qy[:3], qr[:3], lo[:3], hi[:3])
    assert cs[-1] == 0 and all(len(g) == 0 for g in got)


def _random_tree(rng, k, depth, early_leaf=0.1):
    """A vocabulary tree in flat form with up to k children per node, leaves at depth `depth` and a few earlier."""
    desc = [np.zeros(32, np.uint8)]
    kids = [[]]
    level = [0]
    frontier = [0]
    for lv in range(1, depth + 1):
        nxt = []
        for p in frontier:
            if lv > 1 and rng.random() < early_leaf:
                continue
            for _ in range(int(rng.integers(2, k + 1))):
                nid = len(desc)
                base = desc[p] if lv > 1 and rng.random() < 0.5 else rng.integers(0, 256, size=32, dtype=np.uint8)
                row = base.copy()
                row[rng.integers(0, 32)] ^= np.uint8(rng.integers(1, 256))  # siblings close to each other: ties
                desc.append(row)
                kids.append([])
                level.append(lv)
                kids[p].append(nid)
                nxt.append(nid)
        frontier = nxt
    n = len(desc)
    child_start = np.zeros(n + 1, np.int32)
    child_start[1:] = np.cumsum([len(c) for c in kids])
    children = np.array([c for cs in kids for c in cs], np.int32)
    word_id = np.full(n, -1, np.int32)
    leaves = [i for i in range(n) if not kids[i]]
    word_id[leaves] = np.arange(len(leaves))
    weight = np.where(word_id >= 0, rng.random(n) * 5, 0.0)
    weight[leaves[::17]] = 0.0          # stopped words
    return dict(desc=np.stack(desc), child_start=child_start, children=children, word_id=word_id, weight=weight,
                L=depth)


def test_vocabulary_transform_vs_oracle(oracle):
    rng = np.random.default_rng(33)
    tree = _random_tree(rng, 10, 4)
    voc = orbx.ORBVocabulary(tree["desc"], tree["child_start"], tree["children"], tree["word_id"], tree["weight"],
                             tree["L"])
    feats = rng.integers(0, 256, size=(3000, 32), dtype=np.uint8)
    feats[:500] = tree["desc"][rng.integers(1, len(tree["desc"]), size=500)]      # exact node descriptors
    for levelsup in (0, 1, 2, 4, 7):
        w, nd, wt = voc.transform_features(feats, levelsup)
        ow, ond, owt = oracle.vocabulary_transform(tree, feats, levelsup)
        assert np.array_equal(w, ow) and np.array_equal(nd, ond) and np.array_equal(wt, owt), levelsup
    assert (nd == 0).all()          # levelsup beyond the depth: the root
    bow, fv = voc.transform(feats, 2)
    assert abs(sum(bow.values()) - 1.0) < 1e-9 and sum(len(v) for v in fv.values()) == int((owt > 0).sum())
    # malformed trees are refused: a child that precedes its parent
    bad = dict(tree)
    bad["children"] = tree["children"].copy()
    bad["children"][-1] = 0
    with pytest.raises(orbx.OrbxError):
        orbx.ORBVocabulary(bad["desc"], bad["child_start"], bad["children"], bad["word_id"], bad["weight"], 4)


@pytest.mark.parametrize("dist", [(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05),        # EuRoC.yaml
                                  (0.262383, -0.953104, -0.005358, 0.002628, 1.163314),         # TUM1.yaml
                                  (0.0, 0.0, 0.0, 0.0)])                                         # KITTI: no distortion
def test_undistort_keypoints_vs_oracle(oracle, dist):
    """Frame::UndistortKeyPoints / ComputeImageBounds with the shipped camera models: the arithmetic is
    double with the oracle's operation order, so the float results must be equal bit for bit."""
    import torch
    rng = np.random.default_rng(11)
    n = 3000
    kps = np.zeros(n, orbx.KEYPOINT_DTYPE)
    kps["x"] = rng.uniform(0, 752, n).astype(np.float32)
    kps["y"] = rng.uniform(0, 480, n).astype(np.float32)
    kps["size"], kps["angle"], kps["octave"], kps["class_id"] = 31, rng.uniform(0, 360, n), rng.integers(0, 8, n), -1
    cam = (458.654, 457.296, 367.215, 248.375) if len(dist) == 4 else (517.306408, 516.469215, 318.643040, 255.313989)
    m = orbx.ORBmatcher()
    un = m.UndistortKeyPoints(kps, cam, dist)
    want = oracle.undistort_points(np.stack([kps["x"], kps["y"]], 1), cam, dist)
    assert np.array_equal(un["x"], want[:, 0]) and np.array_equal(un["y"], want[:, 1])
    for f in ("size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(un[f], kps[f])
    if dist[0] == 0:
        assert np.array_equal(un["x"], kps["x"])
    else:
        assert np.abs(un["x"] - kps["x"]).max() > 1.0      # the model moves points near the border by pixels
    # device form, in place
    d = torch.from_numpy(kps.view(np.uint8).reshape(n, -1).copy()).cuda()
    m.undistort_keypoints_device(d.data_ptr(), n, cam, dist, d.data_ptr())
    torch.cuda.synchronize()
    got = d.cpu().numpy().reshape(-1).view(orbx.KEYPOINT_DTYPE)
    assert np.array_equal(got["x"], want[:, 0]) and np.array_equal(got["y"], want[:, 1])
    assert np.array_equal(got["octave"], kps["octave"])
    # image bounds: min / max of the undistorted corners (src/Frame.cc:503-523)
    b = m.ComputeImageBounds(752, 480, cam, dist)
    c = oracle.undistort_points(np.array([[0, 0], [752, 0], [0, 480], [752, 480]], np.float32), cam, dist)
    if dist[0] == 0:
        assert b == (0.0, 752.0, 0.0, 480.0)
    else:
        assert b == (min(c[0, 0], c[2, 0]), max(c[1, 0], c[3, 0]), min(c[0, 1], c[1, 1]), max(c[2, 1], c[3, 1]))


def _projection_scene(seed, n=2500, ns=1800, stereo=True):
    """A frame of n features and ns sources cut from them: projections a few pixels off, descriptors a few
    bits off, levels around the feature's; clusters of sources aim at the same feature so that the
    first-come exclusion decides, some features hold a map point already, some sources do not block."""
    rng = np.random.default_rng(seed)
    kps = np.zeros(n, orbx.KEYPOINT_DTYPE)
    kps["x"] = rng.uniform(0, 1241, n).astype(np.float32)
    kps["y"] = rng.uniform(0, 376, n).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, n)
    kps["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    kps["size"], kps["class_id"] = 31, -1
    desc = orbx.synth_descriptors(seed, n)
    uright = np.where(rng.random(n) < 0.6, kps["x"] - rng.uniform(1, 60, n), -1).astype(np.float32) if stereo else None
    taken = (rng.random(n) < 0.15).astype(np.uint8)
    src = rng.integers(0, n, ns)
    src[1::7] = src[0:-1:7][:len(src[1::7])]          # neighbours in the list aim at the same feature
    x = (kps["x"][src] + rng.normal(0, 3, ns)).astype(np.float32)
    y = (kps["y"][src] + rng.normal(0, 3, ns)).astype(np.float32)
    lvl = np.clip(kps["octave"][src] + rng.integers(-1, 2, ns), 0, 7).astype(np.int32)
    sd = desc[src].copy()
    for i in range(ns):
        bits = np.unpackbits(sd[i])
        bits[rng.choice(256, size=int(rng.integers(0, 70)), replace=False)] ^= 1
        sd[i] = np.packbits(bits)
    xr = (x - (kps["x"][src] - np.where(uright[src] > 0, uright[src], 0)) + rng.normal(0, 2, ns)).astype(np.float32) \
        if stereo else None
    valid = (rng.random(ns) < 0.9).astype(np.uint8)
    blocks = (rng.random(ns) < 0.85).astype(np.uint8)
    angle = ((kps["angle"][src] + rng.normal(0, 6, ns) + np.where(rng.random(ns) < 0.1, 140, 0)) % 360).astype(np.float32)
    view_cos = rng.uniform(0.99, 1.0, ns).astype(np.float32)
    return dict(kps=kps, desc=desc, uright=uright, taken=taken, x=x, y=y, lvl=lvl, sd=sd, xr=xr, valid=valid,
                blocks=blocks, angle=angle, view_cos=view_cos)


@pytest.mark.parametrize("stereo,th", [(True, 1.0), (False, 3.0), (True, 5.0)])
def test_search_by_projection_map_points_vs_oracle(oracle, stereo, th):
    """src/ORBmatcher.cc:46-137 (TrackLocalMap): matches, feature assignments and the count equal the oracle's."""
    sc = _projection_scene(21 + int(th), stereo=stereo)
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    bounds = (0.0, 1241.0, 0.0, 376.0)
    m = orbx.ORBmatcher(0.8)
    nm, match, fm = m.SearchByProjectionMapPoints(sc["kps"], sc["desc"], bounds, scale, sc["valid"], sc["x"], sc["y"],
                                                  sc["lvl"], sc["view_cos"], sc["sd"], th, uright=sc["uright"],
                                                  proj_xr=sc["xr"], taken=sc["taken"], blocks=sc["blocks"])
    r = np.where(sc["view_cos"] > np.float32(0.998), np.float32(2.5), np.float32(4.0))
    if th != 1.0:
        r = r * np.float32(th)
    r = (r.astype(np.float32) * scale[sc["lvl"]]).astype(np.float32)
    onm, omatch, ofm = oracle.search_by_projection(sc["kps"], sc["desc"], bounds, sc["x"], sc["y"], r, sc["lvl"] - 1,
                                                   sc["lvl"], sc["sd"], 0.8, uright=sc["uright"], taken=sc["taken"],
                                                   valid=sc["valid"], xr=sc["xr"], blocks=sc["blocks"], use_second=True)
    assert nm == onm and nm > 300
    assert np.array_equal(match, omatch) and np.array_equal(fm, ofm)
    assert (match[sc["valid"] == 0] == -1).all()
    assert (sc["taken"][match[match >= 0]] == 0).all()


@pytest.mark.parametrize("mode,ori", [("none", True), ("forward", True), ("backward", False)])
def test_search_by_projection_last_frame_vs_oracle(oracle, mode, ori):
    """src/ORBmatcher.cc:1412-1566 (TrackWithMotionModel): best-only search, level window by motion direction,
    rotation histogram with factor HISTO_LENGTH/360."""
    sc = _projection_scene(31, stereo=(mode != "none"))
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    bounds = (0.0, 1241.0, 0.0, 376.0)
    m = orbx.ORBmatcher(0.9)
    nm, match, fm = m.SearchByProjectionLastFrame(sc["kps"], sc["desc"], bounds, scale, sc["valid"], sc["x"], sc["y"],
                                                  sc["lvl"], sc["angle"], sc["sd"], 7.0, forward=mode == "forward",
                                                  backward=mode == "backward", uright=sc["uright"], ur=sc["xr"],
                                                  taken=sc["taken"], blocks=sc["blocks"], check_orientation=ori)
    r = (np.float32(7.0) * scale[sc["lvl"]]).astype(np.float32)
    if mode == "forward":
        lo, hi = sc["lvl"], np.full_like(sc["lvl"], -1)
    elif mode == "backward":
        lo, hi = np.zeros_like(sc["lvl"]), sc["lvl"]
    else:
        lo, hi = sc["lvl"] - 1, sc["lvl"] + 1
    onm, omatch, ofm = oracle.search_by_projection(sc["kps"], sc["desc"], bounds, sc["x"], sc["y"], r, lo, hi, sc["sd"],
                                                   0.9, uright=sc["uright"], taken=sc["taken"], valid=sc["valid"],
                                                   xr=sc["xr"], blocks=sc["blocks"], src_angle=sc["angle"],
                                                   use_second=False, check_orientation=ori)
    assert nm == onm and nm > 300
    assert np.array_equal(match, omatch) and np.array_equal(fm, ofm)
    if ori:
        assert (fm == -2).sum() > 10        # the planted 140-degree rotations fall outside the three maxima


@pytest.mark.parametrize("ori,orbdist", [(True, 100), (True, 64), (False, 100)])
def test_search_by_projection_key_frame_form_vs_oracle(oracle, ori, orbdist):
    """src/ORBmatcher.cc:1568-1690 (Tracking::Relocalization): best only against ORBdist, features that hold any
    map point skipped, rotation histogram with factor 1/HISTO_LENGTH (13 bins in use)."""
    sc = _projection_scene(41, stereo=False)
    scale = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    bounds = (0.0, 1241.0, 0.0, 376.0)
    lvl = np.clip(sc["lvl"], 0, 6)
    # rotations spread over the whole circle so that more than three of the 13 bins fill
    rng = np.random.default_rng(5)
    angle = ((sc["angle"] + rng.choice([0, 0, 0, 40, 80, 170, 250], size=len(sc["angle"]))) % 360).astype(np.float32)
    m = orbx.ORBmatcher(0.9, ori)
    nm, match, fm = m.SearchByProjectionKeyFrame(sc["kps"], sc["desc"], bounds, scale, sc["valid"], sc["x"], sc["y"], lvl,
                                                 angle, sc["sd"], 10.0, orbdist, taken=sc["taken"])
    r = (np.float32(10.0) * scale[lvl]).astype(np.float32)
    onm, omatch, ofm = oracle.search_by_projection(sc["kps"], sc["desc"], bounds, sc["x"], sc["y"], r, lvl - 1, lvl + 1,
                                                   sc["sd"], 0.9, taken=sc["taken"], valid=sc["valid"], src_angle=angle,
                                                   use_second=False, th_dist=orbdist, check_orientation=2 if ori else 0)
    assert nm == onm and nm > 300
    assert np.array_equal(match, omatch) and np.array_equal(fm, ofm)
    if ori:
        assert (fm == -2).sum() > 10
        # the other histogram would have cleared a different set
        _, _, ofm1 = oracle.search_by_projection(sc["kps"], sc["desc"], bounds, sc["x"], sc["y"], r, lvl - 1, lvl + 1,
                                                 sc["sd"], 0.9, taken=sc["taken"], valid=sc["valid"], src_angle=angle,
                                                 use_second=False, th_dist=orbdist, check_orientation=1)
        assert not np.array_equal(ofm, ofm1)


def test_search_by_projection_edges(oracle):
    """No sources, no features, and a window that holds more candidates than the kernel's lists (an error,
    not a wrong answer)."""
    m = orbx.ORBmatcher(0.8)
    kps = np.zeros(300, orbx.KEYPOINT_DTYPE)
    kps["x"] = 100 + (np.arange(300) % 20)
    kps["y"] = 100 + (np.arange(300) // 20)
    desc = orbx.synth_descriptors(2, 300)
    bounds = (0.0, 640.0, 0.0, 480.0)
    e = np.zeros(0, np.float32)
    ei = np.zeros(0, np.int32)
    nm, match, fm = m.search_by_projection(kps, desc, bounds, e, e, e, ei, ei, np.zeros((0, 32), np.uint8))
    assert nm == 0 and len(match) == 0 and (fm == -1).all()
    one = np.array([105.0], np.float32)
    nm, match, fm = m.search_by_projection(kps[:0], desc[:0], bounds, one, one, np.array([5.0], np.float32),
                                           np.array([0], np.int32), np.array([-1], np.int32), desc[:1])
    assert nm == 0 and list(match) == [-1] and len(fm) == 0
    # 300 features inside one window: the lists hold 256
    with pytest.raises(orbx.OrbxError):
        m.search_by_projection(kps, desc, bounds, np.array([110.0], np.float32), np.array([107.0], np.float32),
                               np.array([60.0], np.float32), np.array([0], np.int32), np.array([-1], np.int32), desc[:1])
    # 200 of them: the crowded-window path (more than the 32 staged candidates) agrees with the oracle
    x, y, r = np.array([110.0], np.float32), np.array([104.5], np.float32), np.array([30.0], np.float32)
    sub = kps[:200]
    nm, match, fm = m.search_by_projection(sub, desc[:200], bounds, x, y, r, np.array([0], np.int32),
                                           np.array([-1], np.int32), desc[150:151], use_second=False)
    onm, omatch, ofm = oracle.search_by_projection(sub, desc[:200], bounds, x, y, r, np.array([0], np.int32),
                                                   np.array([-1], np.int32), desc[150:151], 0.8, use_second=False)
    assert nm == onm == 1 and list(match) == list(omatch) == [150] and np.array_equal(fm, ofm)


def test_rccl_sharded_matcher_c_abi(oracle):
    """orbx_sharded_* (the C path: top-2, ncclAllGather on the same stream, merge) with a real RCCL communicator
    of one rank, from Python and from the C++ program (which also checks a host loop with the reference's
    bookkeeping, src/ORBmatcher.cc:217-242).  More ranks need more GPUs: RCCL takes one rank per device."""
    import subprocess

    import torch
    from orbslam2_amd.sharded import RcclShardedMatcher
    ndb, nq = 30011, 501
    db, q = make_db(ndb, nq, seed=9)
    db[25000:25003] = q[3]
    d_db, d_q = torch.from_numpy(db).cuda(), torch.from_numpy(q).cuda()
    best = torch.zeros(nq, dtype=torch.int32, device="cuda")
    d1 = torch.zeros(nq, dtype=torch.int16, device="cuda")
    d2 = torch.zeros(nq, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    sm = RcclShardedMatcher(0, 1, 0, max_queries=nq)
    ob, o1, o2 = oracle.match_top2(q, db)
    for n in (nq, 77):  # the second call smaller: the blocks are laid out for the call's own query count
        sm.match_device(d_q.data_ptr(), n, d_db.data_ptr(), ndb, 0, best.data_ptr(), d1.data_ptr(), d2.data_ptr())
        sm.synchronize()
        assert np.array_equal(best.cpu().numpy()[:n], ob[:n])
        assert np.array_equal(d1.cpu().numpy().view(np.uint16)[:n], o1[:n])
        assert np.array_equal(d2.cpu().numpy().view(np.uint16)[:n], o2[:n])
    sm.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "test_sharded_rccl")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    out = subprocess.run([exe, "1", "333", "20000"], capture_output=True, text=True, timeout=110)
    # RCCL prints a version banner on the same stream first
    assert out.returncode == 0 and any(ln.startswith("OK world=1") for ln in out.stdout.splitlines()), \
        out.stdout + out.stderr


def test_hand_worked_cases_through_the_hip_path():
    """The paper cases of tests/kat_cases.py (the only pins of SearchForInitialization and GetFeaturesInArea that
    do not come from the oracle) through the C ABI: host-array forms, and the device batch form of the search."""
    import torch
    import kat_cases as K
    from orbslam2_amd import _lib as L
    kps, cells, queries = K.features_in_area_case()
    q = np.array([a for a, _ in queries], np.float32)
    m = orbx.ORBmatcher(0.9, True)
    lists, cs, ci = m.GetFeaturesInArea(kps, K.BOUNDS, q[:, 0], q[:, 1], q[:, 2], q[:, 3].astype(np.int32),
                                        q[:, 4].astype(np.int32))
    for (args, want), got in zip(queries, lists):
        assert list(got) == want, args
    for (c, r), items in cells.items():
        assert list(ci[cs[c * 48 + r]:cs[c * 48 + r + 1]]) == items
    assert cs[-1] == 8
    k1, d1, k2, d2, win, ratio, no_ori, with_ori = K.search_for_initialization_case()
    for ori, (want_n, want_m) in ((False, no_ori), (True, with_ori)):
        mt = orbx.ORBmatcher(ratio, ori)
        prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], axis=1), np.float32)
        before = prev.copy()
        n, m12, pairs = mt.SearchForInitialization(k1, d1, k2, d2, K.BOUNDS, prev, win)
        assert n == want_n and list(m12) == want_m and pairs == 15
        for i, j in enumerate(want_m):
            assert tuple(prev[i]) == ((k2["x"][j], k2["y"][j]) if j >= 0 else tuple(before[i]))
        # device batch form: the two frames as a batch of two in the extractor's result layout
        cap = 16
        kp = np.zeros((2, cap), L.KEYPOINT_DTYPE)
        de = np.zeros((2, cap, 32), np.uint8)
        kp[0, :len(k1)], kp[1, :len(k2)] = k1, k2
        de[0, :len(k1)], de[1, :len(k2)] = d1, d2
        d_kp = torch.from_numpy(kp.view(np.uint8).reshape(2, -1)).cuda()
        d_de = torch.from_numpy(de).cuda()
        d_cnt = torch.tensor([len(k1), len(k2)], dtype=torch.int32, device="cuda")
        o12 = torch.full((1, cap), -7, dtype=torch.int32, device="cuda")
        onm = torch.zeros(1, dtype=torch.int32, device="cuda")
        onp = torch.zeros(1, dtype=torch.int32, device="cuda")
        pout = torch.full((1, cap, 2), -1.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        mt.set_stream(torch.cuda.current_stream().cuda_stream)
        mt.search_for_initialization_batch_device(d_kp.data_ptr(), d_de.data_ptr(), d_cnt.data_ptr(), cap, 2, 16,
                                                  K.BOUNDS, o12.data_ptr(), onm.data_ptr(), onp.data_ptr(),
                                                  windowSize=win, d_prev_matched_out=pout.data_ptr())
        torch.cuda.synchronize()
        assert int(onm.item()) == want_n and list(o12.cpu().numpy()[0, :len(k1)]) == want_m
        assert int(onp.item()) == 15
        po = pout.cpu().numpy()[0]
        for i, j in enumerate(want_m):
            if j >= 0:
                assert tuple(po[i]) == (k2["x"][j], k2["y"][j])


@pytest.mark.parametrize("strict,ratio,ori", [(False, 0.75, True), (True, 0.8, True), (False, 0.7, False)])
def test_search_by_bow_device_chain_vs_oracle(oracle, strict, ratio, ori):
    """Relocalisation-style chain without a host hop: descriptors in device memory -> vocabulary transform ->
    feature vectors (sort by node id) -> SearchByBoW, all through device-pointer entry points; against the
    oracle's transform, a numpy FeatureVector (ascending node ids, features in index order) and the oracle's
    SearchByBoW per key frame."""
    import torch
    from orbslam2_amd import _lib as L
    rng = np.random.default_rng(77 + int(strict))
    tree = _random_tree(rng, 8, 5, early_leaf=0.05)
    voc = orbx.ORBVocabulary(tree["desc"], tree["child_start"], tree["children"], tree["word_id"], tree["weight"],
                             tree["L"])
    nb, cap1, cap2, n2 = 5, 700, 1600, 1500
    leaves = np.flatnonzero(tree["word_id"] >= 0)
    d2 = tree["desc"][rng.choice(leaves, size=n2)].copy()
    d2[:, 5:17] ^= rng.integers(0, 256, size=(n2, 12), dtype=np.uint8)    # near a leaf, so nodes are shared
    d2[:400] = tree["desc"][leaves[3]]                                      # one crowded node
    d2[:400, 20:] = rng.integers(0, 256, size=(400, 12), dtype=np.uint8)
    counts1 = np.array([700, 0, 333, 1, 650], np.int32)
    kp1 = np.zeros((nb, cap1), L.KEYPOINT_DTYPE)
    de1 = rng.integers(0, 256, size=(nb, cap1, 32), dtype=np.uint8)         # rows past the count: garbage
    kp1["angle"] = (rng.random((nb, cap1)) * 360).astype(np.float32)
    kp2 = np.zeros(cap2, L.KEYPOINT_DTYPE)
    kp2["angle"] = (rng.random(cap2) * 360).astype(np.float32)
    for b in range(nb):
        pick = rng.integers(0, n2, size=counts1[b])
        rows = d2[pick].copy()
        rows[:, 28:] ^= rng.integers(0, 4, size=(counts1[b], 4), dtype=np.uint8)
        fresh = rng.random(counts1[b]) < 0.15
        rows[fresh] = rng.integers(0, 256, size=(int(fresh.sum()), 32), dtype=np.uint8)
        de1[b, :counts1[b]] = rows
        # a key frame is the frame seen rotated by 30 degrees, give or take: most matches share a few rotation bins
        kp1["angle"][b, :counts1[b]] = (kp2["angle"][pick] + 30 + rng.normal(0, 6, size=counts1[b])) % 360
    de2 = rng.integers(0, 256, size=(cap2, 32), dtype=np.uint8)
    de2[:n2] = d2
    valid1 = (rng.random((nb, cap1)) < 0.85).astype(np.uint8)
    valid2 = (rng.random(cap2) < 0.9).astype(np.uint8) if strict else None
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()  # noqa: E731
    t_kp1, t_de1, t_c1, t_v1 = dev(kp1), dev(de1), dev(counts1), dev(valid1)
    t_kp2, t_de2, t_c2 = dev(kp2), dev(de2), dev(np.array([n2], np.int32))
    t_v2 = dev(valid2) if valid2 is not None else None
    i32 = lambda n: torch.zeros(n, dtype=torch.int32, device="cuda")  # noqa: E731
    word1, node1, word2, node2 = i32(nb * cap1), i32(nb * cap1), i32(cap2), i32(cap2)
    wt1 = torch.zeros(nb * cap1, dtype=torch.float64, device="cuda")
    wt2 = torch.zeros(cap2, dtype=torch.float64, device="cuda")
    m = orbx.ORBmatcher(ratio, ori)
    st = torch.cuda.current_stream().cuda_stream
    m.set_stream(st)
    voc.transform_features_device(t_de1.data_ptr(), nb * cap1, word1.data_ptr(), node1.data_ptr(), wt1.data_ptr(),
                                  levelsup=2, stream=st)
    voc.transform_features_device(t_de2.data_ptr(), cap2, word2.data_ptr(), node2.data_ptr(), wt2.data_ptr(),
                                  levelsup=2, stream=st)
    node1 = torch.where(wt1 > 0, node1, torch.full_like(node1, -1))   # a word of weight 0 is in no node
    node2 = torch.where(wt2 > 0, node2, torch.full_like(node2, -1))
    ids1, start1, items1, num1 = i32(nb * cap1), i32(nb * (cap1 + 1)), i32(nb * cap1), i32(nb)
    ids2, start2, items2, num2 = i32(cap2), i32(cap2 + 1), i32(cap2), i32(1)
    m.feature_vectors_build_device(node1.data_ptr(), t_c1.data_ptr(), nb, cap1, ids1.data_ptr(), start1.data_ptr(),
                                   items1.data_ptr(), num1.data_ptr())
    m.feature_vectors_build_device(node2.data_ptr(), t_c2.data_ptr(), 1, cap2, ids2.data_ptr(), start2.data_ptr(),
                                   items2.data_ptr(), num2.data_ptr())
    m12, m21, nm = i32(nb * cap1), i32(nb * cap2), i32(nb)
    m.search_by_bow_device((t_kp1.data_ptr(), t_de1.data_ptr(), t_c1.data_ptr(), t_v1.data_ptr(), ids1.data_ptr(),
                            start1.data_ptr(), items1.data_ptr()), nb, cap1,
                           (t_kp2.data_ptr(), t_de2.data_ptr(), t_v2.data_ptr() if t_v2 is not None else 0,
                            ids2.data_ptr(), start2.data_ptr(), items2.data_ptr()), cap2,
                           m12.data_ptr(), m21.data_ptr(), nm.data_ptr(), strict_lt=strict)
    torch.cuda.synchronize()
    # expected: oracle transform, numpy feature vectors, oracle search per key frame
    def fv_of(desc):
        _, node, wt = oracle.vocabulary_transform(tree, desc, 2)
        keep = np.flatnonzero(wt > 0)
        ids, start, order = _feature_vector(node[keep])
        return ids, start, keep[order].astype(np.int32)
    e_ids2, e_start2, e_items2 = fv_of(d2)
    E2 = int(num2.item())
    assert E2 == len(e_ids2)
    assert np.array_equal(ids2.cpu().numpy()[:E2], e_ids2) and (ids2.cpu().numpy()[E2:] == 0x7fffffff).all()
    assert np.array_equal(start2.cpu().numpy()[:E2 + 1], e_start2) and (start2.cpu().numpy()[E2:] == e_start2[-1]).all()
    assert np.array_equal(items2.cpu().numpy()[:e_start2[-1]], e_items2)
    h12, h21, hn = m12.cpu().numpy().reshape(nb, cap1), m21.cpu().numpy().reshape(nb, cap2), nm.cpu().numpy()
    hid1, hst1, hit1 = (ids1.cpu().numpy().reshape(nb, cap1), start1.cpu().numpy().reshape(nb, cap1 + 1),
                        items1.cpu().numpy().reshape(nb, cap1))
    total = 0
    for b in range(nb):
        n1 = int(counts1[b])
        f_ids, f_start, f_items = fv_of(de1[b, :n1]) if n1 else (np.zeros(0, np.int32), np.zeros(1, np.int32),
                                                                 np.zeros(0, np.int32))
        E1 = int(num1.cpu().numpy()[b])
        assert E1 == len(f_ids) and np.array_equal(hid1[b, :E1], f_ids) and np.array_equal(hst1[b, :E1 + 1], f_start)
        assert np.array_equal(hit1[b, :f_start[-1]], f_items)
        if n1 == 0:
            assert hn[b] == 0 and (h21[b] == -1).all()
            continue
        en, e12, e21 = oracle.search_by_bow(de1[b, :n1], kp1["angle"][b, :n1], valid1[b, :n1], (f_ids, f_start, f_items),
                                            d2, kp2["angle"][:n2], None if valid2 is None else valid2[:n2],
                                            (e_ids2, e_start2, e_items2), strict, ratio, ori)
        assert hn[b] == en, b
        assert np.array_equal(h12[b, :n1], e12) and (h12[b, n1:] == -1).all()
        assert np.array_equal(h21[b, :n2], e21) and (h21[b, n2:] == -1).all()
        total += en
    assert total > 200  # (the orientation check and the valid flags thin the matches out)
    voc.close()
