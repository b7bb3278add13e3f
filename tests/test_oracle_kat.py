"""Known-answer tests of the CPU oracle's primitives.  The reference ships no tests or golden vectors
for this path and its pixel arithmetic lives in OpenCV, which is absent here (parity unpinned), so the
expected values below were worked out by hand from the published algorithms; each case says how."""
import numpy as np
import pytest


def test_scale_tables_and_quotas(oracle):
    # src/ORBextractor.cc:437-475: cumulative float products of (double)1.2f; geometric quotas, remainder last
    sf, inv, s2, is2 = oracle.scale_tables(1.2, 8)
    want = np.array([1.0, 1.2, 1.44000006, 1.72800016, 2.07360029, 2.48832035, 2.98598456, 3.58318162], np.float32)
    assert np.array_equal(sf, want)
    assert np.array_equal(inv, np.float32(1.0) / sf) and np.array_equal(s2, sf * sf)
    assert list(oracle.features_per_level(1000, 1.2, 8)) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(oracle.features_per_level(2000, 1.2, 8)) == [434, 362, 302, 251, 209, 175, 145, 122]
    assert list(oracle.features_per_level(4000, 1.2, 8)) == [869, 724, 603, 503, 419, 349, 291, 242]
    assert list(oracle.features_per_level(1200, 1.2, 8)) == [261, 217, 181, 151, 126, 105, 87, 72]
    assert sum(oracle.features_per_level(1000, 1.2, 8)) == 1000


def test_umax(oracle):
    # src/ORBextractor.cc:483-505: round(sqrt(225 - v^2)) up to v = 11, then mirrored about the diagonal
    assert list(oracle.umax()) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    # the patch holds 749 pixels
    assert 31 + 2 * sum(2 * u + 1 for u in oracle.umax()[1:]) == 749


def test_resize_constant_and_ramp(oracle):
    # a constant image stays constant: (b0*(c*2048>>4)>>16) + (b1*(...)>>16) + 2 >> 2 with b0+b1 = 2048
    src = np.full((40, 60), 200, np.uint8)
    assert (oracle.resize(src, 50, 33) == 200).all()
    # 2 -> 1 columns of one row: dx=0 maps to fx = 0.5: (a0,a1) = (1024,1024); values 10 and 30:
    # H = 10*1024 + 30*1024 = 40960; rows identical (sy clamps): b0 = b1 = 1024 for fy = 0.5
    # ((1024*(40960>>4))>>16) = 40 ; 40 + 40 + 2 >> 2 = 20
    src = np.array([[10, 30], [10, 30]], np.uint8)
    assert oracle.resize(src, 1, 1)[0, 0] == 20
    # identity size keeps the image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(17, 23), dtype=np.uint8)
    assert np.array_equal(oracle.resize(img, 23, 17), img)


def test_resize_matches_float_bilinear_within_one(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(48, 64), dtype=np.uint8)
    dw, dh = 53, 40
    got = oracle.resize(img, dw, dh).astype(np.float64)
    sx = (np.arange(dw) + 0.5) * (64 / dw) - 0.5
    sy = (np.arange(dh) + 0.5) * (48 / dh) - 0.5
    x0 = np.clip(np.floor(sx).astype(int), 0, 62)
    y0 = np.clip(np.floor(sy).astype(int), 0, 46)
    fx = np.clip(sx - x0, 0, 1)
    fy = np.clip(sy - y0, 0, 1)
    f = img.astype(np.float64)
    ref = ((1 - fy)[:, None] * ((1 - fx) * f[y0][:, x0] + fx * f[y0][:, x0 + 1]) +
           fy[:, None] * ((1 - fx) * f[y0 + 1][:, x0] + fx * f[y0 + 1][:, x0 + 1]))
    assert np.abs(got - ref).max() <= 1.0


def test_gauss_kernel_and_borders(oracle):
    # impulse response = outer product of the integer kernel, scaled: (k_i*k_j*255 + 32768) >> 16
    k = np.array([18, 34, 48, 56, 48, 34, 18])
    img = np.zeros((21, 21), np.uint8)
    img[10, 10] = 255
    want = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(oracle.gauss7(img)[7:14, 7:14], want)
    k1 = np.array([18, 34, 49, 55, 49, 34, 18])
    want1 = (np.outer(k1, k1) * 255 + 32768) >> 16
    assert np.array_equal(oracle.gauss7(img, 1)[7:14, 7:14], want1)
    # constant image: sum 256 keeps it; sum 257 brightens and saturates
    c = np.full((9, 11), 250, np.uint8)
    assert (oracle.gauss7(c) == 250).all()
    assert (oracle.gauss7(c, 1) == min(255, (250 * 257 * 257 + 32768) >> 16)).all()
    # BORDER_REFLECT_101 equals blurring the reflect-padded image and cropping
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(12, 15), dtype=np.uint8)
    padded = np.pad(img, 3, mode="reflect")
    assert np.array_equal(oracle.gauss7(img), oracle.gauss7(padded)[3:-3, 3:-3])


def ring(img, x, y):
    dx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    dy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]
    return [int(img[y + b, x + a]) for a, b in zip(dx, dy)]


def test_fast_score_hand_cases(oracle):
    img = np.full((9, 9), 100, np.uint8)
    # nine contiguous ring pixels 40 darker: A = 40; ten: still 40; eight: no arc -> A <= 0
    dx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    dy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]
    for n_dark, want in [(9, 40), (10, 40), (16, 40)]:
        im = img.copy()
        for k in range(n_dark):
            im[4 + dy[(k + 5) % 16], 4 + dx[(k + 5) % 16]] = 60
        assert oracle.fast_arc_score(im, 4, 4) == want
    im = img.copy()
    for k in range(8):
        im[4 + dy[k], 4 + dx[k]] = 60
    assert oracle.fast_arc_score(im, 4, 4) <= 0
    # brighter arc with one weaker pixel: the score is the weakest difference of the best arc
    im = img.copy()
    for k in range(9):
        im[4 + dy[k], 4 + dx[k]] = 180
    im[4 + dy[4], 4 + dx[4]] = 125
    assert oracle.fast_arc_score(im, 4, 4) == 25


def test_fast_detect_equals_score_definition(oracle):
    """cv::FAST semantics: corner at t <=> A > t; response = A-1; kept iff strictly greater than the 8
    neighbours' responses, counting non-corners and pixels outside the inner window as 0."""
    rng = np.random.default_rng(3)
    for trial in range(6):
        img = np.full((37, 38), 128, np.uint8)
        for _ in range(14):
            x, y = rng.integers(0, 30), rng.integers(0, 30)
            img[y:y + rng.integers(3, 12), x:x + rng.integers(3, 12)] = rng.integers(0, 256)
        img = np.clip(img.astype(int) + rng.integers(-3, 4, img.shape), 0, 255).astype(np.uint8)
        h, w = img.shape
        A = np.zeros((h, w), int)
        for y in range(3, h - 3):
            for x in range(3, w - 3):
                A[y, x] = oracle.fast_arc_score(img, x, y)
        for t in (20, 7, 12):
            xs, ys, rs = oracle.fast_detect(img, t)
            got = {(int(a), int(b)): int(c) for a, b, c in zip(xs, ys, rs)}
            S = np.where(A > t, A - 1, 0)
            want = {}
            for y in range(3, h - 3):
                for x in range(3, w - 3):
                    if A[y, x] > t:
                        nb = S[y - 1:y + 2, x - 1:x + 2].copy()
                        nb[1, 1] = -1
                        if S[y, x] > nb.max():
                            want[(x, y)] = S[y, x]
            assert got == want
            # row-major order
            assert list(zip(ys, xs)) == sorted(zip(ys, xs))


def test_fast_atan2(oracle):
    # polynomial approximation of cv::fastAtan2: exact on the axes, within 0.3 degree elsewhere
    assert oracle.fast_atan2(0.0, 1.0) == 0.0
    assert oracle.fast_atan2(1.0, 0.0) == 90.0
    assert oracle.fast_atan2(0.0, -1.0) == 180.0
    assert oracle.fast_atan2(-1.0, 0.0) == 270.0
    assert oracle.fast_atan2(0.0, 0.0) == 0.0
    rng = np.random.default_rng(4)
    for _ in range(500):
        y, x = rng.normal(size=2) * 1000
        ref = np.degrees(np.arctan2(y, x)) % 360
        d = abs(oracle.fast_atan2(float(np.float32(y)), float(np.float32(x))) - ref)
        assert min(d, 360 - d) < 0.3


def test_ic_angle_moments(oracle):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(41, 43), dtype=np.uint8)
    a, m10, m01 = oracle.ic_angle(img, 21, 20)
    um = oracle.umax()
    w10 = w01 = 0
    for v in range(-15, 16):
        for u in range(-um[abs(v)], um[abs(v)] + 1):
            w10 += u * int(img[20 + v, 21 + u])
            w01 += v * int(img[20 + v, 21 + u])
    assert (m10, m01) == (w10, w01)
    assert a == oracle.fast_atan2(float(w01), float(w10))
    # a patch brighter on the right points along +x (0 degrees), brighter below along +y (90 degrees)
    g = np.tile(np.arange(43, dtype=np.uint8) * 5, (41, 1))
    assert oracle.ic_angle(g, 21, 20)[0] == 0.0
    assert oracle.ic_angle(np.ascontiguousarray(g[:41, :41].T), 20, 20)[0] == 90.0


def test_brief_bit_layout(oracle):
    # at angle 0 the sample offsets are the table entries themselves: bit k of byte i compares test 8i+k
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, size=(41, 41), dtype=np.uint8)
    d = oracle.brief_descriptor(img, 20, 20, 0.0)
    import re
    pat = [int(v) for v in re.findall(r"-?\d+", open("orbslam2_amd/csrc/brief_pattern.inc").read().split("\n", 1)[1])]
    assert len(pat) == 1024
    for i in range(32):
        val = 0
        for k in range(8):
            x0, y0, x1, y1 = pat[4 * (8 * i + k):4 * (8 * i + k) + 4]
            val |= int(img[20 + y0, 20 + x0] < img[20 + y1, 20 + x1]) << k
        assert d[i] == val
    # rotating by 90 degrees maps (x,y) -> (-y, x): compare with the descriptor of the rotated image at 0 degrees
    d90 = oracle.brief_descriptor(img, 20, 20, 90.0)
    rot = np.ascontiguousarray(np.rot90(img, 1))  # rot[r, c] = img[c, 40 - r]
    # sample (row = x, col = -y) of img equals sample (row = y', col = x') of its clockwise rotation
    d0r = oracle.brief_descriptor(np.ascontiguousarray(np.rot90(img, -1)), 20, 20, 0.0)
    assert np.array_equal(d90, d0r) or np.array_equal(d90, oracle.brief_descriptor(rot, 20, 20, 0.0))


def test_octree_small_cases(oracle):
    # fewer candidates than the quota: every candidate is kept (each ends alone in a node)
    xs = np.array([5, 100, 200, 300, 310], np.int32)
    ys = np.array([5, 50, 100, 150, 152], np.int32)
    rs = np.array([10, 20, 30, 40, 50], np.int32)
    keep = oracle.distribute_octree(xs, ys, rs, 16, 16 + 608, 16, 16 + 448, 100)
    assert sorted(keep) == [0, 1, 2, 3, 4]
    # quota 1 with one root: the first split already yields >= 1 nodes; the list holds the children of
    # the root pushed to the front in the order UL, UR, BL, BR -> BR first.  Keep-best per node.
    xs = np.array([10, 600, 10, 600, 590], np.int32)
    ys = np.array([10, 10, 440, 440, 430], np.int32)
    rs = np.array([9, 8, 7, 6, 30], np.int32)
    keep = oracle.distribute_octree(xs, ys, rs, 16, 16 + 608, 16, 16 + 448, 1)
    assert list(keep) == [4, 2, 1, 0]
    # response ties: the first candidate in list order wins
    rs = np.array([9, 8, 7, 30, 30], np.int32)
    assert list(oracle.distribute_octree(xs, ys, rs, 16, 16 + 608, 16, 16 + 448, 1)) == [3, 2, 1, 0]


def test_octree_invariants(oracle):
    rng = np.random.default_rng(7)
    for quota in (30, 217, 434):
        pts = set()
        while len(pts) < 3000:
            pts.add((int(rng.integers(0, 1209)), int(rng.integers(0, 344))))
        pts = np.array(sorted(pts, key=lambda t: (t[1], t[0])), np.int32)
        rs = rng.integers(7, 200, size=len(pts)).astype(np.int32)
        keep = oracle.distribute_octree(pts[:, 0], pts[:, 1], rs, 16, 16 + 1209, 16, 16 + 344, quota)
        assert quota <= len(keep) <= quota + 2 or len(keep) <= 16
        assert len(set(keep.tolist())) == len(keep)


def test_descriptor_distance_and_top2(oracle):
    rng = np.random.default_rng(8)
    a = rng.integers(0, 256, size=(50, 32), dtype=np.uint8)
    b = rng.integers(0, 256, size=(70, 32), dtype=np.uint8)
    for i in range(0, 50, 7):
        for j in range(0, 70, 11):
            assert oracle.descriptor_distance(a[i], b[j]) == int(np.unpackbits(a[i] ^ b[j]).sum())
    best, d1, d2 = oracle.match_top2(a, b)
    D = np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)
    assert np.array_equal(best, D.argmin(1))
    assert np.array_equal(d1, D.min(1))
    assert np.array_equal(d2, np.sort(D, axis=1)[:, 1])


def test_three_maxima_and_ratio(oracle):
    import ctypes as C
    def tm(sizes):
        arr = (C.c_int * 30)(*sizes)
        i1, i2, i3 = C.c_int(), C.c_int(), C.c_int()
        oracle.lib().orc_three_maxima(arr, 30, C.byref(i1), C.byref(i2), C.byref(i3))
        return i1.value, i2.value, i3.value
    s = [0] * 30
    s[3], s[7], s[20] = 50, 40, 30
    assert tm(s) == (3, 7, 20)
    s[7], s[20] = 4, 6          # 6 >= 10 % of 50 stays second, 4 is dropped
    assert tm(s) == (3, 20, -1)
    s[20] = 4                   # second and third below 10 % of the first are both dropped
    assert tm(s) == (3, -1, -1)
    s[20] = 5
    s[7] = 5                    # ties: the earlier bin ranks higher
    assert tm(s) == (3, 7, 20)
    assert oracle.lib().orc_ratio_accept(50, 67, 50, 0.75) == 1
    assert oracle.lib().orc_ratio_accept(51, 200, 50, 0.75) == 0
    assert oracle.lib().orc_ratio_accept(30, 40, 50, 0.75) == 0   # 30 < 0.75*40 = 30 is false


def test_stereo_matches_known_answer(oracle):
    """Frame::ComputeStereoMatches on a hand-built case: one left keypoint, two right candidates (the
    farther descriptor loses), the right image = the left shifted by 8 px and the right keypoint 2 px
    off, so the window search must land on +2; the sub-pixel offset and the depth are recomputed here
    from the three window distances in float32.  (A single match at window distance 0 would be removed by
    the median rule, 0 < 1.5*1.4*0 being false, so the right image carries a little noise.)"""
    rng = np.random.default_rng(11)
    w, h = 200, 100
    wide = rng.integers(0, 256, size=(h, w + 8), dtype=np.uint8)
    left, right = wide[:, :w].copy(), wide[:, 8:].copy()      # right[x] = left[x + 8]
    right = np.clip(right.astype(np.int16) + rng.integers(-1, 2, size=right.shape), 0, 255).astype(np.uint8)
    KP = oracle.KEYPOINT_DTYPE
    kl = np.zeros(1, KP)
    kl["x"], kl["y"], kl["octave"] = 100.0, 50.0, 0
    kr = np.zeros(3, KP)
    kr["x"] = [90.0, 60.0, 90.0]
    kr["y"] = [50.4, 50.0, 70.0]          # the third one lies outside the row band
    kr["octave"] = 0
    dl = rng.integers(0, 256, size=(1, 32), dtype=np.uint8)
    dr = np.repeat(dl, 3, axis=0)
    dr[0, 0] ^= 0x0F                       # distance 4
    dr[1, :2] ^= 0xFF                      # distance 16
    sf = np.ones(1, np.float32)
    mb, mbf = np.float32(0.5), np.float32(40.0)   # maxD = 80
    n, ur, dp, sad = oracle.compute_stereo_matches(kl, dl, kr, dr, [left], [right], sf, sf, float(mb), float(mbf))
    assert n == 1

    def window_dist(inc):
        a = left[45:56, 95:106].astype(np.int32) - int(left[50, 100])
        b = right[45:56, 85 + inc:96 + inc].astype(np.int32) - int(right[50, 90 + inc])
        return np.float32(np.abs(a - b).sum())

    d1, d2, d3 = window_dist(1), window_dist(2), window_dist(3)
    assert sad[0] == int(d2) and 0 < d2 < min(d1, d3)
    delta = (d1 - d3) / (np.float32(2.0) * (d1 + d3 - np.float32(2.0) * d2))
    best_ur = np.float32(1.0) * (np.float32(90.0) + np.float32(2.0) + delta)
    assert ur[0] == best_ur
    assert dp[0] == mbf / (np.float32(100.0) - best_ur)
    # a disparity outside [0, mbf/mb) is rejected: with mbf/mb = 4 the match disappears
    n, ur, dp, _ = oracle.compute_stereo_matches(kl, dl, kr, dr, [left], [right], sf, sf, 10.0, 40.0)
    assert n == 0 and ur[0] == -1 and dp[0] == -1


def test_distinctive_descriptor_known_answer(oracle):
    """MapPoint::ComputeDistinctiveDescriptors by hand: rows A, B = A with 2 bits flipped, C = A with 2
    other bits flipped, D = all different.  Distances from A: 0,2,2,256 -> sorted 0,2,2,256, median index
    (int)(0.5*3) = 1 -> 2; from B: 0,2,4,254 -> 2; from C: 0,2,4,254 -> 2; from D: 0,254,254,256 -> 254.
    The first row with the least median (A) wins."""
    a = np.zeros(32, np.uint8)
    b, c = a.copy(), a.copy()
    b[0] = 0x03
    c[1] = 0x03
    d = np.full(32, 0xFF, np.uint8)
    assert oracle.distinctive_descriptor(np.stack([a, b, c, d])) == 0
    assert oracle.distinctive_descriptor(np.stack([d, a, b, c])) == 1
    # two rows: the median index is (int)(0.5*1) = 0, the distance of a row to itself: index 0 wins
    assert oracle.distinctive_descriptor(np.stack([d, a])) == 0
    assert oracle.distinctive_descriptor(np.zeros((0, 32), np.uint8)) == -1


def test_search_by_bow_known_answer(oracle):
    """ORBmatcher::SearchByBoW by hand.  Node 5 holds two frame features: f0 = X and f1 = X with bits
    100..109 flipped, and two key-frame features: k0 = X with bit 0 flipped (distances 1 and 11: takes f0),
    k1 = X with bits 1,2 flipped (f0 is taken, so only f1 at distance 12 remains: best 12, second 256).
    Node 6 exists only in the key frame, node 9 only in the frame: both are skipped."""
    def flip(row, bits):
        b = np.unpackbits(row.copy())
        b[list(bits)] ^= 1
        return np.packbits(b)

    x = np.random.default_rng(2).integers(0, 256, size=32, dtype=np.uint8)
    d2 = np.stack([x, flip(x, range(100, 110)), flip(x, range(0, 256, 2))])
    d1 = np.stack([flip(x, [0]), flip(x, [1, 2]), flip(x, [3])])
    nodes1 = (np.array([5, 6], np.int32), np.array([0, 2, 3], np.int32), np.array([0, 1, 2], np.int32))
    nodes2 = (np.array([5, 9], np.int32), np.array([0, 2, 3], np.int32), np.array([0, 1, 2], np.int32))
    ang = np.zeros(3, np.float32)
    n, m12, m21 = oracle.search_by_bow(d1, ang, None, nodes1, d2, ang, None, nodes2, False, 0.6, True)
    assert n == 2 and list(m12) == [0, 1, -1] and list(m21) == [0, 1, -1]
    # without a usable map point k0 is skipped and k1 takes f0 (distance 2 against 12: 2 < 0.6 * 12)
    n, m12, m21 = oracle.search_by_bow(d1, ang, np.array([0, 1, 1], np.uint8), nodes1, d2, ang, None, nodes2, False,
                                       0.6, True)
    assert n == 1 and list(m12) == [-1, 0, -1]
    # the ratio test: 0.1 * 11 is not above 1
    n, m12, _ = oracle.search_by_bow(d1, ang, None, nodes1, d2, ang, None, nodes2, False, 0.05, True)
    assert n == 0 and list(m12) == [-1, -1, -1]


def test_vocabulary_transform_known_answer(oracle):
    """A two-level tree by hand: root -> A (all zero), B (all ones); A -> A1 (zero), A2 (one bit set);
    B is a leaf.  A feature with 3 bits set goes to A (3 < 253), then to A2 (2 < 3); the all-ones feature
    goes to the leaf B; a feature at equal distance from A and B (128 bits set, bit 0 among them) takes the
    first child A and there A2 (127 < 128)."""
    z, o = np.zeros(32, np.uint8), np.full(32, 0xFF, np.uint8)
    a2 = z.copy()
    a2[0] = 0x01
    desc = np.stack([z, z, o, z, a2])                  # 0 root, 1 A, 2 B, 3 A1, 4 A2
    tree = dict(desc=desc, child_start=np.array([0, 2, 4, 4, 4, 4], np.int32), children=np.array([1, 2, 3, 4], np.int32),
                word_id=np.array([-1, -1, 0, 1, 2], np.int32), weight=np.array([0, 0, 1.5, 2.5, 3.5]), L=2)
    f1 = z.copy()
    f1[0] = 0x07
    half = z.copy()
    half[:16] = 0xFF
    word, node, wt = oracle.vocabulary_transform(tree, np.stack([f1, o, half]), 1)
    assert list(word) == [2, 0, 2] and list(wt) == [3.5, 1.5, 3.5]
    assert list(node) == [1, 2, 1]                     # the node at level L - 1
    _, node0, _ = oracle.vocabulary_transform(tree, np.stack([f1, o, half]), 0)
    assert list(node0) == [4, 0, 4]                    # level 2: the leaf itself; B stops at level 1 -> stays 0


def test_undistort_points_known_answers():
    """cv::undistortPoints as Frame::UndistortKeyPoints calls it: the principal point is a fixed point,
    zero k1 copies the input (src/Frame.cc:469-473), and re-applying the forward distortion model to
    the result reproduces the input pixel."""
    from oracle import pyoracle as orc
    cam = np.array([458.654, 457.296, 367.215, 248.375], np.float32)
    dist = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)
    pp = np.array([[cam[2], cam[3]]], np.float32)
    assert np.array_equal(orc.undistort_points(pp, cam, dist), pp)
    rng = np.random.default_rng(4)
    xy = np.stack([rng.uniform(100, 650, 500), rng.uniform(60, 420, 500)], 1).astype(np.float32)
    assert np.array_equal(orc.undistort_points(xy, cam, np.zeros(4, np.float32)), xy)
    for d in (dist, np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)):
        un = orc.undistort_points(xy, cam, d).astype(np.float64)
        k1, k2, p1, p2 = (float(v) for v in d[:4])
        k3 = float(d[4]) if len(d) > 4 else 0.0
        x = (un[:, 0] - float(cam[2])) / float(cam[0])
        y = (un[:, 1] - float(cam[3])) / float(cam[1])
        r2 = x * x + y * y
        rad = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
        xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        back = np.stack([xd * float(cam[0]) + float(cam[2]), yd * float(cam[1]) + float(cam[3])], 1)
        # five iterations leave a residual that grows towards the image border: well below a pixel here
        assert np.abs(back - xy).max() < 0.2 and np.median(np.abs(back - xy)) < 1e-3


def test_search_by_projection_known_answers():
    """Hand-made scene for ORBmatcher::SearchByProjection (src/ORBmatcher.cc:46-137, :1412-1566): first-come
    exclusion, the same-level ratio rule, the right-coordinate rule and the level window."""
    from oracle import pyoracle as orc
    import orbslam2_amd as orbx
    kps = np.zeros(4, orbx.KEYPOINT_DTYPE)
    kps["x"] = [100, 102, 300, 500]
    kps["y"] = [100, 101, 200, 300]
    kps["octave"] = [1, 1, 2, 0]
    desc = np.zeros((4, 32), np.uint8)
    desc[1, 0] = 0x0F        # 4 bits from feature 0
    desc[2, :4] = 0xFF       # 32 bits
    desc[3, :2] = 0xFF       # 16 bits
    bounds = (0.0, 640.0, 0.0, 480.0)
    zero = np.zeros((1, 32), np.uint8)

    def run(x, y, r, lo, hi, sd, **kw):
        return orc.search_by_projection(kps, desc, bounds, np.float32(x), np.float32(y), np.float32(r), np.int32(lo),
                                        np.int32(hi), sd, kw.pop("ratio", 0.8), **kw)

    # one source over features 0 and 1 (same level): best 0 bits, second 4 bits: 0 > 0.8*4 is false -> accepted
    nm, match, fm = run([101], [100], [5], [0], [1], zero)
    assert nm == 1 and list(match) == [0] and list(fm) == [0, -1, -1, -1]
    # descriptor 3 bits from feature 0 and 1 bit from feature 1: best = feature 1 (1), second = feature 0 (3),
    # same level and 1 > 0.3*3: rejected by the ratio rule; accepted when the levels differ
    sd = np.zeros((1, 32), np.uint8)
    sd[0, 0] = 0x07
    nm, match, _ = run([101], [100], [5], [0], [1], sd, ratio=0.3)
    assert nm == 0 and list(match) == [-1]
    kps["octave"][1] = 0
    nm, match, _ = run([101], [100], [5], [0], [1], sd, ratio=0.3)
    assert nm == 1 and list(match) == [1]
    kps["octave"][1] = 1
    # two sources aim at feature 0: the first wins it, the second falls back to feature 1 ...
    two = np.zeros((2, 32), np.uint8)
    nm, match, fm = run([101, 101], [100, 100], [5, 5], [0, 0], [1, 1], two, use_second=False)
    assert nm == 2 and list(match) == [0, 1] and list(fm[:2]) == [0, 1]
    # ... unless the first source's map point has no observations: then both write feature 0
    nm, match, fm = run([101, 101], [100, 100], [5, 5], [0, 0], [1, 1], two, use_second=False,
                        blocks=np.array([0, 1], np.uint8))
    assert nm == 2 and list(match) == [0, 0] and fm[0] == 1
    # features that already hold an observed map point are skipped
    nm, match, _ = run([101], [100], [5], [0], [1], zero, taken=np.array([1, 0, 0, 0], np.uint8), use_second=False)
    assert list(match) == [1]
    # level window: feature 2 (level 2) is outside [0, 1]; distance above TH_HIGH is refused
    nm, match, _ = run([300], [200], [5], [0], [1], zero)
    assert nm == 0
    nm, match, _ = run([300], [200], [5], [1], [2], zero, th_dist=31)
    assert nm == 0
    nm, match, _ = run([300], [200], [5], [1], [2], zero, th_dist=32)
    assert list(match) == [2]
    # right-coordinate rule: |xr - uright| > r drops the candidate, uright <= 0 (no stereo match) does not
    ur = np.array([-1, -1, -1, 480], np.float32)
    nm, match, _ = run([500], [300], [5], [0], [0], zero, uright=ur, xr=np.array([486], np.float32))
    assert nm == 0
    nm, match, _ = run([500], [300], [5], [0], [0], zero, uright=ur, xr=np.array([484], np.float32))
    assert list(match) == [3]


def test_search_by_projection_histogram_factor_of_the_key_frame_form():
    """The relocalisation form bins rotations with 1.0f/HISTO_LENGTH (src/ORBmatcher.cc:1581), the last-frame
    form with HISTO_LENGTH/360.0f (:1420): rotations of 5 and 10 degrees share bin 0 in the first and fall into
    bins 0 and 1 in the second, where the single 10-degree match is then outside the three maxima."""
    from oracle import pyoracle as orc
    import orbslam2_amd as orbx
    n = 10
    kps = np.zeros(n, orbx.KEYPOINT_DTYPE)
    kps["x"] = 50 + 40 * np.arange(n)
    kps["y"] = 100
    desc = orbx.synth_descriptors(5, n)
    rot = np.array([5, 5, 5, 5, 10, 100, 100, 100, 200, 200], np.float32)
    bounds = (0.0, 640.0, 0.0, 480.0)
    args = (kps, desc, bounds, kps["x"], kps["y"], np.full(n, 5, np.float32), np.zeros(n, np.int32),
            np.zeros(n, np.int32), desc, 0.9)
    nm1, m1, f1 = orc.search_by_projection(*args, src_angle=rot, use_second=False, check_orientation=1)
    assert nm1 == 9 and list(m1) == list(range(n)) and f1[4] == -2 and (np.delete(f1, 4) == np.delete(np.arange(n), 4)).all()
    nm2, m2, f2 = orc.search_by_projection(*args, src_angle=rot, use_second=False, check_orientation=2)
    assert nm2 == 10 and list(f2) == list(range(n))
    nm0, _, f0 = orc.search_by_projection(*args, src_angle=rot, use_second=False, check_orientation=0)
    assert nm0 == 10 and list(f0) == list(range(n))


def test_resize_edge_clamps_known_answers(oracle):
    """cv::resize INTER_LINEAR (8-bit, 11-bit coefficients) on two small cases worked out by hand.
    6x4 -> 5x3 (ratio 1.2 in x, 4/3 in y): fx = .1 .3 .5 .7 .9 at sx = 0..4, coefficient pairs
    (1843,205) (1434,614) (1024,1024) (614,1434) (205,1843); fy = 1/6, 1/2, 5/6 at sy = 0,1,2, pairs
    (1707,341) (1024,1024) (341,1707).  First pixel: H0 = 10*1843 + 60*205 = 30730, H1 = 0*1843 + 128*205 =
    26240; (1707*(30730>>4))>>16 = 50, (341*(26240>>4))>>16 = 8; (50 + 8 + 2) >> 2 = 15."""
    src = np.array([[10, 60, 200, 35, 90, 255], [0, 128, 77, 250, 13, 140], [99, 31, 180, 64, 222, 5],
                    [240, 17, 150, 201, 48, 111]], np.uint8)
    want = np.array([[15, 104, 125, 75, 220], [52, 94, 143, 129, 77], [197, 60, 167, 107, 92]], np.uint8)
    assert np.array_equal(oracle.resize(src, 5, 3), want)
    # 3x2 -> 5x4 (enlarging: every clamp of the coordinate computation is taken).  Columns: fx = -0.2 -> sx = 0,
    # weight (2048,0); 0.4 -> (1229,819); 1.0 -> sx = 1, (2048,0); 1.6 -> sx = 1, (819,1229); 2.2 -> sx = 2 = sw-1:
    # xmax = 4, the column takes S[2]*2048.  Rows: fy = -0.25 -> sy = -1, both rows clamp to row 0 with (512,1536);
    # 0.25 -> rows 0,1 (1536,512); 0.75 -> rows 0,1 (512,1536); 1.25 -> sy = 1, sy+1 clamps to row 1: (1536,512)
    # of the same row.  Second pixel of the first row: H = 10*1229 + 200*819 = 176090 for both rows;
    # (512*(176090>>4))>>16 = 85, (1536*11005)>>16 = 257; (85 + 257 + 2) >> 2 = 86.
    src = np.array([[10, 200, 90], [250, 40, 160]], np.uint8)
    want = np.array([[10, 86, 200, 134, 90], [70, 106, 160, 128, 108], [190, 146, 80, 117, 143],
                     [250, 166, 40, 112, 160]], np.uint8)
    assert np.array_equal(oracle.resize(src, 5, 4), want)


def _cells_literal(oracle, img, ini_th, min_th):
    """The loop of ORBextractor::ComputeKeyPointsOctTree (src/ORBextractor.cc:826-893) written out literally on
    top of cv::FAST for one window (pinned above by test_fast_detect_equals_score_definition): candidates in
    coordinates relative to (16,16), in the order the reference appends them."""
    h, w = img.shape
    minB, maxBX, maxBY = 16, w - 16, h - 16
    width, height = np.float32(maxBX - minB), np.float32(maxBY - minB)
    nCols, nRows = int(width / np.float32(30)), int(height / np.float32(30))
    wCell, hCell = int(np.ceil(width / nCols)), int(np.ceil(height / nRows))
    out = []
    for i in range(nRows):
        iniY = minB + i * hCell
        maxY = iniY + hCell + 6
        if iniY >= maxBY - 3:
            continue
        maxY = min(maxY, maxBY)
        for j in range(nCols):
            iniX = minB + j * wCell
            maxX = iniX + wCell + 6
            if iniX >= maxBX - 6:
                continue
            maxX = min(maxX, maxBX)
            xs, ys, rs = oracle.fast_detect(img[iniY:maxY, iniX:maxX], ini_th)
            if len(xs) == 0:
                xs, ys, rs = oracle.fast_detect(img[iniY:maxY, iniX:maxX], min_th)
            out += [(int(x) + j * wCell, int(y) + i * hCell, int(r)) for x, y, r in zip(xs, ys, rs)]
    return out, wCell, hCell, nCols, nRows


def test_cell_grid_literal_loop_and_plateau_across_a_cell_border(oracle):
    """Per-cell FAST: the candidate list equals the literal cell loop, including a width (813) at which the last
    cell column starts 6 pixels before the border and is skipped by the `iniX >= maxBorderX-6` rule while the
    same remainder would not skip a row (`iniY >= maxBorderY-3`), and an image that is mirror-symmetric about a
    cell border, where two equal responses face each other across the border and both survive (non-maximum
    suppression never looks into the neighbouring cell), while the same pair inside a cell suppresses itself."""
    rng = np.random.default_rng(11)
    for (w, h) in [(813, 100), (200, 813), (176, 144)]:
        img = np.full((h, w), 128, np.uint8)
        for _ in range(w * h // 250):
            x, y = rng.integers(0, w - 4), rng.integers(0, h - 4)
            img[y:y + rng.integers(3, 30), x:x + rng.integers(3, 30)] = rng.integers(0, 256)
        img = np.clip(img.astype(int) + rng.integers(-3, 4, img.shape), 0, 255).astype(np.uint8)
        want, wCell, hCell, nCols, nRows = _cells_literal(oracle, img, 20, 7)
        xs, ys, rs = oracle.level_candidates(img, 20, 7)
        assert list(zip(xs.tolist(), ys.tolist(), rs.tolist())) == want and len(want) > 50
        if w == 813:
            assert 16 + (nCols - 1) * wCell >= (w - 16) - 6  # the last column is the skipped one
    # Two equally bright pixels side by side on a flat background: each sees sixteen darker ring pixels, so both
    # are corners with the same response.  Astride the border between the first two cell columns (detect regions
    # [19, 19+wCell) and [19+wCell, ...)) both survive, each alone in its cell; ten columns further, inside one
    # cell, each has a neighbour that is not smaller and neither survives.
    w, h = 176, 112
    _, wCell, hCell, _, _ = _cells_literal(oracle, np.zeros((h, w), np.uint8), 20, 7)
    b = 19 + wCell
    img = np.full((h, w), 50, np.uint8)
    img[40, b - 1:b + 1] = 200
    img[70, b + 10:b + 12] = 200
    assert (40 - 19) // hCell == (40 - 19 + 1) // hCell and (b + 11 - 19) // wCell == (b + 10 - 19) // wCell
    want, *_ = _cells_literal(oracle, img, 20, 7)
    xs, ys, rs = oracle.level_candidates(img, 20, 7)
    got = list(zip(xs.tolist(), ys.tolist(), rs.tolist()))
    assert got == want
    assert got == [(b - 1 - 16, 40 - 16, 149), (b - 16, 40 - 16, 149)]  # response = A - 1 = 150 - 1


def test_brief_rounds_half_to_even(oracle):
    """computeOrbDescriptor (src/ORBextractor.cc:128-169) rounds the rotated sample offsets with cvRound, i.e. to
    the nearest integer with ties to even.  At 30 degrees the float sine is exactly 0.5, so pattern points with
    y = 0 and odd x land on a tie: point (1,0) of test 31 has row offset 0.5 -> 0 (not 1), point (-9,0) of test
    15 has row offset -4.5 -> -4 (not -5).  An independent restatement in float32 with np.rint must equal the
    oracle's descriptor, and a half-away-from-zero variant of it must differ, on an image built for that."""
    import re
    pat = [int(v) for v in re.findall(r"-?\d+", open("orbslam2_amd/csrc/brief_pattern.inc").read().split("\n", 1)[1])]
    f32 = np.float32
    rad = f32(f32(30.0) * f32(np.pi / f32(180.0)))
    a, b = f32(np.cos(np.float64(rad))), f32(np.sin(np.float64(rad)))
    assert b == f32(0.5)
    assert pat[4 * 31:4 * 31 + 2] == [1, 0] and pat[4 * 15:4 * 15 + 2] == [-9, 0]
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, size=(41, 41), dtype=np.uint8)
    # make the two tied samples decide their tests: the tie-to-even pixel bright, the half-away pixel dark
    img[20 + 0, 20 + 1], img[20 + 1, 20 + 1] = 0, 255
    img[20 - 4, 20 - 8], img[20 - 5, 20 - 8] = 0, 255

    def describe(rounding):
        d = np.zeros(32, np.uint8)
        for t in range(256):
            v = []
            for px, py in (pat[4 * t:4 * t + 2], pat[4 * t + 2:4 * t + 4]):
                row = f32(f32(px) * b) + f32(f32(py) * a)
                col = f32(f32(px) * a) - f32(f32(py) * b)
                v.append(int(img[20 + int(rounding(row)), 20 + int(rounding(col))]))
            d[t // 8] |= int(v[0] < v[1]) << (t % 8)
        return d

    even = describe(np.rint)
    away = describe(lambda x: np.sign(x) * np.floor(np.abs(x) + f32(0.5)))
    got = oracle.brief_descriptor(img, 20, 20, 30.0)
    assert np.array_equal(got, even)
    assert not np.array_equal(even, away)


def test_features_in_area_hand_case(oracle):
    """Frame::PosInGrid / GetFeaturesInArea (src/Frame.cc:349-420) against a case worked out on paper
    (tests/kat_cases.py): round-to-nearest binning, keypoints that fall off the grid's last half cell, the
    column-outer / row-inner visiting order, strict radius, level switches, the four early returns."""
    import kat_cases as K
    kps, cells, queries = K.features_in_area_case()
    q = np.array([a for a, _ in queries], np.float32)
    lists, cs, ci = oracle.grid_queries(kps, K.BOUNDS, q[:, 0], q[:, 1], q[:, 2], q[:, 3].astype(np.int32),
                                        q[:, 4].astype(np.int32))
    for (args, want), got in zip(queries, lists):
        assert list(got) == want, args
    # the grid itself: cell = column * 48 + row, items in index order
    for c in range(64):
        for r in range(48):
            items = list(ci[cs[c * 48 + r]:cs[c * 48 + r + 1]])
            assert items == cells.get((c, r), []), (c, r)
    assert cs[-1] == 8  # keypoints 6 and 8 are in no cell


def test_search_for_initialization_hand_case(oracle):
    """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:433-549) against a case worked out on paper
    (tests/kat_cases.py): a match taken over by a later keypoint, a candidate skipped by vMatchedDistance, a
    level-1 keypoint ignored, a second-best of INT_MAX passing the ratio, TH_LOW, a ratio failure, and a match
    cleared by the three-maxima histogram in which a taken-over entry still counts."""
    import kat_cases as K
    k1, d1, k2, d2, win, ratio, no_ori, with_ori = K.search_for_initialization_case()
    for ori, (want_n, want_m) in ((0, no_ori), (1, with_ori)):
        prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], axis=1), np.float32)
        before = prev.copy()
        n, m12, pairs = oracle.search_for_initialization(k1, d1, k2, d2, K.BOUNDS, prev, win, ratio, ori)
        assert n == want_n and list(m12) == want_m
        assert pairs == 2 + 2 + 2 + 1 + 2 + 6  # i0, i1, i2: j0 and j1; i4: j2; i5: j3 and j4; six isolated pairs
        for i, j in enumerate(want_m):
            want_xy = (k2["x"][j], k2["y"][j]) if j >= 0 else tuple(before[i])
            assert tuple(prev[i]) == want_xy, i
