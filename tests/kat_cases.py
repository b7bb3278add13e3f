"""Hand-worked cases for the two order-dependent pieces of ORB-SLAM2 logic on the matching path: every expected
value below is derived on paper from the reference's source lines (cited), not produced by running any code.
Used by tests/test_oracle_kat.py (pins the oracle) and tests/test_gpu_matcher.py (the same inputs through the
HIP path)."""
import numpy as np

KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"),
               ("class_id", "<i4")])
BOUNDS = (0.0, 640.0, 0.0, 480.0)  # mnMinX, mnMaxX, mnMinY, mnMaxY: 64 x 48 cells of 10 x 10 pixels


def keypoints(rows):
    """rows: (x, y, octave, angle)"""
    k = np.zeros(len(rows), KP)
    for i, (x, y, o, a) in enumerate(rows):
        k[i] = (x, y, 31.0, a, 50.0, o, -1)
    return k


def bits(*ranges):
    """32-byte descriptor with the given bit ranges [lo, hi) set"""
    b = np.zeros(256, np.uint8)
    for lo, hi in ranges:
        b[lo:hi] = 1
    return np.packbits(b, bitorder="little")


def features_in_area_case():
    """Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea, src/Frame.cc:239-253, 349-420.

    PosInGrid rounds (x - mnMinX) * 0.1 to the nearest cell (:412-413), so a keypoint at x = 25.0 (2.5 -> 3)
    sits in column 3 and one at x = 635.0 (63.5 -> 64 >= FRAME_GRID_COLS) is in no cell at all (:416-417).
    The query walks columns floor((x - r) * 0.1) .. ceil((x + r) * 0.1) outer, rows inner (:355-379), a cell's
    keypoints in insertion (= index) order, and keeps |dx| < r and |dy| < r, both strict (:400)."""
    kps = keypoints([
        (25.0, 35.0, 0, 0),   # 0: cell (3, 4)   [2.5 -> 3, 3.5 -> 4: halves round away from zero]
        (22.0, 12.0, 0, 0),   # 1: cell (2, 1)
        (31.0, 8.0, 1, 0),    # 2: cell (3, 1)   level 1
        (24.0, 31.0, 0, 0),   # 3: cell (2, 3)
        (23.0, 11.0, 2, 0),   # 4: cell (2, 1)   level 2, behind keypoint 1 in the same cell
        (45.0, 20.0, 0, 0),   # 5: cell (5, 2)   [4.5 -> 5]; exactly r = 20 right of the first query: excluded
        (635.0, 100.0, 0, 0),  # 6: column 64: in no cell
        (634.9, 100.0, 0, 0),  # 7: cell (63, 10)
        (5.0, 475.0, 0, 0),   # 8: row round(47.5) = 48: in no cell
        (4.9, 474.9, 0, 0),   # 9: cell (0, 47)
    ])
    cells = {(3, 4): [0], (2, 1): [1, 4], (3, 1): [2], (2, 3): [3], (5, 2): [5], (63, 10): [7], (0, 47): [9]}
    q = [  # (x, y, r, minLevel, maxLevel) -> expected indices in order
        # columns floor(0.5) = 0 .. ceil(4.5) = 5, rows 0 .. ceil(4.0) = 4: column 2: rows 1 (1, 4), 3 (3);
        # column 3: rows 1 (2), 4 (0); column 5 row 2: keypoint 5 has |dx| = 20, not < 20
        ((25.0, 20.0, 20.0, 0, -1), [1, 4, 3, 2, 0]),
        ((25.0, 20.0, 20.0, 0, 0), [1, 3, 0]),      # maxLevel >= 0 switches the level check on (:374): level 0 only
        ((25.0, 20.0, 20.0, 1, -1), [4, 2]),        # minLevel > 0: levels 1 and up
        ((25.0, 20.0, 20.0, 1, 1), [2]),
        ((25.0, 20.0, 20.001, 0, -1), [1, 4, 3, 2, 0, 5]),  # a hair more radius takes keypoint 5 in (column 5 last)
        ((630.0, 100.0, 6.0, 0, -1), [7]),          # keypoint 6 is 5 px away but was never binned
        ((700.0, 100.0, 5.0, 0, -1), []),           # floor(69.5) = 69 >= 64: early return (:356-357)
        ((-30.0, 100.0, 5.0, 0, -1), []),           # ceil(-2.5) = -2 < 0 (:361-362)
        ((100.0, 600.0, 5.0, 0, -1), []),           # floor(59.5) = 59 >= 48 (:366-367)
        ((100.0, -40.0, 5.0, 0, -1), []),           # ceil(-3.5) = -3 < 0 (:371-372)
        ((3.0, 477.0, 4.0, 0, -1), [9]),            # keypoint 8 (2 px away) is in no cell
        ((-3.0, 470.0, 8.0, 0, -1), [9]),           # window partly left of the grid: column range clamps to 0 (:355)
    ]
    return kps, cells, q


def search_for_initialization_case():
    """ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:433-549, window 30, mfNNratio 0.9.

    F2 (all level 0, angle 10): j0 (100,100) = no bits, j1 (110,100) = bits 0..49, j2 (300,300) = bits 100..199,
    j3 (500,100) = bits 0..99, j4 (500,110) = j3 with bits 0..2 cleared, j5..j10 isolated at (50+80k, 400).
    F1, search centres = own positions:
      i0 (100,100): d(j0) = 10, d(j1) = 40 -> 10 <= 50 and 10 < 0.9*40: takes j0, vMatchedDistance[j0] = 10
      i1 (105,100): d(j0) = 5 (< 10, so not skipped), d(j1) = 45 -> takes j0 over from i0 (:492-499): i0 unmatched,
                    the match count stays, vMatchedDistance[j0] = 5
      i2 (102,100): d(j0) = 7 is skipped because vMatchedDistance[j0] = 5 <= 7 (:471-472), before the best/second
                    bookkeeping; d(j1) = 43 is then the only candidate: bestDist2 stays INT_MAX and
                    43 < (float)INT_MAX*0.9 passes (:488): takes j1
      i3 (300,300): level 1, ignored although identical to j2 (:450-451)
      i4 (300,300): d(j2) = 51 > TH_LOW (:486)
      i5 (500,105): candidates in visiting order j3 (row 10), j4 (row 11): 30 and 33; 30 < 0.9*33 = 29.7 fails (:488)
      i6..i11: identical to j5..j10: distance 0, no other candidate
    Without the orientation check: 8 matches.  With it (:502-541), rot = angle1 - 10, bin = round(rot/12):
      bin 0: i6, i7, i8 (3)   bin 2: i9, i10 (2)   bin 10: i0, i2 (2: i0's entry stays in the histogram after its
      match was taken over, :511 has no counterpart in :492-499)   bin 5: i1 (1)   bin 15: i11 (1)
    Three maxima (:1698-1739): bins 0, 2, 10 (equal counts: the earlier bin ranks higher, and 2 >= 0.1*3 keeps
    both).  Bins 5 and 15 are cleared: i1 and i11 lose their matches -> 6 matches.  Had the taken-over entry not
    counted, bin 10 would tie with bin 5 at one entry and i1 would survive instead of i2.
    vbPrevMatched of matched keypoints becomes the partner's position (:544-546), the rest is untouched."""
    f2 = [((100, 100), bits()), ((110, 100), bits((0, 50))), ((300, 300), bits((100, 200))),
          ((500, 100), bits((0, 100))), ((500, 110), bits((3, 100)))]
    iso = [bits((8 * k, 8 * k + 120)) for k in range(6)]
    for k in range(6):
        f2.append(((50 + 80 * k, 400), iso[k]))
    k2 = keypoints([(x, y, 0, 10.0) for (x, y), _ in f2])
    d2 = np.stack([d for _, d in f2])
    i4 = bits((151, 200))  # j2 with its bits 100..150 cleared: distance 51
    f1 = [((100, 100), 0, 130.0, bits((0, 10))),
          ((105, 100), 0, 70.0, bits((0, 5))),
          ((102, 100), 0, 130.0, bits((0, 7))),
          ((300, 300), 1, 10.0, bits((100, 200))),
          ((300, 300), 0, 10.0, i4),
          ((500, 105), 0, 10.0, bits((0, 10), (40, 100)))]
    rots = [0.0, 0.0, 0.0, 24.0, 24.0, 180.0]
    for k in range(6):
        f1.append(((50 + 80 * k, 400), 0, 10.0 + rots[k], iso[k]))
    k1 = keypoints([(x, y, o, a) for (x, y), o, a, _ in f1])
    d1 = np.stack([d for _, _, _, d in f1])
    no_ori = [-1, 0, 1, -1, -1, -1, 5, 6, 7, 8, 9, 10]
    with_ori = [-1, -1, 1, -1, -1, -1, 5, 6, 7, 8, 9, -1]
    return k1, d1, k2, d2, 30, 0.9, (8, no_ori), (6, with_ori)
