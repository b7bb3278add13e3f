"""orbx_match_top2_device_noalloc (the entry point that allocates nothing: partial results in a device array
inside the library) against orbx_match_top2_device, through the C ABI: indices and distances bit-exact, on
calls in a row that share the array -- the second smaller than the first."""
import numpy as np
import pytest

import orbslam2_amd as orbx
from orbslam2_amd import _lib as L

pytestmark = pytest.mark.gpu


def _data(ndb, nq, seed):
    db = orbx.synth_descriptors(seed, ndb)
    rng = np.random.default_rng(seed)
    q = db[rng.integers(0, ndb, size=nq)].copy()
    flips = rng.integers(0, 256, size=(nq, 24))  # up to 24 flipped bits per query
    for i in range(nq):
        bits = np.unpackbits(q[i])
        bits[flips[i, :rng.integers(0, 25)]] ^= 1
        q[i] = np.packbits(bits)
    db[1::97] = db[0::97][:len(db[1::97])]  # duplicates: ties between neighbouring rows
    return db, q


def _both(m, db, q, idx_base=0):
    import torch
    nq, ndb = len(q), len(db)
    d_db, d_q = torch.from_numpy(db).cuda(), torch.from_numpy(q).cuda()
    outs = []
    for noalloc in (False, True):
        best = torch.full((nq,), -7, dtype=torch.int32, device="cuda")
        d1 = torch.full((nq,), 999, dtype=torch.int16, device="cuda")
        d2 = torch.full((nq,), 999, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        stream = torch.cuda.current_stream().cuda_stream
        if noalloc:
            orbx.ORBmatcher.match_top2_device_noalloc(0, stream, d_q.data_ptr(), nq, d_db.data_ptr(), ndb, idx_base,
                                                      best.data_ptr(), d1.data_ptr(), d2.data_ptr())
        else:
            m.set_stream(stream)
            m.match_top2_device(d_q.data_ptr(), nq, d_db.data_ptr(), ndb, idx_base, best.data_ptr(), d1.data_ptr(),
                                d2.data_ptr())
        torch.cuda.synchronize()
        outs.append((best.cpu().numpy(), d1.cpu().numpy().view(np.uint16), d2.cpu().numpy().view(np.uint16)))
    return outs


def test_noalloc_top2_equals_handle_form_on_consecutive_calls(monkeypatch, oracle):
    m = orbx.ORBmatcher(0.75)
    # once against the oracle itself (the handle form is checked against it elsewhere): index base added by hand
    db, q = _data(5000, 100, 11)
    _, (b1, a1, s1) = _both(m, db, q, 40)
    ob, o1, o2 = oracle.match_top2(q, db)
    assert np.array_equal(b1, ob + 40) and np.array_equal(a1, o1) and np.array_equal(s1, o2)
    # two calls in a row on the same array, the second smaller; then the vector-ALU kernel (below 32
    # queries), an offset index base, and a larger call again
    for ndb, nq, base, seed in [(200000, 2000, 0, 3), (5000, 100, 0, 4), (3001, 20, 1000, 5), (60000, 999, 7, 6)]:
        db, q = _data(ndb, nq, seed)
        (b0, a0, s0), (b1, a1, s1) = _both(m, db, q, base)
        assert np.array_equal(b0, b1) and np.array_equal(a0, a1) and np.array_equal(s0, s1), (ndb, nq)
        assert (b0 >= base).all() and (a0 < 60).sum() > nq // 2
    # a cut into so many chunks that the array would overflow is coarsened, with the same result
    monkeypatch.setenv("ORBX_TOP2_BLOCKS", "8192")
    db, q = _data(400000, 2000, 8)
    (b0, a0, s0), (b1, a1, s1) = _both(m, db, q)
    assert np.array_equal(b0, b1) and np.array_equal(a0, a1) and np.array_equal(s0, s1)
    monkeypatch.delenv("ORBX_TOP2_BLOCKS")
    # an empty database: no best row
    import torch
    d_q = torch.zeros((40, 32), dtype=torch.uint8, device="cuda")
    best = torch.zeros(40, dtype=torch.int32, device="cuda")
    d1, d2 = torch.zeros(40, dtype=torch.int16, device="cuda"), torch.zeros(40, dtype=torch.int16, device="cuda")
    orbx.ORBmatcher.match_top2_device_noalloc(0, torch.cuda.current_stream().cuda_stream, d_q.data_ptr(), 40, 0, 0, 0,
                                              best.data_ptr(), d1.data_ptr(), d2.data_ptr())
    torch.cuda.synchronize()
    assert (best.cpu().numpy() == -1).all() and (d1.cpu().numpy() == 256).all() and (d2.cpu().numpy() == 256).all()


def test_noalloc_top2_rejects_what_does_not_fit():
    import torch
    lib = L.lib()
    nq, ndb = 600000, 40000  # two chunks of the longest kind x 600000 queries > 1,048,576 partial results
    d_q = torch.zeros((nq, 32), dtype=torch.uint8, device="cuda")
    d_db = torch.zeros((ndb, 32), dtype=torch.uint8, device="cuda")
    best = torch.full((nq,), -7, dtype=torch.int32, device="cuda")
    d1, d2 = torch.zeros(nq, dtype=torch.int16, device="cuda"), torch.zeros(nq, dtype=torch.int16, device="cuda")
    args = (0, None, d_q.data_ptr(), nq, d_db.data_ptr(), ndb, 0, best.data_ptr(), d1.data_ptr(), d2.data_ptr())
    assert lib.orbx_match_top2_device_noalloc(*args) == L.E_CAPACITY
    torch.cuda.synchronize()
    assert (best.cpu().numpy() == -7).all()  # nothing was launched
    assert lib.orbx_match_top2_device_noalloc(0, None, None, 4, d_db.data_ptr(), ndb, 0, best.data_ptr(),
                                              d1.data_ptr(), d2.data_ptr()) == L.E_ARG
    assert lib.orbx_match_top2_device_noalloc(0, None, d_q.data_ptr(), 2000000, d_db.data_ptr(), ndb, 0,
                                              best.data_ptr(), d1.data_ptr(), d2.data_ptr()) == L.E_CAPACITY
