"""GPU parity of Frame::ComputeStereoMatches through the C ABI against the oracle's restatement."""
import numpy as np
import pytest

import orbslam2_amd as orbx

pytestmark = pytest.mark.gpu

MB, MBF = 0.537, 386.1448  # KITTI 00-02: baseline and baseline * fx (Examples/Stereo/KITTI00-02.yaml)


def stereo_pair(seed, w, h, disparities, noise_seed):
    """Left image and a right image whose horizontal bands are the left one shifted by the given disparities."""
    wide = orbx.synth_image(seed, w + 160, h, 0)
    left = wide[:, 48:48 + w].copy()
    right = np.empty_like(left)
    nb = len(disparities)
    for b, d in enumerate(disparities):
        y0, y1 = b * h // nb, (b + 1) * h // nb
        right[y0:y1] = wide[y0:y1, 48 + d:48 + d + w]
    rng = np.random.default_rng(noise_seed)
    right = np.clip(right.astype(np.int16) + rng.integers(-2, 3, size=right.shape), 0, 255).astype(np.uint8)
    return left, right


def oracle_stereo(oracle, exl, exr, f, mb, mbf):
    kl, dl = exl.batch_download(f)
    kr, dr = exr.batch_download(f)
    nl = exl.nlevels
    pl = [exl.pyramid_level(f, l, 0) for l in range(nl)]
    pr = [exr.pyramid_level(f, l, 0) for l in range(nl)]
    return oracle.compute_stereo_matches(kl, dl, kr, dr, pl, pr, exl.GetScaleFactors(), exl.GetInverseScaleFactors(),
                                         mb, mbf)


@pytest.mark.parametrize("w,h,nfeat", [(1241, 376, 2000), (752, 480, 1000)])
def test_stereo_matches_vs_oracle(oracle, w, h, nfeat):
    lefts, rights = [], []
    for i, disp in enumerate([(12, 31), (0, 5, 44), (3,), (70, 9)]):
        a, b = stereo_pair(0x5EED + i, w, h, disp, i)
        lefts.append(a), rights.append(b)
    # a right image without texture: no candidates at all
    lefts.append(lefts[0]), rights.append(orbx.synth_image(7, w, h, 1))
    nb = len(lefts)
    exl = orbx.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=nb)
    exr = orbx.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=nb)
    exl.extract_batch_host(lefts)
    exr.extract_batch_host(rights)
    total = 0
    for f in range(nb):
        nm, ur, dp = orbx.ComputeStereoMatches(exl, exr, MB, MBF, frame=f)
        on, our, odp, _ = oracle_stereo(oracle, exl, exr, f, MB, MBF)
        assert nm == on
        assert np.array_equal(ur, our)    # same float operations in the same order: bit-exact
        assert np.array_equal(dp, odp)
        assert ((ur >= 0) == (dp > 0)).all()
        total += nm
    assert total > 1000
    assert orbx.ComputeStereoMatches(exl, exr, MB, MBF, frame=nb - 1)[0] == 0


def test_stereo_device_form_and_errors():
    import torch
    w, h = 640, 480
    a, b = stereo_pair(3, w, h, (20,), 1)
    exl = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
    exr = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
    with pytest.raises(orbx.OrbxError):
        orbx.ComputeStereoMatches(exl, exr, MB, MBF)  # no batch yet
    exl.extract_batch_host([a, a])
    exr.extract_batch_host([b, b])
    cap = exl.max_keypoints
    ur = torch.zeros((2, cap), dtype=torch.float32, device="cuda")
    dp = torch.zeros((2, cap), dtype=torch.float32, device="cuda")
    nm = torch.zeros(2, dtype=torch.int32, device="cuda")
    orbx.compute_stereo_matches_device(exl, exr, MB, MBF, ur.data_ptr(), dp.data_ptr(), nm.data_ptr())
    torch.cuda.synchronize()
    n0, u0, d0 = orbx.ComputeStereoMatches(exl, exr, MB, MBF, frame=0)
    assert int(nm[0]) == n0 == int(nm[1]) and n0 > 100
    assert np.array_equal(ur[0].cpu().numpy()[:len(u0)], u0) and np.array_equal(ur[1].cpu().numpy()[:len(u0)], u0)
    assert (ur[0].cpu().numpy()[len(u0):] == -1).all()
    ok = u0 >= 0
    kl, _ = exl.batch_download(0)
    disp = kl["x"][ok] - u0[ok]
    assert abs(np.median(disp) - 20) < 0.5     # the synthetic disparity comes back
    # mismatched handles
    exs = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    exs.extract_batch_host([b])
    with pytest.raises(orbx.OrbxError):
        orbx.ComputeStereoMatches(exl, exs, MB, MBF)
