"""Cross-checks of the oracle's OpenCV restatements against independent floating-point definitions
computed with torch / numpy (not bit-level pins: the fixed-point results must stay within the rounding
distance of the real-valued definition).

  resize   cv::resize INTER_LINEAR samples at (dx + 0.5) * scale - 0.5 with clamped neighbours: the same
           positions as torch's bilinear interpolation with align_corners=False and no antialiasing.
  blur     GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101): a separable convolution with the normalised
           samples of exp(-x^2 / 8) on a reflect-padded image.
  atan2    cv::fastAtan2 documents an accuracy of about 0.3 degrees.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

import orbslam2_amd as orbx


def test_resize_against_float_bilinear(oracle):
    img = orbx.synth_image(5, 640, 480, 0)
    for dw, dh in [(533, 400), (320, 240), (246, 185)]:
        got = oracle.resize(img, dw, dh).astype(np.float64)
        ref = F.interpolate(torch.from_numpy(img.astype(np.float64))[None, None], size=(dh, dw), mode="bilinear",
                            align_corners=False, antialias=False)[0, 0].numpy()
        err = np.abs(got - ref)
        # 11-bit coefficients and two roundings: well below one gray level on average, never above one
        assert err.max() <= 1.0 and err.mean() < 0.3, (dw, dh, err.max(), err.mean())


def test_blur_against_float_gaussian(oracle):
    img = orbx.synth_image(6, 320, 200, 0)
    x = np.arange(-3, 4, dtype=np.float64)
    k = np.exp(-x * x / (2.0 * 2.0 * 2.0))
    k /= k.sum()
    t = torch.from_numpy(img.astype(np.float64))[None, None]
    t = F.pad(t, (3, 3, 3, 3), mode="reflect")              # reflect without repeating the border pixel (101)
    kk = torch.from_numpy(k)
    t = F.conv2d(t, kk.view(1, 1, 1, 7))
    ref = F.conv2d(t, kk.view(1, 1, 7, 1))[0, 0].numpy()
    for variant in (0, 1):
        got = oracle.gauss7(img, variant).astype(np.float64)
        # the kernel of variant 1 sums to 257/256 (and saturates)
        err = np.abs(got - (np.minimum(ref * 257.0 / 256.0, 255.0) if variant else ref))
        # 8-bit kernel (18,34,48|49,56|55,...)/256 against the real-valued one: a gray level or two
        assert err.max() <= 2.0 and err.mean() < 0.6, (variant, err.max(), err.mean())


def test_fast_atan2_against_libm(oracle):
    rng = np.random.default_rng(3)
    worst = 0.0
    for _ in range(4000):
        y, x = (float(v) for v in rng.integers(-30000, 30001, size=2))
        if x == 0 and y == 0:
            continue
        ref = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(oracle.fast_atan2(y, x) - ref)
        worst = max(worst, min(d, 360.0 - d))
    assert worst < 0.3, worst


def test_ic_angle_against_float_centroid(oracle):
    img = orbx.synth_image(8, 200, 200, 0)
    um = oracle.umax()
    for (x, y) in [(50, 60), (100, 100), (150, 40), (33, 170)]:
        m10 = m01 = 0.0
        for v in range(-15, 16):
            for u in range(-um[abs(v)], um[abs(v)] + 1):
                m10 += u * float(img[y + v, x + u])
                m01 += v * float(img[y + v, x + u])
        ref = math.degrees(math.atan2(m01, m10)) % 360.0
        d = abs(oracle.ic_angle(img, x, y)[0] - ref)
        assert min(d, 360.0 - d) < 0.3
