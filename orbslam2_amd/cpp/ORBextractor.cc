// Drop-in for src/ORBextractor.cc of ORB-SLAM2: the cv:: surface on top of orbx::Extractor.
// Compiled only where OpenCV is installed (it is the one file of this repository that needs it).
#include "ORBextractor.h"

#include <algorithm>
#include <cassert>
#include <climits>
#include <cstring>
#include <stdexcept>

#include "orbx.hpp"

static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "orbx_keypoint mirrors cv::KeyPoint");

namespace ORB_SLAM2 {

static const int EDGE_THRESHOLD = 19;  // src/ORBextractor.cc:74

void ImagePyramid::fetch() const
{
    if (stale_ && owner_)
        owner_->MaterializePyramid();
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST),
      minThFAST(_minThFAST)
{
    // the same arithmetic as the library's tables (checked against each other by the adaptor's test);
    // follows src/ORBextractor.cc:437-456.  No device is touched until the first image arrives.
    mvScaleFactor.resize(nlevels), mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels), mvInvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);
        mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    for (int i = 0; i < nlevels; i++) {
        mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
        mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    mvImagePyramid.owner_ = this;
    mvImagePyramid.levels_.resize(nlevels);  // src/ORBextractor.cc:458
}

ORBextractor::~ORBextractor() {}

::orbx_extractor *ORBextractor::handle() { return impl_ ? impl_->handle() : nullptr; }

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &_keypoints,
                              cv::OutputArray _descriptors)
{
    if (_image.empty())
        return;  // outputs untouched (src/ORBextractor.cc:1117-1118)
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);

    if (!impl_ || image.cols > impl_w_ || image.rows > impl_h_) {
        impl_w_ = std::max(impl_w_, image.cols), impl_h_ = std::max(impl_h_, image.rows);
        impl_.reset(new orbx::Extractor(nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST, impl_w_, impl_h_));
    }
    const orbx_keypoint *kps = nullptr;
    const uint8_t *desc = nullptr;
    const int n = impl_->extract_raw(image.data, image.cols, image.rows, (int)image.step, kps, desc);
    mvImagePyramid.stale_ = true;  // the levels on the host (if any) are those of an earlier image

    _keypoints.resize(n);
    if (n == 0)
        _descriptors.release();  // src/ORBextractor.cc:1135-1136
    else {
        std::memcpy(static_cast<void *>(_keypoints.data()), kps, sizeof(orbx_keypoint) * n);
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        if (d.isContinuous())
            std::memcpy(d.data, desc, (size_t)n * 32);
        else
            for (int i = 0; i < n; i++)
                std::memcpy(d.ptr(i), desc + (size_t)i * 32, 32);
    }
}

bool ORBextractor::MaterializePyramid()
{
    if (!impl_)
        return false;
    // one buffer for all padded levels (src/ORBextractor.cc:1199-1219 allocates one per level), filled
    // by one download; every level shares its reference count
    std::vector<int64_t> off(nlevels);
    std::vector<int> w(nlevels), h(nlevels);
    int64_t total = 0;
    orbx::check(orbx_pyramid_layout(impl_->handle(), EDGE_THRESHOLD, off.data(), w.data(), h.data(), &total),
                "orbx_pyramid_layout");
    // a cv::Mat row holds at most INT_MAX columns (a padded pyramid of more than 2 GiB: frames beyond 38000 x 38000)
    if (total <= 0 || total > (int64_t)INT_MAX)
        throw std::length_error("ORBextractor: the padded pyramid does not fit one cv::Mat row");
    cv::Mat all(1, (int)total, CV_8UC1);
    orbx::check(orbx_pyramid_download(impl_->handle(), 0, EDGE_THRESHOLD, all.data, total), "orbx_pyramid_download");
    for (int l = 0; l < nlevels; l++) {
        const int pw = w[l] + 2 * EDGE_THRESHOLD, ph = h[l] + 2 * EDGE_THRESHOLD;
        cv::Mat whole = all.colRange((int)off[l], (int)off[l] + pw * ph).reshape(1, ph);
        mvImagePyramid.levels_[l] = whole(cv::Rect(EDGE_THRESHOLD, EDGE_THRESHOLD, w[l], h[l]));
    }
    mvImagePyramid.stale_ = false;
    return true;
}

}  // namespace ORB_SLAM2
