// Drop-in for src/ORBextractor.cc of ORB-SLAM2: the cv:: surface on top of orbx::Extractor.
// Compiled only where OpenCV is installed (it is the one file of this repository that needs it).
#include "ORBextractor.h"

#include <cassert>
#include <cstring>

#include "orbx.hpp"

static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "orbx_keypoint mirrors cv::KeyPoint");

namespace ORB_SLAM2 {

static const int EDGE_THRESHOLD = 19;  // src/ORBextractor.cc:74

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST),
      minThFAST(_minThFAST)
{
    // the tables come from the library so that they are the ones the kernels use; they follow
    // src/ORBextractor.cc:437-456.  No device is touched until the first image arrives.
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
    mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor() {}

orbx_extractor *ORBextractor::handle() { return impl_ ? impl_->handle() : nullptr; }

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
    std::vector<orbx_keypoint> kps;
    std::vector<uint8_t> desc;
    (*impl_)(image.data, image.cols, image.rows, (int)image.step, kps, desc);

    const int n = (int)kps.size();
    _keypoints.resize(n);
    if (n == 0)
        _descriptors.release();  // src/ORBextractor.cc:1135-1136
    else {
        std::memcpy(static_cast<void *>(_keypoints.data()), kps.data(), sizeof(orbx_keypoint) * n);
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        for (int i = 0; i < n; i++)
            std::memcpy(d.ptr(i), desc.data() + (size_t)i * 32, 32);
    }
    // padded pyramid for the stereo matcher (src/ORBextractor.cc:1199-1219)
    for (int l = 0; l < nlevels; l++) {
        int pw = 0, ph = 0;
        std::vector<uint8_t> buf = impl_->pyramid_level(l, EDGE_THRESHOLD, pw, ph);
        cv::Mat whole(ph, pw, CV_8UC1);
        std::memcpy(whole.data, buf.data(), buf.size());
        mvImagePyramid[l] = whole(cv::Rect(EDGE_THRESHOLD, EDGE_THRESHOLD, pw - 2 * EDGE_THRESHOLD,
                                           ph - 2 * EDGE_THRESHOLD));
    }
}

}  // namespace ORB_SLAM2
