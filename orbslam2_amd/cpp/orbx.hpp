// orbx.hpp -- OpenCV-free C++ host layer over the C ABI of liborbx (include/orbx.h).
//
// orbx::Extractor has the constructor arguments and the call shape of ORB_SLAM2::ORBextractor
// (include/ORBextractor.h:45-111 of the reference) with plain pointers in place of cv::Mat;
// orbx::descriptor_distance is ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1744-1760) on two
// 32-byte rows.  ORBextractor.h / ORBextractor.cc next to this file put the cv:: types back on top
// of it for builds that have OpenCV.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbx.h"

namespace orbx {

struct Error : std::runtime_error {
    int code;
    Error(int c, const char *where) : std::runtime_error(std::string(where) + ": " + orbx_strerror(c)), code(c) {}
};
inline void check(int rc, const char *where)
{
    if (rc < 0)
        throw Error(rc, where);
}

class Extractor {
public:
    Extractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int max_width = 1920,
              int max_height = 1088, int max_batch = 1, int device = 0, int blur_variant = 0)
    {
        orbx_params p{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, blur_variant,
                      max_width, max_height,  max_batch, device};
        check(orbx_extractor_create(&p, &h_), "orbx_extractor_create");
        nlevels_ = nlevels;
        scale_.resize(nlevels), inv_scale_.resize(nlevels), sigma2_.resize(nlevels), inv_sigma2_.resize(nlevels);
        check(orbx_extractor_scale_tables(h_, scale_.data(), inv_scale_.data(), sigma2_.data(), inv_sigma2_.data()),
              "orbx_extractor_scale_tables");
        cap_ = orbx_extractor_max_keypoints(h_);
    }
    ~Extractor() { orbx_extractor_destroy(h_); }
    Extractor(const Extractor &) = delete;
    Extractor &operator=(const Extractor &) = delete;

    // operator(): 8-bit gray image of height rows, stride bytes apart.  Returns false for an empty
    // image and leaves the outputs untouched, as the reference does (src/ORBextractor.cc:1117-1118);
    // zero keypoints clear both outputs.
    bool operator()(const uint8_t *image, int width, int height, int stride, std::vector<orbx_keypoint> &keypoints,
                    std::vector<uint8_t> &descriptors)
    {
        if (!image || width <= 0 || height <= 0)
            return false;
        kp_buf_.resize(cap_);
        desc_buf_.resize((size_t)cap_ * 32);
        int n = 0;
        check(orbx_extract(h_, image, width, height, stride, kp_buf_.data(), desc_buf_.data(), cap_, &n),
              "orbx_extract");
        keypoints.assign(kp_buf_.begin(), kp_buf_.begin() + n);
        descriptors.assign(desc_buf_.begin(), desc_buf_.begin() + (size_t)n * 32);
        return true;
    }

    // level of the pyramid of the last call with a reflect-101 frame of `border` pixels
    // (mvImagePyramid keeps border 19, src/ORBextractor.cc:1199-1219)
    std::vector<uint8_t> pyramid_level(int level, int border, int &padded_w, int &padded_h)
    {
        int w = 0, h = 0;
        check(orbx_pyramid_level(h_, 0, level, 0, nullptr, 0, &w, &h), "orbx_pyramid_level");
        padded_w = w + 2 * border, padded_h = h + 2 * border;
        std::vector<uint8_t> out((size_t)padded_w * padded_h);
        check(orbx_pyramid_level(h_, 0, level, border, out.data(), padded_w, nullptr, nullptr), "orbx_pyramid_level");
        return out;
    }

    int GetLevels() const { return nlevels_; }
    float GetScaleFactor() const { return nlevels_ > 1 ? scale_[1] : 1.f; }
    std::vector<float> GetScaleFactors() const { return scale_; }
    std::vector<float> GetInverseScaleFactors() const { return inv_scale_; }
    std::vector<float> GetScaleSigmaSquares() const { return sigma2_; }
    std::vector<float> GetInverseScaleSigmaSquares() const { return inv_sigma2_; }
    orbx_extractor *handle() { return h_; }
    int max_keypoints() const { return cap_; }

private:
    orbx_extractor *h_ = nullptr;
    int nlevels_ = 0, cap_ = 0;
    std::vector<float> scale_, inv_scale_, sigma2_, inv_sigma2_;
    std::vector<orbx_keypoint> kp_buf_;
    std::vector<uint8_t> desc_buf_;
};

inline int descriptor_distance(const uint8_t *a, const uint8_t *b) { return orbx_descriptor_distance(a, b); }

// Frame::ComputeStereoMatches (src/Frame.cc:534-708) on the images the two extractors processed last:
// fills mvuRight / mvDepth (one entry per left keypoint, -1 where unmatched), returns the match count.
inline int compute_stereo_matches(Extractor &left, Extractor &right, float mb, float mbf, std::vector<float> &uright,
                                  std::vector<float> &depth)
{
    const int cap = left.max_keypoints();
    uright.assign(cap, -1.0f);
    depth.assign(cap, -1.0f);
    int n = 0;
    const int nm = orbx_compute_stereo_matches(left.handle(), right.handle(), 0, mb, mbf, uright.data(), depth.data(),
                                               cap, &n);
    check(nm, "orbx_compute_stereo_matches");
    uright.resize(n);
    depth.resize(n);
    return nm;
}

}  // namespace orbx
