// Drop-in for include/ORBextractor.h of ORB-SLAM2 (lianbin/orbslam2): the same class name, constructor,
// operator(), getters and public mvImagePyramid (include/ORBextractor.h:45-111 of the reference),
// computed by liborbx on an MI355X.  Needs OpenCV headers, like the file it replaces; everything
// OpenCV-free lives in orbx.hpp.  ExtractorNode (reference :32-43) is gone: no other file uses it.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <cstddef>
#include <memory>
#include <vector>

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>

#include "orbx.h"  // ::orbx_extractor, the C-ABI handle type

namespace orbx {
class Extractor;
}

namespace ORB_SLAM2 {

class ORBextractor;

// What the reference declares as std::vector<cv::Mat> mvImagePyramid (include/ORBextractor.h:85).  The
// levels live on the GPU; the only reader outside the extractor is Frame::ComputeStereoMatches
// (src/Frame.cc:541,631,643,648), so they are fetched on the first access after a call to operator()
// -- all levels in one device-to-host copy -- and not at all by callers that never look.  Indexing,
// size() and iteration behave like the vector's; each level is a ROI inside a buffer padded by 19
// reflected pixels (src/ORBextractor.cc:1199-1219).
class ImagePyramid {
public:
    typedef std::vector<cv::Mat>::iterator iterator;
    typedef std::vector<cv::Mat>::const_iterator const_iterator;

    std::size_t size() const { return levels_.size(); }
    bool empty() const { return levels_.empty(); }
    cv::Mat &operator[](std::size_t level) { fetch(); return levels_[level]; }
    const cv::Mat &operator[](std::size_t level) const { fetch(); return levels_[level]; }
    cv::Mat &at(std::size_t level) { fetch(); return levels_.at(level); }
    const cv::Mat &at(std::size_t level) const { fetch(); return levels_.at(level); }
    iterator begin() { fetch(); return levels_.begin(); }
    iterator end() { fetch(); return levels_.end(); }
    const_iterator begin() const { fetch(); return levels_.begin(); }
    const_iterator end() const { fetch(); return levels_.end(); }
    // for code that copies the whole pyramid: std::vector<cv::Mat> p = extractor.mvImagePyramid;
    operator const std::vector<cv::Mat> &() const { fetch(); return levels_; }

private:
    friend class ORBextractor;
    void fetch() const;
    ORBextractor *owner_ = nullptr;
    mutable bool stale_ = false;  // operator() ran since the levels were fetched
    mutable std::vector<cv::Mat> levels_;
};

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();

    // Compute the ORB features and descriptors on an image.  Mask is ignored, as in the reference.
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints,
                    cv::OutputArray descriptors);

    int GetLevels() { return nlevels; }
    float GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Levels of the last image (fetched from the GPU when first read, see ImagePyramid).
    ImagePyramid mvImagePyramid;

    // Fetches the levels of the last image now (one device-to-host copy); reading mvImagePyramid does
    // the same implicitly.  Returns false when no image has been processed yet.
    bool MaterializePyramid();

    // The C-ABI handle behind this extractor (null before the first image): lets
    // Frame::ComputeStereoMatches call orbx_compute_stereo_matches on what both extractors hold on the GPU.
    ::orbx_extractor *handle();

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

private:
    ORBextractor(const ORBextractor &);             // the pyramid proxy points back at its extractor
    ORBextractor &operator=(const ORBextractor &);  // (the reference's callers hold extractors by pointer)
    std::unique_ptr<orbx::Extractor> impl_;  // created on first use, sized for the first image
    int impl_w_ = 0, impl_h_ = 0;
};

}  // namespace ORB_SLAM2

#endif
