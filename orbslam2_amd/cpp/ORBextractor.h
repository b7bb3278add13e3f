// Drop-in for include/ORBextractor.h of ORB-SLAM2 (lianbin/orbslam2): the same class name, constructor,
// operator(), getters and public mvImagePyramid (include/ORBextractor.h:45-111 of the reference),
// computed by liborbx on an MI355X.  Needs OpenCV headers, like the file it replaces; everything
// OpenCV-free lives in orbx.hpp.  ExtractorNode (reference :32-43) is gone: no other file uses it.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <memory>
#include <vector>

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>

namespace orbx {
class Extractor;
}

namespace ORB_SLAM2 {

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

    // Levels of the last image, each a ROI inside a buffer padded by 19 reflected pixels
    // (read by Frame::ComputeStereoMatches, src/Frame.cc:541,631,643,648).
    std::vector<cv::Mat> mvImagePyramid;

    // The C-ABI handle behind this extractor (null before the first image): lets
    // Frame::ComputeStereoMatches call orbx_compute_stereo_matches on what both extractors hold on the GPU.
    struct orbx_extractor *handle();

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

private:
    std::unique_ptr<orbx::Extractor> impl_;  // created on first use, sized for the first image
    int impl_w_ = 0, impl_h_ = 0;
};

}  // namespace ORB_SLAM2

#endif
