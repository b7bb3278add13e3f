// The cv:: adaptor (orbslam2_amd/cpp/ORBextractor.{h,cc}) driven the way ORB-SLAM2's Frame drives the class
// it replaces (src/Frame.cc:62-118, 256-262, 534-660 of the reference), compiled against the test-only cv::
// stand-in under tests/cpp/cv_stub.  Self-checking: every result is compared with the same call made
// straight through the C ABI; prints "OK ..." or "FAIL ...".  Built by __graft_entry__.build(), run on the
// GPU by tests/test_gpu_cpp_host.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "ORBextractor.h"
#include "orbx.hpp"

using ORB_SLAM2::ORBextractor;

// the two bodies the integration guide gives a maintainer, in a Frame / ORBmatcher of the reference's shape
namespace ORB_SLAM2 {
class ORBmatcher {
public:
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);
};
class Frame {
public:
    Frame(ORBextractor *l, ORBextractor *r) : mpORBextractorLeft(l), mpORBextractorRight(r), N(0), mb(0), mbf(0) {}
    void ExtractORB(int flag, const cv::Mat &im);
    void ComputeStereoMatches();
    ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<float> mvuRight, mvDepth;
    int N;
    float mb, mbf;
};
}  // namespace ORB_SLAM2
#include "integration_snippets.inc"

namespace ORB_SLAM2 {
// src/Frame.cc:256-262, unchanged
void Frame::ExtractORB(int flag, const cv::Mat &im)
{
    if (flag == 0)
        (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
    else
        (*mpORBextractorRight)(im, cv::Mat(), mvKeysRight, mDescriptorsRight);
}
}  // namespace ORB_SLAM2

#define REQUIRE(cond, what)               \
    do {                                  \
        if (!(cond)) {                    \
            printf("FAIL %s\n", what);    \
            return 1;                     \
        }                                 \
    } while (0)

int main(int argc, char **argv)
{
    const int w = argc > 1 ? atoi(argv[1]) : 752, h = argc > 2 ? atoi(argv[2]) : 480;
    const int stride = w + 24;  // a cv::Mat ROI: rows further apart than the width
    std::vector<uint8_t> store((size_t)stride * h), rstore((size_t)stride * h);
    orbx::check(orbx_synth_image(0x0B5EED07ull, w, h, 0, store.data(), stride), "orbx_synth_image");
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            rstore[(size_t)y * stride + x] = store[(size_t)y * stride + (x + 12 < w ? x + 12 : w - 1)];
    cv::Mat imLeft(h, w, CV_8UC1, store.data(), stride), imRight(h, w, CV_8UC1, rstore.data(), stride);
    try {
        orbx::Extractor ref(1200, 1.2f, 8, 20, 7, w, h);  // first: without a GPU this throws here, not in a thread
        ORBextractor left(1200, 1.2f, 8, 20, 7), right(1200, 1.2f, 8, 20, 7);  // Examples/Stereo/EuRoC.yaml
        REQUIRE(left.handle() == nullptr, "no device work before the first image");
        REQUIRE(left.mvImagePyramid.size() == 8, "mvImagePyramid.size() == nlevels before the first image");
        REQUIRE(!left.MaterializePyramid(), "MaterializePyramid without an image");

        // empty image: outputs untouched (src/ORBextractor.cc:1117-1118)
        std::vector<cv::KeyPoint> keep(3);
        cv::Mat keepd(2, 32, CV_8U);
        left(cv::Mat(), cv::Mat(), keep, keepd);
        REQUIRE(keep.size() == 3 && keepd.rows == 2, "empty image must leave the outputs untouched");

        // the stereo constructor: two extractors in two threads (src/Frame.cc:79-82)
        ORB_SLAM2::Frame F(&left, &right);
        std::thread threadLeft(&ORB_SLAM2::Frame::ExtractORB, &F, 0, imLeft);
        std::thread threadRight(&ORB_SLAM2::Frame::ExtractORB, &F, 1, imRight);
        threadLeft.join();
        threadRight.join();
        F.N = (int)F.mvKeys.size();
        REQUIRE(F.N > 800 && F.mDescriptors.rows == F.N && F.mDescriptors.cols == 32, "left result shape");
        REQUIRE((int)F.mvKeysRight.size() == F.mDescriptorsRight.rows && F.mDescriptorsRight.rows > 800,
                "right result shape");

        // the same image straight through the C ABI
        std::vector<orbx_keypoint> rk;
        std::vector<uint8_t> rd;
        ref(store.data(), w, h, stride, rk, rd);
        REQUIRE((int)rk.size() == F.N, "keypoint count equals the C ABI's");
        REQUIRE(memcmp(static_cast<const void *>(F.mvKeys.data()), rk.data(), rk.size() * sizeof(orbx_keypoint)) == 0,
                "keypoints equal the C ABI's");
        for (int i = 0; i < F.N; i++)
            REQUIRE(memcmp(F.mDescriptors.ptr(i), rd.data() + (size_t)i * 32, 32) == 0, "descriptors equal the C ABI's");
        REQUIRE(F.mvKeys[0].class_id == -1 && F.mvKeys[0].octave == 0 && F.mvKeys[0].size == 31.f, "cv::KeyPoint fields");

        // getters against the library's tables (include/ORBextractor.h:63-83)
        REQUIRE(left.GetLevels() == 8 && left.GetScaleFactor() == 1.2f, "GetLevels / GetScaleFactor");
        REQUIRE(left.GetScaleFactors() == ref.GetScaleFactors() &&
                    left.GetInverseScaleFactors() == ref.GetInverseScaleFactors() &&
                    left.GetScaleSigmaSquares() == ref.GetScaleSigmaSquares() &&
                    left.GetInverseScaleSigmaSquares() == ref.GetInverseScaleSigmaSquares(),
                "scale tables equal the library's");

        // DescriptorDistance on two cv::Mat rows (src/ORBmatcher.cc:1744-1760)
        const int d01 = ORB_SLAM2::ORBmatcher::DescriptorDistance(F.mDescriptors.row(0), F.mDescriptors.row(1));
        REQUIRE(d01 == orbx_descriptor_distance(rd.data(), rd.data() + 32) && d01 > 0, "DescriptorDistance");

        // ComputeStereoMatches on the device data of both handles: no pyramid travels
        F.mb = 0.11f, F.mbf = 47.9f;
        F.ComputeStereoMatches();
        int nmatch = 0, near12 = 0;
        for (int i = 0; i < F.N; i++)
            if (F.mvuRight[i] >= 0) {
                nmatch++;
                const float d = F.mvKeys[i].pt.x - F.mvuRight[i];
                near12 += d > 11.5f && d < 12.5f;
            }
        REQUIRE(nmatch > 200 && near12 * 2 > nmatch, "stereo matches find the 12-pixel shift");

        // mvImagePyramid as src/Frame.cc:541,631,643,648 reads it: fetched now, on first access
        const int nRows = left.mvImagePyramid[0].rows;
        REQUIRE(nRows == h && left.mvImagePyramid[0].cols == w, "level 0 size");
        for (int l = 0; l < 8; l++) {
            int pw = 0, ph = 0;
            std::vector<uint8_t> want = ref.pyramid_level(l, 19, pw, ph);
            const cv::Mat &L = left.mvImagePyramid[l];
            REQUIRE(L.cols == pw - 38 && L.rows == ph - 38 && (int)L.step == pw, "level is a ROI of its padded buffer");
            // the ROI and the 19 reflected pixels around it
            for (int y = -19; y < L.rows + 19; y++)
                REQUIRE(memcmp(L.data + (ptrdiff_t)y * (ptrdiff_t)L.step - 19, want.data() + (size_t)(y + 19) * pw, pw) == 0,
                        "padded level equals orbx_pyramid_level");
        }
        {   // the window reads of the sliding search (src/Frame.cc:631)
            const cv::KeyPoint &kpL = F.mvKeys[F.N / 2];
            const float sf = left.GetInverseScaleFactors()[kpL.octave];
            const int u = (int)(kpL.pt.x * sf + 0.5f), v = (int)(kpL.pt.y * sf + 0.5f), ww = 5;
            cv::Mat IL = left.mvImagePyramid[kpL.octave].rowRange(v - ww, v + ww + 1).colRange(u - ww, u + ww + 1);
            REQUIRE(IL.rows == 11 && IL.cols == 11 && IL.at<uchar>(ww, ww) == left.mvImagePyramid[kpL.octave].at<uchar>(v, u),
                    "window view");
        }
        // a second image: the levels read afterwards are the new image's
        left(imRight, cv::Mat(), F.mvKeys, F.mDescriptors);
        std::vector<cv::Mat> copy = left.mvImagePyramid;  // the vector conversion
        REQUIRE(copy.size() == 8 && memcmp(copy[0].ptr(7), rstore.data() + (size_t)7 * stride, w) == 0,
                "levels follow the last image");

        // zero keypoints: keypoints cleared, descriptors released (src/ORBextractor.cc:1135-1136)
        std::vector<uint8_t> flat((size_t)w * h, 128);
        left(cv::Mat(h, w, CV_8UC1, flat.data()), cv::Mat(), F.mvKeys, F.mDescriptors);
        REQUIRE(F.mvKeys.empty() && F.mDescriptors.empty(), "constant image gives empty outputs");

        // a larger image than the first one: the adaptor re-creates its handle
        std::vector<uint8_t> big((size_t)(w + 64) * (h + 32));
        orbx::check(orbx_synth_image(5, w + 64, h + 32, 0, big.data(), w + 64), "orbx_synth_image");
        left(cv::Mat(h + 32, w + 64, CV_8UC1, big.data()), cv::Mat(), F.mvKeys, F.mDescriptors);
        REQUIRE(F.mvKeys.size() > 800 && left.mvImagePyramid[0].cols == w + 64, "larger image");

        printf("OK n=%d right=%d stereo=%d near12=%d d01=%d\n", F.N, (int)F.mvKeysRight.size(), nmatch, near12, d01);
    } catch (const std::exception &e) {
        printf("FAIL %s\n", e.what());
        return 1;
    }
    return 0;
}
