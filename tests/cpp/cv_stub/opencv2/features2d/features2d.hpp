// TEST-ONLY stand-in (see ../core/core.hpp): cv::KeyPoint with the seven fields and 28-byte layout of
// OpenCV's class.
#ifndef ORBX_TEST_CV_STUB_FEATURES2D_HPP
#define ORBX_TEST_CV_STUB_FEATURES2D_HPP

#include "../core/core.hpp"

namespace cv {

class KeyPoint {
public:
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_)
    {
    }
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};

}  // namespace cv

#endif
