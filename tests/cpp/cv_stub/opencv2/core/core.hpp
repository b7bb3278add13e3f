// TEST-ONLY stand-in for the handful of OpenCV declarations that orbslam2_amd/cpp/ORBextractor.{h,cc}
// use, so that the adaptor can be type-checked, linked and run in an image without OpenCV.  It is not
// an oracle and not part of the product: it holds no image processing, only a reference-counted 8-bit
// matrix header with the member names of cv::Mat, and the argument proxies of cv::InputArray /
// cv::OutputArray.  With a real OpenCV on the include path this directory is simply not used.
#ifndef ORBX_TEST_CV_STUB_CORE_HPP
#define ORBX_TEST_CV_STUB_CORE_HPP

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0

typedef unsigned char uchar;

namespace cv {

struct Point2f {
    float x, y;
    Point2f() : x(0), y(0) {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

struct Rect {
    int x, y, width, height;
    Rect() : x(0), y(0), width(0), height(0) {}
    Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
};

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w_, int h_) : width(w_), height(h_) {}
};

class Mat {
public:
    int rows, cols;
    uchar *data;
    std::size_t step;

    Mat() : rows(0), cols(0), data(nullptr), step(0) {}
    Mat(int r, int c, int type) : rows(0), cols(0), data(nullptr), step(0) { create(r, c, type); }
    // header over memory the caller owns
    Mat(int r, int c, int /*type*/, void *ext, std::size_t st = 0)
        : rows(r), cols(c), data(static_cast<uchar *>(ext)), step(st ? st : (std::size_t)c)
    {
    }
    void create(int r, int c, int /*type*/)
    {
        if (data && r == rows && c == cols && isContinuous())
            return;  // cv::Mat::create keeps a buffer of the right size
        store_ = std::shared_ptr<std::vector<uchar>>(new std::vector<uchar>((std::size_t)r * c));
        rows = r, cols = c, step = (std::size_t)c;
        data = store_->data();
    }
    void release()
    {
        store_.reset();
        rows = cols = 0, data = nullptr, step = 0;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    bool isContinuous() const { return step == (std::size_t)cols || rows == 1; }
    uchar *ptr(int y = 0) { return data + (std::size_t)y * step; }
    const uchar *ptr(int y = 0) const { return data + (std::size_t)y * step; }
    template <typename T> T *ptr(int y = 0) { return reinterpret_cast<T *>(data + (std::size_t)y * step); }
    template <typename T> const T *ptr(int y = 0) const { return reinterpret_cast<const T *>(data + (std::size_t)y * step); }
    template <typename T> T &at(int y, int x) { return reinterpret_cast<T *>(data + (std::size_t)y * step)[x]; }
    template <typename T> const T &at(int y, int x) const { return reinterpret_cast<const T *>(data + (std::size_t)y * step)[x]; }
    Mat operator()(const Rect &r) const
    {
        Mat m(*this);
        m.data = data + (std::size_t)r.y * step + r.x;
        m.rows = r.height, m.cols = r.width;
        return m;
    }
    Mat rowRange(int a, int b) const { return (*this)(Rect(0, a, cols, b - a)); }
    Mat colRange(int a, int b) const { return (*this)(Rect(a, 0, b - a, rows)); }
    Mat row(int y) const { return rowRange(y, y + 1); }
    // single channel only; the source must be continuous, as cv::Mat::reshape requires
    Mat reshape(int /*cn*/, int new_rows) const
    {
        Mat m(*this);
        const std::size_t total = (std::size_t)rows * cols;
        m.rows = new_rows, m.cols = (int)(total / new_rows), m.step = (std::size_t)m.cols;
        return m;
    }
    Mat clone() const
    {
        Mat m(rows, cols, CV_8UC1);
        for (int y = 0; y < rows; y++)
            std::memcpy(m.ptr(y), ptr(y), (std::size_t)cols);
        return m;
    }

private:
    std::shared_ptr<std::vector<uchar>> store_;
};

// cv::InputArray is `const _InputArray &`; the proxies bind to a Mat
class _InputArray {
public:
    _InputArray() : m_(nullptr) {}
    _InputArray(const Mat &m) : m_(&m) {}
    bool empty() const { return !m_ || m_->empty(); }
    Mat getMat() const { return m_ ? *m_ : Mat(); }

protected:
    const Mat *m_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() {}
    _OutputArray(Mat &m) : _InputArray(m) {}
    void create(int rows, int cols, int type) const { const_cast<Mat *>(m_)->create(rows, cols, type); }
    void release() const { const_cast<Mat *>(m_)->release(); }
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;

}  // namespace cv

#endif
