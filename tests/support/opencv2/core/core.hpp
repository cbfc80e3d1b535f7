// TEST DOUBLE, not OpenCV.  The image has no OpenCV, so the drop-in shims in
// orb-slam2-chinesenotes_amd/host/ are compile- and run-checked against this minimal stand-in for the
// handful of cv:: types they touch (cv::Mat as an 8-bit matrix, KeyPoint, Point2f, Input/OutputArray).
// It is used ONLY by tests/test_shim_*.py to build tests/support/shim_driver.cpp; it is not part of
// the product and nothing of the reference is compiled against it.
#pragma once
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0

namespace cv {

struct Point2f {
    float x, y;
    Point2f() : x(0), y(0) {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
};

class Mat {
public:
    int rows, cols;
    unsigned char* data;
    size_t step;
    Mat() : rows(0), cols(0), data(nullptr), step(0) {}
    Mat(int r, int c, int /*type*/) : rows(0), cols(0), data(nullptr), step(0) { create(r, c, CV_8UC1); }
    Mat(int r, int c, int /*type*/, void* ext, size_t st) : rows(r), cols(c), data((unsigned char*)ext), step(st) {}
    void create(int r, int c, int /*type*/)
    {
        if (r == rows && c == cols && data && step == (size_t)c) return;
        buf = std::make_shared<std::vector<unsigned char>>((size_t)r * c);
        rows = r; cols = c; step = (size_t)c; data = buf->data();
    }
    void release() { buf.reset(); rows = cols = 0; data = nullptr; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    bool isContinuous() const { return step == (size_t)cols || rows <= 1; }
    template <typename T> T* ptr(int i = 0) { return reinterpret_cast<T*>(data + (size_t)i * step); }
    template <typename T> const T* ptr(int i = 0) const { return reinterpret_cast<const T*>(data + (size_t)i * step); }
    Mat row(int i) const { Mat m(1, cols, CV_8UC1, data + (size_t)i * step, step); m.buf = buf; return m; }

private:
    std::shared_ptr<std::vector<unsigned char>> buf;
};

class _InputArray {
public:
    _InputArray(const Mat& m) : m_(&m) {}
    bool empty() const { return m_->empty(); }
    Mat getMat() const { return *m_; }
private:
    const Mat* m_;
};
typedef const _InputArray& InputArray;

class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int r, int c, int t) const { m_->create(r, c, t); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }
private:
    Mat* m_;
};
typedef const _OutputArray& OutputArray;

}  // namespace cv
