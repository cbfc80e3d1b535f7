// TEST DOUBLE, not OpenCV.  The image has no OpenCV, so the drop-in shims in
// orb-slam2-chinesenotes_amd/host/ are compile- and run-checked against this minimal stand-in for the
// handful of cv:: types they touch: cv::Mat as an 8-bit image / descriptor matrix AND as a small float
// matrix with the algebra the reference's projection lines use (rowRange / colRange / col / row, t(), *, +, -,
// scalar * and /, dot, cv::norm, at<float>), KeyPoint, Point2f, Input/OutputArray.
// Float arithmetic here is plain left-to-right float loops; it is NOT claimed to round like OpenCV's gemm.
// It is used ONLY by tests/test_shim*.py to build tests/support/shim_driver.cpp; it is not part of
// the product and nothing of the reference is compiled against it.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5

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
    size_t step;                                  // bytes per row
    Mat() : rows(0), cols(0), data(nullptr), step(0), type_(CV_8U) {}
    Mat(int r, int c, int type) : rows(0), cols(0), data(nullptr), step(0), type_(CV_8U) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t st = 0) : rows(r), cols(c), data((unsigned char*)ext), step(st ? st : (size_t)c * esz(type)), type_(type) {}   // (0 = AUTO_STEP)
    static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
    void create(int r, int c, int type)
    {
        if (r == rows && c == cols && data && type == type_ && step == (size_t)c * esz(type)) return;
        buf = std::make_shared<std::vector<unsigned char>>((size_t)r * c * esz(type));
        rows = r; cols = c; type_ = type; step = (size_t)c * esz(type); data = buf->data();
    }
    void release() { buf.reset(); rows = cols = 0; data = nullptr; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    bool isContinuous() const { return step == (size_t)cols * esz(type_) || rows <= 1; }
    template <typename T> T* ptr(int i = 0) { return reinterpret_cast<T*>(data + (size_t)i * step); }
    template <typename T> const T* ptr(int i = 0) const { return reinterpret_cast<const T*>(data + (size_t)i * step); }
    template <typename T> T& at(int i, int j) { return ptr<T>(i)[j]; }
    template <typename T> const T& at(int i, int j) const { return ptr<T>(i)[j]; }
    template <typename T> T& at(int i) { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }
    template <typename T> const T& at(int i) const { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }
    // sub-matrices are headers over the same storage, as in OpenCV
    Mat rowRange(int a, int b) const { Mat m(b - a, cols, type_, data + (size_t)a * step, step); m.buf = buf; return m; }
    Mat colRange(int a, int b) const { Mat m(rows, b - a, type_, data + (size_t)a * esz(type_), step); m.buf = buf; return m; }
    Mat row(int i) const { return rowRange(i, i + 1); }
    Mat col(int j) const { return colRange(j, j + 1); }
    Mat clone() const
    {
        Mat m(rows, cols, type_);
        for (int i = 0; i < rows; i++) std::memcpy(m.data + (size_t)i * m.step, data + (size_t)i * step, (size_t)cols * esz(type_));
        return m;
    }
    Mat t() const
    {
        Mat m(cols, rows, CV_32F);
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) m.at<float>(j, i) = at<float>(i, j);
        return m;
    }
    float dot(const Mat& o) const
    {
        float s = 0;
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) s += at<float>(i, j) * o.at<float>(i, j);
        return s;
    }
    static Mat eye(int r, int c, int type)
    {
        Mat m(r, c, type);
        std::memset(m.data, 0, (size_t)r * m.step);
        for (int i = 0; i < r && i < c; i++) m.at<float>(i, i) = 1.f;
        return m;
    }
    static Mat zeros(int r, int c, int type)
    {
        Mat m(r, c, type);
        std::memset(m.data, 0, (size_t)r * m.step);
        return m;
    }

private:
    int type_;
    std::shared_ptr<std::vector<unsigned char>> buf;
};

inline Mat operator*(const Mat& a, const Mat& b)
{
    Mat m(a.rows, b.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < b.cols; j++) {
            float s = 0;
            for (int k = 0; k < a.cols; k++) s += a.at<float>(i, k) * b.at<float>(k, j);
            m.at<float>(i, j) = s;
        }
    return m;
}
inline Mat ewise(const Mat& a, const Mat& b, float sb)
{
    Mat m(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = a.at<float>(i, j) + sb * b.at<float>(i, j);
    return m;
}
inline Mat operator+(const Mat& a, const Mat& b) { return ewise(a, b, 1.f); }
inline Mat operator-(const Mat& a, const Mat& b) { return ewise(a, b, -1.f); }
inline Mat operator*(double s, const Mat& a)
{
    Mat m(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = (float)(s * a.at<float>(i, j));
    return m;
}
inline Mat operator*(const Mat& a, double s) { return s * a; }
inline Mat operator/(const Mat& a, double s)
{
    Mat m(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = (float)(a.at<float>(i, j) / s);
    return m;
}
inline Mat operator-(const Mat& a) { return -1.0 * a; }
inline double norm(const Mat& a) { return std::sqrt((double)a.dot(a)); }

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
