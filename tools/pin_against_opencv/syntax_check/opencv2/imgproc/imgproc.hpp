// DECLARATIONS ONLY, not OpenCV: lets tests/test_pin_kit.py run `g++ -fsyntax-only` over pin_orb.cpp in a container that has
// no OpenCV (together with tests/support/opencv2/core/core.hpp).  Nothing is compiled to code or linked against these.
#pragma once
#include <opencv2/core/core.hpp>
namespace cv {
struct Point2i { int x, y; Point2i() : x(0), y(0) {} Point2i(int a, int b) : x(a), y(b) {} };
typedef Point2i Point;
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
enum { BORDER_REFLECT_101 = 4, BORDER_ISOLATED = 16, INTER_LINEAR = 1 };
void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_REFLECT_101);
void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void copyMakeBorder(InputArray src, OutputArray dst, int top, int bottom, int left, int right, int borderType);
float fastAtan2(float y, float x);
}  // namespace cv
