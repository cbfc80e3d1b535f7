// compat/opencv/cv.h -- OpenCV 4 removed the legacy <opencv/cv.h> that the reference's include/ORBextractor.h includes
// (ORB-SLAM2 predates it).  CMakeLists.txt puts this directory on the include path only when OpenCV >= 4 is found.
#pragma once
#include <opencv2/opencv.hpp>
