// DECLARATIONS ONLY, not OpenCV (see ../opencv2/imgproc/imgproc.hpp): the legacy header the reference's ORBextractor.h includes.
#pragma once
#include <opencv2/core/core.hpp>
#include <opencv2/imgproc/imgproc.hpp>
#include <opencv2/features2d/features2d.hpp>
