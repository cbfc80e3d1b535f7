// DECLARATIONS ONLY, not OpenCV (see ../imgproc/imgproc.hpp).
#pragma once
#include <vector>
#include <opencv2/core/core.hpp>
namespace cv {
void FAST(InputArray image, std::vector<KeyPoint>& keypoints, int threshold, bool nonmaxSuppression = true);
}
