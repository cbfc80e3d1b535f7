// DECLARATIONS ONLY, not OpenCV (see ../imgproc/imgproc.hpp).
#pragma once
