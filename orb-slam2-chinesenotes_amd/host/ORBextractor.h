// ORBextractor.h -- drop-in replacement for the reference's include/ORBextractor.h.
//
// Same namespace, class name, constructor, operator(), getters and the public data member
// mvImagePyramid as reference include/ORBextractor.h:46-112, so Frame.cc (:73-79, :262-268,
// :520-633) and Tracking.cc (:117-126) compile and link unchanged.  The body runs on an MI355X
// through the C ABI of include/orb_hip.h; there is NO CPU fallback: a missing GPU / library
// failure throws std::runtime_error from the constructor or from operator().
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <vector>
#include <opencv2/core/core.hpp>

struct orb_extractor;   // include/orb_hip.h

namespace ORB_SLAM2
{

class ORBextractor
{
public:
    enum {HARRIS_SCORE=0, FAST_SCORE=1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // Compute the ORB features and descriptors on an image (mask is ignored, as in the reference).
    void operator()(cv::InputArray image, cv::InputArray mask,
                    std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Interiors of the pyramid levels of the last image (no 19-px border; nothing on the path
    // reads it).  Filled after every operator() unless SetPyramidDownload(false) was called
    // (monocular/RGB-D tracking never reads it; stereo does: Frame.cc:520,611,626,633).
    std::vector<cv::Mat> mvImagePyramid;
    void SetPyramidDownload(bool on) { mbDownloadPyramid = on; }
    orb_extractor* Handle() { return mpHandle; }      // for orb_stereo_match on the device-resident pyramids
    // GPU that extractors constructed from now on live on (the constructor signature is the reference's and has no
    // room for it); also read from the environment variable ORB_HIP_DEVICE.  Default 0.
    static void SetDefaultDevice(int device);
    // Additional: which OpenCV generation's integer Gaussian the descriptors are blurred with (orb_gaussian_preset in orb_hip.h:
    // 0 = {18,34,49,55}, OpenCV 2.4 ... the first fixed-point versions; 1 = {18,34,48,56}, the error-diffused fixed-point kernel of
    // later 3.4.x / 4.x).  Applies to extractors constructed afterwards; also read from ORB_HIP_GAUSS.  Default 0.
    static void SetGaussianPreset(int preset);

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor;
    std::vector<float> mvInvScaleFactor;
    std::vector<float> mvLevelSigma2;
    std::vector<float> mvInvLevelSigma2;

    orb_extractor* mpHandle;
    bool mbDownloadPyramid;
    // per-object pinned staging, kept across calls: keypoints, descriptors and the whole pyramid slab (the level
    // cv::Mats of mvImagePyramid are headers over it)
    void* mpStage;
    size_t mnStageBytes;
    size_t mnPyramidBytes;
};

} //namespace ORB_SLAM

#endif
