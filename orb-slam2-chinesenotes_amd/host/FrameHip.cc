// FrameHip.cc -- replacement bodies for two member functions of the reference's ORB_SLAM2::Frame, compiled against the
// reference's own, unchanged include/Frame.h (optional bindings):
//
//   void Frame::ComputeStereoMatches()      src/Frame.cc:513-699   -> orb_stereo_match on the pyramids that the two
//                                                                     extractor handles keep on the device
//   void Frame::ComputeBoW()                src/Frame.cc:425-433   -> hipbow::transform (BowHip.h): DBoW2's
//                                                                     transform(features, BowVector&, FeatureVector&, 4) with
//                                                                     the tree descent on the GPU; the std::map containers
//                                                                     are filled on the host through DBoW2's own methods
// Build: wrap the two definitions of src/Frame.cc in `#ifndef ORB_HIP_FRAME` and add this file.  No CPU fallback.
#include <stdexcept>
#include <string>

#include "Frame.h"
#include "ORBextractor.h"
#include "BowHip.h"

namespace ORB_SLAM2
{

void Frame::ComputeStereoMatches()
{
    mvuRight = std::vector<float>(N, -1.0f);      // :515-516
    mvDepth = std::vector<float>(N, -1.0f);
    if (N == 0) return;
    static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint layout");
    hipbow::check(orb_stereo_match(mpORBextractorLeft->Handle(), mpORBextractorRight->Handle(),
                                   reinterpret_cast<const orb_keypoint*>(mvKeys.data()), mDescriptors.data, N,
                                   reinterpret_cast<const orb_keypoint*>(mvKeysRight.data()), mDescriptorsRight.data, (int)mvKeysRight.size(), mb,
                                   mbf, mvuRight.data(), mvDepth.data()), "orb_stereo_match");
}

void Frame::ComputeBoW()
{
    if (mBowVec.empty())                          // :427
        hipbow::transform(mpORBvocabulary, mDescriptors, mBowVec, mFeatVec, 4);
}

}  // namespace ORB_SLAM2
