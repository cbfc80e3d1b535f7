// KeyFrameHip.cc -- replacement body for void KeyFrame::ComputeBoW(), reference src/KeyFrame.cc:64-73, compiled against the
// reference's own include/KeyFrame.h: the same DBoW2 transform(..., 4) as Frame::ComputeBoW (BowHip.h: tree descent on the GPU,
// one flattened vocabulary per ORBVocabulary object shared with FrameHip.cc).
// Build: wrap the definition in src/KeyFrame.cc in `#ifndef ORB_HIP_KEYFRAME` and add this file.  No CPU fallback.
#include "KeyFrame.h"
#include "BowHip.h"

namespace ORB_SLAM2
{

void KeyFrame::ComputeBoW()
{
    if (mBowVec.empty() || mFeatVec.empty())      // :66
        hipbow::transform(mpORBvocabulary, mDescriptors, mBowVec, mFeatVec, 4);      // :70
}

}  // namespace ORB_SLAM2
