// ORBmatcherHip.cc -- replacement bodies for FOUR member functions of the reference's
// ORB_SLAM2::ORBmatcher, compiled against the reference's own, unchanged include/ORBmatcher.h:
//
//   static int DescriptorDistance(const cv::Mat&, const cv::Mat&)          src/ORBmatcher.cc:46-63
//   int SearchByBoW(KeyFrame*, Frame&, std::vector<MapPoint*>&)            src/ORBmatcher.cc:552-687
//   int SearchByBoW(KeyFrame*, KeyFrame*, std::vector<MapPoint*>&)         src/ORBmatcher.cc:690-832
//   int SearchForInitialization(Frame&, Frame&, std::vector<cv::Point2f>&,
//                               std::vector<int>&, int)                    src/ORBmatcher.cc:1055-1180
//
// Build: compile the reference's src/ORBmatcher.cc with -DORB_HIP_MATCHER (after wrapping those four
// definitions in `#ifndef ORB_HIP_MATCHER`) together with this file.  The other
// routines (SearchByProjection x4, SearchForTriangulation, SearchBySim3, Fuse x2, ComputeThreeMaxima)
// keep their CPU bodies and keep calling DescriptorDistance, which stays a host function.
//
// The shim's job is only marshalling: snapshot MapPoint validity under the KeyFrame's own locks
// (GetMapPointMatches / isBad), flatten the DBoW2::FeatureVector maps to CSR, translate indices back
// to MapPoint*.  No CPU fallback: a failing GPU call throws std::runtime_error.
#include "ORBmatcher.h"

#include <cstdlib>
#include <stdexcept>
#include <string>

#include "orb_hip.h"

namespace ORB_SLAM2
{

namespace
{
// the GPU the matcher handles live on: the environment variable ORB_HIP_DEVICE (the same one the extractor shim reads;
// the reference's constructors have no room for a device index), default 0
int matcherDevice()
{
    const char* e = std::getenv("ORB_HIP_DEVICE");
    return e ? std::atoi(e) : 0;
}

struct MatcherHandle {                       // one stream + scratch per host thread (Tracking,
    orb_matcher* m = nullptr;                // LocalMapping and LoopClosing each own a thread)
    MatcherHandle()
    {
        if (orb_matcher_create(matcherDevice(), &m) != ORB_OK)
            throw std::runtime_error(std::string("ORBmatcher(HIP): orb_matcher_create failed: ") + orb_last_error());
    }
    ~MatcherHandle() { orb_matcher_destroy(m); }
};

orb_matcher* handle()
{
    static thread_local MatcherHandle h;
    return h.m;
}

void check(int rc, const char* what)
{
    if (rc != ORB_OK)
        throw std::runtime_error(std::string("ORBmatcher(HIP): ") + what + " failed: " + orb_last_error());
}

struct FlatFeatVec {
    std::vector<uint32_t> ids;
    std::vector<int32_t> offs, idx;
    orb_featvec view;
    explicit FlatFeatVec(const DBoW2::FeatureVector& fv)
    {
        offs.push_back(0);
        for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
            ids.push_back(it->first);
            for (size_t k = 0; k < it->second.size(); k++) idx.push_back((int32_t)it->second[k]);
            offs.push_back((int32_t)idx.size());
        }
        view.node_ids = ids.data();
        view.offsets = offs.data();
        view.indices = idx.data();
        view.n_nodes = (int32_t)ids.size();
    }
};

const unsigned char* rows32(const cv::Mat& d)
{
    if (!d.empty() && !d.isContinuous())
        throw std::runtime_error("ORBmatcher(HIP): descriptor matrix must be continuous (it is: ORBextractor creates it)");
    return d.data;
}
}  // namespace

int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b)
{
    return orb_hamming(a.ptr<unsigned char>(), b.ptr<unsigned char>());
}

int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches)
{
    const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
    const int nKF = (int)vpMapPointsKF.size();
    std::vector<unsigned char> valid(nKF);
    std::vector<float> angKF(nKF), angF(F.N);
    for (int i = 0; i < nKF; i++) {
        MapPoint* pMP = vpMapPointsKF[i];
        valid[i] = (pMP && !pMP->isBad()) ? 1 : 0;           // :590-595
        angKF[i] = pKF->mvKeysUn[i].angle;                   // :632
    }
    for (int i = 0; i < F.N; i++) angF[i] = F.mvKeys[i].angle;   // :634 (mvKeys, not mvKeysUn)
    FlatFeatVec fvKF(pKF->mFeatVec), fvF(F.mFeatVec);
    std::vector<int32_t> match(F.N > 0 ? F.N : 1);
    int nmatches = 0;
    check(orb_match_bow(handle(), rows32(pKF->mDescriptors), angKF.data(), valid.data(), nKF, &fvKF.view,
                        rows32(F.mDescriptors), angF.data(), F.N, &fvF.view, mfNNratio, mbCheckOrientation ? 1 : 0,
                        match.data(), &nmatches), "orb_match_bow");
    for (int i = 0; i < F.N; i++)
        if (match[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[match[i]];
    return nmatches;
}

int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12)
{
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
    vpMatches12 = std::vector<MapPoint*>(n1, static_cast<MapPoint*>(NULL));
    std::vector<unsigned char> v1(n1), v2(n2);
    std::vector<float> a1(n1), a2(n2);
    for (int i = 0; i < n1; i++) {
        v1[i] = (vpMapPoints1[i] && !vpMapPoints1[i]->isBad()) ? 1 : 0;
        a1[i] = pKF1->mvKeysUn[i].angle;
    }
    for (int i = 0; i < n2; i++) {
        v2[i] = (vpMapPoints2[i] && !vpMapPoints2[i]->isBad()) ? 1 : 0;
        a2[i] = pKF2->mvKeysUn[i].angle;
    }
    FlatFeatVec fv1(pKF1->mFeatVec), fv2(pKF2->mFeatVec);
    std::vector<int32_t> match(n1 > 0 ? n1 : 1);
    int nmatches = 0;
    check(orb_match_bow_kk(handle(), rows32(pKF1->mDescriptors), a1.data(), v1.data(), n1, &fv1.view,
                           rows32(pKF2->mDescriptors), a2.data(), v2.data(), n2, &fv2.view, mfNNratio,
                           mbCheckOrientation ? 1 : 0, match.data(), &nmatches), "orb_match_bow_kk");
    for (int i = 0; i < n1; i++)
        if (match[i] >= 0) vpMatches12[i] = vpMapPoints2[match[i]];
    return nmatches;
}

int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched,
                                        std::vector<int>& vnMatches12, int windowSize)
{
    static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint layout");
    static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "cv::Point2f layout");
    const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
    vnMatches12 = std::vector<int>(n1, -1);
    const float grid4[4] = {Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv};
    int nmatches = 0;
    check(orb_match_init(handle(), reinterpret_cast<const orb_keypoint*>(F1.mvKeysUn.data()), rows32(F1.mDescriptors), n1,
                         reinterpret_cast<const orb_keypoint*>(F2.mvKeysUn.data()), rows32(F2.mDescriptors), n2, grid4,
                         reinterpret_cast<float*>(vbPrevMatched.data()), windowSize, mfNNratio,
                         mbCheckOrientation ? 1 : 0, vnMatches12.data(), &nmatches), "orb_match_init");
    return nmatches;
}

} //namespace ORB_SLAM
