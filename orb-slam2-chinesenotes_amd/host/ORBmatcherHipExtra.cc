// ORBmatcherHipExtra.cc -- replacement bodies for the OTHER eight member functions of the reference's
// ORB_SLAM2::ORBmatcher (SURVEY 8b lets them keep their CPU bodies; these are the optional bindings of INTEGRATION.md 2b
// as compiled code), against the reference's own, unchanged include/ORBmatcher.h:
//
//   int SearchByProjection(Frame&, const vector<MapPoint*>&, float th)                         src/ORBmatcher.cc:73-157
//   int SearchByProjection(Frame&, const Frame&, float th, bool bMono)                         :160-300
//   int SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, float th, int ORBdist)    :303-440
//   int SearchByProjection(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, vector<MapPoint*>&, int th)   :443-550
//   int SearchBySim3(KeyFrame*, KeyFrame*, vector<MapPoint*>&, s12, R12, t12, th)               :835-1025
//   int SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat F12, vector<pair<size_t,size_t>>&, bool)      :1183-1359
//   int Fuse(KeyFrame*, const vector<MapPoint*>&, float th)                                     :1364-1480
//   int Fuse(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, float th, vector<MapPoint*>&)    :1483-1633
//
// Division of labour (include/orb_hip.h): everything that is cv::Mat arithmetic on MapPoints -- the projection, the
// frustum / distance / viewing-angle gates, PredictScale -- stays here on the host, written with the reference's own
// expressions (cited per function), and produces one orb_proj_query per MapPoint; the grid query, the Hamming search,
// the order-dependent assignment rules and the rotation histogram run on the GPU; the map surgery of Fuse
// (Replace / AddObservation / AddMapPoint) is applied here afterwards in the reference's order.
// Build: with ORBmatcherHip.cc, after wrapping these eight definitions of src/ORBmatcher.cc in
// `#ifndef ORB_HIP_MATCHER_EXTRA` (or weakening their symbols, tools/weaken_matcher_symbols.sh).  No CPU fallback.
#include "ORBmatcher.h"

#include <climits>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include <string>

#include "ORBmatcherHipDebug.h"
#include "orb_hip.h"

namespace ORB_SLAM2
{

namespace hipshim
{
static thread_local std::vector<orb_proj_query> g_lastQueries;
const std::vector<orb_proj_query>& LastProjectionQueries() { return g_lastQueries; }
static thread_local float g_lastEx = 0.f, g_lastEy = 0.f;
void LastEpipole(float* ex, float* ey) { *ex = g_lastEx; *ey = g_lastEy; }
}  // namespace hipshim

namespace
{
int matcherDevice()
{
    const char* e = std::getenv("ORB_HIP_DEVICE");
    return e ? std::atoi(e) : 0;
}
struct MatcherHandle {                       // one stream + scratch per host thread, as in ORBmatcherHip.cc
    orb_matcher* m = nullptr;
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
    if (rc != ORB_OK) throw std::runtime_error(std::string("ORBmatcher(HIP): ") + what + " failed: " + orb_last_error());
}
const unsigned char* rows32(const cv::Mat& d)
{
    if (!d.empty() && !d.isContinuous()) throw std::runtime_error("ORBmatcher(HIP): descriptor matrix must be continuous");
    return d.data;
}
static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint layout");
const orb_keypoint* kps(const std::vector<cv::KeyPoint>& v) { return reinterpret_cast<const orb_keypoint*>(v.data()); }

orb_proj_query deadQuery()
{
    orb_proj_query q;
    q.x = q.y = q.r = 0.f;
    q.min_level = q.max_level = -1;
    q.ur = 0.f;
    q.er_max = std::numeric_limits<float>::infinity();
    q.flags = 0;
    return q;
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
        view.node_ids = ids.data(); view.offsets = offs.data(); view.indices = idx.data(); view.n_nodes = (int32_t)ids.size();
    }
};
}  // namespace

float ORBmatcher::RadiusByViewingCos(const float& viewCos)                    // :1653-1660
{
    if (viewCos > 0.998) return 2.5;
    else return 4.0;
}

// ------------------------------------------------------------------------------------------------ :73-157
int ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th)
{
    const bool bFactor = th != 1.0;
    const int nq = (int)vpMapPoints.size();
    std::vector<orb_proj_query> q(nq, deadQuery());
    std::vector<unsigned char> qd((size_t)nq * 32, 0);
    for (int iMP = 0; iMP < nq; iMP++) {
        MapPoint* pMP = vpMapPoints[iMP];
        if (!pMP->mbTrackInView) continue;                                    // :85-88
        if (pMP->isBad()) continue;
        const int& nPredictedLevel = pMP->mnTrackScaleLevel;
        float r = RadiusByViewingCos(pMP->mTrackViewCos);                     // :91-93
        if (bFactor) r *= th;
        q[iMP].x = pMP->mTrackProjX;                                          // :96-98
        q[iMP].y = pMP->mTrackProjY;
        q[iMP].r = r * F.mvScaleFactors[nPredictedLevel];
        q[iMP].min_level = nPredictedLevel - 1;
        q[iMP].max_level = nPredictedLevel;
        q[iMP].ur = pMP->mTrackProjXR;                                        // :113-118
        q[iMP].er_max = r * F.mvScaleFactors[nPredictedLevel];
        q[iMP].flags = 1 | (pMP->Observations() > 0 ? 2 : 0);
        const cv::Mat d = pMP->GetDescriptor();
        std::memcpy(&qd[(size_t)iMP * 32], d.ptr<unsigned char>(0), 32);
    }
    hipshim::g_lastQueries = q;
    std::vector<unsigned char> occ(F.N > 0 ? F.N : 1, 0);
    for (int i = 0; i < F.N; i++) occ[i] = (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) ? 1 : 0;   // :109-111
    const float grid4[4] = {Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv};
    std::vector<int32_t> cur(F.N > 0 ? F.N : 1, -1);
    int nmatches = 0;
    check(orb_match_projection(handle(), 1, q.data(), qd.data(), nullptr, nq, kps(F.mvKeysUn), rows32(F.mDescriptors),
                               F.mvuRight.empty() ? nullptr : F.mvuRight.data(), occ.data(), F.N, grid4, mfNNratio, TH_HIGH, 0,
                               cur.data(), &nmatches), "orb_match_projection");
    for (int i = 0; i < F.N; i++)
        if (cur[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[cur[i]];              // :148
    return nmatches;
}

// ------------------------------------------------------------------------------------------------ :160-300
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono)
{
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);       // :172-181
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat twc = -Rcw.t() * tcw;
    const cv::Mat Rlw = LastFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tlw = LastFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat tlc = Rlw * twc + tlw;
    const bool bForward = tlc.at<float>(2) > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc.at<float>(2) > CurrentFrame.mb && !bMono;

    const int nq = LastFrame.N;
    std::vector<orb_proj_query> q(nq > 0 ? nq : 1, deadQuery());
    std::vector<unsigned char> qd((size_t)(nq > 0 ? nq : 1) * 32, 0);
    std::vector<float> qa(nq > 0 ? nq : 1, 0.f);
    for (int i = 0; i < nq; i++) {
        MapPoint* pMP = LastFrame.mvpMapPoints[i];
        if (!pMP || LastFrame.mvbOutlier[i]) continue;                         // :186-189
        cv::Mat x3Dw = pMP->GetWorldPos();                                     // :192-212
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0);
        const float yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        if (invzc < 0) continue;
        float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        int nLastOctave = LastFrame.mvKeys[i].octave;
        float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];          // :215
        q[i].x = u; q[i].y = v; q[i].r = radius;
        if (bForward) { q[i].min_level = nLastOctave; q[i].max_level = -1; }   // :219-225
        else if (bBackward) { q[i].min_level = 0; q[i].max_level = nLastOctave; }
        else { q[i].min_level = nLastOctave - 1; q[i].max_level = nLastOctave + 1; }
        q[i].ur = u - CurrentFrame.mbf * invzc;                                // :238-244
        q[i].er_max = radius;
        q[i].flags = 1 | (pMP->Observations() > 0 ? 2 : 0);
        const cv::Mat d = pMP->GetDescriptor();
        std::memcpy(&qd[(size_t)i * 32], d.ptr<unsigned char>(0), 32);
        qa[i] = LastFrame.mvKeysUn[i].angle;                                   // :264
    }
    hipshim::g_lastQueries = q;
    const int n = CurrentFrame.N;
    std::vector<unsigned char> occ(n > 0 ? n : 1, 0);
    for (int i = 0; i < n; i++)
        occ[i] = (CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations() > 0) ? 1 : 0;   // :233-235
    const float grid4[4] = {Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv};
    std::vector<int32_t> cur(n > 0 ? n : 1, -1);
    int nmatches = 0;
    check(orb_match_projection(handle(), 0, q.data(), qd.data(), qa.data(), nq, kps(CurrentFrame.mvKeysUn), rows32(CurrentFrame.mDescriptors),
                               CurrentFrame.mvuRight.empty() ? nullptr : CurrentFrame.mvuRight.data(), occ.data(), n, grid4, 0.f,
                               TH_HIGH, mbCheckOrientation ? 1 : 0, cur.data(), &nmatches), "orb_match_projection");
    for (int i = 0; i < n; i++) {
        if (cur[i] >= 0) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[cur[i]];        // :257
        else if (cur[i] == -2) CurrentFrame.mvpMapPoints[i] = static_cast<MapPoint*>(NULL);    // :289
    }
    return nmatches;
}

// ------------------------------------------------------------------------------------------------ :303-440
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                                   const int ORBdist)
{
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);       // :308-310
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -Rcw.t() * tcw;
    const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
    const int nq = (int)vpMPs.size();
    std::vector<orb_proj_query> q(nq > 0 ? nq : 1, deadQuery());
    std::vector<unsigned char> qd((size_t)(nq > 0 ? nq : 1) * 32, 0);
    std::vector<float> qa(nq > 0 ? nq : 1, 0.f);
    for (int i = 0; i < nq; i++) {
        MapPoint* pMP = vpMPs[i];
        if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;         // :322-326
        cv::Mat x3Dw = pMP->GetWorldPos();                                     // :329-357
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0);
        const float yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        cv::Mat PO = x3Dw - Ow;
        float dist3D = cv::norm(PO);
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
        const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
        q[i].x = u; q[i].y = v; q[i].r = radius;
        q[i].min_level = nPredictedLevel - 1;                                  // :362-364
        q[i].max_level = nPredictedLevel + 1;
        q[i].flags = 1 | 2;                                                    // every assigned feature blocks (:377-378)
        const cv::Mat d = pMP->GetDescriptor();
        std::memcpy(&qd[(size_t)i * 32], d.ptr<unsigned char>(0), 32);
        qa[i] = pKF->mvKeysUn[i].angle;                                        // :399
    }
    hipshim::g_lastQueries = q;
    const int n = CurrentFrame.N;
    std::vector<unsigned char> occ(n > 0 ? n : 1, 0);
    for (int i = 0; i < n; i++) occ[i] = CurrentFrame.mvpMapPoints[i] ? 1 : 0;
    const float grid4[4] = {Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv};
    std::vector<int32_t> cur(n > 0 ? n : 1, -1);
    int nmatches = 0;
    check(orb_match_projection(handle(), 0, q.data(), qd.data(), qa.data(), nq, kps(CurrentFrame.mvKeysUn), rows32(CurrentFrame.mDescriptors),
                               nullptr, occ.data(), n, grid4, 0.f, ORBdist, mbCheckOrientation ? 1 : 0, cur.data(), &nmatches),
          "orb_match_projection");
    for (int i = 0; i < n; i++) {
        if (cur[i] >= 0) CurrentFrame.mvpMapPoints[i] = vpMPs[cur[i]];          // :392
        else if (cur[i] == -2) CurrentFrame.mvpMapPoints[i] = NULL;             // :430
    }
    return nmatches;
}

namespace
{
// the gates shared by :443-550, :1364-1480 and :1483-1633: project pMP with (Rcw, tcw), keep it if it lies in front of the
// camera, inside the image, within its scale-invariance distances and is seen from less than 60 degrees off its normal
bool projectIntoKeyFrame(KeyFrame* pKF, MapPoint* pMP, const cv::Mat& Rcw, const cv::Mat& tcw, const cv::Mat& Ow, float th,
                         orb_proj_query& q, float* invzOut)
{
    const float& fx = pKF->fx;
    const float& fy = pKF->fy;
    const float& cx = pKF->cx;
    const float& cy = pKF->cy;
    cv::Mat p3Dw = pMP->GetWorldPos();
    cv::Mat p3Dc = Rcw * p3Dw + tcw;
    if (p3Dc.at<float>(2) < 0.0f) return false;
    const float invz = 1 / p3Dc.at<float>(2);
    const float x = p3Dc.at<float>(0) * invz;
    const float y = p3Dc.at<float>(1) * invz;
    const float u = fx * x + cx;
    const float v = fy * y + cy;
    if (!pKF->IsInImage(u, v)) return false;
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    cv::Mat PO = p3Dw - Ow;
    const float dist3D = cv::norm(PO);
    if (dist3D < minDistance || dist3D > maxDistance) return false;
    cv::Mat Pn = pMP->GetNormal();
    if (PO.dot(Pn) < 0.5 * dist3D) return false;
    int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
    const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
    q.x = u; q.y = v; q.r = radius;
    q.min_level = nPredictedLevel - 1;
    q.max_level = nPredictedLevel;
    q.flags = 1 | 2;
    if (invzOut) *invzOut = invz;
    return true;
}
void keyFrameGrid(KeyFrame* pKF, float grid4[4])
{
    grid4[0] = (float)pKF->mnMinX; grid4[1] = (float)pKF->mnMinY;
    grid4[2] = pKF->mfGridElementWidthInv; grid4[3] = pKF->mfGridElementHeightInv;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ :443-550
int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched,
                                   int th)
{
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);                           // :452-456
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
    const int nq = (int)vpPoints.size();
    std::vector<orb_proj_query> q(nq > 0 ? nq : 1, deadQuery());
    std::vector<unsigned char> qd((size_t)(nq > 0 ? nq : 1) * 32, 0);
    std::vector<float> qa(nq > 0 ? nq : 1, 0.f);
    for (int iMP = 0; iMP < nq; iMP++) {
        MapPoint* pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;                // :468-470
        if (!projectIntoKeyFrame(pKF, pMP, Rcw, tcw, Ow, (float)th, q[iMP], nullptr)) { q[iMP] = deadQuery(); continue; }
        const cv::Mat d = pMP->GetDescriptor();
        std::memcpy(&qd[(size_t)iMP * 32], d.ptr<unsigned char>(0), 32);
    }
    hipshim::g_lastQueries = q;
    const int n = pKF->N;
    std::vector<unsigned char> occ(n > 0 ? n : 1, 0);
    for (int i = 0; i < n; i++) occ[i] = vpMatched[i] ? 1 : 0;                   // :517-518
    float grid4[4];
    keyFrameGrid(pKF, grid4);
    std::vector<int32_t> cur(n > 0 ? n : 1, -1);
    int nmatches = 0;
    check(orb_match_projection(handle(), 0, q.data(), qd.data(), qa.data(), nq, kps(pKF->mvKeysUn), rows32(pKF->mDescriptors), nullptr,
                               occ.data(), n, grid4, 0.f, TH_LOW, 0, cur.data(), &nmatches), "orb_match_projection");
    for (int i = 0; i < n; i++)
        if (cur[i] >= 0) vpMatched[i] = vpPoints[cur[i]];                        // :537
    return nmatches;
}

// ------------------------------------------------------------------------------------------------ :1364-1480
int ORBmatcher::Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th)
{
    cv::Mat Rcw = pKF->GetRotation();
    cv::Mat tcw = pKF->GetTranslation();
    const float& bf = pKF->mbf;
    cv::Mat Ow = pKF->GetCameraCenter();
    const int nMPs = (int)vpMapPoints.size();
    std::vector<orb_proj_query> q(nMPs > 0 ? nMPs : 1, deadQuery());
    std::vector<unsigned char> qd((size_t)(nMPs > 0 ? nMPs : 1) * 32, 0);
    for (int i = 0; i < nMPs; i++) {
        MapPoint* pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;                    // :1386-1387
        float invz = 0.f;
        if (!projectIntoKeyFrame(pKF, pMP, Rcw, tcw, Ow, th, q[i], &invz)) { q[i] = deadQuery(); continue; }
        q[i].ur = q[i].x - bf * invz;                                            // :1408
        const cv::Mat d = pMP->GetDescriptor();
        std::memcpy(&qd[(size_t)i * 32], d.ptr<unsigned char>(0), 32);
    }
    hipshim::g_lastQueries = q;
    float grid4[4];
    keyFrameGrid(pKF, grid4);
    std::vector<int32_t> best(nMPs > 0 ? nMPs : 1, -1), dist(nMPs > 0 ? nMPs : 1, 256);
    check(orb_match_projection_best(handle(), q.data(), qd.data(), nMPs, kps(pKF->mvKeysUn), rows32(pKF->mDescriptors),
                                    pKF->mvuRight.empty() ? nullptr : pKF->mvuRight.data(), pKF->N, grid4, TH_LOW, 1,
                                    pKF->mvInvLevelSigma2.data(), (int)pKF->mvInvLevelSigma2.size(), best.data(), dist.data()),
          "orb_match_projection_best");
    // the map surgery, in the reference's order (:1456-1476).  It does not feed back into the SEARCH (a MapPoint's position and
    // descriptor do not change), but it does feed back into the gates of :1386-1387, which the reference evaluates per
    // iteration: a pointer listed twice is in the keyframe (or replaced, i.e. bad) by its second turn, and Replace() can turn
    // a later entry bad -- so the gates are evaluated again here, at the point of the reference's loop.
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {
        if (best[i] < 0) continue;
        MapPoint* pMP = vpMapPoints[i];
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        MapPoint* pMPinKF = pKF->GetMapPoint(best[i]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                else pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, best[i]);
            pKF->AddMapPoint(pMP, best[i]);
        }
        nFused++;
    }
    return nFused;
}

// ------------------------------------------------------------------------------------------------ :1483-1633
int ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint)
{
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);                           // :1494-1498
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    std::set<MapPoint*> spAlreadyFound;                                        // pKF->GetMapPoints() (:1501): its good MapPoints
    {
        const std::vector<MapPoint*> v = pKF->GetMapPointMatches();
        for (size_t i = 0; i < v.size(); i++)
            if (v[i] && !v[i]->isBad()) spAlreadyFound.insert(v[i]);
    }
    const int nPoints = (int)vpPoints.size();
    std::vector<orb_proj_query> q(nPoints > 0 ? nPoints : 1, deadQuery());
    std::vector<unsigned char> qd((size_t)(nPoints > 0 ? nPoints : 1) * 32, 0);
    for (int iMP = 0; iMP < nPoints; iMP++) {
        MapPoint* pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;                // :1512-1513
        if (!projectIntoKeyFrame(pKF, pMP, Rcw, tcw, Ow, th, q[iMP], nullptr)) { q[iMP] = deadQuery(); continue; }
        const cv::Mat d = pMP->GetDescriptor();
        std::memcpy(&qd[(size_t)iMP * 32], d.ptr<unsigned char>(0), 32);
    }
    hipshim::g_lastQueries = q;
    float grid4[4];
    keyFrameGrid(pKF, grid4);
    std::vector<int32_t> best(nPoints > 0 ? nPoints : 1, -1), dist(nPoints > 0 ? nPoints : 1, 256);
    check(orb_match_projection_best(handle(), q.data(), qd.data(), nPoints, kps(pKF->mvKeysUn), rows32(pKF->mDescriptors), nullptr, pKF->N,
                                    grid4, TH_LOW, 0, nullptr, 0, best.data(), dist.data()), "orb_match_projection_best");
    int nFused = 0;
    for (int iMP = 0; iMP < nPoints; iMP++) {                                    // :1608-1627
        if (best[iMP] < 0) continue;
        MapPoint* pMP = vpPoints[iMP];
        MapPoint* pMPinKF = pKF->GetMapPoint(best[iMP]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, best[iMP]);
            pKF->AddMapPoint(pMP, best[iMP]);
        }
        nFused++;
    }
    return nFused;
}

// ------------------------------------------------------------------------------------------------ :835-1025
int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                             const cv::Mat& t12, const float th)
{
    const float& fx = pKF1->fx;
    const float& fy = pKF1->fy;
    const float& cx = pKF1->cx;
    const float& cy = pKF1->cy;
    cv::Mat R1w = pKF1->GetRotation();                                         // :844-852
    cv::Mat t1w = pKF1->GetTranslation();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat sR12 = s12 * R12;
    cv::Mat sR21 = (1.0 / s12) * R12.t();
    cv::Mat t21 = -sR21 * t12;
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size();
    const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false);
    std::vector<bool> vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {                                             // :862-873
        MapPoint* pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    // one direction: MapPoints of KFa (camera pose Raw, taw) through (sRba, tba) into KFb, best feature of KFb within TH_HIGH
    auto direction = [&](const std::vector<MapPoint*>& vpA, const std::vector<bool>& already, const cv::Mat& Raw, const cv::Mat& taw,
                         const cv::Mat& sRba, const cv::Mat& tba, KeyFrame* pKFb, std::vector<int>& vnMatch) {
        const int NA = (int)vpA.size();
        std::vector<orb_proj_query> q(NA > 0 ? NA : 1, deadQuery());
        std::vector<unsigned char> qd((size_t)(NA > 0 ? NA : 1) * 32, 0);
        for (int i1 = 0; i1 < NA; i1++) {
            MapPoint* pMP = vpA[i1];
            if (!pMP || already[i1]) continue;                                  // :882-886
            if (pMP->isBad()) continue;
            cv::Mat p3Dw = pMP->GetWorldPos();                                  // :888-916
            cv::Mat p3Dc1 = Raw * p3Dw + taw;
            cv::Mat p3Dc2 = sRba * p3Dc1 + tba;
            if (p3Dc2.at<float>(2) < 0.0) continue;
            const float invz = 1.0 / p3Dc2.at<float>(2);
            const float x = p3Dc2.at<float>(0) * invz;
            const float y = p3Dc2.at<float>(1) * invz;
            const float u = fx * x + cx;
            const float v = fy * y + cy;
            if (!pKFb->IsInImage(u, v)) continue;
            const float maxDistance = pMP->GetMaxDistanceInvariance();
            const float minDistance = pMP->GetMinDistanceInvariance();
            const float dist3D = cv::norm(p3Dc2);
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            const int nPredictedLevel = pMP->PredictScale(dist3D, pKFb);
            const float radius = th * pKFb->mvScaleFactors[nPredictedLevel];
            q[i1].x = u; q[i1].y = v; q[i1].r = radius;
            q[i1].min_level = nPredictedLevel - 1;                               // :930-931
            q[i1].max_level = nPredictedLevel;
            q[i1].flags = 1 | 2;
            const cv::Mat d = pMP->GetDescriptor();
            std::memcpy(&qd[(size_t)i1 * 32], d.ptr<unsigned char>(0), 32);
        }
        hipshim::g_lastQueries = q;
        float grid4[4];
        keyFrameGrid(pKFb, grid4);
        std::vector<int32_t> best(NA > 0 ? NA : 1, -1);
        check(orb_match_projection_best(handle(), q.data(), qd.data(), NA, kps(pKFb->mvKeysUn), rows32(pKFb->mDescriptors), nullptr, pKFb->N,
                                        grid4, TH_HIGH, 0, nullptr, 0, best.data(), nullptr), "orb_match_projection_best");
        vnMatch.assign(NA, -1);
        for (int i = 0; i < NA; i++) vnMatch[i] = best[i];
    };
    std::vector<int> vnMatch1, vnMatch2;
    direction(vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, pKF2, vnMatch1);
    direction(vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12, pKF1, vnMatch2);
    int nFound = 0;                                                            // :1006-1022
    for (int i1 = 0; i1 < N1; i1++) {
        int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            int idx1 = vnMatch2[idx2];
            if (idx1 == i1) {
                vpMatches12[i1] = vpMapPoints2[idx2];
                nFound++;
            }
        }
    }
    return nFound;
}

// ------------------------------------------------------------------------------------------------ :1183-1359
int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                                       const bool bOnlyStereo)
{
    cv::Mat Cw = pKF1->GetCameraCenter();                                       // :1190-1197
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat C2 = R2w * Cw + t2w;
    const float invz = 1.0f / C2.at<float>(2);
    const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
    const float ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
    hipshim::g_lastEx = ex; hipshim::g_lastEy = ey;
    const int n1 = pKF1->N, n2 = pKF2->N;
    std::vector<unsigned char> mp1(n1 > 0 ? n1 : 1, 0), mp2(n2 > 0 ? n2 : 1, 0);
    for (int i = 0; i < n1; i++) mp1[i] = pKF1->GetMapPoint(i) ? 1 : 0;           // :1226-1229
    for (int i = 0; i < n2; i++) mp2[i] = pKF2->GetMapPoint(i) ? 1 : 0;           // :1247-1250
    float F[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) F[3 * r + c] = F12.at<float>(r, c);
    FlatFeatVec fv1(pKF1->mFeatVec), fv2(pKF2->mFeatVec);
    std::vector<int32_t> m12(n1 > 0 ? n1 : 1, -1);
    int nmatches = 0;
    check(orb_match_triangulation(handle(), kps(pKF1->mvKeysUn), rows32(pKF1->mDescriptors), mp1.data(),
                                  pKF1->mvuRight.empty() ? nullptr : pKF1->mvuRight.data(), n1, &fv1.view, kps(pKF2->mvKeysUn),
                                  rows32(pKF2->mDescriptors), mp2.data(), pKF2->mvuRight.empty() ? nullptr : pKF2->mvuRight.data(), n2,
                                  &fv2.view, F, ex, ey, pKF2->mvScaleFactors.data(), pKF2->mvLevelSigma2.data(),
                                  (int)pKF2->mvScaleFactors.size(), bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0, m12.data(), &nmatches),
          "orb_match_triangulation");
    vMatchedPairs.clear();                                                     // :1345-1352
    vMatchedPairs.reserve(nmatches);
    for (int i = 0; i < n1; i++) {
        if (m12[i] < 0) continue;
        vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
    }
    return nmatches;
}

}  // namespace ORB_SLAM2
