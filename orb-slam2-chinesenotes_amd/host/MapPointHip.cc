// MapPointHip.cc -- replacement body for void MapPoint::ComputeDistinctiveDescriptors(), reference src/MapPoint.cc:275-342,
// compiled against the reference's own include/MapPoint.h and KeyFrame.h: the observed descriptors are gathered on the host
// exactly as :283-304 does (under mMutexFeatures, bad keyframes skipped), the N x N Hamming distances and the
// least-median choice (:310-335) run on the GPU (orb_distinctive_descriptors), the winner is cloned into mDescriptor
// under the lock (:337-340).
// hipshim::ComputeDistinctiveDescriptors(points) does the same for MANY MapPoints with ONE device call -- what the loops of
// LocalMapping::ProcessNewKeyFrame / SearchInNeighbors (src/LocalMapping.cc) and the map updates after a loop closure
// amount to; results per point are identical to the member function's.
// Build: wrap the definition in src/MapPoint.cc in `#ifndef ORB_HIP_MAPPOINT` and add this file.  No CPU fallback.
#include <map>
#include <mutex>
#include <vector>

#include "MapPoint.h"
#include "KeyFrame.h"
#include "BowHip.h"

namespace ORB_SLAM2
{
namespace hipshim
{
void ComputeDistinctiveDescriptors(const std::vector<MapPoint*>& points);
}

namespace
{
// :283-304 for one point: rows of the observing keyframes' descriptor matrices (empty: the function returns early)
std::vector<cv::Mat> observedDescriptors(MapPoint* pMP, std::mutex& mu, bool& bad, std::map<KeyFrame*, size_t>& obsMember)
{
    std::vector<cv::Mat> vDescriptors;
    std::map<KeyFrame*, size_t> observations;
    {
        std::unique_lock<std::mutex> lock1(mu);
        if (bad) return vDescriptors;
        observations = obsMember;
    }
    (void)pMP;
    vDescriptors.reserve(observations.size());
    for (std::map<KeyFrame*, size_t>::iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
        KeyFrame* pKF = mit->first;
        if (!pKF->isBad()) vDescriptors.push_back(pKF->mDescriptors.row((int)mit->second));
    }
    return vDescriptors;
}
}  // namespace

void MapPoint::ComputeDistinctiveDescriptors()
{
    std::vector<cv::Mat> vDescriptors = observedDescriptors(this, mMutexFeatures, mbBad, mObservations);
    if (vDescriptors.empty()) return;
    const size_t N = vDescriptors.size();
    int32_t best = 0;
    std::vector<unsigned char> rows(N * 32);
    for (size_t i = 0; i < N; i++) std::memcpy(&rows[i * 32], vDescriptors[i].ptr<unsigned char>(0), 32);
    const int32_t offsets[2] = {0, (int32_t)N};
    hipbow::check(orb_distinctive_descriptors(hipbow::matcher(), rows.data(), offsets, 1, &best), "orb_distinctive_descriptors");
    {
        std::unique_lock<std::mutex> lock(mMutexFeatures);
        mDescriptor = vDescriptors[best].clone();
    }
}

namespace hipshim
{
class MapPointAccess : public MapPoint {          // (the members are protected in include/MapPoint.h)
public:
    std::vector<cv::Mat> gather() { return observedDescriptors(this, mMutexFeatures, mbBad, mObservations); }
    void set(const cv::Mat& d)
    {
        std::unique_lock<std::mutex> lock(mMutexFeatures);
        mDescriptor = d.clone();
    }
};

void ComputeDistinctiveDescriptors(const std::vector<MapPoint*>& points)
{
    std::vector<std::vector<cv::Mat> > all(points.size());
    std::vector<int32_t> offsets(1, 0);
    std::vector<unsigned char> rows;
    for (size_t p = 0; p < points.size(); p++) {
        if (points[p]) all[p] = static_cast<MapPointAccess*>(points[p])->gather();
        for (size_t i = 0; i < all[p].size(); i++) {
            rows.resize(rows.size() + 32);
            std::memcpy(&rows[rows.size() - 32], all[p][i].ptr<unsigned char>(0), 32);
        }
        offsets.push_back((int32_t)(rows.size() / 32));
    }
    if (rows.empty()) return;
    std::vector<int32_t> best(points.size(), -1);
    hipbow::check(orb_distinctive_descriptors(hipbow::matcher(), rows.data(), offsets.data(), (int)points.size(), best.data()),
                  "orb_distinctive_descriptors");
    for (size_t p = 0; p < points.size(); p++)
        if (!all[p].empty()) static_cast<MapPointAccess*>(points[p])->set(all[p][best[p]]);
}
}  // namespace hipshim

}  // namespace ORB_SLAM2
