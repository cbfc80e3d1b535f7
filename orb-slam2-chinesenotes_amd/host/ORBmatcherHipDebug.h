// ORBmatcherHipDebug.h -- diagnostics of the optional matcher bindings (ORBmatcherHipExtra.cc): the orb_proj_query array
// the calling thread's last projection-based routine handed to the GPU (one per projected MapPoint; flags == 0 for points
// the host-side gates rejected).  Tests feed it to the CPU oracle; a maintainer can diff it against the reference's own
// GetFeaturesInArea arguments.
#pragma once
#include <vector>

#include "orb_hip.h"

namespace ORB_SLAM2
{
namespace hipshim
{
const std::vector<orb_proj_query>& LastProjectionQueries();
// the epipole (ex, ey) the calling thread's last SearchForTriangulation computed (reference src/ORBmatcher.cc:1193-1197)
void LastEpipole(float* ex, float* ey);
}
}  // namespace ORB_SLAM2
