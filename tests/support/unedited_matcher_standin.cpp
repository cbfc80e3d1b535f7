// TEST-ONLY stand-in for "the reference's src/ORBmatcher.cc compiled UNCHANGED": it defines the four member functions
// that orb-slam2-chinesenotes_amd/host/ORBmatcherHip.cc replaces (with marker bodies) plus one that is NOT replaced.
// tests/test_shim.py weakens the four symbols in this object (tools/weaken_matcher_symbols.sh) and links it beside
// ORBmatcherHip.o: the HIP-backed definitions must win, the other function must stay -- the binding recipe of
// INTEGRATION.md section 2 that needs no edit of the reference's source file.
#include "ORBmatcher.h"

namespace ORB_SLAM2 {
// (in the reference these live in src/Frame.cc)
float Frame::mnMinX = 0, Frame::mnMinY = 0, Frame::mfGridElementWidthInv = 0.1f, Frame::mfGridElementHeightInv = 0.1f;
int ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&) { return -12345; }
int ORBmatcher::SearchByBoW(KeyFrame*, Frame&, std::vector<MapPoint*>&) { return -12345; }
int ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, std::vector<MapPoint*>&) { return -12345; }
int ORBmatcher::SearchForInitialization(Frame&, Frame&, std::vector<cv::Point2f>&, std::vector<int>&, int) { return -12345; }
}  // namespace ORB_SLAM2

extern "C" int standin_not_replaced() { return 777; }          // plays the role of the 8 matcher methods that keep their CPU bodies
