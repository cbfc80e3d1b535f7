// orbref -- CPU ORACLE for the ORB front-end hot path.  TEST INFRASTRUCTURE ONLY.
//
// A dependency-free C++17 restatement of the reference's ORBextractor / ORBmatcher
// algorithms (Hello-Water/ORB-SLAM2-ChineseNotes, src/ORBextractor.cc, src/ORBmatcher.cc,
// src/Frame.cc) and of the OpenCV / DBoW2 primitive semantics those files call
// (SURVEY.md Appendix A/B).  It exists to CHECK the HIP path; it is never the thing that is
// shipped or measured.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may build, link, import or call anything in this directory.
//
// PARITY UNPINNED at the OpenCV/DBoW2 boundary: the reference ships no tests, golden
// vectors or fixtures, and it cannot be compiled here (OpenCV + DBoW2 absent, no network),
// so this restatement -- pinned only by the source-derivable known answers of SURVEY.md
// Appendix C (tests/test_oracle_kat.py) -- IS the definition the GPU path is compared with.
//
// Build flags that are part of the definition: -O2 -ffp-contract=off -fno-fast-math.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace orbref {

// Same 28-byte layout as cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id).
struct KeyPoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px;  // row-major, pitch == w
    uint8_t at(int y, int x) const { return px[(size_t)y * w + x]; }
};

// One FAST candidate of a level, in coordinates relative to (16,16) as the reference's
// vToDistributeKeys holds them (src/ORBextractor.cc:866-871).
struct Candidate {
    int x, y, response;
};

struct ResizeTab {  // per-axis fixed-point bilinear coefficients
    std::vector<int> ofs;
    std::vector<short> c0, c1;
};

class Extractor {
public:
    // reference ctor: src/ORBextractor.cc:498-559
    Extractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    // the integer taps cv::GaussianBlur(7x7, sigma 2) uses for CV_8U: OpenCV-version dependent, see orb_gaussian_preset
    int gaussTaps[4] = {18, 34, 49, 55};

    // reference operator(): src/ORBextractor.cc:1084-1150
    void extract(const uint8_t* img, int rows, int cols, size_t stride,
                 std::vector<KeyPoint>& kps, std::vector<uint8_t>& desc);

    // ---- stages, public so that differential tests can compare stage by stage ----
    void computePyramid(const uint8_t* img, int rows, int cols, size_t stride);   // :1153-1180
    std::vector<Candidate> cellCandidates(int level) const;                       // :795-875
    std::vector<Candidate> distribute(const std::vector<Candidate>& cands,        // :562-792
                                      int minX, int maxX, int minY, int maxY, int N) const;
    float icAngle(int level, int x, int y) const;                                 // :78-105
    void descriptor(const Image& blurred, int x, int y, float angleDeg, uint8_t out[32]) const;  // :120-161

    int nfeatures, nlevels, iniThFAST, minThFAST;
    double scaleFactor;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<Image> pyramid;               // mvImagePyramid interiors (no 19-px border)
    std::vector<int> levelCounts;             // keypoints per level of the last extract()
    std::vector<int> levelCandidates;         // FAST candidates per level of the last extract()
};

// ---- primitives ----
int cvRoundF(float v);                                   // round-half-even
int cvRoundD(double v);
ResizeTab resizeTab(int srcLen, int dstLen);             // A.2
void resizeLinear(const Image& src, Image& dst, int dw, int dh);   // cv::resize INTER_LINEAR 8UC1
int fastScoreV(const uint8_t* p, int pitch);             // A.4: V(p) (may be <= 0)
// A.7; taps4 = {k0, k1, k2, k3} of the symmetric 8.8 fixed-point kernel k0 k1 k2 k3 k2 k1 k0 (nullptr: {18, 34, 49, 55})
void gaussianBlur7(const Image& src, Image& dst, const int* taps4 = nullptr);
float fastAtan2(float y, float x);                       // A.5

// ---- matcher ----
int hamming256(const uint8_t* a, const uint8_t* b);      // src/ORBmatcher.cc:46-63
void threeMaxima(const int counts[30], int& i1, int& i2, int& i3);   // :1663-1707

// DBoW2::FeatureVector flattened to CSR: ascending node ids; indices ascending inside a node.
struct FeatVec {
    std::vector<uint32_t> nodeIds;
    std::vector<int32_t> offsets;   // size nodeIds.size()+1
    std::vector<int32_t> indices;
};

// SearchByBoW(KeyFrame*, Frame&, ...)  src/ORBmatcher.cc:552-687.
// outF[iF] = index of the matched KF feature or -1.  Returns nmatches.
int searchByBoW(const uint8_t* descKF, const float* angleKF, const uint8_t* validKF, const FeatVec& fvKF,
                const uint8_t* descF, const float* angleF, int nF, const FeatVec& fvF,
                float nnRatio, bool checkOri, std::vector<int32_t>& outF);

// SearchByBoW(KeyFrame*, KeyFrame*, ...)  src/ORBmatcher.cc:690-832.
// out12[i1] = index of matched KF2 feature or -1.
int searchByBoWKK(const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1, const FeatVec& fv1,
                  const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2, const FeatVec& fv2,
                  float nnRatio, bool checkOri, std::vector<int32_t>& out12);

// Frame grid (src/Frame.cc:243-259, 348-422; include/Frame.h:37-38)
struct FrameGrid {
    float minX, minY, invW, invH;
    std::vector<std::vector<int32_t>> cells;   // [ix*48+iy], insertion order
    void assign(const KeyPoint* kps, int n);
    std::vector<int32_t> inArea(const KeyPoint* kps, float x, float y, float r, int minLevel, int maxLevel) const;
};

// SearchForInitialization  src/ORBmatcher.cc:1055-1180.  prevMatched is in/out (x,y pairs).
int searchForInitialization(const KeyPoint* kps1, const uint8_t* desc1, int n1,
                            const KeyPoint* kps2, const uint8_t* desc2, int n2,
                            const FrameGrid& grid2, float* prevMatchedXY, int windowSize,
                            float nnRatio, bool checkOri, std::vector<int32_t>& matches12);

// One projected MapPoint of the tracking matchers (what the reference computes with cv::Mat before the search).
struct ProjQuery {
    float x, y, r;              // projection and window radius handed to GetFeaturesInArea
    int32_t minLevel, maxLevel; // level range handed to GetFeaturesInArea (-1 = open)
    float ur, erMax;            // stereo check: skip if uRight>0 && |ur - uRight| > erMax
    int32_t flags;              // bit0: query is live; bit1: its MapPoint has Observations() > 0
};
int searchByProjectionMap(const ProjQuery* q, const uint8_t* qDesc, int nq, const KeyPoint* kps, const uint8_t* desc,
                          const float* uRight, const uint8_t* occupiedIn, int n, const FrameGrid& grid, float ratio,
                          int maxDist, std::vector<int32_t>& matchCur);
int searchByProjectionLast(const ProjQuery* q, const uint8_t* qDesc, const float* qAngle, int nq, const KeyPoint* kps,
                           const uint8_t* desc, const float* uRight, const uint8_t* occupiedIn, int n,
                           const FrameGrid& grid, int maxDist, bool checkOri, std::vector<int32_t>& matchCur);

void searchByProjectionBest(const ProjQuery* q, const uint8_t* qDesc, int nq, const KeyPoint* kps, const uint8_t* desc,
                            const float* uRight, int n, const FrameGrid& grid, int maxDist, bool chi2,
                            const float* invSigma2, int32_t* bestIdx, int32_t* bestDist);

// SearchForTriangulation  src/ORBmatcher.cc:1183-1359 (+ CheckDistEpipolarLine :1636-1650).  F12 row-major 3x3.
int searchForTriangulation(const KeyPoint* k1, const uint8_t* d1, const uint8_t* hasMP1, const float* uR1, int n1,
                           const FeatVec& fv1, const KeyPoint* k2, const uint8_t* d2, const uint8_t* hasMP2,
                           const float* uR2, int n2, const FeatVec& fv2, const float* F12, float ex, float ey,
                           const float* scaleFactors2, const float* levelSigma2_2, bool onlyStereo, bool checkOri,
                           std::vector<int32_t>& matches12);

// A DBoW2 vocabulary tree flattened to arrays (node 0 = root; children of node v = children[childBegin[v] ..
// childBegin[v+1]) in stored order; wordId[v] >= 0 for leaves).
struct VocabTree {
    const uint8_t* nodeDesc;      // [nNodes][32]
    const int32_t* childBegin;    // [nNodes + 1]
    const int32_t* children;
    const int32_t* wordId;        // [nNodes]
    int nNodes, L;
};
void vocabTransform(const VocabTree& t, const uint8_t* desc, int n, int levelsup, int32_t* wordOf, int32_t* nodeOf);

// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:275-342) for a batch of MapPoints: descriptors of point p
// are rows offsets[p]..offsets[p+1] of desc; bestIdx[p] = index inside that list (-1 for an empty list).
void distinctiveDescriptors(const uint8_t* desc, const int32_t* offsets, int nPoints, int32_t* bestIdx);

// Synthetic stand-in for the (absent) ORB vocabulary: 2-level k=10 tree of 256-bit centroids
// (SURVEY §8d).  nodeId = 11 + 10*c1 + c2.  centroids: 10 level-1 then 100 level-2, 32 B each.
FeatVec bowTransform(const uint8_t* desc, int n, const uint8_t* centroids /*110 x 32*/);

}  // namespace orbref
