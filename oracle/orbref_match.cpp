// orbref_match.cpp -- CPU ORACLE (test infrastructure only; see orbref.hpp header note).
// Restates reference src/ORBmatcher.cc:37-63,552-832,1055-1180,1663-1707 and the Frame grid
// of src/Frame.cc:243-259,348-422 per SURVEY.md Appendix B.  PARITY UNPINNED (no reference
// fixtures exist); DBoW2's FeatureVector is replaced by a CSR view with the same iteration order.
#include "orbref.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>

namespace orbref {

static const int TH_LOW = 50;          // src/ORBmatcher.cc:38
static const int HISTO_LENGTH = 30;    // :39

int hamming256(const uint8_t* a, const uint8_t* b)
{
    // :46-63 -- eight 32-bit SWAR popcounts of the XOR (little-endian loads of the 32-byte rows)
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        std::memcpy(&pa, a + 4 * i, 4);
        std::memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

void threeMaxima(const int counts[30], int& ind1, int& ind2, int& ind3)
{
    // :1663-1707.  ind1..3 keep the caller's initial -1 unless assigned.
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < HISTO_LENGTH; i++) {
        const int s = counts[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s; ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

static int rotBin(float angA, float angB)
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = angA - angB;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);     // std::round(float): half away from zero
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// position of the first node id >= want, starting the search anywhere (std::map::lower_bound)
static size_t lowerBound(const std::vector<uint32_t>& ids, uint32_t want)
{
    return (size_t)(std::lower_bound(ids.begin(), ids.end(), want) - ids.begin());
}

int searchByBoW(const uint8_t* descKF, const float* angleKF, const uint8_t* validKF, const FeatVec& fvKF,
                const uint8_t* descF, const float* angleF, int nF, const FeatVec& fvF,
                float nnRatio, bool checkOri, std::vector<int32_t>& outF)
{
    outF.assign(nF, -1);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    size_t a = 0, b = 0;
    while (a < fvKF.nodeIds.size() && b < fvF.nodeIds.size()) {
        if (fvKF.nodeIds[a] == fvF.nodeIds[b]) {
            for (int p = fvKF.offsets[a]; p < fvKF.offsets[a + 1]; p++) {
                const int iKF = fvKF.indices[p];
                if (!validKF[iKF]) continue;               // no MapPoint / isBad()  (:592-595)
                int best1 = 256, best2 = 256, bestIdx = -1;
                for (int q = fvF.offsets[b]; q < fvF.offsets[b + 1]; q++) {
                    const int iF = fvF.indices[q];
                    if (outF[iF] >= 0) continue;            // already matched (:607)
                    const int d = hamming256(descKF + 32 * (size_t)iKF, descF + 32 * (size_t)iF);
                    if (d < best1) { best2 = best1; best1 = d; bestIdx = iF; }
                    else if (d < best2) best2 = d;
                }
                if (best1 <= TH_LOW && (float)best1 < nnRatio * (float)best2) {
                    outF[bestIdx] = iKF;
                    if (checkOri) rotHist[rotBin(angleKF[iKF], angleF[bestIdx])].push_back(bestIdx);
                    nmatches++;
                }
            }
            a++; b++;
        } else if (fvKF.nodeIds[a] < fvF.nodeIds[b]) {
            a = lowerBound(fvKF.nodeIds, fvF.nodeIds[b]);
        } else {
            b = lowerBound(fvF.nodeIds, fvKF.nodeIds[a]);
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        int i1 = -1, i2 = -1, i3 = -1;
        threeMaxima(counts, i1, i2, i3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int idx : rotHist[i]) { outF[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

int searchByBoWKK(const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1, const FeatVec& fv1,
                  const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2, const FeatVec& fv2,
                  float nnRatio, bool checkOri, std::vector<int32_t>& out12)
{
    out12.assign(n1, -1);
    std::vector<uint8_t> matched2(n2, 0);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    size_t a = 0, b = 0;
    while (a < fv1.nodeIds.size() && b < fv2.nodeIds.size()) {
        if (fv1.nodeIds[a] == fv2.nodeIds[b]) {
            for (int p = fv1.offsets[a]; p < fv1.offsets[a + 1]; p++) {
                const int i1 = fv1.indices[p];
                if (!valid1[i1]) continue;
                int best1 = 256, best2 = 256, bestIdx = -1;
                for (int q = fv2.offsets[b]; q < fv2.offsets[b + 1]; q++) {
                    const int i2 = fv2.indices[q];
                    if (matched2[i2] || !valid2[i2]) continue;    // :750-754
                    const int d = hamming256(desc1 + 32 * (size_t)i1, desc2 + 32 * (size_t)i2);
                    if (d < best1) { best2 = best1; best1 = d; bestIdx = i2; }
                    else if (d < best2) best2 = d;
                }
                if (best1 < TH_LOW && (float)best1 < nnRatio * (float)best2) {   // strict (:772)
                    out12[i1] = bestIdx;
                    matched2[bestIdx] = 1;
                    if (checkOri) rotHist[rotBin(angle1[i1], angle2[bestIdx])].push_back(i1);
                    nmatches++;
                }
            }
            a++; b++;
        } else if (fv1.nodeIds[a] < fv2.nodeIds[b]) {
            a = lowerBound(fv1.nodeIds, fv2.nodeIds[b]);
        } else {
            b = lowerBound(fv2.nodeIds, fv1.nodeIds[a]);
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        int i1 = -1, i2 = -1, i3 = -1;
        threeMaxima(counts, i1, i2, i3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int idx : rotHist[i]) { out12[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// ------------------------------------------------------------------ Frame grid
static const int GRID_COLS = 64, GRID_ROWS = 48;   // include/Frame.h:37-38

void FrameGrid::assign(const KeyPoint* kps, int n)
{
    cells.assign((size_t)GRID_COLS * GRID_ROWS, {});
    for (int i = 0; i < n; i++) {
        int px = (int)std::round((kps[i].x - minX) * invW);      // src/Frame.cc:414-415
        int py = (int)std::round((kps[i].y - minY) * invH);
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
        cells[(size_t)px * GRID_ROWS + py].push_back(i);
    }
}

std::vector<int32_t> FrameGrid::inArea(const KeyPoint* kps, float x, float y, float r,
                                       int minLevel, int maxLevel) const
{
    // src/Frame.cc:348-409
    std::vector<int32_t> out;
    const int nMinCellX = std::max(0, (int)std::floor((x - minX - r) * invW));
    if (nMinCellX >= GRID_COLS) return out;
    const int nMaxCellX = std::min(GRID_COLS - 1, (int)std::ceil((x - minX + r) * invW));
    if (nMaxCellX < 0) return out;
    const int nMinCellY = std::max(0, (int)std::floor((y - minY - r) * invH));
    if (nMinCellY >= GRID_ROWS) return out;
    const int nMaxCellY = std::min(GRID_ROWS - 1, (int)std::ceil((y - minY + r) * invH));
    if (nMaxCellY < 0) return out;
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
            for (int32_t j : cells[(size_t)ix * GRID_ROWS + iy]) {
                const KeyPoint& kp = kps[j];
                if (checkLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                const float dx = kp.x - x, dy = kp.y - y;
                if (std::fabs(dx) < r && std::fabs(dy) < r) out.push_back(j);
            }
    return out;
}

int searchForInitialization(const KeyPoint* kps1, const uint8_t* desc1, int n1,
                            const KeyPoint* kps2, const uint8_t* desc2, int n2,
                            const FrameGrid& grid2, float* prevXY, int windowSize,
                            float nnRatio, bool checkOri, std::vector<int32_t>& m12)
{
    // src/ORBmatcher.cc:1055-1180
    int nmatches = 0;
    m12.assign(n1, -1);
    std::vector<int> rotHist[HISTO_LENGTH];
    std::vector<int> matchedDist(n2, INT_MAX), m21(n2, -1);
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = kps1[i1].octave;
        if (level1 > 0) continue;
        std::vector<int32_t> cand = grid2.inArea(kps2, prevXY[2 * i1], prevXY[2 * i1 + 1],
                                                 (float)windowSize, level1, level1);
        if (cand.empty()) continue;
        int best = INT_MAX, best2 = INT_MAX, bestIdx = -1;
        for (int32_t i2 : cand) {
            const int d = hamming256(desc1 + 32 * (size_t)i1, desc2 + 32 * (size_t)i2);
            if (matchedDist[i2] <= d) continue;
            if (d < best) { best2 = best; best = d; bestIdx = i2; }
            else if (d < best2) best2 = d;
        }
        if (best <= TH_LOW) {
            if ((float)best < (float)best2 * nnRatio) {
                if (m21[bestIdx] >= 0) { m12[m21[bestIdx]] = -1; nmatches--; }
                m12[i1] = bestIdx;
                m21[bestIdx] = i1;
                matchedDist[bestIdx] = best;
                nmatches++;
                if (checkOri) rotHist[rotBin(kps1[i1].angle, kps2[bestIdx].angle)].push_back(i1);
            }
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();   // stale entries count
        int i1 = -1, i2 = -1, i3 = -1;
        threeMaxima(counts, i1, i2, i3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int idx1 : rotHist[i])
                if (m12[idx1] >= 0) { m12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (m12[i1] >= 0) {
            prevXY[2 * i1] = kps2[m12[i1]].x;
            prevXY[2 * i1 + 1] = kps2[m12[i1]].y;
        }
    return nmatches;
}

// ------------------------------------------------------------------ synthetic vocabulary (§8d, B.5)
FeatVec bowTransform(const uint8_t* desc, int n, const uint8_t* cent)
{
    std::vector<std::pair<uint32_t, int32_t>> pairs;
    pairs.reserve(n);
    for (int i = 0; i < n; i++) {
        const uint8_t* d = desc + 32 * (size_t)i;
        int c1 = 0, b1 = 257;
        for (int c = 0; c < 10; c++) {
            int h = hamming256(d, cent + 32 * (size_t)c);
            if (h < b1) { b1 = h; c1 = c; }           // first minimum wins
        }
        int c2 = 0, b2 = 257;
        for (int c = 0; c < 10; c++) {
            int h = hamming256(d, cent + 32 * (size_t)(10 + 10 * c1 + c));
            if (h < b2) { b2 = h; c2 = c; }
        }
        pairs.push_back({(uint32_t)(11 + 10 * c1 + c2), i});
    }
    std::stable_sort(pairs.begin(), pairs.end(),
                     [](const auto& a, const auto& b) { return a.first < b.first; });
    FeatVec fv;
    for (size_t k = 0; k < pairs.size(); k++) {
        if (k == 0 || pairs[k].first != pairs[k - 1].first) {
            fv.nodeIds.push_back(pairs[k].first);
            fv.offsets.push_back((int32_t)k);
        }
        fv.indices.push_back(pairs[k].second);
    }
    fv.offsets.push_back((int32_t)pairs.size());
    return fv;
}

}  // namespace orbref

// ------------------------------------------------------------------ SearchByProjection (tracking matchers)
// The reference projects MapPoints with cv::Mat arithmetic (src/ORBmatcher.cc:195-212; Frame::isInFrustum for the
// local-map variant).  That O(N) host arithmetic stays in the caller/shim; the oracle (and the GPU entry point)
// start from the projected query: position, window radius, level range, stereo check, descriptor.
namespace orbref {

static void bestTwo(const std::vector<int32_t>& cand, const KeyPoint* kps, const uint8_t* desc, const float* uRight,
                    const std::vector<uint8_t>& occupied, const ProjQuery& q, const uint8_t* qd,
                    int& best, int& best2, int& level, int& level2, int& bestIdx)
{
    best = 256; best2 = 256; level = -1; level2 = -1; bestIdx = -1;
    for (int32_t idx : cand) {
        if (occupied[idx]) continue;                                   // mvpMapPoints[idx] && Observations()>0
        if (uRight && uRight[idx] > 0) {
            const float er = std::fabs(q.ur - uRight[idx]);
            if (er > q.erMax) continue;
        }
        const int d = hamming256(qd, desc + 32 * (size_t)idx);
        if (d < best) { best2 = best; best = d; level2 = level; level = kps[idx].octave; bestIdx = idx; }
        else if (d < best2) { level2 = kps[idx].octave; best2 = d; }
    }
}

// SearchByProjection(Frame&, const vector<MapPoint*>&, th)  src/ORBmatcher.cc:73-157.  matchCur[idx] = query index.
int searchByProjectionMap(const ProjQuery* q, const uint8_t* qDesc, int nq, const KeyPoint* kps, const uint8_t* desc,
                          const float* uRight, const uint8_t* occupiedIn, int n, const FrameGrid& grid, float ratio,
                          int maxDist, std::vector<int32_t>& matchCur)
{
    matchCur.assign(n, -1);
    std::vector<uint8_t> occupied(occupiedIn, occupiedIn + n);
    int nmatches = 0;
    for (int i = 0; i < nq; i++) {
        if (!(q[i].flags & 1)) continue;                               // mbTrackInView && !isBad()
        std::vector<int32_t> cand = grid.inArea(kps, q[i].x, q[i].y, q[i].r, q[i].minLevel, q[i].maxLevel);
        if (cand.empty()) continue;
        int best, best2, level, level2, bestIdx;
        bestTwo(cand, kps, desc, uRight, occupied, q[i], qDesc + 32 * (size_t)i, best, best2, level, level2, bestIdx);
        if (best <= maxDist) {                                         // TH_HIGH
            if (level == level2 && (float)best > ratio * (float)best2) continue;
            matchCur[bestIdx] = i;
            occupied[bestIdx] = (q[i].flags & 2) ? 1 : 0;              // the new MapPoint's Observations()>0
            nmatches++;
        }
    }
    return nmatches;
}

// SearchByProjection(Frame& Current, const Frame& Last, th, bMono)  src/ORBmatcher.cc:160-300.
// matchCur[i2] = last-frame index, -2 = reset to NULL by the rotation filter, -1 = untouched.
int searchByProjectionLast(const ProjQuery* q, const uint8_t* qDesc, const float* qAngle, int nq, const KeyPoint* kps,
                           const uint8_t* desc, const float* uRight, const uint8_t* occupiedIn, int n,
                           const FrameGrid& grid, int maxDist, bool checkOri, std::vector<int32_t>& matchCur)
{
    matchCur.assign(n, -1);
    std::vector<uint8_t> occupied(occupiedIn, occupiedIn + n);
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0;
    for (int i = 0; i < nq; i++) {
        if (!(q[i].flags & 1)) continue;                               // has MapPoint, not outlier, projects inside
        std::vector<int32_t> cand = grid.inArea(kps, q[i].x, q[i].y, q[i].r, q[i].minLevel, q[i].maxLevel);
        if (cand.empty()) continue;
        int best = 256, bestIdx = -1;
        for (int32_t i2 : cand) {
            if (occupied[i2]) continue;
            if (uRight && uRight[i2] > 0) {
                const float er = std::fabs(q[i].ur - uRight[i2]);
                if (er > q[i].erMax) continue;
            }
            const int d = hamming256(qDesc + 32 * (size_t)i, desc + 32 * (size_t)i2);
            if (d < best) { best = d; bestIdx = i2; }
        }
        if (best <= maxDist) {                                         // TH_HIGH / ORBdist / TH_LOW
            matchCur[bestIdx] = i;
            occupied[bestIdx] = (q[i].flags & 2) ? 1 : 0;
            nmatches++;
            if (checkOri) rotHist[rotBin(qAngle[i], kps[bestIdx].angle)].push_back(bestIdx);
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        int i1 = -1, i2 = -1, i3 = -1;
        threeMaxima(counts, i1, i2, i3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int idx : rotHist[i]) { matchCur[idx] = -2; nmatches--; }   // duplicates decrement twice (:287-291)
        }
    }
    return nmatches;
}

}  // namespace orbref

// ------------------------------------------------------------------ DBoW2 vocabulary-tree descent (B.5)
// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) of ORB-SLAM2's vendored DBoW2 (absent from
// the reference tree; restated from its published algorithm): from the root, repeatedly move to the child with
// the smallest Hamming distance (first minimum wins, children in stored order) until a leaf; report the node
// reached at level L - levelsup.  Called by Frame::ComputeBoW (reference src/Frame.cc:425-433) with levelsup = 4.
namespace orbref {

void vocabTransform(const VocabTree& t, const uint8_t* desc, int n, int levelsup, int32_t* wordOf, int32_t* nodeOf)
{
    const int nidLevel = t.L - levelsup;
    for (int i = 0; i < n; i++) {
        const uint8_t* d = desc + 32 * (size_t)i;
        int finalId = 0, level = 0, nid = (nidLevel <= 0) ? 0 : -1;
        do {
            ++level;
            const int b = t.childBegin[finalId], e = t.childBegin[finalId + 1];
            finalId = t.children[b];
            int best = hamming256(d, t.nodeDesc + 32 * (size_t)finalId);
            for (int c = b + 1; c < e; c++) {
                const int id = t.children[c];
                const int h = hamming256(d, t.nodeDesc + 32 * (size_t)id);
                if (h < best) { best = h; finalId = id; }
            }
            if (level == nidLevel) nid = finalId;
        } while (t.childBegin[finalId + 1] > t.childBegin[finalId]);          // !isLeaf()
        wordOf[i] = t.wordId[finalId];
        nodeOf[i] = nid;                                                     // -1: leaf reached above that level
    }
}

}  // namespace orbref

// ------------------------------------------------------------------ MapPoint::ComputeDistinctiveDescriptors
// reference src/MapPoint.cc:275-342: among the N observed descriptors of a MapPoint pick the one whose MEDIAN Hamming
// distance to all N (itself included, distance 0) is smallest; median = sorted[(int)(0.5*(N-1))]; first minimum wins.
namespace orbref {

void distinctiveDescriptors(const uint8_t* desc, const int32_t* offsets, int nPoints, int32_t* bestIdx)
{
    for (int p = 0; p < nPoints; p++) {
        const int b = offsets[p], N = offsets[p + 1] - b;
        if (N <= 0) { bestIdx[p] = -1; continue; }
        int bestMedian = INT_MAX, best = 0;
        std::vector<int> row(N);
        for (int i = 0; i < N; i++) {
            for (int j = 0; j < N; j++) row[j] = (i == j) ? 0 : hamming256(desc + 32 * (size_t)(b + i), desc + 32 * (size_t)(b + j));
            std::sort(row.begin(), row.end());
            const int median = row[(int)(0.5 * (N - 1))];
            if (median < bestMedian) { bestMedian = median; best = i; }
        }
        bestIdx[p] = best;
    }
}

}  // namespace orbref

// ------------------------------------------------------------------ independent best candidate per projected point
// The search loops of ORBmatcher::Fuse (src/ORBmatcher.cc:1386-1432, with the chi-square reprojection gate, and
// :1570-1600 without) and of ORBmatcher::SearchBySim3 (:905-932, :975-1002): window query, level range, first
// minimum; no state is carried between points.
namespace orbref {

void searchByProjectionBest(const ProjQuery* q, const uint8_t* qDesc, int nq, const KeyPoint* kps, const uint8_t* desc,
                            const float* uRight, int /*n*/, const FrameGrid& grid, int maxDist, bool chi2,
                            const float* invSigma2, int32_t* bestIdx, int32_t* bestDist)
{
    for (int i = 0; i < nq; i++) {
        bestIdx[i] = -1;
        if (bestDist) bestDist[i] = 256;
        if (!(q[i].flags & 1)) continue;
        std::vector<int32_t> cand = grid.inArea(kps, q[i].x, q[i].y, q[i].r, q[i].minLevel, q[i].maxLevel);
        int best = INT_MAX, idxBest = -1;
        for (int32_t idx : cand) {
            const KeyPoint& kp = kps[idx];
            if (chi2) {
                const float u = q[i].x, v = q[i].y;
                if (uRight && uRight[idx] >= 0) {
                    const float ex = u - kp.x, ey = v - kp.y, er = q[i].ur - uRight[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * invSigma2[kp.octave] > 7.8) continue;
                } else {
                    const float ex = u - kp.x, ey = v - kp.y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * invSigma2[kp.octave] > 5.99) continue;
                }
            } else if (uRight && uRight[idx] > 0) {
                if (std::fabs(q[i].ur - uRight[idx]) > q[i].erMax) continue;
            }
            const int d = hamming256(qDesc + 32 * (size_t)i, desc + 32 * (size_t)idx);
            if (d < best) { best = d; idxBest = idx; }
        }
        if (idxBest >= 0 && bestDist) bestDist[i] = best;
        if (idxBest >= 0 && best <= maxDist) bestIdx[i] = idxBest;
    }
}

}  // namespace orbref

// ------------------------------------------------------------------ SearchForTriangulation (src/ORBmatcher.cc:1183-1359)
namespace orbref {

static bool checkDistEpipolarLine(const KeyPoint& kp1, const KeyPoint& kp2, const float* F12, const float* levelSigma2)
{
    // :1636-1650
    const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
    const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
    const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
    const float num = a * kp2.x + b * kp2.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * levelSigma2[kp2.octave];
}

int searchForTriangulation(const KeyPoint* k1, const uint8_t* d1, const uint8_t* hasMP1, const float* uR1, int n1,
                           const FeatVec& fv1, const KeyPoint* k2, const uint8_t* d2, const uint8_t* hasMP2,
                           const float* uR2, int n2, const FeatVec& fv2, const float* F12, float ex, float ey,
                           const float* scaleFactors2, const float* levelSigma2_2, bool onlyStereo, bool checkOri,
                           std::vector<int32_t>& matches12)
{
    matches12.assign(n1, -1);
    std::vector<uint8_t> matched2(n2, 0);            // never set by the reference (:1198, :1235): kept for fidelity
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0;
    size_t a = 0, b = 0;
    while (a < fv1.nodeIds.size() && b < fv2.nodeIds.size()) {
        if (fv1.nodeIds[a] == fv2.nodeIds[b]) {
            for (int p = fv1.offsets[a]; p < fv1.offsets[a + 1]; p++) {
                const int idx1 = fv1.indices[p];
                if (hasMP1[idx1]) continue;
                const bool stereo1 = uR1 && uR1[idx1] >= 0;
                if (onlyStereo && !stereo1) continue;
                const KeyPoint& kp1 = k1[idx1];
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int q = fv2.offsets[b]; q < fv2.offsets[b + 1]; q++) {
                    const int idx2 = fv2.indices[q];
                    if (matched2[idx2] || hasMP2[idx2]) continue;
                    const bool stereo2 = uR2 && uR2[idx2] >= 0;
                    if (onlyStereo && !stereo2) continue;
                    const int dist = hamming256(d1 + 32 * (size_t)idx1, d2 + 32 * (size_t)idx2);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const KeyPoint& kp2 = k2[idx2];
                    if (!stereo1 && !stereo2) {
                        const float distex = ex - kp2.x, distey = ey - kp2.y;
                        if (distex * distex + distey * distey < 100 * scaleFactors2[kp2.octave]) continue;
                    }
                    if (checkDistEpipolarLine(kp1, kp2, F12, levelSigma2_2)) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    matches12[idx1] = bestIdx2;
                    nmatches++;
                    if (checkOri) rotHist[rotBin(kp1.angle, k2[bestIdx2].angle)].push_back(idx1);
                }
            }
            a++; b++;
        } else if (fv1.nodeIds[a] < fv2.nodeIds[b]) {
            a = lowerBound(fv1.nodeIds, fv2.nodeIds[b]);
        } else {
            b = lowerBound(fv2.nodeIds, fv1.nodeIds[a]);
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        int i1 = -1, i2 = -1, i3 = -1;
        threeMaxima(counts, i1, i2, i3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int idx : rotHist[i]) { matches12[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

}  // namespace orbref
