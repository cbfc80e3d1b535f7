// orbref_stereo.cpp -- CPU ORACLE (test infrastructure only; see orbref.hpp header note).
// Restates Frame::ComputeStereoMatches, reference src/Frame.cc:513-699 ("S1" of SURVEY 8a): row table,
// Hamming coarse match, 11x11 SAD sliding +-5 on the pyramid level of the left keypoint, parabola
// sub-pixel fit, median outlier cut.  PARITY UNPINNED (no reference fixtures exist).
#include <algorithm>
#include <climits>
#include <cmath>
#include <utility>
#include <vector>

#include "orbref.hpp"

namespace orbref {

void stereoMatches(const Extractor& exL, const Extractor& exR,
                   const KeyPoint* kL, const uint8_t* dL, int N,
                   const KeyPoint* kR, const uint8_t* dR, int Nr,
                   float mb, float mbf, float* uRight, float* depth)
{
    for (int i = 0; i < N; i++) { uRight[i] = -1.0f; depth[i] = -1.0f; }
    const int thOrbDist = (100 + 50) / 2;                          // (TH_HIGH + TH_LOW)/2  (:518)
    const int nRows = exL.pyramid[0].h;
    std::vector<std::vector<int>> rowIdx(nRows);
    for (int iR = 0; iR < Nr; iR++) {
        const float kpY = kR[iR].y;
        const float r = 2.0f * exL.mvScaleFactor[kR[iR].octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) rowIdx[yi].push_back(iR);   // reference has no bounds check (cannot trigger)
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<std::pair<int, int>> distIdx;
    for (int iL = 0; iL < N; iL++) {
        const KeyPoint& kpL = kL[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        const size_t row = (size_t)vL;
        if (row >= (size_t)nRows) continue;
        const std::vector<int>& cand = rowIdx[row];
        if (cand.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = 100;                                        // TH_HIGH
        int bestIdxR = 0;
        for (int iR : cand) {
            const KeyPoint& kpR = kR[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = hamming256(dL + 32 * (size_t)iL, dR + 32 * (size_t)iR);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = kR[bestIdxR].x;
            const float scaleFactor = exL.mvInvScaleFactor[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor);
            const float scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const Image& imL = exL.pyramid[kpL.octave];
            const Image& imR = exR.pyramid[kpL.octave];
            const int y0 = (int)(scaledvL - w), xL0 = (int)(scaleduL - w);
            float IL[11][11];
            const float cL = (float)imL.at(y0 + w, xL0 + w);
            for (int y = 0; y < 11; y++)
                for (int x = 0; x < 11; x++) IL[y][x] = (float)imL.at(y0 + y, xL0 + x) - cL;
            int bestSad = INT_MAX, bestincR = 0;
            float vDists[2 * 5 + 1];
            const float iniu = scaleduR0 + L - w;                  // sic (:624)
            const float endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= imR.w) continue;
            for (int incR = -L; incR <= L; incR++) {
                const int xR0 = (int)(scaleduR0 + incR - w);
                const float cR = (float)imR.at(y0 + w, xR0 + w);
                double acc = 0;                                    // cv::norm(NORM_L1) accumulates in double
                for (int y = 0; y < 11; y++)
                    for (int x = 0; x < 11; x++) acc += std::fabs(IL[y][x] - ((float)imR.at(y0 + y, xR0 + x) - cR));
                const float dist = (float)acc;
                if (dist < bestSad) { bestSad = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = exL.mvScaleFactor[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = 0.01;
                    bestuR = uL - 0.01;
                }
                depth[iL] = mbf / disparity;
                uRight[iL] = bestuR;
                distIdx.push_back({bestSad, iL});
            }
        }
    }
    if (distIdx.empty()) return;                                   // reference: UB on an empty vector (:686)
    std::sort(distIdx.begin(), distIdx.end());
    const float median = (float)distIdx[distIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)distIdx.size() - 1; i >= 0; i--) {
        if (distIdx[i].first < thDist) break;
        uRight[distIdx[i].second] = -1;
        depth[distIdx[i].second] = -1;
    }
}

}  // namespace orbref

extern "C" void orbref_stereo(void* hL, void* hR, const orbref::KeyPoint* kL, const uint8_t* dL, int N,
                              const orbref::KeyPoint* kR, const uint8_t* dR, int Nr, float mb, float mbf,
                              float* uRight, float* depth)
{
    orbref::stereoMatches(*(orbref::Extractor*)hL, *(orbref::Extractor*)hR, kL, dL, N, kR, dR, Nr, mb, mbf, uRight, depth);
}
