// orb_matcher_init.hip -- ORBmatcher::SearchForInitialization on gfx950
// (reference src/ORBmatcher.cc:1055-1180) together with the Frame grid it queries
// (src/Frame.cc:243-259 AssignFeaturesToGrid, :348-409 GetFeaturesInArea, :412-422 PosInGrid;
// 64x48 cells, include/Frame.h:37-38).
//
// Three launches on the matcher's stream:
//   k_init_grid        frame-2 level-0 keypoints -> (cell, index) keys, sorted: the grid as one
//                      sorted array, so the cells (ix, iyMin..iyMax) of a window query are ONE
//                      contiguous range per column ix, already in the reference's iteration order
//                      (ix outer, iy inner, insertion order inside a cell).
//   k_init_candidates  one wave per level-0 keypoint of frame 1: window query + Hamming distance
//                      to every candidate, written in reference order (parallel part).
//   k_init_resolve     one wave replays the order-dependent part serially over i1: the
//                      vMatchedDistance skip rule, best/second-best, ratio test, match stealing,
//                      rotation histogram (stale entries keep counting), top-3 filter, vbPrevMatched.
#include <algorithm>
#include <vector>

#include "orb_matcher_internal.h"

#pragma clang fp contract(off)

#define WAVE 64
#define TH_LOW 50
#define HISTO_LENGTH 30
#define DIST_NONE 0x7FFFFFFF

#include "orb_grid_device.h"

// candList[i1*stride + k] = (i2 << 16) | dist, k < candCount[i1], in GetFeaturesInArea order
__global__ __launch_bounds__(WAVE) void k_init_candidates(const orb_keypoint* __restrict__ kps1,
                                                          const uint8_t* __restrict__ desc1, int n1,
                                                          const orb_keypoint* __restrict__ kps2,
                                                          const uint8_t* __restrict__ desc2,
                                                          const uint32_t* __restrict__ keys,
                                                          const int* __restrict__ nKeysPtr, InitGrid g,
                                                          const float* __restrict__ prevXY, float r,
                                                          uint32_t* __restrict__ candList, int stride,
                                                          int* __restrict__ candCount)
{
    const int i1 = blockIdx.x, lane = threadIdx.x;
    if (i1 >= n1) return;
    if (kps1[i1].octave > 0) { if (lane == 0) candCount[i1] = 0; return; }      // :1074-1076
    const int nKeys = *nKeysPtr;
    const float x = prevXY[2 * i1], y = prevXY[2 * i1 + 1];
    // cell window (:355-372)
    const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, g.minX), r), g.invW)));
    const int maxCX = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, g.minX), r), g.invW)));
    const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, g.minY), r), g.invH)));
    const int maxCY = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, g.minY), r), g.invH)));
    int count = 0;
    if (!(minCX >= GRID_COLS || maxCX < 0 || minCY >= GRID_ROWS || maxCY < 0)) {
        uint32_t d1[8];
        load_desc8(desc1 + (size_t)i1 * 32, d1);
        uint32_t* out = candList + (size_t)i1 * stride;
        for (int ix = minCX; ix <= maxCX; ix++) {
            if (minCY > maxCY) break;
            const int a = lower_key(keys, nKeys, (uint32_t)(ix * GRID_ROWS + minCY) << 16);
            const int b = lower_key(keys, nKeys, (uint32_t)(ix * GRID_ROWS + maxCY + 1) << 16);
            for (int base = a; base < b; base += WAVE) {
                const int k = base + lane;
                bool ok = false;
                uint32_t rec = 0;
                if (k < b) {
                    const int i2 = (int)(keys[k] & 0xFFFFu);
                    const float dx = __fsub_rn(kps2[i2].x, x), dy = __fsub_rn(kps2[i2].y, y);
                    if (fabsf(dx) < r && fabsf(dy) < r) {                     // :401-403
                        uint32_t d2[8];
                        load_desc8(desc2 + (size_t)i2 * 32, d2);
                        int dist = 0;
#pragma unroll
                        for (int w = 0; w < 8; w++) dist += __popc(d1[w] ^ d2[w]);
                        ok = true;
                        rec = ((uint32_t)i2 << 16) | (uint32_t)dist;
                    }
                }
                const unsigned long long bal = __ballot(ok);
                if (ok) out[count + __popcll(bal & ((1ull << lane) - 1))] = rec;
                count += __popcll(bal);
            }
        }
    }
    if (lane == 0) candCount[i1] = count;
}

__global__ __launch_bounds__(WAVE) void k_init_resolve(const orb_keypoint* __restrict__ kps1, int n1,
                                                       const orb_keypoint* __restrict__ kps2, int n2,
                                                       const uint32_t* __restrict__ candList, int stride,
                                                       const int* __restrict__ candCount, float ratio, int checkOri,
                                                       float* __restrict__ prevXY, int32_t* __restrict__ m12,
                                                       int* __restrict__ matchedDist /*[n2]*/,
                                                       int* __restrict__ m21 /*[n2]*/,
                                                       uint8_t* __restrict__ binOf /*[n1]*/,
                                                       int32_t* __restrict__ nmatchesOut)
{
    __shared__ int hist[HISTO_LENGTH];
    const int lane = threadIdx.x;
    for (int i = lane; i < n2; i += WAVE) { matchedDist[i] = DIST_NONE; m21[i] = -1; }
    for (int i = lane; i < n1; i += WAVE) { m12[i] = -1; binOf[i] = 0xFF; }
    if (lane < HISTO_LENGTH) hist[lane] = 0;
    __threadfence_block();
    __syncthreads();
    for (int i1 = 0; i1 < n1; i1++) {
        const int nc = candCount[i1];
        if (nc == 0) continue;                                        // also covers level1 > 0
        const uint32_t* list = candList + (size_t)i1 * stride;
        unsigned best = 0xFFFFFFFFu;                                  // (dist << 16 | position); none yet
        unsigned second = 0xFFFFFFFFu;                                // dist only; none yet
        int bestI2 = -1;
        for (int base = 0; base < nc; base += WAVE) {
            const int q = base + lane;
            unsigned mine = 0xFFFFFFFFu;
            int i2 = -1;
            if (q < nc) {
                const uint32_t rec = list[q];
                i2 = (int)(rec >> 16);
                const int dist = (int)(rec & 0xFFFFu);
                if (!(matchedDist[i2] <= dist)) mine = ((unsigned)dist << 16) | (unsigned)(q - base);   // :1094
            }
            const unsigned m1 = orb_wave_umin(mine);
            const unsigned m2 = orb_wave_umin(mine == m1 ? 0xFFFFFFFFu : mine);
            if (m1 != 0xFFFFFFFFu) {
                const unsigned d1 = m1 >> 16;
                const unsigned d2 = (m2 == 0xFFFFFFFFu) ? 0xFFFFFFFFu : (m2 >> 16);
                const unsigned bd = (best == 0xFFFFFFFFu) ? 0xFFFFFFFFu : (best >> 16);
                if (d1 < bd) {
                    second = min(bd, d2);
                    best = m1;
                    bestI2 = __shfl(i2, (int)(m1 & 0xFFFFu));
                } else {
                    second = min(second, d1);
                }
            }
        }
        if (best == 0xFFFFFFFFu) continue;
        const int bestDist = (int)(best >> 16);
        if (bestDist <= TH_LOW) {
            const float second_f = (second == 0xFFFFFFFFu) ? (float)DIST_NONE : (float)(int)second;
            if ((float)bestDist < __fmul_rn(second_f, ratio)) {       // :1112
                if (lane == 0) {
                    const int old = m21[bestI2];
                    if (old >= 0) m12[old] = -1;                      // steal (:1115-1119)
                    m12[i1] = bestI2;
                    m21[bestI2] = i1;
                    matchedDist[bestI2] = bestDist;
                    if (checkOri) {
                        float rot = __fsub_rn(kps1[i1].angle, kps2[bestI2].angle);
                        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                        int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
                        if (bin == HISTO_LENGTH) bin = 0;
                        binOf[i1] = (uint8_t)bin;
                        hist[bin]++;                                  // entries of stolen matches stay counted
                    }
                }
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    __syncthreads();
    // the reference's running count (++ on accept, -- on steal) equals the number of live matches
    int alive = 0;
    for (int i = lane; i < n1; i += WAVE) alive += (m12[i] >= 0);
    alive = orb_wave_sum(alive);
    int nmatches = alive;
    if (checkOri) {
        int i1 = -1, i2 = -1, i3 = -1;
        {
            int max1 = 0, max2 = 0, max3 = 0;
            for (int i = 0; i < HISTO_LENGTH; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
                else if (s > max3) { max3 = s; i3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { i3 = -1; }
        }
        int dropped = 0;
        for (int i = lane; i < n1; i += WAVE) {
            const int b = binOf[i];
            if (b != 0xFF && b != i1 && b != i2 && b != i3 && m12[i] >= 0) { m12[i] = -1; dropped++; }   // :1163-1167
        }
        dropped = orb_wave_sum(dropped);
        nmatches -= dropped;
    }
    __threadfence_block();
    __syncthreads();
    for (int i = lane; i < n1; i += WAVE) {
        const int j = m12[i];
        if (j >= 0) { prevXY[2 * i] = kps2[j].x; prevXY[2 * i + 1] = kps2[j].y; }     // :1175-1177
    }
    if (lane == 0) *nmatchesOut = nmatches;
}

// ------------------------------------------------------------------ host side
extern "C" int orb_match_init(orb_matcher* m, const orb_keypoint* kps1, const uint8_t* desc1, int n1,
                              const orb_keypoint* kps2, const uint8_t* desc2, int n2, const float* grid4,
                              float* prev_xy, int window_size, float ratio, int check_ori, int32_t* match_12,
                              int* nmatches)
{
    if (!m || n1 < 0 || n2 < 0 || !nmatches || !grid4) return ORB_ERR_INVALID;
    *nmatches = 0;
    if (n1 > 0 && (!match_12 || !prev_xy)) return ORB_ERR_INVALID;
    for (int i = 0; i < n1; i++) match_12[i] = -1;
    if (n1 == 0 || n2 == 0) return ORB_OK;
    if (!kps1 || !desc1 || !kps2 || !desc2) return ORB_ERR_INVALID;
    if (n1 > 65535 || n2 > 65535) return ORB_ERR_UNSUPPORTED;
    ORB_HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    MBuf* buf = m->init;
    const size_t sz[12] = {sizeof(orb_keypoint) * (size_t)n1, (size_t)32 * n1, sizeof(orb_keypoint) * (size_t)n2,
                           (size_t)32 * n2, (size_t)8 * n1, (size_t)4 * n2 + 4, (size_t)4 * n1 * n2, (size_t)4 * n1,
                           (size_t)4 * n1 + 4, (size_t)4 * n2, (size_t)4 * n2, (size_t)n1};
    int rc;
    for (int i = 0; i < 12; i++)
        if ((rc = buf[i].ensure(sz[i])) != ORB_OK) return rc;
    orb_keypoint* dK1 = (orb_keypoint*)buf[0].p;
    uint8_t* dD1 = (uint8_t*)buf[1].p;
    orb_keypoint* dK2 = (orb_keypoint*)buf[2].p;
    uint8_t* dD2 = (uint8_t*)buf[3].p;
    float* dPrev = (float*)buf[4].p;
    uint32_t* dKeys = (uint32_t*)buf[5].p;
    int* dNKeys = (int*)((uint8_t*)buf[5].p + (size_t)4 * n2);
    uint32_t* dCand = (uint32_t*)buf[6].p;
    int* dCandCount = (int*)buf[7].p;
    int32_t* dM12 = (int32_t*)buf[8].p;
    int32_t* dNm = dM12 + n1;
    int* dMatchedDist = (int*)buf[9].p;
    int* dM21 = (int*)buf[10].p;
    uint8_t* dBin = (uint8_t*)buf[11].p;
    ORB_HIP_TRY(hipMemcpyAsync(dK1, kps1, sz[0], hipMemcpyHostToDevice, st));
    ORB_HIP_TRY(hipMemcpyAsync(dD1, desc1, sz[1], hipMemcpyHostToDevice, st));
    ORB_HIP_TRY(hipMemcpyAsync(dK2, kps2, sz[2], hipMemcpyHostToDevice, st));
    ORB_HIP_TRY(hipMemcpyAsync(dD2, desc2, sz[3], hipMemcpyHostToDevice, st));
    ORB_HIP_TRY(hipMemcpyAsync(dPrev, prev_xy, sz[4], hipMemcpyHostToDevice, st));
    InitGrid g = {grid4[0], grid4[1], grid4[2], grid4[3]};
    hipLaunchKernelGGL(k_init_grid, dim3(1), dim3(256), 0, st, dK2, n2, g, 1, dKeys, dNKeys);
    hipLaunchKernelGGL(k_init_candidates, dim3(n1), dim3(WAVE), 0, st, dK1, dD1, n1, dK2, dD2, dKeys, dNKeys, g, dPrev,
                       (float)window_size, dCand, n2, dCandCount);
    hipLaunchKernelGGL(k_init_resolve, dim3(1), dim3(WAVE), 0, st, dK1, n1, dK2, n2, dCand, n2, dCandCount, ratio,
                       check_ori, dPrev, dM12, dMatchedDist, dM21, dBin, dNm);
    ORB_HIP_TRY(hipGetLastError());
    std::vector<int32_t> host((size_t)n1 + 1);
    ORB_HIP_TRY(hipMemcpyAsync(host.data(), dM12, ((size_t)n1 + 1) * 4, hipMemcpyDeviceToHost, st));
    ORB_HIP_TRY(hipMemcpyAsync(prev_xy, dPrev, sz[4], hipMemcpyDeviceToHost, st));
    ORB_HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < n1; i++) match_12[i] = host[i];
    *nmatches = host[n1];
    return ORB_OK;
}
