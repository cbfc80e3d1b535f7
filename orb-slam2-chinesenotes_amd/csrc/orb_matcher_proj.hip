// orb_matcher_proj.hip -- the two tracking matchers on gfx950:
//   mode 0  ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th, bMono)   src/ORBmatcher.cc:160-300
//   mode 1  ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)          src/ORBmatcher.cc:73-157
//
// The projection itself (cv::Mat arithmetic, :195-212; Frame::isInFrustum) is O(N) host work that stays in the
// shim, written with the reference's own expressions; the entry point starts from the projected query.
// Same split as SearchForInitialization: a parallel window-query + Hamming phase (one wave per query, candidates in
// GetFeaturesInArea order, static filters applied), then one wave replays the order-dependent part: the
// "feature already has a MapPoint with observations" rule changes as matches are assigned (:109-111, :233-235).
#include <algorithm>
#include <vector>

#include "orb_grid_device.h"
#include "orb_matcher_internal.h"

#pragma clang fp contract(off)

#define WAVE 64
#define HISTO_LENGTH 30

struct Sigma2Tab { float inv[16]; };   // mvInvLevelSigma2 of the searched KeyFrame

struct ProjQuery {                 // == orb_proj_query
    float x, y, r;
    int32_t minLevel, maxLevel;
    float ur, erMax;
    int32_t flags;
};
static_assert(sizeof(ProjQuery) == sizeof(orb_proj_query), "orb_proj_query layout");

static __device__ __forceinline__ unsigned pj_umin_dpp(unsigned v)
{
#define PJ_DPP(ctrl, rmask) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(v), ctrl, rmask, 0xf, false))
    PJ_DPP(0x111, 0xf); PJ_DPP(0x112, 0xf); PJ_DPP(0x114, 0xf); PJ_DPP(0x118, 0xf); PJ_DPP(0x142, 0xa); PJ_DPP(0x143, 0xc);
#undef PJ_DPP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// candList[i*stride + k] = (i2 << 16) | dist, k < candCount[i], in GetFeaturesInArea order, after the static
// filters: level range (:390-397 of Frame.cc), |dx|<r && |dy|<r, stereo check (:113-118 / :238-244 of ORBmatcher.cc)
__global__ __launch_bounds__(WAVE) void k_proj_candidates(const ProjQuery* __restrict__ q,
                                                          const uint8_t* __restrict__ qDesc, int nq,
                                                          const orb_keypoint* __restrict__ kps,
                                                          const uint8_t* __restrict__ desc,
                                                          const float* __restrict__ uRight,
                                                          const uint32_t* __restrict__ keys,
                                                          const int* __restrict__ nKeysPtr, InitGrid g,
                                                          uint32_t* __restrict__ candList, int stride,
                                                          int* __restrict__ candCount, int chi2, Sigma2Tab sig)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= nq) return;
    const ProjQuery Q = q[i];
    if (!(Q.flags & 1)) { if (lane == 0) candCount[i] = 0; return; }
    const int nKeys = *nKeysPtr;
    const float x = Q.x, y = Q.y, r = Q.r;
    const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, g.minX), r), g.invW)));
    const int maxCX = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, g.minX), r), g.invW)));
    const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, g.minY), r), g.invH)));
    const int maxCY = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, g.minY), r), g.invH)));
    const bool checkLevels = (Q.minLevel > 0) || (Q.maxLevel >= 0);
    int count = 0;
    if (!(minCX >= GRID_COLS || maxCX < 0 || minCY >= GRID_ROWS || maxCY < 0) && minCY <= maxCY) {
        uint32_t d1[8];
        load_desc8(qDesc + (size_t)i * 32, d1);
        uint32_t* out = candList + (size_t)i * stride;
        for (int ix = minCX; ix <= maxCX; ix++) {
            const int a = lower_key(keys, nKeys, (uint32_t)(ix * GRID_ROWS + minCY) << 16);
            const int b = lower_key(keys, nKeys, (uint32_t)(ix * GRID_ROWS + maxCY + 1) << 16);
            for (int base = a; base < b; base += WAVE) {
                const int k = base + lane;
                bool ok = false;
                uint32_t rec = 0;
                if (k < b) {
                    const int i2 = (int)(keys[k] & 0xFFFFu);
                    const orb_keypoint kp = kps[i2];
                    bool pass = true;
                    if (checkLevels) {
                        if (kp.octave < Q.minLevel) pass = false;
                        if (Q.maxLevel >= 0 && kp.octave > Q.maxLevel) pass = false;
                    }
                    const float dx = __fsub_rn(kp.x, x), dy = __fsub_rn(kp.y, y);
                    if (!(fabsf(dx) < r && fabsf(dy) < r)) pass = false;
                    if (pass) {
                        const float ur2 = uRight ? uRight[i2] : -1.0f;
                        if (!chi2) {
                            if (ur2 > 0 && fabsf(__fsub_rn(Q.ur, ur2)) > Q.erMax) pass = false;
                        } else {
                            // reprojection-error gate of ORBmatcher::Fuse (:1401-1426): chi-square 3 dof / 2 dof
                            const float ex = __fsub_rn(x, kp.x), ey = __fsub_rn(y, kp.y);
                            float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                            const float is2 = sig.inv[min(max(kp.octave, 0), 15)];
                            if (ur2 >= 0) {
                                const float er = __fsub_rn(Q.ur, ur2);
                                e2 = __fadd_rn(e2, __fmul_rn(er, er));
                                if ((double)__fmul_rn(e2, is2) > 7.8) pass = false;
                            } else if ((double)__fmul_rn(e2, is2) > 5.99) pass = false;
                        }
                    }
                    if (pass) {
                        uint32_t d2[8];
                        load_desc8(desc + (size_t)i2 * 32, d2);
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
    if (lane == 0) candCount[i] = count;
}

// serial replay by one wave.  matchCur: -1 untouched, >= 0 query index, -2 reset by the rotation filter (mode 0)
__global__ __launch_bounds__(WAVE) void k_proj_resolve(int mode, const ProjQuery* __restrict__ q,
                                                       const float* __restrict__ qAngle, int nq,
                                                       const orb_keypoint* __restrict__ kps, int n,
                                                       const uint32_t* __restrict__ candList, int stride,
                                                       const int* __restrict__ candCount, float ratio, int maxDist,
                                                       int checkOri,
                                                       uint8_t* __restrict__ occupied /*in/out copy*/,
                                                       int32_t* __restrict__ matchCur,
                                                       uint32_t* __restrict__ events /*[nq] bin<<16|idx*/,
                                                       int32_t* __restrict__ nmatchesOut)
{
    __shared__ int hist[HISTO_LENGTH];
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += WAVE) matchCur[i] = -1;
    if (lane < HISTO_LENGTH) hist[lane] = 0;
    __threadfence_block();
    __syncthreads();
    int nmatches = 0, nEvents = 0;
    for (int i = 0; i < nq; i++) {
        const int nc = candCount[i];
        if (nc == 0) continue;
        const uint32_t* list = candList + (size_t)i * stride;
        unsigned b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu;              // two smallest (dist << 16 | position)
        for (int base = 0; base < nc; base += WAVE) {
            const int p = base + lane;
            unsigned mine = 0xFFFFFFFFu;
            if (p < nc) {
                const uint32_t rec = list[p];
                if (!occupied[rec >> 16]) mine = ((rec & 0xFFFFu) << 16) | (unsigned)p;      // :109-111 / :233-235
            }
            const unsigned m1 = pj_umin_dpp(mine);
            const unsigned m2 = pj_umin_dpp(mine == m1 ? 0xFFFFFFFFu : mine);
            // merge the chunk's two smallest into the running two smallest
            if (m1 < b1) { b2 = min(b1, m2); b1 = m1; }
            else { b2 = min(b2, m1); }
        }
        if (b1 == 0xFFFFFFFFu) continue;
        const int best = (int)(b1 >> 16);
        const int bestIdx = (int)(list[b1 & 0xFFFFu] >> 16);
        if (best > maxDist) continue;                             // TH_HIGH (:136, :258), ORBdist (:379), TH_LOW (:493)
        if (mode == 1) {
            const int best2 = (b2 == 0xFFFFFFFFu) ? 256 : (int)(b2 >> 16);
            const int level = kps[bestIdx].octave;
            const int level2 = (b2 == 0xFFFFFFFFu) ? -1 : kps[list[b2 & 0xFFFFu] >> 16].octave;
            if (level == level2 && (float)best > __fmul_rn(ratio, (float)best2)) continue;    // :148-149
        }
        if (lane == 0) {
            matchCur[bestIdx] = i;
            occupied[bestIdx] = (q[i].flags & 2) ? 1 : 0;
            if (mode == 0 && checkOri) {
                float rot = __fsub_rn(qAngle[i], kps[bestIdx].angle);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
                hist[bin]++;
                events[nEvents] = ((uint32_t)bin << 16) | (uint32_t)bestIdx;
            }
        }
        nEvents++;
        nmatches++;
        __threadfence_block();
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    if (mode == 0 && checkOri) {
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
        for (int e = lane; e < nEvents; e += WAVE) {
            const uint32_t ev = events[e];
            const int b = (int)(ev >> 16);
            if (b != i1 && b != i2 && b != i3) { matchCur[ev & 0xFFFFu] = -2; dropped++; }       // :285-293
        }
        dropped = orb_wave_sum(dropped);
        nmatches -= dropped;
    }
    if (lane == 0) *nmatchesOut = nmatches;
}

// independent best candidate per query (no assignment state): the search loops of ORBmatcher::Fuse x2 and of
// ORBmatcher::SearchBySim3 (first minimum wins, `dist < bestDist`)
__global__ __launch_bounds__(WAVE) void k_proj_best(const uint32_t* __restrict__ candList, int stride,
                                                    const int* __restrict__ candCount, int nq, int maxDist,
                                                    int32_t* __restrict__ bestIdx, int32_t* __restrict__ bestDist)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= nq) return;
    const int nc = candCount[i];
    const uint32_t* list = candList + (size_t)i * stride;
    unsigned b1 = 0xFFFFFFFFu;
    for (int base = 0; base < nc; base += WAVE) {
        const int p = base + lane;
        unsigned mine = 0xFFFFFFFFu;
        if (p < nc) mine = ((list[p] & 0xFFFFu) << 16) | (unsigned)p;
        b1 = min(b1, pj_umin_dpp(mine));
    }
    if (lane == 0) {
        const bool ok = b1 != 0xFFFFFFFFu && (int)(b1 >> 16) <= maxDist;
        bestIdx[i] = ok ? (int32_t)(list[b1 & 0xFFFFu] >> 16) : -1;
        if (bestDist) bestDist[i] = (b1 == 0xFFFFFFFFu) ? 256 : (int32_t)(b1 >> 16);
    }
}

static int proj_common(orb_matcher* m, const orb_proj_query* queries, const uint8_t* q_desc, const float* q_angle, int nq,
                       const orb_keypoint* kps_un, const uint8_t* desc, const float* u_right, const uint8_t* occupied,
                       int n, const float* grid4, int chi2, const float* inv_sigma2, int n_levels, MBuf* buf)
{
    const size_t sz[12] = {sizeof(ProjQuery) * (size_t)nq, (size_t)32 * nq, (size_t)4 * nq, sizeof(orb_keypoint) * (size_t)n,
                           (size_t)32 * n, (size_t)4 * n, (size_t)n, (size_t)4 * n + 4, (size_t)4 * nq * n, (size_t)4 * nq,
                           (size_t)4 * std::max(n, nq) + 4, (size_t)4 * nq};
    int rc;
    for (int i = 0; i < 12; i++)
        if ((rc = buf[i].ensure(sz[i])) != ORB_OK) return rc;
    hipStream_t st = m->stream;
    ORB_HIP_TRY(hipMemcpyAsync(buf[0].p, queries, sz[0], hipMemcpyHostToDevice, st));
    ORB_HIP_TRY(hipMemcpyAsync(buf[1].p, q_desc, sz[1], hipMemcpyHostToDevice, st));
    if (q_angle) ORB_HIP_TRY(hipMemcpyAsync(buf[2].p, q_angle, sz[2], hipMemcpyHostToDevice, st));
    ORB_HIP_TRY(hipMemcpyAsync(buf[3].p, kps_un, sz[3], hipMemcpyHostToDevice, st));
    ORB_HIP_TRY(hipMemcpyAsync(buf[4].p, desc, sz[4], hipMemcpyHostToDevice, st));
    if (u_right) ORB_HIP_TRY(hipMemcpyAsync(buf[5].p, u_right, sz[5], hipMemcpyHostToDevice, st));
    if (occupied) ORB_HIP_TRY(hipMemcpyAsync(buf[6].p, occupied, sz[6], hipMemcpyHostToDevice, st));
    InitGrid g = {grid4[0], grid4[1], grid4[2], grid4[3]};
    Sigma2Tab sig;
    for (int i = 0; i < 16; i++) sig.inv[i] = (inv_sigma2 && i < n_levels) ? inv_sigma2[i] : 1.0f;
    uint32_t* dKeys = (uint32_t*)buf[7].p;
    int* dNKeys = (int*)((uint8_t*)buf[7].p + (size_t)4 * n);
    hipLaunchKernelGGL(k_init_grid, dim3(1), dim3(256), 0, st, (const orb_keypoint*)buf[3].p, n, g, 0, dKeys, dNKeys);
    hipLaunchKernelGGL(k_proj_candidates, dim3(nq), dim3(WAVE), 0, st, (const ProjQuery*)buf[0].p, (const uint8_t*)buf[1].p, nq,
                       (const orb_keypoint*)buf[3].p, (const uint8_t*)buf[4].p, u_right ? (const float*)buf[5].p : nullptr, dKeys,
                       dNKeys, g, (uint32_t*)buf[8].p, n, (int*)buf[9].p, chi2, sig);
    return ORB_OK;
}

extern "C" int orb_match_projection_best(orb_matcher* m, const orb_proj_query* queries, const uint8_t* q_desc, int nq,
                                         const orb_keypoint* kps_un, const uint8_t* desc, const float* u_right, int n,
                                         const float* grid4, int max_dist, int chi2, const float* inv_level_sigma2,
                                         int n_levels, int32_t* best_idx, int32_t* best_dist)
{
    if (!m || nq < 0 || n < 0 || !grid4) return ORB_ERR_INVALID;
    if (nq > 0 && !best_idx) return ORB_ERR_INVALID;
    for (int i = 0; i < nq; i++) { best_idx[i] = -1; if (best_dist) best_dist[i] = 256; }
    if (nq == 0 || n == 0) return ORB_OK;
    if (!queries || !q_desc || !kps_un || !desc || (chi2 && !inv_level_sigma2)) return ORB_ERR_INVALID;
    if (nq > 65535 || n > 65535 || n_levels > 16) return ORB_ERR_UNSUPPORTED;
    ORB_HIP_TRY(hipSetDevice(m->device));
    MBuf* buf = m->init;
    int rc = proj_common(m, queries, q_desc, nullptr, nq, kps_un, desc, u_right, nullptr, n, grid4, chi2, inv_level_sigma2,
                         n_levels, buf);
    if (rc != ORB_OK) return rc;
    hipStream_t st = m->stream;
    int32_t* dBest = (int32_t*)buf[10].p;
    int32_t* dDist = (int32_t*)buf[11].p;
    hipLaunchKernelGGL(k_proj_best, dim3(nq), dim3(WAVE), 0, st, (const uint32_t*)buf[8].p, n, (const int*)buf[9].p, nq, max_dist,
                       dBest, dDist);
    ORB_HIP_TRY(hipGetLastError());
    ORB_HIP_TRY(hipMemcpyAsync(best_idx, dBest, (size_t)4 * nq, hipMemcpyDeviceToHost, st));
    if (best_dist) ORB_HIP_TRY(hipMemcpyAsync(best_dist, dDist, (size_t)4 * nq, hipMemcpyDeviceToHost, st));
    ORB_HIP_TRY(hipStreamSynchronize(st));
    return ORB_OK;
}

extern "C" int orb_match_projection(orb_matcher* m, int mode, const orb_proj_query* queries, const uint8_t* q_desc,
                                    const float* q_angle, int nq, const orb_keypoint* kps_un, const uint8_t* desc,
                                    const float* u_right, const uint8_t* occupied, int n, const float* grid4, float ratio,
                                    int max_dist, int check_ori, int32_t* match_cur, int* nmatches)
{
    if (!m || (mode != 0 && mode != 1) || nq < 0 || n < 0 || !nmatches || !grid4) return ORB_ERR_INVALID;
    *nmatches = 0;
    if (n > 0 && !match_cur) return ORB_ERR_INVALID;
    for (int i = 0; i < n; i++) match_cur[i] = -1;
    if (nq == 0 || n == 0) return ORB_OK;
    if (!queries || !q_desc || !kps_un || !desc || !occupied || (mode == 0 && check_ori && !q_angle))
        return ORB_ERR_INVALID;
    if (nq > 65535 || n > 65535) return ORB_ERR_UNSUPPORTED;
    ORB_HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    MBuf* buf = m->init;                                           // shares the SearchForInitialization scratch
    const size_t sz[12] = {sizeof(ProjQuery) * (size_t)nq, (size_t)32 * nq, (size_t)4 * nq, sizeof(orb_keypoint) * (size_t)n,
                           (size_t)32 * n, (size_t)4 * n, (size_t)n, (size_t)4 * n + 4, (size_t)4 * nq * n, (size_t)4 * nq,
                           (size_t)4 * n + 4, (size_t)4 * nq};
    int rc;
    for (int i = 0; i < 12; i++)
        if ((rc = buf[i].ensure(sz[i])) != ORB_OK) return rc;
    ProjQuery* dQ = (ProjQuery*)buf[0].p;
    uint8_t* dQD = (uint8_t*)buf[1].p;
    float* dQA = (float*)buf[2].p;
    orb_keypoint* dK = (orb_keypoint*)buf[3].p;
    uint8_t* dD = (uint8_t*)buf[4].p;
    float* dUR = (float*)buf[5].p;
    uint8_t* dOcc = (uint8_t*)buf[6].p;
    uint32_t* dKeys = (uint32_t*)buf[7].p;
    int* dNKeys = (int*)((uint8_t*)buf[7].p + (size_t)4 * n);
    uint32_t* dCand = (uint32_t*)buf[8].p;
    int* dCandCount = (int*)buf[9].p;
    int32_t* dMatch = (int32_t*)buf[10].p;
    int32_t* dNm = dMatch + n;
    uint32_t* dEvents = (uint32_t*)buf[11].p;
    ORB_HIP_TRY(hipMemcpyAsync(dQ, queries, sz[0], hipMemcpyHostToDevice, st));
    ORB_HIP_TRY(hipMemcpyAsync(dQD, q_desc, sz[1], hipMemcpyHostToDevice, st));
    if (q_angle) ORB_HIP_TRY(hipMemcpyAsync(dQA, q_angle, sz[2], hipMemcpyHostToDevice, st));
    ORB_HIP_TRY(hipMemcpyAsync(dK, kps_un, sz[3], hipMemcpyHostToDevice, st));
    ORB_HIP_TRY(hipMemcpyAsync(dD, desc, sz[4], hipMemcpyHostToDevice, st));
    if (u_right) ORB_HIP_TRY(hipMemcpyAsync(dUR, u_right, sz[5], hipMemcpyHostToDevice, st));
    ORB_HIP_TRY(hipMemcpyAsync(dOcc, occupied, sz[6], hipMemcpyHostToDevice, st));
    InitGrid g = {grid4[0], grid4[1], grid4[2], grid4[3]};
    hipLaunchKernelGGL(k_init_grid, dim3(1), dim3(256), 0, st, dK, n, g, 0, dKeys, dNKeys);
    hipLaunchKernelGGL(k_proj_candidates, dim3(nq), dim3(WAVE), 0, st, dQ, dQD, nq, dK, dD, u_right ? dUR : (const float*)nullptr, dKeys, dNKeys, g, dCand, n,
                       dCandCount, 0, Sigma2Tab{});
    hipLaunchKernelGGL(k_proj_resolve, dim3(1), dim3(WAVE), 0, st, mode, dQ, dQA, nq, dK, n, dCand, n, dCandCount, ratio,
                       max_dist, check_ori, dOcc, dMatch, dEvents, dNm);
    ORB_HIP_TRY(hipGetLastError());
    std::vector<int32_t> host((size_t)n + 1);
    ORB_HIP_TRY(hipMemcpyAsync(host.data(), dMatch, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, st));
    ORB_HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < n; i++) match_cur[i] = host[i];
    *nmatches = host[n];
    return ORB_OK;
}
