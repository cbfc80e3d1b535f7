// orb_stereo.hip -- "ORBmatcher stereo search" of BASELINE config 3: Frame::ComputeStereoMatches,
// reference src/Frame.cc:513-699, on gfx950, reading the two extractors' DEVICE-RESIDENT pyramids
// (the reference reads mpORBextractorLeft/Right->mvImagePyramid, :520,611,626,633).
//
// Frame.cc must link unchanged, so this is an ADDITIONAL entry point, parity-
// checked against the oracle's restatement; a maintainer may call it from ComputeStereoMatches.
//
//   k_stereo_rows      one workgroup per pair: per right keypoint its row band, octave and x as one record, and the
//                      reference's vRowIndices (:528-540) as a CSR over image rows
//   k_stereo_match     one wave64 per left keypoint: the right keypoints of its row (64 candidates per step; the minimum of
//                      (distance << 16 | index) IS "first minimum wins" whatever the order inside a row's list),
//                      Hamming coarse match, 11x11 SAD at 11 offsets from LDS
//                      patches (exact integers: the float patches of the reference hold integers),
//                      parabola fit and depth in float with the reference's operation order.
//   k_stereo_outliers  one workgroup per pair: median SAD by histogram selection, cut >= 1.5*1.4*median (:685-698).
#include <algorithm>
#include <vector>

#include "orb_extractor_internal.h"
#include "orb_wave.h"

#pragma clang fp contract(off)

#define WAVE 64

__device__ __forceinline__ unsigned st_umin_dpp(unsigned v)
{
#define ST_DPP(ctrl, rmask) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(v), ctrl, rmask, 0xf, false))
    ST_DPP(0x111, 0xf); ST_DPP(0x112, 0xf); ST_DPP(0x114, 0xf); ST_DPP(0x118, 0xf); ST_DPP(0x142, 0xa); ST_DPP(0x143, 0xc);
#undef ST_DPP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Right-image side of the coarse search, once per pair instead of once per LEFT keypoint.  Per right keypoint its row band
// (:528-540: r = 2 * scale[octave], rows floor(y - r) .. ceil(y + r)), octave and x as one 16-byte record, and
// vRowIndices (:528-540) of the pair as a CSR: rowStart[row] .. rowStart[row + 1] index the right keypoints whose band
// covers the row (u16 indices, any order inside a row).  A left keypoint then looks at the ~40 right keypoints of its row
// instead of testing the bands of all of them (1200 of 2000 at KITTI size with the level windows of the earlier version:
// three quarters of the search kernel's instructions).  One workgroup per pair, counters in LDS.
// dynamic LDS: cnt[nRows + 1] | fill[nRows] | part[1024 + 64] ints
__global__ __launch_bounds__(1024) void k_stereo_rows(const OrbGeom G, const orb_keypoint* __restrict__ kR0,
                                                      uint4* __restrict__ rec0, size_t recStride, int nR,
                                                      const int32_t* __restrict__ countsR, size_t stride, int nRows,
                                                      int maxBand, int* __restrict__ rowStart0,
                                                      unsigned short* __restrict__ rowList0, size_t listCap)
{
    extern __shared__ int rsm[];
    int* cnt = rsm;
    int* fill = cnt + nRows + 1;
    int* part = fill + nRows;
    const int pr = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
    const int Nr = countsR ? min(countsR[pr], (int)stride) : nR;
    uint4* recs = rec0 + recStride * pr;
    const orb_keypoint* kR = kR0 + stride * pr;
    int* rowStart = rowStart0 + (size_t)pr * (nRows + 1);
    unsigned short* rowList = rowList0 + listCap * pr;
    for (int y = tid; y <= nRows; y += T) cnt[y] = 0;
    for (int y = tid; y < nRows; y += T) fill[y] = 0;
    __syncthreads();
    // (a band longer than maxBand cannot come from a finite y: such a keypoint gets no rows -- the reference would index
    // vRowIndices out of range with it; the list holds maxBand entries per keypoint)
    for (int i = tid; i < Nr; i += T) {
        const orb_keypoint kp = kR[i];
        const int oc = min(max(kp.octave, 0), G.nlevels - 1);
        const float r = __fmul_rn(2.0f, G.L[oc].scale);                      // :531
        const int maxr = (int)ceilf(__fadd_rn(kp.y, r)), minr = (int)floorf(__fsub_rn(kp.y, r));
        const uint4 rc = make_uint4((unsigned)minr, (unsigned)maxr, (unsigned)kp.octave, __float_as_uint(kp.x));
        recs[i] = rc;                                              // (read back below by the thread that wrote it)
        const int lo = max((int)rc.x, 0), hi = min((int)rc.y, nRows - 1);
        if (hi - lo < maxBand)                              // (clamped values: no overflow whatever y was)
            for (int y = lo; y <= hi; y++) atomicAdd(&cnt[y], 1);
    }
    __syncthreads();
    {   // exclusive scan of cnt[0 .. nRows) in place, the total into cnt[nRows]
        const int C = (nRows + T - 1) / T;
        const int b = min(tid * C, nRows), e = min(b + C, nRows);
        int sum = 0;
        for (int y = b; y < e; y++) sum += cnt[y];
        // (a lane of wave 0 owns 16 consecutive partials: stored 17 apart, so that the 64 lanes of a read hit 64 different
        //  banks -- 16 apart they all met in two banks, and this scan was 5.7 us of the kernel's 11.6)
        part[tid + (tid >> 4)] = sum;
        __syncthreads();
        if (tid < WAVE) {
            int mine = 0;
            for (int j = 0; j < 16; j++) mine += part[17 * tid + j];
            const int incl = orb_wave_scan_incl(mine);
            int run = incl - mine;
            for (int j = 0; j < 16; j++) { const int v = part[17 * tid + j]; part[17 * tid + j] = run; run += v; }
            if (tid == WAVE - 1) cnt[nRows] = incl;
        }
        __syncthreads();
        int run = part[tid + (tid >> 4)];
        for (int y = b; y < e; y++) { const int v = cnt[y]; cnt[y] = run; run += v; }
    }
    __syncthreads();
    for (int y = tid; y <= nRows; y += T) rowStart[y] = cnt[y];
    for (int i = tid; i < Nr; i += T) {
        const uint4 rc = recs[i];
        const int lo = max((int)rc.x, 0), hi = min((int)rc.y, nRows - 1);
        if (hi - lo < maxBand)                              // (clamped values: no overflow whatever y was)
            for (int y = lo; y <= hi; y++) rowList[cnt[y] + atomicAdd(&fill[y], 1)] = (unsigned short)i;
    }
}

// grid (left keypoint, pair): pair p uses frames frame0 + p of the two pyramids and rows [p * stride, ...) of the
// keypoint / descriptor / result arrays; the keypoint counts come from the host (single pair) or from device arrays
// (batch: the extractors' d_counts, no host round trip between extraction and search).
__global__ __launch_bounds__(WAVE) void k_stereo_match(const OrbGeom G, const uint8_t* __restrict__ pyrL0, size_t slabL,
                                                       const uint8_t* __restrict__ pyrR0, size_t slabR,
                                                       const orb_keypoint* __restrict__ kL0,
                                                       const uint8_t* __restrict__ dL0, int nL,
                                                       const int32_t* __restrict__ countsL,
                                                       const orb_keypoint* __restrict__ kR0,
                                                       const uint8_t* __restrict__ dR0, int nR,
                                                       const int32_t* __restrict__ countsR, size_t stride, float maxD,
                                                       float mbf, float* __restrict__ uRight0,
                                                       float* __restrict__ depth0,
                                                       unsigned long long* __restrict__ pairs0,
                                                       const uint4* __restrict__ rec0,
                                                       size_t recStride, const int* __restrict__ rowStart0,
                                                       const unsigned short* __restrict__ rowList0, size_t listCap)
{
    __shared__ int IL[11][11];
    __shared__ int IR[11][21];
    __shared__ int part[11][11];
    const int iL = blockIdx.x, lane = threadIdx.x, pr = blockIdx.y;
    const int N = countsL ? min(countsL[pr], (int)stride) : nL;
    const uint8_t* pyrL = pyrL0 + slabL * pr;
    const uint8_t* pyrR = pyrR0 + slabR * pr;
    const orb_keypoint* kL = kL0 + stride * pr;
    const uint8_t* dL = dL0 + stride * pr * 32;
    const uint8_t* dR = dR0 + stride * pr * 32;
    float* uRight = uRight0 + stride * pr;
    float* depth = depth0 + stride * pr;
    unsigned long long* pairs = pairs0 + stride * pr;
    if (iL >= N) return;
    // (every left keypoint owns slot iL of `pairs`: ~0 = no match.  One atomic slot counter per pair used to serialise all
    // the waves of a batch on ONE cache line of the L2: 64 k returning atomics took longer than the search itself)
    if (lane == 0) { uRight[iL] = -1.0f; depth[iL] = -1.0f; pairs[iL] = ~0ull; }
    const orb_keypoint kpL = kL[iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const int nRows = G.L[0].h;
    const int row = (int)vL;                                       // vRowIndices[vL] (:560)
    if (vL < 0.0f || row >= nRows) return;
    const float minU = __fsub_rn(uL, maxD), maxU = uL;             // minD == 0 (:543-544, :565-566)
    if (maxU < 0) return;

    // ---- coarse match (:573-595): candidates in ascending iR, first minimum below TH_HIGH wins
    uint32_t dl[8];
    {
        const uint4 lo = reinterpret_cast<const uint4*>(dL + (size_t)iL * 32)[0], hi = reinterpret_cast<const uint4*>(dL + (size_t)iL * 32)[1];
        dl[0] = lo.x; dl[1] = lo.y; dl[2] = lo.z; dl[3] = lo.w; dl[4] = hi.x; dl[5] = hi.y; dl[6] = hi.z; dl[7] = hi.w;
    }
    // candidates = the right keypoints of the row (k_stereo_rows).  A wave is a chain of dependent global round trips and
    // little else, so two chunks of 64 list entries are requested at once, then their records, then the descriptors of
    // those that pass the octave and disparity tests; every lane keeps its own best (one wave reduction at the end).
    const uint4* recs = rec0 + recStride * pr;
    const unsigned short* rowList = rowList0 + listCap * pr;
    int s0, s1;
    {
        const int* rs = rowStart0 + (size_t)pr * (nRows + 1) + row;
        s0 = rs[0]; s1 = rs[1];
    }
    unsigned best = 0xFFFFFFFFu;
    float bestX = 0.0f;                                            // x of this lane's best candidate (its record is at hand)
    for (int base = s0; base < s1; base += 2 * WAVE) {
        int idx[2];
#pragma unroll
        for (int k = 0; k < 2; k++) idx[k] = rowList[min(base + k * WAVE + lane, s1 - 1)];
        uint4 rc[2];
#pragma unroll
        for (int k = 0; k < 2; k++) rc[k] = recs[idx[k]];
        bool cand[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int oct = (int)rc[k].z;
            const float x = __uint_as_float(rc[k].w);
            cand[k] = base + k * WAVE + lane < s1 && oct >= levelL - 1 && oct <= levelL + 1 && x >= minU && x <= maxU;
        }
        uint4 lo[2], hi[2];
#pragma unroll
        for (int k = 0; k < 2; k++)
            if (cand[k]) {
                const uint4* d = reinterpret_cast<const uint4*>(dR + (size_t)idx[k] * 32);
                lo[k] = d[0]; hi[k] = d[1];
            }
#pragma unroll
        for (int k = 0; k < 2; k++)
            if (cand[k]) {
                const int dist = __popc(dl[0] ^ lo[k].x) + __popc(dl[1] ^ lo[k].y) + __popc(dl[2] ^ lo[k].z) + __popc(dl[3] ^ lo[k].w) +
                                 __popc(dl[4] ^ hi[k].x) + __popc(dl[5] ^ hi[k].y) + __popc(dl[6] ^ hi[k].z) + __popc(dl[7] ^ hi[k].w);
                const unsigned key = ((unsigned)dist << 16) | (unsigned)idx[k];
                if (key < best) { best = key; bestX = __uint_as_float(rc[k].w); }
            }
    }
    const unsigned mineBest = best;
    best = st_umin_dpp(best);
    const int bestDist = (best == 0xFFFFFFFFu) ? 100 : min(100, (int)(best >> 16));   // init TH_HIGH, strict <
    if (!(bestDist < 75) || best == 0xFFFFFFFFu) return;                              // thOrbDist (:518, :599)
    const int bestIdxR = (int)(best & 0xFFFFu);

    // ---- SAD refinement on the pyramid level of the LEFT keypoint (:601-648)
    const OrbLevelGeom& Lv = G.L[levelL];
    // mvKeysRight[bestIdxR].pt.x (:608) from the lane that holds the winner (keys are unique): no load
    const unsigned long long owner = __ballot(mineBest == best);
    const float uR0 = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(bestX), (int)__builtin_ctzll(owner)));
    (void)bestIdxR;
    const float sf = Lv.invScale;
    const float scaleduL = roundf(__fmul_rn(kpL.x, sf)), scaledvL = roundf(__fmul_rn(kpL.y, sf));
    const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
    const int w = 5, Lr = 5;
    const float iniu = __fsub_rn(__fadd_rn(scaleduR0, (float)Lr), (float)w);          // sic (:624)
    const float endu = __fadd_rn(__fadd_rn(__fadd_rn(scaleduR0, (float)Lr), (float)w), 1.0f);
    if (iniu < 0 || endu >= (float)Lv.w) return;
    const int y0 = (int)__fsub_rn(scaledvL, (float)w), xL0 = (int)__fsub_rn(scaleduL, (float)w);
    const int xR0 = (int)__fsub_rn(__fsub_rn(scaleduR0, (float)Lr), (float)w);        // leftmost column of the strip
    const uint8_t* imL = pyrL + Lv.pyrOff;
    const uint8_t* imR = pyrR + Lv.pyrOff;
    // the reference would read out of the Mat here; inside the supported envelope keypoints sit >= 19 px inside
    if (y0 < 0 || y0 + 11 > Lv.h || xL0 < 0 || xL0 + 11 > Lv.w || xR0 < 0 || xR0 + 21 > Lv.w) return;
    for (int i = lane; i < 121; i += WAVE) IL[i / 11][i % 11] = imL[(size_t)(y0 + i / 11) * Lv.pitch + xL0 + i % 11];
    for (int i = lane; i < 231; i += WAVE) IR[i / 21][i % 21] = imR[(size_t)(y0 + i / 21) * Lv.pitch + xR0 + i % 21];
    __syncthreads();
    const int cL = IL[5][5];
    for (int i = lane; i < 121; i += WAVE) {
        const int inc = i / 11, dy = i % 11;                     // inc = incR + 5
        const int cR = IR[5][inc + 5];
        int s = 0;
#pragma unroll
        for (int dx = 0; dx < 11; dx++) s += abs((IL[dy][dx] - cL) - (IR[dy][dx + inc] - cR));
        part[inc][dy] = s;
    }
    __syncthreads();
    // the 11 SADs by 11 lanes, their first minimum (:641-645: strict <, so the lower offset wins ties) as one wave minimum
    // of (SAD << 4 | offset), its neighbours by v_readlane -- one lane summing 121 values serially was a third of the
    // kernel's instructions (the kernel is issue-bound: 64 k waves per 32 KITTI pairs)
    int mySad = 0;
    if (lane < 11) {
#pragma unroll
        for (int dy = 0; dy < 11; dy++) mySad += part[lane][dy];
    }
    const unsigned kb = st_umin_dpp(lane < 11 ? ((unsigned)mySad << 4) | (unsigned)lane : 0xFFFFFFFFu);
    const int bestinc = (int)(kb & 15u), bestSad = (int)(kb >> 4);
    const int bestincR = bestinc - 5;
    if (bestincR == -Lr || bestincR == Lr) return;                 // :651
    const float dist1 = (float)__builtin_amdgcn_readlane(mySad, bestinc - 1), dist2 = (float)bestSad,
                dist3 = (float)__builtin_amdgcn_readlane(mySad, bestinc + 1);
    if (lane != 0) return;
    const float deltaR = __fdiv_rn(__fsub_rn(dist1, dist3),
                                   __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2))));
    if (deltaR < -1 || deltaR > 1) return;
    float bestuR = __fmul_rn(Lv.scale, __fadd_rn(__fadd_rn(scaleduR0, (float)bestincR), deltaR));
    float disparity = __fsub_rn(uL, bestuR);
    if (disparity >= 0.0f && disparity < maxD) {
        if (disparity <= 0) {
            disparity = (float)0.01;
            bestuR = (float)((double)uL - 0.01);
        }
        depth[iL] = __fdiv_rn(mbf, disparity);
        uRight[iL] = bestuR;
        pairs[iL] = ((unsigned long long)(unsigned)bestSad << 32) | (unsigned)iL;
    }
}

// one workgroup per pair: median of the matched keypoints' SAD and the cut >= 1.5 * 1.4 * median (:685-698).  The
// reference sorts (SAD, index) pairs only to read the element in the middle: its SAD is the (n/2)-th order statistic, which
// two 256-bin histograms find (high byte, then low byte inside the bin the middle falls into; a SAD is at most
// 121 * 510 < 2^16) -- six barriers instead of the 66 steps of a sorting network over ~2000 slots, and no LDS or
// global-memory sort whatever the number of keypoints.  Unmatched slots are ~0.
__global__ __launch_bounds__(1024) void k_stereo_outliers(const unsigned long long* __restrict__ pairs0, size_t stride, int nL,
                                                          const int32_t* __restrict__ countsL,
                                                          float* __restrict__ uRight0, float* __restrict__ depth0)
{
    __shared__ int hist[256];
    __shared__ int sel[3];                                         // matched count, chosen bin, rank inside it
    const int pr = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
    const unsigned long long* gp = pairs0 + stride * pr;
    float* uRight = uRight0 + stride * pr;
    float* depth = depth0 + stride * pr;
    const int nAll = countsL ? min(countsL[pr], (int)stride) : nL;
    if (nAll <= 0) return;
    int median = 0;
    for (int pass = 0; pass < 2; pass++) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const int wantHi = pass ? sel[1] : 0;
        for (int i = tid; i < nAll; i += T) {
            const unsigned long long e = gp[i];
            if (e == ~0ull) continue;
            const unsigned sad = min((unsigned)(e >> 32), 0xFFFFu);
            if (pass == 0) atomicAdd(&hist[sad >> 8], 1);
            else if ((int)(sad >> 8) == wantHi) atomicAdd(&hist[sad & 0xFFu], 1);
        }
        __syncthreads();
        if (tid < 64) {                                            // one wave: the bin in which the wanted rank falls
            const int c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const int mine = c0 + c1 + c2 + c3;
            const int incl = orb_wave_scan_incl(mine);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            const int want = pass ? sel[2] : total / 2;            // rank (0-based) of the element in the middle (:687)
            const int before = incl - mine;
            if (total > 0 && want >= before && want < incl) {      // exactly one lane
                int r = want - before, bin = 4 * tid;
                if (r >= c0) { r -= c0; bin++; if (r >= c1) { r -= c1; bin++; if (r >= c2) { r -= c2; bin++; } } }
                sel[1] = bin;
                sel[2] = r;
            }
            if (tid == 0 && pass == 0) sel[0] = total;
        }
        __syncthreads();
        if (sel[0] == 0) return;                                   // reference: UB on the empty vector (:686)
        median = pass ? (median << 8) | sel[1] : sel[1];
        __syncthreads();
    }
    const float thDist = __fmul_rn(1.5f * 1.4f, (float)median);
    for (int i = tid; i < nAll; i += T) {
        const unsigned long long e = gp[i];
        if (e == ~0ull) continue;
        if (!((float)(int)(e >> 32) < thDist)) {
            uRight[i] = -1.0f;                                     // slot i = left keypoint i
            depth[i] = -1.0f;
        }
    }
}

static int stereo_check(orb_extractor* left, orb_extractor* right)
{
    if (left->rows == 0 || left->rows != right->rows || left->cols != right->cols ||
        left->prm.nlevels != right->prm.nlevels || left->prm.scale_factor != right->prm.scale_factor) {
        orb_set_error("stereo: both extractors must have processed images of the same size with the same pyramid");
        return ORB_ERR_INVALID;
    }
    if (left->device != right->device) return ORB_ERR_UNSUPPORTED;
    return ORB_OK;
}

// launches both kernels for nPairs pairs; counts from the host (cL/cR null) or from the device
static int stereo_launch(orb_extractor* left, orb_extractor* right, int frameL, int frameR, int nPairs, size_t stride,
                         const orb_keypoint* kL, const uint8_t* dL, int nL, const int32_t* cL, const orb_keypoint* kR,
                         const uint8_t* dR, int nR, const int32_t* cR, float mb, float mbf, float* uR, float* dep)
{
    ORB_HIP_TRY(hipSetDevice(left->device));
    int rc;
    const size_t perPair = stride;                                  // (SAD, index) slots per pair
    // scratch: pairs[perPair * nPairs] (u64, slot iL of pair p = left keypoint iL) | rec[recStride * nPairs] (uint4)
    //          | rowStart[nPairs][nRows + 1] (int) | rowList[nPairs][listCap] (u16)
    const size_t nSlots = perPair * nPairs;
    const size_t recStride = std::max<size_t>(stride, (size_t)(cR ? 0 : nR));       // single pair: stride is the LEFT count
    const size_t nRecs = recStride * nPairs;
    const int nRows = left->G.L[0].h;
    // rows a right keypoint's band can cover: floor(y - r) .. ceil(y + r), r = 2 * scale <= 2 * scale of the last level
    const int maxBand = std::min(nRows, 2 * (int)std::ceil(2.0 * left->G.L[left->G.nlevels - 1].scale) + 3);
    const size_t listCap = (recStride * (size_t)maxBand + 7) & ~(size_t)7;
    const size_t rowsLds = ((size_t)2 * nRows + 1 + 1024 + 64) * 4;
    if (rowsLds > 64 * 1024) { orb_set_error("stereo: %d image rows exceed the row table's LDS budget", nRows); return ORB_ERR_UNSUPPORTED; }
    if ((rc = left->dStereo.ensure((size_t)8 * (nSlots + 1) + (size_t)16 * nRecs + (size_t)4 * nPairs * (nRows + 1) + (size_t)2 * nPairs * listCap + 64)) != ORB_OK)
        return rc;
    hipStream_t st = left->stream;
    ORB_HIP_TRY(hipEventRecord(left->waitEv, right->stream));      // the right pyramid must be complete
    ORB_HIP_TRY(hipStreamWaitEvent(st, left->waitEv, 0));
    unsigned long long* pairs = (unsigned long long*)left->dStereo.p;
    uint4* rec = (uint4*)(pairs + ((nSlots + 1) & ~(size_t)1));           // 16-byte aligned
    int* rowStart = (int*)(rec + nRecs);
    unsigned short* rowList = (unsigned short*)(rowStart + (size_t)nPairs * (nRows + 1));
    const float maxD = mbf / mb;                                   // :546
    const int gridX = cL ? (int)stride : nL;
    hipLaunchKernelGGL(k_stereo_rows, dim3(nPairs), dim3(1024), rowsLds, st, left->G, kR, rec, recStride, nR, cR, stride, nRows, maxBand,
                       rowStart, rowList, listCap);
    hipLaunchKernelGGL(k_stereo_match, dim3(gridX, nPairs), dim3(WAVE), 0, st, left->G,
                       (const uint8_t*)left->dPyr.p + left->pyrSlab * frameL, left->pyrSlab,
                       (const uint8_t*)right->dPyr.p + right->pyrSlab * frameR, right->pyrSlab, kL, dL, nL, cL, kR, dR, nR, cR,
                       stride, maxD, mbf, uR, dep, pairs, rec, recStride, rowStart, rowList, listCap);
    hipLaunchKernelGGL(k_stereo_outliers, dim3(nPairs), dim3(1024), 0, st, pairs, perPair, nL, cL, uR, dep);
    ORB_HIP_TRY(hipGetLastError());
    // The search reads the RIGHT handle's pyramid (and the right keypoints / descriptors) on the LEFT handle's stream: whatever
    // the caller issues next on the right handle -- typically the next extraction, which overwrites all three -- is ordered
    // behind it.  (Without this edge a pipelined caller raced the search of step k against the right extraction of step k + 1.)
    if (right->stream != st) {
        ORB_HIP_TRY(hipEventRecord(right->waitEv, st));
        ORB_HIP_TRY(hipStreamWaitEvent(right->stream, right->waitEv, 0));
    }
    return ORB_OK;
}

extern "C" int orb_stereo_match_device(orb_extractor* left, orb_extractor* right, int frame_l, int frame_r,
                                       const orb_keypoint* d_kps_l, const uint8_t* d_desc_l, int n_l,
                                       const orb_keypoint* d_kps_r, const uint8_t* d_desc_r, int n_r, float mb,
                                       float mbf, float* d_u_right, float* d_depth)
{
    if (!left || !right || n_l < 0 || n_r < 0) return ORB_ERR_INVALID;
    if (n_l == 0) return ORB_OK;
    if (!d_kps_l || !d_desc_l || !d_u_right || !d_depth || (n_r > 0 && (!d_kps_r || !d_desc_r))) return ORB_ERR_INVALID;
    int rc;
    if ((rc = stereo_check(left, right)) != ORB_OK) return rc;
    frame_l -= left->frameBase;
    frame_r -= right->frameBase;
    if (frame_l < 0 || frame_l >= left->lastFrames || frame_r < 0 || frame_r >= right->lastFrames) return ORB_ERR_INVALID;
    if (n_l > 65535 || n_r > 65535) return ORB_ERR_UNSUPPORTED;
    return stereo_launch(left, right, frame_l, frame_r, 1, (size_t)n_l, d_kps_l, d_desc_l, n_l, nullptr, d_kps_r, d_desc_r, n_r,
                         nullptr, mb, mbf, d_u_right, d_depth);
}

// Batch: pair p = frames (first_frame_l + p, first_frame_r + p) of the two handles' last batches; keypoints, descriptors
// and results of pair p at rows [p * cap, (p + 1) * cap) of the arrays the extractors wrote (orb_extract_batch_device
// layout), keypoint counts read on the DEVICE from the extractors' d_counts.  One launch for all pairs.
extern "C" int orb_stereo_match_batch_device(orb_extractor* left, orb_extractor* right, int first_frame_l, int first_frame_r,
                                             int n_pairs, const orb_keypoint* d_kps_l, const uint8_t* d_desc_l,
                                             const int32_t* d_counts_l, const orb_keypoint* d_kps_r,
                                             const uint8_t* d_desc_r, const int32_t* d_counts_r, int cap, float mb, float mbf,
                                             float* d_u_right, float* d_depth)
{
    if (!left || !right || n_pairs < 0 || cap <= 0) return ORB_ERR_INVALID;
    if (n_pairs == 0) return ORB_OK;
    if (!d_kps_l || !d_desc_l || !d_counts_l || !d_kps_r || !d_desc_r || !d_counts_r || !d_u_right || !d_depth) return ORB_ERR_INVALID;
    int rc;
    if ((rc = stereo_check(left, right)) != ORB_OK) return rc;
    first_frame_l -= left->frameBase;
    first_frame_r -= right->frameBase;
    if (first_frame_l < 0 || first_frame_l + n_pairs > left->lastFrames || first_frame_r < 0 ||
        first_frame_r + n_pairs > right->lastFrames)
        return ORB_ERR_INVALID;
    if (cap > 65535 || n_pairs > 65535) return ORB_ERR_UNSUPPORTED;
    return stereo_launch(left, right, first_frame_l, first_frame_r, n_pairs, (size_t)cap, d_kps_l, d_desc_l, 0, d_counts_l, d_kps_r,
                         d_desc_r, 0, d_counts_r, mb, mbf, d_u_right, d_depth);
}

extern "C" int orb_stereo_match(orb_extractor* left, orb_extractor* right, const orb_keypoint* kps_l,
                                const uint8_t* desc_l, int n_l, const orb_keypoint* kps_r, const uint8_t* desc_r,
                                int n_r, float mb, float mbf, float* u_right, float* depth)
{
    if (!left || !right || n_l < 0 || n_r < 0) return ORB_ERR_INVALID;
    if (n_l == 0) return ORB_OK;
    if (!kps_l || !desc_l || !u_right || !depth) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(left->device));
    const size_t szl = sizeof(orb_keypoint) * (size_t)n_l, szr = sizeof(orb_keypoint) * (size_t)std::max(n_r, 1);
    int rc;
    if ((rc = left->dStereoIn.ensure(szl + szr + (size_t)32 * n_l + (size_t)32 * std::max(n_r, 1) + (size_t)8 * n_l + 64)) != ORB_OK)
        return rc;
    uint8_t* base = (uint8_t*)left->dStereoIn.p;
    orb_keypoint* dKl = (orb_keypoint*)base;
    orb_keypoint* dKr = (orb_keypoint*)(base + szl);
    uint8_t* dDl = base + szl + szr;
    uint8_t* dDr = dDl + (size_t)32 * n_l;
    float* dU = (float*)(((uintptr_t)(dDr + (size_t)32 * std::max(n_r, 1)) + 15) & ~(uintptr_t)15);
    float* dZ = dU + n_l;
    hipStream_t st = left->stream;
    ORB_HIP_TRY(hipMemcpyAsync(dKl, kps_l, szl, hipMemcpyHostToDevice, st));
    ORB_HIP_TRY(hipMemcpyAsync(dDl, desc_l, (size_t)32 * n_l, hipMemcpyHostToDevice, st));
    if (n_r > 0) {
        ORB_HIP_TRY(hipMemcpyAsync(dKr, kps_r, sizeof(orb_keypoint) * (size_t)n_r, hipMemcpyHostToDevice, st));
        ORB_HIP_TRY(hipMemcpyAsync(dDr, desc_r, (size_t)32 * n_r, hipMemcpyHostToDevice, st));
    }
    rc = orb_stereo_match_device(left, right, 0, 0, dKl, dDl, n_l, dKr, dDr, n_r, mb, mbf, dU, dZ);
    if (rc != ORB_OK) return rc;
    ORB_HIP_TRY(hipMemcpyAsync(u_right, dU, (size_t)4 * n_l, hipMemcpyDeviceToHost, st));
    ORB_HIP_TRY(hipMemcpyAsync(depth, dZ, (size_t)4 * n_l, hipMemcpyDeviceToHost, st));
    ORB_HIP_TRY(hipStreamSynchronize(st));
    return ORB_OK;
}
