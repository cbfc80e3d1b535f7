// orb_matcher.hip -- matcher half of include/orb_hip.h on gfx950.
//
//   orb_hamming / orb_three_maxima      host restatements of the two tiny static helpers
//                                       (reference src/ORBmatcher.cc:46-63, :1663-1707)
//   k_bow_assign                        synthetic 2-level vocabulary descent
//   k_match_bow<KK>                     SearchByBoW(KF,F) / SearchByBoW(KF,KF)  (:552-687, :690-832)
//   k_init_candidates / k_init_resolve  SearchForInitialization + Frame grid     (:1055-1180; Frame.cc:348-422)
//
// SearchByBoW: one 256-thread workgroup per (keyframe, frame) pair.  Both feature vectors are
// rebuilt in LDS as CSR from a per-feature vocabulary-node array; vocabulary nodes are independent
// (a Frame feature belongs to exactly one node), so the four waves take nodes round-robin.  Inside a
// node the keyframe loop is serial (greedy "already matched" rule, :607); the lanes of the wave hold
// the Frame-side descriptors of the node and find best / second-best Hamming distance with popcount
// + cross-lane min.  Rotation histogram, ComputeThreeMaxima and the top-3 filter run in the same launch.
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "orb_block_sort.h"
#include "orb_matcher_internal.h"
#include "orb_bow_device.h"

#pragma clang fp contract(off)

#define MAX_NODES 1024
#define NODE_NONE 0xFFFFu

// ------------------------------------------------------------------ host helpers
extern "C" int orb_hamming(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y;
        std::memcpy(&x, a + 4 * i, 4);
        std::memcpy(&y, b + 4 * i, 4);
        dist += __builtin_popcount(x ^ y);
    }
    return dist;
}

static inline void three_maxima(const int* counts, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < HISTO_LENGTH; i++) {
        const int s = counts[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

extern "C" void orb_three_maxima(const int32_t* counts30, int32_t* ind3)
{
    int a, b, c;
    three_maxima(counts30, a, b, c);
    ind3[0] = a; ind3[1] = b; ind3[2] = c;
}

// ------------------------------------------------------------------ vocabulary stand-in
__global__ __launch_bounds__(256) void k_bow_assign(const uint8_t* __restrict__ desc, const int32_t* __restrict__ counts,
                                                    int cap, const uint8_t* __restrict__ cent,
                                                    uint16_t* __restrict__ nodeOf)
{
    __shared__ uint32_t c[110 * 8];
    for (int i = threadIdx.x; i < 110 * 8; i += blockDim.x) c[i] = reinterpret_cast<const uint32_t*>(cent)[i];
    __syncthreads();
    const int f = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const size_t row = (size_t)f * cap + i;
    if (i >= counts[f]) { nodeOf[row] = NODE_NONE; return; }
    uint32_t d[8];
    load_desc(desc + row * 32, d);
    int c1 = 0, b1 = 257;
    for (int k = 0; k < 10; k++) {
        const int h = hamming8(d, c + 8 * k);
        if (h < b1) { b1 = h; c1 = k; }                 // first minimum wins
    }
    int c2 = 0, b2 = 257;
    for (int k = 0; k < 10; k++) {
        const int h = hamming8(d, c + 8 * (10 + 10 * c1 + k));
        if (h < b2) { b2 = h; c2 = k; }
    }
    nodeOf[row] = (uint16_t)(11 + 10 * c1 + c2);
}

// ------------------------------------------------------------------ SearchByBoW
struct BowSide {
    const uint8_t* desc;      // [n][32]
    const float* angle;       // angle of feature i at angle[i*angleStride]
    int angleStride;
    const uint8_t* valid;     // or nullptr
    const uint16_t* nodeOf;   // vocabulary node per feature, NODE_NONE = not in the feature vector
    int n;                    // -1: the pair is invalid (index outside the store, count outside [0, cap])
    // the side's feature vector as CSR, built once per frame by k_build_csr (nullptr: built here, once per pair)
    const uint32_t* csrKeys;  // [n] (node << 16 | index), grouped by node, ascending index inside a node
    const uint16_t* csrStart; // [nNodes]
    const uint16_t* csrCnt;   // [nNodes]
};

// One side's feature vector as CSR in LDS: keys[start[node] .. start[node] + cnt[node]) = (node << 16 | index) with
// ascending index inside a node (DBoW2's push order).  A counting sort -- six barrier steps instead of the ~45 of a
// bitonic network over ~1000 keys, which used to be 40 % of this latency-bound kernel:
//   1 histogram of the nodes with LDS atomics (the returned slot is an arbitrary order inside the node)
//   2 exclusive scan of the histogram by one wave (DPP), groups by descending size  -> start[], cnt[]
//   3 scatter the indices to tmp[start[node] + slot]
//   4 rank every index inside its node's segment (segments are ~10 long; O(m) per feature) -> keys[]
// Features whose node is not in [0, nNodes) take no part (they are in no feature vector).
#define ORB_CSR_MAXPT 8        // features per thread: cap <= 8192, 1024 threads
__device__ void build_csr(const uint16_t* __restrict__ nodeOf, int n, int nNodes, uint32_t* keys, uint32_t* tmp,
                          uint32_t* cntw, uint16_t* start, uint16_t* cnt, bool bySize)
{
    const int T = blockDim.x, tid = threadIdx.x;
    for (int t = tid; t < nNodes; t += T) cntw[t] = 0;
    __syncthreads();
    unsigned nd[ORB_CSR_MAXPT], slot[ORB_CSR_MAXPT];
#pragma unroll
    for (int m = 0; m < ORB_CSR_MAXPT; m++) {
        const int i = tid + m * T;
        nd[m] = 0xFFFFu;
        slot[m] = 0;
        if (i < n) {
            const unsigned v = gload(nodeOf + i);
            if (v < (unsigned)nNodes) { nd[m] = v; slot[m] = atomicAdd(&cntw[v], 1u); }
        }
    }
    __syncthreads();
    // The node groups are laid out by DESCENDING size (ties: ascending node), not by node: start[] / cnt[] stay indexed by node,
    // so no consumer sees the order -- except k_match_bow_query's phase 1, whose lanes walk a keyframe's positions 64 at a time
    // and run as long as the LARGEST query-side node among them: nodes that are large in the keyframe are large in the query
    // too, so neighbours in this order have similar scan lengths (round 4).  Only where that kernel will read the result
    // (bySize: the store's CSR with the node-sorted descriptor copy); elsewhere plain node order (cnt[rank] = rank) -- the
    // ranking and its barriers doubled k_build_csr on a 512-frame batch (8.4 -> 17.2 us).
    for (int t = tid; t < nNodes; t += T) {            // rank of node t by (count descending, node ascending) -> cnt[rank] = t
        if (!bySize) { cnt[t] = (uint16_t)t; continue; }
        const unsigned c = cntw[t];
        int r = 0, u = 0;
        for (; u + 8 <= nNodes; u += 8) {              // 8 independent broadcast reads in flight (one dependent read per
            unsigned v[8];                             // iteration is a chain of LDS round trips: +3 us on a one-frame launch)
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = cntw[u + k];
#pragma unroll
            for (int k = 0; k < 8; k++) r += (v[k] > c) || (v[k] == c && u + k < t);
        }
        for (; u < nNodes; u++) {
            const unsigned cu = cntw[u];
            r += (cu > c) || (cu == c && u < t);
        }
        cnt[r] = (uint16_t)t;
    }
    __syncthreads();
    if (tid < 64) {                                    // exclusive scan over the nodes in that order: lane owns a contiguous chunk
        const int C = (nNodes + 63) / 64;
        const int b = min(tid * C, nNodes), e = min(b + C, nNodes);
        int sum = 0;
        for (int r = b; r < e; r++) sum += (int)cntw[cnt[r]];
        const int incl = orb_wave_scan_incl(sum);
        int run = incl - sum;
        for (int r = b; r < e; r++) {
            const int t = cnt[r];
            start[t] = (uint16_t)run;
            run += (int)cntw[t];
        }
    }
    __syncthreads();
    for (int t = tid; t < nNodes; t += T) cnt[t] = (uint16_t)cntw[t];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < ORB_CSR_MAXPT; m++)
        if (nd[m] != 0xFFFFu) tmp[start[nd[m]] + slot[m]] = (uint32_t)(tid + m * T);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < ORB_CSR_MAXPT; m++)
        if (nd[m] != 0xFFFFu) {
            const unsigned i = (unsigned)(tid + m * T);
            const int s0 = start[nd[m]], c = cnt[nd[m]];
            int rank = 0;
            for (int j = 0; j < c; j++) rank += tmp[s0 + j] < i;
            keys[s0 + rank] = (nd[m] << 16) | i;
        }
    __syncthreads();
}

// the CSR of one side in LDS: copied when the feature store holds it (one coalesced read), else built
__device__ __forceinline__ void load_or_build_csr(const BowSide& S, int nNodes, uint32_t* keys, uint32_t* tmp, uint32_t* cntw,
                                                  uint16_t* start, uint16_t* cnt)
{
    if (S.csrKeys) {
        for (int i = threadIdx.x; i < S.n; i += blockDim.x) keys[i] = gload(S.csrKeys + i);
        for (int t = threadIdx.x; t < nNodes; t += blockDim.x) { start[t] = gload(S.csrStart + t); cnt[t] = gload(S.csrCnt + t); }
        __syncthreads();
    } else {
        build_csr(S.nodeOf, S.n, nNodes, keys, tmp, cntw, start, cnt, false);
    }
}

// one workgroup per frame of a feature store: its CSR, kept in HBM next to node_of
__global__ __launch_bounds__(1024) void k_build_csr(const uint16_t* __restrict__ nodeOf, const int32_t* __restrict__ counts,
                                                    int cap, int nNodes, uint32_t* __restrict__ keysOut,
                                                    uint16_t* __restrict__ startOut, uint16_t* __restrict__ cntOut,
                                                    const uint8_t* __restrict__ descIn, uint8_t* __restrict__ descOut)
{
    extern __shared__ uint32_t csm[];
    uint32_t* keys = csm;
    uint32_t* tmp = keys + cap;
    uint32_t* cntw = tmp + cap;
    uint16_t* start = reinterpret_cast<uint16_t*>(cntw + nNodes);
    uint16_t* cnt = start + nNodes;
    const int f = blockIdx.x;
    const int n = min(max(counts[f], 0), cap);
    for (int i = threadIdx.x; i < cap; i += blockDim.x) keys[i] = 0xFFFFFFFFu;
    build_csr(nodeOf + (size_t)f * cap, n, nNodes, keys, tmp, cntw, start, cnt, descOut != nullptr);
    for (int i = threadIdx.x; i < cap; i += blockDim.x) keysOut[(size_t)f * cap + i] = keys[i];
    for (int t = threadIdx.x; t < nNodes; t += blockDim.x) {
        startOut[(size_t)f * nNodes + t] = start[t];
        cntOut[(size_t)f * nNodes + t] = cnt[t];
    }
    if (descOut) {                                             // the descriptors in key order (orb_featstore.csr_desc)
        const uint4* src = reinterpret_cast<const uint4*>(descIn) + (size_t)f * cap * 2;
        uint4* dst = reinterpret_cast<uint4*>(descOut) + (size_t)f * cap * 2;
        for (int i = threadIdx.x; i < 2 * cap; i += blockDim.x) {
            const uint32_t k = keys[i >> 1];
            if (k != 0xFFFFFFFFu) dst[i] = src[2 * (k & 0xFFFFu) + (i & 1)];
        }
    }
}

// minimum over the 16 lanes of a DPP row, in every lane of the row (row_ror 8, 4, 2, 1)
__device__ __forceinline__ unsigned row16_umin_all(unsigned v)
{
    ORB_DPP_STEP_UMIN(v, 0x128, 0xf);
    ORB_DPP_STEP_UMIN(v, 0x124, 0xf);
    ORB_DPP_STEP_UMIN(v, 0x122, 0xf);
    ORB_DPP_STEP_UMIN(v, 0x121, 0xf);
    return v;
}

// A node with at most 16 NB features of the frame (the usual case: ~10 features per node and frame; NB = 2 covers the
// nodes of up to 32): the wave works as 4 rows of 16 lanes, every row holding the node's B features (NB per lane), and
// takes 4 A features at a time -- one per row, their distances computed side by side.  Best and second best of all four
// rows come from one pair of row reductions (row_ror: every lane of a row ends up with its row's minimum) and every lane
// runs the acceptance test (:625-627 / :772-775) for its own row.  The greedy "already taken" rule couples the rows only
// when the column a row takes is the best or second-best column of a LATER row of the group (a column that is neither
// changes neither): two ballots tell; if not -- the usual case -- all rows commit at once, else the rows are resolved one
// after the other as the reference's loop does.
template <bool KK, int NB>
__device__ __forceinline__ void match_small_node(const BowSide& A, const BowSide& B, int na, int nb, int a0, int b0,
                                                 const uint32_t* keysA, const uint32_t* keysB, const uint8_t* okALds,
                                                 uint8_t* takenB, int16_t* res, float ratio, int lane)
{
    constexpr int GA = NB == 1 ? 4 : 2;                        // groups of 4 A features whose descriptors are requested together
    const int rowI = lane >> 4, col = lane & 15;
    uint32_t dB[NB][8];
    int jB[NB];
    bool taken[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) {
#pragma unroll
        for (int w = 0; w < 8; w++) dB[k][w] = 0;
        jB[k] = -1;
        taken[k] = true;                                       // columns >= nb count as taken (distance 256)
        if (col + 16 * k < nb) {
            jB[k] = (int)(keysB[b0 + col + 16 * k] & 0xFFFFu);
            load_desc(B.desc + (size_t)jB[k] * 32, dB[k]);
            taken[k] = takenB[jB[k]] != 0;
        }
    }
    auto reduce2 = [&](const uint32_t* dAg, unsigned& r1, unsigned& r2) {
        unsigned v[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            unsigned d = 0;
#pragma unroll
            for (int w = 0; w < 8; w++) d += __popc(dB[k][w] ^ dAg[w]);
            v[k] = ((taken[k] ? 256u : d) << 16) | (unsigned)(col + 16 * k);
        }
        const unsigned lo = NB == 1 ? v[0] : min(v[0], v[NB - 1]), hi = NB == 1 ? 0xFFFFFFFFu : max(v[0], v[NB - 1]);
        r1 = row16_umin_all(lo);
        r2 = row16_umin_all(lo == r1 ? hi : lo);
    };
    for (int abase = 0; abase < na; abase += 4 * GA) {
        uint32_t dA[GA][8];
        int iAg[GA];
        bool okAg[GA];
#pragma unroll
        for (int g = 0; g < GA; g++) {
            // branch-free: rows past the node's last A feature load its last one again and are masked by okAg
            const int p = abase + 4 * g + rowI;
            iAg[g] = (int)(keysA[a0 + min(p, na - 1)] & 0xFFFFu);
            okAg[g] = p < na && okALds[iAg[g]] != 0;           // :590-595
            load_desc(A.desc + (size_t)iAg[g] * 32, dA[g]);
        }
#pragma unroll
        for (int g = 0; g < GA; g++) {
            if (abase + 4 * g >= na) continue;
            const int iA = iAg[g];
            // No free column within TH_LOW of any of the four A features: nothing can be accepted, nothing changes (most
            // groups of a pair of unrelated frames, e.g. a query against a keyframe DB) -- the two row reductions, the
            // acceptance test and the commit logic (~50 of a group's ~70 instructions) are skipped.
            {
                bool near = false;
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    unsigned d = 0;
#pragma unroll
                    for (int w = 0; w < 8; w++) d += __popc(dB[k][w] ^ dA[g][w]);
                    near = near || (!taken[k] && (KK ? d < (unsigned)TH_LOW : d <= (unsigned)TH_LOW));
                }
                if (__ballot(near && okAg[g]) == 0) continue;
            }
            unsigned r1, r2;
            reduce2(dA[g], r1, r2);
            const int b1 = (int)(r1 >> 16), b2 = (int)(r2 >> 16);                 // 256 when nothing is left
            const bool accept = okAg[g] && (KK ? (b1 < TH_LOW) : (b1 <= TH_LOW)) && (float)b1 < __fmul_rn(ratio, (float)b2);
            const unsigned c1 = r1 & 31u, c2 = r2 & 31u;
            unsigned long long winsK[NB], wins = 0;
#pragma unroll
            for (int k = 0; k < NB; k++) {
                winsK[k] = __ballot(accept && c1 == (unsigned)(col + 16 * k));     // bit 16 r + lane column
                wins |= winsK[k];
            }
            const unsigned long long cares = __ballot(okAg[g] && ((unsigned)col == (c1 & 15u) || (unsigned)col == (c2 & 15u)));
            const unsigned w0 = (unsigned)wins & 0xFFFFu, w1 = (unsigned)(wins >> 16) & 0xFFFFu, w2 = (unsigned)(wins >> 32) & 0xFFFFu;
            const unsigned i1 = (unsigned)(cares >> 16) & 0xFFFFu, i2 = (unsigned)(cares >> 32) & 0xFFFFu, i3 = (unsigned)(cares >> 48);
            if (((w0 & (i1 | i2 | i3)) | (w1 & (i2 | i3)) | (w2 & i3)) == 0) {
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    if (accept && c1 == (unsigned)(col + 16 * k)) {
                        takenB[jB[k]] = 1;
                        const int rIdx = KK ? iA : jB[k];
                        res[rIdx] = (int16_t)(KK ? jB[k] : iA);
                    }
                    const unsigned all = (unsigned)(winsK[k] | (winsK[k] >> 16) | (winsK[k] >> 32) | (winsK[k] >> 48)) & 0xFFFFu;
                    taken[k] = taken[k] || ((all >> col) & 1u);                   // in every row: the column is gone
                }
                continue;
            }
            const unsigned long long okMask = __ballot(okAg[g]);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (abase + 4 * g + r >= na) continue;
                if (!((okMask >> (16 * r)) & 1)) continue;
                unsigned q1, q2;
                reduce2(dA[g], q1, q2);                                           // with the flags as they are now
                const unsigned m1 = (unsigned)__builtin_amdgcn_readlane((int)q1, 16 * r);
                const unsigned m2 = (unsigned)__builtin_amdgcn_readlane((int)q2, 16 * r);
                const int best1 = (int)(m1 >> 16), best2 = (int)(m2 >> 16);
                const bool pass = KK ? (best1 < TH_LOW) : (best1 <= TH_LOW);      // :772 vs :625
                if (pass && (float)best1 < __fmul_rn(ratio, (float)best2)) {
                    const int win = (int)(m1 & 31u);
#pragma unroll
                    for (int k = 0; k < NB; k++)
                        if (col + 16 * k == win) {
                            taken[k] = true;
                            if (rowI == r) {
                                takenB[jB[k]] = 1;
                                const int rIdx = KK ? iA : jB[k];
                                res[rIdx] = (int16_t)(KK ? jB[k] : iA);
                            }
                        }
                }
            }
        }
    }
}

// KK == false: SearchByBoW(KeyFrame*, Frame&)   -> out[iB] = iA   (B = Frame, A = KeyFrame)
// KK == true : SearchByBoW(KeyFrame*, KeyFrame*) -> out[iA] = iB  (A = KF1, B = KF2), B needs valid, strict <
//
// 16 waves per pair; a wave owns whole vocabulary nodes.  Inside a node the A-side loop is serial (the
// greedy rule), but nothing in it touches memory: each lane keeps ONE B descriptor and its "taken" flag
// in registers, the A descriptors of the node are preloaded one per lane and broadcast with v_readlane,
// best / second-best are two DPP min-reductions on packed (distance << 16 | position).
// (two 16-wave workgroups per CU need 8 waves per SIMD: at most 64 VGPRs)
template <bool KK>
__device__ __forceinline__ void match_bow_pair(const BowSide& A, const BowSide& B, int nNodes, int capLds, float ratio, int checkOri,
                                               int32_t* __restrict__ match, int matchStride, int32_t* __restrict__ nmatchesOut)
{
    extern __shared__ uint32_t msm[];
    // carve-up: keysA[cap] keysB[cap] tmp[cap] cntw[nNodes] (u32) | startA cntA startB cntB [nNodes] (u16) |
    //           res[cap] (i16) | bin[cap] takenB[cap] okA[cap] (u8)
    uint32_t* keysA = msm;
    uint32_t* keysB = keysA + capLds;
    uint32_t* tmp = keysB + capLds;
    uint32_t* cntw = tmp + capLds;
    uint16_t* startA = reinterpret_cast<uint16_t*>(cntw + nNodes);
    uint16_t* cntA = startA + nNodes;
    uint16_t* startB = cntA + nNodes;
    uint16_t* cntB = startB + nNodes;
    int16_t* res = reinterpret_cast<int16_t*>(cntB + nNodes);
    uint8_t* bin = reinterpret_cast<uint8_t*>(res + capLds);
    uint8_t* takenB = bin + capLds;
    uint8_t* okALds = takenB + capLds;                // A-side "has a good MapPoint" flags (:590-595), read once
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int keepBins[3];
    __shared__ int nm;
    __shared__ int nextNode;

    const int tid = threadIdx.x, lane = tid & 63;
    if (A.n < 0 || B.n < 0 || A.n > capLds || B.n > capLds) {      // invalid pair (bow_side_of_store): reported, never run
        for (int i = tid; i < matchStride; i += blockDim.x) match[(size_t)blockIdx.x * matchStride + i] = -1;
        if (tid == 0) nmatchesOut[blockIdx.x] = -1;
        return;
    }
    const int nRes = KK ? A.n : B.n;
    for (int i = tid; i < nRes; i += blockDim.x) { res[i] = -1; bin[i] = 0xFF; }
    for (int i = tid; i < B.n; i += blockDim.x) takenB[i] = (KK && B.valid && !gload(B.valid + i)) ? 1 : 0;   // :750
    for (int i = tid; i < A.n; i += blockDim.x) okALds[i] = (A.valid && !gload(A.valid + i)) ? 0 : 1;
    if (tid < HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) { nm = 0; nextNode = 0; }
    load_or_build_csr(A, nNodes, keysA, tmp, cntw, startA, cntA);
    load_or_build_csr(B, nNodes, keysB, tmp, cntw, startB, cntB);

    // nodes differ a lot in size: waves take the next node from a shared counter instead of a fixed stride
    // (a fixed stride left the slowest wave 2.6x behind the fastest)
    while (true) {
        int node = 0;
        if (lane == 0) node = atomicAdd(&nextNode, 1);
        node = __builtin_amdgcn_readfirstlane(node);
        if (node >= nNodes) break;
        const int na = cntA[node], nb = cntB[node];
        if (na == 0 || nb == 0) continue;
        const int a0 = startA[node], b0 = startB[node];
        if (nb <= 16) {
            match_small_node<KK, 1>(A, B, na, nb, a0, b0, keysA, keysB, okALds, takenB, res, ratio, lane);
        } else if (nb <= 32) {
            match_small_node<KK, 2>(A, B, na, nb, a0, b0, keysA, keysB, okALds, takenB, res, ratio, lane);
        } else if (nb <= 2 * WAVE) {
            // ---- medium node: the node's B features fit the wave's registers, one or two per lane (positions lane and
            // lane + 64); nothing in the serial A loop touches memory
            uint32_t dB0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dB1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int jB0 = -1, jB1 = -1;
            bool taken0 = true, taken1 = true;
            if (lane < nb) {
                jB0 = (int)(keysB[b0 + lane] & 0xFFFFu);
                load_desc(B.desc + (size_t)jB0 * 32, dB0);
                taken0 = takenB[jB0] != 0;
            }
            const bool two = nb > WAVE;                            // wave-uniform
            if (lane + WAVE < nb) {
                jB1 = (int)(keysB[b0 + WAVE + lane] & 0xFFFFu);
                load_desc(B.desc + (size_t)jB1 * 32, dB1);
                taken1 = takenB[jB1] != 0;
            }
            for (int abase = 0; abase < na; abase += WAVE) {
                const int nChunk = min(WAVE, na - abase);
                uint32_t dA[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int iAl = -1;
                bool okA = false;
                if (lane < nChunk) {
                    iAl = (int)(keysA[a0 + abase + lane] & 0xFFFFu);
                    okA = okALds[iAl] != 0;                               // :590-595
                    if (okA) load_desc(A.desc + (size_t)iAl * 32, dA);
                }
                const unsigned long long okMask = __ballot(okA);
                for (int p = 0; p < nChunk; p++) {
                    if (!((okMask >> p) & 1)) continue;
                    uint32_t a8[8];
#pragma unroll
                    for (int w = 0; w < 8; w++) a8[w] = (uint32_t)__builtin_amdgcn_readlane((int)dA[w], p);
                    unsigned d0 = 256u, d1 = 256u;
                    if (!taken0) {
                        d0 = 0;
#pragma unroll
                        for (int w = 0; w < 8; w++) d0 += __popc(dB0[w] ^ a8[w]);
                    }
                    if (two && !taken1) {
                        d1 = 0;
#pragma unroll
                        for (int w = 0; w < 8; w++) d1 += __popc(dB1[w] ^ a8[w]);
                    }
                    // (nothing within TH_LOW: no acceptance possible, see match_small_node)
                    if (__ballot(KK ? min(d0, d1) < (unsigned)TH_LOW : min(d0, d1) <= (unsigned)TH_LOW) == 0) continue;
                    // packed (distance << 16 | position): the lane's smaller and larger value, then the wave's two smallest
                    const unsigned v0 = (d0 << 16) | (unsigned)lane, v1 = (d1 << 16) | (unsigned)(lane + WAVE);
                    const unsigned lo = min(v0, v1), hi = max(v0, v1);
                    const unsigned m1 = orb_wave_umin(lo);
                    const unsigned m2 = orb_wave_umin(lo == m1 ? hi : lo);
                    const int best1 = (int)(m1 >> 16), best2 = (int)min(256u, m2 >> 16);   // 256 when nothing is left
                    const bool pass = KK ? (best1 < TH_LOW) : (best1 <= TH_LOW);          // :772 vs :625
                    if (pass && (float)best1 < __fmul_rn(ratio, (float)best2)) {
                        const int win = (int)(m1 & 0xFFFFu);
                        const int iA = __builtin_amdgcn_readlane(iAl, p);
                        if ((win & (WAVE - 1)) == lane) {
                            const int jB = win < WAVE ? jB0 : jB1;
                            if (win < WAVE) taken0 = true; else taken1 = true;
                            takenB[jB] = 1;
                            const int rIdx = KK ? iA : jB;
                            res[rIdx] = (int16_t)(KK ? jB : iA);
                        }
                    }
                }
            }
        } else {
            // ---- general path (> 128 B features in one node): chunked, taken flags in LDS
            for (int p = a0; p < a0 + na; p++) {
                const int iA = (int)(keysA[p] & 0xFFFFu);
                if (!okALds[iA]) continue;
                uint32_t dA[8], dB[8];
                load_desc(A.desc + (size_t)iA * 32, dA);
                unsigned best = (256u << 16) | 0xFFFFu, second = 256u;
                for (int base = 0; base < nb; base += WAVE) {
                    const int q = base + lane;
                    unsigned d = 256u;
                    if (q < nb) {
                        const int j = (int)(keysB[b0 + q] & 0xFFFFu);
                        if (!takenB[j]) {
                            load_desc(B.desc + (size_t)j * 32, dB);
                            d = (unsigned)hamming8(dA, dB);
                        }
                    }
                    const unsigned mine = (d << 16) | (unsigned)q;
                    const unsigned m1 = orb_wave_umin(mine);
                    const unsigned m2 = orb_wave_umin(mine == m1 ? 0xFFFFFFFFu : mine);
                    const unsigned d1 = m1 >> 16, d2 = min(256u, m2 >> 16);
                    if (d1 < (best >> 16)) { second = min(best >> 16, d2); best = m1; }
                    else second = min(second, d1);
                }
                const int best1 = (int)(best >> 16), best2 = (int)second;
                const bool pass = KK ? (best1 < TH_LOW) : (best1 <= TH_LOW);
                if (pass && (float)best1 < __fmul_rn(ratio, (float)best2)) {
                    const int jB = (int)(keysB[b0 + (int)(best & 0xFFFFu)] & 0xFFFFu);
                    if (lane == 0) {
                        takenB[jB] = 1;
                        const int rIdx = KK ? iA : jB;
                        res[rIdx] = (int16_t)(KK ? jB : iA);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();
            }
        }
    }
    __syncthreads();

    // ---- rotation histogram + top-3 filter (:663-684)
    // the rotation bin of every match (:634-641) is computed here, for all matches at once: inside the node loop the two
    // angle loads were a dependent global round trip per accepted match
    int local = 0;
    for (int i = tid; i < nRes; i += blockDim.x)
        if (res[i] >= 0) {
            local++;
            if (checkOri) {
                const int iA = KK ? i : res[i], jB = KK ? res[i] : i;
                const int bb = rot_bin(gload(A.angle + (size_t)iA * A.angleStride), gload(B.angle + (size_t)jB * B.angleStride));
                bin[i] = (uint8_t)bb;
                atomicAdd(&hist[bb], 1);
            }
        }
    if (local) atomicAdd(&nm, local);
    __syncthreads();
    if (checkOri) {
        if (tid < WAVE) three_maxima_wave(hist, keepBins, tid);
        __syncthreads();
        int dropped = 0;
        for (int i = tid; i < nRes; i += blockDim.x)
            if (res[i] >= 0) {
                const int bb = bin[i];
                if (bb != keepBins[0] && bb != keepBins[1] && bb != keepBins[2]) { res[i] = -1; dropped++; }
            }
        if (dropped) atomicSub(&nm, dropped);
        __syncthreads();
    }
    int32_t* out = match + (size_t)blockIdx.x * matchStride;
    for (int i = tid; i < nRes; i += blockDim.x) out[i] = res[i];
    if (tid == 0) nmatchesOut[blockIdx.x] = nm;
}

// the BowSide of frame `idx` of a feature store (side 0 = keyframe: carries the "has a good MapPoint" flags)
__device__ __forceinline__ BowSide bow_side_of_store(const orb_featstore& S, int idx, int s, int nNodes)
{
    BowSide b;
    const bool inStore = idx >= 0 && idx < S.n_frames;
    const size_t row = inStore ? (size_t)idx * S.cap : 0;
    b.desc = S.desc + row * 32;
    b.angle = &S.kps[row].angle;
    b.angleStride = sizeof(orb_keypoint) / sizeof(float);
    b.valid = (s == 0 && S.valid) ? S.valid + row : nullptr;
    b.nodeOf = S.node_of + row;
    const int n = inStore ? S.counts[idx] : -1;
    b.n = (n >= 0 && n <= S.cap) ? n : -1;                 // an index or count outside the store: the pair reports nmatches = -1
    b.csrKeys = S.csr_keys ? S.csr_keys + row : nullptr;
    b.csrStart = (S.csr_keys && inStore) ? S.csr_start + (size_t)idx * nNodes : nullptr;
    b.csrCnt = (S.csr_keys && inStore) ? S.csr_cnt + (size_t)idx * nNodes : nullptr;
    if (!S.csr_start || !S.csr_cnt) b.csrKeys = nullptr;
    return b;
}

template <bool KK>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_match_bow(const BowSide* __restrict__ sidesA, const BowSide* __restrict__ sidesB,
                                                    int nNodes, int capLds, float ratio, int checkOri,
                                                    int32_t* __restrict__ match, int matchStride,
                                                    int32_t* __restrict__ nmatchesOut)
{
    const BowSide A = sidesA[blockIdx.x], B = sidesB[blockIdx.x];
    match_bow_pair<KK>(A, B, nNodes, capLds, ratio, checkOri, match, matchStride, nmatchesOut);
}

// SearchByBoW(KeyFrame*, Frame&) over pairs (kfIndex[p], fIndex[p]) of a feature store: the pair's two sides are derived
// by the workgroup itself (a separate fill kernel was a 5 us launch in front of every batch, a tenth of a 1000-pair query)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_match_bow_store(orb_featstore S, const int32_t* __restrict__ kfIndex,
                                                    const int32_t* __restrict__ fIndex, int nNodes, float ratio, int checkOri,
                                                    int32_t* __restrict__ match, int32_t* __restrict__ nmatchesOut)
{
    const BowSide A = bow_side_of_store(S, kfIndex[blockIdx.x], 0, nNodes), B = bow_side_of_store(S, fIndex[blockIdx.x], 1, nNodes);
    match_bow_pair<false>(A, B, nNodes, S.cap, ratio, checkOri, match, S.cap, nmatchesOut);
}

// ------------------------------------------------------------------ host side
static size_t match_lds_bytes(int capLds, int nNodes)
{
    return (size_t)capLds * (4 + 4 + 4 + 2 + 1 + 1 + 1) + (size_t)nNodes * (4 + 8) + 16;
}

extern "C" int orb_matcher_create(int device_id, orb_matcher** out)
{
    if (!out) return ORB_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        orb_set_error("no HIP device: liborbhip has no CPU fallback");
        return ORB_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= ndev) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(device_id));
    orb_matcher* m = new (std::nothrow) orb_matcher();
    if (!m) return ORB_ERR_INTERNAL;
    m->device = device_id;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) m->cus = cus; }
    {
        // LDS a workgroup may take: 160 KB on gfx950 (what this library is written for).  Elsewhere the largest figure the runtime
        // reports (the plain per-block attribute is the 64 KB default on some parts, the opt-in / per-CU attributes the real size),
        // so that the query-form matcher falls back to the pair kernel earlier instead of failing its launch.
        hipDeviceProp_t prop;
        const bool known = hipGetDeviceProperties(&prop, device_id) == hipSuccess;
        if (!known || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            int best = 64 * 1024, v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id) == hipSuccess) best = std::max(best, v);
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, device_id) == hipSuccess) best = std::max(best, v);
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device_id) == hipSuccess) best = std::max(best, v);
            m->ldsMax = (size_t)std::min(best, 160 * 1024);
        }
        (void)hipGetLastError();
    }
    hipError_t e = orb_stream_create(&m->stream, device_id, 1);
    if (e != hipSuccess) { delete m; orb_set_error("hipStreamCreate: %s", hipGetErrorString(e)); return ORB_ERR_HIP; }
    (void)hipEventCreateWithFlags(&m->waitEv, hipEventDisableTiming);
    *out = m;
    return ORB_OK;
}

extern "C" int orb_matcher_wait_for(orb_matcher* m, void* other_stream)
{
    if (!m) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(m->device));
    ORB_HIP_TRY(hipEventRecord(m->waitEv, (hipStream_t)other_stream));
    ORB_HIP_TRY(hipStreamWaitEvent(m->stream, m->waitEv, 0));
    return ORB_OK;
}

extern "C" void orb_matcher_destroy(orb_matcher* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    m->sidesA.release(); m->sidesB.release(); m->out.release(); m->nm.release(); m->plan.release(); m->qctr.release();
    for (auto& b : m->stage) b.release();
    for (auto& b : m->init) b.release();
    if (m->waitEv) (void)hipEventDestroy(m->waitEv);
    if (m->stream) orb_stream_destroy(m->stream, m->device);
    delete m;
}

extern "C" int orb_matcher_sync(orb_matcher* m)
{
    if (!m) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(m->device));
    ORB_HIP_TRY(hipStreamSynchronize(m->stream));
    return ORB_OK;
}

extern "C" void* orb_matcher_stream(orb_matcher* m) { return m ? (void*)m->stream : nullptr; }

extern "C" int orb_bow_assign_device(orb_matcher* m, const uint8_t* d_desc, const int32_t* d_counts, int nFrames,
                                     int cap, const uint8_t* d_cent, uint16_t* d_nodeOf)
{
    if (!m || !d_desc || !d_counts || !d_cent || !d_nodeOf || nFrames < 0 || cap <= 0) return ORB_ERR_INVALID;
    if (nFrames == 0) return ORB_OK;
    ORB_HIP_TRY(hipSetDevice(m->device));
    hipLaunchKernelGGL(k_bow_assign, dim3((cap + 255) / 256, nFrames), dim3(256), 0, m->stream, d_desc, d_counts, cap,
                       d_cent, d_nodeOf);
    ORB_HIP_TRY(hipGetLastError());
    return ORB_OK;
}

static int launch_match(orb_matcher* m, bool kk, const BowSide* dA, const BowSide* dB, int nPairs, int nNodes,
                        int capLds, float ratio, int checkOri, int32_t* dMatch, int matchStride, int32_t* dNm)
{
    const size_t lds = match_lds_bytes(capLds, nNodes);
    if (lds > 156 * 1024 || capLds > 1024 * ORB_CSR_MAXPT) {
        orb_set_error("feature capacity %d x %d nodes exceeds the match kernel's LDS budget", capLds, nNodes);
        return ORB_ERR_UNSUPPORTED;
    }
    if (lds > 64 * 1024) {                                 // frames of > ~3900 features: the CU's whole LDS for one pair
        ORB_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_match_bow<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ORB_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_match_bow<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (kk)
        hipLaunchKernelGGL(k_match_bow<true>, dim3(nPairs), dim3(1024), lds, m->stream, dA, dB, nNodes, capLds, ratio,
                           checkOri, dMatch, matchStride, dNm);
    else
        hipLaunchKernelGGL(k_match_bow<false>, dim3(nPairs), dim3(1024), lds, m->stream, dA, dB, nNodes, capLds, ratio,
                           checkOri, dMatch, matchStride, dNm);
    ORB_HIP_TRY(hipGetLastError());
    return ORB_OK;
}

extern "C" int orb_match_bow_batch_device(orb_matcher* m, const orb_featstore* store, const int32_t* d_kf,
                                          const int32_t* d_f, int nPairs, float ratio, int checkOri,
                                          int32_t* d_match, int32_t* d_nm)
{
    if (!m || !store || !d_kf || !d_f || !d_match || !d_nm || nPairs < 0) return ORB_ERR_INVALID;
    if (nPairs == 0) return ORB_OK;
    if (store->cap <= 0 || store->cap > 8192) { orb_set_error("featstore cap must be 1..8192"); return ORB_ERR_UNSUPPORTED; }
    ORB_HIP_TRY(hipSetDevice(m->device));
    const int nNodes = store->n_nodes > 0 ? store->n_nodes : 128;
    const size_t lds = match_lds_bytes(store->cap, nNodes);
    if (lds > 156 * 1024 || store->cap > 1024 * ORB_CSR_MAXPT) {
        orb_set_error("feature capacity %d x %d nodes exceeds the match kernel's LDS budget", store->cap, nNodes);
        return ORB_ERR_UNSUPPORTED;
    }
    if (lds > 64 * 1024)                                   // frames of > ~3900 features: the CU's whole LDS for one pair
        ORB_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_match_bow_store), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_match_bow_store, dim3(nPairs), dim3(1024), lds, m->stream, *store, d_kf, d_f, nNodes, ratio, checkOri, d_match,
                       d_nm);
    ORB_HIP_TRY(hipGetLastError());
    return ORB_OK;
}

extern "C" int orb_bow_build_csr_desc_device(orb_matcher* m, const uint16_t* d_node_of, const int32_t* d_counts, const uint8_t* d_desc,
                                             int nFrames, int cap, int nNodes, uint32_t* d_keys, uint16_t* d_start, uint16_t* d_cnt,
                                             uint8_t* d_csr_desc)
{
    if (!m || !d_node_of || !d_counts || !d_keys || !d_start || !d_cnt || nFrames < 0 || nNodes <= 0) return ORB_ERR_INVALID;
    if ((d_csr_desc != nullptr) != (d_desc != nullptr)) return ORB_ERR_INVALID;
    if (nFrames == 0) return ORB_OK;
    if (cap <= 0 || cap > 8192) { orb_set_error("featstore cap must be 1..8192"); return ORB_ERR_UNSUPPORTED; }
    const size_t lds = (size_t)cap * 8 + (size_t)nNodes * 8;
    if (lds > 156 * 1024) { orb_set_error("feature capacity %d x %d nodes exceeds the CSR kernel's LDS budget", cap, nNodes); return ORB_ERR_UNSUPPORTED; }
    ORB_HIP_TRY(hipSetDevice(m->device));                      // the attribute below is per device
    if (lds > 64 * 1024) ORB_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_csr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_build_csr, dim3(nFrames), dim3(1024), lds, m->stream, d_node_of, d_counts, cap, nNodes, d_keys, d_start,
                       d_cnt, d_desc, d_csr_desc);
    ORB_HIP_TRY(hipGetLastError());
    return ORB_OK;
}

extern "C" int orb_bow_build_csr_device(orb_matcher* m, const uint16_t* d_node_of, const int32_t* d_counts, int nFrames, int cap,
                                        int nNodes, uint32_t* d_keys, uint16_t* d_start, uint16_t* d_cnt)
{
    return orb_bow_build_csr_desc_device(m, d_node_of, d_counts, nullptr, nFrames, cap, nNodes, d_keys, d_start, d_cnt, nullptr);
}

// CSR feature vectors of both sides -> per-feature compact node index over the COMMON node ids
// (the merge-walk of :578-659 only ever pairs equal ids).  Requires ascending indices inside a node,
// which is what DBoW2's transform() produces (fv[node].push_back(i) for ascending i).
static int csr_to_nodeof(const orb_featvec* fa, int na, const orb_featvec* fb, int nb,
                         std::vector<uint16_t>& nodeA, std::vector<uint16_t>& nodeB, int& nCommon)
{
    nodeA.assign(std::max(na, 1), NODE_NONE);
    nodeB.assign(std::max(nb, 1), NODE_NONE);
    nCommon = 0;
    int a = 0, b = 0;
    auto check = [](const orb_featvec* fv, int k, int n) -> bool {
        for (int p = fv->offsets[k]; p < fv->offsets[k + 1]; p++) {
            if (fv->indices[p] < 0 || fv->indices[p] >= n) return false;
            if (p > fv->offsets[k] && fv->indices[p] <= fv->indices[p - 1]) return false;
        }
        return true;
    };
    while (a < fa->n_nodes && b < fb->n_nodes) {
        if (a > 0 && fa->node_ids[a] <= fa->node_ids[a - 1]) return ORB_ERR_INVALID;
        if (b > 0 && fb->node_ids[b] <= fb->node_ids[b - 1]) return ORB_ERR_INVALID;
        if (fa->node_ids[a] == fb->node_ids[b]) {
            if (nCommon >= MAX_NODES) { orb_set_error("more than %d common vocabulary nodes", MAX_NODES); return ORB_ERR_UNSUPPORTED; }
            if (!check(fa, a, na) || !check(fb, b, nb)) {
                orb_set_error("feature-vector indices must be in range and ascending inside a node");
                return ORB_ERR_UNSUPPORTED;
            }
            for (int p = fa->offsets[a]; p < fa->offsets[a + 1]; p++) nodeA[fa->indices[p]] = (uint16_t)nCommon;
            for (int p = fb->offsets[b]; p < fb->offsets[b + 1]; p++) nodeB[fb->indices[p]] = (uint16_t)nCommon;
            nCommon++; a++; b++;
        } else if (fa->node_ids[a] < fb->node_ids[b]) a++;
        else b++;
    }
    return ORB_OK;
}

static int match_host(orb_matcher* m, bool kk,
                      const uint8_t* descA, const float* angA, const uint8_t* validA, int nA, const orb_featvec* fvA,
                      const uint8_t* descB, const float* angB, const uint8_t* validB, int nB, const orb_featvec* fvB,
                      float ratio, int checkOri, int32_t* matchOut, int* nmatches)
{
    if (!m || nA < 0 || nB < 0 || !fvA || !fvB || !nmatches) return ORB_ERR_INVALID;
    const int nRes = kk ? nA : nB;
    *nmatches = 0;
    if (nRes > 0 && !matchOut) return ORB_ERR_INVALID;
    for (int i = 0; i < nRes; i++) matchOut[i] = -1;
    if (nA == 0 || nB == 0) return ORB_OK;
    if (!descA || !descB || !angA || !angB) return ORB_ERR_INVALID;
    if (nA > 32767 || nB > 32767) return ORB_ERR_UNSUPPORTED;
    std::vector<uint16_t> nodeA, nodeB;
    int nCommon = 0, rc;
    if ((rc = csr_to_nodeof(fvA, nA, fvB, nB, nodeA, nodeB, nCommon)) != ORB_OK) return rc;
    if (nCommon == 0) return ORB_OK;
    ORB_HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    const size_t sz[10] = {(size_t)nA * 32, (size_t)nA * 4, (size_t)nA, (size_t)nA * 2,
                           (size_t)nB * 32, (size_t)nB * 4, (size_t)nB, (size_t)nB * 2,
                           sizeof(BowSide) * 2, (size_t)nRes * 4 + 4};
    const void* src[8] = {descA, angA, validA, nodeA.data(), descB, angB, validB, nodeB.data()};
    for (int i = 0; i < 10; i++)
        if ((rc = m->stage[i].ensure(sz[i])) != ORB_OK) return rc;
    for (int i = 0; i < 8; i++)
        if (src[i]) ORB_HIP_TRY(hipMemcpyAsync(m->stage[i].p, src[i], sz[i], hipMemcpyHostToDevice, st));
    BowSide sides[2];
    sides[0] = {(const uint8_t*)m->stage[0].p, (const float*)m->stage[1].p, 1,
                validA ? (const uint8_t*)m->stage[2].p : nullptr, (const uint16_t*)m->stage[3].p, nA, nullptr, nullptr, nullptr};
    sides[1] = {(const uint8_t*)m->stage[4].p, (const float*)m->stage[5].p, 1,
                validB ? (const uint8_t*)m->stage[6].p : nullptr, (const uint16_t*)m->stage[7].p, nB, nullptr, nullptr, nullptr};
    ORB_HIP_TRY(hipMemcpyAsync(m->stage[8].p, sides, sizeof(sides), hipMemcpyHostToDevice, st));
    int32_t* dOut = (int32_t*)m->stage[9].p;
    const int capLds = std::max(nA, nB);
    rc = launch_match(m, kk, (const BowSide*)m->stage[8].p, (const BowSide*)m->stage[8].p + 1, 1, nCommon, capLds, ratio,
                      checkOri, dOut, nRes, dOut + nRes);
    if (rc != ORB_OK) return rc;
    std::vector<int32_t> host((size_t)nRes + 1);
    ORB_HIP_TRY(hipMemcpyAsync(host.data(), dOut, ((size_t)nRes + 1) * 4, hipMemcpyDeviceToHost, st));
    ORB_HIP_TRY(hipStreamSynchronize(st));
    std::memcpy(matchOut, host.data(), (size_t)nRes * 4);
    *nmatches = host[nRes];
    return ORB_OK;
}

extern "C" int orb_match_bow(orb_matcher* m, const uint8_t* desc_kf, const float* angle_kf, const uint8_t* valid_kf,
                             int n_kf, const orb_featvec* fv_kf, const uint8_t* desc_f, const float* angle_f, int n_f,
                             const orb_featvec* fv_f, float ratio, int check_ori, int32_t* match_f, int* nmatches)
{
    return match_host(m, false, desc_kf, angle_kf, valid_kf, n_kf, fv_kf, desc_f, angle_f, nullptr, n_f, fv_f, ratio,
                      check_ori, match_f, nmatches);
}

extern "C" int orb_match_bow_kk(orb_matcher* m, const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1,
                                const orb_featvec* fv1, const uint8_t* desc2, const float* angle2, const uint8_t* valid2,
                                int n2, const orb_featvec* fv2, float ratio, int check_ori, int32_t* match_12,
                                int* nmatches)
{
    return match_host(m, true, desc1, angle1, valid1, n1, fv1, desc2, angle2, valid2, n2, fv2, ratio, check_ori,
                      match_12, nmatches);
}

// SearchForInitialization: implemented in orb_matcher_init.hip
