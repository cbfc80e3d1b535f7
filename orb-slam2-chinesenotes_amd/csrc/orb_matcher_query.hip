// orb_matcher_query.hip -- SearchByBoW(KeyFrame*, Frame&) of ONE query frame against MANY keyframes of a feature store:
// the Relocalization / loop-candidate loop of the reference (src/Tracking.cc:1471-1492: one
// ORBmatcher::SearchByBoW(pKF, mCurrentFrame, vvpMapPointMatches[i]) per candidate keyframe, src/ORBmatcher.cc:552-687).
//
// The pair kernel of orb_matcher.hip gives every (keyframe, frame) pair a 1024-thread workgroup that rebuilds both
// feature vectors and walks the ~100 common nodes with 16 waves: ~1800 vector instructions per wave of bookkeeping around
// ~10 x 10 distances per node, every descriptor load a dependent round trip inside a serial loop.  k_match_bow_query cuts
// the work by FEATURE instead:
//
//   once per workgroup   the query side -- descriptors in feature-vector order (csr_desc), the feature index of every
//                        position, node starts / counts, angles -- is staged in LDS and kept for all keyframes the
//                        workgroup handles (two 512-thread workgroups per CU, each taking keyframes from a counter).
//   phase 1 (parallel)   a LANE = one keyframe feature, taken in feature-vector order, so a wave reads 64 consecutive rows
//                        of the keyframe's csr_desc (2 KB, coalesced; waves take such chunks from a counter) and nothing in
//                        the loop depends on a load: the lane
//                        scans the query features of ITS node out of LDS (lanes of a node read the same address) and keeps
//                        best / second best as packed (distance << 16 | column) with one v_min + one v_med3 -- as if no
//                        query feature were taken yet.  22 vector instructions per 64 distances, no cross-lane operation.
//   phase 2 (per node)   the reference's serial loop (:586-650), but only where it can matter: nothing of a node is taken
//                        before the first feature that phase 1 would accept (nodes without one are done: in a pair of
//                        unrelated frames that is 9 nodes in 10), a feature whose phase-1 best distance exceeds TH_LOW can
//                        never be accepted (taking columns away only raises it), and taking a column away changes a
//                        later feature's best / second best only if the column IS its best or second-best column -- only
//                        then is that feature rescanned over the free columns, by a whole wave with a lane per column.
//                        A wave walks the candidates of a node in the reference's order with v_readlane; the "already
//                        matched" flags (:607) are one bit per column of a wave-uniform mask (bytes in LDS for nodes of more
//                        than 64 query features).
//   finish (same launch) rotation histogram (:634-641), ComputeThreeMaxima, top-3 filter (:663-684), match count; the result
//                        row leaves LDS once, coalesced -- no fill pass, no second kernel.
//
// Results are identical to the pair kernel's and the oracle's.  Measured on an MI355X (tools/qk_stamps.py, one 752x480 frame
// against 1000 keyframes, nothing else on the GPU): 50-54 us per query, the pair kernel 78; per keyframe of a workgroup
// ~9.4 us phase 1 + 1.4 waiting for its slowest wave (at the issue rate of its 21 instructions per 64 distances: ~24 columns
// per wave because a wave's lanes sit in several nodes of different sizes), 0.6 replay (16 for the keyframe of the query's own
// scene: an acceptance in nearly every node) + 2.6 waiting, 2.3 histogram + filter + write-out; 4.7 once for the query side.
#include <algorithm>
#include <cstdlib>
#include <new>

#include "orb_matcher_internal.h"
#include "orb_bow_device.h"

#pragma clang fp contract(off)

// G keyframes are in flight per workgroup, one per 1 / G of its threads (template parameter: 2, or 1 where two do not fit
// the LDS; 4 -- one round over a 1000-keyframe list -- measured slower: 86 against 64-70 us, every barrier then waits for the
// slowest of four keyframes)
// diagnostics (orb_matcher_set_stage_stamps): thread 0 of a workgroup leaves the 100 MHz clock at the stage boundaries of its FIRST keyframe pair
#define QK_STAMP(k) do { if (stamps && tid == 0 && iter == 0) stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)

__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// popcount(x) + acc in ONE instruction (the compiler emits v_bcnt x, 0 and separate v_add3)
__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc)
{
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

__device__ __forceinline__ unsigned hamming_acc(const uint4& alo, const uint4& ahi, const uint4& blo, const uint4& bhi)
{
    unsigned d = bcnt_acc(alo.x ^ blo.x, 0u);                       // 8 xor + 8 bcnt
    d = bcnt_acc(alo.y ^ blo.y, d);
    d = bcnt_acc(alo.z ^ blo.z, d);
    d = bcnt_acc(alo.w ^ blo.w, d);
    d = bcnt_acc(ahi.x ^ bhi.x, d);
    d = bcnt_acc(ahi.y ^ bhi.y, d);
    d = bcnt_acc(ahi.z ^ bhi.z, d);
    d = bcnt_acc(ahi.w ^ bhi.w, d);
    return d;
}

__device__ __forceinline__ unsigned wave_umax(unsigned v) { return ~orb_wave_umin(~v); }

__device__ __forceinline__ uint4 readlane4(const uint4& v, int l)
{
    return make_uint4((unsigned)__builtin_amdgcn_readlane((int)v.x, l), (unsigned)__builtin_amdgcn_readlane((int)v.y, l),
                      (unsigned)__builtin_amdgcn_readlane((int)v.z, l), (unsigned)__builtin_amdgcn_readlane((int)v.w, l));
}

// the acceptance test of :625-627 on packed (distance << 16 | column) values; anything >= 256 is "none" (the reference's
// initial bestDist1 = bestDist2 = 256 and its strict compares)
__device__ __forceinline__ bool accept2(unsigned best, unsigned second, float ratio)
{
    const int b1 = (int)min(best >> 16, 256u), b2 = (int)min(second >> 16, 256u);
    return b1 <= TH_LOW && (float)b1 < __fmul_rn(ratio, (float)b2);
}

// LDS per workgroup: the query side (38 bytes per feature of capacity) + per keyframe in flight 15 bytes per feature
static size_t query_lds_bytes(int cap, int nNodes, int G)
{
    return (size_t)cap * (32 + 4 + 2 + G * (8 + 4 + 2 + 1)) + (size_t)nNodes * (4 + G * 8) + 64;
}

// grid: x = workgroups, one per CU, each taking groups of G keyframe slots (p, p + nGroups, ...: far apart in the list, so
// neighbours of one scene -- the expensive pairs -- land in different groups) from a counter until the list is done: a
// workgroup that drew the keyframe of the query's own scene (16 us of phase 2 instead of 0.4) takes fewer groups afterwards;
// with a fixed two groups per workgroup those few set the kernel's end 15-20 us after everyone else's.  y = query
// ONE workgroup per CU is resident (hipOccupancyMaxActiveBlocksPerMultiprocessor): the fused kernel needs ~100 scalar and ~96 vector
// registers, a second 16-wave workgroup would need 8 waves per SIMD (<= 96 / 64), and forcing that (amdgpu_waves_per_eu) spills.
template <int QK_G, int QK_THREADS>
__global__ __launch_bounds__(QK_THREADS) void k_match_bow_query(orb_featstore S, const int32_t* __restrict__ kfIndex, int nKf,
                                                               const int32_t* __restrict__ fIndex, int nNodes, float ratio, int checkOri,
                                                               int32_t* __restrict__ match, int32_t* __restrict__ nmatches,
                                                               unsigned* __restrict__ groupCtr, unsigned* __restrict__ ctrToClear,
                                                               int ctrStride, unsigned long long* __restrict__ stamps)
{
    extern __shared__ uint4 qsm[];
    constexpr int QK_HALF = QK_THREADS / QK_G;
    const int cap = S.cap;
    const int tid = threadIdx.x, g = tid / QK_HALF, t = tid % QK_HALF, lane = tid & 63;
    uint4* bDesc = qsm;                                            // [2 cap] query descriptors, feature-vector order
    // [G][cap] uint2 behind bDesc: phase 1's (best, second) of the keyframe feature at a position (resAll below)
    float* angB = reinterpret_cast<float*>(reinterpret_cast<uint2*>(bDesc + 2 * cap) + (size_t)QK_G * cap);   // [cap] query angles by feature index
    // [G][cap] u32 behind angB: its feature index | "has a good MapPoint" << 16 (resIAll below)
    uint32_t* firstAcc = reinterpret_cast<uint32_t*>(angB + cap) + (size_t)QK_G * cap + (size_t)g * nNodes;   // [G][nNodes] first position phase 1 would accept (~0: none)
    uint16_t* jB = reinterpret_cast<uint16_t*>(reinterpret_cast<uint32_t*>(angB + cap) + (size_t)QK_G * (cap + nNodes));   // [cap] query feature index of a position
    uint16_t* startB = jB + cap;                                   // [nNodes]
    uint16_t* cntB = startB + nNodes;
    int16_t* row = reinterpret_cast<int16_t*>(cntB + nNodes) + (size_t)g * cap;             // [G][cap] the result row, by query feature index
    uint16_t* startA = reinterpret_cast<uint16_t*>(reinterpret_cast<int16_t*>(cntB + nNodes) + (size_t)QK_G * cap) + (size_t)g * 2 * nNodes;
    uint16_t* cntA = startA + nNodes;                              // [G][2][nNodes]
    uint8_t* takenB = reinterpret_cast<uint8_t*>(reinterpret_cast<uint16_t*>(reinterpret_cast<int16_t*>(cntB + nNodes) + (size_t)QK_G * cap) + (size_t)QK_G * 2 * nNodes) +
                      (size_t)g * cap;                             // [G][cap] by query position; after phase 2: the rotation bin by feature
    __shared__ int histS[QK_G][HISTO_LENGTH + 2];
    __shared__ int keepS[QK_G][4];
    __shared__ int nmS[QK_G];
    __shared__ int kfS[QK_G];                                      // the part's keyframe, -1: nothing to do (read by all waves in phases 1 and 2)
    __shared__ int nAS[QK_G];                                      // its feature count
    __shared__ int chunkCtr;                                       // phase 1: next chunk of 64 positions
    int* hist = histS[g];
    int* keepBins = keepS[g];

    const int q = blockIdx.y;
    const int fq = fIndex[q];
    int nQ = -1;
    if (fq >= 0 && fq < S.n_frames) nQ = gload(S.counts + fq);
    const bool queryOk = nQ >= 0 && nQ <= cap;
    const size_t rowB = (size_t)(queryOk ? fq : 0) * cap;

    if (queryOk) {                                                 // ---- the query side, once
        const uint4* src = reinterpret_cast<const uint4*>(S.csr_desc) + rowB * 2;
        for (int i = tid; i < 2 * nQ; i += QK_THREADS) bDesc[i] = gload(src + i);
        for (int i = tid; i < nQ; i += QK_THREADS) {
            jB[i] = (uint16_t)(gload(S.csr_keys + rowB + i) & 0xFFFFu);
            angB[i] = gload(&S.kps[rowB + i].angle);
        }
        for (int n = tid; n < nNodes; n += QK_THREADS) {
            startB[n] = gload(S.csr_start + (size_t)fq * nNodes + n);
            cntB[n] = gload(S.csr_cnt + (size_t)fq * nNodes + n);
        }
    }
    for (int n = t; n < nNodes; n += QK_HALF) firstAcc[n] = 0xFFFFFFFFu;      // (again after every phase 2)

    // The first group of a workgroup is its own index (all workgroups of a launch used to ask the counter for it within the same
    // microsecond: ~512 returning atomics on one address, served one after the other); the groups after it come from the
    // counter when the workgroup is free -- asking for the next group EARLY, behind the current one's work, was measured and
    // lost 4 us: a workgroup that drew an expensive keyframe then sits on a group that idle workgroups could have taken.
    __shared__ int nextGroup;
    const int nGroups = (nKf + QK_G - 1) / QK_G;
    // The counters a later launch will use (this launch's are groupCtr[]): the WHOLE slot, not only this launch's n_queries
    // entries -- a launch with fewer queries would otherwise leave the higher entries of a slot as an earlier, wider launch
    // left them (>= its gridDim.x), and the next wide launch on that slot would skip keyframe groups.
    if (blockIdx.x == 0 && q == 0)
        for (int i = tid; i < ctrStride; i += QK_THREADS) ctrToClear[i] = 0;
    int iter = 0;
    QK_STAMP(0);
    for (;; iter++) {
        if (iter > 0 && tid == 0) nextGroup = (int)gridDim.x + (int)atomicAdd(&groupCtr[q], 1u);
        __syncthreads();
        const int s0 = iter == 0 ? (int)blockIdx.x : nextGroup;
        if (s0 >= nGroups) break;
        const int slot = s0 + g * nGroups;
        const size_t pair = (size_t)q * nKf + (size_t)min(slot, nKf - 1);
        int32_t* out = match + pair * cap;
        int nA = -1, kf = 0;
        if (slot < nKf) {
            kf = kfIndex[slot];
            if (kf >= 0 && kf < S.n_frames) nA = gload(S.counts + kf);
        }
        const bool live = slot < nKf && queryOk && nA >= 0 && nA <= cap;
        if (slot < nKf && !live) {                                 // invalid pair: reported, never run
            for (int i = t; i < cap; i += QK_HALF) out[i] = -1;
            if (t == 0) nmatches[pair] = -1;
        }
        const size_t rowA = (size_t)(live ? kf : 0) * cap;
        if (t == 0) { kfS[g] = live ? kf : -1; nAS[g] = live ? nA : 0; }     // (last read before the previous phase 2's closing barrier)
        if (tid == 0) chunkCtr = 0;
        __syncthreads();                                           // the previous keyframes' rows have left LDS (and the query side is in)
        QK_STAMP(1);
        uint16_t sA0 = 0, cA0 = 0;                                 // the keyframe's node table: requested now, stored after phase 1
        if (live && t < nNodes) {
            sA0 = gload(S.csr_start + (size_t)kf * nNodes + t);
            cA0 = gload(S.csr_cnt + (size_t)kf * nNodes + t);
        }

        // ---- phase 1: every keyframe feature against the query features of its node, nothing taken.  Positions behind the
        // feature vector's end carry the key ~0 (k_build_csr), so a chunk needs nothing but the keyframe's feature count.
        // The waves take chunks of 64 positions (of all keyframes in flight, round robin) from a counter: a wave's time is
        // its largest node's size, which a fixed assignment left 2x apart between the waves.  (Requesting the next chunk's rows
        // ahead cost the registers that let two workgroups share a CU: the other workgroup's waves hide the latency instead.)
        {
            uint2* resAll = reinterpret_cast<uint2*>(bDesc + 2 * cap);
            uint32_t* resIAll = reinterpret_cast<uint32_t*>(angB + cap);
            uint32_t* firstAll = resIAll + (size_t)QK_G * cap;
            int maxA = 0;
#pragma unroll
            for (int k = 0; k < QK_G; k++) maxA = max(maxA, nAS[k]);
            const int nChunks = QK_G * ((maxA + 63) / 64);
            struct Chunk { uint4 alo, ahi; uint32_t key; uint8_t vflag; int gg, pos; };
            auto grab = [&]() {
                int c = 0;
                if (lane == 0) c = atomicAdd(&chunkCtr, 1);
                return __builtin_amdgcn_readfirstlane(c);
            };
            auto load_chunk = [&](int c, Chunk& C) {
                C.gg = c % QK_G;
                C.pos = (c / QK_G) * 64 + lane;
                C.key = 0xFFFFFFFFu;
                C.alo = make_uint4(0, 0, 0, 0);
                C.ahi = C.alo;
                C.vflag = 1;
                if (C.pos < nAS[C.gg]) {
                    const size_t r = (size_t)kfS[C.gg] * cap;
                    const uint4* d = reinterpret_cast<const uint4*>(S.csr_desc) + (r + C.pos) * 2;
                    C.key = gload(S.csr_keys + r + C.pos);
                    C.alo = gload(d);
                    C.ahi = gload(d + 1);
                    if (C.key != 0xFFFFFFFFu && S.valid) C.vflag = gload(S.valid + r + (C.key & 0xFFFFu));      // :590-595 (needed after the scan only)
                }
            };
            for (int c = grab(); c < nChunks; c = grab()) {
                Chunk C;
                load_chunk(c, C);
                const bool act = C.key != 0xFFFFFFFFu;
                const int node = act ? (int)(C.key >> 16) : 0, iA = (int)(C.key & 0xFFFFu);
                const int nb = act ? (int)cntB[node] : 0;
                const uint4* pB = bDesc + 2 * (int)startB[node];
                const int nbMax = (int)wave_umax((unsigned)nb);
                unsigned best = 0xFFFFFFFFu, second = 0xFFFFFFFFu;    // invariant: best <= second
                for (int j = 0; j < nbMax; j++) {
                    if (j < nb) {
                        const uint4 blo = pB[2 * j], bhi = pB[2 * j + 1];
                        const unsigned v = (hamming_acc(C.alo, C.ahi, blo, bhi) << 16) | (unsigned)j;
                        second = umed3(best, second, v);
                        best = min(best, v);
                    }
                }
                if (act) {
                    const bool ok = C.vflag != 0;
                    resAll[(size_t)C.gg * cap + C.pos] = make_uint2(best, second);
                    resIAll[(size_t)C.gg * cap + C.pos] = (uint32_t)iA | (ok ? 0x10000u : 0u);
                    // nothing of a node is taken before its first acceptance: phase 2 starts there, and skips nodes without one
                    if (ok && accept2(best, second, ratio)) atomicMin(&firstAll[C.gg * nNodes + node], (uint32_t)C.pos);
                }
            }
        }
        if (live) {
            if (t < nNodes) { startA[t] = sA0; cntA[t] = cA0; }
            for (int n = t + QK_HALF; n < nNodes; n += QK_HALF) {
                startA[n] = gload(S.csr_start + (size_t)kf * nNodes + n);
                cntA[n] = gload(S.csr_cnt + (size_t)kf * nNodes + n);
            }
            for (int i = t; i < nQ; i += QK_HALF) { row[i] = -1; takenB[i] = 0; }
            if (t < HISTO_LENGTH) hist[t] = 0;
            if (t == 0) nmS[g] = 0;
        }
        QK_STAMP(2);
        __syncthreads();
        QK_STAMP(3);

        // ---- phase 2: the reference's serial loop, a wave per (node, keyframe) that has an acceptance; all waves share
        // the tasks of all keyframes in flight
        {
            uint2* resAll = reinterpret_cast<uint2*>(bDesc + 2 * cap);
            uint32_t* resIAll = reinterpret_cast<uint32_t*>(angB + cap);
            uint32_t* firstAll = resIAll + (size_t)QK_G * cap;
            int16_t* rowAll = reinterpret_cast<int16_t*>(cntB + nNodes);
            uint16_t* startAAll = reinterpret_cast<uint16_t*>(rowAll + (size_t)QK_G * cap);
            uint8_t* takenAll = reinterpret_cast<uint8_t*>(startAAll + (size_t)QK_G * 2 * nNodes);
            struct Pre { uint2 bs; uint32_t ri; uint4 alo, ahi; bool cand, loaded; int first, aEnd; };
            const int nTasks = QK_G * nNodes;
            auto preload = [&](int task, int c0, Pre& P) {
                const int gg = task % QK_G, node = task / QK_G;
                P.first = c0 >= 0 ? c0 : (int)firstAll[gg * nNodes + node];
                P.aEnd = (int)startAAll[gg * 2 * nNodes + node] + (int)startAAll[gg * 2 * nNodes + nNodes + node];
                const int p = P.first + lane;
                P.bs = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                P.ri = 0;
                if (p < P.aEnd) { P.bs = resAll[(size_t)gg * cap + p]; P.ri = resIAll[(size_t)gg * cap + p]; }
                // a best distance above TH_LOW only grows when columns are taken away: never accepted
                P.cand = (P.ri >> 16) != 0 && (P.bs.x >> 16) <= (unsigned)TH_LOW;
                P.alo = make_uint4(0, 0, 0, 0);
                P.ahi = P.alo;
                // the candidates' descriptors, for the rescans (broadcast with v_readlane): requested ahead only where rescans are
                // likely -- three candidates or more; a node with one or two fetches them when the first rescan comes
                P.loaded = __builtin_popcountll(__ballot(P.cand)) >= 3;
                if (P.loaded && P.cand) {
                    const uint4* d = reinterpret_cast<const uint4*>(S.csr_desc) + ((size_t)kfS[gg] * cap + p) * 2;
                    P.alo = gload(d);
                    P.ahi = gload(d + 1);
                }
            };
            auto load_late = [&](int task, Pre& P) {
                if (P.loaded) return;
                P.loaded = true;
                if (P.cand) {
                    const uint4* d = reinterpret_cast<const uint4*>(S.csr_desc) + ((size_t)kfS[task % QK_G] * cap + P.first + lane) * 2;
                    P.alo = gload(d);
                    P.ahi = gload(d + 1);
                }
            };
            // wave w owns the tasks w, w + 16, ...; its lanes look at 64 of them at once
            constexpr int NW = QK_THREADS / 64;
            for (int tbase = tid >> 6; tbase < nTasks; tbase += 64 * NW) {
              const int mine = tbase + NW * lane;
              bool has = false;
              if (mine < nTasks) has = kfS[mine % QK_G] >= 0 && firstAll[(mine % QK_G) * nNodes + mine / QK_G] != 0xFFFFFFFFu;
              unsigned long long todo = __ballot(has);
              // (requesting the next task's data before walking the current one changed nothing: 16.9 against 15.9 us for the
              // keyframe of the query's own scene -- the candidate loop, not the round trip in front of it, is what a task costs)
              while (todo) {
                const int cur = tbase + NW * __builtin_ctzll(todo);
                todo &= todo - 1;
                Pre P;
                preload(cur, -1, P);
                const int gg = cur % QK_G, node = cur / QK_G;
                const int nb = cntB[node], b0 = startB[node];
                const uint4* pB = bDesc + 2 * b0;
                int16_t* rowG = rowAll + (size_t)gg * cap;
                if (nb <= 64) {
                    // the node's columns one per lane, their "already matched" flags (:607) one bit each of a wave-uniform mask
                    uint4 blo = make_uint4(0, 0, 0, 0), bhi = blo;
                    if (lane < nb) { blo = pB[2 * lane]; bhi = pB[2 * lane + 1]; }
                    unsigned long long taken = 0;
                    while (true) {
                        unsigned long long mask = __ballot(P.cand);
                        while (mask) {
                            const int i = __builtin_ctzll(mask);
                            mask &= mask - 1;
                            unsigned best = (unsigned)__builtin_amdgcn_readlane((int)P.bs.x, i), second = (unsigned)__builtin_amdgcn_readlane((int)P.bs.y, i);
                            const unsigned c1 = best & 0xFFFFu, c2 = second == 0xFFFFFFFFu ? c1 : (second & 0xFFFFu);
                            if (((taken >> c1) | (taken >> c2)) & 1ull) {   // a column it ranked first or second is gone: rescan the free ones
                                load_late(cur, P);
                                const uint4 a0 = readlane4(P.alo, i), a1 = readlane4(P.ahi, i);
                                const bool freeCol = lane < nb && !((taken >> lane) & 1ull);
                                const unsigned v = freeCol ? ((hamming_acc(a0, a1, blo, bhi) << 16) | (unsigned)lane) : 0xFFFFFFFFu;
                                best = orb_wave_umin(v);
                                second = orb_wave_umin(v == best ? 0xFFFFFFFFu : v);
                            }
                            if (accept2(best, second, ratio)) {
                                const unsigned c = best & 0xFFFFu;
                                taken |= 1ull << c;
                                if (lane == 0) rowG[jB[b0 + c]] = (int16_t)((unsigned)__builtin_amdgcn_readlane((int)P.ri, i) & 0xFFFFu);   // vpMapPointMatches[realIdxF] = pMP (:629)
                            }
                        }
                        if (P.first + 64 >= P.aEnd) break;
                        preload(cur, P.first + 64, P);             // (a node with more than 64 keyframe features)
                    }
                } else {
                    uint8_t* tk = takenAll + (size_t)gg * cap + b0;
                    while (true) {
                        unsigned long long mask = __ballot(P.cand);
                        while (mask) {
                            const int i = __builtin_ctzll(mask);
                            mask &= mask - 1;
                            unsigned best = (unsigned)__builtin_amdgcn_readlane((int)P.bs.x, i), second = (unsigned)__builtin_amdgcn_readlane((int)P.bs.y, i);
                            const unsigned c1 = best & 0xFFFFu, c2 = second == 0xFFFFFFFFu ? c1 : (second & 0xFFFFu);
                            if (tk[c1] | tk[c2]) {                // rescan, a lane per column, 64 columns at a time
                                load_late(cur, P);
                                const uint4 a0 = readlane4(P.alo, i), a1 = readlane4(P.ahi, i);
                                unsigned lb = 0xFFFFFFFFu, ls = 0xFFFFFFFFu;
                                for (int j = lane; j < nb; j += 64) {
                                    if (tk[j]) continue;
                                    const unsigned v = (hamming_acc(a0, a1, pB[2 * j], pB[2 * j + 1]) << 16) | (unsigned)j;
                                    ls = umed3(lb, ls, v);
                                    lb = min(lb, v);
                                }
                                best = orb_wave_umin(lb);
                                second = orb_wave_umin(lb == best ? ls : lb);
                            }
                            if (accept2(best, second, ratio)) {
                                const unsigned c = best & 0xFFFFu;
                                if (lane == 0) {
                                    tk[c] = 1;
                                    rowG[jB[b0 + c]] = (int16_t)((unsigned)__builtin_amdgcn_readlane((int)P.ri, i) & 0xFFFFu);
                                }
                                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            }
                        }
                        if (P.first + 64 >= P.aEnd) break;
                        preload(cur, P.first + 64, P);
                    }
                }
              }
            }
        }
        QK_STAMP(4);
        __syncthreads();
        QK_STAMP(5);

        // ---- rotation histogram + top-3 filter (:634-641, :663-684); takenB is dead: it holds the bin of a match now
        uint8_t* bin = takenB;
        int local = 0;
        for (int n = t; n < nNodes; n += QK_HALF) firstAcc[n] = 0xFFFFFFFFu;  // for the next keyframe's phase 1
        if (live) {
            for (int i = t; i < nQ; i += QK_HALF) {
                const int r = row[i];
                if (r >= 0) {
                    local++;
                    if (checkOri) {
                        const int bb = rot_bin(gload(&S.kps[rowA + r].angle), angB[i]);
                        bin[i] = (uint8_t)bb;
                        atomicAdd(&hist[bb], 1);
                    }
                }
            }
            if (local) atomicAdd(&nmS[g], local);
        }
        __syncthreads();
        if (checkOri) {
            if (live && t < WAVE) three_maxima_wave(hist, keepBins, t);
            __syncthreads();
            if (live) {
                int dropped = 0;
                for (int i = t; i < nQ; i += QK_HALF)
                    if (row[i] >= 0) {
                        const int bb = bin[i];
                        if (bb != keepBins[0] && bb != keepBins[1] && bb != keepBins[2]) { row[i] = -1; dropped++; }
                    }
                if (dropped) atomicSub(&nmS[g], dropped);
            }
            __syncthreads();
        }
        if (live) {
            for (int i = t; i < cap; i += QK_HALF) out[i] = i < nQ ? (int32_t)row[i] : -1;
            if (t == 0) nmatches[pair] = nmS[g];
        }
        QK_STAMP(6);
    }
}

extern "C" int orb_matcher_set_stage_stamps(orb_matcher* m, unsigned long long* d_stamps, size_t capacity)
{
    if (!m) return ORB_ERR_INVALID;
    m->stamps = d_stamps;
    m->stampCap = d_stamps ? capacity : 0;
    return ORB_OK;
}

// pair list of the query form for the pair kernel (frames too large for this kernel's LDS): pair q * nKf + k = (kfIndex[k], fIndex[q])
__global__ void k_query_pairs(const int32_t* __restrict__ kfIndex, int nKf, const int32_t* __restrict__ fIndex, int nQ,
                              int32_t* __restrict__ kfPairs, int32_t* __restrict__ fPairs)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nKf * nQ) return;
    kfPairs[i] = kfIndex[i % nKf];
    fPairs[i] = fIndex[i / nKf];
}

extern "C" int orb_match_bow_query_device(orb_matcher* m, const orb_featstore* store, const int32_t* d_kf_index, int n_kf,
                                          const int32_t* d_f_index, int n_queries, float ratio, int check_ori, int32_t* d_match,
                                          int32_t* d_nmatches)
{
    if (!m || !store || !d_kf_index || !d_f_index || !d_match || !d_nmatches || n_kf < 0 || n_queries < 0) return ORB_ERR_INVALID;
    if (n_kf == 0 || n_queries == 0) return ORB_OK;
    if (store->cap <= 0 || store->cap > 8192) { orb_set_error("featstore cap must be 1..8192"); return ORB_ERR_UNSUPPORTED; }
    if (!store->csr_keys || !store->csr_start || !store->csr_cnt || !store->csr_desc) {
        orb_set_error("orb_match_bow_query_device needs the store's per-frame feature vectors with node-sorted descriptors "
                      "(orb_bow_build_csr_desc_device)");
        return ORB_ERR_INVALID;
    }
    if (n_queries > 65535 || (unsigned long long)n_kf * (unsigned long long)n_queries >= (1ull << 31)) return ORB_ERR_UNSUPPORTED;
    const int nNodes = store->n_nodes > 0 ? store->n_nodes : 128;
    ORB_HIP_TRY(hipSetDevice(m->device));
    // (G keyframes in flight, threads per workgroup, workgroups per CU): ORB_QK_CFG=g,t,b overrides (tuning)
    static const char* envCfg = getenv("ORB_QK_CFG");
    static const int envDbg = getenv("ORB_QK_DBG") ? atoi(getenv("ORB_QK_DBG")) : 0;
    // Default: ONE keyframe per 512-thread workgroup, two workgroups per CU (each stages the query side itself): independent
    // workgroups drift apart, so the issue-bound phase 1 of one runs beside the latency-bound replay / histogram / write-out of
    // the other -- 54 against 65 us per 1000 keyframes for two keyframes in flight in one 1024-thread workgroup, whose halves
    // wait for each other at every barrier (256 threads x 2: 71 us).  Frames too large for two such workgroups per CU
    // (> ~1450 features) take the 1024-thread forms.
    int G = 1, TH = 512, perCu = 2;
    // the LDS budget comes from the device (160 KB on gfx950; a part with less takes the pair kernel earlier instead of failing)
    const size_t ldsCu = m->ldsMax, ldsWg = m->ldsMax - 4 * 1024;
    if (2 * query_lds_bytes(store->cap, nNodes, 1) > ldsCu) { G = 2; TH = 1024; perCu = 1; }
    if (envCfg) sscanf(envCfg, "%d,%d,%d", &G, &TH, &perCu);
    if (!((G == 1 && (TH == 256 || TH == 512 || TH == 1024)) || (G == 2 && TH == 1024) || (G == 4 && TH == 1024))) { G = 2; TH = 1024; perCu = 1; }
    while (G > 1 && query_lds_bytes(store->cap, nNodes, G) > ldsWg) G >>= 1;
    if (G == 1 && TH != 512 && TH != 256) TH = 1024;
    const size_t lds = query_lds_bytes(store->cap, nNodes, G);
    if (lds > ldsWg) {
        // frames of more than ~2900 features: the same pairs through the pair kernel (17 bytes of LDS per feature)
        const size_t nPairs = (size_t)n_kf * n_queries;
        int rc = m->plan.ensure(nPairs * 8);
        if (rc != ORB_OK) return rc;
        int32_t* kfPairs = (int32_t*)m->plan.p;
        int32_t* fPairs = kfPairs + nPairs;
        hipLaunchKernelGGL(k_query_pairs, dim3((unsigned)((nPairs + 255) / 256)), dim3(256), 0, m->stream, d_kf_index, n_kf, d_f_index,
                           n_queries, kfPairs, fPairs);
        ORB_HIP_TRY(hipGetLastError());
        return orb_match_bow_batch_device(m, store, kfPairs, fPairs, (int)nPairs, ratio, check_ori, d_match, d_nmatches);
    }
    const void* fn = G == 4 ? reinterpret_cast<const void*>(k_match_bow_query<4, 1024>)
                   : G == 2 ? reinterpret_cast<const void*>(k_match_bow_query<2, 1024>)
                   : TH == 512 ? reinterpret_cast<const void*>(k_match_bow_query<1, 512>)
                   : TH == 256 ? reinterpret_cast<const void*>(k_match_bow_query<1, 256>) : reinterpret_cast<const void*>(k_match_bow_query<1, 1024>);
    if (lds > 64 * 1024) ORB_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // a workgroup stages the query side once and keeps it for its keyframes: perCu resident workgroups per CU, each taking groups
    perCu = std::max(1, std::min(perCu, (int)(ldsCu / lds)));
    const long long slots = (long long)m->cus * perCu;
    const int blocks = (int)std::max<long long>(1, std::min<long long>((n_kf + G - 1) / G, std::max<long long>(1, slots / n_queries)));      // per query
    unsigned long long* stamps = (unsigned long long)blocks * n_queries * 8 <= m->stampCap ? m->stamps : nullptr;
    const dim3 grid(blocks, n_queries), block(TH);
    if ((size_t)n_queries > m->qctrStride) {                       // the counter ring (zeroed once; every launch clears a slot for a later one)
        const size_t stride = std::max<size_t>(64, (size_t)n_queries);
        ORB_HIP_TRY(hipStreamSynchronize(m->stream));
        int rc = m->qctr.ensure(8 * stride * sizeof(unsigned));
        if (rc != ORB_OK) return rc;
        ORB_HIP_TRY(hipMemsetAsync(m->qctr.p, 0, 8 * stride * sizeof(unsigned), m->stream));
        m->qctrStride = stride;
        m->qserial = 0;
    }
    unsigned* groupCtr = (unsigned*)m->qctr.p + (size_t)(m->qserial % 8) * m->qctrStride;
    unsigned* ctrToClear = (unsigned*)m->qctr.p + (size_t)((m->qserial + 4) % 8) * m->qctrStride;
    m->qserial++;
    if (envDbg) {
        int nb = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, TH, lds);
        fprintf(stderr, "[orb] k_match_bow_query<%d>: lds %zu B, blocks %d, occupancy %d workgroups per CU (%s)\n", G, lds, blocks, nb, hipGetErrorString(e));
    }
#define QK_LAUNCH(GG, TT) hipLaunchKernelGGL((k_match_bow_query<GG, TT>), grid, block, lds, m->stream, *store, d_kf_index, n_kf, d_f_index, nNodes, \
                                             ratio, check_ori, d_match, d_nmatches, groupCtr, ctrToClear, (int)m->qctrStride, stamps)
    if (G == 4) QK_LAUNCH(4, 1024);
    else if (G == 2) QK_LAUNCH(2, 1024);
    else if (TH == 512) QK_LAUNCH(1, 512);
    else if (TH == 256) QK_LAUNCH(1, 256);
    else QK_LAUNCH(1, 1024);
#undef QK_LAUNCH
    ORB_HIP_TRY(hipGetLastError());
    return ORB_OK;
}

// Frame::ComputeBoW of the query frames (descent + feature vector, written into the store) and the search, one call
// (reference src/Tracking.cc:1471-1492 per frame: ComputeBoW, then SearchByBoW against every candidate keyframe).
extern "C" int orb_bow_query_frames_device(orb_matcher* m, orb_vocab* v, const orb_featstore* store, int first_query, int n_queries,
                                           int levelsup, const int32_t* d_kf_index, int n_kf, const int32_t* d_f_index, float ratio,
                                           int check_ori, int32_t* d_match, int32_t* d_nmatches)
{
    if (!m || !v || !store || first_query < 0 || n_queries < 0 || first_query + n_queries > store->n_frames) return ORB_ERR_INVALID;
    if (!store->desc || !store->counts || !store->node_of || !store->csr_keys || !store->csr_start || !store->csr_cnt || !store->csr_desc ||
        store->cap <= 0 || store->n_nodes <= 0)
        return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(m->device));
    if (n_queries == 0) return ORB_OK;
    const size_t cap = (size_t)store->cap, f0 = (size_t)first_query, nn = (size_t)store->n_nodes;
    int rc = orb_bow_transform_device(m, v, store->desc + f0 * cap * 32, store->counts + f0, n_queries, store->cap, levelsup, nullptr, nullptr,
                                      const_cast<uint16_t*>(store->node_of) + f0 * cap);
    if (rc != ORB_OK) return rc;
    rc = orb_bow_build_csr_desc_device(m, store->node_of + f0 * cap, store->counts + f0, store->desc + f0 * cap * 32, n_queries, store->cap,
                                       store->n_nodes, const_cast<uint32_t*>(store->csr_keys) + f0 * cap,
                                       const_cast<uint16_t*>(store->csr_start) + f0 * nn, const_cast<uint16_t*>(store->csr_cnt) + f0 * nn,
                                       const_cast<uint8_t*>(store->csr_desc) + f0 * cap * 32);
    if (rc != ORB_OK) return rc;
    return orb_match_bow_query_device(m, store, d_kf_index, n_kf, d_f_index, n_queries, ratio, check_ori, d_match, d_nmatches);
}
