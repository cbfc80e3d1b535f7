// orb_fast.hip -- FAST-9/16 detection over strips of cells on gfx950.
// Reference: the cell loop of ORBextractor::ComputeKeyPointsOctTree, src/ORBextractor.cc:795-875
// (cv::FAST(cell ROI, iniThFAST, nonmax=true), fallback to minThFAST when it returns nothing).
//
// One wave64 per STRIP = a run of horizontally adjacent cells of one cell row (OrbStrip, orb_common.h).  The detection
// zones of adjacent cells tile the plane without overlap, so the score V of a pixel does not depend on its cell; only
// the 3x3 NMS (cell-local: it must not see the neighbouring cell) and the iniTh -> minTh fallback (per cell) do.
// The kernel is bound by VALU issue and, below ~20 waves per CU, by latency (rocprofv3, profiles/r02_*_valu.json:
// time follows 1/occupancy), so it is arranged for few instructions, DENSE lanes and a small LDS footprint:
//   * the strip ROI is staged in LDS with 8-byte loads; a lane works on a QUAD of 4 horizontally adjacent pixels
//     as two packed pairs;
//   * phase A is an exact cheap rejection on every pair: a 9-arc contains ring pixel k or k+8 for every k, so
//     V <= U = max(I - max_k min(r_k, r_k+8), min_k max(r_k, r_k+8) - I); with the four even k (11 dword reads)
//     about 5 pairs in 6 have U <= min(iniTh, minTh) in both pixels and are finished;
//   * surviving pairs go to two small ring queues (left / right pair of a quad) that are drained 64 at a time as
//     soon as 64 are waiting: phase B, the exact V(p) = max(I_p - min_arcs max_arc ring, max_arcs min_arc ring - I_p)
//     (7 rows x 12 bytes: 17 dword reads), always runs on full waves except for one last partial step per queue and
//     strip (one wave per CELL left phase B at 2.6 steps for 1.5 waves of work);
//   * both phases use v_pk_maximum3_f16 / v_pk_minimum3_f16: a u8 stored in a 16-bit half is a positive f16
//     subnormal whose order is the integer order, so the packed 3-input float min/max is exact and moves 2 pixels x
//     3 operands per instruction (tools/ubench).  One v_perm_b32 builds each packed ring operand from the window;
//   * V is threshold-free, and only pixels with V > min(iniTh, minTh) can ever be keypoints OR suppress one (a
//     neighbour at or below the threshold in force is below the pixel it would suppress): phase B appends exactly
//     those, (row, col, score), to a candidate queue.  There is NO score map next to the tile: when all scores are
//     known the tile is dead, it is zeroed, the candidates' scores are scattered into it and the NMS reads it as the
//     score map -- LDS per wave is one tile, not two (23 instead of 13 waves per CU at 3 cells per strip);
//   * NMS + emission run over the queue, not over the zone.  A strip whose candidates overflow the queue (noise
//     images) is put on a list and redone by k_fast_strips_dense (tile + full score map, dense scan): same results;
//   * the quadtree path of a candidate is two table look-ups (x and y bisect independently);
//   * everything derived from the strip rectangle alone comes precomputed in the 48-byte OrbStrip record.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "orb_kernels.h"
#include "orb_wave.h"

#define WAVE 64
#define FT_PAD 4                       // dwords of slack around the tile (edge quads read one dword outside)
#define FT_QRING 128                   // entries of one pair ring: < 64 left over + <= 64 appended per phase-A step
// LDS is sized per image geometry (dynamic): the tile uses a row pitch of `pdw` dwords (even: rows are staged with
// 8-byte stores) that covers the widest strip of the frame.

__device__ __forceinline__ unsigned pk_max3(unsigned a, unsigned b, unsigned c)
{
    unsigned d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned pk_min3(unsigned a, unsigned b, unsigned c)
{
    unsigned d;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned pk_max2(unsigned a, unsigned b)
{
    unsigned d;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ unsigned pk_min2(unsigned a, unsigned b)
{
    unsigned d;
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ unsigned pk_sub_i16(unsigned a, unsigned b)
{
    unsigned d;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ unsigned pk_add_u16(unsigned a, unsigned b)
{
    unsigned d;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ unsigned pk_add_u16_s(unsigned a, unsigned sUniform)
{
    unsigned d;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(sUniform));
    return d;
}
typedef unsigned short orb_us2 __attribute__((ext_vector_type(2)));
// v_pk_add_u16 through the vector type: the compiler sees the operands (the inline-asm form with a scalar operand made it
// pad the preceding v_cmp with five s_nop)
__device__ __forceinline__ unsigned pk_add_u16_n(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(orb_us2, a) + __builtin_bit_cast(orb_us2, b));
}
__device__ __forceinline__ unsigned pk_max_i16(unsigned a, unsigned b)
{
    unsigned d;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// bytes I and I+1 of the 12-byte window (w0,w1,w2) as two zero-extended 16-bit halves
template <int I>
__device__ __forceinline__ unsigned pick2(unsigned w0, unsigned w1, unsigned w2)
{
    if constexpr (I + 1 <= 7)
        return __builtin_amdgcn_perm(w1, w0, (unsigned)(I | 0x0c00 | ((I + 1) << 16) | 0x0c000000));
    else
        return __builtin_amdgcn_perm(w2, w1, (unsigned)((I - 4) | 0x0c00 | ((I - 3) << 16) | 0x0c000000));
}

// S = max(V, 0) for a pair of pixels whose window bytes are (C, C+1); W[r][0..2] are rows y-3..y+3.
template <int C>
__device__ __forceinline__ unsigned fast_pair(const unsigned (&W)[7][3])
{
    // ring k = 0..15: (dx,dy) = (0,3),(1,3),(2,2),(3,1),(3,0),(3,-1),(2,-2),(1,-3),(0,-3),(-1,-3),(-2,-2),
    //                           (-3,-1),(-3,0),(-3,1),(-2,2),(-1,3)      
    unsigned r[16];
#define RING(k, dx, dy) r[k] = pick2<C + (dx)>(W[(dy) + 3][0], W[(dy) + 3][1], W[(dy) + 3][2]);
    RING(0, 0, 3) RING(1, 1, 3) RING(2, 2, 2) RING(3, 3, 1) RING(4, 3, 0) RING(5, 3, -1) RING(6, 2, -2) RING(7, 1, -3)
    RING(8, 0, -3) RING(9, -1, -3) RING(10, -2, -2) RING(11, -3, -1) RING(12, -3, 0) RING(13, -3, 1) RING(14, -2, 2)
    RING(15, -1, 3)
#undef RING
    const unsigned c = pick2<C>(W[3][0], W[3][1], W[3][2]);
    unsigned hi3[16], lo3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        hi3[k] = pk_max3(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
        lo3[k] = pk_min3(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
    }
    unsigned hi9[16], lo9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        hi9[k] = pk_max3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);    // max of the 9-arc starting at k
        lo9[k] = pk_min3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
    }
    // M = min over arcs of the arc maximum, m = max over arcs of the arc minimum
    unsigned M = pk_min3(pk_min3(hi9[0], hi9[1], hi9[2]), pk_min3(hi9[3], hi9[4], hi9[5]), pk_min3(hi9[6], hi9[7], hi9[8]));
    M = pk_min3(M, pk_min3(hi9[9], hi9[10], hi9[11]), pk_min3(hi9[12], hi9[13], hi9[14]));
    M = pk_min2(M, hi9[15]);
    unsigned m = pk_max3(pk_max3(lo9[0], lo9[1], lo9[2]), pk_max3(lo9[3], lo9[4], lo9[5]), pk_max3(lo9[6], lo9[7], lo9[8]));
    m = pk_max3(m, pk_max3(lo9[9], lo9[10], lo9[11]), pk_max3(lo9[12], lo9[13], lo9[14]));
    m = pk_max2(m, lo9[15]);
    // V = max(c - M, m - c) per 16-bit half (plain integers again), clamped at 0
    const unsigned v = pk_max_i16(pk_sub_i16(c, M), pk_sub_i16(m, c));
    return pk_max_i16(v, 0u);
}


__device__ __forceinline__ int mbcnt64(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// 3x3 strict NMS test of the score at (row, col), cell-local: columns outside [cs, ce) belong to the neighbouring
// cell, which cv::FAST on this cell's ROI never scores.  sc = S-1 must exceed max(0, scores of the neighbours above
// th); a neighbour at or below th is below S anyway, so this is S > max(1, raw neighbour values).
__device__ __forceinline__ bool fast_nms_ok(const uint8_t* s, int pitch, bool first, bool last, int S)
{
    int m = max(max((int)s[-pitch], (int)s[pitch]), 1);
    const int l = max(max((int)s[-pitch - 1], (int)s[-1]), (int)s[pitch - 1]);
    const int r = max(max((int)s[-pitch + 1], (int)s[1]), (int)s[pitch + 1]);
    if (!first) m = max(m, l);
    if (!last) m = max(m, r);
    return S > m;
}

// where the scores of phase B go: DENSE = score map (smapZ[row * spitch + col]); otherwise the candidate queue
struct FastSink {
    uint8_t* smapZ;             // DENSE
    int spitch;
    uint16_t* candPos;          // queue: row << 8 | col
    uint8_t* candScore;
    int candCap;
    int nCand;                  // may exceed candCap: overflow, nothing beyond candCap was stored
};

// one phase-B step: exact scores of up to 64 queued pairs (H = 0: pixels 4q, 4q+1; H = 1: 4q+2, 4q+3)
template <int H, bool DENSE>
__device__ __forceinline__ void fast_bstep(const uint32_t* tileDw, const uint16_t* Q, int head, int n, int lane, int pdw,
                                           int zLo, int zHi, int lowTh, FastSink& K)
{
    const bool act = lane < n;
    int sLo = 0, sHi = 0, row = 0, cx = 0;
    if (act) {
        const int ent = Q[(head + lane) & (FT_QRING - 1)];
        row = ent >> 8;
        const int q = ent & 0xff;
        unsigned W[7][3];
#pragma unroll
        for (int r = 0; r < 7; r++) {
            const uint32_t* p = tileDw + (row - 3 + r) * pdw + q - 1;
            W[r][0] = p[0]; W[r][1] = p[1]; W[r][2] = p[2];
        }
        const unsigned s2 = H ? fast_pair<6>(W) : fast_pair<4>(W);      // two scores, one per 16-bit half
        cx = 4 * q + 2 * H;
        sLo = (int)(s2 & 0xff);
        sHi = (int)((s2 >> 16) & 0xff);
        if (!(cx >= zLo && cx < zHi)) sLo = 0;                           // pixel of the neighbouring strip
        if (!(cx + 1 >= zLo && cx + 1 < zHi)) sHi = 0;
        if (DENSE) *reinterpret_cast<uint16_t*>(K.smapZ + row * K.spitch + cx) = (uint16_t)(sLo | (sHi << 8));
    }
    if (!DENSE) {
        const bool pLo = sLo > lowTh, pHi = sHi > lowTh;
        const unsigned long long bLo = __ballot(pLo), bHi = __ballot(pHi);
        const int nLo = __popcll(bLo), nHi = __popcll(bHi);
        if (K.nCand + nLo + nHi <= K.candCap) {
            const int e = (row << 8) | cx;
            if (pLo) {
                const int w = K.nCand + mbcnt64(bLo);
                K.candPos[w] = (uint16_t)e;
                K.candScore[w] = (uint8_t)sLo;
            }
            if (pHi) {
                const int w = K.nCand + nLo + mbcnt64(bHi);
                K.candPos[w] = (uint16_t)(e + 1);
                K.candScore[w] = (uint8_t)sHi;
            }
        }
        K.nCand += nLo + nHi;
    }
}

// ---- stage the ROI rows [y0, y0+h) as aligned 8-byte groups; item -> (row, group) advanced incrementally
__device__ __forceinline__ void fast_stage(const OrbStrip& S, const uint8_t* img, int pitch, uint32_t* tileDw, int pdw,
                                           int lane)
{
    const int nx = S.nx8, total = nx * S.h;
    int r = (int)(((unsigned)lane * S.invX8) >> 20), g = lane - r * nx;
    const int stR = S.stepG, stG = WAVE - stR * nx;
    const uint8_t* base = img + (size_t)S.y0 * pitch + (S.x0 - S.xoff);
#pragma unroll 4
    for (int i = lane; i < total; i += WAVE) {
        const uint2 v = *reinterpret_cast<const uint2*>(base + r * pitch + 8 * g);
        *reinterpret_cast<uint2*>(tileDw + r * pdw + 2 * g) = v;
        g += stG;
        r += stR;
        if (g >= nx) { g -= nx; r++; }
    }
}

// ---- phases A and B over the zone of a staged strip
template <bool DENSE>
__device__ __forceinline__ void fast_detect(const OrbStrip& S, const uint32_t* tileDw, int FT_PDW, uint16_t* pairQ,
                                            uint32_t* smapQ, int sdw, int lowTh, int lane, FastSink& K)
{
    // phase A: cheap exact rejection on every pixel pair (what cv::FAST's threshold tests amount to).
    // A 9-arc of the 16-ring contains ring pixel k or k+8 for every k, so with lo_k = min(r_k, r_k+8),
    // hi_k = max(r_k, r_k+8):   V <= U := max(I - max_k lo_k, min_k hi_k - I).
    // Only the 4 even k are used here (rows y, y+-2, y+-3: 11 dword reads instead of 21); pairs with
    // U <= lowTh in both pixels score 0 (never a corner at either threshold) and skip the exact V (about 5 in 6 pairs).
    const int zh = S.zh, zLo = S.zLo, zHi = S.zHi, qLo = S.qLo, nq = S.nq;
    const unsigned thK = (unsigned)(0x7fff - lowTh) * 0x10001u;
    const int nItems = nq * zh;
    int cntA = 0, cntB = 0, headA = 0, headB = 0;                  // wave-uniform ring state
    // item -> (zone row, quad) is advanced incrementally (64 items per step), as ONE packed counter ent = row << 8 | quad
    // (also the queue entry) and ONE running dword offset of the item's window: no per-item division or multiply
    const int ry0 = (int)(((unsigned)lane * S.invQ) >> 20), qi0 = lane - ry0 * nq;
    const int stepR = S.stepR, stepQ = WAVE - stepR * nq;
    const int qHi = qLo + nq;
    int ent = ((3 + ry0) << 8) | (qLo + qi0);
    int pOff = (3 + ry0) * FT_PDW + qLo + qi0 - 1;
    const int entStep = (stepR << 8) + stepQ, offStep = stepR * FT_PDW + stepQ;
    const int entWrap = 256 - nq, offWrap = FT_PDW - nq;
    const int offMax = (3 + zh - 1) * FT_PDW + qHi - 2;            // lanes past the last item read here (results unused)
    for (int base = 0;; base += WAVE) {
        const bool more = base < nItems;
        if (more) {
            const uint32_t* p = tileDw + min(pOff, offMax);
            const unsigned c0 = p[0], c1 = p[1], c2 = p[2];                                   // row y
            const unsigned u1 = p[-3 * FT_PDW + 1], d1 = p[3 * FT_PDW + 1];                   // rows y-3, y+3: x .. x+3
            const unsigned a0 = p[-2 * FT_PDW], a1 = p[-2 * FT_PDW + 1], a2 = p[-2 * FT_PDW + 2];   // row y-2
            const unsigned b0 = p[2 * FT_PDW], b1 = p[2 * FT_PDW + 1], b2 = p[2 * FT_PDW + 2];      // row y+2
            const bool valid = base + lane < nItems;
            if (DENSE && valid) smapQ[(ent >> 8) * sdw + (ent & 0xff)] = 0;
            unsigned u[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                // window byte index of the pair's first pixel: 4 (pixels 4q,4q+1) or 6 (4q+2,4q+3)
                const unsigned cc = h ? pick2<6>(c0, c1, c2) : pick2<4>(c0, c1, c2);
                const unsigned r0 = h ? pick2<6>(0, d1, 0) : pick2<4>(0, d1, 0);              // k=0  (0,+3)
                const unsigned r8 = h ? pick2<6>(0, u1, 0) : pick2<4>(0, u1, 0);              // k=8  (0,-3)
                const unsigned r4 = h ? pick2<9>(c0, c1, c2) : pick2<7>(c0, c1, c2);          // k=4  (+3,0)
                const unsigned r12 = h ? pick2<3>(c0, c1, c2) : pick2<1>(c0, c1, c2);         // k=12 (-3,0)
                const unsigned r2 = h ? pick2<8>(b0, b1, b2) : pick2<6>(b0, b1, b2);          // k=2  (+2,+2)
                const unsigned r10 = h ? pick2<4>(a0, a1, a2) : pick2<2>(a0, a1, a2);         // k=10 (-2,-2)
                const unsigned r6 = h ? pick2<8>(a0, a1, a2) : pick2<6>(a0, a1, a2);          // k=6  (+2,-2)
                const unsigned r14 = h ? pick2<4>(b0, b1, b2) : pick2<2>(b0, b1, b2);         // k=14 (-2,+2)
                const unsigned mlo = pk_max3(pk_min2(r0, r8), pk_min2(r4, r12), pk_max2(pk_min2(r2, r10), pk_min2(r6, r14)));
                const unsigned mhi = pk_min3(pk_max2(r0, r8), pk_max2(r4, r12), pk_min2(pk_max2(r2, r10), pk_max2(r6, r14)));
                u[h] = pk_max_i16(pk_sub_i16(cc, mlo), pk_sub_i16(mhi, cc));                  // U per 16-bit half (signed)
            }
            // a pair is queued if one of its pixels has U > lowTh: adding 0x7fff - lowTh to a signed half in
            // [-255, 255] sets bit 15 exactly then.  Pixels of a neighbouring strip inside an edge quad may queue
            // a pair needlessly; phase B zeroes their scores, so zone membership is not tested here.
            const bool pa = valid && (pk_add_u16_s(u[0], thK) & 0x80008000u) != 0;
            const bool pb = valid && (pk_add_u16_s(u[1], thK) & 0x80008000u) != 0;
            const unsigned long long ba = __ballot(pa), bb = __ballot(pb);
            if (pa) pairQ[(headA + cntA + mbcnt64(ba)) & (FT_QRING - 1)] = (uint16_t)ent;   // entries carry (row, quad) directly
            if (pb) pairQ[FT_QRING + ((headB + cntB + mbcnt64(bb)) & (FT_QRING - 1))] = (uint16_t)ent;
            cntA += __popcll(ba);
            cntB += __popcll(bb);
            ent += entStep;
            pOff += offStep;
            if ((ent & 0xff) >= qHi) { ent += entWrap; pOff += offWrap; }
        }
        // phase B: exact V for queued pairs, a full wave at a time (the last pass flushes the remainders).  LDS
        // operations of one wave execute in order; the barrier only keeps the compiler from reordering them.
        __syncthreads();
        while (cntA >= WAVE || (!more && cntA > 0)) {
            const int n = min(cntA, WAVE);
            fast_bstep<0, DENSE>(tileDw, pairQ, headA, n, lane, FT_PDW, zLo, zHi, lowTh, K);
            headA = (headA + n) & (FT_QRING - 1);
            cntA -= n;
        }
        while (cntB >= WAVE || (!more && cntB > 0)) {
            const int n = min(cntB, WAVE);
            fast_bstep<1, DENSE>(tileDw, pairQ + FT_QRING, headB, n, lane, FT_PDW, zLo, zHi, lowTh, K);
            headB = (headB + n) & (FT_QRING - 1);
            cntB -= n;
        }
        if (!more) break;
    }
    __syncthreads();
}

// key of the candidate at tile (row, col) in cell c of the strip; S1 = score + 1
#define FAST_KEY(row, col, c, S1)                                                                                         \
    (((unsigned long long)(xtab[(col) + S.cxBase] | ytab[(row) + cy0]) << ORB_KEY_PATH_SHIFT) |                           \
     ((unsigned long long)S.ci << ORB_KEY_CI_SHIFT) | ((unsigned long long)(S.cj0 + (c)) << ORB_KEY_CJ_SHIFT) | ((unsigned long long)(row) << 14) |   \
     ((unsigned long long)((col) - S.xoff - (c) * wCell) << 8) | (unsigned long long)((S1) - 1))

__global__ __launch_bounds__(WAVE) void k_fast_strips(const OrbGeom G, const uint8_t* __restrict__ pyr,
                                                      size_t pyrSlab, const OrbStrip* __restrict__ strips,
                                                      const uint32_t* __restrict__ pathTab,
                                                      unsigned long long* __restrict__ cand, size_t candSlab,
                                                      int* __restrict__ candCount, int* __restrict__ errFlags,
                                                      int* __restrict__ ovfCount, int* __restrict__ ovfList, int iniTh, int minTh, int pdw,
                                                      int rowsMax, int candCap, int nStrips, int nFrames, unsigned invPerFrame)
{
    // dynamic LDS: [pad | tile | pad | pair rings | candidate positions | candidate scores]
    extern __shared__ uint32_t fsm[];
    const int tileDwords = rowsMax * pdw;
    uint32_t* tileDw = fsm + FT_PAD;
    uint16_t* pairQ = reinterpret_cast<uint16_t*>(tileDw + tileDwords + FT_PAD);   // [2][FT_QRING]
    const int FT_PDW = pdw, FT_PITCH = 4 * pdw;
    const int lane = threadIdx.x;
    int f, si;
    if (invPerFrame) {                                             // 1-D XCD-aware grid: a frame's strips share one L2
        if (!orb_xcd_decode(blockIdx.x, (unsigned)nStrips, invPerFrame, nFrames, f, si)) return;
    } else {
        f = blockIdx.y;
        si = blockIdx.x;
    }
    const OrbStrip S = strips[si];
    const OrbLevelGeom& L = G.L[S.level];
    fast_stage(S, pyr + (size_t)f * pyrSlab + L.pyrOff, L.pitch, tileDw, FT_PDW, lane);
    __syncthreads();

    FastSink K;
    K.smapZ = nullptr; K.spitch = 0;
    K.candPos = pairQ + 2 * FT_QRING;
    K.candScore = reinterpret_cast<uint8_t*>(K.candPos + candCap);
    K.candCap = candCap;
    K.nCand = 0;
    const int lowTh = min(iniTh, minTh);
    fast_detect<false>(S, tileDw, FT_PDW, pairQ, nullptr, 0, lowTh, lane, K);
    const int nCand = K.nCand;
    if (nCand == 0) return;
    if (nCand > candCap) {                                         // redone by k_fast_strips_dense
        if (lane == 0) {
            ovfList[atomicAdd(ovfCount, 1)] = (int)(((unsigned)f << 16) | (unsigned)si);   // f, si <= 65535 (checked on the host)
            atomicAdd(&ovfCount[8 + S.level], 1);             // statistics for the host: which levels need shorter strips
        }
        return;
    }

    // ---- the tile is dead: it becomes the score map (0 everywhere but at the candidates)
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        uint4* t4 = reinterpret_cast<uint4*>(fsm);                  // pad + tile + pad: 16-byte aligned, a multiple of 16 bytes
        const int n4 = (FT_PAD + S.h * FT_PDW + FT_PAD + 3) >> 2;
        for (int i = lane; i < n4; i += WAVE) t4[i] = z;
    }
    __syncthreads();
    uint8_t* smap = reinterpret_cast<uint8_t*>(tileDw);
    for (int e = lane; e < nCand; e += WAVE) {
        const unsigned ent = K.candPos[e];
        smap[(ent >> 8) * FT_PITCH + (ent & 0xff)] = K.candScore[e];
    }
    __syncthreads();

    const int zLo = S.zLo, zHi = S.zHi, wCell = S.wCell;
    // ---- cell-local 3x3 strict NMS over the candidate queue (any order), both thresholds at once; which one counts
    // is decided per cell afterwards: minTh only where iniTh leaves NO KEYPOINT after NMS (:857-861, so a plateau of
    // equal scores that suppresses itself also triggers the fallback).
    unsigned long long keep0 = 0, keep1 = 0;                       // one bit per queue step (<= 64 steps)
    unsigned has = 0;
    {
        int it = 0;
        for (int base = 0; base < nCand; base += WAVE, it++) {
            const int e = base + lane;
            if (e < nCand) {
                const unsigned ent = K.candPos[e];
                const int row = ent >> 8, col = ent & 0xff;
                const int c = (int)(((unsigned)(col - zLo) * S.invW) >> 16);
                const int cs = zLo + c * wCell, ce = min(cs + wCell, zHi);
                const int Sv = K.candScore[e];
                const bool ok = fast_nms_ok(smap + row * FT_PITCH + col, FT_PITCH, col == cs, col == ce - 1, Sv);
                const unsigned k0 = ok && Sv > iniTh, k1 = ok && Sv > minTh;
                keep0 |= (unsigned long long)k0 << it;
                keep1 |= (unsigned long long)k1 << it;
                has |= k0 << c;
            }
        }
    }
    const unsigned fb = ~orb_wave_or(has);                         // cells without a keypoint at iniTh
    unsigned long long keepF = 0;
    int mine = 0;
    for (unsigned long long mm = keep0 | keep1; mm;) {             // the few entries of this lane that passed the NMS
        const int it = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        const int col = K.candPos[it * WAVE + lane] & 0xff;
        const unsigned c = ((unsigned)(col - zLo) * S.invW) >> 16;
        const unsigned long long k = ((((fb >> c) & 1u) ? keep1 : keep0) >> it) & 1ull;
        keepF |= k << it;
        mine += (int)k;
    }
    const int incl = orb_wave_scan_incl(mine);
    const int total = __builtin_amdgcn_readlane(incl, WAVE - 1);
    if (total == 0) return;
    int base0 = 0;
    if (lane == 0) base0 = atomicAdd(&candCount[f * ORB_MAX_LEVELS + S.level], total);
    base0 = __builtin_amdgcn_readfirstlane(base0);
    if (base0 + total > L.candCap) {                               // cannot happen: candCap is the NMS bound
        if (lane == 0) orb_flag_error(errFlags, f, 1);
        return;
    }
    const uint32_t* xtab = pathTab + L.pathXOff;
    const uint32_t* ytab = pathTab + L.pathYOff;
    unsigned long long* out = cand + (size_t)f * candSlab + L.candBase;
    const int cy0 = S.ci * L.hCell;
    int w = base0 + incl - mine;
    while (keepF) {
        const int it = __ffsll((long long)keepF) - 1;
        keepF &= keepF - 1;
        const unsigned ent = K.candPos[it * WAVE + lane];
        const int row = ent >> 8, col = ent & 0xff;
        const int c = (int)(((unsigned)(col - zLo) * S.invW) >> 16);
        const int Sv = K.candScore[it * WAVE + lane];
        out[w++] = FAST_KEY(row, col, c, Sv);
    }
}


// =====================================================================================================================
// k_fast_strips_p<P>: the same strip detector with a COMPILE-TIME tile pitch of P dwords (P % 8 == 4) and leaner
// bookkeeping around the same arithmetic (round 3: the kernel runs at the issue rate of its instruction mix, so the
// only lever is fewer vector instructions; r02 counters: 1980 per strip-wave, of which staging 150, phase A 84 per 64
// quads, phase B 145 per 64 pairs).  What changed against k_fast_strips:
//   * every LDS access is (one address VGPR + immediate offset): the 11 / 21 window reads of phases A / B need no
//     per-row address arithmetic;
//   * staging moves 16-byte chunks, 64 / CL rows per step, with loop-invariant lane -> (row, chunk) decode: ~3 vector
//     instructions per load instead of 15;
//   * phase A walks the zone in blocks of 8 rows, COLUMN-major inside a block (lane -> row = lane & 7, quad += 8 per step):
//     with P = 4 (mod 8) the 32 lanes of an LDS access group cover 4 columns x 8 rows = 32 distinct banks (row-major
//     enumeration put a few lanes of the next row on the banks of the previous one: every read 2-way conflicted), and
//     the whole per-lane state is ONE register ((window address << 8) | quad): add, compare, select, add, shift;
//   * the queue entry of a surviving pair IS the LDS address of its 7 x 3-dword window; phase B decodes (row, column)
//     from it only for the few pixels that become candidates (constant division by the pitch);
//   * pixels of an edge quad that lie outside the zone are not tested in phase B any more: their candidates are dropped
//     when the candidates are scattered into the score map (their queue score is zeroed, so the NMS ignores them).
// Same results as k_fast_strips bit for bit (tests/test_gpu_extractor.py::test_fast_strip_sizes_and_queue_overflow runs
// both); geometries with strips wider than the largest instantiated pitch use the generic kernel.
#define F2_RING 128                    // entries of one pair ring

template <int P>
struct FastP {
    static constexpr int ROWB = 4 * P;                                     // tile row pitch in bytes
    static constexpr int HDR = ((2 * F2_RING * 2 + 16 + ROWB - 1) / ROWB) * ROWB;   // rings + pad, a multiple of ROWB
    static constexpr int CL = (P / 4 <= 8) ? 8 : 16;                       // lanes per staged row (16-byte chunks)
    static constexpr int RI = 64 / CL;                                     // rows per staging step
    static constexpr int MAXIT = (66 + RI - 1) / RI;                       // strips are at most 66 rows tall
    static_assert(P % 8 == 4 && P / 4 <= 16, "pitch must be 4 (mod 8) dwords and at most 16 chunks");
    static_assert(HDR % 16 == 0 && ROWB % 16 == 0, "16-byte aligned tile rows");
};
size_t orb_fast_p_lds_bytes(int P, int rowsMax, int candCap)
{
    const int rowb = 4 * P, hdr = ((2 * F2_RING * 2 + 16 + rowb - 1) / rowb) * rowb;
    // phase A's last block may run up to 7 rows past the zone: those windows are read (never used); they land in the
    // candidate arrays behind the tile, which therefore cover at least 8 tile rows
    const size_t tail = std::max<size_t>((size_t)3 * candCap + 16, (size_t)8 * rowb + 16);
    return (size_t)hdr + (size_t)rowsMax * rowb + 16 + tail;
}

typedef unsigned int orb_u32x4 __attribute__((ext_vector_type(4)));
typedef orb_u32x4 __attribute__((aligned(8))) orb_u32x4_a8;
// LDS accesses of the hot loops by raw byte address (address space 3 pointers made from integers): the kernel owns no
// static LDS, so its dynamic block starts at address 0 (checked once at the top of the kernel) and no access pays an add
// of the block's (relocatable) base
typedef __attribute__((address_space(3))) uint32_t orb_lds_u32;
typedef __attribute__((address_space(3))) uint16_t orb_lds_u16;
__device__ __forceinline__ const orb_lds_u32* lds_dw(unsigned byteAddr) { return reinterpret_cast<const orb_lds_u32*>(byteAddr); }
__device__ __forceinline__ orb_lds_u16* lds_hw(unsigned byteAddr) { return reinterpret_cast<orb_lds_u16*>(byteAddr); }

template <int P, int OCC>
__global__ __launch_bounds__(WAVE) void k_fast_strips_p(const OrbGeom G, const uint8_t* __restrict__ pyr, size_t pyrSlab,
                                                        const OrbStrip* __restrict__ strips,
                                                        const uint32_t* __restrict__ pathTab,
                                                        unsigned long long* __restrict__ cand, size_t candSlab,
                                                        int* __restrict__ candCount, int* __restrict__ errFlags,
                                                        int* __restrict__ ovfCount, int* __restrict__ ovfList, int iniTh,
                                                        int minTh, int rowsMax, int candCap, int nStrips, int nFrames,
                                                        unsigned invPerFrame)
{
    // OCC: a clobbered high register makes the kernel claim 80 / 96 / 128 VGPRs, i.e. at most 6 / 5 / 4 waves per SIMD (see
    // orb_launch_fast_strips: 5 balances the CU's SIMDs; 4 -- 48 KB of LDS left to other kernels' workgroups -- was tried for
    // co-scheduling with the other lanes' kernels and lost: 0.498 ms, step 1.348 against 1.311 ms)
    if (OCC == 6) asm volatile("" ::: "v79");
    if (OCC == 5) asm volatile("" ::: "v95");
    if (OCC == 4) asm volatile("" ::: "v127");
    // dynamic LDS (bytes): [ring A 256 | ring B 256 | pad .. HDR) | tile rowsMax x ROWB | 16 | candPos 2 candCap | candScore candCap]
    extern __shared__ uint32_t fsm[];
    constexpr int ROWB = FastP<P>::ROWB, TB = FastP<P>::HDR;
    uint8_t* lds = reinterpret_cast<uint8_t*>(fsm);
    const int lane = threadIdx.x;
    int f, si;
    if (invPerFrame) {                                             // 1-D XCD-aware grid: a frame's strips share one L2
        if (!orb_xcd_decode(blockIdx.x, (unsigned)nStrips, invPerFrame, nFrames, f, si)) return;
    } else {
        f = blockIdx.y;
        si = blockIdx.x;
    }
    if ((unsigned)(uintptr_t)((__attribute__((address_space(3))) uint32_t*)fsm) != 0u) {   // see lds_dw
        if (lane == 0) orb_flag_error(errFlags, f, 1);
        return;
    }
    // The 48-byte strip record as THREE SCALAR loads (round 5): its fields are bytes and shorts, which the compiler fetched with
    // five vector loads off a vector index -- a vector-memory round trip at the head of every wave, in front of the staging
    // loads that depend on it; the index is wave-uniform, the scalar cache serves the record (shared by all frames) at once.
    f = __builtin_amdgcn_readfirstlane(f);
    si = __builtin_amdgcn_readfirstlane(si);
    OrbStrip S;
    {
        const uint4* rec = reinterpret_cast<const uint4*>(strips + si);
        const uint4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        uint4 tmp[3] = {r0, r1, r2};
        __builtin_memcpy(&S, tmp, sizeof(S));
    }
    const OrbLevelGeom& L = G.L[S.level];

    // ---- stage rows [y0, y0 + h) from the 8-byte aligned column x0 - xoff, ROWB bytes per row, 16 bytes per lane
    {
        constexpr int CL = FastP<P>::CL, RI = FastP<P>::RI, MAXIT = FastP<P>::MAXIT, NC = P / 4;
        const int rr = lane / CL, c = lane % CL;
        const int h = S.h, pitch = L.pitch;
        const uint8_t* src = pyr + (size_t)f * pyrSlab + L.pyrOff + (size_t)(S.y0 + rr) * pitch + (S.x0 - S.xoff) + 16 * c;
        uint8_t* dst = lds + TB + rr * ROWB + 16 * c;
        const bool act = c < NC;
        orb_u32x4 v[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; it++)
            if (it * RI < h) {                                     // (uniform)
                if (act && it * RI + rr < h) v[it] = *reinterpret_cast<const orb_u32x4_a8*>(src + (size_t)it * RI * pitch);
            }
#pragma unroll
        for (int it = 0; it < MAXIT; it++)
            if (it * RI < h) {
                if (act && it * RI + rr < h) *reinterpret_cast<orb_u32x4*>(dst + it * RI * ROWB) = v[it];
            }
    }
    __syncthreads();

    uint16_t* ringA = reinterpret_cast<uint16_t*>(lds);
    uint16_t* ringB = ringA + F2_RING;
    const int CB = TB + rowsMax * ROWB + 16;                       // candidate positions (u16: row << 8 | col), then scores (u8)
    uint16_t* candPos = reinterpret_cast<uint16_t*>(lds + CB);
    uint8_t* candScore = lds + CB + 2 * candCap;
    const int lowTh = min(iniTh, minTh);
    int nCand = 0;

    // ---- phases A and B (see fast_detect): x = (LDS byte address of the window's first dword << 8) | quad index in the zone
    {
        const int zh = S.zh, nq = S.nq, qLo = S.qLo;
        unsigned thKV = (unsigned)(0x7fff - lowTh) * 0x10001u;
        asm volatile("" : "+v"(thKV));                             // (a vector register: no scalar operand in the loop's packed adds)
        const int nBlocks = (zh + 7) >> 3, nItems = nBlocks * 8 * nq;
        // item j = base + lane -> column c = j >> 3 over all blocks, block = c / nq, quad = c % nq, row = 8 block + (lane & 7).
        // Strips of >= 8 quads advance incrementally (a step of 8 columns wraps into the next block at most once); the few
        // narrower ones (clipped last cells) decode every step from scratch.
        const bool narrow = nq < 8;
        const float rcpNq = __frcp_rn((float)nq);
        auto decode = [&](int base) -> unsigned {
            const int c = (base + lane) >> 3;
            const int blk = (int)(((float)c + 0.5f) * rcpNq);       // c <= 600, nq <= 7: 0.5 / nq is far above the rounding error
            const int qi = c - blk * nq;
            return ((unsigned)(TB + ((blk * 8 + (lane & 7)) * P + qLo + qi - 1) * 4) << 8) | (unsigned)qi;
        };
        unsigned x = narrow ? decode(0) : ((unsigned)(TB + ((lane & 7) * P + qLo + (lane >> 3) - 1) * 4) << 8) | (unsigned)(lane >> 3);
        const unsigned xStep = (32u << 8) | 8u;
        unsigned xWrapV = ((unsigned)(8 * ROWB - 4 * nq) << 8) - (unsigned)nq;
        asm volatile("" : "+v"(xWrapV));                           // stays in a vector register (v_cndmask cannot take it as a scalar)
        const unsigned aEnd = (unsigned)(TB + zh * ROWB - 4);       // windows of zone rows start below this address (quad - 1 >= -1)
        int cntA = 0, cntB = 0, headA = 0, headB = 0;               // wave-uniform ring state
        constexpr unsigned kRowMagic = (unsigned)(((1ull << 32) + ROWB - 1) / ROWB);   // floor(a / ROWB) for a < 2^16
        // Htag 0 / 1: n entries of ring A (left pairs of their quads) / ring B (right pairs).  Htag 2: the LAST step of a strip,
        // the remainders of both rings at once -- lanes [0, n) take ring A, lanes [n, n + nB2) ring B with their windows
        // shifted down by two bytes, so that every lane scores the pair at window bytes 4, 5 (21 extra instructions once per
        // strip instead of a second, mostly empty step of ~124)
        auto bstep = [&](auto Htag, const uint16_t* ring, int head, int n, int nB2) {
            constexpr int H = decltype(Htag)::value;
            const bool fromB = H == 2 && lane >= n;
            const bool act = lane < n + (H == 2 ? nB2 : 0);
            unsigned s2 = 0, a = 0;
            if (act) {
                a = fromB ? ringB[(headB + lane - n) & (F2_RING - 1)] : ring[(head + lane) & (F2_RING - 1)];
                const orb_lds_u32* p = lds_dw(a);
                unsigned W[7][3];
#pragma unroll
                for (int r = 0; r < 7; r++) { W[r][0] = p[r * P]; W[r][1] = p[r * P + 1]; W[r][2] = p[r * P + 2]; }
                if (H == 2) {
                    const unsigned shB = fromB ? 2u : 0u;
#pragma unroll
                    for (int r = 0; r < 7; r++) {
                        W[r][0] = __builtin_amdgcn_alignbyte(W[r][1], W[r][0], shB);
                        W[r][1] = __builtin_amdgcn_alignbyte(W[r][2], W[r][1], shB);
                        W[r][2] >>= 8u * shB;
                    }
                }
                s2 = H == 1 ? fast_pair<6>(W) : fast_pair<4>(W);    // two scores, one per 16-bit half
            }
            const int sLo = (int)(s2 & 0xffffu), sHi = (int)(s2 >> 16);
            const bool pLo = sLo > lowTh, pHi = sHi > lowTh;
            const unsigned long long bLo = __ballot(pLo), bHi = __ballot(pHi);
            if ((bLo | bHi) == 0) return;
            const int nLo = __popcll(bLo), nHi = __popcll(bHi);
            if (nCand + nLo + nHi <= candCap) {
                // (row, col) of the pair's first pixel: a + 4 = TB + zoneRow * ROWB + 4 quad
                const unsigned a4 = a + 4u, rowAbs = __umulhi(a4, kRowMagic);
                const unsigned e = ((rowAbs + (unsigned)(3 - TB / ROWB)) << 8) + (a4 - rowAbs * (unsigned)ROWB) +
                                   (H == 2 ? (fromB ? 2u : 0u) : (unsigned)(2 * H));
                if (pLo) {
                    const int w = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bLo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bLo, (unsigned)nCand));
                    candPos[w] = (uint16_t)e;
                    candScore[w] = (uint8_t)sLo;
                }
                if (pHi) {
                    const int w = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bHi >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bHi, (unsigned)(nCand + nLo)));
                    candPos[w] = (uint16_t)(e + 1);
                    candScore[w] = (uint8_t)sHi;
                }
            }
            nCand += nLo + nHi;
        };
        for (int base = 0;; base += WAVE) {
            const bool more = base < nItems;
            if (more) {
                const unsigned a = x >> 8;
                const orb_lds_u32* p = lds_dw(a);
                const unsigned c0 = p[3 * P], c1 = p[3 * P + 1], c2 = p[3 * P + 2];           // row y
                const unsigned u1 = p[1], d1 = p[6 * P + 1];                                  // rows y-3, y+3: x .. x+3
                const unsigned a0 = p[P], a1 = p[P + 1], a2 = p[P + 2];                       // row y-2
                const unsigned b0 = p[5 * P], b1 = p[5 * P + 1], b2 = p[5 * P + 2];           // row y+2
                unsigned u[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const unsigned cc = h ? pick2<6>(c0, c1, c2) : pick2<4>(c0, c1, c2);
                    const unsigned r0 = h ? pick2<6>(0, d1, 0) : pick2<4>(0, d1, 0);              // k=0  (0,+3)
                    const unsigned r8 = h ? pick2<6>(0, u1, 0) : pick2<4>(0, u1, 0);              // k=8  (0,-3)
                    const unsigned r4 = h ? pick2<9>(c0, c1, c2) : pick2<7>(c0, c1, c2);          // k=4  (+3,0)
                    const unsigned r12 = h ? pick2<3>(c0, c1, c2) : pick2<1>(c0, c1, c2);         // k=12 (-3,0)
                    const unsigned r2 = h ? pick2<8>(b0, b1, b2) : pick2<6>(b0, b1, b2);          // k=2  (+2,+2)
                    const unsigned r10 = h ? pick2<4>(a0, a1, a2) : pick2<2>(a0, a1, a2);         // k=10 (-2,-2)
                    const unsigned r6 = h ? pick2<8>(a0, a1, a2) : pick2<6>(a0, a1, a2);          // k=6  (+2,-2)
                    const unsigned r14 = h ? pick2<4>(b0, b1, b2) : pick2<2>(b0, b1, b2);         // k=14 (-2,+2)
                    const unsigned mlo = pk_max3(pk_min2(r0, r8), pk_min2(r4, r12), pk_max2(pk_min2(r2, r10), pk_min2(r6, r14)));
                    const unsigned mhi = pk_min3(pk_max2(r0, r8), pk_max2(r4, r12), pk_min2(pk_max2(r2, r10), pk_max2(r6, r14)));
                    u[h] = pk_max_i16(pk_sub_i16(cc, mlo), pk_sub_i16(mhi, cc));                  // U per 16-bit half (signed)
                }
                // a pair is queued if one of its pixels has U > lowTh (bit 15 of the half after adding 0x7fff - lowTh) and
                // it lies on a zone row (the last block of 8 rows may run past the zone)
                // (the ballots of the two compares are ANDed as scalars: a ballot of their conjunction makes the compiler
                // materialise the predicate in a register)
                const bool valid = a < aEnd;
                const bool qa = (pk_add_u16_n(u[0], thKV) & 0x80008000u) != 0, qb = (pk_add_u16_n(u[1], thKV) & 0x80008000u) != 0;
                const unsigned long long bv = __ballot(valid), ba = __ballot(qa) & bv, bb = __ballot(qb) & bv;
                // ring slot: ((head + cnt + rank among the queued lanes) mod 128) as a byte offset
                const unsigned ra = __builtin_amdgcn_mbcnt_hi((unsigned)(ba >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ba, 0u));
                const unsigned rb = __builtin_amdgcn_mbcnt_hi((unsigned)(bb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bb, 0u));
                if (qa & valid) *lds_hw(((ra << 1) + (unsigned)(2 * (headA + cntA))) & (2 * F2_RING - 2)) = (uint16_t)a;
                if (qb & valid) lds_hw(((rb << 1) + (unsigned)(2 * (headB + cntB))) & (2 * F2_RING - 2))[F2_RING] = (uint16_t)a;
                cntA += __popcll(ba);
                cntB += __popcll(bb);
                if (!narrow) {
                    x += xStep;
                    x += ((x & 0xffu) >= (unsigned)nq) ? xWrapV : 0u;
                } else {
                    x = decode(base + WAVE);
                }
            }
            __syncthreads();       // LDS operations of one wave execute in order; this only keeps the compiler from reordering
            while (cntA >= WAVE) {
                bstep(std::integral_constant<int, 0>(), ringA, headA, WAVE, 0);
                headA = (headA + WAVE) & (F2_RING - 1);
                cntA -= WAVE;
            }
            while (cntB >= WAVE) {
                bstep(std::integral_constant<int, 1>(), ringB, headB, WAVE, 0);
                headB = (headB + WAVE) & (F2_RING - 1);
                cntB -= WAVE;
            }
            if (!more) {                                           // the remainders (< 64 each)
                if (cntA > 0 && cntB > 0 && cntA + cntB <= WAVE) {
                    bstep(std::integral_constant<int, 2>(), ringA, headA, cntA, cntB);
                } else {
                    if (cntA > 0) bstep(std::integral_constant<int, 0>(), ringA, headA, cntA, 0);
                    if (cntB > 0) bstep(std::integral_constant<int, 1>(), ringB, headB, cntB, 0);
                }
                break;
            }
        }
        __syncthreads();
    }
    if (nCand == 0) return;
    if (nCand > candCap) {                                         // redone by k_fast_strips_dense
        if (lane == 0) {
            ovfList[atomicAdd(ovfCount, 1)] = (int)(((unsigned)f << 16) | (unsigned)si);
            atomicAdd(&ovfCount[8 + S.level], 1);
        }
        return;
    }

    // ---- the tile is dead: it becomes the score map (0 everywhere but at the candidates inside the zone)
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        uint4* t4 = reinterpret_cast<uint4*>(lds + TB - 16);       // one row of slack on both sides of the rows the NMS reads
        const int n4 = (16 + S.h * ROWB + 16) >> 4;
        for (int i = lane; i < n4; i += WAVE) t4[i] = z;
    }
    __syncthreads();
    uint8_t* smap = lds + TB;
    const int zLo = S.zLo, zHi = S.zHi, wCell = S.wCell;
    for (int e = lane; e < nCand; e += WAVE) {
        const unsigned ent = candPos[e];
        const unsigned col = ent & 0xff;
        if (col - (unsigned)zLo < (unsigned)(zHi - zLo)) smap[(ent >> 8) * ROWB + col] = candScore[e];
        else candScore[e] = 0;                                     // pixel of the neighbouring strip inside an edge quad
    }
    __syncthreads();

    // ---- cell-local 3x3 strict NMS over the candidate queue, both thresholds at once (as k_fast_strips)
    unsigned long long keep0 = 0, keep1 = 0;                       // one bit per queue step (<= 64 steps)
    unsigned has = 0;
    {
        int it = 0;
        for (int base = 0; base < nCand; base += WAVE, it++) {
            const int e = base + lane;
            if (e < nCand) {
                const unsigned ent = candPos[e];
                const int row = ent >> 8, col = ent & 0xff;
                const int c = (int)(((unsigned)(col - zLo) * S.invW) >> 16);
                const int cs = zLo + c * wCell, ce = min(cs + wCell, zHi);
                const int Sv = candScore[e];
                const bool ok = fast_nms_ok(smap + row * ROWB + col, ROWB, col == cs, col == ce - 1, Sv);
                const unsigned k0 = ok && Sv > iniTh, k1 = ok && Sv > minTh;
                keep0 |= (unsigned long long)k0 << it;
                keep1 |= (unsigned long long)k1 << it;
                has |= k0 << c;
            }
        }
    }
    const unsigned fb = ~orb_wave_or(has);                         // cells without a keypoint at iniTh
    unsigned long long keepF = 0;
    int mine = 0;
    for (unsigned long long mm = keep0 | keep1; mm;) {             // the few entries of this lane that passed the NMS
        const int it = __ffsll((long long)mm) - 1;
        mm &= mm - 1;
        const int col = candPos[it * WAVE + lane] & 0xff;
        const unsigned c = ((unsigned)(col - zLo) * S.invW) >> 16;
        const unsigned long long k = ((((fb >> c) & 1u) ? keep1 : keep0) >> it) & 1ull;
        keepF |= k << it;
        mine += (int)k;
    }
    const int incl = orb_wave_scan_incl(mine);
    const int total = __builtin_amdgcn_readlane(incl, WAVE - 1);
    if (total == 0) return;
    int base0 = 0;
    if (lane == 0) base0 = atomicAdd(&candCount[f * ORB_MAX_LEVELS + S.level], total);
    base0 = __builtin_amdgcn_readfirstlane(base0);
    if (base0 + total > L.candCap) {                               // cannot happen: candCap is the NMS bound
        if (lane == 0) orb_flag_error(errFlags, f, 1);
        return;
    }
    const uint32_t* xtab = pathTab + L.pathXOff;
    const uint32_t* ytab = pathTab + L.pathYOff;
    unsigned long long* out = cand + (size_t)f * candSlab + L.candBase;
    const int cy0 = S.ci * L.hCell;
    int w = base0 + incl - mine;
    while (keepF) {
        const int it = __ffsll((long long)keepF) - 1;
        keepF &= keepF - 1;
        const unsigned ent = candPos[it * WAVE + lane];
        const int row = ent >> 8, col = ent & 0xff;
        const int c = (int)(((unsigned)(col - zLo) * S.invW) >> 16);
        const int Sv = candScore[it * WAVE + lane];
        out[w++] = FAST_KEY(row, col, c, Sv);
    }
}

// =====================================================================================================================
// k_fast_strips_mw<P>: the strip detector of k_fast_strips_p for launches of a FEW frames (single frames, config 5's
// stream), where a frame's ~300 strips are all the chip has to do and the time of the launch is the time of ONE strip's
// chain of ~1600 dependent-ish instructions on one wave.  Here FOUR waves share a strip: the tile is staged by all 256
// threads, the steps of phase A (8 rows x 8 quads each) are dealt round-robin over the waves, every wave keeps its own
// two pair rings and drains them itself (phase B), candidates go to ONE queue through an LDS counter (their order is
// irrelevant: the NMS reads the score map, the quadtree sorts the keys), and the tail (zeroing, scatter, NMS, emission)
// runs over the queue with 256 threads.  Same arithmetic, same results; real barriers between the stages.
#define MW_WAVES 4
template <int P>
struct FastMW {
    static constexpr int ROWB = 4 * P;
    static constexpr int RINGB = MW_WAVES * 2 * F2_RING * 2;                // per wave: ring A 256 B | ring B 256 B
    static constexpr int HDR = ((RINGB + 16 + ROWB - 1) / ROWB) * ROWB;
    static constexpr int CL = FastP<P>::CL;
    static constexpr int RI = (WAVE * MW_WAVES) / CL;                       // rows per staging step
    static constexpr int MAXIT = (66 + RI - 1) / RI;
};
size_t orb_fast_mw_lds_bytes(int P, int rowsMax, int candCap)
{
    const int rowb = 4 * P, hdr = ((MW_WAVES * 2 * F2_RING * 2 + 16 + rowb - 1) / rowb) * rowb;
    const size_t tail = std::max<size_t>((((size_t)3 * candCap + 3) & ~(size_t)3) + 32, (size_t)8 * rowb + 32);   // (see orb_fast_p_lds_bytes)
    return (size_t)hdr + (size_t)rowsMax * rowb + 16 + tail;
}

template <int P>
__device__ __forceinline__ void fast_mw_strip(uint32_t* fsm, const OrbGeom& G, const uint8_t* __restrict__ pyr, size_t pyrSlab,
                                              const OrbStrip& S, int f, int si,
                                              const uint32_t* __restrict__ pathTab,
                                              unsigned long long* __restrict__ cand, size_t candSlab,
                                              int* __restrict__ candCount, int* __restrict__ errFlags,
                                              int* __restrict__ ovfCount, int* __restrict__ ovfList, int iniTh,
                                              int minTh, int rowsMax, int candCap)
{
    // dynamic LDS (bytes): [4 x (ring A 256 | ring B 256) | pad .. HDR) | tile rowsMax x ROWB | 16 | candPos 2 candCap |
    //                       candScore candCap (rounded up to 4) | shared: candidate counter, cells-with-a-keypoint mask, the
    //                       waves' keypoint counts, the output base]
    constexpr int ROWB = FastMW<P>::ROWB, TB = FastMW<P>::HDR, T = WAVE * MW_WAVES;
    uint8_t* lds = reinterpret_cast<uint8_t*>(fsm);
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((unsigned)(uintptr_t)((__attribute__((address_space(3))) uint32_t*)fsm) != 0u) {   // see lds_dw
        if (tid == 0) orb_flag_error(errFlags, f, 1);
        return;
    }
    const OrbLevelGeom& L = G.L[S.level];
    const int CB = TB + rowsMax * ROWB + 16;                       // candidate positions (u16: row << 8 | col), then scores (u8)
    uint16_t* candPos = reinterpret_cast<uint16_t*>(lds + CB);
    uint8_t* candScore = lds + CB + 2 * candCap;
    int* shared = reinterpret_cast<int*>(lds + CB + ((3 * candCap + 3) & ~3));
    if (tid == 0) { shared[0] = 0; shared[1] = 0; }

    // ---- stage rows [y0, y0 + h): 16 bytes per thread, T / CL rows per step.  UNCONDITIONAL loads (a thread outside the tile
    // re-reads the tile's last row / last chunk): with `if (...) v[it] = load` the compiler waited for every load before it
    // issued the next -- three round trips in a row at the head of a kernel whose whole life is ~10 us (round 5, tools/isa_waits.py)
    {
        constexpr int CL = FastMW<P>::CL, RI = FastMW<P>::RI, MAXIT = FastMW<P>::MAXIT, NC = P / 4;
        const int rr = tid / CL, c = tid % CL;
        const int h = S.h, pitch = L.pitch;
        const uint8_t* src = pyr + (size_t)f * pyrSlab + L.pyrOff + (size_t)S.y0 * pitch + (S.x0 - S.xoff) + 16 * min(c, NC - 1);
        uint8_t* dst = lds + TB + rr * ROWB + 16 * c;
        const bool act = c < NC;
        orb_u32x4 v[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; it++)
            v[it] = *reinterpret_cast<const orb_u32x4_a8*>(src + (size_t)min(it * RI + rr, h - 1) * pitch);
#pragma unroll
        for (int it = 0; it < MAXIT; it++)
            if (it * RI < h) {
                if (act && it * RI + rr < h) *reinterpret_cast<orb_u32x4*>(dst + it * RI * ROWB) = v[it];
            }
    }
    __syncthreads();

    const int lowTh = min(iniTh, minTh);
    {
        const unsigned ringBase = (unsigned)wv * (unsigned)(4 * F2_RING);      // this wave's rings (bytes)
        const uint16_t* ringA = reinterpret_cast<const uint16_t*>(lds + ringBase);
        const uint16_t* ringB = ringA + F2_RING;
        const int zh = S.zh, nq = S.nq, qLo = S.qLo;
        unsigned thKV = (unsigned)(0x7fff - lowTh) * 0x10001u;
        asm volatile("" : "+v"(thKV));
        const int nsc = (nq + 7) >> 3, nSteps = ((zh + 7) >> 3) * nsc;        // steps of 8 rows x 8 quads; wave w takes w, w + 4, ...
        int cntA = 0, cntB = 0, headA = 0, headB = 0;                           // wave-uniform ring state
        constexpr unsigned kRowMagic = (unsigned)(((1ull << 32) + ROWB - 1) / ROWB);
        auto bstep = [&](auto Htag, int head, int n, int nB2) {                 // (as in k_fast_strips_p)
            constexpr int H = decltype(Htag)::value;
            const bool fromB = H == 2 && lane >= n;
            const bool act = lane < n + (H == 2 ? nB2 : 0);
            unsigned s2 = 0, a = 0;
            if (act) {
                a = fromB ? ringB[(headB + lane - n) & (F2_RING - 1)] : (H == 1 ? ringB : ringA)[(head + lane) & (F2_RING - 1)];
                const orb_lds_u32* p = lds_dw(a);
                unsigned W[7][3];
#pragma unroll
                for (int r = 0; r < 7; r++) { W[r][0] = p[r * P]; W[r][1] = p[r * P + 1]; W[r][2] = p[r * P + 2]; }
                if (H == 2) {
                    const unsigned shB = fromB ? 2u : 0u;
#pragma unroll
                    for (int r = 0; r < 7; r++) {
                        W[r][0] = __builtin_amdgcn_alignbyte(W[r][1], W[r][0], shB);
                        W[r][1] = __builtin_amdgcn_alignbyte(W[r][2], W[r][1], shB);
                        W[r][2] >>= 8u * shB;
                    }
                }
                s2 = H == 1 ? fast_pair<6>(W) : fast_pair<4>(W);
            }
            const int sLo = (int)(s2 & 0xffffu), sHi = (int)(s2 >> 16);
            const bool pLo = sLo > lowTh, pHi = sHi > lowTh;
            const unsigned long long bLo = __ballot(pLo), bHi = __ballot(pHi);
            if ((bLo | bHi) == 0) return;
            const int nLo = __popcll(bLo), nHi = __popcll(bHi);
            int base = 0;
            if (lane == 0) base = atomicAdd(&shared[0], nLo + nHi);            // the workgroup's candidate queue
            base = __builtin_amdgcn_readfirstlane(base);
            if (base + nLo + nHi <= candCap) {
                const unsigned a4 = a + 4u, rowAbs = __umulhi(a4, kRowMagic);
                const unsigned e = ((rowAbs + (unsigned)(3 - TB / ROWB)) << 8) + (a4 - rowAbs * (unsigned)ROWB) +
                                   (H == 2 ? (fromB ? 2u : 0u) : (unsigned)(2 * H));
                if (pLo) {
                    const int w = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bLo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bLo, (unsigned)base));
                    candPos[w] = (uint16_t)e;
                    candScore[w] = (uint8_t)sLo;
                }
                if (pHi) {
                    const int w = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bHi >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bHi, (unsigned)(base + nLo)));
                    candPos[w] = (uint16_t)(e + 1);
                    candScore[w] = (uint8_t)sHi;
                }
            }
        };
        for (int st = wv;; st += MW_WAVES) {
            const bool more = st < nSteps;
            if (more) {
                const int blk = st / nsc, cs = (st - blk * nsc) * 8;            // (wave-uniform)
                const int col = cs + (lane >> 3), row = blk * 8 + (lane & 7);
                const bool valid = col < nq && row < zh;
                const unsigned a = (unsigned)(TB + (row * P + qLo + min(col, nq - 1) - 1) * 4);
                const orb_lds_u32* p = lds_dw(a);
                const unsigned c0 = p[3 * P], c1 = p[3 * P + 1], c2 = p[3 * P + 2];           // row y
                const unsigned u1 = p[1], d1 = p[6 * P + 1];                                  // rows y-3, y+3: x .. x+3
                const unsigned a0 = p[P], a1 = p[P + 1], a2 = p[P + 2];                       // row y-2
                const unsigned b0 = p[5 * P], b1 = p[5 * P + 1], b2 = p[5 * P + 2];           // row y+2
                unsigned u[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const unsigned cc = h ? pick2<6>(c0, c1, c2) : pick2<4>(c0, c1, c2);
                    const unsigned r0 = h ? pick2<6>(0, d1, 0) : pick2<4>(0, d1, 0);
                    const unsigned r8 = h ? pick2<6>(0, u1, 0) : pick2<4>(0, u1, 0);
                    const unsigned r4 = h ? pick2<9>(c0, c1, c2) : pick2<7>(c0, c1, c2);
                    const unsigned r12 = h ? pick2<3>(c0, c1, c2) : pick2<1>(c0, c1, c2);
                    const unsigned r2 = h ? pick2<8>(b0, b1, b2) : pick2<6>(b0, b1, b2);
                    const unsigned r10 = h ? pick2<4>(a0, a1, a2) : pick2<2>(a0, a1, a2);
                    const unsigned r6 = h ? pick2<8>(a0, a1, a2) : pick2<6>(a0, a1, a2);
                    const unsigned r14 = h ? pick2<4>(b0, b1, b2) : pick2<2>(b0, b1, b2);
                    const unsigned mlo = pk_max3(pk_min2(r0, r8), pk_min2(r4, r12), pk_max2(pk_min2(r2, r10), pk_min2(r6, r14)));
                    const unsigned mhi = pk_min3(pk_max2(r0, r8), pk_max2(r4, r12), pk_min2(pk_max2(r2, r10), pk_max2(r6, r14)));
                    u[h] = pk_max_i16(pk_sub_i16(cc, mlo), pk_sub_i16(mhi, cc));
                }
                const bool qa = (pk_add_u16_n(u[0], thKV) & 0x80008000u) != 0, qb = (pk_add_u16_n(u[1], thKV) & 0x80008000u) != 0;
                const unsigned long long bv = __ballot(valid), ba = __ballot(qa) & bv, bb = __ballot(qb) & bv;
                const unsigned ra = __builtin_amdgcn_mbcnt_hi((unsigned)(ba >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ba, 0u));
                const unsigned rb = __builtin_amdgcn_mbcnt_hi((unsigned)(bb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bb, 0u));
                if (qa & valid) *lds_hw(ringBase + (((ra << 1) + (unsigned)(2 * (headA + cntA))) & (2 * F2_RING - 2))) = (uint16_t)a;
                if (qb & valid) *lds_hw(ringBase + 2 * F2_RING + (((rb << 1) + (unsigned)(2 * (headB + cntB))) & (2 * F2_RING - 2))) = (uint16_t)a;
                cntA += __popcll(ba);
                cntB += __popcll(bb);
            }
            // LDS operations of one wave execute in order: this only keeps the compiler from moving the ring reads of phase B
            // above the ring writes (NOT a workgroup barrier: the waves run different numbers of steps)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            while (cntA >= WAVE) {
                bstep(std::integral_constant<int, 0>(), headA, WAVE, 0);
                headA = (headA + WAVE) & (F2_RING - 1);
                cntA -= WAVE;
            }
            while (cntB >= WAVE) {
                bstep(std::integral_constant<int, 1>(), headB, WAVE, 0);
                headB = (headB + WAVE) & (F2_RING - 1);
                cntB -= WAVE;
            }
            if (!more) {
                if (cntA > 0 && cntB > 0 && cntA + cntB <= WAVE) {
                    bstep(std::integral_constant<int, 2>(), headA, cntA, cntB);
                } else {
                    if (cntA > 0) bstep(std::integral_constant<int, 0>(), headA, cntA, 0);
                    if (cntB > 0) bstep(std::integral_constant<int, 1>(), headB, cntB, 0);
                }
                break;
            }
        }
    }
    __syncthreads();
    const int nCand = shared[0];
    if (nCand == 0) return;
    if (nCand > candCap) {                                         // redone by k_fast_strips_dense
        if (tid == 0) {
            ovfList[atomicAdd(ovfCount, 1)] = (int)(((unsigned)f << 16) | (unsigned)si);
            atomicAdd(&ovfCount[8 + S.level], 1);
        }
        return;
    }

    // ---- the tile is dead: it becomes the score map (0 everywhere but at the candidates inside the zone)
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        uint4* t4 = reinterpret_cast<uint4*>(lds + TB - 16);
        const int n4 = (16 + S.h * ROWB + 16) >> 4;
        for (int i = tid; i < n4; i += T) t4[i] = z;
    }
    __syncthreads();
    uint8_t* smap = lds + TB;
    const int zLo = S.zLo, zHi = S.zHi, wCell = S.wCell;
    for (int e = tid; e < nCand; e += T) {
        const unsigned ent = candPos[e];
        const unsigned col = ent & 0xff;
        if (col - (unsigned)zLo < (unsigned)(zHi - zLo)) smap[(ent >> 8) * ROWB + col] = candScore[e];
        else candScore[e] = 0;
    }
    __syncthreads();

    // ---- cell-local 3x3 strict NMS over the candidate queue, both thresholds at once (as k_fast_strips)
    unsigned keep0 = 0, keep1 = 0;                                 // one bit per queue step (candCap <= 4096: <= 16 steps)
    unsigned has = 0;
    {
        int it = 0;
        for (int base = 0; base < nCand; base += T, it++) {
            const int e = base + tid;
            if (e < nCand) {
                const unsigned ent = candPos[e];
                const int row = ent >> 8, col = ent & 0xff;
                const int c = (int)(((unsigned)(col - zLo) * S.invW) >> 16);
                const int cs = zLo + c * wCell, ce = min(cs + wCell, zHi);
                const int Sv = candScore[e];
                const bool ok = fast_nms_ok(smap + row * ROWB + col, ROWB, col == cs, col == ce - 1, Sv);
                const unsigned k0 = ok && Sv > iniTh, k1 = ok && Sv > minTh;
                keep0 |= k0 << it;
                keep1 |= k1 << it;
                has |= k0 << c;
            }
        }
    }
    has = orb_wave_or(has);
    if (lane == 0 && has) atomicOr(&shared[1], (int)has);
    __syncthreads();
    const unsigned fb = ~(unsigned)shared[1];                      // cells without a keypoint at iniTh
    unsigned keepF = 0;
    int mine = 0;
    for (unsigned mm = keep0 | keep1; mm;) {
        const int it = __ffs((int)mm) - 1;
        mm &= mm - 1;
        const int col = candPos[it * T + tid] & 0xff;
        const unsigned c = ((unsigned)(col - zLo) * S.invW) >> 16;
        const unsigned k = ((((fb >> c) & 1u) ? keep1 : keep0) >> it) & 1u;
        keepF |= k << it;
        mine += (int)k;
    }
    const int incl = orb_wave_scan_incl(mine);
    // ONE atomic per workgroup on the level's counter: the ~300 strips of a frame finish within a microsecond of each other
    // and their returning atomics on the 8 counters of a frame (one cache line) are served one after the other (~4.4 ns
    // each: 5.4 us of a 14 us launch with one atomic per wave)
    if (lane == WAVE - 1) shared[2 + wv] = incl;
    __syncthreads();
    const int t0 = shared[2], t1 = shared[3], t2 = shared[4], t3 = shared[5];
    const int total = t0 + t1 + t2 + t3;
    if (total == 0) return;
    if (tid == 0) shared[6] = atomicAdd(&candCount[f * ORB_MAX_LEVELS + S.level], total);
    __syncthreads();
    if (shared[6] + total > L.candCap) {                           // cannot happen: candCap is the NMS bound
        if (tid == 0) orb_flag_error(errFlags, f, 1);
        return;
    }
    const int base0 = shared[6] + (wv > 0 ? t0 : 0) + (wv > 1 ? t1 : 0) + (wv > 2 ? t2 : 0);
    const uint32_t* xtab = pathTab + L.pathXOff;
    const uint32_t* ytab = pathTab + L.pathYOff;
    unsigned long long* out = cand + (size_t)f * candSlab + L.candBase;
    const int cy0 = S.ci * L.hCell;
    int w = base0 + incl - mine;
    while (keepF) {
        const int it = __ffs((int)keepF) - 1;
        keepF &= keepF - 1;
        const unsigned ent = candPos[it * T + tid];
        const int row = ent >> 8, col = ent & 0xff;
        const int c = (int)(((unsigned)(col - zLo) * S.invW) >> 16);
        const int Sv = candScore[it * T + tid];
        out[w] = FAST_KEY(row, col, c, Sv);
        w++;
    }
}

template <int P>
__global__ __launch_bounds__(WAVE * MW_WAVES) void k_fast_strips_mw(const OrbGeom G, const uint8_t* __restrict__ pyr, size_t pyrSlab,
                                                                   const OrbStrip* __restrict__ strips,
                                                                   const uint32_t* __restrict__ pathTab,
                                                                   unsigned long long* __restrict__ cand, size_t candSlab,
                                                                   int* __restrict__ candCount, int* __restrict__ errFlags,
                                                                   int* __restrict__ ovfCount, int* __restrict__ ovfList, int iniTh,
                                                                   int minTh, int rowsMax, int candCap)
{
    extern __shared__ uint32_t fsm[];
    const int f = blockIdx.y, si = blockIdx.x;
    OrbStrip S;                                                    // three scalar loads (see k_fast_strips_p)
    {
        const uint4* rec = reinterpret_cast<const uint4*>(strips + si);
        const uint4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        uint4 tmp[3] = {r0, r1, r2};
        __builtin_memcpy(&S, tmp, sizeof(S));
    }
    fast_mw_strip<P>(fsm, G, pyr, pyrSlab, S, f, si, pathTab, cand, candSlab, candCount, errFlags, ovfCount, ovfList, iniTh, minTh,
                     rowsMax, candCap);
}

// The strips of ovfList ((frame << 16 | strip) entries) again, with a full score map next to the tile and
// a dense scan of it, one cell at a time: any number of candidates.  One wave per workgroup, grid-stride over the list.
__global__ __launch_bounds__(WAVE) void k_fast_strips_dense(const OrbGeom G, const uint8_t* __restrict__ pyr,
                                                            size_t pyrSlab, const OrbStrip* __restrict__ strips,
                                                            const uint32_t* __restrict__ pathTab,
                                                            unsigned long long* __restrict__ cand, size_t candSlab,
                                                            int* __restrict__ candCount, int* __restrict__ errFlags,
                                                            const int* __restrict__ ovfCount, const int* __restrict__ ovfList, int iniTh,
                                                            int minTh, int pdw, int rowsMax, int sdw)
{
    // dynamic LDS: [pad | tile | pad | score map | pair rings].  The score map only covers the zone and its 1-px
    // halo: tile rows [2, 4 + zh) and the quads [hLo, hLo + nh), at its own pitch of sdw dwords.
    extern __shared__ uint32_t fsm[];
    const int tileDwords = rowsMax * pdw;
    uint32_t* tileDw = fsm + FT_PAD;
    uint32_t* smapDw = tileDw + tileDwords + FT_PAD;
    uint16_t* pairQ = reinterpret_cast<uint16_t*>(smapDw + (rowsMax - 4) * sdw);   // [2][FT_QRING]
    const int FT_PDW = pdw, SPITCH = 4 * sdw;
    const int lane = threadIdx.x;
    const int nList = *ovfCount;
    const int lowTh = min(iniTh, minTh);
    for (int li = blockIdx.x; li < nList; li += gridDim.x) {
        const unsigned ent = (unsigned)ovfList[li];
        const int f = (int)(ent >> 16), si = (int)(ent & 0xffffu);
        const OrbStrip S = strips[si];
        const OrbLevelGeom& L = G.L[S.level];
        __syncthreads();                                           // previous strip's LDS reads are done
        fast_stage(S, pyr + (size_t)f * pyrSlab + L.pyrOff, L.pitch, tileDw, FT_PDW, lane);
        const int zh = S.zh, zLo = S.zLo, zHi = S.zHi;
        // Quads [qLo, qHi) cover the zone columns [zLo, zHi); the halo columns zLo-1 and zHi (and the rows above and
        // below the zone) only ever read as score 0 by the NMS -- cv::FAST scores nothing outside the ROI interior.
        const int qLo = S.qLo, nq = S.nq, qHi = qLo + nq;
        const int hLo = S.hLo, nh = S.nh, hHi = hLo + nh;          // halo-inclusive quad range
        uint8_t* smapZ = reinterpret_cast<uint8_t*>(smapDw) - 2 * SPITCH - 4 * hLo;    // score of tile (row, byte col)
        uint32_t* smapQ = smapDw - 2 * sdw - hLo;                  // the same, addressed by (row, quad)
        for (int i = lane; i < 2 * nh; i += WAVE) {
            const int row = (i < nh) ? 2 : 3 + zh, q = hLo + (i < nh ? i : i - nh);
            smapQ[row * sdw + q] = 0;
        }
        for (int i = lane; i < 2 * zh; i += WAVE) {                // left / right halo quads of the zone rows
            const int row = 3 + (i >> 1), q = (i & 1) ? hHi - 1 : hLo;
            if (q < qLo || q >= qHi) smapQ[row * sdw + q] = 0;
        }
        __syncthreads();
        FastSink K;
        K.smapZ = smapZ; K.spitch = SPITCH;
        K.candPos = nullptr; K.candScore = nullptr; K.candCap = 0; K.nCand = 0;
        fast_detect<true>(S, tileDw, FT_PDW, pairQ, smapQ, sdw, lowTh, lane, K);

        const uint8_t* smap = smapZ;
        const int nc = S.nc, wCell = S.wCell;
        const uint32_t* xtab = pathTab + L.pathXOff;
        const uint32_t* ytab = pathTab + L.pathYOff;
        unsigned long long* out = cand + (size_t)f * candSlab + L.candBase;
        int* cnt = &candCount[f * ORB_MAX_LEVELS + S.level];
        const int cy0 = S.ci * L.hCell;
        // The reference re-runs cv::FAST with minThFAST when the iniThFAST call returns NO KEYPOINT (:857-861), i.e. after NMS
        for (int c = 0; c < nc; c++) {
            const int cs = zLo + c * wCell, ce = min(cs + wCell, zHi), wz = ce - cs, npx = wz * zh;
            int t0 = 0, t1 = 0;
            for (int base = 0; base < npx; base += WAVE) {
                const int i = base + lane;
                bool k0 = false, k1 = false;
                if (i < npx) {
                    const int r = i / wz, col = cs + i - r * wz;
                    const uint8_t* s = smap + (3 + r) * SPITCH + col;
                    const int Sv = s[0];
                    const bool ok = fast_nms_ok(s, SPITCH, col == cs, col == ce - 1, Sv);
                    k0 = ok && Sv > iniTh;
                    k1 = ok && Sv > minTh;
                }
                t0 += __popcll(__ballot(k0));
                t1 += __popcll(__ballot(k1));
            }
            const int th = t0 ? iniTh : minTh, tot = t0 ? t0 : t1;
            if (tot == 0) continue;
            int base0 = 0;
            if (lane == 0) base0 = atomicAdd(cnt, tot);
            base0 = __builtin_amdgcn_readfirstlane(base0);
            if (base0 + tot > L.candCap) {                         // cannot happen: candCap is the NMS bound
                if (lane == 0) orb_flag_error(errFlags, f, 1);
                break;
            }
            int run = base0;
            for (int base = 0; base < npx; base += WAVE) {
                const int i = base + lane;
                bool k = false;
                int row = 0, col = 0, Sv = 0;
                if (i < npx) {
                    const int r = i / wz;
                    row = 3 + r;
                    col = cs + i - r * wz;
                    const uint8_t* s = smap + row * SPITCH + col;
                    Sv = s[0];
                    k = fast_nms_ok(s, SPITCH, col == cs, col == ce - 1, Sv) && Sv > th;
                }
                const unsigned long long b = __ballot(k);
                if (k) out[run + mbcnt64(b)] = FAST_KEY(row, col, c, Sv);
                run += __popcll(b);
            }
        }
    }
}
#undef FAST_KEY

size_t orb_fast_lds_bytes(int pdw, int rowsMax, int candCap)
{
    return (size_t)4 * (FT_PAD + rowsMax * pdw + FT_PAD) + (size_t)2 * 2 * FT_QRING + (size_t)3 * candCap;
}
size_t orb_fast_dense_lds_bytes(int pdw, int rowsMax, int sdw)
{
    return (size_t)4 * (FT_PAD + rowsMax * pdw + FT_PAD + (rowsMax - 4) * sdw) + (size_t)2 * 2 * FT_QRING;
}

// ovfList: nStrips * nFrames ints; *ovfCount zeroed by the caller before the launch
void orb_launch_fast_strips(hipStream_t st, const OrbGeom& G, const uint8_t* pyr, size_t pyrSlab,
                            const OrbStrip* strips, int nStrips, const uint32_t* pathTab, unsigned long long* cand,
                            size_t candSlab, int* candCount, int* errFlags, int* ovfCount, int* ovfList, int iniTh, int minTh,
                            int pdw, int rowsMax, int sdw, int candCap, int nFrames, int fixedPitch, bool skipDense)
{
    if (nStrips == 0) return;
    unsigned inv = 0;
    const unsigned wgs = orb_xcd_grid((unsigned)nStrips, nFrames, &inv);
    const dim3 grid = wgs ? dim3(wgs) : dim3(nStrips, nFrames);
    if (fixedPitch && orb_fast_p_lds_bytes(fixedPitch, rowsMax, candCap) > 64 * 1024) fixedPitch = 0;
    // a few frames: four waves per strip (k_fast_strips_mw) -- the launch lasts as long as one strip's chain.  ORB_FAST_MW=0|1 pins.
    const char* mwEnv = std::getenv("ORB_FAST_MW");
    const int mwPin = mwEnv ? std::atoi(mwEnv) : -1;
    const bool mw = (fixedPitch == 28 || fixedPitch == 20) && candCap <= 4096 && orb_fast_mw_lds_bytes(fixedPitch, rowsMax, candCap) <= 64 * 1024 &&
                    (mwPin >= 0 ? mwPin != 0 : (long long)nStrips * nFrames <= 1536);
    // waves per SIMD of k_fast_strips_p<28> (ORB_FAST_OCC=0|4|5|6): 5 by default.  Its single-wave workgroups are limited by LDS
    // (7 KB each: 22 per CU), and the dispatcher does not spread 22 waves evenly over the CU's four SIMDs; claiming 96 VGPRs
    // caps every SIMD at 5 -- 20 per CU, evenly -- and the launch is 3 % SHORTER with fewer waves (0.466 against 0.480 ms per 512
    // frames, natural content 0.540 against 0.555; 20 waves per CU reached through LDS padding instead: 0.506)
    static const int occ = [] { const char* e = std::getenv("ORB_FAST_OCC"); return e ? std::atoi(e) : 5; }();
    if (mw) {
        const size_t lds = orb_fast_mw_lds_bytes(fixedPitch, rowsMax, candCap);
#define ORB_FAST_MW_LAUNCH(PP)                                                                                                  \
        hipLaunchKernelGGL((k_fast_strips_mw<PP>), dim3(nStrips, nFrames), dim3(WAVE * MW_WAVES), lds, st, G, pyr, pyrSlab, strips, pathTab, \
                           cand, candSlab, candCount, errFlags, ovfCount, ovfList, iniTh, minTh, rowsMax, candCap)
        if (fixedPitch == 28) ORB_FAST_MW_LAUNCH(28);
        else ORB_FAST_MW_LAUNCH(20);
#undef ORB_FAST_MW_LAUNCH
    }
    else
#define ORB_FAST_P_LAUNCH(PP, OO)                                                                                               \
    hipLaunchKernelGGL((k_fast_strips_p<PP, OO>), grid, dim3(WAVE), orb_fast_p_lds_bytes(PP, rowsMax, candCap), st, G, pyr, pyrSlab, strips, pathTab, \
                       cand, candSlab, candCount, errFlags, ovfCount, ovfList, iniTh, minTh, rowsMax, candCap, nStrips, nFrames, inv)
    if (fixedPitch == 28) { if (occ == 4) ORB_FAST_P_LAUNCH(28, 4); else if (occ == 5) ORB_FAST_P_LAUNCH(28, 5); else if (occ == 6) ORB_FAST_P_LAUNCH(28, 6); else ORB_FAST_P_LAUNCH(28, 0); }
    else if (fixedPitch == 20) ORB_FAST_P_LAUNCH(20, 0);
#undef ORB_FAST_P_LAUNCH
    else
    hipLaunchKernelGGL(k_fast_strips, grid, dim3(WAVE),
                       orb_fast_lds_bytes(pdw, rowsMax, candCap), st, G, pyr, pyrSlab, strips, pathTab, cand, candSlab,
                       candCount, errFlags, ovfCount, ovfList, iniTh, minTh, pdw, rowsMax, candCap, nStrips, nFrames, inv);
    if (skipDense) return;                                 // the caller looks at *ovfCount afterwards and redoes the batch if it is not 0
    const long long all = (long long)nStrips * nFrames;
    // (the list is almost always empty: a small grid keeps this launch short in a single frame's chain; the kernel
    // strides over the list whatever its length)
    hipLaunchKernelGGL(k_fast_strips_dense, dim3((unsigned)std::max<long long>(16, std::min<long long>(all / 16, 2048))), dim3(WAVE),
                       orb_fast_dense_lds_bytes(pdw, rowsMax, sdw), st, G, pyr, pyrSlab, strips, pathTab, cand, candSlab,
                       candCount, errFlags, ovfCount, ovfList, iniTh, minTh, pdw, rowsMax, sdw);
}
