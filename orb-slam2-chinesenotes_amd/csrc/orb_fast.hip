// orb_fast.hip -- per-cell FAST-9/16 detection on gfx950.
// Reference: the cell loop of ORBextractor::ComputeKeyPointsOctTree, src/ORBextractor.cc:795-875
// (cv::FAST(cell ROI, iniThFAST, nonmax=true), fallback to minThFAST when it returns nothing).
//
// One wave64 per FAST cell (<= 66x66 px ROI).  What bounds this kernel is VALU issue (rocprofv3: 95 % VALU-busy,
// ~1090 vector instructions per cell), so the arithmetic is arranged for the cheapest instruction mix CDNA4 offers:
//   * the ROI is staged in LDS with aligned dword loads; a lane works on a QUAD of 4 horizontally adjacent pixels
//     as two packed pairs;
//   * phase A is an exact cheap rejection on every pair: a 9-arc contains ring pixel k or k+8 for every k, so
//     V <= U = max(I - max_k min(r_k, r_k+8), min_k max(r_k, r_k+8) - I); with the four even k (11 dword reads)
//     about 3 pairs in 4 have U <= min(iniTh, minTh) in both pixels and are finished;
//   * phase B computes the exact V(p) = max(I_p - min_arcs max_arc ring, max_arcs min_arc ring - I_p) only for the
//     queued pairs (7 rows x 12 bytes: 21 dword reads), on dense lanes again;
//   * both phases use v_pk_maximum3_f16 / v_pk_minimum3_f16: a u8 stored in a 16-bit half is a positive f16
//     subnormal whose order is the integer order, so the packed 3-input float min/max is exact and moves 2 pixels x
//     3 operands per instruction (tools/ubench).  One v_perm_b32 builds each packed ring operand from the window;
//   * V is threshold-free: both thresholds and the NMS read the same u8 score map;
//   * only pixels with V > min(iniTh, minTh) can ever be keypoints: they are queued (the queue re-uses the image
//     tile's LDS) and NMS + emission run over the queue, not over the zone;
//   * the quadtree path of a candidate is two table look-ups (x and y bisect independently);
//   * everything derived from the cell rectangle alone comes precomputed in the 32-byte OrbCell record.
#include <algorithm>

#include "orb_kernels.h"
#include "orb_wave.h"

#define WAVE 64
#define FT_PAD 4                       // dwords of slack around the tile (edge quads read one dword outside)
// LDS is sized per image geometry (dynamic): tile and score map use a row pitch of `pdw` dwords that covers the
// widest ROI of the frame (13 dwords at 640x480 instead of the 18 a 66-px ROI would need), which roughly doubles
// the number of resident waves.

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

__global__ __launch_bounds__(WAVE) void k_fast_cells(const OrbGeom G, const uint8_t* __restrict__ pyr,
                                                     size_t pyrSlab, const OrbCell* __restrict__ cells,
                                                     const uint32_t* __restrict__ pathTab,
                                                     unsigned long long* __restrict__ cand, size_t candSlab,
                                                     int* __restrict__ candCount, int* __restrict__ errFlags,
                                                     int iniTh, int minTh, int maxItems, int pdw, int rowsMax, int tileDwords,
                                                     int nCells, int nFrames, unsigned invPerFrame)
{
    // dynamic LDS: [pad | tile (>= 2 bytes per zone pixel: it later holds the candidate queue) | pad | score map | pair queues]
    extern __shared__ uint32_t fsm[];
    uint32_t* tileDw = fsm + FT_PAD;
    uint32_t* smapDw = tileDw + tileDwords + FT_PAD;
    uint16_t* pairQ = reinterpret_cast<uint16_t*>(smapDw + rowsMax * pdw);   // [2][maxItems]
    const int FT_PDW = pdw, FT_PITCH = 4 * pdw;
    const int lane = threadIdx.x;
    int f, ci;
    if (invPerFrame) {                                             // 1-D XCD-aware grid: a frame's cells share one L2
        if (!orb_xcd_decode(blockIdx.x, (unsigned)nCells, invPerFrame, nFrames, f, ci)) return;
    } else {
        f = blockIdx.y;
        ci = blockIdx.x;
    }
    const OrbCell cell = cells[ci];
    const OrbLevelGeom& L = G.L[cell.level];
    const uint8_t* img = pyr + (size_t)f * pyrSlab + L.pyrOff;

    // ---- stage the ROI rows [y0, y0+h) as aligned dwords; lane -> (row in pass, dword column)
    const int xoff = cell.xoff, xa = cell.x0 - xoff;
    const int ndw = cell.ndw;                                      // <= 18
    {
        const int rp = (int)(((unsigned)lane * cell.invDw) >> 20), c = lane - rp * ndw;
        const int rowsPerPass = cell.rowsPerPass;
        if (rp < rowsPerPass) {
            const uint8_t* src = img + (size_t)(cell.y0 + rp) * L.pitch + xa + 4 * c;
            const size_t step = (size_t)rowsPerPass * L.pitch;
            uint32_t* dstp = tileDw + rp * FT_PDW + c;
            const int dstep = rowsPerPass * FT_PDW;
#pragma unroll 4
            for (int r = rp; r < cell.h; r += rowsPerPass, src += step, dstp += dstep)
                *dstp = *reinterpret_cast<const uint32_t*>(src);
        }
    }
    const int zh = cell.zh;                                        // detection zone: tile rows [3,3+zh), cols [zLo,zHi)
    const int zLo = cell.zLo, zHi = cell.zHi;
    // Quads [qLo, qHi) cover the zone columns [zLo, zHi); the halo columns zLo-1 and zHi (and the rows above and
    // below the zone) only ever read as score 0 by the NMS -- cv::FAST scores nothing outside the ROI interior -- so
    // they are zeroed here instead of being run through the detector (one quad per row less for 3 alignments in 4).
    const int qLo = cell.qLo, nq = cell.nq, qHi = qLo + nq;
    const int hLo = cell.hLo, nh = cell.nh, hHi = hLo + nh;        // halo-inclusive quad range
    for (int i = lane; i < 2 * nh; i += WAVE) {
        const int row = (i < nh) ? 2 : 3 + zh, q = hLo + (i < nh ? i : i - nh);
        smapDw[row * FT_PDW + q] = 0;
    }
    for (int i = lane; i < 2 * zh; i += WAVE) {                                   // left / right halo quads of the zone rows
        const int row = 3 + (i >> 1), q = (i & 1) ? hHi - 1 : hLo;
        if (q < qLo || q >= qHi) smapDw[row * FT_PDW + q] = 0;
    }
    __syncthreads();

    // ---- phase A: cheap exact rejection on every pixel pair (what cv::FAST's threshold tests amount to).
    // A 9-arc of the 16-ring contains ring pixel k or k+8 for every k, so with lo_k = min(r_k, r_k+8),
    // hi_k = max(r_k, r_k+8):   V <= U := max(I - max_k lo_k, min_k hi_k - I).
    // Only the 4 even k are used here (rows y, y+-2, y+-3: 11 dword reads instead of 21); pairs with
    // U <= lowTh in both pixels score 0 (never a corner at either threshold) and skip the exact V (about 5 in 6 pairs).
    const int lowTh = min(iniTh, minTh);
    const unsigned thK = (unsigned)(0x7fff - lowTh) * 0x10001u;
    const int nItems = nq * zh;
    const unsigned invq = cell.invQ;
    int nA = 0, nB = 0;                                            // wave-uniform queue lengths
    // item -> (zone row ry, quad qi) is advanced incrementally (64 items per step): no per-item division
    int ry = (int)(((unsigned)lane * invq) >> 20);
    int qi = lane - ry * nq;
    const int stepR = cell.stepR, stepQ = WAVE - stepR * nq;
    for (int base = 0; base < nItems; base += WAVE) {
        const int item = base + lane;
        bool pa = false, pb = false;
        const int q = qLo + qi;
        const int row = 3 + ry;
        if (item < nItems) {
            const uint32_t* p = tileDw + row * FT_PDW + q - 1;
            const unsigned c0 = p[0], c1 = p[1], c2 = p[2];                                   // row y
            const unsigned u1 = p[-3 * FT_PDW + 1], d1 = p[3 * FT_PDW + 1];                   // rows y-3, y+3: x .. x+3
            const unsigned a0 = p[-2 * FT_PDW], a1 = p[-2 * FT_PDW + 1], a2 = p[-2 * FT_PDW + 2];   // row y-2
            const unsigned b0 = p[2 * FT_PDW], b1 = p[2 * FT_PDW + 1], b2 = p[2 * FT_PDW + 2];      // row y+2
            smapDw[row * FT_PDW + q] = 0;
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
            // [-255, 255] sets bit 15 exactly then.  Pixels of a neighbouring cell inside an edge quad may queue
            // a pair needlessly; phase B zeroes their scores, so zone membership is not tested here.
            pa = (pk_add_u16(u[0], thK) & 0x80008000u) != 0;
            pb = (pk_add_u16(u[1], thK) & 0x80008000u) != 0;
        }
        const unsigned long long ba = __ballot(pa), bb = __ballot(pb);
        const unsigned long long lt = (1ull << lane) - 1;
        const uint16_t ent = (uint16_t)((row << 8) | q);          // queue entries carry (row, quad) directly
        if (pa) pairQ[nA + __popcll(ba & lt)] = ent;
        if (pb) pairQ[maxItems + nB + __popcll(bb & lt)] = ent;
        nA += __popcll(ba);
        nB += __popcll(bb);
        qi += stepQ;
        ry += stepR;
        if (qi >= nq) { qi -= nq; ry++; }
    }
    __syncthreads();

    // ---- phase B: exact V for the queued pairs (dense lanes again); remember pixels above the lower threshold
    unsigned long long cmask = 0;                                  // bit 2*step+j, steps over queue A then queue B
    int step = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int nQ = h ? nB : nA;
        const uint16_t* Q = pairQ + (h ? maxItems : 0);
        for (int base = 0; base < nQ; base += WAVE, step++) {
            const int e = base + lane;
            if (e < nQ) {
                const int ent = Q[e];
                const int row = ent >> 8, q = ent & 0xff;
                unsigned W[7][3];
#pragma unroll
                for (int r = 0; r < 7; r++) {
                    const uint32_t* p = tileDw + (row - 3 + r) * FT_PDW + q - 1;
                    W[r][0] = p[0]; W[r][1] = p[1]; W[r][2] = p[2];
                }
                const unsigned s2 = h ? fast_pair<6>(W) : fast_pair<4>(W);      // two scores, one per 16-bit half
                const int cx = 4 * q + 2 * h;
                int sLo = (int)(s2 & 0xff), sHi = (int)((s2 >> 16) & 0xff);
                if (!(cx >= zLo && cx < zHi)) sLo = 0;             // pixel of the neighbouring cell
                if (!(cx + 1 >= zLo && cx + 1 < zHi)) sHi = 0;
                reinterpret_cast<uint16_t*>(smapDw)[(row * FT_PITCH + cx) >> 1] = (uint16_t)(sLo | (sHi << 8));
                unsigned fl = 0;
                if (sLo > lowTh) fl |= 1;
                if (sHi > lowTh) fl |= 2;
                cmask |= (unsigned long long)fl << (2 * step);
            }
        }
    }
    const int stepsA = (nA + WAVE - 1) / WAVE;                     // steps [0, stepsA) belong to queue A
    __syncthreads();                                               // tile is dead from here on: it becomes the queue

    // ---- queue of candidate pixels, entry = row << 8 | col (tile coordinates); order is irrelevant
    uint16_t* queue = reinterpret_cast<uint16_t*>(tileDw);
    const uint8_t* smap = reinterpret_cast<const uint8_t*>(smapDw);
    const int mine = __popcll(cmask);
    const int incl = orb_wave_scan_incl(mine);
    const int nCand = __builtin_amdgcn_readlane(incl, WAVE - 1);
    if (nCand == 0) return;
    if (nCand > 2 * tileDwords) {                                  // cannot happen: the tile region holds 2 B per zone pixel
        if (lane == 0) atomicOr(&errFlags[f], 16);
        return;
    }
    {
        int w = incl - mine;
        unsigned long long mm = cmask;
        while (mm) {
            const int bit = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const int st = bit >> 1;
            const int h = st >= stepsA;
            const int e = (h ? st - stepsA : st) * WAVE + lane;
            const int ent = pairQ[(h ? maxItems : 0) + e];
            queue[w++] = (uint16_t)((ent & 0xff00) | (4 * (ent & 0xff) + 2 * h + (bit & 1)));
        }
    }
    __syncthreads();

    // ---- cell-local 3x3 strict NMS on score = (V > th) ? V-1 : 0.  The reference re-runs cv::FAST with
    // minThFAST when the iniThFAST call returns NO KEYPOINT (:857-861) -- i.e. after NMS, so a plateau of
    // equal scores that suppresses itself also triggers the fallback.
    unsigned long long keep = 0;
    int total = 0;
    for (int attempt = 0; attempt < 2 && total == 0; attempt++) {
        const int th = attempt ? minTh : iniTh;
        keep = 0;
        int it = 0;
        for (int base = 0; base < nCand; base += WAVE, it++) {
            const int e = base + lane;
            bool k = false;
            if (e < nCand) {
                const unsigned ent = queue[e];
                const uint8_t* s = smap + (ent >> 8) * FT_PITCH + (ent & 0xff);
                const int S = s[0];
                if (S > th) {
                    // sc = S-1 must exceed max(0, scores of the neighbours above th); a neighbour at or below th
                    // is below S anyway, so this is S > max(1, raw neighbour values)
                    const int m0 = max(max((int)s[-FT_PITCH - 1], (int)s[-FT_PITCH]), (int)s[-FT_PITCH + 1]);
                    const int m1 = max(max((int)s[-1], (int)s[1]), 1);
                    const int m2 = max(max((int)s[FT_PITCH - 1], (int)s[FT_PITCH]), (int)s[FT_PITCH + 1]);
                    k = S > max(max(m0, m1), m2);
                }
            }
            if (k) keep |= 1ull << it;
            total += __popcll(__ballot(k));
        }
    }
    if (total == 0) return;

    int base0 = 0;
    if (lane == 0) base0 = atomicAdd(&candCount[f * ORB_MAX_LEVELS + cell.level], total);
    base0 = __builtin_amdgcn_readfirstlane(base0);
    if (base0 + total > L.candCap) {                               // cannot happen: candCap is the NMS bound
        if (lane == 0) atomicOr(&errFlags[f], 1);
        return;
    }
    unsigned long long* out = cand + (size_t)f * candSlab + L.candBase + base0;
    const uint32_t* xtab = pathTab + L.pathXOff;
    const uint32_t* ytab = pathTab + L.pathYOff;
    int run = 0, it = 0;
    for (int base = 0; base < nCand; base += WAVE, it++) {
        const bool k = (keep >> it) & 1;
        const unsigned long long b = __ballot(k);
        if (k) {
            const unsigned ent = queue[base + lane];
            const int row = ent >> 8, col = ent & 0xff;
            const int xin = col - xoff, yin = row;                 // cv::FAST keypoint coords inside the ROI
            const int S = smap[row * FT_PITCH + col];
            const int cx = xin + cell.cj * L.wCell, cy = yin + cell.ci * L.hCell;   // :868-869
            unsigned long long key = (unsigned long long)(xtab[cx] | ytab[cy]) << ORB_KEY_PATH_SHIFT;
            key |= ((unsigned long long)cell.ci << 27) | ((unsigned long long)cell.cj << 20) |
                   ((unsigned long long)yin << 14) | ((unsigned long long)xin << 8) | (unsigned long long)(S - 1);
            out[run + __popcll(b & ((1ull << lane) - 1))] = key;
        }
        run += __popcll(b);
    }
}

void orb_launch_fast_cells(hipStream_t st, const OrbGeom& G, const uint8_t* pyr, size_t pyrSlab,
                           const OrbCell* cells, int nCells, const uint32_t* pathTab, unsigned long long* cand,
                           size_t candSlab, int* candCount, int* errFlags, int iniTh, int minTh, int maxItems,
                           int pdw, int rowsMax, int maxZonePx, int nFrames)
{
    if (nCells == 0) return;
    const int tileDwords = std::max(rowsMax * pdw, (maxZonePx + 1) / 2);
    const size_t lds = (size_t)4 * (FT_PAD + tileDwords + FT_PAD + rowsMax * pdw) + (size_t)4 * maxItems;
    unsigned inv = 0;
    const unsigned wgs = orb_xcd_grid((unsigned)nCells, nFrames, &inv);
    hipLaunchKernelGGL(k_fast_cells, wgs ? dim3(wgs) : dim3(nCells, nFrames), dim3(WAVE), lds, st, G, pyr, pyrSlab, cells,
                       pathTab, cand, candSlab, candCount, errFlags, iniTh, minTh, maxItems, pdw, rowsMax, tileDwords, nCells,
                       nFrames, inv);
}
