// orb_extract_kernels.hip -- the pyramid kernels of the ORB extractor on gfx950 (wave64; no MFMA: fixed-point
// bilinear resampling, bound by the memory system).
//
// Stage map (reference src/ORBextractor.cc):
//   k_copy_level0 / k_resize_level4p (fallbacks k_resize_level4, k_resize_level)   ComputePyramid :1153-1180
//                                                                                   (cv::resize INTER_LINEAR)
//   (k_fast_cells lives in orb_fast.hip)
//   (k_quadtree lives in orb_quadtree.hip)
//   (k_orient_desc lives in orb_desc.hip)
// Every kernel takes blockIdx.y (or .z) = frame: batched frames are independent.
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>

#include "orb_kernels.h"

#pragma clang fp contract(off)


#define WAVE 64

// ------------------------------------------------------------------------------------------------
// Level 0: copy of the input into the pitch-aligned pyramid slab (reference :1173 copies the image
// into its bordered buffer).  One thread per 16 output bytes.
__global__ __launch_bounds__(256) void k_copy_level0(const uint8_t* __restrict__ src, size_t rowStride,
                                                     size_t frameStride, uint8_t* __restrict__ pyr,
                                                     size_t pyrSlab, int w, int h, int pitch, int vec16, int x16n,
                                                     unsigned invx, int* __restrict__ clr, int clrInts,
                                                     const char4* __restrict__ pat8, float4* __restrict__ patF)
{
    // the first kernel of the chain also clears the batch's status words (error flags, candidate / keypoint counters,
    // overflow list head): one launch less than a memset node in front of it (5 us of a single frame's chain)
    {
        const unsigned g = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        if (g < (unsigned)clrInts) clr[g] = 0;
        // ... and spreads the 256 BRIEF pairs (int8 x0,y0,x1,y1) into the float4 table k_orient_desc reads: done per
        // batch, so whoever wrote the int8 pattern (orb_extractor_set_pattern*, an RCCL broadcast) needs no hook
        if (g < 256u) {
            const char4 q = pat8[g];
            patF[g] = make_float4((float)q.x, (float)q.z, (float)q.y, (float)q.w);   // {x0, x1, y0, y1}: the two points of a pair side by side per axis (orb_desc.hip)
        }
    }
    // flattened (row, 16-byte column) index: every lane of every wave has work (a 640-px row is only 40 columns)
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    const int y = invx ? (int)(((unsigned long long)idx * invx) >> 32) : (int)idx;     // invx == 0: one item per row
    const int x16 = (int)idx - y * x16n;
    if (y >= h) return;
    const uint8_t* s = src + (size_t)f * frameStride + (size_t)y * rowStride + (size_t)x16 * 16;
    uint8_t* d = pyr + (size_t)f * pyrSlab + (size_t)y * pitch + (size_t)x16 * 16;
    if (vec16 && x16 * 16 + 16 <= w) {
        *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
    } else {
        const int n = min(16, w - x16 * 16);
        for (int i = 0; i < n; i++) d[i] = s[i];
    }
}

// ------------------------------------------------------------------------------------------------
// cv::resize(INTER_LINEAR, 8UC1) of level l-1 into level l, fixed point exactly as OpenCV's
// HResizeLinear/VResizeLinear.  One thread -> 4 horizontally adjacent output pixels.
// xtab[dx] = {sx, a0 | a1<<16}; ytab[dy] = {sy0 | sy1<<16, b0 | b1<<16} (built on the host).
__global__ __launch_bounds__(256) void k_resize_level(uint8_t* __restrict__ pyr, size_t pyrSlab,
                                                      int srcOff, int srcPitch, int dstOff, int dstPitch,
                                                      int dw, int dh, const int2* __restrict__ xtab,
                                                      const int2* __restrict__ ytab)
{
    const int x4 = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int f = blockIdx.z;
    if (x4 * 4 >= dw || y >= dh) return;
    const uint8_t* src = pyr + (size_t)f * pyrSlab + srcOff;
    uint8_t* dst = pyr + (size_t)f * pyrSlab + dstOff;
    const int2 ty = ytab[y];
    const uint8_t* r0 = src + (size_t)(ty.x & 0xffff) * srcPitch;
    const uint8_t* r1 = src + (size_t)(ty.x >> 16) * srcPitch;
    const int b0 = (short)(ty.y & 0xffff), b1 = (short)(ty.y >> 16);
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = x4 * 4 + i;
        if (x < dw) {
            const int2 tx = xtab[x];
            const int sx = tx.x;
            const int a0 = (short)(tx.y & 0xffff), a1 = (short)(tx.y >> 16);
            const int sx1 = sx + (a1 != 0);           // a1 == 0 whenever sx+1 would be out of range
            const int h0 = r0[sx] * a0 + r0[sx1] * a1;
            const int h1 = r1[sx] * a0 + r1[sx1] * a1;
            const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            out |= (uint32_t)(v & 0xff) << (8 * i);
        }
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)y * dstPitch + (size_t)x4 * 4) = out;
}

// ------------------------------------------------------------------------------------------------
// Same arithmetic, memory-lean form used whenever 4 consecutive output pixels draw on <= 12 aligned
// source bytes per row (scale <= 2.2; ORB-SLAM2 uses 1.2): one thread loads its four {sx,a0,a1}
// entries with two 16-byte loads and each source row as three aligned dwords, then funnel-shifts the
// byte pairs out of the window (10 memory instructions per 4 pixels instead of 22).
__global__ __launch_bounds__(256) void k_resize_level4(uint8_t* __restrict__ pyr, size_t pyrSlab,
                                                       int srcOff, int srcPitch, int dstOff, int dstPitch,
                                                       int dw, int dh, const int2* __restrict__ xtab,
                                                       const int2* __restrict__ ytab, int x4n, unsigned invx)
{
    // flattened (row, pixel quad) index so that every lane has work whatever the level width
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    const int y = invx ? (int)(((unsigned long long)idx * invx) >> 32) : (int)idx;     // invx == 0: one item per row
    const int x4 = (int)idx - y * x4n;
    if (y >= dh) return;
    const uint8_t* src = pyr + (size_t)f * pyrSlab + srcOff;
    uint8_t* dst = pyr + (size_t)f * pyrSlab + dstOff;
    const int2 ty = ytab[y];
    const int b0 = (short)(ty.y & 0xffff), b1 = (short)(ty.y >> 16);
    const uint4 ta = reinterpret_cast<const uint4*>(xtab)[2 * x4], tb = reinterpret_cast<const uint4*>(xtab)[2 * x4 + 1];
    const int sx[4] = {(int)ta.x, (int)ta.z, (int)tb.x, (int)tb.z};
    const unsigned co[4] = {ta.y, ta.w, tb.y, tb.w};
    const int sxBase = sx[0] & ~3;
    const uint32_t* r0 = reinterpret_cast<const uint32_t*>(src + (size_t)(ty.x & 0xffff) * srcPitch + sxBase);
    const uint32_t* r1 = reinterpret_cast<const uint32_t*>(src + (size_t)(ty.x >> 16) * srcPitch + sxBase);
    const unsigned u0 = r0[0], u1 = r0[1], u2 = r0[2];
    const unsigned v0 = r1[0], v1 = r1[1], v2 = r1[2];
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int o = sx[i] - sxBase;                     // 0..10 by construction (scale <= 2.2); o+1 <= 11
        const bool lo = o < 4, mid = o < 8;
        const unsigned pu = __builtin_amdgcn_alignbyte(lo ? u1 : (mid ? u2 : 0u), lo ? u0 : (mid ? u1 : u2), (unsigned)(o & 3));
        const unsigned pv = __builtin_amdgcn_alignbyte(lo ? v1 : (mid ? v2 : 0u), lo ? v0 : (mid ? v1 : v2), (unsigned)(o & 3));
        const int a0 = (short)(co[i] & 0xffff), a1 = (short)(co[i] >> 16);
        const int h0 = (int)(pu & 0xff) * a0 + (int)((pu >> 8) & 0xff) * a1;
        const int h1 = (int)(pv & 0xff) * a0 + (int)((pv >> 8) & 0xff) * a1;
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        if (x4 * 4 + i < dw) out |= (uint32_t)(v & 0xff) << (8 * i);
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)y * dstPitch + (size_t)x4 * 4) = out;
}

// ------------------------------------------------------------------------------------------------
// Lean form of the same arithmetic (the window kernel above is VALU-issue bound: ~150 vector instructions per
// 4 pixels, most of them byte-window selects).  Everything that depends only on the output column is moved into
// a per-thread table built once on the host (xq, 3 x 16 B per 4 pixels):
//   {baseA, baseB, sel0, sel1} {sel2, sel3, co0, co1} {co2, co3, -, -}
// baseA/baseB = 4-aligned source byte offsets of the 8-byte windows of pixels (0,1) / (2,3); sel_i = the
// v_perm_b32 selector that pulls source bytes (o_i, o_i + 1) out of the window straight into a u16 pair;
// co_i = a0 | a1 << 16.  One v_perm_b32 + one v_dot2_u32_u16 then give the horizontal pass of a row
// (p0*a0 + p1*a1), and all products fit 24-bit multiplies.  Valid while o_i + 1 <= 7, i.e. scale <= 3.
typedef unsigned short orb_u16x2 __attribute__((ext_vector_type(2)));

// bs0 / bs1: the row coefficients b0 / b1 (0..2048) shifted left by 16, so that (x * b) >> 16 is ONE v_mul_hi_u32
__device__ __forceinline__ unsigned resize_px(uint2 wa, uint2 wb, unsigned sel, unsigned co, unsigned bs0, unsigned bs1)
{
    const unsigned pa = __builtin_amdgcn_perm(wa.y, wa.x, sel), pb = __builtin_amdgcn_perm(wb.y, wb.x, sel);
    const unsigned h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u16x2, pa), __builtin_bit_cast(orb_u16x2, co), 0u, false);
    const unsigned h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u16x2, pb), __builtin_bit_cast(orb_u16x2, co), 0u, false);
    return (__umulhi(h0 >> 4, bs0) + __umulhi(h1 >> 4, bs1) + 2) >> 2;                        // <= 255
}

// the horizontal pass alone, in the form the vertical pass consumes: (p0 a0 + p1 a1) >> 4
__device__ __forceinline__ unsigned resize_h(uint2 w, unsigned sel, unsigned co)
{
    const unsigned p = __builtin_amdgcn_perm(w.y, w.x, sel);
    return __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u16x2, p), __builtin_bit_cast(orb_u16x2, co), 0u, false) >> 4;
}

#define RESIZE_ROWS 4      // output rows per thread: the 40-byte column entry and the index decode are paid once

__global__ __launch_bounds__(256) void k_resize_level4p(uint8_t* __restrict__ pyr, size_t pyrSlab, int srcOff,
                                                        int srcPitch, int dstOff, int dstPitch, int dh,
                                                        const uint4* __restrict__ xq, const int2* __restrict__ ytab,
                                                        int x4n, unsigned invx)
{
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    const unsigned yg = invx ? (unsigned)(((unsigned long long)idx * invx) >> 32) : idx;   // invx == 0: one quad per row
    const unsigned x4 = idx - __umul24(yg, (unsigned)x4n);
    const unsigned y0 = yg * RESIZE_ROWS;
    if ((int)y0 >= dh) return;
    const uint8_t* src = pyr + (size_t)f * pyrSlab + srcOff;
    uint8_t* dst = pyr + (size_t)f * pyrSlab + dstOff;
    const uint4 q0 = xq[3 * x4], q1 = xq[3 * x4 + 1];
    const uint2 q2 = *reinterpret_cast<const uint2*>(xq + 3 * x4 + 2);
    // all loads of the 4 rows are issued before the first use (the kernel is latency-bound, not issue-bound:
    // row-by-row code left the vector ALU 31 % busy); rows past the end are clamped for the loads, skipped for the store
    int2 ty[RESIZE_ROWS];
#pragma unroll
    for (int r = 0; r < RESIZE_ROWS; r++) ty[r] = ytab[min(y0 + r, (unsigned)dh - 1u)];
    uint2 wA0[RESIZE_ROWS], wA1[RESIZE_ROWS], wB0[RESIZE_ROWS], wB1[RESIZE_ROWS];
#pragma unroll
    for (int r = 0; r < RESIZE_ROWS; r++) {
        const unsigned rowA = __umul24((unsigned)ty[r].x & 0xffffu, (unsigned)srcPitch);
        const unsigned rowB = __umul24((unsigned)ty[r].x >> 16, (unsigned)srcPitch);
        wA0[r] = *reinterpret_cast<const uint2*>(src + (rowA + q0.x));
        wA1[r] = *reinterpret_cast<const uint2*>(src + (rowA + q0.y));
        wB0[r] = *reinterpret_cast<const uint2*>(src + (rowB + q0.x));
        wB1[r] = *reinterpret_cast<const uint2*>(src + (rowB + q0.y));
    }
#pragma unroll
    for (int r = 0; r < RESIZE_ROWS; r++) {
        const unsigned y = y0 + r;
        const unsigned b0 = (unsigned)ty[r].y << 16, b1 = (unsigned)ty[r].y & 0xffff0000u;   // pre-shifted, see resize_px
        const unsigned v0 = resize_px(wA0[r], wB0[r], q0.z, q1.z, b0, b1);
        const unsigned v1 = resize_px(wA0[r], wB0[r], q0.w, q1.w, b0, b1);
        const unsigned v2 = resize_px(wA1[r], wB1[r], q1.x, q2.x, b0, b1);
        const unsigned v3 = resize_px(wA1[r], wB1[r], q1.y, q2.y, b0, b1);
        if ((int)y < dh)
            *reinterpret_cast<uint32_t*>(dst + (__umul24(y, (unsigned)dstPitch) + x4 * 4)) = v0 | (v1 << 8) | (v2 << 16) | (v3 << 24);
    }
}

// ------------------------------------------------------------------------------------------------
// Two levels per launch.  The resize kernels are bound by the memory system (their time does not change when the
// arithmetic is removed), and every level is written once and read back once; here a workgroup produces a BAND of
// PAIR_ROWS rows of level D = M+1 together with the rows of level M those need: the M rows are computed from level
// S = M-1 (global windows, as above), written to HBM (level M is an output of its own) AND kept in LDS, and the D
// band is resampled from the LDS copy.  Level M is never read back from HBM (-23 % of the pyramid's traffic over the
// pairs (1,2) (3,4) (5,6)).  Adjacent bands overlap by 1-2 rows of M, which both workgroups compute and write with
// identical bytes.  Valid for scale <= 2 (consecutive D rows then draw on adjacent or overlapping M rows, so the
// bands cover every row of M).  Measured at 640x480 x 512 frames: bands of 8 / 12 / 16 / 20 / 32 / 48 / 64 rows take
// 0.380 / 0.362 / 0.346 / 0.358 / 0.362 / 0.378 / 0.42 ms for the whole pyramid, the unfused kernels 0.375 ms.
#define PAIR_ROWS 16

__global__ __launch_bounds__(256) void k_resize_pair(uint8_t* __restrict__ pyr, size_t pyrSlab, int srcOff, int srcPitch,
                                                     int midOff, int midPitch, int midH, int dstOff, int dstPitch, int dstH,
                                                     const uint4* __restrict__ xqM, const int2* __restrict__ ytabM, int x4M,
                                                     unsigned invM, const uint4* __restrict__ xqD,
                                                     const int2* __restrict__ ytabD, int x4D, unsigned invD, int ldsPitchDw)
{
    extern __shared__ uint32_t band[];                 // [rows of M][ldsPitchDw]
    const int f = blockIdx.y;
    const int R0 = blockIdx.x * PAIR_ROWS, R1 = min(R0 + PAIR_ROWS, dstH);
    const uint8_t* src = pyr + (size_t)f * pyrSlab + srcOff;
    uint8_t* mid = pyr + (size_t)f * pyrSlab + midOff;
    uint8_t* dst = pyr + (size_t)f * pyrSlab + dstOff;
    const int m0 = ytabD[R0].x & 0xffff, m1 = min((int)((unsigned)ytabD[R1 - 1].x >> 16), midH - 1);
    const int nM = m1 - m0 + 1, gM = (nM + RESIZE_ROWS - 1) / RESIZE_ROWS;

    // ---- rows m0..m1 of level M from level S: one item = 4 pixels x 4 rows, all loads before the first use
    for (unsigned idx = threadIdx.x; idx < (unsigned)(gM * x4M); idx += blockDim.x) {
        const unsigned g = __umulhi(idx, invM), x4 = idx - g * (unsigned)x4M;
        const uint4 q0 = xqM[3 * x4], q1 = xqM[3 * x4 + 1];
        const uint2 q2 = *reinterpret_cast<const uint2*>(xqM + 3 * x4 + 2);
        int2 ty[RESIZE_ROWS];
#pragma unroll
        for (int r = 0; r < RESIZE_ROWS; r++) ty[r] = ytabM[min(m0 + (int)g * RESIZE_ROWS + r, m1)];
        uint2 wA0[RESIZE_ROWS], wA1[RESIZE_ROWS], wB0[RESIZE_ROWS], wB1[RESIZE_ROWS];
#pragma unroll
        for (int r = 0; r < RESIZE_ROWS; r++) {
            const unsigned rowA = __umul24((unsigned)ty[r].x & 0xffffu, (unsigned)srcPitch);
            const unsigned rowB = __umul24((unsigned)ty[r].x >> 16, (unsigned)srcPitch);
            wA0[r] = *reinterpret_cast<const uint2*>(src + (rowA + q0.x));
            wA1[r] = *reinterpret_cast<const uint2*>(src + (rowA + q0.y));
            wB0[r] = *reinterpret_cast<const uint2*>(src + (rowB + q0.x));
            wB1[r] = *reinterpret_cast<const uint2*>(src + (rowB + q0.y));
        }
#pragma unroll
        for (int r = 0; r < RESIZE_ROWS; r++) {
            const unsigned lr = g * RESIZE_ROWS + r;
            const unsigned b0 = (unsigned)ty[r].y << 16, b1 = (unsigned)ty[r].y & 0xffff0000u;   // pre-shifted, see resize_px
            const unsigned out = resize_px(wA0[r], wB0[r], q0.z, q1.z, b0, b1) | (resize_px(wA0[r], wB0[r], q0.w, q1.w, b0, b1) << 8) |
                                 (resize_px(wA1[r], wB1[r], q1.x, q2.x, b0, b1) << 16) | (resize_px(wA1[r], wB1[r], q1.y, q2.y, b0, b1) << 24);
            if ((int)lr < nM) {
                band[lr * (unsigned)ldsPitchDw + x4] = out;
                *reinterpret_cast<uint32_t*>(mid + (__umul24((unsigned)m0 + lr, (unsigned)midPitch) + x4 * 4)) = out;
            }
        }
    }
    __syncthreads();

    // ---- rows R0..R1-1 of level D from the LDS band, again 4 pixels x 4 rows per item
    const int nD = R1 - R0, gD = (nD + RESIZE_ROWS - 1) / RESIZE_ROWS;
    for (unsigned idx = threadIdx.x; idx < (unsigned)(gD * x4D); idx += blockDim.x) {
        const unsigned g = __umulhi(idx, invD), x4 = idx - g * (unsigned)x4D;
        const uint4 q0 = xqD[3 * x4], q1 = xqD[3 * x4 + 1];
        const uint2 q2 = *reinterpret_cast<const uint2*>(xqD + 3 * x4 + 2);
        const unsigned ia = q0.x >> 2, ib = q0.y >> 2;
#pragma unroll
        for (int r = 0; r < RESIZE_ROWS; r++) {
            const int lr = (int)g * RESIZE_ROWS + r;
            const int2 ty = ytabD[min(R0 + lr, R1 - 1)];
            const unsigned b0 = (unsigned)ty.y << 16, b1 = (unsigned)ty.y & 0xffff0000u;         // pre-shifted, see resize_px
            const uint32_t* rowA = band + (unsigned)(((int)((unsigned)ty.x & 0xffffu) - m0) * ldsPitchDw);
            const uint32_t* rowB = band + (unsigned)((min((int)((unsigned)ty.x >> 16), m1) - m0) * ldsPitchDw);
            const uint2 wA0 = make_uint2(rowA[ia], rowA[ia + 1]), wA1 = make_uint2(rowA[ib], rowA[ib + 1]);
            const uint2 wB0 = make_uint2(rowB[ia], rowB[ia + 1]), wB1 = make_uint2(rowB[ib], rowB[ib + 1]);
            const unsigned out = resize_px(wA0, wB0, q0.z, q1.z, b0, b1) | (resize_px(wA0, wB0, q0.w, q1.w, b0, b1) << 8) |
                                 (resize_px(wA1, wB1, q1.x, q2.x, b0, b1) << 16) | (resize_px(wA1, wB1, q1.y, q2.y, b0, b1) << 24);
            if (lr < nD) *reinterpret_cast<uint32_t*>(dst + (__umul24((unsigned)(R0 + lr), (unsigned)dstPitch) + x4 * 4)) = out;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Chains of levels out of LDS (round 3; OrbPyrChain in orb_common.h).  k_resize_pair above spends its time waiting: table
// loads -> window loads that depend on them -> compute -> barrier -> ..., a chain of dependent global round trips per
// workgroup at 2.1 TB/s.  Here the only global loads a band waits for are issued at once from addresses that depend on
// nothing but the workgroup index: the source rows (coalesced 16-byte chunks), the row tables (turned into LDS byte
// offsets and pre-shifted coefficients, one uint4 per row) and the first column entries.  Every level is then resampled
// out of LDS with the same fixed-point arithmetic (resize_px), written to HBM and kept for the next step.  The first
// chain of a batch stages the caller's image (any alignment, any stride) and writes level 0 on the way, so the separate
// copy pass and its 0.6 MB of traffic per frame are gone; it also clears the status block and spreads the BRIEF pattern
// as k_copy_level0 does.
typedef unsigned int orb_u32x4 __attribute__((ext_vector_type(4)));
typedef orb_u32x4 __attribute__((aligned(1))) orb_u32x4_a1;
#define PYR_STAGE_MAX 8            // 16-byte chunks a thread has in flight while staging

// One level of a chain out of LDS: one item = 4 pixels x RG rows (RG = 4 amortises the column entry best, RG = 1 / 2 leave
// fewer threads idle in the last pass over a band: the host picks per chain).  hook() runs once, behind the request for the
// first item's column entry (k_pyr_chain_p puts the NEXT band's source loads there: younger than that request, so the wait
// for it does not wait for them).
template <int RG, bool XL, class Hook>
__device__ __forceinline__ void pyr_chain_step(const OrbPyrChain& C, int k, const int2* bt, uint8_t* slab, uint8_t* lds,
                                               const uint4* __restrict__ xqAll, int tid, Hook&& hook)
{
    const OrbPyrStep& T = C.st[k];
    const int2 mr = bt[1 + k];
    const int nM = mr.y - mr.x + 1, gM = (nM + RG - 1) / RG;
    const uint4* xq = xqAll + T.xqOff;
    uint8_t* dst = slab + T.dstOff + (size_t)mr.x * T.dstPitch;
    const bool keep = k + 1 < C.nSteps;
    const uint8_t* rp = lds + T.rpOff;
    uint8_t* keepL = lds + T.ldsOff;
    const unsigned x4n = (unsigned)T.x4, inv = T.invX4, dpitch = (unsigned)T.dstPitch, kpitch = 4u * (unsigned)T.ldsPitchDw;
    // the column entry of the NEXT item is requested before the current one is computed (an L2 round trip per item
    // otherwise: the entry's selectors are needed by the item's first instructions)
    const unsigned nItems = (unsigned)gM * x4n;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    uint2 q2 = make_uint2(0, 0);
    unsigned g = 0, x4 = 0;
    const uint8_t* xl = lds + C.xqLdsOff + 16 * (T.xqOff - C.st[0].xqOff);
    if (!XL && (unsigned)tid < nItems) {
        g = inv ? __umulhi((unsigned)tid, inv) : (unsigned)tid;
        x4 = (unsigned)tid - g * x4n;
        q0 = xq[3 * x4]; q1 = xq[3 * x4 + 1];
        q2 = *reinterpret_cast<const uint2*>(xq + 3 * x4 + 2);
    }
    hook();
    for (unsigned idx = tid; idx < nItems; idx += 256) {
        const unsigned idxN = idx + 256;
        uint4 n0 = q0, n1 = q1;
        uint2 n2 = q2;
        unsigned gN = g, x4N = x4;
        if constexpr (XL) {
            g = inv ? __umulhi(idx, inv) : idx;
            x4 = idx - g * x4n;
            q0 = *reinterpret_cast<const uint4*>(xl + 48 * x4);
            q1 = *reinterpret_cast<const uint4*>(xl + 48 * x4 + 16);
            q2 = *reinterpret_cast<const uint2*>(xl + 48 * x4 + 32);
        } else if (idxN < nItems) {
            gN = inv ? __umulhi(idxN, inv) : idxN;
            x4N = idxN - gN * x4n;
            n0 = xq[3 * x4N]; n1 = xq[3 * x4N + 1];
            n2 = *reinterpret_cast<const uint2*>(xq + 3 * x4N + 2);
        }
        uint4 e[RG];
#pragma unroll
        for (int r = 0; r < RG; r++) e[r] = *reinterpret_cast<const uint4*>(rp + 16 * (g * RG + r));
        // horizontal pass of a source row for the item's 4 pixels: (S[sx] a0 + S[sx + 1] a1) >> 4, the form the vertical
        // pass consumes.  At scale 1.2 the lower source row of an output row is the upper source row of the next output row
        // four times in five: its horizontal pass (2 LDS reads, 4 x (v_perm, v_dot2, shift)) is then reused, not redone.
        auto hrow = [&](unsigned rowOff, unsigned (&h)[4]) {
            const uint32_t* w0 = reinterpret_cast<const uint32_t*>(lds + rowOff + q0.x);
            const uint32_t* w1 = reinterpret_cast<const uint32_t*>(lds + rowOff + q0.y);
            const uint2 wa = make_uint2(w0[0], w0[1]), wb = make_uint2(w1[0], w1[1]);
            h[0] = resize_h(wa, q0.z, q1.z); h[1] = resize_h(wa, q0.w, q1.w);
            h[2] = resize_h(wb, q1.x, q2.x); h[3] = resize_h(wb, q1.y, q2.y);
        };
        // two register sets that the rows alternate between: row r's upper source row lives in hh[r & 1], its lower one in
        // hh[(r + 1) & 1] -- which IS the upper set of row r + 1, so a reused horizontal pass stays where it is (it used to be
        // copied, four register moves per reuse, three reuses in four rows)
        unsigned hh[2][4];
#pragma unroll
        for (int r = 0; r < RG; r++) {
            const unsigned lr = g * RG + r;
            unsigned (&hA)[4] = hh[r & 1];
            unsigned (&hB)[4] = hh[(r + 1) & 1];
            if (r == 0 || e[r].x != e[r - 1].y) hrow(e[r].x, hA);
            hrow(e[r].y, hB);
            unsigned out = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) out |= ((__umulhi(hA[j], e[r].z) + __umulhi(hB[j], e[r].w) + 2) >> 2) << (8 * j);
            if (lr < (unsigned)nM) {
                *reinterpret_cast<uint32_t*>(dst + (lr * dpitch + x4 * 4)) = out;
                if (keep) *reinterpret_cast<uint32_t*>(keepL + (lr * kpitch + x4 * 4)) = out;
            }
        }
        if constexpr (!XL) { q0 = n0; q1 = n1; q2 = n2; g = gN; x4 = x4N; }
    }
}

template <int RG, bool XL>
__global__ __launch_bounds__(256) void k_pyr_chain(const OrbPyrChain C, const uint8_t* __restrict__ img, size_t rowStride,
                                                   size_t frameStride, uint8_t* __restrict__ pyr, size_t pyrSlab,
                                                   const uint4* __restrict__ xqAll, const int2* __restrict__ ytAll,
                                                   const int2* __restrict__ bandTab, int* __restrict__ clr, int clrInts,
                                                   const char4* __restrict__ pat8, float4* __restrict__ patF,
                                                   unsigned long long* __restrict__ stamps)
{
    extern __shared__ uint32_t ldsDw[];
    uint8_t* lds = reinterpret_cast<uint8_t*>(ldsDw);
    const int tid = threadIdx.x, f = blockIdx.y, b = blockIdx.x;
    // diagnostics (orb_extractor_set_pyr_stamps): thread 0 leaves the 100 MHz clock at the workgroup's phase boundaries:
    // 0 start, 1 staging loads requested and stored, 2 staged (barrier passed), 3 + k level k of the chain done (barrier passed)
#define PYR_STAMP(k) do { if (stamps && tid == 0) stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    PYR_STAMP(0);
    if (clr) {                                                     // first kernel of the batch (see k_copy_level0)
        const unsigned g = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        if (g < (unsigned)clrInts) clr[g] = 0;
        if (g < 256u) {
            const char4 q = pat8[g];
            patF[g] = make_float4((float)q.x, (float)q.z, (float)q.y, (float)q.w);   // {x0, x1, y0, y1}: the two points of a pair side by side per axis (orb_desc.hip)
        }
    }
    // XL: the column tables of the chain's levels go to LDS (requested first: their addresses depend on nothing)
    orb_u32x4 xv[4];
    if constexpr (XL) {
        const orb_u32x4* xg = reinterpret_cast<const orb_u32x4*>(xqAll + C.st[0].xqOff);
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (tid + 256 * i < C.xqLdsN) xv[i] = xg[tid + 256 * i];
    }
    const int2* bt = bandTab + C.tabOff + b * (C.nSteps + 2);
    const int2 sr = bt[0];
    uint8_t* slab = pyr + (size_t)f * pyrSlab;

    // ---- row parameters: wave k fills those of step k (LDS byte offsets of the two source rows, coefficients << 16).  The band
    // table entries are wave-uniform (scalar loads: they do not count against the vector-memory counter), the row-table entry is
    // REQUESTED here and used behind the staging loads below -- a wait for it in front of them was two more round trips in a row
    // at the head of every workgroup.
    const int rpK = __builtin_amdgcn_readfirstlane(tid >> 6), rpT = tid & 63;
    int2 rpTy = make_int2(0, 0);
    int rpSrcRow0 = 0;
    bool rpAct = false;
    if (rpK < C.nSteps) {
        const int2 mr = bt[1 + rpK];
        const int n4 = (mr.y - mr.x + 4) & ~3;
        rpSrcRow0 = bt[rpK].x;                                     // the step's source band: bt[0] or the band of step k - 1
        rpAct = rpT < n4;
        rpTy = ytAll[C.st[rpK].ytOff + min(mr.x + min(rpT, n4 - 1), mr.y)];
    }
    // ---- stage the source rows sr.x .. sr.y (and write the level-0 rows this band owns)
    {
        const int w = C.srcW, cpr = C.cpr, total = (sr.y - sr.x + 1) * cpr;
        const uint8_t* src;
        size_t stride;
        if (C.copy0) { src = img + (size_t)f * frameStride + (size_t)sr.x * rowStride; stride = rowStride; }
        else { src = slab + C.srcOff + (size_t)sr.x * C.srcPitch; stride = (size_t)C.srcPitch; }
        const int2 cp = bt[C.nSteps + 1];
        const int pitchB = 4 * C.srcLdsPitchDw;
        uint8_t* dstL = lds + C.srcLdsOff;
        uint8_t* l0 = slab + C.srcOff + (size_t)sr.x * C.srcPitch;
        // Every load of a thread must be IN FLIGHT before the first is waited for.  (Round 5: the stamps of
        // orb_extractor_set_pyr_stamps showed a workgroup spending 7-11 of its 17-19 us here -- the last chunk of a row whose
        // width is not a multiple of 16 used to be read byte by byte in an else branch, and with loads on both sides of that
        // branch the compiler put an s_waitcnt vmcnt(0) behind EVERY 16-byte load: eight round trips in a row, not one.)
        // Now the loop holds one load per chunk and nothing else: a partial last chunk is read as the LAST 16 bytes of its row
        // (inside the caller's buffer whatever the stride) and shifted into place in registers afterwards.  The rows of a
        // pyramid level are padded to their pitch, so chains that read the slab never have a partial chunk.
        const bool canPart = C.copy0 && (w & 15) != 0 && w >= 16;
        for (int base = 0; base < total; base += 256 * PYR_STAGE_MAX) {
            orb_u32x4 v[PYR_STAGE_MAX];
            if (C.copy0 && w < 16) {                               // (rows narrower than one chunk: byte by byte, no chain is built for them in practice)
#pragma unroll
                for (int i = 0; i < PYR_STAGE_MAX; i++) {
                    const int idx = base + i * 256 + tid;
                    if (idx < total) {
                        const int r = C.invCpr ? (int)__umulhi((unsigned)idx, C.invCpr) : idx, c = idx - r * cpr;
                        const uint8_t* s = src + (size_t)r * stride + 16 * c;
                        const int n = w - 16 * c;
                        unsigned t[4] = {0, 0, 0, 0};
                        for (int j = 0; j < 15; j++)
                            if (j < n) t[j >> 2] |= (unsigned)s[j] << (8 * (j & 3));
                        v[i] = orb_u32x4{t[0], t[1], t[2], t[3]};
                    }
                }
            } else {
                // UNCONDITIONAL loads (a thread beyond the band's last chunk re-reads that chunk): an `if (idx < total) v[i] = ...`
                // makes every v[i] a phi of the whole register array, and the compiler then waits for each load where it merges
#pragma unroll
                for (int i = 0; i < PYR_STAGE_MAX; i++) {
                    const int idx = min(base + i * 256 + tid, total - 1);
                    const int r = C.invCpr ? (int)__umulhi((unsigned)idx, C.invCpr) : idx, c = idx - r * cpr;
                    const bool part = canPart && 16 * c + 16 > w;
                    v[i] = *reinterpret_cast<const orb_u32x4_a1*>(src + (size_t)r * stride + (part ? w - 16 : 16 * c));
                }
                if (canPart) {
#pragma unroll
                    for (int i = 0; i < PYR_STAGE_MAX; i++) {
                        const int idx = min(base + i * 256 + tid, total - 1);
                        const int r = C.invCpr ? (int)__umulhi((unsigned)idx, C.invCpr) : idx, c = idx - r * cpr;
                        // bytes [w - 16, w) of the row are in v; the chunk wants bytes [16 c, w) in front and zeros behind:
                        // a 128-bit shift right by k = 16 - (w - 16 c) bytes (k = 0 .. 15; whole chunks: k <= 0, left as they are)
                        const int k = 16 - (w - 16 * c), dw = k >> 2;
                        const unsigned sh = (unsigned)k & 3u;
                        const unsigned t0 = v[i].x, t1 = v[i].y, t2 = v[i].z, t3 = v[i].w;
                        const unsigned a0 = dw == 0 ? t0 : dw == 1 ? t1 : dw == 2 ? t2 : t3;
                        const unsigned a1 = dw == 0 ? t1 : dw == 1 ? t2 : dw == 2 ? t3 : 0u;
                        const unsigned a2 = dw == 0 ? t2 : dw == 1 ? t3 : 0u;
                        const unsigned a3 = dw == 0 ? t3 : 0u;
                        const bool part = k > 0;
                        v[i].x = part ? __builtin_amdgcn_alignbyte(a1, a0, sh) : t0;
                        v[i].y = part ? __builtin_amdgcn_alignbyte(a2, a1, sh) : t1;
                        v[i].z = part ? __builtin_amdgcn_alignbyte(a3, a2, sh) : t2;
                        v[i].w = part ? __builtin_amdgcn_alignbyte(0u, a3, sh) : t3;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < PYR_STAGE_MAX; i++) {
                const int idx = base + i * 256 + tid;
                if (idx < total) {
                    const int r = C.invCpr ? (int)__umulhi((unsigned)idx, C.invCpr) : idx, c = idx - r * cpr;
                    *reinterpret_cast<orb_u32x4*>(dstL + r * pitchB + 16 * c) = v[i];
                    if (C.copy0 && (unsigned)(sr.x + r - cp.x) < (unsigned)(cp.y - cp.x))
                        *reinterpret_cast<orb_u32x4*>(l0 + (size_t)r * C.srcPitch + 16 * c) = v[i];   // (row padding up to the pitch gets zeros)
                }
            }
        }
    }
    if (rpAct) {
        const int k = rpK;
        const int pitchB = 4 * (k == 0 ? C.srcLdsPitchDw : C.st[k - 1].ldsPitchDw);
        const int base = k == 0 ? C.srcLdsOff : C.st[k - 1].ldsOff;
        uint4 e;
        e.x = (unsigned)(base + ((rpTy.x & 0xffff) - rpSrcRow0) * pitchB);
        e.y = (unsigned)(base + ((int)((unsigned)rpTy.x >> 16) - rpSrcRow0) * pitchB);
        e.z = (unsigned)rpTy.y << 16;
        e.w = (unsigned)rpTy.y & 0xffff0000u;
        *reinterpret_cast<uint4*>(lds + C.st[k].rpOff + 16 * rpT) = e;
    }
    if constexpr (XL) {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (tid + 256 * i < C.xqLdsN) *reinterpret_cast<orb_u32x4*>(lds + C.xqLdsOff + 16 * (tid + 256 * i)) = xv[i];
    }
    PYR_STAMP(1);
    __syncthreads();
    PYR_STAMP(2);

    // ---- level after level out of LDS (pyr_chain_step above)
    for (int k = 0; k < C.nSteps; k++) {
        pyr_chain_step<RG, XL>(C, k, bt, slab, lds, xqAll, tid, [] {});
        __syncthreads();
        PYR_STAMP(3 + k);
    }
#undef PYR_STAMP
}

// ------------------------------------------------------------------------------------------------
// The batch form of k_pyr_chain (round 5): PERSISTENT workgroups, one per resident slot, that walk the (frame, band) units
// u = blockIdx.x, + gridDim.x, ... and hold the NEXT unit's source chunks (and its row-table entry) in registers, requested
// at the start of the current unit's first level: the stamps of k_pyr_chain showed a workgroup waiting 3-4 of its 10-14 us
// for the one round trip of its staging loads, five workgroups per CU notwithstanding.  Here that round trip runs under the
// resampling of the unit before.  The compute phase keeps its own loads out of the way of the prefetch: column entries are
// either in LDS (XL: copied once per workgroup, not once per band) or requested one item ahead -- the first before the
// prefetch (older: waited for alone), the later ones when the prefetch has long arrived.  NPF = 16-byte chunks per thread
// (the host picks it from the chain's largest band; chains whose bands need more than 8 use k_pyr_chain).
template <int RG, bool XL, int NPF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_pyr_chain_p(const OrbPyrChain C, const uint8_t* __restrict__ img, size_t rowStride,
                                                     size_t frameStride, uint8_t* __restrict__ pyr, size_t pyrSlab,
                                                     const uint4* __restrict__ xqAll, const int2* __restrict__ ytAll,
                                                     const int2* __restrict__ bandTab, int* __restrict__ clr, int clrInts,
                                                     const char4* __restrict__ pat8, float4* __restrict__ patF,
                                                     unsigned long long* __restrict__ stamps, int nUnits, unsigned invBands)
{
    extern __shared__ uint32_t ldsDw[];
    uint8_t* lds = reinterpret_cast<uint8_t*>(ldsDw);
    const int tid = threadIdx.x;
    if (clr) {                                                     // first kernel of the batch (see k_copy_level0)
        for (unsigned g = blockIdx.x * 256u + (unsigned)tid; g < (unsigned)clrInts; g += gridDim.x * 256u) clr[g] = 0;
        const unsigned g = blockIdx.x * 256u + (unsigned)tid;
        if (g < 256u) {
            const char4 q = pat8[g];
            patF[g] = make_float4((float)q.x, (float)q.z, (float)q.y, (float)q.w);   // {x0, x1, y0, y1} (orb_desc.hip)
        }
    }
    if constexpr (XL) {                                            // the chain's column tables: once per workgroup
        const orb_u32x4* xg = reinterpret_cast<const orb_u32x4*>(xqAll + C.st[0].xqOff);
        orb_u32x4 xv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) xv[i] = xg[min(tid + 256 * i, C.xqLdsN - 1)];
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (tid + 256 * i < C.xqLdsN) *reinterpret_cast<orb_u32x4*>(lds + C.xqLdsOff + 16 * (tid + 256 * i)) = xv[i];
    }
    const int w = C.srcW, cpr = C.cpr;
    const bool canPart = C.copy0 && (w & 15) != 0;                 // (w >= 16: the host sends narrower sources to k_pyr_chain)
    const int rpK = __builtin_amdgcn_readfirstlane(tid >> 6), rpT = tid & 63;
    struct Unit { int f, b; const int2* bt; int2 sr; };
    auto unit = [&](int u) {
        Unit U;
        U.f = invBands ? (int)__umulhi((unsigned)u, invBands) : u;
        U.b = u - U.f * C.bands;
        U.bt = bandTab + C.tabOff + U.b * (C.nSteps + 2);
        U.sr = U.bt[0];
        return U;
    };
    auto chunk = [&](int idx, int& r, int& c) { r = C.invCpr ? (int)__umulhi((unsigned)idx, C.invCpr) : idx; c = idx - r * cpr; };
    // what a unit's staging needs from memory: NPF chunks and one row-table entry per thread, all requested at once
    orb_u32x4 v[NPF];
    int2 rpTy = make_int2(0, 0);
    auto request = [&](int u) {
        const Unit U = unit(u);
        if (rpK < C.nSteps) {
            const int2 mr = U.bt[1 + rpK];
            const int n4 = (mr.y - mr.x + 4) & ~3;
            rpTy = ytAll[C.st[rpK].ytOff + min(mr.x + min(rpT, n4 - 1), mr.y)];
        }
        const int total = (U.sr.y - U.sr.x + 1) * cpr;
        const uint8_t* src;
        size_t stride;
        if (C.copy0) { src = img + (size_t)U.f * frameStride + (size_t)U.sr.x * rowStride; stride = rowStride; }
        else { src = pyr + (size_t)U.f * pyrSlab + C.srcOff + (size_t)U.sr.x * C.srcPitch; stride = (size_t)C.srcPitch; }
#pragma unroll
        for (int i = 0; i < NPF; i++) {                            // unconditional, clamped (see k_pyr_chain)
            int r, c;
            chunk(min(i * 256 + tid, total - 1), r, c);
            const bool part = canPart && 16 * c + 16 > w;
            v[i] = *reinterpret_cast<const orb_u32x4_a1*>(src + (size_t)r * stride + (part ? w - 16 : 16 * c));
        }
    };
#define PYR_STAMP(k) do { if (stamps && tid == 0) stamps[(size_t)u * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    int u = blockIdx.x;
    request(u);
    for (;;) {
        PYR_STAMP(0);
        const Unit U = unit(u);
        uint8_t* slab = pyr + (size_t)U.f * pyrSlab;
        // ---- the unit's source rows, out of the registers: LDS (and the level-0 rows this band owns)
        {
            const int total = (U.sr.y - U.sr.x + 1) * cpr;
            const int2 cp = U.bt[C.nSteps + 1];
            const int pitchB = 4 * C.srcLdsPitchDw;
            uint8_t* dstL = lds + C.srcLdsOff;
            uint8_t* l0 = slab + C.srcOff + (size_t)U.sr.x * C.srcPitch;
#pragma unroll
            for (int i = 0; i < NPF; i++) {
                const int idx = i * 256 + tid;
                int r, c;
                chunk(min(idx, total - 1), r, c);
                orb_u32x4 o = v[i];
                if (canPart) {                                     // a partial last chunk was read as the row's last 16 bytes (k_pyr_chain)
                    const int k = 16 - (w - 16 * c), dw = k >> 2;
                    const unsigned sh = (unsigned)k & 3u;
                    const unsigned t0 = o.x, t1 = o.y, t2 = o.z, t3 = o.w;
                    const unsigned a0 = dw == 0 ? t0 : dw == 1 ? t1 : dw == 2 ? t2 : t3;
                    const unsigned a1 = dw == 0 ? t1 : dw == 1 ? t2 : dw == 2 ? t3 : 0u;
                    const unsigned a2 = dw == 0 ? t2 : dw == 1 ? t3 : 0u;
                    const unsigned a3 = dw == 0 ? t3 : 0u;
                    const bool part = k > 0;
                    o.x = part ? __builtin_amdgcn_alignbyte(a1, a0, sh) : t0;
                    o.y = part ? __builtin_amdgcn_alignbyte(a2, a1, sh) : t1;
                    o.z = part ? __builtin_amdgcn_alignbyte(a3, a2, sh) : t2;
                    o.w = part ? __builtin_amdgcn_alignbyte(0u, a3, sh) : t3;
                }
                if (idx < total) {
                    *reinterpret_cast<orb_u32x4*>(dstL + r * pitchB + 16 * c) = o;
                    if (C.copy0 && (unsigned)(U.sr.x + r - cp.x) < (unsigned)(cp.y - cp.x))
                        *reinterpret_cast<orb_u32x4*>(l0 + (size_t)r * C.srcPitch + 16 * c) = o;   // (row padding up to the pitch gets zeros)
                }
            }
        }
        if (rpK < C.nSteps) {                                      // row parameters: wave k fills those of step k (k_pyr_chain)
            const int k = rpK;
            const int2 mr = U.bt[1 + k];
            const int n4 = (mr.y - mr.x + 4) & ~3;
            if (rpT < n4) {
                const int srcRow0 = U.bt[k].x;
                const int pitchB = 4 * (k == 0 ? C.srcLdsPitchDw : C.st[k - 1].ldsPitchDw);
                const int base = k == 0 ? C.srcLdsOff : C.st[k - 1].ldsOff;
                uint4 e;
                e.x = (unsigned)(base + ((rpTy.x & 0xffff) - srcRow0) * pitchB);
                e.y = (unsigned)(base + ((int)((unsigned)rpTy.x >> 16) - srcRow0) * pitchB);
                e.z = (unsigned)rpTy.y << 16;
                e.w = (unsigned)rpTy.y & 0xffff0000u;
                *reinterpret_cast<uint4*>(lds + C.st[k].rpOff + 16 * rpT) = e;
            }
        }
        PYR_STAMP(1);
        __syncthreads();
        PYR_STAMP(2);
        // ---- the next unit's requests go out under this unit's first level (the last unit asks for itself again: a branch
        // around the loads would make the registers conditional, and conditional loads are waited for one by one)
        const int uNext = u + (int)gridDim.x;
        const bool more = uNext < nUnits;
        const int uReq = more ? uNext : u;
        pyr_chain_step<RG, XL>(C, 0, U.bt, slab, lds, xqAll, tid, [&] { request(uReq); });
        __syncthreads();
        PYR_STAMP(3);
        for (int k = 1; k < C.nSteps; k++) {
            pyr_chain_step<RG, XL>(C, k, U.bt, slab, lds, xqAll, tid, [] {});
            __syncthreads();
            PYR_STAMP(3 + k);
        }
        if (!more) break;
        u = uNext;
    }
#undef PYR_STAMP
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (keep <<<>>> syntax inside this translation unit)
// exact division of idx < 2^31 by d via multiply-high: q = (idx * ceil(2^32/d)) >> 32 is exact while idx*d < 2^32
// d == 1 has no 32-bit inverse: 0 tells the kernels that a row holds a single item (index == row)
static unsigned inv32(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); }

void orb_launch_copy_level0(hipStream_t st, const uint8_t* src, size_t rowStride, size_t frameStride,
                            uint8_t* pyr, size_t pyrSlab, int w, int h, int pitch, int nFrames, int* clr, int clrInts,
                            const int8_t* pat8, float* patF)
{
    const int vec16 = ((reinterpret_cast<uintptr_t>(src) | rowStride | frameStride) & 15) == 0;
    const int x16 = (w + 15) / 16;
    const int total = x16 * h;
    dim3 grid((total + 255) / 256, nFrames);
    hipLaunchKernelGGL(k_copy_level0, grid, dim3(256), 0, st, src, rowStride, frameStride, pyr, pyrSlab, w, h, pitch, vec16,
                       x16, inv32(x16), clr, clrInts, reinterpret_cast<const char4*>(pat8), reinterpret_cast<float4*>(patF));
}

void orb_launch_resize(hipStream_t st, uint8_t* pyr, size_t pyrSlab, const OrbLevelGeom& src,
                       const OrbLevelGeom& dst, const int2* xtab, const int2* ytab, const uint4* xq, int nFrames)
{
    const int x4 = (dst.w + 3) / 4;
    dim3 grid((x4 + 63) / 64, (dst.h + 3) / 4, nFrames);
    // the flattened index decode (multiply-high) is exact while total * x4 < 2^32; 24-bit multiplies need
    // rows, pitches and row offsets below 2^24 / 2^32 (any realistic image)
    const double scale = (double)src.w / dst.w;
    const long long total = (long long)x4 * dst.h;
    const bool flat = total * x4 < (1ll << 32);
    if (xq && flat && (long long)src.h * src.pitch < (1ll << 31) && src.pitch < (1 << 24) && dst.h < (1 << 24)) {
        const long long groups = (long long)x4 * ((dst.h + RESIZE_ROWS - 1) / RESIZE_ROWS);
        hipLaunchKernelGGL(k_resize_level4p, dim3((unsigned)((groups + 255) / 256), nFrames), dim3(256), 0, st, pyr, pyrSlab,
                           src.pyrOff, src.pitch, dst.pyrOff, dst.pitch, dst.h, xq, ytab, x4, inv32(x4));
    }
    // window form needs sx(dx+3)+1 - (sx(dx) & ~3) <= 11, i.e. 3 + ceil(3*scale) + 1 <= 11
    else if (scale <= 2.2 && flat) {
        hipLaunchKernelGGL(k_resize_level4, dim3((unsigned)((total + 255) / 256), nFrames), dim3(256), 0, st, pyr, pyrSlab,
                           src.pyrOff, src.pitch, dst.pyrOff, dst.pitch, dst.w, dst.h, xtab, ytab, x4, inv32(x4));
    }
    else
        hipLaunchKernelGGL(k_resize_level, grid, dim3(64, 4), 0, st, pyr, pyrSlab, src.pyrOff, src.pitch,
                           dst.pyrOff, dst.pitch, dst.w, dst.h, xtab, ytab);
}

// resident workgroups of a persistent kernel on the current device (occupancy x CUs), remembered per (device, kernel, LDS)
static int pyr_resident_slots(const void* kern, size_t ldsBytes)
{
    static std::mutex mu;
    static std::map<std::tuple<int, const void*, size_t>, int> memo;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_tuple(dev, kern, ldsBytes);
    auto it = memo.find(key);
    if (it != memo.end()) return it->second;
    int occ = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, ldsBytes) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); occ = 0; }
    const int slots = occ > 0 && cus > 0 ? occ * cus : 0;
    memo[key] = slots;
    return slots;
}

// persist: the batch form (k_pyr_chain_p) where the chain allows it -- the caller offers it for batches that fill the chip
// several times over, ORB_PYR_PERSIST=1 takes the offer.  OFF by default: it hides the staging round trip as designed and the
// launches take as long as before (profiles/r05_pyr_persistent_attempt.txt).  Returns true when the
// persistent form was launched.
bool orb_launch_pyr_chain(hipStream_t st, const OrbPyrChain& C, const uint8_t* img, size_t rowStride, size_t frameStride,
                          uint8_t* pyr, size_t pyrSlab, const uint4* xqAll, const int2* ytAll, const int2* bandTab, int nFrames,
                          int* clr, int clrInts, const int8_t* pat8, float* patF, unsigned long long* stamps, bool persist)
{
    static const int rg = std::getenv("ORB_PYR_RG") ? std::atoi(std::getenv("ORB_PYR_RG")) : 4;
    // ORB_PYR_LDSMIN=<KB> (tuning): claim at least that much LDS per workgroup, i.e. FEWER workgroups of this latency-bound kernel
    // per CU, so that other lanes' kernels find LDS and wave slots beside it
    static const size_t ldsMin = std::getenv("ORB_PYR_LDSMIN") ? (size_t)std::max(0, std::min(64, std::atoi(std::getenv("ORB_PYR_LDSMIN")))) * 1024 : 0;
    // (read per OFFERED launch, not once: the tests switch them within one process; the few-frame launches never get here --
    // a getenv is a scan of the environment, three of them per launch were 2 us of a single frame's chain)
    const char* envP = persist ? std::getenv("ORB_PYR_PERSIST") : nullptr;
    const bool persistOn = envP && std::atoi(envP) != 0;
    // ORB_PYR_SLOTS=<percent> (tuning, tests): persistent workgroups as a share of the resident slots
    const char* envS = persistOn ? std::getenv("ORB_PYR_SLOTS") : nullptr;
    const bool debug = persistOn && std::getenv("ORB_PYR_DEBUG") != nullptr;
    const int slotPct = envS ? std::max(10, std::min(400, std::atoi(envS))) : 100;
    const size_t ldsBytes = std::max((size_t)C.ldsBytes, ldsMin);
    const long long nUnits = (long long)C.bands * nFrames;
    const long long chunks = (long long)C.srcRowsMax * C.cpr;
    if (persist && persistOn && rg == 4 && C.srcW >= 16 && chunks <= 256 * PYR_STAGE_MAX && nUnits * C.bands < (1ll << 32) && nUnits < (1ll << 31)) {
        const unsigned invBands = C.bands <= 1 ? 0u : (unsigned)(((1ull << 32) + C.bands - 1) / C.bands);
        auto gop = [&](auto kern) -> bool {
            const int slots = pyr_resident_slots(reinterpret_cast<const void*>(kern), ldsBytes);
            if (slots <= 0) return false;
            if (debug) std::fprintf(stderr, "[orb] k_pyr_chain_p: %d bands x %d frames, %lld chunks per band, LDS %zu, %d resident slots\n", C.bands, nFrames, chunks, ldsBytes, slots);
            const long long want = std::max(1ll, (long long)slots * slotPct / 100);
            if (nUnits < 2 * want) return false;                   // (nothing to prefetch for: the plain form)
            hipLaunchKernelGGL(kern, dim3((unsigned)want), dim3(256), ldsBytes, st, C, img, rowStride, frameStride, pyr, pyrSlab, xqAll,
                               ytAll, bandTab, clr, clrInts, reinterpret_cast<const char4*>(pat8), reinterpret_cast<float4*>(patF),
                               stamps, (int)nUnits, invBands);
            return true;
        };
        const int npf = (int)((chunks + 255) / 256);
        bool done;
        if (C.xqLdsN > 0) done = npf <= 2 ? gop(k_pyr_chain_p<4, true, 2>) : npf <= 4 ? gop(k_pyr_chain_p<4, true, 4>) : npf <= 6 ? gop(k_pyr_chain_p<4, true, 6>) : gop(k_pyr_chain_p<4, true, 8>);
        else done = npf <= 2 ? gop(k_pyr_chain_p<4, false, 2>) : npf <= 3 ? gop(k_pyr_chain_p<4, false, 3>) : npf <= 4 ? gop(k_pyr_chain_p<4, false, 4>) : npf <= 6 ? gop(k_pyr_chain_p<4, false, 6>) : gop(k_pyr_chain_p<4, false, 8>);
        if (done) return true;
    }
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(C.bands, nFrames), dim3(256), ldsBytes, st, C, img, rowStride, frameStride, pyr,
                           pyrSlab, xqAll, ytAll, bandTab, clr, clrInts, reinterpret_cast<const char4*>(pat8),
                           reinterpret_cast<float4*>(patF), stamps);
    };
    if (C.xqLdsN > 0) {
        if (rg == 1) go(k_pyr_chain<1, true>);
        else if (rg == 2) go(k_pyr_chain<2, true>);
        else go(k_pyr_chain<4, true>);
    }
    else if (rg == 1) go(k_pyr_chain<1, false>);
    else if (rg == 2) go(k_pyr_chain<2, false>);
    else go(k_pyr_chain<4, false>);
    return false;
}

// Levels M and M+1 in one launch (k_resize_pair); returns false when the pair is not eligible (the caller then
// uses orb_launch_resize for each level).
bool orb_launch_resize_pair(hipStream_t st, uint8_t* pyr, size_t pyrSlab, const OrbLevelGeom& S, const OrbLevelGeom& M,
                            const OrbLevelGeom& D, const uint4* xqM, const int2* ytabM, const uint4* xqD, const int2* ytabD,
                            int nFrames)
{
    if (!xqM || !xqD) return false;
    const int x4M = (M.w + 3) / 4, x4D = (D.w + 3) / 4;
    const double scaleD = (double)M.h / D.h;
    if (x4M < 2 || x4D < 2 || scaleD > 2.0 || (double)M.w / D.w > 2.0) return false;
    // rows of M one band needs: PAIR_ROWS * scale + 2, + 1 for rounding
    const int maxRows = (int)(PAIR_ROWS * scaleD) + 4;
    const int ldsPitchDw = x4M + 2;                             // 8-byte windows overrun a row by up to one dword
    const size_t lds = (size_t)maxRows * ldsPitchDw * 4;
    if (lds > 48 * 1024) return false;
    if ((long long)maxRows * x4M * x4M >= (1ll << 32) || (long long)PAIR_ROWS * x4D * x4D >= (1ll << 32)) return false;
    if ((long long)S.h * S.pitch >= (1ll << 31) || S.pitch >= (1 << 24) || M.pitch >= (1 << 24) || D.pitch >= (1 << 24)) return false;
    const int bands = (D.h + PAIR_ROWS - 1) / PAIR_ROWS;
    hipLaunchKernelGGL(k_resize_pair, dim3(bands, nFrames), dim3(256), lds, st, pyr, pyrSlab, S.pyrOff, S.pitch, M.pyrOff,
                       M.pitch, M.h, D.pyrOff, D.pitch, D.h, xqM, ytabM, x4M, inv32(x4M), xqD, ytabD, x4D, inv32(x4D), ldsPitchDw);
    return true;
}
