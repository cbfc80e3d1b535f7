// orb_desc_level.hip -- the descriptor stage LEVEL-RESIDENT, as the reference structures it: one Gaussian blur per pyramid
// level, then every keypoint of that level (reference src/ORBextractor.cc:1118-1136: GaussianBlur(workingMat, ..., 7x7, 2, 2,
// BORDER_REFLECT_101) once per level, computeDescriptors over the level's keypoints; orientation :78-115 on the raw level).
//
// k_orient_desc (orb_desc.hip) gives every keypoint a wave of its own: it stages a 43 x 43 patch, blurs its rows, and evaluates
// fastAtan2 + the double-precision cos / sin of ONE angle on all 64 lanes.  On the upper pyramid levels the patches of a frame
// overlap several times over (1.5x on level 0 ... 5.2x on level 7 at 640x480 / 1000 features), so the per-keypoint form stages
// and row-blurs the same pixels again and again.  k_desc_level gives a whole REGION of a level (the whole level where it fits
// the CU's LDS: levels >= 3 at 640x480) to one 512-thread workgroup:
//
//   stage     the region's rows once, 16-byte global loads, into LDS rows of an ODD dword pitch (column walks and the 31 rows of
//             an IC_Angle then spread over the banks), with the 3-px BORDER_REFLECT_101 frame written around it where the region
//             touches the image border, so that nothing later treats the border specially;
//   IC_Angle  a wave per keypoint straight out of the resident level (v_dot4_u32_u8 against the mask / weight tables of
//             orb_desc.hip, exact integers);
//   angles    ONE LANE per keypoint: fastAtan2 and orb_sincos cost ~75 vector instructions -- per 64 keypoints here, per keypoint
//             in the per-keypoint kernel; the 28-byte keypoint records leave from here;
//   blur      the full 7 x 7 Gaussian (8.8 fixed point, taps from the handle) of the region: a thread walks one dword column
//             (4 pixels) of a band of rows, the row pass with v_dot4_u32_u8 on funnel-shifted windows, the column pass with
//             v_dot2_u32_u16 over a register window of vertically packed row pairs; the blurred dwords WAIT IN REGISTERS
//             (<= DL_OUTS per thread) until every thread has read its raw rows -- a barrier -- and then overwrite the raw level in
//             place: no second buffer, no hazard;
//   sample    a wave per keypoint: the 512 rotated pattern points are single byte reads of the blurred level (the per-keypoint
//             kernel pays a 7-tap column pass per sample), four ballots are the descriptor.
//
// Arithmetic is that of k_orient_desc (same tables, same float sequence), so results are bit-identical; the extractor runs
// this kernel for the levels its plan covers and k_orient_desc for the others (and for launches of a few frames, where a
// workgroup's chain of phases would be the latency of the call).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "orb_kernels.h"
#include "orb_wave.h"

#pragma clang fp contract(off)

#include "../../include/orb_sincos.h"

#define WAVE 64
#define DL_THREADS 512
#define DL_WAVES (DL_THREADS / WAVE)
#define DL_OUTS ORB_DESC_LEVEL_OUTS
#define DL_STAGE_BATCH 10
// diagnostics (orb_extractor_set_desc_stamps): thread 0 of a workgroup leaves the 100 MHz clock at its phase boundaries
#define DL_STAMP(k) do { if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)

typedef unsigned short dl_u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int dl_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int dl_reflect101(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// cv::fastAtan2 -- the same operation sequence as orb_desc.hip's
__device__ __forceinline__ float dl_fast_atan2_deg(float y, float x)
{
    const float p1 = __uint_as_float(0x4265226fu), p3 = __uint_as_float(0xc19556eeu);
    const float p5 = __uint_as_float(0x410e9fbfu), p7 = __uint_as_float(0xc0228ad9u);
    const float eps = 2.2204460492503131e-16f;
    const float ax = fabsf(x), ay = fabsf(y);
    const bool xGe = ax >= ay;
    const float c = __fdiv_rn(xGe ? ay : ax, __fadd_rn(xGe ? ax : ay, eps));
    const float c2 = __fmul_rn(c, c);
    float a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    if (!xGe) a = __fsub_rn(90.f, a);
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// 7-tap row blur at the 4 byte positions of the middle dword d1 of the window (d0, d1, d2) (see hblur4 in orb_desc.hip)
__device__ __forceinline__ void dl_hblur4(unsigned d0, unsigned d1, unsigned d2, unsigned out[4], unsigned K0, unsigned K1)
{
    const unsigned lo0 = __builtin_amdgcn_alignbyte(d1, d0, 1), hi0 = __builtin_amdgcn_alignbyte(d2, d1, 1);
    const unsigned lo1 = __builtin_amdgcn_alignbyte(d1, d0, 2), hi1 = __builtin_amdgcn_alignbyte(d2, d1, 2);
    const unsigned lo2 = __builtin_amdgcn_alignbyte(d1, d0, 3), hi2 = __builtin_amdgcn_alignbyte(d2, d1, 3);
    out[0] = __builtin_amdgcn_udot4(hi0, K1, __builtin_amdgcn_udot4(lo0, K0, 0u, false), false);
    out[1] = __builtin_amdgcn_udot4(hi1, K1, __builtin_amdgcn_udot4(lo1, K0, 0u, false), false);
    out[2] = __builtin_amdgcn_udot4(hi2, K1, __builtin_amdgcn_udot4(lo2, K0, 0u, false), false);
    out[3] = __builtin_amdgcn_udot4(d2, K1, __builtin_amdgcn_udot4(d1, K0, 0u, false), false);
}

// grid: x = region + nRegions * frame (regions of a frame side by side: a big level's workgroup and a small one's share a CU).
// LDS: [image: imgRows x pd dwords] [kpP: u32 x kcap] [kpA: float4 x kcap] [kpK: u16 x kcap]
__global__ __launch_bounds__(DL_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_desc_level(const OrbGeom G, const OrbDescPlan P, const uint8_t* __restrict__ pyr, size_t pyrSlab,
                  const uint32_t* __restrict__ kpl, const int* __restrict__ kpCount, const float4* __restrict__ patF,
                  const uint4* __restrict__ angTab, orb_keypoint* __restrict__ kpsOut, uint8_t* __restrict__ descOut, int cap,
                  int nFrames, OrbGaussK gk, unsigned long long* __restrict__ stamps)
{
    extern __shared__ uint32_t dlsm[];
    __shared__ int nKs;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ri = blockIdx.x % P.nRegions, f = blockIdx.x / P.nRegions;
    const OrbDescRegion R = P.R[ri];
    const int level = R.level;
    const OrbLevelGeom& L = G.L[level];
    const int pd = R.pd, rw4 = R.rw4, rh = R.rh, imgRows = R.imgRows, kcap = L.kpCap;
    uint32_t* img = dlsm;
    uint32_t* kpP = img + ((imgRows * pd + 3) & ~3);
    float4* kpA = reinterpret_cast<float4*>(kpP + ((kcap + 3) & ~3));
    uint16_t* kpK = reinterpret_cast<uint16_t*>(kpA + kcap);
    uint8_t* imgB = reinterpret_cast<uint8_t*>(img);

    const int* cnt = kpCount + f * ORB_MAX_LEVELS;
    const int nLevel = min(cnt[level], kcap);
    int off = 0;
    for (int l = 0; l < level; l++) off += cnt[l];
    if (tid == 0) nKs = 0;
    DL_STAMP(0);
    __syncthreads();

    // ---- the region's keypoints (order is irrelevant: every keypoint keeps its list position k, its output slot)
    const uint32_t* kl = kpl + (size_t)f * G.kpSlab + L.kpBase;
    for (int k = tid; k < nLevel; k += DL_THREADS) {
        const uint32_t packed = kl[k];
        const int x = (int)(packed >> 20), y = (int)(packed >> 8) & 0xFFF;
        if (x >= R.cx0 && x < R.cx1 && y >= R.cy0 && y < R.cy1 && off + k < cap) {
            const int s = atomicAdd(&nKs, 1);
            kpP[s] = packed;
            kpK[s] = (uint16_t)k;
        }
    }
    // (a region without keypoints still has to reach the barriers below with every thread: uniform exit after the count is known)

    // ---- stage: LDS row lr holds level row reflect101(ry0 + lr - 3); a row is rw4 dwords behind one pad dword
    {
        const uint8_t* src = pyr + (size_t)f * pyrSlab + L.pyrOff;
        const int nC = (rw4 + 3) >> 2;                             // 16-byte chunks per row
        const unsigned invC = R.invC;                              // ceil(2^32 / nC) (0: nC == 1)
        const int items = imgRows * nC;
        // every load of a thread is requested before the first is stored (a loop of load -> wait -> store was a chain of
        // ~10 memory round trips per workgroup); DL_STAGE_BATCH x 512 chunks cover the 79 KB a region may take
        auto where = [&](int i, int& lr, int& c) {
            lr = nC == 1 ? i : (int)__umulhi((unsigned)i, invC);
            c = i - lr * nC;
        };
        for (int i0 = tid; i0 < items; i0 += DL_THREADS * DL_STAGE_BATCH) {
            dl_u32x4 v[DL_STAGE_BATCH];
#pragma unroll
            for (int b = 0; b < DL_STAGE_BATCH; b++) {
                const int i = i0 + b * DL_THREADS;
                if (i < items) {
                    int lr, c;
                    where(i, lr, c);
                    const int sr = dl_reflect101(R.ry0 + lr - 3, L.h);
                    v[b] = *reinterpret_cast<const dl_u32x4*>(src + (size_t)sr * L.pitch + R.rx0 + 16 * c);
                }
            }
#pragma unroll
            for (int b = 0; b < DL_STAGE_BATCH; b++) {
                const int i = i0 + b * DL_THREADS;
                if (i < items) {
                    int lr, c;
                    where(i, lr, c);
                    uint32_t* dst = img + lr * pd + 1 + 4 * c;
                    const int left = rw4 - 4 * c;
                    dst[0] = v[b].x;
                    if (left > 1) dst[1] = v[b].y;
                    if (left > 2) dst[2] = v[b].z;
                    if (left > 3) dst[3] = v[b].w;
                }
            }
        }
    }
    DL_STAMP(1);
    __syncthreads();
    const int nK = nKs;
    if (nK == 0) return;

    // ---- the reflect-101 frame left / right of the image (only where the region's edge is the image's; a region edge inside
    // the image has 21 real pixels of halo instead, and the blurred pixels within 3 of it are never sampled)
    for (int i = tid; i < 2 * imgRows; i += DL_THREADS) {
        const int lr = i >> 1, side = i & 1;
        uint8_t* rowB = imgB + (size_t)(lr * pd + 1) * 4;          // byte of region column 0
        if (side == 0) {
            if (R.padL) { const uint8_t a = rowB[1], b = rowB[2], c = rowB[3]; rowB[-1] = a; rowB[-2] = b; rowB[-3] = c; }
        } else if (R.padR) {
            const int w = R.rwPx;                                   // region columns [0, w)
            const uint8_t a = rowB[w - 2], b = rowB[w - 3], c = rowB[w - 4];
            rowB[w] = a; rowB[w + 1] = b; rowB[w + 2] = c;
        }
    }

    // ---- IC_Angle (:78-105), a wave per keypoint, from the resident level (see orb_desc.hip for the table scheme)
    {
        const int angV = (min(lane, 61) >> 1) - 15, angH = lane & 1;
        const uint4* angT = angTab + ((angV < 0 ? -angV : angV) * 2 + angH) * 2;
        const uint4 mk = angT[0], wt = angT[1];
        for (int j = wv; j < nK; j += DL_WAVES) {
            const uint32_t packed = kpP[j];
            const int x0 = (int)(packed >> 20), y0 = (int)(packed >> 8) & 0xFFF;
            int m10 = 0, m01 = 0;
            if (lane < 62) {
                const int b0 = ((y0 + angV - R.ry0 + 3) * pd + 1) * 4 + (x0 - 15 + 16 * angH - R.rx0);
                const unsigned sh = (unsigned)b0 & 3u;
                const uint32_t* p = img + (b0 >> 2);
                const unsigned d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
                const unsigned n0 = __builtin_amdgcn_alignbyte(d1, d0, sh), n1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
                const unsigned n2 = __builtin_amdgcn_alignbyte(d3, d2, sh), n3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
                const unsigned s0 = __builtin_amdgcn_udot4(n0, mk.x, __builtin_amdgcn_udot4(n1, mk.y,
                                    __builtin_amdgcn_udot4(n2, mk.z, __builtin_amdgcn_udot4(n3, mk.w, 0u, false), false), false), false);
                const unsigned s1 = __builtin_amdgcn_udot4(n0, wt.x, __builtin_amdgcn_udot4(n1, wt.y,
                                    __builtin_amdgcn_udot4(n2, wt.z, __builtin_amdgcn_udot4(n3, wt.w, 0u, false), false), false), false);
                m10 = (int)s1 - 15 * (int)s0;
                m01 = angV * (int)s0;
            }
            m10 = orb_wave_sum(m10);
            m01 = orb_wave_sum(m01);
            if (lane == 0) kpA[j] = make_float4(__int_as_float(m10), __int_as_float(m01), 0.f, 0.f);
        }
    }
    DL_STAMP(2);
    __syncthreads();

    // ---- angles: a lane per keypoint; the keypoint records (:1138-1148) leave from here
    if (tid < nK) {
        const float4 m = kpA[tid];
        const float angle = dl_fast_atan2_deg((float)__float_as_int(m.y), (float)__float_as_int(m.x));
        const float rad = __fmul_rn(angle, __uint_as_float(0x3c8efa35u));      // (float)(CV_PI/180.f)
        float a, b;
        orb_sincos(rad, &a, &b);
        kpA[tid] = make_float4(angle, a, b, 0.f);
        const uint32_t packed = kpP[tid];
        const int x0 = (int)(packed >> 20), y0 = (int)(packed >> 8) & 0xFFF;
        orb_keypoint kp;
        kp.x = (float)x0;
        kp.y = (float)y0;
        if (level != 0) {
            kp.x = __fmul_rn(kp.x, L.scale);
            kp.y = __fmul_rn(kp.y, L.scale);
        }
        kp.size = L.sizeField;
        kp.angle = angle;
        kp.response = (float)(packed & 0xFF);
        kp.octave = level;
        kp.class_id = -1;
        kpsOut[(size_t)f * cap + off + kpK[tid]] = kp;
    }

    DL_STAMP(3);
    // ---- blur: thread = (band, dword column); the results wait in registers for the barrier
    uint32_t outv[DL_OUTS];
    const int bh = R.bh;
    const bool blurT = tid < R.nBands * rw4;
    int band = 0, col = 0;
    if (blurT) {
        band = rw4 == 1 ? tid : (int)__umulhi((unsigned)tid, R.invW);
        col = tid - band * rw4;
        const uint32_t* wp = img + band * bh * pd + col;             // LDS row band * bh = region row band * bh - 3
        unsigned hp[4] = {0, 0, 0, 0};                              // the previous row's sums
        unsigned pr[6][4];                                          // vertically packed row pairs (s-6, s-5) ... (s-1, s): a ring
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int x = 0; x < 4; x++) pr[i][x] = 0;
#pragma unroll
        for (int s = 0; s < DL_OUTS + 6; s++) {
            if (s < bh + 6) {
                unsigned hn[4];
                dl_hblur4(wp[0], wp[1], wp[2], hn, gk.h0, gk.h1);
                wp += pd;
                if (s >= 6) {
                    // rows s-6 .. s with taps k0 k1 k2 k3 k2 k1 k0: pairs (s-6, s-5), (s-4, s-3), (s-2, s-1) and row s alone
                    unsigned acc[4];
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        unsigned a = __umul24(hn[x], gk.v3) + 32768u;
                        a = __builtin_amdgcn_udot2(__builtin_bit_cast(dl_u16x2, pr[(s + 0) % 6][x]), __builtin_bit_cast(dl_u16x2, gk.v0), a, false);
                        a = __builtin_amdgcn_udot2(__builtin_bit_cast(dl_u16x2, pr[(s + 2) % 6][x]), __builtin_bit_cast(dl_u16x2, gk.v1), a, false);
                        a = __builtin_amdgcn_udot2(__builtin_bit_cast(dl_u16x2, pr[(s + 4) % 6][x]), __builtin_bit_cast(dl_u16x2, gk.v2), a, false);
                        acc[x] = min(a, 0x00FFFFFFu);               // saturate_cast<uchar>(acc >> 16): byte 2
                    }
                    const unsigned lo = __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u);
                    const unsigned hi = __builtin_amdgcn_perm(acc[3], acc[2], 0x0c0c0602u);
                    outv[s - 6] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
                }
                // the pair (s-1, s) takes the ring slot of (s-7, s-6), i.e. index (s-1) % 6 == (s+5) % 6
#pragma unroll
                for (int x = 0; x < 4; x++) { pr[(s + 5) % 6][x] = hp[x] | (hn[x] << 16); hp[x] = hn[x]; }
            }
        }
    }
    DL_STAMP(4);
    __syncthreads();                                               // every raw row has been read
    if (blurT) {
        uint32_t* op = img + (band * bh + 3) * pd + col + 1;
        const int r0 = band * bh;
#pragma unroll
        for (int o = 0; o < DL_OUTS; o++)
            if (o < bh && r0 + o < rh) op[o * pd] = outv[o];
    }
    __syncthreads();
    DL_STAMP(5);

    // ---- steered BRIEF (:120-161), a wave per keypoint: lane handles pairs lane, lane+64, lane+128, lane+192
    {
        const float4 q0 = patF[lane], q1 = patF[64 + lane], q2 = patF[128 + lane], q3 = patF[192 + lane];
        const unsigned P4 = 4u * (unsigned)pd, kBias = 0x4B400000u;
        for (int j = wv; j < nK; j += DL_WAVES) {
            const uint32_t packed = kpP[j];
            const float4 ang = kpA[j];
            const int x0 = (int)(packed >> 20), y0 = (int)(packed >> 8) & 0xFFF;
            const float a = ang.y, b = ang.z;
            // byte of (x0 + ic, y0 + ir) = ir * 4 pd + ic + base; the rounded coordinates arrive biased (cvRound by adding
            // 1.5 * 2^23: the low 24 bits of the sum are 0x400000 + i), the biases fold into one wave-uniform constant
            const unsigned base = (unsigned)(((y0 - R.ry0 + 3) * pd + 1) * 4 + (x0 - R.rx0));
            const unsigned eC = base - 0x400000u * P4 - kBias;
            // (a pair's two points rotated together: patF = {x0, x1, y0, y1}, see orb_desc.hip)
            typedef float dl_f2 __attribute__((ext_vector_type(2)));
            const dl_f2 a2 = {a, a}, b2 = {b, b}, magic2 = {12582912.f, 12582912.f};
            auto pair_bit = [&](const float4& q) -> bool {
                const dl_f2 X = {q.x, q.y}, Y = {q.z, q.w};
                const dl_f2 fr = X * b2 + Y * a2;
                const dl_f2 fc = X * a2 - Y * b2;
                const dl_f2 rr = fr + magic2, cc = fc + magic2;
                const int t0 = (int)imgB[__umul24(__float_as_uint(rr.x), P4) + __float_as_uint(cc.x) + eC];
                const int t1 = (int)imgB[__umul24(__float_as_uint(rr.y), P4) + __float_as_uint(cc.y) + eC];
                return t0 < t1;
            };
            const unsigned long long w0 = __ballot(pair_bit(q0));
            const unsigned long long w1 = __ballot(pair_bit(q1));
            const unsigned long long w2 = __ballot(pair_bit(q2));
            const unsigned long long w3 = __ballot(pair_bit(q3));
            if (lane < 4) {
                const unsigned long long w = lane == 0 ? w0 : lane == 1 ? w1 : lane == 2 ? w2 : w3;
                reinterpret_cast<unsigned long long*>(descOut + ((size_t)f * cap + off + kpK[j]) * ORB_DESC_BYTES)[lane] = w;
            }
        }
    }
    DL_STAMP(6);
    if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)ri << 32) | (unsigned)nK;
}

// ---- host side: which levels run level-resident, and how --------------------------------------------------------------------
// A level is cut into nT vertical tiles (1 = the whole level): tile t owns the keypoints of its core rows and computes the
// blur of the core +- 18 rows (a rotated pattern point lies within 18 px of its keypoint), reading 3 more on each side.  A
// region must fit the blur walk (nBands = 512 / dwords-per-row bands of bh <= DL_OUTS rows) and HALF the CU's LDS, so that two
// workgroups share a CU whatever their levels (the kernel's registers allow two 8-wave workgroups per CU, not more).  A level
// is taken while the model below says its keypoints get cheaper than the per-keypoint kernel's ~435 vector instructions;
// level 0 always stays with k_orient_desc (which also writes the frame's keypoint count).
static size_t region_lds(int imgRows, int pd, int kpCap)
{
    return (size_t)((imgRows * pd + 3) & ~3) * 4 + (size_t)((kpCap + 3) & ~3) * 4 + (size_t)kpCap * (16 + 2) + 16;
}

static int plan_level(const OrbLevelGeom& L, int level, size_t ldsCap, OrbDescRegion* out, int maxOut)
{
    const int nD = (L.w + 3) / 4;
    if (L.w < 44 || L.h < 44 || nD > DL_THREADS) return 0;
    if (((nD + 3) / 4) * 16 > L.pitch) return 0;                   // the last 16-byte chunk of a row stays inside the row pitch
    const int nBands = DL_THREADS / nD;
    int pd = nD + 2;
    if (!(pd & 1)) pd++;
    for (int nT = 1; nT <= maxOut && nT <= 8; nT++) {
        const int core = (L.h + nT - 1) / nT;
        if (nT > 1 && core < 40) break;
        bool ok = true;
        long long rowsSum = 0;
        for (int t = 0; t < nT && ok; t++) {
            const int c0 = t * core, c1 = std::min(L.h, c0 + core);
            if (c1 <= c0) { ok = false; break; }
            const int r0 = std::max(0, c0 - 18), r1 = std::min(L.h, c1 + 18);
            const int rh = r1 - r0, bh = (rh + nBands - 1) / nBands;
            const int imgRows = std::max(rh, nBands * bh) + 6;
            const size_t lds = region_lds(imgRows, pd, L.kpCap);
            if (bh > DL_OUTS || lds > ldsCap) { ok = false; break; }
            OrbDescRegion& R = out[t];
            std::memset(&R, 0, sizeof(R));
            R.level = (short)level;
            R.rx0 = 0; R.ry0 = (short)r0;
            R.rw4 = (short)nD; R.rh = (short)rh; R.rwPx = (short)L.w;
            R.cx0 = 0; R.cx1 = (short)L.w; R.cy0 = (short)c0; R.cy1 = (short)(t == nT - 1 ? L.h : c1);
            R.pd = (short)pd;
            R.nBands = (short)nBands; R.bh = (short)bh;
            R.imgRows = (short)imgRows;
            R.padL = R.padR = 1;
            const int nC = (nD + 3) / 4;
            R.invC = nC > 1 ? (unsigned)(((1ull << 32) + nC - 1) / nC) : 0u;
            R.invW = nD > 1 ? (unsigned)(((1ull << 32) + nD - 1) / nD) : 0u;
            R.ldsBytes = (int)lds;
            rowsSum += (long long)rh * (1.0 + 2.4 / bh) * 1000;     // + the 6 rows of row blur a band computes beyond its own (40 % of a row's work)
        }
        if (!ok) continue;
        // vector instructions per keypoint: ~135 (IC_Angle, samples, shares) + the blur of rowsSum x w pixels at ~10.25 per pixel
        // and lane, over the level's quota of keypoints -- against ~435 of the per-keypoint kernel
        const double perKp = 135.0 + (double)rowsSum / 1000.0 * L.w * 10.25 / 64.0 / std::max(1, L.quota);
        return perKp < 0.88 * 435.0 ? nT : 0;
    }
    return 0;
}

void orb_desc_level_plan(const OrbGeom& G, OrbDescPlan* P)
{
    std::memset(P, 0, sizeof(*P));
    P->firstLevel = G.nlevels;
    // OFF by default -- measured on an MI355X (512 x 640x480): the per-keypoint kernel alone 0.391 ms per batch;
    // levels 2..7 through this kernel + levels 0..1 per keypoint 0.396 ms in one stream, 0.412 side by side on two: the kernel
    // issues 18 % fewer vector instructions for the batch but at 61 % of the issue rate (two 8-wave workgroups per CU, a third of
    // their life in staging and at barriers), against 94 % for 32 independent waves per CU.  ORB_DESC_LEVEL=1 switches it on
    // (read at every geometry build); tests/test_gpu_desc_level.py runs it.
    const int envOn = std::getenv("ORB_DESC_LEVEL") ? std::atoi(std::getenv("ORB_DESC_LEVEL")) : 0;
    if (!envOn) return;
    // two workgroups per CU (160 KB) by default; ORB_DESC_LEVEL_LDS=<KB> for experiments (e.g. 100: one per CU, fewer tiles)
    const size_t ldsCap = (size_t)(std::getenv("ORB_DESC_LEVEL_LDS") ? std::max(16, std::min(156, std::atoi(std::getenv("ORB_DESC_LEVEL_LDS")))) : 79) * 1024;
    // levels from the top (smallest) down while they qualify; regions then in level order, so that neighbours in the grid are
    // a frame's large and small regions
    OrbDescRegion tmp[ORB_MAX_LEVELS][8];
    int nT[ORB_MAX_LEVELS] = {0};
    int first = G.nlevels, total = 0;
    for (int l = G.nlevels - 1; l >= 1; l--) {
        const int n = plan_level(G.L[l], l, ldsCap, tmp[l], 8);
        if (n == 0 || total + n > ORB_DESC_MAX_REGIONS) break;
        nT[l] = n;
        total += n;
        first = l;
    }
    for (int l = first; l < G.nlevels; l++)
        for (int t = 0; t < nT[l]; t++) {
            P->R[P->nRegions++] = tmp[l][t];
            P->ldsMax = std::max(P->ldsMax, tmp[l][t].ldsBytes);
        }
    P->firstLevel = first;
}

int orb_launch_desc_level(hipStream_t st, const OrbGeom& G, const OrbDescPlan& P, const uint8_t* pyr, size_t pyrSlab, const uint32_t* kpl,
                          const int* kpCount, const float* patternF, const uint4* angTab, orb_keypoint* kps, uint8_t* desc, int cap,
                          int nFrames, const int* gaussTaps4, unsigned long long* stamps, size_t stampCap)
{
    if (P.nRegions == 0 || nFrames == 0) return 0;
    static const int legacy[4] = {18, 34, 49, 55};
    const int* t = gaussTaps4 ? gaussTaps4 : legacy;
    OrbGaussK gk;
    gk.h0 = (unsigned)t[0] | ((unsigned)t[1] << 8) | ((unsigned)t[2] << 16) | ((unsigned)t[3] << 24);
    gk.h1 = (unsigned)t[2] | ((unsigned)t[1] << 8) | ((unsigned)t[0] << 16);
    gk.v0 = (unsigned)t[0] | ((unsigned)t[1] << 16);
    gk.v1 = (unsigned)t[2] | ((unsigned)t[3] << 16);
    gk.v2 = (unsigned)t[2] | ((unsigned)t[1] << 16);
    gk.v3 = (unsigned)t[0];
    static bool attrSet = false;
    if (!attrSet) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_desc_level), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess)
            return -1;
        attrSet = true;
    }
    const char* minEnv = std::getenv("ORB_DESC_LEVEL_LDSMIN");       // (tuning: claim more LDS than needed -> fewer workgroups per CU)
    const size_t ldsLaunch = std::max<size_t>((size_t)P.ldsMax, minEnv ? (size_t)std::min(156, std::atoi(minEnv)) * 1024 : 0);
    hipLaunchKernelGGL(k_desc_level, dim3((unsigned)P.nRegions * (unsigned)nFrames), dim3(DL_THREADS), ldsLaunch, st, G, P, pyr, pyrSlab,
                       kpl, kpCount, reinterpret_cast<const float4*>(patternF), angTab, kps, desc, cap, nFrames, gk,
                       (size_t)P.nRegions * nFrames * 8 <= stampCap ? stamps : nullptr);
    return 0;
}
