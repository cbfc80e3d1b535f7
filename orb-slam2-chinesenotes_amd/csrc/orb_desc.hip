// orb_desc.hip -- orientation + steered BRIEF, fused per keypoint, on gfx950.
// Reference: IC_Angle / computeOrientation src/ORBextractor.cc:78-115, GaussianBlur :1129-1130,
// computeOrbDescriptor :120-161, the tail of operator() :1118-1148.
//
// One wave64 per keypoint slot.  The 43x43 source patch (radius 15 for IC_Angle, 18 for the pattern,
// +3 for the 7-tap blur) is staged in LDS as aligned dwords (BORDER_REFLECT_101 resolved at load time
// on the rare keypoints within 21 px of the image border); the horizontal blur pass runs once over
// the patch with v_dot4_u32_u8 on funnel-shifted byte windows (8.8 fixed-point taps fit a byte), the
// vertical pass only at the 512 sampled points.  The blurred level is never written to HBM.
// Four __ballot()s of 64 comparisons ARE the 256-bit descriptor.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "orb_kernels.h"
#include "orb_wave.h"

#pragma clang fp contract(off)

#include "../../include/orb_sincos.h"

#define WAVE 64
typedef unsigned short orb_u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int orb_u32x4 __attribute__((ext_vector_type(4)));
typedef orb_u32x4 __attribute__((aligned(4))) orb_u32x4_a4;
#define PR 21                  // patch radius
#define PW 43                  // patch rows / useful columns
#define PB 48                  // LDS row pitch in bytes (12 aligned dwords cover xoff + 43 <= 46 bytes)
#define PDW (PB / 4)
#define HT_RP 50               // the row-blurred patch is kept TRANSPOSED: u16 per column (25 dwords: an odd dword pitch
                               // spreads the columns over the LDS banks); rows 0..47, of which 0..42 exist
#define HT_COLS 40             // columns = the 10 quads of LDS bytes that the pattern can reach

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    // cv::fastAtan2; constants are the float products p_k * (float)(180/pi)
    const float p1 = __uint_as_float(0x4265226fu), p3 = __uint_as_float(0xc19556eeu);
    const float p5 = __uint_as_float(0x410e9fbfu), p7 = __uint_as_float(0xc0228ad9u);
    const float eps = 2.2204460492503131e-16f;
    const float ax = fabsf(x), ay = fabsf(y);
    // one division for both octant halves: c = min(ax, ay) / (max(ax, ay) + eps), picked as the reference picks them
    const bool xGe = ax >= ay;
    const float c = __fdiv_rn(xGe ? ay : ax, __fadd_rn(xGe ? ax : ay, eps));
    const float c2 = __fmul_rn(c, c);
    float a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    if (!xGe) a = __fsub_rn(90.f, a);
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// 7-tap row blur at the 4 byte positions 4q..4q+3 of a row, from the 12-byte window (d0,d1,d2) =
// bytes 4q-4 .. 4q+7.  Taps k0 k1 k2 k3 k2 k1 k0 (8.8 fixed point; 18,34,49,55 unless the handle was given
// another OpenCV version's): two dot4 per output.  K0 = k0 | k1 << 8 | k2 << 16 | k3 << 24 (bytes b-3 .. b),
// K1 = k2 | k1 << 8 | k0 << 16 (bytes b+1 .. b+3).
__device__ __forceinline__ void hblur4(unsigned d0, unsigned d1, unsigned d2, unsigned out[4], unsigned K0, unsigned K1)
{
    const unsigned lo0 = __builtin_amdgcn_alignbyte(d1, d0, 1), hi0 = __builtin_amdgcn_alignbyte(d2, d1, 1);
    const unsigned lo1 = __builtin_amdgcn_alignbyte(d1, d0, 2), hi1 = __builtin_amdgcn_alignbyte(d2, d1, 2);
    const unsigned lo2 = __builtin_amdgcn_alignbyte(d1, d0, 3), hi2 = __builtin_amdgcn_alignbyte(d2, d1, 3);
    out[0] = __builtin_amdgcn_udot4(hi0, K1, __builtin_amdgcn_udot4(lo0, K0, 0u, false), false);
    out[1] = __builtin_amdgcn_udot4(hi1, K1, __builtin_amdgcn_udot4(lo1, K0, 0u, false), false);
    out[2] = __builtin_amdgcn_udot4(hi2, K1, __builtin_amdgcn_udot4(lo2, K0, 0u, false), false);
    out[3] = __builtin_amdgcn_udot4(d2, K1, __builtin_amdgcn_udot4(d1, K0, 0u, false), false);
}

__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_orient_desc(const OrbGeom G, const uint8_t* __restrict__ pyr,
                                                      size_t pyrSlab, const uint32_t* __restrict__ kpl,
                                                      const int* __restrict__ kpCount,
                                                      const float4* __restrict__ patF,
                                                      const uint4* __restrict__ angTab,
                                                      const uint32_t* __restrict__ hbTab,
                                                      orb_keypoint* __restrict__ kpsOut,
                                                      uint8_t* __restrict__ descOut, int cap,
                                                      int32_t* __restrict__ countsOut, int* __restrict__ errFlags,
                                                      int nFrames, unsigned invPerFrame, OrbGaussK gk, int perFrame)
{
    // ONE LDS region: first the raw patch (2 KB, dword rows with one dword of slack on both sides), later the
    // row-blurred patch H (4 KB, u16) written over it once every lane holds its blur outputs in registers.
    // 4.1 KB per workgroup instead of 6.2 KB lets the wave-slot limit (32 per CU), not LDS, set the occupancy.
    __shared__ __attribute__((aligned(16))) uint32_t ldsBuf[1032];
    static_assert(1032 >= 4 + 48 * PDW + 2 && 1032 * 4 >= HT_COLS * HT_RP * 2, "raw patch rows 0..47 (43..47: slack that is read, never used) and H");
    uint32_t* Pdw = ldsBuf + 4;                        // 16-byte aligned rows (48-byte pitch): staged with 16-byte stores
    const int lane = threadIdx.x;
    int slot, f;
    if (invPerFrame) {                                 // 1-D XCD-aware grid: a frame's keypoints share one L2
        if (!orb_xcd_decode(blockIdx.x, (unsigned)perFrame, invPerFrame, nFrames, f, slot)) return;
    } else {
        slot = blockIdx.x;
        f = blockIdx.y;
    }
    int level = 0;
    while (level + 1 < G.nlevels && slot >= G.L[level + 1].kpBase) level++;
    const OrbLevelGeom& L = G.L[level];
    const int k = slot - L.kpBase;
    const int* cnt = kpCount + f * ORB_MAX_LEVELS;
    int off = 0;
    for (int l = 0; l < level; l++) off += cnt[l];
    if (slot == 0) {
        int tot = 0;
        for (int l = 0; l < G.nlevels; l++) tot += cnt[l];
        if (lane == 0) {
            countsOut[f] = min(tot, cap);
            if (tot > cap) orb_flag_error(errFlags, f, 4);
        }
    }
    if (k >= cnt[level] || off + k >= cap) return;

    const uint32_t packed = kpl[(size_t)f * G.kpSlab + slot];
    // constant-table loads are issued here, long before their use, so that their latency hides behind the patch
    // staging (one wave per workgroup: nothing else would cover it)
    // (the pattern as floats: k_copy_level0 spreads the int8 pairs once per batch, 16 conversions per lane saved here)
    const float4 q0 = patF[lane], q1 = patF[64 + lane], q2 = patF[128 + lane], q3 = patF[192 + lane];
    const int angV = (min(lane, 61) >> 1) - 15, angH = lane & 1;
    const uint4* angT = angTab + ((angV < 0 ? -angV : angV) * 2 + angH) * 2;
    const uint4 mk = angT[0], wt = angT[1];
    const int x0 = (int)(packed >> 20), y0 = (int)(packed >> 8) & 0xFFF;
    const int resp = (int)(packed & 0xFF);
    const uint8_t* img = pyr + (size_t)f * pyrSlab + L.pyrOff;

    // ---- stage the patch: LDS row r = image row y0-21+r, LDS byte b = image column xa+b
    const int xl = x0 - PR;                            // image column of patch column 0 (may be < 0)
    const int xa = xl & ~3, xoff = xl - xa;            // (two's complement: floor to a multiple of 4)
    const bool interior = xl >= 0 && x0 + PR < L.w && y0 - PR >= 0 && y0 + PR < L.h;
    // horizontal-blur work items of this lane (see below): requested now, used after the staging
    const uint32_t* hbt = hbTab + xoff * 192 + lane;
    const unsigned hb0 = hbt[0], hb1 = hbt[64], hb2 = hbt[128];
    if (interior) {
        // 43 rows x 3 chunks of 16 bytes (the patch row starts at a 4-byte aligned column: unaligned 16-byte global loads),
        // 21 rows per pass, all three passes in flight
        const int row = (lane * 171) >> 9, c = lane - row * 3;     // lane / 3 for lane < 64
        const uint8_t* src = img + (size_t)(y0 - PR + row) * L.pitch + xa + 16 * c;
        uint8_t* dst = reinterpret_cast<uint8_t*>(Pdw) + row * PB + 16 * c;
        orb_u32x4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0;
        if (lane < 63) {
            v0 = *reinterpret_cast<const orb_u32x4_a4*>(src);
            v1 = *reinterpret_cast<const orb_u32x4_a4*>(src + (size_t)21 * L.pitch);
        }
        if (lane < 3) v2 = *reinterpret_cast<const orb_u32x4_a4*>(src + (size_t)42 * L.pitch);
        if (lane < 63) {
            *reinterpret_cast<orb_u32x4*>(dst) = v0;
            *reinterpret_cast<orb_u32x4*>(dst + 21 * PB) = v1;
        }
        if (lane < 3) *reinterpret_cast<orb_u32x4*>(dst + 42 * PB) = v2;
    } else if (lane < PB) {                            // BORDER_REFLECT_101, byte by byte
        uint8_t* Pb = reinterpret_cast<uint8_t*>(Pdw);
        const int gx = reflect101(xa + lane, L.w);
        for (int r = 0; r < PW; r++) {
            const int gy = reflect101(y0 - PR + r, L.h);
            Pb[r * PB + lane] = img[(size_t)gy * L.pitch + gx];
        }
    }
    __syncthreads();

    // ---- IC_Angle (:78-105): two lanes per patch row v = -15..15 (u = -15..0 and u = 1..16).  The 16 bytes of a
    // half row are funnel-shifted into 4 dwords and reduced with v_dot4_u32_u8 against a per-(|v|, half) table of
    // byte masks (1 inside the circular patch |u| <= umax[|v|]) and byte weights (u + 15 inside, 0 outside):
    //   s0 = sum I,  s1 = sum (u + 15) I   =>   m10 += s1 - 15 s0,  m01 += v s0       (exact integers)
    int m10 = 0, m01 = 0;
    if (lane < 62) {
        const int v = angV, hh = angH;
        const int b0 = xoff + PR - 15;                  // LDS byte of u = -15
        const unsigned sh = (unsigned)b0 & 3u;
        const uint32_t* p = Pdw + (PR + v) * PDW + (b0 >> 2) + 4 * hh;
        const unsigned d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
        const unsigned n0 = __builtin_amdgcn_alignbyte(d1, d0, sh), n1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
        const unsigned n2 = __builtin_amdgcn_alignbyte(d3, d2, sh), n3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
        const unsigned s0 = __builtin_amdgcn_udot4(n0, mk.x, __builtin_amdgcn_udot4(n1, mk.y,
                            __builtin_amdgcn_udot4(n2, mk.z, __builtin_amdgcn_udot4(n3, mk.w, 0u, false), false), false), false);
        const unsigned s1 = __builtin_amdgcn_udot4(n0, wt.x, __builtin_amdgcn_udot4(n1, wt.y,
                            __builtin_amdgcn_udot4(n2, wt.z, __builtin_amdgcn_udot4(n3, wt.w, 0u, false), false), false), false);
        m10 = (int)s1 - 15 * (int)s0;
        m01 = v * (int)s0;
    }
    m10 = orb_wave_sum(m10);
    m01 = orb_wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // ---- horizontal 7-tap pass.  The rotated pattern stays within 18 px of the keypoint (|x|,|y| <= 13), so only
    // LDS bytes xoff+3 .. xoff+39 are ever sampled: 10 quads starting at qFirst = (xoff+3)/4.  Static mapping lane ->
    // (row phase rp of 6, quad): row PAIRS 12 i + 2 rp, + 1 for i = 0..3 (LDS addresses are base + constant: no index
    // arithmetic), written TRANSPOSED -- H[column][row], a dword = two vertically adjacent rows of one column -- so that
    // the 7 vertical taps of a sample are 4 consecutive dwords.  The kernel is bound by the LDS pipe (r02 counters: 89 %
    // busy, 2/3 of it the 56 conflict-ridden u16 reads per lane of the row-major layout); this way a sample is two
    // ds_read2_b32 + 4 v_alignbit + 4 v_dot2_u32_u16.
    const int qFirst = (xoff + 3) >> 2;
    {
        // Work items = (row pair, quad) that the rotated pattern can reach: a sample lies within 18.4 px of the keypoint
        // (|x|, |y| <= 13 for every pattern point, checked by orb_extractor_set_pattern), so the row pairs near the top and
        // the bottom of the patch need 4..9 of the 10 quads -- 189 / 190 items instead of 240, three per lane instead of four.
        // The host tabulates them per xoff (orb_desc_hblur_table): source dword (relative to Pdw) | destination dword << 16.
        const unsigned items[3] = {hb0, hb1, hb2};
        uint32_t hv[3][4];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (items[i] != 0xffffffffu) {
                const uint32_t* p = Pdw + (items[i] & 0xffffu);
                unsigned o0[4], o1[4];
                hblur4(p[0], p[1], p[2], o0, gk.h0, gk.h1);
                hblur4(p[PDW], p[PDW + 1], p[PDW + 2], o1, gk.h0, gk.h1);
#pragma unroll
                for (int j = 0; j < 4; j++) hv[i][j] = o0[j] | (o1[j] << 16);
            }
        }
        __syncthreads();                                       // every read of the raw patch is done: H may overwrite it
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (items[i] != 0xffffffffu) {
                uint32_t* col = ldsBuf + (items[i] >> 16);      // dword (column 4 ql, rows 2 p / 2 p + 1)
#pragma unroll
                for (int j = 0; j < 4; j++) col[j * (HT_RP / 2)] = hv[i][j];
            }
        }
    }
    __syncthreads();

    // ---- steered BRIEF (:120-161): lane handles pairs lane, lane+64, lane+128, lane+192
    const float rad = __fmul_rn(angle, __uint_as_float(0x3c8efa35u));       // (float)(CV_PI/180.f)
    float a, b;
    orb_sincos(rad, &a, &b);
    // blurred sample at pattern point (px,py) rotated by the keypoint angle (GET_VALUE, :132-134)
    // cvRound(float) = round half to even = what adding 1.5 * 2^23 does to the mantissa (|value| < 2^22): the low
    // bits of the sum ARE the integer (biased by 0x4B400000), so rounding costs one v_add_f32 per coordinate and the
    // biases fold into one wave-uniform constant of the LDS index (all arithmetic mod 2^32).
    const unsigned kBias = 0x4B400000u;
    // column index through a 24-bit multiply-add: the low 24 bits of the biased column are 0x400000 + ic
    const unsigned eC = (unsigned)((xoff + PR - 4 * qFirst) * HT_RP + (PR - 3)) - 0x400000u * (unsigned)HT_RP - kBias;
    const unsigned K0 = gk.v0, K1 = gk.v1, K2 = gk.v2, K3 = gk.v3;    // taps, two per dot2: k0 | k1 << 16, k2 | k3 << 16, k2 | k1 << 16, k0
    // The two points of a pair are rotated TOGETHER: patF holds a pair as {x0, x1, y0, y1}, so row = x * b + y * a and
    // col = x * a - y * b of both points are four packed multiplies and two packed adds on aligned register pairs (+ two
    // for the rounding constant).  With the pair stored {x0, y0, x1, y1} the compiler packed (x, y) of ONE point instead and
    // then had to move six registers around per pair to add across the halves: 24 of the kernel's ~435 instructions (round 5).
    typedef float orb_f2 __attribute__((ext_vector_type(2)));
    const orb_f2 a2 = {a, a}, b2 = {b, b}, magic2 = {12582912.f, 12582912.f};
    auto blurred = [&](unsigned br, unsigned bc) -> int {
        // u16 index of the topmost tap: H[xoff + 21 + ic - 4 qFirst][21 + ir - 3]; its 7 rows lie in 4 consecutive dwords,
        // starting in the low (even index) or the high half of the first
        const unsigned e = __umul24(bc, (unsigned)HT_RP) + br + eC;
        const uint32_t* hw = ldsBuf + (e >> 1);
        const unsigned d0 = hw[0], d1 = hw[1], d2 = hw[2], d3 = hw[3];
        const unsigned sh = e << 4;                            // v_alignbit uses the low 5 bits: 0 or 16
        const unsigned p0 = __builtin_amdgcn_alignbit(d1, d0, sh), p1 = __builtin_amdgcn_alignbit(d2, d1, sh);
        const unsigned p2 = __builtin_amdgcn_alignbit(d3, d2, sh), p3 = __builtin_amdgcn_alignbit(d3, d3, sh);   // (high half x tap 0)
        unsigned acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u16x2, p3), __builtin_bit_cast(orb_u16x2, K3), 32768u, false);
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u16x2, p2), __builtin_bit_cast(orb_u16x2, K2), acc, false);
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u16x2, p1), __builtin_bit_cast(orb_u16x2, K1), acc, false);
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(orb_u16x2, p0), __builtin_bit_cast(orb_u16x2, K0), acc, false);
        return (int)min(255u, acc >> 16);
    };
    // t0 < t1 of a pair (GET_VALUE of both points, :132-134, :141-160)
    auto pair_bit = [&](const float4& q) -> bool {
        const orb_f2 X = {q.x, q.y}, Y = {q.z, q.w};
        const orb_f2 fr = X * b2 + Y * a2;                      // (float)(x * b + y * a): two roundings of the products, one of the sum
        const orb_f2 fc = X * a2 - Y * b2;
        const orb_f2 rr = fr + magic2, cc = fc + magic2;        // cvRound: see kBias above
        const int t0 = blurred(__float_as_uint(rr.x), __float_as_uint(cc.x));
        const int t1 = blurred(__float_as_uint(rr.y), __float_as_uint(cc.y));
        return t0 < t1;
    };
    const unsigned long long w0 = __ballot(pair_bit(q0));
    const unsigned long long w1 = __ballot(pair_bit(q1));
    const unsigned long long w2 = __ballot(pair_bit(q2));
    const unsigned long long w3 = __ballot(pair_bit(q3));

    if (lane < 4) {
        const unsigned long long w = lane == 0 ? w0 : lane == 1 ? w1 : lane == 2 ? w2 : w3;
        reinterpret_cast<unsigned long long*>(descOut + ((size_t)f * cap + off + k) * ORB_DESC_BYTES)[lane] = w;
    }
    if (lane == 0) {
        orb_keypoint kp;
        kp.x = (float)x0;
        kp.y = (float)y0;
        if (level != 0) {                               // :1140-1146
            kp.x = __fmul_rn(kp.x, L.scale);
            kp.y = __fmul_rn(kp.y, L.scale);
        }
        kp.size = L.sizeField;
        kp.angle = angle;
        kp.response = (float)resp;
        kp.octave = level;
        kp.class_id = -1;
        kpsOut[(size_t)f * cap + off + k] = kp;
    }
}

// The horizontal-blur work items of k_orient_desc for xoff = 0..3: [4][3][64] entries, source dword offset (relative to the
// raw patch) | destination dword offset << 16, 0xffffffff = none.  Row offset dy (from the keypoint) of the row-blurred
// patch is read by samples of rows dy-3 .. dy+3; a sample (ic, ir) is the rounding of a point within R = sqrt(13^2 + 13^2)
// of the keypoint, so |ic| <= round(sqrt(R^2 - (|ir| - 0.5)^2)).
void orb_desc_hblur_table(uint32_t* tab768)
{
    const double R2 = 13.0 * 13.0 + 13.0 * 13.0;
    int cmax[43];
    for (int r = 0; r < 43; r++) {
        const int ady = r > PR ? r - PR : PR - r, ir = ady > 3 ? ady - 3 : 0;
        const double m = ir > 0 ? ir - 0.5 : 0.0;
        int c = (int)(std::sqrt(R2 - m * m) + 0.5);
        cmax[r] = c > 18 ? 18 : c;
    }
    for (int xoff = 0; xoff < 4; xoff++) {
        uint32_t* t = tab768 + xoff * 192;
        for (int i = 0; i < 192; i++) t[i] = 0xffffffffu;
        const int qFirst = (xoff + 3) >> 2;
        int n = 0;
        for (int p = 0; p < 22; p++) {
            const int c = std::max(cmax[2 * p], 2 * p + 1 < 43 ? cmax[2 * p + 1] : 0);
            const int lo = (xoff + PR - c) >> 2, hi = (xoff + PR + c) >> 2;
            for (int q = lo; q <= hi; q++, n++) {
                const unsigned src = (unsigned)(2 * p * PDW + q - 1), dst = (unsigned)(4 * (q - qFirst) * (HT_RP / 2) + p);
                if (n < 192) t[(n >> 6) * 64 + (n & 63)] = src | (dst << 16);      // item n -> pass n / 64, lane n % 64
            }
        }
        if (n > 192) std::abort();                         // 189 / 190 by construction
    }
}

void orb_launch_orient_desc(hipStream_t st, const OrbGeom& G, const uint8_t* pyr, size_t pyrSlab,
                            const uint32_t* kpl, const int* kpCount, const float* patternF, const uint4* angTab, const uint32_t* hbTab,
                            orb_keypoint* kps, uint8_t* desc, int cap, int32_t* counts, int* errFlags,
                            int nFrames, const int* gaussTaps4, int slotLimit)
{
    // slotLimit > 0: only the keypoint slots below it (the levels that k_desc_level does not take)
    const int perFrame = slotLimit > 0 ? std::min(slotLimit, G.kpSlab) : G.kpSlab;
    static const int legacy[4] = {18, 34, 49, 55};
    const int* t = gaussTaps4 ? gaussTaps4 : legacy;
    OrbGaussK gk;                                                  // the taps packed as the kernel's dot4 / dot2 operands
    gk.h0 = (unsigned)t[0] | ((unsigned)t[1] << 8) | ((unsigned)t[2] << 16) | ((unsigned)t[3] << 24);
    gk.h1 = (unsigned)t[2] | ((unsigned)t[1] << 8) | ((unsigned)t[0] << 16);
    gk.v0 = (unsigned)t[0] | ((unsigned)t[1] << 16);
    gk.v1 = (unsigned)t[2] | ((unsigned)t[3] << 16);
    gk.v2 = (unsigned)t[2] | ((unsigned)t[1] << 16);
    gk.v3 = (unsigned)t[0];
    unsigned inv = 0;
    const unsigned wgs = orb_xcd_grid((unsigned)perFrame, nFrames, &inv);
    if (wgs)
        hipLaunchKernelGGL(k_orient_desc, dim3(wgs), dim3(WAVE), 0, st, G, pyr, pyrSlab, kpl, kpCount, reinterpret_cast<const float4*>(patternF), angTab, hbTab, kps,
                           desc, cap, counts, errFlags, nFrames, inv, gk, perFrame);
    else
        hipLaunchKernelGGL(k_orient_desc, dim3(perFrame, nFrames), dim3(WAVE), 0, st, G, pyr, pyrSlab, kpl, kpCount,
                           reinterpret_cast<const float4*>(patternF), angTab, hbTab, kps, desc, cap, counts, errFlags, nFrames, 0u, gk, perFrame);
}
