// orb_extract_kernels.hip -- gfx950 kernels of the ORB extractor (wave64, LDS-tiled, no MFMA:
// this is integer/bitwise stencil + gather work bounded by HBM/LDS, not a dense contraction).
//
// Stage map (reference src/ORBextractor.cc):
//   k_copy_level0 / k_resize_level   ComputePyramid            :1153-1180  (cv::resize INTER_LINEAR)
//   (k_fast_cells lives in orb_fast.hip)
//   (k_quadtree lives in orb_quadtree.hip)
//   k_orient_desc                    IC_Angle, GaussianBlur, computeOrbDescriptor, operator() tail
//                                    :78-171, :1118-1148
// Every kernel takes blockIdx.y (or .z) = frame: batched frames are independent.
#include "orb_kernels.h"

#pragma clang fp contract(off)

#include "../../include/orb_sincos.h"

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// Level 0: copy of the input into the pitch-aligned pyramid slab (reference :1173 copies the image
// into its bordered buffer).  One thread per 16 output bytes.
__global__ __launch_bounds__(256) void k_copy_level0(const uint8_t* __restrict__ src, size_t rowStride,
                                                     size_t frameStride, uint8_t* __restrict__ pyr,
                                                     size_t pyrSlab, int w, int h, int pitch, int vec16)
{
    const int x16 = blockIdx.x * blockDim.x + threadIdx.x;   // 16-byte column
    const int y = blockIdx.y;
    const int f = blockIdx.z;
    if (x16 * 16 >= w) return;
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
// Orientation + steered BRIEF, fused per keypoint.  One wave64 per keypoint slot.
// The 43x43 source patch (radius 15 for IC_Angle, 18 for the pattern, +3 for the 7-tap blur) is
// staged in LDS with BORDER_REFLECT_101 resolved at load time; the horizontal blur pass is done
// once for the patch (u16, max 255*257 fits), the vertical pass only at the 512 sampled points.
// The blurred level is never written to HBM.
#define PR 21
#define PW 43
#define PP 44
#define HW 37
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
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(WAVE) void k_orient_desc(const OrbGeom G, const uint8_t* __restrict__ pyr,
                                                      size_t pyrSlab, const uint32_t* __restrict__ kpl,
                                                      const int* __restrict__ kpCount,
                                                      const int8_t* __restrict__ pattern,
                                                      orb_keypoint* __restrict__ kpsOut,
                                                      uint8_t* __restrict__ descOut, int cap,
                                                      int32_t* __restrict__ countsOut, int* __restrict__ errFlags)
{
    __shared__ uint8_t P[PW * PP];
    __shared__ uint16_t H[PW * (HW + 1)];
    const int lane = threadIdx.x;
    const int slot = blockIdx.x, f = blockIdx.y;
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
            if (tot > cap) atomicOr(&errFlags[f], 4);
        }
    }
    if (k >= cnt[level] || off + k >= cap) return;

    const uint32_t packed = kpl[(size_t)f * G.kpSlab + slot];
    const int x0 = (int)(packed >> 20), y0 = (int)(packed >> 8) & 0xFFF;
    const int resp = (int)(packed & 0xFF);
    const uint8_t* img = pyr + (size_t)f * pyrSlab + L.pyrOff;

    // ---- stage the patch (lanes over columns)
    if (lane < PW) {
        const int gx = reflect101(x0 - PR + lane, L.w);
        for (int r = 0; r < PW; r++) {
            const int gy = reflect101(y0 - PR + r, L.h);
            P[r * PP + lane] = img[(size_t)gy * L.pitch + gx];
        }
    }
    __syncthreads();

    // ---- IC_Angle (:78-105): lane v handles patch row v-15
    int m10 = 0, m01 = 0;
    if (lane < 31) {
        const int v = lane - 15;
        const int d = (int)(G.umaxPacked >> (4 * (v < 0 ? -v : v))) & 15;
        const uint8_t* row = P + (PR + v) * PP + PR;
        int s0 = 0, s1 = 0;
        for (int u = -d; u <= d; u++) {
            const int val = row[u];
            s0 += val;
            s1 += u * val;
        }
        m10 = s1;
        m01 = v * s0;
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // ---- horizontal 7-tap pass (8.8 fixed point taps 18,34,49,55,49,34,18)
    for (int idx = lane; idx < PW * HW; idx += WAVE) {
        const int r = idx / HW, c = idx - r * HW;
        const uint8_t* s = P + r * PP + c;
        const int acc = 18 * (s[0] + s[6]) + 34 * (s[1] + s[5]) + 49 * (s[2] + s[4]) + 55 * s[3];
        H[r * (HW + 1) + c] = (uint16_t)acc;
    }
    __syncthreads();

    // ---- steered BRIEF (:120-161): lane handles pairs lane, lane+64, lane+128, lane+192
    const float rad = __fmul_rn(angle, __uint_as_float(0x3c8efa35u));       // (float)(CV_PI/180.f)
    float a, b;
    orb_sincos(rad, &a, &b);
    // blurred sample at pattern point (px,py) rotated by the keypoint angle (GET_VALUE, :132-134)
    auto sample = [&](int pxi, int pyi) -> int {
        const float px = (float)pxi, py = (float)pyi;
        const float fr = __fadd_rn(__fmul_rn(px, b), __fmul_rn(py, a));
        const float fc = __fsub_rn(__fmul_rn(px, a), __fmul_rn(py, b));
        const int ir = __float2int_rn(fr), ic = __float2int_rn(fc);
        const uint16_t* h = H + (PR + ir - 3) * (HW + 1) + (PR + ic - 3);
        const int acc = 18 * (h[0] + h[6 * (HW + 1)]) + 34 * (h[HW + 1] + h[5 * (HW + 1)]) +
                        49 * (h[2 * (HW + 1)] + h[4 * (HW + 1)]) + 55 * h[3 * (HW + 1)];
        return min(255, (acc + 32768) >> 16);
    };
    const char4* pat4 = reinterpret_cast<const char4*>(pattern);
    const char4 q0 = pat4[lane], q1 = pat4[64 + lane], q2 = pat4[128 + lane], q3 = pat4[192 + lane];
    const unsigned long long w0 = __ballot(sample(q0.x, q0.y) < sample(q0.z, q0.w));
    const unsigned long long w1 = __ballot(sample(q1.x, q1.y) < sample(q1.z, q1.w));
    const unsigned long long w2 = __ballot(sample(q2.x, q2.y) < sample(q2.z, q2.w));
    const unsigned long long w3 = __ballot(sample(q3.x, q3.y) < sample(q3.z, q3.w));

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

// ------------------------------------------------------------------------------------------------
// launch wrappers (keep <<<>>> syntax inside this translation unit)
void orb_launch_copy_level0(hipStream_t st, const uint8_t* src, size_t rowStride, size_t frameStride,
                            uint8_t* pyr, size_t pyrSlab, int w, int h, int pitch, int nFrames)
{
    const int vec16 = ((reinterpret_cast<uintptr_t>(src) | rowStride | frameStride) & 15) == 0;
    const int x16 = (w + 15) / 16;
    dim3 grid((x16 + 255) / 256, h, nFrames);
    hipLaunchKernelGGL(k_copy_level0, grid, dim3(256), 0, st, src, rowStride, frameStride, pyr, pyrSlab, w, h, pitch, vec16);
}

void orb_launch_resize(hipStream_t st, uint8_t* pyr, size_t pyrSlab, const OrbLevelGeom& src,
                       const OrbLevelGeom& dst, const int2* xtab, const int2* ytab, int nFrames)
{
    const int x4 = (dst.w + 3) / 4;
    dim3 grid((x4 + 63) / 64, (dst.h + 3) / 4, nFrames);
    hipLaunchKernelGGL(k_resize_level, grid, dim3(64, 4), 0, st, pyr, pyrSlab, src.pyrOff, src.pitch,
                       dst.pyrOff, dst.pitch, dst.w, dst.h, xtab, ytab);
}

void orb_launch_orient_desc(hipStream_t st, const OrbGeom& G, const uint8_t* pyr, size_t pyrSlab,
                            const uint32_t* kpl, const int* kpCount, const int8_t* pattern,
                            orb_keypoint* kps, uint8_t* desc, int cap, int32_t* counts, int* errFlags,
                            int nFrames)
{
    hipLaunchKernelGGL(k_orient_desc, dim3(G.kpSlab, nFrames), dim3(WAVE), 0, st, G, pyr, pyrSlab, kpl, kpCount,
                       pattern, kps, desc, cap, counts, errFlags);
}
