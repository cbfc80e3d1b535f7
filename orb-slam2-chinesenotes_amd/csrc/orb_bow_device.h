// orb_bow_device.h -- device helpers shared by the SearchByBoW kernels (orb_matcher.hip: one workgroup per pair;
// orb_matcher_query.hip: one query frame against many keyframes).
#pragma once
#include "orb_wave.h"
#include "orb_common.h"

#define WAVE 64
#define TH_LOW 50
#define HISTO_LENGTH 30

__device__ __forceinline__ int hamming8(const uint32_t* a, const uint32_t* b)
{
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d += __popc(a[i] ^ b[i]);
    return d;
}

// A load through a pointer that is known to point to global (HBM) memory.  The BowSide pointers reach the kernel inside a
// struct read from memory, so the compiler cannot tell their address space and would emit flat_load (which also takes an
// LDS-aperture check and counts on both wait counters); this makes it a global_load.
template <class T>
__device__ __forceinline__ T gload(const T* p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const __attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p));
#else
    return *p;                                         // (host pass of the single-source compile: never executed)
#endif
}

__device__ __forceinline__ void load_desc(const uint8_t* p, uint32_t v[8])
{
    const uint4 lo = gload(reinterpret_cast<const uint4*>(p)), hi = gload(reinterpret_cast<const uint4*>(p) + 1);
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
    v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
}

// rotation-histogram bin (reference :634-641): factor is 1/HISTO_LENGTH, so only bins 0..12 occur
__device__ __forceinline__ int rot_bin(float angA, float angB)
{
    float rot = __fsub_rn(angA, angB);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// ComputeThreeMaxima (:1663-1707) by one wave: the reference's sequential scan with strict > keeps, among equal counts,
// the lower bin first, i.e. it selects by (count descending, bin ascending) -- three wave maxima of (count << 8 | 63 - bin),
// empty bins (the scan never takes a count of 0) excluded.  (A serial 30-step loop by one thread, unrolled by the compiler
// under this kernel's 64-VGPR limit, spilled ~150 scratch accesses into the tail of every pair.)
__device__ __forceinline__ void three_maxima_wave(const int* hist, int* keep, int lane)
{
    const int cnt = lane < HISTO_LENGTH ? hist[lane] : 0;
    const unsigned key = cnt > 0 ? ((unsigned)cnt << 8) | (unsigned)(63 - lane) : 0u;
    const unsigned k1 = ~orb_wave_umin(~key);
    const unsigned key2 = key == k1 ? 0u : key;
    const unsigned k2 = ~orb_wave_umin(~key2);
    const unsigned key3 = key2 == k2 ? 0u : key2;
    const unsigned k3 = ~orb_wave_umin(~key3);
    int ind1 = k1 ? 63 - (int)(k1 & 0xFFu) : -1, ind2 = k2 ? 63 - (int)(k2 & 0xFFu) : -1, ind3 = k3 ? 63 - (int)(k3 & 0xFFu) : -1;
    const float max1 = (float)(k1 >> 8), max2 = (float)(k2 >> 8), max3 = (float)(k3 >> 8);
    if (max2 < __fmul_rn(0.1f, max1)) { ind2 = -1; ind3 = -1; }
    else if (max3 < __fmul_rn(0.1f, max1)) { ind3 = -1; }
    if (lane == 0) { keep[0] = ind1; keep[1] = ind2; keep[2] = ind3; }
}

