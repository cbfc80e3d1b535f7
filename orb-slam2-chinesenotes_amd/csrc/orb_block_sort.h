// orb_block_sort.h -- ascending bitonic sort of a[0..n) by one workgroup, arbitrary n (all merges ascending,
// virtual +inf padding).  Used by the quadtree (u64 keys), the matchers' CSR build and the frame grid (u32 keys).
//
// A block sort of ~1000 keys is latency-bound (one dependent LDS round trip + barrier per step), so the steps are
// arranged to need few round trips: a step is indexed by PAIR (no idle half of the threads) and a thread loads all
// its operands before it stores any; the steps with partner distance 2 and 1 (and the whole k = 2, 4 stages) run
// in registers on 4 consecutive elements per thread: 45 barrier steps instead of 55 for 1024 keys, each about
// half as long.  Force-inlined so that the address space of `a` (LDS or global) is known at the call site.
#pragma once
#include <hip/hip_runtime.h>

#define ORB_SORT_CX(x, y) { if (x > y) { const T t_ = x; x = y; y = t_; } }

// steps on groups of 4 consecutive elements: first = true runs stages k = 2 and k = 4, else the j = 2, 1 tail
template <class T>
__device__ __forceinline__ void orb_sort_local4(T* a, int n, int np2, bool first)
{
    for (int g = threadIdx.x * 4; g < np2; g += blockDim.x * 4) {
        if (g < n) {
            const T inf = (T)~(T)0;
            T v0 = a[g], v1 = g + 1 < n ? a[g + 1] : inf, v2 = g + 2 < n ? a[g + 2] : inf, v3 = g + 3 < n ? a[g + 3] : inf;
            if (first) {
                ORB_SORT_CX(v0, v1) ORB_SORT_CX(v2, v3)        // k = 2
                ORB_SORT_CX(v0, v3) ORB_SORT_CX(v1, v2)        // k = 4 flip
                ORB_SORT_CX(v0, v1) ORB_SORT_CX(v2, v3)        // k = 4, j = 1
            } else {
                ORB_SORT_CX(v0, v2) ORB_SORT_CX(v1, v3)        // j = 2
                ORB_SORT_CX(v0, v1) ORB_SORT_CX(v2, v3)        // j = 1
            }
            a[g] = v0;                                         // +inf never moves below a real key
            if (g + 1 < n) a[g + 1] = v1;
            if (g + 2 < n) a[g + 2] = v2;
            if (g + 3 < n) a[g + 3] = v3;
        }
    }
    __syncthreads();
}

// one compare-exchange step over all pairs; flip: partner = i ^ (2d - 1) (d = k/2), else partner = i | d
template <class T>
__device__ __forceinline__ void orb_sort_step(T* a, int n, int half, int d, bool flip)
{
    const int S = blockDim.x;
    for (int t = threadIdx.x; t < half; t += 2 * S) {
        const int t1 = t + S;
        const int i0 = ((t & ~(d - 1)) << 1) | (t & (d - 1)), i1 = ((t1 & ~(d - 1)) << 1) | (t1 & (d - 1));
        const int p0 = flip ? i0 ^ (2 * d - 1) : i0 | d, p1 = flip ? i1 ^ (2 * d - 1) : i1 | d;
        const bool ok0 = p0 < n, ok1 = t1 < half && p1 < n;
        T x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        if (ok0) { x0 = a[i0]; y0 = a[p0]; }
        if (ok1) { x1 = a[i1]; y1 = a[p1]; }
        if (ok0 && x0 > y0) { a[i0] = y0; a[p0] = x0; }
        if (ok1 && x1 > y1) { a[i1] = y1; a[p1] = x1; }
    }
    __syncthreads();
}

// call with all threads of the block; a[0..n) must be visible (barrier before), result visible after return
template <class T>
__device__ __forceinline__ void orb_block_sort(T* a, int n)
{
    int np2 = 4;
    while (np2 < n) np2 <<= 1;
    const int half = np2 >> 1;
    orb_sort_local4(a, n, np2, true);
    for (int k = 8; k <= np2; k <<= 1) {
        orb_sort_step(a, n, half, k >> 1, true);
        for (int j = k >> 2; j >= 4; j >>= 1) orb_sort_step(a, n, half, j, false);
        orb_sort_local4(a, n, np2, false);
    }
}
