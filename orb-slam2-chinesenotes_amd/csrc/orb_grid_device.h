// orb_grid_device.h -- the Frame feature grid on the device (reference src/Frame.cc:243-259 AssignFeaturesToGrid,
// :348-409 GetFeaturesInArea, :412-422 PosInGrid; 64x48 cells, include/Frame.h:37-38), shared by the
// SearchForInitialization and SearchByProjection kernels (each translation unit gets its own static copy).
//
// The grid is ONE sorted array of (cell << 16 | index) keys with cell = ix*48 + iy: the cells (ix, iyMin..iyMax) of
// a window query are a contiguous range per column ix, already in the reference's iteration order (ix outer, iy
// inner, insertion order inside a cell).
#pragma once
#include "orb_block_sort.h"
#include "orb_wave.h"
#include "orb_common.h"

#define GRID_COLS 64
#define GRID_ROWS 48

struct InitGrid { float minX, minY, invW, invH; };

static __device__ __forceinline__ void load_desc8(const uint8_t* p, uint32_t v[8])
{
    const uint4 lo = reinterpret_cast<const uint4*>(p)[0], hi = reinterpret_cast<const uint4*>(p)[1];
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
    v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
}

// keys[k] = cell<<16 | index for the level-0 keypoints of frame 2 that fall inside the grid
// (PosInGrid, :412-422), sorted ascending; *nKeys = how many.
static __global__ __launch_bounds__(256) void k_init_grid(const orb_keypoint* __restrict__ kps2, int n2, InitGrid g,
                                                          int levelZeroOnly, uint32_t* __restrict__ keys,
                                                          int* __restrict__ nKeys)
{
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        const orb_keypoint kp = kps2[i];
        if (levelZeroOnly && kp.octave != 0) continue;             // SearchForInitialization queries level 0 only, :1079
        const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, g.minX), g.invW));
        const int py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, g.minY), g.invH));
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
        const int slot = atomicAdd(&cnt, 1);
        keys[slot] = ((uint32_t)(px * GRID_ROWS + py) << 16) | (uint32_t)i;
    }
    __syncthreads();
    const int n = cnt;
    orb_block_sort(keys, n);                                       // global memory, one workgroup
    if (threadIdx.x == 0) *nKeys = n;
}

static __device__ __forceinline__ int lower_key(const uint32_t* keys, int n, uint32_t want)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    return lo;
}


