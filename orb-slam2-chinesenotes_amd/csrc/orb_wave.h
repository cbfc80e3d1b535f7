// orb_wave.h -- wave64 reductions and scans with DPP (row_shr 1,2,4,8 inside the four rows of 16 lanes, then
// row_bcast 15 / 31 across rows).  No LDS round trips: a __shfl-style ds_bpermute reduction costs an LDS access
// per step, these cost one vector instruction per step (tools/ubench/valu_rate.hip).
#pragma once
#include <hip/hip_runtime.h>

#define ORB_DPP_STEP_ADD(v, ctrl, rmask) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false)

// inclusive prefix sum over the wave (lane i gets v[0] + ... + v[i])
__device__ __forceinline__ int orb_wave_scan_incl(int v)
{
    ORB_DPP_STEP_ADD(v, 0x111, 0xf);
    ORB_DPP_STEP_ADD(v, 0x112, 0xf);
    ORB_DPP_STEP_ADD(v, 0x114, 0xf);
    ORB_DPP_STEP_ADD(v, 0x118, 0xf);
    ORB_DPP_STEP_ADD(v, 0x142, 0xa);
    ORB_DPP_STEP_ADD(v, 0x143, 0xc);
    return v;
}

// sum over the wave, returned in every lane (wave-uniform)
__device__ __forceinline__ int orb_wave_sum(int v) { return __builtin_amdgcn_readlane(orb_wave_scan_incl(v), 63); }

// bitwise OR over the wave, returned in every lane (wave-uniform)
__device__ __forceinline__ unsigned orb_wave_or(unsigned x)
{
    int v = (int)x;
#define ORB_DPP_STEP_OR(ctrl, rmask) v |= __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false)
    ORB_DPP_STEP_OR(0x111, 0xf);
    ORB_DPP_STEP_OR(0x112, 0xf);
    ORB_DPP_STEP_OR(0x114, 0xf);
    ORB_DPP_STEP_OR(0x118, 0xf);
    ORB_DPP_STEP_OR(0x142, 0xa);
    ORB_DPP_STEP_OR(0x143, 0xc);
#undef ORB_DPP_STEP_OR
    return (unsigned)__builtin_amdgcn_readlane(v, 63);
}

#define ORB_DPP_STEP_UMIN(v, ctrl, rmask) \
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(v), ctrl, rmask, 0xf, false))

// unsigned minimum over the wave, returned in every lane (wave-uniform)
__device__ __forceinline__ unsigned orb_wave_umin(unsigned v)
{
    ORB_DPP_STEP_UMIN(v, 0x111, 0xf);
    ORB_DPP_STEP_UMIN(v, 0x112, 0xf);
    ORB_DPP_STEP_UMIN(v, 0x114, 0xf);
    ORB_DPP_STEP_UMIN(v, 0x118, 0xf);
    ORB_DPP_STEP_UMIN(v, 0x142, 0xa);
    ORB_DPP_STEP_UMIN(v, 0x143, 0xc);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
