// orb_matcher_internal.h -- the matcher handle, shared by orb_matcher.hip and orb_matcher_init.hip.
#pragma once
#include "orb_common.h"

struct MBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return ORB_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        ORB_HIP_TRY(hipMalloc(&p, need));
        bytes = need;
        return ORB_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

struct orb_matcher {
    int device = 0;
    int cus = 256;                       // compute units of the device (grid sizing of the query-form matcher)
    size_t ldsMax = 160 * 1024;          // LDS a workgroup may take on this device (hipDeviceAttributeMaxSharedMemoryPerBlock; gfx950: 160 KB)
    hipStream_t stream = nullptr;
    MBuf sidesA, sidesB;                 // BowSide arrays of a batch
    MBuf stage[12];                      // host-API staging (SearchByBoW)
    MBuf init[12];                       // host-API staging + scratch (SearchForInitialization)
    MBuf out, nm;
    MBuf plan;                           // pair lists of orb_match_bow_query_device's large-frame fallback (orb_matcher_query.hip)
    hipEvent_t waitEv = nullptr;
    MBuf qctr;                           // ring of 8 x qctrStride group counters of k_match_bow_query (launch L uses slot L % 8 and clears slot (L + 4) % 8)
    size_t qctrStride = 0;
    unsigned qserial = 0;
    unsigned long long* stamps = nullptr;   // diagnostics: orb_matcher_set_stage_stamps
    size_t stampCap = 0;
};
