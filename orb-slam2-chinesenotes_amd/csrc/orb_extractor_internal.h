// orb_extractor_internal.h -- the extractor handle (shared by orb_extractor.hip and orb_stereo.hip).
#pragma once
#include <vector>

#include "orb_kernels.h"

struct DevBuf {
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
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct orb_extractor {
    orb_extractor_params prm;
    int device = 0;
    hipStream_t stream = nullptr;
    static const int kProfSlots = 64;
    hipEvent_t ev[kProfSlots][5];           // ring of per-batch stage boundaries
    hipEvent_t waitEv = nullptr;
    int profFrames = 0;
    bool profiling = false;
    int profCount = 0;                      // batches recorded since profiling was (re)enabled

    // constructor tables (reference :503-558)
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> quota;
    int umax[16];

    // geometry for the current image size
    int rows = 0, cols = 0;
    OrbGeom G;
    std::vector<OrbStrip> strips;               // FAST work items (runs of cells of one cell row)
    int nCells = 0;
    size_t pyrSlab = 0, candSlab = 0;
    int sortCap = 4096, nodeCap = 0, maxKp = 0;
    int fastPdw = 20, fastRows = 66, fastSdw = 18, fastCandCap = 640;   // LDS sizing of k_fast_strips
    // cells per strip aimed at, per level.  Starts at 3 and is lowered for a level whose strips keep overflowing the
    // candidate queue (coarse levels have several times more corners per pixel); ORB_FAST_STRIP=k fixes it (tuning).
    int fastStripK[ORB_MAX_LEVELS];
    int fastStripsOfLevel[ORB_MAX_LEVELS];
    bool fastStripFixed = false;
    bool geomDirty = false;                 // strip lengths changed: rebuild the geometry on the next call

    // device memory
    DevBuf dPattern, dAngTab, dCells, dXtab, dYtab, dXq, dPath;   // constants
    std::vector<size_t> xtabOff, ytabOff;       // per level offsets (in int2 units)
    std::vector<long long> xqOff;               // per level offset into dXq (uint4 units), -1 = level not eligible
    DevBuf dPyr, dCand, dKpl, dOvf;             // per-batch scratch (dOvf: FAST strips to redo densely)
    // per-batch status words, ONE allocation so that one memset clears it and one copy fetches it:
    // [err: n][FAST candidates per level: 16n][keypoints per level: 16n][FAST overflow: list length, 7 pad, 16 per-level
    // counts] for the n frames of the current batch
    DevBuf dStat;
    int* errP() const { return (int*)dStat.p; }
    int* candCountP() const { return (int*)dStat.p + lastFrames; }
    int* kpCountP() const { return (int*)dStat.p + (size_t)(1 + ORB_MAX_LEVELS) * lastFrames; }
    int* ovfCountP() const { return (int*)dStat.p + (size_t)(1 + 2 * ORB_MAX_LEVELS) * lastFrames; }
    static const int kOvfInts = 8 + ORB_MAX_LEVELS;
    static size_t statInts(int n) { return (size_t)(1 + 2 * ORB_MAX_LEVELS) * n + kOvfInts; }
    DevBuf dImgs, dKps, dDesc, dCounts;         // staging for the host-buffer API
    DevBuf dStereo, dStereoIn;                  // stereo search: (SAD, index) pairs; host-API staging
    const int8_t* patternPtr = nullptr;         // device pointer in use (own copy or caller's)
    int framesCap = 0, lastFrames = 0;
    std::vector<int> hStat;                     // host copy of dStat (orb_extractor_sync)
    bool statFetched = false;
    void* hStage = nullptr;                     // pinned staging of the host-buffer API
    size_t hStageBytes = 0;
};

