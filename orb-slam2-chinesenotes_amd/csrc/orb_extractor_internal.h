// orb_extractor_internal.h -- the extractor handle (shared by orb_extractor.hip and orb_stereo.hip).
#pragma once
#include <vector>

#include "orb_kernels.h"
#include "orb_geometry_host.h"

#define ORB_QT_LDS_MAX (156 * 1024)  // k_quadtree may take (almost) the CU's whole 160 KB for huge per-level quotas
#define ORB_PIPE_CHUNK_MIN 8       // host batches of >= 2 chunks of this size are pipelined

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
    std::vector<hipStream_t> retiredStreams;   // streams replaced after a foreign capture invalidated them: destroyed with the handle
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
    int gaussTaps[4] = {18, 34, 49, 55};    // orb_extractor_set_gaussian
    OrbHostTables tables;                   // the same, as orb_geometry_host.h computes them

    // geometry for the current image size
    int rows = 0, cols = 0;
    int lastGeomRows = 0, lastGeomCols = 0;     // size of the last successful geometry build (survives a failed one)
    OrbGeom G;
    int pyrPersistent = 0;                      // launches of the last batch's pyramid that took the persistent form (k_pyr_chain_p)
    unsigned long long* pyrStamps = nullptr;    // diagnostics: phase stamps of the k_pyr_chain launches (orb_extractor_set_pyr_stamps)
    size_t pyrStampCap = 0;
    int pyrStampChains = 0, pyrStampBands[8] = {}, pyrStampSteps[8] = {};
    unsigned long long* descStamps = nullptr;   // diagnostics: phase stamps of k_desc_level (orb_extractor_set_desc_stamps)
    size_t descStampCap = 0;
    OrbDescPlan descPlan;                       // level-resident descriptor stage: regions of the upper levels (orb_desc_level_plan)
    std::vector<OrbStrip> strips;               // FAST work items (runs of cells of one cell row)
    int nCells = 0;
    size_t pyrSlab = 0, candSlab = 0;
    int sortCap = 4096, nodeCap = 0, maxKp = 0;
    int fastPdw = 20, fastRows = 66, fastSdw = 18, fastCandCap = 640;   // LDS sizing of k_fast_strips
    int fastP = 0;                          // compile-time pitch of k_fast_strips_p in use (0: the generic kernel)
    // cells per strip aimed at, per level.  Starts at 3 and is lowered for a level whose strips keep overflowing the
    // candidate queue (coarse levels have several times more corners per pixel); ORB_FAST_STRIP=k fixes it (tuning).
    int fastStripK[ORB_MAX_LEVELS];
    int fastStripsOfLevel[ORB_MAX_LEVELS];
    bool fastStripFixed = false;
    bool geomDirty = false;                 // strip lengths changed: rebuild the geometry on the next call
    // Strip-length feedback without a sync: a device-path batch leaves its overflow counters in pinned memory behind an
    // event; the next device-path call applies them if they have arrived (callers that never call orb_extractor_sync
    // between batches -- device pipelines -- would otherwise keep redoing overflowing strips densely for ever).
    bool specNoDense = false;               // single-frame host call: leave k_fast_strips_dense out, redo the frame if a strip overflowed
    int* ovfHost = nullptr;                 // pinned copy of the overflow block (kOvfInts ints)
    hipEvent_t ovfEv = nullptr;
    hipStream_t sideStream = nullptr;           // the level-resident descriptor kernel of a batch runs here, beside k_orient_desc on `stream`
    hipEvent_t sideFork = nullptr, sideJoin = nullptr;
    unsigned batchSerial = 0, ovfPendingSerial = 0, ovfAppliedSerial = 0;
    unsigned statSerial = 0;                // batch that h->hStat belongs to (0 = the latest one)
    bool hostCall = false;                  // inside orb_extract_batch: status travels with the results, no feedback copy

    // device memory
    DevBuf dPattern, dPatternF, dAngTab, dCells, dXtab, dYtab, dXq, dPath, dBand;   // constants
    std::vector<OrbPyrChain> pyrChains;         // empty: per-level pyramid kernels (k_copy_level0, k_resize_*)
    std::vector<OrbPyrChain> pyrChainsLat;      // the same levels in bands of 4 rows, for batches of a few frames
    std::vector<OrbPyrChain> pyrChainsOne;      // one or two frames: up to ORB_PYR_MAXCHAIN levels per launch
    std::vector<size_t> xtabOff, ytabOff;       // per level offsets (in int2 units)
    std::vector<long long> xqOff;               // per level offset into dXq (uint4 units), -1 = level not eligible
    DevBuf dPyr, dCand, dKpl, dOvf;             // per-batch scratch (dOvf: FAST strips to redo densely)
    DevBuf dQt;                                 // node lists of k_quadtree_gnodes (quotas beyond one workgroup's LDS)
    bool qtGlobal = false;
    // per-batch status words, ONE allocation so that one memset clears it and one copy fetches it:
    // [err: n][FAST candidates per level: 16n][keypoints per level: 16n][FAST overflow: list length, 7 pad, 16 per-level
    // counts] for the n frames of the current batch
    DevBuf dStat;
    // In front of the block: kStickyInts ints that no batch clears; the last of them (errP()[-1]) = OR of every error
    // flag since the last sync.
    static const int kStickyInts = 4;
    int* errP() const { return (int*)dStat.p + kStickyInts; }
    int* candCountP() const { return errP() + lastFrames; }
    int* kpCountP() const { return errP() + (size_t)(1 + ORB_MAX_LEVELS) * lastFrames; }
    int* ovfCountP() const { return errP() + (size_t)(1 + 2 * ORB_MAX_LEVELS) * lastFrames; }
    static const int kOvfInts = 8 + ORB_MAX_LEVELS;
    static size_t batchInts(int n) { return (size_t)(1 + 2 * ORB_MAX_LEVELS) * n + kOvfInts; }     // what a batch clears
    static size_t statInts(int n) { return kStickyInts + batchInts(n); }
    DevBuf dImgs, dKps, dDesc, dCounts;         // staging for the host-buffer API
    DevBuf dStereo, dStereoIn;                  // stereo search: (SAD, index) pairs; host-API staging
    const int8_t* patternPtr = nullptr;         // device pointer in use (own copy or caller's)
    int framesCap = 0, lastFrames = 0;
    std::vector<int> hStat;                     // host copy of dStat (orb_extractor_sync)
    bool statFetched = false;
    void* hStage = nullptr;                     // pinned staging of the host-buffer API
    size_t hStageBytes = 0;
    // single-frame host calls (the reference's operator() path): the launch chain + the copies back, captured once as a
    // HIP graph and replayed -- one launch instead of ~17 API calls.  Re-captured when anything it baked in changes.
    struct Graph {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        int rows = 0, cols = 0, cap = 0, sortCap = 0, geomVersion = -1;
        const void* pattern = nullptr;
        const void* stage = nullptr;
        // every device buffer the captured nodes address: a later, larger batch re-allocates them (ensure_scratch, the
        // staging of the host-buffer API) and a replay would run on freed memory
        const void* bufs[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        int seenSame = 0;                       // eager calls with the current key (capture after the first)
        bool broken = false;                    // a graph API call failed once: stay eager
    } graph1;
    int geomVersion = 0;                        // bumped by every geometry build
    void graph_bufs(const void* (&b)[10]) const
    {
        const void* cur[10] = {dPyr.p, dCand.p, dKpl.p, dOvf.p, dStat.p, dImgs.p, dKps.p, dDesc.p, dCounts.p, dQt.p ? dQt.p : dPatternF.p};
        for (int i = 0; i < 10; i++) b[i] = cur[i];
    }

    int frameBase = 0;                          // batch index of the device-resident frame 0 (pipelined host batches keep
                                                // only their last chunk on the device)

    // pipelined host batches (orb_host_pipe.hip): two slots of device in/out buffers and pinned staging, copy streams
    static const int kPipeSlots = 3;            // chunks in flight in the host pipeline (issue k, retire k - 2)
    struct Pipe {
        hipStream_t h2d = nullptr, d2h = nullptr;
        hipEvent_t evIn[kPipeSlots] = {}, evK[kPipeSlots] = {}, evOut[kPipeSlots] = {};
        DevBuf dImg[kPipeSlots], dKps[kPipeSlots], dDesc[kPipeSlots], dCnt[kPipeSlots];
        void* pinIn[kPipeSlots] = {};
        void* pinOut[kPipeSlots] = {};
        size_t pinInBytes = 0, pinOutBytes = 0;
        bool ready = false;
    } pipe;
};

int orb_check_status(orb_extractor* h);        // interprets h->hStat of the batch of h->lastFrames frames
int orb_extract_batch_pipelined(orb_extractor* h, const uint8_t* imgs, int nFrames, int rows, int cols, size_t rowStride,
                                size_t frameStride, orb_keypoint* kps, uint8_t* desc, int cap, int32_t* counts);
void orb_pipe_release(orb_extractor* h);

