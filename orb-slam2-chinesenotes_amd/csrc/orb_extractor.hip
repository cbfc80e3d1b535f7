// orb_extractor.hip -- host side of the extractor half of include/orb_hip.h:
// handle life cycle, the constructor tables of reference src/ORBextractor.cc:498-559, level / cell
// geometry (:805-849), resize coefficient tables (cv::resize set-up), scratch slabs in HBM and
// the launch sequence of one batch.  All device work of a handle goes to its own HIP stream.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstddef>
#include <chrono>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/orb_brief_pattern.h"
#include "orb_extractor_internal.h"
#include "orb_geometry_host.h"

// ORB_TIMING=1: where a single-frame host call spends its time (image copy-in, launch, wait, copy-out), printed by destroy
static const bool g_timing = std::getenv("ORB_TIMING") != nullptr;
static double g_tAcc[4];
static long g_tCalls;
static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#pragma clang fp contract(off)

// ------------------------------------------------------------------ error string
static thread_local char g_err[512] = "";
void orb_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* orb_last_error(void) { return g_err; }
extern "C" const char* orb_version(void) { return "orbhip 0.3 (gfx950)"; }
extern "C" int orb_abi_version(void) { return ORB_HIP_ABI_VERSION; }
extern "C" size_t orb_sizeof_featstore(void) { return sizeof(orb_featstore); }

// level sizes, FAST strips, quadtree boxes, slab layout for a rows x cols input: planned on the host
// (orb_geometry_host.h), LDS budgets checked, constants uploaded, then committed to the handle -- a failed call leaves
// the handle without a geometry (rows = cols = 0), never with a half-built one.
static int build_geometry(orb_extractor* h, int rows, int cols)
{
    h->rows = h->cols = 0;
    OrbGeomPlan P;
    // strips cut by the self-tuning default (not by ORB_FAST_STRIP) stay within the 112-byte rows of k_fast_strips_p<28>
    int rc = orb_plan_geometry(h->prm, h->tables, h->fastStripK, rows, cols, P, h->fastStripFixed ? 0 : 112);
    if (rc != ORB_OK) return rc;
    const OrbGeom& G = P.G;
    std::vector<OrbStrip>& strips = P.strips;
    std::vector<uint32_t>& pathTab = P.pathTab;
    const int nodeCap = P.nodeCap, maxPdw = P.fastPdw, maxRows = P.fastRows, maxSdw = P.fastSdw;
    // LDS sort capacity per (frame, level) instance; larger candidate sets are sorted in global memory.
    // It starts small (more resident workgroups: the kernel is latency-bound) and orb_extractor_sync() grows it to
    // the largest candidate count actually seen, so steady-state batches sort in LDS.
    // A rebuild for shorter FAST strips (same image size) keeps what the feedback has already grown it to.
    int sortCap = (rows == h->lastGeomRows && cols == h->lastGeomCols) ? std::max(h->sortCap, 1024) : 1024;
    // per-workgroup LDS budget: 60 KB (several workgroups per CU) unless the node arrays alone need more
    const size_t qtBudget = orb_quadtree_lds_bytes(256, nodeCap) > 60 * 1024 ? (size_t)ORB_QT_LDS_MAX : (size_t)60 * 1024;
    while (sortCap > 256 && orb_quadtree_lds_bytes(sortCap, nodeCap) > qtBudget) sortCap -= 256;
    // quotas whose node lists do not fit one workgroup's LDS (nFeatures >~ 12 000): the lists go to a global scratch slab
    // (k_quadtree_gnodes: same results, every step a round trip to L2)
    const bool qtGlobal = orb_quadtree_lds_bytes(sortCap, nodeCap) > ORB_QT_LDS_MAX;
    if (qtGlobal) sortCap = 4096;
    if (orb_fast_lds_bytes(maxPdw, maxRows, 4096) > 64 * 1024 || orb_fast_dense_lds_bytes(maxPdw, maxRows, maxSdw) > 64 * 1024) {
        orb_set_error("FAST strip tile needs more than 64 KB of LDS");
        return ORB_ERR_UNSUPPORTED;
    }

    // upload constants for this geometry
    if (!strips.empty()) {
        if ((rc = h->dCells.ensure(strips.size() * sizeof(OrbStrip))) != ORB_OK) return rc;
        ORB_HIP_TRY(hipMemcpyAsync(h->dCells.p, strips.data(), strips.size() * sizeof(OrbStrip), hipMemcpyHostToDevice,
                                   h->stream));
    }
    if (!pathTab.empty()) {
        if ((rc = h->dPath.ensure(pathTab.size() * 4)) != ORB_OK) return rc;
        ORB_HIP_TRY(hipMemcpyAsync(h->dPath.p, pathTab.data(), pathTab.size() * 4, hipMemcpyHostToDevice, h->stream));
    }
    std::vector<int2>& xt = P.xt;
    std::vector<int2>& yt = P.yt;
    std::vector<uint32_t>& xq = P.xq;
    h->xtabOff = P.xtabOff; h->ytabOff = P.ytabOff; h->xqOff = P.xqOff;
    if (!xt.empty()) {
        if ((rc = h->dXtab.ensure(xt.size() * sizeof(int2))) != ORB_OK) return rc;
        if ((rc = h->dYtab.ensure(yt.size() * sizeof(int2))) != ORB_OK) return rc;
        ORB_HIP_TRY(hipMemcpyAsync(h->dXtab.p, xt.data(), xt.size() * sizeof(int2), hipMemcpyHostToDevice, h->stream));
        ORB_HIP_TRY(hipMemcpyAsync(h->dYtab.p, yt.data(), yt.size() * sizeof(int2), hipMemcpyHostToDevice, h->stream));
        if (!xq.empty()) {
            if ((rc = h->dXq.ensure(xq.size() * 4)) != ORB_OK) return rc;
            ORB_HIP_TRY(hipMemcpyAsync(h->dXq.p, xq.data(), xq.size() * 4, hipMemcpyHostToDevice, h->stream));
        }
    }
    if (!P.bandTab.empty() && !std::getenv("ORB_PYR_LEGACY")) {
        if ((rc = h->dBand.ensure(P.bandTab.size() * sizeof(int2))) != ORB_OK) return rc;
        ORB_HIP_TRY(hipMemcpyAsync(h->dBand.p, P.bandTab.data(), P.bandTab.size() * sizeof(int2), hipMemcpyHostToDevice, h->stream));
        h->pyrChains = P.chains;
        h->pyrChainsLat = P.chainsLat;
        h->pyrChainsOne = P.chainsOne;
    } else {
        h->pyrChains.clear();
        h->pyrChainsLat.clear();
        h->pyrChainsOne.clear();
    }
    ORB_HIP_TRY(hipStreamSynchronize(h->stream));     // host vectors go out of scope
    // commit
    h->G = G;
    orb_desc_level_plan(h->G, &h->descPlan);                  // which levels the level-resident descriptor kernel takes
    h->strips.swap(strips);
    h->nCells = P.nCells;
    std::memcpy(h->fastStripsOfLevel, P.stripsOfLevel, sizeof(P.stripsOfLevel));
    h->pyrSlab = P.pyrSlab;
    h->candSlab = P.candSlab;
    h->nodeCap = nodeCap;
    h->qtGlobal = qtGlobal;
    h->maxKp = P.maxKp;
    h->fastPdw = maxPdw;
    h->fastRows = maxRows;
    h->fastSdw = maxSdw;
    // the fixed-pitch detector (k_fast_strips_p<P>) needs rows within its pitch
    h->fastP = 0;
    if (!std::getenv("ORB_FAST_GENERIC")) h->fastP = maxPdw <= 20 ? 20 : maxPdw <= 28 ? 28 : 0;
    // candidate queue of a strip (pixels whose score exceeds the lower threshold); a strip with more switches to
    // the dense scan.  At most 64 queue steps (one bit per step in the kernel).
    h->fastCandCap = 640;
    if (const char* e = std::getenv("ORB_FAST_CANDCAP")) h->fastCandCap = std::max(64, std::min(4096, std::atoi(e)));
    h->sortCap = sortCap;
    h->framesCap = 0;                                  // slabs changed size: re-allocate lazily
    h->rows = rows;
    h->cols = cols;
    h->lastGeomRows = rows;
    h->lastGeomCols = cols;
    h->geomDirty = false;
    h->geomVersion++;
    return ORB_OK;
}

static int ensure_scratch(orb_extractor* h, int nFrames)
{
    if (nFrames <= h->framesCap) return ORB_OK;
    int rc;
    if ((rc = h->dPyr.ensure(h->pyrSlab * nFrames + 256)) != ORB_OK) return rc;   // + slack: window loads overrun a row by <= 11 B
    if ((rc = h->dCand.ensure(h->candSlab * 8 * nFrames)) != ORB_OK) return rc;
    if ((rc = h->dKpl.ensure((size_t)h->G.kpSlab * 4 * nFrames)) != ORB_OK) return rc;
    if ((rc = h->dOvf.ensure((size_t)4 * std::max<size_t>(1, h->strips.size()) * nFrames)) != ORB_OK) return rc;
    if (h->qtGlobal && (rc = h->dQt.ensure(orb_quadtree_scratch_stride(h->nodeCap) * h->G.nlevels * nFrames)) != ORB_OK) return rc;
    if (orb_extractor::statInts(nFrames) * 4 > h->dStat.bytes) {
        // a larger status block: the sticky words (error flags of batches that were never synchronised) move with it
        DevBuf nb;
        if ((rc = nb.ensure(orb_extractor::statInts(nFrames) * 4)) != ORB_OK) return rc;
        if (h->dStat.p) ORB_HIP_TRY(hipMemcpyAsync(nb.p, h->dStat.p, orb_extractor::kStickyInts * 4, hipMemcpyDeviceToDevice, h->stream));
        else ORB_HIP_TRY(hipMemsetAsync(nb.p, 0, orb_extractor::kStickyInts * 4, h->stream));
        ORB_HIP_TRY(hipStreamSynchronize(h->stream));          // earlier batches still write the old block; rare (growth only)
        h->dStat.release();
        h->dStat = nb;
    }
    h->framesCap = nFrames;
    return ORB_OK;
}

// ------------------------------------------------------------------ C ABI
extern "C" int orb_builtin_pattern(int8_t* out)
{
    if (!out) return ORB_ERR_INVALID;
    std::memcpy(out, ORB_BRIEF_PATTERN_XY, 1024);
    return ORB_OK;
}

extern "C" int orb_extractor_create(const orb_extractor_params* p, int device_id, orb_extractor** out)
{
    if (!p || !out) return ORB_ERR_INVALID;
    *out = nullptr;
    if (p->nlevels < 1 || p->nlevels > ORB_MAX_LEVELS || p->nfeatures < 0 || !(p->scale_factor > 1.0f) ||
        p->ini_th_fast < 0 || p->min_th_fast < 0 || p->ini_th_fast > 255 || p->min_th_fast > 255) {
        orb_set_error("invalid extractor parameters");
        return ORB_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        orb_set_error("no HIP device: liborbhip has no CPU fallback");
        return ORB_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= ndev) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(device_id));
    orb_extractor* h = new (std::nothrow) orb_extractor();
    if (!h) return ORB_ERR_INTERNAL;
    h->prm = *p;
    h->device = device_id;
    orb_build_tables(h->prm, h->tables);
    h->scale = h->tables.scale; h->invScale = h->tables.invScale; h->sigma2 = h->tables.sigma2;
    h->invSigma2 = h->tables.invSigma2; h->quota = h->tables.quota;
    std::memcpy(h->umax, h->tables.umax, sizeof(h->umax));
    {
        // cells per FAST strip: 3, and 2 on the coarse levels (scale >= 2.4: several times more corners per pixel, so
        // three cells overflow the candidate queue now and then -- few cells live there, the shorter strips cost ~1 %)
        int k = 0;
        if (const char* e = std::getenv("ORB_FAST_STRIP")) { k = std::max(1, std::min(8, std::atoi(e))); h->fastStripFixed = true; }
        for (int l = 0; l < ORB_MAX_LEVELS; l++) h->fastStripK[l] = k ? k : ((l < h->prm.nlevels && h->scale[l] >= 2.4f) ? 2 : 3);
    }
    hipError_t e = orb_stream_create(&h->stream, device_id, 0);
    if (e != hipSuccess) { delete h; orb_set_error("hipStreamCreate: %s", hipGetErrorString(e)); return ORB_ERR_HIP; }
    for (int k = 0; k < orb_extractor::kProfSlots; k++)
        for (int i = 0; i < 5; i++) (void)hipEventCreate(&h->ev[k][i]);
    (void)hipEventCreateWithFlags(&h->waitEv, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&h->ovfEv, hipEventDisableTiming);
    if (hipHostMalloc((void**)&h->ovfHost, orb_extractor::kOvfInts * 4, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h->ovfHost = nullptr; }
    int rc = h->dPattern.ensure(1024);
    if (rc == ORB_OK) {
        if (orb_copy_blocking(h->dPattern.p, ORB_BRIEF_PATTERN_XY, 1024, hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = ORB_ERR_HIP;
    }
    if (rc == ORB_OK) rc = h->dPatternF.ensure(256 * 16);
    if (rc == ORB_OK) rc = h->dAngTab.ensure(16 * 2 * 32 + 768 * 4);     // + the descriptor kernel's horizontal-blur item table
    if (rc == ORB_OK) {
        // IC_Angle tables (k_orient_desc): per (|v|, half row) 16 mask bytes (1 inside |u| <= umax[|v|]) and
        // 16 weight bytes (u + 15 inside, 0 outside) for u = -15 + 16*half + byte
        uint8_t tab[16][2][32];
        for (int a = 0; a < 16; a++)
            for (int hh = 0; hh < 2; hh++)
                for (int b = 0; b < 16; b++) {
                    const int u = -15 + 16 * hh + b;
                    const bool in = std::abs(u) <= h->umax[a];
                    tab[a][hh][b] = in ? 1 : 0;
                    tab[a][hh][16 + b] = in ? (uint8_t)(u + 15) : 0;
                }
        if (orb_copy_blocking(h->dAngTab.p, tab, sizeof(tab), hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = ORB_ERR_HIP;
        uint32_t hb[768];
        orb_desc_hblur_table(hb);
        if (rc == ORB_OK && orb_copy_blocking((uint8_t*)h->dAngTab.p + sizeof(tab), hb, sizeof(hb), hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = ORB_ERR_HIP;
    }
    if (rc != ORB_OK) { orb_set_error("extractor tables: upload failed: %s", hipGetErrorString(hipGetLastError())); orb_extractor_destroy(h); return rc; }
    h->patternPtr = (const int8_t*)h->dPattern.p;
    *out = h;
    return ORB_OK;
}

extern "C" void orb_extractor_destroy(orb_extractor* h)
{
    if (!h) return;
    if (g_timing && g_tCalls) {
        fprintf(stderr, "orbhip timing over %ld single-frame calls (us): copy-in %.1f  launch %.1f  wait %.1f  copy-out %.1f\n", g_tCalls,
                g_tAcc[0] / g_tCalls, g_tAcc[1] / g_tCalls, g_tAcc[2] / g_tCalls, g_tAcc[3] / g_tCalls);
        g_tCalls = 0; g_tAcc[0] = g_tAcc[1] = g_tAcc[2] = g_tAcc[3] = 0;
    }
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    DevBuf* bufs[] = {&h->dPattern, &h->dPatternF, &h->dAngTab, &h->dCells, &h->dXtab, &h->dYtab, &h->dXq, &h->dPath, &h->dBand, &h->dPyr, &h->dCand, &h->dKpl, &h->dOvf, &h->dQt,
                      &h->dStat, &h->dImgs, &h->dKps, &h->dDesc, &h->dCounts,
                      &h->dStereo, &h->dStereoIn};
    for (DevBuf* b : bufs) b->release();
    for (int k = 0; k < orb_extractor::kProfSlots; k++)
        for (int i = 0; i < 5; i++)
            if (h->ev[k][i]) (void)hipEventDestroy(h->ev[k][i]);
    if (h->waitEv) (void)hipEventDestroy(h->waitEv);
    if (h->ovfEv) (void)hipEventDestroy(h->ovfEv);
    if (h->sideFork) (void)hipEventDestroy(h->sideFork);
    if (h->sideJoin) (void)hipEventDestroy(h->sideJoin);
    if (h->sideStream) orb_stream_destroy(h->sideStream, h->device);
    if (h->ovfHost) (void)hipHostFree(h->ovfHost);
    orb_pipe_release(h);
    if (h->graph1.exec) (void)hipGraphExecDestroy(h->graph1.exec);
    if (h->graph1.graph) (void)hipGraphDestroy(h->graph1.graph);
    if (h->hStage) (void)hipHostFree(h->hStage);
    if (h->stream) orb_stream_destroy(h->stream, h->device);
    for (hipStream_t st : h->retiredStreams) orb_stream_destroy(st, h->device);
    delete h;
}

extern "C" int orb_extractor_get_tables(const orb_extractor* h, float* scale, float* inv, float* s2, float* is2,
                                        int32_t* fpl)
{
    if (!h) return ORB_ERR_INVALID;
    for (int i = 0; i < h->prm.nlevels; i++) {
        if (scale) scale[i] = h->scale[i];
        if (inv) inv[i] = h->invScale[i];
        if (s2) s2[i] = h->sigma2[i];
        if (is2) is2[i] = h->invSigma2[i];
        if (fpl) fpl[i] = h->quota[i];
    }
    return ORB_OK;
}

extern "C" int orb_extractor_max_keypoints(const orb_extractor* h)
{
    if (!h) return ORB_ERR_INVALID;
    int tot = 0;
    for (int l = 0; l < h->prm.nlevels; l++) tot += std::max(h->quota[l] + 3, 4 * 15) + 5;
    return tot;
}

// the reference's table stays within +-13 (src/ORBextractor.cc:175-432); the descriptor kernel's patch and its row-blur item
// table are sized for that extent (csrc/orb_desc.hip: a sample lies within sqrt(13^2 + 13^2) px of the keypoint)
static int check_pattern(const int8_t* pat)
{
    for (int i = 0; i < 1024; i++)
        if (pat[i] < -13 || pat[i] > 13) {
            orb_set_error("BRIEF pattern coordinate %d outside [-13, 13]: the descriptor kernel's patch covers 18 px around a keypoint", (int)pat[i]);
            return ORB_ERR_UNSUPPORTED;
        }
    return ORB_OK;
}

extern "C" int orb_extractor_set_pattern(orb_extractor* h, const int8_t* pat)
{
    if (!h || !pat) return ORB_ERR_INVALID;
    int rc = check_pattern(pat);
    if (rc != ORB_OK) return rc;
    ORB_HIP_TRY(hipSetDevice(h->device));
    ORB_HIP_TRY(hipMemcpyAsync(h->dPattern.p, pat, 1024, hipMemcpyHostToDevice, h->stream));
    ORB_HIP_TRY(hipStreamSynchronize(h->stream));
    h->patternPtr = (const int8_t*)h->dPattern.p;
    return ORB_OK;
}

// The table that arrives over RCCL (csrc/orb_multi.hip) or from a caller's device buffer: it is copied to the host first and
// held to the same extent rule as the host setter -- a coordinate beyond +-13 would let the descriptor kernel index LDS outside
// the row-blurred patch and return wrong descriptors without any error.  The handle keeps its previous pattern when
// the table is rejected.
extern "C" int orb_extractor_set_pattern_device(orb_extractor* h, const int8_t* dpat)
{
    if (!h || !dpat) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(h->device));
    int8_t host[1024];
    ORB_HIP_TRY(hipMemcpyAsync(host, dpat, 1024, hipMemcpyDeviceToHost, h->stream));
    ORB_HIP_TRY(hipStreamSynchronize(h->stream));
    int rc = check_pattern(host);
    if (rc != ORB_OK) return rc;
    ORB_HIP_TRY(hipMemcpyAsync(h->dPattern.p, dpat, 1024, hipMemcpyDeviceToDevice, h->stream));
    ORB_HIP_TRY(hipStreamSynchronize(h->stream));
    h->patternPtr = (const int8_t*)h->dPattern.p;
    return ORB_OK;
}

extern "C" int orb_gaussian_preset(int preset, int32_t* taps4)
{
    if (!taps4) return ORB_ERR_INVALID;
    static const int32_t legacy[4] = {18, 34, 49, 55}, ed[4] = {18, 34, 48, 56};
    if (preset != ORB_GAUSS_OPENCV_LEGACY && preset != ORB_GAUSS_OPENCV_FIXEDPOINT_ED) return ORB_ERR_INVALID;
    std::memcpy(taps4, preset == ORB_GAUSS_OPENCV_LEGACY ? legacy : ed, sizeof(legacy));
    return ORB_OK;
}

extern "C" int orb_extractor_set_gaussian(orb_extractor* h, const int32_t* taps4)
{
    if (!h || !taps4) return ORB_ERR_INVALID;
    const int sum = 2 * (taps4[0] + taps4[1] + taps4[2]) + taps4[3];
    for (int i = 0; i < 4; i++)
        if (taps4[i] < 0 || taps4[i] > 255) { orb_set_error("Gaussian taps must be 8.8 fixed-point values in 0..255"); return ORB_ERR_UNSUPPORTED; }
    // the row pass keeps its sums in 16 bits (255 * 257 = 65535 is the largest that fits: OpenCV's own 8.8 kernels sum to 256 or 257)
    if (sum < 1 || sum > 257) { orb_set_error("Gaussian taps sum to %d: the row pass holds 255 * sum in 16 bits (sum <= 257)", sum); return ORB_ERR_UNSUPPORTED; }
    ORB_HIP_TRY(hipSetDevice(h->device));
    ORB_HIP_TRY(hipStreamSynchronize(h->stream));
    for (int i = 0; i < 4; i++) h->gaussTaps[i] = taps4[i];
    h->geomVersion++;                                  // (a captured single-frame graph baked the old taps in: re-capture)
    return ORB_OK;
}

extern "C" int orb_extractor_set_profiling(orb_extractor* h, int enable)
{
    if (!h) return ORB_ERR_INVALID;
    h->profiling = enable != 0;
    h->profCount = 0;
    return ORB_OK;
}

// average per-batch stage times over the (at most kProfSlots most recent) profiled batches
extern "C" int orb_extractor_get_stage_ms(orb_extractor* h, float* ms5)
{
    if (!h || !ms5) return ORB_ERR_INVALID;
    if (h->profCount == 0) { orb_set_error("no profiled batch"); return ORB_ERR_INVALID; }
    ORB_HIP_TRY(hipSetDevice(h->device));
    ORB_HIP_TRY(hipStreamSynchronize(h->stream));
    const int n = std::min(h->profCount, (int)orb_extractor::kProfSlots);
    double acc[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < n; k++) {
        const int slot = (h->profCount - 1 - k) % orb_extractor::kProfSlots;
        float t;
        for (int i = 0; i < 4; i++) {
            ORB_HIP_TRY(hipEventElapsedTime(&t, h->ev[slot][i], h->ev[slot][i + 1]));
            acc[i] += t;
        }
        ORB_HIP_TRY(hipEventElapsedTime(&t, h->ev[slot][0], h->ev[slot][4]));
        acc[4] += t;
    }
    for (int i = 0; i < 5; i++) ms5[i] = (float)(acc[i] / n);
    return ORB_OK;
}

extern "C" int orb_extractor_profiled_frames(const orb_extractor* h) { return h ? h->profFrames : ORB_ERR_INVALID; }

// make this handle's stream wait for everything already enqueued on `other_stream` (hipStream_t)
extern "C" int orb_extractor_wait_for(orb_extractor* h, void* other_stream)
{
    if (!h) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(h->device));
    ORB_HIP_TRY(hipEventRecord(h->waitEv, (hipStream_t)other_stream));
    ORB_HIP_TRY(hipStreamWaitEvent(h->stream, h->waitEv, 0));
    return ORB_OK;
}

extern "C" void* orb_extractor_stream(orb_extractor* h) { return h ? (void*)h->stream : nullptr; }

// the quadtree's LDS sort capacity follows the largest candidate count seen (within the LDS budget)
static void grow_sort_cap(orb_extractor* h, int maxCandidates)
{
    if (h->qtGlobal) return;                           // (the keys alone: 4096 of them in LDS, fixed)
    // the largest count seen + 1/8, in steps of 256 keys (2 KB) -- NOT the next power of two: the sorts take any n, and a level-0
    // count of 1086 (drawn content) / 2115 (natural statistics) used to claim 2048 / 4096 key slots, i.e. 31 / 47 KB of LDS per
    // workgroup = 5 / 3 workgroups of this latency-bound kernel per CU where 23 / 34 KB (6 / 4) do (round 5, tools/qt_stamps.py)
    int want = std::max(1024, std::min(4096, (maxCandidates + maxCandidates / 8 + 255) & ~255));
    const size_t budget = orb_quadtree_lds_bytes(256, h->nodeCap) > 60 * 1024 ? (size_t)ORB_QT_LDS_MAX : (size_t)60 * 1024;
    while (want > 256 && orb_quadtree_lds_bytes(want, h->nodeCap) > budget) want -= 256;
    if (want > h->sortCap) h->sortCap = want;
}

// levels whose strips overflowed their candidate queue get shorter strips (results never depend on the strip length).
// `serial` = the batch the counters belong to: counters of batches launched before an adjustment took effect are ignored.
static void apply_fast_overflows(orb_extractor* h, const int* perLevel, unsigned serial)
{
    if (h->fastStripFixed || (int)(serial - h->ovfAppliedSerial) <= 0) return;
    h->ovfAppliedSerial = serial;
    bool changed = false;
    for (int l = 0; l < h->prm.nlevels; l++)
        if (h->fastStripK[l] > 1 && perLevel[l] > 0) {   // one redone strip already costs the batch ~35 us of serial latency
            h->fastStripK[l]--;
            changed = true;
        }
    if (changed) {
        h->geomDirty = true;                           // the next call rebuilds the geometry
        h->ovfAppliedSerial = h->batchSerial;          // batches already launched still ran with the longer strips
    }
}

extern "C" int orb_extract_batch_device(orb_extractor* h, const uint8_t* d_imgs, int nFrames, int rows, int cols,
                                        size_t rowStride, size_t frameStride, orb_keypoint* d_kps,
                                        uint8_t* d_desc, int cap, int32_t* d_counts)
{
    if (!h || nFrames < 0 || cap < 0 || !d_counts) return ORB_ERR_INVALID;
    if (nFrames == 0) return ORB_OK;
    if (nFrames > 65535) { orb_set_error("at most 65535 frames per batch"); return ORB_ERR_UNSUPPORTED; }
    ORB_HIP_TRY(hipSetDevice(h->device));
    if (!d_imgs || rows <= 0 || cols <= 0) {                   // reference :1087-1088: silent return
        ORB_HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)nFrames * 4, h->stream));
        h->lastFrames = 0;
        return ORB_OK;
    }
    if (!d_kps || !d_desc || rowStride < (size_t)cols) return ORB_ERR_INVALID;
    int rc;
    if (h->ovfPendingSerial && !h->hostCall) {           // overflow counters of an earlier batch, if they have arrived
        if (hipEventQuery(h->ovfEv) == hipSuccess) {
            apply_fast_overflows(h, h->ovfHost + 8, h->ovfPendingSerial);
            grow_sort_cap(h, h->ovfHost[1]);                 // (word 1: largest candidate set that did not fit the LDS sort)
            h->ovfPendingSerial = 0;
        } else {
            (void)hipGetLastError();                     // hipErrorNotReady
        }
    }
    if (rows != h->rows || cols != h->cols || h->geomDirty)
        if ((rc = build_geometry(h, rows, cols)) != ORB_OK) return rc;
    if ((rc = ensure_scratch(h, nFrames)) != ORB_OK) return rc;
    const OrbGeom& G = h->G;
    hipStream_t st = h->stream;
    uint8_t* pyr = (uint8_t*)h->dPyr.p;

    // (Cutting a large batch into sub-batches on several streams was measured: no gain, every kernel already fills
    // the chip -- one stream, one launch chain.)
    h->lastFrames = nFrames;                                   // fixes the layout of the status block
    h->frameBase = 0;
    h->statFetched = false;
    const int n = nFrames;
    int* scc = h->candCountP();
    int* skc = h->kpCountP();
    int* serr = h->errP();
    unsigned long long* scand = (unsigned long long*)h->dCand.p;
    uint32_t* skpl = (uint32_t*)h->dKpl.p;
    const bool prof = h->profiling;
    hipEvent_t* pe = h->ev[h->profCount % orb_extractor::kProfSlots];
    if (prof) ORB_HIP_TRY(hipEventRecord(pe[0], st));
    // (also clears the status block behind the sticky word -- error flags, counters, overflow list head -- and spreads the
    // int8 BRIEF pattern into the float table the descriptor kernel reads)
    if (!h->pyrChains.empty()) {
        // small batches cannot fill the chip with 16-row bands (21 workgroups per frame at 640x480, 5 resident per CU):
        // below ~10 workgroups per CU the 4-row bands (4x the workgroups, each a quarter as long) finish sooner
        // up to 32 frames: fewer launches of longer chains (column tables in LDS where they fit)
        static const int oneMax = std::getenv("ORB_PYR_ONE_MAX") ? std::atoi(std::getenv("ORB_PYR_ONE_MAX")) : 32;
        // ORB_PYR_SET=batch|few|one pins the variant (tests: every variant on the same frames)
        const char* pin = std::getenv("ORB_PYR_SET");
        const int which = pin ? (pin[0] == 'b' ? 0 : pin[0] == 'f' ? 1 : 2)
                          : n <= oneMax ? 2 : 0;     // (round 4, on the round-3 chains: the 4-row-band set no longer wins anywhere --
                                                     //  16 / 32 frames: one 0.032 / 0.043 ms, few 0.035 / 0.047, batch 0.042 / 0.046;
                                                     //  40 / 64 / 96 frames: batch 0.049 / 0.061 / 0.071, few 0.051 / 0.069 / 0.093)
        const std::vector<OrbPyrChain>& chains = which == 2 ? h->pyrChainsOne : which == 1 ? h->pyrChainsLat : h->pyrChains;
        size_t stampOff = 0;                                   // diagnostics: 8 words per workgroup, launch after launch
        int persistent = 0;
        for (size_t c = 0; c < chains.size(); c++) {
            const size_t words = (size_t)chains[c].bands * n * 8;
            unsigned long long* stp = h->pyrStamps && stampOff + words <= h->pyrStampCap ? h->pyrStamps + stampOff : nullptr;
            stampOff += words;
            persistent += orb_launch_pyr_chain(st, chains[c], d_imgs, rowStride, frameStride, pyr, h->pyrSlab, (const uint4*)h->dXq.p,
                                 (const int2*)h->dYtab.p, (const int2*)h->dBand.p, n, c == 0 ? h->errP() : nullptr,
                                 (int)orb_extractor::batchInts(n), h->patternPtr, (float*)h->dPatternF.p, stp, which == 0) ? 1 : 0;
        }
        h->pyrPersistent = persistent;
        h->pyrStampChains = (int)chains.size();
        for (size_t c = 0; c < chains.size() && c < 8; c++) { h->pyrStampBands[c] = chains[c].bands; h->pyrStampSteps[c] = chains[c].nSteps; }
    } else {
    orb_launch_copy_level0(st, d_imgs, rowStride, frameStride, pyr, h->pyrSlab, G.L[0].w, G.L[0].h, G.L[0].pitch, n, h->errP(),
                           (int)orb_extractor::batchInts(n), h->patternPtr, (float*)h->dPatternF.p);
    {
        auto xq_of = [&](int l) { return h->xqOff[l] >= 0 ? (const uint4*)h->dXq.p + h->xqOff[l] : (const uint4*)nullptr; };
        auto ytab_of = [&](int l) { return (const int2*)h->dYtab.p + h->ytabOff[l]; };
        for (int l = 1; l < G.nlevels;) {
            if (l + 1 < G.nlevels &&
                orb_launch_resize_pair(st, pyr, h->pyrSlab, G.L[l - 1], G.L[l], G.L[l + 1], xq_of(l), ytab_of(l), xq_of(l + 1),
                                       ytab_of(l + 1), n)) {
                l += 2;
                continue;
            }
            orb_launch_resize(st, pyr, h->pyrSlab, G.L[l - 1], G.L[l], (const int2*)h->dXtab.p + h->xtabOff[l], ytab_of(l),
                              xq_of(l), n);
            l++;
        }
    }
    }
    if (prof) ORB_HIP_TRY(hipEventRecord(pe[1], st));
    orb_launch_fast_strips(st, G, pyr, h->pyrSlab, (const OrbStrip*)h->dCells.p, (int)h->strips.size(),
                           (const uint32_t*)h->dPath.p, scand, h->candSlab, scc, serr, h->ovfCountP(), (int*)h->dOvf.p,
                           h->prm.ini_th_fast, h->prm.min_th_fast, h->fastPdw, h->fastRows, h->fastSdw, h->fastCandCap, n, h->fastP, h->specNoDense);
    if (prof) ORB_HIP_TRY(hipEventRecord(pe[2], st));
    orb_launch_quadtree(st, G, scand, h->candSlab, scc, skpl, skc, serr, h->sortCap, h->nodeCap, n, h->ovfCountP(),
                        h->qtGlobal ? (unsigned char*)h->dQt.p : nullptr);
    if (prof) ORB_HIP_TRY(hipEventRecord(pe[3], st));
    // Batches: the upper pyramid levels go through the level-resident kernel (one staging + one blur per level region, the
    // angle arithmetic once per 64 keypoints; orb_desc_level.hip), the lower ones -- whose patches hardly overlap -- keep a
    // wave per keypoint.  A few frames: every keypoint its own wave (a level's workgroup would be the latency of the call).
    // (the knobs are read only where a plan exists, i.e. with ORB_DESC_LEVEL=1: a getenv is a scan of the environment, and this
    // function is a single frame's whole host cost in config 5)
    const bool havePlan = h->descPlan.nRegions > 0;
    const char* lmfEnv = havePlan ? std::getenv("ORB_DESC_LEVEL_MIN_FRAMES") : nullptr;
    const int levelMinFrames = lmfEnv ? std::atoi(lmfEnv) : 24;
    const bool useLevel = havePlan && n >= levelMinFrames;
    // The two kernels are independent (disjoint keypoint slots).  ORB_DESC_LEVEL_SIDE=1 runs the level-resident one on a side
    // stream BESIDE the per-keypoint one (the idea: single-wave workgroups of k_orient_desc fill the issue cycles that staging
    // and barriers leave) -- measured slower than one after the other (0.412 against 0.396 ms per 512 frames: the level kernel's
    // workgroups hold half a CU's LDS each and keep the other kernel's waves OUT), so it is off.
    const char* sideEnv = useLevel ? std::getenv("ORB_DESC_LEVEL_SIDE") : nullptr;
    const bool wantSide = useLevel && n >= 24 && sideEnv && std::atoi(sideEnv) != 0;
    if (wantSide && !h->sideStream) {
        // created on first use only: HIP deals streams round-robin over a few hardware queues, and a stream that exists but is
        // never used still shifts which queues the OTHER handles' streams land on (two lanes of a 64-frame batch overlapped
        // worse with an idle side stream per handle: 251 k against 328 k frames/s)
        if (orb_stream_create(&h->sideStream, h->device, 2) != hipSuccess) h->sideStream = nullptr;
        (void)hipEventCreateWithFlags(&h->sideFork, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&h->sideJoin, hipEventDisableTiming);
        (void)hipGetLastError();
    }
    const bool side = wantSide && h->sideStream && h->sideFork && h->sideJoin;
    hipStream_t lst = st;
    if (side) {
        ORB_HIP_TRY(hipEventRecord(h->sideFork, st));
        ORB_HIP_TRY(hipStreamWaitEvent(h->sideStream, h->sideFork, 0));
        lst = h->sideStream;
    }
    if (useLevel && orb_launch_desc_level(lst, G, h->descPlan, pyr, h->pyrSlab, skpl, skc, (const float*)h->dPatternF.p, (const uint4*)h->dAngTab.p, d_kps,
                                          d_desc, cap, n, h->gaussTaps, h->descStamps, h->descStampCap) != 0) {
        orb_set_error("k_desc_level: the launch could not be set up");
        return ORB_ERR_HIP;
    }
    orb_launch_orient_desc(st, G, pyr, h->pyrSlab, skpl, skc, (const float*)h->dPatternF.p, (const uint4*)h->dAngTab.p, (const uint32_t*)((const uint8_t*)h->dAngTab.p + 16 * 2 * 32), d_kps, d_desc, cap,
                           d_counts, serr, n, h->gaussTaps, useLevel ? G.L[h->descPlan.firstLevel].kpBase : 0);
    if (side) {
        ORB_HIP_TRY(hipEventRecord(h->sideJoin, lst));
        ORB_HIP_TRY(hipStreamWaitEvent(st, h->sideJoin, 0));
    }
    if (prof) {
        ORB_HIP_TRY(hipEventRecord(pe[4], st));
        h->profCount++;
        h->profFrames = n;
    }
    h->batchSerial++;
    // (every fourth batch is enough: overflowing content keeps overflowing, and the copy is a bubble at the end of the chain)
    if (!h->hostCall && !h->fastStripFixed && h->ovfHost && h->ovfPendingSerial == 0 && (h->batchSerial & 3u) == 1u) {
        ORB_HIP_TRY(hipMemcpyAsync(h->ovfHost, h->ovfCountP(), orb_extractor::kOvfInts * 4, hipMemcpyDeviceToHost, st));
        ORB_HIP_TRY(hipEventRecord(h->ovfEv, st));
        h->ovfPendingSerial = h->batchSerial;
    }
    ORB_HIP_TRY(hipGetLastError());
    return ORB_OK;
}

// host-side part of a sync: interpret the status block (already in h->hStat)
int orb_check_status(orb_extractor* h)
{
    const int n = h->lastFrames;
    const int sticky = h->hStat[orb_extractor::kStickyInts - 1];
    const int* err = h->hStat.data() + orb_extractor::kStickyInts;
    const int* cand = err + n;
    {                                                  // adapt the quadtree's LDS sort capacity to the data
        int mx = 0;
        for (size_t i = 0; i < (size_t)ORB_MAX_LEVELS * n; i++) mx = std::max(mx, cand[i]);
        grow_sort_cap(h, mx);
    }
    // (h->statSerial: the batch this block belongs to -- a pipelined host batch retires chunk k while chunk k+1, launched
    // with the same strips, is in flight; its counters must not shorten the strips a second time)
    apply_fast_overflows(h, err + (size_t)(1 + 2 * ORB_MAX_LEVELS) * n + 8, h->statSerial ? h->statSerial : h->batchSerial);
    for (int f = 0; f < n; f++)
        if (err[f]) {
            orb_set_error("device-side overflow flag 0x%x on frame %d (1 candidates, 2 nodes, 4 output cap)", err[f], f);
            return (err[f] & 4) ? ORB_ERR_CAPACITY : ORB_ERR_INTERNAL;
        }
    if (sticky) {                                      // raised by an earlier batch that was never synchronised
        orb_set_error("device-side overflow flag 0x%x in an earlier, unsynchronised batch (1 candidates, 2 nodes, 4 output cap)", sticky);
        return (sticky & 4) ? ORB_ERR_CAPACITY : ORB_ERR_INTERNAL;
    }
    return ORB_OK;
}

extern "C" int orb_extractor_sync(orb_extractor* h)
{
    if (!h) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(h->device));
    ORB_HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->lastFrames > 0 && !h->statFetched) {
        h->hStat.resize(orb_extractor::statInts(h->lastFrames));
        ORB_HIP_TRY(orb_copy_blocking(h->hStat.data(), h->dStat.p, h->hStat.size() * 4, hipMemcpyDeviceToHost, h->stream));
        h->statFetched = true;
        h->statSerial = 0;
        if (h->hStat[orb_extractor::kStickyInts - 1]) ORB_HIP_TRY(orb_fill_blocking(h->dStat.p, 0, orb_extractor::kStickyInts * 4, h->stream));
        return orb_check_status(h);
    }
    return ORB_OK;
}

static int ensure_stage(orb_extractor* h, size_t bytes)
{
    if (bytes <= h->hStageBytes) return ORB_OK;
    if (h->hStage) { (void)hipHostFree(h->hStage); h->hStage = nullptr; h->hStageBytes = 0; }
    ORB_HIP_TRY(hipHostMalloc(&h->hStage, bytes, hipHostMallocDefault));
    h->hStageBytes = bytes;
    return ORB_OK;
}

// Host-buffer entry: H2D, the launch chain, and ONE round trip back.  Status words, keypoints and descriptors are
// fetched with three asynchronous copies into pinned staging followed by a single stream synchronisation (the
// capacity-sized slabs are copied whole; 66 KB per frame at nFeatures = 1000), then the valid prefixes are
// memcpy'd into the caller's (pageable) buffers.  Batches whose slabs exceed kStageLimit fall back to
// count-then-copy per frame.
extern "C" int orb_extract_batch(orb_extractor* h, const uint8_t* imgs, int nFrames, int rows, int cols,
                                 size_t rowStride, size_t frameStride, orb_keypoint* kps, uint8_t* desc, int cap,
                                 int32_t* counts)
{
    if (!h || nFrames < 0 || !counts) return ORB_ERR_INVALID;
    if (nFrames == 0) return ORB_OK;
    if (!imgs || rows <= 0 || cols <= 0) {
        for (int f = 0; f < nFrames; f++) counts[f] = 0;
        return ORB_OK;
    }
    if (!kps || !desc || cap <= 0) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(h->device));
    struct HostCall { orb_extractor* h; explicit HostCall(orb_extractor* x) : h(x) { h->hostCall = true; } ~HostCall() { h->hostCall = false; } } hostCallGuard(h);
    if (nFrames >= 2 * ORB_PIPE_CHUNK_MIN)                       // large batch: H2D(k+1) | kernels(k) | D2H(k-1)
        return orb_extract_batch_pipelined(h, imgs, nFrames, rows, cols, rowStride, frameStride, kps, desc, cap, counts);
    h->frameBase = 0;
    int rc;
    const size_t imgBytes = (size_t)rows * cols;
    const size_t kStageLimit = (size_t)96 << 20;
    const size_t statB = orb_extractor::statInts(nFrames) * 4, cntB = (size_t)4 * nFrames;
    const size_t kpB = sizeof(orb_keypoint) * (size_t)cap * nFrames, dsB = (size_t)ORB_DESC_BYTES * cap * nFrames;
    // where the pieces sit in the pinned staging: keypoints and descriptors on 16-byte boundaries (the descriptor kernel stores
    // 8-byte words straight into it in the single-frame path; 28 * cap bytes of keypoints would leave them 4-byte aligned)
    const size_t kpOff = (statB + cntB + 15) & ~(size_t)15, dsOff = (kpOff + kpB + 15) & ~(size_t)15, stgEnd = dsOff + dsB;
    const bool whole = stgEnd <= kStageLimit;
    // Single frames (the reference's operator() path) skip both DMA legs: the image is copied by the CPU into the handle's
    // pinned staging and the first pyramid kernel reads it from there over PCIe; keypoints, descriptors and the count are
    // written by the descriptor kernel straight into pinned staging.  What is left on the copy engines is the 200-byte
    // status block.  (A pageable hipMemcpyAsync of 307 KB costs more than the CPU copy + the kernel's reads.)
    const bool zero = nFrames == 1 && whole && !std::getenv("ORB_NO_ZEROCOPY");
    const size_t imgOff = (stgEnd + 255) & ~(size_t)255;
    if ((rc = ensure_stage(h, whole ? (zero ? imgOff + imgBytes + 256 : stgEnd) : statB + cntB)) != ORB_OK) return rc;
    uint8_t* stg = (uint8_t*)h->hStage;
    const double tm0 = g_timing ? now_us() : 0.0;
    if (zero) {
        uint8_t* dst = stg + imgOff;
        if (rowStride == (size_t)cols) std::memcpy(dst, imgs, imgBytes);
        else
            for (int y = 0; y < rows; y++) std::memcpy(dst + (size_t)y * cols, imgs + (size_t)y * rowStride, (size_t)cols);
    } else {
        if ((rc = h->dImgs.ensure(imgBytes * nFrames)) != ORB_OK) return rc;
        if ((rc = h->dKps.ensure(sizeof(orb_keypoint) * (size_t)cap * nFrames)) != ORB_OK) return rc;
        if ((rc = h->dDesc.ensure((size_t)ORB_DESC_BYTES * cap * nFrames)) != ORB_OK) return rc;
        if ((rc = h->dCounts.ensure((size_t)4 * nFrames)) != ORB_OK) return rc;
        if (rowStride == (size_t)cols && (frameStride == imgBytes || nFrames == 1)) {
            // contiguous frames: one linear copy (a 2-D copy of an odd width such as 1241 takes a slow row-wise path)
            ORB_HIP_TRY(hipMemcpyAsync(h->dImgs.p, imgs, imgBytes * nFrames, hipMemcpyHostToDevice, h->stream));
        } else {
            for (int f = 0; f < nFrames; f++) {
                if (rowStride == (size_t)cols)
                    ORB_HIP_TRY(hipMemcpyAsync((uint8_t*)h->dImgs.p + imgBytes * f, imgs + frameStride * f, imgBytes,
                                               hipMemcpyHostToDevice, h->stream));
                else
                    ORB_HIP_TRY(hipMemcpy2DAsync((uint8_t*)h->dImgs.p + imgBytes * f, cols, imgs + frameStride * f, rowStride,
                                                 cols, rows, hipMemcpyHostToDevice, h->stream));
            }
        }
    }
    // the chain and the copies back; issued eagerly, or (single frames, the reference's per-call path) as one graph
    auto chain = [&]() -> int {
        if (zero) {
            int r = orb_extract_batch_device(h, stg + imgOff, 1, rows, cols, cols, imgBytes, (orb_keypoint*)(stg + kpOff),
                                             stg + dsOff, cap, (int32_t*)(stg + statB));
            if (r != ORB_OK) return r;
            ORB_HIP_TRY(hipMemcpyAsync(stg, h->dStat.p, statB, hipMemcpyDeviceToHost, h->stream));
            return ORB_OK;
        }
        int r = orb_extract_batch_device(h, (const uint8_t*)h->dImgs.p, nFrames, rows, cols, cols, imgBytes,
                                         (orb_keypoint*)h->dKps.p, (uint8_t*)h->dDesc.p, cap, (int32_t*)h->dCounts.p);
        if (r != ORB_OK) return r;
        ORB_HIP_TRY(hipMemcpyAsync(stg, h->dStat.p, statB, hipMemcpyDeviceToHost, h->stream));
        ORB_HIP_TRY(hipMemcpyAsync(stg + statB, h->dCounts.p, cntB, hipMemcpyDeviceToHost, h->stream));
        if (whole) {
            ORB_HIP_TRY(hipMemcpyAsync(stg + kpOff, h->dKps.p, kpB, hipMemcpyDeviceToHost, h->stream));
            ORB_HIP_TRY(hipMemcpyAsync(stg + dsOff, h->dDesc.p, dsB, hipMemcpyDeviceToHost, h->stream));
        }
        return ORB_OK;
    };
    const double tm1 = g_timing ? now_us() : 0.0;
    // Single frames leave the (almost always idle) launch for overflowed FAST strips out of the chain: ~4.5 us of a ~95 us chain.
    // The overflow counter comes back with the status block; if a strip did overflow the frame is redone with that kernel
    // (and the strips get shorter: apply_fast_overflows).
    const bool noSpec = std::getenv("ORB_NO_SPEC") != nullptr;
    struct SpecGuard { orb_extractor* h; ~SpecGuard() { h->specNoDense = false; } } specGuard{h};
    h->specNoDense = zero && !noSpec;
    orb_extractor::Graph& Gr = h->graph1;
    const bool graphable = nFrames == 1 && whole && !h->profiling && !Gr.broken && !std::getenv("ORB_NO_GRAPH");
    const void* curBufs[10];
    h->graph_bufs(curBufs);                                    // (after the ensure() calls above: what this call will address)
    const bool sameKey = graphable && Gr.rows == rows && Gr.cols == cols && Gr.cap == cap && Gr.sortCap == h->sortCap &&
                         Gr.geomVersion == h->geomVersion && Gr.pattern == (const void*)h->patternPtr && Gr.stage == h->hStage &&
                         rows == h->rows && cols == h->cols && !h->geomDirty && h->framesCap >= 1 &&
                         std::memcmp(Gr.bufs, curBufs, sizeof(curBufs)) == 0;
    bool launched = false;
    if (sameKey && !Gr.exec) {
        // an eager call with this key has been through: everything is allocated and built, so the same calls can be
        // captured (capturing does not execute them)
        if (Gr.graph) { (void)hipGraphDestroy(Gr.graph); Gr.graph = nullptr; }
        ORB_HIP_TRY(hipStreamSynchronize(h->stream));
        bool ok = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            const int r = chain();
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(h->stream, &g);
            ok = r == ORB_OK && e == hipSuccess && g != nullptr;
            if (ok) { Gr.graph = g; ok = hipGraphInstantiate(&Gr.exec, g, nullptr, nullptr, 0) == hipSuccess; }
            else if (g) (void)hipGraphDestroy(g);
        }
        if (!ok) {                                             // stay eager for the rest of the handle's life
            if (std::getenv("ORB_DEBUG_CAPTURE")) {
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                const hipError_t q = hipStreamIsCapturing(h->stream, &cs);
                std::fprintf(stderr, "[orb] capture failed on handle %p: last error '%s', isCapturing -> %s, status %d; message '%s'\n", (void*)h,
                             hipGetErrorString(hipPeekAtLastError()), hipGetErrorString(q), (int)cs, orb_last_error());
            }
            (void)hipGetLastError();
            Gr.broken = true;
            Gr.exec = nullptr;
            // a capture that something outside this library invalidated can leave the stream in the "invalidated" state even
            // after hipStreamEndCapture (seen on ROCm 7.2: every later launch on it fails): the stream is idle here, replace it
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(h->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
                (void)hipGetLastError();
                // The old stream stays alive until orb_extractor_destroy: its handle is public (orb_extractor_stream: torch
                // ExternalStream wrappers, orb_matcher_wait_for, the stereo partner) and a caller that cached it must not be left
                // with a dangling one; orb_extractor_stream returns the replacement from now on.
                hipStream_t fresh = nullptr;
                if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) {
                    (void)hipGetLastError();
                    orb_set_error("orb_extract: the handle's stream was invalidated by a foreign capture and no replacement could be created");
                    return ORB_ERR_HIP;
                }
                h->retiredStreams.push_back(h->stream);
                h->stream = fresh;
                (void)hipGetLastError();
            }
        }
    }
    if (sameKey && Gr.exec) {
        h->lastFrames = 1; h->frameBase = 0; h->statFetched = false;
        if (hipGraphLaunch(Gr.exec, h->stream) == hipSuccess) launched = true;
        else { (void)hipGetLastError(); Gr.broken = true; }
    }
    if (!launched) {
        if ((rc = chain()) != ORB_OK) return rc;
        if (graphable && !Gr.broken && !sameKey && h->lastFrames == 1) {     // remember the key: the next call captures
            Gr.rows = rows; Gr.cols = cols; Gr.cap = cap; Gr.sortCap = h->sortCap; Gr.geomVersion = h->geomVersion;
            Gr.pattern = (const void*)h->patternPtr; Gr.stage = h->hStage;
            h->graph_bufs(Gr.bufs);                            // (after the eager chain: ensure_scratch has run)
            if (Gr.exec) { (void)hipGraphExecDestroy(Gr.exec); Gr.exec = nullptr; }
        }
    }
    if (h->lastFrames == 0) {                                  // nothing was launched
        for (int f = 0; f < nFrames; f++) counts[f] = 0;
        return ORB_OK;
    }
    const double tm2 = g_timing ? now_us() : 0.0;
    ORB_HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->specNoDense && ((const int*)stg)[orb_extractor::kStickyInts + (size_t)(1 + 2 * ORB_MAX_LEVELS) * nFrames] > 0) {
        h->specNoDense = false;                                // (the graph keeps the short chain; this call is eager)
        if ((rc = chain()) != ORB_OK) return rc;
        ORB_HIP_TRY(hipStreamSynchronize(h->stream));
    }
    const double tm3 = g_timing ? now_us() : 0.0;
    h->hStat.assign((const int*)stg, (const int*)stg + orb_extractor::statInts(nFrames));
    h->statFetched = true;
    h->statSerial = 0;                                         // = the batch just run
    if (h->hStat[orb_extractor::kStickyInts - 1]) ORB_HIP_TRY(orb_fill_blocking(h->dStat.p, 0, orb_extractor::kStickyInts * 4, h->stream));
    std::memcpy(counts, stg + statB, cntB);
    if ((rc = orb_check_status(h)) != ORB_OK) return rc;
    for (int f = 0; f < nFrames; f++) {
        const int n = counts[f];
        if (n <= 0) continue;
        if (whole) {
            std::memcpy(kps + (size_t)cap * f, stg + kpOff + sizeof(orb_keypoint) * (size_t)cap * f, sizeof(orb_keypoint) * n);
            std::memcpy(desc + (size_t)ORB_DESC_BYTES * cap * f, stg + dsOff + (size_t)ORB_DESC_BYTES * cap * f,
                        (size_t)ORB_DESC_BYTES * n);
        } else {
            ORB_HIP_TRY(hipMemcpyAsync(kps + (size_t)cap * f, (orb_keypoint*)h->dKps.p + (size_t)cap * f,
                                       sizeof(orb_keypoint) * n, hipMemcpyDeviceToHost, h->stream));
            ORB_HIP_TRY(hipMemcpyAsync(desc + (size_t)ORB_DESC_BYTES * cap * f,
                                       (uint8_t*)h->dDesc.p + (size_t)ORB_DESC_BYTES * cap * f,
                                       (size_t)ORB_DESC_BYTES * n, hipMemcpyDeviceToHost, h->stream));
        }
    }
    if (!whole) ORB_HIP_TRY(hipStreamSynchronize(h->stream));
    if (g_timing && nFrames == 1) {
        const double tm4 = now_us();
        g_tAcc[0] += tm1 - tm0; g_tAcc[1] += tm2 - tm1; g_tAcc[2] += tm3 - tm2; g_tAcc[3] += tm4 - tm3;
        g_tCalls++;
    }
    return ORB_OK;
}

extern "C" int orb_extract(orb_extractor* h, const uint8_t* img, int rows, int cols, size_t stride,
                           orb_keypoint* kps, uint8_t* desc, int cap, int* n)
{
    if (!h || !n) return ORB_ERR_INVALID;
    int32_t cnt = 0;
    const int rc = orb_extract_batch(h, img, 1, rows, cols, stride, 0, kps, desc, cap, &cnt);
    *n = cnt;
    return rc;
}

extern "C" int orb_get_pyramid_level(orb_extractor* h, int frame, int level, uint8_t* dst, size_t dstStride,
                                     int* rows, int* cols)
{
    if (!h || level < 0 || level >= h->prm.nlevels || h->rows == 0) return ORB_ERR_INVALID;
    const OrbLevelGeom& L = h->G.L[level];
    if (rows) *rows = L.h;
    if (cols) *cols = L.w;
    if (!dst) return ORB_OK;
    frame -= h->frameBase;                                     // a pipelined host batch keeps its last chunk on the device
    if (frame < 0 || frame >= h->lastFrames || dstStride < (size_t)L.w) {
        orb_set_error("frame not resident (device-resident frames of the last batch: %d..%d)", h->frameBase, h->frameBase + h->lastFrames - 1);
        return ORB_ERR_INVALID;
    }
    ORB_HIP_TRY(hipSetDevice(h->device));
    ORB_HIP_TRY(hipMemcpy2DAsync(dst, dstStride, (const uint8_t*)h->dPyr.p + h->pyrSlab * frame + L.pyrOff, L.pitch,
                                 L.w, L.h, hipMemcpyDeviceToHost, h->stream));
    ORB_HIP_TRY(hipStreamSynchronize(h->stream));
    return ORB_OK;
}

extern "C" int orb_get_pyramid(orb_extractor* h, int frame, uint8_t* dst, size_t dstBytes, size_t* bytes, int32_t* offsets,
                               int32_t* pitches, int32_t* rows, int32_t* cols)
{
    if (!h || h->rows == 0) return ORB_ERR_INVALID;
    for (int l = 0; l < h->prm.nlevels; l++) {
        const OrbLevelGeom& L = h->G.L[l];
        if (offsets) offsets[l] = L.pyrOff;
        if (pitches) pitches[l] = L.pitch;
        if (rows) rows[l] = L.h;
        if (cols) cols[l] = L.w;
    }
    if (bytes) *bytes = h->pyrSlab;
    if (!dst) return ORB_OK;
    frame -= h->frameBase;
    if (frame < 0 || frame >= h->lastFrames || dstBytes < h->pyrSlab) {
        orb_set_error("orb_get_pyramid: frame not resident or buffer smaller than %zu bytes", h->pyrSlab);
        return ORB_ERR_INVALID;
    }
    ORB_HIP_TRY(hipSetDevice(h->device));
    ORB_HIP_TRY(hipMemcpyAsync(dst, (const uint8_t*)h->dPyr.p + h->pyrSlab * frame, h->pyrSlab, hipMemcpyDeviceToHost, h->stream));
    ORB_HIP_TRY(hipStreamSynchronize(h->stream));
    return ORB_OK;
}

extern "C" void* orb_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

extern "C" void orb_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

// diagnostics: FAST strips per level of the last SYNCHRONISED batch that overflowed their candidate queue and were redone
// by k_fast_strips_dense (results are the same either way; many overflows cost time), and strips per level and frame
extern "C" int orb_get_fast_overflows(orb_extractor* h, int32_t* overflowed, int32_t* strips_per_frame)
{
    if (!h || h->hStat.size() < orb_extractor::statInts(h->lastFrames) || !h->statFetched) return ORB_ERR_INVALID;
    const int* ovf = h->hStat.data() + orb_extractor::kStickyInts + (size_t)(1 + 2 * ORB_MAX_LEVELS) * h->lastFrames + 8;
    for (int l = 0; l < h->prm.nlevels; l++) {
        if (overflowed) overflowed[l] = ovf[l];
        if (strips_per_frame) strips_per_frame[l] = h->fastStripsOfLevel[l];
    }
    return ORB_OK;
}

extern "C" int orb_extractor_set_desc_stamps(orb_extractor* h, unsigned long long* d_stamps, size_t capacity)
{
    if (!h) return ORB_ERR_INVALID;
    h->descStamps = d_stamps;
    h->descStampCap = d_stamps ? capacity : 0;
    return ORB_OK;
}

extern "C" int orb_extractor_set_qt_stamps(orb_extractor* h, unsigned long long* d_stamps, size_t capacity)
{
    if (!h) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(h->device));
    // 8 words per workgroup of a batch (frames x levels); the pointer is process-wide (one diagnostic user at a time)
    if (d_stamps && capacity < (size_t)8 * ORB_MAX_LEVELS) return ORB_ERR_INVALID;
    if (orb_quadtree_set_stamps(d_stamps, h->stream) != 0) return ORB_ERR_HIP;
    return ORB_OK;
}

extern "C" int orb_extractor_set_pyr_stamps(orb_extractor* h, unsigned long long* d_stamps, size_t capacity)
{
    if (!h) return ORB_ERR_INVALID;
    h->pyrStamps = d_stamps;
    h->pyrStampCap = d_stamps ? capacity : 0;
    return ORB_OK;
}

extern "C" int orb_extractor_pyr_stamp_layout(const orb_extractor* h, int32_t* n_chains, int32_t* bands8, int32_t* steps8)
{
    if (!h || !n_chains) return ORB_ERR_INVALID;
    *n_chains = h->pyrStampChains;
    for (int c = 0; c < 8; c++) {
        if (bands8) bands8[c] = c < h->pyrStampChains ? h->pyrStampBands[c] : 0;
        if (steps8) steps8[c] = c < h->pyrStampChains ? h->pyrStampSteps[c] : 0;
    }
    return ORB_OK;
}

extern "C" int orb_extractor_pyr_persistent(const orb_extractor* h, int32_t* launches)
{
    if (!h || !launches) return ORB_ERR_INVALID;
    *launches = h->pyrPersistent;
    return ORB_OK;
}

extern "C" int orb_extractor_desc_plan(const orb_extractor* h, int32_t* first_level, int32_t* n_regions)
{
    if (!h || h->rows <= 0) return ORB_ERR_INVALID;                // (no geometry yet)
    if (first_level) *first_level = h->descPlan.firstLevel;
    if (n_regions) *n_regions = h->descPlan.nRegions;
    return ORB_OK;
}

extern "C" int orb_get_level_counts(orb_extractor* h, int frame, int32_t* kept, int32_t* cands)
{
    if (!h) return ORB_ERR_INVALID;
    frame -= h->frameBase;                                     // as orb_get_pyramid_level: index within the whole batch
    if (frame < 0 || frame >= h->lastFrames) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(h->device));
    ORB_HIP_TRY(hipStreamSynchronize(h->stream));
    int32_t buf[ORB_MAX_LEVELS];
    if (kept) {
        ORB_HIP_TRY(orb_copy_blocking(buf, h->kpCountP() + (size_t)ORB_MAX_LEVELS * frame, sizeof(buf), hipMemcpyDeviceToHost, h->stream));
        for (int l = 0; l < h->prm.nlevels; l++) kept[l] = buf[l];
    }
    if (cands) {
        ORB_HIP_TRY(orb_copy_blocking(buf, h->candCountP() + (size_t)ORB_MAX_LEVELS * frame, sizeof(buf), hipMemcpyDeviceToHost, h->stream));
        for (int l = 0; l < h->prm.nlevels; l++) cands[l] = buf[l];
    }
    return ORB_OK;
}
