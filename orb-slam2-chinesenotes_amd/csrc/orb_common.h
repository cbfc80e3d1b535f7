// orb_common.h -- shared declarations of the HIP implementation behind include/orb_hip.h.
// gfx950 only.  There is deliberately NO CPU fallback in this library: if no device is usable
// every entry point fails with ORB_ERR_NO_DEVICE / ORB_ERR_HIP.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/orb_hip.h"

void orb_set_error(const char* fmt, ...);

#define ORB_HIP_TRY(expr)                                                                  \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            orb_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return ORB_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

// Blocking copies / fills go through a stream of the caller's (hipMemcpyAsync + hipStreamSynchronize), never through the
// legacy null stream: hipMemcpy / hipMemset on the null stream FAIL while any stream of the process is being captured into
// a graph -- and invalidate that capture -- even a thread-local capture of another thread's non-blocking stream (two
// extractor handles on two threads, src/Frame.cc:82-85: one thread's table upload met the other's graph capture).
static inline hipError_t orb_copy_blocking(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t st)
{
    const hipError_t e = hipMemcpyAsync(dst, src, n, kind, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}
static inline hipError_t orb_fill_blocking(void* dst, int v, size_t n, hipStream_t st)
{
    const hipError_t e = hipMemsetAsync(dst, v, n, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
}

// The handles' streams (csrc/orb_streams.hip): created on the hardware queue that the fewest library streams of the same role
// use (0 extractor, 1 matcher, 2 copy / side, 3 other) -- HIP spreads streams over 4 hardware queues by a policy of its own, and
// streams on one queue serialise; which of a pipeline's streams share a queue decides how it overlaps (round 5).
hipError_t orb_stream_create(hipStream_t* out, int device, int role);
void orb_stream_destroy(hipStream_t s, int device);

// ---- geometry of one pyramid level, shared by host set-up code and all kernels ----
struct OrbLevelGeom {
    int w, h, pitch;            // image size and row pitch (bytes, multiple of 64)
    int pyrOff;                 // byte offset of the level inside one frame's pyramid slab
    int nCols, nRows, wCell, hCell;   // FAST cell grid (reference src/ORBextractor.cc:820-823)
    int candBase, candCap;      // key slots inside one frame's candidate slab
    int pathXOff, pathYOff;     // offsets of the level's quadtree path tables (u32 per x / per y)
    int quota;                  // mnFeaturesPerLevel[level]
    int kpBase, kpCap;          // slots inside one frame's per-level keypoint list
    int nIni;                   // quadtree roots (reference :567)
    float hX;                   // root width (reference :568)
    int boxW, boxH;             // maxX-minX, maxY-minY
    float scale;                // mvScaleFactor[level]
    float invScale;             // mvInvScaleFactor[level]
    float sizeField;            // (float)(int)(31*scale)
};

struct OrbGeom {
    int nlevels;
    int kpSlab;                 // sum of kpCap
    unsigned long long umaxPacked;  // umax[v] in nibble v (values <= 15), reference :544-558
    OrbLevelGeom L[ORB_MAX_LEVELS];
};

// ---- pyramid chains (k_pyr_chain, orb_extract_kernels.hip): one launch produces up to ORB_PYR_MAXCHAIN consecutive levels.
// A workgroup owns a BAND of rows of the chain's last level, stages the rows of the chain's source level that the band
// draws on in LDS (coalesced 16-byte loads, no dependent address), and resamples level after level out of LDS: every
// produced level is written to HBM once and never read back inside the chain.  The first chain of a batch reads the
// caller's image and also writes level 0 (reference src/ORBextractor.cc:1173).
#define ORB_PYR_MAXCHAIN 4
struct OrbPyrStep {
    int dstOff, dstPitch, dstH; // produced level inside one frame's pyramid slab
    int x4;                     // pixel quads per row
    unsigned invX4;             // ceil(2^32 / x4), 0 when x4 == 1
    int xqOff, ytOff;           // the level's column table (uint4 units) and row table (int2 units)
    int ldsOff, ldsPitchDw;     // where the level's band lives in LDS while it is the source of the next step (bytes, dwords)
    int rpOff;                  // the step's row parameters in LDS (bytes): uint4 {source row A, source row B (LDS byte offsets), b0 << 16, b1 << 16}
};
struct OrbPyrChain {
    int nSteps, copy0;          // levels produced; 1: the source is the caller's image and level 0 is written as well
    int srcOff, srcPitch, srcW, srcH;   // source level inside the slab (copy0: where level 0 goes)
    int srcLdsOff, srcLdsPitchDw;
    int cpr;                    // 16-byte chunks per source row
    int srcRowsMax;             // source rows of the largest band (k_pyr_chain_p: chunks per thread)
    unsigned invCpr;            // ceil(2^32 / cpr), 0 when cpr == 1
    int bands, tabOff;          // workgroups per frame; offset (int2 units) of the band table [band][nSteps + 2]:
                                //   [0] source rows (first, last), [1 + k] rows of step k, [nSteps + 1] level-0 rows to copy
    int ldsBytes;
    int xqLdsOff, xqLdsN;       // > 0 entries: the column tables of the chain's levels (contiguous from st[0].xqOff, uint4 units) are
                                //   copied to LDS at xqLdsOff first (the few-frames variants: an LDS read per item, not an L2 round trip)
    OrbPyrStep st[ORB_PYR_MAXCHAIN];
};

// ---- level-resident descriptor stage (k_desc_level, orb_desc_level.hip): a REGION is a run of rows of one pyramid level (the
// whole level, or a vertical tile of it with 18 rows of halo) that one workgroup stages in LDS, blurs in place and samples.
#define ORB_DESC_LEVEL_OUTS 56      // blurred dwords a thread keeps in registers across the barrier of the in-place blur
#define ORB_DESC_MAX_REGIONS 24
struct OrbDescRegion {
    short level;
    short rx0, ry0;             // origin in the level (rx0: multiple of 16); LDS row lr holds level row reflect101(ry0 + lr - 3)
    short rw4, rh, rwPx;        // dwords per row, rows whose blur is computed, pixels per row
    short cx0, cx1, cy0, cy1;   // core: the keypoints with cx0 <= x < cx1, cy0 <= y < cy1 are this region's
    short pd;                   // LDS row pitch in dwords (odd; one pad dword on each side of the rw4)
    short nBands, bh;           // blur walk: thread = (band, dword column), bands of bh rows
    short imgRows;              // LDS rows = max(rh, nBands * bh) + 6
    unsigned char padL, padR;   // the region's left / right edge is the image's: write the reflect-101 frame there
    unsigned short pad0;
    unsigned invC, invW;        // ceil(2^32 / chunks per row), ceil(2^32 / rw4): thread -> (row, chunk), (band, column)
    int ldsBytes;
};
struct OrbDescPlan {
    int nRegions, firstLevel;   // levels firstLevel .. nlevels-1 run level-resident (firstLevel == nlevels: none)
    int ldsMax;                 // dynamic LDS of the launch: the largest region's
    int pad0;
    OrbDescRegion R[ORB_DESC_MAX_REGIONS];
};

// the 7-tap Gaussian of the descriptor kernel (k0 k1 k2 k3 k2 k1 k0, 8.8 fixed point) packed as its dot4 / dot2 operands
struct OrbGaussK {
    unsigned h0, h1;            // horizontal pass (v_dot4_u32_u8): k0 | k1 << 8 | k2 << 16 | k3 << 24,  k2 | k1 << 8 | k0 << 16
    unsigned v0, v1, v2, v3;    // vertical pass (v_dot2_u32_u16): k0 | k1 << 16, k2 | k3 << 16, k2 | k1 << 16, k0
};

// One work item of k_fast_strips: a run of `nc` horizontally adjacent FAST cells of one cell row (reference
// :826-861: cell (ci, cj) has the ROI [16 + cj*wCell, +wCell+6) x [16 + ci*hCell, +hCell+6), clipped).  The detection
// zones of adjacent cells (ROI minus cv::FAST's 3-px rim) tile the plane without overlap, so the strip is ONE tile
// whose zone columns [zLo, zHi) are cut into cells every wCell columns; only the NMS and the threshold fallback
// look at cell boundaries.  Everything the kernel would derive from the rectangle comes precomputed (the kernel is
// vector-issue bound and has no integer division).
struct OrbStrip {
    short x0, y0, w, h;         // ROI of the strip inside the level image (first cell's ROI start, last cell's ROI end)
    unsigned char level, ci, cj0, nc;
    unsigned char xoff;         // x0 & 7: the tile is staged from the 8-byte aligned column x0 - xoff
    unsigned char nx8;          // 8-byte groups per staged row = (xoff + w + 7) / 8
    unsigned char stepG;        // 64 / nx8
    unsigned char nq, stepR;    // quads (4 aligned columns) covering the zone columns; 64 / nq
    unsigned char qLo, hLo, nh; // first zone quad; first quad / number of quads of the halo-inclusive range
    unsigned char zh;           // zone rows = h - 6 (tile rows [3, 3 + zh))
    unsigned char zLo, zHi;     // zone columns [zLo, zHi) in tile bytes (zLo = xoff + 3)
    unsigned char wCell;        // zone width of every cell of the strip but the (possibly clipped) last
    short cxBase;               // cj0 * wCell - xoff: tile column -> x relative to the level's minBorderX (:868)
    unsigned short zonePx;      // (zHi - zLo) * zh
    unsigned int invX8, invQ;   // ceil(2^20 / nx8), ceil(2^20 / nq): lane -> (row, column) without division
    unsigned int invW;          // ceil(2^16 / wCell): zone column -> cell of the strip
    unsigned int pad[2];
};
static_assert(sizeof(OrbStrip) == 48, "OrbStrip is loaded as one scalar record");


// Candidate key layout (64 bit), sorted ascending by the quadtree kernel:
//   [63:60] quadtree root   [59:36] 12 x 2-bit quadrant path (depth 0 in the top bits)
//   [35:28] cell row  [27:20] cell col  [19:14] y in ROI  [13:8] x in ROI   [7:0] FAST score
// (round 3: cell row / column are 8 bits -- 256 cells of >= 30 px per axis, i.e. levels of up to 4112 px such as a
// 4096 x 2160 frame; they were 7 bits, 3870 px)
// (cell row, cell col, y, x) ascending == the order in which the reference appends candidates
// (src/ORBextractor.cc:826-871), so ties resolve exactly as its "first maximum wins" scan.
#define ORB_KEY_ROOT_SHIFT 60
#define ORB_KEY_PATH_SHIFT 36
#define ORB_KEY_CI_SHIFT 28
#define ORB_KEY_CJ_SHIFT 20
#define ORB_KEY_ORD_MASK 0xFFFFFFFu      // (cell row, cell col, y, x): the order the reference appends candidates in
#define ORB_KEY_PATH_LEVELS 12

// XCD-aware decode of a 1-D grid (frames x items).  Workgroups are dealt round-robin over the 8 XCDs of an
// MI355X (observed placement, MI355X_MICROARCH.md; used for speed only, never for correctness), and each XCD has
// its own 4 MiB L2.  Mapping id -> frame = 8 * (id / (8 * perFrame)) + id % 8 keeps every workgroup of a frame on
// ONE XCD, so overlapping patch / ROI reads of a frame (0.95 MB pyramid) hit that XCD's L2 instead of being
// fetched up to 8 times.  Grid size: perFrame * 8 * ceil(nFrames / 8); invPerFrame = ceil(2^32 / perFrame),
// exact while (id >> 3) * perFrame < 2^32 (checked by orb_xcd_grid on the host).
#ifdef __HIPCC__
// Device-side failure of frame f (1 candidate slab, 2 quadtree nodes, 4 output capacity): the per-batch word that the
// next batch's memset clears, and the handle's STICKY word in front of the status block (errFlags[-1]), which only
// orb_extractor_sync() clears -- batches issued back to back without a sync in between cannot lose a flag.
__device__ __forceinline__ void orb_flag_error(int* errFlags, int f, int bits)
{
    atomicOr(&errFlags[f], bits);
    atomicOr(&errFlags[-1], bits);
}
__device__ __forceinline__ bool orb_xcd_decode(unsigned id, unsigned perFrame, unsigned invPerFrame, int nFrames,
                                               int& frame, int& item)
{
    const unsigned xcd = id & 7u, j = id >> 3;
    const unsigned grp = __umulhi(j, invPerFrame);
    item = (int)(j - grp * perFrame);
    frame = (int)(grp * 8u + xcd);
    return frame < nFrames;
}
#endif
// host side: number of workgroups, or 0 (use the plain 2-D grid) when perFrame < 2 (its inverse does not fit 32 bits)
// or the batch is too large for the 32-bit decode
static inline unsigned orb_xcd_grid(unsigned perFrame, int nFrames, unsigned* invPerFrame)
{
    const unsigned long long groups = ((unsigned long long)nFrames + 7) / 8;
    const unsigned long long total = groups * 8ull * perFrame;
    // Few frames: frame -> XCD leaves XCDs idle (a single frame would run on ONE of the eight: 32 CUs for its ~300 strips /
    // ~1000 keypoint waves, the other 224 idle); the plain grid deals its workgroups over all of them.  A frame's pyramid
    // is then read through up to eight L2s -- irrelevant when a handful of frames is all there is.
    if ((unsigned long long)nFrames * 5 < groups * 8ull * 4) return 0;      // less than 80 % of the frame slots used
    if (perFrame < 2 || total >= (1ull << 31) || groups * perFrame * (unsigned long long)perFrame >= (1ull << 32)) return 0;
    *invPerFrame = (unsigned)(((1ull << 32) + perFrame - 1) / perFrame);
    return (unsigned)total;
}
