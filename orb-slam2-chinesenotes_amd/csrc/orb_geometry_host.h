// orb_geometry_host.h -- the HOST-ONLY set-up arithmetic of the extractor: the constructor tables of reference
// src/ORBextractor.cc:498-559, per-level sizes and the FAST cell grid (:805-849) cut into strips, quadtree roots and
// path tables (:567-579, :438-463), cv::resize's coefficient tables and the slab layout.  No HIP call in
// here: orb_extractor.hip uploads the plan, and tools/asan_geometry.cpp runs the same code under ASan + UBSan on the
// CPU (sanitizers cannot run on the GPU side of this pool).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <vector>

#include "orb_common.h"

#pragma clang fp contract(off)

struct OrbHostTables {                      // constructor tables (reference :503-558)
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> quota;
    int umax[16];
};

struct OrbGeomPlan {                        // everything build_geometry derives from (rows, cols) before touching the device
    OrbGeom G;
    std::vector<OrbStrip> strips;
    int nCells = 0;
    int stripsOfLevel[ORB_MAX_LEVELS];
    size_t pyrSlab = 0, candSlab = 0;
    int nodeCap = 0, maxKp = 0, fastPdw = 4, fastRows = 7, fastSdw = 1;
    int fastMinNq = 255;                    // fewest zone quads of any strip (k_fast_strips_p wants >= 8)
    std::vector<uint32_t> pathTab;
    std::vector<int2> xt, yt;               // resize tables of all levels, back to back
    std::vector<uint32_t> xq;               // per-4-pixel column entries of k_resize_level4p
    std::vector<size_t> xtabOff, ytabOff;
    std::vector<long long> xqOff;
    // pyramid chains (k_pyr_chain): empty when some level is not eligible -- the per-level kernels are used then
    std::vector<OrbPyrChain> chains;        // bands of 16 rows: batches (throughput)
    std::vector<OrbPyrChain> chainsLat;     // bands of 4 rows: a few frames (4x the workgroups, each a quarter as long: latency)
    std::vector<OrbPyrChain> chainsOne;     // one or two frames: up to ORB_PYR_MAXCHAIN levels per launch, column tables in LDS
    std::vector<int2> bandTab;
};

static inline int cv_round_f(float v) { return (int)lrintf(v); }      // round-half-even
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------ ctor tables (A.1)
static void orb_build_tables(const orb_extractor_params& prm, OrbHostTables& T)
{
    const int nl = prm.nlevels;
    const double scaleFactor = (double)prm.scale_factor;            // member is double (:99 of the header)
    T.scale.assign(nl, 1.0f);
    T.sigma2.assign(nl, 1.0f);
    for (int i = 1; i < nl; i++) {
        T.scale[i] = (float)(T.scale[i - 1] * scaleFactor);
        T.sigma2[i] = T.scale[i] * T.scale[i];
    }
    T.invScale.resize(nl);
    T.invSigma2.resize(nl);
    for (int i = 0; i < nl; i++) {
        T.invScale[i] = 1.0f / T.scale[i];
        T.invSigma2[i] = 1.0f / T.sigma2[i];
    }
    T.quota.resize(nl);
    const float factor = (float)(1.0f / scaleFactor);
    float nDesired = prm.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        T.quota[l] = cv_round_f(nDesired);
        sum += T.quota[l];
        nDesired *= factor;
    }
    T.quota[nl - 1] = std::max(prm.nfeatures - sum, 0);

    // umax (:544-558)
    const int HP = 15;
    int v, v0;
    const int vmax = (int)std::floor(HP * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int)std::ceil(HP * std::sqrt(2.f) / 2);
    const double hp2 = HP * HP;
    for (v = 0; v <= vmax; ++v) T.umax[v] = cv_round_d(std::sqrt(hp2 - v * v));
    for (v = HP, v0 = 0; v >= vmin; --v) {
        while (T.umax[v0] == T.umax[v0 + 1]) ++v0;
        T.umax[v] = v0;
        ++v0;
    }
}

// cv::resize INTER_LINEAR coefficient set-up for one axis
static void axis_table(int srcLen, int dstLen, bool isX, std::vector<int2>& out)
{
    out.resize(dstLen);
    const double invScale = (double)dstLen / srcLen;
    const double scale = 1.0 / invScale;
    for (int d = 0; d < dstLen; d++) {
        float fr = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(fr);
        fr -= s;
        if (isX) {
            if (s < 0) { fr = 0; s = 0; }
            if (s >= srcLen - 1) { fr = 0; s = srcLen - 1; }
        }
        int c0 = cv_round_f((1.f - fr) * 2048.f), c1 = cv_round_f(fr * 2048.f);
        c0 = std::min(32767, std::max(-32768, c0));
        c1 = std::min(32767, std::max(-32768, c1));
        if (isX) {
            out[d] = make_int2(s, (c0 & 0xffff) | (c1 << 16));
        } else {
            const int s0 = std::min(std::max(s, 0), srcLen - 1), s1 = std::min(std::max(s + 1, 0), srcLen - 1);
            out[d] = make_int2(s0 | (s1 << 16), (c0 & 0xffff) | (c1 << 16));
        }
    }
}

// One row of FAST cells (reference :826-861) cut into strips of about `target` cells (k_fast_strips works on a strip)
struct FastCellRow {
    int level, ci, y0, h;
    std::vector<int> x0, w;     // ROI start / width of the cells that the reference runs cv::FAST on (cj = index)
};

// maxRowBytes > 0: no strip is staged wider than that (xoff <= 7 + ROI width), so that the fixed-pitch kernel fits
static int make_strips(const FastCellRow& row, int wCell, int target, std::vector<OrbStrip>& out, int& maxPdw, int& maxRows,
                       int& maxSdw, int& minNq, int maxRowBytes = 0)
{
    const int J = (int)row.x0.size();
    if (J == 0) return ORB_OK;
    // columns are addressed with 8 bits inside the tile: xoff (<= 7) + strip ROI width <= 255
    int kMax = std::max(1, std::min(8, (248 - 6) / wCell));
    if (maxRowBytes > 0) kMax = std::max(1, std::min(kMax, (maxRowBytes - 7 - 6) / wCell));
    const int K = std::max(1, std::min(target, kMax));
    const int nStrips = (J + K - 1) / K;
    for (int s = 0, j0 = 0; s < nStrips; s++) {
        const int nc = J / nStrips + (s < J % nStrips ? 1 : 0), j1 = j0 + nc - 1;
        OrbStrip S;
        std::memset(&S, 0, sizeof(S));
        S.x0 = (short)row.x0[j0];
        S.y0 = (short)row.y0;
        S.w = (short)(row.x0[j1] + row.w[j1] - row.x0[j0]);
        S.h = (short)row.h;
        S.level = (unsigned char)row.level; S.ci = (unsigned char)row.ci; S.cj0 = (unsigned char)j0; S.nc = (unsigned char)nc;
        const int xoff = S.x0 & 7, zLo = xoff + 3, zHi = zLo + S.w - 6;
        if (xoff + S.w > 255 || S.h > 66 || wCell > 255) {
            orb_set_error("FAST strip %dx%d too large for the kernel", (int)S.w, (int)S.h);
            return ORB_ERR_UNSUPPORTED;
        }
        const int nx8 = (xoff + S.w + 7) / 8, qLo = zLo >> 2, nq = ((zHi - 1) >> 2) + 1 - qLo;
        const int hLo = (zLo - 1) >> 2, nh = (zHi >> 2) + 1 - hLo;
        S.xoff = (unsigned char)xoff;
        S.nx8 = (unsigned char)nx8;
        S.stepG = (unsigned char)(64 / nx8);
        S.nq = (unsigned char)nq;
        S.stepR = (unsigned char)(64 / nq);
        S.qLo = (unsigned char)qLo; S.hLo = (unsigned char)hLo; S.nh = (unsigned char)nh;
        S.zh = (unsigned char)(S.h - 6);
        S.zLo = (unsigned char)zLo; S.zHi = (unsigned char)zHi;
        S.wCell = (unsigned char)wCell;
        S.cxBase = (short)(j0 * wCell - xoff);
        S.zonePx = (unsigned short)((zHi - zLo) * (S.h - 6));
        S.invX8 = ((1u << 20) + nx8 - 1) / nx8;
        S.invQ = ((1u << 20) + nq - 1) / nq;
        S.invW = ((1u << 16) + wCell - 1) / wCell;
        for (int n = 0; n < zHi - zLo; n++)                              // the kernel's division-free cell index
            if ((int)(((unsigned)n * S.invW) >> 16) != n / wCell) { orb_set_error("FAST cell index reciprocal inexact"); return ORB_ERR_INTERNAL; }
        for (int l = 0; l < 64; l++)
            if ((int)(((unsigned)l * S.invX8) >> 20) != l / nx8 || (int)(((unsigned)l * S.invQ) >> 20) != l / nq) {
                orb_set_error("FAST lane decode reciprocal inexact");
                return ORB_ERR_INTERNAL;
            }
        maxPdw = std::max(maxPdw, 2 * nx8);
        maxRows = std::max(maxRows, (int)S.h);
        maxSdw = std::max(maxSdw, nh);
        minNq = std::min(minNq, nq);
        out.push_back(S);
        j0 += nc;
    }
    return ORB_OK;
}

// ---- pyramid chains.  Levels [first, first + nSteps) in one launch, bands of `bandRows` rows of the last one.  Walks the
// row tables back from every band of the last level to the rows of each earlier level (and of the source) that the band
// needs, and widens the ranges where the bands would otherwise leave a row of an intermediate level (or of level 0, for
// the copying chain) unwritten.  Returns false when the chain is not eligible (no column table, a band too large for LDS).
static bool plan_pyr_chain(const OrbGeom& G, const std::vector<int2>& yt, const std::vector<size_t>& ytabOff,
                           const std::vector<long long>& xqOff, int first, int nSteps, bool copy0, int bandRows,
                           size_t ldsLimit, OrbPyrChain& C, std::vector<int2>& tab, bool xqLds = false)
{
    std::memset(&C, 0, sizeof(C));
    if (first < 1 || nSteps < 1 || nSteps > ORB_PYR_MAXCHAIN || first + nSteps > G.nlevels) return false;
    const OrbLevelGeom& S = G.L[first - 1];
    for (int k = 0; k < nSteps; k++) {
        const OrbLevelGeom& D = G.L[first + k];
        if (xqOff[first + k] < 0) return false;
        const int x4 = (D.w + 3) / 4;
        if ((long long)x4 * x4 * 64 >= (1ll << 32)) return false;     // multiply-high decode of (row group, quad)
        if (D.pitch >= (1 << 20) || D.h >= (1 << 15) || G.L[first + k - 1].h >= (1 << 15)) return false;
    }
    if (copy0 && first != 1) return false;
    const OrbLevelGeom& Last = G.L[first + nSteps - 1];
    const int bands = (Last.h + bandRows - 1) / bandRows;
    const int ent = nSteps + 2;
    const size_t tab0 = tab.size();
    tab.resize(tab0 + (size_t)bands * ent);
    std::vector<int> maxRows(nSteps + 1, 0);                         // [0] source, [1 + k] step k
    for (int b = 0; b < bands; b++) {
        int2* e = &tab[tab0 + (size_t)b * ent];
        const int2* prev = b ? &tab[tab0 + (size_t)(b - 1) * ent] : nullptr;
        int lo = b * bandRows, hi = std::min(lo + bandRows, Last.h) - 1;
        for (int k = nSteps - 1; k >= 0; k--) {
            const OrbLevelGeom& D = G.L[first + k];
            if (k < nSteps - 1) {                                    // an intermediate level is an output: no row may be left out
                if (b == 0) lo = 0;
                else if (lo > prev[1 + k].y + 1) lo = prev[1 + k].y + 1;
                if (b == bands - 1) hi = D.h - 1;
            }
            e[1 + k] = make_int2(lo, hi);
            maxRows[1 + k] = std::max(maxRows[1 + k], hi - lo + 1);
            const int2* t = &yt[ytabOff[first + k]];
            const int srcH = G.L[first + k - 1].h;
            const int nlo = t[lo].x & 0xffff, nhi = std::min((int)((unsigned)t[hi].x >> 16), srcH - 1);
            lo = nlo; hi = nhi;
        }
        if (copy0) {                                                 // level 0 is written by the bands that stage it
            if (b == 0) lo = 0;
            else if (lo > prev[0].y + 1) lo = prev[0].y + 1;
            if (b == bands - 1) hi = S.h - 1;
        }
        e[0] = make_int2(lo, hi);
        maxRows[0] = std::max(maxRows[0], hi - lo + 1);
    }
    for (int b = 0; b < bands; b++) {                                // level-0 rows a band copies: from its first staged row up to the next band's
        int2* e = &tab[tab0 + (size_t)b * ent];
        e[nSteps + 1] = make_int2(b == 0 ? 0 : e[0].x, b == bands - 1 ? S.h : e[ent].x);
    }
    // LDS: [row parameters of every step | region R0 | region R1]; source and step 1 live in R0, step 0 in R1
    C.nSteps = nSteps; C.copy0 = copy0 ? 1 : 0;
    C.srcOff = S.pyrOff; C.srcPitch = S.pitch; C.srcW = S.w; C.srcH = S.h;
    C.bands = bands; C.tabOff = (int)tab0;
    size_t off = 0;
    for (int k = 0; k < nSteps; k++) {
        C.st[k].rpOff = (int)off;
        off += (size_t)16 * (maxRows[1 + k] + 3);                    // uint4 per row (+ the clamped rows of the last group)
    }
    const size_t rowParamBytes = off;
    C.srcLdsPitchDw = align_up((S.w + 3) / 4 + 2, 4);
    C.cpr = (S.w + 15) / 16;
    C.srcRowsMax = maxRows[0];
    C.invCpr = C.cpr <= 1 ? 0u : (unsigned)(((1ull << 32) + C.cpr - 1) / C.cpr);
    for (int k = 0; k <= nSteps; k++)
        if (maxRows[k] + 3 > 64 || (long long)maxRows[0] * C.cpr * C.cpr >= (1ll << 32)) { tab.resize(tab0); return false; }   // one wave fills a step's row parameters
    size_t r0 = (size_t)4 * C.srcLdsPitchDw * maxRows[0], r1 = 0;
    for (int k = 0; k + 1 < nSteps; k++) {
        OrbPyrStep& T = C.st[k];
        T.ldsPitchDw = (G.L[first + k].w + 3) / 4 + 2;
        const size_t bytes = (size_t)4 * T.ldsPitchDw * maxRows[1 + k];
        if (k & 1) r0 = std::max(r0, bytes); else r1 = std::max(r1, bytes);
    }
    r0 = (r0 + 15) & ~(size_t)15; r1 = (r1 + 15) & ~(size_t)15;
    C.srcLdsOff = (int)rowParamBytes;
    for (int k = 0; k < nSteps; k++) {
        const OrbLevelGeom& D = G.L[first + k];
        OrbPyrStep& T = C.st[k];
        T.dstOff = D.pyrOff; T.dstPitch = D.pitch; T.dstH = D.h;
        T.x4 = (D.w + 3) / 4;
        T.invX4 = T.x4 <= 1 ? 0u : (unsigned)(((1ull << 32) + T.x4 - 1) / T.x4);
        T.xqOff = (int)xqOff[first + k];
        T.ytOff = (int)ytabOff[first + k];
        T.ldsOff = (int)(rowParamBytes + ((k & 1) ? 0 : r0));
    }
    C.ldsBytes = (int)(rowParamBytes + r0 + r1);
    if (xqLds) {                                                     // the levels' column tables follow each other in the xq array
        const long long n = xqOff[first + nSteps - 1] + 3ll * C.st[nSteps - 1].x4 - xqOff[first];
        bool contiguous = true;
        for (int k = 0; k + 1 < nSteps; k++) contiguous = contiguous && xqOff[first + k] + 3ll * C.st[k].x4 <= xqOff[first + k + 1];
        if (contiguous && n > 0 && n <= 1024) {                      // (4 entries of 16 bytes per thread in flight)
            C.xqLdsOff = C.ldsBytes;
            C.xqLdsN = (int)n;
            C.ldsBytes += (int)(16 * n);
        }
    }
    if ((size_t)C.ldsBytes > ldsLimit) { tab.resize(tab0); return false; }
    return true;
}

// level sizes, FAST strips, quadtree boxes, resize tables, slab layout for a rows x cols input
static int orb_plan_geometry(const orb_extractor_params& prm, const OrbHostTables& T, const int* stripK, int rows, int cols,
                             OrbGeomPlan& P, int maxStripRowBytes = 0)
{
    const int nl = prm.nlevels;
    OrbGeom& G = P.G;
    std::memset(&G, 0, sizeof(G));
    G.nlevels = nl;
    for (int i = 0; i < 16; i++) G.umaxPacked |= (unsigned long long)(T.umax[i] & 15) << (4 * i);
    std::vector<OrbStrip>& strips = P.strips;
    strips.clear();
    int nCells = 0;
    size_t pyrOff = 0;
    size_t candOff = 0;
    int kpOff = 0, nodeCap = 0;
    std::vector<uint32_t>& pathTab = P.pathTab;
    pathTab.clear();
    int maxPdw = 4, maxRows = 7, maxSdw = 1, minNq = 255;
    int* stripsOfLevel = P.stripsOfLevel;
    std::memset(P.stripsOfLevel, 0, sizeof(P.stripsOfLevel));
    for (int l = 0; l < nl; l++) {
        OrbLevelGeom& L = G.L[l];
        L.w = cv_round_f((float)cols * T.invScale[l]);                 // :1158
        L.h = cv_round_f((float)rows * T.invScale[l]);
        // (keypoint coordinates are packed in 12 bits: x <= w - 17 < 4096)
        if (L.w < 1 || L.h < 1 || L.w > 4112 || L.h > 4112) {
            orb_set_error("level %d of a %dx%d image is %dx%d: outside the supported 1..4112 px", l, cols, rows, L.w, L.h);
            return ORB_ERR_UNSUPPORTED;
        }
        L.pitch = align_up(L.w, 64);
        L.pyrOff = (int)pyrOff;
        pyrOff += (size_t)L.pitch * L.h;
        if (pyrOff > 0x7fffffffu) return ORB_ERR_UNSUPPORTED;
        L.quota = T.quota[l];
        L.scale = T.scale[l];
        L.invScale = T.invScale[l];
        L.sizeField = (float)(int)(31 * T.scale[l]);                  // :886, :895

        // FAST cell grid (:805-849)
        const int minB = 16, maxBX = L.w - 16, maxBY = L.h - 16;
        const float width = (float)(maxBX - minB), height = (float)(maxBY - minB);
        const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
        int candCap = 0;
        const size_t firstStrip = strips.size();
        const int firstCell = nCells;
        if (nCols > 0 && nRows > 0) {
            const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
            if (wCell > 60 || hCell > 60 || nCols > 256 || nRows > 256) {
                orb_set_error("FAST cell grid %dx%d cells of %dx%d px unsupported", nCols, nRows, wCell, hCell);
                return ORB_ERR_UNSUPPORTED;
            }
            L.nCols = nCols; L.nRows = nRows; L.wCell = wCell; L.hCell = hCell;
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minB + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBY - 3) continue;
                if (maxY > maxBY) maxY = (float)maxBY;
                FastCellRow row;
                row.level = l; row.ci = i; row.y0 = (int)iniY; row.h = (int)maxY - (int)iniY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minB + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBX - 6) continue;
                    if (maxX > maxBX) maxX = (float)maxBX;
                    const int cw = (int)maxX - (int)iniX;
                    if (cw < 7 || row.h < 7) continue;                   // cv::FAST finds nothing in such a ROI
                    // the skip rules are monotone in j, so the cells of a row are cj = 0 .. J-1 without gaps
                    if ((int)row.x0.size() != j) { orb_set_error("FAST cell row with a gap"); return ORB_ERR_INTERNAL; }
                    row.x0.push_back((int)iniX);
                    row.w.push_back(cw);
                    candCap += ((cw - 6 + 1) / 2) * ((row.h - 6 + 1) / 2);   // 3x3 strict NMS bound
                    nCells++;
                }
                const int rc = make_strips(row, wCell, stripK[l], strips, maxPdw, maxRows, maxSdw, minNq, maxStripRowBytes);
                if (rc != ORB_OK) return rc;
            }
        }
        L.candBase = (int)candOff;
        L.candCap = candCap;
        candOff += (size_t)align_up(std::max(candCap, 1), 2);

        // quadtree roots (:567-568)
        L.boxW = maxBX - minB;
        L.boxH = maxBY - minB;
        int nIni = 0;
        if (L.boxW > 0 && L.boxH > 0) nIni = (int)std::round((float)L.boxW / L.boxH);
        if (nIni > 15) { orb_set_error("aspect ratio %d:1 unsupported", nIni); return ORB_ERR_UNSUPPORTED; }
        // (12 ceil-halvings bring a box of up to 4096 px down to single pixels: 4096, 2048, ..., 2, 1)
        if (std::max(nIni > 0 ? (L.boxW + std::max(nIni, 1) - 1) / std::max(nIni, 1) + 1 : 0, L.boxH) > 4096) {
            orb_set_error("quadtree box %dx%d needs more than %d path levels", L.boxW, L.boxH, ORB_KEY_PATH_LEVELS);
            return ORB_ERR_UNSUPPORTED;
        }
        if (nIni <= 0) {
            // reference divides by zero here (portrait images): defined as "no keypoints"
            L.nIni = 0; L.hX = 1.f; L.candCap = 0;
            strips.resize(firstStrip);
            nCells = firstCell;
        }
        stripsOfLevel[l] = (int)(strips.size() - firstStrip);
        if (nIni > 0) {
            L.nIni = nIni;
            L.hX = (float)L.boxW / nIni;
        }
        // quadtree path tables (DivideNode bisects x and y independently, :438-463): for every candidate
        // coordinate the 12 x-decisions (even bits of the 24-bit path, + root << 24) and the 12 y-decisions
        L.pathXOff = (int)pathTab.size();
        for (int x = 0; x < std::max(L.boxW, 0); x++) {
            uint32_t code = 0;
            if (L.nIni > 0) {
                int root = (int)((float)x / L.hX);                           // :593
                if (root >= L.nIni) root = L.nIni - 1;                       // not reachable for candidate x
                int ulx = (int)(L.hX * (float)root), urx = (int)(L.hX * (float)(root + 1));   // :578-579
                for (int d = 0; d < ORB_KEY_PATH_LEVELS; d++) {
                    const int mid = ulx + ((urx - ulx + 1) >> 1);            // UL.x + ceil((UR.x-UL.x)/2)
                    const int right = !(x < mid);
                    code |= (uint32_t)right << (2 * (ORB_KEY_PATH_LEVELS - 1 - d));
                    if (right) ulx = mid; else urx = mid;
                }
                code |= (uint32_t)root << 24;
            }
            pathTab.push_back(code);
        }
        L.pathYOff = (int)pathTab.size();
        for (int y = 0; y < std::max(L.boxH, 0); y++) {
            uint32_t code = 0;
            int uly = 0, bry = L.boxH;
            for (int d = 0; d < ORB_KEY_PATH_LEVELS; d++) {
                const int mid = uly + ((bry - uly + 1) >> 1);
                const int down = !(y < mid);
                code |= (uint32_t)down << (2 * (ORB_KEY_PATH_LEVELS - 1 - d) + 1);
                if (down) uly = mid; else bry = mid;
            }
            pathTab.push_back(code);
        }
        L.kpBase = kpOff;
        L.kpCap = std::max(L.quota + 3, 4 * std::max(nIni, 1)) + 5;
        kpOff += L.kpCap;
        nodeCap = std::max(nodeCap, L.kpCap);
    }
    G.kpSlab = kpOff;
    std::vector<int2>& xt = P.xt;
    std::vector<int2>& yt = P.yt;
    std::vector<int2> t;
    xt.clear(); yt.clear();
    P.xtabOff.assign(nl, 0);
    P.ytabOff.assign(nl, 0);
    P.xqOff.assign(nl, -1);
    std::vector<uint32_t>& xq = P.xq;
    xq.clear();
    for (int l = 1; l < nl; l++) {
        axis_table(G.L[l - 1].w, G.L[l].w, true, t);
        while (t.size() % 4) t.push_back(t.back());           // k_resize_level4 reads four entries at a time
        P.xtabOff[l] = xt.size();                             // stays a multiple of 4 -> 32-byte aligned
        xt.insert(xt.end(), t.begin(), t.end());
        {   // per-thread (4 px) table of k_resize_level4p: window bases, v_perm selectors, coefficient pairs
            std::vector<uint32_t> q;
            bool ok = true;
            for (size_t i = 0; i + 3 < t.size() && ok; i += 4) {
                const int baseA = t[i].x & ~3, baseB = t[i + 2].x & ~3;
                uint32_t sel[4];
                for (int k = 0; k < 4; k++) {
                    const int o = t[i + k].x - (k < 2 ? baseA : baseB);
                    if (o < 0 || o > 6) { ok = false; break; }
                    sel[k] = (uint32_t)o | (0x0cu << 8) | ((uint32_t)(o + 1) << 16) | (0x0cu << 24);
                }
                const uint32_t e[12] = {(uint32_t)baseA, (uint32_t)baseB, sel[0], sel[1], sel[2], sel[3], (uint32_t)t[i].y,
                                        (uint32_t)t[i + 1].y, (uint32_t)t[i + 2].y, (uint32_t)t[i + 3].y, 0u, 0u};
                q.insert(q.end(), e, e + 12);
            }
            for (const int2& e : t) ok = ok && (e.y & 0x8000) == 0 && e.y >= 0;        // coefficients are 0..2048
            if (ok) {
                P.xqOff[l] = (long long)(xq.size() / 4);
                xq.insert(xq.end(), q.begin(), q.end());
            }
        }
        axis_table(G.L[l - 1].h, G.L[l].h, false, t);
        P.ytabOff[l] = yt.size();
        yt.insert(yt.end(), t.begin(), t.end());
    }
    if (strips.size() > 65535) {                                       // the FAST overflow list packs (frame << 16 | strip)
        orb_set_error("%zu FAST strips per frame: more than the 65535 the overflow list can address", strips.size());
        return ORB_ERR_UNSUPPORTED;
    }
    // pyramid chains: (1, 2) with the level-0 copy, then pairs, the last three levels together when an odd one remains
    P.chains.clear();
    P.chainsLat.clear();
    P.chainsOne.clear();
    P.bandTab.clear();
    // maxLen 0: pairs as above (throughput); > 0: as few launches as chains of <= maxLen levels allow, the earlier (larger)
    // levels in the shorter chains (a single frame's duration is launches x dependent round trips, not arithmetic)
    auto plan_set = [&](std::vector<OrbPyrChain>& set, int bandHi, int bandLo, int maxLen, size_t ldsLimit, bool xqLds) -> bool {
        for (int l = 1; l < nl;) {
            int n = std::min(2, nl - l);
            if (maxLen > 0) {
                const int left = nl - l, launches = (left + maxLen - 1) / maxLen;
                n = left / launches;
            } else {
                if (nl - (l + n) == 1) n = (l == 1) ? n : 3;      // never leave a single level for a launch of its own ...
                if (l + n > nl) n = nl - l;
            }
            OrbPyrChain C;
            bool done = false;
            for (int tryN = n; tryN >= 1 && !done; tryN--)
                for (int br = bandHi; br >= bandLo && !done; br >>= 1)
                    if (plan_pyr_chain(G, yt, P.ytabOff, P.xqOff, l, tryN, l == 1, br, ldsLimit, C, P.bandTab, xqLds)) {
                        set.push_back(C);
                        l += tryN;
                        done = true;
                    }
            if (!done) return false;
        }
        return true;
    };
    if (nl >= 2) {
        const int bandRows = std::getenv("ORB_PYR_BAND") ? std::max(2, std::min(64, std::atoi(std::getenv("ORB_PYR_BAND")))) : 16;
        const int bandOne = std::getenv("ORB_PYR_BAND_ONE") ? std::max(2, std::min(16, std::atoi(std::getenv("ORB_PYR_BAND_ONE")))) : 4;
        const bool xl = std::getenv("ORB_PYR_XQLDS") != nullptr;
        if (!plan_set(P.chains, bandRows, std::min(bandRows, 8), 0, (size_t)40 * 1024, xl) || !plan_set(P.chainsLat, 4, 2, 0, (size_t)40 * 1024, xl) ||
            !plan_set(P.chainsOne, bandOne, 2, ORB_PYR_MAXCHAIN, (size_t)60 * 1024, true)) {
            P.chains.clear(); P.chainsLat.clear(); P.chainsOne.clear(); P.bandTab.clear();
        }
    }
    P.nCells = nCells;
    P.pyrSlab = (size_t)align_up((int)pyrOff, 256);
    P.candSlab = candOff;
    P.nodeCap = nodeCap;
    P.maxKp = kpOff;
    P.fastPdw = maxPdw;
    P.fastRows = maxRows;
    P.fastSdw = maxSdw;
    P.fastMinNq = minNq;
    return ORB_OK;
}
