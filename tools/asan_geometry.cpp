// asan_geometry.cpp -- the extractor's HOST-ONLY set-up code (orb_geometry_host.h: constructor tables, level sizes, FAST
// strips, quadtree path tables, cv::resize coefficient tables, slab layout) under AddressSanitizer + UBSan on the CPU,
// over a sweep of image sizes and parameters, with the invariants the kernels rely on checked explicitly.
// build + run: make -C orb-slam2-chinesenotes_amd asan-geometry   (hipcc --cuda-host-only -fsanitize=address,undefined)
#include <cstdarg>
#include <cstdio>
#include <random>

#include "../orb-slam2-chinesenotes_amd/csrc/orb_geometry_host.h"

static char g_err[512];
void orb_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define CHECK(c)                                                                                        \
    do {                                                                                                \
        if (!(c)) { fprintf(stderr, "asan_geometry: check failed at %s:%d: %s (%dx%d nf %d nl %d sf %g)\n", __FILE__, __LINE__, #c, \
                            cols, rows, prm.nfeatures, prm.nlevels, (double)prm.scale_factor); return 1; }                        \
    } while (0)

static int one(const orb_extractor_params& prm, int rows, int cols, const int* stripK, int& planned)
{
    OrbHostTables T;
    orb_build_tables(prm, T);
    int sum = 0;
    for (int l = 0; l < prm.nlevels; l++) { CHECK(T.quota[l] >= 0); sum += T.quota[l]; }
    CHECK(sum >= prm.nfeatures || prm.nfeatures == 0 || T.quota[prm.nlevels - 1] == 0);
    for (int v = 0; v < 16; v++) CHECK(T.umax[v] >= 0 && T.umax[v] <= 15);
    OrbGeomPlan P;
    const int rc = orb_plan_geometry(prm, T, stripK, rows, cols, P);
    if (rc != ORB_OK) return rc == ORB_ERR_UNSUPPORTED ? 0 : 1;          // outside the envelope: reported, fine
    planned++;
    const OrbGeom& G = P.G;
    size_t strip = 0;
    for (int l = 0; l < prm.nlevels; l++) {
        const OrbLevelGeom& L = G.L[l];
        CHECK(L.w >= 1 && L.h >= 1 && L.pitch >= L.w && L.pitch % 64 == 0);
        CHECK((size_t)L.pyrOff + (size_t)L.pitch * L.h <= P.pyrSlab);
        CHECK((size_t)L.pathXOff + std::max(L.boxW, 0) <= P.pathTab.size() && (size_t)L.pathYOff + std::max(L.boxH, 0) <= P.pathTab.size());
        CHECK(L.kpBase + L.kpCap <= G.kpSlab && L.candBase + L.candCap <= (int)P.candSlab + 1);
        // the strips of the level: every cell of every cell row exactly once, inside the image, within the tile limits
        int cells = 0;
        for (int s = 0; s < P.stripsOfLevel[l]; s++, strip++) {
            const OrbStrip& S = P.strips[strip];
            CHECK(S.level == l && S.nc >= 1 && S.nc <= 8);
            CHECK(S.x0 >= 16 && S.y0 >= 16 && S.x0 + S.w <= L.w - 16 && S.y0 + S.h <= L.h - 16 && S.w >= 7 && S.h >= 7);
            CHECK(S.xoff == (S.x0 & 7) && S.xoff + S.w <= 255 && S.h <= 66);
            CHECK(S.zLo == S.xoff + 3 && S.zHi == S.zLo + S.w - 6 && S.zh == S.h - 6);
            CHECK(2 * S.nx8 <= P.fastPdw && S.h <= P.fastRows && S.nh <= P.fastSdw);
            CHECK(S.qLo * 4 <= S.zLo && (S.qLo + S.nq) * 4 >= S.zHi && S.hLo <= S.qLo && S.hLo + S.nh >= S.qLo + S.nq);
            CHECK((S.zHi - S.zLo + S.wCell - 1) / S.wCell == S.nc);       // zone columns cut into exactly nc cells
            CHECK(S.cxBase + S.zLo >= 0 && S.cxBase + S.zHi <= L.boxW && S.ci * L.hCell + 3 + S.zh <= L.boxH);
            CHECK(S.zonePx == (S.zHi - S.zLo) * S.zh);
            cells += S.nc;
        }
        (void)cells;
        // resize tables of the level: every source index inside the previous level
        if (l >= 1) {
            const OrbLevelGeom& Sp = G.L[l - 1];
            for (int x = 0; x < L.w; x++) {
                const int2 e = P.xt[P.xtabOff[l] + x];
                CHECK(e.x >= 0 && e.x <= Sp.w - 1);
            }
            for (int y = 0; y < L.h; y++) {
                const int2 e = P.yt[P.ytabOff[l] + y];
                CHECK((e.x & 0xffff) <= Sp.h - 1 && ((unsigned)e.x >> 16) <= (unsigned)(Sp.h - 1));
            }
            if (P.xqOff[l] >= 0) CHECK((size_t)P.xqOff[l] * 4 + (size_t)((L.w + 3) / 4) * 12 <= P.xq.size());
        }
    }
    CHECK(strip == P.strips.size());
    // pyramid chains (k_pyr_chain): the chains produce levels 1 .. nl-1 in order; inside a chain every band's row ranges stay
    // inside their levels and inside the LDS regions planned for them, the rows a step reads are rows its source band holds,
    // the bands of every produced level (and the level-0 copy partitions) cover the level without a gap
    CHECK(P.chains.empty() == P.chainsLat.empty() && P.chains.empty() == P.chainsOne.empty());
    for (const std::vector<OrbPyrChain>* set : {&P.chains, &P.chainsLat, &P.chainsOne}) if (!set->empty()) {
        int next = 1;
        for (const OrbPyrChain& C : *set) {
            CHECK(C.nSteps >= 1 && C.nSteps <= ORB_PYR_MAXCHAIN && next + C.nSteps <= prm.nlevels);
            const int first = next, ent = C.nSteps + 2;
            CHECK(C.copy0 == (first == 1 ? 1 : 0));
            CHECK(C.srcOff == G.L[first - 1].pyrOff && C.srcW == G.L[first - 1].w && C.srcH == G.L[first - 1].h);
            CHECK((size_t)C.tabOff + (size_t)C.bands * ent <= P.bandTab.size() && C.ldsBytes <= (set == &P.chainsOne ? 60 : 40) * 1024);
            const int bandBytes = C.xqLdsN > 0 ? C.xqLdsOff : C.ldsBytes;   // the row bands end where the column tables start
            if (C.xqLdsN > 0) {
                CHECK(C.xqLdsN <= 1024 && (C.xqLdsOff % 16) == 0 && C.xqLdsOff + 16 * C.xqLdsN == C.ldsBytes);
                for (int k = 0; k < C.nSteps; k++)
                    CHECK(C.st[k].xqOff >= C.st[0].xqOff && C.st[k].xqOff - C.st[0].xqOff + 3 * C.st[k].x4 <= C.xqLdsN);
                CHECK((size_t)(C.st[0].xqOff + C.xqLdsN) * 4 <= P.xq.size());
            }
            CHECK(C.cpr == (C.srcW + 15) / 16 && 16 * C.cpr <= 4 * C.srcLdsPitchDw && (C.srcLdsPitchDw % 4) == 0);
            std::vector<int> covered(C.nSteps + 2, 0);                 // next uncovered row of [source copy, steps...]
            for (int b = 0; b < C.bands; b++) {
                const int2* e = &P.bandTab[C.tabOff + (size_t)b * ent];
                CHECK(e[0].x >= 0 && e[0].x <= e[0].y && e[0].y < C.srcH);
                size_t srcBytes = (size_t)4 * C.srcLdsPitchDw * (e[0].y - e[0].x + 1);
                size_t srcOff = C.srcLdsOff;
                int srcRow0 = e[0].x, srcRow1 = e[0].y;
                for (int k = 0; k < C.nSteps; k++) {
                    const OrbPyrStep& T = C.st[k];
                    const OrbLevelGeom& D = G.L[first + k];
                    CHECK(T.dstOff == D.pyrOff && T.dstPitch == D.pitch && T.dstH == D.h && T.x4 == (D.w + 3) / 4);
                    CHECK(e[1 + k].x >= 0 && e[1 + k].x <= e[1 + k].y && e[1 + k].y < D.h);
                    CHECK(e[1 + k].y - e[1 + k].x + 1 + 3 <= 64);
                    CHECK((size_t)T.rpOff + 16 * (size_t)((e[1 + k].y - e[1 + k].x + 4) & ~3) <= (size_t)C.srcLdsOff);
                    CHECK(srcOff + srcBytes <= (size_t)bandBytes);
                    for (int y = e[1 + k].x; y <= e[1 + k].y; y++) {       // rows read lie inside the source band
                        const int2 t = P.yt[P.ytabOff[first + k] + y];
                        CHECK((t.x & 0xffff) >= srcRow0 && (int)((unsigned)t.x >> 16) <= srcRow1);
                    }
                    // the 8-byte windows of the last pixel quad stay inside the source row's LDS pitch
                    const int srcPitchDw = k == 0 ? C.srcLdsPitchDw : C.st[k - 1].ldsPitchDw;
                    for (int q = 0; q < T.x4; q++) {
                        const uint32_t* xe = &P.xq[(size_t)T.xqOff * 4 + (size_t)q * 12];
                        CHECK(xe[0] / 4 + 1 < (unsigned)srcPitchDw && xe[1] / 4 + 1 < (unsigned)srcPitchDw);
                    }
                    if (b == 0) CHECK(e[1 + k].x == 0);
                    else CHECK(e[1 + k].x <= covered[1 + k]);
                    covered[1 + k] = std::max(covered[1 + k], e[1 + k].y + 1);
                    if (k + 1 < C.nSteps) {
                        CHECK(T.ldsPitchDw >= T.x4 + 2);
                        srcOff = T.ldsOff; srcBytes = (size_t)4 * T.ldsPitchDw * (e[1 + k].y - e[1 + k].x + 1);
                        CHECK((size_t)T.ldsOff >= (size_t)C.srcLdsOff);
                        // ping-pong: the band kept by step k does not overlap the band step k reads
                        const size_t rdOff = k == 0 ? (size_t)C.srcLdsOff : (size_t)C.st[k - 1].ldsOff;
                        const size_t rdBytes = (size_t)4 * srcPitchDw * (srcRow1 - srcRow0 + 1);
                        CHECK(srcOff + srcBytes <= rdOff || rdOff + rdBytes <= srcOff);
                    }
                    srcRow0 = e[1 + k].x; srcRow1 = e[1 + k].y;
                }
                if (C.copy0) {
                    const int2 cp = e[C.nSteps + 1];
                    CHECK(cp.x == covered[0] && cp.x >= e[0].x && cp.y <= e[0].y + 1 && cp.x <= cp.y);
                    covered[0] = cp.y;
                }
            }
            for (int k = 0; k < C.nSteps; k++) CHECK(covered[1 + k] == G.L[first + k].h);
            if (C.copy0) CHECK(covered[0] == C.srcH);
            next += C.nSteps;
        }
        CHECK(next == prm.nlevels);
    }
    return 0;
}

int main()
{
    std::mt19937 rng(20261004);
    int planned = 0, total = 0;
    const int fixed[][2] = {{640, 480}, {752, 480}, {1241, 376}, {320, 240}, {333, 257}, {211, 157}, {90, 80}, {4095, 300}, {4200, 100},
                            {100, 900}, {64, 64}, {33, 33}, {1, 1}, {1920, 1080}};
    for (const auto& f : fixed)
        for (int k = 1; k <= 8; k += 2) {
            orb_extractor_params prm = {1000, 1.2f, 8, 20, 7};
            int K[ORB_MAX_LEVELS];
            for (int& v : K) v = k;
            total++;
            if (one(prm, f[1], f[0], K, planned)) return 1;
        }
    for (int t = 0; t < 600; t++) {
        orb_extractor_params prm;
        prm.nfeatures = (int)(rng() % 6000);
        const float sfs[] = {1.05f, 1.1f, 1.2f, 1.25f, 1.5f, 2.0f, 2.5f};
        prm.scale_factor = sfs[rng() % 7];
        prm.nlevels = 1 + (int)(rng() % ORB_MAX_LEVELS);
        prm.ini_th_fast = 20; prm.min_th_fast = 7;
        int K[ORB_MAX_LEVELS];
        for (int& v : K) v = 1 + (int)(rng() % 8);
        const int cols = 1 + (int)(rng() % 2500), rows = 1 + (int)(rng() % 1500);
        total++;
        if (one(prm, rows, cols, K, planned)) return 1;
    }
    printf("asan_geometry: %d configurations, %d inside the envelope, no sanitizer report, all invariants hold\n", total, planned);
    return 0;
}
