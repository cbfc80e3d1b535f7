// orbref_capi.cpp -- flat C entry points over the CPU ORACLE so tests/ and bench.py's
// cpu_baseline leg can drive it through ctypes.  TEST INFRASTRUCTURE ONLY (see orbref.hpp).
#include <cstring>

#include "../include/orb_sincos.h"
#include "orbref.hpp"

using namespace orbref;

static FeatVec makeFV(const uint32_t* ids, const int32_t* offs, const int32_t* idx, int nn)
{
    FeatVec fv;
    fv.nodeIds.assign(ids, ids + nn);
    fv.offsets.assign(offs, offs + nn + 1);
    fv.indices.assign(idx, idx + offs[nn]);
    return fv;
}

extern "C" {

void* orbref_create(int nf, float sf, int nl, int iniTh, int minTh) { return new Extractor(nf, sf, nl, iniTh, minTh); }
void orbref_set_gaussian(void* h, const int* taps4) { for (int i = 0; i < 4; i++) static_cast<Extractor*>(h)->gaussTaps[i] = taps4[i]; }
void orbref_destroy(void* h) { delete (Extractor*)h; }

// returns the number of keypoints (also when it exceeds cap; then nothing is copied)
int orbref_extract(void* h, const uint8_t* img, int rows, int cols, size_t stride,
                   KeyPoint* kps, uint8_t* desc, int cap)
{
    Extractor* e = (Extractor*)h;
    std::vector<KeyPoint> k;
    std::vector<uint8_t> d;
    e->extract(img, rows, cols, stride, k, d);
    if ((int)k.size() <= cap) {
        if (!k.empty()) {
            std::memcpy(kps, k.data(), k.size() * sizeof(KeyPoint));
            std::memcpy(desc, d.data(), d.size());
        }
    }
    return (int)k.size();
}

void orbref_tables(void* h, float* scale, float* invScale, float* sigma2, float* invSigma2, int* quota, int* umax16)
{
    Extractor* e = (Extractor*)h;
    for (int i = 0; i < e->nlevels; i++) {
        scale[i] = e->mvScaleFactor[i];
        invScale[i] = e->mvInvScaleFactor[i];
        sigma2[i] = e->mvLevelSigma2[i];
        invSigma2[i] = e->mvInvLevelSigma2[i];
        quota[i] = e->mnFeaturesPerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
}

void orbref_compute_pyramid(void* h, const uint8_t* img, int rows, int cols, size_t stride)
{
    ((Extractor*)h)->computePyramid(img, rows, cols, stride);
}
void orbref_pyramid_dims(void* h, int level, int* w, int* hh)
{
    Extractor* e = (Extractor*)h;
    *w = e->pyramid[level].w;
    *hh = e->pyramid[level].h;
}
void orbref_pyramid_copy(void* h, int level, uint8_t* dst)
{
    Extractor* e = (Extractor*)h;
    std::memcpy(dst, e->pyramid[level].px.data(), e->pyramid[level].px.size());
}
void orbref_level_counts(void* h, int* kept, int* cands)
{
    Extractor* e = (Extractor*)h;
    for (int i = 0; i < e->nlevels; i++) {
        kept[i] = e->levelCounts.empty() ? 0 : e->levelCounts[i];
        cands[i] = e->levelCandidates.empty() ? 0 : e->levelCandidates[i];
    }
}

// stage: FAST candidates of a level of the current pyramid, (x,y,response) triplets rel. (16,16)
int orbref_cell_candidates(void* h, int level, int32_t* xyz, int cap)
{
    std::vector<Candidate> c = ((Extractor*)h)->cellCandidates(level);
    if ((int)c.size() <= cap)
        for (size_t i = 0; i < c.size(); i++) { xyz[3 * i] = c[i].x; xyz[3 * i + 1] = c[i].y; xyz[3 * i + 2] = c[i].response; }
    return (int)c.size();
}

int orbref_distribute(void* h, const int32_t* xyz, int n, int minX, int maxX, int minY, int maxY, int N,
                      int32_t* out, int cap)
{
    std::vector<Candidate> c(n);
    for (int i = 0; i < n; i++) c[i] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    std::vector<Candidate> r = ((Extractor*)h)->distribute(c, minX, maxX, minY, maxY, N);
    if ((int)r.size() <= cap)
        for (size_t i = 0; i < r.size(); i++) { out[3 * i] = r[i].x; out[3 * i + 1] = r[i].y; out[3 * i + 2] = r[i].response; }
    return (int)r.size();
}

float orbref_ic_angle(void* h, int level, int x, int y) { return ((Extractor*)h)->icAngle(level, x, y); }

void orbref_descriptor(void* h, const uint8_t* blurred, int w, int hh, int x, int y, float angle, uint8_t* out32)
{
    Image im;
    im.w = w; im.h = hh;
    im.px.assign(blurred, blurred + (size_t)w * hh);
    ((Extractor*)h)->descriptor(im, x, y, angle, out32);
}

void orbref_resize(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
    Image s, d;
    s.w = sw; s.h = sh;
    s.px.assign(src, src + (size_t)sw * sh);
    resizeLinear(s, d, dw, dh);
    std::memcpy(dst, d.px.data(), d.px.size());
}

void orbref_resize_tab(int srcLen, int dstLen, int32_t* ofs, int16_t* c0, int16_t* c1)
{
    ResizeTab t = resizeTab(srcLen, dstLen);
    for (int i = 0; i < dstLen; i++) { ofs[i] = t.ofs[i]; c0[i] = t.c0[i]; c1[i] = t.c1[i]; }
}

void orbref_blur(const uint8_t* src, int w, int h, uint8_t* dst)
{
    Image s, d;
    s.w = w; s.h = h;
    s.px.assign(src, src + (size_t)w * h);
    gaussianBlur7(s, d);
    std::memcpy(dst, d.px.data(), d.px.size());
}

void orbref_blur_taps(const uint8_t* src, int w, int h, uint8_t* dst, const int* taps4)
{
    Image s, d;
    s.w = w; s.h = h;
    s.px.assign(src, src + (size_t)w * h);
    gaussianBlur7(s, d, taps4);
    std::memcpy(dst, d.px.data(), d.px.size());
}

// dense V map (int16) over the interior [3,w-3)x[3,h-3); rim = -256
void orbref_fast_vmap(const uint8_t* img, int w, int h, int16_t* out)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            out[(size_t)y * w + x] = (x < 3 || y < 3 || x >= w - 3 || y >= h - 3)
                                         ? (int16_t)-256
                                         : (int16_t)fastScoreV(img + (size_t)y * w + x, w);
}

float orbref_fast_atan2(float y, float x) { return fastAtan2(y, x); }
void orbref_sincos(float x, float* c, float* s) { orb_sincos(x, c, s); }
int orbref_cvround(float v) { return cvRoundF(v); }

int orbref_hamming(const uint8_t* a, const uint8_t* b) { return hamming256(a, b); }
void orbref_three_maxima(const int* counts30, int* out3)
{
    int a = -1, b = -1, c = -1;
    threeMaxima(counts30, a, b, c);
    out3[0] = a; out3[1] = b; out3[2] = c;
}

// returns number of nodes; node_ids/offsets need room for 101(+1) entries, indices for n
int orbref_bow_transform(const uint8_t* desc, int n, const uint8_t* centroids,
                         uint32_t* nodeIds, int32_t* offsets, int32_t* indices)
{
    FeatVec fv = bowTransform(desc, n, centroids);
    std::memcpy(nodeIds, fv.nodeIds.data(), fv.nodeIds.size() * 4);
    std::memcpy(offsets, fv.offsets.data(), fv.offsets.size() * 4);
    if (n) std::memcpy(indices, fv.indices.data(), fv.indices.size() * 4);
    return (int)fv.nodeIds.size();
}

int orbref_search_by_bow(const uint8_t* descKF, const float* angleKF, const uint8_t* validKF,
                         const uint32_t* idsKF, const int32_t* offsKF, const int32_t* idxKF, int nnKF,
                         const uint8_t* descF, const float* angleF, int nF,
                         const uint32_t* idsF, const int32_t* offsF, const int32_t* idxF, int nnF,
                         float ratio, int checkOri, int32_t* outF)
{
    std::vector<int32_t> out;
    int nm = searchByBoW(descKF, angleKF, validKF, makeFV(idsKF, offsKF, idxKF, nnKF),
                         descF, angleF, nF, makeFV(idsF, offsF, idxF, nnF), ratio, checkOri != 0, out);
    if (nF) std::memcpy(outF, out.data(), (size_t)nF * 4);
    return nm;
}

int orbref_search_by_bow_kk(const uint8_t* d1, const float* a1, const uint8_t* v1, int n1,
                            const uint32_t* ids1, const int32_t* offs1, const int32_t* idx1, int nn1,
                            const uint8_t* d2, const float* a2, const uint8_t* v2, int n2,
                            const uint32_t* ids2, const int32_t* offs2, const int32_t* idx2, int nn2,
                            float ratio, int checkOri, int32_t* out12)
{
    std::vector<int32_t> out;
    int nm = searchByBoWKK(d1, a1, v1, n1, makeFV(ids1, offs1, idx1, nn1),
                           d2, a2, v2, n2, makeFV(ids2, offs2, idx2, nn2), ratio, checkOri != 0, out);
    if (n1) std::memcpy(out12, out.data(), (size_t)n1 * 4);
    return nm;
}

int orbref_search_for_init(const KeyPoint* k1, const uint8_t* d1, int n1,
                           const KeyPoint* k2, const uint8_t* d2, int n2,
                           float minX, float minY, float invW, float invH,
                           float* prevXY, int window, float ratio, int checkOri, int32_t* m12)
{
    FrameGrid g;
    g.minX = minX; g.minY = minY; g.invW = invW; g.invH = invH;
    g.assign(k2, n2);
    std::vector<int32_t> out;
    int nm = searchForInitialization(k1, d1, n1, k2, d2, n2, g, prevXY, window, ratio, checkOri != 0, out);
    if (n1) std::memcpy(m12, out.data(), (size_t)n1 * 4);
    return nm;
}

int orbref_search_by_projection(int mode, const ProjQuery* q, const uint8_t* qDesc, const float* qAngle, int nq,
                                const KeyPoint* kps, const uint8_t* desc, const float* uRight, const uint8_t* occupied,
                                int n, float minX, float minY, float invW, float invH, float ratio, int maxDist,
                                int checkOri, int32_t* matchCur)
{
    FrameGrid g;
    g.minX = minX; g.minY = minY; g.invW = invW; g.invH = invH;
    g.assign(kps, n);
    std::vector<int32_t> out;
    const int nm = mode == 0 ? searchByProjectionLast(q, qDesc, qAngle, nq, kps, desc, uRight, occupied, n, g, maxDist, checkOri != 0, out)
                             : searchByProjectionMap(q, qDesc, nq, kps, desc, uRight, occupied, n, g, ratio, maxDist, out);
    if (n) std::memcpy(matchCur, out.data(), (size_t)n * 4);
    return nm;
}

void orbref_vocab_transform(const uint8_t* nodeDesc, const int32_t* childBegin, const int32_t* children,
                            const int32_t* wordId, int nNodes, int L, const uint8_t* desc, int n, int levelsup,
                            int32_t* wordOf, int32_t* nodeOf)
{
    VocabTree t{nodeDesc, childBegin, children, wordId, nNodes, L};
    vocabTransform(t, desc, n, levelsup, wordOf, nodeOf);
}

void orbref_distinctive(const uint8_t* desc, const int32_t* offsets, int nPoints, int32_t* bestIdx)
{
    distinctiveDescriptors(desc, offsets, nPoints, bestIdx);
}

void orbref_search_by_projection_best(const ProjQuery* q, const uint8_t* qDesc, int nq, const KeyPoint* kps,
                                      const uint8_t* desc, const float* uRight, int n, float minX, float minY, float invW,
                                      float invH, int maxDist, int chi2, const float* invSigma2, int32_t* bestIdx,
                                      int32_t* bestDist)
{
    FrameGrid g;
    g.minX = minX; g.minY = minY; g.invW = invW; g.invH = invH;
    g.assign(kps, n);
    searchByProjectionBest(q, qDesc, nq, kps, desc, uRight, n, g, maxDist, chi2 != 0, invSigma2, bestIdx, bestDist);
}

int orbref_search_for_triangulation(const KeyPoint* k1, const uint8_t* d1, const uint8_t* mp1, const float* ur1, int n1,
                                    const uint32_t* ids1, const int32_t* offs1, const int32_t* idx1, int nn1,
                                    const KeyPoint* k2, const uint8_t* d2, const uint8_t* mp2, const float* ur2, int n2,
                                    const uint32_t* ids2, const int32_t* offs2, const int32_t* idx2, int nn2,
                                    const float* F12, float ex, float ey, const float* sf2, const float* sig2,
                                    int onlyStereo, int checkOri, int32_t* m12)
{
    std::vector<int32_t> out;
    const int nm = searchForTriangulation(k1, d1, mp1, ur1, n1, makeFV(ids1, offs1, idx1, nn1), k2, d2, mp2, ur2, n2,
                                          makeFV(ids2, offs2, idx2, nn2), F12, ex, ey, sf2, sig2, onlyStereo != 0,
                                          checkOri != 0, out);
    if (n1) std::memcpy(m12, out.data(), (size_t)n1 * 4);
    return nm;
}

// grid query exposed for the grid unit tests: returns count, indices in reference order
int orbref_features_in_area(const KeyPoint* k, int n, float minX, float minY, float invW, float invH,
                            float x, float y, float r, int minLevel, int maxLevel, int32_t* out, int cap)
{
    FrameGrid g;
    g.minX = minX; g.minY = minY; g.invW = invW; g.invH = invH;
    g.assign(k, n);
    std::vector<int32_t> v = g.inArea(k, x, y, r, minLevel, maxLevel);
    if ((int)v.size() <= cap && !v.empty()) std::memcpy(out, v.data(), v.size() * 4);
    return (int)v.size();
}

}  // extern "C"

// Pin of include/orb_sincos.h: walks the float bit patterns [loBits, hiBits] of the ANGLE IN DEGREES with
// the given stride, converts to radians exactly as computeOrbDescriptor does (src/ORBextractor.cc:124), and
// counts inputs where cos or sin differs from
//   mode 0: the correctly rounded value, taken as (float)cos((double)rad) / (float)sin((double)rad)
//   mode 1: the host libm's cosf/sinf (what the reference itself calls at :125; glibc documents <= 1 ulp,
//           i.e. it is NOT always correctly rounded)
#include <cmath>
extern "C" long orbref_sincos_sweep(uint32_t loBits, uint32_t hiBits, uint32_t stride, int mode, uint32_t* firstBad)
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    long bad = 0;
    for (uint64_t b = loBits; b <= hiBits; b += stride) {
        uint32_t bits = (uint32_t)b;
        float deg;
        std::memcpy(&deg, &bits, 4);
        const float rad = deg * factorPI;
        float c, s;
        orb_sincos(rad, &c, &s);
        const float rc = mode ? cosf(rad) : (float)cos((double)rad);
        const float rs = mode ? sinf(rad) : (float)sin((double)rad);
        if (c != rc || s != rs) {
            if (bad == 0 && firstBad) *firstBad = bits;
            bad++;
        }
    }
    return bad;
}
