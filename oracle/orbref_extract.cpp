// orbref_extract.cpp -- CPU ORACLE (test infrastructure only; see orbref.hpp header note).
// Restates reference src/ORBextractor.cc:72-171,436-902,1084-1180 plus the OpenCV
// primitives it calls, per SURVEY.md Appendix A.  PARITY UNPINNED at the OpenCV boundary.
#include "orbref.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <list>
#include <utility>

#include "../include/orb_brief_pattern.h"
#include "../include/orb_sincos.h"

namespace orbref {

static const int kPatchSize = 31;       // src/ORBextractor.cc:72
static const int kHalfPatch = 15;       // :73
static const int kEdge = 19;            // :74 EDGE_THRESHOLD

int cvRoundF(float v) { return (int)std::lrint((double)v); }   // default FE_TONEAREST: half-even
int cvRoundD(double v) { return (int)std::lrint(v); }

// ------------------------------------------------------------------ ctor (A.1, :498-559)
Extractor::Extractor(int nf, float sf, int nl, int iniTh, int minTh)
    : nfeatures(nf), nlevels(nl), iniThFAST(iniTh), minThFAST(minTh), scaleFactor(sf)
{
    mvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f;
    mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);   // float*double -> float
        mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    mvInvScaleFactor.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
        mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
        mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    pyramid.resize(nlevels);

    mnFeaturesPerLevel.resize(nlevels);
    float factor = (float)(1.0f / scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        mnFeaturesPerLevel[level] = cvRoundF(nDesired);
        sum += mnFeaturesPerLevel[level];
        nDesired *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);

    umax.resize(kHalfPatch + 1);
    int v, v0;
    int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v <= vmax; ++v) umax[v] = cvRoundD(std::sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

// ------------------------------------------------------------------ resize (A.2)
static short satShort(int v) { return (short)std::min(32767, std::max(-32768, v)); }

ResizeTab resizeTab(int srcLen, int dstLen)
{
    // cv::resize, INTER_LINEAR, 8UC1, generic path: coefficient set-up of resizeGeneric_Linear.
    ResizeTab t;
    t.ofs.resize(dstLen);
    t.c0.resize(dstLen);
    t.c1.resize(dstLen);
    const double invScale = (double)dstLen / srcLen;
    const double scale = 1.0 / invScale;
    for (int d = 0; d < dstLen; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        t.ofs[d] = s;
        float w0 = 1.f - f, w1 = f;
        t.c0[d] = satShort(cvRoundF(w0 * 2048.f));
        t.c1[d] = satShort(cvRoundF(w1 * 2048.f));
    }
    return t;
}

void resizeLinear(const Image& src, Image& dst, int dw, int dh)
{
    ResizeTab tx = resizeTab(src.w, dw), ty = resizeTab(src.h, dh);
    // x axis: out-of-range taps are clamped AND their fraction zeroed; y axis: rows are
    // clipped only (fraction kept) -- SURVEY A.2.
    for (int d = 0; d < dw; d++) {
        if (tx.ofs[d] < 0) { tx.ofs[d] = 0; tx.c0[d] = 2048; tx.c1[d] = 0; }
        if (tx.ofs[d] >= src.w - 1) { tx.ofs[d] = src.w - 1; tx.c0[d] = 2048; tx.c1[d] = 0; }
    }
    dst.w = dw; dst.h = dh;
    dst.px.assign((size_t)dw * dh, 0);
    std::vector<int> h0(dw), h1(dw);
    for (int y = 0; y < dh; y++) {
        int sy0 = std::min(std::max(ty.ofs[y], 0), src.h - 1);
        int sy1 = std::min(std::max(ty.ofs[y] + 1, 0), src.h - 1);
        const uint8_t* r0 = &src.px[(size_t)sy0 * src.w];
        const uint8_t* r1 = &src.px[(size_t)sy1 * src.w];
        for (int x = 0; x < dw; x++) {
            int s = tx.ofs[x];
            int a0 = tx.c0[x], a1 = tx.c1[x];
            int s1 = (s + 1 < src.w) ? s + 1 : s;   // a1 == 0 whenever s+1 is out of range
            h0[x] = r0[s] * a0 + r0[s1] * a1;
            h1[x] = r1[s] * a0 + r1[s1] * a1;
        }
        int b0 = ty.c0[y], b1 = ty.c1[y];
        for (int x = 0; x < dw; x++)
            dst.px[(size_t)y * dw + x] =
                (uint8_t)((((b0 * (h0[x] >> 4)) >> 16) + ((b1 * (h1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

// ------------------------------------------------------------------ pyramid (:1153-1180)
void Extractor::computePyramid(const uint8_t* img, int rows, int cols, size_t stride)
{
    for (int level = 0; level < nlevels; ++level) {
        float scale = mvInvScaleFactor[level];
        int w = cvRoundF((float)cols * scale), h = cvRoundF((float)rows * scale);
        Image& L = pyramid[level];
        if (level == 0) {
            L.w = cols; L.h = rows;
            L.px.resize((size_t)rows * cols);
            for (int y = 0; y < rows; y++) std::memcpy(&L.px[(size_t)y * cols], img + y * stride, cols);
        } else {
            resizeLinear(pyramid[level - 1], L, w, h);
        }
        // the 19-px BORDER_REFLECT_101 frame (:1168-1174) is never read by this path (A.3)
    }
}

// ------------------------------------------------------------------ FAST-9/16 score (A.4)
static const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

int fastScoreV(const uint8_t* p, int pitch)
{
    // V = max over the 16 contiguous 9-arcs of min(+d) and of min(-d), d_k = I(p) - I(ring_k).
    // 9-arc minima via minima of 3 (k,k+1,k+2) combined at k, k+3, k+6 -- same value as the plain
    // 16 x 9 scan, an order of magnitude fewer operations (keeps the cpu_baseline honest).
    int d[16], lo3[16], hi3[16];
    const int c = p[0];
    for (int k = 0; k < 16; k++) d[k] = c - p[kRingDy[k] * pitch + kRingDx[k]];
    for (int k = 0; k < 16; k++) {
        const int a = d[k], b = d[(k + 1) & 15], e = d[(k + 2) & 15];
        lo3[k] = std::min(a, std::min(b, e));
        hi3[k] = std::max(a, std::max(b, e));
    }
    int best = -256;
    for (int k = 0; k < 16; k++) {
        const int mn = std::min(lo3[k], std::min(lo3[(k + 3) & 15], lo3[(k + 6) & 15]));
        const int mx = std::max(hi3[k], std::max(hi3[(k + 3) & 15], hi3[(k + 6) & 15]));
        best = std::max(best, std::max(mn, -mx));
    }
    return best;   // corner at threshold th  <=>  best > th;  cornerScore == best-1
}

// cv::FAST(roi, kps, th, nonmax=true) on one cell ROI; appends (x,y,score) in ROI coordinates,
// ascending y then x (FAST_t + cornerScore<16> of OpenCV, generic path).
static void fastCell(const Image& im, int x0, int y0, int w, int h, int th,
                     std::vector<Candidate>& out)
{
    if (w < 7 || h < 7) return;
    static thread_local std::vector<uint8_t> score, corner;       // per-call scratch, reused
    score.assign((size_t)w * h, 0);
    corner.assign((size_t)w * h, 0);
    const int pitch = im.w;
    const int o0 = kRingDy[0] * pitch + kRingDx[0], o8 = kRingDy[8] * pitch + kRingDx[8];
    const int o4 = kRingDy[4] * pitch + kRingDx[4], o12 = kRingDy[12] * pitch + kRingDx[12];
    const int o2 = kRingDy[2] * pitch + kRingDx[2], o10 = kRingDy[10] * pitch + kRingDx[10];
    const int o6 = kRingDy[6] * pitch + kRingDx[6], o14 = kRingDy[14] * pitch + kRingDx[14];
    for (int y = 3; y < h - 3; y++) {
        const uint8_t* p = &im.px[(size_t)(y0 + y) * pitch + x0 + 3];
        for (int x = 3; x < w - 3; x++, p++) {
            // Cheap exact rejection (what cv::FAST's threshold-table tests amount to): every 9-arc of the
            // 16-ring contains ring pixel k or k+8, for each k; and it is all-darker or all-brighter.
            const int lo = p[0] - th, hi = p[0] + th;
            int a = p[o0], b = p[o8];
            bool dark = a < lo || b < lo, bright = a > hi || b > hi;
            if (!(dark || bright)) continue;
            a = p[o4]; b = p[o12];
            dark = dark && (a < lo || b < lo); bright = bright && (a > hi || b > hi);
            if (!(dark || bright)) continue;
            a = p[o2]; b = p[o10];
            dark = dark && (a < lo || b < lo); bright = bright && (a > hi || b > hi);
            if (!(dark || bright)) continue;
            a = p[o6]; b = p[o14];
            dark = dark && (a < lo || b < lo); bright = bright && (a > hi || b > hi);
            if (!(dark || bright)) continue;
            const int V = fastScoreV(p, pitch);
            if (V > th) {
                score[(size_t)y * w + x] = (uint8_t)(V - 1);
                corner[(size_t)y * w + x] = 1;
            }
        }
    }
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            // only pixels that passed the arc test are considered (their score may be 0 at th==0)
            if (!corner[(size_t)y * w + x]) continue;
            int s = score[(size_t)y * w + x];
            const uint8_t* r = &score[(size_t)y * w + x];
            if (s > r[-1] && s > r[1] && s > r[-w - 1] && s > r[-w] && s > r[-w + 1] &&
                s > r[w - 1] && s > r[w] && s > r[w + 1])
                out.push_back({x, y, s});
        }
}

std::vector<Candidate> Extractor::cellCandidates(int level) const
{
    // src/ORBextractor.cc:799-875
    const Image& im = pyramid[level];
    std::vector<Candidate> all;
    const float W = 30;
    const int minBX = kEdge - 3, minBY = minBX;
    const int maxBX = im.w - kEdge + 3, maxBY = im.h - kEdge + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols <= 0 || nRows <= 0) return all;
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    std::vector<Candidate> cell;
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = (float)maxBY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBX - 6) continue;
            if (maxX > maxBX) maxX = (float)maxBX;
            cell.clear();
            fastCell(im, (int)iniX, (int)iniY, (int)maxX - (int)iniX, (int)maxY - (int)iniY, iniThFAST, cell);
            if (cell.empty())
                fastCell(im, (int)iniX, (int)iniY, (int)maxX - (int)iniX, (int)maxY - (int)iniY, minThFAST, cell);
            for (auto& c : cell) all.push_back({c.x + j * wCell, c.y + i * hCell, c.response});
        }
    }
    return all;
}

// ------------------------------------------------------------------ quadtree (A.6, :436-495, :562-792)
namespace {
struct Node {
    int ulx, uly, urx, ury, blx, bly, brx, bry;
    std::vector<int> keys;             // indices into the candidate array, order preserved
    bool noMore = false;
    long seq = 0;                      // creation sequence: canonical tie-break (A.6)
    std::list<Node>::iterator self;
};

void divide(const Node& p, const std::vector<Candidate>& c, Node& n1, Node& n2, Node& n3, Node& n4)
{
    const int halfX = (int)std::ceil((float)(p.urx - p.ulx) / 2);
    const int halfY = (int)std::ceil((float)(p.bry - p.uly) / 2);
    n1.ulx = p.ulx; n1.uly = p.uly;
    n1.urx = p.ulx + halfX; n1.ury = p.uly;
    n1.blx = p.ulx; n1.bly = p.uly + halfY;
    n1.brx = p.ulx + halfX; n1.bry = p.uly + halfY;
    n2.ulx = n1.urx; n2.uly = n1.ury;
    n2.urx = p.urx; n2.ury = p.ury;
    n2.blx = n1.brx; n2.bly = n1.bry;
    n2.brx = p.urx; n2.bry = p.uly + halfY;
    n3.ulx = n1.blx; n3.uly = n1.bly;
    n3.urx = n1.brx; n3.ury = n1.bry;
    n3.blx = p.blx; n3.bly = p.bly;
    n3.brx = n1.brx; n3.bry = p.bly;
    n4.ulx = n3.urx; n4.uly = n3.ury;
    n4.urx = n2.brx; n4.ury = n2.bry;
    n4.blx = n3.brx; n4.bly = n3.bry;
    n4.brx = p.brx; n4.bry = p.bry;
    for (int k : p.keys) {
        const float x = (float)c[k].x, y = (float)c[k].y;
        if (x < n1.urx) {
            if (y < n1.bry) n1.keys.push_back(k); else n3.keys.push_back(k);
        } else if (y < n1.bry) n2.keys.push_back(k);
        else n4.keys.push_back(k);
    }
    if (n1.keys.size() == 1) n1.noMore = true;
    if (n2.keys.size() == 1) n2.noMore = true;
    if (n3.keys.size() == 1) n3.noMore = true;
    if (n4.keys.size() == 1) n4.noMore = true;
}
}  // namespace

std::vector<Candidate> Extractor::distribute(const std::vector<Candidate>& cands,
                                             int minX, int maxX, int minY, int maxY, int N) const
{
    std::vector<Candidate> result;
    const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));
    if (nIni <= 0) return result;   // reference: UB for portrait images (A.6); defined here as "no keys"
    const float hX = (float)(maxX - minX) / nIni;

    std::list<Node> nodes;
    std::vector<Node*> roots(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; i++) {
        Node n;
        n.ulx = (int)(hX * (float)i); n.uly = 0;
        n.urx = (int)(hX * (float)(i + 1)); n.ury = 0;
        n.blx = n.ulx; n.bly = maxY - minY;
        n.brx = n.urx; n.bry = maxY - minY;
        n.seq = seq++;
        nodes.push_back(n);
        roots[i] = &nodes.back();
    }
    for (int k = 0; k < (int)cands.size(); k++) {
        size_t r = (size_t)((float)cands[k].x / hX);
        assert(r < roots.size());
        roots[r]->keys.push_back(k);
    }
    for (auto it = nodes.begin(); it != nodes.end();) {
        if (it->keys.size() == 1) { it->noMore = true; ++it; }
        else if (it->keys.empty()) it = nodes.erase(it);
        else ++it;
    }

    bool finish = false;
    std::vector<std::pair<int, Node*>> sizeAndNode;
    auto pushChild = [&](Node& ch, int& nToExpand) {
        if (ch.keys.empty()) return;
        ch.seq = seq++;
        nodes.push_front(ch);
        if (ch.keys.size() > 1) {
            nToExpand++;
            sizeAndNode.push_back({(int)ch.keys.size(), &nodes.front()});
            nodes.front().self = nodes.begin();
        }
    };
    auto lessSizeSeq = [](const std::pair<int, Node*>& a, const std::pair<int, Node*>& b) {
        if (a.first != b.first) return a.first < b.first;
        return a.second->seq < b.second->seq;    // canonical replacement of the pointer compare (:711)
    };

    while (!finish) {
        int prevSize = (int)nodes.size();
        int nToExpand = 0;
        sizeAndNode.clear();
        for (auto it = nodes.begin(); it != nodes.end();) {
            if (it->noMore) { ++it; continue; }
            Node n1, n2, n3, n4;
            divide(*it, cands, n1, n2, n3, n4);
            pushChild(n1, nToExpand);
            pushChild(n2, nToExpand);
            pushChild(n3, nToExpand);
            pushChild(n4, nToExpand);
            it = nodes.erase(it);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
            finish = true;
        } else if ((int)nodes.size() + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = (int)nodes.size();
                std::vector<std::pair<int, Node*>> prev = sizeAndNode;
                sizeAndNode.clear();
                std::sort(prev.begin(), prev.end(), lessSizeSeq);
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    int dummy = 0;
                    divide(*prev[j].second, cands, n1, n2, n3, n4);
                    pushChild(n1, dummy);
                    pushChild(n2, dummy);
                    pushChild(n3, dummy);
                    pushChild(n4, dummy);
                    nodes.erase(prev[j].second->self);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
            }
        }
    }

    result.reserve(nodes.size());
    for (auto& n : nodes) {
        int best = n.keys[0];
        float maxResp = (float)cands[best].response;
        for (size_t k = 1; k < n.keys.size(); k++)
            if ((float)cands[n.keys[k]].response > maxResp) {
                best = n.keys[k];
                maxResp = (float)cands[best].response;
            }
        result.push_back(cands[best]);
    }
    return result;
}

// ------------------------------------------------------------------ orientation (A.5, :78-105)
float fastAtan2(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

float Extractor::icAngle(int level, int x, int y) const
{
    const Image& im = pyramid[level];
    const uint8_t* center = &im.px[(size_t)y * im.w + x];
    const int step = im.w;
    int m01 = 0, m10 = 0;
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * center[u];
    for (int v = 1; v <= kHalfPatch; ++v) {
        int vsum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int plus = center[u + v * step], minus = center[u - v * step];
            vsum += (plus - minus);
            m10 += u * (plus + minus);
        }
        m01 += v * vsum;
    }
    return fastAtan2((float)m01, (float)m10);
}

// ------------------------------------------------------------------ Gaussian blur (A.7)
static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * n - 2 - i;
    }
    return i;
}

void gaussianBlur7(const Image& src, Image& dst, const int* taps4)
{
    static const int legacy[4] = {18, 34, 49, 55};
    const int* k = taps4 ? taps4 : legacy;
    // OpenCV 2.4 / 3.0-3.3 8-bit separable path: 8.8 fixed-point taps {18,34,49,55,49,34,18}.
    // Row pass on a REFLECT_101-padded copy of each row, column pass over seven row pointers; the
    // arithmetic (int32 sums, +32768 >> 16, saturate) is unchanged, only the loop structure is CPU-friendly
    // so that the cpu_baseline timing is not inflated by a naive restatement.
    const int w = src.w, h = src.h;
    std::vector<int> rowbuf((size_t)w * h);
    std::vector<uint8_t> pad((size_t)w + 6);
    for (int y = 0; y < h; y++) {
        const uint8_t* s = &src.px[(size_t)y * w];
        for (int i = 0; i < 3; i++) {
            pad[i] = s[reflect101(i - 3, w)];
            pad[w + 3 + i] = s[reflect101(w + i, w)];
        }
        std::memcpy(&pad[3], s, w);
        int* r = &rowbuf[(size_t)y * w];
        const uint8_t* p = pad.data();
        for (int x = 0; x < w; x++)
            r[x] = k[0] * (p[x] + p[x + 6]) + k[1] * (p[x + 1] + p[x + 5]) + k[2] * (p[x + 2] + p[x + 4]) + k[3] * p[x + 3];
    }
    dst.w = w; dst.h = h;
    dst.px.resize((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const int* r[7];
        for (int t = 0; t < 7; t++) r[t] = &rowbuf[(size_t)reflect101(y + t - 3, h) * w];
        uint8_t* d = &dst.px[(size_t)y * w];
        for (int x = 0; x < w; x++) {
            const int acc = k[0] * (r[0][x] + r[6][x]) + k[1] * (r[1][x] + r[5][x]) + k[2] * (r[2][x] + r[4][x]) + k[3] * r[3][x];
            const int v = (acc + 32768) >> 16;
            d[x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
}

// ------------------------------------------------------------------ descriptor (A.8, :120-161)
void Extractor::descriptor(const Image& blurred, int x, int y, float angleDeg, uint8_t out[32]) const
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float angle = (float)angleDeg * factorPI;
    float a, b;
    orb_sincos(angle, &a, &b);
    const uint8_t* center = &blurred.px[(size_t)y * blurred.w + x];
    const int step = blurred.w;
    auto value = [&](int idx) -> int {
        const float px = (float)ORB_BRIEF_PATTERN_XY[2 * idx], py = (float)ORB_BRIEF_PATTERN_XY[2 * idx + 1];
        const float fr = px * b + py * a;     // two roundings + one add; no FMA (-ffp-contract=off)
        const float fc = px * a - py * b;
        return center[cvRoundF(fr) * step + cvRoundF(fc)];
    };
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            int t0 = value(16 * i + 2 * k), t1 = value(16 * i + 2 * k + 1);
            val |= (t0 < t1) << k;
        }
        out[i] = (uint8_t)val;
    }
}

// ------------------------------------------------------------------ operator() (:1084-1150)
void Extractor::extract(const uint8_t* img, int rows, int cols, size_t stride,
                        std::vector<KeyPoint>& kps, std::vector<uint8_t>& desc)
{
    kps.clear();
    desc.clear();
    levelCounts.assign(nlevels, 0);
    levelCandidates.assign(nlevels, 0);
    if (!img || rows <= 0 || cols <= 0) return;
    computePyramid(img, rows, cols, stride);

    std::vector<std::vector<KeyPoint>> all(nlevels);
    for (int level = 0; level < nlevels; ++level) {
        const Image& im = pyramid[level];
        const int minBX = kEdge - 3, minBY = minBX;
        const int maxBX = im.w - kEdge + 3, maxBY = im.h - kEdge + 3;
        std::vector<Candidate> cands = cellCandidates(level);
        // portrait levels (box more than twice as tall as wide): the reference's nIni rounds to 0 and it divides by
        // zero; defined here as "the level yields nothing", candidates included
        if ((int)std::round((float)(maxBX - minBX) / (maxBY - minBY)) <= 0) cands.clear();
        levelCandidates[level] = (int)cands.size();
        std::vector<Candidate> kept = distribute(cands, minBX, maxBX, minBY, maxBY, mnFeaturesPerLevel[level]);
        const int scaledPatch = (int)(kPatchSize * mvScaleFactor[level]);
        for (auto& c : kept) {
            KeyPoint kp;
            kp.x = (float)c.x + minBX;
            kp.y = (float)c.y + minBY;
            kp.size = (float)scaledPatch;
            kp.angle = -1;
            kp.response = (float)c.response;
            kp.octave = level;
            kp.class_id = -1;
            all[level].push_back(kp);
        }
    }
    for (int level = 0; level < nlevels; ++level)
        for (auto& kp : all[level]) kp.angle = icAngle(level, cvRoundF(kp.x), cvRoundF(kp.y));

    for (int level = 0; level < nlevels; ++level) {
        auto& lk = all[level];
        levelCounts[level] = (int)lk.size();
        if (lk.empty()) continue;
        Image blurred;
        gaussianBlur7(pyramid[level], blurred, gaussTaps);
        for (auto& kp : lk) {
            uint8_t d[32];
            descriptor(blurred, cvRoundF(kp.x), cvRoundF(kp.y), kp.angle, d);
            desc.insert(desc.end(), d, d + 32);
        }
        if (level != 0) {
            float scale = mvScaleFactor[level];
            for (auto& kp : lk) { kp.x *= scale; kp.y *= scale; }
        }
        kps.insert(kps.end(), lk.begin(), lk.end());
    }
}

}  // namespace orbref
