// orb_matcher_tri.hip -- ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:1183-1359, with
// CheckDistEpipolarLine :1636-1650) on gfx950.
//
// In the reference as given, vbMatched2 is never set, so every KF1 feature is matched independently: one wave per
// KF1 feature scans the KF2 features 64 at a time; "same vocabulary node" is a compare on the per-feature node
// index (ascending index == the node list's order).  The running `dist > bestDist -> skip` rule means: among the
// candidates that pass every test, the smallest distance wins and on ties the LAST one in list order, i.e. a
// DPP min-reduction on (dist << 16 | 0xFFFF - position).  Rotation histogram + top-3 filter follow in a
// one-workgroup kernel.  The epipole (ex, ey) and F12 come from the caller (cv::Mat arithmetic stays host-side).
#include <algorithm>
#include <vector>

#include "orb_matcher_internal.h"

#pragma clang fp contract(off)

#define WAVE 64
#define TH_LOW 50
#define HISTO_LENGTH 30
#define NODE_NONE 0xFFFFu

struct TriParams {
    float F12[9];
    float ex, ey;
    float scaleFactors2[16];
    float levelSigma2[16];
    int onlyStereo, checkOri;
};

static __device__ __forceinline__ unsigned tr_umin_dpp(unsigned v)
{
#define TR_DPP(ctrl, rmask) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(v), ctrl, rmask, 0xf, false))
    TR_DPP(0x111, 0xf); TR_DPP(0x112, 0xf); TR_DPP(0x114, 0xf); TR_DPP(0x118, 0xf); TR_DPP(0x142, 0xa); TR_DPP(0x143, 0xc);
#undef TR_DPP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__global__ __launch_bounds__(WAVE) void k_tri_match(const orb_keypoint* __restrict__ k1, const uint8_t* __restrict__ d1,
                                                    const uint8_t* __restrict__ mp1, const float* __restrict__ ur1,
                                                    const uint16_t* __restrict__ node1, int n1,
                                                    const orb_keypoint* __restrict__ k2, const uint8_t* __restrict__ d2,
                                                    const uint8_t* __restrict__ mp2, const float* __restrict__ ur2,
                                                    const uint16_t* __restrict__ node2, int n2, const TriParams P,
                                                    int32_t* __restrict__ m12, uint8_t* __restrict__ binOf)
{
    const int idx1 = blockIdx.x, lane = threadIdx.x;
    if (idx1 >= n1) return;
    if (lane == 0) { m12[idx1] = -1; binOf[idx1] = 0xFF; }
    const unsigned nd = node1[idx1];
    if (nd == NODE_NONE || mp1[idx1]) return;                          // not in a common node / already has a MapPoint (:1218)
    const bool stereo1 = ur1 && ur1[idx1] >= 0;
    if (P.onlyStereo && !stereo1) return;
    const orb_keypoint kp1 = k1[idx1];
    // epipolar line l2 = x1' F12 (:1640-1642)
    const float a = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, P.F12[0]), __fmul_rn(kp1.y, P.F12[3])), P.F12[6]);
    const float b = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, P.F12[1]), __fmul_rn(kp1.y, P.F12[4])), P.F12[7]);
    const float c = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, P.F12[2]), __fmul_rn(kp1.y, P.F12[5])), P.F12[8]);
    const float den = __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
    const uint4 lo = reinterpret_cast<const uint4*>(d1 + (size_t)idx1 * 32)[0], hi = reinterpret_cast<const uint4*>(d1 + (size_t)idx1 * 32)[1];
    unsigned best = 0xFFFFFFFFu;
    for (int base = 0; base < n2; base += WAVE) {
        const int j = base + lane;
        unsigned mine = 0xFFFFFFFFu;
        if (j < n2 && node2[j] == nd && !mp2[j]) {                     // same node, no MapPoint (:1239)
            const bool stereo2 = ur2 && ur2[j] >= 0;
            if (!(P.onlyStereo && !stereo2)) {
                const uint4 l2 = reinterpret_cast<const uint4*>(d2 + (size_t)j * 32)[0], h2 = reinterpret_cast<const uint4*>(d2 + (size_t)j * 32)[1];
                const int dist = __popc(lo.x ^ l2.x) + __popc(lo.y ^ l2.y) + __popc(lo.z ^ l2.z) + __popc(lo.w ^ l2.w) +
                                 __popc(hi.x ^ h2.x) + __popc(hi.y ^ h2.y) + __popc(hi.z ^ h2.z) + __popc(hi.w ^ h2.w);
                if (dist <= TH_LOW) {
                    const orb_keypoint kp2 = k2[j];
                    const int oct = min(max(kp2.octave, 0), 15);
                    bool ok = true;
                    if (!stereo1 && !stereo2) {                        // too close to the epipole (:1256-1262)
                        const float dx = __fsub_rn(P.ex, kp2.x), dy = __fsub_rn(P.ey, kp2.y);
                        if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < __fmul_rn(100.0f, P.scaleFactors2[oct])) ok = false;
                    }
                    if (ok) {                                          // CheckDistEpipolarLine (:1644-1649)
                        const float num = __fadd_rn(__fadd_rn(__fmul_rn(a, kp2.x), __fmul_rn(b, kp2.y)), c);
                        if (den == 0) ok = false;
                        else {
                            const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
                            ok = (double)dsqr < 3.84 * (double)P.levelSigma2[oct];
                        }
                    }
                    if (ok) mine = ((unsigned)dist << 16) | (0xFFFFu - (unsigned)j);   // ties: the later candidate wins
                }
            }
        }
        best = min(best, tr_umin_dpp(mine));
    }
    if (best == 0xFFFFFFFFu) return;
    if (lane == 0) {
        const int idx2 = (int)(0xFFFFu - (best & 0xFFFFu));
        m12[idx1] = idx2;
        if (P.checkOri) {
            float rot = __fsub_rn(kp1.angle, k2[idx2].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
            if (bin == HISTO_LENGTH) bin = 0;
            binOf[idx1] = (uint8_t)bin;
        }
    }
}

__global__ __launch_bounds__(1024) void k_tri_filter(int32_t* __restrict__ m12, const uint8_t* __restrict__ binOf, int n1,
                                                     int checkOri, int32_t* __restrict__ nmatchesOut)
{
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int keep[3];
    __shared__ int nm;
    if (threadIdx.x < HISTO_LENGTH) hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) nm = 0;
    __syncthreads();
    int local = 0;
    for (int i = threadIdx.x; i < n1; i += blockDim.x)
        if (m12[i] >= 0) {
            local++;
            if (checkOri) atomicAdd(&hist[binOf[i]], 1);
        }
    if (local) atomicAdd(&nm, local);
    __syncthreads();
    if (checkOri) {
        if (threadIdx.x == 0) {
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < HISTO_LENGTH; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
                else if (s > max3) { max3 = s; i3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { i3 = -1; }
            keep[0] = i1; keep[1] = i2; keep[2] = i3;
        }
        __syncthreads();
        int dropped = 0;
        for (int i = threadIdx.x; i < n1; i += blockDim.x)
            if (m12[i] >= 0) {
                const int bb = binOf[i];
                if (bb != keep[0] && bb != keep[1] && bb != keep[2]) { m12[i] = -1; dropped++; }
            }
        if (dropped) atomicSub(&nm, dropped);
        __syncthreads();
    }
    if (threadIdx.x == 0) *nmatchesOut = nm;
}

// common vocabulary nodes of the two CSR feature vectors -> compact per-feature node index (NODE_NONE elsewhere)
static int common_nodes(const orb_featvec* fa, int na, const orb_featvec* fb, int nb, std::vector<uint16_t>& nodeA,
                        std::vector<uint16_t>& nodeB)
{
    nodeA.assign(std::max(na, 1), NODE_NONE);
    nodeB.assign(std::max(nb, 1), NODE_NONE);
    int a = 0, b = 0, k = 0;
    while (a < fa->n_nodes && b < fb->n_nodes) {
        if (fa->node_ids[a] == fb->node_ids[b]) {
            if (k >= 65534) return ORB_ERR_UNSUPPORTED;
            for (int p = fa->offsets[a]; p < fa->offsets[a + 1]; p++) {
                if (fa->indices[p] < 0 || fa->indices[p] >= na || (p > fa->offsets[a] && fa->indices[p] <= fa->indices[p - 1])) return ORB_ERR_UNSUPPORTED;
                nodeA[fa->indices[p]] = (uint16_t)k;
            }
            for (int p = fb->offsets[b]; p < fb->offsets[b + 1]; p++) {
                if (fb->indices[p] < 0 || fb->indices[p] >= nb || (p > fb->offsets[b] && fb->indices[p] <= fb->indices[p - 1])) return ORB_ERR_UNSUPPORTED;
                nodeB[fb->indices[p]] = (uint16_t)k;
            }
            k++; a++; b++;
        } else if (fa->node_ids[a] < fb->node_ids[b]) a++;
        else b++;
    }
    return ORB_OK;
}

extern "C" int orb_match_triangulation(orb_matcher* m, const orb_keypoint* kps1, const uint8_t* desc1, const uint8_t* has_mp1,
                                       const float* u_right1, int n1, const orb_featvec* fv1, const orb_keypoint* kps2,
                                       const uint8_t* desc2, const uint8_t* has_mp2, const float* u_right2, int n2,
                                       const orb_featvec* fv2, const float* F12, float ex, float ey,
                                       const float* scale_factors2, const float* level_sigma2_2, int n_levels, int only_stereo,
                                       int check_ori, int32_t* match_12, int* nmatches)
{
    if (!m || n1 < 0 || n2 < 0 || !nmatches || !fv1 || !fv2 || !F12 || !scale_factors2 || !level_sigma2_2) return ORB_ERR_INVALID;
    *nmatches = 0;
    if (n1 > 0 && !match_12) return ORB_ERR_INVALID;
    for (int i = 0; i < n1; i++) match_12[i] = -1;
    if (n1 == 0 || n2 == 0) return ORB_OK;
    if (!kps1 || !desc1 || !has_mp1 || !kps2 || !desc2 || !has_mp2) return ORB_ERR_INVALID;
    if (n1 > 65534 || n2 > 65534 || n_levels < 1 || n_levels > 16) return ORB_ERR_UNSUPPORTED;
    std::vector<uint16_t> nodeA, nodeB;
    int rc = common_nodes(fv1, n1, fv2, n2, nodeA, nodeB);
    if (rc != ORB_OK) { orb_set_error("feature-vector indices must be in range and ascending inside a node"); return rc; }
    ORB_HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    MBuf* buf = m->init;
    const size_t sz[12] = {sizeof(orb_keypoint) * (size_t)n1, (size_t)32 * n1, (size_t)n1, (size_t)4 * n1, (size_t)2 * n1,
                           sizeof(orb_keypoint) * (size_t)n2, (size_t)32 * n2, (size_t)n2, (size_t)4 * n2, (size_t)2 * n2,
                           (size_t)4 * n1 + 4, (size_t)n1};
    const void* src[10] = {kps1, desc1, has_mp1, u_right1, nodeA.data(), kps2, desc2, has_mp2, u_right2, nodeB.data()};
    for (int i = 0; i < 12; i++)
        if ((rc = buf[i].ensure(sz[i])) != ORB_OK) return rc;
    for (int i = 0; i < 10; i++)
        if (src[i]) ORB_HIP_TRY(hipMemcpyAsync(buf[i].p, src[i], sz[i], hipMemcpyHostToDevice, st));
    TriParams P;
    for (int i = 0; i < 9; i++) P.F12[i] = F12[i];
    P.ex = ex; P.ey = ey;
    for (int i = 0; i < 16; i++) {
        P.scaleFactors2[i] = i < n_levels ? scale_factors2[i] : 1.0f;
        P.levelSigma2[i] = i < n_levels ? level_sigma2_2[i] : 1.0f;
    }
    P.onlyStereo = only_stereo; P.checkOri = check_ori;
    int32_t* dM = (int32_t*)buf[10].p;
    hipLaunchKernelGGL(k_tri_match, dim3(n1), dim3(WAVE), 0, st, (const orb_keypoint*)buf[0].p, (const uint8_t*)buf[1].p,
                       (const uint8_t*)buf[2].p, u_right1 ? (const float*)buf[3].p : nullptr, (const uint16_t*)buf[4].p, n1,
                       (const orb_keypoint*)buf[5].p, (const uint8_t*)buf[6].p, (const uint8_t*)buf[7].p,
                       u_right2 ? (const float*)buf[8].p : nullptr, (const uint16_t*)buf[9].p, n2, P, dM, (uint8_t*)buf[11].p);
    hipLaunchKernelGGL(k_tri_filter, dim3(1), dim3(1024), 0, st, dM, (const uint8_t*)buf[11].p, n1, check_ori, dM + n1);
    ORB_HIP_TRY(hipGetLastError());
    std::vector<int32_t> host((size_t)n1 + 1);
    ORB_HIP_TRY(hipMemcpyAsync(host.data(), dM, ((size_t)n1 + 1) * 4, hipMemcpyDeviceToHost, st));
    ORB_HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < n1; i++) match_12[i] = host[i];
    *nmatches = host[n1];
    return ORB_OK;
}
