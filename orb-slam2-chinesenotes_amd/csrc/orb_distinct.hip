// orb_distinct.hip -- MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:275-342) for a batch of
// MapPoints on gfx950: one wave64 per MapPoint.  Lane i owns observation i: it computes its row
// of Hamming distances (descriptors staged in LDS), finds the row median by rank counting (no sort: the k-th
// smallest is the value v with #(d < v) <= k < #(d <= v)), and a DPP min-reduction on (median << 16 | i) picks the
// first minimum.  Lists longer than 64 observations are processed in lane-strided passes; lists longer than DD_MAXN (a MapPoint
// seen from more than 256 keyframes: long sessions with revisits) take a second form of the same wave -- the descriptors pass
// through LDS in tiles of DD_MAXN and a lane counts its row's distances into a 257-bin histogram (its row of R), whose running
// sum crosses k at the median -- so the reference's "any N" (src/MapPoint.cc:306-335) holds up to 65 535 observations.
#include <algorithm>

#include "orb_matcher_internal.h"

#define WAVE 64
#define DD_MAXN 256            // observations per MapPoint whose descriptors and distance rows fit LDS at once (a MapPoint rarely has > 100)
#define DD_HARDMAX 65535       // (median << 16 | index) packs the index into 16 bits

static __device__ __forceinline__ unsigned dd_umin_dpp(unsigned v)
{
#define DD_DPP(ctrl, rmask) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(v), ctrl, rmask, 0xf, false))
    DD_DPP(0x111, 0xf); DD_DPP(0x112, 0xf); DD_DPP(0x114, 0xf); DD_DPP(0x118, 0xf); DD_DPP(0x142, 0xa); DD_DPP(0x143, 0xc);
#undef DD_DPP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__global__ __launch_bounds__(WAVE) void k_distinctive(const uint8_t* __restrict__ desc, const int32_t* __restrict__ offsets,
                                                      int nPoints, int32_t* __restrict__ bestIdx, int* __restrict__ err)
{
    __shared__ uint32_t D[DD_MAXN * 8];
    __shared__ uint16_t R[WAVE][DD_MAXN + 2];         // one distance row per lane (pitch +2: odd dword stride)
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= nPoints) return;
    const int b = offsets[p], N = offsets[p + 1] - b;
    if (N <= 0) { if (lane == 0) bestIdx[p] = -1; return; }
    if (N > DD_HARDMAX) { if (lane == 0) { bestIdx[p] = -1; atomicOr(err, 1); } return; }
    if (N > DD_MAXN) {
        const int k = (int)(0.5 * (N - 1));
        unsigned best = 0xFFFFFFFFu;
        for (int base = 0; base < N; base += WAVE) {
            const int i = base + lane;
            uint32_t di[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (i < N) {
#pragma unroll
                for (int w = 0; w < 8; w++) di[w] = reinterpret_cast<const uint32_t*>(desc + (size_t)(b + i) * 32)[w];
            }
            for (int v = 0; v < DD_MAXN + 2; v++) R[lane][v] = 0;
            for (int t0 = 0; t0 < N; t0 += DD_MAXN) {
                const int nt = min(DD_MAXN, N - t0);
                __syncthreads();                               // (one wave: the previous tile's reads are done)
                for (int q = lane; q < nt * 8; q += WAVE) D[q] = reinterpret_cast<const uint32_t*>(desc + (size_t)(b + t0) * 32)[q];
                __syncthreads();
                if (i < N)
                    for (int j = 0; j < nt; j++) {
                        int d = 0;
#pragma unroll
                        for (int w = 0; w < 8; w++) d += __popc(di[w] ^ D[j * 8 + w]);
                        R[lane][d]++;                          // d in 0..256; N <= 65535 fits the 16-bit bin
                    }
            }
            unsigned mine = 0xFFFFFFFFu;
            if (i < N) {
                int cum = 0, median = 256;
                for (int v = 0; v <= 256; v++) {
                    cum += R[lane][v];
                    if (cum > k) { median = v; break; }        // the k-th smallest of the row (:326)
                }
                mine = ((unsigned)median << 16) | (unsigned)i;
            }
            best = min(best, dd_umin_dpp(mine));
        }
        if (lane == 0) bestIdx[p] = (int)(best & 0xFFFFu);
        return;
    }
    for (int i = lane; i < N * 8; i += WAVE) D[i] = reinterpret_cast<const uint32_t*>(desc + (size_t)b * 32)[i];
    __syncthreads();
    const int k = (int)(0.5 * (N - 1));               // index of the median in the sorted row (:326)
    unsigned best = 0xFFFFFFFFu;
    for (int base = 0; base < N; base += WAVE) {
        const int i = base + lane;
        unsigned mine = 0xFFFFFFFFu;
        if (i < N) {
            uint32_t di[8];
#pragma unroll
            for (int w = 0; w < 8; w++) di[w] = D[i * 8 + w];
            for (int j = 0; j < N; j++) {
                int d = 0;
#pragma unroll
                for (int w = 0; w < 8; w++) d += __popc(di[w] ^ D[j * 8 + w]);
                R[lane][j] = (uint16_t)d;              // d(i,i) = 0 as Distances[i][i] = 0 (:314)
            }
            int median = 0;
            for (int j = 0; j < N; j++) {
                const int v = R[lane][j];
                int less = 0, leq = 0;
                for (int t = 0; t < N; t++) {
                    const int u = R[lane][t];
                    less += (u < v);
                    leq += (u <= v);
                }
                if (less <= k && k < leq) { median = v; break; }
            }
            mine = ((unsigned)median << 16) | (unsigned)i;
        }
        best = min(best, dd_umin_dpp(mine));
    }
    if (lane == 0) bestIdx[p] = (int)(best & 0xFFFFu);
}

extern "C" int orb_distinctive_descriptors_device(orb_matcher* m, const uint8_t* d_desc, const int32_t* d_offsets,
                                                  int n_points, int32_t* d_best_idx)
{
    if (!m || n_points < 0) return ORB_ERR_INVALID;
    if (n_points == 0) return ORB_OK;
    if (!d_desc || !d_offsets || !d_best_idx) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(m->device));
    int rc;
    if ((rc = m->nm.ensure(4)) != ORB_OK) return rc;
    ORB_HIP_TRY(hipMemsetAsync(m->nm.p, 0, 4, m->stream));
    hipLaunchKernelGGL(k_distinctive, dim3(n_points), dim3(WAVE), 0, m->stream, d_desc, d_offsets, n_points, d_best_idx,
                       (int*)m->nm.p);
    ORB_HIP_TRY(hipGetLastError());
    return ORB_OK;
}

extern "C" int orb_distinctive_descriptors(orb_matcher* m, const uint8_t* desc, const int32_t* offsets, int n_points,
                                           int32_t* best_idx)
{
    if (!m || n_points < 0) return ORB_ERR_INVALID;
    if (n_points == 0) return ORB_OK;
    if (!desc || !offsets || !best_idx) return ORB_ERR_INVALID;
    const int total = offsets[n_points];
    for (int p = 0; p < n_points; p++)
        if (offsets[p + 1] - offsets[p] > DD_HARDMAX) {
            orb_set_error("a MapPoint with more than %d observations", DD_HARDMAX);
            return ORB_ERR_UNSUPPORTED;
        }
    ORB_HIP_TRY(hipSetDevice(m->device));
    int rc;
    if ((rc = m->stage[0].ensure((size_t)32 * std::max(total, 1))) != ORB_OK ||
        (rc = m->stage[1].ensure((size_t)4 * (n_points + 1))) != ORB_OK || (rc = m->stage[2].ensure((size_t)4 * n_points)) != ORB_OK)
        return rc;
    if (total > 0) ORB_HIP_TRY(hipMemcpyAsync(m->stage[0].p, desc, (size_t)32 * total, hipMemcpyHostToDevice, m->stream));
    ORB_HIP_TRY(hipMemcpyAsync(m->stage[1].p, offsets, (size_t)4 * (n_points + 1), hipMemcpyHostToDevice, m->stream));
    if ((rc = orb_distinctive_descriptors_device(m, (const uint8_t*)m->stage[0].p, (const int32_t*)m->stage[1].p, n_points,
                                                 (int32_t*)m->stage[2].p)) != ORB_OK)
        return rc;
    ORB_HIP_TRY(hipMemcpyAsync(best_idx, m->stage[2].p, (size_t)4 * n_points, hipMemcpyDeviceToHost, m->stream));
    ORB_HIP_TRY(hipStreamSynchronize(m->stream));
    return ORB_OK;
}
