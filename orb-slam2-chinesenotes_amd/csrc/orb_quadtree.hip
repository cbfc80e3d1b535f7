// orb_quadtree.hip -- quadtree keypoint distribution on gfx950.
// Reference: ORBextractor::DistributeOctTree src/ORBextractor.cc:562-792, ExtractorNode::DivideNode :436-495.
//
// One 256-thread workgroup per (frame, level); grid (frames, levels), see k_quadtree.
//
// The reference splits std::list nodes and copies key vectors.  Here every candidate carries its
// quadrant path (k_fast_cells), so after ONE sort by key every node of the tree -- at any depth --
// is a contiguous range [lo,hi) of the sorted array and DivideNode is three binary searches on a
// 2-bit digit.  The list evolution (push_front order, erase in place, the "expand the largest
// nodes first" phase with its early break and its (size, creation order) tie-break)
// is replayed exactly, but data-parallel:
//   * a full pass divides every expandable node at once (one thread per node), an exclusive scan
//     of the child counts gives each child its push_front position: new list =
//     reverse(children in creation order) ++ (untouched single-key nodes in old order);
//   * the careful phase sorts (size, seq) with the block bitonic sort, divides ALL candidates
//     in parallel, and a scan of the size gains finds where the reference's `break` (:758) falls.
#include <cstdlib>
#include "orb_quadtree_device.h"

__global__ __launch_bounds__(256) void k_quadtree(const OrbGeom G, unsigned long long* __restrict__ cand,
                                                  size_t candSlab, const int* __restrict__ candCount,
                                                  uint32_t* __restrict__ kpl, int* __restrict__ kpCount,
                                                  int* __restrict__ errFlags, int sortCap, int nodeCap,
                                                  int* __restrict__ ovfBlock)
{
    extern __shared__ unsigned long long qsm[];
    __shared__ int sh[4];                              // root count, two child counters, t* (see qt_body)
    // grid = (frames, levels): workgroups are dealt round-robin over the 8 XCDs in linear order, so with the frame
    // index fastest every XCD gets the same mix of levels (level fastest would send ALL level-0 workgroups, the
    // longest ones, to one XCD), and the big levels are dispatched first (longest-processing-time-first).
    const int level = blockIdx.y, f = blockIdx.x;
    qt_instance_lds(qsm, sh, G, level, f, cand, candSlab, candCount[f * ORB_MAX_LEVELS + level], kpl, kpCount, errFlags, sortCap, nodeCap,
                    ovfBlock);
}

// Slow-but-correct variant for per-level quotas beyond what one workgroup's LDS holds (nFeatures >~ 12 000): the node lists, cut
// points and scan arrays of an instance live in a global scratch slab (60 bytes per node slot); the keys stay in LDS when they
// fit.  Same qt_body: the lists are reached through global pointers, every step a round trip to L2 instead of LDS.
__global__ __launch_bounds__(256) void k_quadtree_gnodes(const OrbGeom G, unsigned long long* __restrict__ cand, size_t candSlab,
                                                         const int* __restrict__ candCount, uint32_t* __restrict__ kpl,
                                                         int* __restrict__ kpCount, int* __restrict__ errFlags, int sortCap, int nodeCap,
                                                         int* __restrict__ ovfBlock, unsigned char* __restrict__ scratchAll,
                                                         size_t scratchStride)
{
    extern __shared__ unsigned long long qsm[];                    // keys[sortCap] | part[258]
    unsigned long long* ldsKeys = qsm;
    int* part = reinterpret_cast<int*>(qsm + sortCap);
    __shared__ int sh[4];
    const int level = blockIdx.y, f = blockIdx.x;
    const OrbLevelGeom& L = G.L[level];
    const int tid = threadIdx.x, T = blockDim.x;
    int n = candCount[f * ORB_MAX_LEVELS + level];
    if (n > L.candCap) n = L.candCap;
    int* outCount = &kpCount[f * ORB_MAX_LEVELS + level];
    if (n == 0) {
        if (tid == 0) *outCount = 0;
        return;
    }
    unsigned char* sc = scratchAll + ((size_t)f * G.nlevels + level) * scratchStride;
    unsigned long long* prevA = reinterpret_cast<unsigned long long*>(sc);
    unsigned long long* prevB = prevA + nodeCap;
    QtNode* A = reinterpret_cast<QtNode*>(prevB + nodeCap);
    QtNode* B = A + nodeCap;
    int3* cuts = reinterpret_cast<int3*>(B + nodeCap);
    int* va = reinterpret_cast<int*>(cuts + nodeCap);
    int* vb = va + nodeCap;
    unsigned long long* gk = cand + (size_t)f * candSlab + L.candBase;
    if (n <= sortCap) {
        for (int i = tid; i < n; i += T) ldsKeys[i] = gk[i];
        qt_body(ldsKeys, n, G, L, f, prevA, prevB, A, B, cuts, va, vb, part, kpl, outCount, errFlags, sh, nullptr, 0, nodeCap);
    } else {
        if (tid == 0) atomicMax(&ovfBlock[1], n);
        qt_body(gk, n, G, L, f, prevA, prevB, A, B, cuts, va, vb, part, kpl, outCount, errFlags, sh, nullptr, 0, nodeCap);
    }
}

// diagnostics: where the stage stamps of k_quadtree go (nullptr: none); see QT_STAMP in orb_quadtree_device.h
int orb_quadtree_set_stamps(unsigned long long* d_stamps, hipStream_t st)
{
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(g_qtStamps), &d_stamps, sizeof(d_stamps), 0, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    return hipStreamSynchronize(st) == hipSuccess ? 0 : -1;
}

size_t orb_quadtree_scratch_stride(int nodeCap) { return ((size_t)nodeCap * (16 + 2 * sizeof(QtNode) + sizeof(int3) + 8) + 15) & ~(size_t)15; }

size_t orb_quadtree_lds_bytes(int sortCap, int nodeCap)
{
    return (size_t)sortCap * 8 + (size_t)nodeCap * (16 + 2 * sizeof(QtNode) + sizeof(int3) + 8) + 258 * 4;
}

void orb_launch_quadtree(hipStream_t st, const OrbGeom& G, unsigned long long* cand, size_t candSlab,
                         const int* candCount, uint32_t* kpl, int* kpCount, int* errFlags, int sortCap,
                         int nodeCap, int nFrames, int* ovfBlock, unsigned char* globalScratch)
{
    if (globalScratch) {
        hipLaunchKernelGGL(k_quadtree_gnodes, dim3(nFrames, G.nlevels), dim3(256), (size_t)sortCap * 8 + 258 * 4, st, G, cand, candSlab,
                           candCount, kpl, kpCount, errFlags, sortCap, nodeCap, ovfBlock, globalScratch, orb_quadtree_scratch_stride(nodeCap));
        return;
    }
    const size_t lds = orb_quadtree_lds_bytes(sortCap, nodeCap);
    // quotas beyond ~1000 per level (nFeatures >~ 4500) need more than the default 64 KB of dynamic LDS: a workgroup may
    // use the CU's whole 160 KB (one workgroup per CU then -- only the huge-quota configurations pay that)
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_quadtree, dim3(nFrames, G.nlevels), dim3(256), lds, st, G, cand, candSlab, candCount,
                       kpl, kpCount, errFlags, sortCap, nodeCap, ovfBlock);
}
