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
#include "orb_block_sort.h"
#include "orb_kernels.h"
#include "orb_wave.h"

struct QtNode {
    int lo, hi;      // key range in the sorted candidate array
    int depth;       // number of path digits already consumed
};

__device__ __forceinline__ unsigned qt_digit(unsigned long long k, int depth)
{
    return (unsigned)(k >> (ORB_KEY_PATH_SHIFT + 2 * (ORB_KEY_PATH_LEVELS - 1 - depth))) & 3u;
}

// first index in [lo,hi) whose digit at `depth` is >= q
__device__ __forceinline__ int qt_lower(const unsigned long long* keys, int lo, int hi, int depth, unsigned q)
{
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (qt_digit(keys[mid], depth) < q) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// DivideNode: the three interior cut points of a node's key range
__device__ __forceinline__ int3 qt_cuts(const unsigned long long* keys, const QtNode nd)
{
    int3 c;
    if (nd.depth >= ORB_KEY_PATH_LEVELS) {            // unreachable inside the supported image envelope
        c.x = c.y = c.z = nd.hi;
    } else {
        c.y = qt_lower(keys, nd.lo, nd.hi, nd.depth, 2);
        c.x = qt_lower(keys, nd.lo, c.y, nd.depth, 1);
        c.z = qt_lower(keys, c.y, nd.hi, nd.depth, 3);
    }
    return c;
}

// in-place exclusive scan of a[0..n) by the whole block; returns the total.  part = int[blockDim.x+1].
__device__ int qt_scan(int* a, int n, int* part)
{
    const int T = blockDim.x, t = threadIdx.x;
    const int C = (n + T - 1) / T;
    const int b = min(t * C, n), e = min(b + C, n);
    int s = 0;
    for (int i = b; i < e; i++) s += a[i];
    part[t] = s;
    __syncthreads();
    if (t < 64) {                                      // T == 256: each of 64 lanes owns 4 partials
        const int q0 = part[4 * t], q1 = part[4 * t + 1], q2 = part[4 * t + 2], q3 = part[4 * t + 3];
        const int mine = q0 + q1 + q2 + q3;
        const int incl = orb_wave_scan_incl(mine);
        const int ex = incl - mine;
        part[4 * t] = ex; part[4 * t + 1] = ex + q0; part[4 * t + 2] = ex + q0 + q1; part[4 * t + 3] = ex + q0 + q1 + q2;
        if (t == 63) part[T] = incl;
    }
    __syncthreads();
    int run = part[t];
    for (int i = b; i < e; i++) { const int v = a[i]; a[i] = run; run += v; }
    const int total = part[T];
    __syncthreads();
    return total;
}

// Everything after the keys are in place.  Force-inlined into both call sites so that the compiler knows the
// address space of `keys` (LDS: ds_* instructions; a runtime-selected generic pointer would turn every access
// of the sort and of the binary searches into slow flat_* operations).
__device__ __forceinline__ void qt_body(unsigned long long* keys, int n, const OrbGeom& G, const OrbLevelGeom& L, int f,
                                        unsigned long long* prevA, unsigned long long* prevB, QtNode* A, QtNode* B,
                                        int3* cuts, int* va, int* vb, int* part, uint32_t* __restrict__ kpl,
                                        int* outCount, int* __restrict__ errFlags, int* sh)
{
    int& sh_size = sh[0]; int& sh_prevCount = sh[1]; int& sh_state = sh[2]; int& sh_inB = sh[3];
    int& sh_prevInB = sh[4]; int& sh_tstar = sh[5];
    const int tid = threadIdx.x, T = blockDim.x;
    __syncthreads();
    orb_block_sort(keys, n);

    const int N = L.quota;
    // ---- roots (reference :575-612): empty roots vanish, single-key roots are bNoMore
    if (tid == 0) {
        int cnt = 0, lo = 0;
        for (int r = 0; r < L.nIni; r++) {
            int a = lo, b = n;                        // first key whose root is > r
            while (a < b) {
                const int mid = (a + b) >> 1;
                if ((int)(keys[mid] >> ORB_KEY_ROOT_SHIFT) <= r) a = mid + 1; else b = mid;
            }
            if (a > lo) { QtNode nd; nd.lo = lo; nd.hi = a; nd.depth = 0; A[cnt++] = nd; }
            lo = a;
        }
        sh_size = cnt;
        sh_state = 0;
        sh_inB = 0;
        sh_prevInB = 0;
        sh_prevCount = 0;
    }
    __syncthreads();

    while (true) {
        const int state = sh_state;
        const int size0 = sh_size;
        const int pc = sh_prevCount;
        QtNode* cur = sh_inB ? B : A;
        QtNode* nxt = sh_inB ? A : B;
        unsigned long long* prev = sh_prevInB ? prevB : prevA;
        unsigned long long* prevNew = sh_prevInB ? prevA : prevB;
        __syncthreads();                               // everyone holds the loop state before it is rewritten
        if (state == 2) break;
        if (state == 1 && pc == 0) {                   // nothing left to expand: size cannot change (:762)
            if (tid == 0) sh_state = 2;
            __syncthreads();
            continue;
        }

        if (state == 0) {
            // ---------------- one full pass over the list (:631-691), all nodes at once
            for (int i = tid; i < size0; i += T) {
                const QtNode nd = cur[i];
                if (nd.hi - nd.lo == 1) {              // bNoMore: stays where it is
                    va[i] = 1 << 16;                   // packed scan value: keep count in the high half
                } else {
                    const int3 c = qt_cuts(keys, nd);
                    cuts[i] = c;
                    va[i] = (c.x > nd.lo) + (c.y > c.x) + (c.z > c.y) + (nd.hi > c.z);
                }
            }
            if (tid == 0) sh_prevCount = 0;
            __syncthreads();
            const int tot = qt_scan(va, size0, part);
            const int sTot = tot & 0xFFFF, kTot = tot >> 16;
            for (int i = tid; i < size0; i += T) {
                const QtNode nd = cur[i];
                const int off = va[i];
                if (nd.hi - nd.lo == 1) {
                    nxt[sTot + (off >> 16)] = nd;
                } else {
                    const int3 c = cuts[i];
                    const int edge[5] = {nd.lo, c.x, c.y, c.z, nd.hi};
                    int j = off & 0xFFFF;              // creation sequence number of the next child
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int a = edge[q], b = edge[q + 1];
                        if (b > a) {
                            const int pos = sTot - 1 - j;      // push_front: later children end up in front
                            QtNode ch; ch.lo = a; ch.hi = b; ch.depth = nd.depth + 1;
                            nxt[pos] = ch;
                            if (b - a > 1) {
                                const int slot = atomicAdd(&sh_prevCount, 1);
                                prev[slot] = ((unsigned long long)(b - a) << 48) | ((unsigned long long)j << 24) |
                                             (unsigned long long)pos;
                            }
                            j++;
                        }
                    }
                }
            }
            __syncthreads();
            if (tid == 0) {
                const int size = sTot + kTot;
                sh_size = size;
                sh_inB ^= 1;
                if (size >= N || size == size0) sh_state = 2;                 // :695
                else if (size + 3 * sh_prevCount > N) sh_state = 1;           // :701
            }
            __syncthreads();
        } else {
            // ---------------- careful phase (:703-765): largest first, stop as soon as size >= N
            orb_block_sort(prev, pc);                    // ascending (size, seq); processed from the back (:711-713)
            for (int t = tid; t < pc; t += T) {
                const int idx = (int)(prev[pc - 1 - t] & 0xFFFFFF);
                const QtNode nd = cur[idx];
                const int3 c = qt_cuts(keys, nd);
                cuts[t] = c;
                va[t] = (c.x > nd.lo) + (c.y > c.x) + (c.z > c.y) + (nd.hi > c.z);   // children of candidate t
            }
            for (int i = tid; i < size0; i += T) vb[i] = 1;                   // alive flags of the current list
            if (tid == 0) sh_tstar = pc;
            __syncthreads();
            const int sAll = qt_scan(va, pc, part);    // va[t] = children created before candidate t
            (void)sAll;
            // size after candidate t has been divided = size0 + (va[t] + children(t)) - (t + 1)
            for (int t = tid; t < pc; t += T) {
                const int idx = (int)(prev[pc - 1 - t] & 0xFFFFFF);
                const QtNode nd = cur[idx];
                const int3 c = cuts[t];
                const int ch = (c.x > nd.lo) + (c.y > c.x) + (c.z > c.y) + (nd.hi > c.z);
                if (size0 + va[t] + ch - (t + 1) >= N) atomicMin(&sh_tstar, t);
            }
            __syncthreads();
            const int P = min(pc, sh_tstar + 1);       // candidates actually divided before the break (:758)
            __syncthreads();
            for (int t = tid; t < P; t += T) vb[(int)(prev[pc - 1 - t] & 0xFFFFFF)] = 0;     // erased parents
            if (tid == 0) sh_prevCount = 0;
            __syncthreads();
            // children of the first P candidates
            int sTot;
            {
                const int idxLast = (int)(prev[pc - P] & 0xFFFFFF);          // candidate t = P-1
                const QtNode nd = cur[idxLast];
                const int3 c = cuts[P - 1];
                sTot = va[P - 1] + (c.x > nd.lo) + (c.y > c.x) + (c.z > c.y) + (nd.hi > c.z);
            }
            const int kTot = qt_scan(vb, size0, part); // vb[i] = position of alive node i among the alive ones
            for (int t = tid; t < P; t += T) {
                const int idx = (int)(prev[pc - 1 - t] & 0xFFFFFF);
                const QtNode nd = cur[idx];
                const int3 c = cuts[t];
                const int edge[5] = {nd.lo, c.x, c.y, c.z, nd.hi};
                int j = va[t];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int a = edge[q], b = edge[q + 1];
                    if (b > a) {
                        const int pos = sTot - 1 - j;
                        QtNode ch; ch.lo = a; ch.hi = b; ch.depth = nd.depth + 1;
                        nxt[pos] = ch;
                        if (b - a > 1) {
                            const int slot = atomicAdd(&sh_prevCount, 1);
                            const unsigned long long e = ((unsigned long long)(b - a) << 48) |
                                                         ((unsigned long long)j << 24) | (unsigned long long)pos;
                            prevNew[slot] = e;
                        }
                        j++;
                    }
                }
            }
            for (int i = tid; i < size0; i += T) {
                const QtNode nd = cur[i];
                // alive test: recompute from the scan (vb[i+1]-vb[i] is 1 for alive nodes)
                const int here = vb[i];
                const int next = (i + 1 < size0) ? vb[i + 1] : kTot;
                if (next != here) nxt[sTot + here] = nd;
            }
            __syncthreads();
            if (tid == 0) {
                const int size = sTot + kTot;
                sh_size = size;
                sh_inB ^= 1;
                sh_prevInB ^= 1;
                if (size >= N || size == size0) sh_state = 2;                 // :762
            }
            __syncthreads();
        }
    }

    // ---- keep the best key of every node, in list order (:770-789)
    const QtNode* fin = sh_inB ? B : A;
    const int size = sh_size;
    if (size > L.kpCap) {
        if (tid == 0) { orb_flag_error(errFlags, f, 2); *outCount = 0; }
        return;
    }
    uint32_t* out = kpl + (size_t)f * G.kpSlab + L.kpBase;
    for (int i = tid; i < size; i += T) {
        const QtNode nd = fin[i];
        unsigned long long bestKey = 0;
        for (int k = nd.lo; k < nd.hi; k++) {
            // max response; among equals the candidate the reference appended first: smallest (ci,cj,y,x)
            const unsigned long long key = keys[k];
            const unsigned resp = (unsigned)(key & 0xFF), ord = (unsigned)(key >> 8) & 0x3FFFFFFu;
            const unsigned bresp = (unsigned)(bestKey & 0xFF), bord = (unsigned)(bestKey >> 8) & 0x3FFFFFFu;
            if (k == nd.lo || resp > bresp || (resp == bresp && ord < bord)) bestKey = key;
        }
        const int ci = (int)(bestKey >> 27) & 0x7F, cj = (int)(bestKey >> 20) & 0x7F;
        const int yin = (int)(bestKey >> 14) & 0x3F, xin = (int)(bestKey >> 8) & 0x3F;
        const int x = xin + cj * L.wCell + 16, y = yin + ci * L.hCell + 16;     // + minBorder (:892-893)
        out[i] = ((uint32_t)x << 20) | ((uint32_t)y << 8) | (uint32_t)(bestKey & 0xFF);
    }
    if (tid == 0) *outCount = size;
}

__global__ __launch_bounds__(256) void k_quadtree(const OrbGeom G, unsigned long long* __restrict__ cand,
                                                  size_t candSlab, const int* __restrict__ candCount,
                                                  uint32_t* __restrict__ kpl, int* __restrict__ kpCount,
                                                  int* __restrict__ errFlags, int sortCap, int nodeCap)
{
    extern __shared__ unsigned long long qsm[];
    // LDS carve-up: keys[sortCap] | prevA,prevB[nodeCap] (u64) | A,B[nodeCap] (QtNode) | cuts[nodeCap] (int3)
    //               | va[nodeCap] | vb[nodeCap] (int) | part[257]
    unsigned long long* ldsKeys = qsm;
    unsigned long long* prevA = qsm + sortCap;
    unsigned long long* prevB = prevA + nodeCap;
    QtNode* A = reinterpret_cast<QtNode*>(prevB + nodeCap);
    QtNode* B = A + nodeCap;
    int3* cuts = reinterpret_cast<int3*>(B + nodeCap);
    int* va = reinterpret_cast<int*>(cuts + nodeCap);
    int* vb = va + nodeCap;
    int* part = vb + nodeCap;
    __shared__ int sh[8];                              // size, prevCount, state (0 full pass, 1 careful, 2 done), inB, prevInB, tstar

    // grid = (frames, levels): workgroups are dealt round-robin over the 8 XCDs in linear order, so with the frame
    // index fastest every XCD gets the same mix of levels (level fastest would send ALL level-0 workgroups, the
    // longest ones, to one XCD), and the big levels are dispatched first (longest-processing-time-first).
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
    unsigned long long* gk = cand + (size_t)f * candSlab + L.candBase;
    if (n <= sortCap) {
        for (int i = tid; i < n; i += T) ldsKeys[i] = gk[i];
        qt_body(ldsKeys, n, G, L, f, prevA, prevB, A, B, cuts, va, vb, part, kpl, outCount, errFlags, sh);
    } else {
        qt_body(gk, n, G, L, f, prevA, prevB, A, B, cuts, va, vb, part, kpl, outCount, errFlags, sh);   // rare: sort in global memory
    }
}

size_t orb_quadtree_lds_bytes(int sortCap, int nodeCap)
{
    return (size_t)sortCap * 8 + (size_t)nodeCap * (16 + 2 * sizeof(QtNode) + sizeof(int3) + 8) + 258 * 4;
}

void orb_launch_quadtree(hipStream_t st, const OrbGeom& G, unsigned long long* cand, size_t candSlab,
                         const int* candCount, uint32_t* kpl, int* kpCount, int* errFlags, int sortCap,
                         int nodeCap, int nFrames)
{
    const size_t lds = orb_quadtree_lds_bytes(sortCap, nodeCap);
    // quotas beyond ~1000 per level (nFeatures >~ 4500) need more than the default 64 KB of dynamic LDS: a workgroup may
    // use the CU's whole 160 KB (one workgroup per CU then -- only the huge-quota configurations pay that)
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_quadtree, dim3(nFrames, G.nlevels), dim3(256), lds, st, G, cand, candSlab, candCount,
                       kpl, kpCount, errFlags, sortCap, nodeCap);
}
