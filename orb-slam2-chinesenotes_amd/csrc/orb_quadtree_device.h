// orb_quadtree_device.h -- device code of the quadtree keypoint distribution (see orb_quadtree.hip for the design notes):
// included by orb_quadtree.hip (k_quadtree, k_quadtree_gnodes).
#pragma once
#include "orb_block_sort.h"
#include "orb_kernels.h"
#include "orb_wave.h"

// diagnostics (orb_extractor_set_qt_stamps): thread 0 of a workgroup leaves the 100 MHz clock at the stage boundaries of its
// (frame, level) instance: 0 start, 1 keys sorted, 2 full passes done (closed form or roots), 3 careful phase done, 4 keypoints
// emitted; word 5 = candidates | careful iterations << 16 | expandable nodes at the first careful iteration << 32 | list size
// there << 48.  8 words per workgroup, workgroup = level * frames + frame.
__device__ unsigned long long* g_qtStamps = nullptr;
#define QT_STAMP(k) do { if (qtStamps && threadIdx.x == 0) qtStamps[(k)] = __builtin_amdgcn_s_memrealtime(); } while (0)

struct QtNode {
    int lo, hi;      // key range in the sorted candidate array
    int depth;       // number of path digits already consumed
};

__device__ __forceinline__ unsigned qt_digit(unsigned long long k, int depth)
{
    return (unsigned)(k >> (ORB_KEY_PATH_SHIFT + 2 * (ORB_KEY_PATH_LEVELS - 1 - depth))) & 3u;
}

// DivideNode: the three interior cut points of a node's key range.  The three searches (first key whose digit is >= 1,
// >= 2, >= 3) run side by side over the whole range: three independent LDS reads in flight per step instead of a chain
// of three dependent searches (the division passes are chains of dependent LDS round trips, nothing else).
__device__ __forceinline__ int3 qt_cuts(const unsigned long long* keys, const QtNode nd)
{
    int3 c;
    if (nd.depth >= ORB_KEY_PATH_LEVELS) {            // unreachable inside the supported image envelope
        c.x = c.y = c.z = nd.hi;
        return c;
    }
    const int sh = ORB_KEY_PATH_SHIFT + 2 * (ORB_KEY_PATH_LEVELS - 1 - nd.depth);
    int l1 = nd.lo, h1 = nd.hi, l2 = nd.lo, h2 = nd.hi, l3 = nd.lo, h3 = nd.hi;
    while (l1 < h1 || l2 < h2 || l3 < h3) {
        const int m1 = (l1 + h1) >> 1, m2 = (l2 + h2) >> 1, m3 = (l3 + h3) >> 1;
        // (a finished search has l == h: its probe index may be nd.hi, one past the node -- clamp, the result is unused)
        const unsigned d1 = (unsigned)(keys[min(m1, nd.hi - 1)] >> sh) & 3u;
        const unsigned d2 = (unsigned)(keys[min(m2, nd.hi - 1)] >> sh) & 3u;
        const unsigned d3 = (unsigned)(keys[min(m3, nd.hi - 1)] >> sh) & 3u;
        if (l1 < h1) { if (d1 < 1u) l1 = m1 + 1; else h1 = m1; }
        if (l2 < h2) { if (d2 < 2u) l2 = m2 + 1; else h2 = m2; }
        if (l3 < h3) { if (d3 < 3u) l3 = m3 + 1; else h3 = m3; }
    }
    c.x = l1; c.y = l2; c.z = l3;
    return c;
}

// The initial sort by key, without a sorting network: the bitonic sort of ~1000 keys is 45 barrier-separated steps of
// dependent LDS round trips (13 us of a 45 us workgroup).  The top of a key is (root, quadrant path), so a bucket =
// (root, first D path digits) is a contiguous range of the sorted order:
//   1 histogram of the buckets with LDS atomics (the returned slot is an arbitrary order inside the bucket)
//   2 exclusive scan of the histogram by one wave
//   3 scatter to tmp[start[bucket] + slot]
//   4 rank every key inside its bucket (keys are unique; buckets hold ~n / 64 keys) -> keys[]
// Five barriers.  Returns false -- keys untouched -- when the scratch is too small or a bucket holds more than
// QT_BUCKET_MAX keys (clustered corners: the rank step is quadratic in the bucket size); the caller then sorts with the
// network.  scratch = everything behind keys in the workgroup's LDS (node lists, unused until the sort is done).
#define QT_BUCKET_MAX 96
__device__ __forceinline__ bool qt_bucket_sort(unsigned long long* keys, int n, int nIni, unsigned char* scratch, int scratchBytes,
                                               int* shFlag)
{
    const int T = blockDim.x, tid = threadIdx.x;
    const int D = n >= 256 ? 3 : 2;                    // path digits in the bucket index: 64 or 16 buckets per root
    const int nb = nIni << (2 * D);
    if ((size_t)n * 10 + (size_t)nb * 8 + 16 > (size_t)scratchBytes || nb > 1024) return false;
    const int shift = ORB_KEY_PATH_SHIFT + 2 * (ORB_KEY_PATH_LEVELS - D);      // key >> shift = root << 2D | first D digits
    unsigned long long* tmp = reinterpret_cast<unsigned long long*>(scratch);
    int* cnt = reinterpret_cast<int*>(tmp + n);
    int* start = cnt + nb;
    unsigned short* slot = reinterpret_cast<unsigned short*>(start + nb);
    for (int t = tid; t < nb; t += T) cnt[t] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += T) slot[i] = (unsigned short)atomicAdd(&cnt[(int)(keys[i] >> shift)], 1);
    __syncthreads();
    if (tid < 64) {                                    // exclusive scan over the buckets: lane owns a contiguous chunk
        const int C = (nb + 63) / 64;
        const int b = min(tid * C, nb), e = min(b + C, nb);
        int sum = 0, mx = 0;
        for (int t = b; t < e; t++) { const int c = cnt[t]; sum += c; mx = max(mx, c); }
        const int incl = orb_wave_scan_incl(sum);
        int run = incl - sum;
        for (int t = b; t < e; t++) { start[t] = run; run += cnt[t]; }
        mx = (int)~orb_wave_umin(~(unsigned)mx);
        if (tid == 0) *shFlag = mx;
    }
    __syncthreads();
    if (*shFlag > QT_BUCKET_MAX) return false;         // (uniform; nothing has been moved yet)
    for (int i = tid; i < n; i += T) {
        const unsigned long long k = keys[i];
        tmp[start[(int)(k >> shift)] + slot[i]] = k;
    }
    __syncthreads();
    for (int i = tid; i < n; i += T) {
        const unsigned long long k = tmp[i];
        const int b = (int)(k >> shift), s0 = start[b], c = cnt[b];
        int rank = 0;
        for (int j = 0; j < c; j++) rank += tmp[s0 + j] < k;
        keys[s0 + rank] = k;
    }
    __syncthreads();
    return true;
}

// The same sort with the keys in REGISTERS (round 5): the candidates come from global memory QT_RK at a time per thread and never
// need a second LDS copy -- the version above scatters into tmp[] and ranks from tmp[] into keys[], i.e. wants 10 bytes of
// scratch per key, and a level-0 workgroup of a natural-statistics frame (1900 candidates against ~950 on the drawn content)
// did not have them: it fell back to the sorting network, 33 us of a 46 us workgroup (tools/qt_stamps.py).  Here a thread keeps
// its keys in registers from the load to the final store: histogram -> scan -> scatter into keys[] (grouped by bucket) ->
// barrier -> rank inside the bucket (reads only) -> barrier -> store to the final slot.  Scratch: 8 bytes per bucket + 2 per
// key.  Returns false -- keys[] then holds the UNSORTED candidates -- when a bucket exceeds QT_BUCKET_MAX or the scratch is too
// small; the caller sorts with the network.  n <= QT_RK * blockDim.x (QT_RK = 8 keys per thread up to 2048 candidates, 16 up to
// 4096: natural-statistics frames reach 2100 on level 0).
template <int QT_RK>
__device__ __forceinline__ bool qt_bucket_sort_regs(unsigned long long* keys, const unsigned long long* __restrict__ gk, int n, int nIni,
                                                    unsigned char* scratch, int scratchBytes, int* shFlag)
{
    const int T = blockDim.x, tid = threadIdx.x;
    unsigned long long kv[QT_RK];
#pragma unroll
    for (int b = 0; b < QT_RK; b++) kv[b] = gk[min(tid + b * T, n - 1)];          // all loads in flight (unconditional, clamped)
    const int D = n >= 256 ? 3 : 2;
    const int nb = nIni << (2 * D);
    const bool fits = (size_t)n * 2 + (size_t)nb * 8 + 16 <= (size_t)scratchBytes && nb <= 1024;
    const int shift = ORB_KEY_PATH_SHIFT + 2 * (ORB_KEY_PATH_LEVELS - D);
    int* cnt = reinterpret_cast<int*>(scratch);
    int* start = cnt + nb;
    unsigned short slotv[QT_RK];
    if (fits) {
        for (int t = tid; t < nb; t += T) cnt[t] = 0;
        __syncthreads();
#pragma unroll
        for (int b = 0; b < QT_RK; b++)
            if (tid + b * T < n) slotv[b] = (unsigned short)atomicAdd(&cnt[(int)(kv[b] >> shift)], 1);
        __syncthreads();
        if (tid < 64) {                                    // exclusive scan over the buckets + the largest bucket
            const int C = (nb + 63) / 64;
            const int b0 = min(tid * C, nb), e = min(b0 + C, nb);
            int sum = 0, mx = 0;
            for (int t = b0; t < e; t++) { const int c = cnt[t]; sum += c; mx = max(mx, c); }
            const int incl = orb_wave_scan_incl(sum);
            int run = incl - sum;
            for (int t = b0; t < e; t++) { start[t] = run; run += cnt[t]; }
            mx = (int)~orb_wave_umin(~(unsigned)mx);
            if (tid == 0) *shFlag = mx;
        }
        __syncthreads();
    }
    if (!fits || *shFlag > QT_BUCKET_MAX) {                // (uniform) the network sorts them: hand the candidates over as they came
#pragma unroll
        for (int b = 0; b < QT_RK; b++)
            if (tid + b * T < n) keys[tid + b * T] = kv[b];
        __syncthreads();
        return false;
    }
#pragma unroll
    for (int b = 0; b < QT_RK; b++)
        if (tid + b * T < n) keys[start[(int)(kv[b] >> shift)] + slotv[b]] = kv[b];
    __syncthreads();
    int rankv[QT_RK];
#pragma unroll
    for (int b = 0; b < QT_RK; b++) {
        rankv[b] = 0;
        if (tid + b * T < n) {
            const int bk = (int)(kv[b] >> shift), s0 = start[bk], c = cnt[bk];
            int r = 0;
            for (int j = 0; j < c; j++) r += keys[s0 + j] < kv[b];
            rankv[b] = s0 + r;
        }
    }
    __syncthreads();                                       // every read of the grouped order is done
#pragma unroll
    for (int b = 0; b < QT_RK; b++)
        if (tid + b * T < n) keys[rankv[b]] = kv[b];
    __syncthreads();
    return true;
}

// in-place exclusive scan of a[0..n) by the whole block; returns the total.  part = int[blockDim.x+1].
static __device__ int qt_scan(int* a, int n, int* part)
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

// ---- The full passes in closed form --------------------------------------------------------------------------------
// While the reference is in its "divide every node" passes (:631-701) the list after pass D holds exactly the nodes of
// depth D -- one per distinct (root, first D path digits) among the keys -- plus the single-key nodes that stopped
// earlier.  With the keys sorted, all of that can be read off the number of leading symbols (root, then 2-bit digits)
// that neighbouring keys share:
//   c[i] = common leading symbols of keys i-1 and i (0: different roots; c[0] = c[n] = 0)
//   keys i-1, i lie in the same depth-d node  <=>  c[i] >= d + 1
//   nodes at depth d:            C[d] = #{i : c[i] <= d}          (every node has one first key)
//   nodes with >= 2 keys:        E[d] = #{i : c[i] <= d < c[i+1]}
//   a single-key node {i} was born (became bNoMore) at depth max(c[i], c[i+1])
// so the number of full passes D follows from C and E by the reference's own tests (:695-701), without dividing
// anything.  The ORDER of the list after pass D follows from push_front: a pass reverses the order of the parents it
// divides and puts their children in front of everything else, last child first; nodes that were already single keep
// their relative order behind them.  By induction the nodes born at depth b appear sorted by
//   (root, digit 1, ..., digit b) with digit j DESCENDING iff b - j is even (the root: iff b is odd),
// newest generation first.  Flipping the descending symbols turns that into one ascending 32-bit key per node
// ((D - b) << 28 | flipped prefix), and a rank sort of the <= N nodes puts them where D passes would have.
// The creation number of a depth-D node in pass D (the tie-break of the careful phase) is
// sTot - 1 - position, as in the pass itself.  Level 0 of a 640 x 480 frame: root search 1.3 us + 4 full passes 8.2 us
// (chains of dependent binary searches, four barriers per pass) become 5.4 us (measured with early exits).
__device__ __forceinline__ int qt_common(unsigned long long a, unsigned long long b)
{
    const unsigned x = (unsigned)((a ^ b) >> ORB_KEY_PATH_SHIFT);              // root (4 bits) | 12 digits (24 bits)
    if (x >> 24) return 0;
    if (x == 0) return 1 + ORB_KEY_PATH_LEVELS;
    return 1 + ((__clz((int)x) - 8) >> 1);
}

// exclusive scan of one value per thread (256 threads); *total = sum.  part = int[257].
__device__ __forceinline__ int qt_scan256(int v, int* part, int* total)
{
    const int t = threadIdx.x;
    part[t] = v;
    __syncthreads();
    if (t < 64) {
        const int q0 = part[4 * t], q1 = part[4 * t + 1], q2 = part[4 * t + 2], q3 = part[4 * t + 3];
        const int mine = q0 + q1 + q2 + q3;
        const int incl = orb_wave_scan_incl(mine);
        const int ex = incl - mine;
        part[4 * t] = ex; part[4 * t + 1] = ex + q0; part[4 * t + 2] = ex + q0 + q1; part[4 * t + 3] = ex + q0 + q1 + q2;
        if (t == 63) part[256] = incl;
    }
    __syncthreads();
    const int r = part[t];
    *total = part[256];
    __syncthreads();
    return r;
}

struct QtClosed { int size, pc, state, ok; };

// keys sorted; c8 = n + 1 bytes of scratch; cnt = int[42]; A / prev / va / vb / part as in qt_body.  Returns ok = 0
// (nothing but scratch touched) when the passes would go deeper than the path has digits or the list does not fit.
__device__ __forceinline__ QtClosed qt_full_passes_closed(const unsigned long long* keys, int n, int N, int nodeCap, unsigned char* c8,
                                                         int* cnt, QtNode* A, unsigned long long* prev, int* va, int* vb, int* part,
                                                         int* shCount)
{
    const int tid = threadIdx.x, T = blockDim.x;
    constexpr int ND = ORB_KEY_PATH_LEVELS + 1;                                 // depths 0 .. 12
    QtClosed R; R.size = 0; R.pc = 0; R.state = 0; R.ok = 0;
    // cnt: h[v] = #{i : c[i] = v} | P[v] = #{i : c[i] = v < c[i+1]} | M[v] = #{i : c[i] < c[i+1] = v}, v = 0 .. 13:
    //   C[d] = sum of h[0 .. d],   E[d] = sum of (P - M)[0 .. d]      (a key with c[i] < c[i+1] opens a node of >= 2 keys at
    //   every depth c[i] .. c[i+1] - 1)
    constexpr int NB = ND + 1;
    for (int t = tid; t < 3 * NB; t += T) cnt[t] = 0;
    if (tid == 0) { c8[n] = 0; *shCount = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += T) {
        const unsigned long long k = keys[i];
        const int ci = i > 0 ? qt_common(keys[i - 1], k) : 0;
        const int cn = i + 1 < n ? qt_common(k, keys[i + 1]) : 0;
        c8[i] = (unsigned char)ci;
        atomicAdd(&cnt[ci], 1);
        if (ci < cn) { atomicAdd(&cnt[NB + ci], 1); atomicAdd(&cnt[2 * NB + cn], 1); }
    }
    __syncthreads();
    int Cs[ND], Es[ND];
    {
        int c = 0, e = 0;
#pragma unroll
        for (int d = 0; d < ND; d++) { c += cnt[d]; e += cnt[NB + d] - cnt[2 * NB + d]; Cs[d] = c; Es[d] = e; }
    }
    // the reference's loop tests after every full pass (:695, :701)
    int D = 0, st = 0, Cprev = Cs[0], Eprev = Es[0], M = 0, pcD = 0;
#pragma unroll
    for (int p = 1; p < ND; p++) {
        if (!D) {
            const int Cd = Cs[p], Ed = Es[p];
            if (Cd >= N || Cd == Cprev) { D = p; st = 2; }
            else if (Cd + 3 * Ed > N) { D = p; st = 1; }
            if (D) { M = Cd; pcD = Ed; }
            else { Cprev = Cd; Eprev = Ed; }
        }
    }
    if (!D) return R;                                                          // deeper than the path: the caller divides node by node
    if (M > nodeCap) return R;
    const int sTot = M - (Cprev - Eprev);                                      // nodes born in pass D = all but the singles of depth D - 1
    // node m = the m-th key with c <= D: contiguous chunk of keys per thread, scan of the per-thread counts
    {
        const int Cz = (n + T - 1) / T, b0 = min(tid * Cz, n), e0 = min(b0 + Cz, n);
        int mine = 0;
        for (int i = b0; i < e0; i++) mine += c8[i] <= D;
        int tot;
        int at = qt_scan256(mine, part, &tot);
        for (int i = b0; i < e0; i++)
            if (c8[i] <= D) va[at++] = i;
    }
    __syncthreads();
    auto node_of = [&](int m, QtNode& nd) -> unsigned {                         // node m and its order key
        nd.lo = va[m];
        nd.hi = m + 1 < M ? va[m + 1] : n;
        int b = D;
        if (nd.hi - nd.lo == 1) b = max((int)c8[nd.lo], (int)c8[nd.lo + 1]);   // born single at that depth (<= D)
        nd.depth = b;
        const unsigned kp = (unsigned)(keys[nd.lo] >> ORB_KEY_PATH_SHIFT);      // 28 bits
        const unsigned flip = (b & 1) ? 0xFCCCCCCu : 0x0333333u;
        const unsigned keep = ~((1u << (2 * (ORB_KEY_PATH_LEVELS - b))) - 1u) & 0xFFFFFFFu;
        return ((unsigned)(D - b) << 28) | ((kp ^ flip) & keep);
    };
    for (int m = tid; m < M; m += T) { QtNode nd; vb[m] = (int)node_of(m, nd); }
    __syncthreads();
    for (int m = tid; m < M; m += T) {
        QtNode nd;
        const unsigned mine = node_of(m, nd);
        int rank = 0, j = 0;
        for (; j + 8 <= M; j += 8) {
            unsigned v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = (unsigned)vb[j + u];
#pragma unroll
            for (int u = 0; u < 8; u++) rank += v[u] < mine;
        }
        for (; j < M; j++) rank += (unsigned)vb[j] < mine;
        A[rank] = nd;
        if (nd.hi - nd.lo > 1) {
            const int slot = atomicAdd(shCount, 1);
            prev[slot] = ((unsigned long long)(nd.hi - nd.lo) << 48) | ((unsigned long long)(sTot - 1 - rank) << 24) |
                         (unsigned long long)rank;
        }
    }
    __syncthreads();
    R.size = M; R.pc = pcD; R.state = st; R.ok = 1;
    return R;
}

// Everything after the keys are in place.  Force-inlined into both call sites so that the compiler knows the
// address space of `keys` (LDS: ds_* instructions; a runtime-selected generic pointer would turn every access
// of the sort and of the binary searches into slow flat_* operations).
__device__ __forceinline__ void qt_body(unsigned long long* keys, int n, const OrbGeom& G, const OrbLevelGeom& L, int f,
                                        unsigned long long* prevA, unsigned long long* prevB, QtNode* A, QtNode* B,
                                        int3* cuts, int* va, int* vb, int* part, uint32_t* __restrict__ kpl,
                                        int* outCount, int* __restrict__ errFlags, int* sh, unsigned char* scratch, int scratchBytes, int nodeCap,
                                        int sorted = 0)     // 0: keys unsorted, 1: sorted already, 2: unsorted and the bucket sort already refused them
{
    // shared words: sh[0] number of roots, sh[1] / sh[2] the two "expandable children" counters (passes alternate between
    // them: the one a pass does not count into is cleared for the next pass), sh[3] t* of the careful phase.
    // The loop state itself (phase, list size, which buffers hold the lists) is kept in registers, computed redundantly by
    // every thread from values all of them read after the same barrier: no thread-0 update + barrier at the end of a pass
    // and none at its start (a pass is a chain of barrier-separated steps; two of seven were only that bookkeeping).
    int& sh_tstar = sh[3];
    const int tid = threadIdx.x, T = blockDim.x;
    unsigned long long* qtStamps = g_qtStamps ? g_qtStamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
    int dbgIters = 0, dbgPc = 0, dbgSize = 0;
    QT_STAMP(0);
    __syncthreads();
    if (sorted == 2) orb_block_sort(keys, n);
    else if (sorted == 0 && !(scratch && qt_bucket_sort(keys, n, L.nIni, scratch, scratchBytes, &sh[3]))) orb_block_sort(keys, n);
    QT_STAMP(1);

    const int N = L.quota;
    int state = 0;                                     // 0 full passes, 1 careful phase, 2 done
    int size0 = 0, pc = 0, inB = 0, prevInB = 0, par = 0;
    bool closed = false;
    // the full passes in closed form (LDS instances: c[] lives in the second node list and the cut points, unused until
    // the careful phase; the 26 counters in the scan partials)
    if (scratch && (size_t)n + 2 <= (size_t)nodeCap * (sizeof(QtNode) + sizeof(int3)) && nodeCap >= 32) {
        if (tid == 0) { sh[1] = 0; sh[2] = 0; }
        const QtClosed R = qt_full_passes_closed(keys, n, N, nodeCap, reinterpret_cast<unsigned char*>(B), reinterpret_cast<int*>(prevB), A,
                                                 prevA, va, vb, part, &sh[3]);
        if (R.ok) { closed = true; size0 = R.size; pc = R.pc; state = R.state; }
    }
    if (!closed) {
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
        sh[0] = cnt;
        sh[1] = 0;
        sh[2] = 0;
    }
    __syncthreads();
    size0 = sh[0];
    }
    QT_STAMP(2);
    while (state != 2) {
        if (state == 1 && pc == 0) break;              // nothing left to expand: size cannot change (:762)
        if (state == 1) { if (dbgIters == 0) { dbgPc = pc; dbgSize = size0; } dbgIters++; }
        QtNode* cur = inB ? B : A;
        QtNode* nxt = inB ? A : B;
        unsigned long long* prev = prevInB ? prevB : prevA;
        unsigned long long* prevNew = prevInB ? prevA : prevB;
        int* cntNow = &sh[1 + par];                    // this pass counts the expandable children here ...
        int* cntNext = &sh[1 + (par ^ 1)];             // ... and clears the counter of the next pass

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
            __syncthreads();
            if (tid == 0) *cntNext = 0;                // every thread has read it (before this barrier) in the pass before
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
                                const int slot = atomicAdd(cntNow, 1);
                                prev[slot] = ((unsigned long long)(b - a) << 48) | ((unsigned long long)j << 24) |
                                             (unsigned long long)pos;
                            }
                            j++;
                        }
                    }
                }
            }
            __syncthreads();
            const int size = sTot + kTot;
            pc = *cntNow;
            inB ^= 1;
            par ^= 1;
            if (size >= N || size == size0) state = 2;                        // :695
            else if (size + 3 * pc > N) state = 1;                            // :701
            size0 = size;
        } else {
            // ---------------- careful phase (:703-765): largest first, stop as soon as size >= N
            // ascending (size, seq); processed from the back (:711-713).  Up to a workgroup's worth of candidates (the rule:
            // the phase starts when 3 x candidates could overshoot N, ~170 on level 0) are rank-sorted -- every thread
            // counts the entries below its own, 8 independent broadcast reads in flight -- : two barriers instead of the
            // 36 steps of the network (6.6 -> ~1 us)
            if (pc <= T && size0 <= 512) {
                // ---- the short form (round 4): a thread keeps ITS candidate in registers from the rank to the children (no
                // sorted copy), the candidates' child counts are scanned by ONE wave (<= 256 values, four per lane), the erased
                // parents are bits of a 512-bit mask and an alive node's new position is a popcount.  5 barriers instead of 12;
                // same list, same candidate entries.
                unsigned long long mine = 0;
                int t = -1, idx = 0, ch = 0;
                QtNode nd = {0, 0, 0};
                int3 c = make_int3(0, 0, 0);
                if (tid < pc) {
                    mine = prev[tid];
                    int rank = 0, j = 0;
                    for (; j + 8 <= pc; j += 8) {
                        unsigned long long v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) v[u] = prev[j + u];
#pragma unroll
                        for (int u = 0; u < 8; u++) rank += v[u] < mine;
                    }
                    for (; j < pc; j++) rank += prev[j] < mine;
                    t = pc - 1 - rank;                                 // processed from the back (:711-713)
                    idx = (int)(mine & 0xFFFFFF);
                    nd = cur[idx];
                    c = qt_cuts(keys, nd);
                    ch = (c.x > nd.lo) + (c.y > c.x) + (c.z > c.y) + (nd.hi > c.z);
                    va[t] = ch;
                }
                if (tid < 16) vb[tid] = 0;                             // the mask of erased parents
                if (tid == 0) { sh_tstar = pc; *cntNext = 0; }
                __syncthreads();
                if (tid < 64) {                                        // exclusive scan of va[0 .. pc): lane owns 4 consecutive values
                    int q[4], mineS = 0;
#pragma unroll
                    for (int u = 0; u < 4; u++) { q[u] = 4 * tid + u < pc ? va[4 * tid + u] : 0; mineS += q[u]; }
                    const int incl = orb_wave_scan_incl(mineS);
                    int run = incl - mineS;
#pragma unroll
                    for (int u = 0; u < 4; u++) { if (4 * tid + u < pc) va[4 * tid + u] = run; run += q[u]; }
                }
                __syncthreads();
                int before = 0;
                if (t >= 0) {
                    before = va[t];                                    // children created before candidate t
                    if (size0 + before + ch - (t + 1) >= N) atomicMin(&sh_tstar, t);
                }
                __syncthreads();
                const int P = min(pc, sh_tstar + 1);                   // candidates actually divided before the break (:758)
                if (t >= 0 && t < P) {
                    atomicOr(&vb[idx >> 5], (int)(1u << (idx & 31)));
                    if (t == P - 1) part[0] = before + ch;             // sTot
                }
                __syncthreads();
                const int sTot = part[0], kTot = size0 - P;
                if (t >= 0 && t < P) {
                    const int edge[5] = {nd.lo, c.x, c.y, c.z, nd.hi};
                    int j = before;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int a = edge[q], b = edge[q + 1];
                        if (b > a) {
                            const int pos = sTot - 1 - j;
                            QtNode chn; chn.lo = a; chn.hi = b; chn.depth = nd.depth + 1;
                            nxt[pos] = chn;
                            if (b - a > 1) {
                                const int slot = atomicAdd(cntNow, 1);
                                prevNew[slot] = ((unsigned long long)(b - a) << 48) | ((unsigned long long)j << 24) | (unsigned long long)pos;
                            }
                            j++;
                        }
                    }
                }
                for (int i = tid; i < size0; i += T) {
                    const unsigned w = (unsigned)vb[i >> 5];
                    if (!((w >> (i & 31)) & 1u)) {                     // alive: its position = i - erased parents before it
                        int gone = __popc(w & ((1u << (i & 31)) - 1u));
                        for (int k = 0; k < (i >> 5); k++) gone += __popc((unsigned)vb[k]);
                        nxt[sTot + i - gone] = cur[i];
                    }
                }
                __syncthreads();
                const int size = sTot + kTot;
                pc = *cntNow;
                inB ^= 1;
                prevInB ^= 1;
                par ^= 1;
                if (size >= N || size == size0) state = 2;                        // :762
                size0 = size;
                continue;
            }
            if (pc <= T) {
                unsigned long long mine = 0;
                int rank = 0;
                if (tid < pc) {
                    mine = prev[tid];
                    int j = 0;
                    for (; j + 8 <= pc; j += 8) {
                        unsigned long long v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) v[u] = prev[j + u];
#pragma unroll
                        for (int u = 0; u < 8; u++) rank += v[u] < mine;
                    }
                    for (; j < pc; j++) rank += prev[j] < mine;
                }
                __syncthreads();
                if (tid < pc) prev[rank] = mine;
                __syncthreads();
            } else {
                orb_block_sort(prev, pc);
            }
            for (int t = tid; t < pc; t += T) {
                const int idx = (int)(prev[pc - 1 - t] & 0xFFFFFF);
                const QtNode nd = cur[idx];
                const int3 c = qt_cuts(keys, nd);
                cuts[t] = c;
                va[t] = (c.x > nd.lo) + (c.y > c.x) + (c.z > c.y) + (nd.hi > c.z);   // children of candidate t
            }
            for (int i = tid; i < size0; i += T) vb[i] = 1;                   // alive flags of the current list
            if (tid == 0) { sh_tstar = pc; *cntNext = 0; }
            __syncthreads();
            (void)qt_scan(va, pc, part);               // va[t] = children created before candidate t
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
            for (int t = tid; t < P; t += T) vb[(int)(prev[pc - 1 - t] & 0xFFFFFF)] = 0;     // erased parents
            // children of the first P candidates
            int sTot;
            {
                const int idxLast = (int)(prev[pc - P] & 0xFFFFFF);          // candidate t = P-1
                const QtNode nd = cur[idxLast];
                const int3 c = cuts[P - 1];
                sTot = va[P - 1] + (c.x > nd.lo) + (c.y > c.x) + (c.z > c.y) + (nd.hi > c.z);
            }
            __syncthreads();
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
                            const int slot = atomicAdd(cntNow, 1);
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
            const int size = sTot + kTot;
            pc = *cntNow;
            inB ^= 1;
            prevInB ^= 1;
            par ^= 1;
            if (size >= N || size == size0) state = 2;                        // :762
            size0 = size;
        }
    }

    QT_STAMP(3);
    // ---- keep the best key of every node, in list order (:770-789)
    const QtNode* fin = inB ? B : A;
    const int size = size0;
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
            const unsigned resp = (unsigned)(key & 0xFF), ord = (unsigned)(key >> 8) & ORB_KEY_ORD_MASK;
            const unsigned bresp = (unsigned)(bestKey & 0xFF), bord = (unsigned)(bestKey >> 8) & ORB_KEY_ORD_MASK;
            if (k == nd.lo || resp > bresp || (resp == bresp && ord < bord)) bestKey = key;
        }
        const int ci = (int)(bestKey >> ORB_KEY_CI_SHIFT) & 0xFF, cj = (int)(bestKey >> ORB_KEY_CJ_SHIFT) & 0xFF;
        const int yin = (int)(bestKey >> 14) & 0x3F, xin = (int)(bestKey >> 8) & 0x3F;
        const int x = xin + cj * L.wCell + 16, y = yin + ci * L.hCell + 16;     // + minBorder (:892-893)
        out[i] = ((uint32_t)x << 20) | ((uint32_t)y << 8) | (uint32_t)(bestKey & 0xFF);
    }
    if (tid == 0) *outCount = size;
    QT_STAMP(4);
    if (qtStamps && tid == 0)
        qtStamps[5] = (unsigned long long)(unsigned)n | ((unsigned long long)(dbgIters & 0xFFFF) << 16) | ((unsigned long long)(dbgPc & 0xFFFF) << 32) |
                      ((unsigned long long)(dbgSize & 0xFFFF) << 48);
}

// One (frame, level) instance with its keys and lists in the workgroup's LDS at qsm (carve-up below; sh = four shared words):
// the body of k_quadtree
__device__ __forceinline__ void qt_instance_lds(unsigned long long* qsm, int* sh, const OrbGeom& G, int level, int f,
                                                unsigned long long* __restrict__ cand, size_t candSlab, int n,
                                                uint32_t* __restrict__ kpl, int* __restrict__ kpCount, int* __restrict__ errFlags,
                                                int sortCap, int nodeCap, int* __restrict__ ovfBlock)
{
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
    const OrbLevelGeom& L = G.L[level];
    const int tid = threadIdx.x, T = blockDim.x;
    if (n > L.candCap) n = L.candCap;
    int* outCount = &kpCount[f * ORB_MAX_LEVELS + level];
    if (n == 0) {
        if (tid == 0) *outCount = 0;
        return;
    }
    unsigned long long* gk = cand + (size_t)f * candSlab + L.candBase;
    if (n <= sortCap) {
        // scratch of the bucket sort: the node lists behind the keys (prevA .. part), unused until the sort is done
        unsigned char* scr = reinterpret_cast<unsigned char*>(prevA);
        const int scrBytes = nodeCap * (16 + 2 * (int)sizeof(QtNode) + (int)sizeof(int3) + 8) + 257 * 4;
        int sorted = 0;
        if (n <= 16 * T) {
            // the candidates go from global memory through registers straight to their sorted place (qt_bucket_sort_regs)
            __syncthreads();
            if (n <= 8 * T) sorted = qt_bucket_sort_regs<8>(ldsKeys, gk, n, L.nIni, scr, scrBytes, &sh[3]) ? 1 : 2;
            else sorted = qt_bucket_sort_regs<16>(ldsKeys, gk, n, L.nIni, scr, scrBytes, &sh[3]) ? 1 : 2;
        } else {
            for (int i0 = tid; i0 < n; i0 += 8 * T) {
                unsigned long long kv[8];
#pragma unroll
                for (int b = 0; b < 8; b++) kv[b] = gk[min(i0 + b * T, n - 1)];
#pragma unroll
                for (int b = 0; b < 8; b++)
                    if (i0 + b * T < n) ldsKeys[i0 + b * T] = kv[b];
            }
        }
        qt_body(ldsKeys, n, G, L, f, prevA, prevB, A, B, cuts, va, vb, part, kpl, outCount, errFlags, sh, scr, scrBytes, nodeCap, sorted);
    } else {
        // more candidates than the LDS holds: the host grows the sort capacity when it hears of it (word 1 of the overflow
        // block: at a sync, or through the unsynchronised feedback of orb_extract_batch_device)
        if (tid == 0) atomicMax(&ovfBlock[1], n);
        qt_body(gk, n, G, L, f, prevA, prevB, A, B, cuts, va, vb, part, kpl, outCount, errFlags, sh, nullptr, 0, nodeCap);   // rare: sort in global memory
    }
}
