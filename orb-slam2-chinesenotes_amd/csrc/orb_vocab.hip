// orb_vocab.hip -- DBoW2 vocabulary-tree descent on gfx950: the part of
// TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) that touches descriptors, as called
// by Frame::ComputeBoW (reference src/Frame.cc:425-433) and KeyFrame::ComputeBoW (src/KeyFrame.cc:70): per feature
// the word (leaf) it falls into and the node it passes at level L - levelsup.  DBoW2 itself is absent from the
// reference tree; the algorithm is restated from its published source (first-minimum Hamming descent, children in
// stored order).  Building the std::map containers (and the tf-idf BowVector, which uses doubles) stays on the host.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <vector>

#include "orb_matcher_internal.h"

struct orb_vocab {
    int device = 0, nNodes = 0, L = 0;
    // The tree is re-laid out at creation: nodes are numbered in breadth-first SLOT order, so the children of a node
    // are consecutive slots in their stored order (what "first minimum wins" iterates over) and one level of a descent
    // reads ONE contiguous run of k x 32 bytes instead of k scattered 32-byte rows (ids come from file order in DBoW2).
    MBuf slotDesc;          // [nNodes][32]
    MBuf slotKids;          // int2 per slot: {first child slot, number of children | ORB_VOCAB_LEAFKIDS when every child is a leaf}
    MBuf slotNode;          // int32 per slot: the caller's node id
    MBuf slotWord;          // int32 per slot: word id of the slot's node (one load at the end of a descent, not two)
    std::vector<int> depth;                         // host: depth of every node
    int maxKids = 0;                                // largest number of children of a node
    int nTop = 0;                                   // slots of depth <= 2 (a prefix of the breadth-first order), at most 400: staged in LDS by the descent
    std::vector<int32_t> slotNodeHost;              // host copy of slotNode
    std::map<int, std::pair<MBuf, int>> compact;    // levelsup -> (device int32 compactOfSlot[nSlots], K)
};

#include "orb_wave.h"

// bit 30 of a slot's child count: all its children are leaves -- the descent then needs no child record from that level (they
// are all {0, 0}: in a complete k = 10, L = 6 tree 8 MB of zeros that the last, least cached level of every descent used to fetch)
#define ORB_VOCAB_LEAFKIDS 0x40000000

// minimum over the 16 lanes of a DPP row, returned in every lane of the row (row_ror 8, 4, 2, 1)
__device__ __forceinline__ unsigned row16_umin(unsigned v)
{
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false));
    return v;
}

// General form (nodes with any number of children; the kernel for vocabularies with k > 16): 16 lanes (one DPP row)
// per feature, lane r scores children r, r + 16, ... of the current node (their 32-byte descriptors are one contiguous
// run), the row minimum of (distance << 16 | child index) is the first-minimum child.
__global__ __launch_bounds__(256) void k_vocab_transform(const uint4* __restrict__ slotDesc, const int2* __restrict__ slotKids,
                                                         const int32_t* __restrict__ slotNode,
                                                         const int32_t* __restrict__ slotWord, int nidLevel,
                                                         const int32_t* __restrict__ compactOfSlot,
                                                         const uint8_t* __restrict__ desc,
                                                         const int32_t* __restrict__ counts, int cap,
                                                         int32_t* __restrict__ wordOf, int32_t* __restrict__ nodeId,
                                                         uint16_t* __restrict__ nodeOf)
{
    const int f = blockIdx.y, r = threadIdx.x & 15, i = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (i >= cap) return;
    const size_t row = (size_t)f * cap + i;
    const int n = counts ? counts[f] : cap;
    if (i >= n) {
        if (nodeOf && r == 0) nodeOf[row] = 0xFFFF;
        return;
    }
    const uint4 lo = reinterpret_cast<const uint4*>(desc + row * 32)[0], hi = reinterpret_cast<const uint4*>(desc + row * 32)[1];
    int slot = 0, level = 0;
    int nid = (nidLevel <= 0) ? slotNode[0] : -1, cix = (nidLevel <= 0 && compactOfSlot) ? compactOfSlot[0] : -1;
    int2 kids = slotKids[0];
    kids.y &= 0xFFFF;
    while (kids.y > 0) {                                          // do { ... } while (!isLeaf())
        ++level;
        unsigned best = 0xFFFFFFFFu;
        for (int c0 = 0; c0 < kids.y; c0 += 16) {
            const int c = c0 + r;
            unsigned key = 0xFFFFFFFFu;
            if (c < kids.y) {
                const uint4* nd = slotDesc + (size_t)(kids.x + c) * 2;
                const uint4 nl = nd[0], nh = nd[1];
                const int h = __popc(lo.x ^ nl.x) + __popc(lo.y ^ nl.y) + __popc(lo.z ^ nl.z) + __popc(lo.w ^ nl.w) +
                              __popc(hi.x ^ nh.x) + __popc(hi.y ^ nh.y) + __popc(hi.z ^ nh.z) + __popc(hi.w ^ nh.w);
                key = ((unsigned)h << 16) | (unsigned)c;
            }
            best = min(best, row16_umin(key));
        }
        slot = kids.x + (int)(best & 0xffffu);
        kids = slotKids[slot];
        kids.y &= 0xFFFF;
        if (level == nidLevel) {
            nid = slotNode[slot];
            if (compactOfSlot) cix = compactOfSlot[slot];
        }
    }
    if (r != 0) return;
    if (wordOf) wordOf[row] = slotWord[slot];
    if (nodeId) nodeId[row] = nid;
    if (nodeOf) nodeOf[row] = (nid >= 0 && compactOfSlot) ? (uint16_t)cix : (uint16_t)0xFFFF;
}

// The same descent for vocabularies whose nodes have at most 16 children (every DBoW2 vocabulary: k = 10), written
// branch-free and for NF features per DPP row: all loads of a level -- NF x (child descriptor + child record) -- are
// issued back to back before any is waited for, so a wave keeps 4 NF chains in flight; a lane fetches its child's
// {first child, count} record together with the child's descriptor and the winner's record is passed along the row
// (ds_bpermute), so a level costs ONE global round trip, not two (155 -> 103 us per 512 frames; NF = 1 / 2 / 4: 106 /
// 103 / 129 us -- at 4.4 TB/s of 320-byte runs from a 35 MB table the Infinity Cache, not the chain, is the limit).
// Finished or absent features keep loading slot 0 and ignore what comes back.
template <int NF>
__global__ __launch_bounds__(256) void k_vocab_transform_k16(const uint4* __restrict__ slotDesc, const int2* __restrict__ slotKids,
                                                             const int32_t* __restrict__ slotNode,
                                                             const int32_t* __restrict__ slotWord, int nidLevel,
                                                             const int32_t* __restrict__ compactOfSlot,
                                                             const uint8_t* __restrict__ desc,
                                                             const int32_t* __restrict__ counts, int cap,
                                                             int32_t* __restrict__ wordOf, int32_t* __restrict__ nodeId,
                                                             uint16_t* __restrict__ nodeOf, int nTop)
{
    // The first nTop slots (breadth-first: the root and the top two levels, 111 nodes of 40 bytes for k = 10) are staged in
    // LDS: two of a descent's six levels then cost no global load at all (the kernel is bound by the memory side: a third
    // of its 16-byte lanes through the texture path gone)
    extern __shared__ uint4 vsm[];
    int2* kidsL = reinterpret_cast<int2*>(vsm + 2 * nTop);
    for (int i = threadIdx.x; i < 2 * nTop; i += 256) vsm[i] = slotDesc[i];
    for (int i = threadIdx.x; i < nTop; i += 256) kidsL[i] = slotKids[i];
    if (nTop) __syncthreads();
    const int f = blockIdx.y, r = threadIdx.x & 15, rowIdx = threadIdx.x >> 4;
    const int base = blockIdx.x * (16 * NF);
    const int n = counts ? min(counts[f], cap) : cap;
    const int rowBase = (int)(threadIdx.x & 63u) & ~15;           // first lane of this DPP row
    const int2 rootKids = slotKids[0];
    int idx[NF], slot[NF], lvl[NF], nidSlot[NF];
    int2 kids[NF];
    uint4 lo[NF], hi[NF];
#pragma unroll
    for (int j = 0; j < NF; j++) {
        idx[j] = base + j * 16 + rowIdx;
        const size_t row = (size_t)f * cap + min(idx[j], cap - 1);
        lo[j] = reinterpret_cast<const uint4*>(desc + row * 32)[0];
        hi[j] = reinterpret_cast<const uint4*>(desc + row * 32)[1];
        kids[j] = idx[j] < n ? rootKids : make_int2(0, 0);
        slot[j] = 0; lvl[j] = 0;
        nidSlot[j] = nidLevel <= 0 ? 0 : -1;
    }
    while (true) {
        bool any = false;
#pragma unroll
        for (int j = 0; j < NF; j++) any |= (kids[j].y & 0xFFFF) != 0;
        if (!any) break;
        uint4 nl[NF], nh[NF];
        int2 mine[NF];
        // all loads of this level, back to back.  The LDS copy of the top levels is taken only when EVERY lane of the wave is
        // inside it (a wave-uniform branch: in a balanced tree all features are on the same level), otherwise every lane reads
        // global memory -- a per-lane `if (slot < nTop) LDS else global` put loads on both sides of a divergent branch, and the
        // compiler then waited for the loads of one feature before it requested those of the next (round 5, tools/isa_waits.py)
        int s2[NF];
        bool top = true, leafKids = true;
#pragma unroll
        for (int j = 0; j < NF; j++) {
            const int cj = kids[j].y & 0xFFFF;
            s2[j] = kids[j].x + min(r, max(cj - 1, 0));
            top = top && s2[j] < nTop;
            leafKids = leafKids && (cj == 0 || (kids[j].y & ORB_VOCAB_LEAFKIDS) != 0);
        }
        if (__all(top)) {
#pragma unroll
            for (int j = 0; j < NF; j++) {
                nl[j] = vsm[2 * s2[j]];
                nh[j] = vsm[2 * s2[j] + 1];
                mine[j] = kidsL[s2[j]];
            }
        } else if (__all(leafKids)) {                              // the last level: the children's records are all {0, 0}, not fetched
#pragma unroll
            for (int j = 0; j < NF; j++) {
                nl[j] = slotDesc[(size_t)s2[j] * 2];
                nh[j] = slotDesc[(size_t)s2[j] * 2 + 1];
                mine[j] = make_int2(0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NF; j++) {
                nl[j] = slotDesc[(size_t)s2[j] * 2];
                nh[j] = slotDesc[(size_t)s2[j] * 2 + 1];
                mine[j] = slotKids[s2[j]];
            }
        }
#pragma unroll
        for (int j = 0; j < NF; j++) {
            const int cj = kids[j].y & 0xFFFF;
            const bool active = cj > 0;
            const int h = __popc(lo[j].x ^ nl[j].x) + __popc(lo[j].y ^ nl[j].y) + __popc(lo[j].z ^ nl[j].z) + __popc(lo[j].w ^ nl[j].w) +
                          __popc(hi[j].x ^ nh[j].x) + __popc(hi[j].y ^ nh[j].y) + __popc(hi[j].z ^ nh[j].z) + __popc(hi[j].w ^ nh[j].w);
            const unsigned key = r < cj ? ((unsigned)h << 16) | (unsigned)r : 0xFFFFFFFFu;   // first minimum wins
            const int b = (int)(row16_umin(key) & 15u);
            const int nx = __builtin_amdgcn_ds_bpermute((rowBase + b) << 2, mine[j].x);
            const int ny = __builtin_amdgcn_ds_bpermute((rowBase + b) << 2, mine[j].y);
            lvl[j] += active ? 1 : 0;
            slot[j] = active ? kids[j].x + b : slot[j];
            nidSlot[j] = (active && lvl[j] == nidLevel) ? slot[j] : nidSlot[j];
            kids[j].x = active ? nx : kids[j].x;
            kids[j].y = active ? ny : 0;
        }
    }
#pragma unroll
    for (int j = 0; j < NF; j++) {
        if (r != 0 || idx[j] >= cap) continue;
        const size_t row = (size_t)f * cap + idx[j];
        if (idx[j] >= n) {
            if (nodeOf) nodeOf[row] = 0xFFFF;
            continue;
        }
        if (wordOf) wordOf[row] = slotWord[slot[j]];
        const int nid = nidSlot[j] >= 0 ? slotNode[nidSlot[j]] : -1;
        if (nodeId) nodeId[row] = nid;
        if (nodeOf) nodeOf[row] = (nid >= 0 && compactOfSlot) ? (uint16_t)compactOfSlot[nidSlot[j]] : (uint16_t)0xFFFF;
    }
}

extern "C" int orb_vocab_create(int device, const uint8_t* node_desc, const int32_t* child_begin, const int32_t* children,
                                const int32_t* word_id, int n_nodes, int L, orb_vocab** out)
{
    if (!out || !node_desc || !child_begin || !children || !word_id || n_nodes < 1 || L < 1) return ORB_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { orb_set_error("no HIP device"); return ORB_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) return ORB_ERR_INVALID;
    if (child_begin[0] != 0 || child_begin[n_nodes] < 1) { orb_set_error("vocabulary: root has no children"); return ORB_ERR_INVALID; }
    ORB_HIP_TRY(hipSetDevice(device));
    // validate the tree and compute depths (host)
    std::vector<int> depth(n_nodes, -1);
    depth[0] = 0;
    std::vector<int> order{0};
    const int nch = child_begin[n_nodes];
    for (size_t q = 0; q < order.size(); q++) {
        const int v = order[q];
        if (child_begin[v + 1] < child_begin[v] || child_begin[v + 1] > nch) return ORB_ERR_INVALID;
        for (int c = child_begin[v]; c < child_begin[v + 1]; c++) {
            const int id = children[c];
            if (id <= 0 || id >= n_nodes || depth[id] != -1) { orb_set_error("vocabulary: not a tree"); return ORB_ERR_INVALID; }
            depth[id] = depth[v] + 1;
            order.push_back(id);
        }
    }
    for (int i = 0; i < n_nodes; i++)
        if (child_begin[i + 1] - child_begin[i] > 65535) { orb_set_error("vocabulary: more than 65535 children"); return ORB_ERR_UNSUPPORTED; }
    // slot layout: `order` is a breadth-first order in which the children of a node are consecutive, in stored order
    // (nodes unreachable from the root get no slot: a descent never visits them)
    const int nSlots = (int)order.size();
    std::vector<int32_t> slotOf(n_nodes, -1), slotNode(nSlots);
    for (int s2 = 0; s2 < nSlots; s2++) { slotOf[order[s2]] = s2; slotNode[s2] = order[s2]; }
    std::vector<int32_t> kids(2 * (size_t)nSlots), sword(nSlots);
    std::vector<uint8_t> sdesc((size_t)32 * nSlots);
    for (int s2 = 0; s2 < nSlots; s2++) {
        const int node = order[s2], nc = child_begin[node + 1] - child_begin[node];
        kids[2 * (size_t)s2] = nc > 0 ? slotOf[children[child_begin[node]]] : 0;
        kids[2 * (size_t)s2 + 1] = nc;
        std::memcpy(&sdesc[(size_t)32 * s2], node_desc + (size_t)32 * node, 32);
        sword[s2] = word_id[node];
    }
    if (!std::getenv("ORB_VOCAB_NO_LEAFFLAG"))
        for (int s2 = 0; s2 < nSlots; s2++) {                          // ORB_VOCAB_LEAFKIDS: every child of the slot is a leaf
            const int nc = kids[2 * (size_t)s2 + 1], c0 = kids[2 * (size_t)s2];
            bool all = nc > 0;
            for (int r = 0; r < nc && all; r++) all = kids[2 * (size_t)(c0 + r) + 1] == 0;
            if (all) kids[2 * (size_t)s2 + 1] |= ORB_VOCAB_LEAFKIDS;
        }
    orb_vocab* v = new (std::nothrow) orb_vocab();
    if (!v) return ORB_ERR_INTERNAL;
    v->device = device; v->nNodes = n_nodes; v->L = L; v->depth = depth; v->slotNodeHost = slotNode;
    for (int i = 0; i < n_nodes; i++) v->maxKids = std::max(v->maxKids, child_begin[i + 1] - child_begin[i]);
    while (v->nTop < nSlots && v->nTop < 400 && depth[order[v->nTop]] <= 2) v->nTop++;
    if (std::getenv("ORB_VOCAB_NO_LDS")) v->nTop = 0;
    int rc;
    if ((rc = v->slotDesc.ensure((size_t)32 * nSlots)) != ORB_OK || (rc = v->slotKids.ensure((size_t)8 * nSlots)) != ORB_OK ||
        (rc = v->slotNode.ensure((size_t)4 * nSlots)) != ORB_OK || (rc = v->slotWord.ensure((size_t)4 * nSlots)) != ORB_OK) {
        orb_vocab_destroy(v);
        return rc;
    }
    // (uploads through a stream of their own, not the null stream: see orb_copy_blocking)
    hipStream_t up = nullptr;
    bool upOk = hipStreamCreateWithFlags(&up, hipStreamNonBlocking) == hipSuccess;
    upOk = upOk && orb_copy_blocking(v->slotDesc.p, sdesc.data(), (size_t)32 * nSlots, hipMemcpyHostToDevice, up) == hipSuccess &&
           orb_copy_blocking(v->slotKids.p, kids.data(), (size_t)8 * nSlots, hipMemcpyHostToDevice, up) == hipSuccess &&
           orb_copy_blocking(v->slotNode.p, slotNode.data(), (size_t)4 * nSlots, hipMemcpyHostToDevice, up) == hipSuccess &&
           orb_copy_blocking(v->slotWord.p, sword.data(), (size_t)4 * nSlots, hipMemcpyHostToDevice, up) == hipSuccess;
    if (up) (void)hipStreamDestroy(up);
    if (!upOk) {
        orb_set_error("vocabulary upload failed: %s", hipGetErrorString(hipGetLastError()));
        orb_vocab_destroy(v);
        return ORB_ERR_HIP;
    }
    *out = v;
    return ORB_OK;
}

extern "C" void orb_vocab_destroy(orb_vocab* v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    v->slotDesc.release(); v->slotKids.release(); v->slotNode.release(); v->slotWord.release();
    for (auto& kv : v->compact) kv.second.first.release();
    delete v;
}

// compact index (ascending node id) of the nodes at level L - levelsup: the vocabulary-node space of the matcher
static int compact_table(orb_vocab* v, int levelsup, const int32_t** dTab, int* K)
{
    auto it = v->compact.find(levelsup);
    if (it == v->compact.end()) {
        const int lvl = std::max(v->L - levelsup, 0);
        std::vector<int32_t> tab(v->nNodes, -1);
        int k = 0;
        for (int i = 0; i < v->nNodes; i++)
            if (v->depth[i] == lvl) tab[i] = k++;
        const size_t nSlots = v->slotNodeHost.size();
        std::vector<int32_t> tabSlot(nSlots);                  // indexed by slot: what the descent holds
        for (size_t s2 = 0; s2 < nSlots; s2++) tabSlot[s2] = tab[v->slotNodeHost[s2]];
        std::pair<MBuf, int> entry;
        int rc = entry.first.ensure((size_t)4 * nSlots);
        if (rc != ORB_OK) return rc;
        hipStream_t up = nullptr;
        const bool upOk = hipStreamCreateWithFlags(&up, hipStreamNonBlocking) == hipSuccess &&
                          orb_copy_blocking(entry.first.p, tabSlot.data(), (size_t)4 * nSlots, hipMemcpyHostToDevice, up) == hipSuccess;
        if (up) (void)hipStreamDestroy(up);
        if (!upOk) {
            entry.first.release();
            orb_set_error("vocabulary level table upload failed");
            return ORB_ERR_HIP;
        }
        entry.second = k;
        it = v->compact.emplace(levelsup, entry).first;
    }
    *dTab = (const int32_t*)it->second.first.p;
    *K = it->second.second;
    return ORB_OK;
}

extern "C" int orb_vocab_level_nodes(orb_vocab* v, int levelsup)
{
    if (!v) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(v->device));                          // (compact_table allocates and uploads on first use; found by tests/test_lint_setdevice.py)
    const int32_t* t; int K;
    const int rc = compact_table(v, levelsup, &t, &K);
    return rc == ORB_OK ? K : rc;
}

extern "C" int orb_bow_transform_device(orb_matcher* m, orb_vocab* v, const uint8_t* d_desc, const int32_t* d_counts,
                                        int n_frames, int cap, int levelsup, int32_t* d_word_of, int32_t* d_node_id,
                                        uint16_t* d_node_of)
{
    if (!m || !v || !d_desc || n_frames < 0 || cap <= 0) return ORB_ERR_INVALID;
    if (n_frames == 0) return ORB_OK;
    if (m->device != v->device) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(m->device));
    const int32_t* tab = nullptr;
    int K = 0, rc;
    if (d_node_of) {
        if ((rc = compact_table(v, levelsup, &tab, &K)) != ORB_OK) return rc;
        if (K > 65534) { orb_set_error("more than 65534 vocabulary nodes at that level"); return ORB_ERR_UNSUPPORTED; }
    }
    if (v->maxKids <= 16)
        hipLaunchKernelGGL(k_vocab_transform_k16<2>, dim3((cap + 31) / 32, n_frames), dim3(256), (size_t)v->nTop * 40, m->stream,
                           (const uint4*)v->slotDesc.p, (const int2*)v->slotKids.p, (const int32_t*)v->slotNode.p,
                           (const int32_t*)v->slotWord.p, v->L - levelsup, tab, d_desc, d_counts, cap, d_word_of, d_node_id, d_node_of, v->nTop);
    else
        hipLaunchKernelGGL(k_vocab_transform, dim3((cap + 15) / 16, n_frames), dim3(256), 0, m->stream,
                           (const uint4*)v->slotDesc.p, (const int2*)v->slotKids.p, (const int32_t*)v->slotNode.p,
                           (const int32_t*)v->slotWord.p, v->L - levelsup, tab, d_desc, d_counts, cap, d_word_of, d_node_id, d_node_of);
    ORB_HIP_TRY(hipGetLastError());
    return ORB_OK;
}

extern "C" int orb_bow_transform(orb_matcher* m, orb_vocab* v, const uint8_t* desc, int n, int levelsup, int32_t* word_of,
                                 int32_t* node_id)
{
    if (!m || !v || n < 0) return ORB_ERR_INVALID;
    if (n == 0) return ORB_OK;
    if (!desc || !word_of || !node_id) return ORB_ERR_INVALID;
    ORB_HIP_TRY(hipSetDevice(m->device));
    int rc;
    if ((rc = m->stage[10].ensure((size_t)32 * n)) != ORB_OK || (rc = m->stage[11].ensure((size_t)8 * n)) != ORB_OK) return rc;
    uint8_t* dD = (uint8_t*)m->stage[10].p;
    int32_t* dW = (int32_t*)m->stage[11].p;
    int32_t* dN = dW + n;
    ORB_HIP_TRY(hipMemcpyAsync(dD, desc, (size_t)32 * n, hipMemcpyHostToDevice, m->stream));
    if ((rc = orb_bow_transform_device(m, v, dD, nullptr, 1, n, levelsup, dW, dN, nullptr)) != ORB_OK) return rc;
    ORB_HIP_TRY(hipMemcpyAsync(word_of, dW, (size_t)4 * n, hipMemcpyDeviceToHost, m->stream));
    ORB_HIP_TRY(hipMemcpyAsync(node_id, dN, (size_t)4 * n, hipMemcpyDeviceToHost, m->stream));
    ORB_HIP_TRY(hipStreamSynchronize(m->stream));
    return ORB_OK;
}
