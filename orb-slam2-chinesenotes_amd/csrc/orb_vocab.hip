// orb_vocab.hip -- DBoW2 vocabulary-tree descent on gfx950: the part of
// TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) that touches descriptors, as called
// by Frame::ComputeBoW (reference src/Frame.cc:425-433) and KeyFrame::ComputeBoW (src/KeyFrame.cc:70): per feature
// the word (leaf) it falls into and the node it passes at level L - levelsup.  DBoW2 itself is absent from the
// reference tree; the algorithm is restated from its published source (first-minimum Hamming descent, children in
// stored order).  Building the std::map containers (and the tf-idf BowVector, which uses doubles) stays on the host.
#include <algorithm>
#include <map>
#include <new>
#include <vector>

#include "orb_matcher_internal.h"

struct orb_vocab {
    int device = 0, nNodes = 0, L = 0;
    MBuf desc, childBegin, children, wordId;
    std::vector<int> depth;                         // host: depth of every node
    std::map<int, std::pair<MBuf, int>> compact;    // levelsup -> (device int32 compactOf[nNodes], K)
};

__global__ __launch_bounds__(256) void k_vocab_transform(const uint8_t* __restrict__ nodeDesc,
                                                         const int32_t* __restrict__ childBegin,
                                                         const int32_t* __restrict__ children,
                                                         const int32_t* __restrict__ wordId, int nidLevel,
                                                         const int32_t* __restrict__ compactOf,
                                                         const uint8_t* __restrict__ desc,
                                                         const int32_t* __restrict__ counts, int cap,
                                                         int32_t* __restrict__ wordOf, int32_t* __restrict__ nodeId,
                                                         uint16_t* __restrict__ nodeOf)
{
    const int f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const size_t row = (size_t)f * cap + i;
    const int n = counts ? counts[f] : cap;
    if (i >= n) {
        if (nodeOf) nodeOf[row] = 0xFFFF;
        return;
    }
    const uint4 lo = reinterpret_cast<const uint4*>(desc + row * 32)[0], hi = reinterpret_cast<const uint4*>(desc + row * 32)[1];
    int finalId = 0, level = 0, nid = (nidLevel <= 0) ? 0 : -1;
    int b = childBegin[0], e = childBegin[1];
    while (e > b) {                                               // do { ... } while (!isLeaf())
        ++level;
        int best = 257, bestId = -1;
        for (int c = b; c < e; c++) {
            const int id = children[c];
            const uint4 nl = reinterpret_cast<const uint4*>(nodeDesc + (size_t)id * 32)[0], nh = reinterpret_cast<const uint4*>(nodeDesc + (size_t)id * 32)[1];
            const int h = __popc(lo.x ^ nl.x) + __popc(lo.y ^ nl.y) + __popc(lo.z ^ nl.z) + __popc(lo.w ^ nl.w) +
                          __popc(hi.x ^ nh.x) + __popc(hi.y ^ nh.y) + __popc(hi.z ^ nh.z) + __popc(hi.w ^ nh.w);
            if (h < best) { best = h; bestId = id; }              // first minimum wins
        }
        finalId = bestId;
        if (level == nidLevel) nid = finalId;
        b = childBegin[finalId];
        e = childBegin[finalId + 1];
    }
    if (wordOf) wordOf[row] = wordId[finalId];
    if (nodeId) nodeId[row] = nid;
    if (nodeOf) nodeOf[row] = (nid >= 0 && compactOf) ? (uint16_t)compactOf[nid] : (uint16_t)0xFFFF;
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
    orb_vocab* v = new (std::nothrow) orb_vocab();
    if (!v) return ORB_ERR_INTERNAL;
    v->device = device; v->nNodes = n_nodes; v->L = L; v->depth = depth;
    int rc;
    if ((rc = v->desc.ensure((size_t)32 * n_nodes)) != ORB_OK || (rc = v->childBegin.ensure((size_t)4 * (n_nodes + 1))) != ORB_OK ||
        (rc = v->children.ensure((size_t)4 * nch)) != ORB_OK || (rc = v->wordId.ensure((size_t)4 * n_nodes)) != ORB_OK) {
        delete v;
        return rc;
    }
    ORB_HIP_TRY(hipMemcpy(v->desc.p, node_desc, (size_t)32 * n_nodes, hipMemcpyHostToDevice));
    ORB_HIP_TRY(hipMemcpy(v->childBegin.p, child_begin, (size_t)4 * (n_nodes + 1), hipMemcpyHostToDevice));
    ORB_HIP_TRY(hipMemcpy(v->children.p, children, (size_t)4 * nch, hipMemcpyHostToDevice));
    ORB_HIP_TRY(hipMemcpy(v->wordId.p, word_id, (size_t)4 * n_nodes, hipMemcpyHostToDevice));
    *out = v;
    return ORB_OK;
}

extern "C" void orb_vocab_destroy(orb_vocab* v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    v->desc.release(); v->childBegin.release(); v->children.release(); v->wordId.release();
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
        std::pair<MBuf, int> entry;
        int rc = entry.first.ensure((size_t)4 * v->nNodes);
        if (rc != ORB_OK) return rc;
        ORB_HIP_TRY(hipMemcpy(entry.first.p, tab.data(), (size_t)4 * v->nNodes, hipMemcpyHostToDevice));
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
    hipLaunchKernelGGL(k_vocab_transform, dim3((cap + 255) / 256, n_frames), dim3(256), 0, m->stream,
                       (const uint8_t*)v->desc.p, (const int32_t*)v->childBegin.p, (const int32_t*)v->children.p,
                       (const int32_t*)v->wordId.p, v->L - levelsup, tab, d_desc, d_counts, cap, d_word_of, d_node_id, d_node_of);
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
