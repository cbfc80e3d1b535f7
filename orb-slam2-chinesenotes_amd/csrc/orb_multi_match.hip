// orb_multi_match.hip -- BASELINE configs[4] over the GPUs of one node: a keyframe descriptor database sharded BY KEYFRAME
// ("each (query, KF) result is independent; results are concatenated on the host"), the query frame replicated,
// no collective at all.  What runs per stream frame is the Relocalization / loop-candidate loop of the reference
// (src/Tracking.cc:1471-1492: one ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) per candidate keyframe,
// src/ORBmatcher.cc:552-687) as ONE orb_match_bow_batch_device per shard.
// A database object owns, per entry of the orb_multi's device list, a matcher handle and the shard's slice of the
// feature store (descriptors, keypoints, valid flags, counts, vocabulary node per feature and the per-frame feature
// vectors as CSR, built once) plus one extra frame slot that receives the query.  One host thread per shard, each
// writing its keyframes' rows of the caller's result arrays.
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "orb_matcher_internal.h"
#include "orb_host_threads.h"

struct OrbDbShard {
    int device = 0, first = 0, count = 0;      // keyframes [first, first + count) of the database
    orb_matcher* mt = nullptr;
    MBuf desc, kps, valid, counts, nodeOf, ckeys, cstart, ccnt, cdesc, kfIdx, fIdx, match, nm;
    orb_featstore store;
};

struct orb_multi_db {
    std::vector<OrbDbShard> sh;
    int nKf = 0, cap = 0, nNodes = 0;
};

extern "C" void orb_multi_db_destroy(orb_multi_db* db)
{
    if (!db) return;
    for (OrbDbShard& s : db->sh) {
        (void)hipSetDevice(s.device);
        if (s.mt) { (void)orb_matcher_sync(s.mt); orb_matcher_destroy(s.mt); }
        MBuf* bufs[] = {&s.desc, &s.kps, &s.valid, &s.counts, &s.nodeOf, &s.ckeys, &s.cstart, &s.ccnt, &s.cdesc, &s.kfIdx, &s.fIdx, &s.match, &s.nm};
        for (MBuf* b : bufs) b->release();
    }
    delete db;
}

static int build_shard(OrbDbShard& s, const uint8_t* desc, const orb_keypoint* kps, const uint8_t* valid, const int32_t* counts,
                       const uint16_t* nodeOf, int cap, int nNodes)
{
    ORB_HIP_TRY(hipSetDevice(s.device));
    int rc = orb_matcher_create(s.device, &s.mt);
    if (rc != ORB_OK) return rc;
    const size_t F = (size_t)s.count + 1;                          // + the query slot
    const size_t c = (size_t)cap;
    if ((rc = s.desc.ensure(F * c * ORB_DESC_BYTES)) != ORB_OK || (rc = s.kps.ensure(F * c * sizeof(orb_keypoint))) != ORB_OK ||
        (rc = s.valid.ensure(F * c)) != ORB_OK || (rc = s.counts.ensure(F * 4)) != ORB_OK || (rc = s.nodeOf.ensure(F * c * 2)) != ORB_OK ||
        (rc = s.ckeys.ensure(F * c * 4)) != ORB_OK || (rc = s.cstart.ensure(F * (size_t)nNodes * 2)) != ORB_OK ||
        (rc = s.ccnt.ensure(F * (size_t)nNodes * 2)) != ORB_OK || (rc = s.cdesc.ensure(F * c * ORB_DESC_BYTES)) != ORB_OK || (rc = s.kfIdx.ensure(std::max<size_t>(s.count, 1) * 4)) != ORB_OK ||
        (rc = s.fIdx.ensure(std::max<size_t>(s.count, 1) * 4)) != ORB_OK || (rc = s.match.ensure(std::max<size_t>(s.count, 1) * c * 4)) != ORB_OK ||
        (rc = s.nm.ensure(std::max<size_t>(s.count, 1) * 4)) != ORB_OK)
        return rc;
    hipStream_t st = (hipStream_t)orb_matcher_stream(s.mt);
    const size_t f0 = (size_t)s.first, n = (size_t)s.count;
    ORB_HIP_TRY(hipMemsetAsync(s.valid.p, 1, F * c, st));           // valid == NULL: every feature has a good MapPoint; the query slot: unused
    ORB_HIP_TRY(hipMemsetAsync(s.counts.p, 0, F * 4, st));
    if (n) {
        ORB_HIP_TRY(hipMemcpyAsync(s.desc.p, desc + f0 * c * ORB_DESC_BYTES, n * c * ORB_DESC_BYTES, hipMemcpyHostToDevice, st));
        ORB_HIP_TRY(hipMemcpyAsync(s.kps.p, kps + f0 * c, n * c * sizeof(orb_keypoint), hipMemcpyHostToDevice, st));
        if (valid) ORB_HIP_TRY(hipMemcpyAsync(s.valid.p, valid + f0 * c, n * c, hipMemcpyHostToDevice, st));
        ORB_HIP_TRY(hipMemcpyAsync(s.counts.p, counts + f0, n * 4, hipMemcpyHostToDevice, st));
        ORB_HIP_TRY(hipMemcpyAsync(s.nodeOf.p, nodeOf + f0 * c, n * c * 2, hipMemcpyHostToDevice, st));
        std::vector<int32_t> kf(n), fq(n, (int32_t)n);              // pair p: keyframe p of the shard against the query slot
        for (size_t i = 0; i < n; i++) kf[i] = (int32_t)i;
        ORB_HIP_TRY(hipMemcpyAsync(s.kfIdx.p, kf.data(), n * 4, hipMemcpyHostToDevice, st));
        ORB_HIP_TRY(hipMemcpyAsync(s.fIdx.p, fq.data(), n * 4, hipMemcpyHostToDevice, st));
        ORB_HIP_TRY(hipStreamSynchronize(st));                      // the index vectors go out of scope
        // the keyframes' feature vectors, once (the reference computes them once per KeyFrame, src/KeyFrame.cc:70)
        rc = orb_bow_build_csr_desc_device(s.mt, (const uint16_t*)s.nodeOf.p, (const int32_t*)s.counts.p, (const uint8_t*)s.desc.p, (int)n,
                                           cap, nNodes, (uint32_t*)s.ckeys.p, (uint16_t*)s.cstart.p, (uint16_t*)s.ccnt.p, (uint8_t*)s.cdesc.p);
        if (rc != ORB_OK) return rc;
    }
    s.store.desc = (const uint8_t*)s.desc.p;
    s.store.kps = (const orb_keypoint*)s.kps.p;
    s.store.valid = (const uint8_t*)s.valid.p;
    s.store.counts = (const int32_t*)s.counts.p;
    s.store.node_of = (const uint16_t*)s.nodeOf.p;
    s.store.cap = cap;
    s.store.n_frames = (int32_t)F;
    s.store.n_nodes = nNodes;
    s.store.csr_keys = (const uint32_t*)s.ckeys.p;
    s.store.csr_start = (const uint16_t*)s.cstart.p;
    s.store.csr_cnt = (const uint16_t*)s.ccnt.p;
    s.store.csr_desc = (const uint8_t*)s.cdesc.p;
    return orb_matcher_sync(s.mt);
}

extern "C" int orb_multi_db_create(const int* devices, int n_devices, const uint8_t* desc, const orb_keypoint* kps,
                                   const uint8_t* valid, const int32_t* counts, const uint16_t* node_of, int n_kf, int cap,
                                   int n_nodes, orb_multi_db** out)
{
    if (!devices || n_devices < 1 || n_devices > 64 || !out || n_kf < 0 || cap <= 0 || n_nodes <= 0) return ORB_ERR_INVALID;
    if (n_kf > 0 && (!desc || !kps || !counts || !node_of)) return ORB_ERR_INVALID;
    if (cap > 8192) { orb_set_error("featstore cap must be 1..8192"); return ORB_ERR_UNSUPPORTED; }
    *out = nullptr;
    orb_multi_db* db = new (std::nothrow) orb_multi_db();
    if (!db) return ORB_ERR_INTERNAL;
    db->nKf = n_kf; db->cap = cap; db->nNodes = n_nodes;
    db->sh.resize(n_devices);
    for (int r = 0; r < n_devices; r++) {
        OrbDbShard& s = db->sh[r];
        s.device = devices[r];
        orb_shard_range(n_kf, n_devices, r, &s.first, &s.count);
    }
    std::vector<int> rcs;
    std::vector<std::string> errs;
    {
        const int bad = orb_fan_out(n_devices, [=](int r) -> int { return build_shard(db->sh[r], desc, kps, valid, counts, node_of, cap, n_nodes); },
                                    [] { return std::string(orb_last_error()); }, rcs, errs);
        if (bad >= 0) {
            orb_set_error("device %d (shard %d): %s", devices[bad], bad, errs[bad].c_str());
            const int rc = rcs[bad];
            orb_multi_db_destroy(db);
            return rc;
        }
    }
    *out = db;
    return ORB_OK;
}

extern "C" int orb_multi_db_shards(const orb_multi_db* db) { return db ? (int)db->sh.size() : ORB_ERR_INVALID; }

static int query_shard(OrbDbShard& s, int cap, int nNodes, const uint8_t* qDesc, const orb_keypoint* qKps, int qCount,
                       const uint16_t* qNodeOf, float ratio, int checkOri, int32_t* match, int32_t* nmatches)
{
    if (s.count == 0) return ORB_OK;
    ORB_HIP_TRY(hipSetDevice(s.device));
    hipStream_t st = (hipStream_t)orb_matcher_stream(s.mt);
    const size_t slot = (size_t)s.count, c = (size_t)cap, nq = (size_t)qCount;
    if (nq) {
        ORB_HIP_TRY(hipMemcpyAsync((uint8_t*)s.desc.p + slot * c * ORB_DESC_BYTES, qDesc, nq * ORB_DESC_BYTES, hipMemcpyHostToDevice, st));
        ORB_HIP_TRY(hipMemcpyAsync((orb_keypoint*)s.kps.p + slot * c, qKps, nq * sizeof(orb_keypoint), hipMemcpyHostToDevice, st));
        ORB_HIP_TRY(hipMemcpyAsync((uint16_t*)s.nodeOf.p + slot * c, qNodeOf, nq * 2, hipMemcpyHostToDevice, st));
    }
    const int32_t qc = qCount;
    ORB_HIP_TRY(hipMemcpyAsync((int32_t*)s.counts.p + slot, &qc, 4, hipMemcpyHostToDevice, st));
    ORB_HIP_TRY(hipStreamSynchronize(st));                          // &qc is a stack variable
    int rc = orb_bow_build_csr_desc_device(s.mt, (const uint16_t*)s.nodeOf.p + slot * c, (const int32_t*)s.counts.p + slot,
                                           (const uint8_t*)s.desc.p + slot * c * ORB_DESC_BYTES, 1, cap, nNodes,
                                           (uint32_t*)s.ckeys.p + slot * c, (uint16_t*)s.cstart.p + slot * (size_t)nNodes,
                                           (uint16_t*)s.ccnt.p + slot * (size_t)nNodes, (uint8_t*)s.cdesc.p + slot * c * ORB_DESC_BYTES);
    if (rc != ORB_OK) return rc;
    // one query against the shard's keyframes: the candidate loop as ONE launch pair (orb_matcher_query.hip)
    rc = orb_match_bow_query_device(s.mt, &s.store, (const int32_t*)s.kfIdx.p, s.count, (const int32_t*)s.fIdx.p, 1, ratio, checkOri,
                                    (int32_t*)s.match.p, (int32_t*)s.nm.p);
    if (rc != ORB_OK) return rc;
    ORB_HIP_TRY(hipMemcpyAsync(match + (size_t)s.first * c, s.match.p, (size_t)s.count * c * 4, hipMemcpyDeviceToHost, st));
    ORB_HIP_TRY(hipMemcpyAsync(nmatches + s.first, s.nm.p, (size_t)s.count * 4, hipMemcpyDeviceToHost, st));
    ORB_HIP_TRY(hipStreamSynchronize(st));
    return ORB_OK;
}

extern "C" int orb_multi_match_bow_batch(orb_multi_db* db, const uint8_t* q_desc, const orb_keypoint* q_kps, int q_count,
                                         const uint16_t* q_node_of, float ratio, int check_ori, int32_t* match, int32_t* nmatches)
{
    if (!db || q_count < 0 || q_count > db->cap || !match || !nmatches) return ORB_ERR_INVALID;
    if (q_count > 0 && (!q_desc || !q_kps || !q_node_of)) return ORB_ERR_INVALID;
    const int W = (int)db->sh.size();
    std::vector<int> rcs;
    std::vector<std::string> errs;
    const int bad = orb_fan_out(W, [=](int r) -> int {
        return query_shard(db->sh[r], db->cap, db->nNodes, q_desc, q_kps, q_count, q_node_of, ratio, check_ori, match, nmatches);
    }, [] { return std::string(orb_last_error()); }, rcs, errs);
    if (bad >= 0) {
        orb_set_error("device %d (shard %d): %s", db->sh[bad].device, bad, errs[bad].c_str());
        return rcs[bad];
    }
    return ORB_OK;
}
