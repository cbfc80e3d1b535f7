// orb_multi.hip -- the batched-frames mode over the GPUs of ONE node, at the product level (C ABI): frames are
// independent (reference src/ORBextractor.cc:1084-1150 keeps no state between calls), so a batch is cut into
// contiguous blocks, one per GPU, with NO data-path collective; the single collective is the start-up broadcast of
// the 1 KiB BRIEF pattern from device 0 over xGMI with RCCL (ncclBroadcast), as BASELINE.json's north_star asks.
// One extractor handle and one host thread per GPU; every thread runs the host-batch pipeline of orb_host_pipe.hip on
// its block and writes straight into the caller's arrays at its frames' offsets, so results are concatenated by
// construction.
// librccl is opened with dlopen at orb_multi_create: liborbhip.so itself carries no link-time dependency on it (the
// library must load on a machine without a GPU for the build / symbol checks).
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "orb_extractor_internal.h"
#include "orb_host_threads.h"

extern "C" void orb_shard_range(int total, int world, int rank, int* first, int* count)
{
    orb_shard_range_impl(total, world, rank, first, count);   // (csrc/orb_host_threads.h: HIP-free, also run by tools/tsan_host.cpp)
}

// the few RCCL entry points, resolved at run time (types as in rccl.h: ncclComm_t is an opaque pointer, results and
// enums are ints; ncclInt8 = 0)
typedef void* rccl_comm_t;
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(rccl_comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load()
    {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && Broadcast && GroupStart && GroupEnd;
    }
};

struct orb_multi {
    std::vector<orb_extractor*> ex;             // one per entry of `devices`
    std::vector<int> devices;
    std::vector<int> uniq;                      // distinct devices, in order of first appearance (the RCCL ranks)
    std::vector<rccl_comm_t> comms;             // one per distinct device
    Rccl rccl;
    std::string lastError;
};

static int broadcast_pattern(orb_multi* m)
{
    // device 0's pattern (rank 0) -> every distinct device with ncclBroadcast; handles sharing a device with an
    // earlier one take a device-to-device copy
    const int nu = (int)m->uniq.size();
    std::vector<int> firstOf(nu, -1);
    for (int i = 0; i < (int)m->ex.size(); i++)
        for (int u = 0; u < nu; u++)
            if (m->devices[i] == m->uniq[u] && firstOf[u] < 0) firstOf[u] = i;
    if (m->rccl.GroupStart() != 0) { orb_set_error("ncclGroupStart failed"); return ORB_ERR_HIP; }
    int grc = ORB_OK;
    for (int u = 0; u < nu && grc == ORB_OK; u++) {
        orb_extractor* h = m->ex[firstOf[u]];
        if (hipSetDevice(h->device) != hipSuccess) { orb_set_error("hipSetDevice(%d) failed", h->device); grc = ORB_ERR_HIP; break; }
        const int r = m->rccl.Broadcast(h->dPattern.p, h->dPattern.p, 1024, /*ncclInt8*/ 0, /*root*/ 0, m->comms[u], h->stream);
        if (r != 0) { orb_set_error("ncclBroadcast: %s", m->rccl.GetErrorString ? m->rccl.GetErrorString(r) : "error"); grc = ORB_ERR_HIP; }
    }
    if (grc != ORB_OK) { (void)m->rccl.GroupEnd(); return grc; }          // never leave the group open behind an error
    if (m->rccl.GroupEnd() != 0) { orb_set_error("ncclGroupEnd failed"); return ORB_ERR_HIP; }
    for (int u = 0; u < nu; u++) {
        orb_extractor* h = m->ex[firstOf[u]];
        ORB_HIP_TRY(hipSetDevice(h->device));
        ORB_HIP_TRY(hipStreamSynchronize(h->stream));
    }
    for (int i = 0; i < (int)m->ex.size(); i++) {
        int u = 0;
        while (m->uniq[u] != m->devices[i]) u++;
        if (firstOf[u] == i) continue;
        const int rc = orb_extractor_set_pattern_device(m->ex[i], (const int8_t*)m->ex[firstOf[u]]->dPattern.p);
        if (rc != ORB_OK) return rc;
    }
    return ORB_OK;
}

extern "C" void orb_multi_destroy(orb_multi* m)
{
    if (!m) return;
    for (size_t u = 0; u < m->comms.size(); u++)
        if (m->comms[u] && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(m->comms[u]);
    for (orb_extractor* h : m->ex) orb_extractor_destroy(h);
    delete m;
}

extern "C" int orb_multi_create(const orb_extractor_params* p, const int* devices, int n_devices, orb_multi** out)
{
    if (!p || !devices || !out || n_devices < 1 || n_devices > 64) return ORB_ERR_INVALID;
    *out = nullptr;
    orb_multi* m = new (std::nothrow) orb_multi();
    if (!m) return ORB_ERR_INTERNAL;
    int rc = ORB_OK;
    for (int i = 0; i < n_devices && rc == ORB_OK; i++) {
        orb_extractor* h = nullptr;
        rc = orb_extractor_create(p, devices[i], &h);
        if (rc == ORB_OK) {
            m->ex.push_back(h);
            m->devices.push_back(devices[i]);
            if (std::find(m->uniq.begin(), m->uniq.end(), devices[i]) == m->uniq.end()) m->uniq.push_back(devices[i]);
        }
    }
    if (rc == ORB_OK && !m->rccl.load()) { orb_set_error("librccl.so not found: %s", dlerror()); rc = ORB_ERR_HIP; }
    if (rc == ORB_OK) {
        m->comms.assign(m->uniq.size(), nullptr);
        const int r = m->rccl.CommInitAll(m->comms.data(), (int)m->uniq.size(), m->uniq.data());
        if (r != 0) { orb_set_error("ncclCommInitAll: %s", m->rccl.GetErrorString ? m->rccl.GetErrorString(r) : "error"); rc = ORB_ERR_HIP; }
    }
    if (rc == ORB_OK) rc = broadcast_pattern(m);              // rank 0 holds the built-in pattern after create
    if (rc != ORB_OK) { orb_multi_destroy(m); return rc; }
    *out = m;
    return ORB_OK;
}

extern "C" int orb_multi_devices(const orb_multi* m) { return m ? (int)m->ex.size() : ORB_ERR_INVALID; }
extern "C" orb_extractor* orb_multi_handle(orb_multi* m, int i) { return (m && i >= 0 && i < (int)m->ex.size()) ? m->ex[i] : nullptr; }

extern "C" int orb_multi_set_pattern(orb_multi* m, const int8_t* pattern)
{
    if (!m || !pattern) return ORB_ERR_INVALID;
    const int rc = orb_extractor_set_pattern(m->ex[0], pattern);      // to rank 0 ...
    return rc != ORB_OK ? rc : broadcast_pattern(m);                   // ... and over xGMI to the others
}

extern "C" int orb_multi_extract_batch(orb_multi* m, const uint8_t* imgs, int n_frames, int rows, int cols, size_t row_stride,
                                       size_t frame_stride, orb_keypoint* kps, uint8_t* desc, int cap, int32_t* counts)
{
    if (!m || n_frames < 0 || !counts) return ORB_ERR_INVALID;
    if (n_frames == 0) return ORB_OK;
    const int W = (int)m->ex.size();
    std::vector<int> rcs;
    std::vector<std::string> errs;
    // one host thread per device, the error of the first failing rank reported (csrc/orb_host_threads.h)
    const int bad = orb_fan_out(W, [=](int r) -> int {
        int first = 0, count = 0;
        orb_shard_range(n_frames, W, r, &first, &count);
        if (count == 0) return ORB_OK;
        return orb_extract_batch(m->ex[r], imgs ? imgs + frame_stride * (size_t)first : nullptr, count, rows, cols, row_stride,
                                 frame_stride, kps ? kps + (size_t)cap * first : nullptr,
                                 desc ? desc + (size_t)ORB_DESC_BYTES * cap * first : nullptr, cap, counts + first);
    }, [] { return std::string(orb_last_error()); }, rcs, errs);
    if (bad >= 0) {
        orb_set_error("device %d (rank %d): %s", m->devices[bad], bad, errs[bad].c_str());
        return rcs[bad];
    }
    return ORB_OK;
}
