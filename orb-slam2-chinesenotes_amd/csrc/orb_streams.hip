// orb_streams.hip -- the handles' streams, spread over the GPU's hardware queues ON PURPOSE.
//
// HIP deals the streams of a process over a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) by a policy of its own;
// streams that land on the same queue run one after the other, whatever the program meant.  Round 5 found that this -- not the
// kernels -- decided how much the library's pipelines overlap: config 5 (two extractor and two matcher handles) ran at 0.049
// or 0.080 or 0.095 ms per frame depending on how many OTHER streams had been created (and destroyed) before its handles;
// an idle extra stream per handle took the 64-frame batch from 349 k to 251 k frames/s; more hardware queues made
// everything slower (tools/experiments/hwq.sh, c5_pads.sh).  So the library finds out which queue a stream is on and chooses:
//
//   * once per device, REFERENCE streams are found, one per hardware queue: streams are created until `maxQ` of them are
//     pairwise concurrent (probe below); they stay alive, idle, for the life of the process;
//   * a handle's stream is picked among a few candidate streams: each candidate is classified by the reference stream it is
//     NOT concurrent with (= its queue), the one on the queue that the fewest library streams OF THE SAME ROLE use is kept
//     (ties: fewest library streams of any role), the others are destroyed.  Handles of one role (extractors, matchers) thus
//     spread over the queues first -- two matcher handles never share a queue while a free one exists -- and an extractor
//     shares a queue with a matcher rather than with another extractor.
//
// probe(a, b): a kernel that spins ~150 us on `a`, a kernel that stamps the clock on `b`, launched in that order; the streams
// are concurrent iff b's stamp is earlier than the end of a's spin.  Reference streams carry no work of the library, so
// probing never delays a running pipeline.  ORB_STREAM_BALANCE=0 switches all of it off (plain hipStreamCreateWithFlags).
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "orb_common.h"

__global__ void k_stream_spin(unsigned long long* ts, unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    ts[0] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    ts[1] = __builtin_amdgcn_s_memrealtime();
}

__global__ void k_stream_stamp(unsigned long long* ts) { ts[0] = __builtin_amdgcn_s_memrealtime(); }

namespace {

constexpr int kMaxQueues = 8, kRoles = 4;

struct DeviceQueues {
    bool ready = false, usable = false;
    std::vector<hipStream_t> ref;                   // one idle stream per hardware queue found
    int usedRole[kMaxQueues][kRoles] = {};
    int usedAll[kMaxQueues] = {};
    std::map<hipStream_t, std::pair<int, int>> owner;   // library stream -> (queue, role)
    unsigned long long* ts = nullptr;               // pinned, device-visible: [0..1] spin start / end, [2] stamp
};

std::mutex g_mu;
std::map<int, DeviceQueues> g_dev;

// are a and b concurrent (different hardware queues)?  -1: the probe itself failed
int probe(DeviceQueues& D, hipStream_t a, hipStream_t b)
{
    D.ts[0] = D.ts[1] = D.ts[2] = 0;
    hipLaunchKernelGGL(k_stream_spin, dim3(1), dim3(1), 0, a, D.ts, 15000ull);      // 150 us of the 100 MHz clock
    hipLaunchKernelGGL(k_stream_stamp, dim3(1), dim3(1), 0, b, D.ts + 2);
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (D.ts[0] == 0 || D.ts[1] == 0 || D.ts[2] == 0) return -1;
    return D.ts[2] < D.ts[1] ? 1 : 0;
}

// the queue (index into D.ref) of stream s; a stream concurrent with every reference is on a queue of its own: -1
int classify(DeviceQueues& D, hipStream_t s)
{
    for (size_t q = 0; q < D.ref.size(); q++) {
        const int c = probe(D, D.ref[q], s);
        if (c < 0) return -2;
        if (c == 0) return (int)q;
    }
    return -1;
}

void discover(DeviceQueues& D)
{
    D.ready = true;
    if (hipHostMalloc((void**)&D.ts, 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return; }
    int maxQ = 4;
    if (const char* e = std::getenv("GPU_MAX_HW_QUEUES")) maxQ = std::max(1, std::min(kMaxQueues, std::atoi(e)));
    std::vector<hipStream_t> extra;
    for (int tries = 0; tries < 4 * maxQ && (int)D.ref.size() < maxQ; tries++) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        const int q = classify(D, s);
        if (q == -2) { extra.push_back(s); break; }
        if (q == -1) D.ref.push_back(s); else extra.push_back(s);
    }
    for (hipStream_t s : extra) (void)hipStreamDestroy(s);
    D.usable = D.ref.size() >= 2;
    if (std::getenv("ORB_STREAM_DEBUG")) std::fprintf(stderr, "[orb] hardware queues found: %zu\n", D.ref.size());
}

}  // namespace

// role: 0 extractor, 1 matcher, 2 copy / side streams, 3 other
hipError_t orb_stream_create(hipStream_t* out, int device, int role)
{
    *out = nullptr;
    static const bool balance = [] { const char* e = std::getenv("ORB_STREAM_BALANCE"); return !e || std::atoi(e) != 0; }();
    if (!balance) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    std::lock_guard<std::mutex> lock(g_mu);
    DeviceQueues& D = g_dev[device];
    if (!D.ready) discover(D);
    if (!D.usable) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    role = std::max(0, std::min(kRoles - 1, role));
    const int nq = (int)D.ref.size();
    auto cost = [&](int q) { return D.usedRole[q][role] * 1000 + D.usedAll[q]; };
    int bestCost = cost(0);
    for (int q = 1; q < nq; q++) bestCost = std::min(bestCost, cost(q));
    std::vector<std::pair<hipStream_t, int>> cand;
    hipStream_t pick = nullptr;
    int pickQ = -1;
    for (int tries = 0; tries < 3 * nq && !pick; tries++) {
        hipStream_t s = nullptr;
        const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e != hipSuccess) { for (auto& c : cand) (void)hipStreamDestroy(c.first); return e; }
        const int q = classify(D, s);
        if (q >= 0 && cost(q) == bestCost) { pick = s; pickQ = q; }
        else cand.push_back({s, q});
    }
    if (!pick) {                                    // no candidate on a least-used queue: the best of what came
        int bi = -1;
        for (size_t i = 0; i < cand.size(); i++)
            if (cand[i].second >= 0 && (bi < 0 || cost(cand[i].second) < cost(cand[(size_t)bi].second))) bi = (int)i;
        if (bi < 0) bi = 0;
        pick = cand[(size_t)bi].first;
        pickQ = cand[(size_t)bi].second;
        cand.erase(cand.begin() + bi);
    }
    for (auto& c : cand) (void)hipStreamDestroy(c.first);
    if (pickQ >= 0) { D.usedRole[pickQ][role]++; D.usedAll[pickQ]++; D.owner[pick] = {pickQ, role}; }
    if (std::getenv("ORB_STREAM_DEBUG")) std::fprintf(stderr, "[orb] stream %p: role %d on hardware queue %d\n", (void*)pick, role, pickQ);
    *out = pick;
    return hipSuccess;
}

void orb_stream_destroy(hipStream_t s, int device)
{
    if (!s) return;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_dev.find(device);
        if (it != g_dev.end()) {
            auto o = it->second.owner.find(s);
            if (o != it->second.owner.end()) {
                it->second.usedRole[o->second.first][o->second.second]--;
                it->second.usedAll[o->second.first]--;
                it->second.owner.erase(o);
            }
        }
    }
    (void)hipStreamDestroy(s);
}
