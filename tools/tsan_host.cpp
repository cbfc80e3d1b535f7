// tsan_host.cpp -- the library's HOST-THREAD logic (csrc/orb_host_threads.h: the chunk pipeline of orb_extract_batch for large
// host batches with its copy threads, the per-device fan-out of orb_multi_* with its error merge, the batch partition) run
// against a FAKE device under -fsanitize=thread, on a machine without a GPU ("run host
// code under TSan").  make -C orb-slam2-chinesenotes_amd tsan-host builds and runs it; tests/test_tsan_host.py is the CPU-suite
// hook.
//
// The fake device: a STREAM is a worker thread with a FIFO of closures, an EVENT is a (mutex, condition variable, counter) that
// a stream signals when it reaches the record and that streams or the host wait for; "copies" are memcpys and the "kernel
// chain" a per-frame checksum, all run by the stream threads.  The pipeline code under test is exactly what
// csrc/orb_host_pipe.hip runs (orb_pipe_run with the same slot / event protocol); a protocol violation -- a slot's staging
// rewritten while its upload is in flight, results unpacked before the download finished -- is a data race between a host
// thread and a stream thread here, which the sanitizer reports.  Results are also checked for content.
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../orb-slam2-chinesenotes_amd/csrc/orb_host_threads.h"

namespace {

struct Event {
    std::mutex m;
    std::condition_variable cv;
    unsigned long long done = 0, recorded = 0;       // generations: a wait captures the generation recorded so far
};

struct Stream {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false;
    std::thread th;
    Stream() : th([this] { run(); }) {}
    ~Stream()
    {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv.notify_all();
        th.join();
    }
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [this] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
            }
            f();
        }
    }
    void push(std::function<void()> f)
    {
        { std::lock_guard<std::mutex> l(m); q.push_back(std::move(f)); }
        cv.notify_one();
    }
    void record(Event& e)
    {
        unsigned long long gen;
        { std::lock_guard<std::mutex> l(e.m); gen = ++e.recorded; }
        push([&e, gen] {
            std::lock_guard<std::mutex> l(e.m);
            e.done = std::max(e.done, gen);
            e.cv.notify_all();
        });
    }
    void wait(Event& e)                               // the stream waits for what has been recorded on e so far
    {
        unsigned long long gen;
        { std::lock_guard<std::mutex> l(e.m); gen = e.recorded; }
        push([&e, gen] {
            std::unique_lock<std::mutex> l(e.m);
            e.cv.wait(l, [&e, gen] { return e.done >= gen; });
        });
    }
    void sync()                                       // (its event outlives the signalling stream thread's last touch: shared)
    {
        auto e = std::make_shared<Event>();
        push([e] {
            std::lock_guard<std::mutex> l(e->m);
            e->done = 1;
            e->cv.notify_all();
        });
        std::unique_lock<std::mutex> l(e->m);
        e->cv.wait(l, [&e] { return e->done >= 1; });
    }
};

void host_wait(Event& e)
{
    unsigned long long gen;
    { std::lock_guard<std::mutex> l(e.m); gen = e.recorded; }
    std::unique_lock<std::mutex> l(e.m);
    e.cv.wait(l, [&e, gen] { return e.done >= gen; });
}

uint32_t checksum(const uint8_t* p, size_t n, uint32_t seed)
{
    uint32_t h = 2166136261u ^ seed;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 16777619u;
    return h;
}

constexpr int NS = 3;
constexpr int OUT_WORDS = 64;                          // "keypoints + descriptors" of a frame: 64 words derived from its pixels

// the fake counterpart of HipOps in csrc/orb_host_pipe.hip
struct FakeOps {
    const uint8_t* imgs;                               // caller's (pageable) images
    uint32_t* results;                                 // caller's results, OUT_WORDS per frame
    int32_t* counts;
    size_t imgBytes;
    int C;
    bool inPinned = false, outPinned = false, failAt = false;
    int failChunkFrame = -1;                           // extract() fails for the chunk that starts at this frame
    Stream h2d, cs, d2h;
    Event evIn[NS], evK[NS], evOut[NS];
    std::vector<uint8_t> pinIn[NS], dImg[NS];
    std::vector<uint32_t> dOut[NS], pinOut[NS];
    std::vector<int32_t> dCnt[NS], pinCnt[NS];
    int chunkFirst[NS] = {-1, -1, -1};
    std::atomic<int> drained{0};

    FakeOps(const uint8_t* im, uint32_t* res, int32_t* cnt, size_t ib, int c) : imgs(im), results(res), counts(cnt), imgBytes(ib), C(c)
    {
        for (int s = 0; s < NS; s++) {
            pinIn[s].resize(imgBytes * C); dImg[s].resize(imgBytes * C);
            dOut[s].resize((size_t)OUT_WORDS * C); pinOut[s].resize((size_t)OUT_WORDS * C);
            dCnt[s].resize(C); pinCnt[s].resize(C);
        }
    }
    bool in_pinned() const { return inPinned; }
    bool out_pinned() const { return outPinned; }
    size_t in_bytes_per_frame() const { return std::max<size_t>(imgBytes, (size_t)3 << 20); }   // (pretend: so that several copy threads start)
    size_t out_bytes_per_frame() const { return (size_t)3 << 20; }
    void stage_frame(int s, int f, int frame) { std::memcpy(pinIn[s].data() + imgBytes * f, imgs + imgBytes * frame, imgBytes); }
    int upload(int s, int f0, int c)
    {
        const uint8_t* src = inPinned ? imgs + imgBytes * f0 : pinIn[s].data();
        uint8_t* dst = dImg[s].data();
        const size_t n = imgBytes * c;
        h2d.push([=] { std::memcpy(dst, src, n); });
        return 0;
    }
    int mark_uploaded(int s) { h2d.record(evIn[s]); return 0; }
    // TSAN_HOST_BREAK=1 (tests/test_tsan_host.py): the compute stream does NOT wait for the upload -- the sanitizer must notice
    int compute_waits_upload(int s) { if (!std::getenv("TSAN_HOST_BREAK")) cs.wait(evIn[s]); return 0; }
    int compute_waits_download(int s) { cs.wait(evOut[s]); return 0; }
    int extract(int s, int c)
    {
        if (failChunkFrame >= 0 && chunkFirstOf(s) == failChunkFrame) return -7;
        const uint8_t* src = dImg[s].data();
        uint32_t* out = dOut[s].data();
        int32_t* cnt = dCnt[s].data();
        const size_t ib = imgBytes;
        cs.push([=] {
            for (int f = 0; f < c; f++) {
                for (int w = 0; w < OUT_WORDS; w++) out[(size_t)f * OUT_WORDS + w] = checksum(src + ib * f, ib, (uint32_t)w);
                cnt[f] = (int32_t)(checksum(src + ib * f, ib, 999u) % OUT_WORDS) + 1;
            }
        });
        return 0;
    }
    int mark_computed(int s) { cs.record(evK[s]); return 0; }
    int download_waits_compute(int s) { d2h.wait(evK[s]); return 0; }
    int download(int s, int f0, int c)
    {
        const uint32_t* so = dOut[s].data();
        const int32_t* sc = dCnt[s].data();
        uint32_t* po = outPinned ? results + (size_t)OUT_WORDS * f0 : pinOut[s].data();
        int32_t* pc = pinCnt[s].data();
        d2h.push([=] {
            std::memcpy(pc, sc, sizeof(int32_t) * c);
            std::memcpy(po, so, sizeof(uint32_t) * OUT_WORDS * c);
        });
        return 0;
    }
    int mark_downloaded(int s) { d2h.record(evOut[s]); return 0; }
    int wait_downloaded(int s) { host_wait(evOut[s]); return 0; }
    int finish(int s, int f0, int c)
    {
        std::memcpy(counts + f0, pinCnt[s].data(), sizeof(int32_t) * c);
        return 0;
    }
    void unpack_frame(int s, int f, int frame)
    {
        std::memcpy(results + (size_t)OUT_WORDS * frame, pinOut[s].data() + (size_t)OUT_WORDS * f, sizeof(uint32_t) * counts[frame]);
    }
    void drain() { h2d.sync(); cs.sync(); d2h.sync(); drained++; }
    // (which chunk a slot holds: set by the test driver through the frame index passed to upload)
    int chunkFirstOf(int s) const { return chunkFirst[s]; }
};

// upload() learns the chunk's first frame; wrap to record it for the failure injection
struct FakeOpsTracked : FakeOps {
    using FakeOps::FakeOps;
    int upload(int s, int f0, int c) { chunkFirst[s] = f0; return FakeOps::upload(s, f0, c); }
};

int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); fails++; } } while (0)

void test_pipeline(int nFrames, int C, bool inPinned, bool outPinned)
{
    const size_t imgBytes = 4096 + 17;
    std::vector<uint8_t> imgs(imgBytes * nFrames);
    for (size_t i = 0; i < imgs.size(); i++) imgs[i] = (uint8_t)((i * 2654435761u) >> 13);
    std::vector<uint32_t> res((size_t)OUT_WORDS * nFrames, 0xDEADBEEFu);
    std::vector<int32_t> counts(nFrames, -1);
    FakeOpsTracked ops(imgs.data(), res.data(), counts.data(), imgBytes, C);
    ops.inPinned = inPinned; ops.outPinned = outPinned;
    const int rc = orb_pipe_run(ops, nFrames, C, NS);
    CHECK(rc == 0, "pipeline rc %d", rc);
    for (int f = 0; f < nFrames; f++) {
        const int want = (int)(checksum(imgs.data() + imgBytes * f, imgBytes, 999u) % OUT_WORDS) + 1;
        CHECK(counts[f] == want, "frame %d: count %d, want %d", f, counts[f], want);
        const int nw = outPinned ? OUT_WORDS : want;   // (pageable results: only the frame's own `count` words are copied out)
        for (int w = 0; w < nw; w++)
            CHECK(res[(size_t)f * OUT_WORDS + w] == checksum(imgs.data() + imgBytes * f, imgBytes, (uint32_t)w), "frame %d word %d", f, w);
    }
}

void test_pipeline_error(int nFrames, int C, int failFrame)
{
    const size_t imgBytes = 1024;
    std::vector<uint8_t> imgs(imgBytes * nFrames, 3);
    std::vector<uint32_t> res((size_t)OUT_WORDS * nFrames, 0);
    std::vector<int32_t> counts(nFrames, -1);
    FakeOpsTracked ops(imgs.data(), res.data(), counts.data(), imgBytes, C);
    ops.failChunkFrame = failFrame;
    const int rc = orb_pipe_run(ops, nFrames, C, NS);
    CHECK(rc == -7, "an error of chunk %d must come back (rc %d)", failFrame / C, rc);
    CHECK(ops.drained.load() == 1, "the streams are drained once behind an error");
    for (int f = 0; f < std::max(0, failFrame - 2 * C); f++) CHECK(counts[f] >= 1, "chunks issued before the failure are retired (frame %d)", f);
}

void test_par_items()
{
    for (int n : {1, 2, 7, 64, 257}) {
        std::vector<std::vector<uint8_t>> src(n, std::vector<uint8_t>(1000)), dst(n, std::vector<uint8_t>(1000, 0));
        for (int i = 0; i < n; i++) std::memset(src[i].data(), i & 255, 1000);
        std::atomic<int> calls{0};
        orb_par_items(n, (size_t)4 << 20, [&](int i) { std::memcpy(dst[i].data(), src[i].data(), 1000); calls++; });
        CHECK(calls.load() == n, "par_items(%d): %d calls", n, calls.load());
        for (int i = 0; i < n; i++) CHECK(dst[i] == src[i], "par_items(%d): item %d", n, i);
    }
}

thread_local std::string tlsError;                     // the library's orb_last_error() is thread-local in the same way

void test_fan_out()
{
    for (int W : {1, 2, 5, 8}) {
        const int total = 41;
        std::vector<int> got(total, -1);
        std::vector<int> rcs;
        std::vector<std::string> errs;
        int bad = orb_fan_out(W, [&](int r) -> int {
            int first = 0, count = 0;
            orb_shard_range_impl(total, W, r, &first, &count);
            for (int i = first; i < first + count; i++) got[i] = r;       // disjoint blocks, as orb_multi_extract_batch writes them
            return 0;
        }, [] { return tlsError; }, rcs, errs);
        CHECK(bad == -1, "fan_out(%d) reports rank %d", W, bad);
        int prev = 0;
        for (int i = 0; i < total; i++) { CHECK(got[i] >= prev && got[i] < W, "partition of %d over %d at %d", total, W, i); prev = got[i]; }
        // ranks 1 and 3 fail with their own (thread-local) messages: the first failing rank is reported with ITS message
        bad = orb_fan_out(W, [&](int r) -> int {
            if (r == 1 || r == 3) { tlsError = "rank " + std::to_string(r) + " failed"; return -100 - r; }
            return 0;
        }, [] { return tlsError; }, rcs, errs);
        if (W >= 2) {
            CHECK(bad == 1 && rcs[1] == -101 && errs[1] == "rank 1 failed", "fan_out(%d): first failing rank %d '%s'", W, bad, bad >= 0 ? errs[bad].c_str() : "");
            if (W >= 4) CHECK(rcs[3] == -103 && errs[3] == "rank 3 failed", "fan_out(%d): rank 3's own message", W);
        } else {
            CHECK(bad == -1, "fan_out(1) has no rank 1");
        }
    }
    // the partition rule itself
    for (int total : {0, 1, 7, 512, 513}) {
        for (int W : {1, 2, 3, 8}) {
            int sum = 0, next = 0;
            for (int r = 0; r < W; r++) {
                int first = -1, count = -1;
                orb_shard_range_impl(total, W, r, &first, &count);
                CHECK(first == next && count >= total / W && count <= total / W + 1, "shard_range(%d, %d, %d)", total, W, r);
                next = first + count;
                sum += count;
            }
            CHECK(sum == total, "shard_range covers %d over %d", total, W);
        }
    }
}

}  // namespace

int main()
{
    test_par_items();
    test_fan_out();
    for (bool inP : {false, true})
        for (bool outP : {false, true}) {
            test_pipeline(100, 16, inP, outP);         // 7 chunks, the last one short
            test_pipeline(33, 16, inP, outP);          // 3 chunks: the slots are used once each
            test_pipeline(16, 16, inP, outP);          // a single chunk
            test_pipeline(260, 8, inP, outP);          // 33 chunks: every slot reused ten times
        }
    test_pipeline_error(100, 16, 48);                  // the chunk of frames 48..63 fails to launch
    test_pipeline_error(100, 16, 0);
    if (fails) { std::fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
    std::printf("tsan_host: pipeline, copy threads, fan-out and partition clean\n");
    return 0;
}
