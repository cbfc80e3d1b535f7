// threads_stress.cpp -- the host threads of liborbhip all at once, from a C++ caller of the C ABI (run ON THE GPU BOX:
// `make -C orb-slam2-chinesenotes_amd threads-stress && tools/threads_stress`; tests/test_gpu_threads.py does).
// The scenario of the reference -- two extractors driven from two threads, src/Frame.cc:82-85 -- while a third thread runs
// host batches (the staging threads of csrc/orb_host_pipe.hip), a fourth the multi-device entry (one thread per listed
// device, csrc/orb_multi.hip) and a fifth creates and destroys handles.  Every call must succeed and repeat its results.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/orb_hip.h"

static std::vector<uint8_t> frame(int w, int h, unsigned seed)
{
    std::vector<uint8_t> im((size_t)w * h, 128);
    unsigned s = seed * 2654435761u + 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int k = 0; k < 300; k++) {                                // rectangles of random grey: corners for FAST
        const int x0 = rnd() % w, y0 = rnd() % h, rw = 4 + rnd() % 90, rh = 4 + rnd() % 90, v = rnd() % 256;
        for (int y = y0; y < y0 + rh && y < h; y++) std::memset(&im[(size_t)y * w + x0], v, (size_t)std::min(rw, w - x0));
    }
    return im;
}

int main()
{
    orb_extractor_params p;
    std::memset(&p, 0, sizeof(p));
    p.nfeatures = 1000; p.scale_factor = 1.2f; p.nlevels = 8; p.ini_th_fast = 20; p.min_th_fast = 7;
    const int W = 640, H = 480;
    std::atomic<int> failures{0};
    std::atomic<bool> stop{false};
    auto single = [&](int id) {                                    // one ORBextractor object per thread, call after call
        orb_extractor* h = nullptr;
        if (orb_extractor_create(&p, 0, &h) != ORB_OK) { std::fprintf(stderr, "create: %s\n", orb_last_error()); failures++; return; }
        const int cap = orb_extractor_max_keypoints(h);
        std::vector<orb_keypoint> kps(cap);
        std::vector<uint8_t> desc((size_t)cap * 32), first;
        const std::vector<uint8_t> im = frame(W, H, 100 + id);
        for (int it = 0; it < 25; it++) {
            int n = 0;
            if (orb_extract(h, im.data(), H, W, W, kps.data(), desc.data(), cap, &n) != ORB_OK) { std::fprintf(stderr, "extract: %s\n", orb_last_error()); failures++; break; }
            std::vector<uint8_t> now(desc.begin(), desc.begin() + (size_t)n * 32);
            if (it == 0) first = now;
            else if (now != first) { std::fprintf(stderr, "thread %d: call %d differs from call 0\n", id, it); failures++; break; }
        }
        orb_extractor_destroy(h);
    };
    auto batches = [&]() {                                         // host batches: the staging threads of orb_host_pipe
        orb_extractor* h = nullptr;
        if (orb_extractor_create(&p, 0, &h) != ORB_OK) { failures++; return; }
        const int cap = orb_extractor_max_keypoints(h), F = 48;
        std::vector<uint8_t> imgs;
        for (int f = 0; f < F; f++) { const std::vector<uint8_t> im = frame(W, H, 500 + f); imgs.insert(imgs.end(), im.begin(), im.end()); }
        std::vector<orb_keypoint> kps((size_t)cap * F);
        std::vector<uint8_t> desc((size_t)cap * 32 * F);
        std::vector<int32_t> counts(F), first;
        for (int it = 0; it < 4; it++) {
            if (orb_extract_batch(h, imgs.data(), F, H, W, W, (size_t)W * H, kps.data(), desc.data(), cap, counts.data()) != ORB_OK) { std::fprintf(stderr, "batch: %s\n", orb_last_error()); failures++; break; }
            if (it == 0) first = counts;
            else if (counts != first) { std::fprintf(stderr, "batch %d: counts differ\n", it); failures++; break; }
        }
        orb_extractor_destroy(h);
    };
    auto multi = [&]() {                                           // one handle + host thread per listed device (the same GPU twice)
        const int devs[2] = {0, 0};
        orb_multi* m = nullptr;
        if (orb_multi_create(&p, devs, 2, &m) != ORB_OK) { std::fprintf(stderr, "multi: %s\n", orb_last_error()); failures++; return; }
        const int cap = orb_extractor_max_keypoints(orb_multi_handle(m, 0)), F = 32;
        std::vector<uint8_t> imgs;
        for (int f = 0; f < F; f++) { const std::vector<uint8_t> im = frame(W, H, 900 + f); imgs.insert(imgs.end(), im.begin(), im.end()); }
        std::vector<orb_keypoint> kps((size_t)cap * F);
        std::vector<uint8_t> desc((size_t)cap * 32 * F);
        std::vector<int32_t> counts(F);
        for (int it = 0; it < 3; it++)
            if (orb_multi_extract_batch(m, imgs.data(), F, H, W, W, (size_t)W * H, kps.data(), desc.data(), cap, counts.data()) != ORB_OK) { std::fprintf(stderr, "multi batch: %s\n", orb_last_error()); failures++; break; }
        orb_multi_destroy(m);
    };
    auto churn = [&]() {                                           // handles come and go meanwhile
        const std::vector<uint8_t> im = frame(320, 240, 7);
        while (!stop.load()) {
            orb_extractor* h = nullptr;
            if (orb_extractor_create(&p, 0, &h) != ORB_OK) { failures++; return; }
            const int cap = orb_extractor_max_keypoints(h);
            std::vector<orb_keypoint> kps(cap);
            std::vector<uint8_t> desc((size_t)cap * 32);
            int n = 0;
            if (orb_extract(h, im.data(), 240, 320, 320, kps.data(), desc.data(), cap, &n) != ORB_OK) failures++;
            orb_extractor_destroy(h);
        }
    };
    std::thread tc(churn), t0(single, 0), t1(single, 1), tb(batches), tm(multi);
    t0.join(); t1.join(); tb.join(); tm.join();
    stop.store(true);
    tc.join();
    std::printf("threads_stress: %d failures\n", failures.load());
    return failures.load() ? 1 : 0;
}
