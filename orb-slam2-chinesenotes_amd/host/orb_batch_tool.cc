// orb_batch_tool.cc -- C++ host program for the batched-frames mode (north_star: "host stays C++ calling HIP through a
// thin C-ABI; a batched-frames mode shards independent frames across the GPUs of one node with RCCL broadcast of the
// BRIEF pattern"): reads a file of raw 8-bit grayscale frames, extracts ORB features of all of them with
// orb_multi_extract_batch over the listed GPUs, writes counts / keypoints / descriptors.  What a dataset-processing
// caller does instead of looping over ORBextractor::operator() (reference src/ORBextractor.cc:1084-1150, called per
// frame from src/Frame.cc:262-268).  Plain C++ over include/orb_hip.h: no OpenCV, no HIP headers.
//
//   orb_batch_tool <frames.raw> <width> <height> <n_frames> <nfeatures> <out_prefix> [device,device,...]
//   build: g++ -std=c++17 -O2 -I include orb-slam2-chinesenotes_amd/host/orb_batch_tool.cc \
//              -L orb-slam2-chinesenotes_amd -lorbhip -Wl,-rpath,$PWD/orb-slam2-chinesenotes_amd -o orb_batch_tool
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "orb_hip.h"

static void writeFile(const std::string& path, const void* p, size_t bytes)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f || std::fwrite(p, 1, bytes, f) != bytes) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); std::exit(2); }
    std::fclose(f);
}

int main(int argc, char** argv)
{
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s frames.raw width height n_frames nfeatures out_prefix [dev,dev,...]\n", argv[0]);
        return 2;
    }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), N = std::atoi(argv[4]), nf = std::atoi(argv[5]);
    const std::string out = argv[6];
    std::vector<int> devs;
    if (argc > 7)
        for (char* t = std::strtok(argv[7], ","); t; t = std::strtok(nullptr, ",")) devs.push_back(std::atoi(t));
    if (devs.empty()) devs.push_back(0);

    const orb_extractor_params prm = {nf, 1.2f, 8, 20, 7};          // the reference's defaults (Tracking.cc:117-126 reads them from the settings file)
    orb_multi* m = nullptr;
    if (orb_multi_create(&prm, devs.data(), (int)devs.size(), &m) != ORB_OK) {
        std::fprintf(stderr, "orb_multi_create: %s\n", orb_last_error());
        return 1;
    }
    const int cap = orb_extractor_max_keypoints(orb_multi_handle(m, 0));
    const size_t frameBytes = (size_t)W * H;
    // pinned host buffers: copied from / to directly, at PCIe rate
    uint8_t* imgs = static_cast<uint8_t*>(orb_host_alloc(frameBytes * N));
    orb_keypoint* kps = static_cast<orb_keypoint*>(orb_host_alloc(sizeof(orb_keypoint) * (size_t)cap * N));
    uint8_t* desc = static_cast<uint8_t*>(orb_host_alloc((size_t)ORB_DESC_BYTES * cap * N));
    std::vector<int32_t> counts(N, 0);
    if (!imgs || !kps || !desc) { std::fprintf(stderr, "pinned allocation failed\n"); return 1; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(imgs, 1, frameBytes * N, f) != frameBytes * N) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    std::fclose(f);

    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {                              // the first pass also builds geometry / allocates
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = orb_multi_extract_batch(m, imgs, N, H, W, (size_t)W, frameBytes, kps, desc, cap, counts.data());
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc != ORB_OK) { std::fprintf(stderr, "orb_multi_extract_batch: %s\n", orb_last_error()); return 1; }
        best = std::min(best, dt);
    }
    long long total = 0;
    for (int i = 0; i < N; i++) total += counts[i];
    std::printf("{\"frames\": %d, \"devices\": %d, \"keypoints\": %lld, \"frames_per_s\": %.1f}\n", N, (int)devs.size(), total, N / best);

    writeFile(out + ".counts", counts.data(), sizeof(int32_t) * (size_t)N);
    // valid prefixes only, frame after frame
    std::vector<uint8_t> k, d;
    for (int i = 0; i < N; i++) {
        const uint8_t* kp = reinterpret_cast<const uint8_t*>(kps + (size_t)cap * i);
        k.insert(k.end(), kp, kp + sizeof(orb_keypoint) * (size_t)counts[i]);
        d.insert(d.end(), desc + (size_t)ORB_DESC_BYTES * cap * i, desc + (size_t)ORB_DESC_BYTES * ((size_t)cap * i + counts[i]));
    }
    writeFile(out + ".kps", k.data(), k.size());
    writeFile(out + ".desc", d.data(), d.size());
    orb_host_free(imgs); orb_host_free(kps); orb_host_free(desc);
    orb_multi_destroy(m);
    return 0;
}
