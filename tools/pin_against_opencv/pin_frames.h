// pin_frames.h -- the seeded synthetic frame generator of the benchmark (Python: orbhip/synth.py synth_frame) in
// plain C++, no OpenCV: splitmix64 stream, 400 rectangles then 200 discs on mid-grey, uniform noise in [-6, 6].
// tests/test_pin_kit.py compiles this header alone and compares its frames with the Python generator's, byte for byte.
#pragma once
#include <cstdint>
#include <vector>

namespace pin
{
static const uint64_t SEED0 = 0x0B5EED00ull;

// element `index` of the splitmix64 stream of `seed` (state = seed + (index + 1) * golden gamma)
inline uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

inline std::vector<uint8_t> synth_frame(int index, int width, int height, int n_rect = 400, int n_disc = 200, int noise = 6)
{
    const uint64_t seed = SEED0 + (uint64_t)index;
    const uint64_t nshape = 5ull * n_rect + 4ull * n_disc;
    std::vector<int> img((size_t)width * height, 128);
    uint64_t p = 0;
    for (int k = 0; k < n_rect; k++) {
        const int x0 = (int)(splitmix64_at(seed, p) % (uint64_t)width), y0 = (int)(splitmix64_at(seed, p + 1) % (uint64_t)height);
        const int w = 4 + (int)(splitmix64_at(seed, p + 2) % 117), h = 4 + (int)(splitmix64_at(seed, p + 3) % 117);
        const int g = (int)(splitmix64_at(seed, p + 4) % 256);
        p += 5;
        for (int y = y0; y < y0 + h && y < height; y++)
            for (int x = x0; x < x0 + w && x < width; x++) img[(size_t)y * width + x] = g;
    }
    for (int k = 0; k < n_disc; k++) {
        const int cx = (int)(splitmix64_at(seed, p) % (uint64_t)width), cy = (int)(splitmix64_at(seed, p + 1) % (uint64_t)height);
        const int rad = 2 + (int)(splitmix64_at(seed, p + 2) % 59), g = (int)(splitmix64_at(seed, p + 3) % 256);
        p += 4;
        for (int y = cy - rad < 0 ? 0 : cy - rad; y <= cy + rad && y < height; y++)
            for (int x = cx - rad < 0 ? 0 : cx - rad; x <= cx + rad && x < width; x++)
                if ((x - cx) * (x - cx) + (y - cy) * (y - cy) <= rad * rad) img[(size_t)y * width + x] = g;
    }
    std::vector<uint8_t> out((size_t)width * height);
    for (size_t i = 0; i < out.size(); i++) {
        int v = img[i];
        if (noise > 0) v += (int)(splitmix64_at(seed, nshape + i) % (uint64_t)(2 * noise + 1)) - noise;
        out[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
    return out;
}

// FNV-1a over a row: what the vectors file holds per image row (a differing row is named, not just "the image differs")
inline uint32_t row_hash(const uint8_t* p, int n)
{
    uint32_t h = 2166136261u;
    for (int i = 0; i < n; i++) { h ^= p[i]; h *= 16777619u; }
    return h;
}
}  // namespace pin
