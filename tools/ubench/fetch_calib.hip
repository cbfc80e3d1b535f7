// fetch_calib.hip -- known-byte-count reads for calibrating rocprofv3's FETCH_SIZE on gfx950 per load width
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ...
// other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Each kernel reads every byte of a 1 GiB buffer (4x the 256 MiB Infinity Cache) exactly once:
//   k_read4 / k_read8 / k_read16 : coalesced streams of 4 / 8 / 16 bytes per lane
//   k_rows48                     : the descriptor kernel's pattern, 48-byte row segments read as dwords
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib
// run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -o run -- tools/ubench/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdio>

template <typename T>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ src, size_t n, unsigned* __restrict__ sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T v = src[i];
        const unsigned* w = reinterpret_cast<const unsigned*>(&v);
#pragma unroll
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x12345678u) sink[0] = acc;                       // keeps the loads alive, practically never taken
}

// rows of 48 bytes (12 dwords) at a pitch of 64 bytes cover the buffer; lane -> (row, dword) with 5 rows per 60 lanes
__global__ __launch_bounds__(64) void k_rows48(const unsigned* __restrict__ src, size_t nRows, unsigned* __restrict__ sink)
{
    unsigned acc = 0;
    const int lane = threadIdx.x, r = lane / 12, c = lane - 12 * r;
    if (r < 5)
        for (size_t row = (size_t)blockIdx.x * 5 + r; row < nRows; row += (size_t)gridDim.x * 5) acc ^= src[row * 16 + c];
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    void* buf;
    unsigned* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_read<unsigned>, dim3(256 * 32), dim3(256), 0, 0, (const unsigned*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(k_read<uint2>, dim3(256 * 32), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, sink);
        hipLaunchKernelGGL(k_read<uint4>, dim3(256 * 32), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(k_rows48, dim3(256 * 64), dim3(64), 0, 0, (const unsigned*)buf, bytes / 64, sink);
    }
    (void)hipDeviceSynchronize();
    printf("true_bytes k_read<unsigned>=%zu k_read<uint2>=%zu k_read<uint4>=%zu k_rows48=%zu (48 of every 64 bytes)\n", bytes, bytes,
           bytes, bytes / 64 * 48);
    return 0;
}
