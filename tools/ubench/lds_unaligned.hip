// Micro-benchmark: what an LDS read costs on gfx950 when its address is not aligned to its size (the descriptor kernel
// reads 7 vertically adjacent u16 of the transposed row-blurred patch; one 16-byte read at a 2-byte aligned address would do).
// One wave64 per workgroup-quarter, 8 waves per SIMD, every lane a different pseudo-random address with the given alignment.
// Prints ns per wave-instruction per CU for ds_read_b32 / ds_read2_b32 / ds_read_b64 / ds_read_b128 at each alignment.
//   hipcc -O3 --offload-arch=gfx950 -o lds_unaligned lds_unaligned.hip && ./lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned align, unsigned misalign)
{
    __shared__ __attribute__((aligned(16))) unsigned buf[8192];                      // 32 KB per workgroup: 4 workgroups per CU
    for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = i * 2654435761u;
    __syncthreads();
    unsigned a = ((threadIdx.x * 1103515245u + 12345u) >> 8) % 30000u;
    a = a / align * align + misalign;                                                // byte address inside the buffer
    unsigned acc = 0;
    for (int i = 0; i < iters; i++) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned addr = a + 64 * j;
            v[j] = u32x4{0, 0, 0, 0};
            if (KIND == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(v[j].x) : "v"(addr));
            else if (KIND == 1) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(*reinterpret_cast<u32x2*>(&v[j])) : "v"(addr));
            else if (KIND == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(*reinterpret_cast<u32x2*>(&v[j])) : "v"(addr));
            else asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(addr));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; j++) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        a = (a + 4096u) % 30000u / align * align + misalign;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND>
static void run(const char* name, unsigned align, unsigned mis, unsigned* d)
{
    const int iters = 2000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10, align, mis);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, align, mis);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waveInstr = (double)blocks * 4 * iters * 8;                       // per chip
    printf("%-14s address = %2u k + %2u   %8.3f ms   %6.2f ns per wave-instruction per CU\n", name, align, mis, ms, ms * 1e6 / (waveInstr / 256.0));
}

int main()
{
    unsigned* d;
    hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0>("ds_read_b32", 4, 0, d);
    run<0>("ds_read_b32", 4, 2, d);
    run<0>("ds_read_b32", 4, 1, d);
    run<1>("ds_read2_b32", 4, 0, d);
    run<2>("ds_read_b64", 8, 0, d);
    run<2>("ds_read_b64", 8, 4, d);
    run<2>("ds_read_b64", 8, 2, d);
    run<2>("ds_read_b64", 8, 1, d);
    run<3>("ds_read_b128", 16, 0, d);
    run<3>("ds_read_b128", 16, 8, d);
    run<3>("ds_read_b128", 16, 4, d);
    run<3>("ds_read_b128", 16, 2, d);
    run<3>("ds_read_b128", 16, 1, d);
    hipFree(d);
    return 0;
}
