// stamp_cost.hip -- what an in-kernel time stamp costs on gfx950: s_memrealtime (100 MHz real-time counter, what
// wall_clock64() reads) against s_memtime (shader clock, what clock64() reads), each read 16 times back to back with the
// value consumed (s_waitcnt lgkmcnt(0)) before the next read, by one wave on an otherwise idle GPU and by one wave per CU.
//   hipcc --offload-arch=gfx950 -O3 -o stamp_cost stamp_cost.hip && ./stamp_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_stamps(unsigned long long* out, int useReal)
{
    unsigned long long t[17];
    for (int i = 0; i < 17; i++) {
        t[i] = useReal ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (threadIdx.x == 0)
        for (int i = 0; i < 17; i++) out[blockIdx.x * 17 + i] = t[i];
}

int main()
{
    unsigned long long* d;
    hipMalloc(&d, 256 * 17 * 8);
    std::vector<unsigned long long> h(256 * 17);
    for (int real = 0; real < 2; real++)
        for (int grid : {1, 256}) {
            for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_stamps, dim3(grid), dim3(64), 0, 0, d, real);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, grid * 17 * 8, hipMemcpyDeviceToHost);
            double sum = 0, mx = 0;
            for (int b = 0; b < grid; b++)
                for (int i = 1; i < 17; i++) { const double dlt = (double)(h[b * 17 + i] - h[b * 17 + i - 1]); sum += dlt; mx = dlt > mx ? dlt : mx; }
            const double avg = sum / (grid * 16.0);
            printf("%s, %3d waves: %.1f ticks per dependent read (max %.0f)%s\n", real ? "s_memrealtime" : "s_memtime    ", grid, avg, mx,
                   real ? "  [tick = 10 ns]" : "  [tick = 1 shader-clock unit]");
        }
    return 0;
}
