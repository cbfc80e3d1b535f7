// Does v_pk_maximum3_f16 / v_pk_minimum3_f16 order small integers stored as f16 SUBNORMAL bit patterns
// (bits == value, 0..255) under the default HIP float mode?  Exhaustive over triples from a sample.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* bad)
{
    const int a = threadIdx.x, b = blockIdx.x;
    int errs = 0;
    for (int c = 0; c < 256; c++) {
        unsigned pa = (unsigned)a | ((unsigned)(255 - a) << 16), pb = (unsigned)b | ((unsigned)c << 16), pc = (unsigned)c | ((unsigned)b << 16);
        unsigned mx, mn, mx2, mn2;
        asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(mx) : "v"(pa), "v"(pb), "v"(pc));
        asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(mn) : "v"(pa), "v"(pb), "v"(pc));
        asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(mx2) : "v"(pa), "v"(pb));
        asm volatile("v_pk_min_f16 %0, %1, %2" : "=v"(mn2) : "v"(pa), "v"(pb));
        const int lo[3] = {a, b, c}, hi[3] = {255 - a, c, b};
        int wmxl = max(max(lo[0], lo[1]), lo[2]), wmnl = min(min(lo[0], lo[1]), lo[2]);
        int wmxh = max(max(hi[0], hi[1]), hi[2]), wmnh = min(min(hi[0], hi[1]), hi[2]);
        if ((int)(mx & 0xffff) != wmxl || (int)(mx >> 16) != wmxh) errs++;
        if ((int)(mn & 0xffff) != wmnl || (int)(mn >> 16) != wmnh) errs++;
        if ((int)(mx2 & 0xffff) != max(a, b) || (int)(mx2 >> 16) != max(255 - a, c)) errs += 1000;
        if ((int)(mn2 & 0xffff) != min(a, b) || (int)(mn2 >> 16) != min(255 - a, c)) errs += 1000;
    }
    if (errs) atomicAdd(bad, errs);
}
int main()
{
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d);
    int h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("pk3 f16 on subnormal-coded u8: mismatches = %d (pk3 counts 1, pk2 counts 1000)\n", h);
    return 0;
}
