// Micro-benchmark: issue rate of candidate min/max instructions on gfx950 (one dependent-free stream
// of 8 independent chains per wave, 8 waves per SIMD).  Prints wave-instructions per ns for the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed)
{
    int r0 = threadIdx.x + seed, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19;
    int a = r0 ^ 0x55, b = r1 ^ 0x33;
    for (int i = 0; i < iters; i++) {
#define I3(n) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define F3(n) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define I2(n) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define F2(n) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define PI(n) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define PU(n) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define PF(n) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define H3(n) asm volatile("v_max3_f16 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define P3(n) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define U3(n) asm volatile("v_max3_u16 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define AD(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define FM(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define SD(n) asm volatile("v_max_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(r##n) : "v"(a));
#define PA(n) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define PS(n) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define PM(n) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
        if (KIND == 0) { REP8(I3) REP8(I3) }
        if (KIND == 1) { REP8(F3) REP8(F3) }
        if (KIND == 2) { REP8(I2) REP8(I2) }
        if (KIND == 3) { REP8(F2) REP8(F2) }
        if (KIND == 4) { REP8(PI) REP8(PI) }
        if (KIND == 5) { REP8(PF) REP8(PF) }
        if (KIND == 6) { REP8(H3) REP8(H3) }
        if (KIND == 7) { REP8(P3) REP8(P3) }
        if (KIND == 8) { REP8(U3) REP8(U3) }
        if (KIND == 9) { REP8(AD) REP8(AD) }
        if (KIND == 10) { REP8(FM) REP8(FM) }
        if (KIND == 11) { REP8(SD) REP8(SD) }
        if (KIND == 12) { REP8(PU) REP8(PU) }
        if (KIND == 13) { REP8(PA) REP8(PA) }
        if (KIND == 14) { REP8(PS) REP8(PS) }
        if (KIND == 15) { REP8(PM) REP8(PM) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

template <int KIND>
void run(const char* name, int* d)
{
    const int blocks = 256 * 8, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 100, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waveInstr = (double)blocks * 4 * iters * 16;
    printf("%-22s %8.3f ms  %8.1f wave-instr/ns  (=> %.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", name, ms,
           waveInstr / (ms * 1e6), 1024.0 * 2.4 / (waveInstr / (ms * 1e6)));
}

int main()
{
    int* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_max3_i32", d); run<1>("v_max3_f32", d); run<2>("v_max_i32", d); run<3>("v_max_f32", d);
    run<4>("v_pk_max_i16", d); run<12>("v_pk_max_u16", d); run<5>("v_pk_max_f16", d); run<6>("v_max3_f16", d);
    run<7>("v_pk_maximum3_f16", d); run<8>("v_max3_u16", d); run<9>("v_add_u32", d); run<10>("v_fma_f32", d);
    run<11>("v_max_u16_sdwa", d); run<13>("v_pk_add_u16", d); run<14>("v_pk_sub_i16", d); run<15>("v_perm_b32", d);
    return 0;
}
