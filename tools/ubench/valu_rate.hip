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
    unsigned long long mask = 0x5555555555555555ull ^ seed, m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, m6 = 0, m7 = 0;
    int sl0 = 0, sl1 = 0, sl2 = 0, sl3 = 0, sl4 = 0, sl5 = 0, sl6 = 0, sl7 = 0;
    unsigned long long q0 = r0, q1 = r1, q2 = r2, q3 = r3, q4 = r4, q5 = r5, q6 = r6, q7 = r7;
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
#define ML(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define MH(n) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define M24(n) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define MA24(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define D2(n) asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define D4(n) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define AB(n) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define BF(n) asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define LA(n) asm volatile("v_lshl_add_u32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define A3(n) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define CM(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##n) : "v"(a));
#define M64(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##n) : "v"(a), "v"(b) : "vcc");
#define L64(n) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q##n) : "v"(q7));
#define CI(n) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r##n));
#define MF(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define AD64(n) asm volatile("v_add_f64 %0, %0, %1" : "+v"(q##n) : "v"(q7));
#define CE(n) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "s"(mask));
#define CD(n) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r##n) : "v"(a), "v"(b));
#define CP(n) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(r##n) : "v"(a), "v"(b) : "vcc");
#define CV(n) asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(r##n), "v"(a) : "vcc");
#define CS(n) asm volatile("v_cmp_gt_i32 %0, %1, %2" : "=s"(m##n) : "v"(r##n), "v"(a));
#define AN(n) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define SH(n) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(r##n));
#define AO(n) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r##n) : "v"(a), "v"(b));
#define SB(n) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define MU(n) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define XR(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r##n) : "v"(a));
#define MV(n) asm volatile("v_mov_b32 %0, %1" : "=v"(r##n) : "v"(a));
#define RL(n) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sl##n) : "v"(r##n));
#define BC(n) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r##n) : "v"(a));
        if (KIND == 0) { REP8(I3) REP8(I3) }
        if (KIND == 32) { REP8(CE) REP8(CE) }
        if (KIND == 33) { REP8(CD) REP8(CD) }
        if (KIND == 34) { REP8(CP) REP8(CP) }
        if (KIND == 35) { REP8(CV) REP8(CV) }
        if (KIND == 36) { REP8(CS) REP8(CS) }
        if (KIND == 37) { REP8(AN) REP8(AN) }
        if (KIND == 38) { REP8(SH) REP8(SH) }
        if (KIND == 39) { REP8(AO) REP8(AO) }
        if (KIND == 40) { REP8(SB) REP8(SB) }
        if (KIND == 41) { REP8(MU) REP8(MU) }
        if (KIND == 42) { REP8(XR) REP8(XR) }
        if (KIND == 43) { REP8(MV) REP8(MV) }
        if (KIND == 44) { REP8(RL) REP8(RL) }
        if (KIND == 45) { REP8(BC) REP8(BC) }
        if (KIND == 16) { REP8(ML) REP8(ML) }
        if (KIND == 17) { REP8(MH) REP8(MH) }
        if (KIND == 18) { REP8(M24) REP8(M24) }
        if (KIND == 19) { REP8(MA24) REP8(MA24) }
        if (KIND == 20) { REP8(D2) REP8(D2) }
        if (KIND == 21) { REP8(D4) REP8(D4) }
        if (KIND == 22) { REP8(AB) REP8(AB) }
        if (KIND == 23) { REP8(BF) REP8(BF) }
        if (KIND == 24) { REP8(LA) REP8(LA) }
        if (KIND == 25) { REP8(A3) REP8(A3) }
        if (KIND == 26) { REP8(CM) REP8(CM) }
        if (KIND == 27) { REP8(M64) REP8(M64) }
        if (KIND == 28) { REP8(L64) REP8(L64) }
        if (KIND == 29) { REP8(CI) REP8(CI) }
        if (KIND == 30) { REP8(MF) REP8(MF) }
        if (KIND == 31) { REP8(AD64) REP8(AD64) }
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
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (int)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^
        (int)(m0 ^ m1 ^ m2 ^ m3 ^ m4 ^ m5 ^ m6 ^ m7) ^ sl0 ^ sl1 ^ sl2 ^ sl3 ^ sl4 ^ sl5 ^ sl6 ^ sl7;
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
    run<16>("v_mul_lo_u32", d); run<17>("v_mul_hi_u32", d); run<18>("v_mul_u32_u24", d); run<19>("v_mad_u32_u24", d);
    run<20>("v_dot2_u32_u16", d); run<21>("v_dot4_u32_u8", d); run<22>("v_alignbyte_b32", d); run<23>("v_bfe_u32", d);
    run<24>("v_lshl_add_u32", d); run<25>("v_add3_u32", d); run<26>("v_cndmask_b32", d); run<27>("v_mad_u64_u32", d);
    run<28>("v_lshl_add_u64", d); run<29>("v_cvt_f32_u32", d); run<30>("v_mul_f32", d); run<31>("v_add_f64", d);
    run<32>("v_cndmask_e64 sgpr", d); run<33>("v_cndmask vcc 3-addr", d); run<34>("v_cmp+v_cndmask pair", d);
    run<35>("v_cmp -> vcc", d); run<36>("v_cmp -> sgpr", d); run<37>("v_and_b32", d); run<38>("v_lshrrev_b32", d);
    run<39>("v_and_or_b32", d); run<40>("v_sub_u32", d); run<41>("v_min_u32", d); run<42>("v_xor_b32", d);
    run<43>("v_mov_b32", d); run<44>("v_readlane_b32", d); run<45>("v_bcnt_u32_b32", d);
    return 0;
}
