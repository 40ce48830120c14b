// valu_rate.hip -- issue rate of the integer / f64 VALU instructions the matvec kernels are made of, on one MI355X.
// For each instruction: a loop of 16 independent copies per iteration, W waves per SIMD (W = 1, 2, 4, 8), every CU busy;
// reports shader cycles per wave-instruction per SIMD (s_memtime around the loop, one wave per SIMD sampled) and the
// chip-wide rate from the wall clock.  Answers "is a wave64 VALU instruction 2 or 4 cycles of a SIMD here?".
//   hipcc --offload-arch=gfx950 -O3 scripts/tools/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(S) S S S S S S S S S S S S S S S S
enum { ADD32, XOR32, SHL32, SHL64, ADD64, MUL32, MULHI32, BFE32, CMPSEL, FFBH, BCNT, ADDF64, FMAF64, MOV32, AND64, NOPS };

template <int OP>
__global__ __launch_bounds__(256) void k(int iters, unsigned long long *cycles, uint32_t *sink) {
    uint32_t a = threadIdx.x, b = blockIdx.x | 1, c = 3;
    uint64_t q = threadIdx.x * 0x9e3779b97f4a7c15ull, r = blockIdx.x + 5;
    double d = threadIdx.x, e = 1.0000001;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (OP == ADD32) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == XOR32) { REP16(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == SHL32) { REP16(asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a));) }
        if (OP == SHL64) { REP16(asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q));) }
        if (OP == ADD64) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q) : "v"(r));) }
        if (OP == MUL32) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == MULHI32) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == BFE32) { REP16(asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(a));) }
        if (OP == CMPSEL) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");) }
        if (OP == FFBH) { REP16(asm volatile("v_ffbh_u32 %0, %0" : "+v"(a));) }
        if (OP == BCNT) { REP16(asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == ADDF64) { REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(e));) }
        if (OP == FMAF64) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d) : "v"(e));) }
        if (OP == MOV32) { REP16(asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == AND64) { REP16(asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == NOPS) { REP16(asm volatile("s_nop 0");) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
    if (a + (uint32_t)q + (uint32_t)d == 0x12345678u) sink[0] = a;
}

template <int OP>
static void run(char const *name, int per_op, unsigned long long *d_cycles, uint32_t *d_sink, int cus) {
    int const iters = 4096;
    printf("%-28s", name);
    for (int w = 1; w <= 8; w *= 2) {
        int const blocks = cus * w; // 256-thread blocks = 4 waves = one per SIMD; w blocks per CU -> w waves per SIMD
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, 16, d_cycles, d_sink);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, iters, d_cycles, d_sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cyc; hipMemcpy(&cyc, d_cycles, 8, hipMemcpyDeviceToHost);
        double const insts = (double)iters * 16 * per_op;            // per wave
        double const per_simd = insts * w;                           // wave-instructions one SIMD executed
        printf("  W=%d: %6.2f cyc/inst/SIMD (wave clock), %7.1f G wave-inst/s chip", w, (double)cyc / per_simd, per_simd * cus * 4 / (ms * 1e6));
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int const cus = p.multiProcessorCount;
    printf("%s: %d CUs, clockRate %d kHz; peak at 2 cyc/inst = %.0f G wave-inst/s, at 4 cyc/inst = %.0f\n", p.name, cus, p.clockRate,
           cus * 4 * (p.clockRate * 1e-6) / 2, cus * 4 * (p.clockRate * 1e-6) / 4);
    unsigned long long *d_cycles; uint32_t *d_sink;
    hipMalloc(&d_cycles, 8); hipMalloc(&d_sink, 4);
    run<ADD32>("v_add_u32", 1, d_cycles, d_sink, cus);
    run<XOR32>("v_xor_b32", 1, d_cycles, d_sink, cus);
    run<AND64>("v_and_b32", 1, d_cycles, d_sink, cus);
    run<MOV32>("v_mov_b32", 1, d_cycles, d_sink, cus);
    run<SHL32>("v_lshlrev_b32", 1, d_cycles, d_sink, cus);
    run<SHL64>("v_lshlrev_b64", 1, d_cycles, d_sink, cus);
    run<ADD64>("v_lshl_add_u64", 1, d_cycles, d_sink, cus);
    run<MUL32>("v_mul_lo_u32", 1, d_cycles, d_sink, cus);
    run<MULHI32>("v_mul_hi_u32", 1, d_cycles, d_sink, cus);
    run<BFE32>("v_bfe_u32", 1, d_cycles, d_sink, cus);
    run<CMPSEL>("v_cmp_lt_u32+v_cndmask", 2, d_cycles, d_sink, cus);
    run<FFBH>("v_ffbh_u32", 1, d_cycles, d_sink, cus);
    run<BCNT>("v_bcnt_u32_b32", 1, d_cycles, d_sink, cus);
    run<ADDF64>("v_add_f64", 1, d_cycles, d_sink, cus);
    run<FMAF64>("v_fma_f64", 1, d_cycles, d_sink, cus);
    run<NOPS>("s_nop 0", 1, d_cycles, d_sink, cus);
    return 0;
}
