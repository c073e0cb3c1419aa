// Issue cost of packed-f32 VALU instructions for ONE wave per SIMD (the launch shape of the step kernels), against v_fma_f32.
// hipcc --offload-arch=gfx950 -O3 -o pk_issue pk_issue.hip && ./pk_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, long long *cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a3}, p5 = {a5, a7}, p6 = {a0, a2}, p7 = {a4, a6};
    const float m = 1.0001f, c = 0.5f;
    const f2 pm = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {          // 8 independent v_fma_f32 chains, 64 instructions per iteration
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (MODE == 1) {   // 8 independent v_pk_fma_f32 chains
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));)
        } else if (MODE == 2) {   // v_pk_mul_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));)
        } else if (MODE == 3) {   // v_pk_add_f32
            REP8(asm volatile("v_pk_add_f32 %0, %0, %9\n v_pk_add_f32 %1, %1, %9\n v_pk_add_f32 %2, %2, %9\n v_pk_add_f32 %3, %3, %9\n"
                              "v_pk_add_f32 %4, %4, %9\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %9\n v_pk_add_f32 %7, %7, %9\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));)
        } else if (MODE == 4) {   // one dependent v_fma_f32 chain
            REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));)
        } else if (MODE == 5) {   // one dependent v_pk_fma_f32 chain
            REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pm), "v"(pc));)
        } else if (MODE == 6) {   // v_fmac_f32_dpp, independent (source registers are not written here: no hazard)
            REP8(asm volatile("v_fmac_f32_dpp %0, %8, %9 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
                              "v_fmac_f32_dpp %2, %8, %9 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
                              "v_fmac_f32_dpp %4, %8, %9 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
                              "v_fmac_f32_dpp %6, %8, %9 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (MODE == 7) {   // half packed, half plain, interleaved (does a pk next to a plain op cost the sum?)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_fma_f32 %4, %4, %10, %11\n v_pk_fma_f32 %1, %1, %8, %9\n v_fma_f32 %5, %5, %10, %11\n"
                              "v_pk_fma_f32 %2, %2, %8, %9\n v_fma_f32 %6, %6, %10, %11\n v_pk_fma_f32 %3, %3, %8, %9\n v_fma_f32 %7, %7, %10, %11\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(pm), "v"(pc), "v"(m), "v"(c));)
        } else if (MODE == 8) {   // v_pk_fma_f32 with op_sel (cross halves) and neg modifiers
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n"
                              "v_pk_fma_f32 %2, %2, %8, %9 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n"
                              "v_pk_fma_f32 %4, %4, %8, %9 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n"
                              "v_pk_fma_f32 %6, %6, %8, %9 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));)
        } else if (MODE == 9) {   // v_mov_b32_dpp
            REP8(asm volatile("v_mov_b32_dpp %0, %8 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %8 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %8 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %6, %8 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (MODE == 10) {  // s_nop 1
            REP64(asm volatile("s_nop 1");)
        } else if (MODE == 11) {  // v_pk_mul_f32 by a broadcast scalar pair from SGPR-like constant (literal via op_sel_hi 0: both halves use lo)
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %8 op_sel_hi:[1,0]\n"
                              "v_pk_mul_f32 %4, %4, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %5, %5, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %6, %6, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %7, %8 op_sel_hi:[1,0]\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}

template <int MODE>
void run(const char *name, int blocks, float *out, long long *cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long v : h) mean += (double)v;
    mean /= blocks;
    // s_memtime / readcyclecounter ticks at 100 MHz on gfx9: report the wall time per instruction, which does not depend on that
    printf("%-44s blocks %5d: %7.3f ns per instruction  (%.2f cycles at 2.4 GHz; counter %.1f ticks/iter)\n", name, blocks, ms * 1e6 / ((double)iters * 64), ms * 1e6 / ((double)iters * 64) * 2.4, mean / iters);
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 4096 * 64 * sizeof(float)); hipMalloc(&cyc, 4096 * sizeof(long long));
    for (int blocks : {1024, 2048}) {   // one wave per SIMD, two waves per SIMD
        run<0>("v_fma_f32 x8 independent", blocks, out, cyc);
        run<1>("v_pk_fma_f32 x8 independent", blocks, out, cyc);
        run<2>("v_pk_mul_f32 x8 independent", blocks, out, cyc);
        run<3>("v_pk_add_f32 x8 independent", blocks, out, cyc);
        run<4>("v_fma_f32 dependent chain", blocks, out, cyc);
        run<5>("v_pk_fma_f32 dependent chain", blocks, out, cyc);
        run<6>("v_fmac_f32_dpp x8 independent", blocks, out, cyc);
        run<7>("v_pk_fma_f32 / v_fma_f32 interleaved", blocks, out, cyc);
        run<8>("v_pk_fma_f32 op_sel + neg_lo", blocks, out, cyc);
        run<9>("v_mov_b32_dpp x8 independent", blocks, out, cyc);
        run<10>("s_nop 1", blocks, out, cyc);
        run<11>("v_pk_mul_f32 op_sel_hi broadcast", blocks, out, cyc);
    }
    return 0;
}
