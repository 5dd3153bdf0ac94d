// Integer-instruction issue cost on gfx950 (development aid).  Inline asm so nothing is folded.
// 8 waves per SIMD, 8 independent chains per lane; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 1024
#define OPS(STR)                                                                            \
    for (int it = 0; it < ITER; ++it) {                                                     \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(STR : "+v"(x[i]) : "v"(b), "v"(c)); \
    }
#define OPS64(STR)                                                                          \
    for (int it = 0; it < ITER; ++it) {                                                     \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(STR : "+v"(y[i]) : "v"(b), "v"(c)); \
    }

template <int OP>
__global__ void k(uint32_t* out, uint32_t a0, uint32_t b0) {
    uint32_t x[8];
    uint64_t y[8];
    for (int i = 0; i < 8; ++i) { x[i] = a0 + threadIdx.x * 7 + i; y[i] = x[i]; }
    uint32_t b = b0 | 1, c = b0 + 17;
    if (OP == 0) OPS("v_add_u32 %0, %0, %1")
    if (OP == 1) OPS("v_mul_lo_u32 %0, %0, %1")
    if (OP == 2) OPS("v_mul_hi_u32 %0, %0, %1")
    if (OP == 3) OPS64("v_mad_u64_u32 %0, vcc, %1, %2, %0")
    if (OP == 4) OPS("v_mul_u32_u24 %0, %0, %1")
    if (OP == 5) OPS("v_mad_u32_u24 %0, %0, %1, %2")
    if (OP == 6) OPS("v_min_u32 %0, %0, %1")
    if (OP == 7) OPS("v_lshl_add_u32 %0, %0, 1, %1")
    if (OP == 8) OPS("v_bfe_i32 %0, %0, 0, 7")
    if (OP == 9) OPS("v_add3_u32 %0, %0, %1, %2")
    if (OP == 10) OPS("v_sub_u32 %0, %0, %1")
    if (OP == 11) OPS("v_mul_hi_i32 %0, %0, %1")
    if (OP == 12) OPS("v_cndmask_b32 %0, %0, %1, vcc")
    if (OP == 13) OPS("v_xad_u32 %0, %0, %1, %2")
    if (OP == 14) OPS("v_mad_i32_i24 %0, %0, %1, %2")
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + (uint32_t)y[i] + (uint32_t)(y[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint32_t* d_out) {
    dim3 grid(256 * 8), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d_out, 3u, 5u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d_out, 3u, 5u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr_per_simd = 5.0 * (grid.x * block.x / 64.0) * ITER * 8 / 1024.0;
    double ns = ms * 1e6 / wave_instr_per_simd;
    printf("%-18s %.3f ns per wave-instr per SIMD = %.2f cycles @2.4GHz, %.2f @2.1GHz\n", name, ns, ns * 2.4, ns * 2.1);
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", d); run<10>("v_sub_u32", d); run<6>("v_min_u32", d); run<9>("v_add3_u32", d);
    run<7>("v_lshl_add_u32", d); run<13>("v_xad_u32", d); run<8>("v_bfe_i32", d); run<12>("v_cndmask_b32", d);
    run<1>("v_mul_lo_u32", d); run<2>("v_mul_hi_u32", d); run<11>("v_mul_hi_i32", d);
    run<3>("v_mad_u64_u32", d); run<4>("v_mul_u32_u24", d); run<5>("v_mad_u32_u24", d); run<14>("v_mad_i32_i24", d);
    return 0;
}
