// Issue cost of candidate VALU primitives on gfx950 at 4 waves per SIMD (development aid): cycles per
// wave-instruction per SIMD, from s_memtime over a long unrolled loop with 8 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define BODY(INS)                                                                                   \
    for (int i = 0; i < iters; ++i) {                                                               \
        REP8(asm volatile(INS : "+v"(d0) : "v"(a), "v"(b)); asm volatile(INS : "+v"(d1) : "v"(a), "v"(b)); \
             asm volatile(INS : "+v"(d2) : "v"(a), "v"(b)); asm volatile(INS : "+v"(d3) : "v"(a), "v"(b)); \
             asm volatile(INS : "+v"(d4) : "v"(a), "v"(b)); asm volatile(INS : "+v"(d5) : "v"(a), "v"(b)); \
             asm volatile(INS : "+v"(d6) : "v"(a), "v"(b)); asm volatile(INS : "+v"(d7) : "v"(a), "v"(b));) \
    }

template <int KIND>
__global__ void k32(unsigned long long* ticks, uint32_t* sink, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 1, b = (a ^ 0x9e3779b9u) | 1u;
    uint32_t d0 = a, d1 = b, d2 = a + 1, d3 = b + 1, d4 = a + 2, d5 = b + 2, d6 = a + 3, d7 = b + 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (KIND == 0) { BODY("v_mul_lo_u32 %0, %1, %2") }
    if (KIND == 1) { BODY("v_mul_u32_u24 %0, %1, %2") }
    if (KIND == 2) { BODY("v_mad_u32_u24 %0, %1, %2, %0") }
    if (KIND == 3) { BODY("v_mul_hi_u32_u24 %0, %1, %2") }
    if (KIND == 4) { BODY("v_mad_u32_u16 %0, %1, %2, %0") }
    if (KIND == 5) { BODY("v_dot2_u32_u16 %0, %1, %2, %0") }
    if (KIND == 6) { BODY("v_dot4_u32_u8 %0, %1, %2, %0") }
    if (KIND == 7) { BODY("v_fma_f32 %0, %1, %2, %0") }
    if (KIND == 8) { BODY("v_pk_mul_lo_u16 %0, %1, %2") }
    if (KIND == 9) { BODY("v_add3_u32 %0, %1, %2, %0") }
    if (KIND == 10) { BODY("v_lshl_add_u32 %0, %1, 3, %0") }
    if (KIND == 11) { BODY("v_mul_hi_u32 %0, %1, %2") }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
}
template <int KIND>
__global__ void k64(unsigned long long* ticks, uint32_t* sink, int iters) {
    double a = threadIdx.x * 1.0000001 + 1.0, b = 0.99999;
    double d0 = a, d1 = b, d2 = a + 1, d3 = b + 1, d4 = a + 2, d5 = b + 2, d6 = a + 3, d7 = b + 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (KIND == 0) { BODY("v_fma_f64 %0, %1, %2, %0") }
    if (KIND == 1) { BODY("v_mul_f64 %0, %1, %2") }
    if (KIND == 2) { BODY("v_add_f64 %0, %1, %0") }
    if (KIND == 3) { BODY("v_rndne_f64 %0, %1") }
    if (KIND == 4) { BODY("v_pk_fma_f32 %0, %1, %2, %0") }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

template <typename K>
void run(const char* name, K kern) {
    const int blocks = 256, threads = 1024, iters = 20000;  // 4 waves per SIMD; LAST-finishing wave via kernel time
    unsigned long long* dt; uint32_t* sink;
    hipMalloc(&dt, blocks * 8); hipMalloc(&sink, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dt, sink, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dt, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // whole-kernel time / instructions per SIMD (4 waves x iters x 64) -> ns; cycles at the ~2.1 GHz the chip holds under load
    const double ns = ms * 1e6 / (4.0 * iters * 64.0);
    printf("%-22s %6.2f ns per wave-instruction per SIMD  (~%.1f cycles @2.1 GHz)\n", name, ns, ns * 2.1);
    hipFree(dt); hipFree(sink);
}

int main() {
    run("v_mul_lo_u32", k32<0>); run("v_mul_hi_u32", k32<11>); run("v_mul_u32_u24", k32<1>); run("v_mad_u32_u24", k32<2>);
    run("v_mul_hi_u32_u24", k32<3>); run("v_mad_u32_u16", k32<4>); run("v_dot2_u32_u16", k32<5>); run("v_dot4_u32_u8", k32<6>);
    run("v_pk_mul_lo_u16", k32<8>); run("v_add3_u32", k32<9>); run("v_lshl_add_u32", k32<10>); run("v_fma_f32", k32<7>);
    run("v_fma_f64", k64<0>); run("v_mul_f64", k64<1>); run("v_add_f64", k64<2>); run("v_rndne_f64", k64<3>); run("v_pk_fma_f32", k64<4>);
    return 0;
}
