// Core clock under load (development aid): a pure-VALU loop timed with s_memtime ticks, s_getreg-free,
// and with HIP events, for 1 workgroup and for a full chip.  ticks/instr tells the true issue cost,
// ticks/second tells the clock the chip actually ran at.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>  // 0: v_mad_u64_u32 chains, 1: v_add_u32 chains, 2: v_mul_hi_u32
__global__ void k(unsigned long long* ticks, unsigned long long* real, uint32_t* sink, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9e3779b9u;
    uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3;
    uint32_t d0 = a, d1 = b, d2 = a + 1, d3 = b + 1, d4 = a + 2, d5 = b + 2, d6 = a + 3, d7 = b + 3;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {
                uint64_t carry;
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c0), "=s"(carry) : "v"(a), "v"(b));
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c1), "=s"(carry) : "v"(a), "v"(b));
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c2), "=s"(carry) : "v"(a), "v"(b));
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c3), "=s"(carry) : "v"(a), "v"(b));
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c4), "=s"(carry) : "v"(a), "v"(b));
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c5), "=s"(carry) : "v"(a), "v"(b));
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c6), "=s"(carry) : "v"(a), "v"(b));
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c7), "=s"(carry) : "v"(a), "v"(b));
            } else if (KIND == 1) {
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(d0) : "v"(a));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(d1) : "v"(a));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(d2) : "v"(a));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(d3) : "v"(a));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(d4) : "v"(a));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(d5) : "v"(a));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(d6) : "v"(a));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(d7) : "v"(a));
            } else {
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(d0) : "v"(a));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(d1) : "v"(a));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(d2) : "v"(a));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(d3) : "v"(a));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(d4) : "v"(a));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(d5) : "v"(a));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(d6) : "v"(a));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(d7) : "v"(a));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { ticks[blockIdx.x] = t1 - t0; real[blockIdx.x] = r1 - r0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7) + d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
}

template <int KIND>
void run(const char* name, int blocks, int threads, int iters) {
    unsigned long long *dt, *dr; uint32_t* sink;
    hipMalloc(&dt, blocks * 8); hipMalloc(&dr, blocks * 8); hipMalloc(&sink, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, dt, dr, sink, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, dt, dr, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t, r; hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost); hipMemcpy(&r, dr, 8, hipMemcpyDeviceToHost);
    const double instr_per_wave = 64.0 * iters;
    const int waves_per_simd = threads / 64 / 4 > 0 ? threads / 64 / 4 : 1;
    printf("%-14s grid %5d x %4d: kernel %8.3f ms, block0 %8.3f ms (100 MHz counter), %10llu core ticks -> core clock %.0f MHz, %.2f core cycles per wave-instr per SIMD\n",
           name, blocks, threads, ms, r / 1e5, t, t / (r / 100.0), (double)t / (instr_per_wave * waves_per_simd));
    hipFree(dt); hipFree(dr); hipFree(sink);
}

int main() {
    for (int rep = 0; rep < 1; ++rep) {
        run<0>("v_mad_u64_u32", 1, 256, 200000);
        run<0>("v_mad_u64_u32", 256, 256, 200000);
        run<0>("v_mad_u64_u32", 256, 1024, 100000);
        run<0>("v_mad_u64_u32", 512, 1024, 100000);
        run<0>("v_mad_u64_u32", 512, 512, 100000);
        run<0>("v_mad_u64_u32", 1024, 256, 100000);
        run<1>("v_add_u32", 1, 256, 200000);
        run<1>("v_add_u32", 256, 1024, 100000);
        run<2>("v_mul_hi_u32", 1, 256, 200000);
        run<2>("v_mul_hi_u32", 256, 1024, 100000);
    }
    return 0;
}
