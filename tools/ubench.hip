// Integer-instruction throughput on gfx950 (development aid): wave64 issue cost of the
// operations the NTT / MAC loops are made of.  Prints Gop/s chip-wide and cycles per
// wave-instruction per SIMD assuming the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITER 2048
template <int OP>
__global__ void k(uint32_t* out, uint32_t a0, uint32_t b0) {
    uint32_t x[8];
    uint64_t y[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { x[i] = a0 + threadIdx.x * 7 + i; y[i] = x[i]; d[i] = (double)x[i]; }
    uint32_t b = b0 | 1;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) x[i] = x[i] + b;                                   // v_add_u32
            if (OP == 1) x[i] = x[i] * b;                                   // v_mul_lo_u32
            if (OP == 2) x[i] = __umulhi(x[i], b);                          // v_mul_hi_u32
            if (OP == 3) y[i] = (uint64_t)(uint32_t)y[i] * b + y[i];        // v_mad_u64_u32
            if (OP == 4) x[i] = __umul24(x[i], b);          // v_mul_u32_u24
            if (OP == 5) d[i] = fma(d[i], 1.0000001, 0.5);                  // v_fma_f64
            if (OP == 6) x[i] = min(x[i], x[i] - b);                        // sub + min
            if (OP == 7) { uint32_t q = __umulhi(x[i], 0x9E3779B9u); x[i] = x[i] * b - q * 134215681u; }  // shoup mul
            if (OP == 8) x[i] = __umul24(x[i], b) + x[i];
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + (uint32_t)y[i] + (uint32_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int ops_per_iter, uint32_t* d_out) {
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
    double lane_ops = 5.0 * grid.x * block.x * (double)ITER * 8 * ops_per_iter;
    double gops = lane_ops / (ms * 1e-3) / 1e9;
    // wave-instructions per second per SIMD: chip has 1024 SIMDs
    double winst = lane_ops / 64.0 / (ms * 1e-3) / 1024.0;
    printf("%-16s %9.1f Glane-op/s   %.2f cycles per wave-instr per SIMD @2.4GHz (%.2f @2.1GHz)\n", name, gops, 2.4e9 / winst, 2.1e9 / winst);
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", 1, d);
    run<1>("v_mul_lo_u32", 1, d);
    run<2>("v_mul_hi_u32", 1, d);
    run<3>("v_mad_u64_u32", 1, d);
    run<4>("v_mul_u32_u24", 1, d);
    run<8>("v_mad_u32_u24", 1, d);
    run<5>("v_fma_f64", 1, d);
    run<6>("sub+min", 2, d);
    run<7>("shoup(3mul+sub)", 4, d);
    return 0;
}
