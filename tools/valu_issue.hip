// VALU issue cost per opcode on gfx950, measured so that block placement cannot distort it (development aid,
// feeds tools/valu_model.py and the `valu` roof of bench.py).
//
// Every configuration pins the number of resident waves per SIMD: ONE workgroup per CU is forced with a 100 KiB
// dynamic-LDS request (two cannot share a CU's 160 KiB), TWO with 64 KiB; the grid is exactly #CUs x that count.
// The time is the LONGEST per-block interval (s_memrealtime inside the kernel, max over blocks), not the kernel
// wall time, and the clock is taken from s_memtime of the same interval.  Each wave runs 8 independent
// dependency chains of the opcode.   build: hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define CHAINS(INS)                                                                                     \
    asm volatile(INS : "+v"(d0) : "v"(a), "v"(b)); asm volatile(INS : "+v"(d1) : "v"(a), "v"(b));       \
    asm volatile(INS : "+v"(d2) : "v"(a), "v"(b)); asm volatile(INS : "+v"(d3) : "v"(a), "v"(b));       \
    asm volatile(INS : "+v"(d4) : "v"(a), "v"(b)); asm volatile(INS : "+v"(d5) : "v"(a), "v"(b));       \
    asm volatile(INS : "+v"(d6) : "v"(a), "v"(b)); asm volatile(INS : "+v"(d7) : "v"(a), "v"(b));
#define BODY(INS) for (int i = 0; i < iters; ++i) { REP8(CHAINS(INS)) }

enum Op { MAD64, MUL_HI, MUL_LO, ADD, SUB, MIN, LSHL_ADD, ADD_LSHL, BFE, CNDMASK, AND, MOV, LSHLREV, ADD3,
          FMA64, MUL64, ADD64, RNDNE64, N_OPS };
static const char* kNames[N_OPS] = {"v_mad_u64_u32", "v_mul_hi_u32", "v_mul_lo_u32", "v_add_u32_e32", "v_sub_u32_e32", "v_min_u32_e32",
                                    "v_lshl_add_u32", "v_add_lshl_u32", "v_bfe_u32", "v_cndmask_b32_e32", "v_and_b32_e32", "v_mov_b32_e32",
                                    "v_lshlrev_b32_e32", "v_add3_u32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_rndne_f64"};

template <int OP>
__global__ void k(unsigned long long* ticks, unsigned long long* real, uint32_t* sink, int iters) {
    extern __shared__ uint32_t lds[];
    uint32_t a = threadIdx.x * 2654435761u + 1, b = (a ^ 0x9e3779b9u) | 1u;
    if (threadIdx.x == 99999) lds[0] = a;  // keep the allocation
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    if constexpr (OP == MAD64) {
        uint64_t d0 = a, d1 = b, d2 = a + 1, d3 = b + 1, d4 = a + 2, d5 = b + 2, d6 = a + 3, d7 = b + 3;
        for (int i = 0; i < iters; ++i) {
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d0) : "v"(a), "v"(b) : "vcc");
                 asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d1) : "v"(a), "v"(b) : "vcc");
                 asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d2) : "v"(a), "v"(b) : "vcc");
                 asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d3) : "v"(a), "v"(b) : "vcc");
                 asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d4) : "v"(a), "v"(b) : "vcc");
                 asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d5) : "v"(a), "v"(b) : "vcc");
                 asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d6) : "v"(a), "v"(b) : "vcc");
                 asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d7) : "v"(a), "v"(b) : "vcc");)
        }
        sink[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    } else if constexpr (OP >= FMA64) {
        double fa = threadIdx.x * 1.0000001 + 1.0, fb = 0.99999;
        double d0 = fa, d1 = fb, d2 = fa + 1, d3 = fb + 1, d4 = fa + 2, d5 = fb + 2, d6 = fa + 3, d7 = fb + 3;
        {
            const double a = fa, b = fb;
            if constexpr (OP == FMA64) { BODY("v_fma_f64 %0, %1, %2, %0") }
            if constexpr (OP == MUL64) { BODY("v_mul_f64 %0, %1, %2") }
            if constexpr (OP == ADD64) { BODY("v_add_f64 %0, %1, %0") }
            if constexpr (OP == RNDNE64) { BODY("v_rndne_f64 %0, %1") }
        }
        sink[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    } else {
        uint32_t d0 = a, d1 = b, d2 = a + 1, d3 = b + 1, d4 = a + 2, d5 = b + 2, d6 = a + 3, d7 = b + 3;
        if constexpr (OP == MUL_HI) { BODY("v_mul_hi_u32 %0, %0, %1") }
        if constexpr (OP == MUL_LO) { BODY("v_mul_lo_u32 %0, %0, %1") }
        if constexpr (OP == ADD) { BODY("v_add_u32_e32 %0, %0, %1") }
        if constexpr (OP == SUB) { BODY("v_sub_u32_e32 %0, %0, %1") }
        if constexpr (OP == MIN) { BODY("v_min_u32_e32 %0, %0, %1") }
        if constexpr (OP == LSHL_ADD) { BODY("v_lshl_add_u32 %0, %1, 3, %0") }
        if constexpr (OP == ADD_LSHL) { BODY("v_add_lshl_u32 %0, %0, %1, 1") }
        if constexpr (OP == BFE) { BODY("v_bfe_u32 %0, %0, 3, 7") }
        if constexpr (OP == CNDMASK) { BODY("v_cndmask_b32_e32 %0, %0, %1, vcc") }
        if constexpr (OP == AND) { BODY("v_and_b32_e32 %0, %0, %1") }
        if constexpr (OP == MOV) { BODY("v_mov_b32_e32 %0, %1") }
        if constexpr (OP == LSHLREV) { BODY("v_lshlrev_b32_e32 %0, 1, %0") }
        if constexpr (OP == ADD3) { BODY("v_add3_u32 %0, %1, %2, %0") }
        sink[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {  // the LAST wave of the block to finish defines the block's interval
        atomicMax(&ticks[blockIdx.x], t1 - t0);
        atomicMax(&real[blockIdx.x], r1 - r0);
    }
}

template <int OP>
void run(int cus, int threads, int blocks_per_cu, int iters) {
    const int blocks = cus * blocks_per_cu;
    const size_t lds = blocks_per_cu == 1 ? 100 * 1024 : 64 * 1024;
    unsigned long long *dt, *dr; uint32_t* sink;
    hipMalloc(&dt, blocks * 8); hipMalloc(&dr, blocks * 8); hipMalloc(&sink, (size_t)blocks * threads * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), lds, 0, dt, dr, sink, 16);
    hipDeviceSynchronize();
    hipMemset(dt, 0, blocks * 8); hipMemset(dr, 0, blocks * 8);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), lds, 0, dt, dr, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(blocks), r(blocks);
    hipMemcpy(t.data(), dt, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dr, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(t.begin(), t.end()); std::sort(r.begin(), r.end());
    const double waves_per_simd = threads / 64.0 / 4.0 * blocks_per_cu;
    const double insts_per_simd = 64.0 * iters * waves_per_simd;
    const double ns_med = r[blocks / 2] * 10.0 / insts_per_simd, ns_max = r[blocks - 1] * 10.0 / insts_per_simd;
    const double cyc_med = (double)t[blocks / 2] / insts_per_simd, mhz = (double)t[blocks / 2] / (r[blocks / 2] / 100.0);
    printf("{\"op\": \"%s\", \"waves_per_simd\": %.0f, \"ns_per_wave_inst_per_simd\": %.4f, \"ns_slowest_block\": %.4f, \"cycles\": %.3f, \"clock_mhz\": %.0f}\n",
           kNames[OP], waves_per_simd, ns_med, ns_max, cyc_med, mhz);
    fflush(stdout);
    hipFree(dt); hipFree(dr); hipFree(sink);
}

template <int OP>
void sweep(int cus) {
    run<OP>(cus, 256, 1, 40000);    // 1 wave per SIMD
    run<OP>(cus, 512, 1, 40000);    // 2
    run<OP>(cus, 1024, 1, 20000);   // 4
    run<OP>(cus, 1024, 2, 20000);   // 8
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    sweep<MAD64>(cus); sweep<MUL_HI>(cus); sweep<MUL_LO>(cus); sweep<ADD>(cus); sweep<SUB>(cus); sweep<MIN>(cus);
    sweep<LSHL_ADD>(cus); sweep<ADD_LSHL>(cus); sweep<BFE>(cus); sweep<CNDMASK>(cus); sweep<AND>(cus); sweep<MOV>(cus);
    sweep<LSHLREV>(cus); sweep<ADD3>(cus); sweep<FMA64>(cus); sweep<MUL64>(cus); sweep<ADD64>(cus); sweep<RNDNE64>(cus);
    return 0;
}
