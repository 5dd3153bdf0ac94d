// LDS access-shape cost on gfx950 (development aid): cycles per wave-instruction per CU for the
// load/store shapes the NTT passes use.  16 waves per CU, 256 CUs, 8 instructions between waits, inline asm.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32;
__device__ __forceinline__ u32 phys_(u32 j) { return j + ((j >> 6) << 2); }
__device__ __forceinline__ u32 xlay1_(u32 p) { return (p & 15u) + 20u * ((p >> 4) & 3u) + 80u * ((p >> 6) & 3u) + 320u * (p >> 8); }
__device__ __forceinline__ u32 xlay2_(u32 p) { return (p & 63u) + 80u * ((p >> 6) & 3u) + 320u * (p >> 8); }
__device__ __forceinline__ uint32_t swz(uint32_t L) { return ((L >> 1) & 3u) ^ (((L >> 3) & 1u) * 3u); }

// shape -> byte address for lane `l`, sub-instruction g (0..3)
__device__ __forceinline__ uint32_t addr_of(int shape, uint32_t l, uint32_t g) {
    switch (shape) {
        case 0: return 4u * (16u * l + 4u * (l >> 2) + 4u * g);            // b128 lane-major, pad 4/64 (phys)
        case 1: return 4u * (16u * l + 4u * (g ^ swz(l)));                 // b128 lane-major, xor swizzle
        case 2: return 4u * (20u * l + 4u * g);                            // b128 lane stride 20 words
        case 3: return 4u * (4u * l + 256u * g + 4u * ((4u * l) >> 6) * 0); // b128 consecutive lanes (16 B stride)
        case 4: return 4u * (l + 68u * g);                                 // b32 consecutive lanes
        case 5: return 4u * (2u * l + 136u * g);                           // b32 / b64 stride 2 words
        case 6: return 4u * (68u * (l >> 2) + 4u * g + (l & 3u));          // b32 pass-2 pattern (phys, LO=2)
        case 7: return 4u * (64u * g + 16u * (l & 3u) + 4u * (((l >> 2) & 7u) >> 1) + 2u * ((l >> 2) & 1u) + (l >> 5));  // b32 scatter into lane-major B
        case 8: return 4u * (68u * (l >> 2) + 4u * g + (((l & 1u) << 1) + ((l & 3u) >> 1)));  // b32 scatter into C
        case 9: return 4u * (16u * l + 4u * (l >> 2) + g);                 // b32 strided 16 words (transposed read)
        // ---- the split inverse transform's exchange layouts (kernels.hip), thread t = lane, register r = g ----
        case 10: { u32 pb = ((l >> 2) << 4) | (l & 3u); return 4u * (phys_(pb) + 4u * g); }            // e0 load
        case 11: { u32 pb = ((l >> 2) << 4) | (l & 3u); return 4u * (xlay1_(pb) + 4u * g); }           // e1 store
        case 12: { u32 pb = ((l >> 4) << 6) | (l & 15u); return 4u * (xlay1_(pb) + 20u * g); }         // e1 load
        case 13: { u32 pb = ((l >> 4) << 6) | (l & 15u); return 4u * (xlay2_(pb) + 16u * g); }         // e2 store
        case 14: return 4u * (xlay2_(l & 63u) + 80u * g);                                                // e2 load
        case 15: return 4u * ((l & 63u) + 64u * g);                                                      // e3 store
        case 16: return 4u * (l + 256u * g);                                                             // e3 load
        case 17: return 4u * (phys_(l) + 272u * g);                                                      // digit store
        case 18: { u32 j = ((l >> 4) << 8) | (g << 4) | (l & 15u); return 4u * phys_(j); }              // forward pass A load/store (r = g)
        case 19: { u32 j = ((l >> 4) << 8) | ((g + 4u) << 4) | (l & 15u); return 4u * phys_(j); }       // same, r = g + 4
        case 20: return 4u * phys_(4u * l) + 16u * 0u + 4352u * g;                                       // MAC b128 reads of dct rows (consecutive 16 B, padded)
        // candidates
        case 30: { u32 pb = ((l >> 2) << 4) | (l & 3u); return 4u * (pb + 4u * (pb >> 4) + 4u * g); }   // e0' load: pad 4 words per 16
        case 31: return 4u * (4u * l + 4u * (l >> 2)) + 5120u * g;                                       // e0' b128 store (thread t -> 4t)
        case 32: return 16u * l + 16u * (l >> 3) + 8192u * g;    // consecutive 16 B, 16-B pad every 8 lanes
        case 33: return 16u * l + 16u * (l >> 5) + 8192u * g;    // ... every 32 lanes
        case 34: return 16u * l + 32u * (l >> 4) + 8192u * g;    // 32-B pad every 16 lanes
        case 35: return 16u * l + 64u * (l >> 4) + 8192u * g;    // 64-B pad every 16 lanes
        case 36: return 16u * l + 128u * (l >> 4) + 8192u * g;   // 128-B pad every 16 lanes
        default: return 4u * l;
    }
}

template <int KIND>  // 0 read_b32, 1 read_b64, 2 read_b128, 3 write_b32, 4 write_b64, 5 write_b128, 6 read2_b64
__global__ void k(uint32_t* out, int shape, int iters) {
    extern __shared__ uint32_t lds[];
    const uint32_t l = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = i;
    __syncthreads();
    uint32_t a[4];
    for (int g = 0; g < 4; ++g) a[g] = addr_of(shape, l, g) + w * 8192u;  // each wave its own 8 KiB window (16 waves = 128 KiB)
    uint32_t acc = 0;
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    typedef uint32_t v2 __attribute__((ext_vector_type(2)));
    v4 d4 = {l, l + 1, l + 2, l + 3};
    v2 d2 = {l, l + 1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) {
            const int g = gg & 3;
            if (KIND == 0) { uint32_t r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(a[g])); acc += r; }
            if (KIND == 1) { v2 r; asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(a[g])); acc += r.x; }
            if (KIND == 2) { v4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(a[g])); acc += r.x; }
            if (KIND == 3) asm volatile("ds_write_b32 %0, %1" ::"v"(a[g]), "v"(l));
            if (KIND == 4) asm volatile("ds_write_b64 %0, %1" ::"v"(a[g]), "v"(d2));
            if (KIND == 5) asm volatile("ds_write_b128 %0, %1" ::"v"(a[g]), "v"(d4));
            if (KIND == 6) { v4 r; asm volatile("ds_read2_b64 %0, %1 offset1:68" : "=v"(r) : "v"(a[g])); acc += r.x; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int KIND>
void run(const char* name, int shape, uint32_t* d) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 131072, 0, d, shape, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 131072, 0, d, shape, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per CU: 16 waves x iters x 8 wave-instructions
    double ns_per_inst = ms * 1e6 / (16.0 * iters * 8);
    printf("%-14s shape %d : %.2f ns per wave-instr per CU  = %.1f cycles @2.1GHz\n", name, shape, ns_per_inst, ns_per_inst * 2.1);
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 1024 * 4);
    run<0>("ds_read_b32", 30, d);
    run<5>("ds_write_b128", 31, d);
    run<2>("ds_read_b128", 31, d);
    for (int s : {32, 33, 34, 35, 36}) run<2>("ds_read_b128", s, d);
    for (int s : {10, 12, 14, 16, 18, 19}) run<0>("ds_read_b32", s, d);
    for (int s : {11, 13, 15, 17, 18, 19}) run<3>("ds_write_b32", s, d);
    run<2>("ds_read_b128", 20, d);
    run<5>("ds_write_b128", 20, d);
    for (int s : {0, 1, 2, 3}) run<2>("ds_read_b128", s, d);
    for (int s : {0, 1, 2, 3}) run<5>("ds_write_b128", s, d);
    for (int s : {4, 5, 6, 7, 8, 9}) run<0>("ds_read_b32", s, d);
    for (int s : {4, 5, 6, 7, 8, 9}) run<3>("ds_write_b32", s, d);
    for (int s : {5, 3}) run<1>("ds_read_b64", s, d);
    for (int s : {5}) run<6>("ds_read2_b64", s, d);
    for (int s : {5, 3}) run<4>("ds_write_b64", s, d);
    return 0;
}
