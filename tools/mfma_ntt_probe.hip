// Bounded experiment: could the matrix pipe carry part of the forward transform of the digit
// polynomials?  TIMING PROBE ONLY -- both kernels below execute the instruction streams of the two candidates on
// arbitrary data; neither produces a transform (results are meaningless by construction).
//
// Setting: one blind-rotation step transforms 8 digit polynomials (N = 1024, 7-bit signed digits).  The first five
// Cooley-Tukey stages (bits 9..5) are, for every residue n_lo of the low five index bits, a 32-point linear map
// y[b][n_lo] = sum_{n_hi} M[b][n_hi] x[32 n_hi + n_lo] with the CONSTANT matrix M[b][n_hi] = psi^(32 n_hi (2 brv5(b) + 1)):
// a 32 x 32 by 32 x 256 integer GEMM per step.  With M split into four balanced base-128 limbs (int8) and the raw
// digits as int8, v_mfma_i32_32x32x32_i8 computes the four limb products exactly (|sum| <= 32 * 64 * 64 = 2^17); each
// output element is then y = Y0 + 2^7 Y1 + 2^14 Y2 + 2^21 Y3 (|y| < 2^38) reduced mod Q by one Barrett step.
//
//   k_valu5 : what the engine does today for five stages of one polynomial per wave: 16 coefficients per lane,
//             5 x 8 lazy Shoup butterflies on register pairs (v_mul_hi_u32 + 2 v_mad_u64_u32 + v_add_lshl_u32 +
//             v_sub_u32), twiddles from LDS.
//   k_mfma5 : the candidate: per wave 4 x v_mfma_i32_32x32x32_i8 (A = limb fragments resident in registers, B = the
//             digit bytes of one polynomial from LDS), then for each of the lane's 16 outputs the recombination and
//             reduction (2 x v_lshl_add, sign extension, v_mad_i64_i32, 64-bit offset add, v_alignbit, v_mul_hi_u32,
//             v_mad_u64_u32), plus the byte packing of the lane's digits and the LDS round trip of the int8 image.
// Both run 8 waves per workgroup, two workgroups per CU (the saturated launch shape), ITER iterations; the figure
// reported is shader cycles per iteration per workgroup (s_memtime of the slowest wave of workgroup 0) and the whole-
// launch time.     build: hipcc --offload-arch=gfx950 -O3 -o mfma_ntt_probe mfma_ntt_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
using u32 = uint32_t;
using u64 = uint64_t;

__device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c) {
    u64 r, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return r;
}

constexpr u32 Q = 134215681u;

__global__ __launch_bounds__(512, 4) void k_valu5(unsigned long long* ticks, u32* sink, int iters) {
    __shared__ uint2 tw[1024];
    __shared__ u32 poly[8][1088];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (u32 i = tid; i < 1024; i += 512) tw[i] = make_uint2(i * 2654435761u % Q, i * 40503u);
    for (u32 i = lane; i < 1088; i += 64) poly[wave][i] = (i * 7919u + wave) % Q;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    u32 acc = 0;
    for (int it = 0; it < iters; ++it) {
        u64 x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = poly[wave][((u32)r << 6) + lane + (((u32)r << 6) >> 6 << 2)];
        // five stages on the four register bits + one more on a lane-varying twiddle set: 40 butterflies
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int rb = s & 3;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r & (1 << rb)) continue;
                const uint2 w = tw[(1u << (s + 1)) + ((lane >> s) & ((1u << (s + 1)) - 1u)) + (u32)(r >> (rb + 1))];
                const u32 X = (u32)x[r], Y = (u32)x[r | (1 << rb)];
                const u64 t = mad64(__umulhi(Y, w.y), 0u - Q, mad64(Y, w.x, x[r]));
                x[r | (1 << rb)] = (x[r | (1 << rb)] & 0xFFFFFFFF00000000ull) | (u32)(((X + Q) << 1) - (u32)t);
                x[r] = t;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) poly[wave][((u32)r << 6) + lane + ((u32)r << 2)] = (u32)x[r];
        acc += (u32)x[3];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && lane == 0) atomicMax(ticks, t1 - t0);
    sink[blockIdx.x * 512 + tid] = acc;
}

__global__ __launch_bounds__(512, 4) void k_mfma5(unsigned long long* ticks, u32* sink, int iters) {
    __shared__ uint8_t img[8][1024];   // int8 image of one polynomial's digits per wave: [n_hi][n_lo]
    __shared__ u32 poly[8][1088];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (u32 i = lane; i < 1088; i += 64) poly[wave][i] = (i * 7919u + wave) % Q;
    __syncthreads();
    // limb fragments of the constant matrix: resident in registers for the whole kernel (4 limbs x 16 bytes per lane)
    v4i A[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) A[l] = v4i{(int)(lane * 0x01010101u + l), (int)(lane * 0x02030405u), (int)(l * 0x11111111u), (int)lane};
    const u32 mu = (u32)((1ull << 57) / Q);   // Barrett constant for x >> 25
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    u32 acc = 0;
    for (int it = 0; it < iters; ++it) {
        // the lane's 16 digits arrive as four words of the inverse transform (4 coefficients x 4 digits in the engine):
        // pack the four digits of one position group into one dword and store the int8 image (4 x ds_write_b32)
        u32 d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32 u = poly[wave][lane + 64 * k] + it;
            d[k] = (__builtin_amdgcn_ubfe(u, 0, 7)) | (__builtin_amdgcn_ubfe(u, 7, 7) << 8) | (__builtin_amdgcn_ubfe(u, 14, 7) << 16) | (__builtin_amdgcn_ubfe(u, 21, 7) << 24);
            reinterpret_cast<u32*>(img[wave])[lane + 64 * k] = d[k];
        }
        __builtin_amdgcn_wave_barrier();
        const v4i B = *reinterpret_cast<const v4i*>(&img[wave][lane * 16]);
        v16i Y[4];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            Y[l] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            Y[l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[l], B, Y[l], 0, 0, 0);
        }
        // recombination + one Barrett step per output element (16 per lane)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int t = Y[0][e] + (Y[1][e] << 7), u = Y[2][e] + (Y[3][e] << 7);
            long long y = (long long)t + (long long)u * 16384 + ((long long)Q << 12);
            const u32 x1 = (u32)((u64)y >> 25);
            const u32 r = (u32)mad64(__umulhi(x1, mu), 0u - Q, (u64)y);
            poly[wave][((u32)e << 6) + lane + ((u32)e << 2)] = r;
            acc += r;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && lane == 0) atomicMax(ticks, t1 - t0);
    sink[blockIdx.x * 512 + tid] = acc;
}

template <typename K>
void run(const char* name, K kern, int blocks, int iters) {
    unsigned long long* dt; u32* sink;
    hipMalloc(&dt, 8); hipMalloc(&sink, (size_t)blocks * 512 * 4);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, dt, sink, 8);
    hipDeviceSynchronize();
    hipMemset(dt, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, dt, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
    printf("%-8s grid %4d x 512: %8.3f ms, %8.1f shader cycles per iteration per workgroup (8 polynomials, 5 stages each)\n", name, blocks, ms,
           (double)t / iters);
    hipFree(dt); hipFree(sink);
}

int main() {
    const int iters = 20000;
    for (int blocks : {256, 512}) {
        run("valu5", k_valu5, blocks, iters);
        run("mfma5", k_mfma5, blocks, iters);
    }
    return 0;
}
