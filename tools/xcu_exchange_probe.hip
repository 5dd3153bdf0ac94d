// xcu_exchange_probe.hip -- what ONE step of a bootstrap split over two compute units would pay for its exchange
// (development measurement, not part of the library; VERDICT r2 "missing 3": measure, do not estimate).
//
// A two-CU split of AddToAcc gives each CU one accumulator component: its inverse transform, its digits' forward
// transforms and its half of the MAC.  The MAC of either half needs ALL transformed digit rows, so per step each CU sends
// the partner its 3 transformed digit rows + its evaluation-form accumulator row (folded key): 4 x 4 KiB = 16 KiB, and
// receives as much, 502 times per bootstrap.  This probe runs exactly that traffic and nothing else: pairs of 512-thread
// workgroups, per iteration   store 16 KiB (write-through, sc1) -> every wave drains -> barrier -> one lane stores the
// flag (sc1)  ||  poll the partner's flag (relaxed, sc1) -> barrier -> load the partner's 16 KiB (sc1 loads)   -- the
// hand-off protocol of the MI355X guide (Guideline 16, R1: sc1 payload + drained flag, consumer loads all sc1).
//
// Budget: one bootstrap alone takes 1.87 ms = 3.7 us per step today; the split halves at best the arithmetic (1.85 us per
// step), so reaching 1.4 ms (2.8 us per step) leaves 0.95 us per step for the exchange.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/xcu_exchange_probe tools/xcu_exchange_probe.hip && tools/xcu_exchange_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned int u32;
typedef __attribute__((address_space(1))) u32 gu32;

constexpr int T = 512, WORDS = 4096;   // 16 KiB per direction and step

__global__ __launch_bounds__(T) void k_exchange(u32* buf, u32* flags, u32* bad, u32* xcc_of, int steps, int partner_xor, int spin_limit) {
    const u32 me = blockIdx.x, other = me ^ (u32)partner_xor;
    u32* mine = buf + (size_t)me * 2 * WORDS;          // double-buffered by step parity
    const u32* theirs = buf + (size_t)other * 2 * WORDS;
    if (threadIdx.x == 0) xcc_of[me] = __builtin_amdgcn_s_getreg((31u << 11) | 20u) & 15u;
    u32 acc = 0, errors = 0;
    for (int s = 1; s <= steps; ++s) {
        u32* dst = mine + (s & 1) * WORDS;
        // payload: 512 threads x 32 bytes, write-through
        for (int k = 0; k < 2; ++k) {
            const u32 i = (threadIdx.x * 2 + k) * 4;
            typedef u32 v4 __attribute__((ext_vector_type(4)));
            const v4 v = {(u32)s * 131u + i, (u32)s * 131u + i + 1, (u32)s * 131u + i + 2, (u32)s * 131u + i + 3};
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(dst, 0, WORDS * 4, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(v, r, i * 4, 0, 16);     // aux 16 = sc1
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store((gu32*)(flags + me * 32), (u32)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load((const gu32*)(flags + other * 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u32)s) {
                if (++spins > spin_limit) { bad[0] = 1; break; }             // bounded: a lost partner ends the probe
            }
        }
        __syncthreads();
        const u32* src = theirs + (s & 1) * WORDS;
        for (int k = 0; k < 2; ++k) {
            const u32 i = (threadIdx.x * 2 + k) * 4;
            typedef u32 v4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(src), 0, WORDS * 4, 0x00020000);
            const v4 v = __builtin_amdgcn_raw_buffer_load_b128(r, i * 4, 0, 16);   // sc1: bypass this CU's L1
            errors += (v.x != (u32)s * 131u + i) + (v.w != (u32)s * 131u + i + 3);
            acc += v.y;
        }
        if (bad[0]) break;
    }
    if (errors) atomicAdd(bad + 1, errors);
    if (acc == 0x12345u) bad[2] = acc;   // keep the loads
}

int main(int argc, char** argv) {
    const int steps = 502 * 4;
    u32 *buf, *flags, *bad, *xcc;
    const int max_blocks = 512;
    CK(hipMalloc(&buf, (size_t)max_blocks * 2 * WORDS * 4));
    CK(hipMalloc(&flags, max_blocks * 32 * 4));
    CK(hipMalloc(&bad, 16));
    CK(hipMalloc(&xcc, max_blocks * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::printf("%8s %10s %12s %12s %s\n", "blocks", "pairing", "us/step", "ms/502steps", "partner XCD");
    for (int blocks : {2, 16, 64, 256}) {
        for (int px : {1, 8}) {
            if (px >= blocks) continue;
            float best = 1e30f;
            int same = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(flags, 0, max_blocks * 32 * 4));
                CK(hipMemset(bad, 0, 16));
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k_exchange, dim3(blocks), dim3(T), 0, 0, buf, flags, bad, xcc, steps, px, 50000000);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, a, b));
                u32 hb[4];
                CK(hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost));
                if (hb[0] || hb[1]) { std::printf("probe failed: timeout %u, payload errors %u\n", hb[0], hb[1]); return 2; }
                best = ms < best ? ms : best;
                std::vector<u32> hx(blocks);
                CK(hipMemcpy(hx.data(), xcc, blocks * 4, hipMemcpyDeviceToHost));
                same = hx[0] == hx[px];
            }
            std::printf("%8d %10s %12.3f %12.3f %s\n", blocks, px == 1 ? "b ^ 1" : "b ^ 8", best * 1e3f / steps, best / steps * 502,
                        same ? "same" : "other");
        }
    }
    std::printf("budget for a two-CU bootstrap of <= 1.4 ms: 0.95 us per step for the exchange (1.85 us of arithmetic per step at best)\n");
    return 0;
}
