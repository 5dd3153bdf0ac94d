// kernels64.hip -- 64-bit-modulus variants (ring modulus 2^28 <= Q < 2^62, e.g. STD192's 37-bit Q)
// of the blind-rotation / NTT kernels in kernels.hip.
//
// Same organisation as the 32-bit path (one workgroup per gate bootstrap, one 64-wide wavefront
// per RGSW row polynomial, E = N/64 coefficients per lane in registers, log2(E) radix-2 stages
// per register pass, LDS re-shuffles between passes) and the same reference hot path
// (EvalBinGate, src/gate.cpp:133,146,172,200-202).  Differences: words are u64; the 64-bit
// headroom makes BOTH transforms fully lazy (no per-stage correction at all); RGSW MAC sums are
// accumulated in 128 bits and reduced once; twiddles are read from global memory (L1/L2) because
// (2 + 2*dG) padded polynomials of N = 2048 u64 words already take 139 KiB of the 160 KiB LDS.
#include <hip/hip_runtime.h>

#include <utility>

#include <type_traits>

#include "kernels.hpp"
#include "phase_prof.hpp"

namespace bce {
#include "fused_tail.hpp"
#include "dag_sched.hpp"
#ifdef BCE_PHASE_PROF
__device__ unsigned long long g_phase_prof64[BCE_PROF_WAVES * BCE_PROF_SLOTS];
#define BCE_PROF_ARRAY ::bce::g_phase_prof64
#endif
namespace w64 {

__device__ __forceinline__ u64 csub(u64 x, u64 m) { return min(x, x - m); }

// y * w mod Q lazily, result in [0, 2Q) for any 64-bit y; w = (value, floor(value * 2^64 / Q))
__device__ __forceinline__ u64 mul_shoup_lazy(u64 y, ulonglong2 w, u64 Q) {
    return w.x * y - __umul64hi(w.y, y) * Q;
}

struct U128 {
    u64 lo, hi;
};
__device__ __forceinline__ void mac128(U128& a, u64 x, u64 y) {
    const u64 lo = x * y;
    a.lo += lo;
    a.hi += __umul64hi(x, y) + (a.lo < lo ? 1 : 0);
}
__device__ __forceinline__ void add128(U128& a, u64 x) {
    a.lo += x;
    a.hi += (a.lo < x ? 1 : 0);
}
// (hi * 2^64 + lo) mod Q for hi * c64 < 2^63, c64 = 2^64 mod Q, mu64 = floor(2^64 / Q)
__device__ __forceinline__ u64 reduce128(U128 a, u64 Q, u64 c64, u64 mu64) {
    const u64 t = a.hi * c64;
    u64 s = t + a.lo;
    if (s < t) s += c64;  // wrapped once: 2^64 = c64 (mod Q); s < t < 2^63 so no second wrap
    u64 r = s - __umul64hi(s, mu64) * Q;  // quotient estimate low by at most 2
    r = csub(r, 2 * Q);
    return csub(r, Q);
}

__device__ __forceinline__ u32 phys(u32 j) { return j + ((j >> 6) << 2); }

template <int LOGN>
struct Cfg {
    static constexpr int N = 1 << LOGN;
    static constexpr int LE = LOGN - 6;
    static constexpr int E = 1 << LE;
    static constexpr int NP = N + (N >> 6) * 4;
    static constexpr int F2LO = (6 > LE) ? 6 - LE : 0;
    static_assert(LOGN >= 9 && LOGN <= 11, "supported ring sizes: 512, 1024, 2048");
};

// Workgroup barrier that orders LDS traffic only (__syncthreads() would also drain the key-row loads in flight)
__device__ __forceinline__ void block_sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// 16 bytes of a key row through a buffer resource: wave-uniform base + SGPR row offset + one per-thread offset
__device__ __forceinline__ ulonglong2 key_row(__amdgpu_buffer_rsrc_t rsrc, u32 voff, u32 soff) {
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return make_ulonglong2(((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z);
}

template <typename F, u32... K>
__device__ __forceinline__ void for_each_index(F&& f, std::integer_sequence<u32, K...>) {
    (f(std::integral_constant<u32, K>{}), ...);
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int LOGN, int LO>
__device__ __forceinline__ u32 elem_j(u32 lane, int r) {
    constexpr int LE = Cfg<LOGN>::LE;
    return ((lane >> LO) << (LO + LE)) | ((u32)r << LO) | (lane & ((1u << LO) - 1u));
}
// LW: the LDS word of the row (u64, or u32 for the digit rows of the narrow build: values below 2^32 only)
template <int LOGN, int LO, typename LW>
__device__ __forceinline__ void load_pass(const LW* poly, u32 lane, u64 (&x)[Cfg<LOGN>::E]) {
#pragma unroll
    for (int r = 0; r < Cfg<LOGN>::E; ++r) x[r] = poly[phys(elem_j<LOGN, LO>(lane, r))];
}
template <int LOGN, int LO, typename LW>
__device__ __forceinline__ void store_pass(LW* poly, u32 lane, const u64 (&x)[Cfg<LOGN>::E]) {
#pragma unroll
    for (int r = 0; r < Cfg<LOGN>::E; ++r) poly[phys(elem_j<LOGN, LO>(lane, r))] = (LW)x[r];
}
// [0, Q) from any 64-bit value (the quotient estimate is low by at most 2)
template <int E>
__device__ __forceinline__ void normalise(u64 (&x)[E], u64 Q, u64 mu64) {
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const u64 v = x[r] - __umul64hi(x[r], mu64) * Q;
        x[r] = csub(csub(v, 2 * Q), Q);
    }
}

// forward (Cooley-Tukey) stage on index bit B, no correction: values grow by < 2Q per stage
template <int LOGN, int LO, int B>
__device__ __forceinline__ void fwd_stage(u64 (&x)[Cfg<LOGN>::E], u32 lane, const ulonglong2* __restrict__ tw, u64 Q) {
    constexpr int LE = Cfg<LOGN>::LE, E = Cfg<LOGN>::E, rb = B - LO;
    constexpr u32 m = 1u << (LOGN - 1 - B);
    const u32 hi = (lane >> LO) << (LO + LE - B - 1);
    const u64 Q2 = 2 * Q;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if (r & (1 << rb)) continue;
        const ulonglong2 w = tw[m + (hi | (u32)(r >> (rb + 1)))];
        const u64 X = x[r];
        const u64 T = mul_shoup_lazy(x[r | (1 << rb)], w, Q);
        x[r] = X + T;
        x[r | (1 << rb)] = X + Q2 - T;
    }
}
// inverse (Gentleman-Sande) stage number S (0 = first): inputs < 2^(S+1) Q, the sum is left
// uncorrected, the difference is offset by 2^(S+1) Q; inverse twiddles derived from the forward
// table (psi^-k = -psi^(N-k), Shoup companion = bitwise complement)
template <int LOGN, int LO, int B, int S>
__device__ __forceinline__ void inv_stage(u64 (&x)[Cfg<LOGN>::E], u32 lane, const ulonglong2* __restrict__ tw, u64 Q) {
    constexpr int LE = Cfg<LOGN>::LE, E = Cfg<LOGN>::E, rb = B - LO;
    constexpr u32 m = 1u << (LOGN - 1 - B);
    const u32 hi = (lane >> LO) << (LO + LE - B - 1);
    const u64 off = Q << (S + 1);
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if (r & (1 << rb)) continue;
        const ulonglong2 f = tw[(2 * m - 1) - (hi | (u32)(r >> (rb + 1)))];
        const ulonglong2 w = make_ulonglong2(Q - f.x, ~f.y);
        const u64 X = x[r], Y = x[r | (1 << rb)];
        x[r] = X + Y;
        x[r | (1 << rb)] = mul_shoup_lazy(X + off - Y, w, Q);
    }
}
template <int LOGN, int LO, int BHI, int BLO>
__device__ __forceinline__ void fwd_stages(u64 (&x)[Cfg<LOGN>::E], u32 lane, const ulonglong2* tw, u64 Q) {
    if constexpr (BHI >= BLO) {
        fwd_stage<LOGN, LO, BHI>(x, lane, tw, Q);
        fwd_stages<LOGN, LO, BHI - 1, BLO>(x, lane, tw, Q);
    }
}
template <int LOGN, int LO, int BLO, int BHI>
__device__ __forceinline__ void inv_stages(u64 (&x)[Cfg<LOGN>::E], u32 lane, const ulonglong2* tw, u64 Q) {
    if constexpr (BLO <= BHI) {
        inv_stage<LOGN, LO, BLO, BLO>(x, lane, tw, Q);  // stage number == bit index
        inv_stages<LOGN, LO, BLO + 1, BHI>(x, lane, tw, Q);
    }
}

// forward NTT by one wave in LDS; input < 2Q, output bit-reversed, < (2*LOGN+2) Q unless NORM.
// Rows of 32-bit words (narrow build, Q < 2^31): every store is of normalised values, NORM is implied.
template <int LOGN, bool NORM, typename LW>
__device__ __forceinline__ void ntt_forward_wave(LW* poly, const ulonglong2* tw, u32 lane, u64 Q, u64 mu64) {
    using C = Cfg<LOGN>;
    constexpr bool NARROW = sizeof(LW) == 4;
    u64 x[C::E];
    load_pass<LOGN, 6>(poly, lane, x);
    fwd_stages<LOGN, 6, LOGN - 1, 6>(x, lane, tw, Q);
    if constexpr (NARROW) normalise<C::E>(x, Q, mu64);
    store_pass<LOGN, 6>(poly, lane, x);
    wave_sync();
    load_pass<LOGN, C::F2LO>(poly, lane, x);
    fwd_stages<LOGN, C::F2LO, 5, C::F2LO>(x, lane, tw, Q);
    if constexpr (C::F2LO > 0) {
        if constexpr (NARROW) normalise<C::E>(x, Q, mu64);
        store_pass<LOGN, C::F2LO>(poly, lane, x);
        wave_sync();
        load_pass<LOGN, 0>(poly, lane, x);
        fwd_stages<LOGN, 0, C::F2LO - 1, 0>(x, lane, tw, Q);
    }
    if constexpr (NORM || NARROW) normalise<C::E>(x, Q, mu64);
    store_pass<LOGN, 0>(poly, lane, x);
    wave_sync();
}

// inverse NTT by one wave; src bit-reversed < 2Q; coefficient j = (r << 6) | lane in x[r], in [0, Q)
template <int LOGN>
__device__ __forceinline__ void ntt_inverse_wave(const u64* src, u64* tmp, const ulonglong2* tw, u32 lane, u64 Q,
                                                 ulonglong2 ninv, u64 (&x)[Cfg<LOGN>::E]) {
    using C = Cfg<LOGN>;
    constexpr int LE = C::LE;
    load_pass<LOGN, 0>(src, lane, x);
    inv_stages<LOGN, 0, 0, LE - 1>(x, lane, tw, Q);
    store_pass<LOGN, 0>(tmp, lane, x);
    wave_sync();
    load_pass<LOGN, LE>(tmp, lane, x);
    inv_stages<LOGN, LE, LE, 2 * LE - 1>(x, lane, tw, Q);
    store_pass<LOGN, LE>(tmp, lane, x);
    wave_sync();
    load_pass<LOGN, 6>(tmp, lane, x);
    inv_stages<LOGN, 6, 2 * LE, LOGN - 1>(x, lane, tw, Q);
#pragma unroll
    for (int r = 0; r < C::E; ++r) x[r] = csub(mul_shoup_lazy(x[r], ninv, Q), Q);
}

template <int LOGN>
__device__ __forceinline__ u64 psi_pow(const ulonglong2* tw, u32 e, u64 Q) {
    constexpr u32 N = 1u << LOGN;
    const u64 v = tw[__brev(e & (N - 1)) >> (32 - LOGN)].x;
    return (e & N) ? Q - v : v;
}

__device__ __forceinline__ u32 gate_const(u32 op, u32 q) {
    const u32 e = q >> 3;
    switch (op) {
        case BCE_OR: case BCE_XOR_FAST: return 5 * e;
        case BCE_NOR: case BCE_XNOR_FAST: return e;
        case BCE_NAND: return 3 * e;
        default: return 7 * e;
    }
}

// NBUF_ / NPRE_: depth of the key-row software pipeline (items in flight / requested before the transforms);
// the defaults are the deepest that compile without scratch at N = 2048 (GINX items carry two keys' rows)
// NARROW: the R digit rows are 32-bit words (Q < 2^31: a digit + Q and every normalised transform value fit), the inverse
// transforms get a scratch of their own -- what lets N = 2048 with FOUR gadget digits (STD256, STD256_OPT: 29-bit Q, base 2^8)
// into the 160 KiB of LDS: 2 x 17 KiB accumulator + 8 x 8.5 KiB digit rows + 2 x 17 KiB scratch instead of 10 x 17 KiB.
template <int LOGN, int DG, bool AP, u32 NBUF_ = (AP ? 3 : 2), u32 NPRE_ = (AP ? 2 : 1), bool NARROW = false>
__global__ __launch_bounds__(128 * DG) void k_blind_rotate64(DevParams P, const bce_gate_desc* __restrict__ descs, u32 n_desc,
                                                              u32 slot_stride, u64* __restrict__ acc_out, u32* /*dbg_lweN: no fused tail*/, u32* /*dbg_ks*/) {
    using C = Cfg<LOGN>;
    constexpr int N = C::N, NP = C::NP, E = C::E;
    constexpr u32 R = 2 * DG, T = 64 * R;
    extern __shared__ __align__(16) u64 smem64[];
    using DW = std::conditional_t<NARROW, u32, u64>;
    u64* acc = smem64;            // [2][NP] EVALUATION domain, [0, Q)
    DW* dct = reinterpret_cast<DW*>(acc + 2 * NP);      // [R][NP]
    // exchange rows of the two inverse transforms: the digit rows themselves, or (narrow build) two 64-bit rows behind them
    u64* inv_tmp = reinterpret_cast<u64*>(NARROW ? dct + R * NP : dct);
    u32* av = NARROW ? reinterpret_cast<u32*>(inv_tmp + 2 * NP) : reinterpret_cast<u32*>(dct + R * NP);
    const ulonglong2* __restrict__ tw = P.tw64;

    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 Q = P.Q64;
    const u32 q = P.q, qm = q - 1, n = P.n;
    const bce_gate_desc g = descs[blockIdx.x % n_desc];
    const u32 soff = (blockIdx.x / n_desc) * slot_stride;
    {
        const u32* in0 = P.pool + (size_t)(g.in0 + soff) * P.pool_stride;
        const u32* in1 = P.pool + (size_t)(g.in1 + soff) * P.pool_stride;
        const bool two = g.op <= BCE_XNOR_FAST;
        for (u32 i = tid; i <= n; i += T) {
            u32 v0 = in0[i];
            if (g.neg0) v0 = ((i == n ? (q >> 2) : 0u) - v0) & qm;
            u32 v = v0;
            if (two) {
                u32 v1 = in1[i];
                if (g.neg1) v1 = ((i == n ? (q >> 2) : 0u) - v1) & qm;
                v = (g.op == BCE_XOR_FAST || g.op == BCE_XNOR_FAST) ? (2u * (v0 - v1)) & qm : (v0 + v1) & qm;
            } else if (i == n) {
                v = (v0 + (q >> 2)) & qm;
            }
            av[i] = v;
        }
    }
    __syncthreads();
    {
        const u32 b = av[n];
        const u32 q1 = gate_const(g.op, q), q2 = (q1 + (q >> 1)) & qm;
        const u64 pos = P.Q8p1_64, neg = Q - P.Q8p1_64;
        for (u32 j = tid; j < (u32)N; j += T) {
            u64 v = 0;
            if (j % P.factor == 0) {
                u32 t = (b - j / P.factor) & qm;
                bool in = (q1 < q2) ? (t >= q1 && t < q2) : !(t >= q2 && t < q1);
                v = in ? neg : pos;
            }
            acc[phys(j)] = 0;
            acc[NP + phys(j)] = v;
        }
    }
    __syncthreads();
    if (wave == 0) ntt_forward_wave<LOGN, true>(acc + NP, tw, lane, Q, P.mu64);
    __syncthreads();

    const ulonglong2 ninv = make_ulonglong2(P.Ninv64, P.Ninv64_s);
    constexpr size_t rgsw = (size_t)R * 2 * N;
    const u64* __restrict__ bsk = P.bsk64;
    const u32 nsteps = AP ? n * P.dR : n;
    BCE_PROF_INIT();
    for (u32 step = 0; step < nsteps; ++step) {
        u32 ap = 0;
        const u64* bk;
        if constexpr (!AP) {
            ap = ((q - av[step]) & qm) * P.factor;
            if (ap == 0) continue;
            bk = bsk + (size_t)step * 2 * rgsw;
        } else {
            const u32 i = step / P.dR, k = step - i * P.dR;
            u32 aI = (q - av[i]) & qm;
            for (u32 t = 0; t < k; ++t) aI /= P.baseR;
            const u32 a0 = aI % P.baseR;
            if (a0 == 0) continue;
            bk = bsk + (((size_t)i * P.baseR + a0) * P.dR + k) * rgsw;
        }
        // ---- key rows: software pipeline.  A thread owns the MAC items tid, tid + T, ... (2 positions each);
        // the rows of one item are ROWS 16-byte loads.  The first NPRE items are requested here, before the
        // transforms (the barriers below order LDS only), up to NBUF at the start of the MAC phase, item k + NBUF
        // when item k is done.
        constexpr u32 ITEMS = (2u * (N / 2) + T - 1) / T;       // per thread, last one guarded
        constexpr u32 ROWS = AP ? R : 2 * R;
        constexpr u32 NBUF = NBUF_;   // items in flight
        constexpr u32 NPRE = NPRE_;   // of which requested before the transforms
        // one resource per step: base = this step's RGSW key(s), bounds = their size
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<u64*>(bk), 0, (int)((AP ? 1 : 2) * rgsw * sizeof(u64)), 0x00020000);
        ulonglong2 kb[NBUF][ROWS];
        // the per-item address arithmetic must stay INSIDE the step loop: hoisted, its ~30 loop-invariant values
        // would live across the transforms, which already use the whole register file (scratch reloads per step)
        u32 tid_v = tid, lane_v = lane;
        asm volatile("" : "+v"(tid_v), "+v"(lane_v));
        auto request = [&](auto kc) {
            constexpr u32 k = decltype(kc)::value;
            if constexpr (k < ITEMS) {
                const u32 item = tid_v + k * T;
                if (k + 1 < ITEMS || item < 2u * (N / 2)) {
                    const u32 c = item / (N / 2), p0 = (item % (N / 2)) * 2;
                    const u32 voff = (c * N + p0) * 8u;
#pragma unroll
                    for (u32 l = 0; l < R; ++l) {
                        kb[k % NBUF][l] = key_row(rsrc, voff, l * (2 * N * 8));
                        if constexpr (!AP) kb[k % NBUF][R + l] = key_row(rsrc, voff, (u32)(rgsw * 8) + l * (2 * N * 8));
                    }
                }
            }
        };
        if constexpr (NPRE >= 1) request(std::integral_constant<u32, 0>{});
        if constexpr (NPRE >= 2) request(std::integral_constant<u32, 1>{});
        if (wave < 2) {
            u64 x[E];
            ntt_inverse_wave<LOGN>(acc + wave * NP, inv_tmp + wave * NP, tw, lane_v, Q, ninv, x);
            const int gsh = 64 - (int)P.gBits;
            const u64 Qh = Q >> 1;
#pragma unroll
            for (int r = 0; r < E; ++r) {
                long long d = (x[r] < Qh) ? (long long)x[r] : (long long)x[r] - (long long)Q;
                const u32 pj = phys(((u32)r << 6) | lane_v);
#pragma unroll
                for (u32 l = 0; l < (u32)DG; ++l) {
                    long long rem = (long long)((u64)d << gsh) >> gsh;
                    d = (d - rem) >> P.gBits;
                    dct[(2 * l + wave) * NP + pj] = (DW)(u64)(rem + (long long)Q);  // in (Q - B/2, Q + B/2)
                }
            }
        }
        BCE_PROF_MARK(0);
        block_sync_lds();
        BCE_PROF_MARK(1);
        ntt_forward_wave<LOGN, NARROW>(dct + wave * NP, tw, lane_v, Q, P.mu64);
        BCE_PROF_MARK(2);
        block_sync_lds();
        BCE_PROF_MARK(3);
        if constexpr (NPRE < 1) request(std::integral_constant<u32, 0>{});
        if constexpr (NPRE < 2 && NBUF >= 2) request(std::integral_constant<u32, 1>{});
        if constexpr (NBUF >= 3) request(std::integral_constant<u32, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        const bool odd = ap & 1u;
        auto mac_item = [&](auto kc) {
            constexpr u32 k = decltype(kc)::value;
            const u32 item = tid_v + k * T;
            if (k + 1 < ITEMS || item < 2u * (N / 2)) {
                const u32 c = item / (N / 2), p0 = (item % (N / 2)) * 2;
                const u32 pp = phys(p0);
                U128 sp[2] = {{0, 0}, {0, 0}}, sn[2] = {{0, 0}, {0, 0}};
#pragma unroll
                for (u32 l = 0; l < R; ++l) {
                    ulonglong2 d;
                    if constexpr (NARROW) {
                        const uint2 dn = *reinterpret_cast<const uint2*>(dct + l * NP + pp);
                        d = make_ulonglong2(dn.x, dn.y);
                    } else {
                        d = *reinterpret_cast<const ulonglong2*>(dct + l * NP + pp);
                    }
                    const ulonglong2 kp = kb[k % NBUF][l];
                    mac128(sp[0], d.x, kp.x);
                    mac128(sp[1], d.y, kp.y);
                    if constexpr (!AP) {
                        const ulonglong2 kn = kb[k % NBUF][R + l];
                        mac128(sn[0], d.x, kn.x);
                        mac128(sn[1], d.y, kn.y);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the pipeline depth: do not hoist later requests above this point
                request(std::integral_constant<u32, k + NBUF>{});  // this item's buffer is free again
                __builtin_amdgcn_sched_barrier(0);
                u64 a[2];
                if constexpr (AP) {
                    a[0] = reduce128(sp[0], Q, P.c64, P.mu64);
                    a[1] = reduce128(sp[1], Q, P.c64, P.mu64);
                } else {
                    // positions p0, p0+1: brv(p0+1) = brv(p0) + N/2, so the monomials differ by psi^(N a') = (-1)^a'
                    const u32 k0 = __brev(p0) >> (32 - LOGN);
                    const u32 ex = ((2 * k0 + 1) * ap) & (2 * N - 1);
                    u64 mp[2], mn[2];
                    mp[0] = psi_pow<LOGN>(tw, ex, Q);
                    mn[0] = psi_pow<LOGN>(tw, (2 * N - ex) & (2 * N - 1), Q);
                    mp[1] = odd ? Q - mp[0] : mp[0];
                    mn[1] = odd ? Q - mn[0] : mn[0];
                    const ulonglong2 av2 = *reinterpret_cast<const ulonglong2*>(acc + c * NP + pp);
                    const u64 old[2] = {av2.x, av2.y};
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const u64 rp = reduce128(sp[e], Q, P.c64, P.mu64), rn = reduce128(sn[e], Q, P.c64, P.mu64);
                        U128 t = {0, 0};
                        mac128(t, rp, mp[e] - 1);
                        mac128(t, rn, mn[e] - 1);
                        add128(t, old[e]);
                        a[e] = reduce128(t, Q, P.c64, P.mu64);
                    }
                }
                *reinterpret_cast<ulonglong2*>(acc + c * NP + pp) = make_ulonglong2(a[0], a[1]);
            }
        };
        for_each_index(mac_item, std::make_integer_sequence<u32, ITEMS>{});
        BCE_PROF_MARK(4);
        block_sync_lds();
        BCE_PROF_MARK(5);
    }
    BCE_PROF_FLUSH();
    if (wave < 2) {
        u64 x[E];
        ntt_inverse_wave<LOGN>(acc + wave * NP, inv_tmp + wave * NP, tw, lane, Q, ninv, x);
        u64* out = acc_out + ((size_t)blockIdx.x * 2 + wave) * N;
#pragma unroll
        for (int r = 0; r < E; ++r) out[((u32)r << 6) | lane] = x[r];
    }
}

// batched NTT over global memory, one wave per polynomial
template <int LOGN>
__global__ __launch_bounds__(256) void k_ntt_batch64(DevParams P, u64* __restrict__ polys, u32 count, int inverse) {
    using C = Cfg<LOGN>;
    constexpr int N = C::N, NP = C::NP, E = C::E;
    extern __shared__ __align__(16) u64 smem64[];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    u64* mine = smem64 + wave * NP;
    for (u32 p = blockIdx.x * W + wave; p < count; p += gridDim.x * W) {
        u64* gp = polys + (size_t)p * N;
        for (int r = 0; r < E; ++r) {
            const u32 j = ((u32)r << 6) | lane;
            mine[phys(j)] = gp[j];
        }
        wave_sync();
        if (!inverse) {
            ntt_forward_wave<LOGN, true>(mine, P.tw64, lane, P.Q64, P.mu64);
            for (int r = 0; r < E; ++r) {
                const u32 j = ((u32)r << 6) | lane;
                gp[j] = mine[phys(j)];
            }
        } else {
            u64 x[E];
            ntt_inverse_wave<LOGN>(mine, mine, P.tw64, lane, P.Q64, make_ulonglong2(P.Ninv64, P.Ninv64_s), x);
#pragma unroll
            for (int r = 0; r < E; ++r) gp[((u32)r << 6) | lane] = x[r];
        }
        wave_sync();
    }
}

__global__ void k_pointwise_mac64(DevParams P, u64* __restrict__ b, const u64* __restrict__ a, const u64* __restrict__ z,
                                  u32 count, u32 b_step) {
    const size_t total = (size_t)count * P.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const u32 k = (u32)(i & (P.N - 1));
        const size_t bi = (i >> P.logN) * b_step * P.N + k;
        U128 t = {0, 0};
        mac128(t, a[i], z[k]);
        add128(t, b[bi]);
        b[bi] = reduce128(t, P.Q64, P.c64, P.mu64);
    }
}

}  // namespace w64

// =======================================================================================
// Double-precision formulation of the same blind rotation (Q < 2^39, selected by DevParams::fp64)
// =======================================================================================
// A 64-bit Shoup butterfly costs ~33 integer VALU instructions (three 64x64 multiplies built from 32-bit
// ones plus carries); gfx950 runs fp64 FMA at the 32-bit integer multiply rate, and with exact-integer
// doubles a modular product is 6 instructions:
//     h = y*w;  l = fma(y, w, -h);  q = rint(y * (w/Q));  r = fma(-q, Q, h) + l        (r = y*w - q*Q exactly)
// h + l is the exact product (fma), q is within 0.5 + 2^-4 of y*w/Q for |y| < 2^49, so h - q*Q is an integer
// below 2^38 in magnitude and therefore exact, as is the final sum: |r| <= 0.57 Q and r = y*w (mod Q).
// Every value in LDS / registers is an integer held in a double, |value| < 2^53 always:
//   forward: T = modmul(Y, w), X' = X + T, Y' = X - T         -> grows by < 0.57 Q per stage (< 7 Q after 11)
//   inverse: X' = X + Y, Y' = modmul(Y - X, -w)               -> sums grow 2x per stage, < 2^11 * 0.6 Q < 2^49;
//            no intermediate reduction at all, one modmul by N^-1 at the end
// so results are the same residues as the integer kernel's; the final accumulator is mapped to [0, Q) as u64,
// and the digit decomposition first maps to the reference's representative in [-(Q+1)/2, (Q-3)/2]
// (SignedDigitDecompose centres the canonical value with `x < Q/2`).  Key rows are stored as doubles in HBM
// (converted once after key generation / import).  Compiled with -ffp-contract=off: products are never fused
// except where fma() is written.
namespace wd {
using w64::Cfg;
using w64::phys;
using w64::elem_j;
using w64::wave_sync;
using w64::block_sync_lds;
// exchange of the split inverse transform that stays inside one wave; -DBCE_STEP_BARRIERS = the workgroup barriers of
// the earlier schedule (A/B runs)
#ifdef BCE_STEP_BARRIERS
constexpr int INV_BARRIERS = 3;
__device__ __forceinline__ void wave_local_sync() { block_sync_lds(); }
#else
constexpr int INV_BARRIERS = 1;
__device__ __forceinline__ void wave_local_sync() { wave_sync(); }
#endif
using w64::for_each_index;
using w64::gate_const;

// Twiddle table access: entries [0, lds_n) are mirrored in LDS (the 8-wave N = 2048 kernel has room for the first
// 1024 = every stage except the one on bit 0), the rest is read from global memory.  The block a stage uses
// ([m, 2m), m a compile-time constant) lies entirely on one side.
struct Tw {
    const double2* __restrict__ g;   // global, [N]
    const double2* l;                // LDS mirror of g[0 .. TW_LDS), or unused
    // W1 builds (folded key, N = 2048; round 4): the LDS mirror holds ALL N entries as plain doubles -- the value w only, 8 bytes
    // instead of (w, w / Q) -- and the quotient estimate comes from the product itself (modmul: rint(h * invQ) instead of
    // rint(y * (w / Q)), same six operations).  Half the twiddle bytes through LDS, no twiddle left in global memory.
    const double* l1 = nullptr;
    double invQ = 0.0;
    template <u32 M, u32 TW_LDS>
    __device__ __forceinline__ const double2* blk() const { return (2 * M <= TW_LDS) ? l : g; }
};

__device__ __forceinline__ double modmul_q(double y, double w, double wq, double Q) {  // wq = w / Q (rounded)
    const double h = y * w;
    const double l = fma(y, w, -h);
    const double q = rint(y * wq);
    return fma(-q, Q, h) + l;
}
__device__ __forceinline__ double modmul(double y, double w, double invQ, double Q) {
    const double h = y * w;
    const double l = fma(y, w, -h);
    const double q = rint(h * invQ);
    return fma(-q, Q, h) + l;
}
__device__ __forceinline__ double modred(double s, double invQ, double Q) { return fma(-rint(s * invQ), Q, s); }
// y * twiddle mod Q for either kind of table entry: (w, w / Q) pairs, or w alone (W1 builds).  Error of the quotient estimate
// for |y| < 2^49, w < Q < 2^39: below 0.125 with the pair, below 0.19 with w alone (three roundings: h, 1 / Q, their product),
// i.e. |result| <= 0.625 Q / 0.69 Q -- both inside the bounds the transforms are laid out for (inverse: sums of 2^11 values
// stay below 2^49; forward: 11 stages grow a 13-bit digit to < 7.6 Q < 2^40).
__device__ __forceinline__ double tmul(double y, double2 w, const Tw&, double Q) { return modmul_q(y, w.x, w.y, Q); }
__device__ __forceinline__ double tmul(double y, double w, const Tw& t, double Q) { return modmul(y, w, t.invQ, Q); }
// a 13-bit digit times a twiddle is below 2^53: the product is exact in ONE multiplication (no low part to recover)
__device__ __forceinline__ double tmul_digit(double d, double2 w, const Tw& t, double Q) { return modmul_q(d, w.x, w.y, Q); }
__device__ __forceinline__ double tmul_digit(double d, double w, const Tw& t, double Q) {
    const double h = d * w;
    return fma(-rint(h * t.invQ), Q, h);
}
template <bool W1> struct TwSel;
template <> struct TwSel<false> {
    typedef double2 T;
    static __device__ __forceinline__ double2 lds(const Tw& t, u32 i) { return t.l[i]; }
    static __device__ __forceinline__ double2 last(const Tw& t, u32 i) { return t.g[i]; }   // block m = N / 2: global memory
};
template <> struct TwSel<true> {
    typedef double T;
    static __device__ __forceinline__ double lds(const Tw& t, u32 i) { return t.l1[i]; }
    static __device__ __forceinline__ double last(const Tw& t, u32 i) { return t.l1[i]; }
};

template <int LOGN, int LO>
__device__ __forceinline__ void load_pass(const double* poly, u32 lane, double (&x)[Cfg<LOGN>::E]) {
#pragma unroll
    for (int r = 0; r < Cfg<LOGN>::E; ++r) x[r] = poly[phys(elem_j<LOGN, LO>(lane, r))];
}
template <int LOGN, int LO>
__device__ __forceinline__ void store_pass(double* poly, u32 lane, const double (&x)[Cfg<LOGN>::E]) {
#pragma unroll
    for (int r = 0; r < Cfg<LOGN>::E; ++r) poly[phys(elem_j<LOGN, LO>(lane, r))] = x[r];
}
template <int LOGN, int LO, int B, u32 TWL>
__device__ __forceinline__ void fwd_stage(double (&x)[Cfg<LOGN>::E], u32 lane, Tw twa, double Q) {
    constexpr int LE = Cfg<LOGN>::LE, E = Cfg<LOGN>::E, rb = B - LO;
    constexpr u32 m = 1u << (LOGN - 1 - B);
    const double2* tw = twa.blk<m, TWL>();
    const u32 hi = (lane >> LO) << (LO + LE - B - 1);
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if (r & (1 << rb)) continue;
        const double2 w = tw[m + (hi | (u32)(r >> (rb + 1)))];
        const double X = x[r];
        const double T = modmul_q(x[r | (1 << rb)], w.x, w.y, Q);
        x[r] = X + T;
        x[r | (1 << rb)] = X - T;
    }
}
// psi^-k = -psi^(N-k): entry (2m-1) - i of the forward table, the sign goes into the operand (Y - X)
template <int LOGN, int LO, int B, u32 TWL>
__device__ __forceinline__ void inv_stage(double (&x)[Cfg<LOGN>::E], u32 lane, Tw twa, double Q) {
    constexpr int LE = Cfg<LOGN>::LE, E = Cfg<LOGN>::E, rb = B - LO;
    constexpr u32 m = 1u << (LOGN - 1 - B);
    const double2* tw = twa.blk<m, TWL>();
    const u32 hi = (lane >> LO) << (LO + LE - B - 1);
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if (r & (1 << rb)) continue;
        const double2 f = tw[(2 * m - 1) - (hi | (u32)(r >> (rb + 1)))];
        const double X = x[r], Y = x[r | (1 << rb)];
        x[r] = X + Y;
        x[r | (1 << rb)] = modmul_q(Y - X, f.x, f.y, Q);
    }
}
template <int LOGN, int LO, int BHI, int BLO, u32 TWL = 0>
__device__ __forceinline__ void fwd_stages(double (&x)[Cfg<LOGN>::E], u32 lane, Tw tw, double Q) {
    if constexpr (BHI >= BLO) {
        fwd_stage<LOGN, LO, BHI, TWL>(x, lane, tw, Q);
        fwd_stages<LOGN, LO, BHI - 1, BLO, TWL>(x, lane, tw, Q);
    }
}
template <int LOGN, int LO, int BLO, int BHI, u32 TWL = 0>
__device__ __forceinline__ void inv_stages(double (&x)[Cfg<LOGN>::E], u32 lane, Tw tw, double Q) {
    if constexpr (BLO <= BHI) {
        inv_stage<LOGN, LO, BLO, TWL>(x, lane, tw, Q);
        inv_stages<LOGN, LO, BLO + 1, BHI, TWL>(x, lane, tw, Q);
    }
}
// forward NTT by one wave in LDS; |input| <= Q, |output| < 7 Q, bit-reversed order
template <int LOGN>
__device__ __forceinline__ void ntt_forward_wave(double* poly, Tw tw, u32 lane, double Q) {
    using C = Cfg<LOGN>;
    double x[C::E];
    load_pass<LOGN, 6>(poly, lane, x);
    fwd_stages<LOGN, 6, LOGN - 1, 6>(x, lane, tw, Q);
    store_pass<LOGN, 6>(poly, lane, x);
    wave_sync();
    load_pass<LOGN, C::F2LO>(poly, lane, x);
    fwd_stages<LOGN, C::F2LO, 5, C::F2LO>(x, lane, tw, Q);
    if constexpr (C::F2LO > 0) {
        store_pass<LOGN, C::F2LO>(poly, lane, x);
        wave_sync();
        load_pass<LOGN, 0>(poly, lane, x);
        fwd_stages<LOGN, 0, C::F2LO - 1, 0>(x, lane, tw, Q);
    }
    store_pass<LOGN, 0>(poly, lane, x);
    wave_sync();
}
// forward NTT (N = 2048) whose stages on bits 10, 9, 8 were already applied by the producer: bits 7..3, then 2..0
template <int LOGN>
__device__ __forceinline__ void ntt_forward_wave_low8(double* poly, Tw tw, u32 lane, double Q) {
    static_assert(LOGN == 11, "laid out for N = 2048");
    double x[Cfg<LOGN>::E];
    load_pass<LOGN, 3>(poly, lane, x);
    fwd_stages<LOGN, 3, 7, 3>(x, lane, tw, Q);
    store_pass<LOGN, 3>(poly, lane, x);
    wave_sync();
    load_pass<LOGN, 0>(poly, lane, x);
    fwd_stages<LOGN, 0, 2, 0>(x, lane, tw, Q);
    store_pass<LOGN, 0>(poly, lane, x);
    wave_sync();
}
// Forward phase of the 8-wave (N = 2048) workgroup: 6 polynomials whose stages on bits 10, 9, 8 are done.  One
// transform per wave would put two full transforms on two of the four SIMDs (waves w and w + 4 share a SIMD), so
// waves 0..3 take polynomials 0..3 (32 coefficients per lane: bits 7..3, then 2..0) and waves 4..7 take HALF of
// polynomial 4 or 5 each (wave 4 + 6: polynomial 4, wave 5 + 7: polynomial 5; 16 coefficients per lane: bits 7..4,
// then 3..0): every SIMD carries 1.5 transforms.  The two passes are separated by ONE workgroup barrier (the half
// transforms exchange data across two waves), which every wave executes.
template <int LOGN>
__device__ __forceinline__ void forward_phase_balanced(double* dct, int NP, Tw twa, u32 wave, u32 lane, double Q) {
    constexpr u32 TWL = 1024;
    static_assert(LOGN == 11, "laid out for N = 2048, 6 polynomials on 8 waves");
    if (wave < 4) {
        double* poly = dct + wave * NP;
        double x[Cfg<LOGN>::E];
        load_pass<LOGN, 3>(poly, lane, x);
        fwd_stages<LOGN, 3, 7, 3, TWL>(x, lane, twa, Q);
        store_pass<LOGN, 3>(poly, lane, x);
        block_sync_lds();
        load_pass<LOGN, 0>(poly, lane, x);
        fwd_stages<LOGN, 0, 2, 0, TWL>(x, lane, twa, Q);
        store_pass<LOGN, 0>(poly, lane, x);
    } else {
        double* poly = dct + (4 + (wave & 1u)) * NP;
        const u32 v = ((wave >> 1) & 1u) * 64u + lane;  // 128 virtual lanes per polynomial
        double x[16];
        {   // pass A: registers = position bits 7..4; v[3:0] = p[3:0], v[6:4] = p[10:8]
            const u32 hi = v >> 4, base = (hi << 8) | (v & 15u);
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = poly[phys(base | ((u32)r << 4))];
#pragma unroll
            for (int B = 7; B >= 4; --B) {           // stage on bit B = register bit B - 4; twiddle tw[m + (p >> (B+1))]
                const int rb = B - 4;
                const u32 m = 1u << (10 - B);
                const double2* tw = twa.l;  // m <= 64: LDS mirror
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r & (1 << rb)) continue;
                    const double2 w = tw[m + ((hi << (7 - B)) | (u32)(r >> (rb + 1)))];
                    const double X = x[r];
                    const double T = modmul_q(x[r | (1 << rb)], w.x, w.y, Q);
                    x[r] = X + T;
                    x[r | (1 << rb)] = X - T;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) poly[phys(base | ((u32)r << 4))] = x[r];
        }
        block_sync_lds();
        {   // pass B: registers = position bits 3..0; v = p[10:4]
            const u32 base = v << 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = poly[phys(base | (u32)r)];
#pragma unroll
            for (int B = 3; B >= 0; --B) {
                const u32 m = 1u << (10 - B);
                const double2* tw = (B == 0) ? twa.g : twa.l;  // only the block of the last stage (m = 1024) is global
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r & (1 << B)) continue;
                    const double2 w = tw[m + ((v << (3 - B)) | (u32)(r >> (B + 1)))];
                    const double X = x[r];
                    const double T = modmul_q(x[r | (1 << B)], w.x, w.y, Q);
                    x[r] = X + T;
                    x[r | (1 << B)] = X - T;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) poly[phys(base | (u32)r)] = x[r];
        }
    }
}

// One HALF of a forward transform (stages on bits 7..0; bits 10, 9, 8 were applied by the producer) by one wave:
// 128 virtual lanes per polynomial, v = half * 64 + lane, 16 coefficients per lane (bits 7..4, then 3..0), one
// workgroup barrier between the two passes (the two halves exchange data), executed by every wave of the workgroup.
template <int LOGN, bool W1 = false>
__device__ __forceinline__ void forward_half(double* poly, Tw twa, u32 v, double Q) {
    static_assert(LOGN == 11, "laid out for N = 2048");
    typedef TwSel<W1> TS;
    typedef typename TS::T TV;
    double x[16];
        {   // pass A: registers = position bits 7..4; v[3:0] = p[3:0], v[6:4] = p[10:8]
            const u32 hi = v >> 4, base = (hi << 8) | (v & 15u);
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = poly[phys(base | ((u32)r << 4))];
#pragma unroll
            for (int B = 7; B >= 4; --B) {           // stage on bit B = register bit B - 4; twiddle tw[m + (p >> (B+1))]
                const int rb = B - 4;
                const u32 m = 1u << (10 - B);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r & (1 << rb)) continue;
                    const TV w = TS::lds(twa, m + ((hi << (7 - B)) | (u32)(r >> (rb + 1))));   // m <= 64: LDS mirror
                    const double X = x[r];
                    const double T = tmul(x[r | (1 << rb)], w, twa, Q);
                    x[r] = X + T;
                    x[r | (1 << rb)] = X - T;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) poly[phys(base | ((u32)r << 4))] = x[r];
        }
        block_sync_lds();
        {   // pass B: registers = position bits 3..0; v = p[10:4]
            const u32 base = v << 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = poly[phys(base | (u32)r)];
#pragma unroll
            for (int B = 3; B >= 0; --B) {
                const u32 m = 1u << (10 - B);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r & (1 << B)) continue;
                    const u32 ti = m + ((v << (3 - B)) | (u32)(r >> (B + 1)));
                    const TV w = (B == 0) ? TS::last(twa, ti) : TS::lds(twa, ti);  // only the block of the last stage (m = 1024) is global (W1: LDS)
                    const double X = x[r];
                    const double T = tmul(x[r | (1 << B)], w, twa, Q);
                    x[r] = X + T;
                    x[r | (1 << B)] = X - T;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) poly[phys(base | (u32)r)] = x[r];
        }
}
// One QUARTER of a forward transform (N = 2048; stages on bits 7..0, bits 10, 9, 8 were applied by the producer) by one
// wave: positions [512 k, 512 k + 512) are two of the eight independent 256-point sub-transforms of the row, so the
// wave needs nobody else's data -- 64 lanes x 8 coefficients, passes on bits 7..5 / 4..2 / 1..0, two WAVE-LOCAL
// re-shuffles, no workgroup barrier.  The folded 16-wave kernel runs its four digit rows as 16 of these, one per wave
// (four per SIMD); the folded 8-wave kernel runs two per wave.
template <typename TV>
__device__ __forceinline__ void fwd_bfly_d(double& X, double& Y, TV w, const Tw& twa, double Q) {
    const double T = tmul(Y, w, twa, Q);
    Y = X - T;
    X = X + T;
}
template <int LOGN, bool W1 = false>
__device__ __forceinline__ void forward_quarter(double* row, u32 k, Tw twa, u32 lane, double Q) {
    static_assert(LOGN == 11, "laid out for N = 2048");
    // blocks m <= 512 sit in the LDS mirror; the block of the last stage (m = 1024) in global memory, or (W1) in LDS too
    typedef TwSel<W1> TS;
    typedef typename TS::T TV;
    double x[8];
    {   // coefficients = bits 7..5, lane = (bit 8, bits 4..0)
        const u32 l8 = lane >> 5, hi = (k << 1) | l8;
        double* const p = row + phys((k << 9) | (l8 << 8) | (lane & 31u));   // coefficient r: + 32 r
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = p[(r >> 1) * 68 + (r & 1) * 32];
        const TV w7 = TS::lds(twa, 8 + hi);                                       // stage on bit B: tw[m + (p >> (B + 1))], m = 2^(10 - B)
#pragma unroll
        for (int r = 0; r < 4; ++r) fwd_bfly_d(x[r], x[r + 4], w7, twa, Q);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const TV w6 = TS::lds(twa, 16 + ((hi << 1) | g));
            fwd_bfly_d(x[4 * g], x[4 * g + 2], w6, twa, Q);
            fwd_bfly_d(x[4 * g + 1], x[4 * g + 3], w6, twa, Q);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) fwd_bfly_d(x[2 * g], x[2 * g + 1], TS::lds(twa, 32 + ((hi << 2) | g)), twa, Q);
#pragma unroll
        for (int r = 0; r < 8; ++r) p[(r >> 1) * 68 + (r & 1) * 32] = x[r];
    }
    wave_sync();
    {   // coefficients = bits 4..2, lane = (bits 8..5, bits 1..0)
        const u32 lh = lane >> 2, hi = (k << 4) | lh;
        double* const p = row + phys((k << 9) | (lh << 5) | (lane & 3u));       // coefficient r: + 4 r, inside one 64-block
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = p[4 * r];
        const TV w4 = TS::lds(twa, 64 + hi);
#pragma unroll
        for (int r = 0; r < 4; ++r) fwd_bfly_d(x[r], x[r + 4], w4, twa, Q);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const TV w3 = TS::lds(twa, 128 + ((hi << 1) | g));
            fwd_bfly_d(x[4 * g], x[4 * g + 2], w3, twa, Q);
            fwd_bfly_d(x[4 * g + 1], x[4 * g + 3], w3, twa, Q);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) fwd_bfly_d(x[2 * g], x[2 * g + 1], TS::lds(twa, 256 + ((hi << 2) | g)), twa, Q);
#pragma unroll
        for (int r = 0; r < 8; ++r) p[4 * r] = x[r];
    }
    wave_sync();
    {   // coefficients = bits 2..0 (bit 2 is done): 8 consecutive words per lane
        const u32 hl = (k << 6) | lane;
        double2* const p = reinterpret_cast<double2*>(row + phys(hl << 3));
        const double2 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
        x[0] = v0.x; x[1] = v0.y; x[2] = v1.x; x[3] = v1.y; x[4] = v2.x; x[5] = v2.y; x[6] = v3.x; x[7] = v3.y;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const TV w1 = TS::lds(twa, 512 + ((hl << 1) | g));
            fwd_bfly_d(x[4 * g], x[4 * g + 2], w1, twa, Q);
            fwd_bfly_d(x[4 * g + 1], x[4 * g + 3], w1, twa, Q);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) fwd_bfly_d(x[2 * g], x[2 * g + 1], TS::last(twa, 1024 + ((hl << 2) | g)), twa, Q);
        p[0] = make_double2(x[0], x[1]); p[1] = make_double2(x[2], x[3]); p[2] = make_double2(x[4], x[5]); p[3] = make_double2(x[6], x[7]);
    }
    wave_sync();
}

// Forward phase of the 16-wave (N = 2048) workgroup: the 6 digit polynomials as 12 half-transforms on waves 0..11
// (three per SIMD); waves 12..15 only take part in the barrier.
template <int LOGN>
__device__ __forceinline__ void forward_phase_halves(double* dct, int NP, Tw twa, u32 wave, u32 lane, double Q) {
    if (wave < 12) forward_half<LOGN>(dct + (wave % 6u) * NP, twa, (wave / 6u) * 64u + lane, Q);
    else block_sync_lds();
}

// inverse NTT by one wave; |src| <= 0.6 Q bit-reversed; coefficient j = (r << 6) | lane in x[r], |x| <= 0.57 Q
template <int LOGN>
__device__ __forceinline__ void ntt_inverse_wave(const double* src, double* tmp, Tw tw, u32 lane, double Q,
                                                 double2 ninv, double (&x)[Cfg<LOGN>::E]) {
    using C = Cfg<LOGN>;
    constexpr int LE = C::LE;
    load_pass<LOGN, 0>(src, lane, x);
    inv_stages<LOGN, 0, 0, LE - 1>(x, lane, tw, Q);
    store_pass<LOGN, 0>(tmp, lane, x);
    wave_sync();
    load_pass<LOGN, LE>(tmp, lane, x);
    inv_stages<LOGN, LE, LE, 2 * LE - 1>(x, lane, tw, Q);
    store_pass<LOGN, LE>(tmp, lane, x);
    wave_sync();
    load_pass<LOGN, 6>(tmp, lane, x);
    inv_stages<LOGN, 6, 2 * LE, LOGN - 1>(x, lane, tw, Q);
#pragma unroll
    for (int r = 0; r < C::E; ++r) x[r] = modmul_q(x[r], ninv.x, ninv.y, Q);
}
template <int LOGN>
__device__ __forceinline__ double psi_pow(const double2* tw, u32 e, double Q) {
    constexpr u32 N = 1u << LOGN;
    const double v = tw[__brev(e & (N - 1)) >> (32 - LOGN)].x;
    return (e & N) ? Q - v : v;
}
__device__ __forceinline__ double2 key_row(__amdgpu_buffer_rsrc_t rsrc, u32 voff, u32 soff) {
    const ulonglong2 v = w64::key_row(rsrc, voff, soff);
    return make_double2(__longlong_as_double((long long)v.x), __longlong_as_double((long long)v.y));
}

// ---- split inverse transform for N = 2048 (doubles): 4 waves x 64 lanes x 8 coefficients per polynomial ----------
// Three-stage passes on position bits (0,1,2), (3,4,5), (6,7,8) and a two-stage pass on (9,10); between passes the
// values go through LDS in REGISTER-MAJOR layouts (register r of thread t at r * stride + t: consecutive lanes,
// conflict-free stores) whose row strides make the next pass's loads conflict-free too:
//   e0: 257 (reader lanes vary p[2:0] and p[8:6]),  e1: 264 (reader lanes vary p[5:0]),  e2: 257 (any stride is
//   conflict-free for it; 257 makes a wave's e2 words the same set as its e0 words).
// The exchanges after passes 0 and 1 swap register bits with LANE bits only (pass 1 reads what its own wave's pass 0
// wrote, pass 2 what its own wave's pass 1 wrote) and every wave's words in a buffer are the same set in every layout:
// they need no workgroup barrier.  Only the exchange before the last pass (position bits 10:9 <-> wave) crosses waves.
// No reduction anywhere (see the bounds above); the caller multiplies by N^-1.
// three Gentleman-Sande stages on the register index bits 0,1,2; twiddles: f0[4] (bit 0), f1[2] (bit 1), f2 (bit 2)
template <typename TV>
__device__ __forceinline__ void inv_pass8(double (&x)[8], const TV (&f0)[4], const TV (&f1)[2], TV f2, const Tw& twa, double Q) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // pairs (2k, 2k+1): index bits above bit 0 = k
        const double X = x[2 * k], Y = x[2 * k + 1];
        x[2 * k] = X + Y;
        x[2 * k + 1] = tmul(Y - X, f0[k], twa, Q);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // pairs (r, r+2), r = (k>>1)*4 + (k&1): twiddle by r >> 2
        const int r = (k >> 1) * 4 + (k & 1);
        const double X = x[r], Y = x[r + 2];
        x[r] = X + Y;
        x[r + 2] = tmul(Y - X, f1[k >> 1], twa, Q);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // pairs (r, r+4)
        const double X = x[r], Y = x[r + 4];
        x[r] = X + Y;
        x[r + 4] = tmul(Y - X, f2, twa, Q);
    }
}
// inverse twiddle of stage B for stage index i: -tw[(2m - 1) - i], m = 2^(LOGN-1-B); the sign sits in the operand
template <int B, bool W1 = false>
__device__ __forceinline__ typename TwSel<W1>::T itw11(Tw tw, u32 i) {
    constexpr u32 m = 1u << (10 - B);
    if constexpr (W1) return tw.l1[(2 * m - 1) - i];
    else return tw.blk<m, 1024>()[(2 * m - 1) - i];
}
// acc: evaluation-form polynomial (padded natural layout); bufA / bufB: >= 2112 doubles each; t = thread in the
// 256-thread group.  Leaves coefficient j = (r << 8) | t in x[r] (before the N^-1 scaling).  INV_BARRIERS workgroup
// barriers; bufA must not be written by anyone until the caller's next barrier (pass 3 reads it).
template <bool W1 = false>
__device__ __forceinline__ void split_inverse11(const double* src, double* bufA, double* bufB, Tw tw, u32 t, double Q,
                                                double (&x)[8]) {
    typedef typename TwSel<W1>::T TV;
    TV f0[4], f1[2], f2;
    {   // pass 0: p = 8t + r
        const double2* sp = reinterpret_cast<const double2*>(src + phys(8 * t));
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double2 v = sp[k]; x[2 * k] = v.x; x[2 * k + 1] = v.y; }
#pragma unroll
        for (int k = 0; k < 4; ++k) f0[k] = itw11<0, W1>(tw, 4 * t + k);
        f1[0] = itw11<1, W1>(tw, 2 * t); f1[1] = itw11<1, W1>(tw, 2 * t + 1);
        f2 = itw11<2, W1>(tw, t);
        inv_pass8(x, f0, f1, f2, tw, Q);
#pragma unroll
        for (int r = 0; r < 8; ++r) bufA[r * 257 + t] = x[r];          // e0
    }
    wave_local_sync();
    {   // pass 1: p = (g << 6) | (r << 3) | l,  g = t >> 3, l = t & 7; e0 address of p: (p & 7) * 257 + (p >> 3)
        const u32 g = t >> 3, l = t & 7u;
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = bufA[l * 257 + ((g << 3) | (u32)r)];
#pragma unroll
        for (int k = 0; k < 4; ++k) f0[k] = itw11<3, W1>(tw, 4 * g + k);
        f1[0] = itw11<4, W1>(tw, 2 * g); f1[1] = itw11<4, W1>(tw, 2 * g + 1);
        f2 = itw11<5, W1>(tw, g);
        inv_pass8(x, f0, f1, f2, tw, Q);
#pragma unroll
        for (int r = 0; r < 8; ++r) bufB[r * 264 + t] = x[r];          // e1 (t = (g << 3) | l)
    }
    wave_local_sync();
    {   // pass 2: p = (h << 9) | (r << 6) | m,  h = t >> 6, m = t & 63; e1 address of p: p[5:3] * 264 + ((p >> 6) << 3 | p[2:0])
        const u32 h = t >> 6, m = t & 63u;
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = bufB[(m >> 3) * 264 + ((((h << 3) | (u32)r) << 3) | (m & 7u))];
#pragma unroll
        for (int k = 0; k < 4; ++k) f0[k] = itw11<6, W1>(tw, 4 * h + k);
        f1[0] = itw11<7, W1>(tw, 2 * h); f1[1] = itw11<7, W1>(tw, 2 * h + 1);
        f2 = itw11<8, W1>(tw, h);
        inv_pass8(x, f0, f1, f2, tw, Q);
#pragma unroll
        for (int r = 0; r < 8; ++r) bufA[r * 257 + t] = x[r];          // e2 (t = (h << 6) | m)
    }
    block_sync_lds();
    {   // pass 3: p = (r << 8) | t; e2 address of p: p[8:6] * 257 + (p[10:9] << 6 | p[5:0]); stages on bits 9, 10
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = bufA[((((u32)r & 1u) << 2) | (t >> 6)) * 257 + ((((u32)r >> 1) << 6) | (t & 63u))];
        const TV g0 = itw11<9, W1>(tw, 0), g1 = itw11<9, W1>(tw, 1), g2 = itw11<10, W1>(tw, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // bit 9 = register bit 1: pairs (r, r+2), r in {0,1,4,5}; stage index = r >> 2
            const int r = (k >> 1) * 4 + (k & 1);
            const double X = x[r], Y = x[r + 2];
            x[r] = X + Y;
            const TV f = (k >> 1) ? g1 : g0;
            x[r + 2] = tmul(Y - X, f, tw, Q);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // bit 10 = register bit 2
            const double X = x[r], Y = x[r + 4];
            x[r] = X + Y;
            x[r + 4] = tmul(Y - X, g2, tw, Q);
        }
    }
}

// SPLIT (N = 2048 only): 512-thread workgroup, inverse transforms on all 8 waves (split_inverse11), forward
// transforms on waves 0..R-1, 4 MAC items per thread.
// W16 (with SPLIT): 1024-thread workgroup = 4 waves per SIMD on the one workgroup a CU's LDS holds.  The inverse
// transforms stay on waves 0..7 (8 coefficients per thread is what keeps them at 4 passes); the forward phase runs as
// 12 half-transforms on waves 0..11 and the MAC as 2 items per thread on all 16 waves, so that the LDS traffic of one
// wave (twiddles + coefficients: ~6 k cycles of the phase, profiles/r02_phase_std192_ap.log) overlaps with the
// arithmetic of three others instead of one.  128-register budget: key rows one (GINX) / two (AP) items deep.
// key-row pipeline depth of the 16-wave build (development knobs, tools/w16_sweep.sh).  Measured on one MI355X, STD192,
// 256 bootstraps per launch (profiles/r02_w16_sweep.log): 8 waves 28.1 ms (AP) / 17.0 ms (GINX); 16 waves with one item
// in flight and nothing requested before the transforms 25.9 / 18.5 ms; every deeper pipeline spills 33..81 registers of
// the 128 a 1024-thread workgroup leaves each thread and is slower than the 8-wave build.
// BCE_W16_EARLY (16-wave AP build): both MAC items' key rows are requested BEFORE the forward phase, where the registers
// are free -- by waves 8..15 at the top of the step (they have no share of the inverse transforms), by waves 0..7 once
// their digits are written -- and held through the quarter-transforms (48 registers), so that the MAC phase finds them
// landed instead of exposing the row latency twice per step.  (Requests in front of the inverse transforms, the
// NPRE knobs below, put 24..48 live registers into the phase that needs all 128: 33..81 spills.)
#ifndef BCE_W16_EARLY
#define BCE_W16_EARLY 1
#endif
#ifndef BCE_W16_NBUF_AP
#define BCE_W16_NBUF_AP 1
#define BCE_W16_NPRE_AP 0
#define BCE_W16_NBUF_GINX 1
#define BCE_W16_NPRE_GINX 0
#endif
// FOLD (with SPLIT): the lowest gadget digit is never transformed -- the key arrives with rows l >= 1 replaced by
// ek_l - B^l ek_0 and the MAC multiplies the digit-0 rows by the evaluation-form accumulator itself (see the FOLD note of
// k_blind_rotate_lat in kernels.hip; SignedDigitDecompose is exact for these parameters, checked by the host).  Four
// forward transforms per step instead of six = eight half-transforms, one per wave (8-wave build), or sixteen quarter-
// transforms, one per wave (16-wave build).  The accumulator is double-buffered between `acc` and digit rows 0, 1; the inverse transform's exchange buffers
// move to digit rows 2..5.
// FUSE (16-wave build): the tail of EvalBinGate runs in this kernel's epilogue (fused_tail.hpp) -- the coefficient-form
// accumulator goes to LDS as u64 words, all 1,024 threads extract, switch the modulus, gather the N dKS key-switching rows
// (33.6 MB for STD192) and write the refreshed ciphertext to the pool; no tail kernels, no accumulator round trip through HBM.
template <int LOGN, int DG, bool AP, bool SPLIT = false, bool W16 = false, bool FOLD = false, bool FUSE = false,
          u32 NBUF_ = (W16 ? (AP ? BCE_W16_NBUF_AP : BCE_W16_NBUF_GINX) : (AP ? 3 : 2)),
          u32 NPRE_ = (W16 ? (AP ? BCE_W16_NPRE_AP : BCE_W16_NPRE_GINX) : (AP ? 2 : 1)),
          bool PERSIST = false, typename PT = DevParams>
__device__ __forceinline__ void bootstrap64d(const PT& P, const bce_gate_desc g, const u32 soff, const u32 boot, double* smemd,
                                             u64* __restrict__ acc_out, u32* __restrict__ dbg_lweN, u32* __restrict__ dbg_ks) {
    // one gate bootstrap by one workgroup; `boot` = index of the bootstrap in the launch (acc_out / dbg rows).
    // PERSIST (dataflow kernel, k_bootstrap_dag64): called from a loop -- the thread index is opaque per call so that the
    // compiler cannot split the caller's loop on it, and nothing leaves through acc_out
    static_assert(!FUSE || (W16 && SPLIT), "fused tail: 16-wave build");
    static_assert(!PERSIST || FUSE, "persistent callers rely on the fused tail");
    using C = Cfg<LOGN>;
    constexpr int N = C::N, NP = C::NP, E = C::E;
    constexpr u32 R = 2 * DG, T = W16 ? 1024 : (SPLIT ? 512 : 64 * R);
    static_assert(!SPLIT || (LOGN == 11 && R <= 8 && R >= 4), "split inverse transform: N = 2048");
    static_assert(!W16 || (SPLIT && R == 6), "16-wave variant: N = 2048, three gadget digits");
    static_assert(!FOLD || (SPLIT && R == 6), "folded gadget digit: N = 2048, three gadget digits");
    u32 tid_o = threadIdx.x;
    if constexpr (PERSIST) asm volatile("" : "+v"(tid_o));
    double* acc = smemd;          // [2][NP] evaluation form, |value| <= 0.6 Q
    double* dct = acc + 2 * NP;   // [R][NP]
    u32* av = reinterpret_cast<u32*>(dct + R * NP);
    const double2* __restrict__ tw = P.tw64d;
    // LDS mirror of the first 1024 twiddle entries (8-wave kernel only: 16 KiB of the 20 KiB left next to the polynomials)
    double2* twl = reinterpret_cast<double2*>(av + ((P.n + 1 + 3) & ~3u));
    // W1 (folded key): all N twiddles in those 16 KiB as plain doubles (see Tw); -DBCE_TWQ keeps the (w, w / Q) pairs (A/B runs)
#ifdef BCE_TWQ
    constexpr bool W1 = false;
#else
    constexpr bool W1 = FOLD && SPLIT;
#endif
    typedef TwSel<W1> TS;
    typedef typename TS::T TV;
    double* twl1 = reinterpret_cast<double*>(twl);
    if constexpr (W1) {
        for (u32 i = tid_o; i < (u32)N; i += T) twl1[i] = tw[i].x;
    } else if constexpr (SPLIT) {
        for (u32 i = tid_o; i < 1024u; i += T) twl[i] = tw[i];
    }
    const u32 tid = tid_o, lane = tid & 63, wave = tid >> 6;
    const double Q = P.Qd, invQ = P.invQd;
    const Tw twa{tw, twl, twl1, invQ};

    const u32 q = P.q, qm = q - 1, n = P.n;
    {
        const u32* in0 = P.pool + (size_t)(g.in0 + soff) * P.pool_stride;
        const u32* in1 = P.pool + (size_t)(g.in1 + soff) * P.pool_stride;
        const bool two = g.op <= BCE_XNOR_FAST;
        for (u32 i = tid; i <= n; i += T) {
            u32 v0 = in0[i];
            if (g.neg0) v0 = ((i == n ? (q >> 2) : 0u) - v0) & qm;
            u32 v = v0;
            if (two) {
                u32 v1 = in1[i];
                if (g.neg1) v1 = ((i == n ? (q >> 2) : 0u) - v1) & qm;
                v = (g.op == BCE_XOR_FAST || g.op == BCE_XNOR_FAST) ? (2u * (v0 - v1)) & qm : (v0 + v1) & qm;
            } else if (i == n) {
                v = (v0 + (q >> 2)) & qm;
            }
            av[i] = v;
        }
    }
    __syncthreads();
    {
        const u32 b = av[n];
        const u32 q1 = gate_const(g.op, q), q2 = (q1 + (q >> 1)) & qm;
        const double pos = (double)P.Q8p1_64, neg = -pos;
        for (u32 j = tid; j < (u32)N; j += T) {
            double v = 0.0;
            if (j % P.factor == 0) {
                u32 t = (b - j / P.factor) & qm;
                bool in = (q1 < q2) ? (t >= q1 && t < q2) : !(t >= q2 && t < q1);
                v = in ? neg : pos;
            }
            acc[phys(j)] = 0.0;
            acc[NP + phys(j)] = v;
        }
    }
    __syncthreads();
    if (wave == 0) ntt_forward_wave<LOGN>(acc + NP, Tw{tw, nullptr}, lane, Q);
    __syncthreads();
    const double2 ninv = make_double2(P.Ninvd, P.Ninvd_q);
    // KN (folded key): the evaluation-form accumulator lives scaled by N^-1 -- rows l >= 1 of the key carry the factor, rows 0, 1
    // multiply the (scaled) accumulator itself -- so the un-normalised inverse transform of a step returns the coefficients
    constexpr bool KN = FOLD && (BCE_KEY_NINV != 0);
    for (u32 j = tid; j < (u32)N; j += T)
        acc[NP + phys(j)] = KN ? modmul_q(acc[NP + phys(j)], ninv.x, ninv.y, Q) : modred(acc[NP + phys(j)], invQ, Q);
    __syncthreads();

    constexpr size_t rgsw = (size_t)R * 2 * N;
    const double* __restrict__ bsk = reinterpret_cast<const double*>(P.bsk64);
    // SignedDigitDecompose: representative of the reference (canonical x, `x < Q/2 ? x : x - Q`), closed-form digits
    const double dlo = -(double)((P.Q64 + 1) / 2), dhi = (double)((P.Q64 - 3) / 2);
    const double Bd = (double)(1u << P.gBits), invB = 1.0 / Bd, halfB = 0.5 * Bd;
    double doff = 0.0;
    {
        double pw = halfB;
        for (u32 l = 0; l < (u32)DG; ++l) { doff += pw; pw *= Bd; }
    }
    const u32 nsteps = AP ? n * P.dR : n;
    // FOLD: the step reads the evaluation-form accumulator at `cur` (also as MAC rows 0, 1) and writes the new one to `nxt`
    double* cur = acc;
    double* nxt = FOLD ? dct : acc;
    // exchange buffers of the split inverse transform: digit rows that are dead until the digits are written
    constexpr int XA = FOLD ? 2 : 0, XB = FOLD ? 4 : 2;
    BCE_PROF_INIT();
    for (u32 step = 0; step < nsteps; ++step) {
        u32 ap = 0;
        const double* bk;
        if constexpr (!AP) {
            ap = ((q - av[step]) & qm) * P.factor;
            if (ap == 0) continue;
            bk = bsk + (size_t)step * 2 * rgsw;
        } else {
            const u32 i = step / P.dR, k = step - i * P.dR;
            u32 aI = (q - av[i]) & qm;
            for (u32 t = 0; t < k; ++t) aI /= P.baseR;
            const u32 a0 = aI % P.baseR;
            if (a0 == 0) continue;
            bk = bsk + (((size_t)i * P.baseR + a0) * P.dR + k) * rgsw;
        }
        // key rows: same software pipeline as the integer kernel
        constexpr u32 ITEMS = (2u * (N / 2) + T - 1) / T;
        constexpr u32 ROWS = AP ? R : 2 * R;
        constexpr bool EARLY = W16 && AP && FOLD && (BCE_W16_EARLY != 0);   // (the plain-key build has six rows to transform: no room)
        constexpr u32 NBUF = EARLY ? ITEMS : NBUF_, NPRE = EARLY ? ITEMS : NPRE_;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double*>(bk), 0, (int)((AP ? 1 : 2) * rgsw * sizeof(double)), 0x00020000);
        double2 kb[NBUF][ROWS];
        u32 tid_v = tid, lane_v = lane;
        asm volatile("" : "+v"(tid_v), "+v"(lane_v));  // keep the address arithmetic inside the step loop
        auto request = [&](auto kc) {
            constexpr u32 k = decltype(kc)::value;
            if constexpr (k < ITEMS) {
                const u32 item = tid_v + k * T;
                if (k + 1 < ITEMS || item < 2u * (N / 2)) {
                    const u32 c = item / (N / 2), p0 = (item % (N / 2)) * 2;
                    const u32 voff = (c * N + p0) * 8u;
#pragma unroll
                    for (u32 l = 0; l < R; ++l) {
                        kb[k % NBUF][l] = key_row(rsrc, voff, l * (2 * N * 8));
                        if constexpr (!AP) kb[k % NBUF][R + l] = key_row(rsrc, voff, (u32)(rgsw * 8) + l * (2 * N * 8));
                    }
                }
            }
        };
        if constexpr (!EARLY && NPRE >= 1) request(std::integral_constant<u32, 0>{});
        if constexpr (!EARLY && NPRE >= 2) request(std::integral_constant<u32, 1>{});
        // the first three FORWARD stages (bits 10, 9, 8 = the register index) on one digit polynomial's eight coefficients of a thread
        auto three_stages = [&](double (&v)[8]) {
            const TV w10 = TS::lds(twa, 1), w9a = TS::lds(twa, 2), w9b = TS::lds(twa, 3);
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // bit 10: (r, r+4), twiddle tw[1]
                const double T = tmul_digit(v[r + 4], w10, twa, Q);   // |digit| <= B / 2 = 2^12 (gBits = 13: checked by the host)
                v[r + 4] = v[r] - T; v[r] = v[r] + T;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // bit 9: (r, r+2), r in {0,1,4,5}, twiddle tw[2 + (r >> 2)]
                const int r = (k >> 1) * 4 + (k & 1);
                const TV w = (k >> 1) ? w9b : w9a;
                const double T = tmul(v[r + 2], w, twa, Q);
                v[r + 2] = v[r] - T; v[r] = v[r] + T;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // bit 8: (2k, 2k+1), twiddle tw[4 + k]
                const TV w = TS::lds(twa, 4 + k);
                const double T = tmul(v[2 * k + 1], w, twa, Q);
                v[2 * k + 1] = v[2 * k] - T; v[2 * k] = v[2 * k] + T;
            }
        };
        if (W16 && tid_v >= 512u) {
            // waves 8..15 have no share of the inverse transforms: they keep the barrier count (and, EARLY, pull their key rows
            // while their registers are free)
            if constexpr (EARLY) for_each_index(request, std::make_integer_sequence<u32, ITEMS>{});
#pragma unroll
            for (int b = 0; b < INV_BARRIERS + 1; ++b) block_sync_lds();
            // (round 4, rejected: handing the top digit's raw coefficients to these waves through its digit rows so that they
            // run its three in-register stages -- one more barrier, 12 of 24 butterflies off waves 0..7 -- is 2.3 % SLOWER per
            // 1,024-bootstrap launch, profiles/r04_cfg5_offload_ab.log)
        } else if constexpr (SPLIT) {
            const u32 c = wave >> 2, t = tid_v & 255u;
            double x[8];
            // exchange buffers live in dct rows 0..3 (dead until the digits are written)
            split_inverse11<W1>(cur + c * NP, dct + (XA + c) * NP, dct + (XB + c) * NP, twa, t, Q, x);
            block_sync_lds();  // every thread has read its pass-3 inputs: the digit rows may be overwritten
            // digits, then the first three FORWARD stages (bits 10, 9, 8 = this thread's register index) on each digit
            // in registers: the forward transform below is left with bits 7..0 (two passes)
            double u[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                double d = KN ? modred(x[r], invQ, Q) : modmul_q(x[r], ninv.x, ninv.y, Q);
                d = d > dhi ? d - Q : d;
                d = d < dlo ? d + Q : d;
                u[r] = d + doff;
            }
            auto next_digit = [&](double (&v)[8]) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const double fl = floor(u[r] * invB);
                    v[r] = fma(-fl, Bd, u[r]) - halfB;  // digit in [-B/2, B/2)
                    u[r] = fl;
                }
            };
#pragma unroll
            for (u32 l = 0; l < (u32)DG; ++l) {
                double v[8];
                next_digit(v);
                if (FOLD && l == 0) continue;   // digit 0 only advances the running quotient
                three_stages(v);
#pragma unroll
                for (int r = 0; r < 8; ++r) dct[(2 * l + c) * NP + phys(((u32)r << 8) | t)] = v[r];
            }
            if constexpr (EARLY) for_each_index(request, std::make_integer_sequence<u32, ITEMS>{});
        } else if (wave < 2) {
            double x[E];
            ntt_inverse_wave<LOGN>(acc + wave * NP, dct + wave * NP, Tw{tw, nullptr}, lane_v, Q, ninv, x);
#pragma unroll
            for (int r = 0; r < E; ++r) {
                double d = x[r];
                d = d > dhi ? d - Q : d;
                d = d < dlo ? d + Q : d;
                double u = d + doff;
                const u32 pj = phys(((u32)r << 6) | lane_v);
#pragma unroll
                for (u32 l = 0; l < (u32)DG; ++l) {
                    const double t = floor(u * invB);
                    dct[(2 * l + wave) * NP + pj] = fma(-t, Bd, u) - halfB;  // digit in [-B/2, B/2)
                    u = t;
                }
            }
        }
        BCE_PROF_MARK(0);
        block_sync_lds();
        BCE_PROF_MARK(1);
        if constexpr (FOLD && W16) {
            // rows 2..5 as sixteen quarter-transforms, one per wave (four per SIMD), no barrier inside the phase (eight
            // half-transforms on waves 0..7 with waves 8..15 idle: 21.0 vs 20.2 ms per launch, profiles/r02_fold_quarters_ab.log)
            forward_quarter<LOGN, W1>(dct + (2 + (wave & 3u)) * NP, wave >> 2, twa, lane_v, Q);
        } else if constexpr (FOLD) {
            // eight waves: rows 2..5 as eight half-transforms, waves w and w + 4 share a row (and a SIMD); two quarter-
            // transforms per wave instead measure the same
            forward_half<LOGN, W1>(dct + (2 + (wave & 3u)) * NP, twa, (wave >> 2) * 64u + lane_v, Q);
        } else if constexpr (W16) {
            forward_phase_halves<LOGN>(dct, NP, twa, wave, lane_v, Q);
        } else if constexpr (SPLIT) {
            forward_phase_balanced<LOGN>(dct, NP, twa, wave, lane_v, Q);
        } else {
            ntt_forward_wave<LOGN>(dct + wave * NP, Tw{tw, nullptr}, lane_v, Q);
        }
        BCE_PROF_MARK(2);
        block_sync_lds();
        BCE_PROF_MARK(3);
        if constexpr (NPRE < 1) request(std::integral_constant<u32, 0>{});
        if constexpr (NPRE < 2 && NBUF >= 2) request(std::integral_constant<u32, 1>{});
        if constexpr (NBUF >= 3) request(std::integral_constant<u32, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        const bool odd = ap & 1u;
        auto mac_item = [&](auto kc) {
            constexpr u32 k = decltype(kc)::value;
            const u32 item = tid_v + k * T;
            if (k + 1 < ITEMS || item < 2u * (N / 2)) {
                const u32 c = item / (N / 2), p0 = (item % (N / 2)) * 2;
                const u32 pp = phys(p0);
                double sp[2] = {0.0, 0.0}, sn[2] = {0.0, 0.0};
#pragma unroll
                for (u32 l = 0; l < R; ++l) {
                    // FOLD: rows 0, 1 are the accumulator components themselves
                    const double2 d = *reinterpret_cast<const double2*>((FOLD && l < 2 ? cur : dct) + l * NP + pp);
                    const double2 kp = kb[k % NBUF][l];
                    sp[0] += modmul(d.x, kp.x, invQ, Q);
                    sp[1] += modmul(d.y, kp.y, invQ, Q);
                    if constexpr (!AP) {
                        const double2 kn = kb[k % NBUF][R + l];
                        sn[0] += modmul(d.x, kn.x, invQ, Q);
                        sn[1] += modmul(d.y, kn.y, invQ, Q);
                    }
                    // 128-register build: keep the rows in program order (interleaving all of them for instruction-level
                    // parallelism needs ~70 temporaries; four waves per SIMD provide the parallelism instead)
                    if constexpr (W16) __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
                request(std::integral_constant<u32, k + NBUF>{});
                __builtin_amdgcn_sched_barrier(0);
                double a[2];
                if constexpr (AP) {
                    a[0] = modred(sp[0], invQ, Q);
                    a[1] = modred(sp[1], invQ, Q);
                } else {
                    const u32 k0 = __brev(p0) >> (32 - LOGN);
                    const u32 ex = ((2 * k0 + 1) * ap) & (2 * N - 1);
                    double mp[2], mn[2];
                    mp[0] = psi_pow<LOGN>(tw, ex, Q);
                    mn[0] = psi_pow<LOGN>(tw, (2 * N - ex) & (2 * N - 1), Q);
                    mp[1] = odd ? Q - mp[0] : mp[0];
                    mn[1] = odd ? Q - mn[0] : mn[0];
                    const double2 old = *reinterpret_cast<const double2*>(cur + c * NP + pp);
                    const double oldv[2] = {old.x, old.y};
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const double rp = modred(sp[e], invQ, Q), rn = modred(sn[e], invQ, Q);
                        const double t = modmul(rp, mp[e] - 1.0, invQ, Q) + modmul(rn, mn[e] - 1.0, invQ, Q) + oldv[e];
                        a[e] = modred(t, invQ, Q);
                    }
                }
                *reinterpret_cast<double2*>(nxt + c * NP + pp) = make_double2(a[0], a[1]);
            }
        };
        for_each_index(mac_item, std::make_integer_sequence<u32, ITEMS>{});
        BCE_PROF_MARK(4);
        block_sync_lds();
        BCE_PROF_MARK(5);
        if constexpr (FOLD) { double* const t = cur; cur = nxt; nxt = t; }
    }
    BCE_PROF_FLUSH();
    if (W16 && tid >= 512u) {
#pragma unroll
        for (int b = 0; b < INV_BARRIERS; ++b) block_sync_lds();
    } else if constexpr (SPLIT) {
        const u32 c = wave >> 2, t = tid & 255u;
        double x[8];
        split_inverse11<W1>(cur + c * NP, dct + (XA + c) * NP, dct + (XB + c) * NP, twa, t, Q, x);
        u64* out = acc_out + ((size_t)boot * 2 + c) * N;
        u64* coef = reinterpret_cast<u64*>(acc);   // FUSE: [2][N] u64 in the accumulator's own rows (dead: `cur` was read by pass 0)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const double y = KN ? modred(x[r], invQ, Q) : modmul_q(x[r], ninv.x, ninv.y, Q);
            const double v = y < 0.0 ? y + Q : y;
            const double hi = floor(v * (1.0 / 4294967296.0));
            const u64 wv = ((u64)(u32)hi << 32) | (u64)(u32)fma(-hi, 4294967296.0, v);
            if constexpr (!PERSIST) out[((u32)r << 8) | t] = wv;
            if constexpr (FUSE) coef[c * N + (((u32)r << 8) | t)] = wv;
        }
    } else if (wave < 2) {
        double x[E];
        ntt_inverse_wave<LOGN>(acc + wave * NP, dct + wave * NP, Tw{tw, nullptr}, lane, Q, ninv, x);
        u64* out = acc_out + ((size_t)boot * 2 + wave) * N;
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const double v = x[r] < 0.0 ? x[r] + Q : x[r];   // [0, Q), an integer below 2^39
            const double hi = floor(v * (1.0 / 4294967296.0));
            out[((u32)r << 6) | lane] = ((u64)(u32)hi << 32) | (u64)(u32)fma(-hi, 4294967296.0, v);
        }
    }
    if constexpr (FUSE) {
        // FOLD: if the last step left the evaluation-form accumulator in digit rows 0, 1 (`cur` == dct), `acc` was free all
        // along; otherwise its last reader was pass 0 of the inverse transform above, three barriers ago.
        __syncthreads();
        u32* rowidx = reinterpret_cast<u32*>(dct);                                      // digit rows + exchange buffers: all dead
        u64* red = reinterpret_cast<u64*>(rowidx + ((N * P.dKS + 3) & ~3u));
        u32* outp = P.pool + (size_t)(g.out + soff) * P.pool_stride;
        const u64* coef = reinterpret_cast<const u64*>(acc);
        if (P.ksk_u16) fused_tail<uint16_t, T>(P, coef, rowidx, red, outp, boot, dbg_lweN, dbg_ks);
        else fused_tail<u32, T>(P, coef, rowidx, red, outp, boot, dbg_lweN, dbg_ks);
    }
}

template <int LOGN, int DG, bool AP, bool SPLIT = false, bool W16 = false, bool FOLD = false, bool FUSE = false>
__global__ __launch_bounds__(W16 ? 1024 : (SPLIT ? 512 : 128 * DG)) void k_blind_rotate64d(DevParams P, const bce_gate_desc* __restrict__ descs, u32 n_desc,
                                                               u32 slot_stride, u64* __restrict__ acc_out,
                                                               u32* __restrict__ dbg_lweN, u32* __restrict__ dbg_ks) {
    extern __shared__ __align__(16) double smemd[];
    bootstrap64d<LOGN, DG, AP, SPLIT, W16, FOLD, FUSE>(P, descs[blockIdx.x % n_desc], (blockIdx.x / n_desc) * slot_stride, blockIdx.x, smemd,
                                                       acc_out, dbg_lweN, dbg_ks);
}

// Dataflow evaluation on the config-5 kernel (dag_sched.hpp): one persistent 1,024-thread workgroup per CU (the LDS holds
// one), AP, folded key, fused tail.  No XCD start gate is needed for key locality here -- an AP step reads one RGSW
// ciphertext chosen by the ciphertext's own digit, so workgroups share few key rows even in lock-step.
__global__ __launch_bounds__(1024) void k_bootstrap_dag64(const DevParams* Pp, const DagParams* Dp) {
    extern __shared__ __align__(16) double smemd[];
    u32* mbox = reinterpret_cast<u32*>(smemd);
    dag_worker(Dp, mbox, [&](ConstDagParams& D, u32 t, u32 k) {
        bootstrap64d<11, 3, true, true, true, true, true, BCE_W16_NBUF_AP, BCE_W16_NPRE_AP, true>(
            *as_constant<ConstDevParams>(Pp), D.tasks[t], D.slot_base + k * D.slot_stride, 0, smemd + kDagMailboxWords / 2, nullptr, nullptr, nullptr);
    });
}

// LDS the fused tail needs inside the digit rows of the 16-wave N = 2048 kernel (T = 1024 threads)
bool fused_tail64_fits(const DevParams& P) {
    const size_t N = P.N, NP = N + (N >> 6) * 4, R = 2 * P.dG, T = 1024;
    const size_t VW = P.ksk_u16 ? 8 : 4, G = (P.n + VW) / VW, Gv = G < T ? G : T, RW = (Gv + 63) / 64, SL = (T / 64) / RW;
    const size_t need = ((N * P.dKS + 3) & ~(size_t)3) * 4 + SL * Gv * VW * 8;
    return P.logN == 11 && (size_t)P.n + 1 <= T * VW && RW <= T / 64 && SL >= 1 && need <= R * NP * 8;
}

// key words u64 <-> double in place (exact: Q < 2^39)
__global__ void k_words_u64_f64(u64* __restrict__ w, size_t count, int to_double) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        if (to_double) {
            const u64 v = w[i];
            const double d = fma((double)(u32)(v >> 32), 4294967296.0, (double)(u32)v);
            w[i] = (u64)__double_as_longlong(d);
        } else {
            const double d = __longlong_as_double((long long)w[i]);
            const double hi = floor(d * (1.0 / 4294967296.0));
            w[i] = ((u64)(u32)hi << 32) | (u64)(u32)fma(-hi, 4294967296.0, d);
        }
    }
}
}  // namespace wd

hipError_t launch_words_u64_f64(u64* words, size_t count, int to_double, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(wd::k_words_u64_f64, dim3(4096), dim3(256), 0, s, words, count, to_double);
    return hipGetLastError();
}

// the narrow build of the integer kernel (32-bit digit rows): four gadget digits on N >= 1024, Q < 2^31
bool blind_rotate64_narrow(const DevParams& P) {
    return P.is64 && !P.fp64 && P.dG == 4 && P.logN >= 10 && P.Q64 < (1ull << 31);
}

size_t blind_rotate64_lds_bytes(const DevParams& P) {
    const size_t N = P.N, NP = N + (N >> 6) * 4, R = 2 * P.dG;
    if (blind_rotate64_narrow(P)) return 2 * NP * sizeof(u64) + R * NP * sizeof(u32) + 2 * NP * sizeof(u64) + ((P.n + 1 + 3) & ~3u) * sizeof(u32);
    // the 8-wave double-precision kernel (N = 2048, 3 gadget digits) also mirrors the first 1024 twiddle entries
    const size_t mirror = (P.fp64 && P.logN == 11 && P.dG == 3) ? 1024 * sizeof(double2) : 0;
    return (2 + R) * NP * sizeof(u64) + ((P.n + 1 + 3) & ~3u) * sizeof(u32) + mirror;
}

bool dag64_kernel_available(const DevParams& P) {
    // what launch_blind_rotate64 would run with the tail fused: AP, 16 waves, folded key
    return P.is64 && P.fp64 && P.logN == 11 && P.dG == 3 && P.method_ap && P.fold && P.fuse_tail && P.variant != 2 && wd::fused_tail64_fits(P);
}

hipError_t launch_bootstrap_dag64(const DevParams& P, const DevParams* d_P, const DagParams* d_params, u32 grid, hipStream_t s) {
    if (!dag64_kernel_available(P)) return hipErrorInvalidValue;
    const size_t lds = blind_rotate64_lds_bytes(P) + kDagMailboxWords * 4;   // + the worker's mailbox in front
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wd::k_bootstrap_dag64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wd::k_bootstrap_dag64, dim3(grid), dim3(1024), lds, s, d_P, d_params);
    return hipGetLastError();
}

hipError_t launch_blind_rotate64(const DevParams& P, const bce_gate_desc* d, u32 n_desc, u32 instances, u32 slot_stride,
                                 u64* acc_out, hipStream_t s, u32* dbg_lweN, u32* dbg_ks, bool* tail_fused, LaunchEvents ev) {
    using K = void (*)(DevParams, const bce_gate_desc*, u32, u32, u64*, u32*, u32*);
    if (tail_fused) *tail_fused = false;
    const bool ap = P.method_ap != 0;
    u32 threads = 128 * P.dG;
    K kern = nullptr;
    if (P.fp64 && P.dG == 3) {
        switch (P.logN) {
            case 9: kern = ap ? wd::k_blind_rotate64d<9, 3, true> : wd::k_blind_rotate64d<9, 3, false>; break;
            case 10: kern = ap ? wd::k_blind_rotate64d<10, 3, true> : wd::k_blind_rotate64d<10, 3, false>; break;
            case 11: {
                // AP (BASELINE config 5): the 16-wave build (-8 % per launch); GINX keeps the 8-wave build, whose two-key MAC
                // items do not fit the 128-register budget (16 waves: +9 %).  BCE_VARIANT=2 / 3 force 8 / 16 waves.
                const bool w16 = P.variant == 3 || (P.variant != 2 && ap);
                if (w16) {
                    // the AP build with the folded key (BASELINE config 5) carries the tail in its epilogue
                    const bool fuse = ap && P.fold && tail_fused && P.fuse_tail && wd::fused_tail64_fits(P);
                    if (fuse) { kern = wd::k_blind_rotate64d<11, 3, true, true, true, true, true>; *tail_fused = true; }
                    else if (P.fold) kern = ap ? wd::k_blind_rotate64d<11, 3, true, true, true, true> : wd::k_blind_rotate64d<11, 3, false, true, true, true>;
                    else kern = ap ? wd::k_blind_rotate64d<11, 3, true, true, true> : wd::k_blind_rotate64d<11, 3, false, true, true>;
                    threads = 1024;
                } else {
                    if (P.fold) kern = ap ? wd::k_blind_rotate64d<11, 3, true, true, false, true> : wd::k_blind_rotate64d<11, 3, false, true, false, true>;
                    else kern = ap ? wd::k_blind_rotate64d<11, 3, true, true> : wd::k_blind_rotate64d<11, 3, false, true>;
                    threads = 512;
                }
                break;
            }
            default: break;
        }
    } else if (P.fp64 && P.dG == 4 && P.logN == 9) {
        kern = ap ? wd::k_blind_rotate64d<9, 4, true> : wd::k_blind_rotate64d<9, 4, false>;
    } else if (P.dG == 3) {
        switch (P.logN) {
            case 9: kern = ap ? w64::k_blind_rotate64<9, 3, true> : w64::k_blind_rotate64<9, 3, false>; break;
            case 10: kern = ap ? w64::k_blind_rotate64<10, 3, true> : w64::k_blind_rotate64<10, 3, false>; break;
            case 11: kern = ap ? w64::k_blind_rotate64<11, 3, true> : w64::k_blind_rotate64<11, 3, false>; break;
            default: break;
        }
    } else if (P.dG == 4 && P.logN == 9) {
        kern = ap ? w64::k_blind_rotate64<9, 4, true> : w64::k_blind_rotate64<9, 4, false>;
    } else if (blind_rotate64_narrow(P)) {
        // STD256 / STD256_OPT (N = 2048) and their N = 1024 siblings: 32-bit digit rows
        if (P.logN == 11) kern = ap ? w64::k_blind_rotate64<11, 4, true, 2, 1, true> : w64::k_blind_rotate64<11, 4, false, 1, 0, true>;
        else kern = ap ? w64::k_blind_rotate64<10, 4, true, 3, 2, true> : w64::k_blind_rotate64<10, 4, false, 2, 1, true>;
    }
    if (!kern || (P.fold && !(P.fp64 && P.dG == 3 && P.logN == 11))) return hipErrorInvalidValue;  // folded key: N = 2048 fp64 kernels only
    const size_t lds = blind_rotate64_lds_bytes(P);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const bool fused = tail_fused && *tail_fused;
    if (ev.start || ev.stop)
        return launch_with_events(kern, dim3(n_desc * instances), dim3(threads), lds, s, ev, P, d, n_desc, slot_stride, acc_out,
                                  fused ? dbg_lweN : nullptr, fused ? dbg_ks : nullptr);
    hipLaunchKernelGGL(kern, dim3(n_desc * instances), dim3(threads), lds, s, P, d, n_desc, slot_stride, acc_out,
                       fused ? dbg_lweN : nullptr, fused ? dbg_ks : nullptr);
    return hipGetLastError();
}

hipError_t launch_ntt_batch64(const DevParams& P, u64* polys, u32 count, int inverse, hipStream_t s) {
    if (count == 0) return hipSuccess;
    const size_t N = P.N, NP = N + (N >> 6) * 4;
    const u32 W = 4;
    const size_t lds = W * NP * sizeof(u64);
    u32 blocks = (count + W - 1) / W;
    if (blocks > 4096) blocks = 4096;
    void (*kern)(DevParams, u64*, u32, int) = nullptr;
    switch (P.logN) {
        case 9: kern = w64::k_ntt_batch64<9>; break;
        case 10: kern = w64::k_ntt_batch64<10>; break;
        case 11: kern = w64::k_ntt_batch64<11>; break;
        default: return hipErrorInvalidValue;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * W), lds, s, P, polys, count, inverse);
    return hipGetLastError();
}

hipError_t launch_pointwise_mac64(const DevParams& P, u64* b, const u64* a, const u64* z, u32 count, u32 b_step,
                                  hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(w64::k_pointwise_mac64, dim3(2048), dim3(256), 0, s, P, b, a, z, count, b_step);
    return hipGetLastError();
}

}  // namespace bce

#ifdef BCE_PHASE_PROF
extern "C" int bce_debug_phase_prof64(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(bce::g_phase_prof64), sizeof(unsigned long long) * 256) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[256] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(bce::g_phase_prof64), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
