// fused_tail.hpp -- the tail of EvalBinGate run by the workgroup that finished the blind rotation (included inside
// namespace bce by kernels.hip and kernels64.hip): transpose + sample extract, ModSwitch(Q -> qKS) (RoundqQ), LWE
// KeySwitch as a row gather, ModSwitch(qKS -> q) into the pool.  Reference: OpenFHE binfhe-base-scheme.cpp EvalBinGate /
// lwe-pke.cpp ModSwitch, KeySwitch as called from src/gate.cpp:133,146,172,200-202.
#pragma once

// LWEEncryptionScheme::RoundqQ restated with the same three IEEE double operations
// (compiled with -ffp-contract=off): floor(0.5 + double(v) * double(q) / double(Q)) mod q
__device__ __forceinline__ u32 round_qQ(u64 v, u32 q, u64 Qfrom) {
    double t = (double)v * (double)q;
    t = t / (double)Qfrom;
    u64 r = (u64)floor(0.5 + t);
    return (u32)(r >= q ? r - q : r);
}

// ---- tail of EvalBinGate fused into the blind-rotation kernel (saturated launches) ------------------------------
// After the last inverse transform the workgroup that ran the blind rotation also extracts the LWE sample,
// switches it to qKS, gathers its N*dKS key-switching rows and writes the refreshed ciphertext: the same arithmetic
// as k_tail_gather / k_tail_finish below (one workgroup per bootstrap, S = 1), but its row gather -- memory-bound,
// 2 MB per bootstrap for STD128 -- runs while the CU's other workgroup keeps the vector ALUs busy, instead of as a
// separate kernel between two dependent blind-rotation launches.
//   coef : [2][N] coefficient-form accumulator in LDS        rowidx : [N * dKS] row numbers in LDS
//   red  : [SL][Gv * VW] u64 partial sums in LDS (SL row slices, one per RW = ceil(Gv / 64) waves)
// T threads (a multiple of 64); every thread of the workgroup must call it.
template <typename KT, u32 T, typename CW, typename PT>
__device__ __forceinline__ void fused_tail(const PT& P, const CW* coef, u32* rowidx, u64* red, u32* out, u32 boot,
                                           u32* __restrict__ dbg_lweN, u32* __restrict__ dbg_ks) {
    constexpr u32 VW = 16 / sizeof(KT), W = T / 64;
    const u32 N = P.N, n = P.n, qKS = P.qKS, B = P.baseKS, D = P.dKS;
    const u64 Q = sizeof(CW) == 8 ? (u64)P.Q64 : (u64)P.Q;            // coefficient words: u32 (Q < 2^28) or u64
    const u64 Q8p1 = sizeof(CW) == 8 ? (u64)P.Q8p1_64 : (u64)P.Q8p1;
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // transpose (X -> X^-1) of acc[0]: a'_0 = a_0, a'_{N-i} = -a_i; ModSwitch(Q -> qKS); digits -> row numbers
    for (u32 i = tid; i < N; i += T) {
        const u64 src = (i == 0) ? coef[0] : coef[N - i];
        const u64 v = (i == 0) ? src : (src ? Q - src : 0);
        u32 at = round_qQ(v, qKS, Q);
        if (dbg_lweN) dbg_lweN[(size_t)boot * (N + 1) + i] = at;
        for (u32 j = 0; j < D; ++j) {
            rowidx[i * D + j] = (i * B + at % B) * D + j;
            at /= B;
        }
    }
    __syncthreads();
    const KT* __restrict__ ksk = reinterpret_cast<const KT*>(P.ksk);
    const u32 G = (n + VW) / VW;                  // 16-byte groups holding elements 0..n
    const u32 Gv = G < T ? G : T;                 // (n + 1 <= T * VW for every parameter set this kernel serves)
    const u32 RW = (Gv + 63) / 64, SL = W / RW;   // waves per row, row slices
    const u32 slice = wave / RW, group = (wave - slice * RW) * 64 + lane;
    const u32 LR = N * D;
    if (slice < SL && group < Gv) {
        u64 tot[VW];
#pragma unroll
        for (u32 e = 0; e < VW; ++e) tot[e] = 0;
        const u32 CH = P.ks_chunk;                // rows whose elements can be summed in 32 bits
#ifndef BCE_FUSED_U
#define BCE_FUSED_U 8
#endif
        constexpr u32 U = BCE_FUSED_U;            // rows in flight per lane
        u32 r = slice;
        while (r < LR) {
            u32 run[VW];
#pragma unroll
            for (u32 e = 0; e < VW; ++e) run[e] = 0;
            const u32 rend = (LR - r > CH * SL) ? r + CH * SL : LR;
            auto add_row = [&](uint4 v) {
                if constexpr (sizeof(KT) == 2) {
                    run[0] += v.x & 0xFFFFu; run[1] += v.x >> 16; run[2] += v.y & 0xFFFFu; run[3] += v.y >> 16;
                    run[4] += v.z & 0xFFFFu; run[5] += v.z >> 16; run[6] += v.w & 0xFFFFu; run[7] += v.w >> 16;
                } else {
                    run[0] += v.x; run[1] += v.y; run[2] += v.z; run[3] += v.w;
                }
            };
            for (; r + (U - 1) * SL < rend; r += U * SL) {
                uint4 v[U];
#pragma unroll
                for (u32 u = 0; u < U; ++u) {
                    const u32 row = __builtin_amdgcn_readfirstlane(rowidx[r + u * SL]);
                    v[u] = reinterpret_cast<const uint4*>(ksk + (size_t)row * P.ksk_stride)[group];
                }
#pragma unroll
                for (u32 u = 0; u < U; ++u) add_row(v[u]);
            }
            for (; r < rend; r += SL) {
                const u32 row = __builtin_amdgcn_readfirstlane(rowidx[r]);
                add_row(reinterpret_cast<const uint4*>(ksk + (size_t)row * P.ksk_stride)[group]);
            }
#pragma unroll
            for (u32 e = 0; e < VW; ++e) tot[e] += run[e];
        }
#pragma unroll
        for (u32 e = 0; e < VW; ++e) red[(size_t)slice * Gv * VW + group * VW + e] = tot[e];
    }
    __syncthreads();
    // KeySwitch: a' = -sum_rows A[row], b' = b - sum_rows B[row] (mod qKS), b = acc[1][0] + Q/8 + 1 mod-switched;
    // then ModSwitch(qKS -> q) into the pool
    for (u32 k = tid; k <= n; k += T) {
        u64 sum = 0;
        for (u32 sl = 0; sl < SL; ++sl) sum += red[(size_t)sl * Gv * VW + k];
        const u32 sm = (u32)(sum % qKS);
        u32 base = 0;
        if (k == n) {
            u64 b = (u64)coef[N] + Q8p1;
            b = b >= Q ? b - Q : b;
            base = round_qQ(b, qKS, Q);
            if (dbg_lweN) dbg_lweN[(size_t)boot * (N + 1) + N] = base;
        }
        const u32 v = base >= sm ? base - sm : base + qKS - sm;
        if (dbg_ks) dbg_ks[(size_t)boot * (n + 1) + k] = v;
        out[k] = round_qQ(v, P.q, qKS);
    }
}

