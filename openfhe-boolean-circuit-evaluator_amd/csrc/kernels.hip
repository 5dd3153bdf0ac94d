// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the gate-bootstrapping engine (32-bit ring modulus).
//
// Hot path replaced: OpenFHE's BinFHEContext::EvalBinGate as called by the reference at
// src/gate.cpp:133,146,172,200-202 (one call per gate inside an OpenMP task,
// src/circuit.cpp:698-710).  Here one workgroup runs one gate bootstrap:
//
//   k_blind_rotate        LWE prep (ct1+ct2, folded NOTs) -> test vector -> n x AddToAcc
//                         (2 INTT, signed digit decomposition, 2*dG NTT, RGSW MAC with the
//                         bootstrapping key streamed from HBM/L2, monomial multiply) -> INTT;
//                         ONE WAVE PER TRANSFORM: every 32-bit parameter set, GINX and AP
//   k_blind_rotate_lat    the same computation for N = 1024, dG = 4 (STD128 / STD128_OPT) with the inverse
//                         transforms SPLIT over all 8 waves, passes fused across phases and key rows
//                         requested ahead of their use; 256- and 128-register builds (one / two
//                         workgroups per CU), chosen by launch size in launch_blind_rotate()
//   k_tail_gather/_finish transpose + sample extract + ModSwitch(Q->qKS) + LWE KeySwitch (row gather,
//                         16-byte pieces per lane, rows split over waves / workgroups) + ModSwitch(qKS->q)
//
// All arithmetic is 32-bit unsigned modular integer (Q < 2^28): lazy Shoup butterflies on register
// pairs (forward, values below 22Q, 5 instructions) and compile-time bound-tracked Gentleman-Sande
// stages (inverse), 64-bit MAC sums folded and reduced by lazy Barrett steps; the non-lazy Harvey
// forms remain for Q >= 2^27.6.  No MFMA: nothing here is a dense contraction.
//
// NTT organisation (64-wide wavefronts): E = N/64 coefficients per lane held in registers, log2(E)
// radix-2 stages per register pass, LDS re-shuffles between passes instead of one barrier per stage.
// A polynomial is stored in LDS with 4 pad words per 64 (see DESIGN.md section 4 for the measured
// access-shape costs and for what is and is not on the critical path).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <utility>

#include "kernels.hpp"

namespace bce {

#include "phase_prof.hpp"
#include "fused_tail.hpp"
#include "dag_sched.hpp"
#ifdef BCE_PHASE_PROF
__device__ unsigned long long g_phase_prof[BCE_PROF_WAVES * BCE_PROF_SLOTS];
#define BCE_PROF_ARRAY ::bce::g_phase_prof
#endif

// ---------------------------------------------------------------------------------------
// modular arithmetic
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ u32 csub(u32 x, u32 m) { return min(x, x - m); }  // x in [0,2m) -> [0,m)

// y * w mod Q, lazily: result in [0, 2Q) for any 32-bit y.  w = (value, floor(value*2^32/Q))
__device__ __forceinline__ u32 mul_shoup_lazy(u32 y, uint2 w, u32 Q) {
    u32 qh = __umulhi(w.y, y);
    return w.x * y - qh * Q;
}

// ---- register-pair helpers for the 5-instruction lazy butterfly -------------------------------
// v_mad_u64_u32 is full rate on gfx950 (same issue cost as v_mul_lo_u32, measured), so using its LOW
// word gives multiply+add in one instruction.  Its addend and result are 64-bit register pairs; NTT
// elements are therefore carried as pairs whose high half is a don't-care, which costs registers
// but no instructions.
__device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c) {
    u64 r, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return r;
}
// a*b as a register pair (addend = inline constant 0)
__device__ __forceinline__ u64 mul64(u32 a, u32 b) {
    u64 r, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(carry) : "v"(a), "v"(b));
    return r;
}
// y * w mod Q lazily in [0, 2Q) with 3 instructions: v_mul_hi_u32 + 2 x v_mad_u64_u32 (low word)
__device__ __forceinline__ u32 mul_shoup_lazy3(u32 y, uint2 w, u32 Q) {
    return (u32)mad64(__umulhi(w.y, y), 0u - Q, mul64(y, w.x));
}
__device__ __forceinline__ u64 pair_of(u32 lo) {
    // an empty asm "defines" a fresh register pair without any instruction; only its low half is then
    // written, so the producer of `lo` can target that half directly (a frozen don't-care value would
    // have to be COPIED into every pair's high half instead)
    u64 p;
    asm volatile("" : "=v"(p));
    return (p & 0xFFFFFFFF00000000ull) | lo;
}
__device__ __forceinline__ u64 with_lo(u64 pair, u32 lo) { return (pair & 0xFFFFFFFF00000000ull) | lo; }

// x < 2^(32+shift) -> value congruent to x mod Q in [0, 3Q): shift, mulhi, one low-word mad
__device__ __forceinline__ u32 barrett_lazy3(u64 x, u32 Q, u32 shift, u32 mu) {
    return (u32)mad64(__umulhi((u32)(x >> shift), mu), 0u - Q, x);
}

// x < 2^(2*bitlen(Q)+3) -> x mod Q in [0, Q)
__device__ __forceinline__ u32 barrett_reduce(u64 x, u32 Q, u32 shift, u32 mu) {
    u32 x1 = (u32)(x >> shift);
    u32 qh = __umulhi(x1, mu);
    u32 r = (u32)x - qh * Q;  // < 3Q
    r = csub(r, 2 * Q);
    return csub(r, Q);
}

// x < 2^64 with (x >> 32) * c32 + (x & 0xFFFFFFFF) < 2^(32+shift), c32 = 2^32 mod Q: fold, then reduce
__device__ __forceinline__ u32 barrett_fold(u64 x, u32 c32, u32 Q, u32 shift, u32 mu) {
    return barrett_reduce((u64)(u32)(x >> 32) * c32 + (u32)x, Q, shift, mu);
}

// ---------------------------------------------------------------------------------------
// LDS polynomial layout and register passes
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ u32 phys(u32 j) { return j + ((j >> 6) << 2); }

template <int LOGN>
struct Cfg {
    static constexpr int N = 1 << LOGN;
    static constexpr int LE = LOGN - 6;          // log2 coefficients per lane
    static constexpr int E = 1 << LE;            // coefficients per lane
    static constexpr int NP = N + (N >> 6) * 4;  // padded words per polynomial
    static constexpr int F2LO = (6 > LE) ? 6 - LE : 0;  // low register bit of forward pass 2
    static_assert(LOGN >= 9 && LOGN <= 11, "supported ring sizes: 512, 1024, 2048");
};

// Twiddle table layout.  Natural index m + i (m = 2^s the Cooley-Tukey block, i < m) holds
// psi^brv(m+i).  Blocks with m >= 64 are read by the passes whose registers hold the low index
// bits: lane L needs entries i = (L << sh) | t (sh = s - 6), a stride-2^sh pattern that is a
// 2^sh-way LDS bank conflict.  Those blocks are therefore stored TRANSPOSED: entry (L, t) at
// m + t*64 + L, so that a wave reads 64 consecutive uint2.  (The host builds the same layout.)
template <u32 M>
__device__ __forceinline__ u32 tw_pos(u32 i) {
    if constexpr (M < 64) {
        return M + i;
    } else {
        constexpr u32 sh = __builtin_ctz(M) - 6;
        return M + ((i & ((1u << sh) - 1u)) << 6) + (i >> sh);
    }
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains outstanding GLOBAL loads
// (s_waitcnt vmcnt(0)), which would serialise key rows requested ahead of their use.
__device__ __forceinline__ void block_sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// exchange that stays inside one wave (split inverse transform, passes 0..3).  -DBCE_STEP_BARRIERS restores the
// workgroup barriers of the round-1 schedule for A/B runs (tools/barrier_ab.sh).
__device__ __forceinline__ void wave_sync();
__device__ __forceinline__ void wave_local_sync() {
#ifdef BCE_STEP_BARRIERS
    block_sync_lds();
#else
    wave_sync();
#endif
}
__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wave execute in order; this only pins the compiler.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// register r of lane `lane` holds coefficient j = (lane_hi << (LO+LE)) | (r << LO) | lane_lo
template <int LOGN, int LO>
__device__ __forceinline__ u32 elem_j(u32 lane, int r) {
    constexpr int LE = Cfg<LOGN>::LE;
    u32 lane_lo = lane & ((1u << LO) - 1u);
    u32 lane_hi = lane >> LO;
    return (lane_hi << (LO + LE)) | ((u32)r << LO) | lane_lo;
}

template <int LOGN, int LO>
__device__ __forceinline__ void load_pass(const u32* poly, u32 lane, u32 (&x)[Cfg<LOGN>::E]) {
    constexpr int E = Cfg<LOGN>::E;
    if constexpr (LO == 0) {
        const uint4* p = reinterpret_cast<const uint4*>(poly + phys(lane * E));
#pragma unroll
        for (int k = 0; k < E / 4; ++k) {
            uint4 v = p[k];
            x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int r = 0; r < E; ++r) x[r] = poly[phys(elem_j<LOGN, LO>(lane, r))];
    }
}

template <int LOGN, int LO>
__device__ __forceinline__ void store_pass(u32* poly, u32 lane, const u32 (&x)[Cfg<LOGN>::E]) {
    constexpr int E = Cfg<LOGN>::E;
    if constexpr (LO == 0) {
        uint4* p = reinterpret_cast<uint4*>(poly + phys(lane * E));
#pragma unroll
        for (int k = 0; k < E / 4; ++k) p[k] = make_uint4(x[4 * k], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3]);
    } else {
#pragma unroll
        for (int r = 0; r < E; ++r) poly[phys(elem_j<LOGN, LO>(lane, r))] = x[r];
    }
}

// Cooley-Tukey stage on coefficient-index bit B (distance 2^B), twiddle tw[m + (j >> (B+1))].
// LAZY: no per-stage correction at all -- every stage adds at most 2Q, so values stay below
// (2*LOGN+1)*Q < 2^32 (host checks); otherwise Harvey's [0,4Q) form with one correction per stage.
template <int LOGN, int LO, int B, bool LAZY>
__device__ __forceinline__ void fwd_stage(u32 (&x)[Cfg<LOGN>::E], u32 lane, const uint2* tw, u32 Q) {
    constexpr int LE = Cfg<LOGN>::LE, E = Cfg<LOGN>::E;
    constexpr int rb = B - LO;
    constexpr u32 m = 1u << (LOGN - 1 - B);
    const u32 hi = (lane >> LO) << (LO + LE - B - 1);
    const u32 Q2 = 2 * Q;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if (r & (1 << rb)) continue;
        uint2 w = tw[tw_pos<m>(hi | (u32)(r >> (rb + 1)))];
        u32 X = LAZY ? x[r] : csub(x[r], Q2);
        u32 T = mul_shoup_lazy(x[r | (1 << rb)], w, Q);
        x[r] = X + T;
        x[r | (1 << rb)] = X + Q2 - T;
    }
}

// Lazy Cooley-Tukey stage on register pairs: X' = lo(mad(floor(Y*w'/2^32), -Q, mad(Y, w, X))),
// Y' = 2X + 2Q - X'  ->  v_mul_hi_u32, 2 x v_mad_u64_u32, v_lshl_add_u32, v_sub_u32.
template <int LOGN, int LO, int B>
__device__ __forceinline__ void fwd_stage_pair(u64 (&x)[Cfg<LOGN>::E], u32 lane, const uint2* tw, u32 Q) {
    constexpr int LE = Cfg<LOGN>::LE, E = Cfg<LOGN>::E;
    constexpr int rb = B - LO;
    constexpr u32 m = 1u << (LOGN - 1 - B);
    const u32 hi = (lane >> LO) << (LO + LE - B - 1);
    const u32 Q2 = 2 * Q, negQ = 0u - Q;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if (r & (1 << rb)) continue;
        const uint2 w = tw[tw_pos<m>(hi | (u32)(r >> (rb + 1)))];
        const u32 X = (u32)x[r], Y = (u32)x[r | (1 << rb)];
        const u64 t = mad64(__umulhi(Y, w.y), negQ, mad64(Y, w.x, x[r]));
        x[r | (1 << rb)] = with_lo(x[r | (1 << rb)], (X << 1) + Q2 - (u32)t);
        x[r] = t;
    }
}
// First stage of a pass whose top register bit is the stage bit (one twiddle per lane: registers r and r + E/2),
// straight from PLAIN registers into pairs: only the X operands are
// needed as 64-bit addends, the Y' results are born in pair low halves -- no register-pair set-up moves
// for the 16-byte loads of the lane-major hand-off.
template <int LOGN>
__device__ __forceinline__ void fwd_first_stage_pair(const u32 (&x)[Cfg<LOGN>::E], u64 (&xp)[Cfg<LOGN>::E], uint2 w, u32 Q) {
    constexpr int E = Cfg<LOGN>::E, Hh = E / 2;
    const u32 Q2 = 2 * Q, negQ = 0u - Q;
#pragma unroll
    for (int r = 0; r < Hh; ++r) {
        const u32 X = x[r], Y = x[r + Hh];
        const u64 t = mad64(__umulhi(Y, w.y), negQ, mad64(Y, w.x, pair_of(X)));
        xp[r] = t;
        xp[r + Hh] = pair_of((X << 1) + Q2 - (u32)t);
    }
}
template <int LOGN, int LO, int BHI, int BLO>
__device__ __forceinline__ void fwd_stages_pair(u64 (&x)[Cfg<LOGN>::E], u32 lane, const uint2* tw, u32 Q) {
    if constexpr (BHI >= BLO) {
        fwd_stage_pair<LOGN, LO, BHI>(x, lane, tw, Q);
        fwd_stages_pair<LOGN, LO, BHI - 1, BLO>(x, lane, tw, Q);
    }
}
template <int LOGN, int LO>
__device__ __forceinline__ void load_pass_pair(const u32* poly, u32 lane, u64 (&x)[Cfg<LOGN>::E]) {
#pragma unroll
    for (int r = 0; r < Cfg<LOGN>::E; ++r) x[r] = pair_of(poly[phys(elem_j<LOGN, LO>(lane, r))]);
}
// Lane-major hand-off layout between the digit decomposition and the first forward pass: both
// hold coefficient (r << 6) | lane in register r, so the layout in between is private to them.
// Word k*256 + lane*4 + e holds register 4k + e: 16-byte accesses, consecutive lanes contiguous
// (measured on gfx950: ds_write_b128 12.5 vs 4 x ds_write_b32 16 cycles per KiB, ds_read_b128 4.1
// vs 4 x ds_read_b32 9.2; tools/lds_shapes.hip).
__device__ __forceinline__ u32 lm_word(u32 lane, int k) { return (u32)k * 256u + lane * 4u; }
template <int LOGN>
__device__ __forceinline__ void load_lm(const u32* poly, u32 lane, u32 (&x)[Cfg<LOGN>::E]) {
#pragma unroll
    for (int k = 0; k < Cfg<LOGN>::E / 4; ++k) {
        const uint4 v = *reinterpret_cast<const uint4*>(poly + lm_word(lane, k));
        x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
    }
}
template <int LOGN, int LO>
__device__ __forceinline__ void store_pass_pair(u32* poly, u32 lane, const u64 (&x)[Cfg<LOGN>::E]) {
#pragma unroll
    for (int r = 0; r < Cfg<LOGN>::E; ++r) poly[phys(elem_j<LOGN, LO>(lane, r))] = (u32)x[r];
}

// Gentleman-Sande stages with COMPILE-TIME BOUND TRACKING.  Inverse twiddles come from the
// FORWARD table: psi^-k = -psi^(N-k), i.e. itw[m+i] = -tw[m + (m-1-i)]; the sign goes into the
// operand, so no inverse table and no per-twiddle negation is needed.
//
// A butterfly is X' = X + Y, Y' = (X - Y) * w = (Y + bQ - X) * (-w) with X, Y < bQ.  The Shoup product accepts any
// 32-bit operand and returns [0, 2Q), so only the SUM side grows: a register that takes the sum
// side t times in a row holds < 2^(t+1) Q.  Inside a register pass the side every register takes
// at every stage is a compile-time fact (the stage bit is a bit of the register index), so the
// per-stage conditional subtraction is dropped and each register is reduced once, at the end of
// the pass, by the cheapest step its own bound needs.  Bounds are capped at GS_CAP*Q = 16Q < 2^32
// (Q < 2^28): a stage whose sum would exceed the cap first halves its two inputs.
// gs_bound: bound (units of Q) of register r before the stage on register bit rb, in a pass whose
// first stage is on register bit rb0 and whose inputs are < 2Q.
constexpr int GS_CAP = 16;
constexpr int gs_bound(int r, int rb0, int rb) {
    int b = 2;
    for (int s = rb0; s < rb; ++s) {
        if ((r >> s) & 1) b = 2;
        else b = (2 * b > GS_CAP) ? b : 2 * b;  // capped stages first halve their inputs (see inv_bfly)
    }
    return b;
}

template <int LOGN, int LO, int B, int RB0, bool LAST, int R>
__device__ __forceinline__ void inv_bfly(u32 (&x)[Cfg<LOGN>::E], u32 hi, const uint2* tw, u32 Q, uint2 ninv, uint2 wlast) {
    constexpr int rb = B - LO;
    if constexpr ((R & (1 << rb)) == 0) {
        constexpr u32 m = 1u << (LOGN - 1 - B);
        constexpr int S = R | (1 << rb);
        constexpr int b0 = gs_bound(R, RB0, rb);
        constexpr bool cap = 2 * b0 > GS_CAP;
        constexpr u32 b = cap ? b0 / 2 : b0;
        u32 X = x[R], Y = x[S];
        if constexpr (cap) { X = csub(X, b * Q); Y = csub(Y, b * Q); }
        if constexpr (LAST) {
            // stage LOGN-1 has the single twiddle -I: N^-1 is folded into both outputs
            x[R] = csub(mul_shoup_lazy3(X + Y, ninv, Q), Q);
            x[S] = csub(mul_shoup_lazy3(X + b * Q - Y, wlast, Q), Q);
        } else {
            // (X - Y) * (-f) = (Y - X) * f: the forward entry is used as it is, the OPERAND is negated
            const uint2 f = tw[tw_pos<m>((m - 1) - (hi | (u32)(R >> (rb + 1))))];
            x[R] = X + Y;
            x[S] = mul_shoup_lazy3(Y + b * Q - X, f, Q);
        }
    }
}
template <int LOGN, int LO, int B, int RB0, bool LAST, int... R>
__device__ __forceinline__ void inv_stage_seq(u32 (&x)[Cfg<LOGN>::E], u32 lane, const uint2* tw, u32 Q, uint2 ninv,
                                              uint2 wlast, std::integer_sequence<int, R...>) {
    constexpr int LE = Cfg<LOGN>::LE;
    const u32 hi = (lane >> LO) << (LO + LE - B - 1);
    (inv_bfly<LOGN, LO, B, RB0, LAST, R>(x, hi, tw, Q, ninv, wlast), ...);
}
// stages BLO..BHI of one register pass (LASTPASS: BHI = LOGN-1 carries the N^-1 scaling)
template <int LOGN, int LO, int BLO, int BHI, int RB0, bool LASTPASS>
__device__ __forceinline__ void inv_stages(u32 (&x)[Cfg<LOGN>::E], u32 lane, const uint2* tw, u32 Q, uint2 ninv, uint2 wlast) {
    if constexpr (BLO <= BHI) {
        inv_stage_seq<LOGN, LO, BLO, RB0, (LASTPASS && BLO == BHI)>(x, lane, tw, Q, ninv, wlast,
                                                                   std::make_integer_sequence<int, Cfg<LOGN>::E>{});
        inv_stages<LOGN, LO, BLO + 1, BHI, RB0, LASTPASS>(x, lane, tw, Q, ninv, wlast);
    }
}
// end of a (non-final) pass: every register back below 2Q
template <int LOGN, int RB0, int RBEND, int... R>
__device__ __forceinline__ void inv_pass_reduce(u32 (&x)[Cfg<LOGN>::E], u32 Q, u32 mu32, std::integer_sequence<int, R...>) {
    auto red = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
        constexpr int b = gs_bound(r, RB0, RBEND);
        if constexpr (b == 4) x[r] = csub(x[r], 2 * Q);
        else if constexpr (b > 4) x[r] = x[r] - __umulhi(x[r], mu32) * Q;  // any 32-bit value -> [0, 2Q)
    };
    (red(std::integral_constant<int, R>{}), ...);
}

template <int LOGN, int LO, int BHI, int BLO, bool LAZY>
__device__ __forceinline__ void fwd_stages(u32 (&x)[Cfg<LOGN>::E], u32 lane, const uint2* tw, u32 Q) {
    if constexpr (BHI >= BLO) {
        fwd_stage<LOGN, LO, BHI, LAZY>(x, lane, tw, Q);
        fwd_stages<LOGN, LO, BHI - 1, BLO, LAZY>(x, lane, tw, Q);
    }
}
// Forward negacyclic NTT of one polynomial by one wave, in place in LDS.
// Input: natural order, values < 2Q (LAZY) or < 4Q.  Output: bit-reversed order, values in [0, Q)
// (or merely < (2*LOGN+2)*Q when LAZY && !NORM).
// mu32 = floor(2^32 / Q) (LAZY only).
// LM_IN: the input sits in the lane-major hand-off layout (lm_word) instead of the padded natural one.
template <int LOGN, bool LAZY, bool NORM = true, bool LM_IN = false>
__device__ __forceinline__ void ntt_forward_wave(u32* poly, const uint2* twf, u32 lane, u32 Q, u32 mu32) {
    using C = Cfg<LOGN>;
    u32 x[C::E];
    if constexpr (LAZY && C::F2LO > 0) {
        // passes 1 and 2 on register pairs (5-instruction butterflies), pass 3 on plain registers
        u64 xp[C::E];
        if constexpr (LM_IN) {
            load_lm<LOGN>(poly, lane, x);
            fwd_first_stage_pair<LOGN>(x, xp, twf[1], Q);
            fwd_stages_pair<LOGN, 6, LOGN - 2, 6>(xp, lane, twf, Q);
        } else {
            load_pass_pair<LOGN, 6>(poly, lane, xp);
            fwd_stages_pair<LOGN, 6, LOGN - 1, 6>(xp, lane, twf, Q);
        }
        store_pass_pair<LOGN, 6>(poly, lane, xp);
        wave_sync();
        load_pass_pair<LOGN, C::F2LO>(poly, lane, xp);
        fwd_stages_pair<LOGN, C::F2LO, 5, C::F2LO>(xp, lane, twf, Q);
        store_pass_pair<LOGN, C::F2LO>(poly, lane, xp);
        wave_sync();
        load_pass<LOGN, 0>(poly, lane, x);
        fwd_stages<LOGN, 0, C::F2LO - 1, 0, LAZY>(x, lane, twf, Q);
    } else {
        if constexpr (LM_IN) load_lm<LOGN>(poly, lane, x);
        else load_pass<LOGN, 6>(poly, lane, x);
        fwd_stages<LOGN, 6, LOGN - 1, 6, LAZY>(x, lane, twf, Q);
        store_pass<LOGN, 6>(poly, lane, x);
        wave_sync();
        load_pass<LOGN, C::F2LO>(poly, lane, x);
        fwd_stages<LOGN, C::F2LO, 5, C::F2LO, LAZY>(x, lane, twf, Q);
        if constexpr (C::F2LO > 0) {
            store_pass<LOGN, C::F2LO>(poly, lane, x);
            wave_sync();
            load_pass<LOGN, 0>(poly, lane, x);
            fwd_stages<LOGN, 0, C::F2LO - 1, 0, LAZY>(x, lane, twf, Q);
        }
    }
    if constexpr (NORM || !LAZY) {
#pragma unroll
        for (int r = 0; r < C::E; ++r) {
            if constexpr (LAZY) {
                // x < (2*LOGN+2)*Q < 2^32: quotient estimate off by at most 2
                u32 v = x[r] - __umulhi(x[r], mu32) * Q;
                x[r] = csub(csub(v, 2 * Q), Q);
            } else {
                x[r] = csub(csub(x[r], 2 * Q), Q);
            }
        }
    }  // else: outputs stay below (2*LOGN+2)*Q; the RGSW MAC folds them (barrett_fold)
    store_pass<LOGN, 0>(poly, lane, x);
    wave_sync();
}

// Forward NTT (N = 1024) of a polynomial whose stages on bits 9 and 8 were already applied by the producer of
// its coefficients: two passes of four stages on register pairs, one re-shuffle.
// In place, natural padded layout in, bit-reversed order out, values un-normalised (see ntt_forward_wave).
__device__ __forceinline__ void ntt_forward_wave_low8(u32* poly, const uint2* twf, u32 lane, u32 Q) {
    constexpr int LOGN = 10;
    {
        u64 xp[16];
        load_pass_pair<LOGN, 4>(poly, lane, xp);
        fwd_stages_pair<LOGN, 4, 7, 4>(xp, lane, twf, Q);
        store_pass_pair<LOGN, 4>(poly, lane, xp);
    }
    wave_sync();
    u32 x[16];
    u64 xp[16];
    load_pass<LOGN, 0>(poly, lane, x);
    fwd_first_stage_pair<LOGN>(x, xp, twf[tw_pos<64>(lane)], Q);  // stage 3: block m = 64, index j >> 4 = lane
    fwd_stages_pair<LOGN, 0, 2, 0>(xp, lane, twf, Q);
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = (u32)xp[r];
    store_pass<LOGN, 0>(poly, lane, x);
    wave_sync();
}
// The same eight stages on ONE HALF of such a row: positions [512 h, 512 h + 512) are two of the four independent
// 256-point sub-transforms left after the stages on bits 9 and 8.  64 lanes x 8 register pairs, three passes (bits
// 7..5, 4..2, 1..0), two wave-local re-shuffles, in place: half the butterflies of ntt_forward_wave_low8 with every
// lane busy.  The FOLD kernel transforms six rows per step with it: four whole rows on waves 0..3, two rows as four
// halves on waves 4..7 -- 1.5 transforms on every SIMD.
__device__ __forceinline__ void fwd_bfly_pair(u64& xa, u64& xb, uint2 w, u32 Q) {
    const u32 X = (u32)xa, Y = (u32)xb;
    const u64 t = mad64(__umulhi(Y, w.y), 0u - Q, mad64(Y, w.x, xa));
    xb = with_lo(xb, (X << 1) + 2 * Q - (u32)t);
    xa = t;
}
__device__ __forceinline__ void ntt_forward_half_low8(u32* row, u32 h, const uint2* twf, u32 lane, u32 Q) {
    // the lane-dependent LDS addresses below are a few shifts and adds each; opaque to the optimiser, so that they are
    // recomputed per call instead of living in ~10 registers across the caller's step loop (the 128-register build
    // spilled them, and a scratch reload waits on vmcnt(0), i.e. on the key rows in flight)
    asm volatile("" : "+v"(lane));
    u64 xp[8];
    {   // registers = bits 7..5, lane = (bit 8, bits 4..0)
        const u32 l8 = lane >> 5, j0 = (h << 9) | (l8 << 8) | (lane & 31u);
        u32* const p = row + phys(j0);               // register r: j0 + 32 r, same 64-block for r, r + 1
#pragma unroll
        for (int r = 0; r < 8; ++r) xp[r] = pair_of(p[(r >> 1) * 68 + (r & 1) * 32]);
        const u32 i7 = (h << 1) | l8;                // stage on bit B: twiddle tw[m + (j >> (B + 1))], m = 2^(9 - B)
        const uint2 w7 = twf[tw_pos<4>(i7)];
#pragma unroll
        for (int r = 0; r < 4; ++r) fwd_bfly_pair(xp[r], xp[r + 4], w7, Q);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint2 w6 = twf[tw_pos<8>((i7 << 1) | g)];
            fwd_bfly_pair(xp[4 * g], xp[4 * g + 2], w6, Q);
            fwd_bfly_pair(xp[4 * g + 1], xp[4 * g + 3], w6, Q);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) fwd_bfly_pair(xp[2 * g], xp[2 * g + 1], twf[tw_pos<16>((i7 << 2) | g)], Q);
#pragma unroll
        for (int r = 0; r < 8; ++r) p[(r >> 1) * 68 + (r & 1) * 32] = (u32)xp[r];
    }
    wave_sync();
    {   // registers = bits 4..2, lane = (bits 8..5, bits 1..0)
        const u32 lh = lane >> 2, j0 = (h << 9) | (lh << 5) | (lane & 3u);
        u32* const p = row + phys(j0);               // register r: j0 + 4 r, inside one 64-block
#pragma unroll
        for (int r = 0; r < 8; ++r) xp[r] = pair_of(p[4 * r]);
        const u32 i4 = (h << 4) | lh;
        const uint2 w4 = twf[tw_pos<32>(i4)];
#pragma unroll
        for (int r = 0; r < 4; ++r) fwd_bfly_pair(xp[r], xp[r + 4], w4, Q);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint2 w3 = twf[tw_pos<64>((i4 << 1) | g)];
            fwd_bfly_pair(xp[4 * g], xp[4 * g + 2], w3, Q);
            fwd_bfly_pair(xp[4 * g + 1], xp[4 * g + 3], w3, Q);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) fwd_bfly_pair(xp[2 * g], xp[2 * g + 1], twf[tw_pos<128>((i4 << 2) | g)], Q);
#pragma unroll
        for (int r = 0; r < 8; ++r) p[4 * r] = (u32)xp[r];
    }
    wave_sync();
    {   // registers = bits 2..0 (bit 2 is done): 8 consecutive words per lane
        const u32 j0 = (h << 9) | (lane << 3);
        uint4* const p = reinterpret_cast<uint4*>(row + phys(j0));
        const uint4 v0 = p[0], v1 = p[1];
        xp[0] = pair_of(v0.x); xp[1] = pair_of(v0.y); xp[2] = pair_of(v0.z); xp[3] = pair_of(v0.w);
        xp[4] = pair_of(v1.x); xp[5] = pair_of(v1.y); xp[6] = pair_of(v1.z); xp[7] = pair_of(v1.w);
        const u32 i1 = (((h << 6) | lane) << 1);     // j >> 2 for registers 0..3; + 1 for 4..7
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint2 w1 = twf[tw_pos<256>(i1 | g)];
            fwd_bfly_pair(xp[4 * g], xp[4 * g + 2], w1, Q);
            fwd_bfly_pair(xp[4 * g + 1], xp[4 * g + 3], w1, Q);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) fwd_bfly_pair(xp[2 * g], xp[2 * g + 1], twf[tw_pos<512>((i1 << 1) | g)], Q);
        p[0] = make_uint4((u32)xp[0], (u32)xp[1], (u32)xp[2], (u32)xp[3]);
        p[1] = make_uint4((u32)xp[4], (u32)xp[5], (u32)xp[6], (u32)xp[7]);
    }
    wave_sync();
}
// one lazy Cooley-Tukey butterfly on plain registers (5 instructions; X' = X + wY, Y' = X - wY + 2Q)
__device__ __forceinline__ void fwd_bfly(u32& X, u32& Y, uint2 w, u32 Q) {
    const u64 t = mad64(__umulhi(Y, w.y), 0u - Q, mad64(Y, w.x, pair_of(X)));
    Y = (X << 1) + 2 * Q - (u32)t;
    X = (u32)t;
}

// Inverse NTT by one wave: reads `src` (bit-reversed order, values < 2Q), uses `tmp` for the
// re-shuffles (may alias src), leaves coefficient j = (r << 6) | lane in x[r], in [0, Q).
// ninv = N^-1, wlast = -I * N^-1 (I = psi^(N/2)), both with Shoup companions; mu32 = floor(2^32 / Q).
template <int LOGN>
__device__ __forceinline__ void ntt_inverse_wave(const u32* src, u32* tmp, const uint2* twi, u32 lane, u32 Q,
                                                 uint2 ninv, uint2 wlast, u32 mu32, u32 (&x)[Cfg<LOGN>::E]) {
    using C = Cfg<LOGN>;
    constexpr int LE = C::LE;
    constexpr auto regs = std::make_integer_sequence<int, C::E>{};
    load_pass<LOGN, 0>(src, lane, x);
    inv_stages<LOGN, 0, 0, LE - 1, 0, false>(x, lane, twi, Q, ninv, wlast);
    inv_pass_reduce<LOGN, 0, LE>(x, Q, mu32, regs);
    store_pass<LOGN, 0>(tmp, lane, x);
    wave_sync();
    load_pass<LOGN, LE>(tmp, lane, x);
    inv_stages<LOGN, LE, LE, 2 * LE - 1, 0, false>(x, lane, twi, Q, ninv, wlast);
    inv_pass_reduce<LOGN, 0, LE>(x, Q, mu32, regs);
    store_pass<LOGN, LE>(tmp, lane, x);
    wave_sync();
    load_pass<LOGN, 6>(tmp, lane, x);
    inv_stages<LOGN, 6, 2 * LE, LOGN - 1, 2 * LE - 6, true>(x, lane, twi, Q, ninv, wlast);
    // every register is an output of the last stage: scaled by N^-1 and in [0, Q)
}

// psi^e for e in [0, 2N) from the forward table (tw_f[brv(i)] = psi^i, psi^(i+N) = -psi^i)
// psi^e for e in [0, 2N) from the natural-order power table psi_tab[0..N) (psi^(e+N) = -psi^e),
// read through an SGPR buffer resource: one VALU op for the byte offset, no LDS, no bit reversal
template <int LOGN>
__device__ __forceinline__ u32 psi_pow(__amdgpu_buffer_rsrc_t psi_rsrc, u32 e, u32 Q) {
    constexpr u32 N = 1u << LOGN;
    const u32 v = __builtin_amdgcn_raw_buffer_load_b32(psi_rsrc, (e & (N - 1)) << 2, 0, 0);
    return (e & N) ? Q - v : v;
}

// one 16-byte BSK fragment: voff = per-thread byte offset, soff = wave-uniform row byte offset
__device__ __forceinline__ uint4 bsk_row(__amdgpu_buffer_rsrc_t rsrc, u32 voff, u32 soff) {
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}


// gate constant q1 of BootstrapGateCore (OR 5q/8, AND 7q/8, NOR q/8, NAND 3q/8)
__device__ __forceinline__ u32 gate_const(u32 op, u32 q) {
    u32 e = q >> 3;
    switch (op) {
        case BCE_OR: case BCE_XOR_FAST: return 5 * e;
        case BCE_NOR: case BCE_XNOR_FAST: return e;
        case BCE_NAND: return 3 * e;
        default: return 7 * e;  // AND, REFRESH
    }
}

// Tail of one GINX MAC item: sp / sn are the 64-bit row sums against key+ / key- at the 4 consecutive
// evaluation positions p0..p0+3; multiplies them by the monomials psi^(+-(2k+1)a') - 1 and accumulates
// into the 4 accumulator words read at accp, written to accw (also returned in a[], values < 2Q when LAZY).
// I^a' and I^-a' for I = psi^(N/2): the 4 positions sit at evaluation points whose exponents differ by
// multiples of (N/2)*a' (brv(p0+e) = brv(p0) + {0,2,1,3}*N/4).
template <int LOGN, bool LAZY, typename PT>
__device__ __forceinline__ void ginx_mac_tail(const PT& P, __amdgpu_buffer_rsrc_t psi_rsrc, u32 Q, u32 ap, uint2 Ia,
                                              uint2 Ina, u32 p0, const u32* accp, u32* accw, const u64 (&sp)[4],
                                              const u64 (&sn)[4], u32 (&a)[4]) {
    constexpr u32 N = 1u << LOGN;
    const bool odd = ap & 1u;
    const u32 k0 = __brev(p0) >> (32 - LOGN);
    const u32 ex = ((2 * k0 + 1) * ap) & (2 * N - 1);
    u32 mp[4], mn[4];
    mp[0] = psi_pow<LOGN>(psi_rsrc, ex, Q);
    mn[0] = psi_pow<LOGN>(psi_rsrc, (2 * N - ex) & (2 * N - 1), Q);
    mp[2] = csub(mul_shoup_lazy(mp[0], Ia, Q), Q);
    mn[2] = csub(mul_shoup_lazy(mn[0], Ina, Q), Q);
    mp[1] = odd ? Q - mp[0] : mp[0];
    mn[1] = odd ? Q - mn[0] : mn[0];
    mp[3] = odd ? Q - mp[2] : mp[2];
    mn[3] = odd ? Q - mn[2] : mn[2];
    const uint4 a4v = *reinterpret_cast<const uint4*>(accp);
    a[0] = a4v.x; a[1] = a4v.y; a[2] = a4v.z; a[3] = a4v.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if constexpr (LAZY) {
            // fold the high word (sum < 2^62), reduce lazily to [0,3Q); the three-term sum
            // 3Q*Q + 3Q*Q + 2Q stays below the 2^(32+shift) Barrett bound; acc is kept in [0,2Q)
            const u32 rp = barrett_lazy3((u64)(u32)(sp[e] >> 32) * P.c32 + (u32)sp[e], Q, P.red_shift, P.red_mu);
            const u32 rn = barrett_lazy3((u64)(u32)(sn[e] >> 32) * P.c32 + (u32)sn[e], Q, P.red_shift, P.red_mu);
            a[e] = csub(barrett_lazy3((u64)rp * (mp[e] - 1) + (u64)rn * (mn[e] - 1) + a[e], Q, P.red_shift, P.red_mu), 2 * Q);
        } else {
            const u32 rp = barrett_reduce(sp[e], Q, P.red_shift, P.red_mu);
            const u32 rn = barrett_reduce(sn[e], Q, P.red_shift, P.red_mu);
            a[e] = barrett_reduce((u64)rp * (mp[e] - 1) + (u64)rn * (mn[e] - 1) + a[e], Q, P.red_shift, P.red_mu);
        }
    }
    *reinterpret_cast<uint4*>(accw) = make_uint4(a[0], a[1], a[2], a[3]);
}

// The same tail in Montgomery form (R = 2^32), for the split-transform kernels: REDC(x) = (x + ((u32) x * (-Q^-1)) Q) / 2^32
// = x R^-1 mod Q, below x / 2^32 + Q, is TWO instructions (v_mul_lo_u32 + v_mad_u64_u32 -- the quotient is the high register of
// the pair, no shift) where fold + Barrett is five.  rp' = REDC(sp) = sp R^-1, rn' likewise; the monomial factors come from a
// table scaled by R^2, M = (psi^e - 1) R^2 lazily in [0, 2Q); REDC(rp' M+ + rn' M-) = sp (psi^e - 1) + sn (psi^-e - 1) mod Q,
// below 2Q (bounds: engine.cpp, ok5), + acc (< 2Q) < 4Q, one conditional subtraction.  The accumulator words it leaves are
// congruent to the Barrett form's and in the same range [0, 2Q); every reduced value downstream is therefore the same.
__device__ __forceinline__ u32 redc(u64 x, u32 Q, u32 qn) { return (u32)(mad64((u32)x * qn, Q, x) >> 32); }
template <int LOGN, typename PT>
__device__ __forceinline__ void ginx_mac_tail_redc(const PT& P, __amdgpu_buffer_rsrc_t psi_r2_rsrc, u32 Q, u32 ap, uint2 Ia,
                                                   uint2 Ina, u32 p0, const u32* accp, u32* accw, const u64 (&sp)[4],
                                                   const u64 (&sn)[4], u32 (&a)[4]) {
    constexpr u32 N = 1u << LOGN;
    const bool odd = ap & 1u;
    const u32 k0 = __brev(p0) >> (32 - LOGN);
    const u32 ex = ((2 * k0 + 1) * ap) & (2 * N - 1);
    const u32 off = P.r2_off, K = Q + 2 * off, qn = P.qinv_neg;     // -(M - off) + off = K - M
    u32 mp[4], mn[4];
    const u32 p0r = psi_pow<LOGN>(psi_r2_rsrc, ex, Q), n0r = psi_pow<LOGN>(psi_r2_rsrc, (2 * N - ex) & (2 * N - 1), Q);
    mp[0] = p0r + off;
    mn[0] = n0r + off;
    mp[2] = csub(mul_shoup_lazy(p0r, Ia, Q), Q) + off;
    mn[2] = csub(mul_shoup_lazy(n0r, Ina, Q), Q) + off;
    mp[1] = odd ? K - mp[0] : mp[0];
    mn[1] = odd ? K - mn[0] : mn[0];
    mp[3] = odd ? K - mp[2] : mp[2];
    mn[3] = odd ? K - mn[2] : mn[2];
    const uint4 a4v = *reinterpret_cast<const uint4*>(accp);
    a[0] = a4v.x; a[1] = a4v.y; a[2] = a4v.z; a[3] = a4v.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const u32 rp = redc(sp[e], Q, qn), rn = redc(sn[e], Q, qn);
        a[e] = csub(redc(mad64(rn, mn[e], mul64(rp, mp[e])), Q, qn) + a[e], 2 * Q);
    }
    *reinterpret_cast<uint4*>(accw) = make_uint4(a[0], a[1], a[2], a[3]);
}

// Common start of a gate bootstrap: twiddles into LDS, EvalBinGate's LWE preparation (ct1 + ct2 with the
// folded EvalNOTs, or ct + q/4 for a refresh) into av[0..n], BootstrapGateCore's test vector into acc
// (evaluation form).  Ends with a workgroup barrier.
template <int LOGN, bool LAZY, u32 T, typename PT>
__device__ __forceinline__ void bootstrap_prologue(const PT& P, const bce_gate_desc& g, u32 soff, uint2* twf, u32* acc,
                                                   u32* av, u32 tid, u32 lane, u32 wave) {
    using C = Cfg<LOGN>;
    constexpr int N = C::N, NP = C::NP;
    const u32 Q = P.Q, q = P.q, qm = q - 1, n = P.n;
    for (u32 i = tid; i < (u32)N; i += T) twf[i] = P.tw_f[i];
    {   // EvalBinGate LWE prep with folded EvalNOT: (-a, q/4 - b)
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
                v = (v0 + (q >> 2)) & qm;  // Bootstrap(): ct + q/4
            }
            av[i] = v;
        }
    }
    __syncthreads();
    {   // BootstrapGateCore: acc = (0, m(X)), m sparse with +-(Q/8+1)
        const u32 b = av[n];
        const u32 q1 = gate_const(g.op, q), q2 = (q1 + (q >> 1)) & qm;
        const u32 pos = P.Q8p1, neg = Q - P.Q8p1;
        for (u32 j = tid; j < (u32)N; j += T) {
            u32 v = 0;
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
    if (wave == 0) ntt_forward_wave<LOGN, LAZY>(acc + NP, twf, lane, Q, P.mu32);
    __syncthreads();
}

// ---------------------------------------------------------------------------------------
// blind rotation (GINX / CGGI): one workgroup = one gate bootstrap, one wave per RGSW row
// ---------------------------------------------------------------------------------------
// OCC = workgroups the register budget is sized for per CU (2 or 3); LDS per workgroup is
// 8 KiB twiddles + (2 + 2*DG) padded polynomials + ctprep = 52.5 KiB at N = 1024, DG = 4.
// AP = false: GINX/CGGI AddToAcc (two RGSW keys per LWE coefficient, monomial multiply, accumulate);
// AP = true : AP/DM AddToAcc (one RGSW key per base-baseR digit of -a_i, acc is replaced).
template <int LOGN, int DG, bool LAZY, int OCC, bool AP>
__global__ __launch_bounds__(128 * DG, (OCC * 2 * DG + 3) / 4) void k_blind_rotate(
    DevParams P, const bce_gate_desc* __restrict__ descs, u32 n_desc, u32 slot_stride, u32* __restrict__ acc_out) {
    using C = Cfg<LOGN>;
    constexpr int N = C::N, NP = C::NP, E = C::E;
    constexpr u32 R = 2 * DG;
    constexpr u32 T = 64 * R;
    extern __shared__ __align__(16) u32 smem[];
    uint2* twf = reinterpret_cast<uint2*>(smem);
    u32* acc = reinterpret_cast<u32*>(twf + N);  // [2][NP]  EVALUATION domain, [0,Q) ([0,2Q) when LAZY)
    u32* dct = acc + 2 * NP;                     // [R][NP]
    u32* av = dct + R * NP;                      // ctprep: a[0..n), b

    const u32 tid = threadIdx.x;
    const u32 lane = tid & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keep it (and wave*NP) in SGPRs
    const u32 Q = P.Q, q = P.q, qm = q - 1, n = P.n;

    const bce_gate_desc g = descs[blockIdx.x % n_desc];
    const u32 soff = (blockIdx.x / n_desc) * slot_stride;

    bootstrap_prologue<LOGN, LAZY, T>(P, g, soff, twf, acc, av, tid, lane, wave);

    const uint2 ninv = make_uint2(P.Ninv, P.Ninv_s), wlast = make_uint2(P.Winv_last, P.Winv_last_s);
    constexpr u32 rgsw = R * 2 * N;  // words per RGSW ciphertext
    // GINX key = n * 2 * rgsw words < 4 GiB: one buffer resource covers it (raw, no stride, bounds-checked)
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(P.bsk), 0, AP ? 0x7FFFFFFF : (int)(n * 2 * rgsw * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t psi_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(P.psi_tab), 0, N * 4, 0x00020000);
    // GINX: one step per LWE coefficient; AP: one step per (coefficient, base-baseR digit)
    const u32 nsteps = AP ? n * P.dR : n;
    BCE_PROF_INIT();
    for (u32 step = 0; step < nsteps; ++step) {
        u32 ap = 0, rowb = 0;
        const u32* bk;
        if constexpr (!AP) {
            ap = ((q - av[step]) & qm) * P.factor;  // exponent of the monomial, in [0, 2N)
            if (ap == 0) continue;                   // X^0 - 1 = 0: AddToAcc adds nothing
            bk = P.bsk + (size_t)step * 2 * rgsw;
            rowb = step * (2 * rgsw * 4);
        } else {
            const u32 i = step / P.dR, k = step - i * P.dR;
            u32 aI = (q - av[i]) & qm;
            for (u32 t = 0; t < k; ++t) aI /= P.baseR;
            const u32 a0 = aI % P.baseR;
            if (a0 == 0) continue;                   // rgsw-acc-dm.cpp EvalAcc: digit 0 is skipped
            bk = P.bsk + (((size_t)i * P.baseR + a0) * P.dR + k) * rgsw;
        }
        // (1) two waves: INTT of acc[c], SignedDigitDecompose -> dct[2l + c] (coefficient form)
        if (wave < 2) {
            u32 x[E];
            ntt_inverse_wave<LOGN>(acc + wave * NP, dct + wave * NP, twf, lane, Q, ninv, wlast, P.mu32, x);
            const u32 Qh = Q >> 1;
            if constexpr (LAZY) {
                // SignedDigitDecompose in closed form.  The balanced digits r_l in [-B/2, B/2) of
                // d = sum r_l B^l are unique mod B^dG, and d + sum (B/2) B^l = sum (r_l + B/2) B^l has the
                // plain digits r_l + B/2: one v_bfe_u32 per digit instead of the extract/subtract/shift
                // chain.  rem + Q (in (Q - B/2, Q + B/2)) is as good an input as rem mod Q for the lazy NTT.
                const u32 g = P.gBits;
                u32 off = 0;
                for (u32 l = 0; l < (u32)DG; ++l) off |= 1u << (l * g + g - 1);
                const u32 offm = off - Q, bias = Q - (1u << (g - 1));
#pragma unroll
                for (int k = 0; k < E / 4; ++k) {
                    u32 u[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) u[e] = x[4 * k + e] + ((x[4 * k + e] < Qh) ? off : offm);
#pragma unroll
                    for (u32 l = 0; l < (u32)DG; ++l) {
                        uint4 v;
                        v.x = __builtin_amdgcn_ubfe(u[0], l * g, g) + bias;
                        v.y = __builtin_amdgcn_ubfe(u[1], l * g, g) + bias;
                        v.z = __builtin_amdgcn_ubfe(u[2], l * g, g) + bias;
                        v.w = __builtin_amdgcn_ubfe(u[3], l * g, g) + bias;
                        *reinterpret_cast<uint4*>(dct + (2 * l + wave) * NP + lm_word(lane, k)) = v;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    int d = (x[r] < Qh) ? (int)x[r] : (int)x[r] - (int)Q;
                    const u32 pj = phys(((u32)r << 6) | lane);
#pragma unroll
                    for (u32 l = 0; l < (u32)DG; ++l) {
                        int rem = __builtin_amdgcn_sbfe(d, 0, P.gBits);  // signed digit in [-B/2, B/2): v_bfe_i32
                        d = (d - rem) >> P.gBits;
                        dct[(2 * l + wave) * NP + pj] = rem < 0 ? (u32)(rem + (int)Q) : (u32)rem;
                    }
                }
            }
        }
        BCE_PROF_MARK(0);  // thread 0's own phase-1 work
        __syncthreads();
        BCE_PROF_MARK(1);
        // (2) one wave per decomposed polynomial: forward NTT in place
        ntt_forward_wave<LOGN, LAZY, false, LAZY>(dct + wave * NP, twf, lane, Q, P.mu32);
        BCE_PROF_MARK(2);
        __syncthreads();
        BCE_PROF_MARK(3);
        // (3) RGSW multiply-accumulate
        if constexpr (AP) {
            // acc[c] = sum_l dct[l] * ek[l][c]   (rgsw-acc-dm.cpp AddToAcc: the product REPLACES acc)
            for (u32 item = tid; item < 2u * (N / 4); item += T) {
                const u32 c = item / (N / 4), p0 = (item % (N / 4)) * 4;
                const u32 pp = phys(p0);
                const u32* bp = bk + (size_t)c * N + p0;
                uint4 kA[R];
#pragma unroll
                for (u32 l = 0; l < R; ++l) kA[l] = *reinterpret_cast<const uint4*>(bp + (size_t)l * 2 * N);
                u64 sp[4] = {0, 0, 0, 0};
#pragma unroll
                for (u32 l = 0; l < R; ++l) {
                    const uint4 d = *reinterpret_cast<const uint4*>(dct + l * NP + pp);
                    sp[0] += (u64)d.x * kA[l].x; sp[1] += (u64)d.y * kA[l].y; sp[2] += (u64)d.z * kA[l].z; sp[3] += (u64)d.w * kA[l].w;
                }
                u32 a[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    a[e] = LAZY ? barrett_fold(sp[e], P.c32, Q, P.red_shift, P.red_mu) : barrett_reduce(sp[e], Q, P.red_shift, P.red_mu);
                *reinterpret_cast<uint4*>(acc + c * NP + pp) = make_uint4(a[0], a[1], a[2], a[3]);
            }
            __syncthreads();
            continue;
        }
        const u32 a4 = ap & 3u;  // I^a' and I^-a' (ginx_mac_tail)
        const uint2 Ia = make_uint2(P.I4[a4], P.I4s[a4]);
        const uint2 Ina = make_uint2(P.I4[(4u - a4) & 3u], P.I4s[(4u - a4) & 3u]);
        for (u32 item = tid; item < 2u * (N / 4); item += T) {
            const u32 c = item / (N / 4), p0 = (item % (N / 4)) * 4;
            const u32 pp = phys(p0);
            // buffer loads: SGPR resource over the whole key, per-row byte offset in an SGPR (SALU
            // arithmetic), ONE 32-bit per-thread offset -> no 64-bit VALU address math for the 16 rows
            const u32 toff = (c * N + p0) * 4u;
            u64 sp[4] = {0, 0, 0, 0}, sn[4] = {0, 0, 0, 0};
            constexpr u32 H = R / 2;
            if constexpr (OCC >= 3) {
                // 80-VGPR budget (3 workgroups per CU): half of each key's rows in flight at a time
#pragma unroll
                for (u32 h = 0; h < 2; ++h) {
                    uint4 kA[H], kB[H];
#pragma unroll
                    for (u32 l = 0; l < H; ++l) kA[l] = bsk_row(rsrc, toff, rowb + (h * H + l) * (2 * N * 4));
#pragma unroll
                    for (u32 l = 0; l < H; ++l) kB[l] = bsk_row(rsrc, toff, rowb + (rgsw + (h * H + l) * 2 * N) * 4);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (u32 l = 0; l < H; ++l) {
                        const uint4 d = *reinterpret_cast<const uint4*>(dct + (h * H + l) * NP + pp);
                        sp[0] += (u64)d.x * kA[l].x; sp[1] += (u64)d.y * kA[l].y; sp[2] += (u64)d.z * kA[l].z; sp[3] += (u64)d.w * kA[l].w;
                        sn[0] += (u64)d.x * kB[l].x; sn[1] += (u64)d.y * kB[l].y; sn[2] += (u64)d.z * kB[l].z; sn[3] += (u64)d.w * kB[l].w;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            // BSK rows: key+ (R loads) and the first half of key- in flight together; the second
            // half of key- reuses key+'s registers once those are consumed (bounds VGPR pressure)
            uint4 kA[R], kB[H];
#pragma unroll
            for (u32 l = 0; l < R; ++l) kA[l] = bsk_row(rsrc, toff, rowb + l * (2 * N * 4));
#pragma unroll
            for (u32 l = 0; l < H; ++l) kB[l] = bsk_row(rsrc, toff, rowb + (rgsw + l * 2 * N) * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (u32 l = 0; l < R; ++l) {
                const uint4 d = *reinterpret_cast<const uint4*>(dct + l * NP + pp);
                sp[0] += (u64)d.x * kA[l].x; sp[1] += (u64)d.y * kA[l].y; sp[2] += (u64)d.z * kA[l].z; sp[3] += (u64)d.w * kA[l].w;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (u32 l = 0; l < H; ++l) kA[l] = bsk_row(rsrc, toff, rowb + (rgsw + (H + l) * 2 * N) * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (u32 l = 0; l < H; ++l) {
                const uint4 d = *reinterpret_cast<const uint4*>(dct + l * NP + pp);
                sn[0] += (u64)d.x * kB[l].x; sn[1] += (u64)d.y * kB[l].y; sn[2] += (u64)d.z * kB[l].z; sn[3] += (u64)d.w * kB[l].w;
            }
#pragma unroll
            for (u32 l = 0; l < H; ++l) {
                const uint4 d = *reinterpret_cast<const uint4*>(dct + (H + l) * NP + pp);
                sn[0] += (u64)d.x * kA[l].x; sn[1] += (u64)d.y * kA[l].y; sn[2] += (u64)d.z * kA[l].z; sn[3] += (u64)d.w * kA[l].w;
            }
            }
            u32 anew[4];
            ginx_mac_tail<LOGN, LAZY>(P, psi_rsrc, Q, ap, Ia, Ina, p0, acc + c * NP + pp, acc + c * NP + pp, sp, sn, anew);
        }
        BCE_PROF_MARK(4);
        __syncthreads();
        BCE_PROF_MARK(5);
    }
    BCE_PROF_FLUSH();

    // accumulator back to COEFFICIENT form for the extraction kernel
    if (wave < 2) {
        u32 x[E];
        ntt_inverse_wave<LOGN>(acc + wave * NP, dct + wave * NP, twf, lane, Q, ninv, wlast, P.mu32, x);
        u32* out = acc_out + ((size_t)blockIdx.x * 2 + wave) * N;
#pragma unroll
        for (int r = 0; r < E; ++r) out[((u32)r << 6) | lane] = x[r];
    }
}

// ---------------------------------------------------------------------------------------
// blind rotation, LATENCY variant (GINX, N = 1024, dG = 4, lazy arithmetic)
// ---------------------------------------------------------------------------------------
// A wave issues at most one VALU instruction every 5-9 cycles (measured, tools/clock_probe.hip), so the
// step time of a lone workgroup is set by how many of its waves work at once.  The throughput kernel above
// runs the two inverse transforms of a step on two waves (41% of a lone workgroup's step time) and exposes
// the key-row load latency twice per step.  This variant is for launches that leave CUs to themselves:
//   * each inverse transform is split over FOUR waves: 256 threads x 4 coefficients, five 2-stage passes
//     exchanged through two ping-pong LDS buffers whose layouts are bank-conflict free on both sides
//     with compile-time register offsets (e0: 320 p[9:8] + padded natural, e1: p[3:0] + 20 p[5:4] + 80 p[7:6] +
//     320 p[9:8], e2: p[5:0] + 80 p[7:6] + 320 p[9:8], e3: 320 p[9:8] + p[7:0]).  Every layout keeps the 256
//     positions of one wave (p[9:8] = wave of the group through passes 0..3) in that wave's own 320 words, and the
//     exchanges after passes 0, 1, 2 only swap register bits with LANE bits: they stay inside the wave and need no
//     workgroup barrier.  Only the exchange before the last pass (position bits 9:8 <-> wave) crosses waves: one
//     barrier per inverse transform, three per step (before pass 4, around the forward transforms);
//   * the thread <-> data mappings line up across phases, so two passes are fused away: the MAC thread owns
//     the 4 consecutive positions of inverse pass 0 and runs it on the words it has just accumulated; after the
//     inverse transform a thread holds coefficients t + 256 r, i.e. bits 9 and 8 in registers, and applies the
//     first two FORWARD stages to each digit there -- the forward transform is left with 8 stages = 2 passes;
//   * the 15 inverse twiddles of a thread do not depend on the step: they live in registers;
//   * all 16 key rows of a step are requested at the top of the step and the barriers inside a step
//     order LDS only, so the L2/HBM latency hides behind the transforms (64 VGPRs of loads in flight,
//     affordable at <= 2 workgroups per CU).
// Results are bit-identical to the throughput kernel (same arithmetic, different schedule).
__device__ __forceinline__ u32 xlay1(u32 p) { return (p & 15u) + 20u * ((p >> 4) & 3u) + 80u * ((p >> 6) & 3u) + 320u * (p >> 8); }
__device__ __forceinline__ u32 xlay2(u32 p) { return (p & 63u) + 80u * ((p >> 6) & 3u) + 320u * (p >> 8); }

// two Gentleman-Sande stages on 4 registers whose index is the 2-bit field (B0+1, B0) of the position:
// stage B0 pairs (0,1) [twiddle fa] and (2,3) [fb], stage B0+1 pairs (0,2), (1,3) [fc].  In < 2Q, out < 2Q.
__device__ __forceinline__ void inv_pass4(u32 (&x)[4], uint2 fa, uint2 fb, uint2 fc, u32 Q, u32 mu32) {
    const u32 Q2 = 2 * Q;
    const u32 a0 = x[0] + x[1];                                  // < 4Q
    const u32 a1 = mul_shoup_lazy3(x[1] + Q2 - x[0], fa, Q);     // < 2Q
    const u32 a2 = x[2] + x[3];
    const u32 a3 = mul_shoup_lazy3(x[3] + Q2 - x[2], fb, Q);
    const u32 b0 = a0 + a2;                                      // < 8Q
    x[2] = mul_shoup_lazy3(a2 + 2 * Q2 - a0, fc, Q);
    const u32 b1 = a1 + a3;                                      // < 4Q
    x[3] = mul_shoup_lazy3(a3 + Q2 - a1, fc, Q);
    x[0] = b0 - __umulhi(b0, mu32) * Q;                          // any 32-bit value -> [0, 2Q)
    x[1] = csub(b1, Q2);
}
// the last two stages (bits 8, 9) with N^-1 folded into the final one; out in [0, Q)
__device__ __forceinline__ void inv_pass4_last(u32 (&x)[4], uint2 fa, uint2 fb, uint2 ninv, uint2 wlast, u32 Q) {
    const u32 Q2 = 2 * Q;
    const u32 a0 = x[0] + x[1];
    const u32 a1 = mul_shoup_lazy3(x[1] + Q2 - x[0], fa, Q);
    const u32 a2 = x[2] + x[3];
    const u32 a3 = mul_shoup_lazy3(x[3] + Q2 - x[2], fb, Q);
    x[0] = csub(mul_shoup_lazy3(a0 + a2, ninv, Q), Q);
    x[2] = csub(mul_shoup_lazy3(a0 + 2 * Q2 - a2, wlast, Q), Q);
    x[1] = csub(mul_shoup_lazy3(a1 + a3, ninv, Q), Q);
    x[3] = csub(mul_shoup_lazy3(a1 + Q2 - a3, wlast, Q), Q);
}

// per-thread, step-invariant state of the split inverse transform.  REGTW: the thread's 14 inverse twiddles stay
// in registers for the whole bootstrap; otherwise only their table positions do and they are re-read per step.
template <bool REGTW>
struct SplitInv {
    u32 a0, x0, l1, s1, l2, s2, l3, s3, t;   // a0: the thread's 4 words in an accumulator row, x0: in exchange layout e0
    uint2 fa[REGTW ? 5 : 1], fb[REGTW ? 5 : 1], fc[REGTW ? 4 : 1];
    u32 ia[REGTW ? 1 : 4], ic[REGTW ? 1 : 4];  // positions of fa (fb sits at the entry with index - 1) and fc
};
template <bool REGTW>
__device__ __forceinline__ void split_inv_setup(SplitInv<REGTW>& S, const uint2* twf, u32 t) {
    S.t = t;
    S.a0 = phys(4 * t);
    const u32 wb = 48u * (t >> 6);               // e0 = phys + 48 per wave block: 272 w -> 320 w
    S.x0 = S.a0 + wb;
    const u32 pb1 = ((t >> 2) << 4) | (t & 3u), pb2 = ((t >> 4) << 6) | (t & 15u), pb3 = ((t >> 6) << 8) | (t & 63u);
    S.l1 = phys(pb1) + wb; S.s1 = xlay1(pb1);
    S.l2 = xlay1(pb2); S.s2 = xlay2(pb2);
    S.l3 = xlay2(pb3); S.s3 = 320u * (t >> 6) + (t & 63u);
    // stage B (block m = 2^(9-B)) uses -tw[m + (m-1-i)], i = position >> (B+1); the sign sits in the operand
    const u32 u0 = t, u1 = t >> 2, u2 = t >> 4, u3 = t >> 6;
    if constexpr (REGTW) {
        S.fa[0] = twf[tw_pos<512>(511 - 2 * u0)]; S.fb[0] = twf[tw_pos<512>(510 - 2 * u0)]; S.fc[0] = twf[tw_pos<256>(255 - u0)];
        S.fa[1] = twf[tw_pos<128>(127 - 2 * u1)]; S.fb[1] = twf[tw_pos<128>(126 - 2 * u1)]; S.fc[1] = twf[tw_pos<64>(63 - u1)];
        S.fa[2] = twf[tw_pos<32>(31 - 2 * u2)];   S.fb[2] = twf[tw_pos<32>(30 - 2 * u2)];   S.fc[2] = twf[tw_pos<16>(15 - u2)];
        S.fa[3] = twf[tw_pos<8>(7 - 2 * u3)];     S.fb[3] = twf[tw_pos<8>(6 - 2 * u3)];     S.fc[3] = twf[tw_pos<4>(3 - u3)];
        S.fa[4] = twf[tw_pos<2>(1)];              S.fb[4] = twf[tw_pos<2>(0)];
    } else {
        S.ia[0] = tw_pos<512>(511 - 2 * u0); S.ic[0] = tw_pos<256>(255 - u0);
        S.ia[1] = tw_pos<128>(127 - 2 * u1); S.ic[1] = tw_pos<64>(63 - u1);
        S.ia[2] = tw_pos<32>(31 - 2 * u2);   S.ic[2] = tw_pos<16>(15 - u2);
        S.ia[3] = tw_pos<8>(7 - 2 * u3);     S.ic[3] = tw_pos<4>(3 - u3);
    }
}
// twiddles of pass K (0..3) / of the last pass
template <bool REGTW, int K>
__device__ __forceinline__ void split_tw(const SplitInv<REGTW>& S, const uint2* twf, uint2& fa, uint2& fb, uint2& fc) {
    if constexpr (REGTW) { fa = S.fa[K]; fb = S.fb[K]; fc = S.fc[K]; }
    else {
        constexpr u32 M = 512u >> (2 * K);
        fa = twf[S.ia[K]];
        // entry i-1: one position lower in natural blocks; in transposed blocks (M >= 64, i odd -> i-1 even)
        // the low index bit is the high position bit: 64 positions lower
        fb = twf[M < 64 ? S.ia[K] - 1 : S.ia[K] - 64];
        fc = twf[S.ic[K]];
    }
}
// Pass 0 (stages on position bits 0, 1) of thread t's 4 consecutive evaluation-form values a[] (< 2Q): result
// into exchange buffer xa (layout e0).  The MAC thread of the same index owns exactly these 4 positions, so
// the kernel runs this on the freshly accumulated words instead of re-reading them.
template <bool REGTW>
__device__ __forceinline__ void split_pass0(const SplitInv<REGTW>& S, const uint2* twf, u32 (&a)[4], u32* xa, u32 Q, u32 mu32) {
    uint2 fa, fb, fc;
    split_tw<REGTW, 0>(S, twf, fa, fb, fc);
    inv_pass4(a, fa, fb, fc, Q, mu32);
    *reinterpret_cast<uint4*>(xa + S.x0) = make_uint4(a[0], a[1], a[2], a[3]);
}
// Passes 1..4: xa holds pass 0's output, written by this thread's own quad (no barrier needed in between); xa / xb
// are the ping-pong exchange buffers (1280 words each).  Leaves coefficient j = (r << 8) | t in x[r], in [0, Q).
// Contains ONE workgroup barrier (before the last pass); the other exchanges are wave-local.
// at_pass(integral_constant<int, k>) is called at the start of pass k = 1..4 (the kernel spreads its key-row
// requests over the passes with it: 16 back-to-back 1-KiB loads per wave would block on the memory queue).
template <bool REGTW, typename F>
__device__ __forceinline__ void split_inverse_rest(const SplitInv<REGTW>& S, const uint2* twf, u32* xa, u32* xb, u32 Q, u32 mu32,
                                                   uint2 ninv, uint2 wlast, u32 (&x)[4], F&& at_pass) {
    uint2 fa, fb, fc;
    wave_sync();   // pass 0's stores (this thread's quad) before the loads below
    at_pass(std::integral_constant<int, 1>{});
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = xa[S.l1 + 4 * r];
    split_tw<REGTW, 1>(S, twf, fa, fb, fc);
    inv_pass4(x, fa, fb, fc, Q, mu32);
#pragma unroll
    for (int r = 0; r < 4; ++r) xb[S.s1 + 4 * r] = x[r];
    wave_local_sync();
    at_pass(std::integral_constant<int, 2>{});
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = xb[S.l2 + 20 * r];
    split_tw<REGTW, 2>(S, twf, fa, fb, fc);
    inv_pass4(x, fa, fb, fc, Q, mu32);
#pragma unroll
    for (int r = 0; r < 4; ++r) xa[S.s2 + 16 * r] = x[r];
    wave_local_sync();
    at_pass(std::integral_constant<int, 3>{});
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = xa[S.l3 + 80 * r];
    split_tw<REGTW, 3>(S, twf, fa, fb, fc);
    inv_pass4(x, fa, fb, fc, Q, mu32);
#pragma unroll
    for (int r = 0; r < 4; ++r) xb[S.s3 + 64 * r] = x[r];
    block_sync_lds();
    at_pass(std::integral_constant<int, 4>{});
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = xb[S.t + 320 * r];
    if constexpr (REGTW) inv_pass4_last(x, S.fa[4], S.fb[4], ninv, wlast, Q);
    else inv_pass4_last(x, twf[tw_pos<2>(1)], twf[tw_pos<2>(0)], ninv, wlast, Q);
}

// AP = true: AP/DM accumulator -- one step per non-zero base-baseR digit of -a_i, a single RGSW key selected by the
// digit, the product REPLACES the accumulator (no monomials); everything else is shared with GINX.
// FUSE: the tail of EvalBinGate (extract, ModSwitch, KeySwitch, ModSwitch) runs in this kernel's epilogue (fused_tail).
// FOLD: the lowest gadget digit is never transformed.  SignedDigitDecompose is exact for these parameters (the host
//   checks it: sum_l B^l dct_l = acc as integers), so NTT(dct_0) = ACC - sum_{l>=1} B^l NTT(dct_l) with ACC the
//   evaluation-form accumulator the kernel already holds, and
//       sum_l NTT(dct_l) ek_l  =  ACC ek_0 + sum_{l>=1} NTT(dct_l) (ek_l - B^l ek_0)      (mod Q, exactly).
//   The key arrives with its rows l >= 1 already replaced by ek_l - B^l ek_0 (engine.cpp, k_fold_gadget); the MAC reads
//   the accumulator rows where the digit-0 rows used to be: 6 forward transforms per step instead of 8, same
//   accumulator words.  The evaluation-form accumulator is double-buffered between `acc` and the two digit rows that
//   became free (other threads read a component's words as a MAC row while its owner writes the new ones).
// lat_bootstrap: ONE gate bootstrap by the calling workgroup (all 128 DG threads), the body shared by the per-frontier
// kernel k_blind_rotate_lat and the dependency-driven persistent kernel k_bootstrap_dag.  g: the gate, soff: slot offset
// of the instance, boot: index of this bootstrap in acc_out / the debug buffers (acc_out may be null).
// WPS = waves per SIMD the register budget allows: 2 (one workgroup per CU) or 4 (two)
template <int DG, int WPS, bool AP, bool FUSE, bool FOLD, bool PERSIST = false, typename PT>
__device__ __forceinline__ void lat_bootstrap(const PT& P, const bce_gate_desc g, const u32 soff, const u32 boot, u32* smem,
                                              u32* __restrict__ acc_out, u32* __restrict__ dbg_lweN, u32* __restrict__ dbg_ks) {
    static_assert(DG == 4, "the split inverse transform is laid out for 8 waves");
    constexpr int LOGN = 10;
    using C = Cfg<LOGN>;
    constexpr int N = C::N, NP = C::NP;
    constexpr u32 R = 2 * DG, T = 64 * R, XB = 1280;
    uint2* twf = reinterpret_cast<uint2*>(smem);
    u32* acc = reinterpret_cast<u32*>(twf + N);  // [2][NP]
    u32* dct = acc + 2 * NP;                     // [R][NP]
    u32* xab = dct + R * NP;                     // [2 polynomials][2 buffers][XB]
    u32* av = xab + 4 * XB;

    u32 tid_ = threadIdx.x;
    // opaque inside the persistent kernel's loop: nothing derived from the thread index is hoisted out of the bootstrap
    // (and then kept alive, i.e. spilled, across it)
    if constexpr (PERSIST) asm volatile("" : "+v"(tid_));
    const u32 tid = tid_;
    const u32 lane = tid & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 Q = P.Q, q = P.q, qm = q - 1, n = P.n;
    if constexpr (WPS >= 4 && !PERSIST) {
        // XCD start gate of a multi-round launch (round 4): the workgroups that take over an XCD's slots as the previous round
        // retires start their bootstraps together, so that they walk the key in lock-step (one L2 fill per row and XCD) like
        // the first round does.  Without it the rounds of a launch drift apart and re-fetch the key from the Infinity Cache:
        // FETCH_SIZE of a 6,144-bootstrap launch 45-54 GB -> 19.3 GB (= 6 x the 3.2 GB of a two-round launch), launch time
        // -0.8 % (profiles/r04_gate_ab.log).  Wave 0 waits, wave-uniform, at most xcd_gate_ticks; the others meet it at the
        // prologue's barrier.
        if (P.xcd_gate != nullptr && wave == 0) {
            const u32 xcc = __builtin_amdgcn_s_getreg((31u << 11) | 20u) & 15u;
            u32* const gate = P.xcd_gate + 32u * xcc;
            const u32 slots = max(2u, 2u * (P.cu_count >> 3));              // resident workgroups of one XCD
            const u32 mine = (gridDim.x + 7u - xcc) >> 3;                   // this launch's workgroups on this XCD (round-robin dispatch)
            const u32 t = u_add(gate, 1u);
            const u32 full = (t / slots + 1u) * slots, target = full < mine ? full : mine;
            const u64 t0 = __builtin_amdgcn_s_memrealtime();
            while (u_ld(gate) < target && __builtin_amdgcn_s_memrealtime() - t0 < P.xcd_gate_ticks) __builtin_amdgcn_s_sleep(8);
        }
    }
    bootstrap_prologue<LOGN, true, T>(P, g, soff, twf, acc, av, tid, lane, wave);

    const u32 c = wave >> 2;  // this wave's inverse-transform group = accumulator component
    constexpr bool REGTW = WPS <= 2;
    SplitInv<REGTW> S;
    split_inv_setup(S, twf, tid & 255u);
    u32* const accc = acc + c * NP;
    u32* const xa = xab + c * 2 * XB;
    u32* const xb = xa + XB;

    const uint2 ninv = make_uint2(P.Ninv, P.Ninv_s), wlast = make_uint2(P.Winv_last, P.Winv_last_s);
    constexpr u32 rgsw = R * 2 * N;
    // GINX key: n * 2 RGSW ciphertexts; AP key: n * baseR * dR of them (< 2^31 bytes for every 32-bit parameter set)
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(P.bsk), 0, AP ? 0x7FFFFFFF : (int)(n * 2 * rgsw * 4), 0x00020000);
#ifdef BCE_BARRETT_TAIL   // development: round 3's fold + Barrett tail, for the same-box A/B (tools/tail_ab.sh)
    const __amdgpu_buffer_rsrc_t psi_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(P.psi_tab), 0, N * 4, 0x00020000);
#else
    const __amdgpu_buffer_rsrc_t psi_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(P.psi_tab_r2), 0, N * 4, 0x00020000);
#endif
    // digit extraction constants (see the throughput kernel)
    const u32 gb = P.gBits, Qh = Q >> 1;
    u32 off = 0;
    for (u32 l = 0; l < (u32)DG; ++l) off |= 1u << (l * gb + gb - 1);
    const u32 offm = off - Q, bias = Q - (1u << (gb - 1));
    // MAC item of this thread: component mc, positions mp0..mp0+3 (T == 2 * N/4: exactly one item each)
    const u32 mc = tid / (N / 4), mp0 = (tid % (N / 4)) * 4, mpp = phys(mp0);
    const u32 dig0 = phys(S.t);  // digit destination of register r: dig0 + 272 r  (phys(t + 256 r))
    // forward twiddles of the stages on bits 9 and 8 (wave-uniform): applied to the digits in registers
    const uint2 w9 = twf[1], w8a = twf[2], w8b = twf[3];
    {   // pass 0 of the first inverse transform (afterwards the MAC tail produces it)
        const uint4 v = *reinterpret_cast<const uint4*>(accc + S.a0);
        u32 a[4] = {v.x, v.y, v.z, v.w};
        split_pass0(S, twf, a, xa, Q, P.mu32);
    }

    BCE_PROF_INIT();
    const u32 nsteps = AP ? n * P.dR : n;
    // FOLD: word offset (from `acc`) of the evaluation-form accumulator the step reads; the step writes the other one
    // (offset 2 NP = digit rows 0, 1).
    u32 cb = 0;
    if constexpr (WPS >= 4) __builtin_amdgcn_s_setprio(2);
    for (u32 step = 0; step < nsteps; ++step) {
        u32 ap = 0, rowb;
        if constexpr (!AP) {
            ap = ((q - av[step]) & qm) * P.factor;
            if (ap == 0) continue;  // acc unchanged: xa still holds its pass 0
            rowb = step * (2 * rgsw * 4);
        } else {
            const u32 i = step / P.dR, kd = step - i * P.dR;
            u32 aI = (q - av[i]) & qm;
            for (u32 t = 0; t < kd; ++t) aI /= P.baseR;
            const u32 a0 = aI % P.baseR;
            if (a0 == 0) continue;      // rgsw-acc-dm.cpp EvalAcc: digit 0 is skipped
            rowb = (((i * P.baseR + a0) * P.dR + kd) * rgsw) * 4u;
        }
        // key rows of this step, requested during phase 1 (a quarter at the start of each inverse pass),
        // consumed in phase 3: all 16 with the 256-register budget, the first half of each key with the
        // 128-register one (the rest is requested at the end of phase 2)
        constexpr u32 PR = (WPS <= 2) ? R : R / 2;
        uint4 kA[R], kB[R];
        auto request_rows = [&](auto kc) {
            constexpr u32 k = decltype(kc)::value - 1, G = PR / 4;  // pass k+1 requests rows [k*G, (k+1)*G) of each key
            BCE_PROF_MARK(8 + k);   // 8: step head, 9: pass 1, 10: pass 2, 11: pass 3 + the barrier; slot 0 is then pass 4 + digits
#pragma unroll
            for (u32 l = k * G; l < (k + 1) * G; ++l) {
                kA[l] = bsk_row(rsrc, tid * 16u, rowb + l * (2 * N * 4));
                if constexpr (!AP) kB[l] = bsk_row(rsrc, tid * 16u, rowb + (rgsw + l * 2 * N) * 4);
            }
        };
        // (1) inverse transforms on all 8 waves (passes 1..4), SignedDigitDecompose in closed form, and the
        //     first two forward stages (bits 9, 8 = this thread's 4 registers) on each digit -> dct[2l + c]
        {
            u32 x[4], u[4];
            split_inverse_rest(S, twf, xa, xb, Q, P.mu32, ninv, wlast, x, request_rows);
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = x[r] + ((x[r] < Qh) ? off : offm);
#pragma unroll
            for (u32 l = FOLD ? 1 : 0; l < (u32)DG; ++l) {
                u32 v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_ubfe(u[r], l * gb, gb) + bias;
                fwd_bfly(v[0], v[2], w9, Q);
                fwd_bfly(v[1], v[3], w9, Q);
                fwd_bfly(v[0], v[1], w8a, Q);
                fwd_bfly(v[2], v[3], w8b, Q);
#pragma unroll
                for (int r = 0; r < 4; ++r) dct[(2 * l + c) * NP + dig0 + 272 * r] = v[r];
            }
        }
        BCE_PROF_MARK(0);
        block_sync_lds();
        BCE_PROF_MARK(1);
        // (2) one wave per decomposed polynomial: the remaining 8 forward stages, in place
        // two workgroups per CU: the multiplier-bound forward transforms run at LOW wave priority, so the other
        // workgroup's latency-bound phases (inverse passes, MAC tail) get their issue slots first and the transform
        // waves fill the gaps (+1..4 %, same-box A/B; the opposite policy costs 7 %, and a lone workgroup loses 6 %
        // with either, hence only in this build)
        // (folded key: the half-row waves 4..7 -- the younger ones, which the arbiter serves last -- one level above the
        // whole-row waves, so that both kinds finish the phase together: -2 % per saturated launch, profiles/r02_prio_ab.log)
        // (round 4, profiles/r04_fwd_prio_ab.log: the whole-row waves above the half-row ones, or both at 1 -- within 0.3 % of this)
        if constexpr (WPS >= 4) { if (FOLD && wave >= 4) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#ifndef BCE_SKIP_FWD   // development: -DBCE_SKIP_FWD / -DBCE_SKIP_MAC leave a phase's LDS traffic out (wrong results) to
                       // attribute the LDS counters to phases, tools/lds_attribution.sh
        if constexpr (FOLD) {
            // six rows (2..7) on eight waves: whole rows 2..5 on waves 0..3, rows 6 and 7 as halves on waves 4..7
            if (wave < 4) ntt_forward_wave_low8(dct + (2 + wave) * NP, twf, lane, Q);
            else ntt_forward_half_low8(dct + (6 + ((wave - 4) >> 1)) * NP, (wave - 4) & 1u, twf, lane, Q);
        } else {
            ntt_forward_wave_low8(dct + wave * NP, twf, lane, Q);
        }
#endif
        if constexpr (WPS >= 4) __builtin_amdgcn_s_setprio(2);
        BCE_PROF_MARK(2);
        if constexpr (PR < R) {
            // 128-register build: the transform's registers are free again -- request the second half of the
            // key rows before waiting for the other waves (the barrier orders LDS only)
#pragma unroll
            for (u32 l = PR; l < R; ++l) kA[l] = bsk_row(rsrc, tid * 16u, rowb + l * (2 * N * 4));
#pragma unroll
            for (u32 l = PR; l < R; ++l) if constexpr (!AP) kB[l] = bsk_row(rsrc, tid * 16u, rowb + (rgsw + l * 2 * N) * 4);
        }
        block_sync_lds();
        BCE_PROF_MARK(3);
        // (3) RGSW multiply-accumulate, then pass 0 of the next inverse transform on the new words
        {
            u64 sp[4] = {0, 0, 0, 0}, sn[4] = {0, 0, 0, 0};
#pragma unroll
            for (u32 l = 0; l < R; ++l) {
                // FOLD: rows 0, 1 are the accumulator components themselves (< 2Q)
#ifdef BCE_SKIP_MAC
                const uint4 d = make_uint4(l + tid, l, tid, 1u);
#else
                const uint4 d = *reinterpret_cast<const uint4*>((FOLD && l < 2 ? acc + cb : dct) + l * NP + mpp);
#endif
                sp[0] += (u64)d.x * kA[l].x; sp[1] += (u64)d.y * kA[l].y; sp[2] += (u64)d.z * kA[l].z; sp[3] += (u64)d.w * kA[l].w;
                if constexpr (!AP) {
                    sn[0] += (u64)d.x * kB[l].x; sn[1] += (u64)d.y * kB[l].y; sn[2] += (u64)d.z * kB[l].z; sn[3] += (u64)d.w * kB[l].w;
                }
            }
            u32 anew[4];
            const u32* const accr = acc + cb + mc * NP + mpp;
            u32* const accw = acc + (FOLD ? 2 * NP - cb : 0) + mc * NP + mpp;
            if constexpr (AP) {
                // acc[c] = sum_l dct[l] * ek[l][c]   (rgsw-acc-dm.cpp AddToAcc: the product REPLACES acc)
#pragma unroll
                for (int e = 0; e < 4; ++e) anew[e] = barrett_fold(sp[e], P.c32, Q, P.red_shift, P.red_mu);
                *reinterpret_cast<uint4*>(accw) = make_uint4(anew[0], anew[1], anew[2], anew[3]);
            } else {
                const u32 a4 = ap & 3u;
                const uint2 Ia = make_uint2(P.I4[a4], P.I4s[a4]);
                const uint2 Ina = make_uint2(P.I4[(4u - a4) & 3u], P.I4s[(4u - a4) & 3u]);
#ifdef BCE_BARRETT_TAIL
                ginx_mac_tail<LOGN, true>(P, psi_rsrc, Q, ap, Ia, Ina, mp0, accr, accw, sp, sn, anew);
#else
                ginx_mac_tail_redc<LOGN>(P, psi_rsrc, Q, ap, Ia, Ina, mp0, accr, accw, sp, sn, anew);
#endif
            }
            if constexpr (FOLD) cb = 2 * NP - cb;
            split_pass0(S, twf, anew, xa, Q, P.mu32);  // mc == c, mp0 == 4 t: this thread's pass-0 registers
        }
        BCE_PROF_MARK(4);
        // no barrier here: the next step's passes 1..3 touch only this wave's own words of xa / xb, and its first
        // cross-wave access (pass 4, then the digit rows) sits behind the barrier inside split_inverse_rest, which
        // every wave reaches only after its MAC reads of the digit rows
        wave_local_sync();
        BCE_PROF_MARK(5);
    }
    BCE_PROF_FLUSH();
    // accumulator back to COEFFICIENT form for the extraction kernel (its pass 0 is already in xa)
    {
        u32 x[4];
        split_inverse_rest(S, twf, xa, xb, Q, P.mu32, ninv, wlast, x, [](auto) {});
        if (acc_out) {
            u32* out = acc_out + ((size_t)boot * 2 + c) * N + S.t;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[256 * r] = x[r];
        }
        if constexpr (FUSE) {
            // coefficient-form accumulator into LDS (the evaluation-form copy in `acc` is dead: its pass 0 went to xa)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c * N + S.t + 256 * r] = x[r];
        }
    }
    if constexpr (FUSE) {
        if constexpr (WPS >= 4) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        // rowidx and the partial sums live in the digit rows + exchange buffers (54 KiB, all dead now)
        u32* rowidx = dct;
        u64* red = reinterpret_cast<u64*>(dct + ((N * P.dKS + 3) & ~3u));
        u32* outp = P.pool + (size_t)(g.out + soff) * P.pool_stride;
        if (P.ksk_u16) fused_tail<uint16_t, T>(P, acc, rowidx, red, outp, boot, dbg_lweN, dbg_ks);
        else fused_tail<u32, T>(P, acc, rowidx, red, outp, boot, dbg_lweN, dbg_ks);
    }
}

template <int DG, int WPS, bool AP = false, bool FUSE = false, bool FOLD = false>
__global__ __launch_bounds__(128 * DG, WPS) void k_blind_rotate_lat(DevParams P, const bce_gate_desc* __restrict__ descs, u32 n_desc,
                                                                   u32 slot_stride, u32* __restrict__ acc_out,
                                                                   u32* __restrict__ dbg_lweN, u32* __restrict__ dbg_ks) {
    extern __shared__ __align__(16) u32 smem[];
    lat_bootstrap<DG, WPS, AP, FUSE, FOLD>(P, descs[blockIdx.x % n_desc], (blockIdx.x / n_desc) * slot_stride, blockIdx.x, smem,
                                           acc_out, dbg_lweN, dbg_ks);
}

// LDS the fused tail needs inside the digit rows + exchange buffers of k_blind_rotate_lat (T = 512 threads)
bool fused_tail_fits(const DevParams& P) {
    const size_t N = P.N, NP = N + (N >> 6) * 4, R = 2 * P.dG;
    const size_t VW = P.ksk_u16 ? 8 : 4, G = (P.n + VW) / VW, Gv = G < 512 ? G : 512, RW = (Gv + 63) / 64, SL = 8 / RW;
    const size_t need = ((N * P.dKS + 3) & ~(size_t)3) * 4 + SL * Gv * VW * 8;
    return (size_t)P.n + 1 <= 512 * VW && RW <= 8 && need <= (R * NP + 4 * 1280) * 4;
}

size_t blind_rotate_lat_lds_bytes(const DevParams& P) {
    const size_t N = P.N, NP = N + (N >> 6) * 4, R = 2 * P.dG;
    return (2 * N + (2 + R) * NP + 4 * 1280 + ((P.n + 1 + 3) & ~3u)) * sizeof(u32);
}

size_t blind_rotate_lds_bytes(const DevParams& P) {
    const size_t N = P.N, NP = N + (N >> 6) * 4, R = 2 * P.dG;
    return (2 * N + (2 + R) * NP + ((P.n + 1 + 3) & ~3u)) * sizeof(u32);
}

// ---------------------------------------------------------------------------------------
// dependency-driven persistent kernel: the whole bootstrap DAG in ONE launch
// ---------------------------------------------------------------------------------------
// What the reference does between two gate evaluations -- Circuit::_ManageGates marks the gates whose inputs have all
// arrived as ready (src/circuit.cpp:575-683), Circuit::_ExecuteGates runs the ready ones in a parallel region and
// retires them (src/circuit.cpp:685-817) -- happens here on the device, per bootstrap instead of per frontier:
//   * dep[instance][task] counts the producers of a task that have not finished (0..2, from the slot numbers);
//   * the workgroup that finished a bootstrap (its refreshed ciphertext is in the pool: fused tail) releases its stores at
//     agent scope, decrements its consumers' counters with agent-scope atomics and pushes every consumer that reached
//     zero to the ready queue of its priority class: idx = tail++ ; slots[idx] = item + 1;
//   * an idle workgroup polls the queues' head / tail words (relaxed agent-scope loads + s_sleep, ONE lane), takes a
//     ticket on a class with a backlog (fetch-add on `head`), waits for that entry if need be, performs ONE agent-scope
//     acquire (this CU's L1 holds no fresh copy of other CUs' stores) and runs the bootstrap.  Queue entries are written
//     once per evaluation and never wrap.
// Hand-off protocol = the plain-payload / release-fence / relaxed-flag form of the MI355X guide (Guideline 16): every
// storing wave drains (vmcnt(0)), workgroup barrier, one wave fences + drains, then the atomics; the consumer polls
// relaxed, fences once, drains, workgroup barrier, then plain loads.
// Progress: a workgroup never waits while it owns something another one needs (it either runs a bootstrap, which ends,
// or polls), so the launch completes for any residency; the launcher still sizes the grid to the resident count so that
// no workgroup sits unscheduled.  Every spin is bounded: no completion anywhere for stall_ticks sets the abort word,
// everyone leaves, the host reports BCE_ERR_STATE.
// Placement (policy bit 0): one bootstrap alone on a CU takes 2.0-2.5 ms, two sharing it ~3.1 ms each, so a narrow
// frontier should spread over CUs first.  A workgroup knows its CU (HW_ID / XCC_ID) and how many workgroups on it are
// running (cu_busy[key]); the control line counts the CUs on which none is (idle_cus).  The first workgroup of an idle CU
// claims at once; any other one only when the backlog exceeds what the idle CUs will take (seen on two polls in a row),
// or after it has watched the same queue head for lazy_ticks.
__global__ void k_dag_rearm(DagParams D) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t items = (size_t)D.n_tasks * D.instances;
    for (size_t i = i0; i < items; i += stride) D.dep[i] = D.dep_init[i % D.n_tasks];
    for (u32 q = 0; q < kDagQueues; ++q) {
        const u32 ninit = (D.init_off[q + 1] - D.init_off[q]) * D.instances;
        for (size_t j = i0; j < D.qcap[q]; j += stride) {
            u32 v = 0;
            if (j < ninit) {   // initially ready: task-major, the instances of one task next to each other
                const u32 ti = (u32)(j / D.instances), k = (u32)(j % D.instances);
                v = k * D.n_tasks + D.init_items[D.init_off[q] + ti] + 1u;
            }
            D.slots[q][j] = v;
        }
        if (i0 == 0) { D.ctl[q] = 0; D.ctl[kDagQueues + q] = ninit; D.ctl[kDagIdleCus] = 0; }
    }
    for (size_t j = i0 + kDagAbort; j < kDagCtlWords; j += stride) D.ctl[j] = 0;
}

template <int WPS, bool AP, bool FOLD>
__global__ __launch_bounds__(512, WPS) void k_bootstrap_dag(const DevParams* Pp, const DagParams* Dp) {
    extern __shared__ __align__(16) u32 smem[];
    // the worker's mailbox sits in front of the LDS layout of lat_bootstrap
    dag_worker(Dp, smem, [&](ConstDagParams& D, u32 t, u32 k) {
        lat_bootstrap<4, WPS, AP, true, FOLD, true>(*as_constant<ConstDevParams>(Pp), D.tasks[t], D.slot_base + k * D.slot_stride, 0,
                                                   smem + kDagMailboxWords, nullptr, nullptr, nullptr);
    });
}

bool dag_kernel_available(const DevParams& P) {
    if (P.is64) return dag64_kernel_available(P);
    return !P.is64 && P.variant != 1 && P.logN == 10 && P.dG == 4 && P.lazy && fused_tail_fits(P);
}

hipError_t launch_dag_rearm(const DagParams& D, hipStream_t s) {
    hipLaunchKernelGGL(k_dag_rearm, dim3(1024), dim3(256), 0, s, D);
    return hipGetLastError();
}

hipError_t launch_bootstrap_dag(const DevParams& P, const DevParams* d_P, const DagParams* d_params, int wps, u32 grid, hipStream_t s) {
    if (!dag_kernel_available(P)) return hipErrorInvalidValue;
    if (P.is64) return launch_bootstrap_dag64(P, d_P, d_params, grid, s);
    using DagKernel = void (*)(const DevParams*, const DagParams*);
    const bool ap = P.method_ap != 0, x1 = wps <= 2;
    DagKernel k;
    if (P.fold) k = ap ? (x1 ? k_bootstrap_dag<2, true, true> : k_bootstrap_dag<4, true, true>) : (x1 ? k_bootstrap_dag<2, false, true> : k_bootstrap_dag<4, false, true>);
    else k = ap ? (x1 ? k_bootstrap_dag<2, true, false> : k_bootstrap_dag<4, true, false>) : (x1 ? k_bootstrap_dag<2, false, false> : k_bootstrap_dag<4, false, false>);
    const size_t lds = blind_rotate_lat_lds_bytes(P) + kDagMailboxWords * 4;   // + the worker's mailbox in front
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, d_P, d_params);
    return hipGetLastError();
}


namespace {
using BrKernel = void (*)(DevParams, const bce_gate_desc*, u32, u32, u32*);
template <int LOGN, int DG, bool AP>
BrKernel pick_br(bool lazy, int occ) {
    if (lazy) return occ >= 3 ? k_blind_rotate<LOGN, DG, true, 3, AP> : k_blind_rotate<LOGN, DG, true, 2, AP>;
    return k_blind_rotate<LOGN, DG, false, 2, AP>;
}
template <int LOGN>
BrKernel pick_br_dg(u32 dG, bool lazy, int occ, bool ap) {
    // instantiated gadget sizes: every OpenFHE parameter set with Q < 2^28 has dG in {3, 4}
    switch (dG) {
        case 3: return ap ? pick_br<LOGN, 3, true>(lazy, occ) : pick_br<LOGN, 3, false>(lazy, occ);
        case 4: return ap ? pick_br<LOGN, 4, true>(lazy, occ) : pick_br<LOGN, 4, false>(lazy, occ);
        default: return nullptr;
    }
}
}  // namespace

hipError_t launch_blind_rotate(const DevParams& P, const bce_gate_desc* d, u32 n_desc, u32 instances, u32 slot_stride,
                               u32* acc_out, hipStream_t s, int* kernel_id, u32* dbg_lweN, u32* dbg_ks, bool* tail_fused, LaunchEvents ev) {
    if (kernel_id) *kernel_id = BCE_BR_WAVE_PER_TRANSFORM;
    if (tail_fused) *tail_fused = false;
    const u32 R = 2 * P.dG;
    const dim3 grid(n_desc * instances), block(64 * R);
    const size_t lds = blind_rotate_lds_bytes(P);
    const int occ = P.occupancy_target;
    const bool ap = P.method_ap != 0;
    BrKernel kern = nullptr;
    if (P.fold && !(P.variant != 1 && P.logN == 10 && P.dG == 4 && P.lazy)) return hipErrorInvalidValue;  // folded key, no kernel for it
    if (P.variant != 1 && P.logN == 10 && P.dG == 4 && P.lazy) {
        // N = 1024, dG = 4 (STD128 class): the split-transform kernel, with the 256-register budget while the
        // launch leaves every workgroup a CU of its own, else with the 128-register one (two per CU);
        // measured against the one-wave-per-transform kernel over launch sizes 64..6144: tools/kernel_sweep.py
        using LatKernel = void (*)(DevParams, const bce_gate_desc*, u32, u32, u32*, u32*, u32*);
        const size_t lds_lat = blind_rotate_lat_lds_bytes(P);
        const bool alone = (P.variant == 2) || (P.variant == 0 && grid.x <= P.cu_count);
        const bool x1 = alone && P.variant != 3;
        // saturated launches run the tail of EvalBinGate in the kernel's epilogue (fused_tail); a launch that leaves
        // CUs to themselves keeps the separate tail kernels, which spread one bootstrap's row gather over many CUs
        const bool fuse = !x1 && tail_fused && P.fuse_tail && fused_tail_fits(P);
        LatKernel lk;
        if (P.fold) {   // key rows l >= 1 hold ek_l - B^l ek_0 (see the kernel's FOLD note)
            if (ap) lk = x1 ? k_blind_rotate_lat<4, 2, true, false, true> : (fuse ? k_blind_rotate_lat<4, 4, true, true, true> : k_blind_rotate_lat<4, 4, true, false, true>);
            else lk = x1 ? k_blind_rotate_lat<4, 2, false, false, true> : (fuse ? k_blind_rotate_lat<4, 4, false, true, true> : k_blind_rotate_lat<4, 4, false, false, true>);
        } else if (ap) lk = x1 ? k_blind_rotate_lat<4, 2, true, false> : (fuse ? k_blind_rotate_lat<4, 4, true, true> : k_blind_rotate_lat<4, 4, true, false>);
        else lk = x1 ? k_blind_rotate_lat<4, 2, false, false> : (fuse ? k_blind_rotate_lat<4, 4, false, true> : k_blind_rotate_lat<4, 4, false, false>);
        if (kernel_id) *kernel_id = x1 ? BCE_BR_SPLIT_X1 : BCE_BR_SPLIT_X2;
        if (tail_fused) *tail_fused = fuse;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lat);
        if (e != hipSuccess) return e;
        DevParams Pl = P;
        if (x1 || grid.x <= 2 * P.cu_count) Pl.xcd_gate = nullptr;            // one round: every workgroup starts at once anyway
        if (Pl.xcd_gate && (e = hipMemsetAsync(Pl.xcd_gate, 0, 16 * 32 * sizeof(u32), s)) != hipSuccess) return e;
        if (ev.start || ev.stop) return launch_with_events(lk, grid, block, lds_lat, s, ev, Pl, d, n_desc, slot_stride, acc_out, fuse ? dbg_lweN : nullptr, fuse ? dbg_ks : nullptr);
        hipLaunchKernelGGL(lk, grid, block, lds_lat, s, Pl, d, n_desc, slot_stride, acc_out, fuse ? dbg_lweN : nullptr, fuse ? dbg_ks : nullptr);
        return hipGetLastError();
    }
    switch (P.logN) {
        case 9: kern = pick_br_dg<9>(P.dG, P.lazy != 0, occ, ap); break;
        case 10: kern = pick_br_dg<10>(P.dG, P.lazy != 0, occ, ap); break;
        case 11: kern = (P.dG == 3) ? pick_br_dg<11>(P.dG, P.lazy != 0, 2, ap) : nullptr; break;
        default: break;
    }
    if (!kern) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (ev.start || ev.stop) return launch_with_events(kern, grid, block, lds, s, ev, P, d, n_desc, slot_stride, acc_out);
    hipLaunchKernelGGL(kern, grid, block, lds, s, P, d, n_desc, slot_stride, acc_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// tail: extract + ModSwitch + KeySwitch + ModSwitch (one workgroup per bootstrap)
// ---------------------------------------------------------------------------------------

// The tail runs as two kernels.
//   k_tail_gather: grid = bootstraps x S.  Workgroup (boot, s) owns the coefficients i in [s N/S, (s+1) N/S):
//     transpose + ModSwitch(Q -> qKS) of those, digit decomposition to key-switch row numbers in LDS, then the
//     row gather  sum_rows K[row][0..n]  for its N dKS / S rows.  A lane reads 16 BYTES of a row per load (8 u16 or
//     4 u32 elements; the padded row stride keeps the last load in bounds), 64..256 lanes cover a row and the
//     remaining waves take other rows -- one element per lane (the first version) needed 8x the load
//     instructions and kept 8x fewer bytes in flight.  S > 1 spreads one bootstrap over several CUs when the
//     launch is small.  Partial sums (u64) go to `partial[boot][s][0..n]`.
//   k_tail_finish: grid = bootstraps.  Sums the S partials, reduces mod qKS, subtracts from (0, b) and applies
//     ModSwitch(qKS -> q) into the pool.
// KT = key-switch key element type (u16 when qKS <= 2^16), AW = accumulator word of the blind rotation.
template <typename KT, typename AW>
__global__ __launch_bounds__(256) void k_tail_gather(DevParams P, u32 S, const AW* __restrict__ acc_in,
                                                     u64* __restrict__ partial, u32* __restrict__ dbg_lweN) {
    extern __shared__ __align__(16) u32 smem[];
    constexpr u32 VW = 16 / sizeof(KT);  // elements per 16-byte load
    const u32 N = P.N, n = P.n, qKS = P.qKS, B = P.baseKS, D = P.dKS;
    const u64 Q = sizeof(AW) == 8 ? P.Q64 : (u64)P.Q;
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 boot = blockIdx.x / S, s = blockIdx.x - boot * S;
    const u32 ni = N / S, i0 = s * ni, LR = ni * D;  // coefficients / rows of this workgroup
    u32* rowidx = smem;                                   // [LR] row number (i*B + digit)*D + j
    u64* red = reinterpret_cast<u64*>(smem + ((LR + 3) & ~3u));  // [slices][Gv*VW] or [256] (scalar pass)
    const AW* a0 = acc_in + (size_t)boot * 2 * N;

    // Transpose (X -> X^-1) of acc[0]: a'_0 = a_0, a'_{N-i} = -a_i ; then ModSwitch(Q -> qKS)
    for (u32 ii = tid; ii < ni; ii += 256) {
        const u32 i = i0 + ii;
        const u64 src = (i == 0) ? a0[0] : a0[N - i];
        const u64 v = (i == 0) ? src : (src ? Q - src : 0);
        u32 at = round_qQ(v, qKS, Q);
        if (dbg_lweN) dbg_lweN[(size_t)boot * (N + 1) + i] = at;
        for (u32 j = 0; j < D; ++j) {
            rowidx[ii * D + j] = (i * B + at % B) * D + j;
            at /= B;
        }
    }
    __syncthreads();

    const KT* __restrict__ ksk = reinterpret_cast<const KT*>(P.ksk);
    const u32 G = (n + VW) / VW;                 // 16-byte groups holding elements 0..n
    const u32 Gv = G < 256 ? G : 256;            // groups of the vector pass (one per lane of up to 4 waves)
    const u32 RW = (Gv + 63) / 64, SL = 4 / RW;  // waves per row, row slices per workgroup
    const u32 slice = wave / RW, group = (wave - slice * RW) * 64 + lane;
    u64 tot[VW];
#pragma unroll
    for (u32 e = 0; e < VW; ++e) tot[e] = 0;
    if (slice < SL && group < Gv) {
        // u32 running sums are folded into the u64 totals every CH rows (CH * qKS <= 2^32: no wrap)
        const u32 CH = P.ks_chunk;
        constexpr u32 U = 8;  // rows in flight per lane
        u32 r = slice;
        while (r < LR) {
            u32 run[VW];
#pragma unroll
            for (u32 e = 0; e < VW; ++e) run[e] = 0;
            const u32 rend = (LR - r > CH * SL) ? r + CH * SL : LR;
            for (; r + (U - 1) * SL < rend; r += U * SL) {
                uint4 v[U];
#pragma unroll
                for (u32 u = 0; u < U; ++u) {
                    const u32 row = __builtin_amdgcn_readfirstlane(rowidx[r + u * SL]);
                    v[u] = reinterpret_cast<const uint4*>(ksk + (size_t)row * P.ksk_stride)[group];
                }
#pragma unroll
                for (u32 u = 0; u < U; ++u) {
                    if constexpr (sizeof(KT) == 2) {
                        run[0] += v[u].x & 0xFFFFu; run[1] += v[u].x >> 16; run[2] += v[u].y & 0xFFFFu; run[3] += v[u].y >> 16;
                        run[4] += v[u].z & 0xFFFFu; run[5] += v[u].z >> 16; run[6] += v[u].w & 0xFFFFu; run[7] += v[u].w >> 16;
                    } else {
                        run[0] += v[u].x; run[1] += v[u].y; run[2] += v[u].z; run[3] += v[u].w;
                    }
                }
            }
            for (; r < rend; r += SL) {
                const u32 row = __builtin_amdgcn_readfirstlane(rowidx[r]);
                const uint4 v = reinterpret_cast<const uint4*>(ksk + (size_t)row * P.ksk_stride)[group];
                if constexpr (sizeof(KT) == 2) {
                    run[0] += v.x & 0xFFFFu; run[1] += v.x >> 16; run[2] += v.y & 0xFFFFu; run[3] += v.y >> 16;
                    run[4] += v.z & 0xFFFFu; run[5] += v.z >> 16; run[6] += v.w & 0xFFFFu; run[7] += v.w >> 16;
                } else {
                    run[0] += v.x; run[1] += v.y; run[2] += v.z; run[3] += v.w;
                }
            }
#pragma unroll
            for (u32 e = 0; e < VW; ++e) tot[e] += run[e];
        }
#pragma unroll
        for (u32 e = 0; e < VW; ++e) red[(size_t)slice * Gv * VW + group * VW + e] = tot[e];
    }
    __syncthreads();
    u64* out = partial + ((size_t)boot * S + s) * (n + 1);
    for (u32 k = tid; k < Gv * VW && k <= n; k += 256) {
        u64 sum = 0;
        for (u32 sl = 0; sl < SL; ++sl) sum += red[(size_t)sl * Gv * VW + k];
        out[k] = sum;
    }
    // elements beyond the vector pass (n + 1 > 256 groups, e.g. the single b column of n = 1024 with u32 rows):
    // all threads split the rows, one element at a time
    for (u32 k = Gv * VW; k <= n; ++k) {
        __syncthreads();
        u64 sum = 0;
        for (u32 r = tid; r < LR; r += 256) sum += (u64)ksk[(size_t)rowidx[r] * P.ksk_stride + k];
        red[tid] = sum;
        __syncthreads();
        if (tid == 0) {
            u64 t = 0;
            for (u32 i = 0; i < 256; ++i) t += red[i];
            out[k] = t;
        }
    }
}

template <typename AW>
__global__ __launch_bounds__(256) void k_tail_finish(DevParams P, const bce_gate_desc* __restrict__ descs, u32 n_desc, u32 slot_stride,
                                                     u32 S, const AW* __restrict__ acc_in, const u64* __restrict__ partial,
                                                     u32* __restrict__ dbg_lweN, u32* __restrict__ dbg_ks) {
    const u32 N = P.N, n = P.n, qKS = P.qKS;
    const u64 Q = sizeof(AW) == 8 ? P.Q64 : (u64)P.Q;
    const u64 Q8p1 = sizeof(AW) == 8 ? P.Q8p1_64 : (u64)P.Q8p1;
    const u32 boot = blockIdx.x;
    const bce_gate_desc g = descs[boot % n_desc];
    u32* out = P.pool + (size_t)(g.out + (boot / n_desc) * slot_stride) * P.pool_stride;
    const u64* part = partial + (size_t)boot * S * (n + 1);
    // KeySwitch: a' = -sum_rows A[row], b' = b - sum_rows B[row]   (mod qKS), b = acc[1][0] + Q/8 + 1 mod-switched
    for (u32 k = threadIdx.x; k <= n; k += 256) {
        u64 sum = 0;
        for (u32 s = 0; s < S; ++s) sum += part[(size_t)s * (n + 1) + k];
        const u32 sm = (u32)(sum % qKS);
        u32 base = 0;
        if (k == n) {
            u64 b = (u64)acc_in[(size_t)boot * 2 * N + N] + Q8p1;
            b = b >= Q ? b - Q : b;
            base = round_qQ(b, qKS, Q);
            if (dbg_lweN) dbg_lweN[(size_t)boot * (N + 1) + N] = base;
        }
        const u32 v = base >= sm ? base - sm : base + qKS - sm;
        if (dbg_ks) dbg_ks[(size_t)boot * (n + 1) + k] = v;
        out[k] = round_qQ(v, P.q, qKS);  // ModSwitch(qKS -> q)
    }
}

u32 tail_split(const DevParams& P, u32 boots) {
    // spread a bootstrap's rows over S workgroups while the launch has fewer than ~4 workgroups per CU
    u32 S = 1;
    while (S < 16 && boots * S * 2 <= 4 * P.cu_count && P.N / (S * 2) >= 64) S *= 2;
    return S;
}
size_t tail_partial_words(const DevParams& P, u32 boots) { return (size_t)boots * tail_split(P, boots) * (P.n + 1); }

hipError_t launch_tail(const DevParams& P, const bce_gate_desc* d, u32 n_desc, u32 instances, u32 slot_stride,
                       const void* acc_in, u64* partial, u32* dbg_lweN, u32* dbg_ks, hipStream_t s, LaunchEvents ev) {
    const u32 boots = n_desc * instances, S = tail_split(P, boots);
    const dim3 grid(boots * S), block(256);
    const u32 VW = P.ksk_u16 ? 8 : 4, G = (P.n + VW) / VW, Gv = G < 256 ? G : 256, RW = (Gv + 63) / 64, SL = 4 / RW;
    const size_t LR = (size_t)P.N / S * P.dKS;
    const size_t red_words = std::max<size_t>((size_t)SL * Gv * VW, 256);
    const size_t lds = ((LR + 3) & ~(size_t)3) * sizeof(u32) + red_words * sizeof(u64);
    const LaunchEvents first{ev.start, nullptr}, last{nullptr, ev.stop};
    // the gather carries the start timestamp, the finish the stop (plain launches when none was asked for)
    auto go = [&](auto kern, dim3 g, size_t l, LaunchEvents e, auto... args) -> hipError_t {
        if (e.start || e.stop) return launch_with_events(kern, g, block, l, s, e, args...);
        hipLaunchKernelGGL(kern, g, block, l, s, args...);
        return hipGetLastError();
    };
    hipError_t rc;
    if (P.is64) {
        const u64* a = static_cast<const u64*>(acc_in);
        if (P.ksk_u16) rc = go(k_tail_gather<uint16_t, u64>, grid, lds, first, P, S, a, partial, dbg_lweN);
        else rc = go(k_tail_gather<u32, u64>, grid, lds, first, P, S, a, partial, dbg_lweN);
        if (rc != hipSuccess) return rc;
        return go(k_tail_finish<u64>, dim3(boots), 0, last, P, d, n_desc, slot_stride, S, a, partial, dbg_lweN, dbg_ks);
    }
    const u32* a = static_cast<const u32*>(acc_in);
    if (P.ksk_u16) rc = go(k_tail_gather<uint16_t, u32>, grid, lds, first, P, S, a, partial, dbg_lweN);
    else rc = go(k_tail_gather<u32, u32>, grid, lds, first, P, S, a, partial, dbg_lweN);
    if (rc != hipSuccess) return rc;
    return go(k_tail_finish<u32>, dim3(boots), 0, last, P, d, n_desc, slot_stride, S, a, partial, dbg_lweN, dbg_ks);
}

// ---------------------------------------------------------------------------------------
// EvalNOT / COPY
// ---------------------------------------------------------------------------------------
__global__ void k_lwe_unary(DevParams P, const bce_gate_desc* __restrict__ descs, u32 n_desc, u32 slot_stride) {
    const bce_gate_desc g = descs[blockIdx.x % n_desc];
    const u32 soff = (blockIdx.x / n_desc) * slot_stride;
    const u32* in = P.pool + (size_t)(g.in0 + soff) * P.pool_stride;
    u32* out = P.pool + (size_t)(g.out + soff) * P.pool_stride;
    const u32 qm = P.q - 1;
    const bool neg = (g.op == BCE_OP_NOT) != (g.neg0 != 0);
    for (u32 i = threadIdx.x; i <= P.n; i += blockDim.x) {
        u32 v = in[i];
        if (neg) v = ((i == P.n ? (P.q >> 2) : 0u) - v) & qm;
        out[i] = v;
    }
}

hipError_t launch_lwe_unary(const DevParams& P, const bce_gate_desc* d, u32 n_desc, u32 instances, u32 slot_stride,
                            hipStream_t s) {
    hipLaunchKernelGGL(k_lwe_unary, dim3(n_desc * instances), dim3(256), 0, s, P, d, n_desc, slot_stride);
    return hipGetLastError();
}

// pool rows <-> dense buffer [count][n+1] (multi-rank exchange); slot = descs[i].in0
__global__ void k_pool_pack(DevParams P, const bce_gate_desc* __restrict__ descs, u32* __restrict__ buf, int to_pool) {
    u32* row = P.pool + (size_t)descs[blockIdx.x].in0 * P.pool_stride;
    u32* b = buf + (size_t)blockIdx.x * (P.n + 1);
    for (u32 i = threadIdx.x; i <= P.n; i += blockDim.x) {
        if (to_pool) row[i] = b[i]; else b[i] = row[i];
    }
}

hipError_t launch_pool_pack(const DevParams& P, const bce_gate_desc* d, u32 count, u32* buf, int to_pool, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pool_pack, dim3(count), dim3(256), 0, s, P, d, buf, to_pool);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// batched NTT over global memory (key import / key generation / debug), one wave per poly
// ---------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(256) void k_ntt_batch(DevParams P, u32* __restrict__ polys, u32 count, int inverse) {
    using C = Cfg<LOGN>;
    constexpr int N = C::N, NP = C::NP, E = C::E;
    extern __shared__ __align__(16) u32 smem[];
    uint2* tw = reinterpret_cast<uint2*>(smem);
    u32* buf = reinterpret_cast<u32*>(tw + N);
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W = blockDim.x >> 6;
    for (u32 i = tid; i < (u32)N; i += blockDim.x) tw[i] = P.tw_f[i];
    __syncthreads();
    u32* mine = buf + wave * NP;
    for (u32 p = blockIdx.x * W + wave; p < count; p += gridDim.x * W) {
        u32* gp = polys + (size_t)p * N;
        for (int r = 0; r < E; ++r) {
            u32 j = ((u32)r << 6) | lane;
            mine[phys(j)] = gp[j];
        }
        wave_sync();
        if (!inverse) {
            ntt_forward_wave<LOGN, false>(mine, tw, lane, P.Q, 0);
            for (int r = 0; r < E; ++r) {
                u32 j = ((u32)r << 6) | lane;
                gp[j] = mine[phys(j)];
            }
        } else {
            u32 x[E];
            ntt_inverse_wave<LOGN>(mine, mine, tw, lane, P.Q, make_uint2(P.Ninv, P.Ninv_s),
                                   make_uint2(P.Winv_last, P.Winv_last_s), P.mu32, x);
#pragma unroll
            for (int r = 0; r < E; ++r) gp[((u32)r << 6) | lane] = x[r];
        }
        wave_sync();
    }
}

hipError_t launch_ntt_batch(const DevParams& P, u32* polys, u32 count, int inverse, hipStream_t s) {
    if (count == 0) return hipSuccess;
    const size_t N = P.N, NP = N + (N >> 6) * 4;
    const u32 W = 4;
    const size_t lds = (2 * N + W * NP) * sizeof(u32);
    u32 blocks = (count + W - 1) / W;
    if (blocks > 4096) blocks = 4096;
    void (*kern)(DevParams, u32*, u32, int) = nullptr;
    switch (P.logN) {
        case 9: kern = k_ntt_batch<9>; break;
        case 10: kern = k_ntt_batch<10>; break;
        case 11: kern = k_ntt_batch<11>; break;
        default: return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * W), lds, s, P, polys, count, inverse);
    return hipGetLastError();
}

__global__ void k_pointwise_mac(DevParams P, u32* __restrict__ b, const u32* __restrict__ a,
                                const u32* __restrict__ z, u32 count, u32 b_step) {
    const size_t total = (size_t)count * P.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const u32 k = (u32)(i & (P.N - 1));
        const size_t bi = (i >> P.logN) * b_step * P.N + k;
        b[bi] = barrett_reduce((u64)a[i] * z[k] + b[bi], P.Q, P.red_shift, P.red_mu);
    }
}

hipError_t launch_pointwise_mac(const DevParams& P, u32* b, const u32* a, const u32* z, u32 count, u32 b_step,
                                hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pointwise_mac, dim3(2048), dim3(256), 0, s, P, b, a, z, count, b_step);
    return hipGetLastError();
}

// Key layout of the FOLD kernels (one-off, after key generation / import; undone on export).  One thread per
// (RGSW ciphertext, component c, column, position): row(2l + c) -+= B^l row(c) mod Q, l = 1..dG-1.  Plain 64-bit
// remainders: this runs once per key.
template <typename W>
__global__ __launch_bounds__(256) void k_fold_gadget(DevParams P, W* __restrict__ bsk, u64 rgsw, int dir) {
    const u64 Q = P.is64 ? P.Q64 : (u64)P.Q;
    const u64 per = 4ull * P.N, total = rgsw * per;   // (c, col, k) per RGSW ciphertext
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u64 g = i / per, rem = i - g * per;
        const u32 c = (u32)(rem / (2ull * P.N));
        const u64 ck = rem - (u64)c * 2 * P.N;        // col * N + k
        W* base = bsk + g * (2ull * P.dG) * 2 * P.N;
        const u64 e0 = (u64)base[(u64)c * 2 * P.N + ck];
        u64 bl = 1;
        for (u32 l = 1; l < P.dG; ++l) {
            bl = (bl << P.gBits) % Q;
            const u64 t = (u64)(((unsigned __int128)bl * e0) % Q);
            W* w = base + (u64)(2 * l + c) * 2 * P.N + ck;
            u64 v = (u64)*w;
            if (dir > 0) {
                v = v >= t ? v - t : v + Q - t;
                if (P.fold_ninv) v = (u64)(((unsigned __int128)v * P.Ninv64) % Q);     // ... and scaled by N^-1
            } else {
                if (P.fold_ninv) v = (u64)(((unsigned __int128)v * P.N) % Q);
                v = v + t >= Q ? v + t - Q : v + t;
            }
            *w = (W)v;
        }
    }
}

hipError_t launch_fold_gadget(const DevParams& P, void* bsk, u64 rgsw, int dir, hipStream_t s) {
    if (rgsw == 0) return hipSuccess;
    if (P.is64) hipLaunchKernelGGL(k_fold_gadget<u64>, dim3(4096), dim3(256), 0, s, P, static_cast<u64*>(bsk), rgsw, dir);
    else hipLaunchKernelGGL(k_fold_gadget<u32>, dim3(4096), dim3(256), 0, s, P, static_cast<u32*>(bsk), rgsw, dir);
    return hipGetLastError();
}

}  // namespace bce

#ifdef BCE_PHASE_PROF
extern "C" int bce_debug_phase_prof(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(bce::g_phase_prof), sizeof(unsigned long long) * 256) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[256] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(bce::g_phase_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
