// keygen.hip -- on-device sampling for BTKeyGen: the bootstrapping-key rows (RGSW masks and noise)
// and the whole LWE key-switching key are drawn on the GPU, straight into the layouts the kernels read.
//
// Replaces what the reference obtains from cc.BTKeyGen(sk) (src/circuit.cpp:91 -> OpenFHE KeyGenAcc /
// KeySwitchGen).  The streams are the ones of DESIGN.md "PRNG spec" -- ChaCha20 (RFC 7539 block function), key =
// 32-byte seed, nonce = (domain, index), words consumed in order; masked-rejection uniform draws, CDF-inversion
// Gaussian -- so the keys are bit-identical to the host samplers of prng.hpp and to the oracle's (the keygen
// parity tests compare every word).  One THREAD owns one stream (= one key row): draws inside a stream are
// sequential by construction (rejection sampling consumes a data-dependent number of words), rows are independent.
// A 64-thread block stages 64 rows x 64 elements in LDS and writes them out as 64 coalesced row segments.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "keygen.hpp"

namespace bce {
namespace {

constexpr int KT = 64;  // threads per block = rows per tile = elements per tile row

// ChaCha20 stream with the output block parked in LDS (dynamic word index without scratch): word i of thread t at
// obuf[i * KT + t]
struct DevChaCha {
    u32 in[16];
    u32* obuf;
    int have;
    __device__ void init(const u32* seed8, u32 domain, u64 index, u32* lds, u32 t) {
        in[0] = 0x61707865u; in[1] = 0x3320646eu; in[2] = 0x79622d32u; in[3] = 0x6b206574u;
#pragma unroll
        for (int i = 0; i < 8; ++i) in[4 + i] = seed8[i];
        in[12] = 0; in[13] = domain; in[14] = (u32)index; in[15] = (u32)(index >> 32);
        obuf = lds + t;
        have = 0;
    }
    static __device__ __forceinline__ u32 rotl(u32 v, int c) { return (v << c) | (v >> (32 - c)); }
    __device__ void refill() {
        u32 x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = in[i];
#define BCE_QR(a, b, c, d)                                \
    x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);           \
    x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);           \
    x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);            \
    x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        for (int round = 0; round < 10; ++round) {
            BCE_QR(0, 4, 8, 12) BCE_QR(1, 5, 9, 13) BCE_QR(2, 6, 10, 14) BCE_QR(3, 7, 11, 15)
            BCE_QR(0, 5, 10, 15) BCE_QR(1, 6, 11, 12) BCE_QR(2, 7, 8, 13) BCE_QR(3, 4, 9, 14)
        }
#undef BCE_QR
#pragma unroll
        for (int i = 0; i < 16; ++i) obuf[i * KT] = x[i] + in[i];
        ++in[12];
        have = 16;
    }
    __device__ __forceinline__ u32 next32() {
        if (!have) refill();
        const u32 w = obuf[(16 - have) * KT];
        --have;
        return w;
    }
    __device__ __forceinline__ u64 next64() {
        const u64 lo = next32();
        return lo | ((u64)next32() << 32);
    }
};

// uniform over [0, M) by masked rejection (prng.hpp draw_uniform); bits = bit length of M - 1
__device__ __forceinline__ u64 draw_uniform(DevChaCha& s, u64 M, int bits) {
    if (bits == 0) return 0;
    if (bits <= 32) {
        const u32 mask = bits == 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
        u32 w;
        do { w = s.next32() & mask; } while (w >= M);
        return w;
    }
    const u64 mask = bits == 64 ? ~0ull : ((1ull << bits) - 1ull);
    u64 w;
    do { w = s.next64() & mask; } while (w >= M);
    return w;
}
// discrete Gaussian by CDF inversion on a 64-bit uniform (prng.hpp GaussSampler::draw); cdf: 81 entries in LDS
__device__ __forceinline__ int draw_gauss(DevChaCha& s, const u64* cdf) {
    const u64 u = s.next64();
    int lo = 0, hi = 80;
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (u < cdf[mid]) hi = mid; else lo = mid + 1;
    }
    return lo - 40;
}
__device__ __forceinline__ u64 lift_signed_dev(int v, u64 M) { return v >= 0 ? (u64)v : M - (u64)(-(long long)v); }

// ---- bootstrapping-key rows -------------------------------------------------------------------------------
// Row `rowid` of the key (engine.cpp keygen_bsk): stream (kDomBSK, rowid) gives N uniform words mod Q (the mask a)
// then N Gaussian words (e); the gadget power is added to the message coefficient of a (even rows) or e (odd rows).
// Output: bsk[local][0][N] = a (+ gadget), bsk[local][1][N] = e (+ gadget), ta[local][N] = a  -- coefficient domain;
// the caller transforms all three and adds NTT(a) * NTT(z) into the second polynomial.
template <typename W>
__global__ __launch_bounds__(KT) void k_gen_bsk_rows(KeygenParams kp, u64 r0, u32 cnt, W* __restrict__ bsk, W* __restrict__ ta) {
    __shared__ u32 obuf[16 * KT];
    __shared__ u64 cdf[81];
    __shared__ W tile[KT][KT + 1];
    const u32 t = threadIdx.x;
    for (u32 i = t; i < 81; i += KT) cdf[i] = kp.cdf[i];
    __syncthreads();
    const u32 N = kp.N, R = kp.R;
    const u64 Q = kp.Q;
    const u32 local = blockIdx.x * KT + t;
    const u64 rowid = r0 + local;
    bool active = local < cnt, one = false, negate = false;
    u32 mm = 0;
    const u32 r = (u32)(rowid % R);
    const u64 ek = rowid / R;
    if (active) {
        if (!kp.ap) {
            const u32 i = (u32)(ek / 2), key = (u32)(ek % 2);
            one = key == 0 ? (kp.s[i] == 1) : (kp.s[i] == -1);
        } else {
            const u32 k = (u32)(ek % kp.dR), v = (u32)((ek / kp.dR) % kp.baseR), i = (u32)(ek / kp.dR / kp.baseR);
            if (v == 0) {
                active = false;  // never read: stays zero, and its stream is never drawn (as on the host)
            } else {
                long long pw = 1;
                for (u32 j = 0; j < k; ++j) pw *= kp.baseR;
                const long long qq = (long long)kp.q, m = (long long)kp.s[i] * (long long)v * pw;
                long long e = (((m % qq) + qq) % qq) * (long long)(2 * N / qq);
                if (e >= (long long)N) { e -= N; negate = true; }
                mm = (u32)e;
                one = true;
            }
        }
    }
    const u64 g = kp.gpow[r >> 1], gadd = negate ? Q - g : g;
    DevChaCha st;
    st.init(kp.seed, kDomBSKdev, rowid, obuf, t);
    const u32 rows_here = min((u32)KT, cnt - min(cnt, blockIdx.x * KT));
    for (u32 phase = 0; phase < 2; ++phase) {   // 0: mask a, 1: noise e
        for (u32 k0 = 0; k0 < N; k0 += KT) {
            if (active) {
                for (u32 j = 0; j < KT; ++j) {
                    u64 v;
                    if (phase == 0) v = draw_uniform(st, Q, kp.qbits);
                    else v = lift_signed_dev(draw_gauss(st, cdf), Q);
                    tile[t][j] = (W)v;
                }
            } else {
                for (u32 j = 0; j < KT; ++j) tile[t][j] = (W)0;
            }
            __syncthreads();
            if (phase == 0)
                for (u32 rr = 0; rr < rows_here; ++rr) ta[((size_t)blockIdx.x * KT + rr) * N + k0 + t] = tile[rr][t];
            __syncthreads();
            if (active && one && (r & 1u) == phase && mm >= k0 && mm < k0 + KT)
                tile[t][mm - k0] = (W)(((u64)tile[t][mm - k0] + gadd) % Q);
            __syncthreads();
            for (u32 rr = 0; rr < rows_here; ++rr) bsk[(((size_t)blockIdx.x * KT + rr) * 2 + phase) * N + k0 + t] = tile[rr][t];
            __syncthreads();
        }
    }
}

// ---- LWE key-switching key -----------------------------------------------------------------------------------
// Row idx = (i * baseKS + v) * dKS + j (engine.cpp bce_keygen): stream (kDomKSK, idx) gives n uniform words mod qKS
// (the mask) and one Gaussian word; b = <a, s> + e + z_i * v * baseKS^j  (mod qKS).  Written straight into the
// padded device layout [row][ksk_stride] of u16 / u32 elements.
template <typename KE>
__global__ __launch_bounds__(KT) void k_gen_ksk_rows(KeygenParams kp, u64 rows, KE* __restrict__ ksk) {
    __shared__ u32 obuf[16 * KT];
    __shared__ u64 cdf[81];
    __shared__ u32 tile[KT][KT + 1];
    extern __shared__ int s_lds[];  // n secret coefficients
    const u32 t = threadIdx.x;
    for (u32 i = t; i < 81; i += KT) cdf[i] = kp.cdf[i];
    for (u32 i = t; i < kp.n; i += KT) s_lds[i] = kp.s[i];
    __syncthreads();
    const u32 n = kp.n, B = kp.baseKS, D = kp.dKS;
    const u64 qKS = kp.qKS;
    const u64 idx = (u64)blockIdx.x * KT + t;
    const bool active = idx < rows;
    const u32 rows_here = (u32)min((u64)KT, rows - min(rows, (u64)blockIdx.x * KT));
    DevChaCha st;
    st.init(kp.seed, kDomKSKdev, idx, obuf, t);
    u64 acc = 0;
    for (u32 k0 = 0; k0 <= n; k0 += KT) {
        if (active) {
            for (u32 j = 0; j < KT; ++j) {
                const u32 k = k0 + j;
                u32 v = 0;
                if (k < n) {
                    const u64 a = draw_uniform(st, qKS, kp.ksbits);
                    const int sk = s_lds[k];
                    acc += sk == 1 ? a : (sk == -1 ? (a ? qKS - a : 0) : 0);
                    v = (u32)a;
                } else if (k == n) {
                    const u32 j_d = (u32)(idx % D), vv = (u32)((idx / D) % B), i = (u32)(idx / D / B);
                    const u64 e = lift_signed_dev(draw_gauss(st, cdf), qKS);
                    u64 pw = 1;
                    for (u32 x = 0; x < j_d; ++x) pw *= B;
                    const u64 zi = lift_signed_dev(kp.z[i], qKS);
                    const u64 msg = (zi * (((u64)vv * pw) % qKS)) % qKS;
                    v = (u32)((acc % qKS + e + msg) % qKS);
                }
                tile[t][j] = v;
            }
        }
        __syncthreads();
        if (k0 + t <= n)
            for (u32 rr = 0; rr < rows_here; ++rr) ksk[((size_t)blockIdx.x * KT + rr) * kp.ksk_stride + k0 + t] = (KE)tile[rr][t];
        __syncthreads();
    }
}

}  // namespace

hipError_t launch_gen_bsk_rows(const KeygenParams& kp, u64 r0, u32 cnt, void* bsk, void* ta, int words64, hipStream_t s) {
    if (cnt == 0) return hipSuccess;
    const dim3 grid((cnt + KT - 1) / KT), block(KT);
    if (words64) hipLaunchKernelGGL(k_gen_bsk_rows<u64>, grid, block, 0, s, kp, r0, cnt, static_cast<u64*>(bsk), static_cast<u64*>(ta));
    else hipLaunchKernelGGL(k_gen_bsk_rows<u32>, grid, block, 0, s, kp, r0, cnt, static_cast<u32*>(bsk), static_cast<u32*>(ta));
    return hipGetLastError();
}

hipError_t launch_gen_ksk_rows(const KeygenParams& kp, u64 rows, void* ksk, int u16rows, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    const dim3 grid((u32)((rows + KT - 1) / KT)), block(KT);
    const size_t lds = kp.n * sizeof(int);
    if (u16rows) hipLaunchKernelGGL(k_gen_ksk_rows<uint16_t>, grid, block, lds, s, kp, rows, static_cast<uint16_t*>(ksk));
    else hipLaunchKernelGGL(k_gen_ksk_rows<u32>, grid, block, lds, s, kp, rows, static_cast<u32*>(ksk));
    return hipGetLastError();
}

}  // namespace bce
