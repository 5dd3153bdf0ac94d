// host_math.hpp -- host-side number theory for context construction.
//
// Follows what OpenFHE's GenerateBinFHEContext needs (reference call site
// src/circuit.cpp:88): Q = PreviousPrime(FirstPrime(bits, 2N), 2N), a primitive
// 2N-th root of unity for the negacyclic NTT, and digit counts.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace bce {

using u32 = uint32_t;
using u64 = uint64_t;
using u128 = unsigned __int128;

inline u64 mul_mod(u64 a, u64 b, u64 m) { return (u64)((u128)a * b % m); }

inline u64 pow_mod(u64 a, u64 e, u64 m) {
    u64 r = 1 % m;
    a %= m;
    for (; e; e >>= 1) {
        if (e & 1) r = mul_mod(r, a, m);
        a = mul_mod(a, a, m);
    }
    return r;
}

// deterministic Miller-Rabin for 64-bit integers
inline bool is_prime_u64(u64 n) {
    static const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return false;
    for (u64 b : bases) {
        if (n == b) return true;
        if (n % b == 0) return false;
    }
    u64 d = n - 1;
    int s = 0;
    while (!(d & 1)) { d >>= 1; ++s; }
    for (u64 b : bases) {
        u64 x = pow_mod(b, d, n);
        if (x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int r = 1; r < s && witness; ++r) {
            x = mul_mod(x, x, n);
            if (x == n - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}

// smallest prime > 2^bits congruent to 1 mod m
inline u64 first_prime(u32 bits, u64 m) {
    u64 q = (u64)1 << bits;
    u64 r = q % m;
    q += r ? (m - r) + 1 : 1;
    while (!is_prime_u64(q)) q += m;
    return q;
}

// next smaller prime in the same residue class
inline u64 previous_prime(u64 q, u64 m) {
    do { q -= m; } while (!is_prime_u64(q));
    return q;
}

// minimal primitive m-th root of unity mod Q (m a power of two dividing Q-1)
inline u64 min_primitive_root(u64 Q, u64 m) {
    u64 cof = (Q - 1) / m, g = 0;
    for (u64 x = 2; x < Q && !g; ++x) {
        u64 c = pow_mod(x, cof, Q);
        if (pow_mod(c, m / 2, Q) == Q - 1) g = c;
    }
    u64 best = g, cur = g, g2 = mul_mod(g, g, Q);
    for (u64 k = 1; k < m; k += 2) {
        if (cur < best) best = cur;
        cur = mul_mod(cur, g2, Q);
    }
    return best;
}

inline u32 bit_reverse(u32 x, int bits) {
    u32 r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

inline int bit_length(u64 v) { return v ? 64 - __builtin_clzll(v) : 0; }

// ceil(log(modulus)/log(base)) in double, as the cryptoparameter classes compute it
inline u32 digit_count(double modulus, double base) { return (u32)std::ceil(std::log(modulus) / std::log(base)); }

inline u64 lift_signed(int v, u64 M) { return v >= 0 ? (u64)v : M - (u64)(-(int64_t)v); }

}  // namespace bce
