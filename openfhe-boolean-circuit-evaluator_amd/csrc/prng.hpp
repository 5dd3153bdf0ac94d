// prng.hpp -- deterministic samplers of the engine (DESIGN.md "PRNG spec").
//
// OpenFHE draws keys and noise from an unseeded PRNG, so the reference pins no
// stream; this engine defines one so that keys and ciphertexts are
// reproducible: ChaCha20 (RFC 7539 block function) keyed with the 32-byte
// seed, nonce = (domain, index_lo, index_hi), block counter from 0, output
// words consumed in order.  Distributions follow the reference's use of
// OpenFHE: uniform ternary secrets, uniform masks,
// discrete Gaussian noise with sigma = 3.19.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace bce {

enum StreamDomain : uint32_t { kDomSK = 1, kDomZ = 2, kDomBSK = 3, kDomKSK = 4, kDomENC = 5 };

class ChaChaStream {
public:
    ChaChaStream(const uint8_t seed[32], uint32_t domain, uint64_t index) {
        static const uint32_t sigma[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 4; ++i) in_[i] = sigma[i];
        for (int i = 0; i < 8; ++i) {
            uint32_t w;
            std::memcpy(&w, seed + 4 * i, 4);  // little-endian host
            in_[4 + i] = w;
        }
        in_[12] = 0;
        in_[13] = domain;
        in_[14] = (uint32_t)index;
        in_[15] = (uint32_t)(index >> 32);
        have_ = 0;
    }
    uint32_t next32() {
        if (!have_) refill();
        return out_[16 - have_--];
    }
    uint64_t next64() {
        uint64_t lo = next32();
        return lo | ((uint64_t)next32() << 32);
    }

private:
    static uint32_t rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
    static void quarter(uint32_t* x, int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    }
    void refill() {
        uint32_t x[16];
        std::memcpy(x, in_, sizeof x);
        for (int round = 0; round < 10; ++round) {
            quarter(x, 0, 4, 8, 12); quarter(x, 1, 5, 9, 13); quarter(x, 2, 6, 10, 14); quarter(x, 3, 7, 11, 15);
            quarter(x, 0, 5, 10, 15); quarter(x, 1, 6, 11, 12); quarter(x, 2, 7, 8, 13); quarter(x, 3, 4, 9, 14);
        }
        for (int i = 0; i < 16; ++i) out_[i] = x[i] + in_[i];
        ++in_[12];
        have_ = 16;
    }
    uint32_t in_[16];
    uint32_t out_[16];
    int have_;
};

// uniform over {-1, 0, 1}
inline int draw_ternary(ChaChaStream& s) {
    uint32_t w;
    do { w = s.next32(); } while (w == 0xFFFFFFFFu);
    return (int)(w % 3u) - 1;
}

// uniform over [0, M) by masked rejection
inline uint64_t draw_uniform(ChaChaStream& s, uint64_t M) {
    int bits = (M > 1) ? 64 - __builtin_clzll(M - 1) : 0;
    if (bits == 0) return 0;
    if (bits <= 32) {
        uint32_t mask = bits == 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
        uint32_t w;
        do { w = s.next32() & mask; } while (w >= M);
        return w;
    }
    uint64_t mask = bits == 64 ? ~0ull : ((1ull << bits) - 1ull);
    uint64_t w;
    do { w = s.next64() & mask; } while (w >= M);
    return w;
}

// discrete Gaussian over [-40, 40] by CDF inversion on a 64-bit uniform
class GaussSampler {
public:
    static constexpr int kTail = 40;
    explicit GaussSampler(double sigma) {
        double p[2 * kTail + 1], total = 0.0;
        for (int k = 0; k <= 2 * kTail; ++k) {
            double x = (double)(k - kTail);
            p[k] = std::exp(-(x * x) / (2.0 * sigma * sigma));
            total += p[k];
        }
        double cum = 0.0;
        for (int k = 0; k <= 2 * kTail; ++k) {
            cum += p[k] / total;
            cdf_[k] = (cum >= 1.0) ? ~0ull : (uint64_t)std::ldexp(cum, 64);
        }
        cdf_[2 * kTail] = ~0ull;
    }
    int draw(ChaChaStream& s) const {
        uint64_t u = s.next64();
        int lo = 0, hi = 2 * kTail;
        while (lo < hi) {
            int mid = (lo + hi) / 2;
            if (u < cdf_[mid]) hi = mid; else lo = mid + 1;
        }
        return lo - kTail;
    }
    const uint64_t* table() const { return cdf_; }  // 2 * kTail + 1 entries (device copy for keygen.hip)

private:
    uint64_t cdf_[2 * kTail + 1];
};

}  // namespace bce
