// keygen.hpp -- launch interface of the on-device key samplers (keygen.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace bce {

// stream domains of DESIGN.md "PRNG spec" used on the device (same values as prng.hpp StreamDomain)
constexpr uint32_t kDomBSKdev = 3, kDomKSKdev = 4;

struct KeygenParams {
    uint32_t seed[8];        // 32-byte key-generation seed as little-endian words
    uint32_t n, N;
    uint64_t Q, q, qKS;
    int qbits, ksbits;       // bit lengths of Q - 1 and qKS - 1 (masked rejection)
    uint32_t R;              // RGSW rows per ciphertext = 2 * dG
    uint32_t ap, baseR, dR;  // AP/DM key layout
    uint32_t baseKS, dKS, ksk_stride;
    uint64_t gpow[4];        // baseG^l mod Q
    const int32_t* s;        // device copies of the secret keys
    const int32_t* z;
    const uint64_t* cdf;     // device copy of the 81-entry Gaussian CDF table (prng.hpp GaussSampler)
};

// rows [r0, r0 + cnt) of the bootstrapping key, coefficient domain: bsk[cnt][2][N] (mask, noise, gadget added) and
// ta[cnt][N] (the mask again, for the a * z product); words are u32 or u64
hipError_t launch_gen_bsk_rows(const KeygenParams& kp, uint64_t r0, uint32_t cnt, void* bsk, void* ta, int words64, hipStream_t s);
// all `rows` = N * baseKS * dKS rows of the key-switching key, device layout [row][ksk_stride] of u16 / u32
hipError_t launch_gen_ksk_rows(const KeygenParams& kp, uint64_t rows, void* ksk, int u16rows, hipStream_t s);

}  // namespace bce
