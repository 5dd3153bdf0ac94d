/*
 * bce_keyfile.h -- on-disk exchange format of the key material libbce_amd.so imports (bce_import_keys_file,
 * include/bce_gpu.h).  Shared by the engine (csrc/keyfile.cpp), the OpenFHE-side exporter in this directory and
 * the Python writer of the loader test (tests/test_gpu_keyfile.py).
 *
 * All integers little-endian.  Polynomials are in COEFFICIENT representation with coefficients in natural order
 * and values in [0, Q): the importer transforms them on the device, so the producer's NTT ordering and choice of
 * root never have to match the engine's.
 *
 *   offset  size  field
 *   0       8     magic "BCEKEYS1"
 *   8       4     version (1)
 *   12      4     method: 1 = AP (DM), 2 = GINX (CGGI)                       lbcrypto::BINFHE_METHOD
 *   16      8x8   n, N, q, Q, qKS, baseKS, baseG, baseR                       the context's parameters
 *   80      8     bsk_words   (GINX: n*2*R*2*N, AP: n*baseR*dR*R*2*N with R = 2*ceil(log_baseG Q))
 *   88      8     ksk_words   (N*baseKS*dKS*(n+1))
 *   96      4     has_z       (1: the ring secret follows s; it is only needed to RE-export keys, not to evaluate)
 *   100     4     bsk_format  (0: COEFFICIENT representation as described below; 1: EVALUATION representation in the
 *                                bit-reversed order of OpenFHE's forward transform for the minimal primitive 2N-th root:
 *                                what DCRTPoly / NativePoly hold after BTKeyGen, dumped without SetFormat)
 *   104     4n    s[n]        LWE secret, entries in {-1, 0, 1}
 *           4N    z[N]        if has_z
 *           pad to a multiple of 8
 *           8*bsk_words   bootstrapping key, u64 words:
 *                           GINX [i < n][key < 2][row < R][col < 2][N]   key 0 = RGSW(s_i == +1), key 1 = RGSW(s_i == -1)
 *                           AP   [i < n][v < baseR][k < dR][row < R][col < 2][N]   (v = 0 entries all zero)
 *                         row r of an RGSW ciphertext = RLWE pair (a_r, a_r * z + e_r) with the gadget power
 *                         baseG^(r/2) * message added to column r mod 2
 *           4*ksk_words   key-switching key, u32 words: [i < N][v < baseKS][j < dKS][a_0 .. a_{n-1}, b]  mod qKS
 */
#ifndef BCE_KEYFILE_H
#define BCE_KEYFILE_H
#include <stdint.h>

#define BCE_KEYFILE_MAGIC "BCEKEYS1"
#define BCE_KEYFILE_VERSION 1u
#define BCE_KEYFILE_BSK_COEFFICIENT 0u
#define BCE_KEYFILE_BSK_EVALUATION 1u

#pragma pack(push, 1)
typedef struct bce_keyfile_header {
    char     magic[8];
    uint32_t version;
    uint32_t method;
    uint64_t n, N, q, Q, qKS, baseKS, baseG, baseR;
    uint64_t bsk_words;
    uint64_t ksk_words;
    uint32_t has_z;
    uint32_t bsk_format;   /* 0 coefficient, 1 evaluation (OpenFHE order) */
} bce_keyfile_header; /* 104 bytes */
#pragma pack(pop)

#endif
