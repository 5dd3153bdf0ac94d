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
 *           optional trailer (recommended with bsk_format = 1): magic "BCENTTCK", u32 count, u32 0, then count pairs
 *                         (coef[N], eval[N]) of u64 words -- one polynomial in COEFFICIENT representation and the same
 *                         polynomial as the producer holds it in EVALUATION representation.  The importer transforms coef
 *                         with its own forward transform and REFUSES an evaluation-form key whose pairs do not match: the
 *                         producer's evaluation order is then not the engine's, and the key must be exported with bsk_format = 0.
 */
#ifndef BCE_KEYFILE_H
#define BCE_KEYFILE_H
#include <stdint.h>

#define BCE_KEYFILE_MAGIC "BCEKEYS1"
#define BCE_KEYFILE_VERSION 1u
#define BCE_KEYFILE_BSK_COEFFICIENT 0u
#define BCE_KEYFILE_BSK_EVALUATION 1u
#define BCE_KEYFILE_NTTCHECK_MAGIC "BCENTTCK"

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


/*
 * ---- gate-vector file ("BCEGVEC1") --------------------------------------------------------------------------------
 * What OpenFHE itself RETURNED for the calls the reference makes on BinFHEContext (EvalBinGate src/gate.cpp:133,146,172,
 * 200-202; EvalNOT :112,198-199; Encrypt src/circuit.cpp:506; Decrypt :800), recorded by export_keys.cpp next to the
 * keys of the same context and replayed on the engine by tools/openfhe_export/compare.py: every output word must be
 * identical.  This file is the ciphertext-level parity statement against the reference's own encrypted path.
 *
 *   offset  size  field
 *   0       8     magic "BCEGVEC1"
 *   8       4     version (1)
 *   12      4     method (1 AP, 2 GINX)
 *   16      8x8   n, N, q, Q, qKS, baseKS, baseG, baseR      must equal the key file's
 *   80      8     count                                      number of records that follow
 *   88            records, back to back
 *
 * record:
 *   0       4     kind            (enum below)
 *   4       4     in_bits         bit 0 / bit 1: the plaintexts behind ct1 / ct2 (documentation; Encrypt: the bit)
 *   8       4     decrypted       what OpenFHE's Decrypt(sk, out) returned (TAIL, NTT records: 0)
 *   12      4     payload_words   u64 words that follow
 *   16      8*payload_words       payload, u64 words:
 *
 *   kind 0..5   EvalBinGate(gate = kind: OR, AND, NOR, NAND, XOR_FAST, XNOR_FAST -- lbcrypto::BINGATE order)
 *               ct1[n+1], ct2[n+1], out[n+1]           every ciphertext as (a_0 .. a_{n-1}, b) mod q
 *   kind 16     EvalNOT(ct1):                          ct1[n+1], out[n+1]
 *   kind 17     Bootstrap(ct1):                        ct1[n+1], out[n+1]
 *   kind 32     Encrypt(sk, bit) in the library's DEFAULT output mode: out[n+1]   (not replayable: checked for its
 *               plaintext and its noise, which tells FRESH from BOOTSTRAPPED defaults apart)
 *   kind 33     Encrypt(sk, bit, FRESH):               out[n+1]
 *   kind 48     TAIL -- the calls EvalBinGate makes after the accumulator, through the public LWE scheme:
 *               ctQ[N+1] (an LWE ciphertext of dimension N mod Q), lweN[N+1] = ModSwitch(qKS, ctQ),
 *               ks[n+1] = KeySwitch(params, K, lweN), out[n+1] = ModSwitch(q, ks).  Localises a gate mismatch:
 *               tail records equal + gate records different => the difference is in the blind rotation.
 *   kind 64     NTT -- coef[N], eval[N] = the same NativePoly after SetFormat(EVALUATION): pins the evaluation order
 *               bsk_format = 1 key files rely on (bce_debug_ntt of coef must equal eval).
 */
#define BCE_GATEVEC_MAGIC "BCEGVEC1"
#define BCE_GATEVEC_VERSION 1u
enum {
    BCE_GATEVEC_OR = 0, BCE_GATEVEC_AND = 1, BCE_GATEVEC_NOR = 2, BCE_GATEVEC_NAND = 3, BCE_GATEVEC_XOR_FAST = 4,
    BCE_GATEVEC_XNOR_FAST = 5, BCE_GATEVEC_NOT = 16, BCE_GATEVEC_BOOTSTRAP = 17, BCE_GATEVEC_ENCRYPT_DEFAULT = 32,
    BCE_GATEVEC_ENCRYPT_FRESH = 33, BCE_GATEVEC_TAIL = 48, BCE_GATEVEC_NTT = 64
};

#pragma pack(push, 1)
typedef struct bce_gatevec_header {
    char     magic[8];
    uint32_t version;
    uint32_t method;
    uint64_t n, N, q, Q, qKS, baseKS, baseG, baseR;
    uint64_t count;
} bce_gatevec_header; /* 88 bytes */
typedef struct bce_gatevec_record {
    uint32_t kind;
    uint32_t in_bits;
    uint32_t decrypted;
    uint32_t payload_words;
} bce_gatevec_record; /* 16 bytes, followed by payload_words u64 words */
#pragma pack(pop)

#endif
