/*
 * binfhe_oracle.c -- TEST INFRASTRUCTURE ONLY (see binfhe_oracle.h header).
 *
 * Plain-C restatement of OpenFHE v1.0.x `binfhe` as used by the reference
 * (openfhe-boolean-circuit-evaluator: src/circuit.cpp:88-91,506,800;
 * src/gate.cpp:112,133,146,172,198-202).  OpenFHE is a third-party dependency
 * that is absent from /root/reference and from this image; upstream file names
 * are cited per function ("upstream:") from the published v1.0.x sources.
 * Pinned to the reference's functional known answers by the oracle-alone walks
 * of tests/test_oracle.py (see the header); PARITY UNPINNED at ciphertext level
 * (no golden ciphertexts exist anywhere in the reference).
 *
 * Word layout follows the reference: every ring / LWE element is a uint64_t
 * (OpenFHE NativeInteger).  Internal NTT ordering follows OpenFHE (forward =
 * Cooley-Tukey natural->bit-reversed with psi = minimal primitive 2N-th root,
 * inverse = Gentleman-Sande), although no output depends on it.
 */
#include "binfhe_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint32_t u32;

/* ------------------------------------------------------------------ */
/* PRNG: ChaCha20 block function (RFC 7539) used as a counter stream.  */
/* Spec shared with the product's own independent implementation       */
/* (DESIGN.md "PRNG spec"): key = 32-byte seed, nonce = (domain,       */
/* index_lo, index_hi), block counter from 0, words consumed in order. */
/* ------------------------------------------------------------------ */
typedef struct {
    u32 st[16];
    u32 buf[16];
    int pos;
} stream_t;

#define ROTL32(v, c) (((v) << (c)) | ((v) >> (32 - (c))))
#define QR(a, b, c, d)                                                                                  \
    a += b; d ^= a; d = ROTL32(d, 16); c += d; b ^= c; b = ROTL32(b, 12);                               \
    a += b; d ^= a; d = ROTL32(d, 8);  c += d; b ^= c; b = ROTL32(b, 7);

static void chacha_block(stream_t* s) {
    u32 x[16];
    memcpy(x, s->st, sizeof x);
    for (int i = 0; i < 10; i++) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) s->buf[i] = x[i] + s->st[i];
    s->st[12]++;
    s->pos = 0;
}

static void stream_init(stream_t* s, const uint8_t seed[32], u32 domain, u64 index) {
    s->st[0] = 0x61707865u; s->st[1] = 0x3320646eu; s->st[2] = 0x79622d32u; s->st[3] = 0x6b206574u;
    for (int i = 0; i < 8; i++)
        s->st[4 + i] = (u32)seed[4 * i] | ((u32)seed[4 * i + 1] << 8) | ((u32)seed[4 * i + 2] << 16) | ((u32)seed[4 * i + 3] << 24);
    s->st[12] = 0;
    s->st[13] = domain;
    s->st[14] = (u32)index;
    s->st[15] = (u32)(index >> 32);
    s->pos = 16;
}

static inline u32 next32(stream_t* s) {
    if (s->pos == 16) chacha_block(s);
    return s->buf[s->pos++];
}
static inline u64 next64(stream_t* s) {
    u64 lo = next32(s);
    u64 hi = next32(s);
    return lo | (hi << 32);
}

enum { DOM_SK = 1, DOM_Z = 2, DOM_BSK = 3, DOM_KSK = 4, DOM_ENC = 5 };

/* upstream: core/lib/math/ternaryuniformgenerator-impl.h (uniform over {-1,0,1}) */
static inline int sample_ternary(stream_t* s) {
    for (;;) {
        u32 w = next32(s);
        if (w == 0xFFFFFFFFu) continue; /* 2^32 = 3*1431655765 + 1 */
        return (int)(w % 3u) - 1;
    }
}

static inline int bitlen64(u64 v) { return v ? 64 - __builtin_clzll(v) : 0; }

/* upstream: core/lib/math/discreteuniformgenerator-impl.h (rejection sampling) */
static inline u64 sample_uniform(stream_t* s, u64 M) {
    int bits = bitlen64(M - 1);
    if (bits == 0) return 0;
    if (bits <= 32) {
        u32 mask = bits == 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
        for (;;) {
            u32 w = next32(s) & mask;
            if (w < M) return w;
        }
    }
    u64 mask = bits == 64 ? ~0ull : ((1ull << bits) - 1ull);
    for (;;) {
        u64 w = next64(s) & mask;
        if (w < M) return w;
    }
}

/* upstream: core/lib/math/discretegaussiangenerator-impl.h (inversion sampling,
 * "Peikert" branch used for sigma = 3.19).  CDF table over [-R, R]. */
#define DGG_R 40
typedef struct {
    u64 thr[2 * DGG_R + 1];
} dgg_t;

static void dgg_init(dgg_t* g, double sigma) {
    double p[2 * DGG_R + 1];
    double S = 0.0;
    for (int k = 0; k <= 2 * DGG_R; k++) {
        double x = (double)(k - DGG_R);
        p[k] = exp(-(x * x) / (2.0 * sigma * sigma));
        S += p[k];
    }
    double cum = 0.0;
    for (int k = 0; k <= 2 * DGG_R; k++) {
        cum += p[k] / S;
        g->thr[k] = (cum >= 1.0) ? ~0ull : (u64)ldexp(cum, 64);
    }
    g->thr[2 * DGG_R] = ~0ull;
}

static inline int sample_gauss(const dgg_t* g, stream_t* s) {
    u64 U = next64(s);
    int lo = 0, hi = 2 * DGG_R; /* first k with U < thr[k] (last bucket catches everything) */
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (U < g->thr[mid]) hi = mid; else lo = mid + 1;
    }
    return lo - DGG_R;
}

/* ------------------------------------------------------------------ */
/* number theory (upstream: core/lib/math/nbtheory-impl.h)             */
/* ------------------------------------------------------------------ */
static inline u64 mulmod(u64 a, u64 b, u64 m) { return (u64)((u128)a * b % m); }
static u64 powmod(u64 a, u64 e, u64 m) {
    u64 r = 1 % m;
    a %= m;
    while (e) {
        if (e & 1) r = mulmod(r, a, m);
        a = mulmod(a, a, m);
        e >>= 1;
    }
    return r;
}

static int is_prime(u64 n) {
    if (n < 2) return 0;
    static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (unsigned i = 0; i < 12; i++) {
        if (n == small[i]) return 1;
        if (n % small[i] == 0) return 0;
    }
    u64 d = n - 1;
    int r = 0;
    while ((d & 1) == 0) { d >>= 1; r++; }
    for (unsigned i = 0; i < 12; i++) { /* deterministic for 64-bit n */
        u64 x = powmod(small[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int j = 1; j < r; j++) {
            x = mulmod(x, x, n);
            if (x == n - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

/* FirstPrime(nBits, m): smallest prime q > 2^nBits with q = 1 (mod m) */
uint64_t bo_first_prime(uint32_t bits, uint64_t m) {
    u64 q = 1ull << bits;
    u64 r = q % m;
    q = r ? q + (m - r) + 1 : q + 1;
    while (!is_prime(q)) q += m;
    return q;
}
/* PreviousPrime(q, m): largest prime < q with the same residue mod m */
uint64_t bo_previous_prime(uint64_t q, uint64_t m) {
    q -= m;
    while (!is_prime(q)) q -= m;
    return q;
}

/* RootOfUnity(m, Q): minimal primitive m-th root of unity mod Q (m power of two) */
uint64_t bo_min_primitive_root(uint64_t Q, uint64_t m) {
    /* find any element of order exactly m: x^((Q-1)/m) with x a non-residue of order 2-power full */
    u64 e = (Q - 1) / m;
    u64 psi = 0;
    for (u64 x = 2; x < Q; x++) {
        u64 c = powmod(x, e, Q);
        if (powmod(c, m / 2, Q) == Q - 1) { psi = c; break; }
    }
    /* all primitive m-th roots are psi^k for odd k; return the minimum */
    u64 best = psi, cur = psi, sq = mulmod(psi, psi, Q);
    for (u64 k = 1; k < m; k += 2) {
        if (cur < best) best = cur;
        cur = mulmod(cur, sq, Q);
    }
    return best;
}

/* ------------------------------------------------------------------ */
/* context                                                             */
/* ------------------------------------------------------------------ */
struct bo_ctx {
    u32 n, N, logN;
    u64 q, Q, qKS;
    u32 baseKS, dKS, baseG, gBits, dG, baseR, dR;
    int method;
    double sigma;
    dgg_t dgg;
    /* Barrett for 128-bit products mod Q */
    int bq;      /* bit length of Q */
    u64 mu;      /* floor(2^(2bq+5)/Q) */
    /* NTT tables (bit-reversed order), Shoup companions */
    u64 psi, *tw, *tws, *itw, *itws, Ninv, Ninvs;
    u64* Gpow;   /* baseG^i */
    u64* mono;   /* GINX: [2N][N] EVALUATION-format X^k - 1 */
    u64 gateConst[6];
    /* keys */
    uint8_t seed[32];
    int have_keys;
    int32_t *s, *z;
    u64 bsk_polys; /* number of N-word polys in bsk */
    u64* bsk;      /* EVALUATION format. GINX [i][key][row][col][N]; AP [i][v][k][row][col][N] */
    u32* ksk;      /* [i][v][j][n+1] */
};

static inline u64 shoup_pre(u64 w, u64 Q) { return (u64)(((u128)w << 64) / Q); }
static inline u64 mul_shoup(u64 y, u64 w, u64 ws, u64 Q) {
    u64 qh = (u64)(((u128)ws * y) >> 64);
    u64 t = w * y - qh * Q;
    return t >= Q ? t - Q : t;
}
static inline u64 barrett128(const bo_ctx* c, u128 x) {
    /* x < 2^(2bq+4); q^ = ((x >> (bq-1)) * mu) >> (bq+6) */
    u128 xh = x >> (c->bq - 1);
    u128 qh = (xh * c->mu) >> (c->bq + 6);
    u64 r = (u64)(x - qh * c->Q);
    while (r >= c->Q) r -= c->Q;
    return r;
}
static inline u64 mulmodQ(const bo_ctx* c, u64 a, u64 b) { return barrett128(c, (u128)a * b); }

static u32 bitrev(u32 x, int bits) {
    u32 r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

/* upstream: core/include/math/hal/intnat/transformnat-impl.h
 * ForwardTransformToBitReverseInPlace (Cooley-Tukey) */
void bo_ntt_forward(const bo_ctx* c, uint64_t* x) {
    const u64 Q = c->Q;
    u32 N = c->N, t = N >> 1;
    for (u32 m = 1; m < N; m <<= 1, t >>= 1) {
        for (u32 i = 0; i < m; i++) {
            u64 w = c->tw[m + i], ws = c->tws[m + i];
            u32 j1 = 2 * i * t, j2 = j1 + t;
            for (u32 j = j1; j < j2; j++) {
                u64 u = x[j];
                u64 v = mul_shoup(x[j + t], w, ws, Q);
                u64 a = u + v;
                x[j] = a >= Q ? a - Q : a;
                x[j + t] = u >= v ? u - v : u + Q - v;
            }
        }
    }
}
/* InverseTransformFromBitReverseInPlace (Gentleman-Sande) */
void bo_ntt_inverse(const bo_ctx* c, uint64_t* x) {
    const u64 Q = c->Q;
    u32 N = c->N, t = 1;
    for (u32 m = N >> 1; m >= 1; m >>= 1, t <<= 1) {
        for (u32 i = 0; i < m; i++) {
            u64 w = c->itw[m + i], ws = c->itws[m + i];
            u32 j1 = 2 * i * t, j2 = j1 + t;
            for (u32 j = j1; j < j2; j++) {
                u64 u = x[j], v = x[j + t];
                u64 a = u + v;
                x[j] = a >= Q ? a - Q : a;
                u64 d = u >= v ? u - v : u + Q - v;
                x[j + t] = mul_shoup(d, w, ws, Q);
            }
        }
    }
    for (u32 j = 0; j < N; j++) x[j] = mul_shoup(x[j], c->Ninv, c->Ninvs, Q);
}

static u32 digits_for(double modulus, double base) { return (u32)ceil(log(modulus) / log(base)); }

static bo_ctx* ctx_build(u32 n, u32 N, u64 q, u64 Q, u64 qKS, u32 baseKS, u32 baseG, u32 baseR, int method) {
    bo_ctx* c = (bo_ctx*)calloc(1, sizeof *c);
    c->n = n; c->N = N; c->q = q; c->Q = Q; c->qKS = qKS;
    c->baseKS = baseKS; c->baseG = baseG; c->baseR = baseR; c->method = method;
    c->sigma = 3.19;
    c->logN = 0;
    while ((1u << c->logN) < N) c->logN++;
    /* upstream: lwe-cryptoparameters.h / rgsw-cryptoparameters.h digit counts */
    c->dKS = digits_for((double)qKS, (double)baseKS);
    c->dG = digits_for((double)Q, (double)baseG);
    c->dR = digits_for((double)q, (double)baseR);
    c->gBits = 0;
    while ((1u << c->gBits) < baseG) c->gBits++;
    dgg_init(&c->dgg, c->sigma);
    c->bq = bitlen64(Q);
    c->mu = (u64)(((u128)1 << (2 * c->bq + 5)) / Q);

    /* NTT tables */
    c->psi = bo_min_primitive_root(Q, 2ull * N);
    u64 ipsi = powmod(c->psi, Q - 2, Q);
    c->tw = (u64*)malloc(sizeof(u64) * N); c->tws = (u64*)malloc(sizeof(u64) * N);
    c->itw = (u64*)malloc(sizeof(u64) * N); c->itws = (u64*)malloc(sizeof(u64) * N);
    u64 p = 1, ip = 1;
    for (u32 i = 0; i < N; i++) {
        u32 r = bitrev(i, (int)c->logN);
        c->tw[r] = p; c->tws[r] = shoup_pre(p, Q);
        c->itw[r] = ip; c->itws[r] = shoup_pre(ip, Q);
        p = mulmod(p, c->psi, Q);
        ip = mulmod(ip, ipsi, Q);
    }
    c->Ninv = powmod(N, Q - 2, Q);
    c->Ninvs = shoup_pre(c->Ninv, Q);

    /* upstream: rgsw-cryptoparameters.h PreCompute(): Gpower, gate constants, monomials */
    c->Gpow = (u64*)malloc(sizeof(u64) * c->dG);
    u64 v = 1;
    for (u32 i = 0; i < c->dG; i++) { c->Gpow[i] = v; v = mulmod(v, baseG, Q); }
    c->gateConst[BO_OR] = 5 * (q >> 3);
    c->gateConst[BO_AND] = 7 * (q >> 3);
    c->gateConst[BO_NOR] = 1 * (q >> 3);
    c->gateConst[BO_NAND] = 3 * (q >> 3);
    c->gateConst[BO_XOR_FAST] = 5 * (q >> 3);
    c->gateConst[BO_XNOR_FAST] = 1 * (q >> 3);
    if (method == BO_GINX) {
        c->mono = (u64*)malloc(sizeof(u64) * 2ull * N * N);
        for (u32 k = 0; k < 2 * N; k++) {
            u64* mp = c->mono + (u64)k * N;
            memset(mp, 0, sizeof(u64) * N);
            if (k < N) mp[k] = 1; else mp[k - N] = Q - 1;          /* +-X^k */
            mp[0] = mp[0] ? mp[0] - 1 : Q - 1;                     /* -1 (mod Q) */
            bo_ntt_forward(c, mp);
        }
    }
    return c;
}

/* upstream: binfhe/lib/binfhecontext.cpp GenerateBinFHEContext(set, method) parameter table
 * { numberBits, cyclOrder, latticeParam n, mod q, modKS, baseKS, gadgetBase, baseRK } */
bo_ctx* bo_ctx_create(int set, int method) {
    struct row { u32 bits, M, n; u64 q, qKS; u32 baseKS, baseG, baseR; };
    static const struct row T[] = {
        /* TOY          */ {27, 1024, 64, 512, 0 /*PRIME*/, 25, 1u << 9, 23},
        /* MEDIUM       */ {28, 2048, 422, 1024, 1u << 14, 1u << 7, 1u << 10, 32},
        /* STD128_AP    */ {27, 2048, 512, 1024, 1u << 14, 1u << 7, 1u << 9, 32},
        /* STD128_APOPT */ {27, 2048, 502, 1024, 1u << 14, 1u << 7, 1u << 9, 32},
        /* STD128       */ {27, 2048, 512, 1024, 1u << 14, 1u << 7, 1u << 7, 32},
        /* STD128_OPT   */ {27, 2048, 502, 1024, 1u << 14, 1u << 7, 1u << 7, 32},
        /* STD192       */ {37, 4096, 1024, 1024, 1u << 19, 28, 1u << 13, 32},
        /* STD192_OPT   */ {37, 4096, 805, 1024, 1u << 15, 32, 1u << 13, 32},
        /* STD256       */ {29, 4096, 1024, 2048, 1u << 14, 1u << 7, 1u << 8, 46},
        /* STD256_OPT   */ {29, 4096, 990, 2048, 1u << 14, 1u << 7, 1u << 8, 46},
    };
    if (set < 0 || set > BO_STD256_OPT) return NULL;
    if (method != BO_AP && method != BO_GINX) return NULL;
    const struct row* r = &T[set];
    /* Q = PreviousPrime(FirstPrime(numberBits, cyclOrder), cyclOrder) */
    u64 Q = bo_previous_prime(bo_first_prime(r->bits, r->M), r->M);
    u64 qKS = r->qKS ? r->qKS : Q;
    return ctx_build(r->n, r->M / 2, r->q, Q, qKS, r->baseKS, r->baseG, r->baseR, method);
}

bo_ctx* bo_ctx_create_custom(uint32_t n, uint32_t N, uint64_t q, uint64_t Q, uint64_t qKS, uint32_t baseKS,
                             uint32_t baseG, uint32_t baseR, int method) {
    if (method != BO_AP && method != BO_GINX) return NULL;
    if (!is_prime(Q) || (Q - 1) % (2ull * N) != 0) return NULL;
    return ctx_build(n, N, q, Q, qKS ? qKS : Q, baseKS, baseG, baseR, method);
}

void bo_ctx_destroy(bo_ctx* c) {
    if (!c) return;
    free(c->tw); free(c->tws); free(c->itw); free(c->itws); free(c->Gpow); free(c->mono);
    free(c->s); free(c->z); free(c->bsk); free(c->ksk);
    free(c);
}

void bo_get_params(const bo_ctx* c, uint64_t out[BO_P_COUNT]) {
    out[BO_P_n] = c->n; out[BO_P_N] = c->N; out[BO_P_q] = c->q; out[BO_P_Q] = c->Q; out[BO_P_qKS] = c->qKS;
    out[BO_P_baseKS] = c->baseKS; out[BO_P_dKS] = c->dKS; out[BO_P_baseG] = c->baseG; out[BO_P_dG] = c->dG;
    out[BO_P_baseR] = c->baseR; out[BO_P_dR] = c->dR; out[BO_P_method] = (u64)c->method; out[BO_P_psi] = c->psi;
}

/* ------------------------------------------------------------------ */
/* key generation                                                      */
/* ------------------------------------------------------------------ */
static inline u64 lift(int v, u64 M) { return v >= 0 ? (u64)v : M - (u64)(-v); }

/* One RGSW row pair list for message monomial +-X^mm * [msg != 0].
 * upstream: rgsw-acc-cggi.cpp KeyGenCGGI / rgsw-acc-dm.cpp KeyGenDM.
 * out: 2*dG rows x 2 polys, EVALUATION format. */
static void rgsw_encrypt(const bo_ctx* c, const u64* zntt, int nonzero, u32 mm, int negate, u64 stream_base, u64* out) {
    const u32 N = c->N, R = 2 * c->dG;
    const u64 Q = c->Q;
    u64* tmpA = (u64*)malloc(sizeof(u64) * N);
    for (u32 r = 0; r < R; r++) {
        stream_t st;
        stream_init(&st, c->seed, DOM_BSK, stream_base + r);
        u64* a = out + ((u64)r * 2 + 0) * N;
        u64* b = out + ((u64)r * 2 + 1) * N;
        for (u32 k = 0; k < N; k++) a[k] = sample_uniform(&st, Q);
        for (u32 k = 0; k < N; k++) b[k] = lift(sample_gauss(&c->dgg, &st), Q);
        memcpy(tmpA, a, sizeof(u64) * N);
        if (nonzero) {
            u64 g = c->Gpow[r >> 1];
            u64* tgt = (r & 1) ? b : a; /* row 2i: column 0; row 2i+1: column 1 */
            if (!negate) tgt[mm] = (tgt[mm] + g) % Q;
            else tgt[mm] = (tgt[mm] + Q - g) % Q;
        }
        bo_ntt_forward(c, a);
        bo_ntt_forward(c, b);
        bo_ntt_forward(c, tmpA);
        for (u32 k = 0; k < N; k++) b[k] = (b[k] + mulmodQ(c, tmpA[k], zntt[k])) % Q;
    }
    free(tmpA);
}

void bo_keygen(bo_ctx* c, const uint8_t seed[32]) {
    memcpy(c->seed, seed, 32);
    const u32 n = c->n, N = c->N;
    free(c->s); free(c->z); free(c->bsk); free(c->ksk);
    c->s = (int32_t*)malloc(sizeof(int32_t) * n);
    c->z = (int32_t*)malloc(sizeof(int32_t) * N);
    stream_t st;
    /* upstream: lwe-pke.cpp KeyGen (ternary uniform, n) and KeyGenN (N) */
    stream_init(&st, seed, DOM_SK, 0);
    for (u32 i = 0; i < n; i++) c->s[i] = sample_ternary(&st);
    stream_init(&st, seed, DOM_Z, 0);
    for (u32 i = 0; i < N; i++) c->z[i] = sample_ternary(&st);

    /* upstream: lwe-pke.cpp KeySwitchGen: K[i][v][j] = LWE_s( z_i * v * baseKS^j ) mod qKS */
    const u64 qKS = c->qKS;
    const u32 B = c->baseKS, D = c->dKS;
    c->ksk = (u32*)malloc(sizeof(u32) * (u64)N * B * D * (n + 1));
    u64* digitsKS = (u64*)malloc(sizeof(u64) * D);
    { u64 v = 1; for (u32 j = 0; j < D; j++) { digitsKS[j] = v; v *= B; } }
#pragma omp parallel for schedule(dynamic, 4)
    for (u32 i = 0; i < N; i++) {
        u64 zi = lift(c->z[i], qKS);
        for (u32 v = 0; v < B; v++)
            for (u32 j = 0; j < D; j++) {
                stream_t ks;
                u64 idx = ((u64)i * B + v) * D + j;
                stream_init(&ks, seed, DOM_KSK, idx);
                u32* row = c->ksk + idx * (n + 1);
                u128 acc = 0;
                for (u32 k = 0; k < n; k++) {
                    u64 a = sample_uniform(&ks, qKS);
                    row[k] = (u32)a;
                    acc += (u128)a * lift(c->s[k], qKS);
                }
                u64 e = lift(sample_gauss(&c->dgg, &ks), qKS);
                u64 msg = (u64)((u128)zi * ((u128)v * digitsKS[j] % qKS) % qKS);
                row[n] = (u32)((u64)((acc + e + msg) % qKS));
            }
    }
    free(digitsKS);

    /* ring key in EVALUATION format (upstream: binfhe-base-scheme.cpp KeyGen) */
    u64* zntt = (u64*)malloc(sizeof(u64) * N);
    for (u32 i = 0; i < N; i++) zntt[i] = lift(c->z[i], c->Q);
    bo_ntt_forward(c, zntt);

    const u32 R = 2 * c->dG;
    const u64 rgsw_words = (u64)R * 2 * N;
    if (c->method == BO_GINX) {
        /* upstream: rgsw-acc-cggi.cpp KeyGenAcc: ek[0][0][i] = RGSW(s_i == 1), ek[0][1][i] = RGSW(s_i == -1) */
        c->bsk_polys = (u64)n * 2 * R * 2;
        c->bsk = (u64*)malloc(sizeof(u64) * c->bsk_polys * N);
#pragma omp parallel for schedule(dynamic, 2)
        for (u32 i = 0; i < n; i++)
            for (u32 key = 0; key < 2; key++) {
                int m = (key == 0) ? (c->s[i] == 1) : (c->s[i] == -1);
                u64 id = (u64)i * 2 + key;
                rgsw_encrypt(c, zntt, m, 0, 0, id * R, c->bsk + id * rgsw_words);
            }
    } else {
        /* upstream: rgsw-acc-dm.cpp KeyGenAcc: ek[i][v][k] = RGSW(X^{s_i * v * baseR^k * (2N/q)}), v >= 1 */
        const u32 BR = c->baseR, DR = c->dR;
        c->bsk_polys = (u64)n * BR * DR * R * 2;
        c->bsk = (u64*)calloc(c->bsk_polys * N, sizeof(u64));
        int64_t q = (int64_t)c->q;
#pragma omp parallel for schedule(dynamic, 1)
        for (u32 i = 0; i < n; i++)
            for (u32 v = 1; v < BR; v++) {
                int64_t pw = 1;
                for (u32 k = 0; k < DR; k++, pw *= BR) {
                    int64_t m = (int64_t)c->s[i] * (int64_t)v * pw;
                    int64_t mm = (((m % q) + q) % q) * (int64_t)(2 * N / q);
                    int neg = 0;
                    if (mm >= (int64_t)N) { mm -= N; neg = 1; }
                    u64 id = ((u64)i * BR + v) * DR + k;
                    rgsw_encrypt(c, zntt, 1, (u32)mm, neg, id * R, c->bsk + id * rgsw_words);
                }
            }
    }
    free(zntt);
    c->have_keys = 1;
}

void bo_export_sk(const bo_ctx* c, int32_t* s) { memcpy(s, c->s, sizeof(int32_t) * c->n); }
void bo_export_z(const bo_ctx* c, int32_t* z) { memcpy(z, c->z, sizeof(int32_t) * c->N); }
uint64_t bo_bsk_words(const bo_ctx* c) { return c->bsk_polys * c->N; }
void bo_export_bsk(const bo_ctx* c, uint64_t* out) {
    const u32 N = c->N;
#pragma omp parallel for schedule(static)
    for (u64 p = 0; p < c->bsk_polys; p++) {
        memcpy(out + p * N, c->bsk + p * N, sizeof(u64) * N);
        bo_ntt_inverse(c, out + p * N);
    }
}
uint64_t bo_ksk_words(const bo_ctx* c) { return (u64)c->N * c->baseKS * c->dKS * (c->n + 1); }
void bo_export_ksk(const bo_ctx* c, uint32_t* out) { memcpy(out, c->ksk, sizeof(u32) * bo_ksk_words(c)); }

/* ------------------------------------------------------------------ */
/* LWE layer (upstream: binfhe/lib/lwe-pke.cpp)                        */
/* ------------------------------------------------------------------ */
void bo_encrypt(const bo_ctx* c, int bit, uint64_t enc_index, uint64_t* ct) {
    const u32 n = c->n;
    const u64 q = c->q;
    stream_t st;
    stream_init(&st, c->seed, DOM_ENC, enc_index);
    u128 acc = 0;
    for (u32 i = 0; i < n; i++) {
        ct[i] = sample_uniform(&st, q);
        acc += (u128)ct[i] * lift(c->s[i], q);
    }
    u64 e = lift(sample_gauss(&c->dgg, &st), q);
    u64 m = ((u64)(bit % 4)) * (q / 4);
    ct[n] = (u64)((acc + e + m) % q);
}

static u64 lwe_phase(const bo_ctx* c, const u64* ct) {
    const u32 n = c->n;
    const u64 q = c->q;
    u128 inner = 0;
    for (u32 i = 0; i < n; i++) inner += (u128)ct[i] * lift(c->s[i], q);
    u64 in = (u64)(inner % q);
    return (ct[n] + q - in) % q;
}

int bo_decrypt(const bo_ctx* c, const uint64_t* ct) {
    const u64 q = c->q;
    u64 r = (lwe_phase(c, ct) + q / 8) % q; /* Round(4/q x) = Floor(4/q (x + q/8)) */
    return (int)((4 * r) / q);
}

int64_t bo_noise(const bo_ctx* c, const uint64_t* ct, int bit) {
    const u64 q = c->q;
    u64 r = (lwe_phase(c, ct) + q - (u64)bit * (q / 4)) % q;
    return r > q / 2 ? (int64_t)r - (int64_t)q : (int64_t)r;
}

/* upstream: binfhe-base-scheme.cpp EvalNOT: (-a, q/4 - b) */
void bo_eval_not(const bo_ctx* c, const uint64_t* ct, uint64_t* out) {
    const u64 q = c->q;
    for (u32 i = 0; i < c->n; i++) out[i] = ct[i] ? q - ct[i] : 0;
    out[c->n] = ((q >> 2) + q - ct[c->n]) % q;
}

/* ------------------------------------------------------------------ */
/* accumulator (upstream: rgsw-acc.cpp, rgsw-acc-cggi.cpp, rgsw-acc-dm.cpp) */
/* ------------------------------------------------------------------ */

/* SignedDigitDecompose: input 2 polys COEFFICIENT, output 2*dG polys, out[2l + j] */
static void signed_digit_decompose(const bo_ctx* c, const u64* ct /*[2][N]*/, u64* dct /*[2dG][N]*/) {
    const u32 N = c->N, dG = c->dG;
    const int64_t Q = (int64_t)c->Q;
    const u64 QHalf = c->Q >> 1;
    const int gBits = (int)c->gBits;
    const int sh = 64 - gBits;
    for (u32 j = 0; j < 2; j++)
        for (u32 k = 0; k < N; k++) {
            u64 t = ct[(u64)j * N + k];
            int64_t d = (t < QHalf) ? (int64_t)t : (int64_t)t - Q;
            for (u32 l = 0; l < dG; l++) {
                int64_t r = (int64_t)((u64)d << sh) >> sh; /* signed remainder in [-B/2, B/2) */
                d -= r;
                d >>= gBits;
                dct[((u64)(2 * l + j)) * N + k] = r >= 0 ? (u64)r : (u64)(r + Q);
            }
        }
}

/* test hook: SignedDigitDecompose of one RLWE pair, ct [2][N] coefficient form -> dct [2 dG][N] (digits mod Q) */
void bo_signed_digit_decompose(const bo_ctx* c, const uint64_t* ct, uint64_t* dct) { signed_digit_decompose(c, ct, dct); }

typedef struct {
    u64 *ct, *dct;
} scratch_t;

static void scratch_alloc(const bo_ctx* c, scratch_t* s) {
    s->ct = (u64*)malloc(sizeof(u64) * 2 * c->N);
    s->dct = (u64*)malloc(sizeof(u64) * 2 * c->dG * c->N);
}
static void scratch_free(scratch_t* s) { free(s->ct); free(s->dct); }

static void decompose_acc(const bo_ctx* c, const u64* acc, scratch_t* s) {
    const u32 N = c->N, R = 2 * c->dG;
    memcpy(s->ct, acc, sizeof(u64) * 2 * N);
    bo_ntt_inverse(c, s->ct);
    bo_ntt_inverse(c, s->ct + N);
    signed_digit_decompose(c, s->ct, s->dct);
    for (u32 l = 0; l < R; l++) bo_ntt_forward(c, s->dct + (u64)l * N);
}

/* rgsw-acc-cggi.cpp AddToAcc: acc += (dct x ek1) * mono[a] + (dct x ek2) * mono[-a] */
static void add_to_acc_cggi(const bo_ctx* c, const u64* ek1, const u64* ek2, u64 a, u64* acc, scratch_t* s) {
    const u32 N = c->N, R = 2 * c->dG;
    const u64 M = 2ull * N;
    decompose_acc(c, acc, s);
    u64 ipos = a % M;
    u64 ineg = (M - ipos) % M;
    const u64* mp = c->mono + ipos * N;
    const u64* mn = c->mono + ineg * N;
    for (u32 j = 0; j < 2; j++) {
        u64* aj = acc + (u64)j * N;
        for (u32 k = 0; k < N; k++) {
            u128 t1 = 0, t2 = 0;
            for (u32 l = 0; l < R; l++) {
                u64 d = s->dct[(u64)l * N + k];
                t1 += (u128)d * ek1[((u64)l * 2 + j) * N + k];
                t2 += (u128)d * ek2[((u64)l * 2 + j) * N + k];
            }
            u64 r1 = barrett128(c, t1), r2 = barrett128(c, t2);
            u128 t = (u128)r1 * mp[k] + (u128)r2 * mn[k] + aj[k];
            aj[k] = barrett128(c, t);
        }
    }
}

/* rgsw-acc-dm.cpp AddToAcc: acc = dct x ek */
static void add_to_acc_dm(const bo_ctx* c, const u64* ek, u64* acc, scratch_t* s) {
    const u32 N = c->N, R = 2 * c->dG;
    decompose_acc(c, acc, s);
    for (u32 j = 0; j < 2; j++) {
        u64* aj = acc + (u64)j * N;
        for (u32 k = 0; k < N; k++) {
            u128 t = 0;
            for (u32 l = 0; l < R; l++) t += (u128)s->dct[(u64)l * N + k] * ek[((u64)l * 2 + j) * N + k];
            aj[k] = barrett128(c, t);
        }
    }
}

/* binfhe-base-scheme.cpp BootstrapGateCore + {cggi,dm} EvalAcc. acc: EVALUATION format [2][N] */
static void bootstrap_gate_core(const bo_ctx* c, int gate, const u64* ctprep, u64* acc, scratch_t* s) {
    const u32 N = c->N, n = c->n;
    const u64 q = c->q, Q = c->Q;
    const u32 qHalf = (u32)(q >> 1);
    const u64 q1 = c->gateConst[gate];
    const u64 q2 = (q1 + qHalf) % q;
    const u64 Q2p = Q / 8 + 1, Q2pNeg = Q - Q2p;
    const u32 factor = (u32)(2 * N / q);
    const u64 b = ctprep[n];
    memset(acc, 0, sizeof(u64) * 2 * N);
    u64* m = acc + N;
    for (u32 j = 0; j < qHalf; j++) {
        u64 temp = (b + q - j) % q;
        if (q1 < q2) m[j * factor] = (temp >= q1 && temp < q2) ? Q2pNeg : Q2p;
        else m[j * factor] = (temp >= q2 && temp < q1) ? Q2p : Q2pNeg;
    }
    bo_ntt_forward(c, m);

    const u64 rg = (u64)2 * c->dG * 2 * N; /* words per RGSW */
    if (c->method == BO_GINX) {
        /* rgsw-acc-cggi.cpp EvalAcc */
        for (u32 i = 0; i < n; i++) {
            u64 aI = ((q - ctprep[i]) % q) * factor;
            add_to_acc_cggi(c, c->bsk + ((u64)i * 2 + 0) * rg, c->bsk + ((u64)i * 2 + 1) * rg, aI, acc, s);
        }
    } else {
        /* rgsw-acc-dm.cpp EvalAcc */
        for (u32 i = 0; i < n; i++) {
            u64 aI = (q - ctprep[i]) % q;
            for (u32 k = 0; k < c->dR; k++, aI /= c->baseR) {
                u32 a0 = (u32)(aI % c->baseR);
                if (a0) add_to_acc_dm(c, c->bsk + (((u64)i * c->baseR + a0) * c->dR + k) * rg, acc, s);
            }
        }
    }
}

/* binfhe-base-scheme.cpp EvalBinGate prep: ct1 + ct2, or 2*(ct1 - ct2) for XOR_FAST/XNOR_FAST */
void bo_gate_prep(const bo_ctx* c, int gate, const uint64_t* ct1, const uint64_t* ct2, uint64_t* out) {
    const u64 q = c->q;
    for (u32 i = 0; i <= c->n; i++) {
        if (gate == BO_XOR_FAST || gate == BO_XNOR_FAST) {
            u64 d = (ct1[i] + q - ct2[i]) % q;
            out[i] = (2 * d) % q;
        } else {
            out[i] = (ct1[i] + ct2[i]) % q;
        }
    }
}

static void acc_to_coeff(const bo_ctx* c, u64* acc) {
    bo_ntt_inverse(c, acc);
    bo_ntt_inverse(c, acc + c->N);
}

void bo_blind_rotate(const bo_ctx* c, int gate, const uint64_t* ctprep, uint64_t* acc) {
    scratch_t s;
    scratch_alloc(c, &s);
    bootstrap_gate_core(c, gate, ctprep, acc, &s);
    acc_to_coeff(c, acc);
    scratch_free(&s);
}

/* lwe-pke.cpp RoundqQ: literal double arithmetic in this order */
static inline u64 round_qQ(u64 v, u64 q, u64 Q) {
    return (u64)floor(0.5 + (double)v * (double)q / (double)Q) % q;
}

/* tail of EvalBinGate: Transpose(acc[0]) (X -> X^{-1}), b = acc[1][0] + Q/8 + 1, ModSwitch(Q -> qKS).
 * acc in COEFFICIENT format. */
void bo_extract_modswitch(const bo_ctx* c, const uint64_t* acc, uint64_t* lweN) {
    const u32 N = c->N;
    const u64 Q = c->Q, qKS = c->qKS;
    /* a'(X) = a(X^{-1}): a'_0 = a_0, a'_{N-i} = -a_i */
    lweN[0] = round_qQ(acc[0], qKS, Q);
    for (u32 i = 1; i < N; i++) {
        u64 v = acc[i] ? Q - acc[i] : 0;
        lweN[N - i] = round_qQ(v, qKS, Q);
    }
    u64 b = (Q / 8 + 1 + acc[N]) % Q;
    lweN[N] = round_qQ(b, qKS, Q);
}

/* lwe-pke.cpp KeySwitch */
void bo_keyswitch(const bo_ctx* c, const uint64_t* lweN, uint64_t* out) {
    const u32 n = c->n, N = c->N, B = c->baseKS, D = c->dKS;
    const u64 qKS = c->qKS;
    memset(out, 0, sizeof(u64) * n);
    u64 b = lweN[N];
    for (u32 i = 0; i < N; i++) {
        u64 at = lweN[i];
        for (u32 j = 0; j < D; j++, at /= B) {
            u32 a0 = (u32)(at % B);
            const u32* row = c->ksk + (((u64)i * B + a0) * D + j) * (n + 1);
            for (u32 k = 0; k < n; k++) {
                u64 v = out[k] + qKS - row[k];
                out[k] = v >= qKS ? v - qKS : v;
            }
            b = (b + qKS - row[n]) % qKS;
        }
    }
    out[n] = b;
}

void bo_modswitch_final(const bo_ctx* c, const uint64_t* in, uint64_t* out) {
    for (u32 i = 0; i <= c->n; i++) out[i] = round_qQ(in[i], c->q, c->qKS);
}

static void gate_from_prep(const bo_ctx* c, int gate, const u64* ctprep, u64* out, scratch_t* s, u64* acc, u64* lweN, u64* ks) {
    bootstrap_gate_core(c, gate, ctprep, acc, s);
    acc_to_coeff(c, acc);
    bo_extract_modswitch(c, acc, lweN);
    bo_keyswitch(c, lweN, ks);
    bo_modswitch_final(c, ks, out);
}

typedef struct {
    scratch_t s;
    u64 *acc, *lweN, *ks, *prep, *t0, *t1;
} work_t;
static void work_alloc(const bo_ctx* c, work_t* w) {
    scratch_alloc(c, &w->s);
    w->acc = (u64*)malloc(sizeof(u64) * 2 * c->N);
    w->lweN = (u64*)malloc(sizeof(u64) * (c->N + 1));
    w->ks = (u64*)malloc(sizeof(u64) * (c->n + 1));
    w->prep = (u64*)malloc(sizeof(u64) * (c->n + 1));
    w->t0 = (u64*)malloc(sizeof(u64) * (c->n + 1));
    w->t1 = (u64*)malloc(sizeof(u64) * (c->n + 1));
}
static void work_free(work_t* w) {
    scratch_free(&w->s);
    free(w->acc); free(w->lweN); free(w->ks); free(w->prep); free(w->t0); free(w->t1);
}

/* binfhe-base-scheme.cpp EvalBinGate */
void bo_eval_bingate(const bo_ctx* c, int gate, const uint64_t* ct1, const uint64_t* ct2, uint64_t* out) {
    work_t w;
    work_alloc(c, &w);
    bo_gate_prep(c, gate, ct1, ct2, w.prep);
    gate_from_prep(c, gate, w.prep, out, &w.s, w.acc, w.lweN, w.ks);
    work_free(&w);
}

/* binfhe-base-scheme.cpp Bootstrap: ctprep = ct + q/4 (b only), gate constant AND */
void bo_bootstrap(const bo_ctx* c, const uint64_t* ct, uint64_t* out) {
    work_t w;
    work_alloc(c, &w);
    memcpy(w.prep, ct, sizeof(u64) * (c->n + 1));
    w.prep[c->n] = (w.prep[c->n] + (c->q >> 2)) % c->q;
    gate_from_prep(c, BO_AND, w.prep, out, &w.s, w.acc, w.lweN, w.ks);
    work_free(&w);
}

uint64_t bo_eval_gates(const bo_ctx* c, uint64_t* pool, uint32_t n_desc, const bo_gate_desc* d, int nthreads) {
    const u32 W = c->n + 1;
    u64 nboot = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads) reduction(+ : nboot)
    {
        work_t w;
        work_alloc(c, &w);
#pragma omp for schedule(dynamic, 1)
        for (u32 g = 0; g < n_desc; g++) {
            const bo_gate_desc* e = &d[g];
            const u64* i0 = pool + (u64)e->in0 * W;
            const u64* i1 = pool + (u64)e->in1 * W;
            u64* o = pool + (u64)e->out * W;
            if (e->neg0) { bo_eval_not(c, i0, w.t0); i0 = w.t0; }
            if (e->op == BO_OP_NOT) { bo_eval_not(c, i0, w.t1); memcpy(o, w.t1, sizeof(u64) * W); continue; }
            if (e->op == BO_OP_COPY) { memmove(o, i0, sizeof(u64) * W); continue; }
            if (e->op == BO_OP_REFRESH) {
                memcpy(w.prep, i0, sizeof(u64) * W);
                w.prep[c->n] = (w.prep[c->n] + (c->q >> 2)) % c->q;
                gate_from_prep(c, BO_AND, w.prep, w.t1, &w.s, w.acc, w.lweN, w.ks);
                memcpy(o, w.t1, sizeof(u64) * W);
                nboot++;
                continue;
            }
            if (e->neg1) { bo_eval_not(c, i1, w.t1); i1 = w.t1; }
            bo_gate_prep(c, (int)e->op, i0, i1, w.prep);
            gate_from_prep(c, (int)e->op, w.prep, w.t0, &w.s, w.acc, w.lweN, w.ks);
            memcpy(o, w.t0, sizeof(u64) * W);
            nboot++;
        }
        work_free(&w);
    }
    return nboot;
}
