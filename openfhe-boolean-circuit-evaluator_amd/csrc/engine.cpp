// engine.cpp -- host side of the C ABI in include/bce_gpu.h.
//
// Replaces what the reference obtains from lbcrypto::BinFHEContext (OpenFHE):
// GenerateBinFHEContext / KeyGen / BTKeyGen (src/circuit.cpp:88-91), Encrypt / Decrypt
// (src/circuit.cpp:506,800) and the per-gate EvalBinGate / EvalNOT calls
// (src/gate.cpp:112,133,172,198-202), the latter batched per ready frontier.
// There is no CPU compute path here: without a HIP device every call fails loudly.
#include <hip/hip_runtime_api.h>
#include <sys/random.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_set>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/bce_circuit.h"
#include "../../include/bce_gpu.h"
#include "host_math.hpp"
#include "kernels.hpp"
#include "keygen.hpp"
#include "prng.hpp"

using namespace bce;

namespace {

thread_local std::string g_create_error;

struct ParamRow { u32 bits, cyclo, n; u64 q, qKS; u32 baseKS, baseG, baseR; };
// OpenFHE v1.0.x GenerateBinFHEContext table: {numberBits, cyclOrder, latticeParam, mod, modKS(0 = PRIME), baseKS, gadgetBase, baseRK}
const ParamRow kParamTable[] = {
    {27, 1024, 64, 512, 0, 25, 1u << 9, 23},              // TOY
    {28, 2048, 422, 1024, 1u << 14, 1u << 7, 1u << 10, 32},  // MEDIUM
    {27, 2048, 512, 1024, 1u << 14, 1u << 7, 1u << 9, 32},   // STD128_AP
    {27, 2048, 502, 1024, 1u << 14, 1u << 7, 1u << 9, 32},   // STD128_APOPT
    {27, 2048, 512, 1024, 1u << 14, 1u << 7, 1u << 7, 32},   // STD128
    {27, 2048, 502, 1024, 1u << 14, 1u << 7, 1u << 7, 32},   // STD128_OPT
    {37, 4096, 1024, 1024, 1u << 19, 28, 1u << 13, 32},      // STD192
    {37, 4096, 805, 1024, 1u << 15, 32, 1u << 13, 32},       // STD192_OPT
    {29, 4096, 1024, 2048, 1u << 14, 1u << 7, 1u << 8, 46},  // STD256
    {29, 4096, 990, 2048, 1u << 14, 1u << 7, 1u << 8, 46},   // STD256_OPT
};

struct EventPair { hipEvent_t a, b; int kind; };

// 32 bytes from the operating system's entropy pool (getrandom, then /dev/urandom).  There is no fixed
// fallback value: a context that cannot get entropy fails to draw keys / encryption randomness.
bool os_entropy(uint8_t out[32]) {
    size_t got = 0;
    while (got < 32) {
        const ssize_t r = getrandom(out + got, 32 - got, 0);
        if (r <= 0) break;
        got += (size_t)r;
    }
    if (got == 32) return true;
    FILE* f = std::fopen("/dev/urandom", "rb");
    if (!f) return false;
    const size_t n = std::fread(out, 1, 32, f);
    std::fclose(f);
    return n == 32;
}

}  // namespace

// Contexts that exist, and the schedules each one owns.  A bce_plan / bce_dag belongs to the context it was created on: the
// context's destruction releases whatever its callers left behind, and bce_plan_destroy / bce_dag_destroy on a context that is
// already gone (a Circuit destroyed after its engine) is a no-op instead of a read of freed memory.
struct bce_plan;
struct bce_dag;
namespace {
std::mutex g_live_mu;
std::unordered_set<const bce_ctx*> g_live;
}  // namespace

struct bce_ctx {
    std::unordered_set<bce_plan*> plans;
    std::unordered_set<bce_dag*> dags;
    // parameters
    u32 n = 0, N = 0, logN = 0;
    u64 q = 0, Q = 0, qKS = 0, psi = 0;
    u32 baseKS = 0, dKS = 0, baseG = 0, gBits = 0, dG = 0, baseR = 0, dR = 0;
    int method = 0, device = 0;
    std::string err;

    hipStream_t stream = nullptr;
    DevParams P{};
    // device tables / keys
    uint2* d_twf = nullptr;
    u32* d_psi = nullptr;
    u32* d_psi_r2 = nullptr;
    u32* d_xcd_gate = nullptr;
    void* d_bsk = nullptr;       // u32 words (Q < 2^28) or u64 words (is64)
    ulonglong2* d_tw64 = nullptr;
    double2* d_tw64d = nullptr;   // (w, w / Q) for the double-precision formulation
    bool is64 = false;
    size_t wbytes = 4;
    void* d_ksk = nullptr;
    u64 bsk_polys = 0;
    bool have_keys = false;
    // host secrets
    std::vector<int32_t> s, z;
    uint8_t seed[32] = {0};       // key-generation seed (bce_keygen); zero after bce_import_keys
    // Encryption randomness is independent of the key seed: drawn from OS entropy when the context is created,
    // consumed through a per-context counter that only moves forward, so no (a, e) pair is ever reused.
    // bce_set_encrypt_seed() switches to the deterministic, caller-indexed streams of the parity tests.
    uint8_t enc_seed[32] = {0};
    bool enc_seed_ok = false, enc_deterministic = false;
    uint64_t enc_counter = 0;
    void* rccl_comm = nullptr;   // ncclComm_t of the in-library all-gather (rccl_xchg.cpp), if enabled
    // pool
    u32* d_pool = nullptr;
    u32 pool_slots = 0;
    // work buffers
    void* d_acc = nullptr;
    size_t acc_cap = 0;  // bootstraps
    u64* d_tail_partial = nullptr;  // partial key-switch sums (kernels.hip, k_tail_gather)
    bool events_on = true;                 // per-launch HIP events (bce_timing_set_events): off = counters only, no event packets between dependent kernels
    u32 *d_io = nullptr, *h_io = nullptr;  // staging of bce_lwe_read for scattered slots (device gather + one pinned copy)
    size_t io_cap = 0;
    size_t tail_cap = 0;            // u64 words
    static constexpr int kRing = 4;
    bce_gate_desc* d_descs[kRing] = {nullptr, nullptr, nullptr, nullptr};
    bce_gate_desc* h_descs[kRing] = {nullptr, nullptr, nullptr, nullptr};
    size_t desc_cap[kRing] = {0, 0, 0, 0};
    hipEvent_t ring_ev[kRing] = {nullptr, nullptr, nullptr, nullptr};
    bool ring_busy[kRing] = {false, false, false, false};
    int ring_pos = 0;
    // timing
    std::vector<EventPair> pending, free_events;
    bce_timing timing{};
    // dependency-driven runs (bce_dag_*): device copy of P, knobs, and the status words of runs not yet checked
    DevParams* d_P = nullptr;
    int dag_wg_per_cu = 0, dag_placement = 1;
    uint32_t dag_lazy_us = 20, dag_stall_ms = 4000;
    struct DagStatus { uint32_t abort, done, lazy_waits, pad; uint64_t busy_ticks, wait_ticks, gate_ticks; };
    static constexpr int kDagRuns = 32;
    DagStatus* h_dag_status = nullptr;        // pinned, kDagRuns entries
    struct DagStage { DevParams P; DagParams D; };
    DagStage* h_dag_stage = nullptr;          // pinned, kDagRuns entries: sources of the stream-ordered parameter uploads
    uint64_t dag_expected[kDagRuns] = {0};
    int dag_wps_used[kDagRuns] = {0};
    int dag_pending = 0;
    uint64_t dag_last[7] = {0, 0, 0, 0, 0, 0, 0};

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define HIP_TRY(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) return (ctx)->fail(BCE_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

namespace {

int build_ctx(u32 n, u32 N, u64 q, u64 Q, u64 qKS, u32 baseKS, u32 baseG, u32 baseR, int method, int device,
              bce_ctx** out) {
    if (!out) return BCE_ERR_ARG;
    *out = nullptr;
    if (method != BCE_AP && method != BCE_GINX) { g_create_error = "bad method (expect AP=1 or GINX=2)"; return BCE_ERR_ARG; }
    if (N < 512 || N > 2048 || (N & (N - 1))) { g_create_error = "ring dimension N must be 512, 1024 or 2048"; return BCE_ERR_UNSUPPORTED; }
    if ((q & (q - 1)) || q > 2 * (u64)N || q < 8) { g_create_error = "LWE modulus q must be a power of two dividing 2N"; return BCE_ERR_ARG; }
    if (!is_prime_u64(Q) || (Q - 1) % (2ull * N)) { g_create_error = "Q must be a prime = 1 mod 2N"; return BCE_ERR_ARG; }
    if (Q >= (1ull << 40)) { g_create_error = "ring modulus Q must be below 2^40"; return BCE_ERR_UNSUPPORTED; }
    if (baseG & (baseG - 1)) { g_create_error = "gadget base must be a power of two"; return BCE_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "no HIP device visible: the engine has no CPU fallback";
        return BCE_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { g_create_error = "bad device ordinal"; return BCE_ERR_ARG; }

    // partially built contexts are released through bce_ctx_destroy (frees whatever was allocated)
    struct CtxDeleter { void operator()(bce_ctx* p) const { bce_ctx_destroy(p); } };
    std::unique_ptr<bce_ctx, CtxDeleter> c(new bce_ctx);
    { std::lock_guard<std::mutex> lk(g_live_mu); g_live.insert(c.get()); }
    c->n = n; c->N = N; c->q = q; c->Q = Q; c->qKS = qKS ? qKS : Q;
    c->baseKS = baseKS; c->baseG = baseG; c->baseR = baseR; c->method = method; c->device = device;
    while ((1u << c->logN) < N) ++c->logN;
    while ((1u << c->gBits) < baseG) ++c->gBits;
    c->dKS = digit_count((double)c->qKS, (double)baseKS);
    c->dG = digit_count((double)Q, (double)baseG);
    c->dR = digit_count((double)q, (double)baseR);
    c->is64 = Q >= (1ull << 28);
    c->wbytes = c->is64 ? 8 : 4;
    // four gadget digits on N >= 1024 with a modulus of 28..30 bits (STD256, STD256_OPT: N = 2048, 29-bit Q, base 2^8): the
    // integer 64-bit kernel with 32-bit digit rows (kernels64.hip, NARROW)
    const bool narrow64 = c->is64 && c->dG == 4 && N >= 1024 && Q < (1ull << 31);
    if (c->dG < 3 || c->dG > 4 || (N == 2048 && c->dG != 3 && !narrow64)) { g_create_error = "gadget digit count must be 3 or 4 (N = 2048: 3, or 4 with a ring modulus of 28..30 bits)"; return BCE_ERR_UNSUPPORTED; }
    if (c->is64 && !(c->dG == 3 || (c->dG == 4 && N == 512) || narrow64)) { g_create_error = "64-bit path: four gadget digits need N = 512 or a ring modulus below 2^31"; return BCE_ERR_UNSUPPORTED; }
    if (c->qKS > 0xFFFFFFFFull) { g_create_error = "qKS must fit 32 bits"; return BCE_ERR_UNSUPPORTED; }
    c->psi = min_primitive_root(Q, 2ull * N);

    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return BCE_ERR_HIP; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_create_error = "hipStreamCreate failed"; return BCE_ERR_HIP; }

    // twiddle tables, OpenFHE ordering: tw[brv(i)] = psi^i
    std::vector<uint2> twf(N);
    std::vector<u32> psitab(N);
    u64 p = 1;
    for (u32 i = 0; i < N; ++i) {
        psitab[i] = c->is64 ? 0 : (u32)p;
        u32 r = bit_reverse(i, (int)c->logN);
        // device layout: blocks m >= 64 transposed to [slot][lane] (kernels.hip tw_pos)
        u32 pos = r;
        if (r >= 64) {
            const u32 s = 31u - (u32)__builtin_clz(r), sh = s - 6, ii = r - (1u << s);
            pos = (1u << s) + ((ii & ((1u << sh) - 1u)) << 6) + (ii >> sh);
        }
        twf[pos] = c->is64 ? make_uint2(0, 0) : make_uint2((u32)p, (u32)(((u128)p << 32) / Q));
        p = mul_mod(p, c->psi, Q);
    }
    if (hipMalloc(&c->d_twf, sizeof(uint2) * N) != hipSuccess) {
        g_create_error = "hipMalloc(twiddles) failed";
        return BCE_ERR_HIP;
    }
    hipMemcpy(c->d_twf, twf.data(), sizeof(uint2) * N, hipMemcpyHostToDevice);
    if (hipMalloc(&c->d_psi, sizeof(u32) * N) != hipSuccess) { g_create_error = "hipMalloc(psi table) failed"; return BCE_ERR_HIP; }
    hipMemcpy(c->d_psi, psitab.data(), sizeof(u32) * N, hipMemcpyHostToDevice);
    const u64 r2modq = c->is64 ? 0 : (u64)((((u128)1) << 64) % Q);     // R^2 mod Q, R = 2^32
    {
        std::vector<u32> t(N);
        for (u32 i = 0; i < N; ++i) t[i] = c->is64 ? 0 : (u32)mul_mod(psitab[i], r2modq, Q);
        if (hipMalloc(&c->d_psi_r2, sizeof(u32) * N) != hipSuccess) { g_create_error = "hipMalloc(psi table) failed"; return BCE_ERR_HIP; }
        hipMemcpy(c->d_psi_r2, t.data(), sizeof(u32) * N, hipMemcpyHostToDevice);
    }
    for (int i = 0; i < bce_ctx::kRing; ++i) hipEventCreateWithFlags(&c->ring_ev[i], hipEventDisableTiming);

    DevParams& P = c->P;
    P.n = n; P.N = N; P.logN = c->logN; P.q = (u32)q; P.Q = (u32)Q; P.qKS = (u32)c->qKS;
    P.baseKS = baseKS; P.dKS = c->dKS;
    P.ksk_stride = (n + 1 + 63) & ~63u;
    P.ksk_u16 = c->qKS <= 65536 ? 1 : 0;
    {
        const u64 ch = (((u64)1) << 32) / c->qKS;
        if (ch < 8) { g_create_error = "qKS too large for the key-switch gather (needs qKS <= 2^29)"; return BCE_ERR_ARG; }
        P.ks_chunk = (u32)std::min<u64>(ch & ~(u64)7, 1u << 20);
    }
    P.gBits = c->gBits; P.dG = c->dG; P.baseR = baseR; P.dR = c->dR;
    P.method_ap = method == BCE_AP ? 1 : 0;
    P.factor = (u32)(2 * N / q);
    P.Q8p1 = (u32)(Q / 8 + 1);
    int bq = bit_length(Q);
    P.red_shift = (u32)std::max(2 * bq + 3 - 32, 0);
    P.red_mu = c->is64 ? 0 : (u32)((((u128)1) << (32 + P.red_shift)) / Q);
    u64 ninv = pow_mod(N, Q - 2, Q);
    P.Ninv = (u32)ninv;
    P.Ninv_s = c->is64 ? 0 : (u32)(((u128)ninv << 32) / Q);
    P.mu32 = (u32)((((u64)1) << 32) / Q);
    P.c32 = (u32)((((u64)1) << 32) % Q);
    {
        // lazy forward NTT: inputs < 2Q, every stage adds < 2Q -> outputs < B*Q with B = 2*logN+2.
        // needs (1) B*Q < 2^32, (2) R*B*Q*Q < 2^64 for the 64-bit MAC sums, (3) the folded sum
        // (x>>32)*c32 + 2^32 below the Barrett input bound 2^(32+red_shift)
        const u128 B = 2 * c->logN + 2, R = 2 * c->dG;
        const u128 sum = R * B * Q * Q;
        const bool ok1 = B * Q < ((u128)1 << 32);
        const bool ok2 = sum < ((u128)1 << 64);
        const u128 folded = (sum >> 32) * P.c32 + ((u128)1 << 32);
        const bool ok3 = folded < ((u128)1 << (32 + P.red_shift));
        // (4) MAC keeps rp, rn < 3Q and acc < 2Q: 3Q*Q + 3Q*Q + 2Q below the Barrett bound, 3Q < 2^32
        const bool ok4 = (u128)6 * Q * Q + 2 * Q < ((u128)1 << (32 + P.red_shift)) && (u128)3 * Q < ((u128)1 << 32);
        // (5) Montgomery form of the MAC tail (split-transform kernels): rp', rn' = REDC(sum) < sum / 2^32 + Q; the monomial
        // factors are lazy in [0, 2Q); y = rp' M+ + rn' M- must fit 64 bits and REDC(y) < y / 2^32 + Q must stay <= 2Q, so that
        // REDC(y) + acc (< 2Q) < 4Q < 2^32 and one conditional subtraction of 2Q returns it to [0, 2Q)
        const u128 rmax = (sum >> 32) + 1 + Q, ymax = 2 * rmax * 2 * Q;
        const bool ok5 = ymax < ((u128)1 << 64) && (ymax >> 32) + 1 + Q <= 2 * (u128)Q && (u128)4 * Q < ((u128)1 << 32);
        P.lazy = (ok1 && ok2 && ok3 && ok4 && ok5) ? 1 : 0;
    }
    if (!c->is64) {
        u32 inv = (u32)Q;                                  // Newton: inv = Q^-1 mod 2^32 (Q odd)
        for (int i = 0; i < 5; ++i) inv *= 2u - (u32)Q * inv;
        P.qinv_neg = 0u - inv;
        P.r2_off = (u32)(Q - r2modq);
    }
    {
        const char* occ = std::getenv("BCE_OCCUPANCY");  // development knob: 2 or 3 workgroups per CU
        P.occupancy_target = (occ && occ[0] == '2') ? 2 : 3;
        const size_t lds = (2 * (size_t)N + (2 + 2 * c->dG) * ((size_t)N + (N >> 6) * 4) + ((n + 1 + 3) & ~3u)) * 4;
        if (3 * lds > 160 * 1024) P.occupancy_target = 2;
        const char* var = std::getenv("BCE_VARIANT");  // development knob, see DevParams::variant
        P.variant = var ? (u32)std::atoi(var) : 0;
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
        P.cu_count = (u32)cus;
        const char* ft = std::getenv("BCE_FUSE_TAIL");  // development / parity knob: 0 keeps the separate tail kernels
        P.fuse_tail = (ft && ft[0] == '0') ? 0 : 1;
    }
    {
        u64 I = pow_mod(c->psi, N / 2, Q), v = 1;
        for (int k = 0; k < 4; ++k) {
            P.I4[k] = (u32)v;
            P.I4s[k] = c->is64 ? 0 : (u32)(((u128)v << 32) / Q);
            v = mul_mod(v, I, Q);
        }
        const u64 wl = mul_mod(P.I4[3], ninv, Q);  // -I * N^-1 (I^3 = -I)
        P.Winv_last = (u32)wl;
        P.Winv_last_s = c->is64 ? 0 : (u32)(((u128)wl << 32) / Q);
    }
    P.tw_f = c->d_twf;
    P.psi_tab = c->d_psi;
    P.psi_tab_r2 = c->d_psi_r2;
    P.xcd_gate = nullptr;
    P.xcd_gate_ticks = 10000;   // 100 us
    if (const char* e = std::getenv("BCE_XCD_GATE"); !c->is64 && !(e && e[0] == '0')) {   // on by default (32-bit path); BCE_XCD_GATE=0: A/B runs
        if (hipMalloc(&c->d_xcd_gate, 16 * 32 * sizeof(u32)) != hipSuccess) { g_create_error = "hipMalloc(xcd gate) failed"; return BCE_ERR_HIP; }
        P.xcd_gate = c->d_xcd_gate;
        if (const char* t = std::getenv("BCE_XCD_GATE_US")) P.xcd_gate_ticks = (u32)std::atoi(t) * 100u;
    }
    P.is64 = c->is64 ? 1 : 0;
    P.Q64 = Q;
    P.Q8p1_64 = Q / 8 + 1;
    P.mu64 = (u64)((((u128)1) << 64) / Q);
    P.c64 = (u64)((((u128)1) << 64) % Q);
    P.Ninv64 = ninv;
    P.Ninv64_s = (u64)(((u128)ninv << 64) / Q);
    if (c->is64) {
        std::vector<ulonglong2> tw64(N);
        u64 pw = 1;
        for (u32 i = 0; i < N; ++i) {
            tw64[bit_reverse(i, (int)c->logN)] = make_ulonglong2(pw, (u64)(((u128)pw << 64) / Q));
            pw = mul_mod(pw, c->psi, Q);
        }
        if (hipMalloc(&c->d_tw64, sizeof(ulonglong2) * N) != hipSuccess) { g_create_error = "hipMalloc(twiddles64) failed"; return BCE_ERR_HIP; }
        hipMemcpy(c->d_tw64, tw64.data(), sizeof(ulonglong2) * N, hipMemcpyHostToDevice);
        P.tw64 = c->d_tw64;
        P.lazy = 1;
        // double-precision formulation (kernels64.hip, namespace wd): exact for Q < 2^39; BCE_FP64=0 keeps the
        // integer kernel (development / parity knob)
        const char* fp = std::getenv("BCE_FP64");
        P.fp64 = (Q < (1ull << 39) && !(fp && fp[0] == '0')) ? 1 : 0;
        if (narrow64) P.fp64 = 0;   // no doubles kernel for this class: the key words stay 64-bit integers
        // includes the 16 KiB twiddle mirror of the 8-wave N = 2048 kernel (n <= ~1020 there)
        if (blind_rotate64_lds_bytes(P) > 160 * 1024) { g_create_error = "64-bit path: polynomials (+ twiddle mirror) do not fit the 160 KiB LDS"; return BCE_ERR_UNSUPPORTED; }
        P.Qd = (double)Q;
        P.invQd = 1.0 / (double)Q;
        P.Ninvd = (double)ninv;
        P.Ninvd_q = (double)ninv / (double)Q;
        std::vector<double2> twd(N);
        for (u32 i = 0; i < N; ++i) twd[i] = make_double2((double)tw64[i].x, (double)tw64[i].x / (double)Q);
        if (hipMalloc(&c->d_tw64d, sizeof(double2) * N) != hipSuccess) { g_create_error = "hipMalloc(twiddles64d) failed"; return BCE_ERR_HIP; }
        hipMemcpy(c->d_tw64d, twd.data(), sizeof(double2) * N, hipMemcpyHostToDevice);
        P.tw64d = c->d_tw64d;
    }
    // Lowest gadget digit folded into the key (kernels.hip, FOLD): needs a kernel that has the variant and an EXACT
    // SignedDigitDecompose -- every centred residue d in [-(Q - Q/2), Q/2) must equal sum_l r_l B^l with dG digits
    // r_l in [-B/2, B/2) (then the carry the reference drops after the last digit is always zero).  TOY (27-bit Q,
    // B = 2^9, 3 digits) fails it for the top 0.1 % of residues; STD128* (B = 2^7, 4 digits) and STD192* (37-bit Q,
    // B = 2^13, 3 digits) pass.  BCE_FOLD=0 keeps the plain key (development / parity knob).
    {
        const u128 Bg = (u128)1 << c->gBits;
        u128 span = 0, pw = 1;
        for (u32 l = 0; l < c->dG; ++l) { span += pw; pw *= Bg; }
        const u128 hi = (Bg / 2 - 1) * span, lo = (Bg / 2) * span;       // largest / smallest (negated) representable value
        const bool exact = (u128)(Q >> 1) <= hi + 1 && (u128)(Q - (Q >> 1)) <= lo;
        // (the folded fp64 kernels multiply a digit by a twiddle in one exact multiplication: |digit| Q <= (B / 2) Q < 2^53)
        const bool has_kernel = c->is64 ? (P.fp64 && c->logN == 11 && c->dG == 3 && (Bg / 2) * (u128)Q < ((u128)1 << 53))   // kernels64.hip, N = 2048
                                         : (c->logN == 10 && c->dG == 4 && P.lazy && P.variant != 1);  // kernels.hip, split transform
        const char* fo = std::getenv("BCE_FOLD");
        P.fold = (exact && has_kernel && !(fo && fo[0] == '0')) ? 1 : 0;
        P.fold_ninv = (BCE_KEY_NINV && P.fold && c->is64 && P.fp64) ? 1 : 0;
    }
    P.pool_stride = n + 1;
    c->enc_seed_ok = os_entropy(c->enc_seed);
    *out = c.release();
    return BCE_OK;
}

u64 rgsw_rows_total(const bce_ctx* c);
// evaluation-form key words -> what the kernels read: rows l >= 1 of every RGSW ciphertext minus B^l times row 0
// when the lowest gadget digit is folded (P.fold); IEEE doubles for the double-precision 64-bit kernels (exact, Q < 2^39)
int bsk_words_to_kernel_layout(bce_ctx* c) {
    if (c->P.fold) HIP_TRY(c, launch_fold_gadget(c->P, c->d_bsk, rgsw_rows_total(c) / (2ull * c->dG), +1, c->stream));
    if (!c->is64 || !c->P.fp64) { HIP_TRY(c, hipStreamSynchronize(c->stream)); return BCE_OK; }
    HIP_TRY(c, launch_words_u64_f64(static_cast<u64*>(c->d_bsk), (size_t)c->bsk_polys * c->N, 1, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BCE_OK;
}

u64 rgsw_rows_total(const bce_ctx* c) {
    const u64 R = 2ull * c->dG;
    return c->method == BCE_GINX ? (u64)c->n * 2 * R : (u64)c->n * c->baseR * c->dR * R;
}

int alloc_keys(bce_ctx* c) {
    if (!c->d_bsk) {
        c->bsk_polys = rgsw_rows_total(c) * 2;
        HIP_TRY(c, hipMalloc(&c->d_bsk, c->wbytes * c->bsk_polys * c->N));
    }
    if (!c->d_ksk) {
        size_t rows = (size_t)c->N * c->baseKS * c->dKS;
        HIP_TRY(c, hipMalloc(&c->d_ksk, rows * c->P.ksk_stride * (c->P.ksk_u16 ? 2 : 4)));
    }
    c->P.bsk = static_cast<const u32*>(c->d_bsk);
    c->P.bsk64 = static_cast<const u64*>(c->d_bsk);
    c->P.ksk = c->d_ksk;
    return BCE_OK;
}

// host KSK rows (u32, canonical [row][n+1]) -> padded device layout
int upload_ksk(bce_ctx* c, const u32* ksk) {
    const size_t rows = (size_t)c->N * c->baseKS * c->dKS, W = c->n + 1, S = c->P.ksk_stride;
    if (c->P.ksk_u16) {
        std::vector<uint16_t> buf(rows * S, 0);
        for (size_t r = 0; r < rows; ++r)
            for (size_t k = 0; k < W; ++k) buf[r * S + k] = (uint16_t)ksk[r * W + k];
        HIP_TRY(c, hipMemcpy(c->d_ksk, buf.data(), buf.size() * 2, hipMemcpyHostToDevice));
    } else {
        std::vector<u32> buf(rows * S, 0);
        for (size_t r = 0; r < rows; ++r) std::memcpy(&buf[r * S], &ksk[r * W], W * 4);
        HIP_TRY(c, hipMemcpy(c->d_ksk, buf.data(), buf.size() * 4, hipMemcpyHostToDevice));
    }
    return BCE_OK;
}

int ensure_acc(bce_ctx* c, size_t boots) {
    if (boots <= c->acc_cap) return BCE_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_acc) hipFree(c->d_acc);
    c->d_acc = nullptr;
    size_t cap = std::max(boots, c->acc_cap * 2);
    HIP_TRY(c, hipMalloc(&c->d_acc, cap * 2 * c->N * c->wbytes));
    c->acc_cap = cap;
    return BCE_OK;
}

// stage a descriptor list into the next ring slot; returns device pointer
int stage_descs(bce_ctx* c, const bce_gate_desc* d, size_t n, bce_gate_desc** dev, int* slot) {
    int k = c->ring_pos;
    c->ring_pos = (k + 1) % bce_ctx::kRing;
    if (c->ring_busy[k]) {
        HIP_TRY(c, hipEventSynchronize(c->ring_ev[k]));
        c->ring_busy[k] = false;
    }
    if (n > c->desc_cap[k]) {
        if (c->d_descs[k]) hipFree(c->d_descs[k]);
        if (c->h_descs[k]) hipHostFree(c->h_descs[k]);
        size_t cap = std::max(n, (size_t)1024);
        HIP_TRY(c, hipMalloc(&c->d_descs[k], cap * sizeof(bce_gate_desc)));
        HIP_TRY(c, hipHostMalloc(&c->h_descs[k], cap * sizeof(bce_gate_desc)));
        c->desc_cap[k] = cap;
    }
    std::memcpy(c->h_descs[k], d, n * sizeof(bce_gate_desc));
    HIP_TRY(c, hipMemcpyAsync(c->d_descs[k], c->h_descs[k], n * sizeof(bce_gate_desc), hipMemcpyHostToDevice, c->stream));
    *dev = c->d_descs[k];
    *slot = k;
    return BCE_OK;
}

EventPair get_events(bce_ctx* c, int kind) {
    EventPair p;
    if (!c->free_events.empty()) {
        p = c->free_events.back();
        c->free_events.pop_back();
    } else {
        hipEventCreate(&p.a);
        hipEventCreate(&p.b);
    }
    p.kind = kind;
    return p;
}

void drain_timing(bce_ctx* c) {
    for (auto& p : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            if (p.kind < BCE_BR_KERNELS) { c->timing.blind_rotate_ms += ms; c->timing.br_ms[p.kind] += ms; }
            else c->timing.tail_ms += ms;
        }
        c->free_events.push_back(p);
    }
    c->pending.clear();
}


// ---- word-size generic device helpers (u32: kernels.hip, u64: kernels64.hip) -----------------
int dev_ntt(bce_ctx* c, void* polys, u64 count, int inverse) {
    // count may exceed what one launch indexes comfortably; split in slabs of 2^20 polys
    const u64 slab = 1u << 20;
    for (u64 o = 0; o < count; o += slab) {
        const u32 cnt = (u32)std::min<u64>(slab, count - o);
        if (c->is64) HIP_TRY(c, launch_ntt_batch64(c->P, static_cast<u64*>(polys) + o * c->N, cnt, inverse, c->stream));
        else HIP_TRY(c, launch_ntt_batch(c->P, static_cast<u32*>(polys) + o * c->N, cnt, inverse, c->stream));
    }
    return BCE_OK;
}

// device copies of what the key samplers read (keygen.hip): secrets, Gaussian CDF table, parameters
struct KeygenDev {
    int32_t *s = nullptr, *z = nullptr;
    u64* cdf = nullptr;
    void *ta = nullptr, *zq = nullptr;
    ~KeygenDev() { hipFree(s); hipFree(z); hipFree(cdf); hipFree(ta); hipFree(zq); }
};

int keygen_params(bce_ctx* c, const GaussSampler& gauss, KeygenDev& D, KeygenParams& kp) {
    HIP_TRY(c, hipMalloc(&D.s, c->n * sizeof(int32_t)));
    HIP_TRY(c, hipMalloc(&D.z, c->N * sizeof(int32_t)));
    HIP_TRY(c, hipMalloc(&D.cdf, 81 * sizeof(u64)));
    HIP_TRY(c, hipMemcpy(D.s, c->s.data(), c->n * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(D.z, c->z.data(), c->N * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(D.cdf, gauss.table(), 81 * sizeof(u64), hipMemcpyHostToDevice));
    std::memcpy(kp.seed, c->seed, 32);
    kp.n = c->n; kp.N = c->N; kp.Q = c->Q; kp.q = c->q; kp.qKS = c->qKS;
    kp.qbits = bit_length(c->Q - 1); kp.ksbits = bit_length(c->qKS - 1);
    kp.R = 2 * c->dG; kp.ap = c->method == BCE_AP ? 1 : 0; kp.baseR = c->baseR; kp.dR = c->dR;
    kp.baseKS = c->baseKS; kp.dKS = c->dKS; kp.ksk_stride = c->P.ksk_stride;
    { u64 v = 1; for (u32 i = 0; i < 4; ++i) { kp.gpow[i] = v; v = mul_mod(v, c->baseG, c->Q); } }
    kp.s = D.s; kp.z = D.z; kp.cdf = D.cdf;
    return BCE_OK;
}

// Bootstrapping key.  GINX: ek[i][0] = RGSW(s_i == 1), ek[i][1] = RGSW(s_i == -1)
// (rgsw-acc-cggi.cpp KeyGenAcc).  AP: ek[i][v][k] = RGSW(X^{s_i * v * baseR^k * 2N/q}), v >= 1
// (rgsw-acc-dm.cpp KeyGenAcc; the v = 0 slots stay zero and are never read).
// Everything happens on the device: rows (a, e + gadget) are sampled in place (keygen.hip), transformed, and
// b += NTT(a) * NTT(z).  Rows are produced in chunks whose scratch (the masks once more) stays below 1 GiB
// whatever the key size (STD192/AP: 12.9 GB of key).
int keygen_bsk(bce_ctx* c, const KeygenParams& kp, KeygenDev& D) {
    const u32 N = c->N, R = 2 * c->dG;
    const u64 Q = c->Q;
    const size_t wb = c->wbytes;
    const u64 rows = rgsw_rows_total(c);
    const u64 chunk = std::max<u64>(R, std::min<u64>(rows, ((u64)1 << 30) / ((u64)N * wb)) / R * R);
    HIP_TRY(c, hipMalloc(&D.ta, chunk * N * wb));
    HIP_TRY(c, hipMalloc(&D.zq, N * wb));
    {
        std::vector<u64> zq(N);
        for (u32 k = 0; k < N; ++k) zq[k] = lift_signed(c->z[k], Q);
        if (c->is64) {
            HIP_TRY(c, hipMemcpy(D.zq, zq.data(), N * 8, hipMemcpyHostToDevice));
        } else {
            std::vector<u32> z32(zq.begin(), zq.end());
            HIP_TRY(c, hipMemcpy(D.zq, z32.data(), N * 4, hipMemcpyHostToDevice));
        }
    }
    int rc = dev_ntt(c, D.zq, 1, 0);
    if (rc) return rc;
    char* dev = static_cast<char*>(c->d_bsk);
    for (u64 r0 = 0; r0 < rows; r0 += chunk) {
        const u64 cnt = std::min(chunk, rows - r0);
        char* dst = dev + r0 * 2 * N * wb;
        HIP_TRY(c, launch_gen_bsk_rows(kp, r0, (u32)cnt, dst, D.ta, c->is64 ? 1 : 0, c->stream));
        if ((rc = dev_ntt(c, dst, cnt * 2, 0))) return rc;
        if ((rc = dev_ntt(c, D.ta, cnt, 0))) return rc;
        // b-column (odd polys) += NTT(a) * NTT(z)
        if (c->is64) HIP_TRY(c, launch_pointwise_mac64(c->P, (u64*)dst + N, (const u64*)D.ta, (const u64*)D.zq, (u32)cnt, 2, c->stream));
        else HIP_TRY(c, launch_pointwise_mac(c->P, (u32*)dst + N, (const u32*)D.ta, (const u32*)D.zq, (u32)cnt, 2, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BCE_OK;
}

// after a stream synchronisation: verdict of the dependency-driven runs that finished since the last check
int check_dag_runs(bce_ctx* c) {
    int rc = BCE_OK;
    for (int i = 0; i < c->dag_pending; ++i) {
        const bce_ctx::DagStatus& st = c->h_dag_status[i];
        c->dag_last[0] = st.done; c->dag_last[1] = st.lazy_waits; c->dag_last[2] = st.abort; c->dag_last[3] = (u64)c->dag_wps_used[i] / 2;
        c->dag_last[4] = st.busy_ticks; c->dag_last[5] = st.wait_ticks; c->dag_last[6] = st.gate_ticks;
        if (st.abort != 0 || st.done != c->dag_expected[i])
            rc = c->fail(BCE_ERR_STATE, "bce_dag_run: the device scheduler gave up (abort code %u, %u of %llu bootstraps completed, no progress for %u ms)",
                         st.abort, st.done, (unsigned long long)c->dag_expected[i], c->dag_stall_ms);
    }
    c->dag_pending = 0;
    return rc;
}

int sync_stream(bce_ctx* c) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return check_dag_runs(c);
}

// One frontier of bootstrapped gates whose descriptors already sit on the device: blind rotation (+ the tail kernels
// when the blind-rotation kernel does not carry it).  timed: HIP events around each kernel group + the counters of
// bce_timing; untimed (stream capture of bce_plan_run): launches only, *fused_out says whether the tail was fused.
int launch_bootstraps(bce_ctx* c, const bce_gate_desc* dd, u32 n, u32 instances, u32 slot_stride, void* d_acc,
                      u32* d_lweN, u32* d_ks, bool timed, bool* fused_out, u64* d_partial = nullptr) {
    const size_t nb = (size_t)n * instances;
    int kid = BCE_BR_WORD64;
    bool tail_fused = false;
    const bool events = timed && c->events_on;
    EventPair e0{};
    // the timestamps ride on the kernels' own dispatches (LaunchEvents): no event packet between dependent launches
    LaunchEvents le0{};
    if (events) { e0 = get_events(c, 0); le0 = LaunchEvents{e0.a, e0.b}; }
    {   // a launch that fails hands its event pair back (it would otherwise be neither pending nor free)
        const hipError_t e = c->is64 ? launch_blind_rotate64(c->P, dd, n, instances, slot_stride, static_cast<u64*>(d_acc), c->stream, d_lweN, d_ks, &tail_fused, le0)
                                     : launch_blind_rotate(c->P, dd, n, instances, slot_stride, static_cast<u32*>(d_acc), c->stream, &kid, d_lweN, d_ks, &tail_fused, le0);
        if (e != hipSuccess) {
            if (events) c->free_events.push_back(e0);
            HIP_TRY(c, e);
        }
    }
    if (events) {
        e0.kind = kid;
        c->pending.push_back(e0);
    }
    if (timed) {
        c->timing.br_launches[kid] += 1;
        c->timing.br_bootstraps[kid] += nb;
    }
    if (!tail_fused) {
        EventPair e1{};
        LaunchEvents le1{};
        if (events) {
            e1 = get_events(c, BCE_BR_KERNELS);  // kind >= BCE_BR_KERNELS: tail
            le1 = LaunchEvents{e1.a, e1.b};
        }
        if (!d_partial) {
            const size_t need = tail_partial_words(c->P, (u32)nb);
            if (need > c->tail_cap) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                if (c->d_tail_partial) hipFree(c->d_tail_partial);
                c->d_tail_partial = nullptr;
                const size_t cap = std::max(need, c->tail_cap * 2);
                HIP_TRY(c, hipMalloc(&c->d_tail_partial, cap * sizeof(u64)));
                c->tail_cap = cap;
            }
        }
        const hipError_t e = launch_tail(c->P, dd, n, instances, slot_stride, d_acc, d_partial ? d_partial : c->d_tail_partial, d_lweN, d_ks, c->stream, le1);
        if (e != hipSuccess) {
            if (events) c->free_events.push_back(e1);
            HIP_TRY(c, e);
        }
        if (events) c->pending.push_back(e1);
    } else if (timed) {
        c->timing.fused_tail_launches += 1;
    }
    if (timed) {
        c->timing.blind_rotate_launches += 1;
        c->timing.bootstraps += nb;
    }
    if (fused_out) *fused_out = tail_fused;
    return BCE_OK;
}

int eval_impl(bce_ctx* c, u32 n_desc, const bce_gate_desc* descs, u32 instances, u32 slot_stride, u64* dbg_acc,
              u64* dbg_lweN, u64* dbg_ks) {
    if (n_desc == 0 || instances == 0) return BCE_OK;
    if (!descs) return c->fail(BCE_ERR_ARG, "null descriptor list");
    if (!c->have_keys) return c->fail(BCE_ERR_NO_KEYS, "bce_keygen / bce_import_keys has not been called");
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<bce_gate_desc> boot, unary;
    boot.reserve(n_desc);
    const u64 max_slot = (u64)(instances - 1) * slot_stride;
    for (u32 i = 0; i < n_desc; ++i) {
        const bce_gate_desc& g = descs[i];
        const bool is_boot = g.op <= BCE_XNOR_FAST || g.op == BCE_OP_REFRESH;
        const bool is_unary = g.op == BCE_OP_NOT || g.op == BCE_OP_COPY;
        if (!is_boot && !is_unary) return c->fail(BCE_ERR_ARG, "descriptor %u: unknown op %u", i, g.op);
        u64 hi = std::max<u64>(g.in0, g.out);
        if (g.op <= BCE_XNOR_FAST) hi = std::max<u64>(hi, g.in1);
        if (hi + max_slot >= c->pool_slots) return c->fail(BCE_ERR_POOL, "descriptor %u: slot %llu outside the pool (%u slots)", i, (unsigned long long)(hi + max_slot), c->pool_slots);
        (is_boot ? boot : unary).push_back(g);
    }
    if (!boot.empty()) {
        const size_t nb = boot.size() * (size_t)instances;
        int rc = ensure_acc(c, nb);
        if (rc) return rc;
        bce_gate_desc* dd = nullptr;
        int slot = 0;
        rc = stage_descs(c, boot.data(), boot.size(), &dd, &slot);
        if (rc) return rc;
        u32 *d_lweN = nullptr, *d_ks = nullptr;
        if (dbg_lweN) HIP_TRY(c, hipMalloc(&d_lweN, nb * (c->N + 1) * sizeof(u32)));
        if (dbg_ks) HIP_TRY(c, hipMalloc(&d_ks, nb * (c->n + 1) * sizeof(u32)));
        rc = launch_bootstraps(c, dd, (u32)boot.size(), instances, slot_stride, c->d_acc, d_lweN, d_ks, true, nullptr);
        if (rc) return rc;
        hipEventRecord(c->ring_ev[slot], c->stream);
        c->ring_busy[slot] = true;
        if (dbg_acc || dbg_lweN || dbg_ks) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            std::vector<u32> tmp;
            auto fetch = [&](const u32* dev, size_t words, u64* dst) -> int {
                tmp.resize(words);
                HIP_TRY(c, hipMemcpy(tmp.data(), dev, words * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < words; ++i) dst[i] = tmp[i];
                return BCE_OK;
            };
            if (dbg_acc && c->is64) HIP_TRY(c, hipMemcpy(dbg_acc, c->d_acc, nb * 2 * c->N * 8, hipMemcpyDeviceToHost));
            if (dbg_acc && !c->is64 && (rc = fetch(static_cast<const u32*>(c->d_acc), nb * 2 * c->N, dbg_acc))) return rc;
            if (dbg_lweN && (rc = fetch(d_lweN, nb * (c->N + 1), dbg_lweN))) return rc;
            if (dbg_ks && (rc = fetch(d_ks, nb * (c->n + 1), dbg_ks))) return rc;
            if (d_lweN) hipFree(d_lweN);
            if (d_ks) hipFree(d_ks);
        }
        if (c->pending.size() > 4096) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            drain_timing(c);
        }
    }
    if (!unary.empty()) {
        bce_gate_desc* dd = nullptr;
        int slot = 0;
        int rc = stage_descs(c, unary.data(), unary.size(), &dd, &slot);
        if (rc) return rc;
        HIP_TRY(c, launch_lwe_unary(c->P, dd, (u32)unary.size(), instances, slot_stride, c->stream));
        hipEventRecord(c->ring_ev[slot], c->stream);
        c->ring_busy[slot] = true;
    }
    return BCE_OK;
}

}  // namespace

extern "C" {

int bce_ctx_create(int paramset, int method, int device, bce_ctx** out) {
    if (paramset < 0 || paramset > BCE_STD256_OPT) { g_create_error = "unknown parameter set"; if (out) *out = nullptr; return BCE_ERR_ARG; }
    const ParamRow& r = kParamTable[paramset];
    const u64 Q = previous_prime(first_prime(r.bits, r.cyclo), r.cyclo);
    return build_ctx(r.n, r.cyclo / 2, r.q, Q, r.qKS, r.baseKS, r.baseG, r.baseR, method, device, out);
}

int bce_ctx_create_custom(uint32_t n, uint32_t N, uint64_t q, uint64_t Q, uint64_t qKS, uint32_t baseKS, uint32_t baseG,
                          uint32_t baseR, int method, int device, bce_ctx** out) {
    return build_ctx(n, N, q, Q, qKS, baseKS, baseG, baseR, method, device, out);
}

static void plan_free(bce_plan* p);
static void dag_free(bce_dag* g);

void bce_ctx_destroy(bce_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->rccl_comm) bce_rccl_shutdown(c);
    drain_timing(c);
    {   // schedules nobody destroyed: theirs callers' handles die with the context
        std::unordered_set<bce_plan*> plans;
        std::unordered_set<bce_dag*> dags;
        { std::lock_guard<std::mutex> lk(g_live_mu); plans.swap(c->plans); dags.swap(c->dags); g_live.erase(c); }
        for (bce_plan* p : plans) plan_free(p);
        for (bce_dag* g : dags) dag_free(g);
    }
    for (auto& p : c->free_events) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (int i = 0; i < bce_ctx::kRing; ++i) {
        if (c->d_descs[i]) hipFree(c->d_descs[i]);
        if (c->h_descs[i]) hipHostFree(c->h_descs[i]);
        if (c->ring_ev[i]) hipEventDestroy(c->ring_ev[i]);
    }
    hipFree(c->d_io); if (c->h_io) hipHostFree(c->h_io);
    hipFree(c->d_P); if (c->h_dag_status) hipHostFree(c->h_dag_status); if (c->h_dag_stage) hipHostFree(c->h_dag_stage);
    hipFree(c->d_twf); hipFree(c->d_psi); hipFree(c->d_psi_r2); hipFree(c->d_xcd_gate); hipFree(c->d_tw64); hipFree(c->d_tw64d); hipFree(c->d_bsk); hipFree(c->d_ksk); hipFree(c->d_pool); hipFree(c->d_acc); hipFree(c->d_tail_partial);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

hipStream_t bce_internal_stream(bce_ctx* c) { return c->stream; }
void** bce_internal_comm_slot(bce_ctx* c) { return &c->rccl_comm; }
int bce_internal_device(bce_ctx* c) { return c->device; }
int bce_rccl_shutdown(bce_ctx* c);

int bce_set_error(bce_ctx* c, int code, const char* msg) {  // for the other translation units of the library (keyfile.cpp)
    if (c) c->err = msg ? msg : "";
    return code;
}

const char* bce_last_error(const bce_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int bce_get_params(const bce_ctx* c, uint64_t out[BCE_P_COUNT]) {
    if (!c || !out) return BCE_ERR_ARG;
    out[BCE_P_n] = c->n; out[BCE_P_N] = c->N; out[BCE_P_q] = c->q; out[BCE_P_Q] = c->Q; out[BCE_P_qKS] = c->qKS;
    out[BCE_P_baseKS] = c->baseKS; out[BCE_P_dKS] = c->dKS; out[BCE_P_baseG] = c->baseG; out[BCE_P_dG] = c->dG;
    out[BCE_P_baseR] = c->baseR; out[BCE_P_dR] = c->dR; out[BCE_P_method] = (u64)c->method; out[BCE_P_psi] = c->psi;
    return BCE_OK;
}

uint64_t bce_bsk_words(const bce_ctx* c) { return c ? rgsw_rows_total(c) * 2 * c->N : 0; }
uint64_t bce_ksk_words(const bce_ctx* c) { return c ? (u64)c->N * c->baseKS * c->dKS * (c->n + 1) : 0; }

int bce_keygen(bce_ctx* c, const uint8_t seed_in[32]) {
    if (!c) return BCE_ERR_ARG;
    uint8_t fresh[32];
    const uint8_t* seed = seed_in;
    if (!seed) {  // cc.KeyGen() of the reference: keys from the system's entropy, never from a constant
        if (!os_entropy(fresh)) return c->fail(BCE_ERR_STATE, "bce_keygen: no entropy source (getrandom and /dev/urandom failed)");
        seed = fresh;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = alloc_keys(c);
    if (rc) return rc;
    std::memcpy(c->seed, seed, 32);
    const u32 n = c->n, N = c->N;
    const GaussSampler gauss(3.19);
    c->s.resize(n);
    c->z.resize(N);
    { ChaChaStream st(seed, kDomSK, 0); for (u32 i = 0; i < n; ++i) c->s[i] = draw_ternary(st); }
    { ChaChaStream st(seed, kDomZ, 0); for (u32 i = 0; i < N; ++i) c->z[i] = draw_ternary(st); }

    KeygenDev D;
    KeygenParams kp{};
    if ((rc = keygen_params(c, gauss, D, kp))) return rc;
    // LWE key-switching key: K[i][v][j] = LWE_s(z_i * v * baseKS^j) mod qKS, sampled on the device straight into
    // the padded row layout the tail kernel gathers from
    {
        const u64 rows = (u64)N * c->baseKS * c->dKS;
        HIP_TRY(c, hipMemsetAsync(c->d_ksk, 0, rows * c->P.ksk_stride * (c->P.ksk_u16 ? 2 : 4), c->stream));
        HIP_TRY(c, launch_gen_ksk_rows(kp, rows, c->d_ksk, c->P.ksk_u16 ? 1 : 0, c->stream));
    }
    rc = keygen_bsk(c, kp, D);
    if (!rc) rc = bsk_words_to_kernel_layout(c);
    if (rc) return rc;
    c->have_keys = true;
    return BCE_OK;
}

static int import_keys_impl(bce_ctx* c, const int32_t* s, const int32_t* z, const uint64_t* bsk, uint64_t bsk_words,
                            const uint32_t* ksk, uint64_t ksk_words, bool evaluation_form);

int bce_import_keys(bce_ctx* c, const int32_t* s, const int32_t* z, const uint64_t* bsk, uint64_t bsk_words,
                    const uint32_t* ksk, uint64_t ksk_words) {
    return import_keys_impl(c, s, z, bsk, bsk_words, ksk, ksk_words, false);
}

int bce_import_keys_eval(bce_ctx* c, const int32_t* s, const int32_t* z, const uint64_t* bsk_eval, uint64_t bsk_words,
                         const uint32_t* ksk, uint64_t ksk_words) {
    return import_keys_impl(c, s, z, bsk_eval, bsk_words, ksk, ksk_words, true);
}

static int import_keys_impl(bce_ctx* c, const int32_t* s, const int32_t* z, const uint64_t* bsk, uint64_t bsk_words,
                            const uint32_t* ksk, uint64_t ksk_words, bool evaluation_form) {
    if (!c || !s || !bsk || !ksk) return c ? c->fail(BCE_ERR_ARG, "null key pointer") : BCE_ERR_ARG;
    if (bsk_words != bce_bsk_words(c) || ksk_words != bce_ksk_words(c)) return c->fail(BCE_ERR_ARG, "key sizes do not match the parameter set");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = alloc_keys(c);
    if (rc) return rc;
    c->s.assign(s, s + c->n);
    if (z) c->z.assign(z, z + c->N); else c->z.clear();
    std::memset(c->seed, 0, sizeof c->seed);   // imported keys have no generation seed
    {   // words -> device words, chunked; coefficient-domain words are then transformed in place on the device
        // (evaluation-form words arrive in the engine's own order: OpenFHE's bit-reversed CT order for the minimal
        // primitive 2N-th root, bce_get_params()[BCE_P_psi])
        const u64 chunk = (u64)64 << 20;
        std::vector<u32> tmp32;
        for (u64 o = 0; o < bsk_words; o += chunk) {
            const u64 cnt = std::min(chunk, bsk_words - o);
            for (u64 i = 0; i < cnt; ++i)
                if (bsk[o + i] >= c->Q) return c->fail(BCE_ERR_ARG, "bsk word %llu not reduced mod Q", (unsigned long long)(o + i));
            if (c->is64) {
                HIP_TRY(c, hipMemcpy(static_cast<u64*>(c->d_bsk) + o, bsk + o, cnt * 8, hipMemcpyHostToDevice));
            } else {
                tmp32.resize(cnt);
                for (u64 i = 0; i < cnt; ++i) tmp32[i] = (u32)bsk[o + i];
                HIP_TRY(c, hipMemcpy(static_cast<u32*>(c->d_bsk) + o, tmp32.data(), cnt * 4, hipMemcpyHostToDevice));
            }
        }
        if (!evaluation_form && (rc = dev_ntt(c, c->d_bsk, bsk_words / c->N, 0))) return rc;
        if ((rc = bsk_words_to_kernel_layout(c))) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    rc = upload_ksk(c, ksk);
    if (rc) return rc;
    c->have_keys = true;
    return BCE_OK;
}

int bce_export_sk(const bce_ctx* c, int32_t* s, int32_t* z) {
    if (!c || !c->have_keys) return BCE_ERR_NO_KEYS;
    if (s) std::memcpy(s, c->s.data(), c->n * sizeof(int32_t));
    if (z && !c->z.empty()) std::memcpy(z, c->z.data(), c->N * sizeof(int32_t));
    return BCE_OK;
}

static int export_bsk_impl(bce_ctx* c, uint64_t* bsk, bool evaluation_form);
int bce_export_bsk(bce_ctx* c, uint64_t* bsk) { return export_bsk_impl(c, bsk, false); }

static int export_bsk_impl(bce_ctx* c, uint64_t* bsk, bool evaluation_form) {
    if (!c || !bsk) return BCE_ERR_ARG;
    if (!c->have_keys) return c->fail(BCE_ERR_NO_KEYS, "no keys");
    HIP_TRY(c, hipSetDevice(c->device));
    const u64 words = bce_bsk_words(c);
    const u64 per_rgsw = 4ull * c->dG;  // polynomials of one RGSW ciphertext: chunks hold whole ciphertexts (un-folding works per ciphertext)
    const u64 chunk_polys = std::max<u64>(1, ((u64)256 << 20) / ((u64)c->N * c->wbytes) / per_rgsw) * per_rgsw;
    void* d_tmp = nullptr;
    HIP_TRY(c, hipMalloc(&d_tmp, chunk_polys * c->N * c->wbytes));
    struct Free { void* p; ~Free() { hipFree(p); } } free_tmp{d_tmp};  // released on every return path
    std::vector<u32> tmp32;
    for (u64 p0 = 0; p0 < words / c->N; p0 += chunk_polys) {
        const u64 cnt = std::min(chunk_polys, words / c->N - p0), w = cnt * c->N;
        const char* src = static_cast<const char*>(c->d_bsk) + p0 * c->N * c->wbytes;
        HIP_TRY(c, hipMemcpyAsync(d_tmp, src, w * c->wbytes, hipMemcpyDeviceToDevice, c->stream));
        if (c->P.fp64) HIP_TRY(c, launch_words_u64_f64(static_cast<u64*>(d_tmp), w, 0, c->stream));  // doubles -> u64 words
        if (c->P.fold) HIP_TRY(c, launch_fold_gadget(c->P, d_tmp, cnt / (4ull * c->dG), -1, c->stream));   // rows l >= 1 += B^l row 0
        if (!evaluation_form) { int rc = dev_ntt(c, d_tmp, cnt, 1); if (rc) return rc; }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->is64) {
            HIP_TRY(c, hipMemcpy(bsk + p0 * c->N, d_tmp, w * 8, hipMemcpyDeviceToHost));
        } else {
            tmp32.resize(w);
            HIP_TRY(c, hipMemcpy(tmp32.data(), d_tmp, w * 4, hipMemcpyDeviceToHost));
            for (u64 i = 0; i < w; ++i) bsk[p0 * c->N + i] = tmp32[i];
        }
    }
    return BCE_OK;
}

int bce_export_bsk_eval(bce_ctx* c, uint64_t* bsk) { return export_bsk_impl(c, bsk, true); }

int bce_export_ksk(bce_ctx* c, uint32_t* ksk) {
    if (!c || !ksk) return BCE_ERR_ARG;
    if (!c->have_keys) return c->fail(BCE_ERR_NO_KEYS, "no keys");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t rows = (size_t)c->N * c->baseKS * c->dKS, W = c->n + 1, S = c->P.ksk_stride;
    if (c->P.ksk_u16) {
        std::vector<uint16_t> buf(rows * S);
        HIP_TRY(c, hipMemcpy(buf.data(), c->d_ksk, buf.size() * 2, hipMemcpyDeviceToHost));
        for (size_t r = 0; r < rows; ++r)
            for (size_t k = 0; k < W; ++k) ksk[r * W + k] = buf[r * S + k];
    } else {
        std::vector<u32> buf(rows * S);
        HIP_TRY(c, hipMemcpy(buf.data(), c->d_ksk, buf.size() * 4, hipMemcpyDeviceToHost));
        for (size_t r = 0; r < rows; ++r) std::memcpy(&ksk[r * W], &buf[r * S], W * 4);
    }
    return BCE_OK;
}

int bce_pool_reserve(bce_ctx* c, uint32_t slots) {
    if (!c) return BCE_ERR_ARG;
    if (slots <= c->pool_slots) return BCE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    u32* np = nullptr;
    HIP_TRY(c, hipMalloc(&np, (size_t)slots * c->P.pool_stride * 4));
    HIP_TRY(c, hipMemset(np, 0, (size_t)slots * c->P.pool_stride * 4));
    if (c->d_pool) {
        HIP_TRY(c, hipMemcpy(np, c->d_pool, (size_t)c->pool_slots * c->P.pool_stride * 4, hipMemcpyDeviceToDevice));
        hipFree(c->d_pool);
    }
    c->d_pool = np;
    c->pool_slots = slots;
    c->P.pool = np;
    return BCE_OK;
}

uint32_t bce_pool_slots(const bce_ctx* c) { return c ? c->pool_slots : 0; }

static int pool_pack(bce_ctx* c, const uint32_t* slots, uint32_t count, void* dev, int to_pool);

int bce_lwe_write(bce_ctx* c, const uint32_t* slots, uint32_t count, const uint64_t* cts) {
    if (!c || !slots || !cts) return BCE_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t W = c->n + 1;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<u32> buf;
    for (u32 i = 0; i < count;) {
        u32 run = 1;  // coalesce runs of consecutive slots into one copy
        while (i + run < count && slots[i + run] == slots[i] + run) ++run;
        if ((u64)slots[i] + run > c->pool_slots) return c->fail(BCE_ERR_POOL, "slot %u outside the pool", slots[i] + run - 1);
        buf.resize((size_t)run * W);
        for (size_t k = 0; k < (size_t)run * W; ++k) {
            const u64 v = cts[(size_t)i * W + k];
            if (v >= c->q) return c->fail(BCE_ERR_ARG, "ciphertext word not reduced mod q");
            buf[k] = (u32)v;
        }
        HIP_TRY(c, hipMemcpy(c->d_pool + (size_t)slots[i] * c->P.pool_stride, buf.data(), buf.size() * 4, hipMemcpyHostToDevice));
        i += run;
    }
    return BCE_OK;
}

int bce_lwe_read(bce_ctx* c, const uint32_t* slots, uint32_t count, uint64_t* cts) {
    if (!c || !slots || !cts) return BCE_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t W = c->n + 1;
    { const int rc = sync_stream(c); if (rc) return rc; }
    // one bulk copy when the request is dense enough, else per slot
    u32 lo = ~0u, hi = 0;
    for (u32 i = 0; i < count; ++i) {
        if (slots[i] >= c->pool_slots) return c->fail(BCE_ERR_POOL, "slot %u outside the pool", slots[i]);
        lo = std::min(lo, slots[i]);
        hi = std::max(hi, slots[i]);
    }
    if (count == 0) return BCE_OK;
    std::vector<u32> buf;
    if ((u64)(hi - lo + 1) <= 4ull * count + 64) {
        buf.resize((size_t)(hi - lo + 1) * W);
        HIP_TRY(c, hipMemcpy(buf.data(), c->d_pool + (size_t)lo * W, buf.size() * 4, hipMemcpyDeviceToHost));
        for (u32 i = 0; i < count; ++i)
            for (size_t k = 0; k < W; ++k) cts[i * W + k] = buf[(size_t)(slots[i] - lo) * W + k];
    } else {
        // scattered slots (the outputs of K instances sit one pool stride apart): gather on the device, one copy
        const size_t words = (size_t)count * W;
        if (words > c->io_cap) {
            if (c->d_io) hipFree(c->d_io);
            if (c->h_io) hipHostFree(c->h_io);
            c->d_io = nullptr; c->h_io = nullptr; c->io_cap = 0;
            const size_t cap = std::max(words, (size_t)64 * W);
            HIP_TRY(c, hipMalloc(&c->d_io, cap * sizeof(u32)));
            HIP_TRY(c, hipHostMalloc(&c->h_io, cap * sizeof(u32)));
            c->io_cap = cap;
        }
        { const int rc = pool_pack(c, slots, count, c->d_io, 0); if (rc) return rc; }
        HIP_TRY(c, hipMemcpyAsync(c->h_io, c->d_io, words * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (size_t k = 0; k < words; ++k) cts[k] = c->h_io[k];
    }
    return BCE_OK;
}

int bce_encrypt_bits(bce_ctx* c, const uint8_t* bits, const uint32_t* slots, uint32_t count, uint64_t enc_index_base,
                     int mode) {
    if (!c || !bits || !slots) return BCE_ERR_ARG;
    if (!c->have_keys) return c->fail(BCE_ERR_NO_KEYS, "no keys");
    const u32 n = c->n;
    const u64 q = c->q;
    const size_t W = n + 1;
    const GaussSampler gauss(3.19);
    std::vector<u64> cts((size_t)count * W);
    if (!c->enc_deterministic) {
        // default: the caller's index is ignored; the context's own counter names the stream and never repeats
        if (!c->enc_seed_ok) return c->fail(BCE_ERR_STATE, "no entropy source for encryption randomness (getrandom and /dev/urandom failed)");
        enc_index_base = c->enc_counter;
        c->enc_counter += count;
    }
    for (u32 i = 0; i < count; ++i) {
        ChaChaStream st(c->enc_seed, kDomENC, enc_index_base + i);
        u64* ct = &cts[(size_t)i * W];
        u128 acc = 0;
        for (u32 k = 0; k < n; ++k) {
            ct[k] = draw_uniform(st, q);
            acc += (u128)ct[k] * lift_signed(c->s[k], q);
        }
        u64 e = lift_signed(gauss.draw(st), q);
        ct[n] = (u64)((acc + e + (u64)(bits[i] % 4) * (q / 4)) % q);
    }
    int rc = bce_lwe_write(c, slots, count, cts.data());
    if (rc) return rc;
    if (mode == BCE_BOOTSTRAPPED) {
        std::vector<bce_gate_desc> d(count);
        for (u32 i = 0; i < count; ++i) d[i] = bce_gate_desc{BCE_OP_REFRESH, slots[i], slots[i], slots[i], 0, 0};
        return bce_eval_gates(c, count, d.data());
    }
    return BCE_OK;
}

int bce_set_encrypt_seed(bce_ctx* c, const uint8_t seed[32]) {
    if (!c) return BCE_ERR_ARG;
    if (seed) {
        std::memcpy(c->enc_seed, seed, 32);
        c->enc_seed_ok = true;
        c->enc_deterministic = true;
    } else {
        c->enc_deterministic = false;
        c->enc_counter = 0;
        c->enc_seed_ok = os_entropy(c->enc_seed);
        if (!c->enc_seed_ok) return c->fail(BCE_ERR_STATE, "no entropy source (getrandom and /dev/urandom failed)");
    }
    return BCE_OK;
}

int bce_decrypt_bits(bce_ctx* c, const uint32_t* slots, uint32_t count, uint8_t* bits) {
    if (!c || !slots || !bits) return BCE_ERR_ARG;
    if (!c->have_keys) return c->fail(BCE_ERR_NO_KEYS, "no keys");
    const u32 n = c->n;
    const u64 q = c->q;
    const size_t W = n + 1;
    std::vector<u64> cts((size_t)count * W);
    int rc = bce_lwe_read(c, slots, count, cts.data());
    if (rc) return rc;
    for (u32 i = 0; i < count; ++i) {
        const u64* ct = &cts[(size_t)i * W];
        // <a, s> over the integers with s in {-1, 0, 1}: |sum| < n q, reduced once (same residue as the reference's
        // mod-q accumulation, lwe-pke.cpp Decrypt)
        u64 inner_q;
        if (q <= (1ull << 24)) {
            int64_t inner = 0;
            for (u32 k = 0; k < n; ++k) inner += (int64_t)ct[k] * c->s[k];
            inner_q = (u64)(((inner % (int64_t)q) + (int64_t)q) % (int64_t)q);
        } else {
            u128 inner = 0;
            for (u32 k = 0; k < n; ++k) inner += (u128)ct[k] * lift_signed(c->s[k], q);
            inner_q = (u64)(inner % q);
        }
        u64 r = (ct[n] + q - inner_q) % q;
        r = (r + q / 8) % q;
        bits[i] = (uint8_t)((4 * r) / q);
    }
    return BCE_OK;
}

int bce_eval_gates(bce_ctx* c, uint32_t n_desc, const bce_gate_desc* descs) {
    if (!c) return BCE_ERR_ARG;
    return eval_impl(c, n_desc, descs, 1, 0, nullptr, nullptr, nullptr);
}

int bce_eval_gates_strided(bce_ctx* c, uint32_t n_desc, const bce_gate_desc* descs, uint32_t instances,
                           uint32_t slot_stride) {
    if (!c) return BCE_ERR_ARG;
    return eval_impl(c, n_desc, descs, instances, slot_stride, nullptr, nullptr, nullptr);
}

int bce_synchronize(bce_ctx* c) {
    if (!c) return BCE_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    return sync_stream(c);
}

// ---- a whole step schedule at once (include/bce_gpu.h, "a whole step schedule at once") ---------------------------
}  // extern "C" (the object below is C++)

struct bce_plan {
    std::vector<u32> off, cnt;                 // step s = descriptors [off[s], off[s] + cnt[s])
    u32 instances = 0, slot_stride = 0;
    u32 max_step = 0;                          // descriptors of the largest step
    u64 boots_per_run = 0;
    bce_gate_desc* d_descs = nullptr;          // all steps, resident
    // captured form (bce_plan_run): its own accumulator / partial-sum scratch, so that nothing the graph points to is ever
    // re-allocated by other calls on the context
    void* d_acc = nullptr;
    u64* d_partial = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    u64 fused_launches = 0;
    // the captured kernels hold the context's device pointers and kernel choices by value: what they were at capture
    const void* cap_ptrs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    u32 cap_flags[3] = {0, 0, 0};
};

namespace {
bool plan_capture_is_current(const bce_ctx* c, const bce_plan* p) {
    const void* now[6] = {c->P.pool, c->P.bsk, c->P.bsk64, c->P.ksk, c->P.tw_f, c->P.psi_tab};
    const u32 flags[3] = {c->P.variant, c->P.fuse_tail, c->P.fold};
    return std::memcmp(now, p->cap_ptrs, sizeof now) == 0 && std::memcmp(flags, p->cap_flags, sizeof flags) == 0;
}
}  // namespace

extern "C" {

static void plan_free(bce_plan* p) {
    if (p->exec) hipGraphExecDestroy(p->exec);
    if (p->graph) hipGraphDestroy(p->graph);
    hipFree(p->d_descs); hipFree(p->d_acc); hipFree(p->d_partial);
    delete p;
}

void bce_plan_destroy(bce_ctx* c, bce_plan* p) {
    if (!p || !c) return;
    {   // a context that no longer exists took its plans along (bce_ctx_destroy): nothing left to free, nothing to touch
        std::lock_guard<std::mutex> lk(g_live_mu);
        if (!g_live.count(c) || !c->plans.erase(p)) return;
    }
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    plan_free(p);
}

int bce_plan_create(bce_ctx* c, uint32_t n_steps, const uint32_t* step_sizes, const bce_gate_desc* descs, uint32_t instances,
                    uint32_t slot_stride, uint32_t slot_base, bce_plan** out) {
    if (!c || !out) return BCE_ERR_ARG;
    *out = nullptr;
    if (!step_sizes || !descs || n_steps == 0 || instances == 0) return c->fail(BCE_ERR_ARG, "bce_plan_create: empty schedule");
    if (!c->have_keys) return c->fail(BCE_ERR_NO_KEYS, "bce_keygen / bce_import_keys has not been called");
    HIP_TRY(c, hipSetDevice(c->device));
    struct Deleter { bce_ctx* c; void operator()(bce_plan* p) const { bce_plan_destroy(c, p); } };
    std::unique_ptr<bce_plan, Deleter> p(new bce_plan, Deleter{c});
    { std::lock_guard<std::mutex> lk(g_live_mu); c->plans.insert(p.get()); }
    p->instances = instances; p->slot_stride = slot_stride;
    u64 total = 0;
    for (u32 s = 0; s < n_steps; ++s) {
        if (step_sizes[s] == 0) return c->fail(BCE_ERR_ARG, "bce_plan_create: step %u is empty", s);
        p->off.push_back((u32)total); p->cnt.push_back(step_sizes[s]);
        p->max_step = std::max(p->max_step, step_sizes[s]);
        total += step_sizes[s];
        if (total >= (1ull << 31)) return c->fail(BCE_ERR_ARG, "bce_plan_create: too many descriptors");
    }
    std::vector<bce_gate_desc> d(descs, descs + total);
    const u64 span = (u64)slot_base + (u64)(instances - 1) * slot_stride;
    for (u64 i = 0; i < total; ++i) {
        bce_gate_desc& g = d[i];
        const bool two = g.op <= BCE_XNOR_FAST;
        if (!two && g.op != BCE_OP_REFRESH) return c->fail(BCE_ERR_ARG, "bce_plan_create: descriptor %llu is not a bootstrapped gate (op %u)", (unsigned long long)i, g.op);
        u64 hi = std::max<u64>(g.in0, g.out);
        if (two) hi = std::max<u64>(hi, g.in1);
        if (hi + span >= c->pool_slots) return c->fail(BCE_ERR_POOL, "bce_plan_create: descriptor %llu: slot %llu outside the pool (%u slots)", (unsigned long long)i, (unsigned long long)(hi + span), c->pool_slots);
        g.in0 += slot_base; g.in1 += slot_base; g.out += slot_base;
    }
    p->boots_per_run = total * instances;
    HIP_TRY(c, hipMalloc(&p->d_descs, total * sizeof(bce_gate_desc)));
    HIP_TRY(c, hipMemcpy(p->d_descs, d.data(), total * sizeof(bce_gate_desc), hipMemcpyHostToDevice));
    *out = p.release();
    return BCE_OK;
}

int bce_plan_run_step(bce_ctx* c, bce_plan* p, uint32_t step) {
    if (!c || !p) return BCE_ERR_ARG;
    if (step >= p->cnt.size()) return c->fail(BCE_ERR_ARG, "bce_plan_run_step: step %u of %zu", step, p->cnt.size());
    if (!c->have_keys) return c->fail(BCE_ERR_NO_KEYS, "bce_keygen / bce_import_keys has not been called");
    HIP_TRY(c, hipSetDevice(c->device));
    const int rc = ensure_acc(c, (size_t)p->cnt[step] * p->instances);
    if (rc) return rc;
    const int rc2 = launch_bootstraps(c, p->d_descs + p->off[step], p->cnt[step], p->instances, p->slot_stride, c->d_acc, nullptr, nullptr, true, nullptr);
    if (rc2) return rc2;
    if (c->pending.size() > 4096) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        drain_timing(c);
    }
    return BCE_OK;
}

int bce_plan_run(bce_ctx* c, bce_plan* p) {
    if (!c || !p) return BCE_ERR_ARG;
    if (!c->have_keys) return c->fail(BCE_ERR_NO_KEYS, "bce_keygen / bce_import_keys has not been called");
    HIP_TRY(c, hipSetDevice(c->device));
    if (p->exec && !plan_capture_is_current(c, p)) {
        // the pool grew (bce_pool_reserve) or keys were re-imported since the capture: the graph's kernel arguments are stale
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        hipGraphExecDestroy(p->exec); p->exec = nullptr;
        hipGraphDestroy(p->graph); p->graph = nullptr;
    }
    if (!p->exec) {
        // scratch of the captured launches, sized for the largest step
        const size_t nb = (size_t)p->max_step * p->instances;
        if (!p->d_acc) HIP_TRY(c, hipMalloc(&p->d_acc, nb * 2 * c->N * c->wbytes));
        if (!p->d_partial) HIP_TRY(c, hipMalloc(&p->d_partial, std::max<size_t>(1, tail_partial_words(c->P, (u32)nb)) * sizeof(u64)));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        // relaxed mode: the launchers set kernel attributes (dynamic LDS size) while the stream is capturing
        HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
        int rc = BCE_OK;
        u64 fused = 0;
        for (size_t s = 0; s < p->cnt.size() && rc == BCE_OK; ++s) {
            bool f = false;
            rc = launch_bootstraps(c, p->d_descs + p->off[s], p->cnt[s], p->instances, p->slot_stride, p->d_acc, nullptr, nullptr, false, &f, p->d_partial);
            fused += f ? 1 : 0;
        }
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(c->stream, &g);   // always end the capture, also after a failed launch
        if (rc != BCE_OK) { if (g) hipGraphDestroy(g); return rc; }
        if (e != hipSuccess || !g) return c->fail(BCE_ERR_HIP, "bce_plan_run: stream capture failed: %s", hipGetErrorString(e));
        p->graph = g;
        p->fused_launches = fused;
        { const void* now[6] = {c->P.pool, c->P.bsk, c->P.bsk64, c->P.ksk, c->P.tw_f, c->P.psi_tab}; std::memcpy(p->cap_ptrs, now, sizeof now); }
        p->cap_flags[0] = c->P.variant; p->cap_flags[1] = c->P.fuse_tail; p->cap_flags[2] = c->P.fold;
        const hipError_t e2 = hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0);
        if (e2 != hipSuccess) { p->exec = nullptr; return c->fail(BCE_ERR_HIP, "bce_plan_run: hipGraphInstantiate: %s", hipGetErrorString(e2)); }
    }
    EventPair e0 = get_events(c, BCE_BR_GRAPH);
    hipEventRecord(e0.a, c->stream);
    HIP_TRY(c, hipGraphLaunch(p->exec, c->stream));
    hipEventRecord(e0.b, c->stream);
    c->pending.push_back(e0);
    c->timing.br_launches[BCE_BR_GRAPH] += p->cnt.size();
    c->timing.br_bootstraps[BCE_BR_GRAPH] += p->boots_per_run;
    c->timing.blind_rotate_launches += p->cnt.size();
    c->timing.bootstraps += p->boots_per_run;
    c->timing.fused_tail_launches += p->fused_launches;
    if (c->pending.size() > 4096) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        drain_timing(c);
    }
    return BCE_OK;
}

// ---- dependency-driven evaluation (include/bce_gpu.h, "the hot path, dependency-driven") --------------------------
}  // extern "C" (the object below is C++)

struct bce_dag {
    u32 n_tasks = 0;
    u32 max_slot = 0;
    u32 qcount[kDagQueues] = {0, 0, 0, 0};        // tasks per priority class
    u32 depth = 0;                                 // longest producer chain (bootstraps)
    std::vector<u32> h_dep_init;
    std::vector<u32> init_items;                   // initially ready tasks by class
    u32 init_off[kDagQueues + 1] = {0, 0, 0, 0, 0};
    // immutable device arrays
    bce_gate_desc* d_tasks = nullptr;
    u32 *d_cons_off = nullptr, *d_cons = nullptr, *d_dep_init = nullptr, *d_init = nullptr;
    uint8_t* d_qid = nullptr;
    // per-run state, grown on demand
    u32* d_dep = nullptr; size_t dep_cap = 0;
    u32* d_slots[kDagQueues] = {nullptr, nullptr, nullptr, nullptr}; size_t slots_cap[kDagQueues] = {0, 0, 0, 0};
    u32* d_ctl = nullptr;
    DagParams* d_params = nullptr;
};

extern "C" {

int bce_dag_supported(const bce_ctx* c) { return c && dag_kernel_available(c->P) ? 1 : 0; }

int bce_dag_set_limits(bce_ctx* c, int workgroups_per_cu, int placement, uint32_t lazy_us, uint32_t stall_ms) {
    if (!c || workgroups_per_cu < 0 || workgroups_per_cu > 2) return c ? c->fail(BCE_ERR_ARG, "workgroups_per_cu must be 0, 1 or 2") : BCE_ERR_ARG;
    if (stall_ms == 0 || stall_ms > 40000) return c->fail(BCE_ERR_ARG, "stall_ms must be in 1..40000");
    c->dag_wg_per_cu = workgroups_per_cu; c->dag_placement = placement ? 1 : 0; c->dag_lazy_us = lazy_us; c->dag_stall_ms = stall_ms;
    return BCE_OK;
}

int bce_dag_last_run(bce_ctx* c, uint64_t out[7]) {
    if (!c || !out) return BCE_ERR_ARG;
    for (int i = 0; i < 7; ++i) out[i] = c->dag_last[i];
    return BCE_OK;
}

void bce_dag_destroy(bce_ctx* c, bce_dag* g) {
    if (!g || !c) return;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        if (!g_live.count(c) || !c->dags.erase(g)) return;
    }
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    dag_free(g);
}

static void dag_free(bce_dag* g) {
    hipFree(g->d_tasks); hipFree(g->d_cons_off); hipFree(g->d_cons); hipFree(g->d_dep_init); hipFree(g->d_init); hipFree(g->d_qid);
    hipFree(g->d_dep); hipFree(g->d_ctl); hipFree(g->d_params);
    for (u32 q = 0; q < kDagQueues; ++q) hipFree(g->d_slots[q]);
    delete g;
}

int bce_dag_create(bce_ctx* c, uint32_t n_tasks, const bce_gate_desc* tasks, const uint8_t* prio, bce_dag** out) {
    if (!c || !out) return BCE_ERR_ARG;
    *out = nullptr;
    if (!tasks || n_tasks == 0) return c->fail(BCE_ERR_ARG, "bce_dag_create: empty task list");
    if (!dag_kernel_available(c->P)) return c->fail(BCE_ERR_UNSUPPORTED, "bce_dag_create: no persistent kernel for this parameter class (N = 1024, 4 gadget digits, Q < 2^28; N = 2048, Q < 2^39, AP with the folded key)");
    if ((u64)n_tasks >= (1ull << 31)) return c->fail(BCE_ERR_ARG, "bce_dag_create: too many tasks");
    HIP_TRY(c, hipSetDevice(c->device));
    u32 max_slot = 0;
    for (u32 i = 0; i < n_tasks; ++i) {
        const bce_gate_desc& g = tasks[i];
        const bool two = g.op <= BCE_XNOR_FAST;
        if (!two && g.op != BCE_OP_REFRESH) return c->fail(BCE_ERR_ARG, "bce_dag_create: task %u is not a bootstrapped gate (op %u)", i, g.op);
        if (prio && prio[i] >= kDagQueues) return c->fail(BCE_ERR_ARG, "bce_dag_create: task %u has priority class %u (0..%u)", i, prio[i], kDagQueues - 1);
        max_slot = std::max(max_slot, std::max(g.in0, g.out));
        if (two) max_slot = std::max(max_slot, g.in1);
    }
    // producers by slot: SSA form is required (a slot is written once and read only by later tasks), so that the order in
    // which the device runs independent tasks cannot matter
    std::vector<int32_t> writer((size_t)max_slot + 1, -1);
    std::vector<uint8_t> read_before((size_t)max_slot + 1, 0);
    std::vector<u32> dep(n_tasks, 0), lvl(n_tasks, 1), cons_cnt(n_tasks + 1, 0);
    std::vector<int32_t> p0(n_tasks, -1), p1(n_tasks, -1);
    u32 depth = 0;
    for (u32 i = 0; i < n_tasks; ++i) {
        const bce_gate_desc& g = tasks[i];
        const bool two = g.op <= BCE_XNOR_FAST;
        p0[i] = writer[g.in0]; read_before[g.in0] = 1;
        if (two) { p1[i] = writer[g.in1]; read_before[g.in1] = 1; }
        if (p1[i] == p0[i]) p1[i] = -1;
        if (writer[g.out] >= 0) return c->fail(BCE_ERR_ARG, "bce_dag_create: slot %u is written by tasks %d and %u (SSA form required)", g.out, writer[g.out], i);
        if (read_before[g.out]) return c->fail(BCE_ERR_ARG, "bce_dag_create: task %u writes slot %u that an earlier task (or itself) reads", i, g.out);
        writer[g.out] = (int32_t)i;
        for (int32_t p : {p0[i], p1[i]})
            if (p >= 0) { ++dep[i]; ++cons_cnt[p + 1]; lvl[i] = std::max(lvl[i], lvl[p] + 1); }
        depth = std::max(depth, lvl[i]);
    }
    std::vector<u32> cons_off(cons_cnt);
    for (u32 i = 0; i < n_tasks; ++i) cons_off[i + 1] += cons_off[i];
    std::vector<u32> cons(std::max<u32>(1, cons_off[n_tasks])), fill(cons_off.begin(), cons_off.end() - 1);
    for (u32 i = 0; i < n_tasks; ++i)
        for (int32_t p : {p0[i], p1[i]}) if (p >= 0) cons[fill[p]++] = i;
    std::vector<uint8_t> qid(n_tasks, 0);
    if (prio) qid.assign(prio, prio + n_tasks);

    struct Deleter { bce_ctx* c; void operator()(bce_dag* g) const { bce_dag_destroy(c, g); } };
    std::unique_ptr<bce_dag, Deleter> g(new bce_dag, Deleter{c});
    { std::lock_guard<std::mutex> lk(g_live_mu); c->dags.insert(g.get()); }
    g->n_tasks = n_tasks; g->max_slot = max_slot; g->depth = depth; g->h_dep_init = dep;
    for (u32 q = 0; q < kDagQueues; ++q) {
        g->init_off[q] = (u32)g->init_items.size();
        for (u32 i = 0; i < n_tasks; ++i) {
            if (qid[i] != q) continue;
            ++g->qcount[q];
            if (dep[i] == 0) g->init_items.push_back(i);
        }
    }
    g->init_off[kDagQueues] = (u32)g->init_items.size();
    if (g->init_items.empty()) return c->fail(BCE_ERR_ARG, "bce_dag_create: no task is ready at the start");
    auto up = [&](auto** dst, const auto& v) -> hipError_t {
        using T = std::remove_reference_t<decltype(v[0])>;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), std::max<size_t>(1, v.size()) * sizeof(T));
        if (e != hipSuccess) return e;
        return hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    };
    HIP_TRY(c, hipMalloc(&g->d_tasks, (size_t)n_tasks * sizeof(bce_gate_desc)));
    HIP_TRY(c, hipMemcpy(g->d_tasks, tasks, (size_t)n_tasks * sizeof(bce_gate_desc), hipMemcpyHostToDevice));
    HIP_TRY(c, up(&g->d_cons_off, cons_off));
    HIP_TRY(c, up(&g->d_cons, cons));
    HIP_TRY(c, up(&g->d_dep_init, dep));
    HIP_TRY(c, up(&g->d_init, g->init_items));
    HIP_TRY(c, up(&g->d_qid, qid));
    HIP_TRY(c, hipMalloc(&g->d_ctl, kDagCtlWords * sizeof(u32)));
    HIP_TRY(c, hipMalloc(&g->d_params, sizeof(DagParams)));
    *out = g.release();
    return BCE_OK;
}

int bce_dag_debug_block_task(bce_dag* g, uint32_t t) {
    if (!g || t >= g->n_tasks) return BCE_ERR_ARG;
    g->h_dep_init[t] += 1;
    return hipMemcpy(g->d_dep_init, g->h_dep_init.data(), (size_t)g->n_tasks * sizeof(u32), hipMemcpyHostToDevice) == hipSuccess ? BCE_OK : BCE_ERR_HIP;
}

int bce_dag_run(bce_ctx* c, bce_dag* g, uint32_t instances, uint32_t slot_stride, uint32_t slot_base) {
    if (!c || !g) return BCE_ERR_ARG;
    if (instances == 0) return BCE_OK;
    if (!c->have_keys) return c->fail(BCE_ERR_NO_KEYS, "bce_keygen / bce_import_keys has not been called");
    if (instances > 1 && slot_stride <= g->max_slot) return c->fail(BCE_ERR_ARG, "bce_dag_run: slot_stride %u does not cover the DAG's slots (0..%u)", slot_stride, g->max_slot);
    if ((u64)slot_base + g->max_slot + (u64)(instances - 1) * slot_stride >= c->pool_slots) return c->fail(BCE_ERR_POOL, "bce_dag_run: slot %llu outside the pool (%u slots)", (unsigned long long)((u64)slot_base + g->max_slot + (u64)(instances - 1) * slot_stride), c->pool_slots);
    const u64 items = (u64)g->n_tasks * instances;
    if (items >= 0xFFFFFFF0ull) return c->fail(BCE_ERR_ARG, "bce_dag_run: %llu bootstraps exceed one run's 32-bit item space", (unsigned long long)items);
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->dag_pending >= bce_ctx::kDagRuns) { const int rc = sync_stream(c); if (rc) return rc; }
    if (!c->d_P) HIP_TRY(c, hipMalloc(&c->d_P, sizeof(DevParams)));
    if (!c->h_dag_status) HIP_TRY(c, hipHostMalloc(&c->h_dag_status, sizeof(bce_ctx::DagStatus) * bce_ctx::kDagRuns));
    if (!c->h_dag_stage) HIP_TRY(c, hipHostMalloc(&c->h_dag_stage, sizeof(bce_ctx::DagStage) * bce_ctx::kDagRuns));
    // per-run state (a dag object serves one run at a time: runs are ordered on the engine's stream)
    if (items > g->dep_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        hipFree(g->d_dep); g->d_dep = nullptr;
        HIP_TRY(c, hipMalloc(&g->d_dep, items * sizeof(u32)));
        g->dep_cap = items;
    }
    DagParams D{};
    for (u32 q = 0; q < kDagQueues; ++q) {
        const size_t need = (size_t)g->qcount[q] * instances;
        if (need > g->slots_cap[q]) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            hipFree(g->d_slots[q]); g->d_slots[q] = nullptr;
            HIP_TRY(c, hipMalloc(&g->d_slots[q], need * sizeof(u32)));
            g->slots_cap[q] = need;
        }
        D.slots[q] = g->d_slots[q];
        D.qcap[q] = (u32)need;
        D.init_off[q] = g->init_off[q];
    }
    D.init_off[kDagQueues] = g->init_off[kDagQueues];
    D.tasks = g->d_tasks; D.cons_off = g->d_cons_off; D.cons = g->d_cons; D.qid = g->d_qid; D.dep_init = g->d_dep_init;
    D.dep = g->d_dep; D.init_items = g->d_init; D.ctl = g->d_ctl;
    D.n_tasks = g->n_tasks; D.instances = instances; D.slot_stride = slot_stride; D.slot_base = slot_base;
    D.lazy_ticks = c->dag_lazy_us * 100u;                     // s_memrealtime: 100 MHz
    D.stall_ticks = c->dag_stall_ms * 100000u;
    D.policy = c->dag_placement ? 1u : 0u;
    D.gate_ticks = 15000;        // 150 us: the finish times of bootstraps that started together spread less than that
    D.gate_backlog = 8;          // (64 / 16 / 4 / 1: AES K = 8 at 150 / 153 / 154 / 155 k; with the classes below 157 k, profiles/r03_dataflow_vs_steps.jsonl)
    if (const char* e = std::getenv("BCE_DAG_GATE_US")) D.gate_ticks = (u32)std::atoi(e) * 100u;
    if (const char* e = std::getenv("BCE_DAG_GATE_BACKLOG")) D.gate_backlog = (u32)std::atoi(e);
    // one or two workgroups per CU: two run 512 bootstraps per ~3.1 ms, one runs 256 per ~1.9 ms -- with less work per
    // dependency level than the CUs can hold alone, the shorter bootstrap wins
    int wps = c->dag_wg_per_cu == 1 ? 2 : 4;
    if (c->dag_wg_per_cu == 0) {
        // lower bound of the run under either residency, in units of one bootstrap with a CU to itself: the DAG's longest
        // chain against the work over the chip's rate.  Two workgroups per CU take 1.6 x as long per bootstrap (3.1 vs
        // 1.93 ms) and deliver 1.26 x the rate (165 vs 131 k/s); the cheaper bound wins (AES-expanded: K <= 2 one per CU,
        // K >= 4 two; md5 K = 16 and adder_64 K = 64 one.  Against the earlier rule "work per level <= 3/4 of the CUs":
        // adder_64 K = 64 97.9 -> 117.9 k/s, AES K = 2 115.7 -> 109.6 k, md5 K = 16 114.0 -> 110.1 k,
        // profiles/r03_dataflow_vs_steps.jsonl)
        const double D = (double)std::max<u32>(1, g->depth), W = (double)items, cus = (double)std::max<u32>(1, c->P.cu_count);
        const double t1 = std::max(D, W / cus), t2 = std::max(1.6 * D, W / (1.26 * cus));
        wps = t1 <= t2 ? 2 : 4;
    }
    if (const char* e = std::getenv("BCE_DAG_WPS")) { if (e[0] == '2') wps = 2; else if (e[0] == '4') wps = 4; }
    if (c->P.is64) wps = 2;      // the config-5 kernel: one 1,024-thread workgroup is all a CU's LDS holds
    if (const char* e = std::getenv("BCE_DAG_PLACE")) D.policy = e[0] == '0' ? 0u : 1u;
    const char* dbg = std::getenv("BCE_DAG_DEBUG");   // development: 'r' = re-arm only, 'd' = dry run (no bootstraps)
    if (dbg && dbg[0] == 'd') D.policy |= 2u;
    // the XCD start gate pays where two workgroups per CU share the L2 in the saturated regime (development knob BCE_DAG_GATE=0 / 1)
    { const char* e = std::getenv("BCE_DAG_GATE"); if (e ? e[0] != '0' : wps == 4) D.policy |= 4u; }
    { const char* e = std::getenv("BCE_DAG_TICKETS"); if (e && e[0] == '1') D.policy |= 8u; }   // development: round 3's claim rule
    const u32 grid = c->P.cu_count * (wps == 2 ? 1u : 2u);
    // parameter blocks reach the device in stream order (an earlier run may still be reading the previous ones) from
    // pinned staging entries that stay untouched until the next synchronisation
    const int slot = c->dag_pending++;
    c->h_dag_stage[slot].P = c->P;
    c->h_dag_stage[slot].D = D;
    HIP_TRY(c, hipMemcpyAsync(c->d_P, &c->h_dag_stage[slot].P, sizeof(DevParams), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(g->d_params, &c->h_dag_stage[slot].D, sizeof(DagParams), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, launch_dag_rearm(D, c->stream));
    EventPair e0 = get_events(c, BCE_BR_DAG);
    hipEventRecord(e0.a, c->stream);
    if (!(dbg && dbg[0] == 'r')) HIP_TRY(c, launch_bootstrap_dag(c->P, c->d_P, g->d_params, wps, grid, c->stream));
    hipEventRecord(e0.b, c->stream);
    c->pending.push_back(e0);
    c->dag_expected[slot] = items;
    c->dag_wps_used[slot] = wps;
    HIP_TRY(c, hipMemcpyAsync(&c->h_dag_status[slot], g->d_ctl + kDagAbort, sizeof(bce_ctx::DagStatus), hipMemcpyDeviceToHost, c->stream));
    c->timing.br_launches[BCE_BR_DAG] += 1;
    c->timing.br_bootstraps[BCE_BR_DAG] += items;
    c->timing.blind_rotate_launches += 1;
    c->timing.fused_tail_launches += 1;
    c->timing.bootstraps += items;
    return BCE_OK;
}

int bce_timing_reset(bce_ctx* c) {
    if (!c) return BCE_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    drain_timing(c);
    c->timing = bce_timing{};
    return BCE_OK;
}

int bce_timing_set_events(bce_ctx* c, int on) {
    if (!c) return BCE_ERR_ARG;
    c->events_on = on != 0;
    return BCE_OK;
}

int bce_timing_get(bce_ctx* c, bce_timing* out) {
    if (!c || !out) return BCE_ERR_ARG;
    const int rc = sync_stream(c);
    drain_timing(c);
    *out = c->timing;
    return rc;
}

uint32_t bce_forward_transforms_per_step(const bce_ctx* c) { return c ? 2 * c->dG - (c->P.fold ? 2 : 0) : 0; }

int bce_launch_capacity(const bce_ctx* c, uint32_t* lone, uint32_t* full) {
    if (!c || !lone || !full) return BCE_ERR_ARG;
    const DevParams& P = c->P;
    const u32 cu = P.cu_count;
    u32 per_cu;
    if (c->is64) {
        // 64-bit kernels: LDS-bound residency (N = 2048: one workgroup per CU)
        per_cu = (u32)std::max<size_t>(1, (160 * 1024) / blind_rotate64_lds_bytes(P));
        if (c->logN == 11) per_cu = 1;
    } else if (P.variant != 1 && c->logN == 10 && c->dG == 4 && P.lazy) {
        per_cu = 2;                                    // split-transform kernel, 128-register build
    } else {
        per_cu = P.occupancy_target;                   // one wave per transform: 2 or 3 workgroups per CU
    }
    *lone = cu;
    *full = cu * per_cu;
    return BCE_OK;
}

int bce_bytes_per_bootstrap_parts(const bce_ctx* c, uint64_t out[3]) {
    if (!c || !out) return BCE_ERR_ARG;
    const double rgsws = c->method == BCE_AP ? (double)c->n * c->dR * (c->baseR - 1) / c->baseR : 2.0 * c->n;
    out[0] = (u64)((double)c->wbytes * rgsws * (2 * c->dG) * 2 * c->N);
    out[1] = (c->P.ksk_u16 ? 2ull : 4ull) * c->N * c->dKS * (c->n + 1);
    out[2] = 4ull * 3 * (c->n + 1);
    return BCE_OK;
}

uint64_t bce_bytes_per_bootstrap(const bce_ctx* c) {
    if (!c) return 0;
    // w_bsk*[RGSWs touched * (2dG)*2*N] + w_ks*[N*dKS*(n+1)] + w_ct*[3(n+1)] at this build's widths.
    // GINX touches 2 RGSW ciphertexts per LWE coefficient, AP one per non-zero base-baseR digit
    // (dR * (baseR-1)/baseR on average).
    const double rgsws = c->method == BCE_AP ? (double)c->n * c->dR * (c->baseR - 1) / c->baseR : 2.0 * c->n;
    const u64 bsk = (u64)((double)c->wbytes * rgsws * (2 * c->dG) * 2 * c->N);
    const u64 ks = (c->P.ksk_u16 ? 2ull : 4ull) * c->N * c->dKS * (c->n + 1);
    const u64 ct = 4ull * 3 * (c->n + 1);
    return bsk + ks + ct;
}

int bce_debug_eval_stages(bce_ctx* c, uint32_t n_desc, const bce_gate_desc* descs, uint64_t* acc, uint64_t* lweN,
                          uint64_t* ks) {
    if (!c) return BCE_ERR_ARG;
    for (u32 i = 0; i < n_desc; ++i)
        if (!(descs[i].op <= BCE_XNOR_FAST || descs[i].op == BCE_OP_REFRESH)) return c->fail(BCE_ERR_ARG, "staged outputs need bootstrapped ops only");
    return eval_impl(c, n_desc, descs, 1, 0, acc, lweN, ks);
}

// The tail of EvalBinGate alone (transpose + extract, ModSwitch Q -> qKS, KeySwitch, ModSwitch qKS -> q) on accumulators
// the CALLER supplies: what tools/openfhe_export/compare.py replays OpenFHE's LWEEncryptionScheme::ModSwitch / KeySwitch
// records on, so that a mismatch in a gate vector can be told apart from one in the tail.  Runs the separate tail kernels.
int bce_debug_tail(bce_ctx* c, uint32_t count, const uint64_t* acc, const uint32_t* out_slots, uint64_t* lweN, uint64_t* ks) {
    if (!c || !acc || !out_slots) return BCE_ERR_ARG;
    if (count == 0) return BCE_OK;
    if (!c->have_keys) return c->fail(BCE_ERR_NO_KEYS, "bce_keygen / bce_import_keys has not been called");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t words = (size_t)count * 2 * c->N;
    for (size_t i = 0; i < words; ++i)
        if (acc[i] >= c->Q) return c->fail(BCE_ERR_ARG, "accumulator word not reduced mod Q");
    std::vector<bce_gate_desc> d(count);
    for (u32 i = 0; i < count; ++i) {
        if (out_slots[i] >= c->pool_slots) return c->fail(BCE_ERR_POOL, "slot %u outside the pool", out_slots[i]);
        d[i] = bce_gate_desc{BCE_AND, out_slots[i], out_slots[i], out_slots[i], 0, 0};
    }
    void* d_in = nullptr;
    u32 *d_lweN = nullptr, *d_ks = nullptr;
    u64* d_partial = nullptr;
    HIP_TRY(c, hipMalloc(&d_in, words * c->wbytes));
    HIP_TRY(c, hipMalloc(&d_lweN, (size_t)count * (c->N + 1) * sizeof(u32)));
    HIP_TRY(c, hipMalloc(&d_ks, (size_t)count * (c->n + 1) * sizeof(u32)));
    HIP_TRY(c, hipMalloc(&d_partial, std::max<size_t>(1, tail_partial_words(c->P, count)) * sizeof(u64)));
    std::vector<u32> tmp;
    if (c->is64) {
        HIP_TRY(c, hipMemcpy(d_in, acc, words * 8, hipMemcpyHostToDevice));
    } else {
        tmp.assign(acc, acc + words);
        HIP_TRY(c, hipMemcpy(d_in, tmp.data(), words * 4, hipMemcpyHostToDevice));
    }
    bce_gate_desc* dd = nullptr;
    int slot = 0;
    int rc = stage_descs(c, d.data(), count, &dd, &slot);
    if (rc) return rc;
    HIP_TRY(c, launch_tail(c->P, dd, count, 1, 0, d_in, d_partial, d_lweN, d_ks, c->stream, LaunchEvents{}));
    hipEventRecord(c->ring_ev[slot], c->stream);
    c->ring_busy[slot] = true;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    auto fetch = [&](const u32* dev, size_t n, u64* dst) -> int {
        tmp.resize(n);
        HIP_TRY(c, hipMemcpy(tmp.data(), dev, n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) dst[i] = tmp[i];
        return BCE_OK;
    };
    if (lweN && (rc = fetch(d_lweN, (size_t)count * (c->N + 1), lweN))) return rc;
    if (ks && (rc = fetch(d_ks, (size_t)count * (c->n + 1), ks))) return rc;
    hipFree(d_in); hipFree(d_lweN); hipFree(d_ks); hipFree(d_partial);
    return BCE_OK;
}

static int pool_pack(bce_ctx* c, const uint32_t* slots, uint32_t count, void* dev, int to_pool) {
    if (!c || !slots || !dev) return BCE_ERR_ARG;
    if (count == 0) return BCE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<bce_gate_desc> d(count);
    for (u32 i = 0; i < count; ++i) {
        if (slots[i] >= c->pool_slots) return c->fail(BCE_ERR_POOL, "slot %u outside the pool", slots[i]);
        d[i] = bce_gate_desc{BCE_OP_COPY, slots[i], slots[i], slots[i], 0, 0};
    }
    bce_gate_desc* dd = nullptr;
    int slot = 0;
    int rc = stage_descs(c, d.data(), count, &dd, &slot);
    if (rc) return rc;
    HIP_TRY(c, launch_pool_pack(c->P, dd, count, (u32*)dev, to_pool, c->stream));
    hipEventRecord(c->ring_ev[slot], c->stream);
    c->ring_busy[slot] = true;
    return BCE_OK;
}

int bce_pool_gather(bce_ctx* c, const uint32_t* slots, uint32_t count, void* dev_dst) {
    return pool_pack(c, slots, count, dev_dst, 0);
}
int bce_pool_scatter(bce_ctx* c, const uint32_t* slots, uint32_t count, const void* dev_src) {
    return pool_pack(c, slots, count, const_cast<void*>(dev_src), 1);
}

int bce_debug_ntt(bce_ctx* c, uint64_t* polys, uint32_t count, int inverse) {
    if (!c || !polys) return BCE_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t words = (size_t)count * c->N;
    for (size_t i = 0; i < words; ++i)
        if (polys[i] >= c->Q) return c->fail(BCE_ERR_ARG, "poly word not reduced mod Q");
    void* d = nullptr;
    HIP_TRY(c, hipMalloc(&d, words * c->wbytes));
    std::vector<u32> tmp;
    if (c->is64) {
        HIP_TRY(c, hipMemcpy(d, polys, words * 8, hipMemcpyHostToDevice));
    } else {
        tmp.resize(words);
        for (size_t i = 0; i < words; ++i) tmp[i] = (u32)polys[i];
        HIP_TRY(c, hipMemcpy(d, tmp.data(), words * 4, hipMemcpyHostToDevice));
    }
    int rc = dev_ntt(c, d, count, inverse);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->is64) {
        HIP_TRY(c, hipMemcpy(polys, d, words * 8, hipMemcpyDeviceToHost));
    } else {
        HIP_TRY(c, hipMemcpy(tmp.data(), d, words * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < words; ++i) polys[i] = tmp[i];
    }
    hipFree(d);
    return BCE_OK;
}

}  // extern "C"
