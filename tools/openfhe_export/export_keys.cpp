// export_keys.cpp -- OpenFHE-side exporter: writes the key material of an lbcrypto::BinFHEContext in the exchange
// format of bce_keyfile.h, which libbce_amd.so loads with bce_import_keys_file().
//
// WHY: the reference draws its keys with cc.KeyGen() / cc.BTKeyGen(sk) (/root/reference/src/circuit.cpp:90-91) from
// OpenFHE's unseeded PRNG.  Feeding the SAME keys and ciphertexts to the engine is the only way to state
// "bit-exact against the reference's own encrypted path"; this program is the producer side of that comparison.
//
// STATUS: OpenFHE (and Boost, which the reference also needs) are not installed in the environment this repository
// is developed in, so this file has NOT been compiled there.  It is written against the public binfhe API of
// openfhe-development v1.0.x (the version the reference names, Release_Notes.md:4):
//   BinFHEContext::GetParams(), GetRefreshKey() -> RingGSWACCKey, GetSwitchKey() -> LWESwitchingKey,
//   RingGSWACCKeyImpl::operator[] ([0][key][i] for GINX, [i][v][k] for AP),
//   RingGSWEvalKeyImpl::GetElements() -> std::vector<std::vector<NativePoly>> (EVALUATION format),
//   LWESwitchingKeyImpl::GetElementsA() / GetElementsB(), LWEPrivateKeyImpl::GetElement().
// The consumer side (the file format, its loader and the parity of keys loaded from such a file) IS tested:
// tests/test_gpu_keyfile.py writes the same format from the CPU oracle's keys.
//
// build (on a machine with OpenFHE >= 1.0.1 installed):
//   cmake -S tools/openfhe_export -B build_export && cmake --build build_export
// Environment: BCE_EXPORT_EVALUATION=1 dumps the bootstrapping key as OpenFHE holds it (EVALUATION representation, no
// SetFormat pass over the key) and marks the file bsk_format = 1; the engine imports that without a transform
// (bce_import_keys_eval): the engine's evaluation order is OpenFHE's (bit-reversed CT order, minimal primitive 2N-th root).
// usage:
//   export_keys <TOY|STD128_OPT|...> <AP|GINX> <keys.bce> [--vectors <vectors.bgv> <M>] [<ciphertexts.bin> <bit> ...]
//     generates a context + keys exactly like the reference's Circuit constructor and writes the keys;
//     --vectors: also records what OpenFHE RETURNS for M (e.g. 64) random EvalBinGate calls over all six gates (inputs
//       are fresh encryptions and earlier gate outputs), EvalNOT, Bootstrap, Encrypt in the default and the FRESH mode,
//       Decrypt of every result, plus tail and NTT probes, in the "BCEGVEC1" format of bce_keyfile.h.  Then, on a machine
//       with an MI355X:  python tools/openfhe_export/compare.py keys.bce vectors.bgv   (exit code 0 = every word equal);
//     optionally encrypts the given bits and appends them as u64[n+1] words each (a_0..a_{n-1}, b) mod q,
//     the layout bce_lwe_write() takes.
// The probes use LWEEncryptionScheme::ModSwitch(q, ct) / KeySwitch(params, K, ct) and NativePoly::SetFormat as of v1.0.x;
// build with -DBCE_EXPORT_NO_PROBES if the installed release spells them differently (the gate records do not need them).
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "binfhecontext.h"

#include "bce_keyfile.h"

static bool g_evaluation_form = false;   // BCE_EXPORT_EVALUATION=1

using namespace lbcrypto;

namespace {

void put(FILE* f, const void* p, size_t bytes) {
    if (std::fwrite(p, 1, bytes, f) != bytes) throw std::runtime_error("short write");
}

uint32_t digit_count(double modulus, double base) { return (uint32_t)std::ceil(std::log(modulus) / std::log(base)); }

// one RGSW ciphertext: R rows x 2 polynomials, COEFFICIENT representation, natural coefficient order
void put_rgsw(FILE* f, const RingGSWEvalKey& ek, uint32_t N) {
    std::vector<uint64_t> words(N);
    for (const auto& row : ek->GetElements())
        for (NativePoly poly : row) {  // by value: SetFormat on a copy
            if (!g_evaluation_form) poly.SetFormat(Format::COEFFICIENT);
            for (uint32_t k = 0; k < N; ++k) words[k] = poly[k].ConvertToInt();
            put(f, words.data(), N * sizeof(uint64_t));
        }
}

std::vector<uint64_t> words_of(ConstLWECiphertext ct) {
    const uint32_t n = ct->GetA().GetLength();
    std::vector<uint64_t> w(n + 1);
    for (uint32_t k = 0; k < n; ++k) w[k] = ct->GetA()[k].ConvertToInt();
    w[n] = ct->GetB().ConvertToInt();
    return w;
}

struct VectorFile {
    FILE* f = nullptr;
    uint64_t count = 0;
    void open(const char* path, const bce_keyfile_header& k) {
        f = std::fopen(path, "wb");
        if (!f) throw std::runtime_error("cannot open the vector file for writing");
        bce_gatevec_header h{};
        std::memcpy(h.magic, BCE_GATEVEC_MAGIC, 8);
        h.version = BCE_GATEVEC_VERSION;
        h.method = k.method;
        h.n = k.n; h.N = k.N; h.q = k.q; h.Q = k.Q; h.qKS = k.qKS; h.baseKS = k.baseKS; h.baseG = k.baseG; h.baseR = k.baseR;
        h.count = 0;  // patched by close()
        put(f, &h, sizeof h);
    }
    void record(uint32_t kind, uint32_t in_bits, uint32_t decrypted, const std::vector<const std::vector<uint64_t>*>& parts) {
        bce_gatevec_record r{kind, in_bits, decrypted, 0};
        for (const auto* p : parts) r.payload_words += (uint32_t)p->size();
        put(f, &r, sizeof r);
        for (const auto* p : parts) put(f, p->data(), p->size() * sizeof(uint64_t));
        ++count;
    }
    void close() {
        std::fseek(f, offsetof(bce_gatevec_header, count), SEEK_SET);
        put(f, &count, sizeof count);
        std::fclose(f);
        f = nullptr;
    }
};

// What OpenFHE returns for the calls the reference makes (src/gate.cpp:112,133,146,172,198-202; src/circuit.cpp:506,800)
void write_vectors(BinFHEContext& cc, LWEPrivateKey sk, const bce_keyfile_header& kh, const char* path, uint32_t M) {
    VectorFile vf;
    vf.open(path, kh);
    std::mt19937_64 rng(0xB0017E57ull);   // chooses gates, operands and bits only; all ciphertext randomness is OpenFHE's
    struct Operand { LWECiphertext ct; uint32_t bit; };
    std::vector<Operand> live;
    auto decrypt = [&](ConstLWECiphertext ct) { LWEPlaintext r = 0; cc.Decrypt(sk, ct, &r); return (uint32_t)r; };

    for (uint32_t i = 0; i < 8; ++i) {    // fresh inputs; and what cc.Encrypt(sk, bit) -- the call at src/circuit.cpp:506 -- returns
        const uint32_t bit = (uint32_t)(rng() & 1);
        LWECiphertext fresh = cc.Encrypt(sk, bit, FRESH);
        const auto wf = words_of(fresh);
        vf.record(BCE_GATEVEC_ENCRYPT_FRESH, bit, decrypt(fresh), {&wf});
        live.push_back({fresh, bit});
        LWECiphertext dflt = cc.Encrypt(sk, bit);
        const auto wd = words_of(dflt);
        vf.record(BCE_GATEVEC_ENCRYPT_DEFAULT, bit, decrypt(dflt), {&wd});
        live.push_back({dflt, bit});
    }
    for (uint32_t g = 0; g < M; ++g) {
        const uint32_t gate = (uint32_t)(rng() % 6);
        size_t i = rng() % live.size(), j = rng() % live.size();
        if (i == j) j = (j + 1) % live.size();   // EvalBinGate refuses ct1 == ct2
        LWECiphertext out = cc.EvalBinGate((BINGATE)gate, live[i].ct, live[j].ct);
        const auto w1 = words_of(live[i].ct), w2 = words_of(live[j].ct), wo = words_of(out);
        const uint32_t dec = decrypt(out);
        vf.record(gate, live[i].bit | (live[j].bit << 1), dec, {&w1, &w2, &wo});
        live.push_back({out, dec & 1});
    }
    for (uint32_t k = 0; k < 6; ++k) {
        const Operand& a = live[rng() % live.size()];
        const auto wi = words_of(a.ct);
        LWECiphertext nt = cc.EvalNOT(a.ct);
        const auto wn = words_of(nt);
        vf.record(BCE_GATEVEC_NOT, a.bit, decrypt(nt), {&wi, &wn});
        LWECiphertext bt = cc.Bootstrap(a.ct);
        const auto wb = words_of(bt);
        vf.record(BCE_GATEVEC_BOOTSTRAP, a.bit, decrypt(bt), {&wi, &wb});
    }
#ifndef BCE_EXPORT_NO_PROBES
    {
        const auto lwe = cc.GetParams()->GetLWEParams();
        const auto scheme = cc.GetLWEScheme();
        const uint32_t N = lwe->GetN();
        const NativeInteger Q = lwe->GetQ();
        std::uniform_int_distribution<uint64_t> modQ(0, Q.ConvertToInt() - 1);
        for (uint32_t t = 0; t < 4; ++t) {   // the calls EvalBinGate makes after the accumulator (binfhe-base-scheme.cpp)
            NativeVector a(N, Q);
            for (uint32_t k = 0; k < N; ++k) a[k] = NativeInteger(modQ(rng));
            LWECiphertext ctQ = std::make_shared<LWECiphertextImpl>(std::move(a), NativeInteger(modQ(rng)));
            LWECiphertext ctKS = scheme->ModSwitch(lwe->GetqKS(), ctQ);
            LWECiphertext ks = scheme->KeySwitch(lwe, cc.GetSwitchKey(), ctKS);
            LWECiphertext out = scheme->ModSwitch(lwe->Getq(), ks);
            const auto w0 = words_of(ctQ), w1 = words_of(ctKS), w2 = words_of(ks), w3 = words_of(out);
            vf.record(BCE_GATEVEC_TAIL, 0, 0, {&w0, &w1, &w2, &w3});
        }
        const auto poly_params = cc.GetParams()->GetRingGSWParams()->GetPolyParams();
        for (uint32_t t = 0; t < 2; ++t) {   // the evaluation order a bsk_format = 1 key file relies on
            NativePoly p(poly_params, Format::COEFFICIENT, true);
            std::vector<uint64_t> coef(N), eval(N);
            for (uint32_t k = 0; k < N; ++k) { coef[k] = modQ(rng); p[k] = NativeInteger(coef[k]); }
            p.SetFormat(Format::EVALUATION);
            for (uint32_t k = 0; k < N; ++k) eval[k] = p[k].ConvertToInt();
            vf.record(BCE_GATEVEC_NTT, 0, 0, {&coef, &eval});
        }
    }
#endif
    vf.close();
}

}  // namespace

int main(int argc, char** argv) {
    { const char* e = std::getenv("BCE_EXPORT_EVALUATION"); g_evaluation_form = e && e[0] == '1'; }
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <paramset> <AP|GINX> <keys.bce> [--vectors <vectors.bgv> <M>] [<cts.bin> <bit> ...]\n", argv[0]);
        return 2;
    }
    const std::map<std::string, BINFHE_PARAMSET> sets = {
        {"TOY", TOY}, {"MEDIUM", MEDIUM}, {"STD128_AP", STD128_AP}, {"STD128_APOPT", STD128_APOPT}, {"STD128", STD128},
        {"STD128_OPT", STD128_OPT}, {"STD192", STD192}, {"STD192_OPT", STD192_OPT}, {"STD256", STD256}, {"STD256_OPT", STD256_OPT}};
    const auto ps = sets.find(argv[1]);
    if (ps == sets.end()) throw std::invalid_argument("unknown parameter set");
    const BINFHE_METHOD method = std::string(argv[2]) == "AP" ? AP : GINX;

    // exactly the reference's sequence (src/circuit.cpp:65,88-91)
    BinFHEContext cc;
    cc.GenerateBinFHEContext(ps->second, method);
    LWEPrivateKey sk = cc.KeyGen();
    cc.BTKeyGen(sk);

    const auto params = cc.GetParams();
    const auto lwe = params->GetLWEParams();
    const auto rgsw = params->GetRingGSWParams();
    const uint32_t n = lwe->Getn(), N = lwe->GetN();
    const uint64_t q = lwe->Getq().ConvertToInt(), Q = lwe->GetQ().ConvertToInt(), qKS = lwe->GetqKS().ConvertToInt();
    const uint32_t baseKS = lwe->GetBaseKS(), baseG = rgsw->GetBaseG(), baseR = rgsw->GetBaseR();
    const uint32_t dG = digit_count((double)Q, (double)baseG), R = 2 * dG;
    const uint32_t dR = digit_count((double)q, (double)baseR), dKS = digit_count((double)qKS, (double)baseKS);

    bce_keyfile_header h{};
    std::memcpy(h.magic, BCE_KEYFILE_MAGIC, 8);
    h.version = BCE_KEYFILE_VERSION;
    h.method = method == AP ? 1 : 2;
    h.n = n; h.N = N; h.q = q; h.Q = Q; h.qKS = qKS; h.baseKS = baseKS; h.baseG = baseG; h.baseR = baseR;
    h.bsk_words = (method == AP ? (uint64_t)n * baseR * dR : (uint64_t)n * 2) * R * 2 * N;
    h.ksk_words = (uint64_t)N * baseKS * dKS * (n + 1);
    h.has_z = 0;  // BTKeyGen does not keep the ring secret; evaluation does not need it
    h.bsk_format = g_evaluation_form ? BCE_KEYFILE_BSK_EVALUATION : BCE_KEYFILE_BSK_COEFFICIENT;

    FILE* f = std::fopen(argv[3], "wb");
    if (!f) throw std::runtime_error("cannot open the key file for writing");
    put(f, &h, sizeof h);
    {   // ternary LWE secret, centred
        std::vector<int32_t> s(n);
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t v = sk->GetElement()[i].ConvertToInt();
            s[i] = v == 0 ? 0 : (v == 1 ? 1 : -1);
        }
        put(f, s.data(), n * sizeof(int32_t));
        if ((n * sizeof(int32_t)) % 8) { const uint32_t zero = 0; put(f, &zero, 4); }
    }
    const RingGSWACCKey ek = cc.GetRefreshKey();
    if (method == GINX) {
        // rgsw-acc-cggi.cpp KeyGenAcc: (*ek)[0][0][i] encrypts (s_i == 1), (*ek)[0][1][i] encrypts (s_i == -1)
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t key = 0; key < 2; ++key) put_rgsw(f, (*ek)[0][key][i], N);
    } else {
        // rgsw-acc-dm.cpp KeyGenAcc: (*ek)[i][v][k] encrypts X^{s_i * v * baseR^k * 2N/q}; the v = 0 entries are unused
        std::vector<uint64_t> zeros((size_t)R * 2 * N, 0);
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t v = 0; v < baseR; ++v)
                for (uint32_t k = 0; k < dR; ++k) {
                    if (v == 0) put(f, zeros.data(), zeros.size() * sizeof(uint64_t));
                    else put_rgsw(f, (*ek)[i][v][k], N);
                }
    }
    const LWESwitchingKey ks = cc.GetSwitchKey();
    {
        std::vector<uint32_t> row(n + 1);
        for (uint32_t i = 0; i < N; ++i)
            for (uint32_t v = 0; v < baseKS; ++v)
                for (uint32_t j = 0; j < dKS; ++j) {
                    for (uint32_t k = 0; k < n; ++k) row[k] = (uint32_t)ks->GetElementsA()[i][v][j][k].ConvertToInt();
                    row[n] = (uint32_t)ks->GetElementsB()[i][v][j].ConvertToInt();
                    put(f, row.data(), row.size() * sizeof(uint32_t));
                }
    }
    if (g_evaluation_form) {
        // two (coefficient, evaluation) pairs: the importer checks that its own forward transform reproduces OpenFHE's order
        // before it accepts a key dumped without SetFormat (bce_keyfile.h, trailer)
        put(f, BCE_KEYFILE_NTTCHECK_MAGIC, 8);
        const uint32_t count[2] = {2, 0};
        put(f, count, sizeof count);
        std::mt19937_64 rng(0x77C0FFEEull);
        const auto poly_params = rgsw->GetPolyParams();
        for (uint32_t t = 0; t < 2; ++t) {
            NativePoly p(poly_params, Format::COEFFICIENT, true);
            std::vector<uint64_t> coef(N), eval(N);
            for (uint32_t k = 0; k < N; ++k) { coef[k] = rng() % Q; p[k] = NativeInteger(coef[k]); }
            p.SetFormat(Format::EVALUATION);
            for (uint32_t k = 0; k < N; ++k) eval[k] = p[k].ConvertToInt();
            put(f, coef.data(), N * sizeof(uint64_t));
            put(f, eval.data(), N * sizeof(uint64_t));
        }
    }
    std::fclose(f);

    int rest = 4;
    if (argc >= rest + 3 && std::string(argv[rest]) == "--vectors") {
        write_vectors(cc, sk, h, argv[rest + 1], (uint32_t)std::atoi(argv[rest + 2]));
        rest += 3;
    }
    if (argc >= rest + 2) {  // ciphertexts for a gate-level comparison: u64[n+1] each
        FILE* c = std::fopen(argv[rest], "wb");
        if (!c) throw std::runtime_error("cannot open the ciphertext file for writing");
        std::vector<uint64_t> words(n + 1);
        for (int a = rest + 1; a < argc; ++a) {
            const LWECiphertext ct = cc.Encrypt(sk, std::atoi(argv[a]), FRESH);
            for (uint32_t k = 0; k < n; ++k) words[k] = ct->GetA()[k].ConvertToInt();
            words[n] = ct->GetB().ConvertToInt();
            put(c, words.data(), words.size() * sizeof(uint64_t));
        }
        std::fclose(c);
    }
    return 0;
}
