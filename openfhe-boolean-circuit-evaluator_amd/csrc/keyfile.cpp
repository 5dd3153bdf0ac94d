// keyfile.cpp -- key material on disk (tools/openfhe_export/bce_keyfile.h): bce_import_keys_file /
// bce_export_keys_file of include/bce_gpu.h.
//
// This is the consumer of what an OpenFHE-side exporter writes (tools/openfhe_export/export_keys.cpp): the keys the
// reference obtains from cc.KeyGen() / cc.BTKeyGen(sk) (src/circuit.cpp:90-91), in coefficient representation.
// The file is memory-mapped and handed to bce_import_keys(), which streams the bootstrapping key to the device in
// chunks and transforms it there.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bce_gpu.h"
#include "../../tools/openfhe_export/bce_keyfile.h"

extern "C" int bce_set_error(bce_ctx* c, int code, const char* msg);  // engine.cpp

static_assert(sizeof(bce_keyfile_header) == 104, "on-disk header layout (tools/openfhe_export/bce_keyfile.h)");

namespace {
struct Mapping {
    int fd = -1;
    void* p = MAP_FAILED;
    size_t len = 0;
    ~Mapping() {
        if (p != MAP_FAILED) munmap(p, len);
        if (fd >= 0) close(fd);
    }
};
}  // namespace

extern "C" {

int bce_import_keys_file(bce_ctx* c, const char* path) {
    if (!c || !path) return BCE_ERR_ARG;
    Mapping m;
    m.fd = open(path, O_RDONLY);
    if (m.fd < 0) return bce_set_error(c, BCE_ERR_ARG, (std::string("cannot open key file ") + path).c_str());
    struct stat st;
    if (fstat(m.fd, &st) != 0 || (size_t)st.st_size < sizeof(bce_keyfile_header)) return bce_set_error(c, BCE_ERR_ARG, "key file shorter than its header");
    m.len = (size_t)st.st_size;
    m.p = mmap(nullptr, m.len, PROT_READ, MAP_PRIVATE, m.fd, 0);
    if (m.p == MAP_FAILED) return bce_set_error(c, BCE_ERR_STATE, "mmap of the key file failed");
    const char* base = static_cast<const char*>(m.p);
    bce_keyfile_header h;
    std::memcpy(&h, base, sizeof h);
    if (std::memcmp(h.magic, BCE_KEYFILE_MAGIC, 8) != 0) return bce_set_error(c, BCE_ERR_ARG, "not a BCEKEYS1 key file (bad magic)");
    if (h.version != BCE_KEYFILE_VERSION) return bce_set_error(c, BCE_ERR_UNSUPPORTED, "unsupported key file version");
    uint64_t p[BCE_P_COUNT];
    bce_get_params(c, p);
    const uint64_t want[8] = {p[BCE_P_n], p[BCE_P_N], p[BCE_P_q], p[BCE_P_Q], p[BCE_P_qKS], p[BCE_P_baseKS], p[BCE_P_baseG], p[BCE_P_baseR]};
    const uint64_t got[8] = {h.n, h.N, h.q, h.Q, h.qKS, h.baseKS, h.baseG, h.baseR};
    static const char* names[8] = {"n", "N", "q", "Q", "qKS", "baseKS", "baseG", "baseR"};
    for (int k = 0; k < 8; ++k)
        if (want[k] != got[k]) {
            char buf[160];
            std::snprintf(buf, sizeof buf, "key file parameter %s = %llu does not match the context (%llu)", names[k],
                          (unsigned long long)got[k], (unsigned long long)want[k]);
            return bce_set_error(c, BCE_ERR_ARG, buf);
        }
    if (h.method != p[BCE_P_method]) return bce_set_error(c, BCE_ERR_ARG, "key file method (AP / GINX) does not match the context");
    if (h.bsk_words != bce_bsk_words(c) || h.ksk_words != bce_ksk_words(c)) return bce_set_error(c, BCE_ERR_ARG, "key file word counts do not match the parameter set");
    size_t off = sizeof h;
    const size_t s_bytes = h.n * 4, z_bytes = h.has_z ? h.N * 4 : 0;
    const size_t bsk_off = (off + s_bytes + z_bytes + 7) & ~(size_t)7;
    const size_t need = bsk_off + h.bsk_words * 8 + h.ksk_words * 4;
    if (m.len < need) return bce_set_error(c, BCE_ERR_ARG, "key file is truncated");
    const int32_t* s = reinterpret_cast<const int32_t*>(base + off);
    const int32_t* z = h.has_z ? reinterpret_cast<const int32_t*>(base + off + s_bytes) : nullptr;
    for (uint64_t i = 0; i < h.n; ++i)
        if (s[i] < -1 || s[i] > 1) return bce_set_error(c, BCE_ERR_ARG, "key file: LWE secret entries must be -1, 0 or 1");
    const uint64_t* bsk = reinterpret_cast<const uint64_t*>(base + bsk_off);
    const uint32_t* ksk = reinterpret_cast<const uint32_t*>(base + bsk_off + h.bsk_words * 8);
    for (uint64_t i = 0; i < h.ksk_words; ++i)
        if (ksk[i] >= h.qKS) return bce_set_error(c, BCE_ERR_ARG, "key file: key-switching word not reduced mod qKS");
    if (h.bsk_format > BCE_KEYFILE_BSK_EVALUATION) return bce_set_error(c, BCE_ERR_UNSUPPORTED, "key file: unknown bootstrapping-key representation");
    for (uint64_t k = 0; h.has_z && k < h.N; ++k)
        if (z[k] < -1 || z[k] > 1) return bce_set_error(c, BCE_ERR_ARG, "key file: ring secret entries must be -1, 0 or 1");
    // optional trailer: (coefficient, evaluation) pairs of the producer.  An evaluation-form key is only usable if the
    // producer's evaluation order is this engine's: check it where it can be checked instead of trusting the header
    if (m.len >= need + 16 && std::memcmp(base + need, BCE_KEYFILE_NTTCHECK_MAGIC, 8) == 0) {
        uint32_t count = 0;
        std::memcpy(&count, base + need + 8, 4);
        if (count > 64 || m.len < need + 16 + (size_t)count * 2 * h.N * 8) return bce_set_error(c, BCE_ERR_ARG, "key file: truncated transform-check trailer");
        const char* pairs = base + need + 16;     // byte offsets: the trailer need not be 8-byte aligned in the file
        const size_t poly = (size_t)h.N * 8;
        std::vector<uint64_t> t(h.N);
        for (uint32_t k = 0; k < count && h.bsk_format == BCE_KEYFILE_BSK_EVALUATION; ++k) {
            std::memcpy(t.data(), pairs + (size_t)k * 2 * poly, poly);
            const int rc = bce_debug_ntt(c, t.data(), 1, 0);
            if (rc) return rc;
            if (std::memcmp(t.data(), pairs + ((size_t)k * 2 + 1) * poly, poly) != 0)
                return bce_set_error(c, BCE_ERR_UNSUPPORTED, "key file: the producer's EVALUATION representation is not this engine's (transform-check pair differs): "
                                                              "export the bootstrapping key in coefficient form (bsk_format 0)");
        }
    }
    return h.bsk_format == BCE_KEYFILE_BSK_EVALUATION ? bce_import_keys_eval(c, s, z, bsk, h.bsk_words, ksk, h.ksk_words)
                                                      : bce_import_keys(c, s, z, bsk, h.bsk_words, ksk, h.ksk_words);
}

int bce_export_keys_file(bce_ctx* c, const char* path) {
    if (!c || !path) return BCE_ERR_ARG;
    uint64_t p[BCE_P_COUNT];
    bce_get_params(c, p);
    bce_keyfile_header h{};
    std::memcpy(h.magic, BCE_KEYFILE_MAGIC, 8);
    h.version = BCE_KEYFILE_VERSION;
    h.method = (uint32_t)p[BCE_P_method];
    h.n = p[BCE_P_n]; h.N = p[BCE_P_N]; h.q = p[BCE_P_q]; h.Q = p[BCE_P_Q]; h.qKS = p[BCE_P_qKS];
    h.baseKS = p[BCE_P_baseKS]; h.baseG = p[BCE_P_baseG]; h.baseR = p[BCE_P_baseR];
    h.bsk_words = bce_bsk_words(c);
    h.ksk_words = bce_ksk_words(c);
    // z = N entries of 2 marks "not exported": contexts whose keys were imported without the ring secret have none
    std::vector<int32_t> s(h.n), z(h.N, 2);
    int rc = bce_export_sk(c, s.data(), z.data());
    if (rc) return bce_set_error(c, rc, "no keys to export");
    h.has_z = (h.N > 0 && z[0] != 2) ? 1 : 0;
    std::vector<uint64_t> bsk(h.bsk_words);
    if ((rc = bce_export_bsk(c, bsk.data()))) return rc;
    std::vector<uint32_t> ksk(h.ksk_words);
    if ((rc = bce_export_ksk(c, ksk.data()))) return rc;
    // the file holds the secret keys: owner-only permissions whatever the umask says
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0600);
    FILE* f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
    if (!f) { if (fd >= 0) close(fd); return bce_set_error(c, BCE_ERR_ARG, (std::string("cannot open ") + path + " for writing").c_str()); }
    if (fchmod(fd, 0600) != 0) { std::fclose(f); return bce_set_error(c, BCE_ERR_STATE, "cannot restrict the key file's permissions"); }
    const uint64_t zn = h.has_z ? h.N : 0;
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    ok = ok && std::fwrite(s.data(), 4, h.n, f) == h.n && std::fwrite(z.data(), 4, zn, f) == zn;
    if (ok && ((h.n + zn) & 1)) { const uint32_t zero = 0; ok = std::fwrite(&zero, 4, 1, f) == 1; }
    ok = ok && std::fwrite(bsk.data(), 8, h.bsk_words, f) == h.bsk_words && std::fwrite(ksk.data(), 4, h.ksk_words, f) == h.ksk_words;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? BCE_OK : bce_set_error(c, BCE_ERR_STATE, "short write to the key file");
}

}  // extern "C"
