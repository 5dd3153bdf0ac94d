// circuit_capi.cpp -- extern "C" wrappers of include/bce_circuit.h over bce::Circuit.
// C++ exceptions stop here and become bce_status codes + bce_circuit_last_error().
#include <algorithm>
#include <cstring>
#include <exception>
#include <new>
#include <stdexcept>
#include <string>

#include "../../include/bce_circuit.h"
#include "bristol.hpp"
#include "circuit.hpp"

struct bce_circuit {
    bce::Circuit c;
    std::string err;
    explicit bce_circuit(bce_ctx* e) : c(e) {}
};

namespace {
template <typename F>
int guarded(bce_circuit* h, F fn) {
    if (!h) return BCE_ERR_ARG;
    try {
        fn();
        return BCE_OK;
    } catch (const std::invalid_argument& e) {
        h->err = e.what();
        return BCE_ERR_ARG;
    } catch (const std::out_of_range& e) {
        h->err = e.what();
        return BCE_ERR_ARG;
    } catch (const std::logic_error& e) {
        h->err = e.what();
        return BCE_ERR_STATE;
    } catch (const std::exception& e) {
        h->err = e.what();
        return BCE_ERR_STATE;
    }
}
}  // namespace

extern "C" {

int bce_circuit_create(bce_ctx* engine, bce_circuit** out) {
    if (!out) return BCE_ERR_ARG;
    *out = new (std::nothrow) bce_circuit(engine);
    return *out ? BCE_OK : BCE_ERR_STATE;
}
void bce_circuit_destroy(bce_circuit* h) { delete h; }
const char* bce_circuit_last_error(const bce_circuit* h) { return h ? h->err.c_str() : "null circuit"; }

int bce_circuit_read_file(bce_circuit* h, const char* path) {
    return guarded(h, [&] { if (!path) throw std::invalid_argument("null path"); h->c.ReadFile(path); });
}
int bce_circuit_read_bristol(bce_circuit* h, const char* path, int new_flag) {
    return guarded(h, [&] { if (!path) throw std::invalid_argument("null path"); h->c.ReadBristol(path, new_flag != 0); });
}
int bce_circuit_get_info(const bce_circuit* h, bce_circuit_info* out) {
    if (!h || !out) return BCE_ERR_ARG;
    *out = h->c.info();
    return BCE_OK;
}
int bce_circuit_reset(bce_circuit* h) { return guarded(h, [&] { h->c.Reset(); }); }
int bce_circuit_rearm(bce_circuit* h) { return guarded(h, [&] { h->c.Rearm(); }); }
int bce_circuit_set_plaintext(bce_circuit* h, int on) { return guarded(h, [&] { h->c.setPlaintext(on != 0); }); }
int bce_circuit_set_encrypted(bce_circuit* h, int on) { return guarded(h, [&] { h->c.setEncrypted(on != 0); }); }
int bce_circuit_set_verify(bce_circuit* h, int on) { return guarded(h, [&] { h->c.setVerify(on != 0); }); }
int bce_circuit_get_flags(const bce_circuit* h, int* p, int* e, int* v) {
    if (!h) return BCE_ERR_ARG;
    if (p) *p = h->c.getPlaintext();
    if (e) *e = h->c.getEncrypted();
    if (v) *v = h->c.getVerify();
    return BCE_OK;
}
int bce_circuit_set_batched(bce_circuit* h, int on) { return guarded(h, [&] { h->c.setBatched(on != 0); }); }
int bce_circuit_set_encrypt_mode(bce_circuit* h, int mode) {
    return guarded(h, [&] {
        if (mode != BCE_FRESH && mode != BCE_BOOTSTRAPPED) throw std::invalid_argument("bad encrypt mode");
        h->c.setEncryptMode(mode);
    });
}
int bce_circuit_set_xor_fast(bce_circuit* h, int on) { return guarded(h, [&] { h->c.setXorFast(on != 0); }); }
int bce_circuit_set_relevel(bce_circuit* h, int on) { return guarded(h, [&] { h->c.setRelevel(on != 0); }); }
int bce_circuit_get_relevel(bce_circuit* h) { return h && h->c.getRelevel() ? 1 : 0; }
int bce_circuit_set_instances(bce_circuit* h, uint32_t k) { return guarded(h, [&] { h->c.setInstances(k); }); }
int bce_circuit_set_shard_locality(bce_circuit* h, int on) { return guarded(h, [&] { h->c.setShardLocality(on != 0); }); }
uint64_t bce_circuit_plan_hash(const bce_circuit* h) { return h ? h->c.planHash() : 0; }
int bce_circuit_get_encrypt_mode(const bce_circuit* h) { return h ? h->c.getEncryptMode() : -1; }
int bce_circuit_set_dataflow(bce_circuit* h, int on) { return guarded(h, [&] { h->c.setDataflow(on != 0); }); }
int bce_circuit_dataflow_active(const bce_circuit* h) { return h && h->c.dataflowActive() ? 1 : 0; }
int bce_circuit_set_graph(bce_circuit* h, int on) { return guarded(h, [&] { h->c.setGraph(on != 0); }); }
int bce_circuit_graph_active(const bce_circuit* h) { return h && h->c.graphActive() ? 1 : 0; }
int bce_circuit_dataflow_plan(const bce_circuit* h, bce_gate_desc* tasks, uint8_t* prio, uint32_t cap, uint32_t* n_tasks) {
    if (!h || !n_tasks) return BCE_ERR_ARG;
    const auto& t = h->c.dataflowTasks();
    const auto& p = h->c.dataflowPriorities();
    const uint32_t n = (uint32_t)std::min<size_t>(t.size(), cap);
    if (tasks) std::copy(t.begin(), t.begin() + n, tasks);
    if (prio) std::copy(p.begin(), p.begin() + n, prio);
    *n_tasks = (uint32_t)t.size();
    return BCE_OK;
}
int bce_circuit_set_balance(bce_circuit* h, int on, uint32_t lone, uint32_t full) { return guarded(h, [&] { h->c.setBalance(on != 0, lone, full); }); }
int bce_circuit_relevel_steps(const bce_circuit* h, uint32_t* sizes, uint32_t cap, uint32_t* n_steps) {
    if (!h || !n_steps || (cap && !sizes)) return BCE_ERR_ARG;
    const std::vector<uint32_t> v = h->c.relevelStepSizes();
    for (size_t i = 0; i < v.size() && i < cap; ++i) sizes[i] = v[i];
    *n_steps = (uint32_t)v.size();
    return BCE_OK;
}
int bce_circuit_relevel_publications(const bce_circuit* h, uint32_t* counts, uint32_t cap, uint32_t* n_steps) {
    if (!h || !n_steps || (cap && !counts)) return BCE_ERR_ARG;
    const std::vector<uint32_t> v = h->c.relevelPublications();
    for (size_t i = 0; i < v.size() && i < cap; ++i) counts[i] = v[i];
    *n_steps = (uint32_t)v.size();
    return BCE_OK;
}
int bce_circuit_check_relevel(bce_circuit* h) {
    return guarded(h, [&] {
        std::string why;
        if (!h->c.checkRelevelPlan(&why)) throw std::logic_error("bootstrap-depth schedule: " + why);
    });
}

int bce_circuit_set_input(bce_circuit* h, uint32_t instance, const uint32_t* widths, uint32_t n_buses, const uint8_t* bits) {
    return guarded(h, [&] {
        if (!widths || !bits) throw std::invalid_argument("null input");
        bce::Inputs in(n_buses);
        size_t off = 0;
        for (uint32_t k = 0; k < n_buses; ++k) {
            in[k].assign(bits + off, bits + off + widths[k]);
            off += widths[k];
        }
        h->c.SetInput(instance, in, false);
    });
}
int bce_circuit_clock(bce_circuit* h) { return guarded(h, [&] { h->c.Clock(); }); }
int bce_circuit_get_output(const bce_circuit* h, uint32_t instance, uint8_t* bits) {
    if (!h || !bits) return BCE_ERR_ARG;
    if (instance >= h->c.getInstances()) return BCE_ERR_ARG;
    bce::Outputs o = h->c.getOutputs(instance);   // all output values, concatenated in header order
    size_t pos = 0;
    for (const auto& bus : o)
        for (unsigned b : bus) bits[pos++] = (uint8_t)b;
    return BCE_OK;
}
int bce_circuit_get_buses(const bce_circuit* h, uint32_t* n_in, uint32_t* in_widths, uint32_t in_cap, uint32_t* n_out,
                          uint32_t* out_widths, uint32_t out_cap) {
    if (!h || !n_in || !n_out) return BCE_ERR_ARG;
    const auto& iw = h->c.inputBusBits();
    const auto& ow = h->c.outputBusBits();
    *n_in = (uint32_t)iw.size();
    *n_out = (uint32_t)ow.size();
    for (uint32_t k = 0; in_widths && k < in_cap && k < iw.size(); ++k) in_widths[k] = iw[k];
    for (uint32_t k = 0; out_widths && k < out_cap && k < ow.size(); ++k) out_widths[k] = ow[k];
    return BCE_OK;
}
int bce_circuit_get_counts(const bce_circuit* h, uint32_t out[6]) {
    if (!h || !out) return BCE_ERR_ARG;
    h->c.getCounts(out);
    return BCE_OK;
}
int bce_circuit_get_stats(const bce_circuit* h, bce_circuit_stats* out) {
    if (!h || !out) return BCE_ERR_ARG;
    *out = h->c.stats();
    return BCE_OK;
}
int bce_circuit_dump(const bce_circuit* h, int what) {
    if (!h) return BCE_ERR_ARG;
    if (what == 0) h->c.dumpNetList(); else if (what == 1) h->c.dumpGates(); else h->c.dumpGateCount();
    return BCE_OK;
}

int bce_circuit_set_exchange(bce_circuit* h, uint32_t rank, uint32_t world, int shard_mode, bce_allgather_fn fn, void* user,
                             void* host_send, void* host_recv, void* dev_send, void* dev_recv, uint64_t capacity) {
    return guarded(h, [&] { h->c.setExchange(rank, world, shard_mode, fn, user, host_send, host_recv, dev_send, dev_recv, capacity); });
}
int bce_circuit_enable_rccl(bce_circuit* h, int on) { return guarded(h, [&] { h->c.enableRccl(on != 0); }); }
uint64_t bce_circuit_exchange_capacity(const bce_circuit* h, uint32_t world, int shard_mode, int encrypted) {
    return h ? h->c.exchangeCapacity(world, shard_mode, encrypted != 0) : 0;
}

int bce_assemble_bristol(const char* in_path, int new_flag, int gen_fan_flag, int debug_flag, const char* out_path, char* err,
                         uint32_t err_len) {
    try {
        if (!in_path) throw std::invalid_argument("null path");
        bce::Analysis a = bce::analyze_bristol(in_path, gen_fan_flag != 0, new_flag != 0, true);
        bce::assemble_bristol(a, 0, debug_flag != 0, out_path ? out_path : "", true);
        return BCE_OK;
    } catch (const std::exception& e) {
        if (err && err_len) { std::strncpy(err, e.what(), err_len - 1); err[err_len - 1] = 0; }
        return BCE_ERR_ARG;
    }
}

}  // extern "C"
