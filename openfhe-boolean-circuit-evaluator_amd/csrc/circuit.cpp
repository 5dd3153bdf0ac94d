// circuit.cpp -- host DAG walker (see circuit.hpp).  Behavioural contract: SURVEY.md App. F.
#include "circuit.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <queue>
#include <sstream>
#include <stdexcept>

#include "bristol.hpp"

namespace bce {

namespace {
using Clock_t = std::chrono::steady_clock;
double ms_since(Clock_t::time_point t0) { return std::chrono::duration<double, std::milli>(Clock_t::now() - t0).count(); }
bool contains(const std::string& s, const char* sub) { return s.find(sub) != std::string::npos; }
const char* op_name(GateEnum op) {
    switch (op) {
        case GateEnum::INPUT: return "INPUT";
        case GateEnum::OUTPUT: return "OUTPUT";
        case GateEnum::NOT: return "NOT";
        case GateEnum::AND: return "AND";
        case GateEnum::OR: return "OR";
        case GateEnum::XOR: return "XOR";
        default: return "?";
    }
}
uint32_t gate_weight(GateEnum op, bool xor_fast = false) {  // bootstraps per gate (src/gate.cpp:133,172,200-202)
    switch (op) {
        case GateEnum::AND: case GateEnum::OR: return 1;
        case GateEnum::XOR: return xor_fast ? 1 : 3;
        default: return 0;
    }
}
}  // namespace

// ---- Wire -------------------------------------------------------------------------------
void Wire::updateFanoutGates(const std::string& gateToRemove) {
    auto it = std::find(fanoutGates.begin(), fanoutGates.end(), gateToRemove);
    if (it == fanoutGates.end()) {
        std::cerr << "error can't find " << gateToRemove << " in fanout of wire " << name << std::endl;
        return;
    }
    fanoutGates.erase(it);
}

// ---- Gate -------------------------------------------------------------------------------
static void gate_ck(const GateEvalParams& gep, int rc, const std::string& name) {
    if (rc != BCE_OK) throw std::runtime_error("gate " + name + ": " + bce_last_error(gep.cc));
}

// verify-and-fix (src/gate.cpp:113-120,153-160,174-181,206-213)
static void verify_fix(const GateEvalParams& gep, const char* opn, CipherText slot, unsigned expect, bool fix,
                       const std::string& name) {
    uint8_t res = 0;
    gate_ck(gep, bce_decrypt_bits(gep.cc, &slot, 1, &res), name);
    if (res != expect) {
        std::cerr << "Bad " << opn << " fixing" << std::endl;
        if (gep.fixes) ++*gep.fixes;
        if (fix) {
            uint8_t bit = (uint8_t)expect;
            uint64_t idx = gep.enc_counter ? (*gep.enc_counter)++ : 0;
            gate_ck(gep, bce_encrypt_bits(gep.cc, &bit, &slot, 1, idx, gep.encrypt_mode), name);
        }
    }
}

void Gate::Evaluate(const GateEvalParams& gep) {
    bool all_ready = true;
    for (bool r : ready) all_ready = all_ready && r;
    if (!all_ready) std::cerr << "error, executing gate " << name << " but inputs not ready!" << std::endl;
    const bool pt = gep.plaintext_flag, en = gep.encrypted_flag, vf = gep.verify_flag;
    if (en && !gep.cc) throw std::runtime_error("gate " + name + ": encrypted evaluation needs an engine");
    auto enc_dst = [&]() -> CipherText {
        if (encout.empty() || encout[0] == kNoCipherText) throw std::runtime_error("gate " + name + ": no destination slot");
        return encout[0];
    };
    switch (op) {
        case GateEnum::INPUT:
            std::cerr << "error executing input should not happen" << std::endl;
            break;
        case GateEnum::OUTPUT:
            if (pt) { plainout.resize(1); plainout[0] = plainin[0]; }
            if (en) {
                encout.resize(1);
                encout[0] = encin[0];  // copy of the handle (src/gate.cpp:90-94)
                if (vf) verify_fix(gep, "OUTPUT", encout[0], plainout[0], false, name);
            }
            break;
        case GateEnum::NOT:
            if (pt) { plainout.resize(1); plainout[0] = !plainin[0]; }
            if (en) {
                bce_gate_desc d{BCE_OP_NOT, encin[0], encin[0], enc_dst(), 0, 0};
                gate_ck(gep, bce_eval_gates(gep.cc, 1, &d), name);
                if (vf) verify_fix(gep, "NOT", encout[0], plainout[0], true, name);
            }
            break;
        case GateEnum::AND:
        case GateEnum::OR:
            if (pt) {
                plainout.resize(1);
                plainout[0] = (op == GateEnum::AND) ? (plainin[0] && plainin[1]) : (plainin[0] || plainin[1]);
            }
            if (en) {
                // The reference retries AND after a throw because OpenFHE rejects ct1 == ct2
                // (src/gate.cpp:131-152); the engine computes on values, so equal handles are legal.
                bce_gate_desc d{(uint32_t)(op == GateEnum::AND ? BCE_AND : BCE_OR), encin[0], encin[1], enc_dst(), 0, 0};
                gate_ck(gep, bce_eval_gates(gep.cc, 1, &d), name);
                if (vf) verify_fix(gep, op == GateEnum::AND ? "AND" : "OR", encout[0], plainout[0], true, name);
            }
            break;
        case GateEnum::XOR:
            if (pt) { plainout.resize(1); plainout[0] = plainin[0] ^ plainin[1]; }
            if (en && gep.xor_fast) {
                bce_gate_desc d{BCE_XOR_FAST, encin[0], encin[1], enc_dst(), 0, 0};
                gate_ck(gep, bce_eval_gates(gep.cc, 1, &d), name);
                if (vf) verify_fix(gep, "XOR", encout[0], plainout[0], true, name);
            } else if (en) {
                // (a AND !b) OR (!a AND b), src/gate.cpp:198-202; the NOTs are folded into the prep
                if (tmp.size() < 2) throw std::runtime_error("gate " + name + ": XOR needs two scratch slots");
                bce_gate_desc a[2] = {{BCE_AND, encin[0], encin[1], tmp[0], 0, 1}, {BCE_AND, encin[0], encin[1], tmp[1], 1, 0}};
                gate_ck(gep, bce_eval_gates(gep.cc, 2, a), name);
                bce_gate_desc o{BCE_OR, tmp[0], tmp[1], enc_dst(), 0, 0};
                gate_ck(gep, bce_eval_gates(gep.cc, 1, &o), name);
                if (vf) verify_fix(gep, "XOR", encout[0], plainout[0], true, name);
            }
            break;
        case GateEnum::DFF: std::cerr << "remember to write DFF" << std::endl; break;
        case GateEnum::LUT3: std::cerr << "remember to write LUT3" << std::endl; break;
        case GateEnum::LUT4: std::cerr << "remember to write LUT4" << std::endl; break;
        default: std::cerr << "bad gate eval" << std::endl;
    }
}

// ---- Circuit ------------------------------------------------------------------------------
Circuit::Circuit(int set, int method) {
    std::cout << "Generating crypto context" << std::endl;
    if (set == BCE_TOY) {
        std::cout << "*************************\nWARNING TOY Security used\n*************************" << std::endl;
    } else if (set == BCE_STD128_OPT) {
        std::cout << "STD 128 Optimized Security used" << std::endl;
    } else {
        throw std::invalid_argument("Error Bad security");  // the reference exits here (src/circuit.cpp:75-78)
    }
    if (method == BCE_AP) std::cout << "AP used" << std::endl;
    else if (method == BCE_GINX) std::cout << "GINX used" << std::endl;
    else throw std::invalid_argument("Error Bad method");
    int rc = bce_ctx_create(set, method, 0, &cc);
    if (rc != BCE_OK) throw std::runtime_error(std::string("GenerateBinFHEContext: ") + bce_last_error(nullptr));
    owns_engine_ = true;
    std::cout << "Generating crypto keys" << std::endl;
    rc = bce_keygen(cc, nullptr);  // cc.KeyGen() + BTKeyGen (src/circuit.cpp:90-91): seed from OS entropy
    if (rc != BCE_OK) {
        std::string m = bce_last_error(cc);
        bce_ctx_destroy(cc);
        throw std::runtime_error("BTKeyGen: " + m);
    }
    std::cout << "Done" << std::endl;
    gep.cc = cc;
    gep.enc_counter = &enc_counter_;
    gep.fixes = &stats_.verify_fixes;
}

Circuit::Circuit(bce_ctx* engine) : cc(engine) {
    gep.cc = cc;
    gep.enc_counter = &enc_counter_;
    gep.fixes = &stats_.verify_fixes;
    quiet_ = true;
}

Circuit::~Circuit() {
    dropDag();
    dropPlan();
    if (owns_engine_ && cc) bce_ctx_destroy(cc);
}

void Circuit::requireEngine(const char* what) const {
    if (!cc) throw std::runtime_error(std::string(what) + ": this circuit has no engine (plaintext-only); encrypted mode needs the HIP engine");
}
void Circuit::ck(int rc, const char* what) const {
    if (rc != BCE_OK) throw std::runtime_error(std::string(what) + ": " + bce_last_error(cc));
}

int Circuit::addWire(uint32_t reg) {
    auto it = wire_of_reg_.find(reg);
    if (it != wire_of_reg_.end()) return it->second;
    int id = (int)wire_names_.size();
    wire_of_reg_[reg] = id;
    wire_names_.push_back("R:" + std::to_string(reg));
    return id;
}

int Circuit::wireOf(uint32_t reg, const char* what, unsigned lineNo) const {
    auto it = wire_of_reg_.find(reg);
    if (it == wire_of_reg_.end())
        throw std::runtime_error(std::string(what) + " parse error line " + std::to_string(lineNo) + ": register R" + std::to_string(reg) + " used before it is defined");
    return it->second;
}

bool Circuit::ReadFile(const std::string& inFname) {
    if (!quiet_) std::cout << "Loading circuit description " << inFname << std::endl;
    std::ifstream in(inFname);
    if (!in) throw std::runtime_error("error opening file " + inFname);
    inputGates.clear(); constWires_.clear(); allGates.clear(); wire_names_.clear(); wire_of_reg_.clear();
    unsigned lineNo = 0, gateNo = 0, max_out = 0;
    bool any_out = false;
    std::string t;
    n_buses_ = 0; n_in_bits_.assign(2, 0); out_bus_bits_.clear();
    while (std::getline(in, t)) {
        ++lineNo;
        if (!quiet_ && lineNo % 100 == 0) std::cout << "\r loading line " << lineNo << std::flush;
        if (!t.empty() && t[0] == '#') {
            // "# output buses w1 w2 ...": written by this assembler for circuits with several output values
            if (t.rfind("# output buses", 0) == 0) {
                std::istringstream ob(t.substr(14));
                unsigned w;
                out_bus_bits_.clear();
                while (ob >> w) out_bus_bits_.push_back(w);
            }
            continue;
        }
        unsigned n1 = 0, n2 = 0, n3 = 0;
        auto two_in = [&](const char* fmt, const char* what, GateEnum op) {
            if (std::sscanf(t.c_str(), fmt, &n1, &n2, &n3) != 3) throw std::runtime_error(std::string(what) + " parse error line " + std::to_string(lineNo));
            GateRec g{op, 2, {wireOf(n2, what, lineNo), wireOf(n3, what, lineNo)}, -1, -1, std::string(what) + ":" + std::to_string(gateNo++)};
            g.out = addWire(n1);
            allGates.push_back(g);
        };
        // dispatch order of the reference reader (src/circuit.cpp:144,171,199,223,246,270,292)
        if (contains(t, "CONST")) {
            // extension of the text format (Bristol Fashion EQ): a register holding a public constant
            if (std::sscanf(t.c_str(), "R%u = CONST(%u)", &n1, &n2) != 2 || n2 > 1)
                throw std::runtime_error("CONST parse error line " + std::to_string(lineNo));
            constWires_.push_back({addWire(n1), n2});
        } else if (contains(t, "LOAD")) {
            if (std::sscanf(t.c_str(), "R%u = LOAD(In%u, %u)", &n1, &n2, &n3) != 3 || n2 < 1 || n2 > 64)
                throw std::runtime_error("LOAD parse error line " + std::to_string(lineNo));
            LoadRec l{n2 - 1, n3, addWire(n1), "INPUT:" + std::to_string(gateNo++)};
            n_buses_ = std::max(n_buses_, n2);
            if (n_in_bits_.size() < n2) n_in_bits_.resize(n2, 0);
            n_in_bits_[n2 - 1] = std::max(n_in_bits_[n2 - 1], n3 + 1);
            inputGates.push_back(l);
        } else if (contains(t, "STORE")) {
            if (std::sscanf(t.c_str(), "Out%u = STORE(R%u)", &n1, &n2) != 2) throw std::runtime_error("STORE parse error line " + std::to_string(lineNo));
            GateRec g{GateEnum::OUTPUT, 1, {wireOf(n2, "STORE", lineNo), -1}, -1, (int)n1, "OUTPUT:" + std::to_string(gateNo++)};
            allGates.push_back(g);
            max_out = std::max(max_out, n1);
            any_out = true;
        } else if (contains(t, "NOT")) {
            if (std::sscanf(t.c_str(), "R%u = NOT(R%u)", &n1, &n2) != 2) throw std::runtime_error("NOT parse error line " + std::to_string(lineNo));
            GateRec g{GateEnum::NOT, 1, {wireOf(n2, "NOT", lineNo), -1}, -1, -1, "NOT:" + std::to_string(gateNo++)};
            g.out = addWire(n1);
            allGates.push_back(g);
        } else if (contains(t, "AND")) {
            two_in("R%u = AND(R%u, R%u)", "AND", GateEnum::AND);
        } else if (contains(t, " OR")) {
            two_in("R%u = OR(R%u, R%u)", "OR", GateEnum::OR);
        } else if (contains(t, "XOR")) {
            two_in("R%u = XOR(R%u, R%u)", "XOR", GateEnum::XOR);
        } else if (contains(t, "BOOT")) {
            // no-op
        }
    }
    n_outputs = 1;
    n_output_bits.assign(1, any_out ? max_out + 1 : 0);
    {
        unsigned tot = 0;
        for (unsigned w : out_bus_bits_) tot += w;
        if (out_bus_bits_.empty() || tot != n_output_bits[0]) out_bus_bits_.assign(1, n_output_bits[0]);
    }
    if (!quiet_) {
        std::cout << std::endl << "generating output nbits " << n_output_bits[0] << std::endl;
        std::cout << "generating netlist" << std::endl;
    }
    finalizeNetlist();
    if (!quiet_) std::cout << "Done" << std::endl;
    return true;
}

bool Circuit::ReadBristol(const std::string& path, bool new_flag) {
    Analysis A = analyze_bristol(path, false, new_flag, true);
    const Variable& v = A.variables;
    const Function& f = A.functions;
    inputGates.clear(); constWires_.clear(); allGates.clear(); wire_names_.clear(); wire_of_reg_.clear();
    unsigned gateNo = 0;
    // same register numbering as the assembler: inputs first (bus after bus), then one register per gate
    std::vector<int> node_wire(v.n_tot, -1);
    uint32_t reg = 0;
    {
        unsigned node = 0;
        for (size_t bus = 0; bus < v.in_bits.size(); ++bus)
            for (unsigned b = 0; b < v.in_bits[bus]; ++b, ++node) {
                node_wire[node] = addWire(reg++);
                inputGates.push_back({(unsigned)bus, b, node_wire[node], "INPUT:" + std::to_string(gateNo++)});
            }
    }
    n_in_bits_ = v.in_bits;
    while (n_in_bits_.size() > 1 && n_in_bits_.back() == 0) n_in_bits_.pop_back();   // old format: "n 0 m" = one input
    n_buses_ = (unsigned)n_in_bits_.size();
    if (n_in_bits_.size() < 2) n_in_bits_.resize(2, 0);
    out_bus_bits_ = v.out_bits;
    for (size_t i = 0; i < f.call_list.size(); ++i) {
        const std::string& op = f.call_list[i];
        GateRec g{};
        g.in[0] = g.in[1] = -1; g.out_bit = -1;
        const auto& il = f.in_list[i];
        if (op == " EQ") {
            // constant: a register that is live from the start; encrypted runs hold it as the trivial (noiseless)
            // ciphertext (0, value * q/4) -- a public constant needs no key
            const int w = addWire(reg++);
            node_wire[f.out_list[i].at(0)] = w;
            constWires_.push_back({w, il.at(0)});
            continue;
        }
        for (unsigned w : il) if (node_wire[w] < 0) throw std::runtime_error("ReadBristol: gate " + std::to_string(i) + " uses undefined wire");
        if (op == "XOR" || op == "AND") {
            if (il.size() != 2) throw std::runtime_error("ReadBristol: bad arity");
            g.op = op == "XOR" ? GateEnum::XOR : GateEnum::AND; g.nin = 2; g.in[0] = node_wire[il[0]]; g.in[1] = node_wire[il[1]];
        } else if (op == "NOT") {
            g.op = GateEnum::NOT; g.nin = 1; g.in[0] = node_wire[il.at(0)];
        } else if (op == "EQW") {
            // Bristol Fashion wire copy: no gate, the output node is an alias of the input wire
            // (the reference's assembler only emits a parse-error comment for it, src/assemble.cpp:370-373)
            node_wire[f.out_list[i].at(0)] = node_wire[il.at(0)];
            continue;
        } else {
            throw std::runtime_error("ReadBristol: unsupported op " + op + " at gate " + std::to_string(i));
        }
        g.name = std::string(op_name(g.op)) + ":" + std::to_string(gateNo++);
        g.out = addWire(reg++);
        node_wire[f.out_list[i].at(0)] = g.out;
        allGates.push_back(g);
    }
    for (unsigned o = 0; o < v.n_out1_bits; ++o) {
        int w = node_wire[v.n_tot - v.n_out1_bits + o];
        if (w < 0) throw std::runtime_error("ReadBristol: output node never driven");
        allGates.push_back(GateRec{GateEnum::OUTPUT, 1, {w, -1}, -1, (int)o, "OUTPUT:" + std::to_string(gateNo++)});
    }
    n_outputs = 1;   // internally ONE concatenated bus (bit indices run over all output values); getOutputs() splits it
    n_output_bits.assign(1, v.n_out1_bits);
    finalizeNetlist();
    return true;
}

// CSR fan-out + static ASAP levelisation (what _CircuitManager discovers round by round)
void Circuit::finalizeNetlist() {
    const size_t W = wire_names_.size(), G = allGates.size();
    fan_off_.assign(W + 1, 0);
    for (const auto& g : allGates) for (int k = 0; k < g.nin; ++k) ++fan_off_[g.in[k] + 1];
    for (size_t w = 0; w < W; ++w) fan_off_[w + 1] += fan_off_[w];
    fan_gate_.assign(fan_off_[W], 0);
    std::vector<uint32_t> pos(fan_off_.begin(), fan_off_.end() - 1);
    for (size_t gi = 0; gi < G; ++gi) for (int k = 0; k < allGates[gi].nin; ++k) fan_gate_[pos[allGates[gi].in[k]]++] = (uint32_t)gi;

    levels_.clear();
    gate_level_.assign(G, -1);
    std::vector<int> ready(G, 0), active;
    for (const auto& l : inputGates) active.push_back(l.wire);
    for (const auto& k : constWires_) active.push_back(k.wire);
    max_level_xor_ = 0;
    while (!active.empty()) {
        Level L;
        for (int w : active)
            for (uint32_t e = fan_off_[w]; e < fan_off_[w + 1]; ++e) {
                uint32_t gi = fan_gate_[e];
                if (++ready[gi] == allGates[gi].nin) L.gates.push_back((int)gi);
            }
        if (L.gates.empty()) break;
        std::sort(L.gates.begin(), L.gates.end());
        active.clear();
        for (int gi : L.gates) {
            gate_level_[gi] = (int)levels_.size();
            if (allGates[gi].op == GateEnum::XOR) ++L.n_xor;
            if (allGates[gi].out >= 0) active.push_back(allGates[gi].out);
        }
        max_level_xor_ = std::max(max_level_xor_, L.n_xor);
        levels_.push_back(std::move(L));
    }
    base_stride_ = stride_ = (uint32_t)W + 2 * max_level_xor_;
    inputs_set_ = false;
    rebuildRelevel();  // also sizes the scratch slots the re-levelled schedule needs
    buildShardPlan();
    Reset();
}

void Circuit::Reset() {
    n_input_gates = n_output_gates = n_and_gates = n_or_gates = n_xor_gates = n_not_gates = 0;
    plaintext_flag = encrypted_flag = verify_flag = false;  // gep's copies are left alone, like the reference
    done = false;
    inputs_set_ = false;
    ++epoch_;
    plain_.assign(instances_, std::vector<uint8_t>(wire_names_.size(), 0));
    circuitOut.assign(instances_, std::vector<uint8_t>(n_output_bits.empty() ? 0 : n_output_bits[0], 0));
    stats_ = bce_circuit_stats{};
}

void Circuit::Rearm() {
    if (!inputs_set_) throw std::logic_error("Rearm: SetInput has not been called");
    n_output_gates = n_and_gates = n_or_gates = n_xor_gates = n_not_gates = 0;
    done = false;
    stats_ = bce_circuit_stats{};
}

void Circuit::setVerify(bool b) {
    verify_flag = gep.verify_flag = b;
    if (b) { setPlaintext(true); setEncrypted(true); }
}

void Circuit::setInstances(unsigned k) {
    if (k == 0) throw std::invalid_argument("setInstances: need at least one instance");
    instances_ = k;
    plain_.assign(instances_, std::vector<uint8_t>(wire_names_.size(), 0));
    circuitOut.assign(instances_, std::vector<uint8_t>(n_output_bits.empty() ? 0 : n_output_bits[0], 0));
    buildShardPlan();
    rebuildRelevel();   // the balanced schedule depends on K
}

void Circuit::instanceRange(unsigned& lo, unsigned& hi) const {
    lo = 0; hi = instances_;
    if (world_ > 1 && shard_mode_ == 0) {
        unsigned per = instances_ / world_;
        lo = rank_ * per; hi = lo + per;
    }
}

void Circuit::SetInput(const Inputs& input, bool verbose) { SetInput(0, input, verbose); }

void Circuit::SetInput(unsigned inst, const Inputs& input, bool verbose) {
    if (inst >= instances_) throw std::out_of_range("SetInput: instance out of range");
    size_t total_bits = 0;
    for (size_t k = 0; k < input.size(); ++k) {
        if (verbose) std::cout << "setting input " << k << " size " << input[k].size() << std::endl;
        total_bits += input[k].size();
    }
    if (verbose) std::cout << "set input total of " << input.size() << " inputs" << std::endl;
    unsigned lo, hi;
    instanceRange(lo, hi);
    const bool mine = inst >= lo && inst < hi;
    std::vector<uint8_t> bits;
    std::vector<uint32_t> slots;
    n_input_gates = 0;
    for (const auto& l : inputGates) {
        if (l.bus >= input.size() || l.bit >= input[l.bus].size())
            throw std::out_of_range("SetInput: " + l.name + " reads In" + std::to_string(l.bus + 1) + " bit " + std::to_string(l.bit) + " which was not supplied");
        uint8_t v = input[l.bus][l.bit] ? 1 : 0;
        ++n_input_gates;
        plain_[inst][l.wire] = v;
        bits.push_back(v);
        slots.push_back(inst * stride_ + (uint32_t)l.wire);
    }
    for (const auto& k : constWires_) plain_[inst][k.wire] = (uint8_t)k.value;
    if (total_bits != inputGates.size())
        std::cerr << "error: total_inputs: " << total_bits << " #used: " << inputGates.size() << std::endl;
    else if (verbose)
        std::cout << "input confirmed" << std::endl;
    if (encrypted_flag && mine) {  // encrypted mode must be on at call time (src/circuit.cpp:505-507)
        requireEngine("SetInput");
        ck(bce_pool_reserve(cc, instances_ * stride_), "SetInput(pool)");
        // stream index = (evaluation epoch, instance, input position): every rank draws the same ciphertexts
        uint64_t base = ((epoch_ & 0xFFFFFull) << 44) | ((uint64_t)inst << 24);
        ck(bce_encrypt_bits(cc, bits.data(), slots.data(), (uint32_t)slots.size(), base, encrypt_mode_), "SetInput(Encrypt)");
        if (!constWires_.empty()) {
            // public constants: trivial ciphertexts (a = 0, b = value * q/4), exact and noiseless
            uint64_t p[BCE_P_COUNT];
            ck(bce_get_params(cc, p), "SetInput(params)");
            const size_t W = (size_t)p[BCE_P_n] + 1;
            std::vector<uint64_t> cts(constWires_.size() * W, 0);
            std::vector<uint32_t> cslots;
            for (size_t k = 0; k < constWires_.size(); ++k) {
                cts[k * W + W - 1] = constWires_[k].value ? p[BCE_P_q] / 4 : 0;
                cslots.push_back(inst * stride_ + (uint32_t)constWires_[k].wire);
            }
            ck(bce_lwe_write(cc, cslots.data(), (uint32_t)cslots.size(), cts.data()), "SetInput(constants)");
        }
    }
    inputs_set_ = true;
}

// ---- sharding plan --------------------------------------------------------------------------
void Circuit::buildShardPlan() {
    owner_.clear();
    xwires_.clear();
    if (world_ <= 1 || shard_mode_ != 1) return;
    const size_t Lc = levels_.size();
    owner_.resize(Lc);
    std::vector<uint8_t> gate_owner(allGates.size(), 0xFF);  // 0xFF = everyone (OUTPUT)
    for (size_t l = 0; l < Lc; ++l) {
        const auto& gl = levels_[l].gates;
        uint64_t total = 0;
        for (int gi : gl) total += 4 * gate_weight(allGates[gi].op, xor_fast_) + 1;  // NOT/OUTPUT weigh 1/4 bootstrap
        uint64_t cum = 0;
        owner_[l].resize(gl.size());
        for (size_t k = 0; k < gl.size(); ++k) {
            const auto& g = allGates[gl[k]];
            uint8_t o = (uint8_t)std::min<uint64_t>(world_ - 1, cum * world_ / std::max<uint64_t>(total, 1));
            cum += 4 * gate_weight(g.op, xor_fast_) + 1;
            if (g.op == GateEnum::OUTPUT) o = 0xFF;
            owner_[l][k] = o;
            gate_owner[gl[k]] = o;
        }
    }
    xwires_.assign(Lc, std::vector<std::vector<int>>(world_));
    for (size_t l = 0; l < Lc; ++l) {
        const auto& gl = levels_[l].gates;
        for (size_t k = 0; k < gl.size(); ++k) {
            const auto& g = allGates[gl[k]];
            if (g.out < 0) continue;
            const uint8_t o = owner_[l][k];
            bool cross = false;
            for (uint32_t e = fan_off_[g.out]; e < fan_off_[g.out + 1] && !cross; ++e) cross = gate_owner[fan_gate_[e]] != o;
            if (cross) xwires_[l][o].push_back(g.out);
        }
    }
}

uint64_t Circuit::exchangeCapacity(uint32_t world, int shard_mode, bool encrypted) const {
    uint64_t W = 4;
    if (encrypted && cc) { uint64_t p[BCE_P_COUNT]; bce_get_params(cc, p); W = 4 * (p[BCE_P_n] + 1); }
    const uint64_t nout = n_output_bits.empty() ? 0 : n_output_bits[0];
    if (world <= 1) return 0;
    if (shard_mode == 0) return std::max<uint64_t>(64, (uint64_t)(instances_ / world) * nout);
    // mode 1: widest per-rank publication over all levels (plan must be built for this world)
    uint64_t widest = 0;
    for (const auto& lv : xwires_) for (const auto& r : lv) widest = std::max<uint64_t>(widest, r.size());
    if (xwires_.empty()) for (const auto& lv : levels_) widest = std::max<uint64_t>(widest, lv.gates.size());
    // bootstrap-depth schedule under gate sharding: per-step publications (before the plan exists for this world: a step's
    // outputs are at most its descriptors; slack filling under a larger world can widen steps up to the whole circuit's width)
    for (const auto& st : relevel_xw_) for (const auto& r : st) widest = std::max<uint64_t>(widest, r.size());
    if (relevel_xw_.empty()) for (const auto& st : relevel_plan_) widest = std::max<uint64_t>(widest, st.descs.size());
    // (callers size their buffers with this BEFORE set_exchange and ask again afterwards: the plan for the new world may
    // publish more per step than the estimate -- dist.Exchange does)
    return std::max<uint64_t>(64, widest * instances_ * (encrypted ? W : 1));
}

void Circuit::setExchange(uint32_t rank, uint32_t world, int shard_mode, bce_allgather_fn fn, void* user, void* host_send,
                          void* host_recv, void* dev_send, void* dev_recv, uint64_t capacity) {
    if (world == 0 || rank >= world) throw std::invalid_argument("setExchange: bad rank/world");
    if (world > 1 && !fn) throw std::invalid_argument("setExchange: allgather callback missing");
    if (world > 1 && shard_mode == 0 && instances_ % world) throw std::invalid_argument("setExchange: instance sharding needs instances divisible by world");
    if (world > 250) throw std::invalid_argument("setExchange: world too large");
    rank_ = rank; world_ = world; shard_mode_ = shard_mode; xfn_ = fn; xuser_ = user;
    host_send_ = host_send; host_recv_ = host_recv; dev_send_ = dev_send; dev_recv_ = dev_recv; xcap_ = capacity;
    buildShardPlan();
    rebuildRelevel();   // the number of instances this rank evaluates may have changed
}

uint64_t Circuit::planHash() const {
    uint64_t h = 0xcbf29ce484222325ull;   // FNV-1a over 64-bit words
    auto mix = [&](uint64_t v) { h = (h ^ v) * 0x100000001b3ull; };
    mix(world_); mix((uint64_t)shard_mode_); mix(stride_); mix(instances_); mix(relevel_ ? 1 : 0); mix(xor_fast_ ? 1 : 0);
    for (const auto& lv : owner_) { mix(lv.size()); for (uint8_t o : lv) mix(o); }
    for (const auto& lv : xwires_) for (const auto& r : lv) { mix(r.size()); for (int w : r) mix((uint64_t)w); }
    for (const auto& st : relevel_xw_) for (const auto& r : st) { mix(r.size()); for (int w : r) mix((uint64_t)w); }
    return h;
}

// after a level: publish wires whose consumers sit on other ranks (shard_mode 1)
void Circuit::exchangeLevel(size_t level) {
    if (world_ <= 1 || shard_mode_ != 1) return;
    exchangeWires(xwires_[level]);
}

// one exchange: pub[r] = the wires rank r publishes (every rank knows every list: the plan is static)
void Circuit::exchangeWires(const std::vector<std::vector<int>>& pub) {
    if (world_ <= 1 || shard_mode_ != 1) return;
    size_t widest = 0;
    for (uint32_t r = 0; r < world_; ++r) widest = std::max(widest, pub[r].size());
    if (widest == 0) return;
    const auto& mine = pub[rank_];
    const unsigned K = instances_;
    if (plaintext_flag) {
        const uint64_t bytes = (uint64_t)widest * K;
        if (bytes > xcap_ || !host_send_ || !host_recv_) throw std::runtime_error("exchange: host buffers too small");
        uint8_t* s = (uint8_t*)host_send_;
        std::memset(s, 0, bytes);
        for (unsigned i = 0; i < K; ++i) for (size_t k = 0; k < mine.size(); ++k) s[i * widest + k] = plain_[i][mine[k]];
        if (xfn_(xuser_, bytes, 0) != 0) throw std::runtime_error("exchange: allgather callback failed");
        const uint8_t* rcv = (const uint8_t*)host_recv_;
        for (uint32_t r = 0; r < world_; ++r) {
            if (r == rank_) continue;
            const auto& theirs = pub[r];
            for (unsigned i = 0; i < K; ++i) for (size_t k = 0; k < theirs.size(); ++k) plain_[i][theirs[k]] = rcv[(uint64_t)r * bytes + i * widest + k];
        }
        ++stats_.exchanges;
    }
    if (encrypted_flag) {
        uint64_t p[BCE_P_COUNT];
        bce_get_params(cc, p);
        const uint64_t W = 4 * (p[BCE_P_n] + 1);
        const uint64_t bytes = (uint64_t)widest * K * W;
        if (bytes > xcap_ || !dev_send_ || !dev_recv_) throw std::runtime_error("exchange: device buffers too small");
        std::vector<uint32_t> slots;
        for (unsigned i = 0; i < K; ++i) for (size_t k = 0; k < widest; ++k) slots.push_back(i * stride_ + (uint32_t)(k < mine.size() ? mine[k] : (mine.empty() ? 0 : mine[0])));
        ck(bce_pool_gather(cc, slots.data(), (uint32_t)slots.size(), dev_send_), "exchange(gather)");
        if (rccl_) {
            // stream-ordered: pack kernel -> ncclAllGather -> scatter kernels, all on the engine's stream
            ck(bce_rccl_allgather(cc, dev_send_, dev_recv_, bytes), "exchange(RCCL all-gather)");
        } else {
            ck(bce_synchronize(cc), "exchange(sync)");
            if (xfn_(xuser_, bytes, 1) != 0) throw std::runtime_error("exchange: allgather callback failed");
        }
        for (uint32_t r = 0; r < world_; ++r) {
            if (r == rank_) continue;
            const auto& theirs = pub[r];
            if (theirs.empty()) continue;
            // rows [i][k<theirs.size()] of rank r's block
            for (unsigned i = 0; i < K; ++i) {
                slots.clear();
                for (size_t k = 0; k < theirs.size(); ++k) slots.push_back(i * stride_ + (uint32_t)theirs[k]);
                const char* src = (const char*)dev_recv_ + (uint64_t)r * bytes + (uint64_t)i * widest * W;
                ck(bce_pool_scatter(cc, slots.data(), (uint32_t)slots.size(), src), "exchange(scatter)");
            }
        }
        ++stats_.exchanges;
        stats_.exchanged_cts += (uint64_t)mine.size() * K;
    }
}

// shard_mode 0: every rank ends with the outputs of all instances
void Circuit::gatherOutputs() {
    if (world_ <= 1 || shard_mode_ != 0) return;
    const uint64_t nout = n_output_bits[0];
    unsigned lo, hi;
    instanceRange(lo, hi);
    const uint64_t bytes = (uint64_t)(hi - lo) * nout;
    if (bytes == 0) return;
    if (bytes > xcap_ || !host_send_ || !host_recv_) throw std::runtime_error("gatherOutputs: host buffers too small");
    uint8_t* s = (uint8_t*)host_send_;
    for (unsigned i = lo; i < hi; ++i) std::memcpy(s + (uint64_t)(i - lo) * nout, circuitOut[i].data(), nout);
    if (xfn_(xuser_, bytes, 0) != 0) throw std::runtime_error("gatherOutputs: allgather callback failed");
    const uint8_t* rcv = (const uint8_t*)host_recv_;
    const unsigned per = hi - lo;
    for (uint32_t r = 0; r < world_; ++r)
        for (unsigned k = 0; k < per; ++k) std::memcpy(circuitOut[r * per + k].data(), rcv + (uint64_t)r * bytes + (uint64_t)k * nout, nout);
    ++stats_.exchanges;
}

// ---- re-levelled (bootstrap-depth) schedule: SURVEY 8(f2) -------------------------------------------
// Units of the schedule: a single bootstrap (AND / OR / XOR_FAST; its output is ready one step later) or an XOR built as
// the reference builds it (two ANDs in step s, their OR in step s + 1; ready two steps later).  ASAP placement gives the
// bootstrap depth of the circuit, D steps.  With balance_ on, the same D steps are filled by SLACK instead: one
// bootstrap is one workgroup, so a frontier call costs a staircase in its size (one bootstrap latency up to `lone`
// bootstraps, then one round per `full` resident workgroups; bce_launch_capacity), and a step that holds K x count
// bootstraps is topped up to the next stair with the ready units of least slack (ALAP order).  Units that must run now
// (ALAP step reached) always do, so the depth stays D; which step a gate runs in does not change its ciphertext.
void Circuit::launchCapacity(uint32_t& lone, uint32_t& full) const {
    lone = cap_lone_; full = cap_full_;
    if (lone == 0 || full == 0) {
        lone = 256; full = 512;
        if (cc) { uint32_t a = 0, b = 0; if (bce_launch_capacity(cc, &a, &b) == BCE_OK && a && b) { lone = a; full = b; } }
    }
}

// units of the bootstrap DAG in topological order (NOT chains resolved into negation flags); returns the bootstrap depth
uint32_t Circuit::buildUnits(std::vector<Unit>& units, std::vector<int>& base, std::vector<uint8_t>& neg) const {
    const size_t W = wire_names_.size(), G = allGates.size();
    // resolve NOT chains: wire -> (base wire, negated)
    base.resize(W);
    neg.assign(W, 0);
    for (size_t w = 0; w < W; ++w) base[w] = (int)w;
    // gates are in topological (file) order per level; walk levels so that bases are resolved first
    std::vector<uint32_t> depth(W, 0);
    units.clear();
    std::vector<int32_t> prod(W, -1);   // base wire -> unit that produces it
    units.reserve(G);
    for (const auto& L : levels_)
        for (int gi : L.gates) {
            const GateRec& g = allGates[gi];
            if (g.op == GateEnum::NOT) {
                base[g.out] = base[g.in[0]];
                neg[g.out] = neg[g.in[0]] ^ 1;
                depth[g.out] = depth[g.in[0]];
            } else if (g.op == GateEnum::AND || g.op == GateEnum::OR || g.op == GateEnum::XOR) {
                const uint32_t b0 = (uint32_t)base[g.in[0]], b1 = (uint32_t)base[g.in[1]];
                const uint32_t n0 = neg[g.in[0]], n1 = neg[g.in[1]];
                const uint32_t d = 1 + std::max(depth[g.in[0]], depth[g.in[1]]);
                Unit u{d, d, 1, 0, {0, b0, b1, (uint32_t)g.out, n0, n1}, prod[b0], prod[b1]};
                if (g.op != GateEnum::XOR) {
                    u.d.op = (uint32_t)(g.op == GateEnum::AND ? BCE_AND : BCE_OR);
                } else if (xor_fast_) {
                    // XOR_FAST of negated inputs: NOT a XOR NOT b = a XOR b; one negation flips the result
                    u.d.op = (uint32_t)((n0 ^ n1) ? BCE_XNOR_FAST : BCE_XOR_FAST);
                    u.d.neg0 = u.d.neg1 = 0;
                } else {
                    u.lat = 2;
                }
                depth[g.out] = d + u.lat - 1;
                prod[g.out] = (int32_t)units.size();
                units.push_back(u);
            }
        }
    uint32_t D = 0;
    for (const auto& u : units) D = std::max(D, u.asap + u.lat - 1);
    return D;
}

// successors (CSR) and ALAP start steps of the units (which are in topological order)
void Circuit::unitSuccessorsAlap(const std::vector<Unit>& units, uint32_t D, std::vector<uint32_t>& soff, std::vector<uint32_t>& succ,
                                 std::vector<uint32_t>& alap) {
    const size_t U = units.size();
    soff.assign(U + 1, 0);
    for (const auto& u : units) { if (u.p0 >= 0) ++soff[u.p0 + 1]; if (u.p1 >= 0 && u.p1 != u.p0) ++soff[u.p1 + 1]; }
    for (size_t i = 0; i < U; ++i) soff[i + 1] += soff[i];
    succ.resize(soff[U]);
    {
        std::vector<uint32_t> fill(soff.begin(), soff.end() - 1);
        for (size_t i = 0; i < U; ++i) {
            const Unit& u = units[i];
            if (u.p0 >= 0) succ[fill[u.p0]++] = (uint32_t)i;
            if (u.p1 >= 0 && u.p1 != u.p0) succ[fill[u.p1]++] = (uint32_t)i;
        }
    }
    alap.resize(U);
    for (size_t i = U; i-- > 0;) {
        uint32_t a = D - units[i].lat + 1;
        for (uint32_t k = soff[i]; k < soff[i + 1]; ++k) a = std::min(a, alap[succ[k]] - units[i].lat);
        alap[i] = a;
    }
}

void Circuit::dropPlan() {
    if (plan_) { bce_plan_destroy(cc, plan_); plan_ = nullptr; }
}

bool Circuit::graphActive() const {
    return graph_ && cc && relevel_ && !dataflowActive() && !verify_flag && !(world_ > 1 && shard_mode_ == 1);
}

void Circuit::buildRelevelPlan() {
    dropPlan();   // the resident copy of the schedule belongs to the plan it was built from
    const size_t W = wire_names_.size();
    std::vector<int> base;
    std::vector<uint8_t> neg;
    std::vector<Unit> units;
    const uint32_t D = buildUnits(units, base, neg);
    const size_t U = units.size();
    unsigned ilo, ihi;
    instanceRange(ilo, ihi);
    const uint64_t K = std::max(1u, ihi - ilo);
    const bool sharded = world_ > 1 && shard_mode_ == 1;   // gate sharding: every step's units are split over the ranks
    if (balance_ && U) {
        uint32_t lone, full;
        launchCapacity(lone, full);
        if (sharded) { lone *= world_; full *= world_; }   // the stairs of world_ devices working on one step
        std::vector<uint32_t> soff, succ, alap;
        unitSuccessorsAlap(units, D, soff, succ, alap);
        // list scheduling, least slack first
        using Key = std::pair<uint32_t, uint32_t>;   // (ALAP step, unit)
        std::priority_queue<Key, std::vector<Key>, std::greater<Key>> ready;
        std::vector<std::vector<uint32_t>> later(D + 2);   // units that become ready at a step
        std::vector<uint32_t> waiting(U), ready_at(U, 1);
        for (size_t i = 0; i < U; ++i) {
            const Unit& u = units[i];
            waiting[i] = (u.p0 >= 0) + (u.p1 >= 0 && u.p1 != u.p0);
            if (!waiting[i]) ready.push({alap[i], (uint32_t)i});
        }
        std::vector<uint32_t> ors_due(D + 2, 0);   // ORs of the XORs started one step earlier
        std::vector<uint32_t> chosen;
        for (uint32_t s = 1; s <= D; ++s) {
            for (uint32_t i : later[s]) ready.push({alap[i], i});
            chosen.clear();
            uint64_t cnt = ors_due[s];
            while (!ready.empty() && ready.top().first <= s) {   // no slack left
                const uint32_t i = ready.top().second; ready.pop();
                chosen.push_back(i); cnt += units[i].lat == 2 ? 2 : 1;
            }
            const uint64_t n = cnt * K;
            const uint64_t cap = (n <= lone ? lone : (n + full - 1) / full * full) / K;
            while (!ready.empty()) {
                const uint32_t i = ready.top().second;
                const uint64_t w = units[i].lat == 2 ? 2 : 1;
                if (cnt + w > cap) break;
                // an XOR started in the last step would put its OR beyond D only if its ALAP allowed it: it does not
                ready.pop(); chosen.push_back(i); cnt += w;
            }
            for (uint32_t i : chosen) {
                Unit& u = units[i];
                u.start = s;
                if (u.lat == 2) ++ors_due[s + 1];
                for (uint32_t k = soff[i]; k < soff[i + 1]; ++k) {
                    const uint32_t q = succ[k];
                    ready_at[q] = std::max(ready_at[q], s + u.lat);
                    if (--waiting[q] == 0) later[ready_at[q]].push_back(q);
                }
            }
        }
        if (!ready.empty()) throw std::logic_error("buildRelevelPlan: units left unscheduled");
    }
    // gate sharding: the units that start in a step are split over the ranks in netlist order by bootstrap weight (an XOR's
    // three bootstraps stay on one rank: its temporaries are local), on top of the ORs each rank carries over from the
    // previous step.  Every rank computes the same plan.
    relevel_xw_.clear();
    if (sharded) {
        std::vector<std::vector<uint32_t>> by_step(D + 2);
        for (size_t i = 0; i < U; ++i) by_step[units[i].start].push_back((uint32_t)i);
        auto assign_owners = [&](bool locality) {
        std::vector<uint64_t> carried(world_, 0), next_carried(world_, 0);
        for (uint32_t st = 1; st <= D; ++st) {
            uint64_t total = 0;
            for (uint32_t r = 0; r < world_; ++r) total += carried[r];
            for (uint32_t i : by_step[st]) total += units[i].lat == 2 ? 2 : 1;
            std::fill(next_carried.begin(), next_carried.end(), 0);
            if (!locality) {
                // contiguous split in netlist order: rank r ends where the running load (carried ORs of ranks <= r + the units
                // given out so far) reaches (r + 1) / world of the step's total (midpoint rule: within one unit of the fair share)
                uint32_t r = 0;
                uint64_t cum = carried[0];
                for (uint32_t i : by_step[st]) {
                    const uint64_t w = units[i].lat == 2 ? 2 : 1;
                    while (r + 1 < world_ && (2 * cum + w) * world_ > 2 * (uint64_t)(r + 1) * total) { ++r; cum += carried[r]; }
                    units[i].owner = (uint8_t)r;
                    cum += w;
                    if (units[i].lat == 2) ++next_carried[r];
                }
            } else {
                // locality first ("schedule a gate on the GPU that produced most of its inputs"), balance as the
                // constraint: every rank may take up to its fair share of the step's bootstraps (rounded up, + one unit so that
                // an XOR's pair never has to split).  Units whose two producers sit on one rank choose first, then those with
                // one producing rank, then the free ones fill the least loaded ranks.  Deterministic: every rank computes it.
                const uint64_t share = (total + world_ - 1) / world_ + 1;
                std::vector<uint64_t> load(carried);
                std::vector<uint32_t> rest;
                auto place = [&](uint32_t i, uint32_t r) {
                    units[i].owner = (uint8_t)r;
                    load[r] += units[i].lat == 2 ? 2 : 1;
                    if (units[i].lat == 2) ++next_carried[r];
                };
                auto owner_of = [&](int32_t p) -> int { return p >= 0 ? (int)units[p].owner : -1; };
                for (int pass = 0; pass < 2; ++pass)
                    for (uint32_t i : by_step[st]) {
                        const uint64_t w = units[i].lat == 2 ? 2 : 1;
                        const int a = owner_of(units[i].p0), b = owner_of(units[i].p1);
                        const bool both = a >= 0 && a == b;
                        if (pass == 0) {
                            if (both && load[a] + w <= share) place(i, (uint32_t)a);
                            else if (!both) continue;
                            else rest.push_back(i);
                        } else if (!both) {
                            // one producing rank, or two different ones: the lighter of them if it has room
                            int c = -1;
                            if (a >= 0 && load[a] + w <= share) c = a;
                            if (b >= 0 && load[b] + w <= share && (c < 0 || load[b] < load[c])) c = b;
                            if (c >= 0) place(i, (uint32_t)c); else rest.push_back(i);
                        }
                    }
                std::sort(rest.begin(), rest.end());   // netlist order
                for (uint32_t i : rest) {
                    uint32_t r = 0;
                    for (uint32_t k = 1; k < world_; ++k) if (load[k] < load[r]) r = k;
                    place(i, r);
                }
            }
            carried.swap(next_carried);
        }
        };
        // outputs that cross ranks under an assignment (consumers elsewhere; OUTPUT gates are read by every rank either way)
        auto crossings = [&]() {
            std::vector<uint8_t> x(U, 0);
            for (size_t i = 0; i < U; ++i) {
                const Unit& u = units[i];
                if (u.p0 >= 0 && units[u.p0].owner != u.owner) x[u.p0] = 1;
                if (u.p1 >= 0 && units[u.p1].owner != u.owner) x[u.p1] = 1;
            }
            uint64_t n = 0;
            for (uint8_t v : x) n += v;
            return n;
        };
        assign_owners(false);
        if (shard_locality_) {
            // keep whichever split publishes less: netlist order already is a locality order for some circuits (sha256 on two
            // ranks), input-following placement wins on others (AES-expanded on eight: 21.0 k -> 11.9 k crossing outputs)
            const uint64_t contiguous = crossings();
            assign_owners(true);
            if (crossings() > contiguous) assign_owners(false);
        }
        // publications: an output crosses when a consumer unit sits on another rank, or when an OUTPUT gate reads it (every
        // rank decrypts every output, as in the gate-level plan); it is published after the step that produces it
        std::vector<uint8_t> feeds_output(W, 0);
        for (const auto& g : allGates) if (g.op == GateEnum::OUTPUT) feeds_output[base[g.in[0]]] = 1;
        std::vector<uint8_t> cross(U, 0);
        for (size_t i = 0; i < U; ++i) {
            const Unit& u = units[i];
            if (u.p0 >= 0 && units[u.p0].owner != u.owner) cross[u.p0] = 1;
            if (u.p1 >= 0 && units[u.p1].owner != u.owner) cross[u.p1] = 1;
            if (feeds_output[u.d.out]) cross[i] = 1;
        }
        relevel_xw_.assign(D, std::vector<std::vector<int>>(world_));
        for (size_t i = 0; i < U; ++i)
            if (cross[i]) relevel_xw_[units[i].start + units[i].lat - 2][units[i].owner].push_back((int)units[i].d.out);
    }
    // temporaries of the XORs: two parity banks (a step's ANDs write one bank while the previous step's ORs read the other)
    std::vector<uint32_t> xor_at(D + 2, 0);
    uint32_t max_x = 0;
    for (const auto& u : units) if (u.lat == 2 && (!sharded || u.owner == rank_)) max_x = std::max(max_x, ++xor_at[u.start]);
    if (sharded) {   // the slot stride must be the same on every rank: size the banks for the fullest step of ANY rank
        std::vector<uint32_t> cnt((size_t)(D + 2) * world_, 0);
        for (const auto& u : units) if (u.lat == 2) max_x = std::max(max_x, ++cnt[(size_t)u.start * world_ + u.owner]);
    }
    const uint32_t tmp0 = (uint32_t)W;
    relevel_stride_ = tmp0 + 4 * max_x;
    relevel_plan_.assign(D, RStep{});
    std::fill(xor_at.begin(), xor_at.end(), 0);
    for (const auto& u : units) {
        if (sharded && u.owner != rank_) continue;
        if (u.lat == 1) {
            relevel_plan_[u.start - 1].descs.push_back(u.d);
        } else {
            const uint32_t idx = xor_at[u.start]++;
            const uint32_t t1 = tmp0 + (u.start & 1) * 2 * max_x + 2 * idx, t2 = t1 + 1;
            // (a AND !b), (!a AND b) with the inputs' own negations folded in, then OR one step later
            relevel_plan_[u.start - 1].descs.push_back({BCE_AND, u.d.in0, u.d.in1, t1, u.d.neg0, u.d.neg1 ^ 1u});
            relevel_plan_[u.start - 1].descs.push_back({BCE_AND, u.d.in0, u.d.in1, t2, u.d.neg0 ^ 1u, u.d.neg1});
            relevel_plan_[u.start].descs.push_back({BCE_OR, t1, t2, u.d.out, 0, 0});
        }
    }
    relevel_K_ = (uint32_t)K;
    // NOT wires consumed by OUTPUT gates need a real ciphertext (decrypt must see EvalNOT's output)
    relevel_nots_.clear();
    std::vector<uint8_t> done_not(W, 0);
    for (const auto& g : allGates)
        if (g.op == GateEnum::OUTPUT && neg[g.in[0]] && !done_not[g.in[0]]) {
            done_not[g.in[0]] = 1;
            relevel_nots_.push_back({BCE_OP_NOT, (uint32_t)base[g.in[0]], (uint32_t)base[g.in[0]], (uint32_t)g.in[0], 0, 0});
        } else if (g.op == GateEnum::OUTPUT && !neg[g.in[0]] && base[g.in[0]] != g.in[0] && !done_not[g.in[0]]) {
            done_not[g.in[0]] = 1;  // double negation: plain copy of the base
            relevel_nots_.push_back({BCE_OP_COPY, (uint32_t)base[g.in[0]], (uint32_t)base[g.in[0]], (uint32_t)g.in[0], 0, 0});
        }
}

// (re)build the bootstrap-depth schedule for the current K / capacities and size the per-instance slot stride for it.
// The stride is part of the pool layout: not after SetInput.
void Circuit::rebuildRelevel() {
    buildRelevelPlan();
    if (dataflow_) buildDagTasks();
    uint32_t need = std::max(base_stride_, relevel_stride_);
    if (dataflow_) need = std::max(need, dag_stride_);
    if (inputs_set_ && need > stride_ && balance_) {
        // the inputs already sit in a pool laid out for a smaller stride: keep the layout, fall back to ASAP placement
        // (whose temporaries the stride of finalizeNetlist() always covers)
        balance_ = false;
        buildRelevelPlan();
        balance_ = true;
        need = std::max(base_stride_, relevel_stride_);
        if (dataflow_) need = std::max(need, dag_stride_);
    }
    if (inputs_set_ && need > stride_) throw std::logic_error("the schedule needs a larger slot stride than the pool was laid out with");
    if (!inputs_set_) stride_ = need;
}

// per-step bootstrap counts of the schedule (for one instance), and a self-check: every input of every step was produced
// by an earlier step (or is a primary input / constant), every XOR temporary is read exactly one step after it is written
std::vector<uint32_t> Circuit::relevelStepSizes() const {
    std::vector<uint32_t> v;
    for (const auto& st : relevel_plan_) v.push_back((uint32_t)st.descs.size());
    return v;
}

std::vector<uint32_t> Circuit::relevelPublications() const {
    std::vector<uint32_t> v(relevel_plan_.size(), 0);
    for (size_t s = 0; s < relevel_xw_.size() && s < v.size(); ++s) v[s] = (uint32_t)relevel_xw_[s][rank_].size();
    return v;
}

bool Circuit::checkRelevelPlan(std::string* why) const {
    const size_t W = wire_names_.size();
    std::vector<int32_t> written(relevel_stride_, -1);   // step that wrote a slot; inputs and constants: step -1 = "before"
    std::vector<uint8_t> is_out(W, 0);
    for (const auto& g : allGates) if ((g.op == GateEnum::AND || g.op == GateEnum::OR || g.op == GateEnum::XOR) && g.out >= 0) is_out[g.out] = 1;
    auto fail = [&](const std::string& m) { if (why) *why = m; return false; };
    for (size_t s = 0; s < relevel_plan_.size(); ++s) {
        for (const auto& d : relevel_plan_[s].descs) {
            for (uint32_t in : {d.in0, d.in1}) {
                if (in >= relevel_stride_) return fail("input slot outside the stride");
                if (in < W) {
                    if (is_out[in] && (written[in] < 0 || written[in] >= (int32_t)s)) return fail("step " + std::to_string(s) + " reads register " + std::to_string(in) + " before it is written");
                } else if (written[in] != (int32_t)s - 1) {
                    return fail("step " + std::to_string(s) + " reads an XOR temporary that was not written in the previous step");
                }
            }
        }
        for (const auto& d : relevel_plan_[s].descs) {
            if (d.out >= relevel_stride_) return fail("output slot outside the stride");
            if (d.out < W && written[d.out] >= 0) return fail("register written twice");
            if (written[d.out] == (int32_t)s) return fail("slot written twice in one step");
            written[d.out] = (int32_t)s;
        }
        // gate sharding: what the other ranks publish after this step arrives before the next one
        if (s < relevel_xw_.size())
            for (uint32_t r = 0; r < world_; ++r) {
                if (r == rank_) {
                    for (int w : relevel_xw_[s][r]) if (written[w] != (int32_t)s) return fail("publishes register " + std::to_string(w) + " in a step that did not write it");
                } else {
                    for (int w : relevel_xw_[s][r]) { if (written[w] >= 0) return fail("receives a register it wrote itself"); written[w] = (int32_t)s; }
                }
            }
    }
    if (relevel_xw_.empty())
        for (size_t w = 0; w < W; ++w) if (is_out[w] && written[w] < 0) return fail("register " + std::to_string(w) + " never written");
    return true;
}

void Circuit::setXorFast(bool b) {
    xor_fast_ = gep.xor_fast = b;
    buildShardPlan();
    rebuildRelevel();
}

void Circuit::setBalance(bool on, uint32_t lone, uint32_t full) {
    balance_ = on; cap_lone_ = lone; cap_full_ = full;
    rebuildRelevel();
}

void Circuit::clockReleveled() {
    if (verify_flag) throw std::logic_error("re-levelled schedule is not available in verify mode");
    unsigned lo, hi;
    instanceRange(lo, hi);
    const uint32_t K = hi - lo;
    if (relevel_plan_.empty() || (balance_ && relevel_K_ != std::max(1u, K))) rebuildRelevel();
    if (relevel_stride_ > stride_) throw std::logic_error("re-levelled schedule needs more scratch slots than the pool stride");
    const bool sharded = world_ > 1 && shard_mode_ == 1;
    // the schedule's descriptors live on the device from the first Clock() on (bce_plan): a step is one call without an
    // upload; with setGraph the whole schedule is one hipGraph launch
    if (plan_ && (plan_lo_ != lo || plan_K_ != K || plan_stride_ != stride_)) dropPlan();
    if (!plan_ && K) {
        std::vector<uint32_t> sizes;
        std::vector<bce_gate_desc> all;
        for (const auto& st : relevel_plan_)
            if (!st.descs.empty()) { sizes.push_back((uint32_t)st.descs.size()); all.insert(all.end(), st.descs.begin(), st.descs.end()); }
        if (!sizes.empty()) {
            ck(bce_plan_create(cc, (uint32_t)sizes.size(), sizes.data(), all.data(), K, stride_, lo * stride_, &plan_), "Clock(schedule upload)");
            plan_lo_ = lo; plan_K_ = K; plan_stride_ = stride_;
        }
    }
    if (plan_ && graphActive()) {
        ck(bce_plan_run(cc, plan_), "Clock(schedule graph)");
        for (const auto& st : relevel_plan_) if (!st.descs.empty()) ++stats_.sublaunches;
    } else {
        uint32_t ps = 0;
        for (size_t s = 0; s < relevel_plan_.size(); ++s) {
            if (plan_ && !relevel_plan_[s].descs.empty()) {
                ck(bce_plan_run_step(cc, plan_, ps++), "Clock(re-levelled step)");
                ++stats_.sublaunches;
            }
            if (sharded) exchangeWires(relevel_xw_[s]);   // outputs of this step whose consumers sit on other ranks
        }
    }
    finishReleveled(lo, hi);
    stats_.levels = (uint32_t)relevel_plan_.size();
}

// ---- dataflow schedule: the whole bootstrap DAG in one persistent launch (bce_dag_*) -------------------------------
// Tasks = the units of the bootstrap-depth schedule in topological order, an XOR as its two ANDs and its OR with
// temporaries of its own (SSA: the device runs independent tasks in any order, so no slot may be reused).  Priority
// class of a task = slack of its unit (ALAP step - ASAP step): the device pulls ready tasks of the critical path first.
void Circuit::buildDagTasks() {
    const size_t W = wire_names_.size();
    std::vector<int> base;
    std::vector<uint8_t> neg;
    std::vector<Unit> units;
    const uint32_t D = buildUnits(units, base, neg);
    std::vector<uint32_t> soff, succ, alap;
    unitSuccessorsAlap(units, D, soff, succ, alap);
    uint32_t cls[3] = {0, 1, 2};   // slack bounds of classes 0, 1, 2 (steps); development knob BCE_DAG_CLASSES=a,b,c
                                   // (0,2,8 and 1,4,16 are 2-3 % slower on AES at K = 4 / 8)
    if (const char* e = std::getenv("BCE_DAG_CLASSES")) std::sscanf(e, "%u,%u,%u", &cls[0], &cls[1], &cls[2]);
    dag_tasks_.clear(); dag_prio_.clear();
    uint32_t nx = 0;
    for (size_t i = 0; i < units.size(); ++i) {
        const Unit& u = units[i];
        const uint32_t slack = alap[i] - u.asap;
        const uint8_t pc = slack <= cls[0] ? 0 : slack <= cls[1] ? 1 : slack <= cls[2] ? 2 : 3;
        if (u.lat == 1) {
            dag_tasks_.push_back(u.d); dag_prio_.push_back(pc);
        } else {
            const uint32_t t1 = (uint32_t)W + 2 * nx, t2 = t1 + 1;
            ++nx;
            // (a AND !b), (!a AND b) with the inputs' own negations folded in, then their OR (src/gate.cpp:198-202)
            dag_tasks_.push_back({BCE_AND, u.d.in0, u.d.in1, t1, u.d.neg0, u.d.neg1 ^ 1u});
            dag_tasks_.push_back({BCE_AND, u.d.in0, u.d.in1, t2, u.d.neg0 ^ 1u, u.d.neg1});
            dag_tasks_.push_back({BCE_OR, t1, t2, u.d.out, 0, 0});
            dag_prio_.insert(dag_prio_.end(), 3, pc);
        }
    }
    dag_stride_ = (uint32_t)W + 2 * nx;
    dropDag();
}

void Circuit::dropDag() {
    if (dag_) { bce_dag_destroy(cc, dag_); dag_ = nullptr; }
}

void Circuit::setDataflow(bool b) {
    if (b && inputs_set_ && !dataflow_) throw std::logic_error("setDataflow: choose the dataflow schedule before SetInput (it lays the pool out with its own temporaries)");
    dataflow_ = b;
    rebuildRelevel();
}

bool Circuit::dataflowActive() const {
    return dataflow_ && cc && !verify_flag && !(world_ > 1 && shard_mode_ == 1) && bce_dag_supported(cc) && !dag_tasks_.empty();
}

void Circuit::clockDataflow() {
    unsigned lo, hi;
    instanceRange(lo, hi);
    if (dag_stride_ > stride_) throw std::logic_error("dataflow schedule needs more scratch slots than the pool stride");
    if (relevel_plan_.empty()) rebuildRelevel();
    if (!dag_) ck(bce_dag_create(cc, (uint32_t)dag_tasks_.size(), dag_tasks_.data(), dag_prio_.data(), &dag_), "Clock(dataflow DAG)");
    if (hi > lo) {
        ck(bce_dag_run(cc, dag_, hi - lo, stride_, lo * stride_), "Clock(dataflow run)");
        ++stats_.sublaunches;
    }
    finishReleveled(lo, hi);
    stats_.levels = 1;
}

// NOT wires the OUTPUT gates read, OUTPUT gates (decrypt), gate counts: common end of the two DAG-level schedules
void Circuit::finishReleveled(unsigned lo, unsigned hi) {
    const uint32_t K = hi - lo;
    if (!relevel_nots_.empty() && K) {
        std::vector<bce_gate_desc> d(relevel_nots_);
        for (auto& e : d) { e.in0 += lo * stride_; e.in1 += lo * stride_; e.out += lo * stride_; }
        ck(bce_eval_gates_strided(cc, (uint32_t)d.size(), d.data(), K, stride_), "Clock(output NOTs)");
        ++stats_.sublaunches;
    }
    // OUTPUT gates
    std::vector<uint32_t> oslots;
    std::vector<std::pair<unsigned, int>> obits;
    for (const auto& g : allGates)
        if (g.op == GateEnum::OUTPUT)
            for (unsigned i = lo; i < hi; ++i) { oslots.push_back(i * stride_ + g.in[0]); obits.push_back({i, g.out_bit}); }
    if (!oslots.empty()) {
        std::vector<uint8_t> res(oslots.size());
        ck(bce_decrypt_bits(cc, oslots.data(), (uint32_t)oslots.size(), res.data()), "Clock(Decrypt)");
        for (size_t k = 0; k < oslots.size(); ++k) circuitOut[obits[k].first][obits[k].second] = res[k];
    }
    for (const auto& g : allGates) {
        switch (g.op) {
            case GateEnum::OUTPUT: ++n_output_gates; break;
            case GateEnum::NOT: ++n_not_gates; break;
            case GateEnum::AND: ++n_and_gates; break;
            case GateEnum::OR: ++n_or_gates; break;
            case GateEnum::XOR: ++n_xor_gates; break;
            default: break;
        }
    }
}

// ---- Clock -------------------------------------------------------------------------------------
void Circuit::managerRound(size_t) {
    // Readiness was resolved once in finalizeNetlist(); the per-round work the reference does here
    // (src/circuit.cpp:575-683: scanning waitingGates for every active wire) has no counterpart.
}

void Circuit::executeRound(size_t level) {
    const Level& L = levels_[level];
    unsigned lo, hi;
    instanceRange(lo, hi);
    const bool sharded_gates = world_ > 1 && shard_mode_ == 1;
    auto mine = [&](size_t k) { return !sharded_gates || owner_[level][k] == 0xFF || owner_[level][k] == rank_; };

    if (plaintext_flag) {
        for (unsigned i = lo; i < hi; ++i) {
            auto& pv = plain_[i];
            for (size_t k = 0; k < L.gates.size(); ++k) {
                if (!mine(k)) continue;
                const GateRec& g = allGates[L.gates[k]];
                switch (g.op) {
                    case GateEnum::NOT: pv[g.out] = !pv[g.in[0]]; break;
                    case GateEnum::AND: pv[g.out] = pv[g.in[0]] && pv[g.in[1]]; break;
                    case GateEnum::OR: pv[g.out] = pv[g.in[0]] || pv[g.in[1]]; break;
                    case GateEnum::XOR: pv[g.out] = pv[g.in[0]] ^ pv[g.in[1]]; break;
                    default: break;
                }
            }
        }
    }
    if (encrypted_flag) {
        requireEngine("Clock");
        const uint32_t tmp0 = (uint32_t)wire_names_.size();
        if (batched_) {
            // stage A: AND / OR gates and the two ANDs of every XOR; stage B: the OR of every XOR
            std::vector<bce_gate_desc> A, B;
            uint32_t x = 0;
            for (size_t k = 0; k < L.gates.size(); ++k) {
                const GateRec& g = allGates[L.gates[k]];
                const bool me = mine(k);
                if (g.op == GateEnum::XOR && xor_fast_) {
                    if (me) A.push_back({BCE_XOR_FAST, (uint32_t)g.in[0], (uint32_t)g.in[1], (uint32_t)g.out, 0, 0});
                } else if (g.op == GateEnum::XOR) {
                    const uint32_t t1 = tmp0 + 2 * x, t2 = t1 + 1;
                    ++x;
                    if (!me) continue;
                    A.push_back({BCE_AND, (uint32_t)g.in[0], (uint32_t)g.in[1], t1, 0, 1});
                    A.push_back({BCE_AND, (uint32_t)g.in[0], (uint32_t)g.in[1], t2, 1, 0});
                    B.push_back({BCE_OR, t1, t2, (uint32_t)g.out, 0, 0});
                } else if (!me) {
                    continue;
                } else if (g.op == GateEnum::AND || g.op == GateEnum::OR) {
                    A.push_back({(uint32_t)(g.op == GateEnum::AND ? BCE_AND : BCE_OR), (uint32_t)g.in[0], (uint32_t)g.in[1], (uint32_t)g.out, 0, 0});
                } else if (g.op == GateEnum::NOT) {
                    A.push_back({BCE_OP_NOT, (uint32_t)g.in[0], (uint32_t)g.in[0], (uint32_t)g.out, 0, 0});
                }
            }
            const uint32_t K = hi - lo;
            if (!A.empty()) {
                // descriptors address instance `lo`; the strided call replicates them K times
                for (auto& d : A) { d.in0 += lo * stride_; d.in1 += lo * stride_; d.out += lo * stride_; }
                ck(bce_eval_gates_strided(cc, (uint32_t)A.size(), A.data(), K, stride_), "Clock(stage A)");
                ++stats_.sublaunches;
            }
            if (!B.empty()) {
                for (auto& d : B) { d.in0 += lo * stride_; d.in1 += lo * stride_; d.out += lo * stride_; }
                ck(bce_eval_gates_strided(cc, (uint32_t)B.size(), B.data(), K, stride_), "Clock(stage B)");
                ++stats_.sublaunches;
            }
        } else {
            // reference shape: one Gate::Evaluate per gate (src/circuit.cpp:698-710)
            GateEvalParams p = gep;
            p.plaintext_flag = false;  // plaintext was done above for every instance
            p.verify_flag = false;
            for (unsigned i = lo; i < hi; ++i) {
                uint32_t x = 0;
                for (size_t k = 0; k < L.gates.size(); ++k) {
                    const GateRec& g = allGates[L.gates[k]];
                    uint32_t xi = (g.op == GateEnum::XOR) ? x++ : 0;
                    if (!mine(k) || g.op == GateEnum::OUTPUT) continue;
                    Gate ge;
                    ge.name = g.name; ge.op = g.op;
                    for (int q = 0; q < g.nin; ++q) { ge.encin.push_back(i * stride_ + g.in[q]); ge.ready.push_back(true); }
                    ge.encout.assign(1, i * stride_ + g.out);
                    ge.tmp = {i * stride_ + tmp0 + 2 * xi, i * stride_ + tmp0 + 2 * xi + 1};
                    ge.Evaluate(p);
                }
            }
        }
        if (verify_flag) {
            // decrypt every output of the level, compare with the plaintext pass, repair mismatches
            std::vector<uint32_t> slots;
            std::vector<uint8_t> expect;
            std::vector<const char*> names;
            for (unsigned i = lo; i < hi; ++i)
                for (size_t k = 0; k < L.gates.size(); ++k) {
                    const GateRec& g = allGates[L.gates[k]];
                    if (!mine(k)) continue;
                    int w = g.op == GateEnum::OUTPUT ? g.in[0] : g.out;
                    slots.push_back(i * stride_ + w);
                    expect.push_back(plain_[i][w]);
                    names.push_back(op_name(g.op));
                }
            std::vector<uint8_t> got(slots.size());
            if (!slots.empty()) ck(bce_decrypt_bits(cc, slots.data(), (uint32_t)slots.size(), got.data()), "Clock(verify)");
            for (size_t k = 0; k < slots.size(); ++k) {
                if (got[k] == expect[k]) continue;
                std::cerr << "Bad " << names[k] << " fixing" << std::endl;
                ++stats_.verify_fixes;
                if (std::strcmp(names[k], "OUTPUT") != 0)
                    ck(bce_encrypt_bits(cc, &expect[k], &slots[k], 1, (1ull << 40) + enc_counter_++, encrypt_mode_), "Clock(fix)");
            }
        }
    }
    exchangeLevel(level);

    // retire: counters (once per evaluation, src/circuit.cpp:722-749) and OUTPUT gates (:796-807)
    std::vector<uint32_t> oslots;
    std::vector<std::pair<unsigned, int>> obits;
    for (size_t k = 0; k < L.gates.size(); ++k) {
        const GateRec& g = allGates[L.gates[k]];
        switch (g.op) {
            case GateEnum::OUTPUT: ++n_output_gates; break;
            case GateEnum::NOT: ++n_not_gates; break;
            case GateEnum::AND: ++n_and_gates; break;
            case GateEnum::OR: ++n_or_gates; break;
            case GateEnum::XOR: ++n_xor_gates; break;
            default: break;
        }
        if (g.op != GateEnum::OUTPUT) continue;
        if (!encrypted_flag && !plaintext_flag) std::cerr << "Error either encrypted or plaintext flag must be set" << std::endl;
        for (unsigned i = lo; i < hi; ++i) {
            if (encrypted_flag) { oslots.push_back(i * stride_ + g.in[0]); obits.push_back({i, g.out_bit}); }
            else circuitOut[i][g.out_bit] = plain_[i][g.in[0]];
        }
    }
    if (!oslots.empty()) {
        std::vector<uint8_t> res(oslots.size());
        ck(bce_decrypt_bits(cc, oslots.data(), (uint32_t)oslots.size(), res.data()), "Clock(Decrypt)");
        for (size_t k = 0; k < oslots.size(); ++k) circuitOut[obits[k].first][obits[k].second] = res[k];
    }
}

Outputs Circuit::Clock() {
    if (done) throw std::logic_error("done ckt clocked! should reset");  // the reference exits (src/circuit.cpp:538-541)
    if (!inputs_set_) throw std::logic_error("Clock: SetInput has not been called (no active wires)");
    if (!plaintext_flag && !encrypted_flag) throw std::logic_error("Error either encrypted or plaintext flag must be set");
    auto t_total = Clock_t::now();
    double management = 0, execution = 0;
    uint64_t boots0 = 0;
    if (encrypted_flag) {
        requireEngine("Clock");
        ck(bce_pool_reserve(cc, instances_ * stride_), "Clock(pool)");
        bce_timing t;
        ck(bce_timing_get(cc, &t), "Clock");
        boots0 = t.bootstraps;
    }
    size_t done_gates = 0;
    // gate-level rounds (the reference's Clock loop) whenever a plaintext pass rides along (verify mode) or the caller asked for
    // one Gate::Evaluate per gate (setBatched(false)); otherwise the bootstrap-depth schedule, unless setRelevel(false)
    const bool releveled = (relevel_ || dataflow_) && encrypted_flag && !plaintext_flag && batched_;
    if (releveled) {
        auto t0 = Clock_t::now();
        if (dataflowActive()) clockDataflow(); else clockReleveled();
        execution += ms_since(t0);
        done_gates = allGates.size();
    }
    for (size_t l = 0; l < levels_.size() && inputs_set_ && !releveled; ++l) {
        auto t0 = Clock_t::now();
        managerRound(l);
        management += ms_since(t0);
        t0 = Clock_t::now();
        executeRound(l);
        execution += ms_since(t0);
        done_gates += levels_[l].gates.size();
        ++stats_.levels;
        if (!quiet_) std::cout << "\rProcessing: " << done_gates << " of " << allGates.size() << std::flush;
    }
    if (encrypted_flag) {
        auto t0 = Clock_t::now();
        bce_timing t;
        ck(bce_timing_get(cc, &t), "Clock");  // synchronizes
        stats_.bootstraps = t.bootstraps - boots0;
        execution += ms_since(t0);
    }
    gatherOutputs();
    if (done_gates == allGates.size()) done = true;
    stats_.total_ms = ms_since(t_total);
    stats_.management_ms = management;
    stats_.execution_ms = execution;
    if (!quiet_) {
        std::cout << std::endl << "### Total time " << (unsigned)std::max(1.0, stats_.total_ms) << " msec" << std::endl;
        std::cout << std::endl << "efficiency " << float(std::max(1.0, execution)) / float(std::max(1.0, stats_.total_ms)) * 100.0 << "%" << std::endl;
    }
    return getOutputs(0);
}

Outputs Circuit::getOutputs(unsigned instance) const {
    // one vector per output value (Bristol Fashion circuits may have several; the reference has one, src/circuit.cpp:183-185)
    Outputs o(std::max<size_t>(1, out_bus_bits_.size()));
    if (instance >= circuitOut.size()) return o;
    const auto& bits = circuitOut[instance];
    size_t pos = 0;
    for (size_t b = 0; b < out_bus_bits_.size(); ++b) {
        const size_t w = std::min<size_t>(out_bus_bits_[b], bits.size() - std::min(bits.size(), pos));
        o[b].assign(bits.begin() + pos, bits.begin() + pos + w);
        pos += w;
    }
    if (out_bus_bits_.empty()) o[0].assign(bits.begin(), bits.end());
    return o;
}

void Circuit::getCounts(uint32_t out[6]) const {
    out[0] = n_input_gates; out[1] = n_output_gates; out[2] = n_not_gates;
    out[3] = n_and_gates; out[4] = n_or_gates; out[5] = n_xor_gates;
}

bce_circuit_info Circuit::info() const {
    bce_circuit_info I{};
    I.n_gates = (uint32_t)allGates.size();
    I.n_input_gates = (uint32_t)inputGates.size();
    I.n_wires = (uint32_t)wire_names_.size();
    I.n_inputs = n_buses_;
    I.n_input_bits[0] = n_in_bits_.size() > 0 ? n_in_bits_[0] : 0;   // every bus: bce_circuit_get_buses()
    I.n_input_bits[1] = n_in_bits_.size() > 1 ? n_in_bits_[1] : 0;
    I.n_output_bits = n_output_bits.empty() ? 0 : n_output_bits[0];
    I.n_levels = (uint32_t)levels_.size();
    I.n_relevel_steps = (uint32_t)relevel_plan_.size();
    I.slot_stride = stride_;
    for (const auto& L : levels_) {
        uint32_t a = 0, b = 0;
        for (int gi : L.gates) {
            GateEnum op = allGates[gi].op;
            if (op == GateEnum::AND || op == GateEnum::OR) ++a;
            if (op == GateEnum::XOR && xor_fast_) ++a;
            else if (op == GateEnum::XOR) { a += 2; ++b; }
        }
        if (a) ++I.n_sublaunches;
        if (b) ++I.n_sublaunches;
        I.max_frontier = std::max(I.max_frontier, std::max(a, b));
        I.n_bootstraps += a + b;
    }
    return I;
}

void Circuit::dumpNetList() const {
    std::cout << "Netlist " << std::endl;
    NetList nl;
    for (size_t w = 0; w < wire_names_.size(); ++w) {
        NameList& f = nl[wire_names_[w]];
        for (uint32_t e = fan_off_[w]; e < fan_off_[w + 1]; ++e) f.push_back(allGates[fan_gate_[e]].name);
    }
    for (const auto& it : nl) {
        std::cout << it.first;
        for (const auto& g : it.second) std::cout << " " << g;
        std::cout << std::endl;
    }
}

void Circuit::dumpGates() const {
    std::cout << "Inputlist " << std::endl;
    for (const auto& l : inputGates) std::cout << l.name << std::endl;
    std::cout << "Alllist " << std::endl;
    for (const auto& g : allGates) std::cout << g.name << std::endl;
}

void Circuit::dumpGateCount() const {
    std::cout << "Number of input gates " << n_input_gates << std::endl;
    std::cout << "Number of output gates " << n_output_gates << std::endl;
    std::cout << "Number of not gates " << n_not_gates << std::endl;
    std::cout << "Number of and gates " << n_and_gates << std::endl;
    std::cout << "Number of or gates " << n_or_gates << std::endl;
    std::cout << "Number of xor gates " << n_xor_gates << std::endl;
}

}  // namespace bce
