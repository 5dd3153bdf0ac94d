// circuit.hpp -- host circuit runtime: Wire, Gate, GateEvalParams, Circuit.
//
// Keeps the reference's driver API (src/wire.h:48-69, src/gate.h:50-80, src/circuit.h:54-114)
// with two deliberate differences:
//   * `CipherText` is a slot of the engine's device-resident LWE pool (include/bce_gpu.h)
//     instead of lbcrypto::LWECiphertext (src/wire.h:46);
//   * the DAG is indexed (gate ids, CSR fan-out, ready counters) and levelised once, and the
//     executor hands each ready frontier to bce_eval_gates() instead of one OpenMP task per
//     gate (src/circuit.cpp:698-710).  The O(G^2) netlist build (src/circuit.cpp:323-354) and
//     the O(wires x waiting gates) manager scan (src/circuit.cpp:593-677) are not inherited.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/bce_circuit.h"
#include "../../include/bce_gpu.h"

namespace bce {

using NameList = std::vector<std::string>;
using CipherText = uint32_t;  // pool slot (reference: lbcrypto::LWECiphertext)
constexpr CipherText kNoCipherText = 0xFFFFFFFFu;

// src/wire.h:48-69
class Wire {
public:
    void setName(const std::string& n) { name = n; }
    std::string getName() const { return name; }
    void setValue(bool b) { value = b; }
    bool getValue() const { return value; }
    void setFanoutGates(const NameList& f) { fanoutGates = f; }
    NameList getFanoutGates() const { return fanoutGates; }
    unsigned int getNumberFanoutGates() const { return (unsigned int)fanoutGates.size(); }
    void setCipherText(CipherText c) { ct = c; }
    CipherText getCipherText() const { return ct; }
    // erases the first match, or reports an error (src/wire.cpp:54-65)
    void updateFanoutGates(const std::string& gateToRemove);

private:
    std::string name;
    NameList fanoutGates;
    bool value = false;
    CipherText ct = kNoCipherText;
};

using ReadyList = std::vector<bool>;
using CipherTextList = std::vector<CipherText>;
using BitList = std::vector<unsigned int>;

enum class GateEnum { INPUT, OUTPUT, NOT, AND, OR, XOR, DFF, LUT3, LUT4 };  // src/gate.h:50

// src/gate.h:52-63; `cc` is the engine handle and plays BinFHEContext + secret key
class GateEvalParams {
public:
    bool plaintext_flag = false;
    bool encrypted_flag = false;
    bool verify_flag = false;
    bce_ctx* cc = nullptr;
    bool xor_fast = false;            // opt-in: XOR as ONE bootstrap of 2*(ct1-ct2) (BCE_XOR_FAST), not the reference's 3
    int encrypt_mode = BCE_BOOTSTRAPPED;  // output mode of the verify-mode re-encryptions (cc.Encrypt(sk, bit), src/gate.cpp:118,...)
    uint64_t* enc_counter = nullptr;  // PRNG stream index for verify-mode re-encryptions
    unsigned int* fixes = nullptr;    // counts "Bad <OP> fixing" events
};

// src/gate.h:65-80
class Gate {
public:
    void Reset() {}
    // plaintext logic and/or encrypted logic for this one gate (src/gate.cpp:49-229).
    // Encrypted: encout[0] must hold the destination slot; XOR also needs tmp[0], tmp[1].
    void Evaluate(const GateEvalParams&);
    std::string name;
    GateEnum op = GateEnum::INPUT;
    NameList inWireNames;
    ReadyList ready;
    NameList outWireNames;
    CipherTextList encin;
    BitList plainin;
    CipherTextList encout;
    BitList plainout;
    CipherTextList tmp;  // scratch slots for the XOR expansion
};

using Inputs = std::vector<std::vector<unsigned int>>;
using Outputs = std::vector<std::vector<unsigned int>>;
using NetList = std::map<std::string, NameList>;

class Circuit {
public:
    // Circuit(set, method), src/circuit.cpp:45-98: creates the engine on device 0 and generates
    // keys.  Accepts only TOY / STD128_OPT and AP / GINX like the reference; throws otherwise.
    Circuit(int paramset, int method);
    // shares an existing engine (keys already generated); nullptr = plaintext-only circuit
    explicit Circuit(bce_ctx* engine);
    ~Circuit();

    bool ReadFile(const std::string& cktName);                 // src/circuit.cpp:102-366
    bool ReadBristol(const std::string& path, bool new_flag);  // direct netlist -> DAG
    void Reset();                                              // src/circuit.cpp:368-419
    void SetInput(const Inputs& input, bool verbose = false);  // src/circuit.cpp:455-530
    void SetInput(unsigned instance, const Inputs& input, bool verbose = false);
    void setPlaintext(bool b) { plaintext_flag = gep.plaintext_flag = b; }
    bool getPlaintext() const { return plaintext_flag; }
    void setEncrypted(bool b) { encrypted_flag = gep.encrypted_flag = b; }
    bool getEncrypted() const { return encrypted_flag; }
    void setVerify(bool b);                                    // src/circuit.cpp:833-840
    bool getVerify() const { return verify_flag; }
    Outputs Clock();                                           // src/circuit.cpp:532-573
    void dumpNetList() const;
    void dumpGates() const;
    void dumpGateCount() const;

    // ---- extensions --------------------------------------------------------------
    void setInstances(unsigned k);            // K lock-step input sets (before SetInput)
    unsigned getInstances() const { return instances_; }
    void setBatched(bool b) { batched_ = b; }
    // cc.Encrypt(sk, bit) of SetInput (src/circuit.cpp:506) and of the verify-mode repairs (src/gate.cpp:118,139,143,158,179,211):
    // BCE_BOOTSTRAPPED by default, as OpenFHE v1.0.x's Encrypt defaults to BOOTSTRAPPED (one Bootstrap per fresh ciphertext);
    // BCE_FRESH is the opt-in that skips it
    void setEncryptMode(int mode) { encrypt_mode_ = gep.encrypt_mode = mode; }
    int getEncryptMode() const { return encrypt_mode_; }
    // opt-in, NOT the reference's semantics: evaluate XOR natively with OpenFHE's XOR_FAST gate
    // (src/gate.cpp:194-196 disables it "for now" because of its higher failure rate)
    void setXorFast(bool b);
    // opt-in: schedule by BOOTSTRAP depth instead of gate level -- NOT gates are folded into their
    // consumers' prep (neg flags) and an XOR's OR shares a launch with the next level's ANDs.  Same
    // ciphertexts as the level schedule (EvalNOT is deterministic), fewer dependent launches.
    void setRelevel(bool b) { relevel_ = b; }
    bool getRelevel() const { return relevel_; }
    // the bootstrap-depth schedule fills its steps by slack up to the launch staircase of the engine (default on; see
    // buildRelevelPlan).  lone / full = 0: ask the engine (bce_launch_capacity), else use these capacities (tests).
    void setBalance(bool on, uint32_t lone = 0, uint32_t full = 0);
    // opt-in: hand the engine the whole bootstrap DAG (bce_dag_*): ONE persistent launch per Clock() in which a finished
    // bootstrap releases its consumers on the device -- the ready-gate rule of the reference's manager
    // (src/circuit.cpp:575-683) applied per gate instead of per frontier.  Same ciphertexts in every register as the
    // other schedules.  XOR temporaries get slots of their own (SSA), so it must be chosen before SetInput; it is
    // ignored (the bootstrap-depth schedule runs) in verify mode, under gate sharding and for parameter classes
    // without the persistent kernel.
    void setDataflow(bool b);
    bool getDataflow() const { return dataflow_; }
    // opt-in: replay the bootstrap-depth schedule's launches as ONE hipGraph per Clock() (bce_plan_run) instead of one
    // host call per step.  Same ciphertexts.  Not with gate sharding (the per-step exchange is a host call) or verify mode.
    void setGraph(bool b) { graph_ = b; }
    bool getGraph() const { return graph_; }
    bool graphActive() const;
    bool dataflowActive() const;
    const std::vector<bce_gate_desc>& dataflowTasks() const { return dag_tasks_; }
    const std::vector<uint8_t>& dataflowPriorities() const { return dag_prio_; }
    bool getBalance() const { return balance_; }
    std::vector<uint32_t> relevelStepSizes() const;          // bootstraps per step, one instance
    std::vector<uint32_t> relevelPublications() const;       // registers this rank publishes per step (gate sharding)
    bool checkRelevelPlan(std::string* why = nullptr) const; // every step reads only what earlier steps wrote
    bool getXorFast() const { return xor_fast_; }
    void setQuiet(bool q) { quiet_ = q; }
    // re-arm for another Clock() on the SAME inputs: keeps mode flags and the input ciphertexts
    // already resident in the device pool (registers are SSA, inputs are never overwritten)
    void Rearm();
    Outputs getOutputs(unsigned instance) const;
    void getCounts(uint32_t out[6]) const;
    const bce_circuit_stats& stats() const { return stats_; }
    bce_circuit_info info() const;
    const std::vector<unsigned>& inputBusBits() const { return n_in_bits_; }
    const std::vector<unsigned>& outputBusBits() const { return out_bus_bits_; }
    void setExchange(uint32_t rank, uint32_t world, int shard_mode, bce_allgather_fn fn, void* user, void* host_send,
                     void* host_recv, void* dev_send, void* dev_recv, uint64_t capacity);
    uint64_t exchangeCapacity(uint32_t world, int shard_mode, bool encrypted) const;
    void enableRccl(bool on) { rccl_ = on; }
    // gate sharding: place a unit on the rank that produced (most of) its inputs, within the fair share of each step
    // (default); off = contiguous split in netlist order.  Every rank must choose the same.
    void setShardLocality(bool on) { shard_locality_ = on; if (world_ > 1) rebuildRelevel(); }
    bool getShardLocality() const { return shard_locality_; }
    // digest of everything the ranks of a gate-sharded run must agree on (owners, publications per step, slot stride,
    // instances): ranks whose devices or environment knobs differ would otherwise build different plans and exchange
    // buffers of different sizes -- compare it across ranks before the first Clock()
    uint64_t planHash() const;
    bce_ctx* engine() const { return cc; }

private:
    struct GateRec {
        GateEnum op;
        int nin;
        int in[2];
        int out;      // wire id, -1 for OUTPUT
        int out_bit;  // OUTPUT only
        std::string name;
    };
    struct LoadRec { unsigned bus, bit; int wire; std::string name; };
    struct ConstRec { int wire; unsigned value; };  // Bristol Fashion EQ: a public constant, no gate
    struct Level {
        std::vector<int> gates;  // file order
        uint32_t n_xor = 0;
    };

    bce_ctx* cc = nullptr;
    bool owns_engine_ = false;
    bool plaintext_flag = false, encrypted_flag = false, verify_flag = false;
    bool done = false, inputs_set_ = false, quiet_ = false, batched_ = true, xor_fast_ = false;
    int encrypt_mode_ = BCE_BOOTSTRAPPED;
    GateEvalParams gep;

    std::vector<LoadRec> inputGates;  // LOADs
    std::vector<ConstRec> constWires_; // constants (live from the start, like inputs)
    std::vector<GateRec> allGates;    // everything else, file order
    std::vector<std::string> wire_names_;
    std::map<uint32_t, int> wire_of_reg_;
    std::vector<uint32_t> fan_off_, fan_gate_;  // CSR wire -> consumer gates
    std::vector<Level> levels_;
    std::vector<int> gate_level_;
    uint32_t max_level_xor_ = 0, stride_ = 0;
    unsigned n_outputs = 1;
    std::vector<unsigned> n_output_bits;
    std::vector<unsigned> n_in_bits_;      // width of every input bus (In1, In2, ...)
    std::vector<unsigned> out_bus_bits_;   // width of every output bus; output bit indices run over their concatenation
    unsigned n_buses_ = 0;

    unsigned instances_ = 1;
    std::vector<std::vector<uint8_t>> plain_;       // [instance][wire]
    std::vector<std::vector<uint8_t>> circuitOut;   // [instance][bit]
    uint64_t enc_counter_ = 0, epoch_ = 0;
    unsigned n_input_gates = 0, n_output_gates = 0, n_and_gates = 0, n_or_gates = 0, n_xor_gates = 0, n_not_gates = 0;
    bce_circuit_stats stats_{};
    std::string err_;

    // multi-rank
    uint32_t rank_ = 0, world_ = 1;
    int shard_mode_ = 0;
    bce_allgather_fn xfn_ = nullptr;
    void* xuser_ = nullptr;
    void *host_send_ = nullptr, *host_recv_ = nullptr, *dev_send_ = nullptr, *dev_recv_ = nullptr;
    uint64_t xcap_ = 0;
    bool rccl_ = false;  // device payloads through bce_rccl_allgather on the engine stream (no host sync, no callback)
    bool shard_locality_ = true;  // gate sharding on the bootstrap-depth schedule: a unit goes to the rank that produced its inputs
    std::vector<std::vector<uint8_t>> owner_;                 // [level][k] owner rank of levels_[level].gates[k]
    std::vector<std::vector<std::vector<int>>> xwires_;       // [level][rank] -> wires that rank must publish

    int addWire(uint32_t reg);
    int wireOf(uint32_t reg, const char* what, unsigned lineNo) const;
    void finalizeNetlist();
    void buildShardPlan();
    void instanceRange(unsigned& lo, unsigned& hi) const;
    struct RStep { std::vector<bce_gate_desc> descs; };
    bool relevel_ = true;   // the bootstrap-depth schedule is the default since round 4 (identical registers, 416 instead of 496
                            // dependent launches on AES-expanded); setRelevel(false) = the reference's gate-level rounds, src/circuit.cpp:532-573
    std::vector<RStep> relevel_plan_;          // bootstrap-depth schedule (built lazily)
    std::vector<bce_gate_desc> relevel_nots_;  // NOT wires that OUTPUT gates read: materialised at the end
    uint32_t relevel_stride_ = 0, base_stride_ = 0, relevel_K_ = 0;
    bool balance_ = true;
    uint32_t cap_lone_ = 0, cap_full_ = 0;
    void launchCapacity(uint32_t& lone, uint32_t& full) const;
    struct Unit { uint32_t asap, start; uint8_t lat, owner; bce_gate_desc d; int32_t p0, p1; };   // XOR (lat 2): d holds (in0, in1, out, n0, n1)
    uint32_t buildUnits(std::vector<Unit>& units, std::vector<int>& base, std::vector<uint8_t>& neg) const;
    static void unitSuccessorsAlap(const std::vector<Unit>& units, uint32_t D, std::vector<uint32_t>& soff, std::vector<uint32_t>& succ,
                                   std::vector<uint32_t>& alap);
    bool dataflow_ = false;
    bce_dag* dag_ = nullptr;
    bool graph_ = false;
    bce_plan* plan_ = nullptr;                 // the schedule's descriptors resident on the device (and its captured graph)
    uint32_t plan_lo_ = 0, plan_K_ = 0, plan_stride_ = 0;
    void dropPlan();
    std::vector<bce_gate_desc> dag_tasks_;
    std::vector<uint8_t> dag_prio_;
    uint32_t dag_stride_ = 0;
    void buildDagTasks();
    void dropDag();
    void clockDataflow();
    void finishReleveled(unsigned lo, unsigned hi);
    void buildRelevelPlan();
    void rebuildRelevel();
    void clockReleveled();
    void managerRound(size_t level);
    void executeRound(size_t level);
    void exchangeLevel(size_t level);
    void exchangeWires(const std::vector<std::vector<int>>& pub);
    std::vector<std::vector<std::vector<int>>> relevel_xw_;  // [step][rank] -> wires that rank publishes after the step (gate sharding)
    void gatherOutputs();
    void requireEngine(const char* what) const;
    void ck(int rc, const char* what) const;
    friend struct ::bce_circuit;
};

}  // namespace bce
