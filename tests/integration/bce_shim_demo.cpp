// Option A of INTEGRATION.md as a program: the shim a maintainer of openfhe-boolean-circuit-evaluator would add in place
// of "binfhecontext.h" (src/wire.h:39, src/utils.h:43), driven the way the reference's Circuit::_ExecuteGates
// (src/circuit.cpp:685-817) and Gate::Evaluate (src/gate.cpp:49-216) drive OpenFHE -- but with one batched call per
// stage instead of one EvalBinGate per OpenMP task.  Evaluates a full adder (2 XOR, 2 AND, 1 OR: the gate mix of
// examples/simple_ckts/adder_2bit) on all 8 inputs through the C ABI only.  g++ -std=c++17, links libbce_amd.so; no torch,
// no Python.  Exit codes: 0 = all 8 sums correct, 3 = no GPU visible (the product has no CPU path), 1 = anything else.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <vector>

#include "bce_gpu.h"

using CipherText = uint32_t;                 // was lbcrypto::LWECiphertext   (src/wire.h:46)
struct BceContext {                          // was lbcrypto::BinFHEContext   (src/circuit.h:75, src/gate.h:61)
    bce_ctx* h = nullptr;
    uint32_t next_slot = 0;                  // SSA registers: one fresh slot per produced ciphertext
    int GenerateBinFHEContext(int set, int method) { return bce_ctx_create(set, method, /*device*/ 0, &h); }   // src/circuit.cpp:88
    void KeyGenAndBTKeyGen() {                                                                                  // src/circuit.cpp:90-91
        if (bce_keygen(h, /*seed*/ nullptr) != BCE_OK) throw std::runtime_error(bce_last_error(h));            // NULL: OS entropy, like cc.KeyGen()
    }
    CipherText Encrypt(unsigned bit) {                                                                          // src/circuit.cpp:506
        CipherText s = next_slot++;
        uint8_t b = (uint8_t)bit;
        if (bce_pool_reserve(h, next_slot) != BCE_OK || bce_encrypt_bits(h, &b, &s, 1, 0, BCE_FRESH) != BCE_OK)
            throw std::runtime_error(bce_last_error(h));
        return s;
    }
    unsigned Decrypt(CipherText ct) {                                                                           // src/circuit.cpp:800
        uint8_t b = 0;
        if (bce_decrypt_bits(h, &ct, 1, &b) != BCE_OK) throw std::runtime_error(bce_last_error(h));
        return b;
    }
    ~BceContext() { bce_ctx_destroy(h); }
};

enum class GateEnum { AND, OR, XOR };
struct Gate { GateEnum op; CipherText in0, in1, out; };

// one ready frontier -> stage A (AND / OR gates and both ANDs of every XOR) and stage B (the ORs of the XORs)
static void ExecuteGates(BceContext& cc, std::vector<Gate>& frontier) {
    std::vector<bce_gate_desc> stageA, stageB;
    for (Gate& g : frontier) {
        g.out = cc.next_slot++;
        switch (g.op) {
            case GateEnum::AND: stageA.push_back({BCE_AND, g.in0, g.in1, g.out, 0, 0}); break;
            case GateEnum::OR: stageA.push_back({BCE_OR, g.in0, g.in1, g.out, 0, 0}); break;
            case GateEnum::XOR: {                             // src/gate.cpp:198-202, the two EvalNOTs folded into the prep
                const uint32_t t1 = cc.next_slot++, t2 = cc.next_slot++;
                stageA.push_back({BCE_AND, g.in0, g.in1, t1, 0, 1});
                stageA.push_back({BCE_AND, g.in0, g.in1, t2, 1, 0});
                stageB.push_back({BCE_OR, t1, t2, g.out, 0, 0});
            } break;
        }
    }
    if (bce_pool_reserve(cc.h, cc.next_slot) != BCE_OK) throw std::runtime_error(bce_last_error(cc.h));
    if (bce_eval_gates(cc.h, (uint32_t)stageA.size(), stageA.data()) != BCE_OK) throw std::runtime_error(bce_last_error(cc.h));
    if (bce_eval_gates(cc.h, (uint32_t)stageB.size(), stageB.data()) != BCE_OK) throw std::runtime_error(bce_last_error(cc.h));
}

int main(int argc, char** argv) {
    const int set = argc > 1 ? std::atoi(argv[1]) : BCE_TOY;   // the reference's flag parser accepts TOY and STD128_OPT (src/utils.cpp:167-172)
    BceContext cc;
    const int rc = cc.GenerateBinFHEContext(set, BCE_GINX);
    if (rc == BCE_ERR_NO_DEVICE) { std::printf("no GPU: %s\n", bce_last_error(nullptr)); return 3; }
    if (rc != BCE_OK) { std::printf("context: %s\n", bce_last_error(nullptr)); return 1; }
    try {
        cc.KeyGenAndBTKeyGen();
        int bad = 0;
        for (unsigned v = 0; v < 8; ++v) {
            const unsigned a = v & 1, b = (v >> 1) & 1, cin = v >> 2;
            const CipherText ca = cc.Encrypt(a), cb = cc.Encrypt(b), cc_in = cc.Encrypt(cin);
            // level 1: t = a XOR b, u = a AND b;  level 2: sum = t XOR cin, w = t AND cin;  level 3: cout = u OR w
            std::vector<Gate> l1 = {{GateEnum::XOR, ca, cb, 0}, {GateEnum::AND, ca, cb, 0}};
            ExecuteGates(cc, l1);
            std::vector<Gate> l2 = {{GateEnum::XOR, l1[0].out, cc_in, 0}, {GateEnum::AND, l1[0].out, cc_in, 0}};
            ExecuteGates(cc, l2);
            std::vector<Gate> l3 = {{GateEnum::OR, l1[1].out, l2[1].out, 0}};
            ExecuteGates(cc, l3);
            const unsigned sum = cc.Decrypt(l2[0].out), cout = cc.Decrypt(l3[0].out);
            if (sum + 2 * cout != a + b + cin) { ++bad; std::printf("a=%u b=%u cin=%u -> sum=%u cout=%u\n", a, b, cin, sum, cout); }
        }
        std::printf("full adder through the shim: %d of 8 wrong\n", bad);
        return bad ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
}
