/*
 * bce_circuit.h -- C ABI over the host circuit runtime (C++ classes Circuit / Gate / Wire in
 * openfhe-boolean-circuit-evaluator_amd/csrc/circuit.hpp), the caller side of the hot path.
 *
 * The C++ classes keep the reference's driver API (src/circuit.h:54-73, src/gate.h:50-80,
 * src/wire.h:48-69) and its observable semantics -- one ASAP level of the DAG per manager /
 * executor round, XOR = 2 NOT + 2 AND + 1 OR, the six per-op counters, plaintext /
 * encrypted / verify modes -- but walk an indexed DAG and hand each ready frontier to
 * bce_eval_gates() (include/bce_gpu.h) instead of spawning one OpenMP task per gate
 * (src/circuit.cpp:698-710).  This header exposes them to non-C++ hosts (the Python tests
 * and bench.py bind it with ctypes).
 */
#ifndef BCE_CIRCUIT_H
#define BCE_CIRCUIT_H

#include <stdint.h>

#include "bce_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bce_circuit bce_circuit;

typedef struct bce_circuit_info {
    uint32_t n_gates;        /* all gates except LOADs (src/circuit.cpp allGates)      */
    uint32_t n_input_gates;  /* LOADs                                                    */
    uint32_t n_wires;        /* registers                                                */
    uint32_t n_inputs;       /* number of input buses referenced (In1, In2)              */
    uint32_t n_input_bits[2]; /* the first two buses; all of them: bce_circuit_get_buses()   */
    uint32_t n_output_bits;  /* max STORE index + 1 = all output values concatenated      */
    uint32_t n_levels;       /* Clock() rounds = ASAP levels incl. NOT and OUTPUT levels */
    uint32_t n_sublaunches;  /* dependent bce_eval_gates launches per evaluation          */
    uint32_t n_relevel_steps;/* dependent launches of the opt-in re-levelled schedule     */
    uint32_t max_frontier;   /* widest sub-launch, in bootstraps                          */
    uint32_t slot_stride;    /* pool slots per instance: register r of instance k = slot k*slot_stride + r */
    uint64_t n_bootstraps;   /* per evaluation: AND=1, OR=1, XOR=3, NOT=0                 */
} bce_circuit_info;

typedef struct bce_circuit_stats {
    double   total_ms;       /* "### Total time" of Clock(), src/circuit.cpp:565        */
    double   management_ms;  /* _CircuitManager share                                    */
    double   execution_ms;   /* _ExecuteGates share                                      */
    uint64_t bootstraps;     /* executed by this rank in the last Clock()                */
    uint32_t levels;
    uint32_t sublaunches;
    uint32_t verify_fixes;   /* "Bad <OP> fixing" events (verify mode)                   */
    uint32_t exchanges;      /* multi-rank: number of allgather calls                    */
    uint64_t exchanged_cts;  /* multi-rank: ciphertexts this rank contributed            */
} bce_circuit_stats;

/* allgather callback for multi-rank runs: every rank contributes `bytes` bytes found at the
 * send buffer registered in bce_circuit_set_exchange and receives world*bytes in the
 * registered receive buffer, rank-major.  on_device tells which buffer pair is meant. */
typedef int (*bce_allgather_fn)(void* user, uint64_t bytes, int on_device);

/* Circuit::Circuit(set, method), src/circuit.cpp:45-98, but the engine (keys included) is
 * passed in so that several circuits can share one.  engine == NULL gives a
 * plaintext-only circuit (host logic, no GPU needed); encrypted mode then fails loudly.
 * Lifetime: a circuit uses its engine until it is destroyed, so destroy circuits first.  The other order is tolerated for
 * tear-down only: bce_ctx_destroy releases the device-side schedules (bce_plan / bce_dag) the engine's circuits created, and
 * a circuit destroyed afterwards touches nothing of the engine; any other call on it is an error. */
int bce_circuit_create(bce_ctx* engine, bce_circuit** out);
void bce_circuit_destroy(bce_circuit*);
const char* bce_circuit_last_error(const bce_circuit*);

/* Circuit::ReadFile (assembler text format), src/circuit.cpp:102-366 */
int bce_circuit_read_file(bce_circuit*, const char* path);
/* direct Bristol netlist -> DAG (old: new_flag=0, "Bristol Fashion": new_flag=1) */
int bce_circuit_read_bristol(bce_circuit*, const char* path, int new_flag);
int bce_circuit_get_info(const bce_circuit*, bce_circuit_info* out);

/* Circuit::Reset / setPlaintext / setEncrypted / setVerify, src/circuit.cpp:368-419,819-842 */
int bce_circuit_reset(bce_circuit*);
/* extension: allow another Clock() on the same inputs (mode flags and the input ciphertexts
 * resident in the device pool are kept; counters and the done flag are cleared) */
int bce_circuit_rearm(bce_circuit*);
int bce_circuit_set_plaintext(bce_circuit*, int on);
int bce_circuit_set_encrypted(bce_circuit*, int on);
int bce_circuit_set_verify(bce_circuit*, int on);
int bce_circuit_get_flags(const bce_circuit*, int* plaintext, int* encrypted, int* verify);
/* 0: one bce_eval_gates call per gate through Gate::Evaluate (reference shape);
 * 1 (default): one call per frontier stage */
int bce_circuit_set_batched(bce_circuit*, int on);
/* Output mode of cc.Encrypt(sk, bit) as SetInput (src/circuit.cpp:506) and the verify-mode repairs
 * (src/gate.cpp:118,139,143,158,179,211) call it.  Default BCE_BOOTSTRAPPED: OpenFHE v1.0.x's Encrypt defaults to
 * BOOTSTRAPPED, i.e. every fresh ciphertext goes through one Bootstrap (one extra gate-bootstrap per input bit, run as
 * one batched launch per SetInput call here; not counted in the Clock() statistics).  BCE_FRESH skips it. */
int bce_circuit_set_encrypt_mode(bce_circuit*, int mode);
int bce_circuit_get_encrypt_mode(const bce_circuit*);

/* opt-in extension, NOT reference semantics: evaluate XOR gates natively with OpenFHE's XOR_FAST
 * (one bootstrap of 2*(ct1-ct2) instead of NOT,NOT,AND,AND,OR); the reference keeps this disabled
 * because of its higher failure rate (src/gate.cpp:194-203) */
int bce_circuit_set_xor_fast(bce_circuit*, int on);
/* Schedule by bootstrap depth (NOTs folded into consumers, an XOR's OR launched with the next level's ANDs): the
 * same ciphertext in every bootstrapped register as the reference's gate-level rounds -- asserted register by register on
 * AES-expanded, tests/test_gpu_circuit.py -- in fewer dependent launches (AES-expanded: 416 instead of 496, one block
 * 0.82 s instead of 1.02 s).  DEFAULT ON since round 4: a caller that writes the reference's own sequence (Circuit; ReadFile;
 * Reset; setEncrypted; SetInput; Clock -- src/test_aes.cpp:338-343) gets it with slack-filled steps and the schedule's
 * descriptors resident on the device.  on = 0 restores the reference's Clock rounds (src/circuit.cpp:532-573: one round per
 * gate level, NOT gates materialised).  Gate-level rounds also run whenever a plaintext pass rides along (verify mode) or
 * with bce_circuit_set_batched(0).  Registers of NOT gates hold a ciphertext only under the gate-level rounds (or when an
 * OUTPUT gate reads them). */
int bce_circuit_set_relevel(bce_circuit*, int on);
int bce_circuit_get_relevel(bce_circuit*);
/* The bootstrap-depth schedule fills its steps BY SLACK up to the launch staircase of the engine (default on): one
 * bootstrap is one workgroup, so a frontier call costs one bootstrap latency up to `lone` bootstraps and one more round
 * per `full` beyond (bce_launch_capacity); a step holding K x count bootstraps is topped up to the next stair with the
 * ready gates of least slack.  Same number of steps, same ciphertexts (which step a gate runs in does not change its
 * result), fewer half-empty launches.  lone = full = 0: capacities from the engine; explicit values are for tests and
 * engine-less (plaintext) circuits.  Call before SetInput (the XOR temporaries of a step are part of the slot stride). */
int bce_circuit_set_balance(bce_circuit*, int on, uint32_t lone, uint32_t full);
/* Bootstraps per step of the current bootstrap-depth schedule, for ONE instance: writes min(*n_steps, cap) entries and
 * sets *n_steps to the number of steps. */
int bce_circuit_relevel_steps(const bce_circuit*, uint32_t* sizes, uint32_t cap, uint32_t* n_steps);
/* Gate sharding on that schedule: registers THIS rank publishes after each step (one allgather per step, padded to the
 * widest rank).  Same calling convention as bce_circuit_relevel_steps; all zero without gate sharding. */
int bce_circuit_relevel_publications(const bce_circuit*, uint32_t* counts, uint32_t cap, uint32_t* n_steps);
/* Self-check of that schedule: every step reads only registers written by earlier steps (or inputs / constants), every XOR
 * temporary is consumed exactly one step after it is produced, every gate output is written once.  BCE_OK or
 * BCE_ERR_STATE with the finding in bce_circuit_last_error. */
int bce_circuit_check_relevel(bce_circuit*);
/* opt-in extension: the DATAFLOW schedule -- the whole bootstrap DAG goes to the engine as one bce_dag (bce_gpu.h)
 * and one Clock() is ONE persistent kernel launch in which a finished bootstrap releases its consumers on the device:
 * the ready-gate rule of Circuit::_ManageGates (src/circuit.cpp:575-683) applied per gate, not per frontier, with no
 * kernel boundary between dependent gates.  Same ciphertexts in every register.  XOR temporaries get slots of their
 * own, so call it before SetInput.  Ignored under gate sharding and for parameter classes without the persistent kernel
 * (the bootstrap-depth step schedule runs instead); verify mode carries a plaintext pass and therefore runs the reference's
 * gate-level rounds whatever schedule was chosen.  bce_circuit_dataflow_active tells. */
int bce_circuit_set_dataflow(bce_circuit*, int on);
int bce_circuit_dataflow_active(const bce_circuit*);   /* 1 if the next encrypted Clock() takes the dataflow path */
/* Opt-in for the bootstrap-depth schedule (set_relevel): its launches are captured once into a hipGraph and every
 * Clock() replays them with one launch (bce_plan_run) -- no per-step host call between the dependent kernels.  Same
 * ciphertexts.  Without it the same resident descriptors are walked step by step (bce_plan_run_step).  Not with gate
 * sharding (the per-step exchange is a host call), verify mode or the dataflow schedule: bce_circuit_graph_active tells. */
int bce_circuit_set_graph(bce_circuit*, int on);
int bce_circuit_graph_active(const bce_circuit*);
/* The task list of the dataflow schedule (what bce_dag_create receives): gates in topological order with SSA slots for
 * ONE instance (slot < slot_stride of bce_circuit_get_info) and their priority classes.  Writes min(*n_tasks, cap)
 * entries to each non-NULL array and sets *n_tasks to the number of tasks (= bootstraps of one evaluation). */
int bce_circuit_dataflow_plan(const bce_circuit*, bce_gate_desc* tasks, uint8_t* prio, uint32_t cap, uint32_t* n_tasks);
/* K independent input sets evaluated in lock-step (call before SetInput) */
int bce_circuit_set_instances(bce_circuit*, uint32_t k);
/* Circuit::SetInput, src/circuit.cpp:455-530: bits = concatenation of the input buses,
 * LSB first (Inputs[k][bit]); widths[k] = number of bits of bus k */
int bce_circuit_set_input(bce_circuit*, uint32_t instance, const uint32_t* widths, uint32_t n_buses,
                          const uint8_t* bits);
/* Circuit::Clock, src/circuit.cpp:532-573 */
int bce_circuit_clock(bce_circuit*);
/* Outputs of one instance after Clock(): the bits of all output values, concatenated in header order
 * (n_output_bits in total; the reference has a single output bus, Outputs[0][bit]) */
int bce_circuit_get_output(const bce_circuit*, uint32_t instance, uint8_t* bits);
/* widths of every input and output value (Bristol Fashion headers may name any number of either;
 * src/analyze.cpp:129-158 hard-codes 2 / 1).  Arrays may be NULL to query the counts only. */
int bce_circuit_get_buses(const bce_circuit*, uint32_t* n_in, uint32_t* in_widths, uint32_t in_cap, uint32_t* n_out,
                          uint32_t* out_widths, uint32_t out_cap);
/* dumpGateCount counters: input, output, not, and, or, xor (src/circuit.cpp:866-873) */
int bce_circuit_get_counts(const bce_circuit*, uint32_t out[6]);
int bce_circuit_get_stats(const bce_circuit*, bce_circuit_stats* out);
/* dumpNetList / dumpGates to stdout */
int bce_circuit_dump(const bce_circuit*, int what);

/* ---- multi-rank (one process per GPU) ---------------------------------------------- */
/* shard_mode 0: instances are split contiguously over ranks (no exchange until the outputs);
 * shard_mode 1: every level's gates are split over ranks and only ciphertexts whose
 * fan-out crosses ranks are exchanged after the level.
 * host_* / dev_* : buffers owned by the caller (e.g. torch tensors), `capacity` bytes for
 * send and world*capacity for recv; dev_* may be NULL for plaintext-only runs. */
int bce_circuit_set_exchange(bce_circuit*, uint32_t rank, uint32_t world, int shard_mode, bce_allgather_fn fn,
                             void* user, void* host_send, void* host_recv, void* dev_send, void* dev_recv,
                             uint64_t capacity);
/* After bce_circuit_set_exchange: exchange DEVICE payloads (boundary ciphertexts of shard_mode 1) with the
 * in-library RCCL all-gather on the engine stream (bce_rccl_init must have been called on the circuit's engine)
 * instead of the callback -- no host synchronisation and no callback per level.  Host payloads (plaintext bits,
 * final outputs) keep using the callback.  on = 0 returns to the callback for everything. */
int bce_circuit_enable_rccl(bce_circuit*, int on);
/* Gate sharding on the bootstrap-depth schedule: 1 (default) places a unit on the rank that produced most of its inputs,
 * within each rank's fair share of the step (fewer ciphertexts cross ranks); 0 = contiguous split in netlist
 * order.  Every rank must use the same setting (bce_circuit_plan_hash covers it). */
int bce_circuit_set_shard_locality(bce_circuit*, int on);
/* Digest of what the ranks of a sharded run must agree on (gate owners, publications per step, slot stride, instance
 * count).  The plans are built independently on every rank from the device's launch capacity; compare the digests
 * across ranks (dist.Exchange does) before the first Clock(): different plans mean exchanges of different sizes. */
uint64_t bce_circuit_plan_hash(const bce_circuit*);
/* bytes one rank may contribute in the largest exchange of this circuit/instance count */
uint64_t bce_circuit_exchange_capacity(const bce_circuit*, uint32_t world, int shard_mode, int encrypted);

/* ---- Bristol front end (src/analyze.cpp:56, src/assemble.cpp:46) ------------------- */
/* analyze_bristol + assemble_bristol: writes the assembler text file.  out_path NULL means
 * "<input without extension>_FHE.out" like the reference. */
int bce_assemble_bristol(const char* in_path, int new_flag, int gen_fan_flag, int debug_flag,
                         const char* out_path, char* err, uint32_t err_len);

/* ---- engine helpers used by the exchange path --------------------------------------- */
/* pack pool rows (slots) into a dense device buffer [count][n+1] u32 and back */
int bce_pool_gather(bce_ctx*, const uint32_t* slots, uint32_t count, void* dev_dst);
int bce_pool_scatter(bce_ctx*, const uint32_t* slots, uint32_t count, const void* dev_src);

#ifdef __cplusplus
}
#endif
#endif
