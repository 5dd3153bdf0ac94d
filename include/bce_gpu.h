/*
 * bce_gpu.h -- C ABI of the MI355X gate-bootstrapping engine (libbce_amd.so).
 *
 * This is the drop-in boundary for the ONE hot path of
 * openfheorg/openfhe-boolean-circuit-evaluator: everything the reference
 * reaches through `lbcrypto::BinFHEContext` (OpenFHE).  Each entry point cites
 * the reference interface it replaces (paths relative to the reference root).
 * Plain C types only; no exception crosses this boundary: every call returns a
 * bce_status and bce_last_error() holds the message.
 *
 * The per-gate call `cc.EvalBinGate(gate, ct1, ct2)` made inside one OpenMP
 * task per gate (src/circuit.cpp:698-710 -> src/gate.cpp:133,172,200-202)
 * becomes ONE call bce_eval_gates() for the whole ready-gate frontier; LWE
 * ciphertexts live in a device-resident pool and are named by slot index
 * (the reference's `CipherText` alias, src/wire.h:46).
 */
#ifndef BCE_GPU_H
#define BCE_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bce_ctx bce_ctx;

/* lbcrypto::BINFHE_PARAMSET as used at src/utils.cpp:167-172, src/circuit.cpp:69-78
 * (the reference accepts TOY and STD128_OPT; the rest are OpenFHE's table rows). */
enum bce_paramset {
    BCE_TOY = 0, BCE_MEDIUM = 1, BCE_STD128_AP = 2, BCE_STD128_APOPT = 3, BCE_STD128 = 4,
    BCE_STD128_OPT = 5, BCE_STD192 = 6, BCE_STD192_OPT = 7, BCE_STD256 = 8, BCE_STD256_OPT = 9
};
/* lbcrypto::BINFHE_METHOD, src/utils.cpp:180-185 */
enum bce_method { BCE_AP = 1, BCE_GINX = 2 };
/* lbcrypto::BINGATE for EvalBinGate (src/gate.cpp:133,172) + the unary ops the driver issues */
enum bce_op {
    BCE_OR = 0, BCE_AND = 1, BCE_NOR = 2, BCE_NAND = 3, BCE_XOR_FAST = 4, BCE_XNOR_FAST = 5,
    BCE_OP_NOT = 16,     /* EvalNOT, src/gate.cpp:112,198-199 : no bootstrap      */
    BCE_OP_REFRESH = 17, /* Bootstrap(ct) as run by Encrypt(BOOTSTRAPPED)         */
    BCE_OP_COPY = 18     /* OUTPUT gate copy, src/gate.cpp:90-94                  */
};
enum bce_status {
    BCE_OK = 0, BCE_ERR_ARG = 1, BCE_ERR_NO_DEVICE = 2, BCE_ERR_HIP = 3, BCE_ERR_NO_KEYS = 4,
    BCE_ERR_POOL = 5, BCE_ERR_UNSUPPORTED = 6, BCE_ERR_STATE = 7
};
/* lbcrypto::BINFHE_OUTPUT of BinFHEContext::Encrypt (v1.0.x default is BOOTSTRAPPED) */
enum bce_encrypt_mode { BCE_FRESH = 0, BCE_BOOTSTRAPPED = 1 };

/* parameter block (all u64), same order as the oracle's */
enum { BCE_P_n = 0, BCE_P_N, BCE_P_q, BCE_P_Q, BCE_P_qKS, BCE_P_baseKS, BCE_P_dKS, BCE_P_baseG, BCE_P_dG,
       BCE_P_baseR, BCE_P_dR, BCE_P_method, BCE_P_psi, BCE_P_COUNT };

/* One gate of the frontier.  `neg0/neg1` fold EvalNOT of an input into the
 * bootstrap's LWE prep, which is how XOR = (a AND !b) OR (!a AND b)
 * (src/gate.cpp:198-202) is issued without materialising the NOTs. */
typedef struct bce_gate_desc {
    uint32_t op;   /* enum bce_op */
    uint32_t in0;  /* pool slot */
    uint32_t in1;  /* pool slot (ignored by 1-input ops) */
    uint32_t out;  /* pool slot */
    uint32_t neg0;
    uint32_t neg1;
} bce_gate_desc;

/* which blind-rotation kernel a launch used (chosen by parameter set and launch size) */
enum bce_br_kernel {
    BCE_BR_WAVE_PER_TRANSFORM = 0, /* k_blind_rotate: one wave per (inverse) transform, up to 3 workgroups / CU */
    BCE_BR_SPLIT_X1 = 1,           /* k_blind_rotate_lat<4,2>: split inverse transform, launches of <= #CU workgroups */
    BCE_BR_SPLIT_X2 = 2,           /* k_blind_rotate_lat<4,4>: same, register budget for two workgroups per CU */
    BCE_BR_WORD64 = 3,             /* k_blind_rotate64: ring modulus >= 2^28 */
    BCE_BR_DAG = 4,                /* k_bootstrap_dag: one persistent launch per bce_dag_run (tail fused) */
    BCE_BR_GRAPH = 5,              /* bce_plan_run: the launches of a whole step schedule replayed as one hipGraph (timed as one) */
    BCE_BR_KERNELS = 6
};

/* cumulative device timing, measured with HIP events on the engine's stream */
typedef struct bce_timing {
    double   blind_rotate_ms;   /* sum over launches of the blind-rotation kernels */
    double   tail_ms;           /* extract + modswitch + keyswitch + modswitch    */
    uint64_t blind_rotate_launches;
    uint64_t bootstraps;        /* gate-bootstraps executed (NOT/COPY not counted) */
    /* the same three blind-rotation figures per kernel (index: enum bce_br_kernel) */
    double   br_ms[BCE_BR_KERNELS];
    uint64_t br_launches[BCE_BR_KERNELS];
    uint64_t br_bootstraps[BCE_BR_KERNELS];
    uint64_t fused_tail_launches; /* launches whose blind-rotation kernel also ran the tail in its epilogue
                                     (their tail time is inside blind_rotate_ms, not tail_ms) */
} bce_timing;

/* ---- context ----------------------------------------------------------- */
/* BinFHEContext() + GenerateBinFHEContext(set, method): src/circuit.cpp:65,88.
 * device = HIP device ordinal.  Fails with BCE_ERR_NO_DEVICE when no GPU is
 * visible: there is no CPU fallback in the product. */
int bce_ctx_create(int paramset, int method, int device, bce_ctx** out);
int bce_ctx_create_custom(uint32_t n, uint32_t N, uint64_t q, uint64_t Q, uint64_t qKS, uint32_t baseKS,
                          uint32_t baseG, uint32_t baseR, int method, int device, bce_ctx** out);
void bce_ctx_destroy(bce_ctx*);
/* message of the last failing call on this ctx (ctx may be NULL: last create failure) */
const char* bce_last_error(const bce_ctx*);
int bce_get_params(const bce_ctx*, uint64_t out[BCE_P_COUNT]);

/* ---- keys -------------------------------------------------------------- */
/* sk = cc.KeyGen(); cc.BTKeyGen(sk): src/circuit.cpp:90-91.
 * seed == NULL (what a drop-in caller passes): the 32-byte seed comes from the
 * operating system's entropy pool, as OpenFHE's KeyGen draws from its own
 * entropy-seeded PRNG.  An explicit seed makes the keys a deterministic function
 * of it (ChaCha20 streams, DESIGN.md "PRNG spec") -- for the oracle parity tests
 * and for replicating one key set on every rank of a multi-GPU run; whoever passes
 * one owns its secrecy.  RGSW rows are transformed and multiplied by the ring key
 * on the device. */
int bce_keygen(bce_ctx*, const uint8_t seed[32]);
/* canonical exchange format = coefficient domain, u64 words (what an OpenFHE
 * export would be converted to): s[n], z[N] in {-1,0,1}; bsk [i][key][row][col][N]
 * (GINX) or [i][v][k][row][col][N] (AP); ksk [i][v][j][n+1] mod qKS. */
int bce_import_keys(bce_ctx*, const int32_t* s, const int32_t* z, const uint64_t* bsk, uint64_t bsk_words,
                    const uint32_t* ksk, uint64_t ksk_words);
/* Same, with the bootstrapping key already in EVALUATION form as OpenFHE holds it after BTKeyGen: every polynomial in
 * the bit-reversed order of OpenFHE's Cooley-Tukey forward transform (transformnat-impl.h) for psi = the minimal
 * primitive 2N-th root of unity mod Q (bce_get_params: BCE_P_psi; 455622 / 341565 / 13167220 for the tabulated sets) --
 * the engine's own order, so an OpenFHE-side dump needs no SetFormat(COEFFICIENT) pass over the key (12.9 GB for
 * STD192 / AP) and the import needs no transform.  bce_export_bsk_eval is its inverse.
 * PARITY UNPINNED: that this IS OpenFHE's order is recalled, not checked against OpenFHE (absent here) -- it is pinned only
 * against the oracle's own transform.  The gate-vector file of tools/openfhe_export carries NTT records (a polynomial before
 * and after OpenFHE's SetFormat(EVALUATION)) and compare.py checks them with bce_debug_ntt: use this entry point with
 * OpenFHE-made keys only after that check has passed; the coefficient-form import does not depend on the order. */
int bce_import_keys_eval(bce_ctx*, const int32_t* s, const int32_t* z, const uint64_t* bsk_eval, uint64_t bsk_words,
                         const uint32_t* ksk, uint64_t ksk_words);
int bce_export_bsk_eval(bce_ctx*, uint64_t* bsk_eval);
/* The same material from / to a file in the format of tools/openfhe_export/bce_keyfile.h -- what the
 * OpenFHE-side exporter (tools/openfhe_export/export_keys.cpp) writes for the keys of an existing deployment
 * (cc.KeyGen() / cc.BTKeyGen(sk), src/circuit.cpp:90-91).  Parameters in the file must match the context. */
int bce_import_keys_file(bce_ctx*, const char* path);
int bce_export_keys_file(bce_ctx*, const char* path);
uint64_t bce_bsk_words(const bce_ctx*);
uint64_t bce_ksk_words(const bce_ctx*);
int bce_export_sk(const bce_ctx*, int32_t* s, int32_t* z);
int bce_export_bsk(bce_ctx*, uint64_t* bsk);
int bce_export_ksk(bce_ctx*, uint32_t* ksk);

/* ---- device LWE pool --------------------------------------------------- */
/* LWECiphertext objects (src/wire.h:46, src/gate.h:75-78) become slots. */
int bce_pool_reserve(bce_ctx*, uint32_t slots);
uint32_t bce_pool_slots(const bce_ctx*);
/* ciphertext = u64[n+1] (a[0..n), b), the reference's NativeInteger layout */
int bce_lwe_write(bce_ctx*, const uint32_t* slots, uint32_t count, const uint64_t* cts);
int bce_lwe_read(bce_ctx*, const uint32_t* slots, uint32_t count, uint64_t* cts);

/* cc.Encrypt(sk, bit): src/circuit.cpp:506, src/gate.cpp:118,139,143,158,179,211.
 * The mask a and the noise e of every ciphertext come from a ChaCha20 stream keyed
 * with the context's ENCRYPTION seed, which is independent of the key seed: it is
 * drawn from OS entropy when the context is created, and the stream number is a
 * per-context counter that only moves forward -- `enc_index_base` is then IGNORED,
 * so no (a, e) pair can repeat whatever the caller passes (also after
 * bce_import_keys).  BCE_BOOTSTRAPPED also runs Bootstrap(ct) on the device as
 * OpenFHE v1.0.x does by default. */
int bce_encrypt_bits(bce_ctx*, const uint8_t* bits, const uint32_t* slots, uint32_t count,
                     uint64_t enc_index_base, int mode);
/* Test / replication mode of bce_encrypt_bits: with an explicit 32-byte seed,
 * ciphertext k of a call uses stream `enc_index_base + k` of that seed, so that
 * (1) the oracle reproduces the ciphertext bit for bit and (2) every rank of a
 * gate-sharded multi-GPU run encrypts identical inputs.  The caller then owns the
 * uniqueness of the indices: reusing one reuses (a, e).  seed == NULL returns to
 * the default (fresh OS entropy, internal counter). */
int bce_set_encrypt_seed(bce_ctx*, const uint8_t seed[32]);
/* cc.Decrypt(sk, ct, &res): src/circuit.cpp:800, src/gate.cpp:72,97,115,... (res in 0..3) */
int bce_decrypt_bits(bce_ctx*, const uint32_t* slots, uint32_t count, uint8_t* bits);

/* ---- the hot path ------------------------------------------------------ */
/* Batched EvalBinGate / EvalNOT over one ready frontier:
 * src/circuit.cpp:698-710 + src/gate.cpp:112,133,146,172,198-202.
 * All descriptors of one call are independent (no `out` is an input of the
 * same call).  Asynchronous on the engine's stream; results are ordered with
 * later calls; bce_synchronize() waits. */
int bce_eval_gates(bce_ctx*, uint32_t n_desc, const bce_gate_desc* descs);
/* Same frontier applied to `instances` independent input sets laid out
 * `slot_stride` slots apart (lock-step evaluation of K circuit instances,
 * the reference's sequential numTestLoops loop, src/test_aes.cpp:179-182). */
int bce_eval_gates_strided(bce_ctx*, uint32_t n_desc, const bce_gate_desc* descs, uint32_t instances,
                           uint32_t slot_stride);
int bce_synchronize(bce_ctx*);

/* ---- the hot path, a whole step schedule at once ----------------------------------------------------------
 * A circuit's schedule is the same list of frontiers every time it is clocked (src/circuit.cpp:575-683 finds the same
 * ready gates in the same order).  A bce_plan keeps the descriptors of ALL its dependent steps resident on the device:
 *   bce_plan_run_step  = bce_eval_gates_strided for step `s` without the per-call descriptor upload (same kernels,
 *                        same per-launch timing), the caller walks the steps (and may exchange ciphertexts between them);
 *   bce_plan_run       = every step's launches captured once into a hipGraph and replayed with one hipGraphLaunch:
 *                        no per-step host call, no upload, no event between dependent kernels.  Timed as one unit
 *                        (BCE_BR_GRAPH).  Same ciphertexts in every register as the step-by-step form.
 * Steps hold bootstrapped ops only (BCE_AND .. BCE_XNOR_FAST, BCE_OP_REFRESH); `descs` = the steps' descriptors back to
 * back, step s has step_sizes[s] of them.  instances / slot_stride as in bce_eval_gates_strided; the slots are fixed at
 * creation (slot_base shifts them all, e.g. a rank's first instance). */
typedef struct bce_plan bce_plan;
int bce_plan_create(bce_ctx*, uint32_t n_steps, const uint32_t* step_sizes, const bce_gate_desc* descs,
                    uint32_t instances, uint32_t slot_stride, uint32_t slot_base, bce_plan** out);
int bce_plan_run_step(bce_ctx*, bce_plan*, uint32_t step);
int bce_plan_run(bce_ctx*, bce_plan*);
void bce_plan_destroy(bce_ctx*, bce_plan*);

/* ---- the hot path, dependency-driven ------------------------------------------------------------------
 * The reference alternates Circuit::_ManageGates (a gate is ready when every input wire has arrived,
 * src/circuit.cpp:575-683) with Circuit::_ExecuteGates (src/circuit.cpp:685-817) once per frontier.  A bce_dag
 * hands the engine the WHOLE bootstrap DAG of a circuit instead: `tasks` are bootstrapped gates in topological
 * order, in SSA form (every `out` slot is written by exactly one task and only read by later ones; inputs no
 * task writes are primary inputs).  bce_dag_run evaluates `instances` input sets `slot_stride` slots apart in ONE
 * persistent kernel launch: dependency counters and ready queues live in device memory, a workgroup that finishes a
 * bootstrap releases its consumers, idle workgroups pull the next ready bootstrap -- no kernel boundary and no
 * device-wide barrier between dependent gates.  Every register holds the same ciphertext as under
 * bce_eval_gates_strided called frontier by frontier.  `prio` (may be NULL) gives each task a priority class
 * 0 (most urgent) .. 3; ready tasks of a lower class number are pulled first.
 * Asynchronous like bce_eval_gates; a run in which the device scheduler made no progress for the stall limit
 * (bce_dag_set_limits) abandons its queues and the next synchronising call returns BCE_ERR_STATE. */
typedef struct bce_dag bce_dag;
int bce_dag_supported(const bce_ctx*);   /* 1 when this context's parameter class has the persistent kernel */
int bce_dag_create(bce_ctx*, uint32_t n_tasks, const bce_gate_desc* tasks, const uint8_t* prio, bce_dag** out);
/* instance k of a run uses the DAG's slot numbers shifted by slot_base + k * slot_stride */
int bce_dag_run(bce_ctx*, bce_dag*, uint32_t instances, uint32_t slot_stride, uint32_t slot_base);
void bce_dag_destroy(bce_ctx*, bce_dag*);
/* workgroups_per_cu: 0 = choose by the work per dependency level, 1 or 2 = force; placement: 1 = idle compute
 * units claim ready bootstraps first (default), 0 = first poller wins; lazy_us: how long a half-busy compute unit
 * leaves a short queue to idle ones; stall_ms: no completion anywhere for this long abandons the run. */
int bce_dag_set_limits(bce_ctx*, int workgroups_per_cu, int placement, uint32_t lazy_us, uint32_t stall_ms);
/* counters of the last finished bce_dag_run (after a synchronising call): out[0] bootstraps completed,
 * out[1] claims a half-busy compute unit delayed in favour of an idle one, out[2] abort code (0 = none),
 * out[3] workgroups per CU the run used, out[4] / out[5] 100 MHz ticks summed over all workgroups between claiming
 * a bootstrap and releasing its consumers / spent looking for a bootstrap they then got, out[6] ticks spent waiting at the
 * XCD start gates (inside out[4]) */
int bce_dag_last_run(bce_ctx*, uint64_t out[7]);
/* test hook: task `t` of the DAG waits for one producer more than it has, so it never becomes ready and the run
 * stalls (exercises the bounded-spin exit) */
int bce_dag_debug_block_task(bce_dag*, uint32_t t);

/* ---- measurement ------------------------------------------------------- */
int bce_timing_reset(bce_ctx*);
int bce_timing_get(bce_ctx*, bce_timing* out); /* synchronizes */
/* on (default): every blind-rotation / tail launch carries a start and a stop timestamp ON ITS OWN DISPATCH (hipExtLaunchKernel
 * events: the *_ms fields are the kernels' own durations, nothing is recorded between dependent kernels; 11 us per dependent step
 * on the device timeline against 15.6 us with events recorded around the launches, as rounds 1-2 did).  off: launches and
 * bootstraps are still counted, the *_ms fields stop growing, plain launches.  bce_dag_run / bce_plan_run keep their one pair. */
int bce_timing_set_events(bce_ctx*, int on);
/* algorithmic bytes one gate-bootstrap reads at the widths this engine ships
 * (SURVEY.md 8(d) formula with w_bsk, w_ks, w_ct of this build) */
uint64_t bce_bytes_per_bootstrap(const bce_ctx*);
/* the same figure split by the kernel that moves the bytes: out[0] = bootstrapping-key rows (blind rotation),
 * out[1] = key-switching-key rows (tail gather), out[2] = ciphertext input / output words */
int bce_bytes_per_bootstrap_parts(const bce_ctx*, uint64_t out[3]);
/* forward transforms one AddToAcc step runs: 2 dG as in the reference (rgsw-acc-cggi.cpp / rgsw-acc-dm.cpp AddToAcc),
 * or 2 dG - 2 when this context keeps its key with the lowest gadget digit folded in (same accumulator words; the
 * decomposition is exact for the parameter set and the digit-0 rows multiply the accumulator itself) */
uint32_t bce_forward_transforms_per_step(const bce_ctx*);
/* Launch granularity of this context's blind-rotation kernels, for callers that shape their frontiers (the host
 * scheduler of bce_circuit.h does): one bootstrap is one workgroup, so a call's time is a staircase in its size.
 * *lone = bootstraps up to which every workgroup has a compute unit to itself (one bootstrap latency per call),
 * *full = bootstraps resident at once when the device is saturated (each further multiple costs one more round). */
int bce_launch_capacity(const bce_ctx*, uint32_t* lone, uint32_t* full);

/* ---- in-library collective (multi-GPU, one process per GPU) ---------------------------------------------
 * RCCL all-gather issued on the engine's own stream, so that the exchange of boundary ciphertexts between two
 * dependent frontiers needs neither a host synchronisation nor the host language (bce_circuit_enable_rccl uses
 * it).  RCCL is dlopen()ed on first use.  Rendezvous: rank 0 calls bce_rccl_unique_id and the host program
 * delivers the 128 bytes to every rank, each of which then calls bce_rccl_init with its rank. */
int bce_rccl_available(void);   /* 1 when the RCCL library and the entry points used here could be loaded */
int bce_rccl_version(void);     /* ncclGetVersion of that library (2.x ABI required), 0 if none.  NOTE: between two or more
                                 * devices this path has not run on hardware yet (no multi-GPU node was available to the
                                 * builder); the torch.distributed callback of bce_circuit_set_exchange is the verified one. */
int bce_rccl_unique_id(uint8_t out[128]);
int bce_rccl_init(bce_ctx*, const uint8_t uid[128], int rank, int world);
/* every rank contributes `bytes` bytes at dev_send and receives world * bytes at dev_recv, rank-major; asynchronous,
 * ordered with the engine's kernels */
int bce_rccl_allgather(bce_ctx*, const void* dev_send, void* dev_recv, uint64_t bytes);
int bce_rccl_shutdown(bce_ctx*);
/* what the context's communicator reports: out[0] = ncclCommCount (ranks RCCL sees), out[1] = ncclCommUserRank,
 * out[2] = ncclCommCuDevice.  Lets a caller (bench.py, tests) prove what the exchange collective spans. */
int bce_rccl_comm_info(bce_ctx*, int out[3]);

/* ---- staged outputs for parity tests ----------------------------------- */
/* Runs the frontier like bce_eval_gates and also returns the intermediates
 * (any pointer may be NULL): acc = accumulator after blind rotation,
 * COEFFICIENT domain, u64 [n_desc][2][N]; lweN = after extract + ModSwitch(Q->qKS),
 * u64 [n_desc][N+1]; ks = after KeySwitch, u64 [n_desc][n+1].  Bootstrapped ops only. */
int bce_debug_eval_stages(bce_ctx*, uint32_t n_desc, const bce_gate_desc* descs, uint64_t* acc,
                          uint64_t* lweN, uint64_t* ks);
/* The tail of EvalBinGate alone on caller-supplied accumulators: acc = u64 [count][2][N], COEFFICIENT domain, words < Q
 * (what `acc` of bce_debug_eval_stages returns).  Runs transpose + extract (a = coefficients of acc[0](X^-1),
 * b = acc[1][0] + Q/8 + 1), ModSwitch(Q -> qKS), KeySwitch, ModSwitch(qKS -> q) -- lbcrypto::LWEEncryptionScheme::
 * ModSwitch / KeySwitch as BinFHEScheme::EvalBinGate calls them after the accumulator -- and writes the refreshed
 * ciphertexts to pool slots out_slots[count]; lweN / ks (may be NULL) as in bce_debug_eval_stages.  Used by
 * tools/openfhe_export/compare.py to replay OpenFHE's own tail records (kind BCE_GATEVEC_TAIL of bce_keyfile.h). */
int bce_debug_tail(bce_ctx*, uint32_t count, const uint64_t* acc, const uint32_t* out_slots, uint64_t* lweN,
                   uint64_t* ks);
/* forward (inverse=0) / inverse negacyclic NTT of `count` polys, in place, u64 [count][N];
 * forward output is in the engine's internal evaluation order. */
int bce_debug_ntt(bce_ctx*, uint64_t* polys, uint32_t count, int inverse);

#ifdef __cplusplus
}
#endif
#endif
