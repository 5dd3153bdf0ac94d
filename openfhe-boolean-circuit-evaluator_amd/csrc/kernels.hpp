// kernels.hpp -- launch interface of the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_ext.h>

#include <tuple>

#include <cstdint>

#include "../../include/bce_gpu.h"

namespace bce {

using u32 = uint32_t;
using u64 = uint64_t;

// Everything a kernel needs, passed by value (lives in SGPRs / kernarg).
#ifndef BCE_KEY_NINV
#define BCE_KEY_NINV 1
#endif
struct DevParams {
    u32 n, N, logN;
    u32 q;            // LWE modulus, power of two
    u32 Q;            // ring modulus, prime < 2^28, Q = 1 mod 2N
    u32 qKS;          // key-switch modulus
    u32 baseKS, dKS;
    u32 ksk_stride;   // elements per KSK row (>= n+1, padded)
    u32 ksk_u16;      // 1: rows are uint16_t, 0: uint32_t
    u32 ks_chunk;     // rows whose elements can be summed in 32 bits: floor(2^32 / qKS), a multiple of 8, >= 8
    u32 gBits, dG;    // gadget: base 2^gBits, dG digits; R = 2*dG RGSW rows
    u32 baseR, dR;    // AP only
    u32 method_ap;    // 1: AP/DM accumulator, 0: GINX/CGGI
    u32 factor;       // 2N / q
    u32 Q8p1;         // Q/8 + 1
    u32 red_shift;    // Barrett for x < 2^(2*bitlen(Q)+3): x1 = x >> red_shift
    u32 red_mu;       // floor(2^(32+red_shift) / Q)
    u32 Ninv, Ninv_s; // N^-1 mod Q and its Shoup companion
    u32 Winv_last, Winv_last_s;  // -psi^(N/2) * N^-1: twiddle of the last inverse stage with the scaling folded in
    u32 mu32;         // floor(2^32 / Q): one-step reduction of any 32-bit value to [0, 2Q)
    u32 lazy;         // 1 when the forward NTT can run without any correction (bounds in engine.cpp)
    u32 c32;          // 2^32 mod Q (folds 64-bit MAC sums of un-normalised NTT outputs)
    u32 occupancy_target;  // workgroups per CU the blind-rotation kernel is compiled for (2 or 3)
    u32 variant;           // blind-rotation kernel choice: 0 automatic; 1 one-wave-per-transform kernel; 2 / 3 split-
                           // transform kernel forced to its 256- / 128-register build (development knob BCE_VARIANT)
    u32 cu_count;          // compute units of the device (automatic choice: a launch of <= cu_count workgroups)
    u32* xcd_gate;         // [16 XCC ids][32 words] arrival counters of the launch's workgroups per XCD, or null: multi-round launches of
                           // the split-transform kernel start their workgroups in per-XCD cohorts (round 4; BCE_XCD_GATE=0 disables)
    u32 xcd_gate_ticks;    // longest wait at that gate (100 MHz ticks)
    u32 fuse_tail;         // 1: saturated launches of the split-transform kernel run the tail in their epilogue (BCE_FUSE_TAIL=0 disables)
    u32 fold_ninv;         // 1 (folded fp64 kernels, kernels64.hip): rows l >= 1 of the key are also multiplied by N^-1 and the
                           // evaluation-form accumulator is kept scaled by N^-1, so that the un-normalised inverse transform yields
                           // the coefficients themselves (no N^-1 product per coefficient and step; -DBCE_KEY_NINV=0 disables)
    u32 fold;              // 1: the key is stored with the lowest gadget digit folded in (rows l >= 1 hold ek_l - B^l ek_0) and
                           // the kernels multiply the digit-0 rows by the evaluation-form accumulator itself (BCE_FOLD=0 disables)
    u32 I4[4], I4s[4];     // powers of I = psi^(N/2) (primitive 4th root of unity) and Shoup companions
    const uint2* tw_f;  // [N] (psi^brv(i), shoup), index m+i as in the CT forward NTT; the inverse
                        // transform derives psi^-k = -psi^(N-k) from the same table
    const u32* psi_tab; // [N] psi^e in natural exponent order (monomial lookups of the GINX MAC)
    // Montgomery form of the GINX MAC tail in the split-transform kernels (R = 2^32, kernels.hip ginx_mac_tail_redc):
    const u32* psi_tab_r2;  // [N] psi^e * R^2 mod Q
    u32 qinv_neg;           // -Q^-1 mod 2^32
    u32 r2_off;             // Q - (R^2 mod Q): psi^e R^2 + r2_off = (psi^e - 1) R^2 mod Q, lazily in [0, 2Q)
    const u32* bsk;     // EVALUATION domain, GINX [n][2][R][2][N]; AP [n][baseR][dR][R][2][N]
    // ---- 64-bit ring modulus path (kernels64.hip), used when is64 != 0 (Q >= 2^28) ----
    u32 is64;
    u64 Q64, Q8p1_64;
    u64 mu64;           // floor(2^64 / Q)
    u64 c64;            // 2^64 mod Q
    u64 Ninv64, Ninv64_s;
    const ulonglong2* tw64;  // [N] (psi^brv(i), floor(. * 2^64 / Q))
    const u64* bsk64;
    // double-precision formulation of the 64-bit path (kernels64.hip, namespace wd), used when fp64 != 0 (Q < 2^39):
    // the key words at bsk64 are then IEEE doubles
    u32 fp64;
    double Qd, invQd;          // Q, 1/Q
    double Ninvd, Ninvd_q;     // N^-1 mod Q and N^-1 / Q
    const double2* tw64d;      // [N] (psi^brv(i), psi^brv(i) / Q)
    const void* ksk;    // [N][baseKS][dKS][ksk_stride]
    u32* pool;          // [slots][pool_stride]
    u32 pool_stride;
};

// LDS bytes the blind-rotation kernel needs for these parameters
size_t blind_rotate_lds_bytes(const DevParams& P);

// ---- dependency-driven evaluation of a whole bootstrap DAG by ONE persistent launch (k_bootstrap_dag) ----------------
// Replaces the manager <-> executor loop of the reference (src/circuit.cpp:575-683 ready-gate rule, :698-710 parallel
// region) on the device: every (task, instance) pair has a counter of unfinished producers; a workgroup that finishes a
// bootstrap decrements its consumers' counters and pushes those that reach zero to a ready queue; idle workgroups pull
// from the queues.  No kernel boundary and no device-wide barrier between dependent bootstraps.
constexpr u32 kDagQueues = 4;        // priority classes (0 = most urgent)
constexpr u32 kDagCtlStride = 32;    // control block: words [0, 4) = heads, [4, 8) = tails of the four classes (one 32-byte block)
constexpr u32 kDagCuKeys = 4096;     // (XCC id, SE, SH, CU) keys of the placement table
// control block (u32 words): heads and tails in the first line; then the words below; then the per-CU tables
constexpr u32 kDagAbort = kDagQueues * kDagCtlStride;      // != 0: a poller gave up (code)
constexpr u32 kDagDone = kDagAbort + 1;                    // bootstraps completed
constexpr u32 kDagLazyWaits = kDagAbort + 2;               // diagnostics: claims a half-busy CU delayed for an idle one
constexpr u32 kDagBusyTicks = kDagAbort + 4;               // u64: 100 MHz ticks workgroups spent between claiming an item and releasing its consumers
constexpr u32 kDagWaitTicks = kDagAbort + 6;               // u64: ticks workgroups spent looking for an item they then got
constexpr u32 kDagIdleCus = 8;                             // compute units none of whose workgroups runs a bootstrap (same line as heads / tails)
constexpr u32 kDagCuArrive = kDagAbort + 32;               // [kDagCuKeys] workgroups that announced themselves per CU
constexpr u32 kDagCuBusy = kDagCuArrive + kDagCuKeys;      // [kDagCuKeys] workgroups running a bootstrap per CU
constexpr u32 kDagGate = kDagCuBusy + kDagCuKeys;          // [16 XCC ids] x 32 words: word pair 0 = the XCD's start gate (u64)
constexpr u32 kDagCohortRing = 64;                         // generations of an XCD's gate whose (size, arrived) pairs are kept
constexpr u32 kDagCohort = kDagGate + 16 * 32;             // [16 XCC ids][kDagCohortRing] x {size, arrived}
constexpr u32 kDagGateWaits = kDagAbort + 8;               // u64: ticks workgroups waited at their XCD's start gate
constexpr u32 kDagCtlWords = kDagCohort + 16 * kDagCohortRing * 2;
struct DagParams {
    const bce_gate_desc* tasks;   // [n_tasks] topological order, SSA slots
    const u32* cons_off;          // [n_tasks + 1]
    const u32* cons;              // consumers of every task
    const uint8_t* qid;           // [n_tasks] priority class
    const u32* dep_init;          // [n_tasks] producers inside the DAG (0..2)
    u32* dep;                     // [instances][n_tasks] unfinished producers
    u32* slots[kDagQueues];       // queue entries: item + 1 (0 = not yet pushed); item = instance * n_tasks + task
    u32 qcap[kDagQueues];         // entries the queue will ever receive (tasks of that class x instances)
    const u32* init_items;        // initially ready tasks of every class, concatenated; init_off[q] .. init_off[q + 1]
    u32 init_off[kDagQueues + 1];
    u32* ctl;                     // control block, kDagCtlWords words
    u32 n_tasks, instances, slot_stride, slot_base;   // instance k: slot numbers + slot_base + k * slot_stride
    u32 lazy_ticks;               // 100 MHz ticks a workgroup on a half-busy CU leaves a lone ready item to an idle CU
    u32 stall_ticks;              // 100 MHz ticks without any push after which a poller sets the abort word
    u32 policy;                   // bit 0: placement-aware claims (idle CUs first); bit 1: dry run (development);
                                  // bit 2: workgroups of one XCD that claimed from a deep queue start their bootstraps together;
                                  // bit 3: tickets for every claim (round 3's rule, kept for the A/B against the hybrid claim)
    u32 gate_ticks;               // longest wait at the XCD start gate (100 MHz ticks)
    u32 gate_backlog;             // queue depth at claim time from which a bootstrap goes through the gate
};
// rearm = reset counters / queues for one evaluation; then the persistent launch.  wps = 2: one workgroup per CU
// (256-register build), 4: two per CU.  grid = resident workgroups (never more).
hipError_t launch_dag_rearm(const DagParams& D, hipStream_t s);
// d_P / d_params: the engine's DevParams and the run's DagParams in DEVICE memory (the kernel reads their fields where
// it needs them instead of holding kernel arguments in registers across its loop)
hipError_t launch_bootstrap_dag(const DevParams& P, const DevParams* d_P, const DagParams* d_params, int wps, u32 grid,
                                hipStream_t s);
bool dag_kernel_available(const DevParams& P);
// the config-5 class (kernels64.hip): N = 2048, Q < 2^39 in doubles, AP, folded key, fused tail; 1,024 threads, one workgroup per CU
bool dag64_kernel_available(const DevParams& P);
hipError_t launch_bootstrap_dag64(const DevParams& P, const DevParams* d_P, const DagParams* d_params, u32 grid, hipStream_t s);

// Timestamps of a launch without event packets of their own in the queue: start / stop are attached to the kernel's
// dispatch (hipExtLaunchKernel) -- what hipEventRecord before and after a kernel costs on the device timeline (5-10 us of
// idle time each side) is what a dependent step of one bootstrap latency cannot afford.  Null members: nothing attached.
struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};
// kern(args...) through hipExtLaunchKernel with the two events attached to the DISPATCH (its start / stop timestamps; either
// may be null: the launch is then simply untimed on that side); args are converted to the kernel's own parameter types first
template <typename... KA, typename... A>
inline hipError_t launch_with_events(void (*kern)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t s, LaunchEvents ev, A... args) {
    static_assert(sizeof...(KA) == sizeof...(A), "argument count");
    std::tuple<KA...> held{static_cast<KA>(args)...};
    void* ptrs[sizeof...(KA)];
    std::apply([&](auto&... a) { size_t i = 0; ((ptrs[i++] = (void*)&a), ...); }, held);
    return hipExtLaunchKernel(reinterpret_cast<const void*>(kern), grid, block, ptrs, lds, s, ev.start, ev.stop, 0);
}

// acc_out: u32 [n_boot][2][N], COEFFICIENT domain, values in [0, Q)
// *kernel_id (optional) receives the enum bce_br_kernel value of the kernel that was launched
// *tail_fused (optional) is set when the launched kernel also ran the tail of EvalBinGate (extract, ModSwitch,
// KeySwitch, ModSwitch -> pool[out]) in its epilogue; the caller then skips launch_tail().  dbg_lweN / dbg_ks as
// for launch_tail (used only when the tail is fused).
hipError_t launch_blind_rotate(const DevParams& P, const bce_gate_desc* d_descs, u32 n_desc, u32 instances,
                               u32 slot_stride, u32* acc_out, hipStream_t s, int* kernel_id = nullptr,
                               u32* dbg_lweN = nullptr, u32* dbg_ks = nullptr, bool* tail_fused = nullptr, LaunchEvents ev = {});

// extract + ModSwitch(Q->qKS) + KeySwitch + ModSwitch(qKS->q) -> pool[out]
// dbg_lweN: u32 [n_boot][N+1] or null; dbg_ks: u32 [n_boot][n+1] or null
// partial: device scratch of tail_partial_words(P, n_desc * instances) u64 words (partial key-switch sums)
size_t tail_partial_words(const DevParams& P, u32 boots);
hipError_t launch_tail(const DevParams& P, const bce_gate_desc* d_descs, u32 n_desc, u32 instances, u32 slot_stride,
                       const void* acc_in /* u32 or u64 words by P.is64 */, u64* partial, u32* dbg_lweN, u32* dbg_ks,
                       hipStream_t s, LaunchEvents ev = {});   // ev.start on the first kernel, ev.stop on the last

// 64-bit-modulus counterparts (kernels64.hip)
size_t blind_rotate64_lds_bytes(const DevParams& P);
bool blind_rotate64_narrow(const DevParams& P);   // integer 64-bit kernel with 32-bit digit rows (four gadget digits, N >= 1024, Q < 2^31)
// *tail_fused (optional) is set when the launched kernel also ran the tail (then dbg_lweN / dbg_ks are its debug outputs)
hipError_t launch_blind_rotate64(const DevParams& P, const bce_gate_desc* d_descs, u32 n_desc, u32 instances,
                                 u32 slot_stride, u64* acc_out, hipStream_t s, u32* dbg_lweN = nullptr, u32* dbg_ks = nullptr,
                                 bool* tail_fused = nullptr, LaunchEvents ev = {});
hipError_t launch_ntt_batch64(const DevParams& P, u64* polys, u32 count, int inverse, hipStream_t s);
// key words u64 <-> IEEE double in place (layout of the double-precision formulation)
hipError_t launch_words_u64_f64(u64* words, size_t count, int to_double, hipStream_t s);
hipError_t launch_pointwise_mac64(const DevParams& P, u64* b, const u64* a, const u64* z, u32 count, u32 b_step,
                                  hipStream_t s);

// EvalNOT / COPY over pool slots
hipError_t launch_lwe_unary(const DevParams& P, const bce_gate_desc* d_descs, u32 n_desc, u32 instances,
                            u32 slot_stride, hipStream_t s);

// dense buffer [count][n+1] u32 <-> pool rows named by d_descs[i].in0
hipError_t launch_pool_pack(const DevParams& P, const bce_gate_desc* d_descs, u32 count, u32* buf, int to_pool,
                            hipStream_t s);

// in-place negacyclic NTT of `count` polys, u32 [count][N] in global memory
hipError_t launch_ntt_batch(const DevParams& P, u32* polys, u32 count, int inverse, hipStream_t s);

// In place on `rgsw` RGSW ciphertexts [R = 2 dG rows][2][N] (evaluation form, integer words: u32, or u64 when P.is64):
// dir = +1: row(2l + c) -= B^l row(c) for l = 1..dG-1 (the layout the FOLD kernels read); dir = -1 undoes it.
hipError_t launch_fold_gadget(const DevParams& P, void* bsk, u64 rgsw, int dir, hipStream_t s);

// b[p*b_step][k] += a[p][k] * z[k] mod Q for count polys (key generation)
hipError_t launch_pointwise_mac(const DevParams& P, u32* b, const u32* a, const u32* z, u32 count, u32 b_step,
                                hipStream_t s);

// host-side decrypt helper kernels are not needed: decryption happens on the host (needs sk)

}  // namespace bce
