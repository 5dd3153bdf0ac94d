// dag_sched.hpp -- device side of the dependency-driven evaluation (bce_dag_*): ready queues, tickets, placement, XCD start
// gate and the persistent loop of a workgroup, shared by the 32-bit kernel (kernels.hip: k_bootstrap_dag) and the config-5
// kernel (kernels64.hip: k_bootstrap_dag64).  Included inside namespace bce.  What it replaces: the reference's manager <->
// executor loop, src/circuit.cpp:575-683 (a gate is ready when its last input arrived) and :685-817 (run and retire).
#pragma once

namespace {
constexpr u32 kDagExit = 0xFFFFFFFFu;
// every shared word is accessed through the GLOBAL address space (global_* instructions, never flat_*)
typedef __attribute__((address_space(1))) u32 gu32;
__device__ __forceinline__ u32 dag_ld(const u32* p) { return __hip_atomic_load((const gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dag_st(u32* p, u32 v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 dag_add(u32* p, u32 v) { return __hip_atomic_fetch_add((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ONE thread: next item of the highest non-empty priority class, or kDagExit when every class has been claimed to
// its end (or the run was aborted).
// Claiming is wait-free: the poller reads the eight head / tail words (one 32-byte block, one cache line for every
// poller of the chip), and only when a class shows a backlog (tail > head) does it take a TICKET with one fetch-add on
// that head; the entry of that ticket is its own word to wait for (normally already written; when more pollers than
// entries raced for the backlog, the ticket is a claim on the next entry the class receives).  A compare-and-swap on
// the head instead makes every idle workgroup retry against every other one: measured 5.6 us per claim, serialised,
// with 512 workgroups (180 k claims/s for the whole chip -- the scheduler itself was the bottleneck).
template <typename DT>
__device__ __forceinline__ u32 dag_acquire(const DT& D, const u32* my_busy, bool first_on_cu, bool& deep) {
    deep = false;
    u32* const ctl = D.ctl;
    u32 last_h[kDagQueues];
    u64 seen[kDagQueues];
    bool over[kDagQueues];   // the backlog of the class exceeded the idle CUs at the previous poll
#pragma unroll
    for (u32 q = 0; q < kDagQueues; ++q) { last_h[q] = kDagExit; seen[q] = 0; over[q] = false; }
    u64 t_progress = __builtin_amdgcn_s_memrealtime();
    u32 last_done = kDagExit;
    for (u32 spin = 0;; ++spin) {
        const u64 now = __builtin_amdgcn_s_memrealtime();
        u32 hd[kDagQueues], tl[kDagQueues];
#pragma unroll
        for (u32 q = 0; q < kDagQueues; ++q) { hd[q] = dag_ld(ctl + q); tl[q] = dag_ld(ctl + kDagQueues + q); }
        bool lazy = false;
        u32 idle_cus = 0;
        if (D.policy & 1u) {
            lazy = !first_on_cu || dag_ld(my_busy) != 0;
            if (lazy) idle_cus = dag_ld(ctl + kDagIdleCus);
        }
        bool alive = false;
        u32 pick = kDagQueues;
#pragma unroll
        for (u32 q = 0; q < kDagQueues; ++q) {
            if (hd[q] >= D.qcap[q]) continue;
            alive = true;
            const u32 backlog = tl[q] - hd[q];
            if ((int)backlog <= 0) { over[q] = false; continue; }
            if (pick != kDagQueues) continue;
            if (lazy) {
                // what the idle compute units can take is left to them: claim only a backlog beyond that (twice in a
                // row: the idle ones need a poll to react), or an entry nobody wanted for lazy_ticks
                const bool was_over = over[q];
                over[q] = (int)backlog > (int)idle_cus;
                if (!(over[q] && was_over)) {
                    if (hd[q] != last_h[q]) { last_h[q] = hd[q]; seen[q] = now; continue; }
                    if (now - seen[q] < D.lazy_ticks) continue;
                    dag_add(ctl + kDagLazyWaits, 1);
                }
            }
            pick = q;
        }
        if (pick != kDagQueues) {
            {   // "saturated" = ready entries over all classes (an urgent class is short by nature)
                u32 ready = 0;
#pragma unroll
                for (u32 q = 0; q < kDagQueues; ++q) { const u32 b = tl[q] - hd[q]; if ((int)b > 0) ready += b; }
                deep = ready >= D.gate_backlog;
            }
            const u32 t = dag_add(ctl + pick, 1u);
            if (t < D.qcap[pick]) {
                const u32* const entry = D.slots[pick] + t;
                for (u32 w = 0;; ++w) {
                    const u32 v = dag_ld(entry);
                    if (v) return v - 1;
                    if ((w & 63u) == 63u) {
                        if (dag_ld(ctl + kDagAbort)) return kDagExit;
                        const u64 tw = __builtin_amdgcn_s_memrealtime();
                        const u32 d = dag_ld(ctl + kDagDone);
                        if (d != last_done) { last_done = d; t_progress = tw; }
                        else if (tw - t_progress > D.stall_ticks) { dag_st(ctl + kDagAbort, 2u); return kDagExit; }
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            continue;   // a ticket beyond the class's last entry: nothing there, look again
        }
        if (!alive) return kDagExit;
        if ((spin & 7u) == 7u) {
            if (dag_ld(ctl + kDagAbort)) return kDagExit;
            const u32 d = dag_ld(ctl + kDagDone);
            if (d != last_done) { last_done = d; t_progress = now; }
            else if (now - t_progress > D.stall_ticks) { dag_st(ctl + kDagAbort, 1u); return kDagExit; }
        }
        // idle: poll gently (the wake-up delay is microseconds on a bootstrap of milliseconds; hundreds of workgroups
        // re-reading the control line every microsecond slow the ones that work)
        if (spin < 4) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(127);
    }
}
// ---- XCD start gate ------------------------------------------------------------------------------------------------
// A step of a bootstrap streams 128 KiB of key rows; the 64 workgroups of an XCD share one 4 MiB L2.  Started together
// (a per-frontier launch does that) they walk the key in lock-step and every row is fetched into that L2 once; left to
// drift -- each workgroup takes its next bootstrap when it happens to finish -- every workgroup streams the whole 62.8 MiB
// key through the L2 by itself: FETCH_SIZE per bootstrap x3, each bootstrap 3.4-3.7 ms instead of 3.1
// (profiles/r03_dataflow_kernel.log).  The gate restores the lock-step where it pays, in the saturated regime (a claim
// made while >= gate_backlog entries were ready): the workgroups of an XCD start their bootstraps in COHORTS.
//   gate word (u64 per XCD): [generation : 32 | waiting : 32].  A workgroup arrives (waiting++), remembers the generation
//   it waits for, and starts when the generation moves on.  Whoever opens the gate (CAS generation + 1, waiting = 0)
//   records the cohort's size; a member that comes back after its bootstrap counts itself into arrived[old generation],
//   and the member that completes the count opens the gate for everybody waiting -- bootstraps that started together end
//   within tens of microseconds of each other, so a cohort re-forms without waiting for anybody else's cohort, and
//   cohorts whose ends overlap merge.  Workgroups without a cohort (first bootstrap, or the last one ran ungated) gather
//   for gate_ticks / 8; nobody waits longer than gate_ticks.  A member that leaves (ungated claim, exit) resigns, so that
//   its cohort's count still completes.
typedef __attribute__((address_space(1))) unsigned long long gu64;
constexpr u32 kDagNoCohort = 0xFFFFFFFFu;
// counts this workgroup into its old cohort; true when that completes the cohort
__device__ __forceinline__ bool dag_cohort_arrive(u32* ctl, u32 xcc, u32 prev) {
    u32* const c = ctl + kDagCohort + (xcc * kDagCohortRing + (prev % kDagCohortRing)) * 2u;
    return dag_add(c + 1, 1u) + 1u >= dag_ld(c);
}
__device__ __forceinline__ bool dag_gate_open(u32* ctl, u32 xcc, unsigned long long seen) {
    gu64* const g = (gu64*)(ctl + kDagGate + 32u * xcc);
    const u32 gen = (u32)(seen >> 32);
    if (!__hip_atomic_compare_exchange_strong(g, &seen, (unsigned long long)(gen + 1u) << 32, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT))
        return false;
    u32* const c = ctl + kDagCohort + (xcc * kDagCohortRing + (gen % kDagCohortRing)) * 2u;
    dag_st(c + 1, 0u);                 // members come back milliseconds later
    dag_st(c, (u32)seen);              // size of the cohort that starts now
    return true;
}
// returns the generation (= cohort) this bootstrap starts in
__device__ __forceinline__ u32 dag_gate_enter(u32* ctl, u32 xcc, u32 prev, u32 gate_ticks) {
    gu64* const g = (gu64*)(ctl + kDagGate + 32u * xcc);
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long old = __hip_atomic_fetch_add(g, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u32 gen = (u32)(old >> 32);
    const bool opener = prev != kDagNoCohort && dag_cohort_arrive(ctl, xcc, prev);
    const u32 limit = prev == kDagNoCohort ? gate_ticks / 8u : gate_ticks;
    for (;;) {
        const unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((u32)(v >> 32) != gen) break;                                  // opened for this group
        if (opener || __builtin_amdgcn_s_memrealtime() - t0 > limit) {
            if (dag_gate_open(ctl, xcc, v)) break;
            continue;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    __hip_atomic_fetch_add((gu64*)(ctl + kDagGateWaits), (unsigned long long)(__builtin_amdgcn_s_memrealtime() - t0), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    return gen;
}
// a member leaves its cohort without coming back to the gate
__device__ __forceinline__ void dag_gate_resign(u32* ctl, u32 xcc, u32 prev) {
    if (!dag_cohort_arrive(ctl, xcc, prev)) return;
    gu64* const g = (gu64*)(ctl + kDagGate + 32u * xcc);
    for (int tries = 0; tries < 4; ++tries) {
        const unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((u32)v == 0 || dag_gate_open(ctl, xcc, v)) return;            // nobody waits / opened
    }
}

// Pp / Dp point to the engine's DevParams and the run's DagParams in device memory.  They are not by-value arguments:
// kernel arguments are loaded at entry and would stay live in scalar registers across the whole persistent loop (the
// per-frontier kernel lets most of them die after its prologue).  Each iteration reads them through a constant-address-
// space pointer made opaque inside the loop: scalar loads where a field is needed, nothing hoisted out of the loop.
typedef const __attribute__((address_space(4))) DevParams ConstDevParams;
typedef const __attribute__((address_space(4))) DagParams ConstDagParams;
template <typename C, typename S>
__device__ __forceinline__ C* as_constant(const S* p) {
    C* c = (C*)(uintptr_t)p;
    asm volatile("" : "+s"(c));
    return c;
}

// The persistent loop of one workgroup: claim -> acquire -> run_bootstrap(D, task, instance) -> publish, until every class
// has been claimed to its end.  mbox: kDagMailboxWords words of LDS owned by the loop.  run_bootstrap must leave the
// refreshed ciphertext in the pool (fused tail) and is called by every thread of the workgroup.
constexpr u32 kDagMailboxWords = 8;
template <typename F>
__device__ __forceinline__ void dag_worker(const DagParams* Dp, u32* smem, F&& run_bootstrap) {
    // eight words in front of the LDS layout of lat_bootstrap: [0] the item the workgroup runs next, [1] "first workgroup
    // of its CU", [2] the CU's key, [3] when the item was claimed, [4] the cohort (gate generation) its last gated bootstrap started in
    u32* const mbox = smem;
    if (threadIdx.x == 0) {
        // which CU this workgroup sits on: HW_ID[15:8] = (SE, SH, CU), XCC_ID[3:0]
        const u32 hwid = __builtin_amdgcn_s_getreg((31u << 11) | 4u), xcc = __builtin_amdgcn_s_getreg((31u << 11) | 20u);
        const u32 key = ((xcc & 15u) << 8) | ((hwid >> 8) & 255u);
        const u32 arrival = dag_add(Dp->ctl + kDagCuArrive + key, 1u);
        if (arrival == 0) dag_add(Dp->ctl + kDagIdleCus, 1u);
        mbox[1] = arrival == 0;   // the CU's first workgroup claims eagerly, later arrivals yield to idle CUs
        mbox[2] = key;
        mbox[4] = kDagNoCohort;
    }
    for (;;) {
        // The thread index is made opaque in every iteration and at every use: a comparison the optimiser can prove
        // loop-invariant lets it thread the back edge past the `== 0` test, i.e. split the loop into a path for thread 0
        // and one for the others; the wave then runs them one after the other -- lanes 1..63 of wave 0 spin in "their" loop
        // through the barriers below while lane 0 never gets to claim an item (observed: a hang, and bootstraps run on
        // stale mailbox contents).
        u32 tid_a = threadIdx.x;
        asm volatile("" : "+v"(tid_a));
        if (tid_a == 0) {
            ConstDagParams& D = *as_constant<ConstDagParams>(Dp);
            u32* const my_busy = D.ctl + kDagCuBusy + mbox[2];
            const u64 t_in = __builtin_amdgcn_s_memrealtime();
            bool deep;
            const u32 it = dag_acquire(D, my_busy, mbox[1] != 0, deep);
            if (it != kDagExit) {
                const u64 t_got = __builtin_amdgcn_s_memrealtime();
                __hip_atomic_fetch_add((__attribute__((address_space(1))) u64*)(D.ctl + kDagWaitTicks), t_got - t_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mbox[3] = (u32)t_got;
                if (dag_add(my_busy, 1u) == 0) __hip_atomic_fetch_sub((gu32*)(D.ctl + kDagIdleCus), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool gated = (D.policy & 4u) && deep;
                const u32 prev = mbox[4];
                if (gated) mbox[4] = dag_gate_enter(D.ctl, mbox[2] >> 8, prev, D.gate_ticks);
                else if (prev != kDagNoCohort) { dag_gate_resign(D.ctl, mbox[2] >> 8, prev); mbox[4] = kDagNoCohort; }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            else if (mbox[4] != kDagNoCohort) { dag_gate_resign(D.ctl, mbox[2] >> 8, mbox[4]); mbox[4] = kDagNoCohort; }
            mbox[0] = it;
        }
        __syncthreads();
        {
            const u32 item = __builtin_amdgcn_readfirstlane(mbox[0]);
            if (item == kDagExit) break;
            ConstDagParams& D = *as_constant<ConstDagParams>(Dp);
            const u32 nt = D.n_tasks, k = item / nt, t = item - k * nt;
            // policy bit 1 (development): walk the DAG without running the bootstraps -- the scheduler's own time
            if (!(D.policy & 2u)) run_bootstrap(D, t, k);
        }
        // publish: every storing wave drains, barrier, one wave releases at agent scope, then the counters
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        u32 tid_p = threadIdx.x;
        asm volatile("" : "+v"(tid_p));
        if (tid_p < 64) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ConstDagParams& D = *as_constant<ConstDagParams>(Dp);
            const u32 item = __builtin_amdgcn_readfirstlane(mbox[0]);
            const u32 nt = D.n_tasks, k = item / nt, t = item - k * nt;
            u32* const dep = D.dep + (size_t)k * nt;
            const u32 c1 = D.cons_off[t + 1];
            for (u32 i = D.cons_off[t] + tid_p; i < c1; i += 64) {
                const u32 c = D.cons[i];
                if (__hip_atomic_fetch_sub((gu32*)(dep + c), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u) {
                    const u32 q = D.qid[c];
                    const u32 idx = dag_add(D.ctl + kDagQueues + q, 1u);
                    dag_st(D.slots[q] + idx, k * nt + c + 1u);
                }
            }
            if (tid_p == 0) {
                const u32 dt = (u32)__builtin_amdgcn_s_memrealtime() - mbox[3];
                __hip_atomic_fetch_add((__attribute__((address_space(1))) u64*)(D.ctl + kDagBusyTicks), (u64)dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dag_add(D.ctl + kDagDone, 1u);
                if (__hip_atomic_fetch_sub((gu32*)(D.ctl + kDagCuBusy + mbox[2]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u)
                    dag_add(D.ctl + kDagIdleCus, 1u);

            }
        }
    }
}

}  // namespace
