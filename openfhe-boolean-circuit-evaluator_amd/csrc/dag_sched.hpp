// dag_sched.hpp -- device side of the dependency-driven evaluation (bce_dag_*): ready queues, tickets, placement, XCD start
// gate and the persistent loop of a workgroup, shared by the 32-bit kernel (kernels.hip: k_bootstrap_dag) and the config-5
// kernel (kernels64.hip: k_bootstrap_dag64).  Included inside namespace bce.  What it replaces: the reference's manager <->
// executor loop, src/circuit.cpp:575-683 (a gate is ready when its last input arrived) and :685-817 (run and retire).
#pragma once

namespace {
constexpr u32 kDagExit = 0xFFFFFFFFu;
// every shared word is accessed through the GLOBAL address space (global_* instructions, never flat_*)
typedef __attribute__((address_space(1))) u32 gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ void dag_st(u32* p, u32 v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 dag_add(u32* p, u32 v) { return __hip_atomic_fetch_add((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- wave-uniform scheduler code -------------------------------------------------------------------------------------
// The scheduler of a workgroup runs on wave 0 with ALL 64 lanes active and every operand wave-uniform: the branch that
// selects it tests the wave number in a scalar register, so whole waves take one side of it and the optimiser has no
// lane-divergent, loop-invariant condition to unswitch the persistent loop on (round 3 found one: `threadIdx.x == 0`
// split the loop into a path for lane 0 and one for lanes 1..63, and lane 0 never claimed).  Loads of a uniform address
// are one request whatever the lane count; their results go through v_readfirstlane, so every decision below is a scalar
// branch.  What must happen ONCE per wave -- the read-modify-writes -- is a single asm statement that narrows EXEC to
// lane 0 around the one memory instruction and restores it (the compiler never sees a changed EXEC): no branch at all.
__device__ __forceinline__ u32 u_ld(const u32* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load((const gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ unsigned long long u_ld64(const void* p) {
    const unsigned long long v = __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ((unsigned long long)__builtin_amdgcn_readfirstlane((u32)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)v);
}
// agent-scope relaxed read-modify-writes of gfx942 / gfx950 go to L2 as they are (no sc1); sc0 = return the old value
#define BCE_ONE_LANE(INSN) "s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, 1\n\t" INSN "\n\ts_waitcnt vmcnt(0)\n\ts_mov_b64 exec, %[sv]"
__device__ __forceinline__ u32 u_add(u32* p, u32 v) {                      // returns the value before
    u32 r; unsigned long long sv;
    asm volatile(BCE_ONE_LANE("global_atomic_add %[r], %[p], %[v], off sc0") : [r] "=&v"(r), [sv] "=&s"(sv) : [p] "v"(p), [v] "v"(v) : "memory");
    return __builtin_amdgcn_readfirstlane(r);
}
__device__ __forceinline__ u32 u_sub(u32* p, u32 v) {
    u32 r; unsigned long long sv;
    asm volatile(BCE_ONE_LANE("global_atomic_sub %[r], %[p], %[v], off sc0") : [r] "=&v"(r), [sv] "=&s"(sv) : [p] "v"(p), [v] "v"(v) : "memory");
    return __builtin_amdgcn_readfirstlane(r);
}
__device__ __forceinline__ u32 u_cas(u32* p, u32 expect, u32 desired) {    // returns the value before
    u32 r; unsigned long long sv;
    const unsigned long long pair = ((unsigned long long)expect << 32) | desired;   // DATA[0] = new value, DATA[1] = compare
    asm volatile(BCE_ONE_LANE("global_atomic_cmpswap %[r], %[p], %[d], off sc0") : [r] "=&v"(r), [sv] "=&s"(sv) : [p] "v"(p), [d] "v"(pair) : "memory");
    return __builtin_amdgcn_readfirstlane(r);
}
__device__ __forceinline__ unsigned long long u_add64(void* p, unsigned long long v) {
    unsigned long long r, sv;
    asm volatile(BCE_ONE_LANE("global_atomic_add_x2 %[r], %[p], %[v], off sc0") : [r] "=&v"(r), [sv] "=&s"(sv) : [p] "v"(p), [v] "v"(v) : "memory");
    return ((unsigned long long)__builtin_amdgcn_readfirstlane((u32)(r >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)r);
}
typedef u32 dag_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long u_cas64(void* p, unsigned long long expect, unsigned long long desired) {
    unsigned long long r, sv;
    dag_u32x4 d;
    d.x = (u32)desired; d.y = (u32)(desired >> 32); d.z = (u32)expect; d.w = (u32)(expect >> 32);
    asm volatile(BCE_ONE_LANE("global_atomic_cmpswap_x2 %[r], %[p], %[d], off sc0") : [r] "=&v"(r), [sv] "=&s"(sv) : [p] "v"(p), [d] "v"(d) : "memory");
    return ((unsigned long long)__builtin_amdgcn_readfirstlane((u32)(r >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)r);
}

// Wave-uniform: next item of the highest non-empty priority class, or kDagExit when every class has been claimed to
// its end (or the run was aborted).
// The poller reads the eight head / tail words (one 32-byte block, one cache line for every poller of the chip) and
// claims only from a class that shows a backlog (tail > head):
//   * backlog >= the launch's workgroups: a TICKET, one fetch-add on the head -- wait-free, and every poller that can
//     race for this backlog finds an entry behind its ticket (a poller takes one ticket per look);
//   * a shorter backlog: compare-and-swap of exactly the head it saw; a loser looks again at its normal polling pace.
//     Round 3 took tickets here too: every poller that saw the short backlog got one, and the losers then sat on tickets
//     for FUTURE entries of that one class without looking at the others -- at the start of a run all eager workgroups
//     pick class 0, and the ones beyond its initial entries stayed parked there while ready work of classes 1..3 was
//     served by the few that were not (policy bit 3 = that behaviour, for the A/B of tests/test_gpu_dataflow.py).
//   A compare-and-swap for EVERY claim makes every idle workgroup retry against every other one when the queues are deep:
//   measured 5.6 us per claim, serialised, with 512 workgroups (180 k claims/s for the whole chip).
// A head never passes its tail this way, so `head >= qcap` is "every entry of the class has been claimed".
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
    const u32 pollers = gridDim.x;
    for (u32 spin = 0;; ++spin) {
        const u64 now = __builtin_amdgcn_s_memrealtime();
        u32 hd[kDagQueues], tl[kDagQueues];
#pragma unroll
        for (u32 q = 0; q < kDagQueues; ++q) { hd[q] = u_ld(ctl + q); tl[q] = u_ld(ctl + kDagQueues + q); }
        bool lazy = false;
        u32 idle_cus = 0;
        if (D.policy & 1u) {
            lazy = !first_on_cu || u_ld(my_busy) != 0;
            if (lazy) idle_cus = u_ld(ctl + kDagIdleCus);
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
                    u_add(ctl + kDagLazyWaits, 1);
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
            u32 t;
            bool mine = true;
            if ((D.policy & 8u) || tl[pick] - hd[pick] >= pollers) {
                t = u_add(ctl + pick, 1u);
            } else {
                t = hd[pick];
                mine = u_cas(ctl + pick, t, t + 1u) == t;
            }
            if (mine && t < D.qcap[pick]) {
                const u32* const entry = D.slots[pick] + t;
                for (u32 w = 0;; ++w) {
                    const u32 v = u_ld(entry);
                    if (v) return v - 1;
                    if ((w & 63u) == 63u) {
                        if (u_ld(ctl + kDagAbort)) return kDagExit;
                        const u64 tw = __builtin_amdgcn_s_memrealtime();
                        const u32 d = u_ld(ctl + kDagDone);
                        if (d != last_done) { last_done = d; t_progress = tw; }
                        else if (tw - t_progress > D.stall_ticks) { dag_st(ctl + kDagAbort, 2u); return kDagExit; }
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            if (mine) continue;   // a ticket beyond the class's last entry: nothing there, look again
            // lost the race for a short backlog: poll on
        }
        else if (!alive) return kDagExit;
        if ((spin & 7u) == 7u) {
            if (u_ld(ctl + kDagAbort)) return kDagExit;
            const u32 d = u_ld(ctl + kDagDone);
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
constexpr u32 kDagNoCohort = 0xFFFFFFFFu;
// (all wave-uniform, see above)
// counts this workgroup into its old cohort; true when that completes the cohort
__device__ __forceinline__ bool dag_cohort_arrive(u32* ctl, u32 xcc, u32 prev) {
    u32* const c = ctl + kDagCohort + (xcc * kDagCohortRing + (prev % kDagCohortRing)) * 2u;
    return u_add(c + 1, 1u) + 1u >= u_ld(c);
}
__device__ __forceinline__ bool dag_gate_open(u32* ctl, u32 xcc, unsigned long long seen) {
    u32* const g = ctl + kDagGate + 32u * xcc;
    const u32 gen = (u32)(seen >> 32);
    if (u_cas64(g, seen, (unsigned long long)(gen + 1u) << 32) != seen) return false;
    u32* const c = ctl + kDagCohort + (xcc * kDagCohortRing + (gen % kDagCohortRing)) * 2u;
    dag_st(c + 1, 0u);                 // members come back milliseconds later
    dag_st(c, (u32)seen);              // size of the cohort that starts now
    return true;
}
// returns the generation (= cohort) this bootstrap starts in
__device__ __forceinline__ u32 dag_gate_enter(u32* ctl, u32 xcc, u32 prev, u32 gate_ticks) {
    u32* const g = ctl + kDagGate + 32u * xcc;
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long old = u_add64(g, 1ull);
    const u32 gen = (u32)(old >> 32);
    const bool opener = prev != kDagNoCohort && dag_cohort_arrive(ctl, xcc, prev);
    const u32 limit = prev == kDagNoCohort ? gate_ticks / 8u : gate_ticks;
    for (;;) {
        const unsigned long long v = u_ld64(g);
        if ((u32)(v >> 32) != gen) break;                                  // opened for this group
        if (opener || __builtin_amdgcn_s_memrealtime() - t0 > limit) {
            if (dag_gate_open(ctl, xcc, v)) break;
            continue;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    u_add64(ctl + kDagGateWaits, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - t0));
    return gen;
}
// a member leaves its cohort without coming back to the gate
__device__ __forceinline__ void dag_gate_resign(u32* ctl, u32 xcc, u32 prev) {
    if (!dag_cohort_arrive(ctl, xcc, prev)) return;
    u32* const g = ctl + kDagGate + 32u * xcc;
    for (int tries = 0; tries < 4; ++tries) {
        const unsigned long long v = u_ld64(g);
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
    // wave number in a scalar register: `wave == 0` below is a scalar branch
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x) >> 6;
    u32 first_on_cu = 0, cu_key = 0, cohort = kDagNoCohort;   // scheduler state of wave 0, wave-uniform
    if (wave == 0) {
        // which CU this workgroup sits on: HW_ID[15:8] = (SE, SH, CU), XCC_ID[3:0]
        const u32 hwid = __builtin_amdgcn_s_getreg((31u << 11) | 4u), xcc = __builtin_amdgcn_s_getreg((31u << 11) | 20u);
        cu_key = ((xcc & 15u) << 8) | ((hwid >> 8) & 255u);
        const u32 arrival = u_add(Dp->ctl + kDagCuArrive + cu_key, 1u);
        if (arrival == 0) u_add(Dp->ctl + kDagIdleCus, 1u);
        first_on_cu = arrival == 0;   // the CU's first workgroup claims eagerly, later arrivals yield to idle CUs
    }
    u32 t_claim = 0;
    for (;;) {
        if (wave == 0) {
            ConstDagParams& D = *as_constant<ConstDagParams>(Dp);
            u32* const my_busy = D.ctl + kDagCuBusy + cu_key;
            const u64 t_in = __builtin_amdgcn_s_memrealtime();
            bool deep;
            const u32 it = dag_acquire(D, my_busy, first_on_cu != 0, deep);
            if (it != kDagExit) {
                const u64 t_got = __builtin_amdgcn_s_memrealtime();
                u_add64(D.ctl + kDagWaitTicks, t_got - t_in);
                t_claim = (u32)t_got;
                if (u_add(my_busy, 1u) == 0) u_sub(D.ctl + kDagIdleCus, 1u);
                const bool gated = (D.policy & 4u) && deep;
                const u32 prev = cohort;
                if (gated) cohort = dag_gate_enter(D.ctl, cu_key >> 8, prev, D.gate_ticks);
                else if (prev != kDagNoCohort) { dag_gate_resign(D.ctl, cu_key >> 8, prev); cohort = kDagNoCohort; }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            else if (cohort != kDagNoCohort) { dag_gate_resign(D.ctl, cu_key >> 8, cohort); cohort = kDagNoCohort; }
            mbox[0] = it;             // 64 lanes, one address, one value
        }
        __syncthreads();
        const u32 item = __builtin_amdgcn_readfirstlane(mbox[0]);
        if (item == kDagExit) break;
        {
            ConstDagParams& D = *as_constant<ConstDagParams>(Dp);
            const u32 nt = D.n_tasks, k = item / nt, t = item - k * nt;
            // policy bit 1 (development): walk the DAG without running the bootstraps -- the scheduler's own time
            if (!(D.policy & 2u)) run_bootstrap(D, t, k);
        }
        // publish: every storing wave drains, barrier, one wave releases at agent scope, then the counters
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (wave == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ConstDagParams& D = *as_constant<ConstDagParams>(Dp);
            const u32 nt = D.n_tasks, k = item / nt, t = item - k * nt;
            u32* const dep = D.dep + (size_t)k * nt;
            const u32 c1 = D.cons_off[t + 1];
            // one consumer per lane (a data-dependent lane mask, nothing loop-invariant).  The decrement is acquire-release:
            // the producer that brings a consumer's count to zero has, by this acquire, the OTHER producer's ciphertext
            // store (released before that one's decrement) ordered before its own release below, so the consumer's single
            // acquire fence after reading its queue entry covers both inputs.
            for (u32 i = D.cons_off[t] + (threadIdx.x & 63u); i < c1; i += 64) {
                const u32 c = D.cons[i];
                if (__hip_atomic_fetch_sub((gu32*)(dep + c), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 1u) {
                    const u32 q = D.qid[c];
                    const u32 idx = dag_add(D.ctl + kDagQueues + q, 1u);
                    __hip_atomic_store((gu32*)(D.slots[q] + idx), k * nt + c + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const u32 dt = (u32)__builtin_amdgcn_s_memrealtime() - t_claim;
            u_add64(D.ctl + kDagBusyTicks, (u64)dt);
            u_add(D.ctl + kDagDone, 1u);
            if (u_sub(D.ctl + kDagCuBusy + cu_key, 1u) == 1u) u_add(D.ctl + kDagIdleCus, 1u);
        }
    }
}

}  // namespace
