// phase_prof.hpp -- development aid shared by kernels.hip / kernels64.hip (never defined in the shipped build):
// -DBCE_PHASE_PROF accumulates, for workgroup 0, the cycles between the barriers of a blind-rotation step in the
// including file's own counter array BCE_PROF_ARRAY (no relocatable device code: one array per translation
// unit); read back with bce_debug_phase_prof() / bce_debug_phase_prof64() (tools/phase_prof.py).
#pragma once
#ifdef BCE_PHASE_PROF
// The cycle counts are accumulated in (scalar) registers and written once, after the step loop: a mark is one
// s_memtime, not a global read-modify-write whose latency would land in the phase it closes.
#define BCE_PROF_SLOTS 16
#define BCE_PROF_WAVES 16
#define BCE_PROF_INIT()                                   \
    unsigned long long prof_acc_[BCE_PROF_SLOTS] = {0};   \
    unsigned long long prof_t_ = __builtin_readcyclecounter()
#define BCE_PROF_MARK(slot)                                                         \
    do {                                                                            \
        const unsigned long long now_ = __builtin_readcyclecounter();               \
        prof_acc_[slot] += now_ - prof_t_;                                          \
        prof_t_ = now_;                                                             \
    } while (0)
#define BCE_PROF_FLUSH()                                                            \
    do {                                                                            \
        if (blockIdx.x == 0 && (threadIdx.x & 63u) == 0)                            \
            for (int s_ = 0; s_ < BCE_PROF_SLOTS; ++s_)                             \
                BCE_PROF_ARRAY[(threadIdx.x >> 6) * BCE_PROF_SLOTS + s_] += prof_acc_[s_]; /* one row per wave */ \
    } while (0)
#else
#define BCE_PROF_INIT() do {} while (0)
#define BCE_PROF_MARK(slot) do {} while (0)
#define BCE_PROF_FLUSH() do {} while (0)
#endif
