// phase_prof.hpp -- development aid shared by kernels.hip / kernels64.hip (never defined in the shipped build):
// -DBCE_PHASE_PROF accumulates, for workgroup 0, the cycles between the barriers of a blind-rotation step in the
// including file's own counter array BCE_PROF_ARRAY (no relocatable device code: one array per translation
// unit); read back with bce_debug_phase_prof() / bce_debug_phase_prof64() (tools/phase_prof.py).
#pragma once
#ifdef BCE_PHASE_PROF
#define BCE_PROF_INIT() unsigned long long prof_t_ = __builtin_readcyclecounter()
#define BCE_PROF_MARK(slot)                                                         \
    do {                                                                            \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                  \
            const unsigned long long now_ = __builtin_readcyclecounter();           \
            BCE_PROF_ARRAY[slot] += now_ - prof_t_;                                 \
            prof_t_ = now_;                                                         \
        }                                                                           \
    } while (0)
#else
#define BCE_PROF_INIT() do {} while (0)
#define BCE_PROF_MARK(slot) do {} while (0)
#endif
