// mod_sf_debug.h — diagnostic entry points of libmod_sf.so.  NOT part of the product ABI (include/mod_sf.h): they exist only in
// builds made with `make PHASE_COUNTERS=1`, `make ABLATE=1` (tools/dbg_*.py, tools/ablate.sh; they also read MOD_DEBUG) or
// `make CHECKED=1` (index assertions, mod_device.h MOD_CHECK: counters 64..95 of mod_debug_counters; the export copies all 96 words).
#pragma once
#if defined(MOD_PHASE_COUNTERS) || defined(MOD_ABLATION) || defined(MOD_CHECKED)
#include "../../include/mod_sf.h"
extern "C" {
#pragma GCC visibility push(default)
// copy an internal scratch buffer to the host: 0 member norms, 4 member pixels, 1 cluster table, 2 counters
int mod_debug_read(ModContext *ctx, int which, void *dst, unsigned long long bytes);
// 64 cycle / event counters written by the instrumented kernels; reading resets them
int mod_debug_counters(ModContext *ctx, unsigned long long *out64);
#pragma GCC visibility pop
}
#endif
