// ABI identity, launch-error diagnostics and the "which kernel variant ran" debug hook.
#include "dm_common.h"

#include <stdarg.h>
#include <stdio.h>

static thread_local int g_last_hip_error = 0;
static thread_local const char* g_last_variant = "";
extern "C" void dm_note_hip_error(int code) { g_last_hip_error = code; }
extern "C" int dm_abi_version(void) { return 1; }
extern "C" const char* dm_target_arch(void) { return "gfx950"; }
extern "C" const char* dm_last_hip_error(void) {
  return g_last_hip_error ? hipGetErrorString((hipError_t)g_last_hip_error) : "";
}

// Size-dependent dispatch (64- vs 128-cout conv tile, 16- vs 32-row streaming GEMM tile, streaming vs implicit-GEMM
// 1x1) is invisible from outside; every launcher records the name of the instantiation it picked, so a parity test can
// assert that the kernel it checked is the kernel the B=64 training step runs.
void dm_set_variant(const char* name) { g_last_variant = name; }
extern "C" const char* dm_last_variant(void) { return g_last_variant; }

DmVariantName::DmVariantName(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(s, sizeof(s), fmt, ap);
  va_end(ap);
}
