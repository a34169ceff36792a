// ABI identity and launch-error diagnostics.
#include "dm_common.h"
static thread_local int g_last_hip_error = 0;
extern "C" void dm_note_hip_error(int code) { g_last_hip_error = code; }
extern "C" int dm_abi_version(void) { return 1; }
extern "C" const char* dm_target_arch(void) { return "gfx950"; }
extern "C" const char* dm_last_hip_error(void) {
  return g_last_hip_error ? hipGetErrorString((hipError_t)g_last_hip_error) : "";
}
