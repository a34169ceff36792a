// ABI identity, launch-error diagnostics and the "which kernel variant ran" debug hook.
#include "dm_common.h"

#include <stdarg.h>
#include <stdio.h>

static thread_local int g_last_hip_error = 0;
static thread_local const char* g_last_variant = "";
extern "C" void dm_note_hip_error(int code) { g_last_hip_error = code; }
extern "C" int dm_abi_version(void) { return 4; }   // 4: fp32 compute mode entry points (dm_f32.hip); 2: dm_conv_args / dm_wgrad_args grew norm_scale, norm_shift; 3: dm_attn_args grew workspace, ws_bytes
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

// Device-side dropout counter (see dm_common.h: dm_eff_seed).  Library-global on purpose: every kernel that draws a
// dropout mask must see the same counter within one micro-step (forward and backward regenerate the same masks).
static const uint32_t* g_drop_ctr = nullptr;
const uint32_t* dm_dropout_counter() { return g_drop_ctr; }
extern "C" int dm_set_dropout_counter(const unsigned int* dev_counter) {
  g_drop_ctr = dev_counter;
  return DM_OK;
}
namespace {
__global__ void bump_counter_kernel(unsigned int* c) {
  if (threadIdx.x == 0) c[0] = c[0] + 1u;
}
}  // namespace
extern "C" int dm_bump_dropout_counter(unsigned int* dev_counter, dm_stream_t stream) {
  DM_REQUIRE(dev_counter);
  hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dev_counter);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
