// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dm_hip.h"

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

extern "C" void dm_note_hip_error(int code);  // dm_api.hip: remembered for dm_last_hip_error()
#define DM_LAUNCH_CHECK()                                   \
  do {                                                      \
    const hipError_t dm_e_ = hipGetLastError();             \
    if (dm_e_ != hipSuccess) {                              \
      dm_note_hip_error((int)dm_e_);                        \
      return DM_ERR_LAUNCH;                                 \
    }                                                       \
  } while (0)

// Every entry point validates its arguments with DM_REQUIRE before its first launch; the macro also
// drains the thread's sticky "last error" (the host framework's own benign failures, e.g.
// hipErrorNotReady from an event query, would otherwise be mis-reported by DM_LAUNCH_CHECK).
#define DM_REQUIRE(cond)            \
  do {                              \
    (void)hipGetLastError();        \
    if (!(cond)) return DM_ERR_ARG; \
  } while (0)

static inline int dm_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float dm_silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float dm_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

// wave64 butterfly sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Counter-based RNG (splitmix64 finaliser over (seed, index)); returns 32 uniform bits.
// Used for dropout masks so the backward can regenerate the forward's mask from (seed, index)
// without storing it.  (torch's Philox stream cannot be matched bit-for-bit)
__device__ __forceinline__ uint32_t dm_rng32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
// keep-probability threshold: keep iff rng32 >= p * 2^32
__device__ __forceinline__ bool dm_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
  return dm_rng32(seed, idx) >= thresh;
}
