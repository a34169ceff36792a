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

// dm_api.hip: name of the kernel instantiation the calling thread launched last (dm_last_variant()).  A launcher
// formats its name once (function-local static DmVariantName) and records the pointer per launch.
struct DmVariantName {
  char s[96];
  DmVariantName(const char* fmt, ...) __attribute__((format(printf, 2, 3)));
};
void dm_set_variant(const char* name);

// Per-DEVICE once-flag for hipFuncSetAttribute: a function attribute belongs to the device's code object, so a process
// that drives several GPUs must set it on each.  `mask` is a function-local static word, one bit per device ordinal.
static inline bool dm_first_on_device(unsigned long long& mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;   // unknown ordinal: set every time
  const unsigned long long bit = 1ull << dev;
  if (__atomic_load_n(&mask, __ATOMIC_RELAXED) & bit) return false;
  __atomic_fetch_or(&mask, bit, __ATOMIC_RELAXED);   // (the caller sets the attribute right after; idempotent under a race)
  return true;
}

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
// The mask of element (row, col) is a pure function of (seed, row, col): two 32-bit
// multiply-xorshift stages; one 32-bit draw serves the column PAIR (col>>1) with 16 bits each, so a
// lane that owns consecutive columns pays ~4 VALU ops per element (the 64-bit splitmix used first
// cost ~30 and dominated the attention kernels).  keep iff bits16 >= thresh16 = round(p * 65536).
__device__ __forceinline__ uint32_t dm_drop_row(uint32_t seed, uint32_t row) {
  uint32_t x = row * 0x9E3779B1u + seed;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  return x;
}
__device__ __forceinline__ uint32_t dm_drop_mix(uint32_t x) {
  // one multiply/xorshift round: v_mul_lo_u32 is quarter rate, and these bits are drawn once per two
  // score elements inside the exp-bound attention loops
  x ^= x >> 16;
  x *= 0x21F0AAADu;
  x ^= x >> 15;
  return x;
}
__device__ __forceinline__ uint32_t dm_drop_pairkey(uint32_t pair) { return pair * 0x9E3779B9u; }  // Weyl step
__device__ __forceinline__ uint32_t dm_drop_bits(uint32_t rowstate, uint32_t pair) {
  return dm_drop_mix(rowstate + dm_drop_pairkey(pair));
}
// Attention-probability dropout draws its 32-bit word for (row, column pair) as rowword ^ colword: the row
// word is dm_drop_row(seed, row), the column word below depends on (seed, pair) only.  Marginals are uniform
// and any two cells are independent (pairwise independence); what is given up against a full hash per cell is
// higher-order independence (the four words on a rectangle of the row x pair grid XOR to zero), in exchange
// for ONE v_xor per two score elements inside the VALU-bound attention loops.
__device__ __forceinline__ uint32_t dm_attn_colword(uint32_t seed, uint32_t pair) {
  return dm_drop_mix((seed ^ 0x85EBCA6Bu) + dm_drop_pairkey(pair));
}
__device__ __forceinline__ bool dm_keep16(uint32_t bits32, uint32_t col, uint32_t thresh16) {
  // 16-bit field (col & 1) of bits32 >= thresh16, compared in the high half (no field extraction)
  return ((col & 1u) ? bits32 : (bits32 << 16)) >= (thresh16 << 16);
}
// Dropout seeds under HIP-graph replay: a captured launch replays its kernel arguments, so a host-side seed alone
// would repeat the same mask on every replay.  Kernels therefore add a DEVICE counter (dm_set_dropout_counter; the
// caller bumps it once per micro-step, dm_bump_dropout_counter) to the hashed host seed.  NULL = no counter.
__device__ __forceinline__ uint32_t dm_eff_seed(uint32_t seed, const uint32_t* __restrict__ ctr) {
  return ctr ? seed + ctr[0] * 0x9E3779B1u : seed;
}
const uint32_t* dm_dropout_counter();   // dm_api.hip: the registered device counter (or NULL)
// host side of the same convention
static inline uint32_t dm_thresh16(float p) { return p > 0.f ? (uint32_t)((double)p * 65536.0 + 0.5) : 0u; }
static inline float dm_inv_keep(float p) { return p > 0.f ? 65536.0f / (65536.0f - (float)dm_thresh16(p)) : 1.f; }
static inline uint32_t dm_seed32(unsigned long long s) { return (uint32_t)(s ^ (s >> 32)) * 0x9E3779B1u + 0x7F4A7C15u; }
