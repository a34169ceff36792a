// Flash-style fused attention (forward + backward) for the UNet's self- and cross-attention:
// 8 heads, head_dim in {8,16,32} (flash / few-token kernels) or {64,128} (wide-head kernels), Nq in {576,2304,9216}, Nk = Nq (self) or the number of
// context tokens (cross; 1 for the CLIP table).  No N x N matrix ever reaches memory.
//
// Register-resident P: the score tile is computed TRANSPOSED (S^T = K Q^T, keys on MFMA rows,
// queries on the lane axis), so after the online softmax each lane already holds, for its
// query, exactly the 8 keys per 32-key step that the P operand of the second MFMA
// (O^T = V^T P^T) wants -- up to a fixed permutation of the reduction index, which is applied
// to the V^T operand instead, for free, through the row addresses of ds_read_b64_tr_b16.
// The same trick (with q and key swapped) serves dV/dK, and again dQ.
//
// With head dims this small the kernels are VALU-issue-bound, not MFMA-bound (64 MFMA flop per
// score element); the flash kernels therefore push per-element work onto the MFMA pipe (see
// "shared pieces of the flash kernels" below and DESIGN.md section 3).  Three kernel families by
// key count: one context token (streaming kernels, the softmax is identically 1), 2..16 tokens
// (VALU kernels), otherwise the flash kernels.
//
// Reference semantics replaced: F.scaled_dot_product_attention(q,k,v,dropout_p) with scale
// 1/sqrt(hd), non-causal (unet.py:119-126,136,165-172,214) and its autograd backward.
#include "dm_common.h"
#include <type_traits>

namespace {

constexpr int RS = 96;         // LDS row stride (bytes) of a [rows][32 d] bf16 block: 64 B + 32 pad makes the ds_read_b128 lane groups
                               // ({0-3,12-15,20-27},...) and the ds_read_b64_tr_b16 halves conflict-free (80 B is 2-way on both)
constexpr float NEG = -1e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct AttnParams {
  const bf16* q; long ldq; const bf16* k; long ldk; const bf16* v; long ldv;
  bf16* o; long ldo; float* lse;
  const bf16* d_o; long lddo; bf16* dq; long lddq; bf16* dk; long lddk; bf16* dv; long lddv;
  float* delta;
  int B, H, Nq, Nk, hd;
  float scale;
  uint32_t seed; uint32_t thresh; float inv_keep;  // dm_common.h dropout convention; thresh = 2*t15
  const uint32_t* seed_ctr;                        // device-side seed counter (graph replay) or null
  uint32_t t15, tm1x2; float keep;  // flash kernels: 15-bit threshold, (t15-1) in both halves, 1-p
};

__device__ __forceinline__ bf16x8 tr_pair(const char* a0, const char* a1) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = (bf16)a[j]; r[4 + j] = (bf16)b[j]; }
  return r;
}
__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16)0.f;
  return r;
}
// row fragment straight from global: 8 head-dim elements [8g, 8g+8) of token `tok`
__device__ __forceinline__ bf16x8 load_row_frag(const bf16* base, long ld, long tok, bool tok_ok, int hoff, int hd,
                                                int g) {
  if (!tok_ok || 8 * g >= hd) return zero8();
  return *reinterpret_cast<const bf16x8*>(base + tok * ld + hoff + 8 * g);
}
// One 64-row x 32 (padded) head-dim block of one head travels global -> registers -> LDS, one 16-byte piece per
// thread (head dim 32: four pieces per row).  Loads are issued two blocks ahead (software pipelining).
__device__ __forceinline__ void store_piece(char* dst, int tid, u32x4 v) {
  *reinterpret_cast<u32x4*>(dst + (tid >> 2) * RS + (tid & 3) * 16) = v;
}
// A staged block PAIR (K+V, or Q+dO) in registers.  Head dims <= 16 (DT == 1) fill only two 16-byte pieces per
// row, so the 256 threads cover both tensors with ONE load each (threads 0-127 the first tensor, 128-255 the
// second) and bytes 32.. of the LDS rows are never written -- nor read: every row read of that variant stays
// inside bytes 0..31.  Head dim 32 takes two loads per thread.
template <int DT> struct Stage { u32x4 a, b; };
// The loads go through buffer descriptors (one per tensor, based at this (batch, head) slice): a 32-bit offset
// per thread, and rows past the end or pieces past the head dim come back as zeros from the range check --
// no 64-bit address arithmetic and no selects in the block loop.
template <int DT> struct Stager {
  __amdgpu_buffer_rsrc_t ra, rb;        // DT == 1: ra is this wave's tensor (waves 0-1 the first, 2-3 the second)
  uint32_t va, vb, stepa, stepb;        // byte offset of this thread's piece in block 0; bytes per 64-row block
};
template <int DT>
__device__ __forceinline__ Stager<DT> make_stager(const bf16* A, long lda, const bf16* B, long ldb, long tokbase,
                                                  int n_tok, int hoff, int hd, int tid, int wave) {
  Stager<DT> s;
  constexpr uint32_t OOB = 0x7FFFFFFFu;  // beyond any descriptor's range, also after the block steps are added
  if constexpr (DT == 1) {
    // wave-uniform, and SAID so: with `wave` derived from threadIdx the descriptor is a per-lane value to the compiler,
    // every buffer load becomes a waterfall loop (v_readfirstlane + s_cbranch_execnz), and the control flow makes the
    // waitcnt pass drain ALL loads (vmcnt(0)) where the two-blocks-ahead prefetch needs a counted wait
    const bool second = __builtin_amdgcn_readfirstlane(wave) >= 2;
    const bf16* base = (second ? B + tokbase * ldb : A + tokbase * lda) + hoff;
    const long ld = second ? ldb : lda;
    const int nrec = __builtin_amdgcn_readfirstlane((int)(((long)(n_tok - 1) * ld + hd) * 2));
    s.ra = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nrec, 0x00020000);
    s.rb = s.ra;
    const int row = (tid & 127) >> 1, pc = tid & 1;
    s.va = pc * 8 < hd ? (uint32_t)(row * ld * 2 + pc * 16) : OOB;
    s.vb = s.va;
    s.stepa = (uint32_t)(64 * ld * 2);
    s.stepb = s.stepa;
  } else {
    s.ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + tokbase * lda + hoff), 0,
                                             (int)(((long)(n_tok - 1) * lda + hd) * 2), 0x00020000);
    s.rb = __builtin_amdgcn_make_buffer_rsrc((void*)(B + tokbase * ldb + hoff), 0,
                                             (int)(((long)(n_tok - 1) * ldb + hd) * 2), 0x00020000);
    const int row = tid >> 2, pc = tid & 3;
    s.va = pc * 8 < hd ? (uint32_t)(row * lda * 2 + pc * 16) : OOB;
    s.vb = pc * 8 < hd ? (uint32_t)(row * ldb * 2 + pc * 16) : OOB;
    s.stepa = (uint32_t)(64 * lda * 2);
    s.stepb = (uint32_t)(64 * ldb * 2);
  }
  return s;
}
template <int DT>
__device__ __forceinline__ void stage_load(Stage<DT>& r, const Stager<DT>& s, int blk) {
  r.a = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(s.ra, (int)(s.va + blk * s.stepa), 0, 0));
  if constexpr (DT == 2)
    r.b = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(s.rb, (int)(s.vb + blk * s.stepb), 0, 0));
}
template <int DT>
__device__ __forceinline__ void stage_store(const Stage<DT>& s, char* sA, char* sB, int tid) {
  if constexpr (DT == 1) {
    *reinterpret_cast<u32x4*>((tid >= 128 ? sB : sA) + ((tid & 127) >> 1) * RS + (tid & 1) * 16) = s.a;
  } else {
    store_piece(sA, tid, s.a);
    store_piece(sB, tid, s.b);
  }
}

// ---------------------------------------------------------------- shared pieces of the flash kernels
// With head dims this small the loops are bound by VALU issue, so per-element work is moved into the MFMAs,
// whose pipe is two-thirds idle:
//  (1) The lane-owned score operand (the query in fwd / bwd_q, the key in bwd_kv) is pre-multiplied by
//      c = scale*log2(e) in fp32 and split into bf16 hi + lo parts, so the MFMA returns c*s to ~2^-17.  With
//      head dims <= 16 the lo part rides in k-slots 16..31 of the 16x16x32 MFMA, which zero padding would
//      waste (the LDS-side operand re-reads d 0..15 there); with head dim 32 it is a second MFMA.
//  (2) The accumulator starts at -m (fwd: the running reference maximum) or -LSE (bwd), so the probability
//      is exp2(acc) with no per-element fma; dP leaves its MFMA as dP - delta*keep the same way.
//  (3) fwd rescales lazily: the reference maximum of a query moves only when a block exceeds it by more than
//      2^TAU (wave-uniform branch), so the common block does max / exp / add / pack and nothing else.
//  (4) Dropout compares 15-bit fields (row word ^ column word, both pre-masked): fwd builds the keep mask of
//      a key PAIR with v_pk_sub_i16 + v_pk_ashrrev_i16 and ANDs it onto the packed bf16 pair; the backward
//      kernels compare 16-bit halves in place (SDWA operand select, no shifts).
constexpr uint32_t M15 = 0x7FFF7FFFu;
constexpr float TAU = 8.f;
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t keepmask15x2(uint32_t x, uint32_t tm1x2) {
  const s16x2 d = __builtin_bit_cast(s16x2, tm1x2) - __builtin_bit_cast(s16x2, x);  // < 0 iff field >= t15
  return __builtin_bit_cast(uint32_t, d >> 15);
}
__device__ __forceinline__ bool keep15(uint32_t x, int hi, uint32_t t15) {
  return (uint16_t)(hi ? (x >> 16) : x) >= (uint16_t)t15;
}
template <int DT>
__device__ __forceinline__ void load_scaled_frag(const bf16* base, long ld, long tok, bool tok_ok, int hoff, int hd,
                                                 int g, float c, bf16x8& f0, bf16x8& f1) {
  const bf16x8 raw = load_row_frag(base, ld, tok, tok_ok, hoff, hd, DT == 1 ? (g & 1) : g);
  bf16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float s = c * (float)raw[j];
    hi[j] = (bf16)s;
    lo[j] = (bf16)(s - (float)hi[j]);
  }
  if (DT == 1) { f0 = g < 2 ? hi : lo; f1 = f0; } else { f0 = hi; f1 = lo; }
}
template <int DT>
__device__ __forceinline__ f32x4 score_mfma(bf16x8 a, bf16x8 b0, bf16x8 b1, f32x4 init) {
  f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, init, 0, 0, 0);
  if (DT == 2) s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, s, 0, 0, 0);
  return s;
}

// ---------------------------------------------------------------- forward
// DROP and EDGE (Nk not a multiple of 64: key-range masks, applied in every block of that rarer variant) are
// compile-time: the compiler if-converts runtime `if (p.thresh)` / `if (edge)` guards into unconditional hash +
// select code, which the eval path then pays for.
template <int DT, bool DROP, bool EDGE>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(const AttnParams p) {
  [[maybe_unused]] const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  __shared__ __attribute__((aligned(16))) char sKb[2][64 * RS];
  __shared__ __attribute__((aligned(16))) char sVb[2][64 * RS];
  __shared__ __attribute__((aligned(16))) uint32_t sCb[2][32];  // dropout column words of the staged key block
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  const int gk = DT == 1 ? (g & 1) : g;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H, hoff = h * p.hd;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float c = p.scale * LOG2E;
  const long qbase = (long)b * p.Nq, kbase = (long)b * p.Nk;

  bf16x8 qf0[2], qf1[2];
  int qi[2];
  uint32_t rs[2] = {0u, 0u};
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    qi[qt] = q0 + qt * 16 + li;
    load_scaled_frag<DT>(p.q, p.ldq, qbase + qi[qt], qi[qt] < p.Nq, hoff, p.hd, g, c, qf0[qt], qf1[qt]);
    if constexpr (DROP) rs[qt] = dm_drop_row(seed_eff, (uint32_t)(bh * p.Nq + qi[qt])) & M15;
  }
  float m[2] = {0.f, 0.f};  // reference maximum (exp2 domain), set by the first block
  // row sums of P ride on the MFMA pipe too: an all-ones A operand makes every row of lacc the sum over the
  // block's keys of the (bf16, undropped) probabilities -- no per-element add, no cross-lane reduction
  f32x4 lacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.f;
  f32x4 acc[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) acc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkb = (p.Nk + 63) / 64;
  // K/V blocks are prefetched two blocks ahead into registers and one block ahead into the other LDS
  // buffer: one barrier per key block, no exposed global latency.
  Stage<DT> r0, r1;
  const Stager<DT> stg = make_stager<DT>(p.k, p.ldk, p.v, p.ldv, kbase, p.Nk, hoff, p.hd, tid, wave);
  stage_load<DT>(r0, stg, 0);
  r1 = r0;
  if (nkb > 1) stage_load<DT>(r1, stg, 1);
  stage_store<DT>(r0, sKb[0], sVb[0], tid);
  if (DROP && wave == 0 && tid < 32) sCb[0][tid] = dm_attn_colword(seed_eff, (uint32_t)tid) & M15;
  __syncthreads();
  auto block = [&](int kb, Stage<DT>& r_free, const Stage<DT>& r_next) {
    stage_load<DT>(r_free, stg, kb + 2);   // unconditional: past the end the descriptor's range check returns zeros (no
                                           // traffic), and a branch here would make the waitcnt pass drain the prefetch
    const char* sK = sKb[kb & 1];
    const char* sV = sVb[kb & 1];
    f32x4 st[4][2];  // c*s - m, straight out of the MFMA
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (kt * 16 + li) * RS + gk * 16);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float nm = -m[qt];
        st[kt][qt] = score_mfma<DT>(kf, qf0[qt], qf1[qt], f32x4{nm, nm, nm, nm});
      }
    }
    float bm[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      if constexpr (EDGE) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kb * 64 + kt * 16 + 4 * g + r >= p.Nk) st[kt][qt][r] = NEG;
      }
      float t = NEG;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) t = fmaxf(t, st[kt][qt][r]);
      t = fmaxf(t, __shfl_xor(t, 16, 64));
      bm[qt] = fmaxf(t, __shfl_xor(t, 32, 64));
    }
    // lazy rescale: move a query's reference maximum only in the first block or when this block tops it by > TAU
    const bool first = kb == 0;
    if (__builtin_amdgcn_ballot_w64(first || bm[0] > TAU || bm[1] > TAU) != 0ull) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float sh = (first || bm[qt] > TAU) ? bm[qt] : 0.f;
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-sh);  // first block: l = acc = 0 (and -sh may be huge)
        m[qt] += sh;
        lacc[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt][qt] *= alpha;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[kt][qt][r] -= sh;
      }
    }
    // Per-element VALU budget (the bound of these small-head kernels): half a max3, v_exp, half a cvt_pk
    // (+ with dropout half each of xor, pk_sub, pk_ashr, and).  v_exp_f32 is issued raw (no denormal range
    // fix-up: p < 2^-126 is 0 anyway).
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[kt][qt][r] = __builtin_amdgcn_exp2f(st[kt][qt][r]);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 pb[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        pb[qt] = pack8(st[2 * ks][qt], st[2 * ks + 1][qt]);
        lacc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[qt], lacc[qt], 0, 0, 0);
        if constexpr (DROP) {
          const u32x2 ca = *reinterpret_cast<const u32x2*>(sCb[kb & 1] + (2 * ks) * 8 + 2 * g);
          const u32x2 cb = *reinterpret_cast<const u32x2*>(sCb[kb & 1] + (2 * ks + 1) * 8 + 2 * g);
          u32x4 w = __builtin_bit_cast(u32x4, pb[qt]);
          w[0] &= keepmask15x2(rs[qt] ^ ca[0], p.tm1x2);
          w[1] &= keepmask15x2(rs[qt] ^ ca[1], p.tm1x2);
          w[2] &= keepmask15x2(rs[qt] ^ cb[0], p.tm1x2);
          w[3] &= keepmask15x2(rs[qt] ^ cb[1], p.tm1x2);
          pb[qt] = __builtin_bit_cast(bf16x8, w);
        }
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int qq = li >> 2, pp = li & 3;
        const char* a0 = sV + (ks * 32 + 4 * g + qq) * RS + dt * 32 + pp * 8;
        const bf16x8 vT = tr_pair(a0, a0 + 16 * RS);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vT, pb[qt], acc[dt][qt], 0, 0, 0);
      }
    }
    if (kb + 1 < nkb) {
      stage_store<DT>(r_next, sKb[(kb + 1) & 1], sVb[(kb + 1) & 1], tid);
      if (DROP && wave == 0 && tid < 32) sCb[(kb + 1) & 1][tid] = dm_attn_colword(seed_eff, (uint32_t)((kb + 1) * 32 + tid)) & M15;
    }
    __syncthreads();
  };
  {  // block pairs, an odd last block peeled off: a mid-loop exit is one more path into the loop header on which the
     // youngest prefetch is still in flight, and the waitcnt pass then drains it at the header on every trip
    int kb = 0;
    for (; kb + 1 < nkb; kb += 2) {
      block(kb, r0, r1);
      block(kb + 1, r1, r0);
    }
    if (kb < nkb) block(kb, r0, r1);
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float lt = lacc[qt][0];
    const float inv = (DROP ? p.inv_keep : 1.f) / lt;  // the dropout rescale rides in the normalisation
    if (qi[qt] < p.Nq) {
      if (g == 0 && p.lse) p.lse[(long)bh * p.Nq + qi[qt]] = (m[qt] + log2f(lt)) * LN2;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + 4 * g;
        if (d < p.hd) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16)(acc[dt][qt][r] * inv);
          *reinterpret_cast<bf16x4*>(p.o + (qbase + qi[qt]) * p.ldo + hoff + d) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------- backward
// delta[bh][q] = sum_d dO[q][d] * O[q][d].  thread = (token, 8 channels): coalesced 16-byte loads of whole rows,
// the hd/8 threads of a head combine with shuffles.
__global__ __launch_bounds__(256) void attn_delta_rows_kernel(const AttnParams p) {
  [[maybe_unused]] const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  const int C = p.H * p.hd, TPQ = C >> 3, QPB = 256 / TPQ, sub = p.hd >> 3;
  const int tid = threadIdx.x, c8 = (tid % TPQ) * 8, qs = tid / TPQ, h = c8 / p.hd;
  const long rows = (long)p.B * p.Nq;
  for (long row = (long)blockIdx.x * QPB + qs; row < rows; row += (long)gridDim.x * QPB) {
    const bf16x8 ov = *reinterpret_cast<const bf16x8*>(p.o + row * p.ldo + c8);
    const bf16x8 dv = *reinterpret_cast<const bf16x8*>(p.d_o + row * p.lddo + c8);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += (float)ov[t] * (float)dv[t];
    for (int o2 = 1; o2 < sub; o2 <<= 1) s += __shfl_xor(s, o2, 64);  // sub = 1..16 threads per head
    if ((tid % sub) == 0) {
      const long b = row / p.Nq, q = row % p.Nq;
      p.delta[(b * p.H + h) * p.Nq + q] = s;
    }
  }
}
// generic shapes: thread = (batch, head, query)
__global__ void attn_delta_kernel(const AttnParams p) {
  [[maybe_unused]] const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  const long total = (long)p.B * p.H * p.Nq;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % p.Nq);
    const int bh = (int)(i / p.Nq);
    const int b = bh / p.H, h = bh % p.H;
    const bf16* o = p.o + ((long)b * p.Nq + q) * p.ldo + h * p.hd;
    const bf16* d = p.d_o + ((long)b * p.Nq + q) * p.lddo + h * p.hd;
    float s = 0.f;
    for (int j = 0; j < p.hd; j += 8) {
      const bf16x8 ov = *reinterpret_cast<const bf16x8*>(o + j);
      const bf16x8 dv = *reinterpret_cast<const bf16x8*>(d + j);
#pragma unroll
      for (int t = 0; t < 8; ++t) s += (float)ov[t] * (float)dv[t];
    }
    p.delta[i] = s;
  }
}

// dK, dV: one wave = 16 keys (MFMA columns), workgroup = 64 keys, loop over query blocks of 64.
// With dropout mask Z and keep = 1-p:  dS = P o (Z o dP / keep - delta) = (P o (Z o dP - delta*keep)) / keep;
// the 1/keep rides in the dK epilogue, delta*keep in the dP accumulator's start value.
template <int DT, bool DROP>
__global__ __launch_bounds__(256, DT == 1 ? 4 : 3) void attn_bwd_kv_kernel(const AttnParams p) {
  [[maybe_unused]] const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  __shared__ __attribute__((aligned(16))) char sQb[2][64 * RS];
  __shared__ __attribute__((aligned(16))) char sOb[2][64 * RS];
  __shared__ __attribute__((aligned(16))) float sLb[2][64], sDb[2][64];  // -LSE*log2e, -delta*keep
  // per-query dropout row words: [0..63] as is, [ROFF..ROFF+63] >> 16.  ROFF = 72 (not 64): the lanes of one
  // ds_read_b128 group read the even-key and the odd-key half at the same query offset; 64 dwords apart they hit the
  // same banks (2-way conflict on every row-word read: 13 % of this kernel's LDS cycles in round 1's PMC run)
  constexpr int ROFF = 72;
  __shared__ __attribute__((aligned(16))) uint32_t sRb[2][ROFF + 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  const int gk = DT == 1 ? (g & 1) : g;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H, hoff = h * p.hd;
  const int key = blockIdx.x * 64 + wave * 16 + li;
  const bool key_ok = key < p.Nk;
  const float c = p.scale * LOG2E;
  const long qbase = (long)b * p.Nq, kbase = (long)b * p.Nk;
  bf16x8 kf0, kf1;
  load_scaled_frag<DT>(p.k, p.ldk, kbase + key, key_ok, hoff, p.hd, g, c, kf0, kf1);
  const bf16x8 vfrag = load_row_frag(p.v, p.ldv, kbase + key, key_ok, hoff, p.hd, g);
  f32x4 dkT[DT], dvT[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { dkT[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvT[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nqb = (p.Nq + 63) / 64;
  // this lane's 15-bit column field, in the low half; the row words come from the matching half-array of sRb
  uint32_t cwl = 0u;
  if constexpr (DROP) {
    const uint32_t cw = dm_attn_colword(seed_eff, (uint32_t)key >> 1) & M15;
    cwl = (key & 1) ? (cw >> 16) : cw;
  }
  const int rpar = (key & 1) * ROFF;
  const float keepf = DROP ? p.keep : 1.f;
  // Q / dO / LSE / delta blocks: two blocks ahead in registers, one ahead in the other LDS buffer.
  const Stager<DT> stg = make_stager<DT>(p.q, p.ldq, p.d_o, p.lddo, qbase, p.Nq, hoff, p.hd, tid, wave);
  struct Regs { Stage<DT> st; float l, d; uint32_t rs; bool ok; };
  auto fetch = [&](int qb, Regs& r) {
    stage_load<DT>(r.st, stg, qb);
    if (wave == 0) {  // wave 0 carries the per-query scalars
      const int q = qb * 64 + tid;
      const bool ok = q < p.Nq;
      const long li_ = (long)bh * p.Nq + (ok ? q : 0);
      // RAW values only: any arithmetic on them here makes wave 0 wait for the loads (and, vmcnt being in order, for the
      // block data ahead of them) before the block it should overlap -- and the other three waves wait for wave 0 at
      // the barrier.  The scaling happens in put(), two blocks later.
      r.l = p.lse[li_]; r.d = p.delta[li_]; r.ok = ok;
      if (DROP) r.rs = dm_drop_row(seed_eff, (uint32_t)(bh * p.Nq + q)) & M15;
    }
  };
  auto put = [&](int buf, const Regs& r) {
    stage_store<DT>(r.st, sQb[buf], sOb[buf], tid);
    if (wave == 0) {
      sLb[buf][tid] = r.ok ? -(r.l * LOG2E) : -1e30f;  // out-of-range query: p = exp2(-huge) = 0
      sDb[buf][tid] = r.ok ? -r.d * keepf : 0.f;
      if (DROP) { sRb[buf][tid] = r.rs; sRb[buf][ROFF + tid] = r.rs >> 16; }
    }
  };
  Regs r0{}, r1{};
  fetch(0, r0);
  if (nqb > 1) fetch(1, r1); else r1 = r0;
  put(0, r0);
  __syncthreads();
  auto block = [&](int qb, Regs& r_free, const Regs& r_next) {
    fetch(qb + 2, r_free);   // unconditional (clamped scalars, range-checked block data): see stage_load in the forward
    const char* sQ = sQb[qb & 1];
    const char* sO = sOb[qb & 1];
    const float* sL = sLb[qb & 1];
    const float* sD = sDb[qb & 1];
    f32x4 pt[4], ds[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const bf16x8 qa = *reinterpret_cast<const bf16x8*>(sQ + (qt * 16 + li) * RS + gk * 16);
      const bf16x8 da = *reinterpret_cast<const bf16x8*>(sO + (qt * 16 + li) * RS + gk * 16);  // vfrag is zero in k-slots >= hd
      const f32x4 L4 = *reinterpret_cast<const f32x4*>(sL + qt * 16 + 4 * g);
      const f32x4 D4 = *reinterpret_cast<const f32x4*>(sD + qt * 16 + 4 * g);
      const f32x4 s = score_mfma<DT>(qa, kf0, kf1, L4);                        // c*s - LSE
      const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vfrag, D4, 0, 0, 0);  // dP - delta*keep
      u32x4 R4 = {0u, 0u, 0u, 0u};  // per-query dropout row fields
      if constexpr (DROP) R4 = *reinterpret_cast<const u32x4*>(sRb[qb & 1] + rpar + qt * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // no per-element masks: invalid queries carry -LSE = -1e30 (p = 0); invalid keys only pollute their
        // own (never stored) dK/dV columns
        const float pv = __builtin_amdgcn_exp2f(s[r]);
        float u = pv, t = dp[r];
        if constexpr (DROP) {
          const bool keep = keep15(R4[r] ^ cwl, 0, p.t15);
          u = keep ? pv : 0.f;
          t = keep ? t : D4[r];
          asm("" : "+v"(u));  // keep the select in fp32: one cvt_pk per pair below instead of cvt + select + perm
        }
        pt[qt][r] = u;
        ds[qt][r] = pv * t;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 pB = pack8(pt[2 * ks], pt[2 * ks + 1]);
      const bf16x8 dsB = pack8(ds[2 * ks], ds[2 * ks + 1]);
      const int qq = li >> 2, pp = li & 3;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int off = (ks * 32 + 4 * g + qq) * RS + dt * 32 + pp * 8;
        const bf16x8 doT = tr_pair(sO + off, sO + off + 16 * RS);
        dvT[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(doT, pB, dvT[dt], 0, 0, 0);
        const bf16x8 qT = tr_pair(sQ + off, sQ + off + 16 * RS);
        dkT[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT, dsB, dkT[dt], 0, 0, 0);
      }
    }
    if (qb + 1 < nqb) put((qb + 1) & 1, r_next);
    __syncthreads();
  };
  {  // block pairs, an odd last block peeled off: a mid-loop exit is one more path into the loop header on which the
     // youngest prefetch is still in flight, and the waitcnt pass then drains it at the header on every trip
    int qb = 0;
    for (; qb + 1 < nqb; qb += 2) {
      block(qb, r0, r1);
      block(qb + 1, r1, r0);
    }
    if (qb < nqb) block(qb, r0, r1);
  }
  const float ik = DROP ? p.inv_keep : 1.f;  // dropout rescale of P (dV) and of dS (dK) folded into the epilogue
  if (key_ok) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + 4 * g;
      if (d < p.hd) {
        bf16x4 ok, ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) { ok[r] = (bf16)(dkT[dt][r] * (p.scale * ik)); ov[r] = (bf16)(dvT[dt][r] * ik); }
        *reinterpret_cast<bf16x4*>(p.dk + (kbase + key) * p.lddk + hoff + d) = ok;
        *reinterpret_cast<bf16x4*>(p.dv + (kbase + key) * p.lddv + hoff + d) = ov;
      }
    }
  }
}

// dQ: mirrors the forward (one wave = 32 queries, workgroup = 128 queries, loop over key blocks of 64).
template <int DT, bool DROP, bool EDGE>
__global__ __launch_bounds__(256, DT == 1 ? 4 : 3) void attn_bwd_q_kernel(const AttnParams p) {
  [[maybe_unused]] const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  __shared__ __attribute__((aligned(16))) char sKb[2][64 * RS];
  __shared__ __attribute__((aligned(16))) char sVb[2][64 * RS];
  __shared__ __attribute__((aligned(16))) uint32_t sCb[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  const int gk = DT == 1 ? (g & 1) : g;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H, hoff = h * p.hd;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float c = p.scale * LOG2E;
  const long qbase = (long)b * p.Nq, kbase = (long)b * p.Nk;
  const float keepf = DROP ? p.keep : 1.f;
  bf16x8 qf0[2], qf1[2], dof[2];
  int qi[2];
  float Lq[2], Dq[2];  // -LSE*log2e, -delta*keep: the MFMA accumulators' start values
  uint32_t rsq[2] = {0u, 0u};
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    qi[qt] = q0 + qt * 16 + li;
    const bool ok = qi[qt] < p.Nq;
    load_scaled_frag<DT>(p.q, p.ldq, qbase + qi[qt], ok, hoff, p.hd, g, c, qf0[qt], qf1[qt]);
    dof[qt] = load_row_frag(p.d_o, p.lddo, qbase + qi[qt], ok, hoff, p.hd, g);
    Lq[qt] = ok ? -p.lse[(long)bh * p.Nq + qi[qt]] * LOG2E : -1e30f;  // out-of-range query: p = 0
    Dq[qt] = ok ? -p.delta[(long)bh * p.Nq + qi[qt]] * keepf : 0.f;
    if constexpr (DROP) rsq[qt] = dm_drop_row(seed_eff, (uint32_t)(bh * p.Nq + qi[qt])) & M15;
  }
  f32x4 dqT[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) dqT[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nkb = (p.Nk + 63) / 64;
  Stage<DT> r0, r1;
  const Stager<DT> stg = make_stager<DT>(p.k, p.ldk, p.v, p.ldv, kbase, p.Nk, hoff, p.hd, tid, wave);
  stage_load<DT>(r0, stg, 0);
  r1 = r0;
  if (nkb > 1) stage_load<DT>(r1, stg, 1);
  stage_store<DT>(r0, sKb[0], sVb[0], tid);
  if (DROP && wave == 0 && tid < 32) sCb[0][tid] = dm_attn_colword(seed_eff, (uint32_t)tid) & M15;
  __syncthreads();
  auto block = [&](int kb, Stage<DT>& r_free, const Stage<DT>& r_next) {
    stage_load<DT>(r_free, stg, kb + 2);   // unconditional: past the end the descriptor's range check returns zeros (no
                                           // traffic), and a branch here would make the waitcnt pass drain the prefetch
    const char* sK = sKb[kb & 1];
    const char* sV = sVb[kb & 1];
    f32x4 ds[4][2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (kt * 16 + li) * RS + gk * 16);
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + (kt * 16 + li) * RS + gk * 16);  // dof is zero in k-slots >= hd
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float nl = Lq[qt], nd = Dq[qt];
        const f32x4 s = score_mfma<DT>(kf, qf0[qt], qf1[qt], f32x4{nl, nl, nl, nl});
        const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[qt], f32x4{nd, nd, nd, nd}, 0, 0, 0);
        uint32_t db0 = 0u, db1 = 0u;
        if constexpr (DROP) {
          const u32x2 cw = *reinterpret_cast<const u32x2*>(sCb[kb & 1] + kt * 8 + 2 * g);
          db0 = rsq[qt] ^ cw[0]; db1 = rsq[qt] ^ cw[1];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb * 64 + kt * 16 + 4 * g + r;
          float pv = __builtin_amdgcn_exp2f(s[r]);
          if constexpr (EDGE) { if (key >= p.Nk) pv = 0.f; }  // key-range mask only in the rarer ragged variant
          float t = dp[r];
          if constexpr (DROP) t = keep15(r < 2 ? db0 : db1, r & 1, p.t15) ? t : nd;
          ds[kt][qt][r] = pv * t;
        }
      }
      // keep the 16 score / dP MFMA results of later key tiles out of registers (3 waves per SIMD)
      if constexpr (DROP) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int qq = li >> 2, pp = li & 3;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int off = (ks * 32 + 4 * g + qq) * RS + dt * 32 + pp * 8;
        const bf16x8 kT = tr_pair(sK + off, sK + off + 16 * RS);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          dqT[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT, pack8(ds[2 * ks][qt], ds[2 * ks + 1][qt]),
                                                                dqT[dt][qt], 0, 0, 0);
      }
    }
    if (kb + 1 < nkb) {
      stage_store<DT>(r_next, sKb[(kb + 1) & 1], sVb[(kb + 1) & 1], tid);
      if (DROP && wave == 0 && tid < 32) sCb[(kb + 1) & 1][tid] = dm_attn_colword(seed_eff, (uint32_t)((kb + 1) * 32 + tid)) & M15;
    }
    __syncthreads();
  };
  {  // block pairs, an odd last block peeled off: a mid-loop exit is one more path into the loop header on which the
     // youngest prefetch is still in flight, and the waitcnt pass then drains it at the header on every trip
    int kb = 0;
    for (; kb + 1 < nkb; kb += 2) {
      block(kb, r0, r1);
      block(kb + 1, r1, r0);
    }
    if (kb < nkb) block(kb, r0, r1);
  }
  const float osc = p.scale * (DROP ? p.inv_keep : 1.f);
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    if (qi[qt] >= p.Nq) continue;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + 4 * g;
      if (d < p.hd) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(dqT[dt][qt][r] * osc);
        *reinterpret_cast<bf16x4*>(p.dq + (qbase + qi[qt]) * p.lddq + hoff + d) = o;
      }
    }
  }
}

// ---------------------------------------------------------------- fused backward (head dim 16, 2304 keys per head)
// attn_bwd_kv_kernel and attn_bwd_q_kernel each recompute the probabilities (exp2, dropout compare / selects): with
// head dim 16 that VALU work IS the run time of both.  This kernel computes them once.  A workgroup of NW waves (4 as
// launched) owns a SLICE of the keys of one (batch, head) -- a quarter: every wave keeps 9 key tiles (144 keys) resident in registers -- the scaled
// hi/lo key fragment for the score MFMA, V^T for dP, K^T for dQ, and the dK^T / dV^T accumulators -- and the workgroup
// sweeps all queries in blocks of 64 (staged as in attn_bwd_kv_kernel).  Per (16 queries x 16 keys):
//   S, dP (MFMA) -> P, dS (VALU, once) -> dV^T += dO^T P, dK^T += Q^T dS  (P, dS are already B operands), and
//   dQ^T += K^T dS^T, whose B operand is the TRANSPOSE of the tile: 8 bytes per lane through a per-wave LDS tile
//   and one ds_read_b64_tr_b16 back.
// The dQ^T partial of a wave covers its 144 keys; the waves' partials are summed in wave order through LDS once per
// 64-query block, and the workgroups of a (batch, head) -- one per key split -- leave one fp32 partial array each, which
// attn_dq_merge_kernel adds in split order (fixed order everywhere: bit-reproducible).  All 16x16x16 MFMAs except the score (hi/lo split in k-slots 16-31).
constexpr int FB_NK = 2304;                    // keys per head this kernel family is built for

__device__ __forceinline__ s16x4 tr16(const char* a) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pair(float a, float b) {   // one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ s16x4 pack4(const float (&v)[4]) {
  return __builtin_bit_cast(s16x4, u32x2{cvt_pair(v[0], v[1]), cvt_pair(v[2], v[3])});
}

// NW waves per workgroup, FB_KT key tiles (16 keys) per wave: a workgroup owns NW * FB_KT * 16 keys, and
// FB_NK / that many workgroups (key splits) share one (batch, head).
template <bool DROP, int NW, int FB_KT>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_fused_kernel(const AttnParams p, float* __restrict__ dq_part) {
  constexpr int DT = 1;
  constexpr int FB_KEYS = NW * FB_KT * 16;
  [[maybe_unused]] const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);
  constexpr int ROFF = 72;                                       // see attn_bwd_kv_kernel
  extern __shared__ __attribute__((aligned(16))) char fb_smem[];
  char (*sQb)[64 * RS] = reinterpret_cast<char (*)[64 * RS]>(fb_smem);                     // [2]
  char (*sOb)[64 * RS] = reinterpret_cast<char (*)[64 * RS]>(fb_smem + 2 * 64 * RS);       // [2]
  float (*sLb)[64] = reinterpret_cast<float (*)[64]>(fb_smem + 4 * 64 * RS);               // [2]
  float (*sDb)[64] = reinterpret_cast<float (*)[64]>(fb_smem + 4 * 64 * RS + 512);         // [2]
  uint32_t (*sRb)[ROFF + 64] = reinterpret_cast<uint32_t (*)[ROFF + 64]>(fb_smem + 4 * 64 * RS + 1024);   // [2]
  constexpr int OFF_T = 4 * 64 * RS + 1024 + 2 * (ROFF + 64) * 4;
  static_assert(OFF_T % 16 == 0, "alignment of the transpose tiles");
  char (*sT)[1024] = reinterpret_cast<char (*)[1024]>(fb_smem + OFF_T);                    // [NW] two [16 keys][16 queries] bf16 tiles per wave
  float (*sDQ)[64 * 16] = reinterpret_cast<float (*)[64 * 16]>(fb_smem + OFF_T + NW * 1024);   // [NW] per-wave dQ partial of the query block
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  const int gk = g & 1;
  const int split = blockIdx.x, bh = blockIdx.y, b = bh / p.H, h = bh % p.H, hoff = h * p.hd;
  const int key0 = split * FB_KEYS + wave * (FB_KT * 16);         // first key of this wave
  const float c = p.scale * LOG2E;
  const long qbase = (long)b * p.Nq, kbase = (long)b * p.Nk;
  const float keepf = DROP ? p.keep : 1.f;

  bf16x8 kf[FB_KT];          // c * K, hi (k-slots 0-15) / lo (16-31): B operand of the score MFMA
  s16x4 vf[FB_KT], kq[FB_KT];  // V^T (B operand of dP), K^T (A operand of dQ^T)
  f32x4 dkT[FB_KT], dvT[FB_KT];
  uint32_t cwl[FB_KT];
#pragma unroll
  for (int kt = 0; kt < FB_KT; ++kt) {
    const int key = key0 + kt * 16 + li;
    bf16x8 unused;
    load_scaled_frag<DT>(p.k, p.ldk, kbase + key, true, hoff, p.hd, g, c, kf[kt], unused);
    vf[kt] = *reinterpret_cast<const s16x4*>(p.v + (kbase + key) * p.ldv + hoff + 4 * g);
    bf16x4 kc;
#pragma unroll
    for (int j = 0; j < 4; ++j) kc[j] = p.k[(kbase + key0 + kt * 16 + 4 * g + j) * p.ldk + hoff + li];
    kq[kt] = __builtin_bit_cast(s16x4, kc);
    dkT[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    dvT[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    cwl[kt] = 0u;
    if constexpr (DROP) {
      const uint32_t cw = dm_attn_colword(seed_eff, (uint32_t)key >> 1) & M15;
      cwl[kt] = (key & 1) ? (cw >> 16) : cw;
    }
  }
  const int rpar = (li & 1) * ROFF;   // key parity = lane parity (tiles start at multiples of 16)
  const int nqb = p.Nq / 64;

  const Stager<DT> stg = make_stager<DT>(p.q, p.ldq, p.d_o, p.lddo, qbase, p.Nq, hoff, p.hd, tid & 255, wave & 3);
  const bool stager = wave < 4;       // waves 0-3 stage the Q / dO blocks (one 16-byte piece per thread)
  struct Regs { Stage<DT> st; float l, d; uint32_t rs; bool ok; };
  auto fetch = [&](int qb, Regs& r) {
    if (stager) stage_load<DT>(r.st, stg, qb);
    if (wave == 0) {
      const int q = qb * 64 + tid;
      const bool ok = q < p.Nq;
      const long li_ = (long)bh * p.Nq + (ok ? q : 0);
      r.l = p.lse[li_]; r.d = p.delta[li_]; r.ok = ok;   // raw values only (see attn_bwd_kv_kernel)
      if (DROP) r.rs = dm_drop_row(seed_eff, (uint32_t)(bh * p.Nq + q)) & M15;
    }
  };
  auto put = [&](int buf, const Regs& r) {
    if (stager) stage_store<DT>(r.st, sQb[buf], sOb[buf], tid);
    if (wave == 0) {
      sLb[buf][tid] = r.ok ? -(r.l * LOG2E) : -1e30f;
      sDb[buf][tid] = r.ok ? -r.d * keepf : 0.f;
      if (DROP) { sRb[buf][tid] = r.rs; sRb[buf][ROFF + tid] = r.rs >> 16; }
    }
  };
  Regs r0{}, r1{};
  fetch(0, r0);
  if (nqb > 1) fetch(1, r1); else r1 = r0;
  put(0, r0);
  __syncthreads();
  char* sTw = sT[wave];
  const int qq = li >> 2, pp = li & 3;
  float* part = dq_part + (((long)split * p.B * p.H + bh) * p.Nq) * 16;
  auto block = [&](int qb, Regs& r_free, const Regs& r_next) {
    fetch(qb + 2, r_free);
    const char* sQ = sQb[qb & 1];
    const char* sO = sOb[qb & 1];
    const float* sL = sLb[qb & 1];
    const float* sD = sDb[qb & 1];
#pragma unroll 1
    for (int qt = 0; qt < 4; ++qt) {
      const bf16x8 qa = *reinterpret_cast<const bf16x8*>(sQ + (qt * 16 + li) * RS + gk * 16);
      const s16x4 da = *reinterpret_cast<const s16x4*>(sO + (qt * 16 + li) * RS + g * 8);
      const f32x4 L4 = *reinterpret_cast<const f32x4*>(sL + qt * 16 + 4 * g);
      const f32x4 D4 = *reinterpret_cast<const f32x4*>(sD + qt * 16 + 4 * g);
      u32x4 R4 = {0u, 0u, 0u, 0u};
      if constexpr (DROP) R4 = *reinterpret_cast<const u32x4*>(sRb[qb & 1] + rpar + qt * 16 + 4 * g);
      const int toff = (qt * 16 + 4 * g + qq) * RS + pp * 8;
      const s16x4 doT = tr16(sO + toff);   // [d x 16 queries]: A operand of dV^T
      const s16x4 qT = tr16(sQ + toff);    //                   A operand of dK^T
      f32x4 dq4 = {0.f, 0.f, 0.f, 0.f};
      // software pipeline over the key tiles: the score / dP MFMAs of tile kt + 1 are issued before the VALU work of
      // tile kt (their results are not waited for), and the transposed dS of tile kt is consumed one tile later
      // (two alternating LDS tiles), so neither the MFMA nor the LDS round trip is exposed
      f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[0], L4, 0, 0, 0);       // c*s - LSE
      f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(da, vf[0], D4, 0, 0, 0);    // dP - delta*keep
      s16x4 dsT = {0, 0, 0, 0};
#pragma unroll
      for (int kt = 0; kt < FB_KT; ++kt) {
        f32x4 sn = s, dpn = dp;
        if (kt + 1 < FB_KT) {
          sn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[kt + 1], L4, 0, 0, 0);
          dpn = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(da, vf[kt + 1], D4, 0, 0, 0);
        }
        float u[4], w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(s[r]);
          float uu = pv, t = dp[r];
          if constexpr (DROP) {
            const bool keep = keep15(R4[r] ^ cwl[kt], 0, p.t15);
            uu = keep ? pv : 0.f;
            t = keep ? t : D4[r];
          }
          u[r] = uu;
          w[r] = pv * t;
        }
        const s16x4 pB = pack4(u), dsB = pack4(w);
        if (kt > 0) dq4 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kq[kt - 1], dsT, dq4, 0, 0, 0);
        dvT[kt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(doT, pB, dvT[kt], 0, 0, 0);
        dkT[kt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qT, dsB, dkT[kt], 0, 0, 0);
        // dS tile [query 4g+r][key li] -> LDS as T[key][query]; the transposing read returns, for query li, keys 4g..4g+3
        char* tt = sTw + (kt & 1) * 512;
        *reinterpret_cast<s16x4*>(tt + li * 32 + g * 8) = dsB;
        dsT = tr16(tt + (4 * g + qq) * 32 + pp * 8);
        s = sn; dp = dpn;
      }
      dq4 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kq[FB_KT - 1], dsT, dq4, 0, 0, 0);
      *reinterpret_cast<f32x4*>(&sDQ[wave][(qt * 16 + li) * 16 + 4 * g]) = dq4;   // dQ^T: rows d = 4g.., column = query li
    }
    if (qb + 1 < nqb) put((qb + 1) & 1, r_next);
    __syncthreads();
    // the waves' partials, in wave order -> this workgroup's fp32 partial of dQ (64 queries x 16 dims, coalesced)
    for (int idx = tid; idx < 1024; idx += NW * 64) {
      float a = sDQ[0][idx];
#pragma unroll
      for (int wv = 1; wv < NW; ++wv) a += sDQ[wv][idx];
      part[(long)qb * 1024 + idx] = a;
    }
    __syncthreads();
  };
  {
    int qb = 0;
    for (; qb + 1 < nqb; qb += 2) {
      block(qb, r0, r1);
      block(qb + 1, r1, r0);
    }
    if (qb < nqb) block(qb, r0, r1);
  }
  const float ik = DROP ? p.inv_keep : 1.f;
#pragma unroll
  for (int kt = 0; kt < FB_KT; ++kt) {
    const int key = key0 + kt * 16 + li;
    bf16x4 ok, ov;
#pragma unroll
    for (int r = 0; r < 4; ++r) { ok[r] = (bf16)(dkT[kt][r] * (p.scale * ik)); ov[r] = (bf16)(dvT[kt][r] * ik); }
    *reinterpret_cast<bf16x4*>(p.dk + (kbase + key) * p.lddk + hoff + 4 * g) = ok;
    *reinterpret_cast<bf16x4*>(p.dv + (kbase + key) * p.lddv + hoff + 4 * g) = ov;
  }
}

// dq = scale / keep * (sum of the key splits' partials, in split order); thread = (batch*head, query, 4 dims)
__global__ __launch_bounds__(256) void attn_dq_merge_kernel(const AttnParams p, const float* __restrict__ part, float osc,
                                                            int nsplit) {
  const long n4 = (long)p.B * p.H * p.Nq * 4, hs = (long)p.B * p.H * p.Nq * 16;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 a = *reinterpret_cast<const f32x4*>(part + i * 4);
    for (int sp = 1; sp < nsplit; ++sp) a += *reinterpret_cast<const f32x4*>(part + sp * hs + i * 4);
    const f32x4 b2 = {0.f, 0.f, 0.f, 0.f};
    const int d4 = (int)(i & 3);
    const long rq = i >> 2;
    const int q = (int)(rq % p.Nq);
    const int bh = (int)(rq / p.Nq);
    const int b = bh / p.H, h = bh % p.H;
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (bf16)((a[r] + b2[r]) * osc);
    *reinterpret_cast<bf16x4*>(p.dq + ((long)b * p.Nq + q) * p.lddq + h * 16 + d4 * 4) = o;
  }
}

// ---------------------------------------------------------------- few-token cross-attention
// Nk <= 16 context tokens (the CLIP table gives exactly ONE): a 64-key flash tile would be >= 75 %
// masked, so these kernels drop MFMA altogether.  HBM-bound: q / dO are read, o / dq written, once.
//   phase 1, thread = (query, head): scores against the Nk keys (K, V of the sample live in LDS as
//            fp32), softmax, dropout mask -> per-(query, head, key) scalars in LDS;
//   phase 2, thread = channel (coalesced rows): o = sum_k p_k V_k  (fwd) /
//            dq = sum_k ds_k K_k, and dK_k, dV_k accumulated in registers over the block's queries,
//            then ONE fp32 atomic per (key, channel) per block into a [B,Nk,C] scratch.
constexpr int SMALL_NK = 16;
constexpr int SMALL_CHUNKS = 8;  // query chunks (of 256/H queries) per block

template <int HD>
__global__ __launch_bounds__(256) void attn_small_fwd_kernel(const AttnParams p) {
  [[maybe_unused]] const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int H = p.H, C = H * HD, Nk = p.Nk, QB = 256 / H;
  float* sK = sm; float* sV = sK + Nk * C; float* sP = sV + Nk * C;
  const int tid = threadIdx.x, b = blockIdx.y;
  for (int i = tid; i < Nk * C; i += 256) {
    const int k = i / C, c = i % C;
    sK[i] = (float)p.k[((long)b * Nk + k) * p.ldk + c];
    sV[i] = (float)p.v[((long)b * Nk + k) * p.ldv + c];
  }
  __syncthreads();
  const int ql = tid / H, h = tid % H;
  for (int ch = 0; ch < SMALL_CHUNKS; ++ch) {
    const int q0 = (blockIdx.x * SMALL_CHUNKS + ch) * QB;
    if (q0 >= p.Nq) break;
    const int q = q0 + ql;
    if (q < p.Nq && ql < QB) {
      float qv[HD];
      const bf16* qp = p.q + ((long)b * p.Nq + q) * p.ldq + h * HD;
#pragma unroll
      for (int j = 0; j < HD; j += 8) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(qp + j);
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[j + e] = (float)t[e];
      }
      float s[SMALL_NK];
      float m = NEG;
#pragma unroll
      for (int k = 0; k < SMALL_NK; ++k) {
        s[k] = NEG;
        if (k < Nk) {
          float a = 0.f;
#pragma unroll
          for (int d = 0; d < HD; ++d) a += qv[d] * sK[k * C + h * HD + d];
          s[k] = a * p.scale;
          m = fmaxf(m, s[k]);
        }
      }
      float l = 0.f;
#pragma unroll
      for (int k = 0; k < SMALL_NK; ++k)
        if (k < Nk) { s[k] = __expf(s[k] - m); l += s[k]; }
      const float inv = 1.f / l;
      const int bh = b * H + h;
      if (p.lse) p.lse[(long)bh * p.Nq + q] = m + __logf(l);
      const uint32_t rs = p.thresh ? dm_drop_row(seed_eff, (uint32_t)(bh * p.Nq + q)) : 0u;
#pragma unroll
      for (int k = 0; k < SMALL_NK; ++k)
        if (k < Nk) {
          float pv = s[k] * inv;
          if (p.thresh) pv = dm_keep16(dm_drop_bits(rs, (uint32_t)k >> 1), (uint32_t)k, p.thresh) ? pv * p.inv_keep : 0.f;
          sP[(ql * H + h) * Nk + k] = pv;
        }
    }
    __syncthreads();
    for (int i = tid; i < QB * C; i += 256) {
      const int qq = i / C, c = i % C, hh = c / HD;
      if (q0 + qq < p.Nq) {
        float o = 0.f;
        for (int k = 0; k < Nk; ++k) o += sP[(qq * H + hh) * Nk + k] * sV[k * C + c];
        p.o[((long)b * p.Nq + q0 + qq) * p.ldo + c] = (bf16)o;
      }
    }
    __syncthreads();
  }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const AttnParams p, float* __restrict__ ws) {
  [[maybe_unused]] const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int H = p.H, C = H * HD, Nk = p.Nk, QB = 256 / H;
  float* sK = sm; float* sV = sK + Nk * C; float* sDS = sV + Nk * C; float* sPD = sDS + QB * H * Nk;
  const int tid = threadIdx.x, b = blockIdx.y;
  for (int i = tid; i < Nk * C; i += 256) {
    const int k = i / C, c = i % C;
    sK[i] = (float)p.k[((long)b * Nk + k) * p.ldk + c];
    sV[i] = (float)p.v[((long)b * Nk + k) * p.ldv + c];
  }
  __syncthreads();
  const int ql = tid / H, h = tid % H;
  const int P = 256 / C;                 // phase-2 thread groups (C <= 256)
  const int c2 = tid % C, part = tid / C, h2 = c2 / HD;
  float accK[SMALL_NK], accV[SMALL_NK];
#pragma unroll
  for (int k = 0; k < SMALL_NK; ++k) { accK[k] = 0.f; accV[k] = 0.f; }
  for (int ch = 0; ch < SMALL_CHUNKS; ++ch) {
    const int q0 = (blockIdx.x * SMALL_CHUNKS + ch) * QB;
    if (q0 >= p.Nq) break;
    const int q = q0 + ql;
    if (q < p.Nq && ql < QB) {
      float qv[HD], dv_[HD];
      const bf16* qp = p.q + ((long)b * p.Nq + q) * p.ldq + h * HD;
      const bf16* dp = p.d_o + ((long)b * p.Nq + q) * p.lddo + h * HD;
#pragma unroll
      for (int j = 0; j < HD; j += 8) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(qp + j);
        const bf16x8 u = *reinterpret_cast<const bf16x8*>(dp + j);
#pragma unroll
        for (int e = 0; e < 8; ++e) { qv[j + e] = (float)t[e]; dv_[j + e] = (float)u[e]; }
      }
      const int bh = b * H + h;
      const float L = p.lse[(long)bh * p.Nq + q];
      const uint32_t rs = p.thresh ? dm_drop_row(seed_eff, (uint32_t)(bh * p.Nq + q)) : 0u;
      float pk[SMALL_NK], dP[SMALL_NK];
      float delta = 0.f;
#pragma unroll
      for (int k = 0; k < SMALL_NK; ++k) {
        pk[k] = 0.f; dP[k] = 0.f;
        if (k < Nk) {
          float a = 0.f, d2 = 0.f;
#pragma unroll
          for (int d = 0; d < HD; ++d) {
            a += qv[d] * sK[k * C + h * HD + d];
            d2 += dv_[d] * sV[k * C + h * HD + d];
          }
          pk[k] = __expf(a * p.scale - L);
          float keepf = 1.f;
          if (p.thresh) keepf = dm_keep16(dm_drop_bits(rs, (uint32_t)k >> 1), (uint32_t)k, p.thresh) ? p.inv_keep : 0.f;
          dP[k] = d2 * keepf;
          delta += pk[k] * dP[k];
          sPD[(ql * H + h) * Nk + k] = pk[k] * keepf;
        }
      }
#pragma unroll
      for (int k = 0; k < SMALL_NK; ++k)
        if (k < Nk) sDS[(ql * H + h) * Nk + k] = pk[k] * (dP[k] - delta) * p.scale;
    }
    __syncthreads();
    if (part < P) {
      for (int qq = part; qq < QB; qq += P) {
        const int q2 = q0 + qq;
        if (q2 >= p.Nq) break;
        const float qc = (float)p.q[((long)b * p.Nq + q2) * p.ldq + c2];
        const float dc = (float)p.d_o[((long)b * p.Nq + q2) * p.lddo + c2];
        float dq = 0.f;
#pragma unroll
        for (int k = 0; k < SMALL_NK; ++k)
          if (k < Nk) {
            const float ds = sDS[(qq * H + h2) * Nk + k];
            dq += ds * sK[k * C + c2];
            accK[k] += ds * qc;
            accV[k] += sPD[(qq * H + h2) * Nk + k] * dc;
          }
        p.dq[((long)b * p.Nq + q2) * p.lddq + c2] = (bf16)dq;
      }
    }
    __syncthreads();
  }
  if (part < P) {
#pragma unroll
    for (int k = 0; k < SMALL_NK; ++k)
      if (k < Nk) {
        atomicAdd(ws + ((long)b * Nk + k) * C + c2, accK[k]);
        atomicAdd(ws + (long)p.B * Nk * C + ((long)b * Nk + k) * C + c2, accV[k]);
      }
  }
}

__global__ void attn_small_finish_kernel(const AttnParams p, const float* __restrict__ ws, int C) {
  [[maybe_unused]] const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  const long n = (long)p.B * p.Nk * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long bk = i / C;
    if (p.dk) p.dk[bk * p.lddk + c] = (bf16)ws[i];
    p.dv[bk * p.lddv + c] = (bf16)ws[n + i];
  }
}


// ---------------------------------------------------------------- wide heads (hd = 64, 128)
// ---------------------------------------------------------------- wide heads (64 / 128) on the MFMA pipe
// The latent-diffusion model's self-attention (utils.py:145-156: 512 and 1024 channels over 8 heads, 256 and 64 tokens
// at 32x32 latents).  With head dims this wide a score costs 4*hd = 256..512 MFMA flop, so these kernels are MFMA /
// LDS-read bound, not VALU bound, and are kept plain: same register-resident P trick as the flash kernels (score tile
// computed transposed, keys on MFMA rows, so a lane already owns the 8 keys per 32-key step that the second MFMA
// wants), but the raw q.k accumulator is scaled on the VALU (no hi/lo split), the softmax rescales in every key block,
// and the keys / values of a (batch, head) are staged KCH rows at a time into one LDS image (no double buffering:
// at <= 256 keys a workgroup stages once or twice).  Row stride: hd*2 bytes + 32 (hd 64: 160) or + 96 (hd 128: 352) --
// 10 resp. 22 sixteen-byte units, i.e. +-6 mod 16, which keeps both the ds_read_b128 operand reads and the
// ds_read_b64_tr_b16 halves bank-conflict free (same argument as RS above).
// Dropout words: the few-token convention (row word of the query, one 32-bit draw per key pair).
template <int HD> struct Wide {
  static constexpr int RSW = HD == 64 ? 160 : 352;   // LDS row stride, bytes
  static constexpr int KC = HD / 32;                 // 32-wide reduction chunks of a q.k product
  static constexpr int DT = HD / 16;                 // 16-row tiles of a head-dim-major output
  static constexpr int KCH = HD == 64 ? 128 : 64;    // rows staged per LDS image
};

// rows [r0, r0 + nrows) of one (batch, head) slice -> LDS rows 0..nrows-1; rows past n_valid are zero-filled
template <int HD>
__device__ __forceinline__ void wide_stage_rows(char* dst, const bf16* src, long ld, long rowbase, int r0, int nrows,
                                                int n_valid, int hoff, int tid) {
  constexpr int PPR = HD / 8, RSW = Wide<HD>::RSW;
  for (int i = tid; i < nrows * PPR; i += 256) {
    const int r = i / PPR, pc = i % PPR;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r0 + r < n_valid) v = *reinterpret_cast<const u32x4*>(src + (rowbase + r0 + r) * ld + hoff + pc * 8);
    *reinterpret_cast<u32x4*>(dst + r * RSW + pc * 16) = v;
  }
}
// fragment of row `tok` straight from global: head-dim elements [kc*32 + 8g, +8) for every reduction chunk kc
template <int HD>
__device__ __forceinline__ void wide_row_frags(const bf16* base, long ld, long tok, bool ok, int hoff, int g,
                                               bf16x8 (&f)[HD / 32]) {
#pragma unroll
  for (int kc = 0; kc < HD / 32; ++kc)
    f[kc] = ok ? *reinterpret_cast<const bf16x8*>(base + tok * ld + hoff + kc * 32 + 8 * g) : zero8();
}
__device__ __forceinline__ float wide_zf(uint32_t rs, int key, uint32_t thresh, float inv_keep) {
  return dm_keep16(dm_drop_bits(rs, (uint32_t)key >> 1), (uint32_t)key, thresh) ? inv_keep : 0.f;
}

// forward: a wave owns QT tiles of 16 queries (queries on the lane axis), the workgroup 64*QT queries
template <int HD, int QT>
__global__ __launch_bounds__(256, 2) void attn_wide_fwd_kernel(const AttnParams p) {
  constexpr int RSW = Wide<HD>::RSW, KC = Wide<HD>::KC, DT = Wide<HD>::DT, KCH = Wide<HD>::KCH;
  const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  __shared__ __attribute__((aligned(16))) char sK[KCH * RSW];
  __shared__ __attribute__((aligned(16))) char sV[KCH * RSW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int qq = li >> 2, pp = li & 3;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H, hoff = h * HD;
  const long qbase = (long)b * p.Nq, kbase = (long)b * p.Nk;
  const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);
  const float c = p.scale * LOG2E;
  bf16x8 qf[QT][KC];
  int qi[QT];
  uint32_t rs[QT];
  float m[QT], l[QT];
  f32x4 acc[DT][QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qi[qt] = q0 + qt * 16 + li;
    wide_row_frags<HD>(p.q, p.ldq, qbase + qi[qt], qi[qt] < p.Nq, hoff, g, qf[qt]);
    rs[qt] = dm_drop_row(seed_eff, (uint32_t)(bh * p.Nq + qi[qt]));
    m[qt] = NEG;
    l[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int k0 = 0; k0 < p.Nk; k0 += KCH) {
    const int nrows = min(KCH, (p.Nk - k0 + 63) & ~63);
    __syncthreads();
    wide_stage_rows<HD>(sK, p.k, p.ldk, kbase, k0, nrows, p.Nk, hoff, tid);
    wide_stage_rows<HD>(sV, p.v, p.ldv, kbase, k0, nrows, p.Nk, hoff, tid);
    __syncthreads();
    for (int kb = 0; kb < nrows; kb += 64) {
      f32x4 st[4][QT];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) st[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (kb + kt * 16 + li) * RSW + kc * 64 + g * 16);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt)
            st[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][kc], st[kt][qt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float t = NEG;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = k0 + kb + kt * 16 + 4 * g + r;
            const float s = key < p.Nk ? st[kt][qt][r] * c : NEG;
            st[kt][qt][r] = s;
            t = fmaxf(t, s);
          }
        t = fmaxf(t, __shfl_xor(t, 16, 64));
        t = fmaxf(t, __shfl_xor(t, 32, 64));
        const float mn = fmaxf(m[qt], t);
        const float alpha = __builtin_amdgcn_exp2f(m[qt] - mn);   // first block: m = NEG -> 0 (and l = acc = 0)
        m[qt] = mn;
        l[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt][qt] *= alpha;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = k0 + kb + kt * 16 + 4 * g + r;
            // the row sum takes the bf16-rounded, undropped probability: what the second MFMA multiplies
            float pj = (float)(bf16)__builtin_amdgcn_exp2f(st[kt][qt][r] - mn);
            if (key >= p.Nk) pj = 0.f;
            l[qt] += pj;
            if (p.thresh && wide_zf(rs[qt], key, p.thresh, 1.f) == 0.f) pj = 0.f;
            st[kt][qt][r] = pj;
          }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 pb[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) pb[qt] = pack8(st[2 * ks][qt], st[2 * ks + 1][qt]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const char* a0 = sV + (kb + ks * 32 + 4 * g + qq) * RSW + dt * 32 + pp * 8;
          const bf16x8 vT = tr_pair(a0, a0 + 16 * RSW);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt)
            acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vT, pb[qt], acc[dt][qt], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float lt = l[qt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    if (qi[qt] < p.Nq) {
      const float inv = (p.thresh ? p.inv_keep : 1.f) / lt;
      if (g == 0 && p.lse) p.lse[(long)bh * p.Nq + qi[qt]] = (m[qt] + log2f(lt)) * LN2;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(acc[dt][qt][r] * inv);
        *reinterpret_cast<bf16x4*>(p.o + (qbase + qi[qt]) * p.ldo + hoff + dt * 16 + 4 * g) = o;
      }
    }
  }
}

// dQ: the forward's mapping; P is rebuilt from the saved log-sum-exp, dP^T = V dO^T comes from a second MFMA against
// the staged values, and dQ^T = K^T dS^T reads the staged keys through the transposing LDS read.
template <int HD, int QT>
__global__ __launch_bounds__(256, 2) void attn_wide_bwd_q_kernel(const AttnParams p) {
  constexpr int RSW = Wide<HD>::RSW, KC = Wide<HD>::KC, DT = Wide<HD>::DT, KCH = Wide<HD>::KCH;
  const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  __shared__ __attribute__((aligned(16))) char sK[KCH * RSW];
  __shared__ __attribute__((aligned(16))) char sV[KCH * RSW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int qq = li >> 2, pp = li & 3;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H, hoff = h * HD;
  const long qbase = (long)b * p.Nq, kbase = (long)b * p.Nk;
  const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);
  const float c = p.scale * LOG2E;
  bf16x8 qf[QT][KC], df[QT][KC];
  int qi[QT];
  uint32_t rs[QT];
  float L2[QT], dl[QT];
  f32x4 acc[DT][QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qi[qt] = q0 + qt * 16 + li;
    const bool ok = qi[qt] < p.Nq;
    wide_row_frags<HD>(p.q, p.ldq, qbase + qi[qt], ok, hoff, g, qf[qt]);
    wide_row_frags<HD>(p.d_o, p.lddo, qbase + qi[qt], ok, hoff, g, df[qt]);
    rs[qt] = dm_drop_row(seed_eff, (uint32_t)(bh * p.Nq + qi[qt]));
    L2[qt] = ok ? p.lse[(long)bh * p.Nq + qi[qt]] * LOG2E : 0.f;
    dl[qt] = ok ? p.delta[(long)bh * p.Nq + qi[qt]] : 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int k0 = 0; k0 < p.Nk; k0 += KCH) {
    const int nrows = min(KCH, (p.Nk - k0 + 63) & ~63);
    __syncthreads();
    wide_stage_rows<HD>(sK, p.k, p.ldk, kbase, k0, nrows, p.Nk, hoff, tid);
    wide_stage_rows<HD>(sV, p.v, p.ldv, kbase, k0, nrows, p.Nk, hoff, tid);
    __syncthreads();
    for (int kb = 0; kb < nrows; kb += 64) {
      f32x4 st[4][QT], dp[4][QT];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) { st[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kt][qt] = st[kt][qt]; }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const int off = (kb + kt * 16 + li) * RSW + kc * 64 + g * 16;
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + off);
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + off);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            st[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][kc], st[kt][qt], 0, 0, 0);
            dp[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, df[qt][kc], dp[kt][qt], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = k0 + kb + kt * 16 + 4 * g + r;
            const float pj = key < p.Nk ? __builtin_amdgcn_exp2f(st[kt][qt][r] * c - L2[qt]) : 0.f;
            const float zf = p.thresh ? wide_zf(rs[qt], key, p.thresh, p.inv_keep) : 1.f;
            st[kt][qt][r] = pj * (zf * dp[kt][qt][r] - dl[qt]);
          }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 sb[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) sb[qt] = pack8(st[2 * ks][qt], st[2 * ks + 1][qt]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const char* a0 = sK + (kb + ks * 32 + 4 * g + qq) * RSW + dt * 32 + pp * 8;
          const bf16x8 kT = tr_pair(a0, a0 + 16 * RSW);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt)
            acc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT, sb[qt], acc[dt][qt], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
    if (qi[qt] < p.Nq) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(acc[dt][qt][r] * p.scale);
        *reinterpret_cast<bf16x4*>(p.dq + (qbase + qi[qt]) * p.lddq + hoff + dt * 16 + 4 * g) = o;
      }
    }
}

// dK, dV: keys on the lane axis (a wave owns KT tiles of 16 keys, their k / v fragments resident in registers), the
// queries and their dO rows staged 64 at a time; S = Q K^T and dP = dO V^T read the staged rows as MFMA A operands,
// dV^T = dO^T P and dK^T = Q^T dS read them again through the transposing LDS read.
template <int HD, int KT>
__global__ __launch_bounds__(256, 2) void attn_wide_bwd_kv_kernel(const AttnParams p) {
  constexpr int RSW = Wide<HD>::RSW, KC = Wide<HD>::KC, DT = Wide<HD>::DT;
  const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  __shared__ __attribute__((aligned(16))) char sQ[64 * RSW];
  __shared__ __attribute__((aligned(16))) char sD[64 * RSW];
  __shared__ float sL[64], sDl[64];
  __shared__ uint32_t sRS[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int qq = li >> 2, pp = li & 3;
  const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H, hoff = h * HD;
  const long qbase = (long)b * p.Nq, kbase = (long)b * p.Nk;
  const int kk0 = blockIdx.x * (64 * KT) + wave * (16 * KT);
  const float c = p.scale * LOG2E;
  bf16x8 kf[KT][KC], vf[KT][KC];
  int key[KT];
  f32x4 dk[DT][KT], dv[DT][KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    key[kt] = kk0 + kt * 16 + li;
    wide_row_frags<HD>(p.k, p.ldk, kbase + key[kt], key[kt] < p.Nk, hoff, g, kf[kt]);
    wide_row_frags<HD>(p.v, p.ldv, kbase + key[kt], key[kt] < p.Nk, hoff, g, vf[kt]);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dk[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt][kt] = dk[dt][kt]; }
  }
  for (int q0 = 0; q0 < p.Nq; q0 += 64) {
    __syncthreads();
    wide_stage_rows<HD>(sQ, p.q, p.ldq, qbase, q0, 64, p.Nq, hoff, tid);
    wide_stage_rows<HD>(sD, p.d_o, p.lddo, qbase, q0, 64, p.Nq, hoff, tid);
    if (tid < 64) {
      const bool ok = q0 + tid < p.Nq;   // rows past the end: q = dO = 0 and delta = 0, so dS = P * (0 - 0) = 0 and P * dO = 0
      sL[tid] = ok ? p.lse[(long)bh * p.Nq + q0 + tid] * LOG2E : 0.f;
      sDl[tid] = ok ? p.delta[(long)bh * p.Nq + q0 + tid] : 0.f;
      sRS[tid] = dm_drop_row(seed_eff, (uint32_t)(bh * p.Nq + q0 + tid));
    }
    __syncthreads();
    f32x4 st[4][KT], dp[4][KT];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) { st[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qt][kt] = st[qt][kt]; }
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const int off = (qt * 16 + li) * RSW + kc * 64 + g * 16;
        const bf16x8 qa = *reinterpret_cast<const bf16x8*>(sQ + off);
        const bf16x8 da = *reinterpret_cast<const bf16x8*>(sD + off);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          st[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[kt][kc], st[qt][kt], 0, 0, 0);
          dp[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[kt][kc], dp[qt][kt], 0, 0, 0);
        }
      }
    }
    // st -> dS, dp -> dropped P (both in place)
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = qt * 16 + 4 * g + r;
        const float L = sL[ql], Dl = sDl[ql];
        const uint32_t rsq = sRS[ql];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const float pj = key[kt] < p.Nk ? __builtin_amdgcn_exp2f(st[qt][kt][r] * c - L) : 0.f;
          const float zf = p.thresh ? wide_zf(rsq, key[kt], p.thresh, p.inv_keep) : 1.f;
          st[qt][kt][r] = pj * (zf * dp[qt][kt][r] - Dl);
          dp[qt][kt][r] = pj * zf;
        }
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 sb[KT], pb[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        sb[kt] = pack8(st[2 * ks][kt], st[2 * ks + 1][kt]);
        pb[kt] = pack8(dp[2 * ks][kt], dp[2 * ks + 1][kt]);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int off = (ks * 32 + 4 * g + qq) * RSW + dt * 32 + pp * 8;
        const bf16x8 qT = tr_pair(sQ + off, sQ + off + 16 * RSW);
        const bf16x8 dT = tr_pair(sD + off, sD + off + 16 * RSW);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          dk[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT, sb[kt], dk[dt][kt], 0, 0, 0);
          dv[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dT, pb[kt], dv[dt][kt], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
    if (key[kt] < p.Nk) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        bf16x4 ok_, ov_;
#pragma unroll
        for (int r = 0; r < 4; ++r) { ok_[r] = (bf16)(dk[dt][kt][r] * p.scale); ov_[r] = (bf16)dv[dt][kt][r]; }
        *reinterpret_cast<bf16x4*>(p.dk + (kbase + key[kt]) * p.lddk + hoff + dt * 16 + 4 * g) = ok_;
        *reinterpret_cast<bf16x4*>(p.dv + (kbase + key[kt]) * p.lddv + hoff + dt * 16 + 4 * g) = ov_;
      }
    }
}

// ---------------------------------------------------------------- one context token (the CLIP table)
// With a single key the softmax is identically 1 -- exactly, in any arithmetic: exp(s - s) / 1 -- so
//   o[q] = Z[q,h] * v / keep,   lse[q] = scale * q.k,   dS = P o (dP - delta) = 0  =>  dq = dk = 0,
//   dv = sum_q Z[q,h] * dO[q] / keep
// (Z: dropout mask of the one probability per (query, head), same words as the few-token kernels).  These are
// pure streaming kernels: thread = (query, 8 channels), one 16-byte load and one 16-byte store per element
// group, no LDS in the loop.
__device__ __forceinline__ bool onekey_keep(const AttnParams& p, uint32_t seed_eff, int bh, int q) {
  const uint32_t rs = dm_drop_row(seed_eff, (uint32_t)(bh * p.Nq + q));
  return dm_keep16(dm_drop_bits(rs, 0u), 0u, p.thresh);
}
__global__ __launch_bounds__(256) void attn_onekey_fwd_kernel(const AttnParams p) {
  [[maybe_unused]] const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  const int C = p.H * p.hd, TPQ = C >> 3, QPB = 256 / TPQ;
  const int tid = threadIdx.x, b = blockIdx.y, c8 = (tid % TPQ) * 8, qs = tid / TPQ, h = c8 / p.hd;
  const int sub = p.hd >> 3;  // threads per head: 1, 2, 4, 8 or 16
  const bf16x8 vr = *reinterpret_cast<const bf16x8*>(p.v + (long)b * p.ldv + c8);
  float kf[8];
  bf16x8 vs, vz;
#pragma unroll
  for (int j = 0; j < 8; ++j) { kf[j] = 0.f; vs[j] = (bf16)((float)vr[j] * p.inv_keep); vz[j] = (bf16)0.f; }
  if (p.lse) {  // q and k are only read for the log-sum-exp output (optional with one key: dm_hip.h)
    const bf16x8 kr = *reinterpret_cast<const bf16x8*>(p.k + (long)b * p.ldk + c8);
#pragma unroll
    for (int j = 0; j < 8; ++j) kf[j] = (float)kr[j] * p.scale;
  }
  const bool drop = p.thresh != 0u;
  const int bh = b * p.H + h;
  for (int q = blockIdx.x * QPB + qs; q < p.Nq; q += gridDim.x * QPB) {
    const long row = (long)b * p.Nq + q;
    if (p.lse) {
      const bf16x8 qr = *reinterpret_cast<const bf16x8*>(p.q + row * p.ldq + c8);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)qr[j] * kf[j];
      for (int o2 = 1; o2 < sub; o2 <<= 1) s += __shfl_xor(s, o2, 64);
      if ((tid % sub) == 0) p.lse[(long)bh * p.Nq + q] = s;
    }
    const bool keep = !drop || onekey_keep(p, seed_eff, bh, q);
    if (p.o) *reinterpret_cast<bf16x8*>(p.o + row * p.ldo + c8) = keep ? (drop ? vs : vr) : vz;
  }
}
// ws: [gridDim.x][B][C] fp32 per-block partial sums of dV (plain stores: no memset before, no atomics, a fixed
// summation order), summed by attn_onekey_finish_kernel
__global__ __launch_bounds__(256) void attn_onekey_bwd_kernel(const AttnParams p, float* __restrict__ ws) {
  [[maybe_unused]] const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  __shared__ float sAcc[256 * 8];
  const int C = p.H * p.hd, TPQ = C >> 3, QPB = 256 / TPQ;
  const int tid = threadIdx.x, b = blockIdx.y, c8 = (tid % TPQ) * 8, qs = tid / TPQ, h = c8 / p.hd;
  const bool drop = p.thresh != 0u;
  const int bh = b * p.H + h;
  float acc[8];
  bf16x8 vz;
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[j] = 0.f; vz[j] = (bf16)0.f; }
  for (int q = blockIdx.x * QPB + qs; q < p.Nq; q += gridDim.x * QPB) {
    const long row = (long)b * p.Nq + q;
    const bf16x8 d = *reinterpret_cast<const bf16x8*>(p.d_o + row * p.lddo + c8);
    if (!drop || onekey_keep(p, seed_eff, bh, q)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)d[j];
    }
    if (p.dq) *reinterpret_cast<bf16x8*>(p.dq + row * p.lddq + c8) = vz;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sAcc[(qs * TPQ * 8) + (tid % TPQ) * 8 + j] = acc[j];
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    for (int i = 0; i < QPB; ++i) s += sAcc[i * C + c];
    ws[((long)blockIdx.x * p.B + b) * C + c] = s * (drop ? p.inv_keep : 1.f);
  }
}
__global__ void attn_onekey_finish_kernel(const AttnParams p, const float* __restrict__ ws, int C, int nblk) {
  [[maybe_unused]] const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  const long n = (long)p.B * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nblk; ++k) s += ws[k * n + i];
    const int c = (int)(i % C);
    const long b = i / C;
    if (p.dk) p.dk[b * p.lddk + c] = (bf16)0.f;
    p.dv[b * p.lddv + c] = (bf16)s;
  }
}
bool onekey_ok(const AttnParams& p) {
  const int C = p.H * p.hd;
  return p.Nk == 1 && C <= 2048 && 256 % (C >> 3) == 0 && (!p.o || (p.ldo & 7) == 0) && (!p.dq || ((p.lddq & 7) == 0 && (p.lddo & 7) == 0));
}

bool small_ok(const AttnParams& p) {
  const int C = p.H * p.hd;
  return p.Nk <= SMALL_NK && C <= 256 && 256 % p.H == 0 && 256 % C == 0;
}

int fill(AttnParams& p, const dm_attn_args* a, bool bwd) {
  // With ONE key and the log-sum-exp wanted, the forward's output may be declined (o = NULL): a caller that forms the
  // attention output elsewhere (dm_cross1_fwd) then pays for the q.k pass alone.
  DM_REQUIRE(a && a->v && (a->o || (!bwd && a->Nk == 1 && a->lse && a->q && a->k)));
  DM_REQUIRE(a->hd == 8 || a->hd == 16 || a->hd == 32 || a->hd == 64 || a->hd == 128);
  DM_REQUIRE(a->B > 0 && a->heads > 0 && a->Nq > 0 && a->Nk > 0);
  // With ONE key the output does not depend on q or k (softmax over one element), so a caller that has no use for
  // the log-sum-exp may pass q = k = lse = NULL, and in backward dq = dk = NULL (both are identically zero).
  const bool onekey_min = a->Nk == 1 && !a->lse && (!bwd || (!a->dq && !a->dk));
  DM_REQUIRE(onekey_min || (a->q && a->k && a->lse));
  DM_REQUIRE(a->ldq % 8 == 0 && a->ldk % 8 == 0 && a->ldv % 8 == 0 && a->ldo % 4 == 0);
  DM_REQUIRE(a->drop_p >= 0.f && a->drop_p < 1.f);
  if (bwd) {
    DM_REQUIRE(a->d_o && a->dv && a->delta && (onekey_min || (a->dq && a->dk)));
    DM_REQUIRE(a->lddo % 8 == 0 && a->ldo % 8 == 0 && a->lddq % 4 == 0 && a->lddk % 4 == 0 && a->lddv % 4 == 0);
  }
  p.q = (const bf16*)a->q; p.ldq = a->ldq; p.k = (const bf16*)a->k; p.ldk = a->ldk;
  p.v = (const bf16*)a->v; p.ldv = a->ldv; p.o = (bf16*)a->o; p.ldo = a->ldo; p.lse = a->lse;
  p.d_o = (const bf16*)a->d_o; p.lddo = a->lddo; p.dq = (bf16*)a->dq; p.lddq = a->lddq;
  p.dk = (bf16*)a->dk; p.lddk = a->lddk; p.dv = (bf16*)a->dv; p.lddv = a->lddv; p.delta = a->delta;
  p.B = a->B; p.H = a->heads; p.Nq = a->Nq; p.Nk = a->Nk; p.hd = a->hd; p.scale = a->scale;
  p.seed = dm_seed32(a->drop_seed);
  p.seed_ctr = a->drop_p > 0.f ? dm_dropout_counter() : nullptr;
  // attention draws 15-bit fields (the flash kernels compare them with packed 16-bit signed arithmetic)
  uint32_t t15 = a->drop_p > 0.f ? (uint32_t)((double)a->drop_p * 32768.0 + 0.5) : 0u;
  if (t15 > 32767u) t15 = 32767u;
  p.t15 = t15;
  p.tm1x2 = ((t15 - 1u) & 0xFFFFu) * 0x10001u;
  p.thresh = 2u * t15;  // the same probability as a 16-bit threshold (few-token kernels)
  p.keep = (32768.f - (float)t15) / 32768.f;
  p.inv_keep = 32768.f / (32768.f - (float)t15);
  return DM_OK;
}

template <bool DROP, int NW, int KT>
int launch_fused(const AttnParams& p, float* part, hipStream_t st) {
  constexpr int LDS = 4 * 64 * RS + 1024 + 2 * (72 + 64) * 4 + NW * 1024 + NW * 4096;
  auto k = attn_bwd_fused_kernel<DROP, NW, KT>;
  static unsigned long long attr_devs = 0ull;
  if (dm_first_on_device(attr_devs)) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return DM_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(k, dim3(FB_NK / (NW * KT * 16), p.B * p.H), dim3(NW * 64), LDS, st, p, part);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

bool fused_bwd_enabled() {   // DM_ATTN_BWD_FUSED=0: the two-kernel backward, for A/B measurements
  static const bool on = [] { const char* e = getenv("DM_ATTN_BWD_FUSED"); return !(e && e[0] == '0'); }();
  return on;
}

}  // namespace

extern "C" long dm_attn_bwd_workspace_bytes(int B, int heads, int Nq, int Nk, int hd) {
  if (hd == 16 && Nk == FB_NK && Nq % 64 == 0) return (long)4 * B * heads * Nq * 16 * 4;   // four key splits
  return 0;
}

extern "C" int dm_attn_fwd(const dm_attn_args* a, dm_stream_t stream) {
  AttnParams p{};
  const int rc = fill(p, a, false);
  if (rc != DM_OK) return rc;
  if ((!p.q || !p.o) && !onekey_ok(p)) return DM_ERR_ARG;  // the q-less and o-less forms exist only on the one-token path
  if (onekey_ok(p)) {
    const int QPB = 256 / ((p.H * p.hd) >> 3);
    int gx = (p.Nq + QPB - 1) / QPB;
    if (gx > 32) gx = 32;
    dm_set_variant("attn_onekey_fwd");
    hipLaunchKernelGGL(attn_onekey_fwd_kernel, dim3(gx, p.B), dim3(256), 0, (hipStream_t)stream, p);
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  if (small_ok(p)) {
    const int C = p.H * p.hd, QB = 256 / p.H;
    const dim3 g((p.Nq + QB * SMALL_CHUNKS - 1) / (QB * SMALL_CHUNKS), p.B);
    const size_t lds = (size_t)(2 * p.Nk * C + QB * p.H * p.Nk) * sizeof(float);
    dm_set_variant("attn_small_fwd");
    if (p.hd == 8) hipLaunchKernelGGL(attn_small_fwd_kernel<8>, g, dim3(256), lds, (hipStream_t)stream, p);
    else if (p.hd == 16) hipLaunchKernelGGL(attn_small_fwd_kernel<16>, g, dim3(256), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(attn_small_fwd_kernel<32>, g, dim3(256), lds, (hipStream_t)stream, p);
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  if (p.hd >= 64) {  // wide heads (latent-model widths): MFMA kernels, two query tiles per wave when there are enough queries
    const bool two = p.Nq >= 128;
    const dim3 g((p.Nq + (two ? 127 : 63)) / (two ? 128 : 64), p.B * p.H);
    dm_set_variant(two ? "attn_wide_mfma_fwd<QT=2>" : "attn_wide_mfma_fwd<QT=1>");
#define DM_WF(HD, QT) hipLaunchKernelGGL((attn_wide_fwd_kernel<HD, QT>), g, dim3(256), 0, (hipStream_t)stream, p)
    if (p.hd == 64) { if (two) DM_WF(64, 2); else DM_WF(64, 1); }
    else { if (two) DM_WF(128, 2); else DM_WF(128, 1); }
#undef DM_WF
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  const dim3 grid((p.Nq + 127) / 128, p.B * p.H);
  const bool edge = (p.Nk & 63) != 0;
#define DM_FWD(DT, DROP, EDGE) hipLaunchKernelGGL((attn_fwd_kernel<DT, DROP, EDGE>), grid, dim3(256), 0, (hipStream_t)stream, p)
#define DM_FWD2(DT, DROP) do { if (edge) DM_FWD(DT, DROP, true); else DM_FWD(DT, DROP, false); } while (0)
  dm_set_variant(p.hd > 16 ? (p.thresh ? "attn_flash_fwd<DT=2,DROP=1>" : "attn_flash_fwd<DT=2,DROP=0>")
                           : (p.thresh ? "attn_flash_fwd<DT=1,DROP=1>" : "attn_flash_fwd<DT=1,DROP=0>"));
  if (p.hd > 16) { if (p.thresh) DM_FWD2(2, true); else DM_FWD2(2, false); }
  else { if (p.thresh) DM_FWD2(1, true); else DM_FWD2(1, false); }
#undef DM_FWD2
#undef DM_FWD
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_attn_bwd(const dm_attn_args* a, dm_stream_t stream) {
  AttnParams p{};
  const int rc = fill(p, a, true);
  if (rc != DM_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (!p.q && !onekey_ok(p)) return DM_ERR_ARG;  // the q-less form exists only on the one-token path
  if (onekey_ok(p)) {
    const int C = p.H * p.hd, QPB = 256 / (C >> 3);
    float* ws = p.delta;  // fp32 [<=32][B][C] per-block partials (the caller sizes it: dm_hip.h)
    int gx = (p.Nq + QPB - 1) / QPB;
    if (gx > 32) gx = 32;
    dm_set_variant("attn_onekey_bwd");
    hipLaunchKernelGGL(attn_onekey_bwd_kernel, dim3(gx, p.B), dim3(256), 0, st, p, ws);
    DM_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_onekey_finish_kernel, dim3(64), dim3(256), 0, st, p, (const float*)ws, C, gx);
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  if (small_ok(p)) {
    // p.delta doubles as the fp32 [2][B,Nk,C] reduction scratch (the caller sizes it: dm_hip.h)
    const int C = p.H * p.hd, QB = 256 / p.H;
    float* ws = p.delta;
    if (hipMemsetAsync(ws, 0, (size_t)2 * p.B * p.Nk * C * sizeof(float), st) != hipSuccess) return DM_ERR_LAUNCH;
    const dim3 g((p.Nq + QB * SMALL_CHUNKS - 1) / (QB * SMALL_CHUNKS), p.B);
    const size_t lds = (size_t)(2 * p.Nk * C + 2 * QB * p.H * p.Nk) * sizeof(float);
    dm_set_variant("attn_small_bwd");
    if (p.hd == 8) hipLaunchKernelGGL(attn_small_bwd_kernel<8>, g, dim3(256), lds, st, p, ws);
    else if (p.hd == 16) hipLaunchKernelGGL(attn_small_bwd_kernel<16>, g, dim3(256), lds, st, p, ws);
    else hipLaunchKernelGGL(attn_small_bwd_kernel<32>, g, dim3(256), lds, st, p, ws);
    DM_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_small_finish_kernel, dim3(64), dim3(256), 0, st, p, (const float*)ws, C);
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  const long total = (long)p.B * p.H * p.Nq;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  const int Cc = p.H * p.hd;
  if (Cc <= 2048 && 256 % (Cc >> 3) == 0) {
    const int QPB = 256 / (Cc >> 3);
    long nb = ((long)p.B * p.Nq + QPB - 1) / QPB;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(attn_delta_rows_kernel, dim3((unsigned)nb), dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL(attn_delta_kernel, dim3(blocks), dim3(256), 0, st, p);
  }
  DM_LAUNCH_CHECK();
  if (p.hd >= 64) {  // wide heads: MFMA kernels (two key tiles per wave at head dim 64 when there are enough keys)
    const bool twoq = p.hd == 64 && p.Nq >= 128, twok = p.hd == 64 && p.Nk >= 128;
    const dim3 gk((p.Nk + (twok ? 127 : 63)) / (twok ? 128 : 64), p.B * p.H);
    const dim3 gqw((p.Nq + (twoq ? 127 : 63)) / (twoq ? 128 : 64), p.B * p.H);
    dm_set_variant("attn_wide_mfma_bwd");
    if (p.hd == 64) {
      if (twok) hipLaunchKernelGGL((attn_wide_bwd_kv_kernel<64, 2>), gk, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((attn_wide_bwd_kv_kernel<64, 1>), gk, dim3(256), 0, st, p);
      DM_LAUNCH_CHECK();
      if (twoq) hipLaunchKernelGGL((attn_wide_bwd_q_kernel<64, 2>), gqw, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((attn_wide_bwd_q_kernel<64, 1>), gqw, dim3(256), 0, st, p);
    } else {
      hipLaunchKernelGGL((attn_wide_bwd_kv_kernel<128, 1>), gk, dim3(256), 0, st, p);
      DM_LAUNCH_CHECK();
      hipLaunchKernelGGL((attn_wide_bwd_q_kernel<128, 1>), gqw, dim3(256), 0, st, p);   // two query tiles would spill at 256 VGPRs
    }
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  // head dim 16 with 2304 keys per head (the 48x48 self-attention of the STL10 UNet): one pass over the scores
  // instead of two (attn_bwd_fused_kernel), when the caller brought the workspace for the two fp32 dQ partials
  if (p.hd == 16 && p.Nk == FB_NK && p.Nq % 64 == 0 && a->workspace &&
      a->ws_bytes >= dm_attn_bwd_workspace_bytes(p.B, p.H, p.Nq, p.Nk, p.hd) && fused_bwd_enabled()) {
    float* part = (float*)a->workspace;
    dm_set_variant(p.thresh ? "attn_bwd_fused<DROP=1>" : "attn_bwd_fused<DROP=0>");
    // 8 waves x 9 key tiles, two key splits.  Measured against it (B=64, dropout 0.1): 12 waves x 4 tiles x 3 splits
    // (168 VGPRs, 3 waves per SIMD) 1162 us and 16 x 3 x 3 (128 VGPRs) 1166 us against 1050 us -- the per-query-tile
    // fixed work and the third dQ partial cost more than the extra waves hide
    // 4 waves x 9 key tiles, four key splits: a workgroup takes HALF a CU (4 waves x 256 VGPRs), so a convolution
    // workgroup of the main stream can sit beside it and the VALU-bound attention waves share SIMDs with MFMA-bound
    // ones.  Against 8 waves x 9 tiles x two splits (a whole CU per workgroup): alone 1011 vs 1058 us (B=64, dropout 0.1),
    // in the training step 15.10 vs 15.37 ms.  (12 x 4 x 3 and 16 x 3 x 3: 1162 / 1166 us alone.)
    constexpr int NWF = 4, KTF = 9;
    const int rc2 = p.thresh ? launch_fused<true, NWF, KTF>(p, part, st) : launch_fused<false, NWF, KTF>(p, part, st);
    if (rc2 != DM_OK) return rc2;
    const int nsplit = FB_NK / (NWF * KTF * 16);
    long nb = ((long)p.B * p.H * p.Nq * 4 + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(attn_dq_merge_kernel, dim3((unsigned)nb), dim3(256), 0, st, p, (const float*)part,
                       p.scale * (p.thresh ? p.inv_keep : 1.f), nsplit);
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  const dim3 gkv((p.Nk + 63) / 64, p.B * p.H), gq((p.Nq + 127) / 128, p.B * p.H);
  const bool edge = (p.Nk & 63) != 0;
#define DM_BWD(DT, DROP)                                                                   \
  do {                                                                                     \
    hipLaunchKernelGGL((attn_bwd_kv_kernel<DT, DROP>), gkv, dim3(256), 0, st, p);          \
    DM_LAUNCH_CHECK();                                                                     \
    if (edge) hipLaunchKernelGGL((attn_bwd_q_kernel<DT, DROP, true>), gq, dim3(256), 0, st, p);  \
    else hipLaunchKernelGGL((attn_bwd_q_kernel<DT, DROP, false>), gq, dim3(256), 0, st, p);      \
  } while (0)
  dm_set_variant(p.hd > 16 ? (p.thresh ? "attn_flash_bwd<DT=2,DROP=1>" : "attn_flash_bwd<DT=2,DROP=0>")
                           : (p.thresh ? "attn_flash_bwd<DT=1,DROP=1>" : "attn_flash_bwd<DT=1,DROP=0>"));
  if (p.hd > 16) { if (p.thresh) DM_BWD(2, true); else DM_BWD(2, false); }
  else { if (p.thresh) DM_BWD(1, true); else DM_BWD(1, false); }
#undef DM_BWD
  DM_LAUNCH_CHECK();
  return DM_OK;
}
