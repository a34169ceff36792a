// Weight / bias gradient of the 3x3 convolutions, 1x1 convolutions and linears.
//
//   dW[co][ci][tap] += sum_pixels dy[pixel][co] * xv[pixel + tap][ci]
//
// is a GEMM whose reduction dimension is the PIXEL index, while both operands are stored
// pixel-major (NHWC).  The MFMA fragments (8 consecutive k per lane) are therefore read from
// the LDS images with the gfx950 transposing read ds_read_b64_tr_b16; the k -> pixel map is a
// free permutation (the same for both operands), chosen so that the 8 pixels one half-wave
// touches per read are consecutive => with a 160-byte pixel stride every read is
// bank-conflict free.
//
// One workgroup (256 threads, 2 per CU) owns a 64(co) x 64(ci) x TAPS slice of dW, keeps it
// in 144 accumulator registers per lane, and sweeps a strided subset of the spatial tiles
// (split-K).  The input patch (with halo) is staged once per tile and re-read for all 9 taps.
// Partials go to a per-workgroup fp32 slab in fragment order (16-byte coalesced stores); a
// second kernel sums the slabs in a fixed order (no atomics: gradients are bit-reproducible) into the
// fp32 gradient in the reference's OIHW layout.  The bias gradient
// rides along as one extra MFMA per k-step against an all-ones operand.
//
// Reference semantics replaced: autograd's weight/bias gradients of nn.Conv2d / nn.Linear
// under scaled_loss.backward() (main_stl10.py:502-503).
#include "dm_common.h"

namespace {

enum { WM_S1 = 0, WM_S2 = 1, WM_UP = 2, WM_GEMM = 4 };
constexpr int DYS = 160;           // dy tile pixel stride (bytes)
constexpr int SLAB_BIAS = 1024;    // floats

struct WgradParams {
  const bf16* x; long ldx;
  const bf16* dy; long lddy;
  float* slab;
  int B, Hi, Wi, Cin, Cout, Ho, Wo;
  int tiles_h, tiles_w, ntiles, nsplit;
  long M;
  int act_silu_in;
  const float* nsc; const float* nsh;  // [B][Cin] fp32 or null: the forward read silu(x * nsc + nsh) (dm_conv_args.norm_*)
};

__device__ __forceinline__ bf16x8 tr_pair(const char* a0, const char* a1) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// CIW (1x1 / linear gradients only): 64-channel ci sub-tiles per workgroup.  With CIW = 2 a workgroup owns 64 co x 128 ci:
// the dy fragments of a k-step feed 8 MFMAs instead of 4 (the 1x1 kernel is LDS-read bound: 10 transposed reads per 4
// MFMAs), dy is re-read from memory by half as many ci tiles, and half as many slabs are written.  The second 64 channels
// of a pixel live in a second LDS plane with the same 160-byte pixel stride (a 320-byte stride would be 2-way conflicted).
// The accumulator index plays the role the tap index plays for 3x3 weights.
template <int TH, int TW, int S, int XPITCH, int XSTRIDE, int MODE, int TAPS, bool NORM = false, int CIW = 1>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
  static_assert(CIW == 1 || (MODE == WM_GEMM && TAPS == 1 && !NORM), "ci sub-tiles are a 1x1 / linear option");
  constexpr int NA = TAPS * CIW;   // accumulator blocks (taps, or ci sub-tiles)
  constexpr int PH = (MODE == WM_GEMM) ? TH : ((S == 1) ? TH + 2 : 2 * TH + 1);
  constexpr int PW = (MODE == WM_GEMM) ? TW : ((S == 1) ? TW + 2 : 2 * TW + 1);
  constexpr int NPIX = TH * TW;
  constexpr int KSTEPS = NPIX / 32;
  constexpr int X_PLANE = ((PH * XPITCH * XSTRIDE + 15) / 16) * 16;
  constexpr int X_BYTES = X_PLANE * CIW;
  static_assert(NPIX % 32 == 0 && TW % 8 == 0, "k-step = 32 pixels, 8 consecutive pixels per row segment");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;
  char* sD = smem + X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x, cot = blockIdx.y, cit = blockIdx.z;
  const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;

  f32x4 acc[NA][4];
  f32x4 accb[4];
#pragma unroll
  for (int t = 0; t < NA; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) accb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = (cit == 0) && (wave == 0);
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;

  // per-lane pixel (within a k-step) for the two transposed reads
  int kk0 = 16 * (g >> 1) + 4 * (g & 1) + qq;  // half 0; half 1 = +8
  int xo[2], dq[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int kk = kk0 + 8 * h;
    dq[h] = kk * DYS + pp * 8;
    xo[h] = 0;  // filled per k-step (depends on row/col)
  }

  // Staging is software-pipelined over the tiles this workgroup sweeps: the x patch and dy tile of
  // the NEXT tile are fetched into registers (all loads in flight together) while the MFMAs of the
  // current tile run, and written to LDS after the barrier that retires the current images.
  constexpr int PPP = 8 * CIW;           // 16-byte pieces per pixel of the x image
  constexpr int X_PIECES = PH * PW * PPP;
  constexpr int NPX = (X_PIECES + 255) / 256;
  constexpr int NPD = (NPIX * 8 + 255) / 256;
  static_assert((NPIX * 8) % 256 == 0, "dy tile pieces must divide evenly over the workgroup");

  static_assert(NPX <= 32, "validity bits of a thread's x pieces live in one word");
  uint32_t xmask = 0u;   // NORM: bit i = piece i of the fetched x patch lies inside the image (padding stays zero)
  int xb = 0;            // NORM: sample of the fetched x patch (row of the scale / shift tables)
  auto load_tile = [&](int tile, u32x4(&px)[NPX], u32x4(&pd)[NPD], bool do_x, bool do_d) {
    int b = 0, h0 = 0, w0 = 0;
    long m0 = 0;
    if constexpr (MODE == WM_GEMM) {
      m0 = (long)tile * NPIX;
    } else {
      int t = tile;
      const int tw_i = t % p.tiles_w; t /= p.tiles_w;
      const int th_i = t % p.tiles_h;
      b = t / p.tiles_h;
      h0 = th_i * TH; w0 = tw_i * TW;
    }
    const int vh0 = (MODE == WM_S2) ? 2 * h0 - 1 : h0 - 1;
    const int vw0 = (MODE == WM_S2) ? 2 * w0 - 1 : w0 - 1;
    // Buffer loads: an out-of-range offset returns 0, so padding / overhang / ragged rows need no select on the
    // loaded value.  (A `valid ? v : 0` right after the load made the compiler wait for the whole prefetch BEFORE the
    // MFMA phase it was meant to overlap -- s_waitcnt vmcnt(0) ahead of the k-loop.)  GEMM mode: 2 GiB windows.
    constexpr int OOB = 0x7ffffff0;
    if (do_x) { xmask = 0u; xb = b; }
    if (do_x) {
      const bf16* xbase;
      long xspan;   // elements addressable from xbase
      long xrow0 = 0;
      if constexpr (MODE == WM_GEMM) {
        xrow0 = m0;
        xbase = p.x + m0 * p.ldx;
        xspan = (p.M - m0 - 1) * p.ldx + p.Cin;
      } else {
        xbase = p.x + (long)b * p.Hi * p.Wi * p.ldx;
        xspan = ((long)p.Hi * p.Wi - 1) * p.ldx + p.Cin;
      }
      (void)xrow0;
      const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, (int)(xspan * 2 > 0x7fffffffL ? 0x7fffffffL : xspan * 2), 0x00020000);
#pragma unroll
      for (int i = 0; i < NPX; ++i) {
        const int idx = tid + i * 256;
        const int piece = idx % PPP;
        const int pix = idx / PPP;
        const int pr = pix / PW, pc = pix % PW;
        bool valid = idx < X_PIECES;
        int off;
        const int ch = cit * 64 * CIW + piece * 8;
        if constexpr (MODE == WM_GEMM) {
          valid = valid && (m0 + pix < p.M) && (ch < p.Cin);
          off = (pix * (int)p.ldx + ch) * 2;
        } else {
          const int vh = vh0 + pr, vw = vw0 + pc;
          int sh = vh, sw = vw;
          if constexpr (MODE == WM_UP) {
            valid = valid && (vh >= 0) && (vh < 2 * p.Hi) && (vw >= 0) && (vw < 2 * p.Wi);
            sh = vh >> 1; sw = vw >> 1;
          } else {
            valid = valid && (vh >= 0) && (vh < p.Hi) && (vw >= 0) && (vw < p.Wi);
          }
          off = ((sh * p.Wi + sw) * (int)p.ldx + ch) * 2;
        }
        px[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, valid ? off : OOB, 0, 0));
        if constexpr (NORM) xmask |= (valid ? 1u : 0u) << i;
      }
    }
    if (do_d) {
      const bf16* dbase;
      long dspan;
      if constexpr (MODE == WM_GEMM) {
        dbase = p.dy + m0 * p.lddy;
        dspan = (p.M - m0 - 1) * p.lddy + p.Cout;
      } else {
        dbase = p.dy + (long)b * p.Ho * p.Wo * p.lddy;
        dspan = ((long)p.Ho * p.Wo - 1) * p.lddy + p.Cout;
      }
      const auto rd = __builtin_amdgcn_make_buffer_rsrc((void*)dbase, 0, (int)(dspan * 2 > 0x7fffffffL ? 0x7fffffffL : dspan * 2), 0x00020000);
#pragma unroll
      for (int i = 0; i < NPD; ++i) {
        const int idx = tid + i * 256;
        const int piece = idx & 7;
        const int q = idx >> 3;
        const int ch = cot * 64 + piece * 8;
        int off;
        bool valid = ch < p.Cout;
        if constexpr (MODE == WM_GEMM) {
          valid = valid && (m0 + q < p.M);
          off = (q * (int)p.lddy + ch) * 2;
        } else {
          const int r = q / TW, c = q % TW;
          valid = valid && (h0 + r < p.Ho) && (w0 + c < p.Wo);  // tile overhangs the image
          off = (((h0 + r) * p.Wo + (w0 + c)) * (int)p.lddy + ch) * 2;
        }
        pd[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, valid ? off : OOB, 0, 0));
      }
    }
  };
  auto store_tile = [&](u32x4(&px)[NPX], u32x4(&pd)[NPD], bool do_x, bool do_d) {
    if constexpr (NORM) {
      if (do_x) {
        // the thread's pieces all hold channels [cit*64 + (tid&7)*8, +8) of sample xb; four channels at a time
        // (the kernel has no registers to spare next to its 160 accumulators)
        const long o = (long)xb * p.Cin + cit * 64 + (tid & 7) * 8;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(p.nsc + o + 4 * hf);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(p.nsh + o + 4 * hf);
#pragma unroll
          for (int i = 0; i < NPX; ++i) {
            bf16x8 h = __builtin_bit_cast(bf16x8, px[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) h[4 * hf + j] = (bf16)dm_silu((float)h[4 * hf + j] * sc[j] + sh[j]);
            px[i] = __builtin_bit_cast(u32x4, h);
          }
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i)
          if (!((xmask >> i) & 1u)) px[i] = u32x4{0u, 0u, 0u, 0u};   // zero padding stays zero
      }
    }
    if (do_x)
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int idx = tid + i * 256;
      const int piece = idx % PPP;
      const int pix = idx / PPP;
      const int pr = pix / PW, pc = pix % PW;
      u32x4 v = px[i];
      if (p.act_silu_in) {
        bf16x8 h = __builtin_bit_cast(bf16x8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (bf16)dm_silu((float)h[j]);
        v = __builtin_bit_cast(u32x4, h);
      }
      if (X_PIECES % 256 == 0 || idx < X_PIECES)
        *reinterpret_cast<u32x4*>(sX + (piece >> 3) * X_PLANE + (pr * XPITCH + pc) * XSTRIDE + (piece & 7) * 16) = v;
    }
    if (do_d)
#pragma unroll
    for (int i = 0; i < NPD; ++i) {
      const int idx = tid + i * 256;
      *reinterpret_cast<u32x4*>(sD + (idx >> 3) * DYS + (idx & 7) * 16) = pd[i];
    }
  };

  // The prefetch registers stay live across the MFMA phase; tile shapes whose images need more than
  // 40 of them would spill next to the 160 accumulators, so those fetch (batched) right before the write.
  constexpr bool PREFETCH = (NPX + NPD) <= (NA >= 9 ? 10 : 16);   // (few accumulators: room for a deeper prefetch)
  u32x4 px[NPX], pd[NPD];
  if (PREFETCH && split < p.ntiles) load_tile(split, px, pd, true, true);
  for (int tile = split; tile < p.ntiles; tile += p.nsplit) {
    __syncthreads();
    if constexpr (PREFETCH) {
      store_tile(px, pd, true, true);
    } else {
      load_tile(tile, px, pd, true, false);
      store_tile(px, pd, true, false);
      load_tile(tile, px, pd, false, true);
      store_tile(px, pd, false, true);
    }
    __syncthreads();
    if (PREFETCH && tile + p.nsplit < p.ntiles) load_tile(tile + p.nsplit, px, pd, true, true);

#pragma unroll 1
    for (int ksi = 0; ksi < KSTEPS; ++ksi) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kk = ksi * 32 + kk0 + 8 * h;
        const int r = kk / TW, c = kk % TW;
        xo[h] = ((S * r) * XPITCH + S * c) * XSTRIDE + wave * 32 + pp * 8;
      }
      bf16x8 a[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        a[t] = tr_pair(sD + ksi * 32 * DYS + dq[0] + t * 32, sD + ksi * 32 * DYS + dq[1] + t * 32);
      if (do_bias) {
#pragma unroll
        for (int t = 0; t < 4; ++t) accb[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], ones, accb[t], 0, 0, 0);
      }
      // the x fragment of tap + 1 is read while the four MFMAs of tap run (two alternating registers quads: with one,
      // every tap paid the LDS latency between its read and its first MFMA)
      auto xfrag = [&](int tap) {
        const int kh = (TAPS == 9) ? tap / 3 : 0, kw = (TAPS == 9) ? tap % 3 : 0;
        const int tapoff = (TAPS == 9) ? (kh * XPITCH + kw) * XSTRIDE : tap * X_PLANE;   // 1x1: ci sub-tile = LDS plane
        return tr_pair(sX + xo[0] + tapoff, sX + xo[1] + tapoff);
      };
      bf16x8 bv = xfrag(0);
#pragma unroll
      for (int tap = 0; tap < NA; ++tap) {
        bf16x8 bn = bv;
        if (tap + 1 < NA) bn = xfrag(tap + 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[tap][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], bv, acc[tap][t], 0, 0, 0);
        bv = bn;
      }
    }
  }

  // ---- write the slab (fragment order): [tap][cosub][wave][lane][4] then bias [cosub][lane][4]
  const long slab_floats = (long)NA * 4 * 4 * 256 + SLAB_BIAS;
  float* slab = p.slab + (((long)split * gridDim.y + cot) * gridDim.z + cit) * slab_floats;
#pragma unroll
  for (int tap = 0; tap < NA; ++tap)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      *reinterpret_cast<f32x4*>(slab + (((tap * 4 + t) * 4 + wave) * 64 + lane) * 4) = acc[tap][t];
  if (do_bias) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      *reinterpret_cast<f32x4*>(slab + (long)NA * 4 * 4 * 256 + (t * 64 + lane) * 4) = accb[t];
  }
}

// dw[(co*K + ci)*taps + tap] += sum_split slab[...]; dbias[co] += ...
// Deterministic (bit-reproducible) reduction, one launch, no atomics: a 1024-thread block owns Q = 1024 / L
// consecutive accumulator quads; its L "slab lanes" each sum a strided subset of the split-K slabs (four independent
// 16-byte loads in flight per thread, runs of Q * 16 coalesced bytes), and the first Q threads add the L partial sums
// in lane order before the single read-modify-write of dw.  L follows the split count (64+ slabs: L = 16; the few
// slabs of a large gradient: L = 4), so every shape has enough loads in flight.  The blocks behind
// the accumulator range do the same for the bias column (64 channels x 16 slab lanes).
// ciw > 1 (1x1 gradients with ci sub-tiles): the slab's block index is the ci sub-tile, not a tap (`taps` = ciw there).
template <int L>
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                            float* __restrict__ dbias, int nsplit, int ncot, int ncit,
                                                            int taps, int Kreal, int Nreal, int nblk_main, int ciw) {
  constexpr int Q = 1024 / L;
  __shared__ f32x4 red[1024];
  const long slab_floats = (long)taps * 4 * 4 * 256 + SLAB_BIAS;
  const long per_tile = (long)taps * 4 * 4 * 256;
  const long sp_stride = (long)ncot * ncit * slab_floats;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < nblk_main) {
    const int ql = tid % Q, sl = tid / Q;
    const long total4 = (long)ncot * ncit * per_tile / 4;  // one quad = the 4 accumulator registers of an MFMA lane
    const long i = (long)blockIdx.x * Q + ql;
    const bool in = i < total4;
    const long tile = in ? i / (per_tile / 4) : 0;
    const int e = in ? (int)(i % (per_tile / 4)) * 4 : 0;
    const float* src = slab + tile * slab_floats + e;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (in) {
      int sp = sl;
      for (; sp + 3 * L < nsplit; sp += 4 * L) {
        s0 += *reinterpret_cast<const f32x4*>(src + (long)sp * sp_stride);
        s1 += *reinterpret_cast<const f32x4*>(src + (long)(sp + L) * sp_stride);
        s2 += *reinterpret_cast<const f32x4*>(src + (long)(sp + 2 * L) * sp_stride);
        s3 += *reinterpret_cast<const f32x4*>(src + (long)(sp + 3 * L) * sp_stride);
      }
      for (; sp < nsplit; sp += L) s0 += *reinterpret_cast<const f32x4*>(src + (long)sp * sp_stride);
    }
    red[tid] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && in) {
      f32x4 sum = red[ql];
#pragma unroll 4
      for (int w = 1; w < L; ++w) sum += red[w * Q + ql];
      const int cot = (int)(tile / ncit), cit = (int)(tile % ncit);
      const int ln = (e >> 2) & 63, wv = (e >> 8) & 3, t = (e >> 10) & 3, tap = e >> 12;
      const int co = cot * 64 + t * 16 + (ln >> 4) * 4;
      const int ci = (ciw > 1 ? (cit * ciw + tap) * 64 : cit * 64) + wv * 16 + (ln & 15);
      if (ci < Kreal) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (co + r < Nreal) {
            if (ciw > 1) dw[(long)(co + r) * Kreal + ci] += sum[r];
            else dw[((long)(co + r) * Kreal + ci) * taps + tap] += sum[r];
          }
      }
    }
    return;
  }
  // bias column: 64 channels per block, 16 slab lanes
  const int cot = (int)blockIdx.x - nblk_main, cl = tid & 63, sl = tid >> 6;
  const int co = cot * 64 + cl;
  const int t = cl >> 4, q = (cl & 15) >> 2, reg = cl & 3;
  const float* src = slab + ((long)cot * ncit) * slab_floats + per_tile + (t * 64 + q * 16) * 4 + reg;  // lane q*16 = column 0
  float s0 = 0.f, s1 = 0.f;
  int sp = sl;
  for (; sp + 16 < nsplit; sp += 32) {
    s0 += src[(long)sp * sp_stride];
    s1 += src[(long)(sp + 16) * sp_stride];
  }
  if (sp < nsplit) s0 += src[(long)sp * sp_stride];
  red[tid] = f32x4{s0 + s1, 0.f, 0.f, 0.f};
  __syncthreads();
  if (sl == 0 && co < Nreal) {
    float sum = red[cl][0];
#pragma unroll
    for (int w = 1; w < 16; ++w) sum += red[w * 64 + cl][0];
    dbias[co] += sum;
  }
}

struct Plan { int th, tw, ntiles, nsplit, ncot, ncit, ciw; long slab_floats; };

Plan make_plan(int B, int H, int W, int Cin, int Cout, int taps, int mode) {
  Plan pl{};
  // 1x1 / linear gradients over 256 input channels and more: two ci sub-tiles per workgroup (wgrad_kernel's CIW).  Measured
  // (tools/convbench.py, B=64, same box, CIW 1 -> 2): 256->128 M=147k 33.5 -> 29.8 us, 256->256 M=37k 21.1 -> 20.5; at 128 input
  // channels the single ci tile leaves twice the splits and the doubled slab loses: 128->64 M=590k 41.2 -> 46.7, 128->128 22.5 -> 23.5.
  static const int ciw_min = [] { const char* e = getenv("DM_WGRAD_CIW_MINCIN"); return e ? atoi(e) : 256; }();   // developer switch
  pl.ciw = (taps == 1 && Cin % 128 == 0 && Cin >= ciw_min) ? 2 : 1;
  pl.ncot = (Cout + 63) / 64; pl.ncit = (Cin + 64 * pl.ciw - 1) / (64 * pl.ciw);
  pl.slab_floats = (long)taps * pl.ciw * 4 * 4 * 256 + SLAB_BIAS;
  int Ho = H, Wo = W;
  if (mode == 1) { Ho = H / 2; Wo = W / 2; }
  if (mode == 2) { Ho = 2 * H; Wo = 2 * W; }
  if (taps == 1) {
    const long M = (long)B * H * W;
    pl.th = 8; pl.tw = 16; pl.ntiles = (int)((M + 127) / 128);
  } else {
    auto up = [](int v, int t) { return (v + t - 1) / t; };
    const long a16 = (long)up(Ho, 8) * 8 * up(Wo, 16) * 16, a24 = (long)up(Ho, 8) * 8 * up(Wo, 24) * 24,
               a8 = (long)up(Ho, 8) * 8 * up(Wo, 8) * 8;
    pl.th = 8;
    if (mode == 1) pl.tw = 8;
    else if (a16 <= a24 && a16 <= a8) pl.tw = 16;
    else if (a24 <= a8) pl.tw = 24;
    else pl.tw = 8;
    pl.ntiles = B * up(Ho, pl.th) * up(Wo, pl.tw);
  }
  // 512 workgroups (two per CU) in total: measured best against 384 / 256 on every conv layer but one
  int ns = 512 / (pl.ncot * pl.ncit);
  if (ns < 1) ns = 1;
  if (ns > pl.ntiles) ns = pl.ntiles;
  pl.nsplit = ns;
  return pl;
}

template <int TH, int TW, int S, int XPITCH, int XSTRIDE, int MODE, int TAPS, bool NORM = false, int CIW = 1>
int launch_wgrad(const WgradParams& p, const Plan& pl, hipStream_t st) {
  constexpr int PH = (MODE == WM_GEMM) ? TH : ((S == 1) ? TH + 2 : 2 * TH + 1);
  constexpr int X_BYTES = ((PH * XPITCH * XSTRIDE + 15) / 16) * 16 * CIW;
  constexpr int LDS = X_BYTES + TH * TW * DYS;
  static_assert(LDS <= 80 * 1024, "two workgroups per CU");
  auto k = wgrad_kernel<TH, TW, S, XPITCH, XSTRIDE, MODE, TAPS, NORM, CIW>;
  static unsigned long long attr_devs = 0ull;
  if (dm_first_on_device(attr_devs)) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return DM_ERR_LAUNCH;
  }
  static const DmVariantName vn(NORM ? "wgrad<TH=%d,TW=%d,S=%d,MODE=%d,TAPS=%d,NORM>"
                                     : (CIW > 1 ? "wgrad<TH=%d,TW=%d,S=%d,MODE=%d,TAPS=%d,CIW=2>" : "wgrad<TH=%d,TW=%d,S=%d,MODE=%d,TAPS=%d>"),
                                TH, TW, S, MODE, TAPS);
  dm_set_variant(vn.s);
  hipLaunchKernelGGL(k, dim3(pl.nsplit, pl.ncot, pl.ncit), dim3(256), LDS, st, p);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

}  // namespace

extern "C" long dm_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int taps, int mode) {
  const Plan pl = make_plan(B, H, W, Cin, Cout, taps, mode);
  return (long)pl.nsplit * pl.ncot * pl.ncit * pl.slab_floats * 4 + 256;
}

extern "C" int dm_wgrad(const dm_wgrad_args* a, dm_stream_t stream) {
  DM_REQUIRE(a && a->x && a->dy && a->dw && a->workspace);
  DM_REQUIRE((a->taps == 9 || a->taps == 1) && a->B > 0 && a->ldx % 8 == 0 && a->lddy % 8 == 0);
  DM_REQUIRE(a->Cin % 8 == 0 && a->Cout % 8 == 0);
  DM_REQUIRE(a->taps == 1 || (a->mode >= 0 && a->mode <= 2 && a->Cin % 64 == 0 && a->Cout % 64 == 0));
  DM_REQUIRE(a->ws_bytes >= dm_wgrad_workspace_bytes(a->B, a->H, a->W, a->Cin, a->Cout, a->taps, a->mode));
  const Plan pl = make_plan(a->B, a->H, a->W, a->Cin, a->Cout, a->taps, a->mode);
  WgradParams p{};
  p.x = (const bf16*)a->x; p.ldx = a->ldx; p.dy = (const bf16*)a->dy; p.lddy = a->lddy;
  p.slab = (float*)a->workspace;
  p.B = a->B; p.Hi = a->H; p.Wi = a->W; p.Cin = a->Cin; p.Cout = a->Cout;
  p.Ho = a->H; p.Wo = a->W;
  if (a->taps == 9 && a->mode == 1) { DM_REQUIRE(a->H % 2 == 0 && a->W % 2 == 0); p.Ho = a->H / 2; p.Wo = a->W / 2; }
  if (a->taps == 9 && a->mode == 2) { p.Ho = a->H * 2; p.Wo = a->W * 2; }
  p.tiles_h = (p.Ho + pl.th - 1) / pl.th; p.tiles_w = (p.Wo + pl.tw - 1) / pl.tw; p.ntiles = pl.ntiles; p.nsplit = pl.nsplit;
  p.M = (long)a->B * a->H * a->W;
  p.act_silu_in = a->act_silu_in;
  DM_REQUIRE((a->norm_scale == nullptr) == (a->norm_shift == nullptr));
  DM_REQUIRE(!a->norm_scale || (a->taps == 9 && a->mode == 0 && !a->act_silu_in));
  p.nsc = a->norm_scale; p.nsh = a->norm_shift;
  hipStream_t st = (hipStream_t)stream;
  int rc = DM_ERR_ARG;
  if (p.nsc) {
    if (pl.tw == 16) rc = launch_wgrad<8, 16, 1, 18, 160, WM_S1, 9, true>(p, pl, st);
    else if (pl.tw == 24) rc = launch_wgrad<8, 24, 1, 26, 160, WM_S1, 9, true>(p, pl, st);
    else rc = launch_wgrad<8, 8, 1, 10, 160, WM_S1, 9, true>(p, pl, st);
  } else
  if (a->taps == 1) rc = pl.ciw == 2 ? launch_wgrad<8, 16, 1, 16, 160, WM_GEMM, 1, false, 2>(p, pl, st)
                                     : launch_wgrad<8, 16, 1, 16, 160, WM_GEMM, 1>(p, pl, st);
  else if (a->mode == 1) rc = launch_wgrad<8, 8, 2, 17, 144, WM_S2, 9>(p, pl, st);
  else if (pl.tw == 16) rc = (a->mode == 0) ? launch_wgrad<8, 16, 1, 18, 160, WM_S1, 9>(p, pl, st)
                                            : launch_wgrad<8, 16, 1, 18, 160, WM_UP, 9>(p, pl, st);
  else if (pl.tw == 24) rc = (a->mode == 0) ? launch_wgrad<8, 24, 1, 26, 160, WM_S1, 9>(p, pl, st)
                                            : launch_wgrad<8, 24, 1, 26, 160, WM_UP, 9>(p, pl, st);
  else rc = (a->mode == 0) ? launch_wgrad<8, 8, 1, 10, 160, WM_S1, 9>(p, pl, st)
                           : launch_wgrad<8, 8, 1, 10, 160, WM_UP, 9>(p, pl, st);
  if (rc != DM_OK) return rc;
  const int Nreal = a->n_valid > 0 ? a->n_valid : a->Cout;
  const int Kreal = a->k_valid > 0 ? a->k_valid : a->Cin;
  const int slab_taps = a->taps * pl.ciw;
  const long total4 = (long)pl.ncot * pl.ncit * slab_taps * 1024;  // accumulator quads
  // slab lanes per block (measured, tools/convbench.py: 64 lanes were no faster than 16 on the 512-slab shapes)
  const int L = pl.nsplit >= 64 ? 16 : 4;
  const int Q = 1024 / L;
  const int nblk_main = (int)((total4 + Q - 1) / Q);
  const int nblk = nblk_main + (a->dbias ? pl.ncot : 0);
#define DM_RED(LL)                                                                                               \
  hipLaunchKernelGGL(wgrad_reduce_kernel<LL>, dim3(nblk), dim3(1024), 0, st, (const float*)a->workspace, a->dw, \
                     a->dbias, pl.nsplit, pl.ncot, pl.ncit, slab_taps, Kreal, Nreal, nblk_main, pl.ciw)
  if (L == 16) DM_RED(16); else DM_RED(4);
#undef DM_RED
  DM_LAUNCH_CHECK();
  return DM_OK;
}
