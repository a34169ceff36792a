// Implicit-GEMM convolution / GEMM forward (and data-gradient) for gfx950.
//
// Data layout: activations NHWC bf16 with an explicit pixel stride (ld*), so producers
// can write straight into channel slices of a concat buffer; weights are a bf16 shadow
// of the fp32 OIHW master, pre-packed in MFMA-fragment order (dm_pack_weight) so one
// B-stage is a contiguous, lane-linear 8 KiB block.
//
// Kernel structure (one workgroup = 256 threads = 4 waves, 2 workgroups / CU):
//   for each 64-channel chunk of Cin:
//     stage the input HALO PATCH ((TH+2)x(TW+2) pixels x 64 ch) into LDS once,
//     then sweep the 9 taps: every tap re-reads the patch at a shifted LDS address
//     (9x LDS reuse of each staged byte), weights stream through a double-buffered
//     8 KiB LDS stage.
//   MFMA: v_mfma_f32_16x16x32_bf16, weights as the A operand (rows = cout) and pixels
//   as the B operand (cols = pixel), so each lane ends up with 4 consecutive output
//   channels of one pixel -> 8-byte NHWC stores, per-channel bias/temb adds stay in-lane.
//   The patch pixel stride (160 B / 144 B) and pitch are chosen so that every
//   ds_read_b128 fragment read is bank-conflict free (brute-forced against the gfx950
//   lane-group table).
//
// Reference semantics replaced: nn.Conv2d 3x3 s1/s2 p1, nearest-x2 + conv3x3
// (unet.py:74-75,225-229,234,317-319), nn.Conv2d 1x1 / nn.Linear (unet.py:79,113-114,
// 160-163,261,320,333-335) and their autograd data-gradients.
#include "dm_common.h"

#include <stdio.h>
#include <stdlib.h>

namespace {

// MODE_Q8: stride-1 conv on 8x8 images, FOUR images per workgroup tile (the latent model's innermost stage): each image
// gets its own 10x10 halo patch inside a 20x20 LDS patch and one wave; the weight stages are shared by 256 pixels
// instead of 64 (a quarter of the L2 -> LDS weight traffic, four m-tiles per weight fragment).
enum { MODE_S1 = 0, MODE_S2 = 1, MODE_UP = 2, MODE_ZI = 3, MODE_GEMM = 4, MODE_Q8 = 5 };

struct ConvParams {
  const bf16* x; long ldx;       // input pixel stride (elements)
  const char* w;                 // packed weights
  const float* bias;             // [Cout] or null
  const float* chan_add;         // [B, Cout] fp32 (time-embedding projection) or null
  const bf16* res; long ldr;     // residual to add (same pixel indexing as y) or null
  bf16* y; long ldy;             // bf16 output (null if yf used)
  float* yf; long ldyf;          // optional fp32 output
  int B, Hi, Wi, Cin, Cout, Ho, Wo;
  int tiles_h, tiles_w;
  int ntiles;                    // conv_ws_kernel: tiles along grid.x's walk (set by launch_conv_ws)
  long M;                        // GEMM mode: rows
  int rows_per_batch;            // GEMM mode: rows per chan_add batch index
  int n_valid;                   // number of real output channels to store
  int pool2;                     // sum 2x2 output pixels (dgrad of nearest-x2)
  int act_silu_in;               // apply SiLU to the input while staging
  const float* nsc; const float* nsh;  // [B][Cin] fp32 or null: stage silu(x * nsc + nsh) (GroupNorm + SiLU folded into
                                       // the consumer: dm_gn_prepare supplies the per-(sample, channel) scale / shift)
  // dropout on the output (after bias/chan_add, before res): keep iff rng>=thresh, scale
  uint32_t drop_seed; uint32_t drop_thresh; float drop_scale;  // dm_common.h dropout convention
  const uint32_t* seed_ctr;      // device-side seed counter (graph replay) or null
};

// Epilogue shared by the conv and the streaming GEMM kernels: lane holds couts [cq..cq+3] of pixel
// (lane & 15) of each of its wave's m-tiles.
template <int MT, int NT, int TW, int MODE, bool WIDE_ONLY = false>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x4 (&acc)[MT][NT], int nb, int wave, int lane,
                                              int b, int h0, int w0, long m0) {
  const int cq = (lane >> 4) * 4;
  // ---- wide path: bf16 output, whole channel tiles.  A lane holds 4 couts of tile nt and 4 of tile nt+1 (two
  // 8-byte stores); one v_permlane16_swap per dword trades halves with the lane 16 away, after which every lane
  // owns 8 CONSECUTIVE couts -> one 16-byte store (half the store instructions, 64-byte row segments).
  if constexpr (NT % 2 == 0) {
    if (p.y && !p.yf && !p.pool2 && p.n_valid == p.Cout && (p.ldy & 7) == 0 &&
        (reinterpret_cast<unsigned long long>(p.y) & 15) == 0) {
      const int g = lane >> 4;
      const bool wide_res = p.res && (p.ldr & 7) == 0 && (reinterpret_cast<unsigned long long>(p.res) & 15) == 0;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int q = (wave * MT + mt) * 16 + (lane & 15);
        long pix;
        int bb = b;
        bool ok;
        if constexpr (MODE == MODE_GEMM) {
          pix = m0 + q;
          ok = pix < p.M;
          bb = ok ? (int)(pix / p.rows_per_batch) : 0;
        } else if constexpr (MODE == MODE_Q8) {
          bb = b + (q >> 6);                           // image of the quad
          ok = bb < p.B;
          pix = (long)bb * 64 + (q & 63);
          if (!ok) bb = 0;
        } else {
          const int r = q / TW, c = q % TW;
          ok = (h0 + r < p.Ho) && (w0 + c < p.Wo);  // tile may overhang the image
          pix = ((long)b * p.Ho + (h0 + r)) * p.Wo + (w0 + c);
        }
        if (!ok) pix = 0;
        uint32_t rs = 0u;
        if (p.drop_thresh) rs = dm_drop_row(dm_eff_seed(p.drop_seed, p.seed_ctr), (uint32_t)pix);
#pragma unroll
        for (int nt = 0; nt < NT; nt += 2) {
          u32x2 pk[2];
          const int co8 = (nb * NT + nt + (g & 1)) * 16 + 4 * (g & ~1);  // this lane's 8 consecutive couts
          u32x2 rres[2] = {{0u, 0u}, {0u, 0u}};
          if (p.res) {
            // the residual comes in the same 16-byte shape and is dealt back to the MFMA layout by the same swap
            u32x4 rv = {0u, 0u, 0u, 0u};
            if (wide_res && ok) rv = *reinterpret_cast<const u32x4*>(p.res + pix * p.ldr + co8);
            const auto r0 = __builtin_amdgcn_permlane16_swap(rv[0], rv[2], false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(rv[1], rv[3], false, false);
            rres[0] = u32x2{r0[0], r1[0]};
            rres[1] = u32x2{r0[1], r1[1]};
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int co = (nb * NT + nt + h) * 16 + cq;
            f32x4 v = acc[mt][nt + h];
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + co);
            if (p.chan_add) v += *reinterpret_cast<const f32x4*>(p.chan_add + (long)bb * p.Cout + co);
            if (p.drop_thresh) {
              const uint32_t b0 = dm_drop_bits(rs, (uint32_t)co >> 1), b1 = dm_drop_bits(rs, ((uint32_t)co >> 1) + 1);
#pragma unroll
              for (int j = 0; j < 4; ++j)
                v[j] = dm_keep16(j < 2 ? b0 : b1, (uint32_t)j, p.drop_thresh) ? v[j] * p.drop_scale : 0.f;
            }
            if (p.res) {
              bf16x4 rv = __builtin_bit_cast(bf16x4, rres[h]);
              if (!wide_res) rv = *reinterpret_cast<const bf16x4*>(p.res + pix * p.ldr + co);
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] += (float)rv[j];
            }
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
            pk[h] = __builtin_bit_cast(u32x2, o);
          }
          // odd 16-lane rows of pk[0] <-> even rows of pk[1]
          const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
          if (ok) {
            *reinterpret_cast<u32x4*>(p.y + pix * p.ldy + co8) = u32x4{s0[0], s1[0], s0[1], s1[1]};
          }
          // streaming GEMM: one column pair at a time (the scheduler otherwise batches all pairs' addresses, residual
          // loads and packs up front: ~60 VGPRs that cost the HBM-bound kernel a resident wave per SIMD)
          if constexpr (MODE == MODE_GEMM) __builtin_amdgcn_sched_barrier(0);
        }
      }
      return;
    }
  }
  if constexpr (WIDE_ONLY) return;  // the caller checked the wide path's conditions on the host
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = (nb * NT + nt) * 16 + cq;
    if (co >= p.n_valid) continue;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + co);
    if (p.pool2) {
      if constexpr (MODE != MODE_GEMM && MODE != MODE_Q8 && TW == 16 && (MT % 2) == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; mt += 2) {
          f32x4 v = acc[mt][nt] + acc[mt + 1][nt];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += __shfl_xor(v[j], 1, 64);
          const int r = wave * MT + mt;  // even output row of the pair
          if ((lane & 1) == 0 && (h0 + r) < p.Ho && (w0 + (lane & 15)) < p.Wo) {
            const int oh = (h0 + r) >> 1, ow = (w0 + (lane & 15)) >> 1;
            const long pix = ((long)b * (p.Ho >> 1) + oh) * (p.Wo >> 1) + ow;
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16)(v[j] + bias4[j]);
            *reinterpret_cast<bf16x4*>(p.y + pix * p.ldy + co) = o;
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int q = (wave * MT + mt) * 16 + (lane & 15);
      long pix;
      int bb = b;
      if constexpr (MODE == MODE_GEMM) {
        pix = m0 + q;
        if (pix >= p.M) continue;
        bb = (int)(pix / p.rows_per_batch);
      } else if constexpr (MODE == MODE_Q8) {
        bb = b + (q >> 6);
        if (bb >= p.B) continue;
        pix = (long)bb * 64 + (q & 63);
      } else {
        const int r = q / TW, c = q % TW;
        if (h0 + r >= p.Ho || w0 + c >= p.Wo) continue;  // tile overhangs the image
        pix = ((long)b * p.Ho + (h0 + r)) * p.Wo + (w0 + c);
      }
      f32x4 v = acc[mt][nt] + bias4;
      if (p.chan_add) v += *reinterpret_cast<const f32x4*>(p.chan_add + (long)bb * p.Cout + co);
      if (p.drop_thresh) {
        const uint32_t rs = dm_drop_row(dm_eff_seed(p.drop_seed, p.seed_ctr), (uint32_t)pix);
        const uint32_t b0 = dm_drop_bits(rs, (uint32_t)co >> 1), b1 = dm_drop_bits(rs, ((uint32_t)co >> 1) + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          v[j] = dm_keep16(j < 2 ? b0 : b1, (uint32_t)j, p.drop_thresh) ? v[j] * p.drop_scale : 0.f;
      }
      if (p.res) {
        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(p.res + pix * p.ldr + co);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += (float)rv[j];
      }
      if (p.yf) {
        if (co + 3 < p.n_valid) {
          *reinterpret_cast<f32x4*>(p.yf + pix * p.ldyf + co) = v;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co + j < p.n_valid) p.yf[pix * p.ldyf + co + j] = v[j];
        }
      }
      if (p.y) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
        *reinterpret_cast<bf16x4*>(p.y + pix * p.ldy + co) = o;
      }
    }
  }
}

template <int TH, int TW, int S, int PITCH, int STRIDE, int NT, int MODE, bool NORM = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvParams p) {
  constexpr int TAPS = (MODE == MODE_GEMM) ? 1 : 9;
  constexpr int PH = (MODE == MODE_GEMM) ? TH : (MODE == MODE_Q8) ? 20 : ((S == 1) ? TH + 2 : 2 * TH + 1);
  constexpr int PW = (MODE == MODE_GEMM) ? TW : (MODE == MODE_Q8) ? 20 : ((S == 1) ? TW + 2 : 2 * TW + 1);
  constexpr int MT = TH * TW / 64;  // m-tiles (16 pixels) per wave
  constexpr int A_BYTES = ((PH * PITCH * STRIDE + 15) / 16) * 16;
  constexpr int KSPS = (NT <= 4) ? 2 : 1;       // k-steps (of 32) per B stage
  constexpr int B_STAGE = KSPS * NT * 1024;     // bytes
  constexpr int B_PIECES = B_STAGE / 16;        // 16-byte pieces per stage
  constexpr int B_PER_THREAD = (B_PIECES + 255) / 256;
  constexpr int NS = TAPS * 2 / KSPS;           // stages per channel chunk
  static_assert(TH * TW % 64 == 0, "tile must hold a whole number of m-tiles per wave");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b, h0, w0;
  long m0 = 0;
  if constexpr (MODE == MODE_GEMM) {
    m0 = (long)blockIdx.x * (TH * TW);
    b = 0; h0 = 0; w0 = 0;
  } else if constexpr (MODE == MODE_Q8) {
    b = blockIdx.x * 4; h0 = 0; w0 = 0;  // first image of the quad
  } else {
    int tile = blockIdx.x;
    const int tw_i = tile % p.tiles_w; tile /= p.tiles_w;
    const int th_i = tile % p.tiles_h;
    b = tile / p.tiles_h;
    h0 = th_i * TH; w0 = tw_i * TW;
  }
  const int nb = blockIdx.y;
  const int NTG = p.Cout >> 4;

  int aoff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int q = (wave * MT + mt) * 16 + (lane & 15);
    int r = q / TW, c = q % TW;
    if constexpr (MODE == MODE_Q8) {  // image q >> 6 of the quad sits at patch origin (10 * (i >> 1), 10 * (i & 1))
      r = ((q >> 6) >> 1) * 10 + ((q & 63) >> 3);
      c = ((q >> 6) & 1) * 10 + (q & 7);
    }
    aoff[mt] = ((S * r) * PITCH + S * c) * STRIDE + (lane >> 4) * 16;
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ncc = (p.Cin + 63) >> 6;
  const int vh0 = (MODE == MODE_S2) ? 2 * h0 - 1 : h0 - 1;
  const int vw0 = (MODE == MODE_S2) ? 2 * w0 - 1 : w0 - 1;

  // ---- software pipeline over the flat stage index g = cc * NS + s.  Weight stages are linear in g
  // (packed k-step order), so their prefetch runs two stages ahead straight through channel-chunk
  // boundaries: stage j travels global -> br[j & 1] (issued at step j-2) -> LDS buffer j & 1 (written
  // at the end of step j-1) -> MFMA at step j.  The halo patch of the next chunk is fetched with all
  // its loads in flight at once, then written after the barrier that retires the current patch.
  constexpr int A_PIECES = PH * PW * 8;
  constexpr int NPA = (A_PIECES + 255) / 256;
  const int G = ncc * NS;

  static_assert(NPA <= 32, "validity bits of a thread's patch pieces live in one word");
  uint32_t pmask = 0u;  // bit i: piece i of this thread lies inside the image (zero padding must stay zero after nsc/nsh)
  auto load_patch = [&](int cc, u32x4(&pa)[NPA]) {
    pmask = 0u;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int idx = tid + i * 256;
      const int piece = idx & 7;
      const int pp = idx >> 3;
      const int pr = pp / PW, pc = pp % PW;
      bool valid = idx < A_PIECES;
      long src;
      const int ch = cc * 64 + piece * 8;
      if constexpr (MODE == MODE_GEMM) {
        const long m = m0 + pp;
        valid = valid && (m < p.M) && (ch < p.Cin);
        src = m * p.ldx + ch;
      } else if constexpr (MODE == MODE_Q8) {
        const int img = b + (pr / 10) * 2 + (pc / 10), lr = pr % 10 - 1, lc = pc % 10 - 1;
        valid = valid && (img < p.B) && (lr >= 0) && (lr < 8) && (lc >= 0) && (lc < 8);
        src = ((long)img * 64 + lr * 8 + lc) * p.ldx + ch;
      } else {
        const int vh = vh0 + pr, vw = vw0 + pc;
        int sh = vh, sw = vw;
        if constexpr (MODE == MODE_UP) {
          valid = valid && (vh >= 0) && (vh < 2 * p.Hi) && (vw >= 0) && (vw < 2 * p.Wi);
          sh = vh >> 1; sw = vw >> 1;
        } else if constexpr (MODE == MODE_ZI) {
          valid = valid && (vh >= 0) && (vh < 2 * p.Hi) && (vw >= 0) && (vw < 2 * p.Wi) &&
                  (((vh | vw) & 1) == 0);
          sh = vh >> 1; sw = vw >> 1;
        } else {
          valid = valid && (vh >= 0) && (vh < p.Hi) && (vw >= 0) && (vw < p.Wi);
        }
        src = (((long)b * p.Hi + sh) * p.Wi + sw) * p.ldx + ch;
      }
      // unconditional load (offset 0 is always readable) so that all NPA loads are in flight together
      const u32x4 v = *reinterpret_cast<const u32x4*>(p.x + (valid ? src : 0));
      pa[i] = valid ? v : u32x4{0u, 0u, 0u, 0u};
      if constexpr (NORM) pmask |= (valid ? 1u : 0u) << i;
    }
  };
  auto store_patch = [&](u32x4(&pa)[NPA], int cc) {
    if constexpr (NORM) {
      // the thread's pieces all hold channels [cc*64 + (tid&7)*8, +8) of sample b; four channels at a time
      const long o = (long)b * p.Cin + cc * 64 + (tid & 7) * 8;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.nsc + o + 4 * hf);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(p.nsh + o + 4 * hf);
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
          bf16x8 h = __builtin_bit_cast(bf16x8, pa[i]);
#pragma unroll
          for (int j = 0; j < 4; ++j) h[4 * hf + j] = (bf16)dm_silu((float)h[4 * hf + j] * sc[j] + sh[j]);
          pa[i] = __builtin_bit_cast(u32x4, h);
        }
      }
#pragma unroll
      for (int i = 0; i < NPA; ++i)
        if (!((pmask >> i) & 1u)) pa[i] = u32x4{0u, 0u, 0u, 0u};   // zero padding stays zero
    }
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int idx = tid + i * 256;
      const int piece = idx & 7;
      const int pp = idx >> 3;
      const int pr = pp / PW, pc = pp % PW;
      u32x4 v = pa[i];
      if (p.act_silu_in) {
        bf16x8 h = __builtin_bit_cast(bf16x8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (bf16)dm_silu((float)h[j]);
        v = __builtin_bit_cast(u32x4, h);
      }
      if (A_PIECES % 256 == 0 || idx < A_PIECES)
        *reinterpret_cast<u32x4*>(sA + (pr * PITCH + pc) * STRIDE + piece * 16) = v;
    }
  };
  const char* wbase = p.w + ((long)nb * NT * 64 + 0) * 16;
  auto load_b = [&](int g, u32x4(&br)[B_PER_THREAD]) {
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) {
      const int piece = tid + i * 256;
      const int ksl = piece / (NT * 64), rem = piece % (NT * 64);
      const long off = ((long)(g * KSPS + ksl) * NTG) * 1024 + rem * 16;
      if (B_PIECES % 256 == 0 || piece < B_PIECES) br[i] = *reinterpret_cast<const u32x4*>(wbase + off);
    }
  };
  auto store_b = [&](int buf, u32x4(&br)[B_PER_THREAD]) {
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) {
      const int piece = tid + i * 256;
      if (B_PIECES % 256 == 0 || piece < B_PIECES)
        *reinterpret_cast<u32x4*>(sB + buf * B_STAGE + piece * 16) = br[i];
    }
  };

  u32x4 br0[B_PER_THREAD], br1[B_PER_THREAD];
  {
    u32x4 pa[NPA];
    load_patch(0, pa);
    if (MODE == MODE_S2 && G > 1) load_b(1, br1);   // stage 1 first: the wait for stage 0 then covers it (vmcnt is in order)
    load_b(0, br0);
    if (MODE != MODE_S2 && G > 1) load_b(1, br1);
    store_patch(pa, 0);
    store_b(0, br0);
  }
  __syncthreads();

  int s = 0, cc = 0;
  auto step = [&](int g, u32x4(&br_free)[B_PER_THREAD], u32x4(&br_next)[B_PER_THREAD]) {
    // br_free: register set of stage g (already in LDS) -> refill with stage g + 2
    // br_next: holds stage g + 1 -> goes to LDS at the end of this step
    // Stride-2 tiles (short steps: 4-8 MFMAs) prefetch unconditionally with a clamped index: a branch around the loads
    // makes the waitcnt pass drain them (vmcnt(0)) at the store below.  Measured +8 % there; the 16x16 x 128-cout tiles
    // measured 6 % SLOWER with the exact two-stage waits and keep the branch.
    if constexpr (MODE == MODE_S2) load_b(min(g + 2, G - 1), br_free);
    else if (g + 2 < G) load_b(g + 2, br_free);
    const char* sBc = sB + (g & 1) * B_STAGE;
#pragma unroll
    for (int ksl = 0; ksl < KSPS; ++ksl) {
      const int tap = (KSPS == 2) ? s : (s >> 1);
      const int ks = (KSPS == 2) ? ksl : (s & 1);
      const int kh = (TAPS == 9) ? (tap * 11) >> 5 : 0;  // tap / 3 for tap < 9
      const int kw = (TAPS == 9) ? tap - 3 * kh : 0;
      const int tapoff = (kh * PITCH + kw) * STRIDE + ks * 64;
      bf16x8 av[MT], bv[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        av[mt] = *reinterpret_cast<const bf16x8*>(sA + aoff[mt] + tapoff);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bv[nt] = *reinterpret_cast<const bf16x8*>(sBc + (ksl * NT + nt) * 1024 + lane * 16);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[nt], av[mt], acc[mt][nt], 0, 0, 0);
    }
    ++s;
    if (s == NS && cc + 1 < ncc) {
      // chunk boundary: fetch the next patch, retire the current one, swap
      u32x4 pa[NPA];
      load_patch(cc + 1, pa);
      __syncthreads();
      store_patch(pa, cc + 1);
      s = 0; ++cc;
    }
    if (MODE == MODE_S2 || g + 1 < G) store_b((g + 1) & 1, br_next);   // (after the last step nobody reads that buffer)
    __syncthreads();
  };
  if constexpr (MODE == MODE_S2) {   // pairs + peeled tail: no mid-loop exit path into the loop header
    int g = 0;
    for (; g + 1 < G; g += 2) {
      step(g, br0, br1);
      step(g + 1, br1, br0);
    }
    if (g < G) step(g, br0, br1);
  } else {
    for (int g = 0; g < G; g += 2) {
      step(g, br0, br1);
      if (g + 1 < G) step(g + 1, br1, br0);
    }
  }

  conv_epilogue<MT, NT, TW, MODE>(p, acc, nb, wave, lane, b, h0, w0, m0);
}

template <int TH, int TW, int S, int PITCH, int STRIDE, int NT, int MODE, bool NORM = false>
int launch_conv(const ConvParams& p, hipStream_t st) {
  constexpr int PH = (MODE == MODE_GEMM) ? TH : (MODE == MODE_Q8) ? 20 : ((S == 1) ? TH + 2 : 2 * TH + 1);
  constexpr int A_BYTES = ((PH * PITCH * STRIDE + 15) / 16) * 16;
  constexpr int KSPS = (NT <= 4) ? 2 : 1;
  constexpr int LDS = A_BYTES + 2 * KSPS * NT * 1024;
  static_assert(LDS <= 80 * 1024, "two workgroups per CU must fit in 160 KiB of LDS");
  auto k = conv_mfma_kernel<TH, TW, S, PITCH, STRIDE, NT, MODE, NORM>;
  static unsigned long long attr_devs = 0ull;
  if (dm_first_on_device(attr_devs)) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return DM_ERR_LAUNCH;
  }
  dim3 grid;
  if (MODE == MODE_GEMM) grid = dim3((unsigned)((p.M + TH * TW - 1) / (TH * TW)), (unsigned)(p.Cout / (16 * NT)));
  else if (MODE == MODE_Q8) grid = dim3((unsigned)((p.B + 3) / 4), (unsigned)(p.Cout / (16 * NT)));
  else grid = dim3((unsigned)(p.B * p.tiles_h * p.tiles_w), (unsigned)(p.Cout / (16 * NT)));
  static const DmVariantName vn(NORM ? "conv_mfma<TH=%d,TW=%d,S=%d,NT=%d,MODE=%d,NORM>" : "conv_mfma<TH=%d,TW=%d,S=%d,NT=%d,MODE=%d>",
                                TH, TW, S, NT, MODE);
  dm_set_variant(vn.s);
  hipLaunchKernelGGL(k, grid, dim3(256), LDS, st, p);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// ------------------------------------------------------------------ wave-specialised persistent 3x3 conv
// Why: ablation of conv_mfma_kernel (profiles/r02_conv_ablation.txt) shows a SERIAL time budget -- for the 64->64 layer at
// 96x96, B=64: k-loop ~39 us + halo-patch fetch ~13 us + epilogue ~14 us = 66 us.  The fetch and the epilogue are
// HBM phases (the layer moves 170 MB: 27 us at the achievable 6.3 TB/s, against 17 us of MFMA at peak -- it is an
// HBM-roofline layer) and nothing overlaps them with the k-loop: a wave's vmcnt counter retires IN ORDER, so a patch
// prefetch or the epilogue's stores issued by a wave that also streams the weight stages block that wave's next weight
// wait until they are done; two co-resident workgroups do not hide it for each other either (staggering them changes
// nothing: a lone workgroup's k-loop is latency-bound and runs at half the pair's rate).
// So the roles get their own waves, i.e. their own vmcnt counters, inside one PERSISTENT workgroup of six waves:
//   waves 0-3  consumers: LDS -> MFMA k-loop and the epilogue.  They never wait on vmcnt in the loop, so a tile's
//              stores drain while the next tile's taps already run;
//   wave 4     weight loader: streams the packed weight stages global -> registers -> LDS double buffer, two stages
//              ahead, straight through tile boundaries (the stages repeat with period G);
//   wave 5     patch loader: holds the NEXT unit's (tile, 64-channel chunk) halo patch in its registers while the
//              current unit runs and writes it to LDS at the unit boundary.
// All six waves meet at one barrier per weight stage plus one per unit boundary.  Tile 8x24 output pixels (three
// m-tiles per consumer wave) so that the loader's patch fits its registers and the whole kernel stays under 168 VGPRs:
// three waves per SIMD, two workgroups (12 waves) per CU.
struct ConvWalk {   // this workgroup's walk over its (tile, channel-chunk) units
  int t_cur, t_nxt, tstride, ntiles, cc, ncc;
  __device__ bool has_next() const { return cc + 1 < ncc || t_nxt < ntiles; }
  __device__ bool tile_end() const { return cc + 1 == ncc; }
  __device__ void next(int& t, int& c) const {
    if (cc + 1 < ncc) { t = t_cur; c = cc + 1; } else { t = t_nxt; c = 0; }
  }
  __device__ void advance() {
    if (cc + 1 < ncc) { ++cc; } else { t_cur = t_nxt; t_nxt += tstride; cc = 0; }
  }
};

template <int NT, int MODE, int TW = 24>
__global__ __launch_bounds__(512, 4) void conv_ws_kernel(const ConvParams p) {
  // patch pitch / pixel stride: conflict-free for the ds_read_b128 lane groups (brute-forced over the gfx950 group table:
  // 24-wide tiles need pitch 32 at 160 B -- pitch 26 at 144 B, the first version, was up to 3-way conflicted on the
  // pixel-operand reads; 16-wide tiles pitch 18 at 160 B)
  constexpr int TH = 8, PH = TH + 2, PW = TW + 2, PITCH = (TW == 24) ? 32 : PW, STRIDE = 160, MT = TH * TW / 64;
  constexpr int A_BYTES = PH * PITCH * STRIDE;        // 37,440
  constexpr int KSPS = 2;                             // k-steps (of 32) per weight stage
  constexpr int B_STAGE = KSPS * NT * 1024;
  constexpr int NS = 9;                               // stages per 64-channel chunk (one tap each)
  constexpr int A_PIECES = PH * PW * 8;               // 16-byte pieces of a patch
  constexpr int NPA = (A_PIECES + 127) / 128;         // per lane of a patch loader wave (two of them)
  constexpr int NPB = B_STAGE / 16 / 128;             // per lane of a weight loader wave (two of them)
  static_assert(MODE == MODE_S1 || MODE == MODE_UP || MODE == MODE_ZI, "stride-1 family only");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = blockIdx.y;
  const int NTG = p.Cout >> 4;
  const int ncc = (p.Cin + 63) >> 6;
  const int G = ncc * NS;

  ConvWalk wk;
  wk.t_cur = blockIdx.x; wk.tstride = gridDim.x; wk.t_nxt = blockIdx.x + gridDim.x;
  wk.ntiles = p.ntiles; wk.cc = 0; wk.ncc = ncc;
  auto decode = [&](int t, int& b, int& h0, int& w0) {
    const int tw_i = t % p.tiles_w; t /= p.tiles_w;
    const int th_i = t % p.tiles_h;
    b = t / p.tiles_h; h0 = th_i * TH; w0 = tw_i * TW;
  };

  if (wave < 4) {
    // ================================================================= consumers
    int aoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int q = (wave * MT + mt) * 16 + (lane & 15);
      const int r = q / TW, c = q % TW;
      aoff[mt] = (r * PITCH + c) * STRIDE + (lane >> 4) * 16;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int gbuf = 0;
    __syncthreads();   // first patch and weight stage 0 are in LDS
    for (;;) {
      for (int s = 0; s < NS; ++s) {
        const char* sBc = sB + gbuf * B_STAGE;
        const int kh = (s * 11) >> 5, kw = s - 3 * kh;   // tap s: (s / 3, s % 3)
#pragma unroll
        for (int ksl = 0; ksl < KSPS; ++ksl) {
          const int tapoff = (kh * PITCH + kw) * STRIDE + ksl * 64;
          bf16x8 av[MT], bv[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            av[mt] = *reinterpret_cast<const bf16x8*>(sA + aoff[mt] + tapoff);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            bv[nt] = *reinterpret_cast<const bf16x8*>(sBc + (ksl * NT + nt) * 1024 + lane * 16);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[nt], av[mt], acc[mt][nt], 0, 0, 0);
        }
        gbuf ^= 1;
        __syncthreads();   // stage barrier
      }
      if (wk.tile_end()) {
        int b, h0, w0;
        decode(wk.t_cur, b, h0, w0);
        conv_epilogue<MT, NT, TW, MODE>(p, acc, nb, wave, lane, b, h0, w0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (!wk.has_next()) break;
      __syncthreads();     // unit barrier: the patch loader has swapped the patch
      wk.advance();
    }
  } else if (wave < 6) {
    // ================================================================= weight loaders (each half a stage)
    const int l2 = lane + (wave - 4) * 64;
    const char* wbase = p.w + ((long)nb * NT * 64) * 16;
    auto load_b = [&](int g, u32x4(&br)[NPB]) {
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        const int piece = l2 + i * 128;
        const int ksl = piece / (NT * 64), rem = piece % (NT * 64);
        br[i] = *reinterpret_cast<const u32x4*>(wbase + ((long)(g * KSPS + ksl) * NTG) * 1024 + rem * 16);
      }
    };
    auto store_b = [&](int buf, u32x4(&br)[NPB]) {
#pragma unroll
      for (int i = 0; i < NPB; ++i)
        *reinterpret_cast<u32x4*>(sB + buf * B_STAGE + (l2 + i * 128) * 16) = br[i];
    };
    // units of this workgroup -> number of stages it will run
    const int nmine = (wk.ntiles - wk.t_cur + wk.tstride - 1) / wk.tstride;
    const int total = nmine * G;
    // ONE register set: during stage g it holds stage g + 1 (in flight since the top of stage g - 1's slot), is written
    // to the LDS buffer stage g - 1 ran from, and is refilled with stage g + 2 right away -- a full stage of flight time
    u32x4 bx[NPB];
    load_b(0, bx);
    store_b(0, bx);
    load_b(1 % G, bx);
    int gl = 2 % G;
    __syncthreads();
    int sc = 0;
    for (int g = 0; g < total; ++g) {
      store_b((g + 1) & 1, bx);            // stage g + 1 -> the buffer the consumers are NOT reading during stage g
      load_b(gl, bx); gl = (gl + 1 == G) ? 0 : gl + 1;
      __syncthreads();                     // stage barrier
      if (++sc == NS) { sc = 0; if (g + 1 < total) __syncthreads(); }   // unit barrier
    }
  } else {
    // ================================================================= patch loaders (each half the pieces)
    const int l2 = lane + (wave - 6) * 64;
    // Buffer loads: one descriptor per sample (uniform), a 32-bit byte offset per piece, and an out-of-range offset
    // for the zero padding -- the hardware returns 0, so no validity masks or 64-bit addresses stay live next to the
    // 132 registers the patch itself needs.
    const int sample_bytes = (int)((((long)p.Hi * p.Wi - 1) * p.ldx + p.Cin) * 2);
    auto load_patch = [&](int t, int cc, u32x4(&pa)[NPA]) {
      int b, h0, w0;
      decode(t, b, h0, w0);
      const int vh0 = h0 - 1, vw0 = w0 - 1;
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (long)b * p.Hi * p.Wi * p.ldx), 0, sample_bytes, 0x00020000);
      const int ldb = (int)p.ldx * 2;
      // piece i of this lane: pixel (lane >> 3) + 8 i of the patch (row-major, PW wide), 16-byte piece lane & 7.  The
      // walk is incremental and fenced per piece (sched_barrier), so that one offset is live at a time.
      const int chb = (cc * 64 + (lane & 7) * 8) * 2;
      int pr = 0, pc = l2 >> 3;
      asm volatile("" : "+v"(pc));
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        bool valid = (A_PIECES % 128 == 0) || (pr < PH);
        const int vh = vh0 + pr, vw = vw0 + pc;
        int sh = vh, sw = vw;
        if constexpr (MODE == MODE_UP) {
          valid = valid && (vh >= 0) && (vh < 2 * p.Hi) && (vw >= 0) && (vw < 2 * p.Wi);
          sh = vh >> 1; sw = vw >> 1;
        } else if constexpr (MODE == MODE_ZI) {
          valid = valid && (vh >= 0) && (vh < 2 * p.Hi) && (vw >= 0) && (vw < 2 * p.Wi) && (((vh | vw) & 1) == 0);
          sh = vh >> 1; sw = vw >> 1;
        } else {
          valid = valid && (vh >= 0) && (vh < p.Hi) && (vw >= 0) && (vw < p.Wi);
        }
        const int off = valid ? (sh * p.Wi + sw) * ldb + chb : 0x7ffffff0;
        pa[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
        pc += 16;
        if (pc >= PW) { pc -= PW; ++pr; }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto store_patch = [&](u32x4(&pa)[NPA]) {
      int pr = 0, pc = l2 >> 3;
      asm volatile("" : "+v"(pc));
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        if ((A_PIECES % 128 == 0) || (pr < PH))
          *reinterpret_cast<u32x4*>(sA + (pr * PITCH + pc) * STRIDE + (lane & 7) * 16) = pa[i];
        pc += 16;
        if (pc >= PW) { pc -= PW; ++pr; }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    u32x4 pa[NPA];
    load_patch(wk.t_cur, 0, pa);
    store_patch(pa);
    __syncthreads();
    int tn, cn;
    bool has = wk.has_next();
    if (has) { wk.next(tn, cn); load_patch(tn, cn, pa); }
    for (;;) {
      for (int s = 0; s < NS; ++s) __syncthreads();   // the unit's stage barriers
      if (!has) break;
      store_patch(pa);     // every consumer is past the unit's last stage barrier: the old patch is dead
      __syncthreads();     // unit barrier
      wk.advance();
      has = wk.has_next();
      if (has) { wk.next(tn, cn); load_patch(tn, cn, pa); }
    }
  }
}

template <int NT, int MODE, int TW = 24>
int launch_conv_ws(const ConvParams& p0, hipStream_t st) {
  constexpr int LDS = 10 * ((TW == 24) ? 32 : TW + 2) * 160 + 2 * 2 * NT * 1024;
  static_assert(LDS <= 80 * 1024, "two workgroups per CU");
  auto k = conv_ws_kernel<NT, MODE, TW>;
  static unsigned long long attr_devs = 0ull;
  if (dm_first_on_device(attr_devs)) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return DM_ERR_LAUNCH;
  }
  ConvParams p = p0;
  const long ntiles = (long)p.B * p.tiles_h * p.tiles_w;
  if (ntiles <= 0 || ntiles > 0x7fffffffL) return DM_ERR_ARG;
  p.ntiles = (int)ntiles;
  static int resident = 0;   // co-resident workgroups: two per CU (12 of the 16 wave slots; VGPRs allow no third)
  if (!resident) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    resident = 2 * cus;
    if (const char* f = getenv("DM_CONV_WS_RESIDENT")) resident = atoi(f);   // developer override
  }
  const int gy = p.Cout / (16 * NT);
  int gx = resident / gy;
  if (gx < 1) gx = 1;
  if (gx > p.ntiles) gx = p.ntiles;
  const int rounds = (p.ntiles + gx - 1) / gx;
  gx = (p.ntiles + rounds - 1) / rounds;   // same number of rounds, evenly filled
  static const DmVariantName vn("conv_ws<TH=8,TW=%d,NT=%d,MODE=%d>", TW, NT, MODE);
  dm_set_variant(vn.s);
  hipLaunchKernelGGL(k, dim3((unsigned)gx, (unsigned)gy), dim3(512), LDS, st, p);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// ------------------------------------------------------------------ streaming GEMM (1x1 conv / linear)
// Y[M, N] = X[M, K] W^T with K <= 512: HBM-bound (X read once, Y written once), so the kernel is built
// around the memory stream, not the MFMA pipe:
//   * the workgroup's weight slice (all NKS k-steps of its 16*NT columns) is copied to LDS once and is
//     read-only afterwards -> no barrier in the main loop, the four waves run decoupled;
//   * every wave walks its own 16*MT-row tiles (global wave id, + total waves, ...) and loads the X
//     fragments straight from global memory in MFMA operand layout (16 rows x 64 B per load; the two
//     halves of a 128-byte line are fetched by consecutive loads), one whole tile ahead of the MFMAs;
//   * the epilogue (bias / time-embedding / residual / fp32 out) is the conv kernel's.
template <int MT, int NT, int NKS, bool WIDE>
__global__ __launch_bounds__(256, (WIDE && MT == 1 && NKS <= 4) ? 3 : 2) void gemm_stream_kernel(const ConvParams p, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char sW[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = blockIdx.y;
  const int NTG = p.Cout >> 4;
  const int nwaves = gridDim.x * 4;
  const int nks_real = ((p.Cin + 63) >> 6) * 2;
  const int row_l = lane & 15, kq = (lane >> 4) * 8;

  // X fragments come through a buffer descriptor over the tile's rows: rows past M and channels past Cin read as 0
  // by the hardware range check (an out-of-range offset), with no select on the loaded value -- a `valid ? v : 0` makes
  // the compiler wait for the prefetch where it is issued, not where it is used.
  auto load_x = [&](int tile, bf16x8(&xa)[MT][NKS]) {
    const long row0 = (long)__builtin_amdgcn_readfirstlane(tile) * (MT * 16);
    long rows = p.M - row0;
    rows = rows < 0 ? 0 : (rows > MT * 16 ? MT * 16 : rows);
    const int nrec = rows > 0 ? (int)(((rows - 1) * p.ldx + p.Cin) * 2) : 0;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (rows > 0 ? row0 : 0) * p.ldx), 0, nrec, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int ch = ks * 32 + kq;
        const int off = (ch < p.Cin) ? ((mt * 16 + row_l) * (int)p.ldx + ch) * 2 : 0x7ffffff0;
        xa[mt][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
      }
    }
  };

  bf16x8 xa0[MT][NKS], xa1[MT][NKS];
  int tile = blockIdx.x * 4 + wave;
  if (tile < ntiles) load_x(tile, xa0);
  // weight slice: k-step ks holds NT KiB at ((ks * NTG + nb * NT) * 64) * 16
  {
    constexpr int WP = NKS * NT * 64 / 256;  // 16-byte pieces per thread
    constexpr int WB = WP < 8 ? WP : 8;      // loads kept in flight together
#pragma unroll 1
    for (int i0 = 0; i0 < WP; i0 += WB) {
      u32x4 wv[WB];
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        const int i = tid + (i0 + u) * 256;
        const int ks = i / (NT * 64), rem = i - ks * (NT * 64);
        // k-steps past the packed K (NKS is rounded up to 2/4/8/16) must read as 0, not as whatever follows
        const bool ok = ks < nks_real;
        const u32x4 v = *reinterpret_cast<const u32x4*>(p.w + (ok ? (((long)ks * NTG + nb * NT) * 64 + rem) * 16 : 0));
        wv[u] = ok ? v : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < WB; ++u) *reinterpret_cast<u32x4*>(sW + (long)(tid + (i0 + u) * 256) * 16) = wv[u];
    }
  }
  __syncthreads();

  auto work = [&](int t, bf16x8(&xa)[MT][NKS], bf16x8(&xn)[MT][NKS]) {
    asm volatile("" ::: "memory");  // keep the epilogue's per-column loads (bias) inside the tile loop: hoisted, they cost 32+ VGPRs
    load_x(t + nwaves, xn);   // unconditional: past the last tile the descriptor is empty (zeros, no traffic)
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      bf16x8 bv[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bv[nt] = *reinterpret_cast<const bf16x8*>(sW + ((ks * NT + nt) * 64 + lane) * 16);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        bf16x8 av = xa[mt][ks];
        if (p.act_silu_in) {
#pragma unroll
          for (int e = 0; e < 8; ++e) av[e] = (bf16)dm_silu((float)av[e]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[nt], av, acc[mt][nt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the weight fragments of later k-steps out of registers
    }
    conv_epilogue<MT, NT, 16, MODE_GEMM, WIDE>(p, acc, nb, 0, lane, 0, 0, 0, (long)t * MT * 16);
  };
  for (; tile + nwaves < ntiles; tile += 2 * nwaves) {   // tile pairs, an odd last tile peeled off
    work(tile, xa0, xa1);
    work(tile + nwaves, xa1, xa0);
  }
  if (tile < ntiles) work(tile, xa0, xa1);
}

template <int MT, int NT, int NKS, bool WIDE>
int launch_gemm_stream_v(const ConvParams& p, hipStream_t st) {
  constexpr int LDS = NKS * NT * 1024;
  static_assert(LDS <= 64 * 1024, "two workgroups per CU");
  auto k = gemm_stream_kernel<MT, NT, NKS, WIDE>;
  static unsigned long long attr_devs = 0ull;
  if (dm_first_on_device(attr_devs)) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return DM_ERR_LAUNCH;
  }
  const int ntiles = (int)((p.M + 16 * MT - 1) / (16 * MT));  // wave tiles
  const int gy = p.Cout / (16 * NT);
  // Workgroups in total: four per CU when every wave still gets four or more row tiles (M >= 256k rows), else two per
  // CU -- each workgroup stages its whole weight slice first, and with 2.25 tiles per wave (M = 147k at 1024
  // workgroups) a quarter of the waves run one tile longer than the rest.  tools/convbench.py, 1024 -> 512 workgroups:
  // qkv 128->384 M=147k 36.7 -> 31.2 us, 256->128 M=147k 29.7 -> 22.7, 64->128 M=147k 15.9 -> 12.8, 256->768 M=37k
  // 33.5 -> 28.2; 768 / 640 / 384 / 256 in between or worse; M = 590k: 36.4 -> 37.0 (stays at 1024).
  static const int gxcap = [] { const char* e = getenv("DM_GEMM_GX"); return e ? atoi(e) : 0; }();   // developer override
  int gx = (gxcap > 0 ? gxcap : (p.M >= 262144 ? 1024 : 512)) / gy;
  if (gx < 1) gx = 1;
  if (gx > (ntiles + 3) / 4) gx = (ntiles + 3) / 4;
  // Workgroups are dealt to the 8 XCDs round-robin by linear id (x + y * gx): with gx a multiple of 8 the gy
  // workgroups that stream the same X rows sit on one XCD and share them through its L2.
  if (gy > 1 && gx > 8) gx &= ~7;
  static const DmVariantName vn("gemm_stream<MT=%d,NT=%d,NKS=%d,WIDE=%d>", MT, NT, NKS, (int)WIDE);
  dm_set_variant(vn.s);
  hipLaunchKernelGGL(k, dim3(gx, gy), dim3(256), LDS, st, p, ntiles);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// The variant whose epilogue holds only the 16-byte-store path (bf16 output, whole channel tiles, aligned rows) needs
// ~30 fewer VGPRs: a third resident wave per SIMD for this HBM-bound kernel.  Same conditions as conv_epilogue's.
template <int MT, int NT, int NKS>
int launch_gemm_stream(const ConvParams& p, hipStream_t st) {
  const bool wide = (NT % 2 == 0) && p.y && !p.yf && !p.pool2 && p.n_valid == p.Cout && (p.ldy & 7) == 0 &&
                    (reinterpret_cast<unsigned long long>(p.y) & 15) == 0;
  return wide ? launch_gemm_stream_v<MT, NT, NKS, true>(p, st) : launch_gemm_stream_v<MT, NT, NKS, false>(p, st);
}

template <int NT>
int dispatch_gemm_stream(const ConvParams& p, hipStream_t st) {
  const int nks = ((p.Cin + 63) / 64) * 2;
  const bool big = p.M >= 262144;  // enough rows for 32-row wave tiles to balance over the CUs
  // 32-row wave tiles only where both X tiles in flight plus the accumulators fit the register file
  if (nks <= 2) return big ? launch_gemm_stream<2, NT, 2>(p, st) : launch_gemm_stream<1, NT, 2>(p, st);
  if (nks <= 4) {
    if constexpr (NT <= 4) { if (big) return launch_gemm_stream<2, NT, 4>(p, st); }
    return launch_gemm_stream<1, NT, 4>(p, st);
  }
  if (nks <= 8) return launch_gemm_stream<1, NT, 8>(p, st);
  if constexpr (NT <= 4) {
    if (nks <= 16) return launch_gemm_stream<1, NT, 16>(p, st);
  }
  return 1;  // not handled here
}

// ------------------------------------------------------------------ weight packing
// dst element (kstep, ntg, lane, j):  n = ntg*16 + (lane&15), kk = kstep*32 + 8*(lane>>4) + j,
// kstep = (cc*TAPS + tap)*2 + ks, channel k = cc*64 + ks*32 + 8*(lane>>4) + j.
//   transpose = 0: value = W[n][k][kh][kw]          (forward; W is [N][K][kh][kw] fp32)
//   transpose = 1: value = W[k][n][2-kh][2-kw]      (data-gradient; W is [K][N][kh][kw])
// Out-of-range n (>= N) or k (>= K) pack as zero (padding to 16 / 64).
__device__ __forceinline__ void pack_weight_range(const float* __restrict__ w, bf16* __restrict__ dst, int N, int K,
                                                  int taps, int transpose, long first, long stride) {
  const int Npad = (N + 15) / 16 * 16, Kpad = (K + 63) / 64 * 64;
  const long total = (long)Npad * Kpad * taps;
  const int NTG = Npad >> 4;
  for (long i = first; i < total; i += stride) {
    const int j = (int)(i & 7);
    const int ln = (int)((i >> 3) & 63);
    long r = i >> 9;
    const int ntg = (int)(r % NTG); r /= NTG;
    const int ks = (int)(r & 1); r >>= 1;
    const int tap = (int)(r % taps);
    const int cc = (int)(r / taps);
    const int n = ntg * 16 + (ln & 15);
    const int k = cc * 64 + ks * 32 + 8 * (ln >> 4) + j;
    float v = 0.f;
    if (n < N && k < K) {
      const int kh = tap / 3, kw = tap % 3;
      if (!transpose) v = w[((long)n * K + k) * taps + tap];
      else v = (taps == 9) ? w[((long)k * N + n) * 9 + (2 - kh) * 3 + (2 - kw)] : w[(long)k * N + n];
    }
    dst[i] = (bf16)v;
  }
}

__global__ void pack_weight_kernel(const float* __restrict__ w, bf16* __restrict__ dst, int N, int K,
                                   int taps, int transpose) {
  pack_weight_range(w, dst, N, K, taps, transpose, (long)blockIdx.x * blockDim.x + threadIdx.x,
                    (long)gridDim.x * blockDim.x);
}

// All shadows of a model in ONE launch (after every optimizer step): block -> table entry by binary search
// over block_begin.
__global__ void pack_weight_batch_kernel(const dm_pack_desc* __restrict__ d, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (d[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const dm_pack_desc e = d[lo];
  pack_weight_range(e.w, (bf16*)e.dst, e.N, e.K, e.taps, e.transpose,
                    (long)(blockIdx.x - e.block_begin) * blockDim.x + threadIdx.x, (long)e.nblocks * blockDim.x);
}

// Tile choice for the stride-1 family (modes S1 / UP / ZI).  tile: 0 = 16x16, 1 = 8x24, 2 = 8x8; nt: 4 or 8.
// Rule 1 (round 1): the tile with the least padded area (edge tiles compute their padding), ties to the larger tile
// (fewer halo bytes per pixel); 128 couts when that grid still has two rounds of the 512 resident workgroups.
// Rule 2 (tools/convtile.py sweep, B=64, profiles/r02_convtile.txt): when 8x24 tiles cover the image as exactly as
// 16x16 tiles do and the 16x16 grid is under three rounds, the smaller 8x24 workgroups balance better over the CUs
// (48x48 outputs: 3.0 rounds of 192-pixel tiles instead of 2.25 of 256-pixel ones) and 128 couts per workgroup pay
// again: L2.conv2 49.8 -> 46.6 us, L4.conv1 87.0 -> 81.2, zero-insert dgrad of L2.down 49.5 -> 43.9, dgrad of L2.conv1
// 33.3 -> 29.3.  At 96x96 (4.5+ rounds) 16x16 stays faster (58.6 vs 62.5 us), at 24x24 8x24/64 couts (rule 1).
void pick_tile(int B, int Ho, int Wo, int Cin, int Cout, bool pool2, int* tile, int* nt) {
  auto up = [](int v, int t) { return (v + t - 1) / t; };
  static const int TH_[3] = {16, 8, 8}, TW_[3] = {16, 24, 8};
  const bool can128 = (Cout % 128 == 0);
  const long a16 = (long)up(Ho, 16) * 16 * up(Wo, 16) * 16, a24 = (long)up(Ho, 8) * 8 * up(Wo, 24) * 24,
             a8 = (long)up(Ho, 8) * 8 * up(Wo, 8) * 8;
  int t = 2;
  if (pool2 || (a16 <= a24 && a16 <= a8)) t = 0;
  else if (a24 <= a8) t = 1;
  long tiles = (long)B * up(Ho, TH_[t]) * up(Wo, TW_[t]);
  int n = (can128 && tiles * (Cout / 128) >= 1024) ? 8 : 4;
  if (t == 0 && !pool2 && a24 == a16 && tiles * (Cout / (16 * n)) < 1536) {
    t = 1;
    tiles = (long)B * up(Ho, 8) * up(Wo, 24);
    n = (can128 && tiles * (Cout / 128) >= 512) ? 8 : 4;
  }
  *tile = t;
  *nt = n;
  (void)Cin;
}

}  // namespace

extern "C" long dm_packed_weight_bytes(int N, int K, int taps) {
  const long Npad = (N + 15) / 16 * 16, Kpad = (K + 63) / 64 * 64;
  return Npad * Kpad * taps * 2;
}

extern "C" int dm_pack_weight(const float* w, void* dst, int N, int K, int taps, int transpose,
                              dm_stream_t stream) {
  DM_REQUIRE(w && dst && N > 0 && K > 0 && (taps == 1 || taps == 9));
  const int Npad = (N + 15) / 16 * 16, Kpad = (K + 63) / 64 * 64;
  const long total = (long)Npad * Kpad * taps;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (bf16*)dst, N, K,
                     taps, transpose);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_pack_weights_batch(const dm_pack_desc* table_dev, int n, int total_blocks, dm_stream_t stream) {
  DM_REQUIRE(table_dev && n > 0 && total_blocks > 0);
  hipLaunchKernelGGL(pack_weight_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// mode: 0 = 3x3 s1 p1, 1 = 3x3 s2 p1, 2 = nearest-x2 then 3x3 s1 p1, 3 = zero-insert-x2 then
// 3x3 s1 p1 (data-gradient of mode 1 with transposed+flipped weights).
extern "C" int dm_conv3x3(const dm_conv_args* a, dm_stream_t stream) {
  DM_REQUIRE(a && a->x && a->w && (a->y || a->yf));
  DM_REQUIRE(a->Cin % 64 == 0 && a->Cout % 64 == 0 && a->B > 0);
  DM_REQUIRE(a->ldx % 8 == 0 && a->ldx >= a->Cin);
  DM_REQUIRE(!a->y || (a->ldy % 4 == 0));
  DM_REQUIRE(!a->res || (a->ldr % 4 == 0));
  DM_REQUIRE(a->mode >= 0 && a->mode <= 3);
  ConvParams p{};
  p.x = (const bf16*)a->x; p.ldx = a->ldx; p.w = (const char*)a->w; p.bias = a->bias;
  p.chan_add = a->chan_add; p.res = (const bf16*)a->res; p.ldr = a->ldr;
  p.y = (bf16*)a->y; p.ldy = a->ldy; p.yf = a->yf; p.ldyf = a->ldyf;
  p.B = a->B; p.Hi = a->H; p.Wi = a->W; p.Cin = a->Cin; p.Cout = a->Cout;
  p.n_valid = a->n_valid > 0 ? a->n_valid : a->Cout;
  p.pool2 = a->pool2; p.act_silu_in = a->act_silu_in;
  DM_REQUIRE((a->norm_scale == nullptr) == (a->norm_shift == nullptr));
  DM_REQUIRE(!a->norm_scale || (a->mode == 0 && !a->act_silu_in && !a->pool2));  // GroupNorm feeds plain 3x3 convs only
  p.nsc = a->norm_scale; p.nsh = a->norm_shift;
  p.drop_seed = dm_seed32(a->drop_seed); p.drop_scale = 1.0f; p.seed_ctr = dm_dropout_counter();
  p.drop_thresh = 0;
  if (a->drop_p > 0.f) {
    DM_REQUIRE(a->drop_p < 1.f);
    p.drop_thresh = dm_thresh16(a->drop_p);
    p.drop_scale = dm_inv_keep(a->drop_p);
  }
  hipStream_t st = (hipStream_t)stream;
  const int mode = a->mode;
  if (mode == 1) { p.Ho = a->H / 2; p.Wo = a->W / 2; DM_REQUIRE(a->H % 2 == 0 && a->W % 2 == 0); }
  else if (mode == 2 || mode == 3) { p.Ho = a->H * 2; p.Wo = a->W * 2; }
  else { p.Ho = a->H; p.Wo = a->W; }
  DM_REQUIRE(!p.pool2 || (p.Ho % 2 == 0 && p.Wo % 2 == 0 && p.y && !p.yf && !p.res && !p.chan_add && !p.drop_thresh));
  if (mode == 0 && a->H == 8 && a->W == 8 && a->B >= 4 && !p.pool2 && !p.nsc) {
    // 8x8 images (the latent model's innermost stage: 4096 pixels against K*N up to 9216 x 1024): four images per
    // workgroup; 32-cout workgroups when the 64-cout grid would leave CUs idle
    const int quads = (a->B + 3) / 4;
    if ((long)quads * (a->Cout / 64) >= 256) return launch_conv<16, 16, 1, 20, 160, 4, MODE_Q8>(p, st);
    return launch_conv<16, 16, 1, 20, 160, 2, MODE_Q8>(p, st);
  }
  // ---- tile shape and cout tile.  Candidates: 16x16, 8x24, 8x8 output pixels x {64, 128} couts (stride 2: 8x8
  // only; pool2 needs whole row pairs per wave: 16x16); pick_tile() holds the rule and the measurements behind it.
  int tile = 2, nt = 4;   // stride 2: 8x8 outputs
  const bool can128 = (a->Cout % 128 == 0);
  if (mode == 1) {
    p.tiles_h = (p.Ho + 7) / 8; p.tiles_w = (p.Wo + 7) / 8;
    nt = (can128 && (long)p.B * p.tiles_h * p.tiles_w * (a->Cout / 128) >= 1024) ? 8 : 4;
  } else {
    pick_tile(p.B, p.Ho, p.Wo, a->Cin, a->Cout, p.pool2 != 0, &tile, &nt);
  }
  if (const char* f = getenv("DM_CONV_FORCE")) {   // developer override "tile,nt" (tools/convtile.py); -1 = keep
    int ft = -1, fn = -1;
    if (sscanf(f, "%d,%d", &ft, &fn) >= 1) {
      if (mode != 1 && !p.pool2 && ft >= 0 && ft <= 2) tile = ft;
      if ((fn == 4 || (fn == 8 && can128))) nt = fn;
    }
  }
  // 64-cout tiles go to the wave-specialised persistent kernel when 8x24 tiles cover the image exactly (measured,
  // tools/convbench.py, B=64: 64->64 @96x96 60.6 -> 55.8 us, 64->128 @48x48 34.2 -> 31.4 us, zero-insert dgrad 64->64
  // 59.8 -> 54.9 us; with 128-cout tiles the 256-VGPR conv_mfma_kernel stays ahead: its accumulators do not fit the
  // 128 registers that two eight-wave workgroups per CU leave each wave)
  // ... and when every persistent workgroup gets at least two tiles to pipeline (a smaller launch is all prologue)
  if (mode != 1 && nt == 4 && !p.pool2 && !p.nsc && !p.act_silu_in && p.Ho % 8 == 0 && p.Wo % 24 == 0 &&
      (long)p.B * (p.Ho / 8) * (p.Wo / 24) * (a->Cout / 64) >= 1024) {
    const char* f = getenv("DM_CONV_WS");   // developer switch: "0" = off
    if (!f || f[0] != '0') {
      p.tiles_h = p.Ho / 8; p.tiles_w = p.Wo / 24;
      if (mode == 0) return launch_conv_ws<4, MODE_S1>(p, st);
      if (mode == 2) return launch_conv_ws<4, MODE_UP>(p, st);
      return launch_conv_ws<4, MODE_ZI>(p, st);
    }
  }
  static const int TH_[3] = {16, 8, 8}, TW_[3] = {16, 24, 8};
  p.tiles_h = (p.Ho + TH_[tile] - 1) / TH_[tile]; p.tiles_w = (p.Wo + TW_[tile] - 1) / TW_[tile];
#define DM_GO(TH, TW, S, PITCH, STRIDE, MODE)                                          \
  do {                                                                                 \
    if (nt == 8) return launch_conv<TH, TW, S, PITCH, STRIDE, 8, MODE>(p, st);         \
    return launch_conv<TH, TW, S, PITCH, STRIDE, 4, MODE>(p, st);                      \
  } while (0)
  if (mode == 1) DM_GO(8, 8, 2, 24, 144, MODE_S2);
#define DM_GON(TH, TW, S, PITCH, STRIDE)                                                      \
  do {                                                                                      \
    if (nt == 8) return launch_conv<TH, TW, S, PITCH, STRIDE, 8, MODE_S1, true>(p, st);     \
    return launch_conv<TH, TW, S, PITCH, STRIDE, 4, MODE_S1, true>(p, st);                  \
  } while (0)
  if (p.nsc) {
    // (the 16x16 x 128-cout tile has no registers left for the scale / shift rows: 64 couts there)
    if (tile == 0) return launch_conv<16, 16, 1, 18, 160, 4, MODE_S1, true>(p, st);
    if (tile == 1) DM_GON(8, 24, 1, 32, 160);
    DM_GON(8, 8, 1, 10, 160);
  }
#undef DM_GON
  if (tile == 0) {
    if (mode == 0) DM_GO(16, 16, 1, 18, 160, MODE_S1);
    if (mode == 2) DM_GO(16, 16, 1, 18, 160, MODE_UP);
    DM_GO(16, 16, 1, 18, 160, MODE_ZI);
  }
  if (tile == 1) {
    if (mode == 0) DM_GO(8, 24, 1, 32, 160, MODE_S1);
    if (mode == 2) DM_GO(8, 24, 1, 32, 160, MODE_UP);
    DM_GO(8, 24, 1, 32, 160, MODE_ZI);
  }
  if (mode == 0) DM_GO(8, 8, 1, 10, 160, MODE_S1);
  if (mode == 2) DM_GO(8, 8, 1, 10, 160, MODE_UP);
  DM_GO(8, 8, 1, 10, 160, MODE_ZI);
#undef DM_GO
  return DM_ERR_ARG;
}

// Y[M, N] = X[M, K] * W[N, K]^T (+bias)(+chan_add[row / rows_per_batch])(+res); bf16 in, fp32 accumulate.
extern "C" int dm_gemm(const dm_gemm_args* a, dm_stream_t stream) {
  DM_REQUIRE(a && a->x && a->w && (a->y || a->yf) && a->M > 0);
  DM_REQUIRE(a->K % 8 == 0 && a->N % 16 == 0 && a->ldx % 8 == 0);
  DM_REQUIRE(!a->y || (a->ldy % 4 == 0));
  DM_REQUIRE(!a->res || (a->ldr % 4 == 0));
  ConvParams p{};
  p.x = (const bf16*)a->x; p.ldx = a->ldx; p.w = (const char*)a->w; p.bias = a->bias;
  p.chan_add = a->chan_add; p.res = (const bf16*)a->res; p.ldr = a->ldr;
  p.y = (bf16*)a->y; p.ldy = a->ldy; p.yf = a->yf; p.ldyf = a->ldyf;
  p.M = a->M; p.Cin = a->K; p.Cout = a->N; p.B = 1;
  p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : 1 << 30;
  p.n_valid = a->n_valid > 0 ? a->n_valid : a->N;
  p.act_silu_in = a->act_silu_in;
  p.drop_thresh = 0; p.drop_scale = 1.f;
  hipStream_t st = (hipStream_t)stream;
  {
    // streaming kernel when the workgroup's weight slice fits in 64 KiB of LDS ...
    const int nks = ((a->K + 63) / 64) * 2;
    int rc = 1;
    // ... and the shape is HBM-bound: from K = 512 up, or K = 256 with more than one 128-column block and many rows, the barrier-
    // pipelined implicit-GEMM kernel below (weights re-staged per k-step, 64-row wave tiles) is faster
    // (measured: 16384x512x512 21 vs 27 us, 65536x512x256 28 vs 47 us, 16384x512x1536 43 vs 69 us, 65536x256x256 20 vs 25 us)
    if ((nks >= 16 && a->M >= 4096) || (nks == 8 && a->N > 128 && a->M >= 65536)) rc = 1;
    else
    if (a->N % 128 == 0 && nks <= 8 && a->M > 4096) rc = dispatch_gemm_stream<8>(p, st);
    else if (a->N % 64 == 0 && nks <= 16) rc = dispatch_gemm_stream<4>(p, st);
    if (rc <= 0) return rc;
  }
  if (a->N % 128 == 0) return launch_conv<16, 16, 1, 16, 160, 8, MODE_GEMM>(p, st);
  if (a->N % 64 == 0) return launch_conv<16, 16, 1, 16, 160, 4, MODE_GEMM>(p, st);
  if (a->N % 32 == 0) return launch_conv<16, 16, 1, 16, 160, 2, MODE_GEMM>(p, st);
  return launch_conv<16, 16, 1, 16, 160, 1, MODE_GEMM>(p, st);
}
