// Fused GroupNorm + SiLU (+ optional per-(b,c) time-embedding add) forward / backward on NHWC
// bf16 activations.  HBM-bound: every access is a 16-byte (8 x bf16) vector; statistics are
// reduced with per-thread fp32 partials -> Chan/Welford merges across the workgroup through
// LDS -> a small per-(sample, split, group) partial table that the apply pass re-merges (no
// atomics: bit-reproducible).
//
// Reference semantics replaced: nn.GroupNorm(8, C) (biased variance, eps 1e-5) followed by
// nn.SiLU, and the `h + time_emb_proj` add in front of norm2 (unet.py:77-78,94,101-102,106),
// plus their autograd backward.
#include "dm_common.h"

namespace {

constexpr int PART_LDS = 512;  // (split, group) partials staged in LDS by the apply kernels
constexpr int MAX_C = 1024;    // widest GroupNorm (latent-model widths, utils.py:145-156); 256 threads x 8 channels x rows
struct Wf { float n, mean, m2; };
__device__ __forceinline__ bf16x8 zero8_norm() {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16)0.f;
  return r;
}
__device__ __forceinline__ Wf wf_merge(Wf a, Wf b) {
  const float n = a.n + b.n;
  if (n == 0.f) return a;
  const float d = b.mean - a.mean;
  const float fb = b.n / n;
  Wf r;
  r.n = n;
  r.mean = a.mean + d * fb;
  r.m2 = a.m2 + b.m2 + d * d * a.n * fb;
  return r;
}

// x' = x + chan_add[b][c] (optional) ; statistics of x' per (b, group)
// grid (nsplit, B). partial[b][split][g] = {n, mean, M2}
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16* __restrict__ x, long ldx,
                                                        const float* __restrict__ chan_add,
                                                        float* __restrict__ partial, int HW, int C, int G,
                                                        int nsplit) {
  __shared__ Wf red[256];
  const int b = blockIdx.y, split = blockIdx.x, tid = threadIdx.x;
  const int tpp = C >> 3;            // threads per pixel
  const int ppi = 256 / tpp;         // pixels per iteration
  const int cpg = C / G;             // channels per group
  const int tpg = cpg >> 3;          // threads per group (per pixel)
  const int c0 = (tid % tpp) * 8;
  const int prow = tid / tpp;
  const int per = (HW + nsplit - 1) / nsplit;
  const int p_begin = split * per, p_end = min(HW, p_begin + per);
  float add[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) add[j] = chan_add ? chan_add[(long)b * C + c0 + j] : 0.f;
  float s = 0.f, ss = 0.f, cnt = 0.f;
  const bf16* xb = x + (long)b * HW * ldx + c0;
  int pix = p_begin + prow;
  // four independent 16-byte loads in flight per thread (the blocks are short: latency, not issue, bounds them)
  for (; pix + 3 * ppi < p_end; pix += 4 * ppi) {
    bf16x8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(xb + (long)(pix + u * ppi) * ldx);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[u][j] + add[j];
        s += f; ss += f * f;
      }
    cnt += 32.f;
  }
  for (; pix < p_end; pix += ppi) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(xb + (long)pix * ldx);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)v[j] + add[j];
      s += f; ss += f * f;
    }
    cnt += 8.f;
  }
  Wf w;
  w.n = cnt;
  w.mean = cnt > 0.f ? s / cnt : 0.f;
  w.m2 = cnt > 0.f ? fmaxf(ss - s * w.mean, 0.f) : 0.f;
  red[tid] = w;
  __syncthreads();
  if (tid < G) {
    Wf acc = {0.f, 0.f, 0.f};
    for (int pr = 0; pr < ppi; ++pr)
      for (int k = 0; k < tpg; ++k) acc = wf_merge(acc, red[pr * tpp + tid * tpg + k]);
    float* o = partial + (((long)b * nsplit + split) * G + tid) * 3;
    o[0] = acc.n; o[1] = acc.mean; o[2] = acc.m2;
  }
}

// grid (nchunk, B): y = silu((x' - mean) * rstd * gamma + beta); chunk 0 also publishes stats[b][g] = {mean, rstd}
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16* __restrict__ x, long ldx,
                                                        const float* __restrict__ chan_add,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ partial, int nsplit,
                                                        float* __restrict__ stats, bf16* __restrict__ y, long ldy,
                                                        int HW, int C, int G, float eps, int nchunk, int act) {
  __shared__ float s_mean[64], s_rstd[64];
  __shared__ float s_part[PART_LDS * 3];
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  // the sample's partial table comes in with one parallel load per thread, not nsplit dependent ones per group
  const bool staged = nsplit * G <= PART_LDS;
  if (staged) {
    for (int i = tid; i < nsplit * G * 3; i += 256) s_part[i] = partial[(long)b * nsplit * G * 3 + i];
    __syncthreads();
  }
  if (tid < G) {
    Wf acc = {0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s) {
      const float* o = staged ? s_part + (s * G + tid) * 3 : partial + (((long)b * nsplit + s) * G + tid) * 3;
      Wf w = {o[0], o[1], o[2]};
      acc = wf_merge(acc, w);
    }
    const float var = acc.m2 / acc.n;
    const float rstd = rsqrtf(var + eps);
    s_mean[tid] = acc.mean; s_rstd[tid] = rstd;
    if (chunk == 0 && stats) { stats[((long)b * G + tid) * 2] = acc.mean; stats[((long)b * G + tid) * 2 + 1] = rstd; }
  }
  __syncthreads();
  const int tpp = C >> 3, ppi = 256 / tpp, cpg = C / G;
  const int c0 = (tid % tpp) * 8, prow = tid / tpp;
  const int g = c0 / cpg;
  const float mean = s_mean[g], rstd = s_rstd[g];
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = gamma[c0 + j] * rstd;
    const float a = chan_add ? chan_add[(long)b * C + c0 + j] : 0.f;
    sh[j] = beta[c0 + j] + (a - mean) * sc[j];
  }
  const int per = (HW + nchunk - 1) / nchunk;
  const int p_begin = chunk * per, p_end = min(HW, p_begin + per);
  const bf16* xb = x + (long)b * HW * ldx + c0;
  bf16* yb = y + (long)b * HW * ldy + c0;
  auto one = [&](bf16x8 v, int pix) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float z = (float)v[j] * sc[j] + sh[j];
      o[j] = (bf16)(act ? dm_silu(z) : z);
    }
    *reinterpret_cast<bf16x8*>(yb + (long)pix * ldy) = o;
  };
  int pix = p_begin + prow;
  for (; pix + 3 * ppi < p_end; pix += 4 * ppi) {
    bf16x8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(xb + (long)(pix + u * ppi) * ldx);
#pragma unroll
    for (int u = 0; u < 4; ++u) one(v[u], pix + u * ppi);
  }
  for (; pix < p_end; pix += ppi) one(*reinterpret_cast<const bf16x8*>(xb + (long)pix * ldx), pix);
}

// ---------------------------------------------------------------- one pass: the (sample, group) slab lives in registers
// GroupNorm(8, C) on the UNet's tensors: one (sample, group) slab is at most 8..32 channels x 9216 pixels x 2 B = 144 KiB,
// which a 1024-thread workgroup holds in registers (9 sixteen-byte pieces per thread).  One workgroup per (sample, group)
// reads its slab ONCE, reduces the statistics through LDS, and writes silu(affine(x)) from the registers: 4 B/elem of
// HBM traffic instead of the 6 B/elem of gn_stats_kernel + gn_apply_kernel.  A slab is a 16..64-byte piece of every
// pixel row (NHWC), i.e. a strided stream that uses only part of each 128-byte line; the other groups of the same sample
// use the rest, so the eight workgroups of a sample are given consecutive slots on ONE XCD (workgroup id -> XCD is
// id mod 8) and meet in that XCD's L2.
constexpr int OP_THREADS = 1024, OP_MAXP = 9;
__global__ __launch_bounds__(OP_THREADS) void gn_onepass_fwd_kernel(const bf16* __restrict__ x, long ldx,
                                                                     const float* __restrict__ chan_add,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta,
                                                                     float* __restrict__ stats, bf16* __restrict__ y, long ldy,
                                                                     int B, int HW, int C, float eps, int act) {
  __shared__ Wf red[OP_THREADS / 64];
  __shared__ float s_ms[2];
  const int id = blockIdx.x, xcd = id & 7, jj = id >> 3, g = jj & 7, b = (jj >> 3) * 8 + xcd;
  if (b >= B) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cpg = C >> 3, ppp = cpg >> 3;          // channels per group; 16-byte pieces per pixel of this group
  const int sub = tid % ppp, c0 = g * cpg + sub * 8;
  const int npieces = HW * ppp;
  float add[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) add[j] = chan_add ? chan_add[(long)b * C + c0 + j] : 0.f;
  const bf16* xb = x + (long)b * HW * ldx + c0;
  bf16x8 v[OP_MAXP];
#pragma unroll
  for (int k = 0; k < OP_MAXP; ++k) {
    const int i = tid + k * OP_THREADS;
    v[k] = zero8_norm();
    if (i < npieces) v[k] = *reinterpret_cast<const bf16x8*>(xb + (long)(i / ppp) * ldx);
  }
  float s = 0.f, ss = 0.f, cnt = 0.f;
#pragma unroll
  for (int k = 0; k < OP_MAXP; ++k) {
    if (tid + k * OP_THREADS < npieces) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[k][j] + add[j];
        s += f; ss += f * f;
      }
      cnt += 8.f;
    }
  }
  Wf w;
  w.n = cnt;
  w.mean = cnt > 0.f ? s / cnt : 0.f;
  w.m2 = cnt > 0.f ? fmaxf(ss - s * w.mean, 0.f) : 0.f;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    Wf t;
    t.n = __shfl_xor(w.n, o, 64); t.mean = __shfl_xor(w.mean, o, 64); t.m2 = __shfl_xor(w.m2, o, 64);
    w = wf_merge(w, t);
  }
  if (lane == 0) red[wave] = w;
  __syncthreads();
  if (tid == 0) {
    Wf acc = red[0];
    for (int i = 1; i < OP_THREADS / 64; ++i) acc = wf_merge(acc, red[i]);
    const float rstd = rsqrtf(acc.m2 / acc.n + eps);
    s_ms[0] = acc.mean; s_ms[1] = rstd;
    if (stats) { stats[((long)b * 8 + g) * 2] = acc.mean; stats[((long)b * 8 + g) * 2 + 1] = rstd; }
  }
  __syncthreads();
  const float mean = s_ms[0], rstd = s_ms[1];
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = gamma[c0 + j] * rstd;
    sh[j] = beta[c0 + j] + (add[j] - mean) * sc[j];
  }
  bf16* yb = y + (long)b * HW * ldy + c0;
#pragma unroll
  for (int k = 0; k < OP_MAXP; ++k) {
    const int i = tid + k * OP_THREADS;
    if (i < npieces) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float z = (float)v[k][j] * sc[j] + sh[j];
        o[j] = (bf16)(act ? dm_silu(z) : z);
      }
      *reinterpret_cast<bf16x8*>(yb + (long)(i / ppp) * ldy) = o;
    }
  }
}
// HALF-CU form of the same one-pass kernel: 512 threads at <= 128 VGPRs (4 waves per SIMD), so a workgroup is half a CU's
// register file and a workgroup of another stream -- or a second one of these -- runs beside it (the whole-CU form above
// time-slices with the other streams of the training step).  MAXP pieces per thread, instantiated per slab size (5 / 9 / 18:
// the UNet's 256ch@24x24, 128ch@48x48, 256ch@48x48) so that a small slab does not pay the largest one's registers.  The slab
// is addressed through buffer descriptors over the sample: ONE per-lane byte offset plus a scalar offset per piece (the
// pieces of a thread are a constant pixel stride apart) instead of MAXP 64-bit addresses, and a piece past the sample's
// last pixel is out of range -- the hardware returns zeros for the load and drops the store.
constexpr int OP2_THREADS = 512;
template <int MAXP>
__global__ __launch_bounds__(OP2_THREADS, 4) void gn_onepass2_fwd_kernel(const bf16* __restrict__ x, long ldx,
                                                                          const float* __restrict__ chan_add,
                                                                          const float* __restrict__ gamma,
                                                                          const float* __restrict__ beta,
                                                                          float* __restrict__ stats, bf16* __restrict__ y, long ldy,
                                                                          int B, int HW, int C, float eps, int act) {
  constexpr int T = OP2_THREADS;
  __shared__ Wf red[T / 64];
  __shared__ float s_ms[2];
  const int id = blockIdx.x, xcd = id & 7, jj = id >> 3, g = jj & 7, b = (jj >> 3) * 8 + xcd;
  if (b >= B) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cpg = C >> 3, ppp = cpg >> 3;          // channels per group; 16-byte pieces per pixel of this group
  const int sub = tid % ppp, c0 = g * cpg + sub * 8;
  const int pix0 = tid / ppp, pstep = T / ppp;      // piece k of this thread: pixel pix0 + k * pstep
  float add[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) add[j] = chan_add ? chan_add[(long)b * C + c0 + j] : 0.f;
  const int xbytes = (int)((((long)HW - 1) * ldx + C) * 2), ybytes = (int)((((long)HW - 1) * ldy + C) * 2);
  const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)b * HW * ldx), 0, xbytes, 0x00020000);
  const auto ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (long)b * HW * ldy), 0, ybytes, 0x00020000);
  const int vx = (int)((pix0 * ldx + c0) * 2), vy = (int)((pix0 * ldy + c0) * 2);
  const int sx = (int)(pstep * ldx * 2), sy = (int)(pstep * ldy * 2);
  u32x4 v[MAXP];
#pragma unroll
  for (int k = 0; k < MAXP; ++k) v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vx, k * sx, 0));
  float s = 0.f, ss = 0.f, cnt = 0.f;
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    if (pix0 + k * pstep < HW) {
      const bf16x8 h = __builtin_bit_cast(bf16x8, v[k]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)h[j] + add[j];
        s += f; ss += f * f;
      }
      cnt += 8.f;
    }
  }
  Wf w;
  w.n = cnt;
  w.mean = cnt > 0.f ? s / cnt : 0.f;
  w.m2 = cnt > 0.f ? fmaxf(ss - s * w.mean, 0.f) : 0.f;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    Wf t;
    t.n = __shfl_xor(w.n, o, 64); t.mean = __shfl_xor(w.mean, o, 64); t.m2 = __shfl_xor(w.m2, o, 64);
    w = wf_merge(w, t);
  }
  if (lane == 0) red[wave] = w;
  __syncthreads();
  if (tid == 0) {
    Wf acc = red[0];
    for (int i = 1; i < T / 64; ++i) acc = wf_merge(acc, red[i]);
    const float rstd = rsqrtf(acc.m2 / acc.n + eps);
    s_ms[0] = acc.mean; s_ms[1] = rstd;
    if (stats) { stats[((long)b * 8 + g) * 2] = acc.mean; stats[((long)b * 8 + g) * 2 + 1] = rstd; }
  }
  __syncthreads();
  const float mean = s_ms[0], rstd = s_ms[1];
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = gamma[c0 + j] * rstd;
    sh[j] = beta[c0 + j] + (add[j] - mean) * sc[j];
  }
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    if (pix0 + k * pstep < HW) {
      const bf16x8 h = __builtin_bit_cast(bf16x8, v[k]);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float z = (float)h[j] * sc[j] + sh[j];
        o[j] = (bf16)(act ? dm_silu(z) : z);
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(ry, 0, 0, 0)), o),
                                             ry, vy, k * sy, 0);
    }
  }
}
int gn_onepass_mode() {   // 0: two passes, 1: whole-CU one pass, 2: half-CU one pass (the default for 16-channel groups and wider)
  const char* e = getenv("DM_GN_ONEPASS");
  return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2;
}
bool gn_onepass_ok(int HW, int C, int G) {
  // OPT-IN (DM_GN_ONEPASS=1; read per call: tests switch it at run time).  Alone the kernel beats the two passes (table
  // below), but its workgroup is a whole CU's registers (1024 threads x 126 VGPRs), so inside the step it cannot share a
  // CU with the other streams' workgroups and waits for CUs to drain: training step and sampling measured equal at best
  // (14.52 vs 14.52 ms, 28.2 vs 28.3 img/s), its kernel time in the sampling loop 265 us on average (max 1.36 ms) against
  // ~100 us for the pair; capped at 64 VGPRs (two workgroups per CU) it spills and loses outright (15.38 vs 15.22 ms).
  const int mode = gn_onepass_mode();
  const bool on = mode != 0;
  const int cpg = G > 0 ? C / G : 0;
  if (mode == 2) {   // half-CU form: up to 18 pieces per thread of a 512-thread workgroup
    static const int min_cpg = [] { const char* e = getenv("DM_GN_ONEPASS_MINCPG"); return e ? atoi(e) : 16; }();   // developer switch
    return G == 8 && cpg >= min_cpg && cpg % 8 == 0 && OP2_THREADS % (cpg / 8) == 0 && (long)HW * (cpg / 8) <= (long)OP2_THREADS * 18;
  }
  // 16-channel groups and wider (32-byte pieces and up).  With 8-channel groups (64 channels) a slab is a 16-byte piece
  // of every 128-byte pixel row and the strided stream loses: 64ch@96x96, B=64: 55.7 us against 44.9 us in two passes;
  // 128ch@48x48 30.4 -> 24.0 us, 256ch@24x24 19.5 -> 14.2, 256ch@48x48 44.7 -> 34.9 (tools/hbmbench.py).
  return on && G == 8 && cpg >= 16 && cpg % 8 == 0 && OP_THREADS % (cpg / 8) == 0 && (long)HW * (cpg / 8) <= (long)OP_THREADS * OP_MAXP;
}

// grid (B): merge the sample's (split, group) partials -> stats[b][g] = {mean, rstd}; scale[b][c] = gamma*rstd,
// shift[b][c] = beta + (chan_add - mean)*scale  (the affine a consumer applies in place of gn_apply_kernel)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int nsplit,
                                                           const float* __restrict__ chan_add,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ stats,
                                                           float* __restrict__ scale, float* __restrict__ shift, int C,
                                                           int G, float eps) {
  __shared__ float s_mean[64], s_rstd[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < G) {
    Wf acc = {0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s) {
      const float* o = partial + (((long)b * nsplit + s) * G + tid) * 3;
      Wf w = {o[0], o[1], o[2]};
      acc = wf_merge(acc, w);
    }
    const float rstd = rsqrtf(acc.m2 / acc.n + eps);
    s_mean[tid] = acc.mean; s_rstd[tid] = rstd;
    stats[((long)b * G + tid) * 2] = acc.mean; stats[((long)b * G + tid) * 2 + 1] = rstd;
  }
  __syncthreads();
  const int cpg = C / G;
  for (int c = tid; c < C; c += 256) {
    const int g = c / cpg;
    const float sc = gamma[c] * s_rstd[g];
    const float a = chan_add ? chan_add[(long)b * C + c] : 0.f;
    scale[(long)b * C + c] = sc;
    shift[(long)b * C + c] = beta[c] + (a - s_mean[g]) * sc;   // the same expression gn_apply_kernel evaluates
  }
}

// ---------------------------------------------------------------- backward
// pass 1: grid (nsplit, B): ws[b][split][c] = { sum_p dz, sum_p dz*xhat }
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const bf16* __restrict__ x, long ldx,
                                                             const float* __restrict__ chan_add,
                                                             const bf16* __restrict__ dy, long lddy,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ stats,
                                                             float* __restrict__ ws, float* __restrict__ wsg,
                                                             int HW, int C, int G, int nsplit, int act) {
  __shared__ float red[2][256 * 8 + 8];
  __shared__ float s_wp[(MAX_C / 64) * 2];   // wide groups (C/G > 64): one partial per 64-channel wave slice, summed in order
  const int b = blockIdx.y, split = blockIdx.x, tid = threadIdx.x;
  const int tpp = C >> 3, ppi = 256 / tpp, cpg = C / G;
  const int c0 = (tid % tpp) * 8, prow = tid / tpp;
  const int g = c0 / cpg;
  const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
  float ga[8], be[8], add[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ga[j] = gamma[c0 + j]; be[j] = beta[c0 + j];
    add[j] = chan_add ? chan_add[(long)b * C + c0 + j] : 0.f;
    s1[j] = 0.f; s2[j] = 0.f;
  }
  const int per = (HW + nsplit - 1) / nsplit;
  const int p_begin = split * per, p_end = min(HW, p_begin + per);
  const bf16* xb = x + (long)b * HW * ldx + c0;
  const bf16* db_ = dy + (long)b * HW * lddy + c0;
  auto one = [&](bf16x8 v, bf16x8 d) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = ((float)v[j] + add[j] - mean) * rstd;
      float dz = (float)d[j];
      if (act) {
        const float z = xh * ga[j] + be[j];
        const float sg = dm_sigmoid(z);
        dz *= sg * (1.f + z * (1.f - sg));
      }
      s1[j] += dz; s2[j] += dz * xh;
    }
  };
  int pix = p_begin + prow;
  for (; pix + ppi < p_end; pix += 2 * ppi) {  // four independent loads in flight
    const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(xb + (long)pix * ldx);
    const bf16x8 d0 = *reinterpret_cast<const bf16x8*>(db_ + (long)pix * lddy);
    const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(xb + (long)(pix + ppi) * ldx);
    const bf16x8 d1 = *reinterpret_cast<const bf16x8*>(db_ + (long)(pix + ppi) * lddy);
    one(v0, d0);
    one(v1, d1);
  }
  if (pix < p_end)
    one(*reinterpret_cast<const bf16x8*>(xb + (long)pix * ldx), *reinterpret_cast<const bf16x8*>(db_ + (long)pix * lddy));
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[0][tid * 8 + j] = s1[j]; red[1][tid * 8 + j] = s2[j]; }
  __syncthreads();
  // thread t < C sums channel t over the ppi pixel-rows
  for (int c = tid; c < C; c += 256) {
    float a1 = 0.f, a2 = 0.f;
    const int t0 = c >> 3, j = c & 7;
    for (int pr = 0; pr < ppi; ++pr) {
      a1 += red[0][(pr * tpp + t0) * 8 + j];
      a2 += red[1][(pr * tpp + t0) * 8 + j];
    }
    float* o = ws + (((long)b * nsplit + split) * C + c) * 2;
    o[0] = a1; o[1] = a2;
    // group partials {sum_c gamma*S1, sum_c gamma*S2}: all the apply pass needs.  Consecutive lanes hold consecutive
    // channels: a group of cpg <= 64 channels sits inside one wave (shuffles only); wider groups (C/G > 64, the
    // latent-model widths) combine their per-wave sums through LDS.
    float g1 = gamma[c] * a1, g2 = gamma[c] * a2;
    const int span = cpg < 64 ? cpg : 64;
    for (int o2 = span >> 1; o2 > 0; o2 >>= 1) { g1 += __shfl_xor(g1, o2, 64); g2 += __shfl_xor(g2, o2, 64); }
    if (cpg <= 64) {
      if ((c % cpg) == 0) {
        float* og = wsg + (((long)b * nsplit + split) * G + c / cpg) * 2;
        og[0] = g1; og[1] = g2;
      }
    } else if ((c & 63) == 0) {
      s_wp[(c >> 6) * 2] = g1;
      s_wp[(c >> 6) * 2 + 1] = g2;
    }
  }
  if (cpg > 64) {
    __syncthreads();
    if (tid < 2 * G) {   // group tid >> 1, component tid & 1: its cpg / 64 wave slices in ascending order (no atomics)
      const int g = tid >> 1, k = tid & 1, per = cpg >> 6;
      float a = 0.f;
      for (int i = 0; i < per; ++i) a += s_wp[(g * per + i) * 2 + k];
      wsg[((long)b * nsplit + split) * G * 2 + tid] = a;
    }
  }
}

// pass 2: grid (nchunk, B): dx = rstd * (dz*gamma - m1 - xhat*m2) (+ add_in), optional dropout mask on dx.
// The group means m1, m2 come from the reduce pass's (split, group) partials.
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(
    const bf16* __restrict__ x, long ldx, const float* __restrict__ chan_add, const bf16* __restrict__ dy,
    long lddy, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ stats,
    const float* __restrict__ wsg, int nsplit, const bf16* __restrict__ add_in, long ldadd,
    bf16* __restrict__ dx, long lddx, int HW, int C, int G, int nchunk, int act, uint32_t drop_seed,
    uint32_t drop_thresh, float drop_scale, float* __restrict__ dx_colsum, const uint32_t* __restrict__ seed_ctr) {
  if (drop_thresh) drop_seed = dm_eff_seed(drop_seed, seed_ctr);
  __shared__ float s_red[256 * 8], s_m1[64], s_m2[64];
  __shared__ float s_part[PART_LDS * 2];
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int cpg = C / G;
  // group means of the sample from the reduce pass's (split, group) partials: one parallel load per thread
  const bool staged = nsplit * G <= PART_LDS;
  if (staged) {
    for (int i = tid; i < nsplit * G * 2; i += 256) s_part[i] = wsg[(long)b * nsplit * G * 2 + i];
    __syncthreads();
  }
  if (tid < G) {
    float m1 = 0.f, m2 = 0.f;
    for (int s = 0; s < nsplit; ++s) {
      const float* o = staged ? s_part + (s * G + tid) * 2 : wsg + (((long)b * nsplit + s) * G + tid) * 2;
      m1 += o[0]; m2 += o[1];
    }
    const float inv = 1.f / ((float)cpg * (float)HW);
    s_m1[tid] = m1 * inv; s_m2[tid] = m2 * inv;
  }
  __syncthreads();
  const int tpp = C >> 3, ppi = 256 / tpp;
  const int c0 = (tid % tpp) * 8, prow = tid / tpp;
  const int g = c0 / cpg;
  const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
  const float m1 = s_m1[g], m2 = s_m2[g];
  float ga[8], be[8], add[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ga[j] = gamma[c0 + j]; be[j] = beta[c0 + j];
    add[j] = chan_add ? chan_add[(long)b * C + c0 + j] : 0.f;
  }
  const int per = (HW + nchunk - 1) / nchunk;
  const int p_begin = chunk * per, p_end = min(HW, p_begin + per);
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // per-(sample, channel) sums of dx (optional output)
  const bf16x8 z8 = {};
  auto one = [&](bf16x8 v, bf16x8 d, bf16x8 a8, long gp) {
    bf16x8 o;
    uint32_t db[4] = {0u, 0u, 0u, 0u};
    if (drop_thresh) {
      const uint32_t rs = dm_drop_row(drop_seed, (uint32_t)gp);
#pragma unroll
      for (int j = 0; j < 4; ++j) db[j] = dm_drop_bits(rs, ((uint32_t)c0 >> 1) + j);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = ((float)v[j] + add[j] - mean) * rstd;
      float dz = (float)d[j];
      if (act) {
        const float z = xh * ga[j] + be[j];
        const float sg = dm_sigmoid(z);
        dz *= sg * (1.f + z * (1.f - sg));
      }
      float r = rstd * (dz * ga[j] - m1 - xh * m2);
      if (drop_thresh) r = dm_keep16(db[j >> 1], (uint32_t)j, drop_thresh) ? r * drop_scale : 0.f;
      if (add_in) r += (float)a8[j];
      o[j] = (bf16)r;
      csum[j] += (float)o[j];
    }
    *reinterpret_cast<bf16x8*>(dx + gp * lddx + c0) = o;
  };
  int pix = p_begin + prow;
  for (; pix + ppi < p_end; pix += 2 * ppi) {  // two pixels per iteration: four to six independent loads in flight
    const long g0 = (long)b * HW + pix, g1 = g0 + ppi;
    const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(x + g0 * ldx + c0);
    const bf16x8 d0 = *reinterpret_cast<const bf16x8*>(dy + g0 * lddy + c0);
    const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(x + g1 * ldx + c0);
    const bf16x8 d1 = *reinterpret_cast<const bf16x8*>(dy + g1 * lddy + c0);
    bf16x8 a0 = z8, a1 = z8;
    if (add_in) {
      a0 = *reinterpret_cast<const bf16x8*>(add_in + g0 * ldadd + c0);
      a1 = *reinterpret_cast<const bf16x8*>(add_in + g1 * ldadd + c0);
    }
    one(v0, d0, a0, g0);
    one(v1, d1, a1, g1);
  }
  if (pix < p_end) {
    const long g0 = (long)b * HW + pix;
    one(*reinterpret_cast<const bf16x8*>(x + g0 * ldx + c0), *reinterpret_cast<const bf16x8*>(dy + g0 * lddy + c0),
        add_in ? *reinterpret_cast<const bf16x8*>(add_in + g0 * ldadd + c0) : z8, g0);
  }
  if (dx_colsum) {
    // column sums of the written dx (the time-embedding / bias gradient of the ResBlock, instead of a separate pass
    // over dx): the block's pixel rows are combined in order through LDS and the (chunk, channel) partial goes to the
    // workspace; gn_param_final_kernel adds the chunks in order -- no atomics anywhere: bit-reproducible
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) s_red[tid * 8 + j] = csum[j];
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      float a = 0.f;
      for (int pr = 0; pr < ppi; ++pr) a += s_red[(pr * tpp + (c >> 3)) * 8 + (c & 7)];
      dx_colsum[((long)b * nchunk + chunk) * C + c] = a;   // (dx_colsum = the workspace's column-sum partial table here)
    }
  }
}

// ONE-PASS backward for the slabs the half-CU forward takes (16-channel groups and wider, up to 9 pieces per thread):
// a 512-thread workgroup at <= 128 VGPRs owns one (sample, group) slab.  The dy slab rides in registers, the x slab in
// LDS (each thread writes and re-reads its own 16-byte pieces: conflict-free by construction), so x and dy are read
// ONCE: 6 B/element (8 with the residual add) instead of the two-pass 10 (12).  Phase 1 forms the per-channel sums
// {sum dz, sum dz*xhat}; the workgroup holds every pixel of its channels, so the sums are complete: they go to the
// workspace row of the sample (the dgamma / dbeta reduction reads one row per sample) and, weighted by gamma, give the
// group means the apply phase needs -- no (split, group) partial table.  All merges are a fixed shuffle tree plus ordered
// LDS sums: bit-reproducible like the two-pass kernels.  Out-of-range pieces (pixel >= HW) read zeros through the buffer
// descriptor (dz = 0 adds nothing to the sums) and their stores are dropped.
constexpr int OPB_THREADS = 512;
// MINW = waves per SIMD the register budget is cut for: 4 (<= 128 VGPRs, two workgroups per CU) up to 9 pieces per thread;
// the 18-piece slabs (256ch@48x48: 147 KB of LDS, one workgroup per CU anyway) take MINW = 2.
__device__ __forceinline__ void opb_wait_vmcnt(int n) {   // s_waitcnt vmcnt(n) for a compile-time-foldable n
#define DM_VMC(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    DM_VMC(0) DM_VMC(2) DM_VMC(4) DM_VMC(6) DM_VMC(8) DM_VMC(10) DM_VMC(12) DM_VMC(14) DM_VMC(16) DM_VMC(18) DM_VMC(20)
    DM_VMC(22) DM_VMC(24) DM_VMC(26) DM_VMC(28) DM_VMC(30) DM_VMC(32) DM_VMC(34)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#undef DM_VMC
}
template <int MAXP, int MINW, bool ACT>
__global__ __launch_bounds__(OPB_THREADS, MINW) void gn_onepass2_bwd_kernel(
    const bf16* __restrict__ x, long ldx, const float* __restrict__ chan_add, const bf16* __restrict__ dy, long lddy,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ stats,
    const bf16* __restrict__ add_in, long ldadd, bf16* __restrict__ dx, long lddx, float* __restrict__ ws,
    float* __restrict__ colpart, int B, int HW, int C, uint32_t drop_seed, uint32_t drop_thresh, float drop_scale,
    const uint32_t* __restrict__ seed_ctr) {
  constexpr int T = OPB_THREADS, NW = T / 64, CH = 8;   // CH: channels per 16-byte piece
  typedef u32x4 U;
  typedef bf16x8 Hv;
  extern __shared__ __attribute__((aligned(16))) char op_smem[];
  U* sx = reinterpret_cast<U*>(op_smem);                                             // [MAXP][T] x pieces
  float* red = reinterpret_cast<float*>(op_smem + (size_t)MAXP * T * sizeof(U));     // [NW][cpg][2] wave partials
  __shared__ float s_g[2 * 128], s_m[2];
  if (drop_thresh) drop_seed = dm_eff_seed(drop_seed, seed_ctr);
  const int id = blockIdx.x, xcd = id & 7, jj = id >> 3, g = jj & 7, b = (jj >> 3) * 8 + xcd;
  if (b >= B) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cpg = C >> 3, ppp = cpg / CH;
  const int sub = tid % ppp, c0 = g * cpg + sub * CH;
  const int pix0 = tid / ppp, pstep = T / ppp;
  const float mean = stats[((long)b * 8 + g) * 2], rstd = stats[((long)b * 8 + g) * 2 + 1];
  float ga[CH], be[CH], add[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    ga[j] = gamma[c0 + j]; be[j] = beta[c0 + j];
    add[j] = chan_add ? chan_add[(long)b * C + c0 + j] : 0.f;
  }
  const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)b * HW * ldx), 0, (int)((((long)HW - 1) * ldx + C) * 2), 0x00020000);
  const auto rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (long)b * HW * lddy), 0, (int)((((long)HW - 1) * lddy + C) * 2), 0x00020000);
  const auto ro = __builtin_amdgcn_make_buffer_rsrc((void*)(dx + (long)b * HW * lddx), 0, (int)((((long)HW - 1) * lddx + C) * 2), 0x00020000);
  const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)(add_in ? add_in + (long)b * HW * ldadd : x), 0,
                                                    add_in ? (int)((((long)HW - 1) * ldadd + C) * 2) : 0, 0x00020000);
  const int vx = (int)((pix0 * ldx + c0) * 2), vd = (int)((pix0 * lddy + c0) * 2), vo = (int)((pix0 * lddx + c0) * 2),
            va = (int)((pix0 * ldadd + c0) * 2);
  const int sxs = (int)(pstep * ldx * 2), sds = (int)(pstep * lddy * 2), sos = (int)(pstep * lddx * 2), sas = (int)(pstep * ldadd * 2);
  // the two per-element quantities both phases need (same expressions as the two-pass kernels)
  auto dz_xh = [&](float xv, float dv, int j, float& xh) {
    xh = (xv + add[j] - mean) * rstd;
    float dz = dv;
    if constexpr (ACT) {
      const float z = xh * ga[j] + be[j];
      const float sg = dm_sigmoid(z);
      dz *= sg * (1.f + z * (1.f - sg));
    }
    return dz;
  };
  // ---- phase 1: dy slab -> registers; x slab -> LDS by direct-to-LDS buffer loads (no registers: the whole slab is in
  // flight at once; a wave's 64 lanes land contiguously at its wave-uniform base, which is the [piece][thread] image);
  // then the channel sums.  Each lane reads back only what its own wave's loads wrote: no barrier, a vmcnt wait suffices --
  // but the compiler does not order a ds_read behind the LDS-DMA that fills it, so the waits are explicit: the loads are
  // issued strictly as (x0, dy0, x1, dy1, ...) and piece k waits for vmcnt(2 * (MAXP - 1 - k)), i.e. for its own two loads
  // only, while the later pieces are still in flight (load and arithmetic overlap inside every wave).
  U dv[MAXP];
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(sx + k * T + wave * 64), 16, vx, k * sxs, 0, 0);
    dv[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, vd, k * sds, 0));
    __builtin_amdgcn_sched_barrier(0);   // keep the issue order the counted waits below assume
  }
  float s1[CH], s2[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  constexpr int XB = 3;   // (phase 2: residual pieces in flight beside the dy slab)
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    opb_wait_vmcnt(2 * (MAXP - 1 - k));
    __builtin_amdgcn_sched_barrier(0);
    if (pix0 + k * pstep < HW) {   // (a piece past the last pixel: nothing was written to its LDS slot)
      const Hv h = __builtin_bit_cast(Hv, sx[k * T + tid]), d = __builtin_bit_cast(Hv, dv[k]);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        float xh;
        const float dz = dz_xh((float)h[j], (float)d[j], j, xh);
        s1[j] += dz; s2[j] += dz * xh;
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // one piece at a time: interleaving the unrolled pieces spills
  }
  // fixed shuffle tree over the lanes that hold the same CH channels, then the waves in order through LDS
  auto channel_sums = [&](float (&a)[CH], float (&c)[CH], bool two) {
    for (int o = ppp; o < 64; o <<= 1) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        a[j] += __shfl_xor(a[j], o, 64);
        if (two) c[j] += __shfl_xor(c[j], o, 64);
      }
    }
    if (lane < ppp) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        red[((wave * cpg) + lane * CH + j) * 2] = a[j];
        if (two) red[((wave * cpg) + lane * CH + j) * 2 + 1] = c[j];
      }
    }
  };
  channel_sums(s1, s2, true);
  __syncthreads();
  if (tid < cpg) {
    float a1 = 0.f, a2 = 0.f;
    for (int w = 0; w < NW; ++w) { a1 += red[(w * cpg + tid) * 2]; a2 += red[(w * cpg + tid) * 2 + 1]; }
    float* o = ws + ((long)b * C + g * cpg + tid) * 2;   // the sample's row of {sum dz, sum dz*xhat} per channel
    o[0] = a1; o[1] = a2;
    const float gm = gamma[g * cpg + tid];
    s_g[tid * 2] = gm * a1; s_g[tid * 2 + 1] = gm * a2;
  }
  __syncthreads();
  if (tid < 2) {
    float m = 0.f;
    for (int c = 0; c < cpg; ++c) m += s_g[c * 2 + tid];
    s_m[tid] = m / ((float)cpg * (float)HW);
  }
  __syncthreads();
  const float m1 = s_m[0], m2 = s_m[1];
  // (the dy registers are made opaque here: otherwise the compiler keeps phase 1's bf16 -> fp32 conversions of the whole slab
  // alive for phase 2 -- twice the registers -- instead of converting again)
#pragma unroll
  for (int k = 0; k < MAXP; ++k)
#pragma unroll
    for (int q = 0; q < CH / 2; ++q) asm volatile("" : "+v"(dv[k][q]));
  // ---- phase 2: dx = rstd * (dz*gamma - m1 - xhat*m2), dropout mask, residual add, column sums of the written dx
  float csum[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) csum[j] = 0.f;
#pragma unroll
  for (int k0 = 0; k0 < MAXP; k0 += XB) {
    U av[XB];
    if (add_in) {
#pragma unroll
      for (int i = 0; i < XB; ++i)
        if (k0 + i < MAXP) av[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, va, (k0 + i) * sas, 0));
    }
#pragma unroll
    for (int i = 0; i < XB; ++i) {
      const int k = k0 + i;
      if (k < MAXP) {
        const int pix = pix0 + k * pstep;
        if (pix < HW) {
          const Hv h = __builtin_bit_cast(Hv, sx[k * T + tid]), d = __builtin_bit_cast(Hv, dv[k]);
          Hv a8 = {};
          if (add_in) a8 = __builtin_bit_cast(Hv, av[i]);
          uint32_t db[CH / 2];
#pragma unroll
          for (int j = 0; j < CH / 2; ++j) db[j] = 0u;
          if (drop_thresh) {
            const uint32_t rs = dm_drop_row(drop_seed, (uint32_t)((long)b * HW + pix));
#pragma unroll
            for (int j = 0; j < CH / 2; ++j) db[j] = dm_drop_bits(rs, ((uint32_t)c0 >> 1) + j);
          }
          Hv o;
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            float xh;
            const float dz = dz_xh((float)h[j], (float)d[j], j, xh);
            float r = rstd * (dz * ga[j] - m1 - xh * m2);
            if (drop_thresh) r = dm_keep16(db[j >> 1], (uint32_t)j, drop_thresh) ? r * drop_scale : 0.f;
            if (add_in) r += (float)a8[j];
            o[j] = (bf16)r;
            csum[j] += (float)o[j];
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(ro, 0, 0, 0)), o),
                                                 ro, vo, k * sos, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (colpart) {
    __syncthreads();   // (red is re-used)
    channel_sums(csum, csum, false);
    __syncthreads();
    if (tid < cpg) {
      float a = 0.f;
      for (int w = 0; w < NW; ++w) a += red[(w * cpg + tid) * 2];
      colpart[(long)b * C + g * cpg + tid] = a;
    }
  }
}
bool gn_onepass_bwd_ok(int HW, int C, int G) {
  // DM_GN_ONEPASS_BWD (developer switch, read per call: tests switch it at run time): the largest slab, in 16-byte pieces per
  // thread, the one-pass backward takes; 0 = two passes everywhere.  Default 9: measured alone (tools/hbmbench.py, B=64, whole
  // backward incl. the parameter-gradient kernels, two passes -> one): 128ch@48x48 54.4 -> 47.0 us, 256ch@24x24 39.0 -> 26.5;
  // the 18-piece slabs (256ch@48x48: one workgroup per CU) 81.8 -> 85.7 and stay on the two passes.  The kernel is bound by
  // its arithmetic (the SiLU derivative is evaluated in both phases, as in the two passes) plus the un-overlapped load
  // phase, not by HBM; inside the multi-stream training step the change is neutral (14.79 vs 14.79 ms, tools/gnb_ab.sh).
  const char* e = getenv("DM_GN_ONEPASS_BWD");
  const int maxp = e ? atoi(e) : 9;
  if (maxp <= 0 || gn_onepass_mode() != 2 || !gn_onepass_ok(HW, C, G)) return false;
  const int cpg = C / G;
  return cpg <= 128 && (long)HW * (cpg / 8) <= (long)OPB_THREADS * (maxp < 18 ? maxp : 18);
}

// dgamma[c] += sum over (sample, split) rows of ws[row][c][1], dbeta[c] += ... [0], in two ORDERED stages (no atomics:
// bit-reproducible).  Stage 1, grid (ceil(2C/64), RB): a block owns 64 consecutive floats of the 2C-float rows and R/RB
// rows (4 row lanes, loads unrolled, ordered LDS combine) and writes its share to wsp[blockIdx.y][2C].  Stage 2
// (gn_param_final_kernel): thread f adds the RB shares in order into dgamma / dbeta, and the per-(sample, chunk) column
// sums of dx in chunk order into dx_colsum.
__global__ __launch_bounds__(256) void gn_param_grad_kernel(const float* __restrict__ ws, float* __restrict__ wsp, int R,
                                                            int C) {
  __shared__ float red[256];
  const int tid = threadIdx.x, f = blockIdx.x * 64 + (tid & 63), rl = tid >> 6;
  const int rows_per = (R + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
  float a = 0.f;
  if (f < 2 * C) {
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 4) a += ws[(long)r * C * 2 + f];
  }
  red[tid] = a;
  __syncthreads();
  if (tid < 64 && f < 2 * C) wsp[(long)blockIdx.y * 2 * C + f] = red[tid] + red[tid + 64] + red[tid + 128] + red[tid + 192];
}
__global__ __launch_bounds__(256) void gn_param_final_kernel(const float* __restrict__ wsp, int RB, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, const float* __restrict__ colpart,
                                                             float* __restrict__ dx_colsum, int B, int C, int nchunk) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 2 * C) {
    float a = 0.f;
    for (int r = 0; r < RB; ++r) a += wsp[(long)r * 2 * C + i];
    ((i & 1) ? dgamma : dbeta)[i >> 1] += a;
  } else if (dx_colsum && i - 2 * C < B * C) {
    const int k = i - 2 * C, b = k / C, c = k - b * C;
    float a = 0.f;
    for (int ch = 0; ch < nchunk; ++ch) a += colpart[((long)b * nchunk + ch) * C + c];
    dx_colsum[k] += a;
  }
}

// One-stage form for few rows (the one-pass backward leaves ONE row per sample): thread f adds the R rows in order into
// dgamma / dbeta, the threads behind add the per-sample column sums of dx.  One launch instead of two on the backward's
// critical path (the small-shape configurations are launch-gap bound).
__global__ __launch_bounds__(256) void gn_param_rows_kernel(const float* __restrict__ ws, int R, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, const float* __restrict__ colpart,
                                                            float* __restrict__ dx_colsum, int B, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 2 * C) {
    float a = 0.f;
#pragma unroll 8
    for (int r = 0; r < R; ++r) a += ws[(long)r * 2 * C + i];
    ((i & 1) ? dgamma : dbeta)[i >> 1] += a;
  } else if (dx_colsum && i - 2 * C < B * C) {
    dx_colsum[i - 2 * C] += colpart[i - 2 * C];
  }
}

// Pixel chunks per sample.  Every workgroup pays a fixed prologue (per-channel affine rows, the merge of the partial
// table, a barrier), so few fat workgroups beat many thin ones: tools/hbmbench.py, B=64, workgroups in total
// 2048 -> 512 (forward) / 768 (backward): 128ch@48x48 40.8 -> 29.9 us / 65.2 -> 49.3, 256ch@24x24 24.5 -> 19.3 / 36.8 ->
// 34.6, 64ch@96x96 49.8 -> 44.8 / 79.7 -> 75.3, 256ch@48x48 50.3 -> 44.8 / 87.2 -> 76.5 (backward at 512: 78.8, at 384:
// 95.8; forward at 256: 47.0).
int pick_split(int B, int HW, bool bwd) {
  static const int tf = [] { const char* e = getenv("DM_GN_WGS"); return e ? atoi(e) : 512; }();      // developer overrides
  static const int tb = [] { const char* e = getenv("DM_GN_WGS_BWD"); return e ? atoi(e) : 768; }();
  int ns = (bwd ? tb : tf) / (B > 0 ? B : 1);
  if (ns < 1) ns = 1;
  if (ns > 64) ns = 64;
  while (ns > 1 && HW / ns < 32) ns >>= 1;
  return ns;
}

}  // namespace

extern "C" long dm_gn_workspace_bytes(int B, int HW, int C, int G) {
  const int nf = pick_split(B, HW, false), nb = pick_split(B, HW, true);
  const long fwd = (long)B * nf * G * 3 * 4;
  const long bwd = (long)B * nb * C * 2 * 4 + (long)B * nb * G * 2 * 4      // (split, channel) and (split, group) partials
                   + (long)B * nb * C * 4 + 16L * 2 * C * 4;               // column-sum partials, parameter-gradient shares
  return (fwd > bwd ? fwd : bwd) + 256;
}

extern "C" int dm_gn_silu_fwd(const void* x, long ldx, const float* chan_add, const float* gamma,
                              const float* beta, void* y, long ldy, float* stats, int B, int HW, int C, int G,
                              float eps, int act_silu, void* workspace, long ws_bytes, dm_stream_t stream) {
  DM_REQUIRE(x && y && gamma && beta && stats && workspace);
  DM_REQUIRE(C % 8 == 0 && C <= MAX_C && G > 0 && G <= 64 && C % G == 0 && (C / G) % 8 == 0);
  DM_REQUIRE(256 % (C / 8) == 0 && ldx % 8 == 0 && ldy % 8 == 0);
  DM_REQUIRE(ws_bytes >= dm_gn_workspace_bytes(B, HW, C, G));
  const int ns = pick_split(B, HW, false);
  hipStream_t st = (hipStream_t)stream;
  if (gn_onepass_ok(HW, C, G)) {
    if (gn_onepass_mode() == 2) {
      const long pieces = ((long)HW * (C / G / 8) + OP2_THREADS - 1) / OP2_THREADS;
      const dim3 grid(((B + 7) / 8) * 64), blk(OP2_THREADS);
#define DM_OP2(P) hipLaunchKernelGGL((gn_onepass2_fwd_kernel<P>), grid, blk, 0, st, (const bf16*)x, ldx, chan_add, gamma, beta, \
                                     stats, (bf16*)y, ldy, B, HW, C, eps, act_silu)
      if (pieces <= 5) { dm_set_variant("gn_onepass2_fwd<5>"); DM_OP2(5); }
      else if (pieces <= 9) { dm_set_variant("gn_onepass2_fwd<9>"); DM_OP2(9); }
      else { dm_set_variant("gn_onepass2_fwd<18>"); DM_OP2(18); }
#undef DM_OP2
    } else {
      dm_set_variant("gn_onepass_fwd");
      hipLaunchKernelGGL(gn_onepass_fwd_kernel, dim3(((B + 7) / 8) * 64), dim3(OP_THREADS), 0, st,
                         (const bf16*)x, ldx, chan_add, gamma, beta, stats, (bf16*)y, ldy, B, HW, C, eps, act_silu);
    }
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  dm_set_variant("gn_stats+gn_apply");
  hipLaunchKernelGGL(gn_stats_kernel, dim3(ns, B), dim3(256), 0, st, (const bf16*)x, ldx, chan_add,
                     (float*)workspace, HW, C, G, ns);
  DM_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_apply_kernel, dim3(ns, B), dim3(256), 0, st, (const bf16*)x, ldx, chan_add, gamma, beta,
                     (const float*)workspace, ns, stats, (bf16*)y, ldy, HW, C, G, eps, ns, act_silu);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_gn_prepare(const void* x, long ldx, const float* chan_add, const float* gamma, const float* beta,
                             float* stats, float* scale, float* shift, int B, int HW, int C, int G, float eps,
                             void* workspace, long ws_bytes, dm_stream_t stream) {
  DM_REQUIRE(x && gamma && beta && stats && scale && shift && workspace);
  DM_REQUIRE(C % 8 == 0 && C <= MAX_C && G > 0 && G <= 64 && C % G == 0 && (C / G) % 8 == 0);
  DM_REQUIRE(256 % (C / 8) == 0 && ldx % 8 == 0);
  DM_REQUIRE(ws_bytes >= dm_gn_workspace_bytes(B, HW, C, G));
  const int ns = pick_split(B, HW, false);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(ns, B), dim3(256), 0, st, (const bf16*)x, ldx, chan_add,
                     (float*)workspace, HW, C, G, ns);
  DM_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, st, (const float*)workspace, ns, chan_add, gamma, beta,
                     stats, scale, shift, C, G, eps);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_gn_silu_bwd(const void* x, long ldx, const float* chan_add, const void* dy, long lddy,
                              const float* gamma, const float* beta, const float* stats, const void* add_in,
                              long ldadd, void* dx, long lddx, float* dgamma, float* dbeta, int B, int HW, int C,
                              int G, int act_silu, float drop_p, unsigned long long drop_seed, float* dx_colsum,
                              void* workspace, long ws_bytes, dm_stream_t stream) {
  DM_REQUIRE(x && dy && dx && gamma && beta && stats && workspace && dgamma && dbeta);
  DM_REQUIRE(C % 8 == 0 && G > 0 && G <= 64 && C % G == 0 && (C / G) % 8 == 0 && C <= MAX_C);
  DM_REQUIRE(256 % (C / 8) == 0 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && (!add_in || ldadd % 8 == 0));
  DM_REQUIRE(ws_bytes >= dm_gn_workspace_bytes(B, HW, C, G) && drop_p >= 0.f && drop_p < 1.f);
  const int ns = pick_split(B, HW, true);
  float* ws = (float*)workspace;
  float* wsg = ws + (long)B * ns * C * 2;  // (split, group) partials behind the (split, channel) ones
  float* colpart = wsg + (long)B * ns * G * 2;   // per-(sample, chunk) column sums of dx
  float* wsp = colpart + (long)B * ns * C;       // parameter-gradient shares of stage 1
  hipStream_t st = (hipStream_t)stream;
  const uint32_t thresh = dm_thresh16(drop_p);
  const float scale = dm_inv_keep(drop_p);
  if (gn_onepass_bwd_ok(HW, C, G)) {
    // one (sample, group) slab per workgroup: one workspace row and one column-sum row per sample (ns = 1 below)
    const long pieces = ((long)HW * (C / G / 8) + OPB_THREADS - 1) / OPB_THREADS;
    float* colpart1 = ws + (long)B * C * 2;
    float* wsp1 = colpart1 + (long)B * C;
    const dim3 grid(((B + 7) / 8) * 64), blk(OPB_THREADS);
    int rc = DM_OK;
#define DM_OPB(P, MW)                                                                                                           \
  do {                                                                                                                      \
    constexpr int LDS = P * OPB_THREADS * 16 + (OPB_THREADS / 64) * 128 * 2 * 4;                                            \
    auto k = act_silu ? gn_onepass2_bwd_kernel<P, MW, true> : gn_onepass2_bwd_kernel<P, MW, false>;                         \
    static unsigned long long attr_devs0 = 0ull, attr_devs1 = 0ull;                                                         \
    unsigned long long& attr_devs = act_silu ? attr_devs1 : attr_devs0;                                                     \
    if (dm_first_on_device(attr_devs) &&                                                                                    \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {               \
      rc = DM_ERR_LAUNCH;                                                                                                   \
      break;                                                                                                                \
    }                                                                                                                       \
    const int lds = P * OPB_THREADS * 16 + (OPB_THREADS / 64) * (C / G) * 2 * 4;   /* <= 80 KB: two workgroups per CU */     \
    hipLaunchKernelGGL(k, grid, blk, lds, st, (const bf16*)x, ldx, chan_add, (const bf16*)dy, lddy, gamma, beta, stats,     \
                       (const bf16*)add_in, ldadd, (bf16*)dx, lddx, ws, dx_colsum ? colpart1 : nullptr, B, HW, C,           \
                       dm_seed32(drop_seed), thresh, scale, dm_dropout_counter());                                          \
  } while (0)
    if (pieces <= 5) { dm_set_variant("gn_onepass2_bwd<5>"); DM_OPB(5, 4); }
    else if (pieces <= 9) { dm_set_variant("gn_onepass2_bwd<9>"); DM_OPB(9, 4); }
    else { dm_set_variant("gn_onepass2_bwd<18>"); DM_OPB(18, 2); }
#undef DM_OPB
    if (rc != DM_OK) return rc;
    DM_LAUNCH_CHECK();
    const int nfin1 = 2 * C + (dx_colsum ? B * C : 0);
    if (B <= 256) {
      hipLaunchKernelGGL(gn_param_rows_kernel, dim3((nfin1 + 255) / 256), dim3(256), 0, st, ws, B, dgamma, dbeta, colpart1,
                         dx_colsum, B, C);
      DM_LAUNCH_CHECK();
      return DM_OK;
    }
    const int RB1 = 16;
    hipLaunchKernelGGL(gn_param_grad_kernel, dim3((2 * C + 63) / 64, RB1), dim3(256), 0, st, ws, wsp1, B, C);
    DM_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_param_final_kernel, dim3((nfin1 + 255) / 256), dim3(256), 0, st, wsp1, RB1, dgamma, dbeta, colpart1,
                       dx_colsum, B, C, 1);
    DM_LAUNCH_CHECK();
    return DM_OK;
  }
  dm_set_variant("gn_bwd_reduce+gn_bwd_apply");
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(ns, B), dim3(256), 0, st, (const bf16*)x, ldx, chan_add,
                     (const bf16*)dy, lddy, gamma, beta, stats, ws, wsg, HW, C, G, ns, act_silu);
  DM_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(ns, B), dim3(256), 0, st, (const bf16*)x, ldx, chan_add,
                     (const bf16*)dy, lddy, gamma, beta, stats, wsg, ns, (const bf16*)add_in, ldadd, (bf16*)dx,
                     lddx, HW, C, G, ns, act_silu, dm_seed32(drop_seed), thresh, scale, dx_colsum ? colpart : nullptr,
                     dm_dropout_counter());
  DM_LAUNCH_CHECK();
  const int R = B * ns, RB = R < 16 ? R : 16;
  hipLaunchKernelGGL(gn_param_grad_kernel, dim3((2 * C + 63) / 64, RB), dim3(256), 0, st, ws, wsp, R, C);
  DM_LAUNCH_CHECK();
  const int nfin = 2 * C + (dx_colsum ? B * C : 0);
  hipLaunchKernelGGL(gn_param_final_kernel, dim3((nfin + 255) / 256), dim3(256), 0, st, wsp, RB, dgamma, dbeta, colpart,
                     dx_colsum, B, C, ns);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
