// One-token cross-attention, output side, as ONE streaming pass (reference unet.py:197-220 with a single context
// token -- the CLIP-table configuration of main_stl10.py).
//
// With one key the softmax is identically 1 (dm_attn.hip, "one context token"), so per pixel n of sample b
//   attention(q, k, v)[n, head h] = Z[b, n, h] / keep * v[b, head h]          (Z: dropout mask of the one probability)
// and to_out of it is a sum of at most `heads` fixed vectors:
//   y[b, n, :] = res[b, n, :] + sum_h Z[b, n, h] * U[b, h, :],   U[b, h, :] = Wout[:, head h] . v[b, head h] / keep.
// Forward: U (tiny), then one pass res -> y (4 B/elem) instead of attention output + to_out GEMM (8 B/elem).
// Backward: one pass over dy,  dU[b, h, :] = sum_n Z[b, n, h] / keep * dy[b, n, :]  (2 B/elem; per-block partials summed
// in a fixed order: bit-reproducible), then  dWout += sum_b dU[b, h(ci), :] (x) v[b, ci]  and
// dv[b, ci] = sum_co dU[b, h(ci), co] * Wout[co, ci]  -- instead of the to_out data-gradient GEMM, its weight gradient
// and the one-key attention backward (three more passes over [B*N, C] tensors).
// The dropout words are those of dm_attn_fwd / dm_attn_bwd with Nk == 1 (row = (b*heads + h)*N + n, pair 0, column 0),
// so the two formulations draw the same mask from the same (seed, device counter).
#include "dm_common.h"

namespace {

constexpr int H8 = 8;          // heads (the kernels keep one U row per head in registers)
constexpr int MAX_BLK = 16;    // pixel blocks per sample in the backward pass (= partial sums per output)

struct Cross1Params {
  const bf16* v; long ldv;
  const float* w;
  const bf16* res; long ldr;
  bf16* y; long ldy;
  const bf16* dy; long lddy;
  float* u;
  float* dw;
  bf16* dv; long lddv;
  float* ws;
  int B, N, C, hd;
  uint32_t seed, thresh; float inv_keep;
  const uint32_t* seed_ctr;
};

// U[b][h][co] = inv_keep * sum_j w[co][h*hd + j] * v[b][h*hd + j]
__global__ __launch_bounds__(256) void cross1_u_kernel(const Cross1Params p) {
  const long n = (long)p.B * H8 * p.C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int co = (int)(i % p.C), h = (int)((i / p.C) % H8);
    const long b = i / ((long)p.C * H8);
    const float* wr = p.w + (long)co * p.C + h * p.hd;
    const bf16* vr = p.v + b * p.ldv + h * p.hd;
    float s = 0.f;
    for (int j = 0; j < p.hd; j += 8) {   // hd is a multiple of 8 (C % 64 == 0, eight heads)
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + j), w1 = *reinterpret_cast<const f32x4*>(wr + j + 4);
      const bf16x8 vv = *reinterpret_cast<const bf16x8*>(vr + j);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += w0[e] * (float)vv[e] + w1[e] * (float)vv[4 + e];
    }
    p.u[i] = s * p.inv_keep;
  }
}

// keep bits of the 8 heads of this lane's pixel: lane t of a pixel's TPQ lanes draws head t & 7, one ballot collects them
template <bool DROP>
__device__ __forceinline__ uint32_t cross1_mask(const Cross1Params& p, uint32_t seed_eff, int b, int q, bool valid, int myhead,
                                                int lane_base) {
  if constexpr (!DROP) return 0xFFu;
  bool keep = false;
  if (valid) {
    const uint32_t rs = dm_drop_row(seed_eff, (uint32_t)((b * H8 + myhead) * p.N + q));
    keep = dm_keep16(dm_drop_bits(rs, 0u), 0u, p.thresh);
  }
  const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
  return (uint32_t)(bal >> lane_base) & 0xFFu;
}

template <bool DROP>
__global__ __launch_bounds__(256) void cross1_fwd_kernel(const Cross1Params p) {
  const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  const int TPQ = p.C >> 3, QPB = 256 / TPQ;
  const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.y, tq = tid % TPQ, c8 = tq * 8, qs = tid / TPQ;
  const int lane_base = TPQ >= 64 ? 0 : lane - (lane % TPQ), myhead = tq & 7;
  float u[DROP ? H8 : 1][8];
  if constexpr (DROP) {
#pragma unroll
    for (int h = 0; h < H8; ++h) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p.u + ((long)b * H8 + h) * p.C + c8);
      const f32x4 c = *reinterpret_cast<const f32x4*>(p.u + ((long)b * H8 + h) * p.C + c8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { u[h][j] = a[j]; u[h][4 + j] = c[j]; }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) u[0][j] = 0.f;
#pragma unroll
    for (int h = 0; h < H8; ++h) {   // head order: the sum every pixel would form with an all-ones mask
      const f32x4 a = *reinterpret_cast<const f32x4*>(p.u + ((long)b * H8 + h) * p.C + c8);
      const f32x4 c = *reinterpret_cast<const f32x4*>(p.u + ((long)b * H8 + h) * p.C + c8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { u[0][j] += a[j]; u[0][4 + j] += c[j]; }
    }
  }
  for (int q0 = blockIdx.x * QPB; q0 < p.N; q0 += gridDim.x * QPB) {   // uniform trip count: the ballot needs every lane
    const int q = q0 + qs;
    const bool valid = q < p.N;
    const uint32_t mask = cross1_mask<DROP>(p, seed_eff, b, q, valid, myhead, lane_base);
    if (!valid) continue;
    const long row = (long)b * p.N + q;
    float acc[8];
    if (p.res) {
      const bf16x8 r = *reinterpret_cast<const bf16x8*>(p.res + row * p.ldr + c8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = (float)r[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    }
    if constexpr (DROP) {
#pragma unroll
      for (int h = 0; h < H8; ++h) {
        const float zf = ((mask >> h) & 1u) ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += zf * u[h][j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += u[0][j];
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[j];
    *reinterpret_cast<bf16x8*>(p.y + row * p.ldy + c8) = o;
  }
}

// ws[blk][b][h][c] = inv_keep * sum over this block's pixels of Z[b, n, h] * dy[b, n, c]   (plain stores, no atomics)
template <bool DROP>
__global__ __launch_bounds__(256) void cross1_bwd_kernel(const Cross1Params p) {
  const uint32_t seed_eff = dm_eff_seed(p.seed, p.seed_ctr);  // host seed + device counter (graph replay)
  __shared__ float sAcc[256 * 8];
  const int TPQ = p.C >> 3, QPB = 256 / TPQ;
  const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.y, tq = tid % TPQ, c8 = tq * 8, qs = tid / TPQ;
  const int lane_base = TPQ >= 64 ? 0 : lane - (lane % TPQ), myhead = tq & 7;
  constexpr int NH = DROP ? H8 : 1;   // without dropout every head's mask is all ones: one sum serves the eight
  float acc[NH][8];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[h][j] = 0.f;
  for (int q0 = blockIdx.x * QPB; q0 < p.N; q0 += gridDim.x * QPB) {
    const int q = q0 + qs;
    const bool valid = q < p.N;
    const uint32_t mask = cross1_mask<DROP>(p, seed_eff, b, q, valid, myhead, lane_base);
    if (!valid) continue;
    const bf16x8 d = *reinterpret_cast<const bf16x8*>(p.dy + ((long)b * p.N + q) * p.lddy + c8);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float zf = ((mask >> h) & 1u) ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[h][j] += zf * (float)d[j];
    }
  }
#pragma unroll
  for (int h = 0; h < NH; ++h) {   // (unrolled: acc[h] must stay a register index)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) sAcc[qs * p.C + c8 + j] = acc[h][j];
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
      float s = 0.f;
      for (int i = 0; i < QPB; ++i) s += sAcc[i * p.C + c];
      s *= p.inv_keep;
      float* dst = p.ws + (((long)blockIdx.x * p.B + b) * H8) * p.C + c;
      if constexpr (DROP) dst[(long)h * p.C] = s;
      else
        for (int hh = 0; hh < H8; ++hh) dst[(long)hh * p.C] = s;
    }
  }
}

// dU = the block partials summed in block order
__global__ __launch_bounds__(256) void cross1_du_kernel(const Cross1Params p, int nblk) {
  const long n = (long)p.B * H8 * p.C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < nblk; ++k) s += p.ws[k * n + i];   // independent loads, one fixed order of additions
    p.u[i] = s;
  }
}

// outputs [0, C*C): dw[co][ci] += sum_b dU[b][h(ci)][co] * v[b][ci];  outputs [C*C, C*C + B*C): dv[b][ci] =
// sum_co dU[b][h(ci)][co] * w[co][ci].  Four lanes per output (adjacent outputs = adjacent ci: the v / w reads stay
// coalesced, the dU reads are near-broadcasts), each lane an unrolled strided share of the reduction so that eight
// independent loads are in flight, combined by a fixed xor butterfly.  (One thread per output with a plain loop was a
// serial chain of dependent L2 loads: 64 us per call.)
__global__ __launch_bounds__(256) void cross1_grads_kernel(const Cross1Params p) {
  const long nw = (long)p.C * p.C, n = nw + (long)p.B * p.C;
  const int sub = threadIdx.x & 3;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) >> 2; i < n; i += ((long)gridDim.x * 256) >> 2) {
    float s = 0.f;
    if (i < nw) {
      const int ci = (int)(i % p.C), co = (int)(i / p.C), h = ci / p.hd;
      if (p.dw) {
        const float* du = p.u + (long)h * p.C + co;
        const bf16* vv = p.v + ci;
#pragma unroll 8
        for (int b = sub; b < p.B; b += 4) s += du[(long)b * H8 * p.C] * (float)vv[(long)b * p.ldv];
      }
    } else {
      const long k = i - nw;
      const int ci = (int)(k % p.C), h = ci / p.hd;
      const float* du = p.u + ((k / p.C) * H8 + h) * p.C;
      const float* wc = p.w + ci;
#pragma unroll 8
      for (int co = sub; co < p.C; co += 4) s += du[co] * wc[(long)co * p.C];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (sub == 0) {
      if (i < nw) { if (p.dw) p.dw[i] += s; }
      else { const long k = i - nw; p.dv[(k / p.C) * p.lddv + (k % p.C)] = (bf16)s; }
    }
  }
}

int fill(Cross1Params& p, const dm_cross1_args* a) {
  DM_REQUIRE(a && a->v && a->w_out && a->u);
  DM_REQUIRE(a->B > 0 && a->N > 0 && a->heads == H8);
  DM_REQUIRE(a->C == 64 || a->C == 128 || a->C == 256 || a->C == 512 || a->C == 1024);
  DM_REQUIRE(a->ldv % 8 == 0 && a->drop_p >= 0.f && a->drop_p < 1.f);
  p.v = (const bf16*)a->v; p.ldv = a->ldv; p.w = a->w_out;
  p.res = (const bf16*)a->res; p.ldr = a->ldr; p.y = (bf16*)a->y; p.ldy = a->ldy;
  p.dy = (const bf16*)a->dy; p.lddy = a->lddy; p.u = a->u; p.dw = a->dw_out;
  p.dv = (bf16*)a->dv; p.lddv = a->lddv; p.ws = (float*)a->workspace;
  p.B = a->B; p.N = a->N; p.C = a->C; p.hd = a->C / H8;
  // the attention convention (dm_attn.hip fill()): a 15-bit probability, compared as a 16-bit threshold
  p.seed = dm_seed32(a->drop_seed);
  p.seed_ctr = a->drop_p > 0.f ? dm_dropout_counter() : nullptr;
  uint32_t t15 = a->drop_p > 0.f ? (uint32_t)((double)a->drop_p * 32768.0 + 0.5) : 0u;
  if (t15 > 32767u) t15 = 32767u;
  p.thresh = 2u * t15;
  p.inv_keep = 32768.f / (32768.f - (float)t15);
  return DM_OK;
}

int blocks_for(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" long dm_cross1_bwd_workspace_bytes(int B, int heads, int C) {
  return (long)MAX_BLK * B * heads * C * 4;
}

extern "C" int dm_cross1_fwd(const dm_cross1_args* a, dm_stream_t stream) {
  Cross1Params p{};
  const int rc = fill(p, a);
  if (rc != DM_OK) return rc;
  DM_REQUIRE(a->y && a->ldy % 8 == 0 && (!a->res || a->ldr % 8 == 0));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cross1_u_kernel, dim3(blocks_for((long)p.B * H8 * p.C)), dim3(256), 0, st, p);
  DM_LAUNCH_CHECK();
  const int QPB = 256 / (p.C >> 3);
  int gx = (p.N + QPB - 1) / QPB;
  if (gx > 32) gx = 32;
  dm_set_variant(p.thresh ? "cross1_fwd<DROP=1>" : "cross1_fwd<DROP=0>");
  if (p.thresh) hipLaunchKernelGGL(cross1_fwd_kernel<true>, dim3(gx, p.B), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(cross1_fwd_kernel<false>, dim3(gx, p.B), dim3(256), 0, st, p);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_cross1_bwd(const dm_cross1_args* a, dm_stream_t stream) {
  Cross1Params p{};
  const int rc = fill(p, a);
  if (rc != DM_OK) return rc;
  DM_REQUIRE(a->dy && a->dv && a->workspace && a->lddy % 8 == 0);
  DM_REQUIRE(a->ws_bytes >= dm_cross1_bwd_workspace_bytes(a->B, a->heads, a->C));
  hipStream_t st = (hipStream_t)stream;
  const int QPB = 256 / (p.C >> 3);
  int gx = (p.N + QPB - 1) / QPB;
  if (gx > MAX_BLK) gx = MAX_BLK;
  dm_set_variant(p.thresh ? "cross1_bwd<DROP=1>" : "cross1_bwd<DROP=0>");
  if (p.thresh) hipLaunchKernelGGL(cross1_bwd_kernel<true>, dim3(gx, p.B), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(cross1_bwd_kernel<false>, dim3(gx, p.B), dim3(256), 0, st, p);
  DM_LAUNCH_CHECK();
  hipLaunchKernelGGL(cross1_du_kernel, dim3(blocks_for((long)p.B * H8 * p.C)), dim3(256), 0, st, p, gx);
  DM_LAUNCH_CHECK();
  hipLaunchKernelGGL(cross1_grads_kernel, dim3(blocks_for(4 * ((long)p.C * p.C + (long)p.B * p.C))), dim3(256), 0, st, p);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
