// HBM-bound elementwise / layout kernels: NCHW fp32 <-> NHWC bf16, strided add, SiLU backward,
// q-sample + v-target, weighted MSE loss + gradient, fused CFG + v->eps + DDPM/eta step,
// fused clip + AdamW + EMA over a flat fp32 parameter buffer, squared-norm reduction.
//
// Reference semantics replaced (file:line into the reference):
//   q-sample / v-target ........ main_stl10.py:456-465
//   loss ....................... main_stl10.py:477-482
//   CFG combine, v->eps ........ main_stl10.py:130-136
//   DDPM/eta step, final step .. visualization.py:92-109,144-152
//   clip_grad_norm_/AdamW/EMA .. main_stl10.py:508-510 (torch + timm semantics)
#include "dm_common.h"

#include <mutex>

namespace {

constexpr int EW_BLOCK = 256;
inline int ew_grid(long n_vec) {
  long g = (n_vec + EW_BLOCK - 1) / EW_BLOCK;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// x: [B, C, H, W] fp32 -> y: [B, H*W, ld] bf16 (channels >= C zero-filled up to Cpad)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, bf16* __restrict__ y, int B, int C, int HW,
                                    long ldy, int Cpad) {
  const long total = (long)B * HW * (Cpad >> 3);
  const int vpp = Cpad >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % vpp);
    const long bp = i / vpp;
    const int pix = (int)(bp % HW);
    const int b = (int)(bp / HW);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cv * 8 + j;
      o[j] = (bf16)(c < C ? x[((long)b * C + c) * HW + pix] : 0.f);
    }
    *reinterpret_cast<bf16x8*>(y + bp * ldy + cv * 8) = o;
  }
}

// x: [B*HW, ld] (bf16 or fp32) -> y: [B, C, HW] fp32 ; accumulate: y += (used for nothing else)
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, long ldx, float* __restrict__ y, int B, int C, int HW) {
  const long total = (long)B * C * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int pix = (int)(i % HW);
    const long bc = i / HW;
    const int c = (int)(bc % C);
    const int b = (int)(bc / C);
    y[i] = (float)x[((long)b * HW + pix) * ldx + c];
  }
}

// out[p][c] = a[p][c] + b[p][c] over [P, C] with pixel strides
__global__ void add_kernel(const bf16* __restrict__ a, long lda, const bf16* __restrict__ b, long ldb,
                           bf16* __restrict__ o, long ldo, long P, int C) {
  const int vpp = C >> 3;
  const long total = P * vpp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % vpp) * 8;
    const long p = i / vpp;
    const bf16x8 va = *reinterpret_cast<const bf16x8*>(a + p * lda + cv);
    bf16x8 r = va;
    if (b) {
      const bf16x8 vb = *reinterpret_cast<const bf16x8*>(b + p * ldb + cv);
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (bf16)((float)va[j] + (float)vb[j]);
    }
    *reinterpret_cast<bf16x8*>(o + p * ldo + cv) = r;
  }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (bf16)x[i];
}
__global__ void cast_bf16_f32_kernel(const bf16* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (float)x[i];
}

// dx = dy * silu'(x) (+ add)
__global__ void silu_bwd_kernel(const bf16* __restrict__ dy, long lddy, const bf16* __restrict__ x, long ldx,
                                bf16* __restrict__ dx, long lddx, long P, int C) {
  const int vpp = C >> 3;
  const long total = P * vpp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % vpp) * 8;
    const long p = i / vpp;
    const bf16x8 d = *reinterpret_cast<const bf16x8*>(dy + p * lddy + cv);
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + p * ldx + cv);
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float z = (float)v[j];
      const float sg = dm_sigmoid(z);
      r[j] = (bf16)((float)d[j] * sg * (1.f + z * (1.f - sg)));
    }
    *reinterpret_cast<bf16x8*>(dx + p * lddx + cv) = r;
  }
}

// FiLM forward (reference unet.py:32-42 followed by the Dropout of unet.py:105): out[b][p][c] =
// dropout(gamma[b][c] * h[b][p][c] + beta[b][c]); gb holds one fp32 row [gamma | beta] of 2C values per sample.
// The dropout word of element (pixel, channel) is the one dm_conv3x3's epilogue and dm_gn_silu_bwd draw.
__global__ void film_fwd_kernel(const bf16* __restrict__ h, long ldh, const float* __restrict__ gb,
                                bf16* __restrict__ o, long ldo, long P, int HW, int C, uint32_t seed, uint32_t thresh,
                                float scale, const uint32_t* __restrict__ ctr) {
  const int vpp = C >> 3;
  const long total = P * vpp;
  const uint32_t seed_eff = dm_eff_seed(seed, ctr);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % vpp) * 8;
    const long p = i / vpp;
    const float* row = gb + (p / HW) * 2 * C;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(h + p * ldh + cv);
    const uint32_t rs = thresh ? dm_drop_row(seed_eff, (uint32_t)p) : 0u;
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = row[cv + j] * (float)v[j] + row[C + cv + j];
      if (thresh) f = dm_keep16(dm_drop_bits(rs, ((uint32_t)(cv + j)) >> 1), (uint32_t)j, thresh) ? f * scale : 0.f;
      r[j] = (bf16)f;
    }
    *reinterpret_cast<bf16x8*>(o + p * ldo + cv) = r;
  }
}

// FiLM backward: dh0 = gamma * dh, part[chunk][b][c] = sum over the chunk's pixels of dh * h0 (dgamma partials, summed
// by the caller in chunk order: no atomics).  grid (nchunk, B).
__global__ __launch_bounds__(256) void film_bwd_kernel(const bf16* __restrict__ dh, long lddh, const bf16* __restrict__ h0,
                                                        long ldh, const float* __restrict__ gb, bf16* __restrict__ dh0,
                                                        long ldo, float* __restrict__ part, int HW, int C) {
  __shared__ float red[256 * 8];
  const int chunk = blockIdx.x, nchunk = gridDim.x, b = blockIdx.y, B = gridDim.y, tid = threadIdx.x;
  const int tpp = C >> 3, ppi = 256 / tpp;
  const int c0 = (tid % tpp) * 8, prow = tid / tpp;
  const int per = (HW + nchunk - 1) / nchunk;
  const int p_end = min(HW, (chunk + 1) * per);
  float s[8], ga[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  if (tid < tpp * ppi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) ga[j] = gb[(long)b * 2 * C + c0 + j];
    for (int r = chunk * per + prow; r < p_end; r += ppi) {
      const long gp = (long)b * HW + r;
      const bf16x8 d = *reinterpret_cast<const bf16x8*>(dh + gp * lddh + c0);
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(h0 + gp * ldh + c0);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s[j] += (float)d[j] * (float)v[j];
        o[j] = (bf16)(ga[j] * (float)d[j]);
      }
      *reinterpret_cast<bf16x8*>(dh0 + gp * ldo + c0) = o;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tid * 8 + j] = s[j];
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f;
    for (int pr = 0; pr < ppi; ++pr) a += red[(pr * tpp + (c >> 3)) * 8 + (c & 7)];
    part[((long)chunk * B + b) * C + c] = a;
  }
}

// out[b][c] (fp32) = sum_p x[b][p][c]   (bias / chan_add gradients); grid (C/8 vec groups..., B)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, long ldx, float* __restrict__ out,
                                                      int rows, int C, int accumulate) {
  // grid.x = batch index (rows per batch contiguous), block sums all rows of that batch
  __shared__ float red[256 * 8];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int tpp = C >> 3, ppi = 256 / tpp;
  const int c0 = (tid % tpp) * 8, prow = tid / tpp;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  if (tid < tpp * ppi) {
    for (int r = prow; r < rows; r += ppi) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((long)b * rows + r) * ldx + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tid * 8 + j] = s[j];
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f;
    for (int pr = 0; pr < ppi; ++pr) a += red[(pr * tpp + (c >> 3)) * 8 + (c & 7)];
    if (accumulate) out[(long)b * C + c] += a; else out[(long)b * C + c] = a;
  }
}

// out[c] += sum_b in[b][c]
__global__ __launch_bounds__(256) void batchsum_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                       int C) {
  // 64 channels x 4 row lanes per block, row loads unrolled (independent, in flight together), LDS combine
  __shared__ float red[256];
  const int tid = threadIdx.x, c = blockIdx.x * 64 + (tid & 63), rl = tid >> 6;
  float a = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int b = rl; b < B; b += 4) a += in[(long)b * C + c];
  }
  red[tid] = a;
  __syncthreads();
  if (tid < 64 && c < C) out[c] += red[tid] + red[tid + 64] + red[tid + 128] + red[tid + 192];
}

// out1[c] += sum_b in[b][c]; out2[c] += the same sum (optional); cast[b][c] = bf16(in[b][c]) (optional): the three
// consumers of a ResBlock's per-(sample, channel) gradient sums in one pass
__global__ __launch_bounds__(256) void batchsum2_cast_kernel(const float* __restrict__ in, float* __restrict__ out1,
                                                             float* __restrict__ out2, bf16* __restrict__ cast, int B,
                                                             int C) {
  __shared__ float red[256];
  const int tid = threadIdx.x, c = blockIdx.x * 64 + (tid & 63), rl = tid >> 6;
  float a = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int b = rl; b < B; b += 4) {
      const float v = in[(long)b * C + c];
      a += v;
      if (cast) cast[(long)b * C + c] = (bf16)v;
    }
  }
  red[tid] = a;
  __syncthreads();
  if (tid < 64 && c < C) {
    const float s = red[tid] + red[tid + 64] + red[tid + 128] + red[tid + 192];
    out1[c] += s;
    if (out2) out2[c] += s;
  }
}

// ---------------------------------------------------------------- diffusion math (fp32 NCHW)
__global__ void qsample_kernel(const float* __restrict__ x, const float* __restrict__ e,
                               const float* __restrict__ abar_t, float* __restrict__ xt, float* __restrict__ v,
                               int B, long chw) {
  const long nv = chw >> 2;
  const long total = (long)B * nv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / nv);
    const float a = abar_t[b];
    const float sa = sqrtf(a), so = sqrtf(1.f - a);
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 ev = reinterpret_cast<const f32x4*>(e)[i];
    reinterpret_cast<f32x4*>(xt)[i] = sa * xv + so * ev;
    reinterpret_cast<f32x4*>(v)[i] = sa * ev - so * xv;
  }
}

// loss = sum_b w[b] * sum_chw (out - v)^2 ; dout = 2 * w[b] * gscale * (out - v)
// grid (nblk, B); partial[b*nblk + blk] holds per-block sums (deterministic second stage on host stream).
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ out, const float* __restrict__ v,
                                                   const float* __restrict__ w, float gscale,
                                                   float* __restrict__ dout, float* __restrict__ partial, long chw) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const long nv = chw >> 2;
  const float wb = w[b];
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    const f32x4 d = reinterpret_cast<const f32x4*>(out)[(long)b * nv + i] - reinterpret_cast<const f32x4*>(v)[(long)b * nv + i];
    s += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    if (dout) reinterpret_cast<f32x4*>(dout)[(long)b * nv + i] = (2.f * wb * gscale) * d;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long)b * gridDim.x + blockIdx.x] = wb * (red[0] + red[1] + red[2] + red[3]);
}
__global__ void sum_partials_kernel(const float* __restrict__ partial, int n, float* __restrict__ loss) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = red[0] + red[1] + red[2] + red[3];
}

// One fused sampler step. coef[b] = {sqrt(1-abar), sqrt(abar), 1/sqrt(alpha), beta/sqrt(1-abar), noise_std, g, clamp}
__global__ void cfg_ddpm_step_kernel(const float* __restrict__ z, const float* __restrict__ vu,
                                     const float* __restrict__ vc, const float* __restrict__ noise,
                                     const float* __restrict__ coef, float* __restrict__ zo, int B, long chw) {
  const long nv = chw >> 2;
  const long total = (long)B * nv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / nv);
    const float* c = coef + (long)b * 8;
    const float sig = c[0], sa = c[1], ria = c[2], bs = c[3], ns = c[4], g = c[5], cl = c[6];
    const f32x4 zv = reinterpret_cast<const f32x4*>(z)[i];
    const f32x4 u = reinterpret_cast<const f32x4*>(vu)[i];
    const f32x4 cv = reinterpret_cast<const f32x4*>(vc)[i];
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float vhat = u[j] + g * (cv[j] - u[j]);
      const float eps = sig * zv[j] + sa * vhat;
      float t = ria * (zv[j] - bs * eps);
      r[j] = t;
    }
    if (ns != 0.f) {
      const f32x4 n = reinterpret_cast<const f32x4*>(noise)[i];
      r += ns * n;
    }
    if (cl != 0.f) {
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = fminf(fmaxf(r[j], -1.f), 1.f);
    }
    reinterpret_cast<f32x4*>(zo)[i] = r;
  }
}

// One fused DPM-Solver++(2M, midpoint) step for a v-prediction model with CFG.
// coef[b] = {g, alpha_s, sigma_s, sigma_t/sigma_s, alpha_t*(exp(-h)-1), 1/r0, second-order flag, unused}
//   vhat = g == 1 ? v_c : v_u + g (v_c - v_u);  x0 = alpha_s z - sigma_s vhat
//   z'   = ratio z - c x0 [- 0.5 c (1/r0)(x0 - x0_prev)]
__global__ void cfg_dpmpp2m_step_kernel(const float* __restrict__ z, const float* __restrict__ vu,
                                        const float* __restrict__ vc, const float* __restrict__ m1,
                                        const float* __restrict__ coef, float* __restrict__ zo,
                                        float* __restrict__ m0o, int B, long chw) {
  const long nv = chw >> 2;
  const long total = (long)B * nv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / nv);
    const float* c = coef + (long)b * 8;
    const float g = c[0], as = c[1], ss = c[2], ratio = c[3], cc = c[4], ir0 = c[5];
    const bool second = c[6] != 0.f;
    const f32x4 zv = reinterpret_cast<const f32x4*>(z)[i];
    const f32x4 cv = reinterpret_cast<const f32x4*>(vc)[i];
    f32x4 vh = cv;
    if (g != 1.f) {
      const f32x4 u = reinterpret_cast<const f32x4*>(vu)[i];
      vh = u + g * (cv - u);
    }
    const f32x4 x0 = as * zv - ss * vh;
    f32x4 r = ratio * zv - cc * x0;
    if (second) {
      const f32x4 pm = reinterpret_cast<const f32x4*>(m1)[i];
      r -= (0.5f * cc) * (ir0 * (x0 - pm));
    }
    reinterpret_cast<f32x4*>(zo)[i] = r;
    reinterpret_cast<f32x4*>(m0o)[i] = x0;
  }
}

// ---------------------------------------------------------------- optimizer
// partial[blk] = sum g^2 over the block's grid-stride slice
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[4];
  const long nv = n >> 2;
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long i = nv << 2; i < n; ++i) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// norm_out[0] = sqrt(sum partial) ; norm_out[1] = clip coefficient min(1, max_norm/(norm+1e-6)) (1 if max_norm<=0)
__global__ void finish_norm_kernel(const float* __restrict__ partial, int n, float max_norm, float pre_scale,
                                   float* __restrict__ norm_out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]) * pre_scale;
    norm_out[0] = nrm;
    norm_out[1] = (max_norm > 0.f) ? fminf(1.f, max_norm / (nrm + 1e-6f)) : 1.f;
  }
}

struct AdamArgs {
  float lr, b1, b2, eps, wd, bc1, bc2_sqrt, ema_decay, ema_rest, pre_scale;
  int adamw;  // 1: decoupled decay (AdamW), 0: Adam (L2 decay folded in the gradient, wd usually 0)
};
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, float* __restrict__ ema, const float* __restrict__ clip,
                             long n, AdamArgs a) {
  const float coef = (clip ? clip[1] : 1.f) * a.pre_scale;
  const long nv = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * coef;
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gg = gv[j];
      if (a.adamw) pv[j] *= (1.f - a.lr * a.wd); else gg += a.wd * pv[j];
      mv[j] = a.b1 * mv[j] + (1.f - a.b1) * gg;
      vv[j] = a.b2 * vv[j] + (1.f - a.b2) * gg * gg;
      const float denom = sqrtf(vv[j]) / a.bc2_sqrt + a.eps;
      pv[j] -= (a.lr / a.bc1) * (mv[j] / denom);
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (ema) {
      f32x4 ev = reinterpret_cast<f32x4*>(ema)[i];
      ev = a.ema_decay * ev + a.ema_rest * pv;
      reinterpret_cast<f32x4*>(ema)[i] = ev;
    }
  }
}

// ---- optimizer schedule resident on the device (whole-step HIP-graph replay): the scalars that change every step --
// step count, learning rate, bias corrections -- live in a small device struct that a ONE-thread kernel advances, so
// clip + AdamW + EMA take no host-computed argument and the whole optimizer step can be captured and replayed.
struct OptState {
  int step;         // optimizer steps taken (Adam's t)
  int sched_step;   // LambdaLR step (main_stl10.py:405-414, 441-446)
  float lr, bc1, bc2_sqrt;
  int pad[3];
};
__global__ void opt_state_init_kernel(OptState* s, int step, int sched_step) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { s->step = step; s->sched_step = sched_step; s->lr = 0.f; s->bc1 = 1.f; s->bc2_sqrt = 1.f; }
}
// lr = base_lr * lambda(sched_step): `warm` linear warm-up steps then cosine over total_steps (trainer.lr_lambda, the
// reference's lr_lambda restated); total_steps == 0: constant rate (reference main.py:153)
__global__ void opt_state_advance_kernel(OptState* s, double base_lr, int warm, int total_steps, double beta1, double beta2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int step = s->step + 1, ss = s->sched_step;
  double lam = 1.0;
  if (total_steps > 0) {
    if (ss < warm) lam = (double)(ss + 1) / (double)(warm > 1 ? warm : 1);
    else {
      const int den = total_steps - warm > 1 ? total_steps - warm : 1;
      lam = 0.5 * (1.0 + cos(3.141592653589793 * ((double)(ss - warm) / (double)den)));
    }
  }
  s->lr = (float)(base_lr * lam);
  s->bc1 = (float)(1.0 - pow(beta1, (double)step));
  s->bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
  s->step = step;
  s->sched_step = ss + 1;
}
__global__ void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, float* __restrict__ ema, const float* __restrict__ clip,
                                 long n, AdamArgs a, const OptState* __restrict__ st) {
  a.lr = st->lr; a.bc1 = st->bc1; a.bc2_sqrt = st->bc2_sqrt;
  const float coef = (clip ? clip[1] : 1.f) * a.pre_scale;
  const long nv = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * coef;
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // (the arithmetic of adamw_kernel, statement for statement)
      float gg = gv[j];
      if (a.adamw) pv[j] *= (1.f - a.lr * a.wd); else gg += a.wd * pv[j];
      mv[j] = a.b1 * mv[j] + (1.f - a.b1) * gg;
      vv[j] = a.b2 * vv[j] + (1.f - a.b2) * gg * gg;
      const float denom = sqrtf(vv[j]) / a.bc2_sqrt + a.eps;
      pv[j] -= (a.lr / a.bc1) * (mv[j] / denom);
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (ema) {
      f32x4 ev = reinterpret_cast<f32x4*>(ema)[i];
      ev = a.ema_decay * ev + a.ema_rest * pv;
      reinterpret_cast<f32x4*>(ema)[i] = ev;
    }
  }
}
// dst <- ring slot (counter % nslots): the input of a captured step comes from a device-resident ring of batches,
// selected by a DEVICE counter (the dropout counter the captured step bumps), so a replay needs no host copy
__global__ void ring_fetch_kernel(const u32x4* __restrict__ ring, u32x4* __restrict__ dst, long slot_vec, int nslots,
                                  const uint32_t* __restrict__ ctr) {
  const u32x4* src = ring + (long)(ctr[0] % (uint32_t)nslots) * slot_vec;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < slot_vec; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

}  // namespace

extern "C" int dm_opt_state_bytes(void) { return (int)sizeof(OptState); }

extern "C" int dm_opt_state_init(void* state_dev, int step, int sched_step, dm_stream_t stream) {
  DM_REQUIRE(state_dev && step >= 0 && sched_step >= 0);
  hipLaunchKernelGGL(opt_state_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (OptState*)state_dev, step, sched_step);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_opt_state_advance(void* state_dev, double base_lr, int warmup_steps, int total_steps, double beta1,
                                    double beta2, dm_stream_t stream) {
  DM_REQUIRE(state_dev && total_steps >= 0 && warmup_steps >= 0);
  hipLaunchKernelGGL(opt_state_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (OptState*)state_dev, base_lr,
                     warmup_steps, total_steps, beta1, beta2);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_adamw_ema_step_dev(float* p, const float* g, float* m, float* v, float* ema, const float* clip, long n,
                                     const void* state_dev, float beta1, float beta2, float eps, float wd, double ema_decay,
                                     float pre_scale, int adamw, dm_stream_t stream) {
  DM_REQUIRE(p && g && m && v && state_dev && n > 0 && n % 4 == 0);
  AdamArgs a;
  a.lr = 0.f; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = wd; a.bc1 = 1.f; a.bc2_sqrt = 1.f;
  a.ema_decay = (float)ema_decay; a.ema_rest = (float)(1.0 - ema_decay);
  a.pre_scale = pre_scale; a.adamw = adamw;
  hipLaunchKernelGGL(adamw_dev_kernel, dim3(ew_grid(n >> 2)), dim3(EW_BLOCK), 0, (hipStream_t)stream, p, g, m, v, ema, clip,
                     n, a, (const OptState*)state_dev);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_ring_fetch(const void* ring, void* dst, long slot_bytes, int nslots, const unsigned int* counter_dev,
                             dm_stream_t stream) {
  DM_REQUIRE(ring && dst && counter_dev && nslots > 0 && slot_bytes > 0 && slot_bytes % 16 == 0);
  DM_REQUIRE(((uintptr_t)ring & 15) == 0 && ((uintptr_t)dst & 15) == 0);
  hipLaunchKernelGGL(ring_fetch_kernel, dim3(ew_grid(slot_bytes / 16)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                     (const u32x4*)ring, (u32x4*)dst, slot_bytes / 16, nslots, counter_dev);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_nchw_f32_to_nhwc_bf16(const float* x, void* y, int B, int C, int HW, long ldy, int Cpad,
                                        dm_stream_t stream) {
  DM_REQUIRE(x && y && Cpad % 8 == 0 && Cpad >= C && ldy >= Cpad && ldy % 8 == 0);
  const long total = (long)B * HW * (Cpad >> 3);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, (bf16*)y,
                     B, C, HW, ldy, Cpad);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_nhwc_to_nchw_f32(const void* x, int x_is_f32, long ldx, float* y, int B, int C, int HW,
                                   dm_stream_t stream) {
  DM_REQUIRE(x && y);
  const long total = (long)B * C * HW;
  if (x_is_f32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       (const float*)x, ldx, y, B, C, HW);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       (const bf16*)x, ldx, y, B, C, HW);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_cast_f32_to_bf16(const float* x, void* y, long n, dm_stream_t stream) {
  DM_REQUIRE(x && y && n > 0);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, (bf16*)y, n);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_cast_bf16_to_f32(const void* x, float* y, long n, dm_stream_t stream) {
  DM_REQUIRE(x && y && n > 0);
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16*)x, y, n);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_add_bf16(const void* a, long lda, const void* b, long ldb, void* out, long ldo, long P, int C,
                           dm_stream_t stream) {
  DM_REQUIRE(a && out && C % 8 == 0 && lda % 8 == 0 && ldo % 8 == 0 && (!b || ldb % 8 == 0));
  hipLaunchKernelGGL(add_kernel, dim3(ew_grid(P * (C >> 3))), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const bf16*)a,
                     lda, (const bf16*)b, ldb, (bf16*)out, ldo, P, C);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_silu_bwd_bf16(const void* dy, long lddy, const void* x, long ldx, void* dx, long lddx, long P,
                                int C, dm_stream_t stream) {
  DM_REQUIRE(dy && x && dx && C % 8 == 0 && lddy % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0);
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(ew_grid(P * (C >> 3))), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                     (const bf16*)dy, lddy, (const bf16*)x, ldx, (bf16*)dx, lddx, P, C);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_film_fwd(const void* h, long ldh, const float* gamma_beta, void* out, long ldo, int B, int HW, int C,
                          float drop_p, unsigned long long drop_seed, dm_stream_t stream) {
  DM_REQUIRE(h && gamma_beta && out && B > 0 && HW > 0 && C % 8 == 0 && ldh % 8 == 0 && ldo % 8 == 0);
  DM_REQUIRE(drop_p >= 0.f && drop_p < 1.f);
  const long P = (long)B * HW;
  hipLaunchKernelGGL(film_fwd_kernel, dim3(ew_grid(P * (C >> 3))), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                     (const bf16*)h, ldh, gamma_beta, (bf16*)out, ldo, P, HW, C, dm_seed32(drop_seed), dm_thresh16(drop_p),
                     dm_inv_keep(drop_p), drop_p > 0.f ? dm_dropout_counter() : nullptr);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_film_bwd_chunks(int HW) { return HW >= 4096 ? 16 : HW >= 256 ? 4 : 1; }

extern "C" int dm_film_bwd(const void* dh, long lddh, const void* h0, long ldh, const float* gamma_beta, void* dh0,
                          long ldo, float* dgamma_partials, int B, int HW, int C, dm_stream_t stream) {
  DM_REQUIRE(dh && h0 && gamma_beta && dh0 && dgamma_partials && B > 0 && HW > 0);
  DM_REQUIRE(C % 8 == 0 && C / 8 <= 256 && lddh % 8 == 0 && ldh % 8 == 0 && ldo % 8 == 0);
  hipLaunchKernelGGL(film_bwd_kernel, dim3(dm_film_bwd_chunks(HW), B), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)dh, lddh, (const bf16*)h0, ldh, gamma_beta, (bf16*)dh0, ldo, dgamma_partials, HW, C);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_colsum_bf16(const void* x, long ldx, float* out, int nbatch, int rows_per_batch, int C,
                              int accumulate, dm_stream_t stream) {
  DM_REQUIRE(x && out && C % 8 == 0 && C <= 2048 && ldx % 8 == 0 && 256 % (C / 8 > 256 ? 256 : C / 8) == 0 && C <= 256 * 8);
  DM_REQUIRE(C / 8 <= 256);
  hipLaunchKernelGGL(colsum_kernel, dim3(nbatch), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, out,
                     rows_per_batch, C, accumulate);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_batchsum2_cast_f32(const float* in, float* out1, float* out2, void* cast_bf16, int B, int C,
                                     dm_stream_t stream) {
  DM_REQUIRE(in && out1 && B > 0 && C > 0);
  hipLaunchKernelGGL(batchsum2_cast_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, in, out1, out2,
                     (bf16*)cast_bf16, B, C);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_batchsum_f32(const float* in, float* out, int B, int C, dm_stream_t stream) {
  DM_REQUIRE(in && out);
  hipLaunchKernelGGL(batchsum_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, in, out, B, C);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_qsample_vtarget(const float* x, const float* e, const float* abar_t, float* x_t, float* v, int B,
                                  long chw, dm_stream_t stream) {
  DM_REQUIRE(x && e && abar_t && x_t && v && chw % 4 == 0);
  hipLaunchKernelGGL(qsample_kernel, dim3(ew_grid((long)B * (chw >> 2))), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, e,
                     abar_t, x_t, v, B, chw);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" long dm_mse_workspace_bytes(int B) { return (long)B * 32 * 4; }

extern "C" int dm_mse_loss_fwd_bwd(const float* out, const float* v, const float* w, float grad_scale, float* dout,
                                   float* loss, int B, long chw, void* workspace, long ws_bytes,
                                   dm_stream_t stream) {
  DM_REQUIRE(out && v && w && loss && workspace && chw % 4 == 0 && ws_bytes >= dm_mse_workspace_bytes(B));
  hipLaunchKernelGGL(mse_kernel, dim3(32, B), dim3(256), 0, (hipStream_t)stream, out, v, w, grad_scale, dout,
                     (float*)workspace, chw);
  DM_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, B * 32,
                     loss);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

// ---------------------------------------------------------------- allocation-free variant of dm_mse_loss_fwd_bwd
// For callers that cannot allocate (HIP graph capture, tight serving / monitoring loops): the per-(sample, block)
// partial sums live in a file-scope device array instead of a caller-provided workspace.  `scratch` below is a HOST-side
// stand-in for the symbol: only the address hipGetSymbolAddress returns (per device: every GPU holds its own copy) is
// ever passed to a kernel.
constexpr int DM_MSE_NOALLOC_MAX_B = 4096;                      // 4096 samples x 32 block partials x 4 B = 512 KiB per GPU
static __device__ __align__(256) float scratch[DM_MSE_NOALLOC_MAX_B * 32];
constexpr int DM_MAX_DEVICES = 64;
static float* g_scratch_ptr[DM_MAX_DEVICES];                    // device address of `scratch`, per device ordinal
static std::mutex g_scratch_mu;

// Resolve the scratch array's address on the CURRENT device (once per device).  Returns NULL on failure.
static float* scratch_for_current_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DM_MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
      dm_note_hip_error((int)e);
      return nullptr;
    }
    g_scratch_ptr[dev] = (float*)ptr;
  }
  return g_scratch_ptr[dev];
}

extern "C" int dm_init_device(void) {
  (void)hipGetLastError();
  return scratch_for_current_device() ? DM_OK : DM_ERR_LAUNCH;
}

extern "C" int dm_mse_noalloc_max_batch(void) { return DM_MSE_NOALLOC_MAX_B; }

extern "C" int dm_mse_loss_fwd_bwd_noalloc(const float* out, const float* v, const float* w, float grad_scale,
                                           float* dout, float* loss, int B, long chw, dm_stream_t stream) {
  DM_REQUIRE(out && v && w && loss && B > 0 && chw > 0 && chw % 4 == 0);
  if (B > DM_MSE_NOALLOC_MAX_B) return DM_ERR_TOO_LARGE;  // the static scratch holds 32 partials for 4096 samples
  float* ws = scratch_for_current_device();
  if (!ws) return DM_ERR_LAUNCH;
  // every partial the second kernel reads is written by the first one in this call (B * 32 of them), so nothing of
  // the previous call's contents has to be cleared; calls that overlap on one GPU (two streams) race on the array.
  hipLaunchKernelGGL(mse_kernel, dim3(32, B), dim3(256), 0, (hipStream_t)stream, out, v, w, grad_scale, dout, ws, chw);
  DM_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, B * 32, loss);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_cfg_dpmpp2m_step(const float* z, const float* v_uncond, const float* v_cond, const float* x0_prev,
                                   const float* coef, float* z_out, float* x0_out, int B, long chw, dm_stream_t stream) {
  DM_REQUIRE(z && v_uncond && v_cond && x0_prev && coef && z_out && x0_out && B > 0 && chw > 0 && chw % 4 == 0);
  hipLaunchKernelGGL(cfg_dpmpp2m_step_kernel, dim3(ew_grid((long)B * (chw >> 2))), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                     z, v_uncond, v_cond, x0_prev, coef, z_out, x0_out, B, chw);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_cfg_ddpm_step(const float* z, const float* v_uncond, const float* v_cond, const float* noise,
                                const float* coef, float* z_out, int B, long chw, dm_stream_t stream) {
  DM_REQUIRE(z && v_uncond && v_cond && coef && z_out && chw % 4 == 0);
  hipLaunchKernelGGL(cfg_ddpm_step_kernel, dim3(ew_grid((long)B * (chw >> 2))), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                     z, v_uncond, v_cond, noise, coef, z_out, B, chw);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" long dm_sqnorm_workspace_bytes(void) { return 1024 * 4; }

extern "C" int dm_grad_norm_clipcoef(const float* g, long n, float max_norm, float pre_scale, float* norm_out,
                                     void* workspace, long ws_bytes, dm_stream_t stream) {
  DM_REQUIRE(g && norm_out && workspace && ws_bytes >= dm_sqnorm_workspace_bytes() && n > 0);
  DM_REQUIRE(((uintptr_t)g & 15) == 0);
  hipLaunchKernelGGL(sqnorm_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, g, n, (float*)workspace);
  DM_LAUNCH_CHECK();
  hipLaunchKernelGGL(finish_norm_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, 1024,
                     max_norm, pre_scale, norm_out);
  DM_LAUNCH_CHECK();
  return DM_OK;
}

extern "C" int dm_adamw_ema_step(float* p, const float* g, float* m, float* v, float* ema, const float* clip,
                                 long n, float lr, float beta1, float beta2, float eps, float wd, int step,
                                 double ema_decay, float pre_scale, int adamw, dm_stream_t stream) {
  DM_REQUIRE(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1);
  AdamArgs a;
  a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = wd;
  a.bc1 = (float)(1.0 - pow((double)beta1, step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
  a.ema_decay = (float)ema_decay; a.ema_rest = (float)(1.0 - ema_decay);  // 1 - d in double: timm's (1.0 - decay)
  a.pre_scale = pre_scale; a.adamw = adamw;
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n >> 2)), dim3(EW_BLOCK), 0, (hipStream_t)stream, p, g, m, v, ema, clip,
                     n, a);
  DM_LAUNCH_CHECK();
  return DM_OK;
}
