// HBM-bound helper kernels around the conv GEMMs: layout conversion, max-pool, GroupNorm(+ReLU),
// bilinear upsample, FeatureAlign offset projection.  All NHWC, 16-byte (8 x bf16) accesses.
#include "common.h"

namespace {

// ---------------------------------------------------------------- NCHW f32 -> NHWC bf16 (pad C)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int B, int C, int HW,
                                    int cpad) {
  const long long total = (long long)B * HW;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(p / HW);
    const int hw = (int)(p - (long long)n * HW);
    const float* xp = x + (long long)n * C * HW + hw;
    uint16_t* yp = y + p * cpad;
    for (int c0 = 0; c0 < cpad; c0 += 8) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (c0 + e < C) ? xp[(long long)(c0 + e) * HW] : 0.f;
      *reinterpret_cast<uint4*>(yp + c0) = pack_bf16x8(f);
    }
  }
}

// ---------------------------------------------------------------- maxpool 3x3 s2 p1
__global__ void maxpool3x3s2_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int B, int H, int W,
                                    int C, int Ho, int Wo) {
  const int c8 = C / 8;
  const long long total = (long long)B * Ho * Wo * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    long long p = i / c8;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int dh = 0; dh < 3; ++dh) {
      const int hi = ho * 2 - 1 + dh;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int wi = wo * 2 - 1 + dw;
        if ((unsigned)wi >= (unsigned)W) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(x + (((long long)n * H + hi) * W + wi) * C + cc * 8);
        float f[8];
        unpack_bf16x8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], f[e]);
      }
    }
    *reinterpret_cast<uint4*>(y + (((long long)n * Ho + ho) * Wo + wo) * C + cc * 8) = pack_bf16x8(m);
  }
}

// ---------------------------------------------------------------- GroupNorm
struct GnArgs {
  int nlev, batch, C, groups, cpg;  // cpg = channels per group (multiple of 8)
  int hw[SM_MAX_LEVELS];
  long long row0[SM_MAX_LEVELS];
  int blk0[SM_MAX_LEVELS + 1];  // first block of each level (per image)
  float eps;
  int relu;
  int rpb;  // rows of one (image, level) per block
};

constexpr int GN_ROWS_PER_BLOCK = 256;

// Each block reduces GN_ROWS_PER_BLOCK rows of one (image, level); a thread owns one
// 16-byte chunk column (8 channels, inside one group) and strides over rows.
__global__ __launch_bounds__(256) void gn_stats_kernel(const uint16_t* __restrict__ x, unsigned long long* __restrict__ stats,
                                                       const GnArgs a) {
  const int n = blockIdx.y;
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && (int)blockIdx.x >= a.blk0[l]) lev = l;
  const int rb = (blockIdx.x - a.blk0[lev]) * a.rpb;
  const int HW = a.hw[lev];
  const int c8 = a.C / 8;            // chunks per row
  const int rows_per_iter = 256 / c8;  // C=256 -> 8 rows per iteration
  const int cc = threadIdx.x % c8;
  const int rr = threadIdx.x / c8;
  const uint16_t* base = x + (a.row0[lev] + (long long)n * HW) * a.C;
  float s = 0.f, ss = 0.f;
  if (rr < rows_per_iter) {
    const int rend = min(rb + a.rpb, HW);
    for (int r = rb + rr; r < rend; r += rows_per_iter) {
      const uint4 v = *reinterpret_cast<const uint4*>(base + (long long)r * a.C + cc * 8);
      float f[8];
      unpack_bf16x8(v, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s += f[e];
        ss = __builtin_fmaf(f[e], f[e], ss);
      }
    }
  }
  __shared__ float sh[2][256];
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = ss;
  __syncthreads();
  // one thread per group sums its chunks over all row-lanes
  const int chunks_per_group = a.cpg / 8;
  if ((int)threadIdx.x < a.groups) {
    const int g = threadIdx.x;
    float ts = 0.f, tss = 0.f;
    for (int r = 0; r < rows_per_iter; ++r)
      for (int k = 0; k < chunks_per_group; ++k) {
        const int t = r * c8 + g * chunks_per_group + k;
        ts += sh[0][t];
        tss += sh[1][t];
      }
    // ts / tss were formed in a fixed order; the cross-block sum is fixed point (common.h: gn_fix) = order independent
    unsigned long long* st = stats + (((long long)n * a.nlev + lev) * a.groups + g) * 2;
    atomicAdd(st, gn_fix(ts));
    atomicAdd(st + 1, gn_fix(tss));
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const unsigned long long* __restrict__ stats, const GnArgs a) {
  const int n = blockIdx.y;
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && (int)blockIdx.x >= a.blk0[l]) lev = l;
  const int rb = (blockIdx.x - a.blk0[lev]) * a.rpb;
  const int HW = a.hw[lev];
  const int c8 = a.C / 8;
  const int rows_per_iter = 256 / c8;
  const int cc = threadIdx.x % c8;
  const int rr = threadIdx.x / c8;
  if (rr >= rows_per_iter) return;
  const int g = (cc * 8) / a.cpg;
  float mean, rstd;
  gn_mean_rstd(stats + (((long long)n * a.nlev + lev) * a.groups + g) * 2, (double)HW * (double)a.cpg, a.eps, &mean, &rstd);
  float sc[8], sf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float ga = gamma[cc * 8 + e] * rstd;
    sc[e] = ga;
    sf[e] = beta[cc * 8 + e] - mean * ga;
  }
  const long long off = (a.row0[lev] + (long long)n * HW) * a.C;
  const int rend = min(rb + a.rpb, HW);
  // 4 rows per iteration, all loads issued before the first use: a pure streaming pass wants bytes in flight
  // (one 16-byte load per thread per iteration measured 2.8 TB/s on the 46 MB tower tensor)
  constexpr int UN = 4;
  for (int r = rb + rr; r < rend; r += rows_per_iter * UN) {
    uint4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int ru = min(r + u * rows_per_iter, rend - 1);        // clamped duplicate load, store masked below
      v[u] = *reinterpret_cast<const uint4*>(x + off + (long long)ru * a.C + cc * 8);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int ru = r + u * rows_per_iter;
      if (ru >= rend) break;
      float f[8];
      unpack_bf16x8(v[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = f[e] * sc[e] + sf[e];
        f[e] = a.relu ? fmaxf(t, 0.f) : t;
      }
      *reinterpret_cast<uint4*>(y + off + (long long)ru * a.C + cc * 8) = pack_bf16x8(f);
    }
  }
}

// ---------------------------------------------------------------- bilinear upsample (integer factor)
template <bool F32>
__global__ void upsample_bilinear_kernel(const void* __restrict__ xin, void* __restrict__ yout, int B, int H, int W,
                                         int C, int factor, int in_cs, int out_cs, int out_coff) {
  const int Ho = H * factor, Wo = W * factor;
  constexpr int VEC = F32 ? 4 : 8;
  const int cv = C / VEC;
  const long long total = (long long)B * Ho * Wo * cv;
  const float inv = 1.f / (float)factor;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cv);
    long long p = i / cv;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    // area_pixel_compute_source_index(align_corners=False): max(0, (dst+0.5)*scale-0.5)
    float sy = fmaxf(((float)ho + 0.5f) * inv - 0.5f, 0.f);
    float sx = fmaxf(((float)wo + 0.5f) * inv - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const long long r00 = ((long long)n * H + y0) * W + x0, r01 = ((long long)n * H + y0) * W + x1;
    const long long r10 = ((long long)n * H + y1) * W + x0, r11 = ((long long)n * H + y1) * W + x1;
    const long long orow = ((long long)n * Ho + ho) * Wo + wo;
    float a[VEC], b[VEC], c[VEC], d[VEC], r[VEC];
    if constexpr (F32) {
      const float* x = (const float*)xin;
      *reinterpret_cast<float4*>(a) = *reinterpret_cast<const float4*>(x + r00 * in_cs + cc * 4);
      *reinterpret_cast<float4*>(b) = *reinterpret_cast<const float4*>(x + r01 * in_cs + cc * 4);
      *reinterpret_cast<float4*>(c) = *reinterpret_cast<const float4*>(x + r10 * in_cs + cc * 4);
      *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(x + r11 * in_cs + cc * 4);
    } else {
      const uint16_t* x = (const uint16_t*)xin;
      unpack_bf16x8(*reinterpret_cast<const uint4*>(x + r00 * in_cs + cc * 8), a);
      unpack_bf16x8(*reinterpret_cast<const uint4*>(x + r01 * in_cs + cc * 8), b);
      unpack_bf16x8(*reinterpret_cast<const uint4*>(x + r10 * in_cs + cc * 8), c);
      unpack_bf16x8(*reinterpret_cast<const uint4*>(x + r11 * in_cs + cc * 8), d);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) r[e] = hy * (hx * a[e] + lx * b[e]) + ly * (hx * c[e] + lx * d[e]);
    if constexpr (F32) {
      *reinterpret_cast<float4*>((float*)yout + orow * out_cs + out_coff + cc * 4) = *reinterpret_cast<float4*>(r);
    } else {
      *reinterpret_cast<uint4*>((uint16_t*)yout + orow * out_cs + out_coff + cc * 8) = pack_bf16x8(r);
    }
  }
}

// ---------------------------------------------------------------- FeatureAlign.conv_offset (1x1, 4 -> nout)
struct OffArgs {
  int nlev, nout, reg_cs;
  long long row0[SM_MAX_LEVELS];
  int rows[SM_MAX_LEVELS];
  float scale[SM_MAX_LEVELS];
  long long total;
};

__global__ void offset_linear_kernel(const float* __restrict__ reg, const float* __restrict__ w,
                                     float* __restrict__ out, const OffArgs a) {
  // thread per (row, out channel); 4 MACs each
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.total * a.nout;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / a.nout;
    const int o = (int)(i - r * a.nout);
    int lev = 0;
#pragma unroll
    for (int l = 1; l < SM_MAX_LEVELS; ++l)
      if (l < a.nlev && r >= a.row0[l]) lev = l;
    const float4 v = *reinterpret_cast<const float4*>(reg + r * a.reg_cs);
    const float4 ww = *reinterpret_cast<const float4*>(w + o * 4);
    // bbox_pred (x Scale) feeds conv_offset: sipmask_head.py:261-263; level_scale is 1 when the
    // producing conv already applied Scale in its epilogue
    const float s = a.scale[lev];
    out[i] = ww.x * (v.x * s) + ww.y * (v.y * s) + ww.z * (v.z * s) + ww.w * (v.w * s);
  }
}

// Adjoint of conv_offset w.r.t. ITS WEIGHT (the box prediction it reads is detached, sipmask_head.py:50):
// grad_w[o][c] = sum_rows gout[row][o] * reg[row][c].  Deterministic: a block reduces OFFB_ROWS rows in a fixed order into
// partial[block][nout*4], a second launch sums the partials in block order (no float atomics).
#define OFFB_ROWS 256
__global__ void offset_linear_bwd_partial_kernel(const float* __restrict__ reg, int reg_cs, const float* __restrict__ gout,
                                                 int nout, long long rows, float* __restrict__ partial) {
  extern __shared__ float sm_off_red[];                 // [4][nout][4]
  const int o = threadIdx.x % nout, sub = threadIdx.x / nout;       // blockDim.x = 4 * nout
  const long long r0 = (long long)blockIdx.x * OFFB_ROWS;
  const long long r1 = r0 + OFFB_ROWS < rows ? r0 + OFFB_ROWS : rows;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (long long r = r0 + sub; r < r1; r += 4) {
    const float g = gout[r * nout + o];
    const float4 v = *reinterpret_cast<const float4*>(reg + r * reg_cs);
    a0 = fmaf(g, v.x, a0);
    a1 = fmaf(g, v.y, a1);
    a2 = fmaf(g, v.z, a2);
    a3 = fmaf(g, v.w, a3);
  }
  float* mine = sm_off_red + ((size_t)sub * nout + o) * 4;
  mine[0] = a0; mine[1] = a1; mine[2] = a2; mine[3] = a3;
  __syncthreads();
  if (sub == 0) {
    float* dst = partial + (size_t)blockIdx.x * nout * 4 + o * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float t = sm_off_red[((size_t)0 * nout + o) * 4 + c];
      for (int q = 1; q < 4; ++q) t += sm_off_red[((size_t)q * nout + o) * 4 + c];
      dst[c] = t;
    }
  }
}

// one WAVE per output: lane l sums partials l, l + 64, ... in order, then a fixed shuffle tree -- deterministic, and 350
// partials are 6 dependent loads per lane instead of 350 (the first version, one thread per output: 81 us per step)
__global__ void offset_linear_bwd_final_kernel(const float* __restrict__ partial, int nblk, int n, float* __restrict__ gw) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (j >= n) return;
  float t = 0.f;
  for (int b = lane; b < nblk; b += 64) t += partial[(size_t)b * n + j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
  if (lane == 0) gw[j] = t;
}

// ================================================================ exact-f32 plan (parity mode, conv_f32.hip)
// NCHW f32 -> NHWC f32 (channels zero padded to cpad, multiple of 4)
__global__ void nchw_to_nhwc_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW,
                                        int cpad) {
  const long long total = (long long)B * HW;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(p / HW);
    const int hw = (int)(p - (long long)n * HW);
    const float* xp = x + (long long)n * C * HW + hw;
    float* yp = y + p * cpad;
    for (int c0 = 0; c0 < cpad; c0 += 4) {
      float4 v;
      v.x = (c0 + 0 < C) ? xp[(long long)(c0 + 0) * HW] : 0.f;
      v.y = (c0 + 1 < C) ? xp[(long long)(c0 + 1) * HW] : 0.f;
      v.z = (c0 + 2 < C) ? xp[(long long)(c0 + 2) * HW] : 0.f;
      v.w = (c0 + 3 < C) ? xp[(long long)(c0 + 3) * HW] : 0.f;
      *reinterpret_cast<float4*>(yp + c0) = v;
    }
  }
}

__global__ void maxpool3x3s2_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C,
                                        int Ho, int Wo) {
  const int c4 = C / 4;
  const long long total = (long long)B * Ho * Wo * c4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4);
    long long p = i / c4;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dh = 0; dh < 3; ++dh) {
      const int hi = ho * 2 - 1 + dh;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int wi = wo * 2 - 1 + dw;
        if ((unsigned)wi >= (unsigned)W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (((long long)n * H + hi) * W + wi) * C + cc * 4);
        m.x = fmaxf(m.x, v.x), m.y = fmaxf(m.y, v.y), m.z = fmaxf(m.z, v.z), m.w = fmaxf(m.w, v.w);
      }
    }
    *reinterpret_cast<float4*>(y + (((long long)n * Ho + ho) * Wo + wo) * C + cc * 4) = m;
  }
}

// GroupNorm on f32 rows: statistics accumulated in DOUBLE (sum, sum of squares per (image, level, group)), so the
// E[x^2] - mean^2 form loses nothing against the reference's f32 two-pass moments.  A thread owns one 16-byte chunk
// column (4 channels inside one group) and strides over rows.
__global__ __launch_bounds__(256) void gn_stats_f32_kernel(const float* __restrict__ x, double* __restrict__ stats,
                                                           const GnArgs a) {
  const int n = blockIdx.y;
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && (int)blockIdx.x >= a.blk0[l]) lev = l;
  const int rb = (blockIdx.x - a.blk0[lev]) * a.rpb;
  const int HW = a.hw[lev];
  const int c4 = a.C / 4;
  const int rows_per_iter = 256 / c4;
  const int cc = threadIdx.x % c4;
  const int rr = threadIdx.x / c4;
  const float* base = x + (a.row0[lev] + (long long)n * HW) * a.C;
  double s = 0.0, ss = 0.0;
  if (rr < rows_per_iter) {
    const int rend = min(rb + a.rpb, HW);
    for (int r = rb + rr; r < rend; r += rows_per_iter) {
      const float4 v = *reinterpret_cast<const float4*>(base + (long long)r * a.C + cc * 4);
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  }
  __shared__ double sh[2][256];
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = ss;
  __syncthreads();
  const int chunks_per_group = a.cpg / 4;
  if ((int)threadIdx.x < a.groups) {
    const int g = threadIdx.x;
    double ts = 0.0, tss = 0.0;
    for (int r = 0; r < rows_per_iter; ++r)
      for (int k = 0; k < chunks_per_group; ++k) {
        const int t = r * c4 + g * chunks_per_group + k;
        ts += sh[0][t];
        tss += sh[1][t];
      }
    double* st = stats + (((long long)n * a.nlev + lev) * a.groups + g) * 2;
    atomicAdd(st, ts);
    atomicAdd(st + 1, tss);
  }
}

__global__ __launch_bounds__(256) void gn_apply_f32_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const double* __restrict__ stats, const GnArgs a) {
  const int n = blockIdx.y;
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && (int)blockIdx.x >= a.blk0[l]) lev = l;
  const int rb = (blockIdx.x - a.blk0[lev]) * a.rpb;
  const int HW = a.hw[lev];
  const int c4 = a.C / 4;
  const int rows_per_iter = 256 / c4;
  const int cc = threadIdx.x % c4;
  const int rr = threadIdx.x / c4;
  if (rr >= rows_per_iter) return;
  const int g = (cc * 4) / a.cpg;
  const double* st = stats + (((long long)n * a.nlev + lev) * a.groups + g) * 2;
  const double cnt = (double)HW * (double)a.cpg;
  const double mean_d = st[0] / cnt;
  const double var_d = fmax(st[1] / cnt - mean_d * mean_d, 0.0);
  // (x - mean) * rstd * gamma + beta, as at::native group_norm evaluates it
  const float mean = (float)mean_d;
  const float rstd = (float)(1.0 / sqrt(var_d + (double)a.eps));
  float ga[4], be[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ga[e] = gamma[cc * 4 + e];
    be[e] = beta[cc * 4 + e];
  }
  const long long off = (a.row0[lev] + (long long)n * HW) * a.C;
  const int rend = min(rb + a.rpb, HW);
  for (int r = rb + rr; r < rend; r += rows_per_iter) {
    const long long o = off + (long long)r * a.C + cc * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = (f[e] - mean) * rstd * ga[e] + be[e];
      f[e] = a.relu ? fmaxf(t, 0.f) : t;
    }
    *reinterpret_cast<float4*>(y + o) = make_float4(f[0], f[1], f[2], f[3]);
  }
}

inline int grid_for(long long n, int block) {
  long long g = (n + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int sm_nchw_f32_to_nhwc_bf16(const float* x, void* y, int batch, int c, int h, int w, int cpad,
                                        sm_stream_t stream) {
  if (!x || !y || cpad % 8 != 0 || cpad < c || batch < 1) return SM_ERR_BAD_ARG;
  const long long n = (long long)batch * h * w;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(n, 256)), dim3(256), 0, sm_hip_stream(stream), x,
                     (uint16_t*)y, batch, c, h * w, cpad);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_maxpool3x3s2(const void* x, void* y, int batch, int h, int w, int c, sm_stream_t stream) {
  if (!x || !y || c % 8 != 0) return SM_ERR_BAD_ARG;
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const long long n = (long long)batch * ho * wo * (c / 8);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(n, 256)), dim3(256), 0, sm_hip_stream(stream),
                     (const uint16_t*)x, (uint16_t*)y, batch, h, w, c, ho, wo);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

static int gn_fill_args(GnArgs& a, int& t, int batch, int nlev, const int32_t* hw, const int64_t* row0, int channels,
                        int groups, float eps, int relu, int vec = 8, bool streaming = false) {
  if (nlev < 1 || nlev > SM_MAX_LEVELS || channels % (vec * groups) != 0 || channels > 256 * vec || 256 % (channels / vec) != 0)
    return SM_ERR_BAD_SHAPE;
  if (groups > 256) return SM_ERR_BAD_SHAPE;
  a.nlev = nlev;
  a.batch = batch;
  a.C = channels;
  a.groups = groups;
  a.cpg = channels / groups;
  a.eps = eps;
  a.relu = relu;
  // a block walks GN_ROWS_PER_BLOCK rows; the pure streaming pass (apply only) wants >= ~4 blocks per CU in flight,
  // so small tensors (a B=2 sub-plan: 176 blocks of 256 rows for the whole tower tensor) get shorter blocks
  a.rpb = GN_ROWS_PER_BLOCK;
  if (streaming) {
    long long blocks = 0;
    for (int l = 0; l < nlev; ++l) blocks += (long long)batch * sm_cdiv(hw[l], GN_ROWS_PER_BLOCK);
    if (blocks < 1024) a.rpb = blocks < 512 ? 32 : 64;
  }
  t = 0;
  for (int l = 0; l < SM_MAX_LEVELS; ++l) {
    a.hw[l] = l < nlev ? hw[l] : 0;
    a.row0[l] = l < nlev ? row0[l] : 0;
    a.blk0[l] = t;
    if (l < nlev) t += sm_cdiv(hw[l], a.rpb);
  }
  a.blk0[SM_MAX_LEVELS] = t;
  return SM_OK;
}

extern "C" int sm_groupnorm(const void* x, void* y, const float* gamma, const float* beta, int64_t* stats_fix, int batch,
                            int nlev, const int32_t* hw, const int64_t* row0, int channels, int groups, float eps,
                            int relu, sm_stream_t stream) {
  if (!x || !y || !gamma || !beta || !stats_fix || !hw || !row0) return SM_ERR_BAD_ARG;
  unsigned long long* stats = reinterpret_cast<unsigned long long*>(stats_fix);
  GnArgs a;
  int t;
  const int st = gn_fill_args(a, t, batch, nlev, hw, row0, channels, groups, eps, relu);
  if (st != SM_OK) return st;
  hipStream_t s = sm_hip_stream(stream);
  if (sm_zero_async(stats, sizeof(unsigned long long) * 2 * batch * nlev * groups, s) != hipSuccess) return SM_ERR_LAUNCH;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(t, batch), dim3(256), 0, s, (const uint16_t*)x, stats, a);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(t, batch), dim3(256), 0, s, (const uint16_t*)x, (uint16_t*)y, gamma, beta,
                     stats, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_groupnorm_apply(const void* x, void* y, const float* gamma, const float* beta, const int64_t* stats_fix,
                                  int batch, int nlev, const int32_t* hw, const int64_t* row0, int channels, int groups,
                                  float eps, int relu, sm_stream_t stream) {
  if (!x || !y || !gamma || !beta || !stats_fix || !hw || !row0) return SM_ERR_BAD_ARG;
  const unsigned long long* stats = reinterpret_cast<const unsigned long long*>(stats_fix);
  GnArgs a;
  int t;
  const int st = gn_fill_args(a, t, batch, nlev, hw, row0, channels, groups, eps, relu, 8, true);
  if (st != SM_OK) return st;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(t, batch), dim3(256), 0, sm_hip_stream(stream), (const uint16_t*)x,
                     (uint16_t*)y, gamma, beta, stats, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_upsample_bilinear(const void* x, void* y, int batch, int h, int w, int c, int factor,
                                    int in_cstride, int out_cstride, int out_coff, int is_f32, sm_stream_t stream) {
  if (!x || !y || factor < 1) return SM_ERR_BAD_ARG;
  const int vec = is_f32 ? 4 : 8;
  if (c % vec || in_cstride % vec || out_cstride % vec || out_coff % vec) return SM_ERR_BAD_SHAPE;
  const long long n = (long long)batch * h * factor * w * factor * (c / vec);
  if (is_f32)
    hipLaunchKernelGGL(upsample_bilinear_kernel<true>, dim3(grid_for(n, 256)), dim3(256), 0, sm_hip_stream(stream), x,
                       y, batch, h, w, c, factor, in_cstride, out_cstride, out_coff);
  else
    hipLaunchKernelGGL(upsample_bilinear_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, sm_hip_stream(stream), x,
                       y, batch, h, w, c, factor, in_cstride, out_cstride, out_coff);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_offset_linear(const float* reg, int reg_cstride, const float* w_off, int nout, const int64_t* row0,
                                const int32_t* rows_per_level, const float* level_scale, int nlev, float* out,
                                sm_stream_t stream) {
  if (!reg || !w_off || !row0 || !rows_per_level || !out || nlev < 1 || nlev > SM_MAX_LEVELS) return SM_ERR_BAD_ARG;
  if (reg_cstride % 4 != 0) return SM_ERR_BAD_SHAPE;
  OffArgs a;
  a.nlev = nlev;
  a.nout = nout;
  a.reg_cs = reg_cstride;
  a.total = 0;
  for (int l = 0; l < SM_MAX_LEVELS; ++l) {
    a.row0[l] = l < nlev ? row0[l] : 0;
    a.rows[l] = l < nlev ? rows_per_level[l] : 0;
    a.scale[l] = (l < nlev && level_scale) ? level_scale[l] : 1.f;
    if (l < nlev) a.total += rows_per_level[l];
  }
  hipLaunchKernelGGL(offset_linear_kernel, dim3(grid_for(a.total * nout, 256)), dim3(256), 0, sm_hip_stream(stream),
                     reg, w_off, out, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int64_t sm_offset_linear_bwd_workspace(int64_t rows, int nout) {
  return (int64_t)((rows + OFFB_ROWS - 1) / OFFB_ROWS) * nout * 4 * (int64_t)sizeof(float);
}

extern "C" int sm_offset_linear_bwd(const float* reg, int reg_cstride, const float* grad_out, int nout, int64_t rows,
                                    float* workspace, float* grad_w, sm_stream_t stream) {
  if (!reg || !grad_out || !workspace || !grad_w || rows < 1) return SM_ERR_BAD_ARG;
  if (reg_cstride % 4 != 0 || nout < 1 || nout > 256) return SM_ERR_BAD_SHAPE;
  const int nblk = (int)((rows + OFFB_ROWS - 1) / OFFB_ROWS);
  hipLaunchKernelGGL(offset_linear_bwd_partial_kernel, dim3(nblk), dim3(4 * nout), sizeof(float) * 16 * nout,
                     sm_hip_stream(stream), reg, reg_cstride, grad_out, nout, (long long)rows, workspace);
  SM_LAUNCH_CHECK();
  hipLaunchKernelGGL(offset_linear_bwd_final_kernel, dim3(sm_cdiv(nout * 4, 4)), dim3(256), 0, sm_hip_stream(stream),
                     workspace, nblk, nout * 4, grad_w);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

/* ReLU of a bf16 row matrix into a second buffer.  FPN: P7 = conv(relu(P6)) (fpn.py:166-170) while P6 itself is a
   pyramid level; with the ReLU'd copy as the conv's input the 4-tile P7 launch takes the LDS-DMA operand path instead
   of the register-staged loader the SM_CONV_IN_RELU flag forces (0.056 -> 0.02 ms). */
namespace {
__global__ void relu_bf16_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, long long n16) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n16) return;
  u32x4 v = x[i];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t neg = (v[e] >> 15) & 0x00010001u;      // sign bit of each half
    v[e] &= ~(neg * 0xffffu);                              // negative (and -0) halves -> +0
  }
  y[i] = v;
}
}  // namespace

extern "C" int sm_relu_bf16(const void* x, void* y, int64_t n, sm_stream_t stream) {
  if (!x || !y || n < 0 || (n & 7) != 0) return SM_ERR_BAD_ARG;
  if (n == 0) return SM_OK;
  hipLaunchKernelGGL(relu_bf16_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, sm_hip_stream(stream),
                     (const u32x4*)x, (u32x4*)y, (long long)(n / 8));
  SM_LAUNCH_CHECK();
  return SM_OK;
}

/* ---- results to the host without the runtime's asynchronous copy path (round 6) ------------------------------------
 * Up to SM_COPY_MAX_SEGS (source, destination, bytes) segments copied by ONE kernel launch; the destinations are
 * pinned (hipHostMalloc) host buffers, which the GPU addresses directly.  What it replaces: six hipMemcpyAsync
 * device -> pinned-host calls behind every step (boxes, labels, counts, run counts, string offsets, string prefix): one host
 * call on the submit path of a pipelined step instead of six.  A kernel's stores to coherent host memory are visible to the
 * host once the launch has completed (the event the caller records behind it). */
namespace {
struct CopySegs {
  const unsigned char* src[SM_COPY_MAX_SEGS];
  unsigned char* dst[SM_COPY_MAX_SEGS];
  long long bytes[SM_COPY_MAX_SEGS];
  int nseg;
};

__global__ __launch_bounds__(256) void copy_segments_kernel(const CopySegs a) {
  for (int sgi = blockIdx.y; sgi < a.nseg; sgi += gridDim.y) {
    const unsigned char* s = a.src[sgi];
    unsigned char* d = a.dst[sgi];
    const long long n = a.bytes[sgi];
    const bool al = ((((unsigned long long)s) | ((unsigned long long)d)) & 15ull) == 0ull;
    const long long nv = al ? (n >> 4) : 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x)
      reinterpret_cast<u32x4*>(d)[i] = reinterpret_cast<const u32x4*>(s)[i];
    for (long long i = (nv << 4) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      d[i] = s[i];
  }
}
}  // namespace

extern "C" int sm_copy_segments(int nseg, const void* const* src, void* const* dst, const int64_t* bytes, sm_stream_t stream) {
  if (nseg < 1 || nseg > SM_COPY_MAX_SEGS || !src || !dst || !bytes) return SM_ERR_BAD_ARG;
  CopySegs a;
  long long most = 0;
  for (int i = 0; i < SM_COPY_MAX_SEGS; ++i) {
    const bool on = i < nseg;
    if (on && (bytes[i] < 0 || (bytes[i] > 0 && (!src[i] || !dst[i])))) return SM_ERR_BAD_ARG;
    a.src[i] = on ? (const unsigned char*)src[i] : nullptr;
    a.dst[i] = on ? (unsigned char*)dst[i] : nullptr;
    a.bytes[i] = on ? bytes[i] : 0;
    if (on && bytes[i] > most) most = bytes[i];
  }
  a.nseg = nseg;
  if (most == 0) return SM_OK;
  long long gx = (most / 16 + 255) / 256;            // one 16-byte word per thread and pass for the longest segment ...
  if (gx < 1) gx = 1;
  if (gx > 64) gx = 64;                              // ... on at most 64 blocks per segment (a 256-KB prefix: 4 passes)
  hipLaunchKernelGGL(copy_segments_kernel, dim3((unsigned)gx, (unsigned)nseg), dim3(256), 0, sm_hip_stream(stream), a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

/* ---- exact-f32 plan (parity mode): the f32 twins of the layout / pool / GroupNorm kernels ------------------- */

extern "C" int sm_nchw_f32_to_nhwc_f32(const float* x, float* y, int batch, int c, int h, int w, int cpad,
                                       sm_stream_t stream) {
  if (!x || !y || cpad % 4 != 0 || cpad < c || batch < 1) return SM_ERR_BAD_ARG;
  const long long n = (long long)batch * h * w;
  hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, sm_hip_stream(stream), x, y, batch, c,
                     h * w, cpad);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_maxpool3x3s2_f32(const float* x, float* y, int batch, int h, int w, int c, sm_stream_t stream) {
  if (!x || !y || c % 4 != 0) return SM_ERR_BAD_ARG;
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const long long n = (long long)batch * ho * wo * (c / 4);
  hipLaunchKernelGGL(maxpool3x3s2_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, sm_hip_stream(stream), x, y, batch, h,
                     w, c, ho, wo);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_groupnorm_f32(const float* x, float* y, const float* gamma, const float* beta, double* stats, int batch,
                                int nlev, const int32_t* hw, const int64_t* row0, int channels, int groups, float eps,
                                int relu, sm_stream_t stream) {
  if (!x || !y || !gamma || !beta || !stats || !hw || !row0) return SM_ERR_BAD_ARG;
  GnArgs a;
  int t;
  const int st = gn_fill_args(a, t, batch, nlev, hw, row0, channels, groups, eps, relu, 4);
  if (st != SM_OK) return st;
  hipStream_t s = sm_hip_stream(stream);
  if (sm_zero_async(stats, sizeof(double) * 2 * batch * nlev * groups, s) != hipSuccess) return SM_ERR_LAUNCH;
  hipLaunchKernelGGL(gn_stats_f32_kernel, dim3(t, batch), dim3(256), 0, s, x, stats, a);
  hipLaunchKernelGGL(gn_apply_f32_kernel, dim3(t, batch), dim3(256), 0, s, x, y, gamma, beta, stats, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
