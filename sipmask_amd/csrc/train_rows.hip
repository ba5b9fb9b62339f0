// Training step on NHWC bf16 ROW tensors (gfx950): the small kernels that let the autograd ops of the training graph
// (sipmask_amd/ops_rows.py) hand each other [positions][channels] bf16 matrices -- the layout the MFMA conv kernels
// read and write -- without NCHW<->NHWC transposes, f32 round trips or chains of ATen launches in between.
//
//   sm_weight_prep        f32 OIHW parameter (x per-cout scale = frozen-BatchNorm fold) -> the bf16 operand layouts of
//                         sm_conv2d / sm_conv2d_bwd (forward, flipped+transposed for dX, K-major for the column path)
//   sm_wgrad_finish       dW^T f32 [K][cout] -> OIHW f32 (x the same scale): the gradient autograd expects
//   sm_relu_bwd_bf16      g * (y > 0)
//   sm_bias_grad_rows     column sums of a bf16 row matrix
//   sm_gn_bwd_rows        GroupNorm (+ReLU) backward over pyramid rows, statistics per (image, level, group)
//   sm_upsample_bilinear_bwd_rows   adjoint of sm_upsample_bilinear (gather form, reads a channel slice of the gradient)
//   sm_nearest_bwd_rows   adjoint of the SM_CONV_RES_NEAREST residual (FPN top-down add)
//   sm_scatter_stride_rows  dX of a strided 1x1 conv: rows of the strided grid scattered into the zeroed input grid
// Reference counterparts: ATen's conv / group_norm / upsample / threshold backward kernels under
// M/mmdet/models/{backbones/resnet.py,necks/fpn.py,anchor_heads/sipmask_head.py} in training mode.
#include "common.h"

namespace {

inline int grid_for(long long n, int block) {
  long long g = (n + block - 1) / block;
  if (g > 1 << 20) g = 1 << 20;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------------------------------------ weight layouts
// out is [rows_pad][kp] bf16, zeroed by the host before the launch (padding rows / K padding / channel padding).
//  mode 0: rows = cout,  k = (r*kw + s)*cin_pad + c            value w[o][c][r][s]
//  mode 1: rows = cin,   k = (r*kw + s)*cout + o               value w[o][c][kh-1-r][kw-1-s]     (dX as a forward conv)
//  mode 2: rows = (r*kw + s)*cin + c,  k = o                   value w[o][c][r][s]               (grad-column GEMM)
// A block owns a tile of 32 couts x 32 cins (all taps): it reads the 32 rows of w coalesced (the (c, tap) axis is
// contiguous in OIHW) into LDS and writes 16-byte pieces of 8 consecutive channels (mode 0) / couts (modes 1, 2).
constexpr int WP_T = 32;
__device__ __forceinline__ void weight_prep_tile(const float* __restrict__ w, const float* __restrict__ scale,
                                                 uint16_t* __restrict__ out, int co, int ci, int kh, int kw, int mode,
                                                 int kp, int cin_pad, int tc, int bx, int by, float* tile) {
  // tile: [32 o][tc c * kk + 1]; tc = channels per tile (32, or 8 for big kernels)
  const int kk = kh * kw;
  const int o0 = by * WP_T, c0 = bx * tc;
  const int cw = min(tc, ci - c0);                   // live channels of the tile
  const int pitch = tc * kk + 1;
  const int nch = tc >> 3;                           // 8-wide pieces along the channel / cout axis of a tile
  const int ncol = cw * kk;
  for (int i = threadIdx.x; i < WP_T * ncol; i += 256) {
    const int r = i / ncol, col = i - r * ncol;
    const int o = o0 + r;
    tile[r * pitch + col] = o < co ? w[((long long)o * ci + c0) * kk + col] * (scale ? scale[o] : 1.f) : 0.f;
  }
  __syncthreads();
  if (mode == 0) {
    // pieces: (o in tile, tap, 8-channel chunk)
    for (int i = threadIdx.x; i < WP_T * kk * nch; i += 256) {
      const int ch = i % nch, t = (i / nch) % kk, r = (i / nch) / kk;
      const int o = o0 + r, c = c0 + ch * 8;
      if (o >= co || c >= cin_pad) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (ch * 8 + e < cw) ? tile[r * pitch + (ch * 8 + e) * kk + t] : 0.f;
      *reinterpret_cast<uint4*>(out + (long long)o * kp + (long long)t * cin_pad + c) = pack_bf16x8(v);
    }
  } else {
    // pieces: (c in tile, tap, 8-cout chunk)
    for (int i = threadIdx.x; i < tc * kk * 4; i += 256) {
      const int ch = i & 3, t = (i >> 2) % kk, cl = (i >> 2) / kk;
      const int c = c0 + cl, o = o0 + ch * 8;
      if (cl >= cw || o >= co) continue;
      const int ts = mode == 1 ? kk - 1 - t : t;     // source tap (flipped for the dX operand)
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[(ch * 8 + e) * pitch + cl * kk + ts];
      const long long at = mode == 1 ? (long long)c * kp + (long long)t * co + o : ((long long)t * ci + c) * kp + o;
      *reinterpret_cast<uint4*>(out + at) = pack_bf16x8(v);
    }
  }
}

__global__ __launch_bounds__(256) void weight_prep_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                          uint16_t* __restrict__ out, int co, int ci, int kh, int kw, int mode,
                                                          int kp, int cin_pad, int tc) {
  extern __shared__ float wp_tile[];
  weight_prep_tile(w, scale, out, co, ci, kh, kw, mode, kp, cin_pad, tc, blockIdx.x, blockIdx.y, wp_tile);
}

// every conv of a training step in ONE launch: item table on the device, (item, tile) list per block
struct WPItem {
  const float* w;
  const float* scale;
  uint16_t* out;
  int co, ci, kh, kw, mode, kp, cin_pad, tc, tiles_x, pad_;
};
__global__ __launch_bounds__(256) void weight_prep_multi_kernel(const WPItem* __restrict__ items, const int2* __restrict__ blocks) {
  extern __shared__ float wp_tile[];
  const int2 bk = blocks[blockIdx.x];
  const WPItem it = items[bk.x];
  weight_prep_tile(it.w, it.scale, it.out, it.co, it.ci, it.kh, it.kw, it.mode, it.kp, it.cin_pad, it.tc, bk.y % it.tiles_x,
                   bk.y / it.tiles_x, wp_tile);
}

// gw_t [kh*kw*ci][co] f32 -> out [co][ci][kh][kw] f32 (x scale[o]); thread per output element, reads are strided by co
// (the matrices are <= 2304 x 2048: this is a few hundred KB .. 19 MB once per conv per step)
__global__ void wgrad_finish_kernel(const float* __restrict__ gw_t, const float* __restrict__ scale, float* __restrict__ out,
                                    int co, int ci, int kh, int kw) {
  // tile 32 (o) x 32 (k = tap*ci + c) through LDS so both sides move 128-byte lines
  __shared__ float t[32][33];
  const int K = kh * kw * ci;
  const int k0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 8 rows per pass
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, o = o0 + tx;
    t[r][tx] = (k < K && o < co) ? gw_t[(long long)k * co + o] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int o = o0 + r, k = k0 + tx;
    if (o < co && k < K) {
      const int tap = k / ci, c = k - tap * ci;
      out[((long long)o * ci + c) * (kh * kw) + tap] = t[tx][r] * (scale ? scale[o] : 1.f);
    }
  }
}

// ------------------------------------------------------------------------------------------------ elementwise
__global__ void relu_bwd_kernel(const uint4* __restrict__ g, const uint4* __restrict__ y, uint4* __restrict__ out,
                                long long n16) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n16) return;
  const uint4 gv = g[i], yv = y[i];
  // y is a ReLU output: y > 0 <=> its bf16 bits are neither +0 / -0 nor negative
  auto mask = [](uint32_t gg, uint32_t yy) -> uint32_t {
    const uint32_t lo = yy & 0xffffu, hi = yy >> 16;
    const uint32_t mlo = (lo != 0u && lo < 0x8000u) ? 0x0000ffffu : 0u;
    const uint32_t mhi = (hi != 0u && hi < 0x8000u) ? 0xffff0000u : 0u;
    return gg & (mlo | mhi);
  };
  uint4 r;
  r.x = mask(gv.x, yv.x), r.y = mask(gv.y, yv.y), r.z = mask(gv.z, yv.z), r.w = mask(gv.w, yv.w);
  out[i] = r;
}

// column sums of g [rows][cstride] (first c channels): thread = one 8-channel chunk, strides over the block's rows
constexpr int BG_ROWS = 512;
__global__ __launch_bounds__(256) void bias_grad_rows_kernel(const uint16_t* __restrict__ g, float* __restrict__ out,
                                                             long long rows, int cstride, int c) {
  const int c8 = (c + 7) >> 3;                       // chunks per row (c % 8 == 0 is checked by the host)
  const int lanes = 256 / c8 > 0 ? 256 / c8 : 1;     // row lanes per block
  const int cc = threadIdx.x % c8, rr = threadIdx.x / c8;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long r0 = (long long)blockIdx.x * BG_ROWS;
  const long long r1 = r0 + BG_ROWS < rows ? r0 + BG_ROWS : rows;
  if (rr < lanes) {
    for (long long r = r0 + rr; r < r1; r += lanes) {
      float f[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(g + r * cstride + cc * 8), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += f[e];
    }
  }
  __shared__ float sh[256][9];
#pragma unroll
  for (int e = 0; e < 8; ++e) sh[threadIdx.x][e] = s[e];
  __syncthreads();
  if ((int)threadIdx.x < c8 * 8) {                   // one thread per channel sums the row lanes
    const int ch = threadIdx.x, chunk = ch >> 3, e = ch & 7;
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += sh[l * c8 + chunk][e];
    atomicAdd(out + ch, t);
  }
}

// ------------------------------------------------------------------------------------------------ GroupNorm backward
struct GnbArgs {
  int nlev, batch, C, groups, cpg;
  int hw[SM_MAX_LEVELS];
  long long row0[SM_MAX_LEVELS];
  int blk0[SM_MAX_LEVELS + 1];
  float eps;
  int relu;
  int rpb;
};

// pass 1: per (image, level, group) a1 = sum dy*gamma, a2 = sum dy*gamma*xhat; per channel dgamma = sum dy*xhat,
// dbeta = sum dy (dy already gated by the ReLU).  Thread = one 8-channel chunk column of the block's rows.
__global__ __launch_bounds__(256) void gn_bwd_rows_reduce_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const unsigned long long* __restrict__ stats, float* __restrict__ bins,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 const GnbArgs a) {
  const int n = blockIdx.y;
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && (int)blockIdx.x >= a.blk0[l]) lev = l;
  const int rb = (blockIdx.x - a.blk0[lev]) * a.rpb;
  const int HW = a.hw[lev];
  const int c8 = a.C / 8;
  const int lanes = 256 / c8;
  const int cc = threadIdx.x % c8, rr = threadIdx.x / c8;
  const int g = (cc * 8) / a.cpg;
  float mean, rstd;      // the forward's fixed-point statistics (common.h: gn_fix), the forward's own arithmetic
  gn_mean_rstd(stats + (((long long)n * a.nlev + lev) * a.groups + g) * 2, (double)HW * (double)a.cpg, a.eps, &mean, &rstd);
  float ga[8], be[8], sdy[8], sdyx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ga[e] = gamma[cc * 8 + e];
    be[e] = beta[cc * 8 + e];
    sdy[e] = 0.f;
    sdyx[e] = 0.f;
  }
  const long long off = (a.row0[lev] + (long long)n * HW) * a.C;
  const int rend = min(rb + a.rpb, HW);
  if (rr < lanes) {
    for (int r = rb + rr; r < rend; r += lanes) {
      float xv[8], dv[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(x + off + (long long)r * a.C + cc * 8), xv);
      unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + off + (long long)r * a.C + cc * 8), dv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (xv[e] - mean) * rstd;
        const float d = (a.relu && ga[e] * xh + be[e] <= 0.f) ? 0.f : dv[e];
        sdy[e] += d;
        sdyx[e] += d * xh;
      }
    }
  }
  __shared__ float sh[256][17];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sh[threadIdx.x][e] = sdy[e];
    sh[threadIdx.x][8 + e] = sdyx[e];
  }
  __syncthreads();
  // one thread per channel: sum over row lanes, publish dgamma / dbeta, keep gamma-weighted values for the group sums
  float a1 = 0.f, a2 = 0.f;
  if ((int)threadIdx.x < a.C) {
    const int ch = threadIdx.x, chunk = ch >> 3, e = ch & 7;
    float t1 = 0.f, t2 = 0.f;
    for (int l = 0; l < lanes; ++l) {
      t1 += sh[l * c8 + chunk][e];
      t2 += sh[l * c8 + chunk][8 + e];
    }
    atomicAdd(dbeta + ch, t1);
    atomicAdd(dgamma + ch, t2);
    const float gm = gamma[ch];
    a1 = t1 * gm;
    a2 = t2 * gm;
  }
  __syncthreads();
  sh[threadIdx.x][0] = a1;
  sh[threadIdx.x][1] = a2;
  __syncthreads();
  if ((int)threadIdx.x < a.groups) {
    float t1 = 0.f, t2 = 0.f;
    for (int k = 0; k < a.cpg; ++k) {
      t1 += sh[threadIdx.x * a.cpg + k][0];
      t2 += sh[threadIdx.x * a.cpg + k][1];
    }
    float* b = bins + (((long long)n * a.nlev + lev) * a.groups + threadIdx.x) * 2;
    atomicAdd(b, t1);
    atomicAdd(b + 1, t2);
  }
}

// pass 2: dx = rstd * (dy*gamma - a1/N - xhat * a2/N)
__global__ __launch_bounds__(256) void gn_bwd_rows_apply_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const unsigned long long* __restrict__ stats, const float* __restrict__ bins,
                                                                uint16_t* __restrict__ dx, const GnbArgs a) {
  const int n = blockIdx.y;
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && (int)blockIdx.x >= a.blk0[l]) lev = l;
  const int rb = (blockIdx.x - a.blk0[lev]) * a.rpb;
  const int HW = a.hw[lev];
  const int c8 = a.C / 8;
  const int lanes = 256 / c8;
  const int cc = threadIdx.x % c8, rr = threadIdx.x / c8;
  if (rr >= lanes) return;
  const int g = (cc * 8) / a.cpg;
  const long long sidx = (((long long)n * a.nlev + lev) * a.groups + g) * 2;
  const float cnt = (float)HW * (float)a.cpg;
  float mean, rstd;
  gn_mean_rstd(stats + sidx, (double)HW * (double)a.cpg, a.eps, &mean, &rstd);
  const float m1 = bins[sidx] / cnt, m2 = bins[sidx + 1] / cnt;
  float ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ga[e] = gamma[cc * 8 + e];
    be[e] = beta[cc * 8 + e];
  }
  const long long off = (a.row0[lev] + (long long)n * HW) * a.C;
  const int rend = min(rb + a.rpb, HW);
  for (int r = rb + rr; r < rend; r += lanes) {
    float xv[8], dv[8], o[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(x + off + (long long)r * a.C + cc * 8), xv);
    unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + off + (long long)r * a.C + cc * 8), dv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (xv[e] - mean) * rstd;
      const float d = (a.relu && ga[e] * xh + be[e] <= 0.f) ? 0.f : dv[e];
      o[e] = rstd * (d * ga[e] - m1 - xh * m2);
    }
    *reinterpret_cast<uint4*>(dx + off + (long long)r * a.C + cc * 8) = pack_bf16x8(o);
  }
}

// ------------------------------------------------------------------------------------------------ resampling adjoints
// gin[n][y][x][c] = sum over output pixels (ho, wo) of the bilinear weight they gave input pixel (y, x) * gout
__global__ void upsample_bwd_rows_kernel(const uint16_t* __restrict__ gout, uint16_t* __restrict__ gin, int B, int H, int W,
                                         int C, int factor, int out_cs, int out_coff) {
  const int Ho = H * factor, Wo = W * factor;
  const int cv = C / 8;
  const long long total = (long long)B * H * W * cv;
  const float inv = 1.f / (float)factor;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cv);
    long long p = i / cv;
    const int xi = (int)(p % W);
    p /= W;
    const int yi = (int)(p % H);
    const int n = (int)(p / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int ho0 = max((yi - 1) * factor, 0), ho1 = min((yi + 2) * factor, Ho);
    const int wo0 = max((xi - 1) * factor, 0), wo1 = min((xi + 2) * factor, Wo);
    for (int ho = ho0; ho < ho1; ++ho) {
      const float sy = fmaxf(((float)ho + 0.5f) * inv - 0.5f, 0.f);
      const int y0 = (int)sy, y1 = min(y0 + 1, H - 1);
      const float ly = sy - (float)y0;
      const float wy = (y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int wo = wo0; wo < wo1; ++wo) {
        const float sx = fmaxf(((float)wo + 0.5f) * inv - 0.5f, 0.f);
        const int x0 = (int)sx, x1 = min(x0 + 1, W - 1);
        const float lx = sx - (float)x0;
        const float wx = (x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f);
        if (wx == 0.f) continue;
        float f[8];
        const long long orow = ((long long)n * Ho + ho) * Wo + wo;
        unpack_bf16x8(*reinterpret_cast<const uint4*>(gout + orow * out_cs + out_coff + cc * 8), f);
        const float wgt = wy * wx;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += wgt * f[e];
      }
    }
    *reinterpret_cast<uint4*>(gin + (((long long)n * H + yi) * W + xi) * C + cc * 8) = pack_bf16x8(acc);
  }
}

// coarse[n][sh][sw] = sum of fine[n][ho][wo] over the fine pixels whose nearest source (conv_igemm.hip's
// SM_CONV_RES_NEAREST rule: min(floor(ho * rh / Ho), rh - 1)) is (sh, sw)
__global__ void nearest_bwd_rows_kernel(const uint16_t* __restrict__ gf, uint16_t* __restrict__ gc, int B, int Ho, int Wo,
                                        int rh, int rw, int C) {
  const int cv = C / 8;
  const long long total = (long long)B * rh * rw * cv;
  const float fy = (float)rh / (float)Ho, fx = (float)rw / (float)Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cv);
    long long p = i / cv;
    const int sw = (int)(p % rw);
    p /= rw;
    const int sh = (int)(p % rh);
    const int n = (int)(p / rh);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int h0 = max((int)floorf((float)sh / fy) - 1, 0), h1 = min((int)ceilf((float)(sh + 1) / fy) + 1, Ho);
    const int w0 = max((int)floorf((float)sw / fx) - 1, 0), w1 = min((int)ceilf((float)(sw + 1) / fx) + 1, Wo);
    for (int ho = h0; ho < h1; ++ho) {
      if (min((int)floorf((float)ho * fy), rh - 1) != sh) continue;
      for (int wo = w0; wo < w1; ++wo) {
        if (min((int)floorf((float)wo * fx), rw - 1) != sw) continue;
        float f[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(gf + (((long long)n * Ho + ho) * Wo + wo) * C + cc * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f[e];
      }
    }
    *reinterpret_cast<uint4*>(gc + (((long long)n * rh + sh) * rw + sw) * C + cc * 8) = pack_bf16x8(acc);
  }
}

// out[n][y][x] = (y % s == 0 && x % s == 0 && y/s < ho && x/s < wo) ? in[n][y/s][x/s] : 0      (16-byte chunks)
__global__ void scatter_stride_rows_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int B, int H, int W,
                                           int ho, int wo, int stride, int c8) {
  const long long total = (long long)B * H * W * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    long long p = i / c8;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int n = (int)(p / H);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (y % stride == 0 && x % stride == 0 && y / stride < ho && x / stride < wo)
      v = in[(((long long)n * ho + y / stride) * wo + x / stride) * c8 + cc];
    out[i] = v;
  }
}

int gnb_fill(GnbArgs& a, int& t, int batch, int nlev, const int32_t* hw, const int64_t* row0, int channels, int groups,
             float eps, int relu) {
  if (nlev < 1 || nlev > SM_MAX_LEVELS || batch < 1) return SM_ERR_BAD_SHAPE;
  if (channels % 8 != 0 || channels > 256 || groups < 1 || groups > 256 || channels % groups != 0) return SM_ERR_UNSUPPORTED;
  const int cpg = channels / groups;
  if (cpg % 8 != 0 && 8 % cpg != 0) return SM_ERR_UNSUPPORTED;       // a thread's 8 channels lie in one group
  if (cpg < 8) return SM_ERR_UNSUPPORTED;
  if (256 % (channels / 8) != 0) return SM_ERR_UNSUPPORTED;
  a.nlev = nlev, a.batch = batch, a.C = channels, a.groups = groups, a.cpg = cpg, a.eps = eps, a.relu = relu;
  a.rpb = 128;
  t = 0;
  for (int l = 0; l < SM_MAX_LEVELS; ++l) {
    a.hw[l] = l < nlev ? hw[l] : 1;
    a.row0[l] = l < nlev ? row0[l] : 0;
    a.blk0[l] = t;
    if (l < nlev) t += (hw[l] + a.rpb - 1) / a.rpb;
  }
  a.blk0[SM_MAX_LEVELS] = t;
  return SM_OK;
}

}  // namespace

extern "C" int sm_weight_prep(const float* w, const float* scale, int cout, int cin, int kh, int kw, int mode, void* out,
                              int rows_pad, int kp, int cin_pad, sm_stream_t stream) {
  if (!w || !out || cout < 1 || cin < 1 || kh < 1 || kw < 1 || mode < 0 || mode > 2) return SM_ERR_BAD_ARG;
  if (kp % 8 != 0 || rows_pad < 1) return SM_ERR_BAD_SHAPE;
  const long long klog = mode == 0 ? (long long)kh * kw * cin_pad : (mode == 1 ? (long long)kh * kw * cout : cout);
  const long long rlog = mode == 0 ? cout : (mode == 1 ? cin : (long long)kh * kw * cin);
  if (klog > kp || rlog > rows_pad || (mode == 0 && cin_pad < cin)) return SM_ERR_BAD_SHAPE;
  if (cout % 8 != 0 && mode != 0) return SM_ERR_UNSUPPORTED;        // 16-byte pieces along cout
  if (mode == 0 && cin_pad % 8 != 0) return SM_ERR_BAD_SHAPE;
  hipStream_t s = sm_hip_stream(stream);
  if (sm_zero_async(out, (size_t)rows_pad * kp * 2, s) != hipSuccess) return SM_ERR_LAUNCH;
  const int tc = kh * kw <= 9 ? 32 : 8;               // channels per tile: the LDS tile is 32 couts x tc x (kh*kw) floats
  const size_t lds = sizeof(float) * WP_T * (tc * kh * kw + 1);
  if (lds > 64 * 1024) return SM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(weight_prep_kernel, dim3((cin + tc - 1) / tc, (cout + WP_T - 1) / WP_T), dim3(256), lds, s, w, scale,
                     (uint16_t*)out, cout, cin, kh, kw, mode, kp, cin_pad, tc);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_weight_prep_multi(const void* items, const int32_t* blocks, int nblocks, int lds_bytes, sm_stream_t stream) {
  if (!items || !blocks || lds_bytes < 0 || lds_bytes > 64 * 1024) return SM_ERR_BAD_ARG;
  if (nblocks < 1) return SM_OK;
  hipLaunchKernelGGL(weight_prep_multi_kernel, dim3(nblocks), dim3(256), (size_t)lds_bytes, sm_hip_stream(stream),
                     (const WPItem*)items, (const int2*)blocks);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_wgrad_finish(const float* grad_w_t, const float* scale, int cout, int cin, int kh, int kw, float* out,
                               sm_stream_t stream) {
  if (!grad_w_t || !out || cout < 1 || cin < 1 || kh < 1 || kw < 1) return SM_ERR_BAD_ARG;
  const int K = kh * kw * cin;
  hipLaunchKernelGGL(wgrad_finish_kernel, dim3((K + 31) / 32, (cout + 31) / 32), dim3(256), 0, sm_hip_stream(stream),
                     grad_w_t, scale, out, cout, cin, kh, kw);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_relu_bwd_bf16(const void* g, const void* y, void* out, int64_t n, sm_stream_t stream) {
  if (!g || !y || !out || n < 0 || (n & 7)) return SM_ERR_BAD_ARG;
  if (n == 0) return SM_OK;
  const long long n16 = n / 8;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, sm_hip_stream(stream),
                     (const uint4*)g, (const uint4*)y, (uint4*)out, n16);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_bias_grad_rows(const void* g, int64_t rows, int cstride, int channels, float* out, sm_stream_t stream) {
  if (!g || !out || rows < 0) return SM_ERR_BAD_ARG;
  if (channels < 8 || channels % 8 != 0 || channels > 256 || cstride % 8 != 0 || cstride < channels) return SM_ERR_UNSUPPORTED;
  hipStream_t s = sm_hip_stream(stream);
  if (sm_zero_async(out, sizeof(float) * channels, s) != hipSuccess) return SM_ERR_LAUNCH;
  if (rows == 0) return SM_OK;
  hipLaunchKernelGGL(bias_grad_rows_kernel, dim3((unsigned)((rows + BG_ROWS - 1) / BG_ROWS)), dim3(256), 0, s,
                     (const uint16_t*)g, out, (long long)rows, cstride, channels);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_gn_bwd_rows(const void* x, const void* dy, const float* gamma, const float* beta, const int64_t* stats_fix,
                              int batch, int nlev, const int32_t* hw, const int64_t* row0, int channels, int groups,
                              float eps, int relu, void* dx, float* dgamma, float* dbeta, float* bins,
                              sm_stream_t stream) {
  if (!x || !dy || !gamma || !beta || !stats_fix || !hw || !row0 || !dx || !dgamma || !dbeta || !bins) return SM_ERR_BAD_ARG;
  const unsigned long long* stats = reinterpret_cast<const unsigned long long*>(stats_fix);
  GnbArgs a;
  int t;
  const int st = gnb_fill(a, t, batch, nlev, hw, row0, channels, groups, eps, relu);
  if (st != SM_OK) return st;
  hipStream_t s = sm_hip_stream(stream);
  if (sm_zero_async(bins, sizeof(float) * 2 * batch * nlev * groups, s) != hipSuccess) return SM_ERR_LAUNCH;
  if (sm_zero_async(dgamma, sizeof(float) * channels, s) != hipSuccess) return SM_ERR_LAUNCH;
  if (sm_zero_async(dbeta, sizeof(float) * channels, s) != hipSuccess) return SM_ERR_LAUNCH;
  hipLaunchKernelGGL(gn_bwd_rows_reduce_kernel, dim3(t, batch), dim3(256), 0, s, (const uint16_t*)x, (const uint16_t*)dy,
                     gamma, beta, stats, bins, dgamma, dbeta, a);
  hipLaunchKernelGGL(gn_bwd_rows_apply_kernel, dim3(t, batch), dim3(256), 0, s, (const uint16_t*)x, (const uint16_t*)dy,
                     gamma, beta, stats, bins, (uint16_t*)dx, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_upsample_bilinear_bwd_rows(const void* gout, int out_cstride, int out_coff, int batch, int h, int w, int c,
                                             int factor, void* gin, sm_stream_t stream) {
  if (!gout || !gin || factor < 1 || batch < 1 || h < 1 || w < 1) return SM_ERR_BAD_ARG;
  if (c % 8 || out_cstride % 8 || out_coff % 8) return SM_ERR_BAD_SHAPE;
  const long long n = (long long)batch * h * w * (c / 8);
  hipLaunchKernelGGL(upsample_bwd_rows_kernel, dim3(grid_for(n, 256)), dim3(256), 0, sm_hip_stream(stream),
                     (const uint16_t*)gout, (uint16_t*)gin, batch, h, w, c, factor, out_cstride, out_coff);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_nearest_bwd_rows(const void* g_fine, int batch, int fine_h, int fine_w, int coarse_h, int coarse_w, int c,
                                   void* g_coarse, sm_stream_t stream) {
  if (!g_fine || !g_coarse || batch < 1 || fine_h < 1 || fine_w < 1 || coarse_h < 1 || coarse_w < 1) return SM_ERR_BAD_ARG;
  if (c % 8) return SM_ERR_BAD_SHAPE;
  const long long n = (long long)batch * coarse_h * coarse_w * (c / 8);
  hipLaunchKernelGGL(nearest_bwd_rows_kernel, dim3(grid_for(n, 256)), dim3(256), 0, sm_hip_stream(stream),
                     (const uint16_t*)g_fine, (uint16_t*)g_coarse, batch, fine_h, fine_w, coarse_h, coarse_w, c);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_scatter_stride_rows(const void* in, int batch, int h, int w, int out_h, int out_w, int stride, int c,
                                      void* out, sm_stream_t stream) {
  if (!in || !out || batch < 1 || h < 1 || w < 1 || stride < 1) return SM_ERR_BAD_ARG;
  if (c % 8) return SM_ERR_BAD_SHAPE;
  if ((h - 1) / stride + 1 != out_h || (w - 1) / stride + 1 != out_w) return SM_ERR_BAD_SHAPE;
  const long long n = (long long)batch * h * w * (c / 8);
  hipLaunchKernelGGL(scatter_stride_rows_kernel, dim3(grid_for(n, 256)), dim3(256), 0, sm_hip_stream(stream),
                     (const uint4*)in, (uint4*)out, batch, h, w, out_h, out_w, stride, c / 8);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
