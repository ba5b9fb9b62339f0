// Elementwise kernels of the split-precision ("x3") head plan (include/sipmask_hip.h: SM_CONV_F16).
//
// The reference head is fp32 end to end (M/mmdet/models/anchor_heads/sipmask_head.py:241-287,609-633); the bf16 plan's
// operand rounding (2^-9 per element) is what separates its mask logits from the reference's by ~1.5.  The x3 plan keeps
// every head activation in f32 between layers and feeds the MFMA convolutions with TWO binary16 halves per value,
//     v  =  hi + lo,   hi = f16(v),   lo = f16(v - hi)          (11 + 11 mantissa bits; v - hi is exact in f32)
// laid out along the channel axis as [hi | lo | hi] (3*C channels) against weights [hi | hi | lo]: one ordinary convolution
// over 3*C channels is then  x_hi*w_hi + x_lo*w_hi + x_hi*w_lo  accumulated in f32 by the MFMA -- the product to ~2^-21,
// the dropped lo*lo term is 2^-22.  (bf16 halves give 8 + 8 bits: tools/x3_emulate.py measures 1.0e-3 on the mask logits
// with them, 7.8e-5 with binary16.)  These kernels produce that layout:
//   sm_split3_f16          f32 or bf16 rows -> [hi | lo | hi] rows (a channel slice of a wider destination: the mask branch's
//                          concatenation writes three sources into one)
//   sm_gn_stats_f32_fix    GroupNorm statistics of f32 rows as fixed-point sums (common.h: gn_fix; one POSITION's 8
//                          channels per rounding, integer accumulation: order- and tiling-independent)
//   sm_groupnorm_apply_x3  normalise (+ReLU) f32 rows with such statistics -> f32 rows and / or [hi | lo | hi] rows
// Values beyond binary16's range (|v| > 65504) saturate; activations of a working detector are orders of magnitude below.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float* v, half8& hi, half8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float c = fminf(fmaxf(v[e], -65504.f), 65504.f);
    const _Float16 h = (_Float16)c;                 // v_cvt_f16_f32, round to nearest even
    hi[e] = h;
    lo[e] = (_Float16)(c - (float)h);               // exact difference (<= 13 significant bits), then one rounding
  }
}

struct SplitArgs {
  const void* x;
  uint16_t* y;
  long long rows;
  int c8;          // 8-channel chunks per source row
  int in_cs;       // source row stride (elements)
  int out_cs;      // destination row stride (elements) = 3 * ctot
  int ctot, coff;  // channels of one third of the destination row, first channel of this source inside it
};

// TWO: the two-term layout [hi | hi] for sources whose low half is zero (bf16 rows: every bf16 value above 2^-17 is a
// binary16 value) -- against weights [hi | lo] the conv is x_hi*w_hi + x_hi*w_lo, the three-term product without its zero
// term (sm_split2_f16, round 5: the first tower convs of the x3 plan read the bf16 FPN outputs)
// PAIRS (round 6): the paired layout of sm_conv_desc.x3_pairs -- a row of C values is C/16 groups of [hi 16 | lo 16]
// (2*C binary16 per row; channel c: hi at (c >> 4) * 32 + (c & 15), lo 16 elements behind it), the operand the patch kernel
// reads as (x_hi, x_lo) fragment pairs: 4 bytes written per value instead of the 6 of [hi | lo | hi]
template <bool IN_F32, bool TWO = false, bool PAIRS = false>
__global__ __launch_bounds__(256) void split3_kernel(const SplitArgs a) {
  const long long total = a.rows * a.c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / a.c8;
    const int c = (int)(i - r * a.c8) * 8;
    float v[8];
    if constexpr (IN_F32) {
      const float* p = (const float*)a.x + r * a.in_cs + c;
      const float4 q0 = *reinterpret_cast<const float4*>(p), q1 = *reinterpret_cast<const float4*>(p + 4);
      v[0] = q0.x, v[1] = q0.y, v[2] = q0.z, v[3] = q0.w, v[4] = q1.x, v[5] = q1.y, v[6] = q1.z, v[7] = q1.w;
    } else {
      unpack_bf16x8(*reinterpret_cast<const uint4*>((const uint16_t*)a.x + r * a.in_cs + c), v);
    }
    half8 hi, lo;
    split8(v, hi, lo);
    if constexpr (PAIRS) {
      const int ch = a.coff + c;
      uint16_t* o = a.y + r * a.out_cs + (ch >> 4) * 32 + (ch & 15);
      *reinterpret_cast<half8*>(o) = hi;
      *reinterpret_cast<half8*>(o + 16) = lo;
      continue;
    }
    uint16_t* o = a.y + r * a.out_cs + a.coff + c;
    *reinterpret_cast<half8*>(o) = hi;
    if constexpr (TWO) {
      *reinterpret_cast<half8*>(o + a.ctot) = hi;
    } else {
      *reinterpret_cast<half8*>(o + a.ctot) = lo;
      *reinterpret_cast<half8*>(o + 2 * a.ctot) = hi;
    }
  }
}

// bilinear x`factor` upsampling (align_corners=False; the formula of upsample_bilinear_kernel, misc.hip) of f32 rows
// written straight as [hi | lo | hi]: the mask branch's concatenation [l0 | up2(l1) | up4(l2)] (sipmask_head.py:266-275)
// becomes three launches into one split tensor instead of three upsamples into an f32 tensor + a split pass over it
__global__ __launch_bounds__(256) void upsample_split3_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int B, int H,
                                                              int W, int C, int factor, int in_cs, int ctot, int coff) {
  const int Ho = H * factor, Wo = W * factor, cv = C / 8;
  const long long total = (long long)B * Ho * Wo * cv;
  const float inv = 1.f / (float)factor;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cv);
    long long p = i / cv;
    const int wo = (int)(p % Wo);
    p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const float sy = fmaxf(((float)ho + 0.5f) * inv - 0.5f, 0.f), sx = fmaxf(((float)wo + 0.5f) * inv - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* pa = x + (((long long)n * H + y0) * W + x0) * in_cs + cc * 8;
    const float* pb = x + (((long long)n * H + y0) * W + x1) * in_cs + cc * 8;
    const float* pc = x + (((long long)n * H + y1) * W + x0) * in_cs + cc * 8;
    const float* pd = x + (((long long)n * H + y1) * W + x1) * in_cs + cc * 8;
    float a[8], b[8], c[8], d[8], r[8];
    *reinterpret_cast<float4*>(a) = *reinterpret_cast<const float4*>(pa);
    *reinterpret_cast<float4*>(a + 4) = *reinterpret_cast<const float4*>(pa + 4);
    *reinterpret_cast<float4*>(b) = *reinterpret_cast<const float4*>(pb);
    *reinterpret_cast<float4*>(b + 4) = *reinterpret_cast<const float4*>(pb + 4);
    *reinterpret_cast<float4*>(c) = *reinterpret_cast<const float4*>(pc);
    *reinterpret_cast<float4*>(c + 4) = *reinterpret_cast<const float4*>(pc + 4);
    *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(pd);
    *reinterpret_cast<float4*>(d + 4) = *reinterpret_cast<const float4*>(pd + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = hy * (hx * a[e] + lx * b[e]) + ly * (hx * c[e] + lx * d[e]);
    half8 hi, lo;
    split8(r, hi, lo);
    uint16_t* o = y + (((long long)n * Ho + ho) * Wo + wo) * (3ll * ctot) + coff + cc * 8;
    *reinterpret_cast<half8*>(o) = hi;
    *reinterpret_cast<half8*>(o + ctot) = lo;
    *reinterpret_cast<half8*>(o + 2 * ctot) = hi;
  }
}

struct GnxArgs {
  int nlev, batch, C, groups, cpg;
  int hw[SM_MAX_LEVELS];
  long long row0[SM_MAX_LEVELS];
  int blk0[SM_MAX_LEVELS + 1];
  float eps;
  int relu, rpb;
};

__device__ __forceinline__ int gnx_level(const GnxArgs& a) {
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && (int)blockIdx.x >= a.blk0[l]) lev = l;
  return lev;
}

// thread = one 8-channel chunk column (inside one group) of the block's rows
__global__ __launch_bounds__(256) void gnx_stats_kernel(const float* __restrict__ x, unsigned long long* __restrict__ stats,
                                                        const GnxArgs a) {
  const int n = blockIdx.y, lev = gnx_level(a);
  const int rb = (blockIdx.x - a.blk0[lev]) * a.rpb, HW = a.hw[lev];
  const int c8 = a.C / 8, lanes = 256 / c8;
  const int cc = threadIdx.x % c8, rr = threadIdx.x / c8;
  const float* base = x + (a.row0[lev] + (long long)n * HW) * a.C;
  unsigned long long s = 0ull, ss = 0ull;
  if (rr < lanes) {
    const int rend = min(rb + a.rpb, HW);
    for (int r = rb + rr; r < rend; r += lanes) {
      const float* p = base + (long long)r * a.C + cc * 8;
      const float4 q0 = *reinterpret_cast<const float4*>(p), q1 = *reinterpret_cast<const float4*>(p + 4);
      const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
      float ps = 0.f, pss = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ps += v[e];
        pss = __builtin_fmaf(v[e], v[e], pss);
      }
      s += gn_fix(ps);               // one position's 8 channels per rounding: the conv epilogues' unit (conv_igemm.hip)
      ss += gn_fix(pss);
    }
  }
  __shared__ unsigned long long sh[2][256];
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = ss;
  __syncthreads();
  const int cpg8 = a.cpg / 8;
  if ((int)threadIdx.x < a.groups) {
    const int g = threadIdx.x;
    unsigned long long ts = 0ull, tss = 0ull;
    for (int r = 0; r < lanes; ++r)
      for (int k = 0; k < cpg8; ++k) {
        ts += sh[0][r * c8 + g * cpg8 + k];
        tss += sh[1][r * c8 + g * cpg8 + k];
      }
    unsigned long long* st = stats + (((long long)n * a.nlev + lev) * a.groups + g) * 2;
    atomicAdd(st, ts);
    atomicAdd(st + 1, tss);
  }
}

template <bool PAIRS>
__global__ __launch_bounds__(256) void gnx_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const unsigned long long* __restrict__ stats, float* __restrict__ y32,
                                                        uint16_t* __restrict__ y3, const GnxArgs a) {
  const int n = blockIdx.y, lev = gnx_level(a);
  const int rb = (blockIdx.x - a.blk0[lev]) * a.rpb, HW = a.hw[lev];
  const int c8 = a.C / 8, lanes = 256 / c8;
  const int cc = threadIdx.x % c8, rr = threadIdx.x / c8;
  if (rr >= lanes) return;
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
  const long long row00 = a.row0[lev] + (long long)n * HW;
  const int rend = min(rb + a.rpb, HW);
  for (int r = rb + rr; r < rend; r += lanes) {
    const long long row = row00 + r;
    const float* p = x + row * a.C + cc * 8;
    const float4 q0 = *reinterpret_cast<const float4*>(p), q1 = *reinterpret_cast<const float4*>(p + 4);
    float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = v[e] * sc[e] + sf[e];
      v[e] = a.relu ? fmaxf(t, 0.f) : t;
    }
    if (y32 != nullptr) {
      float* o = y32 + row * a.C + cc * 8;
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (y3 != nullptr) {
      half8 hi, lo;
      split8(v, hi, lo);
      if constexpr (PAIRS) {                        // [hi 16 | lo 16] per 16 channels (sm_conv_desc.x3_pairs)
        uint16_t* o = y3 + row * (2ll * a.C) + (cc >> 1) * 32 + (cc & 1) * 8;
        *reinterpret_cast<half8*>(o) = hi;
        *reinterpret_cast<half8*>(o + 16) = lo;
      } else {
        uint16_t* o = y3 + row * (3ll * a.C) + cc * 8;
        *reinterpret_cast<half8*>(o) = hi;
        *reinterpret_cast<half8*>(o + a.C) = lo;
        *reinterpret_cast<half8*>(o + 2 * a.C) = hi;
      }
    }
  }
}

int gnx_fill(GnxArgs& a, int& t, int batch, int nlev, const int32_t* hw, const int64_t* row0, int channels, int groups,
             float eps, int relu) {
  if (nlev < 1 || nlev > SM_MAX_LEVELS || batch < 1 || groups < 1 || groups > 256) return SM_ERR_BAD_SHAPE;
  if (channels % (8 * groups) != 0 || channels > 2048 || 256 % (channels / 8) != 0) return SM_ERR_BAD_SHAPE;
  a.nlev = nlev, a.batch = batch, a.C = channels, a.groups = groups, a.cpg = channels / groups;
  a.eps = eps, a.relu = relu;
  long long blocks = 0;
  for (int l = 0; l < nlev; ++l) blocks += (long long)batch * sm_cdiv(hw[l], 256);
  a.rpb = blocks < 512 ? 32 : (blocks < 1024 ? 64 : 256);          // short blocks for small tensors (>= ~4 blocks per CU)
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

}  // namespace

extern "C" int sm_split3_f16(const void* x, int x_is_f32, int64_t rows, int channels, int in_cstride, void* y,
                             int ctot, int coff, sm_stream_t stream) {
  if (!x || !y) return SM_ERR_BAD_ARG;
  if (rows < 1 || channels < 8 || channels % 8 || in_cstride % 8 || in_cstride < channels || ctot % 8 || coff % 8 ||
      coff + channels > ctot)
    return SM_ERR_BAD_SHAPE;
  SplitArgs a;
  a.x = x, a.y = (uint16_t*)y, a.rows = rows, a.c8 = channels / 8, a.in_cs = in_cstride, a.out_cs = 3 * ctot;
  a.ctot = ctot, a.coff = coff;
  const long long n = rows * a.c8;
  long long g = (n + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  if (x_is_f32)
    hipLaunchKernelGGL(split3_kernel<true>, dim3((unsigned)g), dim3(256), 0, sm_hip_stream(stream), a);
  else
    hipLaunchKernelGGL(split3_kernel<false>, dim3((unsigned)g), dim3(256), 0, sm_hip_stream(stream), a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_split2_f16(const void* x_bf16, int64_t rows, int channels, int in_cstride, void* y, int ctot, int coff,
                             sm_stream_t stream) {
  if (!x_bf16 || !y) return SM_ERR_BAD_ARG;
  if (rows < 1 || channels < 8 || channels % 8 || in_cstride % 8 || in_cstride < channels || ctot % 8 || coff % 8 ||
      coff + channels > ctot)
    return SM_ERR_BAD_SHAPE;
  SplitArgs a;
  a.x = x_bf16, a.y = (uint16_t*)y, a.rows = rows, a.c8 = channels / 8, a.in_cs = in_cstride, a.out_cs = 2 * ctot;
  a.ctot = ctot, a.coff = coff;
  const long long n = rows * a.c8;
  long long g = (n + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  hipLaunchKernelGGL((split3_kernel<false, true>), dim3((unsigned)g), dim3(256), 0, sm_hip_stream(stream), a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_gn_stats_f32_fix(const float* x, int64_t* stats, int batch, int nlev, const int32_t* hw,
                                   const int64_t* row0, int channels, int groups, sm_stream_t stream) {
  if (!x || !stats || !hw || !row0) return SM_ERR_BAD_ARG;
  GnxArgs a;
  int t;
  const int rc = gnx_fill(a, t, batch, nlev, hw, row0, channels, groups, 0.f, 0);
  if (rc != SM_OK) return rc;
  hipStream_t s = sm_hip_stream(stream);
  if (sm_zero_async(stats, sizeof(unsigned long long) * 2 * batch * nlev * groups, s) != hipSuccess) return SM_ERR_LAUNCH;
  hipLaunchKernelGGL(gnx_stats_kernel, dim3(t, batch), dim3(256), 0, s, x, reinterpret_cast<unsigned long long*>(stats), a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_groupnorm_apply_x3(const float* x, const float* gamma, const float* beta, const int64_t* stats, int batch,
                                     int nlev, const int32_t* hw, const int64_t* row0, int channels, int groups, float eps,
                                     int relu, float* y_f32, void* y_split, sm_stream_t stream) {
  if (!x || !gamma || !beta || !stats || !hw || !row0 || (!y_f32 && !y_split)) return SM_ERR_BAD_ARG;
  GnxArgs a;
  int t;
  const int rc = gnx_fill(a, t, batch, nlev, hw, row0, channels, groups, eps, relu);
  if (rc != SM_OK) return rc;
  hipLaunchKernelGGL(gnx_apply_kernel<false>, dim3(t, batch), dim3(256), 0, sm_hip_stream(stream), x, gamma, beta,
                     reinterpret_cast<const unsigned long long*>(stats), y_f32, (uint16_t*)y_split, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_groupnorm_apply_x3p(const float* x, const float* gamma, const float* beta, const int64_t* stats, int batch,
                                      int nlev, const int32_t* hw, const int64_t* row0, int channels, int groups, float eps,
                                      int relu, float* y_f32, void* y_pairs, sm_stream_t stream) {
  if (!x || !gamma || !beta || !stats || !hw || !row0 || !y_pairs) return SM_ERR_BAD_ARG;
  if (channels % 16 != 0) return SM_ERR_BAD_SHAPE;
  GnxArgs a;
  int t;
  const int rc = gnx_fill(a, t, batch, nlev, hw, row0, channels, groups, eps, relu);
  if (rc != SM_OK) return rc;
  hipLaunchKernelGGL(gnx_apply_kernel<true>, dim3(t, batch), dim3(256), 0, sm_hip_stream(stream), x, gamma, beta,
                     reinterpret_cast<const unsigned long long*>(stats), y_f32, (uint16_t*)y_pairs, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_split_pairs_f16(const void* x, int x_is_f32, int64_t rows, int channels, int in_cstride, void* y,
                                  int ctot, int coff, sm_stream_t stream) {
  if (!x || !y) return SM_ERR_BAD_ARG;
  if (rows < 1 || channels < 8 || channels % 8 || in_cstride % 8 || in_cstride < channels || ctot % 16 || coff % 8 ||
      coff + channels > ctot)
    return SM_ERR_BAD_SHAPE;
  SplitArgs a;
  a.x = x, a.y = (uint16_t*)y, a.rows = rows, a.c8 = channels / 8, a.in_cs = in_cstride, a.out_cs = 2 * ctot;
  a.ctot = ctot, a.coff = coff;
  const long long n = rows * a.c8;
  long long g = (n + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  if (x_is_f32)
    hipLaunchKernelGGL((split3_kernel<true, false, true>), dim3((unsigned)g), dim3(256), 0, sm_hip_stream(stream), a);
  else
    hipLaunchKernelGGL((split3_kernel<false, false, true>), dim3((unsigned)g), dim3(256), 0, sm_hip_stream(stream), a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

// ---------------------------------------------------------------- sip_mask_lat0 by linearity (round 4)
// sip_mask_lat0 is a 1x1 conv over [l0 | up2(l1) | up4(l2)] (sipmask_head.py:275-283), and a 1x1 conv commutes with bilinear
// upsampling: W . [l0 | up2 l1 | up4 l2] = W0 . l0 + up2(W1 . l1) + up4(W2 . l2).  The three products run at their own
// resolutions (23.1 instead of 52.9 GFLOP per 4 images, no 768-channel concatenation: 103 MB written and read back) and this
// kernel adds the two coarse ones onto the fine grid:   out = [relu](a0 + up2(a1) + up4(a2))
//   F32 = false: a1, a2 bf16 rows, a0 absent, out bf16 rows (the residual of the l0 conv: bias / ReLU in its epilogue);
//   F32 = true : a0 (optional), a1, a2 f32 rows, out f32 rows or the split layout [hi | lo | hi] of sm_split3_f16.
//   X3OUT = 2 (round 6): the paired layout [hi 16 | lo 16] per 16 channels (sm_split_pairs_f16), 2 * C per row.
template <bool F32, int X3OUT>
__global__ __launch_bounds__(256) void upsample_sum2_kernel(const void* __restrict__ a0v, const void* __restrict__ a1v,
                                                            const void* __restrict__ a2v, void* __restrict__ outv, int B, int H0,
                                                            int W0, int C, int relu) {
  const int cv = C / 8;
  const long long total = (long long)B * H0 * W0 * cv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cv);
    long long p = i / cv;
    const int wo = (int)(p % W0);
    p /= W0;
    const int ho = (int)(p % H0);
    const int n = (int)(p / H0);
    const long long orow = ((long long)n * H0 + ho) * W0 + wo;
    float r[8];
    if (F32 && a0v != nullptr) {
      const float* q = (const float*)a0v + orow * C + cc * 8;
      *reinterpret_cast<float4*>(r) = *reinterpret_cast<const float4*>(q);
      *reinterpret_cast<float4*>(r + 4) = *reinterpret_cast<const float4*>(q + 4);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = 0.f;
    }
#pragma unroll
    for (int lvl = 1; lvl <= 2; ++lvl) {                  // up2(a1), up4(a2): upsample_bilinear_kernel's source rule
      const int f = 1 << lvl, H = H0 / f, W = W0 / f;
      const float inv = 1.f / (float)f;
      const float sy = fmaxf(((float)ho + 0.5f) * inv - 0.5f, 0.f), sx = fmaxf(((float)wo + 0.5f) * inv - 0.5f, 0.f);
      const int y0 = (int)sy, x0 = (int)sx;
      const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
      const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
      const long long r00 = ((long long)n * H + y0) * W + x0, r01 = ((long long)n * H + y0) * W + x1;
      const long long r10 = ((long long)n * H + y1) * W + x0, r11 = ((long long)n * H + y1) * W + x1;
      const void* src = lvl == 1 ? a1v : a2v;
      float a[8], b[8], c[8], d[8];
      if constexpr (F32) {
        const float* x = (const float*)src;
        auto ld = [&](long long row, float* dst) {
          *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(x + row * C + cc * 8);
          *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(x + row * C + cc * 8 + 4);
        };
        ld(r00, a), ld(r01, b), ld(r10, c), ld(r11, d);
      } else {
        const uint16_t* x = (const uint16_t*)src;
        unpack_bf16x8(*reinterpret_cast<const uint4*>(x + r00 * C + cc * 8), a);
        unpack_bf16x8(*reinterpret_cast<const uint4*>(x + r01 * C + cc * 8), b);
        unpack_bf16x8(*reinterpret_cast<const uint4*>(x + r10 * C + cc * 8), c);
        unpack_bf16x8(*reinterpret_cast<const uint4*>(x + r11 * C + cc * 8), d);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] += hy * (hx * a[e] + lx * b[e]) + ly * (hx * c[e] + lx * d[e]);
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = fmaxf(r[e], 0.f);
    }
    if constexpr (X3OUT == 2) {
      half8 hi, lo;
      split8(r, hi, lo);
      uint16_t* o = (uint16_t*)outv + orow * (2ll * C) + (cc >> 1) * 32 + (cc & 1) * 8;
      *reinterpret_cast<half8*>(o) = hi;
      *reinterpret_cast<half8*>(o + 16) = lo;
    } else if constexpr (X3OUT == 1) {
      half8 hi, lo;
      split8(r, hi, lo);
      uint16_t* o = (uint16_t*)outv + orow * (3ll * C) + cc * 8;
      *reinterpret_cast<half8*>(o) = hi;
      *reinterpret_cast<half8*>(o + C) = lo;
      *reinterpret_cast<half8*>(o + 2 * C) = hi;
    } else if constexpr (F32) {
      float* o = (float*)outv + orow * C + cc * 8;
      *reinterpret_cast<float4*>(o) = *reinterpret_cast<float4*>(r);
      *reinterpret_cast<float4*>(o + 4) = *reinterpret_cast<float4*>(r + 4);
    } else {
      *reinterpret_cast<uint4*>((uint16_t*)outv + orow * C + cc * 8) = pack_bf16x8(r);
    }
  }
}

extern "C" int sm_upsample_sum2(const float* a0, const void* a1, const void* a2, int is_f32, int batch, int h0, int w0, int c,
                                int relu, int out_x3, void* out, sm_stream_t stream) {
  if (!a1 || !a2 || !out) return SM_ERR_BAD_ARG;
  if (batch < 1 || h0 < 4 || w0 < 4 || (h0 & 3) || (w0 & 3) || c < 8 || (c & 7)) return SM_ERR_BAD_SHAPE;   // exact x2 / x4 grids
  if (!is_f32 && (a0 != nullptr || out_x3)) return SM_ERR_UNSUPPORTED;
  const long long n = (long long)batch * h0 * w0 * (c / 8);
  long long g = (n + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  hipStream_t s = sm_hip_stream(stream);
  if (out_x3 == 2 && (c & 15)) return SM_ERR_BAD_SHAPE;
  if (!is_f32)
    hipLaunchKernelGGL((upsample_sum2_kernel<false, 0>), dim3((unsigned)g), dim3(256), 0, s, nullptr, a1, a2, out, batch, h0, w0, c, relu);
  else if (out_x3 == 2)
    hipLaunchKernelGGL((upsample_sum2_kernel<true, 2>), dim3((unsigned)g), dim3(256), 0, s, a0, a1, a2, out, batch, h0, w0, c, relu);
  else if (out_x3)
    hipLaunchKernelGGL((upsample_sum2_kernel<true, 1>), dim3((unsigned)g), dim3(256), 0, s, a0, a1, a2, out, batch, h0, w0, c, relu);
  else
    hipLaunchKernelGGL((upsample_sum2_kernel<true, 0>), dim3((unsigned)g), dim3(256), 0, s, a0, a1, a2, out, batch, h0, w0, c, relu);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_upsample_bilinear_x3(const float* x, void* y, int batch, int h, int w, int c, int factor, int in_cstride,
                                       int ctot, int coff, sm_stream_t stream) {
  if (!x || !y || factor < 1) return SM_ERR_BAD_ARG;
  if (batch < 1 || h < 1 || w < 1 || c < 8 || c % 8 || in_cstride % 4 || in_cstride < c || ctot % 8 || coff % 8 || coff + c > ctot)
    return SM_ERR_BAD_SHAPE;
  const long long n = (long long)batch * h * factor * w * factor * (c / 8);
  long long g = (n + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  hipLaunchKernelGGL(upsample_split3_kernel, dim3((unsigned)g), dim3(256), 0, sm_hip_stream(stream), x, (uint16_t*)y, batch, h, w,
                     c, factor, in_cstride, ctot, coff);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
