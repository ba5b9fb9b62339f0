// Training-path layers of the SipMask head that are not GEMMs: GroupNorm(+ReLU) forward/backward,
// bilinear upsampling forward/backward (align_corners=False, integer factor) and the SGD-momentum update, on f32
// NCHW tensors as autograd hands them over.  All HBM-bound: a GroupNorm group is one contiguous run of
// (C/G)*H*W floats in NCHW, so every (image, group) is one block streaming contiguous memory.
#include <algorithm>

#include "common.h"

namespace {

constexpr int TN_THREADS = 256;

__device__ __forceinline__ float block_sum(float v, float* s_red) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < TN_THREADS / 64; ++w) t += s_red[w];
  return t;
}

// nn.GroupNorm (+ optional ReLU): y = gamma * (x - mean) * rstd + beta, in two fully parallel phases.  A group of an
// image is one contiguous run of m = (C/G)*H*W floats in NCHW; phase 1 reduces it with `nsplit` blocks per group
// (partial sums of x - pivot and (x - pivot)^2, pivot = the group's first element, so the variance does not lose
// precision to a large mean), phase 2 is elementwise.  One block per group would leave the chip idle: 32 groups x
// 4 images = 128 blocks for 256 CUs, each streaming megabytes.
__global__ __launch_bounds__(TN_THREADS) void gn_reduce_kernel(const float* __restrict__ x, float* __restrict__ part, int C,
                                                               int HW, int G, int nsplit) {
  __shared__ float s_red[TN_THREADS / 64];
  const int g = blockIdx.x, n = blockIdx.y, sp = blockIdx.z, tid = threadIdx.x;
  const int cpg = C / G;
  const long long base = ((long long)n * C + (long long)g * cpg) * HW;
  const int m = cpg * HW;
  const int per = (m + nsplit - 1) / nsplit;
  const int lo = sp * per, hi = min(lo + per, m);
  const float pivot = x[base];
  float s = 0.f, ss = 0.f;
  for (int i = lo + tid; i < hi; i += TN_THREADS) {
    const float d = x[base + i] - pivot;
    s += d;
    ss += d * d;
  }
  s = block_sum(s, s_red);
  ss = block_sum(ss, s_red);
  if (tid == 0) {
    unsafeAtomicAdd(part + ((long long)n * G + g) * 2 + 0, s);
    unsafeAtomicAdd(part + ((long long)n * G + g) * 2 + 1, ss);
  }
}

// part (sum, sum of squares of x - pivot) -> stats (mean, rstd); one thread per (n, g)
__global__ void gn_finish_kernel(const float* __restrict__ x, const float* __restrict__ part, float* __restrict__ stats,
                                 int NG, int C, int HW, int G, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NG) return;
  const int n = i / G, g = i - n * G;
  const int cpg = C / G;
  const float m = (float)cpg * (float)HW;
  const float pivot = x[((long long)n * C + (long long)g * cpg) * HW];
  const float ms = part[i * 2] / m;
  const float var = fmaxf(part[i * 2 + 1] / m - ms * ms, 0.f);
  stats[i * 2] = pivot + ms;
  stats[i * 2 + 1] = rsqrtf(var + eps);
}

__global__ void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ stats, float* __restrict__ y, long long total, int C, int HW,
                                int G, int relu) {
  const int cpg = C / G;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    const long long ng = (i / ((long long)C * HW)) * G + c / cpg;
    float v = (x[i] - stats[ng * 2]) * stats[ng * 2 + 1] * gamma[c] + beta[c];
    if (relu) v = fmaxf(v, 0.f);
    y[i] = v;
  }
}

// backward: dy is first masked by (y > 0) when relu.  Phase 1 (nsplit blocks per (n, group)): per-channel
// sums a_c = sum dy*xhat, b_c = sum dy -> atomics into dgamma / dbeta and into the group's (sum gamma*b, sum gamma*a);
// phase 2 elementwise: dx = rstd * (gamma*dy - mean_g(gamma*dy) - xhat * mean_g(gamma*dy*xhat)).
__global__ __launch_bounds__(TN_THREADS) void gn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                   const float* __restrict__ dy,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ stats, float* __restrict__ part,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   int C, int HW, int G, int relu, int nsplit) {
  __shared__ float s_red[TN_THREADS / 64];
  const int g = blockIdx.x, n = blockIdx.y, sp = blockIdx.z, tid = threadIdx.x;
  const int cpg = C / G;
  const long long base = ((long long)n * C + (long long)g * cpg) * HW;
  const float mean = stats[((long long)n * G + g) * 2 + 0], rstd = stats[((long long)n * G + g) * 2 + 1];
  const int per = (HW + nsplit - 1) / nsplit;
  const int lo = sp * per, hi = min(lo + per, HW);
  float s1 = 0.f, s2 = 0.f;
  for (int cl = 0; cl < cpg; ++cl) {
    const int c = g * cpg + cl;
    float a = 0.f, b = 0.f;
    for (int i = lo + tid; i < hi; i += TN_THREADS) {
      const long long o = base + (long long)cl * HW + i;
      float d = dy[o];
      if (relu && !(y[o] > 0.f)) d = 0.f;
      a += d * ((x[o] - mean) * rstd);
      b += d;
    }
    const float ta = block_sum(a, s_red), tb = block_sum(b, s_red);
    if (tid == 0) {
      if (dgamma) unsafeAtomicAdd(dgamma + c, ta);
      if (dbeta) unsafeAtomicAdd(dbeta + c, tb);
    }
    s1 += gamma[c] * tb;
    s2 += gamma[c] * ta;
  }
  if (tid == 0) {
    unsafeAtomicAdd(part + ((long long)n * G + g) * 2 + 0, s1);
    unsafeAtomicAdd(part + ((long long)n * G + g) * 2 + 1, s2);
  }
}

__global__ void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                    const float* __restrict__ gamma, const float* __restrict__ stats,
                                    const float* __restrict__ part, float* __restrict__ dx, long long total, int C, int HW,
                                    int G, int relu) {
  const int cpg = C / G;
  const float inv_m = 1.f / ((float)cpg * (float)HW);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    const long long ng = (i / ((long long)C * HW)) * G + c / cpg;
    float d = dy[i];
    if (relu && !(y[i] > 0.f)) d = 0.f;
    const float rstd = stats[ng * 2 + 1];
    const float xh = (x[i] - stats[ng * 2]) * rstd;
    dx[i] = rstd * (gamma[c] * d - part[ng * 2] * inv_m - xh * part[ng * 2 + 1] * inv_m);
  }
}

// F.interpolate(mode='bilinear', align_corners=False, scale_factor=f) on NCHW f32, one thread per output pixel
__device__ __forceinline__ void bil_src(int o, int n_in, float inv, int& i0, int& i1, float& l) {
  const float s = fmaxf(inv * ((float)o + 0.5f) - 0.5f, 0.f);
  i0 = min((int)s, n_in - 1);
  i1 = min(i0 + 1, n_in - 1);
  l = s - (float)i0;
}

__global__ void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int h, int w,
                                    int f) {
  const int ho = h * f, wo = w * f;
  const long long total = planes * ho * wo;
  const float inv = 1.f / (float)f;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(t % wo), oy = (int)((t / wo) % ho);
    const long long p = t / ((long long)wo * ho);
    int y0, y1, x0, x1;
    float ly, lx;
    bil_src(oy, h, inv, y0, y1, ly);
    bil_src(ox, w, inv, x0, x1, lx);
    const float* s = x + p * h * w;
    y[t] = (1.f - ly) * ((1.f - lx) * s[y0 * w + x0] + lx * s[y0 * w + x1]) +
           ly * ((1.f - lx) * s[y1 * w + x0] + lx * s[y1 * w + x1]);
  }
}

// adjoint of the above in gather form (deterministic, no atomics): input pixel (y, x) collects from the <= 3f x 3f
// output pixels whose interpolation touches it, with the forward's own index/weight function (so clamped borders,
// where both taps land on the same source pixel, get weight 1 as in the forward)
__global__ void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long long planes, int h, int w,
                                    int f) {
  const int ho = h * f, wo = w * f;
  const long long total = planes * h * w;
  const float inv = 1.f / (float)f;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(t % w), y = (int)((t / w) % h);
    const long long p = t / ((long long)w * h);
    const float* g = dy + p * ho * wo;
    float acc = 0.f;
    for (int oy = max(0, f * y - f); oy < min(ho, f * y + 2 * f); ++oy) {
      int y0, y1;
      float ly;
      bil_src(oy, h, inv, y0, y1, ly);
      const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int ox = max(0, f * x - f); ox < min(wo, f * x + 2 * f); ++ox) {
        int x0, x1;
        float lx;
        bil_src(ox, w, inv, x0, x1, lx);
        const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        if (wx != 0.f) row += wx * g[(long long)oy * wo + ox];
      }
      acc += wy * row;
    }
    dx[t] = acc;
  }
}

// torch.optim.SGD step (momentum, weight decay, dampening 0, no nesterov): g += wd*p; buf = mom*buf + g (buf = g on
// the first step); p -= lr*buf.  The reference's optimizer: cfg optimizer=dict(type='SGD', lr=.01, momentum=.9,
// weight_decay=1e-4) with paramwise bias_lr_mult=2, bias_decay_mult=0 (M/mmdet/apis/train.py:92-133).
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long long n, float lr,
                           float momentum, float wd, int first) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] + wd * p[i];
    const float b = first ? gi : momentum * buf[i] + gi;
    buf[i] = b;
    p[i] -= lr * b;
  }
}

// the same update for EVERY parameter tensor in one launch: items[] = per-tensor pointers and hyper-parameters,
// blocks[] = (item, chunk) per thread block (chunks of SGD_CHUNK elements)
constexpr int SGD_CHUNK = 4096;
struct SgdItem {
  float* p;
  const float* g;
  float* buf;
  long long n;
  float lr, wd;
};
__global__ __launch_bounds__(256) void sgd_multi_kernel(const SgdItem* __restrict__ items, const int2* __restrict__ blocks,
                                                        float momentum, int first) {
  const int2 bk = blocks[blockIdx.x];
  const SgdItem it = items[bk.x];
  const long long i0 = (long long)bk.y * SGD_CHUNK;
  const long long i1 = i0 + SGD_CHUNK < it.n ? i0 + SGD_CHUNK : it.n;
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
    const float gi = it.g[i] + it.wd * it.p[i];
    const float b = first ? gi : momentum * it.buf[i] + gi;
    it.buf[i] = b;
    it.p[i] -= it.lr * b;
  }
}

inline int grid_1d(long long n) { return (int)std::min<long long>((n + 255) / 256, 256 * 32); }

}  // namespace

static int gn_nsplit(int batch, int groups, long long m) {
  // enough blocks for the chip (>= ~1024), at least ~16 K elements per block
  long long want = (1024 + (long long)batch * groups - 1) / ((long long)batch * groups);
  long long cap = (m + 16383) / 16384;
  long long ns = want < cap ? want : cap;
  return (int)(ns < 1 ? 1 : (ns > 64 ? 64 : ns));
}

extern "C" int sm_groupnorm_nchw_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                                     int batch, int channels, int hw, int groups, float eps, int relu,
                                     sm_stream_t stream) {
  if (!x || !gamma || !beta || !y || !stats) return SM_ERR_BAD_ARG;
  if (batch < 1 || channels < 1 || hw < 1 || groups < 1 || channels % groups != 0) return SM_ERR_BAD_SHAPE;
  hipStream_t s = sm_hip_stream(stream);
  const int ng = batch * groups;
  const long long m = (long long)(channels / groups) * hw;
  const int ns = gn_nsplit(batch, groups, m);
  // the (mean, rstd) buffer doubles as the partial-sum buffer of phase 1
  if (sm_zero_async(stats, sizeof(float) * 2 * ng, s) != hipSuccess) return SM_ERR_LAUNCH;
  hipLaunchKernelGGL(gn_reduce_kernel, dim3(groups, batch, ns), dim3(TN_THREADS), 0, s, x, stats, channels, hw, groups, ns);
  hipLaunchKernelGGL(gn_finish_kernel, dim3((ng + 63) / 64), dim3(64), 0, s, x, stats, stats, ng, channels, hw, groups, eps);
  const long long total = (long long)batch * channels * hw;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_1d(total)), dim3(256), 0, s, x, gamma, beta, stats, y, total, channels, hw,
                     groups, relu);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_groupnorm_nchw_bwd(const float* x, const float* y, const float* dy, const float* gamma,
                                     const float* stats, float* dx, float* dgamma, float* dbeta, float* scratch,
                                     int batch, int channels, int hw, int groups, int relu, sm_stream_t stream) {
  if (!x || !dy || !gamma || !stats || !scratch || (relu && !y)) return SM_ERR_BAD_ARG;
  if (batch < 1 || channels < 1 || hw < 1 || groups < 1 || channels % groups != 0) return SM_ERR_BAD_SHAPE;
  hipStream_t s = sm_hip_stream(stream);
  if (dgamma && sm_zero_async(dgamma, sizeof(float) * channels, s) != hipSuccess) return SM_ERR_LAUNCH;
  if (dbeta && sm_zero_async(dbeta, sizeof(float) * channels, s) != hipSuccess) return SM_ERR_LAUNCH;
  // scratch f32 [batch][groups][2]: the group sums (sum gamma*dy, sum gamma*dy*xhat) between the two phases
  const int ng = batch * groups;
  if (sm_zero_async(scratch, sizeof(float) * 2 * ng, s) != hipSuccess) return SM_ERR_LAUNCH;
  const int ns = gn_nsplit(batch, groups, hw);
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(groups, batch, ns), dim3(TN_THREADS), 0, s, x, y, dy, gamma, stats, scratch,
                     dgamma, dbeta, channels, hw, groups, relu, ns);
  if (dx) {
    const long long total = (long long)batch * channels * hw;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid_1d(total)), dim3(256), 0, s, x, y, dy, gamma, stats, scratch, dx, total,
                       channels, hw, groups, relu);
  }
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_upsample_bilinear_nchw_fwd(const float* x, float* y, int64_t planes, int h, int w, int factor,
                                             sm_stream_t stream) {
  if (!x || !y) return SM_ERR_BAD_ARG;
  if (planes < 1 || h < 1 || w < 1 || factor < 1) return SM_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(upsample_fwd_kernel, dim3(grid_1d(planes * h * w * factor * factor)), dim3(256), 0,
                     sm_hip_stream(stream), x, y, (long long)planes, h, w, factor);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_upsample_bilinear_nchw_bwd(const float* dy, float* dx, int64_t planes, int h, int w, int factor,
                                             sm_stream_t stream) {
  if (!dy || !dx) return SM_ERR_BAD_ARG;
  if (planes < 1 || h < 1 || w < 1 || factor < 1) return SM_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_1d(planes * h * w)), dim3(256), 0, sm_hip_stream(stream), dy, dx,
                     (long long)planes, h, w, factor);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                           float weight_decay, int first_step, sm_stream_t stream) {
  if (!param || !grad || !momentum_buf) return SM_ERR_BAD_ARG;
  if (n < 1) return SM_OK;
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_1d(n)), dim3(256), 0, sm_hip_stream(stream), param, grad, momentum_buf,
                     (long long)n, lr, momentum, weight_decay, first_step);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_sgd_multi(const void* items, const int32_t* blocks, int nblocks, float momentum, int first_step,
                            sm_stream_t stream) {
  if (!items || !blocks) return SM_ERR_BAD_ARG;
  if (nblocks < 1) return SM_OK;
  hipLaunchKernelGGL(sgd_multi_kernel, dim3(nblocks), dim3(256), 0, sm_hip_stream(stream), (const SgdItem*)items,
                     (const int2*)blocks, momentum, first_step);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
