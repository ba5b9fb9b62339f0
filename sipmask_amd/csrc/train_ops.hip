// Training-side elementwise ops of the SipMask head: CropSplit / CropSplitGt and the fused
// sigmoid focal loss.  All are pure streaming (HBM bound, one read + one write per element).
#include <float.h>

#include "common.h"

namespace {

struct RoiCell {
  bool inside;
  int cell;
};

// index math of M/mmdet/ops/crop/src/crop_split_cuda_kernel.cu:34-52 (float box, integer pixel,
// roi_w computed through double because of the 0.1 literal)
__device__ __forceinline__ RoiCell roi_cell(const float* __restrict__ rois, int n, int ph, int pw, int c) {
  const float4 r = *reinterpret_cast<const float4*>(rois + (long long)n * 4);
  RoiCell o;
  const float fw = (float)pw, fh = (float)ph;
  o.inside = (fw >= r.x) & (fh >= r.y) & (fw < r.z) & (fh < r.w);
  o.cell = 0;
  if (o.inside) {
    const float roi_w = (float)(((double)__fsub_rn(r.z, r.x) + 0.1) / (double)c);
    const float roi_h = (float)(((double)__fsub_rn(r.w, r.y) + 0.1) / (double)c);
    const int iw = (int)__fdiv_rn(__fsub_rn(fw, r.x), roi_w);
    const int ih = (int)__fdiv_rn(__fsub_rn(fh, r.y), roi_h);
    o.cell = ih * c + iw;
  }
  return o;
}

__global__ void crop_split_fwd_kernel(const float* __restrict__ data, const float* __restrict__ rois,
                                      float* __restrict__ out, int H, int W, int c, int N) {
  const long long count = (long long)H * W * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % N);
    const int pw = (int)((i / N) % W);
    const int ph = (int)(i / N / W);
    const RoiCell rc = roi_cell(rois, n, ph, pw, c);
    float v = 0.f;
    if (rc.inside && rc.cell < c * c) v = data[i + (long long)rc.cell * count];
    out[i] = v;  // fully written: the wrapper needs no new_zeros (crop_split.py:22)
  }
}

__global__ void crop_split_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ rois,
                                      float* __restrict__ gin, int H, int W, int c, int N) {
  const long long count = (long long)H * W * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % N);
    const int pw = (int)((i / N) % W);
    const int ph = (int)(i / N / W);
    const RoiCell rc = roi_cell(rois, n, ph, pw, c);
    const float g = gout[i];
    // the scatter is 1:1 (kernel.cu:124 atomicAdd never collides): plain stores, all planes written
    for (int q = 0; q < c * c; ++q) gin[i + (long long)q * count] = (rc.inside && q == rc.cell) ? g : 0.f;
  }
}

__global__ void crop_split_gt_fwd_kernel(const float* __restrict__ data, const float* __restrict__ rois,
                                         float* __restrict__ out, int H, int W, int N) {
  const long long count = (long long)H * W * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % N);
    const int pw = (int)((i / N) % W);
    const int ph = (int)(i / N / W);
    const float4 r = *reinterpret_cast<const float4*>(rois + (long long)n * 4);
    const float fw = (float)pw, fh = (float)ph;
    const bool inside = (fw >= r.x) & (fh >= r.y) & (fw < r.z) & (fh < r.w);
    out[i] = inside ? data[i] : 0.f;
  }
}

// SigmoidFocalLossForward/Backward, sigmoid_focal_loss_cuda.cu:24-97
__global__ void focal_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets,
                                 float* __restrict__ losses, long long total, int C, float gamma, float alpha) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / C;
    const int d = (int)(i - n * C);
    const int t = (int)targets[n];
    const float c1 = (t == d + 1) ? 1.f : 0.f;
    const float c2 = (t >= 0 && t != d + 1) ? 1.f : 0.f;
    const float x = logits[i];
    const float p = 1.f / (1.f + expf(-x));
    const float term1 = powf(1.f - p, gamma) * logf(fmaxf(p, FLT_MIN));
    const float xp = x >= 0.f ? x : 0.f;
    const float term2 = powf(p, gamma) * (-xp - logf(1.f + expf(x - 2.f * xp)));
    losses[i] = -c1 * term1 * alpha - c2 * term2 * (1.f - alpha);
  }
}

__global__ void focal_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets,
                                 const float* __restrict__ dl, float* __restrict__ dx, long long total, int C,
                                 float gamma, float alpha) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / C;
    const int d = (int)(i - n * C);
    const int t = (int)targets[n];
    const float c1 = (t == d + 1) ? 1.f : 0.f;
    const float c2 = (t >= 0 && t != d + 1) ? 1.f : 0.f;
    const float x = logits[i];
    const float p = 1.f / (1.f + expf(-x));
    const float term1 = powf(1.f - p, gamma) * (1.f - p - p * gamma * logf(fmaxf(p, FLT_MIN)));
    const float xp = x >= 0.f ? x : 0.f;
    const float term2 = powf(p, gamma) * ((-xp - logf(1.f + expf(x - 2.f * xp))) * (1.f - p) * gamma - p);
    dx[i] = (-c1 * term1 * alpha - c2 * term2 * (1.f - alpha)) * dl[i];
  }
}

inline int grid_for(long long n) {
  long long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int sm_crop_split_fwd(const float* data, const float* rois, float* out, int h, int w, int c, int n,
                                 sm_stream_t stream) {
  if (!data || !rois || !out || h < 1 || w < 1 || c < 1 || n < 0) return SM_ERR_BAD_ARG;
  if (n == 0) return SM_OK;
  hipLaunchKernelGGL(crop_split_fwd_kernel, dim3(grid_for((long long)h * w * n)), dim3(256), 0, sm_hip_stream(stream),
                     data, rois, out, h, w, c, n);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_crop_split_bwd(const float* grad_out, const float* rois, float* grad_in, int h, int w, int c, int n,
                                 sm_stream_t stream) {
  if (!grad_out || !rois || !grad_in || h < 1 || w < 1 || c < 1 || n < 0) return SM_ERR_BAD_ARG;
  if (n == 0) return SM_OK;
  hipLaunchKernelGGL(crop_split_bwd_kernel, dim3(grid_for((long long)h * w * n)), dim3(256), 0, sm_hip_stream(stream),
                     grad_out, rois, grad_in, h, w, c, n);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_crop_split_gt_fwd(const float* data, const float* rois, float* out, int h, int w, int n,
                                    sm_stream_t stream) {
  if (!data || !rois || !out || h < 1 || w < 1 || n < 0) return SM_ERR_BAD_ARG;
  if (n == 0) return SM_OK;
  hipLaunchKernelGGL(crop_split_gt_fwd_kernel, dim3(grid_for((long long)h * w * n)), dim3(256), 0,
                     sm_hip_stream(stream), data, rois, out, h, w, n);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_sigmoid_focal_loss_fwd(const float* logits, const int64_t* targets, float* losses, int n, int c,
                                         float gamma, float alpha, sm_stream_t stream) {
  if (!logits || !targets || !losses || n < 0 || c < 1) return SM_ERR_BAD_ARG;
  if (n == 0) return SM_OK;
  hipLaunchKernelGGL(focal_fwd_kernel, dim3(grid_for((long long)n * c)), dim3(256), 0, sm_hip_stream(stream), logits,
                     targets, losses, (long long)n * c, c, gamma, alpha);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_sigmoid_focal_loss_bwd(const float* logits, const int64_t* targets, const float* d_losses,
                                         float* d_logits, int n, int c, float gamma, float alpha,
                                         sm_stream_t stream) {
  if (!logits || !targets || !d_losses || !d_logits || n < 0 || c < 1) return SM_ERR_BAD_ARG;
  if (n == 0) return SM_OK;
  hipLaunchKernelGGL(focal_bwd_kernel, dim3(grid_for((long long)n * c)), dim3(256), 0, sm_hip_stream(stream), logits,
                     targets, d_losses, d_logits, (long long)n * c, c, gamma, alpha);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_version(void) { return SM_ABI_VERSION; }

extern "C" const char* sm_strerror(int status) {
  switch (status) {
    case SM_OK: return "ok";
    case SM_ERR_BAD_SHAPE: return "bad shape";
    case SM_ERR_BAD_ARG: return "bad argument (null pointer / inconsistent option)";
    case SM_ERR_LAUNCH: return "HIP launch failure";
    case SM_ERR_UNSUPPORTED: return "unsupported configuration";
    case SM_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown status";
  }
}
