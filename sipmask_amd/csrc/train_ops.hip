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


// ---------------------------------------------------------------- FCOS target assignment
// fcos_target_single, M/mmdet/models/anchor_heads/sipmask_head.py:773-857, for every (image, point) in one launch:
// among the ground-truth boxes whose (centre-sampled) interior holds the point and whose largest regression distance
// lies in the level's range, the one of smallest area wins (first index on ties, as the [S,G] min(dim=1) of the
// reference); label 0 = background.  Same f32 operations in the same order as the tensor formulation, so labels,
// targets and indices are bit-identical to it -- without the [S,G,4] broadcast tensors per image and the python loop.
__global__ __launch_bounds__(256) void fcos_target_kernel(const float* __restrict__ points, const float* __restrict__ pstride,
                                                          const float* __restrict__ lo, const float* __restrict__ hi,
                                                          const float* __restrict__ gtb, const int64_t* __restrict__ gtl,
                                                          const int32_t* __restrict__ ngt, int S, int gmax, int center,
                                                          float radius, int64_t* __restrict__ labels,
                                                          float* __restrict__ targets, int32_t* __restrict__ gt_index) {
  extern __shared__ float s_gt[];          // [gmax][5]: x1, y1, x2, y2, area
  const int b = blockIdx.y;
  const int G = min(ngt[b], gmax);
  for (int i = threadIdx.x; i < G; i += blockDim.x) {
    const float* q = gtb + ((long long)b * gmax + i) * 4;
    s_gt[i * 5 + 0] = q[0];
    s_gt[i * 5 + 1] = q[1];
    s_gt[i * 5 + 2] = q[2];
    s_gt[i * 5 + 3] = q[3];
    s_gt[i * 5 + 4] = __fmul_rn(__fadd_rn(__fsub_rn(q[2], q[0]), 1.f), __fadd_rn(__fsub_rn(q[3], q[1]), 1.f));   // (:791-792)
  }
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S) return;
  const float x = points[2 * p], y = points[2 * p + 1];
  const float r = __fmul_rn(pstride[p], radius), rlo = lo[p], rhi = hi[p];
  float best = 1e8f;                        // INF of the reference (:15)
  int bi = -1;
  float bl = 0.f, bt = 0.f, br = 0.f, bb = 0.f;
  for (int g = 0; g < G; ++g) {
    const float x1 = s_gt[g * 5], y1 = s_gt[g * 5 + 1], x2 = s_gt[g * 5 + 2], y2 = s_gt[g * 5 + 3];
    const float l = __fsub_rn(x, x1), t = __fsub_rn(y, y1), rr = __fsub_rn(x2, x), bo = __fsub_rn(y2, y);
    bool inside;
    if (center) {                           // (:801-835): centre box of half-size radius*stride, clipped to the gt
      const float cx = __fdiv_rn(__fadd_rn(x1, x2), 2.f), cy = __fdiv_rn(__fadd_rn(y1, y2), 2.f);
      const float cl = __fsub_rn(x, fmaxf(__fsub_rn(cx, r), x1)), ct = __fsub_rn(y, fmaxf(__fsub_rn(cy, r), y1));
      const float cr = __fsub_rn(fminf(__fadd_rn(cx, r), x2), x), cb = __fsub_rn(fminf(__fadd_rn(cy, r), y2), y);
      inside = fminf(fminf(cl, ct), fminf(cr, cb)) > 0.f;
    } else {
      inside = fminf(fminf(l, t), fminf(rr, bo)) > 0.f;
    }
    const float far = fmaxf(fmaxf(l, t), fmaxf(rr, bo));
    const bool ok = inside && far >= rlo && far <= rhi;     // (:841-844)
    const float area = ok ? s_gt[g * 5 + 4] : 1e8f;
    if (bi < 0 || area < best) {            // FIRST minimum (min(dim=1) of the reference); g = 0 is always taken, so a
      best = area;                          // point no gt claims gathers the ltrb of gt 0, as the reference does
      bi = g;
      bl = l, bt = t, br = rr, bb = bo;
    }
  }
  const long long o = (long long)b * S + p;
  const bool pos = bi >= 0 && best < 1e8f;
  labels[o] = pos ? gtl[(long long)b * gmax + bi] : 0;
  // bbox_targets = ltrb of the argmin gt (index 0 for a background point with G > 0; zeros when the image has no gt)
  targets[o * 4 + 0] = bl;
  targets[o * 4 + 1] = bt;
  targets[o * 4 + 2] = br;
  targets[o * 4 + 3] = bb;
  gt_index[o] = bi;
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

extern "C" int sm_fcos_target(const float* points, const float* point_stride, const float* range_lo, const float* range_hi,
                              const float* gt_boxes, const int64_t* gt_labels, const int32_t* ngt, int batch, int npoints,
                              int gmax, int center_sampling, float radius, int64_t* labels, float* bbox_targets,
                              int32_t* gt_index, sm_stream_t stream) {
  if (!points || !point_stride || !range_lo || !range_hi || !gt_boxes || !gt_labels || !ngt || !labels || !bbox_targets ||
      !gt_index)
    return SM_ERR_BAD_ARG;
  if (batch < 1 || npoints < 1 || gmax < 1 || gmax * 20 > 64 * 1024) return SM_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(fcos_target_kernel, dim3((npoints + 255) / 256, batch), dim3(256), (size_t)gmax * 20,
                     sm_hip_stream(stream), points, point_stride, range_lo, range_hi, gt_boxes, gt_labels, ngt, npoints, gmax,
                     center_sampling, radius, labels, bbox_targets, gt_index);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
