// Fused SipMask mask loss (training row a13, sipmask_head.py:443-461):
//
//   pred = CropSplit(sigmoid(basis . cof_q), box)      -- 4 x [Hm,Wm,N] probability volumes in the reference
//   gt   = CropSplitGt(gt_mask[idx_gt], box)
//   S[n] = sum over pixels of F.binary_cross_entropy(pred, gt)[.., n]
//
// The reference materialises 4*Hm*Wm*N floats (sigmoid of a [Hm*Wm,32]x[32,N] product per quadrant), crops,
// and reduces.  Outside its box a detection contributes BCE(0,0) = 0, so only the box pixels matter: the
// kernels below never materialise anything -- a pixel's logit is one 32-term dot product with the coefficient
// block of the quadrant (cell) the pixel falls in (crop_split_cuda_kernel.cu:34-52), evaluated where needed:
//   fwd       : one block per detection, box pixels strided over the threads, block reduction -> S[n]
//   bwd (cof) : one block per detection, per quadrant LDS accumulation of dz * basis[pix]   -> grad_cof[n][128]
//   bwd(basis): pixel-major, every thread owns one pixel and loops over the detections (staged in LDS);
//               no atomics, the whole grad_basis tensor is written once (zeros outside every box)
// HBM/L2-bound gather work (32 floats per (pixel, detection) hit); nothing here is GEMM shaped once the crop
// is applied first.
#include "common.h"

namespace {

constexpr int ML_THREADS = 256;
constexpr int ML_DETS = 32;   // detections staged per pass in the pixel-major kernel
constexpr int ML_CHUNKS = 16; // blocks per detection in the detection-major kernels: a training image has 50-500 positives
                              // with boxes of up to the whole grid, so one block per detection leaves most CUs idle and
                              // the largest box sets the time (0.67 ms per image before the split)

struct MLArgs {
  const float* basis;
  long long pix_stride, ch_stride;   // [32][Hm][Wm] (1, Hm*Wm) or [Hm][Wm][32] (32, 1)
  const float* cof;                  // [N][128]
  const float* boxes;                // [N][4] crop boxes in basis-grid coordinates (bbox_dt, sipmask_head.py:407-414)
  const uint8_t* gt;                 // [G][Hm][Wm] 0/1 (gt_mask_new, :432-436)
  const int64_t* idx_gt;             // [N]
  int n, hm, wm;
};

struct BoxGeo {
  float x1, y1, x2, y2, rw, rh;
  int xlo, xhi, ylo, yhi;   // integer pixel bounds of {pw >= x1, pw < x2, ph >= y1, ph < y2} clipped to the grid
};

__device__ __forceinline__ BoxGeo box_geo(const float* b, int hm, int wm) {
  BoxGeo g;
  g.x1 = b[0], g.y1 = b[1], g.x2 = b[2], g.y2 = b[3];
  // roi_w = (x2 - x1 + 0.1) / 2 evaluated in double, stored as float (crop_split_cuda_kernel.cu:47-48)
  g.rw = (float)(((double)__fsub_rn(g.x2, g.x1) + 0.1) / 2.0);
  g.rh = (float)(((double)__fsub_rn(g.y2, g.y1) + 0.1) / 2.0);
  auto ci = [](float v) { return (int)fminf(fmaxf(ceilf(v), -1e6f), 1e6f); };   // NaN -> -1e6: empty range
  g.xlo = max(ci(g.x1), 0);
  g.ylo = max(ci(g.y1), 0);
  g.xhi = min(ci(g.x2) - 1, wm - 1);
  g.yhi = min(ci(g.y2) - 1, hm - 1);
  return g;
}

// cell (quadrant) of an inside pixel, or -1
__device__ __forceinline__ int cell_of(const BoxGeo& g, int px, int py) {
  const float fw = (float)px, fh = (float)py;
  if (!(fw >= g.x1 && fh >= g.y1 && fw < g.x2 && fh < g.y2)) return -1;
  const int iw = (int)__fdiv_rn(__fsub_rn(fw, g.x1), g.rw);
  const int ih = (int)__fdiv_rn(__fsub_rn(fh, g.y1), g.rh);
  const int c = ih * 2 + iw;
  return c < 4 ? c : -2;   // cells past 2x2 are cropped to 0 by the reference kernel (:50-52)
}

__device__ __forceinline__ float logit_of(const MLArgs& a, long long pix, const float* cq) {
  const float* f = a.basis + pix * a.pix_stride;
  float z = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) z = fmaf(f[(long long)k * a.ch_stride], cq[k], z);
  return z;
}

__device__ __forceinline__ float sigmoid_t(float z) { return 1.f / (1.f + expf(-z)); }

// d(BCE)/dz with torch's formulas: binary_cross_entropy_backward = (p - t) / max(p (1-p), 1e-12),
// then sigmoid_backward = * p (1-p)
__device__ __forceinline__ float dbce_dz(float p, float t) {
  const float pq = p * (1.f - p);
  return (p - t) / fmaxf(pq, 1e-12f) * pq;
}

__global__ __launch_bounds__(ML_THREADS) void mask_loss_fwd_kernel(const MLArgs a, float* __restrict__ out) {
  __shared__ float s_cof[128];
  __shared__ float s_red[ML_THREADS / 64];
  const int n = blockIdx.x, tid = threadIdx.x;
  if (tid < 128) s_cof[tid] = a.cof[(long long)n * 128 + tid];
  __syncthreads();
  const BoxGeo g = box_geo(a.boxes + (long long)n * 4, a.hm, a.wm);
  const int bw = g.xhi - g.xlo + 1, bh = g.yhi - g.ylo + 1;
  const uint8_t* gt = a.gt + a.idx_gt[n] * (long long)a.hm * a.wm;
  float acc = 0.f;
  if (bw > 0 && bh > 0) {
    const int npix = bw * bh;
    const int per = (npix + ML_CHUNKS - 1) / ML_CHUNKS;
    const int t1 = min(npix, ((int)blockIdx.y + 1) * per);
    for (int t = (int)blockIdx.y * per + tid; t < t1; t += ML_THREADS) {
      const int py = g.ylo + t / bw, px = g.xlo + t % bw;
      const int c = cell_of(g, px, py);
      if (c == -1) continue;
      const long long pix = (long long)py * a.wm + px;
      const float tg = gt[pix] ? 1.f : 0.f;
      // a cropped (cell >= 4) prediction is exactly 0: BCE(0, t) = t * 100 (log clamped at -100)
      const float p = c >= 0 ? sigmoid_t(logit_of(a, pix, s_cof + c * 32)) : 0.f;
      acc -= tg != 0.f ? fmaxf(logf(p), -100.f) : fmaxf(logf(1.f - p), -100.f);
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < ML_THREADS / 64; ++w) t += s_red[w];
    if (t != 0.f) atomicAdd(out + n, t);        // out is zeroed by the host entry point
  }
}

__global__ __launch_bounds__(ML_THREADS) void mask_loss_bwd_cof_kernel(const MLArgs a, const float* __restrict__ gsum,
                                                                       float* __restrict__ gcof) {
  __shared__ float s_cof[128];
  __shared__ float s_acc[128];
  const int n = blockIdx.x, tid = threadIdx.x;
  if (tid < 128) {
    s_cof[tid] = a.cof[(long long)n * 128 + tid];
    s_acc[tid] = 0.f;
  }
  __syncthreads();
  const BoxGeo g = box_geo(a.boxes + (long long)n * 4, a.hm, a.wm);
  const int bw = g.xhi - g.xlo + 1, bh = g.yhi - g.ylo + 1;
  const uint8_t* gt = a.gt + a.idx_gt[n] * (long long)a.hm * a.wm;
  const float go = gsum[n];
  bool block_any = false;
  if (bw > 0 && bh > 0) {
    const int npix = bw * bh;
    const int per = (npix + ML_CHUNKS - 1) / ML_CHUNKS;
    const int t0 = (int)blockIdx.y * per, t1 = min(npix, t0 + per);
    block_any = t0 < t1;
    // quadrant by quadrant, so that a thread accumulates one 32-vector in registers at a time
    for (int q = 0; q < 4; ++q) {
      float acc[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) acc[k] = 0.f;
      bool any = false;
      for (int t = t0 + tid; t < t1; t += ML_THREADS) {
        const int py = g.ylo + t / bw, px = g.xlo + t % bw;
        if (cell_of(g, px, py) != q) continue;
        const long long pix = (long long)py * a.wm + px;
        const float p = sigmoid_t(logit_of(a, pix, s_cof + q * 32));
        const float dz = go * dbce_dz(p, gt[pix] ? 1.f : 0.f);
        const float* f = a.basis + pix * a.pix_stride;
#pragma unroll
        for (int k = 0; k < 32; ++k) acc[k] = fmaf(dz, f[(long long)k * a.ch_stride], acc[k]);
        any = true;
      }
      if (any) {
#pragma unroll
        for (int k = 0; k < 32; ++k) atomicAdd(&s_acc[q * 32 + k], acc[k]);
      }
    }
  }
  __syncthreads();
  if (tid < 128 && block_any && s_acc[tid] != 0.f) atomicAdd(gcof + (long long)n * 128 + tid, s_acc[tid]);   // gcof zeroed by the host
}

struct DetStage {
  BoxGeo g;
  float go;
  long long gt_off;
};

__global__ __launch_bounds__(ML_THREADS) void mask_loss_bwd_basis_kernel(const MLArgs a, const float* __restrict__ gsum,
                                                                         float* __restrict__ gbasis) {
  __shared__ float s_cof[ML_DETS * 128];
  __shared__ DetStage s_det[ML_DETS];
  // tile: 64 x 4 pixels, lanes along x
  const int tid = threadIdx.x;
  const int px = blockIdx.x * 64 + (tid & 63), py = blockIdx.y * 4 + (tid >> 6);
  const int tx0 = blockIdx.x * 64, ty0 = blockIdx.y * 4;
  const bool live = px < a.wm && py < a.hm;
  const long long pix = (long long)py * a.wm + px;
  float f[32], acc[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    f[k] = live ? a.basis[pix * a.pix_stride + (long long)k * a.ch_stride] : 0.f;
    acc[k] = 0.f;
  }
  for (int n0 = 0; n0 < a.n; n0 += ML_DETS) {
    const int nn = min(ML_DETS, a.n - n0);
    __syncthreads();
    if (tid < nn) {
      DetStage d;
      d.g = box_geo(a.boxes + (long long)(n0 + tid) * 4, a.hm, a.wm);
      d.go = gsum[n0 + tid];
      d.gt_off = a.idx_gt[n0 + tid] * (long long)a.hm * a.wm;
      s_det[tid] = d;
    }
    for (int i = tid; i < nn * 128; i += ML_THREADS) s_cof[i] = a.cof[(long long)n0 * 128 + i];
    __syncthreads();
    for (int n = 0; n < nn; ++n) {
      const BoxGeo& g = s_det[n].g;
      // block-uniform tile / box overlap
      if (g.xhi < tx0 || g.xlo > tx0 + 63 || g.yhi < ty0 || g.ylo > ty0 + 3) continue;
      if (!live) continue;
      const int c = cell_of(g, px, py);
      if (c < 0) continue;
      const float* cq = s_cof + n * 128 + c * 32;
      float z = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) z = fmaf(f[k], cq[k], z);
      const float p = sigmoid_t(z);
      const float dz = s_det[n].go * dbce_dz(p, a.gt[s_det[n].gt_off + pix] ? 1.f : 0.f);
#pragma unroll
      for (int k = 0; k < 32; ++k) acc[k] = fmaf(dz, cq[k], acc[k]);
    }
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < 32; ++k) gbasis[pix * a.pix_stride + (long long)k * a.ch_stride] = acc[k];
  }
}

int fill_args(MLArgs* a, const float* basis, int basis_hwc, const float* cof, const float* boxes, const uint8_t* gt,
              const int64_t* idx_gt, int n, int hm, int wm) {
  if (!basis || !cof || !boxes || !gt || !idx_gt) return SM_ERR_BAD_ARG;
  if (n < 0 || hm < 1 || wm < 1) return SM_ERR_BAD_SHAPE;
  a->basis = basis;
  a->pix_stride = basis_hwc ? 32 : 1;
  a->ch_stride = basis_hwc ? 1 : (long long)hm * wm;
  a->cof = cof;
  a->boxes = boxes;
  a->gt = gt;
  a->idx_gt = idx_gt;
  a->n = n, a->hm = hm, a->wm = wm;
  return SM_OK;
}

}  // namespace

extern "C" int sm_mask_loss_fwd(const float* basis, int basis_hwc, const float* cof, const float* boxes,
                                const uint8_t* gt, const int64_t* idx_gt, int n, int hm, int wm, float* bce_sum,
                                sm_stream_t stream) {
  MLArgs a;
  const int st = fill_args(&a, basis, basis_hwc, cof, boxes, gt, idx_gt, n, hm, wm);
  if (st != SM_OK) return st;
  if (!bce_sum) return SM_ERR_BAD_ARG;
  if (n == 0) return SM_OK;
  if (sm_zero_async(bce_sum, sizeof(float) * n, sm_hip_stream(stream)) != hipSuccess) return SM_ERR_LAUNCH;
  hipLaunchKernelGGL(mask_loss_fwd_kernel, dim3(n, ML_CHUNKS), dim3(ML_THREADS), 0, sm_hip_stream(stream), a, bce_sum);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_mask_loss_bwd(const float* basis, int basis_hwc, const float* cof, const float* boxes,
                                const uint8_t* gt, const int64_t* idx_gt, int n, int hm, int wm,
                                const float* grad_sum, float* grad_cof, float* grad_basis, sm_stream_t stream) {
  MLArgs a;
  const int st = fill_args(&a, basis, basis_hwc, cof, boxes, gt, idx_gt, n, hm, wm);
  if (st != SM_OK) return st;
  if (!grad_sum) return SM_ERR_BAD_ARG;
  hipStream_t s = sm_hip_stream(stream);
  if (grad_cof && n > 0) {
    if (sm_zero_async(grad_cof, sizeof(float) * 128 * n, s) != hipSuccess) return SM_ERR_LAUNCH;
    hipLaunchKernelGGL(mask_loss_bwd_cof_kernel, dim3(n, ML_CHUNKS), dim3(ML_THREADS), 0, s, a, grad_sum, grad_cof);
  }
  if (grad_basis) {
    hipLaunchKernelGGL(mask_loss_bwd_basis_kernel, dim3(sm_cdiv(wm, 64), sm_cdiv(hm, 4)), dim3(ML_THREADS), 0, s, a,
                       grad_sum, grad_basis);
  }
  SM_LAUNCH_CHECK();
  return SM_OK;
}
