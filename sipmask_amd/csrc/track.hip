// SipMask-VIS tracking kernels: the per-detection embedding gather and the comprehensive
// matching score of V/mmdet/models/anchor_heads/sipmask_head.py:616-637 (+ compute_comp_scores :544-562).
// Both are tiny, latency-bound: N <= max_per_img detections, T tracked objects, 512-float embeddings.
#include "common.h"

namespace {

constexpr int TR_THREADS = 256;

// extract_box_feature_center_single (:768-781): feats f32 [B][h][w][C] (NHWC), one block per detection
__global__ __launch_bounds__(TR_THREADS) void track_gather_kernel(const float* __restrict__ feats,
                                                                  const float* __restrict__ det,
                                                                  const int32_t* __restrict__ ndet, int max_num, int h,
                                                                  int w, int C, float box_mul, float stride,
                                                                  float* __restrict__ out) {
  const int i = blockIdx.x, b = blockIdx.y;
  float* o = out + ((long long)b * max_num + i) * C;
  if (i >= ndet[b]) {
    for (int c = threadIdx.x; c < C; c += TR_THREADS) o[c] = 0.f;
    return;
  }
  const float* d = det + ((long long)b * max_num + i) * 5;
  // boxes are taken back to network-input coordinates first (res_det_bboxes[:, :4] *= scale_factor, :612-614)
  const float x1 = d[0] * box_mul, y1 = d[1] * box_mul, x2 = d[2] * box_mul, y2 = d[3] * box_mul;
  int cx = (int)floorf(__fdiv_rn(__fdiv_rn(__fadd_rn(x2, x1), 2.0f), stride));
  int cy = (int)floorf(__fdiv_rn(__fdiv_rn(__fadd_rn(y2, y1), 2.0f), stride));
  cx = min(max(cx, 0), w - 1);   // boxes are clamped to the image, so this only guards the last column/row
  cy = min(max(cy, 0), h - 1);
  const float* f = feats + (((long long)b * h + cy) * w + cx) * C;
  for (int c = threadIdx.x; c < C; c += TR_THREADS) o[c] = f[c];
}

__device__ __forceinline__ float iou_plus1_f(const float* a, const float* b) {
  const float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]) + 1.f, 0.f);
  const float h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]) + 1.f, 0.f);
  const float ov = w * h;
  const float aa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
  const float ab = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
  return ov / (aa + ab - ov);
}

// comp[n][0..T]: log_softmax([0, f_n . g_t]) + c0 log(score_n) + c1 IoU(+1) + c2 [label equal], dummy column 0
// has IoU 0 and label_delta 1.  One block per detection; also the row argmax (first maximum) and its value.
__global__ __launch_bounds__(TR_THREADS) void track_match_kernel(const float* __restrict__ det_feats,
                                                                 const float* __restrict__ prev_feats,
                                                                 const float* __restrict__ det,
                                                                 const int64_t* __restrict__ det_labels,
                                                                 const float* __restrict__ prev_boxes,
                                                                 const int64_t* __restrict__ prev_labels, int T, int C,
                                                                 float c0, float c1, float c2, float* __restrict__ comp,
                                                                 int32_t* __restrict__ match_id,
                                                                 float* __restrict__ match_score) {
  extern __shared__ float s_f[];             // C floats: this detection's embedding
  __shared__ float s_red[TR_THREADS / 64];
  __shared__ int s_idx[TR_THREADS / 64];
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < C; c += TR_THREADS) s_f[c] = det_feats[(long long)n * C + c];
  __syncthreads();
  float* row = comp + (long long)n * (T + 1);
  // pass 1: raw products into the row (column 0 = the dummy logit 0)
  if (tid == 0) row[0] = 0.f;
  for (int t = tid; t < T; t += TR_THREADS) {
    const float* g = prev_feats + (long long)t * C;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc = fmaf(s_f[c], g[c], acc);
    row[t + 1] = acc;
  }
  __syncthreads();
  // log-sum-exp over T+1 logits
  float mx = -INFINITY;
  for (int t = tid; t <= T; t += TR_THREADS) mx = fmaxf(mx, row[t]);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
  if ((tid & 63) == 0) s_red[tid >> 6] = mx;
  __syncthreads();
  mx = s_red[0];
  for (int wv = 1; wv < TR_THREADS / 64; ++wv) mx = fmaxf(mx, s_red[wv]);
  __syncthreads();
  float se = 0.f;
  for (int t = tid; t <= T; t += TR_THREADS) se += expf(row[t] - mx);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) se += __shfl_xor(se, d, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = se;
  __syncthreads();
  se = 0.f;
  for (int wv = 0; wv < TR_THREADS / 64; ++wv) se += s_red[wv];
  const float lse = mx + logf(se);
  __syncthreads();
  // pass 2: comprehensive score, running arg max (first maximum wins)
  const float* db = det + (long long)n * 5;
  const float ls = c0 * logf(db[4]);
  const int64_t lab = det_labels[n];
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int t = tid; t <= T; t += TR_THREADS) {
    float v = row[t] - lse + ls;
    if (t == 0) {
      v += c2;
    } else {
      v += c1 * iou_plus1_f(db, prev_boxes + (long long)(t - 1) * 5) + (prev_labels[t - 1] == lab ? c2 : 0.f);
    }
    row[t] = v;
    if (v > best) {
      best = v;
      bi = t;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const float ob = __shfl_xor(best, d, 64);
    const int oi = __shfl_xor(bi, d, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  if ((tid & 63) == 0) {
    s_red[tid >> 6] = best;
    s_idx[tid >> 6] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int wv = 1; wv < TR_THREADS / 64; ++wv)
      if (s_red[wv] > best || (s_red[wv] == best && s_idx[wv] < bi)) {
        best = s_red[wv];
        bi = s_idx[wv];
      }
    match_id[n] = bi;
    match_score[n] = best;
  }
}

}  // namespace

extern "C" int sm_track_gather(const float* track_feats, const float* det, const int32_t* ndet, int batch, int max_num,
                               int h, int w, int channels, float box_mul, float stride, float* out, sm_stream_t stream) {
  if (!track_feats || !det || !ndet || !out) return SM_ERR_BAD_ARG;
  if (batch < 1 || max_num < 1 || h < 1 || w < 1 || channels < 1 || !(stride > 0.f)) return SM_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(track_gather_kernel, dim3(max_num, batch), dim3(TR_THREADS), 0, sm_hip_stream(stream), track_feats,
                     det, ndet, max_num, h, w, channels, box_mul, stride, out);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_track_match(const float* det_feats, const float* prev_feats, const float* det,
                              const int64_t* det_labels, const float* prev_boxes, const int64_t* prev_labels, int n,
                              int t, int channels, float coeff_score, float coeff_iou, float coeff_label, float* comp,
                              int32_t* match_id, float* match_score, sm_stream_t stream) {
  if (!det_feats || !prev_feats || !det || !det_labels || !prev_boxes || !prev_labels || !comp || !match_id ||
      !match_score)
    return SM_ERR_BAD_ARG;
  if (n < 0 || t < 1 || channels < 1 || channels > 8192) return SM_ERR_BAD_SHAPE;
  if (n == 0) return SM_OK;
  hipLaunchKernelGGL(track_match_kernel, dim3(n), dim3(TR_THREADS), channels * sizeof(float), sm_hip_stream(stream),
                     det_feats, prev_feats, det, det_labels, prev_boxes, prev_labels, t, channels, coeff_score, coeff_iou,
                     coeff_label, comp, match_id, match_score);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
