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
// has IoU 0 and label_delta 1; also the row argmax (first maximum) and its value.  The whole block works on ONE
// detection (feat / db / lab = its embedding, box, label); s_f: C floats of LDS.  Shared by the per-frame kernel and
// the clip tracker so that both assign identical ids.
__device__ __forceinline__ void comp_row(const float* __restrict__ feat, const float* __restrict__ prev_feats,
                                         const float* __restrict__ db, const int64_t lab,
                                         const float* __restrict__ prev_boxes, const int64_t* __restrict__ prev_labels,
                                         int T, int C, float c0, float c1, float c2, float* __restrict__ row, float* s_f,
                                         float* s_red, int* s_idx, int* out_id, float* out_score) {
  const int tid = threadIdx.x;
  for (int c = tid; c < C; c += TR_THREADS) s_f[c] = feat[c];
  __syncthreads();
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
  const float ls = c0 * logf(db[4]);
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
    *out_id = bi;
    *out_score = best;
  }
  __syncthreads();
}

// one block per detection
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
  const int n = blockIdx.x;
  comp_row(det_feats + (long long)n * C, prev_feats, det + (long long)n * 5, det_labels[n], prev_boxes, prev_labels, T, C, c0,
           c1, c2, comp + (long long)n * (T + 1), s_f, s_red, s_idx, match_id + n, match_score + n);
}

// The tracker of a whole clip on the device (V/mmdet/models/anchor_heads/sipmask_head.py:616-667 for frames 0..T-1 in
// order): ONE block walks the frames.  Per frame every detection's comprehensive scores against the object memory are
// formed by one WAVE (16 waves = 16 detections at a time; a lane holds channels lane, lane+64, ... of the embedding, a
// dot product is 8 coalesced loads + a shuffle tree), then the reference's sequential identity assignment -- a detection
// whose best column is the dummy opens a new object, otherwise it claims object o if its score beats the best claim so
// far (the memory slot takes the LAST winner's embedding and box), losers get -1 -- and the memory append.  The memory
// (feats [cap][C], boxes [cap][5], labels [cap], count) lives in HBM between calls: no host round trip per frame, one
// D2H of the ids per clip.  Frames without detections are skipped (as the host loop did).
constexpr int TC_MAXDET = 64, TC_THREADS = 1024, TC_MAXCJ = 16;   // C <= 64 * TC_MAXCJ channels
__global__ __launch_bounds__(TC_THREADS) void track_clip_kernel(const float* __restrict__ det_feats, const float* __restrict__ det,
                                                                const int64_t* __restrict__ det_labels,
                                                                const int32_t* __restrict__ ndet,
                                                                const int32_t* __restrict__ is_first, int nframes, int max_num,
                                                                int C, float c0, float c1, float c2, float* __restrict__ mem_feats,
                                                                float* __restrict__ mem_boxes, int64_t* __restrict__ mem_labels,
                                                                int32_t* __restrict__ mem_count, int cap,
                                                                float* __restrict__ comp_ws, int32_t* __restrict__ ids) {
  extern __shared__ float s_dyn[];           // [cap] best claim per object | [cap] its detection (int) | 16 x [cap + 1] score rows
  float* s_best = s_dyn;
  int* s_win = reinterpret_cast<int*>(s_best + cap);
  float* s_rows = s_dyn + 2 * cap;
  __shared__ int s_mid[TC_MAXDET], s_id[TC_MAXDET], s_new[TC_MAXDET], s_cnt, s_nnew;
  __shared__ float s_msc[TC_MAXDET];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cj = C >> 6;                     // channels per lane (C % 64 == 0)
  if (tid == 0) s_cnt = *mem_count;
  __syncthreads();
  for (int t = 0; t < nframes; ++t) {
    const int n = min(min(ndet[t], max_num), TC_MAXDET);
    int32_t* idt = ids + (long long)t * max_num;
    for (int i = tid; i < max_num; i += TC_THREADS) idt[i] = -1;
    if (n <= 0) continue;                                                   // block-uniform
    const float* ft = det_feats + (long long)t * max_num * C;
    const float* bt = det + (long long)t * max_num * 5;
    const int64_t* lt = det_labels + (long long)t * max_num;
    const int cnt = s_cnt;
    if (is_first[t] != 0 || cnt == 0) {                                     // the frame's detections ARE the memory
      const int m = min(n, cap);
      for (int j = tid; j < m * C; j += TC_THREADS) mem_feats[j] = ft[j];
      for (int j = tid; j < m * 5; j += TC_THREADS) mem_boxes[j] = bt[j];
      for (int j = tid; j < m; j += TC_THREADS) {
        mem_labels[j] = lt[j];
        idt[j] = j;
      }
      __syncthreads();
      if (tid == 0) s_cnt = m;
      __syncthreads();
      continue;
    }
    // ---- comprehensive scores: one wave per detection.  comp[i][0..cnt]: log_softmax([0, f_i . g_o]) + c0 log(score_i)
    // + c1 IoU(+1) + c2 [label equal], dummy column 0 with IoU 0 and label term c2 (compute_comp_scores, :544-562)
    for (int i = wave; i < n; i += TC_THREADS / 64) {
      float f[TC_MAXCJ];
#pragma unroll
      for (int j = 0; j < TC_MAXCJ; ++j) f[j] = j < cj ? ft[(long long)i * C + lane + 64 * j] : 0.f;
      float* row = s_rows + (long long)wave * (cap + 1);           // LDS: in order inside a wave, no cache in between
      if (lane == 0) row[0] = 0.f;
      for (int o = 0; o < cnt; ++o) {
        const float* g = mem_feats + (long long)o * C + lane;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < TC_MAXCJ; ++j)
          if (j < cj) acc = fmaf(f[j], g[64 * j], acc);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if (lane == 0) row[o + 1] = acc;
      }
      __builtin_amdgcn_wave_barrier();                                    // row[] written by lane 0, read by all lanes
      float mx = -INFINITY;
      for (int o = lane; o <= cnt; o += 64) mx = fmaxf(mx, row[o]);
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
      float se = 0.f;
      for (int o = lane; o <= cnt; o += 64) se += expf(row[o] - mx);
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) se += __shfl_xor(se, d, 64);
      const float lse = mx + logf(se);
      const float* db = bt + i * 5;
      const float ls = c0 * logf(db[4]);
      const int64_t lab = lt[i];
      float best = -INFINITY;
      int bi = 0x7fffffff;
      for (int o = lane; o <= cnt; o += 64) {
        float v = row[o] - lse + ls;
        if (o == 0) v += c2;
        else v += c1 * iou_plus1_f(db, mem_boxes + (long long)(o - 1) * 5) + (mem_labels[o - 1] == lab ? c2 : 0.f);
        if (v > best) {                                                     // ascending o per lane: first maximum wins
          best = v;
          bi = o;
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
      if (lane == 0) {
        s_mid[i] = bi;
        s_msc[i] = best;
      }
    }
    for (int o = tid; o < cnt; o += TC_THREADS) {
      s_best[o] = -100.f;
      s_win[o] = -1;
    }
    __syncthreads();
    if (tid == 0) {                                                         // the reference's sequential loop (:641-660)
      int nnew = 0;
      for (int i = 0; i < n; ++i) {
        const int m = s_mid[i];
        int id = -1;
        if (m == 0) {
          if (cnt + nnew < cap) {
            id = cnt + nnew;
            s_new[nnew++] = i;
          }
        } else if (s_msc[i] > s_best[m - 1]) {
          id = m - 1;
          s_best[m - 1] = s_msc[i];
          s_win[m - 1] = i;
        }
        s_id[i] = id;
      }
      s_nnew = nnew;
    }
    __syncthreads();
    for (int i = tid; i < n; i += TC_THREADS) idt[i] = s_id[i];
    // memory update: slot o takes its last winner (wave per slot); new objects are appended in detection order
    for (int o = wave; o < cnt; o += TC_THREADS / 64) {
      const int w = s_win[o];
      if (w < 0) continue;                                                  // wave-uniform (LDS value)
      for (int c = lane; c < C; c += 64) mem_feats[(long long)o * C + c] = ft[(long long)w * C + c];
      if (lane < 5) mem_boxes[o * 5 + lane] = bt[w * 5 + lane];
    }
    const int nnew = s_nnew;
    for (int k = wave; k < nnew; k += TC_THREADS / 64) {
      const int w = s_new[k], o = cnt + k;
      for (int c = lane; c < C; c += 64) mem_feats[(long long)o * C + c] = ft[(long long)w * C + c];
      if (lane < 5) mem_boxes[o * 5 + lane] = bt[w * 5 + lane];
      if (lane == 0) mem_labels[o] = lt[w];
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) s_cnt = cnt + nnew;
    __syncthreads();
  }
  if (tid == 0) *mem_count = s_cnt;
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

extern "C" int sm_track_clip(const float* det_feats, const float* det, const int64_t* det_labels, const int32_t* ndet,
                             const int32_t* is_first, int nframes, int max_num, int channels, float coeff_score,
                             float coeff_iou, float coeff_label, float* mem_feats, float* mem_boxes, int64_t* mem_labels,
                             int32_t* mem_count, int capacity, float* comp_ws, int32_t* ids, sm_stream_t stream) {
  if (!det_feats || !det || !det_labels || !ndet || !is_first || !mem_feats || !mem_boxes || !mem_labels || !mem_count ||
      !ids)
    return SM_ERR_BAD_ARG;
  if (nframes < 1 || max_num < 1 || max_num > TC_MAXDET || channels < 64 || channels % 64 != 0 || channels > 64 * TC_MAXCJ ||
      capacity < max_num || capacity > 4096)
    return SM_ERR_BAD_SHAPE;
  const size_t lds = (size_t)(2 * capacity + (TC_THREADS / 64) * (capacity + 1)) * sizeof(float);
  if (lds > 150 * 1024) return SM_ERR_UNSUPPORTED;
  if (sm_lds_optin((const void*)track_clip_kernel, 150 * 1024) != hipSuccess) return SM_ERR_LAUNCH;
  hipLaunchKernelGGL(track_clip_kernel, dim3(1), dim3(TC_THREADS), lds, sm_hip_stream(stream), det_feats, det, det_labels,
                     ndet, is_first, nframes, max_num, channels, coeff_score, coeff_iou, coeff_label, mem_feats, mem_boxes,
                     mem_labels, mem_count, capacity, comp_ws, ids);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
