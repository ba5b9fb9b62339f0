// Device-resident detection post-processing for SipMaskHead.get_bboxes
// (M/mmdet/models/anchor_heads/sipmask_head.py:543-605):
//   det_score  : max_c sigmoid(cls) * sigmoid(ctr) per position
//   det_topk   : exact per-level top-k (value desc, index asc): 4x8-bit radix select in LDS,
//                ordered tie compaction, bitonic sort of the <=2048 survivors
//   det_gather : sigmoid scores / centerness / distance2bbox / coefficient gather
//   nms_class  : one wave64 per (image, class): threshold -> sort -> greedy NMS where the
//                64-bit suppression word of the reference kernel (nms_kernel.cu:24-68) maps 1:1
//                onto wave64 ballots; no device->host copy, no host scan
//   nms_final  : class-major concat + top max_num (bbox_nms.py:131-140)
// Everything is integer/compare work on <= a few MB: latency bound, not bandwidth bound.
#include "common.h"

namespace {

constexpr int TK_THREADS = 1024;
constexpr int TK_CAP = 2048;

struct TopkSmem {
  unsigned long long sel[TK_CAP];
  unsigned int hist[256];
  unsigned int wsum[TK_THREADS / 64];
  unsigned int prefix, krem, pos_gt, base_eq;
};

__device__ __forceinline__ unsigned long long compose_key(uint32_t okey, uint32_t idx) {
  return ((unsigned long long)okey << 32) | (unsigned long long)(0xffffffffu - idx);
}

// bitonic sort, descending, P = power of two, all threads of the block participate.
// Pair t of a compare-exchange step touches pos = 2t - (t & (stride-1)) and pos + stride: for stride <= 64 the 64
// pairs of a wave lie inside the wave's own 128-element block [128w, 128w+128) (and [128(w+nw), ...) on the next
// trip of the t loop), so consecutive steps with stride <= 64 need no block barrier -- LDS / L1 accesses of ONE wave
// are ordered, a wave-level fence is enough.  Only steps with stride >= 128 exchange data between waves: P = 1024
// takes 9 block barriers instead of 55 (the sort was most of det_topk / nms_class for the heavy classes).
__device__ void block_bitonic_desc(unsigned long long* a, int P, bool in_lds = true) {
  bool cross = true;      // the data this step reads may have been written by another wave
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride >= 128 || cross) sm_syncthreads_flat();      // (`a` may be a generic pointer: common.h)
      else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // flat stores of the step before (generic `a`), same wave
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      cross = stride >= 128 || !in_lds;   // global scratch (heavy NMS classes): block barrier after every step
      for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
        const int pos = 2 * t - (t & (stride - 1));
        const unsigned long long x = a[pos], y = a[pos + stride];
        const bool desc = ((pos & size) == 0);
        if ((x < y) == desc) {
          a[pos] = y;
          a[pos + stride] = x;
        }
      }
    }
  }
  sm_syncthreads_flat();
}

// Exact top-k of keys[0..n) for one block of TK_THREADS threads.  Result: sm.sel[0..k) sorted
// by (key desc, index asc); low word = 0xffffffff - index.  Requires 1 <= k <= min(n, TK_CAP).
__device__ void block_topk(const float* __restrict__ gkeys, int n, int k, TopkSmem& sm, float* lds_keys = nullptr) {
  const int tid = threadIdx.x;
  // the keys are scanned 5 times (4 radix rounds + compaction): stage them in LDS when they fit
  const float* keys = gkeys;
  if (lds_keys != nullptr) {
    for (int i = tid; i < n; i += TK_THREADS) lds_keys[i] = gkeys[i];
    keys = lds_keys;
    __syncthreads();
  }
  if (tid == 0) {
    sm.prefix = 0;
    sm.krem = k;
    sm.pos_gt = 0;
    sm.base_eq = 0;
  }
  for (int round = 0; round < 4; ++round) {
    const int shift = 24 - 8 * round;
    if (tid < 256) sm.hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = sm.prefix;
    // (a wave-aggregated update -- one ballot + one atomic per distinct digit -- was measured SLOWER here, 0.131 vs
    // 0.111 ms for det_select: same-address LDS atomics of one wave are already combined by the hardware)
    for (int i = tid; i < n; i += TK_THREADS) {
      const uint32_t u = float_to_ordered(keys[i]);
      if (round == 0 || (u >> (shift + 8)) == prefix) atomicAdd(&sm.hist[(u >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned int cum = 0, krem = sm.krem;
      int bin = 0;
      for (int b = 255; b >= 0; --b) {
        const unsigned int h = sm.hist[b];
        if (cum + h >= krem) {
          bin = b;
          break;
        }
        cum += h;
      }
      sm.krem = krem - cum;
      sm.prefix = (prefix << 8) | (unsigned int)bin;
    }
    __syncthreads();
  }
  const uint32_t T = sm.prefix;
  const unsigned int krem = sm.krem;
  const unsigned int cnt_gt = (unsigned int)k - krem;
  const int lane = tid & 63, wv = tid >> 6;
  for (int base = 0; base < n; base += TK_THREADS) {
    const int i = base + tid;
    const bool valid = i < n;
    const uint32_t u = valid ? float_to_ordered(keys[i]) : 0u;
    const bool gt = valid && u > T;
    const bool eq = valid && u == T;
    if (gt) {
      const unsigned int slot = atomicAdd(&sm.pos_gt, 1u);
      sm.sel[slot] = compose_key(u, (uint32_t)i);
    }
    const unsigned long long bal = __ballot(eq);
    const unsigned int wrank = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) sm.wsum[wv] = __popcll(bal);
    __syncthreads();
    unsigned int before = sm.base_eq, total = 0;
#pragma unroll
    for (int w = 0; w < TK_THREADS / 64; ++w) {
      const unsigned int c = sm.wsum[w];
      if (w < wv) before += c;
      total += c;
    }
    if (eq) {
      const unsigned int rank = before + wrank;
      if (rank < krem) sm.sel[cnt_gt + rank] = compose_key(u, (uint32_t)i);
    }
    __syncthreads();
    if (tid == 0) sm.base_eq += total;
    __syncthreads();
  }
  int P = 1;
  while (P < k) P <<= 1;
  for (int i = k + tid; i < P; i += TK_THREADS) sm.sel[i] = 0ull;
  block_bitonic_desc(sm.sel, P);
}

struct DetArgs {
  int batch, nlev, C;
  int h[SM_MAX_LEVELS], w[SM_MAX_LEVELS], stride[SM_MAX_LEVELS], hw[SM_MAX_LEVELS];
  long long row0[SM_MAX_LEVELS];
  int pos0[SM_MAX_LEVELS + 1];   // start of level l inside the per-image key array
  int cand0[SM_MAX_LEVELS + 1];  // start of level l inside the per-image candidate list
  int cls_cs, cls_co, cof_cs, cof_co, reg_cs;
  int nms_pre, img_h, img_w, kmax;
  float scale_factor[4];
  int rescale, reg_prescaled;
  const float* per_image;  // [batch][6] = (img_h, img_w, scale_factor[4]) or nullptr (sm_det_desc.per_image)
  int topk_cache_floats;   // dynamic LDS floats available to det_topk_kernel (0 = scan global memory)
};

__global__ void det_score_kernel(const float* __restrict__ cls, const float* __restrict__ reg,
                                 float* __restrict__ keys, const DetArgs a) {
  const int S = a.pos0[a.nlev];
  const long long total = (long long)a.batch * S;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / S);
    const int p = (int)(i - (long long)b * S);
    int lev = 0;
#pragma unroll
    for (int l = 1; l < SM_MAX_LEVELS; ++l)
      if (l < a.nlev && p >= a.pos0[l]) lev = l;
    const long long row = a.row0[lev] + (long long)b * a.hw[lev] + (p - a.pos0[lev]);
    const float* cp = cls + row * a.cls_cs + a.cls_co;
    float mx = -INFINITY;
    for (int c = 0; c < a.C; ++c) mx = fmaxf(mx, cp[c]);
    // max_c(sigmoid(x_c) * s) == sigmoid(max_c x_c) * s : both maps are monotone
    keys[i] = __fmul_rn(sigmoid_rank(mx), sigmoid_rank(reg[row * a.reg_cs + 4]));
  }
}

__global__ __launch_bounds__(TK_THREADS) void det_topk_kernel(const float* __restrict__ keys,
                                                               int32_t* __restrict__ cand_pos, const DetArgs a) {
  __shared__ TopkSmem sm;
  extern __shared__ __attribute__((aligned(16))) float topk_cache[];
  const int lev = blockIdx.x, b = blockIdx.y;
  const int n = a.hw[lev];
  const int S = a.pos0[a.nlev];
  int32_t* out = cand_pos + (long long)b * a.kmax + a.cand0[lev];
  if (a.nms_pre <= 0 || n <= a.nms_pre) {  // sipmask_head.py:571: topk only when more than nms_pre
    for (int i = threadIdx.x; i < n; i += TK_THREADS) out[i] = i;
    return;
  }
  block_topk(keys + (long long)b * S + a.pos0[lev], n, a.nms_pre, sm, a.topk_cache_floats >= n ? topk_cache : nullptr);
  for (int i = threadIdx.x; i < a.nms_pre; i += TK_THREADS)
    out[i] = (int32_t)(0xffffffffu - (uint32_t)(sm.sel[i] & 0xffffffffull));
}

// 256 threads per 64 consecutive candidates of an image: thread = (candidate, class quarter).
// Scores are written CLASS-MAJOR ([B][C][kmax]) so that the per-class NMS blocks stream them
// with coalesced loads.
__global__ __launch_bounds__(256) void det_gather_kernel(const float* __restrict__ cls, const float* __restrict__ reg,
                                                         const float* __restrict__ cof,
                                                         const int32_t* __restrict__ cand_pos,
                                                         float* __restrict__ boxes, float* __restrict__ scores,
                                                         float* __restrict__ ctr, float* __restrict__ cofs,
                                                         const DetArgs a) {
  __shared__ long long s_row[64];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int lc = tid & 63, part = tid >> 6;
  const int k = blockIdx.x * 64 + lc;
  const bool valid = k < a.kmax;
  long long row = -1;
  if (valid) {
    int lev = 0;
#pragma unroll
    for (int l = 1; l < SM_MAX_LEVELS; ++l)
      if (l < a.nlev && k >= a.cand0[l]) lev = l;
    const int pos = cand_pos[(long long)b * a.kmax + k];
    row = a.row0[lev] + (long long)b * a.hw[lev] + pos;
    const long long o = (long long)b * a.kmax + k;
    if (part == 0) {
      const float* rp = reg + row * a.reg_cs;
      ctr[o] = sigmoid_rank(rp[4]);
      const int s = a.stride[lev];
      const int py = pos / a.w[lev], px = pos - py * a.w[lev];
      const float x = (float)(px * s) + (float)(s / 2), y = (float)(py * s) + (float)(s / 2);
      // img_shape / scale_factor of THIS image (img_metas[img_id], sipmask_head.py:517-541) when a table is given
      const float* pi = a.per_image != nullptr ? a.per_image + (long long)b * 6 : nullptr;
      const float xmax = (pi ? pi[1] : (float)a.img_w) - 1.f, ymax = (pi ? pi[0] : (float)a.img_h) - 1.f;
      const float fs = a.reg_prescaled ? 1.f : (float)s;  // bbox_pred.float() * stride (sipmask_head.py:268)
      float x1 = fminf(fmaxf(__fsub_rn(x, __fmul_rn(rp[0], fs)), 0.f), xmax);
      float y1 = fminf(fmaxf(__fsub_rn(y, __fmul_rn(rp[1], fs)), 0.f), ymax);
      float x2 = fminf(fmaxf(__fadd_rn(x, __fmul_rn(rp[2], fs)), 0.f), xmax);
      float y2 = fminf(fmaxf(__fadd_rn(y, __fmul_rn(rp[3], fs)), 0.f), ymax);
      if (a.rescale) {
        x1 /= pi ? pi[2] : a.scale_factor[0];
        y1 /= pi ? pi[3] : a.scale_factor[1];
        x2 /= pi ? pi[4] : a.scale_factor[2];
        y2 /= pi ? pi[5] : a.scale_factor[3];
      }
      *reinterpret_cast<float4*>(boxes + o * 4) = make_float4(x1, y1, x2, y2);
      s_row[lc] = row;
    }
    // class scores: lanes = candidates (coalesced class-major stores; the source rows stay in L1)
    const float* cp = cls + row * a.cls_cs + a.cls_co;
    float* sp = scores + (long long)b * a.C * a.kmax + k;
    for (int c = part; c < a.C; c += 4) sp[(long long)c * a.kmax] = sigmoid_rank(cp[c]);
  } else if (part == 0) {
    s_row[lc] = -1;
  }
  __syncthreads();
  // coefficients: 128 threads sweep the 128 channels of one candidate (coalesced both sides)
  const int ch = tid & 127;
  for (int jj = tid >> 7; jj < 64; jj += 2) {
    const long long r = s_row[jj];
    if (r < 0) continue;
    const long long o = (long long)b * a.kmax + blockIdx.x * 64 + jj;
    cofs[o * 128 + ch] = cof[r * a.cof_cs + a.cof_co + ch];
  }
}

// ------------------------------------------------------------------------------- (location, class) pairs
// Candidate selection of the maskrcnn-benchmark variant (B/ = SipMask-benchmark/,
// B/fcos_core/modeling/rpn/sipmask/inference.py:66-138): per level every (location, class) pair with
// sigmoid(cls) > pre_nms_thresh is a candidate, ranked by sigmoid(cls)*sigmoid(ctr); the best pre_nms_top_n of
// a level survive; the reported score is the square root.  keys [B][S*C] f32: the product, or -1 for a pair
// below the threshold.
__global__ void pair_score_kernel(const float* __restrict__ cls, const float* __restrict__ reg, float* __restrict__ keys,
                                  const DetArgs a, float thr) {
  const int S = a.pos0[a.nlev];
  const long long total = (long long)a.batch * S * a.C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % a.C);
    const long long bp = i / a.C;
    const int b = (int)(bp / S);
    const int p = (int)(bp - (long long)b * S);
    int lev = 0;
#pragma unroll
    for (int l = 1; l < SM_MAX_LEVELS; ++l)
      if (l < a.nlev && p >= a.pos0[l]) lev = l;
    const long long row = a.row0[lev] + (long long)b * a.hw[lev] + (p - a.pos0[lev]);
    const float pc = sigmoid_rank(cls[row * a.cls_cs + a.cls_co + c]);
    keys[i] = pc > thr ? __fmul_rn(pc, sigmoid_rank(reg[row * a.reg_cs + 4])) : -1.f;
  }
}

// one block per (level, image): k = min(#candidates, pre_nms_top_n) best pairs of the level (key desc, pair
// index asc), written to the level's slot range of the per-image candidate list; lvl_cnt[b][l] = k
__global__ __launch_bounds__(TK_THREADS) void pair_topk_kernel(const float* __restrict__ keys,
                                                                int32_t* __restrict__ cand_pair,
                                                                int32_t* __restrict__ lvl_cnt, const DetArgs a) {
  __shared__ TopkSmem sm;
  __shared__ int s_cnt;
  const int lev = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int S = a.pos0[a.nlev];
  const int n = a.hw[lev] * a.C;
  const float* kl = keys + ((long long)b * S + a.pos0[lev]) * a.C;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  int c = 0;
  for (int i = tid; i < n; i += TK_THREADS) c += kl[i] >= 0.f ? 1 : 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
  if ((tid & 63) == 0 && c) atomicAdd(&s_cnt, c);
  __syncthreads();
  const int k = min(s_cnt, a.nms_pre);
  if (tid == 0) lvl_cnt[b * SM_MAX_LEVELS + lev] = k;
  if (k <= 0) return;
  __syncthreads();
  block_topk(kl, n, k, sm);
  int32_t* out = cand_pair + (long long)b * a.kmax + a.cand0[lev];
  for (int i = tid; i < k; i += TK_THREADS) out[i] = (int32_t)(0xffffffffu - (uint32_t)(sm.sel[i] & 0xffffffffull));
}

// decode + gather of the selected pairs.  Slot k of an image belongs to level l iff cand0[l] <= k < cand0[l+1];
// slots past the level's count stay empty: zero box, no score (the dense class-major score matrix is zeroed
// by the host wrapper, and the NMS takes only scores > 0).
__global__ __launch_bounds__(128) void pair_gather_kernel(const float* __restrict__ cls, const float* __restrict__ reg,
                                                          const float* __restrict__ cof, const float* __restrict__ keys,
                                                          const int32_t* __restrict__ cand_pair,
                                                          const int32_t* __restrict__ lvl_cnt, float* __restrict__ boxes,
                                                          float* __restrict__ scores, float* __restrict__ cofs,
                                                          const DetArgs a) {
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && k >= a.cand0[l]) lev = l;
  const long long o = (long long)b * a.kmax + k;
  if (k - a.cand0[lev] >= lvl_cnt[b * SM_MAX_LEVELS + lev]) {
    if (tid < 4) boxes[o * 4 + tid] = 0.f;
    cofs[o * 128 + tid] = 0.f;
    return;
  }
  const int S = a.pos0[a.nlev];
  const int pair = cand_pair[o];
  const int pos = pair / a.C, c = pair - pos * a.C;
  const long long row = a.row0[lev] + (long long)b * a.hw[lev] + pos;
  if (tid == 0) {
    const float* rp = reg + row * a.reg_cs;
    const int s = a.stride[lev];
    const int py = pos / a.w[lev], px = pos - py * a.w[lev];
    const float x = (float)(px * s) + (float)(s / 2), y = (float)(py * s) + (float)(s / 2);
    const float* pi = a.per_image != nullptr ? a.per_image + (long long)b * 6 : nullptr;
    const float xmax = (pi ? pi[1] : (float)a.img_w) - 1.f, ymax = (pi ? pi[0] : (float)a.img_h) - 1.f;   // clip_to_image, TO_REMOVE = 1
    const float fs = a.reg_prescaled ? 1.f : (float)s;                       // bbox_pred * fpn_strides[l] (sipmask.py:161)
    const float x1 = fminf(fmaxf(__fsub_rn(x, __fmul_rn(rp[0], fs)), 0.f), xmax);
    const float y1 = fminf(fmaxf(__fsub_rn(y, __fmul_rn(rp[1], fs)), 0.f), ymax);
    const float x2 = fminf(fmaxf(__fadd_rn(x, __fmul_rn(rp[2], fs)), 0.f), xmax);
    const float y2 = fminf(fmaxf(__fadd_rn(y, __fmul_rn(rp[3], fs)), 0.f), ymax);
    *reinterpret_cast<float4*>(boxes + o * 4) = make_float4(x1, y1, x2, y2);
    const float key = keys[((long long)b * S + a.pos0[lev] + pos) * a.C + c];
    scores[((long long)b * a.C + c) * a.kmax + k] = __fsqrt_rn(key);          // torch.sqrt(per_box_cls), inference.py:133
  }
  cofs[o * 128 + tid] = cof[row * a.cof_cs + a.cof_co + tid];
}

// ------------------------------------------------------------------------------- NMS
// devIoU, nms_kernel.cu:14-22 (float32, +1 widths); written so the compiler cannot contract
__device__ __forceinline__ float iou_plus1(const float4 a, const float4 b) {
  const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  const float width = fmaxf(__fadd_rn(__fsub_rn(right, left), 1.f), 0.f);
  const float height = fmaxf(__fadd_rn(__fsub_rn(bottom, top), 1.f), 0.f);
  const float inter = __fmul_rn(width, height);
  const float sa = __fmul_rn(__fadd_rn(__fsub_rn(a.z, a.x), 1.f), __fadd_rn(__fsub_rn(a.w, a.y), 1.f));
  const float sb = __fmul_rn(__fadd_rn(__fsub_rn(b.z, b.x), 1.f), __fadd_rn(__fsub_rn(b.w, b.y), 1.f));
  return __fdiv_rn(inter, __fsub_rn(__fadd_rn(sa, sb), inter));
}

// IoU(+1)(a, b) > thr with the exact semantics of `iou_plus1(a, b) > thr` (f32 division, round to nearest) at a third
// of its cost: inter - thr * union decides directly unless it is within 1e-5 * union of zero (a quotient within
// 1e-5 of thr; the division's own rounding moves it by 6e-8), and only those borderline pairs -- and non-finite
// boxes -- pay for the division.
__device__ __forceinline__ bool iou_plus1_gt(const float4 a, const float4 b, float thr) {
  const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  const float width = fmaxf(__fadd_rn(__fsub_rn(right, left), 1.f), 0.f);
  const float height = fmaxf(__fadd_rn(__fsub_rn(bottom, top), 1.f), 0.f);
  const float inter = __fmul_rn(width, height);
  const float sa = __fmul_rn(__fadd_rn(__fsub_rn(a.z, a.x), 1.f), __fadd_rn(__fsub_rn(a.w, a.y), 1.f));
  const float sb = __fmul_rn(__fadd_rn(__fsub_rn(b.z, b.x), 1.f), __fadd_rn(__fsub_rn(b.w, b.y), 1.f));
  const float uni = __fsub_rn(__fadd_rn(sa, sb), inter);
  const float t = fmaf(-thr, uni, inter);
  // the sign of t decides only for a POSITIVE union: a box with x2 < x1 - 1 (or y2 < y1 - 1; bbox_pred is not
  // exp'd or ReLU'd, so negative distances are reachable) has a negative area, the union may be <= 0, the
  // reference's quotient (nms_kernel.cu:17-22) is then negative / inf / NaN and compares as such
  if (uni > 0.f && uni < 3e38f && fabsf(t) > 1e-5f * uni) return t > 0.f;      // NaN / inf / uni <= 0 fall through
  return __fdiv_rn(inter, uni) > thr;
}

constexpr int NMS_P_LDS = 2048;   // LDS working-set capacity per (image, class): 56 KB -> 2 blocks per CU
constexpr int NMS_THREADS = 1024;  // 16 waves per (image, class): heavy classes (thousands of boxes) set the tail

// LDS carve (dynamic): keys[P] (u64), kept_box[P] (float4), kept_idx[P] (u32); static: NmsSmem.
struct NmsSmem {
  unsigned long long rowm[64];                    // intra-chunk suppression rows (bits > t only)
  unsigned long long alive_w[NMS_THREADS / 64];   // per-wave "not suppressed by a kept box" masks
  unsigned int n;
  int nkept;
};

// Greedy NMS over the n sorted entries in keys (low word = 0xffffffff - candidate index), all
// NMS_THREADS threads of the block.  Semantics = the reference's bitmask kernel + host scan
// (nms_kernel.cu:24-68,113-138): box i is kept iff no EARLIER KEPT box has IoU(+1) > thr.
// Work split per 64-box chunk (score order): (a) each wave tests the chunk against a slice of
// the kept list, (b) each wave builds 16 rows of the 64x64 intra-chunk suppression matrix with
// ballots (the reference's 64-bit mask word == one wave64 ballot), (c) a scalar 64-step
// resolve on those rows, (d) kept boxes appended.  Returns the number kept (score order).
template <typename BoxFn>
__device__ int block_greedy_nms(const unsigned long long* keys, int n, float thr, float4* kept_box,
                                uint32_t* kept_idx, NmsSmem& sm, BoxFn box_of) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int NW = NMS_THREADS / 64;
  int nkept = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const bool valid = i < n;
    uint32_t idx = 0;
    float4 mine = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
      idx = 0xffffffffu - (uint32_t)(keys[i] & 0xffffffffull);
      mine = box_of(idx);
    }
    // (a) suppression by already-kept boxes, kept list striped over the waves
    bool alive = valid;
    for (int kk = wv; kk < nkept; kk += NW) {
      if (alive && iou_plus1_gt(kept_box[kk], mine, thr)) alive = false;
    }
    const unsigned long long aw = __ballot(alive);
    if (lane == 0) sm.alive_w[wv] = aw;
    // (b) rows t = wv, wv+NW, ... of the intra-chunk matrix: bit j set iff j > t and IoU(t, j) > thr
    for (int t = wv; t < 64; t += NW) {
      float4 bt;
      bt.x = __shfl(mine.x, t);
      bt.y = __shfl(mine.y, t);
      bt.z = __shfl(mine.z, t);
      bt.w = __shfl(mine.w, t);
      const bool hit = valid && lane > t && (base + t) < n && iou_plus1_gt(bt, mine, thr);
      const unsigned long long row = __ballot(hit);
      if (lane == 0) sm.rowm[t] = row;
    }
    sm_syncthreads_flat();
    // (c) scalar resolve + (d) append, by wave 0 alone (the 64 dependent steps are serial anyway; 16 waves doing them
    // redundantly only fought for the issue slots of the 4 SIMDs)
    if (wv == 0) {
      unsigned long long am = sm.alive_w[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) am &= sm.alive_w[w];
      const unsigned long long myrow = sm.rowm[lane];
      for (int t = 0; t < 64; ++t) {
        const unsigned long long rt = __shfl(myrow, t);
        if ((am >> t) & 1ull) am &= ~rt;
      }
      if ((am >> lane) & 1ull) {
        const int slot = nkept + __popcll(am & ((1ull << lane) - 1ull));
        kept_box[slot] = mine;
        kept_idx[slot] = idx;
      }
      if (lane == 0) sm.nkept = nkept + __popcll(am);
    }
    sm_syncthreads_flat();
    nkept = sm.nkept;
  }
  return nkept;
}

// ascending sort of u32 values through the u64 scratch (whole block)
__device__ void block_sort_idx_asc(unsigned long long* scratch, const uint32_t* vals, int n, bool in_lds = true) {
  int P = 1;
  while (P < n) P <<= 1;
  for (int i = threadIdx.x; i < P; i += blockDim.x)
    scratch[i] = i < n ? (unsigned long long)(0xffffffffu - vals[i]) : 0ull;
  block_bitonic_desc(scratch, P, in_lds);  // descending in (max - idx) == ascending idx; zeros (padding) last
}

struct NmsArgs {
  int batch, kmax, C, P, max_num;
  float score_thr, iou_thr;
  int always_sort;   // fast_nms: the final list is always sorted by score (sipmask_head.py:902)
  int top_k;         // fast_nms: boxes kept per class before the IoU test (:871)
  int P_lds;         // per-class capacity of the LDS working set (keys + kept boxes + indices, 28 bytes per entry)
  unsigned char* ext;   // [B][C][P] x 28 bytes in HBM/L2: working set of a class with more than P_lds candidates
  // heavy classes (more than heavy_min candidates above score_thr): the class block only sorts; the IoU tests run
  // chip-wide as a bit matrix (nms_heavy_matrix_kernel) and ONE wave scans it (nms_heavy_scan_kernel)
  int heavy_min, heavy_slots;   // heavy_slots == 0: path off
  int rows, W;                  // per slot: rows = kmax rounded up to 64, W = rows / 64 mask words per row
  int32_t* heavy_cnt;           // [1] slots requested by this launch (may exceed heavy_slots: the rest stay in-block)
  int32_t* heavy_meta;          // [slots][4] = (image, class, n, -)
  float4* hbox;                 // [slots][rows] boxes in score order
  uint32_t* hidx;               // [slots][rows] candidate indices in score order
  unsigned long long* hmat;     // [slots][rows][W] bit j of word w of row i: box w*64+j > i is suppressed by box i
};

// One block per (image, class).  A class typically passes score_thr with a few dozen candidates, rarely with
// thousands, so the LDS working set is sized for P_lds entries (2 blocks per CU: all classes of a batch are resident
// at once) and the rare heavy class works out of its global scratch slice instead (same code, slower memory).
__global__ __launch_bounds__(NMS_THREADS) void nms_class_kernel(const float* __restrict__ boxes,
                                                                const float* __restrict__ scores,
                                                                const float* __restrict__ ctr,
                                                                const int32_t* __restrict__ ncand,
                                                                int32_t* __restrict__ cls_keep,
                                                                int32_t* __restrict__ cls_cnt, const NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  __shared__ NmsSmem sm;
  const int c = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int K = ncand[b];
  const float* sc = scores + ((long long)b * a.C + c) * a.kmax;   // class-major: contiguous over candidates
  const float* ct = ctr + (long long)b * a.kmax;
  const float4* bx = reinterpret_cast<const float4*>(boxes) + (long long)b * a.kmax;
  if (tid == 0) sm.n = 0;
  sm_syncthreads_flat();
  // 0. how many candidates pass the raw class score threshold (bbox_nms.py:111)?  decides LDS vs global scratch
  {
    unsigned int cnt = 0;
    for (int i = tid; i < K; i += NMS_THREADS) cnt += (sc[i] > a.score_thr) ? 1u : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    if (lane == 0 && cnt) atomicAdd(&sm.n, cnt);
  }
  sm_syncthreads_flat();
  const int n = (int)sm.n;
  sm_syncthreads_flat();
  int32_t* outk = cls_keep + ((long long)b * a.C + c) * a.kmax;
  if (n == 0) {
    if (tid == 0) cls_cnt[b * a.C + c] = 0;
    return;
  }
  int P = 1;
  while (P < n) P <<= 1;
  // a heavy class asks for a slot of the chip-wide path; with a slot only the keys are needed here (8 bytes per entry)
  int slot = -1;
  if (a.heavy_slots > 0 && n > a.heavy_min) {
    if (tid == 0) sm.nkept = atomicAdd(a.heavy_cnt, 1);
    sm_syncthreads_flat();
    slot = sm.nkept < a.heavy_slots ? sm.nkept : -1;
    sm_syncthreads_flat();
  }
  if (tid == 0) sm.n = 0;
  const bool in_lds = slot >= 0 ? (size_t)P * 8 <= (size_t)a.P_lds * 28 : P <= a.P_lds;
  unsigned char* ws = in_lds ? dsm : a.ext + ((size_t)b * a.C + c) * (size_t)a.P * 28;
  const size_t cap = in_lds ? (size_t)a.P_lds : (size_t)a.P;
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws);
  float4* kept_box = reinterpret_cast<float4*>(ws + cap * 8);
  uint32_t* kept_idx = reinterpret_cast<uint32_t*>(ws + cap * 24);
  sm_syncthreads_flat();
  // 1. compaction of candidates with raw class score > thr.  The order of insertion is irrelevant: the sort key
  //    (score, index) is a total order.
  for (int base = 0; base < K; base += NMS_THREADS) {
    const int i = base + tid;
    const float s = (i < K) ? sc[i] : 0.f;
    const bool sel = (i < K) && (s > a.score_thr);
    const unsigned long long bal = __ballot(sel);
    unsigned int wbase = 0;
    if (lane == 0 && bal) wbase = atomicAdd(&sm.n, (unsigned int)__popcll(bal));
    wbase = __shfl(wbase, 0);
    if (sel) {
      const float sf = __fmul_rn(s, ct[i]);  // _scores *= score_factors (bbox_nms.py:121-122)
      keys[wbase + __popcll(bal & ((1ull << lane) - 1ull))] = compose_key(float_to_ordered(sf), (uint32_t)i);
    }
  }
  for (int i = n + tid; i < P; i += NMS_THREADS) keys[i] = 0ull;
  sm_syncthreads_flat();
  // 2. sort (score desc, index asc)
  block_bitonic_desc(keys, P, in_lds);
  if (slot >= 0) {   // hand the sorted class over: boxes + indices in score order
    float4* hb = a.hbox + (size_t)slot * a.rows;
    uint32_t* hi = a.hidx + (size_t)slot * a.rows;
    for (int i = tid; i < n; i += NMS_THREADS) {
      const uint32_t idx = 0xffffffffu - (uint32_t)(keys[i] & 0xffffffffull);
      hi[i] = idx;
      hb[i] = bx[idx];
    }
    if (tid == 0) {
      a.heavy_meta[slot * 4 + 0] = b;
      a.heavy_meta[slot * 4 + 1] = c;
      a.heavy_meta[slot * 4 + 2] = n;
    }
    return;
  }
  // 3. greedy NMS
  const int nk = block_greedy_nms(keys, n, a.iou_thr, kept_box, kept_idx, sm, [&](uint32_t idx) { return bx[idx]; });
  // 4. reference returns kept ORIGINAL indices ascending (nms_kernel.cu:135-138)
  block_sort_idx_asc(keys, kept_idx, nk, in_lds);
  for (int i = tid; i < nk; i += NMS_THREADS) outk[i] = (int32_t)(0xffffffffu - (uint32_t)keys[i]);
  if (tid == 0) cls_cnt[b * a.C + c] = nk;
}

// ---- heavy classes: bit matrix over the whole chip + one-wave scan.  In the class block the IoU tests of a class with n
// candidates and k kept boxes (n * k / 2 of them, 22 VALU instructions each) run on ONE CU: 1 000 candidates cost 60 us,
// 3 350 (every candidate of an image in one class) over a millisecond.  Here tile (r, w) of the reference's own n x n/64
// mask (nms_kernel.cu:24-68: bit j of word w of row i = IoU(i, w*64+j) > thr, j > i) is one wave: 64 ballots.
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

__global__ __launch_bounds__(256) void nms_heavy_matrix_kernel(const NmsArgs a) {
  const int slot = blockIdx.y;
  const int nh = min(*a.heavy_cnt, a.heavy_slots);
  if (slot >= nh) return;
  const int n = a.heavy_meta[slot * 4 + 2];
  const int Wn = (n + 63) >> 6;
  const float4* hb = a.hbox + (size_t)slot * a.rows;
  unsigned long long* M = a.hmat + (size_t)slot * a.rows * a.W;
  const int lane = threadIdx.x & 63;
  const int nw = gridDim.x * 4;
  for (int q = blockIdx.x * 4 + (threadIdx.x >> 6); q < Wn * Wn; q += nw) {
    const int r = q / Wn, w = q - r * Wn;
    if (w < r) continue;
    const int col = w * 64 + lane, rowi = r * 64 + lane;
    const bool cv = col < n;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 cb = cv ? hb[col] : z;
    const float4 rb = rowi < n ? hb[rowi] : z;
    unsigned long long mine = 0ull;
#pragma unroll 16
    for (int t = 0; t < 64; ++t) {
      float4 bt;
      bt.x = readlane_f(rb.x, t);
      bt.y = readlane_f(rb.y, t);
      bt.z = readlane_f(rb.z, t);
      bt.w = readlane_f(rb.w, t);
      const int ri = r * 64 + t;
      const bool hit = cv && ri < n && col > ri && iou_plus1_gt(bt, cb, a.iou_thr);
      const unsigned long long bal = __ballot(hit);
      if (lane == t) mine = bal;
    }
    if (rowi < n) M[(size_t)rowi * a.W + w] = mine;
  }
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, l);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// The host scan of the reference (nms_kernel.cu:113-138: `if (!(remv[nblock] & (1 << inblock))) { keep; remv |= row }`)
// by one wave: lane w holds word w (+64, +128 ...) of `remv`, the word of the running chunk is mirrored in a scalar so
// the dependent chain of a step is three scalar instructions; the rows are prefetched G at a time (they do not depend on
// any decision).  The other 3 waves of the block only take part in the final ascending index sort.
constexpr int SCAN_THREADS = 256;   // one wave per SIMD: the scanning wave may use the whole VGPR file for its row buffers
template <int WPL, int G>
__global__ __launch_bounds__(SCAN_THREADS) void nms_heavy_scan_kernel(int32_t* __restrict__ cls_keep,
                                                                     int32_t* __restrict__ cls_cnt, const NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  __shared__ int s_nk;
  const int slot = blockIdx.x;
  const int nh = min(*a.heavy_cnt, a.heavy_slots);
  if (slot >= nh) return;
  const int b = a.heavy_meta[slot * 4 + 0], c = a.heavy_meta[slot * 4 + 1], n = a.heavy_meta[slot * 4 + 2];
  int P = 1;
  while (P < n) P <<= 1;
  unsigned long long* sortbuf = reinterpret_cast<unsigned long long*>(dsm);
  uint32_t* kept_idx = reinterpret_cast<uint32_t*>(dsm + (size_t)a.P * 8);
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < 64) {
    const int Wn = (n + 63) >> 6;
    const unsigned long long* M = a.hmat + (size_t)slot * a.rows * a.W;
    const uint32_t* hi = a.hidx + (size_t)slot * a.rows;
    unsigned long long removed[WPL];
#pragma unroll
    for (int k = 0; k < WPL; ++k) removed[k] = 0ull;
    unsigned long long buf[2][G][WPL];
    auto load = [&](unsigned long long (&dst)[G][WPL], int row0) {
      // branch-free: rows past n are clamped (process() ignores them), words below the diagonal or past the class's
      // last word were never written and are masked out
      const int cr = row0 >> 6;
#pragma unroll
      for (int k = 0; k < WPL; ++k) {
        const int wd = k * 64 + lane;
        const bool live = wd >= cr && wd < Wn;
        const unsigned long long* mp = M + min(wd, a.W - 1);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const unsigned long long v = mp[(size_t)min(row0 + g, n - 1) * a.W];
          dst[g][k] = live ? v : 0ull;
        }
      }
    };
    int nk = 0;
    unsigned long long cw = 0ull, km = 0ull;
    auto process = [&](unsigned long long (&cur)[G][WPL], int row0) {
      const int cc = row0 >> 6, cl = cc & 63, ck = cc >> 6;
      if ((row0 & 63) == 0) {
        unsigned long long v = removed[0];
#pragma unroll
        for (int k = 1; k < WPL; ++k)
          if (ck == k) v = removed[k];
        cw = readlane_u64(v, cl);
        if (cc * 64 + 64 > n) cw |= ~0ull << (n - cc * 64);   // rows past n: "removed", so a step is one bit test
        km = 0ull;
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int t = (row0 & 63) + g;
        unsigned long long dv = cur[g][0];
#pragma unroll
        for (int k = 1; k < WPL; ++k)
          if (ck == k) dv = cur[g][k];
        const unsigned long long dg = readlane_u64(dv, cl);
        if (!((cw >> t) & 1ull)) {
          cw |= dg;
          km |= 1ull << t;
#pragma unroll
          for (int k = 0; k < WPL; ++k) removed[k] |= cur[g][k];
        }
      }
      if (((row0 + G) & 63) == 0 || row0 + G >= n) {   // chunk complete: append its kept boxes in score order
        if ((km >> lane) & 1ull) kept_idx[nk + __popcll(km & ((1ull << lane) - 1ull))] = hi[cc * 64 + lane];
        nk += __popcll(km);
      }
    };
    const int ng = (n + G - 1) / G;
    load(buf[0], 0);
    for (int gi = 0; gi < ng; gi += 2) {
      if (gi + 1 < ng) load(buf[1], (gi + 1) * G);
      process(buf[0], gi * G);
      if (gi + 1 < ng) {
        if (gi + 2 < ng) load(buf[0], (gi + 2) * G);
        process(buf[1], (gi + 1) * G);
      }
    }
    if (lane == 0) s_nk = nk;
  }
  __syncthreads();
  const int nk = s_nk;
  // reference returns kept ORIGINAL indices ascending (nms_kernel.cu:135-138)
  block_sort_idx_asc(sortbuf, kept_idx, nk);
  int32_t* outk = cls_keep + ((long long)b * a.C + c) * a.kmax;
  for (int i = tid; i < nk; i += SCAN_THREADS) outk[i] = (int32_t)(0xffffffffu - (uint32_t)sortbuf[i]);
  if (tid == 0) cls_cnt[b * a.C + c] = nk;
}


__global__ __launch_bounds__(TK_THREADS) void nms_final_kernel(const float* __restrict__ boxes,
                                                               const float* __restrict__ scores,
                                                               const float* __restrict__ ctr,
                                                               const int32_t* __restrict__ cls_keep,
                                                               const int32_t* __restrict__ cls_cnt,
                                                               float* __restrict__ flat_key, float* __restrict__ det,
                                                               int64_t* __restrict__ labels,
                                                               int64_t* __restrict__ keep, int32_t* __restrict__ ndet,
                                                               const NmsArgs a) {
  __shared__ TopkSmem sm;
  __shared__ int pre[257];
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ int cnt_s[256];
  if (tid < a.C) cnt_s[tid] = cls_cnt[b * a.C + tid];
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int c = 0; c < a.C; ++c) {
      pre[c] = t;
      t += cnt_s[c];
    }
    pre[a.C] = t;
  }
  __syncthreads();
  const int total = pre[a.C];
  float* fk = flat_key + (long long)b * a.C * a.kmax;
  auto emit = [&](int slot, int p) {
    // p = position in the class-major concatenation
    int lo = 0, hi = a.C;  // find class c with pre[c] <= p < pre[c+1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pre[mid] <= p) lo = mid; else hi = mid;
    }
    const int c = lo;
    const int idx = cls_keep[((long long)b * a.C + c) * a.kmax + (p - pre[c])];
    const long long o = (long long)b * a.max_num + slot;
    const float4 bb = reinterpret_cast<const float4*>(boxes)[(long long)b * a.kmax + idx];
    const float s = __fmul_rn(scores[((long long)b * a.C + c) * a.kmax + idx], ctr[(long long)b * a.kmax + idx]);
    det[o * 5 + 0] = bb.x;
    det[o * 5 + 1] = bb.y;
    det[o * 5 + 2] = bb.z;
    det[o * 5 + 3] = bb.w;
    det[o * 5 + 4] = s;
    labels[o] = c;
    keep[o] = idx;
  };
  if (total <= a.max_num && !a.always_sort) {
    for (int p = tid; p < total; p += TK_THREADS) emit(p, p);
    if (tid == 0) ndet[b] = total;
    return;
  }
  if (total == 0) {
    if (tid == 0) ndet[b] = 0;
    return;
  }
  const int nout = min(total, a.max_num);
  // more than max_num: sort by score desc (ties: position asc) and keep the top max_num
  for (int p = tid; p < total; p += TK_THREADS) {
    int lo = 0, hi = a.C;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pre[mid] <= p) lo = mid; else hi = mid;
    }
    const int idx = cls_keep[((long long)b * a.C + lo) * a.kmax + (p - pre[lo])];
    fk[p] = __fmul_rn(scores[((long long)b * a.C + lo) * a.kmax + idx], ctr[(long long)b * a.kmax + idx]);
  }
  __syncthreads();
  if (total <= TK_CAP) {   // the usual case (a few hundred survivors): sort them all, no radix select (46 -> 15 us)
    int P = 1;
    while (P < total) P <<= 1;
    for (int p = tid; p < P; p += TK_THREADS)
      sm.sel[p] = p < total ? compose_key(float_to_ordered(fk[p]), (uint32_t)p) : 0ull;
    block_bitonic_desc(sm.sel, P);
  } else {
    block_topk(fk, total, nout, sm);
  }
  for (int i = tid; i < nout; i += TK_THREADS)
    emit(i, (int)(0xffffffffu - (uint32_t)(sm.sel[i] & 0xffffffffull)));
  if (tid == 0) ndet[b] = nout;
}

// ------------------------------------------------------------------------------- fast_nms
// jaccard WITHOUT the +1 (sipmask_head.py:912-960), float32, same operation order
__device__ __forceinline__ float iou_plain(const float4 a, const float4 b) {
  const float w = fmaxf(__fsub_rn(fminf(a.z, b.z), fmaxf(a.x, b.x)), 0.f);
  const float h = fmaxf(__fsub_rn(fminf(a.w, b.w), fmaxf(a.y, b.y)), 0.f);
  const float inter = __fmul_rn(w, h);
  const float aa = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
  const float ab = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
  return __fdiv_rn(inter, __fsub_rn(__fadd_rn(aa, ab), inter));
}

// SipMaskHead.fast_nms (sipmask_head.py:868-910), one block per (image, class): top_k boxes by
// score*centerness (desc, index asc), keep j iff max_{i<j} IoU(i,j) <= thr (a NaN IoU -- degenerate
// boxes -- propagates through torch.max and fails the test) and score > score_thr.  Kept
// candidates are written in rank order; nms_final_kernel (always_sort) does the global top-N.
__global__ __launch_bounds__(TK_THREADS) void fast_nms_class_kernel(const float* __restrict__ boxes,
                                                                    const float* __restrict__ scores,
                                                                    const float* __restrict__ ctr,
                                                                    const int32_t* __restrict__ ncand,
                                                                    float* __restrict__ keybuf,
                                                                    int32_t* __restrict__ cls_keep,
                                                                    int32_t* __restrict__ cls_cnt, const NmsArgs a) {
  __shared__ TopkSmem sm;
  __shared__ float4 s_box[TK_CAP];
  __shared__ unsigned int s_keep[TK_CAP];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int K = ncand[b];
  const float* sc = scores + ((long long)b * a.C + c) * a.kmax;
  const float* ct = ctr + (long long)b * a.kmax;
  float* kb = keybuf + ((long long)b * a.C + c) * a.kmax;
  for (int i = tid; i < K; i += TK_THREADS) kb[i] = __fmul_rn(sc[i], ct[i]);   // mlvl_scores * centerness (:604)
  __syncthreads();
  const int k = min(a.top_k, K);
  int32_t* outk = cls_keep + ((long long)b * a.C + c) * a.kmax;
  if (k <= 0) {
    if (tid == 0) cls_cnt[b * a.C + c] = 0;
    return;
  }
  block_topk(kb, K, k, sm);
  const float4* bx = reinterpret_cast<const float4*>(boxes) + (long long)b * a.kmax;
  for (int i = tid; i < k; i += TK_THREADS) s_box[i] = bx[0xffffffffu - (uint32_t)(sm.sel[i] & 0xffffffffull)];
  __syncthreads();
  for (int j = tid; j < k; j += TK_THREADS) {
    const float4 mine = s_box[j];
    float mx = 0.f;   // triu_(diagonal=1) leaves zeros below the diagonal, so the column max starts at 0
    bool nan = false;
    for (int i = 0; i < j; ++i) {
      const float v = iou_plain(s_box[i], mine);
      nan |= (v != v);
      mx = fmaxf(mx, v);
    }
    const uint32_t idx = 0xffffffffu - (uint32_t)(sm.sel[j] & 0xffffffffull);
    const bool kp = !nan && (mx <= a.iou_thr) && (kb[idx] > a.score_thr);
    s_keep[j] = kp ? 1u : 0u;
  }
  __syncthreads();
  if (tid == 0) {   // k <= 2048: a serial compaction in rank order is cheap
    int n = 0;
    for (int j = 0; j < k; ++j)
      if (s_keep[j]) outk[n++] = (int32_t)(0xffffffffu - (uint32_t)(sm.sel[j] & 0xffffffffull));
    cls_cnt[b * a.C + c] = n;
  }
}

// reference op contract: dets [n][5] -> ascending kept indices (nms_cuda.nms)
__global__ __launch_bounds__(NMS_THREADS) void nms_single_kernel(const float* __restrict__ dets, int n, float thr,
                                                                 int P, int64_t* __restrict__ keep,
                                                                 int32_t* __restrict__ nkeep) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  __shared__ NmsSmem sm;
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(dsm);
  float4* kept_box = reinterpret_cast<float4*>(dsm + (size_t)P * 8);
  uint32_t* kept_idx = reinterpret_cast<uint32_t*>(dsm + (size_t)P * 24);
  const int tid = threadIdx.x;
  for (int i = tid; i < P; i += NMS_THREADS)
    keys[i] = i < n ? compose_key(float_to_ordered(dets[(long long)i * 5 + 4]), (uint32_t)i) : 0ull;
  block_bitonic_desc(keys, P);
  const int nk = block_greedy_nms(keys, n, thr, kept_box, kept_idx, sm, [&](uint32_t idx) {
    const float* p = dets + (long long)idx * 5;
    return make_float4(p[0], p[1], p[2], p[3]);
  });
  block_sort_idx_asc(keys, kept_idx, nk);
  for (int i = tid; i < nk; i += NMS_THREADS) keep[i] = (int64_t)(0xffffffffu - (uint32_t)keys[i]);
  if (tid == 0) *nkeep = nk;
}

__global__ void fill_i32_kernel(int32_t* p, int n, int v) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = v;
}

int fill_det_args(const sm_det_desc* d, DetArgs& a) {
  if (!d || d->nlev < 1 || d->nlev > SM_MAX_LEVELS || d->batch < 1 || d->num_classes < 1) return SM_ERR_BAD_SHAPE;
  if (d->nms_pre > TK_CAP) return SM_ERR_UNSUPPORTED;
  a.batch = d->batch;
  a.nlev = d->nlev;
  a.C = d->num_classes;
  int p = 0, k = 0;
  for (int l = 0; l < SM_MAX_LEVELS; ++l) {
    const bool on = l < d->nlev;
    a.h[l] = on ? d->h[l] : 1;
    a.w[l] = on ? d->w[l] : 1;
    a.stride[l] = on ? d->stride[l] : 1;
    a.hw[l] = on ? d->h[l] * d->w[l] : 0;
    a.row0[l] = on ? d->row0[l] : 0;
    a.pos0[l] = p;
    a.cand0[l] = k;
    if (on) {
      p += a.hw[l];
      k += (d->nms_pre > 0 && a.hw[l] > d->nms_pre) ? d->nms_pre : a.hw[l];
    }
  }
  a.pos0[SM_MAX_LEVELS] = p;
  a.cand0[SM_MAX_LEVELS] = k;
  for (int l = d->nlev; l <= SM_MAX_LEVELS; ++l) {
    a.pos0[l] = p;
    a.cand0[l] = k;
  }
  if (k != d->kmax) return SM_ERR_BAD_SHAPE;
  if (d->reg_cstride < 5 || d->reg_cstride % 4 != 0) return SM_ERR_BAD_SHAPE;
  a.cls_cs = d->cls_cstride;
  a.cls_co = d->cls_coff;
  a.cof_cs = d->cof_cstride;
  a.cof_co = d->cof_coff;
  a.reg_cs = d->reg_cstride;
  a.nms_pre = d->nms_pre;
  a.img_h = d->img_h;
  a.img_w = d->img_w;
  a.kmax = d->kmax;
  for (int i = 0; i < 4; ++i) a.scale_factor[i] = d->scale_factor[i];
  a.rescale = d->rescale;
  a.reg_prescaled = d->reg_prescaled;
  a.per_image = d->per_image;
  a.topk_cache_floats = 0;
  return SM_OK;
}

int next_pow2(int v) {
  int p = 64;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

extern "C" int64_t sm_det_select_workspace(const sm_det_desc* d) {
  DetArgs a;
  if (fill_det_args(d, a) != SM_OK) return -1;
  return (int64_t)d->batch * a.pos0[d->nlev] * sizeof(float);
}

extern "C" int sm_det_select(const sm_det_desc* d, const float* cls, const float* reg, const float* cof, float* boxes,
                             float* scores, float* ctr, float* cofs, int32_t* cand_pos, int32_t* ncand,
                             void* workspace, sm_stream_t stream) {
  if (!cls || !reg || !cof || !boxes || !scores || !ctr || !cofs || !cand_pos || !ncand || !workspace)
    return SM_ERR_BAD_ARG;
  DetArgs a;
  int st = fill_det_args(d, a);
  if (st != SM_OK) return st;
  hipStream_t s = sm_hip_stream(stream);
  float* keys = (float*)workspace;
  const long long total = (long long)a.batch * a.pos0[a.nlev];
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(det_score_kernel, dim3(g), dim3(256), 0, s, cls, reg, keys, a);
  int maxn = 0;
  for (int l = 0; l < a.nlev; ++l) maxn = a.hw[l] > maxn ? a.hw[l] : maxn;
  size_t cache_bytes = (size_t)maxn * sizeof(float);
  if (cache_bytes > 128 * 1024) cache_bytes = 0;   // TopkSmem (17 KiB) + cache must fit 160 KiB
  a.topk_cache_floats = (int)(cache_bytes / sizeof(float));
  if (cache_bytes > 48 * 1024 && sm_lds_optin((const void*)det_topk_kernel, (int)cache_bytes) != hipSuccess)
    return SM_ERR_LAUNCH;
  hipLaunchKernelGGL(det_topk_kernel, dim3(a.nlev, a.batch), dim3(TK_THREADS), cache_bytes, s, keys, cand_pos, a);
  hipLaunchKernelGGL(det_gather_kernel, dim3((a.kmax + 63) / 64, a.batch), dim3(256), 0, s, cls, reg, cof, cand_pos, boxes, scores,
                     ctr, cofs, a);
  // every image has exactly kmax candidates (sum_l min(nms_pre, hw_l))
  if (a.batch > 1024) return SM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(1024), 0, s, ncand, a.batch, a.kmax);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

// per-level candidate capacity of the pair selection: min(pre_nms_top_n, hw*C)
static int fill_pair_args(const sm_det_desc* d, DetArgs& a) {
  if (!d || d->nlev < 1 || d->nlev > SM_MAX_LEVELS || d->batch < 1 || d->num_classes < 1) return SM_ERR_BAD_SHAPE;
  if (d->nms_pre < 1 || d->nms_pre > TK_CAP) return SM_ERR_UNSUPPORTED;
  sm_det_desc t = *d;
  int k = 0;
  for (int l = 0; l < d->nlev; ++l) {
    const long long npair = (long long)d->h[l] * d->w[l] * d->num_classes;
    k += (int)(npair < d->nms_pre ? npair : d->nms_pre);
  }
  if (k != d->kmax) return SM_ERR_BAD_SHAPE;
  // fill_det_args derives kmax from min(nms_pre, hw): give it what it expects, then restore the pair layout
  int kk = 0;
  for (int l = 0; l < d->nlev; ++l) kk += (d->h[l] * d->w[l] > d->nms_pre) ? d->nms_pre : d->h[l] * d->w[l];
  t.kmax = kk;
  const int st = fill_det_args(&t, a);
  if (st != SM_OK) return st;
  int c0 = 0;
  for (int l = 0; l <= SM_MAX_LEVELS; ++l) {
    a.cand0[l] = c0;
    if (l < d->nlev) {
      const long long npair = (long long)d->h[l] * d->w[l] * d->num_classes;
      c0 += (int)(npair < d->nms_pre ? npair : d->nms_pre);
    }
  }
  a.kmax = d->kmax;
  return SM_OK;
}

extern "C" int64_t sm_pairs_select_workspace(const sm_det_desc* d) {
  DetArgs a;
  if (fill_pair_args(d, a) != SM_OK) return -1;
  return (int64_t)d->batch * a.pos0[d->nlev] * d->num_classes * sizeof(float) + (int64_t)d->batch * d->kmax * 4 + 256;
}

extern "C" int sm_pairs_select(const sm_det_desc* d, float pre_nms_thresh, const float* cls, const float* reg,
                               const float* cof, float* boxes, float* scores, float* cofs, int32_t* lvl_cnt,
                               int32_t* ncand, void* workspace, sm_stream_t stream) {
  if (!cls || !reg || !cof || !boxes || !scores || !cofs || !lvl_cnt || !ncand || !workspace) return SM_ERR_BAD_ARG;
  DetArgs a;
  const int st = fill_pair_args(d, a);
  if (st != SM_OK) return st;
  if (a.batch > 1024) return SM_ERR_UNSUPPORTED;
  hipStream_t s = sm_hip_stream(stream);
  float* keys = (float*)workspace;
  const size_t key_bytes = (size_t)a.batch * a.pos0[a.nlev] * a.C * sizeof(float);
  int32_t* cand_pair = (int32_t*)((char*)workspace + (key_bytes + 255) / 256 * 256);
  const long long total = (long long)a.batch * a.pos0[a.nlev] * a.C;
  int g = (int)((total + 255) / 256);
  if (g > 16384) g = 16384;
  if (sm_zero_async(scores, (size_t)a.batch * a.C * a.kmax * sizeof(float), s) != hipSuccess) return SM_ERR_LAUNCH;
  hipLaunchKernelGGL(pair_score_kernel, dim3(g), dim3(256), 0, s, cls, reg, keys, a, pre_nms_thresh);
  hipLaunchKernelGGL(pair_topk_kernel, dim3(a.nlev, a.batch), dim3(TK_THREADS), 0, s, keys, cand_pair, lvl_cnt, a);
  hipLaunchKernelGGL(pair_gather_kernel, dim3(a.kmax, a.batch), dim3(128), 0, s, cls, reg, cof, keys, cand_pair, lvl_cnt,
                     boxes, scores, cofs, a);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(1024), 0, s, ncand, a.batch, a.kmax);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

namespace {
// heavy-class path: classes with more than NMS_HEAVY_MIN candidates above the score threshold; slots beyond
// NMS_HEAVY_SLOTS in one launch stay on the in-block path
constexpr int NMS_HEAVY_SLOTS = 32;
constexpr int NMS_HEAVY_MIN = 512;
int nms_heavy_min() { return NMS_HEAVY_MIN; }
struct HeavyLayout {
  int slots, rows, W;
  size_t off_cnt, off_meta, off_box, off_idx, off_mat, end;
};
HeavyLayout heavy_layout(int batch, int kmax, int num_classes, size_t base) {
  HeavyLayout h;
  const long long cls = (long long)batch * num_classes;
  h.slots = cls < NMS_HEAVY_SLOTS ? (int)cls : NMS_HEAVY_SLOTS;
  h.rows = (kmax + 63) / 64 * 64;
  h.W = h.rows / 64;
  if (nms_heavy_min() <= 0 || kmax <= nms_heavy_min() || h.W > 128) h.slots = 0;   // scan kernel: <= 2 words per lane
  h.off_cnt = (base + 255) / 256 * 256;
  h.off_meta = h.off_cnt + 256;
  h.off_box = h.off_meta + (size_t)h.slots * 16;
  h.off_idx = h.off_box + (size_t)h.slots * h.rows * 16;
  h.off_mat = (h.off_idx + (size_t)h.slots * h.rows * 4 + 255) / 256 * 256;
  h.end = h.off_mat + (size_t)h.slots * h.rows * h.W * 8;
  return h;
}
size_t nms_base_bytes(int batch, int kmax, int num_classes) {
  // cls_keep i32 [B][C][kmax] + flat_key f32 [B][C][kmax] + cls_cnt i32 [B][C]
  size_t n = (size_t)batch * num_classes * kmax * 8 + (size_t)batch * num_classes * 4;
  const size_t P = next_pow2(kmax);
  if (P > NMS_P_LDS) n = (n + 255) / 256 * 256 + (size_t)batch * num_classes * P * 28;   // in-block scratch of big classes
  return n;
}

template <int WPL, int G>
int launch_heavy_scan(int slots, size_t lds, hipStream_t s, int32_t* cls_keep, int32_t* cls_cnt, const NmsArgs& a) {
  if (lds > 48 * 1024 && sm_lds_optin((const void*)nms_heavy_scan_kernel<WPL, G>, 144 * 1024) != hipSuccess)
    return SM_ERR_LAUNCH;
  hipLaunchKernelGGL((nms_heavy_scan_kernel<WPL, G>), dim3(slots), dim3(SCAN_THREADS), lds, s, cls_keep, cls_cnt, a);
  return SM_OK;
}

}  // namespace

extern "C" int64_t sm_multiclass_nms_workspace(int batch, int kmax, int num_classes) {
  return (int64_t)heavy_layout(batch, kmax, num_classes, nms_base_bytes(batch, kmax, num_classes)).end;
}

extern "C" int sm_multiclass_nms(const float* boxes, const float* scores, const float* ctr, const int32_t* ncand,
                                 int batch, int kmax, int num_classes, float score_thr, float iou_thr, int max_num,
                                 float* det, int64_t* labels, int64_t* keep, int32_t* ndet, void* workspace,
                                 sm_stream_t stream) {
  if (!boxes || !scores || !ctr || !ncand || !det || !labels || !keep || !ndet || !workspace) return SM_ERR_BAD_ARG;
  if (batch < 1 || kmax < 1 || num_classes < 1 || num_classes > 256) return SM_ERR_BAD_SHAPE;
  if (max_num < 1 || max_num > TK_CAP) return SM_ERR_UNSUPPORTED;
  NmsArgs a = {};
  a.batch = batch;
  a.kmax = kmax;
  a.C = num_classes;
  a.P = next_pow2(kmax);
  a.max_num = max_num;
  a.score_thr = score_thr;
  a.iou_thr = iou_thr;
  a.always_sort = 0;
  a.top_k = 0;
  a.P_lds = a.P < NMS_P_LDS ? a.P : NMS_P_LDS;
  a.ext = nullptr;
  const size_t lds = (size_t)a.P_lds * 28;
  hipStream_t s = sm_hip_stream(stream);
  int32_t* cls_keep = (int32_t*)workspace;
  float* flat_key = (float*)((char*)workspace + (size_t)batch * num_classes * kmax * 4);
  int32_t* cls_cnt = (int32_t*)((char*)workspace + (size_t)batch * num_classes * kmax * 8);
  if (a.P > NMS_P_LDS)           // classes with more than NMS_P_LDS candidates work out of their global scratch slice
    a.ext = (unsigned char*)workspace + (((size_t)batch * num_classes * kmax * 8 + (size_t)batch * num_classes * 4 + 255) / 256) * 256;
  const HeavyLayout h = heavy_layout(batch, kmax, num_classes, nms_base_bytes(batch, kmax, num_classes));
  unsigned char* wsb = (unsigned char*)workspace;
  a.heavy_min = nms_heavy_min();
  a.heavy_slots = h.slots;
  a.rows = h.rows;
  a.W = h.W;
  a.heavy_cnt = (int32_t*)(wsb + h.off_cnt);
  a.heavy_meta = (int32_t*)(wsb + h.off_meta);
  a.hbox = (float4*)(wsb + h.off_box);
  a.hidx = (uint32_t*)(wsb + h.off_idx);
  a.hmat = (unsigned long long*)(wsb + h.off_mat);
  if (sm_lds_optin((const void*)nms_class_kernel, (int)lds) != hipSuccess) return SM_ERR_LAUNCH;
  if (h.slots > 0) hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(64), 0, s, a.heavy_cnt, 1, 0);
  hipLaunchKernelGGL(nms_class_kernel, dim3(num_classes, batch), dim3(NMS_THREADS), lds, s, boxes, scores, ctr, ncand, cls_keep,
                     cls_cnt, a);
  if (h.slots > 0) {
    hipLaunchKernelGGL(nms_heavy_matrix_kernel, dim3(128, h.slots), dim3(256), 0, s, a);
    const size_t slds = (size_t)a.P * 8 + (size_t)h.rows * 4;
    int st;
    if (h.W <= 64) st = launch_heavy_scan<1, 32>(h.slots, slds, s, cls_keep, cls_cnt, a);
    else st = launch_heavy_scan<2, 16>(h.slots, slds, s, cls_keep, cls_cnt, a);
    if (st != SM_OK) return st;
  }
  hipLaunchKernelGGL(nms_final_kernel, dim3(batch), dim3(TK_THREADS), 0, s, boxes, scores, ctr, cls_keep, cls_cnt,
                     flat_key, det, labels, keep, ndet, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_fast_nms(const float* boxes, const float* scores, const float* ctr, const int32_t* ncand, int batch,
                           int kmax, int num_classes, float score_thr, float iou_thr, int top_k, int max_num,
                           float* det, int64_t* labels, int64_t* keep, int32_t* ndet, void* workspace,
                           sm_stream_t stream) {
  if (!boxes || !scores || !ctr || !ncand || !det || !labels || !keep || !ndet || !workspace) return SM_ERR_BAD_ARG;
  if (batch < 1 || kmax < 1 || num_classes < 1 || num_classes > 256) return SM_ERR_BAD_SHAPE;
  if (max_num < 1 || max_num > TK_CAP || top_k < 1 || top_k > TK_CAP) return SM_ERR_UNSUPPORTED;
  NmsArgs a = {};
  a.batch = batch;
  a.kmax = kmax;
  a.C = num_classes;
  a.P = 0;
  a.max_num = max_num;
  a.score_thr = score_thr;
  a.iou_thr = iou_thr;
  a.always_sort = 1;
  a.top_k = top_k;
  a.P_lds = 0;
  a.ext = nullptr;
  hipStream_t s = sm_hip_stream(stream);
  // same workspace layout as sm_multiclass_nms: cls_keep | flat_key | cls_cnt
  int32_t* cls_keep = (int32_t*)workspace;
  float* flat_key = (float*)((char*)workspace + (size_t)batch * num_classes * kmax * 4);
  int32_t* cls_cnt = (int32_t*)((char*)workspace + (size_t)batch * num_classes * kmax * 8);
  hipLaunchKernelGGL(fast_nms_class_kernel, dim3(num_classes, batch), dim3(TK_THREADS), 0, s, boxes, scores, ctr, ncand,
                     flat_key, cls_keep, cls_cnt, a);
  hipLaunchKernelGGL(nms_final_kernel, dim3(batch), dim3(TK_THREADS), 0, s, boxes, scores, ctr, cls_keep, cls_cnt,
                     flat_key, det, labels, keep, ndet, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

// ---------------------------------------------------------------- evaluation-workload injection (bench.py --det-boxes)
// One launch: while *flag != 0, overwrite x1, y1, x2, y2 of the kept detections det [n][det_stride] with set (*counter % nsets)
// of `sets` [nsets][n][4] and advance the counter; scores (column 4), labels, kept indices and coefficients stay the
// detector's.  Random-weight detections are tiny boxes; an evaluation run's are not -- this gives mask assembly / RLE the
// rectangles of a real one without touching the detector.  Capture-safe (counter and flag live on the device).
namespace {
__global__ __launch_bounds__(256) void det_boxes_override_kernel(float* __restrict__ det, int det_stride,
                                                                 const float* __restrict__ sets, int nsets, int n,
                                                                 int* __restrict__ counter, const unsigned char* __restrict__ flag) {
  const int c = *counter;                         // every thread reads the old value ...
  const bool on = *flag != 0;
  __syncthreads();
  if (threadIdx.x == 0) *counter = c + 1;         // ... before one thread advances it
  if (!on) return;
  const float* src = sets + (long long)((c + 1) % nsets) * n * 4;
  for (int i = threadIdx.x; i < n * 4; i += blockDim.x) det[(long long)(i >> 2) * det_stride + (i & 3)] = src[i];
}
}  // namespace

extern "C" int sm_det_boxes_override(float* det, int det_stride, const float* sets, int nsets, int n, int32_t* counter,
                                     const uint8_t* flag, sm_stream_t stream) {
  if (!det || !sets || !counter || !flag) return SM_ERR_BAD_ARG;
  if (det_stride < 4 || nsets < 1 || n < 1) return SM_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(det_boxes_override_kernel, dim3(1), dim3(256), 0, sm_hip_stream(stream), det, det_stride, sets, nsets, n,
                     counter, flag);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int64_t sm_nms_workspace(int n) { (void)n; return 16; }

extern "C" int sm_nms(const float* dets, int n, float iou_thr, int64_t* keep, int32_t* nkeep, void* workspace,
                      sm_stream_t stream) {
  (void)workspace;
  if (!dets || !keep || !nkeep || n < 0) return SM_ERR_BAD_ARG;
  hipStream_t s = sm_hip_stream(stream);
  if (n == 0) {
    if (sm_zero_async(nkeep, sizeof(int32_t), s) != hipSuccess) return SM_ERR_LAUNCH;
    return SM_OK;
  }
  const int P = next_pow2(n);
  const size_t lds = (size_t)P * 28;
  if (lds > 160 * 1024) return SM_ERR_UNSUPPORTED;
  if (sm_lds_optin((const void*)nms_single_kernel, (int)lds) != hipSuccess) return SM_ERR_LAUNCH;
  hipLaunchKernelGGL(nms_single_kernel, dim3(1), dim3(NMS_THREADS), lds, s, dets, n, iou_thr, P, keep, nkeep);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
