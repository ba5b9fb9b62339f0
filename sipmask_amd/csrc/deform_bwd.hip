// Deformable conv v1 backward (training side of FeatureAlign).
//
// Replaces deform_conv_backward_input_cuda / deform_conv_backward_parameters_cuda
// (M/mmdet/ops/dcn/src/deform_conv_cuda.cpp:262-490) and their kernels deformable_col2im_gpu_kernel,
// deformable_col2im_coord_gpu_kernel, deformable_im2col_gpu_kernel (deform_conv_cuda_kernel.cu:191-433).
//
// Both GEMMs run on the MFMA implicit-GEMM kernel of conv_igemm.hip as 1x1 convolutions:
//   grad columns  gcol[p][k]  = sum_co gout[p][co] * W[co][k]         (A = gout rows, B = W^T prepared by the host)
//   grad weight   dW^T[k][co] = sum_p  col^T[k][p] * gout^T[co][p]    (split-K: S position slices as ONE batched launch,
//                                                                   operands transposed here, partials reduced after)
// and the data-dependent parts are HBM-bound gather/scatter kernels:
//   * col2im: one wave per (position, tap, 64 channels): the 64 lanes scatter their column gradient to the 4
//     bilinear corners of grad_x (f32 NHWC, hardware float atomics on contiguous 256-B segments) and reduce the
//     offset gradient over the channels of the deformable group with wave shuffles;
//   * im2col^T / gout^T: produce the K-major operands of the weight-gradient GEMM.
#include <string.h>

#include <algorithm>

#include "common.h"

namespace {

constexpr int DB_SLICE = 8192;   // positions per split-K slice of the weight-gradient GEMM

struct DBArgs {
  const uint16_t* x;
  const float* offset;
  const uint16_t* gout;
  int nlev, batch;
  int in_h[SM_MAX_LEVELS], in_w[SM_MAX_LEVELS], out_h[SM_MAX_LEVELS], out_w[SM_MAX_LEVELS];
  long long in_row0[SM_MAX_LEVELS], out_row0[SM_MAX_LEVELS];
  int cin, cout, kh, kw, stride, pad, dil, in_cstride, gout_cstride, G;
  long long P;   // total output positions (all levels, all images), compact order = level, image, y, x
};

struct Pos {
  int l, b, oy, ox;
  long long orow;   // row of this position in gout / offset
};

__device__ __forceinline__ Pos locate(const DBArgs& a, long long p) {
  Pos r;
  r.l = 0;
  for (int l = 0; l < a.nlev; ++l) {
    const long long n = (long long)a.batch * a.out_h[l] * a.out_w[l];
    if (p < n || l == a.nlev - 1) {
      r.l = l;
      break;
    }
    p -= n;
  }
  const int hw = a.out_h[r.l] * a.out_w[r.l];
  r.b = (int)(p / hw);
  const int rem = (int)(p - (long long)r.b * hw);
  r.oy = rem / a.out_w[r.l];
  r.ox = rem - r.oy * a.out_w[r.l];
  r.orow = a.out_row0[r.l] + p;
  return r;
}

struct Sample {
  bool valid;
  int hl, wl;
  float lh, lw;
};

// sampling point of (position, tap, deformable group): deform_conv_cuda_kernel.cu:216-229
__device__ __forceinline__ Sample sample_of(const DBArgs& a, const Pos& ps, int tap, int g) {
  const int kk = a.kh * a.kw;
  const int i = tap / a.kw, j = tap - i * a.kw;
  float o0 = 0.f, o1 = 0.f;             // a plain convolution is the offset-free special case
  if (a.offset != nullptr) {
    const float* o = a.offset + ps.orow * (long long)(a.G * kk * 2) + (g * kk + tap) * 2;
    o0 = o[0];
    o1 = o[1];
  }
  const float h_im = (float)(ps.oy * a.stride - a.pad + i * a.dil) + o0;
  const float w_im = (float)(ps.ox * a.stride - a.pad + j * a.dil) + o1;
  Sample s;
  s.valid = h_im > -1.f && w_im > -1.f && h_im < (float)a.in_h[ps.l] && w_im < (float)a.in_w[ps.l];
  const float fh = floorf(h_im), fw = floorf(w_im);
  s.hl = (int)fh;
  s.wl = (int)fw;
  s.lh = h_im - fh;
  s.lw = w_im - fw;
  return s;
}

// ---------------------------------------------------------------- grad_input + grad_offset
__global__ __launch_bounds__(256) void deform_col2im_kernel(const DBArgs a, const uint16_t* __restrict__ gcol,
                                                            float* __restrict__ gx, float* __restrict__ goff,
                                                            long long nunits) {
  const int lane = threadIdx.x & 63;
  const int kk = a.kh * a.kw;
  const int nq = a.cin >> 6;
  const int cpg = a.cin / a.G;
  const long long K = (long long)kk * a.cin;
  for (long long u = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); u < nunits; u += (long long)gridDim.x * 4) {
    const int q = (int)(u % nq);
    const int tap = (int)((u / nq) % kk);
    const long long p = u / ((long long)nq * kk);
    const Pos ps = locate(a, p);
    const int c = q * 64 + lane;
    const int g = (q * 64) / cpg;
    const Sample s = sample_of(a, ps, tap, g);
    if (!s.valid) continue;   // wave-uniform: contributes nothing (kernel.cu:423-426; no in-bounds neighbour :324-327)
    const float top = bf16_bits_to_f32((uint32_t)gcol[p * K + (long long)tap * a.cin + c]);
    const int H = a.in_h[ps.l], W = a.in_w[ps.l];
    const long long base = a.in_row0[ps.l] + (long long)ps.b * H * W;
    float dh = 0.f, dw = 0.f;
    // corners: (hl,wl) (hl,wh) (hh,wl) (hh,wh); weights get_gradient_weight :117-142,
    // coordinate weights get_coordinate_weight :144-188
    const float hh = 1.f - s.lh, hw = 1.f - s.lw;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int hc = s.hl + (k >> 1), wc = s.wl + (k & 1);
      if (hc < 0 || hc > H - 1 || wc < 0 || wc > W - 1) continue;
      const float wy = (k >> 1) ? s.lh : hh, wx = (k & 1) ? s.lw : hw;
      const long long row = base + (long long)hc * W + wc;
      const float v = bf16_bits_to_f32((uint32_t)a.x[row * a.in_cstride + c]);
      if (gx) unsafeAtomicAdd(gx + row * a.cin + c, top * (wy * wx));
      dh += ((k >> 1) ? wx : -wx) * v;
      dw += ((k & 1) ? wy : -wy) * v;
    }
    if (goff) {
      float th = top * dh, tw = top * dw;
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        th += __shfl_down(th, d, 64);
        tw += __shfl_down(tw, d, 64);
      }
      if (lane == 0) {
        float* o = goff + ps.orow * (long long)(a.G * kk * 2) + (g * kk + tap) * 2;
        if (cpg == 64) {
          o[0] = th;
          o[1] = tw;
        } else {
          unsafeAtomicAdd(o, th);
          unsafeAtomicAdd(o + 1, tw);
        }
      }
    }
  }
}

// 3x3 kernels (every deformable conv of the SipMask configs): ONE wave per (position, 64-channel chunk) with the nine
// taps unrolled.  The generic kernel above gives a wave one (position, tap, chunk) unit: one locate(), one offset load,
// one grad-column load and four x loads per unit, each waiting for the one before -- 3.2 M such dependent chains for the
// B=4 head, 3.07 ms (profiles/r02_rocprofv3_kernel_stats_train_step.csv).  Here a wave decodes its position once, fetches
// the group's 18 offsets with one 72-byte load (lanes 0-17; the taps read them with v_readlane), and issues its 9
// grad-column loads and all 36 corner loads UNCONDITIONALLY (clamped addresses, validity folded into the weights) before the
// first use, so a wave has 45 loads in flight instead of one.  The offset gradients -- 18 sums over the 64 channels -- are
// folded with v_permlane32_swap (two values per swap: the low half keeps d/dh, the high half d/dw) and finished by the
// transposing half-wave reduce of common.h: 9 swaps + 16 shuffles instead of 108.
// body: one (position, 64-channel chunk); `scatter(row, hc, wc, value)` takes the d(x) contribution of this lane's channel
// at input pixel (hc, wc) = row `row` of the level's image (wave-uniform arguments except the value)
// far_only: d(x) of the taps whose offsets stay within DX_OMAX is produced by deform_dx_gather_kernel (below); this kernel then
// scatters the remaining, far taps only (and still computes every offset gradient)
constexpr float DX_OMAX = 3.0f;
template <typename Scatter>
__device__ __forceinline__ void col2im9_position(const DBArgs& a, const Pos& ps, const long long p, const int q, const int lane,
                                                 const uint16_t* __restrict__ gcol, const bool want_gx,
                                                 float* __restrict__ goff, Scatter&& scatter, const bool far_only = false) {
  const int l31 = lane & 31;
  const int cpg = a.cin / a.G;
  const long long K = 9ll * a.cin;
  const int c = q * 64 + lane;
  const int g = (q * 64) / cpg;
  const int H = a.in_h[ps.l], W = a.in_w[ps.l];
  const long long base = a.in_row0[ps.l] + (long long)ps.b * H * W;
  const long long orow18 = ps.orow * (long long)(a.G * 18) + g * 18;
  float ofs = 0.f;
  if (a.offset != nullptr && lane < 18) ofs = a.offset[orow18 + lane];
  uint16_t topb[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) topb[t] = gcol[p * K + (long long)t * a.cin + c];
  // phase 1: sampling geometry of the nine taps, all corner loads
  float lhs[9], lws[9];
  int inm[9];                                    // bit 0 / 1: corner row hl / hl+1 inside the image, bit 2 / 3: column wl / wl+1
  int hls[9], wls[9];
  uint16_t xv[9][4];
  unsigned farm = 0u;                            // bit t: tap t samples farther than DX_OMAX from its tap (wave-uniform)
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int i = t / 3, j = t - 3 * (t / 3);
    const float o0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ofs), 2 * t));
    const float o1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ofs), 2 * t + 1));
    if (!(fabsf(o0) <= DX_OMAX && fabsf(o1) <= DX_OMAX)) farm |= 1u << t;
    const float h_im = (float)(ps.oy * a.stride - a.pad + i * a.dil) + o0;
    const float w_im = (float)(ps.ox * a.stride - a.pad + j * a.dil) + o1;
    const bool valid = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;   // kernel.cu:229
    const float fh = floorf(h_im), fw = floorf(w_im);
    const int hl = valid ? (int)fh : 0, wl = valid ? (int)fw : 0;
    hls[t] = hl;
    wls[t] = wl;
    lhs[t] = h_im - fh;
    lws[t] = w_im - fw;
    inm[t] = valid ? ((hl >= 0 ? 1 : 0) | (hl + 1 <= H - 1 ? 2 : 0) | (wl >= 0 ? 4 : 0) | (wl + 1 <= W - 1 ? 8 : 0)) : 0;
    const int h0 = min(max(hl, 0), H - 1), h1 = min(max(hl + 1, 0), H - 1);
    const int w0 = min(max(wl, 0), W - 1), w1 = min(max(wl + 1, 0), W - 1);
    xv[t][0] = a.x[(base + (long long)h0 * W + w0) * a.in_cstride + c];
    xv[t][1] = a.x[(base + (long long)h0 * W + w1) * a.in_cstride + c];
    xv[t][2] = a.x[(base + (long long)h1 * W + w0) * a.in_cstride + c];
    xv[t][3] = a.x[(base + (long long)h1 * W + w1) * a.in_cstride + c];
  }
  // phase 2: scatter d(x), accumulate d(offset).  Corner (row bit, column bit) carries weight wy * wx
  // (get_gradient_weight, kernel.cu:117-142); d/dh = +-wx * x, d/dw = +-wy * x (get_coordinate_weight :144-188); a
  // corner outside the image contributes to neither (its flags, not its weights, decide: lh == 0 is a legal weight).
  float fold[16];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float top = bf16_bits_to_f32((uint32_t)topb[t]);
    float dh = 0.f, dw = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool in = ((inm[t] >> (k >> 1)) & 1) && ((inm[t] >> (2 + (k & 1))) & 1);      // wave-uniform
      if (!in) continue;
      const float wy = (k >> 1) ? lhs[t] : 1.f - lhs[t], wx = (k & 1) ? lws[t] : 1.f - lws[t];
      const float v = bf16_bits_to_f32((uint32_t)xv[t][k]);
      const int hc = hls[t] + (k >> 1), wc = wls[t] + (k & 1);
      if (want_gx && (!far_only || ((farm >> t) & 1u))) scatter(base + (long long)hc * W + wc, hc, wc, top * (wy * wx));
      dh += ((k >> 1) ? wx : -wx) * v;
      dw += ((k & 1) ? wy : -wy) * v;
    }
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(top * dh), __float_as_uint(top * dw), false, false);
    fold[t] = __uint_as_float(r[0]) + __uint_as_float(r[1]);       // low half: d/dh over lanes {l, l+32}; high half: d/dw
  }
  if (goff != nullptr) {
#pragma unroll
    for (int t = 9; t < 16; ++t) fold[t] = 0.f;
    const float tot = gn_half_wave_totals<16>(fold, l31);            // lane pair k = l31 >> 1 holds tap k
    const int k = l31 >> 1;
    if ((l31 & 1) == 0 && k < 9) {
      float* o = goff + orow18 + k * 2 + (lane >> 5);
      if (cpg == 64)
        *o = tot;                                                     // this wave owns the (position, group)
      else
        unsafeAtomicAdd(o, tot);
    }
  }
}

__global__ __launch_bounds__(256) void deform_col2im9_kernel(const DBArgs a, const uint16_t* __restrict__ gcol,
                                                             float* __restrict__ gx, float* __restrict__ goff,
                                                             long long nunits, int far_only) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nq = a.cin >> 6;
  for (long long u = (long long)blockIdx.x * 4 + wv; u < nunits; u += (long long)gridDim.x * 4) {
    const int q = (int)(u % nq);
    const long long p = u / nq;
    const Pos ps = locate(a, p);
    const int c = q * 64 + lane;
    col2im9_position(a, ps, p, q, lane, gcol, gx != nullptr, goff,
                     [&](long long row, int, int, float v) { unsafeAtomicAdd(gx + row * a.cin + c, v); }, far_only != 0);
  }
}

// ---------------------------------------------------------------- d(x) as a GATHER (round 5)
// The scatter above retires 36 x 64 float atomics per (position, deformable group) in the L2: 826 M lane-adds = 1.85 ms of
// the B=4 head's 2.4 ms (profiles/r04_rocprofv3_kernel_stats_train_step.csv); privatising them in LDS was slower still
// (above).  Turned around: d(x)[pixel, c] = sum over the (position, tap) whose bilinear footprint contains the pixel of
// weight * gcol[position, tap, c] -- no atomics, one plain store per element, a deterministic order.  Which (position, tap)
// hit a pixel is data (the offsets), but a sample whose offsets stay within DX_OMAX of its tap can only come from an 8 x 8
// neighbourhood of positions per tap: 64 candidates = one wave-wide test per tap, ~36 hits per pixel on average.
// A block owns an 8 x 8 tile of pixels of one image, level and deformable group: it stages the group's 18 offsets of the
// 17 x 17 positions around the tile in LDS (coalesced), and each of its 4 waves takes 16 pixels in turn -- per tap one
// candidate per lane (offset from LDS, the forward's sampling arithmetic, hit = the sample's corner pair contains the
// pixel), hits compacted into a per-wave list (ballot + mbcnt), then the list is walked four entries at a time: lane c adds
// weight * gcol[entry][c] (one coalesced 128-byte read per entry, four in flight).  Samples farther than DX_OMAX are left to
// the scatter kernel (far_only), which runs BEHIND this kernel and adds them atomically.  3 x 3, stride 1, pad 1, 64
// channels per deformable group (FeatureAlign); everything else keeps the scatter.
constexpr int DXG_T = 8;                          // tile side (pixels)
constexpr int DXG_R = 17;                         // side of the staged position region: 8 + 2 + DX_OMAX below, 1 + DX_OMAX + ... above
constexpr int DXG_LIST = 9 * 64;                  // hit list capacity per wave

struct DxgArgs {
  int tile0[SM_MAX_LEVELS + 1];                   // first block of each level
  int ntx[SM_MAX_LEVELS], nty[SM_MAX_LEVELS];
  long long prow0[SM_MAX_LEVELS];                 // first compact position (row of gcol) of each level
};

__global__ __launch_bounds__(256) void deform_dx_gather_kernel(const DBArgs a, const DxgArgs t, const uint16_t* __restrict__ gcol,
                                                               float* __restrict__ gx) {
  __shared__ float s_off[DXG_R * DXG_R][18];      // the group's offsets of the region's positions (garbage outside the image)
  __shared__ unsigned s_list[4][DXG_LIST];        // per wave: gcol element offset of a hit's (position, tap); < 2^32 (checked by the host)
  __shared__ float s_lw[4][DXG_LIST];             // ... and its bilinear weight
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && (int)blockIdx.x >= t.tile0[l]) lev = l;
  const int H = a.in_h[lev], W = a.in_w[lev];
  int bi = blockIdx.x - t.tile0[lev];
  const int g = bi % a.G;
  bi /= a.G;
  const int tx = bi % t.ntx[lev];
  bi /= t.ntx[lev];
  const int ty = bi % t.nty[lev], b = bi / t.nty[lev];
  const int y0 = ty * DXG_T, x0 = tx * DXG_T;
  const int ry0 = y0 - 2 - (int)DX_OMAX, rx0 = x0 - 2 - (int)DX_OMAX;     // image coordinates of region position (0, 0)
  const long long orow_img = a.out_row0[lev] + (long long)b * H * W;       // row of position (0, 0) of this image in offset / gout
  const long long prow_img = t.prow0[lev] + (long long)b * H * W;         // ... in gcol (compact order)
  // ---- stage the offsets: 289 positions x 18 floats, consecutive threads read consecutive floats of a position
  for (int i = tid; i < DXG_R * DXG_R * 18; i += 256) {
    const int pos = i / 18, k = i - pos * 18;
    const int ry = pos / DXG_R, rx = pos - ry * DXG_R;
    const int oy = ry0 + ry, ox = rx0 + rx;
    float v = 0.f;
    if (oy >= 0 && oy < H && ox >= 0 && ox < W) v = a.offset[(orow_img + (long long)oy * W + ox) * (a.G * 18) + g * 18 + k];
    s_off[pos][k] = v;
  }
  __syncthreads();
  const long long K = 9ll * a.cin;
  const int c = g * 64 + lane;
  const int dy = lane >> 3, dx = lane & 7;
  for (int pi = wv; pi < DXG_T * DXG_T; pi += 4) {
    const int hc = y0 + (pi >> 3), wc = x0 + (pi & 7);
    if (hc >= H || wc >= W) continue;             // wave-uniform
    int n = 0;                                    // entries in this wave's list (wave-uniform)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int i = tp / 3, j = tp - 3 * (tp / 3);
      // candidates of this tap: a sample with |offset| <= DX_OMAX that touches row hc comes from oy in [hc - i - OMAX, hc + 1 - i + OMAX]
      const int oy = hc - i - (int)DX_OMAX + dy, ox = wc - j - (int)DX_OMAX + dx;
      const bool inimg = oy >= 0 && oy < H && ox >= 0 && ox < W;
      const int pos = (oy - ry0) * DXG_R + (ox - rx0);                     // inside the region by construction
      const float o0 = s_off[pos][2 * tp], o1 = s_off[pos][2 * tp + 1];
      const float h_im = (float)(oy - 1 + i) + o0, w_im = (float)(ox - 1 + j) + o1;
      const bool valid = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;      // kernel.cu:229
      const float fh = floorf(h_im), fw = floorf(w_im);
      const int dyc = hc - (int)fh, dxc = wc - (int)fw;                    // which corner of the sample this pixel is
      const bool hit = inimg && valid && fabsf(o0) <= DX_OMAX && fabsf(o1) <= DX_OMAX && (unsigned)dyc <= 1u && (unsigned)dxc <= 1u;
      const float lh = h_im - fh, lw = w_im - fw;
      const float wgt = (dyc ? lh : 1.f - lh) * (dxc ? lw : 1.f - lw);     // get_gradient_weight, kernel.cu:117-142
      const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
      if (hit) {
        const int rank = n + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        s_list[wv][rank] = (unsigned)((prow_img + (long long)oy * W + ox) * K + (long long)tp * a.cin);
        s_lw[wv][rank] = wgt;
      }
      n += __popcll(m);
    }
    // the list is written and read by this wave only (LDS operations of one wave complete in order); the fence + wave barrier
    // pin that order in the SOURCE too: no compiler may move the list reads below above the list stores
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int e = 0;
    for (; e + 4 <= n; e += 4) {
      const float v0 = bf16_bits_to_f32((uint32_t)gcol[s_list[wv][e] + c]);
      const float v1 = bf16_bits_to_f32((uint32_t)gcol[s_list[wv][e + 1] + c]);
      const float v2 = bf16_bits_to_f32((uint32_t)gcol[s_list[wv][e + 2] + c]);
      const float v3 = bf16_bits_to_f32((uint32_t)gcol[s_list[wv][e + 3] + c]);
      acc0 = fmaf(s_lw[wv][e], v0, acc0);
      acc1 = fmaf(s_lw[wv][e + 1], v1, acc1);
      acc2 = fmaf(s_lw[wv][e + 2], v2, acc2);
      acc3 = fmaf(s_lw[wv][e + 3], v3, acc3);
    }
    for (; e < n; ++e) acc0 = fmaf(s_lw[wv][e], bf16_bits_to_f32((uint32_t)gcol[s_list[wv][e] + c]), acc0);
    gx[(a.in_row0[lev] + (long long)b * H * W + (long long)hc * W + wc) * a.cin + c] = (acc0 + acc1) + (acc2 + acc3);
    // ... and the next pixel's list stores stay behind this pixel's list reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// (Round 3 A/B, removed: accumulating d(x) per 8 x 16-position tile in an f32 LDS window -- ds_add_f32, lane = channel --
// and flushing the window with one global atomic per touched pixel, 425 instead of 4608 per tile.  Bit-compatible with the
// direct scatter up to summation order and 2x SLOWER: 5.1 ms against 2.74 for d(x) + d(offset) of the B=4 head with 16 waves
// per window, 6.0 with 4 (profiles/r03_deform_bwd_scatter_ab.txt).  A 64-lane ds_add_f32 costs ~170 cycles here; the L2's
// float atomics retire 826 M lane-adds in 1.85 ms.  What the kernel above spends without any scatter: 0.88 ms.)

// ---------------------------------------------------------------- operands of the weight-gradient GEMM
// col^T[k][pl] (bf16, k = tap*cin + c) for positions p0 .. p0+n of the compact order; columns n..Lp are zero
// col^T, slice-major: colT[s][k][pl] (bf16, k = tap*cin + c, rows K..Kpad of a slice are never read as real data:
// the host zero-fills them once) for position p = s*L + pl; positions >= P are zero.
__global__ __launch_bounds__(256) void deform_im2col_t_kernel(const DBArgs a, int S, int L, int Kpad,
                                                              uint16_t* __restrict__ colT) {
  const int kk = a.kh * a.kw;
  const int nc8 = a.cin >> 3;
  const int cpg = a.cin / a.G;
  const long long SL = (long long)S * L;
  const long long total = (long long)kk * nc8 * SL;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long gp = t % SL;
    const int sl = (int)(gp / L), pl = (int)(gp - (long long)sl * L);
    const int c8 = (int)((t / SL) % nc8);
    const int tap = (int)(t / (SL * nc8));
    const int n = gp < a.P ? pl + 1 : 0;      // "pl < n" below == position exists
    const long long p0 = gp - pl;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (pl < n) {
      const Pos ps = locate(a, p0 + pl);
      const Sample s = sample_of(a, ps, tap, (c8 * 8) / cpg);
      if (s.valid) {
        const int H = a.in_h[ps.l], W = a.in_w[ps.l];
        const long long base = a.in_row0[ps.l] + (long long)ps.b * H * W;
        const float hh = 1.f - s.lh, hw = 1.f - s.lw;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int hc = s.hl + (k >> 1), wc = s.wl + (k & 1);
          if (hc < 0 || hc > H - 1 || wc < 0 || wc > W - 1) continue;
          const float wgt = ((k >> 1) ? s.lh : hh) * ((k & 1) ? s.lw : hw);
          const u32x4 raw = *reinterpret_cast<const u32x4*>(a.x + (base + (long long)hc * W + wc) * a.in_cstride + c8 * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] += wgt * __uint_as_float(raw[e] << 16);
            v[2 * e + 1] += wgt * __uint_as_float(raw[e] & 0xffff0000u);
          }
        }
      }
    }
    uint16_t* o = colT + ((long long)sl * Kpad + (long long)tap * a.cin + c8 * 8) * L + pl;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[(long long)e * L] = (uint16_t)f32_to_bf16_bits(v[e]);
  }
}

// The same operand, one thread per (position, tap, deformable group) instead of per (position, tap, 8 channels): with 64
// channels per group the sampling point -- locate(), the offset load, the bilinear set-up, ~150 VALU -- was recomputed
// for each of the group's 8 chunks (1.29 ms for the B=4 head, profiles/r03_rocprofv3_kernel_stats_train_step.csv).  Here
// it is computed once and the thread walks its group's channels 8 at a time: per chunk 4 corner loads of 16 bytes, the
// blend, 8 two-byte stores (coalesced across the lanes = consecutive positions of a K-major col^T row).
__global__ __launch_bounds__(256) void deform_im2col_t_group_kernel(const DBArgs a, int S, int L, int Kpad,
                                                                    uint16_t* __restrict__ colT) {
  const int kk = a.kh * a.kw;
  const int cpg = a.cin / a.G;                   // multiple of 8
  const long long SL = (long long)S * L;
  // one block = 256 consecutive positions x one (tap, group); XCD-aware order (workgroups go round-robin to the 8 XCDs): the
  // kk * G blocks of a position chunk read the same x rows (shifted by the taps) and run consecutively on XCD chunk % 8
  const int per_chunk = kk * a.G;
  const long long nchunk = (SL + 255) / 256;
  {
    const int xcd = blockIdx.x & 7;
    const long long kq = blockIdx.x >> 3;
    const long long chunk = (kq / per_chunk) * 8 + xcd;
    const int rq = (int)(kq % per_chunk);
    const long long gp = chunk * 256 + threadIdx.x;
    if (chunk >= nchunk || gp >= SL) return;
    const int sl = (int)(gp / L), pl = (int)(gp - (long long)sl * L);
    const int g = rq % a.G;
    const int tap = rq / a.G;
    uint16_t* o = colT + ((long long)sl * Kpad + (long long)tap * a.cin + g * cpg) * L + pl;
    bool live = gp < a.P;
    Pos ps;
    Sample s;
    if (live) {
      ps = locate(a, gp);
      s = sample_of(a, ps, tap, g);
      live = s.valid;
    }
    if (!live) {                                  // positions beyond P and samples outside the image: zero columns
      for (int c = 0; c < cpg; ++c) o[(long long)c * L] = 0;
      return;
    }
    const int H = a.in_h[ps.l], W = a.in_w[ps.l];
    const long long base = a.in_row0[ps.l] + (long long)ps.b * H * W;
    const float hh = 1.f - s.lh, hw = 1.f - s.lw;
    float wgt[4];
    const uint16_t* src[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int hc = s.hl + (k >> 1), wc = s.wl + (k & 1);
      const bool in = hc >= 0 && hc <= H - 1 && wc >= 0 && wc <= W - 1;
      wgt[k] = in ? ((k >> 1) ? s.lh : hh) * ((k & 1) ? s.lw : hw) : 0.f;
      const int hcc = min(max(hc, 0), H - 1), wcc = min(max(wc, 0), W - 1);
      src[k] = a.x + (base + (long long)hcc * W + wcc) * a.in_cstride + g * cpg;       // clamped: the load is unconditional
    }
    for (int c8 = 0; c8 < cpg; c8 += 8) {
      u32x4 raw[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) raw[k] = *reinterpret_cast<const u32x4*>(src[k] + c8);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] += wgt[k] * __uint_as_float(raw[k][e] << 16);
          v[2 * e + 1] += wgt[k] * __uint_as_float(raw[k][e] & 0xffff0000u);
        }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[(long long)(c8 + e) * L] = (uint16_t)f32_to_bf16_bits(v[e]);
    }
  }
}

// Tiled transposition for the offset-free case (every plain convolution): a block moves a 64-position x 64-channel
// tile per tap through LDS, so the reads are full 128-byte channel runs of the NHWC rows and the writes are 16-byte
// position runs of the K-major col^T rows (the generic kernel above gathers 16 B per lane from 64 different rows).
// MODE 0: col^T of x (taps, zero padding outside the image); MODE 1: gout^T (one "tap", no shift, rows >= cout zero).
template <int MODE>
__global__ __launch_bounds__(256) void transpose_tile_kernel(const DBArgs a, int S, int L, int rows_per_slice, int nch,
                                                             uint16_t* __restrict__ out) {
  __shared__ uint16_t tile[64][72];                  // [position][channel], 144-byte pitch: conflict-free both ways
  const int tid = threadIdx.x;
  const long long gp0 = (long long)blockIdx.x * 64;  // first position (global index over S*L) of the tile
  const int c0 = blockIdx.y * 64;                    // first channel
  const int sl = (int)(gp0 / L), pl0 = (int)(gp0 - (long long)sl * L);
  const int ntap = MODE == 0 ? a.kh * a.kw : 1;
  // load role: thread -> (position tid>>2... two passes), 16-byte channel chunk
  const int lp = tid >> 3, lc = (tid & 7) * 8;       // 32 positions per pass, 8 chunks of 8 channels
  // store role: thread -> (channel tid>>2... two passes), 16-byte position chunk
  const int sc = tid >> 3, sp = (tid & 7) * 8;
  Pos ps[2];
  bool pv[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const long long gp = gp0 + lp + 32 * h;
    pv[h] = gp < a.P;
    if (pv[h]) ps[h] = locate(a, gp);
  }
  for (int tap = 0; tap < ntap; ++tap) {
    const int ti = MODE == 0 ? tap / a.kw : 0, tj = MODE == 0 ? tap - ti * a.kw : 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (pv[h] && c0 + lc < nch) {
        if (MODE == 0) {
          const int iy = ps[h].oy * a.stride - a.pad + ti * a.dil, ix = ps[h].ox * a.stride - a.pad + tj * a.dil;
          const int Hh = a.in_h[ps[h].l], Ww = a.in_w[ps[h].l];
          if (iy >= 0 && iy < Hh && ix >= 0 && ix < Ww) {
            const long long row = a.in_row0[ps[h].l] + ((long long)ps[h].b * Hh + iy) * Ww + ix;
            v = *reinterpret_cast<const u32x4*>(a.x + row * a.in_cstride + c0 + lc);
          }
        } else {
          v = *reinterpret_cast<const u32x4*>(a.gout + ps[h].orow * a.gout_cstride + c0 + lc);
        }
      }
      *reinterpret_cast<u32x4*>(&tile[lp + 32 * h][lc]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = sc + 32 * h;
      if (c0 + c < rows_per_slice - (MODE == 0 ? tap * 0 : 0)) {
        uint16_t r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = tile[sp + e][c];
        u32x4 w;
        w[0] = r[0] | ((uint32_t)r[1] << 16);
        w[1] = r[2] | ((uint32_t)r[3] << 16);
        w[2] = r[4] | ((uint32_t)r[5] << 16);
        w[3] = r[6] | ((uint32_t)r[7] << 16);
        const long long krow = MODE == 0 ? (long long)tap * a.cin + c0 + c : (long long)(c0 + c);
        const bool ok = MODE == 0 ? (c0 + c < a.cin) : (c0 + c < rows_per_slice);
        if (ok) *reinterpret_cast<u32x4*>(out + ((long long)sl * rows_per_slice + krow) * L + pl0 + sp) = w;
      }
    }
    __syncthreads();
  }
}

// gout^T, slice-major: goutT[s][co][pl] (bf16); rows cout..cout_pad and positions >= P are zero
__global__ __launch_bounds__(256) void gout_t_kernel(const DBArgs a, int S, int L, int cout_pad,
                                                     uint16_t* __restrict__ goutT) {
  const long long total = (long long)S * cout_pad * L;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int pl = (int)(t % L);
    const int co = (int)((t / L) % cout_pad);
    const int sl = (int)(t / ((long long)L * cout_pad));
    const long long gp = (long long)sl * L + pl;
    uint16_t v = 0;
    if (gp < a.P && co < a.cout) {
      const Pos ps = locate(a, gp);
      v = a.gout[ps.orow * a.gout_cstride + co];
    }
    goutT[t] = v;
  }
}

// grad_w_t[k][co] = sum over the S slices of part[s][k][co] (rows k < K of each Kpad-row slice)
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int S, long long K, int Kpad,
                                    int cout) {
  const long long n = K * cout;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += part[(long long)s * Kpad * cout + i];
    out[i] = acc;
  }
}

// grad_bias[co] = sum over all positions of gout[p][co]: one block per 8 channels, rows strided over the threads
__global__ __launch_bounds__(256) void bias_grad_kernel(const DBArgs a, float* __restrict__ gbias) {
  __shared__ float s_red[256][8];
  const int c0 = blockIdx.x * 8, tid = threadIdx.x;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (long long p = tid; p < a.P; p += 256) {
    const Pos ps = locate(a, p);
    const u32x4 raw = *reinterpret_cast<const u32x4*>(a.gout + ps.orow * a.gout_cstride + c0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[2 * e] += __uint_as_float(raw[e] << 16);
      acc[2 * e + 1] += __uint_as_float(raw[e] & 0xffff0000u);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) s_red[tid][e] = acc[e];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st)
#pragma unroll
      for (int e = 0; e < 8; ++e) s_red[tid][e] += s_red[tid + st][e];
    __syncthreads();
  }
  if (tid < 8 && c0 + tid < a.cout) gbias[c0 + tid] = s_red[0][tid];
}

struct Plan {
  long long P, K;
  int S, L, Kpad, cout_pad2;      // weight-gradient GEMM: S slices of L positions, K rows padded per slice
  int kpad1;                      // grad-column GEMM: K padded to its cout tile
  size_t off_gcol, off_colT, off_goutT, off_part, total;
};

bool make_plan(const sm_conv_desc* d, Plan* pl) {
  pl->P = 0;
  for (int l = 0; l < d->nlev; ++l) pl->P += (long long)d->batch * d->out_h[l] * d->out_w[l];
  pl->K = (long long)d->kh * d->kw * d->cin;
  // split-K: one batched GEMM over S slices of the position axis (enough blocks to fill the chip even though the
  // result is only K x cout), partial sums reduced afterwards
  pl->Kpad = (int)((pl->K + 255) / 256 * 256);   // every position tile (<= 256 rows) stays inside one slice
  const int t2 = sm_conv_cout_tile(d->cout);
  pl->cout_pad2 = (d->cout + t2 - 1) / t2 * t2;
  // slice length: the GEMM of one slice has only ceil(Kpad/128) x ceil(cout/128) output tiles (4 for a 512 -> 128 1x1
  // conv), so small layers take MORE, shorter slices until ~1024 tiles exist -- bounded below by the length at which the
  // f32 partial slabs (S x Kpad x cout, written and re-read) would outweigh the bf16 operands (P x (Kpad + cout))
  {
    const long long tiles = ((pl->Kpad + 127) / 128) * (long long)((pl->cout_pad2 + 127) / 128);
    const long long s_want = (1024 + tiles - 1) / tiles;
    long long L = (pl->P + s_want - 1) / s_want;
    const long long l_min = 4ll * pl->Kpad * pl->cout_pad2 / (pl->Kpad + pl->cout_pad2);
    if (L < l_min) L = l_min;
    if (L < 256) L = 256;
    if (L > DB_SLICE) L = DB_SLICE;
    L = (L + 63) / 64 * 64;
    const long long Pr = (pl->P + 63) / 64 * 64;
    pl->L = (int)(L < Pr ? L : Pr);
  }
  pl->S = (int)((pl->P + pl->L - 1) / pl->L);
  const int t1 = sm_conv_cout_tile((int)pl->K);
  pl->kpad1 = (int)((pl->K + t1 - 1) / t1 * t1);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += (bytes + 255) / 256 * 256;
    return at;
  };
  pl->off_gcol = take((size_t)pl->P * pl->K * 4);
  pl->off_colT = take((size_t)pl->S * pl->Kpad * pl->L * 2);
  pl->off_goutT = take((size_t)pl->S * pl->cout_pad2 * pl->L * 2);
  pl->off_part = take((size_t)pl->S * pl->Kpad * d->cout * 4);
  pl->total = o;
  return true;
}

}  // namespace

extern "C" int64_t sm_deform_conv2d_bwd_workspace(const sm_conv_desc* d) {
  if (!d || d->nlev < 1 || d->nlev > SM_MAX_LEVELS || d->batch < 1) return 0;
  Plan pl;
  make_plan(d, &pl);
  return (int64_t)pl.total;
}

static int conv_bwd_impl(const sm_conv_desc* d, const void* x, const float* offset, const void* w_t, const void* w_dgrad,
                         const void* gout, float* grad_x, float* grad_offset, float* grad_w_t, float* grad_bias,
                         void* workspace, sm_stream_t stream) {
  if (!d || !x || !gout || !workspace) return SM_ERR_BAD_ARG;
  if (d->nlev < 1 || d->nlev > SM_MAX_LEVELS || d->batch < 1) return SM_ERR_BAD_SHAPE;
  const int G = offset ? d->deform_groups : 1;
  if (d->cin % 8 != 0 || d->cout % 8 != 0 || d->out_cstride % 8 != 0 || d->in_cstride % 8 != 0) return SM_ERR_UNSUPPORTED;
  if (offset && G < 1) return SM_ERR_UNSUPPORTED;
  // fast input gradient of a plain stride-1 conv: one forward implicit GEMM over gout with the flipped,
  // transposed weights (w_dgrad); everything else goes through grad columns + col2im
  const bool fast_dgrad = grad_x && !offset && w_dgrad && d->stride == 1;
  const bool col_path = (grad_x && !fast_dgrad) || grad_offset;
  if ((d->flags & SM_CONV_BWD_GX_BF16) && grad_x && !fast_dgrad) return SM_ERR_UNSUPPORTED;   // col2im accumulates in f32
  if (col_path && (!w_t || d->cin % 64 != 0 || (d->cin / G) % 64 != 0)) return w_t ? SM_ERR_UNSUPPORTED : SM_ERR_BAD_ARG;
  hipStream_t s = sm_hip_stream(stream);
  Plan pl;
  make_plan(d, &pl);
  char* ws = (char*)workspace;
  DBArgs a;
  a.x = (const uint16_t*)x;
  a.offset = offset;
  a.gout = (const uint16_t*)gout;
  a.nlev = d->nlev;
  a.batch = d->batch;
  long long in_rows = 0;
  for (int l = 0; l < d->nlev; ++l) {
    a.in_h[l] = d->in_h[l], a.in_w[l] = d->in_w[l], a.out_h[l] = d->out_h[l], a.out_w[l] = d->out_w[l];
    a.in_row0[l] = d->in_row0[l], a.out_row0[l] = d->out_row0[l];
    in_rows = std::max<long long>(in_rows, d->in_row0[l] + (long long)d->batch * d->in_h[l] * d->in_w[l]);
  }
  // the compact position order must coincide with the rows of gout/offset inside a level (it does by
  // construction: row = out_row0[l] + (b*oh + y)*ow + x); levels may sit anywhere
  a.cin = d->cin, a.cout = d->cout, a.kh = d->kh, a.kw = d->kw, a.stride = d->stride, a.pad = d->pad, a.dil = d->dil;
  a.in_cstride = d->in_cstride, a.gout_cstride = d->out_cstride, a.G = G;
  a.P = pl.P;
  const int kk = d->kh * d->kw;

  if (grad_bias) {
    hipLaunchKernelGGL(bias_grad_kernel, dim3((d->cout + 7) / 8), dim3(256), 0, s, a, grad_bias);
    SM_LAUNCH_CHECK();
  }
  if (fast_dgrad) {
    // dX = conv(gout, flip(W)^T) with pad' = dil*(k-1) - pad: same rows as x, f32
    sm_conv_desc g0;
    memset(&g0, 0, sizeof(g0));
    g0.nlev = d->nlev;
    g0.batch = d->batch;
    for (int l = 0; l < d->nlev; ++l) {
      g0.in_h[l] = d->out_h[l], g0.in_w[l] = d->out_w[l];
      g0.out_h[l] = d->in_h[l], g0.out_w[l] = d->in_w[l];
      g0.in_row0[l] = d->out_row0[l];
      g0.out_row0[l] = d->in_row0[l];
    }
    g0.cin = d->cout;
    g0.cout = d->cin;
    const int t0 = sm_conv_cout_tile(d->cin);
    g0.cout_pad = (d->cin + t0 - 1) / t0 * t0;
    g0.kh = d->kh, g0.kw = d->kw, g0.stride = 1, g0.dil = d->dil;
    g0.pad = d->dil * (d->kh - 1) - d->pad;
    if (g0.pad < 0 || d->kh != d->kw) return SM_ERR_UNSUPPORTED;
    g0.in_cstride = d->out_cstride;
    g0.out_cstride = d->cin;
    g0.flags = (d->flags & SM_CONV_BWD_GX_BF16) ? 0u : SM_CONV_OUT_F32;      // bf16 rows for the row-tensor training graph
    g0.flags |= d->flags & (SM_CONV_DBG_TILE256 | SM_CONV_DBG_HAND_PLACED);  // the caller's tile choice for the dX conv (same results)
    const int st = sm_conv2d(&g0, gout, w_dgrad, nullptr, nullptr, grad_x, stream);
    if (st != SM_OK) return st;
  }
  if (col_path) {
    // ---- grad columns = gout @ W  as a 1x1 conv  (cin := cout, cout := K)
    sm_conv_desc g1;
    memset(&g1, 0, sizeof(g1));
    g1.nlev = d->nlev;
    g1.batch = d->batch;
    long long prow = 0;
    for (int l = 0; l < d->nlev; ++l) {
      g1.in_h[l] = g1.out_h[l] = d->out_h[l];
      g1.in_w[l] = g1.out_w[l] = d->out_w[l];
      g1.in_row0[l] = d->out_row0[l];
      g1.out_row0[l] = prow;
      prow += (long long)d->batch * d->out_h[l] * d->out_w[l];
    }
    g1.cin = d->cout;
    g1.cout = (int)pl.K;
    g1.cout_pad = pl.kpad1;
    g1.kh = g1.kw = 1, g1.stride = 1, g1.pad = 0, g1.dil = 1;
    g1.in_cstride = d->out_cstride;
    g1.out_cstride = (int)pl.K;
    g1.flags = 0;                                 // bf16 grad columns: half the bytes the scatter kernel has to read back
    uint16_t* gcol = (uint16_t*)(ws + pl.off_gcol);
    int st = sm_conv2d(&g1, gout, w_t, nullptr, nullptr, gcol, stream);
    if (st != SM_OK) return st;
    float* gx_col = fast_dgrad ? nullptr : grad_x;
    // d(x) of the near samples as a gather (deform_dx_gather_kernel): FeatureAlign's shape
    bool gather = gx_col && offset && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->dil == 1 &&
                  d->cin / G == 64 && !(d->flags & SM_CONV_BWD_DX_SCATTER) && pl.P * pl.K < (1ll << 32);
    long long covered = 0;
    for (int l = 0; l < d->nlev; ++l) {
      gather = gather && d->in_h[l] == d->out_h[l] && d->in_w[l] == d->out_w[l];
      covered += (long long)d->batch * d->in_h[l] * d->in_w[l];
    }
    // (the gather writes every element of every level's rows with plain stores BEFORE the far taps' atomics are
    // launched behind it: nothing to clear when the levels tile the row range)
    if (gx_col && !(gather && covered == in_rows) && sm_zero_async(gx_col, (size_t)in_rows * d->cin * 4, s) != hipSuccess)
      return SM_ERR_LAUNCH;
    if (grad_offset) {   // positions sampling outside the image are skipped by the kernel: their gradient is 0
      long long orows = 0;
      for (int l = 0; l < d->nlev; ++l)
        orows = std::max<long long>(orows, d->out_row0[l] + (long long)d->batch * d->out_h[l] * d->out_w[l]);
      if (sm_zero_async(grad_offset, (size_t)orows * G * kk * 2 * 4, s) != hipSuccess)
        return SM_ERR_LAUNCH;
    }
    if (gather) {
      DxgArgs t;
      long long nb = 0, prow2 = 0;
      for (int l = 0; l < SM_MAX_LEVELS; ++l) {
        const bool on = l < d->nlev;
        t.ntx[l] = on ? sm_cdiv(d->in_w[l], DXG_T) : 1;
        t.nty[l] = on ? sm_cdiv(d->in_h[l], DXG_T) : 1;
        t.tile0[l] = (int)nb;
        t.prow0[l] = prow2;
        if (on) {
          nb += (long long)d->batch * t.ntx[l] * t.nty[l] * G;
          prow2 += (long long)d->batch * d->out_h[l] * d->out_w[l];
        }
      }
      t.tile0[SM_MAX_LEVELS] = (int)nb;
      if (nb > 0x7fffffffLL) return SM_ERR_BAD_SHAPE;
      hipLaunchKernelGGL(deform_dx_gather_kernel, dim3((unsigned)nb), dim3(256), 0, s, a, t, gcol, gx_col);
    }
    if (d->kh == 3 && d->kw == 3) {               // one wave per (position, 64-channel chunk), nine taps unrolled
      const long long nunits = pl.P * (d->cin / 64);
      const int blocks = (int)std::min<long long>((nunits + 3) / 4, 256 * 64);
      // (with the gather in front: every offset gradient, and the d(x) atomics of the far taps only -- the gather leaves
      // those out, so the launch is needed for grad_x alone too; goff == nullptr is handled by the kernel)
      if (gx_col || grad_offset)
        hipLaunchKernelGGL(deform_col2im9_kernel, dim3(blocks), dim3(256), 0, s, a, gcol, gx_col, grad_offset, nunits,
                           gather ? 1 : 0);
    } else {
      const long long nunits = pl.P * kk * (d->cin / 64);
      const int blocks = (int)std::min<long long>((nunits + 3) / 4, 256 * 64);
      hipLaunchKernelGGL(deform_col2im_kernel, dim3(blocks), dim3(256), 0, s, a, gcol, gx_col, grad_offset, nunits);
    }
    SM_LAUNCH_CHECK();
  }
  if (grad_w_t && offset == nullptr && !(d->flags & SM_CONV_BWD_WGRAD_GEMM) &&
      ((d->flags & SM_CONV_BWD_WGRAD_DIRECT) ? sm_wgrad_direct_supported(d) : sm_wgrad_direct_preferred(d))) {
    // plain conv: dW straight from the NHWC rows (wgrad_direct.hip: transposing LDS reads, no im2col^T / gout^T, no slabs)
    const int st = sm_wgrad_direct(d, x, gout, grad_w_t, stream);
    if (st != SM_OK) return st;
  } else if (grad_w_t) {
    // ---- dW^T[k][co] = sum_s col^T_s @ gout_s: ONE launch of the implicit-GEMM kernel as a batch of S 1x1
    // "convolutions" (image s: Kpad rows, cin = L, its own weight matrix gout^T_s via w_batch_stride), then a reduce
    uint16_t* colT = (uint16_t*)(ws + pl.off_colT);
    uint16_t* goutT = (uint16_t*)(ws + pl.off_goutT);
    float* part = (float*)(ws + pl.off_part);
    // rows K..Kpad of every col^T slice are never written: they are extra ROWS of the GEMM's position operand, so they
    // only produce rows K..Kpad of the partial slabs, which wgrad_reduce_kernel never reads (zero-filling them was a
    // memset of the whole S x Kpad x L buffer per conv: 1.1 ms of a 33 ms training step)
    const int ptiles = (int)((long long)pl.S * pl.L / 64);      // L is a multiple of 64
    if (offset == nullptr) {
      hipLaunchKernelGGL(transpose_tile_kernel<0>, dim3(ptiles, (d->cin + 63) / 64), dim3(256), 0, s, a, pl.S, pl.L, pl.Kpad,
                         d->cin, colT);
    } else {
      if ((d->cin / G) % 8 == 0 && d->cin / G >= 32) {     // one thread per (position, tap, deformable group)
        const long long nchunk = ((long long)pl.S * pl.L + 255) / 256;
        const long long nblk = 8 * ((nchunk + 7) / 8) * kk * G;
        if (nblk > 0x7fffffffLL) return SM_ERR_BAD_SHAPE;
        hipLaunchKernelGGL(deform_im2col_t_group_kernel, dim3((unsigned)nblk), dim3(256), 0, s, a, pl.S, pl.L, pl.Kpad, colT);
      } else {
        const long long t1 = (long long)kk * (d->cin / 8) * pl.S * pl.L;
        hipLaunchKernelGGL(deform_im2col_t_kernel, dim3((int)std::min<long long>((t1 + 255) / 256, 256 * 64)), dim3(256), 0, s,
                           a, pl.S, pl.L, pl.Kpad, colT);
      }
    }
    // gout^T: rows cout..cout_pad2 of every slice must be zero (they are weight rows of the GEMM)
    hipLaunchKernelGGL(transpose_tile_kernel<1>, dim3(ptiles, (pl.cout_pad2 + 63) / 64), dim3(256), 0, s, a, pl.S, pl.L,
                       pl.cout_pad2, d->cout, goutT);
    SM_LAUNCH_CHECK();
    sm_conv_desc g2;
    memset(&g2, 0, sizeof(g2));
    g2.nlev = 1;
    g2.batch = pl.S;
    g2.in_h[0] = g2.out_h[0] = 1;
    g2.in_w[0] = g2.out_w[0] = pl.Kpad;
    g2.cin = pl.L;
    g2.cout = d->cout;
    g2.cout_pad = pl.cout_pad2;
    g2.kh = g2.kw = 1, g2.stride = 1, g2.pad = 0, g2.dil = 1;
    g2.in_cstride = pl.L;
    g2.out_cstride = d->cout;
    g2.flags = SM_CONV_OUT_F32;
    g2.w_batch_stride = (long long)pl.cout_pad2 * pl.L;
    const int st = sm_conv2d(&g2, colT, goutT, nullptr, nullptr, part, stream);
    if (st != SM_OK) return st;
    const long long nout = pl.K * d->cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)std::min<long long>((nout + 255) / 256, 4096)), dim3(256), 0, s, part,
                       grad_w_t, pl.S, pl.K, pl.Kpad, d->cout);
    SM_LAUNCH_CHECK();
  }
  return SM_OK;
}

extern "C" int sm_deform_conv2d_bwd(const sm_conv_desc* d, const void* x, const float* offset, const void* w_t,
                                    const void* gout, float* grad_x, float* grad_offset, float* grad_w_t,
                                    void* workspace, sm_stream_t stream) {
  if (!offset) return SM_ERR_BAD_ARG;
  if (d && (d->cin % 64 != 0 || d->deform_groups < 1 || (d->cin / d->deform_groups) % 64 != 0)) return SM_ERR_UNSUPPORTED;
  return conv_bwd_impl(d, x, offset, w_t, nullptr, gout, grad_x, grad_offset, grad_w_t, nullptr, workspace, stream);
}

extern "C" int sm_conv2d_bwd(const sm_conv_desc* d, const void* x, const void* w_t, const void* w_dgrad, const void* gout,
                             float* grad_x, float* grad_w_t, float* grad_bias, void* workspace, sm_stream_t stream) {
  return conv_bwd_impl(d, x, nullptr, w_t, w_dgrad, gout, grad_x, nullptr, grad_w_t, grad_bias, workspace, stream);
}
