// Exact-f32 implicit-GEMM convolution for gfx950: the PARITY MODE of the hot path.
//
// The reference head / backbone are fp32 end to end (M/mmdet/models/anchor_heads/sipmask_head.py:241-287,
// resnet.py:206-229, fpn.py:141-175); north_star asks for mask logits within 1e-3 of it.  bf16 operands cannot
// meet that through ~60 stacked convs, so this kernel runs the same implicit GEMM on v_mfma_f32_32x32x2_f32:
// f32 operands, f32 accumulate, bit-for-bit a k-ordered fmaf chain (guide section 3: "exact f32 at the vector
// rate", 157 TFLOP/s peak = 1/16 of the bf16 MFMA rate).  It is selected per launch plan
// (SipMaskEngine(precision="f32")); the bf16 kernels of conv_igemm.hip remain the throughput path.
//
// GEMM view as in conv_igemm.hip: D[cout][pos] = sum_k W[cout][k] * X[pos][k], k = (kh, kw, cin), cin fastest.
// A = weight tile, B = activation tile  =>  a lane owns 4 consecutive couts of one position (float4 stores).
// K step = 16 floats: LDS rows of 64 bytes, 16-byte slots XOR-swizzled (slot = chunk ^ ((row>>2)&3)) so the
// ds_read_b128 fragment reads are conflict free.  The 32x32x2 instruction takes A[i][k = lane>>5]: a lane of
// half h reads the 8 CONSECUTIVE floats [8h, 8h+8) of its row and feeds element e to instruction e -- the k order
// inside a K step is a free permutation as long as A and B use the same one.
// Loader: global -> VGPR -> LDS (VALU may touch the operand: zero padding, input ReLU, and for the DEFORM variant
// the bilinear gather of deform_conv_cuda_kernel.cu:85-115,216-229 in f32, exactly as the reference samples).
//
// X3 variant (SM_CONV_F16 on sm_conv2d_f32; round 3): the same kernel -- f32 tensors in HBM, the same loader -- with the
// contraction on the f16 matrix pipe in split precision: the loader turns every f32 operand element (the blended
// deformable sample included) into two binary16 halves, hi = f16(v), lo = f16(v - hi), an LDS row holds
// [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] (the same 64 bytes as 16 floats), and a K step is three
// v_mfma_f32_32x32x16_f16 per tile pair: w_hi*x_hi + w_hi*x_lo + w_lo*x_hi (~2^-21 per product, f32 accumulation) instead
// of eight v_mfma_f32_32x32x2_f32 -- 5.3x less matrix-pipe time.  This is FeatureAlign's deformable conv in the x3 head
// plan: the one operand VALU has to produce cannot come through LDS-DMA, so it cannot use the K-concatenated layout of the
// other convs (split_x3.hip).  Weights arrive multiplied by a power of two (binary16's subnormals), undone by acc_scale.
#include "common.h"

namespace {

struct ConvFArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* res;
  float* y;
  const float* offset;
  int nlev, batch;
  int in_h[SM_MAX_LEVELS], in_w[SM_MAX_LEVELS], out_h[SM_MAX_LEVELS], out_w[SM_MAX_LEVELS];
  long long in_row0[SM_MAX_LEVELS], out_row0[SM_MAX_LEVELS], res_row0[SM_MAX_LEVELS];
  int res_h[SM_MAX_LEVELS], res_w[SM_MAX_LEVELS];
  int tile0[SM_MAX_LEVELS + 1];
  int cin, cout, kh, kw, stride, pad, dil;
  int in_cstride, out_cstride, out_coff, res_cstride;
  int Kp, K, ntn, nk;
  unsigned flags;
  int scale_nch;
  float level_scale[SM_MAX_LEVELS];
  int dg, cpg;  // deform groups, channels per deform group
  float acc_scale;  // X3 only: accumulators x this before bias (1 = none)
};

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// 4 f32 -> (hi, lo) binary16 quadruples; |v| beyond binary16's range saturates
__device__ __forceinline__ void split4(const f32x4 v, half4& hi, half4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float c = fminf(fmaxf(v[e], -65504.f), 65504.f);
    const _Float16 h = (_Float16)c;
    hi[e] = h;
    lo[e] = (_Float16)(c - (float)h);
  }
}

template <int WCO, int WPOS, int TCO, int TPOS, bool DEFORM, bool X3 = false>
__global__ __launch_bounds__(256) void conv_f32_kernel(const ConvFArgs a) {
  constexpr int BCO = WCO * TCO * 32;
  constexpr int BPOS = WPOS * TPOS * 32;
  constexpr int NW = (BCO + 63) / 64;   // 16-byte weight chunks per thread per K step
  constexpr int NX = BPOS / 64;         // 16-byte activation chunks per thread per K step
  constexpr int STAGE = (BCO + BPOS) * 64;
  static_assert(WCO * WPOS == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wco = wave / WPOS;
  const int wpos = wave % WPOS;
  const int cj = tid & 3;     // logical 16-byte chunk (4 floats) of the K step this thread loads
  const int r0 = tid >> 2;    // tile row (+64*i)

  // ---- tile decode (wave-uniform), levels enumerated as in conv_igemm.hip
  const int tlin = (int)blockIdx.x;
  const int nt = tlin % a.ntn;
  const int mt = tlin / a.ntn;
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && mt >= a.tile0[l]) lev = l;
  const int H = a.in_h[lev], W = a.in_w[lev], Ho = a.out_h[lev], Wo = a.out_w[lev];
  const int HoWo = Ho * Wo;
  const int M = a.batch * HoWo;
  const int m0 = (mt - a.tile0[lev]) * BPOS;
  const long long in_row0 = a.in_row0[lev];

  int rn[NX], rhi[NX], rwi[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int m = m0 + r0 + 64 * i;
    if (m < M) {
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int ho = rem / Wo;
      rn[i] = n;
      rhi[i] = ho * a.stride - a.pad;
      rwi[i] = (rem - ho * Wo) * a.stride - a.pad;
    } else {
      rn[i] = -1;
      rhi[i] = 0;
      rwi[i] = 0;
    }
  }
  const float* wrow = a.w + (long long)(nt * BCO + r0) * a.Kp + cj * 4;
  const bool in_relu = a.flags & SM_CONV_IN_RELU;
  const int ntap = a.kh * a.kw;

  f32x4 wreg[NW], xreg[NX];
  // deformable variant: sampling geometry of the current (tap, deformable group) per tile row (see load_tile)
  float geo_w[DEFORM ? NX : 1][4];
  int geo_o[DEFORM ? NX : 1][4];
  int geo_tg[DEFORM ? NX : 1];
#pragma unroll
  for (int i = 0; i < (DEFORM ? NX : 1); ++i) geo_tg[i] = -1;
  // a K step's 16 channels lie inside one tap and one deformable group for every thread of the block
  const bool geo_hoist = DEFORM && (a.cin % 16 == 0) && (a.cpg % 16 == 0);
  auto load_tile = [&](int kt) {
    const int k0 = kt * 16 + cj * 4;                 // first K element of this thread's chunk
    const bool kvalid = k0 < a.K;
    const int tap = kvalid ? k0 / a.cin : 0;
    const int ci = k0 - tap * a.cin;
    const int tkh = tap / a.kw;
    const int tkw = tap - tkh * a.kw;
    const int dh = tkh * a.dil, dw = tkw * a.dil;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      if (BCO >= 64 || r0 < BCO)
        wreg[i] = *reinterpret_cast<const f32x4*>(wrow + (long long)i * 64 * a.Kp + kt * 16);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const bool rvalid = kvalid && rn[i] >= 0;
      if constexpr (!DEFORM) {
        const int hi = rhi[i] + dh, wi = rwi[i] + dw;
        const bool ok = rvalid && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        const long long eo = ok ? (in_row0 + ((long long)rn[i] * H + hi) * W + wi) * a.in_cstride + ci : 0ll;
        const f32x4 q = *reinterpret_cast<const f32x4*>(a.x + eo);   // unconditional load, selected below
        if (ok) v = q;
      } else {
        // deformable_im2col_gpu_kernel (deform_conv_cuda_kernel.cu:191-243) for one (position, tap, 4 channels):
        // offset channel layout [g][2*tap + {0: h, 1: w}] (:216,222-223), stride-1 "same" conv => offset row = out row.
        // The sampling geometry -- offset, bilinear weights, corner addresses -- belongs to a (position, tap, deformable
        // group): it is the same for every K step inside the group's channels (4 steps of 16 at 64 channels per group), so
        // it is computed when (tap, group) changes and kept in registers; a K step then is four independent corner loads
        // and the blend instead of offset load -> address -> corner loads (two dependent memory latencies).
        const int g = ci / a.cpg;
        const int tg = kvalid ? tap * a.dg + g : -2;
        if (!geo_hoist || tg != geo_tg[i]) {
          geo_tg[i] = tg;
          const long long orow = a.out_row0[lev] + m0 + r0 + 64 * i;
          const long long oo = rvalid ? orow * (long long)(a.dg * ntap * 2) + (g * ntap + tap) * 2 : 0ll;
          const float2 off = *reinterpret_cast<const float2*>(a.offset + oo);
          const float h_im = (float)(rhi[i] + dh) + off.x;
          const float w_im = (float)(rwi[i] + dw) + off.y;
          const bool inr = rvalid && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;   // :229
          const float hf = floorf(h_im), wf = floorf(w_im);
          const int h_low = inr ? (int)hf : 0, w_low = inr ? (int)wf : 0;
          const int h_high = h_low + 1, w_high = w_low + 1;
          const float lh = h_im - hf, lw = w_im - wf;
          const float hh = 1.f - lh, hw = 1.f - lw;
          const bool t_ok = h_low >= 0, b_ok = h_high <= H - 1, l_ok = w_low >= 0, r_ok = w_high <= W - 1;   // :98-109
          const int hl = min(max(h_low, 0), H - 1), hh_ = min(max(h_high, 0), H - 1);
          const int wl = min(max(w_low, 0), W - 1), wh_ = min(max(w_high, 0), W - 1);
          // a corner outside the image contributes 0 (:98-109), a sample outside (-1, H) x (-1, W) is 0 (:229): zero weights
          geo_w[i][0] = (inr && t_ok && l_ok) ? hh * hw : 0.f;                                           // :111-112
          geo_w[i][1] = (inr && t_ok && r_ok) ? hh * lw : 0.f;
          geo_w[i][2] = (inr && b_ok && l_ok) ? lh * hw : 0.f;
          geo_w[i][3] = (inr && b_ok && r_ok) ? lh * lw : 0.f;
          const int nrow = rvalid ? rn[i] * H * W : 0;
          geo_o[i][0] = nrow + hl * W + wl;
          geo_o[i][1] = nrow + hl * W + wh_;
          geo_o[i][2] = nrow + hh_ * W + wl;
          geo_o[i][3] = nrow + hh_ * W + wh_;
        }
        const float* base = a.x + in_row0 * a.in_cstride + (rvalid ? ci : 0);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(base + (long long)geo_o[i][0] * a.in_cstride);
        const f32x4 q2 = *reinterpret_cast<const f32x4*>(base + (long long)geo_o[i][1] * a.in_cstride);
        const f32x4 q3 = *reinterpret_cast<const f32x4*>(base + (long long)geo_o[i][2] * a.in_cstride);
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(base + (long long)geo_o[i][3] * a.in_cstride);
        v = geo_w[i][0] * q1 + geo_w[i][1] * q2 + geo_w[i][2] * q3 + geo_w[i][3] * q4;                   // :113
      }
      if (in_relu) {
        v.x = fmaxf(v.x, 0.f);
        v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f);
        v.w = fmaxf(v.w, 0.f);
      }
      xreg[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* Wb = smem + buf * STAGE;
    unsigned char* Xb = Wb + BCO * 64;
    if constexpr (X3) {
      // row = [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15]; this thread's 4 floats are k = 4cj .. 4cj+3
      const int hslot = cj >> 1, sub = (cj & 1) * 8;
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const int r = r0 + 64 * i;
        if (BCO >= 64 || r0 < BCO) {
          half4 hi, lo;
          split4(wreg[i], hi, lo);
          const int sw = (r >> 2) & 3;
          *reinterpret_cast<half4*>(Wb + r * 64 + ((hslot ^ sw) * 16) + sub) = hi;
          *reinterpret_cast<half4*>(Wb + r * 64 + (((2 + hslot) ^ sw) * 16) + sub) = lo;
        }
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int r = r0 + 64 * i;
        half4 hi, lo;
        split4(xreg[i], hi, lo);
        const int sw = (r >> 2) & 3;
        *reinterpret_cast<half4*>(Xb + r * 64 + ((hslot ^ sw) * 16) + sub) = hi;
        *reinterpret_cast<half4*>(Xb + r * 64 + (((2 + hslot) ^ sw) * 16) + sub) = lo;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int r = r0 + 64 * i;
      if (BCO >= 64 || r0 < BCO) *reinterpret_cast<f32x4*>(Wb + r * 64 + ((cj ^ ((r >> 2) & 3)) * 16)) = wreg[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int r = r0 + 64 * i;
      *reinterpret_cast<f32x4*>(Xb + r * 64 + ((cj ^ ((r >> 2) & 3)) * 16)) = xreg[i];
    }
  };

  f32x16 acc[TCO][TPOS];
#pragma unroll
  for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
    for (int tp = 0; tp < TPOS; ++tp)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tc][tp][e] = 0.f;

  const int l31 = lane & 31;
  const int khalf = lane >> 5;
  const int rsw = (l31 >> 2) & 3;
  const int wrow_off = (wco * TCO * 32 + l31) * 64;
  const int xrow_off = BCO * 64 + (wpos * TPOS * 32 + l31) * 64;
  auto compute = [&](int buf) {
    const unsigned char* S = smem + buf * STAGE;
    if constexpr (X3) {
      // MFMA 32x32x16: a lane of half h supplies k = 8h .. 8h+7 of its row: slot h (hi) / 2 + h (lo)
      const int sh = (khalf ^ rsw) * 16, sl = ((2 + khalf) ^ rsw) * 16;
      half8 wh[TCO], wl[TCO], xh[TPOS], xl[TPOS];
#pragma unroll
      for (int t = 0; t < TCO; ++t) {
        wh[t] = *reinterpret_cast<const half8*>(S + wrow_off + t * 32 * 64 + sh);
        wl[t] = *reinterpret_cast<const half8*>(S + wrow_off + t * 32 * 64 + sl);
      }
#pragma unroll
      for (int t = 0; t < TPOS; ++t) {
        xh[t] = *reinterpret_cast<const half8*>(S + xrow_off + t * 32 * 64 + sh);
        xl[t] = *reinterpret_cast<const half8*>(S + xrow_off + t * 32 * 64 + sl);
      }
#pragma unroll
      for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
        for (int tp = 0; tp < TPOS; ++tp) {
          acc[tc][tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[tc], xh[tp], acc[tc][tp], 0, 0, 0);
          acc[tc][tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[tc], xl[tp], acc[tc][tp], 0, 0, 0);
          acc[tc][tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[tc], xh[tp], acc[tc][tp], 0, 0, 0);
        }
      return;
    }
    const int s0 = ((2 * khalf) ^ rsw) * 16, s1 = ((2 * khalf + 1) ^ rsw) * 16;
    f32x4 wf[TCO][2], xf[TPOS][2];
#pragma unroll
    for (int t = 0; t < TCO; ++t) {
      wf[t][0] = *reinterpret_cast<const f32x4*>(S + wrow_off + t * 32 * 64 + s0);
      wf[t][1] = *reinterpret_cast<const f32x4*>(S + wrow_off + t * 32 * 64 + s1);
    }
#pragma unroll
    for (int t = 0; t < TPOS; ++t) {
      xf[t][0] = *reinterpret_cast<const f32x4*>(S + xrow_off + t * 32 * 64 + s0);
      xf[t][1] = *reinterpret_cast<const f32x4*>(S + xrow_off + t * 32 * 64 + s1);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
        for (int tp = 0; tp < TPOS; ++tp)
          acc[tc][tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[tc][e >> 2][e & 3], xf[tp][e >> 2][e & 3], acc[tc][tp], 0, 0, 0);
  };

  const int nk = a.nk;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) load_tile(kt + 1);
    compute(buf);
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane owns couts cl..cl+3 (4 per accumulator quad) of position l31 of each 32x32 tile
  const float lscale = a.level_scale[lev];
  const long long out_row0 = a.out_row0[lev];
  const bool has_res = a.flags & (SM_CONV_RES_ADD | SM_CONV_RES_NEAREST);
  const bool vec_out = (a.out_cstride & 3) == 0 && (a.out_coff & 3) == 0;
  const bool vec_res = (a.res_cstride & 3) == 0;
#pragma unroll
  for (int tp = 0; tp < TPOS; ++tp) {
    const int m = m0 + wpos * TPOS * 32 + tp * 32 + l31;
    if (m >= M) continue;
    long long rrow = 0;
    if (has_res) {
      if (a.flags & SM_CONV_RES_ADD) {
        rrow = out_row0 + m;
      } else {   // FPN top-down nearest source (fpn.py:149-152; same index rule as conv_igemm.hip)
        const int n = m / HoWo;
        const int rem = m - n * HoWo;
        const int ho = rem / Wo;
        const int wo = rem - ho * Wo;
        const int rh = a.res_h[lev], rw = a.res_w[lev];
        const int sh = min((int)floorf((float)ho * ((float)rh / (float)Ho)), rh - 1);
        const int sw = min((int)floorf((float)wo * ((float)rw / (float)Wo)), rw - 1);
        rrow = a.res_row0[lev] + ((long long)n * rh + sh) * rw + sw;
      }
    }
#pragma unroll
    for (int tc = 0; tc < TCO; ++tc) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = nt * BCO + wco * TCO * 32 + tc * 32 + 8 * q + 4 * khalf;
        if (c0 >= a.cout) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[tc][tp][4 * q + e];
          if constexpr (X3) v[e] *= a.acc_scale;
          if (c0 + e < a.cout) {
            if (a.bias != nullptr) v[e] += a.bias[c0 + e];
            if (c0 + e < a.scale_nch) v[e] *= lscale;
          }
        }
        const bool full = c0 + 3 < a.cout;
        if (has_res) {
          const float* rp = a.res + rrow * a.res_cstride + c0;
          if (full && vec_res) {
            const float4 r = *reinterpret_cast<const float4*>(rp);
            v[0] += r.x, v[1] += r.y, v[2] += r.z, v[3] += r.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c0 + e < a.cout) v[e] += rp[e];
          }
        }
        if (a.flags & SM_CONV_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (a.flags & SM_CONV_RELU_NCH) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e < a.scale_nch) v[e] = fmaxf(v[e], 0.f);
        }
        float* yp = a.y + (out_row0 + m) * a.out_cstride + a.out_coff + c0;
        if (full && vec_out) {
          *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e < a.cout) yp[e] = v[e];
        }
      }
    }
  }
}

template <bool DEFORM>
int launch_conv_f32(const sm_conv_desc* d, const float* x, const float* offset, const float* w, const float* bias,
                    const float* residual, float* y, hipStream_t stream) {
  if (!d || !x || !w || !y) return SM_ERR_BAD_ARG;
  if (DEFORM && !offset) return SM_ERR_BAD_ARG;
  if (d->nlev < 1 || d->nlev > SM_MAX_LEVELS || d->batch < 1) return SM_ERR_BAD_SHAPE;
  if (d->cin % 4 != 0 || d->cin < 4 || d->cout < 1 || d->in_cstride % 4 != 0) return SM_ERR_BAD_SHAPE;
  const int tile = sm_conv_cout_tile(d->cout);
  if (d->cout_pad % tile != 0 || d->cout_pad < d->cout) return SM_ERR_BAD_SHAPE;
  if ((d->flags & (SM_CONV_RES_ADD | SM_CONV_RES_NEAREST)) && !residual) return SM_ERR_BAD_ARG;
  if (DEFORM) {
    if (d->deform_groups < 1 || d->cin % (4 * d->deform_groups) != 0) return SM_ERR_BAD_ARG;
  }
  for (int l = 0; l < d->nlev; ++l) {
    if (d->out_h[l] < 1 || d->out_w[l] < 1 || d->in_h[l] < 1 || d->in_w[l] < 1) return SM_ERR_BAD_SHAPE;
    const int eh = (d->in_h[l] + 2 * d->pad - (d->dil * (d->kh - 1) + 1)) / d->stride + 1;
    const int ew = (d->in_w[l] + 2 * d->pad - (d->dil * (d->kw - 1) + 1)) / d->stride + 1;
    if (eh != d->out_h[l] || ew != d->out_w[l]) return SM_ERR_BAD_SHAPE;
  }
  ConvFArgs a;
  a.x = x;
  a.w = w;
  a.bias = bias;
  a.res = residual;
  a.y = y;
  a.offset = offset;
  a.nlev = d->nlev;
  a.batch = d->batch;
  a.K = d->kh * d->kw * d->cin;
  a.Kp = (a.K + 15) / 16 * 16;
  constexpr int BPOS = 128;
  int t = 0;
  for (int l = 0; l < SM_MAX_LEVELS; ++l) {
    const bool on = l < d->nlev;
    a.in_h[l] = on ? d->in_h[l] : 1;
    a.in_w[l] = on ? d->in_w[l] : 1;
    a.out_h[l] = on ? d->out_h[l] : 1;
    a.out_w[l] = on ? d->out_w[l] : 1;
    a.in_row0[l] = on ? d->in_row0[l] : 0;
    a.out_row0[l] = on ? d->out_row0[l] : 0;
    a.res_row0[l] = on ? d->res_row0[l] : 0;
    a.res_h[l] = on ? d->res_h[l] : 1;
    a.res_w[l] = on ? d->res_w[l] : 1;
    a.level_scale[l] = on ? d->level_scale[l] : 1.f;
    a.tile0[l] = t;
    if (on) t += sm_cdiv((long long)d->batch * d->out_h[l] * d->out_w[l], BPOS);
  }
  a.tile0[SM_MAX_LEVELS] = t;
  a.cin = d->cin;
  a.cout = d->cout;
  a.kh = d->kh;
  a.kw = d->kw;
  a.stride = d->stride;
  a.pad = d->pad;
  a.dil = d->dil;
  a.in_cstride = d->in_cstride;
  a.out_cstride = d->out_cstride;
  a.out_coff = d->out_coff;
  a.res_cstride = d->res_cstride;
  a.ntn = d->cout_pad / tile;
  a.nk = a.Kp / 16;
  a.flags = d->flags;
  a.scale_nch = d->scale_nch;
  a.dg = DEFORM ? d->deform_groups : 1;
  a.cpg = DEFORM ? d->cin / d->deform_groups : d->cin;
  a.acc_scale = (d->acc_scale == 0.f) ? 1.f : d->acc_scale;
  const long long nblk = (long long)t * a.ntn;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return SM_ERR_BAD_SHAPE;
  dim3 grid((unsigned)nblk), block(256);
  if (d->flags & SM_CONV_F16) {            // split-precision contraction of the f32 operands (X3 above): 128-cout tiles
    if (tile != 128) return SM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv_f32_kernel<2, 2, 2, 2, DEFORM, true>), grid, block, 0, stream, a);
    SM_LAUNCH_CHECK();
    return SM_OK;
  }
  if (tile == 128) hipLaunchKernelGGL((conv_f32_kernel<2, 2, 2, 2, DEFORM>), grid, block, 0, stream, a);
  else if (tile == 64) hipLaunchKernelGGL((conv_f32_kernel<1, 4, 2, 1, DEFORM>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((conv_f32_kernel<1, 4, 1, 1, DEFORM>), grid, block, 0, stream, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

}  // namespace

extern "C" int sm_conv2d_f32(const sm_conv_desc* d, const float* x, const float* offset, const float* w,
                             const float* bias, const float* residual, float* y, sm_stream_t stream) {
  if (offset != nullptr) return launch_conv_f32<true>(d, x, offset, w, bias, residual, y, sm_hip_stream(stream));
  return launch_conv_f32<false>(d, x, nullptr, w, bias, residual, y, sm_hip_stream(stream));
}
