// 3x3 / stride 1 / pad 1 convolution with a handful of output channels (<= 32) over many input channels (gfx950):
// SipMaskHead's sip_mask_lat (512 -> 32 basis channels, sipmask_head.py:284) and fcos_reg + fcos_centerness
// (256 -> 4 + 1, sipmask_head.py:261-268) -- convs whose GEMM has N = 32: there is no cout tile to amortise the input over.
//
// Why a kernel of its own.  On the patch-resident kernel's 32-cout tile (conv3x3_patch.hip) a block owns 256 consecutive
// positions: at the widest level (168 columns) its patch is 598 rows for 256 outputs (2.3x), it re-streams all 295 KB of
// weights through LDS, and with 90 KB of LDS one block per CU runs a chain of 16 channel slices whose LDS-DMA latency
// nothing hides -- 18 MFMAs of work per wave per slice: 0.081 ms for 19.8 GFLOP and 78 MB (245 TFLOP/s, 1 TB/s).
// Here a block is ONE wave and a tile is 2 rows x 32 columns: the 4 x 34-position patch of a 32-channel slice is 8.7 KB
// (three slices in flight: 27 KB, five blocks per CU, no barrier anywhere: a wave waits on its own vmcnt), and the
// weights never touch LDS -- they are stored in MFMA-fragment order ([K step][lane][8 bf16]: one coalesced 1 KB load per
// K step, L1 / L2 resident, shared by every wave of the chip) and prefetched one slice ahead in registers.
//
// Arithmetic: bf16 operands, v_mfma_f32_32x32x16_bf16, K order = (channel slice, tap, 16-channel half); epilogue
// acc * acc_scale + bias, Scale() on the first scale_nch channels, ReLU / ReLU on those channels only -- the order of
// conv3x3_patch.hip / conv_igemm.hip.  Every output is produced by one wave in a fixed order: results do not depend on
// the batch cut or the launch (plan-to-plan bit equality).
#include <cstdlib>

#include "common.h"

namespace {

constexpr int SC_TR = 2, SC_TC = 32;                     // output rows x cols per tile (= two MFMA N tiles)
constexpr int SC_PW = SC_TC + 2, SC_PH = SC_TR + 2;      // patch: 34 x 4 positions
constexpr int SC_NPOS = SC_PW * SC_PH;                   // 136
constexpr int SC_DMA = (SC_NPOS * 4 + 63) / 64;          // 9 LDS-DMA instructions of 64 x 16 bytes per slice
constexpr int SC_BUF = SC_DMA * 1024;                    // 9 216 bytes per slice buffer
constexpr int SC_RING = 3;

__device__ __attribute__((aligned(16))) const unsigned int g_zero16s[4] = {0u, 0u, 0u, 0u};

struct SmallCoArgs {
  const uint16_t* x;
  const uint16_t* w;      // [cin / 32][9 taps][2][64 lanes][8] bf16 (MFMA A fragments, cout rows padded to 32)
  const float* bias;
  void* y;
  int nlev, batch;
  int h[SM_MAX_LEVELS], w_[SM_MAX_LEVELS];
  long long in_row0[SM_MAX_LEVELS], out_row0[SM_MAX_LEVELS];
  int tile0[SM_MAX_LEVELS + 1];     // first tile of each level
  int tiles_x[SM_MAX_LEVELS], tiles_y[SM_MAX_LEVELS];
  int cin, cout, in_cstride, out_cstride, out_coff;
  unsigned flags;
  int scale_nch;
  float level_scale[SM_MAX_LEVELS];
  float acc_scale;
  int ntiles;
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// F16: the operands are IEEE binary16 (SM_CONV_F16: the x3 head plan's split tensors [hi | lo | hi] x [hi | hi | lo] -- the
// kernel never interprets the 16-bit payloads it moves, only the MFMA differs)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
// PAIRS (F16 only, round 6; sm_conv_desc.x3_pairs): a 32-element slice is 16 channels as [hi 16 | lo 16], so the two fragment
// halves of a tap are (x_hi, x_lo) / (w_hi, w_lo) and the tap issues w_hi*x_hi + w_hi*x_lo + w_lo*x_hi -- two thirds of the
// slices, patch bytes and weight loads of the K-concatenated operand [hi | lo | hi] x [hi | hi | lo] for the same products
template <bool F16, bool PAIRS = false>
__global__ __launch_bounds__(64, 1) void conv3x3_smallco_kernel(const SmallCoArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // SC_RING x SC_BUF
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int lane = threadIdx.x, l31 = lane & 31, khalf = lane >> 5;
  // XCD-contiguous tile order: vertically neighbouring tiles share two of their four patch rows
  const int nblk = (int)gridDim.x, xcd = blockIdx.x & 7, xq = nblk >> 3, xr = nblk & 7;
  const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (int)(blockIdx.x >> 3);
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && tile >= a.tile0[l]) lev = l;
  const int H = a.h[lev], W = a.w_[lev];
  int t = tile - a.tile0[lev];
  const int per_img = a.tiles_x[lev] * a.tiles_y[lev];
  const int b = t / per_img;
  t -= b * per_img;
  const int ty = t / a.tiles_x[lev], tx = t - ty * a.tiles_x[lev];
  const int y0 = ty * SC_TR, x0 = tx * SC_TC;
  const long long img_row0 = a.in_row0[lev] + (long long)b * H * W;

  // ---- per-lane source of every 16-byte slot of a patch slice: slot q = j * 64 + lane holds position p = q >> 2,
  // physical chunk q & 3 = logical chunk (8 channels) ^ ((p >> 2) & 3); positions outside the image read zeros
  // (32-bit element offsets, -1 = outside the image: nine registers instead of nine 64-bit pointers + nine flags -- with them the
  // kernel needed 4 dwords of SCRATCH per lane, the only scratch user of the bf16 plan; round 6)
  int soff[SC_DMA];
#pragma unroll
  for (int j = 0; j < SC_DMA; ++j) {
    const int q = j * 64 + lane;
    const int p = q >> 2;
    const int pr = p / SC_PW, pc = p - pr * SC_PW;
    const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
    const bool live = p < SC_NPOS && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    const int chunk = (q & 3) ^ ((p >> 2) & 3);
    soff[j] = live ? (gy * W + gx) * a.in_cstride + chunk * 8 : -1;
  }
  const uint16_t* const ximg = a.x + img_row0 * a.in_cstride;
  const unsigned long long zero_page = (unsigned long long)g_zero16s;
  auto dma_patch = [&](int sl, int buf) {
    const uint16_t* xb = ximg + sl * 32;
    asm volatile("" : "+s"(xb));                  // opaque: keeps hipcc from hoisting the nine 64-bit addresses out of the slice loop
#pragma unroll
    for (int j = 0; j < SC_DMA; ++j) {
      const unsigned long long pm = soff[j] >= 0 ? ~0ull : 0ull;
      const unsigned long long s = ((unsigned long long)(xb + soff[j]) & pm) | (zero_page & ~pm);
      __builtin_amdgcn_global_load_lds((glb_void*)s, (lds_void*)(smem + buf * SC_BUF + j * 1024), 16, 0, 0);
    }
  };
  // ---- B fragment addresses: N tile tp = output row tp, lane column l31, tap (kh, kw) -> patch position p;
  // 16-channel half s = 0: logical chunk khalf; s = 1: chunk 2 + khalf = the same address ^ 32
  unsigned baddr[SC_TR][9];
#pragma unroll
  for (int tp = 0; tp < SC_TR; ++tp) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int p = (tp + tap / 3) * SC_PW + l31 + (tap % 3);
      baddr[tp][tap] = (unsigned)(p * 64 + ((khalf ^ ((p >> 2) & 3)) << 4));
    }
  }
  const int nsl = a.cin >> 5;
  const bf16x8* wfr = reinterpret_cast<const bf16x8*>(a.w) + lane;
  bf16x8 wa[18], wb[18];
  auto load_w = [&](bf16x8 (&dst)[18], int sl) {
#pragma unroll
    for (int k = 0; k < 18; ++k) dst[k] = wfr[(long long)(sl * 18 + k) * 64];
  };
  f32x16 acc[SC_TR];
#pragma unroll
  for (int tp = 0; tp < SC_TR; ++tp)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[tp][e] = 0.f;

  auto slice = [&](int sl, bf16x8 (&wcur)[18], bf16x8 (&wnxt)[18]) {
    const bool p2 = sl + 2 < nsl, w1 = sl + 1 < nsl;
    if (p2) dma_patch(sl + 2, (sl + 2) % SC_RING);
    if (w1) load_w(wnxt, sl + 1);
    if (p2) wait_vm<SC_DMA + 18>();
    else if (w1) wait_vm<18>();
    else wait_vm<0>();
    const unsigned bb = (unsigned)((sl % SC_RING) * SC_BUF);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if constexpr (PAIRS) {
        bf16x8 xh[SC_TR], xl[SC_TR];
#pragma unroll
        for (int tp = 0; tp < SC_TR; ++tp) {
          xh[tp] = *reinterpret_cast<const bf16x8*>(smem + bb + baddr[tp][tap]);
          xl[tp] = *reinterpret_cast<const bf16x8*>(smem + bb + (baddr[tp][tap] ^ 32u));
        }
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int tp = 0; tp < SC_TR; ++tp)
            acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, wcur[tap * 2 + (term == 2 ? 1 : 0)]),
                                                             __builtin_bit_cast(half8, term == 1 ? xl[tp] : xh[tp]), acc[tp], 0, 0, 0);
        continue;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 xf[SC_TR];
#pragma unroll
        for (int tp = 0; tp < SC_TR; ++tp)
          xf[tp] = *reinterpret_cast<const bf16x8*>(smem + bb + (baddr[tp][tap] ^ (unsigned)(s << 5)));
#pragma unroll
        for (int tp = 0; tp < SC_TR; ++tp)
          if constexpr (F16)
            acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, wcur[tap * 2 + s]),
                                                             __builtin_bit_cast(half8, xf[tp]), acc[tp], 0, 0, 0);
          else
            acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wcur[tap * 2 + s], xf[tp], acc[tp], 0, 0, 0);
      }
    }
  };

  // issue order: patch(0), patch(1), weights(0) | slice 0: patch(2), weights(1) | slice 1: patch(3), weights(2) | ...
  // so that at every slice exactly patch(sl + 2) and weights(sl + 1) may still be in flight
  dma_patch(0, 0);
  if (nsl > 1) dma_patch(1, 1);
  load_w(wa, 0);
  for (int sl = 0; sl < nsl; sl += 2) {
    slice(sl, wa, wb);
    if (sl + 1 < nsl) slice(sl + 1, wb, wa);
  }

  // ---- epilogue (register path of conv3x3_patch.hip): lane (position l31, khalf) ends up with 8 consecutive couts
  const bool out_f32 = (a.flags & SM_CONV_OUT_F32) != 0;
  const float lscale = a.level_scale[lev];
#pragma unroll
  for (int tp = 0; tp < SC_TR; ++tp) {
    const int oy = y0 + tp, ox = x0 + l31;
    const bool pvalid = oy < H && ox < W;
    const long long orow = a.out_row0[lev] + (long long)b * H * W + (long long)oy * W + ox;
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t lo = __float_as_uint(acc[tp][4 * (2 * qp) + e]);
        const uint32_t hi = __float_as_uint(acc[tp][4 * (2 * qp + 1) + e]);
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        v[e] = __uint_as_float(r[0]);
        v[4 + e] = __uint_as_float(r[1]);
      }
      const int c0 = 8 * (2 * qp + khalf);
      if (!pvalid || c0 >= a.cout) continue;
      if (a.acc_scale != 1.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= a.acc_scale;
      }
      if (a.bias != nullptr) {
        const float4 b0 = *reinterpret_cast<const float4*>(a.bias + c0);
        const float4 b1 = *reinterpret_cast<const float4*>(a.bias + c0 + 4);
        v[0] += b0.x, v[1] += b0.y, v[2] += b0.z, v[3] += b0.w;
        v[4] += b1.x, v[5] += b1.y, v[6] += b1.z, v[7] += b1.w;
      }
      if (c0 < a.scale_nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (c0 + e < a.scale_nch) v[e] *= lscale;
      }
      if (a.flags & SM_CONV_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (a.flags & SM_CONV_RELU_NCH) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (c0 + e < a.scale_nch) v[e] = fmaxf(v[e], 0.f);
      }
      const long long o = orow * a.out_cstride + a.out_coff + c0;
      if (out_f32) {
        float* yp = reinterpret_cast<float*>(a.y) + o;
        *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(a.y) + o) = pack_bf16x8_v(v);
      }
    }
  }
}

bool smallco_ok(const sm_conv_desc* d) {
  if (!d || d->nlev < 1 || d->nlev > SM_MAX_LEVELS || d->batch < 1) return false;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->dil > 1) return false;
  if (d->cout < 8 || d->cout > 32 || d->cout % 8 != 0 || d->cin < 32 || d->cin % 32 != 0) return false;
  if (d->in_cstride < d->cin || d->in_cstride % 8 != 0) return false;
  // residual / input ReLU / split-precision outputs: not here (launch-plan selector bits are ignored); binary16 operands
  // (SM_CONV_F16) with f32 output only
  if (d->flags & (SM_CONV_RES_ADD | SM_CONV_RES_NEAREST | SM_CONV_IN_RELU | SM_CONV_OUT_X3)) return false;
  if ((d->flags & SM_CONV_F16) && !(d->flags & SM_CONV_OUT_F32)) return false;
  if (d->ngroups > 1 || d->w_batch_stride != 0 || d->w_level_stride != 0 || d->deform_groups > 0) return false;
  if (d->x3_pairs != 0 && (d->x3_pairs != 1 || !(d->flags & SM_CONV_F16))) return false;     // paired operands: binary16 only
  const int calign = (d->flags & SM_CONV_OUT_F32) ? 4 : 8;        // 16-byte stores
  if (d->out_cstride % calign != 0 || d->out_coff % calign != 0 || d->out_coff + d->cout > d->out_cstride) return false;
  for (int l = 0; l < d->nlev; ++l) {
    if (d->in_h[l] != d->out_h[l] || d->in_w[l] != d->out_w[l] || d->in_h[l] < 1 || d->in_w[l] < 1) return false;
    if ((long long)d->in_h[l] * d->in_w[l] * d->in_cstride >= (1ll << 31)) return false;      // 32-bit patch offsets per image
  }
  return true;
}

}  // namespace

extern "C" int sm_conv3x3_smallco_supported(const sm_conv_desc* d) { return smallco_ok(d) ? 1 : 0; }

extern "C" int sm_conv3x3_smallco(const sm_conv_desc* d, const void* x, const void* w_frag, const float* bias, void* y,
                                  sm_stream_t stream) {
  if (!x || !w_frag || !y) return SM_ERR_BAD_ARG;
  if (!smallco_ok(d)) return SM_ERR_UNSUPPORTED;
  SmallCoArgs a;
  a.x = (const uint16_t*)x;
  a.w = (const uint16_t*)w_frag;
  a.bias = bias;
  a.y = y;
  a.nlev = d->nlev;
  a.batch = d->batch;
  long long nt = 0;
  for (int l = 0; l < SM_MAX_LEVELS; ++l) {
    const bool on = l < d->nlev;
    a.h[l] = on ? d->in_h[l] : 1;
    a.w_[l] = on ? d->in_w[l] : 1;
    a.in_row0[l] = on ? d->in_row0[l] : 0;
    a.out_row0[l] = on ? d->out_row0[l] : 0;
    a.level_scale[l] = on ? d->level_scale[l] : 1.f;
    a.tiles_x[l] = on ? (d->in_w[l] + SC_TC - 1) / SC_TC : 0;
    a.tiles_y[l] = on ? (d->in_h[l] + SC_TR - 1) / SC_TR : 0;
    a.tile0[l] = (int)nt;
    nt += (long long)d->batch * a.tiles_x[l] * a.tiles_y[l];
  }
  a.tile0[SM_MAX_LEVELS] = (int)nt;
  if (nt < 1 || nt > 0x7fffffffll) return SM_ERR_BAD_SHAPE;
  a.ntiles = (int)nt;
  a.cin = d->cin;
  a.cout = d->cout;
  a.in_cstride = d->in_cstride;
  a.out_cstride = d->out_cstride;
  a.out_coff = d->out_coff;
  a.flags = d->flags;
  a.scale_nch = d->scale_nch;
  a.acc_scale = (d->acc_scale == 0.f) ? 1.f : d->acc_scale;
  if ((d->flags & SM_CONV_F16) && d->x3_pairs)
    hipLaunchKernelGGL((conv3x3_smallco_kernel<true, true>), dim3((unsigned)nt), dim3(64), SC_RING * SC_BUF, sm_hip_stream(stream), a);
  else if (d->flags & SM_CONV_F16)
    hipLaunchKernelGGL(conv3x3_smallco_kernel<true>, dim3((unsigned)nt), dim3(64), SC_RING * SC_BUF, sm_hip_stream(stream), a);
  else
    hipLaunchKernelGGL(conv3x3_smallco_kernel<false>, dim3((unsigned)nt), dim3(64), SC_RING * SC_BUF, sm_hip_stream(stream), a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
