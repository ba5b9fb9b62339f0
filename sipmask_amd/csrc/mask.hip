// Fused SipMask mask assembly (sipmask_head.py:609-633) -- HBM bound:
//   algorithmic bytes / image = basis read (Hm*Wm*32*4) + u8 masks written (N*Ho*Wo).
// One block owns one TOW x TOH tile of output pixels for ALL detections of an image:
//   * the source (mask-resolution) pixels that tile needs (<= 512) are owned by threads, whose
//     32 basis values stay in VGPRs for the whole detection loop -> the basis is read once;
//   * the detections' boxes and 4x32 coefficients are staged ONCE in LDS (no dependent global
//     load inside the detection loop);
//   * per detection: quadrant select (CropSplit index math, crop_split_cuda_kernel.cu:34-52)
//     -> ONE 32-long dot product (only the selected quadrant is ever needed) -> sigmoid
//     -> LDS tile -> bilinear (align_corners=False) -> > thr -> packed u8 stores; TOW = 128
//     makes every store instruction cover whole 128-byte lines.
// The reference materialises 4 x [Hm*Wm, N] sigmoid planes, the stacked copy, the crop, the
// upsampled float masks and the thresholded copy: ~4 GB/image of traffic vs ~140 MB here.
#include "common.h"

namespace {

constexpr int MA_THREADS = 256;
constexpr int MA_PX = 2;  // source pixels owned per thread
constexpr int MA_SRC_CAP = MA_THREADS * MA_PX;
constexpr int MA_DETS = 64;  // detections staged in LDS per pass (64 * 128 * 4 B = 32 KiB)

struct MaskArgs {
  const float* basis;
  const float* cofs;
  const int64_t* keep;
  const float* det;
  const int32_t* ndet;
  uint8_t* masks;
  float* pos_masks;
  int kmax, max_num, hm, wm, ho, wo, pitch;   // wo = logical width, pitch = row pitch of masks (multiple of 4)
  long long pix_stride, ch_stride;  // basis strides (elements)
  float box_mul_x, box_mul_y, box_div, inv_up_x, inv_up_y, thr;
  const float* per_image;   // [batch][8] = (box_mul_x, box_mul_y, up_h, up_w, Ho, Wo, 1/up_h, 1/up_w) or nullptr
};

struct DetBox {
  float x1, y1, x2, y2, rw, rh;
};

template <int TOW, int TOH>
__global__ __launch_bounds__(MA_THREADS) void mask_assemble_kernel(const MaskArgs a) {
  __shared__ __attribute__((aligned(16))) float s_cof[MA_DETS * 128];
  __shared__ DetBox s_box[MA_DETS];
  __shared__ float s_prob[MA_SRC_CAP];
  const int b = blockIdx.z;
  const int ox0 = blockIdx.x * TOW, oy0 = blockIdx.y * TOH;
  const int tid = threadIdx.x;
  const int nd = min(a.ndet[b], a.max_num);
  if (nd <= 0) return;
  // this image's crop / upsample geometry (img_metas[img_id]['scale_factor'], sipmask_head.py:517-541,621-633); a.ho /
  // a.pitch stay the canvas every mask plane is allocated with
  float g_mul_x = a.box_mul_x, g_mul_y = a.box_mul_y, g_inv_up_x = a.inv_up_x, g_inv_up_y = a.inv_up_y;
  int g_ho = a.ho, g_wo = a.wo;
  if (a.per_image != nullptr) {
    const float* g = a.per_image + (long long)b * 8;
    g_mul_x = g[0], g_mul_y = g[1], g_inv_up_y = g[6], g_inv_up_x = g[7];
    g_ho = min((int)g[4], a.ho), g_wo = min((int)g[5], a.wo);
  }
  if (ox0 >= g_wo || oy0 >= g_ho) return;          // (block-uniform) tile outside this image's mask

  // source window of this output tile
  auto src_x = [&](int o) { return fmaxf(g_inv_up_x * ((float)o + 0.5f) - 0.5f, 0.f); };
  auto src_y = [&](int o) { return fmaxf(g_inv_up_y * ((float)o + 0.5f) - 0.5f, 0.f); };
  const int oxe = min(ox0 + TOW, g_wo) - 1, oye = min(oy0 + TOH, g_ho) - 1;
  const int sx0 = (int)src_x(ox0), sy0 = (int)src_y(oy0);
  const int sx1 = min((int)src_x(oxe) + 1, a.wm - 1), sy1 = min((int)src_y(oye) + 1, a.hm - 1);
  const int spw = sx1 - sx0 + 1, sph = sy1 - sy0 + 1;
  const int nsrc = spw * sph;  // host guarantees <= MA_SRC_CAP

  // owned source pixels + their basis vectors (registers)
  float bas[MA_PX][32];
  int gx[MA_PX], gy[MA_PX];
#pragma unroll
  for (int p = 0; p < MA_PX; ++p) {
    const int li = tid + p * MA_THREADS;
    gx[p] = -1;
    gy[p] = -1;
    if (li < nsrc) {
      const int ly = li / spw, lx = li - ly * spw;
      gx[p] = sx0 + lx;
      gy[p] = sy0 + ly;
      const float* bp = a.basis + (long long)b * a.hm * a.wm * 32 + ((long long)gy[p] * a.wm + gx[p]) * a.pix_stride;
      if (a.ch_stride == 1) {
#pragma unroll
        for (int k = 0; k < 32; k += 4) {
          const float4 v = *reinterpret_cast<const float4*>(bp + k);
          bas[p][k] = v.x;
          bas[p][k + 1] = v.y;
          bas[p][k + 2] = v.z;
          bas[p][k + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) bas[p][k] = bp[(long long)k * a.ch_stride];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 32; ++k) bas[p][k] = 0.f;
    }
  }

  constexpr int GROUPS = TOW * TOH / 4;  // 4-pixel (one u32) groups in the tile
  constexpr int GPT = (GROUPS + MA_THREADS - 1) / MA_THREADS;

  for (int n0 = 0; n0 < nd; n0 += MA_DETS) {
    const int nn = min(MA_DETS, nd - n0);
    __syncthreads();  // previous pass finished with s_cof / s_box / s_prob
    // stage this pass's detections: boxes (CropSplit geometry) + coefficients, coalesced
    if (tid < nn) {
      const float* d = a.det + ((long long)b * a.max_num + n0 + tid) * 5;
      DetBox bx;
      bx.x1 = __fdiv_rn(__fmul_rn(d[0], g_mul_x), a.box_div);
      bx.y1 = __fdiv_rn(__fmul_rn(d[1], g_mul_y), a.box_div);
      bx.x2 = __fdiv_rn(__fmul_rn(d[2], g_mul_x), a.box_div);
      bx.y2 = __fdiv_rn(__fmul_rn(d[3], g_mul_y), a.box_div);
      // roi_w = (x2 - x1 + 0.1) / num_cell in double, rounded to float (kernel.cu:47-48)
      bx.rw = (float)(((double)__fsub_rn(bx.x2, bx.x1) + 0.1) / 2.0);
      bx.rh = (float)(((double)__fsub_rn(bx.y2, bx.y1) + 0.1) / 2.0);
      s_box[tid] = bx;
    }
    for (int i = tid; i < nn * 32; i += MA_THREADS) {  // 32 float4 per detection
      const int dd = i >> 5, q4 = i & 31;
      const long long src = ((long long)b * a.kmax + a.keep[(long long)b * a.max_num + n0 + dd]) * 128;
      *reinterpret_cast<float4*>(s_cof + dd * 128 + q4 * 4) = *reinterpret_cast<const float4*>(a.cofs + src + q4 * 4);
    }
    __syncthreads();

    for (int n = 0; n < nn; ++n) {
      const long long dn = (long long)b * a.max_num + n0 + n;
      const DetBox bx = s_box[n];
      // tile / box overlap in source coordinates (block-uniform, conservative)
      const bool hit = ((float)sx1 >= bx.x1) && ((float)sx0 < bx.x2) && ((float)sy1 >= bx.y1) && ((float)sy0 < bx.y2);
      uint8_t* mrow = a.masks + dn * (long long)a.ho * a.pitch;
      if (!hit) {
        // the whole tile is zero: no LDS traffic, no barrier
#pragma unroll
        for (int g = 0; g < GPT; ++g) {
          const int gi = tid + g * MA_THREADS;
          if (gi < GROUPS) {
            const int oy = oy0 + gi / (TOW / 4), ox = ox0 + (gi % (TOW / 4)) * 4;
            if (oy < g_ho && ox < g_wo) *reinterpret_cast<uint32_t*>(mrow + (long long)oy * a.pitch + ox) = 0u;
          }
        }
        if (a.pos_masks) {
#pragma unroll
          for (int p = 0; p < MA_PX; ++p)
            if (gx[p] >= 0) a.pos_masks[dn * (long long)a.hm * a.wm + (long long)gy[p] * a.wm + gx[p]] = 0.f;
        }
        continue;
      }
#pragma unroll
      for (int p = 0; p < MA_PX; ++p) {
        const int li = tid + p * MA_THREADS;
        if (li < nsrc) {
          const float pw = (float)gx[p], ph = (float)gy[p];
          float prob = 0.f;
          if (pw >= bx.x1 && ph >= bx.y1 && pw < bx.x2 && ph < bx.y2) {
            const int iw = (int)__fdiv_rn(__fsub_rn(pw, bx.x1), bx.rw);
            const int ih = (int)__fdiv_rn(__fsub_rn(ph, bx.y1), bx.rh);
            const float* cq = s_cof + n * 128 + ((ih * 2 + iw) & 3) * 32;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) acc = fmaf(bas[p][k], cq[k], acc);
            prob = sigmoidf_acc(acc);
          }
          s_prob[li] = prob;
          if (a.pos_masks) a.pos_masks[dn * (long long)a.hm * a.wm + (long long)gy[p] * a.wm + gx[p]] = prob;
        }
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < GPT; ++g) {
        const int gi = tid + g * MA_THREADS;
        if (gi >= GROUPS) continue;
        const int oy = oy0 + gi / (TOW / 4), oxb = ox0 + (gi % (TOW / 4)) * 4;
        if (oy >= g_ho || oxb >= g_wo) continue;
        const float sy = src_y(oy);
        const int y0 = (int)sy, y1 = min(y0 + 1, a.hm - 1);
        const float ly = sy - (float)y0, hy = 1.f - ly;
        const float* r0 = s_prob + (y0 - sy0) * spw - sx0;
        const float* r1 = s_prob + (y1 - sy0) * spw - sx0;
        uint32_t packed = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ox = oxb + e;
          if (ox < g_wo) {
            const float sx = src_x(ox);
            const int x0 = (int)sx, x1 = min(x0 + 1, a.wm - 1);
            const float lx = sx - (float)x0, hx = 1.f - lx;
            const float v = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
            packed |= (v > a.thr ? 1u : 0u) << (8 * e);
          }
        }
        *reinterpret_cast<uint32_t*>(mrow + (long long)oy * a.pitch + oxb) = packed;
      }
      __syncthreads();  // s_prob is rewritten by the next overlapping detection
    }
  }
}

// SipMask++ mask rescoring tail (sipmask_head.py:638-641): global max-pool of relu(mask_scoring(...)) over the
// last feature map, the detection's own class channel, times the box score.  One wave per detection.
__global__ __launch_bounds__(64) void mask_rescore_kernel(const float* __restrict__ feat, const int64_t* __restrict__ labels,
                                                          const float* __restrict__ det, const int32_t* __restrict__ ndet,
                                                          int max_num, int hw, int C, float* __restrict__ out) {
  const int i = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const long long n = (long long)b * max_num + i;
  if (i >= ndet[b]) {
    if (lane == 0) out[n] = 0.f;
    return;
  }
  const int c = (int)labels[n];
  float m = -INFINITY;
  for (int p = lane; p < hw; p += 64) m = fmaxf(m, feat[(n * hw + p) * C + c]);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if (lane == 0) out[n] = __fmul_rn(m, det[n * 5 + 4]);
}

}  // namespace

extern "C" int sm_mask_rescore(const float* feat, const int64_t* labels, const float* det, const int32_t* ndet, int batch,
                               int max_num, int hw, int channels, float* mask_scores, sm_stream_t stream) {
  if (!feat || !labels || !det || !ndet || !mask_scores) return SM_ERR_BAD_ARG;
  if (batch < 1 || max_num < 1 || hw < 1 || channels < 1) return SM_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(mask_rescore_kernel, dim3(max_num, batch), dim3(64), 0, sm_hip_stream(stream), feat, labels, det, ndet,
                     max_num, hw, channels, mask_scores);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

extern "C" int sm_mask_assemble(const float* basis, int basis_hwc, const float* cofs, const int64_t* keep,
                                const float* det, const int32_t* ndet, int batch, int kmax, int max_num, int hm,
                                int wm, int ho, int wo, int mask_pitch, float box_mul_x, float box_mul_y, float box_div, double up_scale_h,
                                double up_scale_w, float mask_thr,
                                uint8_t* masks, float* pos_masks, const float* per_image, sm_stream_t stream) {
  if (!basis || !cofs || !keep || !det || !ndet || !masks) return SM_ERR_BAD_ARG;
  if (batch < 1 || hm < 1 || wm < 1 || ho < 1 || wo < 1 || mask_pitch % 4 != 0 || mask_pitch < wo || !(up_scale_h > 0) || !(up_scale_w > 0))
    return SM_ERR_BAD_SHAPE;
  MaskArgs a;
  a.basis = basis;
  a.cofs = cofs;
  a.keep = keep;
  a.det = det;
  a.ndet = ndet;
  a.masks = masks;
  a.pos_masks = pos_masks;
  a.kmax = kmax;
  a.max_num = max_num;
  a.hm = hm;
  a.wm = wm;
  a.ho = ho;
  a.wo = wo;
  a.pitch = mask_pitch;
  a.pix_stride = basis_hwc ? 32 : 1;
  a.ch_stride = basis_hwc ? 1 : (long long)hm * wm;
  a.box_mul_x = box_mul_x;
  a.box_mul_y = box_mul_y;
  a.box_div = box_div;
  a.inv_up_x = (float)(1.0 / up_scale_w);  // area_pixel_compute_scale with an explicit scale_factor
  a.inv_up_y = (float)(1.0 / up_scale_h);
  a.thr = mask_thr;
  a.per_image = per_image;   // (the scalar up_scale then bounds the tile's source window: pass the batch's SMALLEST)
  hipStream_t s = sm_hip_stream(stream);
  // pick the widest output tile whose source window (TOW/up + 3) x (TOH/up + 3) fits the
  // per-thread ownership (512 source pixels); wide tiles = full-line mask stores
  auto fits = [&](int tw, int th) {
    return ((int)((double)tw / up_scale_w) + 3) * ((int)((double)th / up_scale_h) + 3) <= MA_SRC_CAP;
  };
  dim3 block(MA_THREADS);
#define SM_MASK_TRY(TW, TH)                                                                                  \
  if (fits(TW, TH)) {                                                                                        \
    hipLaunchKernelGGL((mask_assemble_kernel<TW, TH>), dim3(sm_cdiv(wo, TW), sm_cdiv(ho, TH), batch), block, \
                       0, s, a);                                                                             \
    SM_LAUNCH_CHECK();                                                                                       \
    return SM_OK;                                                                                            \
  }
  SM_MASK_TRY(128, 8)
  SM_MASK_TRY(64, 8)
  SM_MASK_TRY(32, 8)
  SM_MASK_TRY(16, 8)
  SM_MASK_TRY(8, 8)
  SM_MASK_TRY(8, 4)
#undef SM_MASK_TRY
  return SM_ERR_UNSUPPORTED;
}
