// Input pipeline on the GPU: the reference's CPU-worker transforms Resize(keep_ratio) -> Normalize ->
// Pad(size_divisor) -> ImageToTensor (M/mmdet/datasets/pipelines/transforms.py:24-175,362-403,
// M/configs/sipmask/sipmask_r50_caffe_fpn_gn_1x.py:72-87) fused into one HBM-bound kernel: every output pixel reads
// its 4 source pixels of the uint8 HWC image, interpolates (cv2.INTER_LINEAR geometry: half-pixel centres, edge
// clamp), rounds to uint8 as cv2.resize returns it, normalises and writes float NCHW (zero in the padding).
#include "common.h"

namespace {

struct PreArgs {
  const uint8_t* src;
  int h0, w0;          // source size
  int nh, nw;          // resized size (img_shape)
  int hp, wp;          // padded size (pad_shape)
  float sy, sx;        // source step per destination pixel = h0/nh, w0/nw (cv2: inv_scale)
  float mean[3], inv_std[3];
  int to_rgb;
  float* out;          // [3][hp][wp]
};

__global__ void preprocess_kernel(const PreArgs a) {
  const long long total = (long long)a.hp * a.wp;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(t % a.wp), y = (int)(t / a.wp);
    float v[3] = {0.f, 0.f, 0.f};
    if (x < a.nw && y < a.nh) {
      // cv2 resize, INTER_LINEAR: fx = (dx + 0.5) * inv_scale - 0.5, sx = floor(fx), clamped at the borders
      float fy = ((float)y + 0.5f) * a.sy - 0.5f, fx = ((float)x + 0.5f) * a.sx - 0.5f;
      int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
      float ly = fy - (float)y0, lx = fx - (float)x0;
      if (y0 < 0) { y0 = 0; ly = 0.f; }
      if (x0 < 0) { x0 = 0; lx = 0.f; }
      if (y0 >= a.h0 - 1) { y0 = a.h0 - 1; ly = 0.f; }
      if (x0 >= a.w0 - 1) { x0 = a.w0 - 1; lx = 0.f; }
      const int y1 = min(y0 + 1, a.h0 - 1), x1 = min(x0 + 1, a.w0 - 1);
      const uint8_t* p00 = a.src + ((long long)y0 * a.w0 + x0) * 3;
      const uint8_t* p01 = a.src + ((long long)y0 * a.w0 + x1) * 3;
      const uint8_t* p10 = a.src + ((long long)y1 * a.w0 + x0) * 3;
      const uint8_t* p11 = a.src + ((long long)y1 * a.w0 + x1) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float top = (float)p00[c] + lx * ((float)p01[c] - (float)p00[c]);
        const float bot = (float)p10[c] + lx * ((float)p11[c] - (float)p10[c]);
        const float r = rintf(top + ly * (bot - top));            // uint8 result of cv2.resize (round half to even)
        const int oc = a.to_rgb ? 2 - c : c;                      // mmcv.imnormalize: BGR->RGB before mean/std
        v[oc] = (fminf(fmaxf(r, 0.f), 255.f) - a.mean[oc]) * a.inv_std[oc];
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) a.out[(long long)c * total + t] = v[c];
  }
}

}  // namespace

extern "C" int sm_preprocess_u8(const uint8_t* src, int src_h, int src_w, int new_h, int new_w, int pad_h, int pad_w,
                                const float* mean, const float* std, int to_rgb, float* out_chw, sm_stream_t stream) {
  if (!src || !mean || !std || !out_chw) return SM_ERR_BAD_ARG;
  if (src_h < 1 || src_w < 1 || new_h < 1 || new_w < 1 || pad_h < new_h || pad_w < new_w) return SM_ERR_BAD_SHAPE;
  PreArgs a;
  a.src = src;
  a.h0 = src_h, a.w0 = src_w, a.nh = new_h, a.nw = new_w, a.hp = pad_h, a.wp = pad_w;
  a.sy = (float)((double)src_h / (double)new_h);
  a.sx = (float)((double)src_w / (double)new_w);
  for (int c = 0; c < 3; ++c) {
    a.mean[c] = mean[c];
    a.inv_std[c] = 1.f / std[c];
  }
  a.to_rgb = to_rgb;
  a.out = out_chw;
  const long long total = (long long)pad_h * pad_w;
  hipLaunchKernelGGL(preprocess_kernel, dim3((int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256)), dim3(256), 0,
                     sm_hip_stream(stream), a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
