// Mask assembly of the launch plan (sipmask_head.py:275-285 tail + :609-633), detection-centric:
//
//   feat_masks = bilinear x4 (relu(sip_mask_lat(...)))            (:285)   [B,32,Hm,Wm]  -- never materialised here
//   logits_q   = feat_masks . cof_q                               (:616-619)
//   pos        = CropSplit(sigmoid(logits))                       (:620-624, crop_split_cuda_kernel.cu:34-52)
//   masks      = bilinear x(2/scale_factor)(pos) > 0.4            (:629-633)
//
// Two observations remove almost all of the HBM traffic of sm_mask_assemble (568 MB per 4-image step):
//  (1) bilinear interpolation is linear, so  (bilinear x4 basis_lo) . cof  ==  bilinear x4 (basis_lo . cof)  up to f32
//      rounding: the 32-channel basis is only ever read at its conv resolution (h0 x w0 = Hm/4 x Wm/4, 8.6 MB per
//      4 images instead of 137 MB written by the x4 upsample kernel and read again here), 4 low-resolution quadrant
//      logit planes per tile are formed in LDS and interpolated;
//  (2) a mask is zero outside its (conservative) box rectangle, and the masks buffer is owned by the launch plan: only
//      the tiles of each detection's rectangle are written, plus zeros over the tiles its slot covered in the
//      PREVIOUS call (per-slot tile ranges kept in a small device-side state buffer).  The buffer contents after the
//      call are exactly those of sm_mask_assemble (zeros elsewhere); the bytes written drop from N*Ho*Wo to ~2x the
//      box areas.
// Work decomposition: a plan kernel turns (detections, previous ranges) into a prefix sum of 128x8-pixel output tiles;
// a fixed grid of blocks walks that list (binary search in an LDS copy of the prefix).

#include "common.h"
#include "experiments.h"

namespace {

constexpr int MF_THREADS = 256;
// Output tile: 128 pixels wide (stores cover whole 128-byte lines) x `th` rows, th in {64, 32, 16, 8} = the tallest whose
// LDS tiles fit (mask_lo_caps).  Round 4: the first version used 8-row tiles, and with 100 image-sized boxes per image (the
// worst case a batch can bring) it spent 3.1 ms per 4-image step = 1.7 % of the HBM roof: an 8-row tile reads a 7-row
// probability window and a 4-5-row conv-resolution window for ONE useful conv-resolution row, pays 5 barriers, a binary
// search and a coefficient load per 1 024 pixels.  A 64-row tile amortises all of that over 8 192 pixels.
constexpr int MF_TW = 128, MF_TH_MAX = 64;
constexpr int MF_SRC_MAXW = 320, MF_SRC_MAXH = 96;   // probability window of a tile (mask_lo_caps enforces both)
// floor(n / d) for n < 2^16, d < 2^16 as one multiply-high with m = ceil(2^32 / d) (n * (m * d - 2^32) < 2^32): the
// window sizes are run-time values, and a hardware-less integer division is ~40 VALU instructions per work item
__device__ __forceinline__ unsigned mf_magic(unsigned d) { return (unsigned)((0x100000000ull + d - 1) / d); }
__device__ __forceinline__ int mf_div(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }
// The LDS tiles of one output tile -- its mask-resolution source window and the conv-resolution window under that --
// are DYNAMIC shared memory sized by the launch's up_scale (a keep_ratio COCO resize gives scale_factor 1.6-2.7, i.e.
// up_scale = 2 / scale_factor down to 0.74: a 128x8 output tile then reads a 176x13 source window)
constexpr int MF_DYN_LDS_MAX = 56 * 1024;   // bytes of dynamic LDS a launch may ask for (beside 6 KB static: 64 KB per block)
constexpr int MF_MAX_ENTRIES = 4096;     // 2 * batch * max_num work-list entries (prefix copy lives in LDS)

struct MaskFArgs {
  const float* basis_lo;   // [B][lo_h][lo_w][32]
  const float* cofs;
  const int64_t* keep;
  const float* det;
  const int32_t* ndet;
  uint8_t* masks;
  int32_t* state;          // [B*max_num][4] PIXEL rectangle written by the previous call (x0, y0, x1, y1), exclusive end
  int th;                  // tile height of this launch (rows)
  int32_t* ranges;         // workspace [B*max_num][8]: new range, previous range
  int32_t* prefix;         // workspace [2*B*max_num + 1]
  int batch, kmax, max_num, lo_h, lo_w, factor, hm, wm, ho, wo, pitch;
  float box_mul_x, box_mul_y, box_div, up_x, up_y, inv_up_x, inv_up_y, inv_f, thr;
  int src_cap, lo_cap;     // floats of the dynamic LDS tiles (mask_lo_caps)
  const float* per_image;  // [batch][8] = (box_mul_x, box_mul_y, up_h, up_w, Ho, Wo, 1/up_h, 1/up_w) or nullptr
};

// geometry of image b: the shared scalars, or this image's row of the per-image table (img_metas[img_id]['scale_factor'],
// sipmask_head.py:517-541,621-633); a.ho / a.wo / a.pitch remain the canvas the mask planes are allocated with
struct MfGeom {
  float mul_x, mul_y, up_x, up_y, inv_up_x, inv_up_y;
  int ho, wo;
};
__device__ __forceinline__ MfGeom mf_geom(const MaskFArgs& a, int b) {
  MfGeom g;
  g.mul_x = a.box_mul_x, g.mul_y = a.box_mul_y, g.up_x = a.up_x, g.up_y = a.up_y;
  g.inv_up_x = a.inv_up_x, g.inv_up_y = a.inv_up_y, g.ho = a.ho, g.wo = a.wo;
  if (a.per_image != nullptr) {
    const float* t = a.per_image + (long long)b * 8;
    g.mul_x = t[0], g.mul_y = t[1], g.up_y = t[2], g.up_x = t[3];
    g.ho = min((int)t[4], a.ho), g.wo = min((int)t[5], a.wo);
    g.inv_up_y = t[6], g.inv_up_x = t[7];
  }
  return g;
}

// ---- plan: per slot the tile range of the new rectangle (mask_rects_kernel's conservative rule, rle.hip) and of the
// previous call's; entry 2d = new tiles, 2d+1 = previous tiles; exclusive prefix sum of the tile counts.
__global__ __launch_bounds__(1024) void mask_plan_kernel(const MaskFArgs a) {
  __shared__ int s_part[1024 / 64];
  __shared__ int s_carry;
  const int nslot = a.batch * a.max_num;
  const int n2 = 2 * nslot;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ntx = (a.wo + MF_TW - 1) / MF_TW, nty = (a.ho + a.th - 1) / a.th;
  if (tid == 0) {
    s_carry = 0;
    a.prefix[0] = 0;
  }
  __syncthreads();
  for (int base = 0; base < n2; base += 1024) {
    const int e = base + tid;
    int cnt = 0;
    if (e < n2) {
      const int d = e >> 1;
      int r[4];
      int pxr[4] = {0, 0, 0, 0};                 // the pixel rectangle behind r (new rectangle: next call's state)
      if (e & 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) pxr[k] = a.state[d * 4 + k];
        r[0] = r[1] = r[2] = r[3] = 0;
        if (pxr[2] > pxr[0] && pxr[3] > pxr[1]) {
          r[0] = pxr[0] / MF_TW;
          r[1] = pxr[1] / a.th;
          r[2] = min((pxr[2] + MF_TW - 1) / MF_TW, ntx);
          r[3] = min((pxr[3] + a.th - 1) / a.th, nty);
        }
      } else {
        const int b = d / a.max_num, i = d - b * a.max_num;
        r[0] = r[1] = r[2] = r[3] = 0;
        if (i < min(a.ndet[b], a.max_num)) {
          const MfGeom G = mf_geom(a, b);
          const float* bx = a.det + (long long)d * 5;
          const float x1 = bx[0] * G.mul_x / a.box_div, y1 = bx[1] * G.mul_y / a.box_div;
          const float x2 = bx[2] * G.mul_x / a.box_div, y2 = bx[3] * G.mul_y / a.box_div;
          auto lo = [&](float v, float up) { return (int)fmaxf(fminf(floorf((v - 1.f) * up) - 2.f, 1e9f), -1e9f); };
          auto hi = [&](float v, float up) { return (int)fmaxf(fminf(ceilf((v + 1.f) * up) + 2.f, 1e9f), -1e9f); };
          const int px0 = max(lo(x1, G.up_x), 0), py0 = max(lo(y1, G.up_y), 0);
          const int px1 = min(hi(x2, G.up_x), G.wo), py1 = min(hi(y2, G.up_y), G.ho);
          if (px1 > px0 && py1 > py0) {
            r[0] = px0 / MF_TW;
            r[1] = py0 / a.th;
            r[2] = min((px1 + MF_TW - 1) / MF_TW, ntx);
            r[3] = min((py1 + a.th - 1) / a.th, nty);
            // what the tiles of this rectangle cover, in pixels: the state of the next call (any tile height)
            pxr[0] = r[0] * MF_TW, pxr[1] = r[1] * a.th, pxr[2] = min(r[2] * MF_TW, a.wo), pxr[3] = min(r[3] * a.th, a.ho);
          }
        }
        // (stashed behind the prefix array; copied into the state after every entry has read the old one)
#pragma unroll
        for (int k = 0; k < 4; ++k) a.prefix[n2 + 1 + d * 4 + k] = pxr[k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) a.ranges[d * 8 + (e & 1) * 4 + k] = r[k];
      cnt = max(r[2] - r[0], 0) * max(r[3] - r[1], 0);
    }
    // block-wide inclusive scan of cnt (wave shuffles + per-wave partials)
    int v = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(v, o, 64);
      if (lane >= o) v += t;
    }
    if (lane == 63) s_part[wv] = v;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wv; ++w) woff += s_part[w];
    const int carry = s_carry;
    if (e < n2) a.prefix[e + 1] = carry + woff + v;
    __syncthreads();
    if (tid == 1023) s_carry = carry + woff + v;
    __syncthreads();
  }
  // the new rectangles become the state of the next call (the old ones were read above: single block, barriers in between)
  for (int d = tid; d < nslot; d += 1024) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a.state[d * 4 + k] = a.prefix[n2 + 1 + d * 4 + k];
  }
}

struct FBox {
  float x1, y1, x2, y2, rw, rh;
};

// fast sigmoid of the mask path: v_exp_f32 / v_rcp_f32 (absolute error < 1e-6 on the probability for |x| < 40; the masks
// are held to the oracle away from |up - thr| < 1e-5, tests/test_gpu_kernels.py).  Rankings use sigmoid_rank (common.h).
__device__ __forceinline__ float mf_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }

// VAR (A/B, SIPMASK_MASK_VARIANT): bit 0 = stage (1) per pixel (one read of the 32 basis values, four dot products) instead
// of per (quadrant, pixel); bit 1 = stage (2) reads its conv-resolution neighbours from per-tile tables instead of
// recomputing them per probability
template <int VAR>
__global__ __launch_bounds__(MF_THREADS) void mask_fused_kernel(const MaskFArgs a) {
  __shared__ __attribute__((aligned(16))) float s_cof[128];
  // per-tile coordinate tables: output column / row -> (first source index relative to the window, fraction)
  __shared__ int s_cx0[MF_TW];
  __shared__ float s_clx[MF_TW];
  __shared__ int s_ry0[MF_TH_MAX];
  __shared__ float s_rly[MF_TH_MAX];
  // ... and of stage (2): probability-window column / row -> (conv-resolution index relative to the window, fraction)
  __shared__ int s_mx0[MF_SRC_MAXW], s_my0[MF_SRC_MAXH];
  __shared__ float s_mlx[MF_SRC_MAXW], s_mly[MF_SRC_MAXH];
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  float* const s_lo_base = s_dyn;                     // [4][lo_cap] quadrant logits at conv resolution
  float* const s_prob = s_dyn + 4 * a.lo_cap;         // [src_cap] probabilities at mask resolution
  // the work list's prefix sums, sized by the launch (2 * batch * max_num + 1 entries; a static 16 KB array cost a block of
  // occupancy per CU, and this kernel lives on occupancy: its tiles are chains of short dependent phases)
  int* const s_prefix = reinterpret_cast<int*>(s_dyn + 4 * a.lo_cap + a.src_cap);
  const int tid = threadIdx.x;
  const int n2 = 2 * a.batch * a.max_num;
  const int TH = a.th;
  for (int i = tid; i <= n2; i += MF_THREADS) s_prefix[i] = a.prefix[i];
  __syncthreads();
  const int total = s_prefix[n2];
  auto lo_c = [&](int g) { return fmaxf(a.inv_f * ((float)g + 0.5f) - 0.5f, 0.f); };   // mask-res -> conv-res coordinate

  for (int w = blockIdx.x; w < total; w += gridDim.x) {
    // entry e with prefix[e] <= w < prefix[e+1]
    int lo = 0, hi = n2;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_prefix[mid] <= w) lo = mid; else hi = mid;
    }
    const int e = lo, d = e >> 1, kind = e & 1;
    const int* rg = a.ranges + d * 8;
    const int r0 = rg[kind * 4 + 0], r1 = rg[kind * 4 + 1], r2 = rg[kind * 4 + 2];
    const int t = w - s_prefix[e];
    const int tx = r0 + t % (r2 - r0), ty = r1 + t / (r2 - r0);
    const int ox0 = tx * MF_TW, oy0 = ty * TH;
    uint8_t* mrow = a.masks + (long long)d * a.ho * a.pitch;
    const int ngroups = (MF_TW / 4) * TH;
    if (kind == 1) {
      // a tile the slot covered last time: zero it unless this call's rectangle rewrites it anyway
      if (!(tx >= rg[0] && tx < rg[2] && ty >= rg[1] && ty < rg[3])) {
        for (int gi = tid; gi < ngroups; gi += MF_THREADS) {
          const int oy = oy0 + gi / (MF_TW / 4), ox = ox0 + (gi % (MF_TW / 4)) * 4;
          if (oy < a.ho && ox < a.wo) *reinterpret_cast<uint32_t*>(mrow + (long long)oy * a.pitch + ox) = 0u;
        }
      }
      continue;
    }
    const int b = d / a.max_num;
    const MfGeom G = mf_geom(a, b);                    // block-uniform: this image's crop / upsample geometry
    auto src_x = [&](int o) { return fmaxf(G.inv_up_x * ((float)o + 0.5f) - 0.5f, 0.f); };
    auto src_y = [&](int o) { return fmaxf(G.inv_up_y * ((float)o + 0.5f) - 0.5f, 0.f); };
    __syncthreads();   // previous tile of this block is done with the LDS tiles
    if (tid < 32) {
      const long long src = ((long long)b * a.kmax + a.keep[d]) * 128;
      *reinterpret_cast<float4*>(s_cof + tid * 4) = *reinterpret_cast<const float4*>(a.cofs + src + tid * 4);
    }
    FBox bx;
    {
      const float* dd = a.det + (long long)d * 5;
      bx.x1 = __fdiv_rn(__fmul_rn(dd[0], G.mul_x), a.box_div);
      bx.y1 = __fdiv_rn(__fmul_rn(dd[1], G.mul_y), a.box_div);
      bx.x2 = __fdiv_rn(__fmul_rn(dd[2], G.mul_x), a.box_div);
      bx.y2 = __fdiv_rn(__fmul_rn(dd[3], G.mul_y), a.box_div);
      bx.rw = (float)(((double)__fsub_rn(bx.x2, bx.x1) + 0.1) / 2.0);   // crop_split_cuda_kernel.cu:47-48
      bx.rh = (float)(((double)__fsub_rn(bx.y2, bx.y1) + 0.1) / 2.0);
    }
    // mask-resolution source window of the tile (as sm_mask_assemble) and the conv-resolution window under it
    const int oxe = min(ox0 + MF_TW, G.wo) - 1, oye = min(oy0 + TH, G.ho) - 1;
    const int sx0 = (int)src_x(ox0), sy0 = (int)src_y(oy0);
    const int sx1 = min((int)src_x(oxe) + 1, a.wm - 1), sy1 = min((int)src_y(oye) + 1, a.hm - 1);
    const int spw = sx1 - sx0 + 1, sph = sy1 - sy0 + 1;
    const int nsrc = spw * sph;
    const int lx0 = (int)lo_c(sx0), ly0 = (int)lo_c(sy0);
    const int lx1 = min((int)lo_c(sx1) + 1, a.lo_w - 1), ly1 = min((int)lo_c(sy1) + 1, a.lo_h - 1);
    const int lpw = lx1 - lx0 + 1, lph = ly1 - ly0 + 1;
    const int nlo = lpw * lph;
    // coordinate tables of stage (3): the expressions of the per-pixel version, evaluated once per tile column / row
    if (tid < MF_TW) {
      const float sx = src_x(ox0 + tid);
      const int x0 = (int)sx;
      s_cx0[tid] = x0 - sx0;
      s_clx[tid] = sx - (float)x0;
    } else if (tid < MF_TW + MF_TH_MAX) {
      const int r = tid - MF_TW;
      const float sy = src_y(oy0 + r);
      const int y0 = (int)sy;
      s_ry0[r] = y0 - sy0;
      s_rly[r] = sy - (float)y0;
    }
    // ... and of stage (2): conv-resolution neighbours of every probability column / row (lo_c: the x4 bilinear's source rule)
    for (int i = tid; i < spw + sph; i += MF_THREADS) {
      if (i < spw) {
        const float fx = lo_c(sx0 + i);
        const int x0 = (int)fx;
        s_mx0[i] = x0 - lx0;
        s_mlx[i] = fx - (float)x0;
      } else {
        const int r = i - spw;
        const float fy = lo_c(sy0 + r);
        const int y0 = (int)fy;
        s_my0[r] = y0 - ly0;
        s_mly[r] = fy - (float)y0;
      }
    }
    __syncthreads();   // s_cof, tables
    // (1) the 4 quadrant logits at conv resolution: per pixel ONE read of its 32 basis values (8 loads in flight) and four
    //     32-long dot products against the coefficient quadrants (each in the reference's channel order)
    const unsigned m_lpw = mf_magic((unsigned)lpw), m_spw = mf_magic((unsigned)spw);
    if constexpr ((VAR & 1) == 0) {
      // Round 5.  A work item is (conv pixel, 4-channel chunk): the 8 lanes of a pixel read its 128-byte basis row as one
      // coalesced line (one 16-byte load each) and reduce their four partial dot products with four shuffles.  (Round 4
      // gave a lane one (quadrant, pixel): 8 loads of 16 bytes at a 128-byte lane stride -- 64 cache lines per wave
      // instruction, every pixel fetched four times -- and the launch ran at the L1's line rate, not at the VALU's: that,
      // not stage (3), was what the worst case cost.)
      const int ck = tid & 7;                              // this lane's chunk: channels 4 ck .. 4 ck + 3 (fixed over the loop)
      float4 cq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) cq[q] = *reinterpret_cast<const float4*>(s_cof + q * 32 + ck * 4);
      const int nitem = ((nlo * 8 + 63) >> 6) << 6;        // whole waves stay in the loop: the shuffles need every lane
      for (int i = tid; i < nitem; i += MF_THREADS) {
        const int p = i >> 3;
        const bool ok = p < nlo;
        const int pp = ok ? p : 0;
        const int py = mf_div(pp, m_lpw), px = pp - py * lpw;
        const float4 v = *reinterpret_cast<const float4*>(
            a.basis_lo + (((long long)b * a.lo_h + (ly0 + py)) * a.lo_w + (lx0 + px)) * 32 + ck * 4);
        float sq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sq[q] = fmaf(v.w, cq[q].w, fmaf(v.z, cq[q].z, fmaf(v.y, cq[q].y, v.x * cq[q].x)));
        // transposing butterfly over the pixel's 8 lanes: xor 4 leaves quadrants (0, 1) on chunks 0-3 and (2, 3) on 4-7, xor 2
        // one quadrant per lane pair, xor 1 completes it
        const bool u4 = (ck & 4) != 0, u2 = (ck & 2) != 0;
        const float t0 = __shfl_xor(u4 ? sq[0] : sq[2], 4, 64), t1 = __shfl_xor(u4 ? sq[1] : sq[3], 4, 64);
        const float a0 = (u4 ? sq[2] : sq[0]) + t0, a1 = (u4 ? sq[3] : sq[1]) + t1;
        float c = (u2 ? a1 : a0) + __shfl_xor(u2 ? a0 : a1, 2, 64);
        c += __shfl_xor(c, 1, 64);
        if (ok && (ck & 1) == 0) s_lo_base[((u4 ? 2 : 0) + (u2 ? 1 : 0)) * a.lo_cap + p] = c;
      }
    } else
    for (int p = tid; p < nlo; p += MF_THREADS) {
      const int py = mf_div(p, m_lpw), px = p - py * lpw;
      const float* bp = a.basis_lo + (((long long)b * a.lo_h + (ly0 + py)) * a.lo_w + (lx0 + px)) * 32;
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(bp + 4 * k);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4* cq = reinterpret_cast<const float4*>(s_cof + q * 32);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float4 c = cq[k];
          acc = fmaf(v[k].x, c.x, acc);
          acc = fmaf(v[k].y, c.y, acc);
          acc = fmaf(v[k].z, c.z, acc);
          acc = fmaf(v[k].w, c.w, acc);
        }
        s_lo_base[q * a.lo_cap + p] = acc;
      }
    }
    __syncthreads();
    // (2) mask-resolution probabilities: quadrant select (CropSplit), bilinear xfactor of that quadrant's logits
    //     (upsample_bilinear_kernel's formula, misc.hip), sigmoid.  The quadrant index of the reference,
    //     (int)((p - x1) / rw), is 0 or 1 inside the box (p < x2 and 2 rw = x2 - x1 + 0.1) and a correctly rounded quotient
    //     of positive floats is >= 1 exactly when numerator >= denominator: a comparison replaces each division.
    const int lxlast = a.lo_w - 1 - lx0, lylast = a.lo_h - 1 - ly0;
    // x4 (the plan's factor): mask columns 4m..4m+3 read the conv columns m-1, m, m+1 with the FIXED fractions 5/8, 7/8,
    // 1/8, 3/8 (lo_c(4m + r) = m + r/4 - 3/8), rows likewise -- a work item is four mask pixels of one row: 6 LDS reads
    // and one vertical blend per conv column instead of 4 reads + 4 table reads + a full bilinear per pixel (round 5).
    // Groups that are not uniformly inside one quadrant of the box, or touch the clamped first / last conv column / row, go
    // pixel by pixel through the general expression below; same operand values, same expression tree.
    if (a.factor == 4) {
      const int gx0 = sx0 >> 2, ngx = (sx1 >> 2) - gx0 + 1;
      const unsigned m_ngx = mf_magic((unsigned)ngx);
      for (int gi = tid; gi < ngx * sph; gi += MF_THREADS) {
        const int yy = ngx == 1 ? gi : mf_div(gi, m_ngx), gxi = gi - yy * ngx;      // (mf_magic(1) does not fit 32 bits)
        const int m = gx0 + gxi, n4 = (sy0 + yy) >> 2, sr = (sy0 + yy) & 3;
        const int g0 = 4 * m;                                 // first mask column of the group (absolute)
        const float ph = (float)(sy0 + yy);
        const float pw0 = (float)g0, pw3 = (float)(g0 + 3);
        const int cy0 = (sr < 2 ? n4 - 1 : n4) - ly0, cxm = m - 1 - lx0;
        const bool inside = ph >= bx.y1 && ph < bx.y2 && pw0 >= bx.x1 && pw3 < bx.x2;
        const int iw0 = __fsub_rn(pw0, bx.x1) >= bx.rw ? 1 : 0, iw3 = __fsub_rn(pw3, bx.x1) >= bx.rw ? 1 : 0;
        const bool fast = inside && iw0 == iw3 && g0 >= sx0 && g0 + 3 <= sx1 && m >= 1 && n4 >= 1 && cxm >= 0 && cy0 >= 0 &&
                          cxm + 2 <= lxlast && cy0 + 1 <= lylast && cxm + 2 < lpw && cy0 + 1 < lph;
        if (fast) {
          const int ih = __fsub_rn(ph, bx.y1) >= bx.rh ? 1 : 0;
          const float* q0 = s_lo_base + (ih * 2 + iw0) * a.lo_cap + cy0 * lpw + cxm;
          const float* q1 = q0 + lpw;
          const float ly = sr == 0 ? 0.625f : (sr == 1 ? 0.875f : (sr == 2 ? 0.125f : 0.375f)), hy = 1.f - ly;
          const float a0 = q0[0], a1 = q0[1], a2 = q0[2], b0 = q1[0], b1 = q1[1], b2 = q1[2];
          const float l0 = hy * (0.375f * a0 + 0.625f * a1) + ly * (0.375f * b0 + 0.625f * b1);
          const float l1 = hy * (0.125f * a0 + 0.875f * a1) + ly * (0.125f * b0 + 0.875f * b1);
          const float l2 = hy * (0.875f * a1 + 0.125f * a2) + ly * (0.875f * b1 + 0.125f * b2);
          const float l3 = hy * (0.625f * a1 + 0.375f * a2) + ly * (0.625f * b1 + 0.375f * b2);
          float* dst = s_prob + yy * spw + (g0 - sx0);
          dst[0] = mf_sigmoid(l0), dst[1] = mf_sigmoid(l1), dst[2] = mf_sigmoid(l2), dst[3] = mf_sigmoid(l3);
          continue;
        }
        for (int r = 0; r < 4; ++r) {
          const int xx = g0 + r - sx0;
          if (xx < 0 || xx >= spw) continue;
          const float pw = (float)(g0 + r);
          float prob = 0.f;
          if (pw >= bx.x1 && ph >= bx.y1 && pw < bx.x2 && ph < bx.y2) {
            const int iw = __fsub_rn(pw, bx.x1) >= bx.rw ? 1 : 0;
            const int ih = __fsub_rn(ph, bx.y1) >= bx.rh ? 1 : 0;
            const float* L = s_lo_base + (ih * 2 + iw) * a.lo_cap;
            const int y0 = s_my0[yy], x0 = s_mx0[xx];
            const float ly = s_mly[yy], lx = s_mlx[xx];
            const int y1 = min(y0 + 1, lylast), x1 = min(x0 + 1, lxlast);
            const float hy = 1.f - ly, hx = 1.f - lx;
            const float* q0 = L + y0 * lpw;
            const float* q1 = L + y1 * lpw;
            prob = mf_sigmoid(hy * (hx * q0[x0] + lx * q0[x1]) + ly * (hx * q1[x0] + lx * q1[x1]));
          }
          s_prob[yy * spw + xx] = prob;
        }
      }
    } else
    for (int li = tid; li < nsrc; li += MF_THREADS) {
      const int yy = mf_div(li, m_spw), xx = li - yy * spw;
      const float pw = (float)(sx0 + xx), ph = (float)(sy0 + yy);
      float prob = 0.f;
      if (pw >= bx.x1 && ph >= bx.y1 && pw < bx.x2 && ph < bx.y2) {
        const int iw = __fsub_rn(pw, bx.x1) >= bx.rw ? 1 : 0;
        const int ih = __fsub_rn(ph, bx.y1) >= bx.rh ? 1 : 0;
        const float* L = s_lo_base + (ih * 2 + iw) * a.lo_cap;
        int y0, x0;
        float ly, lx;
        if constexpr ((VAR & 2) != 0) {
          y0 = s_my0[yy], x0 = s_mx0[xx];
          ly = s_mly[yy], lx = s_mlx[xx];
        } else {
          const float fy = lo_c(sy0 + yy), fx = lo_c(sx0 + xx);
          const int ay = (int)fy, ax = (int)fx;
          y0 = ay - ly0, x0 = ax - lx0;
          ly = fy - (float)ay, lx = fx - (float)ax;
        }
        const int y1 = min(y0 + 1, lylast), x1 = min(x0 + 1, lxlast);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* q0 = L + y0 * lpw;
        const float* q1 = L + y1 * lpw;
        const float logit = hy * (hx * q0[x0] + lx * q0[x1]) + ly * (hx * q1[x0] + lx * q1[x1]);
        prob = mf_sigmoid(logit);
      }
      s_prob[li] = prob;
    }
    __syncthreads();
    // (3) image-resolution bilinear + threshold, 4 pixels per 32-bit store.
    // Exact x2 (scale_factor 1, or any keep_ratio batch without `rescale`: up_scale = 2): the source of output pixel o is
    // o/2 - 1/4, so the four pixels of a group 4j..4j+3 read the source columns 2j-1..2j+2 with the FIXED fractions
    // 3/4, 1/4, 3/4, 1/4 and a row reads two source rows with fraction 1/4 (odd rows) or 3/4 (even rows): 8 LDS reads, four
    // vertical blends and four horizontal ones per group, no coordinate tables (round 5: ~10 instead of ~40 VALU per
    // output pixel on the worst case).  The expression tree is the general path's, hy * (hx * a + lx * b) + ly * (hx * c +
    // lx * d), with the same operand values -- bit-identical results; groups on the window's clamped edges (first row /
    // column of the image, last source row / column) take the general path.
    const int xlast = a.wm - 1 - sx0, ylast = a.hm - 1 - sy0;      // clamp of the +1 neighbour, window-relative
    const bool x2 = G.inv_up_x == 0.5f && G.inv_up_y == 0.5f;
    for (int gi = tid; gi < ngroups; gi += MF_THREADS) {
      const int ry = gi / (MF_TW / 4), cx = (gi % (MF_TW / 4)) * 4;
      const int oy = oy0 + ry, oxb = ox0 + cx;
      if (oy >= G.ho || oxb >= G.wo) continue;
      uint32_t packed = 0u;
      const int fy0 = ((oy + 1) >> 1) - 1 - sy0, fx0 = (oxb >> 1) - 1 - sx0;      // exact x2: first source row / column
      if (x2 && oy >= 1 && oxb >= 4 && oxb + 3 < G.wo && fy0 + 1 <= ylast && fx0 + 3 <= xlast) {
        const float ly = (oy & 1) ? 0.25f : 0.75f, hy = 1.f - ly;
        const float* p0 = s_prob + fy0 * spw + fx0;
        const float* p1 = p0 + spw;
        const float a0 = p0[0], a1 = p0[1], a2 = p0[2], a3 = p0[3];
        const float b0 = p1[0], b1 = p1[1], b2 = p1[2], b3 = p1[3];
        // pixel k: x0 = fx0 + {0, 1, 1, 2}[k], lx = {3/4, 1/4, 3/4, 1/4}[k]
        const float v0 = hy * (0.25f * a0 + 0.75f * a1) + ly * (0.25f * b0 + 0.75f * b1);
        const float v1 = hy * (0.75f * a1 + 0.25f * a2) + ly * (0.75f * b1 + 0.25f * b2);
        const float v2 = hy * (0.25f * a1 + 0.75f * a2) + ly * (0.25f * b1 + 0.75f * b2);
        const float v3 = hy * (0.75f * a2 + 0.25f * a3) + ly * (0.75f * b2 + 0.25f * b3);
        packed = (v0 > a.thr ? 1u : 0u) | (v1 > a.thr ? 0x100u : 0u) | (v2 > a.thr ? 0x10000u : 0u) | (v3 > a.thr ? 0x1000000u : 0u);
      } else {
        const int y0 = s_ry0[ry], y1 = min(y0 + 1, ylast);
        const float ly = s_rly[ry], hy = 1.f - ly;
        const float* p0 = s_prob + y0 * spw;
        const float* p1 = s_prob + y1 * spw;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (oxb + k < G.wo) {
            const int x0 = s_cx0[cx + k], x1 = min(x0 + 1, xlast);
            const float lx = s_clx[cx + k], hx = 1.f - lx;
            const float v = hy * (hx * p0[x0] + lx * p0[x1]) + ly * (hx * p1[x0] + lx * p1[x1]);
            packed |= (v > a.thr ? 1u : 0u) << (8 * k);
          }
        }
      }
      *reinterpret_cast<uint32_t*>(mrow + (long long)oy * a.pitch + oxb) = packed;
    }
  }
}

}  // namespace

// LDS tile sizes of a launch (floats): upper bounds of the source windows of one MF_TW x MF_TH output tile; false when
// the geometry does not fit (then the caller assembles from the upsampled basis: sm_mask_assemble)
static bool mask_lo_caps_th(int th, int factor, double up_scale_h, double up_scale_w, int* src_cap, int* lo_cap, int nentries) {
  const double spw_d = (double)MF_TW / up_scale_w + 3.0, sph_d = (double)th / up_scale_h + 3.0;
  if (spw_d * sph_d > 1.0e6) return false;
  const int spw = (int)spw_d, sph = (int)sph_d;
  if (spw > MF_SRC_MAXW || sph > MF_SRC_MAXH) return false;       // the kernel's coordinate tables
  *src_cap = spw * sph;
  *lo_cap = ((spw / factor + 3) * (sph / factor + 3) + 3) & ~3;        // 16-byte aligned quadrant planes
  return (size_t)(*src_cap + 4 * *lo_cap + nentries + 1) * sizeof(float) <= (size_t)MF_DYN_LDS_MAX;
}

static bool mask_lo_caps(int batch, int max_num, int factor, double up_scale_h, double up_scale_w, int* src_cap, int* lo_cap,
                         int* th_out = nullptr) {
  if (batch < 1 || max_num < 1 || factor < 1 || !(up_scale_h > 0) || !(up_scale_w > 0)) return false;
  if (2 * (long long)batch * max_num > MF_MAX_ENTRIES) return false;
  // the tallest tile whose windows fit 28 KB (5 blocks per CU), else the tallest that fits at all
  for (int pass = 0; pass < 2; ++pass) {
    for (int th = MF_TH_MAX; th >= 8; th >>= 1) {
      if (mask_lo_caps_th(th, factor, up_scale_h, up_scale_w, src_cap, lo_cap, 2 * batch * max_num) &&
          (pass == 1 || (size_t)(*src_cap + 4 * *lo_cap + 2 * batch * max_num + 1) * sizeof(float) <= 28u * 1024u)) {
        if (th_out) *th_out = th;
        return true;
      }
    }
  }
  return false;
}

extern "C" int sm_mask_assemble_lo_supported(int batch, int max_num, int factor, double up_scale_h, double up_scale_w) {
  int s, l;
  return mask_lo_caps(batch, max_num, factor, up_scale_h, up_scale_w, &s, &l) ? 1 : 0;
}

extern "C" int64_t sm_mask_assemble_lo_workspace(int batch, int max_num) {
  if (batch < 1 || max_num < 1) return 0;
  const int64_t nslot = (int64_t)batch * max_num;
  return nslot * 8 * 4 + (2 * nslot + 1) * 4 + nslot * 4 * 4 + 64;     // ranges, prefix, next-state stash
}

extern "C" int sm_mask_assemble_lo(const float* basis_lo, int lo_h, int lo_w, int factor, const float* cofs,
                                   const int64_t* keep, const float* det, const int32_t* ndet, int batch, int kmax,
                                   int max_num, int ho, int wo, int mask_pitch, float box_mul_x, float box_mul_y,
                                   float box_div, double up_scale_h, double up_scale_w, float mask_thr, uint8_t* masks,
                                   int32_t* state, void* workspace, const float* per_image, sm_stream_t stream) {
  if (!basis_lo || !cofs || !keep || !det || !ndet || !masks || !state || !workspace) return SM_ERR_BAD_ARG;
  if (batch < 1 || max_num < 1 || lo_h < 1 || lo_w < 1 || factor < 1 || ho < 1 || wo < 1 || mask_pitch % 4 != 0 ||
      mask_pitch < wo || !(up_scale_h > 0) || !(up_scale_w > 0) || !(box_div != 0.f))
    return SM_ERR_BAD_SHAPE;
  int src_cap, lo_cap, th;
  if (!mask_lo_caps(batch, max_num, factor, up_scale_h, up_scale_w, &src_cap, &lo_cap, &th)) return SM_ERR_UNSUPPORTED;
  MaskFArgs a;
  a.th = th;
  a.basis_lo = basis_lo;
  a.cofs = cofs;
  a.keep = keep;
  a.det = det;
  a.ndet = ndet;
  a.masks = masks;
  a.state = state;
  a.ranges = (int32_t*)workspace;
  a.prefix = a.ranges + (size_t)batch * max_num * 8;
  a.batch = batch;
  a.kmax = kmax;
  a.max_num = max_num;
  a.lo_h = lo_h;
  a.lo_w = lo_w;
  a.factor = factor;
  a.hm = lo_h * factor;
  a.wm = lo_w * factor;
  a.ho = ho;
  a.wo = wo;
  a.pitch = mask_pitch;
  a.box_mul_x = box_mul_x;
  a.box_mul_y = box_mul_y;
  a.box_div = box_div;
  a.up_x = (float)up_scale_w;
  a.up_y = (float)up_scale_h;
  a.inv_up_x = (float)(1.0 / up_scale_w);
  a.inv_up_y = (float)(1.0 / up_scale_h);
  a.inv_f = 1.f / (float)factor;
  a.thr = mask_thr;
  a.src_cap = src_cap;       // (with a per-image table the scalar up_scale must be the batch's SMALLEST: it sizes the LDS tiles)
  a.lo_cap = lo_cap;
  a.per_image = per_image;
  hipStream_t s = sm_hip_stream(stream);
  hipLaunchKernelGGL(mask_plan_kernel, dim3(1), dim3(1024), 0, s, a);
  // variant 2 of the four stage layouts (measured in round 4 on 100 image-sized boxes x 4 images: 1.09 / 1.29 / 1.07 / 1.30
  // ms for 0..3, profiles/r04_mask_assemble_variants_ab.txt); the others exist in the experiment build only
  const size_t dyn = (size_t)(src_cap + 4 * lo_cap + 2 * batch * max_num + 1) * sizeof(float);
#ifdef SM_EXPERIMENTS
  static const int variant = sm_experiment_env("SIPMASK_MASK_VARIANT", 2) & 3;
  switch (variant) {
    case 1: hipLaunchKernelGGL(mask_fused_kernel<1>, dim3(2048), dim3(MF_THREADS), dyn, s, a); break;
    case 2: hipLaunchKernelGGL(mask_fused_kernel<2>, dim3(2048), dim3(MF_THREADS), dyn, s, a); break;
    case 3: hipLaunchKernelGGL(mask_fused_kernel<3>, dim3(2048), dim3(MF_THREADS), dyn, s, a); break;
    default: hipLaunchKernelGGL(mask_fused_kernel<0>, dim3(2048), dim3(MF_THREADS), dyn, s, a); break;
  }
#else
  hipLaunchKernelGGL(mask_fused_kernel<2>, dim3(2048), dim3(MF_THREADS), dyn, s, a);
#endif
  SM_LAUNCH_CHECK();
  return SM_OK;
}
