// FeatureAlign's deformable 3x3 convolution in SPLIT PRECISION with the input window resident in LDS (gfx950; round 5).
// Semantics: deformable_im2col + addmm of M/mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:85-115,191-243 on fp32 tensors,
// as M/mmdet/models/anchor_heads/sipmask_head.py:21-55 calls it.
//
// Why.  The x3 head plan (split_x3.hip) keeps the reference head's fp32 activations and feeds the f16 matrix pipe two
// binary16 halves per value (three MFMA terms per product).  Every other conv of that plan reads a pre-split operand by
// LDS-DMA; FeatureAlign is the one conv whose operand VALU has to produce (bilinear samples), and until round 4 it ran on
// conv_f32.hip's gather loader: four corner loads from L2 per (position, tap, 4 channels), 0.80 ms per 4 images -- longer
// than a grouped tower launch with twice its FLOPs.  This kernel is deform_patch.hip's idea on f32 data:
//   * a block owns 256 output positions of one image and holds the f32 window of HALF a deformable group (32 channels =
//     one 128-byte line per pixel, 640 pixels = 80 KB) in LDS; all nine taps blend their corners from there;
//   * K order = (group, channel half, tap, 32 channels); the weights of a K step are 128 contiguous bytes of a cout row
//     laid out [hi 32 | lo 32] binary16 (host: prep_deform_weight_x3), DMAed into a double-buffered [256][128 B] stage;
//   * a lane blends ITS position's 8 channels in f32 (the expression of conv_f32.hip's deformable loader), splits the
//     sample hi = f16(v), lo = f16(v - hi), and multiplies against all 256 couts: w_hi*x_hi + w_hi*x_lo + w_lo*x_hi on
//     v_mfma_f32_32x32x16_f16 (f32 accumulation; the dropped lo*lo term is 2^-22).  48 MFMAs per wave and K step against
//     ~200 VALU: the kernel is bound by the matrix pipe, not by the blend (deform_patch.hip is the other way round);
//   * two tile shapes in one launch: 8 rows x 32 columns, and 32 rows x 8 columns for the strip a level's width leaves
//     beyond a multiple of 32 (168 = 5 x 32 + 8: thirteen row tiles with 24 of 32 columns empty become four column
//     tiles) -- 392 instead of 440 tiles per four 800 x 1344 images.  Both windows are 640 pixels;
//   * a corner outside the window (|offset| > 3) sends that wave through a global gather for that tap, as in deform_patch.hip;
//   * epilogue: acc * acc_scale + bias, ReLU, f32 rows; the output's GroupNorm statistics as fixed-point sums (common.h).
#include <utility>

#include "common.h"
#include "experiments.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int DX_R = 3;
constexpr int DX_PPIX = 640;                      // window pixels of either shape: 16 x 40 (row tile), 40 x 16 (column tile)
constexpr int DX_PPW = DX_PPIX / 8 / 8;           // patch DMA pieces per wave (10)
constexpr int DX_BCO = 256;
constexpr int DX_WSTAGE = DX_BCO * 128;           // one K step of weights: 256 cout rows x [hi 32 | lo 32] binary16
constexpr int DX_THREADS = 512;
constexpr int DX_LDS = 2 * DX_WSTAGE + DX_PPIX * 128;
constexpr int DX_KSTEP = 64;                      // binary16 elements of a cout row per K step

struct DeformX3Args {
  const float* x;
  const uint16_t* w;              // binary16 [cout_pad][dg * 2 * 9][hi 32 | lo 32]
  const float* bias;
  const float* offset;
  float* y;
  unsigned long long* gn_stats;   // fixed point (common.h: gn_fix), or null
  int nlev, batch;
  int h[SM_MAX_LEVELS], w_[SM_MAX_LEVELS];
  long long in_row0[SM_MAX_LEVELS], out_row0[SM_MAX_LEVELS];
  int tile0[SM_MAX_LEVELS + 1];   // first position tile of each level
  int tpi[SM_MAX_LEVELS];         // tiles per image
  int nrow_t[SM_MAX_LEVELS];      // row tiles (8 x 32) per image: the first nrow_t of an image's tiles
  int ntx[SM_MAX_LEVELS];         // row tiles along x
  int xb[SM_MAX_LEVELS];          // first column of the column-tile strip
  int nbx[SM_MAX_LEVELS];         // column tiles (32 x 8) along x inside the strip
  int cout, ntn, dg;
  int in_cstride, out_cstride, out_coff;
  unsigned flags;
  float acc_scale;
  int nblk;
};

template <int N, typename F, int... Is>
__device__ __forceinline__ void xfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void xfor(F&& f) {
  xfor_impl<N>(f, std::make_integer_sequence<int, N>{});
}

__device__ __attribute__((aligned(16))) const unsigned int g_zero16x[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ int dx_xcd_tile(int b, int nblk) {
  const int xcd = b & 7, xq = nblk >> 3, xr = nblk & 7;
  return (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
}

// ABL: ablations for the micro-benchmark (wrong results by construction; `make EXPERIMENTS=1` only): 1 no corner reads /
// blend / set-up in the steps (a constant operand), 2 no weight DMA in the K loop, 4 no MFMAs and no fragment reads.
template <int ABL>
__global__ __launch_bounds__(DX_THREADS, 1) void deform_patch_x3_kernel(const DeformX3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [W stage 0][W stage 1][window]
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  typedef const __attribute__((address_space(3))) f32x4 lds_f32x4;
  typedef const __attribute__((address_space(3))) half8 lds_half8;
  lds_u8* const smem3 = (lds_u8*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;

  // ---- tile decode (wave-uniform)
  const int tlin = dx_xcd_tile(blockIdx.x, a.nblk);
  const int nt = tlin % a.ntn;
  const int mt = tlin / a.ntn;
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && mt >= a.tile0[l]) lev = l;
  const int H = a.h[lev], W = a.w_[lev];
  const int ti = mt - a.tile0[lev];
  const int n = ti / a.tpi[lev];
  const int tt = ti - n * a.tpi[lev];
  const bool col_tile = tt >= a.nrow_t[lev];                      // 32 rows x 8 columns on the right-hand strip
  int y0, x0;
  if (!col_tile) {
    const int ty = tt / a.ntx[lev];
    y0 = ty * 8, x0 = (tt - ty * a.ntx[lev]) * 32;
  } else {
    const int e = tt - a.nrow_t[lev];
    const int by = e / a.nbx[lev];
    y0 = by * 32, x0 = a.xb[lev] + (e - by * a.nbx[lev]) * 8;
  }
  const int lgw = col_tile ? 3 : 5;                               // log2 of the tile width
  const int PW = (col_tile ? 8 : 32) + 2 + 2 * DX_R;              // window pitch in pixels: 16 / 40 (even: swizzle parity)
  const int PH = DX_PPIX / PW;                                    // 40 / 16
  const int py0 = y0 - 1 - DX_R, px0 = x0 - 1 - DX_R;             // image coordinates of window pixel (0, 0)
  const long long img_row0 = a.in_row0[lev] + (long long)n * H * W;
  const float* const ximg = a.x + img_row0 * a.in_cstride;

  unsigned char* const Wb0 = smem;
  unsigned char* const Pb = smem + 2 * DX_WSTAGE;

  // ---- loaders.  Every DMA piece is 8 rows x 128 B: lane L -> row (L >> 3), physical 16-byte slot (L & 7), which holds
  // the logical chunk (L & 7) ^ ((row >> 1) & 7) of the row (swizzle applied on the source side).
  const unsigned long long zero_page = (unsigned long long)g_zero16x;
  auto dma_patch = [&](int g, int ch) {                           // 32 channels (g, ch) of the window -> Pb
    xfor<DX_PPW>([&](auto I) {
      constexpr int i = decltype(I)::value;
      const int pix = (wave + 8 * i) * 8 + (lane >> 3);
      const int pr = col_tile ? (pix >> 4) : (pix / 40);
      const int pc = pix - pr * PW;
      int py0_l = py0;
      asm volatile("" : "+s"(py0_l));                             // opaque: keeps the 10 addresses out of the K loop's registers
      const int ih = py0_l + pr, iw = px0 + pc;
      const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
      const int chunk = (lane & 7) ^ ((pix >> 1) & 7);
      const unsigned long long pm = ok ? ~0ull : 0ull;
      const unsigned long long src =
          ((unsigned long long)(ximg + ((long long)(ih * W + iw) * a.in_cstride + chunk * 4 + g * 64 + ch * 32)) & pm) |
          (zero_page & ~pm);
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(Pb + (wave + 8 * i) * 1024), 16, 0, 0);
    });
  };
  // weights: uniform base (SGPRs) + one per-lane 32-bit offset.  Piece i of this wave = rows wave*32 + 8i .. +7; lane L ->
  // row r = L >> 3 of the piece, physical slot L & 7 = logical chunk (L & 7) ^ (r >> 1) ^ 4*(i & 1)
  const long long Kp = (long long)a.dg * 18 * DX_KSTEP;
  const unsigned wvoff0 = (unsigned)((lane >> 3) * (int)Kp + (((lane & 7) ^ ((lane >> 4) & 3)) * 8));
  const unsigned wvoff1 = wvoff0 ^ 32u;
  auto dma_w = [&](int s, int buf) {                              // K step s -> Wb[buf]
    unsigned char* dst = Wb0 + buf * DX_WSTAGE;
    const uint16_t* const ubase = a.w + ((long long)(nt * DX_BCO + wave * 32) * Kp + (long long)s * DX_KSTEP);
    xfor<4>([&](auto I) {
      constexpr int i = decltype(I)::value;
      __builtin_amdgcn_global_load_lds((glb_void*)(ubase + (long long)(8 * i) * Kp + ((i & 1) ? wvoff1 : wvoff0)),
                                       (lds_void*)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
    });
  };

  // ---- this lane's output position
  const int wy = wave << (5 - lgw);                               // first tile row of this wave
  const int oy = y0 + wy + (l31 >> lgw), ox = x0 + (l31 & ((1 << lgw) - 1));
  const bool row_live = y0 + wy < H;                              // wave-uniform
  const bool pvalid = oy < H && ox < W;
  const long long orow = a.out_row0[lev] + (long long)n * H * W + (long long)oy * W + ox;
  const float* const offp = a.offset + (pvalid ? orow : 0ll) * (a.dg * 18);

  f32x16 acc[8];
#pragma unroll
  for (int tc = 0; tc < 8; ++tc)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[tc][e] = 0.f;

  const int rs8 = (l31 >> 1) & 7;
  const int wrow_off = l31 * 128;

  // ---- per-tap state of this lane's position (deform_conv_cuda_kernel.cu:85-115,216-229 with rhi = oy - 1, rwi = ox - 1)
  struct Tap {
    float w1, w2, w3, w4;
    int c1, c2, c3, c4;           // LDS byte addresses of the four corners, channels khalf*8 .. +3 of sub-step 0
    bool far;                     // wave-uniform: some lane samples outside the LDS window
  };
  auto setup = [&](Tap& t, int g_, int ch_, int tap_, float2 off) {
    const int kh = tap_ / 3, kw = tap_ - kh * 3;
    const float h_im = (float)(oy - 1 + kh) + off.x;
    const float w_im = (float)(ox - 1 + kw) + off.y;
    const bool inr = pvalid && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h_low = inr ? (int)hf : 0, w_low = inr ? (int)wf : 0;
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h_im - hf, lw = w_im - wf;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const bool t_ok = h_low >= 0, b_ok = h_high <= H - 1, l_ok = w_low >= 0, r_ok = w_high <= W - 1;
    t.w1 = (inr && t_ok && l_ok) ? hh * hw : 0.f;
    t.w2 = (inr && t_ok && r_ok) ? hh * lw : 0.f;
    t.w3 = (inr && b_ok && l_ok) ? lh * hw : 0.f;
    t.w4 = (inr && b_ok && r_ok) ? lh * lw : 0.f;
    // corners in window coordinates; a lane whose sample lies outside the image has zero weights and may read anywhere
    const int pr = h_low - py0, pc = w_low - px0;
    const bool in_patch = pr >= 0 && pr <= PH - 2 && pc >= 0 && pc <= PW - 2;
    t.far = __builtin_amdgcn_ballot_w64(inr && !in_patch) != 0ull;
    // pixel p, 16-byte chunk j -> byte p * 128 + ((j ^ ((p >> 1) & 7)) * 16); p and p + PW have the same parity
    const int prc = min(max(pr, 0), PH - 2), pcc = min(max(pc, 0), PW - 2);
    const int p1 = prc * PW + pcc;
    const int a1 = 2 * DX_WSTAGE + p1 * 128 + khalf * 32;        // chunk 2 * khalf of sub-step 0
    t.c1 = a1 ^ (((p1 >> 1) & 7) * 16);
    t.c2 = (a1 + 128) ^ ((((p1 + 1) >> 1) & 7) * 16);
    t.c3 = (a1 + PW * 128) ^ ((((p1 + PW) >> 1) & 7) * 16);
    t.c4 = (a1 + PW * 128 + 128) ^ ((((p1 + PW + 1) >> 1) & 7) * 16);
  };
  // the 4 channels kk*16 + khalf*8 + hf*4 .. +3 of the four corners, from the LDS window ...
  auto corners_lds = [&](const Tap& t, int kk, int hf, f32x4 (&q)[4]) {
    const int x = kk * 64 + hf * 16;
    q[0] = *reinterpret_cast<lds_f32x4*>(smem3 + (t.c1 ^ x));
    q[1] = *reinterpret_cast<lds_f32x4*>(smem3 + (t.c2 ^ x));
    q[2] = *reinterpret_cast<lds_f32x4*>(smem3 + (t.c3 ^ x));
    q[3] = *reinterpret_cast<lds_f32x4*>(smem3 + (t.c4 ^ x));
  };
  // ... or, for a wave with a sample outside the window, from global memory (clamped addresses: the weights carry the zero
  // padding; the top-left pixel is recomputed from the offset -- this path keeps nothing in the fast path's registers).
  // Only the far K step of that wave takes this path.
  auto corners_global = [&](int g_, int ch_, int tap_, float2 off, int kk, int hf, f32x4 (&q)[4]) {
    const int kh = tap_ / 3, kw = tap_ - kh * 3;
    const float h_im = (float)(oy - 1 + kh) + off.x, w_im = (float)(ox - 1 + kw) + off.y;
    const bool inr = pvalid && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
    const int hlo = inr ? (int)floorf(h_im) : 0, wlo = inr ? (int)floorf(w_im) : 0;
    const int hl = min(max(hlo, 0), H - 1), hh_ = min(max(hlo + 1, 0), H - 1);
    const int wl = min(max(wlo, 0), W - 1), wh_ = min(max(wlo + 1, 0), W - 1);
    const float* const gp = ximg + g_ * 64 + ch_ * 32 + khalf * 8 + kk * 16 + hf * 4;
    q[0] = *reinterpret_cast<const f32x4*>(gp + (long long)(hl * W + wl) * a.in_cstride);
    q[1] = *reinterpret_cast<const f32x4*>(gp + (long long)(hl * W + wh_) * a.in_cstride);
    q[2] = *reinterpret_cast<const f32x4*>(gp + (long long)(hh_ * W + wl) * a.in_cstride);
    q[3] = *reinterpret_cast<const f32x4*>(gp + (long long)(hh_ * W + wh_) * a.in_cstride);
  };
  // blend of 8 channels in f32, two at a time (v_pk_mul_f32 / v_pk_fma_f32 with the weight broadcast by op_sel):
  // ((w1*f1 + w2*f2) + w3*f3) + w4*f4, the expression of conv_f32.hip's deformable loader; then the split of split_x3.hip:
  // hi = f16(v) saturated (one v_cvt_pk_f16_f32 per pair), lo = f16(v - hi) (v_fma_mixlo/hi_f16: the binary16 hi read as
  // a source of an f32 fma whose result is rounded once to binary16 -- v - hi is exact in f32)
  auto blend_split = [&](const Tap& t, const f32x4 (&q)[4], int hf, u32x4& xhi, u32x4& xlo) {
    const f32x2 w1v = {t.w1, t.w1}, w2v = {t.w2, t.w2}, w3v = {t.w3, t.w3}, w4v = {t.w4, t.w4};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const f32x2 v1 = {q[0][2 * d], q[0][2 * d + 1]}, v2 = {q[1][2 * d], q[1][2 * d + 1]};
      const f32x2 v3 = {q[2][2 * d], q[2][2 * d + 1]}, v4 = {q[3][2 * d], q[3][2 * d + 1]};
      const f32x2 r = w1v * v1 + w2v * v2 + w3v * v3 + w4v * v4;
      const float c0 = fminf(fmaxf(r[0], -65504.f), 65504.f), c1 = fminf(fmaxf(r[1], -65504.f), 65504.f);
      uint32_t hh, ll;
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hh) : "v"(c0), "v"(c1));
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ll) : "v"(hh), "v"(c0));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(ll) : "v"(hh), "v"(c1));
      xhi[hf * 2 + d] = hh, xlo[hf * 2 + d] = ll;
    }
  };

  // K step index s = (g * 2 + ch) * 9 + tap.  Control flow exists only at the start of a step, right behind the barrier,
  // where no LDS / memory counter is pending: a wave whose tap samples outside the window somewhere (tp.far) gathers both
  // sub-steps' operands from global memory there; the first step of a window blends its first operand there.
  const int nstep = a.dg * 18;
  dma_patch(0, 0);
  dma_w(0, 0);
  Tap tp;
  setup(tp, 0, 0, 0, *reinterpret_cast<const float2*>(offp));
  __syncthreads();

  u32x4 xa_hi, xa_lo, xb_hi, xb_lo;                 // operands of sub-step 0 / 1 as 4 x 2 binary16 (ping-pong: no copies)
  bool have_x = false;                              // (xa_hi, xa_lo) of this step were blended by the step before
  int g = 0, ch = 0, tap = 0;
  for (int s = 0; s < nstep; ++s) {
    int g1 = g, ch1 = ch, tap1 = tap + 1;
    if (tap1 == 9) {
      tap1 = 0;
      ch1 ^= 1;
      if (ch1 == 0) ++g1;
    }
    const bool more = s + 1 < nstep;
    // the next tap's offsets, requested in front of the weight DMA (vmcnt is in order) and read in the middle of the step
    const float2 off_nx = *reinterpret_cast<const float2*>(offp + (more ? (g1 * 9 + tap1) * 2 : 0));   // unconditional
    if (more && !(ABL & 2)) dma_w(s + 1, (s + 1) & 1);
    if (row_live) {
      const int wb = (s & 1) * DX_WSTAGE + wrow_off + ((khalf ^ rs8) * 16);   // hi chunk kk * 2 + khalf: ^ (kk * 32); lo: ^ 64
      auto frag_addr = [&](int kk, int tc) { return (wb ^ (kk * 32)) + tc * 32 * 128; };
      half8 wh[4], wl[4];
      // a quad = four cout tiles: 12 MFMAs -- w_hi * x_hi, w_hi * x_lo (w_hi[t] is dead behind it: the next quad's fragment
      // lands in the same registers while the other MFMAs run), w_lo * x_hi (likewise).  Accumulators of one cout tile are
      // 4 MFMAs apart.
      auto quad = [&](auto KK, auto Q, const u32x4& xhi_, const u32x4& xlo_) {
        const half8 xhi = __builtin_bit_cast(half8, xhi_), xlo = __builtin_bit_cast(half8, xlo_);
        constexpr int kk = decltype(KK)::value, q = decltype(Q)::value;
        constexpr bool reload = !(kk == 1 && q == 1);            // (1, 1): the next fragments belong to the other stage
        constexpr int nkk = q == 1 ? kk + 1 : kk, nq = q ^ 1;
        if constexpr ((ABL & 4) != 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[q * 4][e] += __uint_as_float(xhi_[e] ^ xlo_[e]);   // keep the operands alive
          return;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[q * 4 + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], xhi, acc[q * 4 + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[q * 4 + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], xlo, acc[q * 4 + t], 0, 0, 0);
          if constexpr (reload) wh[t] = *reinterpret_cast<lds_half8*>(smem3 + frag_addr(nkk, nq * 4 + t));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[q * 4 + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t], xhi, acc[q * 4 + t], 0, 0, 0);
          if constexpr (reload) wl[t] = *reinterpret_cast<lds_half8*>(smem3 + (frag_addr(nkk, nq * 4 + t) ^ 64));
        }
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      if constexpr ((ABL & 4) == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          wh[t] = *reinterpret_cast<lds_half8*>(smem3 + frag_addr(0, t));
          wl[t] = *reinterpret_cast<lds_half8*>(smem3 + (frag_addr(0, t) ^ 64));
        }
      }
      f32x4 cq[4];
      // (the fragment reads above are in flight while this prologue runs)
      const bool pre = tp.far;                      // wave-uniform: this tap's operands come from global memory, both sub-steps
      if (pre) {                                    // rare, and nothing hides it: four dependent L2 round trips
        const float2 off = *reinterpret_cast<const float2*>(offp + (g * 9 + tap) * 2);
        corners_global(g, ch, tap, off, 0, 0, cq);
        blend_split(tp, cq, 0, xa_hi, xa_lo);
        corners_global(g, ch, tap, off, 0, 1, cq);
        blend_split(tp, cq, 1, xa_hi, xa_lo);
        corners_global(g, ch, tap, off, 1, 0, cq);
        blend_split(tp, cq, 0, xb_hi, xb_lo);
        corners_global(g, ch, tap, off, 1, 1, cq);
        blend_split(tp, cq, 1, xb_hi, xb_lo);
      } else if (!have_x) {                         // first step of a window: nothing hid this blend
        corners_lds(tp, 0, 0, cq);
        blend_split(tp, cq, 0, xa_hi, xa_lo);
        corners_lds(tp, 0, 1, cq);
        blend_split(tp, cq, 1, xa_hi, xa_lo);
      }
      // ---- the step proper: ONE straight-line body (no branch: the scheduler interleaves VALU / LDS / MFMA freely and every
      // s_waitcnt is exact).  Under sub-step 0's MFMAs the operand of sub-step 1 is blended (kept only when !pre: a select
      // per register, the window reads of a far tap are clamped and harmless); under sub-step 1's the NEXT tap is set up and
      // its first operand blended (in-bounds garbage when the next step reads another window or does not exist: have_x).
      if constexpr ((ABL & 1) != 0) {                // ablation: the MFMA / fragment side alone
        xb_hi = xa_hi, xb_lo = xa_lo;
        quad(I0{}, I0{}, xa_hi, xa_lo);
        quad(I0{}, I1{}, xa_hi, xa_lo);
        quad(I1{}, I0{}, xb_hi, xb_lo);
        quad(I1{}, I1{}, xb_hi, xb_lo);
      } else {
        u32x4 nb_hi, nb_lo;
        corners_lds(tp, 1, 0, cq);
        quad(I0{}, I0{}, xa_hi, xa_lo);
        blend_split(tp, cq, 0, nb_hi, nb_lo);
        corners_lds(tp, 1, 1, cq);
        quad(I0{}, I1{}, xa_hi, xa_lo);
        blend_split(tp, cq, 1, nb_hi, nb_lo);
  #pragma unroll
        for (int e = 0; e < 4; ++e) {
          xb_hi[e] = pre ? xb_hi[e] : nb_hi[e];
          xb_lo[e] = pre ? xb_lo[e] : nb_lo[e];
        }
        setup(tp, g1, ch1, tap1, off_nx);             // tp is dead: the next tap
        corners_lds(tp, 0, 0, cq);
        quad(I1{}, I0{}, xb_hi, xb_lo);
        blend_split(tp, cq, 0, xa_hi, xa_lo);
        corners_lds(tp, 0, 1, cq);
        quad(I1{}, I1{}, xb_hi, xb_lo);
        blend_split(tp, cq, 1, xa_hi, xa_lo);
      }
      have_x = more && tap1 != 0;                   // (a far next tap recomputes its operands anyway)
    }
    if (tap1 == 0 && more) {                     // the next K step reads another window: every wave is done with this one
      __syncthreads();
      dma_patch(g1, ch1);
    }
    __syncthreads();                             // drains the DMA queue (vmcnt(0)) and fences the buffers
    g = g1, ch = ch1, tap = tap1;
  }

  // ---- epilogue (the register epilogue of conv3x3_patch.hip): lanes i / i+32 swap 4-cout groups -> 8 consecutive couts
  unsigned long long* gn_bins = reinterpret_cast<unsigned long long*>(smem);   // [256/8][2]; the K loop's last barrier freed the LDS
  const bool gn = a.gn_stats != nullptr;
  if (gn) {
    if (tid < 64) gn_bins[tid] = 0ull;
    __syncthreads();
  }
  float gpart[32];                                  // (sum, sum of squares) of this position's 8 couts, per (tc, qp)
#pragma unroll
  for (int tc = 0; tc < 8; ++tc) {
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t lo = __float_as_uint(acc[tc][4 * (2 * qp) + e]);
        const uint32_t hi = __float_as_uint(acc[tc][4 * (2 * qp + 1) + e]);
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        v[e] = __uint_as_float(r[0]);
        v[4 + e] = __uint_as_float(r[1]);
      }
      const int cl = tc * 32 + 8 * (2 * qp + khalf);
      const int c0 = nt * DX_BCO + cl;
      const bool live = pvalid && c0 < a.cout;
      if (live) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= a.acc_scale;
        if (a.bias != nullptr) {
          const float4 b0 = *reinterpret_cast<const float4*>(a.bias + c0);
          const float4 b1 = *reinterpret_cast<const float4*>(a.bias + c0 + 4);
          v[0] += b0.x, v[1] += b0.y, v[2] += b0.z, v[3] += b0.w;
          v[4] += b1.x, v[5] += b1.y, v[6] += b1.z, v[7] += b1.w;
        }
      }
      if (gn) {
        float gs = 0.f, gss = 0.f;
        if (live) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            gs += v[e];
            gss = __builtin_fmaf(v[e], v[e], gss);
          }
        }
        gpart[(tc * 2 + qp) * 2 + 0] = gs;
        gpart[(tc * 2 + qp) * 2 + 1] = gss;
      }
      if (!live) continue;
      if (a.flags & SM_CONV_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      float* yp = a.y + orow * a.out_cstride + a.out_coff + c0;
      *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
  if (gn) {                                          // wave-uniform: the shuffles need every lane
    // the 32 totals of the wave's 32 positions (fixed butterfly order, common.h): lane l31 ends up with total
    // k = l31 = (tc * 2 + qp) * 2 + stat and rounds it ONCE to the fixed-point grid
    const float tot = gn_half_wave_totals<32>(gpart, l31);
    const int cl = (l31 >> 2) * 32 + 8 * (2 * ((l31 >> 1) & 1) + khalf);
    if (nt * DX_BCO + cl < a.cout) atomicAdd(&gn_bins[(cl >> 3) * 2 + (l31 & 1)], gn_fix(tot));
    __syncthreads();                                 // the whole tile lies in image n of level lev
    if (tid < 64) {
      const unsigned long long v = gn_bins[tid];
      const int gi = (nt * DX_BCO >> 3) + (tid >> 1);
      if (v != 0ull && gi < (a.cout >> 3))
        atomicAdd(a.gn_stats + (((long long)n * a.nlev + lev) * (a.cout >> 3) + gi) * 2 + (tid & 1), v);
    }
  }
}

bool dx3_ok(const sm_conv_desc* d) {
  if (!d || d->nlev < 1 || d->nlev > SM_MAX_LEVELS || d->batch < 1) return false;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->dil != 1) return false;
  if (d->deform_groups < 1 || d->cin != 64 * d->deform_groups || d->in_cstride % 4 != 0 || d->in_cstride < d->cin) return false;
  if (d->cout < 8 || d->cout_pad % DX_BCO != 0 || d->cout_pad < d->cout) return false;
  if ((d->cout & 7) || (d->out_cstride & 3) || (d->out_coff & 3) || d->out_coff + d->cout > d->out_cstride) return false;
  if (d->flags & (SM_CONV_RES_ADD | SM_CONV_RES_NEAREST | SM_CONV_IN_RELU | SM_CONV_RELU_NCH | SM_CONV_OUT_X3)) return false;
  if (d->scale_nch != 0 || d->w_batch_stride != 0 || d->w_level_stride != 0 || d->ngroups > 1) return false;
  for (int l = 0; l < d->nlev; ++l) {
    if (d->in_h[l] != d->out_h[l] || d->in_w[l] != d->out_w[l] || d->in_h[l] < 1 || d->in_w[l] < 1) return false;
    if ((long long)d->in_h[l] * d->in_w[l] > 0x3fffffffLL) return false;                  // 32-bit pixel indices
  }
  return true;
}

// tile list of one level (host): row tiles (8 x 32) over the columns [0, xb), then either row tiles or column tiles (32 x 8)
// over the strip [xb, W) -- whichever needs fewer blocks
struct LevelTiles {
  int ntx, nrow_t, xb, nbx, tpi;
};
LevelTiles level_tiles(int H, int W) {
  LevelTiles t;
  const int nty = sm_cdiv(H, 8);
  const int nfull = W / 32, rem = W - nfull * 32;
  const int nbx = sm_cdiv(rem, 8), nby = sm_cdiv(H, 32);
  const bool strip = rem > 0 && nbx * nby < nty;                 // the strip as column tiles
  t.ntx = strip ? nfull : nfull + (rem > 0 ? 1 : 0);
  t.nrow_t = t.ntx * nty;
  t.xb = nfull * 32;
  t.nbx = strip ? nbx : 1;
  t.tpi = t.nrow_t + (strip ? nbx * nby : 0);
  return t;
}

}  // namespace

// Which deformable convs sm_deform_conv2d_x3 takes: 3x3 / stride 1 / pad 1, 64 channels per deformable group, f32 rows in and
// out, plain epilogue (bias, ReLU, GroupNorm statistics).  1 / 0.
extern "C" int sm_deform_conv2d_x3_supported(const sm_conv_desc* d) { return dx3_ok(d) ? 1 : 0; }

// Host-side query (no GPU): out4 = {blocks, row tiles, column tiles, LDS window pixels}
extern "C" int sm_deform_conv2d_x3_plan(const sm_conv_desc* d, int64_t* out4) {
  if (!d || !out4) return SM_ERR_BAD_ARG;
  if (!dx3_ok(d)) return SM_ERR_UNSUPPORTED;
  long long rows = 0, cols = 0;
  for (int l = 0; l < d->nlev; ++l) {
    const LevelTiles t = level_tiles(d->in_h[l], d->in_w[l]);
    rows += (long long)d->batch * t.nrow_t;
    cols += (long long)d->batch * (t.tpi - t.nrow_t);
  }
  out4[0] = (rows + cols) * (d->cout_pad / DX_BCO);
  out4[1] = rows;
  out4[2] = cols;
  out4[3] = DX_PPIX;
  return SM_OK;
}

extern "C" int sm_deform_conv2d_x3(const sm_conv_desc* d, const float* x, const float* offset, const void* w_split,
                                   const float* bias, float* y, int64_t* gn_stats, sm_stream_t stream) {
  if (!x || !offset || !w_split || !y) return SM_ERR_BAD_ARG;
  if (!dx3_ok(d)) return SM_ERR_UNSUPPORTED;
  DeformX3Args a;
  a.x = x;
  a.w = (const uint16_t*)w_split;
  a.bias = bias;
  a.offset = offset;
  a.y = y;
  a.gn_stats = reinterpret_cast<unsigned long long*>(gn_stats);
  a.nlev = d->nlev;
  a.batch = d->batch;
  long long t = 0;
  for (int l = 0; l < SM_MAX_LEVELS; ++l) {
    const bool on = l < d->nlev;
    a.h[l] = on ? d->in_h[l] : 1;
    a.w_[l] = on ? d->in_w[l] : 1;
    a.in_row0[l] = on ? d->in_row0[l] : 0;
    a.out_row0[l] = on ? d->out_row0[l] : 0;
    const LevelTiles lt = level_tiles(a.h[l], a.w_[l]);
    a.ntx[l] = lt.ntx > 0 ? lt.ntx : 1;
    a.nrow_t[l] = lt.nrow_t;
    a.xb[l] = lt.xb;
    a.nbx[l] = lt.nbx;
    a.tpi[l] = lt.tpi;
    a.tile0[l] = (int)t;
    if (on) t += (long long)d->batch * lt.tpi;
  }
  a.tile0[SM_MAX_LEVELS] = (int)t;
  a.cout = d->cout;
  a.ntn = d->cout_pad / DX_BCO;
  a.dg = d->deform_groups;
  a.in_cstride = d->in_cstride;
  a.out_cstride = d->out_cstride;
  a.out_coff = d->out_coff;
  a.flags = d->flags;
  a.acc_scale = (d->acc_scale == 0.f) ? 1.f : d->acc_scale;
  const long long nblk = t * a.ntn;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return SM_ERR_BAD_SHAPE;
  a.nblk = (int)nblk;
  hipStream_t s = sm_hip_stream(stream);
  if (gn_stats != nullptr &&
      sm_zero_async(gn_stats, sizeof(unsigned long long) * 2 * d->batch * d->nlev * (d->cout / 8), s) != hipSuccess)
    return SM_ERR_LAUNCH;
  const void* kern = (const void*)deform_patch_x3_kernel<0>;
#ifdef SM_EXPERIMENTS
  switch (((d->flags & SM_CONV_DBG_DX3_NO_BLEND) ? 1 : 0) | ((d->flags & SM_CONV_DBG_PATCH_NO_DMA) ? 2 : 0) |
          ((d->flags & SM_CONV_DBG_PATCH_NO_MFMA) ? 4 : 0)) {
    case 1: kern = (const void*)deform_patch_x3_kernel<1>; break;
    case 2: kern = (const void*)deform_patch_x3_kernel<2>; break;
    case 3: kern = (const void*)deform_patch_x3_kernel<3>; break;
    case 4: kern = (const void*)deform_patch_x3_kernel<4>; break;
    case 6: kern = (const void*)deform_patch_x3_kernel<6>; break;
    default: break;
  }
#endif
  if (sm_lds_optin(kern, DX_LDS) != hipSuccess) return SM_ERR_LAUNCH;
  void* kargs[] = {(void*)&a};
  if (hipLaunchKernel(kern, dim3((unsigned)nblk), dim3(DX_THREADS), kargs, DX_LDS, s) != hipSuccess) return SM_ERR_LAUNCH;
  SM_LAUNCH_CHECK();
  return SM_OK;
}
