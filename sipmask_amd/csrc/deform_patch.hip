// Deformable 3x3 convolution (FeatureAlign, V/mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:85-115,191-243 semantics) with
// the INPUT PATCH of one deformable group resident in LDS (gfx950).
//
// Why a second deformable kernel.  conv_igemm.hip's DEFORM loader gathers the four bilinear corners of every (position,
// tap, 8-channel chunk) from global memory: 4 x 128-byte lines per (position, tap, group), i.e. 36 line requests per
// position and group for ~16 distinct pixels, each behind an L2 round trip, at one block per CU -- the launch runs at
// 290 TFLOP/s, latency-bound.  The offsets of FeatureAlign are a 1x1 conv of the box prediction:
// a few pixels.  So: a block owns an 8 x 32 tile of output positions of one image, DMAs the (8 + 2 + 2R) x (32 + 2 + 2R)
// pixel window of ONE deformable group (64 channels = one 128-byte line per pixel, R = 3) into LDS once and serves all
// nine taps' corner reads from there (ds_read_b128, ~100 cycles instead of ~2000).  A corner that falls outside the window
// (|offset| > R) sends that wave through the global gather for that tap (wave-uniform branch): any offset is handled,
// only the common case is fast.
//
// Layout of the work.  256 couts x 256 positions per block, 8 waves, wave w = output row y0 + w, lane & 31 = column,
// lane >> 5 = K half: every lane blends ITS OWN MFMA B operand (8 channels of its position) in registers -- there is no
// column tile in LDS and no second barrier -- and multiplies it against all 256 couts (8 MFMA tiles of 32 x 32 x 16).
// K order = (group, tap, 64 channels); a K step's weights are 128 contiguous bytes of a cout row in the ordinary
// [cout][tap][cin] operand layout, DMAed as whole lines (8 rows per piece) into a double-buffered [256][128 B] stage.
// LDS: 2 x 32 KB weights + 80 KB patch (640 pixels x 128 B); the patch is reloaded per group (4 x per tile, exposed).
// Arithmetic of the bilinear sample is the DEFORM loader's, expression for expression (f32 blend, one rounding to the bf16
// MFMA operand).
#include <utility>

#include "common.h"
#include "experiments.h"

namespace {

// Two tile shapes in one launch (round 5, as csrc/deform_patch_x3.hip): 8 rows x 32 columns, and 32 rows x 8 columns for the
// strip a level's width leaves beyond a multiple of 32 (168 = 5 x 32 + 8: thirteen row tiles with 24 of 32 columns empty
// become four column tiles) -- 392 instead of 440 tiles per four 800 x 1344 images.  Both windows are 640 pixels.
constexpr int DP_R = 3;
constexpr int DP_PPIX = 640;                      // window pixels: 16 x 40 (row tile) or 40 x 16 (column tile) = 80 DMA pieces of 8
constexpr int DP_PPW = DP_PPIX / 8 / 8;           // patch pieces per wave (10)
constexpr int DP_BCO = 256;
constexpr int DP_WSTAGE = DP_BCO * 128;           // one K step of weights: 256 cout rows x 64 channels
constexpr int DP_THREADS = 512;
constexpr int DP_LDS = 2 * DP_WSTAGE + DP_PPIX * 128;

struct DeformPatchArgs {
  const uint16_t* x;
  const uint16_t* w;
  const float* bias;
  const float* offset;
  void* y;
  unsigned long long* gn_stats;   // fixed point (common.h: gn_fix)
  int nlev, batch;
  int h[SM_MAX_LEVELS], w_[SM_MAX_LEVELS];
  long long in_row0[SM_MAX_LEVELS], out_row0[SM_MAX_LEVELS];
  int tile0[SM_MAX_LEVELS + 1];   // first position tile of each level
  int ntx[SM_MAX_LEVELS];         // row tiles (8 x 32) along x
  int tpi[SM_MAX_LEVELS];         // tiles per image
  int nrow_t[SM_MAX_LEVELS];      // row tiles per image: the first nrow_t of an image's tiles
  int xb[SM_MAX_LEVELS];          // first column of the column-tile strip
  int nbx[SM_MAX_LEVELS];         // column tiles (32 x 8) along x inside the strip
  int cin, cout, ntn, dg;
  int in_cstride, out_cstride, out_coff;
  long long Kp;
  unsigned flags;
  int scale_nch;
  float level_scale[SM_MAX_LEVELS];
  int nblk;
};

template <int N, typename F, int... Is>
__device__ __forceinline__ void dfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void dfor(F&& f) {
  dfor_impl<N>(f, std::make_integer_sequence<int, N>{});
}

__device__ __attribute__((aligned(16))) const unsigned int g_zero16d[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ int dp_xcd_tile(int b, int nblk) {
  const int xcd = b & 7, xq = nblk >> 3, xr = nblk & 7;
  return (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
}

// ABL: ablations for the micro-benchmark (wrong results by construction): 2 no blend (the raw corners are the operand),
// 4 no DMA in the K loop.  Instantiated only by `make EXPERIMENTS=1` (then selected by SIPMASK_DEFORM_ABLATE=n);
// the default library contains deform_patch_kernel<0> alone.
template <int ABL>
__global__ __launch_bounds__(DP_THREADS, 1) void deform_patch_kernel(const DeformPatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [W stage 0][W stage 1][patch]
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  typedef __attribute__((address_space(3))) unsigned char lds_u8;      // typed LDS reads: ds_read_b128, never flat_load
  typedef const __attribute__((address_space(3))) u32x4 lds_u32x4;
  typedef const __attribute__((address_space(3))) bf16x8 lds_bf16x8;
  lds_u8* const smem3 = (lds_u8*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;

  // ---- tile decode (wave-uniform)
  const int tlin = dp_xcd_tile(blockIdx.x, a.nblk);
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
  const int DP_PW = (col_tile ? 8 : 32) + 2 + 2 * DP_R;           // window pitch in pixels: 16 / 40 (even: swizzle parity)
  const int DP_PH = DP_PPIX / DP_PW;                              // 40 / 16
  const int py0 = y0 - 1 - DP_R, px0 = x0 - 1 - DP_R;           // image coordinates of patch pixel (0, 0)
  const long long img_row0 = a.in_row0[lev] + (long long)n * H * W;
  const uint16_t* const ximg = a.x + img_row0 * a.in_cstride;

  unsigned char* const Wb0 = smem;
  unsigned char* const Pb = smem + 2 * DP_WSTAGE;

  // ---- loader state.  Every DMA piece is 8 rows x 128 B: lane L -> row (L >> 3), physical 16-byte slot (L & 7), which
  // holds the logical chunk (L & 7) ^ ((row >> 1) & 7) of the row (swizzle on the source side).
  const unsigned long long zero_page = (unsigned long long)g_zero16d;
  auto dma_patch = [&](int g) {                                   // group g's window -> Pb (pixel addresses recomputed:
    dfor<DP_PPW>([&](auto I) {                                    // 4 x per tile, cheaper than 10 live registers)
      constexpr int i = decltype(I)::value;
      const int pix = (wave + 8 * i) * 8 + (lane >> 3);
      const int pr = col_tile ? (pix >> 4) : (pix / 40), pc = pix - pr * DP_PW;
      int py0_l = py0;
      asm volatile("" : "+s"(py0_l));                             // opaque: keeps the 10 addresses out of the K loop's registers
      const int ih = py0_l + pr, iw = px0 + pc;
      const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
      const int chunk = (lane & 7) ^ ((pix >> 1) & 7);
      const unsigned long long pm = ok ? ~0ull : 0ull;
      const unsigned long long src =
          ((unsigned long long)(ximg + ((ih * W + iw) * a.in_cstride + chunk * 8 + g * 64)) & pm) | (zero_page & ~pm);
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(Pb + (wave + 8 * i) * 1024), 16, 0, 0);
    });
  };
  // weights: uniform base (SGPRs) + one per-lane 32-bit offset.  Piece i of this wave = rows wave*32 + 8i .. +7; lane L ->
  // row r = L >> 3 of the piece, physical slot L & 7 = logical chunk (L & 7) ^ ((row >> 1) & 7) = (L & 7) ^ (r >> 1) ^ 4*(i & 1)
  const unsigned wvoff0 = (unsigned)((lane >> 3) * (int)a.Kp + (((lane & 7) ^ ((lane >> 4) & 3)) * 8));   // Kp % 64 == 0
  const unsigned wvoff1 = wvoff0 ^ 32u;
  auto dma_w = [&](int g, int tap, int buf) {                    // K step (g, tap) -> Wb[buf]
    unsigned char* dst = Wb0 + buf * DP_WSTAGE;
    const uint16_t* const ubase = a.w + ((long long)(nt * DP_BCO + wave * 32) * a.Kp + tap * a.cin + g * 64);
    dfor<4>([&](auto I) {
      constexpr int i = decltype(I)::value;
      __builtin_amdgcn_global_load_lds((glb_void*)(ubase + (long long)(8 * i) * a.Kp + ((i & 1) ? wvoff1 : wvoff0)),
                                       (lds_void*)(dst + (wave * 4 + i) * 1024), 16, 0, 0);
    });
  };

  // ---- this lane's output position
  const int wy = wave << (5 - lgw);                              // first tile row of this wave
  const int oy = y0 + wy + (l31 >> lgw), ox = x0 + (l31 & ((1 << lgw) - 1));
  const bool row_live = y0 + wy < H;                             // wave-uniform
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

  // ---- per-tap state of this lane's position: corner weights, corner addresses (LDS window and global), fallback flag
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 w1v, w2v, w3v, w4v;
  int c1, c2, c3, c4;            // LDS byte addresses of the four corners, K chunk khalf (sub-step kk: ^ (kk * 32))
  int hlo, wlo, gofs;            // the sample's top-left pixel in the image and the channel offset (fallback only)
  bool far;                      // wave-uniform: some lane samples outside the LDS window
  // the DEFORM loader of conv_igemm.hip with rhi = oy - 1, rwi = ox - 1 (deform_conv_cuda_kernel.cu:85-115,216-229)
  auto setup = [&](int g_, int tap_, float2 off) {
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
    const float w1 = (inr && t_ok && l_ok) ? hh * hw : 0.f;
    const float w2 = (inr && t_ok && r_ok) ? hh * lw : 0.f;
    const float w3 = (inr && b_ok && l_ok) ? lh * hw : 0.f;
    const float w4 = (inr && b_ok && r_ok) ? lh * lw : 0.f;
    w1v = f32x2{w1, w1}, w2v = f32x2{w2, w2}, w3v = f32x2{w3, w3}, w4v = f32x2{w4, w4};
    // corners in window coordinates; a lane whose sample lies outside the image has zero weights and may read anywhere
    const int pr = h_low - py0, pc = w_low - px0;
    const bool in_patch = pr >= 0 && pr <= DP_PH - 2 && pc >= 0 && pc <= DP_PW - 2;
    far = __builtin_amdgcn_ballot_w64(inr && !in_patch) != 0ull;
    // pixel p, chunk j -> byte p * 128 + ((j ^ ((p >> 1) & 7)) * 16); p and p + DP_PW have the same parity
    const int prc = min(max(pr, 0), DP_PH - 2), pcc = min(max(pc, 0), DP_PW - 2);
    const int p1 = prc * DP_PW + pcc;
    const int a1 = 2 * DP_WSTAGE + p1 * 128 + khalf * 16;
    c1 = a1 ^ (((p1 >> 1) & 7) * 16);
    c2 = (a1 + 128) ^ ((((p1 + 1) >> 1) & 7) * 16);
    c3 = (a1 + DP_PW * 128) ^ ((((p1 + DP_PW) >> 1) & 7) * 16);
    c4 = (a1 + DP_PW * 128 + 128) ^ ((((p1 + DP_PW + 1) >> 1) & 7) * 16);
    hlo = h_low, wlo = w_low, gofs = g_ * 64 + khalf * 8;
  };
  // corners of K sub-step kk.  Fallback arm: this tap's corners come from global memory for the whole wave (clamped
  // addresses, the weights carry the zero padding); the empty asm makes its results register-defined, so that the code
  // after the join never waits on vmcnt -- that counter is in order, and the next K step's weight DMA is in flight on it.
  // (Round 5 tried the far tap as a step of its own, as csrc/deform_patch_x3.hip handles it: a second body that touches the
  // 128 accumulators -- even a rolled four-iteration loop -- makes hipcc spill 255 registers; the arm stays.)
  u32x4 q1, q2, q3, q4;
  auto corners = [&](int kk) {
    // the window reads are unconditional (clamped addresses): outside any branch the LDS counter stays exact, and the wait
    // for the weight fragments requested before them is lgkmcnt(4), not (0)
    q1 = *reinterpret_cast<const lds_u32x4*>(smem3 + (c1 ^ (kk * 32)));
    q2 = *reinterpret_cast<const lds_u32x4*>(smem3 + (c2 ^ (kk * 32)));
    q3 = *reinterpret_cast<const lds_u32x4*>(smem3 + (c3 ^ (kk * 32)));
    q4 = *reinterpret_cast<const lds_u32x4*>(smem3 + (c4 ^ (kk * 32)));
    if (far) {
      int W_l = W;
      asm volatile("" : "+s"(W_l));                       // opaque: hipcc otherwise speculates this arm's address arithmetic
      const int hl = min(max(hlo, 0), H - 1), hh_ = min(max(hlo + 1, 0), H - 1);       // (12 v_mul_lo_u32) into every step
      const int wl = min(max(wlo, 0), W_l - 1), wh_ = min(max(wlo + 1, 0), W_l - 1);
      const uint16_t* const gp = ximg + gofs + kk * 16;
      q1 = *reinterpret_cast<const u32x4*>(gp + (hl * W_l + wl) * a.in_cstride);
      q2 = *reinterpret_cast<const u32x4*>(gp + (hl * W_l + wh_) * a.in_cstride);
      q3 = *reinterpret_cast<const u32x4*>(gp + (hh_ * W_l + wl) * a.in_cstride);
      q4 = *reinterpret_cast<const u32x4*>(gp + (hh_ * W_l + wh_) * a.in_cstride);
      asm volatile("" : "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4));
    }
  };
  // blend of 8 channels, two at a time (v_pk_mul_f32 / v_pk_fma_f32): per channel ((w1*f1 + w2*f2) + w3*f3) + w4*f4 with
  // the products contracted into FMAs, the scalar expression of conv_igemm.hip's DEFORM loader
  auto blend = [&]() -> bf16x8 {
    u32x4 xq;
    if constexpr (ABL == 2) {
      xq = q1 ^ q2 ^ q3 ^ q4;
    } else {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const f32x2 v1 = {__uint_as_float(q1[d] << 16), __uint_as_float(q1[d] & 0xffff0000u)};
        const f32x2 v2 = {__uint_as_float(q2[d] << 16), __uint_as_float(q2[d] & 0xffff0000u)};
        const f32x2 v3 = {__uint_as_float(q3[d] << 16), __uint_as_float(q3[d] & 0xffff0000u)};
        const f32x2 v4 = {__uint_as_float(q4[d] << 16), __uint_as_float(q4[d] & 0xffff0000u)};
        const f32x2 r = w1v * v1 + w2v * v2 + w3v * v3 + w4v * v4;
        xq[d] = pack_bf16x2(r[0], r[1]);
      }
    }
    return *reinterpret_cast<const bf16x8*>(&xq);
  };

  dma_patch(0);
  dma_w(0, 0, 0);
  float2 off_nx = *reinterpret_cast<const float2*>(offp);
  __syncthreads();

  // K loop.  Step s = (group g, tap): [weight DMA of s+1 and offsets of s+1 requested] [set-up of this tap, corners and
  // fragments of sub-step 0, blend 0] then per 16-channel sub-step kk: [8 MFMAs of kk interleaved with the blend of kk+1]
  // [fragments of kk+1 and corners of kk+2 requested].
  // (A/B, round 2: setting up step s+1 and requesting its first corners behind the last sub-step of s -- so that a step does
  // not open with an exposed LDS round trip -- measured 12 % SLOWER, 0.128 vs 0.114 ms on the B=2 head; not kept.)
  const int nstep = a.dg * 9;
  int g = 0, tap = 0;
  for (int s = 0; s < nstep; ++s) {
    int g1 = g, tap1 = tap + 1;
    if (tap1 == 9) {
      tap1 = 0;
      ++g1;
    }
    const bool more = s + 1 < nstep;
    if (more && ABL != 4) dma_w(g1, tap1, (s + 1) & 1);
    const float2 off = off_nx;                   // requested one step ago: the barrier that closed that step covered it
    off_nx = *reinterpret_cast<const float2*>(offp + (more ? (s + 1) * 2 : 0));   // unconditional: no exec-masked load
    if (row_live) {
      setup(g, tap, off);
      const int wbase = (s & 1) * DP_WSTAGE + wrow_off + ((khalf ^ rs8) * 16);   // K chunk kk * 2 + khalf: ^ (kk * 32)
      bf16x8 wfr[8];
      auto fragments = [&](int kk) {
#pragma unroll
        for (int tc = 0; tc < 8; ++tc)
          wfr[tc] = *reinterpret_cast<const lds_bf16x8*>(smem3 + ((wbase ^ (kk * 32)) + tc * 32 * 128));
      };
      corners(0);
      fragments(0);
      bf16x8 xf = blend();                                      // sub-step 0's blend is the one nothing hides
      __builtin_amdgcn_sched_barrier(0);
      corners(1);
      dfor<4>([&](auto KK) {
        constexpr int kk = decltype(KK)::value;
        // one scheduling region: the 8 MFMAs of kk and the blend of kk+1 (~65 VALU), one MFMA then 8 VALU at a time -- a
        // 32x32x16 MFMA occupies the matrix pipe for 32 cycles, 8 wave64 VALU instructions the vector pipe for as long, so
        // the wave keeps both busy by itself instead of relying on the SIMD's other wave being out of phase (it is not:
        // the barrier of every step re-aligns them, and measured the phases simply added up)
        bf16x8 xf_next = xf;
        if constexpr (kk < 3) xf_next = blend();
#pragma unroll
        for (int tc = 0; tc < 8; ++tc) {
          acc[tc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[tc], xf, acc[tc], 0, 0, 0);
          // rolling fragment prefetch: the MFMA has read wfr[tc] when it issues, so sub-step kk+1's fragment goes into the same
          // registers right behind it and lands while the other MFMAs run -- the LDS phase of kk+1 under the MFMA phase of kk
          if constexpr (kk < 3)
            wfr[tc] = *reinterpret_cast<const lds_bf16x8*>(smem3 + ((wbase ^ ((kk + 1) * 32)) + tc * 32 * 128));
        }
        if constexpr (kk < 3) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kk < 2) corners(kk + 2);
        __builtin_amdgcn_sched_barrier(0);
        xf = xf_next;
      });
    }
    if (tap1 == 0 && more) {                     // next K step starts a new group: every wave is done with the window
      __syncthreads();
      if constexpr (ABL != 4) dma_patch(g1);
    }
    __syncthreads();                             // drains the DMA queue (vmcnt(0)) and fences the buffers
    g = g1;
    tap = tap1;
  }

  // ---- epilogue (the register epilogue of conv3x3_patch.hip): lanes i / i+32 swap 4-cout groups -> 8 consecutive couts
  const float lscale = a.level_scale[lev];
  const bool out_f32 = a.flags & SM_CONV_OUT_F32;
  unsigned long long* gn_bins = reinterpret_cast<unsigned long long*>(smem);                  // [256/8][2]; the K loop's last barrier freed the LDS
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
      const int c0 = nt * DP_BCO + cl;
      const bool live = pvalid && c0 < a.cout;
      if (live) {
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
  if (gn) {                                          // wave-uniform: the shuffles need every lane
    // the 32 totals of the wave's 32 positions (fixed butterfly order, common.h): lane l31 ends up with total
    // k = l31 = (tc * 2 + qp) * 2 + stat and rounds it ONCE to the fixed-point grid
    const float tot = gn_half_wave_totals<32>(gpart, l31);
    const int cl = (l31 >> 2) * 32 + 8 * (2 * ((l31 >> 1) & 1) + khalf);
    if (nt * DP_BCO + cl < a.cout) atomicAdd(&gn_bins[(cl >> 3) * 2 + (l31 & 1)], gn_fix(tot));
  }
  if (gn) {                                          // the whole tile lies in image n of level lev
    __syncthreads();
    if (tid < 64) {
      const unsigned long long v = gn_bins[tid];
      const int gi = (nt * DP_BCO >> 3) + (tid >> 1);
      if (v != 0ull && gi < (a.cout >> 3))
        atomicAdd(a.gn_stats + (((long long)n * a.nlev + lev) * (a.cout >> 3) + gi) * 2 + (tid & 1), v);
    }
  }
}

}  // namespace

// tile list of one level (host; the rule of csrc/deform_patch_x3.hip): row tiles (8 x 32) over the columns [0, xb), then either
// row tiles or column tiles (32 x 8) over the strip [xb, W) -- whichever needs fewer blocks
namespace {
struct DpLevelTiles {
  int ntx, nrow_t, xb, nbx, tpi;
};
DpLevelTiles dp_level_tiles(int H, int W) {
  DpLevelTiles t;
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

// Which deformable convs the LDS-patch kernel takes (conv_igemm.hip's launch_conv<true> asks): 3x3 / stride 1 / pad 1,
// 64 channels per deformable group, 256-cout weight tiles, plain epilogue.  Everything else stays on the gather loader.
bool sm_deform_patch_supported(const sm_conv_desc* d) {
  if (!d || d->nlev < 1 || d->nlev > SM_MAX_LEVELS || d->batch < 1) return false;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->dil != 1) return false;
  if (d->deform_groups < 1 || d->cin != 64 * d->deform_groups || d->in_cstride % 8 != 0) return false;
  if (d->cout < 1 || d->cout_pad % DP_BCO != 0 || d->cout_pad < d->cout) return false;
  if ((d->cout & 7) || (d->out_cstride & 7) || (d->out_coff & 7)) return false;
  if (d->flags & (SM_CONV_RES_ADD | SM_CONV_RES_NEAREST | SM_CONV_IN_RELU)) return false;
  if (d->w_batch_stride != 0 || d->ngroups > 1) return false;
  for (int l = 0; l < d->nlev; ++l) {
    if (d->in_h[l] != d->out_h[l] || d->in_w[l] != d->out_w[l] || d->in_h[l] < 1 || d->in_w[l] < 1) return false;
    if ((long long)d->in_h[l] * d->in_w[l] * d->in_cstride > 0x7fffffffLL) return false;   // 32-bit pixel offsets
  }
  return true;
}

// Host-side query (no GPU): does sm_deform_conv2d / sm_conv2d_gn_stats run this deformable conv on the LDS-window kernel?
// SM_OK and out4 = {blocks, tile rows, tile columns, LDS window pixels per deformable group}, or SM_ERR_UNSUPPORTED when the
// gather loader of conv_igemm.hip takes it (other shapes, SM_CONV_DBG_DEFORM_GATHER).
extern "C" int sm_deform_conv_window_plan(const sm_conv_desc* d, int64_t* out4) {
  if (!d || !out4) return SM_ERR_BAD_ARG;
  if ((d->flags & SM_CONV_DBG_DEFORM_GATHER) || !sm_deform_patch_supported(d)) return SM_ERR_UNSUPPORTED;
  long long t = 0;
  for (int l = 0; l < d->nlev; ++l) t += (long long)d->batch * dp_level_tiles(d->in_h[l], d->in_w[l]).tpi;
  out4[0] = t * (d->cout_pad / DP_BCO);
  out4[1] = 8;                         // (row tiles; the right-hand strip of a level may run as 32 x 8 column tiles)
  out4[2] = 32;
  out4[3] = DP_PPIX;
  return SM_OK;
}

// k_padded: row pitch (elements) of the [cout_pad][k_padded] weight operand (sm_conv_plan.k_padded); gn_stats is zeroed by
// the caller (launch_conv)
int sm_deform_patch_launch(const sm_conv_desc* d, const void* x, const float* offset, const void* w, const float* bias,
                           void* y, hipStream_t stream, unsigned long long* gn_stats, long long k_padded) {
  if (!sm_deform_patch_supported(d)) return SM_ERR_UNSUPPORTED;
  if (gn_stats != nullptr && (d->flags & SM_CONV_OUT_F32)) return SM_ERR_UNSUPPORTED;
  DeformPatchArgs a;
  a.x = (const uint16_t*)x;
  a.w = (const uint16_t*)w;
  a.bias = bias;
  a.offset = offset;
  a.y = y;
  a.gn_stats = gn_stats;
  a.nlev = d->nlev;
  a.batch = d->batch;
  int t = 0;
  for (int l = 0; l < SM_MAX_LEVELS; ++l) {
    const bool on = l < d->nlev;
    a.h[l] = on ? d->in_h[l] : 1;
    a.w_[l] = on ? d->in_w[l] : 1;
    a.in_row0[l] = on ? d->in_row0[l] : 0;
    a.out_row0[l] = on ? d->out_row0[l] : 0;
    a.level_scale[l] = on ? d->level_scale[l] : 1.f;
    const DpLevelTiles lt = dp_level_tiles(a.h[l], a.w_[l]);
    a.ntx[l] = lt.ntx > 0 ? lt.ntx : 1;
    a.nrow_t[l] = lt.nrow_t;
    a.xb[l] = lt.xb;
    a.nbx[l] = lt.nbx;
    a.tpi[l] = lt.tpi;
    a.tile0[l] = t;
    if (on) t += d->batch * a.tpi[l];
  }
  a.tile0[SM_MAX_LEVELS] = t;
  a.cin = d->cin;
  a.cout = d->cout;
  a.ntn = d->cout_pad / DP_BCO;
  a.dg = d->deform_groups;
  a.in_cstride = d->in_cstride;
  a.out_cstride = d->out_cstride;
  a.out_coff = d->out_coff;
  a.Kp = k_padded;
  a.flags = d->flags;
  a.scale_nch = d->scale_nch;
  const long long nblk = (long long)t * a.ntn;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return SM_ERR_BAD_SHAPE;
  a.nblk = (int)nblk;
  const void* kern = (const void*)deform_patch_kernel<0>;
#ifdef SM_EXPERIMENTS
  static const int ablate = sm_experiment_env("SIPMASK_DEFORM_ABLATE", 0);
  if (ablate == 2) kern = (const void*)deform_patch_kernel<2>;
  if (ablate == 4) kern = (const void*)deform_patch_kernel<4>;
#endif
  if (sm_lds_optin(kern, DP_LDS) != hipSuccess) return SM_ERR_LAUNCH;
  void* kargs[] = {(void*)&a};
  if (hipLaunchKernel(kern, dim3((unsigned)nblk), dim3(DP_THREADS), kargs, DP_LDS, stream) != hipSuccess) return SM_ERR_LAUNCH;
  SM_LAUNCH_CHECK();
  return SM_OK;
}
