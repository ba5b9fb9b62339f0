// ResNet bottleneck tail as ONE launch (gfx950): conv2 (3x3 / stride 1 / pad 1, C -> C, folded-BN bias, ReLU) chained
// with conv3 (1x1, C -> 4C, folded-BN bias, + identity, ReLU); optionally the NEXT block's conv1 (1x1, 4C -> C, bias,
// ReLU) chained behind it.  Reference: Bottleneck.forward, M/mmdet/models/backbones/resnet.py:167-200 (caffe style: the
// stride sits on conv1, so conv2 is always stride 1), BN frozen (resnet.py:370,514-521) and folded by the caller.
//
// Why.  layer1 / layer2 are bandwidth-shaped: conv3 moves 155 MB per B=2 launch for 4.4 GFLOP, and the C-channel tensor
// between conv2 and conv3 (and the 4C-channel tensor between conv3 and the next conv1) goes out to HBM and comes back
// for nothing.  Here a block computes the conv2 tile for 128 positions x ALL C channels; the MFMA C layout after the
// v_permlane32_swap step of the register epilogue (lane (p, h) holds 8 consecutive channels 16*kk + 8*h of position p)
// IS the B-operand layout of v_mfma_f32_32x32x16_bf16 for K step kk, so the bias+ReLU'd, bf16-rounded conv2 tile feeds
// conv3's MFMAs straight from registers: no LDS round trip, no barrier, and bit-identical arithmetic to the two-launch
// path (same K order, same rounding points).  The same holds between conv3's output and the next conv1.
// Every wave owns 32 positions for the whole chain; the 1x1 weights stream through LDS in 16 KB slices (LDS-DMA,
// ping-pong between a dedicated buffer and the finished K loop's stages), 128-byte rows with the conv_igemm swizzle.
//
// CDS > 0 (round 4): the FIRST block of layer1, whose shortcut is a 1x1 conv of the block input (resnet.py:453-469
// make_res_layer's downsample, stride 1 in layer1): that conv is one more stretch of K for conv3 -- the weight rows are
// [w3 | w_downsample] (K3 = C2 + CDS), the B fragments of the extra K steps are the block input's rows of this wave's 32
// positions, read straight from HBM into the fragment layout -- so the 4C-channel shortcut tensor is neither written
// (138 MB at four 800 x 1344 images) nor read back, and its launch disappears.  The shortcut stays f32 until the one
// rounding of the block output (the two-launch path rounds it to bf16 in between).
#include "common.h"
#include "experiments.h"

// ablation switches of the 1x1 pair's microbenchmark: compiled in with `make EXPERIMENTS=1` only
#ifdef SM_EXPERIMENTS
#define BT_DBG(a) ((a).dbg)
#else
#define BT_DBG(a) 0
#endif

namespace {

struct BtArgs {
  const uint16_t* x;     // conv2 input  [M][C]
  const uint16_t* w2;    // [C][9C]      (kh, kw, cin) with cin fastest
  const float* b2;
  const uint16_t* w3;    // [4C][C]
  const float* b3;
  const uint16_t* res;   // identity     [M][4C]   (CDS == 0)
  const uint16_t* xds;   // CDS > 0: the block input rows [batch * dsH * dsW][CDS] the shortcut conv reads (at stride ds_stride);
                         // w3 = [4C][C + CDS], b3 = b3 + b_downsample
  int ds_stride, dsH, dsW;
  int dbg;               // SM_EXPERIMENTS: ablation mask of the 1x1 pair (tools/r4 microbench); 0 in the library's own launches
  uint16_t* y;           // block output [M][4C]
  const uint16_t* w1n;   // next conv1   [C][4C] or null
  const float* b1n;
  uint16_t* t1n;         // next conv1 output [M][C]
  int batch, H, W, M;
};

__device__ __attribute__((aligned(16))) const unsigned int g_zero16b[4] = {0u, 0u, 0u, 0u};

constexpr int BT_BPOS = 128;
constexpr int BT_SLICE_BYTES = 16384;

// OCC: blocks per CU the register allocation aims for.  The unchained <64> kernel needs 135 VGPRs at OCC 3; capped at 128
// (OCC 4: four spills outside the loops) a fourth block fits, and these launches are bandwidth-shaped (layer1: 310 MB per
// 4-image launch): more loads in flight per CU.  (Round 4 A/B of a fourth block: neutral, removed.)
// SLB: bytes of a conv3 weight slice.  16 KB by default; the CHAINED variants take 8 KB -- half the couts per pass, so half
// the conv3 accumulators, residual chunks and chained-conv1 B fragments live at a time: 168 VGPRs + spills -> no spills at
// three blocks per CU (round 4; the chain was "neutral" in round 2 because of exactly that).
// CONV2 = false (round 4): no 3x3 conv in the launch -- conv3 (+ identity, ReLU) of one block chained with conv1 of the next,
// both 1x1: layer3's pairs (C2 = 256: 16 800 positions at four 800 x 1344 images, where each of the two launches is a
// 32-step K loop on 132-264 tiles that nothing hides, 0.042 + 0.033 ms, and the 1024-channel block output is read back).
template <int C2, bool CHAIN1, int OCC, int CDS = 0, int SLB = BT_SLICE_BYTES, bool CONV2 = true>
__global__ __launch_bounds__(256, OCC) void bottleneck_tail_kernel(const BtArgs a) {
  static_assert(CDS % 64 == 0, "shortcut input channels");
  constexpr int K3 = C2 + CDS;                  // conv3's K: the conv2 tile (+ the shortcut conv's input channels)
  constexpr int KKD = CDS / 16;
  constexpr int TCO = C2 / 32;                  // MFMA tiles along the conv2 couts (all of them in one wave)
  constexpr int NW = C2 / 64;                   // weight DMA instructions per thread per K step
  constexpr int NX = BT_BPOS / 64;
  constexpr int STAGE = (C2 + BT_BPOS) * 64;    // one K step of 32: [C2 weight rows | 128 position rows] x 64 B
  constexpr int KP2 = 9 * C2;
  constexpr int CPT = C2 / 8;                   // 16-byte chunks per tap
  constexpr int NK = 9 * C2 / 32;
  constexpr int C4 = 4 * C2;
  constexpr int SL = SLB / (K3 * 2);            // conv3 couts per slice (128 | 64 | 32)
  constexpr int W1B = C2 * SL * 2;              // chained conv1: a K slice of SL input channels, [C2 rows][SL k]
  constexpr int NR3 = SLB / 128, NR1 = W1B / 128; // 128-byte rows of a w3 / w1 slice
  static_assert(SL % 32 == 0 && (C2 + CDS) * SL * 2 == SLB && NR3 % 32 == 0 && NR1 % 32 == 0, "slice shape");
  static_assert(!CHAIN1 || SL % 64 == 0, "the chained conv1's K slice is laid out in 128-byte (64-channel) rows");
  constexpr int NPASS = C4 / SL;
  constexpr int CT = SL / 32;                   // MFMA tiles along the slice's couts
  constexpr int KK3 = C2 / 16;                  // K steps of 16 in conv3
  // chained conv1 of the next block: per pass a K slice of SL input channels, weights [C2 rows][SL k]
  constexpr int PRE = CONV2 ? 2 * STAGE : SLB;  // the conv2 K loop's two stages; afterwards (or without conv2) slice buffer B
  static_assert(PRE >= SLB, "the finished stages hold one weight slice");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // bt_lds_bytes(C2, CHAIN1, CDS, SLB, CONV2)
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  unsigned char* const bufA = smem + PRE;
  unsigned char* const bufB = smem;
  unsigned char* const w1buf = smem + PRE + SLB;                      // [2][W1B] (CHAIN1)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  // XCD-contiguous tile ranges (workgroups go round-robin to the 8 XCDs): the tiles of three neighbouring image rows read the
  // same conv2 input rows, and as neighbours in one XCD's range they find them in that L2
  const int nblk_ = (int)gridDim.x, xcd_ = blockIdx.x & 7, xq_ = nblk_ >> 3, xr_ = nblk_ & 7;
  const int tile_ = (xcd_ < xr_ ? xcd_ * (xq_ + 1) : xr_ * (xq_ + 1) + (xcd_ - xr_) * xq_) + (int)(blockIdx.x >> 3);
  const int m0 = tile_ * BT_BPOS;
  const int H = a.H, W = a.W, HW = H * W, M = a.M;

  // ---- weight slices of the 1x1 convs: 16 KB = 128 flat rows of 128 B, lane L of a wave instruction -> row L>>3,
  // physical slot L&7, i.e. it fetches logical chunk (L&7) ^ ((row>>1)&7)
  auto dma_w3_slice = [&](int p, unsigned char* buf) {
#pragma unroll
    for (int r = 0; r < NR3 / 32; ++r) {
      const int fr = (r * 4 + wave) * 8 + (lane >> 3);
      const int sub = fr / SL, row = fr - sub * SL;
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const uint16_t* src = a.w3 + (long long)(p * SL + row) * K3 + sub * 64 + chunk * 8;
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(buf + (r * 4 + wave) * 1024), 16, 0, 0);
    }
  };
  auto dma_w1_slice = [&](int p, unsigned char* buf) {       // rows = next conv1 couts (C2), k = [p*SL, (p+1)*SL)
#pragma unroll
    for (int r = 0; r < NR1 / 32; ++r) {
      const int fr = (r * 4 + wave) * 8 + (lane >> 3);
      const int sub = fr / C2, row = fr - sub * C2;
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const uint16_t* src = a.w1n + (long long)row * C4 + p * SL + sub * 64 + chunk * 8;
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(buf + (r * 4 + wave) * 1024), 16, 0, 0);
    }
  };
  dma_w3_slice(0, bufA);                         // independent of everything: lands under the K loop
  if constexpr (CHAIN1) dma_w1_slice(0, w1buf);

  bf16x8 tfr[KK3];                             // B fragments of conv3: the conv2 tile of this wave's 32 positions
  if constexpr (CONV2) {
    // ---- conv2: implicit GEMM, 32-wide K steps, LDS-DMA double buffer (the conv_dma32_kernel scheme, one level)
    const int j = (lane & 3) ^ ((lane >> 4) & 3);
    const int r0 = tid >> 2;
    int rhi[NX], rwi[NX];
    long long xoff[NX];
  #pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int m = m0 + r0 + 64 * i;
      if (m < M) {
        const int n = m / HW;
        const int rem = m - n * HW;
        const int ho = rem / W;
        const int wo = rem - ho * W;
        rhi[i] = ho - 1;
        rwi[i] = wo - 1;
        xoff[i] = ((long long)n * HW + (long long)rhi[i] * W + rwi[i]) * C2;
      } else {
        rhi[i] = -0x40000000;
        rwi[i] = 0;
        xoff[i] = 0;
      }
    }
    const uint16_t* ld_wp = a.w2 + (long long)r0 * KP2 + j * 8;
    int ld_cc = j, ld_kh = 0, ld_kw = 0;           // CPT >= 8 > j
    const unsigned long long zero_page = (unsigned long long)g_zero16b;
    const int wave_row = wave * 16;
    auto dma_tile = [&](int buf) {
      unsigned char* Wb = smem + buf * STAGE + wave_row * 64;
      unsigned char* Xb = Wb + C2 * 64;
      const long long toff = (long long)((ld_kh * W + ld_kw) * C2 + ld_cc * 8);
  #pragma unroll
      for (int i = 0; i < NW; ++i)
        __builtin_amdgcn_global_load_lds((glb_void*)(ld_wp + (long long)i * 64 * KP2), (lds_void*)(Wb + 64 * i * 64), 16, 0, 0);
  #pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int hi = rhi[i] + ld_kh, wi = rwi[i] + ld_kw;
        const bool ok = ((unsigned)hi < (unsigned)H) & ((unsigned)wi < (unsigned)W);
        const unsigned long long pm = ok ? ~0ull : 0ull;
        const unsigned long long src = ((unsigned long long)(a.x + xoff[i] + toff) & pm) | (zero_page & ~pm);
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(Xb + 64 * i * 64), 16, 0, 0);
      }
      ld_wp += 32;
      ld_cc += 4;
      const int wrap = ld_cc >= CPT ? 1 : 0;
      ld_cc -= wrap * CPT;
      ld_kw += wrap;
      const int wrap2 = ld_kw == 3 ? 1 : 0;
      ld_kw -= wrap2 * 3;
      ld_kh += wrap2;
    };

    f32x16 acc[TCO];
  #pragma unroll
    for (int tc = 0; tc < TCO; ++tc)
  #pragma unroll
      for (int e = 0; e < 16; ++e) acc[tc][e] = 0.f;
    const int rsw = (l31 >> 2) & 3;
    const int wrow_off = l31 * 64;
    const int xrow_off = C2 * 64 + (wave * 32 + l31) * 64;
    auto compute = [&](int buf) {
      const unsigned char* S = smem + buf * STAGE;
  #pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int slot = ((kk * 2 + khalf) ^ rsw) * 16;
        bf16x8 wf[TCO];
  #pragma unroll
        for (int t = 0; t < TCO; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(S + wrow_off + t * 32 * 64 + slot);
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(S + xrow_off + slot);
  #pragma unroll
        for (int tc = 0; tc < TCO; ++tc) acc[tc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tc], xf, acc[tc], 0, 0, 0);
      }
    };
    dma_tile(0);
    __syncthreads();
    for (int kt = 0; kt + 1 < NK; ++kt) {
      const int buf = kt & 1;
      dma_tile(buf ^ 1);
      compute(buf);
      __syncthreads();
    }
    compute((NK - 1) & 1);

    // ---- conv2 epilogue in registers: bias, ReLU, bf16 -> the B fragments of conv3 (K step kk = 2*tc + qp)
  #pragma unroll
    for (int tc = 0; tc < TCO; ++tc)
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
        const int c0 = tc * 32 + 16 * qp + 8 * khalf;
        const float4 b0 = *reinterpret_cast<const float4*>(a.b2 + c0);
        const float4 b1 = *reinterpret_cast<const float4*>(a.b2 + c0 + 4);
        v[0] += b0.x, v[1] += b0.y, v[2] += b0.z, v[3] += b0.w;
        v[4] += b1.x, v[5] += b1.y, v[6] += b1.z, v[7] += b1.w;
  #pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        tfr[2 * tc + qp] = __builtin_bit_cast(bf16x8, pack_bf16x8_v(v));
      }

  } else {
    // no conv2 in this launch (layer3's conv3 + next conv1 pairs): the conv3 operand is the wave's rows of `x` as they lie
    const int mx = min(m0 + wave * 32 + l31, M - 1);
    const uint16_t* xr = a.x + (long long)mx * C2 + 8 * khalf;
#pragma unroll
    for (int kk = 0; kk < KK3; ++kk) tfr[kk] = *reinterpret_cast<const bf16x8*>(xr + 16 * kk);
  }

  // ---- CDS: the B fragments of the shortcut conv -- lane (position, khalf) holds channels 16*kk + 8*khalf .. + 8 of its row
  bf16x8 dfr[KKD > 0 ? KKD : 1];
  if constexpr (CDS > 0) {
    const int md = min(m0 + wave * 32 + l31, M - 1);
    long long drow = md;
    if (a.ds_stride != 1) {                          // 1x1 / stride-s shortcut: output (n, ho, wo) reads input (n, s*ho, s*wo)
      const int n = md / HW, rem = md - n * HW;
      const int ho = rem / W, wo = rem - ho * W;
      drow = ((long long)n * a.dsH + ho * a.ds_stride) * a.dsW + wo * a.ds_stride;
    }
    const uint16_t* xr = a.xds + drow * CDS + 8 * khalf;
#pragma unroll
    for (int kk = 0; kk < KKD; ++kk) dfr[kk] = *reinterpret_cast<const bf16x8*>(xr + 16 * kk);
  }

  // ---- conv3 (+ chained conv1) in passes of SL couts
  const int m = m0 + wave * 32 + l31;
  const bool mok = m < M;
  const long long orow = (long long)(mok ? m : 0) * C4;
  f32x16 acc1[CHAIN1 ? TCO : 1];
  if constexpr (CHAIN1) {
#pragma unroll
    for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[tc][e] = 0.f;
  }
  const int asw = (l31 >> 1) & 7;                // A-fragment swizzle: rows ct*32 + l31, 32 | row base
  constexpr bool RVPF = !CONV2 && CDS == 0;
  u32x4 rvn[RVPF ? CT : 1][2];
  if constexpr (RVPF) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp)
        rvn[ct][qp] = *reinterpret_cast<const u32x4*>(a.res + orow + ct * 32 + 16 * qp + 8 * khalf);
  }
#pragma unroll 1
  for (int p = 0; p < NPASS; ++p) {
    unsigned char* const buf = (p & 1) ? bufB : bufA;
    __syncthreads();                             // slice p has landed; the other buffer's readers (pass p-1) are done
    if (p + 1 < NPASS && !(BT_DBG(a) & 2)) {
      dma_w3_slice(p + 1, (p & 1) ? bufA : bufB);
      if constexpr (CHAIN1) dma_w1_slice(p + 1, w1buf + ((p + 1) & 1) * W1B);
    }
    // identity chunks of this pass.  RVPF (the 1x1 pair: 32 conv3 MFMAs per pass cannot hide an HBM round trip, and it
    // has the registers): they were requested one pass ahead
    u32x4 rv[CT][2];
    if constexpr (CDS == 0 && !RVPF) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp)
          rv[ct][qp] = *reinterpret_cast<const u32x4*>(a.res + orow + p * SL + ct * 32 + 16 * qp + 8 * khalf);
    }
    if constexpr (RVPF) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          rv[ct][qp] = rvn[ct][qp];
          if (p + 1 < NPASS && !(BT_DBG(a) & 4))
            rvn[ct][qp] = *reinterpret_cast<const u32x4*>(a.res + orow + (p + 1) * SL + ct * 32 + 16 * qp + 8 * khalf);
        }
    }
    f32x16 acc3[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc3[ct][e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK3 + KKD; ++kk) {     // K = [conv2 tile | shortcut input]
      const int sub = kk >> 2;
      const int slot = ((((kk & 3) * 2 + khalf) ^ asw)) * 16;
      const bf16x8 bfr = kk < KK3 ? tfr[kk < KK3 ? kk : 0] : dfr[kk >= KK3 ? kk - KK3 : 0];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(buf + (sub * SL + ct * 32 + l31) * 128 + slot);
        acc3[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, bfr, acc3[ct], 0, 0, 0);
      }
    }
    bf16x8 yfr[CT * 2];                          // this pass's outputs as B fragments of the chained conv1
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t lo = __float_as_uint(acc3[ct][4 * (2 * qp) + e]);
          const uint32_t hi = __float_as_uint(acc3[ct][4 * (2 * qp + 1) + e]);
          const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
          v[e] = __uint_as_float(r[0]);
          v[4 + e] = __uint_as_float(r[1]);
        }
        const int c0 = p * SL + ct * 32 + 16 * qp + 8 * khalf;
        const float4 b0 = *reinterpret_cast<const float4*>(a.b3 + c0);
        const float4 b1 = *reinterpret_cast<const float4*>(a.b3 + c0 + 4);
        v[0] += b0.x, v[1] += b0.y, v[2] += b0.z, v[3] += b0.w;
        v[4] += b1.x, v[5] += b1.y, v[6] += b1.z, v[7] += b1.w;
        if constexpr (CDS == 0) {
          float f[8];
          unpack_bf16x8(rv[ct][qp], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] + f[e], 0.f);
        } else {                                   // the shortcut is already in the accumulator (bias folded into b3)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        const u32x4 packed = pack_bf16x8_v(v);
        if (mok && !(BT_DBG(a) & 1)) *reinterpret_cast<u32x4*>(a.y + orow + c0) = packed;
        yfr[ct * 2 + qp] = __builtin_bit_cast(bf16x8, packed);
      }
    if (CHAIN1 && !(BT_DBG(a) & 8)) {
      const unsigned char* wb = w1buf + (p & 1) * W1B;
#pragma unroll
      for (int kk = 0; kk < SL / 16; ++kk) {       // k = p*SL + 16*kk (+ 8*khalf) <-> yfr[kk]
        const int sub = kk >> 2;
        const int slot = ((((kk & 3) * 2 + khalf) ^ asw)) * 16;
#pragma unroll
        for (int tc = 0; tc < TCO; ++tc) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + (sub * C2 + tc * 32 + l31) * 128 + slot);
          acc1[tc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, yfr[kk], acc1[tc], 0, 0, 0);
        }
      }
    }
  }
  if constexpr (CHAIN1) {
#pragma unroll
    for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t lo = __float_as_uint(acc1[tc][4 * (2 * qp) + e]);
          const uint32_t hi = __float_as_uint(acc1[tc][4 * (2 * qp + 1) + e]);
          const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
          v[e] = __uint_as_float(r[0]);
          v[4 + e] = __uint_as_float(r[1]);
        }
        const int c0 = tc * 32 + 16 * qp + 8 * khalf;
        const float4 b0 = *reinterpret_cast<const float4*>(a.b1n + c0);
        const float4 b1 = *reinterpret_cast<const float4*>(a.b1n + c0 + 4);
        v[0] += b0.x, v[1] += b0.y, v[2] += b0.z, v[3] += b0.w;
        v[4] += b1.x, v[5] += b1.y, v[6] += b1.z, v[7] += b1.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        if (mok) *reinterpret_cast<u32x4*>(a.t1n + (long long)m * C2 + c0) = pack_bf16x8_v(v);
      }
  }
}

constexpr int bt_lds_bytes(int c2, bool chain, int cds, int slb, bool conv2) {
  return (conv2 ? 2 * (c2 + BT_BPOS) * 64 : slb) + slb + (chain ? 2 * (c2 * (slb / ((c2 + cds) * 2)) * 2) : 0);
}

template <int C2, bool CHAIN1, int OCC, int CDS = 0, int SLB = BT_SLICE_BYTES, bool CONV2 = true>
int bt_launch(const BtArgs& a, dim3 grid, hipStream_t s) {
  constexpr int lds = bt_lds_bytes(C2, CHAIN1, CDS, SLB, CONV2);
  // > 64 KB of dynamic LDS needs the opt-in, once per (kernel, device)
  if (sm_lds_optin(reinterpret_cast<const void*>(&bottleneck_tail_kernel<C2, CHAIN1, OCC, CDS, SLB, CONV2>), lds) != hipSuccess)
    return SM_ERR_LAUNCH;
  hipLaunchKernelGGL((bottleneck_tail_kernel<C2, CHAIN1, OCC, CDS, SLB, CONV2>), grid, dim3(256), lds, s, a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}

}  // namespace

extern "C" int sm_bottleneck_tail_supported(int channels) { return (channels == 64 || channels == 128) ? 1 : 0; }

extern "C" int sm_bottleneck_tail(int batch, int h, int w, int channels, const void* x, const void* w2, const float* b2,
                                  const void* w3, const float* b3, const void* identity, void* y, const void* w1_next,
                                  const float* b1_next, void* t1_next, sm_stream_t stream) {
  if (!x || !w2 || !b2 || !w3 || !b3 || !identity || !y || batch < 1 || h < 1 || w < 1) return SM_ERR_BAD_ARG;
  if (channels != 64 && channels != 128) return SM_ERR_UNSUPPORTED;
  const bool chain = w1_next != nullptr;
  if (chain && (!b1_next || !t1_next)) return SM_ERR_BAD_ARG;
  const long long M = (long long)batch * h * w;
  if (M * 4 * channels >= (1ll << 31) * 8) return SM_ERR_BAD_SHAPE;
  BtArgs a;
  a.x = (const uint16_t*)x;
  a.w2 = (const uint16_t*)w2;
  a.b2 = b2;
  a.w3 = (const uint16_t*)w3;
  a.b3 = b3;
  a.res = (const uint16_t*)identity;
  a.xds = nullptr;
  a.ds_stride = 1, a.dsH = h, a.dsW = w;
  a.dbg = 0;
  a.y = (uint16_t*)y;
  a.w1n = (const uint16_t*)w1_next;
  a.b1n = b1_next;
  a.t1n = (uint16_t*)t1_next;
  a.batch = batch;
  a.H = h;
  a.W = w;
  a.M = (int)M;
  const dim3 grid(sm_cdiv(M, BT_BPOS));
  hipStream_t s = sm_hip_stream(stream);
  // (three blocks per CU: a fourth -- 128 VGPRs, four spills -- measured neutral in round 4 and is gone)
  if (channels == 64) {
    if (chain) return bt_launch<64, true, 3, 0, 8192>(a, grid, s);     // 8 KB slices: see SLB
    return bt_launch<64, false, 3>(a, grid, s);
  }
  if (chain) return bt_launch<128, true, 2>(a, grid, s);     // 80 KB of LDS: two blocks per CU whatever the registers -- no cap, no spills
  return bt_launch<128, false, 3>(a, grid, s);
}

/* conv2 + conv3 + the block's 1x1 SHORTCUT conv as one launch: the first bottleneck of layer1 (64 -> 256, stride 1) and of
 * layer2 (256 -> 512, stride 2); optionally the next block's conv1 chained behind it like sm_bottleneck_tail (layer1). */
extern "C" int sm_bottleneck_tail_ds(int batch, int h, int w, int channels, const void* x, const void* w2, const float* b2,
                                     const void* w3_ds, const float* b3_ds, const void* x_block, int ds_channels,
                                     int ds_stride, int ds_h, int ds_w, void* y, const void* w1_next, const float* b1_next,
                                     void* t1_next, sm_stream_t stream) {
  if (!x || !w2 || !b2 || !w3_ds || !b3_ds || !x_block || !y || batch < 1 || h < 1 || w < 1) return SM_ERR_BAD_ARG;
  const bool l1 = channels == 64 && ds_channels == 64 && ds_stride == 1;
  const bool l2 = channels == 128 && ds_channels == 256 && ds_stride == 2;
  if (!l1 && !l2) return SM_ERR_UNSUPPORTED;
  if ((ds_h - 1) / ds_stride + 1 != h || (ds_w - 1) / ds_stride + 1 != w) return SM_ERR_BAD_SHAPE;
  const bool chain = w1_next != nullptr;
  if (chain && (!b1_next || !t1_next || !l1)) return SM_ERR_BAD_ARG;
  const long long M = (long long)batch * h * w;
  if (M * 4 * channels >= (1ll << 31) * 8) return SM_ERR_BAD_SHAPE;
  BtArgs a;
  a.x = (const uint16_t*)x;
  a.w2 = (const uint16_t*)w2;
  a.b2 = b2;
  a.w3 = (const uint16_t*)w3_ds;
  a.b3 = b3_ds;
  a.res = nullptr;
  a.xds = (const uint16_t*)x_block;
  a.ds_stride = ds_stride, a.dsH = ds_h, a.dsW = ds_w;
  a.dbg = 0;
  a.y = (uint16_t*)y;
  a.w1n = (const uint16_t*)w1_next;
  a.b1n = b1_next;
  a.t1n = (uint16_t*)t1_next;
  a.batch = batch;
  a.H = h;
  a.W = w;
  a.M = (int)M;
  const dim3 grid(sm_cdiv(M, BT_BPOS));
  hipStream_t s = sm_hip_stream(stream);
  if (l2) return bt_launch<128, false, 3, 256, 24576>(a, grid, s);   // K3 = 384: 32-cout slices of 24 KB
  if (chain) return bt_launch<64, true, 3, 64>(a, grid, s);
  return bt_launch<64, false, 4, 64>(a, grid, s);   // 95 VGPRs, 40 KB of LDS: four blocks per CU
}

/* conv3 (1x1, C -> 4C, + identity, ReLU) of one bottleneck chained with conv1 (1x1, 4C -> C, ReLU) of the next as ONE launch
 * (resnet.py:188-200 -> :175-178): layer3's pairs, channels == 256.  x = conv2's output rows [M][C]. */
extern "C" int sm_conv1x1_pair(long long rows, int channels, const void* x, const void* w3, const float* b3,
                               const void* identity, void* y, const void* w1_next, const float* b1_next, void* t1_next,
                               sm_stream_t stream) {
  if (!x || !w3 || !b3 || !identity || !y || !w1_next || !b1_next || !t1_next || rows < 1) return SM_ERR_BAD_ARG;
  if (channels != 256) return SM_ERR_UNSUPPORTED;
  if (rows * 4 * channels >= (1ll << 31) * 8) return SM_ERR_BAD_SHAPE;
  BtArgs a;
  a.x = (const uint16_t*)x;
  a.w2 = nullptr;
  a.b2 = nullptr;
  a.w3 = (const uint16_t*)w3;
  a.b3 = b3;
  a.res = (const uint16_t*)identity;
  a.xds = nullptr;
  a.ds_stride = 1, a.dsH = 1, a.dsW = 1;
#ifdef SM_EXPERIMENTS
  static const int dbg = sm_experiment_env("SIPMASK_PAIR_DEBUG", 0);
  a.dbg = dbg;
#else
  a.dbg = 0;
#endif
  a.y = (uint16_t*)y;
  a.w1n = (const uint16_t*)w1_next;
  a.b1n = b1_next;
  a.t1n = (uint16_t*)t1_next;
  a.batch = 1;
  a.H = 1;
  a.W = (int)rows;
  a.M = (int)rows;
  return bt_launch<256, true, 1, 0, 32768, false>(a, dim3(sm_cdiv(rows, BT_BPOS)), sm_hip_stream(stream));
}
