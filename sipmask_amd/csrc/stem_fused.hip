// ResNet stem as ONE launch (gfx950): conv 7x7 / stride 2 / pad 3 (3 -> 64, frozen BN folded by the caller) + ReLU +
// max-pool 3x3 / stride 2 / pad 1, straight from the NCHW f32 image to the NHWC bf16 rows layer1 reads.
// Reference: ResNet.forward, M/mmdet/models/backbones/resnet.py:497-505 (conv1 -> norm1 -> relu -> maxpool).
//
// Why.  As three launches (NCHW -> NHWC bf16 with the channels padded to 8, implicit-GEMM conv at K = 49 * 8 = 392,
// max-pool) the stem moves 52 + 69 | 69 + 138 | 138 + 34 MB per four 800 x 1344 images and computes 2.7x the useful
// FLOPs: 0.19 ms, most of it writing the 400 x 672 x 64 conv output and reading it back.  Here a block owns 8 x 14
// POOLED positions: the 17 x 29 conv positions under them are computed from a 39 x 63 input patch held in LDS
// ([row][col][4 channels] bf16: the two kw-adjacent pixels of a tap pair are 16 contiguous bytes, i.e. one MFMA
// B fragment per ds_read_b128, K = 7 kh x 8 kw x 4 c = 224 with zero weights on the padding), staged as bf16 in LDS
// and pooled there; only the pooled rows go to HBM: 52 MB in, 34 MB out.
//
// Numerics: same rounding points as the three-launch path (image and weights rounded to bf16, f32 accumulation,
// bias + ReLU in f32, ONE rounding to bf16, max over the rounded values); the K order of the f32 sum differs, so a
// result can differ from that path by one bf16 ulp where the f32 sums straddle a rounding boundary.

#include "common.h"

namespace {

constexpr int SF_IC = 64;                               // patch columns: 2 * CC + 5 real ones (+ zeros for the kw = 7 slot), 512-byte rows
constexpr int SF_KSTEPS = 14;                           // K = 224 in steps of 16

// PR x PC pooled positions per tile on 64 * NT / 2 threads (every wave owns two MFMA N tiles of 32 conv positions):
//   <8, 14>: 17 x 29 = 493 conv positions = 16 N tiles, 8 waves, 84 KB of LDS: one block per CU
//   <4, 12>:  9 x 25 = 225 conv positions =  8 N tiles, 4 waves, 44 KB: two blocks per CU -- one computes while the
//             other pools / stores / waits for its next patch
template <int PR, int PC>
struct StemCfg {
  static constexpr int CR = 2 * PR + 1, CC = 2 * PC + 1;          // conv positions under the pooled tile
  static constexpr int NPOS = CR * CC;
  static constexpr int NT = (NPOS + 31) / 32;
  static constexpr int THREADS = 64 * NT / 2;
  static constexpr int IR = 2 * CR + 5;                            // input rows
  static constexpr int PATCH_BYTES = IR * SF_IC * 8;
  static constexpr int OUT_BYTES = NT * 32 * 128;                  // [conv position][64 couts] bf16, 16-byte chunks XOR-swizzled
  static constexpr int PIX = IR * SF_IC;
  static constexpr int LOADS = (PIX + THREADS - 1) / THREADS;
  static constexpr int LDS = PATCH_BYTES + OUT_BYTES;
  static_assert(NT % 2 == 0 && 2 * CC + 5 < SF_IC, "tile shape");
};

struct StemArgs {
  const float* img;      // [B][3][H][W] f32
  const uint16_t* w;     // [64][224] bf16: (kh, kw8, c4), zero for kw == 7 and c == 3
  const float* bias;     // [64]
  uint16_t* y;           // [B * H2 * W2][64] bf16
  int batch, H, W, h1, w1, H2, W2;
  int tiles_y, tiles_x, ntiles;
};

typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

template <int PR, int PC>
__global__ __launch_bounds__((StemCfg<PR, PC>::THREADS), (StemCfg<PR, PC>::THREADS <= 256 ? 2 : 1)) void stem_fused_kernel(const StemArgs a) {
  using C = StemCfg<PR, PC>;
  constexpr int SF_PR = PR, SF_PC = PC, SF_CC = C::CC, SF_NPOS = C::NPOS, SF_NT = C::NT, SF_THREADS = C::THREADS;
  constexpr int SF_PATCH_BYTES = C::PATCH_BYTES, SF_PIX = C::PIX, SF_LOADS = C::LOADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const patch = smem;
  unsigned char* const outt = smem + SF_PATCH_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, khalf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- weights: both cout tiles, all 14 K steps, in registers for the whole kernel (112 VGPRs)
  bf16x8 wf[2][SF_KSTEPS];
#pragma unroll
  for (int tc = 0; tc < 2; ++tc) {
#pragma unroll
    for (int s = 0; s < SF_KSTEPS; ++s)
      wf[tc][s] = *reinterpret_cast<const bf16x8*>(a.w + (tc * 32 + l31) * 224 + s * 16 + khalf * 8);
  }
  // the fourth channel slot and the 64th column are never written again
  for (int i = tid; i < SF_PATCH_BYTES / 8; i += SF_THREADS) reinterpret_cast<unsigned long long*>(patch)[i] = 0ull;

  // XCD-contiguous tile ranges (workgroups go round-robin to the 8 XCDs; neighbouring tiles share input halos)
  const int nblk = (int)gridDim.x, xcd = blockIdx.x & 7, per = (nblk + 7) >> 3;
  const int tq = a.ntiles >> 3, tr = a.ntiles & 7;
  const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int t_end = t_begin + (xcd < tr ? tq + 1 : tq);
  int tile = t_begin + (int)(blockIdx.x >> 3);

  float raw[SF_LOADS][3];
  auto fetch = [&](int t) {                   // global loads of tile t's patch into registers (zero outside the image)
    const int b = t / (a.tiles_y * a.tiles_x);
    const int r_ = t - b * (a.tiles_y * a.tiles_x);
    const int ty = r_ / a.tiles_x, tx = r_ - ty * a.tiles_x;
    const int ir0 = 4 * (ty * SF_PR) - 5, ic0 = 4 * (tx * SF_PC) - 5;
    const float* base = a.img + (long long)b * 3 * a.H * a.W;
#pragma unroll
    for (int i = 0; i < SF_LOADS; ++i) {
      const int p = tid + i * SF_THREADS;
      const int r = p >> 6, c = p & 63;
      const int gr = ir0 + r, gc = ic0 + c;
      const bool in = p < SF_PIX && c < 2 * SF_CC + 5 && (unsigned)gr < (unsigned)a.H && (unsigned)gc < (unsigned)a.W;
      const long long o = (long long)gr * a.W + gc;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) raw[i][ch] = in ? base[(long long)ch * a.H * a.W + o] : 0.f;
    }
  };
  auto stash = [&]() {                        // registers -> LDS patch, rounded to bf16
#pragma unroll
    for (int i = 0; i < SF_LOADS; ++i) {
      const int p = tid + i * SF_THREADS;
      if (p < SF_PIX) {
        uint2 v;
        v.x = pack_bf16x2(raw[i][0], raw[i][1]);
        v.y = pack_bf16x2(raw[i][2], 0.f);
        *reinterpret_cast<uint2*>(patch + p * 8) = v;
      }
    }
  };

  // this wave's two N tiles: per-lane LDS base of position n (row-major in the 17 x 29 conv tile)
  unsigned baddr[2];
  int prow[2], pcol[2];
#pragma unroll
  for (int tp = 0; tp < 2; ++tp) {
    const int n = (wave * 2 + tp) * 32 + l31;
    const int nn = n < SF_NPOS ? n : 0;
    prow[tp] = nn / SF_CC;
    pcol[tp] = nn - prow[tp] * SF_CC;
    baddr[tp] = (unsigned)((2 * prow[tp] * SF_IC + 2 * pcol[tp]) * 8 + khalf * 16);
  }

  __syncthreads();
  if (tile < t_end) {
    fetch(tile);
    stash();
  }
  __syncthreads();
  for (; tile < t_end; tile += per) {
    const int next = tile + per;
    if (next < t_end) fetch(next);            // lands under the MFMA phase
    const int b = tile / (a.tiles_y * a.tiles_x);
    const int r_ = tile - b * (a.tiles_y * a.tiles_x);
    const int ty = r_ / a.tiles_x, tx = r_ - ty * a.tiles_x;
    const int py0 = ty * SF_PR, px0 = tx * SF_PC;
    const int cr0 = 2 * py0 - 1, cc0 = 2 * px0 - 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
      for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tc][tp][e] = 0.f;
#pragma unroll
    for (int s = 0; s < SF_KSTEPS; ++s) {
      bf16x8 xf[2];
#pragma unroll
      for (int tp = 0; tp < 2; ++tp)
        xf[tp] = *reinterpret_cast<const bf16x8*>(patch + baddr[tp] + (s >> 1) * (SF_IC * 8) + (s & 1) * 32);
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
          acc[tc][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tc][s], xf[tp], acc[tc][tp], 0, 0, 0);
    }
    __syncthreads();                          // every wave is done with the patch
    if (next < t_end) stash();

    // ---- epilogue: bias + ReLU + bf16 into the LDS tile (positions outside the conv output are the pool's padding: 0,
    // which never wins against a post-ReLU value and every window holds at least one real position)
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      const int n = (wave * 2 + tp) * 32 + l31;
      const int gr = cr0 + prow[tp], gc = cc0 + pcol[tp];
      const bool pvalid = n < SF_NPOS && (unsigned)gr < (unsigned)a.h1 && (unsigned)gc < (unsigned)a.w1;
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) {
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t lo = __float_as_uint(acc[tc][tp][4 * (2 * qp) + e]);
            const uint32_t hi = __float_as_uint(acc[tc][tp][4 * (2 * qp + 1) + e]);
            const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
            v[e] = __uint_as_float(r[0]);
            v[4 + e] = __uint_as_float(r[1]);
          }
          const int c0 = tc * 32 + 8 * (2 * qp + khalf);
          const float4 b0 = *reinterpret_cast<const float4*>(a.bias + c0);
          const float4 b1 = *reinterpret_cast<const float4*>(a.bias + c0 + 4);
          v[0] += b0.x, v[1] += b0.y, v[2] += b0.z, v[3] += b0.w;
          v[4] += b1.x, v[5] += b1.y, v[6] += b1.z, v[7] += b1.w;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (pvalid && v[e] > 0.f) ? v[e] : 0.f;   // +0 for -0 and NaN: the pool compares bits
          if (n < SF_NT * 32)
            *reinterpret_cast<u32x4*>(outt + n * 128 + (((c0 >> 3) ^ (n & 7)) << 4)) = pack_bf16x8_v(v);
        }
      }
    }
    __syncthreads();
    // ---- 3x3 / stride-2 max over the bf16 tile (non-negative values: the unsigned 16-bit order is the float order)
    for (int it = tid; it < SF_PR * SF_PC * 8; it += SF_THREADS) {
      const int pp = it >> 3, ch = it & 7;
      const int py = pp / SF_PC, px = pp - py * SF_PC;
      const int gy = py0 + py, gx = px0 + px;
      u16x8 m = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int n = (2 * py + dy) * SF_CC + 2 * px + dx;
          const u16x8 v = *reinterpret_cast<const u16x8*>(outt + n * 128 + ((ch ^ (n & 7)) << 4));
          m = __builtin_elementwise_max(m, v);
        }
      }
      if (gy < a.H2 && gx < a.W2)
        *reinterpret_cast<u16x8*>(a.y + (((long long)b * a.H2 + gy) * a.W2 + gx) * 64 + ch * 8) = m;
    }
    // (no barrier: the next write to the LDS tile comes after the barrier that follows the next MFMA phase)
  }
}

}  // namespace

namespace {
template <int PR, int PC>
int launch_stem(StemArgs a, sm_stream_t stream) {
  using C = StemCfg<PR, PC>;
  a.tiles_y = (a.H2 + PR - 1) / PR;
  a.tiles_x = (a.W2 + PC - 1) / PC;
  const long long nt = (long long)a.batch * a.tiles_y * a.tiles_x;
  if (nt > 0x7fffffffll) return SM_ERR_BAD_SHAPE;
  a.ntiles = (int)nt;
  if (sm_lds_optin((const void*)stem_fused_kernel<PR, PC>, C::LDS) != hipSuccess) return SM_ERR_LAUNCH;
  const int resident = 256 * (C::THREADS <= 256 ? 2 : 1);          // blocks the chip holds; a multiple of the 8 XCDs
  const int blocks = a.ntiles < resident ? ((a.ntiles + 7) / 8) * 8 : resident;
  hipLaunchKernelGGL((stem_fused_kernel<PR, PC>), dim3(blocks), dim3(C::THREADS), C::LDS, sm_hip_stream(stream), a);
  SM_LAUNCH_CHECK();
  return SM_OK;
}
}  // namespace

extern "C" int sm_stem_fused(const float* img, const void* w_stem, const float* bias, void* y, int batch, int H, int W,
                             sm_stream_t stream) {
  if (!img || !w_stem || !bias || !y) return SM_ERR_BAD_ARG;
  if (batch < 1 || H < 7 || W < 7) return SM_ERR_BAD_SHAPE;
  StemArgs a;
  a.img = img;
  a.w = (const uint16_t*)w_stem;
  a.bias = bias;
  a.y = (uint16_t*)y;
  a.batch = batch;
  a.H = H;
  a.W = W;
  a.h1 = (H + 6 - 7) / 2 + 1;
  a.w1 = (W + 6 - 7) / 2 + 1;
  a.H2 = (a.h1 + 2 - 3) / 2 + 1;
  a.W2 = (a.w1 + 2 - 3) / 2 + 1;
  // (round 4 A/B: 4 x 12 pooled positions per tile at two blocks per CU measured SLOWER -- 0.080 against 0.072 ms for four
  // 800 x 1344 images: the wider halo and twice the per-tile overheads cost more than the second resident block hides)
  return launch_stem<8, 14>(a, stream);
}
