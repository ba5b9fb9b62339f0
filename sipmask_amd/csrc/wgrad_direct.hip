// Weight gradient of a plain convolution straight from the NHWC row tensors (gfx950):
//
//   dW^T[(tap, ci)][co] = sum over output positions p of  x[p (+) tap][ci] * gout[p][co]
//
// Both operands are [position][channel] row matrices, i.e. the contraction index is the SLOW dimension of both, while an
// MFMA lane wants 8 consecutive k of one row / column.  The first version (deform_bwd.hip: transpose_tile_kernel + the
// implicit-GEMM kernel + wgrad_reduce_kernel) therefore materialised im2col^T (9x the activation for a 3x3 conv) and
// gout^T per conv: 3.6 ms of transposes + 1.0 ms of partial-slab reduction in a 33 ms training step.  Here the tiles are
// LDS-DMAed as they lie -- [32 positions][16 channels] sub-tiles of 1 KB, one `global_load_lds_dwordx4` each -- and the
// fragments are read with gfx950's transposing LDS read:
//
//   ds_read_b64_tr_b16: inside a 16-lane group, lane 4r+q supplies the address of 4 consecutive bf16 (row r, columns
//   4q..4q+3 of a 4 x 16 block; the row pitch is free); lane i receives column i = {row 0..3}[i].
//   (measured with tools/tr_probe.hip; MI355X_MICROARCH.md lists the rate: 2 LDS cycles per wave instruction)
//
// so two reads give a lane the 8 consecutive POSITIONS (k) of its channel that v_mfma_f32_32x32x16_bf16 wants, for the
// x tile (A: M = ci) and the gout tile (B: N = co) alike.  Sub-tiles are laid out at a 1152-byte pitch: the two 16-lane
// groups an LDS cycle serves then hit disjoint bank halves.  Split-K over position slices, partial tiles are added into
// the (zeroed) output with hardware float atomics, coalesced along co (lanes = N).
#include "common.h"

namespace {

struct WGArgs {
  const uint16_t* x;
  const uint16_t* g;
  float* out;                      // [kh*kw*cin][cout] f32, zeroed by the host entry point
  int nlev, batch;
  int in_h[SM_MAX_LEVELS], in_w[SM_MAX_LEVELS], out_h[SM_MAX_LEVELS], out_w[SM_MAX_LEVELS];
  long long in_row0[SM_MAX_LEVELS], out_row0[SM_MAX_LEVELS];
  long long pos_end[SM_MAX_LEVELS];   // compact position index one past each level (level, image, y, x order)
  int cin, cout, kh, kw, stride, pad, dil, in_cstride, g_cstride;
  long long P;
  int slice;                       // positions per split-K slice (multiple of 32)
  int ci_tiles, co_tiles;          // tiles along cin / cout
  int per_slice;                   // blocks of one slice = taps * ci_tiles * co_tiles
};

constexpr int WG_PB = 2;           // 32-position blocks per stage
constexpr int WG_KC = 32 * WG_PB;  // positions per stage
constexpr int WG_SUB = 1152;       // sub-tile pitch: 1 KB of data + 128 B so that neighbouring sub-tiles sit in the other bank half

__device__ __attribute__((aligned(16))) const unsigned int g_zero16w[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ unsigned long long tr_read(unsigned addr) {
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ unsigned long long tr_read_128(unsigned addr) {       // + 4 rows of 32 bytes
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:128" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// Tile = (WM * TM * 32) ci x (WN * TN * 32) co on WM x WN waves, each TM x TN MFMA tiles of 32 x 32.
//   <2,2,2,2>: 128 x 128 on 4 waves, two blocks per CU          (64 FLOP per LDS-DMA byte)
//   <2,4,4,2>: 256 x 256 on 8 waves, one block per CU           (128 FLOP per LDS-DMA byte; half the DMA instructions per MFMA)
// The loop is bound by the LDS-DMA path (L2 -> LDS delivers ~43 GB/s per CU with every CU streaming, and every
// global_load_lds costs its wave ~100 issue cycles), so the big tile is taken whenever both channel counts fill it.
template <int WM, int WN, int TM, int TN>
struct WGCfg {
  static constexpr int NW = WM * WN, THREADS = 64 * NW;
  static constexpr int TILE_M = WM * TM * 32, TILE_N = WN * TN * 32;
  static constexpr int SUB_M = TILE_M / 16, SUB_N = TILE_N / 16;          // 16-channel sub-tiles per 32-position block
  static constexpr int REGION_X = WG_PB * SUB_M * WG_SUB, REGION_G = WG_PB * SUB_N * WG_SUB;
  static constexpr int STAGE = REGION_X + REGION_G;
  static constexpr int LDS = 2 * STAGE;                                   // double buffer
  static constexpr int XW = NW / 2;                                       // waves that load x (the rest load gout)
  static constexpr int SUB_PER_WAVE_X = SUB_M / XW, SUB_PER_WAVE_G = SUB_N / (NW - XW);
  static_assert(SUB_M % XW == 0 && SUB_N % (NW - XW) == 0, "sub-tiles split evenly over the loader waves");
};

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN <= 4 ? 2 : 1)) void wgrad_direct_kernel(const WGArgs a) {
  using C = WGCfg<WM, WN, TM, TN>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // 2 stages x (x region + gout region)
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid, decoded XCD-aware: workgroups go round-robin to the 8 XCDs, and the blocks of ONE split-K slice -- its 9 taps
  // and its channel tiles -- read the same gout rows and (shifted) the same x rows.  The first version laid the grid out
  // as (tap, co tile, slice): consecutive taps of a slice landed on eight different L2s and every one of them fetched the
  // slice from HBM -- TCC hit rate 11 %, 1.03 GB of misses per tower dW launch against 92 MB of operands (PMC, round 3).
  // Here block id -> XCD id % 8, and all blocks of slice s run consecutively on XCD s % 8.
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int sl = (kq / a.per_slice) * 8 + xcd;
  const int rq = kq % a.per_slice;
  const int ntap = a.kh * a.kw;
  const int tap = rq % ntap;
  const int ci0 = ((rq / ntap) % a.ci_tiles) * C::TILE_M;
  const int co0 = (rq / (ntap * a.ci_tiles)) * C::TILE_N;
  const int ti = tap / a.kw, tj = tap - ti * a.kw;
  const long long p_begin = (long long)sl * a.slice;
  const long long p_end = p_begin + a.slice < a.P ? p_begin + a.slice : a.P;
  if (p_begin >= p_end) return;
  const int nst = (int)((p_end - p_begin + WG_KC - 1) / WG_KC);
  const unsigned long long zero_page = (unsigned long long)g_zero16w;

  // ---- loader: the first half of the waves fetch the x tile, the second half the gout tile; a wave's lane = (position
  // lane >> 1, 8-channel half lane & 1) of every one of its sub-tiles, so one position decode per thread per 32 positions
  const bool is_x = wave < C::XW;
  const int nsub = is_x ? C::SUB_PER_WAVE_X : C::SUB_PER_WAVE_G;
  const int sub0 = (is_x ? wave : wave - C::XW) * nsub;
  const int sub_all = is_x ? C::SUB_M : C::SUB_N;
  const int lpos = lane >> 1, lhalf = lane & 1;
  auto issue = [&](int st, int buf) {
#pragma unroll
    for (int pb = 0; pb < WG_PB; ++pb) {
      const long long p = p_begin + (long long)st * WG_KC + pb * 32 + lpos;
      unsigned long long src_row = 0;              // byte address of channel 0 of the row, 0 = no row (zero page)
      if (p < p_end) {
        int l = 0;
#pragma unroll
        for (int q = 1; q < SM_MAX_LEVELS; ++q)
          if (q < a.nlev && p >= a.pos_end[q - 1]) l = q;
        const int pl = (int)(p - (l ? a.pos_end[l - 1] : 0));     // < 2^31 per level (checked by the host): 32-bit divisions
        if (is_x) {
          const int hw = a.out_h[l] * a.out_w[l];
          const int b = pl / hw;
          const int rem = pl - b * hw;
          const int oy = rem / a.out_w[l], ox = rem - oy * a.out_w[l];
          const int iy = oy * a.stride - a.pad + ti * a.dil, ix = ox * a.stride - a.pad + tj * a.dil;
          if (iy >= 0 && iy < a.in_h[l] && ix >= 0 && ix < a.in_w[l])
            src_row = (unsigned long long)(a.x + (a.in_row0[l] + ((long long)b * a.in_h[l] + iy) * a.in_w[l] + ix) * a.in_cstride);
        } else {
          src_row = (unsigned long long)(a.g + (a.out_row0[l] + pl) * a.g_cstride);
        }
      }
      const int c_base = (is_x ? ci0 : co0) + lhalf * 8;
      const int c_lim = is_x ? a.cin : a.cout;
      unsigned char* dst = smem + buf * C::STAGE + (is_x ? 0 : C::REGION_X) + pb * sub_all * WG_SUB;
#pragma unroll
      for (int i = 0; i < (C::SUB_PER_WAVE_X > C::SUB_PER_WAVE_G ? C::SUB_PER_WAVE_X : C::SUB_PER_WAVE_G); ++i) {
        if (i < nsub) {                                        // wave-uniform
          const int s = sub0 + i;
          const int c = c_base + s * 16;
          const bool ok = src_row != 0 && c < c_lim;           // channels are multiples of 8: a 16-byte piece is all in or all out
          const unsigned long long src = ok ? src_row + (unsigned long long)c * 2 : zero_page;
          __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dst + s * WG_SUB), 16, 0, 0);
        }
      }
    }
  };

  // ---- fragment addresses.  Inside a 16-lane group: i = lane & 15 -> supplied row r = i >> 2, column quad q = i & 3;
  // group g = lane >> 4 -> channel half (g & 1) of the 32-wide MFMA tile and position half (g >> 1) of the 16-position k step.
  const int wm = wave / WN, wn = wave % WN;
  const int grp = lane >> 4, li = lane & 15;
  const unsigned frag_off = (unsigned)(((grp >> 1) * 8 + (li >> 2)) * 32 + (li & 3) * 8);       // inside a sub-tile, k step 0
  const unsigned smem_base = (unsigned)(uintptr_t)smem;
  unsigned a_addr[TM], b_addr[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) a_addr[t] = smem_base + (unsigned)((wm * TM * 2 + t * 2 + (grp & 1)) * WG_SUB) + frag_off;
#pragma unroll
  for (int t = 0; t < TN; ++t)
    b_addr[t] = smem_base + C::REGION_X + (unsigned)((wn * TN * 2 + t * 2 + (grp & 1)) * WG_SUB) + frag_off;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  issue(0, 0);
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    // stage st has landed (vmcnt(0) of every wave + barrier); the barrier also says every wave is done with stage st - 1,
    // whose buffer the DMA issued below overwrites
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (st + 1 < nst) issue(st + 1, buf ^ 1);
    const unsigned boff = (unsigned)(buf * C::STAGE);
    // fragment pipeline: the transposing reads of k step ks+1 are in flight while the MFMAs of ks issue.  The reads are
    // inline asm (no builtin), so the compiler neither knows they are asynchronous nor places a wait: counted waits by
    // hand, fenced with sched_barrier(0) -- an MFMA is a register-only instruction that the "memory" clobber does not
    // order, and hipcc did hoist it above the wait.
    constexpr int NKS = 2 * WG_PB;
    unsigned long long af[2][TM][2], bf[2][TN][2];   // [set][mfma tile][k half]
    auto rd = [&](int ks, int set) {
      const unsigned ka = boff + (unsigned)((ks >> 1) * C::SUB_M * WG_SUB + (ks & 1) * 512);   // + 16 rows per k step, next block after 2
      const unsigned kb = boff + (unsigned)((ks >> 1) * C::SUB_N * WG_SUB + (ks & 1) * 512);
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        af[set][t][0] = tr_read(a_addr[t] + ka);
        af[set][t][1] = tr_read_128(a_addr[t] + ka);
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        bf[set][t][0] = tr_read(b_addr[t] + kb);
        bf[set][t][1] = tr_read_128(b_addr[t] + kb);
      }
    };
    rd(0, 0);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (ks + 1 < NKS) rd(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < NKS)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (TM + TN)) : "memory");       // the reads of ks+1 may stay outstanding
      else
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
          typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2;
          const u64x2 av = {af[ks & 1][mi][0], af[ks & 1][mi][1]};
          const u64x2 bv = {bf[ks & 1][ni][0], bf[ks & 1][ni][1]};
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                                acc[mi][ni], 0, 0, 0);
        }
    }
  }

  // ---- epilogue: acc[mi][ni][e] = D[m][n], m = 8*(e/4) + 4*(lane/32) + e%4 (ci), n = lane%32 (co): atomics coalesced along co
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int co = co0 + (wn * TN + ni) * 32 + l31;
      if (co >= a.cout) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = ci0 + (wm * TM + mi) * 32 + 8 * (e >> 2) + 4 * lhi + (e & 3);
        if (ci < a.cin) unsafeAtomicAdd(a.out + ((long long)tap * a.cin + ci) * a.cout + co, acc[mi][ni][e]);
      }
    }
}

template <int WM, int WN, int TM, int TN>
int wg_launch(WGArgs& a, const sm_conv_desc* d, long long P, hipStream_t s) {
  using C = WGCfg<WM, WN, TM, TN>;
  a.ci_tiles = (d->cin + C::TILE_M - 1) / C::TILE_M;
  const int co_tiles = (d->cout + C::TILE_N - 1) / C::TILE_N;
  a.co_tiles = co_tiles;
  const long long tiles = (long long)d->kh * d->kw * a.ci_tiles * co_tiles;
  // split K: two rounds of the resident blocks (2 per CU for the 4-wave tile, 1 for the 8-wave tile); every slice ends in
  // TILE_M x TILE_N float atomics per tile, so slices stay >= 512 positions
  const long long resident = 256 * (C::NW <= 4 ? 2 : 1);
  long long S = (2 * resident + tiles - 1) / tiles;
  const long long s_max = (P + 511) / 512;
  if (S > s_max) S = s_max;
  if (S > 512) S = 512;
  if (S < 1) S = 1;
  const long long slice = ((P + S - 1) / S + WG_KC - 1) / WG_KC * WG_KC;
  S = (P + slice - 1) / slice;
  a.slice = (int)slice;
  if (sm_lds_optin((const void*)wgrad_direct_kernel<WM, WN, TM, TN>, C::LDS) != hipSuccess) return SM_ERR_LAUNCH;
  a.per_slice = (int)tiles;
  const long long nblk = 8 * ((S + 7) / 8) * tiles;          // slices beyond S exit at once
  if (nblk > 0x7fffffffLL) return SM_ERR_BAD_SHAPE;
  hipLaunchKernelGGL((wgrad_direct_kernel<WM, WN, TM, TN>), dim3((unsigned)nblk), dim3(C::THREADS), C::LDS, s, a);
  return SM_OK;
}

}  // namespace

extern "C" int sm_wgrad_direct_supported(const sm_conv_desc* d) {
  if (!d || d->nlev < 1 || d->nlev > SM_MAX_LEVELS || d->batch < 1) return 0;
  if (d->cin % 8 != 0 || d->cout % 8 != 0 || d->in_cstride % 8 != 0 || d->out_cstride % 8 != 0) return 0;
  if (d->kh < 1 || d->kw < 1 || d->stride < 1) return 0;
  return 1;
}

// Where the direct kernel beats the im2col^T GEMM path (tools/wgrad_bench.py, B=4 800x1344 shapes): the position axis must
// be long enough to amortise the float-atomic epilogue of every slice, and a 3x3 conv saves 9x the transposition traffic of
// a 1x1 -- >= 16 000 positions for k >= 3 (layer3's 3x3 convs: 197 vs 185 TF/s), >= 60 000 for narrow 1x1 convs; below that
// the GEMM path wins (layer4 1x1 512 -> 2048 at 4 200 positions: 142 vs 182 TF/s).  profiles/r02k_wgrad_direct_vs_gemm.txt
extern "C" int sm_wgrad_direct_preferred(const sm_conv_desc* d) {
  if (!sm_wgrad_direct_supported(d)) return 0;
  long long P = 0;
  for (int l = 0; l < d->nlev; ++l) P += (long long)d->batch * d->out_h[l] * d->out_w[l];
  if (d->kh * d->kw >= 9) return P >= 16000 ? 1 : 0;
  // 1x1: the transposition it saves is only 1x the activation; wins for the narrow convs of layer2 (512 -> 128: 113 vs 86
  // TF/s), loses 5 % on the mask branch's 768 -> 512
  return (P >= 60000 && (long long)d->cin * d->cout <= 131072) ? 1 : 0;
}

extern "C" int sm_wgrad_direct(const sm_conv_desc* d, const void* x, const void* gout, float* grad_w_t, sm_stream_t stream) {
  if (!d || !x || !gout || !grad_w_t) return SM_ERR_BAD_ARG;
  if (!sm_wgrad_direct_supported(d)) return SM_ERR_UNSUPPORTED;
  WGArgs a;
  a.x = (const uint16_t*)x;
  a.g = (const uint16_t*)gout;
  a.out = grad_w_t;
  a.nlev = d->nlev;
  a.batch = d->batch;
  long long P = 0;
  for (int l = 0; l < SM_MAX_LEVELS; ++l) {
    const bool on = l < d->nlev;
    a.in_h[l] = on ? d->in_h[l] : 1, a.in_w[l] = on ? d->in_w[l] : 1;
    a.out_h[l] = on ? d->out_h[l] : 1, a.out_w[l] = on ? d->out_w[l] : 1;
    a.in_row0[l] = on ? d->in_row0[l] : 0, a.out_row0[l] = on ? d->out_row0[l] : 0;
    if (on) {
      const long long n = (long long)d->batch * d->out_h[l] * d->out_w[l];
      if (n >= (1ll << 31)) return SM_ERR_UNSUPPORTED;
      P += n;
    }
    a.pos_end[l] = P;
  }
  a.cin = d->cin, a.cout = d->cout, a.kh = d->kh, a.kw = d->kw, a.stride = d->stride, a.pad = d->pad;
  a.dil = d->dil > 0 ? d->dil : 1;
  a.in_cstride = d->in_cstride, a.g_cstride = d->out_cstride;
  a.P = P;
  const long long K = (long long)d->kh * d->kw * d->cin;
  hipStream_t s = sm_hip_stream(stream);
  if (sm_zero_async(grad_w_t, sizeof(float) * K * d->cout, s) != hipSuccess) return SM_ERR_LAUNCH;
  if (P == 0) return SM_OK;
  // the 256 x 256 tile when both channel counts fill it (>= 3/4) and the position axis feeds >= 512 positions to each of
  // its fewer, larger blocks (tower 3x3 at 89 600 positions: 391 vs 297 TF/s; layer3 3x3 at 16 800: 175 vs 199)
  const bool big = !(d->flags & SM_CONV_BWD_WGRAD_TILE128) && d->cin >= 192 && d->cout >= 192 && P >= 40000;
  const int lrc = big ? wg_launch<2, 4, 4, 2>(a, d, P, s) : wg_launch<2, 2, 2, 2>(a, d, P, s);
  if (lrc != SM_OK) return lrc;
  SM_LAUNCH_CHECK();
  return SM_OK;
}
