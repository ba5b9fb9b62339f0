// Implicit-GEMM convolution for gfx950 (MI355X): NHWC bf16 activations, K-major bf16
// weights, v_mfma_f32_32x32x16_bf16 with f32 accumulation, fused epilogue.
//
// GEMM view:  D[cout][pos] = sum_k W[cout][k] * X[pos][k],  k = (kh, kw, cin) cin fastest.
//   MFMA A operand = weight tile  (rows = cout), B operand = activation tile (cols = pos):
//   with the 32x32 C/D layout (col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)) each
//   lane then owns 4 CONSECUTIVE couts of one position -> 8-byte NHWC stores.
// Tiles: BCO x BPOS x 64(K) per step, 256 threads = 4 wave64; both operands are staged
//   global -> VGPR -> LDS (issue-early / write-late, guide T14) into a double buffer with
//   a 16-byte-chunk XOR swizzle (slot = chunk ^ ((row>>1)&7)) so the ds_read_b128
//   fragment reads of 128-byte rows are bank-conflict free.
// Multi-level: the 5 FPN levels share tower weights, so one launch covers all levels
//   (M tiles are enumerated per level; tiles never straddle a level).
// DEFORM variant: the activation loader performs the deformable bilinear gather
//   (deform_conv_cuda_kernel.cu:85-115,191-243) -- the column buffer never exists.
#include <utility>

#include "common.h"
#include "experiments.h"

// conv_igemm_f16.o (this file compiled with -DSM_OPERAND_F16)
int sm_conv_igemm_f16(const sm_conv_desc* d, const void* x, const void* w, const float* bias, void* y, hipStream_t stream,
                      unsigned long long* gn_stats, void* workspace, long long workspace_bytes);
// deform_patch.hip
bool sm_deform_patch_supported(const sm_conv_desc* d);
int sm_deform_patch_launch(const sm_conv_desc* d, const void* x, const float* offset, const void* w, const float* bias,
                           void* y, hipStream_t stream, unsigned long long* gn_stats, long long k_padded);

// Operand type of this translation unit.  conv_igemm.hip is compiled TWICE (csrc/Makefile): as is -- bf16 operands,
// v_mfma_f32_32x32x16_bf16 -- and with -DSM_OPERAND_F16 into conv_igemm_f16.o -- IEEE binary16 operands on
// v_mfma_f32_32x32x16_f16, the kernels behind SM_CONV_F16 (the split-precision head plan, include/sipmask_hip.h).  The
// operands are 16-bit payloads that the LDS-DMA loader never interprets, so the two builds differ in the MFMA
// instruction only; the f16 build carries the LDS-DMA kernels with f32 output and nothing else (no deformable gather,
// no input ReLU, no bf16 residual / output) and exports one C++ entry point instead of the C ABI.
#ifdef SM_OPERAND_F16
typedef _Float16 frag8 __attribute__((ext_vector_type(8)));
#define SM_MFMA_32x32x16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16((A), (B), (C), 0, 0, 0)
#else
typedef bf16x8 frag8;
#define SM_MFMA_32x32x16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16((A), (B), (C), 0, 0, 0)
#endif

extern "C" int sm_conv_cout_tile(int cout);

namespace {

struct ConvKArgs {
  const uint16_t* x;
  const uint16_t* w;
  const float* bias;
  const uint16_t* res;
  void* y;
  const float* offset;
  int nlev, batch;
  int in_h[SM_MAX_LEVELS], in_w[SM_MAX_LEVELS], out_h[SM_MAX_LEVELS], out_w[SM_MAX_LEVELS];
  long long in_row0[SM_MAX_LEVELS], out_row0[SM_MAX_LEVELS], res_row0[SM_MAX_LEVELS];
  int res_h[SM_MAX_LEVELS], res_w[SM_MAX_LEVELS];
  int tile0[SM_MAX_LEVELS + 1];
  int cin, cout, kh, kw, stride, pad, dil;
  int in_cstride, out_cstride, out_coff, res_cstride;
  int Kp, nchunk, cpt, ntn, nk;
  unsigned flags;
  int scale_nch;
  float level_scale[SM_MAX_LEVELS];
  float acc_scale;   // accumulators are multiplied by this before bias / Scale (1 = none; sm_conv_desc.acc_scale)
  int dg, cpg8;  // deform groups, chunks (of 8 ch) per deform group
  unsigned long long* gn_stats;  // optional fused GroupNorm statistics [batch][nlev][cout/8][2] (sum, sum of squares), fixed point (common.h: gn_fix)
  long long w_bstride;  // elements between the weight matrices of consecutive images (0 = shared): batched / split-K GEMMs
  // group dimension (64-wide-K LDS-DMA kernel only): ngroups problems of identical shape in one launch, e.g. the cls
  // and reg tower convs of one depth.  Tiles [g*tpg, (g+1)*tpg) belong to group g; its operands sit at fixed offsets.
  int ngroups, tpg;
  long long x_grows, y_grows;        // rows between the groups' inputs (0 = shared) / outputs (and RES_ADD residuals)
  long long w_gstride, b_gstride, gn_gstride;   // elements between the groups' weights / biases / GN statistics
  // split-K (64-wide-K LDS-DMA kernel, flat loop): ksplit > 1 -> block b works on K steps [ks*nk/S, (ks+1)*nk/S) of tile
  // b / S and stores its raw f32 accumulators to part[ks][row][cout_pad]; splitk_reduce_kernel applies the epilogue
  int ksplit;
  float* part;
  long long part_rows;   // rows of one partial slab
};

__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {
  uint32_t neg = (v >> 15) & 0x00010001u;
  return v & ~(neg * 0xffffu);
}

// Staging registers: native clang vectors (u32x4), NOT HIP's uint4 class -- an assignment
// `uint4 r = *global_ptr` is emitted as llvm.memcpy(private <- global), which SROA cannot
// promote, so the staging array lands in scratch memory (scratch_store/scratch_load plus an
// immediate vmcnt wait inside the K loop: -25% on the tower convs).  Named members with
// compile-time accessors keep every register statically addressed.
struct Stage8 {
  u32x4 r0, r1, r2, r3, r4, r5, r6, r7;
  template <int I>
  __device__ __forceinline__ u32x4& at() {
    if constexpr (I == 0) return r0;
    else if constexpr (I == 1) return r1;
    else if constexpr (I == 2) return r2;
    else if constexpr (I == 3) return r3;
    else if constexpr (I == 4) return r4;
    else if constexpr (I == 5) return r5;
    else if constexpr (I == 6) return r6;
    else return r7;
  }
};

template <int N, typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// 16 zero bytes in device memory: the source of every zero-padding / K-padding chunk in the
// LDS-DMA loader (an LDS-DMA load cannot be masked, but it can be pointed at zeros)
__device__ __attribute__((aligned(16))) const unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

// PROD = 1: warp-specialised variant of the LDS-DMA kernel -- 8 waves, waves 0-3 only issue MFMAs + fragment
// reads (consumers), waves 4-7 only issue the LDS-DMA loads (producers), so a wave never stalls its MFMA stream
// on the DMA issue port (the two phases of the 4-wave kernel barely overlap: DESIGN.md section 6).
// OPT (LDS-DMA path, cin >= 64 only): bit 0 = control-flow-free loader + peeled K loop (one basic block per K step),
// bit 1 = fragment reads of K sub-step kk+1 issued under the MFMAs of kk (second register set).
template <int WCO, int WPOS, int TCO, int TPOS, bool DEFORM, bool DMA, int PROD = 0, int OPT = 0>
__global__ __launch_bounds__(64 * WCO * WPOS * (1 + PROD), (WCO * WPOS == 4) ? 2 : 1) void conv_igemm_kernel(const ConvKArgs a) {
  static_assert(!PROD || (DMA && !DEFORM), "producer/consumer split exists for the LDS-DMA path only");
  constexpr int NTHR = 64 * WCO * WPOS;            // threads of one role group: 4 waves, or 8 for the 256x256 tile
  constexpr int THREADS = NTHR * (1 + PROD);
  constexpr int LROWS = NTHR / 8;                  // tile rows one loader pass covers (8 16-byte chunks per 64-wide K row)
  constexpr int BCO = WCO * TCO * 32;
  constexpr int BPOS = WPOS * TPOS * 32;
  constexpr int NW = BCO / LROWS;   // 16-byte weight chunks per thread per K step
  constexpr int NX = BPOS / LROWS;  // 16-byte activation chunks per thread per K step
  constexpr int STAGE = (BCO + BPOS) * 128;
  constexpr int EPI_LD = BCO + 4;                  // padded f32 row of the epilogue staging tile
  constexpr int EPI_BYTES = BPOS * EPI_LD * 4;
  constexpr int GN_SEG = 4;                        // images a tile may span before falling back to global atomics
  constexpr int GN_BYTES = GN_SEG * (BCO / 8) * 2 * 8;   // 64-bit fixed-point bins
  // tiles above 128x128 exist with the register epilogue only (the launcher checks its alignment conditions): their
  // f32 staging tile would not fit beside nothing, and they are picked for the reuse, not for odd shapes
  constexpr bool REG_ONLY = BCO * BPOS > 128 * 128;
  constexpr int SMEM_MAIN = (REG_ONLY || 2 * STAGE > EPI_BYTES) ? 2 * STAGE : EPI_BYTES;
  constexpr int SMEM_BYTES = SMEM_MAIN + GN_BYTES;  // ONE LDS object (a second one de-pipelines the DMA loop)
  static_assert(WCO * WPOS == 4 || (WCO * WPOS == 8 && !PROD), "4 waves, or 8 without the producer split");
  static_assert(NW <= 8 && NX <= 8, "Stage8 holds 8 chunks");
  static_assert(OPT == 0 || (DMA && !PROD), "OPT variants exist for the plain LDS-DMA loop only");
  static_assert(WCO * WPOS == 4 || (OPT & 3) == 3 || !DMA, "the 8-wave LDS-DMA tile is built on the flat, pipelined loop");
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

  const int gtid = threadIdx.x;                       // 0..THREADS-1 (epilogue work split)
  const bool is_prod = PROD && gtid >= NTHR;          // wave-uniform role
  const bool is_cons = !PROD || gtid < NTHR;
  const int tid = PROD ? (gtid & (NTHR - 1)) : gtid;   // index inside the role group: loader row/chunk, MFMA wave
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wco = wave / WPOS;
  const int wpos = wave % WPOS;
  // K chunk (8 bf16) of this thread inside the 64-wide K step.  Register path: logical chunk
  // tid&7, written to the swizzled LDS slot.  LDS-DMA path: the hardware writes lane L of a wave
  // to (wave-uniform base + 16*L), i.e. PHYSICAL slot tid&7 of row tid>>3, so the thread must
  // FETCH the logical chunk that belongs there: the swizzle moves to the source side (guide rule 21).
  const int j = DMA ? ((tid & 7) ^ (((tid >> 3) >> 1) & 7)) : (tid & 7);
  const int r0 = tid >> 3;  // tile row handled by this thread (+LROWS*i)
  const int wslot = (j ^ ((r0 >> 1) & 7)) * 16;

  // ---- tile decode (wave-uniform).  Blocks are dispatched round-robin over the 8 XCDs
  // (block b -> XCD b%8, observed); give every XCD one CONTIGUOUS range of tiles so that the
  // 3x3 halo rows and both cout tiles of a position tile hit the same 4 MiB L2 (guide T1,
  // bijective form).  Placement only affects speed, never correctness.
  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, xq = nblk >> 3, xr = nblk & 7;
  const int tlin_s = (a.flags & SM_CONV_DBG_LINEAR_TILES)
                         ? (int)blockIdx.x
                         : (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (int)(blockIdx.x >> 3);
  // split-K: consecutive blocks are the K slices of one tile
  const int ks = a.ksplit > 1 ? tlin_s % a.ksplit : 0;
  const int tlin = a.ksplit > 1 ? tlin_s / a.ksplit : tlin_s;
  const int kt0 = a.ksplit > 1 ? (int)((long long)ks * a.nk / a.ksplit) : 0;
  // group decode (wave-uniform): which problem instance this tile belongs to
  const int grp = a.ngroups > 1 ? tlin / a.tpg : 0;
  const int tl_g = tlin - grp * a.tpg;
  const int nt = tl_g % a.ntn;
  const int mt = tl_g / a.ntn;
  const float* const biasp = a.bias != nullptr ? a.bias + grp * a.b_gstride : nullptr;
  unsigned long long* const gnp = a.gn_stats != nullptr ? a.gn_stats + grp * a.gn_gstride : nullptr;
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && mt >= a.tile0[l]) lev = l;
  const int H = a.in_h[lev], W = a.in_w[lev], Ho = a.out_h[lev], Wo = a.out_w[lev];
  const int HoWo = Ho * Wo;
  const int M = a.batch * HoWo;
  const int m0 = (mt - a.tile0[lev]) * BPOS;
  const long long in_row0 = a.in_row0[lev] + grp * a.x_grows;

  // ---- per-row gather bases (rows r0 + 32*i of the activation tile)
  int rbase[NX], rhi[NX], rwi[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    int m = m0 + r0 + LROWS * i;
    if (m < M) {
      int n = m / HoWo;
      int rem = m - n * HoWo;
      int ho = rem / Wo;
      int wo = rem - ho * Wo;
      rbase[i] = n * H * W;
      rhi[i] = ho * a.stride - a.pad;
      rwi[i] = wo * a.stride - a.pad;
    } else {
      rbase[i] = -1;
      rhi[i] = 0;
      rwi[i] = 0;
    }
  }
  // element offset of each row's (kh=0,kw=0) tap; rows beyond M get a row coordinate that fails
  // every bounds test, so no separate validity flag is needed in the K loop
  long long xoff[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    if (rbase[i] < 0) rhi[i] = -0x40000000;
    xoff[i] = (in_row0 + (rbase[i] < 0 ? 0 : rbase[i]) + (long long)rhi[i] * W + rwi[i]) * a.in_cstride;
  }
  const uint16_t* wrow = a.w + grp * a.w_gstride + (long long)(nt * BCO + r0) * a.Kp + j * 8 +
                         (a.w_bstride != 0 ? (long long)(m0 / HoWo) * a.w_bstride : 0ll);
  const long long wstride = (long long)LROWS * a.Kp;
  // loader K state (this thread's 16-byte chunk j of the current K step), advanced incrementally:
  // no integer division inside the K loop when a tap holds >= 8 chunks (every layer but the stem)
  int ld_cc, ld_kh, ld_kw, ld_kc = j + 8 * kt0;
  {
    const int tap0 = ld_kc / a.cpt;
    ld_cc = ld_kc - tap0 * a.cpt;
    ld_kh = tap0 / a.kw;
    ld_kw = tap0 - ld_kh * a.kw;
  }
  const uint16_t* ld_wp = wrow + 64 * kt0;
  auto advance_k = [&]() {
    ld_kc += 8;
    ld_wp += 64;
    if (a.cpt >= 8) {
      ld_cc += 8;
      if (ld_cc >= a.cpt) {
        ld_cc -= a.cpt;
        if (++ld_kw == a.kw) {
          ld_kw = 0;
          ++ld_kh;
        }
      }
    } else {
      const int tap = ld_kc / a.cpt;
      ld_cc = ld_kc - tap * a.cpt;
      ld_kh = tap / a.kw;
      ld_kw = tap - ld_kh * a.kw;
    }
  };

  // two staging register sets: plain convs keep TWO K tiles of global loads in flight (tile kt+1
  // and kt+2 while tile kt is on the MFMAs); the deformable variant uses set A only
  Stage8 wregA, xregA, wregB, xregB;
  // deformable gather state between "issue" (before the MFMAs) and "finish" (after them)
  constexpr bool DSPLIT = DEFORM && (NX <= 4);
  Stage8 cqa, cqb;                 // 16 corner chunks (rows i, corners 0..3) when DSPLIT
  // PF2 (A/B, compiled out): TWO K steps of corner loads in flight for deformable tiles with <= 2 rows per thread (set
  // 0 in cqa, set 1 in cqb).  Measured on the 8-wave 256 x 128 FeatureAlign tile: 0.330 ms vs 0.275 ms without -- the
  // K step is bound by the loader's VALU (setup + blend) and its registers, not by gather latency.
  constexpr bool PF2 = false && DEFORM && NX <= 2;
  float cw[2][DSPLIT ? NX : 1][4];    // corner weights per set (0 where the corner / sample is invalid)
  using CS0 = std::integral_constant<int, 0>;
  using CS1 = std::integral_constant<int, 1>;
  uint32_t xmaskA[NX], xmaskB[NX];  // plain conv: all-ones where the tap is inside the image
  const uint32_t relu_m = (a.flags & SM_CONV_IN_RELU) ? 0xffffu : 0u;

  // NOTE: load_w / load_x consume the loader K state; call them as a pair, in K order, then advance_k()
  auto load_w = [&](int, Stage8& wreg) {
    static_for<NW>([&](auto I) {
      constexpr int i = decltype(I)::value;
      wreg.template at<i>() = *reinterpret_cast<const u32x4*>(ld_wp + i * wstride);
    });
  };

  // deformable offsets, prefetched one K step ahead (see load_x)
  float2 off_pf[DEFORM ? NX : 1];
  bool off_ready = false;
  auto issue_off = [&](int kc, int kh_, int kw_, int cc_, float2 (&dst)[DEFORM ? NX : 1]) {
    if constexpr (DEFORM) {
      const int ntap = a.kh * a.kw;
      const int tap = kh_ * a.kw + kw_;
      const int g = cc_ / a.cpg8;
      const bool kvalid = kc < a.nchunk;
      const long long orow0 = a.out_row0[lev] + m0 + r0;
      static_for<NX>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const bool ok = kvalid && rhi[i] > -0x20000000;
        const long long oo = ok ? (orow0 + LROWS * i) * (long long)(a.dg * ntap * 2) + (g * ntap + tap) * 2 : 0ll;
        dst[i] = *reinterpret_cast<const float2*>(a.offset + oo);
      });
    }
  };
  // All global loads below are UNCONDITIONAL (clamped address + select): a load inside an
  // exec-masked branch makes hipcc wait vmcnt(0) at the join, which serialises the row loads.
  auto load_x = [&](int, Stage8& xreg, uint32_t (&xmask)[NX], auto CS) {
    constexpr int cs = decltype(CS)::value;
    const int kc = ld_kc;
    const int c0 = ld_cc * 8;
    const bool kvalid = kc < a.nchunk;
    const int dh = ld_kh * a.dil, dw = ld_kw * a.dil;
    if constexpr (!DEFORM) {
      // branch-free: select on the 32-bit row index, AND-masks for zero padding and input ReLU
      const long long toff = (long long)((dh * W + dw) * a.in_cstride + c0);   // same for all rows
      static_for<NX>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int hi = rhi[i] + dh, wi = rwi[i] + dw;
        const bool ok = kvalid && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        const long long eo = ok ? xoff[i] + toff : 0ll;
        // raw load now; the padding / ReLU masks are applied in finish_x() AFTER the MFMAs, so
        // the wait for this load sits behind the matrix work instead of in front of it
        xreg.template at<i>() = *reinterpret_cast<const u32x4*>(a.x + eo);
        xmask[i] = ok ? 0xffffffffu : 0u;
      });
    } else {
      // deformable bilinear gather (deform_conv_cuda_kernel.cu:85-115,216-229); the offset row is
      // the output row (stride-1 "same" conv).  Phase 1: offsets of all rows, phase 2: all corner
      // loads, phase 3 (finish_x, after the MFMAs when DSPLIT): blend to bf16.
      // the offsets of THIS K step were requested one step ago (off_pf); request the next step's now, so that the
      // dependent chain of a step is corner loads -> blend only (offset load -> address -> corner load -> blend was two
      // exposed memory latencies per K step at one block per CU)
      float2 off[NX];
      if (!off_ready) {
        issue_off(kc, ld_kh, ld_kw, ld_cc, off_pf);
        off_ready = true;
      }
      static_for<NX>([&](auto I) { off[decltype(I)::value] = off_pf[decltype(I)::value]; });
      {
        int ncc = ld_cc + 8, nkh = ld_kh, nkw = ld_kw;
        if (a.cpt >= 8) {
          if (ncc >= a.cpt) {
            ncc -= a.cpt;
            if (++nkw == a.kw) {
              nkw = 0;
              ++nkh;
            }
          }
        } else {
          const int ntp = (kc + 8) / a.cpt;
          ncc = (kc + 8) - ntp * a.cpt;
          nkh = ntp / a.kw;
          nkw = ntp - nkh * a.kw;
        }
        issue_off(kc + 8, nkh, nkw, ncc, off_pf);
      }
      static_for<NX>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const bool ok = kvalid && rhi[i] > -0x20000000;
        const float h_im = (float)(rhi[i] + dh) + off[i].x;
        const float w_im = (float)(rwi[i] + dw) + off[i].y;
        const bool inr = ok && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
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
        const int hl = min(max(h_low, 0), H - 1), hh_ = min(max(h_high, 0), H - 1);
        const int wl = min(max(w_low, 0), W - 1), wh_ = min(max(w_high, 0), W - 1);
        const uint16_t* base = a.x + (in_row0 + (ok ? rbase[i] : 0)) * a.in_cstride + (ok ? c0 : 0);
        const u32x4 q1 = *reinterpret_cast<const u32x4*>(base + (long long)(hl * W + wl) * a.in_cstride);
        const u32x4 q2 = *reinterpret_cast<const u32x4*>(base + (long long)(hl * W + wh_) * a.in_cstride);
        const u32x4 q3 = *reinterpret_cast<const u32x4*>(base + (long long)(hh_ * W + wl) * a.in_cstride);
        const u32x4 q4 = *reinterpret_cast<const u32x4*>(base + (long long)(hh_ * W + wh_) * a.in_cstride);
        if constexpr (DSPLIT) {
          cw[cs][i][0] = w1;
          cw[cs][i][1] = w2;
          cw[cs][i][2] = w3;
          cw[cs][i][3] = w4;
          if constexpr (PF2 && cs == 1) {
            cqb.template at<4 * i + 0>() = q1;
            cqb.template at<4 * i + 1>() = q2;
            cqb.template at<4 * i + 2>() = q3;
            cqb.template at<4 * i + 3>() = q4;
          } else if constexpr (i < 2) {
            cqa.template at<4 * i + 0>() = q1;
            cqa.template at<4 * i + 1>() = q2;
            cqa.template at<4 * i + 2>() = q3;
            cqa.template at<4 * i + 3>() = q4;
          } else {
            cqb.template at<4 * (i - 2) + 0>() = q1;
            cqb.template at<4 * (i - 2) + 1>() = q2;
            cqb.template at<4 * (i - 2) + 2>() = q3;
            cqb.template at<4 * (i - 2) + 3>() = q4;
          }
        } else {
          float f1[8], f2[8], f3[8], f4[8], r[8];
          unpack_bf16x8(q1, f1);
          unpack_bf16x8(q2, f2);
          unpack_bf16x8(q3, f3);
          unpack_bf16x8(q4, f4);
#pragma unroll
          for (int e = 0; e < 8; ++e) r[e] = w1 * f1[e] + w2 * f2[e] + w3 * f3[e] + w4 * f4[e];
          xreg.template at<i>() = pack_bf16x8_v(r);
        }
      });
    }
  };

  auto finish_x = [&](Stage8& xreg, uint32_t (&xmask)[NX], auto CS) {
    constexpr int cs = decltype(CS)::value;
    if constexpr (!DEFORM) {
      static_for<NX>([&](auto I) {
        constexpr int i = decltype(I)::value;
        u32x4 v = xreg.template at<i>();
        const uint32_t m = xmask[i];
        if (relu_m) {   // block-uniform; only the FPN P7 conv takes it
          v.x &= ~(((v.x >> 15) & 0x00010001u) * relu_m);
          v.y &= ~(((v.y >> 15) & 0x00010001u) * relu_m);
          v.z &= ~(((v.z >> 15) & 0x00010001u) * relu_m);
          v.w &= ~(((v.w >> 15) & 0x00010001u) * relu_m);
        }
        v.x &= m;
        v.y &= m;
        v.z &= m;
        v.w &= m;
        xreg.template at<i>() = v;
      });
    }
    if constexpr (DSPLIT) {
      static_for<NX>([&](auto I) {
        constexpr int i = decltype(I)::value;
        float f1[8], f2[8], f3[8], f4[8], r[8];
        if constexpr (PF2 && cs == 1) {
          unpack_bf16x8(cqb.template at<4 * i + 0>(), f1);
          unpack_bf16x8(cqb.template at<4 * i + 1>(), f2);
          unpack_bf16x8(cqb.template at<4 * i + 2>(), f3);
          unpack_bf16x8(cqb.template at<4 * i + 3>(), f4);
        } else if constexpr (i < 2) {
          unpack_bf16x8(cqa.template at<4 * i + 0>(), f1);
          unpack_bf16x8(cqa.template at<4 * i + 1>(), f2);
          unpack_bf16x8(cqa.template at<4 * i + 2>(), f3);
          unpack_bf16x8(cqa.template at<4 * i + 3>(), f4);
        } else {
          unpack_bf16x8(cqb.template at<4 * (i - 2) + 0>(), f1);
          unpack_bf16x8(cqb.template at<4 * (i - 2) + 1>(), f2);
          unpack_bf16x8(cqb.template at<4 * (i - 2) + 2>(), f3);
          unpack_bf16x8(cqb.template at<4 * (i - 2) + 3>(), f4);
        }
        const float w1 = cw[cs][i][0], w2 = cw[cs][i][1], w3 = cw[cs][i][2], w4 = cw[cs][i][3];
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = w1 * f1[e] + w2 * f2[e] + w3 * f3[e] + w4 * f4[e];
        xreg.template at<i>() = pack_bf16x8_v(r);
      });
    }
  };

  auto store_tile = [&](int buf, Stage8& wreg, Stage8& xreg) {
    unsigned char* Wb = smem + buf * STAGE;
    unsigned char* Xb = Wb + BCO * 128;
    static_for<NW>([&](auto I) {
      constexpr int i = decltype(I)::value;
      *reinterpret_cast<u32x4*>(Wb + (r0 + LROWS * i) * 128 + wslot) = wreg.template at<i>();
    });
    static_for<NX>([&](auto I) {
      constexpr int i = decltype(I)::value;
      *reinterpret_cast<u32x4*>(Xb + (r0 + LROWS * i) * 128 + wslot) = xreg.template at<i>();
    });
  };

  f32x16 acc[TCO][TPOS];
#pragma unroll
  for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
    for (int tp = 0; tp < TPOS; ++tp)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tc][tp][e] = 0.f;

  const int l31 = lane & 31;
  const int rsw = (l31 >> 1) & 7;
  const int khalf = lane >> 5;
  const int wrow_off = (wco * TCO * 32 + l31) * 128;
  const int xrow_off = BCO * 128 + (wpos * TPOS * 32 + l31) * 128;

  // OPT bit 1: fragments of K sub-step kk+1 are read into a second register set while the MFMAs of kk run
  // (hipcc otherwise re-uses one set: read -> lgkmcnt(0) -> 4 MFMAs, the LDS latency exposed every 128 MFMA cycles);
  // the sched_group_barrier ladder pins "1 MFMA, 1 ds_read" pairs so the reads issue in the MFMA shadows.
  constexpr bool FRAG_PIPE = (OPT & 2) != 0;
  constexpr bool FLAT_LOOP = (OPT & 1) != 0;
  auto compute = [&](int buf) {
    const unsigned char* S = smem + buf * STAGE;
    if constexpr (FRAG_PIPE) {
      frag8 wf[2][TCO], xf[2][TPOS];
      auto rd = [&](int kk, int set) {
        const int slot = ((kk * 2 + khalf) ^ rsw) * 16;
#pragma unroll
        for (int t = 0; t < TCO; ++t) wf[set][t] = *reinterpret_cast<const frag8*>(S + wrow_off + t * 32 * 128 + slot);
#pragma unroll
        for (int t = 0; t < TPOS; ++t) xf[set][t] = *reinterpret_cast<const frag8*>(S + xrow_off + t * 32 * 128 + slot);
      };
      rd(0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, TCO + TPOS, 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) rd(kk + 1, (kk + 1) & 1);
#pragma unroll
        for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
          for (int tp = 0; tp < TPOS; ++tp)
            acc[tc][tp] = SM_MFMA_32x32x16(wf[kk & 1][tc], xf[kk & 1][tp], acc[tc][tp]);
        // ladder: one fragment read of kk+1 behind each MFMA of kk
        constexpr int NFR = TCO + TPOS, NMF = TCO * TPOS, NPAIR = NFR < NMF ? NFR : NMF;
        if (kk < 3) {
#pragma unroll
          for (int i = 0; i < NPAIR; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          if constexpr (NFR > NPAIR) __builtin_amdgcn_sched_group_barrier(0x100, NFR - NPAIR, 0);
          if constexpr (NMF > NPAIR) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NPAIR, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
        }
      }
    } else {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int slot = ((kk * 2 + khalf) ^ rsw) * 16;
      frag8 wf[TCO], xf[TPOS];
#pragma unroll
      for (int t = 0; t < TCO; ++t) wf[t] = *reinterpret_cast<const frag8*>(S + wrow_off + t * 32 * 128 + slot);
#pragma unroll
      for (int t = 0; t < TPOS; ++t) xf[t] = *reinterpret_cast<const frag8*>(S + xrow_off + t * 32 * 128 + slot);
#pragma unroll
      for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
        for (int tp = 0; tp < TPOS; ++tp)
          acc[tc][tp] = SM_MFMA_32x32x16(wf[tc], xf[tp], acc[tc][tp]);
    }
    }
  };

  const int nk = a.ksplit > 1 ? (int)((long long)(ks + 1) * a.nk / a.ksplit) - kt0 : a.nk;
  if constexpr (DMA) {
    // ---- LDS-DMA main loop: global_load_lds_dwordx4 straight into the swizzled LDS image, no
    // staging VGPRs, no ds_write pass, no mask pass.  Two LDS stages; the __syncthreads() that ends
    // a K step also drains the DMA queue (hipcc emits vmcnt(0) in front of the barrier).
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const int wave_row = (tid >> 6) * 8;   // first tile row written by this wave (+LROWS*i)
    const unsigned long long zero_page = (unsigned long long)g_zero16;
    auto dma_tile = [&](int buf) {
      unsigned char* Wb = smem + buf * STAGE;
      unsigned char* Xb = Wb + BCO * 128;
      const int c0 = ld_cc * 8;
      const bool kvalid = ld_kc < a.nchunk;
      const int dh = ld_kh * a.dil, dw = ld_kw * a.dil;
      const long long toff = (long long)((dh * W + dw) * a.in_cstride + c0);
      static_for<NW>([&](auto I) {
        constexpr int i = decltype(I)::value;
        __builtin_amdgcn_global_load_lds((glb_void*)(ld_wp + i * wstride), (lds_void*)(Wb + (wave_row + LROWS * i) * 128),
                                         16, 0, 0);
      });
      static_for<NX>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int hi = rhi[i] + dh, wi = rwi[i] + dw;
        const bool ok = kvalid && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        // bitwise select of the 64-bit address: a ?: between the tensor and the zero page makes
        // hipcc emit TWO exec-masked DMA instructions (one per source) instead of one
        const unsigned long long pm = ok ? ~0ull : 0ull;
        const unsigned long long src = ((unsigned long long)(a.x + xoff[i] + toff) & pm) | (zero_page & ~pm);
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(Xb + (wave_row + LROWS * i) * 128), 16, 0, 0);
      });
      advance_k();
    };
    // OPT bit 0: the same loader without control flow (bitwise bounds test, no division path: the launcher only
    // picks these variants for cin >= 64), and the K loop peeled so that one K step -- DMA issue, fragment reads, MFMAs --
    // is ONE basic block the scheduler can interleave.
    const int wave_row_s = __builtin_amdgcn_readfirstlane(tid >> 6) * 8;   // in an SGPR: the M0 values become SALU work
    auto dma_tile_flat = [&](int buf) {
      unsigned char* Wb = smem + buf * STAGE + wave_row_s * 128;
      unsigned char* Xb = Wb + BCO * 128;
      const bool kvalid = ld_kc < a.nchunk;
      const int dh = ld_kh * a.dil, dw = ld_kw * a.dil;
      const long long toff = (long long)((dh * W + dw) * a.in_cstride + ld_cc * 8);
      static_for<NW>([&](auto I) {
        constexpr int i = decltype(I)::value;
        __builtin_amdgcn_global_load_lds((glb_void*)(ld_wp + i * wstride), (lds_void*)(Wb + LROWS * i * 128),
                                         16, 0, 0);
      });
      static_for<NX>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int hi = rhi[i] + dh, wi = rwi[i] + dw;
        const bool ok = kvalid & ((unsigned)hi < (unsigned)H) & ((unsigned)wi < (unsigned)W);
        const unsigned long long pm = ok ? ~0ull : 0ull;
        const unsigned long long src = ((unsigned long long)(a.x + xoff[i] + toff) & pm) | (zero_page & ~pm);
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(Xb + LROWS * i * 128), 16, 0, 0);
      });
      ld_kc += 8;
      ld_wp += 64;
      ld_cc += 8;
      const int wrap = ld_cc >= a.cpt ? 1 : 0;
      ld_cc -= wrap * a.cpt;
      ld_kw += wrap;
      const int wrap2 = ld_kw == a.kw ? 1 : 0;
      ld_kw -= wrap2 * a.kw;
      ld_kh += wrap2;
    };
    // OPT bit 2 (experiment): the K step written out with every instruction placed by hand -- MFMA m of sub-step kk is
    // followed by fragment read m of kk+1 and, in the first two sub-steps, by one LDS-DMA piece of the NEXT tile every
    // DMA_EVERY MFMAs; sched_barrier(0) after each slot pins the order.  The address VALU of a piece then runs in the
    // shadow of the MFMA issued just before it instead of in front of the whole K step (where, in the 8-wave tile,
    // both waves of a SIMD do it at the same time with the matrix pipe idle).
    constexpr bool HAND_PLACED = (OPT & 4) != 0;
    auto k_step_placed = [&](int buf, bool with_dma) {
      const unsigned char* S = smem + buf * STAGE;
      unsigned char* Wb = smem + (buf ^ 1) * STAGE + wave_row_s * 128;
      unsigned char* Xb = Wb + BCO * 128;
      constexpr int NFR = TCO + TPOS, NMF = TCO * TPOS, NP = NW + NX;
      constexpr int DMA_EVERY = (2 * NMF) / NP > 0 ? (2 * NMF) / NP : 1;
      static_assert(!HAND_PLACED || NFR <= NMF, "one fragment read per MFMA slot");
      frag8 wf[2][TCO], xf[2][TPOS];
      const bool kvalid = ld_kc < a.nchunk;
      const int dh = ld_kh * a.dil, dw = ld_kw * a.dil;
      const long long toff = (long long)((dh * W + dw) * a.in_cstride + ld_cc * 8);
      auto rd1 = [&](int kk, int set, int f) {                       // fragment f of sub-step kk -> register set
        const int slot = ((kk * 2 + khalf) ^ rsw) * 16;
        if (f < TCO) wf[set][f] = *reinterpret_cast<const frag8*>(S + wrow_off + f * 32 * 128 + slot);
        else xf[set][f - TCO] = *reinterpret_cast<const frag8*>(S + xrow_off + (f - TCO) * 32 * 128 + slot);
      };
      static_for<NFR>([&](auto F) { rd1(0, 0, decltype(F)::value); });
      __builtin_amdgcn_sched_barrier(0);
      static_for<4>([&](auto KK) {
        constexpr int kk = decltype(KK)::value;
        static_for<NMF>([&](auto MM) {
          constexpr int m = decltype(MM)::value;
          constexpr int tc = m / TPOS, tp = m % TPOS;
          acc[tc][tp] = SM_MFMA_32x32x16(wf[kk & 1][tc], xf[kk & 1][tp], acc[tc][tp]);
          if constexpr (kk < 3 && m < NFR) rd1(kk + 1, (kk + 1) & 1, m);
          constexpr int slot_idx = kk * NMF + m;
          if constexpr (kk < 2 && (slot_idx % DMA_EVERY) == DMA_EVERY - 1 && slot_idx / DMA_EVERY < NP) {
            constexpr int pc = slot_idx / DMA_EVERY;
            if (with_dma) {                                           // block-uniform: false in the peeled last K step only
              if constexpr (pc < NW) {
                __builtin_amdgcn_global_load_lds((glb_void*)(ld_wp + pc * wstride), (lds_void*)(Wb + LROWS * pc * 128), 16, 0, 0);
              } else {
                constexpr int i = pc - NW;
                const int hi = rhi[i] + dh, wi = rwi[i] + dw;
                const bool ok = kvalid & ((unsigned)hi < (unsigned)H) & ((unsigned)wi < (unsigned)W);
                const unsigned long long pm = ok ? ~0ull : 0ull;
                const unsigned long long src = ((unsigned long long)(a.x + xoff[i] + toff) & pm) | (zero_page & ~pm);
                __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(Xb + LROWS * i * 128), 16, 0, 0);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      });
      if (with_dma) {
        ld_kc += 8;
        ld_wp += 64;
        ld_cc += 8;
        const int wrap = ld_cc >= a.cpt ? 1 : 0;
        ld_cc -= wrap * a.cpt;
        ld_kw += wrap;
        const int wrap2 = ld_kw == a.kw ? 1 : 0;
        ld_kw -= wrap2 * a.kw;
        ld_kh += wrap2;
      }
    };
    if constexpr (HAND_PLACED) {
      dma_tile_flat(0);
      __syncthreads();
      for (int kt = 0; kt + 1 < nk; ++kt) {
        k_step_placed(kt & 1, true);
        __syncthreads();
      }
      k_step_placed((nk - 1) & 1, false);
      if constexpr (!REG_ONLY) __syncthreads();
    } else if constexpr (FLAT_LOOP) {
      dma_tile_flat(0);
      __syncthreads();
      for (int kt = 0; kt + 1 < nk; ++kt) {
        const int buf = kt & 1;
        dma_tile_flat(buf ^ 1);
        compute(buf);
        __syncthreads();
      }
      compute((nk - 1) & 1);
      if constexpr (!REG_ONLY) __syncthreads();   // the LDS-staged epilogue overwrites the stages other waves may still read
    } else if constexpr (PROD) {
      if (is_prod) dma_tile(0);
      __syncthreads();
      for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (is_prod) {
          if (kt + 1 < nk) dma_tile(buf ^ 1);
        } else {
          compute(buf);
        }
        __syncthreads();
      }
    } else {
      dma_tile(0);
      __syncthreads();
      for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) dma_tile(buf ^ 1);
        compute(buf);
        __syncthreads();
      }
    }
  } else {
  load_w(0, wregA);
  load_x(0, xregA, xmaskA, CS0{});
  advance_k();
  finish_x(xregA, xmaskA, CS0{});
  store_tile(0, wregA, xregA);
  // Depth-2 register prefetch was measured NEUTRAL for the plain register-staged convs (tower 616 vs 618 TF/s, layer3/4
  // 3x3 unchanged) and NEGATIVE for the 8-wave deformable tile (PF2 above): compiled out.
  if constexpr (PF2) {
    // register-prefetch depth 2: sets A/B alternate, statically named (loop unrolled by two)
    if (nk > 1) {
      load_w(1, wregA);
      load_x(1, xregA, xmaskA, CS0{});
      advance_k();
    }
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      // even step: LDS[0] = tile kt, set A = tile kt+1 (in flight), issue tile kt+2 -> set B
      if (kt + 2 < nk) {
        load_w(kt + 2, wregB);
        load_x(kt + 2, xregB, xmaskB, CS1{});
        advance_k();
      }
      compute(0);
      if (kt + 1 < nk) {
        finish_x(xregA, xmaskA, CS0{});
        store_tile(1, wregA, xregA);
      }
      __syncthreads();
      if (kt + 1 >= nk) break;
      // odd step: LDS[1] = tile kt+1, set B = tile kt+2 (in flight), issue tile kt+3 -> set A
      if (kt + 3 < nk) {
        load_w(kt + 3, wregA);
        load_x(kt + 3, xregA, xmaskA, CS0{});
        advance_k();
      }
      compute(1);
      if (kt + 2 < nk) {
        finish_x(xregB, xmaskB, CS1{});
        store_tile(0, wregB, xregB);
      }
      __syncthreads();
    }
  } else {
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const bool more = (kt + 1) < nk;
      if (more) {
        load_w(kt + 1, wregA);
        load_x(kt + 1, xregA, xmaskA, CS0{});
        advance_k();
      }
      compute(buf);
      if (more) {
        finish_x(xregA, xmaskA, CS0{});
        store_tile(buf ^ 1, wregA, xregA);
      }
      __syncthreads();
    }
  }
  }  // !DMA

  // ---- epilogue: (acc + bias) * level_scale -> f32 tile in LDS -> coalesced 16-byte row
  // segments (+ residual) (relu) -> bf16 / f32.  The MFMA C layout gives a lane 4 couts of one
  // position (8-byte pieces scattered over 32 rows per store); staging through LDS turns the
  // stores (and the residual loads) into full 128-byte lines.
  const float lscale = a.level_scale[lev];
  const long long out_row0 = a.out_row0[lev] + grp * a.y_grows;
  const bool out_f32 = a.flags & SM_CONV_OUT_F32;
  float* E = reinterpret_cast<float*>(smem);
  unsigned long long* gn_bins = reinterpret_cast<unsigned long long*>(smem + SMEM_MAIN);   // [GN_SEG][BCO/8][2]
  if (gnp != nullptr && gtid < GN_SEG * (BCO / 8) * 2) gn_bins[gtid] = 0ull;
  // ---- register epilogue (same scheme as conv_dma32_kernel: v_permlane32_swap -> 8 consecutive couts per lane ->
  // 16-byte loads/stores, no LDS round trip).  GroupNorm statistics: after the swap a lane's 8 couts are exactly
  // one 8-channel group, so (sum, sum of squares) reduce over the 32 positions of the half-wave with shuffles and
  // land in the LDS bins with one atomic per half-wave (per-lane atomics when a tile straddles two images).
  const bool has_res0 = a.flags & (SM_CONV_RES_ADD | SM_CONV_RES_NEAREST);
  const bool reg_epi = !PROD && !(a.flags & SM_CONV_DBG_LDS_EPILOGUE) && (a.cout & 7) == 0 && (a.out_cstride & 7) == 0 &&
                       (a.out_coff & 7) == 0 && (!has_res0 || (a.res_cstride & 7) == 0);
  if (a.ksplit > 1) {
    // split-K slice: raw accumulators -> part[ks][out row][cout_pad] in the register-epilogue layout (8 consecutive
    // couts per lane, two 16-byte stores); the launcher only splits launches that meet reg_epi's conditions
    const int cpad = a.ntn * BCO;
    float* pp = a.part + (long long)ks * a.part_rows * cpad;
#pragma unroll
    for (int tp = 0; tp < TPOS; ++tp) {
      const int m = m0 + wpos * TPOS * 32 + tp * 32 + l31;
#pragma unroll
      for (int tc = 0; tc < TCO; ++tc) {
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
          if (m >= M) continue;
          const int c0 = nt * BCO + wco * TCO * 32 + tc * 32 + 8 * (2 * qp + khalf);
          float* q = pp + (long long)m * cpad + c0;
          *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(q + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
      }
    }
    return;
  }
  if (reg_epi) {
    const bool gn = gnp != nullptr;
    const int gn_groups = a.cout >> 3;
    const int gn_n0 = m0 / HoWo;
    if (gn) __syncthreads();                       // bins zeroed
#pragma unroll
    for (int tp = 0; tp < TPOS; ++tp) {
      const int m = m0 + wpos * TPOS * 32 + tp * 32 + l31;
      const bool mvalid = m < M;
      const int n_img = mvalid ? m / HoWo : -1;
      long long rrow = 0;
      if (has_res0 && mvalid) {
        if (a.flags & SM_CONV_RES_ADD) {
          rrow = out_row0 + m;
        } else {
          const int rem = m - n_img * HoWo;
          const int ho = rem / Wo;
          const int wo = rem - ho * Wo;
          const int rh = a.res_h[lev], rw = a.res_w[lev];
          const int sh = min((int)floorf((float)ho * ((float)rh / (float)Ho)), rh - 1);
          const int sw = min((int)floorf((float)wo * ((float)rw / (float)Wo)), rw - 1);
          rrow = a.res_row0[lev] + ((long long)n_img * rh + sh) * rw + sw;
        }
      }
      // is the whole wave inside one image?  (then the statistics reduce with shuffles)
      const int n_first = __builtin_amdgcn_readfirstlane(n_img);
      const bool uniform_img = gn && __all(n_img == n_first || n_img < 0) && n_first >= 0;
#pragma unroll
      for (int tc = 0; tc < TCO; ++tc) {
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
          if (a.acc_scale != 1.f) {                 // block-uniform; the x3 plan's power-of-two weight scale, undone exactly
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= a.acc_scale;
          }
          const int cl = wco * TCO * 32 + tc * 32 + 8 * (2 * qp + khalf);   // cout inside the tile
          const int c0 = nt * BCO + cl;
          const bool live = mvalid && c0 < a.cout;
          if (live) {
            if (biasp != nullptr) {
              const float4 b0 = *reinterpret_cast<const float4*>(biasp + c0);
              const float4 b1 = *reinterpret_cast<const float4*>(biasp + c0 + 4);
              v[0] += b0.x, v[1] += b0.y, v[2] += b0.z, v[3] += b0.w;
              v[4] += b1.x, v[5] += b1.y, v[6] += b1.z, v[7] += b1.w;
            }
            if (c0 < a.scale_nch) {
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (c0 + e < a.scale_nch) v[e] *= lscale;
            }
            if (has_res0) {
              float f[8];
              unpack_bf16x8(*reinterpret_cast<const u32x4*>(a.res + rrow * a.res_cstride + c0), f);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += f[e];
            }
          }
          if (gn) {                                 // wave-uniform branch: shuffles below need every lane
            float gs = 0.f, gss = 0.f;
            if (live) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                gs += v[e];
                gss = __builtin_fmaf(v[e], v[e], gss);
              }
            }
            if (uniform_img) {
              // one POSITION's 8 couts (summed in a fixed order) is the unit that is rounded to the fixed-point grid:
              // it does not depend on where tiles start, so plans that cut the same tensor differently (a B=4 plan
              // and two B=2 sub-plans) accumulate identical integers.  Everything after this line is integer addition.
              unsigned long long qs = gn_fix(gs), qss = gn_fix(gss);
#pragma unroll
              for (int d = 16; d > 0; d >>= 1) {   // within the half-wave: xor < 32 never crosses halves
                qs += __shfl_xor(qs, d, 64);
                qss += __shfl_xor(qss, d, 64);
              }
              if (l31 == 0 && c0 < a.cout) {
                const int seg = n_first - gn_n0;
                if (seg < GN_SEG) {
                  atomicAdd(&gn_bins[(seg * (BCO / 8) + (cl >> 3)) * 2 + 0], qs);
                  atomicAdd(&gn_bins[(seg * (BCO / 8) + (cl >> 3)) * 2 + 1], qss);
                } else {
                  unsigned long long* st = gnp + (((long long)n_first * a.nlev + lev) * gn_groups + (c0 >> 3)) * 2;
                  atomicAdd(st, qs);
                  atomicAdd(st + 1, qss);
                }
              }
            } else if (live) {
              const int seg = n_img - gn_n0;
              if (seg < GN_SEG) {
                atomicAdd(&gn_bins[(seg * (BCO / 8) + (cl >> 3)) * 2 + 0], gn_fix(gs));
                atomicAdd(&gn_bins[(seg * (BCO / 8) + (cl >> 3)) * 2 + 1], gn_fix(gss));
              } else {
                unsigned long long* st = gnp + (((long long)n_img * a.nlev + lev) * gn_groups + (c0 >> 3)) * 2;
                atomicAdd(st, gn_fix(gs));
                atomicAdd(st + 1, gn_fix(gss));
              }
            }
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
          const long long o = (out_row0 + m) * a.out_cstride + a.out_coff + c0;
#ifdef SM_OPERAND_F16
          if (a.flags & SM_CONV_OUT_X3) {          // the next layer's split operand: [hi | lo | hi], ctot = out_cstride / 3
            const int ctot = a.out_cstride / 3;
            frag8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float cv = fminf(fmaxf(v[e], -65504.f), 65504.f);
              const _Float16 hv = (_Float16)cv;
              hi[e] = hv;
              lo[e] = (_Float16)(cv - (float)hv);
            }
            uint16_t* yp = reinterpret_cast<uint16_t*>(a.y) + o;
            *reinterpret_cast<frag8*>(yp) = hi;
            *reinterpret_cast<frag8*>(yp + ctot) = lo;
            *reinterpret_cast<frag8*>(yp + 2 * ctot) = hi;
          } else
#endif
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
    if (gn) {                                       // flush the bins (same as the LDS-staged path)
      __syncthreads();
      if (gtid < GN_SEG * (BCO / 8) * 2) {
        const unsigned long long v = gn_bins[gtid];
        const int seg = gtid / ((BCO / 8) * 2), rem = gtid - seg * ((BCO / 8) * 2);
        const int g = (nt * BCO >> 3) + (rem >> 1);
        if (v != 0ull && g < gn_groups && gn_n0 + seg < a.batch)
          atomicAdd(gnp + (((long long)(gn_n0 + seg) * a.nlev + lev) * gn_groups + g) * 2 + (rem & 1), v);
      }
    }
    return;
  }
  if constexpr (REG_ONLY) return;
  if (is_cons) {
#pragma unroll
  for (int tc = 0; tc < TCO; ++tc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cl = wco * TCO * 32 + tc * 32 + 8 * q + 4 * khalf;  // cout inside the tile
      const int c = nt * BCO + cl;
      float bv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = (biasp != nullptr && c + e < a.cout) ? biasp[c + e] : 0.f;
#pragma unroll
      for (int tp = 0; tp < TPOS; ++tp) {
        const int pl = wpos * TPOS * 32 + tp * 32 + l31;
        float4 v;
        v.x = acc[tc][tp][4 * q + 0] * a.acc_scale + bv[0];
        v.y = acc[tc][tp][4 * q + 1] * a.acc_scale + bv[1];
        v.z = acc[tc][tp][4 * q + 2] * a.acc_scale + bv[2];
        v.w = acc[tc][tp][4 * q + 3] * a.acc_scale + bv[3];
        if (c + 0 < a.scale_nch) v.x *= lscale;
        if (c + 1 < a.scale_nch) v.y *= lscale;
        if (c + 2 < a.scale_nch) v.z *= lscale;
        if (c + 3 < a.scale_nch) v.w *= lscale;
        *reinterpret_cast<float4*>(E + pl * EPI_LD + cl) = v;
      }
    }
  }
  }  // is_cons
  __syncthreads();
  constexpr int CPR = BCO / 8;          // 8-cout chunks per tile row
  constexpr int RPP = THREADS / CPR;    // rows per pass (all waves of the block store)
  const int ec = gtid % CPR, er = gtid / CPR;
  const int c0 = nt * BCO + ec * 8;
  const bool has_res = a.flags & (SM_CONV_RES_ADD | SM_CONV_RES_NEAREST);
  const bool vec_ok = (c0 + 7 < a.cout) && ((a.out_cstride & 7) == 0) && ((a.out_coff & 7) == 0) &&
                      (!has_res || (a.res_cstride & 7) == 0);
  // fused GroupNorm statistics (8 channels per group == this thread's 8 couts): accumulate over the
  // thread's rows while the image index is unchanged, flush into LDS bins, one global atomic per bin
  const bool gn = gnp != nullptr;
  const int gn_groups = a.cout >> 3;
  const int gn_n0 = m0 / HoWo;               // first image touched by this tile
  int gn_n = gn_n0, gn_bound = (gn_n0 + 1) * HoWo;
  unsigned long long gn_s = 0ull, gn_ss = 0ull;    // fixed point; one position's 8 couts per rounding (tiling independent)
  auto gn_flush = [&]() {
    if (gn_s != 0ull || gn_ss != 0ull) {
      const int seg = gn_n - gn_n0;
      if (seg < GN_SEG) {
        atomicAdd(&gn_bins[(seg * (BCO / 8) + ec) * 2 + 0], gn_s);
        atomicAdd(&gn_bins[(seg * (BCO / 8) + ec) * 2 + 1], gn_ss);
      } else {
        unsigned long long* st = gnp + (((long long)gn_n * a.nlev + lev) * gn_groups + (c0 >> 3)) * 2;
        atomicAdd(st, gn_s);
        atomicAdd(st + 1, gn_ss);
      }
    }
    gn_s = 0ull;
    gn_ss = 0ull;
  };
  if (c0 < a.cout) {
#pragma unroll 2
    for (int r = er; r < BPOS; r += RPP) {
      const int m = m0 + r;
      if (m >= M) break;
      const float4 lo = *reinterpret_cast<const float4*>(E + r * EPI_LD + ec * 8);
      const float4 hi = *reinterpret_cast<const float4*>(E + r * EPI_LD + ec * 8 + 4);
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      if (has_res) {
        long long rrow;
        if (a.flags & SM_CONV_RES_ADD) {
          rrow = out_row0 + m;
        } else {
          const int n = m / HoWo;
          const int rem = m - n * HoWo;
          const int ho = rem / Wo;
          const int wo = rem - ho * Wo;
          const int rh = a.res_h[lev], rw = a.res_w[lev];
          const int sh = min((int)floorf((float)ho * ((float)rh / (float)Ho)), rh - 1);
          const int sw = min((int)floorf((float)wo * ((float)rw / (float)Wo)), rw - 1);
          rrow = a.res_row0[lev] + ((long long)n * rh + sh) * rw + sw;
        }
        const uint16_t* rp = a.res + rrow * a.res_cstride + c0;
        if (vec_ok) {
          float f[8];
          unpack_bf16x8(*reinterpret_cast<const u32x4*>(rp), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += f[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (c0 + e < a.cout) v[e] += bf16_bits_to_f32(rp[e]);
        }
      }
      if (gn) {
        while (m >= gn_bound) {   // crossed into the next image
          gn_flush();
          ++gn_n;
          gn_bound += HoWo;
        }
        float ps = 0.f, pss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          ps += v[e];
          pss = __builtin_fmaf(v[e], v[e], pss);
        }
        gn_s += gn_fix(ps);
        gn_ss += gn_fix(pss);
      }
      if (a.flags & SM_CONV_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (a.flags & SM_CONV_RELU_NCH) {   // ReLU on the Scale()d channels only (B/ bbox_pred, sipmask.py:157)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (c0 + e < a.scale_nch) v[e] = fmaxf(v[e], 0.f);
      }
      const long long o = (out_row0 + m) * a.out_cstride + a.out_coff + c0;
      if (out_f32) {
        float* yp = reinterpret_cast<float*>(a.y) + o;
        if (vec_ok) {
          *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (c0 + e < a.cout) yp[e] = v[e];
        }
      } else {
        uint16_t* yp = reinterpret_cast<uint16_t*>(a.y) + o;
        if (vec_ok) {
          *reinterpret_cast<u32x4*>(yp) = pack_bf16x8_v(v);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (c0 + e < a.cout) yp[e] = (uint16_t)f32_to_bf16_bits(v[e]);
        }
      }
    }
  }
  if (gn) {
    gn_flush();
    __syncthreads();
    if (gtid < GN_SEG * (BCO / 8) * 2) {
      const unsigned long long v = gn_bins[gtid];
      const int seg = gtid / ((BCO / 8) * 2), rem = gtid - seg * ((BCO / 8) * 2);
      const int g = (nt * BCO >> 3) + (rem >> 1);
      if (v != 0ull && g < gn_groups && gn_n0 + seg < a.batch)
        atomicAdd(gnp + (((long long)(gn_n0 + seg) * a.nlev + lev) * gn_groups + g) * 2 + (rem & 1), v);
    }
  }
}


// NOTE (round 1 experiment, removed): a 256-cout x 128-position, 8-wave, 3-stage LDS-DMA variant
// with counted vmcnt(6) + raw s_barrier measured 736 TF/s on the tower conv vs 716-757 for two
// 128x128 blocks per CU, and lost on small-M layers (half as many blocks): the K loop is not
// bound by prefetch depth or L2->LDS bytes.  See DESIGN.md section 6.


// =====================================================================================
// LDS-DMA kernel with 32-wide K steps: the same 4-wave tiles, but an LDS footprint small enough
// (2 x (BCO+BPOS) x 64 B stages, epilogue staged in two half-tile passes) for FOUR blocks per CU.
// Rationale (ablation): with K=64 steps the MFMA-only loop runs at ~1140
// TF/s and the DMA-only loop at ~1000 TF/s-equivalent, yet together they reach only ~760: phases
// of the two co-resident blocks barely overlap.  More, smaller blocks per CU give more phase
// diversity (the CU always has some wave in its MFMA phase) at the price of twice the barriers.
// LDS rows are 64 bytes; swizzle slot = chunk ^ ((row>>2)&3) keeps ds_read_b128 conflict free.
// =====================================================================================
// OPT = 3 (A/B flag SM_CONV_DBG_K32_OPT, cin >= 32): the K-loop treatment of conv_igemm_kernel's OPT 3 -- flat loader,
// peeled loop, fragments of the second K sub-step read under the MFMAs of the first.
// RESPF (A/B flag SM_CONV_DBG_RES_PREFETCH): the same-row residual of the register epilogue is loaded BEFORE the K
// loop, so its HBM latency overlaps the operand DMA and the MFMAs instead of following them (the 1x1 + residual convs
// of layer1/2 run 2 K steps per tile: load -> MFMA -> residual load -> store was four serial latencies per block).
// NST > 2 (round 5, experiments build): a ring of NST stages with NST - 1 of them in flight -- counted vmcnt wait + raw
// s_barrier per K step instead of __syncthreads (which drains the DMA queue).  Meant for the launches whose blocks are all
// resident at once (the 1x1 convs of layer3: 16 800 positions, 8 / 32 K steps); bit-identical, -14..-18 % on those two launches
// alone, -0.5 % (3 stages) / -3 % (4) on the pipelined step: profiles/r05_conv_ring_ab.txt, DESIGN section 6.
// X3P (binary16 build, round 6): paired split operands -- a K step is one [hi 16 | lo 16] group of 16 channels and issues the
// three cross products on the fragments it reads once (sm_conv_desc.x3_pairs).  A template parameter, not a runtime branch:
// two loop bodies that both update the accumulators made hipcc spill 238 registers in the 128 x 128 tile.
template <int WCO, int WPOS, int TCO, int TPOS, int MINB = 4, int OPT = 0, bool RESPF = false, int NST = 2, bool X3P = false>
__global__ __launch_bounds__(256, MINB) void conv_dma32_kernel(const ConvKArgs a) {
  constexpr int BCO = WCO * TCO * 32;
  constexpr int BPOS = WPOS * TPOS * 32;
  constexpr int NW = (BCO + 63) / 64;   // DMA instructions per thread per K step (weights)
  constexpr int NX = BPOS / 64;  // (activations)
  constexpr int STAGE = (BCO + BPOS) * 64;
  constexpr int EPI_LD = BCO + 4;
  constexpr int HROWS = BPOS / 2;                  // positions per epilogue pass
  constexpr int EPI_BYTES = HROWS * EPI_LD * 4;
  constexpr bool REG_ONLY = BCO * BPOS > 128 * 128;   // register epilogue only (see conv_igemm_kernel)
  constexpr int SMEM_BYTES = (REG_ONLY || NST * STAGE > EPI_BYTES) ? NST * STAGE : EPI_BYTES;
  static_assert(WCO * WPOS == 4, "4 waves");
  static_assert(NST == 2 || (OPT != 0 && BCO >= 64 && SMEM_BYTES <= 65536), "ring: flat loader, every wave owns weight rows");
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wco = wave / WPOS;
  const int wpos = wave % WPOS;
  // a wave-instruction covers 16 rows x 4 chunks; lane L -> row L>>2 (of the group), physical
  // slot L&3, so it fetches logical chunk (L&3) ^ ((row>>2)&3) = (L&3) ^ ((L>>4)&3)
  const int j = (lane & 3) ^ ((lane >> 4) & 3);
  const int r0 = tid >> 2;   // tile row (+64*i)

  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, xq = nblk >> 3, xr = nblk & 7;
  const int tlin = (a.flags & SM_CONV_DBG_LINEAR_TILES)
                       ? (int)blockIdx.x
                       : (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (int)(blockIdx.x >> 3);
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

  int rhi[NX], rwi[NX];
  long long xoff[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int m = m0 + r0 + 64 * i;
    if (m < M) {
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int ho = rem / Wo;
      const int wo = rem - ho * Wo;
      rhi[i] = ho * a.stride - a.pad;
      rwi[i] = wo * a.stride - a.pad;
      xoff[i] = (in_row0 + (long long)n * H * W + (long long)rhi[i] * W + rwi[i]) * a.in_cstride;
    } else {
      rhi[i] = -0x40000000;
      rwi[i] = 0;
      xoff[i] = 0;
    }
  }
  const uint16_t* ld_wp = a.w + (long long)(nt * BCO + r0) * a.Kp + j * 8 + (a.w_bstride != 0 ? (long long)(m0 / HoWo) * a.w_bstride : 0ll);
  const long long wstride = 64ll * a.Kp;
  int ld_kc = j, ld_cc, ld_kh, ld_kw;
  {
    const int tap0 = j / a.cpt;
    ld_cc = j - tap0 * a.cpt;
    ld_kh = tap0 / a.kw;
    ld_kw = tap0 - ld_kh * a.kw;
  }
  const unsigned long long zero_page = (unsigned long long)g_zero16;
  const int wave_row = wave * 16;   // first tile row written by this wave (+64*i)
  auto dma_tile = [&](int buf) {
    unsigned char* Wb = smem + buf * STAGE;
    unsigned char* Xb = Wb + BCO * 64;
    const bool kvalid = ld_kc < a.nchunk;
    const int dh = ld_kh * a.dil, dw = ld_kw * a.dil;
    const long long toff = (long long)((dh * W + dw) * a.in_cstride + ld_cc * 8);
#pragma unroll
    for (int i = 0; i < NW; ++i)
      if (BCO >= 64 || wave_row < BCO)   // 32-cout tile: only waves 0,1 own weight rows (wave-uniform)
        __builtin_amdgcn_global_load_lds((glb_void*)(ld_wp + i * wstride), (lds_void*)(Wb + (wave_row + 64 * i) * 64), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int hi = rhi[i] + dh, wi = rwi[i] + dw;
      const bool ok = kvalid && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      const unsigned long long pm = ok ? ~0ull : 0ull;
      const unsigned long long src = ((unsigned long long)(a.x + xoff[i] + toff) & pm) | (zero_page & ~pm);
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(Xb + (wave_row + 64 * i) * 64), 16, 0, 0);
    }
    ld_kc += 4;
    ld_wp += 32;
    if (a.cpt >= 4) {
      ld_cc += 4;
      if (ld_cc >= a.cpt) {
        ld_cc -= a.cpt;
        if (++ld_kw == a.kw) {
          ld_kw = 0;
          ++ld_kh;
        }
      }
    } else {
      const int tap = ld_kc / a.cpt;
      ld_cc = ld_kc - tap * a.cpt;
      ld_kh = tap / a.kw;
      ld_kw = tap - ld_kh * a.kw;
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
  const int rsw = (l31 >> 2) & 3;
  const int khalf = lane >> 5;
  const int wrow_off = (wco * TCO * 32 + l31) * 64;
  const int xrow_off = BCO * 64 + (wpos * TPOS * 32 + l31) * 64;
  auto compute = [&](int buf) {
    const unsigned char* S = smem + buf * STAGE;
    if constexpr (X3P) {     // hi halves in slots 0-1 of the 64-byte row, lo halves in slots 2-3
      const int s0 = (khalf ^ rsw) * 16, s1 = ((2 + khalf) ^ rsw) * 16;
      frag8 wh[TCO], xh[TPOS];
#pragma unroll
      for (int t = 0; t < TCO; ++t) wh[t] = *reinterpret_cast<const frag8*>(S + wrow_off + t * 32 * 64 + s0);
#pragma unroll
      for (int t = 0; t < TPOS; ++t) xh[t] = *reinterpret_cast<const frag8*>(S + xrow_off + t * 32 * 64 + s0);
      {
        frag8 xl[TPOS];
#pragma unroll
        for (int t = 0; t < TPOS; ++t) xl[t] = *reinterpret_cast<const frag8*>(S + xrow_off + t * 32 * 64 + s1);
#pragma unroll
        for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
          for (int tp = 0; tp < TPOS; ++tp) acc[tc][tp] = SM_MFMA_32x32x16(wh[tc], xh[tp], acc[tc][tp]);
#pragma unroll
        for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
          for (int tp = 0; tp < TPOS; ++tp) acc[tc][tp] = SM_MFMA_32x32x16(wh[tc], xl[tp], acc[tc][tp]);
      }
      {
        frag8 wl[TCO];
#pragma unroll
        for (int t = 0; t < TCO; ++t) wl[t] = *reinterpret_cast<const frag8*>(S + wrow_off + t * 32 * 64 + s1);
#pragma unroll
        for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
          for (int tp = 0; tp < TPOS; ++tp) acc[tc][tp] = SM_MFMA_32x32x16(wl[tc], xh[tp], acc[tc][tp]);
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int slot = ((kk * 2 + khalf) ^ rsw) * 16;
      frag8 wf[TCO], xf[TPOS];
#pragma unroll
      for (int t = 0; t < TCO; ++t) wf[t] = *reinterpret_cast<const frag8*>(S + wrow_off + t * 32 * 64 + slot);
#pragma unroll
      for (int t = 0; t < TPOS; ++t) xf[t] = *reinterpret_cast<const frag8*>(S + xrow_off + t * 32 * 64 + slot);
#pragma unroll
      for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
        for (int tp = 0; tp < TPOS; ++tp)
          acc[tc][tp] = SM_MFMA_32x32x16(wf[tc], xf[tp], acc[tc][tp]);
    }
  };

  // ---- residual prefetch (RESPF): the register epilogue's 16-byte residual pieces, addressed exactly as below
  u32x4 resv[RESPF ? TPOS : 1][RESPF ? TCO : 1][2];
  if constexpr (RESPF) {
#pragma unroll
    for (int tp = 0; tp < TPOS; ++tp) {
      const int m = m0 + wpos * TPOS * 32 + tp * 32 + (lane & 31);
      const long long rrow = a.out_row0[lev] + (m < M ? m : 0);
#pragma unroll
      for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          const int c0 = nt * BCO + wco * TCO * 32 + tc * 32 + 8 * (2 * qp + (lane >> 5));
          const int cc = c0 < a.cout ? c0 : 0;
          resv[tp][tc][qp] = *reinterpret_cast<const u32x4*>(a.res + rrow * a.res_cstride + cc);
        }
    }
  }
  const int nk = a.nk * 2;   // Kp is a multiple of 64
  if constexpr (OPT != 0) {
    const int wave_row_s = __builtin_amdgcn_readfirstlane(wave) * 16;
    auto dma_tile_flat = [&](int buf) {
      unsigned char* Wb = smem + buf * STAGE + wave_row_s * 64;
      unsigned char* Xb = Wb + BCO * 64;
      const bool kvalid = ld_kc < a.nchunk;
      const int dh = ld_kh * a.dil, dw = ld_kw * a.dil;
      const long long toff = (long long)((dh * W + dw) * a.in_cstride + ld_cc * 8);
#pragma unroll
      for (int i = 0; i < NW; ++i)
        if (BCO >= 64 || wave_row_s < BCO)
          __builtin_amdgcn_global_load_lds((glb_void*)(ld_wp + i * wstride), (lds_void*)(Wb + 64 * i * 64), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int hi = rhi[i] + dh, wi = rwi[i] + dw;
        const bool ok = kvalid & ((unsigned)hi < (unsigned)H) & ((unsigned)wi < (unsigned)W);
        const unsigned long long pm = ok ? ~0ull : 0ull;
        const unsigned long long src = ((unsigned long long)(a.x + xoff[i] + toff) & pm) | (zero_page & ~pm);
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(Xb + 64 * i * 64), 16, 0, 0);
      }
      ld_kc += 4;
      ld_wp += 32;
      ld_cc += 4;
      const int wrap = ld_cc >= a.cpt ? 1 : 0;
      ld_cc -= wrap * a.cpt;
      ld_kw += wrap;
      const int wrap2 = ld_kw == a.kw ? 1 : 0;
      ld_kw -= wrap2 * a.kw;
      ld_kh += wrap2;
    };
    auto compute_p = [&](int buf) {
      const unsigned char* S = smem + buf * STAGE;
      frag8 wf[2][TCO], xf[2][TPOS];
      auto rd = [&](int kk, int set) {
        const int slot = ((kk * 2 + khalf) ^ rsw) * 16;
#pragma unroll
        for (int t = 0; t < TCO; ++t) wf[set][t] = *reinterpret_cast<const frag8*>(S + wrow_off + t * 32 * 64 + slot);
#pragma unroll
        for (int t = 0; t < TPOS; ++t) xf[set][t] = *reinterpret_cast<const frag8*>(S + xrow_off + t * 32 * 64 + slot);
      };
      constexpr int NFR = TCO + TPOS, NMF = TCO * TPOS, NPAIR = NFR < NMF ? NFR : NMF;
      rd(0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NFR, 0);
      rd(1, 1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
          for (int tp = 0; tp < TPOS; ++tp)
            acc[tc][tp] = SM_MFMA_32x32x16(wf[kk][tc], xf[kk][tp], acc[tc][tp]);
        if (kk == 0) {
#pragma unroll
          for (int i = 0; i < NPAIR; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          if constexpr (NFR > NPAIR) __builtin_amdgcn_sched_group_barrier(0x100, NFR - NPAIR, 0);
          if constexpr (NMF > NPAIR) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NPAIR, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
        }
      }
    };
    if constexpr (NST > 2) {
      // stage kt + NST - 1 is issued at the top of step kt, behind the barrier that says every wave is done with stage
      // kt - 1 (the buffer it lands in).  vmcnt retires in order and every thread issues PER pieces per stage, so
      // "all but my last (NST - 2) * PER" = stage kt has landed.  The plan guarantees nk >= NST - 1.
      constexpr int PER = NW + NX;
#pragma unroll
      for (int s = 0; s < NST - 1; ++s) dma_tile_flat(s);
      int buf = 0, nbuf = NST - 1;
      for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_sched_barrier(0);
        if (kt + NST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (kt + NST - 1 < nk) dma_tile_flat(nbuf);
        compute_p(buf);
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
      }
      if constexpr (!REG_ONLY) __syncthreads();
    } else {
    dma_tile_flat(0);
    __syncthreads();
    for (int kt = 0; kt + 1 < nk; ++kt) {
      const int buf = kt & 1;
      dma_tile_flat(buf ^ 1);
      compute_p(buf);
      __syncthreads();
    }
    compute_p((nk - 1) & 1);
    if constexpr (!REG_ONLY) __syncthreads();   // the LDS-staged epilogue re-uses the stages
    }
  } else {
  dma_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) dma_tile(buf ^ 1);
    compute(buf);
    __syncthreads();
  }
  }

  const float lscale = a.level_scale[lev];
  const long long out_row0 = a.out_row0[lev];
  const bool out_f32 = a.flags & SM_CONV_OUT_F32;
  // ---- register epilogue (guide T21): the MFMA C layout leaves lanes i / i+32 with adjacent 4-cout groups of one
  // position; one v_permlane32_swap per dword turns each pair of groups into 8 consecutive couts per lane, i.e.
  // ONE 16-byte bf16 store (and one 16-byte residual load) per lane -- no LDS round trip, no barrier.  These
  // launches (1x1 convs, K <= 1152) run 2-36 K steps per tile, so the LDS-staged epilogue was most of their time.
  const bool has_res0 = a.flags & (SM_CONV_RES_ADD | SM_CONV_RES_NEAREST);
  const bool reg_epi = !(a.flags & SM_CONV_DBG_LDS_EPILOGUE) && (a.cout & 7) == 0 && (a.out_cstride & 7) == 0 &&
                       (a.out_coff & 7) == 0 && (!has_res0 || (a.res_cstride & 7) == 0);
  if (reg_epi) {
#pragma unroll
    for (int tp = 0; tp < TPOS; ++tp) {
      const int m = m0 + wpos * TPOS * 32 + tp * 32 + l31;
      long long rrow = 0;
      if (has_res0 && m < M) {
        if (a.flags & SM_CONV_RES_ADD) {
          rrow = out_row0 + m;
        } else {
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
        for (int qp = 0; qp < 2; ++qp) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t lo = __float_as_uint(acc[tc][tp][4 * (2 * qp) + e]);
            const uint32_t hi = __float_as_uint(acc[tc][tp][4 * (2 * qp + 1) + e]);
            const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
            v[e] = __uint_as_float(r[0]);       // lanes 0-31: own group 2qp      | lanes 32-63: lower half's group 2qp+1
            v[4 + e] = __uint_as_float(r[1]);   // lanes 0-31: upper's group 2qp  | lanes 32-63: own group 2qp+1
          }
          if (a.acc_scale != 1.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= a.acc_scale;
          }
          const int c0 = nt * BCO + wco * TCO * 32 + tc * 32 + 8 * (2 * qp + khalf);
          if (m >= M || c0 >= a.cout) continue;
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
          if (has_res0) {
            float f[8];
            if constexpr (RESPF) unpack_bf16x8(resv[tp][tc][qp], f);
            else unpack_bf16x8(*reinterpret_cast<const u32x4*>(a.res + rrow * a.res_cstride + c0), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += f[e];
          }
          if (a.flags & SM_CONV_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if (a.flags & SM_CONV_RELU_NCH) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (c0 + e < a.scale_nch) v[e] = fmaxf(v[e], 0.f);
          }
          const long long o = (out_row0 + m) * a.out_cstride + a.out_coff + c0;
#ifdef SM_OPERAND_F16
          if (a.flags & SM_CONV_OUT_X3) {          // the next layer's split operand: [hi | lo | hi], ctot = out_cstride / 3
            const int ctot = a.out_cstride / 3;
            frag8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float cv = fminf(fmaxf(v[e], -65504.f), 65504.f);
              const _Float16 hv = (_Float16)cv;
              hi[e] = hv;
              lo[e] = (_Float16)(cv - (float)hv);
            }
            uint16_t* yp = reinterpret_cast<uint16_t*>(a.y) + o;
            *reinterpret_cast<frag8*>(yp) = hi;
            *reinterpret_cast<frag8*>(yp + ctot) = lo;
            *reinterpret_cast<frag8*>(yp + 2 * ctot) = hi;
          } else
#endif
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
    return;
  }
  if constexpr (REG_ONLY) return;
  // ---- LDS-staged epilogue in two half-tile passes (positions [p*HROWS, (p+1)*HROWS)): unaligned channel
  // counts / strides, or the A/B debug flag
  float* E = reinterpret_cast<float*>(smem);
  constexpr int CPR = BCO / 8;
  constexpr int RPP = 256 / CPR;
  const int ec = tid % CPR, er = tid / CPR;
  const int c0 = nt * BCO + ec * 8;
  const bool has_res = a.flags & (SM_CONV_RES_ADD | SM_CONV_RES_NEAREST);
  const bool vec_ok = (c0 + 7 < a.cout) && ((a.out_cstride & 7) == 0) && ((a.out_coff & 7) == 0) &&
                      (!has_res || (a.res_cstride & 7) == 0);
  const int wave_p0 = wpos * TPOS * 32;            // first position (in the tile) of this wave
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (p) __syncthreads();                          // pass 0 readers are done with E
    if (wave_p0 / HROWS == p) {
#pragma unroll
      for (int tc = 0; tc < TCO; ++tc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cl = wco * TCO * 32 + tc * 32 + 8 * q + 4 * khalf;
          const int c = nt * BCO + cl;
          float bv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[e] = (a.bias != nullptr && c + e < a.cout) ? a.bias[c + e] : 0.f;
#pragma unroll
          for (int tp = 0; tp < TPOS; ++tp) {
            const int pl = wave_p0 % HROWS + tp * 32 + l31;
            float4 v;
            v.x = acc[tc][tp][4 * q + 0] * a.acc_scale + bv[0];
            v.y = acc[tc][tp][4 * q + 1] * a.acc_scale + bv[1];
            v.z = acc[tc][tp][4 * q + 2] * a.acc_scale + bv[2];
            v.w = acc[tc][tp][4 * q + 3] * a.acc_scale + bv[3];
            if (c + 0 < a.scale_nch) v.x *= lscale;
            if (c + 1 < a.scale_nch) v.y *= lscale;
            if (c + 2 < a.scale_nch) v.z *= lscale;
            if (c + 3 < a.scale_nch) v.w *= lscale;
            *reinterpret_cast<float4*>(E + pl * EPI_LD + cl) = v;
          }
        }
      }
    }
    __syncthreads();
    if (c0 < a.cout) {
#pragma unroll 2
      for (int r = er; r < HROWS; r += RPP) {
        const int m = m0 + p * HROWS + r;
        if (m >= M) break;
        const float4 lo = *reinterpret_cast<const float4*>(E + r * EPI_LD + ec * 8);
        const float4 hi = *reinterpret_cast<const float4*>(E + r * EPI_LD + ec * 8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (has_res) {
          long long rrow;
          if (a.flags & SM_CONV_RES_ADD) {
            rrow = out_row0 + m;
          } else {
            const int n = m / HoWo;
            const int rem = m - n * HoWo;
            const int ho = rem / Wo;
            const int wo = rem - ho * Wo;
            const int rh = a.res_h[lev], rw = a.res_w[lev];
            const int sh = min((int)floorf((float)ho * ((float)rh / (float)Ho)), rh - 1);
            const int sw = min((int)floorf((float)wo * ((float)rw / (float)Wo)), rw - 1);
            rrow = a.res_row0[lev] + ((long long)n * rh + sh) * rw + sw;
          }
          const uint16_t* rp = a.res + rrow * a.res_cstride + c0;
          if (vec_ok) {
            float f[8];
            unpack_bf16x8(*reinterpret_cast<const u32x4*>(rp), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += f[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (c0 + e < a.cout) v[e] += bf16_bits_to_f32(rp[e]);
          }
        }
        if (a.flags & SM_CONV_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (a.flags & SM_CONV_RELU_NCH) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (c0 + e < a.scale_nch) v[e] = fmaxf(v[e], 0.f);
        }
        const long long o = (out_row0 + m) * a.out_cstride + a.out_coff + c0;
        if (out_f32) {
          float* yp = reinterpret_cast<float*>(a.y) + o;
          if (vec_ok) {
            *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (c0 + e < a.cout) yp[e] = v[e];
          }
        } else {
          uint16_t* yp = reinterpret_cast<uint16_t*>(a.y) + o;
          if (vec_ok) {
            *reinterpret_cast<u32x4*>(yp) = pack_bf16x8_v(v);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (c0 + e < a.cout) yp[e] = (uint16_t)f32_to_bf16_bits(v[e]);
          }
        }
      }
    }
  }
}

// ---- split-K epilogue: sum the S partial slabs, then bias / same-row residual / ReLU / store (single-level launches)
struct SplitKEpi {
  const float* part;
  const float* bias;
  const uint16_t* res;
  void* y;
  long long rows, part_rows, out_row0;
  int S, cout, cpad, out_cstride, out_coff, res_cstride;
  unsigned flags;
  float acc_scale;
};

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const SplitKEpi e) {
  const int c8 = e.cout >> 3;
  const long long total = e.rows * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / c8;
    const int c0 = (int)(i - m * c8) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < e.S; ++s) {
      const float* q = e.part + ((long long)s * e.part_rows + m) * e.cpad + c0;
      const float4 a0 = *reinterpret_cast<const float4*>(q), a1 = *reinterpret_cast<const float4*>(q + 4);
      v[0] += a0.x, v[1] += a0.y, v[2] += a0.z, v[3] += a0.w;
      v[4] += a1.x, v[5] += a1.y, v[6] += a1.z, v[7] += a1.w;
    }
    if (e.acc_scale != 1.f) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= e.acc_scale;
    }
    if (e.bias != nullptr) {
      const float4 b0 = *reinterpret_cast<const float4*>(e.bias + c0), b1 = *reinterpret_cast<const float4*>(e.bias + c0 + 4);
      v[0] += b0.x, v[1] += b0.y, v[2] += b0.z, v[3] += b0.w;
      v[4] += b1.x, v[5] += b1.y, v[6] += b1.z, v[7] += b1.w;
    }
    if (e.flags & SM_CONV_RES_ADD) {
      float f[8];
      unpack_bf16x8(*reinterpret_cast<const u32x4*>(e.res + (e.out_row0 + m) * e.res_cstride + c0), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += f[k];
    }
    if (e.flags & SM_CONV_RELU) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    const long long o = (e.out_row0 + m) * e.out_cstride + e.out_coff + c0;
    if (e.flags & SM_CONV_OUT_F32) {
      float* yp = reinterpret_cast<float*>(e.y) + o;
      *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(e.y) + o) = pack_bf16x8_v(v);
    }
  }
}

// ---- launch planning: pure host logic (no device access), exported as sm_conv_plan_query so that the selection
// rules are testable without a GPU.  launch_conv() below executes exactly this plan.
int plan_conv(const sm_conv_desc* d, bool deform, bool with_gn, sm_conv_plan* p, bool allow_split = true) {
  if (!d || !p) return SM_ERR_BAD_ARG;
  if (d->nlev < 1 || d->nlev > SM_MAX_LEVELS || d->batch < 1) return SM_ERR_BAD_SHAPE;
  if (d->cin % 8 != 0 || d->cin < 8 || d->cout < 1) return SM_ERR_BAD_SHAPE;
  if (d->in_cstride % 8 != 0) return SM_ERR_BAD_SHAPE;
  const int tile = sm_conv_cout_tile(d->cout);
  if (d->cout_pad % tile != 0 || d->cout_pad < d->cout) return SM_ERR_BAD_SHAPE;
  if (deform) {
    if (d->deform_groups < 1 || d->cin % (8 * d->deform_groups) != 0) return SM_ERR_BAD_ARG;
    // any stride: the offset rows are OUTPUT rows (deform_conv_cuda_kernel.cu:216-223), the sampling centre is
    // ho * stride - pad + kh * dil in the loaders
  }
  if (with_gn && d->cout % 8 != 0) return SM_ERR_UNSUPPORTED;
  for (int l = 0; l < d->nlev; ++l) {
    if (d->out_h[l] < 1 || d->out_w[l] < 1 || d->in_h[l] < 1 || d->in_w[l] < 1) return SM_ERR_BAD_SHAPE;
    const int eh = (d->in_h[l] + 2 * d->pad - (d->dil * (d->kh - 1) + 1)) / d->stride + 1;
    const int ew = (d->in_w[l] + 2 * d->pad - (d->dil * (d->kw - 1) + 1)) / d->stride + 1;
    if (eh != d->out_h[l] || ew != d->out_w[l]) return SM_ERR_BAD_SHAPE;
  }
  // LDS-DMA loader for plain convs; the register-staged loader where VALU must touch the operand
  // (deformable gather, input ReLU) or when the A/B debug flag asks for it
  const bool dma = !deform && !(d->flags & (SM_CONV_IN_RELU | SM_CONV_DBG_REG_STAGING));
  const int K = d->kh * d->kw * d->cin;
  const int Kp = (K + 63) / 64 * 64;
  const int ngroups = d->ngroups > 1 ? d->ngroups : 1;
  // K-step width: measured on MI355X (profiles/r01_conv_microbench.txt) the 32-wide / 4-blocks-per-CU
  // kernel wins for K <= 1152 (all 1x1 convs, the 3x3 convs of layer1/2, the stem: +5..+28 %) and
  // loses for K >= 2304 (towers, FPN, layer3/4 3x3: -8..-20 %)
  const bool k32 = dma && !with_gn && ngroups == 1 &&
                   ((d->flags & SM_CONV_DBG_K32) || (!(d->flags & SM_CONV_DBG_K64) && Kp <= 1152));
  // ---- tile selection.  The cout tile is fixed by the weight padding contract (32/64/128) but may
  // be split further (128 -> 64); the position tile shrinks until the launch has enough blocks to
  // occupy the chip (256 CUs x 2 or 4 resident blocks): small-M layers (layer3/4, P5-P7, the
  // 32-channel predictors) are latency bound and want blocks, not reuse.
  struct Cfg { int bco, bpos; };
  Cfg cands[4];
  int ncand = 0;
  const bool has_res = d->flags & (SM_CONV_RES_ADD | SM_CONV_RES_NEAREST);
  // alignment conditions of the register epilogue (the tiles above 128x128 have no other)
  const bool reg_ok = !(d->flags & SM_CONV_DBG_LDS_EPILOGUE) && (d->cout & 7) == 0 && (d->out_cstride & 7) == 0 &&
                      (d->out_coff & 7) == 0 && (!has_res || (d->res_cstride & 7) == 0);
  if (!dma) {
    // deformable conv: the bilinear gather (offset load, corner weights, 4 corner loads and a VALU blend per 8-channel
    // chunk) is the loader's cost and it is paid once per BLOCK tile row, whatever the cout extent: a 256-cout x
    // 128-position tile on 8 waves halves the gather work per MFMA of the 128 x 128 tile (the FeatureAlign conv was
    // VALU-bound in the loader: ~520 loader VALU per thread per K step against 512 MFMA cycles)
    if (deform && tile == 128 && d->cout_pad % 256 == 0 && reg_ok && !(d->flags & SM_CONV_DBG_DEFORM_128))
      cands[ncand++] = {256, 128};
    else
      cands[ncand++] = {tile, tile == 128 ? 128 : 256};
  } else if (tile == 128) {
    // 128 couts x 256 positions (64x128 per wave: 0.75 fragment reads and 0.75 DMA bytes per MFMA of the 128x128
    // tile), behind an A/B flag
    if ((d->flags & SM_CONV_DBG_WIDE_POS) && reg_ok) cands[ncand++] = {128, 256};
    if (!(k32 && (d->flags & SM_CONV_DBG_K32_POS64)))
    cands[ncand++] = {128, 128};
    cands[ncand++] = {128, 64};
    if (ncand < 4) cands[ncand++] = {64, 64};
  } else if (tile == 64) {
    cands[ncand++] = {64, 256};
    cands[ncand++] = {64, 128};
    cands[ncand++] = {64, 64};
  } else {
    cands[ncand++] = {32, 256};
    cands[ncand++] = {32, 128};
  }
  const long long want = (d->flags & SM_CONV_DBG_BIG_TILES) ? 0 : (k32 ? 768 : 512);
  // split-K (sm_conv2d_ws): the flat-loop 64-wide-K kernel on single-level launches with the register epilogue's
  // alignment and a plain epilogue (bias / same-row residual / ReLU)
  const int nk64 = Kp / 64;
  int split = 1;
  const bool sk_base = allow_split && dma && !k32 && !with_gn && ngroups == 1 && d->nlev == 1 && reg_ok &&
                       !(d->flags & (SM_CONV_RES_NEAREST | SM_CONV_RELU_NCH | SM_CONV_DBG_NO_SPLITK | SM_CONV_DBG_HAND_PLACED |
                                     SM_CONV_DBG_WARP_SPEC | SM_CONV_DBG_LEGACY_LOOP | SM_CONV_DBG_FLAT_LOOP | SM_CONV_DBG_TILE256)) &&
                       d->scale_nch == 0 && d->w_batch_stride == 0;
  int bco = cands[0].bco, bpos = cands[0].bpos;
  const bool ws = dma && !k32 && (d->flags & SM_CONV_DBG_WARP_SPEC) != 0;
  // 256x256 tile on 8 waves, one block per CU (A/B flag): half the LDS-DMA pieces and 3/4 of the fragment reads per
  // MFMA of the 128x128 tile.  Register epilogue only; 64-wide K steps only.
  bool tile256 = dma && !k32 && !ws && tile == 128 && (d->flags & SM_CONV_DBG_TILE256) && d->cout_pad % 256 == 0 &&
                 d->cin >= 64 && reg_ok;
  if (tile256 && !(d->flags & SM_CONV_DBG_BIG_TILES)) {
    // one block per CU: the launch must fill its rounds of 256 blocks reasonably (a 263-block launch runs two rounds)
    long long nb = 0;
    for (int l = 0; l < d->nlev; ++l) nb += sm_cdiv((long long)d->batch * d->out_h[l] * d->out_w[l], 256);
    nb *= d->cout_pad / 256;
    nb *= ngroups;
    const long long rounds = (nb + 255) / 256;
    tile256 = nb >= 230 && nb * 100 >= rounds * 256 * 65;
  }
  if (tile256) {
    bco = bpos = 256;
    ncand = 0;
  }
  for (int c = 0; c < ncand; ++c) {
    long long nb = 0;
    for (int l = 0; l < d->nlev; ++l)
      nb += sm_cdiv((long long)d->batch * d->out_h[l] * d->out_w[l], cands[c].bpos);
    nb *= d->cout_pad / cands[c].bco;
    bco = cands[c].bco;
    bpos = cands[c].bpos;
    if (nb >= want) break;
    // split-K instead of shrinking the tile further.  Measured (round 2, B=4 R50, HIP events incl. the reduce launch):
    // it only pays for launches of a few blocks with very long K loops -- layer4 3x3 (132 tiles of 128x128, 72 K
    // steps) 0.0595 -> 0.0530 ms, fpn.out2 0.0339 -> 0.0250 ms -- and LOSES where the unsplit launch already fills
    // the chip with smaller tiles (layer3 3x3: 526 tiles of 128x64 0.0415 ms vs 2 x 264 tiles of 128x128 + reduce
    // 0.0536 ms; 1x1 convs with K = 2048 likewise): the reduce kernel's launch boundary costs what the shorter K loop
    // saves.  Hence the narrow rule: <= 160 tiles and >= 36 K steps.
    if (sk_base && cands[c].bco == 128 && cands[c].bpos <= 128 && d->cin >= 64 && nb <= 160 && nk64 >= 36) {
      int S = nk64 / 8 < 4 ? nk64 / 8 : 4;
      if (S > 1) {
        split = S;
        break;
      }
    }
  }
  long long t = 0;
  for (int l = 0; l < d->nlev; ++l) t += sm_cdiv((long long)d->batch * d->out_h[l] * d->out_w[l], bpos);
  const long long nblk = t * (d->cout_pad / bco) * ngroups;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return SM_ERR_BAD_SHAPE;
  if (d->w_level_stride != 0 || d->bias_level_stride != 0) return SM_ERR_UNSUPPORTED;   // per-level weights: sm_conv3x3_patch only
  if (d->x3_pairs != 0 && !(d->flags & SM_CONV_F16)) return SM_ERR_UNSUPPORTED;         // paired split operands: binary16 only
  // groups exist in the 64-wide-K LDS-DMA kernel only (its tile decode carries the group offsets)
  if (ngroups > 1 && (!dma || k32 || ws || (d->flags & SM_CONV_RES_NEAREST) || d->w_batch_stride != 0)) return SM_ERR_UNSUPPORTED;
  // K-loop variant.  64-wide K, 128/64-cout tiles of the tile-128 family: flat loader + peeled K loop + pipelined
  // fragment reads (OPT 3) whenever cin >= 64 (no division path in the flat loader); measured +10..15 % on every 3x3
  // conv with K >= 2304 (profiles/r01_conv_kloop_variants.txt).  A/B flags: FLAT_LOOP = OPT 1, LEGACY_LOOP = OPT 0.
  // 32-wide K: OPT 3 behind SM_CONV_DBG_K32_OPT (cin >= 32).
  int opt = 0;
  if (dma && k32) {
    const bool has_o3 = !(bco == 128 && bpos == 256);
    opt = ((d->flags & SM_CONV_DBG_K32_OPT) && d->cin >= 32 && has_o3) ? 3 : 0;
  } else if (dma && !ws) {
    const bool has_opt = (bco == 128 && (bpos == 128 || bpos == 64)) || (bco == 64 && bpos == 64);
    if (bco == 256) opt = 3;
    else if (has_opt && d->cin >= 64 && !(d->flags & SM_CONV_DBG_LEGACY_LOOP)) opt = (d->flags & SM_CONV_DBG_FLAT_LOOP) ? 1 : 3;
  }
  if (ws && !((bco == 128 && (bpos == 128 || bpos == 64)) || (bco == 64 && bpos == 64))) return SM_ERR_UNSUPPORTED;
  // 32-wide K, experiments build only: a ring of 3 / 4 LDS stages (conv_dma32_kernel, NST; SIPMASK_EXP_K32_RING = 3 | 4)
  int ring = 0;
#ifdef SM_EXPERIMENTS
  if (dma && k32 && !(d->flags & SM_CONV_F16) && d->cin >= 32 && Kp / 32 >= 8 &&
      ((bco == 128 && (bpos == 128 || bpos == 64)) || (bco == 64 && bpos == 64))) {
    const int e = sm_experiment_env("SIPMASK_EXP_K32_RING", 0);
    if (e == 3 || e == 4) ring = e;
  }
#endif
  if (ring) opt = 3;
  p->ring_stages = ring;
  p->lds_dma = dma ? 1 : 0;
  p->k_step = k32 ? 32 : 64;
  p->k_padded = Kp;
  p->tile_cout = bco;
  p->tile_pos = bpos;
  p->threads = (bco == 256 || ws) ? 512 : 256;
  p->k_loop = opt;
  p->warp_spec = ws ? 1 : 0;
  p->blocks = nblk;
  // split-K: S slices per tile (chosen with the tile above), partial slabs [S][rows][cout_pad] f32 in the workspace
  p->split_k = (split > 1 && opt == 3) ? split : 1;
  p->workspace_bytes = 0;
  if (p->split_k > 1)
    p->workspace_bytes = (long long)p->split_k * d->batch * d->out_h[0] * d->out_w[0] * d->cout_pad * 4;
  return SM_OK;
}

template <bool DEFORM>
int launch_conv(const sm_conv_desc* d, const void* x, const float* offset, const void* w, const float* bias,
                const void* residual, void* y, hipStream_t stream, unsigned long long* gn_stats = nullptr, void* workspace = nullptr,
                long long workspace_bytes = 0) {
  if (!d || !x || !w || !y) return SM_ERR_BAD_ARG;
  if (DEFORM && !offset) return SM_ERR_BAD_ARG;
  sm_conv_plan plan;
  int prc = plan_conv(d, DEFORM, gn_stats != nullptr, &plan, !DEFORM && workspace != nullptr);
  if (prc == SM_OK && plan.split_k > 1 && workspace_bytes < plan.workspace_bytes)      // workspace too small: no split
    prc = plan_conv(d, DEFORM, gn_stats != nullptr, &plan, false);
  if (prc != SM_OK) return prc;
#ifdef SM_OPERAND_F16
  // the binary16 build: LDS-DMA kernels with f32 output only (what the x3 head plan launches)
  if (DEFORM || !plan.lds_dma || !(d->flags & (SM_CONV_OUT_F32 | SM_CONV_OUT_X3)) ||
      (d->flags & (SM_CONV_RES_ADD | SM_CONV_RES_NEAREST)))
    return SM_ERR_UNSUPPORTED;
  if (d->x3_pairs != 0) {
    // paired split operands on this kernel family: the 32-wide-K loop without the experiments build's pipelined variant (one
    // K step = one [hi 16 | lo 16] group; sip_mask_lat0's 1x1 convs), plain launches
    if (d->x3_pairs != 1 || plan.k_step != 32 || plan.k_loop != 0 || (d->cin & 31) || d->ngroups > 1 || plan.split_k > 1)
      return SM_ERR_UNSUPPORTED;
  }
  if (d->flags & SM_CONV_OUT_X3) {                  // written by the register epilogue only (8 couts per lane, 16-byte stores)
    if ((d->flags & SM_CONV_DBG_LDS_EPILOGUE) || (d->cout & 7) || (d->out_cstride % 24) || (d->out_coff & 7) || gn_stats ||
        d->out_cstride < 3 * d->cout || plan.split_k > 1)
      return SM_ERR_UNSUPPORTED;
  }
#else
  if (!DEFORM && (d->flags & SM_CONV_F16))          // same plan, same launch code, the other MFMA (conv_igemm_f16.o)
    return sm_conv_igemm_f16(d, x, w, bias, y, stream, gn_stats, (d->flags & SM_CONV_OUT_X3) ? nullptr : workspace,
                             workspace_bytes);
  if (d->flags & SM_CONV_F16) return SM_ERR_UNSUPPORTED;
#endif
  if ((d->flags & (SM_CONV_RES_ADD | SM_CONV_RES_NEAREST)) && !residual) return SM_ERR_BAD_ARG;
  const int tile = sm_conv_cout_tile(d->cout);
  (void)tile;      // (the binary16 build has no register-staged kernels: unused there)
  const bool dma = plan.lds_dma != 0, k32 = plan.k_step == 32;
#ifdef SM_EXPERIMENTS
  const bool ws = plan.warp_spec != 0;
#endif
  const int bco = plan.tile_cout, bpos = plan.tile_pos, opt = plan.k_loop;
  ConvKArgs a;
  a.x = (const uint16_t*)x;
  a.w = (const uint16_t*)w;
  a.w_bstride = d->w_batch_stride;
  a.bias = bias;
  a.res = (const uint16_t*)residual;
  a.y = y;
  a.offset = offset;
  a.gn_stats = gn_stats;
  if (gn_stats != nullptr) {
    const int ng = d->ngroups > 1 ? d->ngroups : 1;
    if (ng > 1 && d->gn_group_stride != 2ll * d->batch * d->nlev * (d->cout / 8)) return SM_ERR_BAD_ARG;   // contiguous
    if (sm_zero_async(gn_stats, sizeof(unsigned long long) * 2 * d->batch * d->nlev * (d->cout / 8) * ng, stream) != hipSuccess)
      return SM_ERR_LAUNCH;
  }
  if constexpr (DEFORM) {
    // FeatureAlign's shape goes to the LDS-patch kernel (deform_patch.hip); everything else, and the A/B flag, gathers
    if (!(d->flags & SM_CONV_DBG_DEFORM_GATHER) && sm_deform_patch_supported(d) && bco == 256)
      return sm_deform_patch_launch(d, x, offset, w, bias, y, stream, gn_stats, plan.k_padded);
  }
  a.nlev = d->nlev;
  a.batch = d->batch;
  const int K = d->kh * d->kw * d->cin;
  a.Kp = plan.k_padded;
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
    if (on) t += sm_cdiv((long long)d->batch * d->out_h[l] * d->out_w[l], bpos);
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
  a.nchunk = K / 8;
  a.cpt = d->cin / 8;
  a.ntn = d->cout_pad / bco;
  a.nk = a.Kp / 64;
  a.flags = d->flags;
  a.scale_nch = d->scale_nch;
  a.acc_scale = (d->acc_scale == 0.f) ? 1.f : d->acc_scale;
  a.dg = DEFORM ? d->deform_groups : 1;
  a.cpg8 = DEFORM ? d->cin / (8 * d->deform_groups) : 1;
  const int S = plan.split_k > 1 ? plan.split_k : 1;
  a.ksplit = S;
  a.part = (float*)workspace;
  a.part_rows = (long long)d->batch * d->out_h[0] * d->out_w[0];
  a.ngroups = d->ngroups > 1 ? d->ngroups : 1;
  a.tpg = t * a.ntn;
  a.x_grows = d->x_group_rows;
  a.y_grows = d->y_group_rows;
  a.w_gstride = d->w_group_stride;
  a.b_gstride = d->bias_group_stride;
  a.gn_gstride = d->gn_group_stride;
  const long long nblk = (long long)t * a.ntn * a.ngroups * S;
  if (nblk != plan.blocks * S) return SM_ERR_BAD_SHAPE;   // plan_conv and this function must agree
  dim3 grid((unsigned)nblk), block(256);
#define SM_LAUNCH(KERNEL) hipLaunchKernelGGL((KERNEL), grid, block, 0, stream, a)
  if (!dma) {
#ifndef SM_OPERAND_F16
    if (DEFORM && bco == 256) { block = dim3(512); SM_LAUNCH((conv_igemm_kernel<4, 2, 2, 2, DEFORM, false>)); }
    else if (tile == 128) SM_LAUNCH((conv_igemm_kernel<2, 2, 2, 2, DEFORM, false>));
    else if (tile == 64) SM_LAUNCH((conv_igemm_kernel<1, 4, 2, 2, DEFORM, false>));
    else SM_LAUNCH((conv_igemm_kernel<1, 4, 1, 2, DEFORM, false>));
#endif
  } else if (k32 && plan.ring_stages > 2) {
#if defined(SM_EXPERIMENTS) && !defined(SM_OPERAND_F16)
    // residual prefetch in front of the K loop where the register epilogue will run with a same-row residual
    const bool rp = (d->flags & SM_CONV_RES_ADD) && !(d->flags & SM_CONV_DBG_LDS_EPILOGUE) && (d->cout & 7) == 0 &&
                    (d->out_cstride & 7) == 0 && (d->out_coff & 7) == 0 && (d->res_cstride & 7) == 0 && d->cout % bco == 0;
    const int r4 = plan.ring_stages == 4;
    if (bco == 128 && bpos == 128) {
      if (r4 && rp) SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 2, 2, 3, true, 4>));
      else if (r4) SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 2, 2, 3, false, 4>));
      else if (rp) SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 2, 2, 3, true, 3>));
      else SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 2, 2, 3, false, 3>));
    } else if (bco == 128 && bpos == 64) {
      if (r4 && rp) SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 1, 3, 3, true, 4>));
      else if (r4) SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 1, 3, 3, false, 4>));
      else if (rp) SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 1, 3, 3, true, 3>));
      else SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 1, 3, 3, false, 3>));
    } else if (bco == 64 && bpos == 64) {
      if (r4 && rp) SM_LAUNCH((conv_dma32_kernel<2, 2, 1, 1, 4, 3, true, 4>));
      else if (r4) SM_LAUNCH((conv_dma32_kernel<2, 2, 1, 1, 4, 3, false, 4>));
      else if (rp) SM_LAUNCH((conv_dma32_kernel<2, 2, 1, 1, 4, 3, true, 3>));
      else SM_LAUNCH((conv_dma32_kernel<2, 2, 1, 1, 4, 3, false, 3>));
    } else return SM_ERR_UNSUPPORTED;
#else
    return SM_ERR_UNSUPPORTED;
#endif
  } else if (k32) {
#ifdef SM_EXPERIMENTS
    const bool o3 = opt == 3;
    if (o3 && bco == 128 && bpos == 128) SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 2, 4, 3>));
    else if (o3 && bco == 128 && bpos == 64) SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 1, 4, 3>));
    else if (o3 && bco == 64 && bpos == 256) SM_LAUNCH((conv_dma32_kernel<1, 4, 2, 2, 4, 3>));
    else if (o3 && bco == 64 && bpos == 128) SM_LAUNCH((conv_dma32_kernel<1, 4, 2, 1, 4, 3>));
    else if (o3 && bco == 64 && bpos == 64) SM_LAUNCH((conv_dma32_kernel<2, 2, 1, 1, 4, 3>));
    else if (o3 && bco == 32 && bpos == 256) SM_LAUNCH((conv_dma32_kernel<1, 4, 1, 2, 4, 3>));
    else if (o3 && bco == 32) SM_LAUNCH((conv_dma32_kernel<1, 4, 1, 1, 4, 3>));
    else if (bco == 128 && bpos == 256) SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 4, 2>));
    else
#endif
#ifdef SM_OPERAND_F16
    if (d->x3_pairs) {
      if (bco == 128 && bpos == 128) SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 2, 4, 0, false, 2, true>));
      else if (bco == 128 && bpos == 64) SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 1, 4, 0, false, 2, true>));
      else if (bco == 64 && bpos == 256) SM_LAUNCH((conv_dma32_kernel<1, 4, 2, 2, 4, 0, false, 2, true>));
      else if (bco == 64 && bpos == 128) SM_LAUNCH((conv_dma32_kernel<1, 4, 2, 1, 4, 0, false, 2, true>));
      else if (bco == 64 && bpos == 64) SM_LAUNCH((conv_dma32_kernel<2, 2, 1, 1, 4, 0, false, 2, true>));
      else if (bco == 32 && bpos == 256) SM_LAUNCH((conv_dma32_kernel<1, 4, 1, 2, 4, 0, false, 2, true>));
      else if (bco == 32) SM_LAUNCH((conv_dma32_kernel<1, 4, 1, 1, 4, 0, false, 2, true>));
      else return SM_ERR_UNSUPPORTED;
    } else
#endif
    if (bco == 128 && bpos == 128) SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 2>));
    else if (bco == 128 && bpos == 64) SM_LAUNCH((conv_dma32_kernel<2, 2, 2, 1>));
    else if (bco == 64 && bpos == 256) SM_LAUNCH((conv_dma32_kernel<1, 4, 2, 2>));
    else if (bco == 64 && bpos == 128) SM_LAUNCH((conv_dma32_kernel<1, 4, 2, 1>));
    else if (bco == 64 && bpos == 64) SM_LAUNCH((conv_dma32_kernel<2, 2, 1, 1>));
    else if (bco == 32 && bpos == 256) SM_LAUNCH((conv_dma32_kernel<1, 4, 1, 2>));
    else if (bco == 32) SM_LAUNCH((conv_dma32_kernel<1, 4, 1, 1>));
    else return SM_ERR_UNSUPPORTED;
  } else if constexpr (!DEFORM) {
#ifdef SM_EXPERIMENTS
    if (ws) block = dim3(512);
    if (ws && bco == 128 && bpos == 128) SM_LAUNCH((conv_igemm_kernel<2, 2, 2, 2, false, true, 1>));
    else if (ws && bco == 128 && bpos == 64) SM_LAUNCH((conv_igemm_kernel<2, 2, 2, 1, false, true, 1>));
    else if (ws && bco == 64 && bpos == 64) SM_LAUNCH((conv_igemm_kernel<2, 2, 1, 1, false, true, 1>));
    else if (ws) return SM_ERR_UNSUPPORTED;
    else if (opt == 1 && bco == 128 && bpos == 128) SM_LAUNCH((conv_igemm_kernel<2, 2, 2, 2, false, true, 0, 1>));
    else if (opt == 1 && bco == 128 && bpos == 64) SM_LAUNCH((conv_igemm_kernel<2, 2, 2, 1, false, true, 0, 1>));
    else if (opt == 1 && bco == 64 && bpos == 64) SM_LAUNCH((conv_igemm_kernel<2, 2, 1, 1, false, true, 0, 1>));
    else if (bco == 128 && bpos == 256) SM_LAUNCH((conv_igemm_kernel<2, 2, 2, 4, false, true>));
    else
#endif
    if (bco == 256 && (d->flags & SM_CONV_DBG_HAND_PLACED)) { block = dim3(512); SM_LAUNCH((conv_igemm_kernel<2, 4, 4, 2, false, true, 0, 7>)); }
    else if (bco == 256) { block = dim3(512); SM_LAUNCH((conv_igemm_kernel<2, 4, 4, 2, false, true, 0, 3>)); }
    else if (opt == 3 && bco == 128 && bpos == 128 && (d->flags & SM_CONV_DBG_HAND_PLACED)) SM_LAUNCH((conv_igemm_kernel<2, 2, 2, 2, false, true, 0, 7>));
    else if (opt == 3 && bco == 128 && bpos == 128) SM_LAUNCH((conv_igemm_kernel<2, 2, 2, 2, false, true, 0, 3>));
    else if (opt == 3 && bco == 128 && bpos == 64) SM_LAUNCH((conv_igemm_kernel<2, 2, 2, 1, false, true, 0, 3>));
    else if (opt == 3 && bco == 64 && bpos == 64) SM_LAUNCH((conv_igemm_kernel<2, 2, 1, 1, false, true, 0, 3>));
    else if (bco == 128 && bpos == 128) SM_LAUNCH((conv_igemm_kernel<2, 2, 2, 2, false, true>));          // (cin < 64: the loop with the
    else if (bco == 128 && bpos == 64) SM_LAUNCH((conv_igemm_kernel<2, 2, 2, 1, false, true>));           //  division path in its loader)
    else if (bco == 64 && bpos == 256) SM_LAUNCH((conv_igemm_kernel<1, 4, 2, 2, false, true>));
    else if (bco == 64 && bpos == 128) SM_LAUNCH((conv_igemm_kernel<1, 4, 2, 1, false, true>));
    else if (bco == 64 && bpos == 64) SM_LAUNCH((conv_igemm_kernel<2, 2, 1, 1, false, true>));
    else if (bco == 32 && bpos == 256) SM_LAUNCH((conv_igemm_kernel<1, 4, 1, 2, false, true>));
    else if (bco == 32) SM_LAUNCH((conv_igemm_kernel<1, 4, 1, 1, false, true>));
    else return SM_ERR_UNSUPPORTED;
  }
#undef SM_LAUNCH
  if (S > 1) {
    SplitKEpi e;
    e.part = a.part;
    e.bias = bias;
    e.res = (const uint16_t*)residual;
    e.y = y;
    e.rows = (long long)d->batch * d->out_h[0] * d->out_w[0];
    e.part_rows = a.part_rows;
    e.out_row0 = d->out_row0[0];
    e.S = S;
    e.cout = d->cout;
    e.cpad = d->cout_pad;
    e.out_cstride = d->out_cstride;
    e.out_coff = d->out_coff;
    e.res_cstride = d->res_cstride;
    e.flags = d->flags;
    e.acc_scale = a.acc_scale;
    const long long items = e.rows * (d->cout >> 3);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((items + 255) / 256 > 4096 ? 4096 : (items + 255) / 256)), dim3(256), 0,
                       stream, e);
  }
  SM_LAUNCH_CHECK();
  return SM_OK;
}

}  // namespace

#ifdef SM_OPERAND_F16
// the binary16 build's only external symbol (C++ linkage, called from the bf16 build's launch_conv)
int sm_conv_igemm_f16(const sm_conv_desc* d, const void* x, const void* w, const float* bias, void* y, hipStream_t stream,
                      unsigned long long* gn_stats, void* workspace, long long workspace_bytes) {
  return launch_conv<false>(d, x, nullptr, w, bias, nullptr, y, stream, gn_stats, workspace, workspace_bytes);
}
#else
extern "C" int sm_conv_cout_tile(int cout) { return cout <= 32 ? 32 : (cout <= 64 ? 64 : 128); }

extern "C" int sm_conv_plan_query(const sm_conv_desc* d, int deformable, int with_gn_stats, sm_conv_plan* out) {
  return plan_conv(d, deformable != 0, with_gn_stats != 0, out);
}

extern "C" int sm_conv2d(const sm_conv_desc* d, const void* x, const void* w, const float* bias,
                         const void* residual, void* y, sm_stream_t stream) {
  return launch_conv<false>(d, x, nullptr, w, bias, residual, y, sm_hip_stream(stream));
}

extern "C" int sm_conv2d_ws(const sm_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                            void* y, void* workspace, int64_t workspace_bytes, sm_stream_t stream) {
  return launch_conv<false>(d, x, nullptr, w, bias, residual, y, sm_hip_stream(stream), nullptr, workspace, workspace_bytes);
}

extern "C" int sm_conv2d_gn_stats(const sm_conv_desc* d, const void* x, const float* offset, const void* w,
                                  const float* bias, const void* residual, void* y, int64_t* gn_stats_fix,
                                  sm_stream_t stream) {
  if (!gn_stats_fix) return SM_ERR_BAD_ARG;
  unsigned long long* gn_stats = reinterpret_cast<unsigned long long*>(gn_stats_fix);
  if (offset != nullptr)
    return launch_conv<true>(d, x, offset, w, bias, nullptr, y, sm_hip_stream(stream), gn_stats);
  return launch_conv<false>(d, x, nullptr, w, bias, residual, y, sm_hip_stream(stream), gn_stats);
}

extern "C" int sm_deform_conv2d(const sm_conv_desc* d, const void* x, const float* offset, const void* w,
                                const float* bias, void* y, sm_stream_t stream) {
  return launch_conv<true>(d, x, offset, w, bias, nullptr, y, sm_hip_stream(stream));
}
#endif  // !SM_OPERAND_F16
