// 3x3 / stride 1 / pad 1 convolution with the INPUT PATCH resident in LDS (gfx950).
//
// Why a second conv kernel.  conv_igemm.hip walks K = (tap, cin) and DMAs, per 64-wide K step, a fresh [positions x 64]
// activation tile: every input pixel crosses L2 -> LDS nine times (once per tap).  Its 256x256 tile moves 128 FLOP per
// L2->LDS byte and the LDS-DMA path (~56 B/clk/CU, MI355X_MICROARCH.md) then caps it near half of the MFMA rate; the
// measured ceiling was 970 TFLOP/s on the tower convs.  Here the loop order is (32-channel
// chunk, tap): the block DMAs ONE patch -- its 256 output positions plus the 3x3 halo, 32 channels deep -- and the nine
// taps read it from LDS at nine row shifts.  Activation bytes per K step drop ~4x (weights now dominate: 203 FLOP per
// DMA byte for W = 168), and the DMA instruction count per MFMA from 8/32 to ~5/32.
//
// Geometry.  Positions are tiled in PADDED-FLAT coordinates per image: q = oh * (W+2) + (ow+1), i.e. every image row
// carries one zero column on each side.  A tile is 256 consecutive q of one image (the two pad columns are dummy
// outputs: 1.2 % at W = 168), so a tap (kh, kw) is the constant row shift kh*(W+2) + kw inside the patch and the 32
// positions of an MFMA tile read 32 CONSECUTIVE patch rows: the XOR swizzle of the 64-byte rows stays conflict free
// for any shift (rows distinct mod 16 per ds_read_b128 lane group).  Patch rows = 256 + 2*(W+2) + 2 (598 for W = 168).
// LDS: 2 patch buffers (chunk c computes while chunk c+1 lands) + 2 weight stages of 256 couts x 128 B (2 taps).
// Weights come in their own layout [cout_pad][cin/32][9][32] (K order = chunk, tap, channel), one DMA piece = 8 cout
// rows x 128 B (the two taps of a stage are one cache line of the row).  8 waves (2 cout x 4 pos), wave tile 128 couts x 64 positions, 1 block per CU, one barrier per 2 taps.
// Epilogue: conv_igemm's register epilogue (v_permlane32_swap -> 8 consecutive couts per lane), bias, per-level Scale,
// ReLU, bf16 / f32 stores, fused GroupNorm statistics; multi-level launches and the group dimension (cls + reg towers).
#include <utility>

#include "common.h"
#include "experiments.h"

// Operand type of this translation unit: compiled twice like conv_igemm.hip (csrc/Makefile) -- bf16 operands on
// v_mfma_f32_32x32x16_bf16, and with -DSM_OPERAND_F16 binary16 operands on v_mfma_f32_32x32x16_f16 (SM_CONV_F16: the
// split-precision head plan, f32 output only).  The LDS-DMA loader never interprets the 16-bit payloads.
#ifdef SM_OPERAND_F16
typedef _Float16 frag8 __attribute__((ext_vector_type(8)));
#define SM_MFMA_32x32x16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16((A), (B), (C), 0, 0, 0)
#else
typedef bf16x8 frag8;
#define SM_MFMA_32x32x16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16((A), (B), (C), 0, 0, 0)
// conv3x3_patch_f16.o
int sm_conv3x3_patch_f16(const sm_conv_desc* d, const void* x, const void* w_patch, const float* bias, void* y,
                         unsigned long long* gn_stats, hipStream_t stream);
#endif

namespace {

struct PatchArgs {
  const uint16_t* x;
  const uint16_t* w;
  const float* bias;
  void* y;
  unsigned long long* gn_stats;   // fixed point (common.h: gn_fix)
  int nlev, batch;
  int h[SM_MAX_LEVELS], w_[SM_MAX_LEVELS];
  long long in_row0[SM_MAX_LEVELS], out_row0[SM_MAX_LEVELS];
  // Two PARTS per launch: part 0 = 256-position tiles, part 1 = the remainder of every (level, image) segment in
  // smaller tiles (128 or 192 positions), dispatched after the big ones -- see plan_parts().
  int tile0[2][SM_MAX_LEVELS + 1];   // first position tile of each level (inside one group)
  int tpi[2][SM_MAX_LEVELS];         // position tiles per image of the level
  int qbase[2][SM_MAX_LEVELS];       // first padded-flat position the part covers in a segment
  int nblk[2];                       // blocks of each part (all groups, all cout tiles)
  int small_pos;                     // 128 or 192
  int cin, cout, nc, ntn;         // nc = cin / 32, ntn = cout_pad / 256
  int in_cstride, out_cstride, out_coff;
  long long Kp;                   // weight row pitch (elements) = 9 * cin
  unsigned flags;
  int scale_nch;
  float level_scale[SM_MAX_LEVELS];
  float acc_scale;                // accumulators x this before bias / Scale (1 = none; sm_conv_desc.acc_scale)
  int prow_cap;                   // rows of one patch buffer (multiple of 16)
  int ngroups, tpg[2];
  long long x_grows, y_grows, w_gstride, b_gstride, gn_gstride;
  long long w_lstride, b_lstride;  // per-level weights / bias (0 = the levels share them)
};

template <int N, typename F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl<N>(f, std::make_integer_sequence<int, N>{});
}

__device__ __attribute__((aligned(16))) const unsigned int g_zero16p[4] = {0u, 0u, 0u, 0u};

constexpr int PT_BCO = 256, PT_BPOS = 256, PT_THREADS = 512;

// One tile: 256 couts x (WPOS * TPOS * 32) positions on 8 waves laid out WCO (cout) x WPOS (position), each wave
// TCO x TPOS MFMA tiles of 32 x 32.  <2,4,4,2> is the 256-position tile; <4,2,2,3> / <4,2,2,2> are the 192- / 128-
// position tiles that finish a launch whose last round of 256-tiles would leave most CUs idle.
template <int VAR>
struct PINGPONG_T {
  static constexpr bool value = (VAR & 16) != 0;
};

template <int WCO, int WPOS, int TCO, int TPOS, int VAR>
__device__ __forceinline__ void patch_tile(const PatchArgs& a, const int part, const int tlin, unsigned char* const smem) {
  constexpr int NWV = WCO * WPOS;               // waves of the block: 8 (shipped), 4 in the one-wave-per-SIMD experiment
  constexpr int MAXPP = 48 / NWV;               // patch DMA pieces per wave (<= 768 patch rows = 48 pieces)
  // couts of the tile: 256 (towers, FPN, predictors) or 32 (round 4: the convs with a handful of output channels --
  // sip_mask_lat 512 -> 32, fcos_reg + centerness 256 -> 5 -- whose cost on the implicit-GEMM kernel is the 9x re-read of
  // their INPUT, 0.10 / 0.065 ms per B=4 launch for 20 / 2 GFLOP; here the input crosses L2 -> LDS once)
  constexpr int BCO = WCO * TCO * 32;
  constexpr int WST = BCO * 128;                // one weight stage: BCO cout rows x 128 B (2 taps x 32 channels)
  constexpr int NPIECE = BCO / 8;               // weight DMA pieces (8 rows x 128 B) of a stage
  constexpr int WPW = (NPIECE + NWV - 1) / NWV; // ... per wave
  constexpr int PPS = MAXPP / 3;                // patch pieces per wave per stage (a chunk lands over three stages)
  // Weight ring.  The 256-cout tile double-buffers its weight stages: it is matrix-pipe / power bound and a third buffer
  // measured 3-8 % SLOWER (HISTORY 6).  The 128- and 32-cout tiles (round 4) are the opposite case: 16 / 2 MFMAs per wave and
  // tap against one weight-DMA landing (~0.9 us) per stage at one block per CU -- stage s+2 is issued at the top of stage s,
  // and the stage ends in a COUNTED wait (everything but the weights just issued) + a raw barrier instead of a full drain.
  constexpr bool RING = (BCO != PT_BCO) && !PINGPONG_T<VAR>::value && ((VAR & 0xf) == 0);
  constexpr int NWB = RING ? 3 : 2;
  static_assert((NWV == 8 || NWV == 4) && (BCO == 256 || BCO == 128 || BCO == 32), "8 or 4 waves; 256, 128 or 32 couts");
  // VAR bits: 1 = software-pipelined stage, 2 = staggered DMA issue (waves 4-7 issue theirs between the two taps of a stage,
  // so the two waves of a SIMD are never both stalled in the LDS-DMA issue); 4 / 8 = ABLATIONS for the micro-benchmark
  // (no DMA / no MFMA in the main loop: wrong results by construction, never used by the library's own launches)
  constexpr bool PIPE = (VAR & 1) != 0, STAGGER = (VAR & 2) != 0, NO_DMA = (VAR & 4) != 0, NO_MFMA = (VAR & 8) != 0;
  constexpr bool PINGPONG = (VAR & 16) != 0;
  // VAR bit 32 (round 6, binary16 build only): PAIRED split operands (sm_conv_desc.x3_pairs).  A 64-byte patch row / weight
  // (row, tap) holds 16 channels as [hi 16 | lo 16] instead of 32 channels, so the two fragment sets a tap reads anyway ARE
  // (w_hi, w_lo) and (x_hi, x_lo), and the tap issues the three products  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi  on them: 24
  // MFMAs per 12 fragment reads and per 128 weight bytes of a cout row, where the K-concatenated form [hi | lo | hi] x
  // [hi | hi | lo] (SM_CONV_F16 alone) needs 18 reads and 192 bytes -- a third off the LDS-DMA stream that co-bounds the tile.
  constexpr bool X3P = (VAR & 32) != 0;
  static_assert(!PINGPONG || WCO * WPOS == 8, "the ping-pong schedule pairs waves w and w + 4");
  static_assert(!X3P || ((VAR & 31) == 0 && BCO == PT_BCO), "paired operands: the plain stage loop of the 256-cout tile");
  constexpr int BPOS = WPOS * TPOS * 32;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave / WPOS, wpos = wave % WPOS;
  const int l31 = lane & 31, khalf = lane >> 5;

  // ---- tile decode (wave-uniform)
  const int grp = a.ngroups > 1 ? tlin / a.tpg[part] : 0;
  const int tl_g = tlin - grp * a.tpg[part];
  const int nt = tl_g % a.ntn;
  const int mt = tl_g / a.ntn;
  int lev = 0;
#pragma unroll
  for (int l = 1; l < SM_MAX_LEVELS; ++l)
    if (l < a.nlev && mt >= a.tile0[part][l]) lev = l;
  const int H = a.h[lev], W = a.w_[lev], Wp = W + 2;
  const int ti = mt - a.tile0[part][lev];
  const int n = ti / a.tpi[part][lev];
  const int q0 = a.qbase[part][lev] + (ti - n * a.tpi[part][lev]) * BPOS;
  const long long img_row0 = a.in_row0[lev] + grp * a.x_grows + (long long)n * H * W;

  const int PB = a.prow_cap * 64;                 // bytes of one patch buffer
  unsigned char* const Wb0 = smem;
  unsigned char* const Pb0 = smem + NWB * WST;

  // ---- loader state.  A DMA piece is 16 rows x 64 B: lane L -> row (L >> 2), physical slot (L & 3), so it fetches
  // the logical 16-byte chunk (L & 3) ^ ((row >> 2) & 3) of that row (swizzle on the source side, guide rule 21).
  const int lrow = lane >> 2;
  const int npieces = (BPOS + 2 * Wp + 2 + 15) >> 4;
  const unsigned long long zero_page = (unsigned long long)g_zero16p;
  // patch: this wave owns pieces wave, wave + 8, ...; per piece the lane's source offset (elements) or -1 (zero page)
  int poff[MAXPP];
#pragma unroll
  for (int i = 0; i < MAXPP; ++i) {
    const int piece = wave + NWV * i;
    const int r = piece * 16 + lrow;                           // patch row
    const int qi = q0 - Wp - 1 + r;                            // padded-flat input index
    const int ih = qi >= 0 ? qi / Wp : -1;
    const int iwp = qi - ih * Wp;
    const bool ok = piece < npieces && ih >= 0 && ih < H && iwp >= 1 && iwp <= W;
    const int chunk = (lane & 3) ^ ((r >> 2) & 3);
    poff[i] = ok ? (int)((ih * W + iwp - 1) * a.in_cstride + chunk * 8) : -1;
  }
  const uint16_t* const xbase = a.x + img_row0 * a.in_cstride;
  // weights: the two taps of a stage are 128 CONTIGUOUS bytes of a cout row in the [cout][chunk][tap][32] layout, so a
  // weight piece is 8 cout rows x one whole 128-byte line (8 lanes per line) instead of 16 rows x half a line: what the
  // L2 -> LDS path charges for is the number of lines a wave-instruction touches (PMC: one TCP access per row; 64-byte
  // rows moved 40 GB/s per CU, conv_igemm's 128-byte rows 92), and the weights are 80 % of this kernel's DMA rows.
  // LDS stage = [256 cout rows][128 B]: 16-byte slot (tap * 4 + K chunk) ^ ((row >> 1) & 7) -- the 16 lanes a
  // ds_read_b128 services together (MI355X_MICROARCH.md, LDS) then cover all 64 banks once.
  // A stage = 32 pieces; this wave owns pieces wave*4 .. wave*4+3.
  const uint16_t* wsrc[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int piece = wave * WPW + i;
    const int row = (piece < NPIECE ? piece : 0) * 8 + (lane >> 3);      // (pieces beyond the stage are never issued)
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    wsrc[i] = a.w + grp * a.w_gstride + lev * a.w_lstride + (long long)(nt * BCO + row) * a.Kp + chunk * 8;
  }
  auto dma_w = [&](int stage, int buf) {            // weight stage `stage` (K steps 2*stage, 2*stage+1) -> Wb[buf]
    unsigned char* dst = Wb0 + buf * WST;
    sfor<WPW>([&](auto I) {
      constexpr int i = decltype(I)::value;
      const int piece = wave * WPW + i;
      if (NPIECE % NWV == 0 || piece < NPIECE)             // wave-uniform; compile-time true for the 256-cout tile
        __builtin_amdgcn_global_load_lds((glb_void*)(wsrc[i] + (long long)stage * 64),
                                         (lds_void*)(dst + piece * 1024), 16, 0, 0);
    });
  };
  auto dma_patch_piece = [&](auto I, int chunk_c, int buf) {     // piece I of this wave, channel chunk c -> Pb[buf]
    constexpr int i = decltype(I)::value;
    const int piece = wave + NWV * i;
    if (piece < npieces) {                                       // wave-uniform
      const unsigned long long pm = poff[i] >= 0 ? ~0ull : 0ull;
      const unsigned long long src = ((unsigned long long)(xbase + poff[i] + chunk_c * 32) & pm) | (zero_page & ~pm);
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(Pb0 + buf * PB + piece * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[TCO][TPOS];
#pragma unroll
  for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
    for (int tp = 0; tp < TPOS; ++tp)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tc][tp][e] = 0.f;

  const int rsw = (l31 >> 1) & 7;
  const int wrow_off = (wco * (TCO * 32) + l31) * 128;
  const int prow0 = wpos * (TPOS * 32) + l31;              // patch row of this lane's position at tap (0,0), tp = 0

  // one tap: 2 K sub-steps of 16, 8 MFMAs each; both sub-steps' fragments are requested up front and hipcc schedules the
  // stage (2 taps = 24 reads + 32 MFMAs, one basic block).  A/B (round 2): the same stage with the reads of sub-step
  // u+1 pinned under the MFMAs of u and the DMA issues pinned behind every second MFMA (sched_barrier(0) per slot)
  // needed 256 VGPRs + 19 spills and ran 764 vs 835 TFLOP/s on the B=2 tower launch -- not kept.
  auto tap = [&](const unsigned char* Wst, const int hx, const unsigned char* P, int shift) {   // hx = 64 * (tap of the stage)
    frag8 wf[2][TCO], xf[2][TPOS];
    int pr[TPOS], psw[TPOS];
#pragma unroll
    for (int t = 0; t < TPOS; ++t) {
      pr[t] = prow0 + t * 32 + shift;
      psw[t] = (pr[t] >> 2) & 3;
    }
    auto rd = [&](int kk, int set) {
#pragma unroll
      for (int t = 0; t < TCO; ++t)
        wf[set][t] = *reinterpret_cast<const frag8*>(Wst + wrow_off + t * 32 * 128 + ((((kk * 2 + khalf) ^ rsw) * 16) ^ hx));
#pragma unroll
      for (int t = 0; t < TPOS; ++t)
        xf[set][t] = *reinterpret_cast<const frag8*>(P + pr[t] * 64 + (((kk * 2 + khalf) ^ psw[t]) * 16));
    };
    rd(0, 0);
    rd(1, 1);
    if constexpr (X3P) {
      // set 0 = the hi halves of the chunk's 16 channels, set 1 = the lo halves: three cross products per accumulator tile,
      // term-major so that consecutive MFMAs never touch the same accumulator
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
          for (int tp = 0; tp < TPOS; ++tp)
            acc[tc][tp] = SM_MFMA_32x32x16(wf[term == 2 ? 1 : 0][tc], xf[term == 1 ? 1 : 0][tp], acc[tc][tp]);
    } else {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
        for (int tp = 0; tp < TPOS; ++tp)
          acc[tc][tp] = SM_MFMA_32x32x16(wf[kk][tc], xf[kk][tp], acc[tc][tp]);
    }
  };

  // ---- main loop over channel-chunk PAIRS: 18 taps = 9 weight stages per iteration.
  // patch(c) lives in Pb[c & 1]; patch(c0+1) is DMAed during stages 0-2 (before tap 9 needs it in stage 4), patch(c0+2)
  // during stages 5-7 (after tap 8, the last reader of Pb[0], finished in stage 4); weight stage s+1 during stage s.
  const int npair = a.nc >> 1;
  const int nstage = npair * 9;
  sfor<MAXPP>([&](auto I) { dma_patch_piece(I, 0, 0); });
  dma_w(0, 0);
  if constexpr (RING) {
    if (1 < nstage) dma_w(1, 1);
  }
  __syncthreads();
  if constexpr (PINGPONG) {
    // ---- ping-pong schedule (MI355X_MICROARCH.md, "Two waves per SIMD"): the block's waves form two groups, A = waves
    // 0-3 and B = waves 4-7 (one of each per SIMD).  Time runs in slots separated by raw s_barriers; in every slot one
    // group issues the 2*TCO*TPOS MFMAs of a tap back to back -- it has the SIMD's matrix pipe to itself -- while the other
    // group reads ITS fragments of the next tap from LDS and issues its share of the LDS-DMA.  Slot 2T: A computes tap T,
    // B loads tap T; slot 2T+1: A loads tap T+1, B computes tap T.  One fragment register set per wave (load and compute
    // alternate).  The profile of the stage-synchronous loop above (tools/patch_variants_bench.py): both waves of a SIMD
    // wait for LDS / the barrier at the same moments, 54 us per tile even with the DMA removed against 31 us of MFMA time.
    // Hazards (T = 2*stage + hh; reads of tap T happen in slots 2T-1 (A) and 2T (B)):
    //  * weight buffer (s+1)&1 was last read for tap 2s-1 in slot 4s-2; A issues stage s+1's DMA in slot 4s-1, B in slot
    //    4s; every wave drains its DMA (vmcnt(0)) at the end of slot 4s+2, the first read of the data is in slot 4s+3;
    //  * patch pieces ride with the weight DMA of the same stages as before (chunk c0+1 with stages 0-2, c0+2 with 5-7);
    //  * a loading wave waits for its own ds_reads (lgkmcnt(0)) before the barrier, so a buffer is never overwritten
    //    while a read of it is in flight.  The barriers are raw s_barrier: they must not drain the DMA queue.
    const bool grp_a = wave < 4;
    frag8 wf[2][TCO], xf[2][TPOS];
    auto load_tap = [&](int st, auto SC) {                 // fragments of tap SC (0..17 inside the pair) of stage st
      constexpr int sidx = decltype(SC)::value;
      constexpr int hh = sidx & 1;
      constexpr int cc = sidx / 9, t9 = sidx % 9, kh = t9 / 3, kw = t9 % 3;
      const unsigned char* Wh = Wb0 + (st & 1) * WST;
      const unsigned char* P = Pb0 + cc * PB;
      // the patch-row addresses of the 18 taps are invariant over the channel-chunk loop and hipcc hoists all of them
      // (36+ VGPRs held across the loop -> spills next to 128 accumulators + 48 fragment registers); an opaque copy of
      // the row pitch makes them a per-tap recomputation of ~5 VALU in the load slot instead
      int wp_l = Wp;
      asm volatile("" : "+s"(wp_l));
      const int shift = kh * wp_l + kw;
      // t*32 rows do not change (row >> 2) & 3: one swizzle per tap, the K-half is one XOR with 32 bytes
      const int pr0 = prow0 + shift;
      const int xa = pr0 * 64 + ((khalf ^ ((pr0 >> 2) & 3)) * 16);
      const int wa = wrow_off + (((hh * 4 + khalf) ^ rsw) * 16);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int t = 0; t < TCO; ++t)
          wf[kk][t] = *reinterpret_cast<const frag8*>(Wh + (wa ^ (kk * 32)) + t * 32 * 128);
#pragma unroll
        for (int t = 0; t < TPOS; ++t)
          xf[kk][t] = *reinterpret_cast<const frag8*>(P + (xa ^ (kk * 32)) + t * 32 * 64);
      }
    };
    auto compute_tap = [&]() {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
          for (int tp = 0; tp < TPOS; ++tp)
            acc[tc][tp] = SM_MFMA_32x32x16(wf[kk][tc], xf[kk][tp], acc[tc][tp]);
    };
    auto issue_stage = [&](int cpv, auto SPC) {            // the DMA the old loop issues at the top of stage (cpv, sp)
      constexpr int sp = decltype(SPC)::value;
      const int st = cpv * 9 + sp, c0 = 2 * cpv;
      if (st + 1 < nstage) dma_w(st + 1, (st + 1) & 1);
      if constexpr (sp < 3) {
        dma_patch_piece(std::integral_constant<int, 2 * sp>{}, c0 + 1, 1);
        dma_patch_piece(std::integral_constant<int, 2 * sp + 1>{}, c0 + 1, 1);
      } else if constexpr (sp >= 5 && sp < 8) {
        if (cpv + 1 < npair) {
          dma_patch_piece(std::integral_constant<int, 2 * (sp - 5)>{}, c0 + 2, 0);
          dma_patch_piece(std::integral_constant<int, 2 * (sp - 5) + 1>{}, c0 + 2, 0);
        }
      }
    };
    // sched_barrier(0): nothing may be scheduled across -- an asm statement orders memory operations only, and hipcc moved
    // MFMAs (register-only) and fragment reads from one slot into the next, which is exactly what the slots forbid
    auto bar = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    auto wait_lds = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto wait_dma = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    // ONE instruction stream for both groups -- L(0) C(0) L(1) C(1) ... with a barrier after every segment -- in which group
    // A skips the very first barrier: from then on A runs one segment ahead of B (A computes tap T while B loads it, A loads
    // tap T+1 while B computes T).  A executes one extra barrier at the end.
    for (int cp = 0; cp < npair; ++cp) {
      sfor<18>([&](auto SC) {
        constexpr int sidx = decltype(SC)::value;           // tap inside the pair
        constexpr int sp = sidx >> 1, hh = sidx & 1;
        const int st = cp * 9 + sp;
        if constexpr (hh == 0 && !NO_DMA) issue_stage(cp, std::integral_constant<int, sp>{});
        load_tap(st, SC);                                   // segment L(T)
        wait_lds();
        if constexpr (hh == 1) {
          if (!grp_a) wait_dma();
        }
        if (!(grp_a && cp == 0 && sidx == 0)) bar();
        if constexpr (!NO_MFMA) compute_tap();              // segment C(T)
        if constexpr (hh == 1) {
          if (grp_a) wait_dma();
        }
        bar();
      });
    }
    if (grp_a) bar();
    __syncthreads();
  } else
  for (int cp = 0; cp < npair; ++cp) {
    const int c0 = 2 * cp;
    sfor<9>([&](auto SP) {
      constexpr int sp = decltype(SP)::value;
      const int st = cp * 9 + sp;                                  // global stage index (parity = W buffer)
      auto issue_dma = [&]() {
        if constexpr (!NO_DMA) {
          if constexpr (!RING) {
            if (st + 1 < nstage) dma_w(st + 1, (st + 1) & 1);
          }
          if constexpr (sp < 3) {                                    // patch of the pair's second chunk
            sfor<PPS>([&](auto J) { dma_patch_piece(std::integral_constant<int, PPS * sp + decltype(J)::value>{}, c0 + 1, 1); });
          } else if constexpr (sp >= 5 && sp < 8) {                  // patch of the NEXT pair's first chunk
            if (cp + 1 < npair) {
              sfor<PPS>([&](auto J) { dma_patch_piece(std::integral_constant<int, PPS * (sp - 5) + decltype(J)::value>{}, c0 + 2, 0); });
            }
          }
          if constexpr (RING) {                                      // LAST in issue order: the counted wait below skips exactly these
            if (st + 2 < nstage) dma_w(st + 2, (st + 2) % 3);
          }
        }
      };
      const bool early = !STAGGER || wave < 4;                      // wave-uniform (SGPR)
      if (early) issue_dma();
      const unsigned char* Wst = Wb0 + (RING ? st % 3 : (st & 1)) * WST;
      if constexpr (PIPE) {
        // software-pipelined stage: 4 sub-steps (tap hh, K half kk) of TCO*TPOS MFMAs; the fragments of sub-step i+1 are
        // read into the other register set behind the MFMAs of sub-step i ("1 MFMA, 1 ds_read" ladder), so only the
        // first sub-step after the barrier waits for the LDS.  hipcc's own schedule (the else branch) re-uses one set per
        // tap: read -> lgkmcnt(0) -> 2-4 MFMAs, eight exposed LDS round trips per stage.
        constexpr int NFR = TCO + TPOS, NMF = TCO * TPOS, NPAIR = NFR < NMF ? NFR : NMF;
        frag8 wf[2][TCO], xf[2][TPOS];
        auto rd = [&](auto IC, int set) {
          constexpr int i = decltype(IC)::value;
          constexpr int hh = i >> 1, kk = i & 1;
          constexpr int s = 2 * sp + hh;
          constexpr int cc = s / 9, t9 = s % 9, kh = t9 / 3, kw = t9 % 3;
          const unsigned char* P = Pb0 + cc * PB;
          const int shift = kh * Wp + kw;
#pragma unroll
          for (int t = 0; t < TCO; ++t)
            wf[set][t] = *reinterpret_cast<const frag8*>(Wst + wrow_off + t * 32 * 128 + (((hh * 4 + kk * 2 + khalf) ^ rsw) * 16));
#pragma unroll
          for (int t = 0; t < TPOS; ++t) {
            const int pr = prow0 + t * 32 + shift;
            xf[set][t] = *reinterpret_cast<const frag8*>(P + pr * 64 + (((kk * 2 + khalf) ^ ((pr >> 2) & 3)) * 16));
          }
        };
        rd(std::integral_constant<int, 0>{}, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NFR, 0);
        sfor<4>([&](auto IC) {
          constexpr int i = decltype(IC)::value;
          if constexpr (i == 2 && STAGGER) {
            if (!early) issue_dma();
          }
          if constexpr (i < 3) rd(std::integral_constant<int, i + 1>{}, (i + 1) & 1);
#pragma unroll
          for (int tc = 0; tc < TCO; ++tc)
#pragma unroll
            for (int tp = 0; tp < TPOS; ++tp)
              if constexpr (!NO_MFMA)
                acc[tc][tp] = SM_MFMA_32x32x16(wf[i & 1][tc], xf[i & 1][tp], acc[tc][tp]);
          if constexpr (i < 3) {
#pragma unroll
            for (int j = 0; j < NPAIR; ++j) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if constexpr (NFR > NPAIR) __builtin_amdgcn_sched_group_barrier(0x100, NFR - NPAIR, 0);
            if constexpr (NMF > NPAIR) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NPAIR, 0);
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
          }
        });
      } else {
      sfor<2>([&](auto HH) {
        constexpr int hh = decltype(HH)::value;
        constexpr int s = 2 * sp + hh;                             // tap index inside the pair, 0..17
        constexpr int cc = s / 9, t = s % 9, kh = t / 3, kw = t % 3;
        if constexpr (hh == 1 && STAGGER) {
          if (!early) issue_dma();
        }
        if constexpr (!NO_MFMA) tap(Wst, hh * 64, Pb0 + cc * PB, kh * Wp + kw);
      });
      }
      // (A/B round 3: a 3-stage weight ring -- stage st+2 issued at the top of stage st, vmcnt(4) + raw s_barrier at its end,
      // so a stage of weights has two stage times to land -- is 3-8 % SLOWER where it fits (image width <= 125):
      // profiles/r03_patch_weight_ring_ab.txt.  The landing latency of the DMA is not what the stage waits for.)
      if constexpr (RING) {
        // weights of stage st+1 (issued a stage ago) and this stage's patch pieces must have landed; the weights of stage
        // st+2, issued last, may stay in flight.  vmcnt retires in order, so "all but my last WPW" is exact for the waves
        // that own weight pieces; the others drain.  Raw barrier: a __syncthreads would drain the queue again.
        __builtin_amdgcn_sched_barrier(0);
        const bool mine = wave * WPW < NPIECE && st + 2 < nstage;     // wave-uniform
        if (mine) {
          if constexpr (WPW == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      } else {
        __syncthreads();                                           // drains the DMA queue (vmcnt(0)) and fences the buffers
      }
    });
  }
  if constexpr (RING) __syncthreads();                             // the epilogue reuses the LDS: everything landed, everyone done

  // ---- epilogue (register epilogue of conv_igemm.hip): lanes i / i+32 swap 4-cout groups -> 8 consecutive couts
  const float lscale = a.level_scale[lev];
  const float* const biasp = a.bias != nullptr ? a.bias + grp * a.b_gstride + lev * a.b_lstride : nullptr;
  unsigned long long* const gnp = a.gn_stats != nullptr ? a.gn_stats + grp * a.gn_gstride : nullptr;
  const bool out_f32 = a.flags & SM_CONV_OUT_F32;
  const long long out_img_row0 = a.out_row0[lev] + grp * a.y_grows + (long long)n * H * W;
  unsigned long long* gn_bins = reinterpret_cast<unsigned long long*>(smem);                  // [256/8][2]; the K loop's last barrier freed the LDS
  const bool gn = gnp != nullptr;
  if (gn) {
    if (tid < 64) gn_bins[tid] = 0ull;
    __syncthreads();
  }
#pragma unroll
  for (int tp = 0; tp < TPOS; ++tp) {
    const int q = q0 + wpos * (TPOS * 32) + tp * 32 + l31;
    const int oh = q / Wp;
    const int owp = q - oh * Wp;
    const bool pvalid = oh < H && owp >= 1 && owp <= W;
    const long long orow = out_img_row0 + (long long)oh * W + owp - 1;
    float gpart[TCO * 4];                           // (sum, sum of squares) of this position's 8 couts, per (tc, qp)
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
        if (a.acc_scale != 1.f) {                   // block-uniform: the x3 plan's power-of-two weight scale, undone exactly
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= a.acc_scale;
        }
        const int cl = wco * (TCO * 32) + tc * 32 + 8 * (2 * qp + khalf);
        const int c0 = nt * BCO + cl;
        const bool live = pvalid && c0 < a.cout;
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
        }
        if (gn) {
          float gs = 0.f, gss = 0.f;
          if (live) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              gs += v[e];
              gss = __builtin_fmaf(v[e], v[e], gss);   // explicit: left to -ffp-contract, instantiations differed in the last bit
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
    if (gn) {                                       // wave-uniform: the shuffles need every lane
      // 32 positions x 8 couts per total, summed over the half-wave in the fixed butterfly order; lane group k = l31 >> sh
      // ends up with total k = ((tc * 2 + qp) * 2 + stat) and rounds it ONCE to the fixed-point grid
      constexpr int sh = TCO == 2 ? 2 : (TCO == 1 ? 3 : (TCO == 4 ? 1 : 0));
      static_assert((TCO * 4) << sh == 32, "TCO");
      const float tot = gn_half_wave_totals<TCO * 4>(gpart, l31);
      const int k = l31 >> sh;
      const int cl = wco * (TCO * 32) + (k >> 2) * 32 + 8 * (2 * ((k >> 1) & 1) + khalf);
      if ((l31 & ((1 << sh) - 1)) == 0 && nt * BCO + cl < a.cout) atomicAdd(&gn_bins[(cl >> 3) * 2 + (k & 1)], gn_fix(tot));
    }
  }
  if (gn) {                                          // the whole tile lies in image n of level lev
    __syncthreads();
    if (tid < 64) {
      const unsigned long long v = gn_bins[tid];
      const int g = (nt * BCO >> 3) + (tid >> 1);
      if (v != 0ull && g < (a.cout >> 3))
        atomicAdd(gnp + (((long long)n * a.nlev + lev) * (a.cout >> 3) + g) * 2 + (tid & 1), v);
    }
  }
}


// XCD-contiguous tile ranges (blockIdx round-robins over the 8 XCDs), as in conv_igemm.hip
__device__ __forceinline__ int xcd_tile(int b, int nblk) {
  const int xcd = b & 7, xq = nblk >> 3, xr = nblk & 7;
  return (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
}

// blocks [0, nblk[0]) are 256-position tiles, blocks [nblk[0], nblk[0] + nblk[1]) the SMALL-position tiles: the
// dispatcher hands out blocks in index order, so a CU that retires a big tile picks up a small one.
template <int SMALL, int PIPE>
__global__ __launch_bounds__(PT_THREADS, 1) void conv3x3_patch_kernel(const PatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [W stage 0][W stage 1][patch 0][patch 1]
  const int b = blockIdx.x;
  if (b < a.nblk[0]) {
    patch_tile<2, 4, 4, 2, PIPE>(a, 0, xcd_tile(b, a.nblk[0]), smem);
  } else {
    if constexpr (SMALL == 128)
      patch_tile<4, 2, 2, 2, PIPE>(a, 1, xcd_tile(b - a.nblk[0], a.nblk[1]), smem);
    else
      patch_tile<4, 2, 2, 3, PIPE>(a, 1, xcd_tile(b - a.nblk[0], a.nblk[1]), smem);
  }
}

// 32 couts x 256 positions on 8 waves (one 32 x 32 MFMA tile per wave per K sub-step), uniform launches only
template <int PIPE>
__global__ __launch_bounds__(PT_THREADS, 1) void conv3x3_patch_n32_kernel(const PatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [W stage 0][W stage 1][patch 0][patch 1]
  patch_tile<1, 8, 1, 1, PIPE>(a, 0, xcd_tile(blockIdx.x, a.nblk[0]), smem);
}

// 128 couts x 256 positions on 8 waves (2 cout x 4 position waves, 64 x 64 per wave), uniform launches only: the 3x3 convs
// whose position count gives too few 256-cout tiles (ResNet layer3 / layer4: 66 / 17 position tiles at B=4)
template <int PIPE>
__global__ __launch_bounds__(PT_THREADS, 1) void conv3x3_patch_n128_kernel(const PatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  patch_tile<2, 4, 2, 2, PIPE>(a, 0, xcd_tile(blockIdx.x, a.nblk[0]), smem);
}

#ifdef SM_EXPERIMENTS
// EXPERIMENT (round 3): the same 256 x 256 tile on FOUR waves, 128 couts x 128 positions each (16 accumulator tiles = 256
// registers, beyond the 256 architectural VGPRs: one wave per SIMD, accumulators in AGPRs), 8 fragment reads per 16 MFMAs
// instead of 6 per 8.  Uniform launches only.
template <int VAR>
__global__ __launch_bounds__(256, 1) void conv3x3_patch_kernel_w4(const PatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  patch_tile<2, 2, 4, 4, VAR>(a, 0, xcd_tile(blockIdx.x, a.nblk[0]), smem);
}
#endif

// cout tile of a descriptor: weights padded to 32 rows select the 32-cout kernel, else 256-row tiles
int patch_bco(const sm_conv_desc* d) { return d->patch_cout_tile == 128 ? 128 : (d->cout_pad == 32 ? 32 : PT_BCO); }

int patch_check(const sm_conv_desc* d) {
  if (!d) return SM_ERR_BAD_ARG;
  if (d->nlev < 1 || d->nlev > SM_MAX_LEVELS || d->batch < 1) return SM_ERR_BAD_SHAPE;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->dil != 1) return SM_ERR_UNSUPPORTED;
  if (d->cin % 64 != 0 || d->cin < 64 || d->in_cstride % 8 != 0) return SM_ERR_UNSUPPORTED;
  if (d->patch_cout_tile != 0 && d->patch_cout_tile != 128) return SM_ERR_UNSUPPORTED;
  if (d->cout < 1 || d->cout_pad < d->cout || d->cout_pad % patch_bco(d) != 0) return SM_ERR_UNSUPPORTED;
  if (patch_bco(d) != PT_BCO && (d->ngroups > 1 || d->w_level_stride != 0)) return SM_ERR_UNSUPPORTED;   // plain launches only
  if ((d->cout & 7) || (d->out_cstride & 7) || (d->out_coff & 7)) return SM_ERR_UNSUPPORTED;
  if (d->flags & (SM_CONV_RES_ADD | SM_CONV_RES_NEAREST | SM_CONV_IN_RELU)) return SM_ERR_UNSUPPORTED;
  if (d->w_batch_stride != 0) return SM_ERR_UNSUPPORTED;
  // paired split operands: binary16 build, 256-cout tiles, f32 output
  if (d->x3_pairs != 0 && (d->x3_pairs != 1 || !(d->flags & SM_CONV_F16) || patch_bco(d) != PT_BCO)) return SM_ERR_UNSUPPORTED;
  for (int l = 0; l < d->nlev; ++l) {
    if (d->in_h[l] != d->out_h[l] || d->in_w[l] != d->out_w[l] || d->in_h[l] < 1 || d->in_w[l] < 1) return SM_ERR_BAD_SHAPE;
    if (PT_BPOS + 2 * (d->in_w[l] + 2) + 2 > 16 * 48) return SM_ERR_UNSUPPORTED;   // patch rows the loader covers (48 pieces)
    if ((long long)d->in_h[l] * d->in_w[l] * d->in_cstride >= (1ll << 31)) return SM_ERR_UNSUPPORTED;   // 32-bit patch offsets
  }
  return SM_OK;
}


// ---- launch shape.  Every (level, image[, group, cout tile]) segment of L = H * (W + 2) padded-flat positions gets
// k256[l] tiles of 256 positions followed by tiles of `small` positions for the rest.  One block per CU: a launch of T
// equal tiles takes ceil(T / CUS) tile times however empty the last round is (the B=2 tower launch: 368 tiles = two
// rounds for 1.44 rounds of work).  The planner tries "r full rounds of big tiles, remainder small" for every r and both
// small sizes and keeps the cheapest by a list-scheduling estimate (cost of a tile = positions + a fixed 24-position
// equivalent for prologue / epilogue).
constexpr int PT_CUS = 256;

struct PatchShape {
  int k256[SM_MAX_LEVELS];
  int small;        // 128 / 192
  long long nbig, nsmall;
};

long long seg_len(const sm_conv_desc* d, int l) { return (long long)d->in_h[l] * (d->in_w[l] + 2); }

double shape_cost(long long nbig, long long nsmall, int small) {
  // list schedule on PT_CUS CUs: big tiles round-robin first, then every small tile to the earliest-free CU
  const double cb = 256 + 24, cs = small + 24;
  const long long rb = nbig / PT_CUS, eb = nbig % PT_CUS;      // eb CUs carry one more big tile
  double t_a = rb * cb, t_b = (rb + 1) * cb;                   // class a: PT_CUS - eb CUs, class b: eb CUs
  const long long n_a = PT_CUS - eb, n_b = eb;
  double end = nbig ? (eb ? t_b : t_a) : 0.0;
  long long left = nsmall;
  while (left > 0) {
    const bool use_a = n_b == 0 || t_a <= t_b;
    double& t = use_a ? t_a : t_b;
    const long long n = use_a ? n_a : n_b;
    left -= left < n ? left : n;
    t += cs;
    end = end > t ? end : t;
  }
  return end;
}

void plan_shape(const sm_conv_desc* d, PatchShape* ps) {
  const int ntn = d->cout_pad / patch_bco(d), ng = d->ngroups > 1 ? d->ngroups : 1;
  const long long segs = (long long)d->batch * ntn * ng;       // segments per level
  long long full[SM_MAX_LEVELS];                               // whole 256-tiles a segment of the level can hold
  long long all_big = 0;
  for (int l = 0; l < d->nlev; ++l) {
    full[l] = seg_len(d, l) / PT_BPOS;
    all_big += segs * sm_cdiv(seg_len(d, l), PT_BPOS);
  }
  // baseline: the uniform launch (every tile 256 positions)
  double best = shape_cost(all_big, 0, 128);
  for (int l = 0; l < SM_MAX_LEVELS; ++l) ps->k256[l] = l < d->nlev ? (int)sm_cdiv(seg_len(d, l), PT_BPOS) : 0;
  ps->small = 128;
  ps->nbig = all_big;
  ps->nsmall = 0;
  if ((d->flags & SM_CONV_DBG_PATCH_UNIFORM) || patch_bco(d) != PT_BCO) return;   // (the 32- / 128-cout kernels have the 256-position tile only)
  const int smalls[2] = {128, 192};
  const long long max_rounds = all_big / PT_CUS + 1;
  for (int si = 0; si < 2; ++si) {
    const int small = smalls[si];
    if ((d->flags & SM_CONV_DBG_PATCH_SMALL128) && small != 128) continue;
    if ((d->flags & SM_CONV_DBG_PATCH_SMALL192) && small != 192) continue;
    for (long long r = 0; r <= max_rounds; ++r) {
      // spend a budget of r * PT_CUS big tiles on the levels in order (largest first = level 0 first)
      long long budget = r * PT_CUS, nbig = 0, nsmall = 0;
      int k[SM_MAX_LEVELS] = {};
      for (int l = 0; l < d->nlev; ++l) {
        long long kl = budget / segs;
        if (kl > full[l]) kl = full[l];
        k[l] = (int)kl;
        budget -= kl * segs;
        nbig += kl * segs;
        const long long rest = seg_len(d, l) - kl * PT_BPOS;
        nsmall += segs * sm_cdiv(rest, small);
      }
      const double c = shape_cost(nbig, nsmall, small);
      if (c < best * 0.97) {                                   // only leave the uniform launch for a real gain
        best = c;
        for (int l = 0; l < SM_MAX_LEVELS; ++l) ps->k256[l] = k[l];
        ps->small = small;
        ps->nbig = nbig;
        ps->nsmall = nsmall;
      }
    }
  }
}

}  // namespace

#ifndef SM_OPERAND_F16
extern "C" int sm_conv3x3_patch_supported(const sm_conv_desc* d) { return patch_check(d) == SM_OK ? 1 : 0; }

extern "C" int64_t sm_conv3x3_patch_tiles(const sm_conv_desc* d) {
  if (patch_check(d) != SM_OK) return 0;
  PatchShape ps;
  plan_shape(d, &ps);
  return ps.nbig + ps.nsmall;
}

extern "C" int sm_conv3x3_patch_plan(const sm_conv_desc* d, int64_t* out) {
  if (!out) return SM_ERR_BAD_ARG;
  const int rc = patch_check(d);
  if (rc != SM_OK) return rc;
  PatchShape ps;
  plan_shape(d, &ps);
  out[0] = ps.nbig;
  out[1] = ps.nsmall;
  out[2] = ps.small;
  out[3] = (int64_t)(shape_cost(ps.nbig, ps.nsmall, ps.small) * 1000.0 / (256 + 24));   // milli tile-times
  return SM_OK;
}

#endif

#ifdef SM_OPERAND_F16
int sm_conv3x3_patch_f16(const sm_conv_desc* d, const void* x, const void* w_patch, const float* bias, void* y,
                         unsigned long long* gn_stats, hipStream_t s) {
  if (!x || !w_patch || !y) return SM_ERR_BAD_ARG;
  const int rc = patch_check(d);
  if (rc != SM_OK) return rc;
  if (!(d->flags & SM_CONV_OUT_F32)) return SM_ERR_UNSUPPORTED;       // binary16 operands: f32 output only
#else
extern "C" int sm_conv3x3_patch(const sm_conv_desc* d, const void* x, const void* w_patch, const float* bias, void* y,
                                int64_t* gn_stats_fix, sm_stream_t stream) {
  if (!x || !w_patch || !y) return SM_ERR_BAD_ARG;
  unsigned long long* gn_stats = reinterpret_cast<unsigned long long*>(gn_stats_fix);
  const int rc = patch_check(d);
  if (rc != SM_OK) return rc;
  hipStream_t s = sm_hip_stream(stream);
  if (d->flags & SM_CONV_F16) return sm_conv3x3_patch_f16(d, x, w_patch, bias, y, gn_stats, s);
#endif
  PatchShape ps;
  plan_shape(d, &ps);
  PatchArgs a;
  a.x = (const uint16_t*)x;
  a.w = (const uint16_t*)w_patch;
  a.bias = bias;
  a.y = y;
  a.gn_stats = gn_stats;
  a.nlev = d->nlev;
  a.batch = d->batch;
  a.small_pos = ps.small;
  int t0 = 0, t1 = 0, maxw = 1;
  for (int l = 0; l < SM_MAX_LEVELS; ++l) {
    const bool on = l < d->nlev;
    a.h[l] = on ? d->in_h[l] : 1;
    a.w_[l] = on ? d->in_w[l] : 1;
    a.in_row0[l] = on ? d->in_row0[l] : 0;
    a.out_row0[l] = on ? d->out_row0[l] : 0;
    a.level_scale[l] = on ? d->level_scale[l] : 1.f;
    const long long len = on ? seg_len(d, l) : 0;
    const int k = on ? ps.k256[l] : 0;
    const long long rest = len - (long long)k * PT_BPOS;
    const int ks = on && rest > 0 ? sm_cdiv(rest, ps.small) : 0;
    a.tile0[0][l] = t0;
    a.tile0[1][l] = t1;
    a.tpi[0][l] = k > 0 ? k : 1;              // divisor only; a level without tiles of a part is never decoded
    a.tpi[1][l] = ks > 0 ? ks : 1;
    a.qbase[0][l] = 0;
    a.qbase[1][l] = k * PT_BPOS;
    if (on) {
      t0 += d->batch * k;
      t1 += d->batch * ks;
      if (d->in_w[l] > maxw) maxw = d->in_w[l];
    }
  }
  a.tile0[0][SM_MAX_LEVELS] = t0;
  a.tile0[1][SM_MAX_LEVELS] = t1;
  a.cin = d->cin;
  a.cout = d->cout;
  a.nc = d->cin / 32;
  const int bco = patch_bco(d);
  a.ntn = d->cout_pad / bco;
  a.in_cstride = d->in_cstride;
  a.out_cstride = d->out_cstride;
  a.out_coff = d->out_coff;
  a.Kp = 9ll * d->cin;
  a.flags = d->flags;
  a.scale_nch = d->scale_nch;
  a.acc_scale = (d->acc_scale == 0.f) ? 1.f : d->acc_scale;
  a.prow_cap = (PT_BPOS + 2 * (maxw + 2) + 2 + 15) / 16 * 16;
  a.ngroups = d->ngroups > 1 ? d->ngroups : 1;
  a.tpg[0] = t0 * a.ntn > 0 ? t0 * a.ntn : 1;
  a.tpg[1] = t1 * a.ntn > 0 ? t1 * a.ntn : 1;
  a.x_grows = d->x_group_rows;
  a.y_grows = d->y_group_rows;
  a.w_gstride = d->w_group_stride;
  a.b_gstride = d->bias_group_stride;
  a.gn_gstride = d->gn_group_stride;
  a.w_lstride = d->w_level_stride;
  a.b_lstride = d->bias_level_stride;
  const long long nb0 = (long long)t0 * a.ntn * a.ngroups, nb1 = (long long)t1 * a.ntn * a.ngroups;
  if (nb0 != ps.nbig || nb1 != ps.nsmall) return SM_ERR_BAD_SHAPE;       // planner / table mismatch: a bug, not a shape
  a.nblk[0] = (int)nb0;
  a.nblk[1] = (int)nb1;
  if (gn_stats != nullptr) {
    if (sm_zero_async(gn_stats, sizeof(unsigned long long) * 2 * a.ngroups * d->batch * d->nlev * (d->cout / 8), s) != hipSuccess)
      return SM_ERR_LAUNCH;
  }
  const size_t lds = (bco == PT_BCO ? 2 : 3) * (size_t)bco * 128 + 2 * (size_t)a.prow_cap * 64;   // (weight ring of the small tiles)
  if (lds > 160 * 1024) return SM_ERR_UNSUPPORTED;
  const long long nblk = nb0 + nb1;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return SM_ERR_BAD_SHAPE;
  const int var = ((d->flags & SM_CONV_DBG_PATCH_PIPE) ? 1 : 0) | ((d->flags & SM_CONV_DBG_PATCH_STAGGER) ? 2 : 0) |
                  ((d->flags & SM_CONV_DBG_PATCH_NO_DMA) ? 4 : 0) | ((d->flags & SM_CONV_DBG_PATCH_NO_MFMA) ? 8 : 0) |
                  ((d->flags & SM_CONV_DBG_PATCH_PINGPONG) ? 16 : 0) | (d->x3_pairs ? 32 : 0);
  auto launch = [&](auto kern, int threads = PT_THREADS) -> int {
    // the attribute is per (kernel, device), not per launch: set once to the most any launch can ask for (common.h)
    if (sm_lds_optin((const void*)kern, 160 * 1024) != hipSuccess) return SM_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(threads), lds, s, a);
    return SM_OK;
  };
  int lrc = SM_ERR_UNSUPPORTED;
  if (bco != PT_BCO) {
    if (nb1 != 0) return SM_ERR_BAD_SHAPE;
    lrc = bco == 32 ? launch(conv3x3_patch_n32_kernel<0>) : launch(conv3x3_patch_n128_kernel<0>);
    if (lrc != SM_OK) return lrc;
    SM_LAUNCH_CHECK();
    return SM_OK;
  }
#define PT_CASE(V)                                                                                                  \
  case V:                                                                                                           \
    lrc = ps.small == 128 ? launch(conv3x3_patch_kernel<128, V>) : launch(conv3x3_patch_kernel<192, V>);            \
    break;
#ifdef SM_EXPERIMENTS
  if ((d->flags & SM_CONV_DBG_PATCH_W4) && nb1 == 0 && (var == 0 || var == 1)) {
    lrc = var == 0 ? launch(conv3x3_patch_kernel_w4<0>, 256) : launch(conv3x3_patch_kernel_w4<1>, 256);
  } else
#endif
  switch (var) {
    PT_CASE(0)
#ifdef SM_OPERAND_F16
    PT_CASE(32)
#endif
#ifdef SM_EXPERIMENTS
    PT_CASE(1) PT_CASE(2) PT_CASE(3) PT_CASE(4) PT_CASE(8) PT_CASE(16) PT_CASE(20) PT_CASE(24)
#endif
    default: break;
  }
#undef PT_CASE
  if (lrc != SM_OK) return lrc;
  SM_LAUNCH_CHECK();
  return SM_OK;
}
