// A/B variants and ablations of rounds 1-2 that did NOT become part of the product (DESIGN.md section 6 has the numbers).
// The default build defines every flag as 0, which folds the code behind them away and leaves their kernel
// instantiations out of libsipmask_hip.so; `make EXPERIMENTS=1` brings them back for tools/ab_conv_variants.sh,
// tools/patch_variants_bench.py and friends.  None of this is declared in include/sipmask_hip.h.
#pragma once
// The product library reads NO environment variable: what a caller may choose travels in descriptors and
// arguments.  The experiment build's ablation masks are the exception, and they go through this one helper.
#ifdef SM_EXPERIMENTS
#include <cstdlib>
static inline int sm_experiment_env(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}
#endif
#ifdef SM_EXPERIMENTS
#define SM_CONV_DBG_LINEAR_TILES 0x40000000u   // disable the XCD-aware tile remap (measured neutral)
#define SM_CONV_DBG_REG_STAGING 0x20000000u    // register-staged loader instead of LDS-DMA
#define SM_CONV_DBG_WIDE_POS 0x00800000u       // 128-cout x 256-position tiles (64x128 per wave)
#define SM_CONV_DBG_FLAT_LOOP 0x00200000u      // flat LDS-DMA loader + peeled K loop WITHOUT the pipelined fragment reads
#define SM_CONV_DBG_LEGACY_LOOP 0x00100000u    // the original K loop (branchy loader, one fragment register set)
#define SM_CONV_DBG_K32_OPT 0x00080000u        // flat loader + peeled loop + pipelined fragment reads in the 32-wide-K kernel
#define SM_CONV_DBG_DEFORM_128 0x00008000u     // deformable conv on the 128 x 128 4-wave tile (default: 256 x 128 on 8 waves)
#define SM_CONV_DBG_PATCH_SMALL128 0x00002000u // sm_conv3x3_patch: only 128-position tiles finish a launch
#define SM_CONV_DBG_PATCH_SMALL192 0x00001000u // sm_conv3x3_patch: only 192-position tiles finish a launch
#define SM_CONV_DBG_PATCH_PIPE 0x00000800u     // sm_conv3x3_patch: fragment reads of sub-step i+1 pinned under the MFMAs of i
#define SM_CONV_DBG_PATCH_STAGGER 0x00000400u  // sm_conv3x3_patch: waves 4-7 issue their LDS-DMA between the two taps of a stage
#define SM_CONV_DBG_PATCH_NO_DMA 0x00000200u   // ABLATION (wrong results): no LDS-DMA in the main loop
#define SM_CONV_DBG_PATCH_NO_MFMA 0x00000100u  // ABLATION (wrong results): no MFMA / fragment reads in the main loop
#define SM_CONV_DBG_PATCH_PINGPONG 0x00000080u // sm_conv3x3_patch: ping-pong schedule of two wave groups
#define SM_CONV_DBG_WARP_SPEC 0x02000000u      // 8-wave producer/consumer variant of the 64-wide-K kernel
#define SM_CONV_DBG_PATCH_W4 0x10000000u       // sm_conv3x3_patch (uniform launches): 4 waves x (128 couts x 128 positions), accumulators in AGPRs
#define SM_CONV_DBG_K32_POS64 0x00000800u       // sm_conv2d, 32-wide K steps: 128-cout tiles start at 64 positions (bit shared with a patch-kernel flag)
#define SM_CONV_DBG_DX3_NO_BLEND 0x00000400u   // ABLATION (wrong results), sm_deform_conv2d_x3: constant operand (with PATCH_NO_DMA / PATCH_NO_MFMA: the other two)
#else
#define SM_CONV_DBG_LINEAR_TILES 0u
#define SM_CONV_DBG_REG_STAGING 0u
#define SM_CONV_DBG_WIDE_POS 0u
#define SM_CONV_DBG_FLAT_LOOP 0u
#define SM_CONV_DBG_LEGACY_LOOP 0u
#define SM_CONV_DBG_K32_OPT 0u
#define SM_CONV_DBG_DEFORM_128 0u
#define SM_CONV_DBG_PATCH_SMALL128 0u
#define SM_CONV_DBG_PATCH_SMALL192 0u
#define SM_CONV_DBG_PATCH_PIPE 0u
#define SM_CONV_DBG_PATCH_STAGGER 0u
#define SM_CONV_DBG_PATCH_NO_DMA 0u
#define SM_CONV_DBG_PATCH_NO_MFMA 0u
#define SM_CONV_DBG_PATCH_PINGPONG 0u
#define SM_CONV_DBG_WARP_SPEC 0u
#define SM_CONV_DBG_PATCH_W4 0u
#define SM_CONV_DBG_K32_POS64 0u
#define SM_CONV_DBG_DX3_NO_BLEND 0u
#endif
