/*
 * sipmask_hip.h -- C ABI of libsipmask_hip.so: the MI355X (gfx950) SipMask hot path.
 *
 * This header is the drop-in boundary: every entry point
 * replaces one reference pybind11/ATen seam, takes raw DEVICE pointers + sizes +
 * an explicit stream, allocates nothing, never synchronises, keeps no global
 * state, and returns an int status (0 = ok, <0 = sm_status).  Citations are into
 * /root/reference (M/ = SipMask-mmdetection/, B/ = SipMask-benchmark/).
 *
 * Data layout in HBM:
 *   activations  "pyramid tensors": rows = sum_l B*H_l*W_l, each row C channels,
 *                bf16 (or f32 where stated), NHWC inside a level, levels
 *                concatenated (level l starts at row row0[l]).
 *   conv weights [cout_pad][Kp] bf16, K = (kh,kw,cin) with cin fastest, Kp = K
 *                rounded up to 64, rows zero padded to the cout tile.
 */
#ifndef SIPMASK_HIP_H
#define SIPMASK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SM_ABI_VERSION 1
#define SM_MAX_LEVELS 5

typedef void* sm_stream_t; /* hipStream_t */

typedef enum {
  SM_OK = 0,
  SM_ERR_BAD_SHAPE = -1,
  SM_ERR_BAD_ARG = -2,
  SM_ERR_LAUNCH = -3,
  SM_ERR_UNSUPPORTED = -4,
  SM_ERR_WORKSPACE = -5
} sm_status;

/* conv flags */
#define SM_CONV_RELU 1u          /* y = max(y, 0) */
#define SM_CONV_OUT_F32 2u       /* y stored as float32 instead of bf16 */
#define SM_CONV_RES_ADD 4u       /* y += residual[row] (bf16, same rows/cstride as y) */
#define SM_CONV_RES_NEAREST 8u   /* y += residual at nearest-neighbour source (FPN top-down,
                                    M/mmdet/models/necks/fpn.py:149-152) */
#define SM_CONV_IN_RELU 16u      /* x = max(x,0) applied on load (fpn.py:174-175 P7) */
#define SM_CONV_BWD_GX_BF16 64u   /* sm_conv2d_bwd only: grad_x is written as bf16 rows (stride-1 convs: the dX GEMM's own
                                    output type; the training graph on row tensors hands it straight to the next op) */
#define SM_CONV_BWD_DX_SCATTER 1024u /* sm_deform_conv2d_bwd only, A/B switch: d(x) of FeatureAlign's shape by the atomic scatter alone
                                      (default, round 5: samples within 3 pixels of their tap by the gather kernel, plain stores in a
                                      fixed order; the scatter adds the far ones) */
#define SM_CONV_BWD_WGRAD_GEMM 128u /* sm_conv2d_bwd only, A/B switch: weight gradient through the materialised im2col^T / gout^T
                                    GEMM (the round-1 path) instead of sm_wgrad_direct */
#define SM_CONV_BWD_WGRAD_DIRECT 256u /* sm_conv2d_bwd only, A/B switch: sm_wgrad_direct wherever it is supported (default: where
                                    sm_wgrad_direct_preferred says it is the faster path) */
#define SM_CONV_BWD_WGRAD_TILE128 512u /* sm_conv2d_bwd / sm_wgrad_direct only, A/B switch: always the 128 x 128 4-wave tile.  (The
                                    SM_CONV_BWD_* bits are read from backward descriptors only; the SM_CONV_DBG_PATCH_* bits that
                                    share their values are read by sm_conv3x3_patch only.) */
#define SM_CONV_RELU_NCH 32u     /* y = max(y, 0) on channels < scale_nch only (the maskrcnn-benchmark variant's
                                    relu(scale(bbox_pred)), SipMask-benchmark/.../sipmask/sipmask.py:155-157) */
/* Launch-plan selectors.  Every one of them picks a different kernel / tile / loop for the SAME arithmetic: results are
 * equal up to f32 accumulation order (TILE256 / HAND_PLACED / PATCH_UNIFORM / BIG_TILES / LDS_EPILOGUE: bit-identical).
 * The rejected variants and ablations of rounds 1-2 (ping-pong, pipelined, staggered, legacy / flat loops, warp
 * specialisation, wide tiles, no-DMA / no-MFMA ablations ...) are NOT part of this interface: they are compiled only
 * with `make EXPERIMENTS=1` (sipmask_amd/csrc/experiments.h) for the A/B tools under tools/. */
#define SM_CONV_DBG_DEFORM_GATHER 0x80000000u /* deformable conv through conv_igemm's global-gather loader where the LDS-window
                                    kernel (deform_patch.hip: 3x3, 64 channels per deformable group) would run: the faster
                                    path when most learned offsets exceed ~3 pixels (sipmask_amd/engine.py measures it) */
#define SM_CONV_DBG_K32 0x10000000u          /* force 32-wide K steps, 4 blocks per CU */
#define SM_CONV_DBG_K64 0x08000000u          /* force 64-wide K steps, 2 blocks per CU */
#define SM_CONV_DBG_BIG_TILES 0x04000000u    /* never shrink tiles for occupancy (tests force large tiles at small shapes) */
#define SM_CONV_DBG_LDS_EPILOGUE 0x01000000u /* LDS-staged epilogue where the register epilogue would run (the path unaligned
                                    shapes take) */
#define SM_CONV_DBG_TILE256 0x00400000u      /* 256x256 tiles on 8 waves, 1 block per CU (64-wide K steps, cout_pad % 256 == 0) */
#define SM_CONV_DBG_HAND_PLACED 0x00040000u  /* with TILE256: hand-placed K step, LDS-DMA pieces between the MFMAs */
#define SM_CONV_DBG_PATCH_UNIFORM 0x00004000u  /* sm_conv3x3_patch: every tile 256 positions (no 128/192-position finishing tiles) */
#define SM_CONV_DBG_NO_SPLITK 0x00010000u   /* sm_conv2d_ws never splits K */
/* Operand type.  Default: bf16 operands (x, w), v_mfma_f32_32x32x16_bf16.  SM_CONV_F16: x and w hold IEEE binary16
 * values and the contraction runs on v_mfma_f32_32x32x16_f16 -- the operand type of the split-precision ("x3") head
 * plan, where every f32 value v travels as two halves hi = f16(v), lo = f16(v - hi) (22 mantissa bits together) laid
 * out as 3*C channels [hi | lo | hi] against weights [hi | hi | lo]: one ordinary convolution over 3*C channels then IS
 * the three-term product hi*hi + lo*hi + hi*lo with f32 accumulation (sm_split3_f16 writes that layout).  Needs
 * SM_CONV_OUT_F32, no residual / input ReLU / deformable gather; sm_conv2d, sm_conv2d_gn_stats, sm_conv3x3_patch. */
#define SM_CONV_F16 0x00020000u
/* With SM_CONV_F16 on sm_conv2d / sm_conv2d_ws (forward descriptors only; the bit is SM_CONV_BWD_GX_BF16 on backward ones):
 * y is written as the NEXT layer's split operand instead of f32 -- binary16 [rows][out_cstride] with out_cstride = 3 * ctot:
 * hi at out_coff + c, lo at ctot + out_coff + c, hi at 2 * ctot + out_coff + c (after bias / Scale / ReLU).  For convs
 * without a GroupNorm behind them (sip_mask_lat0, the SSD-style towers).  Needs the register epilogue's alignment. */
#define SM_CONV_OUT_X3 64u

/* One (multi-level) 2-D convolution as an implicit GEMM.  Replaces the ATen/cuDNN
 * conv calls under M/mmdet/models/backbones/resnet.py:206-229,
 * M/mmdet/models/necks/fpn.py:141-175 and
 * M/mmdet/models/anchor_heads/sipmask_head.py:253-285. */
typedef struct {
  int32_t nlev;                   /* 1..SM_MAX_LEVELS, all levels share the weights */
  int32_t batch;
  int32_t in_h[SM_MAX_LEVELS], in_w[SM_MAX_LEVELS];
  int32_t out_h[SM_MAX_LEVELS], out_w[SM_MAX_LEVELS];
  int64_t in_row0[SM_MAX_LEVELS];   /* first row of level l in x        */
  int64_t out_row0[SM_MAX_LEVELS];  /* first row of level l in y / residual (RES_ADD) */
  int32_t cin;                    /* multiple of 8 */
  int32_t cout;                   /* real output channels */
  int32_t cout_pad;               /* weight rows (multiple of the cout tile: 32, 64 or 128) */
  int32_t kh, kw, stride, pad, dil;
  int32_t in_cstride;             /* elements between consecutive rows of x   */
  int32_t out_cstride, out_coff;  /* y row stride / first channel (concat slices) */
  int32_t res_cstride;
  int32_t res_h[SM_MAX_LEVELS], res_w[SM_MAX_LEVELS]; /* RES_NEAREST source size */
  int64_t res_row0[SM_MAX_LEVELS];
  uint32_t flags;
  int32_t scale_nch;              /* channels < scale_nch are multiplied by level_scale */
  float level_scale[SM_MAX_LEVELS]; /* Scale(), sipmask_head.py:261: y=(acc+bias)*scale */
  int32_t deform_groups;          /* deform conv only: offset layout [row][G][kh*kw][2] f32 */
  int64_t w_batch_stride;         /* elements between the weight matrices of consecutive images; 0 = one shared
                                   * weight (every convolution).  != 0 turns the launch into `batch` independent
                                   * GEMMs (split-K weight gradients); needs out_h*out_w % position tile == 0 */
  /* group dimension (64-wide-K LDS-DMA convs only): `ngroups` > 1 runs that many problems of identical shape in ONE
   * launch -- e.g. the cls and reg tower convs of one depth (sipmask_head.py:252-257), which otherwise are two
   * 1404-block launches.  Group g reads x rows shifted by g*x_group_rows (0 = all groups share the input), writes y
   * rows (and reads a RES_ADD residual) shifted by g*y_group_rows, uses weights at w + g*w_group_stride elements,
   * bias at g*bias_group_stride floats and GroupNorm statistics at g*gn_group_stride floats.  0 or 1 = no groups. */
  int32_t ngroups;
  int64_t x_group_rows, y_group_rows, w_group_stride, bias_group_stride, gn_group_stride;
  float acc_scale;                /* 0 or 1: none.  Otherwise y = epilogue(acc * acc_scale ...): the accumulator is
                                   * multiplied BEFORE bias / Scale / ReLU.  The x3 plan stores weights multiplied by a
                                   * power of two (so that the low half of a ~1e-2 weight stays out of binary16's
                                   * subnormals) and passes the inverse here -- exact. */
  /* per-LEVEL weights (sm_conv3x3_patch only): != 0 -> level l uses the weight matrix at w + l*w_level_stride elements
   * and the bias at bias + l*bias_level_stride floats -- levels of one launch that do NOT share weights: the FPN's three
   * output 3x3 convs (fpn.py:154-157: fpn_convs[i] on the i-th merged lateral) as ONE launch.  0 = shared (towers). */
  int64_t w_level_stride, bias_level_stride;
  /* sm_conv3x3_patch: cout tile of the launch.  0 = by the weight padding (cout_pad == 32: the 32-cout tile, else 256-cout
   * tiles); 128 = 128-cout x 256-position tiles (cout_pad % 128 == 0) for convs whose position count cannot fill the chip
   * with 256-cout tiles -- the 3x3 convs of ResNet layer3 / layer4 (resnet.py:84-239: 4 200 / 1 050 positions per image). */
  int32_t patch_cout_tile;
  /* With SM_CONV_F16 (round 6): 1 = PAIRED split operands.  x rows hold C values as C/16 groups of [hi 16 | lo 16] binary16
   * (in_cstride = cin = 2 * C; sm_split_pairs_f16 / sm_groupnorm_apply_x3p / sm_upsample_sum2(out_x3 = 2) write that layout)
   * and the weights carry the same pairing along their input-channel axis (sm_conv3x3_patch: [cout_pad][C/16][9 taps][hi 16 |
   * lo 16]; sm_conv2d: [cout_pad][K = (kh, kw, 2 * C)]; sm_conv3x3_smallco: its fragment order over 2 * C channels); per 16
   * channels the kernels issue the three products w_hi*x_hi + w_hi*x_lo + w_lo*x_hi on fragments they read once -- the same
   * sum as the K-concatenated form ([hi | lo | hi] against [hi | hi | lo], x3_pairs = 0) with a third less operand traffic.
   * sm_conv3x3_patch (256-cout tiles), sm_conv3x3_smallco, sm_conv2d on its 32-wide-K kernel (K <= 1152); f32 output. */
  int32_t x3_pairs;
} sm_conv_desc;

int sm_version(void);
const char* sm_strerror(int status);
/* which cout tile (32/64/128) the library will use for this cout: weights must be
 * padded to a multiple of it */
int sm_conv_cout_tile(int cout);

/* What the library will launch for a conv descriptor: pure host logic, callable without a GPU (the `-m "not gpu"`
 * tests pin the selection rules with it; sm_conv2d & co. execute exactly this plan).  No reference counterpart: the
 * reference leaves algorithm choice to cuDNN. */
typedef struct sm_conv_plan {
  int32_t lds_dma;    /* 1 = LDS-DMA loader, 0 = register-staged loader (deformable gather, input ReLU) */
  int32_t k_step;     /* 32 or 64 bf16 per K step */
  int32_t k_padded;   /* K rounded up to 64 (the weight row pitch) */
  int32_t tile_cout;  /* block tile */
  int32_t tile_pos;
  int32_t threads;    /* 256, or 512 for the 8-wave tiles */
  int32_t k_loop;     /* 0 legacy loop, 1 flat loader + peeled loop, 3 = 1 + pipelined fragment reads */
  int32_t warp_spec;  /* producer/consumer A/B variant */
  int64_t blocks;     /* grid size (without split-K) */
  int32_t split_k;    /* K slices per tile sm_conv2d_ws would use given a workspace (1 = no split) */
  int32_t ring_stages; /* 32-wide K steps: 0 = two LDS stages and a draining barrier per step (always, in the default build);
                        * 3 / 4 = the experiments build's operand ring (csrc/experiments.h; measured neutral) */
  int64_t workspace_bytes; /* f32 partial slabs [split_k][rows][cout_pad] needed for that; 0 when split_k == 1 */
} sm_conv_plan;
int sm_conv_plan_query(const sm_conv_desc* d, int deformable, int with_gn_stats, sm_conv_plan* out);

/* y = epilogue(conv(x, w) + bias).  x bf16 rows, w bf16 [cout_pad][Kp], bias f32[cout]
 * or NULL, residual bf16 or NULL, y bf16/f32. */
int sm_conv2d(const sm_conv_desc* d, const void* x, const void* w, const float* bias,
              const void* residual, void* y, sm_stream_t stream);

/* sm_conv2d with a caller-provided workspace: launches that leave most of the chip idle (few tiles, long K loops: the
 * 3x3 / 1x1 convs of layer4, FPN lateral 2, ... at M = B*25*42) are split along K into sm_conv_plan.split_k slices per
 * tile (f32 partial slabs in the workspace) followed by a reduce + epilogue kernel.  workspace may be NULL / too small:
 * then this is sm_conv2d.  The summation order differs from sm_conv2d's by the slicing only (f32). */
int sm_conv2d_ws(const sm_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual, void* y,
                 void* workspace, int64_t workspace_bytes, sm_stream_t stream);

/* ResNet bottleneck tail as ONE launch (Bottleneck.forward, M/mmdet/models/backbones/resnet.py:167-200, caffe style,
   BN folded by the caller):  t2 = relu(conv3x3(x; w2) + b2);  y = relu(conv1x1(t2; w3) + b3 + identity);  and, when
   w1_next is given, the NEXT block's  t1_next = relu(conv1x1(y; w1_next) + b1_next)  -- t2 (and y as conv1's input)
   never leave the CU.  channels = C in {64, 128} (layer1 / layer2); x [B*h*w][C], identity / y [B*h*w][4C],
   t1_next [B*h*w][C], all bf16 NHWC rows; w2 [C][9C] with K = (kh, kw, cin), w3 [4C][C], w1_next [C][4C] bf16 (the
   sm_conv2d weight layout for these shapes); biases f32.  Arithmetic (K order, rounding points) is that of the
   separate sm_conv2d launches, so results are bit-identical to them. */
int sm_bottleneck_tail_supported(int channels);
int sm_bottleneck_tail(int batch, int h, int w, int channels, const void* x, const void* w2, const float* b2,
                       const void* w3, const float* b3, const void* identity, void* y, const void* w1_next,
                       const float* b1_next, void* t1_next, sm_stream_t stream);
/* The same with the block's 1x1 SHORTCUT conv fused in (resnet.py:453-469: the first block of a stage): layer1 (channels 64,
 * ds_channels 64, ds_stride 1) and layer2 (channels 128, ds_channels 256, ds_stride 2).  w3_ds = bf16 [4 * channels][channels +
 * ds_channels] = [w3 | w_downsample] row by row, b3_ds = b3 + b_downsample (both BN-folded), x_block = the block input rows
 * [batch * ds_h * ds_w][ds_channels]; output position (n, ho, wo) reads input (n, ds_stride * ho, ds_stride * wo).  The
 * shortcut is never materialised: it stays in the f32 accumulator until the one rounding of the block output.  w1_next /
 * b1_next / t1_next: as in sm_bottleneck_tail (NULL = no chained conv1; layer1 only). */
int sm_bottleneck_tail_ds(int batch, int h, int w, int channels, const void* x, const void* w2, const float* b2,
                          const void* w3_ds, const float* b3_ds, const void* x_block, int ds_channels, int ds_stride, int ds_h,
                          int ds_w, void* y, const void* w1_next, const float* b1_next, void* t1_next, sm_stream_t stream);

/* conv3 (1x1, channels -> 4 * channels, folded BN, + identity, ReLU) of one bottleneck chained with conv1 (1x1, 4 * channels ->
 * channels, folded BN, ReLU) of the NEXT one as one launch (resnet.py:188-200 then :175-178; channels == 256: ResNet layer3).
 * x = conv2's output rows [rows][channels], identity / y [rows][4 * channels], t1_next [rows][channels]; weights in the
 * sm_conv2d layout ([cout][cin]).  Bit-identical to the two sm_conv2d launches (same K order, same rounding points). */
int sm_conv1x1_pair(long long rows, int channels, const void* x, const void* w3, const float* b3, const void* identity, void* y,
                    const void* w1_next, const float* b1_next, void* t1_next, sm_stream_t stream);

/* 3x3 / stride 1 / pad 1 convolution with the input patch resident in LDS (csrc/conv3x3_patch.hip): the throughput
 * kernel for the large 3x3 layers (tower convs -- also as the grouped cls+reg launch --, fcos_cls + sip_cof, FPN
 * output convs).  Same descriptor, activations and epilogue semantics as sm_conv2d / sm_conv2d_gn_stats (bias,
 * per-level Scale on scale_nch channels, ReLU, bf16 / f32 output, fused GroupNorm statistics when gn_stats != NULL,
 * multi-level, group dimension; no residual), but the weights come in the kernel's own K order:
 *   w_patch bf16 [cout_pad][cin/32][9][32]  (32-channel chunk, tap kh*3+kw, channel), cout_pad a multiple of 256.
 * sm_conv3x3_patch_supported: 1 if the descriptor is eligible (3x3 s1 p1, cin % 64 == 0, cout_pad % 256 == 0, 8-aligned
 * output, level widths <= 253); sm_conv3x3_patch_tiles: its grid size.
 * Launch shape: one block per CU, so a launch of equal tiles pays for a whole last round however empty it is.  Every
 * (level, image) segment is therefore cut into 256-position tiles followed by 128- or 192-position tiles, chosen by a
 * list-scheduling estimate over the 256 CUs (sm_conv3x3_patch_plan: out = {big tiles, small tiles, small tile
 * positions, estimated makespan in 1/1000 of a 256-position tile time}); pure host logic, no GPU needed. */
int sm_conv3x3_patch_supported(const sm_conv_desc* d);
int64_t sm_conv3x3_patch_tiles(const sm_conv_desc* d);
int sm_conv3x3_patch_plan(const sm_conv_desc* d, int64_t* out4);
int sm_conv3x3_patch(const sm_conv_desc* d, const void* x, const void* w_patch, const float* bias, void* y,
                     int64_t* gn_stats, sm_stream_t stream);

/* Deformable conv v1 forward, bilinear gather fused into the GEMM operand load
 * (never materialises the column buffer).  Replaces deform_conv_forward_cuda,
 * M/mmdet/ops/dcn/src/deform_conv_cuda.cpp:152-260 + deformable_im2col_gpu_kernel,
 * M/mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:191-243.
 * offset: f32 [rows][G*kh*kw*2] with the reference channel order (kernel.cu:216-223).
 * Two kernels behind this entry point: FeatureAlign's shape (3x3, stride 1, pad 1, 64 channels per deformable group,
 * cout_pad % 256 == 0, plain epilogue) runs on csrc/deform_patch.hip -- one group's pixel window resident in LDS, offsets
 * beyond 3 pixels handled by a per-wave global fallback; every other shape, and SM_CONV_DBG_DEFORM_GATHER, on the
 * gather loader of csrc/conv_igemm.hip (any stride / dilation / kh x kw).  Same samples either way (f32 blend, one rounding
 * to the bf16 operand); the K summation order differs. */
int sm_deform_conv2d(const sm_conv_desc* d, const void* x, const float* offset, const void* w,
                     const float* bias, void* y, sm_stream_t stream);
/* Which of the two kernels takes d (pure host logic, no GPU needed): SM_OK and out4 = {blocks, tile rows, tile columns,
 * LDS window pixels per deformable group} for the LDS-window kernel, SM_ERR_UNSUPPORTED for the gather loader. */
int sm_deform_conv_window_plan(const sm_conv_desc* d, int64_t* out4);

/* Deformable conv v1 backward: replaces deform_conv_backward_input_cuda + deform_conv_backward_parameters_cuda
 * (M/mmdet/ops/dcn/src/deform_conv_cuda.cpp:262-490; kernels deform_conv_cuda_kernel.cu:279-433).
 * d is the FORWARD descriptor (x geometry in in_*, gout/offset geometry in out_*, out_cstride = gout row stride).
 * x bf16 rows, offset f32 [rows][G*kh*kw*2], gout bf16 rows [.][cout];
 * w_t: the weight as the operand of the grad-column GEMM, bf16 [Kpad][cout] with K = (kh,kw,cin) -- i.e. the
 *      tensor weight.permute(2,3,1,0).reshape(K, cout, 1, 1) laid out like a 1x1 sm_conv2d weight (needed for
 *      grad_x / grad_offset only);
 * outputs (each nullable = skipped): grad_x f32 [in rows][cin] (zeroed by the call), grad_offset f32 like offset,
 * grad_w_t f32 [K][cout] = dW^T (overwritten; dW[co][c][i][j] = grad_w_t[(i*kw+j)*cin + c][co]).
 * Needs cin % 64 == 0, (cin/G) % 64 == 0, cout % 8 == 0; any stride / dilation / kh x kw of the descriptor (round 3; the
 * offset rows are OUTPUT rows).  The grad columns W^T gout are kept as bf16 rows in the workspace (like every other
 * gradient row of the training graph); 3x3 kernels take the wave-per-(position, 64 channels) scatter, others the generic one. */
int64_t sm_deform_conv2d_bwd_workspace(const sm_conv_desc* d);
int sm_deform_conv2d_bwd(const sm_conv_desc* d, const void* x, const float* offset, const void* w_t,
                         const void* gout, float* grad_x, float* grad_offset, float* grad_w_t, void* workspace,
                         sm_stream_t stream);

/* Backward of a plain convolution (training row a17; the ATen/cuDNN conv backward under resnet.py / fpn.py /
 * sipmask_head.py).  d is the FORWARD descriptor, x / gout bf16 rows as in sm_deform_conv2d_bwd.
 *   grad_w_t  f32 [K][cout] = dW^T, K = (kh,kw,cin)                       (im2col^T + gout^T + MFMA GEMM)
 *   grad_bias f32 [cout]
 *   grad_x    f32 [in rows][cin]: with w_dgrad (stride 1) one forward implicit GEMM over gout --
 *             w_dgrad = the weight flipped in (kh,kw) and transposed to [cin][cout][kh][kw], laid out like an
 *             sm_conv2d weight; otherwise (strided convs) grad columns + col2im with w_t as in
 *             sm_deform_conv2d_bwd (needs cin % 64 == 0).
 * Every output is nullable.  Workspace: sm_deform_conv2d_bwd_workspace(d). */
int sm_conv2d_bwd(const sm_conv_desc* d, const void* x, const void* w_t, const void* w_dgrad, const void* gout,
                  float* grad_x, float* grad_w_t, float* grad_bias, void* workspace, sm_stream_t stream);

/* Weight gradient of a plain convolution straight from the NHWC rows (csrc/wgrad_direct.hip): d is the FORWARD descriptor,
 * x / gout bf16 rows as in sm_conv2d_bwd, grad_w_t f32 [kh*kw*cin][cout] (zeroed and accumulated by the call: split-K over
 * position slices with float atomics, so the last bits depend on arrival order).  Needs channels % 8 == 0.  sm_conv2d_bwd
 * uses it where sm_wgrad_direct_preferred(d) (long position axes; measured crossover) unless SM_CONV_BWD_WGRAD_GEMM /
 * SM_CONV_BWD_WGRAD_DIRECT force a path. */
int sm_wgrad_direct_supported(const sm_conv_desc* d);
int sm_wgrad_direct_preferred(const sm_conv_desc* d);
int sm_wgrad_direct(const sm_conv_desc* d, const void* x, const void* gout, float* grad_w_t, sm_stream_t stream);

/* ---- training graph on NHWC bf16 row tensors (csrc/train_rows.hip; host side sipmask_amd/ops_rows.py).  These replace
 * the chains of ATen launches (permute / contiguous / to / zeros / mul / threshold_backward / native_group_norm_backward
 * / upsample_bilinear2d_backward ...) autograd runs between the reference's conv layers in training mode
 * (M/mmdet/models/backbones/resnet.py:205-239, necks/fpn.py:137-178, anchor_heads/sipmask_head.py:241-287).
 *
 * sm_weight_prep: f32 OIHW parameter (x scale[cout] when given: the frozen-BatchNorm fold) -> bf16 [rows_pad][kp]:
 *   mode 0  sm_conv2d weight            rows = cout, k = (r*kw+s)*cin_pad + c
 *   mode 1  sm_conv2d_bwd w_dgrad       rows = cin,  k = (r*kw+s)*cout + o, taps flipped
 *   mode 2  sm_conv2d_bwd w_t           rows = (r*kw+s)*cin + c, k = o
 * sm_wgrad_finish: grad_w_t f32 [K][cout] (sm_conv2d_bwd) -> OIHW f32, x scale[cout] when given.
 * sm_relu_bwd_bf16: out = y > 0 ? g : 0 (n elements, n % 8 == 0).
 * sm_bias_grad_rows: out f32[channels] = column sums of g bf16 [rows][cstride] (channels % 8 == 0, <= 256).
 * sm_gn_bwd_rows: GroupNorm(+ReLU) backward.  x = the normalised tensor's INPUT (bf16 pyramid rows, geometry as
 *   sm_groupnorm), stats = the forward's (sum, sum of squares) per (image, level, group); dy bf16; outputs dx bf16,
 *   dgamma / dbeta f32[channels]; bins f32 [batch][nlev][groups][2] is scratch (zeroed by the call).
 * sm_upsample_bilinear_bwd_rows: adjoint of sm_upsample_bilinear: gout bf16 rows of the upsampled grid (row stride
 *   out_cstride, first channel out_coff -> a slice of a concatenated gradient), gin bf16 [batch*h*w][c].
 * sm_nearest_bwd_rows: adjoint of the SM_CONV_RES_NEAREST residual: g_coarse[src(p)] += g_fine[p].
 * sm_scatter_stride_rows: out [batch*h*w][c] = in [batch*out_h*out_w][c] placed at (y*stride, x*stride), zero elsewhere
 *   (dX of a strided 1x1 convolution after the channel GEMM). */
int sm_weight_prep(const float* w, const float* scale, int cout, int cin, int kh, int kw, int mode, void* out,
                   int rows_pad, int kp, int cin_pad, sm_stream_t stream);
/* sm_weight_prep for many weights in ONE launch (a training step re-lays out ~130 operands).  items: device array of
 *   struct { const float* w; const float* scale; uint16_t* out; int32 co, ci, kh, kw, mode, kp, cin_pad, tc, tiles_x, pad; }
 * (tc = channels per tile: 32 for kh*kw <= 9, else 8; tiles_x = ceil(ci / tc)); blocks: device int32 pairs (item, tile) with
 * tile < tiles_x * ceil(co / 32); lds_bytes >= 4 * 32 * (tc*kh*kw + 1) of the largest item.  The outputs' padding is NOT
 * touched: allocate them zeroed once (the host side keeps them across steps). */
int sm_weight_prep_multi(const void* items, const int32_t* blocks, int nblocks, int lds_bytes, sm_stream_t stream);
int sm_wgrad_finish(const float* grad_w_t, const float* scale, int cout, int cin, int kh, int kw, float* out,
                    sm_stream_t stream);
int sm_relu_bwd_bf16(const void* g, const void* y, void* out, int64_t n, sm_stream_t stream);
int sm_bias_grad_rows(const void* g, int64_t rows, int cstride, int channels, float* out, sm_stream_t stream);
int sm_gn_bwd_rows(const void* x, const void* dy, const float* gamma, const float* beta, const int64_t* stats, int batch,
                   int nlev, const int32_t* hw, const int64_t* row0, int channels, int groups, float eps, int relu,
                   void* dx, float* dgamma, float* dbeta, float* bins, sm_stream_t stream);
int sm_upsample_bilinear_bwd_rows(const void* gout, int out_cstride, int out_coff, int batch, int h, int w, int c,
                                  int factor, void* gin, sm_stream_t stream);
int sm_nearest_bwd_rows(const void* g_fine, int batch, int fine_h, int fine_w, int coarse_h, int coarse_w, int c,
                        void* g_coarse, sm_stream_t stream);
int sm_scatter_stride_rows(const void* in, int batch, int h, int w, int out_h, int out_w, int stride, int c, void* out,
                           sm_stream_t stream);

/* sm_conv2d / sm_deform_conv2d (offset != NULL) with the GroupNorm statistics of the output fused in the
 * epilogue: gn_stats int64 [batch][nlev][cout/8][2] = (sum, sum of squares) per (image, level, group of 8
 * channels), zeroed by the call.  Feed it to sm_groupnorm_apply.  Needs cout % 8 == 0.
 * GroupNorm statistics format (every gn_stats / stats argument of the bf16 row kernels): 64-bit FIXED POINT in units
 * of 2^-24.  Tiles contribute partial sums in arrival order; integer addition is associative, so the statistics -- and
 * every result downstream of them, NMS keep indices included -- are bit-reproducible from run to run, as the
 * reference's ATen GroupNorm is (M/mmdet/ops/norm.py:12-55).  Range: |sum| < 5.5e11 per (image, level, group). */
int sm_conv2d_gn_stats(const sm_conv_desc* d, const void* x, const float* offset, const void* w,
                       const float* bias, const void* residual, void* y, int64_t* gn_stats,
                       sm_stream_t stream);

/* offset = W_off (72x4) . (level_scale * reg[row][0:4]) -- FeatureAlign.conv_offset (level_scale
 * NULL = 1 when reg already carries Scale),
 * sipmask_head.py:30-33,50.  reg: f32 rows with stride reg_cstride, out f32 [rows][nout]. */
int sm_offset_linear(const float* reg, int reg_cstride, const float* w_off, int nout,
                     const int64_t* row0, const int32_t* rows_per_level, const float* level_scale,
                     int nlev, float* out, sm_stream_t stream);

/* Weight gradient of FeatureAlign.conv_offset (the 1x1 conv 4 -> nout, no bias, of the DETACHED box prediction:
 * sipmask_head.py:30-33,50 -- autograd reaches only its weight): grad_w[o][c] = sum_rows grad_out[row][o] * reg[row][c]
 * (reg already carries Scale).  Replaces the ATen matmul backward (a vendor GEMM, K = all positions).  Deterministic
 * (fixed-order two-pass reduction); workspace of sm_offset_linear_bwd_workspace(rows, nout) bytes. */
int64_t sm_offset_linear_bwd_workspace(int64_t rows, int nout);
int sm_offset_linear_bwd(const float* reg, int reg_cstride, const float* grad_out, int nout, int64_t rows,
                         float* workspace, float* grad_w, sm_stream_t stream);

/* y = relu(x) over n bf16 elements (n % 8 == 0): halves with the sign bit set become +0 (so a negative NaN becomes 0
   where torch.relu would keep it; activations on this path are finite).  Replaces the `F.relu(outs[-1])` in front of
   the P7 conv (M/mmdet/models/necks/fpn.py:166-170) so that conv takes the LDS-DMA operand path. */
int sm_relu_bf16(const void* x, void* y, int64_t n, sm_stream_t stream);
/* nseg <= SM_COPY_MAX_SEGS byte ranges copied by ONE launch: dst[i][0 .. bytes[i]) = src[i][...].  Meant for the per-step
 * results (sipmask_head.py:645-662 returns them per batch; M/mmdet/apis/test.py:12-72 collects them): the destinations are
 * pinned host buffers (hipHostMalloc: GPU-addressable), so the step's boxes / labels / counts / RLE strings reach the host
 * through ONE kernel launch on the step's stream instead of six hipMemcpyAsync calls.
 * src / dst / bytes are HOST arrays (read during the call). */
#define SM_COPY_MAX_SEGS 8
int sm_copy_segments(int nseg, const void* const* src, void* const* dst, const int64_t* bytes, sm_stream_t stream);

/* GroupNorm(groups) + optional ReLU over each (image, level), in place allowed.
 * nn.GroupNorm in M/mmdet/ops/conv_module.py:116-120 / sipmask_head.py:42,52.
 * stats: int64 fixed-point workspace [batch*nlev*groups*2] (see sm_conv2d_gn_stats), zeroed by the call. */
int sm_groupnorm(const void* x, void* y, const float* gamma, const float* beta, int64_t* stats,
                 int batch, int nlev, const int32_t* hw, const int64_t* row0, int channels,
                 int groups, float eps, int relu, sm_stream_t stream);

/* the normalisation half of sm_groupnorm, given precomputed statistics (same stats layout) */
int sm_groupnorm_apply(const void* x, void* y, const float* gamma, const float* beta, const int64_t* stats,
                       int batch, int nlev, const int32_t* hw, const int64_t* row0, int channels,
                       int groups, float eps, int relu, sm_stream_t stream);

/* ---- split-precision ("x3") head plan: the layout kernels behind SM_CONV_F16 (csrc/split_x3.hip).
 * The reference head computes in fp32 (sipmask_head.py:241-287,609-633).  The x3 plan keeps head activations f32 and
 * hands every convolution its operands as two binary16 halves per value, hi = f16(v), lo = f16(v - hi), along the
 * channel axis as [hi | lo | hi] (3*C channels; weights [hi | hi | lo], prepared by the host).
 * sm_split3_f16: x = f32 (x_is_f32) or bf16 rows [rows][in_cstride], first `channels` channels -> y binary16 rows of
 *   3*ctot channels: hi at [coff, coff+channels), lo at ctot + the same, hi again at 2*ctot + the same (ctot > channels
 *   lets several sources fill one destination: the mask branch's concatenation, sipmask_head.py:266-275).
 * sm_gn_stats_f32_fix: GroupNorm statistics of f32 pyramid rows (geometry as sm_groupnorm) in the fixed-point format of
 *   sm_conv2d_gn_stats; zeroed by the call.
 * sm_groupnorm_apply_x3: y = [relu](GroupNorm(x)) from such statistics; x f32 rows [rows][channels]; writes y_f32 (f32
 *   rows, may alias x) and / or y_split (binary16 [rows][3*channels], [hi | lo | hi]); either may be NULL, not both. */
int sm_split3_f16(const void* x, int x_is_f32, int64_t rows, int channels, int in_cstride, void* y, int ctot, int coff,
                  sm_stream_t stream);
/* The two-term layout for bf16 sources (round 5): y binary16 [rows][2*ctot] = [hi | hi] of x's first `channels` channels.  A
 * bf16 value is a binary16 value (down to 2^-17: below that the low half is at most 2^-24), so against weights [w_hi | w_lo] an
 * ordinary SM_CONV_F16 convolution over 2*C channels is x_hi*w_hi + x_hi*w_lo -- the three-term product of sm_split3_f16
 * without the term that is zero.  The x3 plan's first tower convs read the bf16 FPN outputs this way (2/3 of the MFMA work). */
int sm_split2_f16(const void* x_bf16, int64_t rows, int channels, int in_cstride, void* y, int ctot, int coff,
                  sm_stream_t stream);
/* sm_upsample_bilinear (f32 rows [batch*h*w][in_cstride], first c channels, integer factor, align_corners=False) with
 * the result written as the split layout of sm_split3_f16 (y binary16 [batch*h*factor*w*factor][3*ctot], slice coff). */
int sm_upsample_bilinear_x3(const float* x, void* y, int batch, int h, int w, int c, int factor, int in_cstride, int ctot,
                            int coff, sm_stream_t stream);
/* out = [relu](a0 + up2(a1) + up4(a2)) on the fine grid [batch][h0][w0][c] (bilinear, align_corners=False; h0, w0 multiples of
 * 4; a1 on h0/2 x w0/2, a2 on h0/4 x w0/4, all rows of c channels).  sip_mask_lat0 by linearity: the 1x1 conv over
 * [l0 | up2(l1) | up4(l2)] (sipmask_head.py:275-283) = W0.l0 + up2(W1.l1) + up4(W2.l2), so the three products run at their own
 * resolutions and this adds the coarse ones.  is_f32 = 0: a1, a2, out bf16, a0 NULL (out is the RES_ADD residual of the l0
 * conv); is_f32 = 1: a0 (or NULL), a1, a2 f32, out f32 rows or -- out_x3 = 1 -- the split layout of sm_split3_f16 (3*c channels)
 * or -- out_x3 = 2, round 6 -- the paired layout of sm_split_pairs_f16 (2*c per row, c % 16 == 0). */
int sm_upsample_sum2(const float* a0, const void* a1, const void* a2, int is_f32, int batch, int h0, int w0, int c, int relu,
                     int out_x3, void* out, sm_stream_t stream);
int sm_gn_stats_f32_fix(const float* x, int64_t* stats, int batch, int nlev, const int32_t* hw, const int64_t* row0,
                        int channels, int groups, sm_stream_t stream);
int sm_groupnorm_apply_x3(const float* x, const float* gamma, const float* beta, const int64_t* stats, int batch, int nlev,
                          const int32_t* hw, const int64_t* row0, int channels, int groups, float eps, int relu,
                          float* y_f32, void* y_split, sm_stream_t stream);
/* The PAIRED layout (round 6; sm_conv_desc.x3_pairs): a row of C values as C/16 groups of [hi 16 | lo 16] binary16 -- 2*C per
 * row (4 bytes per value instead of the 6 of [hi | lo | hi]); channel c has its hi half at (c >> 4) * 32 + (c & 15) and its lo
 * half 16 elements behind.  sm_split_pairs_f16 is sm_split3_f16 writing that layout (ctot % 16 == 0, y binary16
 * [rows][2*ctot]); sm_groupnorm_apply_x3p is sm_groupnorm_apply_x3 writing it (channels % 16 == 0; y_pairs required, y_f32
 * optional).  The tower convs of the x3 head plan (sipmask_head.py:252-258) and fcos_cls + sip_cof read these rows. */
int sm_split_pairs_f16(const void* x, int x_is_f32, int64_t rows, int channels, int in_cstride, void* y, int ctot, int coff,
                       sm_stream_t stream);
int sm_groupnorm_apply_x3p(const float* x, const float* gamma, const float* beta, const int64_t* stats, int batch, int nlev,
                           const int32_t* hw, const int64_t* row0, int channels, int groups, float eps, int relu,
                           float* y_f32, void* y_pairs, sm_stream_t stream);

/* 3x3 / stride 1 / pad 1 convolution with 8..32 output channels (cout % 8 == 0) over cin % 32 == 0 input channels, bf16
 * operands: SipMaskHead's sip_mask_lat (512 -> 32, sipmask_head.py:284) and fcos_reg + fcos_centerness (256 -> 4 + 1,
 * :261-268).  Same descriptor and epilogue as sm_conv2d (bias, Scale() on the first scale_nch channels, SM_CONV_RELU /
 * SM_CONV_RELU_NCH / SM_CONV_OUT_F32, acc_scale; no residual, no groups), own kernel: one wave per 2 x 32-position tile, the
 * weights in MFMA-fragment order w_frag = bf16 [cin / 32][9 taps][2][64 lanes][8] (lane = 32 * khalf + cout row, rows >= cout
 * zero; channel = 32 * slice + 16 * half + 8 * khalf + e) read straight from L2 (csrc/conv3x3_smallco.hip). */
int sm_conv3x3_smallco_supported(const sm_conv_desc* d);
int sm_conv3x3_smallco(const sm_conv_desc* d, const void* x, const void* w_frag, const float* bias, void* y,
                       sm_stream_t stream);

/* The ResNet stem as one launch (resnet.py:497-505: conv1 7x7 / stride 2 / pad 3 -> norm1 (frozen, folded into w / bias by the
 * caller) -> ReLU -> maxpool 3x3 / stride 2 / pad 1): img NCHW f32 [batch][3][h][w] -> y NHWC bf16 rows
 * [batch * h2 * w2][64] with h1 = (h - 1) / 2 + 1, h2 = (h1 - 1) / 2 + 1 (same for w).  w_stem: bf16 [64][7][8][4] =
 * (cout, kh, kw, cin) with zeros in the kw = 7 and cin = 3 slots; bias f32 [64].  Same rounding points as
 * sm_nchw_f32_to_nhwc_bf16 + sm_conv2d(SM_CONV_RELU) + sm_maxpool3x3s2; the f32 sum runs in a different K order. */
int sm_stem_fused(const float* img, const void* w_stem, const float* bias, void* y, int batch, int h, int w,
                  sm_stream_t stream);

/* 3x3 stride-2 pad-1 max pool (resnet.py:460), NHWC bf16. */
int sm_maxpool3x3s2(const void* x, void* y, int batch, int h, int w, int c, sm_stream_t stream);

/* NCHW f32 image -> NHWC bf16 with channels zero padded to cpad. */
int sm_nchw_f32_to_nhwc_bf16(const float* x, void* y, int batch, int c, int h, int w, int cpad,
                             sm_stream_t stream);

/* Bilinear (align_corners=False) integer-factor upsample of NHWC rows, written into a channel
 * slice of y (F.interpolate at sipmask_head.py:279,285).  in: bf16 or f32; out same type. */
int sm_upsample_bilinear(const void* x, void* y, int batch, int h, int w, int c, int factor,
                         int in_cstride, int out_cstride, int out_coff, int is_f32,
                         sm_stream_t stream);

/* ---- exact-f32 plan: the parity mode of the hot path -------------------------------------------------------
 * The reference computes the whole path in fp32 (M/mmdet/models/anchor_heads/sipmask_head.py:241-287,609-633;
 * resnet.py:206-229; fpn.py:141-175).  These entry points run the same launch plan with f32 activations and
 * weights on v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulate) so that box / mask logits can be held
 * to the reference within accumulation-order rounding; the bf16 entry points above stay the throughput path.
 * Layout as above with f32 elements: activations f32 rows (cin, in_cstride multiples of 4), weights f32
 * [cout_pad][Kp], K = (kh,kw,cin) cin fastest, Kp = K rounded up to 16, rows padded to sm_conv_cout_tile(cout).
 * sm_conv2d_f32: offset != NULL selects the deformable variant (bilinear samples formed in f32 exactly as
 * deform_conv_cuda_kernel.cu:85-115); residual f32 rows; y f32 (SM_CONV_OUT_F32 is implied).
 * With SM_CONV_F16 in d->flags the SAME tensors are contracted in split precision: the loader turns every f32 operand
 * element (the blended deformable sample included) into binary16 halves hi + lo and a product is three
 * v_mfma_f32_32x32x16_f16 terms (w_hi*x_hi + w_hi*x_lo + w_lo*x_hi, ~2^-21) -- FeatureAlign's deformable conv in the x3
 * head plan (5x less matrix-pipe time than the exact kernel; cout tile 128 only; weights pre-multiplied by a power
 * of two with d->acc_scale its inverse). */
int sm_conv2d_f32(const sm_conv_desc* d, const float* x, const float* offset, const float* w, const float* bias,
                  const float* residual, float* y, sm_stream_t stream);
/* FeatureAlign's deformable conv of the x3 head plan on the LDS-window kernel (csrc/deform_patch_x3.hip, round 5): the
 * computation of sm_conv2d_f32 + SM_CONV_F16 with offset != NULL (deform_conv_forward_cuda on fp32 tensors,
 * M/mmdet/ops/dcn/src/deform_conv_cuda.cpp:152-260, deform_conv_cuda_kernel.cu:85-115,191-243, called by
 * M/mmdet/models/anchor_heads/sipmask_head.py:21-55) for its shape -- 3x3 / stride 1 / pad 1, 64 channels per deformable
 * group, cout_pad % 256 == 0 -- with the f32 window of half a deformable group resident in LDS instead of four L2 corner
 * loads per sample.  x: f32 rows; offset: f32 [rows][G*18] in the reference channel order; y: f32 rows;
 * w_split: binary16 [cout_pad][G*2*9][hi 32 | lo 32] -- per cout row and K step (group g, channel half c, tap t) the 32
 * channels g*64 + c*32 .. +31 of w[cout][.][t] * s as hi = f16(v), lo = f16(v - hi), s a power of two with
 * d->acc_scale = 1 / s; epilogue acc * acc_scale + bias, SM_CONV_RELU.  gn_stats != NULL: the output's GroupNorm
 * statistics [batch][nlev][cout/8][2] in the fixed-point format of sm_conv2d_gn_stats, zeroed by the call.
 * Samples farther than 3 pixels from their tap leave the window: that wave gathers the tap from global memory (any offset
 * is handled; a model whose offsets are mostly that large is faster on sm_conv2d_f32).
 * sm_deform_conv2d_x3_plan (host logic only): out4 = {blocks, 8x32 row tiles, 32x8 column tiles, window pixels}. */
int sm_deform_conv2d_x3_supported(const sm_conv_desc* d);
int sm_deform_conv2d_x3_plan(const sm_conv_desc* d, int64_t* out4);
int sm_deform_conv2d_x3(const sm_conv_desc* d, const float* x, const float* offset, const void* w_split, const float* bias,
                        float* y, int64_t* gn_stats, sm_stream_t stream);
/* NCHW f32 image -> NHWC f32 with channels zero padded to cpad (multiple of 4). */
int sm_nchw_f32_to_nhwc_f32(const float* x, float* y, int batch, int c, int h, int w, int cpad, sm_stream_t stream);
/* 3x3 stride-2 pad-1 max pool on NHWC f32 (resnet.py:460). */
int sm_maxpool3x3s2_f32(const float* x, float* y, int batch, int h, int w, int c, sm_stream_t stream);
/* sm_groupnorm on f32 rows; stats: DOUBLE workspace [batch*nlev*groups*2] (sum, sum of squares), zeroed by the call. */
int sm_groupnorm_f32(const float* x, float* y, const float* gamma, const float* beta, double* stats, int batch,
                     int nlev, const int32_t* hw, const int64_t* row0, int channels, int groups, float eps, int relu,
                     sm_stream_t stream);

/* ---- detection post-processing (sipmask_head.py:543-605) ------------------------------- */

typedef struct {
  int32_t batch, nlev, num_classes; /* num_classes = 80 (without background) */
  int32_t h[SM_MAX_LEVELS], w[SM_MAX_LEVELS], stride[SM_MAX_LEVELS];
  int64_t row0[SM_MAX_LEVELS];   /* level start row in the head output tensors */
  int32_t cls_cstride, cls_coff; /* class logits   f32 [rows][cls_cstride]  */
  int32_t cof_cstride, cof_coff; /* coefficients   f32 [rows][cof_cstride], 128 wide */
  int32_t reg_cstride;           /* f32 [rows][reg_cstride]: ch0-3 = bbox_pred (x Scale, NOT yet
                                    x stride: the kernel applies bbox_pred.float()*stride,
                                    sipmask_head.py:268), ch4 = centerness logit */
  int32_t nms_pre;               /* cfg.nms_pre */
  int32_t img_h, img_w;          /* img_shape for the clamp (transforms.py:219-223) */
  int32_t kmax;                  /* candidate capacity per image = sum_l min(nms_pre, h*w) */
  float scale_factor[4];         /* x1,y1,x2,y2 are divided by these when rescale != 0 (sipmask_head.py:587-588;
                                  * a keep_ratio=False pipeline gives [w,h,w,h] scales, a scalar is repeated) */
  int32_t rescale;
  int32_t reg_prescaled;         /* 1: reg ch0-3 already carry x stride (API path: bbox_preds as
                                    returned by SipMaskHead.forward) */
  const float* per_image;        /* NULL: img_h / img_w / scale_factor above hold for every image of the batch.  Otherwise a
                                  * DEVICE table f32 [batch][6] = (img_h, img_w, scale_factor[0..3]) read per image -- the
                                  * reference takes them from img_metas[img_id] (sipmask_head.py:517-541): a keep_ratio
                                  * resize gives every image of a batch its own. */
} sm_det_desc;

/* workspace bytes for sm_det_select */
int64_t sm_det_select_workspace(const sm_det_desc* d);
/* per level: max_c sigmoid(cls)*sigmoid(ctr) -> top-k (value desc, index asc) -> gather +
 * distance2bbox.  Outputs per image: boxes f32 [B][kmax][4], scores f32 [B][C][kmax] (CLASS-MAJOR)
 * (sigmoid, no background column; class-major so per-class NMS streams it), ctr f32 [B][kmax], cofs f32 [B][kmax][128],
 * cand_pos i32 [B][kmax] (position inside its level), ncand i32 [B]. */
int sm_det_select(const sm_det_desc* d, const float* cls, const float* reg, const float* cof,
                  float* boxes, float* scores, float* ctr, float* cofs, int32_t* cand_pos,
                  int32_t* ncand, void* workspace, sm_stream_t stream);

int64_t sm_multiclass_nms_workspace(int batch, int kmax, int num_classes);
/* Batched device-resident multiclass_nms_idx, M/mmdet/core/post_processing/bbox_nms.py:79-146
 * with the GPU NMS rule IoU(+1) > thr (M/mmdet/ops/nms/src/nms_kernel.cu:14-22,61).
 * Outputs padded to max_num: det f32 [B][max_num][5], labels i64 [B][max_num],
 * keep i64 [B][max_num] (candidate row), ndet i32 [B]. */
int sm_multiclass_nms(const float* boxes, const float* scores, const float* ctr, const int32_t* ncand,
                      int batch, int kmax, int num_classes, float score_thr, float iou_thr,
                      int max_num, float* det, int64_t* labels, int64_t* keep, int32_t* ndet,
                      void* workspace, sm_stream_t stream);

/* Batched SipMaskHead.fast_nms (ssd_flag configs and SipMask-VIS), sipmask_head.py:868-910: per class the
 * top_k boxes by score*centerness, IoU WITHOUT +1 (jaccard :912-960), keep j iff max_{i<j} IoU(i,j) <= iou_thr
 * and score > score_thr; the survivors of all classes sorted by score, top max_num.  Same inputs, outputs and
 * workspace (sm_multiclass_nms_workspace) as sm_multiclass_nms; scores are multiplied by ctr inside. */
int sm_fast_nms(const float* boxes, const float* scores, const float* ctr, const int32_t* ncand, int batch,
                int kmax, int num_classes, float score_thr, float iou_thr, int top_k, int max_num, float* det,
                int64_t* labels, int64_t* keep, int32_t* ndet, void* workspace, sm_stream_t stream);

/* Device-side COCO RLE of the assembled masks: replaces the per-detection D2H + paste + pycocotools
 * mask_util.encode loop of sipmask_head.py:645-657 (algorithm: cocoapi common/maskApi.c rleEncode + rleToString).
 * masks u8 0/1 [batch][max_num][ho][wo]; detection i of image b is encoded iff i < ndet[b].  The mask's
 * top-left min(mask, canvas) window is pasted on a zero canvas_h x canvas_w canvas (:648-653, RLE 'size').
 * rect (nullable) int32 [batch][max_num][4] = x0,y0,x1,y1 (exclusive) outside which the caller guarantees zeros
 * (CropSplit zeroes everything outside the box), so only the box is read.
 * Outputs: counts u32 [batch][max_num][max_runs] (uncompressed run lengths), nruns[d] (or -needed when
 * max_runs is too small), nchars[d], and all compressed strings packed back to back: detection d's 'counts'
 * bytes are packed[offsets[d] .. offsets[d+1]) (offsets has batch*max_num+1 entries; if offsets[last] exceeds
 * packed_cap nothing past the capacity was written and the caller must retry with a larger buffer). */
int64_t sm_rle_workspace(int batch, int max_num, int canvas_w, int max_runs);
/* rect hint for sm_rle_encode from the detections sm_mask_assemble used (same box_mul/box_div/up_scale):
 * a conservative output-pixel rectangle per detection outside which the assembled mask is zero. */
int sm_mask_rects(const float* det, int batch, int max_num, float box_mul_x, float box_mul_y, float box_div,
                  double up_scale_h, double up_scale_w, int32_t* rect, const float* per_image, sm_stream_t stream);
int sm_rle_encode(const uint8_t* masks, const int32_t* ndet, const int32_t* rect, int batch, int max_num, int ho,
                  int wo, int canvas_h, int canvas_w, int max_runs, uint32_t* counts, int32_t* nruns,
                  int32_t* nchars, uint8_t* packed, int64_t packed_cap, int64_t* offsets, void* workspace,
                  sm_stream_t stream);
/* ... with every IMAGE's own mask size and canvas (round 5): per_image int32 [batch][4] = (mask_h, mask_w, canvas_h,
 * canvas_w) on the device, or NULL (= sm_rle_encode).  A keep_ratio batch gives every image its own mask
 * (floor(Hm * 2 / scale_factor), sipmask_head.py:621-633) inside the batch's [ho][wo] planes and its own canvas (img_shape, or
 * ori_shape with rescale: sipmask_head.py:645-653); canvas_h / canvas_w are then the LARGEST canvas of the batch (they size
 * the workspace), the rect hint stays per detection.  One launch per batch instead of one per image. */
int sm_rle_encode_images(const uint8_t* masks, const int32_t* ndet, const int32_t* rect, const int32_t* per_image, int batch,
                         int max_num, int ho, int wo, int canvas_h, int canvas_w, int max_runs, uint32_t* counts,
                         int32_t* nruns, int32_t* nchars, uint8_t* packed, int64_t packed_cap, int64_t* offsets,
                         void* workspace, sm_stream_t stream);

/* Candidate selection of the maskrcnn-benchmark variant (B/ = SipMask-benchmark/,
 * B/fcos_core/modeling/rpn/sipmask/inference.py:66-138): per level, every (location, class) pair with
 * sigmoid(cls) > pre_nms_thresh competes with key sigmoid(cls)*sigmoid(ctr); the best min(count, d->nms_pre)
 * pairs of each level are decoded (clip_to_image) and gathered.  d->kmax = sum_l min(nms_pre, h_l*w_l*C);
 * level l owns the slots [cand0_l, cand0_l + nms_pre), unused slots stay empty (zero box, no score).
 * Outputs: boxes [B][kmax][4], scores [B][C][kmax] class-major DENSE with sqrt(key) at (class of the pair, slot)
 * and 0 elsewhere (so sm_multiclass_nms with score_thr 0 and ctr = 1 is boxlist_ml_nms: same-label NMS,
 * B/fcos_core/csrc/cuda/ml_nms.cu), cofs [B][kmax][128], lvl_cnt int32 [B][SM_MAX_LEVELS], ncand[b] = kmax. */
int64_t sm_pairs_select_workspace(const sm_det_desc* d);
int sm_pairs_select(const sm_det_desc* d, float pre_nms_thresh, const float* cls, const float* reg, const float* cof,
                    float* boxes, float* scores, float* cofs, int32_t* lvl_cnt, int32_t* ncand, void* workspace,
                    sm_stream_t stream);

/* Evaluation-workload injection (no reference counterpart; bench.py --det-boxes): ONE launch that, while *flag != 0,
 * overwrites x1, y1, x2, y2 of the n kept detections det [n][det_stride] (f32; the score in column 4 is kept) with set
 * ((*counter + 1) % nsets) of sets [nsets][n][4] and advances *counter.  Labels, kept indices and coefficients stay the
 * detector's: mask assembly and RLE then see an evaluation run's box sizes behind a random-weight detector.  counter and
 * flag are device memory, so the launch can sit inside a captured graph. */
int sm_det_boxes_override(float* det, int det_stride, const float* sets, int nsets, int n, int32_t* counter,
                          const uint8_t* flag, sm_stream_t stream);

/* Single-class greedy NMS with the reference op's contract, nms_cuda.nms:
 * dets f32 [n][5] -> keep i64 [<=n] ascending original indices, *nkeep (device).
 * M/mmdet/ops/nms/src/nms_kernel.cu:71-139.  workspace: sm_nms_workspace(n) bytes. */
int64_t sm_nms_workspace(int n);
int sm_nms(const float* dets, int n, float iou_thr, int64_t* keep, int32_t* nkeep, void* workspace,
           sm_stream_t stream);

/* Fused mask assembly: coef@basis -> sigmoid -> quadrant crop -> bilinear x up -> > thr.
 * sipmask_head.py:609-633 + CropSplit M/mmdet/ops/crop/src/crop_split_cuda_kernel.cu:19-59.
 * basis f32: [B][Hm][Wm][32] (basis_hwc=1) or [B][32][Hm][Wm] (basis_hwc=0);
 * cofs f32 [B][kmax][128] gathered through keep[B][max_num]; det f32 [B][max_num][5].
 * masks u8 [B][max_num][Ho][mask_pitch] (rows >= ndet[b] untouched; mask_pitch % 4 == 0, >= Wo; the pad
 * columns of a written dword are zeroed);
 * pos_masks f32 [B][max_num][Hm][Wm] (post sigmoid+crop, the CropSplit output) or NULL.
 * crop box = (det[:4] * box_mul_{x,y}) / box_div  (sipmask_head.py:623: * scale_factor / 2, per coordinate for
 * the [w,h,w,h] scale factors of keep_ratio=False pipelines);
 * up_scale_{h,w} = the F.interpolate scale_factor (:629-632; ssd_flag: 2 / scale_factor[3:1:-1]), Ho/Wo the
 * resulting size.
 * per_image (sm_mask_assemble, sm_mask_assemble_lo, sm_mask_rects): NULL = one geometry for the batch (the scalar
 * arguments).  Otherwise a DEVICE table f32 [batch][8] = (box_mul_x, box_mul_y, up_scale_h, up_scale_w, Ho, Wo, 0, 0) read
 * per image (get_bboxes_single runs with img_metas[img_id]['scale_factor'], sipmask_head.py:517-541,621-633); the
 * scalar Ho / Wo then give the CANVAS every mask plane is allocated with (>= every image's Ho / Wo), pixels of a plane
 * outside its image's Ho x Wo are not part of the result. */
int sm_mask_assemble(const float* basis, int basis_hwc, const float* cofs, const int64_t* keep,
                     const float* det, const int32_t* ndet, int batch, int kmax, int max_num,
                     int hm, int wm, int ho, int wo, int mask_pitch, float box_mul_x, float box_mul_y, float box_div,
                     double up_scale_h, double up_scale_w, float mask_thr, uint8_t* masks, float* pos_masks,
                     const float* per_image, sm_stream_t stream);

/* sm_mask_assemble for a launch plan that owns its buffers: the basis is given at its CONV resolution (basis_lo f32
 * [B][lo_h][lo_w][32], the relu(sip_mask_lat) output BEFORE the bilinear x`factor` of sipmask_head.py:285; Hm = lo_h *
 * factor) and interpolated after the coefficient dot product (bilinear interpolation is linear: equal to the reference
 * order up to f32 rounding), and only the 128x8-pixel tiles of each detection's box rectangle are written, plus zeros
 * over the tiles the same slot covered in the previous call.  `state` int32 [batch*max_num][4] carries those tile
 * ranges between calls: zero it together with `masks` when the buffer is created, never touch either in between.
 * After the call `masks` holds exactly what sm_mask_assemble would have written (zeros outside the rectangles).
 * The kernel's LDS tiles are sized by up_scale (dynamic shared memory); sm_mask_assemble_lo_supported (host logic) tells
 * whether a geometry fits: 2 * batch * max_num <= 4096 work-list entries and up_scale above ~0.45 (scale_factor <= ~4.4
 * in get_bboxes' 2 / scale_factor).  Unsupported geometries return SM_ERR_UNSUPPORTED: assemble from the upsampled
 * basis (sm_mask_assemble) instead, as sipmask_amd/engine.py does at plan-build time. */
int sm_mask_assemble_lo_supported(int batch, int max_num, int factor, double up_scale_h, double up_scale_w);
int64_t sm_mask_assemble_lo_workspace(int batch, int max_num);
int sm_mask_assemble_lo(const float* basis_lo, int lo_h, int lo_w, int factor, const float* cofs, const int64_t* keep,
                        const float* det, const int32_t* ndet, int batch, int kmax, int max_num, int ho, int wo,
                        int mask_pitch, float box_mul_x, float box_mul_y, float box_div, double up_scale_h,
                        double up_scale_w, float mask_thr, uint8_t* masks, int32_t* state, void* workspace,
                        const float* per_image, sm_stream_t stream);

/* SipMask++ rescoring tail (sipmask_head.py:638-641): mask_scores[b][i] = max over the hw positions of
 * feat[(b*max_num+i)*hw + p][labels[b][i]] (feat = relu(mask_scoring(convs_scoring(pos_masks))), f32 NHWC rows)
 * times det[b][i][4]; 0 for i >= ndet[b]. */
int sm_mask_rescore(const float* feat, const int64_t* labels, const float* det, const int32_t* ndet, int batch,
                    int max_num, int hw, int channels, float* mask_scores, sm_stream_t stream);

/* ---- training-side ops ---------------------------------------------------------------- */

/* crop_split_cuda_forward/backward, M/mmdet/ops/crop/src/crop_split_cuda.cpp:14-36:
 * data f32 [c*c][H][W][N], rois f32 [N][4], out f32 [H][W][N] (fully written). */
int sm_crop_split_fwd(const float* data, const float* rois, float* out, int h, int w, int c, int n,
                      sm_stream_t stream);
int sm_crop_split_bwd(const float* grad_out, const float* rois, float* grad_in, int h, int w, int c,
                      int n, sm_stream_t stream);
/* crop_split_gt_cuda_forward, M/mmdet/ops/crop/src/crop_split_gt_cuda_kernel.cu:19-49 */
int sm_crop_split_gt_fwd(const float* data, const float* rois, float* out, int h, int w, int n,
                         sm_stream_t stream);

/* sigmoid_focal_loss_cuda.forward/backward, M/mmdet/ops/sigmoid_focal_loss/src/
 * sigmoid_focal_loss_cuda.cu:24-97.  logits f32 [n][c], targets i64 [n]. */
int sm_sigmoid_focal_loss_fwd(const float* logits, const int64_t* targets, float* losses, int n, int c,
                              float gamma, float alpha, sm_stream_t stream);
int sm_sigmoid_focal_loss_bwd(const float* logits, const int64_t* targets, const float* d_losses,
                              float* d_logits, int n, int c, float gamma, float alpha,
                              sm_stream_t stream);

/* FCOS target assignment for a batch, SipMaskHead.fcos_target / fcos_target_single (sipmask_head.py:731-857), device
 * resident: points f32 [S][2] (all levels concatenated), point_stride / range_lo / range_hi f32 [S] (stride and
 * regress range of each point's level), gt_boxes f32 [B][gmax][4], gt_labels i64 [B][gmax], ngt i32 [B].
 * center_sampling / radius as in the head's config.  Outputs: labels i64 [B][S] (0 = background), bbox_targets f32
 * [B][S][4] = (l,t,r,b) against the chosen box, gt_index i32 [B][S] = index of that box (-1 when the image has none).
 * Bit-identical to the reference's broadcast tensor code (same f32 operations, first index on area ties). */
int sm_fcos_target(const float* points, const float* point_stride, const float* range_lo, const float* range_hi,
                   const float* gt_boxes, const int64_t* gt_labels, const int32_t* ngt, int batch, int npoints, int gmax,
                   int center_sampling, float radius, int64_t* labels, float* bbox_targets, int32_t* gt_index,
                   sm_stream_t stream);

/* Fused SipMask mask loss (sipmask_head.py:443-461): for detection n, the sum over the pixels of its crop box
 * of F.binary_cross_entropy(CropSplit(sigmoid(basis . cof_q))[.., n], CropSplitGt(gt[idx_gt[n]])[.., n]) --
 * without materialising the 4 x [Hm,Wm,N] probability volumes.  basis f32 [32][Hm][Wm] (basis_hwc=0) or
 * [Hm][Wm][32]; cof f32 [N][128]; boxes f32 [N][4] in basis-grid coordinates (the CropSplit rois);
 * gt u8 0/1 [G][Hm][Wm]; idx_gt int64 [N].  fwd: bce_sum f32 [N] (zeroed by the call: every box is split over 16
 * blocks that add their share with float atomics).
 * bwd: grad_sum f32 [N] (dL/d bce_sum) -> grad_cof f32 [N][128] (zeroed by the call, accumulated the same way),
 * grad_basis f32 (layout of basis, fully written); either output may be NULL. */
int sm_mask_loss_fwd(const float* basis, int basis_hwc, const float* cof, const float* boxes, const uint8_t* gt,
                     const int64_t* idx_gt, int n, int hm, int wm, float* bce_sum, sm_stream_t stream);
int sm_mask_loss_bwd(const float* basis, int basis_hwc, const float* cof, const float* boxes, const uint8_t* gt,
                     const int64_t* idx_gt, int n, int hm, int wm, const float* grad_sum, float* grad_cof,
                     float* grad_basis, sm_stream_t stream);

/* ---- input pipeline ------------------------------------------------------------------------- */

/* Resize (cv2.INTER_LINEAR geometry, result rounded to uint8 as cv2.resize returns it) -> Normalize ((x - mean) / std,
 * optional BGR->RGB first) -> Pad (zeros) -> float CHW, one launch per image.  Replaces the CPU transforms of
 * M/mmdet/datasets/pipelines/transforms.py (Resize :24-175, Normalize :362-403, Pad :405-455) + ImageToTensor.
 * src uint8 HWC [src_h][src_w][3] on the device; out_chw f32 [3][pad_h][pad_w]; mean/std host float[3]. */
int sm_preprocess_u8(const uint8_t* src, int src_h, int src_w, int new_h, int new_w, int pad_h, int pad_w,
                     const float* mean, const float* std, int to_rgb, float* out_chw, sm_stream_t stream);

/* ---- training-path layers on f32 NCHW tensors ------------------------------------------ */

/* nn.GroupNorm (+ReLU) as used by the head's ConvModules / FeatureAlign (conv_module.py:116-120,
 * sipmask_head.py:42,52).  stats f32 [batch][groups][2] = (mean, rstd) saved for the backward. */
int sm_groupnorm_nchw_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, int batch,
                          int channels, int hw, int groups, float eps, int relu, sm_stream_t stream);
/* dy is masked by (y > 0) when relu.  dx / dgamma / dbeta nullable; dgamma, dbeta are overwritten.
 * scratch: f32 [batch][groups][2] (group sums between the reduction phase and the elementwise phase). */
int sm_groupnorm_nchw_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* stats,
                          float* dx, float* dgamma, float* dbeta, float* scratch, int batch, int channels, int hw,
                          int groups, int relu, sm_stream_t stream);
/* F.interpolate(bilinear, align_corners=False, integer scale_factor) on `planes` = N*C planes of h x w, and its
 * adjoint (dx overwritten). */
int sm_upsample_bilinear_nchw_fwd(const float* x, float* y, int64_t planes, int h, int w, int factor,
                                  sm_stream_t stream);
int sm_upsample_bilinear_nchw_bwd(const float* dy, float* dx, int64_t planes, int h, int w, int factor,
                                  sm_stream_t stream);
/* torch.optim.SGD update with momentum and weight decay (the reference's optimizer, M/mmdet/apis/train.py:92-133):
 * g += wd*p; buf = first_step ? g : momentum*buf + g; p -= lr*buf. */
int sm_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                float weight_decay, int first_step, sm_stream_t stream);

/* sm_sgd_step for ALL parameter tensors in one launch.  items (device): array of
 *   struct { float* param; const float* grad; float* momentum_buf; int64_t n; float lr; float weight_decay; }   (40 bytes)
 * blocks (device) int32 [nblocks][2] = (item index, chunk index): thread block b updates elements
 * [chunk * 4096, min((chunk + 1) * 4096, n)) of its item. */
int sm_sgd_multi(const void* items, const int32_t* blocks, int nblocks, float momentum, int first_step,
                 sm_stream_t stream);

/* ---- SipMask-VIS tracking (V/ = SipMask-VIS/, V/mmdet/models/anchor_heads/sipmask_head.py) -------------- */

/* extract_box_feature_center_single (:768-781) for every detection of the batch: out[b][i][:] = the embedding at
 * floor((x1+x2)/2/stride), floor((y1+y2)/2/stride) of track_feats f32 [batch][h][w][channels] (NHWC), boxes first
 * multiplied by box_mul (res_det_bboxes *= scale_factor, :612-614).  Rows i >= ndet[b] are zeroed. */
int sm_track_gather(const float* track_feats, const float* det, const int32_t* ndet, int batch, int max_num, int h,
                    int w, int channels, float box_mul, float stride, float* out, sm_stream_t stream);

/* Matching scores of one frame against the tracked objects (:623-640, compute_comp_scores :544-562):
 * comp f32 [n][t+1] = log_softmax([0, det_feats . prev_feats^T]) + coeff_score*log(det score)
 *                     + coeff_iou*IoU(+1)(det, prev) + coeff_label*[labels equal]   (column 0 = new object:
 * IoU 0, label term 1), plus the row arg max (first maximum) and its value.  det f32 [n][5], prev_boxes [t][5]. */
int sm_track_match(const float* det_feats, const float* prev_feats, const float* det, const int64_t* det_labels,
                   const float* prev_boxes, const int64_t* prev_labels, int n, int t, int channels,
                   float coeff_score, float coeff_iou, float coeff_label, float* comp, int32_t* match_id,
                   float* match_score, sm_stream_t stream);
/* The tracker of a whole clip on the device (V/mmdet/models/anchor_heads/sipmask_head.py:616-667, frames in order): per
 * frame the comprehensive scores of sm_track_match against the object memory, the reference's sequential identity
 * assignment (best claim per object wins, a detection whose best column is the dummy opens a new object, losers get
 * -1) and the memory update -- one launch per clip instead of a kernel + two device->host syncs per frame.
 * det_feats f32 [T][max_num][C], det f32 [T][max_num][5], det_labels i64 [T][max_num], ndet / is_first i32 [T] (device);
 * the memory mem_feats [cap][C] / mem_boxes [cap][5] / mem_labels [cap] / mem_count i32[1] persists between calls
 * (mem_count = 0: empty, as after reset); comp_ws: reserved, may be NULL (the score rows live in LDS); ids i32 [T][max_num]
 * out (-1 beyond ndet).  max_num <= 64, channels a multiple of 64 up to 1024, cap <= 2000 (LDS); objects beyond cap are
 * not opened (ids -1). */
int sm_track_clip(const float* det_feats, const float* det, const int64_t* det_labels, const int32_t* ndet,
                  const int32_t* is_first, int nframes, int max_num, int channels, float coeff_score, float coeff_iou,
                  float coeff_label, float* mem_feats, float* mem_boxes, int64_t* mem_labels, int32_t* mem_count,
                  int capacity, float* comp_ws, int32_t* ids, sm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SIPMASK_HIP_H */
