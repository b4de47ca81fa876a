/*
 * dynamask_hip.h -- C ABI of libdynamask_hip.so, the MI355X (gfx950) operator
 * library behind the DynaMask mask-head hot path.
 *
 * Contract:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer to fp32
 *     (or int32 / int64 where stated), NCHW-contiguous, owned by the caller;
 *   - no allocation; no state that a caller can observe or has to manage; work is
 *     enqueued on `stream` (a hipStream_t passed as void*; NULL = the null
 *     stream) on the CURRENT device and the call returns without synchronising.
 *     What the library does keep, per device and only as a cache: the device's
 *     compute-unit count and "this kernel's dynamic-LDS limit has been raised"
 *     flags (hipFuncSetAttribute once per device).  A few knobs are read from
 *     the environment on first use, seven in all, none of which changes what is
 *     computed: DM_CONV_TAIL / DM_DCN_TAIL (0: no separate launch for the last,
 *     underfull round of workgroups), DM_CONV1_SMALL_WGS (up to how many 128 x 128
 *     tiles a 1x1 launch takes 128 x 32 tiles instead; default 1.25 per CU, 0: never),
 *     DM_WGRAD_WGS (split-K workgroups of the
 *     weight gradients), DM_ROI_SORT / DM_ROI_SORT_MIN (the RoI ordering launch of
 *     dm_roi_align_fwd_ws; re-read by dm_reload_env_knobs()), DM_CONV_SPLITK is the
 *     host binding's.  (Rounds 2-4 had twenty more -- first-generation kernels and
 *     rejected variants kept for A/B timing; their measurements are in
 *     docs/HISTORY.md and profiles/, the kernels are gone.)
 *     Calls from several host threads are safe (a race only repeats an
 *     idempotent attribute call);
 *   - LDS scatter-accumulators (dm_deform_col2im_coord, the scatter form of
 *     dm_point_sample_bwd) are 64-bit fixed point with 2^-36 resolution: every
 *     addend is the fp32 product gradient * bilinear weight cut to that grid and
 *     integer sums are associative, so a plane's result does not depend on the
 *     order of arrival, for gradient magnitudes in [1.5e-11, 1.3e8]; a non-finite
 *     contribution poisons its output plane with NaN (it is not silently dropped);
 *   - return value: 0 = enqueued, negative = error (DM_ERR_*); the host binding
 *     raises on any non-zero code (dm_error_string()).
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * the reference tree, lslrh/DynaMask).
 */
#ifndef DYNAMASK_HIP_H
#define DYNAMASK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM_OK 0
#define DM_ERR_INVALID_ARG (-1)   /* bad shape / null pointer / unsupported parameter */
#define DM_ERR_LAUNCH (-2)        /* hipLaunch / hipGetLastError reported a failure   */
#define DM_ERR_UNSUPPORTED (-3)   /* valid request the library has no kernel for      */

#define DM_MAX_LEVELS 4
#define DM_MAX_SOURCES 4

typedef void* dm_stream_t; /* hipStream_t */

const char* dm_error_string(int code);
/* ABI version: bumped whenever a signature or the meaning of an argument changes (8: flag bit 3 of dm_conv2d_fwd; 9: RoI assignment / bbox training entry points; 10: dm_detail_target takes the fuse weights from device memory; 11: FCNMaskHead upsample backward; 12: dm_fc_fwd takes a scratch slab, deterministic split-K; 13: dm_deform_coord_grad / dm_deform_col2im, dm_conv2d_fwd_masked, dm_scale, dm_polygon_mask_targets, dm_ignore_columns, dm_upsample2x_bilinear_bwd overwrites; 14: dm_random_sample, dm_bn_relu_maxpool_argmax, the *_fx deterministic accumulators + dm_fx_to_float, dm_mask_loss_fwd_bwd takes a scratch, dm_conv2d_wgrad takes the bias gradient, dm_conv_pack_weight_batch, dm_mask_loss_stage; 15: dm_class_logits_up2x_fwd; 16: dm_conv2d_wgrad_slab / dm_conv2d_wgrad_scratch_floats; 17: dm_class_logits_bwd_slab / dm_class_logits_bwd_scratch_floats; 18: dm_reload_env_knobs, dm_roi_align_fwd_ws / dm_roi_align_workspace_bytes, dm_conv_pack_weight_split / dm_conv_packed_floats_split and flag bits 4, 5 of dm_conv2d_fwd; 19: dm_dcn_bwd_data_fused and its pack; 20: dm_conv2d_fwd_ws / dm_conv2d_splitk_floats; 21: dm_deform_conv_fwd_ws / dm_deform_conv_splitk_floats; 22: the bf16-split layouts (dm_conv_pack_weight_split, dm_conv_packed_floats_split, flag bits 4 / 5 of dm_conv2d_fwd) and the one-kernel DCN data gradient (dm_dcn_bwd_*) REMOVED -- measured, never the parity path, see docs/HISTORY.md; 23: dm_bn_stats takes mean_shift, dm_roi_align_bwd takes the gather form for 16 < P <= 64; 24: dm_build_info; 25: dm_boundary_merge_chain, dm_stage_head_fwd; 26: dm_conv1x1_group_fwd; 27: dm_deform_conv_tout_fwd / dm_deform_conv_tout_supported). */
int dm_abi_version(void);
/* "libdynamask_hip abi=N arch=gfx950 compiler=<clang version> flags=<the product-wide flags of dynamask_amd/build.py>"
 * (static storage).  The library must be compiled WITHOUT packed fp32 instructions (flag "-packed-fp32-ops", see
 * build.py); the host binding checks this string at load time and refuses a library that does not say so.  A build
 * recipe other than build.py passes its flag set as -DDM_BUILD_FLAGS="..."; without it the string says flags=unknown. */
const char* dm_build_info(void);
/* Re-read the DM_ROI_* experiment knobs from the environment (they are otherwise read once, at the first launch, and
 * clamped to validated ranges).  For measurement tools that sweep settings inside one process; no knob changes a result. */
int dm_reload_env_knobs(void);

/* ---------------------------------------------------------------------------
 * K1/K2  multi-level RoIAlign forward (avg pooling, aligned=True, adaptive grid)
 * replaces: SingleRoIExtractor.forward + map_roi_levels
 *           (mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:32-81)
 *           and the per-level mmcv.ops.RoIAlign it builds
 *           (roi_extractors/base_roi_extractor.py:49-55).
 * feats[l]   : [B, C, H[l], W[l]]   l < num_levels (1..4); feats, H, W and
 *              spatial_scales are HOST arrays (feats holds device pointers)
 * rois       : [N, 5] = (batch_idx, x1, y1, x2, y2) in image pixels
 * out        : [N, C, P, P]
 * levels_out : optional [N] int32, the FPN level chosen per RoI (NULL to skip)
 * level = clamp(floor(log2(sqrt(w*h)/finest_scale + 1e-6)), 0, num_levels-1);
 * num_levels == 1 skips the mapping (single-level extractor, base_roi_head.py:53-57).
 * ------------------------------------------------------------------------- */
int dm_roi_align_fwd(const float* const* feats, const int* H, const int* W, const float* spatial_scales,
                     int num_levels, int B, int C, const float* rois, int N, int P, int sampling_ratio,
                     float finest_scale, float* out, int32_t* levels_out, dm_stream_t stream);

/* The same extraction with a caller-provided workspace of dm_roi_align_workspace_bytes(N, P) bytes (device memory,
 * 16-byte aligned, contents irrelevant before and after the call; 0 bytes = no faster path exists for this P).  For the
 * 14x14 extraction of 192 .. 1024 RoIs a first kernel ranks the RoIs by (image, level, 32-pixel row, column) into the
 * workspace and the extraction walks them in that order -- workgroups that run side by side then share their footprints
 * in the XCD's L2 (fabric traffic 261 -> 193 MB per 512 RoIs, 57 -> 50.7 us with the ranking kernel) -- and writes every
 * RoI's rows where dm_roi_align_fwd writes them: the same bits (knob DM_ROI_SORT, default 1).
 * A null / too small workspace falls back to dm_roi_align_fwd's kernels. */
long long dm_roi_align_workspace_bytes(int N, int P);
int dm_roi_align_fwd_ws(const float* const* feats, const int* H, const int* W, const float* spatial_scales,
                        int num_levels, int B, int C, const float* rois, int N, int P, int sampling_ratio,
                        float finest_scale, float* out, int32_t* levels_out, void* workspace,
                        long long workspace_bytes, dm_stream_t stream);

/* K3  RoIAlign backward: scatter-add (float atomics) of grad_out into the
 * per-level feature gradients, which the caller has zero-filled. */
int dm_roi_align_bwd(const float* grad_out, float* const* grad_feats, const int* H, const int* W,
                     const float* spatial_scales, int num_levels, int B, int C, const float* rois, int N,
                     int P, int sampling_ratio, float finest_scale, dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * Weight packing for the implicit-GEMM convolutions: OIHW [Cout, Cin, k, k] ->
 * [k*k][KQ][CoutP][4]: input channels in quads, CoutP = dm_conv_packed_cout(Cout)
 * (zero padded).  The input channels are the concatenation of `num_srcs`
 * sources (src_channels[], HOST array, sum = Cin); each source is padded with
 * zero rows to a multiple of 8 channels, KQ = sum(roundup(Cs, 8)) / 4, total
 * size dm_conv_packed_floats() floats.
 * transpose_flip != 0 packs the weights of the data-gradient convolution
 * (in/out channels swapped, taps rotated by 180 degrees): input is still the
 * forward OIHW tensor, the packed tensor then has "Cout" = Cin of the forward
 * and src_channels must sum to the forward's Cout.
 * ------------------------------------------------------------------------- */
int dm_conv_packed_cout(int Cout);
long long dm_conv_packed_floats(int Cout, int ksize, int num_srcs, const int* src_channels);
int dm_conv_pack_weight(const float* w_oihw, int Cout, int Cin, int ksize, int transpose_flip,
                        int num_srcs, const int* src_channels, float* w_packed, dm_stream_t stream);

/* All the packs of a training step in one launch (the weights change with every optimizer step).  A job is
 * dm_conv_pack_weight's arguments plus a window: the packed [Cout][Cin] tensor may be the input-channel slice
 * [c0, c0 + Cin) of a [Cout][ld][k][k] tensor (the data gradient towards one concat source); ld = Cin, c0 = 0 for
 * a whole tensor.  `jobs_device`: num_jobs structs in DEVICE memory (the caller uploads the table once and reuses
 * it while the tensors stay where they are). */
typedef struct dm_pack_job {
  const float* w;
  float* w_packed;
  int Cout, Cin, ksize, transpose_flip;
  int num_srcs, src_channels[DM_MAX_SOURCES];
  int ld, c0;
} dm_pack_job;
int dm_conv_pack_weight_batch(const dm_pack_job* jobs_device, int num_jobs, dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * K5/K6  dense convolution forward, stride 1, "same" padding, ksize in {1,3},
 * fp32 in / fp32 accumulate on the MFMA units, fused concat + bias + ReLU.
 * replaces: mmcv ConvModule / nn.Conv2d calls of
 *           mask_heads/dynamask_head.py:73,83,86,104,117-121,203-213,221-222,
 *           mask_heads/fcn_mask_head.py:59-71,119-120,102,125.
 * srcs[s]  : [NB, src_channels[s], H, W]; the channel-concatenation of the
 *            sources is the conv input (torch.cat at dynamask_head.py:107-116
 *            is folded into the K loop), sum(src_channels) = Cin.
 *            src_batch_strides[s] (floats; NULL = dense) lets a source be a
 *            channel slice of a wider tensor.  srcs / src_channels /
 *            src_batch_strides are HOST arrays (of device pointers / ints).
 * w_packed : dm_conv_pack_weight layout, bias: [Cout] or NULL
 * relu     : flags -- bit 0: fused ReLU; bit 1: accumulate (out += result; used
 *            for gradient sums in the backward); bit 3: the caller overlaps this
 *            launch with work on another stream (a scheduling hint: the 3x3
 *            kernel then does not split off its last round of workgroups;
 *            results are the same bits either way); any other bit: DM_ERR_INVALID_ARG
 * out      : written at channels [out_ch_offset, out_ch_offset+Cout) of a
 *            tensor [NB, out_ch_total, H, W]
 * ------------------------------------------------------------------------- */
int dm_conv2d_fwd(const float* const* srcs, const int* src_channels, const long long* src_batch_strides,
                  int num_srcs, int NB, int H, int W,
                  const float* w_packed, const float* bias, int Cout, int ksize, int relu, float* out,
                  int out_ch_total, int out_ch_offset, dm_stream_t stream);
/* (ABI 20) dm_conv2d_fwd with a caller-owned workspace, for the calls of real inference (the reference runs the head on at
 * most 100 RoIs, dynamask_roi_head.py:132-135): a launch that would leave most of the chip idle splits its K loop over up
 * to eight workgroups per tile; the splits store bare sums to the workspace ([split][NB][Cout][HW]) and a second kernel
 * adds them in split order and applies bias / accumulate / ReLU.  The same bits every run; they differ from dm_conv2d_fwd's
 * in rounding only (another association of the same products).  dm_conv2d_splitk_floats: the workspace with which this
 * shape splits as far as it wants to, 0 when it would not split (then call dm_conv2d_fwd). */
long long dm_conv2d_splitk_floats(int NB, int H, int W, int Cout, int ksize);
int dm_conv2d_fwd_ws(const float* const* srcs, const int* src_channels, const long long* src_batch_strides,
                     int num_srcs, int NB, int H, int W, const float* w_packed, const float* bias, int Cout, int ksize,
                     int relu, float* out, int out_ch_total, int out_ch_offset, float* workspace,
                     long long workspace_floats, dm_stream_t stream);

/* (ABI 26) Up to three independent single-source 1x1 convolutions (+ bias, + ReLU) as ONE launch -- the FPN-wide
 * semantic_transform_in convolutions of the three SFM stages (mmdet/models/roi_heads/mask_heads/dynamask_head.py:104,
 * relu(conv1x1(P4 / P3 / P2))), which no RoI enters and which otherwise head the inference chain as three launches.
 * x, Cin, H, W, w_packed, bias, Cout, out: HOST arrays of `count` entries (device pointers inside); w_packed[i] as
 * dm_conv_pack_weight(ksize 1, one source) lays it out; bias[i] may be NULL.  Same bits as dm_conv2d_fwd per problem. */
int dm_conv1x1_group_fwd(int count, const float* const* x, const int* Cin, const int* H, const int* W, int NB,
                         const float* const* w_packed, const float* const* bias, const int* Cout, int relu, float* const* out,
                         dm_stream_t stream);

/* dm_conv2d_fwd whose epilogue also applies a ReLU adjoint: outputs where `mask` (same layout, channel count and
 * channel offset as `out`) is not > 0 are stored as 0.  Used for data gradients: the mask is the activation the
 * gradient flows into, so the separate mask pass (read gradient + activation, write gradient) disappears. */
int dm_conv2d_fwd_masked(const float* const* srcs, const int* src_channels, const long long* src_batch_strides,
                         int num_srcs, int NB, int H, int W, const float* w_packed, const float* bias, int Cout,
                         int ksize, int relu, float* out, int out_ch_total, int out_ch_offset, const float* mask,
                         dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * K4  SimpleRoIAlign / point_sample forward (grid_sample bilinear, zero
 * padding, align_corners=False at RoI-relative pixel centres).
 * replaces: mmcv.ops.SimpleRoIAlign at mask_heads/dynamask_head.py:74,105.
 * feat [B, C, H, W], rois [N,5] -> out [N, C, S, S]
 * ------------------------------------------------------------------------- */
int dm_point_sample_fwd(const float* feat, int B, int C, int H, int W, const float* rois, int N, int S,
                        float spatial_scale, float* out, dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * K7  per-RoI class-gathered 1x1 logits, two branches at once.
 * replaces: instance_logits(x)[arange(N), labels][:, None] and the detail twin
 *           (mask_heads/dynamask_head.py:110-113, 236-237).
 * x [N, C, HW]; w_inst/w_det [num_classes, C]; b_* [num_classes];
 * labels [N] int64 (values clamped to [0, num_classes-1] by the caller's contract)
 * inst/det      : [N, HW] raw logits
 * sig_out       : optional; sigmoid(inst), sigmoid(det) written to channels
 *                 sig_ch_offset, sig_ch_offset+1 of a [N, sig_ch_total, HW] tensor
 * ------------------------------------------------------------------------- */
int dm_class_logits_fwd(const float* x, int N, int C, int HW, const float* w_inst, const float* b_inst,
                        const float* w_det, const float* b_det, int num_classes, const int64_t* labels,
                        float* inst, float* det, float* sig_out, int sig_ch_total, int sig_ch_offset,
                        dm_stream_t stream);

/* (ABI 25) One SFM stage's head in one launch: dm_point_sample_fwd(sem, rois -> sampled [N,Cs,S,S]) and
 * dm_class_logits_fwd(x [N,C,S,S] -> inst, det (+ sigmoid slices of sig_out)) -- mmdet/models/roi_heads/mask_heads/
 * dynamask_head.py:104-116; the two share no data, the kernel runs the two bodies in disjoint workgroup ranges (same bits). */
int dm_stage_head_fwd(const float* sem, int B, int Cs, int H, int W, const float* rois, int N, int S, float spatial_scale,
                      float* sampled, const float* x, int C, const float* w_inst, const float* b_inst, const float* w_det,
                      const float* b_det, int num_classes, const int64_t* labels, float* inst, float* det, float* sig_out,
                      int sig_ch_total, int sig_ch_offset, dm_stream_t stream);

/* K7 on relu(upsample2x(x)) (bilinear, align_corners=False) without the upsampled tensor: the logits an exit needs when
 * the stage before it would only have been upsampled for them -- dynamask_head.py:120-122 (F.interpolate + relu) followed
 * by :110-113 of the next stage / :236-237.  x [N, C, H, W] (W even, H, W >= 2: else DM_ERR_UNSUPPORTED and the caller
 * runs the two kernels); inst/det [N, 2H, 2W].  The interpolated values are those of dm_upsample2x_bilinear_fwd bit for
 * bit, and the channel sum runs in the four interleaved subsets of dm_class_logits_fwd. */
int dm_class_logits_up2x_fwd(const float* x, int N, int C, int H, int W, const float* w_inst, const float* b_inst,
                             const float* w_det, const float* b_det, int num_classes, const int64_t* labels,
                             float* inst, float* det, dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * K8  deformable convolution v1 forward, 3x3, stride 1, pad 1, dilation 1,
 * groups 1, no bias, fused ReLU option.
 * replaces: mmcv.ops.DeformConv2dPack (DCN) at mask_heads/dynamask_head.py:84,118-119;
 *           arithmetic spec mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:84-115,190-243,
 *           mmdet/ops/dcn/src/deform_conv_cuda.cpp:152-260.
 * x [NB, C, H, W]; offset [NB, deform_groups*18, H, W] (dh,dw interleaved per tap);
 * w_packed: dm_conv_pack_weight(ksize=3, one source) of the [Cout, C, 3, 3] weight; out [NB, Cout, H, W]
 * relu: bit 0 fused ReLU; bit 3 the same scheduling hint as in dm_conv2d_fwd
 * ------------------------------------------------------------------------- */
int dm_deform_conv_fwd(const float* x, const float* offset, int NB, int C, int H, int W,
                       const float* w_packed, int Cout, int deform_groups, int relu, float* out,
                       dm_stream_t stream);
/* (ABI 27) K8 + the 1x1 behind it in one launch: relu(DCN 3x3) -> 1x1 conv + bias + ReLU, i.e. SFMStage.fuse_conv[1]
 * followed by fuse_transform_out (mmdet/models/roi_heads/mask_heads/dynamask_head.py:117-121), for the 28 x 28 / 56 x 56
 * stages at more than a handful of RoIs (dm_deform_conv_tout_supported: 1 = this shape takes the kernel build in which a
 * wave holds every output channel of its pixels; the 1x1 then runs on the accumulators in registers).
 *   w2t  [Cout][M2P]  the 1x1 weight [M2, Cout] transposed, M2P = M2 rounded up to 32, zeros in the padding
 *   out2 [NB, out2_ch_total, H, W]: channels [0, M2) are written;   out_dcn: NULL (the DCN output is not stored) or
 *   [NB, Cout, H, W].  Returns DM_ERR_UNSUPPORTED for other shapes.  Bits: those of dm_deform_conv_fwd + dm_conv2d_fwd. */
int dm_deform_conv_tout_supported(int NB, int C, int H, int W, int Cout, int M2);
int dm_deform_conv_tout_fwd(const float* x, const float* offset, int NB, int C, int H, int W, const float* w_packed, int Cout,
                            int deform_groups, const float* w2t, const float* b2, int M2, float* out2, int out2_ch_total,
                            float* out_dcn, dm_stream_t stream);

/* (ABI 21) dm_deform_conv_fwd with a caller-owned workspace: a launch that leaves most of the chip idle (the <= 100-RoI
 * inference calls) splits its channel loop over up to eight workgroups per tile; a second kernel adds the splits in index
 * order (+ ReLU).  Same bits every run; they differ from dm_deform_conv_fwd's by the association of the channel sums.
 * dm_deform_conv_splitk_floats: the workspace with which this shape may split as far as it wants to (0: it would not). */
long long dm_deform_conv_splitk_floats(int NB, int C, int H, int W, int Cout);
int dm_deform_conv_fwd_ws(const float* x, const float* offset, int NB, int C, int H, int W, const float* w_packed, int Cout,
                          int deform_groups, int relu, float* out, float* workspace, long long workspace_floats,
                          dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * K11  x2 bilinear upsampling.
 * replaces: nn.Upsample(scale_factor=2, mode='bilinear') + ReLU
 *           (mask_heads/dynamask_head.py:87,123; align_corners=False) and
 *           F.interpolate(scale_factor=2, 'bilinear', align_corners=True)
 *           (mask_heads/dynamask_head.py:239-243).
 * in [NC, H, W] -> out [NC, 2H, 2W]
 * ------------------------------------------------------------------------- */
int dm_upsample2x_bilinear_fwd(const float* in, int NC, int H, int W, int align_corners, int relu,
                               float* out, dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * K15  inference boundary-aware merge of one coarse->fine stage pair.
 * replaces: the loop body of DynaMaskRoIHead.simple_test_mask
 *           (roi_heads/dynamask_roi_head.py:138-149) incl. generate_block_target
 *           (losses/cross_entropy_loss.py:123-154) with boundary_width=1.
 * coarse [n, S, S] logits, fine [n, 2S, 2S] logits (overwritten in place where
 * the x2 align_corners=True upsampled non-boundary mask is >= 0.5).
 * ------------------------------------------------------------------------- */
int dm_boundary_merge(const float* coarse, float* fine, int n, int S, dm_stream_t stream);
/* (ABI 25) The whole inference tail of dynamask_roi_head.py:138-149 in ONE launch: the two dependent merges
 * S -> 2S -> 4S and, when final_2s is given, the align_corners x2 upsample (dynamask_head.py:240-243,
 * F.interpolate(..., scale_factor=2, align_corners=True)) that produces the 4S x 4S logits the second merge overwrites:
 *   p_s [n,S,S], p_2s [n,2S,2S] (read only: the merged 2S logits are temporaries and are not written back),
 *   final_2s [n,2S,2S] or NULL (then out_4s holds the fine logits on entry and is merged in place),  out_4s [n,4S,4S].
 * Bits: those of dm_upsample2x_bilinear_fwd + dm_boundary_merge(S) + dm_boundary_merge(2S). */
int dm_boundary_merge_chain(const float* p_s, const float* p_2s, const float* final_2s, float* out_4s, int n, int S,
                            dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * K16  deconv 2x2 stride 2 (+bias, +ReLU): nn.ConvTranspose2d(C, Cout, 2, 2)
 * replaces: FCNMaskHead.upsample 'deconv' (mask_heads/fcn_mask_head.py:77-83,121-124).
 * x [NB, C, H, W]; w_packed = dm_deconv_pack_weight of the [C, Cout, 2, 2] weight;
 * out [NB, Cout, 2H, 2W]
 * ------------------------------------------------------------------------- */
int dm_deconv_pack_weight(const float* w_iohw, int Cin, int Cout, float* w_packed, dm_stream_t stream);
int dm_deconv2x2_fwd(const float* x, int NB, int C, int H, int W, const float* w_packed, const float* bias,
                     int Cout, int relu, float* out, dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * K17  CARAFE: kernel normaliser (pixel_shuffle + softmax over k*k) fused with
 * the reassembly.  replaces: mmcv CARAFEPack.kernel_normalizer +
 * feature_reassemble (mask_heads/fcn_mask_head.py:84-87,121).
 * x [NB, C, H, W]; enc [NB, k*k*group*scale*scale, H, W] (content encoder
 * output, before pixel shuffle); out [NB, C, scale*H, scale*W]
 * ------------------------------------------------------------------------- */
int dm_carafe_fwd(const float* x, const float* enc, int NB, int C, int H, int W, int up_kernel, int group,
                  int scale, float* out, dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * K10  resolution selector: straight-through Gumbel-softmax (hard), T = 0.5.
 * replaces: DynaMaskRoIHead.get_mask_label / gumbel_softmax
 *           (roi_heads/dynamask_roi_head.py:84-114); U is the explicit uniform
 *           noise the reference draws at :90.
 * logits, U [N, 4] -> y_soft [N,4] (softmax((logits+g)/T)), index [N] int32
 * (first maximum, as torch.max), one_hot [N,4]
 * ------------------------------------------------------------------------- */
int dm_gumbel_select_fwd(const float* logits, const float* U, int N, int K, float temperature,
                         float* y_soft, float* one_hot, int32_t* index, dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * K13  DetailTarget: Laplacian boundary pyramid target, bit-exact {0,1}.
 * replaces: DetailTarget.forward (losses/cross_entropy_loss.py:363-418).
 * masks [N, S, S] in {0,1} -> out [N, S, S]; fuse = the two fuse_kernel weights, either as host
 * values or (fuse_dev != NULL: two floats in device memory, read by the kernel -- the reference keeps
 * them as an nn.Parameter that weight decay changes every step, Quirk Q7; no host round trip)
 * ------------------------------------------------------------------------- */
int dm_detail_target(const float* masks, int N, int S, float fuse0, float fuse1, const float* fuse_dev, float* out,
                     dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * K12  mask losses of one stage, forward + gradients in one pass.
 * replaces: binary_cross_entropy (losses/cross_entropy_loss.py:56-87) and the
 *           fork's eps-BCE mask_cross_entropy (:90-120) as used at :458-462.
 * inst_pred, det_pred, inst_tgt, det_tgt : [N, HW]; weight [N] (mask_labels[:,idx])
 * sums (device, 2 floats, caller zero-fills): sums[0] += sum of BCE-with-logits,
 *   sums[1] += sum_n weight[n] * sum_p -(t log(s+eps) + (1-t) log(1-s+eps))
 * per_roi_det [N] (optional): the un-weighted per-RoI eps-BCE sums (for d/dweight)
 * grad_inst / grad_det (optional): d sums[0]/d inst_pred, d(eps-BCE)/d det_pred
 *   scaled by weight[n]; the host applies the scalar normalisers.
 * ------------------------------------------------------------------------- */
/* scratch: dm_mask_loss_scratch_floats(N) floats (per-workgroup partial sums, added in a fixed order: the loss and
 * d loss / d weight have the same bits on every run). */
long long dm_mask_loss_scratch_floats(int N);
int dm_mask_loss_fwd_bwd(const float* inst_pred, const float* det_pred, const float* inst_tgt,
                         const float* det_tgt, const float* weight, int N, int HW, float* sums,
                         float* per_roi_det, float* grad_inst, float* grad_det, float* scratch, dm_stream_t stream);

/* One stage of DynaCrossEntropyLoss.forward (cross_entropy_loss.py:455-476) with its normalisers applied in the kernel:
 * w_n = mask_labels[n, stage] ([N, num_stages]), den = sum_n w_n + 1e-5 (detached, :462), n_el = N * HW.
 *   loss_terms[0]  = mean BCE-with-logits of this stage (overwritten: only the last stage's reaches the loss, Quirk Q2)
 *   loss_terms[1] += detail_weight * (N / n_el) / den * sum_n w_n * epsBCE_n
 *   grad_mask_labels[n, stage] = detail_weight / (HW * den) * epsBCE_n           (d loss / d mask_labels, den detached)
 *   grad_inst (optional) = (sigmoid(x) - t) / n_el;  grad_det (optional) = d(detail term) / d det_pred
 * scratch: dm_mask_loss_scratch_floats(N).  Replaces dm_mask_loss_fwd_bwd + ~30 host-side tensor operations per stage. */
int dm_mask_loss_stage(const float* inst_pred, const float* det_pred, const float* inst_tgt, const float* det_tgt,
                       const float* mask_labels, int num_stages, int stage, int N, int HW, float detail_weight,
                       float* loss_terms, float* grad_mask_labels, float* grad_inst, float* grad_det, float* scratch,
                       dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * K9  resolution predictor MaskPre (roi_heads/base_roi_head.py:10-27): BatchNorm
 * batch statistics (train mode; biased variance for normalisation, running
 * stats updated with the unbiased one when the pointers are non-NULL) and the
 * fused BN -> ReLU -> max_pool2d(kernel 3, stride 2, pad 1).  The convs / FCs
 * of MaskPre run through dm_conv2d_fwd.
 * x [NB, C, H, W]; mean/var/gamma/beta [C]; out [NB, C, (H-1)/2+1, (W-1)/2+1]
 * ------------------------------------------------------------------------- */
/* mean_shift (ABI 23; [C] or NULL): the running mean is updated with mean + mean_shift.  Used when x is the activation
 * WITHOUT a per-channel bias that train-mode BatchNorm cancels anyway: MaskPre's conv1 applied to the P2 map before the
 * 56 x 56 extraction (conv1 is 1x1, RoIAlign linear: conv1(RoIAlign(x)) = RoIAlign(W1 x) + b1), base_roi_head.py:13-16. */
int dm_bn_stats(const float* x, int NB, int C, int HW, float* mean, float* var, float* running_mean,
                float* running_var, float momentum, const float* mean_shift, float* scratch, dm_stream_t stream);
/* scratch for dm_bn_stats / dm_bn_relu_maxpool_bwd (may be null: one workgroup per channel) */
long long dm_bn_scratch_floats(int C);
int dm_bn_relu_maxpool_fwd(const float* x, int NB, int C, int H, int W, const float* mean, const float* var,
                           const float* gamma, const float* beta, float eps, float* out, dm_stream_t stream);
/* diagnostic: argmax [NB, C, OH, OW] int32 = plane index (y * W + x) of the tap each pooled output took (first
 * maximum in scan order, the choice dm_bn_relu_maxpool_bwd routes the gradient to) -- what
 * F.max_pool2d(..., return_indices=True) returns in the reference (roi_heads/base_roi_head.py:16,19). */
int dm_bn_relu_maxpool_argmax(const float* x, int NB, int C, int H, int W, const float* mean, const float* var,
                              const float* gamma, const float* beta, float eps, int32_t* argmax, dm_stream_t stream);

/* backward of dm_bn_relu_maxpool_fwd with train-mode statistics (mean/var as
 * returned by dm_bn_stats): grad_x [NB,C,H,W] (overwritten), grad_gamma/grad_beta [C]. */
int dm_bn_relu_maxpool_bwd(const float* x, int NB, int C, int H, int W, const float* mean, const float* var,
                           const float* gamma, const float* beta, float eps, const float* grad_out, float* grad_x,
                           float* grad_gamma, float* grad_beta, float* scratch, dm_stream_t stream);

/* K10 backward: gradient of the soft branch of the straight-through estimator
 * (y_hard = (one_hot - y).detach() + y, dynamask_roi_head.py:112-113). */
int dm_gumbel_select_bwd(const float* y_soft, const float* grad_y, int N, int K, float temperature,
                         float* grad_logits, dm_stream_t stream);

/* ---------------------------------------------------------------------------
 * K14  class-balance entropy of the selector and its gradient.
 * replaces: losses/cross_entropy_loss.py:478-481.
 * mask_labels [N, K] -> loss[0] = sum_k p_k log(p_k + 1e-10), p = colsum/total;
 * grad [N, K] (optional) = d loss / d mask_labels
 * ------------------------------------------------------------------------- */
int dm_class_balance_fwd_bwd(const float* mask_labels, int N, int K, float* loss, float* grad,
                             dm_stream_t stream);

/* ===========================================================================
 * Callers either side of the path ("next" rows 1 and 2).
 * =========================================================================== */

/* Mask-target RoIs: (gt_index, box clipped to the mask canvas).
 * replaces: the numpy clip of DynaMaskHead.get_targets (mask_heads/dynamask_head.py:248-256)
 * and the rois assembly of BitmapMasks.crop_and_resize (core/mask/structures.py:270-276);
 * the crop itself is dm_roi_align_fwd on the GT bitmaps (scale 1, adaptive grid),
 * then dm_threshold_ge(0.5) (structures.py:281-284). */
int dm_mask_target_rois(const float* boxes, const int64_t* gt_inds, int N, float max_w, float max_h, float* rois,
                        dm_stream_t stream);
int dm_threshold_ge(const float* x, long long count, float thr, float* out, dm_stream_t stream);

/* Mask targets from POLYGON annotations: out[n] = S x S bitmap (0 / 1) of object inds[n] in the frame of boxes[n].
 * replaces: PolygonMasks.crop_and_resize + to_ndarray -> polygon_to_bitmap -> pycocotools frPyObjects / merge / decode
 * (core/mask/structures.py:469-503, 544-552, 583-599) as DynaMaskHead.get_targets calls them per image and size
 * (mask_heads/dynamask_head.py:248-262), bit-identical to cocoapi's rleFrPoly arithmetic.
 * verts: [V][2] float64 (x, y) of all polygons of the image; poly_start: [P + 1] first vertex of each polygon;
 * inst_start: [num_objects + 1] first polygon of each object; boxes: [N][4] float32 already clipped to the image;
 * inds: [N]; S <= 256.  One workgroup per RoI. */
int dm_polygon_mask_targets(const double* verts, const int* poly_start, const int* inst_start, int num_objects,
                            const float* boxes, const int64_t* inds, int N, int S, float* out, dm_stream_t stream);

/* K18  paste N masks [N, mask_h, mask_w] into their boxes on an [img_h, img_w] canvas
 * and binarise: out[n, y, x] = (grid_sample(mask_n) >= threshold) as uint8.
 * replaces: _do_paste_mask (mask_heads/fcn_mask_head.py:240-308, skip_empty=False) +
 * the threshold of get_seg_masks (mask_heads/dynamask_head.py:325-339).
 * apply_sigmoid != 0 takes logits (mask_pred.sigmoid() of dynamask_head.py:281). */
int dm_paste_masks(const float* masks, const float* boxes, int N, int mask_h, int mask_w, int img_h, int img_w,
                   float threshold, int apply_sigmoid, uint8_t* out, dm_stream_t stream);

/* K19  COCO run-length encoding on the device.
 * replaces: encode_mask_results (mmdet/core/mask/utils.py:36-63: pycocotools rleEncode of a
 * column-major host copy of every bitmap) and, in the fused form, the [N, img_h, img_w]
 * canvas of get_seg_masks (mask_heads/dynamask_head.py:325-342) together with its
 * device->host copy.  Outputs: mask_runs[n] = number of run boundaries of mask n,
 * mask_start[n] (N+1 entries) = offset of its boundaries in `positions` (packed, column-major
 * pixel indices j = x*img_h + y where the value changes; the value before j = 0 is 0).
 * Boundaries past `capacity` are counted but not stored (caller re-runs with a larger buffer).
 * seg_scratch: dm_rle_scratch_ints(N, img_h, img_w) int32.
 * dm_rle_string (host code, no GPU work): boundaries -> run lengths -> the printable
 * `counts` string of the COCO RLE format; returns its length, or -(needed) if cap is short. */
long long dm_rle_scratch_ints(int N, int img_h, int img_w);
int dm_rle_encode_canvas(const uint8_t* canvas, int N, int img_h, int img_w, int* seg_scratch, int* mask_runs,
                         int* mask_start, int* positions, int capacity, dm_stream_t stream);
int dm_paste_rle(const float* masks, const float* boxes, int N, int mask_h, int mask_w, int img_h, int img_w,
                 float threshold, int apply_sigmoid, int* seg_scratch, int* mask_runs, int* mask_start,
                 int* positions, int capacity, dm_stream_t stream);
long long dm_rle_string(const int* positions, int runs, long long total_pixels, char* out, long long cap);

/* K22  fully connected layer out[N, M] = x[N, K] . w[M, K]^T + bias (nn.Linear layouts), optional
 * ReLU; fp32 MFMA.  replaces: the nn.Linear stack of Shared2FCBBoxHead
 * (roi_heads/bbox_heads/convfc_bbox_head.py:101-108,143-186) and MaskPre's fc1 / fc2
 * (roi_heads/base_roi_head.py:17-18,24-26).  K % 4 == 0.  K is split over workgroups in segments
 * whose length depends on K alone; the partial sums go to `scratch` (dm_fc_scratch_floats() floats; may be NULL when that
 * is 0) and are added in a fixed order: an output row has the same bits whatever N is, run to run. */
long long dm_fc_scratch_floats(int N, int K, int M);
int dm_fc_fwd(const float* x, const float* w, const float* bias, int N, int K, int M, int relu, float* out,
              float* scratch, dm_stream_t stream);

/* K20  bbox branch post-processing: softmax over the class logits, DeltaXYWH decode, clip to
 * the image, rescale.  replaces: BBoxHead.get_bboxes up to the NMS
 * (roi_heads/bbox_heads/bbox_head.py:186-217) and delta2bbox
 * (core/bbox/coder/delta_xywh_bbox_coder.py:165-204).  rois [N, roi_stride] with x1 at column
 * roi_x0 (5 / 1 for the reference's [batch, x1, y1, x2, y2] rows); cls_score [N, num_classes+1]
 * or null; bbox_pred [N, 4*num_classes] ([N, 4] if class_agnostic) or null (then the rois
 * themselves are clipped); clip_h/clip_w <= 0: no clipping; scale_x/scale_y: divide the boxes
 * (rescale=True), 1 otherwise.  Outputs scores [N, num_classes+1], bboxes like bbox_pred.
 * K21  dm_nms_mask: suppression bit matrix of M score-sorted boxes [M, 4] (bit j of row i,
 * j > i, set when IoU > iou_threshold; offset 0/1 as mmcv.ops.nms) -- replaces the device half of
 * mmcv.ops.nms as called by multiclass_nms (core/post_processing/bbox_nms.py:5-68);
 * dm_nms_reduce is the host half (greedy pass), returns the number of kept indices. */
int dm_bbox_decode(const float* rois, int roi_stride, int roi_x0, const float* cls_score, const float* bbox_pred,
                   int N, int num_classes, int class_agnostic, const float* means, const float* stds,
                   float wh_ratio_clip, float clip_h, float clip_w, float scale_x, float scale_y, float* scores,
                   float* bboxes, dm_stream_t stream);
int dm_nms_mask(const float* boxes_sorted, int M, float iou_threshold, int offset, unsigned long long* mask,
                dm_stream_t stream);
int dm_nms_reduce(const unsigned long long* mask_host, int M, int* keep, int max_keep);

/* ===========================================================================
 * Backward (training step).  Replaces what autograd derives for the reference
 * modules above plus mmcv's DeformConv2d backward
 * (mmdet/ops/dcn/src/deform_conv_cuda.cpp:262-486, deform_conv_cuda_kernel.cu:117-188,279-436).
 * The conv DATA gradient is dm_conv2d_fwd with dm_conv_pack_weight(transpose_flip=1).
 * =========================================================================== */

/* grad[i] = 0 where out[i] <= 0 (in place). */
int dm_relu_bwd(float* grad, const float* out, long long count, dm_stream_t stream);

/* g_logit[n,p] (+)= (ga[n,p] + gb[n,p]) * s(1-s), s = sig[n,p]; sig / ga / gb may be
 * channel slices of wider tensors (batch strides in floats); gb may be NULL. */
int dm_sigmoid_bwd(const float* sig, long long sig_bs, const float* ga, long long ga_bs, const float* gb,
                   long long gb_bs, int N, int HW, float* g_logit, int accumulate, dm_stream_t stream);

/* out[c] (+)= sum_{n,p} g[n, c, p]  (bias gradient). */
int dm_channel_sum(const float* g, long long batch_stride, int NB, int C, int HW, float* out, int accumulate,
                   dm_stream_t stream);

/* conv weight gradient, fp32 MFMA GEMM over the pixel dimension, atomically
 * accumulated: dw[co*ldw + col_offset + ci*k*k + tap] += sum_{n,y,x} dy[n,co,y,x] *
 * x[n,ci,y+dy-1,x+dx-1].  One call per concat source (col_offset = channel base * k*k);
 * the caller zero-fills dw.  db (optional, [Cout]): the bias gradient db[co] += sum_{n,y,x} dy[n,co,y,x], taken from
 * the dy values the kernel stages anyway (pass it with ONE of the sources of a concat). */
int dm_conv2d_wgrad(const float* dy, long long dy_batch_stride, int Cout, const float* x, long long x_batch_stride,
                    int Cs, int NB, int H, int W, int ksize, float* dw, int ldw, int col_offset, float* db,
                    dm_stream_t stream);

/* The same sums without atomics: every split of the pixel axis writes its partial tile to a slab of `scratch`, a second
 * kernel adds the slabs in index order into dw (and the splits' bias rows into db) -- bit-identical from run to run, as the
 * reference's weight gradient is (a deterministic addmm_, mmdet/ops/dcn/src/deform_conv_cuda.cpp:460-465).  scratch: at
 * least dm_conv2d_wgrad_scratch_floats() floats for ANY shape, 16-byte aligned (a launch that would need more, or a scratch
 * that is not aligned, falls back to the atomics of dm_conv2d_wgrad); contents undefined afterwards; not shared between streams.  dw / db are accumulated into, as above. */
long long dm_conv2d_wgrad_scratch_floats(void);
int dm_conv2d_wgrad_slab(const float* dy, long long dy_batch_stride, int Cout, const float* x, long long x_batch_stride,
                         int Cs, int NB, int H, int W, int ksize, float* dw, int ldw, int col_offset, float* db,
                         float* scratch, long long scratch_floats, dm_stream_t stream);

/* adjoint of dm_upsample2x_bilinear_fwd; fwd_out_for_relu (optional) masks the
 * fused ReLU; grad_in is overwritten (no zero-fill needed; planes up to 64 KB of
 * output gradient are gathered through LDS, larger ones are cleared and scattered). */
int dm_upsample2x_bilinear_bwd(const float* grad_out, const float* fwd_out_for_relu, int NC, int H, int W,
                               int align_corners, float* grad_in, dm_stream_t stream);

/* adjoint of dm_point_sample_fwd, scatter-add into grad_feat (caller zero-fills). */
int dm_point_sample_bwd(const float* grad_out, int B, int C, int H, int W, const float* rois, int N, int S,
                        float spatial_scale, float* grad_feat, dm_stream_t stream);

/* backward of dm_class_logits_fwd: grad_x (+)= W[label] * grad; grad_w/grad_b rows of
 * the RoI's class accumulated atomically (caller zero-fills them). */
int dm_class_logits_bwd(const float* x, int N, int C, int HW, const float* w_inst, const float* w_det,
                        int num_classes, const int64_t* labels, const float* grad_inst, const float* grad_det,
                        float* grad_x, int accumulate_x, float* grad_w_inst, float* grad_b_inst, float* grad_w_det,
                        float* grad_b_det, dm_stream_t stream);
/* The same gradients without contended atomics: every RoI's sums go to `scratch` (dm_class_logits_bwd_scratch_floats(N, C)
 * floats, contents undefined afterwards) and a second launch adds, per class, the RoIs of that class in RoI order.  The RoIs
 * of an image share a few classes: 256 RoIs of one class are 256 serialised atomics per address in dm_class_logits_bwd
 * (27 -> 130 us at 256 x 256 x 14 x 14).  Deterministic (fixed order of additions; the reference's autograd accumulates the
 * gathered rows unordered: mmdet/models/roi_heads/mask_heads/dynamask_head.py:112-113 `[torch.arange(n), labels]`). */
long long dm_class_logits_bwd_scratch_floats(int N, int C);
int dm_class_logits_bwd_slab(const float* x, int N, int C, int HW, const float* w_inst, const float* w_det,
                             int num_classes, const int64_t* labels, const float* grad_inst, const float* grad_det,
                             float* grad_x, int accumulate_x, float* grad_w_inst, float* grad_b_inst, float* grad_w_det,
                             float* grad_b_det, float* scratch, long long scratch_floats, dm_stream_t stream);

/* DCNv1 backward pieces.  col / colgrad: [NB, 9*C, H, W] with rows tap-major
 * (row = tap*C + ci).  dm_dcn_weight_permute converts W[co][ci][tap] <->
 * Wt[(tap*C+ci)][co] so that both GEMMs run as 1x1 convs over the column matrix.
 * dm_deform_col2im_coord overwrites grad_x and grad_offset (no zero-fill needed):
 * the scatter-adds of one (image, channel) plane are accumulated in LDS. */
int dm_deform_im2col(const float* x, const float* offset, int NB, int C, int H, int W, int deform_groups,
                     float* col, dm_stream_t stream);
int dm_deform_col2im_coord(const float* colgrad, const float* x, const float* offset, int NB, int C, int H, int W,
                           int deform_groups, float* grad_x, float* grad_offset, dm_stream_t stream);
/* the two halves of dm_deform_col2im_coord as separate launches (ABI 13): they share only their inputs, so a caller
 * may issue them on two streams -- the coordinate gradient is bound by its gathers, col2im by LDS atomics. */
int dm_deform_coord_grad(const float* colgrad, const float* x, const float* offset, int NB, int C, int H, int W,
                         int deform_groups, float* grad_offset, dm_stream_t stream);
int dm_deform_col2im(const float* colgrad, const float* offset, int NB, int C, int H, int W, int deform_groups,
                     float* grad_x, dm_stream_t stream);
int dm_dcn_weight_permute(const float* src, float* dst, int Cout, int C, int to_colmajor, int accumulate,
                          dm_stream_t stream);

/* SGD(momentum, weight decay) step on a flat fp32 buffer; grad_scale folds the
 * 1/world_size of the gradient all-reduce (apis/train.py:75-79 DDP averaging). */
int dm_sgd_momentum_step(float* params, const float* grads, float* momentum_buf, long long count, float lr,
                         float momentum, float weight_decay, float grad_scale, int first_step, dm_stream_t stream);

/* Backward of the FCNMaskHead upsample layers (mask_heads/fcn_mask_head.py:84-96; the forward of the
 * deconv / CARAFE / bilinear forms is above).
 * dm_carafe_bwd: gradients of dm_carafe_fwd with respect to x and enc (mmcv carafe backward +
 *   kernel_normalizer backward), for scale 2 and H*W <= 256 (the mask head: 14x14 -> 28x28); other shapes
 *   return DM_ERR_UNSUPPORTED.  scratch: dm_carafe_bwd_scratch_floats() floats.
 * dm_upsample2x_nearest_fwd / _bwd: nn.Upsample(scale_factor=2, mode='nearest') and its adjoint.
 * dm_pixel_unshuffle2x: out[n, (dy*2+dx)*C + c, y, x] = in[n, c, 2y+dy, 2x+dx]: with it the backward of the
 *   2x2 stride-2 ConvTranspose2d is dm_conv2d_fwd (data) + dm_conv2d_wgrad (weights) on 1x1 GEMMs. */
long long dm_carafe_bwd_scratch_floats(int NB, int H, int W, int up_kernel, int group);
int dm_carafe_bwd(const float* x, const float* enc, const float* grad_out, int NB, int C, int H, int W, int up_kernel,
                  int group, int scale, float* grad_x, float* grad_enc, float* scratch, dm_stream_t stream);
int dm_upsample2x_nearest_fwd(const float* in, int NC, int H, int W, float* out, dm_stream_t stream);
int dm_upsample2x_nearest_bwd(const float* grad_out, int NC, int H, int W, float* grad_in, dm_stream_t stream);
int dm_pixel_unshuffle2x(const float* in, int NB, int C, int H, int W, float* out, dm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training side of the RoI head around the mask path (SURVEY 8b "forward_train", 8f rank 4):
 * RoI assignment / sampling inputs, bbox regression targets and the two bbox-branch losses.
 * ------------------------------------------------------------------------------------------ */

/* bbox_overlaps(bboxes1 [n1,4], bboxes2 [n2,4], mode, is_aligned=False, eps) -> [n1, n2]
 * (core/bbox/iou_calculators/iou2d_calculator.py:37-131); mode_iof: 0 = 'iou', 1 = 'iof'.
 * Rounded exactly like the reference's separate torch ops (no fused multiply-add). */
int dm_bbox_overlaps(const float* bboxes1, int n1, const float* bboxes2, int n2, int mode_iof, float eps, float* out,
                     dm_stream_t stream);

/* MaxIoUAssigner.assign_wrt_overlaps (core/bbox/assigners/max_iou_assigner.py:129-212) for
 * num_gts > 0 and num_bboxes > 0: overlaps [num_gts, num_bboxes] -> gt_inds [n] (-1 ignore,
 * 0 negative, i+1 = gt i), max_overlaps [n], labels [n] (gt label or -1; NULL with gt_labels
 * NULL).  A float neg_iou_thr t is (neg_iou_lo, neg_iou_hi) = (0, t).  scratch: 2*num_gts floats. */
int dm_max_iou_assign(const float* overlaps, int num_gts, int num_bboxes, float pos_iou_thr, float neg_iou_lo,
                      float neg_iou_hi, float min_pos_iou, int match_low_quality, int gt_max_assign_all,
                      const int64_t* gt_labels, float* scratch, int64_t* gt_inds, float* max_overlaps, int64_t* labels,
                      dm_stream_t stream);

/* RoI sampling: BaseSampler.sample + RandomSampler._sample_pos / _sample_neg + SamplingResult
 * (core/bbox/samplers/base_sampler.py:35-101, random_sampler.py:31-75, sampling_result.py:21-49),
 * stated as a selection by key.  Candidate boxes bboxes [M,4] (the n_prepended ground-truth boxes first
 * when the sampler adds them as proposals) with AssignResult.gt_inds [M] (i+1 = gt i, 0 negative,
 * -1 ignored) and optional labels [M].  quota_pos = int(num * pos_fraction) positives at most, then
 * num - (positives kept) negatives at most, further limited to int(neg_pos_ub * max(1, positives kept))
 * when neg_pos_ub >= 0.  A class within its quota is kept whole; a larger one keeps the boxes with
 * the quota's smallest keys (ties: lower index) -- a uniformly random subset when the keys are
 * i.i.d. noise.  keys_by_class_rank = 0: pos_keys / neg_keys are [M], indexed by box (both may be the
 * same noise tensor); 1: pos_keys[r] / neg_keys[r] belong to the r-th positive / negative in index
 * order (the inverse of the reference's torch.randperm(count) reproduces its choice: test hook).
 * Keys must be finite.  Kept boxes leave in ascending index order (the reference's .unique()).
 * Outputs have capacity `num` rows; counts[4] = (positives kept, negatives kept, positive candidates,
 * negative candidates).  pos_gt_bboxes[o] = gt_bboxes[gt_inds - 1], pos_assigned_gt_inds = gt_inds - 1,
 * pos_is_gt[o] = index < n_prepended, pos_gt_labels (NULL with labels NULL) = labels[index].
 * scratch: 3 * M int32.  One workgroup; O(M^2 / 1024) compares per thread when a class is over quota. */
int dm_random_sample(const int64_t* gt_inds, const float* bboxes, int M, int n_prepended, const float* gt_bboxes,
                     int num_gts, const int64_t* labels, const float* pos_keys, const float* neg_keys,
                     int keys_by_class_rank, int num, int quota_pos, double neg_pos_ub, int32_t* scratch,
                     int64_t* pos_inds, int64_t* neg_inds, int32_t* counts, float* pos_bboxes, float* neg_bboxes,
                     float* pos_gt_bboxes, int64_t* pos_assigned_gt_inds, int64_t* pos_gt_labels, uint8_t* pos_is_gt,
                     dm_stream_t stream);
/* the gt_bboxes_ignore branch of MaxIoUAssigner.assign (max_iou_assigner.py:107-118): overlaps[:, n] = -1 where
 * box n's largest IoF with an ignore region exceeds thr.  iof = dm_bbox_overlaps(mode_iof = 1) of (boxes, regions)
 * [N][I] (boxes_major = 1, ignore_wrt_candidates) or of (regions, boxes) [I][N] (boxes_major = 0). */
int dm_ignore_columns(float* overlaps, int G, int N, const float* iof, int I, int boxes_major, float thr,
                      dm_stream_t stream);

/* bbox2delta (core/bbox/coder/delta_xywh_bbox_coder.py:74-116): proposals, gt [n,4] -> deltas [n,4]. */
int dm_bbox_encode(const float* proposals, const float* gt, int n, const float* means, const float* stds, float* deltas,
                   dm_stream_t stream);

/* cross_entropy(pred, label, weight, reduction='mean', avg_factor) * loss_weight, forward and
 * backward in one pass (losses/cross_entropy_loss.py:9-38, utils.py:26-52), plus
 * accuracy(pred, label) top-1 in percent (losses/accuracy.py:4-49):
 *   loss[0]    = scale * sum_i weight_i * (logsumexp(score_i) - score_i[label_i]),  scale = loss_weight / avg_factor
 *   grad[i, c] = scale * weight_i * (softmax(score_i)[c] - [c == label_i])          (NULL: forward only)
 *   correct[0] = 100 / N * #{i : argmax_c score_i[c] == label_i}                     (NULL: skipped)
 * weight NULL = all ones.  scratch: 2*N floats.  Row sums are added in a fixed order. */
int dm_softmax_ce_fwd_bwd(const float* cls_score, const int64_t* labels, const float* weight, int N, int C, float scale,
                          float* scratch, float* loss, float* correct, float* grad, dm_stream_t stream);

/* BBoxHead.loss, regression half (roi_heads/bbox_heads/bbox_head.py:159-182) with L1Loss
 * (losses/smooth_l1_loss.py:29-42,104-136): rows with 0 <= label < num_classes are positive;
 * their prediction is bbox_pred[i, label_i, :] (num_boxes_per_row = num_classes) or
 * bbox_pred[i, 0, :] (class agnostic, num_boxes_per_row = 1):
 *   loss[0] = scale * sum_pos sum_c |pred - target| * weight,   scale = loss_weight / avg_factor
 *   grad    = d loss / d bbox_pred [N, num_boxes_per_row*4] (overwritten; NULL: forward only)
 * No positive row: loss 0, grad 0 (the reference's `bbox_pred.sum() * 0`).  scratch: N floats. */
int dm_l1_loss_fwd_bwd(const float* bbox_pred, const int64_t* labels, const float* targets, const float* weights, int N,
                       int num_boxes_per_row, int num_classes, float scale, float* scratch, float* loss, float* grad,
                       dm_stream_t stream);

/* Gradient clipping of the flat gradient buffer (optimizer_config grad_clip,
 * configs/dynamask/coco/r50-dynamask-1x.py:274; OptimizerHook.py:10-14 -> clip_grad_norm_):
 * dm_sumsq: out[0] = sum x^2 (fixed-order two-stage sum; scratch dm_sumsq_scratch_floats());
 * dm_clip_scale: x *= min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) with sumsq read on the device
 * (the caller may have added the other parameter groups' sums into it). */
long long dm_sumsq_scratch_floats(void);
int dm_sumsq(const float* x, long long count, float* scratch, float* out, dm_stream_t stream);
int dm_clip_scale(float* x, long long count, const float* sumsq, float max_norm, dm_stream_t stream);
/* x *= factor: a parameter group's gradient scale (the reference's optional OptimizerHook_ multiplies the
 * gradients of roi_head.mask_predictor by 0.05 between clipping and the step, OptimizerHook.py:27-29). */
int dm_scale(float* x, long long count, float factor, dm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Deterministic accumulation (SURVEY 7 "hard parts": a deterministic mode for parity tests; the
 * reference's weight gradient is a deterministic addmm_, mmdet/ops/dcn/src/deform_conv_cuda.cpp:460-465).
 * The entry points above that accumulate across workgroups -- dm_conv2d_wgrad, dm_channel_sum,
 * dm_class_logits_bwd (its four parameter gradients), dm_point_sample_bwd -- add with float atomics, so the
 * last bits of their sums depend on the order in which workgroups arrive.  Each has an *_fx twin with the
 * same arguments whose accumulation target is a buffer of 64-bit FIXED-POINT cells (value * 2^36, two's
 * complement; same indexing as the float target, the caller zero-fills it): integer addition is associative,
 * so the sum is exact in that format and identical on every run.  dm_fx_to_float converts
 * (out[i] (+)= fx[i] * 2^-36, NaN for a cell that received a non-finite addend; `clear` re-zeroes fx).
 * Range: |sum| < 3.3e7 per cell, resolution 1.5e-11 -- far outside what gradients of this path reach.
 * Everything else in the training step is deterministic by construction (fixed-order partial sums, LDS
 * fixed-point scatter, gather-form adjoints) except dm_roi_align_bwd (float atomics into the FPN-map
 * gradients, as in the reference's RoIAlign backward).  The host switch is DM_DETERMINISTIC=1
 * (dynamask_amd.ops.DETERMINISTIC).
 * ------------------------------------------------------------------------------------------ */
int dm_conv2d_wgrad_fx(const float* dy, long long dy_batch_stride, int Cout, const float* x, long long x_batch_stride,
                       int Cs, int NB, int H, int W, int ksize, long long* dw_fx, int ldw, int col_offset,
                       long long* db_fx, dm_stream_t stream);
int dm_channel_sum_fx(const float* g, long long batch_stride, int NB, int C, int HW, long long* out_fx, dm_stream_t stream);
int dm_class_logits_bwd_fx(const float* x, int N, int C, int HW, const float* w_inst, const float* w_det, int num_classes,
                           const int64_t* labels, const float* grad_inst, const float* grad_det, float* grad_x,
                           int accumulate_x, long long* grad_w_inst_fx, long long* grad_b_inst_fx,
                           long long* grad_w_det_fx, long long* grad_b_det_fx, dm_stream_t stream);
int dm_point_sample_bwd_fx(const float* grad_out, int B, int C, int H, int W, const float* rois, int N, int S,
                           float spatial_scale, long long* grad_feat_fx, dm_stream_t stream);
int dm_fx_to_float(long long* fx, long long n, float* out, int accumulate, int clear, dm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DYNAMASK_HIP_H */
