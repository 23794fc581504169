/* tscd_hip.h -- C-ABI of libtscd_hip.so: the MI355X (gfx950) kernels of the TSCD per-clip hot path.
 *
 * The reference (Video-Object-Detection/TSCD) has no FFI boundary: its hot path is stock PyTorch ops called from
 * Python modules.  This ABI therefore sits one layer below the module mirror in yolox/ (which reproduces the
 * reference's import surface) and each entry point names the reference code it replaces (paths relative to the
 * reference tree, file:line).  INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer unless marked (host); `stream` is a hipStream_t.
 *   - all entry points are asynchronous on `stream`, never allocate, never synchronise; scratch memory is passed in
 *     (`*_workspace_bytes` tells how much).  They are re-entrant per stream (no global mutable state).
 *   - return 0 on success, 1 = bad argument, 2 = HIP launch error; tscd_last_error() returns a thread-local message.
 *   - activations are fp16, NHWC ([B,H,W,C] row-major) or a channel slice of such a buffer: `ld*` is the pixel
 *     stride in elements (>= C).  Channel counts / strides that feed 16-byte vector accesses must be multiples of 8.
 */
#ifndef TSCD_HIP_H
#define TSCD_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int tscd_abi_version(void);
const char* tscd_last_error(void);
/* Queues the host function fn(arg) (void (*)(void*)) on `stream`: everything queued after it starts once it has returned.  Used
 * to park the stream on the assignment chain's latch (include/tscd_host.h: tscd_latch_wait_cb + tscd_latch_ticket) while the
 * host keeps queueing the work that follows the assignments; the reference blocks in scipy instead (tscd_matching.py:935). */
int tscd_stream_host_func(void* stream, void* fn, void* arg);

/* ---- convolution / GEMM (MFMA implicit GEMM) -------------------------------------------------------------------
 * Replaces Conv2d inside BaseConv (yolox/models/network_blocks.py:36-61), the prediction convs
 * (yolox/models/tscd_head.py:213-239), WaveletsHFBlock.filter1/2 (yolox/models/surrounding_extraction.py:223-232)
 * and, with ksize=1/H=1, every nn.Linear of the proposal stage (post_trans.py:565-576,1139-1143;
 * tscd_matching.py:24-26,151-153,704; tscd_head.py:123,130-133).
 * epilogue: 0 raw (+ per-M-tile BN partial sums into bn_partial[2][cstride][mtiles] when non-NULL), 1 +bias,
 * 2 +bias,ReLU, 3 folded BN + SiLU (eval), 4 folded BN, 5 Hungarian cost 1 - v/2 with NaN->0
 * (tscd_matching.py:926-930). */
int tscd_conv_tile_m(void);
/* Per-channel BatchNorm partial-sum entries a raw-epilogue launch of this shape writes (= its number of M-tiles, which
 * depends on the main loop it takes): the `mtiles` argument of tscd_bn_finalize.  M = B*Ho*Wo, W = output width. */
int tscd_conv_stat_tiles(long M, int W, int Cin, int Cout, int ksize, int stride);
int tscd_conv_stat_tiles_ex(long M, int W, int Cin, int Cout, int ksize, int stride, int groups, int variant, int tile_m);
int tscd_conv_tile_n(int cout);
int tscd_pack_conv_weight(const void* w, int src_is_f32, void* out, int Cout, int Cin, int CinPad, int KH, int KW,
                          int rows_pad, int Kpad, int mode, void* stream);
/* The same packing for many weights in one launch (after an optimizer step).  jobs: device array of 56-byte records {const void* w;
 * f16* out; int Cout, Cin, CinPad, KH, KW, rows_pad, Kpad, mode, src_is_f32, pad}; blockmap: one {int job, int tile} per
 * workgroup; tile t of a job is the (cout, cin) sub-block [T (t / nI), T (t / nI) + T) x [T (t % nI), T (t % nI) + T) of the weight with
 * T = tscd_pack_block_elems() and nI = ceil(Cin / T); a job has ceil(Cout / T) * nI tiles.  Padding rows / columns of `out` are not
 * written: `out` must come from tscd_pack_conv_weight (which zero-fills them). */
int tscd_pack_block_elems(void);
int tscd_pack_conv_weights_batched(const void* jobs, const void* blockmap, long nblocks, void* stream);
int tscd_conv2d_nhwc_f16(const void* x, const void* wpk, void* y, const float* bias, const float* scale,
                         const float* shift, float* bn_partial, int B, int H, int W, int Cin, int Cout, int ksize,
                         int stride, int ldx, int ldy, int epilogue, void* stream);
/* The general form.  groups > 1: `groups` convolutions of Cin -> Cout channels each in ONE launch (the four towers of
 * TSCDHead.forward, tscd_head.py:315-324): group g reads channels [g*Cin, (g+1)*Cin) of x, the g-th packed matrix of wpk (each
 * padded to a multiple of tscd_conv_tile_n(Cout) rows) and writes channels [g*Cout, (g+1)*Cout) of y; bias / scale / shift run
 * over the groups*Cout concatenated channels, bn_partial is [2][groups * rows_pad][tscd_conv_stat_tiles_ex(...)].
 * Main-loop plan, per call (there is no process-global switch): variant -1 automatic -- LDS-DMA ring (global_load_lds, 4 stages)
 * when Cin % 32 == 0 and Cout > 32, its halo-staged form (input pixels fetched once per 32-channel chunk and shared by the 9 taps,
 * 192- or 256-pixel x 256-channel tiles) for 3x3 / stride-1 convs with Cout % 256 == 0 and at least ~one tile per CU, register
 * staging otherwise; 0 register staging, 1 LDS-DMA 128-wide tiles only, 5 / 6 the halo form with 256- / 192-pixel tiles wherever
 * its shape allows (tests, A/B timing).  korder 1: the register-staged loop walks k as (32-channel chunk, tap) like the halo form
 * -- the two are then bit-identical (tests/test_gpu_conv.py); the other forms accumulate in (tap, channel) order and are
 * bit-identical among themselves.  tile_m 64 / 128 / 192: tile height of the 128-wide LDS-DMA form (0 automatic). */
int tscd_conv2d_nhwc_f16_ex(const void* x, const void* wpk, void* y, const float* bias, const float* scale,
                            const float* shift, float* bn_partial, int B, int H, int W, int Cin, int Cout, int ksize,
                            int stride, int ldx, int ldy, int epilogue, int groups, int variant, int korder, int tile_m,
                            void* stream);

/* ---- backward of the trainable head convs (autograd of network_blocks.py:36-61 / surrounding_extraction.py:223-232
 * under tscd_trainer.py:171-176).  Weight gradient of a stride-1 "same" conv as a pixel-axis GEMM with
 * ds_read_b64_tr_b16 operand fetch and a deterministic split-K reduction; dw is fp32 in torch layout
 * [Cout][Cin][KH][KW].  The data gradient is tscd_conv2d_nhwc_f16 on dY with tscd_pack_conv_weight(mode=1).
 * tscd_bn_silu_bwd_f16: gradient of SiLU(BatchNorm_train(raw)); sums_out = [dbeta | dgamma]. */
int tscd_conv_wgrad_splits(long M, int Cout, int Cin, int ksize);
long tscd_conv_wgrad_workspace_bytes(long M, int Cout, int Cin, int ksize);
int tscd_conv_wgrad_nhwc_f16(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout, int ksize,
                             int ldx, int lddy, void* workspace, long workspace_bytes, void* stream);
/* groups > 1: `groups` weight gradients in one launch (group g from channels [g*Cin, ...) of x and [g*Cout, ...) of dy into
 * dw[g][Cout][Cin][k][k]; Cin / Cout per group).  variant (tests): 1 the LDS-DMA ring (global_load_lds, swizzled lane-linear tiles),
 * 0 the register-staged loop kept as the cross-check -- same tiles, splits and accumulation order, dw bit-identical. */
long tscd_conv_wgrad_workspace_bytes_ex(long M, int Cout, int Cin, int ksize, int groups);
int tscd_conv_wgrad_nhwc_f16_ex(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout, int ksize,
                                int ldx, int lddy, int groups, int variant, void* workspace, long workspace_bytes, void* stream);
long tscd_bn_silu_bwd_workspace_bytes(long M, int C);
/* dY in up to two channel segments ([0, cseg) from dy0, [cseg, C) from dy1, each with its own row stride), gamma in up to two
 * vectors ([0, gseg) / [gseg, C)), raw / draw as channel slices: the two trainable towers of TSCDHead share one conv launch, their
 * BatchNorms stay separate modules. */
int tscd_bn_silu_bwd_f16_ex(const void* dy0, long lddy0, const void* dy1, long lddy1, int cseg, const void* raw, long ldraw,
                            const float* scale_shift, const float* mean, const float* invstd, const float* gamma0,
                            const float* gamma1, int gseg, long M, int C, void* draw, long lddraw, float* sums_out, void* workspace,
                            long workspace_bytes, void* stream);
/* Backward of an nn.Linear (y = x W^T + b) in one host call: dx = dY W (MFMA GEMM on W packed in data-gradient order),
 * dw = dY^T x (pixel-axis wgrad GEMM), db = column sums.  dY [n][outp] is fp16 with row stride lddy and outp = out_f rounded up to 8
 * (extra columns zero); dx [n][in_f] fp16, dw [outp][in_f] fp32, db [outp] fp32; any output may be NULL.
 * tscd_bias_relu_bwd_f16: db[c] = sum_r g[r][c], optionally through the mask of a fused ReLU (y > 0), in which case the masked
 * gradient gm is written as well -- the bias gradient of a Linear / biased conv in one pass. */
long tscd_bias_relu_bwd_workspace_bytes(long M, int C);
int tscd_bias_relu_bwd_f16(const void* g, int ldg, const void* y, int ldy, void* gm, int ldm, long M, int C, float* db,
                           void* workspace, long workspace_bytes, void* stream);
long tscd_linear_bwd_workspace_bytes(int n, int in_f, int outp);
int tscd_linear_bwd_f16(const void* x, int ldx, const void* dy, int lddy, const void* w, int w_is_f32, int n, int in_f, int out_f,
                        int outp, void* dx, float* dw, float* db, void* workspace, long workspace_bytes, void* stream);
/* The same with W already packed in data-gradient order by the caller (tscd_pack_conv_weight mode 1 of the [out_f][in_f][1][1] view,
 * cached per parameter version and refreshed by tscd_pack_conv_weights_batched): no pack launch inside the call.  NULL = pack here. */
int tscd_linear_bwd_packed_f16(const void* x, int ldx, const void* dy, int lddy, const void* w, int w_is_f32,
                               const void* w_dgrad_packed, int n, int in_f, int out_f, int outp, void* dx, float* dw, float* db,
                               void* workspace, long workspace_bytes, void* stream);
int tscd_bn_silu_bwd_f16(const void* dy, const void* raw, const float* scale_shift, const float* mean,
                         const float* invstd, const float* gamma, long M, int C, void* draw, float* sums_out,
                         void* workspace, long workspace_bytes, void* stream);

/* ---- BatchNorm (train-mode statistics / eval fold) + activation: network_blocks.py:52-58 with eps 1e-3,
 * momentum 0.03 (exps/TSCD_OVIS/ovis_tscd_large.py:109-113); Bottleneck shortcut add (network_blocks.py:176-179). */
int tscd_bn_finalize(const float* partial, int mtiles, int cstride, int C, long count, const float* gamma,
                     const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                     float* scale_shift, float* save_mean, float* save_invstd, void* stream);
/* BaseConv in training mode (network_blocks.py:36-61 with BatchNorm in train mode) as one host call: the three launches above
 * (conv with partial sums, finalize, apply + activation + residual).  small: [4][C] floats = scale | shift | save_mean | save_invstd;
 * partial: [2][cstride][mtiles] floats; raw may alias out. */
int tscd_conv_bn_act_train_f16(const void* x, const void* wpk, void* raw, int ldraw, float* partial, int cstride, int B, int H, int W,
                               int Cin, int Cout, int ksize, int stride, int ldx, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, float* small, void* out, int ldo,
                               const void* residual, int ldr, int act, void* stream);
/* The same for SEVERAL BaseConvs in one conv launch (the four towers of TSCDHead.forward, tscd_head.py:315-324): groups == 1 -- convs
 * that share their INPUT run as one dense conv over the nseg * seg_c concatenated output channels (the stem output is staged once for
 * all of them); groups == nseg -- convs with their own input / output channel slices run as one grouped launch.  gamma / beta /
 * running_mean / running_var: (host) arrays of nseg device pointers, one BatchNorm each; small [4][nseg*seg_c]; partial
 * [2][rows][tscd_conv_stat_tiles_ex(...)] with rows = groups * (per-group output channels padded to tscd_conv_tile_n). */
int tscd_conv_bn_act_train_multi_f16(const void* x, const void* wpk, void* raw, int ldraw, float* partial, int B, int H, int W, int Cin,
                                     int seg_c, int nseg, int groups, int ksize, int ldx, const void* const* gamma,
                                     const void* const* beta, float eps, float momentum, void* const* running_mean,
                                     void* const* running_var, float* small, void* out, int ldo, int act, void* stream);
int tscd_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                 float* scale_shift, void* stream);
int tscd_bn_act_nhwc_f16(const void* x, int ldx, void* y, int ldy, const float* scale_shift, const void* residual,
                         int ldr, long M, int C, int act, void* stream);

/* ---- layout / pooling helpers: Focus (network_blocks.py:267-289), nn.Upsample + torch.cat
 * (yolo_pafpn.py:97-115), SPPBottleneck max-pools (network_blocks.py:201-223). */
int tscd_focus_nchw_to_nhwc16(const void* x, int src_is_f32, void* y, int B, int H, int W, void* stream);
int tscd_copy_slice_nhwc_f16(const void* src, int lds_, void* dst, int ldd, int B, int H, int W, int C, int upsample,
                             void* stream);
int tscd_maxpool_nhwc_f16(const void* src, int lds_, void* dst, int ldd, int B, int H, int W, int C, int ks,
                          void* stream);

/* ---- Haar wavelet high-frequency block: DWT_2D / IDWT_2D / WaveletsHFBlock.forward
 * (surrounding_extraction.py:12-127,257-267). */
int tscd_haar_hf_nhwc_f16(const void* x, int ldx, void* hf, int B, int H, int W, int C, void* stream);
int tscd_haar_idwt_mul_nhwc_f16(const void* hf, const void* content, int ldc, void* out, int ldo, int B, int H, int W,
                                int C, void* stream);
/* Backward of the two Haar steps (DWT_Function.backward / IDWT_Function.backward, surrounding_extraction.py:30-41,60-76):
 * d haar_hf = tscd_haar_idwt_mul_nhwc_f16 with content == NULL (the synthesis with LL = 0 is the adjoint of the analysis);
 * out = content * idwt(hf)  =>  dcontent = d * idwt(hf), dhf = dwt_hf(d * content) in one pass. */
/* ModelEMA.update (yolox/utils/ema.py:48-60) over all fp32 state tensors in one launch: v = fma(1-d, model_v, d*v).  table: device
 * array of nchunks records {float* dst; const float* src; long n} with n <= tscd_ema_chunk_elems() (one workgroup per record). */
int tscd_ema_chunk_elems(void);
int tscd_ema_update_f32(const void* table, long nchunks, float d, float one_minus_d, void* stream);
/* Step tail in one launch (reference ovis_tscd_large.py:164-197, tscd_trainer.py:171-178, ema.py:48-60): Nesterov SGD
 * with GradScaler unscale + skip-on-overflow and the EMA of the same parameters.  table: device array of nchunks records
 * {float* p, const float* grad, float* momentum_buffer (zero before the first step), float* ema or NULL, long n, long group}
 * (<= tscd_ema_chunk_elems() elements each); lr / wd: per-group host arrays (<= 8 groups); grad_scale / found_inf: device scalars or
 * NULL; d, one_minus_d: EMA decay of this step. */
int tscd_sgd_ema_step_f32(const void* table, long nchunks, const float* lr, const float* wd, int ngroups, float momentum, int nesterov,
                          const float* grad_scale, const float* found_inf, float d, float one_minus_d, void* stream);
/* The same with the gradients addressed through a device array of per-parameter pointers (grad_base[i] = gradient of parameter i):
 * chunk records then hold the byte offset of their slice in the `grad` field and `group | i << 8` in the group field.  The chunk
 * table no longer depends on where this step's gradients were allocated. */
int tscd_sgd_ema_step_indirect_f32(const void* table, long nchunks, const void* grad_base, const float* lr, const float* wd, int ngroups,
                                   float momentum, int nesterov, const float* grad_scale, const float* found_inf, float d,
                                   float one_minus_d, void* stream);
/* SEVERAL weight gradients in one launch + one slab-reduce launch: the autograd of the proposal stage's nn.Linear layers
 * (tscd_head.py:123,130-133 fc_reg_matcher / cls_pred / matcher_*_pred, post_trans.py:566-577,1119-1125 MCA projections,
 * tscd_matching.py:31-60,704 attention projections -- ~25 per step, a few thousand rows each) and of the trained head convs
 * (tscd_head.py:240-281 cls_convs2 / reg_convs2, surrounding_extraction.py:223-232 wavelet filters, :213-239 prediction convs), none of
 * which the backward pass waits for.  jobs: HOST array of njobs (<= tscd_wgrad_batch_max_jobs()) records {const void* x; const void*
 * dy; float* dw; int B, H, W, cin, cout, ksize, ldx, lddy, groups, pad}: a stride-1 "same" conv over B x H x W pixels (a Linear over M
 * rows: B = H = 1, W = M, ksize = 1); x [pixels][groups*cin], dy [pixels][groups*cout] fp16 with pixel strides ldx / lddy; dw
 * [groups*cout][cin][k][k] fp32.  tscd_wgrad_batch_supported(...) says whether a shape can be a job.  workspace: device memory of
 * tscd_wgrad_batch_workspace_bytes(jobs, njobs) bytes (fp32 slabs of the split jobs, added in slab order: deterministic). */
int tscd_wgrad_batch_max_jobs(void);
int tscd_wgrad_batch_supported(int B, int H, int W, int cin, int cout, int ksize, int ldx, int lddy, int groups);
long tscd_wgrad_batch_workspace_bytes(const void* jobs_host, int njobs);
int tscd_wgrad_batch_f16(const void* jobs_host, int njobs, void* workspace, long workspace_bytes, void* stream);
/* Bias gradients of several nn.Linear layers in one launch (the same deferred pass as tscd_wgrad_batch_f16): jobs = HOST array of njobs
 * records {const void* dy [M][C] fp16; float* db [C]; int M, C (multiple of 8), lddy, pad}; db_j[c] = sum over the M rows, fp32, fixed order. */
int tscd_colsum_batch_f16(const void* jobs_host, int njobs, void* stream);
/* The step's gradients into one flat fp32 arena and / or their non-finite check, one launch: replaces the GradScaler's inf check
 * inside scaler.step (tscd_trainer.py:171-176; torch: 6 multi-tensor launches) and the flattening of gradients into all-reduce
 * buckets that DistributedDataParallel does per parameter (tscd_trainer.py:230-231).  table: device array of nrows records
 * {float* dst or NULL (check only), long src_off (bytes), long n (<= tscd_ema_chunk_elems()), long pi}; the source of a row is
 * src_base[pi] + src_off (src_base: device array of per-parameter gradient pointers); dst = src * scale; *found_inf (device scalar or
 * NULL) is set to 1 when any element read is Inf / NaN and never cleared. */
int tscd_grad_gather_check_f32(const void* table, long nrows, const void* src_base, float scale, float* found_inf, void* stream);
int tscd_haar_idwt_mul_bwd_nhwc_f16(const void* d, const void* hf, const void* content, int ldc, void* dcontent, void* dhf,
                                    int B, int H, int W, int C, void* stream);

/* ---- proposal stage: decode_outputs (tscd_head.py:755-770), postprocess_widx (tscd_head.py:1546-1693),
 * find_feature_score (tscd_head.py:976-1006).  hw3/w3/stride3 are (host) int[3]; maps is a (host) array of 9 device
 * pointers [feature][level]. */
int tscd_decode_score(const void* raw0, const void* raw1, const void* raw2, const int* hw3, const int* w3,
                      const int* stride3, int B, int NC, void* outputs, void* det, void* score, void* stream);
int tscd_select_proposals(const void* score, int B, int A, float conf_thresh, int min_keep, int max_keep, int cap,
                          int* out_idx, int* out_count, void* stream);
/* map_ld: pixel stride (elements) of every map -- W for contiguous maps, wider when the maps are channel slices of one buffer. */
int tscd_gather_rows(const void* const* maps, const int* hw3, const int* sel_idx, int cap, const int* frame_off, int B,
                     int N, int W, int map_ld, int nmaps, void* out0, void* out1, void* out2, int scatter, void* stream);
int tscd_gather_det(const void* det, int A, int ndet, const int* sel_idx, int cap, const int* frame_off, int B, int N,
                    void* out, void* stream);
/* The same, also returning every gathered row's (frame, anchor) as int32 vectors (both or neither may be NULL). */
int tscd_gather_det_rows(const void* det, int A, int ndet, const int* sel_idx, int cap, const int* frame_off, int B, int N, void* out,
                         int* row_b, int* row_a, void* stream);

/* ---- MCA, the fused global-to-local cosine attention: Attention_mca_g2l.forward + find_similar_round2
 * (post_trans.py:581-714) over all local frames of MCA_tscd_g2l_reg.forward (post_trans.py:1127-1162).
 * tiles: device int[ntiles][4] = (row0, nrows, loc_start, loc_end). */
int tscd_mca_dp(int d);
long tscd_mca_workspace_bytes(int N, int N_L, int W, int S);
/* rows_workspace (optional, tscd_mca_rows_bytes): where the per-head normalised q / k / v rows go -- the part of the forward's state the
 * backward reads in place (tscd_mca_backward_fws); NULL = the front of `workspace`.  fp16_logits != 0: the reference-faithful
 * arithmetic of post_trans.py:655-657 -- q.k^T, x 25 and x class score each rounded to fp16 before the softmax (the reference's
 * autocast / .half() tensors); 0 = fp32 logits (bounded in tests/test_deviation_bounds.py). */
long tscd_mca_rows_bytes(int N, int N_L, int W);
int tscd_mca_forward(const void* q_c, const void* kv_c, const void* q_r, const void* kv_r, const void* score,
                     const int* tiles, int ntiles, int N, int N_L, int W, int S, int need_reg, float scale,
                     float sim_thresh, float conf_thresh, void* workspace, long workspace_bytes, void* xcat_c,
                     void* xcat_r, void* cat3_c, void* cat3_r, float* stats_out, void* rows_workspace, int fp16_logits,
                     void* stream);
/* Backward of the fused attention (autograd of post_trans.py:601-714 under tscd_trainer.py:171-176): gradients w.r.t. the
 * projected q / (k|v) rows and the key scores.  stats = stats_out of the forward call; dxcat_* [N_L,2W], dsoft_* / soft_* the
 * first W columns of the [N_L,3W] concat (row strides ld_dsoft / ld_soft). */
long tscd_mca_backward_workspace_bytes(int N, int N_L, int W, int S, int ntiles);
/* Three passes without atomics (delta + dM/Ac/Ar blocks; dq per query tile; dk/dv/dscore per key block): deterministic.
 * dxcat_c = dsoft_c = NULL (need_reg only): no gradient reaches the class branch's attention outputs -- agg_iou, whose class output
 * feeds the assignment cost detached (tscd_head.py:480-483,491-500; tscd_matching.py:919-926).  The passes leave out everything
 * that would only carry those zeros (12 of 32 MFMAs per tile pair in the first pass, dvc, Ac): same bits as zero tensors. */
int tscd_mca_backward(const void* q_c, const void* kv_c, const void* q_r, const void* kv_r, const void* score,
                      const int* tiles, int ntiles, int N, int N_L, int W, int S, int need_reg, float scale, float sim_thresh,
                      float conf_thresh, const float* stats, const void* dxcat_c, const void* dxcat_r, const void* dsoft_c,
                      const void* dsoft_r, int ld_dsoft, const void* soft_c, const void* soft_r, int ld_soft, void* workspace,
                      long workspace_bytes, void* dq_c, void* dkv_c, void* dq_r, void* dkv_r, float* dscore, void* stream);
/* The same reading the normalised q / k / v rows tscd_mca_forward wrote for the same inputs (its rows_workspace, or the front of its
 * workspace; fwd_rows_bytes >= tscd_mca_rows_bytes, checked) instead of re-deriving them (no prep launch); fwd_rows NULL =
 * tscd_mca_backward.  Both files take the layout from one definition (mca_common.h: McaRows). */
int tscd_mca_backward_fws(const void* q_c, const void* kv_c, const void* q_r, const void* kv_r, const void* score,
                          const int* tiles, int ntiles, int N, int N_L, int W, int S, int need_reg, float scale,
                          float sim_thresh, float conf_thresh, const float* stats, const void* dxcat_c,
                          const void* dxcat_r, const void* dsoft_c, const void* dsoft_r, int ld_dsoft, const void* soft_c,
                          const void* soft_r, int ld_soft, void* workspace, long workspace_bytes, void* dq_c, void* dkv_c,
                          void* dq_r, void* dkv_r, float* dscore, const void* fwd_rows, long fwd_rows_bytes, void* stream);

/* Several independent stride-1 GEMM-shaped convs (1x1 / nn.Linear as 1 x n images) in ONE launch of the 64 x 128 LDS-DMA tile: the
 * proposal stage's mutually independent linears (post_trans.py:640-660 q / kv projections, tscd_matching.py:31-60 q / k / v).  jobs = njobs
 * records {x, wpk, y, bias (fp32 or NULL), B, H, W, Cin, Cout, ldx, ldy, epilogue (0 raw, 1 bias, 2 bias + ReLU)}; each job needs
 * tscd_conv2d_jobs_supported(Cin, Cout); at most tscd_conv2d_jobs_max() jobs.  Same bits as one tscd_conv2d_nhwc_f16 call per job. */
typedef struct { const void* x; const void* wpk; void* y; const float* bias; int B, H, W, Cin, Cout, ldx, ldy, epilogue; } TscdConvJob;
int tscd_conv2d_jobs_max(void);
int tscd_conv2d_jobs_supported(int Cin, int Cout);
int tscd_conv2d_jobs_nhwc_f16(const void* jobs /* TscdConvJob[njobs] */, int njobs, void* stream);

/* ---- CAFM / TaskAligned pieces: PositionMHAttention / MHAttention (tscd_matching.py:31-60,140-181), SEModule
 * (:264-283), LayerNorm of forward_post (:566-589,:394-458), cosine normalisation of double_match_embds (:912-937).
 * tiles: device int[ntiles][4] = (row0, nrows, key0, key1). */
int tscd_cos_attn_f16(const void* q, const void* k, const void* v, int C, int heads, const int* tiles, int ntiles,
                      void* out, void* stream);
int tscd_se_fuse_f16(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2,
                     const float* pos, long n, int C, void* out, void* stream);
int tscd_add_layernorm_f16(const void* a, const void* b, const float* gamma, const float* beta, int n, int C, float eps,
                           void* out, void* stream);
/* Backward of the three kernels above (autograd under tscd_trainer.py:171-176).  cos_attn_bwd: gradients w.r.t. the raw
 * q/k (normalisation backward included) and v; q/dO/dq [n,C], k/v/dk/dv [nk,C]; workspace >= 8*nk*C bytes.  se_fuse_bwd / add_layernorm_bwd: parameter
 * gradients are fp32 atomic accumulators (dpos[C], dw0[32*2], dw2[2*32]; dgamma[C], dbeta[C]) that the entry points clear. */
int tscd_cos_attn_bwd_f16(const void* q, const void* k, const void* v, const void* dO, int n, int nk, int C, int heads,
                          const int* tiles, int ntiles, void* workspace, long workspace_bytes, void* dq, void* dk, void* dv,
                          void* stream);
int tscd_cos_attn_bwd_finish_f16(const void* k, const void* workspace, int nk, int C, int heads, void* dk, void* dv,
                                 void* stream);
/* The same attention on the matrix cores (attn_mfma.hip) for head dims 16..160 that are multiples of 16: one wave per
 * (32-query tile, head), tiles = (row0, nrows <= 32, key0, key1).  lsum [n][heads] fp32 receives the softmax denominators
 * the backward needs; the backward writes dq and leaves fp32 [dk^ | dv] accumulators in `workspace` (>= 8*nk*C bytes) for
 * tscd_cos_attn_bwd_finish_f16. */
int tscd_cos_attn_mfma_supported(int head_dim);
int tscd_cos_attn_mfma_f16(const void* q, const void* k, const void* v, int C, int heads, const int* tiles, int ntiles,
                           void* out, float* lsum, void* stream);
int tscd_cos_attn_mfma_bwd_f16(const void* q, const void* k, const void* v, const void* dO, const void* out, const float* lsum,
                               int n, int nk, int C, int heads, const int* tiles, int ntiles, void* workspace,
                               long workspace_bytes, void* dq, void* stream);
/* Two-launch backward for segment lists with disjoint KEY ranges (CAFM's and TaskAligned's per-frame attention,
 * tscd_matching.py:140-181): a query-stationary launch writes dq and delta = dO.O (delta_ws: n*heads floats), a key-stationary
 * launch -- one wave per (32-key tile, head), ktiles = (key row0, nkeys <= 32, query0, query1) -- writes dk and dv in their final
 * fp16 form: no atomics, no fp32 accumulators, no finish pass.  Key rows outside every ktile are not written. */
int tscd_cos_attn_mfma_bwd2_f16(const void* q, const void* k, const void* v, const void* dO, const void* out, const float* lsum,
                                int n, int nk, int C, int heads, const int* tiles, int ntiles, const int* ktiles, int nktiles,
                                float* delta_ws, void* dq, void* dk, void* dv, void* stream);
int tscd_se_fuse_bwd_f16(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2,
                         const void* dout, long n, int C, void* dreg, void* dedge, float* dpos, float* dw0, float* dw2,
                         void* stream);
int tscd_add_layernorm_bwd_f16(const void* a, const void* b, const float* gamma, const void* dy, int n, int C, float eps,
                               void* dx, float* dgamma, float* dbeta, void* stream);
/* The two backward entries above ADDING into their fp32 parameter-gradient accumulators instead of clearing them first: a backward chain
 * (the CAFM recurrence) clears one arena once and lets every frame's call add to its own part. */
int tscd_se_fuse_bwd_acc_f16(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2,
                             const void* dout, long n, int C, void* dreg, void* dedge, float* dpos, float* dw0, float* dw2,
                             void* stream);
int tscd_add_layernorm_bwd_acc_f16(const void* a, const void* b, const float* gamma, const void* dy, int n, int C, float eps,
                                   void* dx, float* dgamma, float* dbeta, void* stream);
int tscd_row_normalize_f16(const void* x, int ldx, int n, int n_pad, int C, float eps, void* out, int ldo, void* stream);
/* Every Hungarian cost matrix of a clip in one launch (double_match_embds, tscd_matching.py:912-937): pairs = npairs x (r0, nr, c0, nc)
 * row ranges of the normalised [reg | cls] rows; matrix p = 1 - rows[r0..r0+nr) . rows[c0..c0+nc)^T / 2 (NaN -> 0), fp16 row-major [nr][nc],
 * written back to back into `out`.  K % 16 == 0, ld >= K, at most 16 pairs. */
int tscd_match_cost_pairs_f16(const void* rows, int ld, int K, const int* pairs, int npairs, void* out, void* stream);
/* The CAFM frame chain's row addressing inside the four kernels above (tscd_matching.py:761-888: every frame's rows are re-ordered by
 * its assignment before they meet the next frame's; the permuted reads / writes ride in the kernels that consume / produce the rows
 * instead of index_select launches).  A row map sends row r (segment s = r / seg_rows, local l = r % seg_rows) to
 * s * seg_rows + idx[off0 + s * off_stride + l]; idx null = identity. */
typedef struct {
  const int64_t* idx;
  int seg_rows, off0, off_stride;
} TscdRowMap;
typedef struct {
  TscdRowMap edge_map;       /* edge row of output row r; backward: also the row d edge of row r is written to */
  int pos_seg_rows;          /* 0: one pos / d pos vector; else row r takes pos + (r / pos_seg_rows) * pos_stride floats */
  int pos_stride;
  const void* add_reg[3];    /* backward: d reg[r] = fp16(d reg + sum_k add_reg[k][r], r < add_reg_rows[k]) -- fp16 [rows, C] */
  int add_reg_rows[3];
  const void* add_edge[2];   /* the same for d edge */
  int add_edge_rows[2];
} TscdSeFuseEx;
typedef struct {
  TscdRowMap a_map;          /* row of `a` read for output row r */
  const void* dy_add;        /* backward: dy row r = fp16(dy[r] + dy_add[r]) */
  void* dx_mapped;           /* backward: a copy of dx row r at row a_map(r) of this buffer (dx itself may then be null) */
} TscdLnEx;
int tscd_se_fuse_ex_f16(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2,
                        const float* pos, long n, int C, void* out, const void* ex /* TscdSeFuseEx */, void* stream);
int tscd_add_layernorm_ex_f16(const void* a, const void* b, const float* gamma, const float* beta, int n, int C, float eps,
                              void* out, const void* ex /* TscdLnEx */, void* stream);
/* backward forms: ADD into the fp32 accumulators (as the _acc entries) */
int tscd_se_fuse_bwd_ex_f16(const void* reg, int ldr, const void* edge, int lde, const float* w0, const float* w2,
                            const void* dout, long n, int C, void* dreg, void* dedge, float* dpos, float* dw0, float* dw2,
                            const void* ex /* TscdSeFuseEx */, void* stream);
int tscd_add_layernorm_bwd_ex_f16(const void* a, const void* b, const float* gamma, const void* dy, int n, int C, float eps,
                                  void* dx, float* dgamma, float* dbeta, const void* ex /* TscdLnEx */, void* stream);

/* ---- eval tail: decode_reg_preds5 (tscd_head.py:914-949) + postprocess (post_process.py:9-85) +
 * torchvision.ops.batched_nms.  mode 0 = refined set, 1 = original detections, 2 = original detections with class-agnostic
 * suppression (utils/boxes.py:59-64).  No candidate or per-class capacity limits (the ImageNet-VID exp sets no upper proposal
 * clamp, exps/TSCD_VID/vid_tscd_large.py:42): capM = what the caller sized the outputs for (rows x classes in mode 0), max_rows =
 * the largest number of detection rows in one frame.  coord_trick = n > 0: frames with at most n candidates are suppressed with
 * torchvision's fp16 coordinate-offset arithmetic (reference-faithful; torchvision's own GPU rule is n = 5000), 0 = exact
 * per-class suppression in fp32 (the product's default). */
long tscd_nms_workspace_bytes(int L, int capM, int NC, int max_rows);
int tscd_batched_nms(const void* det, int ndet, const int* frame_off, int L, const void* cls_logit,
                     const void* obj_logit, const void* reg_delta, int NC, float conf_thresh, float iou_thresh, int capM,
                     int max_rows, int mode, int coord_trick, void* workspace, long workspace_bytes, void* out, int* out_idx,
                     int* out_count, void* stream);

/* Fused losses (tscd_amd/csrc/loss.hip; reference: TSCDHead.get_matcher_refined_losses yolox/models/tscd_head.py:1008-1203,
 * IOUloss yolox/models/losses.py:9-53, encode_reg_preds tscd_head.py:951-974).  Sums are returned un-normalised (the caller
 * divides by the foreground counts, device scalars); per-block partials are reduced in a fixed order (deterministic).
 * tscd_loss_dense_f16: raw [B][A][5+NC] logits, outputs [B][A][5+NC] decoded (cx,cy,w,h,..), SimOTA results fg_mask / matched_gt /
 * pred_iou [B][A], labels [B][Lmax][5] fp32 -> sums[3] = {sum over foreground of 1 - IoU^2, sum over all anchors of the
 * objectness BCE (target = foreground), sum over foreground of the class BCE (target = onehot * matched IoU)}; d_logit (or NULL)
 * [B][A][1+NC] fp16 = gradient of (sums[1] + sums[2]) w.r.t. the raw objectness | class logits.
 * tscd_loss_rows_f16: N refined rows, row n = anchor row_a[n] of frame row_b[n]; part bit 0 class BCE of cls_logit [N][NC]
 * (foreground rows, target gathered through the assignment), bit 1 BCE of obj_logit [N] (target = row is foreground), bit 2
 * SmoothL1 (beta 1) of reg_delta [N][4] against encode_reg_preds(matched box, base xyxy = columns 0..3 of base [N][ndet]) on the
 * foreground rows -> sums[3] and the gradients d_cls / d_obj / d_reg (fp16, NULL to skip). */
long tscd_loss_workspace_bytes(long B, long A, long N);
int tscd_loss_dense_f16(const void* raw, const void* outputs, const unsigned char* fg_mask, const int* matched_gt,
                        const float* pred_iou, const float* labels, int B, int A, int NC, int Lmax, void* workspace,
                        long workspace_bytes, float* sums, void* d_logit, void* stream);
/* The same plus d_reg [B][A][4] fp16 = d(IoU-loss sum) / d raw box logits through the decode (cx = (r + grid) * stride, w = exp(r) *
 * stride): base-detector training, where the box branch trains (yolo_head.py:267-434).  hw / stride: host int[3] per pyramid level. */
int tscd_loss_dense_reg_f16(const void* raw, const void* outputs, const unsigned char* fg_mask, const int* matched_gt,
                            const float* pred_iou, const float* labels, int B, int A, int NC, int Lmax, const int* hw, const int* stride,
                            void* workspace, long workspace_bytes, float* sums, void* d_logit, void* d_reg, void* stream);
int tscd_loss_rows_f16(const int* row_b, const int* row_a, int N, int A, int NC, int Lmax, const unsigned char* fg_mask,
                       const int* matched_gt, const float* pred_iou, const float* labels, const void* cls_logit,
                       const void* obj_logit, const void* reg_delta, const void* base, int ndet, int part, void* workspace,
                       long workspace_bytes, float* sums, void* d_cls, void* d_obj, void* d_reg, void* stream);

/* The scalar tail of get_matcher_refined_losses (tscd_head.py:1157-1203) in one launch: dense[3] / rows[3] loss sums, r_cls (device
 * scalar), num_fg / ngt int[B] -> out[8] = {total, 3 iou, obj, cls, num_fg / max(num_gts, 1), r_cls, matched obj (clipped at 15),
 * 6 matched iou} and weights[6] = d total / d {dense[0..2], rows[0..2]}. */
int tscd_loss_combine_f32(const float* dense, const float* rows, const float* r_cls, const int* num_fg, const int* ngt, int B,
                          int lframe, float* out, float* weights, void* stream);

/* Backward of the fused losses: out[r][pad + c] = fp16(float(d[r][c]) * g[c < split ? i0 : i1]), columns [0, pad) zero.  g: fp32
 * device scalars (loss scale x weight / num_fg): the product is formed in fp32 and rounded once, as the reference's autograd does
 * (tscd_head.py:1157-1203). */
int tscd_scale_grad_f16(const void* d, const float* g, int i0, int i1, int split, long rows, int C, int pad, void* out, void* stream);

/* Class score of every selected proposal and its backward: out[r] = sigmoid(raw[row_b[r]][row_a[r]][5 + class(r)]), class(r) =
 * pred_rows[r][cls_col] (the class id column of the proposal rows, tscd_head.py:1560-1600: pred_result[:, 5] with its gradient kept);
 * the scatter clears d_raw [B][A][nch] and stores g * s * (1 - s) at the same positions (rows are distinct (frame, anchor) pairs).
 * Rounding as torch's fp16 sigmoid / sigmoid_backward (fp32 inside, one rounding at the store). */
int tscd_cls_score_gather_f16(const void* raw, int B, long A, int nch, const int* row_b, const int* row_a, const void* pred_rows,
                              int ld_rows, int cls_col, long N, void* out, void* stream);
int tscd_cls_score_scatter_f16(const void* g, const void* s, int B, long A, int nch, const int* row_b, const int* row_a,
                               const void* pred_rows, int ld_rows, int cls_col, long N, void* d_raw, void* stream);
/* Gradient of one pyramid level's raw prediction map ([B][HW][nch] fp16, a slice of the clip's [B][A][nch] gradient: batch stride in
 * elements) split for the two trainable prediction convs (tscd_head.py:213-239): dY zero-padded to 8 (objectness) / ncp (classes,
 * NC rounded up to 8) channels for the weight-gradient GEMM, and db[4], db[5..] = their bias gradients (column sums, fixed-order
 * two-stage reduction). */
long tscd_pred_grad_split_workspace_bytes(long M, int nch);
int tscd_pred_grad_split_f16(const void* d_raw, long batch_stride, int B, long HW, int nch, int ncp, void* dy_obj, void* dy_cls, float* db,
                             void* workspace, long workspace_bytes, void* stream);

/* ---- SimOTA label assignment: get_fg_idx / get_assignments / get_in_boxes_info / dynamic_k_matching
 * (tscd_head.py:1712-1851,1330-1544), bboxes_iou (yolox/utils/boxes.py:131-154). */
long tscd_simota_workspace_bytes(int B, int A, int G, int topk);
int tscd_simota_assign(const void* outputs, const float* labels, int B, int A, int NC, int Lmax, int G, const int* hw3,
                       const int* w3, const int* stride3, int topk, int emu16, void* workspace, long workspace_bytes,
                       unsigned char* fg_mask, int* matched_gt, float* pred_iou, int* num_fg, int* ngt, void* stream);
/* The same with the centre-prior radius (in strides) as an argument: 4.5 in the TSCD head (what tscd_simota_assign uses), 2.5 in the
 * YOLOX base head (yolo_head.py:594). */
int tscd_simota_assign_r(const void* outputs, const float* labels, int B, int A, int NC, int Lmax, int G, const int* hw3, const int* w3,
                         const int* stride3, int topk, int emu16, float center_radius, void* workspace, long workspace_bytes,
                         unsigned char* fg_mask, int* matched_gt, float* pred_iou, int* num_fg, int* ngt, void* stream);

/* Backbone / neck backward helpers.  zero_insert2: dY [B,Ho,Wo,C] -> [B,2Ho,2Wo,C], dY at even positions: the
 * data / weight gradient of a 3x3 stride-2 pad-1 conv is the stride-1 gradient on it (darknet.py:120-160 downsampling convs,
 * yolo_pafpn.py bu_conv1/2).  maxpool5_bwd: gradient of SPP's stride-1 5x5 max pool (network_blocks.py:183-205), first maximum in scan
 * order wins ties.  upsample2_bwd: gradient of PAFPN's nearest x2 upsample (sum of each 2x2 block). */
int tscd_zero_insert2_nhwc_f16(const void* dy, int lddy, void* out, int ldo, int B, int Ho, int Wo, int C, void* stream);
int tscd_maxpool5_bwd_nhwc_f16(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int B, int H, int W, int C, void* stream);
int tscd_upsample2_bwd_nhwc_f16(const void* dy, int lddy, void* dx, int lddx, int B, int H, int W, int C, void* stream);

/* Clip pre-processing on the device (reference: yolox/data/datasets/vid.py pull_item + data_augment.py:21-31,
 * 196-203, 248-264): decoded uint8 BGR frames [B][hmax][wmax][3] -> letterboxed planar [B][3][H][W] (0..255; fp16 or fp32): bilinear
 * resize to (nh, nw) in OpenCV's INTER_LINEAR geometry, optional 8-bit HSV jitter, optional horizontal flip, 114 fill.
 * meta: device int32 [B][12] = {h, w, nh, nw, flip, hsv_on, hgain, sgain, vgain, bits of fp32(h/nh), bits of fp32(w/nw), 0}. */
int tscd_clip_preproc_u8(const void* src, int B, int hmax, int wmax, const int* meta, void* out, int H, int W, int out_is_f16,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TSCD_HIP_H */
