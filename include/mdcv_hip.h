/* libmdcv_hip.so — C ABI of the MI355X (gfx950) hot path of CVC-YOLOv3 + RektNet training.
 *
 * The reference (cv-core/MIT-Driverless-CV-TrainingInfra) has NO native/FFI layer: its hot path is Python calling
 * stock torch ops.  This ABI is therefore the boundary the reference *would* bind if it had one; each entry point
 * names the reference call site(s) it replaces.  Conventions:
 *   - plain C symbols, plain pointers and sizes; the CALLER owns every buffer (device memory), kernels allocate nothing
 *   - enqueue-only on the caller's HIP stream (`stream` = hipStream_t as void*), no synchronisation inside
 *   - return 0 on success, a negative MDCV_E* for argument errors, or a positive hipError_t
 *   - dtype: 0 = fp32 (parity mode, exact-f32 MFMA), 1 = bf16 (production, fp32 accumulate)
 *   - activations are NHWC with an explicit channel stride `ld*` (elements); channel counts are padded to a multiple
 *     of 8 and pad channels hold exact zeros; weights/gradients at the boundary are OIHW fp32 like torch parameters
 */
#ifndef MDCV_HIP_H
#define MDCV_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define MDCV_OK 0
#define MDCV_EARG (-1)
#define MDCV_F32 0
#define MDCV_BF16 1
/* The library holds NO mutable process state (round 5): a call's result depends on its arguments only.  The dispatch heuristics that were A/B-ed
 * on the training step (csrc/tune.h: one documented knob each) can be overridden PER CALL through the dtype argument of the convolution and
 * weight-gradient entry points: bits 0..7 = MDCV_F32 / MDCV_BF16, the bits above = a signed variant code of that entry point's family (0 =
 * defaults; the code tables are in csrc/tune.h).  Plan-time queries (mdcv_conv2d_stats_rows_geom, mdcv_conv2d_wgrad_splits_geom, ...) take the same
 * tuned dtype as the launch they size buffers for. */
#define MDCV_TUNED(dtype, code) (((dtype) & 0xff) | ((code) * 256))
#define MDCV_ACT_NONE 0
#define MDCV_ACT_LEAKY 1
#define MDCV_ACT_RELU 2

/* ---- convolution (nn.Conv2d fwd/bwd: CVC-YOLOv3/models.py:59-65 ; RektNet/keypoint_net.py:17,25 ; RektNet/resnet.py:12-19)
 * mode 0: forward.  in=[B,Hin,Win,Cin], out=[B,Hout,Wout,Nout], w_packed=[Nout][KH*KW][Cin] (from mdcv_pack_weights w_fwd).
 * mode 1: data gradient.  in=dY [B,Hin,Win,Cin=Cout_pad] on the conv's OUTPUT grid, out=dX on the conv's INPUT grid
 *         [B,Hout,Wout,Nout=Cin_pad], w_packed=[Cin_pad][KH*KW][Cout_pad] (w_dgrad).  stride in {1,2}.
 * bias (fp32[Nout]) is added before statistics; addsrc (same layout as out) is added after (residual / fan-out grads);
 * stats_partial, if given, receives [mdcv_conv2d_stats_rows(M)][2][Nout] fp32 = per-tile (sum, sum^2) per channel. */
int mdcv_conv2d(int dtype, int mode, const void* in, int in_ldc, const void* w_packed, void* out, int out_ldc,
                const float* bias, const void* addsrc, int add_ldc, float* stats_partial,
                int B, int Hin, int Win, int Cin, int Hout, int Wout, int Nout,
                int KH, int KW, int stride, int pad, int dil, void* stream);
/* Inference forward (Darknet.forward / KeypointNet.forward in eval mode: conv -> BatchNorm(running stats) -> activation, models.py:48-72,
 * resnet.py:22-27): out = act(conv(in) * scale[n] + shift[n]) (+ addsrc, e.g. the shortcut input), scale/shift from mdcv_bn_eval_coeffs
 * (or scale = NULL: bias only).  act: MDCV_ACT_*.  The raw conv output never goes to HBM. */
int mdcv_conv2d_affine_act(int dtype, const void* in, int in_ldc, const void* w_packed, void* out, int out_ldc, const float* scale,
                           const float* shift, const void* addsrc, int add_ldc, int act, float slope, int B, int Hin, int Win, int Cin,
                           int Hout, int Wout, int Nout, int KH, int KW, int stride, int pad, int dil, void* stream);
/* Data gradient (mode 1, same geometry arguments as mdcv_conv2d) that also writes the BatchNorm-backward partial sums of the layer
 * whose output gradient it produces: partial[row][0][c] = sum g, partial[row][1][c] = sum g*(y - mean), g = dz * act'(scale*y + shift),
 * one row per 128 output positions (y: that layer's raw conv output, same pixel/channel indexing as the gradient written to `out`).
 * _rows() = number of rows written, or 0 when the geometry / dtype (bf16 only) cannot take the fused path.  Finish with mdcv_bn_bwd_finalize_rows. */
int mdcv_conv2d_dgrad_bnsums_rows(int dtype, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Nout, int KH, int KW, int stride,
                                  int pad, int dil, int in_ldc);
int mdcv_conv2d_dgrad_bnsums(int dtype, const void* in, int in_ldc, const void* w_packed, void* out, int out_ldc, const void* addsrc,
                             int add_ldc, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Nout, int KH, int KW, int stride,
                             int pad, int dil, const void* y, int ldy, const float* scale, const float* shift, const float* mean, int act,
                             float slope, float* partial, void* stream);
/* The network's FIRST conv -> BatchNorm(batch statistics) -> activation (CVC-YOLOv3/models.py:57-71 at index 0; 3 -> 32 channels, 3x3 / stride 1 / pad 1) as two
 * streaming launches that never re-read the layer's 354 MB output: the conv is 25 GFLOP on an 88 MB input, so it is computed twice.  _stats: x -> partial rows
 * [mdcv_first_conv_rows(B, H)][2][32] (sum, sum of squares of the fp32 accumulators; finish with mdcv_bn_stats_finalize).  _bn_act: x -> y (raw conv output, bf16, kept
 * for the backward) AND z = act(scale * y + shift) from the y it stores, in one store loop.  _ok: 1 where the geometry takes this form (bf16, 8 padded input
 * channels at stride 8, 32 output channels); elsewhere: mdcv_conv2d + mdcv_bn_act_fwd. */
int mdcv_first_conv_ok(int dtype, int B, int H, int W, int Cin_pad, int Cout_pad, int KH, int KW, int stride, int pad, int dil, int ldx);
int mdcv_first_conv_rows(int B, int H);
int mdcv_first_conv_stats(int dtype, const void* x, int ldx, const void* w_packed, float* partial, int B, int H, int W, void* stream);
int mdcv_first_conv_bn_act(int dtype, const void* x, int ldx, const void* w_packed, const float* scale, const float* shift, int act, float slope,
                           void* y, int ldy, void* z, int ldz, int B, int H, int W, void* stream);
/* 1 when the data gradient of this geometry runs as the stride-2 form of the 3x3 shift kernel (bf16, 3x3 / stride 2 / pad 1, Hout = 2 Hin, 32 or 64
 * output channels): its store loop writes whole output rows from LDS and carries the fused sums at every size (one partial row per 8 x 31 tile). */
int mdcv_conv2d_dgrad_s2_form_ok(int dtype, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Nout, int KH, int KW, int stride,
                                 int pad, int dil, int in_ldc);
/* Forward statistics through EXACT ACCUMULATORS (csrc/exact_acc.h): the conv's epilogue adds its per-tile sums (per output channel: sum, sum of
 * squares) to xacc = [reps][3][2][Nout] signed 64-bit words (mdcv_xstats_words; ZERO before the launch) with fire-and-forget integer atomics --
 * three 40-bit digits of a fixed-point number with quantum 2^-70, so every addition is exact and the totals do not depend on the order the
 * atomics land in (bit-reproducible, unlike float atomics).  reps (a power of two; mdcv_xstats_reps(rows, C) for a layer that performs `rows`
 * additions per channel, rows = mdcv_conv2d_stats_rows_geom / mdcv_pw_rows) spreads a word's additions over replicas.  Every forward kernel of
 * mdcv_conv2d carries it (both dtypes), and mdcv_pw_conv_fwd_xstats is mdcv_pw_conv_fwd with the same sink.  Consumer: mdcv_bn_act_fwd_xstats --
 * BatchNorm(batch statistics) + activation (+ residual) whose prologue adds the replicas and forms the statistics; it writes scale / shift /
 * mean / invstd / running statistics as mdcv_bn_stats_finalize does (C <= 1024).  Replaces conv -> mdcv_bn_stats_finalize -> mdcv_bn_act_fwd of
 * the reference's nn.Sequential(conv, BatchNorm2d, LeakyReLU) (CVC-YOLOv3/models.py:57-71) by two launches with no hand-off inside a launch. */
int mdcv_xstats_words(int reps, int C);
int mdcv_xstats_reps(int rows, int C);
int mdcv_conv2d_xstats(int dtype, const void* in, int in_ldc, const void* w_packed, void* out, int out_ldc, const float* bias, void* xacc,
                       int reps, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Nout, int KH, int KW, int stride, int pad, int dil,
                       void* stream);
int mdcv_pw_conv_fwd_xstats(int dtype, const void* y, int ldy, const float* scale, const float* shift, const void* resid, int ldr, int act,
                            float slope, void* z_out, int ldz, const void* w_packed, const float* bias, void* out, int out_ldc, void* xacc,
                            int reps, long long M, int K, int N, void* stream);
int mdcv_bn_act_fwd_xstats(int dtype, const void* y, int ldy, const void* xacc, int reps, double count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                           float* invstd, const void* resid, int ldr, void* out, int ldo, int M, int C, int act, float slope, void* stream);
int mdcv_conv2d_stats_rows(int M);      /* generic kernels: one row per 128 output pixels */
/* rows of stats_partial a FORWARD launch with this geometry writes (use this one to size the buffer: the 3x3 / stride-1 /
 * pad-1 shift kernel walks a padded pixel stream and writes more rows than M / 128; every row it returns is written). */
int mdcv_conv2d_stats_rows_geom(int dtype, int B, int Hout, int Wout, int Cin, int Nout, int KH, int KW, int stride, int pad, int dil,
                                int in_ldc);

/* weight gradient: dW (OIHW fp32, real channel counts) = dY^T * im2col(X).  ws = splits*Cout*KH*KW*Cin floats of scratch. */
int mdcv_conv2d_wgrad_splits(int dtype, int M, int Cout, int Ktot);
/* number of fp32 slabs for the kernel mdcv_conv2d_wgrad picks for this geometry (3x3 stride-1 layers with 128-multiple channel
 * counts run a kernel whose kw taps share one activation tile and that wants its own split); ws = splits*Cout*KH*KW*Cin floats. */
int mdcv_conv2d_wgrad_splits_geom(int dtype, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int KH, int KW, int stride,
                                  int pad, int dil, int dy_ldc, int x_ldc);
int mdcv_conv2d_wgrad(int dtype, const void* dy, int dy_ldc, const void* x, int x_ldc, float* ws, int splits,
                      float* dw_oihw, int accumulate, int B, int Hin, int Win, int Cin, int Cin_real,
                      int Hout, int Wout, int Cout, int Cout_real, int KH, int KW, int stride, int pad, int dil, void* stream);

/* OIHW fp32 parameters -> GEMM operand layouts (w_dgrad may be NULL) */
int mdcv_pack_weights(int dtype, const float* w_oihw, void* w_fwd, void* w_dgrad, int Cout, int Cin, int KH, int KW,
                      int Cout_pad, int Cin_pad, void* stream);

/* all convs of a network in ONE launch; table = nlayers device-resident 72-byte records
 *   { const float* w_oihw; void* w_fwd; void* w_dgrad|NULL; int Cout, Cin, KH*KW, Cout_pad, Cin_pad; int reserved[3];
 *     const float* bias|NULL; float* bias_padded; }   (bias: the fp32 bias parameter, copied into the padded operand buffer) */
int mdcv_pack_weights_batched(int dtype, const void* table, int nlayers, int max_taps, void* stream);   /* max_taps = max KH*KW (times Cin_pad/64 scaling is internal) */

/* ---- layout conversion at the API edge (reference tensors are NCHW fp32: train.py:60, train_eval.py:60) */
int mdcv_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int H, int W, int ldc, int Cpad, void* stream);
int mdcv_nhwc_to_nchw(int dtype, const void* src, int ldc, float* dst, int B, int C, int H, int W, void* stream);

/* ---- BatchNorm2d (eps 1e-5, momentum 0.1) + LeakyReLU/ReLU + shortcut add (models.py:66-71,325-327 ; resnet.py:22-27) */
int mdcv_partial_reduce(const float* partial, int rows, int nsums, int C, double* accum, void* stream);
int mdcv_bn_finalize(double* accum, double count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     float momentum, float eps, float* scale, float* shift, float* mean, float* invstd, int C, void* stream);
int mdcv_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                        float* scale, float* shift, int C, void* stream);
/* the same for a conv that has its own bias in front of the BatchNorm (RektNet): shift also absorbs conv_bias * scale, so that
 * mdcv_conv2d_affine_act(scale, shift) == BatchNorm_eval(conv + bias).  conv_bias may be NULL. */
int mdcv_bn_eval_coeffs_bias(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                             const float* conv_bias, float* scale, float* shift, int C, void* stream);
/* out = act(y1*s1+b1 [+ y2*s2+b2]) [+ resid] */
int mdcv_bn_act_fwd(int dtype, const void* y1, int ld1, const float* s1, const float* b1, const void* y2, int ld2, const float* s2,
                    const float* b2, const void* resid, int ldr, void* out, int ldo, int M, int C, int act, float slope, void* stream);
int mdcv_bn_act_bwd_reduce_ws_floats(int dtype, int M, int C, int nsums);
/* partial rows -> statistics -> scale/shift (+ running stats) in ONE launch while rows <= 4096 (a workgroup owns 16 channels and
 * sums their rows itself); larger buffers are first folded to 64 rows IN PLACE by one more launch (no atomics; `partial` is the
 * caller's scratch and is consumed).  `accum` is unused (kept for the two-stage entry points mdcv_partial_reduce / mdcv_bn_finalize). */
int mdcv_bn_stats_finalize(const float* partial, int rows, double* accum, double count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                           float* invstd, int C, void* stream);
/* mdcv_bn_act_bwd_reduce + mdcv_bn_bwd_finalize for one BatchNorm (y2 == NULL) or the fused residual pair, two launches in all. */
int mdcv_bn_act_bwd_reduce_finalize(int dtype, const void* dout, int ldd, const void* y1, int ld1, const float* s1, const float* b1,
                                    const float* mean1, const float* invstd1, const void* y2, int ld2, const float* s2, const float* b2,
                                    const float* mean2, const float* invstd2, float* partial_ws, int M, int C, int act, float slope,
                                    double count, const float* gamma1, float* dgamma1, float* dbeta1, float* cA1, float* cB1, float* cC1,
                                    const float* gamma2, float* dgamma2, float* dbeta2, float* cA2, float* cB2, float* cC2, void* stream);
/* finalize for the fused data-gradient sums: rows x [2][C] partials (sum g, sum g*(y-mean)) -> dgamma, dbeta, cA, cB, cC
 * (more than 4096 rows are folded in place first, as above: `partial` is consumed) */
int mdcv_bn_bwd_finalize_rows(const float* partial, int rows, int C, double count, const float* gamma, const float* mean,
                              const float* invstd, float* dgamma, float* dbeta, float* cA, float* cB, float* cC, void* stream);
int mdcv_bn_act_bwd_reduce(int dtype, const void* dout, int ldd, const void* y1, int ld1, const float* s1, const float* b1,
                           const float* mean1, const float* invstd1, const void* y2, int ld2, const float* s2, const float* b2,
                           const float* mean2, const float* invstd2, double* accum, float* partial_ws, int M, int C, int act, float slope,
                           void* stream);
int mdcv_bn_bwd_finalize(double* accum, int kx, int nsums, int zero_after, double count, const float* gamma, const float* mean,
                         const float* invstd, float* dgamma, float* dbeta, float* cA, float* cB, float* cC, int C, void* stream);
int mdcv_bn_act_bwd_apply(int dtype, const void* dout, int ldd, const void* y1, int ld1, const float* s1, const float* b1,
                          const float* cA1, const float* cB1, const float* cC1, void* dy1, int ldy1,
                          const void* y2, int ld2, const float* s2, const float* b2, const float* cA2, const float* cB2,
                          const float* cC2, void* dy2, int ldy2, int M, int C, int act, float slope, void* stream);
int mdcv_colsum(int dtype, const void* x, int ldc, int M, int C, double* accum, void* stream);
/* the same sums without atomics: per-block partial rows in partial_ws (mdcv_colsum_ws_floats floats), then one reduce launch -> out[C] */
int mdcv_colsum_ws_floats(int dtype, int M, int C);
int mdcv_colsum_f32(int dtype, const void* x, int ldc, int M, int C, float* partial_ws, float* out, void* stream);
int mdcv_accum_to_f32(double* accum, float* out, int n, int zero_after, void* stream);

/* ---- nn.Upsample(scale 2, nearest) fwd/bwd (models.py:86-88); H,W are the LOW-resolution dims */
int mdcv_upsample2x_fwd(int dtype, const void* in, int ldi, void* out, int ldo, int B, int H, int W, int C, void* stream);
int mdcv_upsample2x_bwd(int dtype, const void* dout, int ldo, void* din, int ldi, int B, int H, int W, int C, void* stream);

/* ---- nn.MaxPool2d(2,2) and nn.ZeroPad2d((0,1,0,1)) + nn.MaxPool2d(2,1) of yolo_baseline_tiny.cfg (models.py:74-84).
 * H,W = input dims; output is H/2 x W/2 (stride 2) or H x W (stride 1); idx [B,Ho,Wo,C] bytes = winning window slot for backward */
int mdcv_maxpool2x2_fwd(int dtype, const void* in, int ldi, void* out, int ldo, unsigned char* idx, int B, int H, int W, int C, int stride,
                        void* stream);
int mdcv_maxpool2x2_bwd(int dtype, const void* dout, int ldo, const unsigned char* idx, void* din, int ldi, int B, int H, int W, int C, int stride,
                        void* stream);

/* generic forms of the two (models.py:74-88 builds nn.MaxPool2d(size, stride, (size - 1) // 2) and nn.Upsample(scale_factor = stride) for any
 * size / stride): window k <= 15, -inf padding (pad < k), Ho = (H + 2*pad - k) / stride + 1; idx = kh*k + kw of the first maximum;
 * backward gathers over the windows that contain an input pixel (no atomics).  H, W = input dims. */
int mdcv_maxpool_fwd(int dtype, const void* in, int ldi, void* out, int ldo, unsigned char* idx, int B, int H, int W, int C, int k, int stride,
                     int pad, void* stream);
int mdcv_maxpool_bwd(int dtype, const void* dout, int ldo, const unsigned char* idx, void* din, int ldi, int B, int H, int W, int C, int k, int stride,
                     int pad, void* stream);
int mdcv_upsample_fwd(int dtype, const void* in, int ldi, void* out, int ldo, int B, int H, int W, int C, int scale, void* stream);
int mdcv_upsample_bwd(int dtype, const void* dout, int ldo, void* din, int ldi, int B, int H, int W, int C, int scale, void* stream);

/* ---- YOLOLayer.forward (models.py:140-220) + build_targets / bbox_iou (utils/utils.py:163-275)
 * logits NHWC, channel = a*(5+C)+attr.  anchors_scaled = anchors/stride, fp32 [A][2].  targets fp32 [B][T][5].
 * train: out7[0] += loss, out7[1..6] += (x,y,w,h,obj,noobj) parts; dlogits = d loss / d logits (* *gscale if given). */
long long mdcv_yolo_head_workspace_bytes(int B, int A, int Gh, int Gw);
int mdcv_yolo_head_train(int dtype, const void* logits, int ldc, void* dlogits, int ldd, int Cpad, const float* targets,
                         const float* anchors_scaled, int B, int T, int A, int C, int Gh, int Gw, float thresh, float xy_loss,
                         float wh_loss, float obj_loss, float noobj_loss, void* workspace, float* out7, const float* gscale,
                         void* stream);
int mdcv_yolo_head_grad(int dtype, const void* logits, int ldc, void* dlogits, int ldd, int Cpad, const float* targets,
                        const float* anchors_scaled, int B, int T, int A, int C, int Gh, int Gw, float thresh, float xy_loss,
                        float wh_loss, float obj_loss, float noobj_loss, void* workspace, const float* gscale, void* stream);
int mdcv_yolo_head_decode(int dtype, const void* logits, int ldc, const float* anchors_scaled, float stride, int B, int A, int C, int Gh,
                          int Gw, float* out, int rows_total, int row_off, void* stream);
/* utils.bbox_iou (utils/utils.py:163-193, "+1 pixel" convention) for n = max(n1, n2) box pairs; a side with ONE row is broadcast.  Rows are `stride`
 * floats apart, the first four are the box (corners = 1: x1 y1 x2 y2; 0: cx cy w h).  fp32; every operation rounds where the reference's torch op
 * rounds, NaN propagates as torch.max / torch.min / clamp propagate it: bit-identical to the reference. */
int mdcv_bbox_iou(const float* box1, long long n1, int stride1, const float* box2, long long n2, int stride2, int corners, float* out, void* stream);
long long mdcv_build_targets_workspace_bytes(int B, int T, int A, int Gh, int Gw);
int mdcv_build_targets(const float* targets, const float* anchors, int B, int T, int A, int C, int Gh, int Gw, float thresh,
                       unsigned char* mask, unsigned char* conf_mask, float* tx, float* ty, float* tw, float* th, float* tconf,
                       unsigned char* tcls, void* workspace, int* err_out, void* stream);

/* ---- KeypointNet head (keypoint_net.py:46-56,68-70) and CrossRatioLoss (cross_ratio_loss.py:20-63) */
/* the head's 1x1 conv (keypoint_net.py:40,68: `self.out`, Conv2d(128, 7, 1)) with fp32 logits out of bf16 features: x [M][ldx] bf16 NHWC (C % 32 == 0,
 * C <= 1024), w = the fp32 weights [K][C] as the module holds them, bias [K] or NULL, out fp32 [M][8] (columns >= K are zeros), K <= 8 */
int mdcv_head1x1_f32(const void* x_bf16, int ldx, const float* w, const float* bias, float* out, long long M, int C, int K, void* stream);
int mdcv_softargmax_fwd(int dtype, const void* logits, int ldc, int B, int K, int H, int W, float* hm, float* pts, void* stream);
int mdcv_softargmax_bwd(int dtype, const float* hm, const float* pts, const float* dpts, const float* dhm, float* sdot_ws, int B, int K,
                        int H, int W, void* dlogits, int ldd, void* stream);
int mdcv_cross_ratio_loss(const float* hm, const float* pts, const float* thm, const float* tpts, int B, int H, int W, int loss_type,
                          int include_geo, float gamma_horz, float gamma_vert, double* acc_ws, const float* gscale, float* out3,
                          float* dpts, float* dhm, void* stream);

/* ---- detection post-processing: replaces the per-image Python loop validate.py:80-141, the sequential
 *      greedy NMS utils/nms.py:4-61 and average_precision/compute_ap utils/utils.py:58-119.
 *      Visiting order: descending score, equal scores by descending index (the reference's ascending sort walked from the
 *      back, nms.py:25-32; its CPU sort is unstable past 16 elements, so for ties this is the stable-sort behaviour).
 *      top_k <= MDCV_NMS_MAX_TOPK (the reference always uses 200, nms.py:4 / validate.py:93); larger is MDCV_EARG. */
#define MDCV_NMS_MAX_TOPK 512
long long mdcv_nms_workspace_bytes(int n);
/* boxes [n,4] corner format, scores [n]; keep[0..*count) <- kept indices in visiting order (keep has room for min(n,top_k));
 * count is a device int.  n == 0 gives *count = 0 (nms.py:17-18). */
int mdcv_nms(const float* boxes, const float* scores, int n, float overlap, int top_k, long long* keep, int* count, void* workspace,
             void* stream);
long long mdcv_detect_post_workspace_bytes(int B, int N);
/* pred [B,N,5+C] eval-mode rows (cx,cy,w,h,conf,cls...), targets [B,T,5] zero-padded (cls,cx,cy,w,h) or NULL with T = 0.
 * Per image b: out_count[b] kept detections in descending confidence; out_boxes [B,top_k,4] corner, out_prob, out_cls (first
 * argmax), out_index (row in N), out_correct [B,top_k]; out_stats [B,4] = (AP, recall, precision, valid) with valid = 0 where the
 * reference loop `continue`s (no detection kept, validate.py:97, or no real label, validate.py:120). */
int mdcv_detect_post(const float* pred, int B, int N, int C, const float* targets, int T, float conf_thres, float nms_thres,
                     float iou_thres, float width, float height, int top_k, float* out_boxes, float* out_prob, int* out_cls,
                     long long* out_index, unsigned char* out_correct, int* out_count, float* out_stats, void* workspace,
                     void* stream);
/* utils.py:58-88 on its own: tp u8[m], conf f32[m], 1 <= m <= MDCV_NMS_MAX_TOPK; out3 = (AP, recall, precision). */
int mdcv_average_precision(const unsigned char* tp, const float* conf, int m, int n_gt, float* out3, void* stream);

/* ---- detect -> crop -> keypoints glue.  Cuts box k < count[b] of frame b out of frames [B,C,H,W] (fp32) and
 *      resamples it to out_h x out_w with cv2.resize's default bilinear rule (RektNet/utils.py:73-76 prep_image; layout of
 *      RektNet/detect.py:33-35: [M,C,out_h,out_w]).  Boxes [B,K,4] are corner boxes in detector coordinates and are mapped to
 *      frame pixels as x * scale_x + off_x (CVC-YOLOv3/detect.py:98-101), rounded outwards and clamped to the frame.
 *      Crops are packed image-major: out row m, owner[m] = b, *total = M (device ints; out / owner hold B*K rows).
 *      out_h, out_w <= 256. */
int mdcv_crop_resize(const float* frames, int B, int C, int H, int W, const float* boxes, const int* count, int K, float scale_x,
                     float scale_y, float off_x, float off_y, int out_h, int out_w, float* out, int* owner, int* total, void* stream);

/* The same cut with the rule the reference's loaders actually apply (RektNet/dataset.py:35-38,52, RektNet/detect.py:29-35): the frame
 * is a uint8 image [B,C,H,W], cv2.resize runs its 8-bit fixed-point INTER_LINEAR (11-bit coefficients, result rounded to 8 bits), and
 * only then is the crop divided by 255: out = (float)(u8 / 255.0). */
int mdcv_crop_resize_u8(const unsigned char* frames, int B, int C, int H, int W, const float* boxes, const int* count, int K, float scale_x,
                        float scale_y, float off_x, float off_y, int out_h, int out_w, float* out, int* owner, int* total, void* stream);

/* ---- on-device synthetic cone data with the output contracts of the reference's datasets; every value is a pure
 *      function of (seed, step, index), reproduced bit for bit by oracle/synth_oracle.py.
 *      detector batch (CVC-YOLOv3/utils/datasets.py:124-315): images [B,3,H,W] in [0,1], targets [B,T,5] (cls,cx,cy,w,h), zero rows last.
 *      crop batch (RektNet/dataset.py:34-56, RektNet/utils.py:83-111): images [B,3,80,80], heat-maps [B,7,80,80], points [B,7,2]. */
int mdcv_synth_cone_batch(unsigned int seed, int step, int B, int T, int H, int W, int num_classes, float* images, float* targets, void* stream);
int mdcv_synth_crop_batch(unsigned int seed, int step, int B, int size, float* images, float* heatmaps, float* points, void* stream);

/* ---- optimizer step over the flat fp32 parameter buffer (train.py:180-187,72 ; train_eval.py:263,72) */
int mdcv_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int step, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float grad_scale, void* stream);
int mdcv_sgd_step(float* params, const float* grads, float* momentum_buf, long long n, int step, float lr, float momentum, float weight_decay,
                  float grad_scale, void* stream);

/* ---- runtime plumbing: device query, HIP events on a caller stream, hipGraph capture of a launch sequence */
int mdcv_device_info(int* cu_count, int* wave_size, long long* hbm_bytes, char* arch, int arch_len);
int mdcv_event_create(void** ev);
int mdcv_event_record(void* ev, void* stream);
int mdcv_event_sync(void* ev);
int mdcv_event_elapsed_ms(void* start, void* stop, float* ms);
int mdcv_event_destroy(void* ev);
/* everything enqueued on `from` so far happens before what is enqueued on `to` afterwards (one event record + one stream wait on a
 * ring event without timing; device_scope != 0: the record releases to device scope, all a consumer on the same GPU needs) */
/* On-box peak probe: `blocks` workgroups x 4 waves x iters x 8 independent v_mfma_f32_16x16x32_bf16 on register operands; *flops = the FLOP
 * count of the launch.  Time it with events on `stream`: the dense bf16 MFMA rate this box sustains (spec: ~2.5 PFLOP/s).  sink: >= 1 float. */
int mdcv_probe_mfma(int blocks, int iters, float* sink, double* flops, void* stream);
int mdcv_stream_fork(void* from, void* to, int device_scope);
/* the same ordering without a marker packet in the producer's queue: the next kernel this library launches (exactly one, on `from`) carries
 * the event as its dispatch packet's stop event; mdcv_stream_fork_wait(to, ev) then makes `to` wait for it */
int mdcv_stream_fork_arm(void* from, int device_scope, void** ev_out);
int mdcv_stream_fork_wait(void* to, void* ev);
/* In-library kernel profiler: between _begin and _stop EVERY kernel this library launches carries a start / stop HIP event pair bound
 * to its own dispatch on the stream it is launched on (hipExtLaunchKernel: no extra packets, the durations are the kernel's own).  _count = launches recorded so far (lets a host attribute them to its own calls); _stop ends recording,
 * waits for the events and returns the record count; _read(i) -> duration in ms and the kernel symbol as rocprofv3 prints it.
 * (bench.py's `roofline` / `roofline_kernels`; not meant for timed regions: each record costs two event records.) */
int mdcv_profile_begin(void);
int mdcv_profile_count(void);
int mdcv_profile_stop(void);
int mdcv_profile_read(int i, float* ms, char* name, int name_len);
int mdcv_graph_begin(void* stream);
int mdcv_graph_end(void* stream, void** graph_exec);
int mdcv_graph_launch(void* graph_exec, void* stream);
int mdcv_graph_destroy(void* graph_exec);

/* ---- 1x1 convolution blocks with the neighbouring BatchNorm pass folded into the operand load (csrc/pw_block.hip; bf16 only).
 *      Replaces, for a 1x1 conv whose input is the output of a conv -> BatchNorm -> activation (-> shortcut add) block
 *      (CVC-YOLOv3/models.py:48-72 + the shortcut of :322-327), the pair  mdcv_bn_act_fwd + mdcv_conv2d  by one launch with identical results.
 *      K = channels of the transformed operand (multiple of 32, K/8 divides 256, 64 <= K <= 1024), N = output channels (multiple of 8). */
int mdcv_pw_rows(long long M, int K);          /* rows of stats_partial the entry points below write: one per pixel tile */
/* forward: z = act(y * scale + shift) (+ resid) -> z_out ; out = z . W^T (+ bias) ; stats_partial (may be NULL): [mdcv_pw_rows][2][N] */
int mdcv_pw_conv_fwd(int dtype, const void* y, int ldy, const float* scale, const float* shift, const void* resid, int ldr, int act,
                     float slope, void* z_out, int ldz, const void* w_packed, const float* bias, void* out, int out_ldc,
                     float* stats_partial, long long M, int K, int N, void* stream);

/* ---- backward of a pointwise (1x1, stride 1) nn.Conv2d in ONE launch (CVC-YOLOv3/models.py:59-65, the 34 1x1 layers of yolo_baseline):
 * dx = dy . W (+ addsrc) [M x Cin] AND the fp32 slab partials of dW = dy^T . x (ws[slabs][Cout][Cin]; one slab per run of pixels, summed in
 * fixed order by mdcv_wgrad_reduce: bit-reproducible), a workgroup holding each dy / x pixel tile in LDS once for both products.  fy != NULL:
 * also the BatchNorm-backward partial sums of the layer that produced this conv's input, [slabs][2][Cin] (as mdcv_conv2d_dgrad_bnsums).
 * bf16 only; Cout (channels of dy, padded) in {64, 128, 256, 512}, Cin a multiple of 64.  mdcv_pw_bwd_slabs returns 0 for layers that do not
 * take this form (then: mdcv_conv2d mode 1 + mdcv_conv2d_wgrad). */
int mdcv_pw_bwd_slabs(int dtype, long long M, int Cin, int Cout, int ldy, int ldx, int lddx, int ldadd, int ldfy);
int mdcv_pw_bwd(int dtype, const void* dy, int ldy, const void* x, int ldx, const void* wd_packed, void* dx, int lddx, const void* addsrc,
                int ldadd, float* ws, int slabs, const void* fy, int ldfy, const float* fscale, const float* fshift, const float* fmean, int fact,
                float fslope, float* fpartial, long long M, int Cin, int Cout, void* stream);
/* Weight gradient of a conv -> BatchNorm -> activation layer whose INPUT needs no gradient (a network's first conv, CVC-YOLOv3/models.py:57-71 at
 * index 0), straight from dz (gradient of the activation output) and y (raw conv output): dy = cA*g + cB*y + cC, g = dz * act'(scale*y + shift) is formed
 * in the kernel's operand load, rounded to bf16 as mdcv_bn_act_bwd_apply rounds it -- bit-identical to apply + mdcv_conv2d_wgrad, without the apply
 * pass over the layer's output tensor.  _ok() = 1 where the geometry takes this form (bf16, Cout_pad <= 32); splits: any count
 * mdcv_conv2d_wgrad_splits_geom returns for the geometry -- the plans ask it with MDCV_TUNED(dtype, 20768), a target of three blocks per CU. */
int mdcv_conv2d_wgrad_bnapply_ok(int dtype, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int KH, int KW, int stride,
                                 int pad, int dil, int dz_ldc, int y_ldc, int x_ldc);
int mdcv_conv2d_wgrad_bnapply(int dtype, const void* dz, int dz_ldc, const void* y, int y_ldc, const float* scale, const float* shift,
                              const float* cA, const float* cB, const float* cC, int act, float slope, const void* x, int x_ldc, float* ws,
                              int splits, float* dw_oihw, int accumulate, int B, int Hin, int Win, int Cin, int Cin_real, int Hout, int Wout,
                              int Cout, int Cout_real, int KH, int KW, int stride, int pad, int dil, void* stream);
/* sum of fp32 weight-gradient slabs ws[splits][Cout_pad][KK * Cin_pad] into the OIHW gradient [Cout][Cin][KK] (fixed split order) */
int mdcv_wgrad_reduce(const float* ws, int splits, float* dw_oihw, int accumulate, int Cout_pad, int Cout, int Cin_pad, int Cin, int KK,
                      void* stream);

/* ---- the one exchange step of the data-parallel path, for hosts that do not go through torch.distributed:
 *      nn.DataParallel's gradient reduction (CVC-YOLOv3/train.py:193-195 with `losses[0].sum().backward()`, train.py:70) as an RCCL
 *      all-reduce(SUM) of the flat fp32 gradient buffer over xGMI, one process per GPU.  librccl is bound at run time (dlopen by
 *      soname, sharing the instance a host such as torch already loaded); -2 = librccl not found.  Return values > 0 are ncclResult_t.
 *      id128: 128-byte ncclUniqueId made by rank 0 (mdcv_comm_unique_id) and handed to every rank by the host (file, socket, MPI...). */
int mdcv_comm_unique_id(void* id128);
int mdcv_comm_init(void** comm, int nranks, const void* id128, int rank);     /* on the calling thread's current HIP device */
int mdcv_comm_allreduce_sum(void* comm, float* buf, long long n, void* stream);   /* in place, enqueue-only */
int mdcv_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
