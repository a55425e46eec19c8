/* ssn_hip.h -- C ABI of libssn_hip.so: the MI355X (gfx950) kernels behind the SSN hot path.
 *
 * The reference (yjxiong/action-detection) has no FFI layer: its hot path is Python that calls
 * torch ops, which in turn call cuDNN / cuBLAS.  Each entry point below replaces one of those
 * implicit native calls; the reference call site it stands in for is cited per function
 * (paths relative to /root/reference).  The Python mirror of the reference's nn.Module API
 * (action-detection_amd/ssn_models.py, ops/ssn_ops.py) binds these with ctypes; INTEGRATION.md
 * shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every buffer is device memory owned by the caller (PyTorch); the library allocates nothing
 *    persistent and never synchronises the stream;
 *  - tensors are fp32, NCHW, W-contiguous; "img_stride" arguments are the float distance between
 *    consecutive images, so a channel slice of a wider (concat) tensor is addressed by offsetting
 *    the base pointer to the slice's first channel and passing the wide tensor's C*H*W;
 *  - index / label tensors are int64 as in the reference;
 *  - return value: 0 on success, negative SSN_ERR_* otherwise (message: ssn_last_error());
 *    nothing throws across the ABI;
 *  - all launches go to `stream` (the caller's current HIP stream); re-entrant per stream.
 */
#ifndef SSN_HIP_H
#define SSN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

#define SSN_ERR_ARG (-1)
#define SSN_ERR_LAUNCH (-2)
#define SSN_ERR_WORKSPACE (-3)

const char* ssn_last_error(void);
int ssn_abi_version(void);   /* 9 */

/* ------------------------------------------------------------------ backbone: conv + BN + ReLU
 * Weight re-layout for the implicit-GEMM kernels (what cuDNN does internally with its filter
 * descriptors): w is the torch-layout [Cout][Cin][k][k] weight; `packed` receives
 * ssn_conv_packed_floats() floats.  transposed=0 -> operand of the forward conv, 1 -> of dgrad,
 * 2 -> of the parity-ordered stride-2 dgrad (see ssn_conv_dgrad_layout). */
long ssn_conv_packed_floats(int Cout, int Cin, int ksize, int transposed);
int ssn_conv_pack_weights(const float* w, float* packed, int Cout, int Cin, int ksize, int transposed,
                          hipStream_t stream);

/* The same for `count` layers in ceil(count/40) launches; every argument is a HOST array with one entry per
 * layer.  w1/split describe the optional second source of a fused pair (output channels >= split come from w1);
 * pass w1 = NULL and split = cout otherwise. */
int ssn_conv_pack_weights_multi(int count, const float* const* w0, const float* const* w1, float* const* out,
                                const int* cout, const int* cin, const int* ksize, const int* mode, const int* split,
                                hipStream_t stream);

/* Replaces cuDNN conv fwd + cudnnBatchNorm(eval) + ReLU of every "conv / bn / relu" triple of
 * model_zoo.BNInception, reached from ssn_models.py:266 (train) and :298 (test).
 * y[n][co][ho][wo] = relu?( scale[co] * sum_{ci,r,s} w[co][ci][r][s] * x[n][ci][ho*S-pad+r][wo*S-pad+s]
 *                           + shift[co] )        scale/shift may be NULL (plain convolution).
 * ksize in {1,3,7}, stride in {1,2}.  tile_cfg < 0 selects the tile heuristically.
 * w_packed = ssn_conv_pack_weights(w, transposed=0). */
int ssn_conv_bn_relu_fwd(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                         int N, int Cin, int H, int W, long x_img_stride, int Cout, int Ho, int Wo,
                         long y_img_stride, int ksize, int stride, int pad, int relu, int tile_cfg,
                         float* y_amax, hipStream_t stream);   /* y_amax: see "amax slots" below; NULL = not tracked */

/* Frozen-BN folding (ssn_models.py:156-174 puts every BatchNorm2d in eval mode):
 * scale = gamma / sqrt(var + eps), shift = (conv_bias - mean) * scale + beta. */
int ssn_bn_fold(const float* conv_bias, const float* gamma, const float* beta, const float* mean,
                const float* var, float eps, float* scale, float* shift, int C, hipStream_t stream);

/* ssn_bn_fold for `count` layers in ceil(count/48) launches; every argument is a HOST array with one entry
 * per layer (device pointers, eps, channel count). */
int ssn_bn_fold_multi(int count, const float* const* conv_bias, const float* const* gamma, const float* const* beta,
                      const float* const* mean, const float* const* var, const float* eps, float* const* scale,
                      float* const* shift, const int* channels, hipStream_t stream);

/* Backward of ReLU + frozen BN, in place on dy:  dy <- dy * (y > 0) * scale[c]
 * (autograd of the same triples, entered from ssn_train.py:236 loss.backward()). */
int ssn_relu_bn_bwd(float* dy, const float* y, const float* scale, int N, int C, int HW, long dy_img_stride,
                    long y_img_stride, float* dy_amax, hipStream_t stream);

/* cuDNN dgrad replacement.  wt_packed = ssn_conv_pack_weights(w, transposed=1).
 * dx[n][ci][hi][wi] (+)= sum_{co,r,s} w[co][ci][r][s] * dy[n][co][(hi+pad-r)/S][(wi+pad-s)/S].
 * mask_y / mask_scale (optional, both or neither): when this call is the last writer of dx, the
 * backward of the ReLU + frozen BN that produced the tensor dx belongs to is fused into the store:
 * dx <- dx * (mask_y > 0) * mask_scale[ci]   (mask_scale[ci] NaN: not a ReLU output, dx unchanged;
 * finite scales of either sign are applied as they are).
 * wt_layout = ssn_conv_dgrad_layout(...): 2 selects the parity-ordered stride-2 path (3x3/s2/p1, even input),
 * whose weights must have been packed with transposed = 2; otherwise 1. */
int ssn_conv_dgrad_layout(int ksize, int stride, int pad, int H, int W);
int ssn_conv_dgrad(const float* dy, const float* wt_packed, float* dx, int N, int Cout, int Ho, int Wo,
                   long dy_img_stride, int Cin, int H, int W, long dx_img_stride, int ksize, int stride,
                   int pad, int accumulate, const float* mask_y, long mask_img_stride, const float* mask_scale,
                   int wt_layout, int tile_cfg, float* dx_amax, hipStream_t stream);

/* ---- amax slots (operand scaling of the split kernels below).  A slot is ONE float in device memory holding an
 * upper bound of max |t| of a tensor t; the caller zeroes it before the first kernel writes t, every kernel of this
 * library that stores into t and is handed the slot (the *_amax output arguments; NULL = t is not tracked) raises
 * it to the largest magnitude it stores (unsigned atomic max on the bit pattern), and the split kernels that READ t
 * as a matrix operand derive its power-of-two f16 scale from the slot (the const *_amax arguments, required).
 * ssn_tensor_amax does the same for a tensor no library kernel produced (the frames handed in by the caller). */
int ssn_tensor_amax(const float* x, long n, float* slot, hipStream_t stream);

/* ---- "x6" variants: the same convolutions computed on the f16 matrix cores with fp32-class accuracy ("x6" is the
 * family's historical name: its first version multiplied six bf16 partial products).
 * Every fp32 operand, scaled by a per-tensor power of two, is split into two f16 terms (11+11 significand bits,
 * round to nearest) and the product is accumulated in fp32 from the three partial products a_lo*b_hi + a_hi*b_hi +
 * a_hi*b_lo (csrc/conv_x6.hip); the dropped term is <= 2^-22 |ab|, and the measured error against float64 stays
 * within 2x of an fp32 FMA chain (the 1e-4 budget of the path is untouched) at 3/16 of the f32-MFMA matrix time.
 * ksize in {1,3}; forward stride in {1,2}; dgrad stride 1.
 * Weights: ssn_conv_x6_pack_weights_multi (mode 0 forward / 1 dgrad operand; same w1/split convention as
 * ssn_conv_pack_weights_multi; measures max |w|, scales, splits and packs), ssn_conv_x6_packed_floats() floats per layer.
 * x_amax / dy_amax (required): amax slot of the tensor the gathered operand lives in; y_amax / dx_amax: slot of the
 * output tensor or NULL.
 * x_guard_bytes / dy_guard_bytes: how many bytes directly in FRONT of the gathered tensor the caller guarantees to
 * be readable device memory (e.g. the channels below a channel slice, or an allocation pad).  With >= 256 the
 * stride-1 kernels fetch activations 16 bytes per lane (4 consecutive pixels; border positions are zeroed later),
 * which reaches a few bytes before the first element; with 0 they use 4-byte loads and never leave the tensor.  Same call sites as
 * ssn_conv_bn_relu_fwd / ssn_conv_dgrad (ssn_models.py:266,298; loss.backward at ssn_train.py:223). */
long ssn_conv_x6_packed_floats(int Cout, int Cin, int ksize, int transposed);
void ssn_conv_x6_debug_trace(unsigned long long* per_block_8_words); /* tooling only; NULL = off */
void ssn_conv_x6_debug_flags(int flags);   /* tooling only (tools/ablate_x6.py); 0 = normal operation */
/* The fused launch on an Inception block input (1x1 branch + reduce pair + pool projection = ONE convolution):
 *  - w0..w3, split..split3: up to four weight sources per entry -- output channels [0, split) from w0, [split, split2)
 *    from w1, [split2, split3) from w2, the rest from w3; all splits = cout for a single source;
 *  - raw_from (fwd, > 0): output channels >= raw_from take neither the affine nor the ReLU (the bias-free projection);
 *  - row_split / row_gap (fwd; multiples of 32, 0 = none): output channels >= row_split are stored row_gap channels further
 *    up y's tensor (the 1x1 branch goes to the head of the block output, the rest behind the block's own channels);
 *  - k_split / k_gap (dgrad; multiples of 16): the same displacement on the channel axis of dy;
 *  - g_row_split / g_row_gap (wgrad): ... and on the rows of g;
 *  - y_amax2 / dy_amax2 / g_amax2 (NULL = none): such a tensor has ONE AMAX SLOT PER REGION (the block's own channels; the rows
 *    behind them), so that every slot is complete before anything reads it although the regions are written and read by
 *    concurrent launches; the launch that spans both raises both (fwd) resp. uses the larger of both (dgrad, wgrad). */
int ssn_conv_x6_pack_weights_multi(int count, const float* const* w0, const float* const* w1, const float* const* w2,
                                   const float* const* w3, float* const* out, const int* cout, const int* cin,
                                   const int* ksize, const int* mode, const int* split, const int* split2,
                                   const int* split3, hipStream_t stream);
/* Batched packing: every ssn_conv_x6_pack_* call between _begin and _end only RECORDS its entries; _end issues all of them in three
 * launches (clear the amax tails, max |w| per entry, pack) through a device-resident plan.  plan: >= ssn_conv_x6_pack_batch_entries()
 * * ssn_conv_x6_pack_entry_bytes() bytes of device memory the caller keeps from step to step; it is rewritten (ceil(entries / 40)
 * small launches) only when the recorded entries differ from the ones last written there, or with force_write (a buffer the caller
 * has just allocated).  _abort drops an open batch.  One batch at a time per process (one process per GPU). */
int ssn_conv_x6_pack_batch_begin(void);
int ssn_conv_x6_pack_batch_entries(void);
long ssn_conv_x6_pack_entry_bytes(void);
void ssn_conv_x6_pack_batch_abort(void);
int ssn_conv_x6_pack_batch_end(void* plan, long plan_bytes, int force_write, hipStream_t stream);
int ssn_conv_x6_fwd(const float* x, const float* w_packed, const float* scale, const float* shift, float* y, int N,
                    int Cin, int H, int W, long x_img_stride, int Cout, int Ho, int Wo, long y_img_stride,
                    int ksize, int stride, int pad, int relu, int x_guard_bytes, int tile_cfg, const float* x_amax,
                    float* y_amax, int raw_from, int row_split, int row_gap, float* y_amax2, hipStream_t stream);
int ssn_conv_x6_dgrad(const float* dy, const float* wt_packed, float* dx, int N, int Cout, int Ho, int Wo,
                      long dy_img_stride, int Cin, int H, int W, long dx_img_stride, int ksize, int pad,
                      int accumulate, const float* mask_y, long mask_img_stride, const float* mask_scale,
                      int dy_guard_bytes, int tile_cfg, const float* dy_amax, float* dx_amax, int k_split, int k_gap,
                      const float* dy_amax2, hipStream_t stream);

/* Rectangular taps (csrc/conv_x6_rect.hip): the forward convolutions of the Inception-v3 backbone the reference's
 * tester runs on ActivityNet (ssn_models.py:133-139): kh x kw in {5x5, 1x7, 7x1, 1x3, 3x1}, stride 1, per-axis
 * padding; same kernel, arguments and accuracy class as ssn_conv_x6_fwd.  Their data gradient (training SSN on
 * Inception-v3, ssn_models.py:133-139 + ssn_train.py:236) is the forward correlation of dy with the transposed,
 * tap-reversed weight (ssn_conv_x6_pack_dgrad_rect): ssn_conv_x6_dgrad_rect, same-size layers (2 pad = taps - 1),
 * accumulate / mask_y / mask_scale as ssn_conv_x6_dgrad. */
long ssn_conv_x6_packed_floats_rect(int Cout, int Cin, int kh, int kw);
int ssn_conv_x6_pack_weights_rect(const float* w, float* out, int cout, int cin, int kh, int kw, hipStream_t stream);
int ssn_conv_x6_fwd_rect(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                         int N, int Cin, int H, int W, long x_img_stride, int Cout, int Ho, int Wo,
                         long y_img_stride, int kh, int kw, int pad_h, int pad_w, int relu, int x_guard_bytes,
                         int tile_cfg, const float* x_amax, float* y_amax, hipStream_t stream);
/* `count` rectangular-tap weights in one call (host arrays, one entry per layer); mode 0: forward operand, 2: dgrad operand */
int ssn_conv_x6_pack_rect_multi(int count, const float* const* w, float* const* out, const int* cout, const int* cin,
                                const int* kh, const int* kw, const int* mode, hipStream_t stream);
long ssn_conv_x6_packed_floats_dgrad_rect(int Cout, int Cin, int kh, int kw);
int ssn_conv_x6_pack_dgrad_rect(const float* w, float* out, int cout, int cin, int kh, int kw, hipStream_t stream);
int ssn_conv_x6_dgrad_rect(const float* dy, const float* wt_packed, float* dx, int N, int Cout, int H, int W,
                           long dy_img_stride, int Cin, long dx_img_stride, int kh, int kw, int pad_h, int pad_w,
                           int accumulate, const float* mask_y, long mask_img_stride, const float* mask_scale,
                           int dy_guard_bytes, int tile_cfg, const float* dy_amax, float* dx_amax, hipStream_t stream);

/* Data gradient of the 3x3 / stride-2 / pad-1 layers (even input size) on the x6 kernel: four stride-1 launches, one
 * per parity class of the input pixel, each multiplying only the taps that reach that class (cuDNN dgrad behind
 * loss.backward(), ssn_train.py:236).  wt_packed: ssn_conv_x6_pack_dgrad_s2 (ssn_conv_x6_dgrad_s2_packed_floats
 * floats) of the torch-layout weight [Cout][Cin][3][3].  pad = 1 (BN-Inception; even input size) or 0 (the "valid"
 * stride-2 layers of Inception-v3, any input size >= 3).  Other arguments as ssn_conv_x6_dgrad. */
long ssn_conv_x6_dgrad_s2_packed_floats(int Cout, int Cin);
int ssn_conv_x6_pack_dgrad_s2(const float* w, float* out, int cout, int cin, hipStream_t stream);
int ssn_conv_x6_dgrad_s2(const float* dy, const float* wt_packed, float* dx, int N, int Cout, int Ho, int Wo,
                         long dy_img_stride, int Cin, int H, int W, long dx_img_stride, int accumulate,
                         const float* mask_y, long mask_img_stride, const float* mask_scale, int dy_guard_bytes,
                         int tile_cfg, const float* dy_amax, float* dx_amax, int pad, hipStream_t stream);

/* x6 weight gradient (csrc/conv_wgrad_x6.hip): stride-1 same-size 1x1 / 3x3 convolutions with H*W % 4 == 0, both
 * operands scaled and split to f16 on the fly (g_amax / x_amax: their tensors' amax slots, required), 16-byte loads
 * along the pixel axis.  Same result contract as ssn_conv_wgrad
 * (deterministic split-K).  x_guard_bytes >= 256 is REQUIRED (the shifted taps read up to W+1 floats before x). */
long ssn_conv_wgrad_x6_workspace_bytes(int N, int Cin, int Cout, int H, int W, int ksize, int tile_cfg);
int ssn_conv_wgrad_x6(const float* g, const float* x, float* dw, float* db, int N, int Cin, int H, int W,
                      long x_img_stride, int Cout, long g_img_stride, int ksize, int pad, int x_guard_bytes,
                      void* workspace, long ws_bytes, int tile_cfg, const float* g_amax, const float* x_amax,
                      int g_row_split, int g_row_gap, const float* g_amax2, hipStream_t stream);
/* ... of the stride-1 same-size layers with rectangular taps (kh x kw, 2 pad = taps - 1; dw [Cout][Cin][kh][kw]); x needs
 * (pad_h * W + pad_w) * 4 readable bytes in front of it, rounded up to a multiple of 256.  pad_h = pad_w = 0: the taps reach
 * 0 .. k-1 pixels BEHIND the pixel -- the weight gradient of an unpadded convolution whose output gradient was laid into
 * planes of the input's size (ssn_embed_planes). */
long ssn_conv_wgrad_x6_rect_workspace_bytes(int N, int Cin, int Cout, int H, int W, int kh, int kw, int tile_cfg);
int ssn_conv_wgrad_x6_rect(const float* g, const float* x, float* dw, float* db, int N, int Cin, int H, int W,
                           long x_img_stride, int Cout, long g_img_stride, int kh, int kw, int pad_h, int pad_w,
                           int x_guard_bytes, void* workspace, long ws_bytes, int tile_cfg, const float* g_amax,
                           const float* x_amax, hipStream_t stream);
/* second pass of both wgrad kernels: dw[m][kk] = sum_z part[z][m][kk], db[m] = sum_z part[z][m][K] */
int ssn_wgrad_reduce(const float* part, float* dw, float* db, int M, int K, int splits, hipStream_t stream);
/* the same with TAP-MAJOR slab columns (column t * (K / taps) + ci holds dW[m][ci][t]): the nine-tap planes kernel's slabs */
int ssn_wgrad_reduce_taps(const float* part, float* dw, float* db, int M, int K, int splits, int taps, hipStream_t stream);

/* cuDNN wgrad (+ bias grad) replacement.  dw[co][ci][r][s] = sum_p g * x,  db[co] = sum_p g  (db may be NULL).
 * workspace: ssn_conv_wgrad_workspace_bytes() bytes of scratch for the split-K partial slabs. */
long ssn_conv_wgrad_workspace_bytes(int N, int Cin, int Cout, int Ho, int Wo, int ksize, int tile_cfg);
int ssn_conv_wgrad(const float* g, const float* x, float* dw, float* db, int N, int Cin, int H, int W,
                   long x_img_stride, int Cout, int Ho, int Wo, long g_img_stride, int ksize, int stride, int pad,
                   void* workspace, long ws_bytes, int tile_cfg, hipStream_t stream);
int ssn_conv_pick_tile(int M, long P);
/* tooling only: ablation switches for the conv kernel (tools/ablate_conv.py); 0 = normal operation */
int ssn_conv_debug_flags(int flags);

/* ------------------------------------------------------------------ backbone: training-mode BatchNorm2d
 * bn_mode 'partial' (first BatchNorm2d) / 'full' (all) of ssn_models.py:95-105,156-174 (csrc/bn_train.hip): the
 * layers SSN.train() leaves in training mode normalise with batch statistics, update running_mean / running_var
 * (momentum, unbiased variance) and take the full batch-norm backward.  z is the convolution output WITHOUT the conv
 * bias (it cancels in z - mean(z); conv_bias only enters the running mean).  workspace:
 * ssn_bn_train_workspace_floats(N, C) floats.  Reductions are two-level, atomic-free, combined in double.
 *   stats : mean[c], invstd[c] = 1/sqrt(biased var + eps) over (N, HW); running stats updated in place (or NULL)
 *   apply : y = relu?(gamma * (z - mean) * invstd + beta)
 *   bwd   : g = dy * (y > 0 | 1), dbeta = sum g, dgamma = sum g * xhat, dz = gamma * invstd * (g - dbeta/n - xhat * dgamma/n) */
long ssn_bn_train_workspace_floats(int N, int C);
int ssn_bn_train_stats(const float* z, const float* conv_bias, float* mean, float* invstd, float* running_mean,
                       float* running_var, int N, int C, int HW, long z_img_stride, float eps, float momentum,
                       void* workspace, size_t ws_bytes, hipStream_t stream);
int ssn_bn_train_apply(const float* z, float* y, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, int relu, int N, int C, int HW, long z_img_stride, long y_img_stride,
                       float* y_amax, hipStream_t stream);
int ssn_bn_train_bwd(const float* dy, const float* y, const float* z, const float* mean, const float* invstd,
                     const float* gamma, float* dgamma, float* dbeta, float* dz, int relu, int N, int C, int HW,
                     long dy_img_stride, long y_img_stride, long z_img_stride, long dz_img_stride, void* workspace,
                     size_t ws_bytes, float* dz_amax, hipStream_t stream);

/* ------------------------------------------------------------------ backbone: pooling
 * Max / average pools of BN-Inception (ceil_mode output sizes computed by the caller, avg with
 * count_include_pad=True) and the global average pool before `fc` (ssn_models.py:266 -> backbone).
 * argmax: uint8 [N][C][Ho][Wo] window-local index, written by fwd(max) and consumed by bwd(max). */
int ssn_pool_fwd(int is_max, const float* x, float* y, unsigned char* argmax, int N, int C, int H, int W,
                 long x_img_stride, int Ho, int Wo, long y_img_stride, int ksize, int stride, int pad,
                 float* y_amax, hipStream_t stream);
int ssn_pool_bwd(int is_max, const float* dy, const unsigned char* argmax, float* dx, int N, int C, int H, int W,
                 long dx_img_stride, int Ho, int Wo, long dy_img_stride, int ksize, int stride, int pad,
                 int accumulate, const float* mask_y, long mask_img_stride, const float* mask_scale,
                 float* dx_amax, hipStream_t stream);
/* y = relu?(scale[c] * avgpool(x) + shift[c]) (average pools only): the pool-projection branch of an Inception block
 * (<block>_pool -> <block>_pool_proj + BN + ReLU) evaluated as avgpool(conv1x1(x)) -- identical to conv1x1(avgpool(x))
 * for zero padding with count_include_pad -- so that the pool touches the projection's output channels only. */
int ssn_avgpool_affine_fwd(const float* x, float* y, const float* scale, const float* shift, int relu, int N, int C,
                           int H, int W, long x_img_stride, int Ho, int Wo, long y_img_stride, int ksize, int stride,
                           int pad, float* y_amax, hipStream_t stream);
/* out[c] = sum_{n,hw} g[n][c][hw] (fixed order): bias gradient of such a projection (the gradient BEFORE the pool's
 * backward; the wgrad kernel's bias column sees the pooled gradient). */
int ssn_channel_sum(const float* g, float* out, int N, int C, int HW, long img_stride, void* workspace,
                    size_t ws_bytes, hipStream_t stream);   /* workspace: C * ssn_channel_sum_shares(N) floats */
int ssn_channel_sum_shares(int N);
int ssn_global_avgpool_fwd(const float* x, float* y, int N, int C, int HW, long x_img_stride, hipStream_t stream);
int ssn_global_avgpool_bwd(const float* dy, float* dx, int N, int C, int HW, long dx_img_stride, int accumulate,
                           float* dx_amax, hipStream_t stream);

/* nn.Dropout standing in for the backbone's `fc` (ssn_models.py:71-74). mask: uint8 per element.
 * counter (optional, int64[1] in device memory) is mixed into the Philox key and incremented by the
 * call, so a hipGraph replay draws a fresh mask each time. */
int ssn_dropout_fwd(const float* x, float* y, unsigned char* mask, long total, float p, unsigned long long seed,
                    long* counter, hipStream_t stream);
int ssn_dropout_bwd(const float* dy, const unsigned char* mask, float* dx, long total, float p,
                    hipStream_t stream);

/* ------------------------------------------------------------------ STPP
 * StructuredTemporalPyramidPooling.forward (ops/ssn_ops.py:39-70).  The part table carries the
 * reference's integer tick truncation (ops/ssn_ops.py:53-55), computed once on the host. */
#define SSN_STPP_MAX_PARTS 24
typedef struct SsnStppTable {
    int n_parts;
    int n_seg;
    int act_lo, act_hi;
    int lo[SSN_STPP_MAX_PARTS];
    int hi[SSN_STPP_MAX_PARTS];
    int norm[SSN_STPP_MAX_PARTS];
    int col[SSN_STPP_MAX_PARTS];
} SsnStppTable;
int ssn_stpp_fwd(const float* ft, const float* scaling, float* act_ft, float* stpp_ft, int P, int D,
                 const SsnStppTable* table, hipStream_t stream);
int ssn_stpp_bwd(const float* d_act, const float* d_stpp, const float* scaling, float* d_ft, int P, int D,
                 const SsnStppTable* table, hipStream_t stream);
/* The fused head of SSN.train_forward (ssn_models.py:268-289): STPP + activity / completeness / regression Linear + the prop_type
 * row selection as ONE launch each way (instead of 7 forward, 16 backward), same arithmetic and summation order as ssn_stpp_* /
 * ssn_linear_* / ssn_row_gather|scatter.  w / b / pos / idx / out / O / n: HOST arrays of 3 (activity, completeness, regression;
 * w[2] == NULL: no regression head); pos[h]: DEVICE int [P] = row of proposal p in head h's gathered output or -1; idx[h]: DEVICE
 * long [n_h] = proposal of gathered row r.  Forward: out[h] [n_h][O_h], act_ft [P][D], stpp_ft [P][m D] (kept for the backward).
 * Backward: dout[h] gradients of the gathered outputs -> d_ft [P n_seg][D], dw[h], db[h]. */
int ssn_heads_fwd(const float* ft, const float* scaling, const float* const* w, const float* const* b, const int* const* pos,
                  const long* const* idx, float* const* out, const int* O, const int* n, float* act_ft, float* stpp_ft, int P, int D,
                  const SsnStppTable* table, hipStream_t stream);
int ssn_heads_bwd(const float* ft_unused, const float* scaling, const float* const* w, const float* const* b, const int* const* pos,
                  const long* const* idx, float* const* dout, const int* O, const int* n, float* act_ft, float* stpp_ft, int P, int D,
                  const SsnStppTable* table, float* d_ft, float* const* dw, float* const* db, hipStream_t stream);
/* STPPReorgainzed.forward (ops/ssn_ops.py:109-170), stand-alone activity classifier form.
 * ranges: int32 [P][n_parts][2] row ranges (pr<=pl: skipped); act_range: int32 [P][2]. */
int ssn_stpp_reorg(const float* scores, int T, int D, const int* ranges, const int* act_range,
                   const float* scaling, const int* part_scale_col, int P, int n_parts, int act_len, int comp_len,
                   int reg_len, float* out_act, float* out_comp, float* out_reg, hipStream_t stream);

/* Dense testing (ssn_test.py:84-90): mean over the crops of the per-frame rows, `rst.view(num_crop, -1, D).mean(0)`
 * (x [num_crop][T][D] -> y [T][D]); and the regression de-normalisation reg[..., k] = reg[..., k] * std[k] + mean[k]
 * on reg [n_pairs][2], in place. */
int ssn_crop_mean(const float* x, float* y, int num_crop, int T, int D, hipStream_t stream);
int ssn_reg_denorm(float* reg, long n_pairs, float mean0, float std0, float mean1, float std1, hipStream_t stream);

/* RGBDiff input (SSN._get_diff, ssn_models.py:302-316, keep_rgb = False): in [n_segments][new_length + 1][C][HW] ->
 * out [n_segments][new_length][C][HW], out[g][x] = in[g][x + 1] - in[g][x]. */
int ssn_frame_diff(const float* in, float* out, long n_segments, int new_length, int C, int HW, hipStream_t stream);

/* Input side (csrc/frames.hip): the arithmetic of the reference's transform chain after decoding / scaling --
 * GroupOverSample or crop + horizontal flip, Stack(roll), ToTorchFormatTensor(div=False), GroupNormalize
 * (transforms.py:103-132, 49-64, 256-288, 67-80) -- on decoded uint8 frames.
 * src [n_img][Hs][Ws][C] uint8 HWC -> dst [n_crops][n_img][C][crop_h][crop_w] fp32; crop k is (off_x, off_y, flip);
 * roll: reverse the channel order (RGB -> BGR, Stack(roll=True)); invert_even: 255 - px on the even images of
 * flipped crops (flow x component); mean / stdv (DEVICE arrays of n_mean / n_std floats) repeat over the stacked
 * channels like GroupNormalize. */
int ssn_frames_crop_normalize(const unsigned char* src, float* dst, int n_img, int Hs, int Ws, int C, int crop_h,
                              int crop_w, int n_crops, const int* off_x, const int* off_y, const int* flip, int roll,
                              int invert_even, const float* mean, int n_mean, const float* stdv, int n_std,
                              hipStream_t stream);
/* The TRAINING chain on the GPU: GroupMultiScaleCrop (crop + PIL's bilinear resize to the network input: transforms.py:135-206,
 * restated bit-exactly from Pillow's 8-bit ImagingResample) -> GroupRandomHorizontalFlip (:49-64) -> Stack(roll) ->
 * ToTorchFormatTensor(div=False) -> GroupNormalize, i.e. SSN.get_augmentation() + the tail of ssn_train.py:106-111, on decoded
 * uint8 frames: the loader workers only decode.  src [n_img][Hs][Ws][C] uint8 -> dst [n_img][C][out_h][out_w] fp32; box: DEVICE int
 * [n_img][4] = (x0, y0, crop_w, crop_h) per image (one box per group in the reference), flip: DEVICE int [n_img]; the caller checks
 * that boxes lie inside the frame and crop / output <= 3 per axis.  workspace: ssn_frames_resize_workspace_bytes() device bytes. */
size_t ssn_frames_resize_workspace_bytes(int n_img, int out_h, int out_w);
int ssn_frames_crop_resize_normalize(const unsigned char* src, float* dst, int n_img, int Hs, int Ws, int C, int out_h, int out_w,
                                     const int* box, const int* flip, int roll, int invert_even, const float* mean, int n_mean,
                                     const float* stdv, int n_std, void* workspace, size_t workspace_bytes, hipStream_t stream);

/* Detection post-processing of one video (csrc/detect.hip): score fusion softmax(activity)[1:] * exp(completeness),
 * top-k over all (proposal, class) pairs, temporal NMS per class and location regression
 * (eval_detection_results.py:91-128, 167-178; ops/utils.py:56-82).  act [P][C+1], comp [P][C], reg [P][C][2] or NULL
 * (fp32); rel_prop [P][2] fp64 normalised spans; combined [P][C] fp32 out; dets [C][max_det][5] fp64 = (start, end,
 * score, loc, dur) in descending score order, counts [C] int32; workspace: ssn_detections_workspace_bytes(P, C) bytes
 * of device scratch.  include_bg 1: softmax over all C+1 activity scores (the reference's top_k <= 0 branch) instead of
 * the C class scores (0); 2: no softmax, raw class scores (the --cls_scores branch without --softmax_before_filter, :135); top_k <= 0 keeps every pair, otherwise EXACTLY top_k pairs are kept as np.argsort(...)[-top_k:]
 * does (ties at the k-th score: the higher flat indices, i.e. a stable sort's choice).  Score ties inside a class:
 * higher proposal index first (scores.argsort()[::-1] of a stable sort).  Non-finite scores are ordered as numpy
 * orders them (NaN above +inf) and can never cause an out-of-range access.  Any P is accepted (P > 2048 sorts in the
 * workspace instead of LDS). */
size_t ssn_detections_workspace_bytes(int P, int C);
int ssn_detections(const float* act, const float* comp, const float* reg, const double* rel_prop, float* combined,
                   double* dets, int* counts, void* workspace, size_t ws_bytes, int P, int C, int max_det, int top_k,
                   int include_bg, double nms_thresh, int regress, hipStream_t stream);

/* ------------------------------------------------------------------ heads
 * nn.Linear fwd/bwd for activity_fc / completeness_fc / regressor_fc / test_fc
 * (ssn_models.py:77-78,87,272-273,283,300; cuBLAS GEMMs in the reference). */
int ssn_linear_fwd(const float* x, const float* w, const float* b, float* out, int R, int O, int D,
                   hipStream_t stream);
int ssn_linear_bwd(const float* dout, const float* x, const float* w, float* dx, float* dw, float* db, int R,
                   int O, int D, int accumulate_dx, hipStream_t stream);
/* prop_type row selection (ssn_models.py:275-289). */
int ssn_row_gather(const float* src, const long* index, float* dst, int n_idx, int width, hipStream_t stream);
int ssn_row_scatter(const float* src, const long* index, float* dst, int n_idx, int n_rows, int width,
                    hipStream_t stream);

/* ------------------------------------------------------------------ losses
 * CrossEntropyLoss (ssn_train.py:133,210).  workspace: 2*R floats; the first R (lse) feed the bwd. */
int ssn_ce_loss_fwd(const float* logits, const long* target, float* loss, float* workspace, int R, int C,
                    hipStream_t stream);
int ssn_ce_loss_bwd(const float* logits, const long* target, const float* lse, const float* gout, float* dlogits,
                    int R, int C, hipStream_t stream);
/* CompletenessLoss + OHEMHingeLoss (ops/ssn_ops.py:173-239).  den = pos_cnt + neg_cnt as the
 * reference truncates it (ops/ssn_ops.py:236-239).  coef: R floats kept for bwd; workspace 2*R floats. */
int ssn_completeness_loss_fwd(const float* pred, const long* labels, float* loss, float* coef, float* workspace,
                              int R, int C, int group, int split, int keep_pos, int keep_neg, float den,
                              hipStream_t stream);
int ssn_completeness_loss_bwd(const long* labels, const float* coef, const float* gout, float* dpred, int R, int C,
                              float den, hipStream_t stream);
/* ClassWiseRegressionLoss (ops/ssn_ops.py:242-258).  diff: 2*n floats kept for bwd. */
int ssn_cw_smoothl1_fwd(const float* pred, const long* labels, const float* targets, float* loss, float* diff,
                        int n, int C, hipStream_t stream);
int ssn_cw_smoothl1_bwd(const long* labels, const float* diff, const float* gout, float* dpred, int n, int C,
                        hipStream_t stream);

/* [r6] The training objective of ssn_train.py:210-214 -- activity CE + w_comp * completeness + w_reg * regression -- in ONE launch
 * (the three losses above, same bodies and summation orders, then the mix) and ONE backward launch.  losses[4] = activity,
 * completeness, regression (0 without that head), total.  lse [Ra], coef [Rc], diff [2 n_reg] are kept for the backward; scratch:
 * max(2 Ra, 2 Rc) floats.  reg_pred == NULL (d_reg == NULL in the backward): no regression head (ssn_models.py:288-289). */
int ssn_total_loss_fwd(const float* act_logits, const long* act_target, int Ra, int Ca, const float* comp_pred,
                       const long* comp_labels, int Rc, int Cc, int group, int split, int keep_pos, int keep_neg, float den,
                       const float* reg_pred, const long* reg_labels, const float* reg_targets, int n_reg, int Cr, float w_comp,
                       float w_reg, float* losses, float* lse, float* coef, float* diff, float* scratch, hipStream_t stream);
int ssn_total_loss_bwd(const float* act_logits, const long* act_target, int Ra, int Ca, const long* comp_labels, int Rc, int Cc,
                       float den, const long* reg_labels, int n_reg, int Cr, float w_comp, float w_reg, const float* lse,
                       const float* coef, const float* diff, const float* gout, float* d_act, float* d_comp, float* d_reg,
                       hipStream_t stream);
/* The label / target bookkeeping of SSN.train_forward (ssn_models.py:275-289): target[idx0], target[idx1], target[idx2],
 * reg_target[idx2] in one launch (the reference: four index_select behind three nonzero()).  n2 == 0: no regression. */
int ssn_label_select(const long* target, const float* reg_target, const long* idx0, int n0, const long* idx1, int n1,
                     const long* idx2, int n2, long* out0, long* out1, long* out2, float* out_reg, hipStream_t stream);

/* [r6] Checksum of the parameters a cached derivative (packed weights, folded BatchNorm vectors of the inference cache,
 * planes_exec.py) was built from -- the reference re-reads its parameters in every forward (ssn_models.py:298), so a write through
 * `p.data` is seen at once; torch's version counters do not see it, the bits do.  table: device array of n_entries {const void* ptr;
 * long n_words}; slot: device uint64, 0 on entry.  expected == NULL: slot += checksum (recording).  Otherwise additionally
 * flag |= bit when slot != *expected, and slot <- 0. */
int ssn_param_checksum(const void* table, int n_entries, unsigned long long* slot, const unsigned long long* expected, int* flag,
                       int bit, hipStream_t stream);

/* ------------------------------------------------------------------ optimiser step
 * torch.optim.SGD(momentum, weight_decay) over one flat segment (ssn_train.py:141-144,252);
 * per-group lr_mult / decay_mult of ssn_models.py:240-251 are folded into lr / weight_decay. */
/* skip_flag (device int, may be NULL): read at launch time; non-zero = leave w and the momentum buffer untouched.  It is the word
 * ssn_pl_range_check raises when a tensor of the step left the range of its delayed scale: optimizer.step() (ssn_train.py:252) of a
 * flagged step becomes a no-op, so the host can repeat the step with fresh scales -- also from inside a hipGraph replay. */
int ssn_sgd_step(float* w, const float* grad, float* momentum_buf, long n, float lr, float momentum,
                 float weight_decay, float grad_scale, int first_step, const int* skip_flag, hipStream_t stream);
/* The same update for `count` tensors in ceil(count/48) launches; w / grad / momentum_buf / n / lr / weight_decay
 * are HOST arrays (of device pointers resp. scalars), one entry per tensor. */
int ssn_sgd_step_multi(int count, float* const* w, const float* const* grad, float* const* momentum_buf,
                       const long* n, const float* lr, const float* weight_decay, float momentum, float grad_scale,
                       int first_step, const int* skip_flag, hipStream_t stream);
/* clip_grad_norm support (ssn_train.py:245-249): out[0] (+)= sum(x^2); workspace >= 1024 floats. */
int ssn_sumsq(const float* x, long n, float* out, int accumulate, float* workspace, hipStream_t stream);
int ssn_scale(float* x, long n, const float* coef_dev, float coef, hipStream_t stream);
/* The stem (conv1_7x7_s2 of model_zoo.BNInception, behind ssn_models.py:266) on the split kernels through its space-to-depth
 * form: a k x k / stride-2 / pad (k-1)/2 convolution on C channels is a (k+1)/2-tap stride-1 convolution on the 4C channels
 * xs[(c*2+a)*2+b][h'][w'] = x[c][2h'+a][2w'+b], with 2 padding pixels in front and 1 behind (k = 7): forward =
 * ssn_conv_x6_fwd_rect(xs, pack_rect(ssn_s2d_weights(w)), 4, 4, pad 2, Ho = H/2), weight gradient = ssn_conv_wgrad_x6(ksize 4,
 * pad 2) followed by ssn_s2d_weights_bwd.  (3 input channels fill 3/16 of a split slab directly; 12 fill 12/16.) */
int ssn_space_to_depth2(const float* x, float* xs, int N, int C, int H, int W, float* xs_amax, hipStream_t stream);
int ssn_s2d_weights(const float* w, float* w2, int Cout, int C, int k, hipStream_t stream);
int ssn_s2d_weights_bwd(const float* dw2, float* dw, int Cout, int C, int k, hipStream_t stream);
/* dst[i] += src[i]: sums the conv-gradient buffers of the sub-batches when one backward is executed in chunks (a batch
 * whose activations exceed the 2 GiB a kernel operand can address; DataParallel's reduce of replica gradients). */
int ssn_add_inplace(float* dst, const float* src, long n, hipStream_t stream);
/* out[n][c][h][w] = h < Ho && w < Wo ? g[n][c][h][w] : 0 (planes H x W >= Ho x Wo): the output gradient of an UNPADDED
 * stride-1 convolution on its input's grid, which makes its weight gradient a same-grid problem with taps 0 .. k-1 behind
 * the pixel: ssn_conv_wgrad_x6_rect(kh, kw, pad 0, 0) on (out, x).  The values are g's: out shares g's amax slot. */
int ssn_embed_planes(const float* g, float* out, int N, int C, int Ho, int Wo, long g_img_stride, int H, int W,
                     long out_img_stride, hipStream_t stream);

/* ================================================================== planes tensors (ABI 5)
 * The MFMA-native activation format of the split-precision path (csrc/planes.h): an fp32 activation x of the backbone behind
 * /root/reference/ssn_models.py:266,298 is kept as TWO f16 terms of x * s -- hi = f16(x s), lo = f16(x s - hi), s a power of
 * two per tensor -- in the channel-blocked layout
 *     plane (hi | lo)  |  image n  |  channel group g = c / 8  |  pixel q  |  channel c % 8        (f16; 16 bytes per pixel & group)
 * produced by the kernel that computes x (conv / pool epilogues), so that no consumer converts operands in its inner loop.
 * Pointers `*_hi` / `*_lo` address the two planes at the first channel group of a channel slice (slices start at multiples of 8
 * channels), `*_img_groups` = channel groups (C / 8) of the whole tensor.  Scales are "delayed": `*_scale` points at the
 * tensor's scale slot, fixed while a step runs and derived from the magnitude recorded in the previous step;
 * `*_amax` at its amax slot, raised by every producer (ssn_pl_scales_update turns one into the other between steps and flags
 * tensors that outgrew their head-room).  Packed weights are those of ssn_conv_x6_pack_*. */
int ssn_pl_scales_update(float* amax, float* scale, int* flag, int n, int exact, int slot0, int odd_extra_bits, hipStream_t stream);
/* (slot0 = global index of the first of the n slots; slots with an ODD global index -- where the executor keeps its gradient tensors,
 * whose maxima move far more from step to step than an activation's -- get odd_extra_bits more bits of head-room.) */
/* The range guard of the delayed scales: one launch at the END of a forward / backward pass over the `n` slots of the executor.
 * flag[0] |= 1 if a tensor's recorded (pre-clamp) maximum did not fit the scale it was stored with (its values were clamped to the
 * f16 range), |= 2 if it fell more than 8 bits below the target range (precision draining).  Modifies no slot.  The reference has no
 * counterpart (cuDNN computes in fp32 storage); this is what lets loss.backward() / optimizer.step() (ssn_train.py:236,252) and the
 * alternating train / validate passes (ssn_train.py:191-253, 278-362) run on data whose magnitude changes from call to call: the
 * host repeats a flagged pass with fresh scales, ssn_sgd_step_multi(skip_flag) refuses to consume a flagged step's gradients. */
int ssn_pl_range_check(const float* amax, const float* scale, int* flag, int n, int odd_extra_bits, hipStream_t stream);
/* fp32 NCHW <-> planes (the caller's frames, test inputs, the fp32 feature boundary); s2d: the space-to-depth view of the stem. */
int ssn_pl_from_f32(const float* x, long x_img_stride, void* hi, void* lo, int N, int C, int H, int W, long img_groups, int s2d,
                    const float* scale, float* amax, hipStream_t stream);
/* im2col of a planes slice with few channels: y[n][c kh kw + r kw + s][ho][wo] = x[n][c][ho stride + r - pad_h][wo stride + s - pad_w]
 * (both planes copied, y carries x's scale): the weight gradient of a first convolution that is not run in space-to-depth form becomes a
 * 1x1 problem on C kh kw channels (cuDNN wgrad of the first layer behind ssn_train.py:236 for --arch InceptionV3). */
int ssn_pl_im2col(const void* x_hi, const void* x_lo, long x_img_groups, void* y_hi, void* y_lo, long y_img_groups, int N, int C, int H,
                  int W, int Ho, int Wo, int kh, int kw, int stride, int pad_h, int pad_w, hipStream_t stream);
int ssn_pl_to_f32(const void* hi, const void* lo, long img_groups, float* y, long y_img_stride, int N, int C, int HW,
                  const float* scale, hipStream_t stream);
/* conv + frozen-BN affine + ReLU (cuDNN conv / BN(eval) / ReLU behind ssn_models.py:266) on planes slices: any kh x kw taps,
 * stride 1 / 2; raw_from / row_split / row_gap as ssn_conv_x6_fwd (fused launch on an Inception block input).  tile_cfg < 0: the
 * heuristic; 0 .. ssn_conv_pl_tiles() - 1: that tile; 32 + c: the haloed kernel with tile c on 3x3 / stride 1 / pad 1 layers (each
 * input pixel is staged once per channel group instead of once per tap; ssn_conv_pl_halo_taken says whether the launch takes it,
 * a layer or tile it does not fit runs the plain kernel with tile c); 48 + c: the same with pixel tiles that do not cross images (the
 * 56 x 56 layer: a 128-pixel tile across two images does not fit the halo buffer; prepared in round 4, not yet measured). */
int ssn_conv_pl_fwd(const void* x_hi, const void* x_lo, const float* w_packed, const float* scale, const float* shift, void* y_hi,
                    void* y_lo, int N, int Cin, int H, int W, long x_img_groups, int Cout, int Ho, int Wo, long y_img_groups, int kh,
                    int kw, int stride, int pad_h, int pad_w, int relu, int tile_cfg, const float* x_scale, const float* y_scale,
                    float* y_amax, int raw_from, int row_split, int row_gap, hipStream_t stream);
/* data gradient (loss.backward(), ssn_train.py:236) of a stride-1 layer; mask_hi / mask_scale fuse the ReLU + frozen-BN backward
 * of the layer that produced the input (only the SIGN of its high plane is read); taps_reversed: pack mode 2 operand. */
int ssn_conv_pl_dgrad(const void* dy_hi, const void* dy_lo, const float* wt_packed, void* dx_hi, void* dx_lo, int N, int Cout,
                      int Ho, int Wo, long dy_img_groups, int Cin, int H, int W, long dx_img_groups, int kh, int kw, int pad_h,
                      int pad_w, int accumulate, const void* mask_hi, long mask_img_groups, const float* mask_scale, int tile_cfg,
                      const float* dy_scale, const float* dx_scale, float* dx_amax, int k_split, int k_gap, int taps_reversed,
                      hipStream_t stream);
/* ... of a 3x3 / stride-2 layer (pad 1 on an even input, or pad 0): four parity-class launches (ssn_conv_x6_pack_dgrad_s2). */
int ssn_conv_pl_dgrad_s2(const void* dy_hi, const void* dy_lo, const float* wt_packed, void* dx_hi, void* dx_lo, int N, int Cout,
                         int Ho, int Wo, long dy_img_groups, int Cin, int H, int W, long dx_img_groups, int pad, int accumulate,
                         const void* mask_hi, long mask_img_groups, const float* mask_scale, int tile_cfg, const float* dy_scale,
                         const float* dx_scale, float* dx_amax, hipStream_t stream);
/* weight + bias gradient (cuDNN wgrad behind ssn_train.py:236): the reduction index is the pixel, so the operands are read
 * with the LDS transpose read; tile_cfg >= 100 / < 0: the nine-tap kernel for 3x3 / stride 1 / pad 1 layers. */
int ssn_conv_wgrad_pl(const void* g_hi, const void* g_lo, const void* x_hi, const void* x_lo, float* dw, float* db, int N, int Cin,
                      int H, int W, long x_img_groups, int Cout, int Ho, int Wo, long g_img_groups, int kh, int kw, int stride,
                      int pad_h, int pad_w, void* workspace, long ws_bytes, int tile_cfg, const float* g_scale,
                      const float* x_scale, int g_row_split, int g_row_gap, int* deferred_reduce, hipStream_t stream);
/* deferred_reduce (HOST int[2], may be NULL): when given, the split-K slabs are left in `workspace` -- which then must stay untouched
 * until the caller has reduced them -- and {slabs, taps} for ssn_wgrad_reduce_multi are written there instead of launching the
 * reduction: a backward pass reduces the slabs of all its layers in one launch. */
int ssn_wgrad_reduce_multi(int count, const float* const* part, float* const* dw, float* const* db, const int* M, const int* K,
                           const int* splits, const int* taps, hipStream_t stream);
long ssn_conv_wgrad_pl_workspace_bytes(int N, int Cin, int Cout, int Ho, int Wo, int kh, int kw, int tile_cfg);
/* GROUPED weight gradients: every weight + bias gradient of a backward pass (all the cuDNN wgrad calls behind loss.backward(),
 * /root/reference/ssn_train.py:236) in at most five launches (one per kernel family that has problems) + one reduction.  Problem i takes the arguments of ssn_conv_wgrad_pl as
 * array entries (HOST arrays): plane pointers, dw[i], db[i] (may be NULL), scale pointers, shape[16 i ..] = {N, Cin, H, W, Cout, Ho,
 * Wo, kh, kw, stride, pad_h, pad_w, g_row_split, g_row_gap, tile hint, 0} (hint -1: chosen; 0 / 3 / 8: one-tap 64x64 / 128x128 /
 * 96x128; 100: nine taps; 200: chunked 1x1; 300: the 4x4-tap space-to-depth stem), groups[2 i ..] = {x_img_groups, g_img_groups}.  The problems travel in a device-resident
 * table (`table`: ssn_conv_wgrad_pl_group_table_bytes(count) bytes, written by the call); each kernel family runs ONE grid over all its
 * problems, items ordered longest first, the reduction ranges split only as far as the whole group needs; partial slabs
 * (`workspace`: ssn_conv_wgrad_pl_group_workspace_bytes bytes; plan_out, optional: [count][4] = family, variant, splits, units per
 * split) are reduced in a fixed order: deterministic.  Capturable, no host sync. */
int ssn_conv_wgrad_pl_group(int count, const void* const* g_hi, const void* const* g_lo, const void* const* x_hi,
                            const void* const* x_lo, float* const* dw, float* const* db, const int* shape, const long* groups,
                            const float* const* g_scale, const float* const* x_scale, void* workspace, long ws_bytes, void* table,
                            long table_bytes, hipStream_t stream);
long ssn_conv_wgrad_pl_group_workspace_bytes(int count, const int* shape, const long* groups, int* plan_out);
long ssn_conv_wgrad_pl_group_table_bytes(int count);
void ssn_conv_wgrad_pl_group_tuning(double fixed_nine, double fixed_one, int min_units_nine, int min_units_one); /* tooling / tests: planner constants (<= 0: keep) */
int ssn_conv_wgrad_pl_tiles(void);
int ssn_conv_pl_tiles(void);
int ssn_conv_pl_halo_taken(int N, int H, int W, int tile_cfg);
int ssn_conv_pl_tile_shape(int cfg, int* bm, int* bn);
void ssn_conv_pl_debug_flags(int flags);                    /* tooling (tools/ablate_conv_pl.py) */
void ssn_conv_pl_debug_trace(unsigned long long* buf);
void ssn_conv_wgrad_pl_debug_trace(unsigned long long* buf); /* tooling (tools/trace_wgrad_pl.py) */
void ssn_conv_wgrad_pl_debug_flags(int flags);
/* nn.MaxPool2d(ceil_mode) of the backbone manifest forward / backward (uint8 window-local argmax, torch's tie rule; dx_f32:
 * write the input gradient as fp32 NCHW instead), the 3x3 average pool behind its 1x1 projection (+ affine + ReLU; without
 * affine: its backward stencil), the ReLU / frozen-BN backward of a slice, global average pool forward / backward, and the
 * per-channel sums that give a projection's bias gradient. */
int ssn_pl_maxpool_fwd(const void* x_hi, const void* x_lo, long x_img_groups, void* y_hi, void* y_lo, long y_img_groups,
                       unsigned char* argmax, int N, int C, int H, int W, int Ho, int Wo, int k, int s, int pad,
                       const float* x_scale, const float* y_scale, float* y_amax, hipStream_t stream);
int ssn_pl_maxpool_bwd(const void* dy_hi, const void* dy_lo, long dy_img_groups, const unsigned char* argmax, void* dx_hi,
                       void* dx_lo, long dx_img_groups, int N, int C, int H, int W, int Ho, int Wo, int k, int s, int pad,
                       int accumulate, const void* mask_hi, long mask_img_groups, const float* mask_scale, int mask_pooled,
                       const float* dy_scale, const float* dx_scale, float* dx_amax, float* dx_f32, long dx_f32_img_stride,
                       hipStream_t stream);      /* mask_pooled: mask_hi = hi plane of the pool's OUTPUT (3x3 / stride 2, no accumulation) */
int ssn_pl_avgpool_affine(const void* x_hi, const void* x_lo, long x_img_groups, void* y_hi, void* y_lo, long y_img_groups,
                          const float* scale, const float* shift, int relu, int N, int C, int H, int W, int k, int pad,
                          const float* x_scale, const float* y_scale, float* y_amax, hipStream_t stream);
int ssn_pl_relu_bn_bwd(void* g_hi, void* g_lo, long g_img_groups, const void* y_hi, long y_img_groups, const float* scale, int N,
                       int C, int HW, const float* g_scale, float* g_amax, hipStream_t stream);
int ssn_pl_gap_fwd(const void* x_hi, const void* x_lo, long x_img_groups, float* y, int N, int C, int HW, const float* x_scale,
                   hipStream_t stream);
int ssn_pl_gap_bwd(const float* dy, void* dx_hi, void* dx_lo, long dx_img_groups, int N, int C, int HW, const void* mask_hi,
                   long mask_img_groups, const float* mask_scale, const float* dx_scale, float* dx_amax, hipStream_t stream);
long ssn_pl_channel_sum_workspace_bytes(int C);
int ssn_pl_channel_sum(const void* g_hi, const void* g_lo, long g_img_groups, float* out, int N, int C, int HW,
                       const float* g_scale, void* workspace, long ws_bytes, hipStream_t stream);
/* count slices of one pass in ONE pair of launches (host arrays; the same N; workspace: ssn_pl_channel_sum_workspace_bytes(sum C));
 * per slice bit-identical to ssn_pl_channel_sum. */
int ssn_pl_channel_sum_multi(int count, const void* const* g_hi, const void* const* g_lo, const long* g_img_groups, float* const* out,
                             int N, const int* C, const int* HW, const float* const* g_scale, void* workspace, long ws_bytes,
                             hipStream_t stream);
/* training-mode BatchNorm2d (+ ReLU) on planes slices (csrc/planes_bn.hip): bn_mode 'partial' / 'full' of ssn_models.py:95-105,
 * 156-174 on the planes executor; the mathematics of ssn_bn_train_* above.  C a multiple of 8; z is the convolution output WITHOUT
 * its bias; mean / invstd / running statistics / dgamma / dbeta in real units.  bwd: y_hi = HIGH plane of the layer's output (the
 * ReLU decision; NULL: no ReLU); dz goes to the planes slice (dz_hi, dz_lo) or, with dz_f32, to an fp32 NCHW tensor. */
long ssn_pl_bn_train_workspace_bytes(int C);
int ssn_pl_bn_train_stats(const void* z_hi, const void* z_lo, long z_img_groups, const float* z_scale, const float* conv_bias,
                          float* mean, float* invstd, float* running_mean, float* running_var, int N, int C, int HW, float eps,
                          float momentum, void* workspace, long ws_bytes, hipStream_t stream);
int ssn_pl_bn_train_apply(const void* z_hi, const void* z_lo, long z_img_groups, const float* z_scale, void* y_hi, void* y_lo,
                          long y_img_groups, const float* y_scale, float* y_amax, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, int relu, int N, int C, int HW, hipStream_t stream);
int ssn_pl_bn_train_bwd(const void* dy_hi, const void* dy_lo, long dy_img_groups, const float* dy_scale, const void* y_hi,
                        long y_img_groups, const void* z_hi, const void* z_lo, long z_img_groups, const float* z_scale,
                        const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta, void* dz_hi,
                        void* dz_lo, long dz_img_groups, const float* dz_scale, float* dz_amax, float* dz_f32,
                        long dz_f32_img_stride, int N, int C, int HW, void* workspace, long ws_bytes, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SSN_HIP_H */
