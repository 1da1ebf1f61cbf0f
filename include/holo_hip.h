/*
 * holo_hip.h — C-ABI of the MI355X (gfx950) holographic field-retrieval hot path.
 *
 * The reference (csleemooo/style_transfer_based_holographic_imaging) is pure Python on torch
 * `aten::` ops: it has no FFI.  Each entry point below therefore replaces the aten op sequence
 * of one reference function; the reference-side binding (a ctypes stub) is shown in
 * INTEGRATION.md.  Plain pointers, sizes and a hipStream_t only — no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; the library never allocates,
 *     never synchronises and never copies to the host: callers own all buffers (outputs and
 *     workspace included) and the stream.  `stream` is a hipStream_t passed as void*.
 *   - activations inside the network are NHWC fp32 ("channels last": pixel-major, channel
 *     contiguous); network inputs/outputs are the reference's NCHW planes.
 *   - return value 0 = launched; non-zero = rejected before any launch (bad shape/argument or
 *     a HIP launch error), with the reason available from holo_last_error().
 *   - thread-safety: entry points are re-entrant; holo_last_error() is per thread.
 */
#ifndef HOLO_HIP_H
#define HOLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HOLO_ABI_VERSION 5

/* error codes */
#define HOLO_OK 0
#define HOLO_ERR_ARG 1        /* bad pointer / shape / mode */
#define HOLO_ERR_UNSUPPORTED 2 /* valid request the kernels do not cover (e.g. transform side > 4096) */
#define HOLO_ERR_WORKSPACE 3  /* workspace too small */
#define HOLO_ERR_LAUNCH 4     /* hipGetLastError() after a launch */

int holo_abi_version(void);
const char* holo_last_error(void);
/* name of the device code object's target ("gfx950"); fails the load on any other build */
const char* holo_target_arch(void);
/* hash of the sources, compiler flags and ROCm version this library was built from (the build recipe passes it as
 * -DHOLO_SOURCE_HASH); the Python binding compares it with the hash of the sources next to it on every load, on
 * every rank, and refuses a stale library */
const char* holo_source_hash(void);
/* Test / A-B hook, not part of the reference's surface: force a kernel variant of the convolutions (0 = automatic, the
 * shipped behaviour; 1 = the round-1 kernels with the standard tile, 2 = small tile, 3 = tall bf16 wave tile, 4..7 = 8-wave
 * tiles of the bf16 kernel: all bit-identical to each other (tests/test_hip_bf16.py); 8 = the staged 16x16x32 bf16 kernel,
 * 9 = the weights-resident 64 -> 64 bf16 kernel (csrc/conv_bf16.hip): another summation grouping, within bf16 rounding of the
 * others).  Process-wide. */
int holo_debug_set_conv_tile(int tile);

/* ------------------------------------------------------------------------------------------
 * Weight packing (done once per module; reference keeps OIHW / IOHW tensors, net.py:33-130)
 * ---------------------------------------------------------------------------------------- */
/* Conv2d weight (Cout,Cin,3,3) -> [Cin/KC][tap=ky*3+kx][Cout][KC] fp32, Cin % 16 == 0; KC (16 or 32 channels per
 * pipeline stage) is chosen by the library from (Cin, Cout): pack and holo_conv3x3_f32 always agree */
int holo_pack_conv3x3_weight_f32(const float* w_oihw, float* w_packed, int cout, int cin, void* stream);
size_t holo_conv3x3_packed_elems(int cout, int cin);
/* ConvTranspose2d weight (Cin,Cout,2,2) -> [Cin/16][q=dy*2+dx][Cout][16] fp32, Cin % 16 == 0 */
int holo_pack_convt2x2_weight_f32(const float* w_iohw, float* w_packed, int cin, int cout, void* stream);
size_t holo_convt2x2_packed_elems(int cin, int cout);

/* ------------------------------------------------------------------------------------------
 * Encoder / decoder convolutions
 * ---------------------------------------------------------------------------------------- */
/* vgg[0:4]  (net.py:77-80): Conv2d(1,3,1x1) -> ReflectionPad2d(1) -> Conv2d(3,64,3x3) -> ReLU.
 * x: (B,1,H,W) plane(s), contiguous;  w0,b0: the 1x1 conv (3 + 3 floats);
 * w1: the 3x3 conv weight in its native (64,3,3,3) OIHW order; b1: 64;  y: NHWC (B,H,W,64). */
int holo_conv_first_f32(const float* x, const float* w0, const float* b0, const float* w1,
                        const float* b1, float* y_nhwc, int batch, int h, int w, void* stream);
/* the same layers fed with the hologram INTENSITY: the sqrt of the entry loop (test_field_retrieval_mnist.py:119) applied by the
 * loader, correctly rounded; bit-identical to a correctly rounded sqrt followed by holo_conv_first_* */
int holo_conv_first_sqrt_f32(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                             float* y_nhwc, int batch, int h, int w, void* stream);
int holo_conv_first_sqrt_bf16(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                              uint16_t* y_nhwc, int batch, int h, int w, void* stream);
int holo_conv_first_sqrt_f32s(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                              void* y_nhwc, int batch, int h, int w, void* stream);

/* ReflectionPad2d(1) -> Conv2d(Cin,Cout,3x3) -> [ReLU] -> [MaxPool2d(2,2,ceil_mode=True)]
 * (every "c3" layer of net.py:33-107).  x: NHWC (B,H,W,Cin); w_packed from
 * holo_pack_conv3x3_weight_f32; y: NHWC (B,H,W,Cout) or (B,ceil(H/2),ceil(W/2),Cout) with pool.
 * Requires H,W >= 2, Cin % 16 == 0, Cout % 64 == 0.  fp32 MFMA (v_mfma_f32_32x32x2_f32). */
int holo_conv3x3_f32(const float* x_nhwc, const float* w_packed, const float* bias, float* y_nhwc,
                     int batch, int h, int w, int cin, int cout, int relu, int pool, void* stream);

/* The same layers (ReflectionPad2d(1) -> Conv2d(3x3) -> [ReLU] [-> MaxPool2d(2,2,ceil)], net.py:78-107 / 34-71) as
 * Winograd F(2x2,3x3) in exact fp32 arithmetic on the fp32 matrix cores: 2.25x fewer multiplications; results differ
 * from holo_conv3x3_f32 by fp32 rounding only (both ~1e-5 of the network's outputs away from a float64 evaluation).
 * Weights: (Cout,Cin,3,3) -> G g G^T (float64, rounded once) as [Cout/32][Cin/16][4x4 position][32][16] fp32,
 * 16*Cout*Cin floats.  Cin % 16 == 0 and Cin >= 32, Cout % 32 == 0; shapes and flags as holo_conv3x3_f32. */
int holo_pack_conv3x3_wino_weight_f32(const float* w_oihw, float* w_packed, int cout, int cin, void* stream);
size_t holo_conv3x3_wino_packed_elems(int cout, int cin);
int holo_conv3x3_wino_f32(const float* x_nhwc, const float* w_packed, const float* bias, float* y_nhwc,
                          int batch, int h, int w, int cin, int cout, int relu, int pool, void* stream);
/* The same layer for launches too small to fill the chip (small batches, the deep low-resolution layers: the bundled
 * 5 x 128^2 case has 80 tiles for 256 CUs): split-K.  The Cin/16 channel chunks of every tile are dealt to `ksplit` work items
 * (ksplit in [2,8]; item s sums chunks [s*n/ksplit, (s+1)*n/ksplit) of the n = Cin/16, every part at least 2); each writes raw partial sums into `workspace`
 * (holo_conv3x3_wino_splitk_workspace_bytes = ksplit * B*H*W*Cout floats) and a second launch adds the partials in the fixed
 * order 0, 1, ... with bias, ReLU and pooling — deterministic and independent of the batch a hologram arrives in.  Differs from
 * holo_conv3x3_wino_f32 by fp32 rounding (the summation over channels is re-associated). */
size_t holo_conv3x3_wino_splitk_workspace_bytes(int batch, int h, int w, int cout, int ksplit);
int holo_conv3x3_wino_splitk_f32(const float* x_nhwc, const float* w_packed, const float* bias, float* y_nhwc,
                                 int batch, int h, int w, int cin, int cout, int relu, int pool, int ksplit,
                                 float* workspace, size_t workspace_bytes, void* stream);

/* ConvTranspose2d(Cin,Cout,k=2,s=2) -> [ReLU] (net.py:40,55,64).  x: NHWC (B,H,W,Cin);
 * y: NHWC (B,2H,2W,Cout).  Cin % 16 == 0, Cout % 64 == 0. */
int holo_convt2x2_f32(const float* x_nhwc, const float* w_packed, const float* bias, float* y_nhwc,
                      int batch, int h, int w, int cin, int cout, int relu, void* stream);

/* decoder[36:38] (net.py:72-73): ReflectionPad2d(1) -> Conv2d(64,2,3x3), no activation.
 * x: NHWC (B,H,W,64); w: native (2,64,3,3) OIHW; y: NCHW (B,2,H,W): plane 0 amplitude,
 * plane 1 phase (net.py:250-251). */
int holo_conv_last_f32(const float* x_nhwc, const float* w_oihw, const float* bias, float* y_nchw,
                       int batch, int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------
 * Feature statistics and AdaIN (function.py:4-23, net.py:235-246)
 * ---------------------------------------------------------------------------------------- */
/* calc_mean_std: per (n,c) mean and sqrt(unbiased_var + eps) over HW.  x: NHWC (B,HW,C);
 * mean,std: (B,C).  C % 4 == 0, HW >= 2. */
int holo_channel_mean_std_f32(const float* x_nhwc, float* mean, float* std, int batch, int hw,
                              int c, float eps, void* stream);
/* t = (x-mean)/std*style_std + style_mean ; y = alpha*t + (1-alpha)*x   (net.py:241-246).
 * style_mean/style_std: (B or 1, C) with `style_batch_stride` floats between samples
 * (0 = one style broadcast over the batch: the single-style method, net.py:237-238). */
int holo_adain_apply_f32(const float* x_nhwc, const float* mean, const float* std,
                         const float* style_mean, const float* style_std, long style_batch_stride,
                         float alpha, float* y_nhwc, int batch, int hw, int c, void* stream);
/* calc_mean_std + AdaIN (function.py:4-12 then net.py:240-246) in ONE launch: mean / std (B,c) written as by
 * holo_channel_mean_std_* (the distance head reads them), y as by holo_adain_apply_*; bit-identical to the two calls */
int holo_adain_fused_f32(const float* x_nhwc, float* mean, float* std, const float* style_mean, const float* style_std,
                         long style_batch_stride, float alpha, float* y_nhwc, int batch, int hw, int c, float eps,
                         void* stream);
int holo_adain_fused_bf16(const uint16_t* x_nhwc, float* mean, float* std, const float* style_mean, const float* style_std,
                          long style_batch_stride, float alpha, uint16_t* y_nhwc, int batch, int hw, int c, float eps,
                          void* stream);
int holo_adain_fused_f32s(const void* x_nhwc, float* mean, float* std, const float* style_mean, const float* style_std,
                          long style_batch_stride, float alpha, void* y_nhwc, int batch, int hw, int c, float eps,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * Distance head (net.py:288-308)
 * ---------------------------------------------------------------------------------------- */
/* y[b][n] = bias[n] + sum_k x[b][k]*w[n][k]   (nn.Linear; w is (N,K) row-major) */
int holo_linear_f32(const float* x, const float* w, const float* bias, float* y, int batch, int k,
                    int n, void* stream);
/* InstanceNorm1d on a (B,N) input = per-row (x-mean)/sqrt(biased_var+eps), then ReLU */
int holo_rownorm_relu_f32(const float* x, float* y, int batch, int n, float eps, void* stream);
/* y = sigmoid(bias + x . w) for the final Linear(512,1) */
int holo_linear1_sigmoid_f32(const float* x, const float* w, const float* bias, float* y, int batch,
                             int k, void* stream);

/* The whole head (Distance_G.forward in eval mode, net.py:288-308) on the statistics (B,c) `mean`, `std` in four launches: the input
 * concatenation is read in place and every InstanceNorm1d + ReLU is applied by the Linear that consumes it (each wave holds its input
 * rows in registers and computes their mean / biased variance itself).  Layer shapes as in the reference: w1, w2 (2c,2c), w3 (c,2c),
 * wo (1,c), row-major; out: (B); workspace: holo_distance_head_workspace_bytes = 2 * B * 2c floats.  c % 256 == 0, 2c <= 1024 (the
 * reference has c = 512); other widths take the three per-layer entry points above.  Differs from them by fp32 rounding only (the
 * statistics are summed in a different order). */
size_t holo_distance_head_workspace_bytes(int batch, int c);
int holo_distance_head_f32(const float* mean, const float* std, const float* w1, const float* b1, const float* w2, const float* b2,
                           const float* w3, const float* b3, const float* wo, const float* bo, float* out, int batch, int c, float eps,
                           float* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Angular-spectrum propagation (utils/Angular_Spectrum_Method.py:7-53 + the field packing and
 * abs/angle/intensity epilogues of utils/Forward_model.py:22-39)
 * ---------------------------------------------------------------------------------------- */
#define HOLO_ASM_IN_AMP_PHASE 0 /* in0 = amplitude planes, in1 = phase planes (fp32); O = a*exp(i*phase_scale*p) */
#define HOLO_ASM_IN_COMPLEX 1   /* in0 = interleaved complex64 fields, in1 ignored */
#define HOLO_ASM_OUT_AMP_PHASE 0 /* out0 = |H|, out1 = angle(H)   (Forward_model.py:28-29) */
#define HOLO_ASM_OUT_INTENSITY 1 /* out0 = |H|^2                  (Forward_model.py:39) */
#define HOLO_ASM_OUT_COMPLEX 2   /* out0 = interleaved complex64   (Forward_model.py:36-37) */

/* The float64 phase constants holo_asm_propagate takes, from float32 distances in the reference's units, with the reference's
 * rounding chain (utils/Forward_model.py:18: ((d + add) * mul) * 1e-3 in fp32; utils/Angular_Spectrum_Method.py:29: 1j*2*pi*d is
 * complex64, so the product with 2 pi is an fp32 product, promoted to float64 by the grid): out[i] for i < n from
 * d[(i % period) * d_stride].  One launch instead of five element-wise ones in front of every propagation. */
int holo_asm_phase_constants_f32(const float* d, long d_stride, int period, float add, float mul, double* out, int n, void* stream);
/* bytes of scratch holo_asm_propagate needs (two S x P complex64 slabs per field / output) */
size_t holo_asm_workspace_bytes(int n_fields, int n_dist, int s, int zero_padding);

/* Propagates n_fields square S x S fields, each to n_dist distances (output field index =
 * field*n_dist + dist).  phase_const[field*n_dist+dist] is the float64 constant c of
 * theta = c * sqrt(max(0, 1 - l^2 (fx^2+fy^2))) / l, i.e. 2*pi*d with the reference's rounding
 * (Angular_Spectrum_Method.py:29) — computed by the caller.
 * in*_stride / out*_stride: elements (floats, or complex64 for the complex modes) between
 * consecutive fields, so strided views such as g_t[:, :1] (net.py:250-251) need no copy.
 * zero_padding != 0: replicate-pad to P = 2S (Angular_Spectrum_Method.py:11-14); else P = S.
 * P a power of two in [16, 2048]: fused in-LDS FFT kernels.  Any other 2 <= P <= 4096 (zero_padding needs an
 * even S, as in the reference): direct-DFT kernels, O(P^3) per field — correct, not fast. */
int holo_asm_propagate(const void* in0, const void* in1, int in_mode, long in0_stride, long in1_stride,
                       void* out0, void* out1, int out_mode, long out0_stride, long out1_stride,
                       const double* phase_const, int n_fields, int n_dist, int s, int zero_padding,
                       double wavelength, double pixel_size, float phase_scale,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optional "f32s" storage for the conv stack (same reference lines as the _f32 entry points above).
 * An fp32 value is kept as two fp16 halves, a = hi + lo' * 2^-11 (22-23 significant bits), planar per
 * 16 consecutive channels: [16 x hi | 16 x lo'] = 64 bytes per group, 4 bytes per element, passed as
 * void*.  Activations are NHWC with C % 16 == 0 in 64-byte aligned buffers.  Products are formed as
 * hi*hi + (hi*lo' + lo'*hi) * 2^-11 on v_mfma_f32_32x32x16_f16 with fp32 accumulation and the same
 * per-chunk fold as the fp32 kernel: measured 2.3e-7 of float64 per layer (fp32 MFMA: 3.3e-7) at 2.3x
 * the fp32 kernel's rate.  Not bit-compatible with fp32 arithmetic (the dropped lo'*lo' term, fp16
 * range: |a| saturates at 65504), hence opt-in and never the default.
 * ---------------------------------------------------------------------------------------- */
int holo_cast_f32_to_f32s(const float* x, void* y, long n, void* stream);   /* n % 16 == 0, 64-byte aligned buffers */
int holo_cast_f32s_to_f32(const void* x, float* y, long n, void* stream);
int holo_pack_conv3x3_weight_f32s(const float* w_oihw, void* w_packed, int cout, int cin, void* stream);   /* Cin % 32 == 0 */
int holo_pack_convt2x2_weight_f32s(const float* w_iohw, void* w_packed, int cin, int cout, void* stream);
int holo_conv_first_f32s(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                         void* y_nhwc, int batch, int h, int w, void* stream);
int holo_conv3x3_f32s(const void* x_nhwc, const void* w_packed, const float* bias, void* y_nhwc, int batch, int h,
                      int w, int cin, int cout, int relu, int pool, void* stream);
int holo_convt2x2_f32s(const void* x_nhwc, const void* w_packed, const float* bias, void* y_nhwc, int batch, int h,
                       int w, int cin, int cout, int relu, void* stream);
int holo_conv_last_f32s(const void* x_nhwc, const float* w_oihw, const float* bias, float* y_nchw, int batch, int h,
                        int w, void* stream);
int holo_channel_mean_std_f32s(const void* x_nhwc, float* mean, float* std, int batch, int hw, int c, float eps,
                               void* stream);
int holo_adain_apply_f32s(const void* x_nhwc, const float* mean, const float* std, const float* style_mean,
                          const float* style_std, long style_batch_stride, float alpha, void* y_nhwc, int batch,
                          int hw, int c, void* stream);

/* ------------------------------------------------------------------------------------------
 * 2-D phase unwrapping (utils/functions.py:44-59 `unwrap`, reached with unwrap=True from
 * utils/Forward_model.py:30-31 and test_field_retrieval_mnist.py:126).  The reference loops over the
 * images on the host and calls scikit-image's unwrap_phase (Herraez et al. 2002, reliability-sorted
 * non-continuous path); this entry point builds the same minimum spanning tree of edge reliabilities
 * on the device (Boruvka rounds) for all images at once.  Global multiple of 2 pi per image: the most
 * frequent integer offset is 0.  Border pixels use the constant reliability 9999999 where the library
 * adds rand(); ties are ordered by edge id (csrc/unwrap.hip, oracle/unwrap_oracle.py).
 * ---------------------------------------------------------------------------------------- */
/* bytes of 8-byte aligned scratch for holo_unwrap2d_f32 (24 per pixel) */
size_t holo_unwrap2d_workspace_bytes(int batch, int h, int w);

/* phase/out: `batch` images of h x w fp32, row-major, in_stride / out_stride floats apart (>= h*w);
 * out may not alias phase.  h*w <= 2^24, batch <= 65535. */
int holo_unwrap2d_f32(const float* phase, long in_stride, float* out, long out_stride, int batch, int h, int w,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * bf16 variants (BASELINE config 3: "bf16 convs / fp32 FFT").  Same layouts and semantics as the _f32 entry
 * points; activations and packed weights are bf16 (uint16_t storage), bias, statistics, accumulation
 * (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16) and the last layer's amplitude/phase planes stay fp32.
 * Cin % 32 == 0.  For batches of 13 holograms or more (the open-ended batch bucket) and images large enough to fill the
 * chip, the conv entry points take the 16x16x32 kernels of csrc/conv_bf16.hip — chosen from per-image quantities, so that
 * every batch >= 13 gives a hologram the same bits; they differ from the 32x32x16 kernels by the rounding of partial
 * sums (rare one-ulp differences of the bf16 outputs).  holo_conv_last_bf16 keeps the fp32 weights exactly in both forms.
 * ---------------------------------------------------------------------------------------- */
int holo_pack_conv3x3_weight_bf16(const float* w_oihw, uint16_t* w_packed, int cout, int cin, void* stream);
int holo_pack_convt2x2_weight_bf16(const float* w_iohw, uint16_t* w_packed, int cin, int cout, void* stream);
int holo_conv_first_bf16(const float* x, const float* w0, const float* b0, const float* w1, const float* b1,
                         uint16_t* y_nhwc, int batch, int h, int w, void* stream);
int holo_conv3x3_bf16(const uint16_t* x_nhwc, const uint16_t* w_packed, const float* bias, uint16_t* y_nhwc,
                      int batch, int h, int w, int cin, int cout, int relu, int pool, void* stream);
int holo_convt2x2_bf16(const uint16_t* x_nhwc, const uint16_t* w_packed, const float* bias, uint16_t* y_nhwc,
                       int batch, int h, int w, int cin, int cout, int relu, void* stream);
int holo_conv_last_bf16(const uint16_t* x_nhwc, const float* w_oihw, const float* bias, float* y_nchw,
                        int batch, int h, int w, void* stream);
int holo_channel_mean_std_bf16(const uint16_t* x_nhwc, float* mean, float* std, int batch, int hw, int c,
                               float eps, void* stream);
int holo_adain_apply_bf16(const uint16_t* x_nhwc, const float* mean, const float* std, const float* style_mean,
                          const float* style_std, long style_batch_stride, float alpha, uint16_t* y_nhwc,
                          int batch, int hw, int c, void* stream);
int holo_cast_f32_to_bf16(const float* x, uint16_t* y, long n, void* stream);
int holo_cast_bf16_to_f32(const uint16_t* x, float* y, long n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Layout helpers
 * ---------------------------------------------------------------------------------------- */
int holo_nhwc_to_nchw_f32(const float* x, float* y, int batch, int hw, int c, void* stream);
int holo_nchw_to_nhwc_f32(const float* x, float* y, int batch, int hw, int c, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward pass of the training objective (ABI 5; Net.forward, net.py:199-226 with the losses of net.py:186-197: the
 * reference differentiates it with torch.autograd over aten ops).  fp32, NHWC, csrc/backward.hip.  The data gradient
 * of ReflectionPad2d(1) + Conv2d(3x3) (net.py:34-71, 78-107) is  holo_relu_pool_bwd_pad_f32 (border 2)  ->  the
 * FORWARD convolution (holo_conv3x3_f32 / holo_conv3x3_wino_f32) with the flipped, transposed weights  ->
 * holo_fold_reflect_f32; its weight gradient is holo_pix_gemm_f32 (taps = 9), its bias gradient holo_channel_sum_f32.
 * ---------------------------------------------------------------------------------------- */
/* MaxPool2d(2,2,ceil_mode=True) on NHWC, unfused (training keeps the pre-pool activation): (B,h,w,c) -> (B,ceil(h/2),ceil(w/2),c) */
int holo_maxpool2x2_f32(const float* x_nhwc, float* y_nhwc, int batch, int h, int w, int c, void* stream);
/* out_padded (B, h+2*border, w+2*border, c) = the gradient of the pre-activation, zero outside the interior:
 * dy routed through the max-pool (y_pool != NULL: dy and y_pool are (B,ceil(h/2),ceil(w/2),c); the first maximum of a window
 * in scan order takes the gradient, as torch's stored argmax) and masked by the ReLU (relu != 0: y_act > 0); y_act (B,h,w,c) is
 * the layer's post-ReLU, pre-pool output */
/* pitch_rows: rows from one image's block of out_padded to the next; 0 = h + 2*border (separate blocks).  Less (down to
 * h + border) = the TALL layout: neighbouring images share their zero rows, and the batch is one image of
 * (B-1)*pitch_rows + h + 2*border rows for the convolution that follows (no per-image round-up to whole tiles).
 * per_row > 1 (needs border >= 1): a MOSAIC — that many images side by side as well, w + border columns apart (shared zero
 * columns); image n at cell (n / per_row, n % per_row), cells past the batch zero; out_padded is then
 * (ceil(B/per_row)-1)*pitch_rows + h + 2*border rows of (per_row-1)*(w+border) + w + 2*border pixels.  0 / 1: one image per row */
/* dbias (c) or NULL: the bias gradient = sum of out_padded over its pixels, accumulated in the same pass (float64 partial sums,
 * fixed order); needs a workspace of holo_relu_pool_bwd_pad_workspace_bytes(c) */
size_t holo_relu_pool_bwd_pad_workspace_bytes(int c);
int holo_relu_pool_bwd_pad_f32(const float* dy, const float* y_act, const float* y_pool, float* out_padded, int batch,
                               int h, int w, int c, int relu, int border, int pitch_rows, int per_row, float* dbias,
                               void* workspace, size_t workspace_bytes, void* stream);
/* adjoint of ReflectionPad2d(1): dx (B,h,w,c) = interior of dxp (B,h+2*border,w+2*border,c) plus its pad ring (padded row -1
 * onto row 1, row h onto row h-2, likewise columns); border = 1 or 2; pitch_rows / per_row as above (0 = separate blocks) */
int holo_fold_reflect_f32(const float* dxp, float* dx, int batch, int h, int w, int c, int border, int pitch_rows,
                          int per_row, void* stream);
/* y (B,h,w,cout) = x (B,h,w,cin) . W[cin][cout] + bias [ReLU]: a plain 1x1 convolution on the one-tap MFMA GEMM of the transposed
 * convolution (net.py:46-65 backward: the data gradient of ConvTranspose2d(k2,s2) is a 1x1 convolution of the space-to-depth
 * gradient).  w_packed = holo_pack_convt2x2_weight_f32 of the (cin, cout/4, 2, 2) tensor T[k][c][dy][dx] = W[k][(2dy+dx)*cout/4 + c];
 * cin % 16 == 0, cout % 64 == 0 */
int holo_conv1x1_f32(const float* x_nhwc, const float* w_packed, const float* bias, float* y_nhwc, int batch, int h, int w,
                     int cin, int cout, int relu, void* stream);
/* ConvTranspose2d(k2,s2)+ReLU backward helper: g (B,h,w,4c), g[..,q*c+ch] = dy[n,2Y+dy,2X+dx,ch] * (y > 0), q = dy*2+dx */
int holo_space_to_depth2_relu_f32(const float* dy, const float* y, float* g, int batch, int h, int w, int c, int relu,
                                  void* stream);
/* out[c] = sum over `rows` rows of x (rows, c); c % 64 == 0 (bias gradients) */
size_t holo_channel_sum_workspace_bytes(long rows, int c);
int holo_channel_sum_f32(const float* x, float* out, long rows, int c, void* workspace, size_t workspace_bytes, void* stream);
/* out[m][n][tap] = sum over the B*h*w pixels of P[pixel][m] * Q[reflect(pixel + tap)][n]: weight gradients as a GEMM whose K runs
 * over pixels (v_mfma_f32_32x32x2_f32, deterministic).  P (B,h+2*p_border,w+2*p_border,mc) read at its interior, Q (B,h,w,nc),
 * taps = 9 (tap = ky*3+kx, reflected neighbour: out is the OIHW gradient for m = Cout, n = Cin) or 1; mc, nc % 32 == 0;
 * p_pitch_rows / p_per_row: the layout of P as written by holo_relu_pool_bwd_pad_f32 (0 / 0 = separate blocks) */
size_t holo_pix_gemm_workspace_bytes(int batch, int h, int w, int mc, int nc, int taps);
int holo_pix_gemm_f32(const float* p, int p_border, int p_pitch_rows, int p_per_row, const float* q, float* out, int batch,
                      int h, int w, int mc, int nc, int taps, void* workspace, size_t workspace_bytes, void* stream);
/* decoder[36:38] (net.py:72-73) backward: dy (B,2,h,w) planes, x (B,h,w,64) the layer's input; dxp (B,h+2,w+2,64) = data gradient
 * on the padded domain (fold with holo_fold_reflect_f32, border 1) or NULL; dw (2,64,3,3) + db (2) or NULL */
size_t holo_conv_last_bwd_workspace_bytes(int batch, int h, int w);
int holo_conv_last_bwd_f32(const float* dy, const float* x_nhwc, const float* w_oihw, float* dxp, float* dw, float* db,
                           int batch, int h, int w, void* workspace, size_t workspace_bytes, void* stream);
/* vgg[0:4] (net.py:77-80) backward, data gradient only (the encoder is frozen, net.py:167-170): da (B,h,w,64) the masked gradient
 * of the ReLU'd output, din (B,1,h,w); workspace of B*(h+2)*(w+2)*4 floats */
int holo_conv_first_dgrad_f32(const float* da, const float* w0, const float* w1, float* din, int batch, int h, int w,
                              void* workspace, size_t workspace_bytes, void* stream);
/* calc_mean_std (function.py:4-12) of an NHWC fp32 map with the pixels split over blocks: the large maps of the training losses
 * (net.py:191-197) at small batch, where holo_channel_mean_std_f32's one block per (64 channels, image) leaves the chip idle.
 * Same formulas; float64 partial sums of 1024 pixels added in part order.  c % 64 == 0, hw >= 2 */
size_t holo_channel_mean_std_wide_workspace_bytes(int batch, int hw, int c);
int holo_channel_mean_std_wide_f32(const float* x_nhwc, float* mean, float* std, int batch, int hw, int c, float eps,
                                   void* workspace, size_t workspace_bytes, void* stream);
/* calc_mean_std (function.py:4-12) backward: dx[p][c] = dmean[c]/HW + dstd[c]*(x[p][c]-mean[c])/((HW-1)*std[c]) */
int holo_mean_std_bwd_f32(const float* x_nhwc, const float* mean, const float* stdv, const float* dmean, const float* dstd,
                          float* dx, int batch, int hw, int c, void* stream);
/* Distance_G (net.py:288-308) backward, layer by layer: Linear (dx and / or dw + db), per-row norm + ReLU (z = the Linear's
 * output), sigmoid */
int holo_linear_bwd_f32(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int batch, int fin,
                        int fout, void* stream);
/* nn.Dropout in training mode (net.py:298-304, on each Linear's output): y = x * keep * scale with the drawn 0/1 mask `keep` and
 * scale = 1/(1-p); the same call carries the gradient back */
int holo_mask_scale_f32(const float* x, const float* keep, float scale, float* y, long n, void* stream);
int holo_rownorm_relu_bwd_f32(const float* z, const float* dy, float* dz, int batch, int f, float eps, void* stream);
int holo_sigmoid_bwd_f32(const float* y, const float* dy, float* dz, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HOLO_HIP_H */
