/*
 * maua_hip.h — C ABI of libmaua_hip.so, the MI355X (gfx950) native library behind the audio-reactive StyleGAN2
 * inference path.  Plain pointers and sizes only; every pointer is a DEVICE pointer unless named h_*;
 * `stream` is a hipStream_t passed as void*.  Every launcher is asynchronous on `stream`, never
 * synchronises, never allocates, and returns 0 on success, a hipError_t (> 0) from the launch, or a
 * negative MAUA_E* code for rejected arguments.  All tensors are contiguous fp32 NCHW unless stated.
 *
 * Each entry point names the reference interface (file:line under /root/reference) it replaces.
 */
#ifndef MAUA_HIP_H
#define MAUA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAUA_EINVAL (-22)
#define MAUA_ENOSYS (-38)

/* ABI version of this header; bumped on any signature change. */
int maua_abi_version(void);  /* 5: + the low-resolution entries (maua_*_lowres_*), maua_const_styledconv_f32, maua_torgb_f32's plane-sum form; 4: the style fold (post_s arguments, s == NULL; round 6); 3: + maua_upconv_blur_f32 (round 5); 2: frame source (maua_frame_source_t) arguments; no tuning entry */
/* Number of compute units / name of device 0 (diagnostics for bench.py). */
int maua_device_info(int* cu_count, int* lds_bytes, char* name, int name_len);

/* ------------------------------------------------------------------------------------------------ ops
 * Replaces pybind `upfirdn2d.upfirdn2d(input[major,in_h,in_w,minor], kernel[kh,kw], up_x, up_y, down_x,
 * down_y, pad_x0, pad_x1, pad_y0, pad_y1)` — op/upfirdn2d.cpp:12-22, op/upfirdn2d_kernel.cu:209-369.
 * y must hold major*out_h*out_w*minor floats, out = (in*up + pad0 + pad1 - k)/down + 1 (floor).
 * `k` is the un-flipped [kh,kw] tap matrix in DEVICE memory (true convolution, as the reference).
 * Every argument combination is served; tiled HBM-rate kernels exist for the shapes the reference's own kernel tiles (modes 1-6 of
 * op/upfirdn2d_kernel.cu:313-359), with minor == 1 and planes below 2 GiB: up = down = 1 with 2..4 square taps (Blur), up = 2 / down = 1
 * and up = 1 / down = 2 with up to 4 x 4 taps (Upsample, Downsample), any pads incl. negative ones; the rest takes a one-thread-per-output gather. */
int maua_upfirdn2d_f32(const float* x, const float* k, float* y, int major, int in_h, int in_w, int minor,
                       int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                       int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* The same op for half / double tensors — the reference dispatches half, float and double (upfirdn2d_kernel.cu:313-359,
 * fused_bias_act_kernel.cu:79).  Generic kernels (fp32 accumulation for half); the fp32 entries above are the tuned path. */
int maua_upfirdn2d_f16(const void* x, const void* k, void* y, int major, int in_h, int in_w, int minor, int kh, int kw, int up_x,
                       int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);
int maua_upfirdn2d_f64(const void* x, const void* k, void* y, int major, int in_h, int in_w, int minor, int kh, int kw, int up_x,
                       int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* Replaces pybind `fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)` —
 * op/fused_bias_act.cpp:11-20, op/fused_bias_act_kernel.cu:18-98.  b == NULL or size_b == 0: no bias;
 * ref == NULL: no reference tensor.  bias index = (i / step_b) % size_b. In-place (y == x) is allowed. */
int maua_fused_bias_act_f32(const float* x, const float* b, const float* ref, float* y, int64_t size_x,
                            int size_b, int step_b, int act, int grad, float alpha, float scale, void* stream);

int maua_fused_bias_act_f16(const void* x, const void* b, const void* ref, void* y, int64_t size_x, int size_b, int step_b,
                            int act, int grad, float alpha, float scale, void* stream);
int maua_fused_bias_act_f64(const void* x, const void* b, const void* ref, void* y, int64_t size_x, int size_b, int step_b,
                            int act, int grad, float alpha, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------ frame source
 * Per-frame inputs of a generator forward, read THROUGH device memory instead of through kernel arguments, so that one
 * captured hipGraph serves every batch of a render (and every render): the sequences of render.py:140-182 — latents
 * [n_frames, n_latent, style_dim], truncation [n_frames], one noise map sequence per noise slot — stay resident in HBM and a
 * replay only moves `frame0` (maua_frame_source_seek: one 4-byte device write; the reference re-uploads every sequence slice
 * from pinned host memory per batch).  The struct lives in DEVICE memory; the host fills the pointers once per render.
 * Every launcher below that takes `src` ignores its own latents / trunc / noise / noise_batch_stride arguments when src != NULL:
 *   latent of sample b   = src->latents + (frame0 + b) * n_latent * style_dim
 *   truncation of b      = src->trunc[frame0 + b]            (src->trunc == NULL: no lerp)
 *   noise map of b       = src->noise[slot] + (frame0 + b) * src->noise_stride[slot]   (stride 0: one shared map, e.g. a
 *                          checkpoint's noises.noise_i buffer; src->noise[slot] == NULL: no noise) */
#define MAUA_MAX_NOISE_SLOTS 32
typedef struct {
    int32_t frame0;                               /* first frame of this launch inside every per-frame sequence */
    int32_t pad_;
    const float* latents;
    const float* trunc;
    const float* noise[MAUA_MAX_NOISE_SLOTS];
    int64_t noise_stride[MAUA_MAX_NOISE_SLOTS];   /* floats between the maps of consecutive frames */
} maua_frame_source_t;
/* src->frame0 = frame0, asynchronously on `stream` (a 4-byte hipMemsetD32Async: no host buffer has to outlive the call). */
int maua_frame_source_seek(maua_frame_source_t* src, int frame0, void* stream);

/* ------------------------------------------------------------------------------------------------ generator layers
 * THE STYLE FOLD (ABI 4).  ModulatedConv2d multiplies its input by the per-sample styles before the shared-weight contraction
 * (models/stylegan2.py:220-221, w = scale * W * s  <=>  conv(scale * W, x * s)).  Every feature map of the generator has exactly one
 * modulated 3x3 convolution as consumer (its ToRGB is computed in the producer's epilogue from the un-scaled value), so the multiply can
 * move into the PRODUCER's epilogue — once per element instead of once per element AND output-channel tile of the consumer's K loop:
 *   producers (maua_blur_noise_act_f32, maua_upconv_blur_f32, maua_styledconv_torgb_f32 and _partial_f32 in mode 5) take `post_s`
 *     = the CONSUMER's styles [B, stride] indexed by the producer's output channel, NULL = store the map as it is:
 *       y_stored[b,o] = y[b,o] * post_s[b * stride + o];
 *   consumers (maua_modconv3x3_f32 with up == 5 or 6, maua_upconv_blur_f32, the two ToRGB-fused entries in mode 5) accept s == NULL
 *     = "x arrives multiplied by my styles": kernel instances without the style multiplies in their K loop (4 of 16 transform
 *     instructions per window of the 2-D Winograd kernel, 9 of 23 per K step of the F(2,2)^2 transposed kernel).  d (demodulation)
 *     is unaffected: it is computed from the styles by maua_demod_f32 either way.
 * MAUA_ENOSYS for s == NULL / post_s != NULL in the other modes.  The caller keeps the un-folded form wherever something else reads
 * the feature map (a network bend on that layer id, return_activation_maps).
 *
 * Fused Blur -> NoiseInjection -> FusedLeakyReLU tail of an up-sampling StyledConv
 * (models/stylegan2.py:238,262-266,338-343; op/fused_act.py:74-83):
 *   y[b,c] = lrelu_0.2( upfirdn2d(x[b,c], k, pad=(pad0,pad1)) * gain[b,c] + noise_w * noise[b,0] + bias[c] ) * sqrt(2)
 * x [B,C,in_h,in_w] -> y [B,C,out_h,out_w], up = down = 1. gain (may be NULL = 1) carries the demodulation
 * factor of the preceding shared-weight convolution. noise [B or 1,1,out_h,out_w] (noise_batch_stride 0 = broadcast).
 * Limits (the kernel addresses a plane through 32-bit buffer offsets): kh == kw in 2..4, and planes below 2 GiB — MAUA_ENOSYS otherwise
 * (there is no generic fallback behind this entry: run maua_upfirdn2d_f32, which has one, and maua_fused_bias_act_f32 instead).  The noise
 * map is read as exactly out_h * out_w floats per sample from noise + b * noise_batch_stride: the CALLER guarantees that shape (the Python
 * mirror checks it against the feature map, models/stylegan2.py StyledConv.run; the reference raises a broadcast error there). */
int maua_blur_noise_act_f32(const float* x, const float* k, float* y, int batch, int channels, int in_h, int in_w,
                            int kh, int kw, int pad0, int pad1, const float* gain, const float* noise,
                            int64_t noise_batch_stride, const float* noise_w, const float* bias,
                            const maua_frame_source_t* src, int noise_slot, const float* post_s, int post_stride, void* stream);

/* The WHOLE up-sampling StyledConv in one pass (round 5): transposed 3x3 modulated convolution (stride 2) -> Blur (4x4 SEPARABLE taps, pad
 * (1, 1)) -> NoiseInjection -> FusedLeakyReLU — models/stylegan2.py:229-238,262-266,338-343 — without the raw (2H+1) x (2W+1) map that
 * maua_modconv3x3_f32 (mode 6) writes and maua_blur_noise_act_f32 reads back (548 MB per 1024^2 frame each way):
 *   y[b,o] = lrelu_0.2( blur( conv_transpose2d(x[b] * s[b], W)[o] ) * wscale * d[b,o] + noise_w * noise[b,0] + bias[o] ) * sqrt(2),  y [B,Cout,2H,2W].
 * The blur runs on the accumulators of csrc/modconv_up2d.hip (F(2,2) on both axes of the polyphase form): horizontal pass across the lanes
 * of a wave, vertical pass across the waves of a workgroup through LDS; a workgroup walks a vertical segment of tiles and keeps the three
 * halo rows in LDS; the rows either side of a segment boundary go through `ws` to a small second launch.  `wq` = maua_pack_weight_up2d_f32.
 * `k4`: the [4][4] tap matrix in DEVICE memory; it MUST be separable (outer product, as make_kernel([1,3,3,1]) is) — the caller checks.
 * `ws`: maua_upconv_blur_ws_floats(...) floats.  Shapes: maua_upconv_blur_ok (cin % 8 == 0, cin <= 256, cout % 32 == 0, h % 8 == 0,
 * w % 32 == 0); MAUA_ENOSYS otherwise (the two-launch path serves every shape). */
int maua_upconv_blur_ok(int cin, int cout, int h, int w);
int64_t maua_upconv_blur_ws_floats(int batch, int cin, int cout, int h, int w);
int maua_upconv_blur_f32(const float* x, const float* wq, const float* s, int s_stride, const float* d, float* y, float* ws,
                         const float* k4, const float* noise, int64_t noise_batch_stride, const float* noise_w, const float* bias,
                         const maua_frame_source_t* src, int noise_slot, int batch, int cin, int cout, int h, int w, float wscale,
                         const float* post_s /* [B, s_stride] or NULL */, void* stream);

/* All style affines and demodulation factors of one forward, two launches in total, table-driven.
 *  affine (EqualLinear, models/stylegan2.py:140-146,207,220), with the truncation lerp of Generator.forward
 *  (:541-543) applied to the latent first when trunc != NULL (latent' = trunc_latent + trunc[b]*(latent - trunc_latent)):
 *      s[b*s_stride + s_off_l + i] = (1/sqrt(style_dim)) * sum_j mod_w_l[i,j] * latent'[b, lat_idx_l, j] + mod_b_l[i]
 *  demod (:223-225) in the shared-weight formulation, for layers with wsq != NULL:
 *      d[d_off_l + b*cout_l + o] = rsqrt( wscale_l^2 * sum_i wsq_l[o,i] * s[b, s_off_l + i]^2 + 1e-8 )
 *  `table` lives in DEVICE memory (n_layers entries). batch <= 64 per call. */
typedef struct {
    const float* mod_w; /* [cin, style_dim] */
    const float* mod_b; /* [cin] */
    const float* wsq;   /* [cout, cin] sum over taps of W^2 (maua_pack_weight_f32), NULL = no demodulation */
    int cin;
    int cout;
    int lat_idx;        /* which of the n_latent rows feeds this layer */
    int s_off;          /* offset of this layer's slice inside one batch row of s */
    int64_t d_off;      /* offset of this layer's [batch, cout] block inside d */
    float wscale;       /* 1/sqrt(cin * k * k) */
    int pad_;
} maua_style_layer_t;
int maua_style_affine_f32(const float* latents, int batch, int n_latent, int style_dim, const float* trunc,
                          const float* trunc_latent, const maua_style_layer_t* table, int n_layers, int max_cin,
                          float* s, int s_stride, const maua_frame_source_t* src, void* stream);
int maua_demod_f32(const maua_style_layer_t* table, int n_layers, int max_cout, const float* s, int s_stride, float* d,
                   int batch, void* stream);

/* Sum of squared taps wsq[o,i] and the tap-major repack wp[tap][i][o] of a [cout,cin,k,k] weight (one-off, at load). */
int maua_pack_weight_f32(const float* w, float* wp, float* wsq, int cout, int cin, int ktaps, void* stream);

/* Winograd F(2,3) form of the same weight, transformed along kx: wq[(ky*4+xi)][i][o_pad] (o_pad = cout padded to 32;
 * above 32 channels padded to 64 with the columns of every 64-group interleaved [o % 32][o / 32 % 2], both here and in the
 * F(4,3) pack) with
 * xi 0: g0, 1: (g0+g1+g2)/2, 2: (g0-g1+g2)/2, 3: g2.  The operand of maua_modconv3x3_f32 in mode 2. */
int maua_pack_weight_wino_f32(const float* w, float* wq, int cout, int cin, void* stream);
/* Winograd F(4,3) form (interpolation points 0, +-1, +-2, inf): wq[(ky*6+xi)][i][o_pad] with xi 0: g0/4,
 * 1: -(g0+g1+g2)/6, 2: -(g0-g1+g2)/6, 3: (g0+2g1+4g2)/24, 4: (g0-2g1+4g2)/24, 5: g2.  Operand of mode 3. */
int maua_pack_weight_wino43_f32(const float* w, float* wq, int cout, int cin, void* stream);
/* Transposed-conv form for mode 4 (F(2,2) on the even x-phase of the polyphase decomposition):
 * wq[(ky*4+j)][i][o_pad] with j 0: g2, 1: g0+g2, 2: g0, 3: g1. */
int maua_pack_weight_upwino_f32(const float* w, float* wq, int cout, int cin, void* stream);

/* Transposed-conv form for mode 6 (F(2,2) on BOTH axes of the polyphase decomposition: 25 instead of 36 products per 2x2 block of
 * positions): 16 transformed-kernel entries per (cout, cin) pair — 9 (even, even) + 3 (even row, odd column) + 3 (odd row, even
 * column) + the centre tap, one axis transforming as (tap 2, tap 0 + tap 2, tap 0) — stored as the LDS tile images
 * wq[cout/32][cin/4][16][cin%4][32 columns], followed by the five [cin][cout] tap matrices of the kernel's last row / column that
 * the edge lines (output row 2H, column 2W) use.  maua_pack_weight_up2d_floats() floats.  maua_modconv_up2d_ok(): cin % 4 == 0,
 * cout % 32 == 0, h % 8 == 0, w % 32 == 0. */
int64_t maua_pack_weight_up2d_floats(int cout, int cin);
int maua_pack_weight_up2d_f32(const float* w, float* wq, int cout, int cin, void* stream);
int maua_modconv_up2d_ok(int cin, int cout, int h, int w);
/* SIDE MEASUREMENT (off by default, never the headline): packed weight of the split-bf16 plain convolution (up == 7 below) — every
 * fp32 weight as two bf16 terms w_h + w_l in MFMA lane order, maua_pack_weight_sbf16_bytes() bytes.  maua_modconv_sbf16_ok():
 * cin % 16 == 0, cout % 128 == 0, h % 8 == 0, w % 32 == 0. */
int64_t maua_pack_weight_sbf16_bytes(int cout, int cin);
int maua_pack_weight_sbf16_f32(const float* w, void* wq, int cout, int cin, void* stream);
int maua_modconv_sbf16_ok(int cin, int cout, int h, int w);
int maua_modconv_sbf16_up_ok(int cin, int cout, int h, int w);  /* the transposed form (up == 8): cout % 32 == 0 instead of % 128 */

/* 2-D Winograd F(2x4, 3x3) form for mode 5 (F(2,3) along ky on top of F(4,3) along kx: 24 values per (cout, cin) pair),
 * stored as the LDS tile image the kernel DMAs linearly: wq[cout/BM][cin/4][fy 4][xf 6][cin%4][BM columns] (BM = 64, or 32 for
 * a 32-channel layer; m-tile pairs interleaved inside a row).  Needs 24*cin*cout floats.  maua_modconv_w2d_ok() says whether a
 * layer shape qualifies (cin % 4 == 0, cout == 32 or cout % 64 == 0, w % 32 == 0, h % 8 (16 for cout 32) == 0). */
int maua_pack_weight_wino2d_f32(const float* w, float* wq, int cout, int cin, void* stream);
int maua_modconv_w2d_ok(int cin, int cout, int h, int w);
/* number of output-channel tiles (workgroups per pixel tile) the 2-D Winograd kernel uses for this layer; 0 = shape not accepted */
int maua_modconv_w2d_mtiles(int cin, int cout, int h, int w);

/* ModulatedConv2d 3x3 (models/stylegan2.py:217-254) as input-scale -> shared-weight implicit GEMM on MFMA ->
 * output-demod, with the StyledConv tail (noise + bias + leaky ReLU, :338-343) fused when `fuse_act`:
 *   plain   : x[B,cin,H,W] -> y[B,cout,H,W]        (pad 1)
 *   up == 1 : x[B,cin,H,W] -> y[B,cout,2H+1,2W+1]  (conv_transpose2d stride 2, :229-237) — raw, un-demodulated
 *             when fuse_act == 0 (the blur kernel applies gain/noise/bias/act).
 *   up == 2 : the plain convolution evaluated through Winograd F(2,3) along x (W even): 1.5x fewer MFMA cycles,
 *             same result to fp32 rounding; wp is then the maua_pack_weight_wino_f32 weight.
 *   up == 4 : the transposed convolution of up == 1 with F(2,2) on its even x-phase (W even, fuse_act == 0): 5 instead
 *             of 6 MFMA K-steps per position pair and kernel row; wp from maua_pack_weight_upwino_f32.
 *   up == 3 : the same through Winograd F(4,3) (W % 4 == 0): 2x fewer MFMA cycles than direct, |error| ~2e-5 of
 *             the output scale; wp from maua_pack_weight_wino43_f32.
 *   up == 6 : the transposed convolution of up == 1 with F(2,2) on both axes (shapes accepted by maua_modconv_up2d_ok, fuse_act == 0):
 *             25/36 of the MFMA cycles of up == 1; wp from maua_pack_weight_up2d_f32.
 *   up == 5 : the plain convolution through 2-D Winograd F(2x4, 3x3) (shapes accepted by maua_modconv_w2d_ok): 3x fewer
 *             MFMA cycles than direct, 1.5x fewer than up == 3; wp from maua_pack_weight_wino2d_f32.
 *   up == 7 : SIDE MEASUREMENT — the plain convolution in its direct 9-tap form with split-bf16 products on the bf16 matrix cores
 *             (a b ~= a_h b_h + a_h b_l + a_l b_h, fp32 accumulation; relative error of a product <= 2^-16 + 2^-17); wp from
 *             maua_pack_weight_sbf16_f32.  Not used unless the caller asks for it; the default path computes in fp32.
 *   up == 8 : SIDE MEASUREMENT — the transposed convolution of up == 1 with split-bf16 products (four polyphase phase launches of the
 *             up == 7 kernel + fp32 edge lines), same packed weight as up == 7, fuse_act == 0, ws as for up == 6.
 * wp = tap-major packed weight from maua_pack_weight_f32; s = per-sample input scales [B, s_stride] (NULL with up == 5 / 6: x arrives
 * pre-scaled, see THE STYLE FOLD above);
 * d = demod [B,cout] (NULL = 1).  `ws` is a caller-owned fp32 workspace of at least maua_modconv_ws_floats()
 * floats used for split-K partial sums on small feature maps and, for up == 6, for the exported last input column [B, cin, H]
 * (may be NULL when that returns 0). */
int64_t maua_modconv_ws_floats(int batch, int cin, int cout, int h, int w, int up);
/* Name of the kernel template instance launched by the last modconv call of this process, as rocprofv3 prints it
 * ("modconv_mfma_kernel<BM, BN, WM, MODE, MULTI, FAST, MAXP>") — the key of the per-instance PMC tables in profiles/. */
int maua_modconv_last_instance(char* buf, int buf_len);
int maua_modconv3x3_f32(const float* x, const float* wp, const float* s, int s_stride, const float* d,
                        float* y, int batch, int cin, int cout, int h, int w, int up, float wscale, int fuse_act,
                        const float* noise, int64_t noise_batch_stride, const float* noise_w, const float* bias,
                        float* ws, const maua_frame_source_t* src, int noise_slot, void* stream);

/* Plain StyledConv with the following ToRGB (models/stylegan2.py:338-343 then :356-365) fused into its epilogue: the
 * activated feature map is reduced to RGB while it is still in registers (all channels of a pixel sit in one wave).
 * Only for layers whose channels fit one weight tile in a single wave row (cout <= 64); returns MAUA_ENOSYS otherwise
 * (the caller then runs maua_modconv3x3_f32 + maua_torgb_f32).  rgb_s = the ToRGB layer's styles [B, s_stride] (same
 * stride as s).  frames_u8 != NULL (last layer): the image is not written as fp32 planes at all but leaves as uint8 NHWC frames
 * [B,H,W,3] = clamp(-1,1), (x+1)*127.5, truncating cast (the frame epilogue of render.py:40-43 folded in); rgb_out may then be NULL
 * (the normal case) — when it is not, the fp32 planes are written as well (the parity tests compare them with the oracle in float).
 * mode = 0 (direct), 2, 3 or 5 (Winograd F(2,3) / F(4,3) / 2-D F(2x4,3x3), wp from the matching pack function).  store_features = 0 skips writing y (legal for the last layer: nothing downstream reads it). */
int maua_styledconv_torgb_f32(const float* x, const float* wp, const float* s, int s_stride, const float* d,
                              float* y, int batch, int cin, int cout, int h, int w, int mode, float wscale,
                              const float* noise, int64_t noise_batch_stride, const float* noise_w, const float* bias,
                              const float* rgb_w, const float* rgb_s, float rgb_wscale, const float* rgb_bias,
                              const float* rgb_skip, const float* rgb_k4, float* rgb_out, int store_features,
                              uint8_t* frames_u8, const maua_frame_source_t* src, int noise_slot,
                              const float* post_s /* [B, s_stride] or NULL; mode 5 only */, void* stream);

/* LOW-RESOLUTION LAYERS (ABI 5; the 4^2 .. 32^2 outputs of a generator).  Their convolutions are the split-K direct / polyphase kernels of
 * maua_modconv3x3_f32 (up = 0 / 1), and on maps of a few KB every launch is a fixed 5 .. 20 us whatever it computes.  These entries run the
 * convolution with its split-K slabs left in `ws` (maua_lowres_ws_floats floats: one slab when K is not split) and reduce them in the SAME
 * launch that applies what follows:
 *   maua_upconv_blur_lowres_f32   = maua_modconv3x3_f32(up = 1) + maua_blur_noise_act_f32 (4 x 4 taps, pad (1, 1)) of an up-sampling
 *       StyledConv (models/stylegan2.py:229-238, :338-343): y [B, Cout, 2H, 2W], bit-identical to that pair wherever K is split (without a
 *       split the pair applies wscale * d as one factor, this entry as two);  post_s as THE STYLE FOLD;
 *   maua_styledconv_rgbpart_lowres_f32 = maua_modconv3x3_f32(up = 0, fuse_act = 1) of a plain StyledConv — y bit-identical — that also
 *       leaves per-32-channel-group partial ToRGB sums (models/stylegan2.py:356-365) in rgb_partial [B, 3 * Cout / 32, H, W] (plane 3 g + c),
 *       which maua_torgb_f32's plane-sum form (w = s = NULL) turns into the image.
 * `up` of the up-sampling entry picks the convolution: 1 = the polyphase kernel (wp = maua_pack_weight_f32), 6 = F(2,2) on both axes of the
 * polyphase form (csrc/modconv_up2d.hip on 16 x 16-position tiles, K split over workgroups; wp = maua_pack_weight_up2d_f32): 16-wide
 * inputs, 25 instead of 36 products per 2 x 2 positions and that kernel's operand pipeline (the 16^2 -> 32^2 layer at batch 8: 129 -> 77 us).
 * maua_lowres_ok: up == 1: 2H * 2W <= 1024;  up == 6: W == 16, H % 16 == 0, Cin % 8 == 0, Cout % 32 == 0, 2H * 2W <= 1024;
 * up == 0 / 2 / 3 (the plain entry's `mode`: direct, Winograd F(2,3) / F(4,3) along x — 2 needs an even W, 3 W % 4 == 0): Cout % 32 == 0,
 * H * W % 16 == 0, H * W <= 1024.  MAUA_ENOSYS otherwise.  `wp` of the plain entry = the pack of its mode (maua_pack_weight_f32 /
 * maua_pack_weight_wino_f32 / maua_pack_weight_wino43_f32); s must not be NULL (no pre-scaled instances here). */
int maua_lowres_ok(int cin, int cout, int h, int w, int up);
int64_t maua_lowres_ws_floats(int batch, int cin, int cout, int h, int w, int up);
int maua_upconv_blur_lowres_f32(const float* x, const float* wp, const float* s, int s_stride, const float* d, float* y, float* ws,
                                const float* k4, const float* noise, int64_t noise_batch_stride, const float* noise_w, const float* bias,
                                const maua_frame_source_t* src, int noise_slot, int batch, int cin, int cout, int h, int w, int up,
                                float wscale, const float* post_s, void* stream);
int maua_styledconv_rgbpart_lowres_f32(const float* x, const float* wp, const float* s, int s_stride, const float* d, float* y, float* ws,
                                       const float* noise, int64_t noise_batch_stride, const float* noise_w, const float* bias,
                                       const float* rgb_w, const float* rgb_s, float rgb_wscale, float* rgb_partial,
                                       const maua_frame_source_t* src, int noise_slot, int batch, int cin, int cout, int h, int w,
                                       int mode, float wscale, void* stream);

/* conv1 on a ConstantInput (models/stylegan2.py:269-278, :547-549; csrc/constconv.hip).  The input of a generator's first StyledConv is
 * the learned 4 x 4 constant, the same for every frame and scaled per (frame, channel) by the styles, so
 *   y[b,o,p] = act( (wscale * sum_i T[o,p,i] s[b,i]) * d[b,o] + noise_w * noise[b,p] + bias[o] ),  T[o,p,i] = sum_taps W[o,i,ky,kx] c[i, p + (ky,kx) - 1]
 * with T a function of the checkpoint alone: maua_pack_const_conv_f32 (w [Cout,Cin,3,3], c [Cin,4,4] -> T, Cout * 16 * Cin floats, once per
 * weight version), then maua_const_styledconv_f32 per batch — 1/9 of the convolution's multiply-adds, one read of T, no split-K.
 * rgb_partial != NULL: per-32-channel-group partial ToRGB sums [B, 3 * Cout / 32, 4, 4] as maua_styledconv_rgbpart_lowres_f32 leaves them.
 * maua_const_conv_ok: h == w == 4, Cin % 8 == 0, Cout % 32 == 0 (Cin <= 1536: the styles of eight frames in LDS); MAUA_ENOSYS otherwise. */
int maua_const_conv_ok(int cin, int cout, int h, int w);
int maua_pack_const_conv_f32(const float* w, const float* c, float* T, int cout, int cin, int h, int wd, void* stream);
int maua_const_styledconv_f32(const float* T, const float* s, int s_stride, const float* d, float* y, const float* noise,
                              int64_t noise_batch_stride, const float* noise_w, const float* bias, const float* rgb_w, const float* rgb_s,
                              float rgb_wscale, float* rgb_partial, const maua_frame_source_t* src, int noise_slot, int batch, int cin,
                              int cout, int h, int w, float wscale, void* stream);

/* The same fusion for layers wider than one weight tile (128..512 output channels, mode 5 only): every output-channel tile leaves
 * its share of the ToRGB sum  sum_{i in tile} (rgb_wscale * rgb_w[c,i] * rgb_s[b,i]) * y[b,i,Y,X]  in
 * rgb_partial [B, 3 * maua_modconv_w2d_mtiles(), H, W] (plane 3 m + c), the feature map y is stored as usual.  The caller finishes with
 * maua_torgb_f32 over the 3 m_tiles planes (w = s = NULL: its plane-sum form), which adds bias and the up-sampled skip: the ToRGB of
 * models/stylegan2.py:356-365 then reads 3 m_tiles planes instead of all `cout` feature planes.  MAUA_ENOSYS for mode != 5. */
int maua_styledconv_torgb_partial_f32(const float* x, const float* wp, const float* s, int s_stride, const float* d, float* y,
                                      int batch, int cin, int cout, int h, int w, int mode, float wscale, const float* noise,
                                      int64_t noise_batch_stride, const float* noise_w, const float* bias, const float* rgb_w,
                                      const float* rgb_s, float rgb_wscale, float* rgb_partial,
                                      const maua_frame_source_t* src, int noise_slot, const float* post_s /* [B, s_stride] or NULL */,
                                      void* stream);

/* StyleGAN1 (`--stylegan1`, models/stylegan1.py:258-318 LayerEpilogue) — conv bias, per-channel-weighted noise, LeakyReLU(0.2),
 * instance norm (biased variance, eps 1e-5) and the style modulation in one launch:
 *   y[b,c] = norm( lrelu_0.2( x[b,c] + bias[c] + noise_w[c] * noise[b or 0] ) ) * (style[b, c] + 1) + style[b, C + c]
 * bias / noise / style may be NULL (skipped); instance_norm = 0 skips the normalisation; y == x is allowed. */
int maua_sg1_epilogue_f32(const float* x, const float* bias, const float* noise, int64_t noise_batch_stride,
                          const float* noise_w, const float* style, int style_stride, float* y, int batch, int channels,
                          int h, int w, int instance_norm, void* stream);

/* ToRGB (models/stylegan2.py:356-365): 1x1 modulated conv without demod + bias + 2x FIR-upsampled skip
 * (Upsample :34-52, kernel k4 = 4x4 taps in device memory, pad (2,1)).  skip == NULL: no skip.
 *   y[b,c,Y,X] = sum_i (wscale * w[c,i] * s[b,i]) * x[b,i,Y,X] + bias[c] + up2(skip)[b,c,Y,X]
 * w == NULL and s == NULL: x holds per-tile partial ToRGB sums [B, cin = 3 M, H, W] (maua_styledconv_torgb_partial_f32), plane 3 m + c
 * feeding colour c:  y[b,c] = sum_m x[b, 3 m + c] + bias[c] + up2(skip)[b,c]  (wdt % 4 == 0; one round trip of independent loads). */
int maua_torgb_f32(const float* x, const float* w, const float* s, int s_stride, const float* bias,
                   const float* skip, const float* k4, float* y, int batch, int cin, int h, int wdt,
                   float wscale, void* stream);

/* Frame epilogue of render.py:40-43: [B,3,H,W] fp32 -> [B,H,W,3] uint8 via clamp(-1,1), (x+1)*127.5, truncation. */
int maua_frames_to_u8(const float* img, uint8_t* out, int batch, int h, int w, void* stream);
/* Wide-output delivery of render.py:97-104 on the device: crop [y0, y0+crop_h) x [x0, x0+crop_w) of uint8 NHWC frames
 * in[B,in_h,in_w,3] and resize to out[B,out_h,out_w,3] as PIL's Image.resize(BILINEAR) does for an up-scale (2-tap triangle filter at
 * (dst + 0.5) * in/out - 0.5, clamped taps, horizontal pass rounded to 8 bits, then vertical).  MAUA_ENOSYS for a down-scale. */
int maua_crop_resize_u8(const uint8_t* in, uint8_t* out, int batch, int in_h, int in_w, int x0, int y0, int crop_w, int crop_h,
                        int out_w, int out_h, void* stream);

/* ------------------------------------------------------------------------------------------------ audio / temporal
 * Circular Gaussian FIR along time (audioreactive/signal.py:319-368): x [T, F] -> y [T, F], taps[2*radius+1]
 * already normalised / causal-weighted by the host; circular index when radius <= T, else the reference's
 * single-wrap-then-zero padding (:350-355).  The taps are staged in LDS with 96 floats of zero padding: radius <=
 * MAUA_TEMPORAL_FIR_MAX_RADIUS (a Gaussian of sigma ~2000 frames), MAUA_EINVAL above.  A non-finite sample stays inside the filter's
 * support, as with the reference's conv1d. */
#define MAUA_TEMPORAL_FIR_MAX_RADIUS 8143
int maua_temporal_fir_f32(const float* x, const float* taps, float* y, int n_frames, int64_t features, int radius,
                          void* stream);

/* Power spectrogram |STFT|^2 (librosa.stft as called at audioreactive/signal.py:51,93,119): centred, reflect-padded
 * frames, window[n_fft] in device memory, n_fft a power of two <= 4096.  y[n_samples] -> p[n_bins = n_fft/2+1, n_frames]. */
int maua_stft_power_f32(const float* y, int64_t n_samples, const float* window, int n_fft, int hop, float* p,
                        int n_frames, void* stream);
/* Harmonic / percussive source separation pieces (librosa.effects.percussive / harmonic, audioreactive/signal.py:49,150):
 *   complex STFT -> re/im [n_fft/2+1, n_frames];  inverse STFT (frames_ws: n_frames*n_fft floats of workspace; window-sum-
 *   square normalised overlap-add, centre padding removed, output length n_samples);
 *   median filter of x[rows, cols] along axis 0 (frequency) or 1 (time), size in {3,5,9,17,31}, scipy 'reflect' boundary;
 *   soft mask (x / z)^p / ((x/z)^p + (x_ref*margin/z)^p) applied to the complex spectrum. */
int maua_stft_complex_f32(const float* y, int64_t n_samples, const float* window, int n_fft, int hop, float* out_re,
                          float* out_im, int n_frames, void* stream);
int maua_istft_f32(const float* in_re, const float* in_im, const float* window, int n_fft, int hop, int n_frames,
                   float* frames_ws, float* y, int64_t n_samples, void* stream);
int maua_median_filter_f32(const float* x, float* y, int rows, int cols, int size, int axis, void* stream);
int maua_softmask_apply_f32(const float* re, const float* im, const float* x, const float* x_ref, float margin, float power,
                            int split_zeros, float* out_re, float* out_im, int64_t n, void* stream);
/* out[M,N] = fb[M,K] @ p[K,N] (mel / chroma filterbank projection), optional 10*log10(max(amin, .)) when to_db. */
int maua_filterbank_f32(const float* fb, const float* p, float* out, int m, int k, int n, int to_db, float amin,
                        void* stream);

/* Fourier-method resampling along time, y[num, features] from x[n, features] (row-major, fp64) — scipy.signal.resample
 * as called at audioreactive/signal.py:68,152 — evaluated as a Dirichlet-kernel sum in the time domain. */
int maua_resample_f64(const double* x, int n, int64_t features, double* y, int num, void* stream);

/* |constant-Q transform| (librosa.cqt role inside chroma_cqt / chroma_cens, signal.py:115-117): out[k][t] for n_bins
 * geometrically spaced frequencies freqs[k] (Hz) with window lengths lengths[k] (samples), centred frames every `hop`
 * samples with reflect padding, periodic-Hann kernels of unit L1 norm, scaled by 1/sqrt(length). */
int maua_cqt_mag_f32(const float* y, int64_t n_samples, const float* freqs, const int* lengths, int n_bins, int hop,
                     float sr, float* out, int n_frames, void* stream);

/* Chroma post-processing for audioreactive/signal.py:102-133 (ch / out are [n_bins <= 32, n_frames], fp32):
 * CENS = per-frame L1 normalisation, 4-level quantisation, Hann smoothing over win_len (odd) frames, L2 normalisation;
 * nn_median = per-frame median over the k frames of highest cosine similarity outside |i-j| < width (the
 * aggregate=np.median, metric="cosine" nearest-neighbour filter at :131).  The fp64 similarity row of a frame lives in LDS
 * while n_frames * 8 + k * (4 + 4 n_bins) bytes fit (~16k frames); longer tracks need a caller-owned workspace `ws` of
 * maua_nn_median_ws_doubles() doubles (0 = not needed, ws may be NULL).  A non-NULL ws of min(n_frames, 1024) * n_frames doubles
 * selects the workspace path at any size (same results, bit for bit). */
int maua_chroma_cens_f32(const float* ch, float* out, int n_bins, int n_frames, int win_len, void* stream);
int64_t maua_nn_median_ws_doubles(int n_bins, int n_frames, int k);
int maua_nn_median_f32(const float* ch, float* out, int n_bins, int n_frames, int k, int width, double* ws, void* stream);

/* 3-D tileable Perlin noise (audioreactive/latent.py:188-246): grad [r0+1,r1+1,r2+1,3] -> out [n0,n1,n2]. */
int maua_perlin3d_f32(const float* grad, float* out, int n0, int n1, int n2, int r0, int r1, int r2, void* stream);

/* Network-bending warp (audioreactive/bend.py:52-102): per-frame inverse affine map m[b] = {a00,a01,a02,a10,a11,a12}
 * (output pixel -> source pixel, in pixels), bilinear sampling of the reflect-padded source, zeros outside the padded
 * canvas (kornia warp_affine default padding_mode='zeros', align_corners as encoded by the host in m). */
int maua_affine_reflect_warp_f32(const float* x, const float* m, float* y, int batch, int channels, int h, int w,
                                 int pad_l, int pad_r, int pad_t, int pad_b, const float* add_noise, void* stream);
/* Same, for a canvas built by a CHAIN of ReflectionPad2d (audioreactive/bend.py:60-64 stacks three): pad_* are the total
 * paddings, xmap[w+pad_l+pad_r] / ymap[h+pad_t+pad_b] (int32, device, either may be NULL = single fold) give the source
 * column / row of every canvas column / row.  src != NULL: m is the map SEQUENCE of the whole render [n_frames, 6] and sample
 * b uses row src->frame0 + b (render.py:151-158 rebuilds the transform per batch; a captured graph reads it per replay). */
int maua_affine_reflect_warp_mapped_f32(const float* x, const float* m, float* y, int batch, int channels, int h, int w,
                                        int pad_l, int pad_r, int pad_t, int pad_b, const float* add_noise,
                                        const int* xmap, const int* ymap, const maua_frame_source_t* src, void* stream);

/* ------------------------------------------------------------------------------------------------ hipGraph runtime
 * Capture everything launched on `stream` between begin/end into a hipGraph and replay it (per-frame generator
 * forward, north_star "hipGraph-captured").  Handles are opaque. */
int maua_graph_begin_capture(void* stream);
int maua_graph_end_capture(void* stream, void** graph_exec_out);
int maua_graph_launch(void* graph_exec, void* stream);
int maua_graph_destroy(void* graph_exec);

/* HIP-event timing on an arbitrary stream (bench.py roofline leg; torch.cuda.Event only sees torch's stream). */
int maua_event_create(void** ev);
int maua_event_record(void* ev, void* stream);
int maua_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);
int maua_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* MAUA_HIP_H */
