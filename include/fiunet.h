/*
 * fiunet.h -- C ABI of the MI355X-native UNet frame-pair forward (libfiunet_hip.so).
 *
 * The reference (daultanigaurav/AI-BASED-FRAME-INTERPOLATION) has no FFI / plugin layer: its
 * boundary for this path is a Python class plus a state-dict.  Each entry
 * point below therefore replaces a piece of that Python surface, cited as reference file:line.
 * Plain pointers and sizes only; no torch types; every function returns an int status
 * (0 == FIUNET_OK) and never throws across the ABI.  Device pointers are HIP device memory
 * of the context's device; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * All launches are asynchronous on that stream; one in-flight call per ctx+stream.
 *
 * See INTEGRATION.md for the reference-side binding (a ctypes stub in model/unet.py).
 */
#ifndef FIUNET_H
#define FIUNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FIUNET_ABI_VERSION 5   /* 4: fiunet_prepare_precision; fiunet_debug_read_activation takes the capacity of dst; 5: fiunet_forward_u8_strided */

enum fiunet_status {
    FIUNET_OK = 0,
    FIUNET_ERR_INVALID_ARG = 1,   /* bad pointer / size (the reference raises RuntimeError) */
    FIUNET_ERR_BAD_SHAPE = 2,     /* H or W < 16 (reference: max-pool raises) */
    FIUNET_ERR_MISSING_WEIGHT = 3,/* a state-dict tensor is absent or has the wrong numel */
    FIUNET_ERR_NOT_LOADED = 4,    /* forward before load_weights */
    FIUNET_ERR_WORKSPACE = 5,     /* workspace too small */
    FIUNET_ERR_HIP = 6,           /* a HIP runtime call failed; see fiunet_last_error_string */
    FIUNET_ERR_UNSUPPORTED = 7    /* a value outside what the kernels cover (fiunet_ssim_gauss_f32: even or > 31 window;
                                     read-back of an upsampled half that is never stored) */
};

/* Arithmetic type of the conv path.  FP32: fp32 storage, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).
 * BF16: bf16 activations/weights in HBM, v_mfma_f32_16x16x32_bf16 with fp32 accumulation; the
 * first (Cin=2) conv and the final 1x1 conv stay fp32 arithmetic in both modes. */
enum fiunet_precision {
    FIUNET_FP32 = 0,   /* exact fp32: v_mfma_f32_16x16x4_f32 (the reference's own arithmetic) */
    FIUNET_BF16 = 1,   /* bf16 storage and MFMA operands, fp32 accumulation */
    FIUNET_BF16X2 = 2  /* the fp32 CONTRACT (|d| <= 1e-3) on the bf16 pipe - every activation and weight is two bf16
                          pieces (hi + lo, 16 significant bits; activations [hi planes | lo planes], 4 B per element), a
                          product is wh*xh + wl*xh + wh*xl with fp32 accumulation (~1e-5 relative end to end); fp32 frames
                          in, fp32 logits out, exact-fp32 stem and head.  Needs fiunet_prepare_precision(ctx, FIUNET_BF16X2)
                          once after fiunet_load_weights.  Both decoders; FIUNET_OPT_KEEP_ALL + read-back work, the other
                          A/B options (UNFUSED, GATHER_UPSAMPLE, PAIR_TILES) are ignored in this mode */
};

/* Bit flags for fiunet_set_options. */
enum fiunet_option {
    FIUNET_OPT_UNFUSED = 1, /* ablation: run max-pool, upsample+pad+concat and the 1x1 head as
                               separate kernels instead of fusing them into the consumer conv */
    FIUNET_OPT_KEEP_ALL = 2,/* also store the last 64-ch activation (tap 17) that the fused 1x1
                               head otherwise keeps in registers; for fiunet_debug_read_activation */
    FIUNET_OPT_PAIR_TILES = 8, /* A/B runs: direct >=128-channel convs on the 8-wave tile-pair kernel
                               (conv3x3_pair.hip.h: shared weight ring, double-buffered in-tile); same bits */
    FIUNET_OPT_RNE_WEIGHTS = 32, /* A/B runs: round the bf16 conv weights to nearest instead of with the per-filter
                               error feedback (fiunet.hip, f32_to_bf16_feedback).  Read by fiunet_load_weights: set it
                               BEFORE loading */
    FIUNET_OPT_NO_DITHER = 64, /* A/B runs: switch off the ordered input dither of the bf16 stem (+-2^-9 on the
                               frames, +d on frame 1 / -d on frame 2, an 8x8 Bayer pattern of the global pixel
                               position: it decorrelates the bf16 rounding of values the network carries through
                               its full-resolution skip from the image content; conv3x3_mfma.hip.h, stem_dither) */
    FIUNET_OPT_GATHER_UPSAMPLE = 16 /* A/B runs and tests: always interpolate the upsampled half of a concat
                               input inside the conv's gather; by default the bf16 convs with >= 2 cout tiles
                               on >= 64k pixels read it from a tensor written once by upsample_kernel; same bits */
};

typedef struct fiunet_ctx fiunet_ctx;

/* Replaces FrameInterpolationUNet.__init__ / UNet.__init__ (model/unet.py:99-103, :66-82).
 * frame_channels: 1 = grayscale (the reference's 2->1 network), 3 = RGB (6->3, README variant).
 * bilinear: 1 = bilinear-upsample decoder, what every reference caller constructs (model/inference.py:77);
 * 0 = the constructor's default (model/unet.py:66,99): ConvTranspose2d(C, C/2, 2, 2) decoder (unet.py:42-44), widths
 * 64-128-256-512-1024, 118-tensor state-dict with `unet.up{k}.up.weight/bias`. */
int fiunet_create(fiunet_ctx** out_ctx, int device_id, int frame_channels, int bilinear);

/* Replaces nn.Module teardown (Python GC). */
int fiunet_destroy(fiunet_ctx* ctx);

int fiunet_set_options(fiunet_ctx* ctx, unsigned flags);

/* Replaces model.load_state_dict(...) + .to(device) + .eval() (model/inference.py:83-97).
 * `names[i]` are reference state-dict keys ("unet.inc.double_conv.0.weight", ...),
 * `host_ptrs[i]` contiguous fp32 host arrays, `numels[i]` their element counts.  The
 * `num_batches_tracked` counters may be omitted or passed (ignored).  The library folds the
 * eval-mode BatchNorm into a per-channel scale/shift, repacks OIHW weights into its kernel
 * layout for both precisions and owns the device copies. */
int fiunet_load_weights(fiunet_ctx* ctx, int n, const char* const* names,
                        const float* const* host_ptrs, const int64_t* numels);

/* Builds the weight copies a precision needs beyond what fiunet_load_weights made (FP32, BF16: nothing; BF16X2: the
 * two-piece [wh | wl] copies, ~69 MB, packed on the device from the fp32 copy).  Call it after fiunet_load_weights and
 * before the first forward in that precision (a forward without it returns FIUNET_ERR_NOT_LOADED); idempotent;
 * allocates and synchronises - so not under stream capture.  The reference has no counterpart: its one precision is
 * whatever dtype the module's tensors have (model/inference.py:96). */
int fiunet_prepare_precision(fiunet_ctx* ctx, int precision);

/* Bytes of device scratch fiunet_forward needs for a [B,*,H,W] batch; 0 on bad arguments.
 * Activations whose lifetimes do not overlap share bytes (the reference, under no_grad, frees every
 * non-skip tensor as it goes: model/unet.py:84-95), so the figure depends on the options in force
 * (FIUNET_OPT_KEEP_ALL pins all 18 activations, FIUNET_OPT_UNFUSED adds the concat scratch): query it
 * AFTER fiunet_set_options.  B=8 1080x1920: 6.4 GB bf16 / 12.8 GB fp32. */
size_t fiunet_workspace_bytes(const fiunet_ctx* ctx, int B, int H, int W, int precision);

/* Smallest batch B (1..64; 65 = none) at which NO layer of a forward of H x W frames cuts its K loop over several
 * workgroups (see "Reproducibility" below): batches of at least that many pairs give every pair the same bits whatever
 * the batch.  1 from 1080p up; at 720p 3 in fp32 and 2 in bf16 / bf16x2 (round 6 rule; 5 until round 5); > 8 for the
 * reference's own 256x256.  The value holds for the context's current options, default tile choice included.  Needs loaded weights (it walks the
 * architecture); 0 on bad arguments.  Used by the video loops to pad a ragged last chunk no further than necessary.
 * The reference has no counterpart (aten's conv results do depend on the batch in the last bit). */
int fiunet_min_unsplit_batch(const fiunet_ctx* ctx, int H, int W, int precision);

/* Replaces FrameInterpolationUNet.forward(frame1, frame2) in eval mode
 * (model/unet.py:105-112 -> :84-95), as called by interpolate_frames (model/inference.py:120).
 * frame1, frame2: device fp32 [B, frame_channels, H, W] contiguous (NCHW);
 * out: device fp32 [B, frame_channels, H, W]; raw logits, no activation.
 * workspace: device scratch of at least fiunet_workspace_bytes(...), 256-B aligned.
 * Reproducibility: a call is deterministic (no atomics; fixed summation order).  For frames of >= 1080p the
 * result of a pair does not depend on the batch it is part of or on its position (bit for bit); for smaller
 * frames that holds among batches of at least fiunet_min_unsplit_batch(ctx, H, W, precision) pairs (2 at 720p, bf16).
 * Below that, layers with fewer workgroups than the chip has CUs cut their K loop over several workgroups, and
 * how many depends on B, so the fp32 summation order of a pair - hence its last bit - may differ between batch
 * sizes (1.8e-5 in fp32 on O(1) outputs, bf16 ulp flips; the same holds for a short band of
 * fiunet_forward_strip against the whole frame).  Callers that need a sequence's result to be independent of how
 * it was batched should pad a ragged last batch up to that size, or to the full batch where even a full batch
 * splits (ai_based_frame_interpolation_amd/inference.py does). */
int fiunet_forward(fiunet_ctx* ctx, const float* frame1, const float* frame2, float* out, int B,
                   int H, int W, int precision, void* workspace, size_t workspace_bytes,
                   void* stream);

/* Spatial tiling (2160x3840 pairs cut into horizontal strips, one per
 * GPU).  The reference has no counterpart: it is the same forward (model/unet.py:84-95) evaluated on
 * the band of rows [y_origin, y_origin + H) of an image of H_image rows.  frame1, frame2, out are that
 * band only ([B, C, H, W] contiguous).  The align_corners=True upsampling (unet.py:40) and the F.pad
 * offsets (unet.py:49-53) are evaluated in whole-image coordinates, so every output row whose
 * receptive field (+-109 rows) lies inside the band - or is cut only by a true image border - equals
 * the un-tiled result; rows nearer to a cut edge see zero padding there and must be discarded by the
 * caller (use a halo of >= 112 rows).  y_origin must be a multiple of 16 (four 2x2 max-pools) and H a
 * multiple of 16 unless the band ends the image.  fiunet_forward(..., H, ...) is the band
 * y_origin = 0, H_image = H.  Workspace: fiunet_workspace_bytes(ctx, B, H, W, precision). */
int fiunet_forward_strip(fiunet_ctx* ctx, const float* frame1, const float* frame2, float* out, int B,
                         int H, int W, int y_origin, int H_image, int precision, void* workspace,
                         size_t workspace_bytes, void* stream);

/* Video-path variant: uint8 frames in, uint8 interpolated frame out, with the reference's
 * pre/post-processing on device: x/255*2-1 (model/inference.py:31-35) where the frames are read and
 * trunc(clamp((y+1)/2,0,1)*255) (model/inference.py:54-61) where the output is written, bit for bit the
 * values fiunet_preprocess_u8 -> fiunet_forward -> fiunet_postprocess_u8 give.  frame1/2, out: device uint8
 * [B, frame_channels, H, W].  Where the stem conv is evaluated inside the next conv's gather (bf16, gray)
 * it reads the uint8 frames itself, and every fused 1x1 head writes uint8 itself: one launch chain, no
 * fp32 frame buffers; the other configurations (fp32 / RGB stem kernel, FIUNET_OPT_UNFUSED, FIUNET_OPT_KEEP_ALL)
 * run the two elementwise kernels around the forward on up to three fp32 buffers.  Workspace:
 * fiunet_workspace_bytes_u8 (query it after fiunet_set_options). */
size_t fiunet_workspace_bytes_u8(const fiunet_ctx* ctx, int B, int H, int W, int precision);
int fiunet_forward_u8(fiunet_ctx* ctx, const uint8_t* frame1, const uint8_t* frame2, uint8_t* out,
                      int B, int H, int W, int precision, void* workspace, size_t workspace_bytes,
                      void* stream);
/* The same with the B output images `out_image_stride` BYTES apart (0 = contiguous; >= frame_channels * H * W): the
 * factor-2 video loop (the phantom `interpolate_video` of /root/reference/main.py:118-129) hands
 * over every second frame of its interleaved result F0 M0 F1 M1 ..., so the fused head writes each interpolated frame
 * where it belongs - no temporary, no strided copy afterwards.  Everything else as fiunet_forward_u8. */
int fiunet_forward_u8_strided(fiunet_ctx* ctx, const uint8_t* frame1, const uint8_t* frame2, uint8_t* out,
                              size_t out_image_stride, int B, int H, int W, int precision, void* workspace,
                              size_t workspace_bytes, void* stream);

/* Replaces preprocess_image's arithmetic (model/inference.py:31-35): out = 2*(in/255) - 1. */
int fiunet_preprocess_u8(const uint8_t* in, float* out, size_t n, void* stream);
/* Replaces postprocess_image's arithmetic (model/inference.py:54-61), truncating cast. */
int fiunet_postprocess_u8(const float* in, uint8_t* out, size_t n, void* stream);

/* Quality metrics of the evaluation loop on device.  The reference computes
 * them on the host with scikit-image, one frame at a time (model/evaluation.py:194-218,
 * model/evaluation_simple.py:134-156): peak_signal_noise_ratio(target, pred, data_range=255) and
 * structural_similarity(target, pred, data_range=255) with skimage's defaults (7x7 uniform window,
 * sample covariance, K1 = 0.01, K2 = 0.03, mean over the image minus a 3-pixel border).
 * pred, target: device uint8, `images` planes of H x W (e.g. the [B, C, H, W] output of
 * fiunet_forward_u8 and the ground-truth frames); out: device double[images] (+inf PSNR for identical
 * planes, as skimage).  workspace: device scratch of fiunet_metrics_workspace_bytes(images, H, W),
 * 256-B aligned.  SSIM needs H, W >= 7.  Asynchronous on `stream`. */
size_t fiunet_metrics_workspace_bytes(int images, int H, int W);
int fiunet_psnr_u8(const uint8_t* pred, const uint8_t* target, int images, int H, int W, double* out,
                   void* workspace, size_t workspace_bytes, void* stream);
int fiunet_ssim_u8(const uint8_t* pred, const uint8_t* target, int images, int H, int W, double* out,
                   void* workspace, size_t workspace_bytes, void* stream);

/* The training loss' SSIM on device: SSIMLoss._ssim (model/train.py:37-56) - the
 * window_size x window_size Gaussian window (sigma 1.5, normalised as at train.py:27-35) applied as a
 * depth-wise conv2d with ZERO padding window_size//2 to img1, img2, img1^2, img2^2, img1*img2;
 * C1 = 0.01^2, C2 = 0.03^2; mean of the map.  img1, img2: device fp32, `images` planes of H x W (a
 * [B, C, H, W] tensor is B*C planes: the conv is depth-wise, so planes are independent), in the caller's
 * value range.  out_ssim: device double[images], the mean of each plane's SSIM map (all planes have H*W
 * pixels, so `ssim_map.mean()` (:54) is the mean of these and `.mean(1).mean(1).mean(1)` (:56) the mean
 * over a sample's C planes).  out_sqerr (may be NULL): device double[images], sum (img1 - img2)^2 of each
 * plane - CombinedLoss' MSE term (train.py:75-87) is their total / (images*H*W).  window_size must be odd
 * (the reference's default and only value is 11) and <= 31.  window_1d: HOST pointer to the window_size
 * normalised 1-D Gaussian weights the caller computed (train.py:27-29 with the caller's own torch, whose
 * reduction order decides the last bit of `gauss.sum()`), or NULL to have them computed here (sequential
 * fp32 sum; the mean SSIM then sits within ~1e-6 of the reference's).  Workspace:
 * fiunet_ssim_gauss_workspace_bytes(images, H, W), 256-B aligned.  Asynchronous on `stream`. */
size_t fiunet_ssim_gauss_workspace_bytes(int images, int H, int W);
int fiunet_ssim_gauss_f32(const float* img1, const float* img2, int images, int H, int W, int window_size,
                          const float* window_1d, double* out_ssim, double* out_sqerr, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Parity-test hook: after a fiunet_forward on `workspace` (with FIUNET_OPT_KEEP_ALL set), convert one intermediate
 * activation (blocked channels-last in the compute precision; two-piece in BF16X2) to fp32 NCHW at dst.  tap = 2*block
 * + conv for the 18 fused conv+BN+ReLU stages in state-dict order (0 = unet.inc.double_conv.0 ... 17 =
 * unet.up4.conv.double_conv.3); taps 18..21 = the upsampled + padded half of up1..up4's concat input (`self.up(x1)` +
 * F.pad, unet.py:47-53) where it exists as a tensor (always with the ConvTranspose2d decoder and in BF16X2; for the
 * bilinear decoder only where upsample_kernel materialises it - otherwise FIUNET_ERR_UNSUPPORTED).  out_dims receives
 * {C, H, W} of that activation - C depends on the decoder (the ConvTranspose2d variant is wider at taps 8, 9, 11, 13, 15).  dst_capacity = floats dst can hold; B*C*H*W are written,
 * FIUNET_ERR_INVALID_ARG if it is short.  dst == NULL: dims-only query (nothing is read, workspace may be NULL). */
int fiunet_debug_read_activation(fiunet_ctx* ctx, const void* workspace, int B, int H, int W,
                                 int precision, int tap, float* dst, size_t dst_capacity, int out_dims[3],
                                 void* stream);

/* Measurement hook (bench.py's roofline leg): when enabled, every fiunet_forward records a HIP
 * event on its stream before the first kernel and after each of the 18 conv stages.
 * fiunet_profile_read synchronises on them and returns, per stage, the average duration in
 * ms over the forwards recorded since enable/last read, the stage's algorithmic FLOPs
 * (2*B*H*W*9*Cin*Cout) and the name of the kernel instantiation that ran it
 * (names: 18 strings of name_stride bytes).  Not for use under stream capture. */
int fiunet_profile_enable(fiunet_ctx* ctx, int enable);
int fiunet_profile_read(fiunet_ctx* ctx, int* n_forwards, float* avg_ms, double* flops,
                        char* names, int name_stride);

/* Thread-local description of the last non-OK status. */
const char* fiunet_last_error_string(void);

int fiunet_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FIUNET_H */
