/*
 * srx.h -- C ABI of the MI355X (gfx950) super-resolution path.
 *
 * This is the drop-in boundary: plain pointers and sizes, integer status codes,
 * no exceptions and no framework types.  One srx_ctx owns everything a frame
 * needs on ONE GPU: the folded, truncated and MFMA-packed weights, the
 * activation workspace and the input / output images in HBM.  A ctx is not
 * thread-safe; work is submitted in order on the HIP stream the caller passes
 * (NULL = the default stream).  The host C++ classes SuperResolution /
 * SuperResolutionX in superresolution_amd/csrc sit on top of exactly these calls.
 *
 * Each entry point names the reference interface it replaces; paths are
 * relative to the reference tree (StarsX/SuperResolution).
 */
#ifndef SRX_H
#define SRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct srx_ctx srx_ctx;
typedef int32_t srx_status;

enum {
    SRX_OK = 0,
    SRX_ERR_INVALID = 1,   /* bad argument                                         */
    SRX_ERR_WEIGHTS = 2,   /* weights.bin missing / malformed (LoadWeights == false) */
    SRX_ERR_HIP = 3,       /* a HIP runtime call failed; see srx_last_error()      */
    SRX_ERR_NOMEM = 4,
    SRX_ERR_STATE = 5      /* call made out of order (e.g. read of a tensor not kept) */
};

/* srx_create flags */
enum {
    /* Keep every layer's activations in HBM so srx_read_tensor() can return
     * them (the debug taps used by the per-layer parity tests).  All tensors of
     * the whole batch must fit the workspace limit (16 GB, or SRX_WORKSPACE_MB),
     * about 1.2 KB per input pixel and frame; srx_create fails with
     * SRX_ERR_NOMEM otherwise. */
    SRX_FLAG_KEEP_TENSORS = 1u << 0,
    /* Print "Number of weight tensors" / "Loaded tensor: ..." like
     * Content/LoadWeights.cpp:50,75 while parsing weights.bin. */
    SRX_FLAG_VERBOSE_WEIGHTS = 1u << 1,
    /* Build the graph literally as SuperResolution::createResources does
     * (Content/SuperResolution.cpp:172-275): every layer made by the generic
     * layer factories, up1 materialised, conv_up1 as a direct 5x5 convolution.
     * All tensors are kept (srx_read_tensor works).  Slow VALU kernels: a parity
     * anchor, not the product path. */
    SRX_FLAG_GENERIC_LAYERS = 1u << 2,
    /* fp32 tensors and untruncated fp32 weights: the reference's mode when the
     * device lacks fp16 (Content/SuperResolution.cpp:161-163).  Implies
     * SRX_FLAG_GENERIC_LAYERS; srx_read_tensor then returns fp32. */
    SRX_FLAG_FP32 = 1u << 3,
    /* NHWC tensors and O,kh,kw,I filters inside the generic graph: the
     * reference's TensorLayout::NHWC switch (Content/SuperResolution.cpp:155-159).
     * Implies SRX_FLAG_GENERIC_LAYERS; srx_read_tensor still returns planar CHW. */
    SRX_FLAG_NHWC = 1u << 4
};

/* Tensor ids for srx_read_tensor(), in graph order
 * (Content/SuperResolution.cpp:172-275). */
enum {
    SRX_T_INPUT = 0,    /* [3,H,W]   CSImageToTensor output (m_modelInput)        */
    SRX_T_CONV1,        /* [32,H,W]                                               */
    SRX_T_CONV2,        /* [64,H,W]                                               */
    SRX_T_CONV3,        /* [64,H,W]                                               */
    SRX_T_CONV_UP1,     /* [32,2H,2W]                                             */
    SRX_T_CONV4,        /* [32,2H,2W]                                             */
    SRX_T_CONV5,        /* [32,2H,2W]                                             */
    SRX_T_CONV6,        /* [3,2H,2W] residual                                     */
    SRX_T_OUTPUT,       /* [3,2H,2W] NN2x(input) + residual (m_modelOutput)       */
    SRX_T_COUNT
};

/*
 * Replaces SuperResolution::Init + createResources + initResources
 * (Content/SuperResolution.cpp:25-55, :142-306, :452-712): parses weights.bin
 * (Content/LoadWeights.cpp:21-93), folds batch-norm scale into the filters and
 * truncates to fp16 exactly as ML::Util::CreateWeightTensors does
 * (XUSG/MachineLearning/XUSGMachineLearning.h:856-858), re-lays the filters out
 * for the MFMA kernels (the counterpart of DirectML's managed-weight
 * initialisation, SuperResolution.cpp:615-627) and allocates the workspace for
 * `batch` frames of in_w x in_h pixels on `device`.
 */
srx_status srx_create(srx_ctx** out, int device, const char* weights_path,
                      uint32_t in_w, uint32_t in_h, uint32_t batch, uint32_t flags);
srx_status srx_destroy(srx_ctx* ctx);

/* SuperResolution::GetOutWidth / GetOutHeight (Content/SuperResolution.cpp:132-140). */
srx_status srx_get_out_size(const srx_ctx* ctx, uint32_t* out_w, uint32_t* out_h);
srx_status srx_get_in_size(const srx_ctx* ctx, uint32_t* in_w, uint32_t* in_h, uint32_t* batch);

/*
 * Replace the texture upload of CreateTextureFromFile
 * (XUSG/Advanced/XUSGTextureLoader.h:50-65): copy one decoded 8-bit frame into
 * the ctx's RGBA8 input image in HBM.  `pitch` is the host row pitch in bytes.
 * srx_upload_rgb8 expands 3-channel data to RGBA (alpha 255) as the loader
 * does for 3-channel files (XUSGTextureLoader.h:20).
 */
srx_status srx_upload_rgba8(srx_ctx* ctx, uint32_t frame, const uint8_t* host_rgba,
                            size_t pitch, void* stream);
srx_status srx_upload_rgb8(srx_ctx* ctx, uint32_t frame, const uint8_t* host_rgb,
                           size_t pitch, void* stream);

/* Device-resident RGBA8 input / output images of one frame (for callers that
 * produce or consume frames on the GPU).  Rows are *pitch bytes apart. */
srx_status srx_input_device_ptr(srx_ctx* ctx, uint32_t frame, void** dptr, size_t* pitch);
srx_status srx_output_device_ptr(srx_ctx* ctx, uint32_t frame, void** dptr, size_t* pitch);

/* SuperResolution::ImageToTensors (Content/SuperResolution.cpp:57-71) +
 * CSImageToTensor.hlsl:15-45: RGBA8 image -> fp16 input tensor, all frames. */
srx_status srx_image_to_tensor(srx_ctx* ctx, void* stream);

/* SuperResolution::Process (Content/SuperResolution.cpp:73-103): the two
 * upsamples, seven convolutions and the residual add, all frames.  The last
 * kernel's epilogue also performs the R8G8B8A8_UNORM conversion of
 * PSTensorToImage.hlsl:14-35, so the RGBA8 output image is complete when this
 * returns to the stream. */
srx_status srx_process(srx_ctx* ctx, void* stream);

/* SuperResolution::Render (Content/SuperResolution.cpp:105-130) with the swap
 * chain replaced by a caller buffer: copies frame `frame` of the RGBA8 output
 * image to host memory (rows `pitch` bytes apart).  With
 * SRX_FLAG_KEEP_TENSORS the image is first re-derived from the fp16 output
 * tensor by the stand-alone unpack kernel.  Asynchronous on `stream`. */
srx_status srx_tensor_to_image(srx_ctx* ctx, uint32_t frame, uint8_t* host_rgba,
                               size_t pitch, void* stream);

/* ImageToTensors + Process in one call (what OnRender does on the first frame,
 * SuperResolutionX.cpp:347-353). */
srx_status srx_upscale(srx_ctx* ctx, void* stream);

/* The same for frames [first, first + count) only: lets a caller keep several
 * groups of frames in flight on different streams' copies while one stream
 * computes (the ctx's activation workspace is shared, so calls must be ordered
 * on ONE compute stream).  Not available with SRX_FLAG_KEEP_TENSORS. */
srx_status srx_upscale_frames(srx_ctx* ctx, uint32_t first, uint32_t count, void* stream);

/*
 * The same upscale for callers that must not allocate.  It has no counterpart in the
 * reference: it exists for code under HIP graph capture, tight serving loops and
 * harnesses that forbid hipMalloc.  One w x h RGBA8 image in device memory at `in`
 * (rows in_pitch bytes apart) is upscaled into the 2w x 2h RGBA8 image at `out`
 * (rows out_pitch bytes apart) with the ctx's weights, by the same two fused kernels
 * srx_upscale launches; the caller owns both images.  The one tensor those kernels
 * pass through device memory lives in a workspace that is part of the library's
 * device image, SRX_STATIC_SCRATCH_BYTES on every GPU the library runs on, held until
 * the process ends; the call performs no hipMalloc / hipFree and no other allocation
 * (its set-up was done by srx_create of `ctx`).  The ctx's own size and batch do not
 * matter; pointers and pitches must be multiples of 4.  Images whose workspace need,
 * 64 bytes per OUTPUT pixel plus borders, exceeds the static workspace are refused
 * with SRX_ERR_INVALID (640x360 fits, 1280x720 does not): use srx_upscale for those.
 * The call may be captured into a HIP graph (the workspace is cleared by a kernel,
 * not a memset node).  All calls on one GPU share that GPU's workspace: calls that overlap in time on one
 * GPU, from any ctx, thread or stream, race on it -- order them on one stream.
 * Asynchronous on `stream`.
 */
#define SRX_STATIC_SCRATCH_BYTES (64u << 20)
srx_status srx_upscale_static(srx_ctx* ctx, const void* in_rgba8, size_t in_pitch,
                              void* out_rgba8, size_t out_pitch,
                              uint32_t in_w, uint32_t in_h, void* stream);
/* SRX_STATIC_SCRATCH_BYTES of the loaded library. */
size_t srx_static_scratch_bytes(void);

/* WaitForGpu (SuperResolutionX.cpp:374-385). */
srx_status srx_sync(srx_ctx* ctx, void* stream);

/* Debug tap: copy a kept tensor of one frame to host as planar CHW fp16
 * (the reference's effective layout, Content/SuperResolution.cpp:155-159);
 * fp32 with SRX_FLAG_FP32.  Needs SRX_FLAG_KEEP_TENSORS or a generic-layer
 * ctx, with one exception: SRX_T_CONV_UP1, the only tensor the two fused
 * kernels of the fast path pass through HBM, can be read from any ctx for the
 * frames the workspace still holds: those of the last pass of the last
 * srx_upscale / srx_process / srx_upscale_frames (all frames when batch <=
 * srx_frames_per_launch(); with a larger batch the passes share one
 * workspace).  Any other frame, a read before the first upscale included,
 * fails with SRX_ERR_STATE; srx_upscale_static does not use the workspace.
 * `bytes` must be C*H*W*2 (*4 for fp32).  Synchronous. */
srx_status srx_read_tensor(srx_ctx* ctx, uint32_t frame, int32_t tensor_id,
                           void* fp16_chw_out, size_t bytes);
srx_status srx_tensor_shape(const srx_ctx* ctx, int32_t tensor_id,
                            uint32_t* c, uint32_t* h, uint32_t* w);

/* Host-only (no GPU needed): parse weights.bin and return the folded, truncated
 * fp16 filter (OIHW) and bias of conv layer 0..6 exactly as
 * ML::Util::CreateWeightTensors produces them
 * (XUSG/MachineLearning/XUSGMachineLearning.h:856-858, called at
 * Content/SuperResolution.cpp:318).  Either output may be NULL to query sizes;
 * layer 6 has no bias (*bias_n = 0).  Sizes are in elements. */
srx_status srx_fold_weights(const char* weights_path, int32_t layer, uint16_t* filter_out,
                            size_t filter_cap, uint16_t* bias_out, size_t bias_cap,
                            size_t* filter_n, size_t* bias_n);

/* Host-only: the same layer's filter re-laid out as MFMA A-operand fragments,
 * [tile][kstep][lane 0..63][8 halfs], as the kernels read it (the counterpart
 * of DirectML re-formatting managed weights at operator initialisation,
 * Content/SuperResolution.cpp:615-627).  packed_out may be NULL to query *n. */
srx_status srx_packed_weights(const char* weights_path, int32_t layer, uint16_t* packed_out,
                              size_t cap, size_t* n);

/* Host-only: the fragments of layer 0..6 as the two FUSED kernels keep them in
 * registers (they differ from srx_packed_weights where a fused kernel uses
 * another MFMA shape or order): conv1 as in srx_packed_weights (seven K-steps of four taps); conv_up1 in
 * its row-shared form, [wave = 2 * output half + column phase][30 fragments][lane][8]
 * (fragments 0..17: filter rows 0, 2, 4 with the five filter columns folded into
 * three column taps per phase; 18..29: filter rows 1, 3); conv2, conv3, conv4 and conv5 as
 * v_mfma_f32_16x16x32_f16 fragments [tile][kstep][output half][lane][8] (the
 * rows of the two halves are interleaved: row r of half h is output
 * 8 * (r >> 2) + 4 * h + (r & 3) of the tile, so that a lane's accumulators of
 * both halves are eight consecutive channels); conv6 as row-streaming fragments
 * [row mod 4][tap j][lane][8]. */
srx_status srx_fused_weights(const char* weights_path, int32_t layer, uint16_t* packed_out,
                             size_t cap, size_t* n);

/* Profiling hooks used by bench.py: number of kernels one srx_upscale()
 * launches per batch chunk, their names, and a launch of stage `i` alone. */
int32_t     srx_num_stages(const srx_ctx* ctx);
const char* srx_stage_name(const srx_ctx* ctx, int32_t i);
srx_status  srx_launch_stage(srx_ctx* ctx, int32_t i, void* stream);
/* Matrix-instruction FLOPs one launch of fused stage `i` issues for a pass of
 * srx_frames_per_launch() frames, padding included (items x row steps x the
 * MFMAs a workgroup issues per step): divided by the launch time it is the
 * rate the matrix units actually run at, whatever share of it is useful.
 * No counterpart in the reference (measurement only). */
srx_status  srx_stage_issued_flops(const srx_ctx* ctx, int32_t i, double* flops);
/* Frames one srx_launch_stage() of a convolution stage covers (the activation
 * workspace holds this many frames; srx_process walks the batch in such passes). */
uint32_t    srx_frames_per_launch(const srx_ctx* ctx);
/* What the ctx's device is: compute units, peak engine clock in kHz and the architecture name
 * (the counterpart of the DXGI_ADAPTER_DESC1 the reference reads when it picks its adapter,
 * SuperResolutionX.cpp:76-116).  Any
 * output pointer may be NULL; arch receives at most arch_cap bytes including the terminator. */
srx_status  srx_device_info(const srx_ctx* ctx, int32_t* compute_units, int32_t* clock_khz,
                            char* arch, size_t arch_cap);

/*
 * Frame pipe: the counterpart of the reference's frames in flight
 * (FrameCount = 3 back buffers with one fence value each, SuperResolutionX.h:49,
 * :69-72; MoveToNextFrame, SuperResolutionX.cpp:388-405) for a headless stream of
 * frames that starts and ends in host memory.  The ctx's `batch` frames are
 * divided into slots of `frames_per_slot` frames (at least two slots).  Each
 * slot owns pinned host buffers for its input and output frames; a submitted
 * slot is copied to the GPU (one linear copy), upscaled and copied back (one
 * linear copy) on three HIP streams the pipe owns, so the copies of one slot
 * overlap the kernels of another.  The two copy streams are created at the
 * device's highest stream priority: streams overlap only from different
 * hardware queues, the runtime deals a process's streams over a few queues per
 * priority level, and a caller with streams of its own at the default
 * priority must not be able to push a copy stream onto the compute stream's
 * queue.
 *
 *   srx_pipe_input  : the slot's pinned RGBA8 input frames (tightly packed:
 *                     pitch = in_w*4, frame stride = pitch*in_h) for the caller to fill
 *   srx_pipe_submit : enqueue upload + upscale + download of the slot; returns at once
 *   srx_pipe_wait   : block until the slot's output frames are in its pinned
 *                     output buffer (pitch = out_w*4) and return it; the buffers of
 *                     a slot may be touched again only after this returned
 * A pipe is driven by one host thread; while it exists the ctx must not be
 * used through other entry points, and srx_destroy of the ctx fails with
 * SRX_ERR_STATE until the pipe has been destroyed.
 */
typedef struct srx_pipe srx_pipe;
srx_status srx_pipe_create(srx_ctx* ctx, uint32_t frames_per_slot, srx_pipe** out);
srx_status srx_pipe_destroy(srx_pipe* pipe);
uint32_t   srx_pipe_slots(const srx_pipe* pipe);
/* Channels per pixel of the pinned buffers, 4 = RGBA8 (default) or 3 = RGB8: the
 * loader's expansion of 3-channel data (XUSG/Advanced/XUSGTextureLoader.h:20) and
 * SaveImage's dropped alpha (SuperResolutionX.cpp:423-443) then happen on the GPU,
 * inside the two fused kernels (they read and write three bytes per pixel; there
 * is no repacking pass), and a quarter less data crosses PCIe.  Before the first
 * srx_pipe_submit. */
srx_status srx_pipe_set_channels(srx_pipe* pipe, uint32_t in_channels, uint32_t out_channels);
srx_status srx_pipe_input(srx_pipe* pipe, uint32_t slot, uint8_t** host_rgba,
                          size_t* pitch, size_t* frame_stride);
srx_status srx_pipe_submit(srx_pipe* pipe, uint32_t slot);
srx_status srx_pipe_wait(srx_pipe* pipe, uint32_t slot, const uint8_t** host_rgba_out,
                         size_t* pitch, size_t* frame_stride);

/*
 * Generic layers: the operators the reference's layer factories build
 * (ML::Util::CreateConvolutionLayer / CreateUpsampleLayer / CreateAdditionLayer,
 * XUSG/MachineLearning/XUSGMachineLearning.h:847-854), run directly on dense
 * DEVICE tensors of sizes {N, C, H, W}.  dtype: 0 = fp16 storage, 1 = fp32
 * (arithmetic is fp32 either way); layout: 0 = NCHW, 1 = NHWC.
 *   conv2d     : cross-correlation, stride 1, zero padding ceil((k-1)/2) before and
 *                floor((k-1)/2) after, filter {O, I, kh, kw} (O,kh,kw,I order for NHWC),
 *                kh, kw <= 7; bias != NULL adds the bias and applies ReLU
 *                (useBiasAndActivation)
 *   upsample2d : integer scales; interpolation 0 = nearest neighbour (pixel
 *                replication), 1 = linear (half-pixel centres, clamped at the edges)
 *   add        : element-wise, `out` may alias `a` or `b`
 * Asynchronous on `stream`.
 */
/* Host-only: what the layer factory of `kind` (0 = CreateConvolutionLayer with
 * param = filter sizes {O, I, kh, kw}; 1 = CreateUpsampleLayer with param =
 * {scale_x, scale_y, -, -}; 2 = CreateAdditionLayer) answers for these input
 * sizes: the output sizes, and *in_bytes / *out_bytes raised to the byte sizes
 * of the input and output tensors if those are larger (the reference sizes its
 * ping-pong buffers with these running maxima, Content/SuperResolution.cpp:166-168). */
srx_status srx_layer_plan(int32_t kind, int32_t dtype, const uint32_t in_sizes[4], const uint32_t param[4],
                          uint32_t out_sizes[4], uint64_t* in_bytes, uint64_t* out_bytes);
srx_status srx_layer_conv2d(int32_t dtype, int32_t layout, const void* in, const uint32_t in_sizes[4],
                            const void* filter, const uint32_t filter_sizes[4], const void* bias,
                            void* out, void* stream);
srx_status srx_layer_upsample2d(int32_t dtype, int32_t layout, const void* in, const uint32_t in_sizes[4],
                                uint32_t scale_x, uint32_t scale_y, int32_t interpolation, void* out, void* stream);
srx_status srx_layer_add(int32_t dtype, const void* a, const void* b, void* out, size_t count, void* stream);

/* Message of the last failure on this thread (never NULL). */
const char* srx_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
