/*
 * sr_oracle.h -- CPU restatement of the StarsX/SuperResolution inference path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared
 * against; nothing in the shipped library may call, link or load it.  Only
 * tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py use it.
 *
 * What it restates (all paths relative to /root/reference):
 *   weights.bin parser        SuperResolution/Content/LoadWeights.cpp:21-93
 *   fp32->fp16 truncation     SuperResolution/Common/Float16Compressor.h:41-57
 *   BN fold + bias            XUSG::ML::Util::CreateWeightTensors, declared at
 *                             SuperResolution/XUSG/MachineLearning/XUSGMachineLearning.h:856-858,
 *                             called at SuperResolution/Content/SuperResolution.cpp:318
 *   tensor strides            XUSGMachineLearning.h:860 (GetStrides)
 *   image -> tensor           SuperResolution/Content/Shaders/CSImageToTensor.hlsl:15-45
 *   graph topology + order    SuperResolution/Content/SuperResolution.cpp:142-306, :73-103
 *   tensor -> image           SuperResolution/Content/Shaders/PSTensorToImage.hlsl:14-35
 *                             + R8G8B8A8_UNORM render target (SuperResolution.cpp:423)
 *
 * Pinning status: the weight pre-processing is pinned bit-for-bit against the
 * reference's own Float16Compressor.h compiled from where it lies (oracle/_ref);
 * the graph arithmetic itself runs inside closed-source DirectML in the
 * reference, which ships no golden outputs and cannot run here, so the
 * end-to-end image result is "parity unpinned" against DirectML and is
 * cross-checked only against an independent torch-CPU evaluation
 * (oracle/crosscheck_torch.py).
 */
#ifndef SR_ORACLE_H
#define SR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar conversions ------------------------------------------------ */
uint16_t sro_f16_compress(float v);      /* truncating, Float16Compressor.h:41-57 */
void     sro_f16_compress_array(const float* in, uint16_t* out, size_t n);
uint16_t sro_f32_to_f16_rne(float v);    /* IEEE round-to-nearest-even           */
float    sro_f16_to_f32(uint16_t h);

/* ---- weights.bin ------------------------------------------------------- */
typedef struct sro_weights sro_weights;
/* Returns 0 on success; on failure prints the reference's message to stderr
 * (LoadWeights.cpp:26,38,45,63,82) and returns non-zero. */
int  sro_load_weights(const char* path, sro_weights** out, int verbose);
void sro_free_weights(sro_weights* w);
int  sro_weights_count(const sro_weights* w);
/* i-th tensor in file order */
const char*  sro_weights_name(const sro_weights* w, int i);
uint32_t     sro_weights_len(const sro_weights* w, int i);
const float* sro_weights_data(const sro_weights* w, int i);
const float* sro_weights_find(const sro_weights* w, const char* name, uint32_t* len);

/* layout: 0 = DEFAULT (NCHW), 1 = NHWC.  dtype: 0 = FLOAT16, 1 = FLOAT32.
 * sizes are always {N(out), C(in), H, W}.  filter_out must hold N*C*H*W
 * elements of the dtype, bias_out N elements.  Returns 1 if a bias was
 * produced, 0 if the layer has no scale/shift (conv6), <0 on a missing name. */
int  sro_create_weight_tensors(const sro_weights* w, const char* conv, const char* scale,
                               const char* shift, const uint32_t sizes[4], int layout, int dtype,
                               void* filter_out, void* bias_out);
void sro_get_strides(const uint32_t sizes[4], int layout, uint32_t strides[4]);

/* ---- the model --------------------------------------------------------- */
enum {
    SRO_T_INPUT = 0,   /* [3,H,W]      packed input tensor            */
    SRO_T_CONV1,       /* [32,H,W]                                     */
    SRO_T_CONV2,       /* [64,H,W]                                     */
    SRO_T_CONV3,       /* [64,H,W]                                     */
    SRO_T_CONV_UP1,    /* [32,2H,2W]                                   */
    SRO_T_CONV4,       /* [32,2H,2W]                                   */
    SRO_T_CONV5,       /* [32,2H,2W]                                   */
    SRO_T_CONV6,       /* [3,2H,2W]   residual                         */
    SRO_T_OUTPUT,      /* [3,2H,2W]   NN2x(input) + residual           */
    SRO_T_COUNT
};

/* dtype: 0 = fp16 storage (weights truncated to fp16, every stored activation
 *            rounded to fp16 RNE, fp32 accumulate), 1 = fp32 everywhere.
 * up1_mode: 0 = conv_up1 exactly as the reference graph states it (5x5 on the
 *            NN-upsampled tensor); 1 = the algebraically equivalent four-phase
 *            3x3 form on the low-resolution tensor with phase weights summed in
 *            double and rounded once to fp16 (the layer-wise HIP kernels'
 *            formulation, used to measure what that rounding costs); 2 = the
 *            row-shared form the fused HIP kernel evaluates: only the columns of
 *            the 5x5 filter are folded per output phase (summed in double, rounded
 *            once to fp16), the five row taps are used as they are. */
typedef struct sro_model sro_model;
sro_model* sro_model_create(const sro_weights* w, int dtype, int up1_mode);
void       sro_model_destroy(sro_model* m);

/* rgba: H rows of `pitch` bytes, 4 bytes per pixel (alpha ignored).
 * out_rgba: 2H rows of 2W*4 bytes, alpha = 255.  keep_tensors != 0 keeps every
 * layer's activations for sro_tensor().  Returns 0 on success. */
int sro_run(sro_model* m, const uint8_t* rgba, int w, int h, size_t pitch,
            uint8_t* out_rgba, int keep_tensors);
/* Planar CHW float view of a kept tensor (values are exactly fp16-representable
 * in dtype 0).  *c,*h,*w receive the shape. */
const float* sro_tensor(const sro_model* m, int id, int* c, int* h, int* w);

/* the pack / unpack steps alone (CSImageToTensor / PSTensorToImage + UNORM RT) */
void sro_image_to_tensor(const uint8_t* rgba, int w, int h, size_t pitch, int dtype, float* chw);
void sro_tensor_to_image(const float* chw, int w, int h, uint8_t* rgba);

#ifdef __cplusplus
}
#endif
#endif
