// Launch interface between the C-ABI layer (srx_api.cpp) and the gfx950
// kernels (srx_kernels.hip).  Plain structs, no HIP types beyond hipStream_t.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "srx_weights.h"

namespace srx {

// An activation tensor in HBM: NHWC fp16 with a zero border of `border` pixels
// on every side and the interior padded up to whole tiles.  `base` points at
// channel 0 of image pixel (0,0) of frame 0; pixel (y,x) of frame f is at
// base + f*frame + (y*pitch + x)*channels (halfs), for y,x in
// [-border, padded size + border).  Everything outside the true image stays 0:
// that is the zero padding every convolution of the graph reads.
struct Tensor
{
    uint16_t* base = nullptr;
    int channels = 0;
    int pitch = 0;        // pixels per row, border included
    int64_t frame = 0;    // halfs per frame
};

// 8-bit RGBA image (input texture or output render target) in HBM.
struct Image
{
    uint8_t* base = nullptr;
    int64_t pitch = 0;    // bytes per row
    int64_t frame = 0;    // bytes per frame
    int pixelBytes = 4;   // 4 = RGBA8; 3 = RGB8 (the two fused kernels only; such an image must be followed by at
                          // least one readable byte: its pixels are fetched as unaligned 32-bit words)
};

enum ConvKind
{
    CONV_5X5_RGB,     // conv1: 5x5, 4-half pixels (RGB0) -> 32, bias + ReLU
    CONV_3X3_32_64,   // conv2
    CONV_3X3_64_64,   // conv3
    CONV_UP1_PHASES,  // conv_up1: four phase 3x3 64->32 filters, writes the 2x grid
    CONV_3X3_32_32,   // conv4, conv5
    CONV_3X3_32_RGB   // conv6: 32 -> 3, no bias / ReLU; + NN2x(input) residual add + UNORM8
};

struct ConvLaunch
{
    ConvKind kind;
    Tensor in, out;
    const uint16_t* wpk = nullptr;   // packed MFMA fragments (device)
    const float* bias = nullptr;     // per output row of every 32-row tile (device), fp32
    int width = 0, height = 0;       // true size of the grid the tiles walk (the INPUT grid for CONV_UP1_PHASES)
    int batch = 1;
    // CONV_3X3_32_RGB only
    Tensor x0;                       // packed input tensor (RGB0), low resolution
    Image image;                     // RGBA8 output
    Tensor residual, sum;            // optional fp16 RGB0 taps (base == nullptr: not written)
};

// conv4 + conv5 + conv6 + residual add + UNORM8 in one kernel (rolling row window in LDS).
// a4 needs a border of at least 3 pixels and 64 pixels of slack to the right of the padded width.
struct HrFusedLaunch
{
    Tensor a4;
    Image input;                     // RGBA8 or RGB8 input image: the residual term is NN2x of its tensor value
    const uint16_t *wpk4 = nullptr, *wpk5 = nullptr, *wpk6 = nullptr;   // PackConv3x3K32, PackConv3x3K32, PackConv6Rotating
    const float *bias4 = nullptr, *bias5 = nullptr;
    Image image;                     // RGBA8 or RGB8 output
    int width2 = 0, height2 = 0;     // size of the 2x grid
    int batch = 1;
    bool mayAllocate = true;         // false: the launch must not allocate (no diagnostic probe buffer is created)
};
hipError_t LaunchHrFused(const HrFusedLaunch& l, int numCUs, hipStream_t s);

// image -> tensor + conv1 + conv2 + conv3 + conv_up1 in one kernel (rolling row windows in LDS):
// reads the RGBA8 input image, writes conv_up1's output on the 2x grid.
struct LrFusedLaunch
{
    Image image;                     // RGBA8 or RGB8 input
    Tensor a4;                       // conv_up1 output
    const uint16_t *wpk1 = nullptr;  // conv1 as 7 K-steps (PackConv1)
    const uint16_t *wpk2 = nullptr, *wpk3 = nullptr, *wpkUp1 = nullptr;   // PackConv3x3K32 / PackConvUp1RowSharedK32
    const float* bias = nullptr;     // conv_up1[32] | conv3[64] | conv2[64] | conv1[32], fp32
    int width = 0, height = 0;       // input size
    int batch = 1;
    bool mayAllocate = true;         // false: the launch must not allocate (no diagnostic probe buffer is created)
};
hipError_t LaunchLrFused(const LrFusedLaunch& l, int numCUs, hipStream_t s);

// Load the two fused kernels and the zero-fill kernel on the current device and raise the shared-memory limits
// (otherwise done by the first launch on that device).
hipError_t PrepareFusedKernels();

// How a fused kernel cuts `batch` frames into work items: strips of columns x segments of rows.
// `steps` is the number of row steps one item takes (segment rows + the pipeline's warm-up rows) and
// `rounds` how many items the busiest workgroup slot gets; steps * rounds / batch ranks plans.
struct StripPlan
{
    int strips = 0, segs = 0, segRows = 0, items = 0, steps = 0, rounds = 0;
    long cost = 0;      // estimated duration in row steps x 100 (what the plan minimises)
};
StripPlan PlanLrFused(int width, int height, int batch, int numCUs);
StripPlan PlanHrFused(int width2, int height2, int batch, int numCUs);
// Matrix-instruction work one workgroup issues per row step of an item, padding included (FLOPs = 2 x multiply-adds
// of the instruction shapes): LrFusedKernel 4 x 120 (conv_up1) + 2 x 144 (conv3) + 144 (conv2) v_mfma_f32_16x16x32_f16
// and 21 v_mfma_f32_32x32x16_f16 (conv1); HrFusedKernel 8 waves x 39 v_mfma_f32_16x16x32_f16 (an item of the wide
// layout; conv6's three per wave included).  items x steps x this = what a launch asks of the matrix units.
constexpr double kLrIssuedFlopsPerStep = (4 * 120 + 2 * 144 + 144) * 16384.0 + 21 * 32768.0;
constexpr double kHrIssuedFlopsPerStep = 8 * 39 * 16384.0;

// Zero `bytes` bytes at `p` (both multiples of 16) with a plain kernel.  Unlike a memset node, a kernel node keeps
// its place in a captured stream: inside a HIP graph a hipMemsetAsync of the workspace was seen to overlap the
// kernel that follows it.
hipError_t LaunchZero(void* p, size_t bytes, hipStream_t s);

hipError_t LaunchPack(const Image& src, const Tensor& dst, int width, int height, int batch, hipStream_t s);
hipError_t LaunchUnpack(const Tensor& src, const Image& dst, int width, int height, int batch, hipStream_t s);
hipError_t LaunchConv(const ConvLaunch& l, int numCUs, hipStream_t s);

// Generic layers (srx_layers.hip): run an operator made by ML::Util::Create*Layer on dense device
// tensors.  CONVOLUTION: (input, filter, bias or nullptr, output); UPSAMPLE_2D: (input, -, -, output);
// ELEMENT_WISE_ADD: (a, b, -, output), output may alias a or b.
hipError_t DispatchOperator(const ML::Operator& op, const void* input, const void* filterOrB, const void* bias,
                            void* output, hipStream_t s);
// CSImageToTensor / PSTensorToImage for a {batch, 3, height, width} tensor of any data type and layout.
hipError_t LaunchImageToTensor(const Image& src, void* tensor, ML::TensorDataType t, ML::TensorLayout layout,
                               int width, int height, int batch, hipStream_t s);
hipError_t LaunchTensorToImage(const void* tensor, const Image& dst, ML::TensorDataType t, ML::TensorLayout layout,
                               int width, int height, int batch, hipStream_t s);

constexpr int kTileW = 32;   // output pixels per MFMA column tile
constexpr int kTileH = 8;    // output rows per workgroup tile

}  // namespace srx
