// Host-side weight path of the MI355X super-resolution library: the weights.bin
// parser, the batch-norm fold + fp16 truncation, and the re-layout of the
// filters into MFMA operand fragments.
//
// Interfaces follow the reference so that its call sites read the same:
//   LoadWeights(fpath, weightMap)          Content/LoadWeights.h:13, LoadWeights.cpp:21-93
//   ML::Util::CreateWeightTensors(...)     XUSG/MachineLearning/XUSGMachineLearning.h:856-858
//   ML::Util::GetStrides(...)              XUSGMachineLearning.h:860
//   WeightsType / WeightMapType            XUSGMachineLearning.h:834-835
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace srx {
namespace ML {

using WeightsType = std::vector<float>;
using WeightMapType = std::map<std::string, WeightsType>;

enum class TensorDataType : uint32_t { UNKNOWN, FLOAT32, FLOAT16 };   // XUSGMachineLearning.h:13-24
enum class TensorLayout : uint32_t { DEFAULT, NHWC };                  // XUSGMachineLearning.h:828-832

// fp32 -> fp16 by truncation toward zero, the conversion the reference applies
// to weights and biases (Common/Float16Compressor.h:41-57): overflow beyond
// 65504 gives infinity, values below 2^-14 become truncated half subnormals.
uint16_t Float32ToFloat16Trunc(float v);
float Float16ToFloat32(uint16_t h);
// IEEE round-to-nearest-even, used only for the summed phase weights of conv_up1.
uint16_t Float32ToFloat16(float v);

enum class InterpolationType : uint32_t { NEAREST_NEIGHBOR, LINEAR };   // XUSGMachineLearning.h:177-181

// What a layer factory returns: the counterpart of the reference's compiled DirectML operator
// (XUSG::ML::Operator), reduced to the description a kernel launch needs.  Sizes are {N, C, H, W}.
struct Operator
{
    enum class Kind : uint32_t { NONE, UPSAMPLE_2D, CONVOLUTION, ELEMENT_WISE_ADD };
    Kind kind = Kind::NONE;
    TensorDataType dataType = TensorDataType::FLOAT16;
    TensorLayout layout = TensorLayout::DEFAULT;
    uint32_t inputSizes[4] = {}, outputSizes[4] = {};
    uint32_t filterSizes[4] = {};                  // {O, I, kh, kw}
    uint32_t padTop = 0, padLeft = 0;              // ceil((k-1)/2); bottom / right get floor((k-1)/2)
    bool useBiasAndActivation = false;             // bias + fused ReLU
    uint32_t scaleX = 1, scaleY = 1;
    InterpolationType interpolation = InterpolationType::NEAREST_NEIGHBOR;
};

class Util
{
public:
    Util(TensorDataType dataType = TensorDataType::FLOAT16, TensorLayout layout = TensorLayout::DEFAULT)
        : m_dataType(dataType), m_layout(layout) {}

    // filterSizes = {N (out), C (in), H, W}.  filter = scale[n] * w (one fp32
    // multiply) converted to the tensor data type; bias = shift[n].  With
    // scaleLayerName == nullptr the filter passes through and no bias is made.
    // Returns false when a named tensor is missing or has the wrong length.
    bool CreateWeightTensors(const WeightMapType& weights, const char* convLayerName,
                             const char* scaleLayerName, const char* shiftLayerName,
                             const uint32_t filterSizes[4], std::vector<uint8_t>& filterWeightsOut,
                             std::vector<uint8_t>& biasWeightsOut) const;

    static void GetStrides(const uint32_t sizes[4], TensorLayout layout, uint32_t stridesOut[4]);

    // Layer factories with the reference's signatures minus the D3D12 device
    // (XUSGMachineLearning.h:847-854).  They validate the sizes, report the byte sizes of the
    // input and output tensors (the reference tracks the running maximum: the arguments are
    // raised, never lowered), fill outputSizes and describe the operator.
    bool CreateUpsampleLayer(const uint32_t inputSizes[4], uint64_t& inputBufferRequiredSize,
                             uint64_t& outputBufferRequiredSize, uint32_t outputSizes[4], Operator& opOut,
                             uint32_t scaleSizeX = 2, uint32_t scaleSizeY = 2,
                             InterpolationType interpolationType = InterpolationType::NEAREST_NEIGHBOR) const;
    // Cross-correlation, stride 1, zero padding ceil((k-1)/2) before and floor((k-1)/2) after, so the
    // output has the input's height and width; bias and ReLU iff useBiasAndActivation.
    bool CreateConvolutionLayer(const uint32_t inputSizes[4], const uint32_t* filterSizes, bool useBiasAndActivation,
                                uint64_t& inputBufferRequiredSize, uint64_t& outputBufferRequiredSize,
                                uint32_t outputSizes[4], Operator& opOut) const;
    bool CreateAdditionLayer(const uint32_t inputSizes[4], Operator& opOut) const;

    TensorDataType GetDataType() const { return m_dataType; }
    TensorLayout GetLayout() const { return m_layout; }

private:
    TensorDataType m_dataType;
    TensorLayout m_layout;
};

}  // namespace ML

// Loads weight values from a file.  Same contract as the reference function:
// false (and the reason on stderr) when the file cannot be opened, the count is
// negative, a name is longer than 255 bytes or the data is short.
bool LoadWeights(const std::string& fpath, ML::WeightMapType& weightMap, bool verbose = true);

// ---------------------------------------------------------------------------
// Layer table of the sample CNN (Content/SuperResolution.cpp:188-268).
// ---------------------------------------------------------------------------
struct ConvLayerDesc
{
    const char* conv;
    const char* scale;   // nullptr: no batch-norm, no bias, no activation (conv6)
    const char* shift;
    uint32_t sizes[4];   // {out, in, kh, kw}
};
constexpr int kNumConvLayers = 7;           // c_numConvLayers, Content/SuperResolution.h:64
extern const ConvLayerDesc kConvLayers[kNumConvLayers];

// Folded + truncated fp16 weights of every layer (OIHW) and biases.
struct FoldedWeights
{
    std::vector<uint16_t> filter[kNumConvLayers];
    std::vector<uint16_t> bias[kNumConvLayers];   // empty for conv6
};
bool FoldWeights(const ML::WeightMapType& weights, FoldedWeights& out);

// ---------------------------------------------------------------------------
// MFMA operand packing.  A "fragment" is what one wave feeds v_mfma_f32_32x32x16_f16
// as its A operand for one K-step: 64 lanes x 8 halfs, lane l holding
// A[row = l & 31][k = 8 * (l >> 5) + j], j = 0..7.  Rows are output channels.
// Packed arrays are [tile][kstep][lane][8].
// ---------------------------------------------------------------------------
// 3x3 conv, CIN multiple of 16: kstep = tap * (CIN/16) + cc, k = 16*cc + 8*h + j.
// Output channels are padded with zero rows to a multiple of 32.
std::vector<uint16_t> PackConv3x3(const std::vector<uint16_t>& oihw, int cout, int cin);
// Same filter as fragments of v_mfma_f32_16x16x32_f16: lane l holds
// A[row = l & 15][k = 8 * (l >> 4) + j]; one kstep per tap covers 32 input channels;
// output channels are padded with zero rows to 16.  Layout [tap][lane][8].
std::vector<uint16_t> PackConv3x3Rows16(const std::vector<uint16_t>& oihw, int cout, int cin /* == 32 */);
// conv6 (3x3, 32 -> 3, no bias) for a kernel that streams input rows: fragments of
// v_mfma_f32_16x16x32_f16 whose 16 output rows are four groups of (R, G, B, 0), group g holding the
// output row y with y mod 4 == g.  Input row R (rho = R mod 4) contributes filter row 2 to output row
// R-1, filter row 1 to R and filter row 0 to R+1; the fourth group's rows are zero.  One K-step per
// horizontal tap j covers the 32 input channels.  Layout [rho][j][lane][8]; oihw is [3,32,3,3].
std::vector<uint16_t> PackConv6Rotating(const std::vector<uint16_t>& oihw);
// conv1 (5x5 on RGB0 pixels of 4 halfs): 7 ksteps of 4 taps x 4 channels,
// k = 4 * (tap - 4*kstep) + channel; taps 25..27 and channel 3 are zero.
std::vector<uint16_t> PackConv1(const std::vector<uint16_t>& oihw /*[32,3,5,5]*/);
// conv_up1 in four-phase form: the 5x5 filter on the 2x nearest-neighbour
// upsampled tensor equals, for output phase (py,px), a 3x3 filter on the
// low-resolution tensor whose taps are sums of the 5x5 taps that read the same
// low-resolution pixel.  Sums are formed in double and rounded once to fp16.
// Result: tile = py*2+px, layout as PackConv3x3 with cin = 64.
std::vector<uint16_t> PackConvUp1Phases(const std::vector<uint16_t>& oihw /*[32,64,5,5]*/);
// A 3x3 filter (cin a multiple of 32) as A fragments of v_mfma_f32_16x16x32_f16: lane l holds
// A[row = l & 15][k = 8 * (l >> 4) + j].  Layout [tile of 32 outputs][kstep = tap * (cin/32) + kc]
// [half: outputs 0-15 / 16-31 of the tile][lane][8]; a K-step covers input channels 32*kc .. 32*kc+31
// of one tap.  With `interleave`, row r of half h holds output 8 * (r >> 2) + 4 * h + (r & 3) of the tile instead
// of 16 * h + r: the lane group that gets D rows 4q .. 4q+3 of both halves then owns outputs 8q .. 8q+7.
std::vector<uint16_t> PackConv3x3K32(const std::vector<uint16_t>& oihw, int cout, int cin, bool interleave = false);
// conv_up1 in the row-shared form the fused kernel evaluates.  Only the COLUMNS of the 5x5 filter are folded per
// output phase px (three column taps, each the sum of the 5x5 taps that read the same low-resolution column, formed
// in double and rounded once to fp16: ConvUp1ColumnFilters, [px][32][64][5][3]); the five row taps stay apart.  With
//   S(y) = tap0 x[y-1] + tap2 x[y] + tap4 x[y+1]   and   T(y) = tap1 x[y-1] + tap3 x[y]
// the two output rows of input row y are  out(2y) = S(y) + T(y)  and  out(2y+1) = S(y) + T(y+1): 30 products per
// input pixel and phase px instead of the 36 of two 3x3 phase filters.
// Layout [wave = 2 * outputHalf + px][30 fragments][lane][8], A fragments of v_mfma_f32_16x16x32_f16 (lane l: output
// 16 * outputHalf + (l & 15), channels 32 kc + 8 (l >> 4) ..): fragments 0..17 are S, index (rowSlot * 3 + column tap) *
// 2 + kc with rowSlot 0..2 = input rows y-1, y, y+1; fragments 18..29 are T(y+1), same index with rowSlot 0..1 = input
// rows y, y+1.
std::vector<uint16_t> ConvUp1ColumnFilters(const std::vector<uint16_t>& oihw /*[32,64,5,5]*/);
std::vector<uint16_t> PackConvUp1RowSharedK32(const std::vector<uint16_t>& oihw /*[32,64,5,5]*/);
// The phase filters themselves, [4][32][64][3][3] fp16 (for tests).
std::vector<uint16_t> ConvUp1PhaseFilters(const std::vector<uint16_t>& oihw);

}  // namespace srx
