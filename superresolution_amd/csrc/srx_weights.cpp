#include "srx_weights.h"

#include <algorithm>

#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>

namespace srx {

// ---------------------------------------------------------------------------
// scalar conversions
// ---------------------------------------------------------------------------
namespace ML {

uint16_t Float32ToFloat16Trunc(float v)
{
    uint32_t u;
    std::memcpy(&u, &v, sizeof(u));
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t mag = u & 0x7fffffffu;
    const int exp = static_cast<int>(mag >> 23) - 127;   // unbiased fp32 exponent

    if (mag > 0x7f800000u)                               // NaN keeps its top payload bits
    {
        const uint32_t payload = (mag >> 13) & 0x3ffu;
        return static_cast<uint16_t>(sign | 0x7c00u | (payload ? payload : 1u));
    }
    if (mag > 0x477fe000u) return static_cast<uint16_t>(sign | 0x7c00u);   // > 65504
    if (exp >= -14)                                      // normal half: drop 13 mantissa bits
        return static_cast<uint16_t>(sign | (static_cast<uint32_t>(exp + 15) << 10) | ((mag >> 13) & 0x3ffu));
    if (exp < -24) return static_cast<uint16_t>(sign);   // below the smallest subnormal
    // subnormal half: value / 2^-24, fraction dropped
    const uint32_t mant = (mag & 0x7fffffu) | 0x800000u; // 1.m as a 24-bit integer, value = mant * 2^(exp-23)
    return static_cast<uint16_t>(sign | (mant >> (-exp - 1)));
}

float Float16ToFloat32(uint16_t h)
{
    const uint32_t sign = (static_cast<uint32_t>(h) & 0x8000u) << 16;
    const int exp = (h >> 10) & 0x1f;
    const uint32_t mant = h & 0x3ffu;
    float mag;
    if (exp == 0) mag = std::ldexp(static_cast<float>(mant), -24);
    else if (exp == 31)
    {
        const uint32_t u = 0x7f800000u | (mant << 13);
        std::memcpy(&mag, &u, sizeof(mag));
    }
    else mag = std::ldexp(static_cast<float>(mant | 0x400u), exp - 25);
    uint32_t u;
    std::memcpy(&u, &mag, sizeof(u));
    u |= sign;
    float out;
    std::memcpy(&out, &u, sizeof(out));
    return out;
}

uint16_t Float32ToFloat16(float v)
{
    // Round to nearest even by comparing the two neighbouring halfs.
    if (v != v) return 0x7e00u;
    const uint16_t lo = Float32ToFloat16Trunc(v);        // toward zero
    if ((lo & 0x7fffu) >= 0x7c00u) return lo;
    const uint16_t hi = static_cast<uint16_t>(lo + 1);   // next away from zero (may become inf)
    const double x = std::fabs(static_cast<double>(v));
    const double a = std::fabs(static_cast<double>(Float16ToFloat32(lo)));
    const double b = (hi & 0x7fffu) >= 0x7c00u ? 65536.0 : std::fabs(static_cast<double>(Float16ToFloat32(hi)));
    const double da = x - a, db = b - x;
    if (da < db) return lo;
    if (db < da) return hi;
    return (lo & 1u) ? hi : lo;
}

namespace {

uint64_t tensorBytes(const uint32_t sizes[4], TensorDataType t)
{
    return uint64_t(sizes[0]) * sizes[1] * sizes[2] * sizes[3] * (t == TensorDataType::FLOAT16 ? 2 : 4);
}

bool validSizes(const uint32_t sizes[4])
{
    return sizes && sizes[0] && sizes[1] && sizes[2] && sizes[3] &&
           uint64_t(sizes[0]) * sizes[1] * sizes[2] * sizes[3] < (uint64_t(1) << 40);
}

}  // namespace

bool Util::CreateUpsampleLayer(const uint32_t inputSizes[4], uint64_t& inputBufferRequiredSize,
                               uint64_t& outputBufferRequiredSize, uint32_t outputSizes[4], Operator& opOut,
                               uint32_t scaleSizeX, uint32_t scaleSizeY, InterpolationType interpolationType) const
{
    if (!validSizes(inputSizes) || !outputSizes || scaleSizeX == 0 || scaleSizeY == 0) return false;
    if (m_dataType == TensorDataType::UNKNOWN) return false;
    opOut = Operator{};
    opOut.kind = Operator::Kind::UPSAMPLE_2D;
    opOut.dataType = m_dataType;
    opOut.layout = m_layout;
    opOut.scaleX = scaleSizeX;
    opOut.scaleY = scaleSizeY;
    opOut.interpolation = interpolationType;
    outputSizes[0] = inputSizes[0];
    outputSizes[1] = inputSizes[1];
    outputSizes[2] = inputSizes[2] * scaleSizeY;
    outputSizes[3] = inputSizes[3] * scaleSizeX;
    if (!validSizes(outputSizes)) return false;
    for (int i = 0; i < 4; ++i)
    {
        opOut.inputSizes[i] = inputSizes[i];
        opOut.outputSizes[i] = outputSizes[i];
    }
    inputBufferRequiredSize = std::max(inputBufferRequiredSize, tensorBytes(inputSizes, m_dataType));
    outputBufferRequiredSize = std::max(outputBufferRequiredSize, tensorBytes(outputSizes, m_dataType));
    return true;
}

bool Util::CreateConvolutionLayer(const uint32_t inputSizes[4], const uint32_t* filterSizes, bool useBiasAndActivation,
                                  uint64_t& inputBufferRequiredSize, uint64_t& outputBufferRequiredSize,
                                  uint32_t outputSizes[4], Operator& opOut) const
{
    if (!validSizes(inputSizes) || !validSizes(filterSizes) || !outputSizes) return false;
    if (m_dataType == TensorDataType::UNKNOWN) return false;
    if (filterSizes[1] != inputSizes[1]) return false;                  // group count 1: filter depth = input channels
    opOut = Operator{};
    opOut.kind = Operator::Kind::CONVOLUTION;
    opOut.dataType = m_dataType;
    opOut.layout = m_layout;
    opOut.useBiasAndActivation = useBiasAndActivation;
    opOut.padTop = filterSizes[2] / 2;                                   // ceil((k - 1) / 2)
    opOut.padLeft = filterSizes[3] / 2;
    outputSizes[0] = inputSizes[0];
    outputSizes[1] = filterSizes[0];
    outputSizes[2] = inputSizes[2];
    outputSizes[3] = inputSizes[3];
    for (int i = 0; i < 4; ++i)
    {
        opOut.inputSizes[i] = inputSizes[i];
        opOut.outputSizes[i] = outputSizes[i];
        opOut.filterSizes[i] = filterSizes[i];
    }
    inputBufferRequiredSize = std::max(inputBufferRequiredSize, tensorBytes(inputSizes, m_dataType));
    outputBufferRequiredSize = std::max(outputBufferRequiredSize, tensorBytes(outputSizes, m_dataType));
    return true;
}

bool Util::CreateAdditionLayer(const uint32_t inputSizes[4], Operator& opOut) const
{
    if (!validSizes(inputSizes) || m_dataType == TensorDataType::UNKNOWN) return false;
    opOut = Operator{};
    opOut.kind = Operator::Kind::ELEMENT_WISE_ADD;
    opOut.dataType = m_dataType;
    opOut.layout = m_layout;
    for (int i = 0; i < 4; ++i) opOut.inputSizes[i] = opOut.outputSizes[i] = inputSizes[i];
    return true;
}

bool Util::CreateWeightTensors(const WeightMapType& weights, const char* convLayerName,
                               const char* scaleLayerName, const char* shiftLayerName,
                               const uint32_t filterSizes[4], std::vector<uint8_t>& filterWeightsOut,
                               std::vector<uint8_t>& biasWeightsOut) const
{
    const uint32_t N = filterSizes[0], C = filterSizes[1], H = filterSizes[2], W = filterSizes[3];
    const auto filterIt = weights.find(convLayerName);
    if (filterIt == weights.end() || filterIt->second.size() != size_t(N) * C * H * W) return false;
    const WeightsType& filter = filterIt->second;

    const WeightsType* scale = nullptr;
    const WeightsType* shift = nullptr;
    if (scaleLayerName)
    {
        const auto s = weights.find(scaleLayerName);
        const auto b = shiftLayerName ? weights.find(shiftLayerName) : weights.end();
        if (s == weights.end() || b == weights.end()) return false;
        if (s->second.size() != N || b->second.size() != N) return false;
        scale = &s->second;
        shift = &b->second;
    }

    const bool half = m_dataType == TensorDataType::FLOAT16;
    const size_t elem = half ? sizeof(uint16_t) : sizeof(float);
    const auto put = [&](std::vector<uint8_t>& dst, size_t idx, float v)
    {
        if (half)
        {
            const uint16_t h = Float32ToFloat16Trunc(v);
            std::memcpy(&dst[idx * elem], &h, elem);
        }
        else std::memcpy(&dst[idx * elem], &v, elem);
    };

    filterWeightsOut.assign(size_t(N) * C * H * W * elem, 0);
    size_t o = 0;
    for (uint32_t n = 0; n < N; ++n)
    {
        const auto folded = [&](size_t idx) -> float
        {
            if (!scale) return filter[idx];
            volatile float p = (*scale)[n] * filter[idx];   // a single fp32 multiply, never fused
            return p;
        };
        if (m_layout == TensorLayout::NHWC)
        {
            for (uint32_t h = 0; h < H; ++h)
                for (uint32_t w = 0; w < W; ++w)
                    for (uint32_t c = 0; c < C; ++c)
                        put(filterWeightsOut, o++, folded(((size_t(n) * C + c) * H + h) * W + w));
        }
        else
            for (size_t i = 0; i < size_t(C) * H * W; ++i)
                put(filterWeightsOut, o++, folded(size_t(n) * C * H * W + i));
    }

    biasWeightsOut.clear();
    if (shift)
    {
        biasWeightsOut.assign(size_t(N) * elem, 0);
        for (uint32_t n = 0; n < N; ++n) put(biasWeightsOut, n, (*shift)[n]);
    }
    return true;
}

void Util::GetStrides(const uint32_t sizes[4], TensorLayout layout, uint32_t stridesOut[4])
{
    const uint32_t C = sizes[1], H = sizes[2], W = sizes[3];
    stridesOut[0] = C * H * W;
    if (layout == TensorLayout::NHWC)
    {
        stridesOut[1] = 1;
        stridesOut[2] = W * C;
        stridesOut[3] = C;
    }
    else
    {
        stridesOut[1] = H * W;
        stridesOut[2] = W;
        stridesOut[3] = 1;
    }
}

}  // namespace ML

// ---------------------------------------------------------------------------
// weights.bin
// ---------------------------------------------------------------------------
bool LoadWeights(const std::string& fpath, ML::WeightMapType& weightMap, bool verbose)
{
    std::ifstream input(fpath, std::ifstream::binary);
    if (!input.is_open() || !input.good())
    {
        std::cerr << "Unable to open weight file: " << fpath << std::endl;
        return false;
    }

    const auto readU32 = [&input](uint32_t& v) -> bool
    {
        input.read(reinterpret_cast<char*>(&v), sizeof(v));
        return input.gcount() == static_cast<std::streamsize>(sizeof(v));
    };

    uint32_t rawCount = 0;
    if (!readU32(rawCount) || static_cast<int32_t>(rawCount) < 0)
    {
        std::cerr << "Invalid weight map file: " << fpath << std::endl;
        return false;
    }
    if (verbose) std::cout << "Number of weight tensors: " << rawCount << std::endl;

    for (uint32_t t = 0; t < rawCount; ++t)
    {
        uint32_t nameLen = 0, numFloats = 0;
        if (!readU32(nameLen))
        {
            std::cerr << "Invalid tensor data" << std::endl;
            return false;
        }
        if (nameLen > 255)
        {
            std::cerr << "name_len exceeds c_bufferLength: " << nameLen << " vs " << 255 << std::endl;
            return false;
        }
        std::string name(nameLen, '\0');
        input.read(nameLen ? &name[0] : nullptr, nameLen);
        if (input.gcount() != static_cast<std::streamsize>(nameLen) || !readU32(numFloats))
        {
            std::cerr << "Invalid tensor data" << std::endl;
            return false;
        }
        // The reference builds the key from a C string, so an embedded NUL ends the name.
        name.resize(std::strlen(name.c_str()));

        // Guard against a length field that runs past the end of the file before allocating.
        const auto here = input.tellg();
        input.seekg(0, std::ios::end);
        const auto fileEnd = input.tellg();
        input.seekg(here);
        if (static_cast<uint64_t>(fileEnd - here) < uint64_t(numFloats) * sizeof(float))
        {
            std::cerr << "Invalid tensor data" << std::endl;
            return false;
        }

        ML::WeightsType& dst = weightMap[name];
        dst.assign(numFloats, 0.0f);
        input.read(reinterpret_cast<char*>(dst.data()), std::streamsize(sizeof(float)) * numFloats);
        if (verbose) std::cout << "Loaded tensor: " << name << " -> " << numFloats << std::endl;
    }
    return true;
}

// ---------------------------------------------------------------------------
// layer table + fold
// ---------------------------------------------------------------------------
const ConvLayerDesc kConvLayers[kNumConvLayers] = {
    {"conv1/weights", "conv1/BatchNorm/scale", "conv1/BatchNorm/shift", {32, 3, 5, 5}},
    {"conv2/weights", "conv2/BatchNorm/scale", "conv2/BatchNorm/shift", {64, 32, 3, 3}},
    {"conv3/weights", "conv3/BatchNorm/scale", "conv3/BatchNorm/shift", {64, 64, 3, 3}},
    {"conv_up1/conv/weights", "conv_up1/conv/BatchNorm/scale", "conv_up1/conv/BatchNorm/shift", {32, 64, 5, 5}},
    {"conv4/weights", "conv4/BatchNorm/scale", "conv4/BatchNorm/shift", {32, 32, 3, 3}},
    {"conv5/weights", "conv5/BatchNorm/scale", "conv5/BatchNorm/shift", {32, 32, 3, 3}},
    {"conv6/weights", nullptr, nullptr, {3, 32, 3, 3}},
};

bool FoldWeights(const ML::WeightMapType& weights, FoldedWeights& out)
{
    const ML::Util util(ML::TensorDataType::FLOAT16, ML::TensorLayout::DEFAULT);
    for (int l = 0; l < kNumConvLayers; ++l)
    {
        std::vector<uint8_t> f, b;
        const ConvLayerDesc& d = kConvLayers[l];
        if (!util.CreateWeightTensors(weights, d.conv, d.scale, d.shift, d.sizes, f, b)) return false;
        out.filter[l].resize(f.size() / 2);
        std::memcpy(out.filter[l].data(), f.data(), f.size());
        out.bias[l].resize(b.size() / 2);
        if (!b.empty()) std::memcpy(out.bias[l].data(), b.data(), b.size());
    }
    return true;
}

// ---------------------------------------------------------------------------
// MFMA fragment packing
// ---------------------------------------------------------------------------
std::vector<uint16_t> PackConv3x3(const std::vector<uint16_t>& oihw, int cout, int cin)
{
    const int tiles = (cout + 31) / 32;
    const int cchunks = cin / 16;
    const int ksteps = 9 * cchunks;
    std::vector<uint16_t> out(size_t(tiles) * ksteps * 64 * 8, 0);
    for (int t = 0; t < tiles; ++t)
        for (int tap = 0; tap < 9; ++tap)
            for (int cc = 0; cc < cchunks; ++cc)
                for (int lane = 0; lane < 64; ++lane)
                {
                    const int o = t * 32 + (lane & 31);
                    if (o >= cout) continue;
                    const int h = lane >> 5;
                    const size_t dst = ((size_t(t) * ksteps + tap * cchunks + cc) * 64 + lane) * 8;
                    for (int j = 0; j < 8; ++j)
                    {
                        const int c = cc * 16 + 8 * h + j;
                        out[dst + j] = oihw[(size_t(o) * cin + c) * 9 + tap];
                    }
                }
    return out;
}

std::vector<uint16_t> PackConv3x3Rows16(const std::vector<uint16_t>& oihw, int cout, int cin)
{
    std::vector<uint16_t> out(size_t(9) * 64 * 8, 0);
    for (int tap = 0; tap < 9; ++tap)
        for (int lane = 0; lane < 64; ++lane)
        {
            const int o = lane & 15, q = lane >> 4;
            if (o >= cout) continue;
            for (int j = 0; j < 8; ++j)
                out[(size_t(tap) * 64 + lane) * 8 + j] = oihw[(size_t(o) * cin + 8 * q + j) * 9 + tap];
        }
    return out;
}

std::vector<uint16_t> PackConv6Rotating(const std::vector<uint16_t>& oihw)
{
    // [rho][j][lane][8]: A row 4 * g + c (lane & 15), k = 8 * (lane >> 4) + jj
    std::vector<uint16_t> out(size_t(4) * 3 * 64 * 8, 0);
    for (int rho = 0; rho < 4; ++rho)
        for (int j = 0; j < 3; ++j)
            for (int lane = 0; lane < 64; ++lane)
            {
                const int row = lane & 15, q = lane >> 4;
                const int g = row >> 2, c = row & 3;
                if (c >= 3) continue;
                // the filter row i that input row R (R mod 4 == rho) contributes to output row group g
                int i = -1;
                if (g == ((rho + 3) & 3)) i = 2;        // output row R - 1
                else if (g == rho) i = 1;               // output row R
                else if (g == ((rho + 1) & 3)) i = 0;   // output row R + 1
                if (i < 0) continue;                    // group of row R + 2: not touched by this input row
                for (int jj = 0; jj < 8; ++jj)
                    out[((size_t(rho) * 3 + j) * 64 + lane) * 8 + jj] = oihw[(size_t(c) * 32 + 8 * q + jj) * 9 + i * 3 + j];
            }
    return out;
}

std::vector<uint16_t> PackConv1(const std::vector<uint16_t>& oihw)
{
    constexpr int ksteps = 7;
    std::vector<uint16_t> out(size_t(ksteps) * 64 * 8, 0);
    for (int s = 0; s < ksteps; ++s)
        for (int lane = 0; lane < 64; ++lane)
        {
            const int o = lane & 31, h = lane >> 5;
            for (int j = 0; j < 8; ++j)
            {
                const int tap = 4 * s + 2 * h + (j >> 2);
                const int c = j & 3;
                if (tap < 25 && c < 3) out[(size_t(s) * 64 + lane) * 8 + j] = oihw[(size_t(o) * 3 + c) * 25 + tap];
            }
        }
    return out;
}

// Low-resolution tap (0..2) that 5x5 tap i reads for output phase p:
// floor((p + i - 2) / 2) + 1.
static int PhaseTap(int p, int i) { return ((p + i + 2) >> 1) - 1; }

std::vector<uint16_t> ConvUp1PhaseFilters(const std::vector<uint16_t>& oihw)
{
    constexpr int O = 32, C = 64;
    std::vector<uint16_t> out(size_t(4) * O * C * 9, 0);
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px)
            for (int o = 0; o < O; ++o)
                for (int c = 0; c < C; ++c)
                {
                    double sum[9] = {};
                    for (int i = 0; i < 5; ++i)
                        for (int j = 0; j < 5; ++j)
                            sum[PhaseTap(py, i) * 3 + PhaseTap(px, j)] +=
                                static_cast<double>(ML::Float16ToFloat32(oihw[((size_t(o) * C + c) * 5 + i) * 5 + j]));
                    for (int t = 0; t < 9; ++t)
                        out[(((size_t(py) * 2 + px) * O + o) * C + c) * 9 + t] =
                            ML::Float32ToFloat16(static_cast<float>(sum[t]));
                }
    return out;
}

std::vector<uint16_t> PackConv3x3K32(const std::vector<uint16_t>& oihw, int cout, int cin, bool interleave)
{
    const int tiles = (cout + 31) / 32, kchunks = cin / 32, ksteps = 9 * kchunks;
    std::vector<uint16_t> out(size_t(tiles) * ksteps * 2 * 64 * 8, 0);
    for (int t = 0; t < tiles; ++t)
        for (int tap = 0; tap < 9; ++tap)
            for (int kc = 0; kc < kchunks; ++kc)
                for (int half = 0; half < 2; ++half)
                    for (int lane = 0; lane < 64; ++lane)
                    {
                        const int r = lane & 15, q = lane >> 4;
                        // output of MFMA row r of this half: plain, or interleaved so that the four rows a lane
                        // group gets of either half are eight consecutive outputs
                        const int o = t * 32 + (interleave ? 8 * (r >> 2) + 4 * half + (r & 3) : half * 16 + r);
                        if (o >= cout) continue;
                        const size_t dst = (((size_t(t) * ksteps + tap * kchunks + kc) * 2 + half) * 64 + lane) * 8;
                        for (int j = 0; j < 8; ++j) out[dst + j] = oihw[(size_t(o) * cin + kc * 32 + 8 * q + j) * 9 + tap];
                    }
    return out;
}

std::vector<uint16_t> ConvUp1ColumnFilters(const std::vector<uint16_t>& oihw)
{
    constexpr int O = 32, C = 64;
    std::vector<uint16_t> out(size_t(2) * O * C * 5 * 3, 0);
    for (int px = 0; px < 2; ++px)
        for (int o = 0; o < O; ++o)
            for (int c = 0; c < C; ++c)
                for (int i = 0; i < 5; ++i)
                {
                    double sum[3] = {};
                    for (int j = 0; j < 5; ++j)
                        sum[PhaseTap(px, j)] += static_cast<double>(ML::Float16ToFloat32(oihw[((size_t(o) * C + c) * 5 + i) * 5 + j]));
                    for (int t = 0; t < 3; ++t)
                        out[((((size_t(px) * O + o) * C + c) * 5 + i) * 3) + t] = ML::Float32ToFloat16(static_cast<float>(sum[t]));
                }
    return out;
}

std::vector<uint16_t> PackConvUp1RowSharedK32(const std::vector<uint16_t>& oihw)
{
    constexpr int O = 32, C = 64, kFrags = 30;
    const std::vector<uint16_t> cols = ConvUp1ColumnFilters(oihw);
    std::vector<uint16_t> out(size_t(4) * kFrags * 64 * 8, 0);
    for (int wave = 0; wave < 4; ++wave)
    {
        const int px = wave & 1, oh = wave >> 1;
        for (int f = 0; f < kFrags; ++f)
        {
            // S fragments 0..17: row taps 0, 2, 4 (input rows y-1, y, y+1); T fragments 18..29: row taps 1, 3
            // (input rows y, y+1 as seen from output row 2y+1)
            const int g = f < 18 ? f : f - 18;
            const int rowSlot = g / 6, j = (g % 6) / 2, kc = g % 2;
            const int i = f < 18 ? 2 * rowSlot : 2 * rowSlot + 1;
            for (int lane = 0; lane < 64; ++lane)
            {
                const int o = oh * 16 + (lane & 15), q = lane >> 4;
                for (int jj = 0; jj < 8; ++jj)
                {
                    const int c = kc * 32 + 8 * q + jj;
                    out[((size_t(wave) * kFrags + f) * 64 + lane) * 8 + jj] = cols[((((size_t(px) * O + o) * C + c) * 5 + i) * 3) + j];
                }
            }
        }
    }
    return out;
}

std::vector<uint16_t> PackConvUp1Phases(const std::vector<uint16_t>& oihw)
{
    const std::vector<uint16_t> phases = ConvUp1PhaseFilters(oihw);
    std::vector<uint16_t> out;
    for (int p = 0; p < 4; ++p)
    {
        const std::vector<uint16_t> one(phases.begin() + size_t(p) * 32 * 64 * 9,
                                        phases.begin() + size_t(p + 1) * 32 * 64 * 9);
        const std::vector<uint16_t> packed = PackConv3x3(one, 32, 64);
        out.insert(out.end(), packed.begin(), packed.end());
    }
    return out;
}

}  // namespace srx
