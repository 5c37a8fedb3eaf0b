// gfx950 kernels of the generic layers: the operators the reference's layer factories create
// (ML::Util::CreateUpsampleLayer / CreateConvolutionLayer / CreateAdditionLayer,
// XUSG/MachineLearning/XUSGMachineLearning.h:847-854), for any tensor size, filter size, data type
// (fp16 or fp32 storage, fp32 arithmetic) and layout (NCHW or NHWC), plus the two shader passes
// (CSImageToTensor.hlsl, PSTensorToImage.hlsl) for those data types and layouts.
//
// The convolution of fp16 tensors is an LDS-tiled implicit GEMM on the matrix cores
// (ConvMfmaKernel); fp32 tensors use a direct, LDS-tiled VALU kernel.  They serve the fp32 tensor
// mode, the NHWC switch and other small networks.  The fp16 fast path of the sample network does not
// use them (it runs on the fused MFMA kernels).
#include "srx_device.h"
#include "srx_weights.h"

#include <cstdlib>

namespace srx {

using namespace dev;

namespace {

constexpr int kTW = 32, kTH = 8;      // output pixels per workgroup tile
constexpr int kOT = 16;               // output channels per workgroup
constexpr int kCC = 8;                // input channels staged per pass
constexpr int kKMax = 7;              // largest filter side

__device__ __forceinline__ float loadF(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float loadF(const f16* p, size_t i) { return (float)p[i]; }
__device__ __forceinline__ void storeF(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void storeF(f16* p, size_t i, float v) { p[i] = (f16)v; }   // round to nearest even

// Element (n, c, y, x) of a dense {N, C, H, W} tensor (ML::Util::GetStrides, XUSGMachineLearning.h:860).
template <bool NHWC>
__device__ __forceinline__ size_t at(int n, int c, int y, int x, int C, int H, int W)
{
    return NHWC ? ((size_t(n) * H + y) * W + x) * C + c : ((size_t(n) * C + c) * H + y) * W + x;
}

struct ConvArgs
{
    const void *in, *filter, *bias;
    void* out;
    int N, C, H, W, O, kh, kw, padTop, padLeft, relu;
    int tilesX, tilesY;
};

// out[n,o,y,x] = act(bias[o] + sum_{c,i,j} in[n,c,y+i-padTop,x+j-padLeft] * w[o,c,i,j]), zero padding.
// The filter is OIHW for NCHW tensors and O,kh,kw,I for NHWC tensors (as CreateWeightTensors lays it out).
template <typename T, bool NHWC>
__global__ __launch_bounds__(256) void ConvDirectKernel(const ConvArgs a)
{
    __shared__ float sIn[kCC][kTH + kKMax - 1][kTW + kKMax];
    __shared__ float sW[kCC][kKMax * kKMax][kOT];
    const T* in = static_cast<const T*>(a.in);
    const T* filter = static_cast<const T*>(a.filter);
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int tile = blockIdx.x, n = blockIdx.z, o0 = blockIdx.y * kOT;
    const int x0 = (tile % a.tilesX) * kTW, y0 = (tile / a.tilesX) * kTH;
    const int inH = kTH + a.kh - 1, inW = kTW + a.kw - 1, taps = a.kh * a.kw;

    float acc[kOT];
#pragma unroll
    for (int o = 0; o < kOT; ++o) acc[o] = 0.0f;

    for (int c0 = 0; c0 < a.C; c0 += kCC)
    {
        const int cn = min(kCC, a.C - c0);
        __syncthreads();
        for (int i = tid; i < cn * inH * inW; i += 256)
        {
            const int c = i / (inH * inW), r = i - c * inH * inW, yy = r / inW, xx = r - yy * inW;
            const int y = y0 + yy - a.padTop, x = x0 + xx - a.padLeft;
            sIn[c][yy][xx] = (y >= 0 && y < a.H && x >= 0 && x < a.W) ? loadF(in, at<NHWC>(n, c0 + c, y, x, a.C, a.H, a.W)) : 0.0f;
        }
        for (int i = tid; i < cn * taps * kOT; i += 256)
        {
            const int o = i % kOT, r = i / kOT, t = r % taps, c = r / taps;
            float v = 0.0f;
            if (o0 + o < a.O)
            {
                const int ki = t / a.kw, kj = t - ki * a.kw;
                // the filter seen as a tensor {O, I, kh, kw} in the operator's layout
                v = loadF(filter, at<NHWC>(o0 + o, c0 + c, ki, kj, a.C, a.kh, a.kw));
            }
            sW[c][t][o] = v;
        }
        __syncthreads();
        for (int c = 0; c < cn; ++c)
            for (int ki = 0; ki < a.kh; ++ki)
                for (int kj = 0; kj < a.kw; ++kj)
                {
                    const float v = sIn[c][ty + ki][tx + kj];
                    const float4* w4 = reinterpret_cast<const float4*>(&sW[c][ki * a.kw + kj][0]);
#pragma unroll
                    for (int g = 0; g < kOT / 4; ++g)
                    {
                        const float4 w = w4[g];
                        acc[4 * g + 0] = fmaf(v, w.x, acc[4 * g + 0]);
                        acc[4 * g + 1] = fmaf(v, w.y, acc[4 * g + 1]);
                        acc[4 * g + 2] = fmaf(v, w.z, acc[4 * g + 2]);
                        acc[4 * g + 3] = fmaf(v, w.w, acc[4 * g + 3]);
                    }
                }
    }

    const int x = x0 + tx, y = y0 + ty;
    if (x >= a.W || y >= a.H) return;
    T* out = static_cast<T*>(a.out);
#pragma unroll
    for (int o = 0; o < kOT; ++o)
    {
        if (o0 + o >= a.O) break;
        float v = acc[o];
        if (a.bias) v += loadF(static_cast<const T*>(a.bias), size_t(o0 + o));
        if (a.relu) v = fmaxf(v, 0.0f);
        storeF(out, at<NHWC>(n, o0 + o, y, x, a.O, a.H, a.W), v);
    }
}

// The same convolution for fp16 tensors on the matrix cores: implicit GEMM with
// v_mfma_f32_32x32x16_f16, A = filter (32 output channels x 16 input channels of one tap),
// B = activations (16 input channels x 32 consecutive pixels of a row).  A workgroup (4 waves) makes
// 32 x 8 pixels x 32 output channels; wave w makes rows 2w and 2w+1.  Per pass of 16 input channels
// the input tile with its halo and the filter slice are staged in LDS as [pixel][16 channels] and
// [tap][output][16 channels], 48 bytes apart (an odd number of 16-byte slots: conflict-free
// ds_read_b128 fragments).  fp32 accumulation; bias, ReLU and the rounding to fp16 as in the direct kernel.
constexpr int kMO = 32;               // output channels per workgroup
constexpr int kMC = 16;               // input channels per pass (one MFMA K-step)
constexpr int kSlot = 48;             // bytes per pixel / per filter row in LDS

__host__ __device__ inline int convMfmaInBytes(int kh, int kw) { return (kTH + kh - 1) * (kTW + kw - 1) * kSlot; }
__host__ __device__ inline int convMfmaLdsBytes(int kh, int kw) { return convMfmaInBytes(kh, kw) + kh * kw * kMO * kSlot; }

template <bool NHWC>
__global__ __launch_bounds__(256) void ConvMfmaKernel(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const f16* in = static_cast<const f16*>(a.in);
    const f16* filter = static_cast<const f16*>(a.filter);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x, n = blockIdx.z, o0 = blockIdx.y * kMO;
    const int x0 = (tile % a.tilesX) * kTW, y0 = (tile / a.tilesX) * kTH;
    const int inH = kTH + a.kh - 1, inW = kTW + a.kw - 1, taps = a.kh * a.kw;
    char* sIn = smem;
    char* sW = smem + convMfmaInBytes(a.kh, a.kw);
    const float invInW = 1.0f / float(inW), invTaps = 1.0f / float(taps);

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;

    for (int c0 = 0; c0 < a.C; c0 += kMC)
    {
        const int cn = min(kMC, a.C - c0);
        __syncthreads();
        // Staging.  Channels beyond the tensor's and pixels outside the image are zeros.  Small
        // quotients are taken with a float reciprocal (exact for these ranges, all below 1024).
        if (NHWC && (a.C & 7) == 0)
        {
            // 8 channels = 16 contiguous, aligned bytes per pixel and per filter tap
            for (int i = tid; i < inH * inW * 2; i += 256)
            {
                const int r = i >> 1, hh = i & 1;
                const int yy = int((r + 0.5f) * invInW), xx = r - yy * inW;
                const int y = y0 + yy - a.padTop, x = x0 + xx - a.padLeft;
                uint4 v = uint4{0, 0, 0, 0};
                if (c0 + 8 * hh < a.C && y >= 0 && y < a.H && x >= 0 && x < a.W)
                    v = *reinterpret_cast<const uint4*>(in + at<true>(n, c0 + 8 * hh, y, x, a.C, a.H, a.W));
                *reinterpret_cast<uint4*>(sIn + r * kSlot + hh * 16) = v;
            }
            for (int i = tid; i < kMO * taps * 2; i += 256)
            {
                const int r = i >> 1, hh = i & 1;
                const int o = int((r + 0.5f) * invTaps), t = r - o * taps;
                uint4 v = uint4{0, 0, 0, 0};
                if (o0 + o < a.O && c0 + 8 * hh < a.C)
                    v = *reinterpret_cast<const uint4*>(filter + (size_t(o0 + o) * taps + t) * a.C + c0 + 8 * hh);
                *reinterpret_cast<uint4*>(sW + (t * kMO + o) * kSlot + hh * 16) = v;
            }
        }
        else if (NHWC)
        {
            for (int i = tid; i < kMC * inH * inW; i += 256)
            {
                const int c = i & (kMC - 1), r = i >> 4;
                const int yy = int((r + 0.5f) * invInW), xx = r - yy * inW;
                const int y = y0 + yy - a.padTop, x = x0 + xx - a.padLeft;
                f16 v = f16(0.0f);
                if (c < cn && y >= 0 && y < a.H && x >= 0 && x < a.W) v = in[at<true>(n, c0 + c, y, x, a.C, a.H, a.W)];
                *reinterpret_cast<f16*>(sIn + r * kSlot + c * 2) = v;
            }
            for (int i = tid; i < kMO * kMC * taps; i += 256)
            {
                const int c = i & (kMC - 1), r = i >> 4;
                const int o = int((r + 0.5f) * invTaps), t = r - o * taps;
                f16 v = f16(0.0f);
                if (o0 + o < a.O && c < cn) v = filter[(size_t(o0 + o) * taps + t) * a.C + c0 + c];
                *reinterpret_cast<f16*>(sW + (t * kMO + o) * kSlot + c * 2) = v;
            }
        }
        else
        {
            // NCHW / OIHW: the 8 channels of a fragment half lie in 8 planes; a thread gathers them
            // (lanes run along x, so each of the 8 loads is coalesced) and writes 16 bytes at once
            for (int i = tid; i < inH * inW * 2; i += 256)
            {
                const int r = i % (inH * inW), hh = i / (inH * inW);
                const int yy = int((r + 0.5f) * invInW), xx = r - yy * inW;
                const int y = y0 + yy - a.padTop, x = x0 + xx - a.padLeft;
                f16x8 v;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = f16(0.0f);
                if (y >= 0 && y < a.H && x >= 0 && x < a.W)
                {
                    const f16* src = in + at<false>(n, c0 + 8 * hh, y, x, a.C, a.H, a.W);
                    const size_t plane = size_t(a.H) * a.W;
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (8 * hh + k < cn) v[k] = src[k * plane];
                }
                *reinterpret_cast<f16x8*>(sIn + r * kSlot + hh * 16) = v;
            }
            for (int i = tid; i < kMO * taps * 2; i += 256)
            {
                const int r = i % (kMO * taps), hh = i / (kMO * taps);
                const int o = int((r + 0.5f) * invTaps), t = r - o * taps;
                f16x8 v;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = f16(0.0f);
                if (o0 + o < a.O)
                {
                    const f16* src = filter + (size_t(o0 + o) * a.C + c0 + 8 * hh) * taps + t;
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (8 * hh + k < cn) v[k] = src[k * taps];
                }
                *reinterpret_cast<f16x8*>(sW + (t * kMO + o) * kSlot + hh * 16) = v;
            }
        }
        __syncthreads();
        const char* aBase = sW + px * kSlot + h * 16;
        const char* bBase = sIn + (2 * wave * inW + px) * kSlot + h * 16;
        for (int ki = 0; ki < a.kh; ++ki)
            for (int kj = 0; kj < a.kw; ++kj)
            {
                const f16x8 w = *reinterpret_cast<const f16x8*>(aBase + (ki * a.kw + kj) * kMO * kSlot);
                const f16x8 b0 = *reinterpret_cast<const f16x8*>(bBase + (ki * inW + kj) * kSlot);
                const f16x8 b1 = *reinterpret_cast<const f16x8*>(bBase + ((ki + 1) * inW + kj) * kSlot);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, b1, acc1, 0, 0, 0);
            }
    }

    // D: lane (px, h) holds, for its pixel, output channels (r & 3) + 8 * (r >> 2) + 4 * h
    const int x = x0 + px;
    if (x >= a.W) return;
    f16* out = static_cast<f16*>(a.out);
    const f16* bias = static_cast<const f16*>(a.bias);
#pragma unroll
    for (int row = 0; row < 2; ++row)
    {
        const int y = y0 + 2 * wave + row;
        if (y >= a.H) break;
        if (NHWC && (a.O & 3) == 0)
        {
            // four consecutive output channels of a pixel are 8 contiguous, aligned bytes
#pragma unroll
            for (int g = 0; g < 4; ++g)
            {
                const int o = o0 + 8 * g + 4 * h;
                if (o >= a.O) continue;
                f16x4 v4;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                {
                    float v = row ? acc1[4 * g + k] : acc0[4 * g + k];
                    if (bias) v += (float)bias[o + k];
                    if (a.relu) v = fmaxf(v, 0.0f);
                    v4[k] = (f16)v;
                }
                *reinterpret_cast<f16x4*>(out + at<true>(n, o, y, x, a.O, a.H, a.W)) = v4;
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
            const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (o >= a.O) continue;
            float v = row ? acc1[r] : acc0[r];
            if (bias) v += (float)bias[o];
            if (a.relu) v = fmaxf(v, 0.0f);
            out[at<NHWC>(n, o, y, x, a.O, a.H, a.W)] = (f16)v;
        }
    }
}

struct UpsampleArgs
{
    const void* in;
    void* out;
    int N, C, H, W, sx, sy, linear;
    size_t total;
};

// DML UPSAMPLE_2D.  NEAREST_NEIGHBOR with an integer scale replicates pixels: out[y][x] = in[y / sy][x / sx].
// LINEAR samples with pixel centres at half-integers and clamps at the edges.
template <typename T, bool NHWC>
__global__ void UpsampleKernel(const UpsampleArgs a)
{
    const int OH = a.H * a.sy, OW = a.W * a.sx;
    const T* in = static_cast<const T*>(a.in);
    T* out = static_cast<T*>(a.out);
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < a.total; i += size_t(gridDim.x) * blockDim.x)
    {
        int n, c, y, x;
        size_t r = i;
        if (NHWC) { c = int(r % a.C); r /= a.C; x = int(r % OW); r /= OW; y = int(r % OH); n = int(r / OH); }
        else { x = int(r % OW); r /= OW; y = int(r % OH); r /= OH; c = int(r % a.C); n = int(r / a.C); }
        float v;
        if (!a.linear) v = loadF(in, at<NHWC>(n, c, y / a.sy, x / a.sx, a.C, a.H, a.W));
        else
        {
            const float fy = fminf(fmaxf((y + 0.5f) / a.sy - 0.5f, 0.0f), float(a.H - 1));
            const float fx = fminf(fmaxf((x + 0.5f) / a.sx - 0.5f, 0.0f), float(a.W - 1));
            const int y0 = int(fy), x0 = int(fx), y1 = min(y0 + 1, a.H - 1), x1 = min(x0 + 1, a.W - 1);
            const float wy = fy - y0, wx = fx - x0;
            const float v00 = loadF(in, at<NHWC>(n, c, y0, x0, a.C, a.H, a.W)), v01 = loadF(in, at<NHWC>(n, c, y0, x1, a.C, a.H, a.W));
            const float v10 = loadF(in, at<NHWC>(n, c, y1, x0, a.C, a.H, a.W)), v11 = loadF(in, at<NHWC>(n, c, y1, x1, a.C, a.H, a.W));
            v = (v00 * (1.0f - wx) + v01 * wx) * (1.0f - wy) + (v10 * (1.0f - wx) + v11 * wx) * wy;
        }
        storeF(out, i, v);
    }
}

// DML ELEMENT_WISE_ADD; out may alias a or b (the reference binds it in place, Content/SuperResolution.cpp:700-702).
template <typename T>
__global__ void AddKernel(const T* a, const T* b, T* out, size_t total)
{
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += size_t(gridDim.x) * blockDim.x)
        storeF(out, i, loadF(a, i) + loadF(b, i));
}

// CSImageToTensor.hlsl:15-45: texel.rgb as UNORM floats into a {N, 3, H, W} tensor (plane = H * W).
template <typename T, bool NHWC>
__global__ void ImageToTensorKernel(const uint8_t* img, long pitch, long frame, T* out, int N, int H, int W)
{
    const size_t total = size_t(N) * H * W;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += size_t(gridDim.x) * blockDim.x)
    {
        const int x = int(i % W), y = int((i / W) % H), n = int(i / (size_t(W) * H));
        const unsigned v = *reinterpret_cast<const unsigned*>(img + n * frame + y * pitch + x * 4);
#pragma unroll
        for (int c = 0; c < 3; ++c) storeF(out, at<NHWC>(n, c, y, x, 3, H, W), float((v >> (8 * c)) & 0xff) / 255.0f);
    }
}

// PSTensorToImage.hlsl:14-35 + the R8G8B8A8_UNORM render target.
template <typename T, bool NHWC>
__global__ void TensorToImageKernel(const T* in, uint8_t* img, long pitch, long frame, int N, int H, int W)
{
    const size_t total = size_t(N) * H * W;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += size_t(gridDim.x) * blockDim.x)
    {
        const int x = int(i % W), y = int((i / W) % H), n = int(i / (size_t(W) * H));
        unsigned rgba = 0xff000000u;
#pragma unroll
        for (int c = 0; c < 3; ++c) rgba |= unorm8(loadF(in, at<NHWC>(n, c, y, x, 3, H, W))) << (8 * c);
        *reinterpret_cast<unsigned*>(img + n * frame + y * pitch + x * 4) = rgba;
    }
}

int gridForElements(size_t total)
{
    const size_t blocks = (total + 255) / 256;
    return int(blocks < 1 ? 1 : (blocks > 65536 ? 65536 : blocks));
}

template <typename F16, typename F32>
hipError_t byType(ML::TensorDataType t, F16&& f16, F32&& f32)
{
    if (t == ML::TensorDataType::FLOAT16) f16();
    else if (t == ML::TensorDataType::FLOAT32) f32();
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace

hipError_t DispatchOperator(const ML::Operator& op, const void* input, const void* filterOrB, const void* bias,
                            void* output, hipStream_t s)
{
    using K = ML::Operator::Kind;
    const bool nhwc = op.layout == ML::TensorLayout::NHWC;
    if (!input || !output) return hipErrorInvalidValue;
    if (op.kind == K::CONVOLUTION)
    {
        if (!filterOrB || op.filterSizes[2] > kKMax || op.filterSizes[3] > kKMax) return hipErrorInvalidValue;
        ConvArgs a{};
        a.in = input;
        a.filter = filterOrB;
        a.bias = op.useBiasAndActivation ? bias : nullptr;
        if (op.useBiasAndActivation && !bias) return hipErrorInvalidValue;
        a.out = output;
        a.N = int(op.inputSizes[0]); a.C = int(op.inputSizes[1]); a.H = int(op.inputSizes[2]); a.W = int(op.inputSizes[3]);
        a.O = int(op.filterSizes[0]); a.kh = int(op.filterSizes[2]); a.kw = int(op.filterSizes[3]);
        a.padTop = int(op.padTop); a.padLeft = int(op.padLeft);
        a.relu = op.useBiasAndActivation ? 1 : 0;
        a.tilesX = (a.W + kTW - 1) / kTW;
        a.tilesY = (a.H + kTH - 1) / kTH;
        if (a.N > 65535 || (a.O + kOT - 1) / kOT > 65535) return hipErrorInvalidValue;
        const dim3 grid(a.tilesX * a.tilesY, (a.O + kOT - 1) / kOT, a.N), block(256);
        static const bool forceValu = getenv("SRX_GENERIC_VALU") != nullptr;      // read once per process
        if (op.dataType == ML::TensorDataType::FLOAT16 && !forceValu)
        {
            // fp16 tensors run on the matrix cores (SRX_GENERIC_VALU keeps the direct kernel, for comparison)
            const int lds = convMfmaLdsBytes(a.kh, a.kw);
            const int most = convMfmaLdsBytes(kKMax, kKMax);
            hipError_t e = dev::ensureDynamicLds(reinterpret_cast<const void*>(ConvMfmaKernel<true>), most);
            if (e == hipSuccess) e = dev::ensureDynamicLds(reinterpret_cast<const void*>(ConvMfmaKernel<false>), most);
            if (e != hipSuccess) return e;
            const dim3 mgrid(a.tilesX * a.tilesY, (a.O + kMO - 1) / kMO, a.N);
            if (nhwc) hipLaunchKernelGGL((ConvMfmaKernel<true>), mgrid, block, lds, s, a);
            else hipLaunchKernelGGL((ConvMfmaKernel<false>), mgrid, block, lds, s, a);
            return hipGetLastError();
        }
        return byType(op.dataType,
                      [&] { if (nhwc) hipLaunchKernelGGL((ConvDirectKernel<f16, true>), grid, block, 0, s, a);
                            else hipLaunchKernelGGL((ConvDirectKernel<f16, false>), grid, block, 0, s, a); },
                      [&] { if (nhwc) hipLaunchKernelGGL((ConvDirectKernel<float, true>), grid, block, 0, s, a);
                            else hipLaunchKernelGGL((ConvDirectKernel<float, false>), grid, block, 0, s, a); });
    }
    if (op.kind == K::UPSAMPLE_2D)
    {
        UpsampleArgs a{};
        a.in = input;
        a.out = output;
        a.N = int(op.inputSizes[0]); a.C = int(op.inputSizes[1]); a.H = int(op.inputSizes[2]); a.W = int(op.inputSizes[3]);
        a.sx = int(op.scaleX); a.sy = int(op.scaleY);
        a.linear = op.interpolation == ML::InterpolationType::LINEAR;
        a.total = size_t(a.N) * a.C * a.H * a.sy * a.W * a.sx;
        const dim3 grid(gridForElements(a.total)), block(256);
        return byType(op.dataType,
                      [&] { if (nhwc) hipLaunchKernelGGL((UpsampleKernel<f16, true>), grid, block, 0, s, a);
                            else hipLaunchKernelGGL((UpsampleKernel<f16, false>), grid, block, 0, s, a); },
                      [&] { if (nhwc) hipLaunchKernelGGL((UpsampleKernel<float, true>), grid, block, 0, s, a);
                            else hipLaunchKernelGGL((UpsampleKernel<float, false>), grid, block, 0, s, a); });
    }
    if (op.kind == K::ELEMENT_WISE_ADD)
    {
        if (!filterOrB) return hipErrorInvalidValue;
        const size_t total = size_t(op.inputSizes[0]) * op.inputSizes[1] * op.inputSizes[2] * op.inputSizes[3];
        const dim3 grid(gridForElements(total)), block(256);
        return byType(op.dataType,
                      [&] { hipLaunchKernelGGL(AddKernel<f16>, grid, block, 0, s, static_cast<const f16*>(input),
                                               static_cast<const f16*>(filterOrB), static_cast<f16*>(output), total); },
                      [&] { hipLaunchKernelGGL(AddKernel<float>, grid, block, 0, s, static_cast<const float*>(input),
                                               static_cast<const float*>(filterOrB), static_cast<float*>(output), total); });
    }
    return hipErrorInvalidValue;
}

hipError_t LaunchImageToTensor(const Image& src, void* tensor, ML::TensorDataType t, ML::TensorLayout layout,
                               int width, int height, int batch, hipStream_t s)
{
    const bool nhwc = layout == ML::TensorLayout::NHWC;
    const dim3 grid(gridForElements(size_t(batch) * width * height)), block(256);
    const long pitch = long(src.pitch), frame = long(src.frame);
    return byType(t,
                  [&] { if (nhwc) hipLaunchKernelGGL((ImageToTensorKernel<f16, true>), grid, block, 0, s, src.base, pitch, frame, static_cast<f16*>(tensor), batch, height, width);
                        else hipLaunchKernelGGL((ImageToTensorKernel<f16, false>), grid, block, 0, s, src.base, pitch, frame, static_cast<f16*>(tensor), batch, height, width); },
                  [&] { if (nhwc) hipLaunchKernelGGL((ImageToTensorKernel<float, true>), grid, block, 0, s, src.base, pitch, frame, static_cast<float*>(tensor), batch, height, width);
                        else hipLaunchKernelGGL((ImageToTensorKernel<float, false>), grid, block, 0, s, src.base, pitch, frame, static_cast<float*>(tensor), batch, height, width); });
}

hipError_t LaunchTensorToImage(const void* tensor, const Image& dst, ML::TensorDataType t, ML::TensorLayout layout,
                               int width, int height, int batch, hipStream_t s)
{
    const bool nhwc = layout == ML::TensorLayout::NHWC;
    const dim3 grid(gridForElements(size_t(batch) * width * height)), block(256);
    const long pitch = long(dst.pitch), frame = long(dst.frame);
    return byType(t,
                  [&] { if (nhwc) hipLaunchKernelGGL((TensorToImageKernel<f16, true>), grid, block, 0, s, static_cast<const f16*>(tensor), dst.base, pitch, frame, batch, height, width);
                        else hipLaunchKernelGGL((TensorToImageKernel<f16, false>), grid, block, 0, s, static_cast<const f16*>(tensor), dst.base, pitch, frame, batch, height, width); },
                  [&] { if (nhwc) hipLaunchKernelGGL((TensorToImageKernel<float, true>), grid, block, 0, s, static_cast<const float*>(tensor), dst.base, pitch, frame, batch, height, width);
                        else hipLaunchKernelGGL((TensorToImageKernel<float, false>), grid, block, 0, s, static_cast<const float*>(tensor), dst.base, pitch, frame, batch, height, width); });
}

}  // namespace srx
