// gfx950 (MI355X, CDNA4) kernels of the super-resolution path: the layer-wise kernels (one launch per
// layer, every tensor in HBM; the per-layer parity path), the workspace-clearing kernel and the in-kernel
// clock probe of diagnostic builds.  The product path runs on the two fused kernels in srx_lr_fused.hip and
// srx_hr_fused.hip.
//
// Every convolution here is an implicit GEMM on v_mfma_f32_32x32x16_f16 with
//   A (32 x 16)  = filter fragment, rows = output channels      -> lives in VGPRs
//   B (16 x 32)  = activations, columns = 32 consecutive pixels  -> read from LDS
//   D (32 x 32)  = fp32 accumulators, rows = channels, columns = pixels
// A wave keeps ALL K-steps of its 32 output channels in registers for the whole
// kernel (72 or 144 VGPRs) and walks tiles persistently, so the only operand
// traffic per MFMA is one 1 KiB ds_read_b128 of activations.  Activation tiles
// are staged NHWC into LDS with a pixel stride that is an odd number of 16-byte
// slots, which makes the 32-pixel fragment reads bank-conflict free.
//
// Replaces the DirectML operators the reference dispatches in
// SuperResolution::Process (Content/SuperResolution.cpp:73-103) and the
// CSImageToTensor / PSTensorToImage shaders (Content/Shaders/*.hlsl).
#include "srx_device.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

namespace srx {

using namespace dev;

namespace {

constexpr int TW = kTileW;
constexpr int TH = kTileH;
struct ConvParams
{
    TensorD in, out;
    const f16x8* wpk;
    const float* bias;
    int width, height;
    int tilesX, tilesY, numTiles;
    // final layer only
    TensorD x0, residual, sum;
    uint8_t* image;
    long imagePitch, imageFrame;
};

// ---------------------------------------------------------------------------
// 3x3 convolutions
// ---------------------------------------------------------------------------
enum { MODE_PLAIN = 0, MODE_UP1 = 1, MODE_RGB = 2 };

// CIN: input channels (32 or 64).  NT: 32-row output tiles, one per wave group
// (MODE_UP1: the four output phases).  Each wave owns tile wave % NT and the
// rows wave / NT, wave / NT + 4 / NT, ... of the 8-row workgroup tile.
template <int CIN, int NT, int MODE>
__global__ __launch_bounds__(kThreads, 2) void Conv3x3Kernel(const ConvParams p)
{
    constexpr int PIXB = CIN * 2;              // bytes per pixel in HBM
    constexpr int STRIDE = PIXB + 16;          // LDS pixel stride: odd number of 16-B slots
    constexpr int LW = TW + 2, LH = TH + 2;    // staged tile with its 1-pixel halo
    constexpr int CCH = CIN / 16;
    constexpr int KSTEPS = 9 * CCH;
    constexpr int CPP = PIXB / 16;             // 16-B chunks per pixel
    constexpr int NCHUNK = LH * LW * CPP;
    constexpr int RG = 4 / NT;                 // waves sharing one output tile
    constexpr int COUT = (MODE == MODE_PLAIN) ? NT * 32 : 32;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nt = wave % NT;
    const int rg = wave / NT;
    const int px = lane & 31;
    const int h = lane >> 5;

    // this wave's filter fragments: resident for the whole kernel
    f16x8 wreg[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) wreg[s] = p.wpk[(nt * KSTEPS + s) * 64 + lane];

    float bias[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
        bias[q] = (MODE == MODE_RGB) ? 0.0f : p.bias[nt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h];

    const int tilesPerFrame = p.tilesX * p.tilesY;
    for (TileWalk tw(p.numTiles); tw.valid(); tw.next())
    {
        const int f = tw.cur / tilesPerFrame;
        const int t = tw.cur - f * tilesPerFrame;
        const int ty = t / p.tilesX;
        const int tx = t - ty * p.tilesX;

        __syncthreads();   // every wave is done reading the previous tile
        {
            const f16* src = p.in.base + f * p.in.frame +
                             (long(ty * TH - 1) * p.in.pitch + (tx * TW - 1)) * CIN;
#pragma unroll
            for (int it = 0; it < (NCHUNK + kThreads - 1) / kThreads; ++it)
            {
                const int idx = tid + it * kThreads;
                if (idx < NCHUNK)
                {
                    const int pix = idx / CPP, ch = idx - pix * CPP;
                    const int row = pix / LW, col = pix - row * LW;
                    const uint4 v = *reinterpret_cast<const uint4*>(src + (long(row) * p.in.pitch + col) * CIN + ch * 8);
                    *reinterpret_cast<uint4*>(smem + pix * STRIDE + ch * 16) = v;
                }
            }
        }
        __syncthreads();

        for (int r = rg; r < TH; r += RG)
        {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = bias[q];

            const char* base = smem + (r * LW + px) * STRIDE + h * 16;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
            {
                const int tap = s / CCH, cc = s - tap * CCH;
                const int i = tap / 3, j = tap - 3 * i;
                const f16x8 b = *reinterpret_cast<const f16x8*>(base + (i * LW + j) * STRIDE + cc * 32);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s], b, acc, 0, 0, 0);
            }
            // keep several LDS reads in flight ahead of the MFMA that consumes them
            {
                constexpr int kAhead = 6;
                __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
                for (int k = 0; k < KSTEPS - kAhead; ++k)
                {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 0);
            }

            // D layout: column = lane & 31 (pixel), row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)
            const int y = ty * TH + r;
            const int x = tx * TW + px;
            if (y < p.height && x < p.width)
            {
                if (MODE == MODE_RGB)
                {
                    if (h == 0)
                    {
                        const f16x4 res = to_half4(acc[0], acc[1], acc[2], 0.0f);
                        const f16x4 xin = *reinterpret_cast<const f16x4*>(
                            p.x0.base + f * p.x0.frame + (long(y >> 1) * p.x0.pitch + (x >> 1)) * 4);
                        const f16x4 sum = to_half4((float)res[0] + (float)xin[0], (float)res[1] + (float)xin[1],
                                                   (float)res[2] + (float)xin[2], 0.0f);
                        const unsigned rgba = unorm8((float)sum[0]) | (unorm8((float)sum[1]) << 8) |
                                              (unorm8((float)sum[2]) << 16) | 0xff000000u;
                        *reinterpret_cast<unsigned*>(p.image + f * p.imageFrame + long(y) * p.imagePitch + x * 4) = rgba;
                        if (p.residual.base)
                            *reinterpret_cast<f16x4*>(p.residual.base + f * p.residual.frame +
                                                      (long(y) * p.residual.pitch + x) * 4) = res;
                        if (p.sum.base)
                            *reinterpret_cast<f16x4*>(p.sum.base + f * p.sum.frame +
                                                      (long(y) * p.sum.pitch + x) * 4) = sum;
                    }
                }
                else
                {
                    long pixel;
                    if (MODE == MODE_UP1) pixel = long(2 * y + (nt >> 1)) * p.out.pitch + (2 * x + (nt & 1));
                    else pixel = long(y) * p.out.pitch + x;
                    f16* dst = p.out.base + f * p.out.frame + pixel * COUT + (MODE == MODE_PLAIN ? nt * 32 : 0) + 4 * h;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f16x4*>(dst + 8 * g) =
                            to_half4(fmaxf(acc[4 * g], 0.0f), fmaxf(acc[4 * g + 1], 0.0f),
                                     fmaxf(acc[4 * g + 2], 0.0f), fmaxf(acc[4 * g + 3], 0.0f));
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// conv1: 5x5 on RGB0 pixels (4 halfs).  K = 25 taps x 4 channels, walked as 7
// K-steps of 4 taps; a lane's fragment is two taps x 4 channels.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads, 2) void Conv1Kernel(const ConvParams p)
{
    constexpr int LW = TW + 4, LH = TH + 4;
    constexpr int KSTEPS = 7;
    constexpr int NPIX = LH * LW;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int px = lane & 31;
    const int h = lane >> 5;

    f16x8 wreg[KSTEPS];
    int off0[KSTEPS], off1[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
    {
        wreg[s] = p.wpk[s * 64 + lane];
        int t0 = 4 * s + 2 * h, t1 = t0 + 1;
        t0 = t0 < 25 ? t0 : 0;     // taps 25..27 carry zero weights: read any valid pixel
        t1 = t1 < 25 ? t1 : 0;
        off0[s] = ((t0 / 5) * LW + (t0 % 5)) * 8;
        off1[s] = ((t1 / 5) * LW + (t1 % 5)) * 8;
    }
    float bias[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) bias[q] = p.bias[(q & 3) + 8 * (q >> 2) + 4 * h];

    const int tilesPerFrame = p.tilesX * p.tilesY;
    for (TileWalk tw(p.numTiles); tw.valid(); tw.next())
    {
        const int f = tw.cur / tilesPerFrame;
        const int t = tw.cur - f * tilesPerFrame;
        const int ty = t / p.tilesX;
        const int tx = t - ty * p.tilesX;

        __syncthreads();
        {
            const f16* src = p.in.base + f * p.in.frame + (long(ty * TH - 2) * p.in.pitch + (tx * TW - 2)) * 4;
#pragma unroll
            for (int it = 0; it < (NPIX + kThreads - 1) / kThreads; ++it)
            {
                const int pix = tid + it * kThreads;
                if (pix < NPIX)
                {
                    const int row = pix / LW, col = pix - row * LW;
                    *reinterpret_cast<uint2*>(smem + pix * 8) =
                        *reinterpret_cast<const uint2*>(src + (long(row) * p.in.pitch + col) * 4);
                }
            }
        }
        __syncthreads();

        for (int r = wave; r < TH; r += 4)
        {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = bias[q];
            const char* base = smem + (r * LW + px) * 8;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
            {
                const f16x4 lo = *reinterpret_cast<const f16x4*>(base + off0[s]);
                const f16x4 hi = *reinterpret_cast<const f16x4*>(base + off1[s]);
                const f16x8 b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s], b, acc, 0, 0, 0);
            }
            const int y = ty * TH + r;
            const int x = tx * TW + px;
            if (y < p.height && x < p.width)
            {
                f16* dst = p.out.base + f * p.out.frame + (long(y) * p.out.pitch + x) * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f16x4*>(dst + 8 * g) =
                        to_half4(fmaxf(acc[4 * g], 0.0f), fmaxf(acc[4 * g + 1], 0.0f),
                                 fmaxf(acc[4 * g + 2], 0.0f), fmaxf(acc[4 * g + 3], 0.0f));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// image <-> tensor
// ---------------------------------------------------------------------------
// CSImageToTensor (Content/Shaders/CSImageToTensor.hlsl:15-45): UNORM8 texel ->
// u/255 -> fp16.  One thread per pixel, 4-byte loads and 8-byte stores, both
// fully coalesced along x.  The tensor is stored RGB0 (NHWC, 4 halfs).
__global__ void PackKernel(const uint8_t* src, long srcPitch, long srcFrame, TensorD dst,
                           int width, int height, long total)
{
    for (long i = blockIdx.x * long(blockDim.x) + threadIdx.x; i < total; i += long(gridDim.x) * blockDim.x)
    {
        const long perFrame = long(width) * height;
        const int f = int(i / perFrame);
        const long r = i - f * perFrame;
        const int y = int(r / width), x = int(r - long(y) * width);
        const unsigned v = *reinterpret_cast<const unsigned*>(src + f * srcFrame + y * srcPitch + x * 4);
        const f16x4 o = to_half4(float(v & 0xff) / 255.0f, float((v >> 8) & 0xff) / 255.0f,
                                 float((v >> 16) & 0xff) / 255.0f, 0.0f);
        *reinterpret_cast<f16x4*>(dst.base + f * dst.frame + (long(y) * dst.pitch + x) * 4) = o;
    }
}

// PSTensorToImage (Content/Shaders/PSTensorToImage.hlsl:14-35) + the
// R8G8B8A8_UNORM render-target conversion.
__global__ void UnpackKernel(TensorD src, uint8_t* dst, long dstPitch, long dstFrame,
                             int width, int height, long total)
{
    for (long i = blockIdx.x * long(blockDim.x) + threadIdx.x; i < total; i += long(gridDim.x) * blockDim.x)
    {
        const long perFrame = long(width) * height;
        const int f = int(i / perFrame);
        const long r = i - f * perFrame;
        const int y = int(r / width), x = int(r - long(y) * width);
        const f16x4 v = *reinterpret_cast<const f16x4*>(src.base + f * src.frame + (long(y) * src.pitch + x) * 4);
        const unsigned rgba = unorm8((float)v[0]) | (unorm8((float)v[1]) << 8) | (unorm8((float)v[2]) << 16) | 0xff000000u;
        *reinterpret_cast<unsigned*>(dst + f * dstFrame + y * dstPitch + x * 4) = rgba;
    }
}

template <int CIN, int NT, int MODE>
hipError_t launch3x3(const ConvParams& p, int numCUs, hipStream_t s)
{
    constexpr int smem = (TH + 2) * (TW + 2) * (CIN * 2 + 16);
    hipLaunchKernelGGL((Conv3x3Kernel<CIN, NT, MODE>), dim3(gridFor(p.numTiles, numCUs)), dim3(kThreads), smem, s, p);
    return hipGetLastError();
}

}  // namespace

namespace dev {

namespace {
std::mutex g_launchStateMutex;
#if SRX_DIAG
constexpr int kMaxDevices = 64, kProbeBlocks = 4096;
#endif
}

#if SRX_DIAG
int diagEnvInt(const char* name, int fallback)
{
    const char* v = getenv(name);
    return v ? atoi(v) : fallback;
}

ClockProbe* clockProbeFor(ProbedKernel kernel, int blocks, bool mayAllocate, hipStream_t s)
{
    static const bool on = getenv("SRX_CLOCK") != nullptr;
    if (!on || blocks > kProbeBlocks || !mayAllocate) return nullptr;
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &capture) != hipSuccess || capture != hipStreamCaptureStatusNone) return nullptr;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= kMaxDevices) return nullptr;
    static ClockProbe probes[kMaxDevices][PROBE_COUNT];
    std::lock_guard<std::mutex> lock(g_launchStateMutex);
    ClockProbe& probe = probes[device][kernel];
    if (!probe.dev)
    {
        if (hipMalloc(&probe.dev, 2 * kProbeBlocks * sizeof(unsigned long long)) != hipSuccess)
        {
            probe.dev = nullptr;
            return nullptr;
        }
    }
    return &probe;
}

#endif  // SRX_DIAG

hipError_t ensureDynamicLds(const void* kernel, int bytes)
{
    int device = 0;
    const hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    static std::set<std::pair<int, const void*>> done;
    std::lock_guard<std::mutex> lock(g_launchStateMutex);
    if (done.count({device, kernel})) return hipSuccess;
    const hipError_t r = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (r == hipSuccess) done.insert({device, kernel});
    return r;
}

#if SRX_DIAG
void clockProbeReport(ClockProbe* probePtr, const char* name, int blocks, hipStream_t s, int extraWords)
{
    if (!probePtr) return;
    ClockProbe& probe = *probePtr;
    {
        std::lock_guard<std::mutex> lock(g_launchStateMutex);
        if (++probe.launches != 40) return;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return;
    std::vector<unsigned long long> host(2 * size_t(blocks) + size_t(extraWords));
    if (hipMemcpy(host.data(), probe.dev, sizeof(unsigned long long) * host.size(), hipMemcpyDeviceToHost) != hipSuccess) return;
    // extra words: [workgroup 0..15][wave 0..7][8 totals] of a kernel's diagnostic build
    for (int w = 0; w < 8 && extraWords >= 1024; ++w)
    {
        double sum[8] = {};
        for (int b = 0; b < 16; ++b)
            for (int k = 0; k < 8; ++k) sum[k] += double(host[2 * size_t(blocks) + (size_t(b) * 8 + w) * 8 + k]) / 16;
        fprintf(stderr, "[srx] %s wave %d: cycles by section:", name, w);
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %.0f", sum[k]);
        fprintf(stderr, "\n");
    }
    double sum = 0, cyc = 0, usMin = 1e30, usMax = 0, usSum = 0, ghzMin = 1e30, ghzMax = 0;
    double xcdUs[8] = {};
    for (int b = 0; b < blocks; ++b)
    {
        const double ghz = double(host[2 * b]) / double(host[2 * b + 1]) * 0.1, us = double(host[2 * b + 1]) * 0.01;
        sum += ghz;
        cyc += double(host[2 * b]);
        usSum += us;
        usMin = us < usMin ? us : usMin;
        usMax = us > usMax ? us : usMax;
        ghzMin = ghz < ghzMin ? ghz : ghzMin;
        ghzMax = ghz > ghzMax ? ghz : ghzMax;
        xcdUs[b & 7] += us * 8 / blocks;
    }
    fprintf(stderr, "[srx] %s: in-kernel clock %.3f GHz (%.3f - %.3f), %.0f cycles per workgroup; workgroup lifetime %.0f us "
                    "(%.0f - %.0f); by XCD:", name, sum / blocks, ghzMin, ghzMax, cyc / blocks, usSum / blocks, usMin, usMax);
    for (int x = 0; x < 8; ++x) fprintf(stderr, " %.0f", xcdUs[x]);
    fprintf(stderr, "\n");
}

#endif  // SRX_DIAG

}  // namespace dev

namespace {
__global__ void ZeroKernel(uint4* p, size_t count)
{
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += size_t(gridDim.x) * blockDim.x)
        p[i] = uint4{0, 0, 0, 0};
}
}  // namespace

hipError_t PrepareZero()
{
    hipFuncAttributes attr;      // asking for them loads the kernel on the current device
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(ZeroKernel));
}

hipError_t LaunchZero(void* p, size_t bytes, hipStream_t s)
{
    if ((reinterpret_cast<uintptr_t>(p) & 15) != 0 || (bytes & 15) != 0) return hipErrorInvalidValue;
    const size_t count = bytes / 16;
    if (count == 0) return hipSuccess;
    const int blocks = int(std::min<size_t>((count + 255) / 256, 4096));
    hipLaunchKernelGGL(ZeroKernel, dim3(blocks), dim3(256), 0, s, static_cast<uint4*>(p), count);
    return hipGetLastError();
}

hipError_t LaunchPack(const Image& src, const Tensor& dst, int width, int height, int batch, hipStream_t s)
{
    const long total = long(width) * height * batch;
    const int blocks = int((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(PackKernel, dim3(blocks), dim3(256), 0, s, src.base, long(src.pitch), long(src.frame),
                       toDevice(dst), width, height, total);
    return hipGetLastError();
}

hipError_t LaunchUnpack(const Tensor& src, const Image& dst, int width, int height, int batch, hipStream_t s)
{
    const long total = long(width) * height * batch;
    const int blocks = int((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(UnpackKernel, dim3(blocks), dim3(256), 0, s, toDevice(src), dst.base, long(dst.pitch),
                       long(dst.frame), width, height, total);
    return hipGetLastError();
}

hipError_t LaunchConv(const ConvLaunch& l, int numCUs, hipStream_t s)
{
    ConvParams p{};
    p.in = toDevice(l.in);
    p.out = toDevice(l.out);
    p.wpk = reinterpret_cast<const f16x8*>(l.wpk);
    p.bias = l.bias;
    p.width = l.width;
    p.height = l.height;
    p.tilesX = (l.width + TW - 1) / TW;
    p.tilesY = (l.height + TH - 1) / TH;
    p.numTiles = p.tilesX * p.tilesY * l.batch;
    p.x0 = toDevice(l.x0);
    p.residual = toDevice(l.residual);
    p.sum = toDevice(l.sum);
    p.image = l.image.base;
    p.imagePitch = long(l.image.pitch);
    p.imageFrame = long(l.image.frame);
    if (p.numTiles <= 0) return hipErrorInvalidValue;

    switch (l.kind)
    {
    case CONV_5X5_RGB:
    {
        constexpr int smem = (TH + 4) * (TW + 4) * 8;
        hipLaunchKernelGGL(Conv1Kernel, dim3(gridFor(p.numTiles, numCUs)), dim3(kThreads), smem, s, p);
        return hipGetLastError();
    }
    case CONV_3X3_32_64: return launch3x3<32, 2, MODE_PLAIN>(p, numCUs, s);
    case CONV_3X3_64_64: return launch3x3<64, 2, MODE_PLAIN>(p, numCUs, s);
    case CONV_UP1_PHASES: return launch3x3<64, 4, MODE_UP1>(p, numCUs, s);
    case CONV_3X3_32_32: return launch3x3<32, 1, MODE_PLAIN>(p, numCUs, s);
    case CONV_3X3_32_RGB: return launch3x3<32, 1, MODE_RGB>(p, numCUs, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace srx
