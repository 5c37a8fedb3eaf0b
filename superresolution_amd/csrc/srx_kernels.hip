// gfx950 (MI355X, CDNA4) kernels of the super-resolution path.
//
// Every convolution is an implicit GEMM on v_mfma_f32_32x32x16_f16 with
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

#include <cstdio>
#include <cstdlib>
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

// ---------------------------------------------------------------------------
// conv4 -> conv5 -> conv6 -> + NN2x(input) -> UNORM8, fused.
//
// A workgroup walks a strip of kHrStrip output columns downwards one row per
// step, keeping a rolling window of rows of conv_up1's output and of the two
// intermediate activations in LDS rings, so neither intermediate ever touches
// HBM and no row is computed twice vertically.  Per step the three layers work
// on different rows (conv4 on row t, conv5 on row t-2, conv6 on row t-4) and one
// barrier separates steps.  Wave w: layer w>>1 (conv4 or conv5), 32-pixel half
// w&1 of the 64 pixels computed per row, plus the 16-pixel quarter w of the
// conv6 row (v_mfma_f32_16x16x32_f16, output channels padded 3 -> 16).
// Every wave therefore issues 18 + 9 MFMAs per step: perfectly balanced.
//
// Column bookkeeping for a strip whose first output column is X0:
//   conv6 pixel i  = column X0 + i      (60 of 64 kept), reads a6 index i..i+2
//   a6 ring index i = column X0 - 1 + i = conv5 pixel i, which reads a5 index i..i+2
//   a5 ring index i = column X0 - 2 + i = conv4 pixel i, which reads a4 index i..i+2
//   a4 ring index i = column X0 - 3 + i   (66 pixels loaded per row)
// Activations outside the true image are written as zeros: that is the zero
// padding each of the reference's convolutions applies to its own input.
// ---------------------------------------------------------------------------
constexpr int kHrStrip = 60;
constexpr int kHrRowPix = 66;
constexpr int kHrStride32 = 80;                       // a4 / a5 pixel stride (odd number of 16-B slots)
constexpr int kHrStride6 = 96;                        // a6 pixel stride: conflict-free for the 16x16x32 reads
constexpr int kHrRow32 = kHrRowPix * kHrStride32;     // 5280
constexpr int kHrRow6 = kHrRowPix * kHrStride6;       // 6336
constexpr int kHrA4 = 0;
constexpr int kHrA5 = 4 * kHrRow32;
constexpr int kHrA6 = 8 * kHrRow32;
constexpr int kHrSmem = 8 * kHrRow32 + 4 * kHrRow6;   // 67584

struct HrParams
{
    TensorD a4;
    const uint8_t* input;            // RGBA8 input image
    long inputPitch, inputFrame;
    const f16x8 *wpk4, *wpk5, *wpk6;
    const float *bias4, *bias5;
    uint8_t* image;
    long imagePitch, imageFrame;
    int width2, height2;
    int strips, segs, segRows, numItems;
    unsigned long long* stamps;      // diagnostic build only (EXP == 8)
};

__device__ __forceinline__ unsigned long long stampNow()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

template <int EXP>
__global__ __launch_bounds__(kThreads, 2) void HrFusedKernel(const HrParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int px = lane & 31, h = lane >> 5;      // 32x32x16 fragment coordinates
    const int p16 = lane & 15, q = lane >> 4;     // 16x16x32 fragment coordinates
    const int L = wave >> 1, ti = wave & 1;

    f16x8 wmain[18], w6[9];
    {
        const f16x8* src = L ? p.wpk5 : p.wpk4;
#pragma unroll
        for (int s = 0; s < 18; ++s) wmain[s] = src[s * 64 + lane];
#pragma unroll
        for (int s = 0; s < 9; ++s) w6[s] = p.wpk6[s * 64 + lane];
        // make the loads land here, not behind vmcnt waits inside the step loop
#pragma unroll
        for (int s = 0; s < 18; ++s) asm volatile("" ::"v"(wmain[s]));
#pragma unroll
        for (int s = 0; s < 9; ++s) asm volatile("" ::"v"(w6[s]));
    }
    float bias[16];
    {
        const float* src = L ? p.bias5 : p.bias4;
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] = src[(r & 3) + 8 * (r >> 2) + 4 * h];
    }

    // Two workgroups share a CU, so two waves share each SIMD and its one MFMA pipe.  Left alone
    // they fall into lockstep (both in their MFMA phase, then both in their epilogue, pipe idle).
    // A static priority for every other workgroup lets one wave run its MFMA phase unhindered
    // while the other does its epilogue, and vice versa.  Workgroups b and b + #CUs usually share
    // a CU; if they do not, nothing is lost.
    if (EXP != 6 && ((blockIdx.x / (gridDim.x >> 1 ? gridDim.x >> 1 : 1)) & 1)) __builtin_amdgcn_s_setprio(1);

    // never-written pad columns of the rings must hold finite values
    for (int i = tid * 16; i < kHrSmem; i += kThreads * 16) *reinterpret_cast<uint4*>(smem + i) = uint4{0, 0, 0, 0};

    const int inRing = L ? kHrA5 : kHrA4;
    const int outRing = L ? kHrA6 : kHrA5;
    const int outStride = L ? kHrStride6 : kHrStride32;
    const int outRow = L ? kHrRow6 : kHrRow32;
    const int rowLag = L ? 3 : 0;
    const int colOff = (L ? -1 : -2) + ti * 32 + px;
    const int inLane = (ti * 32 + px) * kHrStride32 + h * 16;
    const int in6Lane = (wave * 16 + p16) * kHrStride6 + q * 16;
    const int outLane = (ti * 32 + px) * outStride + h * 8;
    // row loader: 66 pixels x 4 chunks of 16 B; thread t takes chunk t, wave 0 also the last 8
    const int ldOff = (tid >> 2) * kHrStride32 + (tid & 3) * 16;
    const int ldPixX = min(64 + (lane >> 2), kHrRowPix - 1);
    const int ldOffX = ldPixX * kHrStride32 + (lane & 3) * 16;

    unsigned long long stampSum[4] = {0, 0, 0, 0};
    const int perFrame = p.strips * p.segs;
    for (TileWalk tw(p.numItems); tw.valid(); tw.next())
    {
        const int f = tw.cur / perFrame;
        const int rem = tw.cur - f * perFrame;
        const int seg = rem / p.strips;
        const int X0 = (rem - seg * p.strips) * kHrStrip;
        const int Y0 = seg * p.segRows;
        const int Y1 = min(p.height2, Y0 + p.segRows);

        const f16* a4f = p.a4.base + f * p.a4.frame + (long(X0) - 3) * 32;
        const f16* ldSrc = a4f + (tid >> 2) * 32 + (tid & 3) * 8;
        const f16* ldSrcX = a4f + ldPixX * 32 + (lane & 3) * 8;
        const long rowHalfs = long(p.a4.pitch) * 32;

        const __amdgpu_buffer_rsrc_t imageRsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.image + f * p.imageFrame, 0, int(p.imageFrame), 0x00020000);

        // the residual term: NN2x(input) at this lane's conv6 pixel (clamped: masked lanes load a valid pixel)
        const int col6 = X0 + wave * 16 + p16;
        const uint8_t* x0f = p.input + f * p.inputFrame + long(min(col6, p.width2 - 1) >> 1) * 4;

        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
        {
            const int row = Y0 - 3 + rr;
            char* dst = smem + kHrA4 + (row & 3) * kHrRow32;
            *reinterpret_cast<uint4*>(dst + ldOff) = *reinterpret_cast<const uint4*>(ldSrc + row * rowHalfs);
            if (wave == 0) *reinterpret_cast<uint4*>(dst + ldOffX) = *reinterpret_cast<const uint4*>(ldSrcX + row * rowHalfs);
        }
        // Rows in flight in registers.  Step t issues the load of row t+3 into one register set
        // and parks the other set (row t+2, loaded during step t-1) in LDS, so a load has a whole
        // step to land; the loop is unrolled by two so the sets never need copying.
        uint4 rowA = *reinterpret_cast<const uint4*>(ldSrc + Y0 * rowHalfs), rowB = rowA;
        uint4 rowAX = uint4{0, 0, 0, 0}, rowBX = rowAX;
        rowAX = *reinterpret_cast<const uint4*>(ldSrcX + Y0 * rowHalfs);
        // CSImageToTensor on the fly: UNORM8 -> u/255 -> fp16 (u * (1/255) rounds to the same fp16)
        const auto loadX0 = [&](int r6) {
            const unsigned v = *reinterpret_cast<const unsigned*>(x0f + long(min(max(r6, 0), p.height2 - 1) >> 1) * p.inputPitch);
            constexpr float k = 1.0f / 255.0f;
            return to_half4(float(v & 0xff) * k, float((v >> 8) & 0xff) * k, float((v >> 16) & 0xff) * k, 0.0f);
        };
        f16x4 xinA = loadX0(Y0 - 8), xinB = xinA;
        // Accumulators are finished ONE STEP LATER: step t converts / stores what step t-1
        // accumulated while its own MFMAs run, so the epilogue's VALU work hides behind the matrix
        // pipe instead of following it.  Two accumulator sets alternate (loop unrolled by two).
        f32x16 accA, accB;
        f32x4 acc6A = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, acc6B = acc6A;
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[r] = accB[r] = 0.0f;
        __syncthreads();

        const auto step = [&](const int t, uint4& park, uint4& parkX, uint4& load, uint4& loadX,
                              const f16x4& xin, f16x4& xinNext, const f32x16& accPrev, f32x16& acc,
                              const f32x4& acc6Prev, f32x4& acc6) {
            unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            if (EXP == 8) s0 = stampNow();
            const long nrow = min(t + 3, Y1 + 2);
            if (EXP != 4)
            {
                // every wave fetches (and later parks) the 8 extra chunks: identical bytes to identical
                // LDS addresses, and the step stays one basic block
                load = *reinterpret_cast<const uint4*>(ldSrc + nrow * rowHalfs);
                loadX = *reinterpret_cast<const uint4*>(ldSrcX + nrow * rowHalfs);
            }
            // rows: conv4 accumulates row t, conv5 row t-3, conv6 row t-6; the rows finished in this
            // step are one less each
            const int rc = t - rowLag;
            const int r6 = t - 6;
            if (EXP != 1 && EXP != 2) xinNext = loadX0(r6);

            // ---- finish step t-1: conv4 / conv5 row rc-1 -> ring (ReLU, zero outside the image, fp16)
            if (EXP == 3) asm volatile("" ::"v"(accPrev));
            else
            {
                const int col = X0 + colOff;
                const bool inside = (int(rc - 1 >= 0) & int(rc - 1 < p.height2) & int(col >= 0) & int(col < p.width2)) != 0;
                char* dst = smem + outRing + ((rc - 1) & 3) * outRow + outLane;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                {
                    f16x4 v = to_half4(accPrev[4 * g], accPrev[4 * g + 1], accPrev[4 * g + 2], accPrev[4 * g + 3]);
                    v = __builtin_elementwise_max(v, f16x4{0, 0, 0, 0});
                    if (!inside) v = f16x4{0, 0, 0, 0};
                    *reinterpret_cast<f16x4*>(dst + g * 16) = v;
                }
            }
            // ---- finish step t-1: conv6 row r6-1.  D rows 0..2 (lanes 0..15) are R, G, B of pixel
            // wave*16 + p16.  Branch-free: lanes that own no output pixel store through the buffer
            // descriptor at an offset beyond its range, which the hardware drops.
            if (EXP != 2)
            {
                const int ro = r6 - 1;
                const f16x4 res = to_half4(acc6Prev[0], acc6Prev[1], acc6Prev[2], 0.0f);
                const f16x4 sum = to_half4((float)res[0] + (float)xin[0], (float)res[1] + (float)xin[1],
                                           (float)res[2] + (float)xin[2], 0.0f);
                const unsigned rgba = unorm8((float)sum[0]) | (unorm8((float)sum[1]) << 8) |
                                      (unorm8((float)sum[2]) << 16) | 0xff000000u;
                const bool owns = (int(q == 0) & int(wave * 16 + p16 < kHrStrip) & int(col6 < p.width2) & int(ro >= Y0) & int(ro < Y1)) != 0;
                const unsigned off = owns ? unsigned(ro) * unsigned(p.imagePitch) + unsigned(col6) * 4u : 0xffffffffu;
                __builtin_amdgcn_raw_buffer_store_b32(rgba, imageRsrc, int(off), 0, 0);
            }

            // ---- conv4 (row t) or conv5 (row t - 3)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bias[r];
#pragma unroll
            for (int i = 0; i < 3; ++i)
            {
                const char* base = smem + inRing + ((rc + i - 1) & 3) * kHrRow32 + inLane;
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc)
                    {
                        const f16x8 b = *reinterpret_cast<const f16x8*>(base + j * kHrStride32 + cc * 32);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wmain[(i * 3 + j) * 2 + cc], b, acc, 0, 0, 0);
                    }
            }

            // ---- conv6 (row t - 6), 16 pixels per wave
            acc6 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (EXP != 2)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                {
                    const char* base = smem + kHrA6 + ((r6 + i - 1) & 3) * kHrRow6 + in6Lane;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                    {
                        const f16x8 b = *reinterpret_cast<const f16x8*>(base + j * kHrStride6);
                        acc6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w6[i * 3 + j], b, acc6, 0, 0, 0);
                    }
                }
            if (EXP != 6)
            {
                // LDS reads run kAhead in front of the MFMAs that consume them, and the previous
                // step's epilogue (VALU) is spread over the gaps between MFMAs
                constexpr int kAhead = 6;
                constexpr int kValuPerGap = EXP == 7 ? 0 : 3;
                __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
                for (int k = 0; k < 27; ++k)
                {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (k < 27 - kAhead) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (kValuPerGap) __builtin_amdgcn_sched_group_barrier(0x002, kValuPerGap, 0);
                }
            }
            if (EXP == 8) s1 = stampNow();

            // park row t+2
            {
                char* dst = smem + kHrA4 + ((t + 2) & 3) * kHrRow32;
                *reinterpret_cast<uint4*>(dst + ldOff) = park;
                *reinterpret_cast<uint4*>(dst + ldOffX) = parkX;
            }
            if (EXP == 8) s2 = stampNow();
            if (EXP != 5) barrierLds();
            if (EXP == 8)
            {
                s3 = stampNow();
                stampSum[0] += s1 - s0;
                stampSum[1] += s2 - s1;
                stampSum[2] += s3 - s2;
                stampSum[3] += 1;
            }
        };

        // steps Y0-2 .. Y1+6, rounded up to an even count (a surplus step only touches ring rows
        // nothing reads and stores nothing)
        for (int t = Y0 - 2; t <= Y1 + 6; t += 2)
        {
            step(t, rowA, rowAX, rowB, rowBX, xinA, xinB, accA, accB, acc6A, acc6B);
            step(t + 1, rowB, rowBX, rowA, rowAX, xinB, xinA, accB, accA, acc6B, acc6A);
        }
    }
    if (EXP == 8 && p.stamps && lane == 0)
        for (int k = 0; k < 4; ++k) p.stamps[(blockIdx.x * 4 + wave) * 4 + k] = stampSum[k];
}

template <int CIN, int NT, int MODE>
hipError_t launch3x3(const ConvParams& p, int numCUs, hipStream_t s)
{
    constexpr int smem = (TH + 2) * (TW + 2) * (CIN * 2 + 16);
    hipLaunchKernelGGL((Conv3x3Kernel<CIN, NT, MODE>), dim3(gridFor(p.numTiles, numCUs)), dim3(kThreads), smem, s, p);
    return hipGetLastError();
}

}  // namespace

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

hipError_t LaunchHrFused(const HrFusedLaunch& l, int numCUs, hipStream_t s)
{
    HrParams p{};
    p.a4 = toDevice(l.a4);
    p.input = l.input.base;
    p.inputPitch = long(l.input.pitch);
    p.inputFrame = long(l.input.frame);
    p.wpk4 = reinterpret_cast<const f16x8*>(l.wpk4);
    p.wpk5 = reinterpret_cast<const f16x8*>(l.wpk5);
    p.wpk6 = reinterpret_cast<const f16x8*>(l.wpk6);
    p.bias4 = l.bias4;
    p.bias5 = l.bias5;
    p.image = l.image.base;
    p.imagePitch = long(l.image.pitch);
    p.imageFrame = long(l.image.frame);
    p.width2 = l.width2;
    p.height2 = l.height2;
    p.strips = (l.width2 + kHrStrip - 1) / kHrStrip;
    // split the rows so that there are about two work items per CU; every segment re-computes
    // 4 + 2 warm-up rows, so keep segments at least 32 rows tall
    int segs = (2 * numCUs + p.strips * l.batch - 1) / (p.strips * l.batch);
    const int maxSegs = l.height2 / 32 > 1 ? l.height2 / 32 : 1;
    segs = segs < 1 ? 1 : (segs > maxSegs ? maxSegs : segs);
    p.segRows = (l.height2 + segs - 1) / segs;
    p.segs = (l.height2 + p.segRows - 1) / p.segRows;
    p.numItems = p.strips * p.segs * l.batch;
    if (p.numItems <= 0) return hipErrorInvalidValue;
    const char* env = getenv("SRX_EXP");
    const int exp = env ? atoi(env) : 0;
    const char* perCuEnv = getenv("SRX_HR_PER_CU");
    const dim3 grid(gridFor(p.numItems, numCUs, perCuEnv ? atoi(perCuEnv) : 2)), block(kThreads);
    switch (exp)
    {
    case 1: hipLaunchKernelGGL(HrFusedKernel<1>, grid, block, kHrSmem, s, p); break;
    case 2: hipLaunchKernelGGL(HrFusedKernel<2>, grid, block, kHrSmem, s, p); break;
    case 3: hipLaunchKernelGGL(HrFusedKernel<3>, grid, block, kHrSmem, s, p); break;
    case 4: hipLaunchKernelGGL(HrFusedKernel<4>, grid, block, kHrSmem, s, p); break;
    case 5: hipLaunchKernelGGL(HrFusedKernel<5>, grid, block, kHrSmem, s, p); break;
    case 6: hipLaunchKernelGGL(HrFusedKernel<6>, grid, block, kHrSmem, s, p); break;
    case 7: hipLaunchKernelGGL(HrFusedKernel<7>, grid, block, kHrSmem, s, p); break;
    case 8:
    {
        // diagnostic: where a step spends its cycles (sums per wave of issue / epilogue / barrier time)
        static unsigned long long* d = nullptr;
        const size_t n = size_t(grid.x) * 16;
        if (!d && hipMalloc(&d, 512 * 16 * 8) != hipSuccess) return hipErrorOutOfMemory;
        p.stamps = d;
        hipLaunchKernelGGL(HrFusedKernel<8>, grid, block, kHrSmem, s, p);
        std::vector<unsigned long long> h(n);
        if (hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return hipErrorUnknown;
        double sum[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < n; ++i) sum[i & 3] += double(h[i]);
        static int once = 0;
        if (once++ == 3)
            fprintf(stderr, "HR stamps per step (cycles): mfma-section %.0f  epilogue %.0f  barrier %.0f  (steps %.0f per wave)\n",
                    sum[0] / sum[3], sum[1] / sum[3], sum[2] / sum[3], sum[3] / (n / 4));
        break;
    }
    default: hipLaunchKernelGGL(HrFusedKernel<0>, grid, block, kHrSmem, s, p); break;
    }
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
