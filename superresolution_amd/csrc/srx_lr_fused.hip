// gfx950 kernel: image -> tensor, conv1, conv2, conv3 and conv_up1 fused.
//
// One workgroup of eight waves walks a strip of kLrStrip low-resolution columns
// downwards, one row per step, keeping rolling windows of the packed input and
// of the conv1 / conv2 / conv3 activations in LDS rings (104 KB).  Nothing but the
// 8-bit input image is read from HBM and nothing but conv_up1's output is
// written.  Per step the layers work on different rows, one barrier per step:
//     conv1 row t | conv2 row t-2 | conv3 row t-4 | conv_up1 row t-6
// Every wave keeps the filter fragments of its share resident in VGPRs:
//     waves 0-3  conv_up1, output phase = wave   (36 fragments, 72 MFMAs / step)
//     waves 4-5  conv3, 32 of the 64 outputs     (36 fragments, 72 MFMAs / step)
//     wave  6    conv2, all 64 outputs           (36 fragments, 72 MFMAs / step)
//     wave  7    conv1 + the input row loader    (10 fragments, 30 MFMAs / step)
// Waves w and w+4 share a SIMD, so every SIMD issues 144 MFMAs per step (102 for
// the last pair).
//
// Column bookkeeping for a strip whose first output column is X0:
//   conv_up1 pixel i = column X0 + i (60 of 64 kept), reads a3 index i..i+2
//   a3 index k = column X0 - 1 + k = conv3 pixel k,   reads a2 index k..k+2
//   a2 index m = column X0 - 2 + m = conv2 pixel m,   reads a1 index m..m+2
//   a1 index n = column X0 - 3 + n = conv1 pixel n,   reads x0 index n..n+4 (n < 66 kept)
//   x0 index e = column X0 - 5 + e                     (70 pixels loaded per row)
// Activations outside the true image are written as zeros: that is the zero
// padding each of the reference's convolutions applies to its own input
// (Content/SuperResolution.cpp:191-238 via ML::Util::CreateConvolutionLayer).
#include "srx_device.h"

#include <cstdlib>

namespace srx {

using namespace dev;

namespace {

constexpr int kLrThreads = 512;
constexpr int kLrStrip = 60;

constexpr int kX0Row = 104 * 8;                 // 104 RGB0 pixels of 4 halfs
constexpr int kA1Stride = 80, kA1Row = 66 * kA1Stride;
constexpr int kA2Stride = 144, kA2Row = 66 * kA2Stride;
constexpr int kBiasUp = 0, kBiasC3 = 32, kBiasC2 = 96, kBiasC1 = 160;   // floats into the bias table
constexpr int kX0Off = 192 * 4;
constexpr int kA1Off = kX0Off + 8 * kX0Row;
constexpr int kA2Off = kA1Off + 4 * kA1Row;
constexpr int kA3Off = kA2Off + 4 * kA2Row;
constexpr int kLrSmem = kA3Off + 4 * kA2Row;    // 104576

struct LrParams
{
    const uint8_t* image;
    long imagePitch, imageFrame;
    TensorD a4;
    const f16x8 *wpk1, *wpk2, *wpk3, *wpkU;
    const float* bias;          // conv_up1[32] | conv3[64] | conv2[64] | conv1[32]
    int width, height;
    int strips, segs, segRows, numItems;
};

// Initial accumulator = bias of output rows (q & 3) + 8 * (q >> 2) + 4 * h, from the LDS table.
__device__ __forceinline__ f32x16 biasAcc(const char* smem, int tableOff, int h)
{
    f32x16 acc;
    const float* t = reinterpret_cast<const float*>(smem) + tableOff + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g)
    {
        const f32x4 v = *reinterpret_cast<const f32x4*>(t + 8 * g);
        acc[4 * g] = v[0]; acc[4 * g + 1] = v[1]; acc[4 * g + 2] = v[2]; acc[4 * g + 3] = v[3];
    }
    return acc;
}

// ReLU + fp16 of the 16 accumulators of one lane, as four packed groups of four channels
// (group g = channels 8g + 4h .. 8g + 4h + 3); zero when the pixel lies outside the image.
__device__ __forceinline__ void reluPack(const f32x16& acc, bool inside, f16x4 (&out)[4])
{
#pragma unroll
    for (int g = 0; g < 4; ++g)
    {
        f16x4 v = to_half4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        v = __builtin_elementwise_max(v, f16x4{0, 0, 0, 0});
        out[g] = inside ? v : f16x4{0, 0, 0, 0};
    }
}

// K-loop scheduling hint: keep `ahead` LDS reads in flight in front of the MFMAs that consume
// them; `perRead` MFMAs follow every read.
// The bias reads that seed the accumulators are LDS reads too and come first.
template <int READS, int PER_READ>
__device__ __forceinline__ void interleaveHint()
{
    constexpr int kAhead = 8;
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * PER_READ + kAhead, 0);
#pragma unroll
    for (int k = 0; k < READS - kAhead; ++k)
    {
        __builtin_amdgcn_sched_group_barrier(0x008, PER_READ, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, kAhead * PER_READ, 0);
}

template <int EXP>
__global__ __launch_bounds__(kLrThreads, 2) void LrFusedKernel(const LrParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int px = lane & 31, h = lane >> 5;
    const int role = wave < 4 ? 0 : (wave < 6 ? 1 : (wave == 6 ? 2 : 3));

    // this wave's filter fragments, resident for the whole kernel
    f16x8 wreg[36];
    {
        const f16x8* src = role == 0 ? p.wpkU + wave * 36 * 64
                         : role == 1 ? p.wpk3 + (wave - 4) * 36 * 64
                         : role == 2 ? p.wpk2 : p.wpk1;
        const int n = role == 3 ? 10 : 36;
#pragma unroll
        for (int s = 0; s < 36; ++s) wreg[s] = src[(s < n ? s : 0) * 64 + lane];
        // make the loads land here: a load still pending at the loop would put a vmcnt wait in front
        // of every MFMA of every step, and those waits would also wait for the steps' global stores
#pragma unroll
        for (int s = 0; s < 36; ++s) asm volatile("" ::"v"(wreg[s]));
    }

    // Waves w and w + 4 share a SIMD and its MFMA pipe and have the same amount of work per step;
    // left alone they run in lockstep (both in MFMAs, then both in epilogues with the pipe idle).
    // A static priority for one of them staggers the two.
    if (!(EXP & 32) && role == 0) __builtin_amdgcn_s_setprio(1);

    // pad columns of the rings are never written: they must hold finite values
    for (int i = tid * 16; i < kLrSmem; i += kLrThreads * 16) *reinterpret_cast<uint4*>(smem + i) = uint4{0, 0, 0, 0};
    __syncthreads();
    if (tid < 192) reinterpret_cast<float*>(smem)[tid] = p.bias[tid];

    const int perFrame = p.strips * p.segs;
    for (TileWalk tw(p.numItems); tw.valid(); tw.next())
    {
        const int f = tw.cur / perFrame;
        const int rem = tw.cur - f * perFrame;
        const int seg = rem / p.strips;
        const int X0 = (rem - seg * p.strips) * kLrStrip;
        const int Y0 = seg * p.segRows;
        const int Y1 = min(p.height, Y0 + p.segRows);
        const uint8_t* img = p.image + f * p.imageFrame;

        // input loader (wave 7): row `row`, x0 index e -> column X0 - 5 + e, zero outside the image
        const auto fetch = [&](int row, int e) -> unsigned {
            const int col = X0 - 5 + e;
            const bool inside = row >= 0 && row < p.height && col >= 0 && col < p.width && e < 70;
            return inside ? *reinterpret_cast<const unsigned*>(img + long(row) * p.imagePitch + col * 4) : 0u;
        };
        // CSImageToTensor: UNORM8 -> u/255 -> fp16.  u * (1/255) rounds to the same fp16 as u / 255
        // for all 256 inputs.
        const auto park = [&](int row, int e, unsigned v) {
            constexpr float k = 1.0f / 255.0f;
            const f16x4 o = to_half4(float(v & 0xff) * k, float((v >> 8) & 0xff) * k, float((v >> 16) & 0xff) * k, 0.0f);
            *reinterpret_cast<f16x4*>(smem + kX0Off + (row & 7) * kX0Row + e * 8) = o;
        };

        __syncthreads();
        const int t0 = Y0 - 3;
        if (role == 3)
        {
#pragma unroll
            for (int rr = -2; rr <= 2; ++rr)
            {
                park(t0 + rr, lane, fetch(t0 + rr, lane));
                if (lane < 6) park(t0 + rr, 64 + lane, fetch(t0 + rr, 64 + lane));
            }
        }
        __syncthreads();

        for (int t = t0; t <= Y1 + 5; ++t)
        {
            if (role == 0)
            {
                // ---- conv_up1, phase `wave`, row t - 6 ----
                const int ru = t - 6;
                if (ru >= Y0 && ru < Y1 && !(EXP & 2))
                {
                    const int py = wave >> 1, pxp = wave & 1;
#pragma unroll
                    for (int tile = 0; tile < 2; ++tile)
                    {
                        f32x16 acc = biasAcc(smem, kBiasUp, h);
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                        {
                            const char* base = smem + kA3Off + ((ru + i - 1) & 3) * kA2Row + (tile * 32 + px) * kA2Stride + h * 16;
#pragma unroll
                            for (int j = 0; j < 3; ++j)
#pragma unroll
                                for (int cc = 0; cc < 4; ++cc)
                                {
                                    f16x8 b;
                                    if constexpr ((EXP & 64) != 0) { b = wreg[cc]; asm volatile("" : "+v"(b)); }
                                    else b = *reinterpret_cast<const f16x8*>(base + j * kA2Stride + cc * 32);
                                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[(i * 3 + j) * 4 + cc], b, acc, 0, 0, 0);
                                }
                        }
                        interleaveHint<36, 1>();

                        if constexpr ((EXP & 128) != 0) { asm volatile("" ::"v"(acc)); continue; }
                        const int i = tile * 32 + px;
                        const int col = X0 + i;
                        f16x4 v[4];
                        reluPack(acc, true, v);
                        // widen the stores: after the half-wave swaps a lane holds 8 consecutive channels
                        unsigned d[8];
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                        {
                            const uint2 u = *reinterpret_cast<const uint2*>(&v[g]);
                            d[2 * g] = u.x;
                            d[2 * g + 1] = u.y;
                        }
#pragma unroll
                        for (int pair = 0; pair < 2; ++pair)
#pragma unroll
                            for (int k = 0; k < 2; ++k)
                            {
                                const auto r = __builtin_amdgcn_permlane32_swap(d[4 * pair + k], d[4 * pair + 2 + k], false, false);
                                d[4 * pair + k] = r[0];
                                d[4 * pair + 2 + k] = r[1];
                            }
                        if (i < kLrStrip && col < p.width && !(EXP & 1))
                        {
                            f16* dst = p.a4.base + f * p.a4.frame +
                                       (long(2 * ru + py) * p.a4.pitch + (2 * col + pxp)) * 32 + h * 8;
                            *reinterpret_cast<uint4*>(dst) = uint4{d[0], d[1], d[2], d[3]};
                            *reinterpret_cast<uint4*>(dst + 16) = uint4{d[4], d[5], d[6], d[7]};
                        }
                    }
                }
            }
            else if (role == 1)
            {
                // ---- conv3, outputs 32 * (wave - 4) .., row t - 4 ----
                const int r3 = t - 4;
                if (r3 >= Y0 - 1 && r3 <= Y1 && !(EXP & 4))
                {
                    const int nt = wave - 4;
#pragma unroll
                    for (int tile = 0; tile < 2; ++tile)
                    {
                        f32x16 acc = biasAcc(smem, kBiasC3 + nt * 32, h);
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                        {
                            const char* base = smem + kA2Off + ((r3 + i - 1) & 3) * kA2Row + (tile * 32 + px) * kA2Stride + h * 16;
#pragma unroll
                            for (int j = 0; j < 3; ++j)
#pragma unroll
                                for (int cc = 0; cc < 4; ++cc)
                                {
                                    f16x8 b;
                                    if constexpr ((EXP & 64) != 0) { b = wreg[cc]; asm volatile("" : "+v"(b)); }
                                    else b = *reinterpret_cast<const f16x8*>(base + j * kA2Stride + cc * 32);
                                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[(i * 3 + j) * 4 + cc], b, acc, 0, 0, 0);
                                }
                        }
                        interleaveHint<36, 1>();

                        if constexpr ((EXP & 128) != 0) { asm volatile("" ::"v"(acc)); continue; }
                        const int k = tile * 32 + px;
                        const int col = X0 - 1 + k;
                        f16x4 v[4];
                        reluPack(acc, r3 >= 0 && r3 < p.height && col >= 0 && col < p.width, v);
                        char* dst = smem + kA3Off + (r3 & 3) * kA2Row + k * kA2Stride + nt * 64 + h * 8;
#pragma unroll
                        for (int g = 0; g < 4; ++g) *reinterpret_cast<f16x4*>(dst + g * 16) = v[g];
                    }
                }
            }
            else if (role == 2)
            {
                // ---- conv2, all 64 outputs, row t - 2 ----
                const int r2 = t - 2;
                if (r2 >= Y0 - 2 && r2 <= Y1 + 1 && !(EXP & 8))
                {
#pragma unroll
                    for (int tile = 0; tile < 2; ++tile)
                    {
                        f32x16 acc0 = biasAcc(smem, kBiasC2, h);
                        f32x16 acc1 = biasAcc(smem, kBiasC2 + 32, h);
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                        {
                            const char* base = smem + kA1Off + ((r2 + i - 1) & 3) * kA1Row + (tile * 32 + px) * kA1Stride + h * 16;
#pragma unroll
                            for (int j = 0; j < 3; ++j)
#pragma unroll
                                for (int cc = 0; cc < 2; ++cc)
                                {
                                    f16x8 b;
                                    if constexpr ((EXP & 64) != 0) { b = wreg[cc]; asm volatile("" : "+v"(b)); }
                                    else b = *reinterpret_cast<const f16x8*>(base + j * kA1Stride + cc * 32);
                                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[(i * 3 + j) * 2 + cc], b, acc0, 0, 0, 0);
                                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[18 + (i * 3 + j) * 2 + cc], b, acc1, 0, 0, 0);
                                }
                        }
                        interleaveHint<18, 2>();

                        if constexpr ((EXP & 128) != 0) { asm volatile("" ::"v"(acc0), "v"(acc1)); continue; }
                        const int m = tile * 32 + px;
                        const int col = X0 - 2 + m;
                        const bool inside = r2 >= 0 && r2 < p.height && col >= 0 && col < p.width;
                        f16x4 v[4];
                        char* dst = smem + kA2Off + (r2 & 3) * kA2Row + m * kA2Stride + h * 8;
                        reluPack(acc0, inside, v);
#pragma unroll
                        for (int g = 0; g < 4; ++g) *reinterpret_cast<f16x4*>(dst + g * 16) = v[g];
                        reluPack(acc1, inside, v);
#pragma unroll
                        for (int g = 0; g < 4; ++g) *reinterpret_cast<f16x4*>(dst + 64 + g * 16) = v[g];
                    }
                }
            }
            else
            {
                // ---- the input row t + 3 (issue now, park after the MFMAs) and conv1, row t ----
                const unsigned in0 = fetch(t + 3, lane);
                const unsigned in1 = lane < 6 ? fetch(t + 3, 64 + lane) : 0u;
                const int r1 = t;
                if (r1 <= Y1 + 2 && !(EXP & 8))
                {
#pragma unroll
                    for (int tile = 0; tile < 3; ++tile)
                    {
                        f32x16 acc = biasAcc(smem, kBiasC1, h);
#pragma unroll
                        for (int i = 0; i < 5; ++i)
                        {
                            // K-step (i, u): taps j = 4u + 2h and 4u + 2h + 1 of filter row i, 4 channels each
                            const char* base = smem + kX0Off + ((r1 + i - 2) & 7) * kX0Row + (tile * 32 + px + 2 * h) * 8;
#pragma unroll
                            for (int u = 0; u < 2; ++u)
                            {
                                const f16x4 lo = *reinterpret_cast<const f16x4*>(base + u * 32);
                                const f16x4 hi = *reinterpret_cast<const f16x4*>(base + u * 32 + 8);
                                const f16x8 b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[i * 2 + u], b, acc, 0, 0, 0);
                            }
                        }
                        const int n = tile * 32 + px;
                        const int col = X0 - 3 + n;
                        f16x4 v[4];
                        reluPack(acc, r1 >= 0 && r1 < p.height && col >= 0 && col < p.width, v);
                        if (n < 66)
                        {
                            char* dst = smem + kA1Off + (r1 & 3) * kA1Row + n * kA1Stride + h * 8;
#pragma unroll
                            for (int g = 0; g < 4; ++g) *reinterpret_cast<f16x4*>(dst + g * 16) = v[g];
                        }
                    }
                }
                park(t + 3, lane, in0);
                if (lane < 6) park(t + 3, 64 + lane, in1);
            }
            if (!(EXP & 16)) barrierLds();
        }
    }
}

}  // namespace

hipError_t LaunchLrFused(const LrFusedLaunch& l, int numCUs, hipStream_t s)
{
    LrParams p{};
    p.image = l.image.base;
    p.imagePitch = long(l.image.pitch);
    p.imageFrame = long(l.image.frame);
    p.a4 = toDevice(l.a4);
    p.wpk1 = reinterpret_cast<const f16x8*>(l.wpk1);
    p.wpk2 = reinterpret_cast<const f16x8*>(l.wpk2);
    p.wpk3 = reinterpret_cast<const f16x8*>(l.wpk3);
    p.wpkU = reinterpret_cast<const f16x8*>(l.wpkUp1);
    p.bias = l.bias;
    p.width = l.width;
    p.height = l.height;
    p.strips = (l.width + kLrStrip - 1) / kLrStrip;
    // one work item per CU where the image allows it; a segment re-computes 9 warm-up rows
    int segs = (numCUs + p.strips * l.batch / 2) / (p.strips * l.batch);
    const int maxSegs = l.height / 32 > 1 ? l.height / 32 : 1;
    segs = segs < 1 ? 1 : (segs > maxSegs ? maxSegs : segs);
    p.segRows = (l.height + segs - 1) / segs;
    p.segs = (l.height + p.segRows - 1) / p.segRows;
    p.numItems = p.strips * p.segs * l.batch;
    if (p.numItems <= 0) return hipErrorInvalidValue;
    // raise the dynamic LDS limit once per device
    static bool attr[64] = {};
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) return hipErrorInvalidDevice;
    if (!attr[device])
    {
        attr[device] = true;
    }
    const char* env = getenv("SRX_EXP_LR");
    const int exp = env ? atoi(env) : 0;
    const dim3 grid(gridFor(p.numItems, numCUs, 1)), block(kLrThreads);
#define SRX_LR_CASE(E) case E: hipLaunchKernelGGL(LrFusedKernel<E>, grid, block, kLrSmem, s, p); break;
    switch (exp)
    {
        SRX_LR_CASE(1) SRX_LR_CASE(2) SRX_LR_CASE(4) SRX_LR_CASE(8) SRX_LR_CASE(16) SRX_LR_CASE(32) SRX_LR_CASE(64)
        SRX_LR_CASE(128) SRX_LR_CASE(192) SRX_LR_CASE(208) SRX_LR_CASE(12) SRX_LR_CASE(14) SRX_LR_CASE(6) SRX_LR_CASE(10)
    default:
        SRX_LR_CASE(0)
    }
#undef SRX_LR_CASE
    return hipGetLastError();
}

}  // namespace srx
