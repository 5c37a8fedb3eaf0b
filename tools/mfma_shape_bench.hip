// Development micro-benchmark: the same convolution-like MFMA work (resident filter fragments in
// VGPRs, one 1-KiB activation fragment read from LDS per 32 cycles of MFMA) issued as
// v_mfma_f32_32x32x16_f16 or as v_mfma_f32_16x16x32_f16, two waves per SIMD, random data.
// Prints wall time, in-kernel clock and cycles for both, to decide which shape the kernels should
// use on a power-limited chip; then the 32x32x16 loop with both operands in registers, and both
// loops with 1..6 independent VALU instructions per MFMA (evenly spread or in bursts), to see what
// epilogue work beside the MFMAs costs.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_bench tools/mfma_shape_bench.hip && ./mfma_shape_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLds = 64 * 1024;
constexpr int kIters = 2000;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// SHAPE 0: 32x32x16, 36 MFMAs per iteration per wave (32 pixels x 32 outputs x K = 576).
// SHAPE 1: 16x16x32, 72 MFMAs per iteration per wave (2 pixel tiles x 2 output tiles x 18 K-steps).
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void bench(const f16x8* wsrc, const f16* act, float* out, unsigned long long* clock)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid * 8; i < kLds / 2; i += 256 * 8)
        *reinterpret_cast<f16x8*>(smem + i * 2) = *reinterpret_cast<const f16x8*>(act + i);
    f16x8 w[36];
#pragma unroll
    for (int s = 0; s < 36; ++s) w[s] = wsrc[s * 64 + lane];
#pragma unroll
    for (int s = 0; s < 36; ++s) asm volatile("" ::"v"(w[s]));
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float result = 0.0f;
    if constexpr (SHAPE == 0)
    {
        const int px = lane & 31, h = lane >> 5;
        const int base = px * 144 + h * 16;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        for (int it = 0; it < kIters; ++it)
        {
            const int off = base + (it & 7) * 4608;
#pragma unroll
            for (int k = 0; k < 36; ++k)
            {
                const f16x8 b = *reinterpret_cast<const f16x8*>(smem + off + (k / 4) * 144 + (k % 4) * 32);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[k], b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= 0.001f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) result += acc[r];
    }
    else if constexpr (SHAPE >= 10 && SHAPE < 30)
    {
        // SHAPE 10 + n: the 32x32x16 loop with n independent VALU instructions behind every MFMA
        constexpr int n = SHAPE - 10;
        const int px = lane & 31, h = lane >> 5;
        const int base = px * 144 + h * 16;
        f32x16 acc;
        float d0 = float(lane), d1 = 1.0f, d2 = 2.0f, d3 = 3.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        for (int it = 0; it < kIters; ++it)
        {
            const int off = base + (it & 7) * 4608;
#pragma unroll
            for (int k = 0; k < 36; ++k)
            {
                const f16x8 b = *reinterpret_cast<const f16x8*>(smem + off + (k / 4) * 144 + (k % 4) * 32);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[k], b, acc, 0, 0, 0);
                if constexpr (n >= 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(d0));
                if constexpr (n >= 2) asm volatile("v_add_f32 %0, %0, %0" : "+v"(d1));
                if constexpr (n >= 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(d2));
                if constexpr (n >= 4) asm volatile("v_add_f32 %0, %0, %0" : "+v"(d3));
                if constexpr (n >= 5) asm volatile("v_add_f32 %0, %0, %0" : "+v"(d0));
                if constexpr (n >= 6) asm volatile("v_add_f32 %0, %0, %0" : "+v"(d1));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= 0.001f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) result += acc[r];
        result += d0 + d1 + d2 + d3;
    }
    else if constexpr (SHAPE >= 30)
    {
        // SHAPE 30 + 10*b + n: the 16x16x32 loop with n VALU instructions per MFMA on average, issued
        // in bursts of n << b behind every (1 << b)-th MFMA
        constexpr int n = (SHAPE - 30) % 10, b = (SHAPE - 30) / 10;
        const int p16 = lane & 15, q = lane >> 4;
        const int base = p16 * 160 + q * 16;
        f32x4 acc[4];
        float d[4] = {float(lane), 1.0f, 2.0f, 3.0f};
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < kIters; ++it)
        {
            const int off = base + (it & 7) * 5120;
#pragma unroll
            for (int k = 0; k < 18; ++k)
            {
                const f16x8 b0 = *reinterpret_cast<const f16x8*>(smem + off + (k / 2) * 160 + (k % 2) * 64);
                const f16x8 b1 = *reinterpret_cast<const f16x8*>(smem + off + 16 * 160 + (k / 2) * 160 + (k % 2) * 64);
#pragma unroll
                for (int m = 0; m < 4; ++m)
                {
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2 * k + (m & 1)], m < 2 ? b0 : b1, acc[m], 0, 0, 0);
                    if (((4 * k + m + 1) & ((1 << b) - 1)) == 0)
                    {
#pragma unroll
                        for (int v = 0; v < (n << b); ++v) asm volatile("v_add_f32 %0, %0, %0" : "+v"(d[v & 3]));
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] *= 0.001f;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) result += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
        result += d[0] + d[1] + d[2] + d[3];
    }
    else if constexpr (SHAPE == 3)
    {
        // operands in registers: no LDS reads at all
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        for (int it = 0; it < kIters; ++it)
        {
#pragma unroll
            for (int k = 0; k < 36; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[k], w[(k + 7) % 36], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= 0.001f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) result += acc[r];
    }
    else
    {
        const int p16 = lane & 15, q = lane >> 4;
        const int base = p16 * 160 + q * 16;
        f32x4 acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < kIters; ++it)
        {
            const int off = base + (it & 7) * 5120;
#pragma unroll
            for (int k = 0; k < 18; ++k)
            {
                // K-step k: 32 channels; two pixel tiles share the two filter fragments
                const f16x8 b0 = *reinterpret_cast<const f16x8*>(smem + off + (k / 2) * 160 + (k % 2) * 64);
                const f16x8 b1 = *reinterpret_cast<const f16x8*>(smem + off + 16 * 160 + (k / 2) * 160 + (k % 2) * 64);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2 * k], b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2 * k + 1], b0, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2 * k], b1, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2 * k + 1], b1, acc[3], 0, 0, 0);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] *= 0.001f;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) result += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + tid] = result;
    if (tid == 0) { clock[2 * blockIdx.x] = c1 - c0; clock[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
void run(const char* name, const f16x8* w, const f16* act, float* out, unsigned long long* clock, int blocks)
{
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bench<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(bench<SHAPE>, dim3(blocks), dim3(256), kLds, 0, w, act, out, clock);
    CHECK(hipEventRecord(e0));
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(bench<SHAPE>, dim3(blocks), dim3(256), kLds, 0, w, act, out, clock);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<unsigned long long> host(2 * blocks);
    CHECK(hipMemcpy(host.data(), clock, host.size() * 8, hipMemcpyDeviceToHost));
    double ghz = 0, cyc = 0;
    for (int b = 0; b < blocks; ++b) { ghz += double(host[2 * b]) / double(host[2 * b + 1]) * 0.1; cyc += double(host[2 * b]); }
    const double flop = double(blocks) * 4 * kIters * 36.0 * 32768.0;
    printf("%-10s %8.4f ms  %7.1f TFLOP/s  clock %.3f GHz  %.0f cycles/wg (ideal %d)\n", name, ms, flop / ms / 1e9, ghz / blocks,
           cyc / blocks, kIters * 36 * 32 * 2);
}

int main()
{
    const int blocks = 512;
    std::vector<f16> hw(36 * 64 * 8), ha(kLds / 2);
    srand(1);
    for (auto& v : hw) v = f16((rand() / double(RAND_MAX) - 0.5) * 0.2);
    for (auto& v : ha) { const double u = rand() / double(RAND_MAX); v = f16(u < 0.5 ? 0.0 : u); }
    f16x8* w; f16* act; float* out; unsigned long long* clock;
    CHECK(hipMalloc(&w, hw.size() * 2));
    CHECK(hipMalloc(&act, ha.size() * 2));
    CHECK(hipMalloc(&out, blocks * 256 * 4));
    CHECK(hipMalloc(&clock, blocks * 16));
    CHECK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(act, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    for (int round = 0; round < 1; ++round)
    {
        run<0>("32x32x16", w, act, out, clock, blocks);
        run<1>("16x16x32", w, act, out, clock, blocks);
        run<3>("32x32 reg", w, act, out, clock, blocks);
        run<30>("16: +0", w, act, out, clock, blocks);
        run<31>("16: +1", w, act, out, clock, blocks);
        run<32>("16: +2", w, act, out, clock, blocks);
        run<33>("16: +3", w, act, out, clock, blocks);
        run<51>("16: 4/4", w, act, out, clock, blocks);
        run<52>("16: 8/4", w, act, out, clock, blocks);
        run<61>("16: 8/8", w, act, out, clock, blocks);
        run<62>("16: 16/8", w, act, out, clock, blocks);
        run<10>("+0 VALU", w, act, out, clock, blocks);
        run<11>("+1 VALU", w, act, out, clock, blocks);
        run<12>("+2 VALU", w, act, out, clock, blocks);
        run<14>("+4 VALU", w, act, out, clock, blocks);
        run<15>("+5 VALU", w, act, out, clock, blocks);
        run<16>("+6 VALU", w, act, out, clock, blocks);
    }
    return 0;
}
