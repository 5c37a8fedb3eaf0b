// Development micro-benchmark: cycles per v_mfma_f32_16x16x32_f16 when ONE wave per SIMD (or two) issues a stream of
// them over NACC accumulators, RUN consecutive MFMAs on the same accumulator before moving to the next (operands in
// registers, nothing else in the loop).  Tells what the order of accumulation costs: an MFMA whose C operand is the
// result of the MFMA just before it gets that operand forwarded; any other C operand is read from the register file.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_dep_bench tools/mfma_dep_bench.hip && tools/bin/mfma_dep_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kIters = 4000, kPerIter = 48;

template <int NACC, int RUN, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void bench(float* out, unsigned long long* ticks)
{
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (f16)(0.01f * (lane + j)); b[j] = (f16)(0.02f * (lane - j)); }
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it)
    {
#pragma unroll
        for (int k = 0; k < kPerIter; ++k)
            acc[(k / RUN) % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[(k / RUN) % NACC], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.0f;
    for (int n = 0; n < NACC; ++n) r += acc[n][0] + acc[n][3];
    out[blockIdx.x * THREADS + threadIdx.x] = r;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int NACC, int RUN, int THREADS>
void run(float* out, unsigned long long* ticks, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((bench<NACC, RUN, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, ticks);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((bench<NACC, RUN, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, ticks);
    hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) exit(1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t = 0;
    hipMemcpy(&t, ticks, sizeof(t), hipMemcpyDeviceToHost);
    const double perWave = double(t) / (double(kIters) * kPerIter);
    const double flop = double(blocks) * (THREADS / 64) * kIters * kPerIter * 16384.0;
    printf("%d wave(s) per SIMD, %d accumulator(s), runs of %d: %.2f s_memtime ticks per MFMA of the SIMD; %.3f ms by events = %.0f TFLOP/s\n",
           THREADS / 256, NACC, RUN, perWave / (THREADS / 256), ms, flop / ms * 1e-9);
}

int main()
{
    float* out;
    unsigned long long* ticks;
    const int blocks = 256;
    if (hipMalloc(&out, blocks * 512 * sizeof(float)) != hipSuccess || hipMalloc(&ticks, blocks * sizeof(*ticks)) != hipSuccess) return 1;
    run<1, 1, 256>(out, ticks, blocks);
    run<2, 1, 256>(out, ticks, blocks);
    run<4, 1, 256>(out, ticks, blocks);
    run<4, 2, 256>(out, ticks, blocks);
    run<4, 3, 256>(out, ticks, blocks);
    run<4, 4, 256>(out, ticks, blocks);
    run<4, 6, 256>(out, ticks, blocks);
    run<4, 12, 256>(out, ticks, blocks);
    run<1, 1, 512>(out, ticks, blocks);
    run<4, 1, 512>(out, ticks, blocks);
    run<4, 2, 512>(out, ticks, blocks);
    run<4, 4, 512>(out, ticks, blocks);
    run<4, 12, 512>(out, ticks, blocks);
    return 0;
}
