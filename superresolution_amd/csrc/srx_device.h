// Device-side helpers shared by the gfx950 kernel files (internal header).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "srx_kernels.h"

// SRX_DIAG=1 (make DIAG=1, a separate libsrx_diag.so) compiles the diagnostic variants of the two fused
// kernels, the in-kernel clock probe and the environment switches that select them (SRX_EXP, SRX_EXP_LR,
// SRX_LR_PRIO, SRX_CLOCK; read once per process).  The shipping library holds one instantiation of each fused
// kernel and reads no environment variable on a launch.
#ifndef SRX_DIAG
#define SRX_DIAG 0
#endif

namespace srx {

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace dev {

constexpr int kThreads = 256;

struct TensorD
{
    f16* base;
    int pitch;
    long frame;
};

// Persistent, XCD-aware tile walk.  Workgroups are dealt round-robin over the 8
// XCDs, so workgroups b and b+8 share an L2: give each XCD one contiguous range
// of tiles (neighbouring tiles share their halo rows/columns through that L2).
struct TileWalk
{
    int cur, end, step;
    __device__ TileWalk(int numTiles)
    {
        const int g = gridDim.x, b = blockIdx.x;
        if ((g & 7) == 0)
        {
            const int xcd = b & 7;
            const long lo = (long(numTiles) * xcd) >> 3;
            const long hi = (long(numTiles) * (xcd + 1)) >> 3;
            cur = int(lo) + (b >> 3);
            end = int(hi);
            step = g >> 3;
        }
        else
        {
            cur = b;
            end = numTiles;
            step = g;
        }
    }
    __device__ bool valid() const { return cur < end; }
    __device__ void next() { cur += step; }
};

__device__ __forceinline__ f16x4 to_half4(float a, float b, float c, float d)
{
    f16x4 v;
    v[0] = (f16)a; v[1] = (f16)b; v[2] = (f16)c; v[3] = (f16)d;
    return v;
}

// float -> R8G8B8A8_UNORM channel: NaN -> 0, clamp, scale, +0.5, drop fraction.
__device__ __forceinline__ unsigned unorm8(float v)
{
    v = fminf(fmaxf(v, 0.0f), 1.0f);   // fmaxf returns its other operand for a NaN: NaN -> 0
    return (unsigned)(v * 255.0f + 0.5f);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter, which would make every step of a rolling-window kernel wait for its global stores to be
// acknowledged and for its prefetch loads to land; LDS writes are complete once lgkmcnt reaches 0.
__device__ __forceinline__ void barrierLds()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#if SRX_DIAG
// Diagnostic (SRX_DIAG builds only): when the environment variable SRX_CLOCK is set, a kernel's workgroups record how many
// shader cycles and how many 100-MHz reference ticks they ran for, and the launcher prints the
// resulting in-kernel clock after a number of launches.  The buffer is read by no kernel.  There is
// one probe per (device, kernel): the table behind clockProbeFor is guarded by a mutex, so launches
// from several host threads (one per GPU) do not share a buffer or race on the counters.
struct ClockProbe
{
    unsigned long long* dev = nullptr;   // [2 * maxBlocks] on the probe's device, or nullptr when the probe is off
    int launches = 0;
};
enum ProbedKernel { PROBE_LR_FUSED = 0, PROBE_HR_FUSED = 1, PROBE_COUNT };
// The probe of `kernel` on the current device, with its buffer allocated; nullptr when SRX_CLOCK is unset,
// when `blocks` exceeds the buffer, when `mayAllocate` is false (a launch that must neither allocate nor
// synchronise: the report does both) or when stream `s` is being captured into a graph.
ClockProbe* clockProbeFor(ProbedKernel kernel, int blocks, bool mayAllocate, hipStream_t s);
void clockProbeReport(ClockProbe* probe, const char* name, int blocks, hipStream_t s, int extraWords = 0);
// Integer value of an environment variable, `fallback` when unset.
int diagEnvInt(const char* name, int fallback);
#endif  // SRX_DIAG

// Raise a kernel's dynamic shared memory limit, once per (device, kernel).  Thread-safe.  Called at set-up
// (PrepareFusedKernels, from srx_create), not on a launch.
hipError_t ensureDynamicLds(const void* kernel, int bytes);

inline TensorD toDevice(const Tensor& t) { return TensorD{reinterpret_cast<f16*>(t.base), t.pitch, long(t.frame)}; }

// Workgroups per launch: two per CU (or one when `perCU` is 1), a multiple of 8 so that the
// XCD-aware walk applies.  With fewer work items than that, one workgroup per item, rounded UP to
// a multiple of 8: the few workgroups without an item leave at once (TileWalk gives them an empty
// range), whereas rounding down would send the last items into a second round.
inline int gridFor(int numItems, int numCUs, int perCU = 2)
{
    int most = numCUs * perCU;
    if (most >= 8) most &= ~7;
    if (most < 1) most = 1;
    if (numItems >= most) return most;
    if (numItems <= 8) return numItems < 1 ? 1 : numItems;
    const int up = (numItems + 7) & ~7;
    return up <= most ? up : most;
}

}  // namespace dev
}  // namespace srx
