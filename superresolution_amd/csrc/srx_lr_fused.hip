// gfx950 kernel: image -> tensor, conv1, conv2, conv3 and conv_up1 fused.
//
// One workgroup of eight waves walks a strip of kLrStrip low-resolution columns
// downwards, one row per step, keeping rolling windows of the packed input and
// of the conv1 / conv2 / conv3 activations in LDS rings (135 KB).  Nothing but the
// 8-bit input image is read from HBM and nothing but conv_up1's output is
// written.  Every wave keeps the filter fragments of its share resident in VGPRs
// and runs chains of MFMAs over tiles of 32 pixels x 32 outputs, one 1-KiB
// ds_read_b128 of activations per 32 cycles of MFMA (v_mfma_f32_16x16x32_f16, four 16 x 16
// accumulators per tile; conv1 uses v_mfma_f32_32x32x16_f16):
//     waves 0-3  conv_up1, 16 of the 32 outputs of one column phase   2 chains of 60 MFMAs per step
//     waves 4-5  conv3, 32 of the 64 outputs     2 chains of 72
//     wave  6    conv2, all 64 outputs           4 chains of 36
//     wave  7    conv1 + the input row loader    3 chains of 7 (32x32x16: the 25 taps x 4 channels, four taps per K-step)
// Waves w and w+4 share a SIMD, so every SIMD has 4224 MFMA cycles of work per step (2592 on
// the last pair).
//
// conv_up1 (a 5x5 filter on the nearest-neighbour-upsampled tensor) is evaluated on the low-resolution tensor in
// the row-shared form.  Per output column phase px the five filter columns fold into three column taps (sums of the
// taps that read the same input column, PackConvUp1RowSharedK32); the five filter rows stay apart.  With
//     S(y) = row0 x[y-1] + row2 x[y] + row4 x[y+1]      T(y) = row1 x[y-1] + row3 x[y]
// the two output rows that input row y makes are out(2y) = S(y) + T(y) and out(2y+1) = S(y) + T(y+1).  A wave walks
// the rows downwards, so T(y+1), accumulated beside S(y) from the same activation fragments, is the next step's
// T(y): it stays in the wave's registers and each step costs 30 instead of 36 products per input pixel and phase.
//
// Software pipeline.  A chain's results are converted and stored while the NEXT
// chain of the same wave runs, a few instructions in each gap between its MFMAs,
// and the first activation fragments of the next chain are fetched during the
// last MFMAs of the current one (across the step barrier too: they come from a
// row that was complete a step earlier).  So a wave's MFMA stream never pauses
// for an epilogue or for LDS latency inside a step.  The last chain of a step
// is finished during the next step, which is why each layer trails its producer
// by three rows rather than two:
//     conv1 row t | conv2 row t-2 | conv3 row t-5 | conv_up1 row t-8
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

#include <type_traits>
#include <utility>

namespace srx {

using namespace dev;

namespace {

constexpr int kLrThreads = 512;
constexpr int kLrStrip = 60;

constexpr int kX0Row = 104 * 8;                 // 104 RGB0 pixels of 4 halfs
// pixel strides that make the 16x16x32 B-operand reads (lane = pixel l & 15, channels 8 * (l >> 4) ..) conflict free
constexpr int kA1Stride = 96, kA1Row = 66 * kA1Stride;
constexpr int kA2Stride = 160, kA2Row = 66 * kA2Stride;
constexpr int kBiasUp = 0, kBiasC3 = 32, kBiasC2 = 96, kBiasC1 = 160;   // floats into the bias table
constexpr int kX0Off = 192 * 4;
constexpr int kA1Off = kX0Off + 8 * kX0Row;     // x0: 8 slots, a1: 4 slots, a2 and a3: 5 slots
constexpr int kA2Off = kA1Off + 4 * kA1Row;
constexpr int kA3Off = kA2Off + 5 * kA2Row;
constexpr int kLrSmem = kA3Off + 5 * kA2Row;    // 138368
constexpr int kAhead = 4;                       // LDS reads in flight ahead of their MFMA

struct LrParams
{
    const uint8_t* image;
    long imagePitch, imageFrame;
    TensorD a4;
    const f16x8 *wpk1, *wpk2, *wpk3, *wpkU;
    const float* bias;          // conv_up1[32] | conv3[64] | conv2[64] | conv1[32]
    int width, height;
    int strips, segs, segRows, numItems;
    unsigned long long* clock;  // diagnostic (see ClockProbe), usually nullptr
    int prioMask;               // bit r: waves of role r run at priority 1
};

template <int N>
using IC = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void forEachIdx(F&& f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void forEach(F&& f) { forEachIdx(f, std::make_integer_sequence<int, N>{}); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Slot of row r in a ring of five rows (r >= -5000).  The value is made opaque so that the compiler
// cannot distribute the multiplication by the row size over it: ring addresses then stay a VGPR base
// plus an immediate that fits the 16-bit offset field of ds_read / ds_write.
__device__ __forceinline__ int slot5(int r) { return __builtin_amdgcn_readfirstlane(int(unsigned(r + 5000) % 5u)); }

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

// ReLU + fp16 of accumulators 4g .. 4g+3 of one lane (channels 8g + 4h .. 8g + 4h + 3).
__device__ __forceinline__ f16x4 reluHalf4(const f32x16& acc, int g)
{
    const f16x4 v = to_half4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    return __builtin_elementwise_max(v, f16x4{0, 0, 0, 0});
}

__device__ __forceinline__ f16x4 reluHalf4(const f32x4& acc)
{
    const f16x4 v = to_half4(acc[0], acc[1], acc[2], acc[3]);
    return __builtin_elementwise_max(v, f16x4{0, 0, 0, 0});
}

// ReLU(fp16(a + b)) of four values as two dwords: the addition and the conversion are ONE instruction per value
// (v_fma_mix{lo,hi}_f16: a * 1.0 + b rounded once, to fp16; an fp32 addition followed by a conversion rounds twice and
// differs from it by one fp16 unit in about one value of 25 000), which issues at the full vector rate: the same number
// of instructions as two packed fp32 additions and two packed conversions, but a packed fp32 addition takes about four
// times as long to issue beside the MFMAs.
__device__ __forceinline__ uint2 addReluHalf4(const f32x4& a, const f32x4& b)
{
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    unsigned lo, hi;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, %2" : "=v"(lo) : "v"(a[0]), "v"(b[0]));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, %2" : "+v"(lo) : "v"(a[1]), "v"(b[1]));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, %2" : "=v"(hi) : "v"(a[2]), "v"(b[2]));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, %2" : "+v"(hi) : "v"(a[3]), "v"(b[3]));
    const f16x2 z = {0, 0};
    const f16x2 l = __builtin_elementwise_max(__builtin_bit_cast(f16x2, lo), z), h = __builtin_elementwise_max(__builtin_bit_cast(f16x2, hi), z);
    return uint2{__builtin_bit_cast(unsigned, l), __builtin_bit_cast(unsigned, h)};
}

// Initial accumulators of a 32-pixel x 32-output tile held as four 16 x 16 tiles (index = 2 * pixel
// half + output half).  The MFMA rows of the two output halves are interleaved in the filter fragments
// (PackConv3x3K32 with `interleave`): lane group q holds outputs 8q .. 8q+3 in the first half's accumulator
// and 8q+4 .. 8q+7 in the second's, i.e. 8 consecutive channels of a pixel = one 16-byte store.
__device__ __forceinline__ void biasAcc16(const char* smem, int tableOff, int q, f32x4 (&acc)[4])
{
    const float* t = reinterpret_cast<const float*>(smem) + tableOff + 8 * q;
    acc[0] = acc[2] = *reinterpret_cast<const f32x4*>(t);
    acc[1] = acc[3] = *reinterpret_cast<const f32x4*>(t + 4);
}

// The first kAhead activation fragments of a chain, fetched while the chain before it runs.
struct Frags
{
    f16x8 v[kAhead];
};

// One chain of N MFMAs into `acc` (already holding the bias).  Fragment k of the B operand is
// rd(k) and its filter fragment w(k); the first kAhead fragments arrive in `pre`, and the first
// kAhead of the following chain, rdNext(k), leave in it.  piece(k) is issued in the gap after
// MFMA k: that is where the previous chain's epilogue runs.
template <int N, bool PIN, class W, class RD, class RDN, class PC>
__device__ __forceinline__ void mfmaChain(f32x16& acc, Frags& pre, W&& w, RD&& rd, RDN&& rdNext, PC&& piece)
{
    f16x8 bq[N];
    Frags nxt;
    forEach<N>([&](auto kk) {
        constexpr int K = decltype(kk)::value;
        if constexpr (K + kAhead < N) bq[K + kAhead] = rd(IC<K + kAhead>{});
        else nxt.v[K + kAhead - N] = rdNext(IC<K + kAhead - N>{});
        if constexpr (K < kAhead) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w(kk), pre.v[K], acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w(kk), bq[K], acc, 0, 0, 0);
        piece(kk);
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    });
    pre = nxt;
}

// The same for v_mfma_f32_16x16x32_f16 on a tile of 32 pixels x 32 outputs: NK K-steps of 32 input
// channels, each with two activation fragments (16-pixel halves, reads 2k and 2k+1) and two filter
// fragments w(k, output half), four MFMAs into four 16 x 16 accumulators.  piece(k) follows K-step k.
// The chip holds a higher clock on this MFMA shape than on 32x32x16 at equal work.
// SKIP (timing-only variants of SRX_DIAG builds): the last SKIP K-steps issue no MFMAs (wrong results).
template <int NK, bool PIN, int SKIP = 0, class W, class RD, class RDN, class PC>
__device__ __forceinline__ void mfmaChain16(f32x4 (&acc)[4], Frags& pre, W&& w, RD&& rd, RDN&& rdNext, PC&& piece)
{
    constexpr int N = 2 * NK;
    f16x8 bq[N];
    Frags nxt;
    forEach<NK>([&](auto kk) {
        constexpr int K = decltype(kk)::value;
        forEach<2>([&](auto hh) {
            constexpr int R = 2 * K + decltype(hh)::value;
            if constexpr (R + kAhead < N) bq[R + kAhead] = rd(IC<R + kAhead>{});
            else nxt.v[R + kAhead - N] = rdNext(IC<R + kAhead - N>{});
        });
        const auto frag = [&](auto rr) -> f16x8 {
            constexpr int R = decltype(rr)::value;
            if constexpr (R < kAhead) return pre.v[R];
            else return bq[R];
        };
        const f16x8 b0 = frag(IC<2 * K>{}), b1 = frag(IC<2 * K + 1>{});
        if constexpr (K < NK - SKIP)
        {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w(kk, IC<0>{}), b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w(kk, IC<1>{}), b0, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w(kk, IC<0>{}), b1, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w(kk, IC<1>{}), b1, acc[3], 0, 0, 0);
        }
        else asm volatile("" ::"v"(b0), "v"(b1));      // the fragments are still read
        piece(kk);
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    });
    pre = nxt;
}

// One tile (32 pixels x 16 outputs) of conv_up1 in the row-shared form: 18 K-steps k = (input row slot, column tap,
// channel half), two activation fragments each (16-pixel halves).  wS(k) multiplies into S; the K-steps of the
// second and third input row (k >= 6) also multiply wT(k) into T: 12 + 48 MFMAs.  The K-steps are issued in the order
// kUpOrder, every third one a two-MFMA step of the first input row, so that the fragments a step waits for were
// requested at least six MFMAs earlier wherever the chain stands (a wave is often alone on its SIMD at the end of a
// step, and then nothing else hides LDS latency).  Position P of the chain: K-step kUpOrder[P], reads 2P and 2P + 1.
// S starts from `initS` (the bias) and T from zero, as the C operand of their first MFMA: no accumulator is
// initialised by a vector instruction.
__device__ constexpr int kUpOrder[18] = {6, 7, 0, 8, 9, 1, 10, 11, 2, 12, 13, 3, 14, 15, 4, 16, 17, 5};
constexpr int kUpAhead = 4;                      // LDS reads a conv_up1 chain keeps in flight ahead of their MFMAs
struct UpFrags
{
    f16x8 v[kUpAhead];
};

template <bool PIN, class WS, class WT, class RD, class RDN, class PC>
__device__ __forceinline__ void upChain(f32x4 (&accS)[2], f32x4 (&accT)[2], const f32x4& initS, UpFrags& pre, WS&& wS, WT&& wT, RD&& rd,
                                        RDN&& rdNext, PC&& piece)
{
    constexpr int NK = 18, N = 2 * NK;
    f16x8 bq[N];
    UpFrags nxt;
    forEach<NK>([&](auto pp) {
        constexpr int P = decltype(pp)::value, K = kUpOrder[P];
        forEach<2>([&](auto hh) {
            constexpr int R = 2 * P + decltype(hh)::value;
            if constexpr (R + kUpAhead < N) bq[R + kUpAhead] = rd(IC<R + kUpAhead>{});
            else nxt.v[R + kUpAhead - N] = rdNext(IC<R + kUpAhead - N>{});
        });
        const auto frag = [&](auto rr) -> f16x8 {
            constexpr int R = decltype(rr)::value;
            if constexpr (R < kUpAhead) return pre.v[R];
            else return bq[R];
        };
        const f16x8 b0 = frag(IC<2 * P>{}), b1 = frag(IC<2 * P + 1>{});
        if constexpr (P == 0)
        {
            static_assert(kUpOrder[0] >= 6, "the first step must start both accumulators");
            accS[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wS(IC<K>{}), b0, initS, 0, 0, 0);
            accT[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wT(IC<K>{}), b0, f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
            accS[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wS(IC<K>{}), b1, initS, 0, 0, 0);
            accT[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wT(IC<K>{}), b1, f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
        }
        else
        {
            accS[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wS(IC<K>{}), b0, accS[0], 0, 0, 0);
            if constexpr (K >= 6) accT[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wT(IC<K>{}), b0, accT[0], 0, 0, 0);
            accS[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wS(IC<K>{}), b1, accS[1], 0, 0, 0);
            if constexpr (K >= 6) accT[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wT(IC<K>{}), b1, accT[1], 0, 0, 0);
        }
        piece(pp);
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    });
    pre = nxt;
}

// EXP (variants of SRX_DIAG builds, env SRX_EXP_LR): 1 = no scheduling pins, the compiler orders the
// epilogue pieces itself; 4 = with SRX_CLOCK, the first 16 workgroups also record the cycles each wave
// waits at the step barrier; 8 = timing only, WRONG RESULTS: what a perfect levelling of the four SIMDs would
// give: the conv3 and conv2 waves drop 20 MFMAs per step each (SIMDs 0-2: 4608 -> 4288 MFMA cycles) and the
// conv1 wave issues 60 more on fragments it reads from the conv2 ring (SIMD 3: 3264 -> 4224), same total.
// The shipping library instantiates EXP = 0 only.
// RGB3: the input image holds 3 bytes per pixel (RGB8) instead of 4; a pixel is then fetched as the unaligned 32-bit
// word that starts at its first byte (the fourth byte, the next pixel's red, is ignored like alpha is).
template <int EXP, bool RGB3 = false>
__global__ __launch_bounds__(kLrThreads, 2) void LrFusedKernel(const LrParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool PIN = (EXP & 1) == 0;
    constexpr int LEVEL = (EXP & 8) ? 5 : 0;         // K-steps without MFMAs per step of a conv3 / conv2 wave
    unsigned long long clk0 = 0, ref0 = 0;
    if constexpr (SRX_DIAG != 0)
    {
        clk0 = __builtin_amdgcn_s_memtime();
        ref0 = __builtin_amdgcn_s_memrealtime();
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int px = lane & 31, h = lane >> 5;      // 32x32x16 fragment coordinates (conv1)
    const int p16 = lane & 15, q = lane >> 4;     // 16x16x32 fragment coordinates
    const int role = wave < 4 ? 0 : (wave < 6 ? 1 : (wave == 6 ? 2 : 3));

    // this wave's filter fragments, resident for the whole kernel
    f16x8 wreg[36];
    {
        const f16x8* src = role == 0 ? p.wpkU + wave * 30 * 64
                         : role == 1 ? p.wpk3 + (wave - 4) * 36 * 64
                         : role == 2 ? p.wpk2 : p.wpk1;
        const int n = role == 3 ? 7 : (role == 0 ? 30 : 36);
#pragma unroll
        for (int s = 0; s < 36; ++s) wreg[s] = src[(s < n ? s : 0) * 64 + lane];
        // make the loads land here: a load still pending at the loop would put a vmcnt wait in front
        // of every MFMA of every step, and those waits would also wait for the steps' global stores
#pragma unroll
        for (int s = 0; s < 36; ++s) asm volatile("" ::"v"(wreg[s]));
    }

    // Waves w and w + 4 share a SIMD and its MFMA pipe.  The second-dispatched half (waves 4-7) loses
    // the issue arbitration by age, hence a static priority for it.  Measured on the final kernel, same run
    // (bit r of the mask = role r at priority 1): conv3 + conv2 (+ conv1) above conv_up1, masks 6 and 14: 3.02 - 3.06 ms
    // per launch; none (0), conv_up1 only (1), all (15), conv3 without conv2 (2, 10): 1.7 - 1.9 % slower; priority 3
    // instead of 1: the same as 1.
    if ((p.prioMask >> role) & 1) __builtin_amdgcn_s_setprio(1);

    // pad columns of the rings are never written: they must hold finite values
    for (int i = tid * 16; i < kLrSmem; i += kLrThreads * 16) *reinterpret_cast<uint4*>(smem + i) = uint4{0, 0, 0, 0};
    __syncthreads();
    if (tid < 192) reinterpret_cast<float*>(smem)[tid] = p.bias[tid];

    // EXP & 4 (with SRX_CLOCK): cycles each wave of the first 16 workgroups spends at the step barrier
    unsigned long long waited = 0, lapA = 0, lapB = 0;      // EXP & 4: ticks at the step barrier, in a step's first and second half
    const auto stepBarrier = [&] {
        if constexpr ((EXP & 4) != 0)
        {
            const unsigned long long a = __builtin_amdgcn_s_memtime();
            barrierLds();
            waited += __builtin_amdgcn_s_memtime() - a;
        }
        else barrierLds();
    };
    // The conv_up1 waves write nothing to LDS, and the LDS reads they have in flight at the end of a step are
    // the first fragments of the NEXT step (rows no wave overwrites during it): their barrier need not wait
    // for LDS at all, so that latency passes while the wave waits for the others (0.5 % of the kernel).
    const auto stepBarrierReader = [&] {
        if constexpr ((EXP & 4) != 0)
        {
            const unsigned long long a = __builtin_amdgcn_s_memtime();
            asm volatile("s_barrier" ::: "memory");
            waited += __builtin_amdgcn_s_memtime() - a;
        }
        else asm volatile("s_barrier" ::: "memory");
    };

    const int perFrame = p.strips * p.segs;
    for (TileWalk tw(p.numItems); tw.valid(); tw.next())
    {
        const int f = tw.cur / perFrame;
        const int rem = tw.cur - f * perFrame;
        const int seg = rem / p.strips;
        const int X0 = (rem - seg * p.strips) * kLrStrip;
        const int Y0 = seg * p.segRows;
        const int Y1 = min(p.height, Y0 + p.segRows);
        const int t0 = Y0 - 3, tEnd = Y1 + 8;
        const uint8_t* img = p.image + f * p.imageFrame;

        // input loader (wave 7): row `row`, x0 index e -> column X0 - 5 + e, zero outside the image
        const auto fetch = [&](int row, int e) -> unsigned {
            const int col = X0 - 5 + e;
            const bool inside = row >= 0 && row < p.height && col >= 0 && col < p.width && e < 70;
            typedef unsigned __attribute__((aligned(1))) UnalignedWord;
            if constexpr (RGB3) return inside ? *reinterpret_cast<const UnalignedWord*>(img + long(row) * p.imagePitch + col * 3) : 0u;
            else return inside ? *reinterpret_cast<const unsigned*>(img + long(row) * p.imagePitch + col * 4) : 0u;
        };
        // CSImageToTensor: UNORM8 -> u/255 -> fp16.  u * (1/255) rounds to the same fp16 as u / 255
        // for all 256 inputs.
        const auto park = [&](int row, int e, unsigned v) {
            constexpr float k = 1.0f / 255.0f;
            const f16x4 o = to_half4(float(v & 0xff) * k, float((v >> 8) & 0xff) * k, float((v >> 16) & 0xff) * k, 0.0f);
            *reinterpret_cast<f16x4*>(smem + kX0Off + (row & 7) * kX0Row + e * 8) = o;
        };

        __syncthreads();
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

        if (role == 0)
        {
            // ---- conv_up1, column phase pxp, outputs 16 oh .. 16 oh + 15: input row ru at step ru + 8 makes output
            // rows 2 ru and 2 ru + 1, rows Y0 .. Y1-1; one step before that a pass over row Y0 - 1 leaves T(Y0) in the
            // registers (its outputs are dropped), and one more step at the end finishes the last tile
            const int pxp = wave & 1, oh = wave >> 1;
            const int laneRd = kA3Off + p16 * kA2Stride + q * 16;      // + pixel and tap offsets
            // A lane holds 4 consecutive channels of one pixel (8 B of an a4 row): channels 16 oh + 4q .. + 3.
            // Lanes that own no pixel get an offset no buffer descriptor covers, so their stores are dropped.
            unsigned laneSt[4];
#pragma unroll
            for (int T = 0; T < 4; ++T)        // T = 16-pixel quarter of the 64-pixel row
            {
                const int i = T * 16 + p16, col = X0 + i;
                laneSt[T] = (i < kLrStrip && col < p.width)
                                ? unsigned((2 * col + pxp) * 64 + oh * 32 + q * 8) : 0xC0000000u;
            }
            const long rowBytes = long(p.a4.pitch) * 64;
            char* a4f = reinterpret_cast<char*>(p.a4.base + f * p.a4.frame);
            const f32x4 biasV = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem) + kBiasUp + oh * 16 + 4 * q);
            const f32x4 zeroV = {0.0f, 0.0f, 0.0f, 0.0f};
            // finish a tile: per 16-pixel half and output row parity, S + T, ReLU, fp16 and one 8-B store; then T(y+1)
            // becomes the next row's T(y) (the next chain of the tile starts its accumulators anew)
            // `rowPtr` = the first of the tile's two output rows in a4 (kept as a running pointer: building it from the row
            // number costs a 64-bit multiplication per store, and this wave's instruction count is what its step takes)
            unsigned d[2];
            __amdgpu_buffer_rsrc_t rs0, rs1;
            const auto finish = [&](auto kk, f32x4 (&accS)[2], f32x4 (&accTn)[2], f32x4 (&accTp)[2], char* rowPtr, int T32, bool rowValid) {
                // positions 3, 4 | 6, 7 | 9, 10 | 12, 13 of the running chain (four-MFMA steps): convert, then store
                constexpr int P = decltype(kk)::value;
                if constexpr (P >= 3 && P <= 13 && P % 3 != 2)
                {
                    constexpr int part = P / 3 - 1, half = part >> 1, py = part & 1;
                    if constexpr (P % 3 == 0)
                    {
                        const uint2 u = addReluHalf4(accS[half], py ? accTn[half] : accTp[half]);
                        d[0] = u.x;
                        d[1] = u.y;
                    }
                    else
                    {
                        // the rows' descriptors: base = the a4 row, range = one row (none for a row outside the segment)
                        if constexpr (P == 4)
                        {
                            const int range = rowValid ? int(rowBytes) : 0;
                            rs0 = __builtin_amdgcn_make_buffer_rsrc(rowPtr, 0, range, 0x00020000);
                            rs1 = __builtin_amdgcn_make_buffer_rsrc(rowPtr + rowBytes, 0, range, 0x00020000);
                        }
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{d[0], d[1]}, py ? rs1 : rs0, int(laneSt[T32 * 2 + half]), 0, 0);
                    }
                }
            };
            // read r of a chain: K-step kUpOrder[r / 2] = (input row slot, column tap, channel half), 16-pixel half r % 2 of the tile
            const auto rdT = [&](auto rr, int T32, int c0, int c1, int c2) -> f16x8 {
                constexpr int R = decltype(rr)::value;
                constexpr int k = kUpOrder[R / 2], half = R % 2, tap = k / 2, kc = k % 2, i = tap / 3, j = tap % 3;
                const int b = i == 0 ? c0 : (i == 1 ? c1 : c2);
                return *reinterpret_cast<const f16x8*>(smem + b + (T32 * 32 + half * 16 + j) * kA2Stride + kc * 64);
            };
            const auto wS = [&](auto kk) { return wreg[decltype(kk)::value]; };
            const auto wT = [&](auto kk) { return wreg[12 + decltype(kk)::value]; };       // fragments 18.. for K-steps 6..

            int t = t0;
            for (; t < Y0 + 7; ++t) stepBarrier();
            // S of the two tiles, and two T sets per tile that take turns by row: the set that accumulates T(y+1) in
            // one step is the T(y) of the next, and the other one starts anew (the row loop is unrolled by two so that
            // no register is copied)
            f32x4 sA[2] = {biasV, biasV}, sB[2] = {biasV, biasV};
            f32x4 tA[2][2] = {{zeroV, zeroV}, {zeroV, zeroV}}, tB[2][2] = {{zeroV, zeroV}, {zeroV, zeroV}};
            UpFrags pre;
            {
                // the first fragments of a chain come from its first two input rows (kUpOrder): rows Y0 - 2 (not computed
                // by this item, and used for values that are dropped) and Y0 - 1 of the pass over Y0 - 1
                const int b0 = laneRd + slot5(Y0 - 2) * kA2Row, b1 = laneRd + slot5(Y0 - 1) * kA2Row;
                forEach<kUpAhead>([&](auto rr) { pre.v[decltype(rr)::value] = rdT(rr, 0, b0, b1, b1); });
            }
            char* outRow = a4f + 2 * long(Y0 - 1) * rowBytes;      // output row 2 ru of the row in flight
            // ring offsets of input rows ru - 1, ru, ru + 1, rotated from step to step (no division per row)
            int ring0 = slot5(Y0 - 2) * kA2Row, ring1 = slot5(Y0 - 1) * kA2Row, ring2 = slot5(Y0) * kA2Row;
            const auto rowStep = [&](int ru, f32x4 (&tpA)[2], f32x4 (&tnA)[2], f32x4 (&tpB)[2], f32x4 (&tnB)[2]) {
                const int b0 = laneRd + ring0, b1 = laneRd + ring1, b2 = laneRd + ring2;
                unsigned long long lap0 = 0;
                if constexpr ((EXP & 4) != 0) lap0 = __builtin_amdgcn_s_memtime();
                // tile 0; finishes tile 1 of the previous row, whose T(y+1) is this row's T(y) of that tile
                upChain<PIN>(sA, tnA, biasV, pre, wS, wT, [&](auto rr) { return rdT(rr, 0, b0, b1, b2); },
                             [&](auto rr) { return rdT(rr, 1, b0, b1, b2); },
                             [&](auto kk) { finish(kk, sB, tpB, tnB, outRow - 2 * rowBytes, 1, ru - 1 >= Y0); });
                if constexpr ((EXP & 4) != 0)
                {
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned long long now = __builtin_amdgcn_s_memtime();
                    lapA += now - lap0;
                    lap0 = now;
                    __builtin_amdgcn_sched_barrier(0);
                }
                // tile 1; finishes tile 0 of this row; its tail fetches tile 0 of the next row, whose
                // first fragments come from its first two input rows, ru and ru + 1 (slots b1, b2): both complete
                upChain<PIN>(sB, tnB, biasV, pre, wS, wT, [&](auto rr) { return rdT(rr, 1, b0, b1, b2); },
                             [&](auto rr) { return rdT(rr, 0, b1, b2, b2); },
                             [&](auto kk) { finish(kk, sA, tnA, tpA, outRow, 0, ru >= Y0); });
                if constexpr ((EXP & 4) != 0)
                {
                    __builtin_amdgcn_sched_barrier(0);
                    lapB += __builtin_amdgcn_s_memtime() - lap0;
                    __builtin_amdgcn_sched_barrier(0);
                }
                outRow += 2 * rowBytes;
                ring0 = ring1;
                ring1 = ring2;
                ring2 = ring2 + kA2Row == 5 * kA2Row ? 0 : ring2 + kA2Row;
                stepBarrierReader();
            };
            // The additions of `finish` are inline assembly, which the compiler's hazard recogniser does not look
            // into: they must not read an accumulator before the MFMA that wrote it has retired.  Inside the loop they read
            // what the chain before the running one wrote (at least ten MFMAs earlier); after the loop the last tile is
            // finished right behind its last MFMA, hence the explicit wait states.
            const auto mfmaResultsVisible = [] {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            int ru = Y0 - 1;
            for (; ru + 1 < Y1; ru += 2, t += 2)
            {
                rowStep(ru, tA[0], tA[1], tB[0], tB[1]);
                rowStep(ru + 1, tA[1], tA[0], tB[1], tB[0]);
            }
            if (ru < Y1)
            {
                rowStep(ru, tA[0], tA[1], tB[0], tB[1]);
                ++t;
                mfmaResultsVisible();
                forEach<14>([&](auto kk) { finish(kk, sB, tB[1], tB[0], outRow - 2 * rowBytes, 1, true); });
            }
            else
            {
                mfmaResultsVisible();
                forEach<14>([&](auto kk) { finish(kk, sB, tB[0], tB[1], outRow - 2 * rowBytes, 1, true); });
            }
            stepBarrier();
            for (++t; t <= tEnd; ++t) stepBarrier();
        }
        else if (role == 1)
        {
            // ---- conv3, outputs 32 * (wave - 4) ..: row r3 at step r3 + 5, rows Y0-1 .. Y1, and one more
            // step to finish the last tile
            const int nt = wave - 4;
            const int laneRd = kA2Off + p16 * kA2Stride + q * 16;
            const int laneWr = kA3Off + p16 * kA2Stride + nt * 64 + q * 16;
            bool colInside[4];
#pragma unroll
            for (int T = 0; T < 4; ++T) colInside[T] = X0 - 1 + T * 16 + p16 >= 0 && X0 - 1 + T * 16 + p16 < p.width;
            f16x4 vLo, vHi;
            // accumulator a = (16-pixel half a >> 1, 16-output half a & 1) of the 32 x 32 tile; the two output
            // halves of a pixel make one 16-byte store
            const auto finish = [&](auto kk, const f32x4 (&acc)[4], int row, int T32) {
                constexpr int K = decltype(kk)::value;
                if constexpr (K == 1 || K == 3 || K == 9 || K == 11)
                {
                    constexpr int a = K == 1 ? 0 : (K == 3 ? 1 : (K == 9 ? 2 : 3));
                    const bool inside = colInside[T32 * 2 + (a >> 1)] && row >= 0 && row < p.height;
                    const f16x4 v = reluHalf4(acc[a]);
                    ((a & 1) ? vHi : vLo) = inside ? v : f16x4{0, 0, 0, 0};
                }
                if constexpr (K == 5 || K == 13)
                {
                    constexpr int half = K == 5 ? 0 : 1;
                    *reinterpret_cast<f16x8*>(smem + laneWr + slot5(row) * kA2Row + (T32 * 32 + half * 16) * kA2Stride) =
                        __builtin_shufflevector(vLo, vHi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            };
            const auto rdT = [&](auto rr, int T32, int c0, int c1, int c2) -> f16x8 {
                constexpr int R = decltype(rr)::value;
                constexpr int k = R / 2, half = R % 2, tap = k / 2, kc = k % 2, i = tap / 3, j = tap % 3;
                const int b = i == 0 ? c0 : (i == 1 ? c1 : c2);
                return *reinterpret_cast<const f16x8*>(smem + b + (T32 * 32 + half * 16 + j) * kA2Stride + kc * 64);
            };
            const auto w = [&](auto kk, auto hh) { return wreg[decltype(kk)::value * 2 + decltype(hh)::value]; };

            int t = t0;
            for (; t < Y0 + 4; ++t) stepBarrier();
            f32x4 accA[4], accB[4];
            biasAcc16(smem, kBiasC3 + nt * 32, q, accA);
            biasAcc16(smem, kBiasC3 + nt * 32, q, accB);
            Frags pre;
            {
                const int b0 = laneRd + slot5(Y0 - 2) * kA2Row;
                forEach<kAhead>([&](auto rr) { pre.v[decltype(rr)::value] = rdT(rr, 0, b0, b0, b0); });
            }
            for (int r3 = Y0 - 1; r3 <= Y1; ++r3, ++t)
            {
                const int b0 = laneRd + slot5(r3 - 1) * kA2Row, b1 = laneRd + slot5(r3) * kA2Row,
                          b2 = laneRd + slot5(r3 + 1) * kA2Row;
                mfmaChain16<18, PIN>(accA, pre, w, [&](auto rr) { return rdT(rr, 0, b0, b1, b2); },
                                     [&](auto rr) { return rdT(rr, 1, b0, b1, b2); },
                                     [&](auto kk) {
                                         finish(kk, accB, r3 - 1, 1);
                                         if constexpr (decltype(kk)::value == 16) biasAcc16(smem, kBiasC3 + nt * 32, q, accB);
                                     });
                mfmaChain16<18, PIN, LEVEL>(accB, pre, w, [&](auto rr) { return rdT(rr, 1, b0, b1, b2); },
                                     [&](auto rr) { return rdT(rr, 0, b1, b1, b1); },
                                     [&](auto kk) {
                                         finish(kk, accA, r3, 0);
                                         if constexpr (decltype(kk)::value == 16) biasAcc16(smem, kBiasC3 + nt * 32, q, accA);
                                     });
                stepBarrier();
            }
            forEach<16>([&](auto kk) { finish(kk, accB, Y1, 1); });
            stepBarrier();
            for (++t; t <= tEnd; ++t) stepBarrier();
        }
        else if (role == 2)
        {
            // ---- conv2, all 64 outputs as four chains (tile, half of the outputs): row r2 at step r2 + 2,
            // rows Y0-2 .. Y1+1, and one more step to finish the last chain
            const int laneRd = kA1Off + p16 * kA1Stride + q * 16;
            const int laneWr = kA2Off + p16 * kA2Stride + q * 16;
            bool colInside[4];
#pragma unroll
            for (int T = 0; T < 4; ++T) colInside[T] = X0 - 2 + T * 16 + p16 >= 0 && X0 - 2 + T * 16 + p16 < p.width;
            f16x4 vLo, vHi;
            const auto finish = [&](auto kk, const f32x4 (&acc)[4], int row, int T32, int oTile) {
                constexpr int K = decltype(kk)::value;
                if constexpr (K == 0 || K == 1 || K == 4 || K == 5)
                {
                    constexpr int a = K == 0 ? 0 : (K == 1 ? 1 : (K == 4 ? 2 : 3));
                    const bool inside = colInside[T32 * 2 + (a >> 1)] && row >= 0 && row < p.height;
                    const f16x4 v = reluHalf4(acc[a]);
                    ((a & 1) ? vHi : vLo) = inside ? v : f16x4{0, 0, 0, 0};
                }
                if constexpr (K == 2 || K == 6)
                {
                    constexpr int half = K == 2 ? 0 : 1;
                    *reinterpret_cast<f16x8*>(smem + laneWr + slot5(row) * kA2Row + (T32 * 32 + half * 16) * kA2Stride + oTile * 64) =
                        __builtin_shufflevector(vLo, vHi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            };
            // read r of a chain: K-step r / 2 = tap (32 input channels), 16-pixel half r % 2
            const auto rdT = [&](auto rr, int T32, int c0, int c1, int c2) -> f16x8 {
                constexpr int R = decltype(rr)::value;
                constexpr int tap = R / 2, half = R % 2, i = tap / 3, j = tap % 3;
                const int b = i == 0 ? c0 : (i == 1 ? c1 : c2);
                return *reinterpret_cast<const f16x8*>(smem + b + (T32 * 32 + half * 16 + j) * kA1Stride);
            };
            const auto wLo = [&](auto kk, auto hh) { return wreg[decltype(kk)::value * 2 + decltype(hh)::value]; };
            const auto wHi = [&](auto kk, auto hh) { return wreg[18 + decltype(kk)::value * 2 + decltype(hh)::value]; };

            int t = t0;
            for (; t < Y0; ++t) stepBarrier();
            f32x4 acc0[4], acc1[4];
            biasAcc16(smem, kBiasC2, q, acc0);
            biasAcc16(smem, kBiasC2 + 32, q, acc1);
            Frags pre;
            {
                const int b0 = laneRd + ((Y0 - 3) & 3) * kA1Row;
                forEach<kAhead>([&](auto rr) { pre.v[decltype(rr)::value] = rdT(rr, 0, b0, b0, b0); });
            }
            for (int r2 = Y0 - 2; r2 <= Y1 + 1; ++r2, ++t)
            {
                const int b0 = laneRd + ((r2 - 1) & 3) * kA1Row, b1 = laneRd + (r2 & 3) * kA1Row,
                          b2 = laneRd + ((r2 + 1) & 3) * kA1Row;
                // (tile 0, outputs 0-31); finishes (tile 1, outputs 32-63) of the previous row
                mfmaChain16<9, PIN>(acc0, pre, wLo, [&](auto rr) { return rdT(rr, 0, b0, b1, b2); },
                                    [&](auto rr) { return rdT(rr, 0, b0, b1, b2); },
                                    [&](auto kk) {
                                        finish(kk, acc1, r2 - 1, 1, 1);
                                        if constexpr (decltype(kk)::value == 8) biasAcc16(smem, kBiasC2 + 32, q, acc1);
                                    });
                mfmaChain16<9, PIN>(acc1, pre, wHi, [&](auto rr) { return rdT(rr, 0, b0, b1, b2); },
                                    [&](auto rr) { return rdT(rr, 1, b0, b1, b2); },
                                    [&](auto kk) {
                                        finish(kk, acc0, r2, 0, 0);
                                        if constexpr (decltype(kk)::value == 8) biasAcc16(smem, kBiasC2, q, acc0);
                                    });
                mfmaChain16<9, PIN>(acc0, pre, wLo, [&](auto rr) { return rdT(rr, 1, b0, b1, b2); },
                                    [&](auto rr) { return rdT(rr, 1, b0, b1, b2); },
                                    [&](auto kk) {
                                        finish(kk, acc1, r2, 0, 1);
                                        if constexpr (decltype(kk)::value == 8) biasAcc16(smem, kBiasC2 + 32, q, acc1);
                                    });
                mfmaChain16<9, PIN, LEVEL>(acc1, pre, wHi, [&](auto rr) { return rdT(rr, 1, b0, b1, b2); },
                                    [&](auto rr) { return rdT(rr, 0, b1, b1, b1); },
                                    [&](auto kk) {
                                        finish(kk, acc0, r2, 1, 0);
                                        if constexpr (decltype(kk)::value == 8) biasAcc16(smem, kBiasC2, q, acc0);
                                    });
                stepBarrier();
            }
            forEach<8>([&](auto kk) { finish(kk, acc1, Y1 + 1, 1, 1); });
            stepBarrier();
            for (++t; t <= tEnd; ++t) stepBarrier();
        }
        else
        {
            // ---- conv1 (row t, rows Y0-3 .. Y1+2) and the input row t + 3 (issued first, parked after the MFMAs)
            for (int t = t0; t <= tEnd; ++t)
            {
                const unsigned in0 = fetch(t + 3, lane);
                const unsigned in1 = lane < 6 ? fetch(t + 3, 64 + lane) : 0u;
                const int r1 = t;
                if (r1 <= Y1 + 2)
                {
                    // K-step s of a lane half h: taps 4s + 2h and 4s + 2h + 1 of the 25 (row-major, PackConv1), 4
                    // channels each.  Byte offsets of the two taps' pixels from the lane's pixel in ring row 0; the
                    // ring row of filter row i is uniform, which of two rows a lane reads depends on h only.  Taps
                    // 25 .. 27 carry zero weights and read the last tap's pixel again.
                    int rowOff[5];
#pragma unroll
                    for (int i = 0; i < 5; ++i) rowOff[i] = kX0Off + ((r1 + i - 2) & 7) * kX0Row;
                    int tapLo[7], tapHi[7];
                    forEach<7>([&](auto ss) {
                        constexpr int S = decltype(ss)::value;
                        constexpr int t00 = 4 * S < 24 ? 4 * S : 24, t01 = 4 * S + 1 < 24 ? 4 * S + 1 : 24;
                        constexpr int t10 = 4 * S + 2 < 24 ? 4 * S + 2 : 24, t11 = 4 * S + 3 < 24 ? 4 * S + 3 : 24;
                        tapLo[S] = h ? rowOff[t10 / 5] + (t10 % 5) * 8 : rowOff[t00 / 5] + (t00 % 5) * 8;
                        tapHi[S] = h ? rowOff[t11 / 5] + (t11 % 5) * 8 : rowOff[t01 / 5] + (t01 % 5) * 8;
                    });
#pragma unroll
                    for (int tile = 0; tile < 3; ++tile)
                    {
                        f32x16 acc = biasAcc(smem, kBiasC1, h);
                        const char* base = smem + (tile * 32 + px) * 8;
#pragma unroll
                        for (int s7 = 0; s7 < 7; ++s7)
                        {
                            const f16x4 lo = *reinterpret_cast<const f16x4*>(base + tapLo[s7]);
                            const f16x4 hi = *reinterpret_cast<const f16x4*>(base + tapHi[s7]);
                            const f16x8 b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s7], b, acc, 0, 0, 0);
                        }
                        const int n = tile * 32 + px;
                        const int col = X0 - 3 + n;
                        const bool inside = r1 >= 0 && r1 < p.height && col >= 0 && col < p.width;
                        if (n < 66)
                        {
                            char* dst = smem + kA1Off + (r1 & 3) * kA1Row + n * kA1Stride + h * 8;
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                            {
                                const f16x4 v = reluHalf4(acc, g);
                                *reinterpret_cast<f16x4*>(dst + g * 16) = inside ? v : f16x4{0, 0, 0, 0};
                            }
                        }
                    }
                }
                if constexpr (LEVEL != 0)
                {
                    // timing only: 60 MFMAs on 30 fragments of the conv2 ring, results unused
                    f32x4 dummy[4] = {};
                    const char* src = smem + kA2Off + p16 * kA2Stride + q * 16;
#pragma unroll
                    for (int k = 0; k < 15; ++k)
                    {
                        const f16x8 b0 = *reinterpret_cast<const f16x8*>(src + (k % 3) * kA2Stride + (k / 3) * kA2Row);
                        const f16x8 b1 = *reinterpret_cast<const f16x8*>(src + (16 + k % 3) * kA2Stride + (k / 3) * kA2Row);
                        dummy[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[10 + k], b0, dummy[0], 0, 0, 0);
                        dummy[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[11 + k], b0, dummy[1], 0, 0, 0);
                        dummy[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[10 + k], b1, dummy[2], 0, 0, 0);
                        dummy[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[11 + k], b1, dummy[3], 0, 0, 0);
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) asm volatile("" ::"v"(dummy[a]));
                }
                park(t + 3, lane, in0);
                if (lane < 6) park(t + 3, 64 + lane, in1);
                stepBarrier();
            }
        }
    }
    if constexpr ((EXP & 4) != 0)
        if (p.clock && lane == 0 && blockIdx.x < 16)
            for (int k = 0; k < 8; ++k)
                p.clock[2 * gridDim.x + (blockIdx.x * 8 + wave) * 8 + k] = k == 6 ? waited : (k == 0 ? lapA : (k == 1 ? lapB : 0));
    if constexpr (SRX_DIAG != 0)
        if (p.clock && tid == 0)
        {
            p.clock[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
            p.clock[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - ref0;
        }
}

}  // namespace

StripPlan PlanLrFused(int width, int height, int batch, int numCUs)
{
    StripPlan pl;
    if (width <= 0 || height <= 0 || batch <= 0 || numCUs <= 0) return pl;
    pl.strips = (width + kLrStrip - 1) / kLrStrip;
    // Rows are cut into segments where that fills the CUs better: every segment re-computes 12 warm-up
    // rows, and what counts is rounds x steps (one workgroup per CU).  Segments are at least 8 rows tall: a small
    // image (the sample's own are 91 to 256 pixels a side) then spreads over many more CUs than its few strips, and
    // its launch is 20 row steps long instead of 44; for large images the estimate never picks segments that short.
    const int maxSegs = height / 4 > 1 ? (height / 4 < 128 ? height / 4 : 128) : 1;
    long bestCost = 0;
#if SRX_DIAG
    static const int forced = diagEnvInt("SRX_SEGS_LR", 0);      // a fixed number of row segments, for measurements
#else
    constexpr int forced = 0;
#endif
    for (int segs = forced > 0 ? forced : 1; segs <= (forced > 0 ? forced : maxSegs); ++segs)
    {
        StripPlan c = pl;
        c.segRows = (height + segs - 1) / segs;
        c.segs = (height + c.segRows - 1) / c.segRows;
        c.items = c.strips * c.segs * batch;
        c.steps = c.segRows + 12;
        const int slots = gridFor(c.items, numCUs, 1);
        c.rounds = (c.items + slots - 1) / slots;
        c.cost = 100L * c.rounds * c.steps;
        if (segs == 1 || forced > 0 || c.cost < bestCost)
        {
            bestCost = c.cost;
            pl = c;
        }
    }
    return pl;
}

hipError_t PrepareHrFused();
hipError_t PrepareZero();

namespace {
// Static priority of the roles (bit r: waves of role r run at priority 1): conv3, conv2 and conv1 above conv_up1.
constexpr int kLrPrioMask = 14;
#if SRX_DIAG
struct LrDiag { int exp, prio; };
const LrDiag& lrDiag()
{
    static const LrDiag d{diagEnvInt("SRX_EXP_LR", 0), diagEnvInt("SRX_LR_PRIO", kLrPrioMask)};
    return d;
}
#endif
}  // namespace

hipError_t PrepareFusedKernels()
{
    hipError_t e = ensureDynamicLds(reinterpret_cast<const void*>(LrFusedKernel<0>), kLrSmem);
    if (e == hipSuccess) e = ensureDynamicLds(reinterpret_cast<const void*>(LrFusedKernel<0, true>), kLrSmem);
#if SRX_DIAG
    (void)lrDiag();
    if (e == hipSuccess) e = ensureDynamicLds(reinterpret_cast<const void*>(LrFusedKernel<1>), kLrSmem);
    if (e == hipSuccess) e = ensureDynamicLds(reinterpret_cast<const void*>(LrFusedKernel<4>), kLrSmem);
    if (e == hipSuccess) e = ensureDynamicLds(reinterpret_cast<const void*>(LrFusedKernel<8>), kLrSmem);
#endif
    if (e == hipSuccess) e = PrepareHrFused();
    return e != hipSuccess ? e : PrepareZero();
}

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
    const StripPlan plan = PlanLrFused(l.width, l.height, l.batch, numCUs);
    p.strips = plan.strips;
    p.segs = plan.segs;
    p.segRows = plan.segRows;
    p.numItems = plan.items;
    if (p.numItems <= 0 || (l.image.pixelBytes != 3 && l.image.pixelBytes != 4)) return hipErrorInvalidValue;
    const bool rgb3 = l.image.pixelBytes == 3;
    // a row of a4 must be addressable with a 32-bit byte offset
    if (long(l.a4.pitch) * 64 >= (1L << 31)) return hipErrorInvalidValue;

    const dim3 grid(gridFor(p.numItems, numCUs, 1)), block(kLrThreads);
    // The kernels were loaded and their shared-memory limit raised by PrepareFusedKernels (srx_create): a
    // launch is the launch and nothing else.
#if SRX_DIAG
    const int exp = lrDiag().exp;
    p.prioMask = lrDiag().prio;
    ClockProbe* const probe = clockProbeFor(PROBE_LR_FUSED, int(grid.x), l.mayAllocate, s);
    p.clock = probe ? probe->dev : nullptr;
    switch (exp)
    {
    case 1: hipLaunchKernelGGL(LrFusedKernel<1>, grid, block, kLrSmem, s, p); break;
    case 4: hipLaunchKernelGGL(LrFusedKernel<4>, grid, block, kLrSmem, s, p); break;
    case 8: hipLaunchKernelGGL(LrFusedKernel<8>, grid, block, kLrSmem, s, p); break;
    default:
        if (rgb3) hipLaunchKernelGGL((LrFusedKernel<0, true>), grid, block, kLrSmem, s, p);
        else hipLaunchKernelGGL(LrFusedKernel<0>, grid, block, kLrSmem, s, p);
        break;
    }
    clockProbeReport(probe, "LrFusedKernel", int(grid.x), s, (exp & 4) ? 1024 : 0);
#else
    p.prioMask = kLrPrioMask;
    p.clock = nullptr;
    if (rgb3) hipLaunchKernelGGL((LrFusedKernel<0, true>), grid, block, kLrSmem, s, p);
    else hipLaunchKernelGGL(LrFusedKernel<0>, grid, block, kLrSmem, s, p);
#endif
    return hipGetLastError();
}

}  // namespace srx
