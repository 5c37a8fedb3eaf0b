// gfx950 kernel: conv4 -> conv5 -> conv6 -> + NN2x(input) -> UNORM8, fused.
//
// A workgroup (4 waves, 74 KB LDS, two workgroups per CU) walks a strip of
// kHrStrip output columns downwards one row per step, keeping rolling windows of
// conv_up1's output and of the two intermediate activations in LDS rings, so
// neither intermediate touches HBM and no row is computed twice vertically.  One
// LDS barrier separates steps.  Wave w accumulates, per step, its 32-pixel half
// (w & 1) of one row of conv4 (w < 2) or conv5 (w >= 2) as four 16 x 16
// accumulators with 36 v_mfma_f32_16x16x32_f16 (one K-step per filter tap: all 32
// input channels; the same cycles, LDS bytes and registers as 18 of the 32x32x16
// shape, on which the chip holds a lower clock), and its 16-pixel quarter of conv6
// with 3 more: every wave issues the same work.
//
// conv6 (32 -> 3 channels) streams input rows.  The 16 output rows of its MFMA are four
// groups of (R, G, B, 0); group g accumulates the output row y with y mod 4 == g.  The
// newest row R of conv5's output adds filter row 2 to output row R-1, filter row 1 to R
// and filter row 0 to R+1 in ONE MFMA per horizontal tap (filter fragments per R mod 4:
// PackConv6Rotating), so conv6 costs 3 MFMAs per step instead of 9 and reads each
// activation fragment once.  A finished row is moved to a holding register and its group
// cleared; every fourth step all 64 lanes convert and store a block of four rows.
//
// Software pipeline: what step t accumulates is converted and stored by step t+1,
// a few VALU instructions in each gap between that step's MFMAs, so the epilogue
// runs under the matrix pipe instead of after it.  Rows per step t:
//     accumulate: conv4 row t      conv5 row t-3     conv6 input row t-5
//     finish:     conv4 row t-1    conv5 row t-4     conv6 output row t-7 -> holding register
// The loop is unrolled by four so every ring slot and conv6 group is a compile-time
// constant; segments start at multiples of four rows.
//
// Column bookkeeping for a strip whose first output column is X0:
//   conv6 pixel i   = column X0 + i      (60 of 64 kept), reads a6 index i..i+2
//   a6 ring index i = column X0 - 1 + i = conv5 pixel i, which reads a5 index i..i+2
//   a5 ring index i = column X0 - 2 + i = conv4 pixel i, which reads a4 index i..i+2
//   a4 ring index i = column X0 - 3 + i   (66 pixels loaded per row)
// Activations outside the true image are written as zeros: that is the zero
// padding each of the reference's convolutions applies to its own input
// (Content/SuperResolution.cpp:245-268 via ML::Util::CreateConvolutionLayer); the
// residual add and the conversion are ELEMENT_WISE_ADD (Content/SuperResolution.cpp:700-702)
// and PSTensorToImage.hlsl:14-35 + the R8G8B8A8_UNORM render target.
#include "srx_device.h"

#include <cstdlib>
#include <type_traits>

namespace srx {

using namespace dev;

namespace {

constexpr int kHrStrip = 60;
constexpr int kHrRowPix = 66;
constexpr int kHrStride32 = 96;                       // pixel stride of all three rings: 64 B of channels + 32 B, which makes
constexpr int kHrStride6 = 96;                        // the 16x16x32 B-operand reads (lane = pixel l & 15, channels 8 * (l >> 4) ..) conflict free
constexpr int kHrRow32 = kHrRowPix * kHrStride32;     // 6336
constexpr int kHrRow6 = kHrRowPix * kHrStride6;       // 6336
constexpr int kHrA4 = 0;
constexpr int kHrA5 = 4 * kHrRow32;
constexpr int kHrA6 = 8 * kHrRow32;
constexpr int kHrSmem = 8 * kHrRow32 + 4 * kHrRow6;   // 76032
constexpr int kAhead = 4;                             // LDS reads in flight ahead of their first MFMA
constexpr int kReads = 21;                            // per step: 18 (conv4 or conv5: 9 taps x 2 pixel halves) + 3 (conv6)

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
    unsigned long long* clock;       // diagnostic (see ClockProbe), usually nullptr
};

template <int N>
using IC = std::integral_constant<int, N>;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// EXP: 1 = let the compiler place the epilogue pieces (default: every piece is pinned to its MFMA gap; left to
// itself the compiler bunches them at the head of the step, 10 % slower); 2 = no static priority.  Selected by
// env SRX_EXP (diagnostic).
template <int EXP>
__global__ __launch_bounds__(kThreads, 2) void HrFusedKernel(const HrParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int p16 = lane & 15, q = lane >> 4;     // 16x16x32 fragment coordinates: pixel (or output) within 16, group of 8 channels
    const int L = wave >> 1, ti = wave & 1;

    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), ref0 = __builtin_amdgcn_s_memrealtime();
    f16x8 wmain[18], w6[12];
    {
        const f16x8* src = L ? p.wpk5 : p.wpk4;
#pragma unroll
        for (int s = 0; s < 18; ++s) wmain[s] = src[s * 64 + lane];
#pragma unroll
        for (int s = 0; s < 12; ++s) w6[s] = p.wpk6[s * 64 + lane];
        // make the loads land here, not behind vmcnt waits inside the step loop
#pragma unroll
        for (int s = 0; s < 18; ++s) asm volatile("" ::"v"(wmain[s]));
#pragma unroll
        for (int s = 0; s < 12; ++s) asm volatile("" ::"v"(w6[s]));
    }
    // Rows of the two 16-output halves are interleaved (PackConv3x3K32 with `interleave`): lane group q holds
    // outputs 8q .. 8q+3 of the first half and 8q+4 .. 8q+7 of the second, i.e. 8 consecutive channels = one
    // 16-byte LDS store per pixel instead of two 8-byte ones (which collide four ways at this pixel stride).
    f32x4 biasLo, biasHi;
    {
        const float* src = L ? p.bias5 : p.bias4;
        biasLo = *reinterpret_cast<const f32x4*>(src + 8 * q);
        biasHi = *reinterpret_cast<const f32x4*>(src + 8 * q + 4);
    }

    // Two workgroups share a CU, so two waves share each SIMD and its one MFMA pipe.  Workgroups b
    // and b + #CUs share a CU; the second gets a higher wave priority.  Measured, that changes
    // nothing (priority 0, 1 or 3: one workgroup of each pair gets about 65 % of the pipe and
    // finishes first either way, and a pair made to advance evenly is slower in aggregate); it is
    // kept so that which of the two runs ahead does not depend on arbitration details.
    if ((EXP & 2) == 0 && ((blockIdx.x / (gridDim.x >> 1 ? gridDim.x >> 1 : 1)) & 1)) __builtin_amdgcn_s_setprio(1);

    // never-written pad columns of the rings must hold finite values
    for (int i = tid * 16; i < kHrSmem; i += kThreads * 16) *reinterpret_cast<uint4*>(smem + i) = uint4{0, 0, 0, 0};

    // Ring slots.  Row r of an item lives in slot (r - t0) & 3, t0 = the item's first step.  With
    // the loop unrolled by four the phase PH = (t - t0) & 3 is a compile-time constant, and so is
    // every slot except for the layer-dependent lag (conv5 trails conv4 by 3 rows = 1 slot): that
    // one rotation is folded into these tables, indexed by a compile-time (PH + offset) & 3.
    const int rot = L ? 1 : 0;                     // (-lag) & 3
    const int inRing = L ? kHrA5 : kHrA4;
    const int outRing = L ? kHrA6 : kHrA5;
    constexpr int outStride = kHrStride32;
    constexpr int outRow = kHrRow32;
    static_assert(kHrStride6 == kHrStride32, "one pixel stride for all rings");
    int inSlot[4], outSlot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        inSlot[k] = inRing + ((k + rot) & 3) * kHrRow32;
        outSlot[k] = outRing + ((k + rot) & 3) * outRow;
    }
    const int lag = L ? 3 : 0;
    const int colOff = (L ? -1 : -2) + ti * 32 + p16;              // + 16 for the second pixel half
    const int inLane = (ti * 32 + p16) * kHrStride32 + q * 16;
    const int in6Lane = kHrA6 + (wave * 16 + p16) * kHrStride6 + q * 16;
    const int outLane = (ti * 32 + p16) * outStride + q * 16;
    // row loader: 66 pixels x 4 chunks of 16 B; thread t takes chunk t, and every wave also
    // takes the last 8 chunks (identical bytes to identical addresses: keeps the step branch-free)
    const int ldOff = kHrA4 + (tid >> 2) * kHrStride32 + (tid & 3) * 16;
    const int ldPixX = min(64 + (lane >> 2), kHrRowPix - 1);
    const int ldOffX = kHrA4 + ldPixX * kHrStride32 + (lane & 3) * 16;
    const int ldSrc = (tid >> 2) * 64 + (tid & 3) * 16;         // byte offsets within an a4 row
    const int ldSrcX = ldPixX * 64 + (lane & 3) * 16;
    // conv6: lane group q holds the output row y with y mod 4 == q
    bool qIs[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) qIs[g] = q == g;

    const int perFrame = p.strips * p.segs;
    for (TileWalk tw(p.numItems); tw.valid(); tw.next())
    {
        const int f = tw.cur / perFrame;
        const int rem = tw.cur - f * perFrame;
        const int seg = rem / p.strips;
        const int X0 = (rem - seg * p.strips) * kHrStrip;
        const int Y0 = seg * p.segRows;                       // a multiple of 4
        const int Y1 = min(p.height2, Y0 + p.segRows);
        const int t0 = Y0 - 2;
        const int tLast = ((Y1 - 1) & ~3) + 10;               // the step that stores the last block of four rows

        const long rowBytes = long(p.a4.pitch) * 64;
        // the a4 row to fetch next (uniform): one buffer descriptor per row, lanes add their chunk offset
        const char* rowPtr = reinterpret_cast<const char*>(p.a4.base + f * p.a4.frame + (long(X0) - 3) * 32) + (Y0 - 3) * rowBytes;
        const auto loadRow = [&](int voff) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(rowPtr), 0, kHrRowPix * 64, 0x00020000);
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
            return uint4{v[0], v[1], v[2], v[3]};
        };

        // output rows at or beyond Y1 fall outside this descriptor and are dropped
        const __amdgpu_buffer_rsrc_t imageRsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.image + f * p.imageFrame, 0, int(long(Y1) * p.imagePitch), 0x00020000);
        const __amdgpu_buffer_rsrc_t inputRsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(p.input) + f * p.inputFrame, 0, int(p.inputFrame), 0x00020000);

        // conv6: lane (p16, q) finishes the pixel (row 4B + q, column col6) of a block of rows 4B .. 4B+3
        const int col6 = X0 + wave * 16 + p16;
        const bool laneOwns = wave * 16 + p16 < kHrStrip && col6 < p.width2;
        const unsigned laneOut = unsigned(q) * unsigned(p.imagePitch) + unsigned(col6) * 4u;
        // the residual term: NN2x(input) at that pixel = input pixel (2B + (q >> 1), col6 >> 1)
        const unsigned laneX0 = unsigned(q >> 1) * unsigned(p.inputPitch) + unsigned(min(col6, p.width2 - 1) >> 1) * 4u;
        // conv4 / conv5: are this lane's two columns (one per 16-pixel half) inside the image?
        const bool colInside0 = X0 + colOff >= 0 && X0 + colOff < p.width2;
        const bool colInside1 = X0 + colOff + 16 >= 0 && X0 + colOff + 16 < p.width2;
        // all 32 columns of this wave inside (wave-uniform): then only the rows can fall outside
        const int colFirst = X0 + (L ? -1 : -2) + ti * 32;
        const bool colsAllInside = colFirst >= 0 && colFirst + 31 < p.width2;

        __syncthreads();
        // rows Y0-3 .. Y0-1 go straight to LDS (slots of rows t0-1, t0, t0+1), row Y0 stays in flight
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
        {
            const int slot = (rr - 1) & 3;       // row t0 - 1 + rr
            *reinterpret_cast<uint4*>(smem + slot * kHrRow32 + ldOff) = loadRow(ldSrc);
            *reinterpret_cast<uint4*>(smem + slot * kHrRow32 + ldOffX) = loadRow(ldSrcX);
            rowPtr += rowBytes;
        }
        // Rows in flight in registers: step t issues the load of row t+3 into one register set and
        // parks the other (row t+2, loaded during step t-1), so a load has a whole step to land.
        uint4 rowA = loadRow(ldSrc), rowB = rowA;
        uint4 rowAX = loadRow(ldSrcX), rowBX = rowAX;
        rowPtr += rowBytes;
        unsigned xraw = 0;                        // raw input pixel of the next block to store
        // a 32-pixel x 32-output tile as four 16 x 16 accumulators: index = 2 * pixel half + output half
        f32x4 accA[4], accB[4];
        f32x4 acc6 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        float hold0 = 0.0f, hold1 = 0.0f, hold2 = 0.0f;
#pragma unroll
        for (int a = 0; a < 4; ++a) accA[a] = accB[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();

        // One step.  PH = (t - t0) & 3.
        // MASKED = 0: every value this wave finishes in the step lies inside the image (the common case), so the
        // zeroing of out-of-image activations is compiled out.
        const auto stepImpl = [&](auto phase, auto masked, const int t, uint4& park, uint4& parkX, uint4& load, uint4& loadX,
                                  const f32x4 (&accPrev)[4], f32x4 (&acc)[4]) {
            constexpr int PH = decltype(phase)::value;
            constexpr bool MASKED = decltype(masked)::value != 0;
            // ---- global loads: row t+3 (the last needed row is Y1+2: stop advancing there)
            load = loadRow(ldSrc);
            loadX = loadRow(ldSrcX);
            rowPtr += (t + 3 < Y1 + 2) ? rowBytes : 0;
            // the block of rows stored three steps from now starts at row t - 7 (phase 1 only)
            if constexpr (PH == 1)
                xraw = __builtin_amdgcn_raw_buffer_load_b32(inputRsrc, int(laneX0 + unsigned((t - 7) >> 1) * unsigned(p.inputPitch)), 0, 0);

            // ---- row bookkeeping of what this step finishes
            const int rowDone = t - 1 - lag;                         // conv4 / conv5 row being finished
            const bool rowInside = rowDone >= 0 && rowDone < p.height2;
            const bool inside0 = rowInside & colInside0, inside1 = rowInside & colInside1;

            // ---- LDS addresses of this step
            const char* in0 = smem + inSlot[(PH + 3) & 3] + inLane;  // rows rc-1, rc, rc+1 of the input ring
            const char* in1 = smem + inSlot[PH] + inLane;
            const char* in2 = smem + inSlot[(PH + 1) & 3] + inLane;
            const char* c6 = smem + ((PH + 3) & 3) * kHrRow6 + in6Lane;    // row t-5 of a6
            char* outDst = smem + outSlot[(PH + 3) & 3] + outLane;   // row rc-1 of the output ring

            // read R: filter tap R / 2 = (i, j) on the 16-pixel half R % 2 (all 32 channels: one K-step); 18.. = conv6
            const auto readFrag = [&](auto rr) -> f16x8 {
                constexpr int R = decltype(rr)::value;
                if constexpr (R < 18)
                {
                    constexpr int tap = R / 2, half = R % 2, i = tap / 3, j = tap % 3;
                    const char* b = i == 0 ? in0 : (i == 1 ? in1 : in2);
                    return *reinterpret_cast<const f16x8*>(b + (half * 16 + j) * kHrStride32);
                }
                else
                    return *reinterpret_cast<const f16x8*>(c6 + (R - 18) * kHrStride6);
            };

            // epilogue state carried between the pieces below
            f16x4 v0 = f16x4{0, 0, 0, 0}, v1 = v0, res = v0, sum = v0, xin = v0;
            unsigned c0 = 0, c1 = 0, c2 = 0;
            // accumulator a of the finished row: outputs 16 * (a & 1) + 4q .. + 3 of pixel 16 * (a >> 1) + p16
            const auto reluHalf4 = [&](int a) {
                f16x4 v = to_half4(accPrev[a][0], accPrev[a][1], accPrev[a][2], accPrev[a][3]);
                v = __builtin_elementwise_max(v, f16x4{0, 0, 0, 0});
                if constexpr (!MASKED) return v;
                return ((a >> 1) ? inside1 : inside0) ? v : f16x4{0, 0, 0, 0};
            };
            const auto store8 = [&](char* dst) {
                *reinterpret_cast<f16x8*>(dst) = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
            };
            // Piece K of the previous steps' epilogue, issued in the gap after MFMA K.
            const auto piece = [&](auto kk) {
                constexpr int K = decltype(kk)::value;
                constexpr int g = (PH + 3) & 3;      // conv6 group of the output row t-7, finished last step
                if constexpr (K == 0) v0 = reluHalf4(0);
                if constexpr (K == 1) v1 = reluHalf4(1);
                if constexpr (K == 2) { store8(outDst); v0 = reluHalf4(2); }
                if constexpr (K == 3) v1 = reluHalf4(3);
                if constexpr (K == 4) store8(outDst + 16 * outStride);
                if constexpr (K == 5)
                {
                    hold0 = qIs[g] ? acc6[0] : hold0;
                    hold1 = qIs[g] ? acc6[1] : hold1;
                    hold2 = qIs[g] ? acc6[2] : hold2;
                }
                if constexpr (K == 6)
                {
                    acc6[0] = qIs[g] ? 0.0f : acc6[0];
                    acc6[1] = qIs[g] ? 0.0f : acc6[1];
                    acc6[2] = qIs[g] ? 0.0f : acc6[2];
                }
                // every fourth step: the block of rows t-10 .. t-7 is complete in the holding registers;
                // lane (p16, q) converts and stores pixel (row t-10+q, column col6)
                if constexpr (PH == 0)
                {
                    if constexpr (K == 7)
                    {
                        // CSImageToTensor on the fly: UNORM8 -> u/255 -> fp16 (u * (1/255) rounds to the same fp16)
                        constexpr float k = 1.0f / 255.0f;
                        xin = to_half4(float(xraw & 0xff) * k, float((xraw >> 8) & 0xff) * k, float((xraw >> 16) & 0xff) * k, 0.0f);
                    }
                    if constexpr (K == 8) res = to_half4(hold0, hold1, hold2, 0.0f);
                    if constexpr (K == 9)
                        sum = to_half4((float)res[0] + (float)xin[0], (float)res[1] + (float)xin[1],
                                       (float)res[2] + (float)xin[2], 0.0f);
                    if constexpr (K == 10) c0 = unorm8((float)sum[0]);
                    if constexpr (K == 11) c1 = unorm8((float)sum[1]);
                    if constexpr (K == 12) c2 = unorm8((float)sum[2]);
                    if constexpr (K == 13)
                    {
                        // branch-free: lanes that own no pixel, and blocks above the segment, store at an
                        // offset beyond the descriptor's range, which the hardware drops
                        const int blk = t - 10;
                        const bool ok = laneOwns & (blk >= Y0);
                        const unsigned off = ok ? laneOut + unsigned(blk) * unsigned(p.imagePitch) : 0xffffffffu;
                        __builtin_amdgcn_raw_buffer_store_b32(c0 | (c1 << 8) | (c2 << 16) | 0xff000000u, imageRsrc, int(off), 0, 0);
                    }
                }
            };

            // ---- the MFMA stream: reads run kAhead in front; two epilogue pieces per K-step of four MFMAs
            f16x8 bq[kReads];
            constexpr int rho = (PH + 1) & 3;        // (t - 5) mod 4
            const auto fetch = [&](auto rr) {
                constexpr int R = decltype(rr)::value;
                if constexpr (R < kReads) bq[R] = readFrag(rr);
            };
            // K-step k = filter tap k of the conv4 / conv5 chain; the first takes the bias as its C operand
            const auto body = [&](auto kk) {
                constexpr int K = decltype(kk)::value;
                fetch(IC<2 * K + kAhead>{});
                fetch(IC<2 * K + kAhead + 1>{});
                if constexpr (K == 0)
                {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[0], bq[0], biasLo, 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[1], bq[0], biasHi, 0, 0, 0);
                    piece(IC<0>{});
                    if ((EXP & 1) == 0) __builtin_amdgcn_sched_barrier(0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[0], bq[1], biasLo, 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[1], bq[1], biasHi, 0, 0, 0);
                }
                else
                {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[2 * K], bq[2 * K], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[2 * K + 1], bq[2 * K], acc[1], 0, 0, 0);
                    piece(IC<2 * K>{});
                    if ((EXP & 1) == 0) __builtin_amdgcn_sched_barrier(0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[2 * K], bq[2 * K + 1], acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[2 * K + 1], bq[2 * K + 1], acc[3], 0, 0, 0);
                }
                piece(IC<2 * K + 1>{});
                if ((EXP & 1) == 0) __builtin_amdgcn_sched_barrier(0);
            };
            fetch(IC<0>{}); fetch(IC<1>{}); fetch(IC<2>{}); fetch(IC<3>{});
            if ((EXP & 1) == 0) __builtin_amdgcn_sched_barrier(0);
            body(IC<0>{}); body(IC<1>{}); body(IC<2>{}); body(IC<3>{}); body(IC<4>{});
            body(IC<5>{}); body(IC<6>{}); body(IC<7>{}); body(IC<8>{});
            acc6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w6[rho * 3 + 0], bq[18], acc6, 0, 0, 0);
            acc6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w6[rho * 3 + 1], bq[19], acc6, 0, 0, 0);
            acc6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w6[rho * 3 + 2], bq[20], acc6, 0, 0, 0);

            // ---- park row t+2
            *reinterpret_cast<uint4*>(smem + ((PH + 2) & 3) * kHrRow32 + ldOff) = park;
            *reinterpret_cast<uint4*>(smem + ((PH + 2) & 3) * kHrRow32 + ldOffX) = parkX;
            barrierLds();
        };

        const auto step = [&](auto phase, const int t, uint4& park, uint4& parkX, uint4& load, uint4& loadX,
                              const f32x4 (&accPrev)[4], f32x4 (&acc)[4]) {
            const int rowDone = t - 1 - lag;
            if (colsAllInside && rowDone >= 0 && rowDone < p.height2)
                stepImpl(phase, IC<0>{}, t, park, parkX, load, loadX, accPrev, acc);
            else
                stepImpl(phase, IC<1>{}, t, park, parkX, load, loadX, accPrev, acc);
        };

        // steps t0 .. tLast + 3; the three steps after the last store touch only ring rows nothing
        // reads and store nothing
        for (int t = t0; t <= tLast; t += 4)
        {
            step(IC<0>{}, t, rowA, rowAX, rowB, rowBX, accA, accB);
            step(IC<1>{}, t + 1, rowB, rowBX, rowA, rowAX, accB, accA);
            step(IC<2>{}, t + 2, rowA, rowAX, rowB, rowBX, accA, accB);
            step(IC<3>{}, t + 3, rowB, rowBX, rowA, rowAX, accB, accA);
        }
    }
    if (p.clock && tid == 0)
    {
        p.clock[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
        p.clock[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - ref0;
    }
}

}  // namespace

StripPlan PlanHrFused(int width2, int height2, int batch, int numCUs)
{
    StripPlan pl;
    if (width2 <= 0 || height2 <= 0 || batch <= 0 || numCUs <= 0) return pl;
    pl.strips = (width2 + kHrStrip - 1) / kHrStrip;
    // Rows are cut into segments where that fills the CUs better (two workgroups share a CU): every
    // segment re-computes 16 warm-up rows, so segments are at least 32 rows tall, and they start at
    // multiples of four rows (conv6 stores blocks of four).  What counts is rounds x steps.
    const int maxSegs = height2 / 32 > 1 ? (height2 / 32 < 64 ? height2 / 32 : 64) : 1;
    long bestCost = 0;
    for (int segs = 1; segs <= maxSegs; ++segs)
    {
        StripPlan c = pl;
        c.segRows = ((height2 + segs - 1) / segs + 3) & ~3;
        c.segs = (height2 + c.segRows - 1) / c.segRows;
        c.items = c.strips * c.segs * batch;
        c.steps = c.segRows + 16;
        const int slots = gridFor(c.items, numCUs);
        c.rounds = (c.items + slots - 1) / slots;
        // a workgroup that has its CU to itself takes about 0.72 of the time of one of a pair
        const int last = c.items - (c.rounds - 1) * slots;
        c.cost = (100L * (c.rounds - 1) + (2 * last <= slots ? 72 : 100)) * c.steps;
        if (segs == 1 || c.cost < bestCost)
        {
            bestCost = c.cost;
            pl = c;
        }
    }
    return pl;
}

hipError_t PrepareHrFused()
{
    return ensureDynamicLds(reinterpret_cast<const void*>(HrFusedKernel<0>), kHrSmem);
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
    const StripPlan plan = PlanHrFused(l.width2, l.height2, l.batch, numCUs);
    p.strips = plan.strips;
    p.segs = plan.segs;
    p.segRows = plan.segRows;
    p.numItems = plan.items;
    if (p.numItems <= 0) return hipErrorInvalidValue;
    // one input and one output frame must be addressable with a 32-bit byte offset
    if (p.imageFrame >= (1L << 31) || p.inputFrame >= (1L << 31)) return hipErrorInvalidValue;

    const char* env = getenv("SRX_EXP");
    const int exp = env ? atoi(env) : 0;
    const dim3 grid(gridFor(p.numItems, numCUs)), block(kThreads);
#define SRX_HR_CASE(E)                                                                                       \
    case E: {                                                                                                \
        const hipError_t e = ensureDynamicLds(reinterpret_cast<const void*>(HrFusedKernel<E>), kHrSmem);     \
        if (e != hipSuccess) return e;                                                                       \
        hipLaunchKernelGGL(HrFusedKernel<E>, grid, block, kHrSmem, s, p);                                    \
        break;                                                                                               \
    }
    ClockProbe* const probe = clockProbeFor(PROBE_HR_FUSED, int(grid.x), l.mayAllocate);
    p.clock = probe ? probe->dev : nullptr;
    switch (exp)
    {
        SRX_HR_CASE(1) SRX_HR_CASE(2)
    default:
        SRX_HR_CASE(0)
    }
#undef SRX_HR_CASE
    clockProbeReport(probe, "HrFusedKernel", int(grid.x), s);
    return hipGetLastError();
}

}  // namespace srx
