// gfx950 kernel: conv4 -> conv5 -> conv6 -> + NN2x(input) -> UNORM8, fused.
//
// A workgroup of eight waves (one per CU, 101 KB LDS) walks a strip of kHrWideStrip = 124 output columns
// downwards, one row per step: waves 0-3 (group 0) own its left 60 columns, waves 4-7 (group 1) the
// right 64, with rolling windows of the two intermediate activations in LDS rings, so neither
// intermediate touches HBM and no row is computed twice vertically.  Each group computes 64 columns of
// every layer.  On its own a group keeps 60 of them (each 3x3 layer costs a column on either side); here
// group 1's windows begin inside group 0's columns, so the activations left of its first pixel are the
// ones group 0 stored half a step earlier and all 64 of its output columns are valid: 124 of 128
// computed columns are kept instead of 120 (see the ring layout at kHrWideRowPix).  The layout of two
// independent strips of kHrStrip = 60 columns per workgroup remains as the template's other
// instantiation (diagnostic builds compare the two; outputs are the same bytes).  Wave w of a group accumulates,
// per step, its 32-pixel half (w & 1) of one row of conv4 (w < 2) or conv5 (w >= 2) as four 16 x 16
// accumulators with 36 v_mfma_f32_16x16x32_f16 (one K-step per filter tap: all 32 input channels),
// and its 16-pixel quarter of conv6 with 3 more: every wave issues the same matrix work.
//
// Activation fragments live in registers for the three steps that use their row (as filter row 2,
// then 1, then 0), so a step fetches ONE new input row, six fragments of 1 KiB, instead of three.
// conv5 waves read theirs from the a5 ring in LDS.  conv4 waves take theirs straight from conv_up1's
// output in memory: a fragment (16 pixels x 32 channels) is 1 KiB of consecutive bytes there, 16 B per
// lane, and the request goes out one step early into the register set that is idle (four sets, three
// in use), so there is no copy of conv_up1's rows in LDS at all.
//
// The two groups alternate.  A step of a group has a LOAD phase (the new row's fragments are
// requested, the previous step's accumulators are converted and stored to the rings, finished
// conv6 rows are moved to holding registers and, every fourth step, converted and written to the
// image) and a MATRIX phase (the 39 MFMAs and the six LDS reads of conv6's operands, nothing else).
// Waves w and w + 4 share a SIMD and its one matrix pipe; group 1 runs half a step behind group 0,
// so while one wave of a SIMD streams MFMAs its partner does its vector, LDS-store and memory work,
// and the other way round after the next barrier (one workgroup barrier per phase; the one that ends a
// load phase does not wait for LDS, see phaseBarrierNoWait).  The matrix pipe then idles only for the
// barrier itself, not for LDS latency or epilogues, and the two strips of a CU advance in lock step
// instead of one workgroup starving the other.  A wave in its matrix phase issues nothing but MFMAs, so
// every vector instruction of its partner's load phase takes an issue slot from that stream: the load
// phase is kept as short in instructions as it can be.
//
// conv6 (32 -> 3 channels) streams input rows.  The 16 output rows of its MFMA are four
// groups of (R, G, B, 0); group g accumulates the output row y with y mod 4 == g.  The
// newest row R of conv5's output adds filter row 2 to output row R-1, filter row 1 to R
// and filter row 0 to R+1 in ONE MFMA per horizontal tap (filter fragments per R mod 4:
// PackConv6Rotating; the fragments of R mod 4 == rho are those of rho == 0 with the 16 rows
// rotated by 4 rho, so LDS holds the three of rho == 0 and a lane reads its rotated row), so conv6
// costs 3 MFMAs per step instead of 9 and reads each activation fragment once.  A finished row is
// moved to a holding register and its group cleared; every fourth step all 64 lanes convert and
// store a block of four rows.
//
// Rows per step t of a group:
//     accumulate: conv4 row t      conv5 row t-3     conv6 input row t-5
//     finish:     conv4 row t-1    conv5 row t-4     conv6 output row t-7 -> holding register
// The loop is unrolled by four so every ring slot and conv6 group is a compile-time
// constant; segments start at multiples of four rows.
//
// Column bookkeeping for a strip whose first output column is X0, group G (G = 0 throughout in the layout of
// independent 60-column strips):
//   conv6 pixel i     = column X0 + 60 G + i  (G = 0: 60 of 64 kept; G = 1: all 64), reads a6 position 60 G + i ..+2
//   a6 position 62 G + i = column X0 - 1 + 62 G + i = conv5 pixel i, which reads a5 position 62 G + i ..+2
//   a5 position 64 G + i = column X0 - 2 + 64 G + i = conv4 pixel i, which reads a4 index i..i+2 of its group
//   a4 index i        = column X0 - 3 + 64 G + i   (66 pixels of conv_up1's row, read from memory)
// Activations outside the true image are written as zeros: that is the zero
// padding each of the reference's convolutions applies to its own input
// (Content/SuperResolution.cpp:245-268 via ML::Util::CreateConvolutionLayer); the
// residual add and the conversion are ELEMENT_WISE_ADD (Content/SuperResolution.cpp:700-702)
// and PSTensorToImage.hlsl:14-35 + the R8G8B8A8_UNORM render target.
#include "srx_device.h"

#include <type_traits>

namespace srx {

using namespace dev;

namespace {

constexpr int kHrThreads = 512;
constexpr int kHrStrip = 60;
constexpr int kHrRowPix = 66;
constexpr int kHrStride = 96;                         // pixel stride of all three rings: 64 B of channels + 32 B, which makes the
                                                      // 16x16x32 B-operand reads (lane = pixel l & 15, channels 8 * (l >> 4) ..) conflict free
constexpr int kHrRow = kHrRowPix * kHrStride;         // 6336
constexpr int kHrA5 = 0;                              // per group: two rings of four rows (conv4's and conv5's outputs)
constexpr int kHrGroupSmem = 8 * kHrRow;              // 50688
constexpr int kHrW6 = 2 * kHrGroupSmem;               // conv6 filter fragments of rho == 0: [tap j][lane][16 B]
constexpr int kHrSmem = kHrW6 + 3 * 1024;             // 104448
// Seam-sharing layout (WIDE): the two groups are the halves of ONE strip of kHrWideStrip output columns and share
// their rings, one row of kHrWideRowPix pixel positions per ring slot:
//   a5 positions 0..63 = group 0's conv4 pixels, 64..127 = group 1's: columns X0 - 2 .. X0 + 125, all valid
//   a6 positions 0..61 = group 0's valid conv5 pixels, 62..125 = group 1's: columns X0 - 1 .. X0 + 124
// Group 1's conv5 and conv6 windows start two positions left of its own first pixel, inside group 0's part (conv5
// pixel i of group 1 reads a5 positions 62 + i .., conv6 pixel i reads a6 positions 60 + i ..), so all 64 of its
// output columns are valid: 60 + 64 = 124 of 128 computed columns are kept.  Group 0's two invalid conv5 pixels
// would land on group 1's first two: they are stored to the unused positions 126, 127 instead.
constexpr int kHrWideStrip = 124;
constexpr int kHrWideRowPix = 130;
constexpr int kHrWideRow = kHrWideRowPix * kHrStride;  // 12480
constexpr int kHrWideW6 = 8 * kHrWideRow;              // 99840
constexpr int kHrWideSmem = kHrWideW6 + 3 * 1024;      // 102912

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

// The barrier between a group's load phase and its matrix phase only keeps the two groups in step: what a load
// phase stores to LDS is first read after the NEXT barrier (phaseBarrier, which waits for LDS), and what it reads
// from LDS the matrix phase waits for where it uses it.  So this one does not wait for LDS: the stores drain and
// the reads land while the MFMAs run.
__device__ __forceinline__ void phaseBarrierNoWait()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int N>
using IC = std::integral_constant<int, N>;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// The barrier between two phases.  LDS traffic only is ordered (see barrierLds); the scheduling fences keep the
// compiler from moving matrix or vector instructions from one phase into the other.
__device__ __forceinline__ void phaseBarrier()
{
    __builtin_amdgcn_sched_barrier(0);
    barrierLds();
    __builtin_amdgcn_sched_barrier(0);
}

// EXP (variants of SRX_DIAG builds, env SRX_EXP): 1 = no scheduling pins inside the phases; 2 = no raised priority
// in the matrix phase; 4 = with SRX_CLOCK, the first 16 workgroups also record the cycles each wave spends in the
// load phase, waiting at its barrier, in the matrix phase and waiting at that barrier.  The shipping library
// instantiates EXP = 0 only.
// IN3 / OUT3: the input / output image holds 3 bytes per pixel (RGB8) instead of 4.  An input pixel is then fetched
// as the unaligned 32-bit word that starts at its first byte (the descriptor covers one byte more than the frame for
// the last pixel's word); an output pixel is stored as an unaligned 16-bit word and a byte.
template <int EXP, bool WIDE, bool IN3 = false, bool OUT3 = false>
__global__ __launch_bounds__(kHrThreads, 2) void HrFusedKernel(const HrParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smemAll[];
    constexpr bool PIN = (EXP & 1) == 0;
    constexpr int kRow = WIDE ? kHrWideRow : kHrRow;           // bytes of one ring slot
    constexpr int kA6 = 4 * kRow;                              // the a6 ring follows the four slots of a5
    constexpr int kW6 = WIDE ? kHrWideW6 : kHrW6;
    constexpr int kStrip = WIDE ? kHrWideStrip : kHrStrip;     // output columns per work item

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int G = wave >> 2;                      // group = strip of the pair
    const int w4 = wave & 3;                      // wave within the group
    const int p16 = lane & 15, q = lane >> 4;     // 16x16x32 fragment coordinates: pixel (or output) within 16, group of 8 channels
    const int L = w4 >> 1, ti = w4 & 1;
    char* const smem = smemAll + (WIDE ? 0 : G * kHrGroupSmem);
    // WIDE: first ring position of this group's conv4 pixels (= a5), of its conv5 pixels (= a6; also the a5 position
    // its conv5 window starts at) and the a6 position its conv6 window starts at
    const int pos4 = WIDE ? 64 * G : 0, pos5 = WIDE ? 62 * G : 0, pos6 = WIDE ? 60 * G : 0;

    unsigned long long clk0 = 0, ref0 = 0;
    if constexpr (SRX_DIAG != 0)
    {
        clk0 = __builtin_amdgcn_s_memtime();
        ref0 = __builtin_amdgcn_s_memrealtime();
    }
    f16x8 wmain[18];
    {
        const f16x8* src = L ? p.wpk5 : p.wpk4;
#pragma unroll
        for (int s = 0; s < 18; ++s) wmain[s] = src[s * 64 + lane];
        // make the loads land here, not behind vmcnt waits inside the step loop
#pragma unroll
        for (int s = 0; s < 18; ++s) asm volatile("" ::"v"(wmain[s]));
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

    // never-written pad columns of the rings must hold finite values
    for (int i = tid * 16; i < kW6; i += kHrThreads * 16) *reinterpret_cast<uint4*>(smemAll + i) = uint4{0, 0, 0, 0};
    if (tid < 192) reinterpret_cast<uint4*>(smemAll + kW6)[tid] = reinterpret_cast<const uint4*>(p.wpk6)[tid];
    // conv6 filter fragment of (rho, tap j) for this lane: row (p16 - 4 rho) & 15 of the rho == 0 fragment
    const auto w6Lane = [&](int rho) { return kW6 + ((((p16 - 4 * rho) & 15) + 16 * q) * 16); };

    // Ring slots.  Row r of an item lives in slot (r - t0) & 3, t0 = the item's first step.  With
    // the loop unrolled by four the phase PH = (t - t0) & 3 is a compile-time constant, and so is
    // every slot except for the layer-dependent lag (conv5 trails conv4 by 3 rows = 1 slot): that
    // one rotation is folded into these tables, indexed by a compile-time (PH + offset) & 3.
    const int rot = L ? 1 : 0;                     // (-lag) & 3
    const int inRing = kHrA5;                       // conv5's input; conv4 takes its input straight from HBM / L2
    const int outRing = L ? kA6 : kHrA5;
    int inSlot[4], outSlot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        inSlot[k] = inRing + ((k + rot) & 3) * kRow;
        outSlot[k] = outRing + ((k + rot) & 3) * kRow;
    }
    const int lag = L ? 3 : 0;
    const int colOff = (L ? -1 + pos5 : -2 + pos4) + ti * 32 + p16;  // column of this lane's pixel less the item's X0; + 16 for the second pixel half
    const int inLane = (pos5 + ti * 32 + p16) * kHrStride + q * 16;
    const int in6Lane = kA6 + (pos6 + w4 * 16 + p16) * kHrStride + q * 16;
    const int outLane = ((L ? pos5 : pos4) + ti * 32 + p16) * kHrStride + q * 16;
    // distance from the store of the first 16-pixel half to that of the second.  WIDE: group 0's conv5 pixels 62
    // and 63 are not valid and their positions belong to group 1: those two lanes store to positions 126, 127.
    int half2Off = 16 * kHrStride;
    if (WIDE && G == 0 && L == 1 && ti == 1 && p16 >= 14) half2Off += 64 * kHrStride;
    asm volatile("" : "+v"(half2Off));
    // this lane's store address in each slot of the output ring, kept in registers (made opaque: the compiler
    // would otherwise keep the lane part and add the slot in every step, one more vector instruction)
    int outAddr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        outAddr[k] = (WIDE ? 0 : G * kHrGroupSmem) + outSlot[k] + outLane;
        asm volatile("" : "+v"(outAddr[k]));
    }
    // conv4's activation fragments come straight from conv_up1's output in HBM / L2: fragment (tap j, pixel half)
    // of a row is 16 pixels x 64 B = 1 KiB of consecutive bytes, 16 B per lane.  Byte offset of this lane's part
    // of fragment (0, 0) within the 66 pixels of the strip's row:
    const int a4Lane = (ti * 32 + p16) * 64 + q * 16;
    // conv6: lane group q holds the output row y with y mod 4 == q
    unsigned long long qMask[4];                  // lane masks of the four groups, for the swaps of the load phase
#pragma unroll
    for (int g = 0; g < 4; ++g) qMask[g] = __builtin_amdgcn_ballot_w64(q == g);

    const int perFrame = p.strips * p.segs;
    // iterations of four steps an item takes: steps t0 = Y0 - 2 .. tLast + 3, tLast = the step that stores the last
    // block of four rows
    const auto itersOf = [&](int item) {
        const int rem = item % perFrame;
        const int Y0 = (rem / p.strips) * p.segRows;
        const int Y1 = min(p.height2, Y0 + p.segRows);
        return ((((Y1 - 1) & ~3) + 12 - Y0) >> 2) + 1;
    };
    // A workgroup takes two items of kHrStrip columns, one per group, or (WIDE) one item of kHrWideStrip columns.
    const int numPairs = WIDE ? p.numItems : (p.numItems + 1) >> 1;
    for (TileWalk tw(numPairs); tw.valid(); tw.next())
    {
        // the pair's items; an odd item count leaves the last group 1 a copy of the last item that stores nothing
        const int itemRaw = WIDE ? tw.cur : 2 * tw.cur + G;
        const bool itemValid = itemRaw < p.numItems;
        const int item = itemValid ? itemRaw : p.numItems - 1;
        const int iters = WIDE ? itersOf(item)
                               : max(itersOf(min(2 * tw.cur, p.numItems - 1)), itersOf(min(2 * tw.cur + 1, p.numItems - 1)));
        const int f = item / perFrame;
        const int rem = item - f * perFrame;
        const int seg = rem / p.strips;
        const int X0 = (rem - seg * p.strips) * kStrip;
        const int Y0 = seg * p.segRows;                       // a multiple of 4
        const int Y1 = min(p.height2, Y0 + p.segRows);
        const int t0 = Y0 - 2;

        const long rowBytes = long(p.a4.pitch) * 64;
        // the a4 row to fetch next (uniform): one buffer descriptor per row covering the strip's 66 pixels
        const int a4Col = X0 - 3 + pos4;                      // column of a4 index 0 of this group
        const char* rowPtr = reinterpret_cast<const char*>(p.a4.base + f * p.a4.frame + long(a4Col) * 32) + (Y0 - 3) * rowBytes;
        // bytes of a row from that column to the end of the row's memory (4 border columns precede column 0); a
        // group of a wide item can begin beyond it, and the descriptor then makes its loads return zeros
        const int a4RowBytes = max(0, min(kHrRowPix, p.a4.pitch - 4 - a4Col)) * 64;

        // output rows at or beyond Y1 fall outside this descriptor and are dropped
        const __amdgpu_buffer_rsrc_t imageRsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.image + f * p.imageFrame, 0, int(long(Y1) * p.imagePitch), 0x00020000);
        const __amdgpu_buffer_rsrc_t inputRsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(p.input) + f * p.inputFrame, 0, int(p.inputFrame) + (IN3 ? 1 : 0), 0x00020000);

        // conv6: lane (p16, q) finishes the pixel (row 4B + q, column col6) of a block of rows 4B .. 4B+3
        const int col6 = X0 + pos6 + w4 * 16 + p16;
        const bool laneOwns = itemValid && ((WIDE && G == 1) || w4 * 16 + p16 < kHrStrip) && col6 < p.width2;
        const unsigned laneOut = unsigned(q) * unsigned(p.imagePitch) + unsigned(col6) * (OUT3 ? 3u : 4u);
        // the residual term: NN2x(input) at that pixel = input pixel (2B + (q >> 1), col6 >> 1)
        const unsigned laneX0 = unsigned(q >> 1) * unsigned(p.inputPitch) + unsigned(min(col6, p.width2 - 1) >> 1) * (IN3 ? 3u : 4u);
        // conv4 / conv5: are this lane's two columns (one per 16-pixel half) inside the image?
        const bool colInside0 = X0 + colOff >= 0 && X0 + colOff < p.width2;
        const bool colInside1 = X0 + colOff + 16 >= 0 && X0 + colOff + 16 < p.width2;
        // all 32 columns of this wave inside (wave-uniform): then only the rows can fall outside
        const int colFirst = X0 + (L ? -1 + pos5 : -2 + pos4) + ti * 32;
        const bool colsAllInside = colFirst >= 0 && colFirst + 31 < p.width2;

        __syncthreads();
        unsigned xraw = 0;                        // raw input pixel of the next block to store
        // a 32-pixel x 32-output tile as four 16 x 16 accumulators: index = 2 * pixel half + output half.  One
        // set is enough: a step's load phase converts and stores it before the matrix phase starts it anew.
        f32x4 acc[4];
        f32x4 acc6 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        float hold0 = 0.0f, hold1 = 0.0f, hold2 = 0.0f;
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();
        // Activation fragments stay in registers for the three steps that use their row (as filter row 2, 1,
        // 0): fr[s] holds the row of ring slot s, [tap j * 2 + pixel half]; a step fetches only one new row.
        // conv5 waves read it from the a5 ring in the step that first uses it.  conv4 waves load theirs from
        // memory one step earlier, into the set that is idle (four sets, three in use), so the load has a
        // whole step to land.  The rows the first step needs besides are fetched here.
        f16x8 fr[4][6];
        const auto fetchRow = [&](auto set, const char* rowBase) {
            constexpr int S = decltype(set)::value;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int half = 0; half < 2; ++half)
                    fr[S][j * 2 + half] = *reinterpret_cast<const f16x8*>(rowBase + (half * 16 + j) * kHrStride);
        };
        const auto loadA4Row = [&](auto set) {
            constexpr int S = decltype(set)::value;
            // (the size is uniform, but with the scalar registers all in use the compiler keeps it in a vector
            // register, and a descriptor built from one is applied lane by lane unless it is told otherwise)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(rowPtr), 0,
                                                                                __builtin_amdgcn_readfirstlane(a4RowBytes), 0x00020000);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int half = 0; half < 2; ++half)
                {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, a4Lane + (half * 16 + j) * 64, 0, 0);
                    fr[S][j * 2 + half] = __builtin_bit_cast(f16x8, v);
                }
        };
        if (L == 0)
        {
            // rows Y0-3, Y0-2, Y0-1 = the three input rows of the first step (t0 = Y0 - 2)
            loadA4Row(IC<3>{}); rowPtr += rowBytes;
            loadA4Row(IC<0>{}); rowPtr += rowBytes;
            loadA4Row(IC<1>{}); rowPtr += rowBytes;
        }
        else
        {
            fetchRow(IC<3>{}, smem + inSlot[3] + inLane);
            fetchRow(IC<0>{}, smem + inSlot[0] + inLane);
        }

        unsigned long long spent[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // One step = load phase, barrier, matrix phase, barrier.  PH = (t - t0) & 3.
        const auto step = [&](auto phase, const int t) {
            constexpr int PH = decltype(phase)::value;
            constexpr int rho = (PH + 1) & 3;        // (t - 5) mod 4

            // ---- LDS addresses of this step
            const char* in2 = smem + inSlot[(PH + 1) & 3] + inLane;        // the newest row (rc+1) of the input ring
            const char* c6 = smem + ((PH + 3) & 3) * kRow + in6Lane;       // row t-5 of a6
            char* outDst = smemAll + outAddr[(PH + 3) & 3];               // row rc-1 of the output ring

            unsigned long long stamp = 0;
            const auto lap = [&](int k) {
                if constexpr ((EXP & 4) != 0)
                {
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned long long now = __builtin_amdgcn_s_memtime();
                    if (k >= 0) spent[k] += now - stamp;
                    stamp = now;
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            lap(-1);
            // ================= load phase =================
            // the step's new input row (six fragments): conv5 reads row rc+1 of the a5 ring for this step, conv4
            // requests row t+2 of conv_up1's output for the next one (the last needed row is Y1+2: stop advancing
            // there)
            if (L == 0)
            {
                loadA4Row(IC<(PH + 2) & 3>{});
                rowPtr += (t + 2 < Y1 + 2) ? rowBytes : 0;
            }
            else fetchRow(IC<(PH + 1) & 3>{}, in2);
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
            // the block of rows stored three steps from now starts at row t - 7 (phase 1 only)
            if constexpr (PH == 1)
                xraw = __builtin_amdgcn_raw_buffer_load_b32(inputRsrc, int(laneX0 + unsigned((t - 7) >> 1) * unsigned(p.inputPitch)), 0, 0);
            lap(0);

            // the previous step's epilogue: ReLU, fp16, one 16-byte store per 16-pixel half
            const int rowDone = t - 1 - lag;                         // conv4 / conv5 row being finished
            const bool rowInside = rowDone >= 0 && rowDone < p.height2;
            // accumulator a of the finished row: outputs 8q + 4 * (a & 1) .. + 3 of pixel 16 * (a >> 1) + p16
            f16x4 fin[4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
                fin[a] = __builtin_elementwise_max(to_half4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]), f16x4{0, 0, 0, 0});
            // Activations outside the image are zeros.  In the common case every value this wave finishes lies
            // inside: the zeroing sits behind a wave-uniform branch (the empty asm keeps the compiler from turning
            // it into eight selects that always run).
            if (!(colsAllInside && rowInside))
            {
                asm volatile("" ::: "memory");
                const bool inside0 = rowInside & colInside0, inside1 = rowInside & colInside1;
#pragma unroll
                for (int a = 0; a < 4; ++a) fin[a] = ((a >> 1) ? inside1 : inside0) ? fin[a] : f16x4{0, 0, 0, 0};
            }
            *reinterpret_cast<f16x8*>(outDst) = __builtin_shufflevector(fin[0], fin[1], 0, 1, 2, 3, 4, 5, 6, 7);
            *reinterpret_cast<f16x8*>(outDst + (WIDE ? half2Off : 16 * kHrStride)) = __builtin_shufflevector(fin[2], fin[3], 0, 1, 2, 3, 4, 5, 6, 7);
            lap(1);

            // conv6: the output row t-7 was finished by the last matrix phase; its group is (PH + 3) & 3
            {
                constexpr int g = (PH + 3) & 3;
                // Taken and cleared in place with three swaps under the lane mask of group g: the holding
                // registers of those lanes are zero (cleared after each block store), so the swap moves the
                // finished row out and zeros into the accumulator.  All lanes are active here (full
                // workgroup, uniform control flow), so EXEC is restored to all ones.  Written as C++ selects
                // this is six selects and, because the compiler postpones the taking to the step that converts
                // the block and keeps the old accumulator alive for it, three register copies per step.
                asm volatile("s_mov_b64 exec, %6\n\tv_swap_b32 %0, %3\n\tv_swap_b32 %1, %4\n\tv_swap_b32 %2, %5\n\ts_mov_b64 exec, -1"
                             : "+v"(hold0), "+v"(hold1), "+v"(hold2), "+v"(acc6[0]), "+v"(acc6[1]), "+v"(acc6[2])
                             : "s"(qMask[g]));
            }
            // every fourth step: the block of rows t-10 .. t-7 is complete in the holding registers;
            // lane (p16, q) converts and stores pixel (row t-10+q, column col6)
            if constexpr (PH == 0)
            {
                // CSImageToTensor on the fly: UNORM8 -> u/255 -> fp16 (u * (1/255) rounds to the same fp16)
                constexpr float k = 1.0f / 255.0f;
                const f16x4 xin = to_half4(float(xraw & 0xff) * k, float((xraw >> 8) & 0xff) * k, float((xraw >> 16) & 0xff) * k, 0.0f);
                const f16x4 res = to_half4(hold0, hold1, hold2, 0.0f);
                const f16x4 sum = to_half4((float)res[0] + (float)xin[0], (float)res[1] + (float)xin[1],
                                           (float)res[2] + (float)xin[2], 0.0f);
                const unsigned c0 = unorm8((float)sum[0]), c1 = unorm8((float)sum[1]), c2 = unorm8((float)sum[2]);
                // branch-free: lanes that own no pixel, and blocks above the segment, store at an
                // offset beyond the descriptor's range, which the hardware drops
                const int blk = t - 10;
                const bool ok = laneOwns & (blk >= Y0);
                // (an offset that stays beyond the range with the 2 added for the third byte of an RGB8 pixel)
                const unsigned off = ok ? laneOut + unsigned(blk) * unsigned(p.imagePitch) : 0xc0000000u;
                if constexpr (OUT3)
                {
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(c0 | (c1 << 8)), imageRsrc, int(off), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)c2, imageRsrc, int(off + 2u), 0, 0);
                }
                else __builtin_amdgcn_raw_buffer_store_b32(c0 | (c1 << 8) | (c2 << 16) | 0xff000000u, imageRsrc, int(off), 0, 0);
                hold0 = hold1 = hold2 = 0.0f;         // what the swaps above put into the accumulator
            }
            lap(2);
            if constexpr ((EXP & 4) != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lap(3);
            phaseBarrierNoWait();
            lap(4);

            // ================= matrix phase =================
            if constexpr ((EXP & 2) == 0) __builtin_amdgcn_s_setprio(1);
            // conv6's operands are read here: they are needed after 24 MFMAs of the main chain
            f16x8 w6r[3], b6[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) w6r[j] = *reinterpret_cast<const f16x8*>(smemAll + w6Lane(rho) + j * 1024);
#pragma unroll
            for (int j = 0; j < 3; ++j) b6[j] = *reinterpret_cast<const f16x8*>(c6 + j * kHrStride);
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
            // K-step k = filter tap k = (i, j) of the conv4 / conv5 chain on input row rc-1+i; the first takes the
            // bias as its C operand
            const auto body = [&](auto kk) {
                constexpr int K = decltype(kk)::value;
                constexpr int i = K / 3, j = K % 3, S = (PH + 3 + i) & 3;
                const f16x8 b0 = fr[S][j * 2], b1 = fr[S][j * 2 + 1];
                if constexpr (K == 0)
                {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[0], b0, biasLo, 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[1], b0, biasHi, 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[0], b1, biasLo, 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[1], b1, biasHi, 0, 0, 0);
                }
                else
                {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[2 * K], b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[2 * K + 1], b0, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[2 * K], b1, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wmain[2 * K + 1], b1, acc[3], 0, 0, 0);
                }
            };
            // conv6's three MFMAs go between the last K-steps: at the very end each would wait for the one before
            body(IC<0>{}); body(IC<1>{}); body(IC<2>{}); body(IC<3>{}); body(IC<4>{}); body(IC<5>{});
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
            acc6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w6r[0], b6[0], acc6, 0, 0, 0);
            body(IC<6>{});
            acc6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w6r[1], b6[1], acc6, 0, 0, 0);
            body(IC<7>{});
            acc6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w6r[2], b6[2], acc6, 0, 0, 0);
            body(IC<8>{});
            if constexpr ((EXP & 2) == 0) __builtin_amdgcn_s_setprio(0);
            lap(5);
            phaseBarrier();
            lap(6);
        };

        // Group 1 runs one phase behind group 0: its load phases coincide with group 0's matrix phases.
        if (G == 1) phaseBarrier();
        int t = t0;
        for (int it = 0; it < iters; ++it, t += 4)
        {
            step(IC<0>{}, t);
            step(IC<1>{}, t + 1);
            step(IC<2>{}, t + 2);
            step(IC<3>{}, t + 3);
        }
        if (G == 0) phaseBarrier();
        if constexpr ((EXP & 4) != 0)
            if (p.clock && lane == 0 && blockIdx.x < 16)
                for (int k = 0; k < 8; ++k) p.clock[2 * gridDim.x + (blockIdx.x * 8 + wave) * 8 + k] = spent[k];
    }
    if constexpr (SRX_DIAG != 0)
        if (p.clock && tid == 0)
        {
            p.clock[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
            p.clock[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - ref0;
        }
}

}  // namespace

// The layout the library runs: one item of kHrWideStrip columns per workgroup (true) or two of kHrStrip (false).
constexpr bool kHrWide = true;
#if SRX_DIAG
namespace {
int hrDiagExp()
{
    static const int exp = diagEnvInt("SRX_EXP", 0);
    return exp;
}
bool hrWide()
{
    static const bool wide = diagEnvInt("SRX_HR_WIDE", kHrWide ? 1 : 0) != 0;      // the other layout, for A/B measurements
    return wide;
}
}  // namespace
#else
namespace {
constexpr bool hrWide() { return kHrWide; }
}  // namespace
#endif

StripPlan PlanHrFused(int width2, int height2, int batch, int numCUs)
{
    StripPlan pl;
    if (width2 <= 0 || height2 <= 0 || batch <= 0 || numCUs <= 0) return pl;
    const bool wide = hrWide();
    const int strip = wide ? kHrWideStrip : kHrStrip;
    pl.strips = (width2 + strip - 1) / strip;
    // Rows are cut into segments where that fills the CUs better (a workgroup takes one wide item or two narrow
    // ones, one workgroup per CU): every segment re-computes 16 warm-up rows, so segments are at least 16 rows tall
    // (small images then use many more CUs than they have strips), and they start at multiples of four rows (conv6
    // stores blocks of four).  What counts is rounds x steps.
    const int maxSegs = height2 / 8 > 1 ? (height2 / 8 < 128 ? height2 / 8 : 128) : 1;
    long bestCost = 0;
#if SRX_DIAG
    static const int forced = diagEnvInt("SRX_SEGS_HR", 0);      // a fixed number of row segments, for measurements
#else
    constexpr int forced = 0;
#endif
    for (int segs = forced > 0 ? forced : 1; segs <= (forced > 0 ? forced : maxSegs); ++segs)
    {
        StripPlan c = pl;
        c.segRows = ((height2 + segs - 1) / segs + 3) & ~3;
        c.segs = (height2 + c.segRows - 1) / c.segRows;
        c.items = c.strips * c.segs * batch;
        c.steps = c.segRows + 16;
        const int pairs = wide ? c.items : (c.items + 1) / 2;
        const int slots = gridFor(pairs, numCUs, 1);
        c.rounds = (pairs + slots - 1) / slots;
        c.cost = 100L * c.rounds * c.steps;
        if (segs == 1 || forced > 0 || c.cost < bestCost)
        {
            bestCost = c.cost;
            pl = c;
        }
    }
    return pl;
}

hipError_t PrepareHrFused()
{
    constexpr int kSmemShipped = kHrWide ? kHrWideSmem : kHrSmem;
    hipError_t e = ensureDynamicLds(reinterpret_cast<const void*>(HrFusedKernel<0, kHrWide>), kSmemShipped);
    if (e == hipSuccess) e = ensureDynamicLds(reinterpret_cast<const void*>(HrFusedKernel<0, kHrWide, true, false>), kSmemShipped);
    if (e == hipSuccess) e = ensureDynamicLds(reinterpret_cast<const void*>(HrFusedKernel<0, kHrWide, false, true>), kSmemShipped);
    if (e == hipSuccess) e = ensureDynamicLds(reinterpret_cast<const void*>(HrFusedKernel<0, kHrWide, true, true>), kSmemShipped);
#if SRX_DIAG
    (void)hrDiagExp();
    (void)hrWide();
    if (e == hipSuccess) e = ensureDynamicLds(reinterpret_cast<const void*>(HrFusedKernel<0, !kHrWide>), kHrWide ? kHrSmem : kHrWideSmem);
    if (e == hipSuccess) e = ensureDynamicLds(reinterpret_cast<const void*>(HrFusedKernel<1, kHrWide>), kHrWide ? kHrWideSmem : kHrSmem);
    if (e == hipSuccess) e = ensureDynamicLds(reinterpret_cast<const void*>(HrFusedKernel<2, kHrWide>), kHrWide ? kHrWideSmem : kHrSmem);
    if (e == hipSuccess) e = ensureDynamicLds(reinterpret_cast<const void*>(HrFusedKernel<3, kHrWide>), kHrWide ? kHrWideSmem : kHrSmem);
    if (e == hipSuccess) e = ensureDynamicLds(reinterpret_cast<const void*>(HrFusedKernel<4, kHrWide>), kHrWide ? kHrWideSmem : kHrSmem);
    if (e == hipSuccess) e = ensureDynamicLds(reinterpret_cast<const void*>(HrFusedKernel<4, !kHrWide>), kHrWide ? kHrSmem : kHrWideSmem);
#endif
    return e;
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
    for (const int b : {l.input.pixelBytes, l.image.pixelBytes})
        if (b != 3 && b != 4) return hipErrorInvalidValue;
    const bool in3 = l.input.pixelBytes == 3, out3 = l.image.pixelBytes == 3;
    // one input and one output frame must be addressable with a 32-bit byte offset
    if (p.imageFrame >= (1L << 31) || p.inputFrame >= (1L << 31)) return hipErrorInvalidValue;

    const bool wide = hrWide();
    const dim3 grid(gridFor(wide ? p.numItems : (p.numItems + 1) / 2, numCUs, 1)), block(kHrThreads);
    // the shipped layout with RGB8 images on either side
    const auto launchRgb = [&] {
        constexpr int kSmem = kHrWide ? kHrWideSmem : kHrSmem;
        if (in3 && out3) hipLaunchKernelGGL((HrFusedKernel<0, kHrWide, true, true>), grid, block, kSmem, s, p);
        else if (in3) hipLaunchKernelGGL((HrFusedKernel<0, kHrWide, true, false>), grid, block, kSmem, s, p);
        else hipLaunchKernelGGL((HrFusedKernel<0, kHrWide, false, true>), grid, block, kSmem, s, p);
    };
    // The kernel was loaded and its shared-memory limit raised by PrepareFusedKernels (srx_create): a launch is
    // the launch and nothing else.
#if SRX_DIAG
    const int exp = hrDiagExp();
    ClockProbe* const probe = clockProbeFor(PROBE_HR_FUSED, int(grid.x), l.mayAllocate, s);
    p.clock = probe ? probe->dev : nullptr;
    constexpr int kSmemDefault = kHrWide ? kHrWideSmem : kHrSmem, kSmemOther = kHrWide ? kHrSmem : kHrWideSmem;
    if (in3 || out3)
    {
        if (wide != kHrWide) return hipErrorInvalidValue;      // RGB8 images exist in the shipped layout only
        launchRgb();
    }
    else if (wide != kHrWide)
    {
        if (exp == 4) hipLaunchKernelGGL((HrFusedKernel<4, !kHrWide>), grid, block, kSmemOther, s, p);
        else hipLaunchKernelGGL((HrFusedKernel<0, !kHrWide>), grid, block, kSmemOther, s, p);
    }
    else switch (exp)
    {
    case 1: hipLaunchKernelGGL((HrFusedKernel<1, kHrWide>), grid, block, kSmemDefault, s, p); break;
    case 2: hipLaunchKernelGGL((HrFusedKernel<2, kHrWide>), grid, block, kSmemDefault, s, p); break;
    case 3: hipLaunchKernelGGL((HrFusedKernel<3, kHrWide>), grid, block, kSmemDefault, s, p); break;
    case 4: hipLaunchKernelGGL((HrFusedKernel<4, kHrWide>), grid, block, kSmemDefault, s, p); break;
    default: hipLaunchKernelGGL((HrFusedKernel<0, kHrWide>), grid, block, kSmemDefault, s, p); break;
    }
    clockProbeReport(probe, "HrFusedKernel", int(grid.x), s, exp == 4 ? 1024 : 0);
#else
    p.clock = nullptr;
    if (in3 || out3) launchRgb();
    else hipLaunchKernelGGL((HrFusedKernel<0, kHrWide>), grid, block, kHrWide ? kHrWideSmem : kHrSmem, s, p);
#endif
    return hipGetLastError();
}

}  // namespace srx
