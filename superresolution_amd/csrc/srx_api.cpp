// C ABI of the MI355X super-resolution library (include/srx.h): context,
// memory plan, weight upload and the launch sequence of one upscale.
#include "../../include/srx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "srx_kernels.h"
#include "srx_weights.h"

namespace {

thread_local std::string g_lastError;

srx_status fail(srx_status code, const std::string& msg)
{
    g_lastError = msg;
    return code;
}

// No C++ exception may cross the C ABI: every entry point is a function-try-block that ends in this.
srx_status translateException() noexcept
{
    try
    {
        try { throw; }
        catch (const std::bad_alloc&) { g_lastError = "out of host memory"; return SRX_ERR_NOMEM; }
        catch (const std::exception& e) { g_lastError = std::string("unexpected exception: ") + e.what(); return SRX_ERR_STATE; }
        catch (...) { g_lastError = "unexpected exception"; return SRX_ERR_STATE; }
    }
    catch (...) { return SRX_ERR_NOMEM; }    // not even the message could be stored
}
#define SRX_CATCH catch (...) { return translateException(); }

#define SRX_HIP(expr)                                                                          \
    do {                                                                                       \
        const hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                                  \
            return fail(SRX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

int roundUp(int v, int m) { return (v + m - 1) / m * m; }

// Channels and resolution (1 = input grid, 2 = 2x grid) of each tensor id.
struct TensorInfo { int channels, storedChannels, scale, border, slackRight; };
const TensorInfo kTensorInfo[SRX_T_COUNT] = {
    {3, 4, 1, 2, 0},     // INPUT   RGB0, read by the 5x5 conv1
    {32, 32, 1, 1, 0},   // CONV1
    {64, 64, 1, 1, 0},   // CONV2
    {64, 64, 1, 1, 0},   // CONV3
    {32, 32, 2, 4, 64},  // CONV_UP1 read by the fused conv4-6 kernel: 3-pixel halo, 64-pixel strips
    {32, 32, 2, 1, 0},   // CONV4   (kept tensors only)
    {32, 32, 2, 1, 0},   // CONV5   (kept tensors only)
    {3, 4, 2, 0, 0},     // CONV6   residual (kept tensors only)
    {3, 4, 2, 0, 0},     // OUTPUT  sum      (kept tensors only)
};

// One kernel launch of the upscale sequence.
enum StageKind { ST_PACK, ST_CONV1, ST_CONV2, ST_CONV3, ST_CONV_UP1, ST_CONV4, ST_CONV5, ST_CONV6, ST_HR_FUSED, ST_LR_FUSED, ST_GENERIC };
const char* const kStageNames[] = {"image_to_tensor", "conv1", "conv2", "conv3", "conv_up1",
                                   "conv4", "conv5", "conv6_add_unorm", "conv4_conv5_conv6_add_unorm",
                                   "pack_conv1_conv2_conv3_conv_up1", "generic_layer_graph"};

// Workspace of srx_upscale_static: conv_up1's output (the one tensor the two fused kernels pass through
// device memory) for one image.  Every GPU the library runs on holds its own copy of this array until the
// process ends.  In host code the array's name is only a handle for hipGetSymbolAddress: what kernels and
// memsets get is the per-device address kept in g_staticScratch.
constexpr size_t kStaticScratchBytes = size_t(SRX_STATIC_SCRATCH_BYTES);
constexpr int kMaxDevices = 64;
std::mutex g_staticScratchMutex;
void* g_staticScratch[kMaxDevices] = {};

}  // namespace

static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {

// Address of `scratch` on `device` (which must be the current device); looked up once per device.
srx_status staticScratchFor(int device, void** out)
{
    if (device < 0 || device >= kMaxDevices) return fail(SRX_ERR_INVALID, "device index must be below 64");
    std::lock_guard<std::mutex> lock(g_staticScratchMutex);
    if (!g_staticScratch[device])
    {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr)
            return fail(SRX_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_staticScratch[device] = ptr;
    }
    *out = g_staticScratch[device];
    return SRX_OK;
}

// Layout of tensor `id` for one w x h input frame, as the kernels address it: the one place that knows the
// border, the padding to whole tiles and the slack (the ctx workspace and srx_upscale_static both use it).
struct TensorPlacement { int pitch, rows; int64_t frameHalfs; size_t bytes, baseOffsetHalfs; };
TensorPlacement tensorPlacement(int id, uint32_t w, uint32_t h)
{
    const TensorInfo& ti = kTensorInfo[id];
    TensorPlacement l;
    l.pitch = roundUp(int(w) * ti.scale, srx::kTileW) + 2 * ti.border + ti.slackRight;
    l.rows = roundUp(int(h) * ti.scale, srx::kTileH) + 2 * ti.border;
    l.frameHalfs = int64_t(l.pitch) * l.rows * ti.storedChannels;
    l.bytes = size_t(l.frameHalfs) * 2;
    l.baseOffsetHalfs = (size_t(ti.border) * l.pitch + ti.border) * ti.storedChannels;   // image pixel (0, 0)
    return l;
}

// The Tensor the kernels take for `id` in memory that starts at `mem`.
srx::Tensor tensorAt(int id, const TensorPlacement& l, void* mem)
{
    srx::Tensor t;
    t.channels = kTensorInfo[id].storedChannels;
    t.pitch = l.pitch;
    t.frame = l.frameHalfs;
    t.base = static_cast<uint16_t*>(mem) + l.baseOffsetHalfs;
    return t;
}

}  // namespace

struct srx_ctx
{
    int device = 0;
    int numCUs = 256;
    uint32_t width = 0, height = 0, batch = 0, flags = 0;
    uint32_t chunk = 0;                      // frames of activation workspace (<= batch)
    // frames whose conv_up1 tensor the workspace holds: slot i = frame residentFirst + i (fused path; set by the
    // last launch of the first fused kernel through this ctx's workspace)
    uint32_t residentFirst = 0, residentCount = 0;

    srx::FoldedWeights folded;
    uint16_t* wpk[srx::kNumConvLayers] = {};
    float* bias[srx::kNumConvLayers] = {};
    uint16_t* wpk6Rows16 = nullptr;          // conv6 as row-streaming 16x16x32 fragments (fused kernel)
    uint16_t* wpk1Fused = nullptr;           // conv1 as 7 K-steps of four taps (fused kernel)
    uint16_t* wpkK32[6] = {};                // conv2, conv3, conv_up1 (row-shared form), conv4, conv5 as 16x16x32 fragments (fused kernels), by layer
    float* biasLrFused = nullptr;            // conv_up1 | conv3 | conv2 | conv1 biases (fused kernel)
    std::vector<StageKind> stages;           // launch sequence of one upscale

    srx::Image input, output;
    srx::Tensor tensor[SRX_T_COUNT];
    size_t tensorFrameBytes[SRX_T_COUNT] = {};
    void* tensorAlloc[SRX_T_COUNT] = {};
    int tensorRows[SRX_T_COUNT] = {};

    std::vector<void*> allocs;
    std::vector<uint8_t> hostStage;
    void* staticScratch = nullptr;           // this device's copy of `scratch` (srx_upscale_static)
    int livePipes = 0;                       // frame pipes created on this ctx and not yet destroyed

    // the graph made of generic layers (SRX_FLAG_GENERIC_LAYERS / FP32 / NHWC)
    struct Generic
    {
        srx::ML::TensorDataType dtype = srx::ML::TensorDataType::FLOAT16;
        srx::ML::TensorLayout layout = srx::ML::TensorLayout::DEFAULT;
        srx::ML::Operator up0, up1, conv[srx::kNumConvLayers], add;
        void* filter[srx::kNumConvLayers] = {};
        void* bias[srx::kNumConvLayers] = {};
        void* t[SRX_T_COUNT] = {};            // {batch, C, H, W} each
        void* up1Out = nullptr;               // conv3 upsampled, {batch, 64, 2H, 2W}
        size_t elem() const { return dtype == srx::ML::TensorDataType::FLOAT16 ? 2 : 4; }
    } gen;

    bool keep() const { return (flags & SRX_FLAG_KEEP_TENSORS) != 0; }
    bool generic() const { return (flags & (SRX_FLAG_GENERIC_LAYERS | SRX_FLAG_FP32 | SRX_FLAG_NHWC)) != 0; }
};

namespace {

srx_status allocZero(srx_ctx* c, void** p, size_t bytes)
{
    SRX_HIP(hipMalloc(p, bytes));
    c->allocs.push_back(*p);
    SRX_HIP(hipMemset(*p, 0, bytes));
    return SRX_OK;
}

std::vector<uint16_t> packLayer(const srx::FoldedWeights& folded, int l)
{
    const uint32_t* sz = srx::kConvLayers[l].sizes;
    if (l == 0) return srx::PackConv1(folded.filter[l]);
    if (l == 3) return srx::PackConvUp1Phases(folded.filter[l]);
    return srx::PackConv3x3(folded.filter[l], int(sz[0]), int(sz[1]));
}

srx_status uploadWeights(srx_ctx* c)
{
    using namespace srx;
    for (int l = 0; l < kNumConvLayers; ++l)
    {
        const std::vector<uint16_t> packed = packLayer(c->folded, l);
        const int biasTiles = (l == 3) ? 4 : (int(kConvLayers[l].sizes[0]) + 31) / 32;

        void* d = nullptr;
        if (const srx_status s = allocZero(c, &d, packed.size() * 2)) return s;
        SRX_HIP(hipMemcpy(d, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
        c->wpk[l] = static_cast<uint16_t*>(d);

        // fp32 copy of the (truncated fp16) bias, one value per output row of every 32-row tile
        std::vector<float> b(size_t(biasTiles) * 32, 0.0f);
        for (int t = 0; t < biasTiles; ++t)
            for (int r = 0; r < 32; ++r)
            {
                const int o = (l == 3) ? r : t * 32 + r;
                if (o < int(c->folded.bias[l].size())) b[t * 32 + r] = ML::Float16ToFloat32(c->folded.bias[l][o]);
            }
        if (const srx_status s = allocZero(c, &d, b.size() * 4)) return s;
        SRX_HIP(hipMemcpy(d, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        c->bias[l] = static_cast<float*>(d);
    }
    {
        const std::vector<uint16_t> packed = PackConv6Rotating(c->folded.filter[6]);
        void* d = nullptr;
        if (const srx_status s = allocZero(c, &d, packed.size() * 2)) return s;
        SRX_HIP(hipMemcpy(d, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
        c->wpk6Rows16 = static_cast<uint16_t*>(d);
    }
    for (int l = 1; l <= 5; ++l)
    {
        const uint32_t* sz = kConvLayers[l].sizes;
        const std::vector<uint16_t> packed = l == 3 ? PackConvUp1RowSharedK32(c->folded.filter[l])
                                                    : PackConv3x3K32(c->folded.filter[l], int(sz[0]), int(sz[1]), true);
        void* d = nullptr;
        if (const srx_status s = allocZero(c, &d, packed.size() * 2)) return s;
        SRX_HIP(hipMemcpy(d, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
        c->wpkK32[l] = static_cast<uint16_t*>(d);
    }
    {
        const std::vector<uint16_t> packed = PackConv1(c->folded.filter[0]);
        void* d = nullptr;
        if (const srx_status s = allocZero(c, &d, packed.size() * 2)) return s;
        SRX_HIP(hipMemcpy(d, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
        c->wpk1Fused = static_cast<uint16_t*>(d);

        std::vector<float> b;
        for (int l : {3, 2, 1, 0})
            for (uint16_t v : c->folded.bias[l]) b.push_back(ML::Float16ToFloat32(v));
        if (const srx_status s = allocZero(c, &d, b.size() * 4)) return s;
        SRX_HIP(hipMemcpy(d, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        c->biasLrFused = static_cast<float*>(d);
    }
    return SRX_OK;
}

srx_status allocTensors(srx_ctx* c)
{
    // Per-frame bytes of the activation workspace decide how many frames run per chunk.
    size_t perFrame = 0;
    for (int id = 0; id < SRX_T_COUNT; ++id)
    {
        if (id != SRX_T_CONV_UP1 && !c->keep()) continue;   // the fused kernels keep the rest in LDS
        const TensorPlacement lay = tensorPlacement(id, c->width, c->height);
        c->tensor[id] = tensorAt(id, lay, nullptr);          // base is set once the memory exists
        c->tensor[id].base = nullptr;
        c->tensorRows[id] = lay.rows;
        c->tensorFrameBytes[id] = lay.bytes;
        if (id != SRX_T_INPUT) perFrame += c->tensorFrameBytes[id];
    }
    // Frames resident per pass.  Small images are batched into one launch per stage so that the
    // launch has work for every CU (up to 256 MB of workspace).  A large image could fill the GPU
    // on its own by cutting its rows into segments, but every segment re-computes the pipeline's
    // warm-up rows; several frames per pass let the strips run the full height instead.  Among the
    // pass sizes the workspace limit allows (16 GB, SRX_WORKSPACE_MB overrides) take the one with
    // the least estimated time per frame: a row step of the low-resolution kernel takes 3.2x as
    // long as one of the high-resolution kernel (2.8 against 0.88 us at 1080p).
    size_t limit = size_t(16) << 30;
    if (const char* env = std::getenv("SRX_WORKSPACE_MB"))
        limit = std::max<size_t>(1, std::strtoull(env, nullptr, 10)) << 20;
    perFrame = std::max<size_t>(perFrame, 1);
    const uint32_t maxChunk = uint32_t(std::max<size_t>(1, std::min<size_t>(c->batch, limit / perFrame)));
    uint32_t chunk = uint32_t(std::max<size_t>(1, std::min<size_t>(maxChunk, (size_t(256) << 20) / perFrame)));
    if (c->keep()) chunk = maxChunk;         // every kept tensor of the whole batch must be resident: only the workspace limit applies
    if (!c->keep())
    {
        // cost of one pass of n frames, and of the whole batch in passes of n (the last pass takes the rest)
        const auto passCost = [&](uint32_t n) {
            const srx::StripPlan lr = srx::PlanLrFused(int(c->width), int(c->height), int(n), c->numCUs);
            const srx::StripPlan hr = srx::PlanHrFused(2 * int(c->width), 2 * int(c->height), int(n), c->numCUs);
            return 3.2 * double(lr.cost) + 1.0 * double(hr.cost);
        };
        double best = 0.0;
        const uint32_t first = chunk, last = std::min<uint32_t>(maxChunk, chunk + 64);
        for (uint32_t n = first; n <= last; ++n)
        {
            const uint32_t rest = c->batch % n;
            const double cost = (c->batch / n) * passCost(n) + (rest ? passCost(rest) : 0.0);
            if (n == first || cost < best * 0.98)
            {
                best = cost;
                chunk = n;
            }
        }
    }
    if (c->keep() && chunk < c->batch)
        return fail(SRX_ERR_NOMEM, "SRX_FLAG_KEEP_TENSORS needs every tensor of the whole batch resident, which exceeds the "
                                   "workspace limit (16 GB, or SRX_WORKSPACE_MB): reduce batch or raise the limit");
    c->chunk = chunk;

    for (int id = 0; id < SRX_T_COUNT; ++id)
    {
        if (c->tensorFrameBytes[id] == 0) continue;
        const uint32_t frames = (id == SRX_T_INPUT) ? c->batch : chunk;
        void* d = nullptr;
        if (const srx_status s = allocZero(c, &d, c->tensorFrameBytes[id] * frames)) return s;
        c->tensorAlloc[id] = d;
        c->tensor[id] = tensorAt(id, tensorPlacement(id, c->width, c->height), d);
    }
    return SRX_OK;
}

// The reference's createResources (Content/SuperResolution.cpp:172-275) and createWeightTensors
// (:308-340) with the generic layer factories: same order of layers, same size bookkeeping.
srx_status buildGenericGraph(srx_ctx* c, const srx::ML::WeightMapType& weights)
{
    using namespace srx;
    srx_ctx::Generic& g = c->gen;
    g.dtype = (c->flags & SRX_FLAG_FP32) ? ML::TensorDataType::FLOAT32 : ML::TensorDataType::FLOAT16;
    g.layout = (c->flags & SRX_FLAG_NHWC) ? ML::TensorLayout::NHWC : ML::TensorLayout::DEFAULT;
    const ML::Util util(g.dtype, g.layout);
    uint64_t inBytes = 0, outBytes = 0;
    const uint32_t inputSizes[4] = {c->batch, 3, c->height, c->width};
    uint32_t sizes[SRX_T_COUNT][4] = {};
    std::memcpy(sizes[SRX_T_INPUT], inputSizes, sizeof(inputSizes));
    bool ok = util.CreateUpsampleLayer(inputSizes, inBytes, outBytes, sizes[SRX_T_OUTPUT], g.up0);
    uint32_t cur[4];
    std::memcpy(cur, inputSizes, sizeof(cur));
    uint32_t up1Sizes[4] = {};
    for (int l = 0; ok && l < kNumConvLayers; ++l)
    {
        if (l == 3)
        {
            ok = util.CreateUpsampleLayer(cur, inBytes, outBytes, up1Sizes, g.up1);
            std::memcpy(cur, up1Sizes, sizeof(cur));
        }
        const ConvLayerDesc& d = kConvLayers[l];
        ok = ok && util.CreateConvolutionLayer(cur, d.sizes, d.scale != nullptr, inBytes, outBytes, sizes[SRX_T_CONV1 + l], g.conv[l]);
        std::memcpy(cur, sizes[SRX_T_CONV1 + l], sizeof(cur));
    }
    ok = ok && util.CreateAdditionLayer(sizes[SRX_T_OUTPUT], g.add);
    if (!ok) return fail(SRX_ERR_INVALID, "generic graph: a layer factory rejected its sizes");

    const auto count = [](const uint32_t s[4]) { return size_t(s[0]) * s[1] * s[2] * s[3]; };
    for (int id = 0; id < SRX_T_COUNT; ++id)
        if (const srx_status s = allocZero(c, &g.t[id], count(sizes[id]) * g.elem())) return s;
    if (const srx_status s = allocZero(c, &g.up1Out, count(up1Sizes) * g.elem())) return s;
    for (int l = 0; l < kNumConvLayers; ++l)
    {
        const ConvLayerDesc& d = kConvLayers[l];
        std::vector<uint8_t> f, b;
        if (!util.CreateWeightTensors(weights, d.conv, d.scale, d.shift, d.sizes, f, b))
            return fail(SRX_ERR_WEIGHTS, "loadWeights: a layer's filter, scale or shift tensor is missing or has the wrong size");
        if (const srx_status s = allocZero(c, &g.filter[l], f.size())) return s;
        SRX_HIP(hipMemcpy(g.filter[l], f.data(), f.size(), hipMemcpyHostToDevice));
        if (!b.empty())
        {
            if (const srx_status s = allocZero(c, &g.bias[l], b.size())) return s;
            SRX_HIP(hipMemcpy(g.bias[l], b.data(), b.size(), hipMemcpyHostToDevice));
        }
    }
    return SRX_OK;
}

// Dispatch order of SuperResolution::Process (Content/SuperResolution.cpp:73-103), then Render's pass.
srx_status runGenericGraph(srx_ctx* c, hipStream_t s)
{
    using namespace srx;
    srx_ctx::Generic& g = c->gen;
    SRX_HIP(DispatchOperator(g.up0, g.t[SRX_T_INPUT], nullptr, nullptr, g.t[SRX_T_OUTPUT], s));
    const void* cur = g.t[SRX_T_INPUT];
    for (int l = 0; l < kNumConvLayers; ++l)
    {
        if (l == 3)
        {
            SRX_HIP(DispatchOperator(g.up1, cur, nullptr, nullptr, g.up1Out, s));
            cur = g.up1Out;
        }
        SRX_HIP(DispatchOperator(g.conv[l], cur, g.filter[l], g.bias[l], g.t[SRX_T_CONV1 + l], s));
        cur = g.t[SRX_T_CONV1 + l];
    }
    SRX_HIP(DispatchOperator(g.add, g.t[SRX_T_CONV6], g.t[SRX_T_OUTPUT], nullptr, g.t[SRX_T_OUTPUT], s));
    SRX_HIP(LaunchTensorToImage(g.t[SRX_T_OUTPUT], c->output, g.dtype, g.layout, int(c->width) * 2, int(c->height) * 2,
                                int(c->batch), s));
    return SRX_OK;
}

srx::Tensor frameOffset(const srx::Tensor& t, uint32_t frame)
{
    srx::Tensor r = t;
    if (r.base) r.base += size_t(frame) * t.frame;
    return r;
}

srx::Image frameOffset(const srx::Image& im, uint32_t frame)
{
    srx::Image r = im;
    r.base += size_t(frame) * im.frame;
    return r;
}

// Launch one stage for frames [first, first + count).  `in` / `out`, where given, stand for the ctx's own input /
// output images of those frames (frame 0 of the override = frame `first`): the two fused kernels then read / write
// them instead, RGBA8 or RGB8.
srx_status launchStage(srx_ctx* c, StageKind stage, uint32_t first, uint32_t count, hipStream_t s,
                       const srx::Image* in = nullptr, const srx::Image* out = nullptr)
{
    using namespace srx;
    const int W = int(c->width), H = int(c->height);
    if (stage == ST_GENERIC) return runGenericGraph(c, s);
    if (stage == ST_PACK)
    {
        SRX_HIP(LaunchPack(frameOffset(c->input, first), frameOffset(c->tensor[SRX_T_INPUT], first), W, H, int(count), s));
        return SRX_OK;
    }
    if (stage == ST_HR_FUSED)
    {
        HrFusedLaunch hl;
        hl.a4 = c->tensor[SRX_T_CONV_UP1];
        hl.input = in ? *in : frameOffset(c->input, first);
        hl.wpk4 = c->wpkK32[4];
        hl.wpk5 = c->wpkK32[5];
        hl.wpk6 = c->wpk6Rows16;
        hl.bias4 = c->bias[4];
        hl.bias5 = c->bias[5];
        hl.image = out ? *out : frameOffset(c->output, first);
        hl.width2 = 2 * W;
        hl.height2 = 2 * H;
        hl.batch = int(count);
        SRX_HIP(LaunchHrFused(hl, c->numCUs, s));
        return SRX_OK;
    }
    if (stage == ST_LR_FUSED)
    {
        LrFusedLaunch ll;
        ll.image = in ? *in : frameOffset(c->input, first);
        ll.a4 = c->tensor[SRX_T_CONV_UP1];
        ll.wpk1 = c->wpk1Fused;
        ll.wpk2 = c->wpkK32[1];
        ll.wpk3 = c->wpkK32[2];
        ll.wpkUp1 = c->wpkK32[3];
        ll.bias = c->biasLrFused;
        ll.width = W;
        ll.height = H;
        ll.batch = int(count);
        SRX_HIP(LaunchLrFused(ll, c->numCUs, s));
        c->residentFirst = first;
        c->residentCount = count;
        return SRX_OK;
    }
    const int l = int(stage) - int(ST_CONV1);
    ConvLaunch cl;
    cl.batch = int(count);
    cl.wpk = c->wpk[l];
    cl.bias = c->bias[l];
    static const ConvKind kinds[7] = {CONV_5X5_RGB, CONV_3X3_32_64, CONV_3X3_64_64, CONV_UP1_PHASES,
                                      CONV_3X3_32_32, CONV_3X3_32_32, CONV_3X3_32_RGB};
    cl.kind = kinds[l];
    // workspace tensors are indexed by frame-in-chunk (chunks start at frame 0 of the workspace)
    cl.in = (l == 0) ? frameOffset(c->tensor[SRX_T_INPUT], first) : c->tensor[SRX_T_INPUT + l];
    if (l < 6) cl.out = c->tensor[SRX_T_CONV1 + l];
    const bool hiRes = l >= 4;          // conv_up1 walks the input grid and writes the 2x grid itself
    cl.width = hiRes ? 2 * W : W;
    cl.height = hiRes ? 2 * H : H;
    if (l == 6)
    {
        cl.x0 = frameOffset(c->tensor[SRX_T_INPUT], first);
        cl.image = frameOffset(c->output, first);
        if (c->keep())
        {
            cl.residual = c->tensor[SRX_T_CONV6];
            cl.sum = c->tensor[SRX_T_OUTPUT];
        }
    }
    SRX_HIP(LaunchConv(cl, c->numCUs, s));
    return SRX_OK;
}

bool validCtx(const srx_ctx* c) { return c != nullptr && c->batch > 0; }

// srx_upscale_frames on images other than the ctx's own: `in` / `out` (either may be NULL) hold frames
// first .. first + count - 1 back to back from their frame 0, RGBA8 or RGB8.
srx_status upscaleFramesOn(srx_ctx* c, uint32_t first, uint32_t count, hipStream_t stream, const srx::Image* in, const srx::Image* out)
{
    if (!validCtx(c) || count == 0 || first >= c->batch || count > c->batch - first)
        return fail(SRX_ERR_INVALID, "srx_upscale_frames: bad frame range");
    if (c->keep() || c->generic())
        return fail(SRX_ERR_STATE, "srx_upscale_frames: not available with SRX_FLAG_KEEP_TENSORS or the generic-layer graph (their tensors are indexed by frame of the whole batch)");
    SRX_HIP(hipSetDevice(c->device));
    for (uint32_t f = first; f < first + count; f += c->chunk)
    {
        const uint32_t n = std::min(c->chunk, first + count - f);
        srx::Image inAt, outAt;
        if (in) inAt = frameOffset(*in, f - first);
        if (out) outAt = frameOffset(*out, f - first);
        for (size_t i = 0; i < c->stages.size(); ++i)
            if (const srx_status s = launchStage(c, c->stages[i], f, n, stream, in ? &inAt : nullptr, out ? &outAt : nullptr)) return s;
    }
    return SRX_OK;
}

}  // namespace

extern "C" {

const char* srx_last_error(void) { return g_lastError.c_str(); }

srx_status srx_create(srx_ctx** out, int device, const char* weights_path,
                      uint32_t in_w, uint32_t in_h, uint32_t batch, uint32_t flags) try
{
    if (!out) return fail(SRX_ERR_INVALID, "srx_create: out is NULL");
    *out = nullptr;
    if (!weights_path) return fail(SRX_ERR_INVALID, "srx_create: weights_path is NULL");
    // one output frame (2W x 2H x 4 bytes) must stay addressable with a signed 32-bit byte offset
    if (in_w == 0 || in_h == 0 || batch == 0 || in_w > 16384 || in_h > 16384 ||
        uint64_t(in_w) * in_h * 16 >= (uint64_t(1) << 31))
        return fail(SRX_ERR_INVALID, "srx_create: image size must be 1..16384 per side, fewer than 2^27 pixels, and batch >= 1");

    srx::ML::WeightMapType weights;
    if (!srx::LoadWeights(weights_path, weights, (flags & SRX_FLAG_VERBOSE_WEIGHTS) != 0))
        return fail(SRX_ERR_WEIGHTS, std::string("loadWeights: ") + weights_path);

    struct Destroy { void operator()(srx_ctx* x) const { srx_destroy(x); } };
    std::unique_ptr<srx_ctx, Destroy> owner(new srx_ctx);      // released on success only
    srx_ctx* c = owner.get();
    c->device = device;
    c->width = in_w;
    c->height = in_h;
    c->batch = batch;
    c->flags = flags;
    if (!srx::FoldWeights(weights, c->folded))
    {
        return fail(SRX_ERR_WEIGHTS, "loadWeights: a layer's filter, scale or shift tensor is missing or has the wrong size");
    }

    const auto bail = [&](srx_status s) { return s; };             // `owner` destroys the ctx
    {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return bail(fail(SRX_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e)));
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device);
        if (e != hipSuccess) return bail(fail(SRX_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e)));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return bail(fail(SRX_ERR_HIP, std::string("this library is built for gfx950 only; device is ") + prop.gcnArchName));
        c->numCUs = prop.multiProcessorCount;
    }
    // set-up for srx_upscale_static: this device's address of the static workspace, and the fused kernels
    // loaded on this device with their shared-memory limit raised, so that the entry point itself does neither
    if (const srx_status s = staticScratchFor(device, &c->staticScratch)) return bail(s);
    {
        const hipError_t e = srx::PrepareFusedKernels();
        if (e != hipSuccess) return bail(fail(SRX_ERR_HIP, std::string("PrepareFusedKernels: ") + hipGetErrorString(e)));
    }

    c->input.pitch = int64_t(in_w) * 4;
    c->input.frame = c->input.pitch * in_h;
    c->output.pitch = int64_t(in_w) * 8;
    c->output.frame = c->output.pitch * in_h * 2;
    void* d = nullptr;
    if (const srx_status s = allocZero(c, &d, size_t(c->input.frame) * batch)) return bail(s);
    c->input.base = static_cast<uint8_t*>(d);
    if (const srx_status s = allocZero(c, &d, size_t(c->output.frame) * batch)) return bail(s);
    c->output.base = static_cast<uint8_t*>(d);

    if (c->generic())
    {
        c->stages = {ST_GENERIC};
        c->chunk = batch;
        if (const srx_status s = buildGenericGraph(c, weights)) return bail(s);
    }
    else
    {
        if (c->keep())
            c->stages = {ST_PACK, ST_CONV1, ST_CONV2, ST_CONV3, ST_CONV_UP1, ST_CONV4, ST_CONV5, ST_CONV6};
        else
            c->stages = {ST_LR_FUSED, ST_HR_FUSED};
        if (const srx_status s = uploadWeights(c)) return bail(s);
        if (const srx_status s = allocTensors(c)) return bail(s);
    }
    {
        const hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) return bail(fail(SRX_ERR_HIP, std::string("hipDeviceSynchronize: ") + hipGetErrorString(e)));
    }
    *out = owner.release();
    return SRX_OK;
}
SRX_CATCH

srx_status srx_destroy(srx_ctx* c) try
{
    if (!c) return SRX_OK;
    if (c->livePipes > 0)
        return fail(SRX_ERR_STATE, "srx_destroy: a frame pipe of this ctx is still alive; call srx_pipe_destroy first");
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();          // no buffer is freed while stream work is pending
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
    return SRX_OK;
}
SRX_CATCH

srx_status srx_get_out_size(const srx_ctx* c, uint32_t* w, uint32_t* h) try
{
    if (!validCtx(c)) return fail(SRX_ERR_INVALID, "invalid ctx");
    if (w) *w = c->width << 1;      // Content/SuperResolution.cpp:50-51
    if (h) *h = c->height << 1;
    return SRX_OK;
}
SRX_CATCH

srx_status srx_get_in_size(const srx_ctx* c, uint32_t* w, uint32_t* h, uint32_t* batch) try
{
    if (!validCtx(c)) return fail(SRX_ERR_INVALID, "invalid ctx");
    if (w) *w = c->width;
    if (h) *h = c->height;
    if (batch) *batch = c->batch;
    return SRX_OK;
}
SRX_CATCH

srx_status srx_upload_rgba8(srx_ctx* c, uint32_t frame, const uint8_t* host, size_t pitch, void* stream) try
{
    if (!validCtx(c) || !host || frame >= c->batch || pitch < size_t(c->width) * 4)
        return fail(SRX_ERR_INVALID, "srx_upload_rgba8: bad frame, pointer or pitch");
    SRX_HIP(hipSetDevice(c->device));
    SRX_HIP(hipMemcpy2DAsync(c->input.base + size_t(frame) * c->input.frame, size_t(c->input.pitch), host, pitch,
                             size_t(c->width) * 4, c->height, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    return SRX_OK;
}
SRX_CATCH

srx_status srx_upload_rgb8(srx_ctx* c, uint32_t frame, const uint8_t* host, size_t pitch, void* stream) try
{
    if (!validCtx(c) || !host || frame >= c->batch || pitch < size_t(c->width) * 3)
        return fail(SRX_ERR_INVALID, "srx_upload_rgb8: bad frame, pointer or pitch");
    SRX_HIP(hipSetDevice(c->device));
    // expand to RGBA (alpha 255) on the host, as the loader does for 3-channel files
    c->hostStage.resize(size_t(c->width) * c->height * 4);
    for (uint32_t y = 0; y < c->height; ++y)
    {
        const uint8_t* s = host + size_t(y) * pitch;
        uint8_t* d = c->hostStage.data() + size_t(y) * c->width * 4;
        for (uint32_t x = 0; x < c->width; ++x)
        {
            d[4 * x + 0] = s[3 * x + 0];
            d[4 * x + 1] = s[3 * x + 1];
            d[4 * x + 2] = s[3 * x + 2];
            d[4 * x + 3] = 255;
        }
    }
    const hipStream_t st = static_cast<hipStream_t>(stream);
    SRX_HIP(hipMemcpyAsync(c->input.base + size_t(frame) * c->input.frame, c->hostStage.data(), c->hostStage.size(),
                           hipMemcpyHostToDevice, st));
    SRX_HIP(hipStreamSynchronize(st));     // the staging vector is reused by the next call
    return SRX_OK;
}
SRX_CATCH

srx_status srx_input_device_ptr(srx_ctx* c, uint32_t frame, void** dptr, size_t* pitch) try
{
    if (!validCtx(c) || frame >= c->batch || !dptr) return fail(SRX_ERR_INVALID, "srx_input_device_ptr: bad argument");
    *dptr = c->input.base + size_t(frame) * c->input.frame;
    if (pitch) *pitch = size_t(c->input.pitch);
    return SRX_OK;
}
SRX_CATCH

srx_status srx_output_device_ptr(srx_ctx* c, uint32_t frame, void** dptr, size_t* pitch) try
{
    if (!validCtx(c) || frame >= c->batch || !dptr) return fail(SRX_ERR_INVALID, "srx_output_device_ptr: bad argument");
    *dptr = c->output.base + size_t(frame) * c->output.frame;
    if (pitch) *pitch = size_t(c->output.pitch);
    return SRX_OK;
}
SRX_CATCH

srx_status srx_image_to_tensor(srx_ctx* c, void* stream) try
{
    if (!validCtx(c)) return fail(SRX_ERR_INVALID, "invalid ctx");
    SRX_HIP(hipSetDevice(c->device));
    if (c->generic())
    {
        SRX_HIP(srx::LaunchImageToTensor(c->input, c->gen.t[SRX_T_INPUT], c->gen.dtype, c->gen.layout, int(c->width),
                                         int(c->height), int(c->batch), static_cast<hipStream_t>(stream)));
        return SRX_OK;
    }
    // the fused path converts the image inside its first kernel: nothing to launch here
    if (c->stages[0] != ST_PACK) return SRX_OK;
    return launchStage(c, ST_PACK, 0, c->batch, static_cast<hipStream_t>(stream));
}
SRX_CATCH

srx_status srx_process(srx_ctx* c, void* stream) try
{
    if (!validCtx(c)) return fail(SRX_ERR_INVALID, "invalid ctx");
    SRX_HIP(hipSetDevice(c->device));
    for (uint32_t first = 0; first < c->batch; first += c->chunk)
    {
        const uint32_t count = std::min(c->chunk, c->batch - first);
        for (size_t i = (c->stages[0] == ST_PACK ? 1 : 0); i < c->stages.size(); ++i)
            if (const srx_status s = launchStage(c, c->stages[i], first, count, static_cast<hipStream_t>(stream))) return s;
    }
    return SRX_OK;
}
SRX_CATCH

srx_status srx_upscale(srx_ctx* c, void* stream) try
{
    if (const srx_status s = srx_image_to_tensor(c, stream)) return s;
    return srx_process(c, stream);
}
SRX_CATCH

srx_status srx_upscale_frames(srx_ctx* c, uint32_t first, uint32_t count, void* stream) try
{
    return upscaleFramesOn(c, first, count, static_cast<hipStream_t>(stream), nullptr, nullptr);
}
SRX_CATCH

srx_status srx_upscale_static(srx_ctx* c, const void* in, size_t inPitch, void* out, size_t outPitch,
                              uint32_t w, uint32_t h, void* stream) try
{
    if (!validCtx(c)) return fail(SRX_ERR_INVALID, "invalid ctx");
    if (c->generic()) return fail(SRX_ERR_STATE, "srx_upscale_static: not available on a generic-layer ctx (it holds no fused-kernel weights)");
    if (!in || !out || w == 0 || h == 0 || w > 16384 || h > 16384)
        return fail(SRX_ERR_INVALID, "srx_upscale_static: NULL image or a side outside 1..16384");
    if (inPitch < size_t(w) * 4 || outPitch < size_t(w) * 8 || (inPitch & 3) || (outPitch & 3) ||
        (reinterpret_cast<uintptr_t>(in) & 3) || (reinterpret_cast<uintptr_t>(out) & 3))
        return fail(SRX_ERR_INVALID, "srx_upscale_static: pitches must cover a row (4 bytes per pixel) and, like the pointers, be multiples of 4");
    const TensorPlacement lay = tensorPlacement(SRX_T_CONV_UP1, w, h);
    if (lay.bytes > kStaticScratchBytes)
        return fail(SRX_ERR_INVALID, "srx_upscale_static: image too large for the static workspace: 64 bytes per output pixel plus borders "
                                     "must fit " + std::to_string(kStaticScratchBytes >> 20) + " MiB (e.g. 640x360 fits, 1280x720 does not); "
                                     "use srx_upscale for larger images");
    if (inPitch * h >= (size_t(1) << 31) || outPitch * h * 2 >= (size_t(1) << 31))
        return fail(SRX_ERR_INVALID, "srx_upscale_static: an image must span fewer than 2^31 bytes");
    SRX_HIP(hipSetDevice(c->device));
    const hipStream_t st = static_cast<hipStream_t>(stream);

    // The workspace keeps what the previous call left.  Its border, the padding to whole tiles and the slack
    // to the right must read as zeros (they are the zero padding of conv4), and their position depends on
    // the image size: clear the part this call uses (with a kernel, not hipMemsetAsync: replayed from a HIP graph,
    // the memset was seen to run beside the kernel that follows it).
    SRX_HIP(srx::LaunchZero(c->staticScratch, lay.bytes, st));

    const srx::Tensor a4 = tensorAt(SRX_T_CONV_UP1, lay, c->staticScratch);
    srx::Image src, dst;
    src.base = static_cast<uint8_t*>(const_cast<void*>(in));
    src.pitch = int64_t(inPitch);
    src.frame = int64_t(inPitch) * h;
    dst.base = static_cast<uint8_t*>(out);
    dst.pitch = int64_t(outPitch);
    dst.frame = int64_t(outPitch) * h * 2;

    srx::LrFusedLaunch ll;
    ll.image = src;
    ll.a4 = a4;
    ll.wpk1 = c->wpk1Fused;
    ll.wpk2 = c->wpkK32[1];
    ll.wpk3 = c->wpkK32[2];
    ll.wpkUp1 = c->wpkK32[3];
    ll.bias = c->biasLrFused;
    ll.width = int(w);
    ll.height = int(h);
    ll.batch = 1;
    ll.mayAllocate = false;
    SRX_HIP(srx::LaunchLrFused(ll, c->numCUs, st));

    srx::HrFusedLaunch hl;
    hl.a4 = a4;
    hl.input = src;
    hl.wpk4 = c->wpkK32[4];
    hl.wpk5 = c->wpkK32[5];
    hl.wpk6 = c->wpk6Rows16;
    hl.bias4 = c->bias[4];
    hl.bias5 = c->bias[5];
    hl.image = dst;
    hl.width2 = 2 * int(w);
    hl.height2 = 2 * int(h);
    hl.batch = 1;
    hl.mayAllocate = false;
    SRX_HIP(srx::LaunchHrFused(hl, c->numCUs, st));
    return SRX_OK;
}
SRX_CATCH

size_t srx_static_scratch_bytes(void) { return kStaticScratchBytes; }

srx_status srx_tensor_to_image(srx_ctx* c, uint32_t frame, uint8_t* host, size_t pitch, void* stream) try
{
    if (!validCtx(c) || frame >= c->batch) return fail(SRX_ERR_INVALID, "srx_tensor_to_image: bad frame");
    if (host && pitch < size_t(c->width) * 8) return fail(SRX_ERR_INVALID, "srx_tensor_to_image: pitch too small");
    SRX_HIP(hipSetDevice(c->device));
    const hipStream_t st = static_cast<hipStream_t>(stream);
    if (c->keep() && !c->generic())
        SRX_HIP(srx::LaunchUnpack(frameOffset(c->tensor[SRX_T_OUTPUT], frame), frameOffset(c->output, frame),
                                  int(c->width) * 2, int(c->height) * 2, 1, st));
    if (host)
        SRX_HIP(hipMemcpy2DAsync(host, pitch, c->output.base + size_t(frame) * c->output.frame, size_t(c->output.pitch),
                                 size_t(c->width) * 8, size_t(c->height) * 2, hipMemcpyDeviceToHost, st));
    return SRX_OK;
}
SRX_CATCH

srx_status srx_sync(srx_ctx* c, void* stream) try
{
    if (!validCtx(c)) return fail(SRX_ERR_INVALID, "invalid ctx");
    SRX_HIP(hipSetDevice(c->device));
    SRX_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return SRX_OK;
}
SRX_CATCH

srx_status srx_tensor_shape(const srx_ctx* c, int32_t id, uint32_t* ch, uint32_t* h, uint32_t* w) try
{
    if (!validCtx(c) || id < 0 || id >= SRX_T_COUNT) return fail(SRX_ERR_INVALID, "srx_tensor_shape: bad tensor id");
    const TensorInfo& ti = kTensorInfo[id];
    if (ch) *ch = uint32_t(ti.channels);
    if (h) *h = c->height * uint32_t(ti.scale);
    if (w) *w = c->width * uint32_t(ti.scale);
    return SRX_OK;
}
SRX_CATCH

srx_status srx_read_tensor(srx_ctx* c, uint32_t frame, int32_t id, void* out, size_t bytes) try
{
    if (!validCtx(c) || id < 0 || id >= SRX_T_COUNT || frame >= c->batch || !out)
        return fail(SRX_ERR_INVALID, "srx_read_tensor: bad argument");
    // The two fused kernels keep every tensor but conv_up1's output on the chip.  That one is in HBM, and
    // after srx_upscale / srx_process it still holds every frame when one pass covers the whole batch.
    if (!c->keep() && !c->generic() && id != SRX_T_CONV_UP1)
        return fail(SRX_ERR_STATE, "srx_read_tensor needs SRX_FLAG_KEEP_TENSORS at srx_create (without it only SRX_T_CONV_UP1 reaches HBM)");
    // On the fused path the workspace holds the frames of the last pass only, slot i = frame residentFirst + i.
    uint32_t slot = frame;
    if (!c->keep() && !c->generic())
    {
        if (frame < c->residentFirst || frame - c->residentFirst >= c->residentCount)
            return fail(SRX_ERR_STATE, "srx_read_tensor(SRX_T_CONV_UP1): frame " + std::to_string(frame) + " is not in the workspace, which holds "
                                       "the " + std::to_string(c->residentCount) + " frame(s) from " + std::to_string(c->residentFirst) +
                                       " on of the last pass (nothing before the first upscale; a batch above srx_frames_per_launch() "
                                       "runs in several passes that share one workspace; srx_upscale_static does not use it)");
        slot = frame - c->residentFirst;
    }
    const TensorInfo& ti = kTensorInfo[id];
    const size_t H = size_t(c->height) * ti.scale, W = size_t(c->width) * ti.scale;
    if (c->generic())
    {
        const size_t es = c->gen.elem(), C = size_t(ti.channels), n = C * H * W;
        if (bytes != n * es) return fail(SRX_ERR_INVALID, "srx_read_tensor: bytes != C*H*W*sizeof(element)");
        SRX_HIP(hipSetDevice(c->device));
        SRX_HIP(hipDeviceSynchronize());
        const uint8_t* src = static_cast<const uint8_t*>(c->gen.t[id]) + size_t(frame) * n * es;
        if (c->gen.layout != srx::ML::TensorLayout::NHWC)
        {
            SRX_HIP(hipMemcpy(out, src, n * es, hipMemcpyDeviceToHost));
            return SRX_OK;
        }
        std::vector<uint8_t> raw(n * es);
        SRX_HIP(hipMemcpy(raw.data(), src, n * es, hipMemcpyDeviceToHost));
        uint8_t* dst = static_cast<uint8_t*>(out);
        for (size_t ch = 0; ch < C; ++ch)
            for (size_t i = 0; i < H * W; ++i) std::memcpy(dst + (ch * H * W + i) * es, raw.data() + (i * C + ch) * es, es);
        return SRX_OK;
    }
    if (bytes != size_t(ti.channels) * H * W * 2) return fail(SRX_ERR_INVALID, "srx_read_tensor: bytes != C*H*W*2");
    SRX_HIP(hipSetDevice(c->device));
    SRX_HIP(hipDeviceSynchronize());
    std::vector<uint16_t> raw(c->tensorFrameBytes[id] / 2);
    SRX_HIP(hipMemcpy(raw.data(), static_cast<const uint8_t*>(c->tensorAlloc[id]) + size_t(slot) * c->tensorFrameBytes[id],
                      c->tensorFrameBytes[id], hipMemcpyDeviceToHost));
    const srx::Tensor& t = c->tensor[id];
    uint16_t* dst = static_cast<uint16_t*>(out);
    for (int ch = 0; ch < ti.channels; ++ch)
        for (size_t y = 0; y < H; ++y)
            for (size_t x = 0; x < W; ++x)
                dst[(size_t(ch) * H + y) * W + x] =
                    raw[((y + ti.border) * t.pitch + x + ti.border) * ti.storedChannels + ch];
    return SRX_OK;
}
SRX_CATCH

srx_status srx_fold_weights(const char* path, int32_t layer, uint16_t* filter, size_t filterCap,
                            uint16_t* bias, size_t biasCap, size_t* filterN, size_t* biasN) try
{
    if (!path || layer < 0 || layer >= srx::kNumConvLayers) return fail(SRX_ERR_INVALID, "srx_fold_weights: bad argument");
    srx::ML::WeightMapType weights;
    srx::FoldedWeights folded;
    if (!srx::LoadWeights(path, weights, false) || !srx::FoldWeights(weights, folded))
        return fail(SRX_ERR_WEIGHTS, std::string("loadWeights: ") + path);
    const auto& f = folded.filter[layer];
    const auto& b = folded.bias[layer];
    if (filterN) *filterN = f.size();
    if (biasN) *biasN = b.size();
    if (filter)
    {
        if (filterCap < f.size()) return fail(SRX_ERR_INVALID, "srx_fold_weights: filter buffer too small");
        std::memcpy(filter, f.data(), f.size() * 2);
    }
    if (bias && !b.empty())
    {
        if (biasCap < b.size()) return fail(SRX_ERR_INVALID, "srx_fold_weights: bias buffer too small");
        std::memcpy(bias, b.data(), b.size() * 2);
    }
    return SRX_OK;
}
SRX_CATCH

srx_status srx_packed_weights(const char* path, int32_t layer, uint16_t* out, size_t cap, size_t* n) try
{
    if (!path || layer < 0 || layer >= srx::kNumConvLayers) return fail(SRX_ERR_INVALID, "srx_packed_weights: bad argument");
    srx::ML::WeightMapType weights;
    srx::FoldedWeights folded;
    if (!srx::LoadWeights(path, weights, false) || !srx::FoldWeights(weights, folded))
        return fail(SRX_ERR_WEIGHTS, std::string("loadWeights: ") + path);
    const std::vector<uint16_t> packed = packLayer(folded, layer);
    if (n) *n = packed.size();
    if (out)
    {
        if (cap < packed.size()) return fail(SRX_ERR_INVALID, "srx_packed_weights: buffer too small");
        std::memcpy(out, packed.data(), packed.size() * 2);
    }
    return SRX_OK;
}
SRX_CATCH

srx_status srx_fused_weights(const char* path, int32_t layer, uint16_t* out, size_t cap, size_t* n) try
{
    if (!path || layer < 0 || layer >= srx::kNumConvLayers) return fail(SRX_ERR_INVALID, "srx_fused_weights: bad argument");
    srx::ML::WeightMapType weights;
    srx::FoldedWeights folded;
    if (!srx::LoadWeights(path, weights, false) || !srx::FoldWeights(weights, folded))
        return fail(SRX_ERR_WEIGHTS, std::string("loadWeights: ") + path);
    const uint32_t* sz = srx::kConvLayers[layer].sizes;
    const std::vector<uint16_t> packed =
        layer == 0 ? srx::PackConv1(folded.filter[0])
        : layer == 3 ? srx::PackConvUp1RowSharedK32(folded.filter[3])
        : layer <= 5 ? srx::PackConv3x3K32(folded.filter[layer], int(sz[0]), int(sz[1]), true)
                     : srx::PackConv6Rotating(folded.filter[6]);
    if (n) *n = packed.size();
    if (out)
    {
        if (cap < packed.size()) return fail(SRX_ERR_INVALID, "srx_fused_weights: buffer too small");
        std::memcpy(out, packed.data(), packed.size() * 2);
    }
    return SRX_OK;
}
SRX_CATCH

int32_t srx_num_stages(const srx_ctx* c) { return validCtx(c) ? int32_t(c->stages.size()) : 0; }

const char* srx_stage_name(const srx_ctx* c, int32_t i)
{
    return (validCtx(c) && i >= 0 && i < int32_t(c->stages.size())) ? kStageNames[c->stages[size_t(i)]] : "";
}

uint32_t srx_frames_per_launch(const srx_ctx* c) { return validCtx(c) ? std::min(c->chunk, c->batch) : 0; }

srx_status srx_stage_issued_flops(const srx_ctx* c, int32_t i, double* flops) try
{
    if (!validCtx(c) || i < 0 || i >= int32_t(c->stages.size()) || !flops) return fail(SRX_ERR_INVALID, "srx_stage_issued_flops: bad stage or pointer");
    const StageKind st = c->stages[size_t(i)];
    const int n = int(std::min(c->chunk, c->batch));
    if (st == ST_LR_FUSED)
    {
        const srx::StripPlan pl = srx::PlanLrFused(int(c->width), int(c->height), n, c->numCUs);
        *flops = double(pl.items) * pl.steps * srx::kLrIssuedFlopsPerStep;
    }
    else if (st == ST_HR_FUSED)
    {
        const srx::StripPlan pl = srx::PlanHrFused(2 * int(c->width), 2 * int(c->height), n, c->numCUs);
        *flops = double(pl.items) * pl.steps * srx::kHrIssuedFlopsPerStep;
    }
    else return fail(SRX_ERR_STATE, "srx_stage_issued_flops: only the two fused kernels report their issued work");
    return SRX_OK;
}
SRX_CATCH

srx_status srx_device_info(const srx_ctx* c, int32_t* computeUnits, int32_t* clockKHz, char* arch, size_t archCap) try
{
    if (!validCtx(c)) return fail(SRX_ERR_INVALID, "invalid ctx");
    hipDeviceProp_t prop;
    SRX_HIP(hipGetDeviceProperties(&prop, c->device));
    if (computeUnits) *computeUnits = prop.multiProcessorCount;
    if (clockKHz) *clockKHz = prop.clockRate;
    if (arch && archCap)
    {
        std::strncpy(arch, prop.gcnArchName, archCap - 1);
        arch[archCap - 1] = 0;
    }
    return SRX_OK;
}
SRX_CATCH

srx_status srx_launch_stage(srx_ctx* c, int32_t i, void* stream) try
{
    if (!validCtx(c) || i < 0 || i >= int32_t(c->stages.size())) return fail(SRX_ERR_INVALID, "srx_launch_stage: bad stage");
    SRX_HIP(hipSetDevice(c->device));
    const uint32_t count = (c->stages[size_t(i)] == ST_PACK) ? c->batch : std::min(c->chunk, c->batch);
    return launchStage(c, c->stages[size_t(i)], 0, count, static_cast<hipStream_t>(stream));
}
SRX_CATCH

// ---------------------------------------------------------------------------
// Generic layers on caller tensors.
// ---------------------------------------------------------------------------
namespace {
bool utilFor(int32_t dtype, int32_t layout, srx::ML::Util& out)
{
    if ((dtype != 0 && dtype != 1) || (layout != 0 && layout != 1)) return false;
    out = srx::ML::Util(dtype == 0 ? srx::ML::TensorDataType::FLOAT16 : srx::ML::TensorDataType::FLOAT32,
                        layout == 0 ? srx::ML::TensorLayout::DEFAULT : srx::ML::TensorLayout::NHWC);
    return true;
}
}  // namespace

srx_status srx_layer_plan(int32_t kind, int32_t dtype, const uint32_t inSizes[4], const uint32_t param[4],
                          uint32_t outSizes[4], uint64_t* inBytes, uint64_t* outBytes) try
{
    srx::ML::Util util;
    srx::ML::Operator op;
    uint64_t ib = inBytes ? *inBytes : 0, ob = outBytes ? *outBytes : 0;
    uint32_t os[4] = {};
    bool ok = inSizes && utilFor(dtype, 0, util);
    if (ok && kind == 0) ok = param && util.CreateConvolutionLayer(inSizes, param, true, ib, ob, os, op);
    else if (ok && kind == 1) ok = param && util.CreateUpsampleLayer(inSizes, ib, ob, os, op, param[0], param[1]);
    else if (ok && kind == 2)
    {
        ok = util.CreateAdditionLayer(inSizes, op);
        if (ok) std::memcpy(os, inSizes, sizeof(os));
    }
    else ok = false;
    if (!ok) return fail(SRX_ERR_INVALID, "srx_layer_plan: the layer factory rejected these sizes");
    if (outSizes) std::memcpy(outSizes, os, sizeof(os));
    if (inBytes) *inBytes = ib;
    if (outBytes) *outBytes = ob;
    return SRX_OK;
}
SRX_CATCH

srx_status srx_layer_conv2d(int32_t dtype, int32_t layout, const void* in, const uint32_t inSizes[4], const void* filter,
                            const uint32_t filterSizes[4], const void* bias, void* out, void* stream) try
{
    srx::ML::Util util;
    srx::ML::Operator op;
    uint64_t ib = 0, ob = 0;
    uint32_t outSizes[4];
    if (!in || !inSizes || !filter || !filterSizes || !out || !utilFor(dtype, layout, util) ||
        filterSizes[2] > 7 || filterSizes[3] > 7 ||
        !util.CreateConvolutionLayer(inSizes, filterSizes, bias != nullptr, ib, ob, outSizes, op))
        return fail(SRX_ERR_INVALID, "srx_layer_conv2d: bad pointer, data type, layout or sizes (filter depth must equal the input channels, sides <= 7)");
    SRX_HIP(srx::DispatchOperator(op, in, filter, bias, out, static_cast<hipStream_t>(stream)));
    return SRX_OK;
}
SRX_CATCH

srx_status srx_layer_upsample2d(int32_t dtype, int32_t layout, const void* in, const uint32_t inSizes[4], uint32_t scaleX,
                                uint32_t scaleY, int32_t interpolation, void* out, void* stream) try
{
    srx::ML::Util util;
    srx::ML::Operator op;
    uint64_t ib = 0, ob = 0;
    uint32_t outSizes[4];
    if (!in || !inSizes || !out || !utilFor(dtype, layout, util) || (interpolation != 0 && interpolation != 1) ||
        !util.CreateUpsampleLayer(inSizes, ib, ob, outSizes, op, scaleX, scaleY,
                                  interpolation ? srx::ML::InterpolationType::LINEAR : srx::ML::InterpolationType::NEAREST_NEIGHBOR))
        return fail(SRX_ERR_INVALID, "srx_layer_upsample2d: bad pointer, data type, layout, sizes, scale or interpolation");
    SRX_HIP(srx::DispatchOperator(op, in, nullptr, nullptr, out, static_cast<hipStream_t>(stream)));
    return SRX_OK;
}
SRX_CATCH

srx_status srx_layer_add(int32_t dtype, const void* a, const void* b, void* out, size_t count, void* stream) try
{
    srx::ML::Util util;
    srx::ML::Operator op;
    const uint32_t sizes[4] = {1, 1, uint32_t(count >> 20) + 1, 1u << 20};   // only the element count matters
    if (!a || !b || !out || count == 0 || !utilFor(dtype, 0, util) || !util.CreateAdditionLayer(sizes, op))
        return fail(SRX_ERR_INVALID, "srx_layer_add: bad pointer, data type or count");
    op.inputSizes[0] = op.inputSizes[1] = op.inputSizes[2] = 1;
    if (count >= (size_t(1) << 32)) return fail(SRX_ERR_INVALID, "srx_layer_add: count must be below 2^32");
    op.inputSizes[3] = uint32_t(count);
    SRX_HIP(srx::DispatchOperator(op, a, b, nullptr, out, static_cast<hipStream_t>(stream)));
    return SRX_OK;
}
SRX_CATCH

// ---------------------------------------------------------------------------
// Frame pipe: pinned staging buffers and three streams around srx_upscale_frames.
// ---------------------------------------------------------------------------
struct srx_pipe
{
    srx_ctx* ctx = nullptr;
    int device = 0;                          // the ctx's device, kept here so that destruction never reads the ctx's fields
    bool counted = false;                    // this pipe is included in ctx->livePipes
    uint32_t perSlot = 0, slots = 0;
    size_t inPitch = 0, inFrame = 0, outPitch = 0, outFrame = 0;    // host layouts (tightly packed)
    std::vector<uint8_t*> hostIn, hostOut;
    uint32_t inChannels = 4, outChannels = 4;
    std::vector<uint8_t*> devRgbIn, devRgbOut;                       // per slot, RGB8 mode only
    hipStream_t up = nullptr, comp = nullptr, down = nullptr;
    std::vector<hipEvent_t> evUp, evComp, evDown;
    std::vector<bool> submitted;
};

srx_status srx_pipe_destroy(srx_pipe* p) try
{
    if (!p) return SRX_OK;
    (void)hipSetDevice(p->device);
    for (hipStream_t st : {p->up, p->comp, p->down})
        if (st) (void)hipStreamSynchronize(st);
    if (p->counted && p->ctx) --p->ctx->livePipes;
    for (uint8_t* h : p->hostIn) if (h) (void)hipHostFree(h);
    for (uint8_t* h : p->hostOut) if (h) (void)hipHostFree(h);
    for (uint8_t* d : p->devRgbIn) if (d) (void)hipFree(d);
    for (uint8_t* d : p->devRgbOut) if (d) (void)hipFree(d);
    for (auto* v : {&p->evUp, &p->evComp, &p->evDown})
        for (hipEvent_t e : *v) if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {p->up, p->comp, p->down})
        if (st) (void)hipStreamDestroy(st);
    delete p;
    return SRX_OK;
}
SRX_CATCH

srx_status srx_pipe_create(srx_ctx* c, uint32_t framesPerSlot, srx_pipe** out) try
{
    if (!out) return fail(SRX_ERR_INVALID, "srx_pipe_create: out is NULL");
    *out = nullptr;
    if (!validCtx(c) || framesPerSlot == 0 || c->batch / framesPerSlot < 2)
        return fail(SRX_ERR_INVALID, "srx_pipe_create: the ctx's batch must hold at least two slots of frames_per_slot frames");
    if (c->keep() || c->generic()) return fail(SRX_ERR_STATE, "srx_pipe_create: not available with SRX_FLAG_KEEP_TENSORS or the generic-layer graph");
    SRX_HIP(hipSetDevice(c->device));
    struct Destroy { void operator()(srx_pipe* x) const { srx_pipe_destroy(x); } };
    std::unique_ptr<srx_pipe, Destroy> owner(new srx_pipe);    // released on success only
    srx_pipe* p = owner.get();
    p->ctx = c;
    p->device = c->device;
    p->perSlot = framesPerSlot;
    p->slots = c->batch / framesPerSlot;
    p->inPitch = size_t(c->width) * 4;
    p->inFrame = p->inPitch * c->height;
    p->outPitch = size_t(c->width) * 8;
    p->outFrame = p->outPitch * c->height * 2;
    p->hostIn.assign(p->slots, nullptr);
    p->hostOut.assign(p->slots, nullptr);
    p->evUp.assign(p->slots, nullptr);
    p->evComp.assign(p->slots, nullptr);
    p->evDown.assign(p->slots, nullptr);
    p->submitted.assign(p->slots, false);
    const auto bail = [&](hipError_t e, const char* what) {
        const std::string msg = std::string(what) + ": " + hipGetErrorString(e);
        return fail(e == hipErrorOutOfMemory ? SRX_ERR_NOMEM : SRX_ERR_HIP, msg);
    };
    hipError_t e;
    // The copies of one slot must run beside the kernels of another, which they only do from different hardware
    // queues.  The runtime deals a process's streams over a small pool of hardware queues per priority level (four
    // by default), so in a process with other streams of its own the three streams of a pipe can come to share one,
    // and then copies and kernels take turns (measured: 870 instead of 1540 1080p frames/s with ONE more stream
    // in the process).  The copy streams are therefore created at the highest priority, which has a pool of its
    // own; the compute stream stays at the default priority.
    int leastPriority = 0, greatestPriority = 0;
    if ((e = hipDeviceGetStreamPriorityRange(&leastPriority, &greatestPriority)) != hipSuccess) return bail(e, "hipDeviceGetStreamPriorityRange");
    if ((e = hipStreamCreateWithFlags(&p->comp, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreateWithFlags");
    for (hipStream_t* st : {&p->up, &p->down})
        if ((e = hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatestPriority)) != hipSuccess) return bail(e, "hipStreamCreateWithPriority");
    for (uint32_t s = 0; s < p->slots; ++s)
    {
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&p->hostIn[s]), p->inFrame * framesPerSlot, hipHostMallocDefault)) != hipSuccess)
            return bail(e, "hipHostMalloc(input)");
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&p->hostOut[s]), p->outFrame * framesPerSlot, hipHostMallocDefault)) != hipSuccess)
            return bail(e, "hipHostMalloc(output)");
        for (hipEvent_t* ev : {&p->evUp[s], &p->evComp[s], &p->evDown[s]})
            if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreateWithFlags");
    }
    ++c->livePipes;
    p->counted = true;
    *out = owner.release();
    return SRX_OK;
}
SRX_CATCH

uint32_t srx_pipe_slots(const srx_pipe* p) { return p ? p->slots : 0; }

srx_status srx_pipe_set_channels(srx_pipe* p, uint32_t inChannels, uint32_t outChannels) try
{
    if (!p || (inChannels != 3 && inChannels != 4) || (outChannels != 3 && outChannels != 4))
        return fail(SRX_ERR_INVALID, "srx_pipe_set_channels: channels must be 3 (RGB8) or 4 (RGBA8)");
    for (uint32_t s = 0; s < p->slots; ++s)
        if (p->submitted[s]) return fail(SRX_ERR_STATE, "srx_pipe_set_channels: call it before the first srx_pipe_submit");
    srx_ctx* c = p->ctx;
    SRX_HIP(hipSetDevice(c->device));
    // RGB8 staging buffers, one per slot; a failed allocation leaves none behind, so a later call starts over
    const auto allocPerSlot = [&](std::vector<uint8_t*>& v, size_t bytes) -> hipError_t {
        v.assign(p->slots, nullptr);
        for (uint32_t s = 0; s < p->slots; ++s)
        {
            const hipError_t e = hipMalloc(reinterpret_cast<void**>(&v[s]), bytes);
            if (e != hipSuccess)
            {
                for (uint8_t* d : v) if (d) (void)hipFree(d);
                v.clear();
                return e;
            }
        }
        return hipSuccess;
    };
    if (inChannels == 3 && p->devRgbIn.empty())
        SRX_HIP(allocPerSlot(p->devRgbIn, size_t(c->width) * c->height * 3 * p->perSlot + 4));     // + 4: the last pixel is fetched as a 32-bit word
    if (outChannels == 3 && p->devRgbOut.empty())
        SRX_HIP(allocPerSlot(p->devRgbOut, size_t(c->width) * c->height * 12 * p->perSlot));
    p->inChannels = inChannels;
    p->outChannels = outChannels;
    p->inPitch = size_t(c->width) * inChannels;
    p->inFrame = p->inPitch * c->height;
    p->outPitch = size_t(c->width) * 2 * outChannels;
    p->outFrame = p->outPitch * c->height * 2;
    return SRX_OK;
}
SRX_CATCH

srx_status srx_pipe_input(srx_pipe* p, uint32_t slot, uint8_t** host, size_t* pitch, size_t* frameStride) try
{
    if (!p || slot >= p->slots || !host) return fail(SRX_ERR_INVALID, "srx_pipe_input: bad pipe, slot or pointer");
    *host = p->hostIn[slot];
    if (pitch) *pitch = p->inPitch;
    if (frameStride) *frameStride = p->inFrame;
    return SRX_OK;
}
SRX_CATCH

srx_status srx_pipe_submit(srx_pipe* p, uint32_t slot) try
{
    if (!p || slot >= p->slots) return fail(SRX_ERR_INVALID, "srx_pipe_submit: bad pipe or slot");
    srx_ctx* c = p->ctx;
    SRX_HIP(hipSetDevice(c->device));
    const uint32_t first = slot * p->perSlot;
    // the slot's device input may still be read by its previous upscale, its device output by its
    // previous download
    if (p->submitted[slot])
    {
        SRX_HIP(hipStreamWaitEvent(p->up, p->evComp[slot], 0));
        SRX_HIP(hipStreamWaitEvent(p->comp, p->evDown[slot], 0));
    }
    // Both sides of every copy are contiguous (tightly packed host frames; ctx images with pitch = width x 4 and
    // frames back to back), so each slot moves as ONE linear copy per direction: ROCm 7.2's runtime gives linear copies
    // between page-locked and device memory to the DMA engines, which run beside the kernels (the runtime bundled
    // with torch 2.10 does that for uploads only and downloads with a shader kernel, which shares the CUs with the
    // persistent kernels: with three slots in flight it still hides, with more it does not); rectangular copies may
    // be done by a copy kernel in either runtime.
    if (p->inChannels == 3)
        SRX_HIP(hipMemcpyAsync(p->devRgbIn[slot], p->hostIn[slot], p->inFrame * p->perSlot, hipMemcpyHostToDevice, p->up));
    else
        SRX_HIP(hipMemcpyAsync(c->input.base + size_t(first) * c->input.frame, p->hostIn[slot], p->inFrame * p->perSlot,
                               hipMemcpyHostToDevice, p->up));
    SRX_HIP(hipEventRecord(p->evUp[slot], p->up));
    SRX_HIP(hipStreamWaitEvent(p->comp, p->evUp[slot], 0));
    // RGB8 buffers are read and written by the two fused kernels themselves (3 bytes per pixel, tightly packed
    // frames): no repacking pass, which would cost two more trips through memory per frame on the compute stream.
    srx::Image rgbIn, rgbOut;
    if (p->inChannels == 3)
    {
        rgbIn.base = p->devRgbIn[slot];
        rgbIn.pitch = int64_t(p->inPitch);
        rgbIn.frame = int64_t(p->inFrame);
        rgbIn.pixelBytes = 3;
    }
    if (p->outChannels == 3)
    {
        rgbOut.base = p->devRgbOut[slot];
        rgbOut.pitch = int64_t(p->outPitch);
        rgbOut.frame = int64_t(p->outFrame);
        rgbOut.pixelBytes = 3;
    }
    if (const srx_status s = upscaleFramesOn(c, first, p->perSlot, p->comp, p->inChannels == 3 ? &rgbIn : nullptr,
                                             p->outChannels == 3 ? &rgbOut : nullptr))
        return s;
    SRX_HIP(hipEventRecord(p->evComp[slot], p->comp));
    SRX_HIP(hipStreamWaitEvent(p->down, p->evComp[slot], 0));
    if (p->outChannels == 3)
        SRX_HIP(hipMemcpyAsync(p->hostOut[slot], p->devRgbOut[slot], p->outFrame * p->perSlot, hipMemcpyDeviceToHost, p->down));
    else
        SRX_HIP(hipMemcpyAsync(p->hostOut[slot], c->output.base + size_t(first) * c->output.frame, p->outFrame * p->perSlot,
                               hipMemcpyDeviceToHost, p->down));
    SRX_HIP(hipEventRecord(p->evDown[slot], p->down));
    p->submitted[slot] = true;
    return SRX_OK;
}
SRX_CATCH

srx_status srx_pipe_wait(srx_pipe* p, uint32_t slot, const uint8_t** host, size_t* pitch, size_t* frameStride) try
{
    if (!p || slot >= p->slots) return fail(SRX_ERR_INVALID, "srx_pipe_wait: bad pipe or slot");
    if (!p->submitted[slot]) return fail(SRX_ERR_STATE, "srx_pipe_wait: the slot was never submitted");
    SRX_HIP(hipSetDevice(p->device));
    SRX_HIP(hipEventSynchronize(p->evDown[slot]));
    if (host) *host = p->hostOut[slot];
    if (pitch) *pitch = p->outPitch;
    if (frameStride) *frameStride = p->outFrame;
    return SRX_OK;
}
SRX_CATCH

}  // extern "C"
