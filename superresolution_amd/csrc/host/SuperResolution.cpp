#include "SuperResolution.h"

#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <stdexcept>

#include <cstring>

#include <hip/hip_runtime.h>

#include "ImageIO.h"

using namespace std;

void RenderTarget::release()
{
    if (pixels && pinned) (void)hipHostFree(pixels);
    else std::free(pixels);
    pixels = nullptr;
    size = 0;
    pinned = false;
}

RenderTarget::~RenderTarget()
{
    release();
}

void RenderTarget::Create(uint32_t w, uint32_t h)
{
    release();
    width = w;
    height = h;
    rowPitch = (w * 4 + 255u) & ~255u;        // D3D12 readback rows are 256-byte aligned
    size = size_t(rowPitch) * h;
    void* p = nullptr;
    if (hipHostMalloc(&p, size, hipHostMallocDefault) != hipSuccess || !p)
        throw std::runtime_error("RenderTarget::Create: hipHostMalloc failed");
    pixels = static_cast<uint8_t*>(p);
    pinned = true;
    std::memset(pixels, 0, size);
}

RenderTarget::RenderTarget(const RenderTarget& other)
{
    *this = other;
}

RenderTarget& RenderTarget::operator=(const RenderTarget& other)
{
    if (this == &other) return *this;
    release();
    width = other.width;
    height = other.height;
    rowPitch = other.rowPitch;
    size = other.size;
    if (size)
    {
        pixels = static_cast<uint8_t*>(std::malloc(size));
        if (!pixels) throw std::bad_alloc();
        std::memcpy(pixels, other.pixels, size);
    }
    return *this;
}

SuperResolution::SuperResolution() : m_ctx(nullptr), m_pipe(nullptr), m_framesPerSlot(0), m_imageLayoutIn{}, m_imageLayoutOut{} {}

SuperResolution::~SuperResolution()
{
    srx_pipe_destroy(m_pipe);           // the pipe refers to the ctx: it goes first
    srx_destroy(m_ctx);
}

bool SuperResolution::Init(void* stream, int device, uint32_t vendorId, const char* fileName, bool isFP16Supported)
{
    // Content/SuperResolution.cpp:155-163: NHWC tensors for the vendor that prefers them, fp32 tensors
    // without fp16 support.  Either choice runs the graph of generic layers; the default (fp16, any
    // other vendor) runs the fused MFMA kernels, whose internal layout is their own.
    if (m_ctx)
    {
        cerr << "SuperResolution: already initialised" << endl;        // a second ctx would orphan the first
        return false;
    }
    uint32_t flags = SRX_FLAG_VERBOSE_WEIGHTS;
    if (!isFP16Supported) flags |= SRX_FLAG_FP32;
    if (vendorId == 0x10DE) flags |= SRX_FLAG_NHWC;

    if (!loadInputImage(fileName)) return false;
    const uint32_t width = m_imageLayoutIn.Width;

    if (!createResources(device, flags)) return false;
    if (srx_upload_rgba8(m_ctx, 0, m_inputImage.data(), size_t(width) * 4, stream) != SRX_OK ||
        srx_sync(m_ctx, stream) != SRX_OK)
    {
        cerr << "SuperResolution: " << srx_last_error() << endl;
        return false;
    }

    m_imageLayoutOut.Width = m_imageLayoutIn.Width << 1;
    m_imageLayoutOut.Height = m_imageLayoutIn.Height << 1;
    m_imageLayoutOut.UseNhwc = m_imageLayoutIn.UseNhwc = (flags & SRX_FLAG_NHWC) != 0;
    return true;
}

bool SuperResolution::loadInputImage(const char* fileName)
{
    // Load input image: 3-channel files become RGBA, 1-/2-channel files keep their channels and
    // the missing ones read as 0 (alpha 1) like a D3D R8 / R8G8 texture sampled as float3.
    int width = 0, height = 0, reqChannels = 0;
    const vector<uint8_t> texData = srx::LoadImageFromFile(fileName, width, height, reqChannels);
    if (texData.empty())
    {
        cerr << "Unable to load image: " << fileName << " (" << srx::LastImageError() << ")" << endl;
        return false;
    }
    m_inputImage.assign(size_t(width) * height * 4, 0);
    for (size_t i = 0; i < size_t(width) * height; ++i)
    {
        uint8_t* d = &m_inputImage[i * 4];
        const uint8_t* s = &texData[i * reqChannels];
        d[0] = s[0];
        d[1] = reqChannels >= 2 ? s[1] : 0;
        d[2] = reqChannels >= 3 ? s[2] : 0;
        d[3] = reqChannels >= 4 ? s[3] : 255;
    }
    m_imageLayoutIn.Width = static_cast<uint32_t>(width);
    m_imageLayoutIn.Height = static_cast<uint32_t>(height);
    return true;
}

bool SuperResolution::InitStream(int device, uint32_t vendorId, const char* fileName, bool isFP16Supported,
                                 uint32_t framesInFlight, uint32_t framesPerSlot)
{
    if (m_ctx)
    {
        cerr << "SuperResolution: already initialised" << endl;
        return false;
    }
    if (!isFP16Supported || vendorId == 0x10DE)
    {
        cerr << "SuperResolution: the frame stream runs the fused fp16 kernels only (no -fp32 / -nhwc)" << endl;
        return false;
    }
    if (framesInFlight < 2 || framesPerSlot == 0)
    {
        cerr << "SuperResolution: a frame stream needs at least two slots in flight and one frame per slot" << endl;
        return false;
    }
    if (!loadInputImage(fileName)) return false;
    const uint32_t w = m_imageLayoutIn.Width, h = m_imageLayoutIn.Height;
    if (!createResources(device, SRX_FLAG_VERBOSE_WEIGHTS, framesInFlight * framesPerSlot)) return false;
    if (srx_pipe_create(m_ctx, framesPerSlot, &m_pipe) != SRX_OK || srx_pipe_set_channels(m_pipe, 3, 3) != SRX_OK)
    {
        cerr << "SuperResolution: " << srx_last_error() << endl;
        return false;
    }
    m_framesPerSlot = framesPerSlot;
    // every slot starts out holding the decoded file in each of its frames
    for (uint32_t slot = 0; slot < srx_pipe_slots(m_pipe); ++slot)
    {
        size_t pitch = 0, stride = 0;
        uint8_t* in = GetStreamInput(slot, &pitch, &stride);
        for (uint32_t f = 0; f < framesPerSlot; ++f)
            for (uint32_t y = 0; y < h; ++y)
            {
                uint8_t* d = in + f * stride + y * pitch;
                const uint8_t* s = &m_inputImage[size_t(y) * w * 4];
                for (uint32_t x = 0; x < w; ++x)
                {
                    d[3 * x] = s[4 * x];
                    d[3 * x + 1] = s[4 * x + 1];
                    d[3 * x + 2] = s[4 * x + 2];
                }
            }
    }
    m_imageLayoutOut.Width = w << 1;
    m_imageLayoutOut.Height = h << 1;
    m_imageLayoutOut.UseNhwc = m_imageLayoutIn.UseNhwc = false;
    return true;
}

uint8_t* SuperResolution::GetStreamInput(uint32_t slot, size_t* pRowPitch, size_t* pFrameStride)
{
    uint8_t* host = nullptr;
    if (srx_pipe_input(m_pipe, slot, &host, pRowPitch, pFrameStride) != SRX_OK) throw std::runtime_error(srx_last_error());
    return host;
}

void SuperResolution::SubmitFrames(uint32_t slot)
{
    if (srx_pipe_submit(m_pipe, slot) != SRX_OK) throw std::runtime_error(srx_last_error());
}

const uint8_t* SuperResolution::WaitFrames(uint32_t slot, size_t* pRowPitch, size_t* pFrameStride)
{
    const uint8_t* host = nullptr;
    if (srx_pipe_wait(m_pipe, slot, &host, pRowPitch, pFrameStride) != SRX_OK) throw std::runtime_error(srx_last_error());
    return host;
}

bool SuperResolution::createResources(int device, uint32_t flags, uint32_t batch)
{
    // "Assets/weights.bin" relative to the working directory, as in the reference; SRX_ASSETS names
    // another directory.
    string path = "Assets/weights.bin";
    if (const char* dir = getenv("SRX_ASSETS")) path = string(dir) + "/weights.bin";
    const srx_status s = srx_create(&m_ctx, device, path.c_str(), m_imageLayoutIn.Width, m_imageLayoutIn.Height, batch, flags);
    if (s == SRX_ERR_WEIGHTS) throw std::runtime_error("loadWeights");
    if (s != SRX_OK)
    {
        cerr << "SuperResolution: " << srx_last_error() << endl;
        return false;
    }
    return true;
}

void SuperResolution::ImageToTensors(void* stream)
{
    if (srx_image_to_tensor(m_ctx, stream) != SRX_OK) throw std::runtime_error(srx_last_error());
}

void SuperResolution::Process(void* stream)
{
    if (srx_process(m_ctx, stream) != SRX_OK) throw std::runtime_error(srx_last_error());
}

void SuperResolution::Render(void* stream, RenderTarget* pRenderTarget)
{
    if (srx_tensor_to_image(m_ctx, 0, pRenderTarget->pixels, pRenderTarget->rowPitch, stream) != SRX_OK)
        throw std::runtime_error(srx_last_error());
}

uint32_t SuperResolution::GetOutWidth() const
{
    return m_imageLayoutOut.Width;
}

uint32_t SuperResolution::GetOutHeight() const
{
    return m_imageLayoutOut.Height;
}
