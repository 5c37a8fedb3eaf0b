#include "SuperResolution.h"

#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <stdexcept>

#include "ImageIO.h"

using namespace std;

void RenderTarget::Create(uint32_t w, uint32_t h)
{
    width = w;
    height = h;
    rowPitch = (w * 4 + 255u) & ~255u;        // D3D12 readback rows are 256-byte aligned
    pixels.assign(size_t(rowPitch) * h, 0);
}

SuperResolution::SuperResolution() : m_ctx(nullptr), m_imageLayoutIn{}, m_imageLayoutOut{} {}

SuperResolution::~SuperResolution()
{
    srx_destroy(m_ctx);
}

bool SuperResolution::Init(void* stream, int device, uint32_t vendorId, const char* fileName, bool isFP16Supported)
{
    // Content/SuperResolution.cpp:155-163: NHWC tensors for the vendor that prefers them, fp32 tensors
    // without fp16 support.  Either choice runs the graph of generic layers; the default (fp16, any
    // other vendor) runs the fused MFMA kernels, whose internal layout is their own.
    uint32_t flags = SRX_FLAG_VERBOSE_WEIGHTS;
    if (!isFP16Supported) flags |= SRX_FLAG_FP32;
    if (vendorId == 0x10DE) flags |= SRX_FLAG_NHWC;

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

bool SuperResolution::createResources(int device, uint32_t flags)
{
    // "Assets/weights.bin" relative to the working directory, as in the reference; SRX_ASSETS names
    // another directory.
    string path = "Assets/weights.bin";
    if (const char* dir = getenv("SRX_ASSETS")) path = string(dir) + "/weights.bin";
    const srx_status s = srx_create(&m_ctx, device, path.c_str(), m_imageLayoutIn.Width, m_imageLayoutIn.Height, 1, flags);
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
    if (srx_tensor_to_image(m_ctx, 0, pRenderTarget->pixels.data(), pRenderTarget->rowPitch, stream) != SRX_OK)
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
