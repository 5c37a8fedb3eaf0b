// The model class of the sample, on HIP.  Same public methods and call order as
// the reference (Content/SuperResolution.h:10-25):
//   Init -> [ImageToTensors -> Process]* -> Render*, GetOutWidth / GetOutHeight.
// The D3D12 handles of the reference's signatures have no meaning here; their
// counterparts are
//   XUSG::CommandList*            -> a HIP stream (void*, NULL = default stream)
//   XUSG::ML::CommandRecorder*    -> nothing: kernels are launched directly
//   DescriptorTableLib, uploaders -> nothing: the ctx owns device memory, uploads are synchronous
//   XUSG::RenderTarget*           -> RenderTarget below, an RGBA8 host buffer (the swap-chain back buffer)
// Everything on the device goes through the C ABI in include/srx.h.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/srx.h"

// R8G8B8A8_UNORM back buffer in host memory.
struct RenderTarget
{
    uint32_t width = 0, height = 0;
    uint32_t rowPitch = 0;               // bytes
    std::vector<uint8_t> pixels;
    void Create(uint32_t w, uint32_t h);
};

class SuperResolution
{
public:
    SuperResolution();
    virtual ~SuperResolution();

    // As in the reference (Content/SuperResolution.cpp:155-163) vendorId 0x10DE selects NHWC tensors
    // and isFP16Supported == false fp32 tensors; both run the graph made of generic layers.  The
    // default runs the fused fp16 MFMA kernels.
    bool Init(void* stream, int device, uint32_t vendorId, const char* fileName, bool isFP16Supported = true);

    void ImageToTensors(void* stream);
    void Process(void* stream);
    void Render(void* stream, RenderTarget* pRenderTarget);

    uint32_t GetOutWidth() const;
    uint32_t GetOutHeight() const;

protected:
    struct ImageLayout
    {
        uint32_t Height;
        uint32_t Width;
        bool UseNhwc;
    };

    bool createResources(int device, uint32_t flags);

    srx_ctx* m_ctx;
    std::vector<uint8_t> m_inputImage;   // decoded file, RGBA8
    ImageLayout m_imageLayoutIn;
    ImageLayout m_imageLayoutOut;
};
