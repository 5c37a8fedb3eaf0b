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

// R8G8B8A8_UNORM back buffer in host memory.  Create() takes page-locked memory, so that the copy out of the
// GPU is asynchronous and several render targets can be in flight (the reference's swap-chain back buffers); a
// copy of a RenderTarget (the screen-shot read-back buffer) is an ordinary pageable one.
struct RenderTarget
{
    uint32_t width = 0, height = 0;
    uint32_t rowPitch = 0;               // bytes
    uint8_t* pixels = nullptr;
    size_t size = 0;

    RenderTarget() = default;
    RenderTarget(const RenderTarget& other);
    RenderTarget& operator=(const RenderTarget& other);
    ~RenderTarget();
    void Create(uint32_t w, uint32_t h);

private:
    bool pinned = false;
    void release();
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

    // ---- Stream mode: slots of frames in flight between host memory and the GPU.
    // The reference keeps FrameCount frames in flight, each with its own allocator, back buffer and fence value
    // (SuperResolutionX.h:49,69-72; SuperResolutionX.cpp:388-405).  Here a slot is `framesPerSlot` video frames
    // with page-locked input and output buffers; SubmitFrames enqueues upload, ImageToTensors + Process and the
    // copy back of one slot on three streams and returns at once, WaitFrames is the fence wait of that slot.
    // Frames are RGB8, tightly packed, on both sides (the network reads no alpha, SaveImage keeps none).
    bool InitStream(int device, uint32_t vendorId, const char* fileName, bool isFP16Supported,
                    uint32_t framesInFlight, uint32_t framesPerSlot);
    uint32_t GetFramesPerSlot() const { return m_framesPerSlot; }
    // The slot's input frames: the caller (a video source) writes them before SubmitFrames.  InitStream has
    // filled every slot with the decoded file.
    uint8_t* GetStreamInput(uint32_t slot, size_t* pRowPitch, size_t* pFrameStride);
    void SubmitFrames(uint32_t slot);
    // Blocks until the slot's output frames are in host memory; valid until the slot is submitted again.
    const uint8_t* WaitFrames(uint32_t slot, size_t* pRowPitch, size_t* pFrameStride);

protected:
    struct ImageLayout
    {
        uint32_t Height;
        uint32_t Width;
        bool UseNhwc;
    };

    bool createResources(int device, uint32_t flags, uint32_t batch = 1);
    bool loadInputImage(const char* fileName);

    srx_ctx* m_ctx;
    srx_pipe* m_pipe;
    uint32_t m_framesPerSlot;
    std::vector<uint8_t> m_inputImage;   // decoded file, RGBA8
    ImageLayout m_imageLayoutIn;
    ImageLayout m_imageLayoutOut;
};
