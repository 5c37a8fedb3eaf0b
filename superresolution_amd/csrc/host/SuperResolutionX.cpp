// Adapted from StarsX/SuperResolution (SuperResolution/SuperResolutionX.cpp), whose file header reads
// "Copyright (c) Microsoft. All rights reserved.  This code is licensed under the MIT License (MIT).":
// OnKeyUp, the command-line switches, the screen-shot sequence and the frame statistics follow the reference so that
// the class stays a drop-in; device, queue, swap chain and fences are replaced by a HIP device index and one stream.
#include "SuperResolutionX.h"

#include <cwchar>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cassert>
#include <ctime>
#include <cwctype>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <vector>

#include "ImageIO.h"

using namespace std;

namespace {
// The reference turns a failed bool into ThrowIfFailed(E_FAIL) -> HrException
// (Common/DXFrameworkHelper.h:27-44); here the same failures surface as runtime_error.
void ThrowIfFailed(bool ok, const char* what)
{
    if (!ok) throw std::runtime_error(what);
}
}  // namespace

SuperResolutionX::SuperResolutionX(uint32_t width, uint32_t height, std::wstring name) :
    DXFramework(width, height, name),
    m_vendorId(0x1002),
    m_isFP16Supported(true),
    m_deviceIndex(0),
    m_stream(nullptr),
    m_frameIndex(0),
    m_fenceEvents{},
    m_framePending{},
    m_framesPerSlot(0),
    m_framesCompleted(0),
    m_slotsCompleted(0),
    m_lastOutput(nullptr),
    m_deviceType(DEVICE_DISCRETE),
    m_updateImage(false),
    m_showFPS(true),
    m_pausing(false),
    m_isFirstFrame(true),
    m_fileName("Assets/Sashimi128.png"),
    m_screenShot(0),
    m_frameCnt(0),
    m_lastFps(0.0)
{
}

SuperResolutionX::~SuperResolutionX()
{
}

void SuperResolutionX::OnInit()
{
    LoadPipeline();
    LoadAssets();
}

// Load the rendering pipeline dependencies: pick the device, create the stream, build the model,
// size the render targets to the 2x output.
void SuperResolutionX::LoadPipeline()
{
    // The reference's -warp switch selects the Windows software rasteriser.  There is no software
    // adapter here: the library refuses to run without a gfx950 GPU rather than fall back.
    if (m_deviceType == DEVICE_WARP)
        throw std::runtime_error("-warp: no software adapter exists in the HIP build; run on an MI355X");

    int count = 0;
    ThrowIfFailed(hipGetDeviceCount(&count) == hipSuccess && count > 0, "no HIP device found");
    ThrowIfFailed(m_deviceIndex >= 0 && m_deviceIndex < count, "device index out of range");
    ThrowIfFailed(hipSetDevice(m_deviceIndex) == hipSuccess, "hipSetDevice failed");
    // the stream of the single-image mode; the frame stream brings its own three (upload, compute, download)
    hipStream_t stream = nullptr;
    if (!m_framesPerSlot)
        ThrowIfFailed(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess, "hipStreamCreate failed");
    m_stream = stream;

    ThrowIfFailed(InitTensors(), "InitTensors failed");
    m_width = m_superResolution->GetOutWidth();
    m_height = m_superResolution->GetOutHeight();

    // Create frame resources: one render target and one fence event per frame in flight.  In stream mode the
    // slots' page-locked output buffers are the render targets.
    for (uint8_t n = 0; n < FrameCount; ++n)
    {
        if (!m_framesPerSlot) m_renderTargets[n].Create(m_width, m_height);
        hipEvent_t ev = nullptr;
        ThrowIfFailed(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
        m_fenceEvents[n] = ev;
    }
}

// Load the sample assets: wait until the uploads of Init have reached the GPU.
void SuperResolutionX::LoadAssets()
{
    WaitForGpu();
    m_startTime = m_statsTime = chrono::steady_clock::now();
}

bool SuperResolutionX::InitTensors()
{
    m_superResolution = make_unique<SuperResolution>();
    if (m_framesPerSlot)
        return m_superResolution->InitStream(m_deviceIndex, m_vendorId, m_fileName.c_str(), m_isFP16Supported, FrameCount, m_framesPerSlot);
    // fp16 tensors are always supported on gfx950 (the reference queries DirectML for this)
    return m_superResolution->Init(m_stream, m_deviceIndex, m_vendorId, m_fileName.c_str(), m_isFP16Supported);
}

// Update frame-based values.
void SuperResolutionX::OnUpdate()
{
    CalculateFrameStats();
}

// Render the scene.
void SuperResolutionX::OnRender()
{
    // Record and submit all the work of this frame.
    PopulateCommandList();

    // "Present": the frame now sits in its render target.
    MoveToNextFrame();
}

void SuperResolutionX::OnDestroy()
{
    // Ensure that the GPU is no longer referencing resources that are about to be
    // cleaned up by the destructor.
    WaitForGpu();
    m_superResolution.reset();
    for (uint8_t n = 0; n < FrameCount; ++n)
    {
        if (m_fenceEvents[n]) (void)hipEventDestroy(static_cast<hipEvent_t>(m_fenceEvents[n]));
        m_fenceEvents[n] = nullptr;
    }
    if (m_stream) (void)hipStreamDestroy(static_cast<hipStream_t>(m_stream));
    m_stream = nullptr;
}

// User hot-key interactions.
void SuperResolutionX::OnKeyUp(uint8_t key)
{
    switch (key)
    {
    case VK_SPACE:
        m_pausing = !m_pausing;
        break;
    case VK_F1:
        m_showFPS = !m_showFPS;
        break;
    case VK_F11:
        m_screenShot = 1;
        break;
    case 'U':
        m_updateImage = !m_updateImage;
        break;
    }
}

void SuperResolutionX::ParseCommandLineArgs(wchar_t* argv[], int argc)
{
    const auto lower = [](wstring s)
    {
        transform(s.begin(), s.end(), s.begin(), [](wchar_t c) { return towlower(c); });
        return s;
    };
    // a switch starts with '-' or '/', and matches case-insensitively
    const auto isSwitch = [&](int i, const wchar_t* name)
    {
        const wchar_t* arg = argv[i];
        return (arg[0] == L'-' || arg[0] == L'/') && lower(arg + 1) == lower(name);
    };
    // a value is anything that is not itself a switch; "-1" and "-.5" count as values
    // (SuperResolutionX.cpp:304-310).  There a leading '/' always marks a switch; here it also starts
    // an absolute path, so "/x" is a value when it goes on like a path (another '/' or a '.' in it).
    const auto hasValue = [&](int i)
    {
        if (i + 1 >= argc) return false;
        const wchar_t* arg = argv[i + 1];
        if (arg[0] == L'/') return wcschr(arg + 1, L'/') != nullptr || wcschr(arg + 1, L'.') != nullptr;
        return arg[0] != L'-' || (arg[1] >= L'0' && arg[1] <= L'9') || arg[1] == L'.';
    };

    DXFramework::ParseCommandLineArgs(argv, argc);

    for (int i = 1; i < argc; ++i)
    {
        if (isSwitch(i, L"warp")) m_deviceType = DEVICE_WARP;
        else if (isSwitch(i, L"uma")) m_deviceType = DEVICE_UMA;      // accepted; one kind of device here
        else if (isSwitch(i, L"fp32")) m_isFP16Supported = false;     // the reference takes this from a device feature query
        else if (isSwitch(i, L"nhwc")) m_vendorId = 0x10DE;            // the vendor for which the reference picks NHWC
        else if ((isSwitch(i, L"i") || isSwitch(i, L"image")) && hasValue(i))
        {
            const wstring value = argv[++i];
            m_fileName.assign(value.size(), '\0');
            for (size_t j = 0; j < value.size(); ++j) m_fileName[j] = static_cast<char>(value[j]);
        }
    }
}

void SuperResolutionX::PopulateCommandList()
{
    if (m_framesPerSlot)
    {
        // Stream mode: the slot of this frame index is free (MoveToNextFrame waited for it).  A video source
        // would write the slot's next frames into GetStreamInput(m_frameIndex) here; then the whole slot is
        // recorded and submitted: upload, ImageToTensors + Process, copy into the slot's render targets.
        m_superResolution->SubmitFrames(m_frameIndex);
        m_framePending[m_frameIndex] = true;
        m_isFirstFrame = false;
        return;
    }

    if (m_isFirstFrame || m_updateImage)
    {
        m_superResolution->ImageToTensors(m_stream);
        m_superResolution->Process(m_stream);
        m_isFirstFrame = false;
    }

    RenderTarget* pRenderTarget = &m_renderTargets[m_frameIndex];
    m_superResolution->Render(m_stream, pRenderTarget);
    // the frame's fence: signalled when its render target holds the picture
    ThrowIfFailed(hipEventRecord(static_cast<hipEvent_t>(m_fenceEvents[m_frameIndex]), static_cast<hipStream_t>(m_stream)) == hipSuccess,
                  "hipEventRecord failed");
    m_framePending[m_frameIndex] = true;

    // Screen-shot helper
    if (m_screenShot == 1)
    {
        WaitForGpu();
        m_readBuffer = *pRenderTarget;
        m_screenShot = 2;
    }
}

// Wait for pending GPU work to complete.
void SuperResolutionX::WaitForGpu()
{
    // every frame in flight, oldest first
    for (uint32_t n = 1; n <= FrameCount; ++n) WaitForFrame((m_frameIndex + n) % FrameCount);
    if (m_stream)
        ThrowIfFailed(hipStreamSynchronize(static_cast<hipStream_t>(m_stream)) == hipSuccess, "hipStreamSynchronize failed");
}

// The fence wait of one frame in flight: returns once its render target holds the picture.
void SuperResolutionX::WaitForFrame(uint32_t index)
{
    if (!m_framePending[index]) return;
    if (m_framesPerSlot)
    {
        size_t pitch = 0, stride = 0;
        m_lastOutput = m_superResolution->WaitFrames(index, &pitch, &stride);
        m_lastSlotTime = chrono::steady_clock::now();
        if (m_slotsCompleted++ == 0) m_firstSlotTime = m_lastSlotTime;
        m_framesCompleted += m_framesPerSlot;
    }
    else
    {
        ThrowIfFailed(hipEventSynchronize(static_cast<hipEvent_t>(m_fenceEvents[index])) == hipSuccess, "hipEventSynchronize failed");
        ++m_framesCompleted;
    }
    m_framePending[index] = false;
}

double SuperResolutionX::GetSteadyFramesPerSecond() const
{
    if (m_slotsCompleted < 2) return 0.0;
    const double dt = chrono::duration<double>(m_lastSlotTime - m_firstSlotTime).count();
    return dt > 0.0 ? double(m_slotsCompleted - 1) * m_framesPerSlot / dt : 0.0;
}

// Prepare to render the next frame.
void SuperResolutionX::MoveToNextFrame()
{
    // Update the frame index; if the next frame is not ready to be rendered yet (its render target, or its
    // slot, still has work in flight), wait until it is.  The other frames stay in flight.
    m_frameIndex = (m_frameIndex + 1) % FrameCount;
    WaitForFrame(m_frameIndex);

    // Screen-shot helper: the reference saves FrameCount frames after the request
    if (m_screenShot)
    {
        if (m_screenShot > FrameCount)
        {
            char timeStr[15];
            tm dateTime;
            const auto now = time(nullptr);
            if (localtime_r(&now, &dateTime) && strftime(timeStr, sizeof(timeStr), "%Y%m%d%H%M%S", &dateTime))
            {
                const string name = string("SuperResolutionX_") + timeStr + ".png";
                // stream mode has no read-back copy: the shot is the first frame of the slot that arrived last
                if (m_framesPerSlot) SaveStreamFrame(name.c_str());
                else SaveImage(name.c_str(), &m_readBuffer, m_width, m_height, m_readBuffer.rowPitch);
            }
            m_screenShot = 0;
        }
        else ++m_screenShot;
    }
}

bool SuperResolutionX::SaveImage(char const* fileName, const RenderTarget* pImageBuffer, uint32_t w, uint32_t h,
                                 uint32_t rowPitch, uint8_t comp)
{
    assert(comp == 3 || comp == 4);
    const uint8_t* pData = pImageBuffer->pixels;
    if (!pData)
    {
        cerr << "SaveImage: no picture has been read back yet" << endl;
        return false;
    }

    // strip the row pitch and (comp == 3) the alpha channel
    vector<uint8_t> imageData(size_t(comp) * w * h);
    for (uint32_t i = 0; i < h; ++i)
        for (uint32_t j = 0; j < w; ++j)
            for (uint8_t k = 0; k < comp; ++k)
                imageData[(size_t(w) * i + j) * comp + k] = pData[size_t(rowPitch) * i + 4 * j + k];

    if (!srx::WritePng(fileName, int(w), int(h), comp, imageData.data(), 0))
    {
        cerr << "SaveImage: " << srx::LastImageError() << endl;
        return false;
    }
    cout << "Saved " << fileName << endl;
    return true;
}

// Stream mode: the first frame of the slot that arrived last (RGB8, tightly packed, in the slot's page-locked buffer).
bool SuperResolutionX::SaveStreamFrame(const char* fileName)
{
    if (!m_lastOutput)
    {
        cerr << "SaveImage: no slot has arrived yet" << endl;
        return false;
    }
    if (!srx::WritePng(fileName, int(m_width), int(m_height), 3, m_lastOutput, 0))
    {
        cerr << "SaveImage: " << srx::LastImageError() << endl;
        return false;
    }
    cout << "Saved " << fileName << endl;
    return true;
}

bool SuperResolutionX::SaveFrame(const char* fileName)
{
    WaitForGpu();
    if (m_framesPerSlot) return SaveStreamFrame(fileName);
    const uint32_t last = (m_frameIndex + FrameCount - 1) % FrameCount;
    return SaveImage(fileName, &m_renderTargets[last], m_width, m_height, m_renderTargets[last].rowPitch);
}

double SuperResolutionX::CalculateFrameStats(float* pTimeStep)
{
    const auto now = chrono::steady_clock::now();
    const double totalTime = chrono::duration<double>(now - m_startTime).count();
    ++m_frameCnt;

    const double timeStep = chrono::duration<double>(now - m_statsTime).count();

    // Compute averages over one second period.
    if (timeStep >= 1.0)
    {
        m_lastFps = m_frameCnt / timeStep;
        m_frameCnt = 0;
        m_statsTime = now;

        wstringstream windowText;
        windowText << L"    fps: ";
        if (m_showFPS) windowText << setprecision(2) << fixed << m_lastFps;
        else windowText << L"[F1]";
        windowText << L"    [F11] screen shot";
        SetCustomWindowText(windowText.str().c_str());
    }

    if (pTimeStep) *pTimeStep = static_cast<float>(timeStep);

    return totalTime;
}
