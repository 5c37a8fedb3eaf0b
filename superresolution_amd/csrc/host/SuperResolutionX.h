// Adapted from StarsX/SuperResolution (SuperResolution/SuperResolutionX.h), whose file header reads
// "Copyright (c) Microsoft. All rights reserved.  This code is licensed under the MIT License (MIT).":
// the public interface and the member names are the reference's.
// The application class of the sample, headless.  Public interface as in the
// reference (SuperResolutionX.h:26-39): constructor (width, height, name) and the
// DXFramework overrides OnInit / OnUpdate / OnRender / OnDestroy / OnKeyUp /
// ParseCommandLineArgs.  The D3D12 device, swap chain, command allocators and
// fence of the reference become: a HIP device index, FrameCount page-locked host
// render targets, an in-order HIP stream, and one event per frame in flight that
// is waited for where the reference waits on that frame's fence value.
#pragma once

#include <chrono>
#include <memory>
#include <string>

#include "DXFramework.h"
#include "SuperResolution.h"

class SuperResolutionX : public DXFramework
{
public:
    SuperResolutionX(uint32_t width, uint32_t height, std::wstring name);
    virtual ~SuperResolutionX();

    virtual void OnInit();
    virtual void OnUpdate();
    virtual void OnRender();
    virtual void OnDestroy();

    virtual void OnKeyUp(uint8_t /*key*/);

    virtual void ParseCommandLineArgs(wchar_t* argv[], int argc);

    // Headless additions: which GPU to use, and a direct save of the last rendered frame.
    void SetDeviceIndex(int device) { m_deviceIndex = device; }
    bool SaveFrame(const char* fileName);
    double GetLastFps() const { return m_lastFps; }

    // Stream mode (before OnInit): every OnRender submits one slot of `framesPerSlot` video frames (upload,
    // ImageToTensors + Process, copy back) and returns; FrameCount slots are in flight and MoveToNextFrame waits
    // only for the slot it is about to reuse, as the reference does with its fence values
    // (SuperResolutionX.cpp:388-405).  The input file stands in for the video source.
    void SetStreamMode(uint32_t framesPerSlot) { m_framesPerSlot = framesPerSlot; }
    uint32_t GetFramesPerSlot() const { return m_framesPerSlot; }
    // Frames whose output has arrived in host memory, and when the first and the latest slot arrived.
    uint64_t GetFramesCompleted() const { return m_framesCompleted; }
    double GetSteadyFramesPerSecond() const;
    // Wait for every frame in flight (OnDestroy does so too).
    void Drain() { WaitForGpu(); }

private:
    enum DeviceType : uint8_t
    {
        DEVICE_DISCRETE,
        DEVICE_UMA,
        DEVICE_WARP
    };

    static const uint8_t FrameCount = 3;

    RenderTarget m_renderTargets[FrameCount];

    std::unique_ptr<SuperResolution> m_superResolution;

    uint32_t m_vendorId;
    bool m_isFP16Supported;
    int m_deviceIndex;
    void* m_stream;

    // Synchronization objects.
    uint32_t m_frameIndex;
    void* m_fenceEvents[FrameCount];      // hipEvent_t, recorded after a frame's last command
    bool m_framePending[FrameCount];      // the frame (or slot) of this index has work in flight

    // Stream mode
    uint32_t m_framesPerSlot;             // 0 = one image per frame, as the reference
    uint64_t m_framesCompleted;
    uint32_t m_slotsCompleted;
    std::chrono::steady_clock::time_point m_firstSlotTime, m_lastSlotTime;
    const uint8_t* m_lastOutput;          // first frame of the slot that arrived last (RGB8, tightly packed)

    // Application state
    DeviceType m_deviceType;
    bool m_updateImage;
    bool m_showFPS;
    bool m_pausing;
    bool m_isFirstFrame;

    // User external settings
    std::string m_fileName;

    // Screen-shot helpers and state
    RenderTarget m_readBuffer;
    uint8_t m_screenShot;

    // frame statistics
    std::chrono::steady_clock::time_point m_startTime, m_statsTime;
    uint32_t m_frameCnt;
    double m_lastFps;

    void LoadPipeline();
    void LoadAssets();
    bool InitTensors();

    void PopulateCommandList();
    void WaitForGpu();
    void WaitForFrame(uint32_t index);
    void MoveToNextFrame();
    bool SaveStreamFrame(const char* fileName);
    bool SaveImage(char const* fileName, const RenderTarget* pImageBuffer, uint32_t w, uint32_t h,
                   uint32_t rowPitch, uint8_t comp = 3);
    double CalculateFrameStats(float* fTimeStep = nullptr);
};
