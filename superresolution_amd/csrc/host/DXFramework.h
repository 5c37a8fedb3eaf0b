// Adapted from StarsX/SuperResolution (SuperResolution/Common/DXFramework.h, whose file header reads "Copyright (c)
// Microsoft. All rights reserved.  This code is licensed under the MIT License (MIT)."): the framework base class, without the Win32 window.
// Headless counterpart of the sample framework base class the reference's
// SuperResolutionX derives from (Common/DXFramework.h:17-66): same virtual
// interface and accessors, no window.  The "window title" is kept as a string
// and echoed to stdout when it changes.
#pragma once

#include <cstdint>
#include <string>

class DXFramework
{
public:
    DXFramework(uint32_t width, uint32_t height, std::wstring name);
    virtual ~DXFramework();

    virtual void OnInit() = 0;
    virtual void OnUpdate() = 0;
    virtual void OnRender() = 0;
    virtual void OnDestroy() = 0;

    // Samples override the event handlers to handle specific messages.
    virtual void OnKeyDown(uint8_t /*key*/) {}
    virtual void OnKeyUp(uint8_t /*key*/) {}

    // Accessors.
    uint32_t GetWidth() const { return m_width; }
    uint32_t GetHeight() const { return m_height; }
    const wchar_t* GetTitle() const { return m_title.c_str(); }

    virtual void ParseCommandLineArgs(wchar_t* argv[], int argc);

protected:
    void SetCustomWindowText(const wchar_t* text);

    // Viewport dimensions.
    uint32_t m_width;
    uint32_t m_height;
    float m_aspectRatio;

    // Window title.
    std::wstring m_title;
};

// Virtual-key codes the sample reacts to (SuperResolutionX.cpp:268-285).
enum : uint8_t { VK_SPACE = 0x20, VK_F1 = 0x70, VK_F11 = 0x7A };
