// Adapted from StarsX/SuperResolution (SuperResolution/Common/DXFramework.cpp, whose file header reads "Copyright (c)
// Microsoft. All rights reserved.  This code is licensed under the MIT License (MIT).").
#include "DXFramework.h"

#include <cstdio>

DXFramework::DXFramework(uint32_t width, uint32_t height, std::wstring name)
    : m_width(width), m_height(height), m_title(std::move(name))
{
    m_aspectRatio = static_cast<float>(width) / static_cast<float>(height);
}

DXFramework::~DXFramework() {}

void DXFramework::ParseCommandLineArgs(wchar_t* /*argv*/[], int /*argc*/) {}

// Helper function for setting the window's title text: headless, so the text goes to stdout.
void DXFramework::SetCustomWindowText(const wchar_t* text)
{
    std::printf("%ls:%ls\n", m_title.c_str(), text);
    std::fflush(stdout);
}
