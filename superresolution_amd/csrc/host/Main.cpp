// Headless entry point: drives SuperResolutionX through the same sequence the
// reference's window loop does (Main.cpp:15-20, Common/Win32Application.cpp:17-75):
// ParseCommandLineArgs -> OnInit -> {OnUpdate, OnRender}* -> OnDestroy.
//
//   srx_main [-i|-image <png>] [-uma] [-warp]        the reference's switches
//            [-frames N] [-update] [-o out.png] [-device D] [-noshot]
// By default three frames are rendered with F11 "pressed" before the first one, so
// a SuperResolutionX_<timestamp>.png appears like the reference's screen shot.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "SuperResolutionX.h"

int main(int argc, char* argv[])
{
    int frames = 5, device = 0;
    bool update = false, shot = true;
    std::string outPath;
    std::vector<std::wstring> wide;
    for (int i = 0; i < argc; ++i)
    {
        const char* a = argv[i];
        const bool sw = i > 0 && (a[0] == '-' || a[0] == '/');
        if (sw && !strcasecmp(a + 1, "frames") && i + 1 < argc) { frames = std::atoi(argv[++i]); continue; }
        if (sw && !strcasecmp(a + 1, "device") && i + 1 < argc) { device = std::atoi(argv[++i]); continue; }
        if (sw && !strcasecmp(a + 1, "o") && i + 1 < argc) { outPath = argv[++i]; continue; }
        if (sw && !strcasecmp(a + 1, "update")) { update = true; continue; }
        if (sw && !strcasecmp(a + 1, "noshot")) { shot = false; continue; }
        wide.emplace_back(a, a + std::strlen(a));
    }
    std::vector<wchar_t*> wargv;
    for (auto& w : wide) wargv.push_back(&w[0]);

    try
    {
        SuperResolutionX superResolutionX(1280, 720, L"Super resolution");
        superResolutionX.ParseCommandLineArgs(wargv.data(), int(wargv.size()));
        superResolutionX.SetDeviceIndex(device);
        superResolutionX.OnInit();
        std::printf("output %ux%u\n", superResolutionX.GetWidth(), superResolutionX.GetHeight());
        if (update) superResolutionX.OnKeyUp('U');
        if (shot && outPath.empty()) superResolutionX.OnKeyUp(VK_F11);
        for (int f = 0; f < frames; ++f)
        {
            superResolutionX.OnUpdate();
            superResolutionX.OnRender();
        }
        if (!outPath.empty()) superResolutionX.SaveFrame(outPath.c_str());
        superResolutionX.OnDestroy();
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
