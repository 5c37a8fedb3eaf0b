// Headless entry point: drives SuperResolutionX through the same sequence the
// reference's window loop does (Main.cpp:15-20, Common/Win32Application.cpp:17-75):
// ParseCommandLineArgs -> OnInit -> {OnUpdate, OnRender}* -> OnDestroy.
//
//   srx_main [-i|-image <png>] [-uma] [-warp]        the reference's switches
//            [-frames N] [-update] [-o out.png] [-device D] [-devices D0,D1,...] [-noshot]
//            [-stream N [-slot F] [-shot]]
// By default five frames are rendered with F11 "pressed" before the first one, so
// a SuperResolutionX_<timestamp>.png appears like the reference's screen shot.
//
// -stream N runs the throughput configuration: N video frames (the input file, repeated) go from host
// memory through the GPU and back in slots of F frames (default 8), three slots in flight (the
// reference's FrameCount), and the program prints frames per second, host to host.  With -devices
// every listed device runs its share of the N frames that way.  -shot "presses" F11 before the first slot:
// the screen shot is then the first frame of the slot that arrived last when the shot is due.
//
// -devices lists the GPUs of a multi-GPU run (the reference picks ONE adapter,
// SuperResolutionX.cpp:76-117; frames are independent, so here every listed device gets
// its own SuperResolutionX on its own host thread and the frames {k, k + G, ...} of the
// -frames N sequence, inference re-run for every frame, nothing exchanged between
// devices).  A device may be listed more than once: its instances then share it.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "SuperResolutionX.h"

namespace {

struct Options
{
    int frames = 5;
    int stream = 0, slot = 8;             // -stream N: frames of the throughput run; -slot F: frames per slot
    bool update = false, shot = true;
    bool streamShot = false;              // -shot: F11 in stream mode too
    std::string outPath;
    std::vector<std::wstring> wide;       // what the class's own ParseCommandLineArgs sees
};

// One instance on one device: the window loop's sequence for `frames` frames.
void runInstance(const Options& opt, int device, int frames, bool update, bool shot, const std::string& outPath, bool quiet)
{
    std::vector<std::wstring> wide = opt.wide;
    std::vector<wchar_t*> wargv;
    for (auto& w : wide) wargv.push_back(&w[0]);

    SuperResolutionX superResolutionX(1280, 720, L"Super resolution");
    superResolutionX.ParseCommandLineArgs(wargv.data(), int(wargv.size()));
    superResolutionX.SetDeviceIndex(device);
    if (opt.stream > 0)
    {
        // frames = this instance's share of the stream; every OnRender submits one slot of them
        superResolutionX.SetStreamMode(uint32_t(opt.slot));
        superResolutionX.OnInit();
        if (!quiet) std::printf("output %ux%u\n", superResolutionX.GetWidth(), superResolutionX.GetHeight());
        const int slots = (frames + opt.slot - 1) / opt.slot;
        if (opt.streamShot && shot) superResolutionX.OnKeyUp(VK_F11);
        const auto s0 = std::chrono::steady_clock::now();
        for (int f = 0; f < slots; ++f)
        {
            superResolutionX.OnUpdate();
            superResolutionX.OnRender();
        }
        superResolutionX.Drain();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count();
        std::printf("stream on device %d: %llu frames in %.4f s = %.1f frames/s host to host (%d frames per slot, 3 slots in flight); "
                    "%.1f frames/s between the first and the last slot's arrival\n",
                    device, static_cast<unsigned long long>(superResolutionX.GetFramesCompleted()), dt,
                    double(superResolutionX.GetFramesCompleted()) / dt, opt.slot, superResolutionX.GetSteadyFramesPerSecond());
        if (!outPath.empty()) superResolutionX.SaveFrame(outPath.c_str());
        superResolutionX.OnDestroy();
        return;
    }
    superResolutionX.OnInit();
    if (!quiet) std::printf("output %ux%u\n", superResolutionX.GetWidth(), superResolutionX.GetHeight());
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

// "dir/name.png" -> "dir/name.<k>.png"; the extension is looked for in the file name only.
std::string indexedPath(const std::string& path, size_t k)
{
    const size_t slash = path.find_last_of('/');
    const size_t nameStart = slash == std::string::npos ? 0 : slash + 1;
    const size_t dot = path.find_last_of('.');
    const std::string tag = "." + std::to_string(k);
    return (dot == std::string::npos || dot <= nameStart) ? path + tag : path.substr(0, dot) + tag + path.substr(dot);
}

}  // namespace

int main(int argc, char* argv[])
{
    Options opt;
    int device = 0;
    std::vector<int> devices;
    for (int i = 0; i < argc; ++i)
    {
        const char* a = argv[i];
        const bool sw = i > 0 && (a[0] == '-' || a[0] == '/');
        if (sw && !strcasecmp(a + 1, "frames") && i + 1 < argc) { opt.frames = std::atoi(argv[++i]); continue; }
        if (sw && !strcasecmp(a + 1, "stream") && i + 1 < argc) { opt.stream = std::atoi(argv[++i]); continue; }
        if (sw && !strcasecmp(a + 1, "slot") && i + 1 < argc) { opt.slot = std::atoi(argv[++i]); continue; }
        if (sw && !strcasecmp(a + 1, "device") && i + 1 < argc) { device = std::atoi(argv[++i]); continue; }
        if (sw && !strcasecmp(a + 1, "devices") && i + 1 < argc)
        {
            for (const char* p = argv[++i]; *p;)
            {
                char* end = nullptr;
                const long d = std::strtol(p, &end, 10);
                if (end == p || d < 0) { std::fprintf(stderr, "error: -devices expects a comma-separated list of device indices\n"); return 1; }
                devices.push_back(int(d));
                p = *end == ',' ? end + 1 : end;
            }
            continue;
        }
        if (sw && !strcasecmp(a + 1, "o") && i + 1 < argc) { opt.outPath = argv[++i]; continue; }
        if (sw && !strcasecmp(a + 1, "update")) { opt.update = true; continue; }
        if (sw && !strcasecmp(a + 1, "noshot")) { opt.shot = false; continue; }
        if (sw && !strcasecmp(a + 1, "shot")) { opt.streamShot = true; continue; }
        opt.wide.emplace_back(a, a + std::strlen(a));
    }

    if (opt.stream < 0 || opt.slot < 1 || opt.slot > 64)
    {
        std::fprintf(stderr, "error: -stream expects a frame count, -slot 1..64 frames per slot\n");
        return 1;
    }
    const int total = opt.stream > 0 ? opt.stream : opt.frames;
    if (devices.empty())
    {
        try
        {
            runInstance(opt, device, total, opt.update, opt.shot, opt.outPath, false);
        }
        catch (const std::exception& e)
        {
            std::fprintf(stderr, "error: %s\n", e.what());
            return 1;
        }
        return 0;
    }

    // Frame-sharded run: instance k takes frames k, k + G, ...
    const size_t G = devices.size();
    std::vector<std::string> errors(G);
    std::vector<double> seconds(G, 0.0);
    std::vector<int> share(G, 0);
    std::vector<std::thread> threads;
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t k = 0; k < G; ++k)
    {
        share[k] = (total > int(k)) ? (total - int(k) + int(G) - 1) / int(G) : 0;
        if (share[k] == 0) continue;          // more instances than frames: this one has nothing to render or save
        threads.emplace_back([&, k] {
            try
            {
                const auto s0 = std::chrono::steady_clock::now();
                const std::string out = opt.outPath.empty() ? std::string() : indexedPath(opt.outPath, k);
                runInstance(opt, devices[k], share[k], true, false, out, k != 0);
                seconds[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count();
            }
            catch (const std::exception& e)
            {
                errors[k] = e.what();
            }
        });
    }
    for (auto& t : threads) t.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int rc = 0, done = 0;
    for (size_t k = 0; k < G; ++k)
    {
        if (!errors[k].empty())
        {
            std::fprintf(stderr, "error: instance %zu (device %d): %s\n", k, devices[k], errors[k].c_str());
            rc = 1;
            continue;
        }
        if (share[k] == 0)
        {
            std::printf("instance %zu on device %d: no frames\n", k, devices[k]);
            continue;
        }
        done += share[k];
        std::printf("instance %zu on device %d: %d frames in %.3f s (start-up included)\n", k, devices[k], share[k], seconds[k]);
    }
    std::printf("%d frames on %zu instances in %.3f s\n", done, G, wall);
    return rc;
}
