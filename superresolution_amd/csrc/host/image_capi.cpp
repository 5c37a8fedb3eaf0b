// extern "C" view of ImageIO for the Python tests (no GPU involved).
#include <cstring>

#include "ImageIO.h"

extern "C" {

// Each function returns 0 on success and 1 on failure; no C++ exception leaves them.
int srx_image_info(const char* path, int* w, int* h, int* channels, int* req) try
{
    if (!path || !w || !h || !channels || !req) return 1;
    return srx::LoadImageInfoFromFile(path, *w, *h, *channels, *req) ? 0 : 1;
}
catch (...) { return 1; }

// out must hold w * h * req bytes (query with srx_image_info first)
int srx_image_load(const char* path, unsigned char* out, size_t cap) try
{
    if (!path || !out) return 1;
    int w = 0, h = 0, req = 0;
    const std::vector<unsigned char> px = srx::LoadImageFromFile(path, w, h, req);
    if (px.empty() || px.size() > cap) return 1;
    std::memcpy(out, px.data(), px.size());
    return 0;
}
catch (...) { return 1; }

int srx_image_write_png(const char* path, int w, int h, int comp, const void* data, int stride) try
{
    if (!path || !data) return 1;
    return srx::WritePng(path, w, h, comp, data, stride) ? 0 : 1;
}
catch (...) { return 1; }

const char* srx_image_last_error() { return srx::LastImageError().c_str(); }

}
