// extern "C" view of ImageIO for the Python tests (no GPU involved).
#include <cstring>

#include "ImageIO.h"

extern "C" {

int srx_image_info(const char* path, int* w, int* h, int* channels, int* req)
{
    return srx::LoadImageInfoFromFile(path, *w, *h, *channels, *req) ? 0 : 1;
}

// out must hold w * h * req bytes (query with srx_image_info first)
int srx_image_load(const char* path, unsigned char* out, size_t cap)
{
    int w = 0, h = 0, req = 0;
    const std::vector<unsigned char> px = srx::LoadImageFromFile(path, w, h, req);
    if (px.empty() || px.size() > cap) return 1;
    std::memcpy(out, px.data(), px.size());
    return 0;
}

int srx_image_write_png(const char* path, int w, int h, int comp, const void* data, int stride)
{
    return srx::WritePng(path, w, h, comp, data, stride) ? 0 : 1;
}

const char* srx_image_last_error() { return srx::LastImageError().c_str(); }

}
