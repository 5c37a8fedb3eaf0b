// Thin extern "C" shim around reference code compiled from where it lies under
// /root/reference (never copied into this repository).  TEST INFRASTRUCTURE
// ONLY: used to pin oracle/sr_oracle.c, never by the shipped library.
//   Float16Compressor::compress  SuperResolution/Common/Float16Compressor.h:41-57
//   stbi_info / stbi_load        SuperResolution/Common/stb_image.h (v2.30), as
//                                used by SuperResolution/XUSG/Advanced/XUSGTextureLoader.h:17-32
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include "Float16Compressor.h"
#include "stb_image.h"

extern "C" {

uint16_t ref_f16_compress(float v) { return Float16Compressor::compress(v); }
float ref_f16_decompress(uint16_t h) { return Float16Compressor::decompress(h); }

void ref_f16_compress_array(const float* in, uint16_t* out, size_t n)
{
    for (size_t i = 0; i < n; ++i) out[i] = Float16Compressor::compress(in[i]);
}

// Mirrors LoadImageFromFile: 3-channel files are expanded to 4 channels, other
// channel counts load as they are.  Returns a malloc'd buffer (free with ref_free).
unsigned char* ref_load_image(const char* path, int* w, int* h, int* channels, int* req)
{
    if (!stbi_info(path, w, h, channels)) return nullptr;
    *req = *channels != 3 ? *channels : 4;
    int ch = 0;
    return stbi_load(path, w, h, &ch, *req);
}

void ref_free(void* p) { free(p); }

}
