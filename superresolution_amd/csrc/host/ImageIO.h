// Headless image I/O for the super-resolution host program: PNG (and BMP, TGA, binary PNM, GIF, JPEG) in, PNG out.
//
// Takes the place of the stb-based loader / writer the reference calls
// (XUSG/Advanced/XUSGTextureLoader.h:17-32 -> stbi_info + stbi_load;
// SuperResolutionX.cpp:440 -> stbi_write_png).  The contract kept is the decoded
// result: 8 bits per channel, rows top to bottom, and a 3-channel file is
// expanded to 4 channels with alpha 255 while 1-, 2- and 4-channel files load
// with their own channel count.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace srx {

struct ImageInfo
{
    int width = 0, height = 0;
    int channels = 0;      // channels in the file: 1 = Y, 2 = YA, 3 = RGB, 4 = RGBA
};

// stbi_info equivalent: size and channel count without decoding the pixels.
bool LoadImageInfoFromFile(const char* fileName, int& width, int& height, int& channels, int& reqChannels);

// stbi_load equivalent.  reqChannels receives 4 for 3-channel files, otherwise the file's
// channel count; the returned buffer holds width * height * reqChannels bytes.  Empty on failure
// (the reason is available from LastImageError()).
std::vector<uint8_t> LoadImageFromFile(const char* fileName, int& width, int& height, int& reqChannels);

// stbi_write_png equivalent: comp = 1..4 interleaved 8-bit channels, strideInBytes = 0 means tight.
bool WritePng(const char* fileName, int w, int h, int comp, const void* data, int strideInBytes);

const std::string& LastImageError();

}  // namespace srx
