// BMP, TGA, binary PNM (ImageFormats.cpp) JPEG (ImageJpeg.cpp) and GIF (ImageGif.cpp) decoding; used by ImageIO
// for files that are not PNG.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace srx {
namespace formats {

// Size and channel count (1, 3 or 4) of a BMP / PNM / TGA file image held in memory.
bool Info(const std::vector<uint8_t>& file, int& width, int& height, int& channels, std::string& err);
// The decoded pixels, `channels` interleaved bytes per pixel, rows top to bottom.
bool Decode(const std::vector<uint8_t>& file, int& width, int& height, int& channels, std::vector<uint8_t>& pixels,
            std::string& err);

// Baseline and progressive JPEG: 1 (grey) or 3 (RGB) channels.
bool IsJpeg(const std::vector<uint8_t>& file);
bool JpegInfo(const std::vector<uint8_t>& file, int& width, int& height, int& channels, std::string& err);
bool JpegDecode(const std::vector<uint8_t>& file, int& width, int& height, int& channels, std::vector<uint8_t>& pixels,
                std::string& err);

// GIF: the first image composed on the logical screen, 4 channels.
bool IsGif(const std::vector<uint8_t>& file);
bool GifInfo(const std::vector<uint8_t>& file, int& width, int& height, int& channels, std::string& err);
bool GifDecode(const std::vector<uint8_t>& file, int& width, int& height, int& channels, std::vector<uint8_t>& pixels,
               std::string& err);

}  // namespace formats
}  // namespace srx
