// BMP, TGA and binary PNM decoding for the host program's image loader.
//
// The reference loads images through stb_image (XUSG/Advanced/XUSGTextureLoader.h:17-32 ->
// stbi_info + stbi_load), which accepts these formats beside PNG.  The contract kept here is the
// decoded result of the common, uncompressed (and TGA run-length) flavours: 8 bits per channel,
// rows top to bottom, channel count as stb reports it (BMP: 3, or 4 for 32-bit files; TGA: 1,
// 2 (grey + alpha), 3 or 4, also through a colour map; PNM: 1 or 3).  What is not handled (BMP bit
// fields / RLE / 16-bit, 15/16-bit colour TGA, 16-bit PNM) is refused with a message rather than guessed at.
#include "ImageFormats.h"

#include <cstdint>
#include <cstring>

namespace srx {
namespace formats {

namespace {

uint32_t le16(const uint8_t* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8); }
uint32_t le32(const uint8_t* p) { return le16(p) | (le16(p + 2) << 16); }

bool setError(std::string& err, const char* why)
{
    err = why;
    return false;
}

// ---------------------------------------------------------------- BMP
struct BmpHeader
{
    uint32_t dataOffset = 0, headerSize = 0, bpp = 0, compression = 0, colorsUsed = 0;
    int32_t width = 0, height = 0;
    bool bottomUp = true;
};

bool parseBmp(const std::vector<uint8_t>& f, BmpHeader& h, std::string& err)
{
    if (f.size() < 14 + 12) return setError(err, "BMP: file too short");
    h.dataOffset = le32(&f[10]);
    h.headerSize = le32(&f[14]);
    if (h.headerSize == 12)
    {
        h.width = int32_t(le16(&f[18]));
        h.height = int32_t(le16(&f[20]));
        h.bpp = le16(&f[24]);
    }
    else
    {
        if (h.headerSize < 40 || f.size() < 14 + 40) return setError(err, "BMP: unsupported header");
        h.width = int32_t(le32(&f[18]));
        h.height = int32_t(le32(&f[22]));
        h.bpp = le16(&f[28]);
        h.compression = le32(&f[30]);
        h.colorsUsed = le32(&f[46]);
    }
    h.bottomUp = h.height > 0;
    if (h.height == INT32_MIN) return setError(err, "BMP: bad size");
    if (h.height < 0) h.height = -h.height;
    if (h.width <= 0 || h.height <= 0 || h.width > (1 << 24) || h.height > (1 << 24)) return setError(err, "BMP: bad size");
    if (h.compression != 0) return setError(err, "BMP: compressed and bit-field files are not supported");
    if (h.bpp != 1 && h.bpp != 4 && h.bpp != 8 && h.bpp != 24 && h.bpp != 32)
        return setError(err, "BMP: 16-bit files are not supported");
    return true;
}

bool decodeBmp(const std::vector<uint8_t>& f, const BmpHeader& h, std::vector<uint8_t>& out, int channels, std::string& err)
{
    const size_t W = size_t(h.width), H = size_t(h.height);
    const size_t rowBytes = ((W * h.bpp + 7) / 8 + 3) & ~size_t(3);
    if (h.dataOffset > f.size() || rowBytes * H > f.size() - h.dataOffset) return setError(err, "BMP: pixel data is short");
    const uint8_t* palette = nullptr;
    size_t paletteEntry = 4, paletteCount = 0;
    if (h.bpp <= 8)
    {
        const size_t at = 14 + h.headerSize;
        paletteEntry = h.headerSize == 12 ? 3 : 4;
        size_t n = h.colorsUsed ? h.colorsUsed : 256;
        if (h.headerSize == 12 || !h.colorsUsed) n = (h.dataOffset - at) / paletteEntry;     // as many as lie before the pixels
        if (n == 0 || n > 256 || at + n * paletteEntry > f.size()) return setError(err, "BMP: bad palette");
        palette = &f[at];
        paletteCount = n;
    }
    out.assign(W * H * size_t(channels), 255);
    bool anyAlpha = false;
    for (size_t y = 0; y < H; ++y)
    {
        const uint8_t* src = &f[h.dataOffset + (h.bottomUp ? H - 1 - y : y) * rowBytes];
        uint8_t* dst = &out[y * W * size_t(channels)];
        for (size_t x = 0; x < W; ++x, dst += channels)
        {
            if (h.bpp <= 8)
            {
                // 1- and 4-bit indices are packed most significant first
                size_t idx = src[x];
                if (h.bpp == 4) idx = (src[x >> 1] >> ((x & 1) ? 0 : 4)) & 15;
                else if (h.bpp == 1) idx = (src[x >> 3] >> (7 - (x & 7))) & 1;
                if (idx >= paletteCount) idx = 0;
                const uint8_t* e = palette + idx * paletteEntry;
                dst[0] = e[2]; dst[1] = e[1]; dst[2] = e[0];
            }
            else
            {
                const uint8_t* e = src + x * (h.bpp / 8);
                dst[0] = e[2]; dst[1] = e[1]; dst[2] = e[0];
                if (h.bpp == 32) { dst[3] = e[3]; anyAlpha |= e[3] != 0; }
            }
        }
    }
    // a 32-bit file whose fourth byte is zero everywhere carries no alpha: opaque
    if (h.bpp == 32 && !anyAlpha)
        for (size_t i = 3; i < out.size(); i += 4) out[i] = 255;
    return true;
}

// ---------------------------------------------------------------- binary PNM (P5, P6)
struct PnmHeader
{
    int channels = 0, width = 0, height = 0, maxval = 0;
    size_t dataOffset = 0;
};

bool parsePnm(const std::vector<uint8_t>& f, PnmHeader& h, std::string& err)
{
    if (f.size() < 3 || f[0] != 'P' || (f[1] != '5' && f[1] != '6')) return setError(err, "PNM: not a binary P5 / P6 file");
    h.channels = f[1] == '6' ? 3 : 1;
    size_t p = 2;
    int values[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k)
    {
        for (;;)    // white space and comments
        {
            while (p < f.size() && (f[p] == ' ' || f[p] == '\t' || f[p] == '\n' || f[p] == '\r' || f[p] == '\f' || f[p] == '\v')) ++p;
            if (p < f.size() && f[p] == '#') { while (p < f.size() && f[p] != '\n' && f[p] != '\r') ++p; }
            else break;
        }
        if (p >= f.size() || f[p] < '0' || f[p] > '9') return setError(err, "PNM: bad header");
        long v = 0;
        while (p < f.size() && f[p] >= '0' && f[p] <= '9' && v < (1 << 26)) v = v * 10 + (f[p++] - '0');
        values[k] = int(v);
    }
    ++p;    // the single white-space byte after maxval
    h.width = values[0]; h.height = values[1]; h.maxval = values[2];
    h.dataOffset = p;
    if (h.width <= 0 || h.height <= 0 || h.width > (1 << 24) || h.height > (1 << 24)) return setError(err, "PNM: bad size");
    if (h.maxval <= 0 || h.maxval > 65535) return setError(err, "PNM: max value must be 1..65535");
    // stb_image's 8-bit entry point hands back the LOW byte of big-endian 16-bit samples on a
    // little-endian host: neither copied nor silently corrected here
    if (h.maxval > 255) return setError(err, "PNM: 16-bit samples are not supported");
    return true;
}

bool decodePnm(const std::vector<uint8_t>& f, const PnmHeader& h, std::vector<uint8_t>& out, std::string& err)
{
    const size_t n = size_t(h.width) * h.height * h.channels;
    if (h.dataOffset > f.size() || n > f.size() - h.dataOffset) return setError(err, "PNM: pixel data is short");
    out.assign(f.begin() + long(h.dataOffset), f.begin() + long(h.dataOffset + n));
    return true;
}

// ---------------------------------------------------------------- TGA
struct TgaHeader
{
    int idLength = 0, colorMapType = 0, imageType = 0, width = 0, height = 0, bpp = 0, descriptor = 0;
    int paletteStart = 0, paletteLength = 0, paletteBits = 0;
    int channels = 0;      // of the decoded image: the palette's for a colour-mapped file
    bool rle = false, indexed = false;
};

bool parseTga(const std::vector<uint8_t>& f, TgaHeader& h, std::string& err)
{
    if (f.size() < 18) return setError(err, "unknown image type");
    h.idLength = f[0];
    h.colorMapType = f[1];
    h.imageType = f[2];
    h.paletteStart = int(le16(&f[3]));
    h.paletteLength = int(le16(&f[5]));
    h.paletteBits = f[7];
    h.width = int(le16(&f[12]));
    h.height = int(le16(&f[14]));
    h.bpp = f[16];
    h.descriptor = f[17];
    h.rle = h.imageType >= 8;
    h.indexed = h.colorMapType == 1;
    const int base = h.imageType & 7;
    if (h.colorMapType > 1 || (base != 1 && base != 2 && base != 3) || (h.imageType & ~11) != 0 || h.width <= 0 || h.height <= 0 ||
        (h.indexed != (base == 1)))
        return setError(err, "unknown image type");      // not a TGA either
    if (h.indexed)
    {
        if (h.bpp != 8 && h.bpp != 16) return setError(err, "TGA: colour-map indices must be 8 or 16 bits");
        if (h.paletteBits == 24 || h.paletteBits == 32) h.channels = h.paletteBits / 8;
        else return setError(err, "TGA: only 24- and 32-bit colour maps are supported");
        if (h.paletteLength == 0) return setError(err, "TGA: empty colour map");
    }
    else if (base == 3 && h.bpp == 8) h.channels = 1;
    else if (base == 3 && h.bpp == 16) h.channels = 2;      // grey + alpha
    else if (base == 2 && h.bpp == 24) h.channels = 3;
    else if (base == 2 && h.bpp == 32) h.channels = 4;
    else return setError(err, "TGA: 15- and 16-bit colour files are not supported");
    return true;
}

bool decodeTga(const std::vector<uint8_t>& f, const TgaHeader& h, std::vector<uint8_t>& out, std::string& err)
{
    const size_t W = size_t(h.width), H = size_t(h.height), C = size_t(h.channels), n = W * H;
    const size_t S = h.indexed ? size_t(h.bpp / 8) : C;      // bytes per stored pixel
    size_t p = 18 + size_t(h.idLength);
    const uint8_t* palette = nullptr;
    if (h.indexed)
    {
        // the reference's decoder skips "first entry index" BYTES before the map (stb_image.h:5949); so does this
        p += size_t(h.paletteStart);
        const size_t bytes = size_t(h.paletteLength) * C;
        if (p > f.size() || bytes > f.size() - p) return setError(err, "TGA: colour map is short");
        palette = &f[p];
        p += bytes;
    }
    std::vector<uint8_t> raw(n * S);
    if (!h.rle)
    {
        if (p > f.size() || n * S > f.size() - p) return setError(err, "TGA: pixel data is short");
        std::memcpy(raw.data(), &f[p], n * S);
    }
    else
    {
        size_t i = 0;
        while (i < n)
        {
            if (p >= f.size()) return setError(err, "TGA: pixel data is short");
            const int packet = f[p++];
            size_t count = size_t(packet & 127) + 1;
            if (count > n - i) count = n - i;
            if (packet & 128)
            {
                if (S > f.size() - p) return setError(err, "TGA: pixel data is short");
                for (size_t k = 0; k < count; ++k) std::memcpy(&raw[(i + k) * S], &f[p], S);
                p += S;
            }
            else
            {
                if (count * S > f.size() - p) return setError(err, "TGA: pixel data is short");
                std::memcpy(&raw[i * S], &f[p], count * S);
                p += count * S;
            }
            i += count;
        }
    }
    // rows are stored bottom-up unless bit 5 of the descriptor is set; colours are B, G, R(, A)
    const bool bottomUp = ((h.descriptor >> 5) & 1) == 0;
    out.resize(n * C);
    for (size_t y = 0; y < H; ++y)
    {
        const uint8_t* src = &raw[(bottomUp ? H - 1 - y : y) * W * S];
        uint8_t* dst = &out[y * W * C];
        for (size_t x = 0; x < W; ++x, src += S, dst += C)
        {
            const uint8_t* e = src;
            if (h.indexed)
            {
                size_t idx = S == 1 ? src[0] : le16(src);
                if (idx >= size_t(h.paletteLength)) idx = 0;      // an index beyond the map reads entry 0
                e = palette + idx * C;
            }
            if (C <= 2) { dst[0] = e[0]; if (C == 2) dst[1] = e[1]; }
            else
            {
                dst[0] = e[2]; dst[1] = e[1]; dst[2] = e[0];
                if (C == 4) dst[3] = e[3];
            }
        }
    }
    return true;
}

bool isBmp(const std::vector<uint8_t>& f) { return f.size() >= 2 && f[0] == 'B' && f[1] == 'M'; }
bool isPnm(const std::vector<uint8_t>& f) { return f.size() >= 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6'); }

}  // namespace

bool Info(const std::vector<uint8_t>& file, int& width, int& height, int& channels, std::string& err)
{
    if (IsJpeg(file)) return JpegInfo(file, width, height, channels, err);
    if (IsGif(file)) return GifInfo(file, width, height, channels, err);
    if (isBmp(file))
    {
        BmpHeader h;
        if (!parseBmp(file, h, err)) return false;
        width = h.width; height = h.height; channels = h.bpp == 32 ? 4 : 3;
        return true;
    }
    if (isPnm(file))
    {
        PnmHeader h;
        if (!parsePnm(file, h, err)) return false;
        width = h.width; height = h.height; channels = h.channels;
        return true;
    }
    TgaHeader h;
    if (!parseTga(file, h, err)) return false;
    width = h.width; height = h.height; channels = h.channels;
    return true;
}

bool Decode(const std::vector<uint8_t>& file, int& width, int& height, int& channels, std::vector<uint8_t>& pixels, std::string& err)
{
    if (IsJpeg(file)) return JpegDecode(file, width, height, channels, pixels, err);
    if (IsGif(file)) return GifDecode(file, width, height, channels, pixels, err);
    if (isBmp(file))
    {
        BmpHeader h;
        if (!parseBmp(file, h, err)) return false;
        width = h.width; height = h.height; channels = h.bpp == 32 ? 4 : 3;
        return decodeBmp(file, h, pixels, channels, err);
    }
    if (isPnm(file))
    {
        PnmHeader h;
        if (!parsePnm(file, h, err)) return false;
        width = h.width; height = h.height; channels = h.channels;
        return decodePnm(file, h, pixels, err);
    }
    TgaHeader h;
    if (!parseTga(file, h, err)) return false;
    width = h.width; height = h.height; channels = h.channels;
    return decodeTga(file, h, pixels, err);
}

}  // namespace formats
}  // namespace srx
