#include "ImageIO.h"

#include "ImageFormats.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace srx {
namespace {

thread_local std::string g_error;

bool fail(const std::string& why)
{
    g_error = why;
    return false;
}

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

bool readFile(const char* name, std::vector<uint8_t>& out)
{
    FILE* f = std::fopen(name, "rb");
    if (!f) return fail(std::string("cannot open ") + name);
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? size_t(n) : 0);
    const size_t got = out.empty() ? 0 : std::fread(out.data(), 1, out.size(), f);
    std::fclose(f);
    return got == out.size() ? true : fail(std::string("short read on ") + name);
}

struct PngHeader
{
    uint32_t width = 0, height = 0;
    int depth = 0, colorType = 0, interlace = 0;
    bool hasTrns = false;
};

const uint8_t kSignature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

// Walks the chunk list; collects IDAT, PLTE and tRNS when asked.
bool parseChunks(const std::vector<uint8_t>& file, PngHeader& hdr, std::vector<uint8_t>* idat,
                 std::vector<uint8_t>* plte, std::vector<uint8_t>* trns)
{
    if (file.size() < 33 || std::memcmp(file.data(), kSignature, 8) != 0) return fail("not a PNG file");
    size_t pos = 8;
    bool sawHeader = false;
    while (pos + 12 <= file.size())
    {
        const uint32_t len = be32(&file[pos]);
        const uint8_t* type = &file[pos + 4];
        if (len > file.size() - pos - 12) return fail("truncated PNG chunk");
        const uint8_t* data = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4))
        {
            if (len != 13) return fail("bad IHDR");
            hdr.width = be32(data);
            hdr.height = be32(data + 4);
            hdr.depth = data[8];
            hdr.colorType = data[9];
            hdr.interlace = data[12];
            if (data[10] != 0 || data[11] != 0) return fail("unsupported PNG compression/filter method");
            sawHeader = true;
        }
        else if (!std::memcmp(type, "PLTE", 4)) { if (plte) plte->assign(data, data + len); }
        else if (!std::memcmp(type, "tRNS", 4)) { hdr.hasTrns = true; if (trns) trns->assign(data, data + len); }
        else if (!std::memcmp(type, "IDAT", 4)) { if (idat) idat->insert(idat->end(), data, data + len); }
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += size_t(len) + 12;
    }
    if (!sawHeader || hdr.width == 0 || hdr.height == 0) return fail("PNG without a valid IHDR");
    if (hdr.width > (1u << 24) || hdr.height > (1u << 24)) return fail("PNG too large");
    return true;
}

int samplesPerPixel(int colorType)
{
    switch (colorType)
    {
    case 0: return 1;
    case 2: return 3;
    case 3: return 1;
    case 4: return 2;
    case 6: return 4;
    default: return 0;
    }
}

// Channel count the way stb reports it: palette images are RGB (RGBA with tRNS), a tRNS chunk
// on grey / RGB adds an alpha channel.
int fileChannels(const PngHeader& h)
{
    if (h.colorType == 3) return h.hasTrns ? 4 : 3;
    const int n = samplesPerPixel(h.colorType);
    return (h.hasTrns && (h.colorType == 0 || h.colorType == 2)) ? n + 1 : n;
}

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Undo the per-scanline filters of one (sub)image in place; `raw` holds (1 + rowBytes) * rows.
bool unfilter(uint8_t* raw, size_t rowBytes, size_t rows, size_t bpp)
{
    for (size_t y = 0; y < rows; ++y)
    {
        uint8_t* line = raw + y * (rowBytes + 1);
        const int filter = line[0];
        uint8_t* cur = line + 1;
        const uint8_t* up = y ? cur - (rowBytes + 1) : nullptr;
        for (size_t x = 0; x < rowBytes; ++x)
        {
            const int a = x >= bpp ? cur[x - bpp] : 0;
            const int b = up ? up[x] : 0;
            const int c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = cur[x];
            switch (filter)
            {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: return fail("bad PNG filter type");
            }
            cur[x] = uint8_t(v);
        }
    }
    return true;
}

}  // namespace

const std::string& LastImageError() { return g_error; }

// Larger than anything the upscaler accepts (2^27 pixels); keeps a damaged header from asking for terabytes.
static constexpr uint64_t kMaxPixels = uint64_t(1) << 28;

static bool isPng(const std::vector<uint8_t>& file)
{
    static const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    return file.size() >= 8 && std::memcmp(file.data(), kSig, 8) == 0;
}

bool LoadImageInfoFromFile(const char* fileName, int& width, int& height, int& channels, int& reqChannels)
{
    std::vector<uint8_t> file;
    PngHeader hdr;
    if (!readFile(fileName, file)) return false;
    if (!isPng(file))      // the other formats stb_image accepts: BMP, binary PNM, TGA
    {
        std::string err;
        if (!formats::Info(file, width, height, channels, err)) return fail(err);
        reqChannels = channels != 3 ? channels : 4;
        return true;
    }
    if (!parseChunks(file, hdr, nullptr, nullptr, nullptr)) return false;
    width = int(hdr.width);
    height = int(hdr.height);
    channels = fileChannels(hdr);
    reqChannels = channels != 3 ? channels : 4;     // XUSGTextureLoader.h:20
    return channels != 0;
}

std::vector<uint8_t> LoadImageFromFile(const char* fileName, int& width, int& height, int& reqChannels)
{
    std::vector<uint8_t> file, idat, plte, trns;
    PngHeader hdr;
    if (!readFile(fileName, file)) return {};
    if (!isPng(file))
    {
        std::string err;
        std::vector<uint8_t> px;
        int channels = 0;
        if (!formats::Info(file, width, height, channels, err)) { fail(err); return {}; }
        if (uint64_t(width) * uint64_t(height) > kMaxPixels) { fail("image too large"); return {}; }
        if (!formats::Decode(file, width, height, channels, px, err)) { fail(err); return {}; }
        reqChannels = channels != 3 ? channels : 4;     // XUSGTextureLoader.h:20
        if (channels != 3) return px;
        std::vector<uint8_t> rgba(size_t(width) * height * 4);
        for (size_t i = 0, n = size_t(width) * height; i < n; ++i)
        {
            rgba[4 * i] = px[3 * i]; rgba[4 * i + 1] = px[3 * i + 1]; rgba[4 * i + 2] = px[3 * i + 2]; rgba[4 * i + 3] = 255;
        }
        return rgba;
    }
    if (!parseChunks(file, hdr, &idat, &plte, &trns)) return {};
    if (uint64_t(hdr.width) * uint64_t(hdr.height) > kMaxPixels) { fail("image too large"); return {}; }
    const int spp = samplesPerPixel(hdr.colorType);
    const int depth = hdr.depth;
    const bool depthOk = (hdr.colorType == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                         (hdr.colorType == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                         ((hdr.colorType == 2 || hdr.colorType == 4 || hdr.colorType == 6) && (depth == 8 || depth == 16));
    if (!spp || !depthOk) { fail("unsupported PNG colour type / bit depth"); return {}; }
    if (hdr.interlace > 1) { fail("unsupported PNG interlace method"); return {}; }
    const size_t W = hdr.width, H = hdr.height;
    const size_t bitsPerPixel = size_t(spp) * depth;
    const size_t bpp = (bitsPerPixel + 7) / 8;

    // Adam7 passes: {x0, y0, dx, dy}; a non-interlaced image is one pass {0,0,1,1}.
    static const int kAdam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const int kNone[1][4] = {{0, 0, 1, 1}};
    const int (*passes)[4] = hdr.interlace ? kAdam7 : kNone;
    const int numPasses = hdr.interlace ? 7 : 1;

    size_t rawSize = 0;
    for (int p = 0; p < numPasses; ++p)
    {
        const size_t pw = (W - passes[p][0] + passes[p][2] - 1) / passes[p][2];
        const size_t ph = (H - passes[p][1] + passes[p][3] - 1) / passes[p][3];
        if (pw && ph) rawSize += ((pw * bitsPerPixel + 7) / 8 + 1) * ph;
    }
    std::vector<uint8_t> raw(rawSize);
    {
        z_stream zs{};
        if (inflateInit(&zs) != Z_OK) { fail("inflateInit failed"); return {}; }
        zs.next_in = idat.data();
        zs.avail_in = uInt(idat.size());
        zs.next_out = raw.data();
        zs.avail_out = uInt(raw.size());
        const int rc = inflate(&zs, Z_FINISH);
        const size_t got = zs.total_out;
        inflateEnd(&zs);
        if ((rc != Z_STREAM_END && rc != Z_OK && rc != Z_BUF_ERROR) || got != raw.size())
        {
            fail("corrupt PNG image data");
            return {};
        }
    }

    // samples as 8-bit values, spp per pixel, before palette / tRNS expansion
    std::vector<uint8_t> samples(W * H * spp);
    std::vector<uint16_t> wide;                  // 16-bit samples kept for tRNS colour-key matching
    if (depth == 16 && hdr.hasTrns) wide.resize(W * H * spp);
    size_t off = 0;
    for (int p = 0; p < numPasses; ++p)
    {
        const size_t pw = (W - passes[p][0] + passes[p][2] - 1) / passes[p][2];
        const size_t ph = (H - passes[p][1] + passes[p][3] - 1) / passes[p][3];
        if (!pw || !ph) continue;
        const size_t rowBytes = (pw * bitsPerPixel + 7) / 8;
        if (!unfilter(&raw[off], rowBytes, ph, bpp)) return {};
        for (size_t y = 0; y < ph; ++y)
        {
            const uint8_t* line = &raw[off + y * (rowBytes + 1) + 1];
            for (size_t x = 0; x < pw; ++x)
            {
                const size_t ox = passes[p][0] + x * passes[p][2], oy = passes[p][1] + y * passes[p][3];
                uint8_t* dst = &samples[(oy * W + ox) * spp];
                for (int s = 0; s < spp; ++s)
                {
                    if (depth == 8) dst[s] = line[x * spp + s];
                    else if (depth == 16)
                    {
                        dst[s] = line[(x * spp + s) * 2];            // high byte, as stb does
                        if (!wide.empty()) wide[(oy * W + ox) * spp + s] = uint16_t((line[(x * spp + s) * 2] << 8) | line[(x * spp + s) * 2 + 1]);
                    }
                    else
                    {
                        const size_t bit = x * depth;                // spp == 1 here
                        const int v = (line[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                        // grey samples scale to 0..255, palette indices stay indices
                        dst[s] = hdr.colorType == 3 ? uint8_t(v) : uint8_t(v * 255 / ((1 << depth) - 1));
                    }
                }
            }
        }
        off += (rowBytes + 1) * ph;
    }

    const int channels = fileChannels(hdr);
    reqChannels = channels != 3 ? channels : 4;
    width = int(W);
    height = int(H);
    std::vector<uint8_t> out(W * H * reqChannels);
    for (size_t i = 0; i < W * H; ++i)
    {
        uint8_t px[4] = {0, 0, 0, 255};
        int n = spp;
        if (hdr.colorType == 3)
        {
            const size_t idx = samples[i];
            if (idx * 3 + 2 >= plte.size()) { fail("PNG palette index out of range"); return {}; }
            px[0] = plte[idx * 3]; px[1] = plte[idx * 3 + 1]; px[2] = plte[idx * 3 + 2];
            px[3] = idx < trns.size() ? trns[idx] : 255;
            n = hdr.hasTrns ? 4 : 3;
        }
        else
        {
            for (int s = 0; s < spp; ++s) px[s] = samples[i * spp + s];
            if (hdr.hasTrns && (hdr.colorType == 0 || hdr.colorType == 2))
            {
                // colour key: matching pixels become transparent
                bool match = trns.size() >= size_t(spp) * 2;
                for (int s = 0; match && s < spp; ++s)
                {
                    const uint16_t key = uint16_t((trns[2 * s] << 8) | trns[2 * s + 1]);
                    if (depth == 16) match = wide[i * spp + s] == key;
                    else if (depth == 8) match = samples[i * spp + s] == (key & 0xff);
                    else match = samples[i * spp + s] == uint8_t((key & ((1 << depth) - 1)) * 255 / ((1 << depth) - 1));
                }
                px[spp] = match ? 0 : 255;
                n = spp + 1;
            }
        }
        uint8_t* dst = &out[i * reqChannels];
        if (n == reqChannels) std::memcpy(dst, px, size_t(n));
        else { dst[0] = px[0]; dst[1] = px[1]; dst[2] = px[2]; dst[3] = 255; }   // RGB -> RGBA
    }
    return out;
}

bool WritePng(const char* fileName, int w, int h, int comp, const void* data, int strideInBytes)
{
    if (w <= 0 || h <= 0 || comp < 1 || comp > 4 || !data) return fail("WritePng: bad argument");
    const size_t rowBytes = size_t(w) * comp;
    const size_t stride = strideInBytes ? size_t(strideInBytes) : rowBytes;
    const uint8_t* src = static_cast<const uint8_t*>(data);

    // per-row filter choice by the minimum-sum-of-absolute-differences heuristic
    std::vector<uint8_t> raw((rowBytes + 1) * h), trial(rowBytes);
    for (int y = 0; y < h; ++y)
    {
        const uint8_t* cur = src + size_t(y) * stride;
        const uint8_t* up = y ? cur - stride : nullptr;
        long best = -1;
        for (int f = 0; f < 5; ++f)
        {
            long sum = 0;
            for (size_t x = 0; x < rowBytes; ++x)
            {
                const int a = x >= size_t(comp) ? cur[x - comp] : 0;
                const int b = up ? up[x] : 0;
                const int c = (up && x >= size_t(comp)) ? up[x - comp] : 0;
                int pred = 0;
                if (f == 1) pred = a; else if (f == 2) pred = b; else if (f == 3) pred = (a + b) >> 1; else if (f == 4) pred = paeth(a, b, c);
                trial[x] = uint8_t(cur[x] - pred);
                sum += std::abs(int(int8_t(trial[x])));
            }
            if (best < 0 || sum < best)
            {
                best = sum;
                raw[size_t(y) * (rowBytes + 1)] = uint8_t(f);
                std::memcpy(&raw[size_t(y) * (rowBytes + 1) + 1], trial.data(), rowBytes);
            }
        }
    }
    uLongf zlen = compressBound(uLong(raw.size()));
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), uLong(raw.size()), 6) != Z_OK) return fail("WritePng: deflate failed");

    FILE* f = std::fopen(fileName, "wb");
    if (!f) return fail(std::string("cannot create ") + fileName);
    const auto chunk = [f](const char* type, const uint8_t* d, uint32_t n)
    {
        uint8_t hdr[8] = {uint8_t(n >> 24), uint8_t(n >> 16), uint8_t(n >> 8), uint8_t(n), uint8_t(type[0]), uint8_t(type[1]), uint8_t(type[2]), uint8_t(type[3])};
        std::fwrite(hdr, 1, 8, f);
        if (n) std::fwrite(d, 1, n, f);
        uLong crc = crc32(0L, hdr + 4, 4);
        if (n) crc = crc32(crc, d, n);
        const uint8_t c[4] = {uint8_t(crc >> 24), uint8_t(crc >> 16), uint8_t(crc >> 8), uint8_t(crc)};
        std::fwrite(c, 1, 4, f);
    };
    std::fwrite(kSignature, 1, 8, f);
    static const uint8_t kColorType[5] = {0, 0, 4, 2, 6};
    const uint8_t ihdr[13] = {uint8_t(w >> 24), uint8_t(w >> 16), uint8_t(w >> 8), uint8_t(w), uint8_t(h >> 24), uint8_t(h >> 16),
                              uint8_t(h >> 8), uint8_t(h), 8, kColorType[comp], 0, 0, 0};
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), uint32_t(zlen));
    chunk("IEND", nullptr, 0);
    const bool ok = std::fclose(f) == 0;
    return ok ? true : fail("WritePng: write failed");
}

}  // namespace srx
