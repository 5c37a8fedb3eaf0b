// The LZW raster decoder below (Raster::emit / Raster::run) restates stbi__out_gif_code / stbi__process_gif_raster of
// stb_image v2.30 (Sean Barrett and contributors; public domain / MIT No Attribution), so that GIFs decode to the bytes
// the reference's loader gives.
// GIF decoding (first image of the file) for the host program's image loader.
//
// The reference loads images through stb_image v2.30 (XUSG/Advanced/XUSGTextureLoader.h:17-32);
// for a GIF its stbi_load returns the first image composed on the logical screen as RGBA
// (Common/stb_image.h:6597-6950, 7048-7075).  What decides the bytes, restated from there:
//   * the screen starts as transparent black (0,0,0,0);
//   * a pixel whose palette entry is the transparent index of the preceding graphic control
//     extension is not drawn (the screen keeps what it had), every other pixel is opaque;
//   * when the background index is not 0, the screen pixels that the image's rectangle does not
//     touch get the background palette entry, opaque — copied in the palette's stored order, which
//     is blue, green, red (:6895-6902): such pixels come out with red and blue exchanged, and so
//     they do here;
//   * interlaced images: rows 0, 8, 16 .. then 4, 12 .. then 2, 6 .. then 1, 3 ..
// LZW codes are at most 12 bits; a stream must begin with a clear code.
#include "ImageFormats.h"

#include <cstring>

namespace srx {
namespace formats {

namespace {

struct GifReader
{
    const std::vector<uint8_t>& f;
    size_t pos = 0;
    explicit GifReader(const std::vector<uint8_t>& file) : f(file) {}
    int get8() { return pos < f.size() ? f[pos++] : 0; }
    int get16() { const int a = get8(); return a | (get8() << 8); }
    void skip(size_t n) { pos = (pos >= f.size() || n > f.size() - pos) ? f.size() : pos + n; }
    bool atEnd() const { return pos >= f.size(); }
};

struct Gif
{
    int w = 0, h = 0, flags = 0, bgIndex = 0, transparent = -1, eflags = 0;
    uint8_t pal[256][4] = {}, lpal[256][4] = {};     // stored blue, green, red, alpha
};

bool header(GifReader& r, Gif& g, std::string& err)
{
    if (r.get8() != 'G' || r.get8() != 'I' || r.get8() != 'F' || r.get8() != '8') { err = "GIF: bad signature"; return false; }
    const int version = r.get8();
    if ((version != '7' && version != '9') || r.get8() != 'a') { err = "GIF: bad signature"; return false; }
    g.w = r.get16();
    g.h = r.get16();
    g.flags = r.get8();
    g.bgIndex = r.get8();
    r.get8();      // aspect ratio
    if (g.w <= 0 || g.h <= 0 || g.w > (1 << 24) || g.h > (1 << 24)) { err = "GIF: bad size"; return false; }
    return true;
}

void colorTable(GifReader& r, uint8_t pal[256][4], int entries, int transparent)
{
    for (int i = 0; i < entries; ++i)
    {
        pal[i][2] = uint8_t(r.get8());
        pal[i][1] = uint8_t(r.get8());
        pal[i][0] = uint8_t(r.get8());
        pal[i][3] = transparent == i ? 0 : 255;
    }
}

// The LZW raster of one image, drawn into `out` (RGBA screen) within the image's rectangle.
struct Raster
{
    struct Code { int16_t prefix; uint8_t first, suffix; };
    Code codes[8192];
    uint8_t* out;
    uint8_t* touched;
    const uint8_t (*table)[4];
    long lineSize, startX, startY, maxX, maxY, curX, curY, step;
    int parse;

    void emit(int code)
    {
        // the chain of prefixes is at most 4096 long: walk it with an explicit stack, oldest first
        uint16_t stack[8192];
        int n = 0;
        for (int c = code; c >= 0 && n < 8192; c = codes[c].prefix) stack[n++] = uint16_t(c);
        while (n > 0)
        {
            const int c = stack[--n];
            if (curY >= maxY) return;
            const long idx = curX + curY;
            touched[idx / 4] = 1;
            const uint8_t* e = table[codes[c].suffix];
            if (e[3] > 128)
            {
                out[idx] = e[2]; out[idx + 1] = e[1]; out[idx + 2] = e[0]; out[idx + 3] = e[3];
            }
            curX += 4;
            if (curX >= maxX)
            {
                curX = startX;
                curY += step;
                while (curY >= maxY && parse > 0)
                {
                    step = (1L << parse) * lineSize;
                    curY = startY + (step >> 1);
                    --parse;
                }
            }
        }
    }

    bool run(GifReader& r, std::string& err)
    {
        const int lzwCs = r.get8();
        if (lzwCs > 12) { err = "GIF: bad code size"; return false; }
        const int clear = 1 << lzwCs;
        bool first = true;
        int codeSize = lzwCs + 1, codeMask = (1 << codeSize) - 1, avail = clear + 2, oldCode = -1, validBits = 0, len = 0;
        unsigned bits = 0;
        for (int i = 0; i < clear; ++i) codes[i] = Code{-1, uint8_t(i), uint8_t(i)};
        for (;;)
        {
            if (validBits < codeSize)
            {
                if (len == 0)
                {
                    if (r.atEnd()) return true;          // a file cut short ends the raster
                    len = r.get8();
                    if (len == 0) return true;
                }
                --len;
                bits |= unsigned(r.get8()) << validBits;
                validBits += 8;
                continue;
            }
            const int code = int(bits & unsigned(codeMask));
            bits >>= codeSize;
            validBits -= codeSize;
            if (code == clear)
            {
                codeSize = lzwCs + 1;
                codeMask = (1 << codeSize) - 1;
                avail = clear + 2;
                oldCode = -1;
                first = false;
            }
            else if (code == clear + 1)
            {
                r.skip(size_t(len));
                while ((len = r.get8()) > 0) r.skip(size_t(len));
                return true;
            }
            else if (code <= avail)
            {
                if (first) { err = "GIF: no clear code"; return false; }
                if (oldCode >= 0)
                {
                    if (avail + 1 > 8192) { err = "GIF: too many codes"; return false; }
                    Code& p = codes[avail++];
                    p.prefix = int16_t(oldCode);
                    p.first = codes[oldCode].first;
                    p.suffix = code == avail ? p.first : codes[code].first;
                }
                else if (code == avail) { err = "GIF: illegal code in raster"; return false; }
                emit(code);
                if ((avail & codeMask) == 0 && avail <= 0x0fff)
                {
                    ++codeSize;
                    codeMask = (1 << codeSize) - 1;
                }
                oldCode = code;
            }
            else { err = "GIF: illegal code in raster"; return false; }
        }
    }
};

}  // namespace

bool IsGif(const std::vector<uint8_t>& file) { return file.size() >= 6 && std::memcmp(file.data(), "GIF8", 4) == 0; }

bool GifInfo(const std::vector<uint8_t>& file, int& width, int& height, int& channels, std::string& err)
{
    GifReader r(file);
    Gif g;
    if (!header(r, g, err)) return false;
    width = g.w;
    height = g.h;
    channels = 4;
    return true;
}

bool GifDecode(const std::vector<uint8_t>& file, int& width, int& height, int& channels, std::vector<uint8_t>& pixels,
               std::string& err)
{
    GifReader r(file);
    Gif g;
    if (!header(r, g, err)) return false;
    if (g.flags & 0x80) colorTable(r, g.pal, 2 << (g.flags & 7), -1);
    width = g.w;
    height = g.h;
    channels = 4;
    const size_t count = size_t(g.w) * g.h;
    pixels.assign(count * 4, 0);
    std::vector<uint8_t> touched(count, 0);
    for (;;)
    {
        if (r.atEnd()) { err = "GIF: no image in the file"; return false; }
        const int tag = r.get8();
        if (tag == 0x2c)
        {
            const int x = r.get16(), y = r.get16(), w = r.get16(), h = r.get16();
            if (x + w > g.w || y + h > g.h) { err = "GIF: bad image descriptor"; return false; }
            std::vector<Raster> holder(1);
            Raster& ra = holder[0];
            ra.out = pixels.data();
            ra.touched = touched.data();
            ra.lineSize = long(g.w) * 4;
            ra.startX = long(x) * 4;
            ra.startY = long(y) * ra.lineSize;
            ra.maxX = ra.startX + long(w) * 4;
            ra.maxY = ra.startY + long(h) * ra.lineSize;
            ra.curX = ra.startX;
            ra.curY = w == 0 ? ra.maxY : ra.startY;
            const int lflags = r.get8();
            if (lflags & 0x40) { ra.step = 8 * ra.lineSize; ra.parse = 3; }
            else { ra.step = ra.lineSize; ra.parse = 0; }
            if (lflags & 0x80)
            {
                colorTable(r, g.lpal, 2 << (lflags & 7), (g.eflags & 1) ? g.transparent : -1);
                ra.table = g.lpal;
            }
            else if (g.flags & 0x80) ra.table = g.pal;
            else { err = "GIF: missing colour table"; return false; }
            if (!ra.run(r, err)) return false;
            if (g.bgIndex > 0)
            {
                g.pal[g.bgIndex][3] = 255;
                for (size_t i = 0; i < count; ++i)
                    if (!touched[i]) std::memcpy(&pixels[i * 4], g.pal[g.bgIndex], 4);     // stored order: see the file comment
            }
            return true;
        }
        if (tag == 0x21)
        {
            const int ext = r.get8();
            int len;
            if (ext == 0xf9)
            {
                len = r.get8();
                if (len == 4)
                {
                    g.eflags = r.get8();
                    r.get16();       // delay
                    if (g.transparent >= 0) g.pal[g.transparent][3] = 255;
                    if (g.eflags & 1)
                    {
                        g.transparent = r.get8();
                        g.pal[g.transparent][3] = 0;
                    }
                    else
                    {
                        r.skip(1);
                        g.transparent = -1;
                    }
                }
                else
                {
                    r.skip(size_t(len));
                    continue;
                }
            }
            while ((len = r.get8()) != 0)
            {
                r.skip(size_t(len));
                if (r.atEnd()) break;
            }
            continue;
        }
        if (tag == 0x3b) { err = "GIF: no image in the file"; return false; }
        err = "GIF: unknown block";
        return false;
    }
}

}  // namespace formats
}  // namespace srx
