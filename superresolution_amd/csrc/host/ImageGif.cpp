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

// ---- The raster of one image, in three separate passes.
//
// 1. gatherSubBlocks: the image data is a chain of sub-blocks (length byte + bytes, ended by a zero length or by the end
//    of a file cut short); they are joined into one byte string.
// 2. lzwIndices: the byte string is a variable-width LZW code stream (least significant bit first, at most 12 bits, clear
//    code 2^n, end code 2^n + 1).  The dictionary keeps, per code, its prefix code, its last symbol and the LENGTH of its
//    string, so a code's string is written back to front into its final place in the index buffer: no stack, no
//    recursion.  Decoding stops at the end code, at the end of the data, or once the image's w * h indices are there.
// 3. drawIndices: the indices go to the screen in raster order of the image's rectangle, row order 0, 8, 16 .. / 4, 12 .. /
//    2, 6 .. / 1, 3 .. when interlaced; entries with alpha <= 128 (the transparent index) leave the screen as it is.
// The accept / reject rules are those of the loader the reference uses (Common/stb_image.h:6684-6774), so that a file it
// refuses is refused here: a stream must open with a clear code, a code may exceed the next free one by nothing, and the
// dictionary is allowed to grow to 8192 entries before "too many codes".
std::vector<uint8_t> gatherSubBlocks(GifReader& r)
{
    std::vector<uint8_t> data;
    while (!r.atEnd())
    {
        const int len = r.get8();
        if (len == 0) break;
        // a sub-block cut short by the end of the file reads as zeros up to its declared length, as the reference's
        // loader reads it
        for (int i = 0; i < len; ++i) data.push_back(uint8_t(r.get8()));
    }
    return data;
}

bool lzwIndices(const std::vector<uint8_t>& data, int minCodeSize, size_t wanted, std::vector<uint8_t>& indices, std::string& err)
{
    constexpr int kMaxCodes = 8192;
    struct Entry { int prefix; uint16_t length; uint8_t first, last; };
    std::vector<Entry> dict(kMaxCodes);
    const int clearCode = 1 << minCodeSize, endCode = clearCode + 1;
    for (int i = 0; i < clearCode; ++i) dict[size_t(i)] = Entry{-1, 1, uint8_t(i), uint8_t(i)};

    int width = minCodeSize + 1, nextFree = clearCode + 2, previous = -1;
    bool cleared = false;
    uint32_t window = 0;
    int windowBits = 0;
    size_t at = 0;
    indices.clear();
    indices.reserve(wanted);
    for (;;)
    {
        while (windowBits < width)
        {
            if (at >= data.size()) return true;                     // data exhausted: keep what was decoded
            window |= uint32_t(data[at++]) << windowBits;
            windowBits += 8;
        }
        const int code = int(window & ((1u << width) - 1u));
        window >>= width;
        windowBits -= width;

        if (code == clearCode)
        {
            width = minCodeSize + 1;
            nextFree = clearCode + 2;
            previous = -1;
            cleared = true;
            continue;
        }
        if (code == endCode) return true;
        if (code > nextFree) { err = "GIF: illegal code in raster"; return false; }
        if (!cleared) { err = "GIF: no clear code"; return false; }
        if (previous >= 0)
        {
            if (nextFree + 1 > kMaxCodes) { err = "GIF: too many codes"; return false; }
            // new entry = string(previous) + first symbol of string(code); for code == nextFree that is string(previous)
            // + its own first symbol
            Entry& e = dict[size_t(nextFree)];
            e.prefix = previous;
            e.first = dict[size_t(previous)].first;
            e.length = uint16_t(dict[size_t(previous)].length + 1);
            e.last = code == nextFree ? e.first : dict[size_t(code)].first;
            ++nextFree;
        }
        else if (code == nextFree) { err = "GIF: illegal code in raster"; return false; }

        // the string of `code`, written from its last symbol back to its first; what exceeds the image is dropped
        const size_t start = indices.size(), length = dict[size_t(code)].length;
        const size_t keep = start >= wanted ? 0 : (length < wanted - start ? length : wanted - start);
        indices.resize(start + keep);
        size_t pos = length;
        for (int c = code; c >= 0 && pos > 0; c = dict[size_t(c)].prefix)
        {
            --pos;
            if (pos < keep) indices[start + pos] = dict[size_t(c)].last;
        }
        if ((nextFree & ((1 << width) - 1)) == 0 && nextFree <= 0x0fff) ++width;
        previous = code;
    }
}

// Row of the image that the k-th transmitted row of an interlaced image of `h` rows is.
std::vector<int> interlacedRows(int h)
{
    std::vector<int> rows;
    rows.reserve(size_t(h));
    static const int first[4] = {0, 4, 2, 1}, step[4] = {8, 8, 4, 2};
    for (int pass = 0; pass < 4; ++pass)
        for (int y = first[pass]; y < h; y += step[pass]) rows.push_back(y);
    return rows;
}

void drawIndices(const std::vector<uint8_t>& indices, const uint8_t (*table)[4], int x, int y, int w, int h, bool interlaced,
                 int screenW, uint8_t* screen, uint8_t* touched)
{
    if (w <= 0 || h <= 0) return;
    const std::vector<int> order = interlaced ? interlacedRows(h) : std::vector<int>();
    for (size_t n = 0; n < indices.size(); ++n)
    {
        const int k = int(n / size_t(w)), col = int(n % size_t(w));
        if (k >= h) break;
        const int row = interlaced ? order[size_t(k)] : k;
        const size_t pixel = size_t(y + row) * size_t(screenW) + size_t(x + col);
        touched[pixel] = 1;
        const uint8_t* e = table[indices[n]];                      // stored blue, green, red, alpha
        if (e[3] > 128)
        {
            uint8_t* d = screen + pixel * 4;
            d[0] = e[2]; d[1] = e[1]; d[2] = e[0]; d[3] = e[3];
        }
    }
}

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
            const int lflags = r.get8();
            const uint8_t (*table)[4] = nullptr;
            if (lflags & 0x80)
            {
                colorTable(r, g.lpal, 2 << (lflags & 7), (g.eflags & 1) ? g.transparent : -1);
                table = g.lpal;
            }
            else if (g.flags & 0x80) table = g.pal;
            else { err = "GIF: missing colour table"; return false; }
            const int minCodeSize = r.get8();
            if (minCodeSize > 12) { err = "GIF: bad code size"; return false; }
            const std::vector<uint8_t> data = gatherSubBlocks(r);
            std::vector<uint8_t> indices;
            if (!lzwIndices(data, minCodeSize, size_t(w) * size_t(h), indices, err)) return false;
            drawIndices(indices, table, x, y, w, h, (lflags & 0x40) != 0, g.w, pixels.data(), touched.data());
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
