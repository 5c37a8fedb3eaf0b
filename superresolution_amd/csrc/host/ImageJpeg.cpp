// The inverse DCT (idct1d / idctBlock), the chroma upsampling and the colour conversion restate those of stb_image
// v2.30 (Sean Barrett and contributors; public domain / MIT No Attribution; the DCT is the Independent JPEG Group's
// jidctint), so that JPEGs decode to the bytes the reference's loader gives.
// Baseline and progressive (Huffman, 8-bit) JPEG decoding for the host program's image loader.
//
// The reference loads images through stb_image v2.30 (XUSG/Advanced/XUSGTextureLoader.h:17-32),
// so "the decoded result" of a JPEG is what that decoder produces, and a JPEG decoder has choices
// the standard leaves open.  The ones that decide the bytes are restated here from
// Common/stb_image.h and pinned against it in tests/test_image_io.py:
//   * the integer inverse DCT (jidctint-derived, 12-bit constants, rounding +512 >> 10 after the
//     column pass and +65536 + (128 << 17) >> 17 after the row pass), stb_image.h:2426-2524;
//   * chroma upsampling: 3:1 linear taps for factor 2 in either direction and both (rounding +2 >> 2,
//     +8 >> 4), pixel replication for other factors, stb_image.h:3465-3527, 3646-3655, and the
//     choice of the nearer / farther source row per output row, :3907-3940;
//   * YCbCr -> RGB in 20-bit fixed point with the chroma products masked as in :3657-3683;
//   * three-component files without a JFIF header whose Adobe marker says "no transform", or whose
//     component ids are 'R','G','B', are RGB already (:3879).
// A progressive file's coefficients are gathered over its scans (ITU T.81 G.1.2: spectral selection
// and successive approximation), then multiplied by the quantisation table and transformed at the end,
// as stb does (stb_image.h:2265-2420, 3025-3096).
// Arithmetic-coded, 12-bit and four-component (CMYK / YCCK) files are refused.
#include "ImageFormats.h"

#include <cstring>

namespace srx {
namespace formats {

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman
{
    bool present = false;
    int minCode[17] = {}, maxCode[18] = {}, valPtr[17] = {};
    uint8_t values[256] = {};
};

struct Component
{
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;       // true size in samples, allocated size (whole MCUs)
    int pred = 0;
    std::vector<uint8_t> data;
    std::vector<short> coeff;               // progressive files: 64 coefficients per block, blocks row by row
};

struct Decoder
{
    const std::vector<uint8_t>& f;
    size_t pos = 0;
    std::string& err;
    int width = 0, height = 0, numComp = 0, hMax = 1, vMax = 1, mcuX = 0, mcuY = 0, restartInterval = 0;
    bool jfif = false;
    int adobeTransform = -1;
    uint16_t dequant[4][64] = {};
    bool haveQuant[4] = {};
    Huffman dc[4], ac[4];
    Component comp[3];
    // bit reader
    uint32_t bitBuf = 0;
    int bitCount = 0;
    bool hitMarker = false;
    // progressive files
    bool progressive = false;
    int specStart = 0, specEnd = 63, succHigh = 0, succLow = 0, eobRun = 0;

    Decoder(const std::vector<uint8_t>& file, std::string& e) : f(file), err(e) {}

    bool fail(const char* why) { err = why; return false; }
    int get8() { return pos < f.size() ? f[pos++] : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }

    // ---- entropy-coded bits: 0xFF 0x00 is a literal 0xFF, 0xFF + anything else ends the data (zeros follow)
    int bit()
    {
        if (bitCount == 0)
        {
            int b = 0;
            if (!hitMarker && pos < f.size())
            {
                b = f[pos];
                if (b == 0xff)
                {
                    if (pos + 1 < f.size() && f[pos + 1] == 0) pos += 2;
                    else { hitMarker = true; b = 0; }
                }
                else ++pos;
            }
            bitBuf = uint32_t(b);
            bitCount = 8;
        }
        --bitCount;
        return int((bitBuf >> bitCount) & 1);
    }
    int bits(int n)
    {
        int v = 0;
        for (int i = 0; i < n; ++i) v = (v << 1) | bit();
        return v;
    }
    // a value of n bits whose leading 0 marks the negative half of the range
    int receiveExtend(int n)
    {
        if (n == 0) return 0;
        const int v = bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    int decodeSymbol(const Huffman& h)
    {
        int code = 0;
        for (int len = 1; len <= 16; ++len)
        {
            code = (code << 1) | bit();
            if (h.maxCode[len] >= 0 && code <= h.maxCode[len] && code >= h.minCode[len])
                return h.values[h.valPtr[len] + code - h.minCode[len]];
        }
        return -1;
    }
    void resetEntropy()
    {
        bitCount = 0;
        hitMarker = false;
        for (Component& c : comp) c.pred = 0;
        eobRun = 0;
    }

    bool buildHuffman(Huffman& h, const int counts[16], const uint8_t* vals, int total)
    {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len)
        {
            h.valPtr[len] = k;
            h.minCode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            h.maxCode[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) return fail("JPEG: bad Huffman code lengths");
            code <<= 1;
        }
        std::memcpy(h.values, vals, size_t(total));
        h.present = true;
        return true;
    }

    // ---- marker segments
    bool segment(int marker)
    {
        if (marker == 0xdd)      // DRI
        {
            if (get16() != 4) return fail("JPEG: bad DRI length");
            restartInterval = get16();
            return true;
        }
        int len = get16() - 2;
        if (len < 0 || pos + size_t(len) > f.size()) return fail("JPEG: segment runs past the end of the file");
        const size_t end = pos + size_t(len);
        if (marker == 0xdb)      // DQT
        {
            while (pos < end)
            {
                const int q = get8(), precision = q >> 4, t = q & 15;
                if (precision > 1 || t > 3) return fail("JPEG: bad quantisation table");
                for (int i = 0; i < 64; ++i) dequant[t][kZigzag[i]] = uint16_t(precision ? get16() : get8());
                haveQuant[t] = true;
            }
        }
        else if (marker == 0xc4)  // DHT
        {
            while (pos < end)
            {
                const int q = get8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) return fail("JPEG: bad Huffman table header");
                int counts[16], total = 0;
                for (int i = 0; i < 16; ++i) total += counts[i] = get8();
                if (total > 256 || pos + size_t(total) > f.size()) return fail("JPEG: bad Huffman table");
                if (!buildHuffman(tc ? ac[th] : dc[th], counts, &f[pos], total)) return false;
                pos += size_t(total);
            }
        }
        else if (marker == 0xe0 && len >= 5)   // APP0: JFIF
        {
            if (std::memcmp(&f[pos], "JFIF\0", 5) == 0) jfif = true;
        }
        else if (marker == 0xee && len >= 12)  // APP14: Adobe, the colour transform is its last byte
        {
            if (std::memcmp(&f[pos], "Adobe\0", 6) == 0) adobeTransform = f[pos + 11];
        }
        pos = end;
        return true;
    }

    bool frameHeader()
    {
        const int len = get16();
        if (get8() != 8) return fail("JPEG: only 8 bits per sample are supported");
        height = get16();
        width = get16();
        numComp = get8();
        if (width <= 0 || height <= 0) return fail("JPEG: bad image size");
        if (numComp == 4) return fail("JPEG: four-component (CMYK) files are not supported");
        if (numComp != 1 && numComp != 3) return fail("JPEG: bad component count");
        if (len != 8 + 3 * numComp) return fail("JPEG: bad frame header length");
        for (int i = 0; i < numComp; ++i)
        {
            Component& c = comp[i];
            c.id = get8();
            const int q = get8();
            c.h = q >> 4;
            c.v = q & 15;
            c.tq = get8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("JPEG: bad component description");
            if (c.h > hMax) hMax = c.h;
            if (c.v > vMax) vMax = c.v;
        }
        for (int i = 0; i < numComp; ++i)
            if (hMax % comp[i].h != 0 || vMax % comp[i].v != 0) return fail("JPEG: sampling factors do not divide the largest");
        mcuX = (width + 8 * hMax - 1) / (8 * hMax);
        mcuY = (height + 8 * vMax - 1) / (8 * vMax);
        return true;
    }

    void allocate()
    {
        for (int i = 0; i < numComp; ++i)
        {
            Component& c = comp[i];
            c.x = (width * c.h + hMax - 1) / hMax;
            c.y = (height * c.v + vMax - 1) / vMax;
            c.w2 = mcuX * c.h * 8;
            c.h2 = mcuY * c.v * 8;
            c.data.assign(size_t(c.w2) * c.h2, 0);
            if (progressive) c.coeff.assign(size_t(c.w2) * c.h2, 0);
        }
    }

    // a DC value of a valid file fits 16 bits; the bound only keeps damaged files inside int
    static int boundedPred(int v) { return v < -(1 << 20) ? -(1 << 20) : (v > (1 << 20) ? (1 << 20) : v); }
    static uint8_t clamp(long long x) { return uint8_t(x < 0 ? 0 : (x > 255 ? 255 : x)); }

    // One pass of the jidctint-style 1-D transform on eight values: even part x0..x3, odd part t0..t3.
    // 64-bit intermediates: a valid file stays far inside 32 bits (where stb computes), a damaged one
    // must not overflow.
    typedef long long wide;
    static void idct1d(const wide s[8], wide x[4], wide t[4])
    {
        // the constants are float literals scaled by 4096 and rounded by adding 0.5 and truncating
        // (so the negative ones round toward zero from one half above), as in stb_image.h:2426
        constexpr auto f2f = [](float v) { return int(v * 4096 + 0.5); };
        wide p2 = s[2], p3 = s[6];
        wide p1 = (p2 + p3) * f2f(0.5411961f);
        const wide t2e = p1 + p3 * f2f(-1.847759065f);
        const wide t3e = p1 + p2 * f2f(0.765366865f);
        p2 = s[0];
        p3 = s[4];
        const wide t0e = (p2 + p3) * 4096, t1e = (p2 - p3) * 4096;
        x[0] = t0e + t3e; x[3] = t0e - t3e; x[1] = t1e + t2e; x[2] = t1e - t2e;
        wide t0 = s[7], t1 = s[5], t2 = s[3], t3 = s[1];
        p3 = t0 + t2;
        wide p4 = t1 + t3;
        p1 = t0 + t3;
        p2 = t1 + t2;
        const wide p5 = (p3 + p4) * f2f(1.175875602f);
        t0 = t0 * f2f(0.298631336f);
        t1 = t1 * f2f(2.053119869f);
        t2 = t2 * f2f(3.072711026f);
        t3 = t3 * f2f(1.501321110f);
        p1 = p5 + p1 * f2f(-0.899976223f);
        p2 = p5 + p2 * f2f(-2.562915447f);
        p3 = p3 * f2f(-1.961570560f);
        p4 = p4 * f2f(-0.390180644f);
        t[3] = t3 + p1 + p4; t[2] = t2 + p2 + p3; t[1] = t1 + p2 + p4; t[0] = t0 + p1 + p3;
    }

    static void idctBlock(uint8_t* out, int stride, const short d[64])
    {
        wide val[64];
        for (int i = 0; i < 8; ++i)
        {
            if (!d[i + 8] && !d[i + 16] && !d[i + 24] && !d[i + 32] && !d[i + 40] && !d[i + 48] && !d[i + 56])
            {
                for (int r = 0; r < 8; ++r) val[i + 8 * r] = d[i] * 4;
                continue;
            }
            const wide s[8] = {d[i], d[i + 8], d[i + 16], d[i + 24], d[i + 32], d[i + 40], d[i + 48], d[i + 56]};
            wide x[4], t[4];
            idct1d(s, x, t);
            for (int k = 0; k < 4; ++k) x[k] += 512;
            val[i] = (x[0] + t[3]) >> 10;      val[i + 56] = (x[0] - t[3]) >> 10;
            val[i + 8] = (x[1] + t[2]) >> 10;  val[i + 48] = (x[1] - t[2]) >> 10;
            val[i + 16] = (x[2] + t[1]) >> 10; val[i + 40] = (x[2] - t[1]) >> 10;
            val[i + 24] = (x[3] + t[0]) >> 10; val[i + 32] = (x[3] - t[0]) >> 10;
        }
        for (int r = 0; r < 8; ++r, out += stride)
        {
            wide x[4], t[4];
            idct1d(val + 8 * r, x, t);
            for (int k = 0; k < 4; ++k) x[k] += 65536 + (128 << 17);
            out[0] = clamp((x[0] + t[3]) >> 17); out[7] = clamp((x[0] - t[3]) >> 17);
            out[1] = clamp((x[1] + t[2]) >> 17); out[6] = clamp((x[1] - t[2]) >> 17);
            out[2] = clamp((x[2] + t[1]) >> 17); out[5] = clamp((x[2] - t[1]) >> 17);
            out[3] = clamp((x[3] + t[0]) >> 17); out[4] = clamp((x[3] - t[0]) >> 17);
        }
    }

    bool decodeBlock(Component& c, short d[64])
    {
        std::memset(d, 0, 64 * sizeof(short));
        const int t = decodeSymbol(dc[c.td]);
        if (t < 0 || t > 15) return fail("JPEG: bad Huffman code");
        c.pred = boundedPred(c.pred + receiveExtend(t));
        d[0] = short((long long)c.pred * dequant[c.tq][0]);
        for (int k = 1; k < 64;)
        {
            const int rs = decodeSymbol(ac[c.ta]);
            if (rs < 0) return fail("JPEG: bad Huffman code");
            const int r = rs >> 4, s = rs & 15;
            if (s == 0)
            {
                if (r != 15) break;      // end of block
                k += 16;
            }
            else
            {
                k += r;
                if (k > 63) return fail("JPEG: coefficient index out of range");
                const int z = kZigzag[k++];
                d[z] = short(receiveExtend(s) * dequant[c.tq][z]);
            }
        }
        return true;
    }

    // Progressive DC scan: the first pass stores the prediction-coded value shifted up by the scan's low
    // bit position, later passes append one bit each.
    bool decodeBlockDc(Component& c, short d[64])
    {
        if (succHigh == 0)
        {
            const int t = decodeSymbol(dc[c.td]);
            if (t < 0 || t > 15) return fail("JPEG: bad Huffman code");
            c.pred = boundedPred(c.pred + receiveExtend(t));
            d[0] = short((long long)c.pred * (1 << succLow));
        }
        else if (bit()) d[0] = short(d[0] + (1 << succLow));
        return true;
    }

    // Progressive AC scan over coefficients specStart .. specEnd of one block.
    bool decodeBlockAc(Component& c, short d[64])
    {
        const Huffman& h = ac[c.ta];
        if (succHigh == 0)
        {
            if (eobRun) { --eobRun; return true; }
            int k = specStart;
            do
            {
                const int rs = decodeSymbol(h);
                if (rs < 0) return fail("JPEG: bad Huffman code");
                const int r = rs >> 4, s = rs & 15;
                if (s == 0)
                {
                    if (r < 15)
                    {
                        eobRun = 1 << r;
                        if (r) eobRun += bits(r);
                        --eobRun;
                        break;
                    }
                    k += 16;
                }
                else
                {
                    k += r;
                    if (k > 63) return fail("JPEG: coefficient index out of range");
                    d[kZigzag[k++]] = short(receiveExtend(s) * (1 << succLow));
                }
            } while (k <= specEnd);
            return true;
        }
        // refinement: one more bit for every coefficient that is already non-zero, new +-1 coefficients between them
        const int one = 1 << succLow;
        const auto refine = [&](short& v) {
            if (bit() && (v & one) == 0) v = short(v > 0 ? v + one : v - one);
        };
        if (eobRun)
        {
            --eobRun;
            for (int k = specStart; k <= specEnd; ++k)
                if (d[kZigzag[k]] != 0) refine(d[kZigzag[k]]);
            return true;
        }
        int k = specStart;
        do
        {
            const int rs = decodeSymbol(h);
            if (rs < 0) return fail("JPEG: bad Huffman code");
            int r = rs >> 4, s = rs & 15;
            if (s == 0)
            {
                if (r < 15)
                {
                    eobRun = (1 << r) - 1;
                    if (r) eobRun += bits(r);
                    r = 64;          // run to the end of the band, refining on the way
                }
            }
            else
            {
                if (s != 1) return fail("JPEG: bad refinement code");
                s = bit() ? one : -one;
            }
            while (k <= specEnd)
            {
                short& v = d[kZigzag[k++]];
                if (v != 0) refine(v);
                else
                {
                    if (r == 0) { v = short(s); break; }
                    --r;
                }
            }
        } while (k <= specEnd);
        return true;
    }

    // All scans of a progressive file read: quantisation scale and inverse transform of every block.
    void finishProgressive()
    {
        for (int i = 0; i < numComp; ++i)
        {
            Component& c = comp[i];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3, stride = c.w2 >> 3;
            for (int j = 0; j < bh; ++j)
                for (int b = 0; b < bw; ++b)
                {
                    short* d = &c.coeff[(size_t(j) * stride + b) * 64];
                    for (int k = 0; k < 64; ++k) d[k] = short((long long)d[k] * dequant[c.tq][k]);
                    idctBlock(&c.data[size_t(j) * 8 * c.w2 + size_t(b) * 8], c.w2, d);
                }
        }
    }

    // After every restart interval the data must continue with an RSTn marker.
    bool restartIfDue(int& todo)
    {
        if (restartInterval == 0 || --todo > 0) return true;
        todo = restartInterval;
        // skip to the marker
        while (pos + 1 < f.size() && !(f[pos] == 0xff && f[pos + 1] != 0 && f[pos + 1] != 0xff)) ++pos;
        if (pos + 1 < f.size() && f[pos + 1] >= 0xd0 && f[pos + 1] <= 0xd7) pos += 2;
        else todo = 0x7fffffff;          // no restart marker: whatever follows ends the scan
        resetEntropy();
        return true;
    }

    bool scan()
    {
        const int len = get16();
        const int n = get8();
        if (n < 1 || n > numComp || len != 6 + 2 * n) return fail("JPEG: bad scan header");
        int order[3] = {0, 0, 0};
        for (int i = 0; i < n; ++i)
        {
            const int id = get8(), q = get8();
            int which = -1;
            for (int k = 0; k < numComp; ++k) if (comp[k].id == id) which = k;
            if (which < 0) return fail("JPEG: scan names an unknown component");
            comp[which].td = q >> 4;
            comp[which].ta = q & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) return fail("JPEG: bad Huffman table number");
            if (!haveQuant[comp[which].tq]) return fail("JPEG: component uses a quantisation table that was not defined");
            order[i] = which;
        }
        specStart = get8();
        specEnd = get8();
        const int a = get8();
        succHigh = a >> 4;
        succLow = a & 15;
        if (!progressive)
        {
            if (specStart != 0 || specEnd != 63 || a != 0) return fail("JPEG: bad scan parameters in a sequential file");
        }
        else if (specStart > 63 || specEnd > 63 || specStart > specEnd || succHigh > 13 || succLow > 13 ||
                 (specStart == 0 && specEnd != 0) || (specStart != 0 && n != 1))
            return fail("JPEG: bad progressive scan parameters");
        const bool dcScan = specStart == 0;
        for (int i = 0; i < n; ++i)
        {
            const Component& c = comp[order[i]];
            const bool needDc = !progressive || (dcScan && succHigh == 0), needAc = !progressive || !dcScan;
            if ((needDc && !dc[c.td].present) || (needAc && !ac[c.ta].present))
                return fail("JPEG: scan uses a Huffman table that was not defined");
        }
        resetEntropy();
        int todo = restartInterval ? restartInterval : 0x7fffffff;
        short block[64];
        // one block at (bx, by) of component c: a sequential file transforms it at once, a progressive one
        // adds this scan's part to the stored coefficients
        const auto oneBlock = [&](Component& c, int bx, int by) {
            if (!progressive)
            {
                if (!decodeBlock(c, block)) return false;
                idctBlock(&c.data[size_t(by) * 8 * c.w2 + size_t(bx) * 8], c.w2, block);
                return true;
            }
            short* d = &c.coeff[(size_t(by) * (c.w2 >> 3) + bx) * 64];
            return dcScan ? decodeBlockDc(c, d) : decodeBlockAc(c, d);
        };
        if (n == 1)
        {
            // a scan of one component covers just its own blocks, row by row
            Component& c = comp[order[0]];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i)
                {
                    if (!oneBlock(c, i, j)) return false;
                    restartIfDue(todo);
                }
            return true;
        }
        for (int my = 0; my < mcuY; ++my)
            for (int mx = 0; mx < mcuX; ++mx)
            {
                for (int k = 0; k < n; ++k)
                {
                    Component& c = comp[order[k]];
                    for (int y = 0; y < c.v; ++y)
                        for (int x = 0; x < c.h; ++x)
                            if (!oneBlock(c, mx * c.h + x, my * c.v + y)) return false;
                }
                restartIfDue(todo);
            }
        return true;
    }

    // Header only (stopAtFrame) or the whole file.
    bool run(bool stopAtFrame)
    {
        if (f.size() < 4 || f[0] != 0xff || f[1] != 0xd8) return fail("JPEG: no start-of-image marker");
        pos = 2;
        bool haveFrame = false, haveScan = false;
        for (;;)
        {
            // next marker: 0xFF fill bytes may precede it
            while (pos < f.size() && f[pos] != 0xff) ++pos;
            while (pos < f.size() && f[pos] == 0xff) ++pos;
            if (pos >= f.size()) break;
            const int m = f[pos++];
            if (m == 0xd9) break;                                   // EOI
            if (m == 0x00 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;
            if (m == 0xc0 || m == 0xc1 || m == 0xc2)
            {
                progressive = m == 0xc2;
                if (haveFrame) return fail("JPEG: more than one frame");
                if (!frameHeader()) return false;
                haveFrame = true;
                if (stopAtFrame) return true;
                allocate();
            }
            else if ((m >= 0xc3 && m <= 0xcf) && m != 0xc4 && m != 0xc8 && m != 0xcc)
                return fail("JPEG: lossless, hierarchical and arithmetic-coded files are not supported");
            else if (m == 0xcc) return fail("JPEG: arithmetic-coded files are not supported");
            else if (m == 0xda)
            {
                if (!haveFrame) return fail("JPEG: scan before the frame header");
                if (!scan()) return false;
                haveScan = true;
            }
            else if (!segment(m)) return false;
        }
        if (!haveFrame) return fail("JPEG: no frame header");
        if (!stopAtFrame && !haveScan) return fail("JPEG: no scan");
        if (!stopAtFrame && progressive) finishProgressive();
        return true;
    }

    // ---- upsampling and colour conversion
    static void resampleRow(uint8_t* out, const uint8_t* nearRow, const uint8_t* farRow, int w, int hs, int vs)
    {
        if (hs == 1 && vs == 1) { std::memcpy(out, nearRow, size_t(w)); return; }
        if (hs == 1 && vs == 2)
        {
            for (int i = 0; i < w; ++i) out[i] = uint8_t((3 * nearRow[i] + farRow[i] + 2) >> 2);
            return;
        }
        if (hs == 2 && vs == 1)
        {
            const uint8_t* in = nearRow;
            if (w == 1) { out[0] = out[1] = in[0]; return; }
            out[0] = in[0];
            out[1] = uint8_t((in[0] * 3 + in[1] + 2) >> 2);
            int i = 1;
            for (; i < w - 1; ++i)
            {
                const int n = 3 * in[i] + 2;
                out[i * 2] = uint8_t((n + in[i - 1]) >> 2);
                out[i * 2 + 1] = uint8_t((n + in[i + 1]) >> 2);
            }
            out[i * 2] = uint8_t((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
            out[i * 2 + 1] = in[w - 1];
            return;
        }
        if (hs == 2 && vs == 2)
        {
            if (w == 1) { out[0] = out[1] = uint8_t((3 * nearRow[0] + farRow[0] + 2) >> 2); return; }
            int t1 = 3 * nearRow[0] + farRow[0];
            out[0] = uint8_t((t1 + 2) >> 2);
            for (int i = 1; i < w; ++i)
            {
                const int t0 = t1;
                t1 = 3 * nearRow[i] + farRow[i];
                out[i * 2 - 1] = uint8_t((3 * t0 + t1 + 8) >> 4);
                out[i * 2] = uint8_t((3 * t1 + t0 + 8) >> 4);
            }
            out[w * 2 - 1] = uint8_t((t1 + 2) >> 2);
            return;
        }
        for (int i = 0; i < w; ++i)
            for (int j = 0; j < hs; ++j) out[i * hs + j] = nearRow[i];
    }

    void output(std::vector<uint8_t>& pixels, int& channels)
    {
        channels = numComp == 3 ? 3 : 1;
        pixels.assign(size_t(width) * height * channels, 0);
        const bool isRgb = numComp == 3 && ((comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') ||
                                            (adobeTransform == 0 && !jfif));
        struct Walk { int hs, vs, ystep, wLores, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; } walk[3];
        for (int k = 0; k < numComp; ++k)
        {
            Walk& r = walk[k];
            r.hs = hMax / comp[k].h;
            r.vs = vMax / comp[k].v;
            r.ystep = r.vs >> 1;
            r.wLores = (width + r.hs - 1) / r.hs;
            r.ypos = 0;
            r.line0 = r.line1 = comp[k].data.data();
            r.buf.assign(size_t(width) + 8, 0);
        }
        constexpr auto fixed = [](float v) { return int(v * 4096.0f + 0.5f) << 8; };
        for (int j = 0; j < height; ++j)
        {
            const uint8_t* row[3] = {nullptr, nullptr, nullptr};
            for (int k = 0; k < numComp; ++k)
            {
                Walk& r = walk[k];
                const bool bottom = r.ystep >= (r.vs >> 1);
                resampleRow(r.buf.data(), bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.wLores, r.hs, r.vs);
                row[k] = r.buf.data();
                if (++r.ystep >= r.vs)
                {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < comp[k].y) r.line1 += comp[k].w2;
                }
            }
            uint8_t* out = &pixels[size_t(j) * width * channels];
            if (numComp == 1) std::memcpy(out, row[0], size_t(width));
            else if (isRgb)
                for (int i = 0; i < width; ++i, out += 3) { out[0] = row[0][i]; out[1] = row[1][i]; out[2] = row[2][i]; }
            else
                for (int i = 0; i < width; ++i, out += 3)
                {
                    const int yFixed = (row[0][i] << 20) + (1 << 19);
                    const int cr = row[2][i] - 128, cb = row[1][i] - 128;
                    int r = yFixed + cr * fixed(1.40200f);
                    int g = yFixed + cr * -fixed(0.71414f) + int(unsigned(cb * -fixed(0.34414f)) & 0xffff0000u);
                    int b = yFixed + cb * fixed(1.77200f);
                    out[0] = clamp(r >> 20);
                    out[1] = clamp(g >> 20);
                    out[2] = clamp(b >> 20);
                }
        }
    }
};

}  // namespace

bool IsJpeg(const std::vector<uint8_t>& file) { return file.size() >= 3 && file[0] == 0xff && file[1] == 0xd8 && file[2] == 0xff; }

bool JpegInfo(const std::vector<uint8_t>& file, int& width, int& height, int& channels, std::string& err)
{
    Decoder d(file, err);
    if (!d.run(true)) return false;
    width = d.width;
    height = d.height;
    channels = d.numComp == 3 ? 3 : 1;
    return true;
}

bool JpegDecode(const std::vector<uint8_t>& file, int& width, int& height, int& channels, std::vector<uint8_t>& pixels,
                std::string& err)
{
    Decoder d(file, err);
    if (!d.run(false)) return false;
    width = d.width;
    height = d.height;
    d.output(pixels, channels);
    return true;
}

}  // namespace formats
}  // namespace srx
