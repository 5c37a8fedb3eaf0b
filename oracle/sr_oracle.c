/*
 * sr_oracle.c -- CPU restatement of the StarsX/SuperResolution inference path.
 *
 * TEST INFRASTRUCTURE ONLY: see sr_oracle.h for what may use this file and for
 * the pinning status ("parity unpinned" against DirectML; weight pre-processing
 * pinned against the reference's Float16Compressor.h).
 *
 * Plain C, fp32 accumulate in (c, i, j) order per output element, optional
 * OpenMP over output rows (each output element is still one sequential sum, so
 * the result does not depend on the thread count).
 */
#include "sr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* scalar conversions                                                       */
/* ------------------------------------------------------------------------ */
static uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Float16Compressor::compress (Common/Float16Compressor.h:41-57), restated by
 * what it does rather than by its bit tricks: drop the sign, then
 *   NaN            -> payload >> 13, but at least 0x7C01
 *   > 65504 or inf -> 0x7C00
 *   < 2^-14        -> floor(|x| * 2^24)          (half subnormal, truncated)
 *   otherwise      -> exponent rebased, mantissa TRUNCATED to 10 bits
 * and put the sign back. */
uint16_t sro_f16_compress(float v)
{
    const uint32_t u = f32_bits(v);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7fffffffu;
    uint32_t h;
    if (a > 0x7f800000u) {                       /* NaN */
        h = (a < 0x7f802000u) ? 0x7c01u : (0x7c00u | ((a >> 13) & 0x3ffu));
    } else if (a > 0x477fe000u) {                /* above the largest half, or inf */
        h = 0x7c00u;
    } else if (a < 0x38800000u) {                /* below the smallest normal half */
        const float scaled = bits_f32(a) * 137438953472.0f;      /* 2^37, exact */
        h = ((uint32_t)(int32_t)scaled) >> 13;
    } else {
        h = (a >> 13) - 0x1c000u;
    }
    return (uint16_t)(h | sign);
}

uint16_t sro_f32_to_f16_rne(float v)
{
    const uint32_t u = f32_bits(v);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u | ((a >> 13) & 0x3ffu));
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);     /* rounds to inf */
    if (a < 0x33000001u) return sign;                            /* <= 2^-25 -> 0 */
    if (a < 0x38800000u) {                                       /* subnormal half */
        /* value = m * 2^(e-150); half subnormal unit is 2^-24 */
        const int e = (int)(a >> 23);
        const uint32_t m = (a & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;                               /* 14..24 */
        const uint32_t q = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1);
        uint32_t r = q;
        if (rem > half || (rem == half && (q & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    {
        uint32_t r = (a >> 13) - 0x1c000u;
        const uint32_t rem = a & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;  /* carries into exponent correctly */
        return (uint16_t)(sign | r);
    }
}

float sro_f16_to_f32(uint16_t h)
{
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu;
    const uint32_t m = h & 0x3ffu;
    if (e == 0) {
        const float f = (float)m * 5.9604644775390625e-8f;       /* m * 2^-24 */
        return bits_f32(f32_bits(f) | sign);
    }
    if (e == 31) return bits_f32(sign | 0x7f800000u | (m << 13));
    return bits_f32(sign | ((e + 112u) << 23) | (m << 13));
}

void sro_f16_compress_array(const float* in, uint16_t* out, size_t n)
{
    for (size_t i = 0; i < n; ++i) out[i] = sro_f16_compress(in[i]);
}

static float round_f16(float v) { return sro_f16_to_f32(sro_f32_to_f16_rne(v)); }

/* ------------------------------------------------------------------------ */
/* weights.bin (LoadWeights.cpp:21-93)                                      */
/* ------------------------------------------------------------------------ */
typedef struct {
    char name[256];
    uint32_t len;
    float* data;
} sro_entry;

struct sro_weights {
    int count;
    sro_entry* e;
};

void sro_free_weights(sro_weights* w)
{
    if (!w) return;
    for (int i = 0; i < w->count; ++i) free(w->e[i].data);
    free(w->e);
    free(w);
}

int sro_load_weights(const char* path, sro_weights** out, int verbose)
{
    *out = NULL;
    FILE* f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "Unable to open weight file: %s\n", path);
        return 1;
    }
    int32_t count = 0;
    if (fread(&count, 4, 1, f) != 1 || count < 0) {
        fprintf(stderr, "Invalid weight map file: %s\n", path);
        fclose(f);
        return 2;
    }
    if (verbose) printf("Number of weight tensors: %d\n", count);
    sro_weights* w = (sro_weights*)calloc(1, sizeof(*w));
    w->e = (sro_entry*)calloc((size_t)count + 1, sizeof(sro_entry));
    for (int i = 0; i < count; ++i) {
        uint32_t name_len = 0, w_len = 0;
        if (fread(&name_len, 4, 1, f) != 1) goto bad_data;
        if (name_len > 255) {
            fprintf(stderr, "name_len exceeds c_bufferLength: %u vs 255\n", name_len);
            goto fail;
        }
        char name[256];
        if (name_len && fread(name, 1, name_len, f) != name_len) goto bad_data;
        name[name_len] = '\0';
        if (fread(&w_len, 4, 1, f) != 1) goto bad_data;
        float* data = (float*)malloc(sizeof(float) * (w_len ? w_len : 1));
        if (!data) goto bad_data;
        if (w_len && fread(data, sizeof(float), w_len, f) != w_len) {
            free(data);
            goto bad_data;
        }
        /* the reference keeps a std::map: a repeated name overwrites */
        int slot = -1;
        for (int k = 0; k < w->count; ++k)
            if (strcmp(w->e[k].name, name) == 0) slot = k;
        if (slot < 0) slot = w->count++;
        else free(w->e[slot].data);
        strcpy(w->e[slot].name, name);
        w->e[slot].len = w_len;
        w->e[slot].data = data;
        if (verbose) printf("Loaded tensor: %s -> %u\n", name, w_len);
    }
    fclose(f);
    *out = w;
    return 0;
bad_data:
    fprintf(stderr, "Invalid tensor data\n");
fail:
    fclose(f);
    sro_free_weights(w);
    return 3;
}

int sro_weights_count(const sro_weights* w) { return w->count; }
const char* sro_weights_name(const sro_weights* w, int i) { return w->e[i].name; }
uint32_t sro_weights_len(const sro_weights* w, int i) { return w->e[i].len; }
const float* sro_weights_data(const sro_weights* w, int i) { return w->e[i].data; }

const float* sro_weights_find(const sro_weights* w, const char* name, uint32_t* len)
{
    for (int i = 0; i < w->count; ++i)
        if (strcmp(w->e[i].name, name) == 0) {
            if (len) *len = w->e[i].len;
            return w->e[i].data;
        }
    return NULL;
}

/* ------------------------------------------------------------------------ */
/* BN fold + conversion (Util::CreateWeightTensors) and GetStrides          */
/* ------------------------------------------------------------------------ */
void sro_get_strides(const uint32_t s[4], int layout, uint32_t st[4])
{
    const uint32_t C = s[1], H = s[2], W = s[3];
    if (layout == 1) { st[0] = C * H * W; st[1] = 1; st[2] = W * C; st[3] = C; }
    else             { st[0] = C * H * W; st[1] = H * W; st[2] = W; st[3] = 1; }
}

int sro_create_weight_tensors(const sro_weights* w, const char* conv, const char* scale,
                              const char* shift, const uint32_t sizes[4], int layout, int dtype,
                              void* filter_out, void* bias_out)
{
    const uint32_t N = sizes[0], C = sizes[1], H = sizes[2], W = sizes[3];
    uint32_t flen = 0, slen = 0, blen = 0;
    const float* filt = sro_weights_find(w, conv, &flen);
    if (!filt || flen != N * C * H * W) return -1;
    const float* sc = NULL;
    const float* sh = NULL;
    if (scale) {
        sc = sro_weights_find(w, scale, &slen);
        sh = shift ? sro_weights_find(w, shift, &blen) : NULL;
        if (!sc || !sh || slen != N || blen != N) return -2;
    }
    uint16_t* f16 = (uint16_t*)filter_out;
    float* f32 = (float*)filter_out;
    size_t o = 0;
    for (uint32_t n = 0; n < N; ++n) {
        const float s = sc ? sc[n] : 1.0f;
        if (layout == 1) {
            for (uint32_t h = 0; h < H; ++h)
                for (uint32_t x = 0; x < W; ++x)
                    for (uint32_t c = 0; c < C; ++c, ++o) {
                        const float v0 = filt[((n * C + c) * H + h) * W + x];
                        const volatile float v = sc ? s * v0 : v0;   /* one fp32 multiply */
                        if (dtype == 0) f16[o] = sro_f16_compress(v); else f32[o] = v;
                    }
        } else {
            for (uint32_t i = 0; i < C * H * W; ++i, ++o) {
                const float v0 = filt[n * C * H * W + i];
                const volatile float v = sc ? s * v0 : v0;
                if (dtype == 0) f16[o] = sro_f16_compress(v); else f32[o] = v;
            }
        }
    }
    if (!sc) return 0;
    for (uint32_t n = 0; n < N; ++n) {
        if (dtype == 0) ((uint16_t*)bias_out)[n] = sro_f16_compress(sh[n]);
        else ((float*)bias_out)[n] = sh[n];
    }
    return 1;
}

/* ------------------------------------------------------------------------ */
/* pack / unpack                                                            */
/* ------------------------------------------------------------------------ */
/* CSImageToTensor.hlsl:24-43 with the plane size H*W (the shader's W*W at :37
 * is only right for square images).  UNORM8 -> float is u/255, the typed-UAV
 * store rounds to fp16 (RNE) in dtype 0. */
void sro_image_to_tensor(const uint8_t* rgba, int w, int h, size_t pitch, int dtype, float* chw)
{
    const size_t plane = (size_t)w * h;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const uint8_t* p = rgba + (size_t)y * pitch + (size_t)x * 4;
            for (int c = 0; c < 3; ++c) {
                float v = (float)p[c] / 255.0f;
                if (dtype == 0) v = round_f16(v);
                chw[c * plane + (size_t)y * w + x] = v;
            }
        }
}

/* PSTensorToImage.hlsl:14-35 writes float4(rgb, 1) to an R8G8B8A8_UNORM target:
 * NaN -> 0, clamp to [0,1], scale by 255, add 0.5, drop the fraction. */
static uint8_t unorm8(float v)
{
    if (!(v == v)) return 0;
    if (v < 0.0f) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return (uint8_t)(v * 255.0f + 0.5f);
}

void sro_tensor_to_image(const float* chw, int w, int h, uint8_t* rgba)
{
    const size_t plane = (size_t)w * h;
    for (size_t i = 0; i < plane; ++i) {
        rgba[i * 4 + 0] = unorm8(chw[i]);
        rgba[i * 4 + 1] = unorm8(chw[plane + i]);
        rgba[i * 4 + 2] = unorm8(chw[2 * plane + i]);
        rgba[i * 4 + 3] = 255;
    }
}

/* ------------------------------------------------------------------------ */
/* operators                                                                */
/* ------------------------------------------------------------------------ */
/* DML CONVOLUTION as Util::CreateConvolutionLayer builds it (call sites
 * SuperResolution.cpp:191-268): cross-correlation, stride 1, zero padding
 * ceil((k-1)/2) before / floor((k-1)/2) after, optional bias then ReLU.
 * in: [C,H,W], wt: [O,C,k,k], out: [O,H,W]. */
static void conv2d(const float* in, int C, int H, int W, const float* wt, const float* bias,
                   int O, int k, int relu, int round16, float* out)
{
    const int pb = k / 2;            /* ceil((k-1)/2) for the odd k used here; even k: k/2 */
    const size_t plane = (size_t)H * W;
#pragma omp parallel
    {
        float* acc = (float*)malloc(sizeof(float) * (size_t)W);
#pragma omp for collapse(2) schedule(static)
        for (int o = 0; o < O; ++o)
            for (int y = 0; y < H; ++y) {
                const float b = bias ? bias[o] : 0.0f;
                for (int x = 0; x < W; ++x) acc[x] = b;
                for (int c = 0; c < C; ++c)
                    for (int i = 0; i < k; ++i) {
                        const int yy = y + i - pb;
                        if (yy < 0 || yy >= H) continue;
                        const float* row = in + c * plane + (size_t)yy * W;
                        for (int j = 0; j < k; ++j) {
                            const float wv = wt[((o * C + c) * k + i) * k + j];
                            const int dx = j - pb;
                            const int x0 = dx < 0 ? -dx : 0;
                            const int x1 = dx > 0 ? W - dx : W;
                            for (int x = x0; x < x1; ++x) acc[x] += row[x + dx] * wv;
                        }
                    }
                float* dst = out + o * plane + (size_t)y * W;
                for (int x = 0; x < W; ++x) {
                    float v = acc[x];
                    if (relu && !(v > 0.0f)) v = (v == v) ? 0.0f : v;
                    dst[x] = round16 ? round_f16(v) : v;
                }
            }
        free(acc);
    }
}

/* DML UPSAMPLE_2D, nearest neighbour, scale 2x2 (Util::CreateUpsampleLayer,
 * call sites SuperResolution.cpp:180, :225). */
static void upsample2x(const float* in, int C, int H, int W, float* out)
{
    const int H2 = 2 * H, W2 = 2 * W;
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < H2; ++y) {
            const float* src = in + ((size_t)c * H + (y >> 1)) * W;
            float* dst = out + ((size_t)c * H2 + y) * W2;
            for (int x = 0; x < W2; ++x) dst[x] = src[x >> 1];
        }
}

/* Four-phase form of "5x5 conv on the NN-upsampled tensor": output (2y+py, 2x+px)
 * is a 3x3 conv of the low-resolution tensor with the 5x5 taps that fall on the
 * same low-resolution pixel added together.  For phase p the taps i = 0..4 land
 * on low-resolution offset floor((p + i - 2) / 2) in {-1, 0, +1}. */
static int phase_tap(int p, int i) { return ((p + i - 2 + 4) >> 1) - 2 + 1; }   /* 0..2 */

static void conv_up1_phases(const float* in, int C, int H, int W, const float* wt /*[O,C,5,5]*/,
                            const float* bias, int O, int round16, float* out /*[O,2H,2W]*/)
{
    const size_t plane = (size_t)H * W;
    const int H2 = 2 * H, W2 = 2 * W;
    float* pw = (float*)malloc(sizeof(float) * (size_t)O * C * 9);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)O * plane);
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            for (int o = 0; o < O; ++o)
                for (int c = 0; c < C; ++c) {
                    double s[9] = {0};
                    for (int i = 0; i < 5; ++i)
                        for (int j = 0; j < 5; ++j)
                            s[phase_tap(py, i) * 3 + phase_tap(px, j)] +=
                                (double)wt[((o * C + c) * 5 + i) * 5 + j];
                    for (int t = 0; t < 9; ++t) {
                        const float v = (float)s[t];
                        pw[(o * C + c) * 9 + t] = round16 ? round_f16(v) : v;
                    }
                }
            conv2d(in, C, H, W, pw, bias, O, 3, 1, round16, tmp);
            for (int o = 0; o < O; ++o)
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < W; ++x)
                        out[((size_t)o * H2 + 2 * y + py) * W2 + 2 * x + px] =
                            tmp[o * plane + (size_t)y * W + x];
        }
    free(pw);
    free(tmp);
}

/* Row-shared form of the same operator: only the COLUMNS of the 5x5 filter are folded per output phase px
 * (three column taps, each the sum of the 5x5 taps that read the same low-resolution column, rounded once to
 * fp16); the five ROW taps stay as they are and read low-resolution row y + floor((py + i - 2) / 2).  The two
 * output rows 2y and 2y + 1 then share the terms of row taps 0, 2, 4 and differ in where taps 1 and 3 read,
 * which is how the fused GPU kernel evaluates conv_up1 (30 instead of 36 products per input pixel). */
static void conv_up1_rowshared(const float* in, int C, int H, int W, const float* wt /*[O,C,5,5]*/,
                               const float* bias, int O, int round16, float* out /*[O,2H,2W]*/)
{
    const size_t plane = (size_t)H * W;
    const int H2 = 2 * H, W2 = 2 * W;
    float* pw = (float*)malloc(sizeof(float) * (size_t)O * C * 15);
    for (int px = 0; px < 2; ++px) {
        for (int o = 0; o < O; ++o)
            for (int c = 0; c < C; ++c)
                for (int i = 0; i < 5; ++i) {
                    double s[3] = {0};
                    for (int j = 0; j < 5; ++j) s[phase_tap(px, j)] += (double)wt[((o * C + c) * 5 + i) * 5 + j];
                    for (int t = 0; t < 3; ++t) {
                        const float v = (float)s[t];
                        pw[((o * C + c) * 5 + i) * 3 + t] = round16 ? round_f16(v) : v;
                    }
                }
#pragma omp parallel
        {
            float* acc = (float*)malloc(sizeof(float) * (size_t)W);
#pragma omp for collapse(2) schedule(static)
            for (int o = 0; o < O; ++o)
                for (int y2 = 0; y2 < H2; ++y2) {
                    const int y = y2 >> 1, py = y2 & 1;
                    for (int x = 0; x < W; ++x) acc[x] = bias ? bias[o] : 0.0f;
                    for (int c = 0; c < C; ++c)
                        for (int i = 0; i < 5; ++i) {
                            const int yy = y + phase_tap(py, i) - 1;
                            if (yy < 0 || yy >= H) continue;
                            const float* row = in + c * plane + (size_t)yy * W;
                            for (int t = 0; t < 3; ++t) {
                                const float wv = pw[((o * C + c) * 5 + i) * 3 + t];
                                const int dx = t - 1;
                                const int x0 = dx < 0 ? -dx : 0;
                                const int x1 = dx > 0 ? W - dx : W;
                                for (int x = x0; x < x1; ++x) acc[x] += row[x + dx] * wv;
                            }
                        }
                    float* dst = out + ((size_t)o * H2 + y2) * W2 + px;
                    for (int x = 0; x < W; ++x) {
                        float v = acc[x];
                        if (!(v > 0.0f)) v = (v == v) ? 0.0f : v;
                        dst[2 * x] = round16 ? round_f16(v) : v;
                    }
                }
            free(acc);
        }
    }
    free(pw);
}

/* ------------------------------------------------------------------------ */
/* the model                                                                */
/* ------------------------------------------------------------------------ */
typedef struct {
    int O, C, k, relu;
    float* w;       /* [O,C,k,k] as float (fp16-exact in dtype 0) */
    float* b;       /* [O] or NULL */
} layer;

struct sro_model {
    int dtype, up1_mode;
    layer L[7];
    float* t[SRO_T_COUNT];
    int tc[SRO_T_COUNT], th[SRO_T_COUNT], tw[SRO_T_COUNT];
};

/* layer table: SuperResolution.cpp:188-268 */
static const struct { const char* conv; const char* scale; const char* shift; int O, C, k; } kLayers[7] = {
    {"conv1/weights", "conv1/BatchNorm/scale", "conv1/BatchNorm/shift", 32, 3, 5},
    {"conv2/weights", "conv2/BatchNorm/scale", "conv2/BatchNorm/shift", 64, 32, 3},
    {"conv3/weights", "conv3/BatchNorm/scale", "conv3/BatchNorm/shift", 64, 64, 3},
    {"conv_up1/conv/weights", "conv_up1/conv/BatchNorm/scale", "conv_up1/conv/BatchNorm/shift", 32, 64, 5},
    {"conv4/weights", "conv4/BatchNorm/scale", "conv4/BatchNorm/shift", 32, 32, 3},
    {"conv5/weights", "conv5/BatchNorm/scale", "conv5/BatchNorm/shift", 32, 32, 3},
    {"conv6/weights", NULL, NULL, 3, 32, 3},
};

sro_model* sro_model_create(const sro_weights* w, int dtype, int up1_mode)
{
    sro_model* m = (sro_model*)calloc(1, sizeof(*m));
    m->dtype = dtype;
    m->up1_mode = up1_mode;
    for (int l = 0; l < 7; ++l) {
        layer* L = &m->L[l];
        L->O = kLayers[l].O; L->C = kLayers[l].C; L->k = kLayers[l].k;
        L->relu = kLayers[l].scale != NULL;
        const uint32_t sizes[4] = {(uint32_t)L->O, (uint32_t)L->C, (uint32_t)L->k, (uint32_t)L->k};
        const size_t n = (size_t)L->O * L->C * L->k * L->k;
        L->w = (float*)malloc(sizeof(float) * n);
        L->b = (float*)malloc(sizeof(float) * (size_t)L->O);
        int rc;
        if (dtype == 0) {
            uint16_t* f = (uint16_t*)malloc(2 * n);
            uint16_t* b = (uint16_t*)malloc(2 * (size_t)L->O);
            rc = sro_create_weight_tensors(w, kLayers[l].conv, kLayers[l].scale, kLayers[l].shift,
                                           sizes, 0, 0, f, b);
            if (rc >= 0) {
                for (size_t i = 0; i < n; ++i) L->w[i] = sro_f16_to_f32(f[i]);
                if (rc == 1) for (int i = 0; i < L->O; ++i) L->b[i] = sro_f16_to_f32(b[i]);
            }
            free(f); free(b);
        } else {
            rc = sro_create_weight_tensors(w, kLayers[l].conv, kLayers[l].scale, kLayers[l].shift,
                                           sizes, 0, 1, L->w, L->b);
        }
        if (rc < 0) { sro_model_destroy(m); return NULL; }
        if (rc == 0) { free(L->b); L->b = NULL; }
    }
    return m;
}

void sro_model_destroy(sro_model* m)
{
    if (!m) return;
    for (int l = 0; l < 7; ++l) { free(m->L[l].w); free(m->L[l].b); }
    for (int i = 0; i < SRO_T_COUNT; ++i) free(m->t[i]);
    free(m);
}

static float* talloc(sro_model* m, int id, int c, int h, int w)
{
    free(m->t[id]);
    m->t[id] = (float*)malloc(sizeof(float) * (size_t)c * h * w);
    m->tc[id] = c; m->th[id] = h; m->tw[id] = w;
    return m->t[id];
}

static void tdrop(sro_model* m, int id, int keep)
{
    if (keep) return;
    free(m->t[id]);
    m->t[id] = NULL;
}

/* Dispatch order of SuperResolution::Process (SuperResolution.cpp:73-103):
 * up0, conv1..conv3, up1, conv_up1, conv4..conv6, add. */
int sro_run(sro_model* m, const uint8_t* rgba, int w, int h, size_t pitch,
            uint8_t* out_rgba, int keep)
{
    if (!m || w <= 0 || h <= 0) return 1;
    const int r16 = m->dtype == 0;
    const int H2 = 2 * h, W2 = 2 * w;
    float* x0 = talloc(m, SRO_T_INPUT, 3, h, w);
    sro_image_to_tensor(rgba, w, h, pitch, m->dtype, x0);

    float* y = talloc(m, SRO_T_OUTPUT, 3, H2, W2);
    upsample2x(x0, 3, h, w, y);

    const float* cur = x0;
    for (int l = 0; l < 3; ++l) {
        const layer* L = &m->L[l];
        float* o = talloc(m, SRO_T_CONV1 + l, L->O, h, w);
        conv2d(cur, L->C, h, w, L->w, L->b, L->O, L->k, L->relu, r16, o);
        if (l > 0) tdrop(m, SRO_T_CONV1 + l - 1, keep);
        cur = o;
    }
    {
        const layer* L = &m->L[3];
        float* o = talloc(m, SRO_T_CONV_UP1, L->O, H2, W2);
        if (m->up1_mode == 0) {
            float* up = (float*)malloc(sizeof(float) * (size_t)L->C * H2 * W2);
            upsample2x(cur, L->C, h, w, up);
            conv2d(up, L->C, H2, W2, L->w, L->b, L->O, L->k, L->relu, r16, o);
            free(up);
        } else if (m->up1_mode == 2) {
            conv_up1_rowshared(cur, L->C, h, w, L->w, L->b, L->O, r16, o);
        } else {
            conv_up1_phases(cur, L->C, h, w, L->w, L->b, L->O, r16, o);
        }
        tdrop(m, SRO_T_CONV3, keep);
        cur = o;
    }
    for (int l = 4; l < 7; ++l) {
        const layer* L = &m->L[l];
        float* o = talloc(m, SRO_T_CONV_UP1 + (l - 3), L->O, H2, W2);
        conv2d(cur, L->C, H2, W2, L->w, L->b, L->O, L->k, L->relu, r16, o);
        tdrop(m, SRO_T_CONV_UP1 + (l - 4), keep);
        cur = o;
    }
    /* ELEMENT_WISE_ADD in place on the up0 result (SuperResolution.cpp:700-702) */
    const size_t n = (size_t)3 * H2 * W2;
    for (size_t i = 0; i < n; ++i) {
        const float s = cur[i] + y[i];
        y[i] = r16 ? round_f16(s) : s;
    }
    tdrop(m, SRO_T_CONV6, keep);
    if (out_rgba) sro_tensor_to_image(y, W2, H2, out_rgba);
    return 0;
}

const float* sro_tensor(const sro_model* m, int id, int* c, int* h, int* w)
{
    if (id < 0 || id >= SRO_T_COUNT || !m->t[id]) return NULL;
    if (c) *c = m->tc[id];
    if (h) *h = m->th[id];
    if (w) *w = m->tw[id];
    return m->t[id];
}
