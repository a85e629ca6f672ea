/*
 * vgg_oracle.c -- CPU restatement of the VGG-16 fc7 extractor.  TEST INFRASTRUCTURE ONLY
 * (see nvqa_oracle.c).  PARITY UNPINNED: the Caffe model and Torch7 are not available; this is
 * the public 16-layer VGG definition (13 conv3x3 pad 1 + ReLU, 5 max-pool 2x2/2, fc6+ReLU,
 * fc7+ReLU) that 002_train_vqa_arch1/001_prepro_img_vgg.lua:36-37,109-110 loads and taps at
 * module 38 (post-ReLU fc7; Dropout is the identity in evaluate mode), plus loadim (:47-71).
 * Direct convolution in NCHW with Caffe OIHW weights -- deliberately not the implicit-GEMM /
 * NHWC formulation of the HIP path.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const int kCout[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
static const int kPool[13] = {0, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1};

static int imax(int a, int b) { return a > b ? a : b; }

/* bf16 operand mode (nvqa_vgg16_set_precision): both operands of every product rounded to bf16, round-to-nearest-even */
static int g_vgg_bf16 = 0;
void oracle_vgg16_set_precision(int bf16) { g_vgg_bf16 = bf16; }
static float rb(float v)
{
    if (!g_vgg_bf16) return v;
    uint32_t u;
    memcpy(&u, &v, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&v, &u, 4);
    return v;
}

size_t oracle_vgg16_weight_count(int width_div, int hw)
{
    size_t off = 0;
    int cin = 3;
    for (int i = 0; i < 13; ++i) {
        const int co = imax(1, kCout[i] / width_div);
        off += (size_t)co * cin * 9 + co;
        cin = co;
    }
    const int S = hw / 32, F = imax(4, 4096 / width_div);
    off += (size_t)F * cin * S * S + F;
    off += (size_t)F * F + F;
    return off;
}

/* images n x 3 x hw x hw (preprocessed) -> out n x F */
int oracle_vgg16_fc7(int width_div, int hw, const float *flat, const float *images, int n, float *out)
{
    int cin = 3, H = hw, W = hw;
    const float *p = flat;
    float *cur = (float *)malloc(sizeof(float) * (size_t)n * 3 * hw * hw);
    memcpy(cur, images, sizeof(float) * (size_t)n * 3 * hw * hw);
    for (int i = 0; i < 13; ++i) {
        const int co = imax(1, kCout[i] / width_div);
        const float *Wt = p, *b = p + (size_t)co * cin * 9;
        p = b + co;
        float *nxt = (float *)malloc(sizeof(float) * (size_t)n * co * H * W);
#pragma omp parallel for collapse(2) schedule(static)
        for (int im = 0; im < n; ++im)
            for (int o = 0; o < co; ++o) {
                float *dst = nxt + ((size_t)im * co + o) * H * W;
                for (int q = 0; q < H * W; ++q) dst[q] = b[o];
                for (int c = 0; c < cin; ++c) {
                    const float *src = cur + ((size_t)im * cin + c) * H * W;
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const float w = rb(Wt[(((size_t)o * cin + c) * 3 + ky) * 3 + kx]);
                            for (int y = 0; y < H; ++y) {
                                const int iy = y + ky - 1;
                                if (iy < 0 || iy >= H) continue;
                                const int x0 = kx == 0 ? 1 : 0, x1 = kx == 2 ? W - 1 : W;
                                for (int x = x0; x < x1; ++x) dst[y * W + x] += w * rb(src[iy * W + x + kx - 1]);
                            }
                        }
                }
                for (int q = 0; q < H * W; ++q) dst[q] = dst[q] > 0 ? dst[q] : 0; /* ReLU */
            }
        free(cur);
        cur = nxt;
        cin = co;
        if (kPool[i]) {
            const int Ho = H / 2, Wo = W / 2;
            float *pl = (float *)malloc(sizeof(float) * (size_t)n * cin * Ho * Wo);
            for (size_t ic = 0; ic < (size_t)n * cin; ++ic)
                for (int y = 0; y < Ho; ++y)
                    for (int x = 0; x < Wo; ++x) {
                        const float *s = cur + ic * H * W + (size_t)(2 * y) * W + 2 * x;
                        float m = s[0];
                        if (s[1] > m) m = s[1];
                        if (s[W] > m) m = s[W];
                        if (s[W + 1] > m) m = s[W + 1];
                        pl[ic * Ho * Wo + (size_t)y * Wo + x] = m;
                    }
            free(cur);
            cur = pl;
            H = Ho; W = Wo;
        }
    }
    const int F = imax(4, 4096 / width_div), K6 = cin * H * W;
    float *f6 = (float *)malloc(sizeof(float) * (size_t)n * F);
    const float *W6 = p, *b6 = p + (size_t)F * K6;
    p = b6 + F;
    const float *W7 = p, *b7 = p + (size_t)F * F;
#pragma omp parallel for collapse(2) schedule(static)
    for (int im = 0; im < n; ++im)
        for (int f = 0; f < F; ++f) {
            double acc = b6[f];
            const float *x = cur + (size_t)im * K6, *w = W6 + (size_t)f * K6; /* CHW flatten (nn.View) */
            for (int k = 0; k < K6; ++k) acc += (double)rb(x[k]) * rb(w[k]);
            f6[(size_t)im * F + f] = acc > 0 ? (float)acc : 0.f;
        }
#pragma omp parallel for collapse(2) schedule(static)
    for (int im = 0; im < n; ++im)
        for (int f = 0; f < F; ++f) {
            double acc = b7[f];
            const float *x = f6 + (size_t)im * F, *w = W7 + (size_t)f * F;
            for (int k = 0; k < F; ++k) acc += (double)rb(x[k]) * rb(w[k]);
            out[(size_t)im * F + f] = acc > 0 ? (float)acc : 0.f;
        }
    free(cur);
    free(f6);
    return 0;
}

/* image.scale(src, width, height) in its default 'bilinear' mode (001_prepro_img_vgg.lua:50).  The `image` rock is NOT in
 * /root/reference (third-party, unpinned: SURVEY.md section 2 row 14); this restates the algorithm of its published source
 * (torch/image, generic/image.c: image_(Main_scaleBilinear) over image_(Main_scaleLinear_rowcol)), recalled, not
 * compiled -- PARITY UNPINNED, no reference fixture covers it.  The scale is SEPARABLE, rows (width) first into a
 * [src_height x dst_width] temporary, then columns, and each 1-D pass has two branches:
 *   dst_len > src_len  linear interpolation with step (src_len - 1) / (dst_len - 1), the last sample copied;
 *   dst_len < src_len  AREA AVERAGE: output i is the mean of the source interval [i s, (i + 1) s), s = src_len / dst_len,
 *                      the two end samples weighted by their covered fraction (every real VQA image: COCO 640 x 480 -> 224);
 *   equal              copy.
 * All arithmetic in float, in the source's order (FP contraction off: the HIP kernel follows the same order with
 * unfused operations, so the two agree to the last bit). */
#pragma STDC FP_CONTRACT OFF
static void scale_rowcol(const float *src, float *dst, long src_stride, long dst_stride, long src_len, long dst_len)
{
    if (dst_len > src_len) {
        const float scale = (float)(src_len - 1) / (float)(dst_len - 1);
        if (src_len == 1) {
            for (long di = 0; di < dst_len - 1; ++di) dst[di * dst_stride] = src[0];
        } else {
            for (long di = 0; di < dst_len - 1; ++di) {
                float si_f = (float)di * scale;
                const long si_i = (long)si_f;
                si_f -= (float)si_i;
                const float a = (1.0f - si_f) * src[si_i * src_stride], b = si_f * src[(si_i + 1) * src_stride];
                dst[di * dst_stride] = a + b;
            }
        }
        dst[(dst_len - 1) * dst_stride] = src[(src_len - 1) * src_stride];
    } else if (dst_len < src_len) {
        long si0_i = 0;
        float si0_f = 0.f;
        const float scale = (float)src_len / (float)dst_len;
        for (long di = 0; di < dst_len; ++di) {
            float si1_f = (float)(di + 1) * scale;
            const long si1_i = (long)si1_f;
            si1_f -= (float)si1_i;
            float acc = (1.0f - si0_f) * src[si0_i * src_stride];
            float n = 1.0f - si0_f;
            for (long si = si0_i + 1; si < si1_i; ++si) {
                acc = acc + src[si * src_stride];
                n = n + 1.0f;
            }
            if (si1_i < src_len) {
                const float t = si1_f * src[si1_i * src_stride];
                acc = acc + t;
                n = n + si1_f;
            }
            dst[di * dst_stride] = acc / n;
            si0_i = si1_i;
            si0_f = si1_f;
        }
    } else {
        for (long i = 0; i < dst_len; ++i) dst[i * dst_stride] = src[i * src_stride];
    }
}

/* one plane [H x W] -> [S x S] */
void oracle_image_scale_plane(const float *src, int H, int W, int S, float *dst)
{
    float *tmp = (float *)malloc(sizeof(float) * (size_t)H * S);
    for (int j = 0; j < H; ++j) scale_rowcol(src + (size_t)j * W, tmp + (size_t)j * S, 1, 1, W, S); /* compress / expand rows first */
    for (int i = 0; i < S; ++i) scale_rowcol(tmp + i, dst + i, S, S, H, S);                          /* then columns */
    free(tmp);
}

/* loadim (001_prepro_img_vgg.lua:47-71) minus the file decode: rgb n x 3 x H x W in [0,1] -> image.scale to S x S ->
 * x255 -> n x 3 x S x S BGR planes, mean-subtracted (:65-69). */
void oracle_vgg16_preprocess(const float *rgb, int n, int H, int W, int S, float *out)
{
    static const float mean[3] = {103.939f, 116.779f, 123.68f};
    float *pl = (float *)malloc(sizeof(float) * (size_t)S * S);
    for (int im = 0; im < n; ++im)
        for (int c = 0; c < 3; ++c) {
            oracle_image_scale_plane(rgb + ((size_t)im * 3 + (2 - c)) * H * W, H, W, S, pl);
            float *o = out + ((size_t)im * 3 + c) * S * S;
            for (int i = 0; i < S * S; ++i) {
                const float v = pl[i] * 255.0f;
                o[i] = v - mean[c];
            }
        }
    free(pl);
}
