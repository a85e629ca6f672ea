// vgg.hip -- VGG-16 fc7 feature extractor (forward only) on gfx950.
//
// Replaces 002_train_vqa_arch1/001_prepro_img_vgg.lua: loadcaffe.load(...,'cudnn') + net:forward(ims)
// + net.modules[38].output (:36-37,109-110), i.e. 13 conv3x3/ReLU, 5 max-pools, fc6+ReLU, fc7+ReLU
// (module 38 = the Dropout after fc7's ReLU, identity in evaluate mode => post-ReLU fc7), and the
// per-image preprocessing of loadim (:47-71).
//
// Design: activations NHWC (channels padded to a multiple of 4), every convolution is an implicit
// GEMM on the fp32 MFMA template of gemm_f32.h (A_IM2COL: M = n*H*W output pixels, N = C_out,
// K = 9*C_in), bias+ReLU fused in the epilogue; fc6/fc7 are split-K GEMMs (M = batch is small, the
// 411 MB fc6 weight stream is the cost) with bias+ReLU fused into the slab reduction.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/nvqa.h"
#include "gemm_f32.h"
#include "nvqa_ctx.h"

using namespace nvqa;

namespace {

const int kConvCout[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
const int kPoolAfter[13] = {0, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1};

struct EpiBiasRelu {
    float *C;
    int ldc;
    const float *b;
    __device__ __forceinline__ void operator()(int, int m, int n, float v) const
    {
        C[(size_t)m * ldc + n] = fmaxf(v + b[n], 0.0f);
    }
};
// the same into a bf16 NHWC activation (gfx950 form of the bf16 mode: the next layer's operand is read as stored)
struct EpiBiasReluB16 {
    __bf16 *C;
    int ldc;
    const float *b;
    __device__ __forceinline__ void operator()(int, int m, int n, float v) const
    {
        C[(size_t)m * ldc + n] = (__bf16)fmaxf(v + b[n], 0.0f); // round to nearest even, as the on-the-fly conversion of BF = 1
    }
};
struct EpiSlab {
    float *C;
    int ldc;
    size_t slab;
    __device__ __forceinline__ void operator()(int z, int m, int n, float v) const { C[(size_t)z * slab + (size_t)m * ldc + n] = v; }
};

// NCHW [n][3][H][W] -> NHWC with 4 channels (4th = 0)
__global__ void k_nchw_to_nhwc4(const float *in, int n, int H, int W, float4 *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t hw = (size_t)H * W;
    if (i >= n * hw) return;
    const size_t img = i / hw, p = i % hw;
    const float *src = in + img * 3 * hw + p;
    out[i] = make_float4(src[0], src[hw], src[2 * hw], 0.f);
}

// 2x2 max pool, stride 2, NHWC, C % 4 == 0
__global__ void k_maxpool2_nhwc(const float4 *in, int n, int H, int W, int C4, float4 *out)
{
    const int Ho = H / 2, Wo = W / 2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n * Ho * Wo * C4;
    if (i >= total) return;
    const int c = i % C4;
    size_t p = i / C4;
    const int x = p % Wo; p /= Wo;
    const int y = p % Ho; const size_t img = p / Ho;
    const float4 *b = in + ((img * H + 2 * y) * W + 2 * x) * C4 + c;
    const float4 a0 = b[0], a1 = b[C4], a2 = b[(size_t)W * C4], a3 = b[(size_t)W * C4 + C4];
    float4 o;
    o.x = fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x));
    o.y = fmaxf(fmaxf(a0.y, a1.y), fmaxf(a2.y, a3.y));
    o.z = fmaxf(fmaxf(a0.z, a1.z), fmaxf(a2.z, a3.z));
    o.w = fmaxf(fmaxf(a0.w, a1.w), fmaxf(a2.w, a3.w));
    out[i] = o;
}

// the same on bf16 activations, 8 channels (16 bytes) per thread; the maximum of bf16 values is one of them: exact
__global__ void k_maxpool2_nhwc_b16(const uint4 *in, int n, int H, int W, int C8, uint4 *out)
{
    const int Ho = H / 2, Wo = W / 2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n * Ho * Wo * C8;
    if (i >= total) return;
    const int c = i % C8;
    size_t p = i / C8;
    const int x = p % Wo; p /= Wo;
    const int y = p % Ho; const size_t img = p / Ho;
    const uint4 *b = in + ((img * H + 2 * y) * W + 2 * x) * C8 + c;
    const uint4 a0 = b[0], a1 = b[C8], a2 = b[(size_t)W * C8], a3 = b[(size_t)W * C8 + C8];
    auto mx = [](unsigned p0, unsigned p1, unsigned p2, unsigned p3) {
        auto lo = [](unsigned w) { return __uint_as_float(w << 16); };
        auto hi = [](unsigned w) { return __uint_as_float(w & 0xffff0000u); };
        const float l = fmaxf(fmaxf(lo(p0), lo(p1)), fmaxf(lo(p2), lo(p3)));
        const float h = fmaxf(fmaxf(hi(p0), hi(p1)), fmaxf(hi(p2), hi(p3)));
        return (__float_as_uint(h) & 0xffff0000u) | (__float_as_uint(l) >> 16);
    };
    uint4 o;
    o.x = mx(a0.x, a1.x, a2.x, a3.x); o.y = mx(a0.y, a1.y, a2.y, a3.y);
    o.z = mx(a0.z, a1.z, a2.z, a3.z); o.w = mx(a0.w, a1.w, a2.w, a3.w);
    out[i] = o;
}

// f32 -> bf16 image of a weight array (round to nearest even)
__global__ void k_to_bf16(const float *in, size_t n, __bf16 *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (__bf16)in[i];
}

// sum of split-K slabs + bias + ReLU (fc6 / fc7)
__global__ void k_fc_finish(const float *slabs, int S, int M, int N, const float *bias, float *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * N) return;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += slabs[(size_t)z * M * N + i];
    out[i] = fmaxf(s + bias[i % N], 0.0f);
}

// One output sample of Torch's image.scale 1-D pass (torch/image generic/image.c: image_(Main_scaleLinear_rowcol); the
// `image` rock is third-party and absent from the reference checkout -- the algorithm is its published one, PARITY UNPINNED):
//   dst_len > src_len: linear interpolation with step (src_len - 1) / (dst_len - 1), last sample copied;
//   dst_len < src_len: area average over [di s, (di + 1) s), s = src_len / dst_len, end samples weighted by coverage;
//   equal: copy.
// `at(i)` returns source sample i.  Unfused float operations in the source's order (__f*_rn are never contracted).
// a product that must be ROUNDED before it is added (the source's float arithmetic is unfused): the empty asm hides it from
// the contraction of a * b + c into an fma, which neither __fmul_rn nor `#pragma clang fp contract(off)` prevented in this
// HIP version (measured: identity-size inputs came back 1 ulp off in half the pixels)
__device__ __forceinline__ float rounded(float x)
{
    asm volatile("" : "+v"(x));
    return x;
}
template <class F> __device__ __forceinline__ float scale1d(F at, int src_len, int dst_len, int di)
{
    if (dst_len > src_len) {
        if (di == dst_len - 1) return at(src_len - 1);
        if (src_len == 1) return at(0);
        const float scale = __fdiv_rn((float)(src_len - 1), (float)(dst_len - 1));
        float si_f = rounded(__fmul_rn((float)di, scale)); // (its integer part is subtracted next: no fma)
        const int si_i = (int)si_f;
        si_f = __fsub_rn(si_f, (float)si_i);
        return __fadd_rn(rounded(__fmul_rn(__fsub_rn(1.0f, si_f), at(si_i))), rounded(__fmul_rn(si_f, at(si_i + 1))));
    }
    if (dst_len < src_len) {
        const float scale = __fdiv_rn((float)src_len, (float)dst_len);
        float si0_f = rounded(__fmul_rn((float)di, scale)), si1_f = rounded(__fmul_rn((float)(di + 1), scale));
        const int si0_i = (int)si0_f, si1_i = (int)si1_f;
        si0_f = __fsub_rn(si0_f, (float)si0_i);
        si1_f = __fsub_rn(si1_f, (float)si1_i);
        float acc = rounded(__fmul_rn(__fsub_rn(1.0f, si0_f), at(si0_i))), n = __fsub_rn(1.0f, si0_f);
        for (int si = si0_i + 1; si < si1_i; ++si) {
            acc = __fadd_rn(acc, at(si));
            n = __fadd_rn(n, 1.0f);
        }
        if (si1_i < src_len) {
            acc = __fadd_rn(acc, rounded(__fmul_rn(si1_f, at(si1_i))));
            n = __fadd_rn(n, si1_f);
        }
        return __fdiv_rn(acc, n);
    }
    return at(di);
}

// loadim (001_prepro_img_vgg.lua:47-71): image.scale to S x S ignoring aspect (rows first, then columns: the vertical pass
// runs over horizontally scaled rows, recomputed here per output pixel in the same order), x255, RGB -> BGR, per-channel
// mean subtraction.  in [n][3][H][W] RGB in [0,1]; out [n][3][S][S] (BGR planes).
__global__ void k_vgg_preprocess(const float *in, int n, int H, int W, int S, float *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n * 3 * S * S;
    if (i >= total) return;
    const int x = i % S, y = (i / S) % S, c = (i / ((size_t)S * S)) % 3;
    const size_t img = i / ((size_t)3 * S * S);
    const int src_c = 2 - c; // output plane 0 = B = input plane 2
    const float mean = c == 0 ? 103.939f : (c == 1 ? 116.779f : 123.68f);
    const float *p = in + (img * 3 + src_c) * (size_t)H * W;
    auto hrow = [&](int r) { // sample x of source row r scaled to width S
        const float *row = p + (size_t)r * W;
        return scale1d([&](int k) { return row[k]; }, W, S, x);
    };
    const float v = scale1d(hrow, H, S, y);
    out[i] = __fsub_rn(rounded(__fmul_rn(v, 255.0f)), mean);
}

// C_out >= 128: 8 waves of 32 x 64, 16x16x4 MFMA (tools/kbench3: 117 vs 95 TF for the K-contiguous x K-contiguous form).
// K-tiles of 64 (round 3; 32 before): the per-tile cost outside the MFMAs -- loads, address selects, LDS writes, two barriers --
// is paid half as often; two workgroups of 64 KB still share a CU
// Round 3: SB (the fragment reads of the next q step pinned ahead of the MFMAs) + the A_IM2COLF loader; measured per layer
// at batch 64 (rocprofv3 kernel trace): 19.0 -> 17.9 ms for the 13 convolutions, 0.66 -> 0.70 of the f32 MFMA peak
// (K-tiles of 64 -- half the barriers, but 164 registers and one workgroup per CU -- measured 18.9 ms: not used).
typedef Cfg<16, 128, 128, 32, 4, 2, 1, 1, 0, 0, 0, 1> CfgConv;
typedef Cfg<16, 128, 64, 32, 4, 2, 1, 1, 0, 0, 0, 1> CfgConv64;       // C_out <= 64, or too few 128 x 128 tiles for the chip
typedef Cfg<16, 128, 64, 32, 4, 2, 1, 1, 0, 0, 0, 0> CfgConvFirst;    // conv1_1 (C_in = 3 -> 4: the general loader; K = 36)
typedef Cfg<16, 64, 64, 32, 2, 2, 2, 1> CfgFc;
// gfx950 form of the bf16 mode (BF = 2: bf16 operands in memory; K-tiles of 32 storage floats = 64 k)
typedef Cfg<16, 128, 128, 32, 4, 2, 1, 1, 0, 0, 2, 1> CfgConvB;
typedef Cfg<16, 128, 64, 32, 4, 2, 1, 1, 0, 0, 2, 1> CfgConv64B;
typedef Cfg<16, 64, 64, 32, 2, 2, 2, 1, 0, 0, 2> CfgFcB;

} // namespace

struct nvqa_vgg {
    int device = 0, div = 1, hw = 224, max_batch = 0;
    hipStream_t s = nullptr;
    int cin[13], cinp[13], cout[13], coutp[13];
    int c5 = 0, s5 = 0, fcin = 0, F = 0;
    size_t w_off[15], b_off[15], flat_total = 0; // offsets in the ABI (Caffe-layout) weight vector
    float *Wc[13] = {}, *bc[13] = {};           // repacked conv weights [Cout][3][3][Cin_pad], bias
    float *Wf[2] = {}, *bf[2] = {};             // fc6 [F][s5*s5*c5p] (columns in NHWC order), fc7 [F][F]
    float *img = nullptr, *act[2] = {}, *slabs = nullptr, *fc6o = nullptr, *fc7o = nullptr, *nhwc_in = nullptr;
    size_t act_floats = 0;
    bool have_weights = false;
    bool bf16 = false; // nvqa_vgg16_set_precision: operands of every convolution / fc product rounded to bf16, f32 accumulate
    // gfx950 form of the bf16 mode (g950; every C_in from conv1_2 on and the fc6 width a multiple of 64: the full-width
    // network): bf16 images of the weights, bf16 NHWC activations written by the epilogues, bf16 LDS images,
    // v_mfma_f32_16x16x32_bf16.  The values are the ones the BF = 1 form rounds on the fly.  NVQA_VGG_BF16_FORM=1 keeps BF = 1.
    bool g950 = false;
    __bf16 *Wc16[13] = {}, *Wf16 = nullptr;
    // host images -> device in CHUNKS on a copy stream, chunk k+1 travelling while the network runs on chunk k
    // (001_prepro_img_vgg.lua:101-113 copies image by image and then forwards the batch)
    hipStream_t sc = nullptr;
    hipEvent_t evCopy[2] = {}, evDone = nullptr; // chunk on the device / last forward finished with v->img
    bool done_rec = false;
    int chunk = 256; // measured at B = 512 (configs[4]): one shot 142.0 ms, chunks of 256 140.1, of 128 143.2 (smaller launches fill the chip worse)
};

static int vgg_layout(nvqa_vgg *v)
{
    size_t off = 0;
    int cin = 3;
    for (int i = 0; i < 13; ++i) {
        v->cin[i] = cin;
        v->cinp[i] = (cin + 3) / 4 * 4;
        v->cout[i] = std::max(1, kConvCout[i] / v->div);
        v->coutp[i] = (v->cout[i] + 3) / 4 * 4;
        v->w_off[i] = off; off += (size_t)v->cout[i] * cin * 9;
        v->b_off[i] = off; off += v->cout[i];
        cin = v->cout[i];
    }
    v->c5 = cin;
    v->s5 = v->hw / 32;
    v->fcin = v->c5 * v->s5 * v->s5;
    v->F = std::max(4, 4096 / v->div);
    v->w_off[13] = off; off += (size_t)v->F * v->fcin;
    v->b_off[13] = off; off += v->F;
    v->w_off[14] = off; off += (size_t)v->F * v->F;
    v->b_off[14] = off; off += v->F;
    v->flat_total = off;
    return 0;
}

extern "C" int nvqa_vgg16_create(int device, int width_div, int input_hw, int max_batch, nvqa_vgg **out)
{
    if (!out) { set_error("out is NULL"); return -1; }
    *out = nullptr;
    if (width_div < 1 || 64 % width_div != 0 || input_hw < 32 || input_hw % 32 || max_batch < 1) {
        set_error("bad VGG config (width_div=%d must divide 64, input_hw=%d must be a multiple of 32, max_batch=%d)",
                  width_div, input_hw, max_batch);
        return -1;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device available (libnvqa has no CPU fallback)"); return -2; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range", device); return -1; }
    NVQA_HIP(hipSetDevice(device));
    nvqa_vgg *v = new nvqa_vgg();
    v->device = device; v->div = width_div; v->hw = input_hw; v->max_batch = max_batch;
    vgg_layout(v);
    NVQA_HIP(hipStreamCreateWithFlags(&v->s, hipStreamNonBlocking));
    NVQA_HIP(hipStreamCreateWithFlags(&v->sc, hipStreamNonBlocking));
    for (int p = 0; p < 2; ++p) {
        NVQA_HIP(hipEventCreateWithFlags(&v->evCopy[p], hipEventDisableTiming));
    }
    NVQA_HIP(hipEventCreateWithFlags(&v->evDone, hipEventDisableTiming));
    { const char *e = getenv("NVQA_VGG_CHUNK"); if (e && atoi(e) > 0) v->chunk = atoi(e); }
    for (int i = 0; i < 13; ++i) {
        NVQA_HIP(hipMalloc((void **)&v->Wc[i], (size_t)v->cout[i] * 9 * v->cinp[i] * 4));
        NVQA_HIP(hipMalloc((void **)&v->bc[i], (size_t)v->cout[i] * 4));
    }
    const int c5p = (v->c5 + 3) / 4 * 4;
    NVQA_HIP(hipMalloc((void **)&v->Wf[0], (size_t)v->F * v->s5 * v->s5 * c5p * 4));
    NVQA_HIP(hipMalloc((void **)&v->Wf[1], (size_t)v->F * v->F * 4));
    NVQA_HIP(hipMalloc((void **)&v->bf[0], (size_t)v->F * 4));
    NVQA_HIP(hipMalloc((void **)&v->bf[1], (size_t)v->F * 4));
    const size_t px = (size_t)max_batch * input_hw * input_hw;
    v->act_floats = px * std::max(4, v->coutp[0]); // largest activation: conv1 output at full resolution
    NVQA_HIP(hipMalloc((void **)&v->img, px * 3 * 4));
    NVQA_HIP(hipMalloc((void **)&v->nhwc_in, px * 4 * 4));
    NVQA_HIP(hipMalloc((void **)&v->act[0], v->act_floats * 4));
    NVQA_HIP(hipMalloc((void **)&v->act[1], v->act_floats * 4));
    NVQA_HIP(hipMalloc((void **)&v->slabs, (size_t)16 * max_batch * v->F * 4));
    NVQA_HIP(hipMalloc((void **)&v->fc6o, (size_t)max_batch * v->F * 4));
    NVQA_HIP(hipMalloc((void **)&v->fc7o, (size_t)max_batch * v->F * 4));
    *out = v;
    return 0;
}

extern "C" int nvqa_vgg16_destroy(nvqa_vgg *v)
{
    if (!v) return 0;
    (void)hipSetDevice(v->device);
    (void)hipStreamSynchronize(v->s);
    for (int i = 0; i < 13; ++i) { (void)hipFree(v->Wc[i]); (void)hipFree(v->bc[i]); if (v->Wc16[i]) (void)hipFree(v->Wc16[i]); }
    if (v->Wf16) (void)hipFree(v->Wf16);
    for (int i = 0; i < 2; ++i) { (void)hipFree(v->Wf[i]); (void)hipFree(v->bf[i]); (void)hipFree(v->act[i]); }
    (void)hipFree(v->img); (void)hipFree(v->nhwc_in); (void)hipFree(v->slabs); (void)hipFree(v->fc6o); (void)hipFree(v->fc7o);
    for (int p = 0; p < 2; ++p) {
        if (v->evCopy[p]) (void)hipEventDestroy(v->evCopy[p]);
    }
    if (v->evDone) (void)hipEventDestroy(v->evDone);
    if (v->sc) (void)hipStreamDestroy(v->sc);
    (void)hipStreamDestroy(v->s);
    delete v;
    return 0;
}

// bf16 images of the repacked weights of conv1_2 .. conv5_3 and fc6 (conv1_1 and fc7 read f32 operands: C_in = 3 and an f32 input)
static int vgg_bf16_images(nvqa_vgg *v)
{
    auto conv = [&](const float *src, size_t n, __bf16 **dst) -> int {
        if (!*dst) NVQA_HIP(hipMalloc((void **)dst, n * 2));
        hipLaunchKernelGGL(k_to_bf16, dim3((n + 255) / 256), dim3(256), 0, v->s, src, n, *dst);
        NVQA_HIP(hipGetLastError());
        return 0;
    };
    for (int i = 1; i < 13; ++i) NVQA_TRY(conv(v->Wc[i], (size_t)v->cout[i] * 9 * v->cinp[i], &v->Wc16[i]));
    const int c5p = (v->c5 + 3) / 4 * 4;
    NVQA_TRY(conv(v->Wf[0], (size_t)v->F * v->s5 * v->s5 * c5p, &v->Wf16));
    NVQA_HIP(hipStreamSynchronize(v->s));
    return 0;
}

extern "C" size_t nvqa_vgg16_weight_count(const nvqa_vgg *v) { return v ? v->flat_total : 0; }
extern "C" int nvqa_vgg16_feature_dim(const nvqa_vgg *v) { return v ? v->F : 0; }

// weights: one flat vector in Caffe order and layout -- for each conv W [Cout][Cin][3][3], b [Cout];
// fc6 W [F][C5*S*S] (columns in CHW order), b; fc7 W [F][F], b.
extern "C" int nvqa_vgg16_set_weights(nvqa_vgg *v, const float *flat)
{
    if (!v || !flat) { set_error("NULL argument"); return -1; }
    NVQA_HIP(hipSetDevice(v->device));
    NVQA_HIP(hipStreamSynchronize(v->s));
    for (int i = 0; i < 13; ++i) {
        const int co = v->cout[i], ci = v->cin[i], cp = v->cinp[i];
        std::vector<float> w((size_t)co * 9 * cp, 0.f);
        const float *src = flat + v->w_off[i];
        for (int o = 0; o < co; ++o)
            for (int c = 0; c < ci; ++c)
                for (int t = 0; t < 9; ++t) w[((size_t)o * 9 + t) * cp + c] = src[((size_t)o * ci + c) * 9 + t];
        NVQA_HIP(hipMemcpy(v->Wc[i], w.data(), w.size() * 4, hipMemcpyHostToDevice));
        NVQA_HIP(hipMemcpy(v->bc[i], flat + v->b_off[i], (size_t)co * 4, hipMemcpyHostToDevice));
    }
    {   // fc6: CHW-flattened columns -> (y, x, c_padded) = the NHWC pool5 activation
        const int c5 = v->c5, c5p = (c5 + 3) / 4 * 4, S = v->s5, F = v->F;
        std::vector<float> w((size_t)F * S * S * c5p, 0.f);
        const float *src = flat + v->w_off[13];
        for (int f = 0; f < F; ++f)
            for (int c = 0; c < c5; ++c)
                for (int p = 0; p < S * S; ++p) w[((size_t)f * S * S + p) * c5p + c] = src[(size_t)f * v->fcin + (size_t)c * S * S + p];
        NVQA_HIP(hipMemcpy(v->Wf[0], w.data(), w.size() * 4, hipMemcpyHostToDevice));
        NVQA_HIP(hipMemcpy(v->bf[0], flat + v->b_off[13], (size_t)F * 4, hipMemcpyHostToDevice));
        NVQA_HIP(hipMemcpy(v->Wf[1], flat + v->w_off[14], (size_t)F * F * 4, hipMemcpyHostToDevice));
        NVQA_HIP(hipMemcpy(v->bf[1], flat + v->b_off[14], (size_t)F * 4, hipMemcpyHostToDevice));
    }
    v->have_weights = true;
    if (v->g950) NVQA_TRY(vgg_bf16_images(v));
    return 0;
}

static int xcd_order() // NVQA_XCD=0: natural tile order (A/B measurements)
{
    static const int v = [] { const char *e = getenv("NVQA_XCD"); return (e && e[0] == '0') ? 0 : 1; }();
    return v;
}

// b16: x and W are bf16 arrays (the gfx950 form of fc6); K counts their elements
static int fc_layer(nvqa_vgg *v, const float *x, int M, int K, const float *W, const float *b, float *out, bool b16 = false)
{
    const int N = v->F;
    if (b16) K /= 2; // storage floats
    int ks = 1;
    while (ks < 16 && (size_t)((M + 63) / 64) * ((N + 63) / 64) * ks < 512 && K / (ks * 2) >= 256) ks *= 2;
    int kslice = ((K + ks - 1) / ks + 31) / 32 * 32;
    ks = (K + kslice - 1) / kslice;
    GemmArgs g = {};
    g.A = x; g.lda = K; g.B = W; g.ldb = K; g.M = M; g.N = N; g.K = K; g.kslice = kslice;
    g.xcd = xcd_order();
    if (b16) NVQA_HIP((launch_gemm<CfgFcB, A_KC, B_KC, false, EpiSlab>(v->s, g, EpiSlab{v->slabs, N, (size_t)M * N})));
    else if (v->bf16) NVQA_HIP((launch_gemm<WithBF<CfgFc>::type, A_KC, B_KC, false, EpiSlab>(v->s, g, EpiSlab{v->slabs, N, (size_t)M * N})));
    else NVQA_HIP((launch_gemm<CfgFc, A_KC, B_KC, false, EpiSlab>(v->s, g, EpiSlab{v->slabs, N, (size_t)M * N})));
    hipLaunchKernelGGL(k_fc_finish, dim3(((size_t)M * N + 255) / 256), dim3(256), 0, v->s, v->slabs, ks, M, N, b, out);
    NVQA_HIP(hipGetLastError());
    return 0;
}

// the network on n images that are already on the device (img [n][3][hw][hw]); post-ReLU fc7 features -> out [n x F]
static int vgg_network(nvqa_vgg *v, const float *img, int n, float *out)
{
    int H = v->hw, W = v->hw;
    const size_t px = (size_t)n * H * W;
    hipLaunchKernelGGL(k_nchw_to_nhwc4, dim3((px + 255) / 256), dim3(256), 0, v->s, img, n, H, W, reinterpret_cast<float4 *>(v->nhwc_in));
    const float *cur = v->nhwc_in;
    int which = 0;
    for (int i = 0; i < 13; ++i) {
        float *dst = v->act[which];
        GemmArgs g = {};
        g.A = cur; g.B = v->Wc[i]; g.ldb = 9 * v->cinp[i];
        g.M = n * H * W; g.N = v->cout[i]; g.K = 9 * v->cinp[i]; g.kslice = g.K;
        g.cH = H; g.cW = W; g.cC = v->cinp[i];
        g.xcd = xcd_order(); // neighbouring pixel-row tiles (shared halo rows) and all C_out tiles of a row tile on one XCD
        // output channel stride = padded C_out, so that the next layer reads float4 channels; the pad
        // channels must be zero: they are written by nobody, so clear once when padding exists
        const int ldc = v->coutp[i];
        if (v->g950) {
            // activations are bf16 NHWC from conv1_1's epilogue on; sizes of the products in storage floats (two bf16)
            const EpiBiasReluB16 eb{reinterpret_cast<__bf16 *>(dst), ldc, v->bc[i]};
            const long tiles = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
            if (i == 0) {
                NVQA_HIP((launch_gemm<WithBF<CfgConvFirst>::type, A_IM2COL, B_KC, false, EpiBiasReluB16>(v->s, g, eb)));
            } else {
                g.B = reinterpret_cast<const float *>(v->Wc16[i]);
                g.ldb = g.K = g.kslice = 9 * v->cinp[i] / 2;
                g.cC = v->cinp[i] / 2;
                if (v->cout[i] > 64 && tiles >= 256) NVQA_HIP((launch_gemm<CfgConvB, A_IM2COLF, B_KC, false, EpiBiasReluB16>(v->s, g, eb)));
                else NVQA_HIP((launch_gemm<CfgConv64B, A_IM2COLF, B_KC, false, EpiBiasReluB16>(v->s, g, eb)));
            }
            cur = dst; which ^= 1;
            if (kPoolAfter[i]) {
                float *pd = v->act[which];
                const size_t total = (size_t)n * (H / 2) * (W / 2) * (ldc / 8);
                hipLaunchKernelGGL(k_maxpool2_nhwc_b16, dim3((total + 255) / 256), dim3(256), 0, v->s, reinterpret_cast<const uint4 *>(cur), n, H, W,
                                   ldc / 8, reinterpret_cast<uint4 *>(pd));
                H /= 2; W /= 2;
                cur = pd; which ^= 1;
            }
            continue;
        }
        if (ldc != v->cout[i]) NVQA_HIP(hipMemsetAsync(dst, 0, (size_t)g.M * ldc * 4, v->s));
        // A_IM2COLF: every K-tile inside one tap (C_in a multiple of the K-tile: all layers of the full-width network but conv1_1)
        const EpiBiasRelu ep{dst, ldc, v->bc[i]};
#define NVQA_CONV_GO(CFG)                                                                                              \
    do {                                                                                                               \
        const bool fast = v->cinp[i] % CFG::BK == 0;                                                                   \
        if (v->bf16 && fast) NVQA_HIP((launch_gemm<WithBF<CFG>::type, A_IM2COLF, B_KC, false, EpiBiasRelu>(v->s, g, ep))); \
        else if (v->bf16) NVQA_HIP((launch_gemm<WithBF<CFG>::type, A_IM2COL, B_KC, false, EpiBiasRelu>(v->s, g, ep)));  \
        else if (fast) NVQA_HIP((launch_gemm<CFG, A_IM2COLF, B_KC, false, EpiBiasRelu>(v->s, g, ep)));                  \
        else NVQA_HIP((launch_gemm<CFG, A_IM2COL, B_KC, false, EpiBiasRelu>(v->s, g, ep)));                             \
    } while (0)
        // 128 x 128 tiles when there is at least one per CU, 128 x 64 otherwise (conv5 at batch 32: 49 row tiles x 4 = 196
        // workgroups on 256 CUs; at batch 64 -- 392 tiles -- the wide tile is the faster one again: 600 vs 645 us)
        const long tiles128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
        if (v->cinp[i] % 32 != 0 && v->cout[i] <= 64) NVQA_CONV_GO(CfgConvFirst);
        else if (v->cout[i] > 64 && tiles128 >= 256) NVQA_CONV_GO(CfgConv);
        else NVQA_CONV_GO(CfgConv64);
#undef NVQA_CONV_GO
        cur = dst; which ^= 1;
        if (kPoolAfter[i]) {
            float *pd = v->act[which];
            const size_t total = (size_t)n * (H / 2) * (W / 2) * (ldc / 4);
            hipLaunchKernelGGL(k_maxpool2_nhwc, dim3((total + 255) / 256), dim3(256), 0, v->s, reinterpret_cast<const float4 *>(cur), n, H, W, ldc / 4, reinterpret_cast<float4 *>(pd));
            H /= 2; W /= 2;
            cur = pd; which ^= 1;
        }
    }
    NVQA_HIP(hipGetLastError());
    const int c5p = (v->c5 + 3) / 4 * 4;
    if (v->g950) NVQA_TRY(fc_layer(v, cur, n, v->s5 * v->s5 * c5p, reinterpret_cast<const float *>(v->Wf16), v->bf[0], v->fc6o, true));
    else NVQA_TRY(fc_layer(v, cur, n, v->s5 * v->s5 * c5p, v->Wf[0], v->bf[0], v->fc6o)); // fc6 + ReLU (Dropout = identity)
    NVQA_TRY(fc_layer(v, v->fc6o, n, v->F, v->Wf[1], v->bf[1], out));                  // fc7 + ReLU -> module 38
    return 0;
}

// forward of n host images; the post-ReLU fc7 features stay on the device in v->fc7o [n x F].
// Up to one chunk: one copy, one pass.  More: chunks of v->chunk images (NVQA_VGG_CHUNK), the copy of chunk k+1 on the copy
// stream under the network on chunk k, so that of the 0.6 MB per image only the first chunk's transfer is exposed.
static int vgg_forward(nvqa_vgg *v, const float *images, int n)
{
    if (!v || !images) { set_error("NULL argument"); return -1; }
    if (!v->have_weights) { set_error("nvqa_vgg16_fc7 before nvqa_vgg16_set_weights"); return -1; }
    if (n < 1 || n > v->max_batch) { set_error("n=%d outside 1..%d", n, v->max_batch); return -1; }
    NVQA_HIP(hipSetDevice(v->device));
    const size_t per = (size_t)3 * v->hw * v->hw;
    if (n <= v->chunk) {
        NVQA_HIP(hipMemcpyAsync(v->img, images, (size_t)n * per * 4, hipMemcpyHostToDevice, v->s));
        NVQA_TRY(vgg_network(v, v->img, n, v->fc7o));
        NVQA_HIP(hipEventRecord(v->evDone, v->s));
        v->done_rec = true;
        return 0;
    }
    // v->img is written chunk by chunk below: the previous call's network must have converted all of it
    if (v->done_rec) NVQA_HIP(hipStreamWaitEvent(v->sc, v->evDone, 0));
    int c = 0;
    for (int i0 = 0; i0 < n; i0 += v->chunk, ++c) {
        const int m = std::min(v->chunk, n - i0), p = c & 1;
        // the caller's memory is pageable: the runtime stages it through its own pinned buffers and returns when the chunk
        // has left the host (the network on the chunks before it is already queued and runs meanwhile).  A staging copy
        // of our own (memcpy into hipHostMalloc'ed buffers) measured slower: the single-threaded memcpy of 308 MB takes
        // as long as the whole bf16 extractor.
        NVQA_HIP(hipMemcpyAsync(v->img + (size_t)i0 * per, images + (size_t)i0 * per, (size_t)m * per * 4, hipMemcpyHostToDevice, v->sc));
        NVQA_HIP(hipEventRecord(v->evCopy[p], v->sc));
        NVQA_HIP(hipStreamWaitEvent(v->s, v->evCopy[p], 0));
        NVQA_TRY(vgg_network(v, v->img + (size_t)i0 * per, m, v->fc7o + (size_t)i0 * v->F));
    }
    NVQA_HIP(hipEventRecord(v->evDone, v->s));
    v->done_rec = true;
    return 0;
}

// images: n x 3 x hw x hw, already preprocessed (BGR, mean-subtracted: loadim's output).
extern "C" int nvqa_vgg16_fc7(nvqa_vgg *v, const float *images, int n, float *feats_out)
{
    if (!feats_out) { set_error("NULL argument"); return -1; }
    NVQA_TRY(vgg_forward(v, images, n));
    NVQA_HIP(hipMemcpyAsync(feats_out, v->fc7o, (size_t)n * v->F * 4, hipMemcpyDeviceToHost, v->s));
    NVQA_HIP(hipStreamSynchronize(v->s));
    return 0;
}

// used by nvqa_step_images (nvqa_api.hip): run the extractor, leave the features on the device
int nvqa_vgg_forward_device(nvqa_vgg *v, const float *images, int n, const float **feats_dev, int *F, hipStream_t *stream)
{
    NVQA_TRY(vgg_forward(v, images, n));
    *feats_dev = v->fc7o;
    *F = v->F;
    *stream = v->s;
    return 0;
}

extern "C" int nvqa_vgg16_set_precision(nvqa_vgg *v, int bf16)
{
    if (!v) { set_error("vgg is NULL"); return -1; }
    if (bf16 != 0 && bf16 != 1) { set_error("precision must be 0 (f32) or 1 (bf16 operands)"); return -1; }
    v->bf16 = bf16 != 0;
    const char *form = getenv("NVQA_VGG_BF16_FORM"); // read at every call: the tests compare the two forms in one process
    const bool old_form = form && form[0] == '1';
    bool ok = v->bf16 && !old_form && (v->s5 * v->s5 * ((v->c5 + 3) / 4 * 4)) % 64 == 0;
    for (int i = 1; i < 13 && ok; ++i) ok = v->cinp[i] % 64 == 0 && v->coutp[i] == v->cout[i];
    v->g950 = ok;
    if (v->g950 && v->have_weights) {
        NVQA_HIP(hipSetDevice(v->device));
        NVQA_TRY(vgg_bf16_images(v));
    }
    return 0;
}

// rgb: n x 3 x H x W in [0,1] (image.load output) -> out: n x 3 x hw x hw preprocessed planes
extern "C" int nvqa_vgg16_preprocess(nvqa_vgg *v, const float *rgb, int n, int H, int W, float *out)
{
    if (!v || !rgb || !out) { set_error("NULL argument"); return -1; }
    if (n < 1 || H < 1 || W < 1) { set_error("bad image shape"); return -1; }
    NVQA_HIP(hipSetDevice(v->device));
    float *din = nullptr, *dout = nullptr;
    const size_t nin = (size_t)n * 3 * H * W, nout = (size_t)n * 3 * v->hw * v->hw;
    NVQA_HIP(hipMalloc((void **)&din, nin * 4));
    NVQA_HIP(hipMalloc((void **)&dout, nout * 4));
    NVQA_HIP(hipMemcpyAsync(din, rgb, nin * 4, hipMemcpyHostToDevice, v->s));
    hipLaunchKernelGGL(k_vgg_preprocess, dim3((nout + 255) / 256), dim3(256), 0, v->s, din, n, H, W, v->hw, dout);
    NVQA_HIP(hipGetLastError());
    NVQA_HIP(hipMemcpyAsync(out, dout, nout * 4, hipMemcpyDeviceToHost, v->s));
    NVQA_HIP(hipStreamSynchronize(v->s));
    (void)hipFree(din); (void)hipFree(dout);
    return 0;
}
