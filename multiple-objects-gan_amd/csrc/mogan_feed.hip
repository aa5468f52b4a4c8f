// mogan_feed.hip -- device side of the input pipeline of the coco-attngan train step.
//
// Reference (host, per sample, in DataLoader workers): code/coco/attngan/datasets.py:70-137 get_imgs / crop_imgs --
//   PIL image resized to 268x268 -> ToTensor (u8 / 255) -> random 256x256 crop + horizontal flip -> for the 64x64 and
//   128x128 branches ToPILImage (back to u8) + transforms.Resize (PIL bilinear, antialiased, fixed-point) -> ToTensor +
//   Normalize(0.5, 0.5); the 256x256 branch is the crop itself, normalised.
// Here the host hands over the decoded 268x268 u8 images (215 KB per sample instead of 1 MB of floats over PCIe) and the
// crop offsets / flip flags it drew; crop, flip, the two PIL-compatible resamplings (same coefficient tables, same 22-bit
// fixed-point accumulation and u8 rounding between the horizontal and the vertical pass as Pillow's
// ImagingResampleHorizontal/Vertical_8bpc) and the normalisation run as four small kernels per batch.  Byte work:
// results are bit-identical to the reference pipeline (tests/test_feeder_*.py).  HBM-bound; trivial next to the step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mogan_hip.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;       // Pillow: src/libImaging/Resample.c

static inline int ok_launch() { return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH; }
static inline unsigned nblk(long long n) { long long b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : b); }

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// src (B, ori, ori, 3) u8 HWC; params (B, 3) = {h1 (column offset), w1 (row offset), flip} as in crop_imgs ->
// q (B, size, size, 3) u8 HWC (the cropped / flipped image, = ToPILImage of the float crop) and
// out (B, 3, size, size) f32 = (u8 / 255 - 0.5) / 0.5
__global__ __launch_bounds__(256) void feed_crop_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ params,
                                                        uint8_t* __restrict__ q, float* __restrict__ out, int ori, int size,
                                                        long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // over (b, y, x)
    if (i >= total) return;
    const int x = (int)(i % size); const long long t = i / size; const int y = (int)(t % size); const int b = (int)(t / size);
    const int h1 = params[b * 3], w1 = params[b * 3 + 1], flip = params[b * 3 + 2];
    const int sx = h1 + (flip ? size - 1 - x : x), sy = w1 + y;
    const uint8_t* p = src + (((size_t)b * ori + sy) * ori + sx) * 3;
    uint8_t* pq = q + (size_t)i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint8_t v = p[c];
        pq[c] = v;
        const float f = (float)v / 255.0f;                       // ToTensor
        out[(((size_t)b * 3 + c) * size + y) * size + x] = (f - 0.5f) / 0.5f;   // Normalize
    }
}

// horizontal pass: in (B, H, W, 3) u8 -> out (B, H, OW, 3) u8; bounds (OW, 2) = {xmin, count}, kk (OW, ksize) fixed point
__global__ __launch_bounds__(256) void feed_resample_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                              const int32_t* __restrict__ bounds,
                                                              const int32_t* __restrict__ kk, int ksize, int H, int W,
                                                              int OW, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // over (b, y, ox)
    if (i >= total) return;
    const int ox = (int)(i % OW); const long long row = i / OW;          // row = b * H + y
    const int xmin = bounds[ox * 2], cnt = bounds[ox * 2 + 1];
    const int32_t* k = kk + (size_t)ox * ksize;
    const uint8_t* p = in + ((size_t)row * W + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < cnt; ++x) { s0 += p[x * 3] * k[x]; s1 += p[x * 3 + 1] * k[x]; s2 += p[x * 3 + 2] * k[x]; }
    uint8_t* o = out + (size_t)i * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// vertical pass + ToTensor + Normalize: in (B, H, OW, 3) u8 -> out (B, 3, OH, OW) f32
__global__ __launch_bounds__(256) void feed_resample_v_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                              const int32_t* __restrict__ bounds,
                                                              const int32_t* __restrict__ kk, int ksize, int H, int OH,
                                                              int OW, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // over (b, oy, ox)
    if (i >= total) return;
    const int ox = (int)(i % OW); const long long t = i / OW; const int oy = (int)(t % OH); const int b = (int)(t / OH);
    const int ymin = bounds[oy * 2], cnt = bounds[oy * 2 + 1];
    const int32_t* k = kk + (size_t)oy * ksize;
    const uint8_t* p = in + (((size_t)b * H + ymin) * OW + ox) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < cnt; ++y) {
        const uint8_t* r = p + (size_t)y * OW * 3;
        s0 += r[0] * k[y]; s1 += r[1] * k[y]; s2 += r[2] * k[y];
    }
    const int v[3] = {clip8(s0), clip8(s1), clip8(s2)};
#pragma unroll
    for (int c = 0; c < 3; ++c)
        out[(((size_t)b * 3 + c) * OH + oy) * OW + ox] = ((float)v[c] / 255.0f - 0.5f) / 0.5f;
}

}  // namespace

extern "C" {

int mogan_feed_crop_flip(const uint8_t* src, const int32_t* params, uint8_t* q, float* out, int B, int ori, int size,
                         hipStream_t stream) {
    if (B <= 0 || size <= 0 || ori < size) return MOGAN_ERR_SHAPE;
    const long long n = (long long)B * size * size;
    hipLaunchKernelGGL(feed_crop_kernel, dim3(nblk(n)), dim3(256), 0, stream, src, params, q, out, ori, size, n);
    return ok_launch();
}

int mogan_feed_resample(const uint8_t* in, uint8_t* tmp, float* out, const int32_t* bounds, const int32_t* kk, int ksize,
                        int B, int S, int OS, hipStream_t stream) {
    if (B <= 0 || S <= 0 || OS <= 0 || ksize <= 0) return MOGAN_ERR_SHAPE;
    const long long nh = (long long)B * S * OS, nv = (long long)B * OS * OS;
    hipLaunchKernelGGL(feed_resample_h_kernel, dim3(nblk(nh)), dim3(256), 0, stream, in, tmp, bounds, kk, ksize, S, S, OS, nh);
    hipLaunchKernelGGL(feed_resample_v_kernel, dim3(nblk(nv)), dim3(256), 0, stream, (const uint8_t*)tmp, out, bounds, kk,
                       ksize, S, OS, OS, nv);
    return ok_launch();
}

}  // extern "C"
