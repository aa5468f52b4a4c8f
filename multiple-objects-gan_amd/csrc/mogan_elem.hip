// mogan_elem.hip -- HBM-bound elementwise / small-reduction kernels of the AttnGAN step:
// activations (LeakyReLU, GLU, tanh, sigmoid), bias, add/scale, nearest-upsample backward, strided
// softmax, BCE / KL losses, pooling + bilinear resize (CNN_ENCODER trunk) and the fused Adam(+EMA)
// step over flat fp32 buckets.  Grid-stride / float4 where the layout allows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mogan_hip.h"

namespace {

static inline int ok_launch() { return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH; }

// zero-fill as a kernel node (not hipMemsetAsync): memset nodes captured into a hipGraph were observed
// to race with the atomics kernel that follows them on this ROCm (rare NaN gradients under replay)
__global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
}
static inline void zero_fill(float* p, size_t n, hipStream_t st) {
    size_t b = (n + 255) / 256; if (b > 65536) b = 65536; if (b < 1) b = 1;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)b), dim3(256), 0, st, p, n);
}
#define POOL_LAUNCH(K, total, ...) do { if ((total) < (1ll << 31)) hipLaunchKernelGGL((K<unsigned>), __VA_ARGS__); \
                                         else hipLaunchKernelGGL((K<long long>), __VA_ARGS__); } while (0)
static inline unsigned nblk(long long n, int per = 256) {
    long long b = (n + per - 1) / per;
    return (unsigned)(b < 1 ? 1 : (b > 262144 ? 262144 : b));     // grid-stride beyond this
}
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sumd(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// -------------------------------------------------------------------------------- activations
template <int ACT, bool BWD>
__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                  float* __restrict__ out, long long n, float slope) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        float o;
        if (!BWD) {
            if (ACT == MOGAN_ACT_RELU) o = v > 0.f ? v : 0.f;
            else if (ACT == MOGAN_ACT_LRELU) o = v > 0.f ? v : v * slope;
            else if (ACT == MOGAN_ACT_TANH) o = tanhf(v);
            else o = sigmoidf_(v);
        } else {
            const float d = dy[i];
            if (ACT == MOGAN_ACT_RELU) o = v > 0.f ? d : 0.f;
            else if (ACT == MOGAN_ACT_LRELU) o = v > 0.f ? d : d * slope;
            else if (ACT == MOGAN_ACT_TANH) { const float t = tanhf(v); o = d * (1.f - t * t); }
            else { const float s = sigmoidf_(v); o = d * s * (1.f - s); }
        }
        out[i] = o;
    }
}

// GLU over the channel dim: x (B,C,HW) -> y (B,C/2,HW)
template <bool BWD>
__global__ __launch_bounds__(256) void glu_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                  float* __restrict__ out, long long total, int Ch, long long HW) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long hw = i % HW, bc = i / HW;
        const long long c = bc % Ch, b = bc / Ch;
        const long long ia = (b * 2 * Ch + c) * HW + hw, ig = ia + (long long)Ch * HW;
        const float a = x[ia], s = sigmoidf_(x[ig]);
        if (!BWD) out[i] = a * s;
        else { const float d = dy[i]; out[ia] = d * s; out[ig] = d * a * s * (1.f - s); }
    }
}

__global__ __launch_bounds__(256) void bias_add_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                       long long total, int C, long long HW) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        y[i] += bias[(i / HW) % C];
}

// one block per channel
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int rows,
                                                        int C, int HW, int accumulate) {
    __shared__ double sh[4];
    const int c = blockIdx.x;
    double s = 0;
    const long long per = (long long)rows * HW;
    for (long long e = threadIdx.x; e < per; e += 256) {
        const long long r = e / HW, i = e % HW;
        s += dy[(r * C + c) * HW + i];
    }
    s = wave_sumd(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) db[c] = (accumulate ? db[c] : 0.f) + (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void group_sum_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, int G) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float t = x[i];
        for (int g = 1; g < G; ++g) t += x[(long long)g * n + i];
        y[i] = t;
    }
}
__global__ __launch_bounds__(256) void group_bcast_kernel(const float* __restrict__ dy, float* __restrict__ dx, long long n, int G) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float t = dy[i];
        for (int g = 0; g < G; ++g) dx[(long long)g * n + i] = t;
    }
}
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = a[i] + b[i];
}
__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ a, float alpha, float* __restrict__ y,
                                                    long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = a[i] * alpha;
}

// dx[p,y,x] = sum of the 2x2 block of du[p, 2y..2y+1, 2x..2x+1]
__global__ __launch_bounds__(256) void down2_sum_kernel(const float* __restrict__ du, float* __restrict__ dx,
                                                        long long total, int H, int W) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xw = (int)(i % W);
        const long long t = i / W;
        const int y = (int)(t % H);
        const long long pl = t / H;
        const float* p = du + (pl * 2 * H + 2 * y) * 2 * W + 2 * xw;
        const float2 r0 = *(const float2*)p, r1 = *(const float2*)(p + 2 * W);
        dx[i] = (r0.x + r0.y) + (r1.x + r1.y);
    }
}

// ReLU backward on its own: dx (+)= dz where z > 0 (z = the ReLU's output)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dz,
                                                       float* __restrict__ dx, long long n, int accumulate) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float g = z[i] > 0.f ? dz[i] : 0.f;
        dx[i] = accumulate ? dx[i] + g : g;
    }
}

// rows of n contiguous floats between two batch-strided tensors (channel slices of NCHW tensors)
__global__ __launch_bounds__(256) void copy_strided_kernel(const float* __restrict__ src, long long sbs,
                                                           float* __restrict__ dst, long long dbs, long long n,
                                                           long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / n, r = i - b * n;
        dst[b * dbs + r] = src[b * sbs + r];
    }
}

// -------------------------------------------------------------------------------- strided softmax
// x viewed (outer, L, inner); one thread per (o, i) column, three passes over L (L <= a few hundred).
template <bool BWD>
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                      float* __restrict__ out, const int32_t* __restrict__ lens,
                                                      long long cols, int L, long long inner, float scale) {
    const long long col = (long long)blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    const long long o = col / inner, i = col % inner;
    const long long base = o * L * inner + i;
    const int n = lens ? min(L, max(0, lens[col])) : L;
    if (!BWD) {
        float m = -INFINITY;
        for (int l = 0; l < n; ++l) m = fmaxf(m, x[base + l * inner] * scale);
        float s = 0.f;
        for (int l = 0; l < n; ++l) s += __expf(x[base + l * inner] * scale - m);
        const float inv = 1.f / s;
        for (int l = 0; l < n; ++l) out[base + l * inner] = __expf(x[base + l * inner] * scale - m) * inv;
        for (int l = n; l < L; ++l) out[base + l * inner] = 0.f;
    } else {   // x = y (softmax output)
        float dot = 0.f;
        for (int l = 0; l < n; ++l) dot += x[base + l * inner] * dy[base + l * inner];
        for (int l = 0; l < n; ++l) {
            const float yv = x[base + l * inner];
            out[base + l * inner] = scale * yv * (dy[base + l * inner] - dot);
        }
        for (int l = n; l < L; ++l) out[base + l * inner] = 0.f;
    }
}

// -------------------------------------------------------------------------------- losses (single block)
__device__ __forceinline__ float block_sum_f(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ p, float target, float weight,
                                                      float* __restrict__ loss, int n, int accumulate) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float lp = fmaxf(logf(p[i]), -100.f), l1p = fmaxf(logf(1.f - p[i]), -100.f);
        s -= target * lp + (1.f - target) * l1p;
    }
    s = block_sum_f(s, sh);
    if (threadIdx.x == 0) loss[0] = (accumulate ? loss[0] : 0.f) + weight * s / (float)n;
}
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ p, float target, float weight,
                                                      const float* __restrict__ gout, float* __restrict__ dp, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // torch's BCELoss backward (the reference's loss, miscc/losses.py:158-168): (p - t) / max(p (1 - p), 1e-12) -- NOT the
    // derivative of the clamped logs: for a saturated discriminator (p < 1e-12, logit < -27.6) it is smaller by
    // p (1 - p) / 1e-12, and that is the gradient the reference trains with
    const float pv = p[i];
    const float g = (pv - target) / fmaxf((1.f - pv) * pv, 1e-12f);
    dp[i] = gout[0] * weight * g / (float)n;
}
// BCEWithLogitsLoss(mean) against a constant target, torch's stable form: max(x,0) - x*t + log1p(exp(-|x|))
__global__ __launch_bounds__(256) void bce_logits_fwd_kernel(const float* __restrict__ x, float target, float weight,
                                                             float* __restrict__ loss, int n, int accumulate) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = x[i];
        s += fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v)));
    }
    s = block_sum_f(s, sh);
    if (threadIdx.x == 0) loss[0] = (accumulate ? loss[0] : 0.f) + weight * s / (float)n;
}
__global__ __launch_bounds__(256) void bce_logits_bwd_kernel(const float* __restrict__ x, float target, float weight,
                                                             const float* __restrict__ gout, float* __restrict__ dx,
                                                             int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float sg = 1.f / (1.f + expf(-x[i]));
    dx[i] = gout[0] * weight * (sg - target) / (float)n;
}
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                     float* __restrict__ loss, int n) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += 1.f + lv[i] - mu[i] * mu[i] - __expf(lv[i]);
    s = block_sum_f(s, sh);
    if (threadIdx.x == 0) loss[0] = -0.5f * s / (float)n;
}
__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                     const float* __restrict__ gout, float* __restrict__ dmu,
                                                     float* __restrict__ dlv, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = gout[0] * (-0.5f) / (float)n;
    dmu[i] = g * (-2.f * mu[i]);
    dlv[i] = g * (1.f - __expf(lv[i]));
}

// -------------------------------------------------------------------------------- pooling / resize
// IT = index type: unsigned for tensors below 2^31 elements (every case of the step: 64-bit divisions made these kernels
// integer-bound, maxpool_bwd 112 us on 88 MB), long long otherwise (POOL_LAUNCH picks)
// idx (nullable) records the offset a*k+b of the first maximum of each window for the backward
template <typename IT>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ idx, long long total, int H, int W,
                                                          int OH, int OW, int k, int s, int C, long long ybs) {
    for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < (IT)total; i += (IT)gridDim.x * 256) {
        const int ox = (int)(i % OW); const IT t = i / OW; const int oy = (int)(t % OH); const IT pl = t / OH;
        const long long yo = (pl / C) * ybs + ((pl % C) * OH + oy) * (long long)OW + ox;     // (strided) output position
        const float* px = x + pl * H * W;
        float m = -INFINITY; int am = 0;
        for (int a = 0; a < k; ++a) for (int b = 0; b < k; ++b) {
            const float v = px[(oy * s + a) * W + ox * s + b];
            if (v > m) { m = v; am = a * k + b; }
        }
        y[yo] = m;
        if (idx) idx[i] = (uint8_t)am;
    }
}
// gather form: dx[p,iy,ix] = sum over the windows that contain (iy,ix) and whose first maximum is there.  k <= 2 s (every caller:
// 3x3 stride 2, 2x2 stride 2): an element lies in at most 2 x 2 windows -- their index bytes and gradients are loaded unconditionally
// from clamped addresses (8 loads in flight, no per-window branch) and selected afterwards.
template <typename IT>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const uint8_t* __restrict__ idx, const float* __restrict__ dy,
                                                          float* __restrict__ dx, long long total, int H, int W, int OH,
                                                          int OW, int k, int s, int C, long long dybs,
                                                          const float* __restrict__ relu_of, int accumulate) {
    for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < (IT)total; i += (IT)gridDim.x * 256) {
        const int ix = (int)(i % W); const IT t = i / W; const int iy = (int)(t % H); const IT pl = t / H;
        const int oyb = iy / s, oxb = ix / s;                       // the last window that can contain the element; the other: - 1
        const IT img = pl / C, ch = pl - img * C;
        const long long obase = (long long)pl * OH * OW, dbase = (long long)img * dybs + (long long)ch * OH * OW;
        bool ok[4]; int code[4]; long long off[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oy = oyb - (q >> 1), ox = oxb - (q & 1);
            const int ky = iy - oy * s, kx = ix - ox * s;
            ok[q] = oy >= 0 && ox >= 0 && oy < OH && ox < OW && ky < k && kx < k;
            code[q] = ky * k + kx;
            off[q] = ok[q] ? (long long)oy * OW + ox : 0;
        }
        int am[4]; float gv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) am[q] = (int)idx[obase + off[q]];
#pragma unroll
        for (int q = 0; q < 4; ++q) gv[q] = dy[dbase + off[q]];
        float g = 0.f;
#pragma unroll
        for (int q = 3; q >= 0; --q) if (ok[q] && am[q] == code[q]) g += gv[q];      // (oy, ox) ascending, as the window loops did
        if (relu_of && !(relu_of[i] > 0.f)) g = 0.f;
        dx[i] = accumulate ? dx[i] + g : g;
    }
}
template <typename IT>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          long long total, int H, int W, int OH, int OW, int k, int s,
                                                          int pad) {
    const float inv = 1.f / (float)(k * k);                       // count_include_pad=True
    for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < (IT)total; i += (IT)gridDim.x * 256) {
        const int ox = (int)(i % OW); const IT t = i / OW; const int oy = (int)(t % OH); const IT pl = t / OH;
        const float* px = x + pl * H * W;
        float sum = 0.f;
        for (int a = 0; a < k; ++a) {
            const int yy = oy * s - pad + a; if ((unsigned)yy >= (unsigned)H) continue;
            for (int b = 0; b < k; ++b) { const int xx = ox * s - pad + b; if ((unsigned)xx < (unsigned)W) sum += px[yy * W + xx]; }
        }
        y[i] = sum * inv;
    }
}
template <typename IT>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                          long long total, int H, int W, int OH, int OW, int k, int s,
                                                          int pad, const float* __restrict__ relu_of, int accumulate) {
    const float inv = 1.f / (float)(k * k);
    for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < (IT)total; i += (IT)gridDim.x * 256) {
        const int ix = (int)(i % W); const IT t = i / W; const int iy = (int)(t % H); const IT pl = t / H;
        // windows with oy*s - pad <= iy <= oy*s - pad + k - 1
        const int oy0 = max(0, (iy + pad - k + s) / s), oy1 = min(OH - 1, (iy + pad) / s);
        const int ox0 = max(0, (ix + pad - k + s) / s), ox1 = min(OW - 1, (ix + pad) / s);
        float g = 0.f;
        for (int oy = oy0; oy <= oy1; ++oy)
            for (int ox = ox0; ox <= ox1; ++ox) g += dy[(pl * OH + oy) * OW + ox];
        g *= inv;
        if (relu_of && !(relu_of[i] > 0.f)) g = 0.f;
        dx[i] = accumulate ? dx[i] + g : g;
    }
}
// bilinear, align_corners = False (torch area_pixel_compute_source_index)
__device__ __forceinline__ void bil_src(int o, float scale, int in, int& i0, int& i1, float& l1) {
    float src = ((float)o + 0.5f) * scale - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src; i1 = i0 + (i0 < in - 1 ? 1 : 0); l1 = src - (float)i0;
}
template <typename IT>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           long long total, int H, int W, int OH, int OW) {
    const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
    for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < (IT)total; i += (IT)gridDim.x * 256) {
        const int ox = (int)(i % OW); const IT t = i / OW; const int oy = (int)(t % OH); const IT pl = t / OH;
        int y0, y1, x0, x1; float ly, lx;
        bil_src(oy, sh, H, y0, y1, ly); bil_src(ox, sw, W, x0, x1, lx);
        const float* px = x + pl * H * W;
        y[i] = (1.f - ly) * ((1.f - lx) * px[y0 * W + x0] + lx * px[y0 * W + x1]) +
               ly * ((1.f - lx) * px[y1 * W + x0] + lx * px[y1 * W + x1]);
    }
}
// Gather form (round 4; the scatter form issued four fp32 atomics per output pixel: 211 us for the 256 -> 299 resize in front of
// the Inception trunk, and an order-dependent sum): one thread per INPUT element collects the output pixels whose two taps per
// axis include it.  Output o reads inputs i0(o), i1(o) with weights 1 - l1, l1 (bil_src); src(o) is monotone in o, so the
// outputs touching input i form the contiguous range of o with src(o) in (i - 1, i + 1): bounded here with one index of slack
// on either side and decided exactly by evaluating bil_src, the forward's own arithmetic.
constexpr int BIL_MAXC = 8;          // candidates per axis the register path holds (scale factors >= 1/3: 2/s + 3 <= 8)
__device__ __forceinline__ void bil_cand(int i, float scale, int O, int& lo, int& hi) {
    const float inv = 1.f / scale;
    lo = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
    hi = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
    if (lo < 0) lo = 0;
    if (hi > O - 1) hi = O - 1;
}
__device__ __forceinline__ float bil_w(int o, float scale, int in, int i) {
    int i0, i1; float l1;
    bil_src(o, scale, in, i0, i1, l1);
    return (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
}
template <typename IT>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                           long long total, int H, int W, int OH, int OW) {
    const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
    for (IT i = (IT)blockIdx.x * 256 + threadIdx.x; i < (IT)total; i += (IT)gridDim.x * 256) {
        const int x = (int)(i % W); const IT t = i / W; const int y = (int)(t % H); const IT pl = t / H;
        int xlo, xhi, ylo, yhi;
        bil_cand(x, sw, OW, xlo, xhi); bil_cand(y, sh, OH, ylo, yhi);
        const float* py = dy + (size_t)pl * OH * OW;
        float acc = 0.f;
        if (xhi - xlo < BIL_MAXC) {
            float wx[BIL_MAXC];
#pragma unroll
            for (int k = 0; k < BIL_MAXC; ++k) wx[k] = xlo + k <= xhi ? bil_w(xlo + k, sw, W, x) : 0.f;
            for (int oy = ylo; oy <= yhi; ++oy) {
                const float wy = bil_w(oy, sh, H, y);
                if (wy == 0.f) continue;
                const float* row = py + (size_t)oy * OW + xlo;
                float r = 0.f;
#pragma unroll
                for (int k = 0; k < BIL_MAXC; ++k) if (xlo + k <= xhi) r += wx[k] * row[k];
                acc += wy * r;
            }
        } else {                                           // strong down-scaling of the gradient grid: plain double loop
            for (int oy = ylo; oy <= yhi; ++oy) {
                const float wy = bil_w(oy, sh, H, y);
                if (wy == 0.f) continue;
                float r = 0.f;
                for (int ox = xlo; ox <= xhi; ++ox) r += bil_w(ox, sw, W, x) * py[(size_t)oy * OW + ox];
                acc += wy * r;
            }
        }
        dx[i] = acc;
    }
}

// -------------------------------------------------------------------------------- Adam (+EMA)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float* __restrict__ ema,
                                                   long long n4, long long n, float beta1, float beta2, float eps,
                                                   float step_size, float inv_sqrt_bc2, int eps_mode, float gscale,
                                                   float ema_decay, const float* __restrict__ dev_state) {
    if (dev_state) { step_size = dev_state[1]; inv_sqrt_bc2 = dev_state[2]; }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const long long e = i * 4;
        float pv[4], gv[4], mv[4], vv[4], ev[4];
        const bool full = e + 3 < n;
        if (full) {
            *(float4*)pv = *(const float4*)(p + e); *(float4*)gv = *(const float4*)(g + e);
            *(float4*)mv = *(const float4*)(m + e); *(float4*)vv = *(const float4*)(v + e);
            if (ema) *(float4*)ev = *(const float4*)(ema + e);
        } else {
            for (int k = 0; k < 4; ++k) if (e + k < n) { pv[k] = p[e + k]; gv[k] = g[e + k]; mv[k] = m[e + k]; vv[k] = v[e + k]; if (ema) ev[k] = ema[e + k]; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gg = gv[k] * gscale;
            mv[k] = mv[k] * beta1 + (1.f - beta1) * gg;
            vv[k] = vv[k] * beta2 + (1.f - beta2) * gg * gg;
            const float denom = eps_mode == 0 ? (sqrtf(vv[k]) * inv_sqrt_bc2 + eps) : (sqrtf(vv[k]) + eps);
            pv[k] = pv[k] - step_size * (mv[k] / denom);
            if (ema) ev[k] = ev[k] * ema_decay + (1.f - ema_decay) * pv[k];
        }
        if (full) {
            *(float4*)(p + e) = *(float4*)pv; *(float4*)(m + e) = *(float4*)mv; *(float4*)(v + e) = *(float4*)vv;
            if (ema) *(float4*)(ema + e) = *(float4*)ev;
        } else {
            for (int k = 0; k < 4; ++k) if (e + k < n) { p[e + k] = pv[k]; m[e + k] = mv[k]; v[e + k] = vv[k]; if (ema) ema[e + k] = ev[k]; }
        }
    }
}

// CA_NET.reparametrize (model.py:333-340): c = eps*exp(0.5*logvar) + mu
__global__ __launch_bounds__(256) void reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                          const float* __restrict__ eps, float* __restrict__ c, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) c[i] = eps[i] * __expf(0.5f * lv[i]) + mu[i];
}
__global__ __launch_bounds__(256) void reparam_bwd_kernel(const float* __restrict__ lv, const float* __restrict__ eps,
                                                          const float* __restrict__ dc, float* __restrict__ dmu,
                                                          float* __restrict__ dlv, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { dmu[i] = dc[i]; dlv[i] = dc[i] * eps[i] * 0.5f * __expf(0.5f * lv[i]); }
}

// device-resident step counter (hipGraph replays cannot change kernel arguments):
// state[0] = step count (as float, exact to 2^24), state[1] = step size, state[2] = 1/sqrt(1-b2^t)
__global__ void adam_prep_kernel(float* __restrict__ state, float lr, float beta1, float beta2, int eps_mode) {
    const double t = (double)state[0] + 1.0;
    state[0] = (float)t;
    const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
    state[1] = eps_mode == 0 ? (float)((double)lr / bc1) : (float)((double)lr * sqrt(bc2) / bc1);
    state[2] = (float)(1.0 / sqrt(bc2));
}

}  // namespace

namespace {
// ---------------------------------------------------------------------------------------------- channel concat / broadcast
// dst (N, sum C_i, HW) = the sources side by side along the channel axis; a source is addressed as
//   value(n, c, hw) = p[(n % rows) * sb + (n / rows) * sg + c * sc + (bcast ? 0 : hw)]
// which covers: a plain (N, C, HW) tensor (rows = N), a label / code broadcast over the plane (bcast; the reference's
// .repeat(1, 1, H, W): model.py:109-110, 400, 632-633, 664), one tensor repeated for every object (rows = N / G, sg = 0:
// c_code / the image in the object loops, model.py:395, 663) and the per-object slice label[:, idx] of a (B, G, C) tensor in the
// object-major batch of the batched pathways (rows = B, sb = G C, sg = C; model.py:396, 664).
struct CatSrcP { const float* p; long long sb, sg; int rows, C, bcast, c0; };
struct CatP { CatSrcP s[MOGAN_CAT_MAX]; int nsrc, N, Ctot, HW; };

template <int VEC>
__global__ __launch_bounds__(256) void concat_fwd_kernel(CatP P, float* __restrict__ dst, long long nvec) {
    const int hwv = P.HW / VEC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const int hw = (int)(i % hwv) * VEC;
        const long long r = i / hwv;
        const int c = (int)(r % P.Ctot), n = (int)(r / P.Ctot);
        int k = 0;
#pragma unroll
        for (int q = 1; q < MOGAN_CAT_MAX; ++q) k += (q < P.nsrc && c >= P.s[q].c0) ? 1 : 0;
        const CatSrcP S = P.s[k];
        const float* src = S.p + (long long)(n % S.rows) * S.sb + (long long)(n / S.rows) * S.sg +
                           (long long)(c - S.c0) * (S.bcast ? 1 : P.HW);
        float* d = dst + (r * P.HW + hw);
        if (VEC == 4) {
            float4 v;
            if (S.bcast) { const float t = src[0]; v = make_float4(t, t, t, t); }
            else v = *(const float4*)(src + hw);
            *(float4*)d = v;
        } else {
            d[0] = S.bcast ? src[0] : src[hw];
        }
    }
}

// the gradient: every source receives the sum of the destination gradient over everything that read it -- its channel slice
// (plain), summed over the plane (bcast), summed over the G repeats (rows < N, sg = 0).  One launch: the first `ncopy` work
// items are elements of the non-broadcast sources (one thread each), the rest are (row, channel) sums of the broadcast sources
// (one wave each).  dsrc[i] = NULL: no gradient wanted.
struct CatGradP { CatP c; float* d[MOGAN_CAT_MAX]; long long ebeg[MOGAN_CAT_MAX + 1]; long long rbeg[MOGAN_CAT_MAX + 1]; };

__global__ __launch_bounds__(256) void concat_bwd_kernel(CatGradP G, const float* __restrict__ ddst, unsigned copy_blocks) {
    const CatP& P = G.c;
    if (blockIdx.x < copy_blocks) {
        const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        if (i >= G.ebeg[P.nsrc]) return;
        int k = 0;
#pragma unroll
        for (int q = 1; q < MOGAN_CAT_MAX; ++q) k += (q < P.nsrc && i >= G.ebeg[q]) ? 1 : 0;
        const CatSrcP S = P.s[k];
        const long long e = i - G.ebeg[k];                      // element of the source's own (rows_total, C, HW) layout
        const int hw = (int)(e % P.HW);
        const long long rc = e / P.HW;
        const int c = (int)(rc % S.C);
        const long long row = rc / S.C;                         // row index in the source's storage order
        // destination batch indices n that read this row: sg = 0 and rows < N: n = row, row + rows, ...; else exactly one
        float t = 0.f;
        if (S.sg == 0) {
            for (int n = (int)row; n < P.N; n += S.rows) t += ddst[((long long)n * P.Ctot + S.c0 + c) * P.HW + hw];
        } else {
            // storage row = (n % rows) * (sb / sg) + n / rows  (object-major batch of a (rows, G, C) tensor)
            const int Gn = (int)(S.sb / S.sg);
            const int b = (int)(row / Gn), g = (int)(row % Gn);
            const int n = g * S.rows + b;
            t = ddst[((long long)n * P.Ctot + S.c0 + c) * P.HW + hw];
        }
        G.d[k][e] = t;
        return;
    }
    const long long item = (long long)(blockIdx.x - copy_blocks) * 4 + (threadIdx.x >> 6);
    if (item >= G.rbeg[P.nsrc]) return;
    int k = 0;
#pragma unroll
    for (int q = 1; q < MOGAN_CAT_MAX; ++q) k += (q < P.nsrc && item >= G.rbeg[q]) ? 1 : 0;
    const CatSrcP S = P.s[k];
    const long long e = item - G.rbeg[k];
    const int c = (int)(e % S.C);
    const long long row = e / S.C;
    const int lane = threadIdx.x & 63;
    float t = 0.f;
    auto plane_sum = [&](int n) {
        const float* q = ddst + ((long long)n * P.Ctot + S.c0 + c) * P.HW;
        for (int hw = lane; hw < P.HW; hw += 64) t += q[hw];
    };
    if (S.sg == 0) {
        for (int n = (int)row; n < P.N; n += S.rows) plane_sum(n);
    } else {
        const int Gn = (int)(S.sb / S.sg);
        plane_sum((int)(row % Gn) * S.rows + (int)(row / Gn));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane == 0) G.d[k][e] = t;
}

static int cat_params(CatP& P, const void* const* src, const int* C, const int* rows, const long long* sb, const long long* sg,
                      const int* bcast, int nsrc, int N, int HW) {
    if (nsrc <= 0 || nsrc > MOGAN_CAT_MAX || N <= 0 || HW <= 0) return MOGAN_ERR_SHAPE;
    P.nsrc = nsrc; P.N = N; P.HW = HW;
    int c0 = 0;
    for (int i = 0; i < MOGAN_CAT_MAX; ++i) {
        CatSrcP& S = P.s[i];
        if (i < nsrc) {
            if (C[i] <= 0 || rows[i] <= 0 || N % rows[i] != 0) return MOGAN_ERR_SHAPE;
            S.p = (const float*)src[i]; S.C = C[i]; S.rows = rows[i]; S.sb = sb[i]; S.sg = sg[i]; S.bcast = bcast[i] ? 1 : 0; S.c0 = c0;
            c0 += C[i];
        } else {
            S.p = nullptr; S.C = 0; S.rows = 1; S.sb = 0; S.sg = 0; S.bcast = 0; S.c0 = 0x7fffffff;
        }
    }
    P.Ctot = c0;
    return 0;
}

}  // namespace

extern "C" {

int mogan_abi_version(void) { return 1; }

int mogan_act_fwd(const float* x, float* y, int B, int C, int HW, int act, float slope, hipStream_t stream) {
    if (B <= 0 || C <= 0 || HW <= 0) return MOGAN_ERR_SHAPE;
    const long long n = (long long)B * C * HW;
    switch (act) {
        case MOGAN_ACT_RELU: hipLaunchKernelGGL((act_kernel<MOGAN_ACT_RELU, false>), dim3(nblk(n)), dim3(256), 0, stream, x, nullptr, y, n, slope); break;
        case MOGAN_ACT_LRELU: hipLaunchKernelGGL((act_kernel<MOGAN_ACT_LRELU, false>), dim3(nblk(n)), dim3(256), 0, stream, x, nullptr, y, n, slope); break;
        case MOGAN_ACT_TANH: hipLaunchKernelGGL((act_kernel<MOGAN_ACT_TANH, false>), dim3(nblk(n)), dim3(256), 0, stream, x, nullptr, y, n, slope); break;
        case MOGAN_ACT_SIGMOID: hipLaunchKernelGGL((act_kernel<MOGAN_ACT_SIGMOID, false>), dim3(nblk(n)), dim3(256), 0, stream, x, nullptr, y, n, slope); break;
        case MOGAN_ACT_GLU:
            if (C & 1) return MOGAN_ERR_SHAPE;
            hipLaunchKernelGGL((glu_kernel<false>), dim3(nblk(n / 2)), dim3(256), 0, stream, x, nullptr, y, n / 2, C / 2, (long long)HW);
            break;
        default: return MOGAN_ERR_SHAPE;
    }
    return ok_launch();
}

int mogan_act_bwd(const float* x, const float* dy, float* dx, int B, int C, int HW, int act, float slope,
                  hipStream_t stream) {
    if (B <= 0 || C <= 0 || HW <= 0) return MOGAN_ERR_SHAPE;
    const long long n = (long long)B * C * HW;
    switch (act) {
        case MOGAN_ACT_RELU: hipLaunchKernelGGL((act_kernel<MOGAN_ACT_RELU, true>), dim3(nblk(n)), dim3(256), 0, stream, x, dy, dx, n, slope); break;
        case MOGAN_ACT_LRELU: hipLaunchKernelGGL((act_kernel<MOGAN_ACT_LRELU, true>), dim3(nblk(n)), dim3(256), 0, stream, x, dy, dx, n, slope); break;
        case MOGAN_ACT_TANH: hipLaunchKernelGGL((act_kernel<MOGAN_ACT_TANH, true>), dim3(nblk(n)), dim3(256), 0, stream, x, dy, dx, n, slope); break;
        case MOGAN_ACT_SIGMOID: hipLaunchKernelGGL((act_kernel<MOGAN_ACT_SIGMOID, true>), dim3(nblk(n)), dim3(256), 0, stream, x, dy, dx, n, slope); break;
        case MOGAN_ACT_GLU:
            if (C & 1) return MOGAN_ERR_SHAPE;
            hipLaunchKernelGGL((glu_kernel<true>), dim3(nblk(n / 2)), dim3(256), 0, stream, x, dy, dx, n / 2, C / 2, (long long)HW);
            break;
        default: return MOGAN_ERR_SHAPE;
    }
    return ok_launch();
}

int mogan_bias_add(float* y, const float* bias, int rows, int C, int HW, hipStream_t stream) {
    if (rows <= 0 || C <= 0 || HW <= 0) return MOGAN_ERR_SHAPE;
    const long long n = (long long)rows * C * HW;
    hipLaunchKernelGGL(bias_add_kernel, dim3(nblk(n)), dim3(256), 0, stream, y, bias, n, C, (long long)HW);
    return ok_launch();
}
int mogan_bias_grad(const float* dy, float* dbias, int rows, int C, int HW, int accumulate, hipStream_t stream) {
    if (rows <= 0 || C <= 0 || HW <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(bias_grad_kernel, dim3(C), dim3(256), 0, stream, dy, dbias, rows, C, HW, accumulate);
    return ok_launch();
}
int mogan_add(const float* a, const float* b, float* y, long long n, hipStream_t stream) {
    if (n <= 0) return n == 0 ? 0 : MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(add_kernel, dim3(nblk(n)), dim3(256), 0, stream, a, b, y, n);
    return ok_launch();
}
// y[i] = ((x[i] + x[n+i]) + x[2n+i]) + ... over the G groups of n values (the object pathways' h_0 + h_1 + h_2, in the loop's
// order); its backward: every group receives dy
int mogan_group_sum(const float* x, float* y, long long n, int G, hipStream_t stream) {
    if (n <= 0 || G <= 0) return n == 0 ? 0 : MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(group_sum_kernel, dim3(nblk(n)), dim3(256), 0, stream, x, y, n, G);
    return ok_launch();
}
int mogan_group_bcast(const float* dy, float* dx, long long n, int G, hipStream_t stream) {
    if (n <= 0 || G <= 0) return n == 0 ? 0 : MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(group_bcast_kernel, dim3(nblk(n)), dim3(256), 0, stream, dy, dx, n, G);
    return ok_launch();
}
int mogan_scale(const float* a, float alpha, float* y, long long n, hipStream_t stream) {
    if (n <= 0) return n == 0 ? 0 : MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(scale_kernel, dim3(nblk(n)), dim3(256), 0, stream, a, alpha, y, n);
    return ok_launch();
}
int mogan_down2_sum(const float* du, float* dx, int planes, int H, int W, hipStream_t stream) {
    if (planes <= 0 || H <= 0 || W <= 0) return MOGAN_ERR_SHAPE;
    const long long n = (long long)planes * H * W;
    hipLaunchKernelGGL(down2_sum_kernel, dim3(nblk(n)), dim3(256), 0, stream, du, dx, n, H, W);
    return ok_launch();
}

int mogan_softmax_fwd(const float* x, float* y, const int32_t* lens, long long outer, int L, long long inner,
                      float scale, hipStream_t stream) {
    if (outer <= 0 || L <= 0 || inner <= 0) return MOGAN_ERR_SHAPE;
    const long long cols = outer * inner;
    hipLaunchKernelGGL((softmax_kernel<false>), dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, stream, x, nullptr, y, lens, cols, L, inner, scale);
    return ok_launch();
}
int mogan_softmax_bwd(const float* y, const float* dy, float* dx, const int32_t* lens, long long outer, int L,
                      long long inner, float scale, hipStream_t stream) {
    if (outer <= 0 || L <= 0 || inner <= 0) return MOGAN_ERR_SHAPE;
    const long long cols = outer * inner;
    hipLaunchKernelGGL((softmax_kernel<true>), dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, stream, y, dy, dx, lens, cols, L, inner, scale);
    return ok_launch();
}

int mogan_bce_fwd(const float* p, float target, float weight, float* loss, int n, int accumulate, hipStream_t stream) {
    if (n <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(1), dim3(256), 0, stream, p, target, weight, loss, n, accumulate);
    return ok_launch();
}
int mogan_bce_bwd(const float* p, float target, float weight, const float* gout, float* dp, int n, hipStream_t stream) {
    if (n <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, target, weight, gout, dp, n);
    return ok_launch();
}
int mogan_bce_logits_fwd(const float* x, float target, float weight, float* loss, int n, int accumulate,
                         hipStream_t stream) {
    if (n <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(bce_logits_fwd_kernel, dim3(1), dim3(256), 0, stream, x, target, weight, loss, n, accumulate);
    return ok_launch();
}
int mogan_bce_logits_bwd(const float* x, float target, float weight, const float* gout, float* dx, int n,
                         hipStream_t stream) {
    if (n <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(bce_logits_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, target, weight, gout, dx,
                       n);
    return ok_launch();
}
int mogan_kl_fwd(const float* mu, const float* logvar, float* loss, int n, hipStream_t stream) {
    if (n <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(kl_fwd_kernel, dim3(1), dim3(256), 0, stream, mu, logvar, loss, n);
    return ok_launch();
}
int mogan_kl_bwd(const float* mu, const float* logvar, const float* gout, float* dmu, float* dlogvar, int n,
                 hipStream_t stream) {
    if (n <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(kl_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, mu, logvar, gout, dmu, dlogvar, n);
    return ok_launch();
}

int mogan_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* c, int n, hipStream_t stream) {
    if (n <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, mu, logvar, eps, c, n);
    return ok_launch();
}
int mogan_reparam_bwd(const float* logvar, const float* eps, const float* dc, float* dmu, float* dlogvar, int n,
                      hipStream_t stream) {
    if (n <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, logvar, eps, dc, dmu, dlogvar, n);
    return ok_launch();
}

int mogan_relu_bwd(const float* z, const float* dz, float* dx, long long n, int accumulate, hipStream_t stream) {
    if (n <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(nblk(n)), dim3(256), 0, stream, z, dz, dx, n, accumulate);
    return ok_launch();
}

int mogan_concat_fwd(const void* const* src, const int* C, const int* rows, const long long* sb, const long long* sg,
                     const int* bcast, int nsrc, float* dst, int N, int HW, hipStream_t stream) {
    CatP P;
    const int rc = cat_params(P, src, C, rows, sb, sg, bcast, nsrc, N, HW);
    if (rc) return rc;
    bool v4 = (HW % 4) == 0 && (((uintptr_t)dst) & 15) == 0;
    for (int i = 0; i < nsrc && v4; ++i)
        if (!P.s[i].bcast && ((((uintptr_t)P.s[i].p) & 15) != 0 || (P.s[i].sb % 4) != 0 || (P.s[i].sg % 4) != 0)) v4 = false;
    const long long n = (long long)N * P.Ctot * HW;
    if (v4) hipLaunchKernelGGL((concat_fwd_kernel<4>), dim3(nblk(n / 4)), dim3(256), 0, stream, P, dst, n / 4);
    else hipLaunchKernelGGL((concat_fwd_kernel<1>), dim3(nblk(n)), dim3(256), 0, stream, P, dst, n);
    return ok_launch();
}

int mogan_concat_bwd(const float* ddst, void* const* dsrc, const int* C, const int* rows, const long long* sb, const long long* sg,
                     const int* bcast, int nsrc, int N, int HW, hipStream_t stream) {
    CatGradP G;
    const int rc = cat_params(G.c, (const void* const*)dsrc, C, rows, sb, sg, bcast, nsrc, N, HW);
    if (rc) return rc;
    long long ne = 0, nr = 0;
    for (int i = 0; i <= MOGAN_CAT_MAX; ++i) { G.ebeg[i] = 0; G.rbeg[i] = 0; }
    for (int i = 0; i < MOGAN_CAT_MAX; ++i) {
        G.d[i] = i < nsrc ? (float*)dsrc[i] : nullptr;
        const CatSrcP& S = G.c.s[i];
        // rows of the source's own storage: `rows` when it is repeated (sg = 0), N when every batch index has its own row
        const long long own = i < nsrc && G.d[i] ? (S.sg == 0 ? S.rows : N) : 0;
        G.ebeg[i] = ne; G.rbeg[i] = nr;
        if (i < nsrc && G.d[i]) { if (S.bcast) nr += own * S.C; else ne += own * S.C * HW; }
    }
    for (int i = nsrc; i <= MOGAN_CAT_MAX; ++i) { G.ebeg[i] = ne; G.rbeg[i] = nr; }
    // a source without gradient has an empty range: the `i >= beg[q]` search must skip it -> give it the next begin
    if (ne + nr == 0) return 0;
    if (ne >= (1ll << 40)) return MOGAN_ERR_SHAPE;
    const unsigned cb = (unsigned)((ne + 255) / 256), rb = (unsigned)((nr + 3) / 4);
    hipLaunchKernelGGL(concat_bwd_kernel, dim3(cb + rb), dim3(256), 0, stream, G, ddst, cb);
    return ok_launch();
}

int mogan_copy_strided(const float* src, long long src_bstride, float* dst, long long dst_bstride, int B, long long n,
                       hipStream_t stream) {
    if (B <= 0 || n <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(copy_strided_kernel, dim3(nblk((long long)B * n)), dim3(256), 0, stream, src, src_bstride, dst,
                       dst_bstride, n, (long long)B * n);
    return ok_launch();
}

int mogan_maxpool_fwd_ex(const float* x, float* y, long long y_bstride, uint8_t* idx, int B, int C, int H, int W, int k,
                         int s, hipStream_t stream) {
    if (B <= 0 || C <= 0 || k <= 0 || s <= 0 || H < k || W < k || k * k > 255) return MOGAN_ERR_SHAPE;
    const int OH = (H - k) / s + 1, OW = (W - k) / s + 1;
    if (y_bstride < 0) y_bstride = (long long)C * OH * OW;
    const long long n = (long long)B * C * OH * OW;
    POOL_LAUNCH(maxpool_fwd_kernel, n, dim3(nblk(n)), dim3(256), 0, stream, x, y, idx, n, H, W, OH, OW, k, s, C,
                       y_bstride);
    return ok_launch();
}
int mogan_maxpool_fwd(const float* x, float* y, uint8_t* idx, int planes, int H, int W, int k, int s,
                      hipStream_t stream) {
    return mogan_maxpool_fwd_ex(x, y, -1, idx, 1, planes, H, W, k, s, stream);
}
int mogan_maxpool_bwd_ex(const uint8_t* idx, const float* dy, long long dy_bstride, float* dx, const float* relu_of,
                         int accumulate, int B, int C, int H, int W, int k, int s, hipStream_t stream) {
    if (B <= 0 || C <= 0 || k <= 0 || s <= 0 || H < k || W < k || k > 2 * s) return MOGAN_ERR_SHAPE;      // (k <= 2 s: see the kernel)
    const int OH = (H - k) / s + 1, OW = (W - k) / s + 1;
    if (dy_bstride < 0) dy_bstride = (long long)C * OH * OW;
    const long long n = (long long)B * C * H * W;
    POOL_LAUNCH(maxpool_bwd_kernel, n, dim3(nblk(n)), dim3(256), 0, stream, idx, dy, dx, n, H, W, OH, OW, k, s, C,
                       dy_bstride, relu_of, accumulate);
    return ok_launch();
}
int mogan_maxpool_bwd(const uint8_t* idx, const float* dy, float* dx, int planes, int H, int W, int k, int s,
                      hipStream_t stream) {
    return mogan_maxpool_bwd_ex(idx, dy, -1, dx, nullptr, 0, 1, planes, H, W, k, s, stream);
}
int mogan_avgpool_fwd(const float* x, float* y, int planes, int H, int W, int k, int s, int pad, hipStream_t stream) {
    const int OH = (H + 2 * pad - k) / s + 1, OW = (W + 2 * pad - k) / s + 1;
    if (planes <= 0 || OH <= 0 || OW <= 0) return MOGAN_ERR_SHAPE;
    const long long n = (long long)planes * OH * OW;
    POOL_LAUNCH(avgpool_fwd_kernel, n, dim3(nblk(n)), dim3(256), 0, stream, x, y, n, H, W, OH, OW, k, s, pad);
    return ok_launch();
}
int mogan_avgpool_bwd_ex(const float* dy, float* dx, const float* relu_of, int accumulate, int planes, int H, int W, int k,
                         int s, int pad, hipStream_t stream) {
    if (planes <= 0 || k <= 0 || s <= 0) return MOGAN_ERR_SHAPE;
    const int OH = (H + 2 * pad - k) / s + 1, OW = (W + 2 * pad - k) / s + 1;
    const long long n = (long long)planes * H * W;
    POOL_LAUNCH(avgpool_bwd_kernel, n, dim3(nblk(n)), dim3(256), 0, stream, dy, dx, n, H, W, OH, OW, k, s, pad, relu_of,
                       accumulate);
    return ok_launch();
}
int mogan_avgpool_bwd(const float* dy, float* dx, int planes, int H, int W, int k, int s, int pad,
                      hipStream_t stream) {
    return mogan_avgpool_bwd_ex(dy, dx, nullptr, 0, planes, H, W, k, s, pad, stream);
}
int mogan_bilinear_fwd(const float* x, float* y, int planes, int H, int W, int OH, int OW, hipStream_t stream) {
    if (planes <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return MOGAN_ERR_SHAPE;
    const long long n = (long long)planes * OH * OW;
    POOL_LAUNCH(bilinear_fwd_kernel, n, dim3(nblk(n)), dim3(256), 0, stream, x, y, n, H, W, OH, OW);
    return ok_launch();
}
int mogan_bilinear_bwd(const float* dy, float* dx, int planes, int H, int W, int OH, int OW, hipStream_t stream) {
    if (planes <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return MOGAN_ERR_SHAPE;
    const long long n = (long long)planes * H * W;        // one thread per input element; every element is written
    POOL_LAUNCH(bilinear_bwd_kernel, n, dim3(nblk(n)), dim3(256), 0, stream, dy, dx, n, H, W, OH, OW);
    return ok_launch();
}

int mogan_adam_step(float* p, const float* g, float* m, float* v, float* ema, long long n, float lr, float beta1,
                    float beta2, float eps, int step, float* dev_state, int eps_mode, float grad_scale,
                    float ema_decay, hipStream_t stream) {
    if (n <= 0 || (!dev_state && step < 1)) return MOGAN_ERR_SHAPE;
    if (dev_state) { hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(1), 0, stream, dev_state, lr, beta1, beta2, eps_mode); step = 1; }
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = eps_mode == 0 ? (float)(lr / bc1) : (float)(lr * sqrt(bc2) / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const long long n4 = (n + 3) / 4;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) != 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(adam_kernel, dim3(nblk(n4)), dim3(256), 0, stream, p, g, m, v, ema, n4, n, beta1, beta2, eps,
                       step_size, inv_sqrt_bc2, eps_mode, grad_scale, ema_decay, (const float*)dev_state);
    return ok_launch();
}

}  // extern "C"
