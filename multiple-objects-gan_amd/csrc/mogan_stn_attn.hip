// mogan_stn_attn.hip -- the two ops that are specific to this repo's generator/discriminators:
//
//  * stn: the "object pathway" spatial transformer (code/coco/attngan/model.py:17-21):
//    affine_grid(theta) + grid_sample(bilinear, zero padding).  bbox = -1 (absent object) gives
//    theta_inv = [[-1,0,-4],[0,-1,-4]] whose grid lies wholly outside [-1,1] -> exactly 0.
//    HBM-bound gather; one thread per output pixel loops over the channels (the 4 taps and weights
//    depend only on (b,oy,ox)), so consecutive lanes read consecutive x of the same plane.
//  * GlobalAttentionGeneral core (GlobalAttention.py:96-121): per query pixel q, scores over the
//    T <= 32 words (K = idf <= 128), masked softmax, weighted context.  One thread per query; the
//    (idf x T) word projection sits in LDS and is read as broadcasts; h is read and wc/attn written
//    with q contiguous across lanes.  T is tiny: no MFMA.
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

struct Taps { int x0, y0; float wx1, wy1; bool vx0, vx1, vy0, vy1; };

__device__ __forceinline__ Taps stn_taps(const float* __restrict__ th, int oy, int ox, int Hin, int Win, int Hout,
                                         int Wout, int ac) {
    float xn, yn;
    if (ac) {
        xn = Wout > 1 ? 2.f * ox / (float)(Wout - 1) - 1.f : 0.f;
        yn = Hout > 1 ? 2.f * oy / (float)(Hout - 1) - 1.f : 0.f;
    } else {
        xn = (2.f * ox + 1.f) / (float)Wout - 1.f;
        yn = (2.f * oy + 1.f) / (float)Hout - 1.f;
    }
    const float gx = th[0] * xn + th[1] * yn + th[2];
    const float gy = th[3] * xn + th[4] * yn + th[5];
    float ix, iy;
    if (ac) { ix = (gx + 1.f) * 0.5f * (float)(Win - 1); iy = (gy + 1.f) * 0.5f * (float)(Hin - 1); }
    else { ix = ((gx + 1.f) * (float)Win - 1.f) * 0.5f; iy = ((gy + 1.f) * (float)Hin - 1.f) * 0.5f; }
    Taps t;
    const float fx = floorf(ix), fy = floorf(iy);
    // clamp before the int conversion so wild thetas (absent objects) cannot overflow
    t.x0 = (int)fminf(fmaxf(fx, -2.f), (float)Win + 1.f);
    t.y0 = (int)fminf(fmaxf(fy, -2.f), (float)Hin + 1.f);
    t.wx1 = ix - fx; t.wy1 = iy - fy;
    const bool inx = (fx >= -1.f && fx <= (float)Win), iny = (fy >= -1.f && fy <= (float)Hin);
    t.vx0 = inx && t.x0 >= 0 && t.x0 < Win; t.vx1 = inx && t.x0 + 1 >= 0 && t.x0 + 1 < Win;
    t.vy0 = iny && t.y0 >= 0 && t.y0 < Hin; t.vy1 = iny && t.y0 + 1 >= 0 && t.y0 + 1 < Hin;
    return t;
}

// shared / constant sources (mogan_stn_*_ex): output sample b reads image b % xB; PLANE: x is (xB, C), constant over the plane
template <bool PLANE>
__global__ __launch_bounds__(256) void stn_fwd_ex_kernel(const float* __restrict__ x, const float* __restrict__ theta,
                                                         float* __restrict__ y, int C, int Hin, int Win, int Hout,
                                                         int Wout, int ac, int cchunk, int xB, int tG) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= Hout * Wout) return;
    const int b = blockIdx.z, bx = b % xB, c0 = blockIdx.y * cchunk, c1 = min(C, c0 + cchunk);
    const int oy = pix / Wout, ox = pix - oy * Wout;
    // tG > 0: theta is stored (B', tG, 2, 3) and sample b = g B' + b' (object-major batch) uses theta[b'][g]
    const int nb = tG > 0 ? (int)gridDim.z / tG : 1;
    const int tb = tG > 0 ? (b % nb) * tG + b / nb : b;
    const Taps t = stn_taps(theta + tb * 6, oy, ox, Hin, Win, Hout, Wout, ac);
    const float w00 = (1.f - t.wx1) * (1.f - t.wy1), w01 = t.wx1 * (1.f - t.wy1);
    const float w10 = (1.f - t.wx1) * t.wy1, w11 = t.wx1 * t.wy1;
    const bool v00 = t.vx0 && t.vy0, v01 = t.vx1 && t.vy0, v10 = t.vx0 && t.vy1, v11 = t.vx1 && t.vy1;
    const int o00 = t.y0 * Win + t.x0;
    for (int c = c0; c < c1; ++c) {
        float acc = 0.f;
        if (PLANE) {                       // the same four products in the same order as on the materialised plane
            const float v = x[(size_t)bx * C + c];
            if (v00) acc += v * w00;
            if (v01) acc += v * w01;
            if (v10) acc += v * w10;
            if (v11) acc += v * w11;
        } else {
            const float* px = x + ((size_t)bx * C + c) * Hin * Win;
            if (v00) acc += px[o00] * w00;
            if (v01) acc += px[o00 + 1] * w01;
            if (v10) acc += px[o00 + Win] * w10;
            if (v11) acc += px[o00 + Win + 1] * w11;
        }
        y[((size_t)b * C + c) * Hout * Wout + pix] = acc;
    }
}

template <bool PLANE>
__global__ __launch_bounds__(256) void stn_bwd_ex_kernel(const float* __restrict__ dy, const float* __restrict__ theta,
                                                         float* __restrict__ dx, int C, int Hin, int Win, int Hout,
                                                         int Wout, int ac, int cchunk, int xB, int tG) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const bool in = pix < Hout * Wout;
    const int b = blockIdx.z, bx = b % xB, c0 = blockIdx.y * cchunk, c1 = min(C, c0 + cchunk);
    const int oy = in ? pix / Wout : 0, ox = in ? pix - oy * Wout : 0;
    const int nb = tG > 0 ? (int)gridDim.z / tG : 1;
    const int tb = tG > 0 ? (b % nb) * tG + b / nb : b;
    const Taps t = stn_taps(theta + tb * 6, oy, ox, Hin, Win, Hout, Wout, ac);
    const float w00 = (1.f - t.wx1) * (1.f - t.wy1), w01 = t.wx1 * (1.f - t.wy1);
    const float w10 = (1.f - t.wx1) * t.wy1, w11 = t.wx1 * t.wy1;
    const bool v00 = in && t.vx0 && t.vy0, v01 = in && t.vx1 && t.vy0, v10 = in && t.vx0 && t.vy1, v11 = in && t.vx1 && t.vy1;
    const int o00 = t.y0 * Win + t.x0;
    if (PLANE) {
        // d x[b, c] = sum over the output pixels of dy * (sum of the in-range tap weights): per wave a butterfly, one atomic
        const float wsum = (v00 ? w00 : 0.f) + (v01 ? w01 : 0.f) + (v10 ? w10 : 0.f) + (v11 ? w11 : 0.f);
        for (int c = c0; c < c1; ++c) {
            float g = in ? dy[((size_t)b * C + c) * Hout * Wout + pix] * wsum : 0.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o, 64);
            if ((threadIdx.x & 63) == 0 && g != 0.f) atomicAdd(dx + (size_t)bx * C + c, g);
        }
        return;
    }
    if (!(v00 || v01 || v10 || v11)) return;
    for (int c = c0; c < c1; ++c) {
        float* px = dx + ((size_t)bx * C + c) * Hin * Win;
        const float g = dy[((size_t)b * C + c) * Hout * Wout + pix];
        if (v00) atomicAdd(px + o00, g * w00);
        if (v01) atomicAdd(px + o00 + 1, g * w01);
        if (v10) atomicAdd(px + o00 + Win, g * w10);
        if (v11) atomicAdd(px + o00 + Win + 1, g * w11);
    }
}

__global__ __launch_bounds__(256) void stn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ theta,
                                                      float* __restrict__ y, int C, int Hin, int Win, int Hout,
                                                      int Wout, int ac, int cchunk) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= Hout * Wout) return;
    const int b = blockIdx.z, c0 = blockIdx.y * cchunk, c1 = min(C, c0 + cchunk);
    const int oy = pix / Wout, ox = pix - oy * Wout;
    const Taps t = stn_taps(theta + b * 6, oy, ox, Hin, Win, Hout, Wout, ac);
    const float w00 = (1.f - t.wx1) * (1.f - t.wy1), w01 = t.wx1 * (1.f - t.wy1);
    const float w10 = (1.f - t.wx1) * t.wy1, w11 = t.wx1 * t.wy1;
    const bool v00 = t.vx0 && t.vy0, v01 = t.vx1 && t.vy0, v10 = t.vx0 && t.vy1, v11 = t.vx1 && t.vy1;
    const int o00 = t.y0 * Win + t.x0;
    for (int c = c0; c < c1; ++c) {
        const float* px = x + ((size_t)b * C + c) * Hin * Win;
        float acc = 0.f;
        if (v00) acc += px[o00] * w00;
        if (v01) acc += px[o00 + 1] * w01;
        if (v10) acc += px[o00 + Win] * w10;
        if (v11) acc += px[o00 + Win + 1] * w11;
        y[((size_t)b * C + c) * Hout * Wout + pix] = acc;
    }
}

__global__ __launch_bounds__(256) void stn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ theta,
                                                      float* __restrict__ dx, int C, int Hin, int Win, int Hout,
                                                      int Wout, int ac, int cchunk) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= Hout * Wout) return;
    const int b = blockIdx.z, c0 = blockIdx.y * cchunk, c1 = min(C, c0 + cchunk);
    const int oy = pix / Wout, ox = pix - oy * Wout;
    const Taps t = stn_taps(theta + b * 6, oy, ox, Hin, Win, Hout, Wout, ac);
    const float w00 = (1.f - t.wx1) * (1.f - t.wy1), w01 = t.wx1 * (1.f - t.wy1);
    const float w10 = (1.f - t.wx1) * t.wy1, w11 = t.wx1 * t.wy1;
    const bool v00 = t.vx0 && t.vy0, v01 = t.vx1 && t.vy0, v10 = t.vx0 && t.vy1, v11 = t.vx1 && t.vy1;
    if (!(v00 || v01 || v10 || v11)) return;
    const int o00 = t.y0 * Win + t.x0;
    for (int c = c0; c < c1; ++c) {
        float* px = dx + ((size_t)b * C + c) * Hin * Win;
        const float g = dy[((size_t)b * C + c) * Hout * Wout + pix];
        if (v00) atomicAdd(px + o00, g * w00);
        if (v01) atomicAdd(px + o00 + 1, g * w01);
        if (v10) atomicAdd(px + o00 + Win, g * w10);
        if (v11) atomicAdd(px + o00 + Win + 1, g * w11);
    }
}

__global__ void bbox_to_theta_kernel(const float* __restrict__ bbox, float* __restrict__ th, float* __restrict__ thi,
                                     int N) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float x = bbox[i * 4], y = bbox[i * 4 + 1], w = bbox[i * 4 + 2], h = bbox[i * 4 + 3];
    // same operation order as miscc/utils.py:16-49, no fma contraction
    const float cx = __fadd_rn(x, __fmul_rn(0.5f, w)), cy = __fadd_rn(y, __fmul_rn(0.5f, h));
    float* t = th + i * 6;
    t[0] = w; t[1] = 0.f; t[2] = __fmul_rn(2.f, __fsub_rn(cx, 0.5f));
    t[3] = 0.f; t[4] = h; t[5] = __fmul_rn(2.f, __fsub_rn(cy, 0.5f));
    const float sx = __fdiv_rn(1.0f, w), sy = __fdiv_rn(1.0f, h);
    float* u = thi + i * 6;
    u[0] = sx; u[1] = 0.f; u[2] = __fmul_rn(__fmul_rn(2.f, sx), __fsub_rn(0.5f, cx));
    u[3] = 0.f; u[4] = sy; u[5] = __fmul_rn(__fmul_rn(2.f, sy), __fsub_rn(0.5f, cy));
}

// ------------------------------------------------------------------------------------ attention
// Round 4: the word projections sit in LDS zero-padded to TMAX slots per channel and every multiply-add loop runs over all
// TMAX slots -- the `t < T` tests of the earlier form were uniform branches inside the unrolled loops, which left one global
// load and a dozen FMAs per basic block with nothing in flight between them (61 us per launch against a 20 us HBM bound); the
// channel loops now issue eight loads before their multiplies.
template <int TMAX>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ h, const float* __restrict__ src,
                                                       const uint8_t* __restrict__ mask, float* __restrict__ wc,
                                                       float* __restrict__ attn, int B, int idf, int Q, int T,
                                                       int mask_mode) {
    extern __shared__ __attribute__((aligned(16))) float s_src[];     // [idf][TMAX], slots >= T are zero
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < idf * TMAX; i += 256) {
        const int c = i / TMAX, t = i - c * TMAX;
        s_src[i] = t < T ? src[((size_t)b * idf + c) * T + t] : 0.f;
    }
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    float s[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) s[t] = 0.f;
    const float* ph = h + (size_t)b * idf * Q + q;
    int c = 0;
    for (; c + 8 <= idf; c += 8) {
        float hv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) hv[u] = ph[(size_t)(c + u) * Q];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < TMAX; ++t) s[t] = fmaf(hv[u], s_src[(c + u) * TMAX + t], s[t]);
    }
    for (; c < idf; ++c) {
        const float hv = ph[(size_t)c * Q];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) s[t] = fmaf(hv, s_src[c * TMAX + t], s[t]);
    }
    // reference: mask.repeat(queryL,1) laid over rows b*Q+q  ->  row r uses mask[r % B]  
    const uint8_t* pm = nullptr;
    if (mask) pm = mask + (size_t)(mask_mode == 0 ? (int)(((long long)b * Q + q) % B) : b) * T;
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        bool off = t >= T;
        if (t < T && pm != nullptr && pm[t] != 0) off = true;
        if (off) s[t] = -INFINITY;
        m = fmaxf(m, s[t]);
    }
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) { s[t] = __expf(s[t] - m); sum += s[t]; }     // (slots >= T: exp(-inf) = 0)
    const float inv = 1.f / sum;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) { s[t] *= inv; if (t < T) attn[((size_t)b * T + t) * Q + q] = s[t]; }
    float* pw = wc + (size_t)b * idf * Q + q;
    for (int c2 = 0; c2 < idf; ++c2) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) a = fmaf(s_src[c2 * TMAX + t], s[t], a);
        pw[(size_t)c2 * Q] = a;
    }
}

// dp[t] = sum_c dwc[c,q]*src[c,t] (+dattn[t,q]); ds = p*(dp - sum p*dp); dh[c,q] = sum_t ds[t]*src[c,t]
template <int TMAX>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ src, const float* __restrict__ attn,
                                                       const float* __restrict__ dwc, const float* __restrict__ dattn,
                                                       float* __restrict__ dh, float* __restrict__ dscore, int idf,
                                                       int Q, int T) {
    extern __shared__ __attribute__((aligned(16))) float s_src[];     // [idf][TMAX], slots >= T are zero
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < idf * TMAX; i += 256) {
        const int c = i / TMAX, t = i - c * TMAX;
        s_src[i] = t < T ? src[((size_t)b * idf + c) * T + t] : 0.f;
    }
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    float dp[TMAX], p[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        p[t] = t < T ? attn[((size_t)b * T + t) * Q + q] : 0.f;
        dp[t] = (t < T && dattn) ? dattn[((size_t)b * T + t) * Q + q] : 0.f;
    }
    const float* pd = dwc + (size_t)b * idf * Q + q;
    int c = 0;
    for (; c + 8 <= idf; c += 8) {
        float gv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) gv[u] = pd[(size_t)(c + u) * Q];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < TMAX; ++t) dp[t] = fmaf(gv[u], s_src[(c + u) * TMAX + t], dp[t]);
    }
    for (; c < idf; ++c) {
        const float g = pd[(size_t)c * Q];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) dp[t] = fmaf(g, s_src[c * TMAX + t], dp[t]);
    }
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) dot = fmaf(p[t], dp[t], dot);           // (p = 0 in the slots >= T)
#pragma unroll
    for (int t = 0; t < TMAX; ++t) { dp[t] = p[t] * (dp[t] - dot); if (t < T) dscore[((size_t)b * T + t) * Q + q] = dp[t]; }
    float* po = dh + (size_t)b * idf * Q + q;
    for (int c2 = 0; c2 < idf; ++c2) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) a = fmaf(dp[t], s_src[c2 * TMAX + t], a);
        po[(size_t)c2 * Q] = a;
    }
}

}  // namespace

extern "C" {

int mogan_stn_fwd(const float* x, const float* theta, float* y, int B, int C, int Hin, int Win, int Hout, int Wout,
                  int align_corners, hipStream_t stream) {
    if (B <= 0 || C <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || B > 65535) return MOGAN_ERR_SHAPE;
    const int pb = (Hout * Wout + 255) / 256;
    int csplit = (1024 + pb * B - 1) / (pb * B); if (csplit > C) csplit = C; if (csplit < 1) csplit = 1;
    const int cchunk = (C + csplit - 1) / csplit; csplit = (C + cchunk - 1) / cchunk;
    hipLaunchKernelGGL(stn_fwd_kernel, dim3(pb, csplit, B), dim3(256), 0, stream, x, theta, y, C, Hin, Win, Hout, Wout,
                       align_corners, cchunk);
    return ok_launch();
}

int mogan_stn_bwd(const float* dy, const float* theta, float* dx, int B, int C, int Hin, int Win, int Hout,
                  int Wout, int align_corners, hipStream_t stream) {
    if (B <= 0 || C <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || B > 65535) return MOGAN_ERR_SHAPE;
    zero_fill(dx, (size_t)B * C * Hin * Win, stream);
    const int pb = (Hout * Wout + 255) / 256;
    int csplit = (1024 + pb * B - 1) / (pb * B); if (csplit > C) csplit = C; if (csplit < 1) csplit = 1;
    const int cchunk = (C + csplit - 1) / csplit; csplit = (C + cchunk - 1) / cchunk;
    hipLaunchKernelGGL(stn_bwd_kernel, dim3(pb, csplit, B), dim3(256), 0, stream, dy, theta, dx, C, Hin, Win, Hout, Wout,
                       align_corners, cchunk);
    return ok_launch();
}

int mogan_stn_fwd_ex(const float* x, const float* theta, float* y, int B, int C, int Hin, int Win, int Hout, int Wout,
                     int align_corners, int xB, int x_plane, int theta_G, hipStream_t stream) {
    if (B <= 0 || C <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || B > 65535 || xB <= 0 || B % xB != 0 ||
        theta_G < 0 || (theta_G > 0 && B % theta_G != 0))
        return MOGAN_ERR_SHAPE;
    const int pb = (Hout * Wout + 255) / 256;
    int csplit = (1024 + pb * B - 1) / (pb * B); if (csplit > C) csplit = C; if (csplit < 1) csplit = 1;
    const int cchunk = (C + csplit - 1) / csplit; csplit = (C + cchunk - 1) / cchunk;
    if (x_plane) hipLaunchKernelGGL((stn_fwd_ex_kernel<true>), dim3(pb, csplit, B), dim3(256), 0, stream, x, theta, y, C, Hin, Win,
                                    Hout, Wout, align_corners, cchunk, xB, theta_G);
    else hipLaunchKernelGGL((stn_fwd_ex_kernel<false>), dim3(pb, csplit, B), dim3(256), 0, stream, x, theta, y, C, Hin, Win,
                            Hout, Wout, align_corners, cchunk, xB, theta_G);
    return ok_launch();
}

int mogan_stn_bwd_ex(const float* dy, const float* theta, float* dx, int B, int C, int Hin, int Win, int Hout, int Wout,
                     int align_corners, int xB, int x_plane, int theta_G, hipStream_t stream) {
    if (B <= 0 || C <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || B > 65535 || xB <= 0 || B % xB != 0 ||
        theta_G < 0 || (theta_G > 0 && B % theta_G != 0))
        return MOGAN_ERR_SHAPE;
    zero_fill(dx, (size_t)xB * C * (x_plane ? 1 : (size_t)Hin * Win), stream);
    const int pb = (Hout * Wout + 255) / 256;
    int csplit = (1024 + pb * B - 1) / (pb * B); if (csplit > C) csplit = C; if (csplit < 1) csplit = 1;
    const int cchunk = (C + csplit - 1) / csplit; csplit = (C + cchunk - 1) / cchunk;
    if (x_plane) hipLaunchKernelGGL((stn_bwd_ex_kernel<true>), dim3(pb, csplit, B), dim3(256), 0, stream, dy, theta, dx, C, Hin, Win,
                                    Hout, Wout, align_corners, cchunk, xB, theta_G);
    else hipLaunchKernelGGL((stn_bwd_ex_kernel<false>), dim3(pb, csplit, B), dim3(256), 0, stream, dy, theta, dx, C, Hin, Win,
                            Hout, Wout, align_corners, cchunk, xB, theta_G);
    return ok_launch();
}

int mogan_bbox_to_theta(const float* bbox, float* theta, float* theta_inv, int N, hipStream_t stream) {
    if (N <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(bbox_to_theta_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, bbox, theta, theta_inv, N);
    return ok_launch();
}

int mogan_attn_fwd(const float* h, const float* src, const uint8_t* mask, float* wc, float* attn, int B, int idf,
                   int Q, int T, int mask_mode, hipStream_t stream) {
    if (B <= 0 || idf <= 0 || idf > 128 || Q <= 0 || T <= 0 || T > 32 || B > 65535) return MOGAN_ERR_SHAPE;
    dim3 grid((Q + 255) / 256, B);
    const size_t sh = (size_t)idf * (T <= 8 ? 8 : T <= 16 ? 16 : 32) * sizeof(float);
    if (T <= 8) hipLaunchKernelGGL((attn_fwd_kernel<8>), grid, dim3(256), sh, stream, h, src, mask, wc, attn, B, idf, Q, T, mask_mode);
    else if (T <= 16) hipLaunchKernelGGL((attn_fwd_kernel<16>), grid, dim3(256), sh, stream, h, src, mask, wc, attn, B, idf, Q, T, mask_mode);
    else hipLaunchKernelGGL((attn_fwd_kernel<32>), grid, dim3(256), sh, stream, h, src, mask, wc, attn, B, idf, Q, T, mask_mode);
    return ok_launch();
}

int mogan_attn_bwd(const float* src, const float* attn, const float* dwc, const float* dattn, float* dh,
                   float* dscore, int B, int idf, int Q, int T, hipStream_t stream) {
    if (B <= 0 || idf <= 0 || idf > 128 || Q <= 0 || T <= 0 || T > 32 || B > 65535) return MOGAN_ERR_SHAPE;
    dim3 grid((Q + 255) / 256, B);
    const size_t sh = (size_t)idf * (T <= 8 ? 8 : T <= 16 ? 16 : 32) * sizeof(float);
    if (T <= 8) hipLaunchKernelGGL((attn_bwd_kernel<8>), grid, dim3(256), sh, stream, src, attn, dwc, dattn, dh, dscore, idf, Q, T);
    else if (T <= 16) hipLaunchKernelGGL((attn_bwd_kernel<16>), grid, dim3(256), sh, stream, src, attn, dwc, dattn, dh, dscore, idf, Q, T);
    else hipLaunchKernelGGL((attn_bwd_kernel<32>), grid, dim3(256), sh, stream, src, attn, dwc, dattn, dh, dscore, idf, Q, T);
    return ok_launch();
}

}  // extern "C"
