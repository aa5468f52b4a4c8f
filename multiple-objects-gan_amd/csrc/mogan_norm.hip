// mogan_norm.hip -- BatchNorm (training statistics) fused with GLU / LeakyReLU / ReLU / residual-add,
// forward and backward, plus the eval-mode per-channel affine used by the frozen Inception trunk.
//
// All of it is HBM-bound.  Layout x (B,C,HW), NCHW.  Kernels:
//   bn_partial   grid (C, YS): one block reduces a (batch-range x HW-range) slab of one channel with
//                float4 loads and fp64 accumulators (E[x^2]-mean^2 is then safe), wave64 shuffle
//                reduction, one (sum,sumsq) pair per block into the workspace;
//   bn_finalize  one wave per channel: fixed-order sum of the partials -> mean, invstd, running stats;
//   bn_act_fwd   y = act(gamma*xhat+beta) (+res): one pass, float4;  GLU reads channel c and c+C/2;
//   bn_bwd_partial / bn_bwd_finalize / bn_bwd_apply: the two-reduction BN backward with the activation
//                backward recomputed from x (nothing but x, mean, invstd is saved by the forward).
// Replaces nn.BatchNorm1d/2d + GLU / nn.LeakyReLU / nn.ReLU at code/coco/attngan/model.py:48-81,
// 96-101,364-373,575-611,667-680.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mogan_hip.h"

namespace {

__global__ __launch_bounds__(256) void affine_relu_bwd_out_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                                  const float* __restrict__ scale, float* __restrict__ dx,
                                                                  long long n, int C, int HW) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)((i / HW) % C);
    dx[i] = y[i] > 0.f ? dy[i] * scale[c] : 0.f;
}


__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block (256 threads) reduction of NV doubles; result valid in thread 0
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* sh /* [4*NV] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) sh[wave * NV + i] = v[i];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = sh[i] + sh[NV + i] + sh[2 * NV + i] + sh[3 * NV + i];
}

struct Split { int bs, hs, bper, hper; };   // YS = bs*hs blocks per channel

static Split make_split(int B, int C, int HW) {
    Split s;
    long long want = (1536 + C - 1) / C;                     // ~6 blocks per CU overall
    if (want < 1) want = 1;
    s.bs = (int)(want < B ? want : B);
    long long rest = (want + s.bs - 1) / s.bs;
    long long hmax = (HW + 2047) / 2048; if (hmax < 1) hmax = 1;
    s.hs = (int)(rest < hmax ? rest : hmax);
    s.bper = (B + s.bs - 1) / s.bs; s.bs = (B + s.bper - 1) / s.bper;
    s.hper = (((HW + s.hs - 1) / s.hs) + 3) & ~3; s.hs = (HW + s.hper - 1) / s.hper;
    return s;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

// -------------------------------------------------------------------------------- forward stats
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, int B, int C, int HW,
                                                         Split sp, double* __restrict__ part) {
    __shared__ double sh[8];
    const int c = blockIdx.x, ys = blockIdx.y;
    const int bi = ys / sp.hs, hi = ys % sp.hs;
    const int b0 = bi * sp.bper, b1 = min(B, b0 + sp.bper);
    const int h0 = hi * sp.hper, h1 = min(HW, h0 + sp.hper);
    double acc[2] = {0.0, 0.0};
    const bool vec = ((HW & 3) == 0);
    for (int b = b0; b < b1; ++b) {
        const float* px = x + ((size_t)b * C + c) * HW;
        if (vec) {
            for (int i = h0 + threadIdx.x * 4; i < h1; i += 1024) {
                const float4 v = *(const float4*)(px + i);
                acc[0] += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
                acc[1] += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
            }
        } else {
            for (int i = h0 + threadIdx.x; i < h1; i += 256) {
                const double v = px[i];
                acc[0] += v; acc[1] += v * v;
            }
        }
    }
    block_sum<2>(acc, sh);
    if (threadIdx.x == 0) {
        part[((size_t)c * gridDim.y + ys) * 2 + 0] = acc[0];
        part[((size_t)c * gridDim.y + ys) * 2 + 1] = acc[1];
    }
}

// HW == 1 (BatchNorm1d on (B,C)): one thread per channel, coalesced across channels
__global__ __launch_bounds__(256) void bn1d_partial_kernel(const float* __restrict__ x, int B, int C,
                                                           double* __restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0, q = 0;
    for (int b = 0; b < B; ++b) { const double v = x[(size_t)b * C + c]; s += v; q += v * v; }
    part[(size_t)c * 2] = s; part[(size_t)c * 2 + 1] = q;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ part, int C, int YS, double n,
                                                          float eps, float momentum, float* __restrict__ mean,
                                                          float* __restrict__ invstd, float* __restrict__ rmean,
                                                          float* __restrict__ rvar) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0, q = 0;
    for (int y = 0; y < YS; ++y) { s += part[((size_t)c * YS + y) * 2]; q += part[((size_t)c * YS + y) * 2 + 1]; }
    const double m = s / n;
    double var = q / n - m * m; if (var < 0) var = 0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    if (rvar) {
        const double unb = n > 1 ? var * n / (n - 1.0) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

// Deferred running-statistics update of a BatchNorm call that ran with running_mean / running_var = NULL: from the batch
// statistics it left (mean, invstd), applied later so that the running buffers see the reference's CALL ORDER although the call's
// arithmetic ran earlier (the "wrong pair" head of a discriminator update, miscc/losses.py:152-160, evaluated with the real half
// before the fake images exist).  var = 1 / invstd^2 - eps in double (invstd is a rounded float: relative error of the variance
// <= 1.2e-7 (var + eps) / var).
__global__ __launch_bounds__(256) void bn_running_update_kernel(const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                float* __restrict__ rmean, float* __restrict__ rvar, int C, double n,
                                                                float eps, float momentum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean[c];
    if (rvar) {
        const double is = (double)invstd[c];
        double var = 1.0 / (is * is) - (double)eps; if (var < 0) var = 0;
        const double unb = n > 1 ? var * n / (n - 1.0) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

// -------------------------------------------------------------------------------- forward apply
template <int ACT, bool VEC>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ res, float* __restrict__ y, int C,
                                                         int HW, float slope) {
    constexpr int W = VEC ? 4 : 1;
    constexpr int WA = 4;
    const int Cy = (ACT == MOGAN_ACT_GLU) ? C / 2 : C;
    const int c = blockIdx.y % Cy, b = blockIdx.y / Cy;
    const int i = (blockIdx.x * 256 + threadIdx.x) * W;
    if (i >= HW) return;
    const float sc = gamma[c] * invstd[c], sh = beta[c] - mean[c] * sc;
    const float* px = x + ((size_t)b * C + c) * HW + i;
    float v[WA], o[WA];
    if (VEC) { const float4 t = *(const float4*)px; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else v[0] = *px;
    if (ACT == MOGAN_ACT_GLU) {
        const int cg = c + Cy;
        const float sc2 = gamma[cg] * invstd[cg], sh2 = beta[cg] - mean[cg] * sc2;
        float g[WA];
        if (VEC) { const float4 t = *(const float4*)(px + (size_t)Cy * HW); g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w; }
        else g[0] = px[(size_t)Cy * HW];
#pragma unroll
        for (int k = 0; k < W; ++k) o[k] = (v[k] * sc + sh) * sigmoidf_(g[k] * sc2 + sh2);
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) {
            float t = v[k] * sc + sh;
            if (ACT == MOGAN_ACT_RELU) t = t > 0.f ? t : 0.f;
            if (ACT == MOGAN_ACT_LRELU) t = t > 0.f ? t : t * slope;
            o[k] = t;
        }
    }
    float* py = y + ((size_t)b * Cy + c) * HW + i;
    if (res) {
        const float* pr = res + ((size_t)b * Cy + c) * HW + i;
        if (VEC) { const float4 t = *(const float4*)pr; o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w; }
        else o[0] += *pr;
    }
    if (VEC) *(float4*)py = make_float4(o[0], o[1], o[2], o[3]); else *py = o[0];
}

// -------------------------------------------------------------------------------- backward
// dy_bn (gradient at the BN output) from dy (gradient at the activation output), recomputing the BN output.
// Non-GLU: channel c.  GLU: pair (c, c+Cy): a = bn_c, g = bn_{c+Cy}; y = a*sig(g).
template <int ACT>
__device__ __forceinline__ void act_bwd(float xa, float xg, float dyv, float sc, float sh, float sc2, float sh2,
                                        float slope, float& da, float& dg, float& xha_scaled_dummy) {
    (void)xha_scaled_dummy;
    if (ACT == MOGAN_ACT_GLU) {
        const float a = xa * sc + sh, s = sigmoidf_(xg * sc2 + sh2);
        da = dyv * s; dg = dyv * a * s * (1.f - s);
    } else {
        const float t = xa * sc + sh;
        if (ACT == MOGAN_ACT_RELU) da = t > 0.f ? dyv : 0.f;
        else if (ACT == MOGAN_ACT_LRELU) da = t > 0.f ? dyv : dyv * slope;
        else da = dyv;
        dg = 0.f;
    }
}

// partial sums per (channel, ys): [sum dy_a, sum dy_a*xhat_a, sum dy_g, sum dy_g*xhat_g]
template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int B, int C, int HW,
                                                             Split sp, float slope, double* __restrict__ part) {
    __shared__ double sh_[16];
    const int Cy = (ACT == MOGAN_ACT_GLU) ? C / 2 : C;
    const int c = blockIdx.x, ys = blockIdx.y;          // c in [0, Cy)
    const int bi = ys / sp.hs, hi = ys % sp.hs;
    const int b0 = bi * sp.bper, b1 = min(B, b0 + sp.bper);
    const int h0 = hi * sp.hper, h1 = min(HW, h0 + sp.hper);
    const float mu = mean[c], is = invstd[c];
    const float sc = gamma[c] * is, sh = beta[c] - mu * sc;
    float mu2 = 0, is2 = 0, sc2 = 0, sh2 = 0;
    if (ACT == MOGAN_ACT_GLU) { mu2 = mean[c + Cy]; is2 = invstd[c + Cy]; sc2 = gamma[c + Cy] * is2; sh2 = beta[c + Cy] - mu2 * sc2; }
    double acc[4] = {0, 0, 0, 0};
    float dummy = 0;
    for (int b = b0; b < b1; ++b) {
        const float* pxa = x + ((size_t)b * C + c) * HW;
        const float* pxg = pxa + (size_t)Cy * HW;
        const float* pdy = dy + ((size_t)b * Cy + c) * HW;
        if ((HW & 3) == 0) {                       // (h0, h1 are multiples of 4 then: Split::hper is)
            for (int i = h0 + threadIdx.x * 4; i < h1; i += 1024) {
                const float4 xa4 = *(const float4*)(pxa + i), dy4 = *(const float4*)(pdy + i);
                float4 xg4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ACT == MOGAN_ACT_GLU) xg4 = *(const float4*)(pxg + i);
                const float xa_[4] = {xa4.x, xa4.y, xa4.z, xa4.w}, xg_[4] = {xg4.x, xg4.y, xg4.z, xg4.w};
                const float dy_[4] = {dy4.x, dy4.y, dy4.z, dy4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float da, dg;
                    act_bwd<ACT>(xa_[k], xg_[k], dy_[k], sc, sh, sc2, sh2, slope, da, dg, dummy);
                    acc[0] += da; acc[1] += (double)da * ((xa_[k] - mu) * is);
                    if (ACT == MOGAN_ACT_GLU) { acc[2] += dg; acc[3] += (double)dg * ((xg_[k] - mu2) * is2); }
                }
            }
            continue;
        }
        for (int i = h0 + threadIdx.x; i < h1; i += 256) {
            const float xa = pxa[i], xg = (ACT == MOGAN_ACT_GLU) ? pxg[i] : 0.f;
            float da, dg;
            act_bwd<ACT>(xa, xg, pdy[i], sc, sh, sc2, sh2, slope, da, dg, dummy);
            acc[0] += da; acc[1] += (double)da * ((xa - mu) * is);
            if (ACT == MOGAN_ACT_GLU) { acc[2] += dg; acc[3] += (double)dg * ((xg - mu2) * is2); }
        }
    }
    block_sum<4>(acc, sh_);
    if (threadIdx.x == 0) {
        double* o = part + ((size_t)c * gridDim.y + ys) * 4;
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
    }
}

// -> sums[C][2] = (sum dy_bn, sum dy_bn*xhat) as float, dgamma/dbeta
template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ part, int C, int YS,
                                                              float* __restrict__ sums, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate) {
    const int Cy = (ACT == MOGAN_ACT_GLU) ? C / 2 : C;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Cy) return;
    double a[4] = {0, 0, 0, 0};
    for (int y = 0; y < YS; ++y)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += part[((size_t)c * YS + y) * 4 + k];
    sums[c * 2] = (float)a[0]; sums[c * 2 + 1] = (float)a[1];
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)a[0];
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)a[1];
    if (ACT == MOGAN_ACT_GLU) {
        const int cg = c + Cy;
        sums[cg * 2] = (float)a[2]; sums[cg * 2 + 1] = (float)a[3];
        if (dbeta) dbeta[cg] = (accumulate ? dbeta[cg] : 0.f) + (float)a[2];
        if (dgamma) dgamma[cg] = (accumulate ? dgamma[cg] : 0.f) + (float)a[3];
    }
}

// dx = gamma*invstd * (dy_bn - sum_dy/n - xhat * sum_dyxhat/n)
template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ sums, float* __restrict__ dx, int C,
                                                           int HW, float slope, float inv_n) {
    const int Cy = (ACT == MOGAN_ACT_GLU) ? C / 2 : C;
    const int c = blockIdx.y % Cy, b = blockIdx.y / Cy;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float mu = mean[c], is = invstd[c];
    const float sc = gamma[c] * is, sh = beta[c] - mu * sc;
    float mu2 = 0, is2 = 0, sc2 = 0, sh2 = 0;
    if (ACT == MOGAN_ACT_GLU) { mu2 = mean[c + Cy]; is2 = invstd[c + Cy]; sc2 = gamma[c + Cy] * is2; sh2 = beta[c + Cy] - mu2 * sc2; }
    const size_t ia = ((size_t)b * C + c) * HW + i, ig = ia + (size_t)Cy * HW;
    const float xa = x[ia], xg = (ACT == MOGAN_ACT_GLU) ? x[ig] : 0.f;
    float da, dg, dummy = 0;
    act_bwd<ACT>(xa, xg, dy[((size_t)b * Cy + c) * HW + i], sc, sh, sc2, sh2, slope, da, dg, dummy);
    dx[ia] = sc * (da - sums[c * 2] * inv_n - (xa - mu) * is * sums[c * 2 + 1] * inv_n);
    if (ACT == MOGAN_ACT_GLU) {
        const int cg = c + Cy;
        dx[ig] = sc2 * (dg - sums[cg * 2] * inv_n - (xg - mu2) * is2 * sums[cg * 2 + 1] * inv_n);
    }
}


// -------------------------------------------------------------------------------- finalize folded into the apply pass
// HW % 4 == 0 maps beyond the one-launch size: the per-channel reduction of the partial sums (bn_finalize_kernel /
// bn_bwd_finalize_kernel, one tiny launch each) moves into the apply kernels -- every wave sums its channel's YS partial
// entries (lanes stride the entries, xor-butterfly: all lanes hold the same fp64 totals, no LDS, no barrier) and the block
// (first tile of the first image) also writes what the finalize kernel wrote: mean / invstd / running statistics, dgamma / dbeta.
// Two launches per direction instead of three; each block covers 4096 values of one (image, channel) plane.
__device__ __forceinline__ double wave_allsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int NV, int STRIDE>
__device__ __forceinline__ void part_sums(const double* __restrict__ part, int c, int YS, double (&a)[NV]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NV; ++k) a[k] = 0.0;
    for (int y = lane; y < YS; y += 64)
#pragma unroll
        for (int k = 0; k < NV; ++k) a[k] += part[((size_t)c * YS + y) * STRIDE + k];
#pragma unroll
    for (int k = 0; k < NV; ++k) a[k] = wave_allsum(a[k]);
}
__device__ __forceinline__ void stats_of(const double (&a)[2], double n, float eps, float& mean, float& invstd, double& var) {
    const double m = a[0] / n;
    var = a[1] / n - m * m; if (var < 0) var = 0;
    mean = (float)m; invstd = (float)(1.0 / sqrt(var + (double)eps));
}
__device__ __forceinline__ void write_stats(int c, float mean, float invstd, double var, double n, float momentum,
                                            float* __restrict__ mo, float* __restrict__ io, float* __restrict__ rmean,
                                            float* __restrict__ rvar) {
    mo[c] = mean; io[c] = invstd;
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    if (rvar) {
        const double unb = n > 1 ? var * n / (n - 1.0) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

constexpr int FUSED_PER = 4096;     // values of one plane per block

template <int ACT>
__global__ __launch_bounds__(256) void bn_fwd_apply_fused_kernel(const float* __restrict__ x, const double* __restrict__ part,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ res, float* __restrict__ y,
                                                                 float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                                                 float* __restrict__ rmean, float* __restrict__ rvar, int C, int HW,
                                                                 int YS, double n, float eps, float momentum, float slope,
                                                                 int writer) {
    const int Cy = (ACT == MOGAN_ACT_GLU) ? C / 2 : C;
    const int c = blockIdx.y % Cy, b = blockIdx.y / Cy;
    const bool wr = writer && blockIdx.x == 0 && b == 0 && threadIdx.x == 0;
    double a[2], var;
    float mu, is, mu2 = 0.f, is2 = 0.f;
    part_sums<2, 2>(part, c, YS, a);
    stats_of(a, n, eps, mu, is, var);
    if (wr) write_stats(c, mu, is, var, n, momentum, mean_o, invstd_o, rmean, rvar);
    const float sc = gamma[c] * is, sh = beta[c] - mu * sc;
    float sc2 = 0.f, sh2 = 0.f;
    if (ACT == MOGAN_ACT_GLU) {
        const int cg = c + Cy;
        part_sums<2, 2>(part, cg, YS, a);
        stats_of(a, n, eps, mu2, is2, var);
        if (wr) write_stats(cg, mu2, is2, var, n, momentum, mean_o, invstd_o, rmean, rvar);
        sc2 = gamma[cg] * is2; sh2 = beta[cg] - mu2 * sc2;
    }
    const float* px = x + ((size_t)b * C + c) * HW;
    float* py = y + ((size_t)b * Cy + c) * HW;
    const float* pr = res ? res + ((size_t)b * Cy + c) * HW : nullptr;
    const int i0 = blockIdx.x * FUSED_PER, i1 = min(HW, i0 + FUSED_PER);
    for (int i = i0 + threadIdx.x * 4; i < i1; i += 1024) {
        const float4 t = *(const float4*)(px + i);
        const float v[4] = {t.x, t.y, t.z, t.w};
        float o[4];
        if (ACT == MOGAN_ACT_GLU) {
            const float4 u = *(const float4*)(px + (size_t)Cy * HW + i);
            const float g[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (v[k] * sc + sh) * sigmoidf_(g[k] * sc2 + sh2);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float w = v[k] * sc + sh;
                if (ACT == MOGAN_ACT_RELU) w = w > 0.f ? w : 0.f;
                if (ACT == MOGAN_ACT_LRELU) w = w > 0.f ? w : w * slope;
                o[k] = w;
            }
        }
        if (pr) { const float4 r = *(const float4*)(pr + i); o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w; }
        *(float4*)(py + i) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const double* __restrict__ part, float* __restrict__ dx,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int HW,
                                                                 int YS, float slope, float inv_n, int accumulate, int writer) {
    constexpr int NV = (ACT == MOGAN_ACT_GLU) ? 4 : 2;
    const int Cy = (ACT == MOGAN_ACT_GLU) ? C / 2 : C;
    const int c = blockIdx.y % Cy, b = blockIdx.y / Cy, cg = c + Cy;
    double a[NV];
    part_sums<NV, 4>(part, c, YS, a);
    const float s0 = (float)a[0], s1 = (float)a[1];
    float s2 = 0.f, s3 = 0.f;
    if (ACT == MOGAN_ACT_GLU) { s2 = (float)a[NV - 2]; s3 = (float)a[NV - 1]; }
    if (writer && blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
        if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + s0;
        if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + s1;
        if (ACT == MOGAN_ACT_GLU) {
            if (dbeta) dbeta[cg] = (accumulate ? dbeta[cg] : 0.f) + s2;
            if (dgamma) dgamma[cg] = (accumulate ? dgamma[cg] : 0.f) + s3;
        }
    }
    const float mu = mean[c], is = invstd[c];
    const float sc = gamma[c] * is, sh = beta[c] - mu * sc;
    float mu2 = 0, is2 = 0, sc2 = 0, sh2 = 0;
    if (ACT == MOGAN_ACT_GLU) { mu2 = mean[cg]; is2 = invstd[cg]; sc2 = gamma[cg] * is2; sh2 = beta[cg] - mu2 * sc2; }
    const float* pxa = x + ((size_t)b * C + c) * HW;
    const float* pxg = pxa + (size_t)Cy * HW;
    const float* pdy = dy + ((size_t)b * Cy + c) * HW;
    float* pda = dx + ((size_t)b * C + c) * HW;
    float* pdg = pda + (size_t)Cy * HW;
    const int i0 = blockIdx.x * FUSED_PER, i1 = min(HW, i0 + FUSED_PER);
    float dummy = 0;
    for (int i = i0 + threadIdx.x * 4; i < i1; i += 1024) {
        const float4 xa4 = *(const float4*)(pxa + i), dy4 = *(const float4*)(pdy + i);
        float4 xg4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ACT == MOGAN_ACT_GLU) xg4 = *(const float4*)(pxg + i);
        const float xa_[4] = {xa4.x, xa4.y, xa4.z, xa4.w}, xg_[4] = {xg4.x, xg4.y, xg4.z, xg4.w}, dy_[4] = {dy4.x, dy4.y, dy4.z, dy4.w};
        float oa[4], og[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float da, dg;
            act_bwd<ACT>(xa_[k], xg_[k], dy_[k], sc, sh, sc2, sh2, slope, da, dg, dummy);
            oa[k] = sc * (da - s0 * inv_n - (xa_[k] - mu) * is * s1 * inv_n);
            og[k] = (ACT == MOGAN_ACT_GLU) ? sc2 * (dg - s2 * inv_n - (xg_[k] - mu2) * is2 * s3 * inv_n) : 0.f;
        }
        *(float4*)(pda + i) = make_float4(oa[0], oa[1], oa[2], oa[3]);
        if (ACT == MOGAN_ACT_GLU) *(float4*)(pdg + i) = make_float4(og[0], og[1], og[2], og[3]);
    }
}


// -------------------------------------------------------------------------------- one-launch BatchNorm for small maps
// B*HW <= SMALL_NE values per channel (<= 16x16 maps at B = 16; BatchNorm1d): ONE block per output channel (GLU: per pair)
// does what bn_partial + bn_finalize + bn_act_fwd (and bn_bwd_partial + finalize + apply) do in three launches -- pass 1
// reduces over the channel, pass 2 re-reads the L2-resident values and applies.  Same arithmetic (fp64 sums, float apply).
constexpr int SMALL_NE = 4096;

template <int ACT>
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ res,
                                                           float* __restrict__ y, float* __restrict__ mean,
                                                           float* __restrict__ invstd, float* __restrict__ rmean,
                                                           float* __restrict__ rvar, int B, int C, int HW, float eps,
                                                           float momentum, float slope, int G) {
    __shared__ double sh[16];
    __shared__ float st[4];
    const int Cy = (ACT == MOGAN_ACT_GLU) ? C / 2 : C;
    const int c = blockIdx.x, NE = B * HW;
    constexpr int NCH = (ACT == MOGAN_ACT_GLU) ? 2 : 1;
    // G > 1: the batch holds G groups of B images, each normalised with its OWN statistics -- G BatchNorm calls in sequence (the
    // running statistics are updated group after group, as G calls would), one launch (the object pathways)
    for (int grp = 0; grp < G; ++grp, x += (size_t)B * C * HW, y += (size_t)B * Cy * HW, mean += C, invstd += C) {
    if (grp) __syncthreads();
    // a thread's <= 16 values (x2 for GLU) are loaded ONCE, all loads in flight (clamped addresses, no branch), and stay in registers
    // for the apply pass (a `for (e ...)` loop with one dependent load per iteration made this kernel 13-16 us for 4096 values)
    constexpr int EPT = SMALL_NE / 256;
    float v[NCH][EPT];
    unsigned off[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = threadIdx.x + 256 * i;
        const bool ok = e < NE;
        const int b = e / HW, pos = e - b * HW;
        off[i] = ok ? (unsigned)((b * C + c) * HW + pos) : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const float t = x[ok ? (size_t)off[i] + (size_t)k * Cy * HW : 0];
            v[k][i] = ok ? t : 0.f;
        }
    }
    double acc[2 * NCH];
#pragma unroll
    for (int k = 0; k < 2 * NCH; ++k) acc[k] = 0.0;
#pragma unroll
    for (int i = 0; i < EPT; ++i)
#pragma unroll
        for (int k = 0; k < NCH; ++k) { const double t = v[k][i]; acc[2 * k] += t; acc[2 * k + 1] += t * t; }
    block_sum<2 * NCH>(acc, sh);
    if (threadIdx.x == 0) {
        const double n = (double)NE;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = c + k * Cy;
            const double m = acc[2 * k] / n;
            double var = acc[2 * k + 1] / n - m * m; if (var < 0) var = 0;
            const float mu = (float)m, is = (float)(1.0 / sqrt(var + (double)eps));
            mean[ch] = mu; invstd[ch] = is;
            if (rmean) rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * mu;
            if (rvar) {
                const double unb = n > 1 ? var * n / (n - 1.0) : var;
                rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)unb;
            }
            st[2 * k] = mu; st[2 * k + 1] = is;
        }
    }
    __syncthreads();
    const float sc = gamma[c] * st[1], shf = beta[c] - st[0] * sc;
    float sc2 = 0.f, sh2 = 0.f;
    if (ACT == MOGAN_ACT_GLU) { sc2 = gamma[c + Cy] * st[3]; sh2 = beta[c + Cy] - st[2] * sc2; }
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        if (off[i] == 0xFFFFFFFFu) continue;
        const int e = threadIdx.x + 256 * i;
        const int b = e / HW, pos = e - b * HW;
        const size_t iy = ((size_t)b * Cy + c) * HW + pos;
        float t = v[0][i] * sc + shf;
        if (ACT == MOGAN_ACT_GLU) t = t * sigmoidf_(v[NCH - 1][i] * sc2 + sh2);
        if (ACT == MOGAN_ACT_RELU) t = t > 0.f ? t : 0.f;
        if (ACT == MOGAN_ACT_LRELU) t = t > 0.f ? t : t * slope;
        if (res) t += res[iy];
        y[iy] = t;
    }
    }
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int B, int C, int HW, float slope,
                                                           int accumulate, int G) {
    __shared__ double sh_[16];
    __shared__ float sums[4];
    const int Cy = (ACT == MOGAN_ACT_GLU) ? C / 2 : C;
    const int c = blockIdx.x, NE = B * HW;
    for (int grp = 0; grp < G; ++grp, x += (size_t)B * C * HW, dy += (size_t)B * Cy * HW, dx += (size_t)B * C * HW, mean += C,
             invstd += C, accumulate = 1) {           // (d gamma / d beta: the groups' contributions add up)
    if (grp) __syncthreads();
    const float mu = mean[c], is = invstd[c];
    const float sc = gamma[c] * is, sh = beta[c] - mu * sc;
    float mu2 = 0, is2 = 0, sc2 = 0, sh2 = 0;
    if (ACT == MOGAN_ACT_GLU) { mu2 = mean[c + Cy]; is2 = invstd[c + Cy]; sc2 = gamma[c + Cy] * is2; sh2 = beta[c + Cy] - mu2 * sc2; }
    double acc[4] = {0, 0, 0, 0};
    float dummy = 0;
    // x (both halves for GLU) and dy of a thread's <= 16 elements: loaded once, all in flight, kept for the second pass
    constexpr int EPT = SMALL_NE / 256, NCH = (ACT == MOGAN_ACT_GLU) ? 2 : 1;
    float xv[NCH][EPT], dv[EPT];
    unsigned off[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = threadIdx.x + 256 * i;
        const bool ok = e < NE;
        const int b = e / HW, pos = e - b * HW;
        off[i] = ok ? (unsigned)((b * C + c) * HW + pos) : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const float t = x[ok ? (size_t)off[i] + (size_t)k * Cy * HW : 0];
            xv[k][i] = ok ? t : 0.f;
        }
        const float t = dy[ok ? ((size_t)b * Cy + c) * HW + pos : 0];
        dv[i] = ok ? t : 0.f;
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        if (off[i] == 0xFFFFFFFFu) continue;
        const float xa = xv[0][i], xg = xv[NCH - 1][i];
        float da, dg;
        act_bwd<ACT>(xa, xg, dv[i], sc, sh, sc2, sh2, slope, da, dg, dummy);
        acc[0] += da; acc[1] += (double)da * ((xa - mu) * is);
        if (ACT == MOGAN_ACT_GLU) { acc[2] += dg; acc[3] += (double)dg * ((xg - mu2) * is2); }
    }
    block_sum<4>(acc, sh_);
    if (threadIdx.x == 0) {
        sums[0] = (float)acc[0]; sums[1] = (float)acc[1]; sums[2] = (float)acc[2]; sums[3] = (float)acc[3];
        if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + sums[0];
        if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + sums[1];
        if (ACT == MOGAN_ACT_GLU) {
            if (dbeta) dbeta[c + Cy] = (accumulate ? dbeta[c + Cy] : 0.f) + sums[2];
            if (dgamma) dgamma[c + Cy] = (accumulate ? dgamma[c + Cy] : 0.f) + sums[3];
        }
    }
    __syncthreads();
    const float inv_n = 1.f / ((float)B * (float)HW);
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        if (off[i] == 0xFFFFFFFFu) continue;
        const size_t ia = off[i], ig = ia + (size_t)Cy * HW;
        const float xa = xv[0][i], xg = xv[NCH - 1][i];
        float da, dg;
        act_bwd<ACT>(xa, xg, dv[i], sc, sh, sc2, sh2, slope, da, dg, dummy);
        dx[ia] = sc * (da - sums[0] * inv_n - (xa - mu) * is * sums[1] * inv_n);
        if (ACT == MOGAN_ACT_GLU) dx[ig] = sc2 * (dg - sums[2] * inv_n - (xg - mu2) * is2 * sums[3] * inv_n);
    }
    }
}

// (BatchNorm1d, HW == 1, keeps its thread-per-channel kernels: a block per channel would be 16 values wide)
static bool bn_fused_env() { return true; }
static bool bn_fused_ok(int HW) { return (HW & 3) == 0 && HW >= 64 && bn_fused_env(); }
static bool bn_small_ok(int B, int C, int HW) { return HW >= 16 && (long long)B * HW <= SMALL_NE && C <= 65535 * 2; }

// -------------------------------------------------------------------------------- eval affine
template <int ACT, bool BWD>
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ out, int C,
                                                         int HW, float slope) {
    const int c = blockIdx.y % C;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const size_t idx = (size_t)blockIdx.y * HW + i;
    const float sc = scale[c], t = x[idx] * sc + shift[c];
    if (!BWD) {
        float o = t;
        if (ACT == MOGAN_ACT_RELU) o = t > 0.f ? t : 0.f;
        if (ACT == MOGAN_ACT_LRELU) o = t > 0.f ? t : t * slope;
        out[idx] = o;
    } else {
        float d = dy[idx];
        if (ACT == MOGAN_ACT_RELU) d = t > 0.f ? d : 0.f;
        if (ACT == MOGAN_ACT_LRELU) d = t > 0.f ? d : d * slope;
        out[idx] = d * sc;
    }
}

static inline int ok_launch() { return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH; }

template <int ACT>
static int bn_bwd_impl(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, float* dx, float* dgamma, float* dbeta, int B, int C, int HW, float slope,
                       int accumulate, void* ws, hipStream_t stream) {
    const int Cy = ACT == MOGAN_ACT_GLU ? C / 2 : C;
    if (bn_small_ok(B, C, HW)) {           // small map: one launch (bn_small_bwd_kernel)
        hipLaunchKernelGGL((bn_small_bwd_kernel<ACT>), dim3(Cy), dim3(256), 0, stream, x, dy, mean, invstd, gamma, beta, dx,
                           dgamma, dbeta, B, C, HW, slope, accumulate, 1);
        return ok_launch();
    }
    Split s = make_split(B, C, HW);
    const int YS = s.bs * s.hs;
    double* part = (double*)ws;
    float* sums = (float*)((char*)ws + (size_t)C * YS * 4 * sizeof(double));
    hipLaunchKernelGGL((bn_bwd_partial_kernel<ACT>), dim3(Cy, YS), dim3(256), 0, stream, x, dy, mean, invstd, gamma,
                       beta, B, C, HW, s, slope, part);
    const float inv_n = 1.f / ((float)B * (float)HW);
    const int bchunk = 65535 / Cy;
    if (bchunk < 1) return MOGAN_ERR_SHAPE;
    if (bn_fused_ok(HW)) {                 // finalize folded into the apply pass: two launches
        for (int b0 = 0; b0 < B; b0 += bchunk) {
            const int nb = B - b0 < bchunk ? B - b0 : bchunk;
            hipLaunchKernelGGL((bn_bwd_apply_fused_kernel<ACT>), dim3((HW + FUSED_PER - 1) / FUSED_PER, nb * Cy), dim3(256), 0, stream,
                               x + (size_t)b0 * C * HW, dy + (size_t)b0 * Cy * HW, mean, invstd, gamma, beta, (const double*)part,
                               dx + (size_t)b0 * C * HW, dgamma, dbeta, C, HW, YS, slope, inv_n, accumulate, b0 == 0 ? 1 : 0);
        }
        return ok_launch();
    }
    hipLaunchKernelGGL((bn_bwd_finalize_kernel<ACT>), dim3((Cy + 255) / 256), dim3(256), 0, stream,
                       (const double*)part, C, YS, sums, dgamma, dbeta, accumulate);
    for (int b0 = 0; b0 < B; b0 += bchunk) {
        const int nb = B - b0 < bchunk ? B - b0 : bchunk;
        hipLaunchKernelGGL((bn_bwd_apply_kernel<ACT>), dim3((HW + 255) / 256, nb * Cy), dim3(256), 0, stream,
                           x + (size_t)b0 * C * HW, dy + (size_t)b0 * Cy * HW, mean, invstd, gamma, beta,
                           (const float*)sums, dx + (size_t)b0 * C * HW, C, HW, slope, inv_n);
    }
    return ok_launch();
}

template <bool BWD>
static int affine_impl(const float* x, const float* dy, const float* scale, const float* shift, float* out, int B,
                       int C, int HW, int act, float slope, hipStream_t stream) {
    if (B <= 0 || C <= 0 || HW <= 0) return MOGAN_ERR_SHAPE;
    const int bchunk = 65535 / C;
    if (bchunk < 1) return MOGAN_ERR_SHAPE;
    for (int b0 = 0; b0 < B; b0 += bchunk) {
        const int nb = B - b0 < bchunk ? B - b0 : bchunk;
        dim3 grid((HW + 255) / 256, nb * C);
        const size_t off = (size_t)b0 * C * HW;
        const float* pdy = dy ? dy + off : nullptr;
        switch (act) {
            case MOGAN_ACT_NONE: hipLaunchKernelGGL((affine_act_kernel<MOGAN_ACT_NONE, BWD>), grid, dim3(256), 0, stream, x + off, pdy, scale, shift, out + off, C, HW, slope); break;
            case MOGAN_ACT_RELU: hipLaunchKernelGGL((affine_act_kernel<MOGAN_ACT_RELU, BWD>), grid, dim3(256), 0, stream, x + off, pdy, scale, shift, out + off, C, HW, slope); break;
            case MOGAN_ACT_LRELU: hipLaunchKernelGGL((affine_act_kernel<MOGAN_ACT_LRELU, BWD>), grid, dim3(256), 0, stream, x + off, pdy, scale, shift, out + off, C, HW, slope); break;
            default: return MOGAN_ERR_SHAPE;
        }
    }
    return ok_launch();
}

}  // namespace

extern "C" {

size_t mogan_bn_ws_bytes(int B, int C, int HW) {
    if (B <= 0 || C <= 0 || HW <= 0) return 0;
    const Split s = make_split(B, C, HW);
    return (size_t)C * s.bs * s.hs * 4 * sizeof(double) + (size_t)C * 2 * sizeof(float) + 64;
}

int mogan_bn_stats(const float* x, int B, int C, int HW, float eps, float momentum, float* mean, float* invstd,
                   float* running_mean, float* running_var, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (B <= 0 || C <= 0 || HW <= 0) return MOGAN_ERR_SHAPE;
    if (!ws || ws_bytes < mogan_bn_ws_bytes(B, C, HW)) return MOGAN_ERR_WS;
    double* part = (double*)ws;
    int YS = 1;
    if (HW == 1) {
        hipLaunchKernelGGL(bn1d_partial_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, x, B, C, part);
    } else {
        const Split s = make_split(B, C, HW);
        YS = s.bs * s.hs;
        if (YS > 65535) return MOGAN_ERR_SHAPE;
        hipLaunchKernelGGL(bn_partial_kernel, dim3(C, YS), dim3(256), 0, stream, x, B, C, HW, s, part);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, (const double*)part, C, YS,
                       (double)B * HW, eps, momentum, mean, invstd, running_mean, running_var);
    return ok_launch();
}

int mogan_bn_running_update(const float* mean, const float* invstd, float* running_mean, float* running_var, int C, long long n,
                            float eps, float momentum, hipStream_t stream) {
    if (!mean || !invstd || C <= 0 || n <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, mean, invstd, running_mean,
                       running_var, C, (double)n, eps, momentum);
    return ok_launch();
}

// statistics + apply in one call: small maps (B*HW <= 4096 values per channel) take ONE launch, larger ones the three of
// mogan_bn_stats + mogan_bn_act_fwd.  mean / invstd are written for the backward as by mogan_bn_stats.
int mogan_bn_act_fwd_fused(const float* x, const float* gamma, const float* beta, const float* residual, float* running_mean,
                           float* running_var, float* mean, float* invstd, float* y, int B, int C, int HW, int act, float slope,
                           float eps, float momentum, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (B <= 0 || C <= 0 || HW <= 0 || (act == MOGAN_ACT_GLU && (C & 1))) return MOGAN_ERR_SHAPE;
    const int Cy = act == MOGAN_ACT_GLU ? C / 2 : C;
    if (!bn_small_ok(B, C, HW) && bn_fused_ok(HW) && Cy <= 65535) {
        if (!ws || ws_bytes < mogan_bn_ws_bytes(B, C, HW)) return MOGAN_ERR_WS;
        const Split s = make_split(B, C, HW);
        const int YS = s.bs * s.hs;
        if (YS > 65535) return MOGAN_ERR_SHAPE;
        double* part = (double*)ws;
        hipLaunchKernelGGL(bn_partial_kernel, dim3(C, YS), dim3(256), 0, stream, x, B, C, HW, s, part);
        const int bchunk = 65535 / Cy;
        for (int b0 = 0; b0 < B; b0 += bchunk) {
            const int nb = B - b0 < bchunk ? B - b0 : bchunk;
            const dim3 grid((HW + FUSED_PER - 1) / FUSED_PER, nb * Cy);
#define MOGAN_FUSED_CASE(A) case A: hipLaunchKernelGGL((bn_fwd_apply_fused_kernel<A>), grid, dim3(256), 0, stream,                 \
                                                       x + (size_t)b0 * C * HW, (const double*)part, gamma, beta,                     \
                                                       residual ? residual + (size_t)b0 * Cy * HW : nullptr, y + (size_t)b0 * Cy * HW, \
                                                       mean, invstd, running_mean, running_var, C, HW, YS, (double)B * HW, eps,      \
                                                       momentum, slope, b0 == 0 ? 1 : 0); break;
            switch (act) {
                MOGAN_FUSED_CASE(MOGAN_ACT_NONE)
                MOGAN_FUSED_CASE(MOGAN_ACT_RELU)
                MOGAN_FUSED_CASE(MOGAN_ACT_LRELU)
                MOGAN_FUSED_CASE(MOGAN_ACT_GLU)
                default: return MOGAN_ERR_SHAPE;
            }
#undef MOGAN_FUSED_CASE
        }
        return ok_launch();
    }
    if (!bn_small_ok(B, C, HW)) {
        int rc = mogan_bn_stats(x, B, C, HW, eps, momentum, mean, invstd, running_mean, running_var, ws, ws_bytes, stream);
        return rc ? rc : mogan_bn_act_fwd(x, mean, invstd, gamma, beta, residual, y, B, C, HW, act, slope, stream);
    }
#define MOGAN_SMALL_CASE(A) case A: hipLaunchKernelGGL((bn_small_fwd_kernel<A>), dim3(Cy), dim3(256), 0, stream, x, gamma, beta, \
                                                       residual, y, mean, invstd, running_mean, running_var, B, C, HW, eps,   \
                                                       momentum, slope, 1); break;
    switch (act) {
        MOGAN_SMALL_CASE(MOGAN_ACT_NONE)
        MOGAN_SMALL_CASE(MOGAN_ACT_RELU)
        MOGAN_SMALL_CASE(MOGAN_ACT_LRELU)
        MOGAN_SMALL_CASE(MOGAN_ACT_GLU)
        default: return MOGAN_ERR_SHAPE;
    }
#undef MOGAN_SMALL_CASE
    return ok_launch();
}

// G BatchNorm(train) + activation calls on the G groups of B images of one (G*B, C, HW) tensor in ONE launch each way: group g is
// normalised with its own batch statistics (mean / invstd: G x C, group-major), the running statistics are updated G times in group
// order, d gamma / d beta sum over the groups.  Needs B*HW <= 4096 values per channel and group (any HW >= 1).
int mogan_bn_act_grouped_eligible(int G, int B, int C, int HW) {
    return G >= 1 && B >= 1 && C >= 1 && HW >= 1 && (long long)B * HW <= SMALL_NE && C <= 65535 * 2 &&
           (long long)G * B * C * HW < (1ll << 31) ? 1 : 0;
}

int mogan_bn_act_grouped_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* mean,
                             float* invstd, float* y, int G, int B, int C, int HW, int act, float slope, float eps, float momentum,
                             hipStream_t stream) {
    if (!mogan_bn_act_grouped_eligible(G, B, C, HW) || (act == MOGAN_ACT_GLU && (C & 1))) return MOGAN_ERR_SHAPE;
    const int Cy = act == MOGAN_ACT_GLU ? C / 2 : C;
#define MOGAN_GRP_CASE(A) case A: hipLaunchKernelGGL((bn_small_fwd_kernel<A>), dim3(Cy), dim3(256), 0, stream, x, gamma, beta, \
                                                     (const float*)nullptr, y, mean, invstd, running_mean, running_var, B, C, HW, eps, \
                                                     momentum, slope, G); break;
    switch (act) {
        MOGAN_GRP_CASE(MOGAN_ACT_NONE)
        MOGAN_GRP_CASE(MOGAN_ACT_RELU)
        MOGAN_GRP_CASE(MOGAN_ACT_LRELU)
        MOGAN_GRP_CASE(MOGAN_ACT_GLU)
        default: return MOGAN_ERR_SHAPE;
    }
#undef MOGAN_GRP_CASE
    return ok_launch();
}

int mogan_bn_act_grouped_bwd(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             float* dx, float* dgamma, float* dbeta, int G, int B, int C, int HW, int act, float slope, int accumulate,
                             hipStream_t stream) {
    if (!mogan_bn_act_grouped_eligible(G, B, C, HW) || (act == MOGAN_ACT_GLU && (C & 1))) return MOGAN_ERR_SHAPE;
    const int Cy = act == MOGAN_ACT_GLU ? C / 2 : C;
#define MOGAN_GRP_CASE(A) case A: hipLaunchKernelGGL((bn_small_bwd_kernel<A>), dim3(Cy), dim3(256), 0, stream, x, dy, mean, invstd, gamma, \
                                                     beta, dx, dgamma, dbeta, B, C, HW, slope, accumulate, G); break;
    switch (act) {
        MOGAN_GRP_CASE(MOGAN_ACT_NONE)
        MOGAN_GRP_CASE(MOGAN_ACT_RELU)
        MOGAN_GRP_CASE(MOGAN_ACT_LRELU)
        MOGAN_GRP_CASE(MOGAN_ACT_GLU)
        default: return MOGAN_ERR_SHAPE;
    }
#undef MOGAN_GRP_CASE
    return ok_launch();
}

#define MOGAN_FWD_CASE(A)                                                                                           \
    case A:                                                                                                         \
        if (vec) hipLaunchKernelGGL((bn_act_fwd_kernel<A, true>), grid, dim3(256), 0, stream, x, mean, invstd,      \
                                    gamma, beta, residual, y, C, HW, slope);                                        \
        else hipLaunchKernelGGL((bn_act_fwd_kernel<A, false>), grid, dim3(256), 0, stream, x, mean, invstd, gamma,  \
                                beta, residual, y, C, HW, slope);                                                   \
        break;

int mogan_bn_act_fwd(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     const float* residual, float* y, int B, int C, int HW, int act, float slope, hipStream_t stream) {
    if (B <= 0 || C <= 0 || HW <= 0 || (act == MOGAN_ACT_GLU && (C & 1))) return MOGAN_ERR_SHAPE;
    const int Cy = act == MOGAN_ACT_GLU ? C / 2 : C;
    const bool vec = (HW & 3) == 0;
    if ((long long)B * Cy > 65535) {
        // fold: gridDim.y limit -- loop over batch chunks
        const int bchunk = 65535 / Cy;
        if (bchunk < 1) return MOGAN_ERR_SHAPE;
        for (int b0 = 0; b0 < B; b0 += bchunk) {
            const int nb = B - b0 < bchunk ? B - b0 : bchunk;
            int rc = mogan_bn_act_fwd(x + (size_t)b0 * C * HW, mean, invstd, gamma, beta,
                                      residual ? residual + (size_t)b0 * Cy * HW : nullptr, y + (size_t)b0 * Cy * HW, nb,
                                      C, HW, act, slope, stream);
            if (rc) return rc;
        }
        return 0;
    }
    const int per = vec ? 1024 : 256;
    dim3 grid((HW + per - 1) / per, B * Cy);
    switch (act) {
        MOGAN_FWD_CASE(MOGAN_ACT_NONE)
        MOGAN_FWD_CASE(MOGAN_ACT_RELU)
        MOGAN_FWD_CASE(MOGAN_ACT_LRELU)
        MOGAN_FWD_CASE(MOGAN_ACT_GLU)
        default: return MOGAN_ERR_SHAPE;
    }
    return ok_launch();
}

int mogan_bn_act_bwd(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, float* dx, float* dgamma, float* dbeta, int B, int C, int HW, int act,
                     float slope, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (B <= 0 || C <= 0 || HW <= 0 || (act == MOGAN_ACT_GLU && (C & 1))) return MOGAN_ERR_SHAPE;
    if (!ws || ws_bytes < mogan_bn_ws_bytes(B, C, HW)) return MOGAN_ERR_WS;
    {
        const Split s = make_split(B, C, HW);
        if (s.bs * s.hs > 65535) return MOGAN_ERR_SHAPE;
    }
    switch (act) {
        case MOGAN_ACT_NONE: return bn_bwd_impl<MOGAN_ACT_NONE>(x, dy, mean, invstd, gamma, beta, dx, dgamma, dbeta, B, C, HW, slope, accumulate, ws, stream);
        case MOGAN_ACT_RELU: return bn_bwd_impl<MOGAN_ACT_RELU>(x, dy, mean, invstd, gamma, beta, dx, dgamma, dbeta, B, C, HW, slope, accumulate, ws, stream);
        case MOGAN_ACT_LRELU: return bn_bwd_impl<MOGAN_ACT_LRELU>(x, dy, mean, invstd, gamma, beta, dx, dgamma, dbeta, B, C, HW, slope, accumulate, ws, stream);
        case MOGAN_ACT_GLU: return bn_bwd_impl<MOGAN_ACT_GLU>(x, dy, mean, invstd, gamma, beta, dx, dgamma, dbeta, B, C, HW, slope, accumulate, ws, stream);
        default: return MOGAN_ERR_SHAPE;
    }
}

int mogan_affine_act_fwd(const float* x, const float* scale, const float* shift, float* y, int B, int C, int HW,
                         int act, float slope, hipStream_t stream) {
    return affine_impl<false>(x, nullptr, scale, shift, y, B, C, HW, act, slope, stream);
}
int mogan_affine_act_bwd(const float* x, const float* dy, const float* scale, const float* shift, float* dx, int B,
                         int C, int HW, int act, float slope, hipStream_t stream) {
    return affine_impl<true>(x, dy, scale, shift, dx, B, C, HW, act, slope, stream);
}

// backward of y = relu(scale*conv + shift) given the OUTPUT y: dx = dy * scale[c] * (y > 0)
int mogan_affine_relu_bwd_out(const float* y, const float* dy, const float* scale, float* dx, int B, int C, int HW,
                              hipStream_t stream) {
    if (B <= 0 || C <= 0 || HW <= 0) return MOGAN_ERR_SHAPE;
    const long long n = (long long)B * C * HW;
    hipLaunchKernelGGL(affine_relu_bwd_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, y, dy, scale, dx,
                       n, C, HW);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

}  // extern "C"
