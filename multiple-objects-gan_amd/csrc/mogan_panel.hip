// mogan_panel.hip -- the tail of the frozen trunk's panel GEMMs (mogan_pk_group, mogan_pgemm.hip).
//
// The frozen, eval-mode Inception-v3 trunk of CNN_ENCODER (code/coco/attngan/model.py:258-299) keeps activations and gradients as
// pixel panels between its convolutions.  What stands between two GEMMs of that chain is elementwise (plus one 3x3 box filter) and
// runs here as ONE launch per dependency level of a Mixed block, over up to eight channel slices:
//
//     v = sum over the sources' K-split slabs            raw GEMM results (B, rows, H, W), or any fp32 NCHW slice
//     v = mean of the 3x3 neighbourhood (zero padded)    the pool branch: F.avg_pool2d(x, 3, 1, 1) commutes with the 1x1 convolution
//     v = v * scale[c] + shift[c]; relu                  forward: eval-mode BatchNorm folded + ReLU
//     v += add; v = 0 where mask <= 0                    backward: other contributors of the gradient, ReLU mask of the layer's input
//     -> fp32 NCHW slice and / or pixel-panel slice      what the next GEMM (and the old-path kernels around the trunk) read
//
// Block = 32 channels x 64 pixels (pixels run over the images): phase 1 reads / computes / writes fp32 with the pixel index
// fastest (coalesced), phase 2 takes 8 channels of one pixel from LDS, splits them into their bf16 pieces and stores the three
// 16-byte units of the panel -- the layout pgemm_body gathers (mogan_pgemm.hip: [pixel][group of 32][piece][32 bf16]).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mogan_hip.h"
#include "mogan_internal.h"
#include "mogan_mma.h"

namespace {

constexpr int PT_MAXG = 8;
struct TailGroup { MoganTailArgs m[PT_MAXG]; unsigned end[PT_MAXG]; int n; };

template <bool BOX>
__global__ __launch_bounds__(256, 4) void panel_tail_kernel(const TailGroup g) {
    __shared__ float L[32][65];
    int mi = 0;
#pragma unroll
    for (int i = 0; i < PT_MAXG - 1; ++i) if (i + 1 < g.n && blockIdx.x >= g.end[i]) mi = i + 1;
    const MoganTailArgs& a = g.m[mi];
    const unsigned local = blockIdx.x - (mi ? g.end[mi - 1] : 0u);
    const unsigned HW = (unsigned)(a.H * a.W);
    const unsigned Q = (unsigned)a.B * HW;
    const unsigned nqb = (Q + 63u) / 64u;
    const unsigned cgi = local / nqb, qb = local - cgi * nqb;
    const int c0 = (int)cgi * 32;
    const unsigned q0 = qb * 64u;
    const int tid = threadIdx.x;
    // the thread's 8 elements: pixel q0 + (tid & 63), channels c0 + (tid >> 6) + 4 j.  Every tensor has < 2^31 elements (checked by
    // the entry point): 32-bit offsets; all loads of a step are issued for the 8 elements together (the kernel is latency bound)
    const int ql = tid & 63, cw = tid >> 6;
    const unsigned q = q0 + (unsigned)ql;
    const bool qok = q < Q;
    const unsigned qq = qok ? q : 0u;
    const unsigned img = qq / HW, pix = qq - img * HW;
    float v[8]; bool ok[8]; unsigned inner[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = c0 + cw + 4 * j;
        ok[j] = qok && ch < a.n;
        inner[j] = ok[j] ? (unsigned)ch * HW + pix : 0u;
        v[j] = 0.f;
    }
    if (!BOX || !a.box) {
        for (int s = 0; s < a.nsrc; ++s) {
            const unsigned ib = img * (unsigned)a.src_bs[s];
            const float* __restrict__ p = a.src[s];
            for (int k = 0; k < a.src_nsplit[s]; ++k, p += a.src_slab[s]) {
                float t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = p[ok[j] ? ib + inner[j] : 0u];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += t[j];
            }
        }
    } else {
        const int W = a.W, H = a.H;
        const int y = (int)pix / W, x = (int)pix - y * W;
        for (int s = 0; s < a.nsrc; ++s) {
            const unsigned ib = img * (unsigned)a.src_bs[s];
            const float* __restrict__ p = a.src[s];
            for (int k = 0; k < a.src_nsplit[s]; ++k, p += a.src_slab[s]) {
#pragma unroll 1
                for (int dy = -1; dy <= 1; ++dy) {
                    float t[3][8];
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int yy = y + dy, xx = x + dx;
                        const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                        const int d = dy * W + dx;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float u = p[(ok[j] && in) ? ib + inner[j] + d : 0u];
                            t[dx + 1][j] = in ? u : 0.f;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += t[0][j] + t[1][j] + t[2][j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= (1.f / 9.f);
    }
    if (a.scale != nullptr) {
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int ch = ok[j] ? c0 + cw + 4 * j : 0; sc[j] = a.scale[ch]; sh[j] = a.shift[ch]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
    }
    if (a.relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    if (a.add != nullptr) {
        const unsigned ib = img * (unsigned)a.add_bs;
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = a.add[ok[j] ? ib + inner[j] : 0u];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += t[j];
    }
    if (a.mask != nullptr) {
        const unsigned ib = img * (unsigned)a.mask_bs;
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = a.mask[ok[j] ? ib + inner[j] : 0u];
#pragma unroll
        for (int j = 0; j < 8; ++j) if (!(t[j] > 0.f)) v[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (!ok[j]) v[j] = 0.f;
    if (a.dst != nullptr) {
        const unsigned ib = img * (unsigned)a.dst_bs;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (ok[j]) a.dst[ib + inner[j]] = v[j];
    }
    if (a.panel == nullptr) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) L[cw + 4 * j][ql] = v[j];
    __syncthreads();
    const int pl_q = tid >> 2, g8 = tid & 3;
    const unsigned qp = q0 + (unsigned)pl_q;
    if (qp >= Q) return;
    uint32_t w[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x6_split2(L[g8 * 8 + 2 * j][pl_q], L[g8 * 8 + 2 * j + 1][pl_q], w[0][j], w[1][j], w[2][j]);
    unsigned char* d = (unsigned char*)a.panel + ((size_t)qp * (size_t)a.CGp + (size_t)(a.cg0 + (int)cgi)) * 192 + g8 * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *(uint4*)(d + pl * 64) = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
}

}  // namespace

extern "C" {

int mogan_panel_tail_group(int n, const MoganTailArgs* args, hipStream_t stream) {
    if (n <= 0 || n > PT_MAXG || !args) return MOGAN_ERR_SHAPE;
    TailGroup g{};
    g.n = n;
    long long end = 0;
    for (int i = 0; i < n; ++i) {
        const MoganTailArgs& a = args[i];
        if (a.B <= 0 || a.n <= 0 || a.H <= 0 || a.W <= 0 || a.nsrc <= 0 || a.nsrc > MOGAN_TAIL_MAXSRC || (!a.dst && !a.panel) ||
            ((a.scale == nullptr) != (a.shift == nullptr)))
            return MOGAN_ERR_SHAPE;
        const long long Q = (long long)a.B * a.H * a.W;
        if (Q >= (1ll << 31)) return MOGAN_ERR_SHAPE;
        const long long lim = 1ll << 31;           // 32-bit element offsets inside every tensor
        for (int s = 0; s < a.nsrc; ++s)
            if (!a.src[s] || a.src_nsplit[s] <= 0 || a.src_bs[s] < 0 || a.src_bs[s] * a.B >= lim) return MOGAN_ERR_SHAPE;
        if ((a.add && a.add_bs * a.B >= lim) || (a.mask && a.mask_bs * a.B >= lim) || (a.dst && a.dst_bs * a.B >= lim) ||
            (long long)a.n * a.H * a.W >= lim)
            return MOGAN_ERR_SHAPE;
        if (a.panel && (a.cg0 < 0 || a.cg0 + (a.n + 31) / 32 > a.CGp || Q * a.CGp * 192 >= (1ll << 40))) return MOGAN_ERR_SHAPE;
        g.m[i] = a;
        end += ((Q + 63) / 64) * ((a.n + 31) / 32);
        if (end > 0x7fffffff) return MOGAN_ERR_SHAPE;
        g.end[i] = (unsigned)end;
    }
    bool box = false;
    for (int i = 0; i < n; ++i) box = box || args[i].box != 0;
    if (box) hipLaunchKernelGGL(panel_tail_kernel<true>, dim3((unsigned)end), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL(panel_tail_kernel<false>, dim3((unsigned)end), dim3(256), 0, stream, g);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

}  // extern "C"
