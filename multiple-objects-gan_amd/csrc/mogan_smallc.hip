// mogan_smallc.hip -- direct (VALU) convolution kernels for the layers with <= 4 channels on one side: the image heads
// of the generators (conv3x3 ngf -> 3, model.py:464-475 / S/model.py:196-198) and the first convolution of every
// discriminator (conv4x4 s2 3 -> ndf).  On the MFMA kernels these pad 3 channels to a 32-row tile (5-6 TFLOP/s, 0.2-0.5 ms
// each at 256x256) although they are pure HBM streaming work (200 MB read or written, ~50 us); three of them sit on the
// critical path of the step (end of the G forward, start of the G backward).
//
//   sc_fwd3x3      y (B,COUT<=4,H,W)  = conv3x3 p1 (x (B,Cin,H,W), w)            one thread = one output pixel
//   sc_dgrad3x3    dx (B,Cin,H,W)     = conv3x3^T (dy (B,COUT<=4,H,W), w)        one thread = one pixel, loops over Cin
//   sc_dgrad_k4s2  dx (B,CIN<=4,H,W)  = conv4x4s2p1^T (dy (B,Cout,H/2,W/2), w)   one thread = a 2x2 block of input pixels
// The few-channel operand is staged through LDS with its halo; the weights are read with block-uniform indices, i.e.
// as scalar loads.  fp32 FMA order: ci (or co) outer, taps inner -- sums of <= 27*Cin terms.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/mogan_hip.h"
#include "mogan_internal.h"

namespace {

constexpr int TR = 8, TC = 32;       // thread tile: 8 rows x 32 columns = 256 threads

// Halo tile (NC channels x HH x WW) of an NCHW image into LDS: ALL of a thread's global loads are issued first (clamped addresses, no
// branch), then stored -- a `for (e ...) Xs[..] = ok ? x[..] : 0` loop serialises one load latency per iteration, and with 11-15
// iterations per chunk that chain, not the arithmetic, was the run time of these kernels (80 us for a 64 x 64 image).
template <int NC, int NT, int HH, int WW>
__device__ __forceinline__ void halo_load(float (&v)[(NC * HH * WW + NT - 1) / NT], const float* __restrict__ base, size_t plane,
                                          int nvalid, int y0, int x0, int H, int W, int tid) {
    constexpr int TOT = NC * HH * WW, NE = (TOT + NT - 1) / NT;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int e = tid + NT * j;
        const int c = e / (HH * WW), r = e - c * (HH * WW);
        const int hy = r / WW, hx = r - hy * WW;
        const int iy = y0 + hy, ix = x0 + hx;
        const bool ok = e < TOT && c < nvalid && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const float t = base[ok ? (size_t)c * plane + (size_t)iy * W + ix : 0];
        v[j] = ok ? t : 0.f;
    }
}
template <int NC, int NT, int HH, int WW, int WP>
__device__ __forceinline__ void halo_store(float (*Xs)[HH][WP], const float (&v)[(NC * HH * WW + NT - 1) / NT], int tid) {
    constexpr int TOT = NC * HH * WW, NE = (TOT + NT - 1) / NT;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int e = tid + NT * j;
        const int c = e / (HH * WW), r = e - c * (HH * WW);
        const int hy = r / WW, hx = r - hy * WW;
        if (e < TOT) Xs[c][hy][hx] = v[j];
    }
}
template <int NC, int NT, int HH, int WW, int WP>
__device__ __forceinline__ void stage_halo(float (*Xs)[HH][WP], const float* __restrict__ base, size_t plane, int nvalid, int y0,
                                           int x0, int H, int W, int tid) {
    float v[(NC * HH * WW + NT - 1) / NT];
    halo_load<NC, NT, HH, WW>(v, base, plane, nvalid, y0, x0, H, W, tid);
    halo_store<NC, NT, HH, WW, WP>(Xs, v, tid);
}

template <int COUT>
__global__ __launch_bounds__(256) void sc_fwd3x3(const float* __restrict__ x, const float* __restrict__ w,
                                                 float* __restrict__ y, int Cin, int H, int W, int tiles_x, int tiles_y) {
    constexpr int CK = 16, HH = TR + 2, WW = TC + 2, WP = WW + 1;
    __shared__ float Xs[CK][HH][WP];
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int b = t / tiles_y;
    const int tid = threadIdx.x, ly = tid >> 5, lx = tid & 31;
    const int oy = ty * TR + ly, ox = tx * TC + lx;
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * Cin * plane;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    float pre[(CK * HH * WW + 255) / 256];            // the next chunk's halo: its loads fly while this chunk is computed
    halo_load<CK, 256, HH, WW>(pre, xb, plane, Cin, ty * TR - 1, tx * TC - 1, H, W, tid);
    for (int c0 = 0; c0 < Cin; c0 += CK) {
        halo_store<CK, 256, HH, WW, WP>(Xs, pre, tid);
        __syncthreads();
        if (c0 + CK < Cin)
            halo_load<CK, 256, HH, WW>(pre, xb + (size_t)(c0 + CK) * plane, plane, Cin - c0 - CK, ty * TR - 1, tx * TC - 1, H, W, tid);
        const int nc = min(CK, Cin - c0);
        for (int c = 0; c < nc; ++c) {
            const float* wc = w + (size_t)(c0 + c) * 9;            // + co * Cin * 9: block-uniform -> scalar loads
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float v = Xs[c][ly + kh][lx + kw];
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[co] = fmaf(v, wc[(size_t)co * Cin * 9 + kh * 3 + kw], acc[co]);
                }
        }
        __syncthreads();
    }
    if (oy < H && ox < W) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) y[((size_t)b * COUT + co) * plane + (size_t)oy * W + ox] = acc[co];
    }
}

// The same convolution on maps whose rows are multiples of 128 pixels (the 128 x 128 and 256 x 256 image heads): a thread owns FOUR
// consecutive pixels of a row, a block 8 rows x 128 columns.  Per channel and thread 9 LDS reads (a scalar, an aligned 16-byte
// quad, a scalar per halo row) feed 108 FMAs -- the one-pixel form issues 36 reads for them and is bound by instruction issue
// (141 us at B = 16, 256 x 256, where the 201 MB of input are a 45 us read); the halo is staged as aligned 16-byte quads (columns
// x0 - 4 .. x0 + 131), a quarter of the load / store instructions per byte.
template <int COUT>
__global__ __launch_bounds__(256) void sc_fwd3x3_w4(const float* __restrict__ x, const float* __restrict__ w,
                                                    float* __restrict__ y, int Cin, int H, int W, int tiles_x, int tiles_y) {
    constexpr int CK = 8, HH = TR + 2, TW = 128, NQ = TW / 4 + 2, WP = NQ * 4 + 4;      // 34 quads per halo row, padded pitch
    constexpr int NL = (CK * HH * NQ + 255) / 256;
    __shared__ __attribute__((aligned(16))) float Xs[CK][HH][WP];
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int b = t / tiles_y;
    const int tid = threadIdx.x, ly = tid >> 5, lx = tid & 31;
    const int oy = ty * TR + ly, x0 = tx * TW;
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * Cin * plane;
    float4 pre[NL];
    auto halo_load4 = [&](int c0) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = tid + 256 * j;
            const int c = e / (HH * NQ), r = e - c * (HH * NQ);
            const int hy = r / NQ, q = r - hy * NQ;
            const int iy = ty * TR - 1 + hy, ix = x0 - 4 + 4 * q;
            const bool ok = e < CK * HH * NQ && c0 + c < Cin && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const float4 v = *(const float4*)(xb + (ok ? (size_t)(c0 + c) * plane + (size_t)iy * W + ix : 0));
            pre[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto halo_store4 = [&]() {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = tid + 256 * j;
            const int c = e / (HH * NQ), r = e - c * (HH * NQ);
            const int hy = r / NQ, q = r - hy * NQ;
            if (e < CK * HH * NQ) *(float4*)&Xs[c][hy][4 * q] = pre[j];
        }
    };
    float acc[COUT][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[co][p] = 0.f;
    float wcur[COUT * 9];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int k = 0; k < 9; ++k) wcur[co * 9 + k] = w[(size_t)co * Cin * 9 + k];
    halo_load4(0);
    for (int c0 = 0; c0 < Cin; c0 += CK) {
        halo_store4();
        __syncthreads();
        if (c0 + CK < Cin) halo_load4(c0 + CK);
        const int nc = min(CK, Cin - c0);
        for (int c = 0; c < nc; ++c) {
            float wn[COUT * 9];
            const int cn = min(c0 + c + 1, Cin - 1);
#pragma unroll
            for (int co = 0; co < COUT; ++co)
#pragma unroll
                for (int k = 0; k < 9; ++k) wn[co * 9 + k] = w[((size_t)co * Cin + cn) * 9 + k];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                // pixel p of this thread is halo column 4 lx + 4 + p; its taps are columns 4 lx + 3 + p + kw
                const float* row = &Xs[c][ly + kh][4 * lx];
                const float4 m = *(const float4*)(row + 4);
                const float v[6] = {row[3], m.x, m.y, m.z, m.w, row[8]};
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int co = 0; co < COUT; ++co)
#pragma unroll
                        for (int p = 0; p < 4; ++p) acc[co][p] = fmaf(v[p + kw], wcur[co * 9 + kh * 3 + kw], acc[co][p]);
            }
#pragma unroll
            for (int k = 0; k < COUT * 9; ++k) wcur[k] = wn[k];
        }
        __syncthreads();
    }
    if (oy < H) {
#pragma unroll
        for (int co = 0; co < COUT; ++co)
            *(float4*)(y + ((size_t)b * COUT + co) * plane + (size_t)oy * W + x0 + 4 * lx) =
                make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void sc_dgrad3x3(const float* __restrict__ dy, const float* __restrict__ w,
                                                   float* __restrict__ dx, int Cin, int H, int W, int tiles_x, int tiles_y) {
    constexpr int HH = TR + 2, WW = TC + 2, WP = WW + 1;
    __shared__ float Ys[COUT][HH][WP];
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int b = t / tiles_y;
    const int tid = threadIdx.x, ly = tid >> 5, lx = tid & 31;
    const int iy = ty * TR + ly, ix = tx * TC + lx;
    const size_t plane = (size_t)H * W;
    stage_halo<COUT, 256, HH, WW, WP>(Ys, dy + (size_t)b * COUT * plane, plane, COUT, ty * TR - 1, tx * TC - 1, H, W, tid);
    __syncthreads();
    // dx[iy][ix] = sum_co sum_{kh,kw} dy[iy+1-kh][ix+1-kw] * w[co][ci][kh][kw];  LDS position of dy[iy+1-kh] is ly+2-kh
    float r[COUT][9];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) r[co][kh * 3 + kw] = Ys[co][ly + 2 - kh][lx + 2 - kw];
    const bool inside = iy < H && ix < W;
    float* out = dx + (size_t)b * Cin * plane + (size_t)iy * W + ix;
    for (int ci = 0; ci < Cin; ++ci) {
        float s = 0.f;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float* wc = w + ((size_t)co * Cin + ci) * 9;     // block-uniform -> scalar loads
#pragma unroll
            for (int k = 0; k < 9; ++k) s = fmaf(r[co][k], wc[k], s);
        }
        if (inside) out[(size_t)ci * plane] = s;
    }
}

// one thread = the 2x2 block of input pixels (2Y..2Y+1, 2X..2X+1); it needs dy rows Y-1..Y+1, cols X-1..X+1 of every co:
//   iy = 2Y   (even): kh = 1 -> oy = Y,   kh = 3 -> oy = Y-1        iy = 2Y+1 (odd): kh = 0 -> oy = Y+1, kh = 2 -> oy = Y
template <int CIN>
__global__ __launch_bounds__(256) void sc_dgrad_k4s2(const float* __restrict__ dy, const float* __restrict__ w,
                                                     float* __restrict__ dx, int Cout, int H, int W, int OH, int OW,
                                                     int tiles_x, int tiles_y) {
    constexpr int CK = 8, HH = TR + 2, WW = TC + 2, WP = WW + 1;
    __shared__ float Ys[CK][HH][WP];
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int b = t / tiles_y;
    const int tid = threadIdx.x, ly = tid >> 5, lx = tid & 31;
    const int Y = ty * TR + ly, X = tx * TC + lx;                  // coordinates on the dy grid
    const size_t oplane = (size_t)OH * OW;
    float acc[CIN][2][2];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) acc[ci][0][0] = acc[ci][0][1] = acc[ci][1][0] = acc[ci][1][1] = 0.f;
    float pre[(CK * HH * WW + 255) / 256];            // the next chunk's halo: its loads fly while this chunk is computed
    halo_load<CK, 256, HH, WW>(pre, dy + (size_t)b * Cout * oplane, oplane, Cout, ty * TR - 1, tx * TC - 1, OH, OW, tid);
    for (int c0 = 0; c0 < Cout; c0 += CK) {
        halo_store<CK, 256, HH, WW, WP>(Ys, pre, tid);
        __syncthreads();
        if (c0 + CK < Cout)
            halo_load<CK, 256, HH, WW>(pre, dy + ((size_t)b * Cout + c0 + CK) * oplane, oplane, Cout - c0 - CK, ty * TR - 1, tx * TC - 1, OH,
                                       OW, tid);
        const int nc = min(CK, Cout - c0);
        for (int c = 0; c < nc; ++c) {
            float d[3][3];                                          // d[a][bb] = dy[Y-1+a][X-1+bb]
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int bb = 0; bb < 3; ++bb) d[a][bb] = Ys[c][ly + a][lx + bb];
            const float* wc = w + (size_t)(c0 + c) * CIN * 16;       // block-uniform -> scalar loads
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float* k = wc + ci * 16;
                // rows: py = 0 (iy even): (kh=1, oy=Y -> a=1), (kh=3, oy=Y-1 -> a=0);  py = 1: (kh=0, a=2), (kh=2, a=1)
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        const int kh0 = py ? 0 : 1, a0 = py ? 2 : 1, kh1 = py ? 2 : 3, a1 = py ? 1 : 0;
                        const int kw0 = px ? 0 : 1, b0 = px ? 2 : 1, kw1 = px ? 2 : 3, b1 = px ? 1 : 0;
                        float s = acc[ci][py][px];
                        s = fmaf(d[a0][b0], k[kh0 * 4 + kw0], s);
                        s = fmaf(d[a0][b1], k[kh0 * 4 + kw1], s);
                        s = fmaf(d[a1][b0], k[kh1 * 4 + kw0], s);
                        s = fmaf(d[a1][b1], k[kh1 * 4 + kw1], s);
                        acc[ci][py][px] = s;
                    }
            }
        }
        __syncthreads();
    }
    const size_t plane = (size_t)H * W;
    if (Y < OH && X < OW) {
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                float* o = dx + ((size_t)b * CIN + ci) * plane + (size_t)(2 * Y + py) * W + 2 * X;
                *(float2*)o = make_float2(acc[ci][py][0], acc[ci][py][1]);
            }
    }
}


// dW (COUT<=4,Cin,3,3) = sum over (b, y, x) of dy[b,co,y,x] * x[b,ci,y+kh-1,x+kw-1].  A block owns a band of TR rows of one
// image and walks its TC-wide tiles; per chunk of CK input channels, thread (c, kh, r) slides along row r of the tile with a
// 3-wide window of x[c][r+kh-1][.] (one LDS read per pixel) against the COUT dy values of the pixel (LDS broadcasts):
// 3*COUT FMAs per 1+COUT LDS reads.  The 8 row-threads are summed in LDS, the band's partial (Cin*9*COUT values) goes to
// the workspace and sc_wgrad_reduce adds the bands in a fixed order (deterministic, no atomics).
template <int COUT>
__global__ __launch_bounds__(192) void sc_wgrad3x3(const float* __restrict__ dy, const float* __restrict__ x,
                                                   float* __restrict__ part, int Cin, int H, int W, int tiles_x,
                                                   int tiles_y) {
    constexpr int CK = 8, HH = TR + 2, WW = TC + 2, WP = WW + 1;
    __shared__ float Xs[CK][HH][WP];
    __shared__ float Ys[COUT][TR][TC + 1];
    __shared__ float Rs[CK * 3][TR][3 * COUT + 1];
    const int ty = blockIdx.x % tiles_y, b = blockIdx.x / tiles_y;        // blockIdx.y = chunk of CK input channels
    const int tid = threadIdx.x;
    const int r = tid & 7, kh = (tid >> 3) % 3, c = tid / 24;
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * Cin * plane;
    const float* yb = dy + (size_t)b * COUT * plane;
    float* out = part + (size_t)blockIdx.x * Cin * 9 * COUT;
    {
        const int c0 = blockIdx.y * CK;
        float acc[3][COUT];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[kw][co] = 0.f;
        for (int tx = 0; tx < tiles_x; ++tx) {
            {
                constexpr int NY = (COUT * TR * TC + 191) / 192;
                float yv[NY];
#pragma unroll
                for (int j = 0; j < NY; ++j) {                       // (issued ahead of the halo's loads: all in flight together)
                    const int e = tid + 192 * j;
                    const int co = e / (TR * TC), q = e - co * (TR * TC);
                    const int py = q / TC, px = q - py * TC;
                    const int oy = ty * TR + py, ox = tx * TC + px;
                    const bool ok = e < COUT * TR * TC && oy < H && ox < W;
                    const float t = yb[ok ? (size_t)co * plane + (size_t)oy * W + ox : 0];
                    yv[j] = ok ? t : 0.f;
                }
                stage_halo<CK, 192, HH, WW, WP>(Xs, xb + (size_t)c0 * plane, plane, Cin - c0, ty * TR - 1, tx * TC - 1, H, W, tid);
#pragma unroll
                for (int j = 0; j < NY; ++j) {
                    const int e = tid + 192 * j;
                    const int co = e / (TR * TC), q = e - co * (TR * TC);
                    const int py = q / TC, px = q - py * TC;
                    if (e < COUT * TR * TC) Ys[co][py][px] = yv[j];
                }
            }
            __syncthreads();
            const float* xr = &Xs[c][r + kh][0];
            float x0 = xr[0], x1 = xr[1];
#pragma unroll 8
            for (int px = 0; px < TC; ++px) {
                const float x2 = xr[px + 2];
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const float d = Ys[co][r][px];
                    acc[0][co] = fmaf(d, x0, acc[0][co]);
                    acc[1][co] = fmaf(d, x1, acc[1][co]);
                    acc[2][co] = fmaf(d, x2, acc[2][co]);
                }
                x0 = x1; x1 = x2;
            }
            __syncthreads();
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int co = 0; co < COUT; ++co) Rs[c * 3 + kh][r][kw * COUT + co] = acc[kw][co];
        __syncthreads();
        for (int e = tid; e < CK * 3 * 3 * COUT; e += 192) {
            const int g = e / (3 * COUT), v = e - g * (3 * COUT);          // g = cc*3 + kh, v = kw*COUT + co
            const int cc = g / 3, k = g - cc * 3, kw = v / COUT, co = v - kw * COUT;
            float s = 0.f;
#pragma unroll
            for (int rr = 0; rr < TR; ++rr) s += Rs[g][rr][v];
            if (c0 + cc < Cin) out[((size_t)co * Cin + c0 + cc) * 9 + k * 3 + kw] = s;
        }
    }
}

// dw[i] = (acc ? dw[i] : 0) + sum_blk part[blk][i]: one wave per output, lane l sums blk = l, l+64, ... in order, then a
// fixed shuffle tree
__global__ __launch_bounds__(256) void sc_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int n,
                                                       int nblk, int accumulate) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    float s = 0.f;
    for (int k = lane; k < nblk; k += 64) s += part[(size_t)k * n + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) dw[i] = (accumulate ? dw[i] : 0.f) + s;
}


// dx (B,CIN<=4,H,W) = conv3x3 s2 p0 ^T (dy (B,Cout,OH,OW), w): the data gradient of the Inception trunk's first convolution
// (3 -> 32 at 299x299).  One thread = the 2x2 block of input pixels (2Y..2Y+1, 2X..2X+1); it needs dy rows Y-1, Y and
// columns X-1, X of every co:   even row 2Y: kh = 0 -> oy = Y, kh = 2 -> oy = Y-1;   odd row 2Y+1: kh = 1 -> oy = Y.
template <int CIN>
__global__ __launch_bounds__(256) void sc_dgrad_k3s2(const float* __restrict__ dy, const float* __restrict__ w,
                                                     float* __restrict__ dx, int Cout, int H, int W, int OH, int OW,
                                                     int tiles_x, int tiles_y) {
    constexpr int CK = 8, HH = TR + 1, WW = TC + 1, WP = WW + 2;
    __shared__ float Ys[CK][HH][WP];
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int b = t / tiles_y;
    const int tid = threadIdx.x, ly = tid >> 5, lx = tid & 31;
    const int Y = ty * TR + ly, X = tx * TC + lx;
    const size_t oplane = (size_t)OH * OW;
    float acc[CIN][2][2];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) acc[ci][0][0] = acc[ci][0][1] = acc[ci][1][0] = acc[ci][1][1] = 0.f;
    float pre[(CK * HH * WW + 255) / 256];            // the next chunk's halo: its loads fly while this chunk is computed
    halo_load<CK, 256, HH, WW>(pre, dy + (size_t)b * Cout * oplane, oplane, Cout, ty * TR - 1, tx * TC - 1, OH, OW, tid);
    for (int c0 = 0; c0 < Cout; c0 += CK) {
        halo_store<CK, 256, HH, WW, WP>(Ys, pre, tid);
        __syncthreads();
        if (c0 + CK < Cout)
            halo_load<CK, 256, HH, WW>(pre, dy + ((size_t)b * Cout + c0 + CK) * oplane, oplane, Cout - c0 - CK, ty * TR - 1, tx * TC - 1, OH,
                                       OW, tid);
        const int nc = min(CK, Cout - c0);
        for (int c = 0; c < nc; ++c) {
            const float d00 = Ys[c][ly][lx], d01 = Ys[c][ly][lx + 1];        // (Y-1, X-1), (Y-1, X)
            const float d10 = Ys[c][ly + 1][lx], d11 = Ys[c][ly + 1][lx + 1];  // (Y,   X-1), (Y,   X)
            const float* wc = w + (size_t)(c0 + c) * CIN * 9;                  // block-uniform -> scalar loads
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float* k = wc + ci * 9;
                acc[ci][0][0] += d11 * k[0] + d10 * k[2] + d01 * k[6] + d00 * k[8];
                acc[ci][0][1] += d11 * k[1] + d01 * k[7];
                acc[ci][1][0] += d11 * k[3] + d10 * k[5];
                acc[ci][1][1] += d11 * k[4];
            }
        }
        __syncthreads();
    }
    const size_t plane = (size_t)H * W;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int iy = 2 * Y + py, ix = 2 * X + px;
                if (iy < H && ix < W) dx[((size_t)b * CIN + ci) * plane + (size_t)iy * W + ix] = acc[ci][py][px];
            }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- full-map convolutions with <= 4 output channels (the discriminators' logits heads, model.py:664-680: Conv2d(8 ndf, 1,
// kernel_size=4, stride=4) on a 4x4 map): the filter covers the whole input, the output is 1x1 -- per image one dot product
// of K = Cin*KH*KW elements.  As an implicit GEMM this is M = 1, N = B, K = 12288: a 96-way split-K plus a reduction per call.
__global__ __launch_bounds__(256) void sc_dot_fwd(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                  int K, int Cout) {
    const int b = blockIdx.x / Cout, co = blockIdx.x % Cout;
    const float* xb = x + (size_t)b * K; const float* wc = w + (size_t)co * K;
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) acc = fmaf(xb[k], wc[k], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) y[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
// dx[b][k] = sum_co dy[b][co] w[co][k]
__global__ __launch_bounds__(256) void sc_dot_dgrad(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                    int K, int Cout, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / K), k = (int)(i - (long long)b * K);
    float acc = 0.f;
    for (int co = 0; co < Cout; ++co) acc = fmaf(dy[b * Cout + co], w[(size_t)co * K + k], acc);
    dx[i] = acc;
}
// dw[co][k] (+)= sum_b dy[b][co] x[b][k]   (images in order: deterministic)
__global__ __launch_bounds__(256) void sc_dot_wgrad(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
                                                    int K, int Cout, int B, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cout * K) return;
    const int co = i / K, k = i - co * K;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc = fmaf(dy[b * Cout + co], x[(size_t)b * K + k], acc);
    dw[i] = accumulate ? dw[i] + acc : acc;
}
// ---- the whole logits head in one launch each way (round 4): p[b][co] = sigmoid(<x[b], w[co]> + bias[co]) -- Conv2d(8 ndf, 1,
// kernel_size=4, stride=4) + bias + nn.Sigmoid on a 4x4 map (model.py:626-627, 640-641) were three launches forward (dot,
// bias_add, sigmoid) and four backward on 16-element results, 21 times per step.  K % 4 == 0, 16-byte aligned rows.
__global__ __launch_bounds__(256) void logits_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ p, int K,
                                                              int Cout) {
    const int b = blockIdx.x / Cout, co = blockIdx.x % Cout;
    const float4* xb = (const float4*)(x + (size_t)b * K); const float4* wc = (const float4*)(w + (size_t)co * K);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int k = threadIdx.x; k < K / 4; k += 256) {       // fixed order per thread; four independent chains
        const float4 u = xb[k], v = wc[k];
        a0 = fmaf(u.x, v.x, a0); a1 = fmaf(u.y, v.y, a1); a2 = fmaf(u.z, v.z, a2); a3 = fmaf(u.w, v.w, a3);
    }
    float acc = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float z = (part[0] + part[1]) + (part[2] + part[3]) + (bias ? bias[co] : 0.f);
        p[blockIdx.x] = 1.f / (1.f + __expf(-z));
    }
}
// backward: dz[b][co] = dp * p * (1 - p) on the fly.  Blocks [0, nbx): dx[b][k] = sum_co dz[b][co] w[co][k] (when dx != NULL);
// the next nbw: dw[co][k] (+)= sum_b dz[b][co] x[b][k] (images in order); one more block when db != NULL: db[co] (a bias gradient
// without a weight gradient -- frozen filter, trainable bias -- is its own block, not a rider of the dw blocks).
__global__ __launch_bounds__(256) void logits_head_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ p,
                                                              const float* __restrict__ x, const float* __restrict__ w,
                                                              float* __restrict__ dx, float* __restrict__ dw,
                                                              float* __restrict__ db, int B, int K, int Cout, int accumulate,
                                                              unsigned nbx, unsigned nbw) {
    if (blockIdx.x < nbx) {
        const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        if (i >= (long long)B * K) return;
        const int b = (int)(i / K), k = (int)(i - (long long)b * K);
        float acc = 0.f;
        for (int co = 0; co < Cout; ++co) {
            const float q = p[b * Cout + co];
            acc = fmaf(dp[b * Cout + co] * q * (1.f - q), w[(size_t)co * K + k], acc);
        }
        dx[i] = acc;
        return;
    }
    if (blockIdx.x < nbx + nbw) {
        const int i = (blockIdx.x - nbx) * 256 + threadIdx.x;
        if (i < Cout * K) {
            const int co = i / K, k = i - co * K;
            float acc = 0.f;
            for (int b = 0; b < B; ++b) {
                const float q = p[b * Cout + co];
                acc = fmaf(dp[b * Cout + co] * q * (1.f - q), x[(size_t)b * K + k], acc);
            }
            dw[i] = accumulate ? dw[i] + acc : acc;
        }
        return;
    }
    if (db != nullptr && threadIdx.x < Cout) {
        const int co = threadIdx.x;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) { const float q = p[b * Cout + co]; acc += dp[b * Cout + co] * q * (1.f - q); }
        db[co] = accumulate ? db[co] + acc : acc;
    }
}

static inline bool is_full_map(int Hs, int Ws, int Cout, int KH, int KW, int ph, int pw, int up) {
    return up == 0 && ph == 0 && pw == 0 && KH == Hs && KW == Ws && Cout >= 1 && Cout <= 4;
}

}  // namespace

// ---- internal entry points (hidden visibility): 1 = handled, 0 = not eligible ------------------------------------
int mogan_smallc_fwd_try(const float* x, const float* w, float* y, int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW,
                         int stride, int ph, int pw, int up, hipStream_t st) {
    if (is_full_map(Hs, Ws, Cout, KH, KW, ph, pw, up) && (long long)Cin * KH * KW < (1ll << 30)) {
        hipLaunchKernelGGL(sc_dot_fwd, dim3((unsigned)(B * Cout)), dim3(256), 0, st, x, w, y, Cin * KH * KW, Cout);
        return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
    }
    if (!(KH == 3 && KW == 3 && stride == 1 && ph == 1 && pw == 1 && up == 0 && Cout >= 1 && Cout <= 4)) return 0;
    constexpr int w4 = 1;
    if (w4 && (Ws % 128) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {          // four pixels per thread, 8 x 128 tiles
        const int txs = Ws / 128, tys = cdiv(Hs, TR);
        const long long nb4 = (long long)B * txs * tys;
        if (nb4 <= 0x7fffffffLL) {
            dim3 grid4((unsigned)nb4);
            switch (Cout) {
                case 1: hipLaunchKernelGGL(sc_fwd3x3_w4<1>, grid4, dim3(256), 0, st, x, w, y, Cin, Hs, Ws, txs, tys); break;
                case 2: hipLaunchKernelGGL(sc_fwd3x3_w4<2>, grid4, dim3(256), 0, st, x, w, y, Cin, Hs, Ws, txs, tys); break;
                case 3: hipLaunchKernelGGL(sc_fwd3x3_w4<3>, grid4, dim3(256), 0, st, x, w, y, Cin, Hs, Ws, txs, tys); break;
                default: hipLaunchKernelGGL(sc_fwd3x3_w4<4>, grid4, dim3(256), 0, st, x, w, y, Cin, Hs, Ws, txs, tys); break;
            }
            return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
        }
    }
    const int tiles_x = cdiv(Ws, TC), tiles_y = cdiv(Hs, TR);
    const long long nb = (long long)B * tiles_x * tiles_y;
    if (nb > 0x7fffffffLL) return 0;
    dim3 grid((unsigned)nb);
    switch (Cout) {
        case 1: hipLaunchKernelGGL(sc_fwd3x3<1>, grid, dim3(256), 0, st, x, w, y, Cin, Hs, Ws, tiles_x, tiles_y); break;
        case 2: hipLaunchKernelGGL(sc_fwd3x3<2>, grid, dim3(256), 0, st, x, w, y, Cin, Hs, Ws, tiles_x, tiles_y); break;
        case 3: hipLaunchKernelGGL(sc_fwd3x3<3>, grid, dim3(256), 0, st, x, w, y, Cin, Hs, Ws, tiles_x, tiles_y); break;
        default: hipLaunchKernelGGL(sc_fwd3x3<4>, grid, dim3(256), 0, st, x, w, y, Cin, Hs, Ws, tiles_x, tiles_y); break;
    }
    return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
}

int mogan_smallc_dgrad_try(const float* dy, const float* w, float* dx, int B, int Cin, int Hs, int Ws, int Cout, int KH,
                           int KW, int stride, int ph, int pw, int up, hipStream_t st) {
    if (is_full_map(Hs, Ws, Cout, KH, KW, ph, pw, up) && (long long)Cin * KH * KW < (1ll << 30)) {
        const int K = Cin * KH * KW;
        const long long total = (long long)B * K;
        hipLaunchKernelGGL(sc_dot_dgrad, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, w, dx, K, Cout, total);
        return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
    }
    if (up == 0 && ph == 0 && pw == 0 && KH == 3 && KW == 3 && stride == 2 && Cin >= 1 && Cin <= 4 && Hs >= 3 && Ws >= 3) {
        const int OH = (Hs - 3) / 2 + 1, OW = (Ws - 3) / 2 + 1;
        const int tiles_x = cdiv(cdiv(Ws, 2), TC), tiles_y = cdiv(cdiv(Hs, 2), TR);
        const long long nb = (long long)B * tiles_x * tiles_y;
        if (nb > 0x7fffffffLL) return 0;
        dim3 grid((unsigned)nb);
        switch (Cin) {
            case 1: hipLaunchKernelGGL(sc_dgrad_k3s2<1>, grid, dim3(256), 0, st, dy, w, dx, Cout, Hs, Ws, OH, OW, tiles_x, tiles_y); break;
            case 2: hipLaunchKernelGGL(sc_dgrad_k3s2<2>, grid, dim3(256), 0, st, dy, w, dx, Cout, Hs, Ws, OH, OW, tiles_x, tiles_y); break;
            case 3: hipLaunchKernelGGL(sc_dgrad_k3s2<3>, grid, dim3(256), 0, st, dy, w, dx, Cout, Hs, Ws, OH, OW, tiles_x, tiles_y); break;
            default: hipLaunchKernelGGL(sc_dgrad_k3s2<4>, grid, dim3(256), 0, st, dy, w, dx, Cout, Hs, Ws, OH, OW, tiles_x, tiles_y); break;
        }
        return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
    }
    if (up != 0 || ph != 1 || pw != 1) return 0;
    if (KH == 3 && KW == 3 && stride == 1 && Cout >= 1 && Cout <= 4) {
        const int tiles_x = cdiv(Ws, TC), tiles_y = cdiv(Hs, TR);
        const long long nb = (long long)B * tiles_x * tiles_y;
        if (nb > 0x7fffffffLL) return 0;
        dim3 grid((unsigned)nb);
        switch (Cout) {
            case 1: hipLaunchKernelGGL(sc_dgrad3x3<1>, grid, dim3(256), 0, st, dy, w, dx, Cin, Hs, Ws, tiles_x, tiles_y); break;
            case 2: hipLaunchKernelGGL(sc_dgrad3x3<2>, grid, dim3(256), 0, st, dy, w, dx, Cin, Hs, Ws, tiles_x, tiles_y); break;
            case 3: hipLaunchKernelGGL(sc_dgrad3x3<3>, grid, dim3(256), 0, st, dy, w, dx, Cin, Hs, Ws, tiles_x, tiles_y); break;
            default: hipLaunchKernelGGL(sc_dgrad3x3<4>, grid, dim3(256), 0, st, dy, w, dx, Cin, Hs, Ws, tiles_x, tiles_y); break;
        }
        return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
    }
    if (KH == 4 && KW == 4 && stride == 2 && Cin >= 1 && Cin <= 4 && (Hs % 2) == 0 && (Ws % 2) == 0 &&
        (((uintptr_t)dx) & 7) == 0) {
        const int OH = Hs / 2, OW = Ws / 2;
        const int tiles_x = cdiv(OW, TC), tiles_y = cdiv(OH, TR);
        const long long nb = (long long)B * tiles_x * tiles_y;
        if (nb > 0x7fffffffLL) return 0;
        dim3 grid((unsigned)nb);
        switch (Cin) {
            case 1: hipLaunchKernelGGL(sc_dgrad_k4s2<1>, grid, dim3(256), 0, st, dy, w, dx, Cout, Hs, Ws, OH, OW, tiles_x, tiles_y); break;
            case 2: hipLaunchKernelGGL(sc_dgrad_k4s2<2>, grid, dim3(256), 0, st, dy, w, dx, Cout, Hs, Ws, OH, OW, tiles_x, tiles_y); break;
            case 3: hipLaunchKernelGGL(sc_dgrad_k4s2<3>, grid, dim3(256), 0, st, dy, w, dx, Cout, Hs, Ws, OH, OW, tiles_x, tiles_y); break;
            default: hipLaunchKernelGGL(sc_dgrad_k4s2<4>, grid, dim3(256), 0, st, dy, w, dx, Cout, Hs, Ws, OH, OW, tiles_x, tiles_y); break;
        }
        return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
    }
    return 0;
}

int mogan_smallc_wgrad_try(const float* dy, const float* x, float* dw, int B, int Cin, int Hs, int Ws, int Cout, int KH,
                           int KW, int stride, int ph, int pw, int up, int accumulate, void* ws, size_t ws_bytes,
                           hipStream_t st) {
    if (is_full_map(Hs, Ws, Cout, KH, KW, ph, pw, up) && (long long)Cout * Cin * KH * KW < (1ll << 30)) {
        const int K = Cin * KH * KW;
        hipLaunchKernelGGL(sc_dot_wgrad, dim3((unsigned)cdiv(Cout * K, 256)), dim3(256), 0, st, dy, x, dw, K, Cout, B, accumulate);
        return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
    }
    if (!(KH == 3 && KW == 3 && stride == 1 && ph == 1 && pw == 1 && up == 0 && Cout >= 1 && Cout <= 4)) return 0;
    const int tiles_x = cdiv(Ws, TC), tiles_y = cdiv(Hs, TR);
    const long long nblk = (long long)B * tiles_y;
    const int n = Cout * Cin * 9;
    if (nblk > 65535 || !ws || ws_bytes < (size_t)nblk * n * sizeof(float)) return 0;
    float* part = (float*)ws;
    dim3 grid((unsigned)nblk, (unsigned)cdiv(Cin, 8));
    switch (Cout) {
        case 1: hipLaunchKernelGGL(sc_wgrad3x3<1>, grid, dim3(192), 0, st, dy, x, part, Cin, Hs, Ws, tiles_x, tiles_y); break;
        case 2: hipLaunchKernelGGL(sc_wgrad3x3<2>, grid, dim3(192), 0, st, dy, x, part, Cin, Hs, Ws, tiles_x, tiles_y); break;
        case 3: hipLaunchKernelGGL(sc_wgrad3x3<3>, grid, dim3(192), 0, st, dy, x, part, Cin, Hs, Ws, tiles_x, tiles_y); break;
        default: hipLaunchKernelGGL(sc_wgrad3x3<4>, grid, dim3(192), 0, st, dy, x, part, Cin, Hs, Ws, tiles_x, tiles_y); break;
    }
    hipLaunchKernelGGL(sc_wgrad_reduce, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, st, (const float*)part, dw, n, (int)nblk,
                       accumulate);
    return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
}

// ---- C ABI: the discriminators' logits head ---------------------------------------------------------------------------------
extern "C" {

int mogan_logits_head_fwd(const float* x, const float* w, const float* bias, float* p, int B, int K, int Cout,
                          hipStream_t stream) {
    if (B <= 0 || K <= 0 || (K & 3) || Cout < 1 || Cout > 4 || ((((uintptr_t)x) | ((uintptr_t)w)) & 15)) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(logits_head_fwd_kernel, dim3((unsigned)(B * Cout)), dim3(256), 0, stream, x, w, bias, p, K, Cout);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

int mogan_logits_head_bwd(const float* dp, const float* p, const float* x, const float* w, float* dx, float* dw, float* db,
                          int B, int K, int Cout, int accumulate, hipStream_t stream) {
    if (B <= 0 || K <= 0 || Cout < 1 || Cout > 4 || (long long)B * K >= (1ll << 31)) return MOGAN_ERR_SHAPE;
    const unsigned nbx = dx ? (unsigned)(((long long)B * K + 255) / 256) : 0u;
    const unsigned nbw = dw ? (unsigned)((Cout * K + 255) / 256) : 0u;
    const unsigned nbb = db ? 1u : 0u;
    if (nbx + nbw + nbb == 0) return 0;
    hipLaunchKernelGGL(logits_head_bwd_kernel, dim3(nbx + nbw + nbb), dim3(256), 0, stream, dp, p, x, w, dx, dw, db, B, K, Cout,
                       accumulate, nbx, nbw);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

}  // extern "C"
