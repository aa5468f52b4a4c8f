// mogan_wino22.hip -- Winograd F(2x2, 2x2) for the 4x4 stride-2 pad-1 convolutions (every down-convolution of the
// discriminators, model.py:587-613).  A 4x4 s2 convolution is the sum over the four input phases (p,q) of a 2x2 stride-1
// convolution of the phase image x_pq[i][j] = x[2i+p-1][2j+q-1] with the filter g_pq[a][b] = w[2a+p][2b+q]; each of those is
// evaluated as  Y = A^t [ (G g G^t) .* (B^t d B) ] A  on 3x3 tiles (9 multiplies per 2x2 outputs instead of 16):
//   B^t = [[1,-1,0],[0,1,0],[0,-1,1]]   G = [[1,0],[1,1],[0,1]]   A^t = [[1,1,0],[0,1,1]]       (only +-1: exact adds)
// so the whole convolution is 9 GEMMs  M_xi[co][tile] = sum_{ci,p,q} U_xi[co][ci,p,q] V_xi[ci,p,q][tile]  with K = 4 Cin.
//
// One kernel, same structure as mogan_wino.hip: a block (12 waves) owns 128 output channels x 32 tiles (2x2 outputs each,
// consecutive in (image, tile row, tile column) order -- any even output size, a block may span several small images) for a
// range of K-chunks (4 input channels x 4 phases = 16 k); the 9 x 4 (position, m-tile) units are dealt three per wave.
// Per chunk every tile's 6x6 input window goes to LDS (Xs[ci][6][6][tile]), 512 threads transform one (k-channel, tile)
// patch each into Vs[xi][k][tile], and the weights -- pre-transformed by wino22_weight_kernel into the order the MFMA lanes
// consume them -- come straight from global memory into registers, one chunk ahead.  LDS images double buffered, one barrier
// per chunk, persistent blocks.  Output transform through LDS (Ts[xi][m][tile], aliased with the staging buffers).  With
// nsplit > 1 (deep layers: few tiles, long K) every K-split writes its partial output to a workspace slab and
// wino22_reduce sums the slabs in order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include "../../include/mogan_hip.h"
#include "mogan_internal.h"
#include "mogan_mma.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CI = 4, KC = 4 * CI, NT = 32, BM = 128, NWAVE = 12, NTHR = 64 * NWAVE;
constexpr int XSZ = CI * 36 * NT, VSZ = 9 * KC * NT, TSZ = 9 * BM * NT;

__device__ __forceinline__ float ldg1(__amdgpu_buffer_rsrc_t r, unsigned idx) {          // idx 0x30000000 -> 0.f
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, idx * 4u, 0, 0));
}
__device__ __forceinline__ f32x4 ldg4(__amdgpu_buffer_rsrc_t r, unsigned idx) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, idx * 4u, 0, 0));
}

// U2[mb][wave][chunk][i = 0..5][lane][e = 0..3], v = uu*8 + kk = 4i + e: the A operand of unit u = 3*wave + uu (position
// xi = u >> 2, m-tile u & 3) at k-step kk for lane (h, l31) = U_xi[co = mb*128 + 32*(u&3) + l31][k = 2kk + h], k = ci*4 + 2p + q.
// One thread per 16-byte group of U2 (coalesced stores; the up to four raw taps of a value come through the caches):
// U_(a,b) of phase (p,q) = sum over a' in S(a), b' in S(b) of w[2a'+p][2b'+q],  S(0) = {0}, S(1) = {0,1}, S(2) = {1}.
__global__ __launch_bounds__(256) void wino22_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout,
                                                            int Cin, long long ngroups) {
    const long long gidx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gidx >= ngroups) return;
    const int nchunk = Cin / CI;
    const int lane = (int)(gidx & 63); long long r = gidx >> 6;
    const int i = (int)(r % 6); r /= 6;
    const int chunk = (int)(r % nchunk); r /= nchunk;
    const int wv = (int)(r % NWAVE); const int mb = (int)(r / NWAVE);
    const int h = lane >> 5, l31 = lane & 31;
    f32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int v = 4 * i + e, uu = v >> 3, kk = v & 7;
        const int unit = 3 * wv + uu, xi = unit >> 2, mt = unit & 3, a = xi / 3, b = xi - a * 3;
        const int co = mb * BM + mt * 32 + l31;
        const int k = 2 * kk + h, ci = chunk * CI + (k >> 2), p = (k >> 1) & 1, q = k & 1;
        float u = 0.f;
        if (co < Cout) {
            const float* g = w + ((size_t)co * Cin + ci) * 16;
            const int a0 = a == 2 ? 1 : 0, a1 = a == 0 ? 0 : 1, b0 = b == 2 ? 1 : 0, b1 = b == 0 ? 0 : 1;
            for (int aa = a0; aa <= a1; ++aa)
                for (int bb = b0; bb <= b1; ++bb) u += g[(2 * aa + p) * 4 + 2 * bb + q];
        }
        out[e] = u;
    }
    *(f32x4*)(U + gidx * 4) = out;
}

// data-gradient weights: for output phase (py,px) the 2x2 filter g[a][b] = w[co][ci][3-2a-py][3-2b-px], M = ci, K = co
// (16 co per chunk, k = co % 16).  Same lane order as above, the four phases stored one after the other:
// U2[phase][mb(ci)][wave][chunk(co)][i][lane][e]; one thread per 16-byte group.
__global__ __launch_bounds__(256) void wino22_weight_dg_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout,
                                                               int Cin, long long ngroups) {
    const long long gidx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gidx >= ngroups) return;
    const int nchunk = Cout / KC, mbs = (Cin + BM - 1) / BM;
    const int lane = (int)(gidx & 63); long long r = gidx >> 6;
    const int i = (int)(r % 6); r /= 6;
    const int chunk = (int)(r % nchunk); r /= nchunk;
    const int wv = (int)(r % NWAVE); r /= NWAVE;
    const int mb = (int)(r % mbs); const int phase = (int)(r / mbs), py = phase >> 1, px = phase & 1;
    const int h = lane >> 5, l31 = lane & 31;
    f32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int v = 4 * i + e, uu = v >> 3, kk = v & 7;
        const int unit = 3 * wv + uu, xi = unit >> 2, mt = unit & 3, a = xi / 3, b = xi - a * 3;
        const int ci = mb * BM + mt * 32 + l31;
        const int co = chunk * KC + 2 * kk + h;
        float u = 0.f;
        if (ci < Cin) {
            const float* g = w + ((size_t)co * Cin + ci) * 16;
            const int a0 = a == 2 ? 1 : 0, a1 = a == 0 ? 0 : 1, b0 = b == 2 ? 1 : 0, b1 = b == 0 ? 0 : 1;
            for (int aa = a0; aa <= a1; ++aa)
                for (int bb = b0; bb <= b1; ++bb) u += g[(3 - 2 * aa - py) * 4 + 3 - 2 * bb - px];
        }
        out[e] = u;
    }
    *(f32x4*)(U + gidx * 4) = out;
}

struct W22P {
    const float* X; const float* U; float* Y; float* ws;
    int Cin, H, W, Cout, OH, OW, B;
    int ntile, ntb, mbs, nsplit, cps, nchunk, nitem;
    long long slab;
    unsigned x_bytes, u_bytes;
};

// DG = false: forward.  X = conv input (B,Cin,H,W), K channels = (ci, phase), window 6x6 at (4ty-1, 4tx-1).
// DG = true : data gradient, one output phase (py,px) per work item: dX[2y+py][2x+px] is a plain 2x2 stride-1 convolution of
//             dY with the taps w[3-2a-py][3-2b-px].  X = dY (B,Cout,OH,OW), K channels = co (16 per chunk), window 3x3 at
//             (2ty-1+py, 2tx-1+px) on the phase grid (OH x OW positions), M = ci; p.Cin / p.H / p.W are the K-channel count
//             and the dims of X, p.Cout / p.OH / p.OW the M count and the dims of the phase grid.
template <bool DG>
__global__ __launch_bounds__(NTHR) void wino22_kernel(const W22P p) {
    __shared__ __attribute__((aligned(16))) float raw[TSZ];             // Ts; the staging images alias its head
    float* const Xs = raw;                                              // [2][CI][6][6][tile]
    float* const Vs = raw + 2 * XSZ;                                    // [2][xi][k][tile]
    float* const Ts = raw;
    static_assert(2 * XSZ + 2 * VSZ <= TSZ, "staging fits under Ts");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int plane = p.H * p.W, oplane = p.OH * p.OW;
    const int tx_n = p.OW >> 1, ty_n = p.OH >> 1, tpi = tx_n * ty_n;    // tiles per row / column / image
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, (short)0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)p.U, (short)0, (int)p.u_bytes, 0x00020000);

    // staging role: element e = tid + NTHR*i (i < 6) of Xs = (rest = e >> 5 -> ci, r6, c6; tile = e & 31 = tid & 31)
    constexpr int NXE = XSZ / NTHR;                                     // 6
    static_assert(NXE * NTHR == XSZ, "staging divides");
    // transform role (tid < 512): k-channel kc = tid >> 5 = ci*4 + 2p + q, tile = tid & 31
    const int kc = (tid >> 5) & 15, kci = kc >> 2, kp = (kc >> 1) & 1, kq = kc & 1;
    const bool xform = tid < 16 * NT;
    // MFMA role: units 3*wave + uu
    int uxi[3], umt[3];
#pragma unroll
    for (int uu = 0; uu < 3; ++uu) { const int u = 3 * wave + uu; uxi[uu] = u >> 2; umt[uu] = u & 3; }

    // per work item: (mb, tile block tb, split sp)
    unsigned xg[NXE]; unsigned ubase; int c_beg = 0, c_end = 0, m0 = 0, tb = 0, sp = 0, ph = 0;
    auto plan = [&](int item) {
        sp = item % p.nsplit; const int t2 = item / p.nsplit;
        tb = t2 % p.ntb; const int t3 = t2 / p.ntb;
        const int mb = t3 % p.mbs; ph = t3 / p.mbs;          // (forward: one "phase" item class)
        m0 = mb * BM;
        c_beg = sp * p.cps; c_end = min(p.nchunk, c_beg + p.cps);
        const int tile = tb * NT + (tid & 31);
        const bool tok = tile < p.ntile;
        const int img = tile / tpi, rem = tile - img * tpi;
        const int ty = rem / tx_n, tx = rem - ty * tx_n;
#pragma unroll
        for (int i = 0; i < NXE; ++i) {
            const int rest = (tid >> 5) + (NTHR / 32) * i;              // (channel, row, column) of this thread's i-th element
            int ci, iy, ix;
            if constexpr (DG) {
                ci = rest / 9; const int r = rest - ci * 9, r3 = r / 3, c3 = r - r3 * 3;
                iy = 2 * ty - 1 + (ph >> 1) + r3; ix = 2 * tx - 1 + (ph & 1) + c3;
            } else {
                ci = rest / 36; const int r = rest - ci * 36, r6 = r / 6, c6 = r - r6 * 6;
                iy = 4 * ty - 1 + r6; ix = 4 * tx - 1 + c6;
            }
            const bool ok = tok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            xg[i] = ok ? (unsigned)(img * p.Cin + ci) * plane + (unsigned)(iy * p.W + ix) : 0x30000000u;
        }
        ubase = (unsigned)(((ph * p.mbs + mb) * NWAVE + wave) * p.nchunk) * (6 * 64 * 4) + (unsigned)lane * 4;
    };

    float rx[NXE], rx1[NXE];
    auto load_x = [&](float (&r)[NXE], int chunk) {
#pragma unroll
        for (int i = 0; i < NXE; ++i) r[i] = ldg1(rX, xg[i] == 0x30000000u ? 0x30000000u : xg[i] + (unsigned)(chunk * (DG ? KC : CI)) * plane);
    };
    auto store_x = [&](const float (&r)[NXE], float* Xd) {
#pragma unroll
        for (int i = 0; i < NXE; ++i) Xd[tid + NTHR * i] = r[i];
    };
    auto load_a = [&](f32x4 (&au)[6], int chunk) {
#pragma unroll
        for (int i = 0; i < 6; ++i) au[i] = ldg4(rU, ubase + (unsigned)(chunk * 6 + i) * 256u);
    };
    auto transform = [&](const float* Xc, float* Vd) {      // V = B^t d B of the phase (kp,kq) of channel kci, this tile
        if (xform) {
            float d[3][3];
            if constexpr (DG) {
                const float* px = Xc + (kc * 9) * NT + (tid & 31);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) d[i][j] = px[(i * 3 + j) * NT];
            } else {
                const float* px = Xc + (kci * 36 + kp * 6 + kq) * NT + (tid & 31);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) d[i][j] = px[((2 * i) * 6 + 2 * j) * NT];
            }
            float t[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) { t[0][j] = d[0][j] - d[1][j]; t[1][j] = d[1][j]; t[2][j] = d[2][j] - d[1][j]; }
            float* pv = Vd + kc * NT + (tid & 31);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                pv[(i * 3 + 0) * KC * NT] = t[i][0] - t[i][1];
                pv[(i * 3 + 1) * KC * NT] = t[i][1];
                pv[(i * 3 + 2) * KC * NT] = t[i][2] - t[i][1];
            }
        }
    };

    f32x16 acc[3];
    auto step = [&](int c, int cur, const f32x4 (&ac)[6], f32x4 (&an)[6]) {
        const int nxt = cur ^ 1;
        const float* Vc = Vs + cur * VSZ;
        const float* vb[3];
#pragma unroll
        for (int uu = 0; uu < 3; ++uu) vb[uu] = Vc + (uxi[uu] * KC + h) * NT + l31;
        load_a(an, c + 1);
        store_x(rx, Xs + cur * XSZ);                        // X(c+2)
        load_x(rx, c + 3);
        transform(Xs + nxt * XSZ, Vs + nxt * VSZ);          // X(c+1) -> V(c+1)
#if MOGAN_X6
        // the chunk's 8 k-steps of a unit = one 16-k group of the split-bf16 form (mogan_mma.h)
#pragma unroll
        for (int uu = 0; uu < 3; ++uu) {        // unit by unit: 24 fragment registers live at a time
            float a8[8], b8[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int v = uu * 8 + kk;
                a8[kk] = ac[v >> 2][v & 3]; b8[kk] = vb[uu][2 * kk * NT];
            }
            const X6Frag fa = x6_split8(a8), fb = x6_split8(b8);
#pragma unroll
            for (int term = 0; term < 6; ++term) acc[uu] = x6_mfma(fa, fb, term, acc[uu]);
        }
#else
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int uu = 0; uu < 3; ++uu) {
                const int v = uu * 8 + kk;
                acc[uu] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[v >> 2][v & 3], vb[uu][2 * kk * NT], acc[uu], 0, 0, 0);
            }
#endif
        __syncthreads();
    };

    f32x4 a0[6], a1[6];
    int item = blockIdx.x;
    if (item < p.nitem) {
        plan(item);
        load_x(rx, c_beg); load_x(rx1, c_beg + 1); load_a(a0, c_beg);
    }
    for (; item < p.nitem; item += gridDim.x) {
#pragma unroll
        for (int uu = 0; uu < 3; ++uu)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[uu][r] = 0.f;
        __syncthreads();                                    // the previous item's epilogue reads of Ts are done
        store_x(rx, Xs);
        store_x(rx1, Xs + XSZ);
        load_x(rx, c_beg + 2);
        __syncthreads();
        transform(Xs, Vs);
        __syncthreads();
        for (int c = c_beg; c < c_end; c += 2) {            // (c_end - c_beg) is even (host)
            step(c, 0, a0, a1);
            step(c + 1, 1, a1, a0);
        }
        const int cm0 = m0, ctb = tb, csp = sp, cph = ph;
        if (item + (int)gridDim.x < p.nitem) {
            plan(item + gridDim.x);
            load_x(rx, c_beg); load_x(rx1, c_beg + 1); load_a(a0, c_beg);
        }
        // ---- output transform: M (9 positions) -> Ts -> Y = A^t M A, A^t = [[1,1,0],[0,1,1]]
#pragma unroll
        for (int uu = 0; uu < 3; ++uu)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ts[(uxi[uu] * BM + umt[uu] * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * NT + l31] = acc[uu][r];
        __syncthreads();
        float* __restrict__ Yd = p.nsplit > 1 ? p.ws + (size_t)csp * p.slab : p.Y;
        for (int idx = tid; idx < BM * NT; idx += NTHR) {
            const int m = idx >> 5, tl = idx & 31;
            float M[9];
#pragma unroll
            for (int x = 0; x < 9; ++x) M[x] = Ts[x * BM * NT + idx];
            const int tile = ctb * NT + tl;
            if (cm0 + m < p.Cout && tile < p.ntile) {
                const int img = tile / tpi, rem = tile - img * tpi;
                const int ty = rem / tx_n, tx = rem - ty * tx_n;
                const float y00 = M[0] + M[1] + M[3] + M[4], y01 = M[1] + M[2] + M[4] + M[5];
                const float y10 = M[3] + M[4] + M[6] + M[7], y11 = M[4] + M[5] + M[7] + M[8];
                if constexpr (DG) {     // phase-grid position (2ty+a, 2tx+b) -> dX[2(2ty+a) + py][2(2tx+b) + px]
                    const int WX = 2 * p.OW;
                    float* o = Yd + ((size_t)(img * p.Cout + cm0 + m)) * (4 * (size_t)oplane) +
                               (size_t)(4 * ty + (cph >> 1)) * WX + 4 * tx + (cph & 1);
                    o[0] = y00; o[2] = y01; o[2 * WX] = y10; o[2 * WX + 2] = y11;
                } else {
                    float* o = Yd + ((size_t)(img * p.Cout + cm0 + m)) * oplane + (size_t)(2 * ty) * p.OW + 2 * tx;
                    *(float2*)o = make_float2(y00, y01);
                    *(float2*)(o + p.OW) = make_float2(y10, y11);
                }
            }
        }
    }
}

// y[i] = sum_s ws[s*slab + i] (fixed order)
__global__ __launch_bounds__(256) void wino22_reduce(const float* __restrict__ ws, float* __restrict__ y, long long n,
                                                     long long slab, int nsplit) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 s = *(const f32x4*)(ws + i);
    for (int k = 1; k < nsplit; ++k) s += *(const f32x4*)(ws + (size_t)k * slab + i);
    *(f32x4*)(y + i) = s;
}


// ------------------------------------------------------------------------------------------ weight gradient
// dg_pq = G^t [ sum over tiles (A dY A^t) .* (B^t d_pq B) ] G per input phase: nine GEMMs dU_xi[co][k] = sum_tiles Q_xi[co][tile]
// V_xi[k][tile], k = (ci, p, q), with the tiles on K.  A block (12 waves) owns 128 co x 64 k (16 ci x 4 phases) for all nine
// positions -- 72 (position, m-tile, n-tile) units, six per wave -- and walks a contiguous range of K-chunks of 8 tiles.  Per
// chunk: dY (128 co x 8 tiles x 2x2, from global straight into the threads that transform it) -> Qs[xi][tile][co]; the 6x6
// windows of 16 input channels -> Xs -> Vs[xi][tile][k]; double buffered, one barrier per chunk.  Partial dU per K-split go
// to the workspace; wino22_wgrad_finish sums them in order and applies G^t (.) G.
constexpr int WT = 8, WBM = 128, WBN = 64, WCI = WBN / 4, LDQ2 = WBM + 4, LDV2 = WBN + 4;

__global__ __launch_bounds__(NTHR) void wino22_wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                            float* __restrict__ part, int Cin, int H, int W, int Cout,
                                                            int nchunk, int cps, unsigned y_bytes, unsigned x_bytes) {
    constexpr int QSZ = 9 * WT * LDQ2, VSZ2 = 9 * WT * LDV2, XSZ2 = WCI * 36 * WT;
    __shared__ __attribute__((aligned(16))) float Qs[2 * QSZ];
    __shared__ __attribute__((aligned(16))) float Vs[2 * VSZ2];
    __shared__ __attribute__((aligned(16))) float Xs[2 * XSZ2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int n0c = blockIdx.x * WCI, m0 = blockIdx.y * WBM, sp = blockIdx.z;
    const int k_beg = sp * cps, k_end = min(nchunk, k_beg + cps);
    const int OH = H >> 1, OW = W >> 1, plane = H * W, oplane = OH * OW;
    const int tx_n = OW >> 1, tpi = tx_n * (OH >> 1);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, (short)0, (int)y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)x_bytes, 0x00020000);

    f32x16 acc[6];
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    // unit u of this wave: (xi, mt) pair = 3*wave + (u >> 1), n-tile = u & 1
    int uq[3], uv[3];                       // LDS row bases of the three (xi, mt) pairs: A at Qs[xi*WT rows] + mt*32, B at Vs[xi*WT rows]
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
        const int pair = 3 * wave + pr, xi = pair >> 2, mt = pair & 3;
        uq[pr] = xi * WT * LDQ2 + mt * 32 + l31 + h * LDQ2;
        uv[pr] = xi * WT * LDV2 + l31 + h * LDV2;
    }

    const int tl = tid & 7;                 // tile of the chunk this thread serves in every staging role
    constexpr int NXE = XSZ2 / NTHR;        // 6 window elements per thread
    static_assert(NXE * NTHR == XSZ2, "staging divides");
    float rx[NXE]; float2 ry[2][2];
    int t_img = 0, t_ty = 0, t_tx = 0; bool t_ok = false;
    auto tile_of = [&](int k) {
        const int tile = k * WT + tl;
        t_ok = k < k_end;
        t_img = tile / tpi; const int rem = tile - t_img * tpi;
        t_ty = rem / tx_n; t_tx = rem - t_ty * tx_n;
    };
    auto load_y = [&](int k) {              // dY 2x2 of (co = p >> 3, this tile) for p = tid, tid + 768
        tile_of(k);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int pp = tid + NTHR * r, co = pp >> 3;
            const bool ok = t_ok && pp < WBM * WT && m0 + co < Cout;
            const unsigned g = (unsigned)(t_img * Cout + m0 + co) * oplane + (unsigned)((2 * t_ty) * OW + 2 * t_tx);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
                ry[r][rr] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rY, ok ? (g + rr * OW) * 4u : 0xFFFFFFF8u, 0, 0));
        }
    };
    auto load_x = [&](int k) {              // element e = tid + 768 i: rest = e >> 3 -> (ci, r6, c6)
        tile_of(k);
#pragma unroll
        for (int i = 0; i < NXE; ++i) {
            const int rest = (tid >> 3) + (NTHR / 8) * i;
            const int ci = rest / 36, r = rest - ci * 36, r6 = r / 6, c6 = r - r6 * 6;
            const int iy = 4 * t_ty - 1 + r6, ix = 4 * t_tx - 1 + c6;
            const bool ok = t_ok && n0c + ci < Cin && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            rx[i] = ldg1(rX, ok ? (unsigned)(t_img * Cin + n0c + ci) * plane + (unsigned)(iy * W + ix) : 0x30000000u);
        }
    };
    auto store_x = [&](float* Xd) {
#pragma unroll
        for (int i = 0; i < NXE; ++i) Xd[tid + NTHR * i] = rx[i];
    };
    auto transform_q = [&](float* Qd) {     // Q = A d A^t, A = [[1,0],[1,1],[0,1]]
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int pp = tid + NTHR * r, co = pp >> 3;
            if (pp < WBM * WT) {
                const float d00 = ry[r][0].x, d01 = ry[r][0].y, d10 = ry[r][1].x, d11 = ry[r][1].y;
                const float R[3][2] = {{d00, d01}, {d00 + d10, d01 + d11}, {d10, d11}};
                float* q = &Qd[tl * LDQ2 + co];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    q[(i * 3 + 0) * WT * LDQ2] = R[i][0];
                    q[(i * 3 + 1) * WT * LDQ2] = R[i][0] + R[i][1];
                    q[(i * 3 + 2) * WT * LDQ2] = R[i][1];
                }
            }
        }
    };
    const int vk = tid >> 3, vci = vk >> 2, vp = (vk >> 1) & 1, vq = vk & 1;     // V role (tid < 512): k = ci*4 + 2p + q
    auto transform_v = [&](const float* Xc, float* Vd) {
        if (tid < WBN * WT) {
            const float* px = Xc + (vci * 36 + vp * 6 + vq) * WT + tl;
            float d[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) d[i][j] = px[((2 * i) * 6 + 2 * j) * WT];
            float t[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) { t[0][j] = d[0][j] - d[1][j]; t[1][j] = d[1][j]; t[2][j] = d[2][j] - d[1][j]; }
            float* pv = Vd + tl * LDV2 + vk;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                pv[(i * 3 + 0) * WT * LDV2] = t[i][0] - t[i][1];
                pv[(i * 3 + 1) * WT * LDV2] = t[i][1];
                pv[(i * 3 + 2) * WT * LDV2] = t[i][2] - t[i][1];
            }
        }
    };

    if (k_beg < k_end) {
        load_y(k_beg); load_x(k_beg);
        store_x(Xs);
        load_x(k_beg + 1);
        __syncthreads();
        transform_q(Qs); transform_v(Xs, Vs);
        load_y(k_beg + 1);
        store_x(Xs + XSZ2);
        load_x(k_beg + 2);
        __syncthreads();
#if MOGAN_X6 && defined(MOGAN_X6_W22_WGRAD)
        // (off: with 12 waves = 170 registers per lane the pair form spills -- 6 accumulators + the kept operands of the even
        // chunk + the fragments -- and ran 3.5x slower than the native form below; measured round 2)
        // split-bf16 form: a chunk supplies 4 k-values per lane, the MFMAs of a chunk pair are issued in its second
        // iteration; chunks past k_end are staged as zeros, so an unpaired last chunk is paired with zeros
        float ae[3][WT / 2], b0e[3][WT / 2], b1e[3][WT / 2];
        for (int k = k_beg; k < k_end; k += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int cur = par, nxt = cur ^ 1, kc = k + par;
                const float* Qc = Qs + cur * QSZ;
                const float* Vc = Vs + cur * VSZ2;
                auto do_pr = [&](int pr) {
                    float a8[8], b08[8], b18[8];
#pragma unroll
                    for (int kk = 0; kk < WT / 2; ++kk) {
                        const float a = Qc[uq[pr] + 2 * kk * LDQ2];
                        const float b0 = Vc[uv[pr] + 2 * kk * LDV2], b1 = Vc[uv[pr] + 2 * kk * LDV2 + 32];
                        if (par == 0) { ae[pr][kk] = a; b0e[pr][kk] = b0; b1e[pr][kk] = b1; }
                        else {
                            a8[kk] = ae[pr][kk]; b08[kk] = b0e[pr][kk]; b18[kk] = b1e[pr][kk];
                            a8[4 + kk] = a; b08[4 + kk] = b0; b18[4 + kk] = b1;
                        }
                    }
                    if (par == 1) {
                        const X6Frag fa = x6_split8(a8), fb0 = x6_split8(b08), fb1 = x6_split8(b18);
#pragma unroll
                        for (int term = 0; term < 6; ++term) {
                            acc[pr * 2] = x6_mfma(fa, fb0, term, acc[pr * 2]);
                            acc[pr * 2 + 1] = x6_mfma(fa, fb1, term, acc[pr * 2 + 1]);
                        }
                    }
                };
                do_pr(0);
                transform_q(Qs + nxt * QSZ);                        // dY(kc+1)
                load_y(kc + 2);
                do_pr(1);
                store_x(Xs + cur * XSZ2);                           // X(kc+2)
                load_x(kc + 3);
                do_pr(2);
                transform_v(Xs + nxt * XSZ2, Vs + nxt * VSZ2);      // X(kc+1) -> V(kc+1)
                __syncthreads();
            }
        }
#else
        for (int k = k_beg; k < k_end; ++k) {
            const int cur = (k - k_beg) & 1, nxt = cur ^ 1;
            const float* Qc = Qs + cur * QSZ;
            const float* Vc = Vs + cur * VSZ2;
            transform_q(Qs + nxt * QSZ);                        // dY(k+1)
            load_y(k + 2);
            store_x(Xs + cur * XSZ2);                           // X(k+2)
            load_x(k + 3);
            transform_v(Xs + nxt * XSZ2, Vs + nxt * VSZ2);      // X(k+1) -> V(k+1)
#pragma unroll
            for (int kk = 0; kk < WT / 2; ++kk)
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) {
                    const float a = Qc[uq[pr] + 2 * kk * LDQ2];
                    const float b0 = Vc[uv[pr] + 2 * kk * LDV2], b1 = Vc[uv[pr] + 2 * kk * LDV2 + 32];
                    acc[pr * 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[pr * 2], 0, 0, 0);
                    acc[pr * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[pr * 2 + 1], 0, 0, 0);
                }
            __syncthreads();
        }
#endif
    }
    // partial dU[sp][xi][co][4*Cin]
    const int N4 = 4 * Cin;
    float* out = part + (size_t)sp * 9 * Cout * N4;
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
        const int pair = 3 * wave + pr, xi = pair >> 2, mt = pair & 3;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n = 4 * n0c + nt * 32 + l31;
                if (m < Cout && n < N4) out[((size_t)xi * Cout + m) * N4 + n] = acc[pr * 2 + nt][r];
            }
    }
}

// dW[co][ci][2a+p][2b+q] (+)= (G^t dU_pq G)[a][b],  G^t = [[1,1,0],[0,1,1]]
__global__ __launch_bounds__(64) void wino22_wgrad_finish(const float* __restrict__ part, float* __restrict__ dw, int Cout,
                                                          int Cin, int nsplit, int accumulate) {
    const long long i = (long long)blockIdx.x * 64 + threadIdx.x;          // (co, ci, p, q): i = (co*Cin + ci)*4 + 2p + q
    const long long n = (long long)Cout * Cin * 4;
    if (i >= n) return;
    float u[9];
#pragma unroll
    for (int x = 0; x < 9; ++x) u[x] = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float* ps = part + (size_t)s * 9 * n + i;
#pragma unroll
        for (int x = 0; x < 9; ++x) u[x] += ps[(size_t)x * n];
    }
    const int pq = (int)(i & 3), p = pq >> 1, q = pq & 1;
    float* d = dw + (i >> 2) * 16;
    const float g00 = u[0] + u[1] + u[3] + u[4], g01 = u[1] + u[2] + u[4] + u[5];
    const float g10 = u[3] + u[4] + u[6] + u[7], g11 = u[4] + u[5] + u[7] + u[8];
    float* d00 = d + p * 4 + q; float* d01 = d + p * 4 + 2 + q; float* d10 = d + (2 + p) * 4 + q; float* d11 = d + (2 + p) * 4 + 2 + q;
    *d00 = (accumulate ? *d00 : 0.f) + g00; *d01 = (accumulate ? *d01 : 0.f) + g01;
    *d10 = (accumulate ? *d10 : 0.f) + g10; *d11 = (accumulate ? *d11 : 0.f) + g11;
}

}  // namespace

// The F(2x2,2x2) kernels are OFF by default since the split-bf16 MFMA form (mogan_mma.h): with the matrix pipe 2.7x cheaper the
// direct / implicit-GEMM kernels take the 4x4 s2 layers faster than the 16/9 saving of the transform pays for (same-box A/B
// of the B = 16 step: 336.5 vs 331.3 img/s with them off).  MOGAN_WINO22=1 turns them on; the test hook
// mogan_wino22_debug_min_tiles(n >= 0) also does (and lowers their size threshold), -1 restores the default.
static int g_w22_min_tiles = -1;
extern "C" int mogan_wino22_debug_min_tiles(int n) { g_w22_min_tiles = n; return 0; }
static int w22_min_tiles() {
    static const int env_min = getenv("MOGAN_WINO22_MIN_TILES") ? atoi(getenv("MOGAN_WINO22_MIN_TILES")) : 1024;
    return g_w22_min_tiles >= 0 ? g_w22_min_tiles : env_min;
}
static bool w22_on() {
    static const int env_on = getenv("MOGAN_WINO22") ? atoi(getenv("MOGAN_WINO22")) : 0;
    return env_on != 0 || g_w22_min_tiles >= 0;
}

// ---- internal entry point (hidden visibility): 1 = handled, 0 = not eligible, < 0 = error ----------------------------
// y (B,Cout,H/2,W/2) = conv4x4 s2 p1 (x (B,Cin,H,W), w (Cout,Cin,4,4)).  Workspace: the transformed weights
// (ceil(Cout/128)*128 * 36 * Cin floats) followed by nsplit output slabs when the K range is split.
int mogan_wino22_fwd_try(const float* x, const float* w, float* y, int B, int Cin, int H, int W, int Cout, int KH, int KW,
                         int stride, int ph, int pw, int up, void* ws, size_t ws_bytes, hipStream_t st) {
    const bool on = w22_on();
    if (!on || !(KH == 4 && KW == 4 && stride == 2 && ph == 1 && pw == 1 && up == 0)) return 0;
    if ((Cin % (2 * CI)) || Cin < 64 || Cout < 96 || (H % 4) || (W % 4)) return 0;
    if ((((uintptr_t)y) & 15) != 0) return 0;
    const int OH = H / 2, OW = W / 2;
    const long long mbs = (Cout + BM - 1) / BM;
    const size_t ubytes = (size_t)mbs * BM * 36 * Cin * sizeof(float);
    const long long ynum = (long long)B * Cout * OH * OW;
    if ((long long)B * Cin * H * W >= (1ll << 29) || (long long)Cin * H * W >= (1ll << 26) || ynum >= (1ll << 30) ||
        ubytes >= (1ull << 31))
        return 0;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    W22P p{};
    p.ntile = B * (OH / 2) * (OW / 2);
    // The transformed weights are 36/16 of the raw ones and are rebuilt per call (write + read of 36 x Cout x Cin floats): that
    // only pays when every weight meets many tiles.  Measured at B = 16 (TFLOP/s direct-equivalent, this kernel vs the
    // implicit GEMM): 4096 tiles 156 vs 90, 1024 tiles 135 vs 99 / 105 vs 88, 256 tiles 73 vs 98 / 28 vs 88.
    if (p.ntile < w22_min_tiles()) return 0;
    p.ntb = (p.ntile + NT - 1) / NT; p.mbs = (int)mbs; p.nchunk = Cin / CI;
    const long long blocks = (long long)p.ntb * mbs;
    // K-split: the persistent grid runs ceil(items / ncu) rounds; pick the split count whose last round is fullest
    // (a partial round costs a whole one), a little in favour of fewer splits (partial slabs to write and reduce)
    int nsplit = 1; double best = 1e30;
    for (int s = 1; s <= 16 && s <= std::max(1, p.nchunk / 8); ++s) {
        const double items = (double)blocks * s, rounds = (double)((blocks * s + ncu - 1) / ncu);
        const double cost = rounds * ncu / items * (1.0 + 0.03 * (s - 1));
        if (cost < best - 1e-9) { best = cost; nsplit = s; }
    }
    const size_t ubytes_al = (ubytes + 255) & ~(size_t)255;
    if (!ws || ws_bytes < ubytes_al) return 0;
    if (nsplit > 1) {
        const size_t fit = (ws_bytes - ubytes_al) / ((size_t)ynum * sizeof(float));
        if (fit < 2) nsplit = 1; else nsplit = (int)std::min<size_t>(nsplit, fit);
    }
    int cps = (p.nchunk + nsplit - 1) / nsplit; cps += cps & 1;         // even chunk count per split
    nsplit = (p.nchunk + cps - 1) / cps;
    p.nsplit = nsplit; p.cps = cps; p.nitem = (int)(blocks * nsplit); p.slab = ynum;
    p.X = x; p.U = (const float*)ws; p.Y = y; p.ws = (float*)((char*)ws + ubytes_al);
    p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.OH = OH; p.OW = OW; p.B = B;
    p.x_bytes = (unsigned)(4ull * B * Cin * H * W); p.u_bytes = (unsigned)ubytes;
    const long long n = (long long)Cout * Cin;
    const long long ngroups = mbs * NWAVE * p.nchunk * 6 * 64;
    hipLaunchKernelGGL(wino22_weight_kernel, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, st, w, (float*)ws, Cout, Cin,
                       ngroups);
    hipLaunchKernelGGL(wino22_kernel<false>, dim3((unsigned)std::min<long long>(p.nitem, ncu)), dim3(NTHR), 0, st, p);
    if (nsplit > 1)
        hipLaunchKernelGGL(wino22_reduce, dim3((unsigned)((ynum / 4 + 255) / 256)), dim3(256), 0, st, (const float*)p.ws, y,
                           ynum, ynum, nsplit);
    return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
}

// dx (B,Cin,H,W) = data gradient of conv4x4 s2 p1 for dy (B,Cout,H/2,W/2): four output phases, one work-item class each
int mogan_wino22_dgrad_try(const float* dy, const float* w, float* dx, int B, int Cin, int H, int W, int Cout, int KH, int KW,
                           int stride, int ph, int pw, int up, void* ws, size_t ws_bytes, hipStream_t st) {
    const bool on = w22_on();
    static const int dg_on = getenv("MOGAN_WINO22_DGRAD") ? atoi(getenv("MOGAN_WINO22_DGRAD")) : 1;
    if (!on || !dg_on || !(KH == 4 && KW == 4 && stride == 2 && ph == 1 && pw == 1 && up == 0)) return 0;
    if ((Cout % (2 * KC)) || Cout < 64 || Cin < 96 || (H % 4) || (W % 4)) return 0;
    const int OH = H / 2, OW = W / 2;
    // measured (B = 16, TFLOP/s direct-equivalent): 144 vs 99 and 107 vs 99 at 16x16 phase grids (there the alternative is the
    // implicit GEMM), but 123 vs 121 / 107 vs 111 at 32x32 / 64x64, where the direct 2x2-phase kernel runs: only take the former
    static const int max_ow = getenv("MOGAN_WINO22_DGRAD_MAXOW") ? atoi(getenv("MOGAN_WINO22_DGRAD_MAXOW")) : 16;
    if (g_w22_min_tiles != 1 && OW > max_ow) return 0;          // (the kernel tests lower min_tiles to 1 and take everything)
    const long long mbs = (Cin + BM - 1) / BM;
    const size_t ubytes = (size_t)4 * mbs * BM * 9 * Cout * sizeof(float);
    const long long xnum = (long long)B * Cin * H * W;
    if ((long long)B * Cout * OH * OW >= (1ll << 29) || (long long)Cout * OH * OW >= (1ll << 26) || xnum >= (1ll << 30) ||
        ubytes >= (1ull << 31))
        return 0;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    W22P p{};
    p.ntile = B * (OH / 2) * (OW / 2);
    if (p.ntile < w22_min_tiles()) return 0;
    p.ntb = (p.ntile + NT - 1) / NT; p.mbs = (int)mbs; p.nchunk = Cout / KC;
    const long long blocks = (long long)p.ntb * mbs * 4;
    int nsplit = 1; double best = 1e30;
    for (int s = 1; s <= 16 && s <= std::max(1, p.nchunk / 8); ++s) {
        const double items = (double)blocks * s, rounds = (double)((blocks * s + ncu - 1) / ncu);
        const double cost = rounds * ncu / items * (1.0 + 0.03 * (s - 1));
        if (cost < best - 1e-9) { best = cost; nsplit = s; }
    }
    const size_t ubytes_al = (ubytes + 255) & ~(size_t)255;
    if (!ws || ws_bytes < ubytes_al) return 0;
    if (nsplit > 1) {
        const size_t fit = (ws_bytes - ubytes_al) / ((size_t)xnum * sizeof(float));
        if (fit < 2) nsplit = 1; else nsplit = (int)std::min<size_t>(nsplit, fit);
    }
    int cps = (p.nchunk + nsplit - 1) / nsplit; cps += cps & 1;
    nsplit = (p.nchunk + cps - 1) / cps;
    p.nsplit = nsplit; p.cps = cps; p.nitem = (int)(blocks * nsplit); p.slab = xnum;
    p.X = dy; p.U = (const float*)ws; p.Y = dx; p.ws = (float*)((char*)ws + ubytes_al);
    p.Cin = Cout; p.H = OH; p.W = OW; p.Cout = Cin; p.OH = OH; p.OW = OW; p.B = B;    // K channels / dims of dY, M = ci, phase grid
    p.x_bytes = (unsigned)(4ull * B * Cout * OH * OW); p.u_bytes = (unsigned)ubytes;
    const long long n = (long long)Cout * Cin;
    const long long ngroups = 4 * mbs * NWAVE * p.nchunk * 6 * 64;
    hipLaunchKernelGGL(wino22_weight_dg_kernel, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, st, w, (float*)ws, Cout,
                       Cin, ngroups);
    hipLaunchKernelGGL(wino22_kernel<true>, dim3((unsigned)std::min<long long>(p.nitem, ncu)), dim3(NTHR), 0, st, p);
    if (nsplit > 1)
        hipLaunchKernelGGL(wino22_reduce, dim3((unsigned)((xnum / 4 + 255) / 256)), dim3(256), 0, st, (const float*)p.ws, dx,
                           xnum, xnum, nsplit);
    return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
}

// dw (Cout,Cin,4,4) (+)= weight gradient of conv4x4 s2 p1; workspace: nsplit * 36 * Cout * Cin floats
int mogan_wino22_wgrad_try(const float* dy, const float* x, float* dw, int B, int Cin, int H, int W, int Cout, int KH, int KW,
                           int stride, int ph, int pw, int up, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    const bool on = w22_on();
    static const int wg_on = getenv("MOGAN_WINO22_WGRAD") ? atoi(getenv("MOGAN_WINO22_WGRAD")) : 1;
    if (!on || !wg_on || !(KH == 4 && KW == 4 && stride == 2 && ph == 1 && pw == 1 && up == 0)) return 0;
    if ((Cin % WCI) || Cin < 64 || Cout < 96 || (H % 4) || (W % 4)) return 0;
    if ((((uintptr_t)dy) & 7) != 0) return 0;
    const int OH = H / 2, OW = W / 2;
    const long long ntile = (long long)B * (OH / 2) * (OW / 2);
    if (ntile < w22_min_tiles() || (ntile % WT) != 0) return 0;
    if ((long long)B * Cin * H * W >= (1ll << 29) || (long long)B * Cout * OH * OW >= (1ll << 29)) return 0;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    const int nchunk = (int)(ntile / WT);
    const int tiles_mn = ((Cout + WBM - 1) / WBM) * (Cin / WCI);
    // K-split: fill the last round of 12-wave blocks (one per CU); every split costs a partial dU (36 x Cout x Cin floats)
    int nsplit = 1; double best = 1e30;
    for (int sx = 1; sx <= 64 && sx <= std::max(1, nchunk / 4); ++sx) {
        const double items = (double)tiles_mn * sx, rounds = (double)((tiles_mn * sx + ncu - 1) / ncu);
        const double cost = rounds * ncu / items * (1.0 + 0.04 * (sx - 1));
        if (cost < best - 1e-9) { best = cost; nsplit = sx; }
    }
    const size_t slab = (size_t)36 * Cout * Cin * sizeof(float);
    if (!ws || ws_bytes < slab) return 0;
    if ((size_t)nsplit * slab > ws_bytes) nsplit = (int)(ws_bytes / slab);
    const int cps = (nchunk + nsplit - 1) / nsplit;
    nsplit = (nchunk + cps - 1) / cps;
    dim3 grid((unsigned)(Cin / WCI), (unsigned)((Cout + WBM - 1) / WBM), (unsigned)nsplit);
    hipLaunchKernelGGL(wino22_wgrad_kernel, grid, dim3(NTHR), 0, st, dy, x, (float*)ws, Cin, H, W, Cout, nchunk, cps,
                       (unsigned)(4ull * B * Cout * OH * OW), (unsigned)(4ull * B * Cin * H * W));
    const long long n = (long long)Cout * Cin * 4;
    hipLaunchKernelGGL(wino22_wgrad_finish, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (const float*)ws, dw, Cout, Cin,
                       nsplit, accumulate);
    return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
}
