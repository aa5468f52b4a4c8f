// mogan_wino.hip -- Winograd F(2x2, 3x3) for the 3x3 stride-1 pad-1 convolutions of the generator's ResBlocks
// (model.py:67-81: conv3x3 C -> 2C and C -> C at 64x64 and 128x128, 21 % of the step's multiply-adds), fused in ONE kernel:
//   Y = A^t [ sum_ci (G g G^t) .* (B^t d B) ] A         g = 3x3 filter, d = 4x4 input tile, Y = 2x2 outputs
// 16 multiplies per (ci, co, 2x2 outputs) instead of 36: 2.25x fewer MFMA flops, no intermediate in HBM.
//
// Forward / data gradient (wino3_fwd_kernel, split-bf16 build): a block owns 96 output channels x 32 tiles (2 tile rows x 16 tile
// columns = 4 x 32 output pixels) of one image, 8 waves, wave w the two transform positions 2w, 2w + 1; the transformed filters
// arrive pre-split in lane order (wino3_weight_kernel), the halo is staged and transformed in LDS (see the kernel's comment).
// Weight gradient (wino_wgrad_kernel + wino_wgrad_finish): tiles on K, both operands transformed in LDS, K-split partials.
// One forward form since round 5: the round-2 kernel (fp32 filter planes split in registers), the prepared-planes entry points
// and the lab switches lost their A/Bs and were removed; the native-fp32 build takes the direct kernels for these layers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include "../../include/mogan_hip.h"
#include "mogan_internal.h"
#include "mogan_mma.h"

#ifndef MOGAN_WINO5
#define MOGAN_WINO5 1
#endif
#ifndef WGRAD_XCD
#define WGRAD_XCD 1
#endif
#ifndef W5_XCD
#define W5_XCD 1
#endif
#ifndef W5_NT
#define W5_NT 1
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 8, TROWS = 2, TCOLS = 16, NT = TROWS * TCOLS, BM = 96;
constexpr int XR = 2 * TROWS + 2, XC = 2 * TCOLS + 2, XCP = 36;

__device__ __forceinline__ float ldgx(__amdgpu_buffer_rsrc_t r, unsigned idx, bool ok) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, ok ? idx * 4u : 0xFFFFFFFCu, 0, 0));
}
__device__ __forceinline__ f32x4 ldgx4(__amdgpu_buffer_rsrc_t r, unsigned idx, bool ok) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, ok ? idx * 4u : 0xFFFFFFF0u, 0, 0));
}

#if MOGAN_X6
// ------------------------------------------------------------------------------------------ round 4: forward, third form
// Same tiling (96 output channels x 32 tiles per block, wave w owns positions 2w, 2w+1, 2 x 3 accumulator tiles), new K loop:
//  * a K-step is 16 input channels = the k-extent of ONE v_mfma_f32_32x32x16_bf16 group (lane (h, l31) holds channels
//    8h .. 8h+7), so every step issues its 36 MFMAs -- the earlier form alternated a staging-only step with an MFMA step;
//  * the transformed filters arrive PRE-SPLIT into their three bf16 pieces in lane order (wino3_weight_kernel), 18 16-byte
//    loads per lane and step straight into the MFMA operand registers: no split arithmetic on the A side (it was 3/4 of
//    the kernel's split VALU work), and the weights are split once per (co, ci, position), not once per tile;
//  * one barrier per step; the MFMAs of step p run beside: halo X(p+2) registers -> LDS, halo loads of X(p+3), the input
//    transform X(p+1) -> V(p+1), and the filter loads of step p+1 (position j's registers are reloaded as soon as its MFMAs
//    have been issued).
// U3[mb][wave][step][j][a][piece][lane] x 16 bytes.
constexpr int CK2 = 16, XSZ2 = CK2 * XR * XCP, VSZ2 = 16 * CK2 * NT;

// Filter transform, one wave per unit = (32 output channels) x (16 input channels = one K step): the 32 x 16 x 9 filter values
// come in as contiguous rows (576 / 1152 bytes each; the round-4 form had every lane walk its own filter row -- 32 cache lines per
// load instruction, 10 us per layer for 2 MB of output), go through LDS, and lane (h, l31) -- output channel l31, input channels
// 8 h .. 8 h + 7 -- transforms its 8 filters for ALL 16 positions from registers (G g G^t once per filter) and stores the 48
// fragments of the unit, 1 KB per (position, piece) across the wave.
constexpr int WU_ROW = 16 * 9 + 1;      // LDS floats per output channel (odd: conflict-free lane-per-row reads)

__device__ __forceinline__ void wino3_weight_unit(const float* __restrict__ w, uint4* __restrict__ U3, int Cout, int Cin, int flip,
                                                  int unit, float* __restrict__ T) {
    const int Kin = flip ? Cout : Cin, Kout = flip ? Cin : Cout;
    const int nstep = Kin / CK2;
    const int step = unit % nstep, mt = unit / nstep;            // mt: 32-channel block = (mb, a3)
    const int lane = threadIdx.x & 63, h = lane >> 5, l31 = lane & 31;
    const int k0 = mt * 32, c0 = step * CK2;
    // stage T[kout 32][kin 16][9]: thread i copies 72 consecutive floats of one contiguous source row
    if (!flip) {                                                 // kout = co, kin = ci: row = co, 144 floats from ci = c0
        const int row = lane >> 1, half = lane & 1;
        const bool ok = k0 + row < Kout;
        const float* src = w + ((size_t)(k0 + row) * Cin + c0) * 9 + half * 72;
        float* dst = T + row * WU_ROW + half * 72;
#pragma unroll
        for (int i = 0; i < 72; ++i) dst[i] = ok ? src[i] : 0.f;
    } else {                                                     // kout = ci, kin = co: row = co (16), 288 floats from ci = k0
        const int row = lane >> 2, qt = lane & 3;
        const float* src = w + ((size_t)(c0 + row) * Cin + k0 + qt * 8) * 9;
#pragma unroll
        for (int i = 0; i < 72; ++i) {
            const int ko = qt * 8 + i / 9, t = i - (i / 9) * 9;
            T[ko * WU_ROW + row * 9 + t] = k0 + ko < Kout ? src[i] : 0.f;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    float u[16][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float* gp = T + l31 * WU_ROW + (8 * h + e) * 9;
        float g[3][3];
#pragma unroll
        for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < 3; ++b_) g[a_][b_] = flip ? gp[(2 - a_) * 3 + (2 - b_)] : gp[a_ * 3 + b_];
        float t[4][3];                                           // G g: rows (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2)
#pragma unroll
        for (int b_ = 0; b_ < 3; ++b_) {
            t[0][b_] = g[0][b_];
            t[1][b_] = 0.5f * g[0][b_] + 0.5f * g[1][b_] + 0.5f * g[2][b_];
            t[2][b_] = 0.5f * g[0][b_] - 0.5f * g[1][b_] + 0.5f * g[2][b_];
            t[3][b_] = g[2][b_];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            u[r * 4 + 0][e] = t[r][0];
            u[r * 4 + 1][e] = 0.5f * t[r][0] + 0.5f * t[r][1] + 0.5f * t[r][2];
            u[r * 4 + 2][e] = 0.5f * t[r][0] - 0.5f * t[r][1] + 0.5f * t[r][2];
            u[r * 4 + 3][e] = t[r][2];
        }
    }
    const int mb = mt / 3, a3 = mt - mb * 3;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
        const X6Frag f = x6_split8(u[xi]);
        const size_t base = ((((size_t)(mb * 8 + (xi >> 1)) * nstep + step) * 2 + (xi & 1)) * 3 + a3) * 3 * 64 + lane;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) U3[base + (size_t)pc * 64] = __builtin_bit_cast(uint4, f.p[pc]);
    }
}

// units = ceil(Kout / 96) * 3 * (Kin / 16); four units (waves) per block
__global__ __launch_bounds__(256) void wino3_weight_kernel(const float* __restrict__ w, uint4* __restrict__ U3, int Cout,
                                                           int Cin, int flip, int nunit) {
    __shared__ float T[4][32 * WU_ROW];
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit < nunit) wino3_weight_unit(w, U3, Cout, Cin, flip, unit, T[threadIdx.x >> 6]);
}

// The same transform for up to 32 (weight, direction) pairs in ONE launch: the owner of the weights (trainer.FlatAdam) rebuilds
// every prepared image of its bucket right behind the optimizer step (mogan_wino_prep_group), so that no convolution of the
// step transforms filters on its own chain.  Blocks of the members lie end to end on the grid.
constexpr int WPG_MAX = 32;
struct WinoPrepGroup { const float* w[WPG_MAX]; uint4* u3[WPG_MAX]; int Cout[WPG_MAX], Cin[WPG_MAX], flip[WPG_MAX]; unsigned end[WPG_MAX]; int n; };

__global__ __launch_bounds__(256) void wino3_weight_group_kernel(const WinoPrepGroup g) {
    __shared__ float T[4][32 * WU_ROW];
    int m = 0;
#pragma unroll
    for (int i = 0; i < WPG_MAX - 1; ++i) if (i + 1 < g.n && blockIdx.x >= g.end[i]) m = i + 1;
    const unsigned start = m ? g.end[m - 1] : 0u;
    const int Kin = g.flip[m] ? g.Cout[m] : g.Cin[m], Kout = g.flip[m] ? g.Cin[m] : g.Cout[m];
    const int nunit = ((Kout + BM - 1) / BM) * 3 * (Kin / CK2);
    const int unit = (int)(blockIdx.x - start) * 4 + (threadIdx.x >> 6);
    if (unit < nunit) wino3_weight_unit(g.w[m], g.u3[m], g.Cout[m], g.Cin[m], g.flip[m], unit, T[threadIdx.x >> 6]);
}

__global__ __launch_bounds__(512) void wino3_fwd_kernel(const float* __restrict__ X, const uint4* __restrict__ U3,
                                                        float* __restrict__ Y, int Cin, int H, int W, int Cout, int OH, int OW,
                                                        int pad, int tiles_x, int tiles_y, int ntile, int nimg,
                                                        const float* __restrict__ ep_scale, const float* __restrict__ ep_shift,
                                                        int ep_relu, unsigned x_bytes, unsigned u_bytes) {
    constexpr int VT = 8 * BM * NT > 2 * VSZ2 ? 8 * BM * NT : 2 * VSZ2;
    __shared__ __attribute__((aligned(16))) float Xs[2 * XSZ2];
    __shared__ __attribute__((aligned(16))) float VTs[VT];              // V of the K loop; the epilogue's exchange buffer
    float* const Vs = VTs;
    float* const Ts = VTs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int plane = H * W;
    const int nstep = Cin / CK2;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)U3, (short)0, (int)u_bytes, 0x00020000);

    // halo staging role: position (row, column) sr of the 6 x 34 halo, channels 2i + sg of a step (i < 8)
    const int sg = tid >> 8, sr = tid & 255;
    const bool sact = sr < XR * XC;
    const int shy = sr / XC, shx = sr - shy * XC;
    // (the 2 x 52 idle lanes of this role store their zeros into the two pad columns of the rows, which nobody reads:
    // unconditional stores keep the K loop one basic block)
    const int xl0 = sact ? sg * (XR * XCP) + shy * XCP + shx : sg * (XR * XCP) + (sr % XR) * XCP + XC + ((sr / XR) & 1);
    // transform role: rows 2th, 2th+1 of V = B^t d B for tile tt of channels tc and tc + 8
    const int th = tid >> 8, tc = (tid >> 5) & 7, tt = tid & 31, tty = tt >> 4, ttx = tt & 15;
    const int tx0 = tc * (XR * XCP) + (2 * tty + th) * XCP + 2 * ttx;
    const int tv0 = ((2 * th) * 4 * CK2 + tc) * NT + tt;
    const int bv0 = (wave * 2 * CK2 + 8 * h) * NT + l31;

    // filter loads: per-lane offset (lane x 16 bytes) in the VGPR, everything else -- (mb, wave, step, fragment) -- is uniform
    // and rides in the scalar offset of the buffer instruction (no vector address arithmetic in the K loop)
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int mbs = (Cout + BM - 1) / BM;
    const unsigned ulane = (unsigned)lane * 16u;
    unsigned xg, ubase; int m0, img, oy0, ox0;
    // work item = (spatial tile, output-channel block) with the channel block INNERMOST, a block walks a contiguous range of
    // items: the channel blocks of one spatial tile run back to back on one CU (the second one's halo comes out of L1 / L2, the
    // input leaves HBM once per layer), and vertically adjacent tiles -- which share two halo rows -- stay on the CU as well
    auto plan = [&](int tile, unsigned& g, unsigned& ub, int& m0_, int& img_, int& oy_, int& ox_) {
        int t = tile;
        const int mb = t % mbs; t /= mbs;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y; t /= tiles_y;
        img_ = t;
        m0_ = mb * BM; oy_ = ty * 2 * TROWS; ox_ = tx * 2 * TCOLS;
        const int iy = oy_ + shy - pad, ix = ox_ + shx - pad;
        const bool ok = sact && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        g = ok ? (unsigned)(img_ * Cin + sg) * plane + (unsigned)(iy * W + ix) : 0x30000000u;     // reads 0
        ub = (unsigned)((mb * 8 + wave_s) * nstep) * (18u * 1024u);
    };
    float rx[8], rx1[8];
    auto load_x = [&](float (&r)[8], unsigned g, int c0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = ldgx(rX, g + (unsigned)(c0 + 2 * i) * plane, true);
    };
    auto store_x = [&](const float (&r)[8], float* Xd) {
#pragma unroll
        for (int i = 0; i < 8; ++i) Xd[xl0 + 2 * i * (XR * XCP)] = r[i];
    };
    X6Frag fa[2][3];
    auto load_a = [&](int j, unsigned ub, int step) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                fa[j][a].p[pc] = __builtin_bit_cast(mma_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    rU, ulane, ub + (unsigned)(((step * 2 + j) * 3 + a) * 3 + pc) * 1024u, 0));
    };
    auto transform = [&](const float* Xc, float* Vd, int c8) {          // channel tc + c8
        float ra[4], rb[4], rc[4];
        const float* px = &Xc[tx0 + c8 * (XR * XCP)];
        {
            const float2 a0 = *(const float2*)px, a1 = *(const float2*)(px + 2);
            const float2 b0 = *(const float2*)(px + XCP), b1 = *(const float2*)(px + XCP + 2);
            const float2 c0 = *(const float2*)(px + 2 * XCP), c1 = *(const float2*)(px + 2 * XCP + 2);
            ra[0] = a0.x; ra[1] = a0.y; ra[2] = a1.x; ra[3] = a1.y;
            rb[0] = b0.x; rb[1] = b0.y; rb[2] = b1.x; rb[3] = b1.y;
            rc[0] = c0.x; rc[1] = c0.y; rc[2] = c1.x; rc[3] = c1.y;
        }
        float u[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t1 = ra[j] - rc[j], t2 = rb[j] - ra[j], t3 = rb[j] + rc[j];
            u[0][j] = th ? t2 : t1;
            u[1][j] = th ? t1 : t3;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* pv = &Vd[tv0 + (i * 4 * CK2 + c8) * NT];
            pv[0] = u[i][0] - u[i][2];
            pv[CK2 * NT] = u[i][1] + u[i][2];
            pv[2 * CK2 * NT] = u[i][2] - u[i][1];
            pv[3 * CK2 * NT] = u[i][1] - u[i][3];
        }
    };

    f32x16 acc[2][3];
    auto mma_j = [&](int j, const float* Vc) {
        float b8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[e] = Vc[bv0 + (j * CK2 + e) * NT];
        const X6Frag fb = x6_split8(b8);
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[j][a] = x6_mfma(fa[j][a], fb, term, acc[j][a]);
    };

    const int per = (ntile + (int)gridDim.x - 1) / (int)gridDim.x;
    int tile = blockIdx.x * per;
    const int tile_end = min(ntile, tile + per), tstep = 1;
    if (tile < tile_end) {
        plan(tile, xg, ubase, m0, img, oy0, ox0);
        load_x(rx, xg, 0); load_x(rx1, xg, CK2); load_a(0, ubase, 0); load_a(1, ubase, 0);
    }
    for (; tile < tile_end; tile += tstep) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][a][r] = 0.f;
        store_x(rx, Xs);
        store_x(rx1, Xs + XSZ2);
        load_x(rx, xg, 2 * CK2);
        __syncthreads();                                    // (also: every thread is done with the previous tile's Ts)
        transform(Xs, Vs, 0);
        transform(Xs, Vs, 8);
        __syncthreads();
        for (int p = 0; p < nstep; ++p) {
            const int cur = p & 1;
            const float* Vc = Vs + cur * VSZ2;
            float* Vn = Vs + (cur ^ 1) * VSZ2;
            const float* Xn = Xs + (cur ^ 1) * XSZ2;
            mma_j(0, Vc);
            load_a(0, ubase, p + 1);
            store_x(rx, Xs + cur * XSZ2);                   // X(p+2)
            load_x(rx, xg, (p + 3) * CK2);
            transform(Xn, Vn, 0);                           // X(p+1) -> V(p+1)
            mma_j(1, Vc);
            load_a(1, ubase, p + 1);
            transform(Xn, Vn, 8);
            __syncthreads();
        }
        const int cm0 = m0, cimg = img, coy0 = oy0, cox0 = ox0;
        const bool more = tile + tstep < tile_end;
        if (more) {
            plan(tile + tstep, xg, ubase, m0, img, oy0, ox0);
            load_x(rx, xg, 0); load_x(rx1, xg, CK2);
        }
        // ---- output transform (as in wino_fwd_kernel): row i = wave>>1 of M, T_i[b] = sum_j M[i][j] A[j][b] split over the
        // wave pair; Y[a][b] = sum_i A^t[a][i] T_i[b] through Ts, one pass per output column parity b
        const int wj = wave & 1;
        // item = (output channel m, pair of horizontally adjacent tiles): 96 x 16 = 1536 items, three per thread; the thread
        // ends up with a 2 x 4 output patch per item and stores it as two 16-byte rows
        float yreg[3][2][2][2];                             // [item][output row a][output column b][tile of the pair]
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (b) __syncthreads();
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = wj == 0 ? (b == 0 ? acc[0][a][r] + acc[1][a][r] : acc[1][a][r])
                                            : (b == 0 ? acc[0][a][r] : -acc[0][a][r] - acc[1][a][r]);
                    Ts[(wave * BM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * NT + l31] = p;
                }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int idx = 2 * (tid + 512 * q);        // = m * 32 + first tile of the pair
                float2 tq[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float2 u0 = *(const float2*)&Ts[(2 * i) * BM * NT + idx];
                    const float2 u1 = *(const float2*)&Ts[(2 * i + 1) * BM * NT + idx];
                    tq[i] = make_float2(u0.x + u1.x, u0.y + u1.y);
                }
                yreg[q][0][b][0] = tq[0].x + tq[1].x + tq[2].x; yreg[q][0][b][1] = tq[0].y + tq[1].y + tq[2].y;
                yreg[q][1][b][0] = tq[1].x - tq[2].x - tq[3].x; yreg[q][1][b][1] = tq[1].y - tq[2].y - tq[3].y;
            }
        }
        if (more) { load_a(0, ubase, 0); load_a(1, ubase, 0); }         // (the accumulators are free now)
        const bool fast = (OW & 3) == 0 && coy0 + 2 * TROWS <= OH && cox0 + 2 * TCOLS <= OW && cm0 + BM <= Cout &&
                          ep_scale == nullptr && (((uintptr_t)Y) & 15) == 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int it = tid + 512 * q;
            const int m = it >> 4, tp = it & 15;
            const int oy = coy0 + 2 * (tp >> 3), ox = cox0 + 4 * (tp & 7);
            if (fast) {
                float* o = Y + ((size_t)(cimg * Cout + cm0 + m) * OH + oy) * OW + ox;
                *(float4*)o = make_float4(yreg[q][0][0][0], yreg[q][0][1][0], yreg[q][0][0][1], yreg[q][0][1][1]);
                *(float4*)(o + OW) = make_float4(yreg[q][1][0][0], yreg[q][1][1][0], yreg[q][1][0][1], yreg[q][1][1][1]);
            } else if (cm0 + m < Cout) {
                float sc = 1.f, sh = 0.f;
                if (ep_scale != nullptr) { sc = ep_scale[cm0 + m]; sh = ep_shift[cm0 + m]; }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const int y = oy + a, x = ox + 2 * t + b;
                            if (y < OH && x < OW) {
                                float v = yreg[q][a][b][t];
                                if (ep_scale != nullptr) {
                                    v = fmaf(v, sc, sh);
                                    if (ep_relu) v = fmaxf(v, 0.f);
                                }
                                Y[((size_t)(cimg * Cout + cm0 + m) * OH + y) * OW + x] = v;
                            }
                        }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ round 6: forward, 16 waves per block
// wino3_fwd_kernel is bound by latency, not by a unit: counters of the layer alone (profiles/r06_wino_pmc.txt) show the matrix
// pipe 23 % busy, the texture-address unit ~50 %, LDS 17 %, vector ALU 13 % -- and the waves waiting 47 % of their cycles, two per
// SIMD.  Same tile (96 output channels x 32 tiles of one image), same LDS images and filter layout here, but 16 waves of ONE
// transform position each (48 accumulator registers instead of 96, one position's 9 filter fragments instead of two's), i.e. four
// waves per SIMD at <= 128 registers: twice the waves to hide every load and LDS round trip behind.  The output transform moves
// one 32-channel block of all 16 positions through LDS per pass (three passes, 64 KB each -- the V buffers' own size).
// Full tiles only (OW % 32 == 0, OH % 4 == 0, Kout % 96 == 0), no fused affine epilogue: everything else stays on wino3_fwd_kernel.
__global__ __launch_bounds__(1024) void wino5_fwd_kernel(const float* __restrict__ X, const uint4* __restrict__ U3,
                                                         float* __restrict__ Y, int Cin, int H, int W, int Cout, int OH, int OW,
                                                         int pad, int tiles_x, int tiles_y, int ntile, unsigned x_bytes,
                                                         unsigned u_bytes) {
    __shared__ __attribute__((aligned(16))) float Xs[2 * XSZ2];
    __shared__ __attribute__((aligned(16))) float VTs[2 * VSZ2];     // V of the K loop; the epilogue's exchange buffer
    float* const Vs = VTs;
    float* const Ts = VTs;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);            // this wave's transform position
    const int plane = H * W;
    const int nstep = Cin / CK2;
    const int mbs = Cout / BM;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)U3, (short)0, (int)u_bytes, 0x00020000);

    // halo staging role: position (row, column) sr of the 6 x 34 halo, channels 4 i + sg of a step (i < 4)
    const int sg = tid >> 8, sr = tid & 255;
    const bool sact = sr < XR * XC;
    const int shy = sr / XC, shx = sr - shy * XC;
    const int xl0 = sact ? sg * (XR * XCP) + shy * XCP + shx : sg * (XR * XCP) + (sr % XR) * XCP + XC + ((sr / XR) & 1);
    // transform role: rows 2 th, 2 th + 1 of V = B^t d B for tile tt of channel tc
    const int th = tid >> 9, tc = (tid >> 5) & 15, tt = tid & 31, tty = tt >> 4, ttx = tt & 15;
    const int tx0 = tc * (XR * XCP) + (2 * tty + th) * XCP + 2 * ttx;
    const int tv0 = ((2 * th) * 4 * CK2 + tc) * NT + tt;
    const int bv0 = (xi * CK2 + 8 * h) * NT + l31;
    const unsigned ulane = (unsigned)lane * 16u;
    unsigned xg, ubase; int m0, img, oy0, ox0;
    auto plan = [&](int tile) {
        int t = tile;
        const int mb = t % mbs; t /= mbs;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y; t /= tiles_y;
        img = t;
        m0 = mb * BM; oy0 = ty * 2 * TROWS; ox0 = tx * 2 * TCOLS;
        const int iy = oy0 + shy - pad, ix = ox0 + shx - pad;
        const bool ok = sact && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        xg = ok ? (unsigned)(img * Cin + sg) * plane + (unsigned)(iy * W + ix) : 0x30000000u;     // reads 0
        ubase = (unsigned)((mb * 8 + (xi >> 1)) * nstep) * (18u * 1024u) + (unsigned)(xi & 1) * (9u * 1024u);
    };
    float rx[4];
    auto load_x = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rx[i] = ldgx(rX, xg + (unsigned)(c0 + 4 * i) * plane, true);
    };
    auto store_x = [&](float* Xd) {
#pragma unroll
        for (int i = 0; i < 4; ++i) Xd[xl0 + 4 * i * (XR * XCP)] = rx[i];
    };
    X6Frag fa[3];
    auto load_a = [&](int step) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                fa[a].p[pc] = __builtin_bit_cast(mma_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    rU, ulane, ubase + (unsigned)(step * 18 + a * 3 + pc) * 1024u, 0));
    };
    auto transform = [&](const float* Xc, float* Vd) {
        float ra[4], rb[4], rc[4];
        const float* px = &Xc[tx0];
        {
            const float2 a0 = *(const float2*)px, a1 = *(const float2*)(px + 2);
            const float2 b0 = *(const float2*)(px + XCP), b1 = *(const float2*)(px + XCP + 2);
            const float2 c0 = *(const float2*)(px + 2 * XCP), c1 = *(const float2*)(px + 2 * XCP + 2);
            ra[0] = a0.x; ra[1] = a0.y; ra[2] = a1.x; ra[3] = a1.y;
            rb[0] = b0.x; rb[1] = b0.y; rb[2] = b1.x; rb[3] = b1.y;
            rc[0] = c0.x; rc[1] = c0.y; rc[2] = c1.x; rc[3] = c1.y;
        }
        float u[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t1 = ra[j] - rc[j], t2 = rb[j] - ra[j], t3 = rb[j] + rc[j];
            u[0][j] = th ? t2 : t1;
            u[1][j] = th ? t1 : t3;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* pv = &Vd[tv0 + (i * 4 * CK2) * NT];
            pv[0] = u[i][0] - u[i][2];
            pv[CK2 * NT] = u[i][1] + u[i][2];
            pv[2 * CK2 * NT] = u[i][2] - u[i][1];
            pv[3 * CK2 * NT] = u[i][1] - u[i][3];
        }
    };
    f32x16 acc[3];
    float b8[8];
    auto read_b = [&](const float* Vc) {
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[e] = Vc[bv0 + e * NT];
    };
    auto mma = [&]() {
        const X6Frag fb = x6_split8(b8);
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[a] = x6_mfma(fa[a], fb, term, acc[a]);
    };

    // Which items a block takes, in which order.  The L2 of an XCD (4 MB, 32 CUs) turns over every ~25 us under this kernel (78 KB of
    // halo + 49 KB of output per CU and item): a halo row fetched for one tile is gone before the vertically adjacent tile -- two items
    // later on the same CU -- asks for it, and every item pays its whole halo in 128-byte lines (measured: 3x the input bytes).  So the 32
    // blocks of an XCD (workgroups are dealt round-robin: XCD = blockIdx & 7) work on 32 ADJACENT tiles at the same time -- round r,
    // XCD k: spatial tiles (8 r + k) * 32 .. + 31 = 8 tile rows x 4 tile columns of one image -- so that a shared halo row is one L2 miss
    // for all its readers; the channel blocks of a tile stay innermost.  (Other grids: a contiguous range per block.)
    const int per = (ntile + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nsp = ntile / mbs;
    const bool super = W5_XCD && (gridDim.x & 7u) == 0 && nsp % (int)gridDim.x == 0;
    const int bq = blockIdx.x & 7, bj = blockIdx.x >> 3, pxcd = gridDim.x >> 3;
    auto item_tile = [&](int it) {                          // it-th item of this block -> combined index spatial * mbs + channel block
        if (!super) return (int)blockIdx.x * per + it;
        const int r = it / mbs, mb = it - r * mbs;
        return (((r * 8 + bq) * pxcd + bj) * mbs + mb);
    };
    const int nit = super ? per : max(0, min(ntile, ((int)blockIdx.x + 1) * per) - (int)blockIdx.x * per);
    int it = 0, tile = item_tile(0);
    if (it < nit) { plan(tile); load_x(0); }
    for (; it < nit; ++it) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        load_a(0);
        store_x(Xs);                                        // X(0)
        load_x(CK2);
        __syncthreads();                                    // (also: every thread is done with the previous tile's Ts)
        transform(Xs, Vs);
        store_x(Xs + XSZ2);                                 // X(1)
        load_x(2 * CK2);
        __syncthreads();
        for (int p = 0; p < nstep; ++p) {
            const int cur = p & 1;
            const float* Vc = Vs + cur * VSZ2;
            read_b(Vc);
            mma();
            load_a(p + 1);
            transform(Xs + (cur ^ 1) * XSZ2, Vs + (cur ^ 1) * VSZ2);   // X(p+1) -> V(p+1)
            store_x(Xs + cur * XSZ2);                       // X(p+2)   (X(p) was transformed one step ago)
            load_x((p + 3) * CK2);
            __syncthreads();
        }
        const int cm0 = m0, cimg = img, coy0 = oy0, cox0 = ox0;
        if (it + 1 < nit) { tile = item_tile(it + 1); plan(tile); load_x(0); }
        // ---- output transform, one 32-channel block (a) of all 16 positions per pass: Ts[position][channel 32][tile 32]
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (a) __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) Ts[(xi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * NT + l31] = acc[a][r];
            __syncthreads();
            if (tid < 512) {
                // item = (channel, pair of horizontally adjacent tiles): a 2 x 4 pixel patch
                const int co = tid >> 4, tp = tid & 15;
                float2 m[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) m[i][j] = *(const float2*)&Ts[((i * 4 + j) * 32 + co) * NT + 2 * tp];
                float2 t0[4], t1[4];                        // rows: A^t M
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t0[j] = make_float2(m[0][j].x + m[1][j].x + m[2][j].x, m[0][j].y + m[1][j].y + m[2][j].y);
                    t1[j] = make_float2(m[1][j].x - m[2][j].x - m[3][j].x, m[1][j].y - m[2][j].y - m[3][j].y);
                }
                const int oy = coy0 + 2 * (tp >> 3), ox = cox0 + 4 * (tp & 7);
                float* o = Y + ((size_t)(cimg * Cout + cm0 + a * 32 + co) * OH + oy) * OW + ox;
                const f32x4 y0 = {t0[0].x + t0[1].x + t0[2].x, t0[1].x - t0[2].x - t0[3].x, t0[0].y + t0[1].y + t0[2].y, t0[1].y - t0[2].y - t0[3].y};
                const f32x4 y1 = {t1[0].x + t1[1].x + t1[2].x, t1[1].x - t1[2].x - t1[3].x, t1[0].y + t1[1].y + t1[2].y, t1[1].y - t1[2].y - t1[3].y};
#if W5_NT       // the output is written once and not read again by this kernel: keep it from displacing the halos / filters in L2
                __builtin_nontemporal_store(y0, (f32x4*)o);
                __builtin_nontemporal_store(y1, (f32x4*)(o + OW));
#else
                *(f32x4*)o = y0;
                *(f32x4*)(o + OW) = y1;
#endif
            }
        }
    }
}

#endif  // MOGAN_X6

// ------------------------------------------------------------------------------------------ weight gradient
// dW = G^t [ sum over tiles (A dY A^t) .* (B^t d B) ] G: per position xi a GEMM dU_xi[co][ci] = sum_tiles Q_xi[co][tile]
// V_xi[ci][tile] with the tiles on K.  A block owns 96 co x 32 ci for all 16 positions (wave w: xi = 2w, 2w+1; 2 x 3
// accumulator tiles) and walks a contiguous range of K-chunks; a chunk = 8 tiles of one tile row (2 x 16 output pixels).
// Per chunk: dY (96 x 2 x 16, straight from global into the threads that transform it) -> Qs[xi][tile][co], the input halo
// (32 ci x 4 x 18) -> Xs -> Vs[xi][tile][ci]; both double buffered, one barrier per chunk, preparation of chunk c+1 in the
// shadow of the MFMAs of chunk c.  The partial dU of every block goes to the workspace; wino_wgrad_finish sums the
// K-splits in order and applies G^t (.) G.
constexpr int WCT = 8, WBN = 32, LDQ = BM + 4, WXR = 4, WXC = 2 * WCT + 2, LDX = 33;

__global__ __launch_bounds__(512) void wino_wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                         float* __restrict__ part, int Cin, int H, int W, int Cout,
                                                         int nchunk, int cps, unsigned y_bytes, unsigned x_bytes) {
    constexpr int QSZ = 16 * WCT * LDQ, VSZ = 16 * WCT * WBN, XSZ = WXR * WXC * LDX;
    __shared__ __attribute__((aligned(16))) float Qs[2 * QSZ];
    __shared__ __attribute__((aligned(16))) float Vs[2 * VSZ];
    __shared__ __attribute__((aligned(16))) float Xs[2 * XSZ + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    // (the blocks of one K range -- all (ci, co) blocks read the same dY / x chunks -- on one XCD: mma_xcd_block)
#if WGRAD_XCD
    unsigned bx_, by_, bz_; mma_xcd_block(bx_, by_, bz_);
#else
    const unsigned bx_ = blockIdx.x, by_ = blockIdx.y, bz_ = blockIdx.z;
#endif
    const int n0 = bx_ * WBN, m0 = by_ * BM, sp = bz_;
    const int k_beg = sp * cps, k_end = min(nchunk, k_beg + cps);
    const int plane = H * W;
    const int cgs = W / (2 * WCT), trs = H / 2;             // chunks per tile row, tile rows per image
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, (short)0, (int)y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)x_bytes, 0x00020000);

    f32x16 acc[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][a][r] = 0.f;

    // dY role: pair p = tid + 512 r (r = 0,1; 768 pairs): co = p >> 3, tile column = p & 7
    // X staging role: element e = tid + 512 i (i < 5; 2304 elements): ci = e / 72, row, col (lanes run along the columns)
    // V role: ci = tid & 31, tile column = (tid >> 5) & 7, th = tid >> 8 (rows 2th, 2th+1 of B^t d B)
    constexpr int NXE = (WBN * WXR * WXC + 511) / 512;      // 5
    int xl[NXE]; unsigned xoff[NXE]; int xrow[NXE], xcol[NXE];
#pragma unroll
    for (int i = 0; i < NXE; ++i) {
        const int e = tid + 512 * i;
        const int ci = e / (WXR * WXC), r = e - ci * (WXR * WXC);
        const int hy = r / WXC, hx = r - hy * WXC;
        const bool in = e < WBN * WXR * WXC && n0 + ci < Cin;
        xl[i] = e < WBN * WXR * WXC ? (hy * WXC + hx) * LDX + ci : -1;
        xoff[i] = in ? (unsigned)(n0 + ci) * plane : 0x30000000u;
        xrow[i] = hy - 1; xcol[i] = hx - 1;
    }
    const int vci = tid & 31, vt = (tid >> 5) & 7, th = tid >> 8;

    float rx[NXE]; float2 ry[2][2];
    auto chunk_origin = [&](int k, int& img, int& oy, int& ox) {
        const int cg = k % cgs; const int u = k / cgs;
        const int tr = u % trs; img = u / trs;
        oy = 2 * tr; ox = 2 * WCT * cg;
    };
    auto load_y = [&](int k) {
        int img, oy, ox; chunk_origin(k, img, oy, ox);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int p = tid + 512 * r, co = p >> 3, tc = p & 7;
            const bool ok = p < BM * WCT && m0 + co < Cout && k < k_end;
            const unsigned g = (unsigned)(img * Cout + m0 + co) * plane + (unsigned)(oy * W + ox + 2 * tc);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const unsigned o = ok ? (g + rr * W) * 4u : 0xFFFFFFF8u;
                ry[r][rr] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rY, o, 0, 0));
            }
        }
    };
    auto load_x = [&](int k) {
        int img, oy, ox; chunk_origin(k, img, oy, ox);
#pragma unroll
        for (int i = 0; i < NXE; ++i) {
            const int iy = oy + xrow[i], ix = ox + xcol[i];
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && k < k_end;
            rx[i] = ldgx(rX, xoff[i] + (unsigned)(img * Cin) * plane + (unsigned)(iy * W + ix), ok && xoff[i] < 0x30000000u);
        }
    };
    auto store_x = [&](float* Xd, int dump) {
#pragma unroll
        for (int i = 0; i < NXE; ++i) Xd[xl[i] >= 0 ? xl[i] : dump] = rx[i];
    };
    auto transform_q = [&](float* Qd) {                     // Q = A d A^t, A = [[1,0],[1,1],[1,-1],[0,-1]]
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int p = tid + 512 * r, co = p >> 3, tc = p & 7;
            if (p < BM * WCT) {
                const float d00 = ry[r][0].x, d01 = ry[r][0].y, d10 = ry[r][1].x, d11 = ry[r][1].y;
                const float R[4][2] = {{d00, d01}, {d00 + d10, d01 + d11}, {d00 - d10, d01 - d11}, {-d10, -d11}};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float* q = &Qd[((i * 4) * WCT + tc) * LDQ + co];
                    q[0] = R[i][0];
                    q[WCT * LDQ] = R[i][0] + R[i][1];
                    q[2 * WCT * LDQ] = R[i][0] - R[i][1];
                    q[3 * WCT * LDQ] = -R[i][1];
                }
            }
        }
    };
    auto transform_v = [&](const float* Xc, float* Vd) {    // rows 2th, 2th+1 of V = B^t d B for (ci, tile column vt)
        float ra[4], rb[4], rc[4];
        const float* px = &Xc[((th)*WXC + 2 * vt) * LDX + vci];
#pragma unroll
        for (int j = 0; j < 4; ++j) { ra[j] = px[j * LDX]; rb[j] = px[(WXC + j) * LDX]; rc[j] = px[(2 * WXC + j) * LDX]; }
        float u[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t1 = ra[j] - rc[j], t2 = rb[j] - ra[j], t3 = rb[j] + rc[j];
            u[0][j] = th ? t2 : t1;
            u[1][j] = th ? t1 : t3;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* pv = &Vd[(((2 * th + i) * 4) * WCT + vt) * WBN + vci];
            pv[0] = u[i][0] - u[i][2];
            pv[WCT * WBN] = u[i][1] + u[i][2];
            pv[2 * WCT * WBN] = u[i][2] - u[i][1];
            pv[3 * WCT * WBN] = u[i][1] - u[i][3];
        }
    };

    if (k_beg < k_end) {
        load_y(k_beg); load_x(k_beg);
        store_x(Xs, 2 * XSZ);
        load_x(k_beg + 1);
        __syncthreads();
        transform_q(Qs); transform_v(Xs, Vs);
        load_y(k_beg + 1);
        store_x(Xs + XSZ, XSZ);
        load_x(k_beg + 2);
        __syncthreads();
#if MOGAN_X6
        // split-bf16 form: the MFMAs of a chunk pair are issued in its second iteration (8 k-values per lane = 4 of the even
        // + 4 of the odd chunk); chunks past k_end are staged as zeros, so an unpaired last chunk is simply paired with zeros
        float ave[2][WCT / 2][3], bve[2][WCT / 2];
        for (int k = k_beg; k < k_end; k += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int cur = par, nxt = cur ^ 1, kc = k + par;
                const float* Qc = Qs + cur * QSZ;
                const float* Vc = Vs + cur * VSZ;
                float av[2][WCT / 2][3], bv[2][WCT / 2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int kk = 0; kk < WCT / 2; ++kk) {
                        const int row = (wave * 2 + j) * WCT + 2 * kk + h;
                        bv[j][kk] = Vc[row * WBN + l31];
#pragma unroll
                        for (int a = 0; a < 3; ++a) av[j][kk][a] = Qc[row * LDQ + a * 32 + l31];
                    }
                auto do_j = [&](int j) {
                    X6Frag fa[3], fb;
                    float b8[8];
#pragma unroll
                    for (int kk = 0; kk < WCT / 2; ++kk) { b8[kk] = bve[j][kk]; b8[4 + kk] = bv[j][kk]; }
                    fb = x6_split8(b8);
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float a8[8];
#pragma unroll
                        for (int kk = 0; kk < WCT / 2; ++kk) { a8[kk] = ave[j][kk][a]; a8[4 + kk] = av[j][kk][a]; }
                        fa[a] = x6_split8(a8);
                    }
#pragma unroll
                    for (int term = 0; term < 6; ++term)
#pragma unroll
                        for (int a = 0; a < 3; ++a) acc[j][a] = x6_mfma(fa[a], fb, term, acc[j][a]);
                };
                if (par == 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int kk = 0; kk < WCT / 2; ++kk) {
                            bve[j][kk] = bv[j][kk];
#pragma unroll
                            for (int a = 0; a < 3; ++a) ave[j][kk][a] = av[j][kk][a];
                        }
                } else {
                    do_j(0);
                }
                transform_q(Qs + nxt * QSZ);                        // dY(kc+1) (in registers since the last iteration)
                load_y(kc + 2);
                store_x(Xs + cur * XSZ, 2 * XSZ - cur * XSZ);       // X(kc+2)
                load_x(kc + 3);
                if (par == 1) do_j(1);
                transform_v(Xs + nxt * XSZ, Vs + nxt * VSZ);        // X(kc+1) -> V(kc+1)
                __syncthreads();
            }
        }
#else
        for (int k = k_beg; k < k_end; ++k) {
            const int cur = (k - k_beg) & 1, nxt = cur ^ 1;
            const float* Qc = Qs + cur * QSZ;
            const float* Vc = Vs + cur * VSZ;
            float av[2][WCT / 2][3], bv[2][WCT / 2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kk = 0; kk < WCT / 2; ++kk) {
                    const int row = (wave * 2 + j) * WCT + 2 * kk + h;
                    bv[j][kk] = Vc[row * WBN + l31];
#pragma unroll
                    for (int a = 0; a < 3; ++a) av[j][kk][a] = Qc[row * LDQ + a * 32 + l31];
                }
            transform_q(Qs + nxt * QSZ);                        // dY(k+1) (in registers since the last iteration)
            load_y(k + 2);
            store_x(Xs + cur * XSZ, 2 * XSZ - cur * XSZ);       // X(k+2)
            load_x(k + 3);
            transform_v(Xs + nxt * XSZ, Vs + nxt * VSZ);        // X(k+1) -> V(k+1)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kk = 0; kk < WCT / 2; ++kk)
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        acc[j][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][kk][a], bv[j][kk], acc[j][a], 0, 0, 0);
            __syncthreads();
        }
#endif
    }
    // partial dU[sp][xi][co][ci]
    float* out = part + (size_t)sp * 16 * Cout * Cin;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int xi = wave * 2 + j;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n = n0 + l31;
                if (m < Cout && n < Cin) out[((size_t)xi * Cout + m) * Cin + n] = acc[j][a][r];
            }
    }
}

// dW[co][ci] (+)= G^t (sum_sp dU[sp][.][co][ci]) G,  G^t = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
__global__ __launch_bounds__(64) void wino_wgrad_finish(const float* __restrict__ part, float* __restrict__ dw, int Cout,
                                                         int Cin, int nsplit, int accumulate) {
    const long long i = (long long)blockIdx.x * 64 + threadIdx.x;
    const long long n = (long long)Cout * Cin;
    if (i >= n) return;
    float u[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) u[a][b] = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) {                   // 16 independent loads in flight per split, fixed order
        const float* ps = part + (size_t)sp * 16 * n + i;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) u[a][b] += ps[(size_t)(a * 4 + b) * n];
    }
    float t[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        t[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
        t[1][b] = 0.5f * (u[1][b] - u[2][b]);
        t[2][b] = 0.5f * (u[1][b] + u[2][b]) + u[3][b];
    }
    float* d = dw + i * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float g0 = t[a][0] + 0.5f * (t[a][1] + t[a][2]), g1 = 0.5f * (t[a][1] - t[a][2]),
                    g2 = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
        d[a * 3 + 0] = (accumulate ? d[a * 3 + 0] : 0.f) + g0;
        d[a * 3 + 1] = (accumulate ? d[a * 3 + 1] : 0.f) + g1;
        d[a * 3 + 2] = (accumulate ? d[a * 3 + 2] : 0.f) + g2;
    }
}

}  // namespace

// ---- internal entry points (hidden visibility): 1 = handled, 0 = not eligible, < 0 = error -----------------------------
// dgrad = 0: y (B,Cout,H,W) = conv3x3 s1 p1 (x (B,Cin,H,W), w (Cout,Cin,3,3))
// dgrad = 1: dx (B,Cin,H,W) = conv3x3^T (dy (B,Cout,H,W), w): the same kernel over dY with the rotated / transposed filters
// The pre-split filter planes are rebuilt per call at the head of the workspace (wino3_weight_kernel: one thread per operand
// fragment): the generator uses every weight version once per direction, and planes built behind the optimizer step instead
// measured 0.6 % slower in the step (round 4) -- that variant, the round-2 kernel and the lab switches are gone since round 5
// (decided by measurement: in the step this kernel beats the direct kernels on the same layers by 1 %, its weight gradient by
// another 1.2 %, profiles/r05_ab.txt).  Split-bf16 build only: the native-fp32 build takes the direct kernels.
static int g_wino = -1;

// prep != nullptr: the caller's prepared filter image of THIS weight version and direction (mogan_wino_prep_group): no per-call
// weight transform, the workspace is not touched.
int mogan_wino_try(const float* in, const float* w, const void* prep, float* out, int B, int Cin, int H, int W, int Cout, int KH,
                   int KW, int stride, int ph, int pw, int up, int dgrad, const float* ep_scale, const float* ep_shift, int ep_relu,
                   void* ws, size_t ws_bytes, hipStream_t st) {
#if MOGAN_X6
    if (g_wino < 0) { const char* e = getenv("MOGAN_WINO"); g_wino = (e && e[0] == '0') ? 0 : 1; }
    if (!g_wino || !(KH == 3 && KW == 3 && stride == 1 && ph == pw && (ph == 0 || ph == 1) && up == 0)) return 0;
    const int Kin = dgrad ? Cout : Cin, Kout = dgrad ? Cin : Cout;       // channels the kernel reduces over / produces
    // conv: (H, W) -> (H + 2p - 2); its data gradient runs over dY (the smaller grid) with pad 2 - p and produces (H, W)
    const int cH = H + 2 * ph - 2, cW = W + 2 * pw - 2;
    const int iH = dgrad ? cH : H, iW = dgrad ? cW : W, oH = dgrad ? H : cH, oW = dgrad ? W : cW, pad = dgrad ? 2 - ph : ph;
    if (cH < 2 || cW < 2 || (Kin % CK2) || Kin < 32 || Kout < 64) return 0;
    if ((((uintptr_t)out) & 15) != 0) return 0;
    const int tiles_x = (oW + 2 * TCOLS - 1) / (2 * TCOLS), tiles_y = (oH + 2 * TROWS - 1) / (2 * TROWS);
    // ragged grids waste part of every 4 x 32 tile: below 70 % filling the direct kernels win
    if ((double)oW * oH < 0.7 * (double)tiles_x * 2 * TCOLS * tiles_y * 2 * TROWS) return 0;
    const long long mbs = (Kout + BM - 1) / BM;
    // 32-bit byte offsets: input < 2 GiB, and the "reads as zero" sentinel (0xC0000000 bytes) plus a per-image channel
    // offset must neither land inside the buffer nor wrap
    if ((long long)B * Kin * iH * iW >= (1ll << 29) || (long long)Kin * iH * iW >= (1ll << 26) ||
        (long long)B * Kout * oH * oW >= (1ll << 30))
        return 0;
    // (+ one step of padding: the K loop's last iteration prefetches step nstep, whose scalar offset must stay inside the buffer)
    const size_t u3bytes = (size_t)mbs * 8 * (Kin / CK2) * 18 * 1024 + 18 * 1024;
    if (u3bytes >= (1ull << 31)) return 0;
    if (prep == nullptr && (!ws || u3bytes > ws_bytes)) return 0;
    const uint4* U = prep ? (const uint4*)prep : (const uint4*)ws;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    // the 16-wave form where it applies: full 4 x 32 pixel tiles, 96-channel blocks, no fused affine epilogue
    // (-DMOGAN_WINO5=0 builds the library without it: the A/B reference of profiles/r06_ab.txt)
    const int nunit = (int)mbs * 3 * (Kin / CK2);
    if (prep == nullptr)
        hipLaunchKernelGGL(wino3_weight_kernel, dim3((unsigned)((nunit + 3) / 4)), dim3(256), 0, st, w, (uint4*)ws, Cout, Cin, dgrad, nunit);
    if (MOGAN_WINO5 && ep_scale == nullptr && (oH % (2 * TROWS)) == 0 && (oW % (2 * TCOLS)) == 0 && (Kout % BM) == 0) {
        const long long nt5 = (long long)B * tiles_x * tiles_y * mbs;
        mogan_prof_relabel(3);                              // (bench.py roofline leg: this launch is wino5_fwd_kernel's)
        hipLaunchKernelGGL(wino5_fwd_kernel, dim3((unsigned)std::min<long long>(nt5, ncu)), dim3(1024), 0, st, in, U, out,
                           Kin, iH, iW, Kout, oH, oW, pad, tiles_x, tiles_y, (int)nt5, (unsigned)(4ull * B * Kin * iH * iW),
                           (unsigned)u3bytes);
        return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
    }
    const long long ntile = (long long)B * tiles_x * tiles_y * mbs;
    if (ntile >= (1ll << 30)) return 0;
    // persistent: one 8-wave block per CU walks the tiles
    dim3 grid((unsigned)std::min<long long>(ntile, ncu));
    hipLaunchKernelGGL(wino3_fwd_kernel, grid, dim3(512), 0, st, in, U, out, Kin, iH, iW, Kout, oH, oW, pad, tiles_x,
                       tiles_y, (int)ntile, B, ep_scale, ep_shift, ep_relu, (unsigned)(4ull * B * Kin * iH * iW), (unsigned)u3bytes);
    return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
#else
    return 0;
#endif
}

// dw (Cout,Cin,3,3) (+)= weight gradient of conv3x3 s1 p1; workspace: nsplit * 16 * Cout * Cin floats
int mogan_wino_wgrad_try(const float* dy, const float* x, float* dw, int B, int Cin, int H, int W, int Cout, int KH, int KW,
                         int stride, int ph, int pw, int up, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    if (g_wino < 0) { const char* e = getenv("MOGAN_WINO"); g_wino = (e && e[0] == '0') ? 0 : 1; }
    // MOGAN_WINO_WGRAD=0 falls back to the direct kernel (2 = timing aid: main kernel without the finish pass)
    constexpr int wg_on = 1;
    if (!g_wino || !wg_on || !(KH == 3 && KW == 3 && stride == 1 && ph == 1 && pw == 1 && up == 0)) return 0;
    if (Cin < 32 || Cout < 64 || (H % 2) || (W % (2 * WCT)) || (W % 2)) return 0;
    if ((((uintptr_t)dy) & 7) != 0) return 0;
    if ((long long)B * Cin * H * W >= (1ll << 30) || (long long)B * Cout * H * W >= (1ll << 30)) return 0;
    const int nchunk = B * (H / 2) * (W / (2 * WCT));
    const int tiles_mn = ((Cout + BM - 1) / BM) * ((Cin + WBN - 1) / WBN);
    // one 8-wave block per CU and ONE round of blocks: the K-split partials (16 x Cout x Cin floats per split) are what the
    // finish pass has to read back (256 blocks: 156 / 137 TF direct-equivalent at 128x128 / 64x64; 512: 147 / 117).  In the train step
    // this kernel runs on the weight-gradient side stream BESIDE the generator's data-gradient chain: with 256 persistent blocks it
    // holds every CU for ~240 us and the chain's short kernels queue behind it; 192 blocks leave a quarter of the CUs to the chain:
    // 405.0 / 406.6 vs 402.3 / 399.3 img/s (224: 404.4, 160: 403.6; tools/wino_wg_blocks_probe.sh) -- the default since round 3
    constexpr int wg_blocks = 192;
    int nsplit = wg_blocks / tiles_mn;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > nchunk / 8) nsplit = nchunk / 8 > 0 ? nchunk / 8 : 1;
    const size_t slab = (size_t)16 * Cout * Cin * sizeof(float);
    if (!ws || ws_bytes < slab) return 0;
    if ((size_t)nsplit * slab > ws_bytes) nsplit = (int)(ws_bytes / slab);
    const int cps = (nchunk + nsplit - 1) / nsplit;
    nsplit = (nchunk + cps - 1) / cps;
    dim3 grid((unsigned)((Cin + WBN - 1) / WBN), (unsigned)((Cout + BM - 1) / BM), (unsigned)nsplit);
    hipLaunchKernelGGL(wino_wgrad_kernel, grid, dim3(512), 0, st, dy, x, (float*)ws, Cin, H, W, Cout, nchunk, cps,
                       (unsigned)(4ull * B * Cout * H * W), (unsigned)(4ull * B * Cin * H * W));
    const long long n = (long long)Cout * Cin;
    if (wg_on != 2)     // (2 = timing aid: main kernel only)
    hipLaunchKernelGGL(wino_wgrad_finish, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (const float*)ws, dw, Cout,
                       Cin, nsplit, accumulate);
    return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
}

// ---- prepared filter images owned by the caller (include/mogan_hip.h: mogan_wino_prep_bytes / mogan_wino_prep_group) ---------
size_t mogan_wino_prep_bytes(int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int up, int dgrad) {
#if MOGAN_X6
    if (g_wino < 0) { const char* e = getenv("MOGAN_WINO"); g_wino = (e && e[0] == '0') ? 0 : 1; }
    if (!g_wino || !(KH == 3 && KW == 3 && stride == 1 && ph == pw && (ph == 0 || ph == 1) && up == 0)) return 0;
    const int Kin = dgrad ? Cout : Cin, Kout = dgrad ? Cin : Cout;
    const int cH = Hs + 2 * ph - 2, cW = Ws + 2 * pw - 2;
    const int iH = dgrad ? cH : Hs, iW = dgrad ? cW : Ws, oH = dgrad ? Hs : cH, oW = dgrad ? Ws : cW;
    if (cH < 2 || cW < 2 || (Kin % CK2) || Kin < 32 || Kout < 64) return 0;
    const int tiles_x = (oW + 2 * TCOLS - 1) / (2 * TCOLS), tiles_y = (oH + 2 * TROWS - 1) / (2 * TROWS);
    if ((double)oW * oH < 0.7 * (double)tiles_x * 2 * TCOLS * tiles_y * 2 * TROWS) return 0;
    const long long mbs = (Kout + BM - 1) / BM;
    if ((long long)B * Kin * iH * iW >= (1ll << 29) || (long long)Kin * iH * iW >= (1ll << 26) ||
        (long long)B * Kout * oH * oW >= (1ll << 30))
        return 0;
    const size_t u3bytes = (size_t)mbs * 8 * (Kin / CK2) * 18 * 1024 + 18 * 1024;
    return u3bytes < (1ull << 31) ? u3bytes : 0;
#else
    return 0;
#endif
}

int mogan_wino_prep_group(int n, const float* const* w, void* const* prep, const int* Cout, const int* Cin, const int* dgrad,
                          hipStream_t st) {
#if MOGAN_X6
    if (n < 0 || (n > 0 && (!w || !prep || !Cout || !Cin || !dgrad))) return MOGAN_ERR_SHAPE;
    for (int i0 = 0; i0 < n; i0 += WPG_MAX) {
        WinoPrepGroup g{}; unsigned tot = 0;
        g.n = std::min(WPG_MAX, n - i0);
        for (int i = 0; i < g.n; ++i) {
            const int k = i0 + i;
            const int Kin = dgrad[k] ? Cout[k] : Cin[k], Kout = dgrad[k] ? Cin[k] : Cout[k];
            if (!w[k] || !prep[k] || (Kin % CK2) || Kin < 32 || Kout < 64 || (((uintptr_t)prep[k]) & 15)) return MOGAN_ERR_SHAPE;
            const int nunit = ((Kout + BM - 1) / BM) * 3 * (Kin / CK2);
            g.w[i] = w[k]; g.u3[i] = (uint4*)prep[k]; g.Cout[i] = Cout[k]; g.Cin[i] = Cin[k]; g.flip[i] = dgrad[k] ? 1 : 0;
            tot += (unsigned)((nunit + 3) / 4);
            g.end[i] = tot;
        }
        hipLaunchKernelGGL(wino3_weight_group_kernel, dim3(tot), dim3(256), 0, st, g);
    }
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
#else
    return MOGAN_ERR_SHAPE;
#endif
}
