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
#ifndef WINO_ORDER
#define WINO_ORDER 1
#endif
#ifndef WINO_ROLE
#define WINO_ROLE 1
#endif
#ifndef WINO_SLEEP
#define WINO_SLEEP 8
#endif

__global__ __launch_bounds__(256) void wino3_weight_kernel(const float* __restrict__ w, uint4* __restrict__ U3, int Cout,
                                                           int Cin, int flip, long long nfrag) {
    const long long gidx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gidx >= nfrag) return;
    const int Kin = flip ? Cout : Cin, Kout = flip ? Cin : Cout;
    const int nstep = Kin / CK2;
    const int lane = (int)(gidx & 63); long long rr = gidx >> 6;
    const int a3 = (int)(rr % 3); rr /= 3;
    const int j = (int)(rr & 1); rr >>= 1;
    const int step = (int)(rr % nstep); rr /= nstep;
    const int wv = (int)(rr & 7); const int mb = (int)(rr >> 3);
    const int h = lane >> 5, l31 = lane & 31;
    const int xi = 2 * wv + j, r = xi >> 2, q = xi & 3;
    const int kout = mb * BM + a3 * 32 + l31;
    const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
    float u8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kin = step * CK2 + 8 * h + e;
        float u = 0.f;
        if (kout < Kout) {
            const int co = flip ? kin : kout, ci = flip ? kout : kin;
            const float* g = w + ((size_t)co * Cin + ci) * 9;
            float t[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float tt = 0.f;
#pragma unroll
                for (int a = 0; a < 3; ++a) tt += G[r][a] * (flip ? g[(2 - a) * 3 + (2 - b)] : g[a * 3 + b]);
                t[b] = tt;
            }
            u = G[q][0] * t[0] + G[q][1] * t[1] + G[q][2] * t[2];
        }
        u8[e] = u;
    }
    const X6Frag f = x6_split8(u8);
    const size_t base = ((((size_t)(mb * 8 + wv) * nstep + step) * 2 + j) * 3 + a3) * 3 * 64 + lane;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) U3[base + (size_t)pc * 64] = __builtin_bit_cast(uint4, f.p[pc]);
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
#if WINO_ROLE == 1
    const bool roleB = (wave_s & 4) != 0;       // waves w and w + 4 share a SIMD (round-robin placement)
#elif WINO_ROLE == 2
    const bool roleB = (wave_s & 1) != 0;
#endif
    const unsigned ulane = (unsigned)lane * 16u;
    unsigned xg, ubase; int m0, img, oy0, ox0;
    // work item = (spatial tile, output-channel block) with the channel block INNERMOST, a block walks a contiguous range of
    // items: the channel blocks of one spatial tile run back to back on one CU (the second one's halo comes out of L1 / L2, the
    // input leaves HBM once per layer), and vertically adjacent tiles -- which share two halo rows -- stay on the CU as well
    auto plan = [&](int tile, unsigned& g, unsigned& ub, int& m0_, int& img_, int& oy_, int& ox_) {
        int t = tile;
#if WINO_ORDER
        const int mb = t % mbs; t /= mbs;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y; t /= tiles_y;
        img_ = t;
#else
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y; t /= tiles_y;
        img_ = t % nimg; const int mb = t / nimg;
#endif
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

#if WINO_ORDER
    const int per = (ntile + (int)gridDim.x - 1) / (int)gridDim.x;
    int tile = blockIdx.x * per;
    const int tile_end = min(ntile, tile + per), tstep = 1;
#else
    int tile = blockIdx.x;
    const int tile_end = ntile, tstep = gridDim.x;
#endif
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
#if WINO_ROLE
            // the two waves of a SIMD in lockstep sit in their vector / LDS phases at the same time and the matrix pipe idles;
            // one of them starts every step late by about one MFMA group, so that its MFMAs fall into the other's staging /
            // transform phases (same instruction stream for both: no second loop body, no extra registers)
            if (roleB) __builtin_amdgcn_s_sleep(WINO_SLEEP);
#endif
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

// ------------------------------------------------------------------------------------------ round 6: forward, fourth form
// wino3_fwd_kernel is bound by the vector-memory path, not by the matrix pipe (MFMA busy 0.23): with 32 tiles per block every
// 1 KB filter fragment feeds six MFMAs (0.5 KB per MFMA, 147 KB per K step and CU -- the L1's whole 64 B/clk at the full matrix
// rate, ~12 TB/s out of the L2s over the chip), and the output transform moves all 16 transform positions of every (channel,
// tile) through LDS (16 values x 4 B, written at 64 B/clk) because a wave holds two positions only.  This form changes both:
//   * a block = 4 waves (one per SIMD, the accumulators in the AGPR half of the 512-entry file) owns MT*32 output channels x 64
//     tiles (4 tile rows x 16 tile columns = 8 x 32 output pixels) of one image; wave j owns COLUMN j of the 4 x 4 transform
//     positions, all four rows i, as 4 x MT x 2 accumulator tiles: a filter fragment feeds 12 MFMAs (0.25 KB per MFMA);
//   * the row half of the output transform (Y = A^t M A: P[a][j] = sum_i A^t[a][i] M[i][j]) is done in registers; only the two
//     P values per (channel, tile) and wave cross waves, once, through LDS -- 8 values instead of 16, one pass, two barriers;
//   * the halo arrives as 16-byte quads (9 per row and channel, loaded at the halo's own 4-byte alignment) and is staged with
//     16-byte LDS stores; a transform thread owns (tile, 4 channels) and writes V channel-innermost, so that a B fragment is two
//     ds_read_b128 (the 16-byte slots of a tile's 16 channels are XOR-swizzled with (tile >> 2) & 3: conflict-free in the
//     instruction's 16-lane groups);
//   * work items = (tile group, channel block) with the channel block innermost over a contiguous range per persistent block:
//     the second / third channel block of a group reads its halo out of L1 / L2, the input leaves HBM once per layer.
// LDS: V 2 x 64 KB (fp32, split at the fragment read as before: every V element is read by exactly one wave) + X 22.5 KB.
// Filters: U4[m-tile][K step][j][i][piece][lane] x 16 bytes (wino4_weight_kernel), 12 KB contiguous per (m-tile, step, wave).
// Cout % 64 == 32 (the 96-channel layers): the last 32 channels run as a second launch of the MT = 1 instantiation.
// issue order of a phase: per MFMA `nv` vector-ALU instructions, `nd` LDS instructions and `nm` vector-memory instructions (the
// wave is alone on its SIMD: what is not issued between two MFMAs does not overlap with the matrix pipe at all)
#ifndef W4_SCHED
#define W4_SCHED 1
#endif
#ifndef W4_LAB
#define W4_LAB 0      // lab builds (wrong results): 1 no halo loads, 2 no filter loads, 4 no input transform, 8 no MFMAs in the K loop
#endif
#if W4_SCHED
#define W4_PIPE(nmfma, nv, nd, nm)                                         \
    _Pragma("unroll") for (int m_ = 0; m_ < (nmfma); ++m_) {               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
        __builtin_amdgcn_sched_group_barrier(0x002, (nv), 0);              \
        __builtin_amdgcn_sched_group_barrier(0x080, (nd), 0);              \
        __builtin_amdgcn_sched_group_barrier(0x010, (nm), 0);              \
    }
#else
#define W4_PIPE(nmfma, nv, nd, nm)
#endif
#ifndef W4_V01
#define W4_V01 6
#endif
#ifndef W4_V2
#define W4_V2 6
#endif
#ifndef W4_V3
#define W4_V3 3
#endif
constexpr int W4_TY = 4, W4_TX = 16, W4_NT = W4_TY * W4_TX, W4_XR = 2 * W4_TY + 2, W4_XQ = 9, W4_XW = 4 * W4_XQ;
constexpr int W4_XSZ = CK2 * W4_XR * W4_XW, W4_VSZ = 16 * W4_NT * CK2, W4_NXL = (CK2 * W4_XR * W4_XQ + 255) / 256;

__global__ __launch_bounds__(256) void wino4_weight_kernel(const float* __restrict__ w, uint4* __restrict__ U4, int Cout,
                                                           int Cin, int flip, long long nfrag) {
    const long long gidx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gidx >= nfrag) return;
    const int Kin = flip ? Cout : Cin, Kout = flip ? Cin : Cout;
    const int nstep = Kin / CK2;
    const int lane = (int)(gidx & 63); long long rr = gidx >> 6;
    const int i = (int)(rr & 3); rr >>= 2;
    const int j = (int)(rr & 3); rr >>= 2;
    const int step = (int)(rr % nstep); const int mt = (int)(rr / nstep);
    const int h = lane >> 5, l31 = lane & 31;
    const int kout = mt * 32 + l31;
    const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
    float u8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kin = step * CK2 + 8 * h + e;
        float u = 0.f;
        if (kout < Kout) {
            const int co = flip ? kin : kout, ci = flip ? kout : kin;
            const float* g = w + ((size_t)co * Cin + ci) * 9;
            float t[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float tt = 0.f;
#pragma unroll
                for (int a = 0; a < 3; ++a) tt += G[i][a] * (flip ? g[(2 - a) * 3 + (2 - b)] : g[a * 3 + b]);
                t[b] = tt;
            }
            u = G[j][0] * t[0] + G[j][1] * t[1] + G[j][2] * t[2];
        }
        u8[e] = u;
    }
    const X6Frag f = x6_split8(u8);
    const size_t base = ((((size_t)mt * nstep + step) * 4 + j) * 4 + i) * 3 * 64 + lane;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) U4[base + (size_t)pc * 64] = __builtin_bit_cast(uint4, f.p[pc]);
}

template <int MT>
__global__ __launch_bounds__(256) void wino4_fwd_kernel(const float* __restrict__ X, const uint4* __restrict__ U4,
                                                        float* __restrict__ Y, int Cin, int H, int W, int Cout, int gx, int gy,
                                                        int nitem, int mbs, int mt0, unsigned x_bytes, unsigned u_bytes) {
    __shared__ __attribute__((aligned(16))) float Vs[2 * W4_VSZ];      // V of the K loop; the epilogue's exchange buffer
    __shared__ __attribute__((aligned(16))) float Xs[W4_XSZ + 4 * 256];   // (+ a dump quad per thread: the idle lanes of the last staging slot store unconditionally)
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wj = __builtin_amdgcn_readfirstlane(tid >> 6);         // this wave's column of transform positions
    const int plane = H * W;
    const int nstep = Cin / CK2;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)U4, (short)0, (int)u_bytes, 0x00020000);

    // halo staging role: slot s = tid + 256 k -> (channel, halo row, quad) of a K step; quad q holds columns ox0 - 1 + 4 q .. + 3
    int xs_l[W4_NXL];
#pragma unroll
    for (int k = 0; k < W4_NXL; ++k) {
        const int sl = tid + 256 * k;
        const int c = sl / (W4_XR * W4_XQ), rem = sl - c * (W4_XR * W4_XQ);
        const int r = rem / W4_XQ, q = rem - r * W4_XQ;
        xs_l[k] = sl < CK2 * W4_XR * W4_XQ ? (c * W4_XR + r) * W4_XW + 4 * q : W4_XSZ + 4 * tid;
    }
    // transform role: tile tt (row tty, column ttx), channel pair 2 cq, 2 cq + 1 and 2 cq + 8, 2 cq + 9 of a step
    const int tt = tid & 63, tty = tt >> 4, ttx = tt & 15, cq = tid >> 6;
    const int tsw = (tt >> 2) & 3;
    const int tx0 = (2 * tty) * W4_XW + 2 * ttx;
    // B fragment of tile block tb: tile tb * 32 + l31, channels 8 h .. 8 h + 7 = slots 2 h, 2 h + 1 (swizzled)
    int bsl[2][2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int tile = tb * 32 + l31, sw = (tile >> 2) & 3;
        bsl[tb][0] = tile * CK2 + ((2 * h) ^ sw) * 4;
        bsl[tb][1] = tile * CK2 + ((2 * h + 1) ^ sw) * 4;
    }
    const unsigned ulane = (unsigned)lane * 16u;

    unsigned xg[W4_NXL]; unsigned xedge = 0, xedge_cur = 0;      // bit k: slot k is the left-edge quad, bit 8 + k: the right-edge quad
    unsigned ubase; int m0, img, oy0, ox0;
    auto plan = [&](int item) {
        int t = item;
        const int mb = t % mbs; t /= mbs;
        const int gxi = t % gx; t /= gx;
        const int gyi = t % gy; t /= gy;
        img = t;
        m0 = (mt0 + mb * MT) * 32; oy0 = gyi * 2 * W4_TY; ox0 = gxi * 2 * W4_TX;
        xedge = 0;
#pragma unroll
        for (int k = 0; k < W4_NXL; ++k) {
            const int sl = tid + 256 * k;
            const int c = sl / (W4_XR * W4_XQ), rem = sl - c * (W4_XR * W4_XQ);
            const int r = rem / W4_XQ, q = rem - r * W4_XQ;
            const int iy = oy0 - 1 + r, ix = ox0 - 1 + 4 * q;
            const bool ok = sl < CK2 * W4_XR * W4_XQ && (unsigned)iy < (unsigned)H;
            // a quad that sticks out of its row is loaded from inside the row and shifted (load_x): column -1 of the leftmost
            // group, columns W .. W + 2 of the rightmost one read as zeros, and no load leaves the tensor
            const bool el = ix < 0, er = ix + 4 > W;
            const int ixa = el ? 0 : er ? ix - 3 : ix;
            xg[k] = ok ? ((unsigned)(img * Cin + c) * plane + (unsigned)(iy * W + ixa)) * 4u : 0xC0000000u;     // reads 0
            xedge |= (el ? 1u : 0u) << k | (er ? 256u : 0u) << k;
        }
        ubase = (unsigned)((((mt0 + mb * MT) * nstep) * 4 + wj) * 4) * 3072u;
    };
    f32x4 rx[W4_NXL];
    auto load_x = [&](int c0) {
#pragma unroll
        for (int k = 0; k < W4_NXL; ++k)
            rx[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, xg[k] + (unsigned)c0 * (unsigned)plane * 4u, 0, 0));
    };
    auto store_x = [&]() {                                  // (the edge shift happens here, not at the load: the loads stay in flight)
#pragma unroll
        for (int k = 0; k < W4_NXL; ++k) {
            const f32x4 v = rx[k];
            const bool el = (xedge_cur >> k) & 1u, er = (xedge_cur >> (8 + k)) & 1u;
            f32x4 o;
            o[0] = el ? 0.f : er ? v[3] : v[0];
            o[1] = el ? v[0] : er ? 0.f : v[1];
            o[2] = el ? v[1] : er ? 0.f : v[2];
            o[3] = el ? v[2] : er ? 0.f : v[3];
            *(f32x4*)&Xs[xs_l[k]] = o;
        }
    };
    // filter fragments of position row i, K step `step`: MT m-tiles x 3 pieces
    auto load_a = [&](X6Frag (&fa)[MT], int i, int step) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                fa[a].p[pc] = __builtin_bit_cast(mma_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                    rU, ulane, ubase + (unsigned)((a * nstep + step) * 16 + i) * 3072u + (unsigned)pc * 1024u, 0));
    };
    // input transform of channels c, c + 1 (c = 2 cq + c8) of the staged K step: V = B^t d B, all 16 positions, 8-byte stores
    auto transform = [&](float* Vd, int c8) {
        float v[2][4][4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float* px = &Xs[(2 * cq + c8 + e) * (W4_XR * W4_XW) + tx0];
            float d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 lo = *(const float2*)(px + r * W4_XW), hi = *(const float2*)(px + r * W4_XW + 2);
                d[r][0] = lo.x; d[r][1] = lo.y; d[r][2] = hi.x; d[r][3] = hi.y;
            }
            float t[4][4];                                   // t = B^t d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = d[0][c] - d[2][c]; t[1][c] = d[1][c] + d[2][c];
                t[2][c] = d[2][c] - d[1][c]; t[3][c] = d[1][c] - d[3][c];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[e][r][0] = t[r][0] - t[r][2]; v[e][r][1] = t[r][1] + t[r][2];
                v[e][r][2] = t[r][2] - t[r][1]; v[e][r][3] = t[r][1] - t[r][3];
            }
        }
        const int c = 2 * cq + c8;
        float* pv = &Vd[tt * CK2 + (((c >> 2) ^ tsw) * 4) + (c & 3)];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) *(float2*)(pv + (r * 4 + q) * (W4_NT * CK2)) = make_float2(v[0][r][q], v[1][r][q]);
    };

    f32x16 acc[4][MT][2];
    // B operand of position (i, wj): raw fp32 fragment reads, split one phase ahead of the MFMAs that consume it
    struct RawB { f32x4 lo[2], hi[2]; };
    auto read_b = [&](int i, const float* Vc) {
        const float* pb = Vc + (i * 4 + wj) * (W4_NT * CK2);
        RawB r;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) { r.lo[tb] = *(const f32x4*)(pb + bsl[tb][0]); r.hi[tb] = *(const f32x4*)(pb + bsl[tb][1]); }
        return r;
    };
    auto split_b = [&](const RawB& r, X6Frag (&fb)[2]) {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const float b8[8] = {r.lo[tb][0], r.lo[tb][1], r.lo[tb][2], r.lo[tb][3], r.hi[tb][0], r.hi[tb][1], r.hi[tb][2], r.hi[tb][3]};
            fb[tb] = x6_split8(b8);
        }
    };
    auto mma_i = [&](int i, const X6Frag (&fa)[MT], const X6Frag (&fb)[2]) {
#if W4_LAB & 8
        acc[i][0][0][0] += __builtin_bit_cast(float, (int)fb[0].p[0][0] + (int)fb[1].p[2][1] + (int)fa[0].p[1][0] + (int)fa[MT - 1].p[2][3]);
        return;
#endif
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) acc[i][a][tb] = x6_mfma(fa[a], fb[tb], term, acc[i][a][tb]);
    };

    const int per = (nitem + (int)gridDim.x - 1) / (int)gridDim.x;
    int item = blockIdx.x * per;
    const int item_end = min(nitem, item + per);
    X6Frag fa0[MT], fa1[MT], fb0[2], fb1[2];
    if (item < item_end) { plan(item); load_x(0); }
    for (; item < item_end; ++item) {
        xedge_cur = xedge;                                  // (the registers hold this item's first halo from here on)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][a][tb][r] = 0.f;
        load_a(fa0, 0, 0); load_a(fa1, 1, 0);
        store_x();                                          // X(0)   (Xs is not part of the exchange buffer)
        load_x(CK2);
        __syncthreads();                                    // X(0) staged; every thread is done with the previous item's exchange buffer
        transform(Vs, 0); transform(Vs, 8);
        __syncthreads();                                    // V(0) complete; Xs free
        store_x();                                          // X(1)
        load_x(2 * CK2);
        { const RawB rb = read_b(0, Vs); split_b(rb, fb0); }
        __syncthreads();
        // One K step = four phases, one per position row i: the 12 MT MFMAs of row i run beside the split of row i + 1's B operand
        // and a slice of the staging work.  Two barriers: B1 -- V(p+1) complete (phase 3 already reads it for row 0 of the next
        // step), all transform reads of X(p+1) done; B2 -- X(p+2) staged, all reads of V(p) done.
        for (int p = 0; p < nstep; ++p) {
            const float* Vc = Vs + (p & 1) * W4_VSZ;
            float* Vn = Vs + ((p & 1) ^ 1) * W4_VSZ;
            {   // phase 0
                const RawB rb = read_b(1, Vc);
                mma_i(0, fa0, fb0);
                split_b(rb, fb1);
#if !(W4_LAB & 4)
                transform(Vn, 0);                           // X(p+1) -> V(p+1)
#endif
#if !(W4_LAB & 2)
                load_a(fa0, 2, p);
#endif
                W4_PIPE(12 * MT, W4_V01, 2, 1);
            }
            {   // phase 1
                const RawB rb = read_b(2, Vc);
                mma_i(1, fa1, fb1);
                split_b(rb, fb0);
#if !(W4_LAB & 4)
                transform(Vn, 8);
#endif
#if !(W4_LAB & 2)
                load_a(fa1, 3, p);
#endif
                W4_PIPE(12 * MT, W4_V01, 2, 1);
            }
            __syncthreads();
            {   // phase 2
                const RawB rb = read_b(3, Vc);
                mma_i(2, fa0, fb0);
                split_b(rb, fb1);
                store_x();                                  // X(p+2)
#if !(W4_LAB & 1)
                load_x((p + 3) * CK2);
#endif
#if !(W4_LAB & 2)
                load_a(fa0, 0, p + 1);
#endif
                W4_PIPE(12 * MT, W4_V2, 1, 1);
            }
            {   // phase 3
                const RawB rb = read_b(0, Vn);
                mma_i(3, fa1, fb1);
                split_b(rb, fb0);
#if !(W4_LAB & 2)
                load_a(fa1, 1, p + 1);
#endif
                W4_PIPE(12 * MT, W4_V3, 1, 1);
            }
            __syncthreads();
        }
        const int cm0 = m0, cimg = img, coy0 = oy0, cox0 = ox0;
        if (item + 1 < item_end) { plan(item + 1); load_x(0); }
        // ---- output transform: rows in registers, columns across the four waves through LDS
        //      P[a][j]: a = 0: M0 + M1 + M2, a = 1: M1 - M2 - M3;  Ts[j][a][co][tile]
        float* const Ts = Vs;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float m1 = acc[1][a][tb][r], m2 = acc[2][a][tb][r];
                    const float p0 = acc[0][a][tb][r] + m1 + m2, p1 = m1 - m2 - acc[3][a][tb][r];
                    const int co = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float* q = &Ts[((wj * 2) * (MT * 32) + co) * W4_NT + tb * 32 + l31];
                    q[0] = p0;
                    q[(MT * 32) * W4_NT] = p1;
                }
        __syncthreads();
        // item = (output channel, pair of horizontally adjacent tiles): MT*32 x 32 items, MT*4 per thread; a 2 x 4 pixel patch each
#pragma unroll
        for (int k = 0; k < MT * 4; ++k) {
            const int it = tid + 256 * k;
            const int pr = it & 31, co = it >> 5;
            float2 pj[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < 2; ++a) pj[j][a] = *(const float2*)&Ts[((j * 2 + a) * (MT * 32) + co) * W4_NT + 2 * pr];
            const int ty = pr >> 3, px = pr & 7;
            float* o = Y + ((size_t)(cimg * Cout + cm0 + co) * H + coy0 + 2 * ty) * W + cox0 + 4 * px;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                // Y[a][0] = P0 + P1 + P2, Y[a][1] = P1 - P2 - P3 (columns), for the two tiles of the pair
                const float y00 = pj[0][a].x + pj[1][a].x + pj[2][a].x, y01 = pj[1][a].x - pj[2][a].x - pj[3][a].x;
                const float y10 = pj[0][a].y + pj[1][a].y + pj[2][a].y, y11 = pj[1][a].y - pj[2][a].y - pj[3][a].y;
                *(float4*)(o + (size_t)a * W) = make_float4(y00, y01, y10, y11);
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
    const int n0 = blockIdx.x * WBN, m0 = blockIdx.y * BM, sp = blockIdx.z;
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

int mogan_wino_try(const float* in, const float* w, float* out, int B, int Cin, int H, int W, int Cout, int KH, int KW,
                   int stride, int ph, int pw, int up, int dgrad, const float* ep_scale, const float* ep_shift, int ep_relu,
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
    if (!ws || u3bytes > ws_bytes || u3bytes >= (1ull << 31)) return 0;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    // fourth form (wino4_fwd_kernel): well-filled 8 x 32 pixel groups, pad 1, no fused affine epilogue
    static const int w4_on = getenv("MOGAN_WINO4") ? atoi(getenv("MOGAN_WINO4")) : 1;
    if (w4_on && pad == 1 && ph == 1 && ep_scale == nullptr && (oH % (2 * W4_TY)) == 0 && (oW % (2 * W4_TX)) == 0 && (Kout % 32) == 0 &&
        (((uintptr_t)in) & 3) == 0) {
        const int Mt = Kout / 32, nstep = Kin / CK2;
        const size_t u4bytes = (size_t)Mt * nstep * 16 * 3072;
        if (u4bytes <= ws_bytes && u4bytes < (1ull << 31)) {
            const int gx = oW / (2 * W4_TX), gy = oH / (2 * W4_TY);
            const long long nfrag4 = (long long)Mt * nstep * 16 * 64;
            hipLaunchKernelGGL(wino4_weight_kernel, dim3((unsigned)((nfrag4 + 255) / 256)), dim3(256), 0, st, w, (uint4*)ws, Cout, Cin,
                               dgrad, nfrag4);
            const unsigned xb = (unsigned)(4ull * B * Kin * iH * iW);
            const int mb2 = Mt / 2;                              // channel blocks of 64; a last one of 32 goes out as a second launch
            if (mb2 > 0) {
                const long long nitem = (long long)B * gx * gy * mb2;
                hipLaunchKernelGGL(wino4_fwd_kernel<2>, dim3((unsigned)std::min<long long>(nitem, ncu)), dim3(256), 0, st, in,
                                   (const uint4*)ws, out, Kin, iH, iW, Kout, gx, gy, (int)nitem, mb2, 0, xb, (unsigned)u4bytes);
            }
            if (Mt & 1) {
                const long long nitem = (long long)B * gx * gy;
                hipLaunchKernelGGL(wino4_fwd_kernel<1>, dim3((unsigned)std::min<long long>(nitem, ncu)), dim3(256), 0, st, in,
                                   (const uint4*)ws, out, Kin, iH, iW, Kout, gx, gy, (int)nitem, 1, Mt - 1, xb, (unsigned)u4bytes);
            }
            return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
        }
    }
    const long long ntile = (long long)B * tiles_x * tiles_y * mbs;
    if (ntile >= (1ll << 30)) return 0;
    // persistent: one 8-wave block per CU walks the tiles
    dim3 grid((unsigned)std::min<long long>(ntile, ncu));
    const long long nfrag = mbs * 8 * (Kin / CK2) * 2 * 3 * 64;
    hipLaunchKernelGGL(wino3_weight_kernel, dim3((unsigned)((nfrag + 255) / 256)), dim3(256), 0, st, w, (uint4*)ws, Cout, Cin, dgrad,
                       nfrag);
    hipLaunchKernelGGL(wino3_fwd_kernel, grid, dim3(512), 0, st, in, (const uint4*)ws, out, Kin, iH, iW, Kout, oH, oW, pad, tiles_x,
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
