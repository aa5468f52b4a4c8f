// mogan_dconv.hip -- direct ("halo tile in LDS") fp32 MFMA convolution for the shapes that carry the FLOPs of the
// AttnGAN step: 3x3 s1 p1 (optionally behind a fused nearest-x2 upsample) and 4x4 s2 p1 at >= 16x16 output, plus the
// 2x2 s1 sub-convolutions a 4x4 s2 dgrad decomposes into.
//
// Why a second conv kernel: the implicit-GEMM kernel (mogan_gemm.hip) re-gathers every input pixel KH*KW times from
// global memory with per-element address arithmetic; measured on MI355X it tops out near 100 TFLOP/s while the same
// loop without the gather runs at 141 (tools/lab/gemm_lab.hip).  Here a block stages, per chunk of CK input channels,
//   Xs[CK][HH][WWP]  the input halo tile of its R x Cw output pixels (each input pixel loaded ONCE), and
//   Ws[BM][CK*KH*KW] the weights of its BM output channels (16-byte loads),
// and the MFMA operands are read from LDS with compile-time immediate offsets -- no address VALU in the inner loop:
//   a = Ws[m][(2c+h)*KHW + tap]   b = Xs[2c+h][ry*S+kh][rx*S+kw]      (h = lane>>5 selects the channel of the k-pair)
// K order = (ci, kh, kw) = the natural weight layout, so A rows are contiguous.
//
// Forward, data-gradient and (separate kernel below) weight-gradient all use this tile scheme:
//   dgrad s1  = forward over dY with flipped/transposed weights (a small transform kernel builds them),
//   dgrad s2  = four 2x2 s1 forwards (one per output parity) writing to stride-2 positions,
//   wgrad     = K runs over the pixels of the tile, B = Xs read at [ci][ry*S+kh][rx*S+kw] with (ci,kh,kw) on the lanes.
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

__device__ __forceinline__ float ldg(__amdgpu_buffer_rsrc_t r, unsigned idx) {   // idx = 0x3FFFFFFF -> 0.f
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, idx * 4u, 0, 0));
}
__device__ __forceinline__ f32x4 ldg4(__amdgpu_buffer_rsrc_t r, unsigned idx) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, idx * 4u, 0, 0));
}
#define IDX_OOB 0x3FFFFFFFu   // *4 = 0xFFFFFFFC: beyond any buffer extent -> the load returns 0

struct DConvP {
    const float* X; const float* Wt; float* Y; float* ws;
    int B, Cin, Cout, Hs, Ws, H, W, up;      // stored input dims (Hs,Ws); conv sees H = Hs<<up
    int OH, OW, pt, pl;                      // conv output grid and top/left padding
    int yH, yW, ys, y0, x0;                  // output plane dims, pixel stride and offset (strided dgrad writes)
    int R, Cw, tiles_x, tiles_y;             // block = R rows x Cw cols of one image
    int nsplit, cps;                         // split over channel chunks: chunks per split
    int npar;                                // 4: blockIdx.z also enumerates the stride-2 dgrad parity classes
    long long slab;
    int accumulate;
    unsigned x_bytes, w_bytes;
    const void* Wp; unsigned wp_bytes;       // split-bf16 build: the filters as bf16 pieces in fragment order (dconv_wprep_kernel)
    const void* d2prep; size_t* d2query;     // host side only: handed on to mogan_dconv2_fwd_try (mogan_internal.h)
};

#if MOGAN_X6
// Filters for dconv_fwd_kernel, split-bf16 build: the three bf16 pieces (mogan_mma.h) of every filter value, laid out so that
// a lane's 8-element A fragment of one 16-k group is ONE 16-byte LDS read and the global -> LDS staging is a straight copy:
//   Wp[par][chunk][co][piece 0..2][h 0..1][group g][slot i]   (bf16)
// slot i of group g = k-step 8g + i of the chunk = (channel pair c2, tap) with step = c2*KHW + tap; lane half h takes channel
// 2*c2 + h; steps beyond the chunk's (CK/2)*KHW are zero.  w = [par][co][ci][KHW] fp32 (the conv's own filters, or the
// flipped / parity-split ones of the data gradient).  One thread per (row, chunk, h, group).
__global__ __launch_bounds__(256) void dconv_wprep_kernel(const float* __restrict__ w, uint4* __restrict__ wp, int rows, int Cin,
                                                          int KHW, int CK, int NGRP, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int g = (int)(t % NGRP); long long r = t / NGRP;
    const int h = (int)(r & 1); r >>= 1;
    const int nchunk = Cin / CK;
    const int row = (int)(r % rows); const int chunk = (int)(r / rows);       // row = par*Cout + co; rows = npar*Cout
    const int nstep = (CK / 2) * KHW;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int step = 8 * g + i;
        const int c2 = step / KHW, tap = step - c2 * KHW;
        v[i] = step < nstep ? w[((size_t)row * Cin + chunk * CK + 2 * c2 + h) * KHW + tap] : 0.f;
    }
    const X6Frag f = x6_split8(v);
    // destination: ((chunk*rows + row) * 3 + piece) * 2*NGRP + h*NGRP + g      (16-byte units)
    (void)nchunk;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        wp[(((size_t)chunk * rows + row) * 3 + pl) * (2 * NGRP) + h * NGRP + g] = __builtin_bit_cast(uint4, f.p[pl]);
}
#endif

// ------------------------------------------------------------------------------------------ forward
// CW = tile width in output pixels (32 or 16; the tile is R = 128/CW rows).  DB = LDS double buffering: the halo and
// weight images of chunk c+1 are written into the second buffer *between the MFMA k-steps of chunk c* (one barrier per
// chunk, no store phase).  PMC on the single-buffer version (96->96 3x3 at 256x256): MFMA pipe busy 76 %, every wave
// parked 13 % of its cycles at the two barriers around the store phase, and the two blocks of a CU run in lockstep.
// WIDE (stride 1, no fused upsample, W % 4 == 0): halo rows come in as aligned 16-byte quads covering the columns
// [ox0-4, ox0+CW+4), LDS column c <-> image column ox0-4+c (see dconv_wgrad_kernel).
template <int KH, int KW, int S, int WM, int WN, int TM, int TN, int CK, int CW, bool DB, bool WIDE>
__global__ __launch_bounds__(256) void dconv_fwd_kernel(const DConvP p) {
    constexpr int BM = WM * TM * 32, NTB = WN * TN, PX = NTB * 32, KHW = KH * KW, KC = CK * KHW;
    constexpr int R = PX / CW;
    constexpr int HH = (R - 1) * S + KH, WW = (CW - 1) * S + KW;
    constexpr int QPR = (CW + 8) / 4;                                // quads per halo row (WIDE)
    constexpr int WWP = WIDE ? 4 * QPR : ((WW + 7) & ~7), CPL = HH * WWP;
    constexpr int NXQ = WIDE ? (CK * HH * QPR + 255) / 256 : 1;
    static_assert(!WIDE || S == 1, "WIDE is a stride-1 variant");
    // odd row stride: the A operand read (lane = output channel, stride LDW dwords) then touches 32 distinct banks.
    // PMC with LDW = KC+4 (16-byte rows, b128 stores): 12*lane mod 32 -> 4-way conflicts on every A read, LDS array
    // busy 50 % of the kernel, 69 % of that in conflict cycles.  The price is scalar LDS stores of the weight quads.
    constexpr int LDW = KC + 1;
    constexpr int NXE = (CK * HH * WW + 255) / 256;               // halo elements per thread
    constexpr int NWQ = (BM * (KC / 4) + 255) / 256;              // weight quads per thread
    constexpr int NBUF = DB ? 2 : 1, XSZ = CK * CPL, WSZ = BM * LDW;
    constexpr int NSTEP = (CK / 2) * KHW, FIRST = NSTEP / 2;      // stores of the next chunk ride on steps >= FIRST
    constexpr int NGRP = (NSTEP + 7) / 8, GFIRST = NGRP / 2;       // groups of 8 k-steps; stores of the next chunk ride on groups >= GFIRST
#if MOGAN_X6
    // filters: bf16 pieces in fragment order (dconv_wprep_kernel), row = 3 pieces x 2 lane halves x NGRP groups x 16 B + 16 B pad
    // (an odd multiple of 16 B: the 16-byte fragment reads of 16 consecutive rows are bank-conflict free)
    constexpr int PPR = 3 * 2 * NGRP, WROW = (PPR + 1) * 16, WSZB = BM * WROW;
    constexpr int NWP = (BM * PPR + 255) / 256;                      // 16-byte pieces per thread and chunk
    constexpr int NWQ_ = NWP;
#else
    constexpr int NWQ_ = NWQ;
#endif
    constexpr int XPG = (NXE + (NGRP - GFIRST) - 1) / (NGRP - GFIRST), WPG = (NWQ_ + (NGRP - GFIRST) - 1) / (NGRP - GFIRST);
    static_assert(WM * WN == 4 && KC % 4 == 0 && CK % 2 == 0 && PX == 128, "tile");
    __shared__ __attribute__((aligned(16))) float Xs[NBUF * XSZ];
#if MOGAN_X6
    __shared__ __attribute__((aligned(16))) unsigned char Wb[NBUF * WSZB];
#else
    __shared__ __attribute__((aligned(16))) float Wl[NBUF * WSZ];
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // spatial tile
    int tile = blockIdx.x;
    const int tx = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty = tile % p.tiles_y; const int img = tile / p.tiles_y;
    const int oy0 = ty * R, ox0 = tx * CW;
    const int m0 = blockIdx.y * BM;
    const int sp = blockIdx.z % p.nsplit, par = blockIdx.z / p.nsplit;
    // stride-2 dgrad: parity class (py,px) selects its 2x2 weight set, padding and output phase
    int pt = p.pt, pl = p.pl, y0 = p.y0, x0 = p.x0;
    const float* wptr = p.Wt;
    if (p.npar == 4) {
        const int py = par >> 1, px = par & 1;
        pt = 1 - (py + 1 - ((py + 1) & 1)) / 2; pl = 1 - (px + 1 - ((px + 1) & 1)) / 2;
        y0 = py; x0 = px;
        wptr += (size_t)par * p.Cin * p.Cout * KHW;
    }
    const int wrow0 = par * p.Cout + blockIdx.y * BM;                 // first filter row of this block in the prepped layout
    const int nchunk_all = p.Cin / CK;
    const int c_beg = sp * p.cps, c_end = min(nchunk_all, c_beg + p.cps);

    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, (short)0, (int)p.x_bytes, 0x00020000);
#if MOGAN_X6
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, (short)0, (int)p.wp_bytes, 0x00020000);
    (void)wptr;
#else
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wptr, (short)0, (int)p.w_bytes, 0x00020000);
    (void)wrow0;
#endif

    // ---- per-thread staging plan (chunk independent) ----------------------------------------------------
    // halo: element e -> (c, hy, hx); global offset relative to the chunk's first channel, or OOB (zero padding)
    unsigned xg[NXE]; int xl[NXE];
    const int HsWs = p.Hs * p.Ws;
    {
        constexpr int per_c = HH * WW, total = CK * per_c;
        const int iy_base = oy0 * S - pt, ix_base = ox0 * S - pl;
#pragma unroll
        for (int i = 0; i < NXE; ++i) {
            const int e = tid + 256 * i;
            const int c = e / per_c, r = e - c * per_c;
            const int hy = r / WW, hx = r - hy * WW;
            const int iy = iy_base + hy, ix = ix_base + hx;
            const bool ok = e < total && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            xg[i] = ok ? (unsigned)(c * HsWs + (iy >> p.up) * p.Ws + (ix >> p.up)) : IDX_OOB;
            // stride 2: even and odd halo columns are stored in separate half rows, so that the 32 lanes of a B read
            // (output pixels ox, input column 2*ox+kw) touch consecutive dwords instead of every second one
            const int col = S == 2 ? (hx & 1) * (WWP / 2) + (hx >> 1) : hx;
            xl[i] = e < total ? c * CPL + hy * WWP + col : -1;
        }
    }
    unsigned qg[NXQ]; int ql[NXQ];
    if constexpr (WIDE) {
        const int iy_base = oy0 - pt;
#pragma unroll
        for (int i = 0; i < NXQ; ++i) {
            const int q = tid + 256 * i;
            const int c = q / (HH * QPR), r = q - c * (HH * QPR);
            const int hy = r / QPR, qx = r - hy * QPR;
            const int iy = iy_base + hy, ix = ox0 - 4 + 4 * qx;
            const bool ok = q < CK * HH * QPR && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            qg[i] = ok ? (unsigned)(c * HsWs + iy * p.Ws + ix) : IDX_OOB;
            ql[i] = q < CK * HH * QPR ? c * CPL + hy * WWP + 4 * qx : -1;
        }
    }
    const unsigned x_img = (unsigned)img * p.Cin * HsWs;
#if MOGAN_X6
    // filters: 16-byte piece q = (row, pc) of the chunk's BM x PPR image; global index in dwords
    const int rows_all = p.npar * p.Cout;
    unsigned wg[NWP]; int wl[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int q = tid + 256 * i;
        const int row = q / PPR, pc = q - row * PPR;
        const bool ok = q < BM * PPR && m0 + row < p.Cout;
        wg[i] = ok ? (unsigned)(((wrow0 + row) * PPR + pc) * 4) : IDX_OOB;
        wl[i] = q < BM * PPR ? row * WROW + pc * 16 : -1;
    }
    const unsigned wchunk = (unsigned)rows_all * PPR * 4u;          // dwords per chunk of the prepped filters
    float rx[NXE]; f32x4 rw[NWP]; f32x4 rxq[NXQ];
#else
    // weights: quad q -> (row, kq): 4 consecutive k of one output channel
    unsigned wg[NWQ]; int wl[NWQ];
#pragma unroll
    for (int i = 0; i < NWQ; ++i) {
        const int q = tid + 256 * i;
        const int row = q / (KC / 4), kq = q - row * (KC / 4);
        const bool ok = q < BM * (KC / 4) && m0 + row < p.Cout;
        wg[i] = ok ? (unsigned)((m0 + row) * p.Cin * KHW + 4 * kq) : IDX_OOB;
        wl[i] = q < BM * (KC / 4) ? row * LDW + 4 * kq : -1;
    }

    float rx[NXE]; f32x4 rw[NWQ]; f32x4 rxq[NXQ];
#endif
    auto load_chunk = [&](int c) {
        const unsigned xb = x_img + (unsigned)c * CK * HsWs, wb = (unsigned)c * KC;
        if constexpr (WIDE) {
#pragma unroll
            for (int i = 0; i < NXQ; ++i) rxq[i] = ldg4(rX, qg[i] == IDX_OOB ? IDX_OOB : qg[i] + xb);
        } else
#pragma unroll
        for (int i = 0; i < NXE; ++i) rx[i] = ldg(rX, xg[i] == IDX_OOB ? IDX_OOB : xg[i] + xb);
#if MOGAN_X6
#pragma unroll
        for (int i = 0; i < NWP; ++i) rw[i] = ldg4(rW, wg[i] == IDX_OOB ? IDX_OOB : wg[i] + (unsigned)c * wchunk);
        (void)wb;
#else
#pragma unroll
        for (int i = 0; i < NWQ; ++i) rw[i] = ldg4(rW, wg[i] == IDX_OOB ? IDX_OOB : wg[i] + wb);
#endif
    };
    auto store_x = [&](int i, float* Xd) {
        if constexpr (WIDE) { if (i < NXQ && ql[i] >= 0) *(f32x4*)&Xd[ql[i]] = rxq[i]; }
        else { if (xl[i] >= 0) Xd[xl[i]] = rx[i]; }
    };
#if MOGAN_X6
    auto store_w = [&](int i, unsigned char* Wd) { if (wl[i] >= 0) *(f32x4*)(Wd + wl[i]) = rw[i]; };
    typedef unsigned char wlds_t;
    wlds_t* const Wl = Wb;
    constexpr int WSZ_ = WSZB;
#else
    auto store_w = [&](int i, float* Wd) {
        if (wl[i] >= 0) {
            Wd[wl[i]] = rw[i][0]; Wd[wl[i] + 1] = rw[i][1]; Wd[wl[i] + 2] = rw[i][2]; Wd[wl[i] + 3] = rw[i][3];
        }
    };
    typedef float wlds_t;
    constexpr int WSZ_ = WSZ;
#endif
    auto store_chunk = [&](float* Xd, wlds_t* Wd) {
#pragma unroll
        for (int i = 0; i < NXE; ++i) store_x(i, Xd);
#pragma unroll
        for (int i = 0; i < NWQ_; ++i) store_w(i, Wd);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // lane bases of the MFMA operand reads
    const int h = lane >> 5;
#if MOGAN_X6
    const int abase = (wm * TM * 32 + (lane & 31)) * WROW + h * NGRP * 16;       // bytes
#else
    const int abase = (wm * TM * 32 + (lane & 31)) * LDW + h * KHW;
#endif
    int bbase[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int pb = (wn * TN + t) * 32 + (lane & 31);
        const int ry = pb / CW, rxx = pb - ry * CW;
        bbase[t] = h * CPL + ry * S * WWP + (S == 2 ? rxx : rxx * S) + (WIDE ? 4 - pl : 0);
    }

    if (c_beg < c_end) {
        load_chunk(c_beg);
        store_chunk(Xs, Wl);
        __syncthreads();
        int cur = 0;
        for (int c = c_beg; c < c_end; ++c) {
            const bool more = c + 1 < c_end;
            if constexpr (!DB) load_chunk(c + 1);   // branch-free body (one basic block): a chunk past the end reads harmless data
            else if (more) load_chunk(c + 1);
            const float* Xc = Xs + cur * XSZ;
            const wlds_t* Wc = Wl + cur * WSZ_;
            float* Xn = Xs + (cur ^ (NBUF - 1)) * XSZ;
            wlds_t* Wn = Wl + (cur ^ (NBUF - 1)) * WSZ_;
            // k-steps (channel pair c2, tap kh, kw) in groups of eight = one mma_k16 (mogan_mma.h); a short last group is
            // padded with zeros
#pragma unroll
            for (int g = 0; g < NGRP; ++g) {
#if MOGAN_X6
                // A: the pre-split filter fragments, one 16-byte read per piece; B: eight halo values, split in registers
                X6Frag fa[TM], fb[TN];
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        fa[t].p[pl] = __builtin_bit_cast(mma_bf16x8, *(const uint4*)(Wc + abase + t * 32 * WROW + (pl * 2 * NGRP + g) * 16));
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    float b8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int step = 8 * g + i;                         // compile-time after unrolling
                        const int c2 = step / KHW, tap = step % KHW, kh = tap / KW, kw = tap % KW;
                        b8[i] = step < NSTEP
                            ? Xc[bbase[t] + 2 * c2 * CPL + kh * WWP + (S == 2 ? (kw & 1) * (WWP / 2) + (kw >> 1) : kw)] : 0.f;
                    }
                    fb[t] = x6_split8(b8);
                }
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                        for (int tb = 0; tb < TN; ++tb) acc[ta][tb] = x6_mfma(fa[ta], fb[tb], term, acc[ta][tb]);
#else
                float a[TM][8], b[TN][8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int step = 8 * g + i;                             // compile-time after unrolling
                    const int c2 = step / KHW, tap = step % KHW, kh = tap / KW, kw = tap % KW;
#pragma unroll
                    for (int t = 0; t < TM; ++t) a[t][i] = step < NSTEP ? Wc[abase + t * 32 * LDW + 2 * c2 * KHW + kh * KW + kw] : 0.f;
#pragma unroll
                    for (int t = 0; t < TN; ++t)
                        b[t][i] = step < NSTEP
                            ? Xc[bbase[t] + 2 * c2 * CPL + kh * WWP + (S == 2 ? (kw & 1) * (WWP / 2) + (kw >> 1) : kw)] : 0.f;
                }
                mma_k16<TM, TN>(a, b, acc);
#endif
                if constexpr (DB) {
                    if (g >= GFIRST && more) {
                        const int s0 = g - GFIRST;
#pragma unroll
                        for (int j = 0; j < XPG; ++j)
                            if (s0 * XPG + j < NXE) store_x(s0 * XPG + j, Xn);
#pragma unroll
                        for (int j = 0; j < WPG; ++j)
                            if (s0 * WPG + j < NWQ_) store_w(s0 * WPG + j, Wn);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (!DB) {
                // issue-order template (see dconv_wgrad_kernel): one global load of the next chunk behind each of the
                // first k-steps instead of a load phase in front of the MFMA loop.  +2..4 % over the double-buffered
                // variant on the 3x3 / 2x2 / 4x4-s2 layers (e.g. dgrad 96->192 at 128x128: 134 vs 129 TFLOP/s).
#if !MOGAN_X6
#pragma unroll
                for (int g = 0; g < NSTEP; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
                    if (g < (WIDE ? NXQ : NXE) + NWQ) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#endif
            }
            __syncthreads();
            if constexpr (DB) {
                cur ^= 1;
            } else {
                if (more) { store_chunk(Xs, Wl); __syncthreads(); }
            }
        }
    }

    // ---- epilogue: Y[img][co][oy*ys + y0][ox*ys + x0] -------------------------------------------------------
    float* __restrict__ Yg = (p.nsplit > 1) ? (p.ws + (size_t)sp * p.slab) : p.Y;
    const bool addc = (p.nsplit == 1) && p.accumulate;
    const size_t plane = (size_t)p.yH * p.yW;
#pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
        const int pb = (wn * TN + tb) * 32 + (lane & 31);
        const int ry = pb / CW, rxx = pb - ry * CW;
        const size_t pix = (size_t)((oy0 + ry) * p.ys + y0) * p.yW + (ox0 + rxx) * p.ys + x0;
#pragma unroll
        for (int ta = 0; ta < TM; ++ta)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.Cout) {
                    float* dst = Yg + ((size_t)img * p.Cout + m) * plane + pix;
                    float v = acc[ta][tb][r];
                    if (addc) v += *dst;
                    *dst = v;
                }
            }
    }
}


// ------------------------------------------------------------------------------------------ weight gradient
// dW[co][(ci,kh,kw)] = sum over pixels dY[co][pix] * X[ci][pix*S - p + (kh,kw)]:  M = co, N = (ci,kh,kw), K = pixels.
// K-chunk = a tile of 64 output pixels (RT x CW) of one image: Ys[BM][64] (dY, 16-byte loads) and the halo tile
// Xs[CKW][HHW][WWP] of the <= CKW input channels this block's 128 columns touch.  Lanes carry the column
// (ci,kh,kw) -> per-lane LDS base; the pixel pair of each k-step is a compile-time offset.
struct WGradP {
    const float* dY; const float* X; float* dW; float* ws;
    int B, Cin, Cout, Hs, Ws, H, W, up, OH, OW, pt, pl;
    int N;                    // Cin*KH*KW
    int tiles_x, tiles_y, ntiles, tps, nsplit;     // pixel tiles, tiles per split
    int lg_tx, lg_ty;                              // log2 of tiles_x / tiles_y (both are powers of two)
    long long slab; int accumulate;
    unsigned x_bytes, y_bytes;
};

// DBW = LDS double buffering: the dY / halo images of pixel-tile t+1 are written into the second buffer between the
// MFMA k-steps of tile t (one barrier per tile).  Lab numbers that motivated it (96->96 3x3 at 256x256, TFLOP/s):
// product 87, without the X gather 104, without any global load 113, without the store phase and its two barriers 115.
// WIDE (3x3 s1 p1, no fused upsample, W % 4 == 0): the halo rows are fetched as aligned 16-byte quads covering the
// columns [ox0-4, ox0+CW+4) -- 10 loads per row instead of 34 dword loads; every quad is entirely inside or outside
// the image.  LDS column c of a row then holds image column ox0-4+c.
#ifndef DCONV_WG_XCD
#define DCONV_WG_XCD 1
#endif
#ifndef DCONV_WG_OCC
#define DCONV_WG_OCC 2
#endif
// (two blocks per CU asked for: left alone, the allocator takes 264 registers for the 2 x 2-tile instantiations -- ONE wave per
// SIMD -- where 256 are enough without a spill; these kernels wait on loads and LDS round trips, more waves is what they lack)
template <int KH, int KW, int S, int CW, int WM, int WN, int TM, int TN, bool DBW, bool WIDE>
__global__ __launch_bounds__(256, DCONV_WG_OCC) void dconv_wgrad_kernel(const WGradP p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, KHW = KH * KW, PXK = 64, RT = PXK / CW;
    constexpr int QPR = (CW + 8) / 4;                                      // quads per halo row (WIDE)
    constexpr int HHW = (RT - 1) * S + KH, WW = (CW - 1) * S + KW, WWP = WIDE ? 4 * QPR + 4 : ((WW + 3) & ~3);
    constexpr int CPLW = HHW * WWP, COL0 = WIDE ? 3 : 0;                   // LDS column of image column ox0 - pl
    constexpr int CKW = BN / KHW + 2, LDY = PXK + 4;
    constexpr int NXE = WIDE ? 1 : (CKW * HHW * WW + 255) / 256, NYQ = BM * (PXK / 4) / 256;
    constexpr int NXQ = WIDE ? (CKW * HHW * QPR + 255) / 256 : 1;
    static_assert(!WIDE || (S == 1 && KH == 3 && KW == 3), "WIDE is the 3x3 s1 variant");
    constexpr bool PIPE = (S == 2);
    constexpr int NBUF = DBW ? 2 : 1, YSZ = BM * LDY, XSZ = CKW * CPLW;
    constexpr int NSTEP = PXK / 2, FIRST = NSTEP / 2;
    constexpr int NGRP = NSTEP / 8, GFIRST = NGRP / 2;                     // groups of 8 k-steps (one mma_k16 each)
    constexpr int XPG = (NXE + (NGRP - GFIRST) - 1) / (NGRP - GFIRST), YPG = (NYQ + (NGRP - GFIRST) - 1) / (NGRP - GFIRST);
    static_assert(WM * WN == 4 && BN == 128 && NSTEP % 8 == 0, "tile");
#if MOGAN_X6
    // dY tile as bf16 pieces in fragment order: the split happens once per staged value (instead of once per fragment value,
    // TM x more), an A fragment of a 16-pixel group is one 16-byte read per piece.  Row = [piece][lane half h][group g][8 x bf16]
    // + 16 B pad = 400 B; pixel p of the tile sits at g = p / 16, h = p & 1, slot (p % 16) / 2.
    constexpr int YROW = 3 * 2 * (PXK / 16) * 16 + 16, YSZB = BM * YROW;
    static_assert(PXK == 64, "fragment-order layout of the dY tile");
    __shared__ __attribute__((aligned(16))) unsigned char Ys[NBUF * YSZB];
    typedef unsigned char ylds_t;
    constexpr int YSZ_ = YSZB;
#else
    __shared__ __attribute__((aligned(16))) float Ys[NBUF * YSZ];
    typedef float ylds_t;
    constexpr int YSZ_ = YSZ;
#endif
    __shared__ __attribute__((aligned(16))) float Xs[NBUF * XSZ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // (the blocks of one pixel-tile range -- all (ci, co) blocks read the same dY / x tiles -- on one XCD: mma_xcd_block)
#if DCONV_WG_XCD
    unsigned bx_, by_, bz_; mma_xcd_block(bx_, by_, bz_);
#else
    const unsigned bx_ = blockIdx.x, by_ = blockIdx.y, bz_ = blockIdx.z;
#endif
    const int n0 = bx_ * BN, m0 = by_ * BM, sp = bz_;
    const int ci_first = n0 / KHW;
    const int t_beg = sp * p.tps, t_end = min(p.ntiles, t_beg + p.tps);
    const int HsWs = p.Hs * p.Ws;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, (short)0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)p.dY, (short)0, (int)p.y_bytes, 0x00020000);

    // staging plans.  Halo element e -> (c, hy, hx).  Its global index is pre[i] + (a per-tile scalar): the part that
    // depends on the element is computed once here, so a tile costs one add per element (interior tiles) or an add, two
    // compares and a select (tiles that touch the image border) instead of the full (iy >> up) * Ws + (ix >> up) chain.
    // With the fused upsample (3x3 p1 only: tile origins are even, iyb/ixb odd) (iyb+hy)>>1 = (iyb+1)/2 + ((hy-1)>>1).
    constexpr unsigned PRE_BAD = 0x20000000u;       // * 4 bytes >= 2 GiB > any extent: the buffer load returns 0
    unsigned pre[NXE]; int xhy[NXE], xhx[NXE], xl[NXE];
#pragma unroll
    for (int i = 0; i < NXE; ++i) {
        const int e = tid + 256 * i;
        const int c = e / (HHW * WW), r = e - c * (HHW * WW);
        const int hy = r / WW, hx = r - hy * WW;
        const bool ok = e < CKW * HHW * WW && ci_first + c < p.Cin;
        const int inoff = p.up ? ((hy - 1) >> 1) * p.Ws + ((hx - 1) >> 1) : hy * p.Ws + hx;
        pre[i] = ok ? (unsigned)((ci_first + c) * HsWs + inoff) : PRE_BAD;
        xhy[i] = hy; xhx[i] = hx;
        xl[i] = e < CKW * HHW * WW ? c * CPLW + hy * WWP + hx : -1;
    }
    unsigned qpre[NXQ]; int qhy[NXQ], qx4[NXQ], ql[NXQ];
    if constexpr (WIDE) {
#pragma unroll
        for (int i = 0; i < NXQ; ++i) {
            const int q = tid + 256 * i;
            const int c = q / (HHW * QPR), r = q - c * (HHW * QPR);
            const int hy = r / QPR, qx = r - hy * QPR;
            const bool ok = q < CKW * HHW * QPR && ci_first + c < p.Cin;
            qpre[i] = ok ? (unsigned)((ci_first + c) * HsWs + hy * p.Ws + 4 * qx) : PRE_BAD;
            qhy[i] = hy; qx4[i] = 4 * qx;
            ql[i] = q < CKW * HHW * QPR ? c * CPLW + hy * WWP + 4 * qx : -1;
        }
    }
    unsigned yg[NYQ]; int yl[NYQ];
#pragma unroll
    for (int i = 0; i < NYQ; ++i) {
        const int q = tid + 256 * i;
        const int row = q / (PXK / 4), p0 = 4 * (q - row * (PXK / 4));
        const int ry = p0 / CW, rxx = p0 - ry * CW;
        yg[i] = m0 + row < p.Cout ? (unsigned)((m0 + row) * p.OH * p.OW + ry * p.OW + rxx) : IDX_OOB;
#if MOGAN_X6
        yl[i] = row * YROW + (p0 >> 4) * 16 + ((p0 & 15) >> 1) * 2;      // bytes: group p0/16, slot (p0 % 16) / 2 (even), piece 0, h = 0
#else
        yl[i] = row * LDY + p0;
#endif
    }
    float rx[NXE]; f32x4 ry4[NYQ]; f32x4 rxq[NXQ];
    auto load_tile = [&](int t) {
        const int tx = t & (p.tiles_x - 1); const int u = t >> p.lg_tx;
        const int ty = u & (p.tiles_y - 1); const int img = u >> p.lg_ty;
        const int oy0 = ty * RT, ox0 = tx * CW;
        const int iyb = oy0 * S - p.pt, ixb = ox0 * S - p.pl;
        const unsigned yb = (unsigned)img * p.Cout * p.OH * p.OW + oy0 * p.OW + ox0;
        // block-uniform part of the halo index (may be "negative" for border tiles: unsigned wrap-around is intended,
        // such elements are masked below)
        const unsigned sbase = (unsigned)img * p.Cin * HsWs +
            (unsigned)(p.up ? ((iyb + 1) >> 1) * p.Ws + ((ixb + 1) >> 1) : iyb * p.Ws + ixb);
        const bool interior = iyb >= 0 && ixb >= 0 && iyb + HHW <= p.H && ixb + WW <= p.W;
        if constexpr (WIDE) {
            // quad columns start at image column ox0 - 4 (16-byte aligned); whole quads are in or out of the row
            const unsigned qbase = (unsigned)img * p.Cin * HsWs + (unsigned)(iyb * p.Ws + ox0 - 4);
#pragma unroll
            for (int i = 0; i < NXQ; ++i) {
                const bool ok = (unsigned)(iyb + qhy[i]) < (unsigned)p.H && (unsigned)(ox0 - 4 + qx4[i]) < (unsigned)p.W;
                rxq[i] = ldg4(rX, ok ? qpre[i] + qbase : PRE_BAD);
            }
        } else
        if (!PIPE && interior) {
#pragma unroll
            for (int i = 0; i < NXE; ++i) rx[i] = ldg(rX, pre[i] + sbase);
        } else {
#pragma unroll
            for (int i = 0; i < NXE; ++i) {
                const bool ok = (unsigned)(iyb + xhy[i]) < (unsigned)p.H && (unsigned)(ixb + xhx[i]) < (unsigned)p.W;
                rx[i] = ldg(rX, ok ? pre[i] + sbase : PRE_BAD);
            }
        }
#pragma unroll
        for (int i = 0; i < NYQ; ++i) ry4[i] = ldg4(rY, yg[i] == IDX_OOB ? IDX_OOB : yb + yg[i]);
    };
    auto store_x = [&](int i, float* Xd) { if (xl[i] >= 0) Xd[xl[i]] = rx[i]; };
#if MOGAN_X6
    auto store_y = [&](int i, ylds_t* Yd) {           // pixels p0, p0+2 -> lane half 0, p0+1, p0+3 -> lane half 1
        uint32_t w[2][3];
        x6_split2(ry4[i][0], ry4[i][2], w[0][0], w[0][1], w[0][2]);
        x6_split2(ry4[i][1], ry4[i][3], w[1][0], w[1][1], w[1][2]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) *(uint32_t*)(Yd + yl[i] + (pl * 2 + hh) * 64) = w[hh][pl];
    };
#else
    auto store_y = [&](int i, ylds_t* Yd) { *(f32x4*)&Yd[yl[i]] = ry4[i]; };
#endif
    auto store_tile = [&](float* Xd, ylds_t* Yd) {
        if constexpr (WIDE) {
#pragma unroll
            for (int i = 0; i < NXQ; ++i) if (ql[i] >= 0) *(f32x4*)&Xd[ql[i]] = rxq[i];
        } else
#pragma unroll
        for (int i = 0; i < NXE; ++i) store_x(i, Xd);
#pragma unroll
        for (int i = 0; i < NYQ; ++i) store_y(i, Yd);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#if MOGAN_X6
    const int abase = (wm * TM * 32 + (lane & 31)) * YROW + h * 64;          // bytes
#else
    const int abase = (wm * TM * 32 + (lane & 31)) * LDY + h;
#endif
    int bbase[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int n = n0 + (wn * TN + t) * 32 + (lane & 31);
        const int ci = n / KHW, tap = n - ci * KHW;
        const int kh = tap / KW, kw = tap - kh * KW;
        bbase[t] = n < p.N ? (ci - ci_first) * CPLW + kh * WWP + kw + COL0 + h * S : 0;
    }

    if (t_beg < t_end) {
        load_tile(t_beg);
        store_tile(Xs, Ys);
        __syncthreads();
        int cur = 0;
        for (int t = t_beg; t < t_end; ++t) {
            const bool more = t + 1 < t_end;
            if constexpr (PIPE) load_tile(t + 1 < p.ntiles ? t + 1 : t);      // branch-free body (one basic block)
            else if (more) load_tile(t + 1);
            const float* Xc = Xs + cur * XSZ;
            const ylds_t* Yc = Ys + cur * YSZ_;
            float* Xn = Xs + (cur ^ (NBUF - 1)) * XSZ;
            ylds_t* Yn = Ys + (cur ^ (NBUF - 1)) * YSZ_;
#pragma unroll
            for (int g = 0; g < NSTEP / 8; ++g) {
#if MOGAN_X6
                X6Frag fa[TM], fb[TN];
#pragma unroll
                for (int q = 0; q < TM; ++q)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        fa[q].p[pl] = __builtin_bit_cast(mma_bf16x8, *(const uint4*)(Yc + abase + q * 32 * YROW + pl * 128 + g * 16));
#pragma unroll
                for (int q = 0; q < TN; ++q) {
                    float b8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int pp = 8 * g + i;                           // pixel pair: compile-time after unrolling
                        const int ry = (2 * pp) / CW, rxx = (2 * pp) % CW;
                        b8[i] = Xc[bbase[q] + ry * S * WWP + rxx * S];
                    }
                    fb[q] = x6_split8(b8);
                }
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                        for (int tb = 0; tb < TN; ++tb) acc[ta][tb] = x6_mfma(fa[ta], fb[tb], term, acc[ta][tb]);
#else
                float a[TM][8], b[TN][8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int pp = 8 * g + i;                               // pixel pair: compile-time after unrolling
                    const int ry = (2 * pp) / CW, rxx = (2 * pp) % CW;
#pragma unroll
                    for (int q = 0; q < TM; ++q) a[q][i] = Yc[abase + q * 32 * LDY + 2 * pp];
#pragma unroll
                    for (int q = 0; q < TN; ++q) b[q][i] = Xc[bbase[q] + ry * S * WWP + rxx * S];
                }
                mma_k16<TM, TN>(a, b, acc);
#endif
                if constexpr (DBW) {
                    if (g >= GFIRST && more) {
                        const int s0 = g - GFIRST;
#pragma unroll
                        for (int j = 0; j < XPG; ++j)
                            if (s0 * XPG + j < NXE) store_x(s0 * XPG + j, Xn);
#pragma unroll
                        for (int j = 0; j < YPG; ++j)
                            if (s0 * YPG + j < NYQ) store_y(s0 * YPG + j, Yn);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (PIPE) {
                // issue-order template for the scheduler: per k-step its LDS operand reads and MFMAs, and one of the
                // NXE+NYQ global loads of the next tile behind each of the first k-steps (instead of all of them in
                // front of the MFMA loop, where both blocks of a CU sit in their load phase at the same time).
                // Measured on the 4x4 s2 layers: 83 -> 97 TFLOP/s; no gain on the 3x3 variant (fewer halo loads).
#if !MOGAN_X6
#pragma unroll
                for (int g = 0; g < PXK / 2; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
                    if (g < NXE + NYQ) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#endif
            }
            __syncthreads();
            if constexpr (DBW) {
                cur ^= 1;
            } else {
                if (more) { store_tile(Xs, Ys); __syncthreads(); }
            }
        }
    }

    float* __restrict__ Wg = (p.nsplit > 1) ? (p.ws + (size_t)sp * p.slab) : p.dW;
    const bool addc = (p.nsplit == 1) && p.accumulate;
#pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
        const int n = n0 + (wn * TN + tb) * 32 + (lane & 31);
#pragma unroll
        for (int ta = 0; ta < TM; ++ta)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < p.N && m < p.Cout) {
                    float* dst = Wg + (size_t)m * p.N + n;
                    float v = acc[ta][tb][r];
                    if (addc) v += *dst;
                    *dst = v;
                }
            }
    }
}

// weight transforms for the data gradient ---------------------------------------------------------------------
// s1: Wd[ci][co][kh][kw] = W[co][ci][KH-1-kh][KW-1-kw]
__global__ __launch_bounds__(256) void wflip_kernel(const float* __restrict__ w, float* __restrict__ wd, int Cout, int Cin,
                                                    int KH, int KW) {
    const long long total = (long long)Cout * Cin * KH * KW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int kw = (int)(i % KW); long long t = i / KW;
        const int kh = (int)(t % KH); t /= KH;
        const int co = (int)(t % Cout); const int ci = (int)(t / Cout);
        wd[i] = w[(((size_t)co * Cin + ci) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
    }
}
// s2 (4x4, p1): four parity kernels Wp[par][ci][co][a][b] = W[co][ci][kh0 + 2(1-a)][kw0 + 2(1-b)], kh0 = (py+1)&1
__global__ __launch_bounds__(256) void wparity_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin) {
    const long long per = (long long)Cin * Cout * 4, total = per * 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int par = (int)(i / per); long long r = i - par * per;
        const int b = (int)(r & 1), a = (int)((r >> 1) & 1); r >>= 2;
        const int co = (int)(r % Cout), ci = (int)(r / Cout);
        const int py = par >> 1, px = par & 1;
        const int kh = ((py + 1) & 1) + 2 * (1 - a), kw = ((px + 1) & 1) + 2 * (1 - b);
        wp[i] = w[(((size_t)co * Cin + ci) * 4 + kh) * 4 + kw];
    }
}

__global__ __launch_bounds__(256) void dconv_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                           long long n, long long slab, int nsplit, int acc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // fixed summation order (deterministic)
    if (i >= n) return;
    const float* p = ws + i;
    float s = acc ? out[i] : 0.f;
    int k = 0;
    for (; k + 8 <= nsplit; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * slab];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nsplit; ++k) s += p[(size_t)k * slab];
    out[i] = s;
}

static inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }
static inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

template <int KH, int KW, int S, int CK>
static int launch_fwd(DConvP& p, void* ws, size_t ws_bytes, hipStream_t st) {
    if constexpr (S == 1) {           // round 5: the pre-split form (mogan_dconv2.hip) where its tile grid fits
        if (p.up == 0) {
            const int rc2 = mogan_dconv2_fwd_try(p.X, p.Wt, 0, p.Y, p.B, p.Cin, p.Cout, p.H, p.W, p.OH, p.OW, KH, KW, p.pt, p.pl, p.yH,
                                                 p.yW, p.ys, p.npar, p.accumulate, ws, ws_bytes, st, nullptr, p.d2query);
            if (rc2 != 0) return rc2 < 0 ? rc2 : 0;
        }
    }
    if constexpr (S == 2 && KH == 4 && KW == 4) {     // 4x4 s2 p1: the same kernel over the space-to-depth image of the input
        if (p.up == 0 && p.pt == 1 && p.pl == 1 && p.npar == 1 && p.ys == 1 && !p.accumulate) {
            const int rc2 = mogan_dconv2_fwd_try(p.X, p.Wt, 3, p.Y, p.B, p.Cin, p.Cout, p.H, p.W, p.OH, p.OW, 4, 4, 1, 1, p.yH, p.yW, 1, 1,
                                                 0, ws, ws_bytes, st, p.d2prep, p.d2query);
            if (rc2 != 0) return rc2 < 0 ? rc2 : 0;
        }
    }
    if (p.d2query) return MOGAN_ERR_WS;                  // dry run of the dispatch (mogan_conv_prep_bytes): nothing is launched
    // tile config: 96-wide M when it pads less
    const bool m96 = cdiv(p.Cout, 96) * 96 < cdiv(p.Cout, 128) * 128;
    const int bm = m96 ? 96 : 128;
    p.Cw = std::min(32, p.OW); p.R = 128 / p.Cw;
    p.tiles_x = p.OW / p.Cw; p.tiles_y = p.OH / p.R;
    const long long tiles = (long long)p.B * p.tiles_x * p.tiles_y * cdiv(p.Cout, bm) * p.npar;
    const int nchunk = p.Cin / CK;
    int nsplit = 1;
    constexpr int fwd_target = 512;
    if (tiles < 384 && nchunk >= 8) nsplit = (int)std::min<long long>(cdiv(fwd_target, tiles), nchunk / 4);
    const long long y_numel = (long long)p.B * p.Cout * p.yH * p.yW;
#if MOGAN_X6
    {   // the filters as bf16 pieces in fragment order, at the head of the workspace (dconv_wprep_kernel)
        constexpr int NGRP = ((CK / 2) * KH * KW + 7) / 8, PPR = 6 * NGRP;
        const long long rows = (long long)p.npar * p.Cout;
        const size_t wpb = (size_t)nchunk * rows * PPR * 16;
        if (!ws || ws_bytes < wpb + 256 || wpb >= (1ull << 31)) return MOGAN_ERR_WS;
        const long long total = (long long)nchunk * rows * 2 * NGRP;
        hipLaunchKernelGGL(dconv_wprep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p.Wt, (uint4*)ws, (int)rows,
                           p.Cin, KH * KW, CK, NGRP, total);
        p.Wp = ws; p.wp_bytes = (unsigned)wpb;
        const size_t adv = (wpb + 255) & ~(size_t)255;
        ws = (char*)ws + adv; ws_bytes -= adv;
    }
#endif
    if (nsplit > 1) {
        const long long fit = ws ? (long long)(ws_bytes / (sizeof(float) * (size_t)y_numel)) : 0;
        nsplit = fit < 2 ? 1 : (int)std::min<long long>(nsplit, fit);
    }
    p.cps = (int)cdiv(nchunk, nsplit); p.nsplit = (int)cdiv(nchunk, p.cps);
    p.slab = y_numel; p.ws = (float*)ws;
    dim3 grid((unsigned)(p.B * p.tiles_x * p.tiles_y), (unsigned)cdiv(p.Cout, bm), (unsigned)(p.nsplit * p.npar));
    // 96-wide tiles double-buffer their LDS images (2 x 36 KB, two blocks per CU still fit in 160 KB); the 128-wide
    // ones (2 x 46 KB would leave one block per CU) keep the single buffer
    constexpr bool DB1 = false;     // single LDS buffer + the issue-order template; DB = true keeps the double-buffered loop
    const bool wide = S == 1 && p.up == 0 && (p.W % 4) == 0 && (((uintptr_t)p.X) & 15) == 0;
#define MOGAN_FW(WMv, WNv, TMv, TNv, CWv, DBv, WIDEv) \
    hipLaunchKernelGGL((dconv_fwd_kernel<KH, KW, S, WMv, WNv, TMv, TNv, CK, CWv, DBv, WIDEv>), grid, dim3(256), 0, st, p)
    bool done = false;
    if constexpr (S == 1) {
        if (wide) {
            if (p.Cw == 32) { if (m96) MOGAN_FW(1, 4, 3, 1, 32, DB1, true); else MOGAN_FW(2, 2, 2, 2, 32, false, true); }
            else { if (m96) MOGAN_FW(1, 4, 3, 1, 16, DB1, true); else MOGAN_FW(2, 2, 2, 2, 16, false, true); }
            done = true;
        }
    }
    if (!done) {
        if (p.Cw == 32) { if (m96) MOGAN_FW(1, 4, 3, 1, 32, DB1, false); else MOGAN_FW(2, 2, 2, 2, 32, false, false); }
        else { if (m96) MOGAN_FW(1, 4, 3, 1, 16, DB1, false); else MOGAN_FW(2, 2, 2, 2, 16, false, false); }
    }
#undef MOGAN_FW
    if (p.nsplit > 1)
        hipLaunchKernelGGL(dconv_reduce_kernel, dim3((unsigned)cdiv(y_numel, 256)), dim3(256), 0, st,
                           (const float*)ws, p.Y, y_numel, y_numel, p.nsplit, p.accumulate);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

template <int KH, int KW, int S>
static int launch_wgrad(WGradP& p, void* ws, size_t ws_bytes, hipStream_t st) {
    const bool m96 = cdiv(p.Cout, 96) * 96 < cdiv(p.Cout, 128) * 128;
    const int bm = m96 ? 96 : 128;
    const int cw = std::min(32, p.OW), rt = 64 / cw;
    p.tiles_x = p.OW / cw; p.tiles_y = p.OH / rt; p.ntiles = p.B * p.tiles_x * p.tiles_y;
    p.lg_tx = __builtin_ctz(p.tiles_x); p.lg_ty = __builtin_ctz(p.tiles_y);
    const long long blocks = cdiv(p.N, 128) * cdiv(p.Cout, bm);
    const long long w_numel = (long long)p.Cout * p.N;
    // block target of the pixel-tile split.  640 was the isolated optimum; in the step these launches run on the weight-gradient side
    // stream beside the generator's data-gradient chain, and fewer, longer blocks leave it more of the chip: 384 -> 402.1 / 400.3 /
    // 409.8 against 398.0 / 398.8 / 407.3 img/s (448: 401.2 / 401.1, 320: 398.2 / 400.1, 256: 394.6 / 395.1; tools/split_probe.sh)
    constexpr int wg_target = 384;
    int nsplit = (int)std::min<long long>(cdiv(wg_target, blocks), std::max(1, p.ntiles / 2));
    if (nsplit > 1) {
        const long long fit = ws ? (long long)(ws_bytes / (sizeof(float) * (size_t)w_numel)) : 0;
        nsplit = fit < 2 ? 1 : (int)std::min<long long>(nsplit, fit);
    }
    p.tps = (int)cdiv(p.ntiles, nsplit); p.nsplit = (int)cdiv(p.ntiles, p.tps);
    p.slab = w_numel; p.ws = (float*)ws;
    dim3 grid((unsigned)cdiv(p.N, 128), (unsigned)cdiv(p.Cout, bm), (unsigned)p.nsplit);
    // LDS double buffering (DBW) is implemented but off: measured 65 TF with the stores pinned between the k-steps
    // (the sched_barriers stop the compiler from running the LDS operand reads ahead of the MFMAs) and 87 = no gain
    // without the pinning; the lab variant without global loads runs at 113, so the loads, not the barriers, cost
    constexpr bool DBW = false;
    // 16-byte halo loads for the plain 3x3 s1 p1 convolutions (the ResBlock / D 3x3 layers)
    const bool wide = S == 1 && KH == 3 && p.up == 0 && p.pl == 1 && p.pt == 1 && (p.W % 4) == 0 &&
                      (((uintptr_t)p.X) & 15) == 0;
#define MOGAN_WG(CWv, WMv, WNv, TMv, TNv, DBv, WIDEv) \
    hipLaunchKernelGGL((dconv_wgrad_kernel<KH, KW, S, CWv, WMv, WNv, TMv, TNv, DBv, WIDEv>), grid, dim3(256), 0, st, p)
    if constexpr (S == 1 && KH == 3) {
        if (wide) {
            if (cw == 32) { if (m96) MOGAN_WG(32, 1, 4, 3, 1, false, true); else MOGAN_WG(32, 2, 2, 2, 2, false, true); }
            else { if (m96) MOGAN_WG(16, 1, 4, 3, 1, false, true); else MOGAN_WG(16, 2, 2, 2, 2, false, true); }
        }
    }
    if (!(S == 1 && KH == 3 && wide)) {
        if (cw == 32) { if (m96) MOGAN_WG(32, 1, 4, 3, 1, DBW, false); else MOGAN_WG(32, 2, 2, 2, 2, false, false); }
        else { if (m96) MOGAN_WG(16, 1, 4, 3, 1, DBW, false); else MOGAN_WG(16, 2, 2, 2, 2, false, false); }
    }
#undef MOGAN_WG
    if (p.nsplit > 1)
        hipLaunchKernelGGL(dconv_reduce_kernel, dim3((unsigned)cdiv(w_numel, 256)), dim3(256), 0, st,
                           (const float*)ws, p.dW, w_numel, w_numel, p.nsplit, p.accumulate);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

}  // namespace

// ---- internal entry points (hidden visibility: not part of the C ABI) ---------------------------------------
// return 1 = handled, 0 = not eligible (caller falls back to the implicit-GEMM kernel), <0 = error
int mogan_dconv_fwd_try(const float* x, const float* w, float* y, int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW,
                        int stride, int ph, int pw, int up, void* ws, size_t ws_bytes, hipStream_t st, const void* d2prep,
                        size_t* d2query) {
    const int H = Hs << up, W = Ws << up;
    const int OH = (H + 2 * ph - KH) / stride + 1, OW = (W + 2 * pw - KW) / stride + 1;
    const bool k33 = KH == 3 && KW == 3 && stride == 1 && ph == 1 && pw == 1;
    const bool k44 = KH == 4 && KW == 4 && stride == 2 && ph == 1 && pw == 1 && up == 0;
    if (!(k33 || k44)) return 0;
    if (!pow2(OH) || !pow2(OW) || OW < 16 || OH < 8 || OH * std::min(32, OW) < 128) return 0;
    if ((Cin % 8) != 0 || Cout < 64 || (((uintptr_t)w) & 15) != 0) return 0;
    if ((long long)B * Cin * Hs * Ws >= (1ll << 29) || (long long)Cout * Cin * KH * KW >= (1ll << 29) ||
        (long long)B * Cout * OH * OW >= (1ll << 30)) return 0;
    DConvP p{};
    p.X = x; p.Wt = w; p.Y = y; p.B = B; p.Cin = Cin; p.Cout = Cout; p.Hs = Hs; p.Ws = Ws; p.H = H; p.W = W; p.up = up;
    p.OH = OH; p.OW = OW; p.pt = ph; p.pl = pw; p.yH = OH; p.yW = OW; p.ys = 1; p.y0 = 0; p.x0 = 0; p.accumulate = 0; p.npar = 1;
    p.x_bytes = 4u * B * Cin * Hs * Ws; p.w_bytes = 4u * Cout * Cin * KH * KW;
    p.d2prep = d2prep; p.d2query = d2query;
    // 4x4 s2: round 1 (native fp32 MFMA) took it only at >= 64-pixel rows (124 vs 98 TF against the implicit GEMM at 64x64 output,
    // 98 vs 98 at 32x32, 90 vs 98 at 16x16: more M-blocks re-reading the same halo tile); with the pre-split filters of the
    // split-bf16 build it wins down to 16-pixel rows in the step (339.2 / 337.1 vs 337.8 / 335.6 img/s; MOGAN_DCONV_K44_MINOW)
    constexpr int k44_min_ow = 16;
    if (k44 && OW < k44_min_ow) return 0;
    // 4x4 s2: chunks of 4 channels (32 k-steps, like the 36 of a 3x3 chunk of 8) keep the staging registers and the
    // double-buffered LDS images (2 x 36 KB) within two blocks per CU
    const int rc = k44 ? launch_fwd<4, 4, 2, 4>(p, ws, ws_bytes, st) : launch_fwd<3, 3, 1, 8>(p, ws, ws_bytes, st);
    if (rc == MOGAN_ERR_WS) return 0;       // no room for the prepared filters: the implicit-GEMM path takes it
    return rc ? rc : 1;
}

// data gradient of the same two conv families; dx is (B,Cin,H,W) in the conv-input domain
int mogan_dconv_dgrad_try(const float* dy, const float* w, float* dx, int B, int Cin, int Hs, int Ws, int Cout, int KH,
                          int KW, int stride, int ph, int pw, int up, void* ws, size_t ws_bytes, hipStream_t st, const void* d2prep,
                          size_t* d2query) {
    const int H = Hs << up, W = Ws << up;
    const int OH = (H + 2 * ph - KH) / stride + 1, OW = (W + 2 * pw - KW) / stride + 1;
    const bool k33 = KH == 3 && KW == 3 && stride == 1 && ph == 1 && pw == 1;
    const bool k44 = KH == 4 && KW == 4 && stride == 2 && ph == 1 && pw == 1 && up == 0;
    if (!(k33 || k44)) return 0;
    // the dgrad "forward" runs over dY (OH x OW) and produces H x W (k33) or the (H/2 x W/2) parity grids (k44)
    const int gH = k33 ? H : H / 2, gW = k33 ? W : W / 2;
    if (!pow2(gH) || !pow2(gW) || gW < 16 || gH < 8 || gH * std::min(32, gW) < 128) return 0;
    if ((Cout % 8) != 0 || Cin < 64) return 0;
    const size_t wbytes = (size_t)Cout * Cin * KH * KW * sizeof(float);
    if (!ws || ws_bytes < wbytes + 256) return 0;
    if ((long long)B * Cout * OH * OW >= (1ll << 29) || (long long)Cout * Cin * KH * KW >= (1ll << 29) ||
        (long long)B * Cin * H * W >= (1ll << 30)) return 0;
    {   // round 5: the pre-split form takes the ORIGINAL filters (its prep kernel indexes them flipped / by parity class)
        const int rc2 = k33 ? mogan_dconv2_fwd_try(dy, w, 1, dx, B, Cout, Cin, OH, OW, H, W, 3, 3, KH - 1 - ph, KW - 1 - pw, H, W, 1, 1,
                                                   0, ws, ws_bytes, st, nullptr, d2query)
                            : mogan_dconv2_fwd_try(dy, w, 2, dx, B, Cout, Cin, OH, OW, gH, gW, 2, 2, 0, 0, H, W, 2, 4, 0, ws, ws_bytes, st,
                                                   d2prep, d2query);
        if (rc2 != 0) return rc2;
    }
    if (d2query) return 0;                               // dry run of the dispatch (mogan_conv_prep_bytes): nothing is launched
    float* wt = (float*)ws;                                          // transformed weights live at the head of ws
    void* ws2 = (char*)ws + ((wbytes + 255) & ~(size_t)255);
    const size_t ws2_bytes = ws_bytes - ((wbytes + 255) & ~(size_t)255);
    const long long wn = (long long)Cout * Cin * KH * KW;
    const unsigned nb = (unsigned)std::min<long long>(cdiv(wn, 256), 65536);
    DConvP p{};
    p.X = dy; p.Y = dx; p.B = B; p.Cin = Cout; p.Cout = Cin; p.Hs = OH; p.Ws = OW; p.H = OH; p.W = OW; p.up = 0;
    p.yH = H; p.yW = W; p.accumulate = 0; p.npar = 1;
    p.x_bytes = 4u * B * Cout * OH * OW;
    if (k33) {
        hipLaunchKernelGGL(wflip_kernel, dim3(nb), dim3(256), 0, st, w, wt, Cout, Cin, KH, KW);
        p.Wt = wt; p.w_bytes = (unsigned)wbytes;
        p.OH = H; p.OW = W; p.pt = KH - 1 - ph; p.pl = KW - 1 - pw; p.ys = 1; p.y0 = 0; p.x0 = 0;
        const int rc = launch_fwd<3, 3, 1, 8>(p, ws2, ws2_bytes, st);
        if (rc == MOGAN_ERR_WS) return 0;       // no room for the prepared filters: the implicit-GEMM path takes it
    return rc ? rc : 1;
    }
    hipLaunchKernelGGL(wparity_kernel, dim3(nb), dim3(256), 0, st, w, wt, Cout, Cin);
    // the four parity classes run in ONE launch (blockIdx.z): oyb0 = (py+1-kh0)/2 with kh0 = (py+1)&1 -> pt = 1-oyb0
    p.Wt = wt; p.w_bytes = 4u * Cin * Cout * 4; p.npar = 4;
    p.OH = gH; p.OW = gW; p.ys = 2; p.y0 = 0; p.x0 = 0; p.pt = 0; p.pl = 0;
    const int rc = launch_fwd<2, 2, 1, 8>(p, ws2, ws2_bytes, st);
    if (rc == MOGAN_ERR_WS) return 0;       // no room for the prepared filters: the implicit-GEMM path takes it
    return rc ? rc : 1;
}

int mogan_dconv_wgrad_try(const float* dy, const float* x, float* dw, int B, int Cin, int Hs, int Ws, int Cout, int KH,
                          int KW, int stride, int ph, int pw, int up, int accumulate, void* ws, size_t ws_bytes,
                          hipStream_t st) {
    const int H = Hs << up, W = Ws << up;
    const int OH = (H + 2 * ph - KH) / stride + 1, OW = (W + 2 * pw - KW) / stride + 1;
    const bool k33 = KH == 3 && KW == 3 && stride == 1 && ph == 1 && pw == 1;
    const bool k44 = KH == 4 && KW == 4 && stride == 2 && ph == 1 && pw == 1 && up == 0;
    if (!(k33 || k44)) return 0;
    if (!pow2(OH) || !pow2(OW) || OW < 16 || OH < 4 || Cout < 64 || Cin * KH * KW < 256) return 0;
    if ((((uintptr_t)dy) & 15) != 0) return 0;
    if ((long long)B * Cin * Hs * Ws >= (1ll << 29) || (long long)B * Cout * OH * OW >= (1ll << 29)) return 0;
    WGradP p{};
    p.dY = dy; p.X = x; p.dW = dw; p.B = B; p.Cin = Cin; p.Cout = Cout; p.Hs = Hs; p.Ws = Ws; p.H = H; p.W = W; p.up = up;
    p.OH = OH; p.OW = OW; p.pt = ph; p.pl = pw; p.N = Cin * KH * KW; p.accumulate = accumulate;
    p.x_bytes = 4u * B * Cin * Hs * Ws; p.y_bytes = 4u * B * Cout * OH * OW;
    const int rc = k33 ? launch_wgrad<3, 3, 1>(p, ws, ws_bytes, st) : launch_wgrad<4, 4, 2>(p, ws, ws_bytes, st);
    if (rc == MOGAN_ERR_WS) return 0;       // no room for the prepared filters: the implicit-GEMM path takes it
    return rc ? rc : 1;
}
