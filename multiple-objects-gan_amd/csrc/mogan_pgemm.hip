// mogan_pgemm.hip -- "panel" implicit GEMM for the weight-heavy convolutions (the deep layers of the discriminators).
//
// The deep D layers are GEMMs with a huge weight operand and few columns (768..3072 x 6144..27648 weights against N = B*OH*OW
// = 240..1024 pixels).  In gemm_kernel both operands are gathered as fp32 and split into their three bf16 pieces (mogan_mma.h)
// every time a K-tile is staged: 144 split + 176 gather VALU instructions per 48 MFMAs, and the weights -- which change once
// per optimizer step but are used by the real, fake and generator passes -- are split again in every launch.  Here both
// operands arrive PRE-SPLIT, in the order the matrix instruction consumes them:
//
//   weights      "A panels", written once per weight version by wpack_kernel (mogan_pk_weight_pack):
//                [class][m-tile of 32 rows][k-step of 16][piece 3][lane 64][8 bf16].  One (m-tile, k-step, piece) is the
//                1 KB A operand of one v_mfma_f32_32x32x16_bf16 in lane order, so a wave streams the weights of its rows
//                with coalesced 16-byte loads STRAIGHT INTO REGISTERS (no LDS, no VALU); every weight byte is read by
//                exactly one wave of a block.  K runs tap-major: k = (kh*KW + kw)*C + c.
//   activations  "pixel panels", written per call by apack_kernel into the workspace: channels-last bf16 pieces
//                [pixel][channel group of 32][piece 3][32 bf16] (192 contiguous bytes per pixel and K-tile).  Because K is
//                tap-major a K-tile of 32 is ONE tap and 32 consecutive channels, so the im2col gather of a row is a copy of
//                192 contiguous bytes from the pixel (oy*s - p + kh, ox*s - p + kw): twelve lanes per row, 16-byte loads,
//                16-byte LDS stores, ~8 VALU instructions of address arithmetic per chunk and NO materialised im2col matrix.
//
// pgemm_kernel<TM, TN>: 4 waves, wave w owns rows [w*TM*32, (w+1)*TM*32) x all TN*32 columns of the block tile; the pixel
// panel tile goes through LDS (208-byte rows, double buffered, one barrier per K-tile of 32), weights by direct loads one
// K-tile ahead; per 16 k a wave issues TM*3 global loads, TN*3 ds_read_b128 and TM*TN*6 MFMAs.  Forward: rows = Cout.
// Data gradient: one GEMM per stride-parity class (rows = Cin, K = Cout * taps of the class), all classes in one launch.
// Split-K slabs + mogan_splitk_reduce as in gemm_kernel (deterministic).
//
// Replaces (reference = stock torch ops): nn.Conv2d forward / backward-data of the deep discriminator layers,
// code/coco/attngan/model.py:594-613 (downBlock, Block3x3_leakRelu), 616-642 (D_GET_LOGITS jointConv), 738-760 (D_NET256).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include "../../include/mogan_hip.h"
#include "mogan_internal.h"
#include "mogan_mma.h"
#include <type_traits>
#ifndef PK_NSET
#define PK_NSET 3
#endif
#define PK_TRIP (PK_NSET == 2 ? 2 : 6)      // K-tiles per trip of the main loop

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PkP {
    const unsigned char* A;      // weight panels (class 0)
    const unsigned char* P;      // pixel panels
    float* C; float* ws;
    int M, N, K;                 // per class
    int Mt, KS;                  // ceil(M / 32), K / 16
    unsigned long long a_cls_stride;
    unsigned a_bytes, p_bytes;
    int ntile, kt_per, nsplit;   // K-tiles of 32 in all, per split
    long long slab;
    int accumulate;
    int dgrad, ncls;
    int Cc, CG;                  // channels of the K range (one tap), Cc / 32
    int CGp, cg0;                // channel groups of a pixel in the panel, first group of the K range's slice
    int PH, PW;                  // image dims of the pixel panel
    int RH, RW;                  // rows n -> (img, r, c) with r < RH, c < RW
    int s, ph, pw, nkw;          // nkw: taps per class along x (tap = ta * nkw + tb)
    int outH, outW;              // dims of the output image
    int gx, gy;
};

__device__ __forceinline__ unsigned xcd_order(unsigned L, unsigned total) {
    const unsigned k = L & 7u, j = L >> 3, q = total >> 3, r = total & 7u;
    return k * q + (k < r ? k : r) + j;
}

__device__ __forceinline__ uint4 ldg16(__amdgpu_buffer_rsrc_t r, unsigned off, bool ok) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, ok ? off : 0xFFFFFFF0u, 0, 0));
}

// WM x WN waves (WM * WN = 4): wave (wm, wn) owns rows [wm*TM*32, ..) x columns [wn*TN*32, ..) of the (WM*TM*32) x (WN*TN*32) block
// tile.  WM = 4: every weight byte is read by one wave of the block (the deep D layers: many rows, few columns); WM = 2: 64-row
// tiles for the frozen trunk's convolutions with 48..192 filters, whose last 128-row tile would be up to 62 % padding.
template <int TM, int TN, int WM = 4>
__device__ __forceinline__ void pgemm_body(const PkP& p, unsigned V, const unsigned nblk) {
    constexpr int WN = 4 / WM;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, RS = 208;
    constexpr int NCH = BN * 12 / 256;                 // 16-byte chunks of the pixel panel tile per thread and K-tile
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][BN * RS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    // block -> (n-tile, class, m-tile, K-split); n fastest, then the class: the blocks that stream the same weight rows
    // (the n-tiles; for a strided data gradient the classes read interleaved taps of the same filters) sit on one XCD
    V = xcd_order(V, nblk);
    const unsigned bx = V % p.gx; V /= p.gx;
    const unsigned cls = V % p.ncls; V /= p.ncls;
    const unsigned by = V % p.gy;
    const int sp = V / p.gy;
    const int n0 = bx * BN, m0 = by * BM;
    const int t_beg = sp * p.kt_per, t_end = min(p.ntile, t_beg + p.kt_per);

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + cls * p.a_cls_stride), (short)0,
                                                                         (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)p.P, (short)0, (int)p.p_bytes, 0x00020000);

    // rows of the pixel panel tile this thread stages (fixed over the K loop)
    int py = 0, px = 0, kh0 = 0, kw0 = 0;
    if (p.dgrad) { py = cls / p.s; px = cls % p.s; kh0 = (py + p.ph) % p.s; kw0 = (px + p.pw) % p.s; }
    const int RHW = p.RH * p.RW;
    int ry[NCH], rx[NCH]; unsigned pb[NCH]; unsigned ldso[NCH], cho[NCH]; bool rok[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + 256 * i, row = c / 12, ch = c - row * 12;
        const int n = n0 + row;
        rok[i] = n < p.N;
        const int nn = rok[i] ? n : 0;
        const int img = nn / RHW, rem = nn - img * RHW;
        const int r = rem / p.RW, cc = rem - r * p.RW;
        if (p.dgrad) { ry[i] = (r * p.s + py + p.ph - kh0) / p.s; rx[i] = (cc * p.s + px + p.pw - kw0) / p.s; }
        else { ry[i] = r * p.s - p.ph; rx[i] = cc * p.s - p.pw; }
        pb[i] = (unsigned)img * (unsigned)(p.PH * p.PW);
        ldso[i] = row * RS + ch * 16;
        cho[i] = ch * 16;                                  // the same offset inside the pixel's 192-byte group
    }
    const unsigned pixb = (unsigned)p.CGp * 192u;       // bytes per pixel in the panel
    const int sgn = p.dgrad ? -1 : 1;

    // weight stream of this wave: m-tile mt -> byte offset of its k-step 0, plus this lane's 16 bytes
    unsigned abase[TM]; bool aok[TM];
#pragma unroll
    for (int ta = 0; ta < TM; ++ta) {
        const int mt = (m0 >> 5) + wm * TM + ta;
        aok[ta] = mt < p.Mt;
        abase[ta] = (unsigned)mt * (unsigned)p.KS * 3072u + (unsigned)lane * 16u;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // A operands of one K-tile (2 k-steps x TM tiles x 3 pieces): NSET register sets, the loads run NSET - 1 K-tiles ahead of
    // the MFMAs that consume them (a K-tile lasts ~0.4 us at three waves per SIMD; an HBM round trip under load is longer)
    constexpr int NSET = PK_NSET;
    uint4 ra[NSET][2][TM][3];
    uint4 rb[NCH];

    // Loads past the end of this block's K range are issued all the same, at out-of-range offsets (the buffer unit returns
    // zeros): the loop body has no branch, and a trailing odd K-tile multiplies zeros.  (Uniform `if (more) load` branches
    // made the register sets phi nodes: 64 accumulator-file copies per trip.)
    auto load_a = [&](int t, uint4 (&ra)[2][TM][3]) {
        const bool live = t < t_end;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#ifdef PK_LAB_A2          // lab: two thirds of the weight bytes (the third piece is not loaded; results are wrong)
                    if (pl == 2) { ra[s2][ta][2] = ra[s2][ta][1]; continue; }
#endif
                    ra[s2][ta][pl] = ldg16(rA, abase[ta] + (unsigned)((t * 2 + s2) * 3 + pl) * 1024u, aok[ta] & live);
                }
    };
    auto load_b = [&](int t) {
        const bool live = t < t_end;
        const int tap = t / p.CG, cg = t - tap * p.CG;           // uniform
        const int ta = tap / p.nkw, tb = tap - ta * p.nkw;
        const int dy = sgn * ta, dx = sgn * tb;
        const unsigned goff = (unsigned)(p.cg0 + cg) * 192u;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int iy = ry[i] + dy, ix = rx[i] + dx;
            const bool ok = (int)rok[i] & (int)live & (int)((unsigned)iy < (unsigned)p.PH) & (int)((unsigned)ix < (unsigned)p.PW);   // no short circuit
            unsigned off = (pb[i] + (unsigned)(iy * p.PW + ix)) * pixb + goff + cho[i];
            asm volatile("" : "+v"(off));      // keep the address arithmetic unconditional: otherwise it is sunk behind an
            rb[i] = ldg16(rP, off, ok);        // exec-mask branch per chunk instead of one v_cndmask
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) *(uint4*)(&Bs[buf][ldso[i]]) = rb[i];
    };
    const int brow = (lane & 31) * RS + (lane >> 5) * 16;
    // column tile by column tile: the three pieces of B tile tb + 1 are read from LDS while the TM * 6 MFMAs of tile tb run, so
    // only two B fragments are live; every accumulator still receives its six partial products smallest first
    auto read_b = [&](int buf, int s2, int tb) {
        X6Frag f;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            f.p[pl] = __builtin_bit_cast(mma_bf16x8, *(const uint4*)(&Bs[buf][brow + (wn * TN + tb) * 32 * RS + pl * 64 + s2 * 32]));
        return f;
    };
    auto compute = [&](int buf, const uint4 (&ra)[2][TM][3]) {
        X6Frag fb = read_b(buf, 0, 0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            X6Frag fa[TM];
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fa[ta].p[pl] = __builtin_bit_cast(mma_bf16x8, ra[s2][ta][pl]);
#pragma unroll
            for (int tb = 0; tb < TN; ++tb) {
                const X6Frag cur = fb;
                if (tb + 1 < TN) fb = read_b(buf, s2, tb + 1);
                else if (s2 == 0) fb = read_b(buf, 1, 0);
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int ta = 0; ta < TM; ++ta) acc[ta][tb] = x6_mfma(fa[ta], cur, term, acc[ta][tb]);
            }
        }
    };

    // (No early exit inside a trip: a `break` between the stages gives the loop several exits with different live register
    // sets and the allocator answers with hundreds of spills.  Tiles past t_end load zeros -- see load_a -- and the host rounds
    // the K range of a split to a multiple of the trip length.)
    // one K-tile: B tile t + 1 to registers, A tile t + NSET - 1 to its register set, the MFMAs of tile t, B tile t + 1 to the
    // other LDS buffer.  SET / BUF are compile-time so that every register set is named statically.
    auto stage = [&](auto SET, auto BUF, int t) {
        constexpr int S = decltype(SET)::value, Bf = decltype(BUF)::value;
        load_b(t + 1);
        load_a(t + NSET - 1, ra[(S + NSET - 1) % NSET]);
        compute(Bf, ra[S]);
        store_b(Bf ^ 1);
        __syncthreads();
    };
    if (t_beg < t_end) {
        load_b(t_beg);
#pragma unroll
        for (int j = 0; j < NSET - 1; ++j) load_a(t_beg + j, ra[j]);
        store_b(0);
        __syncthreads();
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        if constexpr (NSET == 2) {
            for (int t = t_beg; t < t_end; t += 2) {
                stage(I0{}, I0{}, t);
                stage(I1{}, I1{}, t + 1);
            }
        } else {
            using I2 = std::integral_constant<int, 2>;
            for (int t = t_beg; t < t_end; t += 6) {
                stage(I0{}, I0{}, t);
                stage(I1{}, I1{}, t + 1);
                stage(I2{}, I0{}, t + 2);
                stage(I0{}, I1{}, t + 3);
                stage(I1{}, I0{}, t + 4);
                stage(I2{}, I1{}, t + 5);
            }
        }
    }

    // ---------------------------------------------------------------- epilogue
    const bool split = p.nsplit > 1;
    float* __restrict__ slab = p.ws + (size_t)sp * p.slab;
    const bool addc = !split && p.accumulate;
    const size_t cms = (size_t)p.outH * p.outW;
#pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
        const int n = n0 + (wn * TN + tb) * 32 + (lane & 31);
        const bool nok = n < p.N;
        const int nn = nok ? n : 0;
        const int img = nn / RHW, rem = nn - img * RHW;
        const int r = rem / p.RW, cc = rem - r * p.RW;
        const size_t pos = p.dgrad ? (size_t)(r * p.s + py) * p.outW + (cc * p.s + px) : (size_t)rem;
        const size_t cbase = (size_t)img * p.M * cms + pos;
#pragma unroll
        for (int ta = 0; ta < TM; ++ta) {
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) {
                const int m = m0 + (wm * TM + ta) * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * (lane >> 5);
                if (nok && m < p.M) {
                    float v = acc[ta][tb][r16];
                    float* dst = (split ? slab : p.C) + cbase + (size_t)m * cms;
                    if (addc) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

template <int TM, int TN, int OCC>
__global__ __launch_bounds__(256, OCC) void pgemm_kernel(const PkP p) { pgemm_body<TM, TN>(p, blockIdx.x, gridDim.x); }

// Several independent GEMMs (own weights, panels, geometry and K-split; one tile shape) as ONE launch: the members' blocks lie
// end to end on the 1-D grid.  The frozen Inception trunk (attngan/inception.py) issues the convolutions of one dependency level
// of a Mixed block this way: 1-5 GFLOP each at B = 16, none of which fills 256 CUs alone.
constexpr int PK_MAXG = 8;
struct PkGroup { PkP p[PK_MAXG]; unsigned end[PK_MAXG]; int n; };

template <int TM, int TN, int OCC, int WM>
__global__ __launch_bounds__(256, OCC) void pgemm_group_kernel(const PkGroup g) {
    int pi = 0;
#pragma unroll
    for (int i = 0; i < PK_MAXG - 1; ++i) if (i + 1 < g.n && blockIdx.x >= g.end[i]) pi = i + 1;
    const unsigned start = pi ? g.end[pi - 1] : 0u;
    pgemm_body<TM, TN, WM>(g.p[pi], blockIdx.x - start, g.end[pi] - start);
}

// ------------------------------------------------------------------------------------------------ pack kernels
// NCHW fp32 -> channels-last bf16 pieces: P[q = (img, y, x)][cg][piece][32].  Block = 32 channels x 64 pixels.
__global__ __launch_bounds__(256) void apack_kernel(const float* __restrict__ x, unsigned char* __restrict__ P, int Bn, int C,
                                                    int HW) {
    __shared__ float L[32][65];
    const int tid = threadIdx.x;
    const long long Q = (long long)Bn * HW;
    const long long q0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = tid + 256 * j, cl = e >> 6, ql = e & 63;
        const long long q = q0 + ql;
        float v = 0.f;
        if (q < Q) {
            const long long img = q / HW, pix = q - img * HW;
            v = x[(img * C + c0 + cl) * HW + pix];
        }
        L[cl][ql] = v;
    }
    __syncthreads();
    const int ql = tid >> 2, g8 = tid & 3;
    const long long q = q0 + ql;
    if (q >= Q) return;
    uint32_t w[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x6_split2(L[g8 * 8 + 2 * j][ql], L[g8 * 8 + 2 * j + 1][ql], w[0][j], w[1][j], w[2][j]);
    unsigned char* d = P + ((size_t)q * (C >> 5) + blockIdx.y) * 192 + g8 * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *(uint4*)(d + pl * 64) = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
}

// W (Cout, Cin, KH, KW) fp32 -> weight panels.  Block = 32 rows (m) x 16 channels (c) x all taps through LDS: coalesced
// reads of the fp32 master, 16-byte stores of whole lane slots.  dgrad = 0: m = co, c = ci, k = tap * Cin + ci.
// dgrad = 1: m = ci, c = co; tap (kh, kw) belongs to the stride-parity class (py, px) with (py + ph) % s == kh % s and is its
// tap (a, b) = (kh / s, kw / s): k = (a * nkw + b) * Cout + co.
struct WpackP { const float* w; unsigned char* A; int Cout, Cin, KH, KW, s, ph, pw, dgrad, M, Cc, Mt, KS; unsigned long long cls_stride; };

__global__ __launch_bounds__(256) void wpack_kernel(const WpackP p) {
    extern __shared__ float L[];                       // [tap][32 m][17]
    const int tid = threadIdx.x;
    const int KHW = p.KH * p.KW;
    const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 16;
    const int total = 32 * 16 * KHW;
    for (int e = tid; e < total; e += 256) {
        int mi, cl, tap; size_t src;
        if (!p.dgrad) {
            mi = e / (16 * KHW); const int rem = e - mi * 16 * KHW;
            cl = rem / KHW; tap = rem - cl * KHW;
            src = ((size_t)(m0 + mi) * p.Cin + c0) * KHW + rem;
        } else {
            cl = e / (32 * KHW); const int rem = e - cl * 32 * KHW;
            mi = rem / KHW; tap = rem - mi * KHW;
            src = ((size_t)(c0 + cl) * p.Cin + m0) * KHW + rem;
        }
        L[(tap * 32 + mi) * 17 + cl] = (m0 + mi < p.M) ? p.w[src] : 0.f;
    }
    __syncthreads();
    const int nkw = p.KW / p.s;
    for (int it = tid; it < KHW * 64; it += 256) {
        const int tap = it >> 6, lane = it & 63, mi = lane & 31, h = lane >> 5;
        const float* v = &L[(tap * 32 + mi) * 17 + 8 * h];
        uint32_t w[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x6_split2(v[2 * j], v[2 * j + 1], w[0][j], w[1][j], w[2][j]);
        int cls = 0, ktap = tap;
        if (p.dgrad) {
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const int py = ((kh - p.ph) % p.s + p.s) % p.s, px = ((kw - p.pw) % p.s + p.s) % p.s;
            cls = py * p.s + px;
            ktap = (kh / p.s) * nkw + (kw / p.s);
        }
        const int ks = (ktap * p.Cc + c0) >> 4;
        unsigned char* d = p.A + (size_t)cls * p.cls_stride + ((size_t)blockIdx.x * p.KS + ks) * 3072 + lane * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *(uint4*)(d + pl * 1024) = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
    }
}


// Both directions from ONE read of the fp32 master (what the optimizer's owner calls after every step): block = 32 co x 16 ci
// x all taps.  Forward panels: rows = co, one (m-tile, k-step) unit per tap.  Data-gradient panels: rows = ci (these 16 are
// half of an m-tile: 32 of a unit's 64 lane slots), two k-steps of 16 co per tap.  KHW is a template parameter (no run-time
// divisions), the master is read with 16-byte loads.
struct Wpack2P { const float* w; unsigned char* Af; unsigned char* Ad; int Cout, Cin, KW, s, ph, pw, KSf, KSd;
                 unsigned long long cls_stride; };

template <int KHW>
__global__ __launch_bounds__(256) void wpack2_kernel(const Wpack2P p) {
    __shared__ float L[KHW * 32 * 17];                 // [tap][32 co][16 ci + 1]
    const int tid = threadIdx.x;
    const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 16;
    constexpr int ROW = 16 * KHW;                      // floats per co row of the tile (contiguous in the master)
    if constexpr (ROW % 4 == 0) {
        for (int q = tid; q < 32 * ROW / 4; q += 256) {
            const int mi = q / (ROW / 4), r4 = (q - mi * (ROW / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co0 + mi < p.Cout) v = *(const float4*)(p.w + ((size_t)(co0 + mi) * p.Cin + ci0) * KHW + r4);
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int r = r4 + j; L[((r % KHW) * 32 + mi) * 17 + r / KHW] = e[j]; }
        }
    } else {
        for (int q = tid; q < 32 * ROW; q += 256) {
            const int mi = q / ROW, r = q - mi * ROW;
            L[((r % KHW) * 32 + mi) * 17 + r / KHW] = (co0 + mi < p.Cout) ? p.w[((size_t)(co0 + mi) * p.Cin + ci0) * KHW + r] : 0.f;
        }
    }
    __syncthreads();
    const int nkw = p.KW / p.s;
    for (int it = tid; it < KHW * 128; it += 256) {
        const int tap = it >> 7, sub = it & 127;
        uint32_t w[3][4];
        unsigned char* d;
        if (sub < 64) {                                  // forward: lane = (co, half of the 16 ci)
            const int mi = sub & 31, h = sub >> 5;
            const float* v = &L[(tap * 32 + mi) * 17 + 8 * h];
#pragma unroll
            for (int j = 0; j < 4; ++j) x6_split2(v[2 * j], v[2 * j + 1], w[0][j], w[1][j], w[2][j]);
            const int ks = (tap * p.Cin + ci0) >> 4;
            d = p.Af + ((size_t)blockIdx.x * p.KSf + ks) * 3072 + sub * 16;
        } else {                                         // data gradient: lane = (ci, half), k-step = co group of 16
            const int u = sub - 64, cil = u & 15, h = (u >> 4) & 1, kstep = u >> 5;
            const float* v = &L[(tap * 32 + kstep * 16 + h * 8) * 17 + cil];
#pragma unroll
            for (int j = 0; j < 4; ++j) x6_split2(v[(2 * j) * 17], v[(2 * j + 1) * 17], w[0][j], w[1][j], w[2][j]);
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const int py = ((kh - p.ph) % p.s + p.s) % p.s, px = ((kw - p.pw) % p.s + p.s) % p.s;
            const int ktap = (kh / p.s) * nkw + (kw / p.s);
            const int ks = ((ktap * p.Cout + co0) >> 4) + kstep;
            const int ci = ci0 + cil, lane = (ci & 31) + 32 * h;
            d = p.Ad + (size_t)(py * p.s + px) * p.cls_stride + ((size_t)(ci >> 5) * p.KSd + ks) * 3072 + lane * 16;
            if (co0 + kstep * 16 >= p.Cout) continue;    // (Cout % 32 == 0 is required, kept for safety)
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *(uint4*)(d + pl * 1024) = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
    }
}


// ------------------------------------------------------------------------------------------------ weight gradient operands
// dW[co][r] (+)= sum_n dY[co][n] * Xcol[r][n], r = (ci, kh, kw), n = (img, oy, ox): the same GEMM kernel with the roles dealt
// anew -- dY takes the place of the weights (A panels over K = n, zero-padded to a multiple of 32), the transposed im2col matrix
// the place of the pixel panel (one "pixel" per row r, its "channels" are the n), the output (Cout x Cin*KH*KW, row-major) IS dW.
// Both operands are packed per call: the deep layers have K = B*OH*OW <= a few thousand, so the im2col matrix is small
// (<= 150 MB) while the 300 MB read-modify-write of dW and the 77 GFLOP stay what they are.
__global__ __launch_bounds__(256) void dypack_kernel(const float* __restrict__ dy, unsigned char* __restrict__ A, int Cout, int OHW,
                                                     int Npix, int KS) {
    const long long it = (long long)blockIdx.x * 256 + threadIdx.x;           // (m-tile, k-step, lane)
    const int lane = (int)(it & 63);
    const long long u = it >> 6;
    const int ks = (int)(u % KS), mt = (int)(u / KS);
    const int co = mt * 32 + (lane & 31);
    const int n0 = ks * 16 + (lane >> 5) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = n0 + j;
        float t = 0.f;
        if (co < Cout && n < Npix) { const int img = n / OHW, pix = n - img * OHW; t = dy[((size_t)img * Cout + co) * OHW + pix]; }
        v[j] = t;
    }
    uint32_t w[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x6_split2(v[2 * j], v[2 * j + 1], w[0][j], w[1][j], w[2][j]);
    unsigned char* d = A + ((size_t)mt * KS + ks) * 3072 + lane * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *(uint4*)(d + pl * 1024) = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
}

struct XtP { const float* x; unsigned char* P; int Cin, Hs, Ws, KH, KW, s, ph, pw, OH, OW, Npix, KG; long long total; };
__global__ __launch_bounds__(256) void xtpack_kernel(const XtP p) {
    const long long it = (long long)blockIdx.x * 256 + threadIdx.x;           // (row r, group of 8 n)
    if (it >= p.total) return;
    const int g8 = (int)(it % (p.KG * 4)), r = (int)(it / (p.KG * 4));
    const int KHW = p.KH * p.KW, ci = r / KHW, tap = r - ci * KHW, kh = tap / p.KW, kw = tap - kh * p.KW;
    const int OHW = p.OH * p.OW;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = g8 * 8 + j;
        float t = 0.f;
        if (n < p.Npix) {
            const int img = n / OHW, pix = n - img * OHW, oy = pix / p.OW, ox = pix - oy * p.OW;
            const int iy = oy * p.s - p.ph + kh, ix = ox * p.s - p.pw + kw;
            if ((unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws)
                t = p.x[(((size_t)img * p.Cin + ci) * p.Hs + iy) * p.Ws + ix];
        }
        v[j] = t;
    }
    uint32_t w[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x6_split2(v[2 * j], v[2 * j + 1], w[0][j], w[1][j], w[2][j]);
    unsigned char* d = p.P + ((size_t)r * p.KG + (g8 >> 2)) * 192 + (g8 & 3) * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *(uint4*)(d + pl * 64) = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
}


// ------------------------------------------------------------------------------------------------ fused tails
// conv -> BatchNorm(train) -> LeakyReLU/ReLU of a deep layer has B*OH*OW <= 2048 values per channel: ONE block computes the
// batch statistics of 8 channels, applies them and hands the next layer its pixel panel -- instead of split-K reduce,
// bn_partial, bn_finalize, bn_act_fwd and apack (5 launches).  32 lanes (half a wave) own a channel, its values stay in
// registers between the statistics and the apply step; arithmetic as mogan_norm.hip (fp64 sums, biased variance for the
// normalisation, unbiased for running_var).
template <int ACT>
__device__ __forceinline__ float act_apply(float t, float slope) {
    if (ACT == MOGAN_ACT_RELU) return t > 0.f ? t : 0.f;
    if (ACT == MOGAN_ACT_LRELU) return t > 0.f ? t : t * slope;
    return t;
}
__device__ __forceinline__ double half_wave_sum(double v) {          // all 32 lanes of the half get the sum
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// 8 channels x NE pixels of fp32 in LDS (L[ch][e]) -> pieces of the pixel panel: one 16-byte unit per pixel and piece
__device__ __forceinline__ void panel_from_lds(const float* L, int NE, unsigned char* P, int C, int c0) {
    const int cg = c0 >> 5, g8 = (c0 & 31) >> 3;
    for (int e = threadIdx.x; e < NE; e += blockDim.x) {
        uint32_t w[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x6_split2(L[(2 * j) * NE + e], L[(2 * j + 1) * NE + e], w[0][j], w[1][j], w[2][j]);
        unsigned char* d = P + ((size_t)e * (C >> 5) + cg) * 192 + g8 * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *(uint4*)(d + pl * 64) = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
    }
}

// Groups (G > 1): the tensor holds G batches of B images one behind the other, each of which the reference passes through the
// layer in a call of its own (D(real) and D(fake.detach()) of a discriminator update, miscc/losses.py:136-174): own batch statistics
// per group, running statistics updated group after group in that order.  256 threads per group (blockDim = 256 * G, G <= 2): the groups'
// loads, sums and stores run side by side, the 8 channels' pixel panel covers all G * B images.
struct TailP {
    const float* src; int nsplit; long long slab;       // conv output = sum of nsplit slabs (nsplit == 1: src is y itself)
    const float* gamma; const float* beta; float* rmean; float* rvar;
    float* y; float* mean; float* invstd; float* z; unsigned char* zpanel;     // mean / invstd: [G][C]
    int B, C, HW, G; float eps, momentum, slope;
};

template <int ACT, int EPT, int GT>
__global__ __launch_bounds__(256 * GT) void deep_tail_fwd_kernel(const TailP p) {
    extern __shared__ float L[];
    __shared__ float S[GT * 8 * 2];                     // (group, channel) -> batch mean, unbiased variance
    const int tid = threadIdx.x & 255, grp = threadIdx.x >> 8, chl = tid >> 5, j = tid & 31;
    const int c0 = blockIdx.x * 8, c = c0 + chl;
    const int NE = p.B * p.HW, b0 = grp * p.B;
    float v[EPT];
    unsigned idx[EPT];                                  // (the tensor has < 2^31 elements: checked by the entry point)
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = j + 32 * i;
        const int b = e / p.HW, pos = e - b * p.HW;
        idx[i] = ((unsigned)(b0 + b) * p.C + c) * p.HW + pos;
        v[i] = e < NE ? p.src[idx[i]] : 0.f;
    }
    // the K-split slabs in their fixed order, EPT independent loads in flight per slab (a dependent chain of nsplit loads per
    // element made this kernel 45 us long)
    constexpr int SB = EPT <= 8 ? 4 : (EPT <= 16 ? 2 : 1);       // slabs per trip: ~32 loads in flight per thread
    int sp = 1;
    for (; sp + SB <= p.nsplit; sp += SB) {
        float a[SB][EPT];
#pragma unroll
        for (int u = 0; u < SB; ++u)
#pragma unroll
            for (int i = 0; i < EPT; ++i) a[u][i] = (j + 32 * i) < NE ? p.src[(size_t)(sp + u) * p.slab + idx[i]] : 0.f;
#pragma unroll
        for (int u = 0; u < SB; ++u)
#pragma unroll
            for (int i = 0; i < EPT; ++i) v[i] += a[u][i];
    }
    for (; sp < p.nsplit; ++sp) {
        const float* sl = p.src + (size_t)sp * p.slab;
        float a[EPT];
#pragma unroll
        for (int i = 0; i < EPT; ++i) a[i] = (j + 32 * i) < NE ? sl[idx[i]] : 0.f;
#pragma unroll
        for (int i = 0; i < EPT; ++i) v[i] += a[i];
    }
    double s1 = 0.0, s2 = 0.0;
    const bool wr = p.nsplit > 1 || p.src != p.y;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        if (j + 32 * i < NE) {
            if (wr) p.y[idx[i]] = v[i];
            s1 += (double)v[i]; s2 += (double)v[i] * v[i];
        }
    }
    s1 = half_wave_sum(s1); s2 = half_wave_sum(s2);
    const double n = (double)NE;
    const double m = s1 / n;
    double var = s2 / n - m * m; if (var < 0) var = 0;
    const float mu = (float)m, is = (float)(1.0 / sqrt(var + (double)p.eps));
    if (j == 0) {
        p.mean[grp * p.C + c] = mu; p.invstd[grp * p.C + c] = is;
        const double unb = n > 1 ? var * n / (n - 1.0) : var;
        if (GT == 1) {
            if (p.rmean) p.rmean[c] = (1.f - p.momentum) * p.rmean[c] + p.momentum * mu;
            if (p.rvar) p.rvar[c] = (1.f - p.momentum) * p.rvar[c] + p.momentum * (float)unb;
        } else {
            S[(grp * 8 + chl) * 2] = mu; S[(grp * 8 + chl) * 2 + 1] = (float)unb;
        }
    }
    const float sc = p.gamma[c] * is, sh = p.beta[c] - mu * sc;
    const int NT = NE * GT;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = j + 32 * i;
        if (e < NE) {
            const float t = act_apply<ACT>(v[i] * sc + sh, p.slope);
            p.z[idx[i]] = t;
            if (p.zpanel) L[chl * NT + grp * NE + e] = t;
        }
    }
    if (p.zpanel || GT > 1) __syncthreads();
    if (GT > 1 && grp == 0 && j == 0) {                // the groups' running-statistics updates, in call order
        float rm = p.rmean ? p.rmean[c] : 0.f, rv = p.rvar ? p.rvar[c] : 0.f;
        for (int g = 0; g < GT; ++g) {
            rm = (1.f - p.momentum) * rm + p.momentum * S[(g * 8 + chl) * 2];
            rv = (1.f - p.momentum) * rv + p.momentum * S[(g * 8 + chl) * 2 + 1];
        }
        if (p.rmean) p.rmean[c] = rm;
        if (p.rvar) p.rvar[c] = rv;
    }
    if (p.zpanel) panel_from_lds(L, NT, p.zpanel, p.C, c0);
}

struct TailBwdP {
    const float* dz; const float* y; const float* mean; const float* invstd; const float* gamma; const float* beta;
    float* dy; unsigned char* dypanel; float* dgamma; float* dbeta; int accumulate;
    int B, C, HW, G; float slope;
};

template <int ACT, int EPT, int GT>
__global__ __launch_bounds__(256 * GT) void deep_tail_bwd_kernel(const TailBwdP p) {
    extern __shared__ float L[];
    __shared__ float S[GT * 8 * 2];
    const int tid = threadIdx.x & 255, grp = threadIdx.x >> 8, chl = tid >> 5, j = tid & 31;
    const int c0 = blockIdx.x * 8, c = c0 + chl;
    const int NE = p.B * p.HW, b0 = grp * p.B, NT = NE * GT;
    const float mu = p.mean[grp * p.C + c], is = p.invstd[grp * p.C + c];
    const float sc = p.gamma[c] * is, sh = p.beta[c] - mu * sc;
    float g[EPT], xh[EPT];
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = j + 32 * i;
        g[i] = 0.f; xh[i] = 0.f;
        if (e < NE) {
            const int b = e / p.HW, pos = e - b * p.HW;
            const size_t idx = ((size_t)(b0 + b) * p.C + c) * p.HW + pos;
            const float xa = p.y[idx], d = p.dz[idx];
            const float t = xa * sc + sh;
            float da = d;
            if (ACT == MOGAN_ACT_RELU) da = t > 0.f ? d : 0.f;
            if (ACT == MOGAN_ACT_LRELU) da = t > 0.f ? d : d * p.slope;
            g[i] = da; xh[i] = (xa - mu) * is;
            a0 += da; a1 += (double)da * xh[i];
        }
    }
    a0 = half_wave_sum(a0); a1 = half_wave_sum(a1);
    const float f0 = (float)a0, f1 = (float)a1;
    if (j == 0) {
        if (GT == 1) {
            if (p.dbeta) p.dbeta[c] = (p.accumulate ? p.dbeta[c] : 0.f) + f0;
            if (p.dgamma) p.dgamma[c] = (p.accumulate ? p.dgamma[c] : 0.f) + f1;
        } else {
            S[(grp * 8 + chl) * 2] = f0; S[(grp * 8 + chl) * 2 + 1] = f1;
        }
    }
    const float inv_n = 1.f / ((float)p.B * (float)p.HW);
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = j + 32 * i;
        if (e < NE) {
            const int b = e / p.HW, pos = e - b * p.HW;
            const float d = sc * (g[i] - f0 * inv_n - xh[i] * f1 * inv_n);
            p.dy[((size_t)(b0 + b) * p.C + c) * p.HW + pos] = d;
            if (p.dypanel) L[chl * NT + grp * NE + e] = d;
        }
    }
    if (p.dypanel || GT > 1) __syncthreads();
    if (GT > 1 && grp == 0 && j == 0) {                // d gamma / d beta: the groups' contributions in call order
        float db = (p.accumulate && p.dbeta) ? p.dbeta[c] : 0.f, dg = (p.accumulate && p.dgamma) ? p.dgamma[c] : 0.f;
        for (int q = 0; q < GT; ++q) { db += S[(q * 8 + chl) * 2]; dg += S[(q * 8 + chl) * 2 + 1]; }
        if (p.dbeta) p.dbeta[c] = db;
        if (p.dgamma) p.dgamma[c] = dg;
    }
    if (p.dypanel) panel_from_lds(L, NT, p.dypanel, p.C, c0);
}

// dynamic LDS beyond 64 KB (two groups of a 8 x 8 map at B = 16: 8 channels x 2048 values) needs the attribute, once per kernel
template <typename K>
static void tail_lds_attr(K kern, size_t lds) {
    if (lds > 60 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
}
template <int ACT, int EPT, int GT>
static void launch_tail_fwd_g(const TailP& p, size_t lds, hipStream_t st) {
    static bool attr = false;
    if (!attr && lds > 60 * 1024) { tail_lds_attr(deep_tail_fwd_kernel<ACT, EPT, GT>, lds); attr = true; }
    hipLaunchKernelGGL((deep_tail_fwd_kernel<ACT, EPT, GT>), dim3(p.C / 8), dim3(256 * GT), lds, st, p);
}
template <int ACT, int EPT>
static void launch_tail_fwd_e(const TailP& p, size_t lds, hipStream_t st) {
    if (p.G == 2) launch_tail_fwd_g<ACT, (EPT < 64 ? EPT : 32), 2>(p, lds, st); else launch_tail_fwd_g<ACT, EPT, 1>(p, lds, st);
}
template <int ACT, int EPT, int GT>
static void launch_tail_bwd_g(const TailBwdP& p, size_t lds, hipStream_t st) {
    static bool attr = false;
    if (!attr && lds > 60 * 1024) { tail_lds_attr(deep_tail_bwd_kernel<ACT, EPT, GT>, lds); attr = true; }
    hipLaunchKernelGGL((deep_tail_bwd_kernel<ACT, EPT, GT>), dim3(p.C / 8), dim3(256 * GT), lds, st, p);
}
template <int ACT, int EPT>
static void launch_tail_bwd_e(const TailBwdP& p, size_t lds, hipStream_t st) {
    if (p.G == 2) launch_tail_bwd_g<ACT, (EPT < 64 ? EPT : 32), 2>(p, lds, st); else launch_tail_bwd_g<ACT, EPT, 1>(p, lds, st);
}
template <int ACT>
static int launch_tail_fwd(const TailP& p, hipStream_t st) {
    const int NE = p.B * p.HW;
    const size_t lds = p.zpanel ? (size_t)8 * NE * p.G * sizeof(float) : 0;
    if (NE <= 256) launch_tail_fwd_e<ACT, 8>(p, lds, st);
    else if (NE <= 512) launch_tail_fwd_e<ACT, 16>(p, lds, st);
    else if (NE <= 1024) launch_tail_fwd_e<ACT, 32>(p, lds, st);
    else launch_tail_fwd_e<ACT, 64>(p, lds, st);
    return 0;
}
template <int ACT>
static int launch_tail_bwd(const TailBwdP& p, hipStream_t st) {
    const int NE = p.B * p.HW;
    const size_t lds = p.dypanel ? (size_t)8 * NE * p.G * sizeof(float) : 0;
    if (NE <= 256) launch_tail_bwd_e<ACT, 8>(p, lds, st);
    else if (NE <= 512) launch_tail_bwd_e<ACT, 16>(p, lds, st);
    else if (NE <= 1024) launch_tail_bwd_e<ACT, 32>(p, lds, st);
    else launch_tail_bwd_e<ACT, 64>(p, lds, st);
    return 0;
}
// B images per group, G groups: a half wave holds a (channel, group)'s B * HW values in registers (<= 2048; two groups = 512
// threads = half the registers per lane: <= 1024), the block's LDS the 8-channel panel of all groups (<= 64 KB)
static bool tail_ok(int B, int C, int HW, int act, int G = 1) {
    return B > 0 && HW > 0 && C > 0 && C % 32 == 0 && G >= 1 && G <= 2 && (long long)B * HW * G <= 2048 &&
           (act == MOGAN_ACT_NONE || act == MOGAN_ACT_RELU || act == MOGAN_ACT_LRELU);
}

// ------------------------------------------------------------------------------------------------ host side
static inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }
static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

static int g_pk_force = 0;       // test hook: 1 = take every shape that meets the hard constraints

struct PkGeom { int ncls, nkh, nkw, M, Cc, K, Mt, KS; unsigned long long cls_bytes; };

// hard constraints of the panel formats (independent of any size heuristic)
static bool pk_geom(int Cout, int Cin, int KH, int KW, int s, int dgrad, PkGeom& g) {
    if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || s <= 0) return false;
    if (!dgrad) {
        if (Cin % 32) return false;
        g.ncls = 1; g.nkh = KH; g.nkw = KW; g.M = Cout; g.Cc = Cin;
    } else {
        if (Cout % 32 || KH % s || KW % s) return false;
        g.ncls = s * s; g.nkh = KH / s; g.nkw = KW / s; g.M = Cin; g.Cc = Cout;
    }
    if (KH * KW > 25) return false;                     // LDS tile of wpack_kernel
    g.K = g.nkh * g.nkw * g.Cc;
    g.Mt = (int)cdiv(g.M, 32); g.KS = g.K / 16;
    g.cls_bytes = (unsigned long long)g.Mt * g.KS * 3072ull;
    return g.cls_bytes < (1ull << 32);
}

// tile shapes (block = 128 * tm rows x 32 * tn columns).  Measured on the deep D layers at B = 16 (tools/time_pk.py): 128 x 64 at
// three waves per SIMD wins on every layer (the kernel is latency bound: 256 x 128 at one wave per SIMD ran at half its rate),
// so the heuristic takes it unless a larger tile has clearly less padding; the others stay reachable for tuning / tests
struct PkCfg { int tm, tn; };
static const PkCfg kPk[] = {{1, 2}, {1, 4}, {2, 2}};
enum { NPK = 3 };
static int g_pk_cfg = -1, g_pk_split = 0;
#ifndef PK_OCC12
#define PK_OCC12 3            // waves per SIMD of the 128 x 64 tile (lab: 4 with -DPK_NSET=2)
#endif
#ifndef PK_G64_OCC
#define PK_G64_OCC 2
#endif
static int g_pk_group_tile = -1;          // lab: MOGAN_PK_GROUP_TILE=0/1 forces the 128x64 / 64x128 tile of mogan_pk_group
static inline int a_prof_mode(int dgrad) { return dgrad ? 11 : 10; }    // launch-profile modes of the grouped launches

// OCC = waves per SIMD the register allocation is held to (512 / OCC registers per lane)
template <int TM, int TN, int OCC>
static void launch_pk(unsigned blocks, hipStream_t st, const PkP& p) {
    hipLaunchKernelGGL((pgemm_kernel<TM, TN, OCC>), dim3(blocks), dim3(256), 0, st, p);
}

// keep_slabs: with nsplit > 1 the K-split slabs stay in `ws` for a consumer that sums them itself (deep_tail_fwd_kernel)
static int run_pk(PkP& p, void* ws, size_t ws_bytes, int prof_mode, hipStream_t st, bool keep_slabs = false) {
    // tile: least padded work; 128-row tiles (two blocks per CU) need ~30 % more LDS traffic per MFMA than 256-row tiles
    int best = 0; double bestw = 1e300;
    for (int c = 0; c < NPK; ++c) {
        const int bm = 128 * kPk[c].tm, bn = 32 * kPk[c].tn;
        double w = (double)cdiv(p.M, bm) * bm * (double)cdiv(p.N, bn) * bn;
        w *= c == 0 ? 1.0 : 1.15;
        if (w < bestw * 0.999) { bestw = w; best = c; }
    }
    if (g_pk_cfg >= 0) best = g_pk_cfg;
    const int bm = 128 * kPk[best].tm, bn = 32 * kPk[best].tn;
    p.gx = (int)cdiv(p.N, bn); p.gy = (int)cdiv(p.M, bm);
    const long long tiles = (long long)p.gx * p.gy * p.ncls;
    const long long c_numel = (long long)p.slab;
    int nsplit = 1;
    // blocks per launch as gemm_kernel's split-K aims at (768 alone on the GPU, 384 in the multi-stream step)
    const int target = mogan_split_target(st);
    if (tiles < target && p.ntile >= 8) {
        nsplit = (int)cdiv(target, tiles);
        nsplit = std::min(nsplit, p.ntile / 4);
        if (nsplit < 1) nsplit = 1;
    }
    if (g_pk_split > 0) nsplit = std::min(g_pk_split, p.ntile);
    if (nsplit > 1) {
        const long long fit = ws ? (long long)(ws_bytes / (sizeof(float) * (size_t)c_numel)) : 0;
        if (fit < 2) nsplit = 1; else nsplit = (int)std::min<long long>(nsplit, fit);
    }
    p.kt_per = (int)cdiv(cdiv(p.ntile, nsplit), PK_TRIP) * PK_TRIP;      // whole trips of the K loop (pgemm_kernel)
    p.nsplit = (int)cdiv(p.ntile, p.kt_per);
    p.ws = (float*)ws;
    const long long blocks = tiles * p.nsplit;
    if (blocks <= 0 || blocks > 0x7fffffff) return MOGAN_ERR_SHAPE;
    const double flops = 2.0 * (double)p.M * (double)p.N * p.ncls * (double)p.K;
    mogan_prof_begin(prof_mode, best, flops, p.M, p.N * p.ncls, p.K, st);
    switch (best) {
        case 1: launch_pk<1, 4, 2>((unsigned)blocks, st, p); break;
        case 2: launch_pk<2, 2, 2>((unsigned)blocks, st, p); break;
        default: launch_pk<1, 2, PK_OCC12>((unsigned)blocks, st, p); break;
    }
    mogan_prof_end(1, st);
    if (p.nsplit > 1 && !keep_slabs) mogan_splitk_reduce_dense((const float*)ws, p.C, c_numel, p.nsplit, p.accumulate, st);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

static int apack(const float* x, unsigned char* P, int B, int C, int HW, hipStream_t st) {
    const long long Q = (long long)B * HW;
    hipLaunchKernelGGL(apack_kernel, dim3((unsigned)cdiv(Q, 64), (unsigned)(C / 32)), dim3(256), 0, st, x, P, B, C, HW);
    return 0;
}

}  // namespace

extern "C" {

int mogan_pk_debug_force(int take_all, int cfg, int split) {
    g_pk_force = take_all; g_pk_cfg = (cfg >= 0 && cfg < NPK) ? cfg : -1; g_pk_split = split > 0 ? split : 0;
    return 0;
}

int mogan_pk_conv_eligible(int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int dgrad) {
    PkGeom g;
    if (B <= 0 || Hs <= 0 || Ws <= 0 || !pk_geom(Cout, Cin, KH, KW, stride, dgrad, g)) return 0;
    const int OH = (Hs + 2 * ph - KH) / stride + 1, OW = (Ws + 2 * pw - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return 0;
    if (dgrad && (Hs % stride || Ws % stride)) return 0;          // every parity class has the same Hs/s x Ws/s grid
    // 32-bit byte offsets into the pixel panel
    const long long pix = dgrad ? (long long)B * OH * OW : (long long)B * Hs * Ws;
    if (pix * g.Cc * 6 >= (1ll << 32)) return 0;
    if (g_pk_force) return 1;
    // weight-heavy GEMMs only: few columns per class, long K, enough rows for the 128/256-row tiles.  Wider layers keep the
    // direct / Winograd / gather kernels, where packing the activations would cost more than it saves.
    const int rows_hw = dgrad ? (Hs / stride) * (Ws / stride) : OH * OW;
    return rows_hw <= 64 && g.K >= 1024 && g.M >= 128;
}

size_t mogan_pk_weight_bytes(int Cout, int Cin, int KH, int KW, int stride, int dgrad) {
    PkGeom g;
    if (!pk_geom(Cout, Cin, KH, KW, stride, dgrad, g)) return 0;
    return (size_t)g.cls_bytes * g.ncls;
}

int mogan_pk_weight_pack(const float* w, void* wpk, int Cout, int Cin, int KH, int KW, int stride, int ph, int pw, int dgrad,
                         hipStream_t stream) {
    PkGeom g;
    if (!w || !wpk || !pk_geom(Cout, Cin, KH, KW, stride, dgrad, g)) return MOGAN_ERR_SHAPE;
    WpackP p{};
    p.w = w; p.A = (unsigned char*)wpk; p.Cout = Cout; p.Cin = Cin; p.KH = KH; p.KW = KW; p.s = dgrad ? stride : 1;
    p.ph = ph; p.pw = pw; p.dgrad = dgrad; p.M = g.M; p.Cc = g.Cc; p.Mt = g.Mt; p.KS = g.KS; p.cls_stride = g.cls_bytes;
    if (g.Cc % 16) return MOGAN_ERR_SHAPE;
    const size_t lds = (size_t)KH * KW * 32 * 17 * sizeof(float);
    hipLaunchKernelGGL(wpack_kernel, dim3((unsigned)g.Mt, (unsigned)(g.Cc / 16)), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

}  // extern "C" (helpers below)

namespace {
// forward GEMM from a ready pixel panel; slabs stay in ws when keep_slabs (see run_pk)
static int pk_fwd_from_panel(const void* panel, const void* wpk, float* y, int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW,
                             int stride, int ph, int pw, void* ws, size_t ws_bytes, hipStream_t stream, bool keep_slabs,
                             int* nsplit_out, int accumulate = 0, int prof_mode = 7) {
    PkGeom g;
    if (!pk_geom(Cout, Cin, KH, KW, stride, 0, g) || B <= 0) return MOGAN_ERR_SHAPE;
    const int OH = (Hs + 2 * ph - KH) / stride + 1, OW = (Ws + 2 * pw - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return MOGAN_ERR_SHAPE;
    const size_t pbytes = (size_t)B * Hs * Ws * Cin * 6;
    if (pbytes >= (1ull << 32) || (long long)B * Cout * OH * OW >= (1ll << 31)) return MOGAN_ERR_SHAPE;
    PkP p{};
    p.A = (const unsigned char*)wpk; p.P = (const unsigned char*)panel; p.C = y;
    p.M = Cout; p.N = B * OH * OW; p.K = g.K; p.Mt = g.Mt; p.KS = g.KS; p.a_cls_stride = g.cls_bytes;
    p.a_bytes = (unsigned)g.cls_bytes; p.p_bytes = (unsigned)pbytes; p.ntile = g.K / 32; p.slab = (long long)B * Cout * OH * OW;
    p.accumulate = accumulate; p.dgrad = 0; p.ncls = 1; p.Cc = Cin; p.CG = Cin / 32; p.CGp = Cin / 32; p.cg0 = 0; p.PH = Hs; p.PW = Ws; p.RH = OH; p.RW = OW;
    p.s = stride; p.ph = ph; p.pw = pw; p.nkw = KW; p.outH = OH; p.outW = OW;
    const int rc = run_pk(p, ws, ws_bytes, prof_mode, stream, keep_slabs);
    if (nsplit_out) *nsplit_out = p.nsplit;
    return rc;
}
static int pk_dgrad_from_panel(const void* panel, const void* wpk, float* dx, int B, int Cin, int Hs, int Ws, int Cout, int KH,
                               int KW, int stride, int ph, int pw, void* ws, size_t ws_bytes, hipStream_t stream) {
    PkGeom g;
    if (!pk_geom(Cout, Cin, KH, KW, stride, 1, g) || B <= 0 || Hs % stride || Ws % stride) return MOGAN_ERR_SHAPE;
    const int OH = (Hs + 2 * ph - KH) / stride + 1, OW = (Ws + 2 * pw - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return MOGAN_ERR_SHAPE;
    const size_t pbytes = (size_t)B * OH * OW * Cout * 6;
    if (pbytes >= (1ull << 32) || (long long)B * Cin * Hs * Ws >= (1ll << 31)) return MOGAN_ERR_SHAPE;
    PkP p{};
    p.A = (const unsigned char*)wpk; p.P = (const unsigned char*)panel; p.C = dx;
    p.M = Cin; p.N = B * (Hs / stride) * (Ws / stride); p.K = g.K; p.Mt = g.Mt; p.KS = g.KS; p.a_cls_stride = g.cls_bytes;
    p.a_bytes = (unsigned)g.cls_bytes; p.p_bytes = (unsigned)pbytes; p.ntile = g.K / 32; p.slab = (long long)B * Cin * Hs * Ws;
    p.accumulate = 0; p.dgrad = 1; p.ncls = stride * stride; p.Cc = Cout; p.CG = Cout / 32; p.CGp = Cout / 32; p.cg0 = 0; p.PH = OH; p.PW = OW;
    p.RH = Hs / stride; p.RW = Ws / stride; p.s = stride; p.ph = ph; p.pw = pw; p.nkw = g.nkw; p.outH = Hs; p.outW = Ws;
    return run_pk(p, ws, ws_bytes, 8, stream);
}
}  // namespace

extern "C" {

int mogan_conv2d_fwd_pk(const float* x, const void* wpk, float* y, int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW,
                        int stride, int ph, int pw, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (B <= 0 || Cin <= 0 || Cin % 32 || Hs <= 0 || Ws <= 0) return MOGAN_ERR_SHAPE;
    const size_t pbytes = (size_t)B * Hs * Ws * Cin * 6;
    if (!ws || ws_bytes < up256(pbytes)) return MOGAN_ERR_WS;
    apack(x, (unsigned char*)ws, B, Cin, Hs * Ws, stream);
    return pk_fwd_from_panel(ws, wpk, y, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, (char*)ws + up256(pbytes),
                             ws_bytes - up256(pbytes), stream, false, nullptr);
}

int mogan_conv2d_dgrad_pk(const float* dy, const void* wpk, float* dx, int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW,
                          int stride, int ph, int pw, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (B <= 0 || Cout <= 0 || Cout % 32 || stride <= 0) return MOGAN_ERR_SHAPE;
    const int OH = (Hs + 2 * ph - KH) / stride + 1, OW = (Ws + 2 * pw - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return MOGAN_ERR_SHAPE;
    const size_t pbytes = (size_t)B * OH * OW * Cout * 6;
    if (!ws || ws_bytes < up256(pbytes)) return MOGAN_ERR_WS;
    apack(dy, (unsigned char*)ws, B, Cout, OH * OW, stream);
    return pk_dgrad_from_panel(ws, wpk, dx, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, (char*)ws + up256(pbytes),
                               ws_bytes - up256(pbytes), stream);
}

int mogan_pk_weight_pack_both(const float* w, void* wpk_fwd, void* wpk_dgrad, int Cout, int Cin, int KH, int KW, int stride, int ph,
                              int pw, hipStream_t stream) {
    PkGeom gf, gd;
    if (!w || !wpk_fwd || !wpk_dgrad || !pk_geom(Cout, Cin, KH, KW, stride, 0, gf) || !pk_geom(Cout, Cin, KH, KW, stride, 1, gd))
        return MOGAN_ERR_SHAPE;
    if (Cin % 32) return MOGAN_ERR_SHAPE;                 // (rows of the data-gradient panels: whole m-tiles)
    Wpack2P p{};
    p.w = w; p.Af = (unsigned char*)wpk_fwd; p.Ad = (unsigned char*)wpk_dgrad; p.Cout = Cout; p.Cin = Cin; p.KW = KW; p.s = stride;
    p.ph = ph; p.pw = pw; p.KSf = gf.KS; p.KSd = gd.KS; p.cls_stride = gd.cls_bytes;
    const dim3 grid((unsigned)cdiv(Cout, 32), (unsigned)(Cin / 16));
    switch (KH * KW) {
        case 16: hipLaunchKernelGGL(wpack2_kernel<16>, grid, dim3(256), 0, stream, p); break;
        case 9: hipLaunchKernelGGL(wpack2_kernel<9>, grid, dim3(256), 0, stream, p); break;
        case 1: hipLaunchKernelGGL(wpack2_kernel<1>, grid, dim3(256), 0, stream, p); break;
        default: {
            int rc = mogan_pk_weight_pack(w, wpk_fwd, Cout, Cin, KH, KW, stride, ph, pw, 0, stream);
            return rc ? rc : mogan_pk_weight_pack(w, wpk_dgrad, Cout, Cin, KH, KW, stride, ph, pw, 1, stream);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

static bool wgrad_pk_sizes(int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int& OH, int& OW,
                           int& Kpad, long long& R, size_t& abytes, size_t& pbytes) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Hs <= 0 || Ws <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return false;
    OH = (Hs + 2 * ph - KH) / stride + 1; OW = (Ws + 2 * pw - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return false;
    const long long npix = (long long)B * OH * OW;
    if (npix > (1 << 20)) return false;
    Kpad = (int)((npix + 31) / 32 * 32);
    R = (long long)Cin * KH * KW;
    abytes = up256((size_t)((Cout + 31) / 32) * 32 * Kpad * 6);
    pbytes = up256((size_t)R * Kpad * 6);
    return pbytes < (1ull << 32) && (long long)Cout * R < (1ll << 31);
}

int mogan_pk_wgrad_eligible(int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, size_t ws_bytes) {
    int OH, OW, Kpad; long long R; size_t ab, pb;
    if (!wgrad_pk_sizes(B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, OH, OW, Kpad, R, ab, pb)) return 0;
    if (ab + pb > ws_bytes) return 0;
    if (g_pk_force) return 1;
    // measured (tools/time_pk_wgrad.py): ahead of the implicit-GEMM kernel only where dW is large (>= 16M elements: the epilogue's
    // read-modify-write is most of the work and the 128x64 tiles stream it best) and K pays for the two packs
    return OH * OW <= 64 && Kpad >= 512 && (long long)Cout * R >= (16ll << 20);
}

int mogan_conv2d_wgrad_pk(const float* dy, const float* x, float* dw, int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW,
                          int stride, int ph, int pw, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    int OH, OW, Kpad; long long R; size_t ab, pb;
    if (!dy || !x || !dw || !wgrad_pk_sizes(B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, OH, OW, Kpad, R, ab, pb)) return MOGAN_ERR_SHAPE;
    if (!ws || ab + pb > ws_bytes) return MOGAN_ERR_WS;
    unsigned char* A = (unsigned char*)ws;
    unsigned char* P = A + ab;
    const int Mt = (Cout + 31) / 32, KS = Kpad / 16, npix = B * OH * OW;
    hipLaunchKernelGGL(dypack_kernel, dim3((unsigned)cdiv((long long)Mt * KS * 64, 256)), dim3(256), 0, stream, dy, A, Cout, OH * OW,
                       npix, KS);
    XtP q{};
    q.x = x; q.P = P; q.Cin = Cin; q.Hs = Hs; q.Ws = Ws; q.KH = KH; q.KW = KW; q.s = stride; q.ph = ph; q.pw = pw; q.OH = OH; q.OW = OW;
    q.Npix = npix; q.KG = Kpad / 32; q.total = R * (Kpad / 8);
    hipLaunchKernelGGL(xtpack_kernel, dim3((unsigned)cdiv(q.total, 256)), dim3(256), 0, stream, q);
    // dW (Cout x R) = "1x1 convolution" of the R-pixel, Kpad-channel panel with the Cout x Kpad "filters" dY
    return pk_fwd_from_panel(P, A, dw, 1, Kpad, (int)R, 1, Cout, 1, 1, 1, 0, 0, (char*)ws + ab + pb, ws_bytes - ab - pb, stream, false,
                             nullptr, accumulate, 9);
}

int mogan_pk_group(int n, MoganPkArgs* args, hipStream_t stream) {
    constexpr int env_tile = -1;
    g_pk_group_tile = env_tile;
    if (n <= 0 || n > PK_MAXG || !args) return MOGAN_ERR_SHAPE;
    PkGroup g{};
    g.n = n;
    long long end = 0; double flops = 0;
    // tile of the group: 128 x 64 (four waves down the rows) or 64 x 128 (2 x 2 waves), whichever pads less; the wide-row tile stages
    // half as many pixel-panel bytes per MFMA and is preferred when both pad alike
    double w128 = 0, w64 = 0;
    for (int i = 0; i < n; ++i) {
        const double N = (double)args[i].B * args[i].outH * args[i].outW, K = (double)args[i].KH * args[i].KW * args[i].Cp;
        w128 += (double)cdiv(args[i].M, 128) * 128 * (double)cdiv((long long)N, 64) * 64 * K;
        w64 += (double)cdiv(args[i].M, 64) * 64 * (double)cdiv((long long)N, 128) * 128 * K;
    }
    // measured on the trunk (B = 16, per launch): the 64-row tile is 15-45 % slower wherever the 128-row tile pads < 40 % and no
    // faster where it pads 50 % (the launches are bound by the pixel-panel gather, not by the MFMAs) -> lab switch only
    (void)w64; (void)w128;
    const int tile64 = g_pk_group_tile > 0 ? 1 : 0;
    const int bm = tile64 ? 64 : 128, bn = tile64 ? 128 : 64;
    for (int i = 0; i < n; ++i) {
        const MoganPkArgs& a = args[i];
        if (!a.wpk || !a.panel || !a.raw || a.B <= 0 || a.M <= 0 || a.Cp <= 0 || a.Cp % 32 || a.cg0 < 0 || a.cg0 + a.Cp / 32 > a.CGp ||
            a.KH <= 0 || a.KW <= 0 || a.KH * a.KW > 25 || a.stride <= 0 || (a.dgrad && a.stride != 1) || a.PH <= 0 || a.PW <= 0 ||
            a.outH <= 0 || a.outW <= 0 || a.nsplit <= 0)
            return MOGAN_ERR_SHAPE;
        const size_t pbytes = (size_t)a.B * a.PH * a.PW * a.CGp * 192;
        const long long slab = (long long)a.B * a.M * a.outH * a.outW;
        PkP& p = g.p[i];
        p.A = (const unsigned char*)a.wpk; p.P = (const unsigned char*)a.panel; p.C = a.raw; p.ws = a.raw;
        p.M = a.M; p.N = a.B * a.outH * a.outW; p.K = a.KH * a.KW * a.Cp; p.Mt = (int)cdiv(p.M, 32); p.KS = p.K / 16;
        const unsigned long long abytes = (unsigned long long)p.Mt * p.KS * 3072ull;
        if (pbytes >= (1ull << 32) || abytes >= (1ull << 32) || slab >= (1ll << 31)) return MOGAN_ERR_SHAPE;
        p.a_cls_stride = abytes; p.a_bytes = (unsigned)abytes; p.p_bytes = (unsigned)pbytes; p.ntile = p.K / 32; p.slab = slab;
        p.accumulate = 0; p.dgrad = a.dgrad; p.ncls = 1; p.Cc = a.Cp; p.CG = a.Cp / 32; p.CGp = a.CGp; p.cg0 = a.cg0;
        p.PH = a.PH; p.PW = a.PW; p.RH = a.outH; p.RW = a.outW; p.s = a.stride; p.ph = a.ph; p.pw = a.pw; p.nkw = a.KW;
        p.outH = a.outH; p.outW = a.outW;
        p.gx = (int)cdiv(p.N, bn); p.gy = (int)cdiv(p.M, bm);
        p.kt_per = (int)cdiv(cdiv(p.ntile, std::min(a.nsplit, p.ntile)), PK_TRIP) * PK_TRIP;
        while (cdiv(p.ntile, p.kt_per) > a.nsplit) p.kt_per += PK_TRIP;            // never more slabs than the caller made room for
        p.nsplit = (int)cdiv(p.ntile, p.kt_per);
        args[i].nsplit = p.nsplit;
        end += (long long)p.gx * p.gy * p.nsplit;
        if (end > 0x7fffffff) return MOGAN_ERR_SHAPE;
        g.end[i] = (unsigned)end;
        flops += 2.0 * (double)p.M * (double)p.N * (double)p.K;
    }
    mogan_prof_begin(a_prof_mode(args[0].dgrad), tile64, flops, g.p[0].M, g.p[0].N, g.p[0].K, stream);
    if (tile64) hipLaunchKernelGGL((pgemm_group_kernel<1, 2, PK_G64_OCC, 2>), dim3((unsigned)end), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((pgemm_group_kernel<1, 2, 3, 4>), dim3((unsigned)end), dim3(256), 0, stream, g);
    mogan_prof_end(1, stream);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

size_t mogan_pk_panel_bytes(int B, int C, int HW) { return (B > 0 && C > 0 && HW > 0) ? (size_t)B * HW * C * 6 : 0; }

int mogan_deep_block_eligible(int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int act,
                              int groups) {
    if (groups < 1 || B <= 0 || B % groups) return 0;
    if (!mogan_pk_conv_eligible(B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, 0)) return 0;
    if (!mogan_pk_conv_eligible(B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, 1)) return 0;
    const int OH = (Hs + 2 * ph - KH) / stride + 1, OW = (Ws + 2 * pw - KW) / stride + 1;
    return tail_ok(B / groups, Cout, OH * OW, act, groups) ? 1 : 0;
}

// B = all images of the tensor; groups > 1: `groups` BatchNorm calls on consecutive B / groups images each (stats: [2][groups][Cout])
int mogan_deep_conv_bn_act_fwd(const float* x, const void* xpanel, const void* wpk, const float* gamma, const float* beta,
                               float* rmean, float* rvar, float* y, float* stats, float* z, void* zpanel, int B, int Cin, int Hs,
                               int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, float eps, float momentum, int act,
                               float slope, int groups, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (B <= 0 || groups < 1 || B % groups || Cin <= 0 || Cin % 32 || Hs <= 0 || Ws <= 0 || stride <= 0) return MOGAN_ERR_SHAPE;
    const int OH = (Hs + 2 * ph - KH) / stride + 1, OW = (Ws + 2 * pw - KW) / stride + 1;
    if (OH <= 0 || OW <= 0 || !tail_ok(B / groups, Cout, OH * OW, act, groups) || !gamma || !beta || !y || !stats || !z)
        return MOGAN_ERR_SHAPE;
    char* w0 = (char*)ws; size_t wn = ws_bytes;
    if (!xpanel) {                       // the producer of x handed over no panel: pack it here
        const size_t pbytes = up256((size_t)B * Hs * Ws * Cin * 6);
        if (!ws || ws_bytes < pbytes) return MOGAN_ERR_WS;
        apack(x, (unsigned char*)ws, B, Cin, Hs * Ws, stream);
        xpanel = ws; w0 += pbytes; wn -= pbytes;
    }
    int nsplit = 1;
    int rc = pk_fwd_from_panel(xpanel, wpk, y, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, w0, wn, stream, true, &nsplit);
    if (rc) return rc;
    TailP t{};
    t.src = nsplit > 1 ? (const float*)w0 : y; t.nsplit = nsplit; t.slab = (long long)B * Cout * OH * OW;
    t.gamma = gamma; t.beta = beta; t.rmean = rmean; t.rvar = rvar; t.y = y; t.mean = stats; t.invstd = stats + (size_t)groups * Cout;
    t.z = z; t.zpanel = (unsigned char*)zpanel; t.B = B / groups; t.G = groups; t.C = Cout; t.HW = OH * OW; t.eps = eps;
    t.momentum = momentum; t.slope = slope;
    if (act == MOGAN_ACT_LRELU) launch_tail_fwd<MOGAN_ACT_LRELU>(t, stream);
    else if (act == MOGAN_ACT_RELU) launch_tail_fwd<MOGAN_ACT_RELU>(t, stream);
    else launch_tail_fwd<MOGAN_ACT_NONE>(t, stream);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

int mogan_deep_conv_bn_act_bwd(const float* dz, const float* y, const float* stats, const float* gamma, const float* beta,
                               const void* wpk_dgrad, float* dy, float* dgamma, float* dbeta, int accumulate, float* dx, int B,
                               int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int act, float slope,
                               int groups, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (B <= 0 || stride <= 0 || groups < 1 || B % groups) return MOGAN_ERR_SHAPE;
    const int OH = (Hs + 2 * ph - KH) / stride + 1, OW = (Ws + 2 * pw - KW) / stride + 1;
    if (OH <= 0 || OW <= 0 || !tail_ok(B / groups, Cout, OH * OW, act, groups) || !dz || !y || !stats || !gamma || !beta || !dy)
        return MOGAN_ERR_SHAPE;
    const size_t pbytes = up256((size_t)B * OH * OW * Cout * 6);
    const bool want_dx = dx != nullptr;
    if (want_dx && (!ws || ws_bytes < pbytes || !wpk_dgrad)) return MOGAN_ERR_WS;
    TailBwdP t{};
    t.dz = dz; t.y = y; t.mean = stats; t.invstd = stats + (size_t)groups * Cout; t.gamma = gamma; t.beta = beta; t.dy = dy;
    t.dypanel = want_dx ? (unsigned char*)ws : nullptr; t.dgamma = dgamma; t.dbeta = dbeta; t.accumulate = accumulate;
    t.B = B / groups; t.G = groups; t.C = Cout; t.HW = OH * OW; t.slope = slope;
    if (act == MOGAN_ACT_LRELU) launch_tail_bwd<MOGAN_ACT_LRELU>(t, stream);
    else if (act == MOGAN_ACT_RELU) launch_tail_bwd<MOGAN_ACT_RELU>(t, stream);
    else launch_tail_bwd<MOGAN_ACT_NONE>(t, stream);
    if (!want_dx) return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
    return pk_dgrad_from_panel(ws, wpk_dgrad, dx, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, (char*)ws + pbytes,
                               ws_bytes - pbytes, stream);
}

}  // extern "C"
