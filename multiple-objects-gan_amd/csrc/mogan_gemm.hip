// mogan_gemm.hip -- fp32 implicit GEMM on the matrix cores of gfx950 (CDNA4).
//
// One LDS-tiled kernel, four operand "gather modes":
//   CONV_FWD    Y[n,co,oy,ox]  = sum_{ci,kh,kw} W[co,ci,kh,kw] * X[n,ci,oy*s-p+kh,ox*s-p+kw]
//               GEMM: M=Cout, N=B*OH*OW, K=Cin*KH*KW   (optional fused nearest-x2 upsample of X)
//   CONV_DGRAD  dX[n,ci,y,x]   = sum_{co,kh,kw} W[co,ci,kh,kw] * dY[n,co,(y+p-kh)/s,(x+p-kw)/s]
//               one GEMM per stride-parity class (blockIdx.z): M=Cin, N=B*Hc*Wc, K=Cout*ceil(KH/s)*ceil(KW/s)
//   CONV_WGRAD  dW[co,ci,kh,kw] = sum_{n,oy,ox} dY[n,co,oy,ox] * X[n,ci,oy*s-p+kh,ox*s-p+kw]
//               GEMM: M=Cout, N=Cin*KH*KW, K=B*OH*OW
//   BMM         generic strided batched GEMM (Linear fwd/dgrad/wgrad, attention / DAMSM products)
//
// Math: fp32 in, fp32 accumulate.  Default build (MOGAN_X6 = 1, mogan_mma.h): every fp32 product is formed on the bf16 matrix
// pipe from the exact three-piece bf16 split of both operands -- six v_mfma_f32_32x32x16_bf16 partial products per 16 k, 6/16
// of the matrix-pipe time of the native v_mfma_f32_32x32x2_f32 form (-DMOGAN_X6=0), same error against fp64.  Block = 256
// threads = 4 wave64; each wave owns TMxTN tiles of 32x32; BK = 32.
//   split form   : the operands are split when a K-tile is staged; LDS holds the bf16 pieces (row = 3 x 64 B + 16 B pad, ONE
//                  buffer), fragments are 16-byte LDS reads.  While tile t's 2 x 6 MFMA groups run, tile t+2 is gathered into
//                  registers (first half) and tile t+1 is split and stored (second half); two barriers per tile.
//   native form  : As[m][k], Bs[n][k] fp32 with row stride 36 floats, double buffered, one barrier per tile, the gather of tile
//                  t+1 sliced between the 16 MFMA k-steps of tile t.
// Split-K writes per-split slabs into the caller's workspace; a second kernel reduces them in a fixed order (deterministic,
// no atomics).
//
// Replaces (reference = stock torch ops): nn.Conv2d / nn.Upsample+conv3x3 / nn.Linear / torch.bmm
// call sites of code/coco/attngan/model.py:35-55,364-380,598-611,664-680 and
// GlobalAttention.py:46,66,100,118.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "../../include/mogan_hip.h"
#include "mogan_internal.h"
#include "mogan_mma.h"
// split-bf16 form of gemm_kernel: 1 = the operands are split when they are staged (bf16 pieces in LDS), 0 = when the
// fragments are read (fp32 in LDS, the layout of the native form)
#ifndef MOGAN_X6_STAGE
#define MOGAN_X6_STAGE 1
#endif
#ifndef MOGAN_DGRAD_PARITY_FAST
#define MOGAN_DGRAD_PARITY_FAST 1
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FastDiv { uint32_t d, mul, shr; };

static FastDiv make_fd(uint32_t d) {
    FastDiv f; f.d = d ? d : 1; f.mul = 0; f.shr = 0;
    if (f.d > 1) {
        uint32_t l = 0; while ((1u << l) < f.d) ++l;          // ceil(log2 d)
        uint32_t p = 31 + l;
        f.mul = (uint32_t)(((1ull << p) + f.d - 1) / f.d);
        f.shr = p - 32;
    }
    return f;
}
// exact for 0 <= n < 2^31
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.shr);
}

enum { CONV_FWD = 0, CONV_DGRAD = 1, CONV_WGRAD = 2, BMM = 3 };

struct GemmP {
    const float* A; const float* B; float* C;
    int M, N, K;
    int kchunk, nsplit;
    long long slab;         // elements per split slab (workspace) when nsplit > 1
    float* ws;
    int accumulate;         // direct (nsplit==1) store adds to C
    // optional fused epilogue of CONV_FWD (frozen eval-mode BN + ReLU of the Inception trunk): y = act(scale[m]*y + shift[m])
    const float* ep_scale; const float* ep_shift; int ep_relu;
    float ep_slope;         // > 0 with ep_scale == nullptr: y = LeakyReLU_slope(conv) in the epilogue (mogan_conv2d_lrelu_fwd)
    // conv geometry: H,W = conv-input dims (after the optional fused upsample), Hs,Ws = stored dims
    int Bn, Cin, Cout, H, W, Hs, Ws, OH, OW, KH, KW, s, ph, pw, up;
    FastDiv fd_ohw, fd_ow, fd_khw, fd_kw, fd_nk, fd_nkw;
    int nkh, nkw;           // dgrad taps per parity class (max over classes)
    // bmm strides (elements)
    long long sAb, sAm, sAk, sBb, sBk, sBn, sCb, sCm, sCn;
    int a_lane_k, b_lane_n;
    // strided convolution I/O (channel slices of larger NCHW tensors): batch stride of the gathered operand (x / dY)
    // and of the output; CONV_FWD may send the output channels m >= msplit to a second destination; CONV_DGRAD may
    // zero the result where mask <= 0 (ReLU backward of the layer that produced the conv input) before accumulating
    unsigned xbs; long long ybs;
    float* C2; long long ybs2; int msplit;
    const float* mask; long long mbs;
    int avec;                    // 16-byte loads legal for the k-contiguous operand(s)
    unsigned a_bytes, b_bytes;   // extents of A and B for the buffer descriptors (< 4 GiB)
};

// branch-free guarded load: out-of-range byte offsets return 0 from the buffer unit (no exec-mask branches)
__device__ __forceinline__ float ldg(__amdgpu_buffer_rsrc_t r, unsigned idx, bool ok) {
    const unsigned off = ok ? idx * 4u : 0xFFFFFFFCu;
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

__device__ __forceinline__ f32x4 ldg4(__amdgpu_buffer_rsrc_t r, unsigned idx, bool ok) {
    const unsigned off = ok ? idx * 4u : 0xFFFFFFF0u;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

#define KH_BAD (0x4000 << 16)   // decode entry whose tap index is far out of range -> the bounds test fails

// LDS images: As[m][k], Bs[n][k], row stride LD = BK+4 floats (144 B): fragment reads are 4 x ds_read_b128
// per 32-row tile (bank-conflict free at this stride), staging writes are ds_write_b128 of 4 consecutive k.
// MFMA k assignment inside a K-tile: lane half h = lane>>5 owns k in [16h, 16h+16); step kk uses k = 16h+kk.
template <int MODE, int WM, int WN, int TM, int TN, bool AVEC>
__device__ __forceinline__ void gemm_block(const GemmP& p, const unsigned bx, const unsigned by, const unsigned bz) {
    constexpr bool SCHED = true;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 32, LD = BK + 4;
    constexpr int QA = BM / 32, QB = BN / 32;            // quads (4 consecutive k of one row) per thread
    static_assert(WM * WN == 4, "4 waves per block");
#if MOGAN_X6 && MOGAN_X6_STAGE
    // split-at-staging form: the LDS images hold the three bf16 pieces of every element, row = [piece 1: 32 k][piece 2]
    // [piece 3] + 16 B pad = 208 B (an odd multiple of 16 B: the 16-byte fragment reads of 16 consecutive rows touch all 64
    // banks once); ONE buffer, two barriers per K-tile (the fragments of a tile are in registers before its second half)
    constexpr int RSB = 3 * 2 * BK + 16;
    __shared__ __attribute__((aligned(16))) unsigned char Ab[BM * RSB];
    __shared__ __attribute__((aligned(16))) unsigned char Bb[BN * RSB];
#else
    __shared__ __attribute__((aligned(16))) float As[2][BM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LD];
#endif
    __shared__ __attribute__((aligned(16))) int Kt[2][3][BK];   // per-K-tile decode of k (conv fwd / dgrad)
    __shared__ __attribute__((aligned(16))) int Nt[2][BN];      // per-block decode of the column n (conv wgrad)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = by * BM, n0 = bx * BN;
    const int zb = bz / p.nsplit, sp = bz % p.nsplit;
    const int kbeg = sp * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, (short)0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, (short)0, (int)p.b_bytes, 0x00020000);
    const int OHW = p.OH * p.OW, HsWs = p.Hs * p.Ws, KHW = p.KH * p.KW;

    // staging maps.  k-fast: 8 lanes cover the 32 k of one row (memory contiguous along k);
    // row-fast: lanes run along the rows (memory contiguous along the rows: pixels)
    constexpr bool A_KFAST_C = (MODE == CONV_FWD || MODE == CONV_WGRAD);
    constexpr bool B_KFAST_C = (MODE == CONV_WGRAD);
    const bool a_kfast = (MODE == BMM) ? (p.a_lane_k != 0) : A_KFAST_C;
    const bool b_kfast = (MODE == BMM) ? (p.b_lane_n == 0) : B_KFAST_C;

    // ---- per-thread, tile-invariant decode -------------------------------------------------
    int Ncls = p.N;
    int py = 0, px = 0, Hc = 0, Wc = 0, kh0 = 0, kw0 = 0;
    if constexpr (MODE == CONV_DGRAD) {
        py = zb / p.s; px = zb % p.s;
        Hc = (p.H - py + p.s - 1) / p.s; Wc = (p.W - px + p.s - 1) / p.s;
        Ncls = p.Bn * Hc * Wc;
        if (n0 >= Ncls) return;     // block-uniform
        kh0 = (py + p.ph) % p.s; kw0 = (px + p.pw) % p.s;
    }
    const bool dg44 = MODE == CONV_DGRAD && p.s == 2 && p.KH == 4 && p.KW == 4 && (p.K & 3) == 0 &&
                      (((uintptr_t)p.A) & 15) == 0;
    // FWD / DGRAD: the QB rows (pixels) this thread stages are fixed: decode them once
    bool nvalid[QB]; unsigned nbase[QB]; int ny0[QB], nx0[QB];
    if constexpr (MODE == CONV_FWD) {
#pragma unroll
        for (int i = 0; i < QB; ++i) {
            const int n = n0 + (tid + 256 * i) % BN;
            nvalid[i] = n < p.N;
            const int nn = nvalid[i] ? n : 0;
            const int img = nn / OHW, pix = nn - img * OHW;
            const int oy = pix / p.OW, ox = pix - oy * p.OW;
            ny0[i] = oy * p.s - p.ph; nx0[i] = ox * p.s - p.pw;
            nbase[i] = (unsigned)img * p.xbs;
        }
    }
    if constexpr (MODE == CONV_DGRAD) {
#pragma unroll
        for (int i = 0; i < QB; ++i) {
            const int n = n0 + (tid + 256 * i) % BN;
            nvalid[i] = n < Ncls;
            const int nn = nvalid[i] ? n : 0;
            const int hw = Hc * Wc;
            const int img = nn / hw, rem = nn - img * hw;
            const int yc = rem / Wc, xc = rem - yc * Wc;
            nbase[i] = (unsigned)img * p.xbs;
            ny0[i] = (yc * p.s + py + p.ph - kh0) / p.s;      // exact
            nx0[i] = (xc * p.s + px + p.pw - kw0) / p.s;
        }
    }
    if constexpr (MODE == CONV_WGRAD) {
        for (int j = tid; j < BN; j += 256) {
            const int n = n0 + j;
            const uint32_t ci = fdiv(n, p.fd_khw);
            const uint32_t r = n - ci * KHW;
            const uint32_t kh = fdiv(r, p.fd_kw), kw = r - kh * p.KW;
            Nt[0][j] = ci * HsWs;
            Nt[1][j] = n < p.N ? (int)((kh << 16) | kw) : KH_BAD;
        }
    }

    // decode of one k of a K-tile into the LDS table (all 8 lane-groups write identical values)
    auto decode_k = [&](int kt, int buf) {
        if constexpr (MODE == CONV_FWD) {
            const int j = tid & 31, k = kt + j;
            const uint32_t ci = fdiv(k, p.fd_khw);
            const uint32_t r = k - ci * KHW;
            const uint32_t kh = fdiv(r, p.fd_kw), kw = r - kh * p.KW;
            Kt[buf][0][j] = ci * HsWs;
            Kt[buf][1][j] = k < kend ? (int)((kh << 16) | kw) : KH_BAD;
        } else if constexpr (MODE == CONV_DGRAD) {
            const int j = tid & 31, k = kt + j;
            const uint32_t co = fdiv(k, p.fd_nk);
            const uint32_t r = k - co * (p.nkh * p.nkw);
            const uint32_t khp = fdiv(r, p.fd_nkw), kwp = r - khp * p.nkw;
            const int kh = kh0 + khp * p.s, kw = kw0 + kwp * p.s;
            const bool ok = k < kend && kh < p.KH && kw < p.KW;
            Kt[buf][0][j] = co * OHW;
            Kt[buf][1][j] = ok ? (int)((khp << 16) | kwp) : KH_BAD;
            Kt[buf][2][j] = ok ? (int)(co * p.Cin * KHW + kh * p.KW + kw) : -1;
        }
    };

    float ra[QA][4], rb[QB][4];
    constexpr int NEA = QA * 4, NEB = QB * 4, NE = NEA + NEB;     // elements staged per thread per K-tile
    constexpr int NSL = MOGAN_X6 ? 12 : 16;                              // slices: the first NSL of the 16 k-steps carry
    constexpr int EPS = (NE + NSL - 1) / NSL;                     // the gather, the rest cover the load latency

    // per-quad / per-tile decode kept in registers between the element slices
    int4 qe0[QB], qe1[QB], qw[QA];
    unsigned wg_ab[4], wg_xb[4]; int wg_iy0[4], wg_ix0[4]; bool wg_ok[4];
    int wg_e0[QB], wg_e1[QB];

    // stage ONE element e of the next K-tile (kt) into ra/rb; e is a compile-time constant after unrolling
    auto stage_elem = [&](int e, int kt, int buf, float (&ra)[QA][4], float (&rb)[QB][4]) {
        const bool isA = e < NEA;
        const int i = isA ? e / 4 : (e - NEA) / 4, j = e & 3;
        if constexpr (MODE == CONV_WGRAD) {
            if (e == 0) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int k = kt + 4 * (tid & 7) + jj;
                    wg_ok[jj] = k < kend;
                    const uint32_t kk = wg_ok[jj] ? k : 0;
                    const uint32_t img = fdiv(kk, p.fd_ohw), pix = kk - img * OHW;
                    const uint32_t oy = fdiv(pix, p.fd_ow), ox = pix - oy * p.OW;
                    wg_iy0[jj] = (int)oy * p.s - p.ph; wg_ix0[jj] = (int)ox * p.s - p.pw;
                    wg_xb[jj] = img * p.Cin * HsWs;
                    wg_ab[jj] = img * p.Cout * OHW + pix;
                }
            }
        }
        if (isA) {
            if constexpr (MODE == CONV_FWD) {
                const int k = kt + 4 * (tid & 7) + j;
                const int m = m0 + (tid >> 3) + 32 * i;
                if constexpr (AVEC) {       // K % 4 == 0: the 4 consecutive k of the quad are one aligned 16-byte load
                    if (j == 0) {
                        const f32x4 q4 = ldg4(rA, (unsigned)m * p.K + k, m < p.M && k < kend);
                        ra[i][0] = q4[0]; ra[i][1] = q4[1]; ra[i][2] = q4[2]; ra[i][3] = q4[3];
                    }
                } else {
                    ra[i][j] = ldg(rA, (unsigned)m * p.K + k, m < p.M && k < kend);
                }
            } else if constexpr (MODE == CONV_DGRAD) {
                const int q = tid + 256 * i;
                const int m = m0 + q % BM;
                if (dg44) {
                    // 4x4 stride 2: the quad's four k are the taps (kh0 + 2a, kw0 + 2b) of ONE filter W[co][ci = m]: two
                    // 16-byte loads (kernel rows kh0 and kh0 + 2) and a pick of the columns kw0, kw0 + 2, instead of four
                    // dword loads that each touch a different 64-byte line per lane
                    if (j == 0) {
                        const int k = kt + 4 * (q / BM);
                        const bool ok = m < p.M && k < kend;
                        const unsigned base = ((unsigned)(k >> 2) * (unsigned)p.M + (unsigned)m) * 16u + (unsigned)kh0 * 4u;
                        const f32x4 r0 = ldg4(rA, base, ok), r1 = ldg4(rA, base + 8u, ok);
                        ra[i][0] = kw0 ? r0[1] : r0[0]; ra[i][1] = kw0 ? r0[3] : r0[2];
                        ra[i][2] = kw0 ? r1[1] : r1[0]; ra[i][3] = kw0 ? r1[3] : r1[2];
                    }
                } else {
                    if (j == 0) qw[i] = *(const int4*)&Kt[buf][2][4 * (q / BM)];
                    const int w = j == 0 ? qw[i].x : j == 1 ? qw[i].y : j == 2 ? qw[i].z : qw[i].w;
                    ra[i][j] = ldg(rA, (unsigned)w + (unsigned)m * KHW, m < p.M && w >= 0);
                }
            } else if constexpr (MODE == CONV_WGRAD) {
                const int m = m0 + (tid >> 3) + 32 * i;
                if constexpr (AVEC) {       // OH*OW % 4 == 0: the 4 pixels of the quad share (img, co) and are contiguous
                    if (j == 0) {
                        const f32x4 q4 = ldg4(rA, wg_ab[0] + (unsigned)m * OHW, wg_ok[0] && m < p.M);
                        ra[i][0] = q4[0]; ra[i][1] = q4[1]; ra[i][2] = q4[2]; ra[i][3] = q4[3];
                    }
                } else {
                    ra[i][j] = ldg(rA, wg_ab[j] + (unsigned)m * OHW, wg_ok[j] && m < p.M);
                }
            } else {
                const int q = tid + 256 * i;
                const int m = m0 + (a_kfast ? (q >> 3) : (q % BM));
                const int k = kt + 4 * (a_kfast ? (q & 7) : (q / BM)) + j;
                const unsigned idx = (unsigned)zb * (unsigned)p.sAb + (unsigned)m * (unsigned)p.sAm + (unsigned)k * (unsigned)p.sAk;
                if constexpr (AVEC) {       // host guarantees: both operands k-contiguous, 16-byte aligned rows, K % 4 == 0
                    if (j == 0) {
                        const f32x4 q4 = ldg4(rA, idx, m < p.M && k < kend);
                        ra[i][0] = q4[0]; ra[i][1] = q4[1]; ra[i][2] = q4[2]; ra[i][3] = q4[3];
                    }
                } else {
                    ra[i][j] = ldg(rA, idx, m < p.M && k < kend);
                }
            }
        } else {
            if constexpr (MODE == CONV_FWD || MODE == CONV_DGRAD) {
                if (j == 0) {
                    const int kq = (tid + 256 * i) / BN;
                    qe0[i] = *(const int4*)&Kt[buf][0][4 * kq];
                    qe1[i] = *(const int4*)&Kt[buf][1][4 * kq];
                }
                const int e0 = j == 0 ? qe0[i].x : j == 1 ? qe0[i].y : j == 2 ? qe0[i].z : qe0[i].w;
                const int e1 = j == 0 ? qe1[i].x : j == 1 ? qe1[i].y : j == 2 ? qe1[i].z : qe1[i].w;
                if constexpr (MODE == CONV_FWD) {
                    const int iy = ny0[i] + (e1 >> 16), ix = nx0[i] + (e1 & 0xFFFF);
                    const bool ok = nvalid[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    rb[i][j] = ldg(rB, nbase[i] + e0 + (iy >> p.up) * p.Ws + (ix >> p.up), ok);
                } else {
                    const int oy = ny0[i] - (e1 >> 16), ox = nx0[i] - (e1 & 0xFFFF);
                    const bool ok = nvalid[i] && (unsigned)oy < (unsigned)p.OH && (unsigned)ox < (unsigned)p.OW;
                    rb[i][j] = ldg(rB, nbase[i] + e0 + oy * p.OW + ox, ok);
                }
            } else if constexpr (MODE == CONV_WGRAD) {
                if (j == 0) { const int nl = (tid >> 3) + 32 * i; wg_e0[i] = Nt[0][nl]; wg_e1[i] = Nt[1][nl]; }
                const int iy = wg_iy0[j] + (wg_e1[i] >> 16), ix = wg_ix0[j] + (wg_e1[i] & 0xFFFF);
                const bool ok = wg_ok[j] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                rb[i][j] = ldg(rB, wg_xb[j] + wg_e0[i] + (iy >> p.up) * p.Ws + (ix >> p.up), ok);
            } else {
                const int q = tid + 256 * i;
                const int n = n0 + (b_kfast ? (q >> 3) : (q % BN));
                const int k = kt + 4 * (b_kfast ? (q & 7) : (q / BN)) + j;
                const unsigned idx = (unsigned)zb * (unsigned)p.sBb + (unsigned)k * (unsigned)p.sBk + (unsigned)n * (unsigned)p.sBn;
                if constexpr (AVEC) {
                    if (j == 0) {
                        const f32x4 q4 = ldg4(rB, idx, n < p.N && k < kend);
                        rb[i][0] = q4[0]; rb[i][1] = q4[1]; rb[i][2] = q4[2]; rb[i][3] = q4[3];
                    }
                } else {
                    rb[i][j] = ldg(rB, idx, n < p.N && k < kend);
                }
            }
        }
    };

#if MOGAN_X6 && MOGAN_X6_STAGE
    // quad i of a register set -> its three 8-byte piece groups in the LDS image
    auto store_quad = [&](int i, bool isA, const float (&v)[4]) {
        const int q = tid + 256 * i;
        const bool kf = isA ? a_kfast : b_kfast;
        const int R = isA ? BM : BN;
        const int row = kf ? (q >> 3) : (q % R), kq = kf ? (q & 7) : (q / R);
        uint32_t w[3][2];
        x6_split2(v[0], v[1], w[0][0], w[1][0], w[2][0]);
        x6_split2(v[2], v[3], w[0][1], w[1][1], w[2][1]);
        unsigned char* d = (isA ? Ab : Bb) + row * RSB + kq * 8;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *(uint2*)(d + pl * 2 * BK) = make_uint2(w[pl][0], w[pl][1]);
    };
#else
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < QA; ++i) {
            const int q = tid + 256 * i;
            const int row = a_kfast ? (q >> 3) : (q % BM), kq = a_kfast ? (q & 7) : (q / BM);
            *(float4*)&As[buf][row * LD + 4 * kq] = make_float4(ra[i][0], ra[i][1], ra[i][2], ra[i][3]);
        }
#pragma unroll
        for (int i = 0; i < QB; ++i) {
            const int q = tid + 256 * i;
            const int row = b_kfast ? (q >> 3) : (q % BN), kq = b_kfast ? (q & 7) : (q / BN);
            *(float4*)&Bs[buf][row * LD + 4 * kq] = make_float4(rb[i][0], rb[i][1], rb[i][2], rb[i][3]);
        }
    };

#endif
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#if MOGAN_X6 && MOGAN_X6_STAGE
    // ---- main loop, split-at-staging form.  Tile t is in LDS as bf16 pieces; while its 2 x 6 MFMA groups run, the gather of
    //      tile t+2 goes into one register set (first half: one slice per MFMA group) and tile t+1, gathered one iteration
    //      earlier into the other set, is split and written to LDS (second half, after the barrier that says every wave has
    //      its fragments of tile t).  Two tiles of load latency slack, the split arithmetic once per staged element instead
    //      of once per fragment element (half as many), no VALU between the LDS fragment reads and the MFMAs.
    const int arow = (wm * TM * 32 + (lane & 31)) * RSB + (lane >> 5) * 32;
    const int brow = (wn * TN * 32 + (lane & 31)) * RSB + (lane >> 5) * 32;
    const int ntile = (kend - kbeg + BK - 1) / BK;
    float ra2[QA][4], rb2[QB][4];
    constexpr int NQ = QA + QB;
    constexpr int EPG = (NE + 5) / 6, QPG = (NQ + 5) / 6;          // gather elements / stored quads per MFMA group
    auto frag = [&](const unsigned char* base, int off) {
        X6Frag f;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) f.p[pl] = __builtin_bit_cast(mma_bf16x8, *(const uint4*)(base + off + pl * 2 * BK));
        return f;
    };
    auto iter = [&](int t, float (&raC)[QA][4], float (&rbC)[QB][4], float (&raN)[QA][4], float (&rbN)[QB][4]) {
        const int kt = kbeg + t * BK, tb = t & 1;
        X6Frag fa[TM], fb[TN];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int q = 0; q < TM; ++q) fa[q] = frag(Ab, arow + q * 32 * RSB + s2 * 16);
#pragma unroll
            for (int q = 0; q < TN; ++q) fb[q] = frag(Bb, brow + q * 32 * RSB + s2 * 16);
            if (s2 == 1) __syncthreads();                          // every wave holds its fragments of tile t
#pragma unroll
            for (int term = 0; term < 6; ++term) {
#pragma unroll
                for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                    for (int tb2 = 0; tb2 < TN; ++tb2) acc[ta][tb2] = x6_mfma(fa[ta], fb[tb2], term, acc[ta][tb2]);
                if (s2 == 0) {
#pragma unroll
                    for (int x = 0; x < EPG; ++x)
                        if (term * EPG + x < NE) stage_elem(term * EPG + x, kt + 2 * BK, tb, raN, rbN);   // past kend: zeros
                } else {
#pragma unroll
                    for (int x = 0; x < QPG; ++x) {
                        const int qi = term * QPG + x;
                        if (qi < QA) store_quad(qi, true, raC[qi < QA ? qi : 0]);
                        else if (qi < NQ) store_quad(qi - QA, false, rbC[qi >= QA && qi < NQ ? qi - QA : 0]);
                    }
                    if (term == 5) decode_k(kt + 3 * BK, tb ^ 1);  // table tb^1 was last read while gathering tile t+1
                }
                if (SCHED) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                           // tile t+1 is in LDS
    };
    if (ntile > 0) {
        decode_k(kbeg, 0); decode_k(kbeg + BK, 1);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NE; ++e) stage_elem(e, kbeg, 0, ra, rb);
#pragma unroll
        for (int e = 0; e < NE; ++e) stage_elem(e, kbeg + BK, 1, ra2, rb2);
        __syncthreads();                                           // both tables read
        decode_k(kbeg + 2 * BK, 0);
#pragma unroll
        for (int i = 0; i < QA; ++i) store_quad(i, true, ra[i]);
#pragma unroll
        for (int i = 0; i < QB; ++i) store_quad(i, false, rb[i]);
        __syncthreads();
        for (int t = 0; t < ntile; t += 2) {
            iter(t, ra2, rb2, ra, rb);
            if (t + 1 < ntile) iter(t + 1, ra, rb, ra2, rb2);
        }
    }

#else
    const int arow = (wm * TM * 32 + (lane & 31)) * LD + (lane >> 5) * (BK / 2);
    const int brow = (wn * TN * 32 + (lane & 31)) * LD + (lane >> 5) * (BK / 2);

    // ---- main loop: one barrier per K-tile, LDS double buffered.  The gather of K-tile t+1 (address
    //      arithmetic + buffer loads) is cut into 16 slices, one behind the MFMAs of each k-step of tile t,
    //      fenced with sched_barrier: left to itself the compiler emits [whole gather][64 MFMAs][LDS writes],
    //      and the two waves of a SIMD then phase-lock and idle the matrix pipe while both gather
    //      (measured: SQ_WAIT_INST_ANY 77 %, MFMA busy 64 %). -------------------------------------------
    const int ntile = (kend - kbeg + BK - 1) / BK;
    if (ntile > 0) {
        decode_k(kbeg, 0);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NE; ++e) stage_elem(e, kbeg, 0, ra, rb);
        decode_k(kbeg + BK, 1);
        store_tile(0);
        __syncthreads();
        for (int t = 0; t < ntile; ++t) {
            const int cur = t & 1, kt = kbeg + t * BK;
            float4 af[TM][4], bf[TN][4];
#pragma unroll
            for (int q = 0; q < TM; ++q)
#pragma unroll
                for (int v = 0; v < 4; ++v) af[q][v] = *(const float4*)&As[cur][arow + q * 32 * LD + 4 * v];
#pragma unroll
            for (int q = 0; q < TN; ++q)
#pragma unroll
                for (int v = 0; v < 4; ++v) bf[q][v] = *(const float4*)&Bs[cur][brow + q * 32 * LD + 4 * v];
#if MOGAN_X6
            // fp32 product from three bf16 pieces per operand (x = x1 + x2 + x3 exactly, 8 significant bits each) and the
            // six partial products above 2^-24: v_mfma_f32_32x32x16_bf16 runs at 16x the fp32-MFMA rate, so 12 of them
            // replace 16 v_mfma_f32_32x32x2_f32 at 3/8 of the matrix-pipe time.  k assignment: step s uses the lane's
            // k = 16h + 8s .. +8 (the same 16 values the fp32 form walks through one at a time).
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                X6Frag a3[TM], b3[TN];
#pragma unroll
                for (int q = 0; q < TM; ++q) {
                    const float4 lo = af[q][2 * s2], hi = af[q][2 * s2 + 1];
                    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    a3[q] = x6_split8(x);
                }
#pragma unroll
                for (int q = 0; q < TN; ++q) {
                    const float4 lo = bf[q][2 * s2], hi = bf[q][2 * s2 + 1];
                    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    b3[q] = x6_split8(x);
                }
#pragma unroll
                for (int term = 0; term < 6; ++term) {
#pragma unroll
                    for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                        for (int tb = 0; tb < TN; ++tb) acc[ta][tb] = x6_mfma(a3[ta], b3[tb], term, acc[ta][tb]);
                    const int sl = s2 * 6 + term;
#pragma unroll
                    for (int x = 0; x < EPS; ++x)
                        if (sl * EPS + x < NE) stage_elem(sl * EPS + x, kt + BK, cur ^ 1, ra, rb);
                    if (sl == 11) decode_k(kt + 2 * BK, cur);
                    if (SCHED) __builtin_amdgcn_sched_barrier(0);
                }
            }
#else
#pragma unroll
            for (int s16 = 0; s16 < 16; ++s16) {
                const int v = s16 >> 2, c = s16 & 3;
#pragma unroll
                for (int ta = 0; ta < TM; ++ta) {
                    const float a = c == 0 ? af[ta][v].x : c == 1 ? af[ta][v].y : c == 2 ? af[ta][v].z : af[ta][v].w;
#pragma unroll
                    for (int tb = 0; tb < TN; ++tb) {
                        const float b = c == 0 ? bf[tb][v].x : c == 1 ? bf[tb][v].y : c == 2 ? bf[tb][v].z : bf[tb][v].w;
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ta][tb], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int x = 0; x < EPS; ++x)
                    if (s16 * EPS + x < NE) stage_elem(s16 * EPS + x, kt + BK, cur ^ 1, ra, rb);   // past kend: masked to 0
                if (s16 == 15) decode_k(kt + 2 * BK, cur);       // table `cur` was last read while staging tile t
                if (SCHED) __builtin_amdgcn_sched_barrier(0);    // (a sched_group_barrier template instead: no gain, lab 8)
            }
#endif
            store_tile(cur ^ 1);                                 // buffer cur^1 was last read in iteration t-1
            __syncthreads();
        }
    }

#endif
    // ---------------------------------------------------------------- epilogue
    const bool split = p.nsplit > 1;
    float* __restrict__ slab = p.ws + (size_t)sp * p.slab;
    const bool addc = !split && p.accumulate;
#pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
        const int n = n0 + wn * TN * 32 + tb * 32 + (lane & 31);
        // cden: dense (slab) position of column n, cstr / cstr2: position in the (possibly strided) destination(s)
        size_t cden = 0, cstr = 0, cstr2 = 0, mb = 0, cms = 0; bool nok;
        if constexpr (MODE == CONV_FWD) {
            nok = n < p.N;
            const int nn = nok ? n : 0;
            const int img = nn / OHW, pix = nn - img * OHW;
            cms = OHW;
            cden = (size_t)img * p.M * OHW + pix; cstr = (size_t)img * p.ybs + pix; cstr2 = (size_t)img * p.ybs2 + pix;
            mb = (size_t)img * p.mbs + pix;
        } else if constexpr (MODE == CONV_DGRAD) {
            nok = n < Ncls;
            const int nn = nok ? n : 0;
            const int hw = Hc * Wc;
            const int img = nn / hw, rem = nn - img * hw;
            const int yc = rem / Wc, xc = rem - yc * Wc;
            cms = (size_t)p.H * p.W;
            const size_t pos = (size_t)(yc * p.s + py) * p.W + (xc * p.s + px);
            cden = (size_t)img * p.M * cms + pos; cstr = (size_t)img * p.ybs + pos; mb = (size_t)img * p.mbs + pos;
        } else if constexpr (MODE == CONV_WGRAD) {
            nok = n < p.N; cden = cstr = n; cms = p.N;
        } else {
            nok = n < p.N; cden = cstr = (size_t)zb * p.sCb + (size_t)n * p.sCn; cms = p.sCm;
        }
#pragma unroll
        for (int ta = 0; ta < TM; ++ta) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (nok && m < p.M) {
                    float v = acc[ta][tb][r];
                    if (split) { slab[cden + (size_t)m * cms] = v; continue; }
                    float* dst = p.C + cstr + (size_t)m * cms;
                    if constexpr (MODE == CONV_FWD) {
                        if (m >= p.msplit) dst = p.C2 + cstr2 + (size_t)(m - p.msplit) * cms;
                    }
                    if constexpr (MODE == CONV_DGRAD || MODE == CONV_FWD) {
                        if (p.mask != nullptr && !(p.mask[mb + (size_t)m * cms] > 0.f)) v = 0.f;
                    }
                    if (addc) v += *dst;
                    if constexpr (MODE == CONV_FWD) {
                        if (p.ep_scale != nullptr) {
                            v = fmaf(v, p.ep_scale[m], p.ep_shift[m]);
                            if (p.ep_relu) v = fmaxf(v, 0.f);
                        } else if (p.ep_slope > 0.f) {
                            v = v > 0.f ? v : v * p.ep_slope;
                        }
                    }
                    *dst = v;
                }
            }
        }
    }
}

// XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so XCD k is given a
// CONTIGUOUS range of the tile sequence (n fastest): the blocks that share one A (weight) panel then hit the same L2
// instead of fetching it once per XCD.
__device__ __forceinline__ unsigned xcd_order(unsigned L, unsigned total) {
    const unsigned k = L & 7u, j = L >> 3, q = total >> 3, r = total & 7u;
    return k * q + (k < r ? k : r) + j;
}

template <int MODE, int WM, int WN, int TM, int TN, bool AVEC>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmP p) {
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const unsigned V = xcd_order(bx + gx * (by + gy * bz), total);
        bx = V % gx; const unsigned t2 = V / gx;
        const unsigned ncls = (MODE == CONV_DGRAD) ? (unsigned)(p.s * p.s) : 1u;
        if (MODE == CONV_DGRAD && ncls > 1 && MOGAN_DGRAD_PARITY_FAST) {
            // strided data gradient: the stride-parity classes of one (m-tile, K-chunk) read the SAME filter lines (each class
            // its own taps of every 64-byte W[co][ci] block): run them next to each other (class fastest after the n-tiles) so
            // that the lines come from the XCD's L2 instead of being fetched from HBM once per class
            const unsigned cls = t2 % ncls, t3 = t2 / ncls;
            by = t3 % gy;
            bz = cls * (unsigned)p.nsplit + t3 / gy;
        } else {
            by = t2 % gy; bz = t2 / gy;
        }
    }
    gemm_block<MODE, WM, WN, TM, TN, AVEC>(p, bx, by, bz);
}

// Several independent convolutions (same mode and tile shape, different operands / geometry) as ONE launch: the tile
// sequences of the problems are laid end to end over a 1-D grid.  The frozen Inception trunk's Mixed blocks consist of
// 2-4 small convolutions per dependency level (1-3 GFLOP each at B = 16): launched one by one each of them needs split-K
// (+ a reduction launch) to fill 256 CUs; laid side by side their tiles fill the chip without it.
constexpr int MAXG = 4;
struct GroupArgs { GemmP p[MAXG]; int tile_end[MAXG]; int gx[MAXG], gy[MAXG]; int nprob; };

template <int MODE, int WM, int WN, int TM, int TN, bool AVEC>
__global__ __launch_bounds__(256) void gemm_group_kernel(const GroupArgs g) {
    const unsigned V = xcd_order(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma unroll
    for (int i = 0; i < MAXG - 1; ++i) if (i + 1 < g.nprob && V >= (unsigned)g.tile_end[i]) pi = i + 1;
    const unsigned local = V - (pi ? (unsigned)g.tile_end[pi - 1] : 0u);
    const unsigned gx = g.gx[pi], gy = g.gy[pi];
    const unsigned bx = local % gx, t2 = local / gx;
    gemm_block<MODE, WM, WN, TM, TN, AVEC>(g.p[pi], bx, t2 % gy, t2 / gy);
}

// out[dst(i)] = (acc ? out[dst(i)] : 0) + sum_s ws[s*slab + i]: fixed summation order (deterministic).  One element per
// thread (small outputs still give hundreds of blocks), 8 independent slab loads in flight per thread.  The dense slab
// index i = (img, m, pos) is mapped to the (possibly strided / two-part) destination like the kernel's own epilogue.
struct ReduceP {
    long long n, slab, Mcms, cms, ybs, ybs2, mbs;
    int nsplit, acc, msplit, ep_relu;
    float ep_slope;
    const float* ep_scale; const float* ep_shift; const float* mask; float* out2;
};
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                            const ReduceP q) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= q.n) return;
    long long img, rem; int m;
    if (q.n < (1ll << 31)) {              // (uniform) 32-bit index arithmetic: two 64-bit divisions per element otherwise
        const unsigned iu = (unsigned)i, im = iu / (unsigned)q.Mcms, r = iu - im * (unsigned)q.Mcms;
        img = im; rem = r; m = (int)(r / (unsigned)q.cms);
    } else {
        img = i / q.Mcms; rem = i - img * q.Mcms; m = (int)(rem / q.cms);
    }
    float* dst = out + img * q.ybs + rem;
    if (m >= q.msplit) dst = q.out2 + img * q.ybs2 + (rem - (long long)q.msplit * q.cms);
    const float* p = ws + i;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= q.nsplit; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * q.slab];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < q.nsplit; ++k) s += p[(size_t)k * q.slab];
    if (q.mask != nullptr && !(q.mask[img * q.mbs + rem] > 0.f)) s = 0.f;
    if (q.acc) s += *dst;
    if (q.ep_scale != nullptr) {
        s = fmaf(s, q.ep_scale[m], q.ep_shift[m]);
        if (q.ep_relu) s = fmaxf(s, 0.f);
    } else if (q.ep_slope > 0.f) {
        s = s > 0.f ? s : s * q.ep_slope;
    }
    *dst = s;
}

// ------------------------------------------------------------------------------ host dispatch
struct Cfg { int wm, wn, tm, tn; };
static const Cfg kCfgs[] = {{2, 2, 2, 2}, {1, 4, 3, 1}, {4, 1, 1, 1}, {1, 4, 1, 1}, {2, 2, 1, 1}, {2, 2, 2, 1}, {2, 2, 1, 2}};
enum { NCFG = 7, NCFG_HEUR = 5 };     // 128x64 and 64x128 are only reached through the tuned table / the test hook

template <int MODE, bool AVEC>
static void launch_cfg2(int c, dim3 grid, hipStream_t st, const GemmP& p) {
    switch (c) {
        case 0: hipLaunchKernelGGL((gemm_kernel<MODE, 2, 2, 2, 2, AVEC>), grid, dim3(256), 0, st, p); break;
        case 1: hipLaunchKernelGGL((gemm_kernel<MODE, 1, 4, 3, 1, AVEC>), grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((gemm_kernel<MODE, 4, 1, 1, 1, AVEC>), grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL((gemm_kernel<MODE, 1, 4, 1, 1, AVEC>), grid, dim3(256), 0, st, p); break;
        case 5: hipLaunchKernelGGL((gemm_kernel<MODE, 2, 2, 2, 1, AVEC>), grid, dim3(256), 0, st, p); break;
        case 6: hipLaunchKernelGGL((gemm_kernel<MODE, 2, 2, 1, 2, AVEC>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_kernel<MODE, 2, 2, 1, 1, AVEC>), grid, dim3(256), 0, st, p); break;
    }
}
template <int MODE>
static void launch_cfg(int c, dim3 grid, hipStream_t st, const GemmP& p) {
    if constexpr (MODE == CONV_DGRAD) launch_cfg2<MODE, false>(c, grid, st, p);
    else if (p.avec) launch_cfg2<MODE, true>(c, grid, st, p);
    else launch_cfg2<MODE, false>(c, grid, st, p);
}

template <int MODE, bool AVEC>
static void launch_group2(int c, unsigned nblocks, hipStream_t st, const GroupArgs& g) {
    switch (c) {
        case 0: hipLaunchKernelGGL((gemm_group_kernel<MODE, 2, 2, 2, 2, AVEC>), dim3(nblocks), dim3(256), 0, st, g); break;
        case 5: hipLaunchKernelGGL((gemm_group_kernel<MODE, 2, 2, 2, 1, AVEC>), dim3(nblocks), dim3(256), 0, st, g); break;
        case 6: hipLaunchKernelGGL((gemm_group_kernel<MODE, 2, 2, 1, 2, AVEC>), dim3(nblocks), dim3(256), 0, st, g); break;
        default: hipLaunchKernelGGL((gemm_group_kernel<MODE, 2, 2, 1, 1, AVEC>), dim3(nblocks), dim3(256), 0, st, g); break;
    }
}

static inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

static int g_force_cfg = -1, g_force_split = 0;

// Opt-in per-launch timing (bench.py roofline leg): HIP events on the launch stream around every
// gemm_kernel launch + its algorithmic flops (2*M*N*K of the true, unpadded GEMM, all z-batches).
struct ProfRec { int mode, cfg; double flops; hipEvent_t e0, e1; int M, N, K, nz, nsplit, Cin, Cout, H, W, KH, KW, s, up, Bn; };
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;
static bool g_prof_on = false;

}  // namespace
void mogan_splitk_reduce_dense(const float* ws, float* out, long long n, int nsplit, int accumulate, hipStream_t st) {
    ReduceP q{};
    q.n = n; q.slab = n; q.nsplit = nsplit; q.acc = accumulate;
    q.Mcms = n; q.cms = n; q.ybs = n; q.msplit = 0x7fffffff;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, ws, out, q);
}
int mogan_use_dconv = 1;
static ProfRec g_prof_open; static bool g_prof_open_valid = false;
void mogan_prof_begin(int mode, int cfg, double flops, int M, int N, int K, hipStream_t st) {
    g_prof_open_valid = false;
    if (!g_prof_on) return;
    ProfRec r{}; r.mode = mode; r.cfg = cfg; r.flops = flops; r.M = M; r.N = N; r.K = K; r.nz = 1; r.nsplit = 1;
    hipEventCreate(&r.e0); hipEventCreate(&r.e1); hipEventRecord(r.e0, st);
    g_prof_open = r; g_prof_open_valid = true;
}
void mogan_prof_relabel(int cfg) { if (g_prof_open_valid) g_prof_open.cfg = cfg; }     // the open record belongs to another kernel
void mogan_prof_end(int taken, hipStream_t st) {       // !taken: the attempt fell through, drop the record
    if (!g_prof_open_valid) return;
    g_prof_open_valid = false;
    if (!taken) { hipEventDestroy(g_prof_open.e0); hipEventDestroy(g_prof_open.e1); return; }
    hipEventRecord(g_prof_open.e1, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(g_prof_open);
}
namespace {

// Tile config: least padded MFMA work; ties -> larger tile.  Split-K: fill >= 2 waves of 256 CUs
// when the tile grid alone cannot, keeping >= 128 of K per split.
// split-K aims at this many blocks per launch (mogan_gemm_set_split_target): 768 = three per CU when the kernel has the
// GPU to itself; a caller that keeps several streams busy asks for 384 (fewer slabs to reduce, the other branches fill the
// chip): +2 % on the multi-stream train step, -7 % on the kernel alone.
static std::atomic<int> g_split_target{768};        // process default
// per-stream override (mogan_stream_set_split_target): the owner of a stream knows whether its kernels have the GPU to
// themselves; keyed by the stream handle, read under a mutex (a few hundred lookups per step)
static std::unordered_map<hipStream_t, int> g_stream_target;
static std::mutex g_stream_mu;
static int split_target_of(hipStream_t st) {
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        auto it = g_stream_target.find(st);
        if (it != g_stream_target.end()) return it->second;
    }
    return g_split_target.load(std::memory_order_relaxed);
}

// tuned dispatch (mogan_gemm_tune_set): key = the GEMM as run_gemm sees it
struct TuneKey { int mode, M, N, K, nz; bool operator==(const TuneKey& o) const { return mode == o.mode && M == o.M && N == o.N && K == o.K && nz == o.nz; } };
struct TuneHash { size_t operator()(const TuneKey& k) const {
    size_t h = (size_t)k.mode * 1000003u; h = h * 31 + (size_t)k.M; h = h * 1000003u + (size_t)k.N; h = h * 31 + (size_t)k.K;
    return h * 7 + (size_t)k.nz; } };
static std::unordered_map<TuneKey, std::pair<int, int>, TuneHash> g_tuned;
static std::mutex g_tuned_mu;

}  // namespace
int mogan_split_target(hipStream_t st) { return split_target_of(st); }
namespace {

static int run_gemm(int mode, GemmP& p, int nz, long long c_numel, void* ws, size_t ws_bytes, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0) return 0;
    int best = 0; double bestw = 1e300;
    for (int c = 0; c < NCFG_HEUR; ++c) {
        const int bm = kCfgs[c].wm * kCfgs[c].tm * 32, bn = kCfgs[c].wn * kCfgs[c].tn * 32;
        double w = (double)cdiv(p.M, bm) * bm * (double)cdiv(p.N, bn) * bn;
        w *= (bm * bn >= 96 * 128) ? 1.0 : (bm * bn >= 64 * 64 ? 1.08 : 1.15);   // small tiles: less reuse
        if (w < bestw * 0.999) { bestw = w; best = c; }
    }
    int tuned_split = 0;
    if (g_force_cfg < 0 && g_force_split <= 0 && !g_tuned.empty()) {
        std::lock_guard<std::mutex> lk(g_tuned_mu);
        auto it = g_tuned.find(TuneKey{mode, p.M, p.N, p.K, nz});
        if (it != g_tuned.end()) { best = it->second.first; tuned_split = it->second.second; }
    }
    if (g_force_cfg >= 0) best = g_force_cfg;
    const int bm = kCfgs[best].wm * kCfgs[best].tm * 32, bn = kCfgs[best].wn * kCfgs[best].tn * 32;
    const long long tiles = cdiv(p.M, bm) * cdiv(p.N, bn) * nz;
    int nsplit = 1;
    const int ktiles = (int)cdiv(p.K, 32);
    const int split_target = split_target_of(st);
    if (tiles < 512 && ktiles >= 8) {             // fewer than two blocks per CU and a K loop worth cutting
        nsplit = (int)cdiv(split_target, tiles);
        nsplit = (int)std::min<long long>(nsplit, ktiles / 4);
        if (nsplit < 1) nsplit = 1;
    }
    if (tuned_split > 0) nsplit = std::min(tuned_split, std::max(1, ktiles));
    if (g_force_split > 0) nsplit = std::min(g_force_split, ktiles);
    if (nsplit > 1) {
        const long long fit = ws ? (long long)(ws_bytes / (sizeof(float) * (size_t)c_numel)) : 0;
        if (fit < 2) nsplit = 1; else nsplit = (int)std::min<long long>(nsplit, fit);
    }
    int kt_per = (int)cdiv(ktiles, nsplit);
    nsplit = (int)cdiv(ktiles, kt_per);
    p.kchunk = kt_per * 32; p.nsplit = nsplit; p.slab = c_numel; p.ws = (float*)ws;
    if (p.K <= 0) { p.kchunk = 32; p.nsplit = 1; }
    dim3 grid((unsigned)cdiv(p.N, bn), (unsigned)cdiv(p.M, bm), (unsigned)(nz * p.nsplit));
    if (grid.y > 65535 || grid.z > 65535) return MOGAN_ERR_SHAPE;
    ProfRec rec{}; const bool prof = g_prof_on;
    if (prof) {
        rec.mode = mode; rec.cfg = best;
        rec.M = p.M; rec.N = p.N; rec.K = p.K; rec.nz = nz; rec.nsplit = p.nsplit; rec.Cin = p.Cin; rec.Cout = p.Cout;
        rec.H = p.H; rec.W = p.W; rec.KH = p.KH; rec.KW = p.KW; rec.s = p.s; rec.up = p.up; rec.Bn = p.Bn;
        // DGRAD: p.N is the column count of parity class (0,0); all classes together cover B*H*W columns
        const double ncols = mode == CONV_DGRAD ? (double)p.Bn * p.H * p.W : (double)p.N * nz;
        rec.flops = 2.0 * (double)p.M * ncols * (double)p.K;
        hipEventCreate(&rec.e0); hipEventCreate(&rec.e1);
        hipEventRecord(rec.e0, st);
    }
    switch (mode) {
        case CONV_FWD: launch_cfg<CONV_FWD>(best, grid, st, p); break;
        case CONV_DGRAD: launch_cfg<CONV_DGRAD>(best, grid, st, p); break;
        case CONV_WGRAD: launch_cfg<CONV_WGRAD>(best, grid, st, p); break;
        default: launch_cfg<BMM>(best, grid, st, p); break;
    }
    if (prof) {
        hipEventRecord(rec.e1, st);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(rec);
    }
    if (p.nsplit > 1) {
        const long long nblk = cdiv(c_numel, 256);
        ReduceP q{};
        q.n = c_numel; q.slab = c_numel; q.nsplit = p.nsplit; q.acc = p.accumulate;
        q.Mcms = c_numel; q.cms = c_numel; q.ybs = c_numel; q.msplit = 0x7fffffff;       // dense: dst = out + i
        if (mode == CONV_FWD || mode == CONV_DGRAD) {
            q.cms = mode == CONV_FWD ? (long long)p.OH * p.OW : (long long)p.H * p.W;
            q.Mcms = (long long)p.M * q.cms; q.ybs = p.ybs; q.msplit = p.msplit; q.out2 = p.C2; q.ybs2 = p.ybs2;
            q.mask = p.mask; q.mbs = p.mbs;
            if (mode == CONV_FWD) { q.ep_scale = p.ep_scale; q.ep_shift = p.ep_shift; q.ep_relu = p.ep_relu; q.ep_slope = p.ep_slope; }
        }
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (const float*)ws, p.C, q);
    }
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

// One launch for up to MAXG prepared problems of one mode (no split-K).  Tile shape: among 128x128 / 128x64 / 64x128 / 64x64
// the one with the least padded work, small tiles and under-filled grids penalised.
// measured on the B = 16 train step (img/s by threshold): 384: 291.7, 800: 294.1, 1600: 298.1, never group: 297.1 -- the members
// launched singly use the tuned (tile, split-K) table, a group uses one generic tile shape and no split-K, so grouping only
// pays where every member is wide (the 35x35 maps of Mixed_5b-d: 19600 columns)
static int g_group_min_tiles = 1600;
static int run_group(int mode, GemmP* ps, const int* nz, int n, hipStream_t st) {
    static const int cand[4] = {0, 5, 6, 4};
    int best = 4; double bestw = 1e300;
    for (int ci = 0; ci < 4; ++ci) {
        const int c = cand[ci];
        const int bm = kCfgs[c].wm * kCfgs[c].tm * 32, bn = kCfgs[c].wn * kCfgs[c].tn * 32;
        double w = 0; long long tiles = 0;
        for (int i = 0; i < n; ++i) {
            const long long t = cdiv(ps[i].M, bm) * cdiv(ps[i].N, bn) * nz[i];
            tiles += t;
            w += (double)t * bm * bn * (double)ps[i].K;
        }
        w *= (bm * bn >= 128 * 128) ? 1.0 : (bm * bn >= 128 * 64 ? 1.04 : 1.10);
        if (tiles < 512) w *= 512.0 / (double)(tiles < 64 ? 64 : tiles);      // fewer than two blocks per CU
        if (w < bestw) { bestw = w; best = c; }
    }
    if (g_force_cfg == 0 || g_force_cfg == 4 || g_force_cfg == 5 || g_force_cfg == 6) best = g_force_cfg;
    const int bm = kCfgs[best].wm * kCfgs[best].tm * 32, bn = kCfgs[best].wn * kCfgs[best].tn * 32;
    {
        // a group that cannot fill the chip (the 8x8 maps of Mixed_7b/c: 1024 columns) is better served by its members one
        // by one, each with split-K: return 1 = "launch them singly"
        long long tiles = 0;
        for (int i = 0; i < n; ++i) tiles += cdiv(ps[i].M, bm) * cdiv(ps[i].N, bn) * nz[i];
        if (tiles < g_group_min_tiles && g_force_cfg < 0) return 1;
    }
    GroupArgs g{};
    g.nprob = n;
    long long end = 0; bool avec = true;
    for (int i = 0; i < n; ++i) {
        GemmP& p = ps[i];
        p.kchunk = (int)cdiv(p.K, 32) * 32; p.nsplit = 1; p.slab = 0; p.ws = nullptr;
        g.gx[i] = (int)cdiv(p.N, bn); g.gy[i] = (int)cdiv(p.M, bm);
        end += (long long)g.gx[i] * g.gy[i] * nz[i];
        g.tile_end[i] = (int)end;
        avec = avec && p.avec;
        g.p[i] = p;
    }
    if (end <= 0 || end > 0x7fffffff) return MOGAN_ERR_SHAPE;
    const bool prof = g_prof_on;
    ProfRec rec{};
    if (prof) {                      // one record for the group: summed flops, geometry of the first problem
        rec.mode = mode; rec.cfg = best; rec.M = ps[0].M; rec.N = ps[0].N; rec.K = ps[0].K; rec.nz = n; rec.nsplit = 1;
        rec.Cin = ps[0].Cin; rec.Cout = ps[0].Cout; rec.H = ps[0].H; rec.W = ps[0].W; rec.KH = ps[0].KH; rec.KW = ps[0].KW;
        rec.s = ps[0].s; rec.up = 0; rec.Bn = ps[0].Bn;
        for (int i = 0; i < n; ++i) {
            const double ncols = mode == CONV_DGRAD ? (double)ps[i].Bn * ps[i].H * ps[i].W : (double)ps[i].N * nz[i];
            rec.flops += 2.0 * (double)ps[i].M * ncols * (double)ps[i].K;
        }
        hipEventCreate(&rec.e0); hipEventCreate(&rec.e1);
        hipEventRecord(rec.e0, st);
    }
    if (mode == CONV_FWD) { if (avec) launch_group2<CONV_FWD, true>(best, (unsigned)end, st, g); else launch_group2<CONV_FWD, false>(best, (unsigned)end, st, g); }
    else launch_group2<CONV_DGRAD, false>(best, (unsigned)end, st, g);
    if (prof) {
        hipEventRecord(rec.e1, st);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(rec);
    }
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

static int conv_geom(GemmP& p, int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW, int s, int ph, int pw, int up) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Hs <= 0 || Ws <= 0 || KH <= 0 || KW <= 0 || s <= 0 || up < 0 || up > 1)
        return MOGAN_ERR_SHAPE;
    p.Bn = B; p.Cin = Cin; p.Cout = Cout; p.Hs = Hs; p.Ws = Ws; p.up = up;
    p.H = Hs << up; p.W = Ws << up; p.KH = KH; p.KW = KW; p.s = s; p.ph = ph; p.pw = pw;
    p.OH = (p.H + 2 * ph - KH) / s + 1; p.OW = (p.W + 2 * pw - KW) / s + 1;
    if (p.OH <= 0 || p.OW <= 0) return MOGAN_ERR_SHAPE;
    p.fd_ohw = make_fd(p.OH * p.OW); p.fd_ow = make_fd(p.OW);
    p.fd_khw = make_fd(KH * KW); p.fd_kw = make_fd(KW);
    p.nkh = (KH + s - 1) / s; p.nkw = (KW + s - 1) / s;
    p.fd_nk = make_fd(p.nkh * p.nkw); p.fd_nkw = make_fd(p.nkw);
    if ((long long)B * Cin * p.H * p.W >= (1ll << 30) || (long long)B * Cout * p.OH * p.OW >= (1ll << 30) ||
        (long long)Cout * Cin * KH * KW >= (1ll << 30))
        return MOGAN_ERR_SHAPE;
    return 0;
}

// dense defaults of the strided-I/O fields (after M / the gathered operand are known)
static void dense_io(GemmP& p, int mode) {
    if (mode == CONV_FWD) { p.xbs = (unsigned)p.Cin * p.Hs * p.Ws; p.ybs = (long long)p.Cout * p.OH * p.OW; }
    else { p.xbs = (unsigned)p.Cout * p.OH * p.OW; p.ybs = (long long)p.Cin * p.H * p.W; }
    p.C2 = nullptr; p.ybs2 = 0; p.msplit = 0x7fffffff; p.mask = nullptr; p.mbs = 0;
}

}  // namespace

extern "C" {

int mogan_gemm_set_split_target(int blocks) {
    if (blocks < 64 || blocks > 8192) return MOGAN_ERR_SHAPE;
    g_split_target.store(blocks, std::memory_order_relaxed);
    return 0;
}

int mogan_stream_set_split_target(hipStream_t stream, int blocks) {
    if (blocks != 0 && (blocks < 64 || blocks > 8192)) return MOGAN_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(g_stream_mu);
    if (blocks == 0) g_stream_target.erase(stream); else g_stream_target[stream] = blocks;
    return 0;
}

int mogan_gemm_tune_set(int mode, int M, int N, int K, int nz, int cfg, int split) {
    if (mode < 0 || mode > 3 || cfg < 0 || cfg >= NCFG || split < 1 || M <= 0 || N <= 0 || K <= 0 || nz <= 0) return MOGAN_ERR_SHAPE;
    std::lock_guard<std::mutex> lk(g_tuned_mu);
    g_tuned[TuneKey{mode, M, N, K, nz}] = std::make_pair(cfg, split);
    return 0;
}

int mogan_reserve_streams(int n) {
    static std::vector<hipStream_t> held;
    for (int i = (int)held.size(); i < n; ++i) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return MOGAN_ERR_LAUNCH;
        held.push_back(st);
    }
    return (int)held.size();
}

int mogan_gemm_tune_clear(void) {
    std::lock_guard<std::mutex> lk(g_tuned_mu);
    g_tuned.clear();
    return 0;
}

int mogan_gemm_debug_force(int cfg, int split) { g_force_cfg = cfg; g_force_split = split; return 0; }
int mogan_mfma_form(void) { return MOGAN_X6 ? 6 : 1; }
int mogan_gemm_group_min_tiles(int tiles) { if (tiles < 0) return MOGAN_ERR_SHAPE; g_group_min_tiles = tiles; return 0; }

int mogan_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    g_prof.clear();
    g_prof_on = on != 0;
    return 0;
}

// per-launch CSV (tools/profile_layers.py): consumes the records like mogan_prof_collect
int mogan_prof_dump(const char* path) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    FILE* f = fopen(path, "w");
    if (!f) return MOGAN_ERR_SHAPE;
    fprintf(f, "mode,cfg,M,N,K,nz,nsplit,B,Cin,Cout,H,W,KH,KW,stride,up,gflop,ms\n");
    for (auto& r : g_prof) {
        hipEventSynchronize(r.e1);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = 0.f;
        fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.6f,%.6f\n", r.mode, r.cfg, r.M, r.N, r.K, r.nz, r.nsplit,
                r.Bn, r.Cin, r.Cout, r.H, r.W, r.KH, r.KW, r.s, r.up, r.flops / 1e9, ms);
        hipEventDestroy(r.e0); hipEventDestroy(r.e1);
    }
    g_prof.clear();
    fclose(f);
    return 0;
}

// out: rows of 5 doubles {mode, cfg, launches, algorithmic flops, milliseconds}, one per (mode,cfg) seen
int mogan_prof_collect(double* out, int max_rows) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    constexpr int NMODE = 12;             // 0-6 implicit GEMM / direct / Winograd, 7-9 packed-operand GEMM, 10-11 its grouped launches
    double acc[NMODE][NCFG][3] = {};
    for (auto& r : g_prof) {
        hipEventSynchronize(r.e1);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = 0.f;
        if (r.mode >= 0 && r.mode < NMODE && r.cfg >= 0 && r.cfg < NCFG) {
            acc[r.mode][r.cfg][0] += 1; acc[r.mode][r.cfg][1] += r.flops; acc[r.mode][r.cfg][2] += ms;
        }
        hipEventDestroy(r.e0); hipEventDestroy(r.e1);
    }
    g_prof.clear();
    int n = 0;
    for (int m = 0; m < NMODE; ++m)
        for (int c = 0; c < NCFG; ++c)
            if (acc[m][c][0] > 0 && n < max_rows) {
                double* o = out + 5 * n++;
                o[0] = m; o[1] = c; o[2] = acc[m][c][0]; o[3] = acc[m][c][1]; o[4] = acc[m][c][2];
            }
    return n;
}

int mogan_conv2d_out_dims(int Hs, int Ws, int KH, int KW, int stride, int ph, int pw, int up, int* OH, int* OW) {
    if (stride <= 0) return MOGAN_ERR_SHAPE;
    *OH = ((Hs << up) + 2 * ph - KH) / stride + 1;
    *OW = ((Ws << up) + 2 * pw - KW) / stride + 1;
    return 0;
}

int mogan_conv2d_fwd(const float* x, const float* w, float* y, int B, int Cin, int Hs, int Ws, int Cout, int KH,
                     int KW, int stride, int ph, int pw, int up, void* ws, size_t ws_bytes, hipStream_t stream) {
    return mogan_conv2d_fwd_wp(x, w, nullptr, y, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, ws, ws_bytes, stream);
}

int mogan_conv2d_fwd_wp(const float* x, const float* w, const void* wprep, float* y, int B, int Cin, int Hs, int Ws, int Cout, int KH,
                        int KW, int stride, int ph, int pw, int up, void* ws, size_t ws_bytes, hipStream_t stream) {
    GemmP p{}; int rc = conv_geom(p, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up); if (rc) return rc;
    if (g_force_cfg < 0) {          // <= 4 output channels: HBM streaming work, direct VALU kernel
        rc = mogan_smallc_fwd_try(x, w, y, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    if (g_force_cfg < 0 && up == 0) {   // 3 input channels, 4x4 s2: the discriminators' first layer as a streaming kernel (mogan_stem.hip)
        rc = mogan_stem_fwd_try(x, w, y, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, 1.f, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    if (g_force_cfg == -1) {        // 3x3 s1 p1 at >= 32 channels: fused Winograd F(2x2,3x3), 2.25x fewer multiplies (-2: test hook, off)
        // (recorded flops = the multiplies the kernel executes: 16 per 2x2 outputs instead of 36)
        mogan_prof_begin(4, 1, (4.0 / 9.0) * 2.0 * Cout * (double)B * p.OH * p.OW * Cin * KH * KW, Cout, B * p.OH * p.OW, Cin * KH * KW, stream);
        rc = mogan_wino_try(x, w, wprep, y, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, 0, nullptr, nullptr, 0, ws, ws_bytes, stream);
        mogan_prof_end(rc == 1, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    if (mogan_use_dconv && g_force_cfg < 0) {
        mogan_prof_begin(4, 0, 2.0 * Cout * (double)B * p.OH * p.OW * Cin * KH * KW, Cout, B * p.OH * p.OW, Cin * KH * KW, stream);
        // (the image of a 4x4 s2 convolution is dconv2_fwd_kernel's: mogan_conv_prep_bytes)
        rc = mogan_dconv_fwd_try(x, w, y, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, ws, ws_bytes, stream,
                                 (KH == 4 && KW == 4 && stride == 2) ? wprep : nullptr);
        mogan_prof_end(rc == 1, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    p.A = w; p.B = x; p.C = y; p.M = Cout; p.N = B * p.OH * p.OW; p.K = Cin * KH * KW; p.accumulate = 0;
    p.a_bytes = 4u * Cout * Cin * KH * KW; p.b_bytes = 4u * B * Cin * Hs * Ws;
    p.avec = (p.K % 4 == 0) && (((uintptr_t)w & 15) == 0);
    dense_io(p, CONV_FWD);
    return run_gemm(CONV_FWD, p, 1, (long long)B * Cout * p.OH * p.OW, ws, ws_bytes, stream);
}

// conv + LeakyReLU in the epilogue of the implicit-GEMM kernel (or of its split-K reduction): the first layer of every
// discriminator (nn.Conv2d(3, ndf, 4, 2, 1) -> nn.LeakyReLU(0.2), model.py:597-598, 660-661) has no BatchNorm to carry the
// activation, and its output is the largest map of the network.  Returns 1 when the geometry is not one for this kernel (the
// caller then runs mogan_conv2d_fwd + mogan_act_fwd).
int mogan_conv2d_lrelu_fwd(const float* x, const float* w, float* z, int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW,
                           int stride, int ph, int pw, float slope, void* ws, size_t ws_bytes, hipStream_t stream) {
    GemmP p{}; int rc = conv_geom(p, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, 0); if (rc) return rc;
    if (!(slope > 0.f)) return MOGAN_ERR_SHAPE;
    // only layers the dispatch of mogan_conv2d_fwd would hand to the implicit-GEMM kernel anyway: few input channels
    if (Cin > 16 || Cout <= 4) return 1;
    if (g_force_cfg < 0) {
        rc = mogan_stem_fwd_try(x, w, z, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, slope, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    p.A = w; p.B = x; p.C = z; p.M = Cout; p.N = B * p.OH * p.OW; p.K = Cin * KH * KW; p.accumulate = 0;
    p.a_bytes = 4u * Cout * Cin * KH * KW; p.b_bytes = 4u * B * Cin * Hs * Ws;
    p.avec = (p.K % 4 == 0) && (((uintptr_t)w & 15) == 0);
    p.ep_slope = slope;
    dense_io(p, CONV_FWD);
    return run_gemm(CONV_FWD, p, 1, (long long)B * Cout * p.OH * p.OW, ws, ws_bytes, stream);
}

// conv + per-channel affine (+ ReLU) in the epilogue of the implicit-GEMM kernel (or of its split-K reduction): the
// BasicConv2d of the frozen Inception trunk = conv, eval-mode BN, ReLU (model.py:258-299) in one pass over the output
int mogan_conv2d_affine_fwd(const float* x, const float* w, const float* scale, const float* shift, float* y, int B,
                            int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int relu,
                            void* ws, size_t ws_bytes, hipStream_t stream) {
    return mogan_conv2d_affine_fwd_ex(x, (long long)Cin * Hs * Ws, w, scale, shift, y, -1, nullptr, 0, Cout, B, Cin, Hs, Ws,
                                      Cout, KH, KW, stride, ph, pw, relu, ws, ws_bytes, stream);
}

// The same with channel-slice addressing: x is a slice of a tensor whose batch stride is x_bstride elements (x points
// at the slice's first channel); output channels [0, msplit) go to y (batch stride y_bstride, -1 = dense), channels
// [msplit, Cout) to y2 (batch stride y2_bstride).  Lets a group of same-input 1x1 convolutions run as ONE launch whose
// parts land in different tensors, and lets every branch of an Inception block write straight into the block's
// concatenated output (model.py:258-299 via torchvision's torch.cat).
int mogan_conv2d_affine_fwd_ex(const float* x, long long x_bstride, const float* w, const float* scale, const float* shift,
                               float* y, long long y_bstride, float* y2, long long y2_bstride, int msplit, int B, int Cin,
                               int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int relu, void* ws,
                               size_t ws_bytes, hipStream_t stream) {
    GemmP p{}; int rc = conv_geom(p, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, 0); if (rc) return rc;
    if (!scale || !shift) return MOGAN_ERR_SHAPE;
    const long long xd = (long long)Cin * Hs * Ws, yd = (long long)Cout * p.OH * p.OW;
    if (y_bstride < 0) y_bstride = yd;
    const bool two = y2 != nullptr && msplit > 0 && msplit < Cout;
    const bool plain = x_bstride == xd && y_bstride == yd && !two;
    if (x_bstride < xd || (long long)(B - 1) * x_bstride + xd >= (1ll << 30)) return MOGAN_ERR_SHAPE;
    if (plain && g_force_cfg < 0) {          // the trunk's 3x3 s1 layers on well-filled grids (147x147, 71x71): fused Winograd
        mogan_prof_begin(4, 1, (4.0 / 9.0) * 2.0 * Cout * (double)B * p.OH * p.OW * Cin * KH * KW, Cout, B * p.OH * p.OW, Cin * KH * KW, stream);
        rc = mogan_wino_try(x, w, nullptr, y, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, 0, 0, scale, shift, relu, ws, ws_bytes, stream);
        mogan_prof_end(rc == 1, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    p.A = w; p.B = x; p.C = y; p.M = Cout; p.N = B * p.OH * p.OW; p.K = Cin * KH * KW; p.accumulate = 0;
    p.a_bytes = 4u * Cout * Cin * KH * KW; p.b_bytes = 4u * (unsigned)((long long)(B - 1) * x_bstride + xd);
    p.avec = (p.K % 4 == 0) && (((uintptr_t)w & 15) == 0);
    p.ep_scale = scale; p.ep_shift = shift; p.ep_relu = relu;
    dense_io(p, CONV_FWD);
    p.xbs = (unsigned)x_bstride; p.ybs = y_bstride;
    if (two) { p.C2 = y2; p.ybs2 = y2_bstride; p.msplit = msplit; }
    return run_gemm(CONV_FWD, p, 1, (long long)B * Cout * p.OH * p.OW, ws, ws_bytes, stream);
}

// Plain forward convolution with channel-slice addressing, a ReLU mask on the result and accumulation: the data
// gradient of a STRIDE-1 convolution evaluated as the forward convolution of dY with the spatially flipped, (ci, co)
// transposed filters (pad' = K - 1 - pad).  In the implicit GEMM's forward mode the filter operand is K-contiguous
// (16-byte coalesced loads); its data-gradient mode has to gather W[co][ci][tap] with a stride of KH*KW floats between
// consecutive rows.  For FROZEN weights the flipped copy is made once (attngan/inception.py).
int mogan_conv2d_fwd_ex(const float* x, long long x_bstride, const float* w, float* y, long long y_bstride,
                        const float* relu_of, long long relu_bstride, int accumulate, int B, int Cin, int Hs, int Ws,
                        int Cout, int KH, int KW, int stride, int ph, int pw, void* ws, size_t ws_bytes, hipStream_t stream) {
    GemmP p{}; int rc = conv_geom(p, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, 0); if (rc) return rc;
    const long long xd = (long long)Cin * Hs * Ws, yd = (long long)Cout * p.OH * p.OW;
    if (y_bstride < 0) y_bstride = yd;
    if (x_bstride < xd || y_bstride < yd || (long long)(B - 1) * x_bstride + xd >= (1ll << 30)) return MOGAN_ERR_SHAPE;
    p.A = w; p.B = x; p.C = y; p.M = Cout; p.N = B * p.OH * p.OW; p.K = Cin * KH * KW; p.accumulate = accumulate;
    p.a_bytes = 4u * Cout * Cin * KH * KW; p.b_bytes = 4u * (unsigned)((long long)(B - 1) * x_bstride + xd);
    p.avec = (p.K % 4 == 0) && (((uintptr_t)w & 15) == 0);
    dense_io(p, CONV_FWD);
    p.xbs = (unsigned)x_bstride; p.ybs = y_bstride; p.mask = relu_of; p.mbs = relu_bstride;
    return run_gemm(CONV_FWD, p, 1, (long long)B * Cout * p.OH * p.OW, ws, ws_bytes, stream);
}

}  // extern "C"
namespace {
// GemmP of one strided / fused forward convolution (the arguments of mogan_conv2d_affine_fwd_ex); 0 or an error
static int prep_fwd(GemmP& p, const MoganConvFwdArgs& a) {
    int rc = conv_geom(p, a.B, a.Cin, a.Hs, a.Ws, a.Cout, a.KH, a.KW, a.stride, a.ph, a.pw, 0); if (rc) return rc;
    if (!a.scale || !a.shift) return MOGAN_ERR_SHAPE;
    const long long xd = (long long)a.Cin * a.Hs * a.Ws, yd = (long long)a.Cout * p.OH * p.OW;
    const long long ybs = a.y_bstride < 0 ? yd : a.y_bstride;
    if (a.x_bstride < xd || (long long)(a.B - 1) * a.x_bstride + xd >= (1ll << 30)) return MOGAN_ERR_SHAPE;
    p.A = a.w; p.B = a.x; p.C = a.y; p.M = a.Cout; p.N = a.B * p.OH * p.OW; p.K = a.Cin * a.KH * a.KW; p.accumulate = 0;
    p.a_bytes = 4u * a.Cout * a.Cin * a.KH * a.KW; p.b_bytes = 4u * (unsigned)((long long)(a.B - 1) * a.x_bstride + xd);
    p.avec = (p.K % 4 == 0) && (((uintptr_t)a.w & 15) == 0);
    p.ep_scale = a.scale; p.ep_shift = a.shift; p.ep_relu = a.relu;
    dense_io(p, CONV_FWD);
    p.xbs = (unsigned)a.x_bstride; p.ybs = ybs;
    if (a.y2 != nullptr && a.msplit > 0 && a.msplit < a.Cout) { p.C2 = a.y2; p.ybs2 = a.y2_bstride; p.msplit = a.msplit; }
    return 0;
}
static int prep_dgrad(GemmP& p, const MoganConvDgradArgs& a, int* nz) {
    int rc = conv_geom(p, a.B, a.Cin, a.Hs, a.Ws, a.Cout, a.KH, a.KW, a.stride, a.ph, a.pw, 0); if (rc) return rc;
    const long long yd = (long long)a.Cout * p.OH * p.OW, xd = (long long)a.Cin * a.Hs * a.Ws;
    if (a.dy_bstride < yd || a.dx_bstride < xd || (long long)(a.B - 1) * a.dy_bstride + yd >= (1ll << 30)) return MOGAN_ERR_SHAPE;
    if (a.stride > a.KH || a.stride > a.KW) return MOGAN_ERR_SHAPE;
    p.A = a.w; p.B = a.dy; p.C = a.dx; p.M = a.Cin; p.K = a.Cout * p.nkh * p.nkw; p.accumulate = a.accumulate;
    p.a_bytes = 4u * a.Cout * a.Cin * a.KH * a.KW; p.b_bytes = 4u * (unsigned)((long long)(a.B - 1) * a.dy_bstride + yd);
    const int Hc = (p.H + a.stride - 1) / a.stride, Wc = (p.W + a.stride - 1) / a.stride;
    p.N = a.B * Hc * Wc;
    dense_io(p, CONV_DGRAD);
    p.xbs = (unsigned)a.dy_bstride; p.ybs = a.dx_bstride; p.mask = a.relu_of; p.mbs = a.relu_bstride;
    *nz = a.stride * a.stride;
    return 0;
}
}  // namespace
extern "C" {

// n <= 4 independent forward convolutions (arguments as mogan_conv2d_affine_fwd_ex) in ONE launch, no split-K.  The outputs
// must not overlap.  n == 1 falls back to the single-problem entry point (tile heuristics, split-K).
int mogan_conv2d_affine_fwd_group(int n, const MoganConvFwdArgs* args, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (n <= 0 || n > MAXG || !args) return MOGAN_ERR_SHAPE;
    if (n == 1) {
        const MoganConvFwdArgs& a = args[0];
        return mogan_conv2d_affine_fwd_ex(a.x, a.x_bstride, a.w, a.scale, a.shift, a.y, a.y_bstride, a.y2, a.y2_bstride, a.msplit,
                                          a.B, a.Cin, a.Hs, a.Ws, a.Cout, a.KH, a.KW, a.stride, a.ph, a.pw, a.relu, ws, ws_bytes,
                                          stream);
    }
    GemmP ps[MAXG]; int nz[MAXG];
    for (int i = 0; i < n; ++i) { ps[i] = GemmP{}; int rc = prep_fwd(ps[i], args[i]); if (rc) return rc; nz[i] = 1; }
    int rc = run_group(CONV_FWD, ps, nz, n, stream);
    if (rc != 1) return rc;
    for (int i = 0; i < n; ++i) { rc = mogan_conv2d_affine_fwd_group(1, args + i, ws, ws_bytes, stream); if (rc) return rc; }
    return 0;
}

// n <= 4 independent data gradients (arguments as mogan_conv2d_dgrad_ex) in ONE launch.  Two problems of a group must not
// write (or accumulate into) the same dx elements.
int mogan_conv2d_dgrad_group(int n, const MoganConvDgradArgs* args, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (n <= 0 || n > MAXG || !args) return MOGAN_ERR_SHAPE;
    if (n == 1) {
        const MoganConvDgradArgs& a = args[0];
        return mogan_conv2d_dgrad_ex(a.dy, a.dy_bstride, a.w, a.dx, a.dx_bstride, a.relu_of, a.relu_bstride, a.accumulate, a.B,
                                     a.Cin, a.Hs, a.Ws, a.Cout, a.KH, a.KW, a.stride, a.ph, a.pw, ws, ws_bytes, stream);
    }
    GemmP ps[MAXG]; int nz[MAXG];
    for (int i = 0; i < n; ++i) { ps[i] = GemmP{}; int rc = prep_dgrad(ps[i], args[i], &nz[i]); if (rc) return rc; }
    int rc = run_group(CONV_DGRAD, ps, nz, n, stream);
    if (rc != 1) return rc;
    for (int i = 0; i < n; ++i) { rc = mogan_conv2d_dgrad_group(1, args + i, ws, ws_bytes, stream); if (rc) return rc; }
    return 0;
}

int mogan_conv2d_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Hs, int Ws, int Cout, int KH,
                       int KW, int stride, int ph, int pw, int up, void* ws, size_t ws_bytes, hipStream_t stream) {
    return mogan_conv2d_dgrad_wp(dy, w, nullptr, dx, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, ws, ws_bytes, stream);
}

int mogan_conv2d_dgrad_wp(const float* dy, const float* w, const void* wprep, float* dx, int B, int Cin, int Hs, int Ws, int Cout,
                          int KH, int KW, int stride, int ph, int pw, int up, void* ws, size_t ws_bytes, hipStream_t stream) {
    GemmP p{}; int rc = conv_geom(p, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up); if (rc) return rc;
    if (g_force_cfg < 0) {          // <= 4 channels on one side (image heads, first D convolution)
        rc = mogan_smallc_dgrad_try(dy, w, dx, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    if (g_force_cfg == -1) {
        mogan_prof_begin(5, 1, (4.0 / 9.0) * 2.0 * Cout * (double)B * p.OH * p.OW * Cin * KH * KW, Cin, B * p.H * p.W, Cout * KH * KW, stream);
        rc = mogan_wino_try(dy, w, wprep, dx, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, 1, nullptr, nullptr, 0, ws, ws_bytes, stream);
        mogan_prof_end(rc == 1, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    if (mogan_use_dconv && g_force_cfg < 0) {
        mogan_prof_begin(5, 0, 2.0 * Cout * (double)B * p.OH * p.OW * Cin * KH * KW, Cin, B * p.H * p.W, Cout * KH * KW, stream);
        rc = mogan_dconv_dgrad_try(dy, w, dx, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, ws, ws_bytes, stream,
                                   (KH == 4 && KW == 4 && stride == 2) ? wprep : nullptr);
        mogan_prof_end(rc == 1, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    // dx is the gradient w.r.t. the conv input in the (upsampled) H x W domain: (B,Cin,H,W)
    p.A = w; p.B = dy; p.C = dx; p.M = Cin; p.K = Cout * p.nkh * p.nkw; p.accumulate = 0;
    p.a_bytes = 4u * Cout * Cin * KH * KW; p.b_bytes = 4u * B * Cout * p.OH * p.OW;
    const int Hc = (p.H + stride - 1) / stride, Wc = (p.W + stride - 1) / stride;
    p.N = B * Hc * Wc;   // class (0,0) has the most columns
    dense_io(p, CONV_DGRAD);
    return run_gemm(CONV_DGRAD, p, stride * stride, (long long)B * Cin * p.H * p.W, ws, ws_bytes, stream);
}

// ---- prepared filter images, any kind (include/mogan_hip.h "Prepared filter images") ------------------------------------------
// Which image mogan_conv2d_fwd_wp / _dgrad_wp take for a geometry follows the dispatch above: 3x3 s1 p1 -> the Winograd kernels'
// (mogan_wino_prep_bytes), 4x4 s2 p1 -> dconv2_fwd_kernel's, found by a dry run of that branch of the dispatch.
size_t mogan_conv_prep_bytes(int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int up, int dgrad) {
    if (KH == 3 && KW == 3) return mogan_wino_prep_bytes(B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, dgrad);
    if (!(KH == 4 && KW == 4 && stride == 2 && ph == 1 && pw == 1 && up == 0) || B <= 0 || Cin <= 0 || Cout <= 0 || Hs <= 0 || Ws <= 0)
        return 0;
    if (!mogan_use_dconv || g_force_cfg >= 0) return 0;
    // (the stages ahead of the direct kernels -- <= 4 channels on one side, the 3-channel first layer -- never meet a geometry the
    // direct kernels accept: those need >= 8 / >= 64 channels on the two sides)
    float* const fake = (float*)(uintptr_t)256;           // never dereferenced: the dry run launches nothing
    size_t bytes = 0;
    const int rc = dgrad ? mogan_dconv_dgrad_try(fake, fake, fake, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, fake, (size_t)1 << 40, nullptr,
                                                 nullptr, &bytes)
                         : mogan_dconv_fwd_try(fake, fake, fake, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, fake, (size_t)1 << 40, nullptr,
                                               nullptr, &bytes);
    return rc == 1 ? bytes : 0;
}

int mogan_conv_prep_group(int n, const float* const* w, void* const* prep, const int* Cout, const int* Cin, const int* KH,
                          const int* dgrad, hipStream_t st) {
    if (n < 0 || (n > 0 && (!w || !prep || !Cout || !Cin || !KH || !dgrad))) return MOGAN_ERR_SHAPE;
    // the members of each kind keep their order; one grouped launch per kind (and per 32 members)
    std::vector<const float*> ws_[2]; std::vector<void*> ps_[2]; std::vector<int> co_[2], ci_[2], dg_[2];
    for (int i = 0; i < n; ++i) {
        if (KH[i] != 3 && KH[i] != 4) return MOGAN_ERR_SHAPE;
        const int k = KH[i] == 4;
        ws_[k].push_back(w[i]); ps_[k].push_back(prep[i]); co_[k].push_back(Cout[i]); ci_[k].push_back(Cin[i]); dg_[k].push_back(dgrad[i]);
    }
    int rc = 0;
    if (!ws_[0].empty()) rc = mogan_wino_prep_group((int)ws_[0].size(), ws_[0].data(), ps_[0].data(), co_[0].data(), ci_[0].data(), dg_[0].data(), st);
    if (!rc && !ws_[1].empty())
        rc = mogan_dconv2_prep_group((int)ws_[1].size(), ws_[1].data(), ps_[1].data(), co_[1].data(), ci_[1].data(), dg_[1].data(), st);
    return rc;
}

// Data gradient with channel-slice addressing and a fused ReLU backward: dy is a slice of a tensor with batch stride
// dy_bstride; the result is written (accumulate = 0) or added (1) to dx, a slice of a tensor with batch stride
// dx_bstride; where relu_of[...] <= 0 (same slice geometry as dx, batch stride relu_bstride; nullable) the NEW
// contribution is zeroed first -- the ReLU backward of the layer whose output is this convolution's input, so the
// branches that consume one activation accumulate their already-masked gradients in place.
int mogan_conv2d_dgrad_ex(const float* dy, long long dy_bstride, const float* w, float* dx, long long dx_bstride,
                          const float* relu_of, long long relu_bstride, int accumulate, int B, int Cin, int Hs, int Ws,
                          int Cout, int KH, int KW, int stride, int ph, int pw, void* ws, size_t ws_bytes,
                          hipStream_t stream) {
    GemmP p{}; int rc = conv_geom(p, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, 0); if (rc) return rc;
    const long long yd = (long long)Cout * p.OH * p.OW, xd = (long long)Cin * Hs * Ws;
    if (dy_bstride < yd || dx_bstride < xd || (long long)(B - 1) * dy_bstride + yd >= (1ll << 30)) return MOGAN_ERR_SHAPE;
    if (dy_bstride == yd && dx_bstride == xd && !relu_of && !accumulate)
        return mogan_conv2d_dgrad(dy, w, dx, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, 0, ws, ws_bytes, stream);
    p.A = w; p.B = dy; p.C = dx; p.M = Cin; p.K = Cout * p.nkh * p.nkw; p.accumulate = accumulate;
    p.a_bytes = 4u * Cout * Cin * KH * KW; p.b_bytes = 4u * (unsigned)((long long)(B - 1) * dy_bstride + yd);
    const int Hc = (p.H + stride - 1) / stride, Wc = (p.W + stride - 1) / stride;
    p.N = B * Hc * Wc;
    dense_io(p, CONV_DGRAD);
    p.xbs = (unsigned)dy_bstride; p.ybs = dx_bstride; p.mask = relu_of; p.mbs = relu_bstride;
    // a strided parity class that has no tap (stride > kernel) would leave its pixels unwritten: not used by the trunk
    if (stride > KH || stride > KW) return MOGAN_ERR_SHAPE;
    return run_gemm(CONV_DGRAD, p, stride * stride, (long long)B * Cin * p.H * p.W, ws, ws_bytes, stream);
}

int mogan_conv2d_wgrad(const float* dy, const float* x, float* dw, int B, int Cin, int Hs, int Ws, int Cout, int KH,
                       int KW, int stride, int ph, int pw, int up, int accumulate, void* ws, size_t ws_bytes,
                       hipStream_t stream) {
    GemmP p{}; int rc = conv_geom(p, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up); if (rc) return rc;
    if (g_force_cfg < 0) {
        rc = mogan_smallc_wgrad_try(dy, x, dw, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, accumulate, ws, ws_bytes, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    if (g_force_cfg == -1) {
        mogan_prof_begin(6, 1, (4.0 / 9.0) * 2.0 * Cout * (double)B * p.OH * p.OW * Cin * KH * KW, Cout, Cin * KH * KW, B * p.OH * p.OW, stream);
        rc = mogan_wino_wgrad_try(dy, x, dw, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, accumulate, ws, ws_bytes, stream);
        mogan_prof_end(rc == 1, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    if (mogan_use_dconv && g_force_cfg < 0) {
        mogan_prof_begin(6, 0, 2.0 * Cout * (double)B * p.OH * p.OW * Cin * KH * KW, Cout, Cin * KH * KW, B * p.OH * p.OW, stream);
        rc = mogan_dconv_wgrad_try(dy, x, dw, B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, accumulate, ws, ws_bytes, stream);
        mogan_prof_end(rc == 1, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }
    p.A = dy; p.B = x; p.C = dw; p.M = Cout; p.N = Cin * KH * KW; p.K = B * p.OH * p.OW; p.accumulate = accumulate;
    p.a_bytes = 4u * B * Cout * p.OH * p.OW; p.b_bytes = 4u * B * Cin * Hs * Ws;
    p.avec = ((p.OH * p.OW) % 4 == 0) && (((uintptr_t)dy & 15) == 0);
    return run_gemm(CONV_WGRAD, p, 1, (long long)Cout * Cin * KH * KW, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// nearest-x2 upsample + conv3x3(p1)  ==  the TRANSPOSED 4x4 stride-2 pad-1 convolution with kernel K = T w T^t
// (T = [[0,0,1],[0,1,1],[1,1,0],[1,0,0]] applied to the rows and to the columns of every 3x3 filter): each output
// phase (py,px) of the upsampled grid only ever sees 2x2 distinct source pixels, so 4 multiply-adds per output
// pixel replace 9 (2.25x fewer FLOPs, same result up to the fp32 rounding of the pre-summed weights).
//   forward  Y = C4^T(X)      -> mogan_conv2d_dgrad of the virtual conv C4 (in = Cout, out = Cin, 4x4 s2 p1)
//   dgrad    dX = C4(dY)       -> mogan_conv2d_fwd  of C4: the gradient comes out at the SOURCE resolution
//   wgrad    dK = wgrad_C4(dy4 = X, x4 = dY);  dW = T^t dK T (accumulated into dw)
// K[ci][co][kh][kw] lives at the head of the caller's workspace.
namespace {
__global__ __launch_bounds__(256) void upconv_k4_kernel(const float* __restrict__ w, float* __restrict__ K4, int Cout,
                                                        int Cin) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;        // one thread per (ci, co)
    if (i >= (long long)Cout * Cin) return;
    const int ci = (int)(i / Cout), co = (int)(i - (long long)ci * Cout);
    const float* s = w + ((size_t)co * Cin + ci) * 9;
    float f[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) f[a][b] = s[a * 3 + b];
    float r[4][3];                                                        // rows: [w2, w1+w2, w0+w1, w0]
#pragma unroll
    for (int b = 0; b < 3; ++b) { r[0][b] = f[2][b]; r[1][b] = f[1][b] + f[2][b]; r[2][b] = f[0][b] + f[1][b]; r[3][b] = f[0][b]; }
    float* d = K4 + i * 16;
#pragma unroll
    for (int a = 0; a < 4; ++a) { d[a * 4 + 0] = r[a][2]; d[a * 4 + 1] = r[a][1] + r[a][2]; d[a * 4 + 2] = r[a][0] + r[a][1]; d[a * 4 + 3] = r[a][0]; }
}
// dW[co][ci] (+)= T^t dK[ci][co] T
__global__ __launch_bounds__(256) void upconv_k4_grad_kernel(const float* __restrict__ dK, float* __restrict__ dw, int Cout,
                                                             int Cin, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;        // one thread per (co, ci): coalesced dW
    if (i >= (long long)Cout * Cin) return;
    const int co = (int)(i / Cin), ci = (int)(i - (long long)co * Cin);
    const float* s = dK + ((size_t)ci * Cout + co) * 16;
    float g[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) g[a][b] = s[a * 4 + b];
    float r[3][4];                                                        // T^t rows: w0 <- k2+k3, w1 <- k1+k2, w2 <- k0+k1
#pragma unroll
    for (int b = 0; b < 4; ++b) { r[0][b] = g[2][b] + g[3][b]; r[1][b] = g[1][b] + g[2][b]; r[2][b] = g[0][b] + g[1][b]; }
    float* d = dw + i * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v0 = r[a][2] + r[a][3], v1 = r[a][1] + r[a][2], v2 = r[a][0] + r[a][1];
        if (accumulate) { d[a * 3 + 0] += v0; d[a * 3 + 1] += v1; d[a * 3 + 2] += v2; }
        else { d[a * 3 + 0] = v0; d[a * 3 + 1] = v1; d[a * 3 + 2] = v2; }
    }
}

// K of up to 32 weights in one launch (mogan_upconv3x3_k4_group)
constexpr int UKG_MAX = 32;
struct UpK4Group { const float* w[UKG_MAX]; float* k4[UKG_MAX]; int Cout[UKG_MAX], Cin[UKG_MAX]; unsigned end[UKG_MAX]; int n; };
__global__ __launch_bounds__(256) void upconv_k4_group_kernel(const UpK4Group g) {
    int m = 0;
#pragma unroll 1
    for (int i = 0; i < g.n - 1; ++i) if (blockIdx.x >= g.end[i]) m = i + 1;
    const unsigned start = m ? g.end[m - 1] : 0u;
    const int Cout = g.Cout[m], Cin = g.Cin[m];
    const long long i = (long long)(blockIdx.x - start) * 256 + threadIdx.x;        // one thread per (ci, co)
    if (i >= (long long)Cout * Cin) return;
    const int ci = (int)(i / Cout), co = (int)(i - (long long)ci * Cout);
    const float* s = g.w[m] + ((size_t)co * Cin + ci) * 9;
    float f[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) f[a][b] = s[a * 3 + b];
    float r[4][3];                                                        // (the arithmetic of upconv_k4_kernel, term by term)
#pragma unroll
    for (int b = 0; b < 3; ++b) { r[0][b] = f[2][b]; r[1][b] = f[1][b] + f[2][b]; r[2][b] = f[0][b] + f[1][b]; r[3][b] = f[0][b]; }
    float* d = g.k4[m] + i * 16;
#pragma unroll
    for (int a = 0; a < 4; ++a) { d[a * 4 + 0] = r[a][2]; d[a * 4 + 1] = r[a][1] + r[a][2]; d[a * 4 + 2] = r[a][0] + r[a][1]; d[a * 4 + 3] = r[a][0]; }
}

inline size_t upconv_k_bytes(int Cout, int Cin) { return (((size_t)Cout * Cin * 16 * sizeof(float)) + 255) & ~(size_t)255; }

int upconv_make_k(const float* w, void* ws, size_t ws_bytes, int Cout, int Cin, hipStream_t st) {
    if (!ws || ws_bytes < upconv_k_bytes(Cout, Cin) || Cout <= 0 || Cin <= 0) return MOGAN_ERR_WS;
    const long long n = (long long)Cout * Cin;
    hipLaunchKernelGGL(upconv_k4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, (float*)ws, Cout, Cin);
    return 0;
}
}  // namespace

size_t mogan_upconv3x3_ws_bytes(int Cout, int Cin) { return upconv_k_bytes(Cout, Cin); }

// K = T w T^t alone, into the caller's buffer (Cin, Cout, 4, 4): an owner of w keeps K (and the filter image of the kernel that runs
// the virtual convolution, mogan_conv_prep_*) per weight version and calls mogan_conv2d_dgrad_wp / mogan_conv2d_fwd_wp on K itself
// -- what mogan_upconv3x3_fwd / _dgrad do after building K at the head of the workspace, per call
int mogan_upconv3x3_k4(const float* w, float* k4, int Cout, int Cin, hipStream_t stream) {
    if (!w || !k4 || Cout <= 0 || Cin <= 0) return MOGAN_ERR_SHAPE;
    const long long n = (long long)Cout * Cin;
    hipLaunchKernelGGL(upconv_k4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, k4, Cout, Cin);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

int mogan_upconv3x3_k4_group(int n, const float* const* w, float* const* k4, const int* Cout, const int* Cin, hipStream_t stream) {
    if (n < 0 || (n > 0 && (!w || !k4 || !Cout || !Cin))) return MOGAN_ERR_SHAPE;
    for (int i0 = 0; i0 < n; i0 += UKG_MAX) {
        UpK4Group g{};
        g.n = std::min(UKG_MAX, n - i0);
        unsigned end = 0;
        for (int j = 0; j < g.n; ++j) {
            const int i = i0 + j;
            if (!w[i] || !k4[i] || Cout[i] <= 0 || Cin[i] <= 0) return MOGAN_ERR_SHAPE;
            g.w[j] = w[i]; g.k4[j] = k4[i]; g.Cout[j] = Cout[i]; g.Cin[j] = Cin[i];
            end += (unsigned)(((long long)Cout[i] * Cin[i] + 255) / 256); g.end[j] = end;
        }
        hipLaunchKernelGGL(upconv_k4_group_kernel, dim3(end), dim3(256), 0, stream, g);
    }
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

int mogan_upconv3x3_fwd(const float* x, const float* w, float* y, int B, int Cin, int Hs, int Ws, int Cout, void* ws,
                        size_t ws_bytes, hipStream_t stream) {
    int rc = upconv_make_k(w, ws, ws_bytes, Cout, Cin, stream); if (rc) return rc;
    const size_t kb = upconv_k_bytes(Cout, Cin);
    // C4: input (B,Cout,2Hs,2Ws) -> output (B,Cin,Hs,Ws); its data gradient maps X to Y
    return mogan_conv2d_dgrad(x, (const float*)ws, y, B, Cout, 2 * Hs, 2 * Ws, Cin, 4, 4, 2, 1, 1, 0, (char*)ws + kb,
                              ws_bytes - kb, stream);
}

int mogan_upconv3x3_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Hs, int Ws, int Cout, void* ws,
                          size_t ws_bytes, hipStream_t stream) {
    int rc = upconv_make_k(w, ws, ws_bytes, Cout, Cin, stream); if (rc) return rc;
    const size_t kb = upconv_k_bytes(Cout, Cin);
    return mogan_conv2d_fwd(dy, (const float*)ws, dx, B, Cout, 2 * Hs, 2 * Ws, Cin, 4, 4, 2, 1, 1, 0, (char*)ws + kb,
                            ws_bytes - kb, stream);
}

int mogan_upconv3x3_wgrad(const float* dy, const float* x, float* dw, int B, int Cin, int Hs, int Ws, int Cout,
                          int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    const size_t kb = upconv_k_bytes(Cout, Cin);
    if (!ws || ws_bytes < kb) return MOGAN_ERR_WS;
    // dK (Cin,Cout,4,4) = weight gradient of C4 with output-gradient X and input dY
    int rc = mogan_conv2d_wgrad(x, dy, (float*)ws, B, Cout, 2 * Hs, 2 * Ws, Cin, 4, 4, 2, 1, 1, 0, 0, (char*)ws + kb,
                                ws_bytes - kb, stream);
    if (rc) return rc;
    const long long n = (long long)Cout * Cin;
    hipLaunchKernelGGL(upconv_k4_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)ws,
                       dw, Cout, Cin, accumulate);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

int mogan_bmm(const float* a, const float* b, float* c, int batch, int M, int N, int K, long long sAb, long long sAm,
              long long sAk, long long sBb, long long sBk, long long sBn, long long sCb, long long sCm, long long sCn,
              int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (batch <= 0 || M < 0 || N < 0 || K < 0) return MOGAN_ERR_SHAPE;
    GemmP p{};
    p.A = a; p.B = b; p.C = c; p.M = M; p.N = N; p.K = K; p.accumulate = accumulate;
    if (sAb < 0 || sAm < 0 || sAk < 0 || sBb < 0 || sBk < 0 || sBn < 0) return MOGAN_ERR_SHAPE;
    const long long ea = 1 + (long long)(batch - 1) * sAb + (long long)(M > 0 ? M - 1 : 0) * sAm + (long long)(K > 0 ? K - 1 : 0) * sAk;
    const long long eb = 1 + (long long)(batch - 1) * sBb + (long long)(K > 0 ? K - 1 : 0) * sBk + (long long)(N > 0 ? N - 1 : 0) * sBn;
    if (ea >= (1ll << 30) || eb >= (1ll << 30)) return MOGAN_ERR_SHAPE;
    p.a_bytes = (unsigned)(4 * ea); p.b_bytes = (unsigned)(4 * eb);
    p.avec = (sAk == 1 && sBk == 1 && K % 4 == 0 && sAm % 4 == 0 && sBn % 4 == 0 && sAb % 4 == 0 && sBb % 4 == 0 &&
              (((uintptr_t)a | (uintptr_t)b) & 15) == 0);
    p.sAb = sAb; p.sAm = sAm; p.sAk = sAk; p.sBb = sBb; p.sBk = sBk; p.sBn = sBn; p.sCb = sCb; p.sCm = sCm; p.sCn = sCn;
    p.a_lane_k = (sAk <= sAm); p.b_lane_n = (sBn <= sBk);
    p.OH = p.OW = p.Hs = p.Ws = p.KH = p.KW = p.s = 1;
    // split-K slabs are addressed like C: only allowed when C is dense (batch,M,N) row-major
    const bool dense = (sCn == 1 && sCm == N && (batch == 1 || sCb == (long long)M * N));
    return run_gemm(BMM, p, batch, (long long)batch * M * N, dense ? ws : nullptr, dense ? ws_bytes : 0, stream);
}

}  // extern "C"
