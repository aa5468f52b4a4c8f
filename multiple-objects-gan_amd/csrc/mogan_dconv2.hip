// mogan_dconv2.hip -- direct convolution, second form (round 5): BOTH operands of the MFMA arrive pre-split.
//
// mogan_dconv.hip's forward kernel was designed around the native fp32 MFMA (512 matrix-pipe cycles per 16 k of a 32 x 32 tile):
// it keeps the input halo as fp32 planes in LDS, reads a lane's eight B values with eight ds_read_b32 and splits them into
// their bf16 pieces in registers -- per tap, i.e. nine times per staged value for a 3 x 3 filter.  On the bf16 pipe the same 16 k
// are 192 cycles and everything around the MFMAs shows: lab builds of that kernel (tools/lab, DESIGN_LOG.md E) run the bare K loop
// at 200 TFLOP/s fp32-equivalent where a loop of 9 + 3 ds_read_b128 per 18 MFMAs sustains 370-385 (tools/lab/mfma_peak.hip), and
// lose another 25 % to a K loop padded from 36 to 40 steps, two barriers per 8 channels, and the split arithmetic.
//
// This kernel (split-bf16 build only; 3 x 3 and 2 x 2 stride-1 filters, i.e. the 3 x 3 convolutions and the four parity
// sub-convolutions of a 4 x 4 stride-2 data gradient):
//   * a block = 8 waves = BM output channels x (8 rows x 32 columns) of one image; wave w owns row w, all BM channels
//     (TM x 1 accumulator tiles): one block per CU, two waves per SIMD, the filters of a stage staged ONCE for 256 pixels;
//   * K runs in stages of NS x 16 input channels; one MFMA group (16 k) = ONE tap x 16 channels (lane half h = channels 8h..8h+7):
//     KH*KW groups per 16 channels, no padded steps;
//   * the halo is split ONCE, when it is staged: LDS holds Xh[sub][piece][half][pixel][8 channels x bf16] -- a lane's B operand of a
//     group is one 16-byte read per piece at (lane base + compile-time tap offset), consecutive lanes = consecutive pixels
//     (conflict-free), no VALU between the reads and the MFMAs;
//   * the filters arrive pre-split from dconv2_wprep_kernel in exactly the order the stage's LDS image has (16-byte copies):
//     Wl[row][group][piece][half][8 x bf16], row stride an odd multiple of 16 bytes.
// Replaces dconv_fwd_kernel for: nn.Conv2d 3x3 s1 p1 forward / data gradient (code/coco/attngan/model.py:35-55, 67-81) and the
// data gradient of the 4x4 s2 p1 convolutions (model.py:575-613, 646-760) on maps whose grid is a multiple of 8 x 32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include "../../include/mogan_hip.h"
#include "mogan_internal.h"
#include "mogan_mma.h"

#if MOGAN_X6
#if defined(DCONV2_LAB) && DCONV2_LAB == 9       // lab: shader-clock cycles of wave 0 per kernel segment, summed over the blocks
__device__ unsigned long long g_d2_seg[8];
#define D2_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define D2_ADD(i, a, b) do { if (tid == 0) atomicAdd(&g_d2_seg[i], (b) - (a)); } while (0)
extern "C" int mogan_lab_d2_segments(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_d2_seg), sizeof(g_d2_seg)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_d2_seg), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define D2_T(var) do {} while (0)
#define D2_ADD(i, a, b) do {} while (0)
#endif
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ldg1(__amdgpu_buffer_rsrc_t r, unsigned idx) {   // idx = 0x3FFFFFFF -> 0.f
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, idx * 4u, 0, 0));
}
__device__ __forceinline__ f32x4 ldg4w(__amdgpu_buffer_rsrc_t r, unsigned idx) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, idx * 4u, 0, 0));
}
#define D2_OOB 0x3FFFFFFFu

struct D2P {
    const float* X; const void* Wp; float* Y; float* ws;
    int B, Cin, Cout, H, W;                  // input (B, Cin, H, W); space-to-depth form: Cin = 4 x the tensor's channels CinX
    int CinX;                                // channels of the tensor X
    int OH, OW, pt, pl;                      // output grid of the (sub-)convolution, top / left padding
    int yH, yW, ys;                          // output plane dims and pixel stride (2: the parity classes of a stride-2 data gradient)
    int tiles_x, tiles_y, nsplit, sps, npar; // spatial tiles, K split over stages (stages per split), parity classes
    int tiles_cw;                            // columns of a spatial tile (32 or 16; rows = 256 / columns)
    int ntiles, tpb;                         // spatial tiles in all (B * tiles_x * tiles_y), tiles per persistent block
    long long slab; int accumulate;
    unsigned x_bytes, wp_bytes;
};

// Filters -> Wp[stage][row][unit u = (sub * KHW + tap) * 6 + piece * 2 + half] of 16 bytes = the 8 channels 16 sub + 8 half + i of the
// stage at that tap, for the convolution the kernel runs: rows = npar * Cout', channels Cin'.  wmode selects how that convolution's
// filters Wt[row][c][tap] come out of the tensor w (one thread per (stage, row, sub, tap, half)):
//   0  Wt = w as it is, [rows][Cin'][KHW]: a forward convolution (or filters some other kernel has already transformed);
//   1  data gradient of a 3x3 s1 convolution w (Co, Ci, 3, 3): Cout' = Ci, Cin' = Co, Wt[ci][co][tap] = w[co][ci][8 - tap]
//      (what wflip_kernel of mogan_dconv.hip materialises);
//   2  data gradient of a 4x4 s2 p1 convolution w (Co, Ci, 4, 4) = four 2x2 s1 convolutions, one per output parity (py, px):
//      Wt[par * Ci + ci][co][a * 2 + b] = w[co][ci][((py + 1) & 1) + 2 (1 - a)][((px + 1) & 1) + 2 (1 - b)] (wparity_kernel);
//   3  FORWARD of a 4x4 s2 p1 convolution w (Co, Ci, 4, 4) as a 2x2 s1 convolution over the space-to-depth image of the padded
//      input (NS = 2; a stage = 8 channels x the four phases (dy, dx) = (sub, half)): unit (stage, row, sub, tap = (a, b), half)
//      = w[row][8 stage + i][2 a + sub][2 b + half], i = 0..7.  Cout1 = Ci.
__global__ __launch_bounds__(256) void dconv2_wprep_kernel(const float* __restrict__ w, uint4* __restrict__ wp, int rows, int Cin,
                                                           int KHW, int NS, int wmode, int Cout1, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int half = (int)(t & 1); long long r = t >> 1;
    const int tap = (int)(r % KHW); r /= KHW;
    const int sub = (int)(r % NS); r /= NS;
    const int row = (int)(r % rows); const int stage = (int)(r / rows);
    const int c0 = (stage * NS + sub) * 16 + half * 8;
    float v[8];
    if (wmode == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = w[((size_t)row * Cin + c0 + i) * KHW + tap];
    } else if (wmode == 1) {             // rows = Ci of w, Cin = Co of w
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = w[((size_t)(c0 + i) * rows + row) * 9 + (8 - tap)];
    } else if (wmode == 3) {             // Cout1 = Ci of w
        const int kh = 2 * (tap >> 1) + sub, kw = 2 * (tap & 1) + half;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = w[((size_t)row * Cout1 + stage * 8 + i) * 16 + kh * 4 + kw];
    } else {                             // Cout1 = Ci of w (rows = 4 * Ci), Cin = Co of w
        const int par = row / Cout1, ci = row - par * Cout1, py = par >> 1, px = par & 1;
        const int kh = ((py + 1) & 1) + 2 * (1 - (tap >> 1)), kw = ((px + 1) & 1) + 2 * (1 - (tap & 1));
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = w[((size_t)(c0 + i) * Cout1 + ci) * 16 + kh * 4 + kw];
    }
    const X6Frag f = x6_split8(v);
    const size_t base = ((size_t)stage * rows + row) * (size_t)(NS * KHW * 6) + (size_t)(sub * KHW + tap) * 6 + half;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) wp[base + pl * 2] = __builtin_bit_cast(uint4, f.p[pl]);
}

// The same for up to D2G_MAX weights in one launch (mogan_dconv2_prep_group: the owner of the weights rebuilds the images of a
// bucket behind its optimizer step instead of one prep launch per convolution call)
constexpr int D2G_MAX = 32;
struct D2PrepGroup { const float* w[D2G_MAX]; uint4* wp[D2G_MAX]; int rows[D2G_MAX], Cin[D2G_MAX], NS[D2G_MAX], wmode[D2G_MAX], Cout1[D2G_MAX];
                     unsigned end[D2G_MAX]; long long total[D2G_MAX]; int n; };

__device__ __forceinline__ void dconv2_wprep_item(const float* __restrict__ w, uint4* __restrict__ wp, int rows, int Cin, int KHW, int NS,
                                                  int wmode, int Cout1, long long t) {
    const int half = (int)(t & 1); long long r = t >> 1;
    const int tap = (int)(r % KHW); r /= KHW;
    const int sub = (int)(r % NS); r /= NS;
    const int row = (int)(r % rows); const int stage = (int)(r / rows);
    const int c0 = (stage * NS + sub) * 16 + half * 8;
    float v[8];
    if (wmode == 3) {             // Cout1 = Ci of w
        const int kh = 2 * (tap >> 1) + sub, kw = 2 * (tap & 1) + half;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = w[((size_t)row * Cout1 + stage * 8 + i) * 16 + kh * 4 + kw];
    } else {                      // wmode 2: Cout1 = Ci of w (rows = 4 * Ci), Cin = Co of w
        const int par = row / Cout1, ci = row - par * Cout1, py = par >> 1, px = par & 1;
        const int kh = ((py + 1) & 1) + 2 * (1 - (tap >> 1)), kw = ((px + 1) & 1) + 2 * (1 - (tap & 1));
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = w[((size_t)(c0 + i) * Cout1 + ci) * 16 + kh * 4 + kw];
    }
    const X6Frag f = x6_split8(v);
    const size_t base = ((size_t)stage * rows + row) * (size_t)(NS * KHW * 6) + (size_t)(sub * KHW + tap) * 6 + half;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) wp[base + pl * 2] = __builtin_bit_cast(uint4, f.p[pl]);
}

__global__ __launch_bounds__(256) void dconv2_wprep_group_kernel(const D2PrepGroup g) {
    int m = 0;
#pragma unroll 1
    for (int i = 0; i < g.n - 1; ++i) if (blockIdx.x >= g.end[i]) m = i + 1;
    const unsigned start = m ? g.end[m - 1] : 0u;
    const long long t = (long long)(blockIdx.x - start) * 256 + threadIdx.x;
    if (t >= g.total[m]) return;
    dconv2_wprep_item(g.w[m], g.wp[m], g.rows[m], g.Cin[m], 4, g.NS[m], g.wmode[m], g.Cout1[m], t);
}

// CW = columns of the spatial tile (32: 8 rows x 32 columns, 16: 16 x 16 -- the maps with 16-pixel rows); S2D = the 2 x 2 filter
// runs over the space-to-depth image of a stride-2 convolution's padded input: "pixel" (y, x) of sub-chunk dy, lane half dx is
// X[c][2 y + dy - 1][2 x + dx - 1], a stage = 8 channels of X (NS = 2, KH = KW = 2)
template <int KH, int KW, int TM, int NS, int CW, bool S2D>
__global__ __launch_bounds__(512) void dconv2_fwd_kernel(const D2P p) {
    static_assert(!S2D || (KH == 2 && KW == 2 && NS == 2), "space-to-depth form: 2 x 2 filter, two sub-chunks");
    constexpr int KHW = KH * KW, BM = TM * 32, R = 256 / CW;
    constexpr int HH = R + KH - 1, WW = CW + KW - 1, NPIX = HH * WW;
    constexpr int NG = NS * KHW, UPR = NG * 6, WROW = (UPR + 1) * 16;       // groups per stage; 16-byte units / bytes per filter row
    constexpr int XSZ = NS * 6 * NPIX * 16;
    constexpr int NWU = (BM * UPR + 511) / 512;                             // filter units per thread and stage
    constexpr int NIT = (NS * 2 * NPIX + 511) / 512;                        // halo items (pixel, 8 channels) per thread and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Xh = smem;
    unsigned char* const Wl = smem + XSZ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int prow = (wave * 32 + (lane & 31)) / CW, pcol = (wave * 32 + (lane & 31)) % CW;     // this lane's pixel of the tile
    D2_T(t_k0);
    const int m0 = blockIdx.y * BM;
    const int sp = blockIdx.z % p.nsplit, par = blockIdx.z / p.nsplit;
    int pt = p.pt, pl_ = p.pl, y0 = 0, x0 = 0;
    if (p.npar == 4) {                       // stride-2 data gradient: parity class (py, px) -> its padding and output phase
        const int py = par >> 1, px = par & 1;
        pt = 1 - (py + 1 - ((py + 1) & 1)) / 2; pl_ = 1 - (px + 1 - ((px + 1) & 1)) / 2;
        y0 = py; x0 = px;
    }
    const int rows_all = p.npar * p.Cout;
    const int wrow0 = par * p.Cout + m0;
    const int nst_all = p.Cin / (16 * NS);
    const int s_beg = sp * p.sps, s_end = min(nst_all, s_beg + p.sps);
    // persistent block: the spatial tiles [t_beg, t_end) of this (channel block, K split, parity class), one after the other --
    // the first stage of tile t + 1 is fetched during the last stage of tile t and tile t's output stores drain under tile t + 1's
    // MFMAs (one block per CU: nobody else would cover the load latency in front of a tile and the stores behind it)
    const int t_beg = blockIdx.x * p.tpb, t_end = min(p.ntiles, t_beg + p.tpb);
    if (t_beg >= t_end || s_beg >= s_end) return;

    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, (short)0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, (short)0, (int)p.wp_bytes, 0x00020000);

    // ---- staging plans ---------------------------------------------------------------------------------------------------
    const int HWin = p.H * p.W;
    // halo item (tile independent part): it -> (sub, half, pixel); LDS offset [sub][piece][half][pixel] x 16 bytes
    int xl[NIT], xhy[NIT], xhx[NIT]; unsigned xc[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int it = tid + 512 * j;
        const int sh = it / NPIX, pix = it - sh * NPIX;                    // sh = sub * 2 + half
        xhy[j] = pix / WW; xhx[j] = pix - xhy[j] * WW;
        const bool in = it < NS * 2 * NPIX;
        xc[j] = S2D ? 0u : (unsigned)(sh * 8 * HWin);
        xl[j] = in ? (((sh >> 1) * 6 + (sh & 1)) * NPIX + pix) * 16 : -1;
    }
    unsigned xg[NIT];                                                       // global dword index of an item's first channel (tile dependent)
    int c_img, c_oy0, c_ox0;
    auto plan_tile = [&](int t) {
        const int tx = t % p.tiles_x; const int r = t / p.tiles_x;
        const int ty = r % p.tiles_y; c_img = r / p.tiles_y;
        c_oy0 = ty * R; c_ox0 = tx * CW;
        const unsigned x_img = (unsigned)c_img * p.CinX * HWin;
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            int iy = c_oy0 - pt + xhy[j], ix = c_ox0 - pl_ + xhx[j];
            if constexpr (S2D) {
                const int sh = (tid + 512 * j) / NPIX;
                iy = 2 * (c_oy0 + xhy[j]) + (sh >> 1) - 1; ix = 2 * (c_ox0 + xhx[j]) + (sh & 1) - 1;
            }
            const bool ok = xl[j] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            xg[j] = ok ? x_img + xc[j] + (unsigned)(iy * p.W + ix) : D2_OOB;
        }
    };
    // filters: unit q = tid + 512 i of the stage's BM x UPR image; the prepped layout is row-major with the same row length,
    // so its global index is linear in q
    const unsigned wq0 = (unsigned)wrow0 * UPR, wqmax = (unsigned)max(0, min(BM, p.Cout - m0)) * UPR;
    const unsigned wstage = (unsigned)rows_all * UPR * 4u;                 // dwords per stage of the prepped filters

    float rx[NIT][8]; f32x4 rw[NWU];
    auto load_stage = [&](int s) {
        const unsigned xb = (unsigned)s * (S2D ? 8 : 16 * NS) * HWin;
#pragma unroll
        for (int j = 0; j < NIT; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) rx[j][k] = ldg1(rX, xg[j] == D2_OOB ? D2_OOB : xg[j] + xb + k * HWin);
#pragma unroll
        for (int i = 0; i < NWU; ++i) {
            const unsigned q = (unsigned)tid + 512u * i;
            rw[i] = ldg4w(rW, q < wqmax ? (wq0 + q) * 4u + (unsigned)s * wstage : D2_OOB);
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            if (xl[j] >= 0) {
                const X6Frag f = x6_split8(rx[j]);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *(uint4*)(Xh + xl[j] + pl * 2 * NPIX * 16) = __builtin_bit_cast(uint4, f.p[pl]);
            }
        }
#pragma unroll
        for (int i = 0; i < NWU; ++i) {
            const int q = tid + 512 * i;
            if (q < BM * UPR) *(f32x4*)(Wl + q * 16 + (q / UPR) * 16) = rw[i];       // row * WROW + u * 16, WROW = (UPR + 1) * 16
        }
    };

    f32x16 acc[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    const unsigned char* const Ab = Wl + (lane & 31) * WROW + h * 16;
    const unsigned char* const Bb = Xh + (h * NPIX + prow * WW + pcol) * 16;
    float* __restrict__ const Yg = (p.nsplit > 1) ? (p.ws + (size_t)sp * p.slab) : p.Y;
    const bool addc = (p.nsplit == 1) && p.accumulate;
    const size_t plane = (size_t)p.yH * p.yW;

    plan_tile(t_beg);
    load_stage(s_beg);
    store_stage();
    __syncthreads();
    D2_T(t_k1); D2_ADD(0, t_k0, t_k1);
    for (int t = t_beg; t < t_end; ++t) {
        const int e_img = c_img, e_oy0 = c_oy0, e_ox0 = c_ox0;             // this tile's output coordinates (the plan moves on)
        const bool next_tile = t + 1 < t_end;
        for (int s = s_beg; s < s_end; ++s) {
            const bool last = s + 1 == s_end;
            D2_T(t_s0);
            if (!last) load_stage(s + 1);
            else if (next_tile) { plan_tile(t + 1); load_stage(s_beg); }
            // fragments of group g + 1 are requested in front of the MFMAs of group g (two register sets; the issue order is pinned:
            // left alone, the scheduler starts a group's reads only behind the previous group's last MFMA and every group begins
            // with an exposed LDS latency)
            X6Frag fa[2][TM], fb[2];
            auto load_frags = [&](int g, int buf) {
                const int sub = g / KHW, tap = g - sub * KHW, kh = tap / KW, kw = tap - kh * KW;   // compile time after unrolling
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fb[buf].p[pl] = __builtin_bit_cast(mma_bf16x8, *(const uint4*)(Bb + ((sub * 6 + pl * 2) * NPIX + kh * WW + kw) * 16));
#pragma unroll
                for (int tt = 0; tt < TM; ++tt)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        fa[buf][tt].p[pl] = __builtin_bit_cast(mma_bf16x8, *(const uint4*)(Ab + tt * 32 * WROW + (g * 6 + pl * 2) * 16));
            };
            load_frags(0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_frags(g + 1, (g + 1) & 1);
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int tt = 0; tt < TM; ++tt) acc[tt] = x6_mfma(fa[g & 1][tt], fb[g & 1], term, acc[tt]);
#if !defined(DCONV2_NOSCHED)
                if (g + 1 < NG) {
#pragma unroll
                    for (int q = 0; q < 3 * (TM + 1); ++q) {        // one LDS read behind every second MFMA, the rest of the MFMAs behind
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM - 2 * 3 * (TM + 1) > 0 ? 6 * TM - 2 * 3 * (TM + 1) : 0, 0);
                }
#endif
            }
            D2_T(t_s1);
            __syncthreads();
            D2_T(t_s2);
            if (!last || next_tile) { store_stage(); D2_T(t_s3); __syncthreads(); D2_T(t_s4); D2_ADD(3, t_s2, t_s3); D2_ADD(4, t_s3, t_s4); }
            D2_ADD(1, t_s0, t_s1); D2_ADD(2, t_s1, t_s2); D2_ADD(6, t_s0, t_s0 + 1);
        }
        D2_T(t_e0);
        // ---- epilogue: Y[img][m][(oy0 + prow) * ys + y0][(ox0 + pcol) * ys + x0]; the stores drain under the next tile's MFMAs ----
        const size_t pix = (size_t)((e_oy0 + prow) * p.ys + y0) * p.yW + (size_t)(e_ox0 + pcol) * p.ys + x0;
        float* const ybase = Yg + ((size_t)e_img * p.Cout + m0 + 4 * h) * plane + pix;
        if (m0 + BM <= p.Cout && !addc) {          // whole channel block, plain stores: no per-element branches
#pragma unroll
            for (int tt = 0; tt < TM; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ybase[(size_t)(tt * 32 + (r & 3) + 8 * (r >> 2)) * plane] = acc[tt][r];
                    acc[tt][r] = 0.f;
                }
        } else {
#pragma unroll
            for (int tt = 0; tt < TM; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mr = tt * 32 + (r & 3) + 8 * (r >> 2);
                    if (m0 + mr + 4 * h < p.Cout) {
                        float* dst = ybase + (size_t)mr * plane;
                        float v = acc[tt][r];
                        if (addc) v += *dst;
                        *dst = v;
                    }
                    acc[tt][r] = 0.f;
                }
        }
        D2_T(t_e1); D2_ADD(5, t_e0, t_e1);
    }
}

__global__ __launch_bounds__(256) void dconv2_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long long n,
                                                            long long slab, int nsplit, int acc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // fixed summation order (deterministic)
    if (i >= n) return;
    float s = acc ? out[i] : 0.f;
    for (int k = 0; k < nsplit; ++k) s += ws[(size_t)k * slab + i];
    out[i] = s;
}

static inline long long cdiv2(long long a, long long b) { return (a + b - 1) / b; }

template <int KH, int KW, int TM, int NS, int CW, bool S2D>
static int launch2g(D2P& p, hipStream_t st) {
    constexpr int KHW = KH * KW, NPIX = (256 / CW + KH - 1) * (CW + KW - 1);
    constexpr size_t lds = (size_t)NS * 6 * NPIX * 16 + (size_t)TM * 32 * ((size_t)NS * KHW * 6 + 1) * 16;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)dconv2_fwd_kernel<KH, KW, TM, NS, CW, S2D>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) return MOGAN_ERR_LAUNCH;
        attr = true;
    }
    dim3 grid((unsigned)cdiv2(p.ntiles, p.tpb), (unsigned)cdiv2(p.Cout, TM * 32), (unsigned)(p.nsplit * p.npar));
    hipLaunchKernelGGL((dconv2_fwd_kernel<KH, KW, TM, NS, CW, S2D>), grid, dim3(512), lds, st, p);
    return 0;
}
template <int KH, int KW, int TM, int NS>
static int launch2(D2P& p, hipStream_t st) {
    return p.tiles_cw == 32 ? launch2g<KH, KW, TM, NS, 32, false>(p, st) : launch2g<KH, KW, TM, NS, 16, false>(p, st);
}

// tile rows (tm) and sub-chunks per stage (ns) of a launch: functions of the channels and the form alone (not of the map size), so
// that a filter image can be prepared ahead of the call.  Cin / Cout as the kernel sees them (s2d: Cin = 4 x the tensor's channels)
static inline void d2_tile(bool s2d, bool k22, int Cin, int Cout, int& tm, int& ns) {
    tm = 4; long long best = cdiv2(Cout, 128) * 128;
    if (s2d) { tm = 3; best = cdiv2(Cout, 96) * 96; }
    else if (cdiv2(Cout, 96) * 96 < best) { best = cdiv2(Cout, 96) * 96; tm = 3; }
    if (cdiv2(Cout, 64) * 64 < best) { best = cdiv2(Cout, 64) * 64; tm = 2; }
    ns = s2d ? 2 : (k22 && tm <= 3 && Cin % 32 == 0) ? 2 : 1;
}

}  // namespace
#endif  // MOGAN_X6

// Filter images of n 4x4 s2 p1 weights w[i] (Cout[i], Cin[i], 4, 4) for dconv2_fwd_kernel: dgrad[i] = 0 the forward form (2x2 filter
// over the space-to-depth image), 1 the data-gradient form (four parity classes).  The layout parameters are the ones
// mogan_dconv2_fwd_try derives from the same channels (d2_tile), so an image fits every map size of that weight.
int mogan_dconv2_prep_group(int n, const float* const* w, void* const* prep, const int* Cout, const int* Cin, const int* dgrad,
                            hipStream_t st) {
#if MOGAN_X6
    for (int i0 = 0; i0 < n; i0 += D2G_MAX) {
        D2PrepGroup g{};
        g.n = std::min(D2G_MAX, n - i0);
        unsigned end = 0;
        for (int j = 0; j < g.n; ++j) {
            const int i = i0 + j;
            if (!w[i] || !prep[i] || Cout[i] <= 0 || Cin[i] <= 0 || (((uintptr_t)prep[i]) & 15)) return MOGAN_ERR_SHAPE;
            int tm, ns;
            long long rows; int kin, cout1, wmode;
            if (!dgrad[i]) {          // forward: rows = Cout, kernel channels 4 Cin (s2d), Cout1 = Ci of w
                if (Cin[i] % 8) return MOGAN_ERR_SHAPE;
                d2_tile(true, true, 4 * Cin[i], Cout[i], tm, ns);
                rows = Cout[i]; kin = 4 * Cin[i]; cout1 = Cin[i]; wmode = 3;
            } else {                  // data gradient: the convolution runs Co -> Ci per parity class: rows = 4 Ci, channels Co
                if (Cout[i] % 16) return MOGAN_ERR_SHAPE;
                d2_tile(false, true, Cout[i], Cin[i], tm, ns);
                rows = 4ll * Cin[i]; kin = Cout[i]; cout1 = Cin[i]; wmode = 2;
            }
            const int nst = kin / (16 * ns);
            g.w[j] = w[i]; g.wp[j] = (uint4*)prep[i]; g.rows[j] = (int)rows; g.Cin[j] = kin; g.NS[j] = ns; g.wmode[j] = wmode; g.Cout1[j] = cout1;
            g.total[j] = (long long)nst * rows * ns * 4 * 2;
            const long long blocks = (g.total[j] + 255) / 256;
            if (blocks <= 0 || (long long)end + blocks > 0x7fffffffLL) return MOGAN_ERR_SHAPE;
            end += (unsigned)blocks; g.end[j] = end;
        }
        hipLaunchKernelGGL(dconv2_wprep_group_kernel, dim3(end), dim3(256), 0, st, g);
    }
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
#else
    return n > 0 ? MOGAN_ERR_SHAPE : 0;
#endif
}

// ---- internal entry point (hidden): 1 = handled, 0 = not eligible (the caller takes dconv_fwd_kernel), < 0 = error ---------------
// X (B, Cin, H, W); the filters of the convolution to run come out of w as wmode says (dconv2_wprep_kernel: 0 = w is
// [npar][Cout][Cin][KH*KW] itself, 1 / 2 = w is the ORIGINAL filter tensor of a 3x3 s1 / 4x4 s2 convolution whose data gradient
// this call computes -- no flipped / parity-split copy is materialised, 3 = w is the filter tensor of a 4x4 s2 p1 convolution whose
// FORWARD this call computes: KH = KW = 4, pt = pl = 1, OH = H / 2); output grid OH x OW per parity class written at pixel
// stride ys into planes yH x yW.
int mogan_dconv2_fwd_try(const float* X, const float* w, int wmode, float* Y, int B, int Cin, int Cout, int H, int W, int OH, int OW,
                         int KH, int KW, int pt, int pl, int yH, int yW, int ys, int npar, int accumulate, void* ws,
                         size_t ws_bytes, hipStream_t st, const void* prep_in, size_t* query) {
#if MOGAN_X6
    // MOGAN_DCONV2: 0 = off, 1 = 8 x 32 tiles of stride-1 filters only (the first form of this kernel), 2 (default) = also the
    // 16 x 16 tiles and the space-to-depth forward of the 4x4 s2 convolutions
    constexpr int on = 2;
    if (!on || (on < 2 && (wmode == 3 || (OW % 32) != 0))) return 0;
    // wmode 3: the forward of a 4x4 s2 p1 convolution (KH = KW = 4, pt = pl = 1 on entry) = a 2 x 2 filter over the space-to-depth
    // image of the padded input, 4 Cin channels, no padding of its own
    const bool s2d = wmode == 3;
    if (s2d) {
        if (KH != 4 || KW != 4 || pt != 1 || pl != 1 || (H & 1) || (W & 1) || OH != H / 2 || OW != W / 2 || (Cin % 8) || npar != 1 ||
            ys != 1) return 0;
        KH = KW = 2; pt = pl = 0;
    }
    const int CinX = Cin;
    if (s2d) Cin *= 4;
    const bool k33 = KH == 3 && KW == 3, k22 = KH == 2 && KW == 2;
    // spatial tile: 8 rows x 32 columns, or 16 x 16 on the maps with 16-pixel rows
    const int cw = (OW % 32) == 0 ? 32 : 16, tr = 256 / cw;
    if (!(k33 || k22) || (OW % cw) || (OH % tr) || (Cin % 16) || Cout < 64 || B <= 0) return 0;
    if ((long long)B * CinX * H * W >= (1ll << 29) || (long long)B * Cout * yH * yW >= (1ll << 30)) return 0;
    const int KHW = KH * KW;
    // BM: least padded rows of 64 / 96 / 128, the larger tile on a tie (space-to-depth form: 64 / 96, its stage is two sub-chunks);
    // stage = NS x 16 channels: the 2 x 2 filters take two sub-chunks per stage (8 groups between barriers) where LDS allows
    int tm, ns; d2_tile(s2d, k22, Cin, Cout, tm, ns);
    const int nst = Cin / (16 * ns);
    const long long rows = (long long)npar * Cout;
    const size_t wpb = (size_t)nst * rows * ns * KHW * 6 * 16;
    if (wpb >= (1ull << 31)) return 0;
    const bool owned = wmode == 2 || wmode == 3;          // the forms whose image the weight's owner may hold (4x4 s2 p1)
    if (query) { *query = owned ? wpb : 0; return 1; }   // dry run of the dispatch (mogan_conv_prep_bytes)
    const void* prep = owned ? prep_in : nullptr;
    if (!prep) {
        if (!ws || ws_bytes < wpb + 256) return 0;
        const long long total = (long long)nst * rows * ns * KHW * 2;
        hipLaunchKernelGGL(dconv2_wprep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, (uint4*)ws, (int)rows, Cin,
                           KHW, ns, wmode, s2d ? CinX : Cout, total);
    }
    D2P p{};
    p.X = X; p.Wp = prep ? prep : ws; p.Y = Y; p.B = B; p.Cin = Cin; p.CinX = CinX; p.Cout = Cout; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.pt = pt; p.pl = pl;
    p.yH = yH; p.yW = yW; p.ys = ys; p.npar = npar; p.accumulate = accumulate;
    p.tiles_cw = cw; p.tiles_x = OW / cw; p.tiles_y = OH / tr;
    p.x_bytes = 4u * (unsigned)B * CinX * H * W; p.wp_bytes = (unsigned)wpb;
    const size_t adv = prep ? 0 : ((wpb + 255) & ~(size_t)255);
    char* ws2 = (char*)ws + adv; size_t ws2_bytes = ws ? ws_bytes - adv : 0;
    p.ntiles = B * p.tiles_x * p.tiles_y;
    const long long groups = cdiv2(Cout, tm * 32) * npar;              // (channel block, parity class) pairs: each walks all spatial tiles
    const long long tiles = (long long)p.ntiles * groups;
    const long long y_numel = (long long)B * Cout * yH * yW;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    int nsplit = 1;
    // one block per CU: with fewer tiles than ~3/4 of the CUs the K range is split over the stages
    if (tiles < (3 * ncu) / 4 && nst >= 4) nsplit = (int)std::min<long long>(cdiv2(ncu, tiles), nst / 2);
    if (nsplit > 1) {
        const long long fit = (long long)(ws2_bytes / (sizeof(float) * (size_t)y_numel));
        nsplit = fit < 2 ? 1 : (int)std::min<long long>(nsplit, fit);
    }
    p.sps = (int)cdiv2(nst, nsplit); p.nsplit = (int)cdiv2(nst, p.sps);
    p.slab = y_numel; p.ws = (float*)ws2;
    // persistent blocks: about one per CU over all (channel block, parity, split) groups, every block the same number of tiles
    {
        const long long per_group = std::max<long long>(1, ncu / (groups * p.nsplit));
        p.tpb = (int)cdiv2(p.ntiles, std::min<long long>(per_group, p.ntiles));
    }
    if (tiles * p.nsplit > 0x7fffffffLL) return 0;
    mogan_prof_relabel(2);
    int rc = 0;
    if (s2d) {
        if (cw == 32) rc = tm == 3 ? launch2g<2, 2, 3, 2, 32, true>(p, st) : launch2g<2, 2, 2, 2, 32, true>(p, st);
        else rc = tm == 3 ? launch2g<2, 2, 3, 2, 16, true>(p, st) : launch2g<2, 2, 2, 2, 16, true>(p, st);
    } else if (k33) {
        if (tm == 4) rc = launch2<3, 3, 4, 1>(p, st); else if (tm == 3) rc = launch2<3, 3, 3, 1>(p, st); else rc = launch2<3, 3, 2, 1>(p, st);
    } else if (ns == 2) {
        if (tm == 3) rc = launch2<2, 2, 3, 2>(p, st); else rc = launch2<2, 2, 2, 2>(p, st);
    } else {
        if (tm == 4) rc = launch2<2, 2, 4, 1>(p, st); else if (tm == 3) rc = launch2<2, 2, 3, 1>(p, st); else rc = launch2<2, 2, 2, 1>(p, st);
    }
    if (rc) return rc;
    if (p.nsplit > 1)
        hipLaunchKernelGGL(dconv2_reduce_kernel, dim3((unsigned)cdiv2(y_numel, 256)), dim3(256), 0, st, (const float*)ws2, Y, y_numel,
                           y_numel, p.nsplit, accumulate);
    return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
#else
    return 0;
#endif
}

