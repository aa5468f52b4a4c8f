// mogan_wino.hip -- Winograd F(2x2, 3x3) for the 3x3 stride-1 pad-1 convolutions of the generator's ResBlocks
// (model.py:67-81: conv3x3 C -> 2C and C -> C at 64x64 and 128x128, 21 % of the step's multiply-adds), fused in ONE kernel:
//   Y = A^t [ sum_ci (G g G^t) .* (B^t d B) ] A         g = 3x3 filter, d = 4x4 input tile, Y = 2x2 outputs
// 16 multiplies per (ci, co, 2x2 outputs) instead of 36: 2.25x fewer MFMA flops, no intermediate in HBM.
//
// A block owns 96 output channels x 32 tiles (2 tile rows x 16 tile columns = 4 x 32 output pixels) of one image.  Per chunk
// of 8 input channels it stages the 6 x 34 input halo (Xs), transforms it in registers (one thread = one (channel, tile):
// 16 LDS reads, 32 add/sub) into Vs[xi][c][tile], stages the pre-transformed weights Us[xi][c][co] (global layout
// U[xi][ci][co], 16-byte loads and stores), and runs the 16 small GEMMs  M_xi[co][tile] += U_xi[co][c] V_xi[c][tile]  on
// v_mfma_f32_32x32x2_f32: wave w owns the four positions xi = 4w..4w+3 (row w of the 4x4 grid), 4 x 3 accumulator tiles =
// 192 registers.  Both operand reads are ds_read_b32 with the lanes on consecutive dwords (conflict-free).  Epilogue: the
// row transform M A is local to a wave, the column transform A^t (.) goes through LDS (one pass per output column parity).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mogan_hip.h"
#include "mogan_internal.h"

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

// U[xi][ci][co] = (G g G^t)[xi],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].  flip = 1: the data-gradient filter
// g'(ci' = co, co' = ci) = rot180(w[co][ci]), i.e. the kernel then maps dY (Cout channels) to dX (Cin channels).
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout,
                                                          int Cin, int flip) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)Cout * Cin) return;
    const int co = (int)(i / Cin), ci = (int)(i - (long long)co * Cin);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = flip ? w[i * 9 + (2 - a) * 3 + (2 - b)] : w[i * 9 + a * 3 + b];
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int kin = flip ? co : ci, kout = flip ? ci : co;           // reduction / output channel of the kernel
    const int Kin = flip ? Cout : Cin, Kout = flip ? Cin : Cout;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]),
                    u3 = t[a][2];
        float* d = U + ((size_t)(a * 4) * Kin + kin) * Kout + kout;
        d[0] = u0; d[(size_t)Kin * Kout] = u1; d[(size_t)2 * Kin * Kout] = u2; d[(size_t)3 * Kin * Kout] = u3;
    }
}

__global__ __launch_bounds__(512) void wino_fwd_kernel(const float* __restrict__ X, const float* __restrict__ U,
                                                       float* __restrict__ Y, int Cin, int H, int W, int Cout, int tiles_x,
                                                       int tiles_y, unsigned x_bytes, unsigned u_bytes) {
    constexpr int XSZ = CK * XR * XCP, VSZ = 16 * CK * NT, USZ = 16 * CK * BM;
    __shared__ __attribute__((aligned(16))) float Xs[2 * XSZ + 4];      // + a dump slot for the idle staging lanes
    __shared__ __attribute__((aligned(16))) float Vs[2 * VSZ];
    __shared__ __attribute__((aligned(16))) float Us[2 * USZ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int img = t / tiles_y;
    const int m0 = blockIdx.y * BM;
    const int oy0 = ty * 2 * TROWS, ox0 = tx * 2 * TCOLS;
    const int plane = H * W;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)U, (short)0, (int)u_bytes, 0x00020000);

    // 8 waves: wave w owns the two positions xi = 2w, 2w+1 (row w>>1 of the 4x4 grid, columns 2(w&1), 2(w&1)+1)
    f32x16 acc[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][a][r] = 0.f;

    // staging plans (chunk-invariant parts)
    constexpr int NXE = (CK * XR * XC + 511) / 512;         // 4
    int xl[NXE]; unsigned xg[NXE];
#pragma unroll
    for (int i = 0; i < NXE; ++i) {
        const int e = tid + 512 * i;
        const int c = e / (XR * XC), r = e - c * (XR * XC);
        const int hy = r / XC, hx = r - hy * XC;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        const bool ok = e < CK * XR * XC && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        xl[i] = e < CK * XR * XC ? c * (XR * XCP) + hy * XCP + hx : 2 * XSZ;      // idle lanes: dump slot (branch-free stores)
        xg[i] = ok ? (unsigned)(img * Cin + c) * plane + (unsigned)(iy * W + ix) : 0x30000000u;   // *4 >= 2 GiB: reads 0
    }
    constexpr int NUQ = 16 * CK * (BM / 4) / 512;            // 6
    int ul[NUQ]; unsigned ug[NUQ];
#pragma unroll
    for (int i = 0; i < NUQ; ++i) {
        const int q = tid + 512 * i;
        const int row = q / (BM / 4), m4 = 4 * (q - row * (BM / 4));      // row = xi*CK + c
        const int xi = row / CK, c = row - xi * CK;
        ul[i] = row * BM + m4;
        ug[i] = m0 + m4 < Cout ? (unsigned)(xi * Cin + c) * Cout + (unsigned)(m0 + m4) : 0x30000000u;   // Cout % 4 == 0
    }
    // transform role: (channel, tile, half): the two threads of a pair compute rows {0,1} / {2,3} of B^t d B
    const int th = tid >> 8, tc = (tid >> 5) & 7, tt = tid & 31, tty = tt >> 4, ttx = tt & 15;

    // Software pipeline, ONE barrier per chunk, everything double buffered (144 KB of LDS, one 8-wave block per CU):
    // while the MFMAs of chunk c run on Us/Vs[c&1], the same iteration stores U(c+1) (loaded an iteration ago), transforms
    // X(c+1) (stored an iteration ago) into Vs[(c+1)&1], stores X(c+2) and issues the loads of U(c+2) and X(c+3).
    float rx[NXE]; f32x4 ru[NUQ];
    auto load_x = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NXE; ++i) rx[i] = ldgx(rX, xg[i] + (unsigned)c0 * plane, true);
    };
    auto load_u = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NUQ; ++i) ru[i] = ldgx4(rU, ug[i] + (unsigned)c0 * Cout, true);
    };
    auto store_x = [&](float* Xd) {
#pragma unroll
        for (int i = 0; i < NXE; ++i) Xd[xl[i] < 2 * XSZ ? xl[i] : 2 * XSZ - (int)(Xd - Xs)] = rx[i];
    };
    auto store_u = [&](float* Ud) {
#pragma unroll
        for (int i = 0; i < NUQ; ++i) *(f32x4*)&Ud[ul[i]] = ru[i];
    };
    auto transform = [&](const float* Xc, float* Vd) {      // rows 2th, 2th+1 of V = B^t d B for (channel tc, tile tt)
        // th = 0 needs the tile rows 0,1,2 (u0 = d0 - d2, u1 = d1 + d2), th = 1 the rows 1,2,3 (u2 = d2 - d1, u3 = d1 - d3):
        // both read three rows starting at row th -- no divergent loads, the variants differ by selects on values
        float ra[4], rb[4], rc[4];
        const float* px = &Xc[tc * (XR * XCP) + (2 * tty + th) * XCP + 2 * ttx];
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
            float* pv = &Vd[(((2 * th + i) * 4) * CK + tc) * NT + tt];
            pv[0] = u[i][0] - u[i][2];
            pv[CK * NT] = u[i][1] + u[i][2];
            pv[2 * CK * NT] = u[i][2] - u[i][1];
            pv[3 * CK * NT] = u[i][1] - u[i][3];
        }
    };
    const int nchunk = Cin / CK;
    load_x(0); load_u(0);
    store_x(Xs); store_u(Us);
    load_x(CK); load_u(CK);
    __syncthreads();
    transform(Xs, Vs);
    store_x(Xs + XSZ);
    load_x(2 * CK);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1, nxt = cur ^ 1;
        const float* Uc = Us + cur * USZ;
        const float* Vc = Vs + cur * VSZ;
        // operands of this chunk -> registers (Us/Vs[cur] are complete since the barrier that ended the last iteration)
        float av[2][CK / 2][3], bv[2][CK / 2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < CK / 2; ++kk) {
                const int row = (wave * 2 + j) * CK + 2 * kk + h;
                bv[j][kk] = Vc[row * NT + l31];
#pragma unroll
                for (int a = 0; a < 3; ++a) av[j][kk][a] = Uc[row * BM + a * 32 + l31];
            }
        // the preparation of chunk c+1 (and the loads two / three chunks ahead) has no dependence on the MFMAs of chunk c:
        // the issue-order template below spreads it between them
        store_u(Us + nxt * USZ);                            // U(c+1)
        store_x(Xs + cur * XSZ);                            // X(c+2); Xs[cur] was last read by transform(c), an iteration ago
        load_u((c + 2) * CK); load_x((c + 3) * CK);
        transform(Xs + nxt * XSZ, Vs + nxt * VSZ);          // X(c+1) -> V(c+1)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < CK / 2; ++kk)
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    acc[j][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][kk][a], bv[j][kk], acc[j][a], 0, 0, 0);
        // issue order: the 16 operand reads of j = 0; then behind every MFMA a slice of the rest
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // operand reads of j = 1 (16 over the first 8 MFMAs)
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // LDS stores of U(c+1) / X(c+2)
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // global loads
        }
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (g < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // transform: halo reads
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // transform: add / sub
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // transform: V stores
        }
        __syncthreads();
    }

    // ---- output transform.  Row i = wave>>1 of M: T_i[b] = sum_j M[i][j] A[j][b] with A^t = [[1,1,1,0],[0,1,-1,-1]] is split
    // over the wave pair (columns {0,1} / {2,3}); Y[a][b] = sum_i A^t[a][i] T_i[b] goes through LDS, one pass per b.
    float yreg[6][2][2];
    float* Ts = Us;                                          // [4][96][32]
    const int wi = wave >> 1, wj = wave & 1;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        __syncthreads();
        float part[3][16];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                part[a][r] = wj == 0 ? (b == 0 ? acc[0][a][r] + acc[1][a][r] : acc[1][a][r])
                                     : (b == 0 ? acc[0][a][r] : -acc[0][a][r] - acc[1][a][r]);
        if (wj == 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ts[(wi * BM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * NT + l31] = part[a][r];
        }
        __syncthreads();
        if (wj == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* p = &Ts[(wi * BM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * NT + l31];
                    *p = *p + part[a][r];
                }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int idx = tid + 512 * q;
            const float t0 = Ts[idx], t1 = Ts[BM * NT + idx], t2 = Ts[2 * BM * NT + idx], t3 = Ts[3 * BM * NT + idx];
            yreg[q][0][b] = t0 + t1 + t2;
            yreg[q][1][b] = t1 - t2 - t3;
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int idx = tid + 512 * q;
        const int m = idx >> 5, tile = idx & 31;
        const int oy = oy0 + 2 * (tile >> 4), ox = ox0 + 2 * (tile & 15);
        if (m0 + m < Cout && oy < H && ox < W) {
            float* o = Y + ((size_t)(img * Cout + m0 + m) * H + oy) * W + ox;
            *(float2*)o = make_float2(yreg[q][0][0], yreg[q][0][1]);
            *(float2*)(o + W) = make_float2(yreg[q][1][0], yreg[q][1][1]);
        }
    }
}

}  // namespace

extern "C" {

// lab entry points (not part of include/mogan_hip.h): U (16, Cin, Cout) floats from w (Cout, Cin, 3, 3); y = conv3x3 p1 (x)
int mogan_lab_wino_weights(const float* w, float* U, int Cout, int Cin, int flip, hipStream_t st) {
    const long long n = (long long)Cout * Cin;
    hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, U, Cout, Cin, flip);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

int mogan_lab_wino_fwd(const float* x, const float* U, float* y, int B, int Cin, int H, int W, int Cout, hipStream_t st) {
    if ((Cin % CK) || (Cout % 4) || (H % (2 * TROWS)) || (W % (2 * TCOLS))) return MOGAN_ERR_SHAPE;
    if ((long long)B * Cin * H * W >= (1ll << 30) || (long long)16 * Cin * Cout >= (1ll << 30)) return MOGAN_ERR_SHAPE;
    const int tiles_x = W / (2 * TCOLS), tiles_y = H / (2 * TROWS);
    dim3 grid((unsigned)(B * tiles_x * tiles_y), (unsigned)((Cout + BM - 1) / BM));
    hipLaunchKernelGGL(wino_fwd_kernel, grid, dim3(512), 0, st, x, U, y, Cin, H, W, Cout, tiles_x, tiles_y,
                       4u * B * Cin * H * W, 4u * 16 * Cin * Cout);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

}  // extern "C"
