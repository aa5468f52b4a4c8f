// mogan_stem.hip -- the first convolution of every discriminator, nn.Conv2d(3, ndf, 4, 2, 1) -> LeakyReLU(0.2) (model.py:597-598,
// 660-661; S/model.py:250-251), as ONE streaming kernel (round 5).
//
// The layer is 48 multiply-adds per output value and writes the largest map of the network (B=16, 256x256 input: 100 MB out, 12.6 MB
// in): HBM-write-bound at ~25 us, while the implicit-GEMM kernel needed 71-82 us for it (a 128-wide N tile per block, K = 48 padded
// to its K tile, an LDS round trip per operand) and a one-thread-per-pixel VALU kernel 92 us (profiles/r05_ab.txt).  Here:
//   * a wave owns 32 consecutive output pixels of one output row and ALL output channels (TM x 32 rows): 3 channels x 6 partial
//     products x TM MFMAs (v_mfma_f32_32x32x16_bf16, one 16-k group = the 4 x 4 taps of one input channel), nothing else to reduce;
//   * the B operand of a lane (pixel, k half h) = rows 2 oy + 2h - 1, + 0 / 1 and columns 2 ox - 1 .. 2 ox + 2 of the image: six
//     8-byte loads per channel straight from global memory (the 12.6 MB input lives in L2), split into bf16 pieces in registers;
//   * the filters are split once per block into LDS in fragment order As[c][piece][h][row] (conflict-free 16-byte reads);
//   * LeakyReLU on the accumulators, 128-byte row segments per store instruction (buffer stores: the row's plane offset is a
//     scalar, no address registers); the next group's loads are in flight under the
//     MFMAs and stores of the current one (two register sets; eight waves per CU).
// B = 16, 256x256: 33.7 us (3.4 TB/s of output) against 73.3 us on the implicit-GEMM kernel; 128x128: 16.3 against 22.8 (tools/check_stem.py).
// Split-bf16 build only (MOGAN_X6); same arithmetic as every other MFMA kernel of that build (mogan_mma.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include "../../include/mogan_hip.h"
#include "mogan_internal.h"
#include "mogan_mma.h"

#if MOGAN_X6
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int TM>
__global__ __launch_bounds__(256, 2) void stem_k4s2_lrelu_kernel(const float* __restrict__ X, const float* __restrict__ w,
                                                              float* __restrict__ Y, int H, int W, int Cout, int OH, int OW,
                                                              float slope, int ngroups, unsigned x_bytes, unsigned y_bytes) {
    constexpr int ROWS = TM * 32;
    __shared__ __attribute__((aligned(16))) uint4 As[3 * 3 * 2 * ROWS];          // [c][piece][h][row]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    for (int e = tid; e < ROWS * 3 * 2; e += 256) {
        const int hh = e & 1, c = (e >> 1) % 3, row = e / 6;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = row < Cout ? w[((size_t)row * 3 + c) * 16 + 8 * hh + i] : 0.f;
        const X6Frag f = x6_split8(v);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) As[((c * 3 + pc) * 2 + hh) * ROWS + row] = __builtin_bit_cast(uint4, f.p[pc]);
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)Y, (short)0, (int)y_bytes, 0x00020000);
    const int gpr = OW >> 5, nwaves = gridDim.x * 4;
    const int plane = OH * OW;
    f32x2 va[3][2][3], vb[3][2][3];
    // the six 8-byte pairs (columns 2 ox - 2 + 2 q, + 1) of the two image rows of this lane's k half, per channel; a pair is
    // either inside the row or wholly outside (W and the pair's first column are even): outside -> an offset past the buffer -> 0
    auto load = [&](int g, f32x2 (&v)[3][2][3]) {
        const int ox = ((g % gpr) << 5) + l31; const int r = g / gpr;
        const int oy = r % OH, b = r / OH;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int iy = 2 * oy + 2 * h + rr - 1;
            const bool rok = (unsigned)iy < (unsigned)H;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ix = 2 * ox - 2 + 2 * q;
                const bool ok = rok && (unsigned)ix < (unsigned)W;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const unsigned idx = ((unsigned)(b * 3 + c) * H + iy) * W + ix;
                    v[c][rr][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rX, ok ? idx * 4u : 0xFFFFFFF8u, 0, 0));
                }
            }
        }
    };
    auto compute = [&](int g, const f32x2 (&v)[3][2][3]) {
        f32x16 acc[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float b8[8] = {v[c][0][0][1], v[c][0][1][0], v[c][0][1][1], v[c][0][2][0],
                                 v[c][1][0][1], v[c][1][1][0], v[c][1][1][1], v[c][1][2][0]};
            const X6Frag fb = x6_split8(b8);
            X6Frag fa[TM];
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    fa[t].p[pc] = __builtin_bit_cast(mma_bf16x8, As[((c * 3 + pc) * 2 + h) * ROWS + t * 32 + l31]);
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int t = 0; t < TM; ++t) acc[t] = x6_mfma(fa[t], fb, term, acc[t]);
        }
        const int ox = ((g % gpr) << 5) + l31; const int r = g / gpr;
        const int oy = r % OH, b = r / OH;
        // row m = 32 t + (r & 3) + 8 (r >> 2) + 4 h: the lane part of the address (pixel, 4 h planes) rides in the vector offset, the
        // row part is uniform -> scalar offset: no per-row address registers (Cout % 8 == 0: rows m and m + 4 are valid together)
        const unsigned vo = (((unsigned)(b * Cout) * OH + oy) * OW + ox + 4u * h * (unsigned)plane) * 4u;
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) {
                const int m = t * 32 + (r16 & 3) + 8 * (r16 >> 2);
                float s = acc[t][r16];
                s = s > 0.f ? s : s * slope;
                if (m < Cout) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s), rY, vo, (unsigned)m * (unsigned)plane * 4u, 0);
            }
    };
    int g = blockIdx.x * 4 + wave;
    if (g < ngroups) load(g, va);
    while (g < ngroups) {                                   // two groups per trip: the register sets alternate without copies
        const int g1 = g + nwaves;
        if (g1 < ngroups) load(g1, vb);
        compute(g, va);
        if (g1 >= ngroups) break;
        const int g2 = g1 + nwaves;
        if (g2 < ngroups) load(g2, va);
        compute(g1, vb);
        g = g2;
    }
}

}  // namespace
#endif  // MOGAN_X6

// ---- internal entry point (hidden): 1 = handled, 0 = not eligible, < 0 = error ----------------------------------------------------
// y (B, Cout, H/2, W/2) = LeakyReLU_slope(conv4x4 s2 p1 (x (B, 3, H, W), w (Cout, 3, 4, 4))); slope = 1 is the plain convolution.
int mogan_stem_fwd_try(const float* x, const float* w, float* y, int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride,
                       int ph, int pw, float slope, hipStream_t st) {
#if MOGAN_X6
    constexpr int on = 1;
    if (!on) return 0;
    if (!(Cin == 3 && KH == 4 && KW == 4 && stride == 2 && ph == 1 && pw == 1) || (H & 1) || (W & 1) || B <= 0) return 0;
    const int OH = H / 2, OW = W / 2;
    if ((OW & 31) || Cout < 32 || Cout > 128 || (Cout & 7) || (((uintptr_t)x) & 7) || !(slope > 0.f)) return 0;
    if ((long long)B * 3 * H * W >= (1ll << 29) || (long long)B * Cout * OH * OW >= (1ll << 30)) return 0;
    const long long ngroups = (long long)B * OH * (OW / 32);
    if (ngroups > 0x7fffffffLL) return 0;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    const int tm = (Cout + 31) / 32;
    const unsigned nblk = (unsigned)std::min<long long>((ngroups + 3) / 4, 2ll * ncu);       // two blocks = eight waves per CU
    const unsigned xb = 4u * (unsigned)B * 3u * (unsigned)H * (unsigned)W, yb = 4u * (unsigned)B * Cout * OH * OW;
#define MOGAN_STEM_CASE(T) case T: hipLaunchKernelGGL((stem_k4s2_lrelu_kernel<T>), dim3(nblk), dim3(256), 0, st, x, w, y, H, W, Cout, OH, \
                                                      OW, slope, (int)ngroups, xb, yb); break;
    switch (tm) {
        MOGAN_STEM_CASE(1)
        MOGAN_STEM_CASE(2)
        MOGAN_STEM_CASE(3)
        MOGAN_STEM_CASE(4)
        default: return 0;
    }
#undef MOGAN_STEM_CASE
    return hipGetLastError() == hipSuccess ? 1 : MOGAN_ERR_LAUNCH;
#else
    return 0;
#endif
}
