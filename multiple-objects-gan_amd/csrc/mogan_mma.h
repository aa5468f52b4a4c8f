// fp32 matrix products on the bf16 matrix pipe of gfx950 (device-side helpers shared by the MFMA kernels).
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD, 157 TFLOP/s); v_mfma_f32_32x32x16_bf16 at
// 16x that.  Every fp32 number is the exact sum of three bf16 numbers (8 significant bits each: x1 = bf16(x), x2 = bf16(x - x1),
// x3 = x - x1 - x2, round to nearest; the remainders are exact and x3 needs at most 8 bits), so
//     a * b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + [a2 b3 + a3 b2 + a3 b3]
// and the bracket is at most 2^-23 |a b| (|x2| <= 2^-8 |x|, |x3| <= 2^-16 |x|; typically 2^-25), i.e. of the size of the rounding of one fp32 fma.  Each bf16 product is exact in fp32,
// the accumulators are fp32, the partial products are added smallest first: the six-product form carries fp32-level
// error (measured against fp64 next to the native fp32-MFMA form: tools/diag_x6_precision.py)
// at 6/16 of its matrix-pipe time.  MOGAN_X6=0 compiles the native fp32-MFMA form of every kernel instead.
//
// One call = 16 k-values of a 32x32 tile product: in the native form eight v_mfma_f32_32x32x2_f32 steps, lane (row,
// h = lane >> 5) supplying one A and one B value per step; in the split form the same eight values of the lane become
// the lane's 8-element operand of ONE v_mfma_f32_32x32x16_bf16 (which k a lane slot stands for is free as long as A
// and B agree, so the LDS layouts and operand reads of the kernels are the same in both forms).
#ifndef MOGAN_MMA_H
#define MOGAN_MMA_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef MOGAN_X6
#define MOGAN_X6 1
#endif

typedef float mma_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 mma_bf16x8 __attribute__((ext_vector_type(8)));

struct X6Frag { mma_bf16x8 p[3]; };      // the three bf16 pieces of a lane's 8 operand values

// two fp32 -> three dwords, each holding the bf16 pieces of (x0 | x1 << 16).  Round-to-nearest pieces (v_cvt_pk_bf16_f32):
// |x2| <= 2^-8 |x|, |x3| <= 2^-16 |x|; both remainders are exact in fp32 and the last one fits 8 bits.
typedef float mma_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 mma_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x6_split2(float x0, float x1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    const mma_f32x2 x = {x0, x1};
    p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, mma_bf16x2));
    const mma_f32x2 r = {x0 - __uint_as_float(p1 << 16), x1 - __uint_as_float(p1 & 0xFFFF0000u)};
    p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, mma_bf16x2));
    const mma_f32x2 s = {r[0] - __uint_as_float(p2 << 16), r[1] - __uint_as_float(p2 & 0xFFFF0000u)};
    p3 = __builtin_bit_cast(uint32_t, __builtin_convertvector(s, mma_bf16x2));
}

__device__ __forceinline__ X6Frag x6_split8(const float (&x)[8]) {
    uint32_t w[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x6_split2(x[2 * i], x[2 * i + 1], w[0][i], w[1][i], w[2][i]);
    X6Frag f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint4 u = make_uint4(w[i][0], w[i][1], w[i][2], w[i][3]);
        f.p[i] = __builtin_bit_cast(mma_bf16x8, u);
    }
    return f;
}

// piece indices of the six partial products, smallest first: (3,1) (2,2) (1,3) (2,1) (1,2) (1,1)
__device__ __forceinline__ constexpr int x6_ia(int term) { return term == 0 ? 2 : term == 1 ? 1 : term == 2 ? 0 : term == 3 ? 1 : 0; }
__device__ __forceinline__ constexpr int x6_ib(int term) { return term == 0 ? 0 : term == 1 ? 1 : term == 2 ? 2 : term == 3 ? 0 : term == 4 ? 1 : 0; }

__device__ __forceinline__ mma_f32x16 x6_mfma(const X6Frag& a, const X6Frag& b, int term, mma_f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[x6_ia(term)], b.p[x6_ib(term)], acc, 0, 0, 0);
}

// XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (own L2 each, 4 MB) by their linear index L; this maps L to
// a logical index such that XCD k runs a CONTIGUOUS range of the logical sequence, in order: blocks that are neighbours in the logical
// order (share an operand slice) sit on one XCD at the same time, and the slice is one L2 miss for all of them.
__device__ __forceinline__ unsigned mma_xcd_order(unsigned L, unsigned total) {
    const unsigned k = L & 7u, j = L >> 3, q = total >> 3, r = total & 7u;
    return k * q + (k < r ? k : r) + j;
}
// the logical (x, y, z) of a block of a 3-D grid under that order, x fastest
__device__ __forceinline__ void mma_xcd_block(unsigned& bx, unsigned& by, unsigned& bz) {
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    unsigned V = mma_xcd_order(L, gridDim.x * gridDim.y * gridDim.z);
    bx = V % gridDim.x; V /= gridDim.x;
    by = V % gridDim.y; bz = V / gridDim.y;
}

// acc[ta][tb] += sum over the 16 k-values (8 per lane half) of a[ta][.] * b[tb][.]
template <int TM, int TN>
__device__ __forceinline__ void mma_k16(const float (&a)[TM][8], const float (&b)[TN][8], mma_f32x16 (&acc)[TM][TN]) {
#if MOGAN_X6
    X6Frag fa[TM], fb[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) fa[t] = x6_split8(a[t]);
#pragma unroll
    for (int t = 0; t < TN; ++t) fb[t] = x6_split8(b[t]);
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
        for (int ta = 0; ta < TM; ++ta)
#pragma unroll
            for (int tb = 0; tb < TN; ++tb) acc[ta][tb] = x6_mfma(fa[ta], fb[tb], term, acc[ta][tb]);
#else
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int ta = 0; ta < TM; ++ta)
#pragma unroll
            for (int tb = 0; tb < TN; ++tb)
                acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta][s], b[tb][s], acc[ta][tb], 0, 0, 0);
#endif
}
#endif
