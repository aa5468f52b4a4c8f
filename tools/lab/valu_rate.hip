// Lab: issue cost of the vector instructions of the bf16 split (mogan_mma.h: x6_split2) on gfx950, alone and beside MFMAs of the
// partner wave.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// OP: 0 v_cvt_pk_bf16_f32, 1 v_pk_add_f32, 2 v_pk_fma_f32, 3 v_and_b32, 4 v_lshlrev_b32, 5 v_fma_f32, 6 v_mov_b32, 7 v_add_f32,
//     8 the split of a pair (11 instructions, dependent chain), 8 independent chains
template <int OP>
__global__ __launch_bounds__(512) void k_valu(float* out, int iters) {
    f32x2 p[8]; uint32_t u[8];
    for (int i = 0; i < 8; ++i) { p[i] = f32x2{(float)threadIdx.x + i, 1.5f * i}; u[i] = threadIdx.x * 2654435761u + i; }
    const f32x2 c = {1.0001f, 0.9999f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(p[i][0]), "v"(p[i][1]));
                if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c));
                if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c));
                if (OP == 3) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i]));
                if (OP == 4) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));
                if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(p[i][0]) : "v"(c[0]));
                if (OP == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(p[i][0]));
                if (OP == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(p[i][0]) : "v"(c[0]));
                if (OP == 8) {
                    uint32_t p1, p2, p3; f32x2 hi, r, s;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1) : "v"(p[i][0]), "v"(p[i][1]));
                    asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(hi[0]) : "v"(p1));
                    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hi[1]) : "v"(p1));
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(p[i]), "v"(hi));
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p2) : "v"(r[0]), "v"(r[1]));
                    asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(hi[0]) : "v"(p2));
                    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hi[1]) : "v"(p2));
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(s) : "v"(r), "v"(hi));
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p3) : "v"(s[0]), "v"(s[1]));
                    u[i] ^= p1 ^ p2 ^ p3;
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves 0..3 (one per SIMD): MFMAs only; waves 4..7: the split chains only (MODE 2), or nothing (MODE 1), or MFMAs too (MODE 3);
// MODE 0: waves 0..3 idle, 4..7 split
template <int MODE, int PRIO = 0>
__global__ __launch_bounds__(512) void k_mix(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    if (PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(3);          // the vector waves above the matrix waves
    if (PRIO == 2 && wave < 4) __builtin_amdgcn_s_setprio(3);           // the matrix waves above the vector waves
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    uint4 uu = make_uint4(threadIdx.x * 2654435761u, 0x3f803f80u, 0x40004000u, threadIdx.x * 40503u);
    bf16x8 x = __builtin_bit_cast(bf16x8, uu), y = x;
    f32x2 p[8]; uint32_t u[8];
    for (int i = 0; i < 8; ++i) { p[i] = f32x2{(float)threadIdx.x + i, 1.5f * i}; u[i] = i; }
    const bool do_mfma = (wave < 4 && MODE != 0) || (wave >= 4 && MODE == 3);
    const bool do_split = wave >= 4 && (MODE == 0 || MODE == 2);
    if (do_mfma) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 6; ++rep)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);     // 24 MFMAs = 768 cycles
    } else if (do_split) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) {                                                                        // 16 x 11 = 176 VALU
                    uint32_t p1, p2, p3; f32x2 hi, r, s;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1) : "v"(p[i][0]), "v"(p[i][1]));
                    asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(hi[0]) : "v"(p1));
                    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hi[1]) : "v"(p1));
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(p[i]), "v"(hi));
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p2) : "v"(r[0]), "v"(r[1]));
                    asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(hi[0]) : "v"(p2));
                    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hi[1]) : "v"(p2));
                    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(s) : "v"(r), "v"(hi));
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p3) : "v"(s[0]), "v"(s[1]));
                    u[i] ^= p1 ^ p2 ^ p3;
                }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave's own MFMAs and its own (independent) split chains: clustered (24 MFMAs, then 16 splits) or interleaved
#define SPLIT_ONE(i) { uint32_t p1, p2, p3; f32x2 hi, r, s_; \
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1) : "v"(p[i][0]), "v"(p[i][1])); \
    asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(hi[0]) : "v"(p1)); \
    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hi[1]) : "v"(p1)); \
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(p[i]), "v"(hi)); \
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p2) : "v"(r[0]), "v"(r[1])); \
    asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(hi[0]) : "v"(p2)); \
    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hi[1]) : "v"(p2)); \
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(s_) : "v"(r), "v"(hi)); \
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p3) : "v"(s_[0]), "v"(s_[1])); \
    u[i] ^= p1 ^ p2 ^ p3; }
template <int INTER>
__global__ __launch_bounds__(512) void k_own(float* out, int iters) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    uint4 uu = make_uint4(threadIdx.x * 2654435761u, 0x3f803f80u, 0x40004000u, threadIdx.x * 40503u);
    bf16x8 x = __builtin_bit_cast(bf16x8, uu), y = x;
    f32x2 p[8]; uint32_t u[8];
    for (int i = 0; i < 8; ++i) { p[i] = f32x2{(float)threadIdx.x + i, 1.5f * i}; u[i] = i; }
    for (int it = 0; it < iters; ++it) {
        if (INTER == 0) {
#pragma unroll
            for (int m = 0; m < 24; ++m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(x), "v"(y));
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) SPLIT_ONE(i)
        } else {
#pragma unroll
            for (int m = 0; m < 24; ++m) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(x), "v"(y));
                if (m % 3 != 2) SPLIT_ONE(m % 8)
            }
        }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> static float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4 * 8);
    const int iters = 4000;
    const char* names[] = {"v_cvt_pk_bf16_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_and_b32", "v_lshlrev_b32", "v_fma_f32", "v_mov_b32", "v_add_f32", "split2 (11 instr)"};
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock attribute %d kHz; 256 blocks; per op: ns per wave-instruction at 1 and 2 waves per SIMD (x GHz = cycles)\n", clk);
#define RUN(OP) { for (int th = 256; th <= 512; th += 256) { float ms = timeit([&] { hipLaunchKernelGGL(k_valu<OP>, dim3(256), dim3(th), 0, 0, out, iters); }); \
        const double n = (double)iters * 64 * (OP == 8 ? 11 : 1); printf("%-20s %d waves/SIMD: %.3f ms, %.2f ns per instruction per wave, %.2f ns per instruction per SIMD\n", names[OP], th / 256, ms, ms * 1e6 / n, ms * 1e6 / n / (th / 256)); } }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    const char* mn[] = {"split only (waves 4-7)", "MFMA only (waves 0-3)", "MFMA (0-3) beside split (4-7)", "MFMA on all 8 waves"};
#define MIX(M) { float ms = timeit([&] { hipLaunchKernelGGL(k_mix<M>, dim3(256), dim3(512), 0, 0, out, iters); }); printf("%-32s %.3f ms\n", mn[M], ms); }
    MIX(0) MIX(1) MIX(2) MIX(3)
    { float ms = timeit([&] { hipLaunchKernelGGL((k_mix<2, 1>), dim3(256), dim3(512), 0, 0, out, iters); }); printf("MFMA beside split, split waves at s_setprio 3: %.3f ms\n", ms); }
    { float ms = timeit([&] { hipLaunchKernelGGL((k_mix<2, 2>), dim3(256), dim3(512), 0, 0, out, iters); }); printf("MFMA beside split, MFMA waves at s_setprio 3: %.3f ms\n", ms); }
    { float ms = timeit([&] { hipLaunchKernelGGL(k_own<0>, dim3(256), dim3(256), 0, 0, out, iters); }); printf("one wave per SIMD: 24 MFMAs, then 176 VALU (clustered): %.3f ms\n", ms); }
    { float ms = timeit([&] { hipLaunchKernelGGL(k_own<1>, dim3(256), dim3(256), 0, 0, out, iters); }); printf("one wave per SIMD: 24 x (1 MFMA + 7-8 VALU) interleaved: %.3f ms\n", ms); }
    { float ms = timeit([&] { hipLaunchKernelGGL(k_own<0>, dim3(256), dim3(512), 0, 0, out, iters); }); printf("two waves per SIMD, each clustered: %.3f ms\n", ms); }
    { float ms = timeit([&] { hipLaunchKernelGGL(k_own<1>, dim3(256), dim3(512), 0, 0, out, iters); }); printf("two waves per SIMD, each interleaved: %.3f ms\n", ms); }
    return 0;
}
