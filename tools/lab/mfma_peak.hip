// Lab: what the bf16 matrix pipe of this MI355X sustains (no memory traffic), and what the "9 x ds_read_b128 + 18 MFMA" group
// of dconv_fwd_kernel<.., TM=3, TN=1> sustains with its operands coming from LDS.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_reg(float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    uint4 u = make_uint4(threadIdx.x, 2, 3, 4);
    bf16x8 x = __builtin_bit_cast(bf16x8, u), y = x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 6; ++rep)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// operands from LDS: per group 3 tiles x 3 pieces of A (b128, row stride 496 B) + 3 pieces of B (b128), 18 MFMAs
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_lds(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char L[96 * 496 + 4096];
    for (int i = threadIdx.x; i < (96 * 496 + 4096) / 4; i += blockDim.x) ((uint32_t*)L)[i] = i * 2654435761u >> 20;
    __syncthreads();
    f32x16 acc[3];
    for (int a = 0; a < 3; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int lane = threadIdx.x & 63;
    const unsigned char* A = L + (lane & 31) * 496 + (lane >> 5) * 80;
    const unsigned char* B = L + 96 * 496 + lane * 16;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            bf16x8 fa[3][3], fb[3];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) fa[t][p] = __builtin_bit_cast(bf16x8, *(const uint4*)(A + t * 32 * 496 + (p * 10 + g) * 16));
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[p] = __builtin_bit_cast(bf16x8, *(const uint4*)(B + ((p + g) & 3) * 1024));
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t][term % 3], fb[term / 2], acc[t], 0, 0, 0);
        }
    }
    float s = 0;
    for (int a = 0; a < 3; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same with addresses the compiler cannot prove loop-invariant (offset advanced by a runtime step): the reads stay in the
// loop; PIPE = 1: fragments of group g+1 are requested before the MFMAs of group g (register double buffering)
template <int WAVES, int PIPE, int NB32, int ORDER = 9>
__global__ __launch_bounds__(64 * WAVES) void k_lds2(float* out, int iters, int step, int rnd) {
    __shared__ __attribute__((aligned(16))) unsigned char L[96 * 496 + 8192];
    for (int i = threadIdx.x; i < (96 * 496 + 8192) / 4; i += blockDim.x) {
        uint32_t hsh = (i + 1 + blockIdx.x * 7919) * 2654435761u; hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
        // rnd: two bf16 values with random sign / mantissa and exponent 125..127 (|x| in [0.25, 2)); else tiny integer patterns
        const uint32_t lo = (hsh & 0x807Fu) | ((125u + ((hsh >> 8) % 3u)) << 7), hi = ((hsh >> 16) & 0x807Fu) | ((125u + ((hsh >> 28) % 3u)) << 7);
        ((uint32_t*)L)[i] = rnd ? (lo | (hi << 16)) : (i * 2654435761u >> 20);
    }
    __syncthreads();
    f32x16 acc[3];
    for (int a = 0; a < 3; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int lane = threadIdx.x & 63;
    const unsigned char* A = L + (lane & 31) * 496 + (lane >> 5) * 80;
    const unsigned char* B = L + 96 * 496 + lane * 16;
    int off = 0;
    bf16x8 fa[2][3][3], fb[2][3];
    auto load = [&](int buf, int o) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[buf][t][p] = __builtin_bit_cast(bf16x8, *(const uint4*)(A + t * 32 * 496 + p * 160 + o));
        if (NB32 == 0) {
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[buf][p] = __builtin_bit_cast(bf16x8, *(const uint4*)(B + p * 1024 + o));
        } else {      // the present kernel's B path: 8 dword reads (then split; here just packed)
            uint32_t w[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = *(const uint32_t*)(B + q * 260 + o);
            fb[buf][0] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
            fb[buf][1] = __builtin_bit_cast(bf16x8, make_uint4(w[4], w[5], w[6], w[7]));
            fb[buf][2] = __builtin_bit_cast(bf16x8, make_uint4(w[0] ^ w[4], w[1] ^ w[5], w[2] ^ w[6], w[3] ^ w[7]));
        }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                // piece indices (A, B) of the six partial products: ORDER 0 = mogan_mma.h today (3,1)(2,2)(1,3)(2,1)(1,2)(1,1);
                // 1 = A piece kept as long as possible (3,1)(2,2)(2,1)(1,3)(1,2)(1,1); 2 = B piece kept (1,3)(2,2)(1,2)(3,1)(2,1)(1,1)
                constexpr int IA0[6] = {2, 1, 0, 1, 0, 0}, IB0[6] = {0, 1, 2, 0, 1, 0};
                constexpr int IA1[6] = {2, 1, 1, 0, 0, 0}, IB1[6] = {0, 1, 0, 2, 1, 0};
                constexpr int IA2[6] = {0, 1, 0, 2, 1, 0}, IB2[6] = {2, 1, 1, 0, 0, 0};
                const int ia = ORDER == 0 ? IA0[term] : ORDER == 1 ? IA1[term] : ORDER == 2 ? IA2[term] : term % 3;
                const int ib = ORDER == 0 ? IB0[term] : ORDER == 1 ? IB1[term] : ORDER == 2 ? IB2[term] : term / 2;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][t][ia], fb[buf][ib], acc[t], 0, 0, 0);
            }
    };
    if (PIPE) load(0, 0);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            off = (off + step) & 63;
            if (PIPE) { load((g + 1) & 1, off & ~15); mma(g & 1); }
            else { load(0, off & ~15); mma(0); }
        }
    }
    float s = 0;
    for (int a = 0; a < 3; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// int8 pipe: v_mfma_i32_32x32x32_i8 (a lane supplies 16 int8 per operand, K = 32 per instruction) with operands from LDS as in k_lds2
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_i8(float* out, int iters, int step, int rnd) {
    __shared__ __attribute__((aligned(16))) unsigned char L[96 * 496 + 8192];
    for (int i = threadIdx.x; i < (96 * 496 + 8192) / 4; i += blockDim.x) {
        uint32_t hsh = (i + 1 + blockIdx.x * 7919) * 2654435761u; hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
        ((uint32_t*)L)[i] = rnd ? hsh : (i & 3);
    }
    __syncthreads();
    i32x16 acc[3];
    for (int a = 0; a < 3; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0;
    const int lane = threadIdx.x & 63;
    const unsigned char* A = L + (lane & 31) * 496 + (lane >> 5) * 80;
    const unsigned char* B = L + 96 * 496 + lane * 16;
    int off = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            off = (off + step) & 63;
            const int o = off & ~15;
            i32x4v fa[3][3], fb[3];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p) fa[t][p] = *(const i32x4v*)(A + t * 32 * 496 + p * 160 + o);
#pragma unroll
            for (int p = 0; p < 3; ++p) fb[p] = *(const i32x4v*)(B + p * 1024 + o);
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[t][term % 3], fb[term / 2], acc[t], 0, 0, 0);
        }
    }
    int s = 0;
    for (int a = 0; a < 3; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}

template <typename F>
static double run(F launch, double flops_per_launch, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s bf16  = %6.1f fp32-equivalent (/6)\n", name, ms, flops_per_launch / ms / 1e9, flops_per_launch / ms / 1e9 / 6);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 16 * 1024 * sizeof(float));
    const int iters = 2000;
    const double mf = 2.0 * 32 * 32 * 16;
    for (int blocks : {256, 512, 1024, 2048}) {
        char n[128];
        snprintf(n, sizeof n, "regs: 4 acc, 4 waves/block, %d blocks", blocks);
        run([&] { hipLaunchKernelGGL((k_reg<4, 4>), dim3(blocks), dim3(256), 0, 0, out, iters); }, mf * 6 * 4 * iters * 4.0 * blocks, n);
    }
    run([&] { hipLaunchKernelGGL((k_reg<3, 4>), dim3(512), dim3(256), 0, 0, out, iters); }, mf * 6 * 3 * iters * 4.0 * 512, "regs: 3 acc, 512 blocks");
    run([&] { hipLaunchKernelGGL((k_reg<2, 4>), dim3(512), dim3(256), 0, 0, out, iters); }, mf * 6 * 2 * iters * 4.0 * 512, "regs: 2 acc, 512 blocks");
    run([&] { hipLaunchKernelGGL((k_reg<1, 4>), dim3(512), dim3(256), 0, 0, out, iters); }, mf * 6 * 1 * iters * 4.0 * 512, "regs: 1 acc (dependent chain), 512 blocks");
    for (int blocks : {256, 512, 1024}) {
        char n[128];
        snprintf(n, sizeof n, "LDS operands (9+3 b128 / 18 MFMA), %d blocks", blocks);
        run([&] { hipLaunchKernelGGL((k_lds<4>), dim3(blocks), dim3(256), 0, 0, out, iters / 5); }, mf * 18 * 5 * (iters / 5) * 4.0 * blocks, n);
    }
    for (int blocks : {512, 1024}) {
        char n[128];
        snprintf(n, sizeof n, "LDS in loop, no pipelining, %d blocks", blocks);
        run([&] { hipLaunchKernelGGL((k_lds2<4, 0, 0>), dim3(blocks), dim3(256), 0, 0, out, iters / 4, 16, 0); }, mf * 18 * 4 * (iters / 4) * 4.0 * blocks, n);
        snprintf(n, sizeof n, "LDS in loop, register double buffer, %d blocks", blocks);
        run([&] { hipLaunchKernelGGL((k_lds2<4, 1, 0>), dim3(blocks), dim3(256), 0, 0, out, iters / 4, 16, 0); }, mf * 18 * 4 * (iters / 4) * 4.0 * blocks, n);
        snprintf(n, sizeof n, "LDS in loop, B as 8 x b32, dbl buffer, %d blocks", blocks);
        run([&] { hipLaunchKernelGGL((k_lds2<4, 1, 1>), dim3(blocks), dim3(256), 0, 0, out, iters / 4, 16, 0); }, mf * 18 * 4 * (iters / 4) * 4.0 * blocks, n);
    }
    for (int blocks : {512, 1024}) {
        char n[128];
        snprintf(n, sizeof n, "RANDOM bf16 data: LDS in loop, dbl buffer, %d blocks", blocks);
        run([&] { hipLaunchKernelGGL((k_lds2<4, 1, 0>), dim3(blocks), dim3(256), 0, 0, out, iters / 4, 16, 1); }, mf * 18 * 4 * (iters / 4) * 4.0 * blocks, n);
    }
    for (int rep = 0; rep < 2; ++rep)
        run([&] { hipLaunchKernelGGL((k_lds2<4, 1, 0>), dim3(2048), dim3(256), 0, 0, out, iters, 16, 1); }, mf * 18 * 4 * (double)iters * 4.0 * 2048, "RANDOM data, long run (2048 blocks x 4x iterations)");
    // order of the six partial products (random data; term-major over the three accumulator tiles as in the kernels)
    run([&] { hipLaunchKernelGGL((k_lds2<4, 1, 0, 0>), dim3(1024), dim3(256), 0, 0, out, iters, 16, 1); }, mf * 18 * 4 * (double)iters * 4.0 * 1024, "RANDOM, x6 order of mogan_mma.h");
    run([&] { hipLaunchKernelGGL((k_lds2<4, 1, 0, 1>), dim3(1024), dim3(256), 0, 0, out, iters, 16, 1); }, mf * 18 * 4 * (double)iters * 4.0 * 1024, "RANDOM, A piece kept across products");
    run([&] { hipLaunchKernelGGL((k_lds2<4, 1, 0, 2>), dim3(1024), dim3(256), 0, 0, out, iters, 16, 1); }, mf * 18 * 4 * (double)iters * 4.0 * 1024, "RANDOM, B piece kept across products");
    run([&] { hipLaunchKernelGGL((k_lds2<4, 1, 0, 0>), dim3(1024), dim3(256), 0, 0, out, iters, 16, 1); }, mf * 18 * 4 * (double)iters * 4.0 * 1024, "RANDOM, x6 order of mogan_mma.h (again)");
    // the int8 pipe (K = 32 per instruction: twice the multiply-adds of the bf16 instruction); "TFLOP/s bf16" column = TOP/s here
    const double mi = 2.0 * 32 * 32 * 32;
    run([&] { hipLaunchKernelGGL((k_i8<4>), dim3(1024), dim3(256), 0, 0, out, iters, 16, 0); }, mi * 18 * 4 * (double)iters * 4.0 * 1024, "int8 MFMA 32x32x32, constant data (TOP/s)");
    run([&] { hipLaunchKernelGGL((k_i8<4>), dim3(1024), dim3(256), 0, 0, out, iters, 16, 1); }, mi * 18 * 4 * (double)iters * 4.0 * 1024, "int8 MFMA 32x32x32, RANDOM data (TOP/s)");
    run([&] { hipLaunchKernelGGL((k_i8<4>), dim3(1024), dim3(256), 0, 0, out, iters, 16, 1); }, mi * 18 * 4 * (double)iters * 4.0 * 1024, "int8 MFMA 32x32x32, RANDOM data (again)");
    return 0;
}
