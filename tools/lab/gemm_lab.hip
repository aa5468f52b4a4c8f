// gemm_lab: standalone fp32 MFMA GEMM experiments (C[M][N] = A[M][K] * B[N][K]^T, both K-contiguous).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_lab.hip -o gemm_lab ; run: ./gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// VAR bits: 1 = skip global loads (ablation), 2 = skip LDS stores, 4 = skip MFMA, 8 = 16B global loads
template <int WM, int WN, int TM, int TN, int VAR, int NBUF>
__global__ __launch_bounds__(WM * WN * 64) void sgemm_nt(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, int M, int N, int K) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 32, LD = BK + 4;
    constexpr int QA = BM * 8 / NT, QB = BN * 8 / NT;      // quads per thread
    __shared__ __attribute__((aligned(16))) float As[NBUF][BM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[NBUF][BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, (short)0, M * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)B, (short)0, N * K * 4, 0x00020000);
    f32x4 ra[QA], rb[QB];
    auto load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < QA; ++i) {
            const int q = tid + NT * i, row = q >> 3, kq = q & 7;
            const unsigned off = ((unsigned)(m0 + row) * K + kt + 4 * kq) * 4u;
            if (VAR & 8) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, off, 0, 0));
            else { for (int j = 0; j < 4; ++j) ra[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, off + 4 * j, 0, 0)); }
        }
#pragma unroll
        for (int i = 0; i < QB; ++i) {
            const int q = tid + NT * i, row = q >> 3, kq = q & 7;
            const unsigned off = ((unsigned)(n0 + row) * K + kt + 4 * kq) * 4u;
            if (VAR & 8) rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, off, 0, 0));
            else { for (int j = 0; j < 4; ++j) rb[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB, off + 4 * j, 0, 0)); }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < QA; ++i) { const int q = tid + NT * i; *(f32x4*)&As[buf][(q >> 3) * LD + 4 * (q & 7)] = ra[i]; }
#pragma unroll
        for (int i = 0; i < QB; ++i) { const int q = tid + NT * i; *(f32x4*)&Bs[buf][(q >> 3) * LD + 4 * (q & 7)] = rb[i]; }
    };
    f32x16 acc[TM][TN];
    for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int arow = (wm * TM * 32 + (lane & 31)) * LD + (lane >> 5) * 16;
    const int brow = (wn * TN * 32 + (lane & 31)) * LD + (lane >> 5) * 16;
    for (int i = 0; i < QA; ++i) ra[i] = 0; for (int i = 0; i < QB; ++i) rb[i] = 0;
    load(0); store(0); __syncthreads();
    const int nt = K / BK;
    for (int t = 0; t < nt; ++t) {
        const int cur = NBUF == 2 ? (t & 1) : 0;
        if (!(VAR & 1)) load((t + 1) * BK < K ? (t + 1) * BK : 0);
        f32x4 af[TM][4], bf[TN][4];
#pragma unroll
        for (int q = 0; q < TM; ++q) for (int v = 0; v < 4; ++v) af[q][v] = *(const f32x4*)&As[cur][arow + q * 32 * LD + 4 * v];
#pragma unroll
        for (int q = 0; q < TN; ++q) for (int v = 0; v < 4; ++v) bf[q][v] = *(const f32x4*)&Bs[cur][brow + q * 32 * LD + 4 * v];
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int tb = 0; tb < TN; ++tb) {
                    if (!(VAR & 4)) acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ta][s >> 2][s & 3], bf[tb][s >> 2][s & 3], acc[ta][tb], 0, 0, 0);
                    else acc[ta][tb][s] += af[ta][s >> 2][s & 3] * bf[tb][s >> 2][s & 3];
                }
        if (NBUF == 1) __syncthreads();
        if (!(VAR & 2)) store(NBUF == 2 ? (cur ^ 1) : 0);
        __syncthreads();
    }
    for (int tb = 0; tb < TN; ++tb) for (int ta = 0; ta < TM; ++ta) for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int n = n0 + wn * TN * 32 + tb * 32 + (lane & 31);
        C[(size_t)m * N + n] = acc[ta][tb][r];
    }
}

template <typename F> static float timeit(F f, int n = 8) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); for (int i = 0; i < n; ++i) f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms / n;
}
#define RUN(name, WM, WN, TM, TN, VAR, NBUF) { \
    dim3 g(N / (WN * TN * 32), M / (WM * TM * 32)); \
    float ms = timeit([&] { hipLaunchKernelGGL((sgemm_nt<WM, WN, TM, TN, VAR, NBUF>), g, dim3(WM * WN * 64), 0, 0, dA, dB, dC, M, N, K); }); \
    CHECK(hipMemcpy(hC.data(), dC, 64 * sizeof(float), hipMemcpyDeviceToHost)); \
    double ref = 0; for (int k = 0; k < K; ++k) ref += (double)hA[k] * hB[(size_t)5 * K + k]; \
    printf("%-34s %7.3f ms %7.1f TF   C[0][5]=%.4f ref=%.4f\n", name, ms, 2.0 * M * N * K / ms / 1e9, hC[5], ref); }
int main() {
    const int M = 4096, N = 4096, K = 4096;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hC(64);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2 - 1; for (auto& v : hB) v = (rand() / (float)RAND_MAX) * 2 - 1;
    float *dA, *dB, *dC; CHECK(hipMalloc(&dA, hA.size() * 4)); CHECK(hipMalloc(&dB, hB.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    RUN("128x128 4w 2x2 dword 2buf", 2, 2, 2, 2, 0, 2)
    RUN("128x128 4w 2x2 b128  2buf", 2, 2, 2, 2, 8, 2)
    RUN("128x128 4w 2x2 b128  1buf", 2, 2, 2, 2, 8, 1)
    RUN("128x128 noload", 2, 2, 2, 2, 9, 2)
    RUN("128x128 noload nostore", 2, 2, 2, 2, 11, 2)
    RUN("128x128 nomfma b128", 2, 2, 2, 2, 12, 2)
    RUN("256x128 8w 2x2 b128 2buf", 4, 2, 2, 2, 8, 2)
    RUN("256x128 4w 4x2 b128 2buf", 2, 2, 4, 2, 8, 2)
    RUN("256x256 8w 4x2 b128 1buf", 2, 4, 4, 2, 8, 1)
    RUN("128x256 4w 2x4 b128 2buf", 2, 2, 2, 4, 8, 2)
    RUN("128x64 4w 2x1 b128 2buf", 2, 2, 2, 1, 8, 2)
    RUN("64x64 4w 1x1 b128 2buf", 2, 2, 1, 1, 8, 2)
    return 0;
}
