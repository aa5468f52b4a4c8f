// mogan_lstm.hip -- the text encoder's recurrent part (eval, no gradient) as ONE launch.
//
// RNN_ENCODER (code/coco/attngan/model.py:120-204) embeds the captions (nn.Embedding(n_words, 300)) and runs a one-layer bidirectional
// LSTM (128 units per direction) over the packed sequences; words_emb = the outputs (B, 256, T_max), zero behind every caption's
// end, sent_emb = the two final hidden states (B, 256).  It runs once per train step, frozen, without gradients: 0.1 % of the step's
// FLOPs -- and, on the stock nn.LSTM (MIOpen), 119 launches, 0.37 ms of kernel time and 0.8 ms of host time per step (B = 16, T = 12:
// one GEMM + one update kernel per time step and direction, pack / unpack copies; tools/time_text.py).  Here a block owns one
// (direction, caption):
//   * 512 threads = the 512 gate rows (i, f, g, o x 128 units).  The embedded caption sits in LDS; thread j streams row j of W_ih ONCE
//     and forms its input projection for every time step in registers (bias b_ih + b_hh folded in), parks it in LDS [t][512],
//   * then holds row j of W_hh (128 values) in registers for the recurrence: per step 128 fmas against the hidden state in LDS
//     (broadcast reads), the pre-activations through LDS to the 128 unit threads, which apply the gates (PyTorch's order i, f, g, o:
//     c' = s(f) c + s(i) tanh(g), h' = s(o) tanh(c')), write h into the output row of step t and back to LDS; two barriers per step.
// The backward direction walks t = len - 1 ... 0; positions t >= len of words_emb are written as zeros (pad_packed_sequence).
// fp32 throughout; the summation order differs from MIOpen's GEMMs (results agree to ~1e-6 relative, tests/test_kernels_gpu.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mogan_hip.h"
#include "mogan_internal.h"

namespace {

constexpr int LS_H = 128, LS_G = 4 * LS_H, LS_TMAX = 32, LS_EMAX = 320, LS_BMAX = 64;

struct LstmP {
    const long long* cap; const float* emb;
    const float* w_ih[2]; const float* w_hh[2]; const float* b_ih[2]; const float* b_hh[2];
    const float* h0; const float* c0;
    float* words; float* sent;
    int B, T, Tmax, V, E;
    int lens[LS_BMAX];
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(LS_G) void lstm_encoder_kernel(const LstmP p) {
    __shared__ __attribute__((aligned(16))) float Xs[LS_TMAX * LS_EMAX];      // the embedded caption [t][E]; later: the projections
    __shared__ __attribute__((aligned(16))) float XP[LS_TMAX * LS_G];         // input projections [t][gate row]
    __shared__ __attribute__((aligned(16))) float Hs[LS_H];
    __shared__ float Gs[LS_G];
    const int j = threadIdx.x, dir = blockIdx.x & 1, b = blockIdx.x >> 1;
    const int E = p.E, len = min(max(p.lens[b], 0), p.Tmax);
    // ---- the caption's embedding rows
    for (int t = 0; t < len; ++t) {
        long long tok = p.cap[(size_t)b * p.T + t];
        tok = tok < 0 ? 0 : (tok >= p.V ? p.V - 1 : tok);
        const float* row = p.emb + (size_t)tok * E;
        if (j < E) Xs[t * LS_EMAX + j] = row[j];
    }
    __syncthreads();
    // ---- input projection of every step: one pass over row j of W_ih
    {
        float acc[LS_TMAX];
        const float bias = p.b_ih[dir][j] + p.b_hh[dir][j];
#pragma unroll
        for (int t = 0; t < LS_TMAX; ++t) acc[t] = bias;
        const float4* wr = (const float4*)(p.w_ih[dir] + (size_t)j * E);
        for (int k4 = 0; k4 < E / 4; ++k4) {
            const float4 w = wr[k4];
#pragma unroll
            for (int t = 0; t < LS_TMAX; ++t)
                if (t < len) {                                   // (the same for the whole block)
                    const float4 x = *(const float4*)&Xs[t * LS_EMAX + 4 * k4];
                    acc[t] = fmaf(w.x, x.x, acc[t]); acc[t] = fmaf(w.y, x.y, acc[t]);
                    acc[t] = fmaf(w.z, x.z, acc[t]); acc[t] = fmaf(w.w, x.w, acc[t]);
                }
        }
#pragma unroll
        for (int t = 0; t < LS_TMAX; ++t) if (t < len) XP[t * LS_G + j] = acc[t];
    }
    // ---- recurrence
    float4 whh[LS_H / 4];
    {
        const float4* hr = (const float4*)(p.w_hh[dir] + (size_t)j * LS_H);
#pragma unroll
        for (int k4 = 0; k4 < LS_H / 4; ++k4) whh[k4] = hr[k4];
    }
    float c = 0.f, h = 0.f;
    if (j < LS_H) {
        const size_t s0 = ((size_t)dir * p.B + b) * LS_H + j;
        if (p.h0) h = p.h0[s0];
        if (p.c0) c = p.c0[s0];
        Hs[j] = h;
    }
    __syncthreads();
    float* wout = p.words + ((size_t)b * 2 * LS_H + (size_t)dir * LS_H) * p.Tmax;
    for (int s = 0; s < len; ++s) {
        const int t = dir ? len - 1 - s : s;
        float a0 = XP[t * LS_G + j], a1 = 0.f, a2 = 0.f, a3 = 0.f;       // four chains, summed at the end
#pragma unroll
        for (int k4 = 0; k4 < LS_H / 4; ++k4) {
            const float4 hv = *(const float4*)&Hs[4 * k4];
            a0 = fmaf(whh[k4].x, hv.x, a0); a1 = fmaf(whh[k4].y, hv.y, a1);
            a2 = fmaf(whh[k4].z, hv.z, a2); a3 = fmaf(whh[k4].w, hv.w, a3);
        }
        Gs[j] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (j < LS_H) {
            const float gi = sigm(Gs[j]), gf = sigm(Gs[LS_H + j]), gg = tanhf(Gs[2 * LS_H + j]), go = sigm(Gs[3 * LS_H + j]);
            c = gf * c + gi * gg;
            h = go * tanhf(c);
            Hs[j] = h;
            wout[(size_t)j * p.Tmax + t] = h;
        }
        __syncthreads();
    }
    if (j < LS_H) {
        p.sent[(size_t)b * 2 * LS_H + dir * LS_H + j] = h;
        for (int t = len; t < p.Tmax; ++t) wout[(size_t)j * p.Tmax + t] = 0.f;
    }
}

}  // namespace

extern "C" {

int mogan_lstm_encoder_fwd(const long long* captions, const int* lens_host, const float* emb, const float* const* w_ih,
                           const float* const* w_hh, const float* const* b_ih, const float* const* b_hh, const float* h0,
                           const float* c0, float* words, float* sent, int B, int T, int Tmax, int V, int E, int H,
                           hipStream_t stream) {
    if (!captions || !lens_host || !emb || !w_ih || !w_hh || !b_ih || !b_hh || !words || !sent) return MOGAN_ERR_SHAPE;
    if (B <= 0 || B > LS_BMAX || H != LS_H || T <= 0 || Tmax <= 0 || Tmax > T || Tmax > LS_TMAX || E <= 0 || E > LS_EMAX || (E % 4) || V <= 0)
        return MOGAN_ERR_SHAPE;
    LstmP p{};
    p.cap = captions; p.emb = emb; p.h0 = h0; p.c0 = c0; p.words = words; p.sent = sent;
    p.B = B; p.T = T; p.Tmax = Tmax; p.V = V; p.E = E;
    for (int d = 0; d < 2; ++d) {
        if (!w_ih[d] || !w_hh[d] || !b_ih[d] || !b_hh[d] || (((uintptr_t)w_ih[d] | (uintptr_t)w_hh[d]) & 15)) return MOGAN_ERR_SHAPE;
        p.w_ih[d] = w_ih[d]; p.w_hh[d] = w_hh[d]; p.b_ih[d] = b_ih[d]; p.b_hh[d] = b_hh[d];
    }
    for (int i = 0; i < B; ++i) { if (lens_host[i] < 0 || lens_host[i] > Tmax) return MOGAN_ERR_SHAPE; p.lens[i] = lens_host[i]; }
    hipLaunchKernelGGL(lstm_encoder_kernel, dim3(2 * B), dim3(LS_G), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH;
}

}  // extern "C"
