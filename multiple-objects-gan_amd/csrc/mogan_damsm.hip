// mogan_damsm.hip -- the DAMSM word / sentence matching losses of the generator step as fused kernels.
//
// Reference (stock torch ops composed in python, one func_attention call per caption in a B-iteration loop):
//   code/coco/attngan/miscc/losses.py:62-132  words_loss       (+ GlobalAttention.py:31-69 func_attention)
//   code/coco/attngan/miscc/losses.py:20-59   sent_loss
//   code/coco/attngan/miscc/losses.py:11-17   cosine_similarity
//
// words, forward: one workgroup per (image b, caption i) pair -- the B x B pairs are independent until the two
// cross-entropies.  With ctx = region features of image b (C x S, S = 17*17) and w = word embeddings of caption i
// (C x T_i, only the T_i = cap_lens[i] valid words):
//     score[s,t] = sum_c ctx[c,s] w[c,t]                                  (GlobalAttention.py:46)
//     a1[s,:]    = softmax_t(score[s,:])                                  (:50-52)
//     a2[t,:]    = softmax_s(gamma1 * a1[:,t])                            (:57-60)
//     wc[c,t]    = sum_s ctx[c,s] a2[t,s]                                 (:66)
//     cos[t]     = <wc[:,t], w[:,t]> / max(|wc[:,t]| |w[:,t]|, 1e-8)      (losses.py:11-17,95)
//     sim[b,i]   = gamma3 * log sum_t exp(gamma2 cos[t])                  (losses.py:99-104,115)
// everything between the region features and sim[b,i] stays in registers / LDS; a1, a2 and wc are written once for
// the backward pass (a2 doubles as the attention maps of losses.py:87-91).  HBM-bound: each workgroup streams the
// 296 KB of its image's features twice (L2-resident: the 16 captions of an image run side by side).
// words, backward: same decomposition; a workgroup turns d sim[b,i] into d wc (C x T) and d score (T x S); the two
// rank-(B*T) updates of d ctx are then plain strided GEMMs (mogan_bmm) over the caption axis -- no atomics.
// Cross-entropies (rows: image -> caption, columns: caption -> image, optional same-class mask, losses.py:116-130)
// and the sentence loss (cosine matrix of the two code vectors, losses.py:36-58) are small per-row kernels.
// Gradients are produced for the IMAGE side only (region features / cnn code): in the generator step the text encoder
// is frozen and its embeddings are detached (trainer.py:281-289).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mogan_hip.h"

namespace {

constexpr int TMAX = 32;      // longest caption (cfg.TEXT.WORDS_NUM: 12 for coco, 18 for birds)
constexpr int NT2 = 256;      // the small per-row kernels (4 wave64)
constexpr int NT = 320;       // threads per workgroup (5 wave64: the 289 regions of a 17x17 map in one pass)

static inline int ok_launch() { return hipGetLastError() == hipSuccess ? 0 : MOGAN_ERR_LAUNCH; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// dynamic LDS layout (floats): ws[C*T] word embeddings (or d wc in the backward), a[T*SP] attention image, red[...]
struct Lds {
    float* w; float* a; float* red;
    __device__ Lds(float* base, int C, int T, int SP) : w(base), a(base + C * T), red(base + C * T + T * SP) {}
};

// ---------------------------------------------------------------------------------------------- words: forward
template <int TT>
__global__ __launch_bounds__(NT) void damsm_words_fwd_kernel(
    const float* __restrict__ ctx, const float* __restrict__ words, const int32_t* __restrict__ lens, int Bc, int C,
    int S, int T, float gamma1, float gamma2, float gamma3, float* __restrict__ sim, float* __restrict__ a1o,
    float* __restrict__ a2o, float* __restrict__ wco, float* __restrict__ wto) {
    extern __shared__ float smem[];
    const int SP = S + 1;
    Lds L(smem, C, T, SP);
    const int i = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ti = min(T, max(0, lens[i]));
    // The multiply-add loops run over ALL TT word slots without the `t < Ti` test (round 4): a uniform branch per slot cut the
    // unrolled loops into hundreds of two-instruction blocks with nothing in flight between them (300 us for 0.2 GFLOP).  Slots
    // past the caption's length multiply values nobody reads; slots past T re-read slot T - 1 (tcl) so that every address is valid.
    int tcl[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) tcl[t] = min(t, T - 1);
    const float* cb = ctx + (size_t)b * C * S;
    const float* wi = words + (size_t)i * C * T;
    for (int e = tid; e < C * T; e += NT) {
        const float v = wi[e];
        L.w[e] = v;                                              // [c][t]
        if (b == 0 && wto) { const int c = e / T, t = e - c * T; wto[((size_t)c * Bc + i) * T + t] = t < Ti ? v : 0.f; }
    }
    __syncthreads();
    // scores + softmax over the words, one thread per region s; the channel loop runs 8 loads ahead of its FMAs
    for (int s = tid; s < S; s += NT) {
        float acc[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) acc[t] = 0.f;
        int c = 0;
        // (round 4: the loads of the NEXT eight channels are in flight while the current eight are multiplied -- with five waves
        // per CU nothing else hides the L2 round trip, and the loop was one exposed latency per eight channels)
        float xn[8];
        if (C >= 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) xn[u] = cb[(size_t)u * S + s];
        }
        for (; c + 8 <= C; c += 8) {
            float xs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xs[u] = xn[u];
            if (c + 16 <= C) {
#pragma unroll
                for (int u = 0; u < 8; ++u) xn[u] = cb[(size_t)(c + 8 + u) * S + s];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* wr = L.w + (c + u) * T;
#pragma unroll
                for (int t = 0; t < TT; ++t) acc[t] = fmaf(xs[u], wr[tcl[t]], acc[t]);
            }
        }
        for (; c < C; ++c) {
            const float x = cb[(size_t)c * S + s];
            const float* wr = L.w + c * T;
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[t] = fmaf(x, wr[tcl[t]], acc[t]);
        }
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < TT; ++t) if (t < Ti) m = fmaxf(m, acc[t]);
        float z = 0.f;
#pragma unroll
        for (int t = 0; t < TT; ++t) if (t < Ti) { acc[t] = __expf(acc[t] - m); z += acc[t]; }
        const float inv = 1.f / z;
        float* o1 = a1o + (((size_t)b * Bc + i) * S + s) * T;
#pragma unroll
        for (int t = 0; t < TT; ++t) if (t < T) {
            const float v = t < Ti ? acc[t] * inv : 0.f;
            o1[t] = v;
            L.a[t * SP + s] = v;
        }
    }
    __syncthreads();
    // softmax over the regions of gamma1 * a1[:, t]: wave w owns the words t = w, w + 4, ...
    for (int t = wave; t < T; t += NT / 64) {
        float* row = L.a + t * SP;
        float* o2 = a2o + (((size_t)b * Bc + i) * T + t) * S;
        if (t >= Ti) { for (int s = lane; s < S; s += 64) { row[s] = 0.f; o2[s] = 0.f; } continue; }
        float m = -INFINITY;
        for (int s = lane; s < S; s += 64) m = fmaxf(m, row[s] * gamma1);
        m = wave_max(m);
        float z = 0.f;
        for (int s = lane; s < S; s += 64) { const float e = __expf(row[s] * gamma1 - m); row[s] = e; z += e; }
        z = wave_sum(z);
        const float inv = 1.f / z;
        for (int s = lane; s < S; s += 64) { const float v = row[s] * inv; row[s] = v; o2[s] = v; }
    }
    __syncthreads();
    // weighted context wc[c, t] = sum_s ctx[c, s] a2[t, s]: wave w owns the channels c = w, w + 5, ...; lanes along s,
    // two channels per trip (their loads in flight together).  On the fly: <wc_t, w_t>, |wc_t|^2, |w_t|^2 (lane t of the
    // wave keeps the sums of word t).
    float dot = 0.f, nwc = 0.f, nw = 0.f;
    constexpr int NW = NT / 64;
    constexpr int SI = 5;                                    // s-iterations of a wave held in registers (S <= 320)
    for (int c0 = wave; c0 < C; c0 += 2 * NW) {
        const int c1 = c0 + NW;
        const bool two = c1 < C;
        float p[TT], q[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) { p[t] = 0.f; q[t] = 0.f; }
        if (S <= 64 * SI) {
            // (round 4) all loads of the channel pair first, then the multiplies: the loop used to wait for one L2 round trip per
            // 64 regions, 130 times per wave
            float xv[SI], yv[SI];
#pragma unroll
            for (int k = 0; k < SI; ++k) {
                const int s = lane + 64 * k;
                xv[k] = s < S ? cb[(size_t)c0 * S + s] : 0.f;
                yv[k] = (two && s < S) ? cb[(size_t)c1 * S + s] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < SI; ++k) {
                const int s = lane + 64 * k;
                if (s < S) {
#pragma unroll
                    for (int t = 0; t < TT; ++t) { const float a = L.a[tcl[t] * SP + s]; p[t] = fmaf(xv[k], a, p[t]); q[t] = fmaf(yv[k], a, q[t]); }
                }
            }
        } else
        for (int s = lane; s < S; s += 64) {
            const float x = cb[(size_t)c0 * S + s], y = two ? cb[(size_t)c1 * S + s] : 0.f;
#pragma unroll
            for (int t = 0; t < TT; ++t) { const float a = L.a[tcl[t] * SP + s]; p[t] = fmaf(x, a, p[t]); q[t] = fmaf(y, a, q[t]); }
        }
        float mine = 0.f, mine2 = 0.f;
#pragma unroll
        for (int t = 0; t < TT; ++t) {                       // (all TT slots, no `t < Ti` branch: see tcl)
            const float v = wave_sum(p[t]), v2 = wave_sum(q[t]);
            if (lane == t) { mine = v; mine2 = v2; }
        }
        if (lane < T) {
            const float v = lane < Ti ? mine : 0.f;
            wco[(((size_t)b * C + c0) * Bc + i) * T + lane] = v;        // [b][c][i][t]
            const float wv = L.w[c0 * T + lane];
            dot = fmaf(v, wv, dot); nwc = fmaf(v, v, nwc); nw = fmaf(wv, wv, nw);
            if (two) {
                const float v2 = lane < Ti ? mine2 : 0.f;
                wco[(((size_t)b * C + c1) * Bc + i) * T + lane] = v2;
                const float wv2 = L.w[c1 * T + lane];
                dot = fmaf(v2, wv2, dot); nwc = fmaf(v2, v2, nwc); nw = fmaf(wv2, wv2, nw);
            }
        }
    }
    if (lane < T) { L.red[(wave * 3 + 0) * TMAX + lane] = dot; L.red[(wave * 3 + 1) * TMAX + lane] = nwc;
                    L.red[(wave * 3 + 2) * TMAX + lane] = nw; }
    __syncthreads();
    if (wave == 0) {
        float e = 0.f;
        if (lane < Ti) {
            float d = 0.f, a = 0.f, w2 = 0.f;
            for (int q = 0; q < NT / 64; ++q) { d += L.red[(q * 3 + 0) * TMAX + lane]; a += L.red[(q * 3 + 1) * TMAX + lane];
                                                w2 += L.red[(q * 3 + 2) * TMAX + lane]; }
            const float cosv = d / fmaxf(sqrtf(a) * sqrtf(w2), 1e-8f);
            e = __expf(gamma2 * cosv);
        }
        e = wave_sum(e);
        if (lane == 0) sim[(size_t)b * Bc + i] = logf(e) * gamma3;
    }
}

// ---------------------------------------------------------------------------------------------- words: backward
// d sim[b,i] -> d wc[b][c][i][t] and d score^T[b][i][t][s]
template <int TT>
__global__ __launch_bounds__(NT) void damsm_words_bwd_kernel(
    const float* __restrict__ ctx, const float* __restrict__ words, const int32_t* __restrict__ lens,
    const float* __restrict__ a1, const float* __restrict__ a2, const float* __restrict__ wc,
    const float* __restrict__ dsim, int Bc, int C, int S, int T, float gamma1, float gamma2, float gamma3,
    float* __restrict__ dwc, float* __restrict__ dst) {
    extern __shared__ float smem[];
    const int SP = S + 1;
    Lds L(smem, C, T, SP);
    const int i = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ti = min(T, max(0, lens[i]));
    int tcl[TT];                                             // (see the forward kernel)
#pragma unroll
    for (int t = 0; t < TT; ++t) tcl[t] = min(t, T - 1);
    const float* cb = ctx + (size_t)b * C * S;
    const float* wi = words + (size_t)i * C * T;
    const float g = dsim[(size_t)b * Bc + i];
    // per word: <wc,w>, |wc|^2, |w|^2 again (thread (t = tid % TMAX, part = tid / TMAX) sums a stripe of c)
    {
        const int t = tid % TMAX, part = tid / TMAX, nparts = NT / TMAX;
        float d = 0.f, a = 0.f, w2 = 0.f;
        if (t < Ti)
            for (int c = part; c < C; c += nparts) {
                const float v = wc[(((size_t)b * C + c) * Bc + i) * T + t], wv = wi[c * T + t];
                d = fmaf(v, wv, d); a = fmaf(v, v, a); w2 = fmaf(wv, wv, w2);
            }
        L.red[(part * 3 + 0) * TMAX + t] = d; L.red[(part * 3 + 1) * TMAX + t] = a; L.red[(part * 3 + 2) * TMAX + t] = w2;
    }
    __syncthreads();
    float* coef = L.red + 3 * (NT / TMAX) * TMAX;            // [3][TMAX]: k1[t] (on w), k2[t] (on wc)
    if (tid < TMAX) {
        const int t = tid;
        float d = 0.f, a = 0.f, w2 = 0.f;
        for (int q = 0; q < NT / TMAX; ++q) { d += L.red[(q * 3 + 0) * TMAX + t]; a += L.red[(q * 3 + 1) * TMAX + t];
                                              w2 += L.red[(q * 3 + 2) * TMAX + t]; }
        const float den = sqrtf(a) * sqrtf(w2);
        const bool clamped = den < 1e-8f;
        const float cosv = d / fmaxf(den, 1e-8f);
        float e = t < Ti ? __expf(gamma2 * cosv) : 0.f;
        float z = e;
#pragma unroll
        for (int o = TMAX / 2; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);     // lanes 0..31 of wave 0
        const float dcos = t < Ti ? g * gamma3 * gamma2 * e / z : 0.f;
        // d cos / d wc = w / den - cos * wc / |wc|^2   (den clamped: only the first term, with 1e-8)
        coef[t] = dcos / fmaxf(den, 1e-8f);
        coef[TMAX + t] = (clamped || a <= 0.f) ? 0.f : -dcos * cosv / a;
    }
    __syncthreads();
    for (int e = tid; e < C * T; e += NT) {
        const int c = e / T, t = e - c * T;
        float v = 0.f;
        if (t < Ti) v = coef[t] * wi[e] + coef[TMAX + t] * wc[(((size_t)b * C + c) * Bc + i) * T + t];
        L.w[e] = v;
        dwc[(((size_t)b * C + c) * Bc + i) * T + t] = v;
    }
    __syncthreads();
    // d a2[t,s] = sum_c d wc[c,t] ctx[c,s], one thread per region (S <= 2 * NT: at most two regions per thread, kept in
    // registers); a2 * d a2 goes to LDS for the per-word sums over s
    float da[2][TT], av[2][TT];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int s = tid + j * NT;
#pragma unroll
        for (int t = 0; t < TT; ++t) { da[j][t] = 0.f; av[j][t] = 0.f; }
        if (s < S) {
            int c = 0;
            float xn[8];                                     // (the next eight channels' loads in flight, as in the forward)
            if (C >= 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) xn[u] = cb[(size_t)u * S + s];
            }
            for (; c + 8 <= C; c += 8) {
                float xs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) xs[u] = xn[u];
                if (c + 16 <= C) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) xn[u] = cb[(size_t)(c + 8 + u) * S + s];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float* wr = L.w + (c + u) * T;
#pragma unroll
                    for (int t = 0; t < TT; ++t) da[j][t] = fmaf(xs[u], wr[tcl[t]], da[j][t]);
                }
            }
            for (; c < C; ++c) {
                const float x = cb[(size_t)c * S + s];
                const float* wr = L.w + c * T;
#pragma unroll
                for (int t = 0; t < TT; ++t) da[j][t] = fmaf(x, wr[tcl[t]], da[j][t]);
            }
#pragma unroll
            for (int t = 0; t < TT; ++t) if (t < Ti) {
                av[j][t] = a2[(((size_t)b * Bc + i) * T + t) * S + s];
                L.a[t * SP + s] = av[j][t] * da[j][t];
            }
        }
    }
    __syncthreads();
    float* rowdot = coef + 2 * TMAX;                           // [TMAX]
    for (int t = wave; t < Ti; t += NT / 64) {
        float z = 0.f;
        for (int s = lane; s < S; s += 64) z += L.a[t * SP + s];
        z = wave_sum(z);
        if (lane == 0) rowdot[t] = z;
    }
    __syncthreads();
    // d(gamma1 a1)[s,t] = a2 (d a2 - rowdot[t]); d score[s,:] = a1[s,:] * (d a1[s,:] - <a1[s,:], d a1[s,:]>)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int s = tid + j * NT;
        if (s < S) {
            const float* p1 = a1 + (((size_t)b * Bc + i) * S + s) * T;
            float inner = 0.f;
#pragma unroll
            for (int t = 0; t < TT; ++t) if (t < Ti) {
                da[j][t] = gamma1 * av[j][t] * (da[j][t] - rowdot[t]);       // now d a1[s,t]
                av[j][t] = p1[t];                                            // now a1[s,t]
                inner = fmaf(av[j][t], da[j][t], inner);
            }
#pragma unroll
            for (int t = 0; t < TT; ++t) if (t < T)
                dst[(((size_t)b * Bc + i) * T + t) * S + s] = t < Ti ? av[j][t] * (da[j][t] - inner) : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------- cross-entropies
// sim (R x Q).  Block r < R: softmax over the row (image -> caption), CE against labels[r]; block R + q: softmax over
// column q (caption -> image), CE against labels[q].  Writes the probabilities and per-row / per-column NLL terms.
__global__ __launch_bounds__(NT2) void damsm_ce_fwd_kernel(const float* __restrict__ sim, const int64_t* __restrict__ labels,
                                                          const uint8_t* __restrict__ mask, int R, int Q,
                                                          float* __restrict__ prow, float* __restrict__ pcol,
                                                          float* __restrict__ nll) {
    __shared__ float sh[NT2 / 64];
    const int blk = blockIdx.x, tid = threadIdx.x;
    const bool col = blk >= R;
    const int r = col ? blk - R : blk;
    const int n = col ? R : Q;
    auto at = [&](int k) -> float {
        const int br = col ? k : r, bq = col ? r : k;
        if (mask && mask[(size_t)br * Q + bq]) return -INFINITY;
        return sim[(size_t)br * Q + bq];
    };
    float m = -INFINITY;
    for (int k = tid; k < n; k += NT2) m = fmaxf(m, at(k));
    m = wave_max(m);
    if ((tid & 63) == 0) sh[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    float z = 0.f;
    for (int k = tid; k < n; k += NT2) z += __expf(at(k) - m);
    z = wave_sum(z);
    if ((tid & 63) == 0) sh[tid >> 6] = z;
    __syncthreads();
    z = sh[0] + sh[1] + sh[2] + sh[3];
    const float lz = logf(z);
    float* pout = col ? pcol : prow;
    for (int k = tid; k < n; k += NT2) {
        const int br = col ? k : r, bq = col ? r : k;
        pout[(size_t)br * Q + bq] = __expf(at(k) - m) / z;
    }
    if (tid == 0) {
        const int lab = (int)labels[r];
        nll[blk] = -(at(lab) - m - lz);
    }
}

// out[0] = loss0 = mean_r nll[r], out[1] = loss1 = mean_q nll[R + q]
__global__ __launch_bounds__(NT2) void damsm_ce_finish_kernel(const float* __restrict__ nll, int R, int Q,
                                                             float* __restrict__ out) {
    __shared__ float sh[2][NT2 / 64];
    float a = 0.f, b = 0.f;
    for (int k = threadIdx.x; k < R; k += NT2) a += nll[k];
    for (int k = threadIdx.x; k < Q; k += NT2) b += nll[R + k];
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = (sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]) / (float)R;
        b = (sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]) / (float)Q;
        out[0] = a; out[1] = b;
    }
}

// d sim[r,q] = g0 (prow[r,q] - [q == labels[r]]) / R + g1 (pcol[r,q] - [r == labels[q]]) / Q   (g0, g1 nullable = 0)
__global__ __launch_bounds__(NT2) void damsm_ce_bwd_kernel(const float* __restrict__ prow, const float* __restrict__ pcol,
                                                          const int64_t* __restrict__ labels,
                                                          const float* __restrict__ g0, const float* __restrict__ g1,
                                                          int R, int Q, float* __restrict__ dsim) {
    const int idx = blockIdx.x * NT2 + threadIdx.x;
    if (idx >= R * Q) return;
    const int r = idx / Q, q = idx - r * Q;
    const float a = g0 ? g0[0] * (prow[idx] - ((int)labels[r] == q ? 1.f : 0.f)) / (float)R : 0.f;
    const float b = g1 ? g1[0] * (pcol[idx] - ((int)labels[q] == r ? 1.f : 0.f)) / (float)Q : 0.f;
    dsim[idx] = a + b;
}

// ---------------------------------------------------------------------------------------------- sentence loss
// sim[b,i] = gamma3 * <cnn_b, rnn_i> / max(|cnn_b| |rnn_i|, eps): block per image b, wave per caption stripe
__global__ __launch_bounds__(NT2) void damsm_sent_fwd_kernel(const float* __restrict__ cnn, const float* __restrict__ rnn,
                                                            int Bc, int C, float gamma3, float eps,
                                                            float* __restrict__ sim) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xb = cnn + (size_t)b * C;
    float n0 = 0.f;
    for (int c = lane; c < C; c += 64) n0 = fmaf(xb[c], xb[c], n0);
    n0 = sqrtf(wave_sum(n0));
    for (int i = wave; i < Bc; i += NT2 / 64) {
        const float* ri = rnn + (size_t)i * C;
        float d = 0.f, n1 = 0.f;
        for (int c = lane; c < C; c += 64) { d = fmaf(xb[c], ri[c], d); n1 = fmaf(ri[c], ri[c], n1); }
        d = wave_sum(d); n1 = sqrtf(wave_sum(n1));
        if (lane == 0) sim[(size_t)b * Bc + i] = d / fmaxf(n0 * n1, eps) * gamma3;
    }
}

// d cnn[b,c] = sum_i dsim[b,i] gamma3 ( rnn[i,c] / den - <cnn_b, rnn_i> cnn[b,c] / (|cnn_b|^2 den) ),  den = |cnn_b||rnn_i|
__global__ __launch_bounds__(NT2) void damsm_sent_bwd_kernel(const float* __restrict__ cnn, const float* __restrict__ rnn,
                                                            const float* __restrict__ dsim, int Bc, int C, float gamma3,
                                                            float eps, float* __restrict__ dcnn) {
    extern __shared__ float smem[];           // k1[Bc], k2 (scalar accumulated per caption) -> k1[i], k2[i]
    float* k1 = smem; float* k2 = smem + Bc;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xb = cnn + (size_t)b * C;
    float n0s = 0.f;
    for (int c = lane; c < C; c += 64) n0s = fmaf(xb[c], xb[c], n0s);
    n0s = wave_sum(n0s);
    const float n0 = sqrtf(n0s);
    for (int i = wave; i < Bc; i += NT2 / 64) {
        const float* ri = rnn + (size_t)i * C;
        float d = 0.f, n1 = 0.f;
        for (int c = lane; c < C; c += 64) { d = fmaf(xb[c], ri[c], d); n1 = fmaf(ri[c], ri[c], n1); }
        d = wave_sum(d); n1 = sqrtf(wave_sum(n1));
        if (lane == 0) {
            const float den = n0 * n1, g = dsim[(size_t)b * Bc + i] * gamma3;
            if (den < eps) { k1[i] = g / eps; k2[i] = 0.f; }
            else { k1[i] = g / den; k2[i] = n0s > 0.f ? -g * d / (den * n0s) : 0.f; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NT2) {
        float acc = 0.f, self = 0.f;
        for (int i = 0; i < Bc; ++i) { acc = fmaf(k1[i], rnn[(size_t)i * C + c], acc); self += k2[i]; }
        dcnn[(size_t)b * C + c] = acc + self * xb[c];
    }
}

// out[0] = sum_k w[k] * (*in[k])  (k < n <= 8): sums of 0-dim loss terms without a chain of scalar launches
struct ScalarSumP { const float* in[8]; float w[8]; int n; };
__global__ void scalar_sum_kernel(ScalarSumP p, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int k = 0; k < p.n; ++k) s += p.w[k] * p.in[k][0];
        out[0] = s;
    }
}
struct ScalarScaleP { float* out[8]; float w[8]; int n; };
__global__ void scalar_scale_kernel(const float* __restrict__ g, ScalarScaleP p) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int k = 0; k < p.n; ++k) p.out[k][0] = p.w[k] * g[0];
}

static size_t words_lds_bytes(int C, int S, int T) {
    return ((size_t)C * T + (size_t)T * (S + 1) + 3 * (NT / 64 > NT / TMAX ? NT / 64 : NT / TMAX) * TMAX + 4 * TMAX) *
           sizeof(float);
}

}  // namespace

extern "C" {

int mogan_damsm_words_fwd(const float* ctx, const float* words, const int32_t* cap_lens, int B, int Bc, int C, int S, int T,
                          float gamma1, float gamma2, float gamma3, float* sim, float* a1, float* a2, float* wc, float* wt,
                          hipStream_t stream) {
    if (B <= 0 || Bc <= 0 || C <= 0 || S <= 0 || T <= 0 || T > TMAX || B > 65535) return MOGAN_ERR_SHAPE;
    const size_t lds = words_lds_bytes(C, S, T);
    if (lds > 64 * 1024) return MOGAN_ERR_SHAPE;
#define MOGAN_DAMSM_FWD(TT) hipLaunchKernelGGL(damsm_words_fwd_kernel<TT>, dim3(Bc, B), dim3(NT), lds, stream, ctx, words, \
                                              cap_lens, Bc, C, S, T, gamma1, gamma2, gamma3, sim, a1, a2, wc, wt)
    if (T <= 8) MOGAN_DAMSM_FWD(8); else if (T <= 12) MOGAN_DAMSM_FWD(12); else if (T <= 16) MOGAN_DAMSM_FWD(16);
    else if (T <= 24) MOGAN_DAMSM_FWD(24); else MOGAN_DAMSM_FWD(32);
#undef MOGAN_DAMSM_FWD
    return ok_launch();
}

int mogan_damsm_words_bwd(const float* ctx, const float* words, const int32_t* cap_lens, const float* a1, const float* a2,
                          const float* wc, const float* dsim, int B, int Bc, int C, int S, int T, float gamma1,
                          float gamma2, float gamma3, float* dwc, float* dscore_t, hipStream_t stream) {
    if (B <= 0 || Bc <= 0 || C <= 0 || S <= 0 || T <= 0 || T > TMAX || B > 65535 || S > 2 * NT) return MOGAN_ERR_SHAPE;
    const size_t lds = words_lds_bytes(C, S, T);
    if (lds > 64 * 1024) return MOGAN_ERR_SHAPE;
#define MOGAN_DAMSM_BWD(TT) hipLaunchKernelGGL(damsm_words_bwd_kernel<TT>, dim3(Bc, B), dim3(NT), lds, stream, ctx, words, \
                                              cap_lens, a1, a2, wc, dsim, Bc, C, S, T, gamma1, gamma2, gamma3, dwc, dscore_t)
    if (T <= 8) MOGAN_DAMSM_BWD(8); else if (T <= 12) MOGAN_DAMSM_BWD(12); else if (T <= 16) MOGAN_DAMSM_BWD(16);
    else if (T <= 24) MOGAN_DAMSM_BWD(24); else MOGAN_DAMSM_BWD(32);
#undef MOGAN_DAMSM_BWD
    return ok_launch();
}

int mogan_damsm_ce_fwd(const float* sim, const int64_t* labels, const uint8_t* mask, int R, int Q, float* prow,
                       float* pcol, float* nll, float* out2, hipStream_t stream) {
    if (R <= 0 || Q <= 0 || R != Q) return MOGAN_ERR_SHAPE;      // labels index both axes (losses.py:128-130)
    hipLaunchKernelGGL(damsm_ce_fwd_kernel, dim3(R + Q), dim3(NT2), 0, stream, sim, labels, mask, R, Q, prow, pcol, nll);
    hipLaunchKernelGGL(damsm_ce_finish_kernel, dim3(1), dim3(NT2), 0, stream, (const float*)nll, R, Q, out2);
    return ok_launch();
}

int mogan_damsm_ce_bwd(const float* prow, const float* pcol, const int64_t* labels, const float* g0, const float* g1, int R,
                       int Q, float* dsim, hipStream_t stream) {
    if (R <= 0 || Q <= 0 || R != Q) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(damsm_ce_bwd_kernel, dim3((R * Q + NT2 - 1) / NT2), dim3(NT2), 0, stream, prow, pcol, labels, g0, g1,
                       R, Q, dsim);
    return ok_launch();
}

int mogan_damsm_sent_fwd(const float* cnn, const float* rnn, int B, int Bc, int C, float gamma3, float eps, float* sim,
                         hipStream_t stream) {
    if (B <= 0 || Bc <= 0 || C <= 0) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(damsm_sent_fwd_kernel, dim3(B), dim3(NT2), 0, stream, cnn, rnn, Bc, C, gamma3, eps, sim);
    return ok_launch();
}

int mogan_damsm_sent_bwd(const float* cnn, const float* rnn, const float* dsim, int B, int Bc, int C, float gamma3,
                         float eps, float* dcnn, hipStream_t stream) {
    if (B <= 0 || Bc <= 0 || C <= 0 || (size_t)Bc * 2 * sizeof(float) > 64 * 1024) return MOGAN_ERR_SHAPE;
    hipLaunchKernelGGL(damsm_sent_bwd_kernel, dim3(B), dim3(NT2), (size_t)Bc * 2 * sizeof(float), stream, cnn, rnn, dsim, Bc,
                       C, gamma3, eps, dcnn);
    return ok_launch();
}

int mogan_scalar_sum(const float* const* in, const float* weights, int n, float* out, hipStream_t stream) {
    if (n <= 0 || n > 8) return MOGAN_ERR_SHAPE;
    ScalarSumP p{};
    p.n = n;
    for (int k = 0; k < n; ++k) { p.in[k] = in[k]; p.w[k] = weights[k]; }
    hipLaunchKernelGGL(scalar_sum_kernel, dim3(1), dim3(64), 0, stream, p, out);
    return ok_launch();
}

int mogan_scalar_scale(const float* g, const float* weights, int n, float* const* out, hipStream_t stream) {
    if (n <= 0 || n > 8) return MOGAN_ERR_SHAPE;
    ScalarScaleP p{};
    p.n = n;
    for (int k = 0; k < n; ++k) { p.out[k] = out[k]; p.w[k] = weights[k]; }
    hipLaunchKernelGGL(scalar_scale_kernel, dim3(1), dim3(64), 0, stream, g, p);
    return ok_launch();
}

}  // extern "C"
