// Internal (non-ABI) cross-translation-unit helpers of libmogan_hip.so.  Hidden visibility: the exported
// symbol set stays exactly what include/mogan_hip.h declares.
#ifndef MOGAN_INTERNAL_H
#define MOGAN_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#define MOGAN_HIDDEN __attribute__((visibility("hidden")))

// direct (halo-tile) convolution; return 1 = handled, 0 = shape not eligible, < 0 = error
MOGAN_HIDDEN int mogan_dconv_fwd_try(const float* x, const float* w, float* y, int B, int Cin, int Hs, int Ws, int Cout,
                                     int KH, int KW, int stride, int ph, int pw, int up, void* ws, size_t ws_bytes,
                                     hipStream_t st, const void* d2prep = nullptr, size_t* d2query = nullptr);
MOGAN_HIDDEN int mogan_dconv_dgrad_try(const float* dy, const float* w, float* dx, int B, int Cin, int Hs, int Ws,
                                       int Cout, int KH, int KW, int stride, int ph, int pw, int up, void* ws,
                                       size_t ws_bytes, hipStream_t st, const void* d2prep = nullptr, size_t* d2query = nullptr);
MOGAN_HIDDEN int mogan_dconv_wgrad_try(const float* dy, const float* x, float* dw, int B, int Cin, int Hs, int Ws,
                                       int Cout, int KH, int KW, int stride, int ph, int pw, int up, int accumulate,
                                       void* ws, size_t ws_bytes, hipStream_t st);
// direct convolution, second form (mogan_dconv2.hip: both MFMA operands pre-split; 3x3 / 2x2 stride-1 filters on 8 x 32 tile grids)
MOGAN_HIDDEN int mogan_dconv2_fwd_try(const float* X, const float* w, int wmode, float* Y, int B, int Cin, int Cout, int H, int W,
                                      int OH, int OW, int KH, int KW, int pt, int pl, int yH, int yW, int ys, int npar,
                                      int accumulate, void* ws, size_t ws_bytes, hipStream_t st, const void* prep = nullptr,
                                      size_t* query = nullptr);
// direct VALU kernels for convolutions with <= 4 channels on one side (mogan_smallc.hip); same return convention
MOGAN_HIDDEN int mogan_smallc_fwd_try(const float* x, const float* w, float* y, int B, int Cin, int Hs, int Ws, int Cout,
                                      int KH, int KW, int stride, int ph, int pw, int up, hipStream_t st);
MOGAN_HIDDEN int mogan_smallc_dgrad_try(const float* dy, const float* w, float* dx, int B, int Cin, int Hs, int Ws,
                                        int Cout, int KH, int KW, int stride, int ph, int pw, int up, hipStream_t st);
MOGAN_HIDDEN int mogan_smallc_wgrad_try(const float* dy, const float* x, float* dw, int B, int Cin, int Hs, int Ws,
                                        int Cout, int KH, int KW, int stride, int ph, int pw, int up, int accumulate,
                                        void* ws, size_t ws_bytes, hipStream_t st);
// fused Winograd F(2x2,3x3) for the 3x3 s1 p1 convolutions (mogan_wino.hip): dgrad = 0 forward, 1 data gradient
// (prep: the caller's prepared filter image of this weight version / direction, or nullptr: transformed per call into ws)
MOGAN_HIDDEN int mogan_wino_try(const float* in, const float* w, const void* prep, float* out, int B, int Cin, int H, int W, int Cout,
                                int KH, int KW, int stride, int ph, int pw, int up, int dgrad, const float* ep_scale,
                                const float* ep_shift, int ep_relu, void* ws, size_t ws_bytes, hipStream_t st);
MOGAN_HIDDEN int mogan_wino_wgrad_try(const float* dy, const float* x, float* dw, int B, int Cin, int H, int W, int Cout,
                                      int KH, int KW, int stride, int ph, int pw, int up, int accumulate, void* ws,
                                      size_t ws_bytes, hipStream_t st);
// out[i] = (accumulate ? out[i] : 0) + sum_s ws[s * n + i], i < n, fixed order (the split-K slabs of a dense output)
MOGAN_HIDDEN void mogan_splitk_reduce_dense(const float* ws, float* out, long long n, int nsplit, int accumulate, hipStream_t st);
// split-K block target of a launch on `st` (mogan_gemm_set_split_target / mogan_stream_set_split_target)
MOGAN_HIDDEN int mogan_split_target(hipStream_t st);
// mogan_stem.hip: conv4x4 s2 p1 from 3 input channels + LeakyReLU(slope) (slope = 1: none) as one streaming kernel; 1 = handled
MOGAN_HIDDEN int mogan_stem_fwd_try(const float* x, const float* w, float* y, int B, int Cin, int H, int W, int Cout, int KH, int KW,
                                    int stride, int ph, int pw, float slope, hipStream_t st);
// Prepared filter images of dconv2_fwd_kernel (round 6; the 4x4 s2 p1 convolutions: forward over the space-to-depth image, data
// gradient by parity classes).  The try functions of the direct kernels take two optional arguments: d2prep = the caller's image of
// THIS weight version (used instead of a prep launch when the call lands on dconv2_fwd_kernel's 4x4 s2 forms), d2query != nullptr = a
// DRY RUN of the dispatch: nothing is launched, *d2query receives the size of that image (0: none) and the return value says whether
// the geometry would have been taken.
// filter images of n 4x4 s2 weights (dgrad[i] = 0: forward form, 1: data-gradient form) in one launch per 32 members
MOGAN_HIDDEN int mogan_dconv2_prep_group(int n, const float* const* w, void* const* prep, const int* Cout, const int* Cin,
                                         const int* dgrad, hipStream_t st);
MOGAN_HIDDEN void mogan_prof_begin(int mode, int cfg, double flops, int M, int N, int K, hipStream_t st);
MOGAN_HIDDEN void mogan_prof_end(int taken, hipStream_t st);
MOGAN_HIDDEN void mogan_prof_relabel(int cfg);     // the open launch record's kernel id (2 = dconv2_fwd_kernel)
MOGAN_HIDDEN extern int mogan_use_dconv;
#endif
