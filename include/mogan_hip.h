/* mogan_hip.h -- C ABI of libmogan_hip.so: the MI355X (gfx950) kernels under the AttnGAN G+D
 * train step of tohinz/multiple-objects-gan.
 *
 * The reference has no FFI: its hot path is stock torch.nn ops composed in python
 * (code/coco/attngan/model.py, GlobalAttention.py, miscc/losses.py, trainer.py:281-342).  Each
 * entry point below replaces the torch op(s) at the cited reference lines; the python package
 * binds them with ctypes (multiple-objects-gan_amd/hip/lib.py) and wraps them in
 * torch.autograd.Function (hip/ops.py).  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - arithmetic: fp32 tensors, fp32 accumulation.  The matrix kernels form every fp32 product from the exact three-piece bf16
 *     split of both operands on the bf16 MFMA pipe (six partial products, dropped terms <= 2^-23 |a b|; csrc/mogan_mma.h):
 *     error against fp64 at the level of the native fp32 MFMA instruction, not bit-identical to an fmaf
 *     chain.  mogan_mfma_form() reports the form; -DMOGAN_X6=0 builds the native one;
 *   - all tensors are dense fp32, NCHW, device pointers, borrowed for the duration of the call
 *     (the caller -- PyTorch -- owns the memory); uint8 masks / int32 lengths where stated;
 *   - every call is asynchronous on `stream` (pass torch.cuda.current_stream().cuda_stream) and
 *     allocates nothing: scratch comes from the caller as (ws, ws_bytes); a null/short workspace
 *     only disables split-K (slower, never wrong) unless stated;
 *   - return 0 on success, negative MOGAN_ERR_* otherwise (no exceptions cross the boundary);
 *   - operators keep no state between calls and are thread-safe / re-entrant.  What the library does
 *     hold is TUNING state, never data: a split-K block target (process default + per-stream override,
 *     mogan_gemm_set_split_target / mogan_stream_set_split_target) and the tuned (tile, split-K) table
 *     (mogan_gemm_tune_set), both read under a mutex -- results do not depend on them beyond the
 *     summation order across K-splits; plus the test / measurement hooks marked as such below
 *     (mogan_gemm_debug_force, mogan_prof_*), which are process-wide and
 *     not meant for concurrent use.
 *
 * Arithmetic.  fp32 in, fp32 out, fp32 accumulators everywhere.  In the default build the MFMA kernels (convolutions, bmm)
 * form every product a*b on the bf16 matrix pipe from the exact three-piece bf16 split of both operands (csrc/mogan_mma.h:
 * six partial products, the dropped ones <= 2^-23 |a b|), mogan_mfma_form() == 6.  Measured against fp64 this is within
 * +-25 % of the native fp32 MFMA (libmogan_hip_f32.so, mogan_mfma_form() == 1) on ordinary and on wide-dynamic-range data;
 * the forward error bound  |err| <= 2^-24 * sum |a||b|  holds in every measured case, but under engineered cancellation
 * (result ~1e-4 of the terms) the error relative to the tiny RESULT is up to 17x the native form's, whose fma chain profits
 * from the exact cancellation of adjacent terms (profiles/r02_precision_*.txt).  Input domain of the split form:
 * |x| <= 3.3895e38 (the largest bf16; larger finite values and +-inf give inf / NaN where the native form may stay finite);
 * |x| >= 2^-110 or 0 for full accuracy (below, the third / second piece drops under the smallest normal bf16 and the product
 * keeps 16 / 8 significant bits -- on values whose products are below fp32's normal range anyway).
 * tests/test_kernels_gpu.py::test_fp32_products_on_the_bf16_pipe_* keep these statements under test.
 */
#ifndef MOGAN_HIP_H
#define MOGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* identical to the typedef in <hip/hip_runtime_api.h>; repeated so plain C / ctypes-side tools can
 * include this header without the HIP SDK */
typedef struct ihipStream_t* hipStream_t;

#define MOGAN_ERR_SHAPE (-1)  /* unsupported / inconsistent dimensions */
#define MOGAN_ERR_LAUNCH (-2) /* hip launch error */
#define MOGAN_ERR_WS (-3)     /* workspace required but too small */

/* activation codes shared by the norm / elementwise families */
#define MOGAN_ACT_NONE 0
#define MOGAN_ACT_RELU 1
#define MOGAN_ACT_LRELU 2 /* LeakyReLU(slope) */
#define MOGAN_ACT_GLU 3   /* x[:, :C/2] * sigmoid(x[:, C/2:])   (model.py:24-32) */
#define MOGAN_ACT_TANH 4
#define MOGAN_ACT_SIGMOID 5

int mogan_abi_version(void);
/* split-K of the implicit-GEMM kernels aims at `blocks` workgroups per launch (default 768 = three per CU for a kernel
 * that has the GPU to itself; a caller that keeps several streams busy lowers it to 384: fewer slabs to reduce). */
int mogan_gemm_set_split_target(int blocks);
/* the same per stream (0 = back to the process default): launches on `stream` use this target.  The owner of a stream
 * knows whether its kernels run alone (768) or beside other streams' kernels (384); two engines / threads with their own
 * streams do not interfere. */
int mogan_stream_set_split_target(hipStream_t stream, int blocks);

/* Tuned dispatch of the implicit-GEMM kernel: for the GEMM (mode 0 conv fwd / 1 conv dgrad / 2 conv wgrad / 3 bmm; M, N, K,
 * nz = batch or parity classes, exactly as the kernel sees them) use tile config `cfg` (0..4) and split-K factor `split`
 * instead of the heuristic.  Entries come from timing every (cfg, split) pair on the device (tools/tune_gemm.py ->
 * multiple-objects-gan_amd/hip/tuned_gemm_gfx950.csv); the host registers them once after loading the library.
 * mogan_gemm_tune_clear() drops all entries.  Results do not depend on the entry (same sums per split, fixed order). */
int mogan_gemm_tune_set(int mode, int M, int N, int K, int nz, int cfg, int split);
int mogan_gemm_tune_clear(void);

/* Process set-up, not an operator: create `n` idle non-blocking HIP streams that live until the process ends.  HIP
 * multiplexes the streams of a process onto GPU_MAX_HW_QUEUES hardware queues in the order the streams are created, and
 * streams that share a queue are processed in order; the multi-stream train step calls this once, right after selecting
 * the device and before any other stream exists, so that its own streams (and RCCL's) land on the queues in the measured
 * arrangement ("hardware queues").  Returns the number of streams held, or a negative error. */
int mogan_reserve_streams(int n);

/* test hook: force a GEMM tile config (0..4; -1 = the default dispatch; -2 = the default dispatch without the Winograd kernels, so
 * that 3x3 stride-1 shapes reach the direct kernels) and a split-K factor (0 = heuristic) */
int mogan_gemm_debug_force(int cfg, int split);
/* tuning hook: grouped launches (mogan_conv2d_*_group) with fewer tiles than this run their members one by one (default 1600, see csrc/mogan_gemm.hip) */
int mogan_gemm_group_min_tiles(int tiles);
/* which matrix instruction the MFMA kernels of this build use for their fp32 products: 6 = split-bf16 form (three bf16 pieces
 * per operand, six v_mfma_f32_32x32x16_bf16 partial products per 16 k; csrc/mogan_mma.h -- the default), 1 = native
 * v_mfma_f32_32x32x2_f32 (-DMOGAN_X6=0).  bench.py prices its roofline against the matching peak. */
int mogan_mfma_form(void);

/* measurement hook (bench.py roofline leg): with profiling enabled every gemm_kernel launch is bracketed by
 * HIP events on its own stream; collect() returns rows of 5 doubles {mode (0 fwd,1 dgrad,2 wgrad,3 bmm),
 * tile config, launches, algorithmic flops = sum 2*M*N*K of the true GEMM dims, milliseconds} and the row count.
 * Not for use under hipGraph capture. */
int mogan_prof_enable(int on);
int mogan_prof_collect(double* out, int max_rows);
/* same records as one CSV row per launch (geometry, algorithmic GFLOP, ms); consumes them */
int mogan_prof_dump(const char* path);

/* ---------------------------------------------------------------- convolution (fp32 MFMA implicit GEMM)
 * x (B,Cin,Hs,Ws), w (Cout,Cin,KH,KW), y (B,Cout,OH,OW), no bias.  up=1 fuses nn.Upsample(x2,nearest)
 * in front of the conv (upBlock, model.py:48-55): the conv then sees H=2Hs, W=2Ws.
 * OH = (H + 2*ph - KH)/stride + 1.  Replaces nn.Conv2d at model.py:35-44,92-99,587,598-609,626,664-677. */
int mogan_conv2d_out_dims(int Hs, int Ws, int KH, int KW, int stride, int ph, int pw, int up, int* OH, int* OW);
int mogan_conv2d_fwd(const float* x, const float* w, float* y, int B, int Cin, int Hs, int Ws, int Cout, int KH,
                     int KW, int stride, int ph, int pw, int up, void* ws, size_t ws_bytes, hipStream_t stream);
/* The discriminators' logits head in one launch each way: p (B,Cout) = sigmoid(<x[b], w[co]> + bias[co]) over K = Cin*KH*KW --
 * nn.Conv2d(8 ndf, 1, kernel_size=4, stride=4) (with bias) on a 4x4 map followed by nn.Sigmoid (model.py:626-627, 640-641);
 * x (B,K) and w (Cout,K) dense, K % 4 == 0, Cout <= 4.  Backward from dp and p: dx (B,K) (NULL = not wanted), dw (Cout,K) and
 * db (Cout) written or accumulated (NULL = not wanted; the images are summed in order). */
int mogan_logits_head_fwd(const float* x, const float* w, const float* bias, float* p, int B, int K, int Cout,
                          hipStream_t stream);
int mogan_logits_head_bwd(const float* dp, const float* p, const float* x, const float* w, float* dx, float* dw, float* db,
                          int B, int K, int Cout, int accumulate, hipStream_t stream);
/* z = LeakyReLU_slope(conv2d(x, w)) in one launch: the first layer of every discriminator -- nn.Conv2d(3, ndf, 4, 2, 1)
 * followed by nn.LeakyReLU(0.2) with no BatchNorm in between (model.py:597-598, 660-661).  Returns 0 = done, 1 = not a geometry
 * for this kernel (run mogan_conv2d_fwd + mogan_act_fwd), < 0 = error.  Backward: mogan_act_bwd with z in place of the
 * pre-activation (same sign), then the convolution's gradients. */
int mogan_conv2d_lrelu_fwd(const float* x, const float* w, float* z, int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW,
                           int stride, int ph, int pw, float slope, void* ws, size_t ws_bytes, hipStream_t stream);
/* conv (no upsample) with y = relu?(scale[co]*conv + shift[co]) applied in the kernel epilogue: BasicConv2d of the frozen,
 * eval-mode Inception trunk (conv -> BN on running statistics -> ReLU; model.py:258-299) in one pass over the output */
int mogan_conv2d_affine_fwd(const float* x, const float* w, const float* scale, const float* shift, float* y, int B,
                            int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int relu,
                            void* ws, size_t ws_bytes, hipStream_t stream);
/* its backward up to the conv: dx = dy * scale[c] * (y > 0), from the saved OUTPUT y (B,C,HW) */
int mogan_affine_relu_bwd_out(const float* y, const float* dy, const float* scale, float* dx, int B, int C, int HW,
                              hipStream_t stream);
/* dx: gradient w.r.t. the conv input in the H x W domain, (B,Cin,H,W); for up=1 follow with mogan_down2_sum */
/* Grouped launches for the frozen trunk: up to 4 INDEPENDENT convolutions of one dependency level of a Mixed block (same
 * direction, arbitrary geometries) run as one launch whose 1-D grid is the concatenation of the problems' tile grids -- at
 * B = 16 each of them alone needs split-K (+ a reduction launch) to fill the chip.  Field meaning = the arguments of
 * mogan_conv2d_affine_fwd_ex / mogan_conv2d_dgrad_ex.  Outputs of a group must not overlap. */
typedef struct MoganConvFwdArgs {
    const float* x; long long x_bstride; const float* w; const float* scale; const float* shift;
    float* y; long long y_bstride; float* y2; long long y2_bstride; int msplit;
    int B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, relu;
} MoganConvFwdArgs;
typedef struct MoganConvDgradArgs {
    const float* dy; long long dy_bstride; const float* w; float* dx; long long dx_bstride;
    const float* relu_of; long long relu_bstride; int accumulate;
    int B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw;
} MoganConvDgradArgs;
int mogan_conv2d_affine_fwd_group(int n, const MoganConvFwdArgs* args, void* ws, size_t ws_bytes, hipStream_t stream);
int mogan_conv2d_dgrad_group(int n, const MoganConvDgradArgs* args, void* ws, size_t ws_bytes, hipStream_t stream);
/* Channel-slice addressing for the frozen Inception trunk (model.py:258-299): x is a slice (batch stride x_bstride
 * elements) of a larger NCHW tensor; output channels [0, msplit) go to y (batch stride y_bstride, -1 = dense), channels
 * [msplit, Cout) to y2 (y2 nullable: everything to y).  One launch then serves a group of same-input 1x1 convolutions whose
 * parts land in different tensors, and every branch of a Mixed block writes straight into the block's concatenated
 * output. */
int mogan_conv2d_affine_fwd_ex(const float* x, long long x_bstride, const float* w, const float* scale, const float* shift,
                               float* y, long long y_bstride, float* y2, long long y2_bstride, int msplit, int B, int Cin,
                               int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int relu, void* ws,
                               size_t ws_bytes, hipStream_t stream);
/* Plain forward convolution with channel-slice addressing, a ReLU mask on the result (relu_of: shaped like y, nullable) and
 * accumulation into y.  Used as the data gradient of the frozen trunk's stride-1 convolutions: dX = conv(dY, flipped and
 * (ci,co)-transposed filters, pad K-1-pad) reads the filter operand K-contiguously (the data-gradient mode gathers it with a
 * stride of KH*KW floats). */
int mogan_conv2d_fwd_ex(const float* x, long long x_bstride, const float* w, float* y, long long y_bstride,
                        const float* relu_of, long long relu_bstride, int accumulate, int B, int Cin, int Hs, int Ws,
                        int Cout, int KH, int KW, int stride, int ph, int pw, void* ws, size_t ws_bytes, hipStream_t stream);
/* Data gradient with channel-slice addressing and a fused ReLU backward: dy is a slice with batch stride dy_bstride, the
 * result is written (accumulate 0) or added (1) to the slice dx (batch stride dx_bstride); where relu_of[...] <= 0 (a
 * slice shaped like dx with batch stride relu_bstride; nullable) the new contribution is zeroed first. */
int mogan_conv2d_dgrad_ex(const float* dy, long long dy_bstride, const float* w, float* dx, long long dx_bstride,
                          const float* relu_of, long long relu_bstride, int accumulate, int B, int Cin, int Hs, int Ws,
                          int Cout, int KH, int KW, int stride, int ph, int pw, void* ws, size_t ws_bytes,
                          hipStream_t stream);
int mogan_conv2d_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Hs, int Ws, int Cout, int KH,
                       int KW, int stride, int ph, int pw, int up, void* ws, size_t ws_bytes, hipStream_t stream);
/* ---- Prepared filter images for the Winograd convolutions (round 6; csrc/mogan_wino.hip).  The 3x3 stride-1 convolutions of the
 * generator's ResBlocks (code/coco/attngan/model.py:67-81) run as fused Winograd F(2x2,3x3); their filters enter the kernel
 * transformed (G g G^t) and pre-split into bf16 pieces.  mogan_conv2d_fwd / mogan_conv2d_dgrad build that image per call at the
 * head of the workspace -- one more launch on the convolution's own chain, per use.  A caller that owns the weights (the
 * optimizer) keeps the image instead, one per weight and direction, rebuilds ALL images of a network in one launch right after
 * its optimizer step, and hands it to the *_wp entry points; the library keeps no weight state.
 *   mogan_wino_prep_bytes   size of the image for this convolution geometry and direction (dgrad = 0 forward, 1 data gradient);
 *                           0 = this convolution does not take the Winograd kernels (no image is needed, wprep stays NULL)
 *   mogan_wino_prep_group   images of n (weight, direction) pairs in one launch (32 per launch beyond that); prep[i] 16-byte aligned
 *   mogan_conv2d_fwd_wp /   mogan_conv2d_fwd / mogan_conv2d_dgrad with the caller's image of w for that direction (wprep NULL: the
 *   mogan_conv2d_dgrad_wp   plain entry points' behaviour).  The image must have been built from the current values of w. */
size_t mogan_wino_prep_bytes(int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int up, int dgrad);
int mogan_wino_prep_group(int n, const float* const* w, void* const* prep, const int* Cout, const int* Cin, const int* dgrad,
                          hipStream_t stream);
/* The same for whichever kernel a convolution's geometry takes (round 6, third session): besides the Winograd image of a 3x3 s1 p1
 * convolution, the pre-split filter image of dconv2_fwd_kernel for the discriminators' 4x4 s2 p1 convolutions
 * (code/coco/attngan/model.py:575-613: forward over the space-to-depth image, data gradient by parity classes; csrc/mogan_dconv2.hip),
 * which mogan_conv2d_fwd / _dgrad otherwise rebuild per call (44 prep launches per train step for weights that change once).
 *   mogan_conv_prep_bytes   as mogan_wino_prep_bytes for any geometry: the size of the image mogan_conv2d_fwd_wp / _dgrad_wp expect as
 *                           wprep for it (found by a dry run of the same dispatch), 0 = none
 *   mogan_conv_prep_group   as mogan_wino_prep_group with the filter size KH[i] (3 or 4) per member: one launch per kind and 32 members */
size_t mogan_conv_prep_bytes(int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int up, int dgrad);
int mogan_conv_prep_group(int n, const float* const* w, void* const* prep, const int* Cout, const int* Cin, const int* KH,
                          const int* dgrad, hipStream_t stream);
int mogan_conv2d_fwd_wp(const float* x, const float* w, const void* wprep, float* y, int B, int Cin, int Hs, int Ws, int Cout, int KH,
                        int KW, int stride, int ph, int pw, int up, void* ws, size_t ws_bytes, hipStream_t stream);
int mogan_conv2d_dgrad_wp(const float* dy, const float* w, const void* wprep, float* dx, int B, int Cin, int Hs, int Ws, int Cout,
                          int KH, int KW, int stride, int ph, int pw, int up, void* ws, size_t ws_bytes, hipStream_t stream);
/* dw (Cout,Cin,KH,KW); accumulate != 0 adds into dw */
int mogan_conv2d_wgrad(const float* dy, const float* x, float* dw, int B, int Cin, int Hs, int Ws, int Cout, int KH,
                       int KW, int stride, int ph, int pw, int up, int accumulate, void* ws, size_t ws_bytes,
                       hipStream_t stream);

/* ---- Weight-heavy convolutions on pre-split "panels" (csrc/mogan_pgemm.hip): the deep discriminator layers
 * (code/coco/attngan/model.py:594-613, 616-642, 738-760) are GEMMs with 768..3072 x 6144..27648 weights against a few hundred
 * pixels.  Their weights change once per optimizer step but are used by the real, the fake and the generator pass, so the
 * caller keeps -- next to the fp32 master -- a PACKED copy per direction: the three bf16 pieces of every weight (see
 * "Arithmetic" above) in the order the matrix instruction consumes them.  The convolution entry points below read the packed
 * copy with plain 16-byte loads and pack the activations per call into the workspace (channels-last bf16 pieces); results are
 * the same fp32 products as mogan_conv2d_fwd / mogan_conv2d_dgrad up to the summation order.
 *   mogan_pk_conv_eligible   1 if the geometry meets the panel formats' constraints (forward: Cin % 32 == 0; data gradient:
 *                            Cout % 32 == 0, KH, KW, Hs, Ws multiples of the stride) AND the layer is weight-heavy enough for
 *                            the path to pay (<= 64 output pixels per image and parity class, K >= 1024, >= 128 rows); else 0
 *   mogan_pk_weight_bytes    size of the packed copy for one direction (dgrad = 0 forward, 1 data gradient); 0 = not packable
 *   mogan_pk_weight_pack     w (Cout,Cin,KH,KW) fp32 -> packed copy; the caller re-packs after every change of w (it owns the
 *                            buffer and the bookkeeping: the library keeps no weight state)
 *   mogan_conv2d_fwd_pk      y = conv2d(x, w) from the forward-packed weights;  workspace: B*Hs*Ws*Cin*6 bytes + split-K slabs
 *   mogan_conv2d_dgrad_pk    dx = conv2d data gradient from the dgrad-packed weights; workspace: B*OH*OW*Cout*6 bytes + slabs
 *   mogan_pk_debug_force     test hook (process-wide): take_all != 0 drops the size heuristic of mogan_pk_conv_eligible (hard
 *                            constraints stay); cfg in 0..2 forces a tile shape (-1 = heuristic), split > 0 a K-split count */
int mogan_pk_conv_eligible(int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int dgrad);
size_t mogan_pk_weight_bytes(int Cout, int Cin, int KH, int KW, int stride, int dgrad);
int mogan_pk_weight_pack(const float* w, void* wpk, int Cout, int Cin, int KH, int KW, int stride, int ph, int pw, int dgrad,
                         hipStream_t stream);
/* both copies from one read of w (Cin % 32 == 0 and Cout % 32 == 0): what the owner of the weight calls after its optimizer step */
int mogan_pk_weight_pack_both(const float* w, void* wpk_fwd, void* wpk_dgrad, int Cout, int Cin, int KH, int KW, int stride, int ph,
                              int pw, hipStream_t stream);
int mogan_conv2d_fwd_pk(const float* x, const void* wpk, float* y, int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW,
                        int stride, int ph, int pw, void* ws, size_t ws_bytes, hipStream_t stream);
int mogan_conv2d_dgrad_pk(const float* dy, const void* wpk, float* dx, int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW,
                          int stride, int ph, int pw, void* ws, size_t ws_bytes, hipStream_t stream);
int mogan_pk_debug_force(int take_all, int cfg, int split);
/* weight gradient of the same weight-heavy layers on the packed kernels: dY and the transposed im2col matrix of x are packed per
 * call into the workspace (ws_bytes >= what mogan_pk_wgrad_eligible checks: ~ (Cout + Cin*KH*KW) * B*OH*OW * 6 bytes), dW
 * (Cout,Cin,KH,KW) is written or (accumulate != 0) added to.  Same fp32 products as mogan_conv2d_wgrad, other summation order. */
int mogan_pk_wgrad_eligible(int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, size_t ws_bytes);
int mogan_conv2d_wgrad_pk(const float* dy, const float* x, float* dw, int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW,
                          int stride, int ph, int pw, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);

/* Deep block = conv -> BatchNorm2d (training statistics) -> LeakyReLU / ReLU / nothing on a map with B*OH*OW <= 2048 values per
 * channel (downBlock / Block3x3_leakRelu of the deep discriminator layers, model.py:575-613, 616-642): the packed-weight GEMM
 * followed by ONE tail kernel that sums the K-split slabs, computes the batch statistics (8 channels per block, a channel's
 * values stay in registers), updates the running statistics, applies BN + activation and writes the NEXT layer's pixel panel --
 * 2 launches where conv + split-K reduce + bn_stats (2) + bn_act_fwd + the activation pack took 6; the backward is one tail
 * kernel (activation + BN backward, d gamma / d beta, the gradient's pixel panel) + the packed data-gradient GEMM.
 *   forward   x (B,Cin,Hs,Ws) fp32 and/or its pixel panel xpanel (nullable: packed here into the workspace), wpk = forward
 *             packed weights; outputs y = conv(x) (B,Cout,OH,OW: the BN input, kept for the backward), stats = mean[Cout] |
 *             invstd[Cout], z = act(BN(y)), zpanel (nullable) = pixel panel of z, mogan_pk_panel_bytes(B, Cout, OH*OW) bytes;
 *             rmean / rvar updated like nn.BatchNorm (momentum, unbiased variance), nullable
 *   backward  dz -> dy (B,Cout,OH,OW: gradient at the conv output, what the weight gradient needs), dgamma / dbeta (nullable;
 *             accumulate != 0 adds), dx (nullable: no data gradient) from wpk_dgrad = data-gradient packed weights
 *   groups    1, or 2 (round 5): the B images are `groups` batches of B / groups images one behind the other which the reference
 *             passes through the layer in separate calls -- D(real) and D(fake.detach()) of a discriminator update,
 *             miscc/losses.py:136-174 --: ONE convolution over all B images (the weights stream once), batch statistics per
 *             group (stats = mean[groups][Cout] | invstd[groups][Cout]), running statistics updated group after group in that
 *             order, d gamma / d beta = the groups' sums.  B / groups * OH*OW <= 2048.
 * Same arithmetic as mogan_bn_stats / mogan_bn_act_fwd / mogan_bn_act_bwd (fp64 sums). */
size_t mogan_pk_panel_bytes(int B, int C, int HW);
int mogan_deep_block_eligible(int B, int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int act,
                              int groups);
int mogan_deep_conv_bn_act_fwd(const float* x, const void* xpanel, const void* wpk, const float* gamma, const float* beta,
                               float* rmean, float* rvar, float* y, float* stats, float* z, void* zpanel, int B, int Cin, int Hs,
                               int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, float eps, float momentum, int act,
                               float slope, int groups, void* ws, size_t ws_bytes, hipStream_t stream);
int mogan_deep_conv_bn_act_bwd(const float* dz, const float* y, const float* stats, const float* gamma, const float* beta,
                               const void* wpk_dgrad, float* dy, float* dgamma, float* dbeta, int accumulate, float* dx, int B,
                               int Cin, int Hs, int Ws, int Cout, int KH, int KW, int stride, int ph, int pw, int act, float slope,
                               int groups, void* ws, size_t ws_bytes, hipStream_t stream);

/* ---- Frozen CNN_ENCODER trunk on pixel panels (code/coco/attngan/model.py:258-299: Mixed_5b .. Mixed_7c of the frozen, eval-mode
 * Inception-v3; trainer.py:329-333 sends only the image gradient through it).  With frozen weights every filter is packed ONCE
 * (mogan_pk_weight_pack of the, where needed zero-padded, filters); activations and gradients travel between the layers as pixel
 * panels (channels-last bf16 pieces, 192 bytes per pixel and group of 32 channels; a tensor's channel slices start at multiples
 * of 32), so a convolution is the packed-operand GEMM alone and everything around it -- K-split sum, eval-mode BatchNorm affine,
 * ReLU / ReLU mask, the 3x3 average pool of the pool branch (moved behind the 1x1 convolution: both are linear), accumulation of a
 * gradient with several contributors, the fp32 copy and the next layer's panel -- is ONE tail launch per dependency level.
 *   mogan_pk_group          n <= 8 independent GEMMs in one launch: forward (dgrad = 0: rows = Cout, K = KH*KW*Cp, tap-major) or
 *                           stride-1 data gradient (dgrad = 1: rows = Cin, K = KH*KW*Cp over the dY panel).  raw receives
 *                           nsplit slabs of (B, M, outH, outW) fp32; nsplit is in/out (asked for / written).
 *   mogan_panel_tail_group  n <= 8 slices: v = sum of the sources' slabs; [box: 3x3 mean, zero padding, divisor 9];
 *                           [v = v * scale[c] + shift[c]]; [relu]; [v += add]; [v = 0 where mask <= 0]; -> dst (fp32, nullable)
 *                           and the panel slice (nullable; channels n .. roundup32(n) are written as zeros). */
typedef struct MoganPkArgs {
    const void* wpk; const void* panel; float* raw;
    int B, M;                 /* images, rows of the result */
    int Cp, CGp, cg0;         /* channels of the K range per tap (multiple of 32); channel groups per pixel of the panel; first group */
    int PH, PW, outH, outW;   /* image dims of the panel / of the result */
    int KH, KW, stride, ph, pw, dgrad;
    int nsplit;
} MoganPkArgs;
#define MOGAN_TAIL_MAXSRC 3
typedef struct MoganTailArgs {
    const float* src[MOGAN_TAIL_MAXSRC]; long long src_bs[MOGAN_TAIL_MAXSRC]; long long src_slab[MOGAN_TAIL_MAXSRC];
    int src_nsplit[MOGAN_TAIL_MAXSRC]; int nsrc;
    const float* add; long long add_bs;
    const float* mask; long long mask_bs;
    const float* scale; const float* shift;
    int relu, box;
    float* dst; long long dst_bs;
    void* panel; int CGp, cg0;
    int B, n, H, W;           /* images, channels of the slice, map */
} MoganTailArgs;
int mogan_pk_group(int n, MoganPkArgs* args, hipStream_t stream);
int mogan_panel_tail_group(int n, const MoganTailArgs* args, hipStream_t stream);

/* nn.Upsample(scale_factor=2, mode='nearest') + conv3x3(padding 1, no bias) -- every upBlock of the reference
 * (code/coco/attngan/model.py:48-55, code/coco/stackgan/model.py:16-22) -- evaluated as the TRANSPOSED 4x4
 * stride-2 pad-1 convolution with kernel K = T w T^t, T = [[0,0,1],[0,1,1],[1,1,0],[1,0,0]]: each phase of the
 * upsampled grid sees only 2x2 distinct source pixels, so 4 multiply-adds per output pixel replace 9 (same result up
 * to the fp32 rounding of the pre-summed weights).  x (B,Cin,Hs,Ws), w (Cout,Cin,3,3), y (B,Cout,2Hs,2Ws).
 * dgrad returns dx at the SOURCE resolution (B,Cin,Hs,Ws) (no mogan_down2_sum).  The workspace must hold
 * mogan_upconv3x3_ws_bytes(Cout,Cin) for K (dK in wgrad) in front of the split-K scratch of the inner conv. */
size_t mogan_upconv3x3_ws_bytes(int Cout, int Cin);
/* K (Cin, Cout, 4, 4) of w alone: a caller that owns w keeps K per weight version (and, with mogan_conv_prep_*, the filter image of
 * the kernel that runs the virtual 4x4 s2 convolution) and then calls mogan_conv2d_dgrad_wp(x, K, image, y, B, Cout, 2 Hs, 2 Ws, Cin,
 * 4, 4, 2, 1, 1, 0, ...) for the forward / mogan_conv2d_fwd_wp(dy, K, image, dx, ...) for the data gradient -- the calls
 * mogan_upconv3x3_fwd / _dgrad make after building K per call (same results) */
int mogan_upconv3x3_k4(const float* w, float* k4, int Cout, int Cin, hipStream_t stream);
/* the K of n weights in one launch (what an owner calls behind its optimizer step) */
int mogan_upconv3x3_k4_group(int n, const float* const* w, float* const* k4, const int* Cout, const int* Cin, hipStream_t stream);
int mogan_upconv3x3_fwd(const float* x, const float* w, float* y, int B, int Cin, int Hs, int Ws, int Cout, void* ws,
                        size_t ws_bytes, hipStream_t stream);
int mogan_upconv3x3_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Hs, int Ws, int Cout, void* ws,
                          size_t ws_bytes, hipStream_t stream);
int mogan_upconv3x3_wgrad(const float* dy, const float* x, float* dw, int B, int Cin, int Hs, int Ws, int Cout,
                          int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);
/* backward of nearest x2 upsample: dx[b,c,y,x] = sum of the 2x2 block of du (B*C planes of 2H x 2W) */
int mogan_down2_sum(const float* du, float* dx, int planes, int H, int W, hipStream_t stream);

/* generic strided batched GEMM: C[z][m][n] (+)= sum_k A[z][m][k] * B[z][k][n]; strides in elements.
 * Replaces nn.Linear (model.py:324,365,371) and torch.bmm (GlobalAttention.py:46,66,100,118). */
int mogan_bmm(const float* a, const float* b, float* c, int batch, int M, int N, int K, long long sAb, long long sAm,
              long long sAk, long long sBb, long long sBk, long long sBn, long long sCb, long long sCm, long long sCn,
              int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);

/* ---------------------------------------------------------------- batch norm (+ fused activation)
 * x (B,C,HW) [HW=1: BatchNorm1d].  Training-mode statistics: mean[C], invstd[C] = 1/sqrt(var_biased+eps);
 * running_mean/var (nullable) updated with `momentum` (unbiased var), as nn.BatchNorm*d does.
 * ws: >= mogan_bn_ws_bytes(B,C,HW) bytes, REQUIRED. */
size_t mogan_bn_ws_bytes(int B, int C, int HW);
int mogan_bn_stats(const float* x, int B, int C, int HW, float eps, float momentum, float* mean, float* invstd,
                   float* running_mean, float* running_var, void* ws, size_t ws_bytes, hipStream_t stream);
/* y = act(gamma*(x-mean)*invstd + beta) (+ residual).  GLU: y has C/2 channels. residual nullable, shaped like y.
 * BN+GLU: model.py:52-54,62-63,72-73,366-367; BN+LeakyReLU: 96-101,579,588-589,602-611; BN+ReLU: 372-373;
 * BN+residual: 75,80. */
int mogan_bn_act_fwd(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     const float* residual, float* y, int B, int C, int HW, int act, float slope, hipStream_t stream);
/* The running-statistics update of a BatchNorm call that was made with running_mean = running_var = NULL, from the batch
 * statistics that call wrote (mean / invstd, n = B*HW values per channel): nn.BatchNorm's momentum update with the unbiased
 * variance.  Lets a call's arithmetic run EARLIER than its place in the reference's call order while the running buffers are
 * updated in that order: round 5 evaluates the real-image terms of discriminator_loss (miscc/losses.py:146-160: D(real), the
 * conditional head on real and on the "wrong" pairs) and back-propagates them before the fake images exist; the reference
 * calls COND_DNET on the fake features BEFORE the wrong pairs, so the wrong call's update is deferred behind the fake call's. */
int mogan_bn_running_update(const float* mean, const float* invstd, float* running_mean, float* running_var, int C, long long n,
                            float eps, float momentum, hipStream_t stream);

/* mogan_bn_stats + mogan_bn_act_fwd in one call (training-mode BatchNorm + activation, model.py:48-81, 575-613): maps with
 * B*HW <= 4096 values per channel (and HW >= 16) take ONE launch -- a block per output channel reduces, finalises and applies --,
 * larger ones the three launches of the two calls above; mean / invstd [C] are written for mogan_bn_act_bwd either way (which
 * takes the matching one-launch kernel for the same shapes).  ws as for mogan_bn_stats. */
int mogan_bn_act_fwd_fused(const float* x, const float* gamma, const float* beta, const float* residual, float* running_mean,
                           float* running_var, float* mean, float* invstd, float* y, int B, int C, int HW, int act, float slope,
                           float eps, float momentum, void* ws, size_t ws_bytes, hipStream_t stream);
/* G BatchNorm(train)+activation calls on the G groups of B images of one (G*B, C, HW) tensor, one launch each way (the object
 * pathways of INIT_STAGE_G / D_NET64, model.py:395-407, 662-672: one BatchNorm call per object): group g uses its own
 * batch statistics (mean / invstd: G x C floats, group-major), the running statistics are updated G times in group order, d gamma /
 * d beta sum over the groups.  B*HW <= 4096 per group (mogan_bn_act_grouped_eligible), any HW >= 1. */
int mogan_bn_act_grouped_eligible(int G, int B, int C, int HW);
int mogan_bn_act_grouped_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* mean,
                             float* invstd, float* y, int G, int B, int C, int HW, int act, float slope, float eps, float momentum,
                             hipStream_t stream);
int mogan_bn_act_grouped_bwd(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             float* dx, float* dgamma, float* dbeta, int G, int B, int C, int HW, int act, float slope, int accumulate,
                             hipStream_t stream);
/* dy (B,Cy,HW) -> dx (B,C,HW), dgamma[C], dbeta[C] (accumulate != 0 adds into dgamma/dbeta).
 * The residual branch's gradient is dy itself. ws REQUIRED (mogan_bn_ws_bytes). */
int mogan_bn_act_bwd(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, float* dx, float* dgamma, float* dbeta, int B, int C, int HW, int act,
                     float slope, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream);
/* eval-mode BN / per-channel affine fused with an activation: y = act(x*scale[c] + shift[c]);
 * bwd: dx = dy * act'(.) * scale[c].  (frozen Inception BasicConv2d under CNN_ENCODER, model.py:227-242) */
int mogan_affine_act_fwd(const float* x, const float* scale, const float* shift, float* y, int B, int C, int HW,
                         int act, float slope, hipStream_t stream);
int mogan_affine_act_bwd(const float* x, const float* dy, const float* scale, const float* shift, float* dx, int B,
                         int C, int HW, int act, float slope, hipStream_t stream);

/* ---------------------------------------------------------------- elementwise
 * act in {RELU, LRELU, GLU (over channel dim: x (B,C,HW) -> y (B,C/2,HW)), TANH, SIGMOID}.
 * bwd takes the forward INPUT x (and recomputes).  model.py:93,328,373,470,599,627,659 */
int mogan_act_fwd(const float* x, float* y, int B, int C, int HW, int act, float slope, hipStream_t stream);
int mogan_act_bwd(const float* x, const float* dy, float* dx, int B, int C, int HW, int act, float slope,
                  hipStream_t stream);
/* y (rows,C,HW) += bias[C];  dbias[c] (+)= sum over rows,HW of dy */
int mogan_bias_add(float* y, const float* bias, int rows, int C, int HW, hipStream_t stream);
int mogan_bias_grad(const float* dy, float* dbias, int rows, int C, int HW, int accumulate, hipStream_t stream);
/* y = a + b (n elements); y = a*alpha */
int mogan_add(const float* a, const float* b, float* y, long long n, hipStream_t stream);
/* y = ((x_0 + x_1) + x_2) + ... over the G groups of n values of x (G*n values); backward: dx_g = dy for every group */
int mogan_group_sum(const float* x, float* y, long long n, int G, hipStream_t stream);
int mogan_group_bcast(const float* dy, float* dx, long long n, int G, hipStream_t stream);
int mogan_scale(const float* a, float alpha, float* y, long long n, hipStream_t stream);

/* softmax over L of x viewed as (outer, L, inner), y = softmax(scale*x) restricted to the first
 * lens[o*inner+i] entries (lens nullable = all L); masked entries get 0.  bwd: dx = scale*y*(dy - sum(y*dy)).
 * GlobalAttention.py:50,58 (func_attention, DAMSM). */
int mogan_softmax_fwd(const float* x, float* y, const int32_t* lens, long long outer, int L, long long inner,
                      float scale, hipStream_t stream);
int mogan_softmax_bwd(const float* y, const float* dy, float* dx, const int32_t* lens, long long outer, int L,
                      long long inner, float scale, hipStream_t stream);

/* ---------------------------------------------------------------- spatial transformer (object pathway)
 * y[b,c] = grid_sample(x[b,c], affine_grid(theta[b])) bilinear, zero padding (model.py:17-21).
 * align_corners: 0 = torch>=1.3 default, 1 = torch 0.4.1 semantics.
 * bwd scatters with fp32 atomics into dx, which the callee zero-fills first. */
int mogan_stn_fwd(const float* x, const float* theta, float* y, int B, int C, int Hin, int Win, int Hout, int Wout,
                  int align_corners, hipStream_t stream);
int mogan_stn_bwd(const float* dy, const float* theta, float* dx, int B, int C, int Hin, int Win, int Hout,
                  int Wout, int align_corners, hipStream_t stream);
/* The same with a shared / constant source (round 4; the object pathways' inputs without their materialised copies):
 *   xB       x holds xB images and output sample b reads image b % xB -- `stn(image, transf_matrices[:, idx], ...)` for every
 *            object idx from ONE copy of the image batch (model.py:663-665); dx then has xB images and collects all objects;
 *   x_plane  x is (xB, C): one value per (image, channel), constant over the Hin x Win plane -- the label vector the
 *            reference first .repeat()s over 16 x 16 (model.py:109-111); dx is (xB, C);
 *   theta_G  > 0: theta is stored (B / theta_G, theta_G, 2, 3) -- the loader's (image, object) order -- while the batch is
 *            object-major, sample b = g (B / theta_G) + b' using theta[b'][g] (the batched object loops); 0: theta[b].
 * Determinism: mogan_stn_bwd (like torch's grid_sample backward) scatters with fp32 atomicAdd, so the sum of the <= 4 output
 * pixels' contributions that meet in one source pixel depends on execution order; the shared-source forms add all OBJECTS'
 * contributions to that same sum (the reference adds G separately computed tensors in object order).  The result is
 * therefore reproducible to fp32 rounding of a sum of <= 4 G terms (measured run to run: <= 2 ulp of the largest term), not
 * bit for bit -- tests/test_kernels_gpu.py::test_stn_shared_source_gradient_is_order_independent_to_rounding pins that. */
int mogan_stn_fwd_ex(const float* x, const float* theta, float* y, int B, int C, int Hin, int Win, int Hout, int Wout,
                     int align_corners, int xB, int x_plane, int theta_G, hipStream_t stream);
int mogan_stn_bwd_ex(const float* dy, const float* theta, float* dx, int B, int C, int Hin, int Win, int Hout, int Wout,
                     int align_corners, int xB, int x_plane, int theta_G, hipStream_t stream);
/* ---------------------------------------------------------------- channel concat with broadcast sources (round 4)
 * dst (N, C_0 + ... + C_{nsrc-1}, HW) = the sources side by side along the channel axis -- torch.cat(..., 1) of model.py:
 * 400-401, 418, 457, 633-634, 666, 703 together with the .repeat() / per-object indexing that feeds it, in ONE launch.
 * Source i (C[i] channels) is addressed as
 *     value(n, c, hw) = src[i][(n % rows[i]) * sb[i] + (n / rows[i]) * sg[i] + c * (bcast[i] ? 1 : HW) + (bcast[i] ? 0 : hw)]
 *   plain (N, C, HW) tensor:                     rows = N,     sb = C HW, sg = 0, bcast = 0
 *   (N, C) code repeated over the plane:         rows = N,     sb = C,    sg = 0, bcast = 1
 *   (N/G, C, HW) tensor repeated for G objects:  rows = N / G, sb = C HW, sg = 0
 *   label[:, g] of a (B, G, C) tensor, batch n = g B + b (object-major): rows = B, sb = G C, sg = C  (x HW when bcast = 0)
 * mogan_concat_bwd: dsrc[i] (NULL = not wanted), in the source's own layout, = the sum of ddst over everything that read it
 * (its channel slice; summed over the plane when bcast; summed over the repeats when rows < N and sg = 0).  nsrc <= MOGAN_CAT_MAX. */
#define MOGAN_CAT_MAX 4
int mogan_concat_fwd(const void* const* src, const int* C, const int* rows, const long long* sb, const long long* sg,
                     const int* bcast, int nsrc, float* dst, int N, int HW, hipStream_t stream);
int mogan_concat_bwd(const float* ddst, void* const* dsrc, const int* C, const int* rows, const long long* sb,
                     const long long* sg, const int* bcast, int nsrc, int N, int HW, hipStream_t stream);
/* bbox (N,4)=(x,y,w,h) -> theta (N,2,3), theta_inv (N,2,3)   (miscc/utils.py:16-49) */
int mogan_bbox_to_theta(const float* bbox, float* theta, float* theta_inv, int N, hipStream_t stream);

/* ---------------------------------------------------------------- word attention over image regions
 * GlobalAttentionGeneral.forward core (GlobalAttention.py:96-121) after conv_context:
 * h (B,idf,Q), src (B,idf,T), mask (B,T) uint8 nullable -> wc (B,idf,Q), attn (B,T,Q).
 * mask_mode 0 = reference indexing (row b*Q+q is masked with mask[(b*Q+q) mod B]),
 *           1 = mask[b].   Limits: idf <= 128, T <= 32. */
int mogan_attn_fwd(const float* h, const float* src, const uint8_t* mask, float* wc, float* attn, int B, int idf,
                   int Q, int T, int mask_mode, hipStream_t stream);
/* -> dh (B,idf,Q) and dscore (B,T,Q) (gradient w.r.t. the pre-softmax scores).  dattn nullable.
 * dsrc is then two mogan_bmm calls: dsrc = h . dscore^T + dwc . attn^T. */
int mogan_attn_bwd(const float* src, const float* attn, const float* dwc, const float* dattn, float* dh,
                   float* dscore, int B, int idf, int Q, int T, hipStream_t stream);

/* ---------------------------------------------------------------- losses
 * BCE (mean) of probabilities p[n] against a constant target (nn.BCELoss with torch's log clamp at -100;
 * miscc/losses.py:158-168,198-201): loss[0] (+)= weight * mean(...); bwd: dp = gout[0]*weight/n * d/dp */
int mogan_bce_fwd(const float* p, float target, float weight, float* loss, int n, int accumulate,
                  hipStream_t stream);
int mogan_bce_bwd(const float* p, float target, float weight, const float* gout, float* dp, int n,
                  hipStream_t stream);
/* BCEWithLogits (mean) of raw logits x[n] against a constant target, torch's stable form
 * max(x,0) - x*t + log1p(exp(-|x|)) -- the StackGAN-family losses (code/coco/stackgan/miscc/utils.py:68-125,
 * code/clevr/miscc/utils.py:91-144, code/multi-mnist/miscc/utils.py:71-123).  bwd: dx = gout*w*(sigmoid(x)-t)/n */
int mogan_bce_logits_fwd(const float* x, float target, float weight, float* loss, int n, int accumulate,
                         hipStream_t stream);
int mogan_bce_logits_bwd(const float* x, float target, float weight, const float* gout, float* dx, int n,
                         hipStream_t stream);
/* KL_loss (miscc/losses.py:230-234): loss = -0.5*mean(1 + logvar - mu^2 - exp(logvar)) */
int mogan_kl_fwd(const float* mu, const float* logvar, float* loss, int n, hipStream_t stream);
int mogan_kl_bwd(const float* mu, const float* logvar, const float* gout, float* dmu, float* dlogvar, int n,
                 hipStream_t stream);

/* CA_NET.reparametrize (model.py:333-340): c = eps*exp(0.5*logvar) + mu; bwd: dmu = dc, dlogvar = dc*eps*0.5*exp(0.5*logvar) */
int mogan_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* c, int n, hipStream_t stream);
int mogan_reparam_bwd(const float* logvar, const float* eps, const float* dc, float* dmu, float* dlogvar, int n,
                      hipStream_t stream);

/* ---------------------------------------------------------------- DAMSM matching losses (generator step)
 * words_loss (miscc/losses.py:62-132 + GlobalAttention.py:31-69 func_attention, which the reference calls once per
 * caption in a python loop) for ALL (image b, caption i) pairs in one launch.
 *   ctx (B,C,S) region features, words (Bc,C,T) word embeddings, cap_lens (Bc) int32 valid words per caption.
 *   fwd -> sim (B,Bc) = gamma3 * log sum_t exp(gamma2 * cos(word_{i,t}, context_{b,i,t})), and for the backward pass
 *          a1 (B,Bc,S,T) softmax over the words, a2 (B,Bc,T,S) softmax over the regions (= the attention maps of
 *          losses.py:87-91), wc (B,C,Bc,T) weighted contexts, wt (C,Bc,T) the word embeddings in GEMM layout
 *          (nullable).  Limits: T <= 32, (C*T + T*(S+1) + 896) * 4 bytes of LDS <= 64 KB.
 *   bwd: dsim (B,Bc) -> dwc (B,C,Bc,T), dscore_t (B,Bc,T,S); the region-feature gradient is then two mogan_bmm calls:
 *          dctx[b] (C,S) = dwc[b] (C, Bc*T) . a2[b] (Bc*T, S)  +  wt (C, Bc*T) . dscore_t[b] (Bc*T, S).
 *        (image side only: the text encoder is frozen in the generator step, trainer.py:281-289).  S <= 512. */
int mogan_damsm_words_fwd(const float* ctx, const float* words, const int32_t* cap_lens, int B, int Bc, int C, int S, int T,
                          float gamma1, float gamma2, float gamma3, float* sim, float* a1, float* a2, float* wc, float* wt,
                          hipStream_t stream);
int mogan_damsm_words_bwd(const float* ctx, const float* words, const int32_t* cap_lens, const float* a1, const float* a2,
                          const float* wc, const float* dsim, int B, int Bc, int C, int S, int T, float gamma1,
                          float gamma2, float gamma3, float* dwc, float* dscore_t, hipStream_t stream);
/* The two cross-entropies over a square similarity matrix sim (R,R) (losses.py:45-58,116-130): rows against labels
 * (image -> caption) and columns against labels (caption -> image), entries with mask[r,q] != 0 (same class, nullable)
 * set to -inf first.  fwd -> prow, pcol (R,R) softmax probabilities, nll (2R) scratch, out2 = {loss0, loss1} (means).
 * bwd: g0, g1 (device scalars, gradients of loss0 / loss1; nullable = 0) -> dsim (R,R). */
int mogan_damsm_ce_fwd(const float* sim, const int64_t* labels, const uint8_t* mask, int R, int Q, float* prow,
                       float* pcol, float* nll, float* out2, hipStream_t stream);
int mogan_damsm_ce_bwd(const float* prow, const float* pcol, const int64_t* labels, const float* g0, const float* g1, int R,
                       int Q, float* dsim, hipStream_t stream);
/* sent_loss similarity (losses.py:36-44): sim[b,i] = gamma3 * <cnn_b, rnn_i> / max(|cnn_b| |rnn_i|, eps); bwd -> dcnn */
int mogan_damsm_sent_fwd(const float* cnn, const float* rnn, int B, int Bc, int C, float gamma3, float eps, float* sim,
                         hipStream_t stream);
int mogan_damsm_sent_bwd(const float* cnn, const float* rnn, const float* dsim, int B, int Bc, int C, float gamma3,
                         float eps, float* dcnn, hipStream_t stream);
/* out[0] = sum_k weights[k] * in[k][0] for n <= 8 device scalars (the loss sums of losses.py:169-174,203,221 and
 * trainer.py:330 in one launch); scalar_scale is its gradient: out[k][0] = weights[k] * g[0].  `in` / `out` / `weights`
 * are HOST arrays (read during the call). */
int mogan_scalar_sum(const float* const* in, const float* weights, int n, float* out, hipStream_t stream);
int mogan_scalar_scale(const float* g, const float* weights, int n, float* const* out, hipStream_t stream);

/* ---------------------------------------------------------------- pooling / resize (CNN_ENCODER trunk)
 * max_pool2d(k,s, no padding): fwd records the offset of the first maximum of each window in idx (uint8,
 * same shape as y; nullable), bwd gathers through it (ties -> first, like torch),
 * avg_pool2d(k,s,pad, count_include_pad), bilinear resize (align_corners=0) -- model.py:256,264,271,301 */
int mogan_maxpool_fwd(const float* x, float* y, uint8_t* idx, int planes, int H, int W, int k, int s,
                      hipStream_t stream);
int mogan_maxpool_bwd(const uint8_t* idx, const float* dy, float* dx, int planes, int H, int W, int k, int s,
                      hipStream_t stream);
int mogan_avgpool_fwd(const float* x, float* y, int planes, int H, int W, int k, int s, int pad, hipStream_t stream);
int mogan_avgpool_bwd(const float* dy, float* dx, int planes, int H, int W, int k, int s, int pad,
                      hipStream_t stream);
/* Channel-slice variants for the frozen encoder's explicit forward / backward (attngan/inception.py): y / dy are slices
 * of tensors with the given batch stride (elements; -1 = dense), the pooling gradients are added to dx when
 * accumulate != 0 after being zeroed where relu_of <= 0 (relu_of: dense, shaped like dx; nullable) -- the ReLU backward
 * of the layer that produced the pooled tensor. */
int mogan_maxpool_fwd_ex(const float* x, float* y, long long y_bstride, uint8_t* idx, int B, int C, int H, int W, int k,
                         int s, hipStream_t stream);
int mogan_maxpool_bwd_ex(const uint8_t* idx, const float* dy, long long dy_bstride, float* dx, const float* relu_of,
                         int accumulate, int B, int C, int H, int W, int k, int s, hipStream_t stream);
int mogan_avgpool_bwd_ex(const float* dy, float* dx, const float* relu_of, int accumulate, int planes, int H, int W, int k,
                         int s, int pad, hipStream_t stream);
/* dx (+)= dz where z > 0 (z: the ReLU's output) */
int mogan_relu_bwd(const float* z, const float* dz, float* dx, long long n, int accumulate, hipStream_t stream);
/* B rows of n contiguous floats between two batch-strided tensors */
int mogan_copy_strided(const float* src, long long src_bstride, float* dst, long long dst_bstride, int B, long long n,
                       hipStream_t stream);
int mogan_bilinear_fwd(const float* x, float* y, int planes, int H, int W, int OH, int OW, hipStream_t stream);
int mogan_bilinear_bwd(const float* dy, float* dx, int planes, int H, int W, int OH, int OW, hipStream_t stream);

/* ---------------------------------------------------------------- input pipeline (device side)
 * datasets.py:70-137 (get_imgs / crop_imgs) after the JPEG decode + resize to `ori` x `ori` (268): the host uploads u8
 * HWC images and the (column offset h1, row offset w1, flip) it drew per sample.
 * crop_flip: src (B,ori,ori,3) u8, params (B,3) int32 -> q (B,size,size,3) u8 (the crop, = ToPILImage of it) and
 *            out (B,3,size,size) f32 = ((u8/255) - 0.5) / 0.5 (ToTensor + Normalize).
 * resample:  Pillow's antialiased BILINEAR resize of the square u8 image `in` (B,S,S,3) to OS x OS (transforms.Resize),
 *            then ToTensor + Normalize -> out (B,3,OS,OS) f32.  bounds (OS,2) / kk (OS,ksize): Pillow's coefficient
 *            tables (precompute_coeffs + normalize_coeffs_8bpc, 22-bit fixed point; the same table serves both passes of
 *            a square resize), built by the host (attngan/feeder.py); tmp (B,S,OS,3) u8 scratch.  Bit-identical to PIL. */
int mogan_feed_crop_flip(const uint8_t* src, const int32_t* params, uint8_t* q, float* out, int B, int ori, int size,
                         hipStream_t stream);
int mogan_feed_resample(const uint8_t* in, uint8_t* tmp, float* out, const int32_t* bounds, const int32_t* kk, int ksize,
                        int B, int S, int OS, hipStream_t stream);

/* ---------------------------------------------------------------- optimizer
 * One fused Adam step over a flat fp32 bucket (trainer.py:137-148: lr 2e-4, betas (0.5,0.999), eps 1e-8),
 * optionally followed by the EMA of trainer.py:341-342: ema = ema_decay*ema + (1-ema_decay)*p (ema nullable).
 * `step` is the 1-based step count; if dev_state (3 floats on the device, zero-initialised) is given it
 * replaces `step`: the count lives in dev_state[0] and is incremented on the device, so the call can be
 * captured in a hipGraph and replayed.  eps_mode 0: denom = sqrt(v)/sqrt(1-b2^t) + eps (torch>=1.x);
 * 1: denom = sqrt(v) + eps with step size lr*sqrt(1-b2^t)/(1-b1^t) (torch 0.4.1).  grad_scale multiplies g
 * first (1/world_size after an RCCL sum all-reduce). */
int mogan_adam_step(float* p, const float* g, float* m, float* v, float* ema, long long n, float lr, float beta1,
                    float beta2, float eps, int step, float* dev_state, int eps_mode, float grad_scale,
                    float ema_decay, hipStream_t stream);

/* ---- text encoder (round 6, third session; csrc/mogan_lstm.hip): nn.Embedding + one-layer bidirectional nn.LSTM over packed
 * captions in eval mode, without gradients -- RNN_ENCODER.forward, code/coco/attngan/model.py:183-204 (pack_padded_sequence,
 * self.rnn, pad_packed_sequence, the transposes) -- as ONE launch instead of the 119 of the stock module (MIOpen).
 *   captions (B, T) int64 token ids (clamped to [0, V)), lens[B] on the HOST (sorted or not; 0 <= lens[i] <= Tmax <= T, Tmax <= 32),
 *   emb (V, E) (E % 4 == 0, E <= 320), per direction d = 0 forward / 1 reverse: w_ih[d] (4H, E), w_hh[d] (4H, H), b_ih[d], b_hh[d]
 *   (4H) in PyTorch's gate order i, f, g, o (16-byte aligned weights), H = 128, B <= 64; h0 / c0 (2, B, H) or NULL (zeros);
 *   words (B, 2H, Tmax): the hidden states, zero for t >= lens[b]; sent (B, 2H): the final states (forward | reverse). */
int mogan_lstm_encoder_fwd(const long long* captions, const int* lens, const float* emb, const float* const* w_ih,
                           const float* const* w_hh, const float* const* b_ih, const float* const* b_hh, const float* h0,
                           const float* c0, float* words, float* sent, int B, int T, int Tmax, int V, int E, int H,
                           hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MOGAN_HIP_H */
