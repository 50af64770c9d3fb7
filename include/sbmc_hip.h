/*
 * sbmc_hip.h -- C ABI of libsbmc_hip.so, the MI355X (gfx950) implementation of
 * SBMC's per-sample kernel-splatting operators.
 *
 * This is the drop-in boundary.  In the reference the boundary is the
 * Halide-generated extension module `sbmc.halide_ops` (setup.py:65-90), whose
 * pybind wrapper (halide_pytorch/halide_pytorch/extension.py:149-175) exposes,
 * per operator, `<op>_{cpu,cuda}_float32(inputs..., preallocated outputs...)`
 * in generator Input<>/Output<> declaration order.  Every function below
 * replaces one of those entry points (cited) or fuses the Python composition
 * that sits directly on top of them (sbmc/modules.py:376-473).
 *
 * Conventions (same contract as the reference wrapper):
 *   - all tensors are contiguous float32 (int32 where stated) DEVICE pointers
 *     in torch index order:
 *         data     [bs, c,  h, w]
 *         weights  [bs, kh, kw, h, w]   (kh, kw odd or even; pad = (k-1)/2)
 *         sum_w    [bs, h, w]
 *   - the caller allocates every output; the kernels write every element of
 *     every output (buffers may arrive uninitialised); inputs are read-only;
 *     outputs must not alias inputs;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *     Launches are asynchronous on it; no device synchronisation, allocation or
 *     global scratch inside the library => re-entrant from any host thread;
 *   - return value: 0 on success; SBMC_HIP_EINVAL for bad arguments; otherwise
 *     the hipError_t of the failed launch.  Nothing throws across this ABI.
 */
#ifndef SBMC_HIP_H
#define SBMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define SBMC_API __attribute__((visibility("default")))
#else
#define SBMC_API
#endif

#define SBMC_HIP_ABI_VERSION 8
#define SBMC_HIP_EINVAL (-1)
/* largest channel count the fused/plain kernels take in one call */
#define SBMC_HIP_MAX_CHANNELS 8

/* Returns SBMC_HIP_ABI_VERSION. */
SBMC_API int sbmc_hip_abi_version(void);

/* Human-readable text for a non-zero return code (static storage). */
SBMC_API const char *sbmc_hip_strerror(int code);

/*
 * scatter2gather -- replaces `scatter2gather_cuda_float32(weights, output)`
 * (reference: src/scatter2gather.cpp:29-52, called at sbmc/functions.py:56-59
 * and, as its own adjoint, :67-70).
 *   output[n,dy,dx,y,x] = Wz[n, kh-1-dy, kw-1-dx, y+dy-ph, x+dx-pw]   (z: 0 outside)
 */
SBMC_API int sbmc_scatter2gather_f32(const float *weights, float *output,
                            int bs, int h, int w, int kh, int kw,
                            void *stream);

/*
 * kernel_weighting forward -- replaces
 * `kernel_weighting_cuda_float32(data, weights, output, sum_w)`
 * (reference: src/kernel_weighting.cpp:28-64, called at sbmc/functions.py:95-98).
 *   output[n,c,y,x] = sum_{ry,rx} W[n,ry,rx,y,x] * Dz[n,c,y+ry-ph,x+rx-pw]
 *   sum_w[n,y,x]    = sum_{ry,rx} W[n,ry,rx,y,x]          (not boundary-masked)
 * c is unrestricted (processed in groups of SBMC_HIP_MAX_CHANNELS).
 */
SBMC_API int sbmc_kernel_weighting_fwd_f32(const float *data, const float *weights,
                                  float *output, float *sum_w,
                                  int bs, int c, int h, int w, int kh, int kw,
                                  void *stream);

/*
 * kernel_weighting backward -- replaces
 * `kernel_weighting_grad_cuda_float32(data, weights, sum_w, d_output, d_sum_w,
 *                                     d_data, d_weights)`
 * (reference: src/kernel_weighting.cpp:68-124, called at sbmc/functions.py:109-114).
 *   d_data[n,c,y,x]       = sum_{ry,rx} Wz[n,kh-1-ry,kw-1-rx,y+ry-ph,x+rx-pw]
 *                                       * dOz[n,c,y+ry-ph,x+rx-pw]
 *   d_weights[n,dy,dx,y,x] = d_sum_w[n,y,x] + sum_c Dz[n,c,y+dy-ph,x+dx-pw] * dO[n,c,y,x]
 * `sum_w` is accepted for signature parity and ignored (as in the reference).
 */
SBMC_API int sbmc_kernel_weighting_bwd_f32(const float *data, const float *weights,
                                  const float *sum_w, const float *d_output,
                                  const float *d_sum_w, float *d_data,
                                  float *d_weights,
                                  int bs, int c, int h, int w, int kh, int kw,
                                  void *stream);

/*
 * The same three operators with half-precision storage -- what the reference's naming scheme
 * (setup.py:65-84: `<op>_{cpu,cuda}_<dtype>`) would call `scatter2gather_cuda_float16`,
 * `kernel_weighting_cuda_float16`, `kernel_weighting_grad_cuda_float16` (the
 * reference only instantiates float32).  EVERY tensor of a call is IEEE half (torch.float16), same
 * shapes and semantics; products and sums are formed in fp32 and rounded once on the store.
 * scatter2gather_f16 is a pure permutation: bit exact.
 */
SBMC_API int sbmc_scatter2gather_f16(const void *weights, void *output,
                            int bs, int h, int w, int kh, int kw, void *stream);
SBMC_API int sbmc_kernel_weighting_fwd_f16(const void *data, const void *weights,
                                  void *output, void *sum_w,
                                  int bs, int c, int h, int w, int kh, int kw, void *stream);
SBMC_API int sbmc_kernel_weighting_bwd_f16(const void *data, const void *weights,
                                  const void *sum_w, const void *d_output,
                                  const void *d_sum_w, void *d_data, void *d_weights,
                                  int bs, int c, int h, int w, int kh, int kw, void *stream);

/*
 * 1 if sbmc_splat_update_{fwd,bwd}_f32 accept this (channels, kernel size): k odd,
 * 1 <= c <= SBMC_HIP_MAX_CHANNELS and the LDS halo tiles fit; 0 otherwise (the
 * caller then composes scatter2gather + kernel_weighting instead).
 */
SBMC_API int sbmc_splat_update_supported(int c, int k);

/*
 * Fused progressive splat update, forward -- one call replaces the whole body of
 * `ProgressiveKernelApply.forward` for splat=True (sbmc/modules.py:422-471):
 * Scatter2Gather -> max over taps -> running-max merge -> sub_/exp_ ->
 * KernelWeighting -> running sums, in ONE pass over the [k*k,h,w] logits.
 *
 *   kernels   [bs, k*k, h, w]  sample-centred (splat) logits, k odd
 *   data      [bs, c, h, w]    sample radiance, 1 <= c <= SBMC_HIP_MAX_CHANNELS
 *   sum_r_in  [bs, c, h, w] | sum_w_in [bs, h, w] | max_w_in [bs, h, w]
 *             running state, or all three NULL for the initialisation call
 *             (modules.py:431-447); a mix of NULL / non-NULL is SBMC_HIP_EINVAL
 *   sum_r_out, sum_w_out, max_w_out   updated state (same shapes)
 *   kmax_out  [bs, h, w]       max over this sample's gather taps (modules.py:429)
 *   atap_out  [bs, h, w] int32 gather tap index dy*k+dx of the first tap attaining kmax
 *             (kmax_out / atap_out are saved for backward: autograd routes the
 *             gradient of the max to that tap)
 * With g[t] the gather-layout logits of destination pixel q (0 where the source
 * pixel lies outside the image) and M = max(kmax, max_w_in):
 *   sum_r_out = sum_r_in * exp(max_w_in - M) + sum_t exp(g[t]-M) * Dz[src(t)]
 *   sum_w_out = sum_w_in * exp(max_w_in - M) + sum_t exp(g[t]-M)
 */
SBMC_API int sbmc_splat_update_fwd_f32(const float *data, const float *kernels,
                              const float *sum_r_in, const float *sum_w_in,
                              const float *max_w_in,
                              float *sum_r_out, float *sum_w_out,
                              float *max_w_out, float *kmax_out,
                              int32_t *atap_out,
                              int bs, int c, int h, int w, int k,
                              void *stream);

/* Size in bytes of the workspace sbmc_splat_update_bwd_f32 needs for these dimensions. */
SBMC_API size_t sbmc_splat_update_bwd_scratch_bytes(int bs, int c, int h, int w, int k);

/*
 * Fused progressive splat update, backward -- the adjoint of the call above,
 * i.e. what torch autograd computes through modules.py:422-471 +
 * functions.py:62-71,102-115 (KernelWeighting.backward, exp/sub/max backward,
 * Scatter2Gather.backward), in one read of the logits and one write of their
 * gradient.
 *
 *   inputs saved from forward: data, kernels, sum_r_in/sum_w_in/max_w_in (or all
 *       NULL), sum_r_out, sum_w_out, max_w_out, kmax, atap
 *   upstream gradients: d_sum_r_out [bs,c,h,w], d_sum_w_out [bs,h,w],
 *       d_max_w_out [bs,h,w]   (all required; pass zeros where unused)
 *   outputs: d_data [bs,c,h,w], d_kernels [bs,k*k,h,w] (splat layout), and --
 *       unless this was the initialisation call -- d_sum_r_in, d_sum_w_in,
 *       d_max_w_in (pass NULL for all three on the initialisation call)
 *   scratch: caller-provided device workspace of
 *       sbmc_splat_update_bwd_scratch_bytes(bs, c, h, w, k) bytes (the library holds
 *       no global state and never allocates); contents undefined on return.
 */
SBMC_API int sbmc_splat_update_bwd_f32(const float *data, const float *kernels,
                              const float *sum_r_in, const float *sum_w_in,
                              const float *max_w_in,
                              const float *sum_r_out, const float *sum_w_out,
                              const float *max_w_out, const float *kmax,
                              const int32_t *atap,
                              const float *d_sum_r_out, const float *d_sum_w_out,
                              const float *d_max_w_out,
                              float *d_data, float *d_kernels,
                              float *d_sum_r_in, float *d_sum_w_in,
                              float *d_max_w_in, float *scratch,
                              int bs, int c, int h, int w, int k,
                              void *stream);

/*
 * Gather-kernel variant (ProgressiveKernelApply(splat=False), the reference's `--gather`
 * ablation, sbmc/modules.py:422-471 without the Scatter2Gather step): `kernels` are
 * pixel-centred, tap (dy,dx) of pixel (y,x) weighs data[y+dy-p, x+dx-p]; every tap counts in the
 * running softmax (only the data outside the image is zero).  Same arguments, outputs and
 * saved tensors as sbmc_splat_update_{fwd,bwd}_f32.  fp32; only where
 * sbmc_gather_update_supported(c, k, h, w) == 1 (otherwise the caller composes
 * kernel_weighting with torch ops, as the reference does).
 */
SBMC_API int sbmc_gather_update_supported(int c, int k, int h, int w);

SBMC_API int sbmc_gather_update_fwd_f32(const float *data, const float *kernels,
                               const float *sum_r_in, const float *sum_w_in, const float *max_w_in,
                               float *sum_r_out, float *sum_w_out, float *max_w_out,
                               float *kmax_out, int32_t *atap_out,
                               int bs, int c, int h, int w, int k, void *stream);

SBMC_API int sbmc_gather_update_bwd_f32(const float *data, const float *kernels,
                               const float *sum_r_in, const float *sum_w_in, const float *max_w_in,
                               const float *sum_r_out, const float *sum_w_out, const float *max_w_out,
                               const float *kmax, const int32_t *atap,
                               const float *d_sum_r_out, const float *d_sum_w_out, const float *d_max_w_out,
                               float *d_data, float *d_kernels,
                               float *d_sum_r_in, float *d_sum_w_in, float *d_max_w_in, float *scratch,
                               int bs, int c, int h, int w, int k, void *stream);

/*
 * All samples of a frame at once.  The running state (sum_r, sum_w, max_w) is an
 * associative log-sum-exp monoid, so S progressive updates (the sample loop of
 * Multisteps.forward, reference sbmc/models.py:195-209) need not be S dependent launches:
 *   1. sbmc_splat_update_fwd_f32 with batch = bs*S and NULL incoming state reduces every
 *      sample on its own (part_r, part_w, part_m = its sum_r_out, sum_w_out, max_w_out);
 *   2. sbmc_splat_merge_fwd_f32 folds the S partial states per pixel, in sample order, and
 *      keeps the running state after each sample (run_*) for the backward;
 *   3. sbmc_splat_all_bwd_f32 = per-pixel reverse chain (the per-step state adjoint of the
 *      progressive update, applied S times) + ONE launch producing every d_kernels/d_data.
 * Values equal S chained sbmc_splat_update_* calls up to fp32 rounding.
 * Layouts: part_r/run_r/data/d_data [bs,S,c,h,w]; part_w/part_m/run_w/run_m/atap [bs,S,h,w];
 * kernels/d_kernels [bs,S,k*k,h,w]; final state and its gradients as in the per-sample calls;
 * scratch: sbmc_splat_update_bwd_scratch_bytes(bs*S, c, h, w, k) bytes.
 * Only where sbmc_splat_all_supported(c, k, h, w) == 1 (the strip kernels: k = 21 with
 * c <= 4, or odd k in 3..19 with c = 3).
 */
SBMC_API int sbmc_splat_all_supported(int c, int k, int h, int w);

SBMC_API int sbmc_splat_merge_fwd_f32(const float *part_r, const float *part_w, const float *part_m,
                             float *sum_r, float *sum_w, float *max_w,
                             float *run_r, float *run_w, float *run_m,
                             int bs, int s, int c, int h, int w, void *stream);

SBMC_API int sbmc_splat_all_bwd_f32(const float *data, const float *kernels,
                           const float *part_m, const int32_t *atap,
                           const float *run_r, const float *run_w, const float *run_m,
                           const float *d_sum_r, const float *d_sum_w, const float *d_max_w,
                           float *d_data, float *d_kernels, float *scratch,
                           int bs, int s, int c, int h, int w, int k, void *stream);

/* ABI 6: sbmc_splat_all_bwd_f32 / sbmc_splat_slab_bwd_f32 (top = bot = 0: the whole frame) that also raise `bound` -- a
 * device word the caller zeroed -- to the bit pattern of an upper bound of |d_kernels|: with every weight
 * exp(S - M) <= 1, |dS| <= |dW| + |d_kmax| + max|data| * sum_c |dR_c| per destination, quantities the per-pixel chain
 * holds; dmax: a device word with the bit pattern of max |data|.  The consumer of d_kernels (the wide 1 x 1 layer's
 * backward, sbmc_pointwise_wide_bwd_f32) takes its power-of-two scale from it without a pass over 1.6 GB per sample. */
SBMC_API int sbmc_splat_all_bwd_bound_f32(const float *data, const float *kernels,
                                 const float *part_m, const int32_t *atap,
                                 const float *run_r, const float *run_w, const float *run_m,
                                 const float *d_sum_r, const float *d_sum_w, const float *d_max_w,
                                 float *d_data, float *d_kernels, float *scratch,
                                 const unsigned *dmax, unsigned *bound,
                                 int bs, int s, int c, int h, int w, int k, int top, int bot, void *stream);

/*
 * Row-slab form of the all-samples splat: ONE frame sharded along H over several GPUs (new
 * functionality; the reference's closest analogue is the overlapped tiling
 * of scripts/denoise.py:54-93).  A rank holds the h rows of its slab -- radiance and logits of its
 * own samples only -- and splats them into a destination slab of hd = top + h + bot rows:
 * destination row r is source row r - top, `top` / `bot` (0 <= . <= (k-1)/2) are the overhang rows
 * that belong to the neighbouring ranks.  Source rows beyond a slab edge:
 *   zero_top / zero_bot = 1  the edge is the image edge: the taps beyond it are the zero-filled
 *                            taps of Scatter2Gather (logit 0, data 0 -- src/scatter2gather.cpp:34-47),
 *                            counted in the running softmax exactly as in the whole-frame calls;
 *   zero_top / zero_bot = 0  the rows belong to a neighbouring slab, whose rank accounts for them:
 *                            they contribute nothing here.
 * Every destination pixel's state is then the log-sum-exp partial over this rank's sources; the
 * ranks exchange the overhang rows and merge them with the rule of sbmc/modules.py:450-471
 * (sbmc_amd/dist.py), which reproduces the whole-frame state up to fp32 rounding.
 *   1. sbmc_splat_slab_fwd_f32: n = bs*S images; part_r [n,c,hd,w], part_w / part_m / atap [n,hd,w]
 *      (part_m = the per-sample max, as kmax in the whole-frame call);
 *   2. sbmc_splat_merge_fwd_f32 with h := hd folds the S partial states (unchanged);
 *   3. sbmc_splat_slab_bwd_f32: as sbmc_splat_all_bwd_f32; run_*, part_m, atap and the upstream
 *      gradients live on the hd destination rows, data / kernels / d_data / d_kernels on the h
 *      source rows; scratch: sbmc_splat_update_bwd_scratch_bytes(bs*S, c, hd, w, k) bytes.
 * Whole frame == (top, bot, zero_top, zero_bot) = (0, 0, 1, 1).  Strip kernels only:
 * sbmc_splat_slab_supported(c, k, h, w, top, bot) == 1, else SBMC_HIP_EINVAL (the caller then
 * halo-pads the inputs and uses the whole-frame calls on the padded slab).
 */
SBMC_API int sbmc_splat_slab_supported(int c, int k, int h, int w, int top, int bot);

SBMC_API int sbmc_splat_slab_fwd_f32(const float *data, const float *kernels,
                            float *part_r, float *part_w, float *part_m, int32_t *atap,
                            int n, int c, int h, int w, int k,
                            int top, int bot, int zero_top, int zero_bot, void *stream);

SBMC_API int sbmc_splat_slab_bwd_f32(const float *data, const float *kernels,
                            const float *part_m, const int32_t *atap,
                            const float *run_r, const float *run_w, const float *run_m,
                            const float *d_sum_r, const float *d_sum_w, const float *d_max_w,
                            float *d_data, float *d_kernels, float *scratch,
                            int bs, int s, int c, int h, int w, int k, int top, int bot, void *stream);

/* the same with half logits / logit gradients (see the fp16 block below) */
SBMC_API int sbmc_splat_slab_fwd_f16(const float *data, const void *kernels,
                            float *part_r, float *part_w, float *part_m, int32_t *atap,
                            int n, int c, int h, int w, int k,
                            int top, int bot, int zero_top, int zero_bot, void *stream);

SBMC_API int sbmc_splat_slab_bwd_f16(const float *data, const void *kernels,
                            const float *part_m, const int32_t *atap,
                            const float *run_r, const float *run_w, const float *run_m,
                            const float *d_sum_r, const float *d_sum_w, const float *d_max_w,
                            float *d_data, void *d_kernels, float *scratch,
                            int bs, int s, int c, int h, int w, int k, int top, int bot, void *stream);

/*
 * fp16 logits ("fp16 activations", BASELINE.json configs[4]).  Same calls as
 * the _f32 ones except that `kernels` and `d_kernels` are IEEE half tensors (torch.float16);
 * radiance, running state, every accumulation and every other gradient stay fp32.  Halves the
 * bytes of the two big streams.  Only where sbmc_splat_f16_supported(c,k,h,w) == 1,
 * SBMC_HIP_EINVAL otherwise (the caller up-casts and uses the _f32 entry points).
 */
/* 1 if the *_f16 entry points accept this (channels, kernel size, frame): k = 21, c <= 4. */
SBMC_API int sbmc_splat_f16_supported(int c, int k, int h, int w);

SBMC_API int sbmc_splat_update_fwd_f16(const float *data, const void *kernels,
                              const float *sum_r_in, const float *sum_w_in, const float *max_w_in,
                              float *sum_r_out, float *sum_w_out, float *max_w_out,
                              float *kmax_out, int32_t *atap_out,
                              int bs, int c, int h, int w, int k, void *stream);

SBMC_API int sbmc_splat_update_bwd_f16(const float *data, const void *kernels,
                              const float *sum_r_in, const float *sum_w_in, const float *max_w_in,
                              const float *sum_r_out, const float *sum_w_out, const float *max_w_out,
                              const float *kmax, const int32_t *atap,
                              const float *d_sum_r_out, const float *d_sum_w_out, const float *d_max_w_out,
                              float *d_data, void *d_kernels,
                              float *d_sum_r_in, float *d_sum_w_in, float *d_max_w_in, float *scratch,
                              int bs, int c, int h, int w, int k, void *stream);

SBMC_API int sbmc_splat_all_bwd_f16(const float *data, const void *kernels,
                           const float *part_m, const int32_t *atap,
                           const float *run_r, const float *run_w, const float *run_m,
                           const float *d_sum_r, const float *d_sum_w, const float *d_max_w,
                           float *d_data, void *d_kernels, float *scratch,
                           int bs, int s, int c, int h, int w, int k, void *stream);

/*
 * Fused bias + activation around the 1x1 convolutions of the per-sample ConvChains (reference:
 * nn.Conv2d(1x1) + ReLU / LeakyReLU, sbmc/modules.py:154-175 as used at sbmc/models.py:79-102),
 * whose matrix product runs as a batched GEMM on the planar [b, c, hw] activations.
 *   act: 0 linear, 1 relu, 2 leaky_relu(slope)
 *   fwd (in place):  y = act(y + bias[c])
 *   bwd:             gx = gy * act'(y)   (y = the forward OUTPUT; gx may alias gy -- for act = 0 the
 *                    aliased call only computes the sums),
 *                    partial[b, c, j] = sum of gx over chunk j of plane (b, c),
 *                    j < sbmc_bias_act_chunks(b, c, hw); the bias gradient is the sum of the
 *                    partials over b and j (done by the caller: no atomics, deterministic)
 * hw must be a multiple of 4 and the tensors 16-byte aligned (SBMC_HIP_EINVAL otherwise).
 */
SBMC_API int sbmc_bias_act_chunks(int b, int c, long hw);
SBMC_API int sbmc_bias_act_fwd_f32(float *y, const float *bias, int b, int c, long hw, int act,
                          float slope, void *stream);
SBMC_API int sbmc_bias_act_bwd_f32(const float *gy, const float *y, float *gx, float *partial,
                          int b, int c, long hw, int act, float slope, void *stream);

/*
 * The same for a chain whose first 1x1 layer sees [per-sample features ; per-pixel context]
 * (reference: th.cat([f, propagated], 1) for every sample, sbmc/models.py:147-153,171-177,
 * 196-199).  The layer is linear, so the context term t = W_c ctx is the same for all s samples
 * of a pixel; it is computed once per pixel by the caller and added here with bias + activation:
 *   fwd (in place):  y[b,s,c,p] = act(y[b,s,c,p] + t[b,c,p] + bias[c])
 *   bwd:             gx = gy * act'(y);  gt[b,c,p] = sum_s gx  (t_per_pixel = 1 only);
 *                    partial[b, c, j] = sum_{s, p in chunk j} gx, j < sbmc_bias_act_chunks(b, c, hw)
 *                    (bias gradient = sum over b, j; for t_per_pixel = 0, gt[b,c] = sum over j)
 * y, gy, gx: [b*s, c, hw];  t, gt: [b, c, hw] (t_per_pixel = 1) or [b, c] (t_per_pixel = 0: the
 * context is constant over the image, e.g. the global features of the first step).
 */
SBMC_API int sbmc_ctx_act_fwd_f32(float *y, const float *t, const float *bias, int b, int s, int c, long hw,
                         int t_per_pixel, int act, float slope, void *stream);
SBMC_API int sbmc_ctx_act_bwd_f32(const float *gy, const float *y, float *gx, float *gt, float *partial,
                         int b, int s, int c, long hw, int t_per_pixel, int act, float slope,
                         void *stream);

/* ---- a whole 1x1 convolution layer in one pass (fp32 MFMA) -------------------------------------
 * The per-sample layers of the reference's ConvChains (sbmc/modules.py:154-175 with ksize = 1;
 * sbmc/models.py:79-102) on planar activations:
 *     y[b, co, p] = act( sum_ci w[co, ci] * x[b, ci, p] + bias[co] + t )
 * x: [b, cin, hw], w: [cout, cin] (the convolution weight), y: [b, cout, hw]; cin <= 128.
 * t (the context half of a chain's first layer, see ctx_act above): t_mode 0 = none,
 * 1 = [b/s, cout] (constant over the image), 2 = [b/s, cout, hw]; batch element b uses t[b / s].
 * act / slope as for bias_act.  y must not alias x.
 */
SBMC_API int sbmc_pointwise_supported(int cin, int cout, long hw);
SBMC_API int sbmc_pointwise_fwd_f32(const float *x, const float *w, const float *bias, const float *t,
                           float *y, int b, int s, int cin, int cout, long hw, int t_mode, int act,
                           float slope, void *stream);
/* The same forward with half-precision STORAGE ("fp16 activations", BASELINE configs[4]): y is
 * _Float16, x is _Float16 (x_is_half = 1) or float (the network's fp32 inputs); w, bias, t stay
 * float and all arithmetic is fp32 (fp32 MFMA).  Backward: sbmc_pointwise_bwd_f16 below. */
SBMC_API int sbmc_pointwise_fwd_f16(const void *x, int x_is_half, const float *w, const float *bias,
                           const float *t, void *y, int b, int s, int cin, int cout, long hw,
                           int t_mode, int act, float slope, void *stream);
/* Backward of the same layer in one pass over gy, y (the forward output) and x; cout <= 128:
 *     gz = gy * act'(y);   gx[b] = w^T gz[b]   (gx may be NULL: not computed);
 *     gw_partial[g] = this workgroup's share of sum_b gz[b] x[b]^T     [groups, cout, cin]
 *     gb_partial[g] = its share of the row sums of gz                  [groups, nb, cout]
 *                     (nb = b / s for t_mode 1 -- the per-image context gradient is the sum over g --
 *                      else 1; the bias gradient is the sum over g and nb)
 *     gt[b / s]     = sum over the s samples of a pixel of gz          (t_mode 2 only) [b / s, cout, hw]
 * gmean (may be NULL; not with t_mode 2): [b / s_mean, cout, hw], the gradient of the mean of y over
 * groups of s_mean consecutive batch elements (the per-pixel mean over samples that feeds the U-net):
 * the kernel then works on gy[b] + gmean[b / s_mean] / s_mean, which saves the pass that would add them.
 * groups = sbmc_pointwise_bwd_groups(b, s, t_mode, hw) -- with gmean and t_mode 0:
 * sbmc_pointwise_bwd_groups(b, s_mean, 1, hw); the caller adds the partials up (no atomics).
 */
SBMC_API int sbmc_pointwise_bwd_supported(int cin, int cout, long hw);
SBMC_API int sbmc_pointwise_bwd_groups(int b, int s, int t_mode, long hw);
/* Weight and bias gradient of a WIDE linear 1x1 layer, 128 < cout <= 512 (the 441-channel kernel regressor output,
 * reference sbmc/models.py:98-102; the fused backward above takes cout <= 128): gw[co][k] = sum_{b, px} gz[b][co][px]
 * x[b][k][px] at fp32 accuracy on the bf16 matrix pipe, gbias[co] = sum gz, in ONE pass over gz and x.
 * gw_partial [groups, cout, cin] and gb_partial [groups, cout] (groups = ..._groups(b, hw)) are written completely;
 * the caller adds the groups up.  cin <= 128, hw % 4 == 0. */
SBMC_API int sbmc_pointwise_gw_wide_supported(int cin, int cout, long hw);
SBMC_API int sbmc_pointwise_gw_wide_groups(int b, long hw);
SBMC_API int sbmc_pointwise_gw_wide_f32(const float *gz, const float *x, float *gw_partial, float *gb_partial, int b,
                               int cin, int cout, long hw, void *stream);
/* ... with gz and x _Float16 (fp16 activations); the partial sums stay fp32 */
SBMC_API int sbmc_pointwise_gw_wide_f16(const void *gz, const void *x, float *gw_partial, float *gb_partial, int b,
                               int cin, int cout, long hw, void *stream);
SBMC_API int sbmc_pointwise_bwd_f32(const float *gy, const float *y, const float *x, const float *w,
                           float *gx, float *gw_partial, float *gb_partial, float *gt,
                           const float *gmean, int s_mean, int b, int s, int cin, int cout, long hw,
                           int t_mode, int act, float slope, void *stream);

/* Forward and backward with SIGN BITS instead of the saved output.  The forward also writes, when signs is not
 * NULL, one bit per output element -- pre-activation > 0 -- as [b, cout, ceil(hw / 32)] 32-bit words (bit i of
 * word j of a row = pixel 32 j + i); the backward's activation adjoint reads those instead of y: 1/32 of the
 * bytes.  (The fp32 forward runs on the bf16 matrix pipe at fp32 accuracy: every fp32 operand is the exact sum
 * of three bf16 values and six of the nine partial products are accumulated -- csrc/pointwise.hip.) */
SBMC_API int sbmc_pointwise_fwd_signs_f32(const float *x, const float *w, const float *bias, const float *t,
                                 float *y, unsigned *signs, int b, int s, int cin, int cout, long hw,
                                 int t_mode, int act, float slope, void *stream);
/* The same forward that also returns ymean [b / s_mean, cout, hw], the mean of y over groups of s_mean consecutive
 * batch elements (the per-pixel mean over samples that feeds the U-net, reference sbmc/models.py:179): the kernel
 * walks the samples of a pixel tile one after the other and accumulates in LDS, so y is not read back (cout <= 128;
 * with t_mode != 0: s_mean == s).  signs may be NULL. */
SBMC_API int sbmc_pointwise_fwd_mean_f32(const float *x, const float *w, const float *bias, const float *t, float *y,
                                unsigned *signs, float *ymean, int s_mean, int b, int s, int cin, int cout,
                                long hw, int t_mode, int act, float slope, void *stream);
/* ABI 6 -- the fp32 layer with MAGNITUDE WORDS (the scheme of the 3 x 3 kernels, sbmc_conv3x3_*: a pass leaves the bit
 * pattern of the largest magnitude it wrote in a device word, the consumer takes its power-of-two scale from it).
 *   xmax  a device word holding the bit pattern of a float >= max |x|, or NULL.  Not NULL: the products run in the 3 x 3
 *         kernels' number format -- two f16 planes under the power-of-two scale, three of the four partial products
 *         (<= 2^-22 |x w| per term dropped), fp32 accumulation -- at half the matrix-pipe work of the three-bf16-plane
 *         form (NULL: that form, as sbmc_pointwise_fwd_signs_f32).  A word SMALLER than max |x| is the caller's error
 *         (values beyond it overflow their f16 plane).
 *   amax  a device word the caller zeroed (or holding a lower bound), raised to the bit pattern of max |y|; or NULL.
 *   signs, ymean / s_mean: as above, or NULL.
 * No reference counterpart (the reference multiplies in fp32: sbmc/modules.py:154-175 through cuDNN). */
SBMC_API int sbmc_pointwise_fwd_scaled_f32(const float *x, const float *w, const float *bias, const float *t, float *y,
                                  unsigned *signs, float *ymean, int s_mean, const unsigned *xmax, unsigned *amax,
                                  int b, int s, int cin, int cout, long hw, int t_mode, int act, float slope,
                                  void *stream);
/* The backward with magnitude words: gmax / xmax (and gmmax when gmean is given) -- device words holding the bit patterns
 * of floats >= max |gy| / max |x| / max |gmean| -- not NULL: both products (gw += gz x^T, gx = w^T gz) in the two-f16-plane
 * form; both NULL: the three-bf16-plane form.  gxmax (needs gx; or NULL): a zeroed word raised to the bit pattern of
 * max |gx| -- the scale of the backward of the layer before (the three-plane form keeps it only without a context or mean
 * gradient: SBMC_HIP_EINVAL otherwise).  signs: the forward's sign bits (ignored when act == 0). */
SBMC_API int sbmc_pointwise_bwd_scaled_f32(const float *gy, const unsigned *signs, const float *x, const float *w,
                                  float *gx, float *gw_partial, float *gb_partial, float *gt, const float *gmean,
                                  int s_mean, const unsigned *gmax, const unsigned *gmmax, const unsigned *xmax,
                                  unsigned *gxmax, int b, int s, int cin, int cout, long hw, int t_mode, int act,
                                  float slope, void *stream);
/* The WHOLE backward of a wide linear layer (128 < cout <= 512: the 441-channel logits, reference sbmc/models.py:98-102)
 * in one pass over the logit gradient gz [b, cout, hw]: gx [b, cin, hw] = w^T gz, and the partial sums of gw / gbias as
 * sbmc_pointwise_gw_wide_f32 ([groups, cout, cin], [groups, cout], groups = sbmc_pointwise_gw_wide_groups(b, hw)), in
 * the two-f16-plane form: gmax / xmax are device words holding the bit patterns of floats >= max |gz| / max |x| (gmax
 * may be a loose bound: sbmc_splat_all_bwd_bound_f32); gxmax (or NULL): a zeroed word raised to max |gx|.
 * ws: sbmc_pointwise_wide_bwd_ws_bytes() bytes of device scratch, 16-byte aligned (the weights' prepared planes). */
SBMC_API size_t sbmc_pointwise_wide_bwd_ws_bytes(void);
SBMC_API int sbmc_pointwise_wide_bwd_f32(const float *gz, const float *x, const float *w, float *gx, float *gw_partial,
                                float *gb_partial, void *ws, const unsigned *gmax, const unsigned *xmax,
                                unsigned *gxmax, int b, int cin, int cout, long hw, void *stream);
/* ABI 8 -- a CHAIN of 2 or 3 per-sample 1 x 1 layers in ONE pass (csrc/pointwise_chain.hip): the reference's per-sample
 * embeddings and the first two layers of its kernel regressor (sbmc/modules.py:154-175 as built at sbmc/models.py:79-102
 * and run at :147-153, 171-177, 196-199), y_l = act_l(w_l y_{l-1} + bias_l (+ t for l = 0)), y_{-1} = x [b, cin, hw].
 * A 64-pixel tile stays in LDS through all layers: x is read once, an intermediate y_l is WRITTEN only where y[l] is not
 * NULL (training: the backward's input) and never read back.
 *   w, bias, y, signs, amax   arrays of nl device pointers (host arrays; signs / amax may be NULL, and so may their
 *         entries and y[l] for l < nl - 1).  signs[l] (needs y[l]): one bit per output, value > 0, [b, cout_l, ceil(hw / 32)]
 *         words as sbmc_pointwise_fwd_signs_f32; amax[l]: a zeroed device word raised to the bit pattern of max |y_l|.
 *   t / t_mode   the FIRST layer's context term as in sbmc_pointwise_fwd_f32 (s images share one); tmax (t_mode != 0): a
 *         device word holding the bit pattern of a float >= max |t| (any bound: it only enters the scale below).
 *   ymean   NULL or [b / s, cout_last, hw]: the mean of the last layer's output over groups of s consecutive images.
 *   cout, act, slope   per layer (host arrays); cin, cout_l <= 128.
 * Arithmetic: two f16 planes per operand under a power-of-two scale, three of the four partial products (<= 2^-22 per
 * term dropped), fp32 accumulation -- the 3 x 3 kernels' format -- with the scale of an operand tile taken PER PIXEL: the input's
 * from the tile itself (no magnitude word: network inputs are taken as they are), an intermediate activation's from a bound
 * (its input's scale x the layer's largest absolute row sum + max |bias| (+ max |t|)), carried from layer to layer. */
SBMC_API int sbmc_pointwise_chain_supported(int cin, int nl, const int *cout, long hw);
SBMC_API int sbmc_pointwise_chain_fwd_f32(const float *x, const float *t, const unsigned *tmax, const float *const *w,
                                 const float *const *bias,
                                 float *const *y, unsigned *const *signs, unsigned *const *amax, float *ymean,
                                 int nl, const int *cout, const int *act, const float *slope, int b, int s, int cin,
                                 long hw, int t_mode, void *stream);
/* ABI 8 -- the forward of a WIDE layer without a context term (128 < cout <= 512: the 441-channel logits, reference
 * sbmc/models.py:98-102) on the chain pass's machinery (csrc/pointwise_chain.hip pw_wide_fwd_kernel): the input tile staged once
 * per 64 pixels for all row tiles, per-pixel scale (no magnitude word needed), the same arithmetic.  amax (or NULL): a zeroed
 * device word raised to the bit pattern of max |y|.  No sign words (a wide layer's backward reads none). */
SBMC_API int sbmc_pointwise_wide_fwd_supported(int cin, int cout, long hw);
SBMC_API int sbmc_pointwise_wide_fwd_f32(const float *x, const float *w, const float *bias, float *y, unsigned *amax, int b,
                                int cin, int cout, long hw, int act, float slope, void *stream);
/* ABI 8 -- the backward of TWO consecutive 128-channel per-sample 1 x 1 layers in one pass (csrc/pointwise_chain_bwd.hip):
 * y_a = act_a(w_a x_a + b_a + t), y_b = act_b(w_b y_a + b_b) (reference sbmc/modules.py:154-175; the embeddings' and the
 * regressor's first two layers, sbmc/models.py:79-102).  The gradient of y_a never leaves the chip.
 *   gy [b, 128, hw]: gradient of y_b; signs_b: y_b's sign words (sbmc_pointwise_fwd_signs_f32; NULL when act_b == 0);
 *   xb = y_a [b, 128, hw] (layer a's activation decisions are taken from it: y_a > 0); xa [b, cin, hw], cin <= 128.
 *   gxa [b, cin, hw] or NULL; gw_partial_b [groups, 128, 128], gb_partial_b [groups, 128], gw_partial_a [groups, 128, cin],
 *   gb_partial_a [groups, bq, 128] (bq = b / s with t_mode 1, else 1), every element written; groups =
 *   sbmc_pointwise_chain_bwd_groups(b, s, t_mode, hw).  gt [b / s, 128, hw] (t_mode 2: the per-pixel context gradient) or
 *   NULL; with t_mode 1 the per-image context gradient is gb_partial_a summed over the groups.
 *   gmax, xbmax, xamax: device words holding the bit patterns of floats >= max |gy|, max |xb|, max |xa| (the two-f16-plane
 *   format's scales; gz_a's scale is a bound: the largest absolute column sum of w_b x the word of gy); gxmax (or NULL):
 *   a zeroed word raised to max |gxa|. */
SBMC_API int sbmc_pointwise_chain_bwd_supported(int cin, long hw);
SBMC_API int sbmc_pointwise_chain_bwd_groups(int b, int s, int t_mode, long hw);
SBMC_API int sbmc_pointwise_chain_bwd_f32(const float *gy, const unsigned *signs_b, const float *xb, const float *wb,
                                 const float *xa, const float *wa, float *gxa, float *gw_partial_b, float *gb_partial_b,
                                 float *gw_partial_a, float *gb_partial_a, float *gt, const unsigned *gmax,
                                 const unsigned *xbmax, const unsigned *xamax, unsigned *gxmax, int b, int s, int cin,
                                 long hw, int t_mode, int act_b, float slope_b, int act_a, float slope_a, void *stream);
/* the all-half layer (x, y _Float16) with the mean as a _Float16 tensor: the mean of the half values as stored */
SBMC_API int sbmc_pointwise_fwd_mean_f16(const void *x, const float *w, const float *bias, const float *t, void *y,
                                void *ymean, int s_mean, int b, int s, int cin, int cout, long hw, int t_mode,
                                int act, float slope, void *stream);
SBMC_API int sbmc_pointwise_bwd_signs_f32(const float *gy, const unsigned *signs, const float *x, const float *w,
                                 float *gx, float *gw_partial, float *gb_partial, float *gt,
                                 const float *gmean, int s_mean, int b, int s, int cin, int cout, long hw,
                                 int t_mode, int act, float slope, void *stream);

/* The same backward with half-precision STORAGE (training under torch.autocast(float16), SURVEY.md row
 * N4): gy, y, gmean are _Float16; x and gx are _Float16 (x_is_half = 1) or float (a chain's first
 * layer, whose input is the network's fp32 features); w, gw_partial, gb_partial and gt stay float, every
 * product and sum is fp32 (fp32 MFMA), gx is rounded once on the store. */
SBMC_API int sbmc_pointwise_bwd_f16(const void *gy, const void *y, const void *x, int x_is_half, const float *w,
                           void *gx, float *gw_partial, float *gb_partial, float *gt,
                           const void *gmean, int s_mean, int b, int s, int cin, int cout, long hw,
                           int t_mode, int act, float slope, void *stream);

/* ---- the U-net's up path: bilinear x2 upsampling + channel concatenation in one pass ---------
 * Reference sbmc/modules.py:300-320: F.interpolate(coarse, scale 2, "bilinear", align_corners=False)
 * followed by th.cat([up, left], 1).
 *   fwd: out[b, :cu] = upsample(coarse[b]), out[b, cu:] = left[b]
 *        coarse [b, cu, h, w], left [b, cl, 2h, 2w], out [b, cu + cl, 2h, 2w]; w even
 *   bwd: gcoarse = adjoint of the upsampling applied to gout[:, :cu] (a gather: no atomics);
 *        the gradient of `left` is the channel slice gout[:, cu:] itself
 */
SBMC_API int sbmc_upsample2x_cat_supported(int h, int w);
SBMC_API int sbmc_upsample2x_cat_fwd_f32(const float *coarse, const float *left, float *out, int b,
                                int cu, int cl, int h, int w, void *stream);
SBMC_API int sbmc_upsample2x_cat_bwd_f32(const float *gout, float *gcoarse, int b, int cu, int cl,
                                int h, int w, void *stream);

/* Row-slab form (one frame sharded along H, sbmc_amd/dist.py): coarse / gcoarse hold hc = top + h + bot
 * rows whose first `top` and last `bot` (0 or 1) are the neighbouring slabs' edge rows; left / out / gout
 * are the 2h fine rows of this slab.  The edge clamp of the interpolation acts at true image borders
 * (top == 0 / bot == 0) only; gcoarse of a halo row holds this slab's share of its gradient (the caller
 * returns it to its owner).  (top, bot) = (0, 0) is the whole-frame call. */
SBMC_API int sbmc_upsample2x_cat_slab_fwd_f32(const float *coarse, const float *left, float *out, int b,
                                     int cu, int cl, int hc, int w, int top, int bot, void *stream);
SBMC_API int sbmc_upsample2x_cat_slab_bwd_f32(const float *gout, float *gcoarse, int b, int cu, int cl,
                                     int hc, int w, int top, int bot, void *stream);

/* ---- channels-last (NHWC) forms of the glue around the U-nets' 3x3 convolutions (csrc/nhwc_ops.hip) ----
 * MIOpen's fastest fp32 3x3 solvers on gfx950 are NHWC-native; with the U-net's activations kept
 * channels-last between its convolutions MIOpen stops transposing every input and output.  Tensors here
 * are [pixels = b*h*w, c] (bias_act) or [b, h, w, c] (upsample), c multiples of 4, 16-byte aligned.
 *   bias_act_nhwc fwd (in place): y = act(y + bias[c]);
 *   bwd: gx = gy * act'(y) (gx may alias gy), partial[j, c] = sums over the pixels of workgroup j,
 *        j < sbmc_bias_act_nhwc_chunks(pixels, c); the bias gradient is the sum over j (no atomics);
 *        only where sbmc_bias_act_nhwc_supported(c) == 1 (c / 4 divides 256);
 *   upsample2x_cat_nhwc: as sbmc_upsample2x_cat_* (reference sbmc/modules.py:300-320); the backward also
 *        writes the skip connection's gradient gleft = gout[..., cu:] as a contiguous tensor (either
 *        output pointer may be NULL: not computed). */
/* Batched 2-d transpose dst[b][cols][rows] = src[b][rows][cols] (rows, cols multiples of 4): the planar <->
 * channels-last conversion at the U-net's entry, (rows, cols) = (c, h*w) one way and (h*w, c) the other. */
SBMC_API int sbmc_transpose2d_f32(const float *src, float *dst, int b, int rows, int cols, void *stream);
/* 2 x 2 / stride 2 max-pooling of a channels-last map x [b, 2 hc, 2 wc, c] -> y [b, hc, wc, c] (the U-nets' pooling,
 * reference sbmc/modules.py:262-263) and its adjoint fused with the addition of the skip connection's gradient:
 * gx = gskip (NULL: 0) + gpool routed to the FIRST maximum of each window (the element torch's max_pool2d records; the
 * arg-max is recomputed from x).  elem: 4 = float, 2 = _Float16; c % 4 == 0. */
SBMC_API int sbmc_maxpool2_nhwc_fwd(const void *x, void *y, int b, int hc, int wc, int c, int elem, void *stream);
SBMC_API int sbmc_maxpool2_nhwc_bwd_add(const void *x, const void *gpool, const void *gskip, void *gx, int b, int hc,
                               int wc, int c, int elem, void *stream);
/* ABI 7: the same routing with the adjoint of the bias + activation pass that PRODUCED x in the same pass (fp32; x is a
 * convolution chain's activated output and the pooling + skip node its only reader): gx = (gskip + routed gpool) *
 * (z > 0 ? 1 : slope) from the producer's sign words (sbmc_conv3x3_bias_act_nhwc_f32), partial
 * [sbmc_bias_act_nhwc_chunks(b hc wc, c)][c]: the bias gradient's partial sums, *amax (a zeroed word) raised to max |gx|.
 * c as sbmc_bias_act_nhwc_supported. */
SBMC_API int sbmc_maxpool2_nhwc_bwd_add_adj_f32(const float *x, const float *gpool, const float *gskip, float *gx,
                                       const unsigned *signs, float slope, float *partial, unsigned *amax, int b, int hc,
                                       int wc, int c, void *stream);
/* the same batched 2-d transpose of _Float16 tensors (rows, cols multiples of 4; 8-byte aligned) */
SBMC_API int sbmc_transpose2d_f16(const void *src, void *dst, int b, int rows, int cols, void *stream);
/* ... and *amax raised to the bit pattern of the largest magnitude of the tensor (see sbmc_bias_act_nhwc_fwd_amax_f32). */
SBMC_API int sbmc_transpose2d_amax_f32(const float *src, float *dst, unsigned *amax, int b, int rows, int cols,
                              void *stream);
SBMC_API int sbmc_bias_act_nhwc_supported(int c);
SBMC_API int sbmc_bias_act_nhwc_chunks(long pixels, int c);
SBMC_API int sbmc_bias_act_nhwc_fwd_f32(float *y, const float *bias, long pixels, int c, int act, float slope,
                               void *stream);
SBMC_API int sbmc_bias_act_nhwc_bwd_f32(const float *gy, const float *y, float *gx, float *partial, long pixels,
                               int c, int act, float slope, void *stream);
/* bias_act_nhwc with SIGN BITS (act = 1 or 2): the forward also writes one bit per element, pre-activation > 0
 * -- float4 number i of y (i = pixel * c / 4 + channel quad) owns nibble i % 8 of word i / 8,
 * ceil(pixels * c / 32) words --, and the backward's activation adjoint reads those instead of y. */
SBMC_API int sbmc_bias_act_nhwc_fwd_signs_f32(float *y, const float *bias, unsigned *signs, long pixels, int c,
                                     int act, float slope, void *stream);
SBMC_API int sbmc_bias_act_nhwc_bwd_signs_f32(const float *gy, const unsigned *signs, float *gx, float *partial,
                                     long pixels, int c, int act, float slope, void *stream);
/* The two passes above which also RAISE *amax (an atomic maximum: the caller hands in a word holding 0, or a
 * bound it wants kept) to the bit pattern of the largest magnitude of what they wrote (y / gx): the 3 x 3
 * convolution that consumes it (sbmc_conv3x3_*) takes its scale from there instead of running
 * sbmc_conv3x3_absmax_f32 over the tensor.  ABI 5: the passes no longer zero the word themselves (a memset launch
 * per pass).  fwd: signs may be NULL (no sign bits; act 0 allowed); bwd: signs NULL <=> act 0. */
SBMC_API int sbmc_bias_act_nhwc_fwd_amax_f32(float *y, const float *bias, unsigned *signs, unsigned *amax, long pixels,
                                    int c, int act, float slope, void *stream);
SBMC_API int sbmc_bias_act_nhwc_bwd_amax_f32(const float *gy, const unsigned *signs, float *gx, float *partial,
                                    unsigned *amax, long pixels, int c, int act, float slope, void *stream);
SBMC_API int sbmc_upsample2x_cat_nhwc_supported(int cu, int cl, int h, int w);
SBMC_API int sbmc_upsample2x_cat_nhwc_fwd_f32(const float *coarse, const float *left, float *out, int b, int cu,
                                     int cl, int h, int w, void *stream);
SBMC_API int sbmc_upsample2x_cat_nhwc_bwd_f32(const float *gout, float *gcoarse, float *gleft, int b, int cu,
                                     int cl, int h, int w, void *stream);
/* row-slab form of the two (see sbmc_upsample2x_cat_slab_*): coarse / gcoarse hold hc = top + h + bot rows */
SBMC_API int sbmc_upsample2x_cat_nhwc_slab_fwd_f32(const float *coarse, const float *left, float *out, int b,
                                          int cu, int cl, int hc, int w, int top, int bot, void *stream);
SBMC_API int sbmc_upsample2x_cat_nhwc_slab_bwd_f32(const float *gout, float *gcoarse, float *gleft, int b,
                                          int cu, int cl, int hc, int w, int top, int bot, void *stream);
/* ABI 7: the whole frame's adjoint (top = bot = 0) with the activation adjoint of the layer that PRODUCED the coarse map in
 * the same pass (a convolution chain's activated output that only the upsampling reads): gcoarse = gather(gout) *
 * (z > 0 ? 1 : slope) from that layer's sign words, partial [sbmc_bias_act_nhwc_chunks(b h w, cu)][cu]: the bias gradient's
 * partial sums, *amax (a zeroed word) raised to max |gcoarse|; gleft or NULL.  cu as sbmc_bias_act_nhwc_supported. */
SBMC_API int sbmc_upsample2x_cat_nhwc_bwd_adj_f32(const float *gout, float *gcoarse, float *gleft, const unsigned *signs,
                                         float slope, float *partial, unsigned *amax, int b, int cu, int cl, int h,
                                         int w, void *stream);
/* the same on _Float16 tensors (fp16 activations under torch.autocast): half storage, fp32 interpolation */
SBMC_API int sbmc_upsample2x_cat_nhwc_slab_fwd_f16(const void *coarse, const void *left, void *out, int b,
                                          int cu, int cl, int hc, int w, int top, int bot, void *stream);
SBMC_API int sbmc_upsample2x_cat_nhwc_slab_bwd_f16(const void *gout, void *gcoarse, void *gleft, int b,
                                          int cu, int cl, int hc, int w, int top, int bot, void *stream);

/* ---- neighbour halo transport for one frame sharded over the GPUs of a node (csrc/halo.hip) ----------
 * New functionality (the reference is single-process; closest analogue: the overlapped tiles of
 * scripts/denoise.py:54-93).  Every rank owns a MAILBOX in uncached device memory that its two neighbours map
 * through HIP IPC; `put` packs rows of a tensor straight into the neighbour's mailbox (peer stores over xGMI)
 * and raises a flag there, `get` waits for the flag on the device, unpacks into the consumer's tensor and hands
 * the slot back: two launches per exchange, no communication library and no host round trip on the data path.
 * These are the ONLY entry points of the library that allocate: the caller owns a mailbox from
 * sbmc_halo_alloc to sbmc_halo_free and a mapping from sbmc_halo_open to sbmc_halo_close.
 *
 * Directions: 0 = up (rank - 1), 1 = down (rank + 1).  A mailbox has two rings of `nslots` slots of
 * `slot_bytes` (a multiple of 16); message number `seq` of a direction (0, 1, 2, ... counted by the caller,
 * the same on both ends of a link) uses slot seq % nslots, and its sender first waits until message
 * seq - nslots was consumed.  A message is `chunks` runs of `chunk_bytes`, `pitch` bytes apart in the tensor
 * and dense in the slot.  Waits are bounded by `timeout_ticks` of the 100 MHz wall clock; after a time-out
 * the kernels run on and sbmc_halo_status reports a non-zero word (1/2: put waited for a free slot up/down,
 * 3/4: get waited for data from above/below).
 */
#define SBMC_HALO_HANDLE_BYTES 64
/* bytes of a mailbox with the given ring geometry (0: bad arguments) */
SBMC_API size_t sbmc_halo_bytes(long long slot_bytes, int nslots);
/* allocates and zeroes a mailbox on the current device, returns its address and its IPC handle */
SBMC_API int sbmc_halo_alloc(size_t bytes, void **base, unsigned char *handle);
SBMC_API int sbmc_halo_free(void *base);
/* maps the mailbox another process exported (any GPU of the node) into this one / unmaps it */
SBMC_API int sbmc_halo_open(const unsigned char *handle, void **base);
SBMC_API int sbmc_halo_close(void *peer_base);
/* reads the time-out word of a mailbox (synchronous 4-byte copy: call at a synchronisation point) */
SBMC_API int sbmc_halo_status(void *box, unsigned *err);
/* the same word as a float (1: some wait has timed out, 0: none) written to device memory ON THE STREAM: the flag can
 * ride in a collective, so that every rank learns of a time-out at the same point of its step */
SBMC_API int sbmc_halo_status_to(void *box, float *dst, void *stream);
/* sends src_up to the mailbox up_box and src_down to down_box (a NULL box: no such neighbour).  amax (or NULL):
 * the device word with the bit pattern of the largest magnitude of the tensor the rows are cut from (what
 * sbmc_conv3x3_* scale by); it travels with the message. */
SBMC_API int sbmc_halo_put(void *box, void *up_box, void *down_box, const void *src_up, const void *src_down,
                  long long chunks, long long chunk_bytes, long long pitch, unsigned seq_up, unsigned seq_down,
                  int nslots, long long slot_bytes, long long timeout_ticks, const unsigned *amax, void *stream);
/* receives into dst_up / dst_down (NULL: nothing expected from there): dst = received, or
 * dst = add + received with add_elem = 4 (float) or 2 (_Float16) -- the adjoint of halo padding.  In the
 * same launch, optionally, a plain 2-d copy body_src -> body_dst (the slab's own rows; body_src NULL: the
 * run body_dst is filled with zeros).  amax (or NULL): a device word that is raised (atomic maximum) to the words
 * the senders attached to the received messages -- handed the word of the receiving tensor's own rows it ends up
 * bounding the padded map, with no pass over it. */
SBMC_API int sbmc_halo_get(void *box, void *up_box, void *down_box, void *dst_up, void *dst_down,
                  const void *add_up, const void *add_down, int add_elem,
                  long long chunks, long long chunk_bytes, long long dst_pitch, long long add_pitch,
                  void *body_dst, const void *body_src, long long body_chunks, long long body_chunk_bytes,
                  long long body_dst_pitch, long long body_src_pitch,
                  unsigned seq_up, unsigned seq_down, int nslots, long long slot_bytes, long long timeout_ticks,
                  unsigned *amax, void *stream);
/* Cross-rank merge of the splat's running state (reference sbmc/modules.py:450-471: M = max(m1, m2),
 * sums rescaled by exp(m - M)).  ext [bs, c + 2, top + rows + bot, w] is this rank's partial state on its slab
 * extended by p rows towards each neighbour (channels: c of sum_r, sum_w, max_w; top, bot = p or 0); its
 * overhang rows were sent with sbmc_halo_put (chunks = bs * (c + 2), chunk = p * w floats).  fwd waits for the
 * neighbours' overhangs, merges them into the first / last p of the slab's own rows -> out [bs, c + 2, rows, w]
 * and keeps what arrived in recv_up / recv_down [bs, c + 2, p, w]; bwd (local) returns the gradient of the own
 * rows into gext[:, :, top : top + rows] and the gradient of what was received (to be sent back and stored
 * into the sender's gext overhang rows with sbmc_halo_put / sbmc_halo_get). */
SBMC_API int sbmc_halo_merge_state_fwd_f32(void *box, void *up_box, void *down_box, const float *ext, float *out,
                                  float *recv_up, float *recv_down, int bs, int c, int rows, int w, int p,
                                  int top, int bot, unsigned seq_up, unsigned seq_down, int nslots,
                                  long long slot_bytes, long long timeout_ticks, void *stream);
SBMC_API int sbmc_halo_merge_state_bwd_f32(const float *ext, const float *recv_up, const float *recv_down,
                                  const float *gout, float *gext, float *grecv_up, float *grecv_down,
                                  int bs, int c, int rows, int w, int p, int top, int bot, void *stream);

/*
 * 3 x 3 convolution, stride 1, zero padding 1, of a channels-last fp32 image -- replaces the cuDNN convolution
 * behind every `nn.Conv2d(..., 3, padding=1)` of the reference's U-nets (sbmc/modules.py:195-320 via
 * ttools ConvChain) at fp32 accuracy on the f16 matrix pipe (csrc/conv3x3.hip: x = h + l in two f16 planes,
 * three products per term, scales from the tensors' largest magnitudes taken on the device).
 *   supported:        cin % 32 == 0, cout % 128 == 0 (and sizes whose tile offsets fit 31 bits)
 *   weights_bytes:    size of the prepared weights of one layer (0: unsupported)
 *   absmax:           *out = bit pattern of max |x| over n floats (x 16-byte aligned)
 *   prepare_weights:  w[co][ci][ky][kx] with element strides s_* (storage_elems floats of dense storage) ->
 *                     wp; flip != 0 prepares the adjoint's weights: call with cin / cout and s_co / s_ci
 *                     exchanged, taps are mirrored (gx = conv(gy, flipped transposed w))
 *   nhwc:             y[n][h][w][cout] = sum x[n][h + ky - 1][w + kx - 1][ci] w[co][ci][ky][kx]
 */
SBMC_API int sbmc_conv3x3_supported(int n, int h, int w, int cin, int cout);
SBMC_API size_t sbmc_conv3x3_weights_bytes(int cin, int cout);
SBMC_API int sbmc_conv3x3_absmax_f32(const float *x, long n, unsigned *out, void *stream);
/* ... RAISING *out instead of setting it (atomic maximum with what it holds) */
SBMC_API int sbmc_conv3x3_absmax_raise_f32(const float *x, long n, unsigned *out, void *stream);
SBMC_API int sbmc_conv3x3_prepare_weights_f32(const float *w, long s_co, long s_ci, long s_ky, long s_kx,
                                     long storage_elems, int cin, int cout, int flip, void *wp, void *stream);
/* Stream-K workspace (optional, `ws` below; NULL: whole tiles are dealt round-robin to the compute units): a launch whose
 * tile count is no multiple of the CU count then cuts its (tile, 32-channel chunk) units into equal ranges over ALL
 * compute units; partial tiles meet in `ws` (workspace_bytes of device memory, 16-byte aligned, used by one launch at
 * a time) and a second, small launch adds them in a fixed order and runs their epilogues: the result does not depend on
 * the run. */
SBMC_API size_t sbmc_conv3x3_workspace_bytes(void);
SBMC_API int sbmc_conv3x3_nhwc_f32(const float *x, const unsigned *xmax, const void *wp, float *y, int n, int h,
                          int w, int cin, int cout, void *ws, void *stream);
/* The same convolution with the bias + activation pass behind it (reference sbmc/modules.py:154-175) in its
 * epilogue: y = act(conv(x) + bias[cout]), act 0 linear / 1 ReLU / 2 LeakyReLU(slope).  signs (or NULL): one bit
 * per output (pre-activation > 0), bit e % 32 of word e / 32 of the channels-last element index e -- what
 * sbmc_bias_act_nhwc_bwd_signs_f32 / _bwd_amax_f32 read; amax (or NULL): raised to the bit pattern of max |y| (a
 * word the caller zeroed). */
SBMC_API int sbmc_conv3x3_bias_act_nhwc_f32(const float *x, const unsigned *xmax, const void *wp, const float *bias,
                                   float *y, unsigned *signs, unsigned *amax, int n, int h, int w, int cin,
                                   int cout, int act, float slope, void *ws, void *stream);
/* ABI 7 -- the DATA GRADIENT of such a layer whose input is the activated output of another one that nothing else reads
 * (a chain of convolutions: reference sbmc/modules.py:195-320, ttools ConvChain), with that producing layer's
 * activation adjoint + bias gradient pass (sbmc_bias_act_nhwc_bwd_amax_f32) in the kernel's epilogue -- the gradient
 * is written once instead of written, read and rewritten:
 *     gz[n][h][w][cout] = (sum gy[n][h + ky - 1][w + kx - 1][ci] wT[co][ci][ky][kx]) * (z > 0 ? 1 : slope)
 *   gy [n, h, w, cin] with its magnitude word gmax; wp: the adjoint's prepared weights (prepare_weights flip = 1 / the
 *   bank's wp_bwd); cout = the producing layer's output channels (adj_supported: conv3x3_supported and cout <= 512);
 *   signs: the words that layer's forward left (sbmc_conv3x3_bias_act_nhwc_f32); slope 0: ReLU;
 *   partial [sbmc_conv3x3_adj_partial_rows()][cout], ZEROED by the caller: partial sums of the bias gradient, sum over
 *   pixels of gz (which rows a launch writes is its own business; fixed order: the result does not depend on the run);
 *   the caller adds the rows up, e.g. as bias_partial of sbmc_conv3x3_wgrad_bias_f32;
 *   amax: raised to the bit pattern of max |gz| (a word the caller zeroed). */
SBMC_API int sbmc_conv3x3_adj_supported(int n, int h, int w, int cin, int cout);
SBMC_API int sbmc_conv3x3_adj_partial_rows(void);
SBMC_API int sbmc_conv3x3_adj_nhwc_f32(const float *gy, const unsigned *gmax, const void *wp, const unsigned *signs,
                              float slope, float *gz, float *partial, unsigned *amax, int n, int h, int w, int cin,
                              int cout, void *ws, void *stream);
/* Half activations (torch.autocast(float16) semantics: half inputs, weights rounded to half once, fp32 accumulation,
 * half outputs): x, y _Float16 channels-last; wp the prepared weights of prepare_weights_f32 / the weight bank (their
 * high plane is f16 of the power-of-two-scaled weight, the scale is divided out); bias fp32; no magnitude words. */
SBMC_API int sbmc_conv3x3_nhwc_f16(const void *x, const void *wp, void *y, int n, int h, int w, int cin, int cout,
                          void *ws, void *stream);
SBMC_API int sbmc_conv3x3_bias_act_nhwc_f16(const void *x, const void *wp, const float *bias, void *y, unsigned *signs,
                                   int n, int h, int w, int cin, int cout, int act, float slope, void *ws,
                                   void *stream);
/* ... its weight gradient (gy, x _Float16; gw and gbias fp32) and the activation adjoint + bias partial sums on half
 * gradients (signs NULL <=> act 0; gx may alias gy) */
SBMC_API int sbmc_conv3x3_wgrad_bias_f16(const void *gy, const void *x, float *gw, long s_co, long s_ci, long s_ky,
                                long s_kx, void *scratch, int n, int h, int w, int cin, int cout,
                                const float *bias_partial, int bias_chunks, int bias_c, float *gbias, void *stream);
SBMC_API int sbmc_bias_act_nhwc_bwd_signs_f16(const void *gy, const unsigned *signs, void *gx, float *partial,
                                     long pixels, int c, int act, float slope, void *stream);
/* Weight gradient of the same convolution: gw[co][ci][ky][kx] (element strides s_*) = sum over the pixels of
 * gy[n][y][x][co] x[n][y + ky - 1][x + kx - 1][ci]; gmax / xmax: the tensors' largest magnitudes as written by
 * sbmc_conv3x3_absmax_f32; scratch: wgrad_scratch_bytes of device memory (partial sums of the pixel ranges,
 * added in a fixed order: the result does not depend on the run).  cin % 128 == 0, cout % 128 == 0. */
SBMC_API int sbmc_conv3x3_wgrad_supported(int n, int h, int w, int cin, int cout);
SBMC_API size_t sbmc_conv3x3_wgrad_scratch_bytes(int n, int h, int w, int cin, int cout);
SBMC_API int sbmc_conv3x3_wgrad_f32(const float *gy, const unsigned *gmax, const float *x, const unsigned *xmax,
                           float *gw, long s_co, long s_ci, long s_ky, long s_kx, void *scratch, int n, int h,
                           int w, int cin, int cout, void *stream);
/* ... which also adds up the bias gradient's per-workgroup partial sums bias_partial [bias_chunks][bias_c] (what
 * sbmc_bias_act_nhwc_bwd_* leave) into gbias [bias_c], in a fixed order, in its reduction launch (a launch of its
 * own otherwise).  bias_partial NULL: exactly sbmc_conv3x3_wgrad_f32. */
SBMC_API int sbmc_conv3x3_wgrad_bias_f32(const float *gy, const unsigned *gmax, const float *x, const unsigned *xmax,
                                float *gw, long s_co, long s_ci, long s_ky, long s_kx, void *scratch, int n, int h,
                                int w, int cin, int cout, const float *bias_partial, int bias_chunks, int bias_c,
                                float *gbias, void *stream);


/*
 * Weight bank (csrc/conv3x3.hip): what a step derives from its weight-normalised convolution weights -- the
 * reference wraps every convolution in torch's weight norm, w = g v / ||v|| per output channel
 * (sbmc/modules.py:85-94, 178-188) -- for up to SBMC_WBANK_MAX layers per call instead of 2 (1 x 1 layers) to 8
 * (3 x 3 layers) launches per layer and step.
 *   forward:  w and ||v|| of every entry; for 3 x 3 entries with wp_fwd / wp_bwd also the prepared weights of
 *             sbmc_conv3x3_prepare_weights_f32 in both orientations (wp_fwd: cin -> cout; wp_bwd: the adjoint's,
 *             cout -> cin, taps mirrored), scaled by the layer's largest |w|.  `norm` holds 2 cout floats (the
 *             norms, then scratch).
 *   backward: gv = (g / n)(gw - v <gw, v> / n^2), gg = <gw, v> / n for every entry; gw in any element strides
 *             (NULL: the loss does not depend on the layer, zeros are written).
 */
#define SBMC_WBANK_MAX 24
typedef struct sbmc_wbank_entry {
    const float *v;     /* weight_v [cout][cin][kh][kw], dense */
    const float *g;     /* weight_g [cout] */
    float *w;           /* out: the weight, layout of v */
    float *norm;        /* out: [2 cout]: ||v|| per output channel, then scratch */
    void *wp_fwd;       /* out or NULL: sbmc_conv3x3_weights_bytes(cin, cout) bytes */
    void *wp_bwd;       /* out or NULL: sbmc_conv3x3_weights_bytes(cout, cin) bytes */
    int cout, cin, kh, kw;
} sbmc_wbank_entry;
typedef struct sbmc_wbank_grad {
    const float *gw;    /* gradient of w, element strides below; or NULL */
    const float *v, *g, *norm;
    float *gv;          /* out: layout of v */
    float *gg;          /* out: [cout] */
    long s_co, s_ci, s_ky, s_kx;
    int cout, cin, kh, kw;
} sbmc_wbank_grad;
SBMC_API int sbmc_wbank_forward_f32(const sbmc_wbank_entry *entries, int n, void *stream);
SBMC_API int sbmc_wbank_backward_f32(const sbmc_wbank_grad *entries, int n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SBMC_HIP_H */
