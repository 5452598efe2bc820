/* emap_hip.h - C ABI of libemap_hip.so, the MI355X (gfx950) render hot path of EMAP.
 *
 * EMAP (cvg/EMAP) is pure Python/PyTorch and has no FFI layer; the drop-in boundary is the
 * Python class API of src/models/{udf_model,udf_renderer_blending,embedder,loss}.py (SURVEY.md
 * par. 8b).  The entry points below are what a binding for that boundary needs: every one names the
 * reference interface it replaces (file:line in the reference repo).  emap_amd/_lib.py binds them
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host; tensors are fp32, contiguous,
 *     row-major; index outputs are int64 (torch.long), like the reference's searchsorted / sort.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); no call synchronises,
 *     allocates device memory or keeps a reference to a caller buffer after the work it enqueued.
 *   - return value: 0 = ok, negative = EMAP_E_* (emap_last_error() has the text).  No C++ exception
 *     crosses the boundary.
 *   - the reference's NaN -> pdb.set_trace() convention (udf_renderer_blending.py:102-107,346-351,
 *     632-633) is replaced by a device error word `err_flags` (int32[1], may be NULL) into which
 *     kernels OR the EMAP_F_* bits; the host reads it lazily, never inside the hot path.
 */
#ifndef EMAP_HIP_H
#define EMAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMAP_ABI_VERSION 10

/* error codes */
#define EMAP_OK 0
#define EMAP_E_INVALID (-1)     /* bad argument / unsupported configuration */
#define EMAP_E_LAUNCH (-2)      /* HIP launch error                          */
#define EMAP_E_WORKSPACE (-3)   /* workspace too small                       */

/* err_flags bits */
#define EMAP_F_NAN_SAMPLES 1    /* sample_pdf produced NaN   (udf_renderer_blending.py:102)  */
#define EMAP_F_NAN_GRADERR 2    /* gradient_error is NaN     (udf_renderer_blending.py:632)  */
#define EMAP_F_MLP_NONFINITE 4  /* the MLP produced inf/NaN (fp16 modes: an activation or tangent left fp16's range;
                                   use EMAP_PREC_BF16X3 / EMAP_PREC_BF16 for such a network)          */

/* arithmetic mode of the MLP GEMMs (accumulation is always fp32) */
#define EMAP_PREC_BF16 0        /* one bf16 MFMA pass                                        */
#define EMAP_PREC_BF16X3 1      /* split bf16: a_hi*w_hi + a_lo*w_hi + a_hi*w_lo (~2^-17)     */
#define EMAP_PREC_F16 2         /* one fp16 MFMA pass (11-bit mantissa; |values| must stay < 65504) */
#define EMAP_PREC_F16X3 3       /* split fp16, three passes (~2^-22): the mode of the 1e-4 parity gate */
#define EMAP_PREC_F16X3M 4      /* EMAP_PREC_F16X3 with the cross terms of the value+gradient pass's FORWARD sweep as MX fp6 too (its
                                   reverse sweep has them in every split-fp16 build): d_hidden = 256 only; the kernel ~10 % faster,
                                   grad_x 6.2e-5 instead of 3.0e-5 of the 1e-4 gate (docs/DESIGN_LOG_r1-r4.md par. 6c); every other kernel = F16X3 */

#define EMAP_PREC_F16X3E 5      /* EMAP_PREC_F16X3 with f16 cross terms in BOTH sweeps of the value+gradient pass (no MX fp6 anywhere): the round-3
                                   arithmetic - grad_x 1.5e-5 (max-normalised) / 7.7e-3 (element-wise p99.9) instead of 3.0e-5 / 1.2e-2, the kernel
                                   ~12 % slower (ABI 7; profiles/r05_elementwise_attribution.txt) */

/* udf_type (udf_model.py:82-88) */
#define EMAP_UDF_ABS 0
#define EMAP_UDF_SQUARE 1
#define EMAP_UDF_SDF 2

#define EMAP_MAX_LIN 12

/* Constructor arguments of UDFNetwork that the kernels need (udf_model.py:8-21,24-45). */
typedef struct EmapNetConfig {
    int32_t d_hidden;   /* 128 or 256                                   */
    int32_t n_lin;      /* number of Linear layers = n_layers + 1       */
    int32_t skip_l;     /* layer whose input is cat([x, PE])/sqrt2, or -1 (skip_in=(4,) -> 4) */
    int32_t multires;   /* 0..10 positional-encoding octaves (0: raw xyz) */
    int32_t d_out;      /* must be 1                                    */
    int32_t udf_type;   /* EMAP_UDF_*                                   */
    float scale;        /* inputs * scale, udf / scale (udf_model.py:91,108) */
} EmapNetConfig;

int emap_abi_version(void);
const char* emap_last_error(void);

/* How UDFNetwork.gradient (udf_model.py:121-135) is evaluated, process-wide: -1 (default) = by launch size (reverse sweep from
 * 8 193 points in the split modes (round 6; 10 240 before) / 16 384 in the single-pass modes, forward-mode tangents below), 0 = always forward-mode tangents,
 * 1 = always the reverse sweep.  Read ONCE at library load from EMAP_GRAD_MODE=fwd|rev; this call changes it afterwards (tests, A/B
 * measurements).  Returns the previous setting.  Not a per-launch environment lookup any more (ABI 6). */
int emap_set_grad_mode(int mode);

/* ---- weights -------------------------------------------------------------------------------
 * Replaces the per-call weight_norm re-evaluation of nn.utils.parametrizations.weight_norm
 * (udf_model.py:73-74): folds W = g*v/||v|| once, permutes it into MFMA fragment order and
 * converts it for `prec`.  g[l] is [out,1], v[l] is [out,in], b[l] is [out] (state-dict
 * original0/original1/bias).  The three pointer tables are HOST arrays of device pointers. */
int emap_packed_bytes(const EmapNetConfig* cfg, int prec, size_t* bytes);
int emap_pack_weights(const EmapNetConfig* cfg, const float* const* g_host, const float* const* v_host,
                      const float* const* b_host, void* packed, int prec, void* stream);

/* ---- fields --------------------------------------------------------------------------------
 * emap_udf_fwd      : UDFNetwork.forward / .udf value  (udf_model.py:90-116)   x (P,3) -> udf (P)
 * emap_udf_fwd_grad : + UDFNetwork.gradient            (udf_model.py:121-135)  -> grad (P,3)
 * emap_embed        : Embedder.embed                   (embedder.py:34-35)     x (P,3) -> (P,3+6L) */
int emap_udf_fwd(const EmapNetConfig* cfg, const void* packed, int prec, const float* x, int64_t P,
                 float* udf, void* stream);
/* scratch: device buffer of emap_udf_scratch_bytes() bytes (0 for small launches; the reverse-mode kernel keeps its
 * per-workgroup sigma' slabs there); may be NULL when that size is 0 */
int emap_udf_scratch_bytes(const EmapNetConfig* cfg, int prec, int64_t P, size_t* bytes);
int emap_udf_fwd_grad(const EmapNetConfig* cfg, const void* packed, int prec, const float* x, int64_t P,
                      float* udf, float* grad3, void* scratch, size_t scratch_bytes, void* stream);
int emap_embed(const float* x, int64_t P, int multires, float* pe, void* stream);

/* ---- sampler -------------------------------------------------------------------------------
 * emap_sample_pdf    : sample_pdf(bins, weights, m, det=True)  (udf_renderer_blending.py:69-109)
 *                      bins (N,n), weights (N,n-1) -> samples (N,m), inds int64 (N,m) (may be NULL)
 * emap_upsample_step : up_sample_unbias                        (udf_renderer_blending.py:228-353)
 *                      z,udf (N,n) -> z_new (N,m), inds (N,m) (may be NULL)
 * emap_merge_sorted  : the cat + sort + gather of cat_z_vals   (udf_renderer_blending.py:361-375)
 *                      z (N,n), z_new (N,m) [, udf (N,n), udf_new (N,m)] -> z_out, udf_out (N,n+m),
 *                      perm int64 (N,n+m) (may be NULL); udf/udf_new/udf_out may be NULL (last=True) */
int emap_sample_pdf(const float* bins, const float* weights, int N, int n, int m, float* samples,
                    int64_t* inds, int32_t* err_flags, void* stream);
/* emap_sample_pdf_u : the same with the caller's uniform draws u (N, m) instead of the deterministic grid: sample_pdf(det=False)
 *                      (udf_renderer_blending.py:84-85; the reference draws u with torch.rand on the CPU generator) */
int emap_sample_pdf_u(const float* bins, const float* weights, const float* u, int N, int n, int m, float* samples, int64_t* inds,
                      int32_t* err_flags, void* stream);
int emap_upsample_step(const float* rays_o, const float* rays_d, const float* z, const float* udf, int N, int n,
                       int m, const float* sample_dist_dev, float inv_s, float beta, float gamma, float* z_new,
                       int64_t* inds, int32_t* err_flags, void* stream);
int emap_merge_sorted(const float* z, const float* z_new, const float* udf, const float* udf_new, int N, int n,
                      int m, float* z_out, float* udf_out, int64_t* perm, void* stream);

/* ---- compositing ---------------------------------------------------------------------------
 * The tail of render_core (udf_renderer_blending.py:435-455,463-677) after the MLP: given the
 * sorted z_vals and (udf, grad) at the interval mid-points.  Scalars are the already-transformed
 * inv_s = exp(10*variance).clip, beta, gamma (udf_renderer_blending.py:466-472).
 * Per-sample outputs are (N,S); per-ray outputs (N) or (N,3); `scalars` receives
 *   [0] gradient_error  [1] gradient_error_near_surface  [2] sparse_error
 *   [3] sum(relax*err) [4] sum(relax) [5] sum(near*err) [6] sum(near)   (for cross-rank reduction)
 *   [7] sum_rays sum_samples exp(-sparse_scale*udf)   [8] 1/inv_s  [9] 1/beta  [10] gamma  [11] inv_s
 * Any output pointer may be NULL. */
typedef struct EmapCompositeOut {
    float* weights;        /* (N,S)   :593-602 */
    float* alpha;          /* (N,S)   :545     */
    float* mid_z;          /* (N,S)   :446     */
    float* dists;          /* (N,S)   :435-444 */
    float* inside_sphere;  /* (N,S)   :568     */
    float* gradient_mag;   /* (N,S)   :463     */
    float* gradients_flip; /* (N,S,3) :637     */
    float* edge;           /* (N)     :606-609 */
    float* depth;          /* (N)     :607 (times depth_scale if given, render() :786) */
    float* weight_sum;     /* (N)     :604     */
    float* normals;        /* (N,3)   :662     */
    float* scalars;        /* (16)             */
} EmapCompositeOut;

int emap_composite_fwd(const float* rays_o, const float* rays_d, const float* z, const float* udf,
                       const float* grad3, const float* depth_scale, int N, int S, const float* sample_dist_dev,
                       float inv_s, float beta, float gamma, float cos_anneal_ratio, int has_cos_anneal,
                       float flip_saturation, float near_surface, float sparse_scale, float background,
                       int has_background, const EmapCompositeOut* out, float* partials /* (N,8) scratch */,
                       int32_t* err_flags, void* stream);


/* ---- fused forward render ------------------------------------------------------------------
 * UDFRendererBlending.render (udf_renderer_blending.py:679-800) for upsampling_type="classical",
 * use_unbias_render=True, n_outside=0: coarse z_vals (:705-720) -> importance_sample (:802-841) ->
 * render_core (:418-677).  near/far are (N) device arrays; t_rand (N) may be NULL (no jitter).
 * The whole sequence is enqueued on `stream` with no host synchronisation.
 * z_vals (N,S), udf (N,S), grad3 (N,S,3) are outputs as well (S = n_samples + n_importance//steps*steps). */
typedef struct EmapRenderParams {
    int32_t n_rays;
    int32_t n_samples;
    int32_t n_importance;
    int32_t up_sample_steps;
    float inv_s, beta, gamma;
    float cos_anneal_ratio;
    int32_t has_cos_anneal;
    float flip_saturation;
    float near_surface;
    float sparse_scale;
    float background;
    int32_t has_background;
    /* optional: raw nn.Parameters on the device (variance, beta, gamma of SingleVarianceNetwork /
     * BetaNetwork, udf_model.py:215,248-253).  When variance_dev != NULL the kernels evaluate
     * inv_s = exp(10*variance).clip(1e-6,1e6), beta = exp(10*beta).clip(0,1/beta_min).clip(1e-6,1e6),
     * gamma = exp(10*gamma).clip(1e-6,1e6) themselves (udf_renderer_blending.py:466-472) and ignore the
     * three by-value fields above, so no device->host read of the parameters is ever needed. */
    const float* variance_dev;
    const float* beta_dev;
    const float* gamma_dev;
    float beta_min;
    int32_t reserved;
} EmapRenderParams;

/* same, with inv_s/beta/gamma taken from raw device parameters (see EmapRenderParams) */
int emap_composite_fwd_p(const float* rays_o, const float* rays_d, const float* z, const float* udf,
                         const float* grad3, const float* depth_scale, int N, int S, const float* sample_dist_dev,
                         const EmapRenderParams* p, const EmapCompositeOut* out, float* partials,
                         int32_t* err_flags, void* stream);

/* ABI v8: emap_render_fwd runs importance_sample (udf_renderer_blending.py:802-841) as ONE launch where the shape allows (ABI 9: 1 <=
 * n_importance / up_sample_steps <= 16 new samples per step; measured launch-size rule in csrc/udf_mlp_kernel.inc:launch_is_mode - round 6:
 * fused up to 512 rays, the chain of dense launches beyond); 0 restores the chain of 2 K - 1 launches (same results bit for bit: tests, A/B), 2 (ABI 9) uses
 * the fused kernel at every launch size.  Process-wide; returns the previous value. */
int emap_set_fused_sampling(int on);
/* ABI v9: emap_render_fwd composites every ray INSIDE the final value + grad_x launch (the workgroup that writes a ray's last point runs
 * render_core's tail for it, udf_renderer_blending.py:463-625; BASELINE config C2: "fused MLP + composite kernel") whenever that launch is
 * the reverse-sweep kernel (>= 8 193 points in the split modes); the workgroup that composites the render's last ray then runs the
 * deterministic cross-ray reduction too (same additions in the same order as composite_reduce_kernel): the render ends with that launch.
 * 0 restores the separate compositing and reduction launches (same results bit for bit: tests, A/B; EMAP_FUSED_COMPOSITE=0 at load).  Process-wide;
 * returns the previous value. */
int emap_set_fused_composite(int on);
/* ABI v10: 1 = the wide value launches (the coarse pass of a render: UDFNetwork.udf on n_samples points per ray, udf_renderer_blending.py:722-725,
 * udf_model.py:90-116; >= 32 768 points, split-fp16 / split-bf16 modes at d_hidden = 256 except f16x3m) run the FORWARD sweep of the 32x32x16
 * value + grad_x kernel as a value kernel (csrc/udf_mlp_rev32.inc, VAL: same tile geometry as the 16x16x32 value kernel, half the MFMA
 * instructions, pinned K-loop).  Default 0 (udf_mlp_fs2_kernel for every value launch; EMAP_VALUE32=1 at load turns it on): measured -0.9 % /
 * 0 / -3.3 % on the 512 / 1024 / 4096-ray render, and the two forms sum a GEMM's products in different orders - udf agrees to fp32 rounding
 * (tests), not bit for bit, so with it a render's z_vals depend on the launch size.  Process-wide; returns the previous value. */
int emap_set_value_tile_mode(int on);
int emap_render_workspace_bytes(const EmapNetConfig* cfg, int prec, const EmapRenderParams* p, size_t* bytes);
int emap_render_fwd(const EmapNetConfig* cfg, const void* packed, int prec, const EmapRenderParams* p,
                    const float* rays_o, const float* rays_d, const float* near, const float* far,
                    const float* t_rand, const float* depth_scale, float* z_vals, float* udf, float* grad3,
                    const EmapCompositeOut* out, void* workspace, size_t workspace_bytes, int32_t* err_flags,
                    void* stream);

/* ---- training backward (SURVEY par. 8 f1) --------------------------------------------------------
 * Replaces torch.autograd through render_core and UDFNetwork.gradient(create_graph=True) under loss.backward()
 * (udf_renderer_blending.py:457-625, udf_model.py:121-135, runner_udf.py:166-167).  importance_sample is @torch.no_grad
 * and z_samples are detached (udf_renderer_blending.py:344,802), so parameters receive gradient only through the final
 * (udf, grad_x udf) evaluation at the sample points and through inv_s / beta / gamma:
 *
 * emap_composite_bwd : dL/d{edge, depth, gradient_error, gradient_error_near_surface}  ->  dL/dudf (N,S),
 *                      dL/d(grad_x udf) (N,S,3), dL/d{variance, beta, gamma}
 * emap_udf_vjp       : (dL/dudf (P), dL/dgrad (P,3)) at points x (P,3)  ->  dL/d{g_l, v_l, bias_l} of every Linear
 *                      (the double backward incl. Softplus'', skip concat, |.|, and the weight-norm VJP)
 * emap_render_bwd    : both, for the z_vals / udf / grad3 that emap_render_fwd returned
 *
 * Gradient outputs are written (accumulate = 0) or added to (accumulate = 1) and multiplied by grad_scale.
 * The pointer tables are HOST arrays of n_lin device pointers with the shapes of emap_pack_weights
 * (g [out,1], v [out,in], bias [out]); dg_host may be NULL when weight_norm = 0 (then dv = dL/dW). */
typedef struct EmapCompositeGrads {
    const float* d_edge;                         /* (N) or NULL                                  */
    const float* d_depth;                        /* (N) or NULL (gradient of the depth_scale'd depth) */
    const float* d_gradient_error;               /* device scalar or NULL                        */
    const float* d_gradient_error_near_surface;  /* device scalar or NULL                        */
    const float* scalars;                        /* EmapCompositeOut.scalars of the forward ([4], [6] = the eikonal mask sums;
                                                    data-parallel callers put the GLOBAL sums there) */
    float* d_variance;                           /* out (1) or NULL: dL/d SingleVarianceNetwork.variance */
    float* d_beta;                               /* out (1) or NULL: dL/d BetaNetwork.beta       */
    float* d_gamma;                              /* out (1) or NULL: dL/d BetaNetwork.gamma      */
    float grad_scale;
    int32_t accumulate;
    float* zero_tail;                            /* ABI v8, optional: n_zero_tail floats that the launch clears when accumulate == 0 - the gradient */
    int64_t n_zero_tail;                         /* slots of scalar parameters the path does not reach, so that a caller's flat gradient needs no memset */
} EmapCompositeGrads;

typedef struct EmapParamGrads {
    const float* const* g_host;   /* parameters (for the weight-norm VJP) */
    const float* const* v_host;
    float* const* dg_host;        /* outputs */
    float* const* dv_host;
    float* const* db_host;
    int32_t weight_norm;          /* 1: W = g v/||v|| (udf_model.py:73-74); 0: W = v */
    int32_t accumulate;
    float grad_scale;
    int32_t reserved;
} EmapParamGrads;

/* partials: (N,4) scratch.  With variance_dev == NULL in `p` the three scalar outputs are dL/d{inv_s, beta, gamma}. */
int emap_composite_bwd(const float* rays_o, const float* rays_d, const float* z, const float* udf, const float* grad3,
                       const float* depth_scale, int N, int S, const float* sample_dist_dev, const EmapRenderParams* p,
                       const EmapCompositeGrads* g, float* d_udf, float* d_grad3, float* partials, void* stream);

/* Workspace of the two backward entry points: *_workspace_bytes() is the PREFERRED size - the stash of the weight-gradient operands,
 * ~17 KiB per point, for up to 524 288 points per launch of the sweep (8.8 GB; sized for 288 GB of HBM).  A caller may pass less: the
 * backward then runs in more, smaller chunks (same results to the rounding of the partial sums; never fewer than 8 192 points
 * per chunk, else EMAP_E_WORKSPACE with the minimum in emap_last_error()). */
int emap_udf_vjp_workspace_bytes(const EmapNetConfig* cfg, int prec, int64_t P, size_t* bytes);
int emap_udf_vjp(const EmapNetConfig* cfg, const void* packed, int prec, const float* x, int64_t P, const float* d_udf,
                 const float* d_grad3, const EmapParamGrads* out, void* workspace, size_t workspace_bytes,
                 int32_t* err_flags, void* stream);

/* z_vals (N,S), udf (N,S), grad3 (N,S,3), sample_dist_dev: what emap_render_fwd produced (sample_dist_dev = the first
 * float of its workspace).  workspace: emap_render_bwd_workspace_bytes(). */
int emap_render_bwd_workspace_bytes(const EmapNetConfig* cfg, int prec, const EmapRenderParams* p, size_t* bytes);
int emap_render_bwd(const EmapNetConfig* cfg, const void* packed, int prec, const EmapRenderParams* p, const float* rays_o,
                    const float* rays_d, const float* depth_scale, const float* z_vals, const float* udf, const float* grad3,
                    const float* sample_dist_dev, const EmapCompositeGrads* g, const EmapParamGrads* out, void* workspace,
                    size_t workspace_bytes, int32_t* err_flags, void* stream);
/* The same call in two stages, for a data-parallel step whose ranks must agree on the fp16 range scale of the MLP backward
 * (the scale is a power of two taken from max|dL/dudf|, max|dL/dgrad| over the launch; with rank-local maxima the step would
 * depend on how the rays are sharded - the loop being sharded is src/runner/runner_udf.py:90-168):
 *   stages & 1 : the compositing adjoint; leaves dL/dudf, dL/dgrad3 and the two maxima (two floats >= 0) in the workspace
 *   stages & 2 : MLP double backward + weight gradients, reading the maxima found in the workspace
 * Between the stages the caller may max-reduce the two floats at byte offset emap_render_bwd_absmax_offset() of the
 * workspace across ranks (stream order is the only synchronisation needed).  stages == 3 is emap_render_bwd. */
int emap_render_bwd_absmax_offset(const EmapNetConfig* cfg, int prec, const EmapRenderParams* p, size_t* offset);
int emap_render_bwd_staged(const EmapNetConfig* cfg, const void* packed, int prec, const EmapRenderParams* p, const float* rays_o,
                           const float* rays_d, const float* depth_scale, const float* z_vals, const float* udf, const float* grad3,
                           const float* sample_dist_dev, const EmapCompositeGrads* g, const EmapParamGrads* out, void* workspace,
                           size_t workspace_bytes, int32_t* err_flags, void* stream, int stages);

/* ---- on-device ray / pixel sampler (SURVEY par. 8 f3) ---------------------------------------------
 * Replaces Dataset.gen_random_rays_patches_at (src/dataset/dataset.py:222-307): pixel draw (uniform, or 50 % uniform + 50 %
 * edge-weighted when importance != 0, :236-263), edge look-up (:270), p = K^-1 [x,y,1] (:272-277), rays_v = R p/|p| (:279-283),
 * rays_o = t (:284-286), depth_scale (:280), ndc uv (:265-267) - one launch, everything resident on the device.
 * The dataset is uploaded once: edges (n_images,H,W) in [0,1] (cv.imread(...,0)/255, :133-135), the inverse intrinsics'
 * upper 3x3 (:119) and the 4x4 camera-to-world poses (:86,121), row-major.  For importance sampling additionally, per image:
 * pixel_order (H*W int32: row-major pixel ids, those with edge > 0.1 first), n_edge (their count) and density (the image's
 * mean edge value).
 * Random numbers: Philox4x32-10 with key `seed`, stream = step, index = ray.  step = *counter_dev when counter_dev != NULL
 * (the kernel chain then increments it: the sampler can live inside a captured graph), else `offset`.
 * img_idx < 0 selects image_perm[step % n_images] (or step % n_images when image_perm is NULL) on the device.
 * pixels_in ((N,2) int64 x,y; may be NULL) bypasses the draw: the deterministic part, used by the parity tests.
 * Any output pointer may be NULL. */
typedef struct EmapRayDataset {
    const float* edges;          /* (n_images,H,W)   */
    const int32_t* pixel_order;  /* (n_images,H*W) or NULL */
    const int32_t* n_edge;       /* (n_images) or NULL     */
    const float* density;        /* (n_images) or NULL     */
    const float* kinv;           /* (n_images,3,3)   */
    const float* pose;           /* (n_images,4,4)   */
    const int32_t* image_perm;   /* (n_images) or NULL (runner_udf.py:79-82 image_perm) */
    int32_t n_images, H, W, reserved;
} EmapRayDataset;

typedef struct EmapRayBatch {
    float* rays_o;       /* (N,3) */
    float* rays_v;       /* (N,3) */
    float* edge;         /* (N)   */
    float* depth_scale;  /* (N)   */
    float* ndc_uv;       /* (N,2) */
    float* p_cam;        /* (N,3)  rays_norm_XYZ_cam */
    int64_t* pixels;     /* (N,2)  x,y */
    int32_t* img_idx;    /* (1)    the image the batch came from */
    float* t_rand;       /* (N)    ABI v8: render()'s per-ray jitter U(-0.5, 0.5) (udf_renderer_blending.py:719: torch.rand([N,1]) - 0.5 on the host
                                   generator there), from the same Philox draw as the ray's pixel */
} EmapRayBatch;

int emap_sample_rays(const EmapRayDataset* ds, int img_idx, int batch, int importance, uint64_t seed, uint64_t offset,
                     uint64_t* counter_dev, const int64_t* pixels_in, const EmapRayBatch* out, void* stream);

/* ---- scalar tail of a training step (SURVEY par. 8 a15; src/runner/runner_udf.py:124-168, src/models/loss.py:14-17,
 *      src/runner/runner_base.py:110-117) ---------------------------------------------------------------
 * emap_train_stats : the rank-local statistics of a step and dL/d(edge) of EdgeLoss("mse") * edge_weight:
 *                      d_edge[i] = (edge[i] - true_edge[i]) * d_scale          (d_scale = 2 * edge_weight / N_global; may be NULL)
 *                      stats5    = [sum(relax), sum(near), sum(relax*err), sum(near*err), sum((edge - true_edge)^2)]
 *                    with the eikonal sums taken from emap_render_fwd's `scalars` - the 5 numbers a data-parallel step sums
 * emap_train_loss  : out2 = [loss, edge_loss] of runner_udf.py:124-159 from (globally summed) stats5:
 *                      edge_loss = stats5[4] * w_over_n  (edge_weight / N_global),
 *                      loss = edge_loss + igr * stats5[2] / (stats5[0] + 1e-5) + igr_ns * stats5[3] / (stats5[1] + 1e-5)
 * emap_adam_step   : torch.optim.Adam([{geo params, lr_geo}, {the rest}], lr) (runner_base.py:110-117; betas / eps as given,
 *                    no weight decay, no amsgrad) on flat buffers: elements [0, n_geo) use lr_geo, [n_geo, n) use lr.
 *                    `step_dev` (float[1], starts at 0) counts the steps on the device and is incremented by the call.
 *                    ABI 7: beta1 / beta2 are DOUBLES - 1 - beta and beta^t are formed in double as torch forms them (1.0f - 0.999f is
 *                    4.7e-5 off 0.001: exp_avg_sq would not match a torch.optim.Adam checkpoint). */
int emap_train_stats(const float* edge, const float* true_edge, const float* scalars, int N, float d_scale, float* d_edge,
                     float* stats5, void* stream);
int emap_train_loss(const float* stats5, float w_over_n, float igr_weight, float igr_ns_weight, float* out2, void* stream);
int emap_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* step_dev, int64_t n, int64_t n_geo,
                   float lr_geo, float lr, double beta1, double beta2, float eps, void* stream);
/* emap_adam_step_masked : the same with per-element state for the tail [n_geo, n) (the scalars variance / beta / gamma, which the runner
 *                    freezes and un-freezes: runner_udf.py:141-154): tail_mask[j] = 1 trainable / 0 frozen (torch.optim.Adam skips a
 *                    parameter without gradient: no update, no state change), tail_step[j] = that element's own step count
 *                    (bias correction restarts when a parameter is un-frozen, as torch's per-parameter `step`).  Both device
 *                    arrays of n - n_geo floats: no host index tensors, graph-capturable (ABI 6). */
int emap_adam_step_masked(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* step_dev, int64_t n, int64_t n_geo,
                          float lr_geo, float lr, double beta1, double beta2, float eps, const float* tail_mask, float* tail_step, void* stream);

/* ---- dense-grid extraction (SURVEY par. 8 f2) ----------------------------------------------------
 * emap_null_direction : `_, _, vh = torch.linalg.svd(grad_ld); F.normalize(vh[:, -1, :])` of get_udf_normals_grid /
 *                       get_udf_normals_slow (src/edge_extraction/extract_pointcloud.py:86-88, 177-179): per point the unit
 *                       right singular vector of the smallest singular value of its (k x 3) gradient matrix
 *                       grads (n, k, 3) -> dir (n, 3); the sign is arbitrary, as in the reference */
int emap_null_direction(const float* grads, int64_t n, int k, float* dir, void* stream);

/* ---- measurement ----------------------------------------------------------------------------
 * While enabled, emap_render_fwd / emap_render_bwd bracket their dominant kernels with hipEvents on the launch
 * stream; emap_profile_read[_kernel] (after the caller synchronised) returns the summed duration and the number of
 * launches.  Used by bench.py for the roofline figure (in a loop separate from the headline timing). */
int emap_profile_enable(int on);
int emap_profile_read(float* total_ms_host, int* launches_host);
/* which: 0 = final value+gradient MLP pass of emap_render_fwd (same as emap_profile_read), 1 = udf_mlp_vjp sweep,
 * 2 = weight-gradient GEMMs (both inside emap_render_bwd / emap_udf_vjp) */
int emap_profile_read_kernel(int which, float* total_ms_host, int* launches_host);
/* Shader clock (MHz) during the LAST launch, while profiling was enabled, of which = 0: the reverse-sweep value+gradient kernel,
 * 1: the udf_mlp_vjp sweep - s_memtime (shader cycles) over s_memrealtime (100 MHz) between the entry and the exit of workgroup 0.
 * 0 if no such launch was recorded.  (The dominant kernel runs at the package power cap on real data: DESIGN.md par. 5.) */
int emap_profile_read_clock(int which, float* mhz_host);

/* ---- one-shot peer-to-peer all-reduce of the gradient bucket (SURVEY par. 8e "xGMI note"; csrc/allreduce.hip; ABI 7) ----------------
 * The data-parallel step's single collective (emap_amd/parallel.py; the reference has none: EMAP is single-GPU, runner_udf.py:166-168)
 * is a 1.85 MB SUM: latency-bound.  Instead of RCCL's ring (2 (R-1) dependent steps) every rank reads every other rank's bucket
 * directly over the xGMI mesh and sums in rank order (bit-identical on all ranks), one synchronisation, one launch, no host work.
 *   emap_ar_local_bytes   : size of a rank's REGION for buckets of up to n_floats: [256 B control][2 x staging]
 *   emap_ar_alloc         : allocates + zeroes a region (uncached device memory) and returns its 64-byte hipIpcMemHandle; the caller
 *                           exchanges the handles of all ranks on the host (torch.distributed all_gather_object, any backend)
 *   emap_ar_open / _close : map / unmap a peer's region from its handle (hipIpcOpenMemHandle; needs HSA_ENABLE_IPC_MODE_LEGACY=0 here)
 *   emap_ar_allreduce_sum : data[0..n) <- sum over ranks, in place; regions_host = HOST array of `world` device pointers, entry r = the
 *                           region of rank r as mapped into THIS process (entry `rank` = the own region).  Every rank calls it in lock
 *                           step with the same n.  Graph-capturable (the step counter lives in the region).  A peer that does not
 *                           show up within the time-out sets the sticky error word AND makes this rank's result NaN (ABI 9: never a
 *                           partial sum - the failure reaches every consumer of the bucket) instead of hanging the device.
 *   emap_ar_set_timeout_ms: that time-out (ABI 9; default 10 000 ms, x6 for a region's first two launches: lazy code-object loads,
 *                           workspace allocations and data-loader stalls skew the ranks most at start-up); process-wide, read at launch.
 *   emap_ar_error         : host read of that error word (synchronises).
 *   emap_ar_alloc FAILS when fine-grained memory is not available (ABI 9; no coarse-grained fall-back: use RCCL then). */
int emap_ar_local_bytes(int64_t n_floats, size_t* bytes);
int emap_ar_alloc(size_t bytes, void** region, void* ipc_handle64);
int emap_ar_open(const void* ipc_handle64, void** peer_region);
int emap_ar_close(void* peer_region);
int emap_ar_free(void* region);
int emap_ar_allreduce_sum(float* data, int64_t n, int rank, int world, void* const* regions_host, size_t region_bytes, void* stream);
int emap_ar_error(void* region, int* error_host);
int emap_ar_set_timeout_ms(int64_t ms);

/* host-only: torch.linspace(start, end, steps) in fp32, the grid of sample_pdf's u / the coarse z_vals */
void emap_linspace_host(float start, float end, int steps, float* out_host);

#ifdef __cplusplus
}
#endif
#endif /* EMAP_HIP_H */
