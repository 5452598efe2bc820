// sampler.hip - per-ray kernels of the EMAP renderer for gfx950: coarse z_vals, occlusion-aware
// up-sampling (up_sample_unbias + sample_pdf), sorted merge (cat_z_vals) and the compositing tail of
// render_core.  HBM-light, latency-bound work: one 64-lane wavefront per ray, the ray's samples live
// in LDS, prefix products / sums are wave scans (no torch.cumprod / sort / searchsorted launches, no
// host synchronisation).
//
// Reference behaviour (cvg/EMAP, src/models/udf_renderer_blending.py):
//   sample_pdf :69-109   up_sample_unbias :228-353   cat_z_vals :355-377   sdf2alpha :379-416
//   udf2logistic :155-170   render :700-720 (coarse z)   render_core :435-455,463-677
//
// Numerics: every elementwise expression is evaluated in fp32 in the reference's operation order
// with separately rounded mul/add (no fma contraction) so that, given identical fp32 inputs, the
// integer outputs (searchsorted indices, merge permutation) are bit-exact; scans accumulate in
// fp64 and round each output to fp32, which is what torch's CPU cumsum/cumprod do
// (acc_type<float, /*is_cuda=*/false> == double).
#include "emap_common.h"

namespace emap {

#include "sampler_dev.inc"

static float linspace_at_host(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    volatile float step = (end - start) / (float)(steps - 1);
    volatile float a = step * (float)i;
    volatile float b = step * (float)(steps - i - 1);
    return (i < steps / 2) ? start + a : end - b;
}

__global__ __launch_bounds__(64) void sample_pdf_kernel(const float* bins, const float* weights, int N, int n, int m,
                                                        float* samples, int64_t* inds, int32_t* err, const float* u) {
    __shared__ float s_bins[MAXS], s_w[MAXS], s_pdf[MAXS], s_cdf[MAXS + 1];
    const int ray = blockIdx.x, lane = threadIdx.x;
    for (int e = lane; e < n; e += 64) s_bins[e] = bins[(size_t)ray * n + e];
    for (int e = lane; e < n - 1; e += 64) s_w[e] = weights[(size_t)ray * (n - 1) + e];
    __syncthreads();
    sample_pdf_wave(s_bins, s_w, s_pdf, s_cdf, n, m, lane, samples + (size_t)ray * m, inds ? inds + (size_t)ray * m : nullptr, err,
                    u ? u + (size_t)ray * m : nullptr);
}

__global__ __launch_bounds__(64) void upsample_kernel(const float* rays_o, const float* rays_d, const float* z,
                                                      const float* udf, int N, int n, int m, const float* sample_dist,
                                                      float inv_s, float beta, float gamma, float* z_new, int64_t* inds,
                                                      int32_t* err) {
    __shared__ float s_z[MAXS], s_u[MAXS];
    __shared__ UpsampleScratch w;
    const int ray = blockIdx.x, lane = threadIdx.x;
    const float ox = rays_o[3 * ray], oy = rays_o[3 * ray + 1], oz = rays_o[3 * ray + 2];
    const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
    const float sd = *sample_dist;
    for (int e = lane; e < n; e += 64) {
        s_z[e] = z[(size_t)ray * n + e];
        s_u[e] = udf[(size_t)ray * n + e];
    }
    __syncthreads();
    upsample_body(ox, oy, oz, dx, dy, dz, sd, s_z, s_u, w, n, m, inv_s, beta, gamma, lane, z_new + (size_t)ray * m,
                  inds ? inds + (size_t)ray * m : nullptr, err);
}

// ---------------------------------------------------------------------------------------------
// cat_z_vals: merge two sorted lists (stable: old samples first on ties), gather udf
// (udf_renderer_blending.py:361-375)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void merge_kernel(const float* z, const float* z_new, const float* udf,
                                                   const float* udf_new, int N, int n, int m, float* z_out,
                                                   float* udf_out, int64_t* perm) {
    __shared__ float s_z[MAXS], s_n[MAXS];
    const int ray = blockIdx.x, lane = threadIdx.x;
    for (int e = lane; e < n; e += 64) s_z[e] = z[(size_t)ray * n + e];
    for (int e = lane; e < m; e += 64) s_n[e] = z_new[(size_t)ray * m + e];
    __syncthreads();
    const size_t ob = (size_t)ray * (n + m);
    for (int e = lane; e < n + m; e += 64) {
        float v;
        const int rank = merge_rank(s_z, s_n, n, m, e, v);
        z_out[ob + rank] = v;
        if (perm) perm[ob + rank] = e;
        if (udf_out) udf_out[ob + rank] = (e < n) ? udf[(size_t)ray * n + e] : udf_new[(size_t)ray * m + (e - n)];
    }
}

// ---------------------------------------------------------------------------------------------
// One step of importance_sample (udf_renderer_blending.py:824-839) as ONE launch per ray wave:
//   [COARSE: the coarse z_vals of render() :705-720 computed in place (and sample_dist, :704)]
//   [MERGE : cat_z_vals of the PREVIOUS step (:355-377): merge z/udf with the previous step's new samples]
//   up_sample_unbias of this step (:228-353) on the merged lists, which never leave LDS
//   [TAIL  : cat_z_vals(last=True) of this step: the final z_vals]
// The kernels the C ABI exposes one by one (emap_upsample_step, emap_merge_sorted) share the bodies.
// ---------------------------------------------------------------------------------------------

template <bool COARSE, bool MERGE, bool TAIL>
__global__ __launch_bounds__(64) void sampler_step_kernel(const StepArgs a) {
    __shared__ float s_z[MAXS], s_u[MAXS], s_n[MAXS];
    __shared__ UpsampleScratch w;
    const int ray = blockIdx.x, lane = threadIdx.x;
    const float ox = a.rays_o[3 * ray], oy = a.rays_o[3 * ray + 1], oz = a.rays_o[3 * ray + 2];
    const float dx = a.rays_d[3 * ray], dy = a.rays_d[3 * ray + 1], dz = a.rays_d[3 * ray + 2];
    int n = a.n;
    const int m = a.m;
    float sd;
    if constexpr (COARSE) {
        // sample_dist = ((far - near) / n_samples).mean() (:704): every wave forms the same fp64 sum in the same order
        double sum = 0.0;
#pragma unroll 8      // loads of 8 iterations in flight (the adds stay in order): at 4096 rays this loop was 15 of the step's 25 us
        for (int r = lane; r < a.N; r += 64) sum += (double)FDIV(FSUB(a.far[r], a.near[r]), (float)n);
        sd = (float)(wave_sum_d(sum) / (double)a.N);
        if (ray == 0 && lane == 0) *a.sample_dist = sd;
        for (int e = lane; e < n; e += 64) {
            const float lin = linspace_at(0.0f, 1.0f, n, e);
            float v = FADD(a.near[ray], FMUL(FSUB(a.far[ray], a.near[ray]), lin));                   // :707
            if (a.t_rand) v = FADD(v, FDIV(FMUL(a.t_rand[ray], 2.0f), (float)n));                    // :720
            s_z[e] = v;
            s_u[e] = a.udf[(size_t)ray * n + e];
            a.z_merged[(size_t)ray * n + e] = v;
        }
    } else {
        sd = *a.sample_dist;
        if constexpr (MERGE) {
            // all four global reads of the step are requested before anything waits on them: one memory round trip, not two (the
            // kernel is one latency chain per ray - 23 k cycles, of which this block was 5.4-6.8 k)
            float uu[(MAXS + 63) / 64];
#pragma unroll
            for (int i = 0; i < (MAXS + 63) / 64; ++i) {
                const int e = lane + 64 * i;
                uu[i] = (e < n) ? a.udf[(size_t)ray * n + e] : ((e < n + m) ? a.udf_prev[(size_t)ray * m + (e - n)] : 0.f);
            }
            for (int e = lane; e < n; e += 64) w.a[e] = a.z[(size_t)ray * n + e];       // scratch as staging: old z
            for (int e = lane; e < m; e += 64) s_n[e] = a.z_prev[(size_t)ray * m + e];
            __syncthreads();
            const size_t ob = (size_t)ray * (n + m);
#pragma unroll
            for (int i = 0; i < (MAXS + 63) / 64; ++i) {
                const int e = lane + 64 * i;
                if (e < n + m) {
                    float v;
                    const int rank = merge_rank(w.a, s_n, n, m, e, v);
                    const float u = uu[i];
                    s_z[rank] = v; s_u[rank] = u;
                    a.z_merged[ob + rank] = v; a.udf_merged[ob + rank] = u;
                }
            }
            n += m;
        } else {
            for (int e = lane; e < n; e += 64) { s_z[e] = a.z[(size_t)ray * n + e]; s_u[e] = a.udf[(size_t)ray * n + e]; }
        }
    }
    __syncthreads();
    upsample_body(ox, oy, oz, dx, dy, dz, sd, s_z, s_u, w, n, m, a.inv_s, a.beta, a.gamma, lane, s_n, nullptr, a.err);
    __syncthreads();
    for (int e = lane; e < m; e += 64) a.z_new[(size_t)ray * m + e] = s_n[e];
    if constexpr (TAIL) {
        const size_t ob = (size_t)ray * (n + m);
        for (int e = lane; e < n + m; e += 64) {
            float v;
            const int rank = merge_rank(s_z, s_n, n, m, e, v);
            a.z_final[ob + rank] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// coarse z_vals + sample_dist (render() :700-720)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void coarse_z_kernel(const float* near, const float* far, const float* t_rand, int N,
                                                       int n_samples, float* z, float* sample_dist) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long long)N * n_samples) {
        const int ray = (int)(i / n_samples), k = (int)(i - (long long)ray * n_samples);
        const float lin = linspace_at(0.0f, 1.0f, n_samples, k);
        float v = FADD(near[ray], FMUL(FSUB(far[ray], near[ray]), lin));                           // :707
        if (t_rand) v = FADD(v, FDIV(FMUL(t_rand[ray], 2.0f), (float)n_samples));                  // :720
        z[i] = v;
    }
    if (blockIdx.x == 0) {  // sample_dist = ((far - near) / n_samples).mean()                          :704
        __shared__ double red[4];
        double s = 0.0;
        for (int r = threadIdx.x; r < N; r += 256) s += (double)FDIV(FSUB(far[r], near[r]), (float)n_samples);
        s = wave_sum_d(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) *sample_dist = (float)((red[0] + red[1] + red[2] + red[3]) / (double)N);
    }
}

// ---------------------------------------------------------------------------------------------
// render_core tail (udf_renderer_blending.py:435-455,463-677)
// ---------------------------------------------------------------------------------------------
#include "composite_dev.inc"     // CompositeArgs + composite_ray<C, COH>: the per-ray body, shared with udf_mlp_rev32.inc's fused tail


template <int C>
__global__ __launch_bounds__(64) void composite_kernel(const CompositeArgs a) {
    composite_ray<C, false>(a, blockIdx.x, threadIdx.x);
}

// deterministic cross-ray reduction of the eikonal terms (:618-625) and sparse_error (:642-644)
__global__ __launch_bounds__(256) void composite_reduce_kernel(const float* partials, int N, float* scalars, int32_t* err,
                                                               const CompositeArgs a) {
    __shared__ double red[4][5];
    composite_reduce_body<false>(partials, N, scalars, err, a, threadIdx.x, 256, red);
}

// ---------------------------------------------------------------------------------------------
// composite_bwd: reverse of render_core's tail (udf_renderer_blending.py:463-625 under autograd) - SURVEY par. 8 f1
// ---------------------------------------------------------------------------------------------
// Given dL/d{edge, depth} per ray and dL/d{gradient_error, gradient_error_near_surface}, produces dL/dudf (N,S),
// dL/d(grad_x udf) (N,S,3) and per-ray partial sums of dL/d{inv_s, beta, gamma}.  The derivation (and its check against
// torch.autograd through the oracle) is oracle/vjp_mirror.py:composite_bwd / tests/test_vjp_math.py.  One wave per ray;
// the forward quantities are recomputed with the forward kernel's own expressions so that every clip / mask decision is
// the one the forward took; the two cumprod adjoints are exclusive suffix sums (fp64 wave scans).
//
// Round 5: the ray lives in REGISTERS - lane l holds the C = 1, 2 or 4 consecutive samples [l C, (l+1) C) (the chunking wave_scan uses,
// so the two prefix products are bit-identical to the forward kernel's), every input is fetched by one burst of loads at the top (one
// memory round trip instead of one per pass and loop iteration), neighbours (z, true_cos of sample e+1) come over the DPP network, and
// the sigmoids / exponentials of the forward recomputation are kept for the adjoint instead of being evaluated a second time.  No LDS,
// no barriers.  The round-4 kernel (13 LDS arrays per ray, strided passes) took 24 us at 512 rays and 108 us at 4096.
struct CompositeBwdArgs {
    const float *rays_o, *rays_d, *z, *udf, *grad, *depth_scale, *sample_dist;
    int N, S;
    float inv_s, beta, gamma, car;
    int anneal;
    float flip_sat, near_surface, background;
    int has_bg;
    const float *var_p, *beta_p, *gamma_p;
    float beta_min;
    const float *d_edge, *d_depth;      // (N) or null
    const float *d_ge, *d_ge_ns;        // device scalars or null
    const float* scalars;               // the forward's scalars: [4] = sum(relax), [6] = sum(near)
    float *d_udf, *d_grad;              // (N,S), (N,S,3)
    float* partials;                    // (N,4): per-ray d_inv_s, d_beta, d_gamma
    uint32_t* absmax;                   // [2]: max|d_udf|, max|d_grad| of the launch, written by the reduce kernel; may be null
    float* raymax;                      // (N,2): the per-ray maxima behind them (null iff absmax is)
    float* zero_tail;                   // EmapCompositeGrads.zero_tail / n_zero_tail (cleared by the reduce kernel), or null
    long long n_zero_tail;
};

// lane l <- lane 63 - l
__device__ __forceinline__ double lane_reverse_d(double v, int lane) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int idx = (63 - lane) << 2;
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(idx, (int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_ds_bpermute(idx, (int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// exclusive suffix sum over the ray: out[e] = sum_{k>e} in[k]; the lanes are reversed around the forward DPP scan
template <int C>
__device__ __forceinline__ void ray_suffix_sum(const float (&in)[C], const bool (&ok)[C], float (&out)[C], int lane) {
    double loc = 0.0;
#pragma unroll
    for (int i = C - 1; i >= 0; --i) if (ok[i]) loc += (double)in[i];
    const double inc = wave_scan_incl_d<false>(lane_reverse_d(loc, lane));
    double run = lane_reverse_d(dpp_d<0x138, 0xf>(0.0, inc), lane);
#pragma unroll
    for (int i = C - 1; i >= 0; --i) { out[i] = (float)run; if (ok[i]) run += (double)in[i]; }
}

// sdf2alpha (the forward's expressions, bit for bit) that also hands back what its adjoint needs
struct Sdf2AlphaKeep { float val, pc, nc, den, en, ep; };
__device__ __forceinline__ float sdf2alpha_keep(float sdf, float true_cos, float dists, float inv_s, bool anneal, float car, Sdf2AlphaKeep& k) {
    float iter_cos = true_cos;
    if (anneal) {
        const float a = FMUL(relu_(FADD(FMUL(-true_cos, 0.5f), 0.5f)), FSUB(1.0f, car));
        const float b = FMUL(relu_(-true_cos), car);
        iter_cos = -FADD(a, b);
    }
    const float h = FMUL(FMUL(iter_cos, dists), 0.5f);
    k.en = FADD(sdf, h);
    k.ep = FSUB(sdf, h);
    k.pc = sigmoidf_(FMUL(k.ep, inv_s));
    k.nc = sigmoidf_(FMUL(k.en, inv_s));
    k.den = FADD(k.pc, 1e-5f);
    k.val = FDIV(FADD(FSUB(k.pc, k.nc), 1e-5f), k.den);
    return clipf(k.val, 0.0f, 1.0f);
}
// backward of sdf2alpha(sdf, -tabs, dists, inv_s) for an upstream gradient dval on its clipped output
__device__ __forceinline__ void sdf2alpha_bwd(const Sdf2AlphaKeep& k, float tabs, float dists, float inv_s, bool anneal, float car, float dval,
                                              float& d_sdf, float& d_tabs, float& d_inv_s) {
    const float dic = anneal ? -(0.5f * (1.0f - car) + ((tabs > 0.f) ? car : 0.f)) : -1.0f;
    const float rden = __builtin_amdgcn_rcpf(k.den);
    const float dv = (k.val >= 0.f && k.val <= 1.f) ? dval : 0.f;
    const float dnc = -dv * rden, dpc = dv * k.nc * rden * rden;
    const float gp = dpc * k.pc * (1.0f - k.pc), gn = dnc * k.nc * (1.0f - k.nc);
    d_inv_s = gp * k.ep + gn * k.en;
    d_sdf = (gp + gn) * inv_s;
    d_tabs = (gn - gp) * inv_s * dists * 0.5f * dic;
}

template <int C>
__global__ __launch_bounds__(64) void composite_bwd_kernel(const CompositeBwdArgs a) {
    const int ray = blockIdx.x, lane = threadIdx.x, S = a.S;
    const size_t rb = (size_t)ray * S;
    // one burst: the ray's samples (clamped to the last one past the end), then the per-ray and per-launch scalars
    float z[C + 1], u[C], gx[C], gy[C], gz[C], tc[C + 1];
    bool ok[C], last[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const int e = lane * C + i;
        ok[i] = e < S; last[i] = !(e < S - 1);
        const size_t q = rb + (ok[i] ? e : S - 1);
        z[i] = a.z[q]; u[i] = a.udf[q];
        gx[i] = a.grad[3 * q]; gy[i] = a.grad[3 * q + 1]; gz[i] = a.grad[3 * q + 2];
    }
    const float ox = a.rays_o[3 * ray], oy = a.rays_o[3 * ray + 1], oz = a.rays_o[3 * ray + 2];
    const float dx = a.rays_d[3 * ray], dy = a.rays_d[3 * ray + 1], dz = a.rays_d[3 * ray + 2];
    const float sd = *a.sample_dist;
    float inv_s_ = a.inv_s, beta_ = a.beta, gamma_ = a.gamma;
    if (a.var_p) {
        inv_s_ = clipf(expf(FMUL(a.var_p[0], 10.0f)), 1e-6f, 1e6f);
        beta_ = clipf(clipf(expf(FMUL(a.beta_p[0], 10.0f)), 0.0f, FDIV(1.0f, a.beta_min)), 1e-6f, 1e6f);
        gamma_ = clipf(expf(FMUL(a.gamma_p[0], 10.0f)), 1e-6f, 1e6f);
    }
    const float g_edge = a.d_edge ? a.d_edge[ray] * (a.has_bg ? (1.0f - a.background) : 1.0f) : 0.f;
    const float g_depth = a.d_depth ? a.d_depth[ray] * (a.depth_scale ? a.depth_scale[ray] : 1.0f) : 0.f;
    const float c_ge = a.d_ge ? a.d_ge[0] / (a.scalars[4] + 1e-5f) : 0.f;
    const float c_ns = a.d_ge_ns ? a.d_ge_ns[0] / (a.scalars[6] + 1e-5f) : 0.f;
    const bool anneal = a.anneal != 0;

#pragma unroll
    for (int i = 0; i < C; ++i) tc[i] = FADD(FADD(FMUL(dx, gx[i]), FMUL(dy, gy[i])), FMUL(dz, gz[i]));
    z[C] = dpp_next_f(0.f, z[0]);       // sample e+1 of a lane's last sample is the next lane's first
    tc[C] = dpp_next_f(0.f, tc[0]);
    float dists[C], E[C], opE[C], raw[C], eq[C], ain[C], av[C], vpr[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        dists[i] = last[i] ? sd : FSUB(z[i + 1], z[i]);
        E[i] = expf(FMUL(-beta_, u[i]));                        // udf2logistic1(u, beta) with its pieces kept
        opE[i] = FADD(1.0f, E[i]);
        raw[i] = FDIV(FMUL(beta_, E[i]), FMUL(opE[i], opE[i]));
        eq[i] = expf(FMUL(FMUL(-relu_(raw[i]), gamma_), dists[i]));            // 1 - alpha_occ up to rounding
        const float vis_mask = last[i] ? 1.0f : ((tc[i + 1] < 0.01f) ? 1.0f : 0.0f);
        const float occ = FSUB(1.0f, eq[i]);
        ain[i] = FADD(FSUB(1.0f, occ), FMUL(a.flip_sat, vis_mask));
        av[i] = FADD(clipf(ain[i], 0.0f, 1.0f), 1e-7f);
    }
    ray_prefix_prod<C>(av, ok, vpr);           // raw (unclipped) visibility product
    Sdf2AlphaKeep kp[C], km[C];
    float ap[C], am[C], vp[C], alpha[C], om[C], T[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        vp[i] = clipf(vpr[i], 0.0f, 1.0f);
        const float tcn = -fabsf(tc[i]);
        ap[i] = sdf2alpha_keep(u[i], tcn, dists[i], inv_s_, anneal, a.car, kp[i]);
        am[i] = sdf2alpha_keep(-u[i], tcn, dists[i], inv_s_, anneal, a.car, km[i]);
        alpha[i] = FADD(FMUL(ap[i], vp[i]), FMUL(am[i], FSUB(1.0f, vp[i])));
        om[i] = FADD(FSUB(1.0f, alpha[i]), 1e-7f);
    }
    ray_prefix_prod<C>(om, ok, T);             // transmittance
    float mid[C], dal[C], x[C], suf[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        mid[i] = FADD(z[i], FMUL(dists[i], 0.5f));
        dal[i] = g_edge + g_depth * mid[i];                     // dL/dw_e for now
        x[i] = dal[i] * alpha[i] * T[i];                        // dw_e * w_e
    }
    ray_suffix_sum<C>(x, ok, suf, lane);
#pragma unroll
    for (int i = 0; i < C; ++i) {
        dal[i] = dal[i] * T[i] - suf[i] * __builtin_amdgcn_rcpf(om[i]);
        const float dvp = (vpr[i] >= 0.f && vpr[i] <= 1.f) ? dal[i] * (ap[i] - am[i]) : 0.f;
        x[i] = dvp * vpr[i];
    }
    ray_suffix_sum<C>(x, ok, suf, lane);
    double p_is = 0.0, p_beta = 0.0, p_gamma = 0.0;
    float mx_u = 0.f, mx_g = 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        // occlusion branch: a_i = clip(1 - occ + fs*vm) + 1e-7
        const float da = suf[i] * __builtin_amdgcn_rcpf(av[i]);
        const float docc = (ain[i] >= 0.f && ain[i] <= 1.f) ? -da : 0.f;
        const float dq = docc * eq[i];
        const float r1 = __builtin_amdgcn_rcpf(opE[i]), r2 = r1 * r1;
        const float draw = (raw[i] > 0.f) ? dq * gamma_ * dists[i] : 0.f;
        const float fE = (1.0f - E[i]) * r2 * r1;
        float du = draw * (-beta_ * beta_ * E[i] * fE);
        const float pb = draw * (E[i] * r2 - beta_ * u[i] * E[i] * fE), pg = dq * relu_(raw[i]) * dists[i];
        // alpha branch
        const float tabs = fabsf(tc[i]);
        float s1, t1, i1, s2, t2, i2;
        sdf2alpha_bwd(kp[i], tabs, dists[i], inv_s_, anneal, a.car, dal[i] * vp[i], s1, t1, i1);
        sdf2alpha_bwd(km[i], tabs, dists[i], inv_s_, anneal, a.car, dal[i] * (1.0f - vp[i]), s2, t2, i2);
        du += s1 - s2;
        const float dtc = (t1 + t2) * ((tc[i] > 0.f) ? 1.f : ((tc[i] < 0.f) ? -1.f : 0.f));
        // eikonal terms (:612-625), masks detached
        const float px = FADD(ox, FMUL(dx, mid[i])), py = FADD(oy, FMUL(dy, mid[i])), pz = FADD(oz, FMUL(dz, mid[i]));
        const float pn = sqrtf(FADD(FADD(FMUL(px, px), FMUL(py, py)), FMUL(pz, pz)));
        const float gm = sqrtf(FADD(FADD(FMUL(gx[i], gx[i]), FMUL(gy[i], gy[i])), FMUL(gz[i], gz[i])));
        const float relax = (pn < 2.4f) ? 1.0f : 0.0f, ns = (u[i] < a.near_surface) ? 1.0f : 0.0f;
        const float coef = (gm > 0.f) ? (c_ge * relax + c_ns * ns) * 2.0f * (gm - 1.0f) * __builtin_amdgcn_rcpf(gm) : 0.f;
        const float ogx = dtc * dx + coef * gx[i], ogy = dtc * dy + coef * gy[i], ogz = dtc * dz + coef * gz[i];
        if (ok[i]) {
            const size_t q = rb + lane * C + i;
            a.d_udf[q] = du;
            a.d_grad[3 * q] = ogx; a.d_grad[3 * q + 1] = ogy; a.d_grad[3 * q + 2] = ogz;
            p_is += (double)(i1 + i2); p_beta += (double)pb; p_gamma += (double)pg;
            const float au = fabsf(du), ag = fmaxf(fmaxf(fabsf(ogx), fabsf(ogy)), fabsf(ogz));
            mx_u = (au < 3.0e38f) ? fmaxf(mx_u, au) : mx_u;
            mx_g = (ag < 3.0e38f) ? fmaxf(mx_g, ag) : mx_g;
        }
    }
    p_is = wave_sum_d(p_is); p_beta = wave_sum_d(p_beta); p_gamma = wave_sum_d(p_gamma);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mx_u = fmaxf(mx_u, __shfl_xor(mx_u, off)); mx_g = fmaxf(mx_g, __shfl_xor(mx_g, off)); }
    if (lane == 0) {
        float* p = a.partials + (size_t)ray * 4;
        p[0] = (float)p_is; p[1] = (float)p_beta; p[2] = (float)p_gamma; p[3] = 0.f;
        // the launch maxima are formed by the reduce kernel: two atomicMax per ray on one cache line cost 10.5 ns EACH, serialised in L2 -
        // 11 of the kernel's 18 us at 512 rays, 85 of 100 us at 4096 (round 5, profiles/r05_composite_kernels.txt)
        if (a.raymax) { a.raymax[2 * (size_t)ray] = mx_u; a.raymax[2 * (size_t)ray + 1] = mx_g; }
    }
}

// deterministic cross-ray sum of the scalar-parameter gradients and the chain through x = exp(10 p).clip(...)
// (udf_model.py:226-227,259-263, udf_renderer_blending.py:466-472): d_param[0..2] = dL/d{variance, beta, gamma}
__global__ __launch_bounds__(256) void composite_bwd_reduce_kernel(const float* partials, int N, const CompositeBwdArgs a,
                                                                   float* d_variance, float* d_beta, float* d_gamma, float grad_scale,
                                                                   int accumulate) {
    __shared__ double red[4][3];
    __shared__ float redm[4][2];
    if (a.zero_tail && !accumulate)      // before the three scalar gradients are written: they may lie inside the range
        for (long long i = threadIdx.x; i < a.n_zero_tail; i += 256) a.zero_tail[i] = 0.f;
    __syncthreads();
    double v[3] = {0, 0, 0};
    float mu = 0.f, mg = 0.f;       // per-ray maxima are finite and >= 0
    for (int i = threadIdx.x; i < N; i += 256) {
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] += (double)partials[(size_t)i * 4 + k];
        if (a.raymax) { mu = fmaxf(mu, a.raymax[2 * (size_t)i]); mg = fmaxf(mg, a.raymax[2 * (size_t)i + 1]); }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = wave_sum_d(v[k]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mu = fmaxf(mu, __shfl_xor(mu, off)); mg = fmaxf(mg, __shfl_xor(mg, off)); }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) red[threadIdx.x >> 6][k] = v[k];
        redm[threadIdx.x >> 6][0] = mu; redm[threadIdx.x >> 6][1] = mg;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (a.absmax) {
            a.absmax[0] = __builtin_bit_cast(uint32_t, fmaxf(fmaxf(redm[0][0], redm[1][0]), fmaxf(redm[2][0], redm[3][0])));
            a.absmax[1] = __builtin_bit_cast(uint32_t, fmaxf(fmaxf(redm[0][1], redm[1][1]), fmaxf(redm[2][1], redm[3][1])));
        }
        double t[3];
        for (int k = 0; k < 3; ++k) t[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        float r_var = 0.f, r_beta = 0.f, r_gamma = 0.f;
        if (a.var_p) {
            const float xs = expf(FMUL(a.var_p[0], 10.0f));
            if (xs >= 1e-6f && xs <= 1e6f) r_var = (float)t[0] * 10.0f * xs;
            const float xb = expf(FMUL(a.beta_p[0], 10.0f)), hi = FDIV(1.0f, a.beta_min);
            if (xb >= 0.f && xb <= hi && xb >= 1e-6f && xb <= 1e6f) r_beta = (float)t[1] * 10.0f * xb;
            const float xg = expf(FMUL(a.gamma_p[0], 10.0f));
            if (xg >= 1e-6f && xg <= 1e6f) r_gamma = (float)t[2] * 10.0f * xg;
        } else {   // by-value scalars: report the gradients w.r.t. inv_s, beta, gamma themselves
            r_var = (float)t[0]; r_beta = (float)t[1]; r_gamma = (float)t[2];
        }
        if (d_variance) d_variance[0] = (accumulate ? d_variance[0] : 0.f) + r_var * grad_scale;
        if (d_beta) d_beta[0] = (accumulate ? d_beta[0] : 0.f) + r_beta * grad_scale;
        if (d_gamma) d_gamma[0] = (accumulate ? d_gamma[0] : 0.f) + r_gamma * grad_scale;
    }
}

// ---------------------------------------------------------------------------------------------
// Embedder.embed (embedder.py:34-35): x (P,3) -> (P, 3+6L), reference column order
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const float* x, long long P, int L, float* pe) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int d0 = 3 + 6 * L;
    const float xs[3] = {x[3 * p], x[3 * p + 1], x[3 * p + 2]};
    float* o = pe + p * d0;
    o[0] = xs[0]; o[1] = xs[1]; o[2] = xs[2];
    for (int k = 0; k < L; ++k) {
        const float f = (float)(1 << k);
        for (int c = 0; c < 3; ++c) {
            float sn, cs;
            sincosf(FMUL(xs[c], f), &sn, &cs);
            o[3 + 6 * k + c] = sn;
            o[3 + 6 * k + 3 + c] = cs;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int launch_sample_pdf(const float* bins, const float* weights, int N, int n, int m, float* samples, int64_t* inds,
                      int32_t* err, hipStream_t st, const float* u) {
    if (n < 2 || n > MAXS || m < 1 || m > MAXS) { set_error("sample_pdf: n=%d m=%d out of range (max %d)", n, m, MAXS); return EMAP_E_INVALID; }
    if (N <= 0) return EMAP_OK;
    hipLaunchKernelGGL(sample_pdf_kernel, dim3(N), dim3(64), 0, st, bins, weights, N, n, m, samples, inds, err, u);
    return check_launch("sample_pdf");
}

int launch_upsample(const float* rays_o, const float* rays_d, const float* z, const float* udf, int N, int n, int m,
                    const float* sample_dist, float inv_s, float beta, float gamma, float* z_new, int64_t* inds,
                    int32_t* err, hipStream_t st) {
    if (n < 2 || n > MAXS || m < 1 || m > MAXS) { set_error("upsample_step: n=%d m=%d out of range (max %d)", n, m, MAXS); return EMAP_E_INVALID; }
    if (N <= 0) return EMAP_OK;
    hipLaunchKernelGGL(upsample_kernel, dim3(N), dim3(64), 0, st, rays_o, rays_d, z, udf, N, n, m, sample_dist, inv_s, beta,
                       gamma, z_new, inds, err);
    return check_launch("upsample_step");
}

int launch_merge(const float* z, const float* z_new, const float* udf, const float* udf_new, int N, int n, int m,
                 float* z_out, float* udf_out, int64_t* perm, hipStream_t st) {
    if (n < 1 || n > MAXS || m < 1 || m > MAXS) { set_error("merge_sorted: n=%d m=%d out of range (max %d)", n, m, MAXS); return EMAP_E_INVALID; }
    if (udf_out && (!udf || !udf_new)) { set_error("merge_sorted: udf_out needs udf and udf_new"); return EMAP_E_INVALID; }
    if (N <= 0) return EMAP_OK;
    hipLaunchKernelGGL(merge_kernel, dim3(N), dim3(64), 0, st, z, z_new, udf, udf_new, N, n, m, z_out, udf_out, perm);
    return check_launch("merge_sorted");
}

int launch_sampler_step(bool coarse, bool tail, const StepArgs& a, hipStream_t st) {
    const int n_out = a.n + (coarse ? 0 : a.m);
    if (a.n < 2 || n_out + a.m > MAXS || a.m < 1) { set_error("sampler_step: n=%d m=%d out of range (max %d)", a.n, a.m, MAXS); return EMAP_E_INVALID; }
    if (a.N <= 0) return EMAP_OK;
    if (coarse) {
        if (tail) hipLaunchKernelGGL((sampler_step_kernel<true, false, true>), dim3(a.N), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((sampler_step_kernel<true, false, false>), dim3(a.N), dim3(64), 0, st, a);
    } else {
        if (tail) hipLaunchKernelGGL((sampler_step_kernel<false, true, true>), dim3(a.N), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((sampler_step_kernel<false, true, false>), dim3(a.N), dim3(64), 0, st, a);
    }
    return check_launch("sampler_step");
}

int launch_coarse(const float* near, const float* far, const float* t_rand, int N, int n_samples, float* z,
                  float* sample_dist, hipStream_t st) {
    if (N <= 0) return EMAP_OK;
    const long long tot = (long long)N * n_samples;
    hipLaunchKernelGGL(coarse_z_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, near, far, t_rand, N, n_samples, z,
                       sample_dist);
    return check_launch("coarse_z");
}

int fill_composite_args(const float* rays_o, const float* rays_d, const float* z, const float* udf, const float* grad3,
                        const float* depth_scale, int N, int S, const float* sample_dist, float inv_s, float beta,
                        float gamma, float car, int anneal, float flip_sat, float near_surface, float sparse_scale,
                        float background, int has_bg, const float* var_p, const float* beta_p, const float* gamma_p,
                        float beta_min, const EmapCompositeOut* out, float* partials, CompositeArgs* pa) {
    if (S < 1 || S > MAXS) { set_error("composite: S=%d out of range (max %d)", S, MAXS); return EMAP_E_INVALID; }
    if (!out || !partials) { set_error("composite: out/partials must not be null"); return EMAP_E_INVALID; }
    if (var_p && (!beta_p || !gamma_p)) { set_error("composite: variance_dev given without beta_dev/gamma_dev"); return EMAP_E_INVALID; }
    CompositeArgs& a = *pa;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.udf = udf; a.grad = grad3; a.depth_scale = depth_scale;
    a.sample_dist = sample_dist; a.N = N; a.S = S; a.inv_s = inv_s; a.beta = beta; a.gamma = gamma; a.car = car;
    a.anneal = anneal; a.flip_sat = flip_sat; a.near_surface = near_surface; a.sparse_scale = sparse_scale;
    a.background = background; a.has_bg = has_bg; a.out = *out; a.partials = partials;
    a.var_p = var_p; a.beta_p = beta_p; a.gamma_p = gamma_p; a.beta_min = beta_min;
    return EMAP_OK;
}

// the deterministic cross-ray reduction alone: what is left to launch when the value + grad_x kernel composited the rays itself (CompositeFuse)
int launch_composite_reduce(const CompositeArgs& a, int32_t* err, hipStream_t st) {
    if (a.N > 0 && a.out.scalars) hipLaunchKernelGGL(composite_reduce_kernel, dim3(1), dim3(256), 0, st, a.partials, a.N, a.out.scalars, err, a);
    return check_launch("composite_reduce");
}

int launch_composite(const float* rays_o, const float* rays_d, const float* z, const float* udf, const float* grad3,
                     const float* depth_scale, int N, int S, const float* sample_dist, float inv_s, float beta,
                     float gamma, float car, int anneal, float flip_sat, float near_surface, float sparse_scale,
                     float background, int has_bg, const float* var_p, const float* beta_p, const float* gamma_p,
                     float beta_min, const EmapCompositeOut* out, float* partials, int32_t* err, hipStream_t st) {
    CompositeArgs a;
    const int rc = fill_composite_args(rays_o, rays_d, z, udf, grad3, depth_scale, N, S, sample_dist, inv_s, beta, gamma, car, anneal, flip_sat,
                                       near_surface, sparse_scale, background, has_bg, var_p, beta_p, gamma_p, beta_min, out, partials, &a);
    if (rc) return rc;
    if (N <= 0) return EMAP_OK;
    if (S <= 64) hipLaunchKernelGGL(composite_kernel<1>, dim3(N), dim3(64), 0, st, a);
    else if (S <= 128) hipLaunchKernelGGL(composite_kernel<2>, dim3(N), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(composite_kernel<4>, dim3(N), dim3(64), 0, st, a);
    if (out->scalars) hipLaunchKernelGGL(composite_reduce_kernel, dim3(1), dim3(256), 0, st, partials, N, out->scalars, err, a);
    return check_launch("composite");
}

int launch_composite_bwd(const float* rays_o, const float* rays_d, const float* z, const float* udf, const float* grad3,
                         const float* depth_scale, int N, int S, const float* sample_dist, const EmapRenderParams* p,
                         const EmapCompositeGrads* gr, float* d_udf, float* d_grad3, float* partials, uint32_t* absmax,
                         hipStream_t st) {
    if (S < 1 || S > MAXS) { set_error("composite_bwd: S=%d out of range (max %d)", S, MAXS); return EMAP_E_INVALID; }
    if (N <= 0) return EMAP_OK;
    CompositeBwdArgs a;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.udf = udf; a.grad = grad3; a.depth_scale = depth_scale;
    a.sample_dist = sample_dist; a.N = N; a.S = S; a.inv_s = p->inv_s; a.beta = p->beta; a.gamma = p->gamma; a.car = p->cos_anneal_ratio;
    a.anneal = p->has_cos_anneal; a.flip_sat = p->flip_saturation; a.near_surface = p->near_surface;
    a.background = p->background; a.has_bg = p->has_background;
    a.var_p = p->variance_dev; a.beta_p = p->beta_dev; a.gamma_p = p->gamma_dev; a.beta_min = p->beta_min;
    if (a.var_p && (!a.beta_p || !a.gamma_p)) { set_error("composite_bwd: variance_dev given without beta_dev/gamma_dev"); return EMAP_E_INVALID; }
    a.d_edge = gr->d_edge; a.d_depth = gr->d_depth; a.d_ge = gr->d_gradient_error; a.d_ge_ns = gr->d_gradient_error_near_surface;
    a.scalars = gr->scalars; a.d_udf = d_udf; a.d_grad = d_grad3; a.partials = partials; a.absmax = absmax;
    a.zero_tail = gr->n_zero_tail > 0 ? gr->zero_tail : nullptr; a.n_zero_tail = gr->n_zero_tail;
    a.raymax = absmax ? partials + (size_t)N * 4 : nullptr;     // internal callers (emap_render_bwd) size `partials` as (N,4) + (N,2)
    if ((a.d_ge || a.d_ge_ns) && !a.scalars) { set_error("composite_bwd: the eikonal gradients need the forward's scalars"); return EMAP_E_INVALID; }
    if (S <= 64) hipLaunchKernelGGL(composite_bwd_kernel<1>, dim3(N), dim3(64), 0, st, a);
    else if (S <= 128) hipLaunchKernelGGL(composite_bwd_kernel<2>, dim3(N), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(composite_bwd_kernel<4>, dim3(N), dim3(64), 0, st, a);
    hipLaunchKernelGGL(composite_bwd_reduce_kernel, dim3(1), dim3(256), 0, st, partials, N, a, gr->d_variance, gr->d_beta,
                       gr->d_gamma, gr->grad_scale, gr->accumulate);
    return check_launch("composite_bwd");
}

int launch_embed(const float* x, int64_t P, int L, float* pe, hipStream_t st) {
    if (L < 1 || L > 16) { set_error("embed: multires=%d out of range", L); return EMAP_E_INVALID; }
    if (P <= 0) return EMAP_OK;
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, x, (long long)P, L, pe);
    return check_launch("embed");
}

// host copy of the u grid used by sample_pdf, for CPU-side tests of the linspace restatement
void linspace_host(float start, float end, int steps, float* out) {
    for (int i = 0; i < steps; ++i) out[i] = linspace_at_host(start, end, steps, i);
}

}  // namespace emap
