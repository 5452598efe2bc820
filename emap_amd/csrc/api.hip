// api.hip - the extern "C" surface of libemap_hip.so (include/emap_hip.h) and the host-side
// orchestration of one forward render: every kernel of UDFRendererBlending.render
// (reference src/models/udf_renderer_blending.py:679-800) is enqueued back-to-back on the caller's
// stream with zero host synchronisation (the reference has >= 11 device->host syncs per render,
// SURVEY.md par. 3.1).
#include "emap_common.h"
#include <stdarg.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <stdlib.h>

namespace emap {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return EMAP_E_LAUNCH;
    }
    return EMAP_OK;
}

// ---- optional per-launch timing of the dominant kernel (the final value+grad MLP pass) ----------
// bench.py enables it for the timed region; hipEvents are recorded on the launch stream right
// before/after that kernel, and read back after the region's final synchronisation.
constexpr int PROF_MAX = 1024;
constexpr int PROF_KERNELS = 3;   // 0: final value+grad pass of render_fwd, 1: udf_mlp_vjp sweep, 2: wgrad GEMMs (render_bwd / udf_vjp)
static bool g_prof_on = false;
static int g_prof_n[PROF_KERNELS] = {0, 0, 0};
static hipEvent_t g_prof_ev[PROF_KERNELS][PROF_MAX][2];
static bool g_prof_init = false;
static long long* g_clk_dev = nullptr;
struct ProfScope {   // records a HIP event pair around one launch on its stream while profiling is enabled
    int k; hipStream_t st; bool on;
    ProfScope(int k_, hipStream_t st_) : k(k_), st(st_), on(g_prof_on && g_prof_n[k_] < PROF_MAX) { if (on) (void)hipEventRecord(g_prof_ev[k][g_prof_n[k]][0], st); }
    ~ProfScope() { if (on) { (void)hipEventRecord(g_prof_ev[k][g_prof_n[k]][1], st); ++g_prof_n[k]; } }
};

int launch_sample_pdf(const float*, const float*, int, int, int, float*, int64_t*, int32_t*, hipStream_t, const float* u = nullptr);
int launch_upsample(const float*, const float*, const float*, const float*, int, int, int, const float*, float, float,
                    float, float*, int64_t*, int32_t*, hipStream_t);
int launch_merge(const float*, const float*, const float*, const float*, int, int, int, float*, float*, int64_t*,
                 hipStream_t);
int launch_coarse(const float*, const float*, const float*, int, int, float*, float*, hipStream_t);
int launch_composite(const float*, const float*, const float*, const float*, const float*, const float*, int, int,
                     const float*, float, float, float, float, int, float, float, float, float, int, const float*,
                     const float*, const float*, float, const EmapCompositeOut*, float*, int32_t*, hipStream_t);
int fill_composite_args(const float*, const float*, const float*, const float*, const float*, const float*, int, int,
                        const float*, float, float, float, float, int, float, float, float, float, int, const float*,
                        const float*, const float*, float, const EmapCompositeOut*, float*, CompositeArgs*);
int launch_composite_reduce(const CompositeArgs&, int32_t*, hipStream_t);
int launch_embed(const float*, int64_t, int, float*, hipStream_t);
void linspace_host(float, float, int, float*);
int launch_sample_rays(const EmapRayDataset*, int, int, int, uint64_t, uint64_t, uint64_t*, const int64_t*, const EmapRayBatch*,
                       hipStream_t);
int launch_composite_bwd(const float*, const float*, const float*, const float*, const float*, const float*, int, int,
                         const float*, const EmapRenderParams*, const EmapCompositeGrads*, float*, float*, float*, uint32_t*,
                         hipStream_t);

// ---- scalar tail of a training step (train.hip) ----
int launch_train_stats(const float*, const float*, const float*, int, float, float*, float*, hipStream_t);
int launch_train_loss(const float*, float, float, float, float*, hipStream_t);
int launch_adam(float*, const float*, float*, float*, float*, int64_t, int64_t, float, float, double, double, float, const float*, float*, hipStream_t);
// ---- training backward (udf_mlp_vjp.inc, wgrad.hip) ----
#define EMAP_VJP_DECL(m) \
    int launch_vjp_sweep_##m(const NetLayout&, const void*, const PointSource&, int64_t, int, int, const float*, const float*, \
                             const VjpLayout&, char*, char*, char*, int, const uint32_t*, float*, hipStream_t, int32_t*);
EMAP_VJP_DECL(bf16) EMAP_VJP_DECL(bf16x3) EMAP_VJP_DECL(f16) EMAP_VJP_DECL(f16x3)
#undef EMAP_VJP_DECL
size_t plan_wgrad(const NetLayout&, const VjpLayout&, int, WgradJob*, int*, int*, int*, int*);
int launch_absmax(const float*, const float*, int64_t, uint32_t*, hipStream_t);
int launch_wgrad(const NetLayout&, const VjpLayout&, const WgradJob*, int, int, const char*, const char*, float*, int, int,
                 hipStream_t, float scale = 1.0f, int no_bias = 0);
int launch_wgrad_reduce(const NetLayout&, const WgradJob*, int, const int*, const int*, const float*, const uint32_t*, const float*, int,
                        const float* const*, const float* const*, float* const*, float* const*, float* const*, int, int, float,
                        hipStream_t);

static int device_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n = pr.multiProcessorCount;
        else n = 256;
    }
    return n;
}

struct VjpPlan {
    VjpLayout V;
    WgradJob jobs[WGRAD_MAX_JOBS];
    int n_jobs, job_h[EMAP_MAX_LIN], job_pe[EMAP_MAX_LIN], wgrad_wg, sweep_grid, chunk_tiles;
    size_t off_absmax, off_slab, off_partial, off_ldot, off_a, off_z, total;
};

// avail = 0: the preferred plan (chunks of up to VJP_CHUNK_TILES tiles); avail > 0: the largest chunk whose stash fits into a workspace of
// `avail` bytes - the caller bounds the memory, the backward runs in more chunks (chunk_tiles = 0: not even VJP_MIN_CHUNK_TILES fit)
constexpr int VJP_MIN_CHUNK_TILES = 256;
static VjpPlan plan_vjp(const NetLayout& L, int64_t P, size_t avail = 0) {
    VjpPlan pl;
    build_vjp_layout(L, &pl.V);
    const int cus = device_cus();
    pl.sweep_grid = (L.H == 256 && VJP_NW_256 == 8) ? cus : 2 * cus;          // 8 waves: one workgroup per CU; 4 waves: two
    const int64_t tiles = (P + VJP_PT - 1) / VJP_PT;
    pl.chunk_tiles = (int)std::min<int64_t>(std::max<int64_t>(tiles, 1), VJP_CHUNK_TILES);
    const size_t pfl = plan_wgrad(L, pl.V, cus, pl.jobs, &pl.n_jobs, pl.job_h, pl.job_pe, &pl.wgrad_wg);
    size_t off = 0;
    pl.off_absmax = off; off += 256;
    pl.off_slab = off; off += (size_t)pl.sweep_grid * pl.V.s_slab_kb * 1024;
    pl.off_partial = off; off += ((pfl * 4 + 255) & ~(size_t)255);
    pl.off_ldot = off; off += (((size_t)std::max<int64_t>(tiles, 1) * 4 + 255) & ~(size_t)255);
    // precise weight gradients (L.wgrad_lo): a second pair of stashes for the lo parts of both operand sets
    const size_t per_tile = ((size_t)pl.V.a_tile_kb + (size_t)pl.V.z_tile_kb) * 1024 * (L.wgrad_lo ? 2 : 1);
    if (avail > 0) {
        const size_t fit = avail > off ? (avail - off) / per_tile : 0;
        const int64_t need = std::min<int64_t>(std::max<int64_t>(tiles, 1), VJP_MIN_CHUNK_TILES);
        pl.chunk_tiles = (int64_t)fit < need ? 0 : (int)std::min<size_t>(fit, (size_t)pl.chunk_tiles);
    }
    pl.off_a = off; off += (size_t)pl.chunk_tiles * pl.V.a_tile_kb * 1024;
    pl.off_z = off; off += (size_t)pl.chunk_tiles * pl.V.z_tile_kb * 1024;
    pl.V.lo_a_delta = pl.V.lo_z_delta = 0;
    if (L.wgrad_lo) {
        pl.V.lo_a_delta = (long long)(off - pl.off_a); off += (size_t)pl.chunk_tiles * pl.V.a_tile_kb * 1024;
        pl.V.lo_z_delta = (long long)(off - pl.off_z); off += (size_t)pl.chunk_tiles * pl.V.z_tile_kb * 1024;
    }
    pl.total = off;
    return pl;
}
// bytes of the smallest workspace plan_vjp accepts for P points
static size_t vjp_min_bytes(const NetLayout& L, int64_t P) {
    const VjpPlan pl = plan_vjp(L, P);
    const int64_t tiles = (P + VJP_PT - 1) / VJP_PT;
    const size_t per_tile = ((size_t)pl.V.a_tile_kb + (size_t)pl.V.z_tile_kb) * 1024 * (L.wgrad_lo ? 2 : 1);
    return pl.off_a + (size_t)std::min<int64_t>(std::max<int64_t>(tiles, 1), VJP_MIN_CHUNK_TILES) * per_tile;
}

// d/dtheta of sum_p du[p] udf(x_p) + dg[p] . grad udf(x_p); absmax must already hold max|du|, max|dg| of the launch
static int run_vjp(const NetLayout& L, const void* packed, int prec, const PointSource& src, int64_t P, const float* d_udf,
                   const float* d_grad, const EmapParamGrads* out, const VjpPlan& pl, char* ws, int32_t* err, hipStream_t st) {
    uint32_t* absmax = reinterpret_cast<uint32_t*>(ws + pl.off_absmax);
    float* partial = reinterpret_cast<float*>(ws + pl.off_partial);
    float* ldot = reinterpret_cast<float*>(ws + pl.off_ldot);
    const int64_t tiles = (P + VJP_PT - 1) / VJP_PT;
    int chunk = 0;
    for (int64_t t0 = 0; t0 < tiles || chunk == 0; t0 += pl.chunk_tiles, ++chunk) {
        const int nt = (int)std::min<int64_t>(pl.chunk_tiles, std::max<int64_t>(tiles - t0, 0));
        int rc = EMAP_OK;
        {
        ProfScope ps(1, st);
        switch (prec) {
            case EMAP_PREC_BF16: rc = launch_vjp_sweep_bf16(L, packed, src, P, (int)t0, nt, d_udf, d_grad, pl.V, ws + pl.off_a, ws + pl.off_z, ws + pl.off_slab, pl.sweep_grid, absmax, ldot, st, err); break;
            case EMAP_PREC_BF16X3: rc = launch_vjp_sweep_bf16x3(L, packed, src, P, (int)t0, nt, d_udf, d_grad, pl.V, ws + pl.off_a, ws + pl.off_z, ws + pl.off_slab, pl.sweep_grid, absmax, ldot, st, err); break;
            case EMAP_PREC_F16: rc = launch_vjp_sweep_f16(L, packed, src, P, (int)t0, nt, d_udf, d_grad, pl.V, ws + pl.off_a, ws + pl.off_z, ws + pl.off_slab, pl.sweep_grid, absmax, ldot, st, err); break;
            default: rc = launch_vjp_sweep_f16x3(L, packed, src, P, (int)t0, nt, d_udf, d_grad, pl.V, ws + pl.off_a, ws + pl.off_z, ws + pl.off_slab, pl.sweep_grid, absmax, ldot, st, err); break;
        }
        }
        if (rc) return rc;
        ProfScope pw(2, st);
        rc = launch_wgrad(L, pl.V, pl.jobs, pl.n_jobs, pl.wgrad_wg, ws + pl.off_a, ws + pl.off_z, partial, nt, chunk > 0 ? 1 : 0, st);
        if (rc) return rc;
        if (L.wgrad_lo && pl.V.lo_a_delta) {
            // precise weight gradients: dW = Z_hi A_hi^T + (Z_hi A_lo^T + Z_lo A_hi^T) / LO_SCALE - the two cross terms from the lo stashes, added
            // to the same K-slice partials (the lo x lo term is below 2^-22 of the product)
            const float inv_lo = 1.0f / 2048.0f;      // LO_SCALE of the split-fp16 modes (udf_mlp_vjp.inc)
            rc = launch_wgrad(L, pl.V, pl.jobs, pl.n_jobs, pl.wgrad_wg, ws + pl.off_a + pl.V.lo_a_delta, ws + pl.off_z, partial, nt, 1, st, inv_lo, 1);
            if (rc) return rc;
            rc = launch_wgrad(L, pl.V, pl.jobs, pl.n_jobs, pl.wgrad_wg, ws + pl.off_a, ws + pl.off_z + pl.V.lo_z_delta, partial, nt, 1, st, inv_lo, 0);
            if (rc) return rc;
        }
        if (tiles == 0) break;
    }
    return launch_wgrad_reduce(L, pl.jobs, pl.n_jobs, pl.job_h, pl.job_pe, partial, absmax, ldot, (int)tiles, out->g_host, out->v_host, out->dg_host,
                               out->dv_host, out->db_host, out->weight_norm, out->accumulate, out->grad_scale, st);
}

static int check_param_grads(const NetLayout& L, const EmapParamGrads* o, const char* who) {
    if (!o || !o->v_host || !o->dv_host || !o->db_host) { set_error("%s: null parameter-gradient table", who); return EMAP_E_INVALID; }
    if (o->weight_norm && (!o->g_host || !o->dg_host)) { set_error("%s: weight_norm needs g_host and dg_host", who); return EMAP_E_INVALID; }
    for (int l = 0; l < L.n_lin; ++l) {
        if (!o->v_host[l] || !o->dv_host[l] || !o->db_host[l] || (o->weight_norm && (!o->g_host[l] || !o->dg_host[l]))) {
            set_error("%s: null tensor for layer %d", who, l);
            return EMAP_E_INVALID;
        }
    }
    return EMAP_OK;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Workspace {
    size_t sample_dist, z_a, z_b, udf_a, udf_b, z_new, z_new2, udf_new, partials, ray_cnt, rev, total;
};

// render_core's tail inside the value + grad_x kernel (CompositeFuse; ABI 9).  EMAP_FUSED_COMPOSITE=0 / emap_set_fused_composite(0): the
// separate composite_kernel launch of rounds 1-5 (same results bit for bit: tests, A/B).  Read once at load.
static int fused_composite_from_env() {
    const char* e = getenv("EMAP_FUSED_COMPOSITE");
    return (e && e[0] == '0') ? 0 : 1;
}
static std::atomic<int> g_fused_composite{fused_composite_from_env()};
#ifndef EMAP_FUSED_REDUCE
#define EMAP_FUSED_REDUCE 1     // the fused tail also runs the cross-ray reduction (CompositeFuse::done_cnt); 0: composite_reduce_kernel as a launch of its own (A/B)
#endif

static Workspace plan_workspace(const EmapRenderParams& p, const NetLayout* L = nullptr) {
    const size_t N = (size_t)std::max(p.n_rays, 0);
    const int m = p.up_sample_steps > 0 ? p.n_importance / p.up_sample_steps : 0;
    const size_t S = (size_t)p.n_samples + (size_t)m * std::max(p.up_sample_steps, 0);
    Workspace w;
    size_t off = 0;
    w.sample_dist = off; off += 256;
    w.z_a = off; off += align256(N * S * 4);
    w.z_b = off; off += align256(N * S * 4);
    w.udf_a = off; off += align256(N * S * 4);
    w.udf_b = off; off += align256(N * S * 4);
    w.z_new = off; off += align256(N * (size_t)std::max(m, 1) * 4);
    w.z_new2 = off; off += align256(N * (size_t)std::max(m, 1) * 4);
    w.udf_new = off; off += align256(N * (size_t)std::max(m, 1) * 4);
    w.partials = off; off += align256(N * 8 * 4);
    w.ray_cnt = off; off += align256((N + 1) * 4);      // arrival counters of the fused compositing tail (int32 per ray) + the launch's count of composited rays
    w.rev = off; off += L ? align256(rev_scratch_bytes(*L)) : 0;   // sigma' slabs of the reverse-mode value+gradient kernel
    w.total = off;
    return w;
}

}  // namespace emap

using namespace emap;

extern "C" {

int emap_abi_version(void) { return EMAP_ABI_VERSION; }
int emap_set_fused_sampling(int on) { return set_fused_sampling(on); }
int emap_set_fused_composite(int on) { return g_fused_composite.exchange(on ? 1 : 0, std::memory_order_relaxed); }
const char* emap_last_error(void) { return g_err; }
int emap_set_grad_mode(int mode) { return set_grad_mode(mode); }
int emap_set_value_tile_mode(int on) { return set_value_tile_mode(on); }

int emap_packed_bytes(const EmapNetConfig* cfg, int prec, size_t* bytes) {
    NetLayout L;
    const int rc = build_layout(cfg, prec, &L);
    if (rc) return rc;
    if (!bytes) { set_error("bytes is null"); return EMAP_E_INVALID; }
    *bytes = layout_bytes(L);
    return EMAP_OK;
}

int emap_pack_weights(const EmapNetConfig* cfg, const float* const* g, const float* const* v, const float* const* b,
                      void* packed, int prec, void* stream) {
    NetLayout L;
    const int rc = build_layout(cfg, prec, &L);
    if (rc) return rc;
    if (!g || !v || !b || !packed) { set_error("pack_weights: null pointer"); return EMAP_E_INVALID; }
    for (int l = 0; l < L.n_lin; ++l)
        if (!g[l] || !v[l] || !b[l]) { set_error("pack_weights: null tensor for layer %d", l); return EMAP_E_INVALID; }
    return launch_pack(L, g, v, b, packed, static_cast<hipStream_t>(stream));
}

static int udf_call(const EmapNetConfig* cfg, const void* packed, int prec, const float* x, int64_t P, float* udf,
                    float* grad3, void* scratch, size_t scratch_bytes, void* stream) {
    NetLayout L;
    const int rc = build_layout(cfg, prec, &L);
    if (rc) return rc;
    if (!packed || !udf || (P > 0 && !x)) { set_error("udf_fwd: null pointer"); return EMAP_E_INVALID; }
    if (P < 0) { set_error("udf_fwd: negative P"); return EMAP_E_INVALID; }
    if (P == 0) return EMAP_OK;
    if (grad3 && mlp_uses_rev(L, prec, P) && (!scratch || scratch_bytes < rev_scratch_bytes(L))) {
        set_error("udf_fwd_grad: scratch %zu < %zu bytes (emap_udf_scratch_bytes)", scratch ? scratch_bytes : (size_t)0, rev_scratch_bytes(L));
        return EMAP_E_WORKSPACE;
    }
    PointSource src;
    memset(&src, 0, sizeof(src));
    src.x = x;
    return launch_mlp(L, packed, prec, src, P, udf, grad3, static_cast<hipStream_t>(stream), nullptr, scratch);
}

int emap_udf_scratch_bytes(const EmapNetConfig* cfg, int prec, int64_t P, size_t* bytes) {
    NetLayout L;
    const int rc = build_layout(cfg, prec, &L);
    if (rc) return rc;
    if (!bytes || P < 0) { set_error("udf_scratch_bytes: bad argument"); return EMAP_E_INVALID; }
    *bytes = mlp_uses_rev(L, prec, P) ? rev_scratch_bytes(L) : 0;
    return EMAP_OK;
}

int emap_udf_fwd(const EmapNetConfig* cfg, const void* packed, int prec, const float* x, int64_t P, float* udf,
                 void* stream) {
    return udf_call(cfg, packed, prec, x, P, udf, nullptr, nullptr, 0, stream);
}

int emap_udf_fwd_grad(const EmapNetConfig* cfg, const void* packed, int prec, const float* x, int64_t P, float* udf,
                      float* grad3, void* scratch, size_t scratch_bytes, void* stream) {
    if (!grad3) { set_error("udf_fwd_grad: grad3 is null"); return EMAP_E_INVALID; }
    return udf_call(cfg, packed, prec, x, P, udf, grad3, scratch, scratch_bytes, stream);
}

int emap_embed(const float* x, int64_t P, int multires, float* pe, void* stream) {
    if (P > 0 && (!x || !pe)) { set_error("embed: null pointer"); return EMAP_E_INVALID; }
    return launch_embed(x, P, multires, pe, static_cast<hipStream_t>(stream));
}

int emap_sample_pdf(const float* bins, const float* weights, int N, int n, int m, float* samples, int64_t* inds,
                    int32_t* err_flags, void* stream) {
    if (N > 0 && (!bins || !weights || !samples)) { set_error("sample_pdf: null pointer"); return EMAP_E_INVALID; }
    return launch_sample_pdf(bins, weights, N, n, m, samples, inds, err_flags, static_cast<hipStream_t>(stream));
}
int emap_sample_pdf_u(const float* bins, const float* weights, const float* u, int N, int n, int m, float* samples, int64_t* inds,
                      int32_t* err_flags, void* stream) {
    if (N > 0 && (!bins || !weights || !samples || !u)) { set_error("sample_pdf_u: null pointer"); return EMAP_E_INVALID; }
    return launch_sample_pdf(bins, weights, N, n, m, samples, inds, err_flags, static_cast<hipStream_t>(stream), u);
}

int emap_upsample_step(const float* rays_o, const float* rays_d, const float* z, const float* udf, int N, int n, int m,
                       const float* sample_dist_dev, float inv_s, float beta, float gamma, float* z_new, int64_t* inds,
                       int32_t* err_flags, void* stream) {
    if (N > 0 && (!rays_o || !rays_d || !z || !udf || !sample_dist_dev || !z_new)) { set_error("upsample_step: null pointer"); return EMAP_E_INVALID; }
    return launch_upsample(rays_o, rays_d, z, udf, N, n, m, sample_dist_dev, inv_s, beta, gamma, z_new, inds, err_flags,
                           static_cast<hipStream_t>(stream));
}

int emap_merge_sorted(const float* z, const float* z_new, const float* udf, const float* udf_new, int N, int n, int m,
                      float* z_out, float* udf_out, int64_t* perm, void* stream) {
    if (N > 0 && (!z || !z_new || !z_out)) { set_error("merge_sorted: null pointer"); return EMAP_E_INVALID; }
    return launch_merge(z, z_new, udf, udf_new, N, n, m, z_out, udf_out, perm, static_cast<hipStream_t>(stream));
}

int emap_composite_fwd(const float* rays_o, const float* rays_d, const float* z, const float* udf, const float* grad3,
                       const float* depth_scale, int N, int S, const float* sample_dist_dev, float inv_s, float beta,
                       float gamma, float cos_anneal_ratio, int has_cos_anneal, float flip_saturation,
                       float near_surface, float sparse_scale, float background, int has_background,
                       const EmapCompositeOut* out, float* partials, int32_t* err_flags, void* stream) {
    if (N > 0 && (!rays_o || !rays_d || !z || !udf || !grad3 || !sample_dist_dev)) { set_error("composite_fwd: null pointer"); return EMAP_E_INVALID; }
    return launch_composite(rays_o, rays_d, z, udf, grad3, depth_scale, N, S, sample_dist_dev, inv_s, beta, gamma,
                            cos_anneal_ratio, has_cos_anneal, flip_saturation, near_surface, sparse_scale, background,
                            has_background, nullptr, nullptr, nullptr, 0.f, out, partials, err_flags,
                            static_cast<hipStream_t>(stream));
}

int emap_composite_fwd_p(const float* rays_o, const float* rays_d, const float* z, const float* udf, const float* grad3,
                         const float* depth_scale, int N, int S, const float* sample_dist_dev, const EmapRenderParams* p,
                         const EmapCompositeOut* out, float* partials, int32_t* err_flags, void* stream) {
    if (!p || (N > 0 && (!rays_o || !rays_d || !z || !udf || !grad3 || !sample_dist_dev))) { set_error("composite_fwd_p: null pointer"); return EMAP_E_INVALID; }
    return launch_composite(rays_o, rays_d, z, udf, grad3, depth_scale, N, S, sample_dist_dev, p->inv_s, p->beta, p->gamma,
                            p->cos_anneal_ratio, p->has_cos_anneal, p->flip_saturation, p->near_surface, p->sparse_scale,
                            p->background, p->has_background, p->variance_dev, p->beta_dev, p->gamma_dev, p->beta_min, out,
                            partials, err_flags, static_cast<hipStream_t>(stream));
}

int emap_render_workspace_bytes(const EmapNetConfig* cfg, int prec, const EmapRenderParams* p, size_t* bytes) {
    NetLayout L;
    const int rc = build_layout(cfg, prec, &L);
    if (rc) return rc;
    if (!p || !bytes) { set_error("render_workspace_bytes: null pointer"); return EMAP_E_INVALID; }
    *bytes = plan_workspace(*p, &L).total;
    return EMAP_OK;
}

int emap_render_fwd(const EmapNetConfig* cfg, const void* packed, int prec, const EmapRenderParams* p,
                    const float* rays_o, const float* rays_d, const float* near, const float* far, const float* t_rand,
                    const float* depth_scale, float* z_vals, float* udf, float* grad3, const EmapCompositeOut* out,
                    void* workspace, size_t workspace_bytes, int32_t* err_flags, void* stream) {
    NetLayout L;
    int rc = build_layout(cfg, prec, &L);
    if (rc) return rc;
    if (!p || !packed || !rays_o || !rays_d || !near || !far || !z_vals || !udf || !grad3 || !out || !workspace) {
        set_error("render_fwd: null pointer");
        return EMAP_E_INVALID;
    }
    const int N = p->n_rays, Sc = p->n_samples, K = p->up_sample_steps;
    if (N <= 0) return EMAP_OK;
    if (Sc < 2 || p->n_importance < 0 || (p->n_importance > 0 && K < 1)) { set_error("render_fwd: bad sampling configuration"); return EMAP_E_INVALID; }
    const int m = (p->n_importance > 0) ? p->n_importance / K : 0;
    const int steps = (m > 0) ? K : 0;
    const int S = Sc + m * steps;
    if (S > 256) { set_error("render_fwd: %d samples per ray exceed the kernel limit of 256", S); return EMAP_E_INVALID; }
    const Workspace w = plan_workspace(*p, &L);
    if (workspace_bytes < w.total) { set_error("render_fwd: workspace %zu < %zu bytes", workspace_bytes, w.total); return EMAP_E_WORKSPACE; }
    char* ws = static_cast<char*>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* sample_dist = reinterpret_cast<float*>(ws + w.sample_dist);
    float* zbuf[2] = {reinterpret_cast<float*>(ws + w.z_a), reinterpret_cast<float*>(ws + w.z_b)};
    float* ubuf[2] = {reinterpret_cast<float*>(ws + w.udf_a), reinterpret_cast<float*>(ws + w.udf_b)};
    float* znew[2] = {reinterpret_cast<float*>(ws + w.z_new), reinterpret_cast<float*>(ws + w.z_new2)};
    float* udf_new = reinterpret_cast<float*>(ws + w.udf_new);
    float* partials = reinterpret_cast<float*>(ws + w.partials);
    // render_core's tail inside the final value + grad_x launch (ABI 9) where that launch is the reverse-sweep kernel: the first launch of
    // the render clears the per-ray arrival counters, the final one composites every ray as its last tile completes
    int32_t* ray_cnt = reinterpret_cast<int32_t*>(ws + w.ray_cnt);
    const bool fuse_comp = g_fused_composite.load(std::memory_order_relaxed) && mlp_uses_rev(L, prec, (int64_t)N * S) &&
                           comp_list_entries((int64_t)N * S, S) <= COMP_LIST_MAX;

    if (steps == 0) {
        // no up-sampling: the coarse samples (render() :700-720) are the final z_vals
        rc = launch_coarse(near, far, t_rand, N, Sc, z_vals, sample_dist, st);
        if (rc) return rc;
        if (fuse_comp && hipMemsetAsync(ray_cnt, 0, ((size_t)N + 1) * 4, st) != hipSuccess) { (void)hipGetLastError(); set_error("render_fwd: memset failed"); return EMAP_E_LAUNCH; }
    } else {
        // importance_sample (:802-841) in 2*steps launches: the coarse z_vals are evaluated where they are consumed (the first
        // MLP pass and the first sampler step, same separately rounded expression), every later sampler step merges the previous
        // step's samples (cat_z_vals :355-377), up-samples (:228-353) and - in the last step - merges again, in one launch
        PointSource src;
        memset(&src, 0, sizeof(src));
        src.rays_o = rays_o; src.rays_d = rays_d; src.n_per_ray = Sc; src.mid = 0; src.sample_dist = sample_dist;
        src.coarse = 1; src.near = near; src.far = far; src.t_rand = t_rand;
        if (fuse_comp) { src.zero_cnt = ray_cnt; src.zero_n = N + 1; }
        rc = launch_mlp(L, packed, prec, src, (int64_t)N * Sc, ubuf[0], nullptr, st, err_flags);
        if (rc) return rc;
        src.coarse = 0; src.zero_cnt = nullptr; src.zero_n = 0;
        // everything from here to the final z_vals in ONE launch where the shape allows (16 new samples per ray and step, below 2048 rays):
        // sampler steps and MLP passes alternate inside the workgroup that owns the rays (udf_mlp_kernel.inc, IS)
        IsLaunch q;
        q.rays_o = rays_o; q.rays_d = rays_d; q.near = near; q.far = far; q.t_rand = t_rand; q.sample_dist = sample_dist;
        q.udf_coarse = ubuf[0]; q.z_final = z_vals; q.N = N; q.Sc = Sc; q.m = m; q.steps = steps;
        rc = launch_importance(L, packed, prec, q, st, err_flags);
        if (rc < 0) return rc;
        int cur = 0, n = Sc;
        for (int i = 0; rc == IS_NOT_FUSED && i < steps; ++i) {
            const bool last = (i + 1 == steps);
            StepArgs a;
            memset(&a, 0, sizeof(a));
            a.rays_o = rays_o; a.rays_d = rays_d; a.N = N; a.m = m; a.err = err_flags; a.sample_dist = sample_dist;
            a.inv_s = 64.0f * (float)(1 << i);                                          // :826
            a.beta = 64.0f * (float)(1 << (i + 1));                                     // :828
            a.gamma = std::min(std::max(20.0f * (float)(1 << (K - i)), 20.0f), 320.0f);  // :830
            a.z_new = znew[i & 1];
            a.z_final = last ? z_vals : nullptr;
            if (i == 0) {
                a.n = Sc; a.udf = ubuf[0]; a.z_merged = zbuf[0]; a.near = near; a.far = far; a.t_rand = t_rand;
            } else {
                a.n = n; a.z = zbuf[cur]; a.udf = ubuf[cur]; a.z_prev = znew[(i - 1) & 1]; a.udf_prev = udf_new;
                a.z_merged = zbuf[cur ^ 1]; a.udf_merged = ubuf[cur ^ 1];
                cur ^= 1;
                n += m;
            }
            int rc2 = launch_sampler_step(i == 0, last, a, st);
            if (rc2) return rc2;
            if (!last) {
                src.z = znew[i & 1]; src.n_per_ray = m;
                rc2 = launch_mlp(L, packed, prec, src, (int64_t)N * m, udf_new, nullptr, st, err_flags);
                if (rc2) return rc2;
            }
        }
    }

    // render_core (:418-677): MLP value + grad at the interval mid-points, then compositing
    PointSource fin;
    memset(&fin, 0, sizeof(fin));
    fin.rays_o = rays_o; fin.rays_d = rays_d; fin.z = z_vals; fin.n_per_ray = S; fin.mid = 1; fin.sample_dist = sample_dist;
    CompositeFuse cf;
    if (fuse_comp) {
        rc = fill_composite_args(rays_o, rays_d, z_vals, udf, grad3, depth_scale, N, S, sample_dist, p->inv_s, p->beta, p->gamma,
                                 p->cos_anneal_ratio, p->has_cos_anneal, p->flip_saturation, p->near_surface, p->sparse_scale,
                                 p->background, p->has_background, p->variance_dev, p->beta_dev, p->gamma_dev, p->beta_min, out,
                                 partials, &cf.c);
        if (rc) return rc;
        cf.ray_cnt = ray_cnt;
        cf.done_cnt = EMAP_FUSED_REDUCE ? ray_cnt + N : nullptr;
    }
    {
        ProfScope ps(0, st);
        rc = launch_mlp(L, packed, prec, fin, (int64_t)N * S, udf, grad3, st, err_flags, ws + w.rev, fuse_comp ? &cf : nullptr);
    }
    if (rc) return rc;
    // 3 launches per render: value pass, importance_sample, value + grad_x + compositing + cross-ray reduction (-DEMAP_FUSED_REDUCE=0: the reduction as a fourth)
    if (fuse_comp) return (cf.done_cnt && cf.c.out.scalars) ? EMAP_OK : launch_composite_reduce(cf.c, err_flags, st);
    return launch_composite(rays_o, rays_d, z_vals, udf, grad3, depth_scale, N, S, sample_dist, p->inv_s, p->beta, p->gamma,
                            p->cos_anneal_ratio, p->has_cos_anneal, p->flip_saturation, p->near_surface, p->sparse_scale,
                            p->background, p->has_background, p->variance_dev, p->beta_dev, p->gamma_dev, p->beta_min, out,
                            partials, err_flags, st);
}

int emap_composite_bwd(const float* rays_o, const float* rays_d, const float* z, const float* udf, const float* grad3,
                       const float* depth_scale, int N, int S, const float* sample_dist_dev, const EmapRenderParams* p,
                       const EmapCompositeGrads* g, float* d_udf, float* d_grad3, float* partials, void* stream) {
    if (!p || !g || (N > 0 && (!rays_o || !rays_d || !z || !udf || !grad3 || !sample_dist_dev || !d_udf || !d_grad3 || !partials))) {
        set_error("composite_bwd: null pointer");
        return EMAP_E_INVALID;
    }
    return launch_composite_bwd(rays_o, rays_d, z, udf, grad3, depth_scale, N, S, sample_dist_dev, p, g, d_udf, d_grad3, partials,
                                nullptr, static_cast<hipStream_t>(stream));
}

int emap_udf_vjp_workspace_bytes(const EmapNetConfig* cfg, int prec, int64_t P, size_t* bytes) {
    NetLayout L;
    const int rc = build_layout(cfg, prec, &L);
    if (rc) return rc;
    if (!bytes || P < 0) { set_error("udf_vjp_workspace_bytes: bad argument"); return EMAP_E_INVALID; }
    *bytes = plan_vjp(L, P).total;
    return EMAP_OK;
}

int emap_udf_vjp(const EmapNetConfig* cfg, const void* packed, int prec, const float* x, int64_t P, const float* d_udf,
                 const float* d_grad3, const EmapParamGrads* out, void* workspace, size_t workspace_bytes, int32_t* err_flags,
                 void* stream) {
    NetLayout L;
    int rc = build_layout(cfg, prec, &L);
    if (rc) return rc;
    if (P < 0 || !packed || !workspace || (P > 0 && (!x || !d_udf || !d_grad3))) { set_error("udf_vjp: null pointer"); return EMAP_E_INVALID; }
    rc = check_param_grads(L, out, "udf_vjp");
    if (rc) return rc;
    const VjpPlan pl = plan_vjp(L, P, workspace_bytes ? workspace_bytes : 1);
    if (pl.chunk_tiles <= 0) { set_error("udf_vjp: workspace %zu < %zu bytes (minimum; emap_udf_vjp_workspace_bytes is the preferred size)", workspace_bytes, vjp_min_bytes(L, P)); return EMAP_E_WORKSPACE; }
    char* ws = static_cast<char*>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = launch_absmax(d_udf, d_grad3, P, reinterpret_cast<uint32_t*>(ws + pl.off_absmax), st);
    if (rc) return rc;
    PointSource src;
    memset(&src, 0, sizeof(src));
    src.x = x;
    return run_vjp(L, packed, prec, src, P, d_udf, d_grad3, out, pl, ws, err_flags, st);
}

static size_t render_bwd_extra(const EmapRenderParams& p, size_t* off_du, size_t* off_dg, size_t* off_part) {
    const size_t N = (size_t)std::max(p.n_rays, 0);
    const int m = p.up_sample_steps > 0 ? p.n_importance / p.up_sample_steps : 0;
    const size_t S = (size_t)p.n_samples + (size_t)m * std::max(p.up_sample_steps, 0);
    size_t off = 0;
    *off_du = off; off += align256(N * S * 4);
    *off_dg = off; off += align256(N * S * 12);
    *off_part = off; off += align256(N * 16 + N * 8);      // composite_bwd's (N,4) partial sums + (N,2) per-ray maxima
    return off;
}

int emap_render_bwd_workspace_bytes(const EmapNetConfig* cfg, int prec, const EmapRenderParams* p, size_t* bytes) {
    NetLayout L;
    const int rc = build_layout(cfg, prec, &L);
    if (rc) return rc;
    if (!p || !bytes) { set_error("render_bwd_workspace_bytes: null pointer"); return EMAP_E_INVALID; }
    size_t a, b, c;
    const size_t extra = render_bwd_extra(*p, &a, &b, &c);
    const int m = p->up_sample_steps > 0 ? p->n_importance / p->up_sample_steps : 0;
    const int64_t S = (int64_t)p->n_samples + (int64_t)m * std::max(p->up_sample_steps, 0);
    *bytes = extra + plan_vjp(L, (int64_t)std::max(p->n_rays, 0) * S).total;
    return EMAP_OK;
}

int emap_render_bwd_absmax_offset(const EmapNetConfig* cfg, int prec, const EmapRenderParams* p, size_t* offset) {
    NetLayout L;
    const int rc = build_layout(cfg, prec, &L);
    if (rc) return rc;
    if (!p || !offset) { set_error("render_bwd_absmax_offset: null pointer"); return EMAP_E_INVALID; }
    size_t a, b, c;
    const size_t extra = render_bwd_extra(*p, &a, &b, &c);
    const int m = p->up_sample_steps > 0 ? p->n_importance / p->up_sample_steps : 0;
    const int64_t S = (int64_t)p->n_samples + (int64_t)m * std::max(p->up_sample_steps, 0);
    *offset = extra + plan_vjp(L, (int64_t)std::max(p->n_rays, 0) * S).off_absmax;
    return EMAP_OK;
}

int emap_render_bwd(const EmapNetConfig* cfg, const void* packed, int prec, const EmapRenderParams* p, const float* rays_o,
                    const float* rays_d, const float* depth_scale, const float* z_vals, const float* udf, const float* grad3,
                    const float* sample_dist_dev, const EmapCompositeGrads* g, const EmapParamGrads* out, void* workspace,
                    size_t workspace_bytes, int32_t* err_flags, void* stream) {
    return emap_render_bwd_staged(cfg, packed, prec, p, rays_o, rays_d, depth_scale, z_vals, udf, grad3, sample_dist_dev, g, out,
                                  workspace, workspace_bytes, err_flags, stream, 3);
}

int emap_render_bwd_staged(const EmapNetConfig* cfg, const void* packed, int prec, const EmapRenderParams* p, const float* rays_o,
                           const float* rays_d, const float* depth_scale, const float* z_vals, const float* udf, const float* grad3,
                           const float* sample_dist_dev, const EmapCompositeGrads* g, const EmapParamGrads* out, void* workspace,
                           size_t workspace_bytes, int32_t* err_flags, void* stream, int stages) {
    NetLayout L;
    int rc = build_layout(cfg, prec, &L);
    if (rc) return rc;
    if (!p || !g || !packed || !rays_o || !rays_d || !z_vals || !udf || !grad3 || !sample_dist_dev || !workspace) {
        set_error("render_bwd: null pointer");
        return EMAP_E_INVALID;
    }
    rc = check_param_grads(L, out, "render_bwd");
    if (rc) return rc;
    const int N = p->n_rays;
    if (N <= 0) return EMAP_OK;
    const int m = p->up_sample_steps > 0 && p->n_importance > 0 ? p->n_importance / p->up_sample_steps : 0;
    const int S = p->n_samples + m * (m > 0 ? p->up_sample_steps : 0);
    size_t o_du, o_dg, o_part;
    const size_t extra = render_bwd_extra(*p, &o_du, &o_dg, &o_part);
    const VjpPlan pl = plan_vjp(L, (int64_t)N * S, workspace_bytes > extra ? workspace_bytes - extra : 1);
    if (pl.chunk_tiles <= 0) { set_error("render_bwd: workspace %zu < %zu bytes (minimum; emap_render_bwd_workspace_bytes is the preferred size)", workspace_bytes, extra + vjp_min_bytes(L, (int64_t)N * S)); return EMAP_E_WORKSPACE; }
    char* ws = static_cast<char*>(workspace);
    char* vws = ws + extra;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* d_udf = reinterpret_cast<float*>(ws + o_du);
    float* d_grad = reinterpret_cast<float*>(ws + o_dg);
    if ((stages & 3) == 0) { set_error("render_bwd_staged: stages must have bit 0 and / or bit 1 set"); return EMAP_E_INVALID; }
    if (stages & 1) {
        // render_core's tail in reverse; it also leaves max|d_udf|, max|d_grad| for the sweep's range scale
        rc = launch_composite_bwd(rays_o, rays_d, z_vals, udf, grad3, depth_scale, N, S, sample_dist_dev, p, g, d_udf, d_grad,
                                  reinterpret_cast<float*>(ws + o_part), reinterpret_cast<uint32_t*>(vws + pl.off_absmax), st);
        if (rc) return rc;
    }
    if (!(stages & 2)) return EMAP_OK;
    PointSource fin;
    memset(&fin, 0, sizeof(fin));
    fin.rays_o = rays_o; fin.rays_d = rays_d; fin.z = z_vals; fin.n_per_ray = S; fin.mid = 1; fin.sample_dist = sample_dist_dev;
    return run_vjp(L, packed, prec, fin, (int64_t)N * S, d_udf, d_grad, out, pl, vws, err_flags, st);
}

int emap_sample_rays(const EmapRayDataset* ds, int img_idx, int batch, int importance, uint64_t seed, uint64_t offset,
                     uint64_t* counter_dev, const int64_t* pixels_in, const EmapRayBatch* out, void* stream) {
    return launch_sample_rays(ds, img_idx, batch, importance, seed, offset, counter_dev, pixels_in, out, static_cast<hipStream_t>(stream));
}

int emap_train_stats(const float* edge, const float* true_edge, const float* scalars, int N, float d_scale, float* d_edge,
                     float* stats5, void* stream) {
    return launch_train_stats(edge, true_edge, scalars, N, d_scale, d_edge, stats5, static_cast<hipStream_t>(stream));
}
int emap_train_loss(const float* stats5, float w_over_n, float igr_weight, float igr_ns_weight, float* out2, void* stream) {
    return launch_train_loss(stats5, w_over_n, igr_weight, igr_ns_weight, out2, static_cast<hipStream_t>(stream));
}
int emap_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* step_dev, int64_t n, int64_t n_geo,
                   float lr_geo, float lr, double beta1, double beta2, float eps, void* stream) {
    return launch_adam(params, grads, exp_avg, exp_avg_sq, step_dev, n, n_geo, lr_geo, lr, beta1, beta2, eps, nullptr, nullptr, static_cast<hipStream_t>(stream));
}
int emap_adam_step_masked(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* step_dev, int64_t n, int64_t n_geo,
                          float lr_geo, float lr, double beta1, double beta2, float eps, const float* tail_mask, float* tail_step, void* stream) {
    return launch_adam(params, grads, exp_avg, exp_avg_sq, step_dev, n, n_geo, lr_geo, lr, beta1, beta2, eps, tail_mask, tail_step,
                       static_cast<hipStream_t>(stream));
}

int emap_null_direction(const float* grads, int64_t n, int k, float* dir, void* stream) {
    if (n < 0) { set_error("null_direction: negative n"); return EMAP_E_INVALID; }
    if (n > 0 && (!grads || !dir)) { set_error("null_direction: null pointer"); return EMAP_E_INVALID; }
    return launch_null_direction(grads, n, k, dir, static_cast<hipStream_t>(stream));
}

int emap_profile_enable(int on) {
    if (on && !g_prof_init) {
        for (int k = 0; k < PROF_KERNELS; ++k)
            for (int i = 0; i < PROF_MAX; ++i)
                for (int e = 0; e < 2; ++e)
                    if (hipEventCreate(&g_prof_ev[k][i][e]) != hipSuccess) { set_error("hipEventCreate failed"); return EMAP_E_LAUNCH; }
        g_prof_init = true;
    }
    g_prof_on = on != 0;
    if (on) for (int k = 0; k < PROF_KERNELS; ++k) g_prof_n[k] = 0;
    // shader-clock stamps of the two big MLP kernels (udf_mlp_kernel.inc:clock_stamp): a device buffer on the current device while
    // profiling is on; the launchers pass null otherwise
    int cur_dev = -1;
    if (on && hipGetDevice(&cur_dev) != hipSuccess) { set_error("hipGetDevice failed"); return EMAP_E_LAUNCH; }
    if (on && g_clk_dev && cur_dev != emap::g_prof_clk_device) { (void)hipFree(g_clk_dev); g_clk_dev = nullptr; }   // profiling moved to another device
    if (on && !g_clk_dev) {
        emap::g_prof_clk_device = cur_dev;
        if (hipMalloc(reinterpret_cast<void**>(&g_clk_dev), 8 * sizeof(long long)) != hipSuccess) { g_clk_dev = nullptr; set_error("hipMalloc failed"); return EMAP_E_LAUNCH; }
    }
    if (on && hipMemset(g_clk_dev, 0, 8 * sizeof(long long)) != hipSuccess) { set_error("hipMemset failed"); return EMAP_E_LAUNCH; }
    emap::g_prof_clk = on ? g_clk_dev : nullptr;
    return EMAP_OK;
}

int emap_profile_read_clock(int which, float* mhz_host) {
    if (!mhz_host || (which != 0 && which != 1)) { set_error("profile_read_clock: bad argument"); return EMAP_E_INVALID; }
    *mhz_host = 0.f;
    if (!g_clk_dev) return EMAP_OK;
    long long h[8];
    if (hipMemcpy(h, g_clk_dev, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { set_error("hipMemcpy failed"); return EMAP_E_LAUNCH; }
    const long long* s = h + 4 * which;        // {memtime, realtime} at entry, {memtime, realtime} at exit, of the LAST launch
    const long long dt = s[2] - s[0], dr = s[3] - s[1];
    if (dt > 0 && dr > 0) *mhz_host = (float)((double)dt / (double)dr * 100.0);
    return EMAP_OK;
}

int emap_profile_read_kernel(int which, float* total_ms_host, int* launches_host) {
    if (!total_ms_host || !launches_host || which < 0 || which >= PROF_KERNELS) { set_error("profile_read: bad argument"); return EMAP_E_INVALID; }
    float tot = 0.f;
    for (int i = 0; i < g_prof_n[which]; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof_ev[which][i][0], g_prof_ev[which][i][1]) != hipSuccess) { set_error("hipEventElapsedTime failed (region not synchronised?)"); return EMAP_E_LAUNCH; }
        tot += ms;
    }
    *total_ms_host = tot;
    *launches_host = g_prof_n[which];
    return EMAP_OK;
}

int emap_profile_read(float* total_ms_host, int* launches_host) { return emap_profile_read_kernel(0, total_ms_host, launches_host); }

/* host-only helper used by the CPU tests: the u grid of sample_pdf / the coarse z grid */
void emap_linspace_host(float start, float end, int steps, float* out_host) { linspace_host(start, end, steps, out_host); }

}  // extern "C"
