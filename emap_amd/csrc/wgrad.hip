// wgrad.hip - weight-gradient GEMMs of the training backward (SURVEY.md par. 8 f1), their reduction and the weight-norm
// VJP.  Replaces what autograd does for the Linear layers of UDFNetwork under loss.backward()
// (reference src/models/udf_model.py:73-74,90-110 differentiated twice, src/runner/runner_udf.py:166-167).
//
//   wgrad_kernel   : dW_l = Zbar_l . A_{l-1}^T  over the columns (points x {value, tangent}) of a K-slice of tiles, straight
//                    from the fragments udf_mlp_vjp_kernel left in global memory (both operands already have K = columns, 8
//                    per lane, so every load is one coalesced 1 KiB fragment and feeds v_mfma_f32_16x16x32 directly);
//                    db_l = sum over the value columns of Zbar_l (VALU, from the same fragments).
//                    *Bound: HBM* - 16 KiB of stash per point are read once (DESIGN.md par. 3.5).
//   wgrad_reduce   : sums the K-slices in a fixed order (deterministic), maps fragment order back to the natural
//                    [out][in] order (incl. the PE slot permutation), undoes the range scale K and the 1/sqrt2 of the
//                    skip concat, and applies the weight-norm VJP  W = g v/||v||  ->  dg, dv  (one wave per output row).
//   absmax_kernel  : max|du|, max|dg| of a launch (the range scale K of the sweep).
#include "emap_common.h"
#include <string.h>

namespace emap {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16w(const bf16x8& a, const bf16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16w(const f16x8& a, const f16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

__global__ __launch_bounds__(256) void absmax_kernel(const float* du, const float* dg, long long P, uint32_t* out) {
    float mu = 0.f, mg = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
        const float u = fabsf(du[i]);
        const float g = fmaxf(fmaxf(fabsf(dg[3 * i]), fabsf(dg[3 * i + 1])), fabsf(dg[3 * i + 2]));
        mu = (u < 3.0e38f) ? fmaxf(mu, u) : mu;   // NaN / inf do not poison the scale (they poison the result, as in autograd)
        mg = (g < 3.0e38f) ? fmaxf(mg, g) : mg;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mu = fmaxf(mu, __shfl_xor(mu, off)); mg = fmaxf(mg, __shfl_xor(mg, off)); }
    if ((threadIdx.x & 63) == 0) {   // non-negative floats order like their bit patterns
        atomicMax(out, __builtin_bit_cast(uint32_t, mu));
        atomicMax(out + 1, __builtin_bit_cast(uint32_t, mg));
    }
}

// ---------------------------------------------------------------------------------------------
// weight-gradient GEMM
// ---------------------------------------------------------------------------------------------
// One job = (layer, column part): part 0 = the hidden features of A level l (output of layer l-1), part 1 = the PE slots
// (A level 0; layers 0 and skip).  A job's tiles are cut into n_slices contiguous K-slices; workgroup = one slice of
// one job; 8 waves, wave w owns row tiles w and w + 8 and all (<= 16) column tiles: up to 128 accumulator VGPRs.
struct WgradArgs {
    const char* stash_a;
    const char* stash_z;
    float* partial;
    int32_t n_tiles, n_jobs, a_tile_kb, z_tile_kb;
    int32_t accumulate;         // 1: add to the partials already there (second and later chunks of a launch)
    WgradJob job[WGRAD_MAX_JOBS];
};

template <class V8>
__global__ __launch_bounds__(512, 1) void wgrad_kernel(const WgradArgs a) {
    int ji = 0;
    while (ji + 1 < a.n_jobs && (int)blockIdx.x >= a.job[ji + 1].first_wg) ++ji;
    const WgradJob J = a.job[ji];
    const int slice = (int)blockIdx.x - J.first_wg;
    const int t0 = (int)(((long long)a.n_tiles * slice) / J.n_slices), t1 = (int)(((long long)a.n_tiles * (slice + 1)) / J.n_slices);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nrt = (wave < J.z_rt ? 1 : 0) + (wave + 8 < J.z_rt ? 1 : 0);
    constexpr int CTM = 16;
    f32x4 acc[2][CTM];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < CTM; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};
    if (nrt > 0) {
        const char* zb = a.stash_z + (size_t)J.z_off * 1024 + (size_t)wave * 2048 + lane * 16;
        const char* ab = a.stash_a + (size_t)J.a_off * 1024 + lane * 16;
        for (int tile = t0; tile < t1; ++tile) {
            const char* zt = zb + (size_t)tile * ((size_t)a.z_tile_kb * 1024);
            const char* at = ab + (size_t)tile * ((size_t)a.a_tile_kb * 1024);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                V8 zf[2];
                zf[0] = *reinterpret_cast<const V8*>(zt + s * 1024);
                zf[1] = (nrt > 1) ? *reinterpret_cast<const V8*>(zt + 8 * 2048 + s * 1024) : V8{};
#pragma unroll
                for (int i = 0; i < 2; ++i)   // value columns are K slots e with e % 4 < 2
                    bsum[i] += ((float)zf[i][0] + (float)zf[i][1]) + ((float)zf[i][4] + (float)zf[i][5]);
#pragma unroll
                for (int c = 0; c < CTM; ++c) {
                    if (c < J.a_ct) {
                        const V8 af = *reinterpret_cast<const V8*>(at + c * 2048 + s * 1024);
                        acc[0][c] = mfma16w(zf[0], af, acc[0][c]);
                        acc[1][c] = mfma16w(zf[1], af, acc[1][c]);
                    }
                }
            }
        }
    }
    // partial block [slice][row tile 0..15][a_ct][64 lanes x 4]
    float* pb = a.partial + J.part_off + (size_t)slice * 16 * J.a_ct * 256;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rt = wave + 8 * i;
        if (rt < J.z_rt) {
#pragma unroll
            for (int c = 0; c < CTM; ++c) {
                if (c < J.a_ct) {
                    f32x4* dst = reinterpret_cast<f32x4*>(pb + ((size_t)rt * J.a_ct + c) * 256 + lane * 4);
                    *dst = a.accumulate ? (*dst + acc[i][c]) : acc[i][c];
                }
            }
        }
    }
    if (J.bias_off >= 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int rt = wave + 8 * i;
            if (rt < J.z_rt && lane < 16) {
                float* dst = a.partial + J.bias_off + (size_t)slice * 256 + rt * 16 + lane;
                *dst = a.accumulate ? (*dst + v) : v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// reduction over K-slices + un-permutation + weight-norm VJP
// ---------------------------------------------------------------------------------------------
struct ReduceArgs {
    const float* partial;
    const uint32_t* absmax;
    float grad_scale;           // extra factor on every gradient (1/world for data-parallel means; 1 otherwise)
    int32_t accumulate;         // 1: add to dg/dv/db instead of overwriting
    int32_t weight_norm;        // 0: dv = dW (g is ignored, dg untouched)
    int32_t n_lin, H, d0, multires, skip_l;
    int32_t row_off[EMAP_MAX_LIN + 1];   // first global row of layer l (prefix sum of out_dim)
    int32_t out_dim[EMAP_MAX_LIN], in_prev[EMAP_MAX_LIN], has_pe[EMAP_MAX_LIN];
    int32_t job_h[EMAP_MAX_LIN], job_pe[EMAP_MAX_LIN];   // indices into job[] or -1
    WgradJob job[WGRAD_MAX_JOBS];
    const float* g[EMAP_MAX_LIN];
    const float* v[EMAP_MAX_LIN];
    float* dg[EMAP_MAX_LIN];
    float* dv[EMAP_MAX_LIN];
    float* db[EMAP_MAX_LIN];
};

__device__ __forceinline__ float vjp_scale_from_w(const uint32_t* absmax) {   // must equal udf_mlp_vjp.inc:vjp_scale_from
    const float mu = __builtin_bit_cast(float, absmax[0]), mg = __builtin_bit_cast(float, absmax[1]);
    const float m = fmaxf(mg, mu * (1.0f / 64.0f));
    if (!(m > 1e-30f) || !(m < 1e30f)) return 1.0f;
    int e;
    (void)frexpf(m, &e);
    return ldexpf(1.0f, -e);
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// element (row m of row tile rt, column n of column tile ct) of a job's summed partial
__device__ __forceinline__ float partial_at(const float* partial, const WgradJob& J, int rt, int m, int ct, int n) {
    const size_t e = ((size_t)rt * J.a_ct + ct) * 256 + ((m >> 2) * 16 + n) * 4 + (m & 3);
    const size_t stride = (size_t)16 * J.a_ct * 256;
    float s = 0.f;
    for (int q = 0; q < J.n_slices; ++q) s += partial[J.part_off + q * stride + e];
    return s;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const ReduceArgs a) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= a.row_off[a.n_lin]) return;
    int l = 0;
    while (row >= a.row_off[l + 1]) ++l;
    const int o = row - a.row_off[l];
    const int in_prev = a.in_prev[l];
    const int n_in = in_prev + (a.has_pe[l] ? a.d0 : 0);
    const float inv_k = a.grad_scale / vjp_scale_from_w(a.absmax);
    const float mult = (l == a.skip_l) ? 0.70710678118654752440f : 1.0f;
    const int rt = o >> 4, m = o & 15;
    constexpr int MAXC = 6;     // ceil((256 + 63) / 64)
    float dw[MAXC], vv[MAXC];
    float dot = 0.f, nrm = 0.f;
    const float* vrow = a.v[l] + (size_t)o * n_in;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int k = lane + 64 * c;
        dw[c] = 0.f; vv[c] = 0.f;
        if (k < n_in) {
            float s;
            if (k < in_prev) {
                s = partial_at(a.partial, a.job[a.job_h[l]], rt, m, k >> 4, k & 15);
            } else {
                // natural PE column -> slot (sp, g, e) of the PE block (udf_mlp.hip:pack_kernel) -> row tile 2*sp + e/4, row 4g + e%4
                const int pc = k - in_prev, M = a.multires;
                int ang, kind;
                if (pc < 3) { ang = (pc == 2) ? 3 * M + 1 : 3 * M; kind = (pc == 1) ? 1 : 0; }
                else { const int q = pc - 3, kk = q / 6, r = q - 6 * kk; kind = r / 3; ang = 3 * kk + (r - 3 * kind); }
                const int g = ang >> 3, qq = ang & 7, sp = qq >> 2, e = 2 * (qq & 3) + kind;
                const int prow = 4 * g + (e & 3), ptile = 2 * sp + (e >> 2);
                s = partial_at(a.partial, a.job[a.job_pe[l]], rt, m, ptile, prow);
            }
            dw[c] = s * inv_k * mult;      // d/dW_l of the reference's Linear (the packed weight holds W_l/sqrt2 for the skip layer)
            vv[c] = vrow[k];
            dot = fmaf(dw[c], vv[c], dot);
            nrm = fmaf(vv[c], vv[c], nrm);
        }
    }
    dot = wave_sum_f(dot);
    nrm = wave_sum_f(nrm);
    float* dvrow = a.dv[l] + (size_t)o * n_in;
    if (a.weight_norm) {
        const float n = sqrtf(nrm), gg = a.g[l][o];
        const float c1 = gg / n, c2 = dot / nrm;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int k = lane + 64 * c;
            if (k < n_in) {
                const float r = c1 * (dw[c] - c2 * vv[c]);
                dvrow[k] = a.accumulate ? dvrow[k] + r : r;
            }
        }
        if (lane == 0) { const float r = dot / n; a.dg[l][o] = a.accumulate ? a.dg[l][o] + r : r; }
    } else {
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int k = lane + 64 * c;
            if (k < n_in) dvrow[k] = a.accumulate ? dvrow[k] + dw[c] : dw[c];
        }
    }
    if (lane == 0) {
        const int jb = (a.job_h[l] >= 0) ? a.job_h[l] : a.job_pe[l];
        const WgradJob& J = a.job[jb];
        float s = 0.f;
        for (int q = 0; q < J.n_slices; ++q) s += a.partial[J.bias_off + (size_t)q * 256 + o];
        const float r = s * inv_k;
        a.db[l][o] = a.accumulate ? a.db[l][o] + r : r;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// jobs and their K-slices for a budget of `wg_budget` workgroups; returns the number of floats of the partial buffer
size_t plan_wgrad(const NetLayout& L, const VjpLayout& V, int wg_budget, WgradJob* jobs, int* n_jobs, int* job_h, int* job_pe,
                  int* total_wg) {
    int n = 0;
    double cost[WGRAD_MAX_JOBS], tot = 0;
    for (int l = 0; l < L.n_lin; ++l) {
        job_h[l] = job_pe[l] = -1;
        const LayerDesc& d = L.layer[l];
        for (int part = 0; part < 2; ++part) {
            if (part == 0 && d.h_ks == 0) continue;
            if (part == 1 && d.pe_ks == 0) continue;
            WgradJob& J = jobs[n];
            J.layer = l; J.part = part;
            J.z_off = V.z_off[l]; J.z_rt = V.z_rt[l];
            J.a_off = (part == 0) ? V.a_off[l] : V.a_off[0];
            J.a_ct = (part == 0) ? (d.in_prev + 15) / 16 : 2 * PE_KS;
            J.bias_off = -1;
            cost[n] = (double)((J.z_rt + 7) / 8) * (1.0 + J.a_ct);   // per tile: Z fragment loads + A fragments streamed per wave
            tot += cost[n];
            (part == 0 ? job_h : job_pe)[l] = n;
            ++n;
        }
    }
    int wg = 0;
    size_t floats = 0;
    for (int i = 0; i < n; ++i) {
        int s = (int)(cost[i] / tot * wg_budget);
        if (s < 1) s = 1;
        jobs[i].n_slices = s;
        jobs[i].first_wg = wg;
        wg += s;
        jobs[i].part_off = (int32_t)floats;
        floats += (size_t)s * 16 * jobs[i].a_ct * 256;
    }
    for (int l = 0; l < L.n_lin; ++l) {   // the bias partials ride on one job per layer
        WgradJob& J = jobs[(job_h[l] >= 0) ? job_h[l] : job_pe[l]];
        J.bias_off = (int32_t)floats;
        floats += (size_t)J.n_slices * 256;
    }
    *n_jobs = n;
    *total_wg = wg;
    return floats;
}

int launch_absmax(const float* du, const float* dg, int64_t P, uint32_t* out, hipStream_t st) {
    if (hipMemsetAsync(out, 0, 8, st) != hipSuccess) { set_error("hipMemsetAsync failed"); return EMAP_E_LAUNCH; }
    if (P <= 0) return EMAP_OK;
    const int grid = (int)((P + 255) / 256 < 512 ? (P + 255) / 256 : 512);
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, st, du, dg, (long long)P, out);
    return check_launch("absmax");
}

int launch_wgrad(const NetLayout& L, const VjpLayout& V, const WgradJob* jobs, int n_jobs, int total_wg, const char* stash_a,
                 const char* stash_z, float* partial, int n_tiles, int accumulate, hipStream_t st) {
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.stash_a = stash_a; a.stash_z = stash_z; a.partial = partial;
    a.n_tiles = n_tiles; a.n_jobs = n_jobs; a.a_tile_kb = V.a_tile_kb; a.z_tile_kb = V.z_tile_kb; a.accumulate = accumulate;
    for (int i = 0; i < n_jobs; ++i) a.job[i] = jobs[i];
    if (L.is_f16) hipLaunchKernelGGL(wgrad_kernel<f16x8>, dim3(total_wg), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(wgrad_kernel<bf16x8>, dim3(total_wg), dim3(512), 0, st, a);
    return check_launch("wgrad");
}

int launch_wgrad_reduce(const NetLayout& L, const WgradJob* jobs, int n_jobs, const int* job_h, const int* job_pe,
                        const float* partial, const uint32_t* absmax, const float* const* g, const float* const* v,
                        float* const* dg, float* const* dv, float* const* db, int weight_norm, int accumulate, float grad_scale,
                        hipStream_t st) {
    ReduceArgs a;
    memset(&a, 0, sizeof(a));
    a.partial = partial; a.absmax = absmax; a.grad_scale = grad_scale; a.accumulate = accumulate; a.weight_norm = weight_norm;
    a.n_lin = L.n_lin; a.H = L.H; a.d0 = L.d0; a.multires = L.multires; a.skip_l = L.skip_l;
    int rows = 0;
    for (int l = 0; l < L.n_lin; ++l) {
        a.row_off[l] = rows; rows += L.layer[l].out_dim;
        a.out_dim[l] = L.layer[l].out_dim; a.in_prev[l] = L.layer[l].in_prev; a.has_pe[l] = L.layer[l].pe_ks ? 1 : 0;
        a.job_h[l] = job_h[l]; a.job_pe[l] = job_pe[l];
        a.g[l] = g ? g[l] : nullptr; a.v[l] = v[l]; a.dg[l] = dg ? dg[l] : nullptr; a.dv[l] = dv[l]; a.db[l] = db[l];
    }
    a.row_off[L.n_lin] = rows;
    for (int i = 0; i < n_jobs; ++i) a.job[i] = jobs[i];
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, a);
    return check_launch("wgrad_reduce");
}

}  // namespace emap
