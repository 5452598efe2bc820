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
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>

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
    // precise weight gradients (three passes per chunk: hi x hi, Z_hi x A_lo, Z_lo x A_hi): this pass's factor (1 or 1 / LO_SCALE) and whether
    // its Z operand contributes to the bias sums (not in the Z_hi x A_lo pass: the hi x hi pass has counted Z_hi already)
    float scale;
    int32_t no_bias;
    WgradJob job[WGRAD_MAX_JOBS];
};

// The K loop is software-pipelined by hand with the same device the MLP kernels use (inline-asm loads retired by counted
// s_waitcnt vmcnt(N) statements that name their destination registers; hipcc otherwise sinks every load to its first use and
// the loop runs at one HBM latency per fragment: measured 1.8 ms instead of 0.3 ms for 65 536 points).  A step = one K-step
// of one tile (2 Z fragments + CT A fragments).  Every A register is reloaded for the NEXT step right after the two MFMAs that
// consumed it and the next step's Z fragments are issued at the top of the step, so CT + 2 coalesced 1 KiB loads per wave are in
// flight at all times.  In-flight order at the top of a step: Z0, Z1, A[0..CT-1]  ->  vmcnt(CT-1) retires Z and A[0]; inside
// the step A[c] has CT+1 younger loads behind it.  Column tiles beyond a_ct and a missing second row tile are clamped to valid
// addresses: their accumulators are garbage and never stored.  scripts/isa_lint.py checks the register discipline.
template <class V8>
__device__ __forceinline__ void wg_load(V8& dst, int voff, const char* sbase) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");   // s_nop: see asm_gload16
}
template <int N, class V8>
__device__ __forceinline__ void wg_wait(V8& r0) { asm volatile("s_waitcnt vmcnt(%c1)" : "+v"(r0) : "i"(N) : "memory"); }
template <int N, class V8>
__device__ __forceinline__ void wg_wait3(V8& r0, V8& r1, V8& r2) { asm volatile("s_waitcnt vmcnt(%c3)" : "+v"(r0), "+v"(r1), "+v"(r2) : "i"(N) : "memory"); }
__device__ __forceinline__ const char* wg_uniform(const char* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

template <int... Is, class F>
__device__ __forceinline__ void wg_static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void wg_static_for(F&& f) { wg_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

template <int CT>
__device__ __forceinline__ void wgrad_store(const WgradArgs& a, const WgradJob& J, int slice, int wave, int lane, f32x4 (&acc)[2][CT], float (&bsum)[2]) {
    // partial block [slice][row tile 0..15][a_ct][64 lanes x 4]
    float* pb = a.partial + J.part_off + (size_t)slice * 16 * J.a_ct * 256;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rt = wave + 8 * i;
        if (rt < J.z_rt) {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (c < J.a_ct) {
                    f32x4* dst = reinterpret_cast<f32x4*>(pb + ((size_t)rt * J.a_ct + c) * 256 + lane * 4);
                    *dst = a.accumulate ? (*dst + acc[i][c] * a.scale) : acc[i][c] * a.scale;
                }
            }
        }
    }
    if (J.bias_off >= 0 && !a.no_bias) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v = bsum[i] * a.scale;
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
// weight-gradient GEMM, operands staged through LDS by DMA
// ---------------------------------------------------------------------------------------------
// If every one of the 8 waves pulled ALL column-tile fragments of a step itself (round 2's first version; they share them, only
// the two Z fragments are a wave's own): 8 x 18 KiB per step through the CU's 64 B/clk vector-memory path = 2.3 k cycles against 1.0 k cycles
// of MFMA per SIMD - L1-bandwidth bound by 2x, and every 1 KiB load costs its wave ~50 issue cycles.  Here each fragment of a
// step is fetched ONCE per workgroup, straight into LDS (global_load_lds_dwordx4: lane-linear 1 KiB, exactly the fragment
// layout), 2 A + 2 Z fragments per wave and step, and read back with ds_read_b128 (256 B/clk).  Ring of NST = D + 1 stages, D
// steps in flight (no registers are involved, so nothing the compiler could copy: loads stay in flight across the loop
// back-edge); per step: s_waitcnt vmcnt (my fragments of this step have landed) -> s_barrier (everyone's have, and everyone
// has finished reading the stage about to be refilled) -> issue step u + D -> 18 ds_read_b128 + 32 MFMAs.
#ifndef EMAP_REDUCE_WAVES
#define EMAP_REDUCE_WAVES 4
#endif
#ifndef EMAP_WGRAD_DEPTH
#define EMAP_WGRAD_DEPTH 3
#endif
constexpr int WGRAD_DEPTH = EMAP_WGRAD_DEPTH;   // steps (32 KiB each) in flight per workgroup; the ring has WGRAD_DEPTH + 1 stages (<= 160 KiB of LDS)
__device__ __forceinline__ void wg_dma16(unsigned lds_dst, const char* gsrc) {
    unsigned keep;   // M0 = LDS byte address of lane 0's 16 bytes; written in the statement that uses it (the compiler owns M0)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <class V8, int NB>
__device__ __forceinline__ void wgrad_lds_run(const WgradArgs& a, const WgradJob& J, int slice, int wave, int lane, char* smem) {
    constexpr int CT = 4 * NB;
    constexpr int NA = (CT + 7) / 8;              // A fragments a wave fetches per step
    constexpr int NLD = NA + 2;                   // DMA loads per wave and step
    constexpr int D = WGRAD_DEPTH, NST = D + 1;
    constexpr int STAGE = (CT + 16) * 1024;       // A[CT] then Z[16 row tiles]
    const int t0 = (int)(((long long)a.n_tiles * slice) / J.n_slices), t1 = (int)(((long long)a.n_tiles * (slice + 1)) / J.n_slices);
    const int nsteps = 2 * (t1 - t0);
    f32x4 acc[2][CT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};
    if (nsteps > 0) {
        const size_t zts = (size_t)a.z_tile_kb * 1024, ats = (size_t)a.a_tile_kb * 1024;
        const char* zsrc = a.stash_z + (size_t)J.z_off * 1024 + lane * 16;
        const char* asrc = a.stash_a + (size_t)J.a_off * 1024 + lane * 16;
        const unsigned lds0 = (unsigned)(size_t)smem;
        // what this wave fetches: A fragments 8i + wave (folded into the valid range: duplicates write identical bytes) and its Z rows
        int ca[NA], zr[2];
#pragma unroll
        for (int i = 0; i < NA; ++i) ca[i] = 8 * i + ((8 * i + 8 <= CT) ? wave : wave % (CT - 8 * i));
        zr[0] = wave < J.z_rt ? wave : J.z_rt - 1;
        zr[1] = wave + 8 < J.z_rt ? wave + 8 : J.z_rt - 1;
        auto issue = [&](int u) __attribute__((always_inline)) {
            const int uu = u < nsteps ? u : nsteps - 1;          // past the end: a harmless re-fetch keeps the load count uniform
            const int tile = t0 + (uu >> 1), sh = (uu & 1) * 1024;
            const unsigned st = lds0 + (unsigned)(u % NST) * STAGE;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int cs = ca[i] < J.a_ct ? ca[i] : J.a_ct - 1;
                wg_dma16(__builtin_amdgcn_readfirstlane(st + ca[i] * 1024), asrc + (size_t)tile * ats + cs * 2048 + sh);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
                wg_dma16(__builtin_amdgcn_readfirstlane(st + (CT + zr[i]) * 1024), zsrc + (size_t)tile * zts + zr[i] * 2048 + sh);
        };
        for (int u = 0; u < D; ++u) issue(u);
        for (int u = 0; u < nsteps; ++u) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NLD * (D - 1)) : "memory");
            __builtin_amdgcn_s_barrier();
            issue(u + D);
            const char* st = smem + (size_t)(u % NST) * STAGE + lane * 16;
            V8 z[2];
            z[0] = *reinterpret_cast<const V8*>(st + (CT + zr[0]) * 1024);
            z[1] = *reinterpret_cast<const V8*>(st + (CT + zr[1]) * 1024);
            {   // the value columns of a tile are exactly its K-step 0 (udf_mlp_vjp.inc: K slot <-> column map)
                const float m = (u & 1) ? 0.0f : 1.0f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    bsum[i] = fmaf(m, (((float)z[i][0] + (float)z[i][1]) + ((float)z[i][2] + (float)z[i][3])) +
                                      (((float)z[i][4] + (float)z[i][5]) + ((float)z[i][6] + (float)z[i][7])), bsum[i]);
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const V8 af = *reinterpret_cast<const V8*>(st + c * 1024);
                acc[0][c] = mfma16w(z[0], af, acc[0][c]);
                acc[1][c] = mfma16w(z[1], af, acc[1][c]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail re-fetches: no DMA may outlive the workgroup's LDS
    }
    wgrad_store<CT>(a, J, slice, wave, lane, acc, bsum);
}

#ifdef EMAP_WGRAD_TIMING
static __device__ long long emap_wg_times[1024 * 2];   // debug builds: per workgroup {job, s_memrealtime ticks (100 MHz)}
#endif
template <class V8>
__global__ __launch_bounds__(512, 1) void wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char wg_smem[];
#ifdef EMAP_WGRAD_TIMING
    const long long wg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    int ji = 0;
    while (ji + 1 < a.n_jobs && (int)blockIdx.x >= a.job[ji + 1].first_wg) ++ji;
    const WgradJob J = a.job[ji];
    const int slice = (int)blockIdx.x - J.first_wg;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nb = (J.a_ct + 3) >> 2;
    if (nb <= 1) wgrad_lds_run<V8, 1>(a, J, slice, wave, lane, wg_smem);
    else if (nb == 2) wgrad_lds_run<V8, 2>(a, J, slice, wave, lane, wg_smem);
    else if (nb == 3) wgrad_lds_run<V8, 3>(a, J, slice, wave, lane, wg_smem);
    else wgrad_lds_run<V8, 4>(a, J, slice, wave, lane, wg_smem);
#ifdef EMAP_WGRAD_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 1024) { emap_wg_times[2 * blockIdx.x] = ji; emap_wg_times[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - wg_t0; }
#endif
}

// ---------------------------------------------------------------------------------------------
// reduction over K-slices + un-permutation + weight-norm VJP
// ---------------------------------------------------------------------------------------------
struct ReduceArgs {
    const float* partial;
    const uint32_t* absmax;
    const float* ldot;          // per-tile exact sums for the last layer's dot(dW, v) (udf_mlp_vjp.inc)
    int32_t n_tiles;
    float grad_scale;           // extra factor on every gradient (1/world for data-parallel means; 1 otherwise)
    int32_t accumulate;         // 1: add to dg/dv/db instead of overwriting
    int32_t weight_norm;        // 0: dv = dW (g is ignored, dg untouched)
    int32_t n_lin, H, d0, multires, skip_l;
    int32_t row_off[EMAP_MAX_LIN + 1];   // first 4-row group of layer l (prefix sum of ceil(out_dim / 4))
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

// 4 consecutive output rows per wave: they are the 4 accumulator registers of one lane of the MFMA output, i.e. ONE 16-byte
// word of the partial blocks per (column, slice) - every load is a float4 and a wave reads 256-byte runs.
constexpr int RED_WAVES = EMAP_REDUCE_WAVES;   // waves of a row group: each sums every RED_WAVES-th K-slice
static_assert(RED_WAVES >= 2 && RED_WAVES <= 16, "EMAP_REDUCE_WAVES: the cross-wave sum needs 2..16 waves (part[RED_WAVES - 1] in LDS)");
__global__ __launch_bounds__(RED_WAVES * 64) void wgrad_reduce_kernel(const ReduceArgs a) {
    // one workgroup per group of 4 rows: its 4 waves each sum every 4th K-slice (there are only ~500 groups: with one wave per group
    // the kernel was a latency chain on two waves per CU), wave 0 adds the four partial sums in a fixed order and finishes the rows
    __shared__ f32x4 part[RED_WAVES - 1][6][64];
    const int grp = blockIdx.x;
    const int wv = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    if (grp >= a.row_off[a.n_lin]) return;       // row_off counts groups of 4 rows here
    int l = 0;
    while (grp >= a.row_off[l + 1]) ++l;
    const int o0 = 4 * (grp - a.row_off[l]);
    const int out_dim = a.out_dim[l];
    const int in_prev = a.in_prev[l];
    const int n_in = in_prev + (a.has_pe[l] ? a.d0 : 0);
    const float inv_k = a.grad_scale / vjp_scale_from_w(a.absmax);
    const float mult = (l == a.skip_l) ? 0.70710678118654752440f : 1.0f;
    const int rt = o0 >> 4, mq = (o0 & 15) >> 2;
    // the last layer's Z level holds its single real row twice: row 0 = hi part, row 1 = lo part of the seeds (udf_mlp_vjp.inc)
    const bool last = (l == a.n_lin - 1);
    constexpr int MAXC = 6;     // ceil((256 + 63) / 64)
    float dw[4][MAXC], vv[4][MAXC];
    float dot[4] = {0.f, 0.f, 0.f, 0.f}, nrm[4] = {0.f, 0.f, 0.f, 0.f};
    // where this lane's columns live: one float4 per (column, slice); all columns' loads of a slice are issued together and the
    // slice loop is unrolled, so ~20 independent 16-byte loads are in flight per lane (there are only ~2 waves per CU: the
    // kernel is a latency chain, not a bandwidth problem)
    const float* src[MAXC];
    size_t str[MAXC];
    int ns[MAXC];
    int nmax = 0;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int k = lane + 64 * c;
        src[c] = a.partial; str[c] = 0; ns[c] = 0;
        if (k < n_in) {
            int jb, ct, n;
            if (k < in_prev) { jb = a.job_h[l]; ct = k >> 4; n = k & 15; }
            else {
                // natural PE column -> slot (sp, g, e) of the PE block (udf_mlp.hip:pack_body) -> row tile 2*sp + e/4, row 4g + e%4
                const int pc = k - in_prev, M = a.multires;
                int ang, kind;
                if (pc < 3) { ang = (pc == 2) ? 3 * M + 1 : 3 * M; kind = (pc == 1) ? 1 : 0; }
                else { const int q = pc - 3, kk = q / 6, r = q - 6 * kk; kind = r / 3; ang = 3 * kk + (r - 3 * kind); }
                const int g = ang >> 3, qq = ang & 7, sp = qq >> 2, e = 2 * (qq & 3) + kind;
                jb = a.job_pe[l]; ct = 2 * sp + (e >> 2); n = 4 * g + (e & 3);
            }
            const WgradJob& J = a.job[jb];
            src[c] = a.partial + J.part_off + (((size_t)rt * J.a_ct + ct) * 64 + mq * 16 + n) * 4;
            str[c] = (size_t)16 * J.a_ct * 256;
            ns[c] = J.n_slices;
        }
        nmax = max(nmax, ns[c]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off));
    f32x4 sum[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) sum[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int q = wv; q < nmax; q += RED_WAVES) {       // fixed order per column and wave: deterministic
        f32x4 v[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c)         // unconditional loads (a column with fewer slices re-reads its last one): they issue as one batch
            v[c] = *reinterpret_cast<const f32x4*>(src[c] + (size_t)min(q, max(ns[c] - 1, 0)) * str[c]);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            sum[c] += (q < ns[c]) ? v[c] : z;
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c) part[wv - 1][c][lane] = sum[c];
    }
    __syncthreads();
    if (wv > 0) return;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        f32x4 t = part[0][c][lane];
#pragma unroll
        for (int w = 1; w < RED_WAVES - 1; ++w) t += part[w][c][lane];      // fixed order
        sum[c] += t;
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int k = lane + 64 * c;
#pragma unroll
        for (int r = 0; r < 4; ++r) { dw[r][c] = 0.f; vv[r][c] = 0.f; }
        if (k < n_in) {
            f32x4 sv = sum[c];
            if (last) sv[0] += sv[1];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (o0 + r < out_dim) {
                    dw[r][c] = sv[r] * inv_k * mult;      // d/dW_l of the reference's Linear (the packed weight holds W_l/sqrt2 for the skip layer)
                    vv[r][c] = a.v[l][(size_t)(o0 + r) * n_in + k];
                    dot[r] = fmaf(dw[r][c], vv[r][c], dot[r]);
                    nrm[r] = fmaf(vv[r][c], vv[r][c], nrm[r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { dot[r] = wave_sum_f(dot[r]); nrm[r] = wave_sum_f(nrm[r]); }
    if (last && a.weight_norm && a.g[l][0] != 0.f) {
        // exact component of dW along W: dot(dW, v) = (||v||/g) sum_tiles ldot  (fixed order: deterministic)
        float ld = 0.f;
#pragma unroll 8
        for (int t = lane; t < a.n_tiles; t += 64) ld += a.ldot[t];
        ld = wave_sum_f(ld);
        dot[0] = ld * inv_k * sqrtf(nrm[0]) / a.g[l][0];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + r;
        if (o >= out_dim) break;
        float* dvrow = a.dv[l] + (size_t)o * n_in;
        if (a.weight_norm) {
            const float n = sqrtf(nrm[r]), gg = a.g[l][o];
            const float c1 = gg / n, c2 = dot[r] / nrm[r];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int k = lane + 64 * c;
                if (k < n_in) {
                    const float x = c1 * (dw[r][c] - c2 * vv[r][c]);
                    dvrow[k] = a.accumulate ? dvrow[k] + x : x;
                }
            }
            if (lane == 0) { const float x = dot[r] / n; a.dg[l][o] = a.accumulate ? a.dg[l][o] + x : x; }
        } else {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int k = lane + 64 * c;
                if (k < n_in) dvrow[k] = a.accumulate ? dvrow[k] + dw[r][c] : dw[r][c];
            }
        }
    }
    {   // bias rows: 16 lanes per row, lane i sums the K-slices q = i, i + 16, ... in order, then a fixed tree over the 16 lanes (deterministic;
        // one lane per row walking all slices was a chain of ~28 dependent global loads in every workgroup)
        const int r = lane >> 4, i = lane & 15;
        const int o = o0 + r;
        const int jb = (a.job_h[l] >= 0) ? a.job_h[l] : a.job_pe[l];
        const WgradJob& J = a.job[jb];
        float s = 0.f;
        if (o < out_dim)
            for (int q = i; q < J.n_slices; q += 16) s += a.partial[J.bias_off + (size_t)q * 256 + o] + (last ? a.partial[J.bias_off + (size_t)q * 256 + 1] : 0.f);
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (i == 0 && o < out_dim) {
            const float x = s * inv_k;
            a.db[l][o] = a.accumulate ? a.db[l][o] + x : x;
        }
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
            // a slice is bound by what it streams per K-step (round 3: with the cost counted in MFMAs the two PE jobs - 20 KiB per step for
            // a quarter of the MFMAs - got a third of the slices their bytes needed and ran 2.2x longer than everyone else: 334 us,
            // 3.6 TB/s; per-workgroup stamps then showed the one-row last layer paying for its 16 clamped Z fetches like a full level)
            // 1 KiB fetches per K-step and workgroup: A fragments + Z fetches; the clamped duplicates of a short Z level hit in L2 and
            // count about half (measured: the one-row last layer ran 294 us with 15 slices, 188 us with 28, next to 240 us)
            cost[n] = (double)J.a_ct + (double)J.z_rt + 0.5 * (double)(16 - (J.z_rt < 16 ? J.z_rt : 16));
            tot += cost[n];
            (part == 0 ? job_h : job_pe)[l] = n;
            ++n;
        }
    }
    // slices ~ cost, the whole budget handed out (largest remainders first)
    int sl[WGRAD_MAX_JOBS], given = 0;
    double rem[WGRAD_MAX_JOBS];
    for (int i = 0; i < n; ++i) {
        const double x = cost[i] / tot * wg_budget;
        sl[i] = (int)x < 1 ? 1 : (int)x;
        rem[i] = x - (double)sl[i];
        given += sl[i];
    }
    while (given < wg_budget) {
        int b = 0;
        for (int i = 1; i < n; ++i) if (rem[i] > rem[b]) b = i;
        ++sl[b]; rem[b] -= 1.0; ++given;
    }
    int wg = 0;
    size_t floats = 0;
    for (int i = 0; i < n; ++i) {
        const int s = sl[i];
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
                 const char* stash_z, float* partial, int n_tiles, int accumulate, hipStream_t st, float scale, int no_bias) {
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.stash_a = stash_a; a.stash_z = stash_z; a.partial = partial; a.scale = scale; a.no_bias = no_bias;
    a.n_tiles = n_tiles; a.n_jobs = n_jobs; a.a_tile_kb = V.a_tile_kb; a.z_tile_kb = V.z_tile_kb; a.accumulate = accumulate;
    for (int i = 0; i < n_jobs; ++i) a.job[i] = jobs[i];
    constexpr size_t lds = (size_t)(WGRAD_DEPTH + 1) * 32 * 1024;   // NST stages of (16 A + 16 Z) KiB
    static uint64_t attr_mask = 0;
    if (attr_needed(attr_mask)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<f16x8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<bf16x8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
            return EMAP_E_LAUNCH;
        }
    }
    if (L.is_f16) hipLaunchKernelGGL(wgrad_kernel<f16x8>, dim3(total_wg), dim3(512), lds, st, a);
    else hipLaunchKernelGGL(wgrad_kernel<bf16x8>, dim3(total_wg), dim3(512), lds, st, a);
    return check_launch("wgrad");
}

int launch_wgrad_reduce(const NetLayout& L, const WgradJob* jobs, int n_jobs, const int* job_h, const int* job_pe,
                        const float* partial, const uint32_t* absmax, const float* ldot, int n_tiles, const float* const* g, const float* const* v,
                        float* const* dg, float* const* dv, float* const* db, int weight_norm, int accumulate, float grad_scale,
                        hipStream_t st) {
    ReduceArgs a;
    memset(&a, 0, sizeof(a));
    a.partial = partial; a.absmax = absmax; a.ldot = ldot; a.n_tiles = n_tiles; a.grad_scale = grad_scale; a.accumulate = accumulate; a.weight_norm = weight_norm;
    a.n_lin = L.n_lin; a.H = L.H; a.d0 = L.d0; a.multires = L.multires; a.skip_l = L.skip_l;
    int rows = 0;   // counted in groups of 4 rows (one wave each)
    for (int l = 0; l < L.n_lin; ++l) {
        a.row_off[l] = rows; rows += (L.layer[l].out_dim + 3) / 4;
        a.out_dim[l] = L.layer[l].out_dim; a.in_prev[l] = L.layer[l].in_prev; a.has_pe[l] = L.layer[l].pe_ks ? 1 : 0;
        a.job_h[l] = job_h[l]; a.job_pe[l] = job_pe[l];
        a.g[l] = g ? g[l] : nullptr; a.v[l] = v[l]; a.dg[l] = dg ? dg[l] : nullptr; a.dv[l] = dv[l]; a.db[l] = db[l];
    }
    a.row_off[L.n_lin] = rows;
    for (int i = 0; i < n_jobs; ++i) a.job[i] = jobs[i];
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rows), dim3(RED_WAVES * 64), 0, st, a);
    return check_launch("wgrad_reduce");
}

}  // namespace emap
#ifdef EMAP_WGRAD_TIMING
extern "C" int emap_debug_wgrad_times(long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(emap::emap_wg_times), (size_t)n * sizeof(long long));
}
#endif
