// emap_common.h - shared host/device definitions for libemap_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/emap_hip.h"

// split-fp16 ("f16x3") arithmetic with ONE accumulator: instead of storing the lo parts x2^11 (f16 subnormals are flushed by the
// MFMA) and keeping the lo x hi products in a second accumulator, BOTH operands are pre-scaled - weights by F16X3_WS at pack
// time, activations by a per-kernel power of two - so that hi and lo parts are normal numbers and all three products land in
// one accumulator at scale WS*XS (undone in the epilogue).  Weights |w| <= 0.5, activations and deltas <= O(1), forward-mode
// tangents <= O(1e3) on the reference networks: 2^13 * 0.5, 2^9 * 64 and 2^4 * 4e3 stay below the f16 maximum.
// Measured (round 1): same accuracy (udf 6e-7, grad 1e-5 / 1e-6), all parity tests green, but 3 % SLOWER than the
// two-accumulator form (the extra scaling multiplies cost more than the freed registers buy), so it is off by default.

namespace emap {
constexpr float F16X3_WS = 8192.0f;

// ---- error plumbing (host) ------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// true the first time it is called for the current device with this mask (hipFuncSetAttribute is per function AND device)
inline bool attr_needed(uint64_t& mask) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return true;
    if (mask & (1ull << d)) return false;
    mask |= 1ull << d;
    return true;
}

// ---- packed-weight layout -------------------------------------------------------------------
// The MLP kernels keep activations in registers in MFMA-fragment order and never transpose them:
// the K dimension of layer l+1 is *permuted* so that the 8 k-values lane group g feeds to K-step s
// are exactly the 8 outputs of layer l it already holds (two 16-feature output tiles 2s, 2s+1, rows
// 4g..4g+3 of each).  The weights are packed once per optimizer step in the matching order.
//
// packed buffer = [bias  n_lin*H f32][rowscale n_lin*H f32][fragments ...]
// fragment      = one MFMA A operand (16 out-features x 32 k) = 64 lanes x 8 bf16 = 1 KiB,
//                 lane l=(g<<4)|i holds W[16*tile+i][kmap(s,g,0..7)]
// fragment order= layer, out-pair p, K-step s (PE block first), tile-in-pair t, part (hi, lo)
constexpr int FRAG_BYTES = 1024;
constexpr int PE_KS = 2;          // the PE block always occupies 2 K-steps (64 slots >= 3+6*10)

struct LayerDesc {
    int32_t pe_ks;      // 2 if the layer's input contains the PE block (layer 0, skip layer)
    int32_t h_ks;       // H/32 if the layer's input contains the previous hidden layer
    int32_t n_pairs;    // ceil(out_dim / 32): output tile pairs
    int32_t out_dim;    // real output features
    int32_t in_prev;    // real features taken from the previous layer (x part)
    int32_t frag_off;   // first fragment of the layer (in fragments, parts included)
    int32_t act;        // 1 = softplus(beta=100) after the layer
    int32_t pad;
};

struct NetLayout {
    int32_t H, n_lin, skip_l, multires, d0, nparts, udf_type, n_chunks;
    float scale;
    int32_t total_frags;
    int32_t bias_off_bytes, rowscale_off_bytes, frag_off_bytes;
    int32_t is_f16;
    int32_t mx_fwd;            // EMAP_PREC_F16X3M: the forward 32x32 section (r32) in the mixed MX layout too
    int32_t mx_bwd;            // 0 = EMAP_PREC_F16X3E: the transposed 32x32 section in plain f16 hi / lo fragments (f16 cross terms in the reverse sweep)
    // transposed section (reverse-mode d(udf)/dx, udf_mlp_rev_kernel): W_l^T fragments for the backward GEMMs
    //   t_off[l]   first fragment of layer l's hidden-row block  [row pair][K-step over out features][t][part], l >= 1
    //   tpe_off[l] first fragment of layer l's PE-row block      [pe pair 0..1][K-step][t][part], l in {0, skip_l}
    // and the last layer's single real row as fp32 (the backward sweep's seed).  has_rev = 0 when the topology is not
    // covered (skip layer == last layer): the forward-mode kernels are used then.
    int32_t has_rev, t_total_frags, t_frag_off_bytes, wlast_off_bytes;
    int32_t t_off[EMAP_MAX_LIN], tpe_off[EMAP_MAX_LIN];
    // the same two fragment sets in the K order of the 32x32x16 kernels (udf_mlp_rev32.inc): same fragment counts and the
    // same per-layer offsets (frag_off, t_off, tpe_off), a fragment = 32 rows x 16 k, index [row tile][K32-step][u][part]
    int32_t r32_frag_off_bytes, r32_t_frag_off_bytes;
    // "swm": the MX-fp6 operands of the TRAINING sweep's cross terms (udf_mlp_vjp.inc, round 5), split-fp16 at d_hidden = 256 (not in
    // EMAP_PREC_F16X3E): one unit of SWM_UNIT_BYTES per (GEMM, tile pair p, K128-step S) - W_hi and W_lo of the pair's two 16-row tiles
    // as e2m3 in the A layout of v_mfma_scale_f32_16x16x128_f8f6f4 (lane l: row l % 16, k-block l / 16 = K-step 4 S + l / 16) + their E8M0 bytes.
    //   swm_unit[l]    first unit of forward layer l's hidden-K GEMM (l >= 1; units [pair][S]), -1 if none
    //   swm_t_unit[l]  first unit of reverse step b = l (rows = input features of layer l, K = its output features), -1 if none
    int32_t sweep_mx, swm_off_bytes, swm_units;
    int32_t wgrad_lo;      // 1: the weight-gradient GEMMs multiply hi + lo parts of both operands (three passes; precision mode f16x3e, round 6)
    int32_t swm_unit[EMAP_MAX_LIN], swm_t_unit[EMAP_MAX_LIN];
    LayerDesc layer[EMAP_MAX_LIN];
};
// unit layout: fp6 registers q0..q3 of block (t, part6) at (2 t + part6) KiB + 16 lane; q4..q5 at 4 KiB + (2 t + part6) 512 + 8 lane;
// scale bytes at 6 KiB + 4 lane + (2 t + part6)      (part6: 0 = W_hi6, 1 = W_lo6, whose byte already undoes the x 2^11 of the lo parts)
constexpr int SWM_UNIT_BYTES = 6656;

// The transposed 32x32 section in the MIXED layout of the MX-fp6 reverse sweep (udf_mlp.hip:pack32_t_body, udf_mlp_rev32.inc):
// split-fp16 at d_hidden = 256 (a sweep wave owns two row tiles = one 32-value MX block per lane)
#ifndef EMAP_SWEEP_MX
#define EMAP_SWEEP_MX 1     // 1: the training sweep's cross terms as MX-fp6 MFMAs (udf_mlp_vjp.inc, round 5); 0: three f16 passes as in rounds 1-4 (A/B builds: all units)
#endif
#ifndef EMAP_REV_MX6
#define EMAP_REV_MX6 1      // 0: f16 cross terms in the backward GEMMs too (A/B builds: compile udf_mlp AND udf_mlp_f16x3 with the flag)
#endif
__host__ __device__ inline bool r32_t_mixed(const NetLayout& L) { return EMAP_REV_MX6 && L.mx_bwd && L.is_f16 && L.nparts == 2 && L.H == 256; }
// The same for the FORWARD sweep of that kernel (udf_mlp.hip:pack32_body writes the mixed layout too): precision mode EMAP_PREC_F16X3M.
__host__ __device__ inline bool r32_mixed(const NetLayout& L) { return L.mx_fwd && r32_t_mixed(L); }
// fixed MX scales of the positional-encoding block (|sin|, |cos| <= 1, raw coordinates up to 1.875 exactly): 2^-2 for the hi parts,
// 2^-3 for the lo parts (x 2^11 in f16, undone in the E8M0 byte)
#define EMAP_PE_HI6_E8M0 125u
#define EMAP_PE_LO6_E8M0 124u
// E8M0 scale of an MX block of e2m3 values with largest magnitude m, as the exponent field of an fp32 (bits 23..30):
// 2^(floor(log2(m * 8/7.5)) - 2), so that m / scale <= 7.5 (the e2m3 maximum)
__host__ __device__ inline uint32_t mx6_scale_bits(float m) {
    const float mm = fmaxf(m, 7.8886090522101181e-31f /* 2^-100 */) * 1.0666667f;
    uint32_t b;
    __builtin_memcpy(&b, &mm, 4);
    return (b & 0x7f800000u) - (2u << 23);
}

// returns 0 or EMAP_E_INVALID (error text set)
int build_layout(const EmapNetConfig* cfg, int prec, NetLayout* L);
inline size_t layout_bytes(const NetLayout& L) {
    const size_t base = (size_t)L.r32_t_frag_off_bytes + (size_t)L.t_total_frags * FRAG_BYTES;
    return L.sweep_mx ? (size_t)L.swm_off_bytes + (size_t)L.swm_units * SWM_UNIT_BYTES : base;
}

// launchers implemented in the .hip files
int launch_pack(const NetLayout& L, const float* const* g, const float* const* v, const float* const* b,
                void* packed, hipStream_t st);

struct PointSource {
    const float* x;            // explicit (P,3) points, or nullptr:
    const float* rays_o;       // (N,3)
    const float* rays_d;       // (N,3)
    const float* z;            // (N,n)
    int32_t n_per_ray;         // n
    int32_t mid;               // 1: evaluate at z + dists/2 with dists[last] = *sample_dist (render_core :435-449)
    const float* sample_dist;  // device scalar (mid=1)
    // coarse=1: z is not read; z[ray][i] = near + (far-near)*linspace(0,1,n)[i] + t_rand*2/n is evaluated on the fly (render() :705-720)
    const float* near;
    const float* far;
    const float* t_rand;       // may be null
    int32_t coarse;
    int32_t zero_n;            // zero_cnt[0 .. zero_n) = 0 at the start of the launch (value passes: the per-ray arrival counters of the
    int32_t* zero_cnt;         // render's fused compositing tail, CompositeFuse::ray_cnt); may be null
};

// ---- render_core's tail (udf_renderer_blending.py:435-455,463-677): arguments of composite_kernel / composite_ray (composite_dev.inc) ----
struct CompositeArgs {
    const float *rays_o, *rays_d, *z, *udf, *grad, *depth_scale, *sample_dist;
    int N, S;
    float inv_s, beta, gamma, car;
    int anneal;
    float flip_sat, near_surface, sparse_scale, background;
    int has_bg;
    const float *var_p, *beta_p, *gamma_p;  // optional raw device parameters (see EmapRenderParams)
    float beta_min;
    EmapCompositeOut out;
    float* partials;
};

// What the fused tail of udf_mlp_rev32_kernel needs beside the compositing arguments: one arrival counter per ray (points of the ray
// whose udf / grad_x have been written; zeroed by the first launch of the render, PointSource::zero_cnt).
struct CompositeFuse {
    CompositeArgs c;
    int32_t* ray_cnt;
    // rays composited so far in this launch (zeroed with ray_cnt), or null.  Not null: the workgroup whose rays bring it to c.N also runs the
    // deterministic cross-ray reduction (composite_reduce_body, the body of composite_reduce_kernel): the render ends with this launch.
    int32_t* done_cnt;
};
// upper bound on the rays ONE workgroup of the value + grad_x kernel can come to own (its list lives in the kernel's exchange buffer): every tile
// it runs (64 points, <= 512 workgroups) can complete at most 64 / S + 2 rays
constexpr int COMP_LIST_MAX = 8192;
inline long long comp_list_entries(long long P, int S) {
    const long long tiles = (P + 63) / 64, grid = tiles < 512 ? tiles : 512;
    return ((tiles + grid - 1) / (grid > 0 ? grid : 1)) * (64 / (S > 0 ? S : 1) + 2);
}
// one fused step of importance_sample (sampler.hip:sampler_step_kernel)
struct StepArgs {
    const float *rays_o, *rays_d;
    const float *z, *udf;            // (N,n)  [coarse: z is not read]
    const float *z_prev, *udf_prev;  // (N,m)  merge: the previous step's new samples and their udf
    float *z_merged, *udf_merged;    // (N,n+m) merge outputs; coarse: z_merged (N,n) receives the coarse z_vals
    const float *near, *far, *t_rand;   // coarse
    float* sample_dist;              // read; coarse: written
    float* z_new;                    // (N,m) this step's new samples
    float* z_final;                  // (N,n'+m) tail
    int N, n, m;
    float inv_s, beta, gamma;
    int32_t* err;
};
int launch_sampler_step(bool coarse, bool tail, const StepArgs& a, hipStream_t st);

// scratch: device buffer of at least rev_scratch_bytes(L) for the reverse-mode grad kernel (required when that kernel is
// selected; mlp_uses_rev() tells)
// importance_sample (udf_renderer_blending.py:802-841) as ONE launch (udf_mlp_kernel.inc, IS instantiations): the sampler steps and the MLP passes between
// them.  Returns EMAP_OK, an error, or IS_NOT_FUSED (> 0: shape not covered - the caller runs the launch chain; nothing was enqueued).
struct IsLaunch {
    const float *rays_o, *rays_d, *near, *far, *t_rand;   // t_rand may be null
    float* sample_dist;            // written (ray 0's wave)
    const float* udf_coarse;       // (N, Sc)
    float* z_final;                // (N, Sc + steps * m)
    int32_t N, Sc, m, steps;
};
constexpr int IS_NOT_FUSED = 1;
int launch_importance(const NetLayout& L, const void* packed, int prec, const IsLaunch& q, hipStream_t st, int32_t* err_flags);
int set_fused_sampling(int on);    // process-wide switch (tests, A/B): 0 chain, 1 fused where the launcher's size rule picks it, 2 fused at every size; returns the previous value
int fused_sampling_mode();
int set_value_tile_mode(int on);   // wide value launches (>= 512 tiles of 64 points, split modes, d_hidden 256): 1 = the 32x32 forward sweep (udf_mlp_rev32.inc, VAL), 0 = udf_mlp_fs2_kernel; returns the previous value
int value_tile_mode();
// fuse (value + grad_x launches that run the reverse-sweep kernel only - mlp_uses_rev()): the workgroup that writes the last point of a ray
// composites that ray (BASELINE config C2: "fused MLP + composite"); the caller then launches only the cross-ray reduction.
int launch_mlp(const NetLayout& L, const void* packed, int prec, const PointSource& src, int64_t P,
               float* udf, float* grad3, hipStream_t st, int32_t* err_flags = nullptr, void* scratch = nullptr,
               const CompositeFuse* fuse = nullptr);
int launch_null_direction(const float* g, int64_t n, int k, float* dir, hipStream_t st);

// ---- training backward (udf_mlp_vjp.inc, wgrad.hip) ---------------------------------------------------
// The sweep kernel leaves the operands of the weight-gradient GEMM per tile of VJP_PT points in MFMA-fragment order
// (K = columns: 64 per tile = 2 K-steps).  A "level" is a set of feature rows: rt row tiles x 2 K-steps x 1 KiB.
//   A level 0 = PE slots (4 row tiles), A level l+1 = output of hidden layer l; Z level l = adjoint of layer l's pre-activation
constexpr int VJP_PT = 32;
#ifndef EMAP_VJP_CHUNK_TILES
#define EMAP_VJP_CHUNK_TILES 16384
#endif
constexpr int VJP_CHUNK_TILES = EMAP_VJP_CHUNK_TILES;   // tiles per sweep launch of the preferred plan (bounds the stash: 16 384 x ~0.54 MiB = 8.8 GB, one chunk for 4096 rays x 128 samples;
                                                          // 12.85 vs 13.7 ms per training step there with chunks of 2048); a caller with a smaller workspace gets smaller chunks
#ifndef EMAP_VJP_NW256
#define EMAP_VJP_NW256 8
#endif
constexpr int VJP_NW_256 = EMAP_VJP_NW256;   // waves per workgroup of the sweep at d_hidden = 256: 8 (one workgroup per CU; measured 876-880 us) or 4 (two tile pairs per wave, two workgroups per CU: 939-958 us, 22 % more cycles)
constexpr int WGRAD_MAX_JOBS = 2 * EMAP_MAX_LIN;
struct VjpLayout {
    int32_t a_rt[EMAP_MAX_LIN + 1], a_off[EMAP_MAX_LIN + 1];   // row tiles / KiB offset inside a tile's A block
    int32_t z_rt[EMAP_MAX_LIN], z_off[EMAP_MAX_LIN];           // same for the Z block
    int32_t s_off[EMAP_MAX_LIN];                               // KiB offset of layer l's sigma' inside a workgroup's slab
    int32_t a_tile_kb, z_tile_kb, s_slab_kb;
    // precise weight gradients (NetLayout::wgrad_lo; precision mode f16x3e): the sweep also leaves the LO parts of both operand sets, in the same
    // layout, lo_a_delta / lo_z_delta bytes behind the hi parts' stashes (0 = not stashed)
    long long lo_a_delta, lo_z_delta;
};
void build_vjp_layout(const NetLayout& L, VjpLayout* V);
// one weight-gradient GEMM job (wgrad.hip)
struct WgradJob {
    int32_t layer, part;
    int32_t z_off, z_rt;        // KiB offset / row tiles of the Z level
    int32_t a_off, a_ct;        // KiB offset / column tiles (= row tiles of the A level that are used)
    int32_t first_wg, n_slices; // workgroups [first_wg, first_wg + n_slices)
    int32_t part_off;           // offset of the job's partial block in floats: [slice][16 row tiles][a_ct][256]
    int32_t bias_off;           // offset of the job's bias partials in floats: [slice][256] (part 0 jobs and layer 0 only, else -1)
};


bool mlp_uses_rev(const NetLayout& L, int prec, int64_t P);
int set_grad_mode(int mode);   // -1 by launch size, 0 forward-mode tangents, 1 reverse sweep; returns the previous setting (udf_mlp.hip)
extern int g_prof_clk_device;
long long* prof_clk_here();   // g_prof_clk if the current device is the one it was allocated on, else null
extern long long* g_prof_clk;   // device buffer of 8 x int64 while emap_profile_enable(1) is in effect, else null (clock_stamp in udf_mlp_kernel.inc)
constexpr int REV_MAX_WG = 768;      // persistent workgroups of the reverse-mode kernel (3 per CU)
// sigma' stash of the reverse-mode kernel per row tile and column tile of 32 points: 16 values per lane x 64 lanes, 3 bytes reserved per value
// (unorm16 planes; precision mode f16x3e adds one plane of low bytes - udf_mlp_rev32.inc, SG24)
constexpr int REV_SG_BYTES_PER_NC = 3072;
inline size_t rev_scratch_bytes(const NetLayout& L) {   // sigmoid stash: [workgroup][layer][pair][6][64 lanes x 16 B]
    // + 8 KiB per workgroup: the lo parts of the tile's PE block (MX6F: their LDS slots hold the fp6 forms; read back by the PE rows)
    return L.has_rev ? (size_t)(REV_MAX_WG * 2 / 3) * ((size_t)(L.n_lin - 1) * (size_t)(L.H / 32) * (2 * REV_SG_BYTES_PER_NC) + 8192) : 0;   // 2 resident workgroups per CU
}

}  // namespace emap
