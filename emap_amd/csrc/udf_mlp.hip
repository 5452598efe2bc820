// udf_mlp.hip - weight packing (weight_norm fold + MFMA-fragment permutation) and the precision dispatcher of the
// fused UDF-MLP kernels.  The kernels themselves live in udf_mlp_kernel.inc, instantiated per precision mode in
// udf_mlp_{bf16,bf16x3,f16,f16x3}.hip.
#include "emap_common.h"
#include <atomic>
#include <stdlib.h>
#include <string.h>

namespace emap {
long long* g_prof_clk = nullptr;
int g_prof_clk_device = -1;
// the clock buffer lives on the device that was current when profiling was enabled: launches on another device of the process get
// no stamps (a foreign device pointer would fault without peer access)
long long* prof_clk_here() {
    if (!g_prof_clk) return nullptr;
    int d = -1;
    return (hipGetDevice(&d) == hipSuccess && d == g_prof_clk_device) ? g_prof_clk : nullptr;
}


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__host__ __device__ constexpr bool prec_is_f16(int mode) { return mode == EMAP_PREC_F16 || mode == EMAP_PREC_F16X3 || mode == EMAP_PREC_F16X3M || mode == EMAP_PREC_F16X3E; }
__host__ __device__ constexpr int prec_nparts(int mode) { return (mode == EMAP_PREC_BF16 || mode == EMAP_PREC_F16) ? 1 : 2; }

// ---------------------------------------------------------------------------------------------
// weight packing:  W = g * v / ||v||  ->  permuted bf16 (hi[, lo]) MFMA fragments
// ---------------------------------------------------------------------------------------------
struct PackArgs {
    const float* g[EMAP_MAX_LIN];
    const float* v[EMAP_MAX_LIN];
    const float* b[EMAP_MAX_LIN];
    int32_t in_dim[EMAP_MAX_LIN];  // row length of v[l]
    NetLayout L;
    char* packed;
};

// one wave per (layer, row): rowscale = g / ||v_row||, bias copy (zero for padded rows)
__global__ __launch_bounds__(256) void rowscale_kernel(const PackArgs a) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int H = a.L.H;
    if (row >= a.L.n_lin * H) return;
    const int l = row / H, o = row - l * H;
    float* bias = reinterpret_cast<float*>(a.packed + a.L.bias_off_bytes);
    float* rs = reinterpret_cast<float*>(a.packed + a.L.rowscale_off_bytes);
    if (o >= a.L.layer[l].out_dim) {
        if (lane == 0) { bias[row] = 0.f; rs[row] = 0.f; }
        return;
    }
    const int n = a.in_dim[l];
    const float* v = a.v[l] + (size_t)o * n;
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s = fmaf(v[i], v[i], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
        bias[row] = a.b[l][o];
        rs[row] = a.g[l][o] / sqrtf(s);
    }
}

// W_l[o][col] * rowscale * mult, or 0 outside the matrix - the two loads are UNCONDITIONAL on clamped indices (a branch per element kept
// the compiler from batching the 8 / 32 independent gathers of a thread: each waited for its own round trip)
__device__ __forceinline__ float packed_weight(const PackArgs& a, const float* rs, int l, int H, int o, int col, int out_dim, int n_in, float mult) {
    const bool ok = (o < out_dim) & (col >= 0) & (col < n_in);
    const int oc = min(max(o, 0), out_dim - 1), cc = min(max(col, 0), n_in - 1);
    const float w = rs[l * H + oc] * a.v[l][(size_t)oc * n_in + cc] * mult;
    return ok ? w : 0.f;
}

__device__ __forceinline__ void pack_body(const PackArgs& a, unsigned block) {
    const long long gid = (long long)block * 256 + threadIdx.x;  // one thread per (fragment, lane)
    const long long F = gid >> 6;
    const int lane = (int)(gid & 63);
    if (F >= a.L.total_frags) return;
    int l = 0;
    while (l + 1 < a.L.n_lin && F >= a.L.layer[l + 1].frag_off) ++l;
    const LayerDesc Ld = a.L.layer[l];
    const int H = a.L.H, NP = a.L.nparts, d0 = a.L.d0, M = a.L.multires;
    int idx = (int)(F - Ld.frag_off);
    const int part = idx % NP; idx /= NP;
    const int t = idx & 1; idx >>= 1;
    const int n_ks = Ld.pe_ks + Ld.h_ks;
    const int s = idx % n_ks;
    const int p = idx / n_ks;
    const int i = lane & 15, g = lane >> 4;
    const int o = 32 * p + 16 * t + i;
    const float* rs = reinterpret_cast<const float*>(a.packed + a.L.rowscale_off_bytes);
    const float mult = (l == a.L.skip_l) ? 0.70710678118654752440f : 1.0f;  // cat([x, PE]) / sqrt(2), udf_model.py:100
    const int n_in = a.in_dim[l];
    const bool f16 = a.L.is_f16 != 0;
    bf16x8 outv;
    f16x8 outh;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int col = -1;
        if (s < Ld.pe_ks) {
            // PE block slot (sp, g, e): pair q = 4*sp + e/2 -> angle a = 8*g + q; kind = e&1 (sin|cos)
            const int q = 4 * s + (e >> 1), ang = 8 * g + q, kind = e & 1;
            int pcol = -1;
            if (ang < 3 * M) {
                const int k = ang / 3, c = ang - 3 * k;
                pcol = 3 + 6 * k + (kind ? 3 + c : c);
            } else if (ang == 3 * M) {
                pcol = kind ? 1 : 0;
            } else if (ang == 3 * M + 1) {
                pcol = kind ? -1 : 2;
            }
            if (pcol >= 0 && pcol < d0) col = (l == 0) ? pcol : Ld.in_prev + pcol;
        } else {
            const int sh = s - Ld.pe_ks;
            const int f = 16 * (2 * sh + (e >> 2)) + 4 * g + (e & 3);
            if (f < Ld.in_prev) col = f;
        }
        const float w = packed_weight(a, rs, l, H, o, col, Ld.out_dim, n_in, mult);
        const __bf16 hi = (__bf16)w;
        outv[e] = (part == 0) ? hi : (__bf16)(w - (float)hi);
        const _Float16 hh = (_Float16)w;
        outh[e] = (part == 0) ? hh : (_Float16)((w - (float)hh) * 2048.0f);  // lo parts scaled by 2^11 (split-fp16)
    }
    char* dst = a.packed + a.L.frag_off_bytes + F * FRAG_BYTES + lane * 16;
    if (f16) *reinterpret_cast<f16x8*>(dst) = outh;
    else *reinterpret_cast<bf16x8*>(dst) = outv;
}

// Transposed fragments for the reverse sweep: A operand of  delta_in = W_l^T * delta_z_l.
//   hidden rows: row f = 32*p + 16*t + i is input feature f of layer l (natural order, = the C-fragment row of the
//                forward output a_{l-1}, so sigma' stashed by the forward sweep lines up lane for lane);
//   PE rows:     row (tile tau = 2*pp + t, i = 4*gc + r) is PE slot (sp = pp, gc, e = 4*t + r) - the slot lane group gc
//                itself produced in the forward PE block;
//   k element (s, g, e) is output feature 16*(2s + e/4) + 4g + e%4 of layer l (the usual permutation).
__device__ __forceinline__ void pack_t_body(const PackArgs& a, unsigned block) {
    const long long gid = (long long)block * 256 + threadIdx.x;
    const long long F = gid >> 6;
    const int lane = (int)(gid & 63);
    if (F >= a.L.t_total_frags) return;
    const int H = a.L.H, NP = a.L.nparts, d0 = a.L.d0, M = a.L.multires, NKS = H / 32;
    // locate the block
    int l = -1; bool pe = false; int base = 0;
    for (int q = 0; q < a.L.n_lin; ++q) {
        if (q >= 1 && q < a.L.n_lin - 1) {
            const int n = ((a.L.layer[q].in_prev + 31) / 32) * NKS * 2 * NP;
            if (F >= a.L.t_off[q] && F < a.L.t_off[q] + n) { l = q; pe = false; base = a.L.t_off[q]; }
        }
        if (q == 0 || q == a.L.skip_l) {
            const int n = 2 * NKS * 2 * NP;
            if (F >= a.L.tpe_off[q] && F < a.L.tpe_off[q] + n) { l = q; pe = true; base = a.L.tpe_off[q]; }
        }
    }
    if (l < 0) return;
    const LayerDesc Ld = a.L.layer[l];
    int idx = (int)(F - base);
    const int part = idx % NP; idx /= NP;
    const int t = idx & 1; idx >>= 1;
    const int s = idx % NKS;
    const int p = idx / NKS;
    const int i = lane & 15, g = lane >> 4;
    const float* rs = reinterpret_cast<const float*>(a.packed + a.L.rowscale_off_bytes);
    const float mult = (l == a.L.skip_l) ? 0.70710678118654752440f : 1.0f;
    const int n_in = a.in_dim[l];
    int col = -1;   // column of W_l this row stands for
    if (!pe) {
        const int f = 32 * p + 16 * t + i;
        if (f < Ld.in_prev) col = f;
    } else {
        const int gc = i >> 2, r = i & 3, e_pe = 4 * t + r;
        const int q = 4 * p + (e_pe >> 1), ang = 8 * gc + q, kind = e_pe & 1;
        int pcol = -1;
        if (ang < 3 * M) {
            const int k = ang / 3, c = ang - 3 * k;
            pcol = 3 + 6 * k + (kind ? 3 + c : c);
        } else if (ang == 3 * M) {
            pcol = kind ? 1 : 0;
        } else if (ang == 3 * M + 1) {
            pcol = kind ? -1 : 2;
        }
        if (pcol >= 0 && pcol < d0) col = (l == 0) ? pcol : Ld.in_prev + pcol;
    }
    const bool f16 = a.L.is_f16 != 0;
    bf16x8 outv;
    f16x8 outh;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int o = 16 * (2 * s + (e >> 2)) + 4 * g + (e & 3);
        const float w = packed_weight(a, rs, l, H, o, col, Ld.out_dim, n_in, mult);
        const __bf16 hi = (__bf16)w;
        outv[e] = (part == 0) ? hi : (__bf16)(w - (float)hi);
        const _Float16 hh = (_Float16)w;
        outh[e] = (part == 0) ? hh : (_Float16)((w - (float)hh) * 2048.0f);
    }
    char* dst = a.packed + a.L.t_frag_off_bytes + F * FRAG_BYTES + lane * 16;
    if (f16) *reinterpret_cast<f16x8*>(dst) = outh;
    else *reinterpret_cast<bf16x8*>(dst) = outv;
}


// ---- fragments in the K order of the 32x32x16 kernels (udf_mlp_rev32.inc) -------------------------------------------------
// A operand of v_mfma_f32_32x32x16: lane (hh = lane>>5, i = lane&31) holds row i, k = 8*hh + e.  One fragment = 32 output
// features x 16 k; fragment order = layer, row tile p, K32-step S (PE block first), u (K16 half), part (hi, lo).
//   hidden k-slot (S, u, hh, e) = feature 32S + (e&3) + 8(2u + (e>>2)) + 4hh  (register r = 8u + e of the lane's C fragment)
//   PE k-slot (S, u, hh, e): angle index 16hh + 8S + 4u + (e>>1), kind (sin|cos) = e&1
__device__ __forceinline__ int pe_col_of(int ang, int kind, int M, int d0) {
    int pcol = -1;
    if (ang < 3 * M) {
        const int k = ang / 3, c = ang - 3 * k;
        pcol = 3 + 6 * k + (kind ? 3 + c : c);
    } else if (ang == 3 * M) {
        pcol = kind ? 1 : 0;
    } else if (ang == 3 * M + 1) {
        pcol = kind ? -1 : 2;
    }
    return (pcol >= 0 && pcol < d0) ? pcol : -1;
}
__device__ __forceinline__ void store_frag(const PackArgs& a, char* dst, const float (&w)[8], int part) {
    bf16x8 outv;
    f16x8 outh;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hi = (__bf16)w[e];
        outv[e] = (part == 0) ? hi : (__bf16)(w[e] - (float)hi);
        const _Float16 hh = (_Float16)w[e];
        outh[e] = (part == 0) ? hh : (_Float16)((w[e] - (float)hh) * 2048.0f);  // lo parts scaled by 2^11 (split-fp16)
    }
    if (a.L.is_f16) *reinterpret_cast<f16x8*>(dst) = outh;
    else *reinterpret_cast<bf16x8*>(dst) = outv;
}

typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef uint32_t u32x6 __attribute__((ext_vector_type(6)));

// The 8 KiB block of one (row tile, K64-step Sg) in the MIXED layout of the MX-fp6 sweeps (layout: see pack32_t_body), written by thread f = 0 of
// each lane - the fragment-granular grid of pack_all_kernel gives a block eight waves, seven of them exit: one thread evaluates the lane's 32
// weights ONCE and writes everything derived from them (the first version spread the block over the eight threads, each recomputing the block
// maxima: 416 weight evaluations per lane and block instead of 32, and the re-pack of a training step took 154 us instead of 17).
// weight(S, u, e) = the element of K32-step S, K16-step u, k-slot e of this lane's row.
template <class WF>
__device__ __forceinline__ void store_mixed_block(const PackArgs& a, char* blk, int lane, int f, int Sg, int NSG, WF&& weight) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    if (f != 0) return;
    f16x32 vh, vl;
    float mh = 0.f, ml = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) {       // element order of the lane's MX block: e = 16 t + 8 u + e'  <->  (S = 2 Sg + t, u, e')
        const float w = weight(2 * Sg + (e >> 4), (e >> 3) & 1, e & 7);
        const _Float16 h16 = (_Float16)w;
        const _Float16 l16 = (_Float16)((w - (float)h16) * 2048.0f);
        vh[e] = h16; vl[e] = l16;
        mh = fmaxf(mh, fabsf((float)h16));
        ml = fmaxf(ml, fabsf((float)l16));
    }
    // @0..3 KiB: the four hi16 fragments (t, u)
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f16x8v o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = vh[8 * q + e];
        *reinterpret_cast<f16x8v*>(blk + q * FRAG_BYTES + lane * 16) = o;
    }
    // fp6 forms and their E8M0 bytes (lo parts are stored x 2^11: undone in the byte)
    const uint32_t sbh = mx6_scale_bits(mh), sbl = mx6_scale_bits(ml);
    const u32x6 qh = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vh, __builtin_bit_cast(float, sbh));
    const u32x6 ql = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vl, __builtin_bit_cast(float, sbl));
    *reinterpret_cast<u32x4*>(blk + 4 * FRAG_BYTES + lane * 16) = u32x4{qh[0], qh[1], qh[2], qh[3]};
    *reinterpret_cast<u32x4*>(blk + 5 * FRAG_BYTES + lane * 16) = u32x4{ql[0], ql[1], ql[2], ql[3]};
    *reinterpret_cast<u32x2*>(blk + 6 * FRAG_BYTES + lane * 8) = u32x2{qh[4], qh[5]};
    *reinterpret_cast<u32x2*>(blk + 6 * FRAG_BYTES + 512 + lane * 8) = u32x2{ql[4], ql[5]};
    // @7 KiB of block Sg' (8 B / lane): bytes 2 j + (hi6 | lo6) = the scales of step Sg' + j, j = 0..3.  This thread owns the bytes of ITS step in
    // the slots of blocks Sg, Sg - 1, Sg - 2, Sg - 3 (byte stores) and zeroes the bytes of its own slot that belong to no step
    const uint8_t bh = (uint8_t)(sbh >> 23), bl = (uint8_t)((sbl >> 23) - 11u);
    for (int j = 0; j < 4; ++j) {
        if (Sg - j >= 0) {
            uint8_t* slot = reinterpret_cast<uint8_t*>(blk - (ptrdiff_t)j * 8 * FRAG_BYTES + 7 * FRAG_BYTES + lane * 8);
            slot[2 * j] = bh; slot[2 * j + 1] = bl;
        }
        if (Sg + j >= NSG) {
            uint8_t* own = reinterpret_cast<uint8_t*>(blk + 7 * FRAG_BYTES + lane * 8);
            own[2 * j] = 0; own[2 * j + 1] = 0;
        }
    }
}

__device__ __forceinline__ void pack32_body(const PackArgs& a, unsigned block) {
    const long long gid = (long long)block * 256 + threadIdx.x;  // one thread per (fragment, lane)
    const long long F = gid >> 6;
    const int lane = (int)(gid & 63);
    if (F >= a.L.total_frags) return;
    int l = 0;
    while (l + 1 < a.L.n_lin && F >= a.L.layer[l + 1].frag_off) ++l;
    const LayerDesc Ld = a.L.layer[l];
    const int H = a.L.H, NP = a.L.nparts, d0 = a.L.d0, M = a.L.multires;
    int idx = (int)(F - Ld.frag_off);
    const int n_ks = Ld.pe_ks + Ld.h_ks;
    const int i = lane & 31, hh = lane >> 5;
    const float* rs = reinterpret_cast<const float*>(a.packed + a.L.rowscale_off_bytes);
    const float mult = (l == a.L.skip_l) ? 0.70710678118654752440f : 1.0f;  // cat([x, PE]) / sqrt(2), udf_model.py:100
    const int n_in = a.in_dim[l];
    // element (K32-step S, K16-step u, k-slot e) of row o
    auto weight = [&](int o, int S, int u, int e) -> float {
        int col = -1;
        if (S < Ld.pe_ks) {
            const int pcol = pe_col_of(16 * hh + 8 * S + 4 * u + (e >> 1), e & 1, M, d0);
            if (pcol >= 0) col = (l == 0) ? pcol : Ld.in_prev + pcol;
        } else {
            const int f = 32 * (S - Ld.pe_ks) + (e & 3) + 8 * (2 * u + (e >> 2)) + 4 * hh;
            if (f < Ld.in_prev) col = f;
        }
        return packed_weight(a, rs, l, H, o, col, Ld.out_dim, n_in, mult);
    };
    if (r32_mixed(a.L)) {      // MX-fp6 forward sweep: one 8 KiB block per (row tile, K64-step) - every layer has an even number of K32-steps
        const int f = idx & 7; idx >>= 3;
        const int NSG = n_ks / 2;
        const int Sg = idx % NSG;
        const int o = 32 * (idx / NSG) + i;
        store_mixed_block(a, a.packed + a.L.r32_frag_off_bytes + (F - f) * FRAG_BYTES, lane, f, Sg, NSG,
                          [&](int S, int u, int e) -> float { return weight(o, S, u, e); });
        return;
    }
    const int part = idx % NP; idx /= NP;
    const int u = idx & 1; idx >>= 1;
    const int S = idx % n_ks;
    const int o = 32 * (idx / n_ks) + i;
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = weight(o, S, u, e);
    store_frag(a, a.packed + a.L.r32_frag_off_bytes + F * FRAG_BYTES + lane * 16, w, part);
}

// Transposed fragments of the 32x32x16 reverse sweep: A operand of  delta_in = W_l^T * delta_z_l.
//   hidden rows: row i of row tile p is input feature 32p + i of layer l (natural order = the A-operand row, so the C
//                fragment of the backward GEMM lines up lane for lane with the sigma' the forward sweep stashed);
//   PE rows:     row i of tile tau: hc = (i>>2)&1, r = (i&3) + 4(i>>3)  ->  PE slot (S = tau, u = r>>3, hc, e = r&7), the slot
//                lane half hc holds in register r of its C fragment and in its own PE fragment;
//   k element (S, u, hh, e) is output feature 32S + (e&3) + 8(2u + (e>>2)) + 4hh of layer l.
__device__ __forceinline__ void pack32_t_body(const PackArgs& a, unsigned block) {
    const long long gid = (long long)block * 256 + threadIdx.x;
    const long long F = gid >> 6;
    const int lane = (int)(gid & 63);
    if (F >= a.L.t_total_frags) return;
    const int H = a.L.H, NP = a.L.nparts, d0 = a.L.d0, M = a.L.multires, NKS = H / 32;
    int l = -1; bool pe = false; int base = 0;
    for (int q = 0; q < a.L.n_lin; ++q) {
        if (q >= 1 && q < a.L.n_lin - 1) {
            const int n = ((a.L.layer[q].in_prev + 31) / 32) * NKS * 2 * NP;
            if (F >= a.L.t_off[q] && F < a.L.t_off[q] + n) { l = q; pe = false; base = a.L.t_off[q]; }
        }
        if (q == 0 || q == a.L.skip_l) {
            const int n = 2 * NKS * 2 * NP;
            if (F >= a.L.tpe_off[q] && F < a.L.tpe_off[q] + n) { l = q; pe = true; base = a.L.tpe_off[q]; }
        }
    }
    if (l < 0) return;
    const LayerDesc Ld = a.L.layer[l];
    int idx = (int)(F - base);
    const int i = lane & 31, hh = lane >> 5;
    const float* rs = reinterpret_cast<const float*>(a.packed + a.L.rowscale_off_bytes);
    const float mult = (l == a.L.skip_l) ? 0.70710678118654752440f : 1.0f;
    const int n_in = a.in_dim[l];
    // row i of row tile p -> column `col` of W_l
    auto col_of = [&](int p) -> int {
        if (!pe) {
            const int f = 32 * p + i;
            return (f < Ld.in_prev) ? f : -1;
        }
        const int hc = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3);
        const int pcol = pe_col_of(16 * hc + 8 * p + 4 * (r >> 3) + ((r & 7) >> 1), r & 1, M, d0);
        return (pcol >= 0) ? ((l == 0) ? pcol : Ld.in_prev + pcol) : -1;
    };
    auto weight = [&](int col, int S, int u, int e) -> float {
        const int o = 32 * S + (e & 3) + 8 * (2 * u + (e >> 2)) + 4 * hh;
        return packed_weight(a, rs, l, H, o, col, Ld.out_dim, n_in, mult);
    };
    char* dst = a.packed + a.L.r32_t_frag_off_bytes + F * FRAG_BYTES + lane * 16;
    if (!r32_t_mixed(a.L)) {
        const int part = idx % NP; idx /= NP;
        const int u = idx & 1; idx >>= 1;
        const int S = idx % NKS;
        const int p = idx / NKS;
        const int col = col_of(p);
        float w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = weight(col, S, u, e);
        store_frag(a, dst, w, part);
        return;
    }
    // ---- mixed layout (split-fp16, d_hidden = 256): one 8 KiB block per (row tile p, K64-step Sg):
    //   @0..3 KiB  hi16 fragments (S = 2Sg, u = 0), (2Sg, 1), (2Sg + 1, 0), (2Sg + 1, 1)
    //   @4 KiB     W_hi6 registers q0..q3 (16 B / lane)        @5 KiB    W_lo6 q0..q3
    //   @6 KiB     W_hi6 q4..q5 (8 B / lane)                   @6.5 KiB  W_lo6 q4..q5
    //   @7 KiB     8 B / lane: E8M0 scales of this lane's blocks of steps Sg .. Sg+3, byte 2 j + (hi6 | lo6) for step Sg + j
    // The cross terms W_hi x_lo + W_lo x_hi of the reverse sweep run as MX-scaled fp6 (e2m3) MFMAs of K = 64
    // (v_mfma_scale_f32_32x32x64_f8f6f4, 4x the f16 rate; docs/DESIGN_LOG_r1-r4.md par. 6c): lane (hh, i) of an fp6 block holds the 32 k-slots
    // e = 16 pi + 8 u + e' <-> (S = 2Sg + pi, u, hh, e') of row i - the order in which a sweep lane holds its two row tiles' outputs -
    // as six registers of e2m3.  hi6 quantises the f16 hi parts, lo6 the lo parts (x 2^11 in f16, undone in the scale byte); both
    // through v_cvt_scalef32_pk32_fp6_f16, the instruction the sweep itself uses for its B operands (probed: natural element order,
    // x / scale, RNE, saturating; profiles/r04_probe_fp6.txt).
    const int f = idx & 7; idx >>= 3;
    const int NSG = NKS / 2;
    const int Sg = idx % NSG;
    const int col = col_of(idx / NSG);
    store_mixed_block(a, a.packed + a.L.r32_t_frag_off_bytes + (F - f) * FRAG_BYTES, lane, f, Sg, NSG,
                      [&](int S, int u, int e) -> float { return weight(col, S, u, e); });
}

// MX operands of the training sweep (emap_common.h "swm"): one thread per (unit, row tile t, lane); lane (kb, i) holds row 16 t + i of the
// pair and the 32 k-values of K-step 4 S + kb in the element order idx = 8 g' + e  <->  feature 16 (2 s + (e >> 2)) + 4 g' + (e & 3): exactly what
// the four lanes (g', i) of the f16 fragment (s, t) hold, and on the B side what the four lanes (g', column) of an output fragment hold.
__device__ __forceinline__ void pack_swm_body(const PackArgs& a, unsigned block) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const long long gid = (long long)block * 256 + threadIdx.x;
    const int unit = (int)(gid >> 7), t = (int)((gid >> 6) & 1), lane = (int)(gid & 63);
    if (unit >= a.L.swm_units) return;
    const int H = a.L.H, n128 = H / 128;
    int l = -1; bool tr = false;
    for (int q = 1; q < a.L.n_lin; ++q) {
        if (a.L.swm_unit[q] >= 0 && unit >= a.L.swm_unit[q] && unit < a.L.swm_unit[q] + a.L.layer[q].n_pairs * n128) { l = q; tr = false; }
        if (a.L.swm_t_unit[q] >= 0 && unit >= a.L.swm_t_unit[q] && unit < a.L.swm_t_unit[q] + ((a.L.layer[q].in_prev + 31) / 32) * n128) { l = q; tr = true; }
    }
    if (l < 0) return;
    const LayerDesc Ld = a.L.layer[l];
    const int u = unit - (tr ? a.L.swm_t_unit[l] : a.L.swm_unit[l]);
    const int p = u / n128, S = u % n128;
    const int i = lane & 15, kb = lane >> 4, s = 4 * S + kb;
    const float* rs = reinterpret_cast<const float*>(a.packed + a.L.rowscale_off_bytes);
    const float mult = (l == a.L.skip_l) ? 0.70710678118654752440f : 1.0f;
    const int n_in = a.in_dim[l];
    const int row = 32 * p + 16 * t + i;          // forward: output feature; transposed: input feature (a column of W_l)
    f16x32 vh, vl;
    float mh = 0.f, ml = 0.f;
#pragma unroll
    for (int idx = 0; idx < 32; ++idx) {
        const int gq = idx >> 3, e = idx & 7;
        const int f = 16 * (2 * s + (e >> 2)) + 4 * gq + (e & 3);
        // forward: W[row][f], f < in_prev; transposed: W[f][row], row < in_prev
        const int wo = tr ? f : row, wc = tr ? row : f;
        const float w = packed_weight(a, rs, l, H, wo, (wc < Ld.in_prev) ? wc : -1, Ld.out_dim, n_in, mult);
        const _Float16 h16 = (_Float16)w;
        const _Float16 l16 = (_Float16)((w - (float)h16) * 2048.0f);
        vh[idx] = h16; vl[idx] = l16;
        mh = fmaxf(mh, fabsf((float)h16));
        ml = fmaxf(ml, fabsf((float)l16));
    }
    const uint32_t sbh = mx6_scale_bits(mh), sbl = mx6_scale_bits(ml);
    const u32x6 qh = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vh, __builtin_bit_cast(float, sbh));
    const u32x6 ql = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vl, __builtin_bit_cast(float, sbl));
    char* ub = a.packed + a.L.swm_off_bytes + (size_t)unit * SWM_UNIT_BYTES;
    *reinterpret_cast<u32x4*>(ub + (2 * t + 0) * 1024 + lane * 16) = u32x4{qh[0], qh[1], qh[2], qh[3]};
    *reinterpret_cast<u32x4*>(ub + (2 * t + 1) * 1024 + lane * 16) = u32x4{ql[0], ql[1], ql[2], ql[3]};
    *reinterpret_cast<u32x2*>(ub + 4096 + (2 * t + 0) * 512 + lane * 8) = u32x2{qh[4], qh[5]};
    *reinterpret_cast<u32x2*>(ub + 4096 + (2 * t + 1) * 512 + lane * 8) = u32x2{ql[4], ql[5]};
    uint8_t* sc = reinterpret_cast<uint8_t*>(ub + 6144 + lane * 4);
    sc[2 * t + 0] = (uint8_t)(sbh >> 23);
    sc[2 * t + 1] = (uint8_t)((sbl >> 23) - 11u);
}

// fp32 copy of the last layer's real row (times its weight-norm scale): the seed of the reverse sweep
__device__ __forceinline__ void pack_wlast_body(const PackArgs& a, unsigned block) {
    const int f = block * 256 + threadIdx.x;
    const int H = a.L.H, l = a.L.n_lin - 1;
    if (f >= H) return;
    const float* rs = reinterpret_cast<const float*>(a.packed + a.L.rowscale_off_bytes);
    float* wl = reinterpret_cast<float*>(a.packed + a.L.wlast_off_bytes);
    const float mult = (l == a.L.skip_l) ? 0.70710678118654752440f : 1.0f;   // skip layer == last layer (d4 networks)
    wl[f] = (f < a.L.layer[l].in_prev) ? rs[l * H] * a.v[l][f] * mult : 0.f;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int build_layout(const EmapNetConfig* cfg, int prec, NetLayout* L) {
    if (!cfg || !L) { set_error("null config"); return EMAP_E_INVALID; }
    if (cfg->d_hidden != 256 && cfg->d_hidden != 128) { set_error("d_hidden must be 128 or 256 (got %d)", cfg->d_hidden); return EMAP_E_INVALID; }
    if (cfg->n_lin < 2 || cfg->n_lin > EMAP_MAX_LIN) { set_error("n_lin out of range (%d)", cfg->n_lin); return EMAP_E_INVALID; }
    if (cfg->multires < 0 || cfg->multires > 10) { set_error("multires must be in 0..10 (got %d)", cfg->multires); return EMAP_E_INVALID; }   // 0: raw coordinates only (udf_model.py:26-29)
    if (cfg->d_out != 1) { set_error("d_out must be 1 (got %d): feature outputs are not on the hot path", cfg->d_out); return EMAP_E_INVALID; }
    if (prec < EMAP_PREC_BF16 || prec > EMAP_PREC_F16X3E) { set_error("unknown precision mode %d", prec); return EMAP_E_INVALID; }
    if (prec == EMAP_PREC_F16X3M && (cfg->d_hidden != 256 || !EMAP_REV_MX6)) {
        set_error("precision f16x3m (MX fp6 cross terms in the forward sweep) needs d_hidden = 256 (got %d)", cfg->d_hidden);
        return EMAP_E_INVALID;
    }
    if (cfg->skip_l == 0 || cfg->skip_l == 1 || cfg->skip_l >= cfg->n_lin) { set_error("unsupported skip layer %d", cfg->skip_l); return EMAP_E_INVALID; }
    if (!(cfg->scale > 0.f)) { set_error("scale must be > 0"); return EMAP_E_INVALID; }
    const int H = cfg->d_hidden;
    L->H = H; L->n_lin = cfg->n_lin; L->skip_l = cfg->skip_l; L->multires = cfg->multires;
    L->d0 = 3 + 6 * cfg->multires; L->nparts = prec_nparts(prec); L->is_f16 = prec_is_f16(prec) ? 1 : 0;
    L->mx_fwd = (prec == EMAP_PREC_F16X3M) ? 1 : 0;
    L->mx_bwd = (prec == EMAP_PREC_F16X3E) ? 0 : 1;
    L->udf_type = cfg->udf_type; L->scale = cfg->scale;
    int frag = 0, chunks = 0;
    for (int l = 0; l < cfg->n_lin; ++l) {
        LayerDesc& d = L->layer[l];
        const bool last = (l == cfg->n_lin - 1);
        d.out_dim = last ? 1 : ((l + 1 == cfg->skip_l) ? H - L->d0 : H);
        d.pe_ks = (l == 0 || l == cfg->skip_l) ? PE_KS : 0;
        d.h_ks = (l == 0) ? 0 : H / 32;
        d.in_prev = (l == 0) ? 0 : ((l == cfg->skip_l) ? H - L->d0 : H);
        d.n_pairs = (d.out_dim + 31) / 32;
        d.frag_off = frag;
        d.act = last ? 0 : 1;
        d.pad = 0;
        frag += d.n_pairs * (d.pe_ks + d.h_ks) * 2 * L->nparts;
        chunks += d.n_pairs;
    }
    L->total_frags = frag;
    L->n_chunks = chunks;
    L->bias_off_bytes = 0;
    L->rowscale_off_bytes = cfg->n_lin * H * 4;
    L->frag_off_bytes = ((2 * cfg->n_lin * H * 4 + 1023) / 1024) * 1024;
    if (frag > 65535) { set_error("network too large for the fragment table"); return EMAP_E_INVALID; }
    // transposed section for the reverse sweep
    L->has_rev = (cfg->skip_l < cfg->n_lin - 1) ? 1 : 0;
    L->wlast_off_bytes = L->frag_off_bytes + frag * FRAG_BYTES;
    L->t_frag_off_bytes = L->wlast_off_bytes + ((H * 4 + 1023) / 1024) * 1024;
    int tf = 0;
    for (int l = 0; l < cfg->n_lin; ++l) {
        L->t_off[l] = -1; L->tpe_off[l] = -1;
        if (l >= 1 && l < cfg->n_lin - 1) { L->t_off[l] = tf; tf += ((L->layer[l].in_prev + 31) / 32) * (H / 32) * 2 * L->nparts; }
        if (l == 0 || l == cfg->skip_l) { L->tpe_off[l] = tf; tf += 2 * (H / 32) * 2 * L->nparts; }
    }
    L->t_total_frags = tf;   // always packed: the training backward (udf_mlp_vjp.inc) needs it for every topology
    L->r32_frag_off_bytes = L->t_frag_off_bytes + tf * FRAG_BYTES;
    L->r32_t_frag_off_bytes = L->r32_frag_off_bytes + frag * FRAG_BYTES;
    // MX operands of the training sweep (emap_common.h): hidden-K GEMMs of the forward layers 1 .. n_lin-1 (the last layer's one
    // real row included: its x_lo fragments no longer exist in the sweep's exchange buffer) and of the reverse steps n_lin-2 .. 1
    // precise weight gradients: hi + lo parts of both operands of dW = sum Z A^T (wgrad.hip: three passes over twice the stash).  The mode without
    // MX fp6 anywhere (f16x3e) is the one that asks for margin; there the sweep is the f16 one, whose lo fragments are f16 as well.
    L->wgrad_lo = (prec == EMAP_PREC_F16X3E && L->is_f16 && L->nparts == 2) ? 1 : 0;
    L->sweep_mx = (EMAP_SWEEP_MX && L->is_f16 && L->nparts == 2 && H == 256 && prec != EMAP_PREC_F16X3E && L->has_rev) ? 1 : 0;
    L->swm_units = 0;
    L->swm_off_bytes = (int32_t)(((size_t)L->r32_t_frag_off_bytes + (size_t)tf * FRAG_BYTES + 255) & ~(size_t)255);
    for (int l = 0; l < cfg->n_lin; ++l) { L->swm_unit[l] = -1; L->swm_t_unit[l] = -1; }
    if (L->sweep_mx) {
        const int n128 = H / 128;
        for (int l = 1; l < cfg->n_lin; ++l) { L->swm_unit[l] = L->swm_units; L->swm_units += L->layer[l].n_pairs * n128; }
        for (int l = 1; l < cfg->n_lin - 1; ++l) { L->swm_t_unit[l] = L->swm_units; L->swm_units += ((L->layer[l].in_prev + 31) / 32) * n128; }
    }
    return EMAP_OK;
}

void build_vjp_layout(const NetLayout& L, VjpLayout* V) {
    memset(V, 0, sizeof(*V));
    int a = 0, z = 0, s = 0;
    V->a_rt[0] = 2 * PE_KS; V->a_off[0] = a; a += V->a_rt[0] * 2;
    for (int l = 0; l < L.n_lin; ++l) {
        const bool last = (l == L.n_lin - 1);
        if (!last) {
            V->a_rt[l + 1] = 2 * L.layer[l].n_pairs; V->a_off[l + 1] = a; a += V->a_rt[l + 1] * 2;
            V->s_off[l] = s; s += 2 * L.layer[l].n_pairs;
        }
        V->z_rt[l] = last ? 1 : 2 * L.layer[l].n_pairs; V->z_off[l] = z; z += V->z_rt[l] * 2;
    }
    V->a_tile_kb = a; V->z_tile_kb = z;
    (void)s;
    // per workgroup: [hidden layer][tile pair][4 x 64 lanes x 16 B] lane-linear (a' as f16 / bf16 hi part, sigma' as unorm16); the f16 sweep of the
    // split-fp16 modes adds two planes with the lo parts of a' (udf_mlp_vjp.inc, SLABLO)
    const bool slablo = !L.sweep_mx && L.is_f16 && L.nparts == 2;
    V->s_slab_kb = (L.n_lin - 1) * (L.H / 32) * (slablo ? 6 : 4);
}

__global__ __launch_bounds__(256) void pack_all_kernel(const PackArgs a, unsigned nb0, unsigned nb1, unsigned nb2) {
    const unsigned b = blockIdx.x;
#ifndef EMAP_PACK_SECTIONS
#define EMAP_PACK_SECTIONS 63      // timing builds: bit per section (scripts/r5/pack_time.py)
#endif
    if (b < nb0) { if (EMAP_PACK_SECTIONS & 1) pack_body(a, b); }
    else if (b < nb0 + nb1) { if (EMAP_PACK_SECTIONS & 2) pack_t_body(a, b - nb0); }
    else if (b < 2 * nb0 + nb1) { if (EMAP_PACK_SECTIONS & 4) pack32_body(a, b - nb0 - nb1); }
    else if (b < 2 * (nb0 + nb1)) { if (EMAP_PACK_SECTIONS & 8) pack32_t_body(a, b - 2 * nb0 - nb1); }
    else if (b < 2 * (nb0 + nb1) + nb2) { if (EMAP_PACK_SECTIONS & 16) pack_wlast_body(a, b - 2 * (nb0 + nb1)); }
    else { if (EMAP_PACK_SECTIONS & 32) pack_swm_body(a, b - 2 * (nb0 + nb1) - nb2); }
}

int launch_pack(const NetLayout& L, const float* const* g, const float* const* v, const float* const* b,
                void* packed, hipStream_t st) {
    PackArgs a;
    for (int l = 0; l < L.n_lin; ++l) {
        a.g[l] = g[l]; a.v[l] = v[l]; a.b[l] = b[l];
        a.in_dim[l] = (l == 0) ? L.d0 : L.H;  // row length of v[l] (udf_model.py:24-45)
    }
    a.L = L;
    a.packed = static_cast<char*>(packed);
    const int rows = L.n_lin * L.H;
    hipLaunchKernelGGL(rowscale_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, a);
    const long long threads = (long long)L.total_frags * 64;
    // forward fragments, transposed fragments and the last layer's fp32 row in ONE launch (they only depend on the row scales): the
    // training step re-packs after every optimizer update, and each launch of these small kernels is ~5 us of latency
    const long long tthreads = (long long)L.t_total_frags * 64;
    const unsigned nb0 = (unsigned)((threads + 255) / 256), nb1 = (unsigned)((tthreads + 255) / 256), nb2 = (unsigned)((L.H + 255) / 256);
    const unsigned nb3 = L.sweep_mx ? (unsigned)(((long long)L.swm_units * 128 + 255) / 256) : 0u;
    hipLaunchKernelGGL(pack_all_kernel, dim3(2 * (nb0 + nb1) + nb2 + nb3), dim3(256), 0, st, a, nb0, nb1, nb2);
    return check_launch("pack_weights");
}

int launch_mlp_bf16(const NetLayout&, const void*, const PointSource&, int64_t, float*, float*, hipStream_t, int, int32_t*, void*, const CompositeFuse*);
int launch_mlp_bf16x3(const NetLayout&, const void*, const PointSource&, int64_t, float*, float*, hipStream_t, int, int32_t*, void*, const CompositeFuse*);
int launch_mlp_f16(const NetLayout&, const void*, const PointSource&, int64_t, float*, float*, hipStream_t, int, int32_t*, void*, const CompositeFuse*);
int launch_mlp_f16x3(const NetLayout&, const void*, const PointSource&, int64_t, float*, float*, hipStream_t, int, int32_t*, void*, const CompositeFuse*);

// kernel variant: 2 = "fs2" (udf_mlp_fs2_kernel: every value launch, forward-mode tangents for small grad launches),
// 3 = "rev" (grad launches only: forward + reverse sweep on 32x32 MFMA tiles, udf_mlp_rev32.inc).
// How d(udf)/dx is computed is a process-wide setting (emap_set_grad_mode; -1 = by launch size, 0 = always forward-mode tangents,
// 1 = always the reverse sweep), initialised ONCE at library load from EMAP_GRAD_MODE=fwd|rev: no getenv on the launch path
// (rounds 1-3 read two environment variables per launch).
static int grad_mode_from_env() {
    const char* gm = getenv("EMAP_GRAD_MODE");
    return !gm ? -1 : (!strcmp(gm, "fwd") ? 0 : (!strcmp(gm, "rev") ? 1 : -1));
}
static std::atomic<int> g_grad_mode{grad_mode_from_env()};   // process-wide; atomic: callers on several threads / devices read it per launch
int set_grad_mode(int mode) { return g_grad_mode.exchange((mode == 0 || mode == 1) ? mode : -1, std::memory_order_relaxed); }
static int mlp_variant(const NetLayout& L, int prec, int64_t P, bool grad) {
    // reverse mode halves the MFMA work of a grad launch but a tile is two dependent sweeps: it wins once most CUs have a
    // workgroup (measured crossover: the split modes between 8k and 12k points, single-pass modes at 16k).  Round 6 (scripts/r6/gpu_grad_mode_sweep.py,
    // profiles/r06_grad_mode_sweep.txt): the forward-mode kernel (16 points per workgroup, two per CU) starts its second round at 8 193 points - 129 us
    // against the reverse sweep's 94 from there on (85 against 94 at 8 192): the split modes switch at 8 193 (rounds 3-5: 10 240).
    const int64_t rev_min = (prec == EMAP_PREC_F16X3 || prec == EMAP_PREC_F16X3M || prec == EMAP_PREC_F16X3E || prec == EMAP_PREC_BF16X3) ? 8193 : 16384;
    const int gm = g_grad_mode.load(std::memory_order_relaxed);
    if (grad && L.has_rev && gm != 0 && (P >= rev_min || gm == 1)) return 3;
    return 2;
}

bool mlp_uses_rev(const NetLayout& L, int prec, int64_t P) { return mlp_variant(L, prec, P, true) == 3; }

int launch_mlp(const NetLayout& L, const void* packed, int prec, const PointSource& src, int64_t P, float* udf,
               float* grad3, hipStream_t st, int32_t* err_flags, void* scratch, const CompositeFuse* fuse) {
    const int v = mlp_variant(L, prec, P, grad3 != nullptr);
    if (fuse && v != 3) { set_error("launch_mlp: the fused compositing tail needs the reverse-sweep kernel (mlp_uses_rev)"); return EMAP_E_INVALID; }
    switch (prec) {
        case EMAP_PREC_BF16: return launch_mlp_bf16(L, packed, src, P, udf, grad3, st, v, err_flags, scratch, fuse);
        case EMAP_PREC_BF16X3: return launch_mlp_bf16x3(L, packed, src, P, udf, grad3, st, v, err_flags, scratch, fuse);
        case EMAP_PREC_F16: return launch_mlp_f16(L, packed, src, P, udf, grad3, st, v, err_flags, scratch, fuse);
        case EMAP_PREC_F16X3:
        case EMAP_PREC_F16X3E:
        case EMAP_PREC_F16X3M: return launch_mlp_f16x3(L, packed, src, P, udf, grad3, st, v, err_flags, scratch, fuse);   // L.mx_fwd selects the kernel
    }
    set_error("unknown precision mode %d", prec);
    return EMAP_E_INVALID;
}

int launch_is_bf16(const NetLayout&, const void*, const IsLaunch&, hipStream_t, int32_t*);
int launch_is_bf16x3(const NetLayout&, const void*, const IsLaunch&, hipStream_t, int32_t*);
int launch_is_f16(const NetLayout&, const void*, const IsLaunch&, hipStream_t, int32_t*);
int launch_is_f16x3(const NetLayout&, const void*, const IsLaunch&, hipStream_t, int32_t*);

static int fused_sampling_from_env() {      // EMAP_FUSED_SAMPLING=0: the launch chain; 2: fused whatever the launch size (A/B) (read ONCE at load, like EMAP_GRAD_MODE; at run time: emap_set_fused_sampling)
    const char* e = getenv("EMAP_FUSED_SAMPLING");
    return (e && e[0] == '0') ? 0 : ((e && e[0] == '2') ? 2 : 1);
}
static std::atomic<int> g_fused_sampling{fused_sampling_from_env()};
int set_fused_sampling(int on) { return g_fused_sampling.exchange(on < 0 ? 0 : (on > 2 ? 2 : on), std::memory_order_relaxed); }
int fused_sampling_mode() { return g_fused_sampling.load(std::memory_order_relaxed); }

// Wide value launches as the forward sweep of the 32x32 kernel (udf_mlp_rev32.inc, VAL).  OFF by default - measured round 6, same box, interleaved
// (profiles/r06_value32_ab.txt): one 32 768-point launch 95.6 vs 97.4 us, but +3 ... +5 % from 65 536 points on; render from a graph 0.5387 vs
// 0.5435 ms at 512 rays (-0.9 %), 1.019 vs 1.019 at 1024, 3.998 vs 4.135 at 4096 (-3.3 %: the cooler value kernels leave the final pass more
// clock under the package power cap).  Its sums run in another order than the 16x16 kernel's (udf differs by <= 4.5e-7 of the maximum): a render's
// z_vals would depend on whether its launch has 512 tiles - a sub-batch would no longer reproduce its rows bit for bit - for a gain inside the
// spread of the boxes.  EMAP_VALUE32=1 (read ONCE at load) / emap_set_value_tile_mode(1) turn it on.
static int value_tile_from_env() {
    const char* e = getenv("EMAP_VALUE32");
    return (e && e[0] == '1') ? 1 : 0;
}
static std::atomic<int> g_value_tile{value_tile_from_env()};
int set_value_tile_mode(int on) { return g_value_tile.exchange(on ? 1 : 0, std::memory_order_relaxed); }
int value_tile_mode() { return g_value_tile.load(std::memory_order_relaxed); }

int launch_importance(const NetLayout& L, const void* packed, int prec, const IsLaunch& q, hipStream_t st, int32_t* err_flags) {
    if (!g_fused_sampling.load(std::memory_order_relaxed)) return IS_NOT_FUSED;
    switch (prec) {
        case EMAP_PREC_BF16: return launch_is_bf16(L, packed, q, st, err_flags);
        case EMAP_PREC_BF16X3: return launch_is_bf16x3(L, packed, q, st, err_flags);
        case EMAP_PREC_F16: return launch_is_f16(L, packed, q, st, err_flags);
        case EMAP_PREC_F16X3:
        case EMAP_PREC_F16X3E:
        case EMAP_PREC_F16X3M: return launch_is_f16x3(L, packed, q, st, err_flags);
    }
    set_error("unknown precision mode %d", prec);
    return EMAP_E_INVALID;
}

}  // namespace emap
