// udf_mlp.hip - positional encoding + weight-normed Softplus(beta=100) UDF MLP, value and grad_x,
// as ONE fused MFMA kernel per call for gfx950 (MI355X).
//
// Reference behaviour being reproduced (cvg/EMAP):
//   Embedder.embed              src/models/embedder.py:10-35
//   UDFNetwork.forward/udf      src/models/udf_model.py:90-116
//   UDFNetwork.gradient         src/models/udf_model.py:121-135   (autograd there; forward-mode here)
//   weight_norm parametrization src/models/udf_model.py:73-74     (folded once in pack_kernel)
//
// Design (see DESIGN.md "MLP kernel"):
//   * Z^T = W . X^T with v_mfma_f32_16x16x32_bf16: A = 16 output features x 32 k (weights, streamed
//     L2 -> LDS with global_load_lds, shared by the 4 waves of a workgroup), B = 32 k x 16 columns
//     (activations, resident in VGPRs for the whole network), C/D = fp32.
//   * A wave owns NCT column tiles of 16 columns.  A column is a point (value kernel) or one of
//     {value, d/dx, d/dy, d/dz} of a point (grad kernel: forward-mode tangents, so no activation
//     is ever stored; a'_k+ = sigmoid(100 z) * z'_k).
//   * No transposes / no LDS round trip between layers: lane group g of the D fragment holds rows
//     4g..4g+3 of each 16-feature tile, which is exactly what it must supply as k-values
//     8 per K-step to the next layer once the next layer's K dimension is permuted accordingly
//     (pack_kernel writes the weights in that order).
//   * EMAP_PREC_BF16   : NCT=4, one MFMA pass.
//     EMAP_PREC_BF16X3 : NCT=2, a = a_hi + a_lo, w = w_hi + w_lo, three MFMA passes (drops lo*lo):
//                        ~2^-17 relative, the mode the 1e-4 parity gate runs in.
//   * one wave per SIMD (up to 512 VGPRs), 4 waves per workgroup, persistent over point tiles.
#include "emap_common.h"
#include <type_traits>
#include <utility>

namespace emap {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NBUF = 3;          // LDS ring depth (weight chunks)
constexpr int MAX_CHUNKS = 96;   // EMAP_MAX_LIN * 8 pairs

struct MlpArgs {
    const char* frags;        // packed fragments
    const float* bias;        // n_lin * H
    PointSource src;
    long long P;
    float* udf;
    float* grad;
    int32_t n_tiles;
    int32_t n_lin;
    int32_t multires;
    int32_t udf_type;
    float scale;
    int32_t n_chunks;
    LayerDesc layer[EMAP_MAX_LIN];
};

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    // one wave-instruction: 64 lanes x 16 B global -> LDS[lds_dst + lane*16]   (LDS-DMA)
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
        : "memory");
}

__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// softplus(beta=100)(z) and its derivative sigmoid(100 z)   (nn.Softplus(beta=100), udf_model.py:78)
__device__ __forceinline__ void softplus_sig(float z, float& a, float& s) {
    const float t = __expf(-100.0f * fabsf(z));
    const float l = __logf(1.0f + t);
    a = fmaxf(z, 0.0f) + 0.01f * l;
    const float r = __builtin_amdgcn_rcpf(1.0f + t);
    s = (z >= 0.0f) ? r : t * r;
}
__device__ __forceinline__ float softplus100(float z) {
    const float t = __expf(-100.0f * fabsf(z));
    return fmaxf(z, 0.0f) + 0.01f * __logf(1.0f + t);
}

// sin and cos of a (|a| up to a few thousand) with ~1e-7 absolute error: Cody-Waite reduction by
// pi/2 in three parts + minimax polynomials on [-pi/4, pi/4].
__device__ __forceinline__ void sincos_acc(float a, float& sn, float& cs) {
    const float n = rintf(a * 0.63661977236758134308f);  // a * 2/pi
    float r = fmaf(-n, 1.5703125f, a);                   // pi/2 = 1.5703125 + 4.83751296997e-4 + 7.549789954e-8
    r = fmaf(-n, 4.83751296997070312e-4f, r);
    r = fmaf(-n, 7.54978995489188216e-8f, r);
    const float r2 = r * r;
    float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(sp, r2, -1.6666654611e-1f);
    sp = fmaf(sp * r2, r, r);
    float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(cp, r2, 4.166664568298827e-2f);
    cp = fmaf(cp * r2, r2, fmaf(-0.5f, r2, 1.0f));
    const int q = ((int)n) & 3;
    const float s0 = (q & 1) ? cp : sp;
    const float c0 = (q & 1) ? sp : cp;
    sn = (q & 2) ? -s0 : s0;
    cs = ((q + 1) & 2) ? -c0 : c0;
}

__device__ __forceinline__ float dpp_ror8(float v) {
    // lane j of each 16-lane row reads lane (j+8)%16 of the same row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
}

// o + d * zz with a separately rounded product, like the reference's elementwise torch ops
// (rays_o[:, None, :] + rays_d[:, None, :] * z, udf_renderer_blending.py:448-450,812).
__device__ __forceinline__ float ray_at(float o, float d, float zz) { return __fadd_rn(o, __fmul_rn(d, zz)); }

// Fetch a point (already multiplied by `scale`).
__device__ __forceinline__ void load_point(const PointSource& s, long long p, float scale, float xs[3]) {
    if (s.x) {
        xs[0] = s.x[3 * p + 0] * scale;
        xs[1] = s.x[3 * p + 1] * scale;
        xs[2] = s.x[3 * p + 2] * scale;
    } else {
        const int n = s.n_per_ray;
        const long long ray = p / n;
        const int i = (int)(p - ray * n);
        float zz = s.z[ray * n + i];
        if (s.mid) {
            const float d = (i + 1 < n) ? __fsub_rn(s.z[ray * n + i + 1], zz) : *s.sample_dist;
            zz = __fadd_rn(zz, __fmul_rn(d, 0.5f));  // mid_z_vals = z_vals + dists * 0.5 (udf_renderer_blending.py:446)
        }
        xs[0] = ray_at(s.rays_o[3 * ray + 0], s.rays_d[3 * ray + 0], zz) * scale;
        xs[1] = ray_at(s.rays_o[3 * ray + 1], s.rays_d[3 * ray + 1], zz) * scale;
        xs[2] = ray_at(s.rays_o[3 * ray + 2], s.rays_d[3 * ray + 2], zz) * scale;
    }
}

// The PE block: lane group g owns angle indices a = 8*g + q, q = 0..7, and for each the slot pair
//   a <  3M        : k = a/3, c = a%3 -> (sin(2^k x_c), cos(2^k x_c))          embedder.py:26-29
//   a == 3M        : (x_0, x_1)   a == 3M+1 : (x_2, 0)   else (0, 0)            embedder.py:14-16
struct PeAngle { float sn, cs, f; int c; bool is_ang; int a; };
__device__ __forceinline__ PeAngle pe_angle(const float xs[3], int a, int M) {
    PeAngle r;
    const int k = a / 3;
    r.c = a - 3 * k;
    r.a = a;
    const float xc = (r.c == 0) ? xs[0] : ((r.c == 1) ? xs[1] : xs[2]);
    r.f = __builtin_bit_cast(float, (127 + k) << 23);  // 2^k exactly
    sincos_acc(xc * r.f, r.sn, r.cs);
    r.is_ang = a < 3 * M;
    return r;
}
// type 0 = value, type 1..3 = d/dx_{type-1}
__device__ __forceinline__ void pe_pair(const PeAngle& r, const float xs[3], int M, int type, float& v0, float& v1) {
    const int a = r.a;
    if (type == 0) {
        v0 = r.is_ang ? r.sn : ((a == 3 * M) ? xs[0] : ((a == 3 * M + 1) ? xs[2] : 0.0f));
        v1 = r.is_ang ? r.cs : ((a == 3 * M) ? xs[1] : 0.0f);
    } else {
        const int tc = type - 1;
        const float d0 = (r.c == tc) ? r.f * r.cs : 0.0f;
        const float d1 = (r.c == tc) ? -r.f * r.sn : 0.0f;
        const float r0 = (a == 3 * M) ? (tc == 0 ? 1.0f : 0.0f) : ((a == 3 * M + 1) ? (tc == 2 ? 1.0f : 0.0f) : 0.0f);
        const float r1 = (a == 3 * M) ? (tc == 1 ? 1.0f : 0.0f) : 0.0f;
        v0 = r.is_ang ? d0 : r0;
        v1 = r.is_ang ? d1 : r1;
    }
}

template <int NPART>
__device__ __forceinline__ void split_store(float v, bf16x8 (&dst)[NPART], int e) {
    const __bf16 hi = (__bf16)v;
    dst[0][e] = hi;
    if constexpr (NPART == 2) dst[1][e] = (__bf16)(v - (float)hi);
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
// softplus(beta=100) and its derivative on the raw v_exp_f32 / v_log_f32 / v_rcp_f32 pipes (nn.Softplus(beta=100),
// udf_model.py:78).  With e = exp(-100 z) (exponent clamped so that e stays finite):
//     softplus(z) = max(z + log(1 + e)/100, 0)        sigmoid(100 z) = 1 / (1 + e)
// The first identity is exact for every z; for z << 0 the sum cancels to an absolute error of one ulp of |z|
// (<= 6e-8 for |z| < 1), far below the bf16 / split-bf16 quantisation of the activations that follows.
__device__ __forceinline__ float vmax0(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));  // plain max(x, 0): no canonicalising second v_max
    return r;
}
__device__ __forceinline__ float softplus_fast(float z) {
    const float t = __builtin_amdgcn_exp2f(fabsf(z) * -144.26950408889634f);  // exp(-100|z|)
    const float l = __builtin_amdgcn_logf(1.0f + t);                          // log2(1+t)
    return fmaf(l, 6.9314718055994531e-3f, vmax0(z));                         // max(z,0) + ln2/100 * log2(1+t)
}
__device__ __forceinline__ void softplus_sig_fast(float z, float& a, float& s) {
    const float e = __builtin_amdgcn_exp2f(fminf(z * -144.26950408889634f, 126.0f));
    const float u = 1.0f + e;
    a = vmax0(fmaf(__builtin_amdgcn_logf(u), 6.9314718055994531e-3f, z));
    s = __builtin_amdgcn_rcpf(u);
}

enum { KIND_FIRST = 0, KIND_NORMAL = 1, KIND_SKIP = 2 };

// compile-time loop: f(integral_constant<int, 0>{}), ..., f(integral_constant<int, N-1>{}) - every register
// array index derived from the loop variable is a constant expression (nothing can fall back to scratch)
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

template <int H, int MODE, int NCT, bool GRAD, int WPB>
__global__ __launch_bounds__(WPB * 64, WPB / 4) void udf_mlp_kernel(const MlpArgs a) {
    constexpr int NPART = (MODE == EMAP_PREC_BF16) ? 1 : 2;
    constexpr int NKS = H / 32;
    constexpr int NPAIR = H / 32;
    constexpr int PW = GRAD ? (NCT == 4 ? 16 : 8) : 16 * NCT;  // points per wave
    constexpr int CHUNK_BYTES = (PE_KS + NKS) * 2 * NPART * FRAG_BYTES;
    constexpr int NFR = 2 * NPART;  // fragments per K-step: [t][part]
    static_assert(!GRAD || NCT == 4 || NCT == 2, "grad kernel: 4 column tiles (types) or 2 (half-tile types)");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bias_lds = reinterpret_cast<float*>(smem);                 // n_lin * H floats
    const int bias_bytes = ((a.n_lin * H * 4 + 1023) / 1024) * 1024;
    char* ring = smem + bias_bytes;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)ring;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int j = lane & 15;

    for (int i = tid; i < a.n_lin * H; i += WPB * 64) bias_lds[i] = a.bias[i];
    __syncthreads();

    // ---- weight-chunk pipeline state (all wave-uniform, no divisions / table look-ups in the hot loop) ----
    // The packed fragments are one linear stream of chunks (layer-major, pair-minor); a prefetch cursor runs
    // exactly two chunks ahead of the compute cursor and wraps to layer 0 for the next point tile.
    const int my_tiles = (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    int pf_left = my_tiles * a.n_chunks;  // chunks still to issue
    int pf_l = 0, pf_p = 0;               // prefetch cursor
    int pf_nf = (a.layer[0].pe_ks + a.layer[0].h_ks) * 2 * NPART, pf_off = a.layer[0].frag_off, pf_np = a.layer[0].n_pairs;
    int pf_slot = 0;                      // ring slot the next issue writes
    int cs_slot = 0;                      // ring slot the next compute reads
    int last_lpc = 0;                     // loads per wave of the most recently issued chunk
    auto issue_chunk = [&]() {
        if (pf_left > 0) {
            const char* gbase = a.frags + (size_t)(pf_off + pf_p * pf_nf) * FRAG_BYTES + lane * 16;
            const uint32_t lbase = ring_lds + (uint32_t)(pf_slot * CHUNK_BYTES);
            for (int f = wave; f < pf_nf; f += WPB) glds16(gbase + f * FRAG_BYTES, lbase + f * FRAG_BYTES);
            last_lpc = (pf_nf - wave + WPB - 1) / WPB;  // this wave's share of the chunk's loads
            --pf_left;
            if (++pf_p == pf_np) {
                pf_p = 0;
                pf_l = (pf_l + 1 == a.n_lin) ? 0 : pf_l + 1;
                pf_nf = (a.layer[pf_l].pe_ks + a.layer[pf_l].h_ks) * 2 * NPART;
                pf_off = a.layer[pf_l].frag_off;
                pf_np = a.layer[pf_l].n_pairs;
            }
        } else {
            last_lpc = 0;
        }
        pf_slot = (pf_slot + 1 == NBUF) ? 0 : pf_slot + 1;
    };

    if (my_tiles > 0) {
        issue_chunk();
        issue_chunk();
    }

    // column -> (point, type) map of this lane
    auto col_type = [&](int ct) -> int { return !GRAD ? 0 : (NCT == 4 ? ct : 2 * ct + (j >> 3)); };
    auto col_point = [&](int ct) -> int { return !GRAD ? 16 * ct + j : (NCT == 4 ? j : (j & 7)); };

    bf16x8 in[NPART][NCT][NKS];
    bf16x8 out[NPART][NCT][NPAIR];
    bf16x8 pe[NPART][NCT][PE_KS];
#pragma unroll
    for (int pt = 0; pt < NPART; ++pt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                in[pt][ct][s] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                out[pt][ct][s] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }

    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const long long pbase = ((long long)tile * WPB + wave) * PW;

        // ---- positional encoding block (values and, for GRAD, tangents) ----
        {
            float xs[3];
            if constexpr (GRAD) {
                long long p = pbase + col_point(0);
                if (p >= a.P) p = a.P - 1;
                load_point(a.src, p, a.scale, xs);
                // one sincos per angle of this lane's point; every column type is derived from it
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const PeAngle ang = pe_angle(xs, 8 * g + q, a.multires);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        float v0, v1;
                        pe_pair(ang, xs, a.multires, col_type(ct), v0, v1);
                        const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
                        pe[0][ct][q >> 2][2 * (q & 3)] = h0;
                        pe[0][ct][q >> 2][2 * (q & 3) + 1] = h1;
                        if constexpr (NPART == 2) {
                            pe[1][ct][q >> 2][2 * (q & 3)] = (__bf16)(v0 - (float)h0);
                            pe[1][ct][q >> 2][2 * (q & 3) + 1] = (__bf16)(v1 - (float)h1);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    long long p = pbase + col_point(ct);
                    if (p >= a.P) p = a.P - 1;
                    load_point(a.src, p, a.scale, xs);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const PeAngle ang = pe_angle(xs, 8 * g + q, a.multires);
                        float v0, v1;
                        pe_pair(ang, xs, a.multires, 0, v0, v1);
                        const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
                        pe[0][ct][q >> 2][2 * (q & 3)] = h0;
                        pe[0][ct][q >> 2][2 * (q & 3) + 1] = h1;
                        if constexpr (NPART == 2) {
                            pe[1][ct][q >> 2][2 * (q & 3)] = (__bf16)(v0 - (float)h0);
                            pe[1][ct][q >> 2][2 * (q & 3) + 1] = (__bf16)(v1 - (float)h1);
                        }
                    }
                }
            }
        }

        // ---- layers ----
        for (int l = 0; l < a.n_lin; ++l) {
            const LayerDesc L = a.layer[l];
            const bool last = (l == a.n_lin - 1);
            const int kind = (L.h_ks == 0) ? KIND_FIRST : (L.pe_ks ? KIND_SKIP : KIND_NORMAL);

            // Two accumulator sets: while pair p accumulates into acc[p&1], the activation epilogue of pair
            // p-1 (acc[(p-1)&1]) is issued in slices between the K-steps, so its VALU/transcendental work
            // runs in the shadow of the MFMA pipe (one wave per SIMD: nobody else would fill it).
            // (with two waves per SIMD the SIMD partner provides that overlap: one accumulator set, epilogue in line)
            constexpr int NACC = (WPB == 8) ? 1 : 2;
            f32x4 acc[NACC][2][NCT];
            int cur_slot = 0;

            // epilogue slice `sl` (of NSL) of pair `pp`, reading acc set `par`.  A slice always produces an even
            // number of consecutive bf16 elements of each output fragment, so they convert with v_cvt_pk_bf16_f32.
            constexpr int NSL = GRAD ? 4 : 2 * NCT;
            auto put2 = [&](auto pp_c, int ct, int e0, float x0, float x1) {
                constexpr int pp = decltype(pp_c)::value;
                const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
                out[0][ct][pp][e0] = h0;
                out[0][ct][pp][e0 + 1] = h1;
                if constexpr (NPART == 2) {
                    out[1][ct][pp][e0] = (__bf16)(x0 - (float)h0);
                    out[1][ct][pp][e0 + 1] = (__bf16)(x1 - (float)h1);
                }
            };
            auto epi_slice = [&](auto par_c, auto pp_c, auto sl_c) {
                constexpr int par = decltype(par_c)::value, sl = decltype(sl_c)::value;
                if constexpr (!GRAD) {
                    constexpr int ct = sl >> 1, t = sl & 1;
#pragma unroll
                    for (int r = 0; r < 4; r += 2)
                        put2(pp_c, ct, 4 * t + r, softplus_fast(acc[par][t][ct][r]), softplus_fast(acc[par][t][ct][r + 1]));
                } else if constexpr (NCT == 4) {
                    constexpr int t = sl >> 1, r = 2 * (sl & 1);
                    float a0, s0, a1, s1;
                    softplus_sig_fast(acc[par][t][0][r], a0, s0);
                    softplus_sig_fast(acc[par][t][0][r + 1], a1, s1);
                    put2(pp_c, 0, 4 * t + r, a0, a1);
#pragma unroll
                    for (int ct = 1; ct < NCT; ++ct) put2(pp_c, ct, 4 * t + r, s0 * acc[par][t][ct][r], s1 * acc[par][t][ct][r + 1]);
                } else {
                    // NCT == 2: tile 0 = [8 value | 8 d/dx], tile 1 = [8 d/dy | 8 d/dz]
                    constexpr int t = sl >> 1, r = 2 * (sl & 1);
                    const bool isv = (j < 8);
                    float av[2], se[2], sw[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float sg;
                        softplus_sig_fast(acc[par][t][0][r + i], av[i], sg);
                        sw[i] = dpp_ror8(sg);
                        se[i] = isv ? sg : sw[i];
                    }
                    put2(pp_c, 0, 4 * t + r, isv ? av[0] : sw[0] * acc[par][t][0][r], isv ? av[1] : sw[1] * acc[par][t][0][r + 1]);
                    put2(pp_c, 1, 4 * t + r, se[0] * acc[par][t][1][r], se[1] * acc[par][t][1][r + 1]);
                }
            };

            // K-loop of one output pair for a layer kind; `par` = accumulator set, `prev` = pair whose epilogue
            // is interleaved (-1: none)
            auto pair_body = [&](auto kind_c, auto p_c, auto epi_c) {
                constexpr int KIND = decltype(kind_c)::value;
                constexpr bool do_epi = decltype(epi_c)::value;
                constexpr int p = decltype(p_c)::value;
                constexpr int NS = (KIND == KIND_FIRST) ? PE_KS : (KIND == KIND_NORMAL ? NKS : PE_KS + NKS);
                constexpr int par = (NACC == 2) ? (p & 1) : 0;
                const char* buf = ring + cur_slot * CHUNK_BYTES + lane * 16;
                // A-fragment prefetch distance: with two waves per SIMD the partner hides most of the LDS latency
                constexpr int PD = (WPB == 8) ? 1 : 2;
                bf16x8 af[PD + 1][NFR];
                auto load_a = [&](auto s_c) {
                    constexpr int s = decltype(s_c)::value;
#pragma unroll
                    for (int f = 0; f < NFR; ++f)
                        af[s % (PD + 1)][f] = *reinterpret_cast<const bf16x8*>(buf + (size_t)(s * NFR + f) * FRAG_BYTES);
                };
                static_for<PD>([&](auto i_c) { if constexpr (decltype(i_c)::value < NS) load_a(i_c); });
                static_for<NS>([&](auto s_c) {
                    constexpr int s = decltype(s_c)::value;
                    constexpr int sa = s % (PD + 1);
                    if constexpr (s + PD < NS) load_a(std::integral_constant<int, s + PD>{});
                    constexpr bool is_pe = (KIND == KIND_FIRST) || (KIND == KIND_SKIP && s < PE_KS);
                    constexpr int sb = is_pe ? s : ((KIND == KIND_SKIP) ? s - PE_KS : s);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        bf16x8 bh, bl;
                        if constexpr (is_pe) { bh = pe[0][ct][sb]; bl = pe[NPART - 1][ct][sb]; }
                        else { bh = in[0][ct][sb]; bl = in[NPART - 1][ct][sb]; }
                        if constexpr (NPART == 1) {
                            acc[par][0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[sa][0], bh, acc[par][0][ct], 0, 0, 0);
                            acc[par][1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[sa][1], bh, acc[par][1][ct], 0, 0, 0);
                        } else {
                            // fragment order [t][part]: 0 = t0 hi, 1 = t0 lo, 2 = t1 hi, 3 = t1 lo
                            acc[par][0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[sa][0], bl, acc[par][0][ct], 0, 0, 0);
                            acc[par][1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[sa][2], bl, acc[par][1][ct], 0, 0, 0);
                            acc[par][0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[sa][1], bh, acc[par][0][ct], 0, 0, 0);
                            acc[par][1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[sa][3], bh, acc[par][1][ct], 0, 0, 0);
                            acc[par][0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[sa][0], bh, acc[par][0][ct], 0, 0, 0);
                            acc[par][1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[sa][2], bh, acc[par][1][ct], 0, 0, 0);
                        }
                    }
                    if constexpr (p > 0 && do_epi && NACC == 2) {
                        static_for<NSL>([&](auto sl_c) {
                            constexpr int sl = decltype(sl_c)::value;
                            if constexpr (sl * NS / NSL == s)
                                epi_slice(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, (p > 0 ? p - 1 : 0)>{}, sl_c);
                        });
                    }
                });
            };

            static_for<NPAIR>([&](auto p_c) {
                constexpr int p = decltype(p_c)::value;
                if (p < L.n_pairs) {
                    // -- chunk (l, p) must have landed in every wave's view; then refill the ring --
                    wait_vmcnt(last_lpc);
                    __builtin_amdgcn_s_barrier();
                    issue_chunk();
                    cur_slot = cs_slot;
                    cs_slot = (cs_slot + 1 == NBUF) ? 0 : cs_slot + 1;

                    // -- accumulators start from the bias (value columns only) --
                    {
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_lds + l * H + 32 * p + 4 * g);
                        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias_lds + l * H + 32 * p + 16 + 4 * g);
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) {
                            const bool isv = (col_type(ct) == 0);
                            acc[(NACC == 2) ? (p & 1) : 0][0][ct] = isv ? b0 : zero;
                            acc[(NACC == 2) ? (p & 1) : 0][1][ct] = isv ? b1 : zero;
                        }
                    }
                    if (last) pair_body(std::integral_constant<int, KIND_NORMAL>{}, p_c, std::false_type{});
                    else if (kind == KIND_NORMAL) pair_body(std::integral_constant<int, KIND_NORMAL>{}, p_c, std::true_type{});
                    else if (kind == KIND_SKIP) pair_body(std::integral_constant<int, KIND_SKIP>{}, p_c, std::true_type{});
                    else pair_body(std::integral_constant<int, KIND_FIRST>{}, p_c, std::true_type{});

                    // the last pair of the layer has no successor to hide behind
                    if (!last && (NACC == 1 || p + 1 == L.n_pairs)) {
                        static_for<NSL>([&](auto sl_c) { epi_slice(std::integral_constant<int, ((NACC == 2) ? (p & 1) : 0)>{}, p_c, sl_c); });
                    }

                    if (last && p == 0) {
                        // ---- network output: feature 0 = tile 0 row 0 -> lane group 0, register 0 ----
                        const float inv_scale = 1.0f / a.scale;
                        if constexpr (!GRAD) {
                            if (g == 0) {
#pragma unroll
                                for (int ct = 0; ct < NCT; ++ct) {
                                    const long long pp = pbase + 16 * ct + j;
                                    const float h = acc[0][0][ct][0];
                                    const float u = (a.udf_type == EMAP_UDF_ABS) ? fabsf(h) : ((a.udf_type == EMAP_UDF_SQUARE) ? h * h : h);
                                    if (pp < a.P) a.udf[pp] = u * inv_scale;
                                }
                            }
                        } else if constexpr (NCT == 4) {
                            if (g == 0) {
                                const long long pp = pbase + j;
                                const float h = acc[0][0][0][0];
                                float u, m;
                                if (a.udf_type == EMAP_UDF_ABS) { u = fabsf(h); m = (h > 0.f) ? 1.f : ((h < 0.f) ? -1.f : 0.f); }
                                else if (a.udf_type == EMAP_UDF_SQUARE) { u = h * h; m = 2.f * h; }
                                else { u = h; m = 1.f; }
                                if (pp < a.P) {
                                    a.udf[pp] = u * inv_scale;
                                    a.grad[3 * pp + 0] = m * acc[0][0][1][0];
                                    a.grad[3 * pp + 1] = m * acc[0][0][2][0];
                                    a.grad[3 * pp + 2] = m * acc[0][0][3][0];
                                }
                            }
                        } else {
                            const float h0 = acc[0][0][0][0];    // lanes j<8: h      ; j>=8: dh/dx
                            const float h_sw = dpp_ror8(h0);     // lanes j>=8: h of their point
                            const float h = (j < 8) ? h0 : h_sw;
                            float u, m;
                            if (a.udf_type == EMAP_UDF_ABS) { u = fabsf(h); m = (h > 0.f) ? 1.f : ((h < 0.f) ? -1.f : 0.f); }
                            else if (a.udf_type == EMAP_UDF_SQUARE) { u = h * h; m = 2.f * h; }
                            else { u = h; m = 1.f; }
                            if (g == 0) {
                                const long long pp = pbase + (j & 7);
                                if (pp < a.P) {
                                    if (j < 8) {
                                        a.udf[pp] = u * inv_scale;
                                        a.grad[3 * pp + 1] = m * acc[0][0][1][0];
                                    } else {
                                        a.grad[3 * pp + 0] = m * h0;
                                        a.grad[3 * pp + 2] = m * acc[0][0][1][0];
                                    }
                                }
                            }
                        }
                    }
                }
            });
            // this layer's outputs are the next layer's inputs
#pragma unroll
            for (int pt = 0; pt < NPART; ++pt)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int s = 0; s < NKS; ++s) in[pt][ct][s] = out[pt][ct][s];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

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

__global__ __launch_bounds__(256) void pack_kernel(const PackArgs a) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;  // one thread per (fragment, lane)
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
    bf16x8 outv;
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
        float w = 0.f;
        if (o < Ld.out_dim && col >= 0 && col < n_in) w = rs[l * H + o] * a.v[l][(size_t)o * n_in + col] * mult;
        const __bf16 hi = (__bf16)w;
        outv[e] = (part == 0) ? hi : (__bf16)(w - (float)hi);
    }
    *reinterpret_cast<bf16x8*>(a.packed + a.L.frag_off_bytes + F * FRAG_BYTES + lane * 16) = outv;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int build_layout(const EmapNetConfig* cfg, int prec, NetLayout* L) {
    if (!cfg || !L) { set_error("null config"); return EMAP_E_INVALID; }
    if (cfg->d_hidden != 256 && cfg->d_hidden != 128) { set_error("d_hidden must be 128 or 256 (got %d)", cfg->d_hidden); return EMAP_E_INVALID; }
    if (cfg->n_lin < 2 || cfg->n_lin > EMAP_MAX_LIN) { set_error("n_lin out of range (%d)", cfg->n_lin); return EMAP_E_INVALID; }
    if (cfg->multires < 1 || cfg->multires > 10) { set_error("multires must be in 1..10 (got %d)", cfg->multires); return EMAP_E_INVALID; }
    if (cfg->d_out != 1) { set_error("d_out must be 1 (got %d): feature outputs are not on the hot path", cfg->d_out); return EMAP_E_INVALID; }
    if (prec != EMAP_PREC_BF16 && prec != EMAP_PREC_BF16X3) { set_error("unknown precision mode %d", prec); return EMAP_E_INVALID; }
    if (cfg->skip_l == 0 || cfg->skip_l == 1 || cfg->skip_l >= cfg->n_lin) { set_error("unsupported skip layer %d", cfg->skip_l); return EMAP_E_INVALID; }
    if (!(cfg->scale > 0.f)) { set_error("scale must be > 0"); return EMAP_E_INVALID; }
    const int H = cfg->d_hidden;
    L->H = H; L->n_lin = cfg->n_lin; L->skip_l = cfg->skip_l; L->multires = cfg->multires;
    L->d0 = 3 + 6 * cfg->multires; L->nparts = (prec == EMAP_PREC_BF16) ? 1 : 2;
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
    L->pad0 = 0;
    if (chunks > MAX_CHUNKS || frag > 65535) { set_error("network too large for the chunk table"); return EMAP_E_INVALID; }
    return EMAP_OK;
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
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a);
    return check_launch("pack_weights");
}

template <int H, int MODE, int NCT, bool GRAD, int WPB>
static int launch_mlp_t(const NetLayout& L, const void* packed, const PointSource& src, int64_t P, float* udf,
                        float* grad3, hipStream_t st) {
    constexpr int NPART = (MODE == EMAP_PREC_BF16) ? 1 : 2;
    constexpr int PW = GRAD ? (NCT == 4 ? 16 : 8) : 16 * NCT;
    constexpr int CHUNK_BYTES = (PE_KS + H / 32) * 2 * NPART * FRAG_BYTES;
    MlpArgs a;
    const char* pk = static_cast<const char*>(packed);
    a.frags = pk + L.frag_off_bytes;
    a.bias = reinterpret_cast<const float*>(pk + L.bias_off_bytes);
    a.src = src; a.P = P; a.udf = udf; a.grad = grad3;
    a.n_tiles = (int)((P + WPB * PW - 1) / (WPB * PW));
    a.n_lin = L.n_lin; a.multires = L.multires; a.udf_type = L.udf_type; a.scale = L.scale;
    a.n_chunks = L.n_chunks;
    for (int l = 0; l < L.n_lin; ++l) a.layer[l] = L.layer[l];
    const size_t lds = ((size_t)(L.n_lin * H * 4 + 1023) / 1024) * 1024 + (size_t)NBUF * CHUNK_BYTES;
    static bool attr_set = false;
    auto kern = udf_mlp_kernel<H, MODE, NCT, GRAD, WPB>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
            return EMAP_E_LAUNCH;
        }
        attr_set = true;
    }
    if (a.n_tiles <= 0) return EMAP_OK;
    int grid = a.n_tiles < 256 ? a.n_tiles : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WPB * 64), lds, st, a);
    return check_launch("udf_mlp");
}

// Geometry: bf16 kernels run 8 waves per workgroup (two per SIMD, <=256 registers each) so that one wave's
// activation epilogue (VALU/transcendental issue slots) overlaps its SIMD partner's MFMAs; a single wave cannot
// issue fast enough to keep the matrix pipe busy on its own.  The split-bf16 kernels need the whole register file
// (hi+lo activations) and run one wave per SIMD.  Small batches take fewer columns per wave / fewer waves per
// workgroup so that every CU still gets a workgroup (a wave's latency per tile is ~proportional to its columns).
int launch_mlp(const NetLayout& L, const void* packed, int prec, const PointSource& src, int64_t P, float* udf,
               float* grad3, hipStream_t st) {
    const bool grad = grad3 != nullptr;
    auto tiles = [&](int pts_per_wg) { return (P + pts_per_wg - 1) / pts_per_wg; };
#define EMAP_DISPATCH(HH)                                                                                        \
    if (L.H == HH) {                                                                                             \
        if (prec == EMAP_PREC_BF16) {                                                                            \
            if (grad) return launch_mlp_t<HH, EMAP_PREC_BF16, 4, true, 4>(L, packed, src, P, udf, grad3, st);     \
            if (tiles(256) >= 256) return launch_mlp_t<HH, EMAP_PREC_BF16, 2, false, 8>(L, packed, src, P, udf, grad3, st); \
            if (tiles(128) >= 128) return launch_mlp_t<HH, EMAP_PREC_BF16, 1, false, 8>(L, packed, src, P, udf, grad3, st); \
            return launch_mlp_t<HH, EMAP_PREC_BF16, 1, false, 4>(L, packed, src, P, udf, grad3, st);              \
        } else {                                                                                                 \
            if (grad) return launch_mlp_t<HH, EMAP_PREC_BF16X3, 2, true, 4>(L, packed, src, P, udf, grad3, st);   \
            if (tiles(128) >= 256) return launch_mlp_t<HH, EMAP_PREC_BF16X3, 2, false, 4>(L, packed, src, P, udf, grad3, st); \
            return launch_mlp_t<HH, EMAP_PREC_BF16X3, 1, false, 4>(L, packed, src, P, udf, grad3, st);            \
        }                                                                                                        \
    }
    EMAP_DISPATCH(256)
    EMAP_DISPATCH(128)
#undef EMAP_DISPATCH
    set_error("no MLP kernel for d_hidden=%d prec=%d", L.H, prec);
    return EMAP_E_INVALID;
}

}  // namespace emap
