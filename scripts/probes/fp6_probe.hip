// fp6_probe.hip - what the MX (block-scaled) fp6 path of gfx950 does, checked on the GPU before udf_mlp_rev32.inc relies on it:
//   (1) v_cvt_scalef32_pk32_fp6_f16: element order, divide-by-scale, RNE, saturation;
//   (2) v_mfma_scale_f32_32x32x64_f8f6f4 (cbsz = blgp = 2: e2m3 x e2m3): k-slot layout of A / B, per-lane E8M0 scale (byte 0 of the
//       scale VGPR), C layout - against a host computation from the decoded operands;
//   (3) issue rates: f16 32x32x16 vs fp6 / fp8 32x32x64 streams (independent accumulators), and the hh+hh+mx mix of the
//       reverse sweep with the f16 and the mx MFMAs accumulating into the SAME registers vs into separate ones.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/fp6_probe.hip -o scripts/probes/bin/fp6_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void cvt_kernel(const _Float16* in, const float* scale, unsigned* out) {
    const int l = threadIdx.x;
    v32h x;
    for (int e = 0; e < 32; ++e) x[e] = in[l * 32 + e];
    v6u q = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, scale[l]);
    for (int r = 0; r < 6; ++r) out[l * 6 + r] = q[r];
}

__global__ void mfma_kernel(const unsigned* a6, const unsigned* b6, const int* sa, const int* sb, float* c) {
    const int l = threadIdx.x;
    v8i a = {(int)a6[l * 6], (int)a6[l * 6 + 1], (int)a6[l * 6 + 2], (int)a6[l * 6 + 3], (int)a6[l * 6 + 4], (int)a6[l * 6 + 5], 0, 0};
    v8i b = {(int)b6[l * 6], (int)b6[l * 6 + 1], (int)b6[l * 6 + 2], (int)b6[l * 6 + 3], (int)b6[l * 6 + 4], (int)b6[l * 6 + 5], 0, 0};
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 2, 2, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 16; ++r) c[l * 16 + r] = acc[r];
}

// ---- rates ----
template <int MODE>   // 0: f16 32x32x16 x12, 1: fp6 mx x12, 2: fp8 mx x12, 3: mix 8 f16 + 4 fp6 separate accumulators, 4: mix, same accumulators
__global__ __launch_bounds__(256) void rate_kernel(int iters, float* out, long long* clk) {
    v8h ah, bh;
    for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)(0.001f * (threadIdx.x * 8 + e) - 0.5f); bh[e] = (_Float16)(0.37f - 0.0007f * (threadIdx.x * 8 + e)); }
    v8i a8, b8;
    for (int e = 0; e < 8; ++e) { a8[e] = 0x12345678 * (threadIdx.x + e + 1); b8[e] = 0x9e3779b9 * (threadIdx.x + 3 * e + 1); }
    const int s = 0x7f7f7f7f;
    v16f c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
            }
        } else if constexpr (MODE == 1 || MODE == 2) {
            constexpr int F = (MODE == 1) ? 2 : 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, F, F, 0, s, 0, s);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c1, F, F, 0, s, 0, s);
                c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c2, F, F, 0, s, 0, s);
                c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c3, F, F, 0, s, 0, s);
            }
        } else if constexpr (MODE == 3) {   // the reverse sweep's K64 step: 8 f16 MFMAs into (c0, c1), 4 fp6 MFMAs into (c2, c3)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c2, 2, 2, 0, s, 0, s);
                c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c3, 2, 2, 0, s, 0, s);
            }
        } else {                             // the same with ONE accumulator per column tile
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 2, 2, 0, s, 0, s);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c1, 2, 2, 0, s, 0, s);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

static float dec_e2m3(unsigned v) {
    const int s = (v >> 5) & 1, e = (v >> 3) & 3, m = v & 7;
    const float a = (e == 0) ? m * 0.125f : ldexpf(1.0f + m * 0.125f, e - 1);
    return s ? -a : a;
}
static unsigned get6(const unsigned* q, int e) {
    const int bit = 6 * e;
    unsigned long long w = q[bit >> 5];
    if ((bit >> 5) + 1 < 6) w |= (unsigned long long)q[(bit >> 5) + 1] << 32;
    return (unsigned)((w >> (bit & 31)) & 63);
}
static float q_e2m3_host(float v) {
    float a = fabsf(v);
    if (a > 7.5f) a = 7.5f;
    const float step = (a < 2.f) ? 0.125f : (a < 4.f ? 0.25f : 0.5f);
    float q = nearbyintf(a / step) * step;
    if (q > 7.5f) q = 7.5f;
    return v < 0 ? -q : q;
}

int main() {
    // ---------- (1) conversion ----------
    std::vector<_Float16> in(64 * 32);
    std::vector<float> sc(64);
    srand(1);
    for (int l = 0; l < 64; ++l) {
        sc[l] = ldexpf(1.0f, (l % 9) - 4);
        for (int e = 0; e < 32; ++e) {
            float v = ((rand() % 20001) - 10000) * 1e-3f;   // [-10, 10]
            if (l == 0) v = (e - 16) * 0.5f;                 // a ramp: shows the element order directly
            in[l * 32 + e] = (_Float16)(v * sc[l]);
        }
    }
    _Float16* d_in; float* d_sc; unsigned* d_q;
    CK(hipMalloc(&d_in, in.size() * 2)); CK(hipMalloc(&d_sc, 64 * 4)); CK(hipMalloc(&d_q, 64 * 6 * 4));
    CK(hipMemcpy(d_in, in.data(), in.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_sc, sc.data(), 64 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, d_in, d_sc, d_q);
    std::vector<unsigned> q(64 * 6);
    CK(hipMemcpy(q.data(), d_q, q.size() * 4, hipMemcpyDeviceToHost));
    int bad_nat = 0, bad_il = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 32; ++e) {
            const float want = q_e2m3_host((float)in[l * 32 + e] / sc[l]);
            if (dec_e2m3(get6(&q[l * 6], e)) != want) ++bad_nat;
            const int e_il = (e < 16) ? 2 * e : 2 * (e - 16) + 1;
            if (dec_e2m3(get6(&q[l * 6], e_il)) != want) ++bad_il;
        }
    printf("cvt_scalef32_pk32_fp6_f16: mismatches natural order %d / 2048, interleaved order %d / 2048\n", bad_nat, bad_il);
    printf("  lane 0 ramp decoded:");
    for (int e = 0; e < 32; ++e) printf(" %g", dec_e2m3(get6(&q[0], e)));
    printf("\n");

    // ---------- (2) scaled MFMA ----------
    std::vector<unsigned> a6(64 * 6), b6(64 * 6);
    std::vector<int> sa(64), sb(64);
    for (int l = 0; l < 64; ++l) {
        for (int r = 0; r < 6; ++r) { a6[l * 6 + r] = (unsigned)rand() * 2654435761u + rand(); b6[l * 6 + r] = (unsigned)rand() * 40503u + ((unsigned)rand() << 16); }
        sa[l] = 127 + (l % 7) - 3;            // byte 0 = the scale; upper bytes garbage on purpose
        sa[l] |= 0x55aa3300;
        sb[l] = (127 + ((l * 5) % 5) - 2) | 0x11220000;
    }
    unsigned *d_a, *d_b; int *d_sa, *d_sb; float* d_c;
    CK(hipMalloc(&d_a, 64 * 24)); CK(hipMalloc(&d_b, 64 * 24)); CK(hipMalloc(&d_sa, 256)); CK(hipMalloc(&d_sb, 256)); CK(hipMalloc(&d_c, 64 * 64));
    CK(hipMemcpy(d_a, a6.data(), 64 * 24, hipMemcpyHostToDevice)); CK(hipMemcpy(d_b, b6.data(), 64 * 24, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(d_sb, sb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, d_a, d_b, d_sa, d_sb, d_c);
    std::vector<float> c(64 * 16);
    CK(hipMemcpy(c.data(), d_c, c.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, cmax = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            double ref = 0;
            for (int hb = 0; hb < 2; ++hb) {
                const int la = i + 32 * hb, lb = j + 32 * hb;
                double s = 0;
                for (int e = 0; e < 32; ++e) s += (double)dec_e2m3(get6(&a6[la * 6], e)) * dec_e2m3(get6(&b6[lb * 6], e));
                ref += s * ldexp(1.0, ((sa[la] & 255) - 127) + ((sb[lb] & 255) - 127));
            }
            worst = fmax(worst, fabs(ref - c[l * 16 + r]));
            cmax = fmax(cmax, fabs(ref));
        }
    printf("mfma_scale_f32_32x32x64 fp6 x fp6: max |C - ref| = %g (max |ref| %g)  [A lane (hb,i): row i, k-block hb; scale = byte 0]\n", worst, cmax);

    // ---------- (3) rates ----------
    float* d_out; long long* d_clk;
    CK(hipMalloc(&d_out, 1024 * 256 * 4)); CK(hipMalloc(&d_clk, 64));
    const int iters = 20000;
    const char* names[5] = {"12 x f16 32x32x16", "12 x fp6 mx 32x32x64", "12 x fp8 mx 32x32x64", "8 f16 + 4 fp6, separate acc", "8 f16 + 4 fp6, same acc"};
    for (int waves = 1; waves <= 2; ++waves)
        for (int mode = 0; mode < 5; ++mode) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const dim3 grid(256 * waves), blk(256);
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                switch (mode) {
                    case 0: hipLaunchKernelGGL(rate_kernel<0>, grid, blk, 0, 0, iters, d_out, d_clk); break;
                    case 1: hipLaunchKernelGGL(rate_kernel<1>, grid, blk, 0, 0, iters, d_out, d_clk); break;
                    case 2: hipLaunchKernelGGL(rate_kernel<2>, grid, blk, 0, 0, iters, d_out, d_clk); break;
                    case 3: hipLaunchKernelGGL(rate_kernel<3>, grid, blk, 0, 0, iters, d_out, d_clk); break;
                    default: hipLaunchKernelGGL(rate_kernel<4>, grid, blk, 0, 0, iters, d_out, d_clk); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            long long clk; CK(hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost));
            printf("%d wave(s)/SIMD  %-30s %8.3f ms  %6.1f cycles per 12-MFMA group per wave (%.2f GHz)\n", waves, names[mode], ms,
                   (double)clk / iters, clk / (ms * 1e6));
        }
    return 0;
}
