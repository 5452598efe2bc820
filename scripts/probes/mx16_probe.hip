// mx16_probe.hip - v_mfma_scale_f32_16x16x128_f8f6f4 with e2m3 operands (the MX instruction a 16x16-tile kernel - the training sweep
// udf_mlp_vjp.inc - would use; docs/DESIGN_LOG_r1-r4.md par. 7):  (1) operand / result layout and per-lane E8M0 scales against a host computation,
// (2) issue rate next to v_mfma_f32_16x16x32_f16.
// Hypothesis checked: A lane l = row l % 16, k-block l / 16 (32 consecutive k); B lane l = column l % 16, k-block l / 16; C as every 16x16
// MFMA (lane l: column l % 16, rows 4 (l / 16) + r); scale = byte 0 of the scale register of the lane that holds the block.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/mx16_probe.hip -o scripts/probes/bin/mx16_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void mfma_kernel(const _Float16* ain, const _Float16* bin, const int* sa, const int* sb, float* c) {
    const int l = threadIdx.x;
    v32h xa, xb;
    for (int e = 0; e < 32; ++e) { xa[e] = ain[l * 32 + e]; xb[e] = bin[l * 32 + e]; }
    const v6u qa = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(xa, 1.0f), qb = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(xb, 1.0f);
    v8i a = {(int)qa[0], (int)qa[1], (int)qa[2], (int)qa[3], (int)qa[4], (int)qa[5], 0, 0};
    v8i b = {(int)qb[0], (int)qb[1], (int)qb[2], (int)qb[3], (int)qb[4], (int)qb[5], 0, 0};
    v4f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 2, 2, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 4; ++r) c[l * 4 + r] = acc[r];
}

template <int MODE>   // 0: 16 x f16 16x16x32, 1: 16 x fp6 mx 16x16x128, 2: 8 f16 + 8 fp6 into the same accumulators
__global__ __launch_bounds__(256) void rate_kernel(int iters, float* out, long long* clk) {
    v8h ah, bh;
    for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)(0.001f * (threadIdx.x * 8 + e) - 0.5f); bh[e] = (_Float16)(0.37f - 0.0007f * (threadIdx.x * 8 + e)); }
    v8i a8, b8;
    for (int e = 0; e < 8; ++e) { a8[e] = 0x12345678 * (threadIdx.x + e + 1); b8[e] = 0x9e3779b9 * (threadIdx.x + 3 * e + 1); }
    const int s = 0x7f7f7f7f;
    v4f c[8] = {};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (MODE == 0 || (MODE == 2 && q < 8)) c[q & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c[q & 7], 0, 0, 0);
            else c[q & 7] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c[q & 7], 2, 2, 0, s, 0, s);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int q = 0; q < 8; ++q) r += c[q][0] + c[q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[MODE] = t1 - t0;
}

int main() {
    // (1) layout
    std::vector<_Float16> ha(64 * 32), hb(64 * 32);
    std::vector<int> sa(64), sb(64);
    const float grid[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};   // e2m3-representable magnitudes
    srand(5);
    for (int i = 0; i < 64 * 32; ++i) {
        ha[i] = (_Float16)(grid[rand() % 8] * ((rand() & 1) ? 1.f : -1.f));
        hb[i] = (_Float16)(grid[rand() % 8] * ((rand() & 1) ? 1.f : -1.f));
    }
    for (int l = 0; l < 64; ++l) { sa[l] = 127 + (rand() % 5) - 2; sb[l] = 127 + (rand() % 5) - 2; }
    _Float16 *da, *db; int *dsa, *dsb; float* dc;
    CK(hipMalloc(&da, ha.size() * 2)); CK(hipMalloc(&db, hb.size() * 2)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dc, 64 * 4 * 4));
    CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
    std::vector<float> c(256);
    CK(hipMemcpy(c.data(), dc, 1024, hipMemcpyDeviceToHost));
    double worst = 0, mref = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int col = l % 16, row = 4 * (l / 16) + r;
            double ref = 0;
            for (int kb = 0; kb < 4; ++kb) {
                const int la = kb * 16 + row, lb = kb * 16 + col;
                double sblk = 0;
                for (int e = 0; e < 32; ++e) sblk += (double)(float)ha[la * 32 + e] * (double)(float)hb[lb * 32 + e];
                ref += sblk * ldexp(1.0, sa[la] - 127) * ldexp(1.0, sb[lb] - 127);
            }
            worst = fmax(worst, fabs(ref - c[l * 4 + r])); mref = fmax(mref, fabs(ref));
        }
    printf("mfma_scale_f32_16x16x128 fp6 x fp6: max |C - ref| = %g (max |ref| %g)  [A lane l: row l %% 16, k-block l / 16; B likewise; scale = byte 0]\n", worst, mref);
    // (2) rates
    float* dout; long long* dclk;
    CK(hipMalloc(&dout, 1024 * 256 * 4)); CK(hipMalloc(&dclk, 64));
    const char* names[3] = {"16 x f16 16x16x32", "16 x fp6 mx 16x16x128", "8 f16 + 8 fp6 mx, same accumulators"};
    for (int waves = 1; waves <= 2; ++waves)
        for (int mode = 0; mode < 3; ++mode) {
            const int iters = 20000, grid_ = 256 * waves;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(grid_), dim3(256), 0, 0, iters, dout, dclk);
                else if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(grid_), dim3(256), 0, 0, iters, dout, dclk);
                else hipLaunchKernelGGL(rate_kernel<2>, dim3(grid_), dim3(256), 0, 0, iters, dout, dclk);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            long long clk[3]; CK(hipMemcpy(clk, dclk, 24, hipMemcpyDeviceToHost));
            printf("%d wave(s)/SIMD  %-40s %7.3f ms   %6.1f s_memtime ticks per 16-MFMA group per wave\n", waves, names[mode], ms, (double)clk[mode] / iters);
        }
    return 0;
}
