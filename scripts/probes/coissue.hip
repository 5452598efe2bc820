// Probe: do one wave's MFMAs and its SIMD partner's VALU work overlap on gfx950?
// 512-thread workgroups (2 waves per SIMD), 1 per CU.  Waves 0-3 run role A, waves 4-7 role B.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__device__ __forceinline__ float run_mfma(int n, float seed) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][3];
    return r;
}
template <int NCH, int KIND>
__device__ __forceinline__ float run_valu(int n, float seed) {
    float x[NCH];
    for (int i = 0; i < NCH; ++i) x[i] = seed + i;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (KIND == 0) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
            else if (KIND == 1) { _Float16 h = (_Float16)x[i]; x[i] = (x[i] - (float)h) * 2048.0f + 1.0f; }
            else x[i] = __builtin_amdgcn_exp2f(x[i]) * 0.5f;
        }
    }
    float r = 0;
    for (int i = 0; i < NCH; ++i) r += x[i];
    return r;
}
// mode bit0: waves 0-3 do MFMA; bit1: waves 4-7 do VALU; bit2: swap roles (MFMA on the younger half); bit3: setprio on MFMA role
template <int KIND>
__global__ __launch_bounds__(512, 1) void k(float* out, int mode, int nm, int nv, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    const bool first = wave < 4;
    const bool swap = mode & 4;
    const bool do_m = (mode & 1) && (first != swap);
    const bool do_v = (mode & 2) && (first == swap);
    float r = 0;
    long long t0 = clock64();
    if (do_m) {
        if (mode & 8) __builtin_amdgcn_s_setprio(2);
        r = run_mfma<8>(nm, (float)threadIdx.x);
    }
    if (do_v) r = run_valu<4, KIND>(nv, (float)threadIdx.x * 1e-3f);
    long long t1 = clock64();
    if (r == 123.456f) out[0] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
template <int KIND>
void run(const char* name) {
    float* out; long long* cyc; hipMalloc(&out, 4); hipMalloc(&cyc, 64);
    const int nm = 2000, nv = 16000;   // 16000 MFMAs ~ 256k cycles; 64000 VALU ~ 256k+ cycles
    printf("== VALU kind: %s\n", name);
    for (int mode : {1, 2, 3, 7, 11, 15}) {
        hipMemset(cyc, 0, 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, mode, nm, nv, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, mode, nm, nv, cyc);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf(" mode %2d (%s%s%s%s): %8.1f us   wave0 %lld  wave4 %lld ticks\n", mode, (mode & 1) ? "M" : "-", (mode & 2) ? "V" : "-",
               (mode & 4) ? " swapped" : "", (mode & 8) ? " prio" : "", ms * 1e3, h[0], h[4]);
    }
}
int main() {
    run<0>("fma chains (4 independent)");
    run<1>("cvt f16 split");
    run<2>("exp2");
    return 0;
}
