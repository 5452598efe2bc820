// Probe: (a) do independent VALU ops between MFMAs hide under the matrix pipe, for the 16x16x32 (16-cycle) and the 32x32x16
// (32-cycle) f16 MFMA; (b) do an MFMA-only wave and a VALU-only wave on the SAME SIMD overlap (max) or add (sum)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NV, int BIG>
__device__ __forceinline__ void body(int n, float& r) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x + i); b[i] = (_Float16)(threadIdx.x - i); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
    if (BIG) {
        f16v acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(i * NV + v) & 7]) : "v"(1.0001f), "v"(0.5f));
            }
        }
        for (int i = 0; i < 4; ++i) r += acc[i][0];
    } else {
        f4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(i * NV + v) & 7]) : "v"(1.0001f), "v"(0.5f));
            }
        }
        for (int i = 0; i < 8; ++i) r += acc[i][0];
    }
    for (int i = 0; i < 8; ++i) r += x[i];
}

template <int NV, int BIG, int WAVES>
__global__ __launch_bounds__(WAVES * 256, 1) void k(float* out, int n, long long* cyc) {
    float r = 0;
    long long t0 = clock64();
    body<NV, BIG>(n, r);
    long long t1 = clock64();
    if (r == 123.456f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// roles: waves 0..3 (one per SIMD) MFMA-only (8 per iteration); waves 4..7 VALU-only (NV*8 per iteration)
template <int NV, int MODE>   // MODE 0: both, 1: MFMA wave only, 2: VALU wave only
__global__ __launch_bounds__(512, 1) void roles(float* out, int n, long long* cyc) {
    const int w = threadIdx.x >> 6;
    float r = 0;
    long long t0 = clock64();
    if (w < 4) {
        if (MODE != 2) body<0, 0>(n, r);
    } else {
        if (MODE != 1) {
            float x[8];
            for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
            for (int it = 0; it < n; ++it) {
#pragma unroll
                for (int v = 0; v < NV * 8; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(1.0001f), "v"(0.5f));
            }
            for (int i = 0; i < 8; ++i) r += x[i];
        }
    }
    long long t1 = clock64();
    if (r == 123.456f) out[0] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}

float* out; long long* cyc;
template <int NV, int BIG, int WAVES>
void run() {
    const int n = 2000;
    hipLaunchKernelGGL((k<NV, BIG, WAVES>), dim3(256), dim3(WAVES * 256), 0, 0, out, n, cyc);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf(" %s waves/SIMD %d, VALU per MFMA %d: %6.1f ticks per MFMA(+VALU group)\n", BIG ? "32x32x16" : "16x16x32", WAVES, NV, (double)h / (n * (BIG ? 4 : 8)));
}
template <int NV, int MODE>
void runr() {
    const int n = 2000;
    hipMemset(cyc, 0, 64);
    hipLaunchKernelGGL((roles<NV, MODE>), dim3(256), dim3(512), 0, 0, out, n, cyc);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf(" roles NV=%d mode %d: MFMA wave %6.1f ticks/iter(8 MFMA), VALU wave %6.1f ticks/iter(%d VALU)\n", NV, MODE, (double)h[0] / n, (double)h[4] / n, NV * 8);
}
int main() {
    hipMalloc(&out, 4); hipMalloc(&cyc, 64);
    printf("clock64 ticks (100 MHz constant clock? compare the NV=0 rows: 16 / 32 shader cycles per MFMA)\n");
    run<0, 0, 1>(); run<1, 0, 1>(); run<2, 0, 1>(); run<3, 0, 1>(); run<4, 0, 1>(); run<6, 0, 1>();
    run<0, 1, 1>(); run<2, 1, 1>(); run<4, 1, 1>(); run<6, 1, 1>(); run<8, 1, 1>();
    run<0, 0, 2>(); run<2, 0, 2>(); run<4, 0, 2>();
    run<0, 1, 2>(); run<4, 1, 2>(); run<8, 1, 2>();
    runr<2, 1>(); runr<2, 2>(); runr<2, 0>();
    runr<4, 1>(); runr<4, 2>(); runr<4, 0>();
    return 0;
}
