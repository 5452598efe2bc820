// Probe: inside ONE wave, do independent VALU ops placed between MFMAs hide under the MFMAs' 16 pipe cycles (gfx950)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NV, int WAVES>
__global__ __launch_bounds__(WAVES * 256, 1) void k(float* out, int n, long long* cyc) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x + i); b[i] = (_Float16)(threadIdx.x - i); }
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
    long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(i * NV + v) & 7]) : "v"(1.0001f), "v"(0.5f));
        }
    }
    long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + x[i];
    if (r == 123.456f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NV, int WAVES>
void run() {
    float* out; long long* cyc; hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    const int n = 2000;
    hipLaunchKernelGGL((k<NV, WAVES>), dim3(256), dim3(WAVES * 256), 0, 0, out, n, cyc);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf(" waves/SIMD %d, VALU per MFMA %d: %6.1f ticks per MFMA(+VALU group)\n", WAVES, NV, (double)h / (n * 8));
}
int main() {
    run<0, 1>(); run<1, 1>(); run<2, 1>(); run<3, 1>(); run<4, 1>(); run<6, 1>();
    run<0, 2>(); run<1, 2>(); run<2, 2>(); run<3, 2>(); run<4, 2>();
    return 0;
}
