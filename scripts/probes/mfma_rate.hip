// Probe: back-to-back issue rate of v_mfma_f32_16x16x32_{f16,bf16} on one wave / two waves per SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int KIND, int WAVES>
__global__ __launch_bounds__(WAVES * 256, 1) void k(float* out, int n, long long* cyc) {
    h8 a, b; b8 c, d;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x + i); b[i] = (_Float16)(threadIdx.x - i); c[i] = (__bf16)(float)(threadIdx.x + i); d[i] = (__bf16)(float)(threadIdx.x - i); }
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c, d, acc[i], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0];
    if (r == 123.456f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND, int WAVES> void run(const char* nm) {
    float* out; long long* cyc; hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((k<KIND, WAVES>), dim3(256), dim3(WAVES * 256), 0, 0, out, 4000, cyc);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s, %d wave(s)/SIMD: %.1f ticks per MFMA (per wave)\n", nm, WAVES, (double)h / (4000 * 8));
}
int main() { run<0, 1>("f16 16x16x32"); run<1, 1>("bf16 16x16x32"); run<0, 2>("f16 16x16x32"); run<1, 2>("bf16 16x16x32"); return 0; }
