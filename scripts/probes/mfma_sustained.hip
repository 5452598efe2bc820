// Probe: sustained dense f16 MFMA rate (v_mfma_f32_16x16x32_f16, independent accumulators) over ~1 s of back-to-back launches,
// by HIP events - what the matrix pipes deliver once the power management has settled, vs the first launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int WAVES>
__global__ __launch_bounds__(WAVES * 256, 1) void k(float* out, int n) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(threadIdx.x * 0.002f - i); }
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0];
    if (r == 123.456f) out[0] = r;
}
template <int WAVES>
void run(float* out) {
    const int n = 20000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = (double)blocks * WAVES * 4 * n * 8 * 16384.0;
    for (int rep = 0; rep < 2; ++rep) {
        const int launches = rep == 0 ? 1 : 60;
        hipEventRecord(e0, 0);
        for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k<WAVES>), dim3(blocks), dim3(WAVES * 256), 0, 0, out, n);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf(" waves/SIMD %d, %2d launch(es): %.1f ms, %.0f TFLOP/s, %.2f ns per MFMA per SIMD\n", WAVES, launches, ms, flop * launches / ms * 1e-9,
               ms * 1e6 / ((double)launches * n * 8 * WAVES));
    }
}
int main() {
    float* out; hipMalloc(&out, 4);
    run<1>(out); run<2>(out); run<1>(out);
    return 0;
}
