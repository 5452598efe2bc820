// mfma_raw_stress.hip - is the compiler's wait between an MFMA and the first VALU read of its result enough when OTHER waves keep the
// same SIMD's matrix pipe busy?  Every wave runs  acc = mfma(a, b, acc); x += acc[r] (VALU read right behind it)  chains whose result
// depends only on (blockIdx.x % 256, threadIdx.x); launch A = one wave per SIMD, launch B = two and more waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0: f16 32x32x16, 1: fp6 mx 32x32x64
__global__ __launch_bounds__(256, 2) void stress(int iters, float* out) {
    const unsigned id = (blockIdx.x & 255u) * 256u + threadIdx.x;
    unsigned h = id * 2654435761u + 777u;
    v8i a, b;
    for (int e = 0; e < 8; ++e) { h = h * 1664525u + 1013904223u; a[e] = (int)h; h = h * 1664525u + 1013904223u; b[e] = (int)h; }
    v8h ah, bh;
    for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)(0.01f * ((id + e) % 37) - 0.2f); bh[e] = (_Float16)(0.3f - 0.01f * ((id * 3 + e) % 41)); }
    float x = 0.f;
    const int s = 0x7f7f7f7f;
    for (int it = 0; it < iters; ++it) {
        v16f c0 = {}, c1 = {};
        for (int r = 0; r < 16; ++r) { c0[r] = x * 1e-3f + r; c1[r] = r - x * 1e-3f; }
        if constexpr (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, c1, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 2, 2, 0, s, 0, s);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, 2, 2, 0, s, 0, s);
        }
        // VALU reads right behind the MFMAs (the compiler pads the documented wait states, nothing more)
        float t = 0.f;
        for (int r = 0; r < 16; ++r) t += c1[r] * (r + 1) - c0[15 - r];
        x = t * 1e-3f;
        ah[it & 7] = (_Float16)(x * 1e-3f);
        a[it & 7] ^= __builtin_bit_cast(int, x);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = x;
}

template <int MODE>
static void run(const char* name) {
    const int iters = 4000, nbig = 2048;
    float *d_a, *d_b;
    CK(hipMalloc(&d_a, 256 * 256 * 4)); CK(hipMalloc(&d_b, (size_t)nbig * 256 * 4));
    std::vector<float> ha(256 * 256), hb((size_t)nbig * 256);
    int total = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stress<MODE>, dim3(256), dim3(256), 0, 0, iters, d_a);
        hipLaunchKernelGGL(stress<MODE>, dim3(nbig), dim3(256), 0, 0, iters, d_b);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ha.data(), d_a, ha.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), d_b, hb.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (size_t i = 0; i < hb.size(); ++i) if (memcmp(&hb[i], &ha[i % ha.size()], 4) != 0) ++bad;
        printf("%s rep %d: %d of %zu differ (sample %g)\n", name, rep, bad, hb.size(), ha[4321]);
        total += bad;
    }
    printf("%s: %s\n", name, total ? "RESULT READ TOO EARLY UNDER CONTENTION" : "identical");
}
int main() { run<0>("f16 32x32x16 -> VALU"); run<1>("fp6 mx 32x32x64 -> VALU"); return 0; }
