// mx_stress.hip - is v_mfma_scale_f32_32x32x64_f8f6f4 safe with TWO waves per SIMD issuing it concurrently with different scales?
// Every block computes a chain of MX MFMAs whose operands / scale bytes depend only on (blockIdx.x % 256, threadIdx.x, iteration):
// launch A = 256 blocks (one wave per SIMD), launch B = 1024 blocks (two and more waves per SIMD, blocks b, b+256, .. identical work).
// Any difference between the results of identical work items = an interaction between the waves of a SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MIX>
__global__ __launch_bounds__(256, 2) void stress(int iters, float* out) {
    const unsigned id = (blockIdx.x & 255u) * 256u + threadIdx.x;
    unsigned h = id * 2654435761u + 12345u;
    v8i a, b;
    for (int e = 0; e < 8; ++e) { h = h * 1664525u + 1013904223u; a[e] = (int)h; h = h * 1664525u + 1013904223u; b[e] = (int)h; }
    v8h ah, bh;
    for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)(0.01f * ((id + e) % 37) - 0.2f); bh[e] = (_Float16)(0.3f - 0.01f * ((id * 3 + e) % 41)); }
    v16f c0 = {}, c1 = {};
    unsigned s = h;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const int sa = 120 + ((s >> 8) & 15) + ((int)(s >> 3) & 0x7f00) * 0 , sb = 120 + ((s >> 16) & 15);   // E8M0 bytes 120..135, different per lane and iteration
        const int sa4 = sa | ((sa ^ 5) << 8) | ((sa + 1) << 16) | ((sa - 2) << 24);
        const int sb4 = sb | ((sb + 3) << 8) | ((sb ^ 9) << 16) | ((sb - 1) << 24);
        if constexpr (MIX) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
        }
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 2, 2, 0, sa4, 2, sb4);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, 2, 2, 3, sa4, 1, sb4);
        c0 *= 0.5f; c1 *= 0.5f;                       // keep the magnitudes bounded
        a[it & 7] ^= (int)s;
    }
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += c0[r] - c1[r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MIX>
static void run(const char* name) {
    const int iters = 3000, nbig = 1024;
    float *d_a, *d_b;
    CK(hipMalloc(&d_a, 256 * 256 * 4)); CK(hipMalloc(&d_b, (size_t)nbig * 256 * 4));
    std::vector<float> ha(256 * 256), hb((size_t)nbig * 256);
    int total_bad = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stress<MIX>, dim3(256), dim3(256), 0, 0, iters, d_a);
        hipLaunchKernelGGL(stress<MIX>, dim3(nbig), dim3(256), 0, 0, iters, d_b);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ha.data(), d_a, ha.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), d_b, hb.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0, nan = 0;
        for (size_t i = 0; i < hb.size(); ++i) {
            const float x = hb[i], y = ha[i % ha.size()];
            if (!(x == x)) ++nan;
            if (memcmp(&x, &y, 4) != 0) ++bad;
        }
        printf("%s rep %d: %d of %zu results of the crowded launch differ from the one-wave-per-SIMD launch (%d NaN), sample %g\n", name, rep, bad, hb.size(), nan, ha[1234]);
        total_bad += bad;
    }
    printf("%s: %s\n", name, total_bad ? "WAVES INTERFERE" : "identical");
}
#include <string.h>
int main() { run<0>("mx only"); run<1>("f16 + mx"); return 0; }
