// clock_probe.hip - what does s_memtime count?  Ratio of s_memtime to s_memrealtime (constant 100 MHz) inside (a) a light kernel
// (one wave per CU, dependent v_fma chain), (b) a dense f16 MFMA loop on every SIMD (2 waves per SIMD), (c) MFMA + LDS reads + VALU, (d) the dense MFMA loop on operands with random mantissas.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(int mode, int iters, long long* out, float* sink) {
    __shared__ float lds[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc0 = {0}, acc1 = {0};
    f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    f16x8 ra[8], rb[8];     // mode 3: operands with random mantissas, a different pair for each of 8 consecutive MFMAs
    {
        unsigned sd = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
        for (int q = 0; q < 8; ++q)
            for (int e = 0; e < 8; ++e) {
                sd = sd * 1664525u + 1013904223u; ra[q][e] = (_Float16)(((int)(sd >> 8) % 2001 - 1000) * 1e-3f);
                sd = sd * 1664525u + 1013904223u; rb[q][e] = (_Float16)(((int)(sd >> 8) % 2001 - 1000) * 1e-3f);
            }
    }
    float x = (float)lane, y = 1.0f;
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
#pragma unroll
            for (int q = 0; q < 32; ++q) x = fmaf(x, 1.0001f, 0.5f);
        } else if (mode == 3) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra[q], rb[q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra[(q + 3) & 7], rb[(q + 5) & 7], acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                if (mode == 2) {
                    y = fmaf(y, 1.0001f, lds[(lane * 4 + q * 64 + it) & 8191]);
                    x = fmaf(x, y, 0.25f); x = fmaf(x, 0.999f, y); x = __expf(x * 1e-6f) + x;
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { out[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)] = t1 - t0; out[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) + 1] = r1 - r0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x + y + acc0[0] + acc1[3];
}

int main() {
    long long* out; float* sink;
    hipMalloc(&out, 256 * 8 * 2 * sizeof(long long)); hipMalloc(&sink, 256 * 512 * sizeof(float));
    struct { const char* name; int mode, threads, iters; } cases[] = {
        {"light: 1 wave per CU, v_fma chain", 0, 64, 40000}, {"dense MFMA, 2 waves per SIMD", 1, 512, 20000}, {"MFMA + LDS + VALU, 2 waves per SIMD", 2, 512, 12000},
        {"dense MFMA, RANDOM operands, 2 waves/SIMD", 3, 512, 20000}, {"dense MFMA, constant operands (again)", 1, 512, 20000}};
    for (auto& c : cases) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(c.threads), 0, 0, c.mode, c.iters, out, sink);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(256 * 8 * 2);
            hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
            const int nw = 256 * c.threads / 64;
            std::vector<double> ratio;
            for (int w = 0; w < nw; ++w) ratio.push_back((double)h[2 * w] / (double)h[2 * w + 1]);
            std::sort(ratio.begin(), ratio.end());
            const double mf = (c.mode ? 16.0 * c.iters * 2 * 32 * 32 * 16 * (double)nw / (ms * 1e-3) / 1e12 : 0.0);
            printf("%-40s rep %d: kernel %.3f ms, s_memtime/s_memrealtime median %.3f (min %.3f max %.3f) -> %.0f MHz if s_memrealtime = 100 MHz; %.0f TFLOP/s\n",
                   c.name, rep, ms, ratio[ratio.size() / 2], ratio.front(), ratio.back(), ratio[ratio.size() / 2] * 100.0, mf);
        }
    }
    return 0;
}
