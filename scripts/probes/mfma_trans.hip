// Probe (round 6): does a SIMD's matrix pipe keep its rate while the SIMD's OTHER wave streams transcendental (v_exp_f32) or plain (v_fma_f32) VALU ops?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run 32 x v_mfma_f32_32x32x16_f16 per iteration on 4 independent accumulators, waves 4-7 run 64 VALU
// ops per iteration on 16 independent chains.  MODE 1: MFMA waves only, 2: VALU waves only, 0: both.  clock64 ticks per iteration.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int VKIND, int MODE, int PRIO = 0>   // PRIO: s_setprio of the VALU waves (the MFMA waves stay at 0); VKIND 0: v_exp_f32, 1: v_fma_f32, 2: the epilogue's mix per element (exp, log, rcp + 9 fma)
__global__ __launch_bounds__(512, 1) void k(float* out, int n, long long* cyc) {
    const int w = threadIdx.x >> 6;
    float r = 0;
    float c1 = 1.0001f, c2 = 0.5f;
    asm volatile("" : "+v"(c1), "+v"(c2));
    __syncthreads();
    long long t0 = clock64();
    if (w < 4) {
        if (MODE != 2) {
            h8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x + i); b[i] = (_Float16)(threadIdx.x - i); }
            f16v acc[4];
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
            for (int it = 0; it < n; ++it) {
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) r += acc[i][0];
        }
    } else {
        if (MODE != 1) {
            float x[16], y[16];
            for (int i = 0; i < 16; ++i) { x[i] = 1.0f + threadIdx.x * 1e-3f + i; y[i] = threadIdx.x * 1e-3f + i; }
            if (PRIO == 3) asm volatile("s_setprio 3");
            if (PRIO == 1) asm volatile("s_setprio 1");
            for (int it = 0; it < n; ++it) {
#pragma unroll
                for (int i = 0; i < 64; ++i) {
                    if (VKIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i & 15]));
                    else if (VKIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[i & 15]) : "v"(c1), "v"(c2));
                    else {
                        if ((i & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i & 15]));
                        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[i & 15]) : "v"(c1), "v"(c2));
                    }
                }
            }
            for (int i = 0; i < 16; ++i) r += x[i] + y[i];
        }
    }
    long long t1 = clock64();
    if (r == 123.456f) out[0] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}

template <int VKIND, int MODE, int PRIO = 0>
void run(const char* name, float* out, long long* cyc) {
    const int n = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<VKIND, MODE, PRIO>), dim3(256), dim3(512), 0, 0, out, n, cyc); (void)hipDeviceSynchronize(); }
    long long h[8];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-60s MFMA wave %7.1f ticks / 32 MFMA (32 x 32 = 1024 cycles)   VALU wave %7.1f ticks / 64 ops\n", name, (double)h[0] / n, (double)h[4] / n);
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 64);
    run<0, 1>("MFMA waves alone", out, cyc);
    run<0, 2>("v_exp waves alone", out, cyc);
    run<0, 0>("MFMA waves + v_exp waves", out, cyc);
    run<1, 2>("v_fma waves alone", out, cyc);
    run<1, 0>("MFMA waves + v_fma waves", out, cyc);
    run<2, 2>("(exp, 3 fma) waves alone", out, cyc);
    run<2, 0>("MFMA waves + (exp, 3 fma) waves", out, cyc);
    run<0, 0, 3>("MFMA waves + v_exp waves at s_setprio 3", out, cyc);
    run<1, 0, 3>("MFMA waves + v_fma waves at s_setprio 3", out, cyc);
    run<2, 0, 3>("MFMA waves + (exp, 3 fma) waves at s_setprio 3", out, cyc);
    run<2, 0, 1>("MFMA waves + (exp, 3 fma) waves at s_setprio 1", out, cyc);
    return 0;
}
