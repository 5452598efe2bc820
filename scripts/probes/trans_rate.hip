// Probe (round 6): issue cost of the transcendental VALU ops (v_exp_f32) on gfx950 and whether plain VALU ops (v_fma_f32) of the SAME wave or of the SIMD's
// other wave issue in their shadow.  One workgroup per CU, WAVES = 4 (one wave per SIMD) or 8 (two per SIMD); each wave runs n iterations of a straight-line
// block; 16 independent chains per op kind (no dependency stalls).  clock64 ticks per iteration of wave 0 (and wave 4).
#include <hip/hip_runtime.h>
#include <stdio.h>

#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(i) & 15]))
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[(i) & 15]) : "v"(c1), "v"(c2))

template <int KIND>
__global__ __launch_bounds__(512, 1) void k(float* out, int n, long long* cyc) {
    float x[16], y[16];
    for (int i = 0; i < 16; ++i) { x[i] = 1.0f + threadIdx.x * 1e-3f + i; y[i] = threadIdx.x * 1e-3f + i; }
    float c1 = 1.0001f, c2 = 0.5f;
    asm volatile("" : "+v"(c1), "+v"(c2));
    const int w = threadIdx.x >> 6;
    __syncthreads();
    long long t0 = clock64();
    if (KIND == 0) {            // 32 exp
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) EXP(i);
        }
    } else if (KIND == 1) {     // 96 fma
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 96; ++i) FMA(i);
        }
    } else if (KIND == 2) {     // 32 exp then 96 fma
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) EXP(i);
#pragma unroll
            for (int i = 0; i < 96; ++i) FMA(i);
        }
    } else if (KIND == 3) {     // 32 x (exp, 3 fma)
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) { EXP(i); FMA(3 * i); FMA(3 * i + 1); FMA(3 * i + 2); }
        }
    } else if (KIND == 4) {     // 32 x (exp, 1 fma)
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) { EXP(i); FMA(i); }
        }
    } else if (KIND == 5) {     // 32 fma
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) FMA(i);
        }
    } else if (KIND == 6) {     // roles: waves 0-3 32 exp, waves 4-7 96 fma
        if (w < 4) {
            for (int it = 0; it < n; ++it) {
#pragma unroll
                for (int i = 0; i < 32; ++i) EXP(i);
            }
        } else {
            for (int it = 0; it < n; ++it) {
#pragma unroll
                for (int i = 0; i < 96; ++i) FMA(i);
            }
        }
    }
    long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += x[i] + y[i];
    if (r == 123.456f) out[0] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}

template <int KIND>
void run(const char* name, int waves, float* out, long long* cyc) {
    const int n = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(waves * 64), 0, 0, out, n, cyc);
        (void)hipDeviceSynchronize();
    }
    long long h[8];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s waves/CU %d: %7.1f ticks/iter (wave 0)", name, waves, (double)h[0] / n);
    if (waves == 8) printf("  %7.1f (wave 4)", (double)h[4] / n);
    printf("\n");
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 64);
    for (int waves = 4; waves <= 8; waves += 4) {
        run<0>("32 v_exp", waves, out, cyc);
        run<5>("32 v_fma", waves, out, cyc);
        run<1>("96 v_fma", waves, out, cyc);
        run<2>("32 v_exp then 96 v_fma", waves, out, cyc);
        run<3>("32 x (v_exp, 3 v_fma)", waves, out, cyc);
        run<4>("32 x (v_exp, v_fma)", waves, out, cyc);
    }
    run<6>("roles: waves 0-3 32 v_exp | waves 4-7 96 v_fma", 8, out, cyc);
    return 0;
}
