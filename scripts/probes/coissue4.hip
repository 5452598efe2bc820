// Probe for the "one wave per SIMD, statically scheduled" candidate (docs/DESIGN_LOG_r1-r4.md par. 7): a K-step of the reverse-sweep kernel
// at a 128-point tile is 12 v_mfma_f32_32x32x16_f16 + 4 fragment loads (1 KiB each, L2-resident) + 16 ds_read_b128 + a share of
// the epilogue VALU (NV per MFMA).  Everything but the MFMAs is placed between them.  How many cycles per K-step does ONE wave
// per SIMD need, against 12 x 32 = 384?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int NV, int LOADS, int DSR>
__global__ __launch_bounds__(256, 1) void k(const char* w, float* out, int n, long long* cyc) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<float*>(lds)[i] = i * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(lane * 0.02f - i); }
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = lane * 1e-3f + i;
    u4 ld[4] = {};
    u4 dr[2] = {};
    const char* base = w + ((blockIdx.x * 4 + wave) & 63) * 65536;   // 4 MiB of "weights": L2 / MALL resident
    const int voff = lane * 16;
    long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
        const unsigned long long pv = (unsigned long long)(base + (it & 15) * 4096);
        const char* p = reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pv >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)pv));
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (LOADS && (m % 3) == 0)
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%c3" : "=v"(ld[m / 3]) : "v"(voff), "s"(p), "i"((m / 3) * 1024) : "memory");
            if (DSR) {
#pragma unroll
                for (int d = 0; d < DSR; ++d)
                    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(dr[d & 1]) : "v"(voff), "i"(((m * 2 + d) & 7) * 1024) : "memory");
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (v == NV - 1 && NV >= 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(m * NV + v) & 7]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(m * NV + v) & 7]) : "v"(1.0001f), "v"(0.5f));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]), "+v"(dr[0]), "+v"(dr[1]));
        a[0] = (_Float16)((float)a[0] + __builtin_bit_cast(float, ld[0][0] & 1u) + __builtin_bit_cast(float, dr[0][0] & 1u));
    }
    long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < 4; ++i) r += acc[i][0];
    for (int i = 0; i < 8; ++i) r += x[i];
    if (r == 123.456f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
char* w; float* out; long long* cyc;
template <int NV, int LOADS, int DSR>
void run() {
    const int n = 2000;
    hipLaunchKernelGGL((k<NV, LOADS, DSR>), dim3(256), dim3(256), 0, 0, w, out, n, cyc);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf(" per K-step (12 MFMA 32x32x16 = 384 cycles): VALU/MFMA %d, global loads %d, ds_read_b128/MFMA %d: %7.1f ticks\n", NV, LOADS ? 4 : 0, DSR, (double)h / n);
}
int main() {
    hipMalloc(&w, 4 << 20); hipMemset(w, 0, 4 << 20); hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    run<0, 0, 0>(); run<0, 0, 0>();
    run<3, 0, 0>(); run<0, 1, 0>(); run<0, 0, 2>();
    run<3, 1, 0>(); run<3, 1, 2>(); run<4, 1, 2>(); run<2, 1, 2>(); run<5, 1, 2>();
    return 0;
}
