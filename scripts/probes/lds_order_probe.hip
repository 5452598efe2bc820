// lds_order_probe.hip - can an LDS read that is still QUEUED when its wave reaches s_barrier be overtaken by LDS writes that other
// waves of the workgroup issue behind the barrier?  (hipcc's __syncthreads() does not wait for outstanding LDS reads: its LDS-only
// release fence assumes that the LDS operations of all waves execute in one total order.)
// Wave 0 of a 4-wave workgroup issues NR x ds_read_b128 over a region holding pattern A and goes to s_barrier WITHOUT waiting; waves
// 1..3 wait at the barrier and then overwrite the region with pattern B (ds_write_b32, 32 lanes, like the reduction of
// udf_mlp_rev32_kernel).  Wave 0 then waits for its reads and counts dwords that show pattern B.  Run with one and with two
// workgroups per CU (the second one adds LDS traffic).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int NR = 12;                 // outstanding 16 B/lane reads of wave 0 at the barrier (12 KiB region)

template <bool DRAIN>
__global__ __launch_bounds__(256, 2) void probe(int iters, unsigned* bad_out, unsigned* lane_hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];      // 80 KiB per workgroup: two workgroups fill a CU's LDS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = tid; i < NR * 256; i += 256) lds[i] = 0xA0000000u | (unsigned)i;       // pattern A
        __syncthreads();
        if (wave == 0) {
            u32x4 r[NR];
            const unsigned addr = lane * 16;
#pragma unroll
            for (int q = 0; q < NR; ++q) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(r[q]) : "v"(addr), "i"(q * 1024));
            if (DRAIN) asm volatile("s_waitcnt lgkmcnt(0)");
            asm volatile("s_barrier" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                         "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]));
#pragma unroll
            for (int q = 0; q < NR; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((r[q][e] >> 28) != 0xAu) { ++bad; atomicAdd(&lane_hist[lane], 1u); }
        } else {
            asm volatile("s_barrier" ::: "memory");
            // behind the barrier: overwrite the region, 32 lanes x 4 B per instruction
            if (lane < 32)
                for (int i = (wave - 1) * 32 + lane; i < NR * 256; i += 96) lds[i] = 0xB0000000u | (unsigned)i;
        }
        __syncthreads();
        // LDS noise for the other workgroup of the CU
        unsigned s = 0;
        for (int i = tid; i < 16384; i += 256) s += lds[NR * 256 + i];
        if (s == 0x12345u) bad_out[1] = s;
        __syncthreads();
    }
    if (bad) atomicAdd(&bad_out[0], bad);
}

template <bool DRAIN>
static void run(const char* name, int blocks) {
    unsigned *d_bad, *d_hist;
    CK(hipMalloc(&d_bad, 8)); CK(hipMalloc(&d_hist, 256));
    CK(hipMemset(d_bad, 0, 8)); CK(hipMemset(d_hist, 0, 256));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<DRAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    hipLaunchKernelGGL(probe<DRAIN>, dim3(blocks), dim3(256), 80 * 1024, 0, 2000, d_bad, d_hist);
    CK(hipDeviceSynchronize());
    unsigned bad[2], hist[64];
    CK(hipMemcpy(bad, d_bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hist, d_hist, 256, hipMemcpyDeviceToHost));
    printf("%-52s %4d workgroups: %u overtaken dwords of %llu read", name, blocks, bad[0], 2000ull * blocks * NR * 256);
    if (bad[0]) { printf("; by lane quarter:"); for (int q = 0; q < 4; ++q) { unsigned t = 0; for (int l = 16 * q; l < 16 * q + 16; ++l) t += hist[l]; printf(" %u", t); } }
    printf("\n");
}
int main() {
    run<false>("reads queued at s_barrier, no wait", 256);
    run<false>("reads queued at s_barrier, no wait", 512);
    run<false>("reads queued at s_barrier, no wait", 2048);
    run<true>("s_waitcnt lgkmcnt(0) in front of s_barrier", 512);
    run<true>("s_waitcnt lgkmcnt(0) in front of s_barrier", 2048);
    return 0;
}
