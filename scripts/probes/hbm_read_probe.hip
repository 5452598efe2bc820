// hbm_read_probe.hip - what a kernel can read from HBM on this box: 1.2 GB streamed once (the size of the weight-gradient GEMM's operand
// set), 16 B per lane, by (a) plain global loads with 8 in flight per lane, (b) LDS-DMA loads (global_load_lds_dwordx4), for several grids.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/hbm_read_probe.hip -o /tmp/hbm_read_probe && /tmp/hbm_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void rd_plain(const f32x4* src, size_t n16, float* sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + 7 * stride < n16; i += 8 * stride) {
        f32x4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load(src + i + q * stride);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = acc[0];
}

__global__ __launch_bounds__(512) void rd_dma(const char* src, size_t bytes, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // 8 waves x DEPTH x 1 KiB
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t chunk = 1024, per_wg = 8 * chunk;
    const size_t nsteps = bytes / ((size_t)gridDim.x * per_wg);
    const char* p = src + ((size_t)blockIdx.x * 8 + wave) * chunk + lane * 16;
    const size_t step = (size_t)gridDim.x * per_wg;
    const unsigned l0 = (unsigned)(size_t)lds + wave * 16 * 1024;
    for (size_t s = 0; s < nsteps; ++s) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(l0 + (unsigned)(s & 15) * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p + s * step), "s"(dst) : "memory");
        if ((s & 15) == 15) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lds[threadIdx.x] == 123 && bytes == 7) sink[1] = 1.f;
}

int main() {
    const size_t bytes = (size_t)1200 << 20;
    char* buf; float* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 64); hipMemset(buf, 1, bytes);
    hipFuncSetAttribute(reinterpret_cast<const void*>(rd_dma), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int grid : {256, 512, 1024, 2048}) {
        for (int kind = 0; kind < 2; ++kind) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(rd_plain, dim3(grid), dim3(512), 0, 0, reinterpret_cast<const f32x4*>(buf), bytes / 16, sink);
                else hipLaunchKernelGGL(rd_dma, dim3(grid > 256 ? 256 : grid), dim3(512), 128 * 1024, 0, buf, bytes, sink);
                hipEventRecord(e1); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%s grid %4d: %.1f us for %.2f GB = %.2f TB/s\n", kind ? "LDS-DMA, 16 x 1 KiB in flight per wave, 1 workgroup per CU" : "plain nontemporal 16 B loads, 8 per lane in flight  ", grid, best * 1e3, bytes / 1e9, bytes / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
