// Probe: what do HW_REG_HW_ID / HW_REG_LDS_ALLOC / XCC_ID hold for co-resident workgroups on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(256, 2) void probe(uint32_t* out, int spin) {
    extern __shared__ char smem[];
    smem[threadIdx.x] = 1;
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const uint32_t la = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);
    const uint32_t xc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if ((threadIdx.x & 63) == 0) {
        uint32_t* o = out + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        o[0] = hw; o[1] = la; o[2] = xc; o[3] = (uint32_t)(t0 & 0xffffffff);
    }
}
int main() {
    const int nb = 1024;
    uint32_t* d; hipMalloc(&d, nb * 16 * 4);
    for (int lds : {80 * 1024, 53 * 1024, 20 * 1024}) {
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(probe, dim3(nb), dim3(256), lds, 0, d, 200000);
        hipDeviceSynchronize();
        static uint32_t h[nb * 16];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("lds=%d\n", lds);
        for (int b = 0; b < 24; ++b) {
            printf(" wg%3d:", b);
            for (int w = 0; w < 4; ++w) printf("  hw=%08x la=%08x xcc=%x", h[(b * 4 + w) * 4], h[(b * 4 + w) * 4 + 1], h[(b * 4 + w) * 4 + 2]);
            printf("\n");
        }
        // histogram of la values
        int nz = 0; for (int b = 0; b < nb; ++b) if ((h[b * 16 + 1] & 0xfff) != 0) ++nz;
        printf(" workgroups with nonzero low-12 bits of LDS_ALLOC: %d / %d\n", nz, nb);
    }
    return 0;
}
