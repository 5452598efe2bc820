// permlane_swap_probe.hip - what v_permlane32_swap_b32 / v_permlane16_swap_b32 (gfx950) move: lane l starts with a = 1000 + l, b = 2000 + l.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/permlane_swap_probe.hip -o scripts/probes/bin/permlane_swap_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
    const unsigned l = threadIdx.x;
    unsigned a = 1000 + l, b = 2000 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[l] = r[0]; o[64 + l] = r[1]; o[128 + l] = s[0]; o[192 + l] = s[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 1024); unsigned h[256];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane32_swap -> first", "permlane32_swap -> second", "permlane16_swap -> first", "permlane16_swap -> second"};
    for (int q = 0; q < 4; ++q) { printf("%s (lanes 0,15,16,31,32,47,48,63): ", names[q]); int ls[8] = {0,15,16,31,32,47,48,63}; for (int i = 0; i < 8; ++i) printf("%u ", h[64*q + ls[i]]); printf("\n"); }
    return 0;
}
