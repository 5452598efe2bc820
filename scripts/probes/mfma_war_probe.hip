// mfma_war_probe.hip - when does an MFMA read its A / B operand registers?  hipcc frees an operand register with the MFMA that reads
// it last and may hand it to the very next VALU instruction (it models no write-after-read hazard on SrcA / SrcB).
// Each wave runs   A, B -> fixed registers;  acc = mfma(A, B, acc);  N wait states;  v_mov junk over the chosen operand's registers
// inside ONE asm block (exact spacing), and the result is compared with the same chain with the overwrite 64 wait states behind the
// MFMA.  Launch "alone" = one wave per SIMD; "crowded" = the SIMD shared with other waves issuing MFMAs (the matrix pipe may still
// be busy with THEIR instruction when this wave's MFMA is issued).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int N> struct Nops;
#define DEF_NOPS(N, S) template <> struct Nops<N> { static constexpr const char* s() { return S; } };
#define CLOB_LIST "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113"

#define LOAD_F16 "v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\tv_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\tv_mov_b32 v106, %7\n\tv_mov_b32 v107, %8\n\ts_nop 4\n\t"
#define MFMA_F16 "v_mfma_f32_32x32x16_f16 %0, v[100:103], v[104:107], %0\n\t"
#define CLOB_A4 "v_mov_b32 v100, %9\n\tv_mov_b32 v101, %9\n\tv_mov_b32 v102, %9\n\tv_mov_b32 v103, %9\n\t"
#define CLOB_B4 "v_mov_b32 v104, %9\n\tv_mov_b32 v105, %9\n\tv_mov_b32 v106, %9\n\tv_mov_b32 v107, %9\n\t"
#define LOAD_MX "v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\tv_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\tv_mov_b32 v108, %7\n\tv_mov_b32 v109, %8\n\tv_mov_b32 v110, %1\n\tv_mov_b32 v111, %3\n\tv_mov_b32 v112, %5\n\tv_mov_b32 v113, %7\n\ts_nop 4\n\t"
#define MFMA_MX "v_mfma_scale_f32_32x32x64_f8f6f4 %0, v[100:105], v[108:113], %0, %10, %10 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
#define CLOB_A6 "v_mov_b32 v100, %9\n\tv_mov_b32 v101, %9\n\tv_mov_b32 v102, %9\n\tv_mov_b32 v103, %9\n\tv_mov_b32 v104, %9\n\tv_mov_b32 v105, %9\n\t"
#define CLOB_B6 "v_mov_b32 v108, %9\n\tv_mov_b32 v109, %9\n\tv_mov_b32 v110, %9\n\tv_mov_b32 v111, %9\n\tv_mov_b32 v112, %9\n\tv_mov_b32 v113, %9\n\t"
#define TAIL "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
#define ASMBLK(LOAD, MFMA, NOPS, CLOB) asm volatile(LOAD MFMA NOPS CLOB TAIL : "+v"(acc) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(junk), "v"(sc) : CLOB_LIST)

#define N0 ""
#define N1 "s_nop 0\n\t"
#define N2 "s_nop 1\n\t"
#define N4 "s_nop 3\n\t"
#define N8 "s_nop 7\n\t"
#define N16 "s_nop 15\n\t"
#define N32 "s_nop 15\n\ts_nop 15\n\t"
#define N64 "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"

template <int MODE, int WHICH, int NOPS>
__global__ __launch_bounds__(256, 2) void probe(int iters, float* out) {
    const unsigned id = (blockIdx.x & 255u) * 256u + threadIdx.x;
    unsigned h = id * 2654435761u + 99u;
    float x = 0.f;
    const unsigned sc = 0x7f7f7f7fu;
    for (int it = 0; it < iters; ++it) {
        unsigned w[8];
        for (int e = 0; e < 8; ++e) { h = h * 1664525u + 1013904223u; w[e] = (MODE == 0) ? ((h & 0x3fff3fffu) | 0x30003000u) : h; }
        v16f acc = {};
        const unsigned junk = (MODE == 0) ? 0x3c003c00u ^ (h & 0x03ff03ffu) : ~h;
#define SEL(NN, TXT) if constexpr (NOPS == NN) { if constexpr (MODE == 0) { if constexpr (WHICH == 0) ASMBLK(LOAD_F16, MFMA_F16, TXT, CLOB_A4); else ASMBLK(LOAD_F16, MFMA_F16, TXT, CLOB_B4); } \
                                                 else { if constexpr (WHICH == 0) ASMBLK(LOAD_MX, MFMA_MX, TXT, CLOB_A6); else ASMBLK(LOAD_MX, MFMA_MX, TXT, CLOB_B6); } }
        SEL(0, N0) SEL(1, N1) SEL(2, N2) SEL(4, N4) SEL(8, N8) SEL(16, N16) SEL(32, N32) SEL(64, N64)
        float t = 0.f;
        for (int r = 0; r < 16; ++r) t += acc[r] * (float)(r + 1);
        x = x * 0.5f + t * 1e-3f;
        h ^= __builtin_bit_cast(unsigned, x);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = x;
}

template <int MODE, int WHICH, int NOPS>
static void one(std::vector<float>& ref_alone, const char* what) {
    const int iters = 1500, nbig = 2048;
    static float *d_a = nullptr, *d_b = nullptr;
    if (!d_a) { CK(hipMalloc(&d_a, 256 * 256 * 4)); CK(hipMalloc(&d_b, (size_t)nbig * 256 * 4)); }
    std::vector<float> ha(256 * 256), hb((size_t)nbig * 256);
    hipLaunchKernelGGL((probe<MODE, WHICH, NOPS>), dim3(256), dim3(256), 0, 0, iters, d_a);
    hipLaunchKernelGGL((probe<MODE, WHICH, NOPS>), dim3(nbig), dim3(256), 0, 0, iters, d_b);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ha.data(), d_a, ha.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), d_b, hb.size() * 4, hipMemcpyDeviceToHost));
    if (NOPS == 64) ref_alone = ha;
    int bad_alone = 0, bad_crowd = 0;
    for (size_t i = 0; i < ha.size(); ++i) if (memcmp(&ha[i], &ref_alone[i], 4)) ++bad_alone;
    for (size_t i = 0; i < hb.size(); ++i) if (memcmp(&hb[i], &ref_alone[i % ref_alone.size()], 4)) ++bad_crowd;
    printf("%-34s overwrite %2d wait states behind the MFMA: wrong results alone %6d / %zu, crowded %7d / %zu\n", what, NOPS, bad_alone, ha.size(), bad_crowd, hb.size());
}
template <int MODE, int WHICH>
static void sweep(const char* what) {
    std::vector<float> ref;
    one<MODE, WHICH, 64>(ref, what);
    one<MODE, WHICH, 32>(ref, what); one<MODE, WHICH, 16>(ref, what); one<MODE, WHICH, 8>(ref, what); one<MODE, WHICH, 4>(ref, what);
    one<MODE, WHICH, 2>(ref, what); one<MODE, WHICH, 1>(ref, what); one<MODE, WHICH, 0>(ref, what);
}
int main() {
    sweep<0, 0>("f16 32x32x16, SrcA overwritten");
    sweep<0, 1>("f16 32x32x16, SrcB overwritten");
    sweep<1, 0>("fp6 mx 32x32x64, SrcA overwritten");
    sweep<1, 1>("fp6 mx 32x32x64, SrcB overwritten");
    return 0;
}
