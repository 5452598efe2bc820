#!/usr/bin/env python3
"""Adds s_memtime stamps (EMAP_TIMELINE) to emap_amd/csrc/udf_mlp_vjp.inc + the export emap_debug_vjp_timeline to udf_mlp_f16x3.hip, in place.
   usage: python scripts/probes/vjp_timeline_instrument.py; scripts/build_variant.sh vtl -DEMAP_TIMELINE; git checkout emap_amd/csrc
   (anchored string replacements instead of a patch file: they survive edits elsewhere in the kernel; reader: scripts/probes/vjp_timeline.py)"""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = os.path.join(ROOT, "emap_amd/csrc/udf_mlp_vjp.inc"); s = open(p).read()
def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:80])
    s = s.replace(old, new, 1)
T = lambda body: "#ifdef EMAP_TIMELINE\n" + body + "\n#endif\n"
rep('struct VjpArgs {', T('''static __device__ long long emap_vtl_buf[32 * 8 * 64];
#define VTL(i) do { if (vtl_on) { const long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) vtlp[i] = t_; } } while (0)
#define VTLIF(c, i) do { if (vtl_on && (c)) { const long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) vtlp[i] = t_; } } while (0)
#else
#define VTL(i) do {} while (0)
#define VTLIF(c, i) do {} while (0)''') + '\nstruct VjpArgs {')
rep('        const long long pbase = ((long long)va.tile0 + tile) * VJP_PT;\n', '        const long long pbase = ((long long)va.tile0 + tile) * VJP_PT;\n' + T('''        const bool vtl_on = (tile == (int)(blockIdx.x + gridDim.x)) && blockIdx.x < 32;
        long long* const vtlp = emap_vtl_buf + (blockIdx.x * NW + wave) * 64;
        int vtb = -1;''') + '        VTL(16);\n')
rep('        auto publish = [&](int n_pairs) __attribute__((always_inline)) {\n', '        auto publish = [&](int n_pairs) __attribute__((always_inline)) {\n' + T('            VTLIF(vtb >= 0, vtb + 3);'))
rep('            if constexpr (SMX) {\n                // MX B operands: lane (g, j) converts column j of column tile g', T('            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");\n            VTLIF(vtb >= 0, vtb + 4);') + '            if constexpr (SMX) {\n                // MX B operands: lane (g, j) converts column j of column tile g')
rep("            __syncthreads();   // barrier B: the next GEMM's input is complete (PP: and everyone has finished reading the other buffer)\n", T('            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");\n            VTLIF(vtb >= 0, vtb + 5);') + "            __syncthreads();   // barrier B: the next GEMM's input is complete (PP: and everyone has finished reading the other buffer)\n" + T('            VTLIF(vtb >= 0, vtb + 6);'))
rep('            const int kind = (L.h_ks == 0) ? KIND_FIRST : (L.pe_ks ? KIND_SKIP : KIND_NORMAL);\n            static_for<PPW>', '            const int kind = (L.h_ks == 0) ? KIND_FIRST : (L.pe_ks ? KIND_SKIP : KIND_NORMAL);\n' + T('            vtb = (l == 2) ? 0 : -1;\n            VTLIF(vtb >= 0, 0);') + '            static_for<PPW>')
rep('                if constexpr (ROLLV) head_after_fwd(l);\n                uint32_t bk[16];', T('                VTLIF(vtb >= 0, 1);') + '                if constexpr (ROLLV) head_after_fwd(l);\n                uint32_t bk[16];')
rep('                {\n                    const V8 xh[4] = {o[pi][0][0], o[pi][1][0], o[pi][2][0], o[pi][3][0]};\n                    stash_store(sa + ', T('                VTLIF(vtb >= 0, 2);') + '                {\n                    const V8 xh[4] = {o[pi][0][0], o[pi][1][0], o[pi][2][0], o[pi][3][0]};\n                    stash_store(sa + ')
rep("            const int lz = b - 1;                         // the layer whose zb, zb' this step produces\n", "            const int lz = b - 1;                         // the layer whose zb, zb' this step produces\n" + T('            vtb = (b == 3) ? 8 : -1;\n            VTLIF(vtb >= 0, 8);'))
rep('                wait_slab();\n', T('                VTLIF(vtb >= 0, 9);') + '                wait_slab();\n')
rep('                {\n                    const V8 xh[4] = {o[pi][0][0], o[pi][1][0], o[pi][2][0], o[pi][3][0]};\n                    stash_store(sz + ', T('                VTLIF(vtb >= 0, 10);') + '                {\n                    const V8 xh[4] = {o[pi][0][0], o[pi][1][0], o[pi][2][0], o[pi][3][0]};\n                    stash_store(sz + ')
rep("        __syncthreads();   // the tile is done with xbuf / pebuf / red before the next tile's PE overwrites them\n", "        VTL(19);\n        __syncthreads();   // the tile is done with xbuf / pebuf / red before the next tile's PE overwrites them\n")
rep('        // ================= last layer: one real output row, split along K over the waves =================\n', T('        vtb = -1;') + '        VTL(17);\n        // ================= last layer: one real output row, split along K over the waves =================\n')
rep('        // ================= reverse sweep: adjoints of the pre-activations', '        VTL(18);\n        // ================= reverse sweep: adjoints of the pre-activations')
open(p, "w").write(s)
q = os.path.join(ROOT, "emap_amd/csrc/udf_mlp_f16x3.hip")
open(q, "a").write('''#ifdef EMAP_TIMELINE
extern "C" int emap_debug_vjp_timeline(long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(emap::emap_vtl_buf), (size_t)n * sizeof(long long));
}
#endif
''')
print("instrumented", p)
