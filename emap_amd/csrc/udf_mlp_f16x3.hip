// udf_mlp_f16x3.hip - instantiates the fused UDF-MLP kernels (udf_mlp_kernel.inc) for EMAP_PREC_F16X3.
#include "udf_mlp_kernel.inc"
namespace emap {
int launch_mlp_f16x3(const NetLayout& L, const void* packed, const PointSource& src, int64_t P, float* udf, float* grad3,
                      hipStream_t st, int variant, int32_t* err, void* scratch, const CompositeFuse* fuse) {
    if (variant == 3) return launch_mlp_rev32_mode<EMAP_PREC_F16X3>(L, packed, src, P, udf, grad3, st, err, scratch, fuse);
    return launch_mlp_fs2_mode<EMAP_PREC_F16X3>(L, packed, src, P, udf, grad3, st, err);
}
int launch_vjp_sweep_f16x3(const NetLayout& L, const void* packed, const PointSource& src, int64_t P, int tile0, int n_tiles,
                            const float* d_udf, const float* d_grad, const VjpLayout& V, char* stash_a, char* stash_z, char* stash_s,
                            int grid, const uint32_t* absmax, float* ldot, hipStream_t st, int32_t* err) {
    return launch_vjp_sweep_mode<EMAP_PREC_F16X3>(L, packed, src, P, tile0, n_tiles, d_udf, d_grad, V, stash_a, stash_z, stash_s, grid,
                                                 absmax, ldot, st, err);
}
int launch_is_f16x3(const NetLayout& L, const void* packed, const IsLaunch& q, hipStream_t st, int32_t* err) {
    return launch_is_mode<EMAP_PREC_F16X3>(L, packed, q, st, err);
}
}  // namespace emap
#ifdef EMAP_TIMELINE      // probe builds only
extern "C" int emap_debug_fs2_timeline(long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(emap::emap_ftl_buf), (size_t)n * sizeof(long long));
}
#endif
