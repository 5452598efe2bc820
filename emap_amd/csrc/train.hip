// train.hip - the scalar tail of a training step (SURVEY.md par. 8 a15): what runner_udf.py:124-168 does between render() and
// loss.backward() and after it, as three small kernels instead of ~25 torch element-wise launches (each ~4.7 us of launch
// latency on a 512-ray step) and a multi-tensor Adam that runs two 463 k-element tensors on 8 workgroups (2 x 25 us):
//   train_stats_kernel : EdgeLoss (loss.py:14-17, mse) numerator and dL/d(edge) of the rank's rays + the eikonal sums render()
//                        left in `scalars` -> the 5 numbers a data-parallel step exchanges
//   train_loss_kernel  : loss = edge_loss * edge_weight + igr_weight * ge + igr_ns_weight * ge_ns   (runner_udf.py:155-159) from
//                        the (global) statistics
//   adam_kernel        : torch.optim.Adam with the reference's two parameter groups (runner_base.py:110-117) on the flat
//                        parameter / gradient buffers; the step counter is a device word (graph-capturable)
#include "emap_common.h"

namespace emap {

__global__ __launch_bounds__(256) void train_stats_kernel(const float* edge, const float* true_edge, const float* scalars, int N,
                                                          float d_scale, float* d_edge, float* stats) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float d = edge[i] - true_edge[i];
        if (d_edge) d_edge[i] = d * d_scale;
        s += (double)d * (double)d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        // scalars[3..6] = sum(relax*err), sum(relax), sum(near*err), sum(near)   (emap_composite_fwd)
        stats[0] = scalars[4]; stats[1] = scalars[6]; stats[2] = scalars[3]; stats[3] = scalars[5];
        stats[4] = (float)(red[0] + red[1] + red[2] + red[3]);
    }
}

__global__ void train_loss_kernel(const float* stats, float w_over_n, float igr, float igr_ns, float* out) {
    const float edge_loss = stats[4] * w_over_n;
    out[0] = edge_loss + igr * stats[2] / (stats[0] + 1e-5f) + igr_ns * stats[3] / (stats[1] + 1e-5f);
    out[1] = edge_loss;
}

// torch.optim.Adam (amsgrad=False, weight_decay=0, maximize=False), the arithmetic of torch's fused kernel:
//   m = lerp(m, g, 1-b1);  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// tail_mask / tail_step (may be null: every element trainable, one global step count): per element of the tail [n_geo, n) - the
// scalars variance / beta / gamma - whether it is trainable (requires_grad) and its OWN step count: torch.optim.Adam skips a parameter
// without gradient and starts its `step` state when the parameter first gets one (runner_udf.py:144-154 un-freezes variance / beta late)
// b1, b2 arrive as DOUBLES and 1 - b, b^t are formed in double like torch does (python floats / the fused kernel's double arguments):
// 1.0f - 0.999f is 4.7e-5 off 0.001, which scaled exp_avg_sq by that factor against a torch.optim.Adam checkpoint (round 5).
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, float* step, long long n, long long n_geo,
                                                   float lr_geo, float lr, double b1d, double b2d, float eps, const float* tail_mask,
                                                   float* tail_step) {
    const float t = *step + 1.0f;
    const float b2 = (float)b2d, omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
    __shared__ float s_bc[2];      // the two double-precision powers once per workgroup, not once per thread (9 -> 12 us otherwise)
    if (threadIdx.x == 0) { s_bc[0] = (float)(1.0 - pow(b1d, (double)t)); s_bc[1] = (float)sqrt(1.0 - pow(b2d, (double)t)); }
    __syncthreads();
    const float bc1 = s_bc[0], bc2s = s_bc[1];
    // the geometry range in float4 words when the four buffers are 16-byte aligned (the flat buffers are): a quarter of the threads and memory
    // instructions of the element-wise form (11 -> 7 us at 463 k parameters); the rest - the last < 4 geometry elements and the scalar tail with
    // its mask and per-element step counts - element by element.  Same arithmetic per element either way.
    typedef float f4 __attribute__((ext_vector_type(4)));
    const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const long long n4 = al ? (n_geo >> 2) : 0;
    const float ss_geo = lr_geo / bc1;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
        const f4 gi = reinterpret_cast<const f4*>(g)[q];
        f4 mi = reinterpret_cast<f4*>(m)[q], vi = reinterpret_cast<f4*>(v)[q], pi = reinterpret_cast<f4*>(p)[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mi[k] = mi[k] + (gi[k] - mi[k]) * omb1;
            vi[k] = b2 * vi[k] + omb2 * gi[k] * gi[k];
            pi[k] -= ss_geo * mi[k] / (sqrtf(vi[k]) / bc2s + eps);
        }
        reinterpret_cast<f4*>(m)[q] = mi; reinterpret_cast<f4*>(v)[q] = vi; reinterpret_cast<f4*>(p)[q] = pi;
    }
    for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float c1 = bc1, c2s = bc2s;
        if (i >= n_geo && tail_mask) {
            if (tail_mask[i - n_geo] == 0.0f) continue;                  // frozen: no update, no state change
            const float tt = tail_step[i - n_geo] + 1.0f;
            tail_step[i - n_geo] = tt;
            c1 = (float)(1.0 - pow(b1d, (double)tt)); c2s = (float)sqrt(1.0 - pow(b2d, (double)tt));
        }
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * omb1;
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi; v[i] = vi;
        const float step_size = ((i < n_geo) ? lr_geo : lr) / c1;
        p[i] -= step_size * mi / (sqrtf(vi) / c2s + eps);
    }
}
__global__ void adam_bump_kernel(float* step) { *step += 1.0f; }

int launch_train_stats(const float* edge, const float* true_edge, const float* scalars, int N, float d_scale, float* d_edge, float* stats,
                       hipStream_t st) {
    if (!edge || !true_edge || !scalars || !stats || N < 0) { set_error("train_stats: null pointer"); return EMAP_E_INVALID; }
    hipLaunchKernelGGL(train_stats_kernel, dim3(1), dim3(256), 0, st, edge, true_edge, scalars, N, d_scale, d_edge, stats);
    return check_launch("train_stats");
}
int launch_train_loss(const float* stats, float w_over_n, float igr, float igr_ns, float* out, hipStream_t st) {
    if (!stats || !out) { set_error("train_loss: null pointer"); return EMAP_E_INVALID; }
    hipLaunchKernelGGL(train_loss_kernel, dim3(1), dim3(1), 0, st, stats, w_over_n, igr, igr_ns, out);
    return check_launch("train_loss");
}
int launch_adam(float* p, const float* g, float* m, float* v, float* step, int64_t n, int64_t n_geo, float lr_geo, float lr, double b1,
                double b2, float eps, const float* tail_mask, float* tail_step, hipStream_t st) {
    if (!p || !g || !m || !v || !step || n < 0 || n_geo < 0 || n_geo > n || ((tail_mask == nullptr) != (tail_step == nullptr))) {
        set_error("adam_step: bad arguments");
        return EMAP_E_INVALID;
    }
    if (n == 0) return EMAP_OK;
    const int64_t work = n_geo / 4 + (n - (n_geo & ~(int64_t)3));      // float4 words of the geometry range + the remaining elements
    const int grid = (int)((work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, st, p, g, m, v, step, (long long)n, (long long)n_geo, lr_geo, lr, b1, b2, eps, tail_mask, tail_step);
    hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(1), 0, st, step);
    return check_launch("adam_step");
}

}  // namespace emap
