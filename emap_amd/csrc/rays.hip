// rays.hip - on-device ray / pixel sampler (SURVEY.md par. 8 f3).
//
// Replaces Dataset.gen_random_rays_patches_at (reference src/dataset/dataset.py:222-307): per training step the reference
// draws pixels on the HOST (torch.randint, and python `random.choices` over all H*W pixel probabilities when
// importance_sample=True), builds the rays with a handful of small torch CPU ops and copies six tensors to the GPU.  Once a
// render step takes well under a millisecond that host work dominates.  Here one kernel does all of it from data that
// lives on the device: pixel draw (Philox4x32-10 counter-based stream), edge look-up, p = K^-1 [x, y, 1], rays_v = R p/|p|,
// rays_o = t, depth_scale, ndc coordinates.  Zero host->device copies per step; the step counter itself is a device word, so
// the sampler can sit inside a captured hipGraph.
//
// Importance sampling (dataset.py:236-263): half of the batch uniform, half from `random.choices(pixels, probabilities)` with
// probabilities = 1 - d on pixels whose edge value is > 0.1 and d elsewhere (d = mean edge value of the image).  Two weight
// classes -> pick the class with probability mass n_e (1-d) : n_n d, then a uniform member of the class: exactly that
// distribution, from a per-image list of pixel indices with the edge pixels first (built once on upload).
// The host generators (torch CPU Mersenne / python random) cannot be reproduced on a device; parity is therefore (i) the
// deterministic part - rays of GIVEN pixels equal the reference formulas - and (ii) distribution tests (SURVEY par. 8c).
#include "emap_common.h"

namespace emap {

struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ static void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __device__ __forceinline__ static void gen(uint64_t seed, uint64_t stream, uint64_t index, uint32_t (&out)[4]) {
        uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            round(c, k0, k1);
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};

// uniform integer in [0, n) from 32 random bits (multiply-shift; bias < n / 2^32)
__device__ __forceinline__ int rand_below(uint32_t r, int n) { return (int)(((uint64_t)r * (uint64_t)n) >> 32); }

struct RayArgs {
    EmapRayDataset ds;
    EmapRayBatch out;
    const int64_t* pixels_in;
    const uint64_t* counter;
    uint64_t* bump;          // the counter again when THIS launch increments it (single-workgroup launches), else null
    uint64_t seed, offset;
    int32_t img_idx, batch, importance;
};

__device__ __forceinline__ void sample_ray(const RayArgs& a, uint64_t step, int i);

// batch <= 1024: ONE workgroup, which also increments the step counter once every lane has read it (a second launch for that cost 4.8 us of
// a 6 us job); larger batches: 256-thread workgroups and bump_counter_kernel behind them
__global__ __launch_bounds__(1024) void sample_rays_kernel(const RayArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t step = a.counter ? *a.counter : a.offset;
    if (i < a.batch) sample_ray(a, step, i);
    if (a.bump) {
        __syncthreads();
        if (threadIdx.x == 0) *a.bump = step + 1;
    }
}

__device__ __forceinline__ void sample_ray(const RayArgs& a, uint64_t step, int i) {
    int img = a.img_idx;
    if (img < 0) img = a.ds.image_perm ? a.ds.image_perm[step % (uint64_t)a.ds.n_images] : (int)(step % (uint64_t)a.ds.n_images);
    const int H = a.ds.H, W = a.ds.W, HW = H * W;
    int px, py;
    uint32_t r[4];
    if (!a.pixels_in || a.out.t_rand) Philox::gen(a.seed, step, (uint64_t)i, r);
    // render()'s per-ray jitter torch.rand([N,1]) - 0.5 (udf_renderer_blending.py:719) from word 2 of the ray's draw: U(-0.5, 0.5) on a 2^-24 grid
    if (a.out.t_rand) a.out.t_rand[i] = (float)(r[2] >> 8) * (1.0f / 16777216.0f) - 0.5f;
    if (a.pixels_in) {
        px = (int)a.pixels_in[2 * i]; py = (int)a.pixels_in[2 * i + 1];
    } else {
        const int half = a.batch / 2;
        if (!a.importance || i < half) {                        // dataset.py:233-234 / 244-245
            px = rand_below(r[0], W); py = rand_below(r[1], H);
        } else {                                                // dataset.py:254-260
            const int ne = a.ds.n_edge[img], nn = HW - ne;
            const float d = a.ds.density[img];
            const double we = (double)ne * (1.0 - (double)d), wn = (double)nn * (double)d;
            const double u = ((double)r[0] + 0.5) * (1.0 / 4294967296.0) * (we + wn);
            const int32_t* order = a.ds.pixel_order + (size_t)img * HW;
            int pix;
            if ((u < we && ne > 0) || nn == 0) pix = order[rand_below(r[1], ne)];
            else pix = order[ne + rand_below(r[1], nn)];
            px = pix % W; py = pix / W;
        }
    }
    const float* K = a.ds.kinv + (size_t)img * 9;
    const float* P = a.ds.pose + (size_t)img * 16;
    const float fx = (float)px, fy = (float)py;
    // p = K^-1 [x, y, 1]   (dataset.py:272-277)
    const float p0 = K[0] * fx + K[1] * fy + K[2], p1 = K[3] * fx + K[4] * fy + K[5], p2 = K[6] * fx + K[7] * fy + K[8];
    const float n = sqrtf(p0 * p0 + p1 * p1 + p2 * p2);
    const float v0 = p0 / n, v1 = p1 / n, v2 = p2 / n;          // :279
    if (a.out.depth_scale) a.out.depth_scale[i] = v2;           // :280
    if (a.out.rays_v) {                                         // :281-283
        a.out.rays_v[3 * i + 0] = P[0] * v0 + P[1] * v1 + P[2] * v2;
        a.out.rays_v[3 * i + 1] = P[4] * v0 + P[5] * v1 + P[6] * v2;
        a.out.rays_v[3 * i + 2] = P[8] * v0 + P[9] * v1 + P[10] * v2;
    }
    if (a.out.rays_o) { a.out.rays_o[3 * i] = P[3]; a.out.rays_o[3 * i + 1] = P[7]; a.out.rays_o[3 * i + 2] = P[11]; }   // :284-286
    if (a.out.edge) a.out.edge[i] = a.ds.edges[(size_t)img * HW + (size_t)py * W + px];                                // :270
    if (a.out.ndc_uv) {                                         // :265-267
        a.out.ndc_uv[2 * i] = 2.0f * fx / (float)(W - 1) - 1.0f;
        a.out.ndc_uv[2 * i + 1] = 2.0f * fy / (float)(H - 1) - 1.0f;
    }
    if (a.out.p_cam) { a.out.p_cam[3 * i] = p0; a.out.p_cam[3 * i + 1] = p1; a.out.p_cam[3 * i + 2] = p2; }
    if (a.out.pixels) { a.out.pixels[2 * i] = px; a.out.pixels[2 * i + 1] = py; }
    if (a.out.img_idx && i == 0) a.out.img_idx[0] = img;
}

__global__ void bump_counter_kernel(uint64_t* c) { *c += 1; }

int launch_sample_rays(const EmapRayDataset* ds, int img_idx, int batch, int importance, uint64_t seed, uint64_t offset,
                       uint64_t* counter, const int64_t* pixels_in, const EmapRayBatch* out, hipStream_t st) {
    if (!ds || !out) { set_error("sample_rays: null pointer"); return EMAP_E_INVALID; }
    if (!ds->edges || !ds->kinv || !ds->pose || ds->n_images < 1 || ds->H < 2 || ds->W < 2) { set_error("sample_rays: incomplete dataset"); return EMAP_E_INVALID; }
    if (img_idx >= ds->n_images) { set_error("sample_rays: image %d out of range", img_idx); return EMAP_E_INVALID; }
    if (importance && !pixels_in && (!ds->pixel_order || !ds->n_edge || !ds->density)) { set_error("sample_rays: importance sampling needs pixel_order / n_edge / density"); return EMAP_E_INVALID; }
    if (batch <= 0) return EMAP_OK;
    RayArgs a;
    a.ds = *ds; a.out = *out; a.pixels_in = pixels_in; a.counter = counter; a.seed = seed; a.offset = offset;
    a.img_idx = img_idx; a.batch = batch; a.importance = importance;
    const bool one_wg = batch <= 1024;
    a.bump = one_wg ? counter : nullptr;
    if (one_wg) hipLaunchKernelGGL(sample_rays_kernel, dim3(1), dim3((batch + 63) / 64 * 64), 0, st, a);
    else {
        hipLaunchKernelGGL(sample_rays_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, a);
        if (counter) hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(1), 0, st, counter);
    }
    return check_launch("sample_rays");
}

}  // namespace emap
