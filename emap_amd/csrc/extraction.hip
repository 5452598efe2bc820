// extraction.hip - the per-point post-processing of the dense-grid extraction queries (SURVEY par. 8 f2).
//
// get_udf_normals_grid / get_udf_normals_slow (src/edge_extraction/extract_pointcloud.py:76-88, 172-179) estimate the local
// line direction at a query point as the right singular vector of the smallest singular value of the (sampling_N x 3)
// matrix G of UDF gradients at jittered copies of the point:  _, _, vh = torch.linalg.svd(grad_ld); vh[:, -1, :], then
// F.normalize.  That vector is the eigenvector of the smallest eigenvalue of the 3x3 matrix G^T G, so no SVD is needed:
// one pass accumulates the 6 distinct entries of G^T G, the eigenvalue comes from the closed form for symmetric 3x3
// matrices and the eigenvector from the largest cross product of two rows of (G^T G - lambda I).  The sign of a singular
// vector is arbitrary (LAPACK's choice in the reference), and so is the vector itself when the smallest singular value is
// repeated; callers compare directions up to sign where it is defined.
//
// Bound: HBM.  12*k bytes read + 12 written per point (k = sampling_N = 50: 612 B), ~6k + 100 flops.
#include "emap_common.h"
#include <math.h>

namespace emap {

constexpr int EV_PTS = 64;   // points per workgroup (one thread per point after a coalesced stage through LDS)

__global__ __launch_bounds__(EV_PTS) void null_direction_kernel(const float* __restrict__ g, long long n, int k,
                                                               float* __restrict__ dir) {
    extern __shared__ float sm[];                       // EV_PTS * 3k floats
    const long long p0 = (long long)blockIdx.x * EV_PTS;
    const int np = (int)((n - p0 < EV_PTS) ? (n - p0) : EV_PTS);
    const int row = 3 * k;
    const float* src = g + p0 * row;
    for (int i = threadIdx.x; i < np * row; i += EV_PTS) sm[i] = src[i];
    __syncthreads();
    if ((int)threadIdx.x >= np) return;
    const float* v = sm + threadIdx.x * row;
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
    for (int i = 0; i < k; ++i) {
        const double x = v[3 * i], y = v[3 * i + 1], z = v[3 * i + 2];
        a00 += x * x; a01 += x * y; a02 += x * z; a11 += y * y; a12 += y * z; a22 += z * z;
    }
    // smallest eigenvalue of the symmetric PSD matrix A (trigonometric closed form)
    const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    const double q = (a00 + a11 + a22) / 3.0;
    const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1;
    double lam = q;
    if (p2 > 0.0) {
        const double p = sqrt(p2 / 6.0);
        const double ip = 1.0 / p;
        const double c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c01 = a01 * ip, c02 = a02 * ip, c12 = a12 * ip;
        double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
        r = fmin(1.0, fmax(-1.0, r));
        const double phi = acos(r) / 3.0;
        lam = q + 2.0 * p * cos(phi + 2.0943951023931954923);   // + 2*pi/3: the smallest root
    }
    // eigenvector: rows of (A - lam I) span the plane orthogonal to it; take the best-conditioned cross product
    const double r0x = a00 - lam, r0y = a01, r0z = a02;
    const double r1x = a01, r1y = a11 - lam, r1z = a12;
    const double r2x = a02, r2y = a12, r2z = a22 - lam;
    double e0x = r0y * r1z - r0z * r1y, e0y = r0z * r1x - r0x * r1z, e0z = r0x * r1y - r0y * r1x;
    double e1x = r0y * r2z - r0z * r2y, e1y = r0z * r2x - r0x * r2z, e1z = r0x * r2y - r0y * r2x;
    double e2x = r1y * r2z - r1z * r2y, e2y = r1z * r2x - r1x * r2z, e2z = r1x * r2y - r1y * r2x;
    const double n0 = e0x * e0x + e0y * e0y + e0z * e0z, n1 = e1x * e1x + e1y * e1y + e1z * e1z, n2 = e2x * e2x + e2y * e2y + e2z * e2z;
    double ex = e0x, ey = e0y, ez = e0z, nn = n0;
    if (n1 > nn) { ex = e1x; ey = e1y; ez = e1z; nn = n1; }
    if (n2 > nn) { ex = e2x; ey = e2y; ez = e2z; nn = n2; }
    const double tr = a00 + a11 + a22;
    if (!(nn > 1e-24 * tr * tr * tr * tr)) {
        // (numerically) repeated smallest eigenvalue - fewer than 3 independent gradients: the null space has dimension >= 2
        // and the SVD returns an arbitrary member of it.  Take a vector orthogonal to the dominant direction (the largest row
        // of A - lam I is parallel to it for a rank-1 matrix); for A = c I (including 0, where LAPACK's vh is the identity
        // and the reference ends up with its last row) return e_z.
        const double q0 = r0x * r0x + r0y * r0y + r0z * r0z, q1 = r1x * r1x + r1y * r1y + r1z * r1z, q2 = r2x * r2x + r2y * r2y + r2z * r2z;
        double dx = r0x, dy = r0y, dz = r0z, qq = q0;
        if (q1 > qq) { dx = r1x; dy = r1y; dz = r1z; qq = q1; }
        if (q2 > qq) { dx = r2x; dy = r2y; dz = r2z; qq = q2; }
        if (qq > 1e-30 * tr * tr && qq > 0.0) {
            const double ax = fabs(dx), ay = fabs(dy), az = fabs(dz);
            if (ax <= ay && ax <= az) { ex = 0.0; ey = -dz; ez = dy; }          // d x e_x
            else if (ay <= az)        { ex = dz; ey = 0.0; ez = -dx; }          // d x e_y
            else                      { ex = -dy; ey = dx; ez = 0.0; }          // d x e_z
        } else { ex = 0.0; ey = 0.0; ez = 1.0; }
        nn = ex * ex + ey * ey + ez * ez;
    }
    const double inv = 1.0 / sqrt(nn);
    float* o = dir + (p0 + threadIdx.x) * 3;
    o[0] = (float)(ex * inv); o[1] = (float)(ey * inv); o[2] = (float)(ez * inv);
}

int launch_null_direction(const float* g, int64_t n, int k, float* dir, hipStream_t st) {
    if (n <= 0) return EMAP_OK;
    if (k < 1 || k > 128) { set_error("null_direction: sampling_N must be in 1..128 (got %d)", k); return EMAP_E_INVALID; }
    const size_t lds = (size_t)EV_PTS * 3 * k * sizeof(float);
    hipLaunchKernelGGL(null_direction_kernel, dim3((unsigned)((n + EV_PTS - 1) / EV_PTS)), dim3(EV_PTS), lds, st, g, (long long)n, k, dir);
    return check_launch("null_direction");
}

}  // namespace emap
