// ba.hip -- motion bundle adjustment with free, marginalised points on one pose (gfx950).
//
// Replaces visualOdometry::BundleAdjust3d2d(points_2d, points_3d, K, R, t),
// src/bundleAdjust.cpp:551-613: g2o's OptimizationAlgorithmLevenberg over BlockSolver<6,3> with one
// VertexSE3Expmap, N VertexSBAPointXYZ (setMarginalized(true): Schur-eliminated, NOT fixed) and N
// EdgeProjectXYZ2UV, optimize(10), only t written back (:609-611).
//
// Mapping: ONE workgroup of 256 threads runs the whole Levenberg-Marquardt loop on the device -- no
// host round trip per iteration or trial.  A thread owns the points i = tid, tid + 256, ...; per
// point the 2x3 / 2x6 Jacobians, the 3x3 block inverse and the 6x3 coupling block live in
// registers and are recomputed from (pose, point) in every pass (~300 f64 operations) instead of
// being stored (216 B per point per pass otherwise).  Per trial two passes over the points:
//   1. Schur complement  S = Hpp + lambda I - sum Hpl (Hll + lambda I)^-1 Hpl^T,  bs likewise: 27
//      sums, reduced in a FIXED order (strided partial sums per thread, xor-butterfly per
//      wavefront = a balanced tree, the four wave totals in order through 1 KB of LDS), so the
//      result is the same on every run and equals the oracle's bit for bit up to libm;
//   2. back-substitution of the points, trial estimates, new chi2 and the gain-ratio scale.
// The 6x6 solve, SE3 exponential and the accept / reject rule are wave-uniform scalar work that
// every thread carries redundantly (no broadcast, no extra barrier).
// Bound: f64 VALU latency of one workgroup (the problem is 4096 x ~1 kflop per pass); HBM traffic
// is the points once (20 B each) + 48 B per point per pass of L2-resident estimates.
#include "svo_internal.h"

namespace {

constexpr int BA_T = 256;

struct BaLin {
    double A[6], B[12], e[2];
};

__device__ __forceinline__ void ba_error(const double *R, const double *t, const double *X, float zx, float zy,
                                         double f, double cx, double cy, double *e)
{
    const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
    const double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
    const double w = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    e[0] = (double)zx - (x / w * f + cx);
    e[1] = (double)zy - (y / w * f + cy);
}

// EdgeProjectXYZ2UV::linearizeOplus (g2o types_six_dof_expmap), [omega; upsilon] ordering
__device__ __forceinline__ void ba_linearize(const double *R, const double *t, const double *X, float zx, float zy,
                                             double f, double cx, double cy, BaLin &L)
{
    const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
    const double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
    const double w = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    L.e[0] = (double)zx - (x / w * f + cx);
    L.e[1] = (double)zy - (y / w * f + cy);
    const double w2 = w * w;
    const double t02 = -x / w * f, t12 = -y / w * f, s = -1. / w;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        L.A[c] = s * (f * R[c] + t02 * R[6 + c]);
        L.A[3 + c] = s * (f * R[3 + c] + t12 * R[6 + c]);
    }
    L.B[0] = x * y / w2 * f;
    L.B[1] = -(1. + (x * x / w2)) * f;
    L.B[2] = y / w * f;
    L.B[3] = -1. / w * f;
    L.B[4] = 0.;
    L.B[5] = x / w2 * f;
    L.B[6] = (1. + y * y / w2) * f;
    L.B[7] = -x * y / w2 * f;
    L.B[8] = -x / w * f;
    L.B[9] = 0.;
    L.B[10] = -1. / w * f;
    L.B[11] = y / w2 * f;
}

__device__ __forceinline__ void ba_point_blocks(const BaLin &L, double *Hll, double *bl, double *Hpl)
{
    const double *A = L.A, *B = L.B;
    Hll[0] = A[0] * A[0] + A[3] * A[3];
    Hll[1] = A[0] * A[1] + A[3] * A[4];
    Hll[2] = A[0] * A[2] + A[3] * A[5];
    Hll[3] = A[1] * A[1] + A[4] * A[4];
    Hll[4] = A[1] * A[2] + A[4] * A[5];
    Hll[5] = A[2] * A[2] + A[5] * A[5];
#pragma unroll
    for (int c = 0; c < 3; c++)
        bl[c] = -(A[c] * L.e[0] + A[3 + c] * L.e[1]);
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            Hpl[3 * r + c] = B[r] * A[c] + B[6 + r] * A[3 + c];
}

__device__ __forceinline__ void ba_sym3_inv(const double *Hll, double lambda, double *Vi)
{
    const double a = Hll[0] + lambda, b = Hll[1], c = Hll[2], d = Hll[3] + lambda, e = Hll[4], g = Hll[5] + lambda;
    const double c00 = d * g - e * e, c01 = c * e - b * g, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    const double id = 1. / det;
    Vi[0] = c00 * id;
    Vi[1] = c01 * id;
    Vi[2] = c02 * id;
    Vi[3] = (a * g - c * c) * id;
    Vi[4] = (b * c - a * e) * id;
    Vi[5] = (a * d - b * b) * id;
}

// K per-thread partials -> K totals in every thread, fixed order: balanced tree inside each
// wavefront (xor butterfly, commutative additions: every lane holds the same bits), then the four
// wave totals ((w0 + w1) + w2) + w3 through LDS.
template <int K> __device__ __forceinline__ void block_sum(double (&v)[K], double *s_red /* [4][K] */)
{
#pragma unroll
    for (int k = 0; k < K; k++) {
        double x = v[k];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1)
            x = x + __shfl_xor(x, m, 64);
        v[k] = x;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();  // the previous reduction's readers are done with s_red
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < K; k++)
            s_red[wave * K + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++)
        v[k] = ((s_red[k] + s_red[K + k]) + s_red[2 * K + k]) + s_red[3 * K + k];
}

__device__ __forceinline__ double block_max(double x, double *s_red)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1)
        x = fmax(x, __shfl_xor(x, m, 64));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0)
        s_red[wave] = x;
    __syncthreads();
    return fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
}

__device__ bool ba_chol6_solve(const double *Ain, const double *b, double *x)
{
    double L[36];
    for (int k = 0; k < 36; k++)
        L[k] = 0;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
            double s = Ain[6 * i + j];
            for (int k = 0; k < j; k++)
                s -= L[6 * i + k] * L[6 * j + k];
            if (i == j) {
                if (!(s > 0))
                    return false;
                L[6 * i + i] = sqrt(s);
            } else
                L[6 * i + j] = s / L[6 * j + j];
        }
    double y[6];
    for (int i = 0; i < 6; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++)
            s -= L[6 * i + k] * y[k];
        y[i] = s / L[6 * i + i];
    }
    for (int i = 5; i >= 0; i--) {
        double s = y[i];
        for (int k = i + 1; k < 6; k++)
            s -= L[6 * k + i] * x[k];
        x[i] = s / L[6 * i + i];
    }
    return true;
}

// SE3Quat::exp([omega; upsilon]) * (R, t)
__device__ void ba_se3_exp_mul(const double *d, const double *R, const double *t, double *Rn, double *tn)
{
    const double wx = d[0], wy = d[1], wz = d[2];
    const double th = sqrt(wx * wx + wy * wy + wz * wz);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9], E[9], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    if (th < 0.00001) {
        for (int k = 0; k < 9; k++) {
            E[k] = ((k % 4) == 0 ? 1. : 0.) + O[k] + O2[k];
            V[k] = E[k];
        }
    } else {
        const double sn = svo_sin(th), cn = svo_cos(th);
        const double a = sn / th, b = (1. - cn) / (th * th), c = (th - sn) / (th * th * th);
        for (int k = 0; k < 9; k++) {
            const double I = (k % 4) == 0 ? 1. : 0.;
            E[k] = I + a * O[k] + b * O2[k];
            V[k] = I + b * O[k] + c * O2[k];
        }
    }
    double u[3];
    for (int i = 0; i < 3; i++)
        u[i] = V[3 * i] * d[3] + V[3 * i + 1] * d[4] + V[3 * i + 2] * d[5];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            Rn[3 * i + j] = E[3 * i] * R[j] + E[3 * i + 1] * R[3 + j] + E[3 * i + 2] * R[6 + j];
        tn[i] = (E[3 * i] * t[0] + E[3 * i + 1] * t[1] + E[3 * i + 2] * t[2]) + u[i];
    }
}

struct BaResult {  // device -> host record
    double t[3], R[9], info[5];
};

struct BaArgs {
    const float *pts2d, *pts3d;
    int n, iterations;
    double f, cx, cy;
    double R[9], t[3];
    double *X, *Xn;  // n*3 doubles each: accepted and trial point estimates
    BaResult *out;
};

__global__ __launch_bounds__(BA_T) void ba_3d2d_kernel(BaArgs a)
{
    __shared__ double s_red[4 * 27];
    const int tid = threadIdx.x, n = a.n;
    const double f = a.f, cx = a.cx, cy = a.cy;
    const float2 *__restrict__ z = reinterpret_cast<const float2 *>(a.pts2d);
    double *__restrict__ X = a.X, *__restrict__ Xn = a.Xn;
    double R[9], t[3];
    for (int k = 0; k < 9; k++)
        R[k] = a.R[k];
    for (int k = 0; k < 3; k++)
        t[k] = a.t[k];
    for (int i = tid; i < n; i += BA_T)
        for (int c = 0; c < 3; c++)
            X[3 * i + c] = (double)a.pts3d[3 * i + c];
    // a thread only ever reads the X / Xn entries it wrote itself: no barrier is needed for them
    double lambda = 0, ni = 2, chi_first = 0, chi_last = 0;
    int it_run = 0, trials_total = 0;
    for (int it = 0; it < a.iterations; it++) {
        // ---- computeActiveErrors ----
        double c1[1] = {0};
        for (int i = tid; i < n; i += BA_T) {
            double e[2];
            ba_error(R, t, X + 3 * i, z[i].x, z[i].y, f, cx, cy, e);
            c1[0] += e[0] * e[0] + e[1] * e[1];
        }
        block_sum<1>(c1, s_red);
        const double chi = c1[0];
        if (it == 0)
            chi_first = chi_last = chi;
        // ---- buildSystem: Hpp, bp ----
        double acc[27], md = 0;
#pragma unroll
        for (int k = 0; k < 27; k++)
            acc[k] = 0;
        for (int i = tid; i < n; i += BA_T) {
            BaLin L;
            ba_linearize(R, t, X + 3 * i, z[i].x, z[i].y, f, cx, cy, L);
            int k = 0;
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = r; c < 6; c++)
                    acc[k++] += L.B[r] * L.B[c] + L.B[6 + r] * L.B[6 + c];
#pragma unroll
            for (int r = 0; r < 6; r++)
                acc[21 + r] += -(L.B[r] * L.e[0] + L.B[6 + r] * L.e[1]);
            const double h0 = L.A[0] * L.A[0] + L.A[3] * L.A[3], h1 = L.A[1] * L.A[1] + L.A[4] * L.A[4],
                         h2 = L.A[2] * L.A[2] + L.A[5] * L.A[5];
            md = fmax(md, fmax(h0, fmax(h1, h2)));
        }
        block_sum<27>(acc, s_red);
        double Hpp[36], bp[6];
        {
            int k = 0;
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = r; c < 6; c++) {
                    Hpp[6 * r + c] = Hpp[6 * c + r] = acc[k];
                    k++;
                }
#pragma unroll
            for (int r = 0; r < 6; r++)
                bp[r] = acc[21 + r];
        }
        if (it == 0) {  // computeLambdaInit: tau * max |diag H|
            double maxdiag = block_max(md, s_red);
            for (int r = 0; r < 6; r++)
                maxdiag = fmax(maxdiag, fabs(Hpp[7 * r]));
            lambda = 1e-5 * maxdiag;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        bool bad = false;
        do {
            // ---- pass 1: Schur complement onto the pose ----
#pragma unroll
            for (int k = 0; k < 27; k++)
                acc[k] = 0;
            for (int i = tid; i < n; i += BA_T) {
                BaLin L;
                double Hll[6], bl[3], Hpl[18], Vi[6], Y[18];
                ba_linearize(R, t, X + 3 * i, z[i].x, z[i].y, f, cx, cy, L);
                ba_point_blocks(L, Hll, bl, Hpl);
                ba_sym3_inv(Hll, lambda, Vi);
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const double *h = Hpl + 3 * r;
                    Y[3 * r] = h[0] * Vi[0] + h[1] * Vi[1] + h[2] * Vi[2];
                    Y[3 * r + 1] = h[0] * Vi[1] + h[1] * Vi[3] + h[2] * Vi[4];
                    Y[3 * r + 2] = h[0] * Vi[2] + h[1] * Vi[4] + h[2] * Vi[5];
                }
                int k = 0;
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = r; c < 6; c++)
                        acc[k++] += Y[3 * r] * Hpl[3 * c] + Y[3 * r + 1] * Hpl[3 * c + 1] + Y[3 * r + 2] * Hpl[3 * c + 2];
#pragma unroll
                for (int r = 0; r < 6; r++)
                    acc[21 + r] += Y[3 * r] * bl[0] + Y[3 * r + 1] * bl[1] + Y[3 * r + 2] * bl[2];
            }
            block_sum<27>(acc, s_red);
            double S[36], bs[6], dp[6];
            {
                int k = 0;
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = r; c < 6; c++) {
                        const double v = Hpp[6 * r + c] + (r == c ? lambda : 0.) - acc[k];
                        S[6 * r + c] = S[6 * c + r] = v;
                        k++;
                    }
#pragma unroll
                for (int r = 0; r < 6; r++)
                    bs[r] = bp[r] - acc[21 + r];
            }
            const bool ok2 = ba_chol6_solve(S, bs, dp);  // wave-uniform
            double Rn[9], tn[3], tempChi, scale = 0;
            if (ok2) {
                ba_se3_exp_mul(dp, R, t, Rn, tn);
                // ---- pass 2: points' back-substitution, trial estimates, new error, scale ----
                double c2[2] = {0, 0};
                for (int i = tid; i < n; i += BA_T) {
                    BaLin L;
                    double Hll[6], bl[3], Hpl[18], Vi[6], r3[3], dl[3], e[2], xn[3];
                    ba_linearize(R, t, X + 3 * i, z[i].x, z[i].y, f, cx, cy, L);
                    ba_point_blocks(L, Hll, bl, Hpl);
                    ba_sym3_inv(Hll, lambda, Vi);
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        double s = 0;
#pragma unroll
                        for (int r = 0; r < 6; r++)
                            s += Hpl[3 * r + c] * dp[r];
                        r3[c] = bl[c] - s;
                    }
                    dl[0] = Vi[0] * r3[0] + Vi[1] * r3[1] + Vi[2] * r3[2];
                    dl[1] = Vi[1] * r3[0] + Vi[3] * r3[1] + Vi[4] * r3[2];
                    dl[2] = Vi[2] * r3[0] + Vi[4] * r3[1] + Vi[5] * r3[2];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        xn[c] = X[3 * i + c] + dl[c];
                        Xn[3 * i + c] = xn[c];
                        c2[0] += dl[c] * (lambda * dl[c] + bl[c]);
                    }
                    ba_error(Rn, tn, xn, z[i].x, z[i].y, f, cx, cy, e);
                    c2[1] += e[0] * e[0] + e[1] * e[1];
                }
                block_sum<2>(c2, s_red);
                scale = c2[0];
                tempChi = c2[1];
                for (int r = 0; r < 6; r++)
                    scale += dp[r] * (lambda * dp[r] + bp[r]);
            } else {
                tempChi = 1.7976931348623157e308;
            }
            trials_total++;
            rho = (chi - tempChi);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(tempChi) && ok2) {
                const double q = 2 * rho - 1;
                double alpha = 1. - q * q * q;
                alpha = fmin(alpha, 2. / 3.);
                const double sf = fmax(1. / 3., alpha);
                lambda *= sf;
                ni = 2;
                for (int k = 0; k < 9; k++)
                    R[k] = Rn[k];
                for (int k = 0; k < 3; k++)
                    t[k] = tn[k];
                for (int i = tid; i < n; i += BA_T)
                    for (int c = 0; c < 3; c++)
                        X[3 * i + c] = Xn[3 * i + c];
                chi_last = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                if (!isfinite(lambda)) {
                    bad = true;
                    break;
                }
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        it_run = it + 1;
        if (qmax == 10 || rho == 0 || bad)
            break;  // OptimizationAlgorithm::Terminate
    }
    if (tid == 0) {
        for (int k = 0; k < 3; k++)
            a.out->t[k] = t[k];
        for (int k = 0; k < 9; k++)
            a.out->R[k] = R[k];
        a.out->info[0] = chi_first;
        a.out->info[1] = chi_last;
        a.out->info[2] = lambda;
        a.out->info[3] = it_run;
        a.out->info[4] = trials_total;
    }
}

}  // namespace

extern "C" int svo_ba_3d2d(svo_ctx *ctx, const float *pts2d, const float *pts3d, int n, const double *K4,
                           const double *R9, double *t3, int iterations, double *R9_out, double *pts3d_out,
                           double *info, int mem)
{
    SVO_CHECK_ARG(ctx && pts2d && pts3d && n >= 1 && K4 && R9 && t3 && iterations >= 0);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    SVO_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ctx->w_a.ensure((size_t)n * 24)) || (rc = ctx->w_b.ensure((size_t)n * 24)) ||
        (rc = ctx->w_c.ensure(sizeof(BaResult))))
        return rc;
    BaArgs a;
    a.pts2d = pts2d;
    a.pts3d = pts3d;
    if (mem == SVO_MEM_HOST) {
        if ((rc = ctx->s_a.ensure((size_t)n * 8)) || (rc = ctx->s_b.ensure((size_t)n * 12)))
            return rc;
        SVO_HIP(hipMemcpyAsync(ctx->s_a.p, pts2d, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        SVO_HIP(hipMemcpyAsync(ctx->s_b.p, pts3d, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
        a.pts2d = ctx->s_a.as<float>();
        a.pts3d = ctx->s_b.as<float>();
    }
    a.n = n;
    a.iterations = iterations;
    a.f = K4[0];  // CameraParameters(K(0,0), (K(0,2), K(1,2)), 0): ONE focal length, :588-590
    a.cx = K4[2];
    a.cy = K4[3];
    memcpy(a.R, R9, sizeof(a.R));
    memcpy(a.t, t3, sizeof(a.t));
    a.X = ctx->w_a.as<double>();
    a.Xn = ctx->w_b.as<double>();
    a.out = reinterpret_cast<BaResult *>(ctx->w_c.p);
    {
        ScopedKernelTime tm(ctx, SVO_K_PNP);
        hipLaunchKernelGGL(ba_3d2d_kernel, dim3(1), dim3(BA_T), 0, ctx->stream, a);
    }
    SVO_HIP(hipGetLastError());
    BaResult *h = reinterpret_cast<BaResult *>(ctx->pinned);
    SVO_HIP(hipMemcpyAsync(h, a.out, sizeof(BaResult), hipMemcpyDeviceToHost, ctx->stream));
    if (pts3d_out)  // HOST doubles in both modes
        SVO_HIP(hipMemcpyAsync(pts3d_out, a.X, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = svo_wait(ctx)))
        return rc;
    memcpy(t3, h->t, sizeof(h->t));  // the only value the reference writes back (:609-611)
    if (R9_out)
        memcpy(R9_out, h->R, sizeof(h->R));
    if (info)
        memcpy(info, h->info, sizeof(h->info));
    return SVO_OK;
}
