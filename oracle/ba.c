/*
 * ba.c -- CPU restatement of visualOdometry::BundleAdjust3d2d, src/bundleAdjust.cpp:551-613.
 * TEST INFRASTRUCTURE (see svo_oracle.h).
 *
 * What the reference builds there (g2o, un-vendored; published algorithm restated):
 *   - one VertexSE3Expmap (world -> camera, estimate (R, t) as passed in, :566-574),
 *     oplus:  T <- exp([omega; upsilon]) * T                                   (g2o types_six_dof_expmap)
 *   - N VertexSBAPointXYZ, one per 3-D point, FREE and marginalised (:577-584),  oplus: X <- X + d
 *   - CameraParameters(f = K(0,0), (cx, cy) = (K(0,2), K(1,2)), 0) (:586-591): ONE focal length
 *   - N EdgeProjectXYZ2UV, information I2 (:593-604): e = z - (f * x/z + cx, f * y/z + cy),
 *     x = T.map(X); analytic Jacobians of g2o's linearizeOplus
 *   - OptimizationAlgorithmLevenberg over BlockSolver<6,3> + LinearSolverDense (:552-557):
 *     lambda_0 = 1e-5 * max diag(H); per iteration up to 10 trials; Schur complement of the
 *     point blocks onto the single 6x6 pose block; rho = (chi2 - chi2') / (dx.(lambda dx + b) + 1e-3);
 *     accepted: lambda *= max(1/3, min(2/3, 1 - (2 rho - 1)^3)), ni = 2; rejected: lambda *= ni, ni *= 2
 *   - optimize(10) (:606), and ONLY t is written back (:609-611).
 * Every point has a single observation, so its 3x3 block J^T J is rank 2: the damping is what
 * makes the system solvable -- reproduced as is.
 *
 * Stated deviations: the rotation is kept as a matrix (g2o: unit quaternion, re-normalised per
 * product); the 6x6 system is solved by Cholesky (g2o's LinearSolverDense: Eigen LDLT); sums over the
 * points run in the order of the GPU's reduction (256 strided partial sums, a balanced tree per 64,
 * then the four totals in order) -- any order is an equally valid rounding of the same sums.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "svo_oracle.h"

#define BA_T 256 /* partial sums, as the threads of the GPU workgroup */

typedef struct {
    double A[6];  /* 2x3: d e / d point  */
    double B[12]; /* 2x6: d e / d [omega; upsilon] */
    double e[2];
} ba_lin;

/* error only: e = z - cam_map(T.map(X)) */
static void ba_error(const double *R, const double *t, const double *X, const float *z, double f, double cx,
                     double cy, double *e)
{
    const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
    const double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
    const double w = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    e[0] = (double)z[0] - (x / w * f + cx);
    e[1] = (double)z[1] - (y / w * f + cy);
}

static void ba_linearize(const double *R, const double *t, const double *X, const float *z, double f, double cx,
                         double cy, ba_lin *L)
{
    const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
    const double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
    const double w = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    L->e[0] = (double)z[0] - (x / w * f + cx);
    L->e[1] = (double)z[1] - (y / w * f + cy);
    const double w2 = w * w;
    /* tmp = [f 0 -x/w f; 0 f -y/w f];  A = -1/w * tmp * R */
    const double t02 = -x / w * f, t12 = -y / w * f, s = -1. / w;
    for (int c = 0; c < 3; c++) {
        L->A[c] = s * (f * R[c] + t02 * R[6 + c]);
        L->A[3 + c] = s * (f * R[3 + c] + t12 * R[6 + c]);
    }
    L->B[0] = x * y / w2 * f;
    L->B[1] = -(1. + (x * x / w2)) * f;
    L->B[2] = y / w * f;
    L->B[3] = -1. / w * f;
    L->B[4] = 0.;
    L->B[5] = x / w2 * f;
    L->B[6] = (1. + y * y / w2) * f;
    L->B[7] = -x * y / w2 * f;
    L->B[8] = -x / w * f;
    L->B[9] = 0.;
    L->B[10] = -1. / w * f;
    L->B[11] = y / w2 * f;
}

/* per-point blocks of the normal equations: Hll = A^T A (6 unique: 00 01 02 11 12 22), bl = -A^T e,
 * Hpl = B^T A (6x3) */
static void ba_point_blocks(const ba_lin *L, double *Hll, double *bl, double *Hpl)
{
    const double *A = L->A, *B = L->B;
    Hll[0] = A[0] * A[0] + A[3] * A[3];
    Hll[1] = A[0] * A[1] + A[3] * A[4];
    Hll[2] = A[0] * A[2] + A[3] * A[5];
    Hll[3] = A[1] * A[1] + A[4] * A[4];
    Hll[4] = A[1] * A[2] + A[4] * A[5];
    Hll[5] = A[2] * A[2] + A[5] * A[5];
    for (int c = 0; c < 3; c++)
        bl[c] = -(A[c] * L->e[0] + A[3 + c] * L->e[1]);
    for (int r = 0; r < 6; r++)
        for (int c = 0; c < 3; c++)
            Hpl[3 * r + c] = B[r] * A[c] + B[6 + r] * A[3 + c];
}

/* inverse of the symmetric 3x3 (Hll + lambda I) by cofactors; V, Vi: 6 unique entries */
static void ba_sym3_inv(const double *Hll, double lambda, double *Vi)
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

static double ba_tree_sum(const double *p)
{
    double tot[4];
    for (int w = 0; w < 4; w++) {
        double v[64];
        memcpy(v, p + 64 * w, sizeof(v));
        for (int s = 1; s < 64; s <<= 1)
            for (int i = 0; i < 64; i += 2 * s)
                v[i] = v[i] + v[i + s];
        tot[w] = v[0];
    }
    return ((tot[0] + tot[1]) + tot[2]) + tot[3];
}

static int ba_chol6_solve(const double *Ain, const double *b, double *x)
{
    double L[36];
    memset(L, 0, sizeof(L));
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
            double s = Ain[6 * i + j];
            for (int k = 0; k < j; k++)
                s -= L[6 * i + k] * L[6 * j + k];
            if (i == j) {
                if (!(s > 0))
                    return 0;
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
    return 1;
}

/* SE3Quat::exp([omega; upsilon]) * (R, t) */
static void ba_se3_exp_mul(const double *d, const double *R, const double *t, double *Rn, double *tn)
{
    const double wx = d[0], wy = d[1], wz = d[2];
    const double th = sqrt(wx * wx + wy * wy + wz * wz);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9], E[9], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    if (th < 0.00001) {
        for (int k = 0; k < 9; k++)
            E[k] = ((k % 4) == 0 ? 1. : 0.) + O[k] + O2[k];
        memcpy(V, E, sizeof(V));
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

static double ba_chi2(const double *R, const double *t, const double *X, const float *z, int n, double f, double cx,
                      double cy)
{
    double part[BA_T];
    for (int tid = 0; tid < BA_T; tid++) {
        double s = 0;
        for (int i = tid; i < n; i += BA_T) {
            double e[2];
            ba_error(R, t, X + 3 * i, z + 2 * i, f, cx, cy, e);
            s += e[0] * e[0] + e[1] * e[1];
        }
        part[tid] = s;
    }
    return ba_tree_sum(part);
}

int orc_ba_3d2d(const float *pts2d, const float *pts3d, int n, const double *K4, const double *R9, double *t3,
                int iterations, double *R9_out, double *pts3d_out, double *info)
{
    if (!pts2d || !pts3d || n < 1 || !K4 || !R9 || !t3 || iterations < 0)
        return -1;
    const double f = K4[0], cx = K4[2], cy = K4[3]; /* K(1,1) is not read upstream (:588-590) */
    double R[9], t[3];
    memcpy(R, R9, sizeof(R));
    memcpy(t, t3, sizeof(t));
    double *X = (double *)malloc(sizeof(double) * 3 * n * 2), *Xn = X + 3 * n;
    for (int i = 0; i < 3 * n; i++)
        X[i] = pts3d[i];
    double lambda = 0, ni = 2, chi_first = 0, chi_last = 0;
    int it_run = 0, trials_total = 0;
    static double part[28][BA_T];
    for (int it = 0; it < iterations; it++) {
        const double chi = ba_chi2(R, t, X, pts2d, n, f, cx, cy);
        if (it == 0)
            chi_first = chi_last = chi;
        /* ---- buildSystem: Hpp (21 unique), bp (6); the point blocks are recomputed where needed ---- */
        double maxdiag = 0;
        for (int tid = 0; tid < BA_T; tid++) {
            double acc[27], md = 0;
            memset(acc, 0, sizeof(acc));
            for (int i = tid; i < n; i += BA_T) {
                ba_lin L;
                ba_linearize(R, t, X + 3 * i, pts2d + 2 * i, f, cx, cy, &L);
                int k = 0;
                for (int r = 0; r < 6; r++)
                    for (int c = r; c < 6; c++)
                        acc[k++] += L.B[r] * L.B[c] + L.B[6 + r] * L.B[6 + c];
                for (int r = 0; r < 6; r++)
                    acc[21 + r] += -(L.B[r] * L.e[0] + L.B[6 + r] * L.e[1]);
                const double h0 = L.A[0] * L.A[0] + L.A[3] * L.A[3], h1 = L.A[1] * L.A[1] + L.A[4] * L.A[4],
                             h2 = L.A[2] * L.A[2] + L.A[5] * L.A[5];
                md = fmax(md, fmax(h0, fmax(h1, h2)));
            }
            for (int k = 0; k < 27; k++)
                part[k][tid] = acc[k];
            maxdiag = fmax(maxdiag, md);
        }
        double Hpp[36], bp[6];
        {
            int k = 0;
            for (int r = 0; r < 6; r++)
                for (int c = r; c < 6; c++) {
                    Hpp[6 * r + c] = Hpp[6 * c + r] = ba_tree_sum(part[k]);
                    k++;
                }
            for (int r = 0; r < 6; r++)
                bp[r] = ba_tree_sum(part[21 + r]);
        }
        if (it == 0) { /* computeLambdaInit: tau * max |diag H| over pose and point blocks */
            for (int r = 0; r < 6; r++)
                maxdiag = fmax(maxdiag, fabs(Hpp[7 * r]));
            lambda = 1e-5 * maxdiag;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0, bad = 0;
        do {
            /* ---- Schur complement onto the pose with lambda on every diagonal ---- */
            for (int tid = 0; tid < BA_T; tid++) {
                double acc[27];
                memset(acc, 0, sizeof(acc));
                for (int i = tid; i < n; i += BA_T) {
                    ba_lin L;
                    double Hll[6], bl[3], Hpl[18], Vi[6], Y[18];
                    ba_linearize(R, t, X + 3 * i, pts2d + 2 * i, f, cx, cy, &L);
                    ba_point_blocks(&L, Hll, bl, Hpl);
                    ba_sym3_inv(Hll, lambda, Vi);
                    for (int r = 0; r < 6; r++) { /* Y = Hpl * V^-1 */
                        const double *h = Hpl + 3 * r;
                        Y[3 * r] = h[0] * Vi[0] + h[1] * Vi[1] + h[2] * Vi[2];
                        Y[3 * r + 1] = h[0] * Vi[1] + h[1] * Vi[3] + h[2] * Vi[4];
                        Y[3 * r + 2] = h[0] * Vi[2] + h[1] * Vi[4] + h[2] * Vi[5];
                    }
                    int k = 0;
                    for (int r = 0; r < 6; r++)
                        for (int c = r; c < 6; c++)
                            acc[k++] += Y[3 * r] * Hpl[3 * c] + Y[3 * r + 1] * Hpl[3 * c + 1] + Y[3 * r + 2] * Hpl[3 * c + 2];
                    for (int r = 0; r < 6; r++)
                        acc[21 + r] += Y[3 * r] * bl[0] + Y[3 * r + 1] * bl[1] + Y[3 * r + 2] * bl[2];
                }
                for (int k = 0; k < 27; k++)
                    part[k][tid] = acc[k];
            }
            double S[36], bs[6], dp[6];
            {
                int k = 0;
                for (int r = 0; r < 6; r++)
                    for (int c = r; c < 6; c++) {
                        const double v = Hpp[6 * r + c] + (r == c ? lambda : 0.) - ba_tree_sum(part[k]);
                        S[6 * r + c] = S[6 * c + r] = v;
                        k++;
                    }
                for (int r = 0; r < 6; r++)
                    bs[r] = bp[r] - ba_tree_sum(part[21 + r]);
            }
            const int ok2 = ba_chol6_solve(S, bs, dp);
            double Rn[9], tn[3], tempChi, scale = 0;
            if (ok2) {
                ba_se3_exp_mul(dp, R, t, Rn, tn);
                /* ---- back-substitution of the points, trial estimates, new error, scale ---- */
                for (int tid = 0; tid < BA_T; tid++) {
                    double sc = 0, se = 0;
                    for (int i = tid; i < n; i += BA_T) {
                        ba_lin L;
                        double Hll[6], bl[3], Hpl[18], Vi[6], r3[3], dl[3], e[2];
                        ba_linearize(R, t, X + 3 * i, pts2d + 2 * i, f, cx, cy, &L);
                        ba_point_blocks(&L, Hll, bl, Hpl);
                        ba_sym3_inv(Hll, lambda, Vi);
                        for (int c = 0; c < 3; c++) {
                            double s = 0;
                            for (int r = 0; r < 6; r++)
                                s += Hpl[3 * r + c] * dp[r];
                            r3[c] = bl[c] - s;
                        }
                        dl[0] = Vi[0] * r3[0] + Vi[1] * r3[1] + Vi[2] * r3[2];
                        dl[1] = Vi[1] * r3[0] + Vi[3] * r3[1] + Vi[4] * r3[2];
                        dl[2] = Vi[2] * r3[0] + Vi[4] * r3[1] + Vi[5] * r3[2];
                        for (int c = 0; c < 3; c++) {
                            Xn[3 * i + c] = X[3 * i + c] + dl[c];
                            sc += dl[c] * (lambda * dl[c] + bl[c]);
                        }
                        ba_error(Rn, tn, Xn + 3 * i, pts2d + 2 * i, f, cx, cy, e);
                        se += e[0] * e[0] + e[1] * e[1];
                    }
                    part[0][tid] = sc;
                    part[1][tid] = se;
                }
                scale = ba_tree_sum(part[0]);
                tempChi = ba_tree_sum(part[1]);
                for (int r = 0; r < 6; r++)
                    scale += dp[r] * (lambda * dp[r] + bp[r]);
            } else {
                tempChi = 1.7976931348623157e308; /* std::numeric_limits<double>::max() */
            }
            trials_total++;
            rho = (chi - tempChi);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(tempChi) && ok2) {
                const double q = 2 * rho - 1;
                double alpha = 1. - q * q * q; /* pow(2 rho - 1, 3) upstream */
                alpha = fmin(alpha, 2. / 3.);
                const double sf = fmax(1. / 3., alpha);
                lambda *= sf;
                ni = 2;
                memcpy(R, Rn, sizeof(R));
                memcpy(t, tn, sizeof(t));
                memcpy(X, Xn, sizeof(double) * 3 * n);
                chi_last = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                if (!isfinite(lambda)) {
                    bad = 1;
                    break;
                }
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        it_run = it + 1;
        if (qmax == 10 || rho == 0 || bad)
            break; /* OptimizationAlgorithm::Terminate */
    }
    memcpy(t3, t, sizeof(t)); /* only t is written back upstream (:609-611) */
    if (R9_out)
        memcpy(R9_out, R, sizeof(R));
    if (pts3d_out)
        memcpy(pts3d_out, X, sizeof(double) * 3 * n);
    if (info) {
        info[0] = chi_first;
        info[1] = chi_last;
        info[2] = lambda;
        info[3] = it_run;
        info[4] = trials_total;
    }
    free(X);
    return 0;
}
