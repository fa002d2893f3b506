/*
 * oracle/geometry.c -- CPU restatement of the two-view geometry on the hot path
 * (TEST INFRASTRUCTURE; see svo_oracle.h.  PARITY UNPINNED.)
 *
 *   orc_fransac        cv::findFundamentalMat(p1, p2, FM_RANSAC, thr, 0.99, mask)
 *                      reference call sites src/tracking.cpp:34 (3.0 px) and :75 (1.0 px)
 *   orc_triangulate    cv::triangulatePoints(P1, P2, x1, x2) + the float dehomogenisation
 *                      of src/triangulation.cpp:152-160
 *   orc_transform_points   src/keyFrameManagement.cpp:20-30 / :33-46
 *   orc_get_colors     include/monoUtils.h:180-193
 *   orc_rodrigues*     cv::Rodrigues as used at src/VisualSLAM.cpp:70-74
 *
 * Upstream behaviour restated: RANSAC over 7-point samples,
 * <= 3 models per sample from the cubic det(lambda*F1 + (1-lambda)*F2) = 0, error = max of
 * the two squared point-to-epipolar-line distances, rounded to float and compared with
 * (float)thr^2, at most 1000 iterations with the adaptive update
 * log(1-conf)/log(1-w^7), first-best-wins ordering, degenerate (collinear) samples
 * re-drawn.  The mask is the only output the reference consumes (src/tracking.cpp:35-40).
 *
 * Stated deviations: (1) OpenCV draws samples from cv::RNG seeded with (uint64)-1, a
 * sequential generator; here sample k of iteration i comes from a counter-based hash of
 * (seed, i, k) so that a GPU can evaluate every iteration concurrently and still
 * reproduce the SEQUENTIAL algorithm's answer exactly ("hard parts").
 * (2) the null space of the 7x9 system is taken by Gauss-Jordan elimination with full
 * pivoting instead of an SVD; both span the same pencil, so the candidate matrices agree
 * up to scale, and the epipolar error is scale-invariant.  (3) models are normalised to
 * unit Frobenius norm instead of F[8] = 1.
 */
#include "svo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- counter-based RNG -- */
uint32_t orc_rng_u32(uint64_t seed, uint32_t iter, uint32_t draw)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((((uint64_t)iter << 32) | draw) + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}

#define ORC_MAX_ATTEMPTS 8
#define ORC_MAX_DRAWS 64

/* cv::PointSetRegistrator getSubset: m distinct indices; up to ORC_MAX_ATTEMPTS attempts,
 * an attempt is rejected when check() says the subset is degenerate.  Returns 1 on success. */
typedef int (*subset_check_fn)(const int *idx, int m, const void *user);
static int draw_subset(uint64_t seed, uint32_t iter, int n, int m, int *idx, subset_check_fn check,
                       const void *user)
{
    uint32_t draw = 0;
    for (int attempt = 0; attempt < ORC_MAX_ATTEMPTS; attempt++) {
        int i = 0, guard = 0;
        while (i < m && guard < ORC_MAX_DRAWS) {
            int v = (int)(orc_rng_u32(seed, iter, draw++) % (uint32_t)n);
            guard++;
            int dup = 0;
            for (int j = 0; j < i; j++)
                if (idx[j] == v)
                    dup = 1;
            if (!dup)
                idx[i++] = v;
        }
        if (i < m)
            return 0;
        if (!check || check(idx, m, user))
            return 1;
    }
    return 0;
}
int orc_draw_subset_plain(uint64_t seed, uint32_t iter, int n, int m, int *idx)
{
    return draw_subset(seed, iter, n, m, idx, 0, 0);
}

/* RANSACUpdateNumIters */
static int update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p < 0 ? 0 : (p > 1 ? 1 : p);
    ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
    double num = 1. - p;
    if (num < DBL_MIN)
        num = DBL_MIN;
    double denom = 1. - svo_powi(1. - ep, model_points);
    if (denom < DBL_MIN)
        return 0;
    num = svo_log(num);
    denom = svo_log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}
int orc_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    return update_num_iters(p, ep, model_points, max_iters);
}

/* ---------------------------------------------------------------- 7-point solver ------ */
static double det3(const double *m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
           m[2] * (m[3] * m[7] - m[4] * m[6]);
}

/* real roots of c0 x^3 + c1 x^2 + c2 x + c3 (cv::solveCubic), two Newton polish steps */
static int solve_cubic(const double *c, double *r)
{
    int n = 0;
    double a = c[0], b = c[1], cc = c[2], d = c[3];
    double scale = fabs(a) + fabs(b) + fabs(cc) + fabs(d);
    if (scale == 0)
        return 0;
    if (fabs(a) <= 1e-14 * scale) {
        if (fabs(b) <= 1e-14 * scale) {
            if (fabs(cc) <= 1e-14 * scale)
                return 0;
            r[0] = -d / cc;
            n = 1;
        } else {
            double disc = cc * cc - 4 * b * d;
            if (disc < 0)
                return 0;
            double sq = sqrt(disc);
            r[0] = (-cc + sq) / (2 * b);
            r[1] = (-cc - sq) / (2 * b);
            n = 2;
        }
    } else {
        double a1 = b / a, a2 = cc / a, a3 = d / a;
        double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        double Qcubed = Q * Q * Q, dd = Qcubed - R * R;
        if (dd > 0) {
            double theta = svo_acos(R / sqrt(Qcubed));
            double sqrtQ = sqrt(Q);
            double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
            r[0] = t0 * svo_cos(t1) - t2;
            r[1] = t0 * svo_cos(t1 + 2. * 3.14159265358979323846 / 3) - t2;
            r[2] = t0 * svo_cos(t1 + 4. * 3.14159265358979323846 / 3) - t2;
            n = 3;
        } else if (dd == 0) {
            double e = svo_cbrt(fabs(R));
            if (R > 0)
                e = -e;
            r[0] = 2 * e - a1 * (1. / 3);
            r[1] = -e - a1 * (1. / 3);
            n = 2;
        } else {
            double e = svo_cbrt(sqrt(-dd) + fabs(R));
            if (R > 0)
                e = -e;
            r[0] = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    for (int k = 0; k < n; k++)
        for (int it = 0; it < 2; it++) {
            double x = r[k];
            double f = ((a * x + b) * x + cc) * x + d;
            double fp = (3 * a * x + 2 * b) * x + cc;
            if (fabs(fp) > 1e-300)
                r[k] = x - f / fp;
        }
    return n;
}

int orc_seven_point(const double *x1, const double *x2, double *F)
{
    /* epipolar constraint x2^T F x1 = 0, F row-major: rows of the 7x9 system */
    double A[7][9];
    for (int i = 0; i < 7; i++) {
        double u0 = x1[2 * i], v0 = x1[2 * i + 1], u1 = x2[2 * i], v1 = x2[2 * i + 1];
        A[i][0] = u1 * u0;
        A[i][1] = u1 * v0;
        A[i][2] = u1;
        A[i][3] = v1 * u0;
        A[i][4] = v1 * v0;
        A[i][5] = v1;
        A[i][6] = u0;
        A[i][7] = v0;
        A[i][8] = 1.;
    }
    /* Gauss-Jordan with full pivoting -> reduced row echelon form */
    int colperm[9];
    for (int j = 0; j < 9; j++)
        colperm[j] = j;
    for (int k = 0; k < 7; k++) {
        int pr = k, pc = k;
        double best = -1;
        for (int i = k; i < 7; i++)
            for (int j = k; j < 9; j++)
                if (fabs(A[i][j]) > best) {
                    best = fabs(A[i][j]);
                    pr = i;
                    pc = j;
                }
        if (best < 1e-12)
            return 0; /* rank deficient sample */
        if (pr != k)
            for (int j = 0; j < 9; j++) {
                double t = A[k][j];
                A[k][j] = A[pr][j];
                A[pr][j] = t;
            }
        if (pc != k) {
            for (int i = 0; i < 7; i++) {
                double t = A[i][k];
                A[i][k] = A[i][pc];
                A[i][pc] = t;
            }
            int t = colperm[k];
            colperm[k] = colperm[pc];
            colperm[pc] = t;
        }
        double inv = 1. / A[k][k];
        for (int j = 0; j < 9; j++)
            A[k][j] *= inv;
        for (int i = 0; i < 7; i++)
            if (i != k) {
                double f = A[i][k];
                if (f != 0)
                    for (int j = 0; j < 9; j++)
                        A[i][j] -= f * A[k][j];
            }
    }
    /* null vectors: free columns 7 and 8 of the permuted system */
    double f1[9], f2[9];
    for (int k = 0; k < 7; k++) {
        f1[colperm[k]] = -A[k][7];
        f2[colperm[k]] = -A[k][8];
    }
    f1[colperm[7]] = 1;
    f1[colperm[8]] = 0;
    f2[colperm[7]] = 0;
    f2[colperm[8]] = 1;
    /* F(l) = l*f1 + (1-l)*f2 = f2 + l*(f1-f2);  det F(l) = c0 l^3 + c1 l^2 + c2 l + c3 */
    double G[9], H[9], M[9], c[4];
    for (int i = 0; i < 9; i++) {
        G[i] = f1[i] - f2[i];
        H[i] = f2[i];
    }
    c[0] = det3(G);
    c[3] = det3(H);
    c[1] = 0;
    c[2] = 0;
    for (int row = 0; row < 3; row++) {
        memcpy(M, G, sizeof(M));
        for (int j = 0; j < 3; j++)
            M[3 * row + j] = H[3 * row + j];
        c[1] += det3(M);
        memcpy(M, H, sizeof(M));
        for (int j = 0; j < 3; j++)
            M[3 * row + j] = G[3 * row + j];
        c[2] += det3(M);
    }
    double roots[3];
    int n = solve_cubic(c, roots), nm = 0;
    for (int k = 0; k < n; k++) {
        double *Fk = F + 9 * nm, nrm = 0;
        for (int i = 0; i < 9; i++) {
            Fk[i] = H[i] + roots[k] * G[i];
            nrm += Fk[i] * Fk[i];
        }
        nrm = sqrt(nrm);
        if (!(nrm > 1e-300) || !isfinite(nrm))
            continue;
        for (int i = 0; i < 9; i++)
            Fk[i] /= nrm;
        nm++;
    }
    return nm;
}

float orc_f_error(const double *F, float x1, float y1, float x2, float y2)
{
    double a, b, c, d1, d2, s1, s2;
    a = F[0] * x1 + F[1] * y1 + F[2];
    b = F[3] * x1 + F[4] * y1 + F[5];
    c = F[6] * x1 + F[7] * y1 + F[8];
    s2 = 1. / (a * a + b * b);
    d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6];
    b = F[1] * x2 + F[4] * y2 + F[7];
    c = F[2] * x2 + F[5] * y2 + F[8];
    s1 = 1. / (a * a + b * b);
    d1 = x1 * a + y1 * b + c;
    double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 > e2 ? e1 : e2);
}

typedef struct {
    const float *p1, *p2;
} fsub_t;

/* cv haveCollinearPoints: the LAST selected point against every pair of earlier ones */
static int collinear_last(const float *p, const int *idx, int m)
{
    int i = m - 1;
    for (int j = 0; j < i; j++) {
        double dx1 = p[2 * idx[j]] - p[2 * idx[i]], dy1 = p[2 * idx[j] + 1] - p[2 * idx[i] + 1];
        for (int k = 0; k < j; k++) {
            double dx2 = p[2 * idx[k]] - p[2 * idx[i]], dy2 = p[2 * idx[k] + 1] - p[2 * idx[i] + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <=
                FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return 1;
        }
    }
    return 0;
}
static int f_subset_ok(const int *idx, int m, const void *user)
{
    const fsub_t *u = (const fsub_t *)user;
    return !collinear_last(u->p1, idx, m) && !collinear_last(u->p2, idx, m);
}
int orc_fransac_draw(const float *p1, const float *p2, int n, uint64_t seed, uint32_t iter, int *idx7)
{
    fsub_t u = {p1, p2};
    return draw_subset(seed, iter, n, 7, idx7, f_subset_ok, &u);
}

/* cv::findFundamentalMat below 15 correspondences (upstream fundam.cpp, recalled -- OpenCV is not in the checkout):
 *   n == 7      the 7-point solver runs once on the seven pairs, the mask is set to all ones
 *   8 <= n < 15 `(method & ~3) == FM_RANSAC && npoints >= 15` fails, so the call goes to the LEAST-MEDIAN estimator
 *               (createLMeDSPointSetRegistrator(cb, 7, confidence)): RANSACUpdateNumIters(confidence, 0.45, 7, 1000)
 *               = 300 iterations of 7-point samples; per model the MEDIAN of the float errors ((a + b) * 0.5 of the two
 *               middle ones for an even count); the first model with the smallest median wins; then
 *               sigma = max(2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(median), 0.001) and the inliers are err <= sigma^2;
 *               the result stands if at least 7 are inliers (else the mask is cleared).
 * The threshold argument is not used on this branch.  Reached when tracking is nearly lost -- exactly where
 * PerspectiveNpointEstimation's 8 px retry fires (src/keyFrameManagement.cpp:85-92; call sites src/tracking.cpp:34,75). */
static int cmp_float(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

int orc_f_small(const float *p1, const float *p2, int n, const orc_fransac_params *prm, uint8_t *mask, double *F,
                int *iters_run)
{
    const int M = 7;
    fsub_t u = {p1, p2};
    if (n == M) {
        double x1[14], x2[14], Fs[27];
        for (int k = 0; k < M; k++) {
            x1[2 * k] = p1[2 * k];
            x1[2 * k + 1] = p1[2 * k + 1];
            x2[2 * k] = p2[2 * k];
            x2[2 * k + 1] = p2[2 * k + 1];
        }
        const int nm = orc_seven_point(x1, x2, Fs);
        if (iters_run)
            *iters_run = 0;
        if (nm <= 0)
            return 0; /* runKernel found nothing: an empty matrix, the mask is not touched upstream (cleared here) */
        memset(mask, 1, (size_t)n);
        if (F)
            memcpy(F, Fs, 9 * sizeof(double));
        return n;
    }
    const int niters = update_num_iters(prm->confidence, 0.45, M, 1000);
    double min_median = DBL_MAX, bestF[9] = {0};
    float err[16], sorted[16];
    int it;
    for (it = 0; it < niters; it++) {
        int idx[7];
        if (!draw_subset(prm->seed, (uint32_t)it, n, M, idx, f_subset_ok, &u))
            break;
        double x1[14], x2[14], Fs[27];
        for (int k = 0; k < M; k++) {
            x1[2 * k] = p1[2 * idx[k]];
            x1[2 * k + 1] = p1[2 * idx[k] + 1];
            x2[2 * k] = p2[2 * idx[k]];
            x2[2 * k + 1] = p2[2 * idx[k] + 1];
        }
        const int nm = orc_seven_point(x1, x2, Fs);
        for (int m = 0; m < nm; m++) {
            for (int i = 0; i < n; i++)
                sorted[i] = orc_f_error(Fs + 9 * m, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
            qsort(sorted, (size_t)n, sizeof(float), cmp_float);
            const double median = (n & 1) ? (double)sorted[n / 2] : (double)(sorted[n / 2 - 1] + sorted[n / 2]) * 0.5;
            if (median < min_median) {
                min_median = median;
                memcpy(bestF, Fs + 9 * m, sizeof(bestF));
            }
        }
    }
    if (iters_run)
        *iters_run = it;
    if (!(min_median < DBL_MAX))
        return 0;
    double sigma = 2.5 * 1.4826 * (1. + 5. / (n - M)) * sqrt(min_median);
    if (sigma < 0.001)
        sigma = 0.001;
    const float t = (float)(sigma * sigma);
    int count = 0;
    for (int i = 0; i < n; i++) {
        err[i] = orc_f_error(bestF, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
        mask[i] = err[i] <= t;
        count += mask[i];
    }
    if (count < M) { /* result = count >= modelPoints: no model */
        memset(mask, 0, (size_t)n);
        return 0;
    }
    if (F)
        memcpy(F, bestF, sizeof(bestF));
    return count;
}

int orc_fransac(const float *p1, const float *p2, int n, const orc_fransac_params *prm,
                uint8_t *mask, double *F, int *iters_run)
{
    const int M = 7;
    memset(mask, 0, (size_t)(n > 0 ? n : 0));
    if (iters_run)
        *iters_run = 0;
    if (n < M)
        return 0;
    if (n < 15 && !prm->ransac_below_15) /* upstream: RANSAC only from 15 correspondences on */
        return orc_f_small(p1, p2, n, prm, mask, F, iters_run);
    const float thr = (float)(prm->threshold * prm->threshold);
    int niters = prm->max_iters, best_count = 0;
    double bestF[9] = {0};
    fsub_t u = {p1, p2};
    int it;
    for (it = 0; it < niters; it++) {
        int idx[7];
        if (!draw_subset(prm->seed, (uint32_t)it, n, M, idx, f_subset_ok, &u))
            break; /* getSubset failed: the upstream loop stops */
        double x1[14], x2[14], Fs[27];
        for (int k = 0; k < M; k++) {
            x1[2 * k] = p1[2 * idx[k]];
            x1[2 * k + 1] = p1[2 * idx[k] + 1];
            x2[2 * k] = p2[2 * idx[k]];
            x2[2 * k + 1] = p2[2 * idx[k] + 1];
        }
        int nm = orc_seven_point(x1, x2, Fs);
        for (int m = 0; m < nm; m++) {
            int count = 0;
            /* an integer count: the order of the summands is free (cpu_baseline runs this over the host's threads) */
#pragma omp parallel for reduction(+ : count) schedule(static) if (n >= 2048)
            for (int i = 0; i < n; i++)
                count += orc_f_error(Fs + 9 * m, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) <= thr;
            if (count > (best_count > M - 1 ? best_count : M - 1)) {
                best_count = count;
                memcpy(bestF, Fs + 9 * m, sizeof(bestF));
                niters = update_num_iters(prm->confidence, (double)(n - count) / n, M, niters);
            }
        }
    }
    if (iters_run)
        *iters_run = it;
    if (best_count <= 0)
        return 0;
    for (int i = 0; i < n; i++)
        mask[i] = orc_f_error(bestF, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) <= thr;
    if (F)
        memcpy(F, bestF, sizeof(bestF));
    return best_count;
}

/* ---------------------------------------------------------------- DLT triangulation --- */
/* one-sided (Hestenes) Jacobi SVD of a 4x4 matrix, as cv::SVD does for small matrices;
 * returns the right singular vector of the smallest singular value */
static void smallest_right_singular_vector4(const double *Ain, double *v4)
{
    double A[4][4], V[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            A[i][j] = Ain[4 * i + j];
            V[i][j] = i == j;
        }
    for (int sweep = 0; sweep < 30; sweep++) {
        int rotated = 0;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 4; i++) {
                    al += A[i][p] * A[i][p];
                    be += A[i][q] * A[i][q];
                    ga += A[i][p] * A[i][q];
                }
                if (fabs(ga) <= DBL_EPSILON * sqrt(al * be) || ga == 0)
                    continue;
                rotated = 1;
                double zeta = (be - al) / (2. * ga);
                double t = (zeta >= 0 ? 1. : -1.) / (fabs(zeta) + sqrt(1. + zeta * zeta));
                double c = 1. / sqrt(1. + t * t), s = c * t;
                for (int i = 0; i < 4; i++) {
                    double ap = A[i][p], aq = A[i][q];
                    A[i][p] = c * ap - s * aq;
                    A[i][q] = s * ap + c * aq;
                    double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq;
                    V[i][q] = s * vp + c * vq;
                }
            }
        if (!rotated)
            break;
    }
    int best = 0;
    double bn = DBL_MAX;
    for (int j = 0; j < 4; j++) {
        double nn = 0;
        for (int i = 0; i < 4; i++)
            nn += A[i][j] * A[i][j];
        if (nn < bn) {
            bn = nn;
            best = j;
        }
    }
    for (int i = 0; i < 4; i++)
        v4[i] = V[i][best];
}

void orc_stereo_projections(double fx, double fy, double cx, double cy, double baseline,
                            double *P1, double *P2)
{
    /* K*[I|0] and K*[I|(-b,0,0)^T], src/triangulation.cpp:142-149 */
    double K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            P1[4 * i + j] = K[3 * i + j];
            P2[4 * i + j] = K[3 * i + j];
        }
        P1[4 * i + 3] = 0;
        P2[4 * i + 3] = K[3 * i] * (-baseline);
    }
}

void orc_triangulate(const double *P1, const double *P2, const float *x1, const float *x2,
                     int n, float *out_xyz, float *out_h)
{
    for (int i = 0; i < n; i++) {
        double A[16], v[4];
        const double *P[2] = {P1, P2};
        const float *x[2] = {x1 + 2 * i, x2 + 2 * i};
        for (int j = 0; j < 2; j++) {
            double px = x[j][0], py = x[j][1];
            for (int k = 0; k < 4; k++) {
                A[4 * (2 * j) + k] = px * P[j][8 + k] - P[j][k];
                A[4 * (2 * j + 1) + k] = py * P[j][8 + k] - P[j][4 + k];
            }
        }
        smallest_right_singular_vector4(A, v);
        float h[4] = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        if (out_h)
            memcpy(out_h + 4 * i, h, sizeof(h));
        /* src/triangulation.cpp:154-160: float division, no w ~ 0 or cheirality filter */
        out_xyz[3 * i] = h[0] / h[3];
        out_xyz[3 * i + 1] = h[1] / h[3];
        out_xyz[3 * i + 2] = h[2] / h[3];
    }
}

void orc_transform_points(const double *Rt, const float *in_xyz, int n, float *out_xyz)
{
    for (int i = 0; i < n; i++) {
        float x = in_xyz[3 * i], y = in_xyz[3 * i + 1], z = in_xyz[3 * i + 2];
        for (int r = 0; r < 3; r++)
            out_xyz[3 * i + r] =
                (float)(Rt[4 * r] * x + Rt[4 * r + 1] * y + Rt[4 * r + 2] * z + Rt[4 * r + 3]);
    }
}

void orc_get_colors(const uint8_t *img, int w, int h, int c, const float *xy, int n,
                    float *out_bgr)
{
    for (int i = 0; i < n; i++) {
        int x = (int)xy[2 * i], y = (int)xy[2 * i + 1];
        if (x < 0)
            x = 0;
        if (y < 0)
            y = 0;
        if (x >= w)
            x = w - 1;
        if (y >= h)
            y = h - 1; /* the reference reads out of bounds here; clamped instead */
        for (int k = 0; k < 3; k++)
            out_bgr[3 * i + k] = (float)img[((size_t)y * w + x) * c + (c >= 3 ? k : 0)];
    }
}

/* ---------------------------------------------------------------- Rodrigues ----------- */
void orc_rodrigues(const double *r, double *R)
{
    double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < DBL_EPSILON) {
        for (int i = 0; i < 9; i++)
            R[i] = (i % 4) == 0;
        return;
    }
    double c = svo_cos(th), s = svo_sin(th), c1 = 1. - c, it = 1. / th;
    double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    R[0] = c + c1 * x * x;
    R[1] = c1 * x * y - s * z;
    R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z;
    R[4] = c + c1 * y * y;
    R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y;
    R[7] = c1 * y * z + s * x;
    R[8] = c + c1 * z * z;
}

void orc_rodrigues_inv(const double *R, double *r)
{
    /* cv::Rodrigues matrix -> vector branch (for exact rotations) */
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    double theta = svo_acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            r[0] = r[1] = r[2] = 0;
        } else {
            double t;
            t = (R[0] + 1) * 0.5;
            rx = sqrt(t > 0 ? t : 0);
            t = (R[4] + 1) * 0.5;
            ry = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0))
                rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            r[0] = rx * theta;
            r[1] = ry * theta;
            r[2] = rz * theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        r[0] = rx * vth;
        r[1] = ry * vth;
        r[2] = rz * vth;
    }
}

void orc_compose_camera_pose(const double *rvec, const double *tvec, double *R, double *t)
{
    /* src/VisualSLAM.cpp:70-74: Rodrigues(rvec,R); R = R.t(); t = -R*tvec */
    double Rm[9];
    orc_rodrigues(rvec, Rm);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[3 * i + j] = Rm[3 * j + i];
    for (int i = 0; i < 3; i++)
        t[i] = -(R[3 * i] * tvec[0] + R[3 * i + 1] * tvec[1] + R[3 * i + 2] * tvec[2]);
}
