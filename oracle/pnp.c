/*
 * oracle/pnp.c -- CPU restatement of cv::solvePnPRansac as the reference calls it
 * (TEST INFRASTRUCTURE; see svo_oracle.h.  PARITY UNPINNED.)
 *
 * Reference call sites: src/keyFrameManagement.cpp:84
 *   solvePnPRansac(obj, img, K, dist=0, rvec, tvec, false, 100, 1.0, 0.99, inliers)
 * and the retry at :88 with (100, 8.0, 0.98).  Default flags = SOLVEPNP_ITERATIVE.
 *
 * Upstream algorithm restated: RANSAC over 5-point samples solved
 * by EPnP (Lepetit/Moreno-Noguer/Fua: 4 control points from the PCA of the sample,
 * barycentric coordinates, 2n x 12 system M, the four eigenvectors of M^T M with the
 * smallest eigenvalues, betas from the N = 1, 2, 3 linearisations each polished by 5
 * Gauss-Newton steps on the 6 control-point distances, rigid alignment by the SVD of the
 * 3x3 cross-covariance, best reprojection error wins); inlier iff the squared
 * reprojection error (float) <= (float)thr^2; at most `iterations` iterations with the
 * adaptive bound log(1-conf)/log(1-w^5); then one iterative refinement (Levenberg-
 * Marquardt on the reprojection error) over the inliers of the best hypothesis, seeded
 * with that hypothesis.  Outputs rvec/tvec (double) and the inlier index list.
 *
 * Stated deviations: counter-based sampling (see geometry.c); symmetric eigenproblems and
 * the 3x3 SVD use cyclic Jacobi with a fixed round-robin pair order (cv::SVD is a Jacobi
 * method too, with a different sweep order); least-squares sub-problems go through the
 * normal equations + Jacobi pseudo-inverse; the refinement perturbs the pose on the left
 * (R <- exp(dw) R) instead of differentiating through Rodrigues -- same minimiser.
 */
#include "svo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_draw_subset_plain(uint64_t seed, uint32_t iter, int n, int m, int *idx);
int orc_update_num_iters(double p, double ep, int model_points, int max_iters);

/* ---- cyclic Jacobi for symmetric matrices, round-robin ("tournament") pair order ------- */
/* The pair schedule is the one a wavefront executes in parallel on the GPU: n-1 rounds
 * (n even; odd n is padded with an idle player) of n/2 disjoint pairs.                    */
static int rr_pairs(int n, int round, int (*pairs)[2])
{
    int m = n + (n & 1), cnt = 0;
    for (int k = 0; k < m / 2; k++) {
        int a = k == 0 ? m - 1 : (round + k) % (m - 1);
        int b = k == 0 ? round % (m - 1) : (round - k + (m - 1)) % (m - 1);
        if (a >= n || b >= n)
            continue;
        pairs[cnt][0] = a < b ? a : b;
        pairs[cnt][1] = a < b ? b : a;
        cnt++;
    }
    return cnt;
}

/* A (n x n, row-major, symmetric) -> eigenvalues w (diagonal), eigenvectors = COLUMNS of V.
 * Per round: the rotations of all (disjoint) pairs are computed from the current matrix,
 * then ALL column updates, then ALL row updates, then V -- the order a wavefront applies
 * them in parallel, so both implementations round identically.                            */
void orc_jacobi_eigen_sym(int n, double *A, double *V, double *w, int sweeps)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            V[i * n + j] = i == j;
    int m = n + (n & 1);
    int pairs[16][2];
    double cs[16][2];
    for (int s = 0; s < sweeps; s++)
        for (int r = 0; r < m - 1; r++) {
            int np = rr_pairs(n, r, pairs);
            for (int e = 0; e < np; e++) {
                int p = pairs[e][0], q = pairs[e][1];
                double apq = A[p * n + q];
                double app = A[p * n + p], aqq = A[q * n + q];
                /* a rotation below the resolution of a double is the identity: skip its arithmetic */
                if (fabs(apq) <= 1e-19 * (fabs(app) + fabs(aqq))) {
                    cs[e][0] = 1.;
                    cs[e][1] = 0.;
                    continue;
                }
                /* t = tan(phi) = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = a / b, written with
                 * ONE division: h = |a| + sqrt(a^2 + b^2), c = h / sqrt(h^2 + b^2), s = +-b / sqrt(h^2 + b^2) */
                double a = aqq - app, b = 2. * apq;
                double h = fabs(a) + sqrt(a * a + b * b);
                double inv = 1. / sqrt(h * h + b * b);
                cs[e][0] = h * inv;
                cs[e][1] = (a >= 0 ? b : -b) * inv;
            }
            for (int e = 0; e < np; e++) { /* columns p,q:  A <- A J */
                int p = pairs[e][0], q = pairs[e][1];
                double c = cs[e][0], sn = cs[e][1];
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - sn * akq;
                    A[k * n + q] = sn * akp + c * akq;
                }
            }
            for (int e = 0; e < np; e++) { /* rows p,q:  A <- J^T A */
                int p = pairs[e][0], q = pairs[e][1];
                double c = cs[e][0], sn = cs[e][1];
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - sn * aqk;
                    A[q * n + k] = sn * apk + c * aqk;
                }
            }
            for (int e = 0; e < np; e++) {
                int p = pairs[e][0], q = pairs[e][1];
                double c = cs[e][0], sn = cs[e][1];
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - sn * vkq;
                    V[k * n + q] = sn * vkp + c * vkq;
                }
            }
        }
    for (int i = 0; i < n; i++)
        w[i] = A[i * n + i];
}

/* min ||A x - b|| (m x n, n <= 5) through the normal equations, Tikhonov-damped by 1e-14 of
 * the mean diagonal (stands in for the SVD / QR solves of upstream EPnP; a singular system
 * yields x = 0) and solved by Cholesky.                                                    */
static void lstsq_small(int m, int n, const double *A, const double *b, double *x)
{
    double N[25], rhs[5], L[25], y[5];
    double tr = 0;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int k = 0; k < m; k++)
                s += A[k * n + i] * A[k * n + j];
            N[i * n + j] = s;
        }
        double s = 0;
        for (int k = 0; k < m; k++)
            s += A[k * n + i] * b[k];
        rhs[i] = s;
        tr += N[i * n + i];
    }
    const double damp = 1e-14 * tr / n;
    for (int i = 0; i < n; i++)
        x[i] = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            double s = N[i * n + j] + (i == j ? damp : 0.);
            for (int k = 0; k < j; k++)
                s -= L[i * n + k] * L[j * n + k];
            if (i == j) {
                if (!(s > 0))
                    return;
                L[i * n + i] = sqrt(s);
            } else
                L[i * n + j] = s / L[j * n + j];
        }
    for (int i = 0; i < n; i++) {
        double s = rhs[i];
        for (int k = 0; k < i; k++)
            s -= L[i * n + k] * y[k];
        y[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = y[i];
        for (int k = i + 1; k < n; k++)
            s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
}

/* SVD of a 3x3 matrix by one-sided Jacobi: A = U diag(s) V^T */
static void svd3(const double *Ain, double *U, double *Vout)
{
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(A, Ain, sizeof(A));
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 12; sweep++) {
        int rotated = 0;
        for (int e = 0; e < 3; e++) {
            int p = PQ[e][0], q = PQ[e][1];
            double al = 0, be = 0, ga = 0;
            for (int i = 0; i < 3; i++) {
                al += A[3 * i + p] * A[3 * i + p];
                be += A[3 * i + q] * A[3 * i + q];
                ga += A[3 * i + p] * A[3 * i + q];
            }
            if (ga == 0 || fabs(ga) <= DBL_EPSILON * sqrt(al * be))
                continue;
            rotated = 1;
            /* same rotation as in orc_jacobi_eigen_sym, one division */
            double a = be - al, b = 2. * ga;
            double h = fabs(a) + sqrt(a * a + b * b);
            double inv = 1. / sqrt(h * h + b * b);
            double c = h * inv, s = (a >= 0 ? b : -b) * inv;
            for (int i = 0; i < 3; i++) {
                double ap = A[3 * i + p], aq = A[3 * i + q];
                A[3 * i + p] = c * ap - s * aq;
                A[3 * i + q] = s * ap + c * aq;
                double vp = V[3 * i + p], vq = V[3 * i + q];
                V[3 * i + p] = c * vp - s * vq;
                V[3 * i + q] = s * vp + c * vq;
            }
        }
        if (!rotated)
            break;
    }
    /* columns of A are u_j * s_j; order by decreasing norm, rebuild a right-handed U */
    double nrm[3];
    int ord[3] = {0, 1, 2};
    for (int j = 0; j < 3; j++)
        nrm[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
    for (int a = 0; a < 2; a++)
        for (int b = a + 1; b < 3; b++)
            if (nrm[ord[b]] > nrm[ord[a]]) {
                int t = ord[a];
                ord[a] = ord[b];
                ord[b] = t;
            }
    double Uc[3][3], Vc[3][3];
    for (int k = 0; k < 3; k++) {
        int j = ord[k];
        for (int i = 0; i < 3; i++) {
            Vc[k][i] = V[3 * i + j];
            Uc[k][i] = nrm[j] > 0 ? A[3 * i + j] / nrm[j] : 0;
        }
    }
    /* a (near-)zero singular value leaves its u undefined: complete the basis */
    if (!(nrm[ord[2]] > 1e-12 * nrm[ord[0]])) {
        Uc[2][0] = Uc[0][1] * Uc[1][2] - Uc[0][2] * Uc[1][1];
        Uc[2][1] = Uc[0][2] * Uc[1][0] - Uc[0][0] * Uc[1][2];
        Uc[2][2] = Uc[0][0] * Uc[1][1] - Uc[0][1] * Uc[1][0];
    }
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < 3; i++) {
            U[3 * i + k] = Uc[k][i];
            Vout[3 * i + k] = Vc[k][i];
        }
}

/* ---- EPnP -------------------------------------------------------------------------------- */
#define EPNP_MAXN 16

static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static void compute_rt_from_betas(const double *v[4], const double *betas, const double (*alphas)[4],
                                  const double (*pws)[3], int n, double *R, double *t, double (*pcs)[3])
{
    double ccs[4][3];
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 3; k++)
            ccs[i][k] = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 3; k++)
                ccs[j][k] += betas[i] * v[i][3 * j + k];
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++)
            pcs[i][k] = alphas[i][0] * ccs[0][k] + alphas[i][1] * ccs[1][k] + alphas[i][2] * ccs[2][k] +
                        alphas[i][3] * ccs[3][k];
    if (pcs[0][2] < 0.)
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 3; k++)
                pcs[i][k] = -pcs[i][k];
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            pc0[k] += pcs[i][k];
            pw0[k] += pws[i][k];
        }
    for (int k = 0; k < 3; k++) {
        pc0[k] /= n;
        pw0[k] /= n;
    }
    double ABt[9] = {0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++)
                ABt[3 * j + k] += (pcs[i][j] - pc0[j]) * (pws[i][k] - pw0[k]);
    double U[9], V[9];
    svd3(ABt, U, V);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
    double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                 R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (det < 0) {
        R[6] = -R[6];
        R[7] = -R[7];
        R[8] = -R[8];
    }
    for (int i = 0; i < 3; i++)
        t[i] = pc0[i] - dot3(R + 3 * i, pw0);
}

static double reprojection_error(const double *R, const double *t, const double (*pws)[3], const double (*us)[2],
                                 int n, const double *K4)
{
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double Xc = dot3(R, pws[i]) + t[0], Yc = dot3(R + 3, pws[i]) + t[1];
        double inv_Zc = 1.0 / (dot3(R + 6, pws[i]) + t[2]);
        double ue = K4[2] + K4[0] * Xc * inv_Zc, ve = K4[3] + K4[1] * Yc * inv_Zc;
        double du = us[i][0] - ue, dv = us[i][1] - ve;
        sum += sqrt(du * du + dv * dv);
    }
    return sum / n;
}

static void gauss_newton_betas(const double (*L)[10], const double *rho, double *b)
{
    for (int it = 0; it < 5; it++) {
        double A[24], r[6], x[4];
        for (int i = 0; i < 6; i++) {
            const double *l = L[i];
            A[4 * i + 0] = 2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3];
            A[4 * i + 1] = l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3];
            A[4 * i + 2] = l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3];
            A[4 * i + 3] = l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3];
            r[i] = rho[i] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] +
                             l[4] * b[1] * b[2] + l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] +
                             l[8] * b[2] * b[3] + l[9] * b[3] * b[3]);
        }
        lstsq_small(6, 4, A, r, x);
        for (int k = 0; k < 4; k++)
            b[k] += x[k];
    }
}

int orc_epnp(const double *obj, const double *img, int n, const double *K4, double *R, double *t)
{
    if (n < 4 || n > EPNP_MAXN)
        return -1;
    const double fu = K4[0], fv = K4[1], uc = K4[2], vc = K4[3];
    double pws[EPNP_MAXN][3], us[EPNP_MAXN][2], alphas[EPNP_MAXN][4], cws[4][3];
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++)
            pws[i][k] = obj[3 * i + k];
        us[i][0] = img[2 * i];
        us[i][1] = img[2 * i + 1];
    }
    /* control points: centroid + principal directions scaled by sqrt(lambda/n) */
    for (int k = 0; k < 3; k++) {
        cws[0][k] = 0;
        for (int i = 0; i < n; i++)
            cws[0][k] += pws[i][k];
        cws[0][k] /= n;
    }
    double C[9] = {0}, Vc[9], wc[3];
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                C[3 * a + b] += (pws[i][a] - cws[0][a]) * (pws[i][b] - cws[0][b]);
    orc_jacobi_eigen_sym(3, C, Vc, wc, 5);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; a++)
        for (int b = a + 1; b < 3; b++)
            if (wc[ord[b]] > wc[ord[a]]) {
                int tt = ord[a];
                ord[a] = ord[b];
                ord[b] = tt;
            }
    for (int i = 1; i < 4; i++) {
        double lam = wc[ord[i - 1]];
        double k = sqrt((lam > 0 ? lam : 0) / n);
        for (int j = 0; j < 3; j++)
            cws[i][j] = cws[0][j] + k * Vc[3 * j + ord[i - 1]];
    }
    /* barycentric coordinates */
    double CC[9], CCi[9];
    for (int i = 0; i < 3; i++)
        for (int j = 1; j < 4; j++)
            CC[3 * i + j - 1] = cws[j][i] - cws[0][i];
    double det = CC[0] * (CC[4] * CC[8] - CC[5] * CC[7]) - CC[1] * (CC[3] * CC[8] - CC[5] * CC[6]) +
                 CC[2] * (CC[3] * CC[7] - CC[4] * CC[6]);
    double scale = fabs(CC[0]) + fabs(CC[4]) + fabs(CC[8]) + fabs(CC[1]) + fabs(CC[2]) + fabs(CC[3]) +
                   fabs(CC[5]) + fabs(CC[6]) + fabs(CC[7]);
    if (!(fabs(det) > 1e-18 * scale * scale * scale) || !isfinite(det))
        return -2; /* coplanar / coincident sample: no control-point basis */
    double id = 1. / det;
    CCi[0] = (CC[4] * CC[8] - CC[5] * CC[7]) * id;
    CCi[1] = (CC[2] * CC[7] - CC[1] * CC[8]) * id;
    CCi[2] = (CC[1] * CC[5] - CC[2] * CC[4]) * id;
    CCi[3] = (CC[5] * CC[6] - CC[3] * CC[8]) * id;
    CCi[4] = (CC[0] * CC[8] - CC[2] * CC[6]) * id;
    CCi[5] = (CC[2] * CC[3] - CC[0] * CC[5]) * id;
    CCi[6] = (CC[3] * CC[7] - CC[4] * CC[6]) * id;
    CCi[7] = (CC[1] * CC[6] - CC[0] * CC[7]) * id;
    CCi[8] = (CC[0] * CC[4] - CC[1] * CC[3]) * id;
    for (int i = 0; i < n; i++) {
        double d[3] = {pws[i][0] - cws[0][0], pws[i][1] - cws[0][1], pws[i][2] - cws[0][2]};
        for (int j = 0; j < 3; j++)
            alphas[i][1 + j] = CCi[3 * j] * d[0] + CCi[3 * j + 1] * d[1] + CCi[3 * j + 2] * d[2];
        alphas[i][0] = 1.0 - alphas[i][1] - alphas[i][2] - alphas[i][3];
    }
    /* M^T M */
    double M[2 * EPNP_MAXN][12], MtM[144], Ve[144], we[12];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 4; j++) {
            M[2 * i][3 * j] = alphas[i][j] * fu;
            M[2 * i][3 * j + 1] = 0;
            M[2 * i][3 * j + 2] = alphas[i][j] * (uc - us[i][0]);
            M[2 * i + 1][3 * j] = 0;
            M[2 * i + 1][3 * j + 1] = alphas[i][j] * fv;
            M[2 * i + 1][3 * j + 2] = alphas[i][j] * (vc - us[i][1]);
        }
    for (int a = 0; a < 12; a++)
        for (int b = 0; b < 12; b++) {
            double s = 0;
            for (int r = 0; r < 2 * n; r++)
                s += M[r][a] * M[r][b];
            MtM[12 * a + b] = s;
        }
    orc_jacobi_eigen_sym(12, MtM, Ve, we, 6); /* quadratic convergence: off-diagonals < 1e-12 after 6 */
    /* the four smallest eigenvalues, ascending (ties: lower index first) */
    int sel[4];
    char used[12] = {0};
    for (int k = 0; k < 4; k++) {
        int best = -1;
        for (int e = 0; e < 12; e++)
            if (!used[e] && (best < 0 || we[e] < we[best]))
                best = e;
        used[best] = 1;
        sel[k] = best;
    }
    double vbuf[4][12];
    const double *v[4];
    for (int k = 0; k < 4; k++) {
        for (int i = 0; i < 12; i++)
            vbuf[k][i] = Ve[12 * i + sel[k]];
        v[k] = vbuf[k];
    }
    /* L (6x10) and rho */
    static const int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {1, 2, 3, 2, 3, 3};
    double L[6][10], rho[6], dv[4][6][3];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 6; j++)
            for (int k = 0; k < 3; k++)
                dv[i][j][k] = v[i][3 * PA[j] + k] - v[i][3 * PB[j] + k];
    for (int i = 0; i < 6; i++) {
        L[i][0] = dot3(dv[0][i], dv[0][i]);
        L[i][1] = 2. * dot3(dv[0][i], dv[1][i]);
        L[i][2] = dot3(dv[1][i], dv[1][i]);
        L[i][3] = 2. * dot3(dv[0][i], dv[2][i]);
        L[i][4] = 2. * dot3(dv[1][i], dv[2][i]);
        L[i][5] = dot3(dv[2][i], dv[2][i]);
        L[i][6] = 2. * dot3(dv[0][i], dv[3][i]);
        L[i][7] = 2. * dot3(dv[1][i], dv[3][i]);
        L[i][8] = 2. * dot3(dv[2][i], dv[3][i]);
        L[i][9] = dot3(dv[3][i], dv[3][i]);
        double d[3] = {cws[PA[i]][0] - cws[PB[i]][0], cws[PA[i]][1] - cws[PB[i]][1], cws[PA[i]][2] - cws[PB[i]][2]};
        rho[i] = dot3(d, d);
    }
    double betas[3][4], Rs[3][9], ts[3][3], errs[3], pcs[EPNP_MAXN][3];
    /* approx 1: betas_approx = [B11 B12 B13 B14] */
    {
        double A[24], x[4];
        for (int i = 0; i < 6; i++) {
            A[4 * i] = L[i][0];
            A[4 * i + 1] = L[i][1];
            A[4 * i + 2] = L[i][3];
            A[4 * i + 3] = L[i][6];
        }
        lstsq_small(6, 4, A, rho, x);
        double *b = betas[0];
        if (x[0] < 0) {
            b[0] = sqrt(-x[0]);
            b[1] = -x[1] / b[0];
            b[2] = -x[2] / b[0];
            b[3] = -x[3] / b[0];
        } else {
            b[0] = sqrt(x[0]);
            b[1] = x[1] / b[0];
            b[2] = x[2] / b[0];
            b[3] = x[3] / b[0];
        }
    }
    /* approx 2: [B11 B12 B22] */
    {
        double A[18], x[3];
        for (int i = 0; i < 6; i++) {
            A[3 * i] = L[i][0];
            A[3 * i + 1] = L[i][1];
            A[3 * i + 2] = L[i][2];
        }
        lstsq_small(6, 3, A, rho, x);
        double *b = betas[1];
        if (x[0] < 0) {
            b[0] = sqrt(-x[0]);
            b[1] = x[2] < 0 ? sqrt(-x[2]) : 0.;
        } else {
            b[0] = sqrt(x[0]);
            b[1] = x[2] > 0 ? sqrt(x[2]) : 0.;
        }
        if (x[1] < 0)
            b[0] = -b[0];
        b[2] = 0.;
        b[3] = 0.;
    }
    /* approx 3: [B11 B12 B22 B13 B23] */
    {
        double A[30], x[5];
        for (int i = 0; i < 6; i++)
            for (int k = 0; k < 5; k++)
                A[5 * i + k] = L[i][k];
        lstsq_small(6, 5, A, rho, x);
        double *b = betas[2];
        if (x[0] < 0) {
            b[0] = sqrt(-x[0]);
            b[1] = x[2] < 0 ? sqrt(-x[2]) : 0.;
        } else {
            b[0] = sqrt(x[0]);
            b[1] = x[2] > 0 ? sqrt(x[2]) : 0.;
        }
        if (x[1] < 0)
            b[0] = -b[0];
        b[2] = x[3] / b[0];
        b[3] = 0.;
    }
    int best = -1;
    for (int a = 0; a < 3; a++) {
        gauss_newton_betas(L, rho, betas[a]);
        compute_rt_from_betas(v, betas[a], alphas, pws, n, Rs[a], ts[a], pcs);
        errs[a] = reprojection_error(Rs[a], ts[a], pws, us, n, K4);
        if (isfinite(errs[a]) && (best < 0 || errs[a] < errs[best]))
            best = a;
    }
    if (best < 0)
        return -3;
    memcpy(R, Rs[best], sizeof(double) * 9);
    memcpy(t, ts[best], sizeof(double) * 3);
    for (int i = 0; i < 9; i++)
        if (!isfinite(R[i]))
            return -3;
    return 0;
}

/* ---- reprojection error as PnPRansacCallback::computeError ------------------------------- */
static inline float reproj_err_sq(const double *R, const double *t, const double *K4, const float *X, const float *x)
{
    double Xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
    double Yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
    double Zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
    double z = Zc != 0 ? 1. / Zc : 1.;
    float px = (float)(Xc * z * K4[0] + K4[2]), py = (float)(Yc * z * K4[1] + K4[3]);
    float dx = x[0] - px, dy = x[1] - py;
    return (float)((double)dx * dx + (double)dy * dy);
}
float orc_reproj_err_sq(const double *R, const double *t, const double *K4, const float *X, const float *x)
{
    return reproj_err_sq(R, t, K4, X, x);
}

/* ---- Levenberg-Marquardt refinement of (R, t) over a point subset ------------------------ */
static int chol6_solve(const double *Ain, const double *b, double *x)
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

/* accumulates J^T J (upper+lower), J^T r and the squared error for the pose (R,t).
 *
 * Summation order = the HIP kernel's (pnp.hip: pnp_finish_kernel / block_reduce), so that the two agree BIT FOR BIT
 * (any fixed order is as good as upstream's sequential one, whose own order varies with the
 * OpenCV build): 256 strided partial sums (element e goes to partial e mod 256, elements in ascending order), the
 * partials of each group of 64 added in ascending order starting from 0, the four group totals as ((g0+g1)+g2)+g3. */
#define PNP_NACC 28 /* error, 21 upper-triangular entries of J^T J, 6 of J^T r */
static double pnp_normal_eq(const float *obj, const float *img, const int *idx, int m, const double *K4,
                            const double *R, const double *t, double *JtJ, double *Jtr)
{
    static _Thread_local double part[256][PNP_NACC];
    memset(part, 0, sizeof(part));
    for (int e = 0; e < m; e++) {
        double *acc = part[e & 255];
        const int i = idx ? idx[e] : e;
        const float *X = obj + 3 * i;
        double rx = R[0] * X[0] + R[1] * X[1] + R[2] * X[2];
        double ry = R[3] * X[0] + R[4] * X[1] + R[5] * X[2];
        double rz = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
        double Xc = rx + t[0], Yc = ry + t[1], Zc = rz + t[2];
        double iz = 1. / Zc;
        double u = K4[0] * Xc * iz + K4[2], v = K4[1] * Yc * iz + K4[3];
        double ru = u - img[2 * i], rv = v - img[2 * i + 1];
        acc[0] += ru * ru + rv * rv;
        if (!JtJ)
            continue;
        /* d(u,v)/d(Xc,Yc,Zc) */
        double a0 = K4[0] * iz, a2 = -K4[0] * Xc * iz * iz;
        double b1 = K4[1] * iz, b2 = -K4[1] * Yc * iz * iz;
        /* dXc/dw = -[R X]x, dXc/dt = I */
        double Ju[6] = {a2 * ry, a0 * rz - a2 * rx, -a0 * ry, a0, 0, a2};
        double Jv[6] = {-b1 * rz + b2 * ry, -b2 * rx, b1 * rx, 0, b1, b2};
        int k = 1;
        for (int p = 0; p < 6; p++)
            for (int q = p; q < 6; q++)
                acc[k++] += Ju[p] * Ju[q] + Jv[p] * Jv[q];
        for (int p = 0; p < 6; p++)
            acc[22 + p] += Ju[p] * ru + Jv[p] * rv;
    }
    double tot[PNP_NACC];
    const int nacc = JtJ ? PNP_NACC : 1;
    for (int k = 0; k < nacc; k++) {
        double g[4];
        for (int w = 0; w < 4; w++) {
            double x = 0;
            for (int j = 0; j < 64; j++)
                x += part[64 * w + j][k];
            g[w] = x;
        }
        tot[k] = ((g[0] + g[1]) + g[2]) + g[3];
    }
    if (JtJ) {
        int k = 1;
        for (int p = 0; p < 6; p++)
            for (int q = p; q < 6; q++) {
                JtJ[6 * p + q] = tot[k];
                JtJ[6 * q + p] = tot[k];
                k++;
            }
        for (int p = 0; p < 6; p++)
            Jtr[p] = tot[22 + p];
    }
    return tot[0];
}

double orc_pnp_refine_Rt(const float *obj, const float *img, const int *idx, int m, const double *K4,
                         double *R, double *t, int max_iters)
{
    double JtJ[36], Jtr[6], lambda = 1e-3;
    double err = pnp_normal_eq(obj, img, idx, m, K4, R, t, JtJ, Jtr);
    for (int it = 0; it < max_iters; it++) {
        double A[36], d[6];
        memcpy(A, JtJ, sizeof(A));
        for (int k = 0; k < 6; k++)
            A[7 * k] += lambda * JtJ[7 * k] + 1e-300;
        double nb[6];
        for (int k = 0; k < 6; k++)
            nb[k] = -Jtr[k];
        if (!chol6_solve(A, nb, d)) {
            lambda *= 10;
            if (lambda > 1e12)
                break;
            continue;
        }
        double dR[9], Rn[9], tn[3];
        orc_rodrigues(d, dR);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                Rn[3 * i + j] = dR[3 * i] * R[j] + dR[3 * i + 1] * R[3 + j] + dR[3 * i + 2] * R[6 + j];
        for (int k = 0; k < 3; k++)
            tn[k] = t[k] + d[3 + k];
        double e2 = pnp_normal_eq(obj, img, idx, m, K4, Rn, tn, 0, 0);
        if (e2 < err || !(err == err)) {
            memcpy(R, Rn, sizeof(Rn));
            memcpy(t, tn, sizeof(tn));
            double step = 0, scale = 0;
            for (int k = 0; k < 6; k++)
                step += d[k] * d[k];
            scale = 1. + t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
            double prev = err;
            err = pnp_normal_eq(obj, img, idx, m, K4, R, t, JtJ, Jtr);
            lambda *= 0.1;
            if (lambda < 1e-12)
                lambda = 1e-12;
            /* CvLevMarq's criterion as cv::solvePnP sets it (20 iterations, FLT_EPSILON): relative change of the
             * parameter vector, FLT_EPSILON^2 = 2^-46, or an error that no longer decreases */
            if (step <= 1.4210854715202004e-14 * scale || prev - err <= 1e-10 * prev)
                break;
        } else {
            lambda *= 10;
            if (lambda > 1e12)
                break;
        }
    }
    return sqrt(err / (m > 0 ? m : 1));
}

double orc_pnp_refine(const float *obj, const float *img, const int *idx, int m, const double *K4, double *rvec,
                      double *tvec, int max_iters)
{
    double R[9];
    orc_rodrigues(rvec, R);
    double rms = orc_pnp_refine_Rt(obj, img, idx, m, K4, R, tvec, max_iters);
    orc_rodrigues_inv(R, rvec);
    return rms;
}

/* hypothesis of RANSAC iteration `it`: 0 ok (R,t filled), <0 no model */
int orc_pnp_hypothesis(const float *obj, const float *img, int n, const double *K4, uint64_t seed, int it,
                       double *R, double *t)
{
    int idx[5];
    if (!orc_draw_subset_plain(seed, (uint32_t)it, n, 5, idx))
        return -1;
    double o[15], u[10];
    for (int k = 0; k < 5; k++) {
        for (int c = 0; c < 3; c++)
            o[3 * k + c] = obj[3 * idx[k] + c];
        u[2 * k] = img[2 * idx[k]];
        u[2 * k + 1] = img[2 * idx[k] + 1];
    }
    return orc_epnp(o, u, 5, K4, R, t) == 0 ? 0 : -2;
}

int orc_pnp_ransac(const float *obj, const float *img, int n, const double *K4, const orc_pnp_params *prm,
                   double *rvec, double *tvec, int *inliers, int *iters_run)
{
    const int M = 5;
    if (iters_run)
        *iters_run = 0;
    if (n < M)
        return 0;
    const float thr = (float)(prm->reproj_err * prm->reproj_err);
    int niters = prm->iterations, best_count = 0, it;
    double bestR[9], bestT[3];
    for (it = 0; it < niters; it++) {
        double R[9], t[3];
        int rc = orc_pnp_hypothesis(obj, img, n, K4, prm->seed, it, R, t);
        if (rc == -1)
            break;
        if (rc != 0)
            continue;
        int count = 0;
        /* an integer count: the order of the summands is free (cpu_baseline runs this over the host's threads) */
#pragma omp parallel for reduction(+ : count) schedule(static) if (n >= 2048)
        for (int i = 0; i < n; i++)
            count += reproj_err_sq(R, t, K4, obj + 3 * i, img + 2 * i) <= thr;
        if (count > (best_count > M - 1 ? best_count : M - 1)) {
            best_count = count;
            memcpy(bestR, R, sizeof(R));
            memcpy(bestT, t, sizeof(t));
            niters = orc_update_num_iters(prm->confidence, (double)(n - count) / n, M, niters);
        }
    }
    if (iters_run)
        *iters_run = it;
    if (best_count <= 0)
        return 0;
    int k = 0;
    for (int i = 0; i < n; i++)
        if (reproj_err_sq(bestR, bestT, K4, obj + 3 * i, img + 2 * i) <= thr)
            inliers[k++] = i;
    orc_pnp_refine_Rt(obj, img, inliers, k, K4, bestR, bestT, prm->refine_iters > 0 ? prm->refine_iters : 20);
    orc_rodrigues_inv(bestR, rvec);
    memcpy(tvec, bestT, sizeof(bestT));
    return k;
}

/* ---- cv::solvePnP(obj, img, K, dist = 0, rvec, tvec): SOLVEPNP_ITERATIVE, no extrinsic guess ------
 * The last rung of the reference's older ladder, src/bundleAdjust.cpp:470-477.  Upstream
 * (cvFindExtrinsicCameraParams2): planarity test on the eigenvalues of the centred second moments of
 * the object points (W[2] / W[1] < 1e-3 -> homography branch, NOT built here: returns -2); otherwise
 * DLT: L rows [X Y Z 1 0 0 0 0 xX xY xZ x], [0 0 0 0 X Y Z 1 yX yY yZ y] with (x, y) = -(normalised
 * image point); the eigenvector of the smallest eigenvalue of L^T L is [RR | tt] (3x4); sign from
 * det(RR); R = U V^T of RR's SVD; t = tt * |R| / |RR|; then Levenberg-Marquardt over all points
 * (here: orc_pnp_refine_Rt, the same minimiser).  Returns 0, -1 bad arguments / n < 6, -2 planar,
 * -3 degenerate.                                                                                   */
int orc_solve_pnp(const float *obj, const float *img, int n, const double *K4, double *rvec, double *tvec,
                  double *rms_out)
{
    if (!obj || !img || !K4 || !rvec || !tvec || n < 6)
        return -1;
    double S[4][10];
    memset(S, 0, sizeof(S));
    const double ifx = 1. / K4[0], ify = 1. / K4[1];
    for (int i = 0; i < n; i++) {
        const double P[4] = {obj[3 * i], obj[3 * i + 1], obj[3 * i + 2], 1.};
        const double x = -(((double)img[2 * i] - K4[2]) * ifx), y = -(((double)img[2 * i + 1] - K4[3]) * ify);
        const double w3 = x * x + y * y;
        int k = 0;
        for (int r = 0; r < 4; r++)
            for (int c = r; c < 4; c++) {
                const double pp = P[r] * P[c];
                S[0][k] += pp;
                S[1][k] += x * pp;
                S[2][k] += y * pp;
                S[3][k] += w3 * pp;
                k++;
            }
    }
    {
        const double inv_n = 1. / n, mx = S[0][3] * inv_n, my = S[0][6] * inv_n, mz = S[0][8] * inv_n;
        double C[9] = {S[0][0] - n * mx * mx, S[0][1] - n * mx * my, S[0][2] - n * mx * mz,
                       S[0][1] - n * mx * my, S[0][4] - n * my * my, S[0][5] - n * my * mz,
                       S[0][2] - n * mx * mz, S[0][5] - n * my * mz, S[0][7] - n * mz * mz};
        double U[9], V[9], w[3];
        svd3(C, U, V);
        for (int c = 0; c < 3; c++) {
            double q = 0;
            for (int r = 0; r < 3; r++) {
                const double e = C[3 * r] * V[c] + C[3 * r + 1] * V[3 + c] + C[3 * r + 2] * V[6 + c];
                q += e * e;
            }
            w[c] = sqrt(q);
        }
        if (!(w[1] > 0))
            return -2;
        if (w[2] / w[1] < 1e-3) {
            /* upstream's planar branch (cvFindExtrinsicCameraParams2): plane-aligned coordinates, homography by the
             * normalised DLT of findHomography(method 0), [h1 h2 h1 x h2] orthonormalised, t = 2 h3 / (|h1| + |h2|),
             * back through the plane transform; then the same refinement as the other branch.  (Upstream polishes
             * the homography with ten LM steps of its own first; the pose refinement minimises the same error.) */
            const double inv_n = 1. / n, mx = S[0][3] * inv_n, my = S[0][6] * inv_n, mz = S[0][8] * inv_n;
            double Rp[9], Tp[3];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++)
                    Rp[3 * r + c] = V[3 * c + r];
            if (Rp[2] * Rp[2] + Rp[5] * Rp[5] < 1e-10)
                for (int i = 0; i < 9; i++)
                    Rp[i] = (i % 4) == 0 ? 1. : 0.;
            const double detp = Rp[0] * (Rp[4] * Rp[8] - Rp[5] * Rp[7]) - Rp[1] * (Rp[3] * Rp[8] - Rp[5] * Rp[6]) +
                                Rp[2] * (Rp[3] * Rp[7] - Rp[4] * Rp[6]);
            if (detp < 0)
                for (int i = 0; i < 9; i++)
                    Rp[i] = -Rp[i];
            for (int r = 0; r < 3; r++)
                Tp[r] = -(Rp[3 * r] * mx + Rp[3 * r + 1] * my + Rp[3 * r + 2] * mz);
            double *q = (double *)malloc(sizeof(double) * 4 * (size_t)n);
            double c4[4] = {0, 0, 0, 0}, d4[4] = {0, 0, 0, 0}, sc4[4];
            for (int i = 0; i < n; i++) {
                const double px = obj[3 * i], py = obj[3 * i + 1], pz = obj[3 * i + 2];
                q[4 * i] = Rp[0] * px + Rp[1] * py + Rp[2] * pz + Tp[0];
                q[4 * i + 1] = Rp[3] * px + Rp[4] * py + Rp[5] * pz + Tp[1];
                q[4 * i + 2] = ((double)img[2 * i] - K4[2]) * ifx;
                q[4 * i + 3] = ((double)img[2 * i + 1] - K4[3]) * ify;
                for (int k = 0; k < 4; k++)
                    c4[k] += q[4 * i + k];
            }
            for (int k = 0; k < 4; k++)
                c4[k] *= inv_n;
            for (int i = 0; i < n; i++)
                for (int k = 0; k < 4; k++)
                    d4[k] += fabs(q[4 * i + k] - c4[k]);
            for (int k = 0; k < 4; k++) {
                if (!(d4[k] > DBL_EPSILON)) {
                    free(q);
                    return -2;
                }
                sc4[k] = n / d4[k];
            }
            double LtL[81], Vh9[81], w9[9];
            memset(LtL, 0, sizeof(LtL));
            for (int i = 0; i < n; i++) {
                const double X = (q[4 * i] - c4[0]) * sc4[0], Y = (q[4 * i + 1] - c4[1]) * sc4[1];
                const double u = (q[4 * i + 2] - c4[2]) * sc4[2], v = (q[4 * i + 3] - c4[3]) * sc4[3];
                const double Lx[9] = {X, Y, 1, 0, 0, 0, -u * X, -u * Y, -u};
                const double Ly[9] = {0, 0, 0, X, Y, 1, -v * X, -v * Y, -v};
                for (int r = 0; r < 9; r++)
                    for (int c = r; c < 9; c++)
                        LtL[9 * r + c] += Lx[r] * Lx[c] + Ly[r] * Ly[c];
            }
            free(q);
            for (int r = 0; r < 9; r++)
                for (int c = 0; c < r; c++)
                    LtL[9 * r + c] = LtL[9 * c + r];
            orc_jacobi_eigen_sym(9, LtL, Vh9, w9, 10);
            int best = 0;
            for (int e = 1; e < 9; e++)
                if (w9[e] < w9[best])
                    best = e;
            double H0[9], H1[9], H[9];
            for (int i = 0; i < 9; i++)
                H0[i] = Vh9[9 * i + best];
            const double Ti[9] = {1. / sc4[2], 0, c4[2], 0, 1. / sc4[3], c4[3], 0, 0, 1};
            const double Tm[9] = {sc4[0], 0, -c4[0] * sc4[0], 0, sc4[1], -c4[1] * sc4[1], 0, 0, 1};
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++)
                    H1[3 * r + c] = H0[3 * r] * Tm[c] + H0[3 * r + 1] * Tm[3 + c] + H0[3 * r + 2] * Tm[6 + c];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++)
                    H[3 * r + c] = Ti[3 * r] * H1[c] + Ti[3 * r + 1] * H1[3 + c] + Ti[3 * r + 2] * H1[6 + c];
            if (!(fabs(H[8]) > DBL_EPSILON))
                return -2;
            const double ih = 1. / H[8];
            for (int i = 0; i < 9; i++)
                H[i] *= ih;
            const double n1 = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]);
            const double n2 = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
            const double i1 = 1. / fmax(n1, DBL_EPSILON), i2 = 1. / fmax(n2, DBL_EPSILON), it = 2. / fmax(n1 + n2, DBL_EPSILON);
            const double h1[3] = {H[0] * i1, H[3] * i1, H[6] * i1}, h2[3] = {H[1] * i2, H[4] * i2, H[7] * i2};
            const double th[3] = {H[2] * it, H[5] * it, H[8] * it};
            const double h3[3] = {h1[1] * h2[2] - h1[2] * h2[1], h1[2] * h2[0] - h1[0] * h2[2], h1[0] * h2[1] - h1[1] * h2[0]};
            const double Hm[9] = {h1[0], h2[0], h3[0], h1[1], h2[1], h3[1], h1[2], h2[2], h3[2]};
            double Uh[9], Vh[9], Rh[9], R[9], t[3];
            svd3(Hm, Uh, Vh);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++)
                    Rh[3 * i + j] = Uh[3 * i] * Vh[3 * j] + Uh[3 * i + 1] * Vh[3 * j + 1] + Uh[3 * i + 2] * Vh[3 * j + 2];
            for (int i = 0; i < 3; i++) {
                t[i] = Rh[3 * i] * Tp[0] + Rh[3 * i + 1] * Tp[1] + Rh[3 * i + 2] * Tp[2] + th[i];
                for (int j = 0; j < 3; j++)
                    R[3 * i + j] = Rh[3 * i] * Rp[j] + Rh[3 * i + 1] * Rp[3 + j] + Rh[3 * i + 2] * Rp[6 + j];
            }
            for (int i = 0; i < 9; i++)
                if (!isfinite(R[i]))
                    return -2;
            const double rms = orc_pnp_refine_Rt(obj, img, 0, n, K4, R, t, 20);
            orc_rodrigues_inv(R, rvec);
            memcpy(tvec, t, sizeof(t));
            if (rms_out)
                *rms_out = rms;
            return 0;
        }
    }
    double A[144], V[144], w[12];
    for (int e = 0; e < 144; e++) {
        const int r = e / 12, c = e % 12, br = r >> 2, bc = c >> 2, i = r & 3, j = c & 3;
        const int lo = i < j ? i : j, hi = i < j ? j : i, k = lo * 4 - lo * (lo - 1) / 2 + (hi - lo);
        int blk = -1;
        if (br == bc)
            blk = br == 2 ? 3 : 0;
        else if (br + bc == 2)
            blk = 1;
        else if (br + bc == 3)
            blk = 2;
        A[e] = blk < 0 ? 0. : S[blk][k];
    }
    orc_jacobi_eigen_sym(12, A, V, w, 10);
    int best = 0;
    for (int e = 1; e < 12; e++)
        if (w[e] < w[best])
            best = e;
    double RR[9], tt[3];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++)
            RR[3 * r + c] = V[12 * (4 * r + c) + best];
        tt[r] = V[12 * (4 * r + 3) + best];
    }
    const double det = RR[0] * (RR[4] * RR[8] - RR[5] * RR[7]) - RR[1] * (RR[3] * RR[8] - RR[5] * RR[6]) +
                       RR[2] * (RR[3] * RR[7] - RR[4] * RR[6]);
    if (det < 0) {
        for (int i = 0; i < 9; i++)
            RR[i] = -RR[i];
        for (int i = 0; i < 3; i++)
            tt[i] = -tt[i];
    }
    double sc = 0;
    for (int i = 0; i < 9; i++)
        sc += RR[i] * RR[i];
    sc = sqrt(sc);
    if (!(sc > DBL_EPSILON))
        return -3;
    double U[9], Vs[9], R[9], t[3];
    svd3(RR, U, Vs);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[3 * i + j] = U[3 * i] * Vs[3 * j] + U[3 * i + 1] * Vs[3 * j + 1] + U[3 * i + 2] * Vs[3 * j + 2];
    for (int i = 0; i < 3; i++)
        t[i] = tt[i] * (sqrt(3.) / sc);
    const double rms = orc_pnp_refine_Rt(obj, img, 0, n, K4, R, t, 20);
    orc_rodrigues_inv(R, rvec);
    memcpy(tvec, t, sizeof(t));
    if (rms_out)
        *rms_out = rms;
    return 0;
}
