/*
 * oracle/posegraph.c -- CPU restatement of the reference's g2o pose graph
 * (TEST INFRASTRUCTURE; see svo_oracle.h.  PARITY UNPINNED.)
 *
 * Reference: /root/reference/include/poseGraph.h
 *   initializeGraph  :69-84   vertex 0 = identity, FIXED
 *   augmentNode      :87-111  new vertex (estimate = global pose) + edge(prev -> cur) whose
 *                             measurement is prev^-1 * cur taken from the CURRENT estimates
 *   addLoopClosure   :113-126 edge(prevVertex -> vertices[fromID]), measurement = IDENTITY
 *   globalOptimize   :128-138 initializeOptimization(); optimize(10)
 *   saveStructure    :140-179 VERTEX_SE3:QUAT / EDGE_SE3:QUAT text
 * Solver set-up (:29-30,52-54): Gauss-Newton, BlockSolver<6,6>, LinearSolverEigen (sparse
 * Cholesky); information matrices are left at the EdgeSE3 default (identity).
 *
 * g2o semantics restated: estimate X in SE3; error of edge (i -> j)
 * with measurement Z is e = toVectorMQT(Z^-1 * Xi^-1 * Xj) = [translation ; xyz of the unit
 * quaternion with w >= 0]; chi2 = e.e; update X <- X * fromVectorMQT(d) with
 * q(d) = (sqrt(1 - |d_q|^2), d_q) (identity rotation when |d_q| > 1, as g2o's
 * fromCompactQuaternion); Jacobians are the exact derivatives of e with respect to that
 * update; one Gauss-Newton step solves H d = -b, H = sum J^T J over all edges with the fixed
 * vertex removed; no damping, no robust kernel, exactly `iters` iterations.
 *
 * Poses are kept as (t, unit quaternion); g2o keeps Isometry3d matrices and converts, which
 * is the same map up to rounding.  The linear solve is a skyline (envelope) Cholesky in
 * vertex order -- an exact factorisation like Eigen's, only the elimination order differs.
 */
#include "svo_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* pose7 = tx ty tz qx qy qz qw */
static void q_mul(const double *a, const double *b, double *o) /* (x,y,z,w) */
{
    double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by - ax * bz + ay * bw + az * bx;
    o[2] = aw * bz + ax * by - ay * bx + az * bw;
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
}
static void q_rot(const double *q, const double *v, double *o)
{
    /* v' = v + 2 w (u x v) + 2 u x (u x v) */
    double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    double cx = uy * v[2] - uz * v[1], cy = uz * v[0] - ux * v[2], cz = ux * v[1] - uy * v[0];
    double dx = uy * cz - uz * cy, dy = uz * cx - ux * cz, dz = ux * cy - uy * cx;
    o[0] = v[0] + 2 * (w * cx + dx);
    o[1] = v[1] + 2 * (w * cy + dy);
    o[2] = v[2] + 2 * (w * cz + dz);
}
static void q_normalize(double *q)
{
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n > 0)
        for (int i = 0; i < 4; i++)
            q[i] /= n;
}
static void se3_mul(const double *A, const double *B, double *O)
{
    double t[3], q[4];
    q_rot(A + 3, B, t);
    for (int i = 0; i < 3; i++)
        t[i] += A[i];
    q_mul(A + 3, B + 3, q);
    memcpy(O, t, sizeof(t));
    memcpy(O + 3, q, sizeof(q));
}
static void se3_inv(const double *A, double *O)
{
    double qi[4] = {-A[3], -A[4], -A[5], A[6]}, t[3];
    q_rot(qi, A, t);
    for (int i = 0; i < 3; i++)
        O[i] = -t[i];
    memcpy(O + 3, qi, sizeof(qi));
}
static void q_to_R(const double *q, double *R)
{
    double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z);
    R[1] = 2 * (x * y - z * w);
    R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);
    R[4] = 1 - 2 * (x * x + z * z);
    R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);
    R[7] = 2 * (y * z + x * w);
    R[8] = 1 - 2 * (x * x + y * y);
}

void orc_se3_oplus(const double *X, const double *v, double *Xout)
{
    double d[7];
    d[0] = v[0];
    d[1] = v[1];
    d[2] = v[2];
    double w = 1. - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
    if (w < 0) { /* g2o fromCompactQuaternion: identity rotation */
        d[3] = d[4] = d[5] = 0;
        d[6] = 1;
    } else {
        d[3] = v[3];
        d[4] = v[4];
        d[5] = v[5];
        d[6] = sqrt(w);
    }
    se3_mul(X, d, Xout);
    q_normalize(Xout + 3);
}

/* e = toVectorMQT(Z^-1 Xi^-1 Xj); Ji = de/d(update of Xi), Jj = de/d(update of Xj), 6x6 row-major */
void orc_se3_edge_error(const double *Xi, const double *Xj, const double *Z, double *e, double *Ji, double *Jj)
{
    double A[7], Xi_inv[7], B[7], E[7];
    se3_inv(Z, A);
    se3_inv(Xi, Xi_inv);
    se3_mul(Xi_inv, Xj, B);
    se3_mul(A, B, E);
    double s = E[6] < 0 ? -1. : 1.;
    e[0] = E[0];
    e[1] = E[1];
    e[2] = E[2];
    e[3] = s * E[3];
    e[4] = s * E[4];
    e[5] = s * E[5];
    if (!Ji && !Jj)
        return;
    double Re[9], Ra[9];
    q_to_R(E + 3, Re);
    q_to_R(A + 3, Ra);
    if (Jj) {
        memset(Jj, 0, sizeof(double) * 36);
        /* translation rows: d te / d u = Re */
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++)
                Jj[6 * r + c] = Re[3 * r + c];
        /* quaternion rows: d vec(qe (x) (1,v)) / dv = w_e I + [u_e]x, times the sign */
        double ux = E[3], uy = E[4], uz = E[5], w = E[6];
        double Q[9] = {w, -uz, uy, uz, w, -ux, -uy, ux, w};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++)
                Jj[6 * (3 + r) + 3 + c] = s * Q[3 * r + c];
    }
    if (Ji) {
        memset(Ji, 0, sizeof(double) * 36);
        /* d te / d u = -Ra ;  d te / d v = 2 Ra [tb]x */
        double tb[3] = {B[0], B[1], B[2]};
        double Tx[9] = {0, -tb[2], tb[1], tb[2], 0, -tb[0], -tb[1], tb[0], 0};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                Ji[6 * r + c] = -Ra[3 * r + c];
                double sacc = 0;
                for (int k = 0; k < 3; k++)
                    sacc += Ra[3 * r + k] * Tx[3 * k + c];
                Ji[6 * r + 3 + c] = 2. * sacc;
            }
        /* d vec(qa (x) (1,-v) (x) qb) / dv = -[L(qa) R(qb)]_{xyz,xyz}  (w,x,y,z ordering) */
        double aw = A[6], ax = A[3], ay = A[4], az = A[5];
        double bw = B[6], bx = B[3], by = B[4], bz = B[5];
        double L[16] = {aw, -ax, -ay, -az, ax, aw, -az, ay, ay, az, aw, -ax, az, -ay, ax, aw};
        double Rm[16] = {bw, -bx, -by, -bz, bx, bw, bz, -by, by, -bz, bw, bx, bz, by, -bx, bw};
        for (int r = 1; r < 4; r++)
            for (int c = 1; c < 4; c++) {
                double sacc = 0;
                for (int k = 0; k < 4; k++)
                    sacc += L[4 * r + k] * Rm[4 * k + c];
                Ji[6 * (3 + r - 1) + 3 + c - 1] = -s * sacc;
            }
    }
}

/* ---- graph ------------------------------------------------------------------------------- */
struct orc_posegraph {
    int nv, ne, capv, cape;
    double *pose;  /* nv x 7 */
    int *from, *to;
    double *meas;  /* ne x 7 */
    int prev;      /* poseGraph.h prevVertex */
};

orc_posegraph *orc_pg_create(void)
{
    orc_posegraph *g = (orc_posegraph *)calloc(1, sizeof(*g));
    g->capv = g->cape = 64;
    g->pose = (double *)malloc(sizeof(double) * 7 * g->capv);
    g->from = (int *)malloc(sizeof(int) * g->cape);
    g->to = (int *)malloc(sizeof(int) * g->cape);
    g->meas = (double *)malloc(sizeof(double) * 7 * g->cape);
    g->prev = -1;
    return g;
}
void orc_pg_destroy(orc_posegraph *g)
{
    if (!g)
        return;
    free(g->pose);
    free(g->from);
    free(g->to);
    free(g->meas);
    free(g);
}
static void pg_add_vertex(orc_posegraph *g, const double *p)
{
    if (g->nv == g->capv) {
        g->capv *= 2;
        g->pose = (double *)realloc(g->pose, sizeof(double) * 7 * g->capv);
    }
    memcpy(g->pose + 7 * g->nv, p, sizeof(double) * 7);
    q_normalize(g->pose + 7 * g->nv + 3);
    g->nv++;
}
static void pg_add_edge(orc_posegraph *g, int from, int to, const double *z)
{
    if (g->ne == g->cape) {
        g->cape *= 2;
        g->from = (int *)realloc(g->from, sizeof(int) * g->cape);
        g->to = (int *)realloc(g->to, sizeof(int) * g->cape);
        g->meas = (double *)realloc(g->meas, sizeof(double) * 7 * g->cape);
    }
    g->from[g->ne] = from;
    g->to[g->ne] = to;
    memcpy(g->meas + 7 * g->ne, z, sizeof(double) * 7);
    g->ne++;
}
void orc_pg_initialize(orc_posegraph *g)
{
    const double id[7] = {0, 0, 0, 0, 0, 0, 1};
    g->nv = g->ne = 0;
    pg_add_vertex(g, id);
    g->prev = 0;
}
void orc_pg_augment_node(orc_posegraph *g, const double *pose7)
{
    pg_add_vertex(g, pose7);
    int cur = g->nv - 1;
    double inv[7], z[7];
    se3_inv(g->pose + 7 * g->prev, inv);
    se3_mul(inv, g->pose + 7 * cur, z); /* prev^-1 * cur from the current estimates */
    q_normalize(z + 3);
    pg_add_edge(g, g->prev, cur, z);
    g->prev = cur;
}
void orc_pg_add_loop_closure(orc_posegraph *g, int from_id)
{
    const double id[7] = {0, 0, 0, 0, 0, 0, 1};
    pg_add_edge(g, g->prev, from_id, id); /* edge(prevVertex -> vertices[fromID]), identity */
}
int orc_pg_num_vertices(const orc_posegraph *g) { return g->nv; }
int orc_pg_num_edges(const orc_posegraph *g) { return g->ne; }
void orc_pg_get_estimates(const orc_posegraph *g, double *out) { memcpy(out, g->pose, sizeof(double) * 7 * g->nv); }
void orc_pg_get_edge(const orc_posegraph *g, int e, int *from, int *to, double *meas7)
{
    *from = g->from[e];
    *to = g->to[e];
    memcpy(meas7, g->meas + 7 * e, sizeof(double) * 7);
}

/* ---- Gauss-Newton with a skyline Cholesky --------------------------------------------------- */
/* unknown block of vertex v (v >= 1) is v-1; scalar row r = 6*(v-1)+k; env[r] = first stored column */
int orc_pg_optimize(orc_posegraph *g, int iters, double *chi2_out)
{
    const int nb = g->nv - 1, n = 6 * nb;
    if (nb <= 0)
        return 0;
    int *benv = (int *)malloc(sizeof(int) * nb); /* block envelope start per block row */
    for (int b = 0; b < nb; b++)
        benv[b] = b;
    for (int e = 0; e < g->ne; e++) {
        int i = g->from[e] - 1, j = g->to[e] - 1;
        if (i < 0 || j < 0)
            continue;
        int hi = i > j ? i : j, lo = i > j ? j : i;
        if (lo < benv[hi])
            benv[hi] = lo;
    }
    size_t *rowoff = (size_t *)malloc(sizeof(size_t) * (n + 1));
    int *env = (int *)malloc(sizeof(int) * n);
    size_t tot = 0;
    for (int r = 0; r < n; r++) {
        env[r] = 6 * benv[r / 6];
        rowoff[r] = tot;
        tot += (size_t)(r - env[r] + 1);
    }
    rowoff[n] = tot;
    double *Hs = (double *)malloc(sizeof(double) * tot), *b = (double *)malloc(sizeof(double) * n),
           *dx = (double *)malloc(sizeof(double) * n);
#define HS(r, c) Hs[rowoff[r] + (size_t)((c) - env[r])]
    for (int it = 0; it <= iters; it++) {
        memset(Hs, 0, sizeof(double) * tot);
        memset(b, 0, sizeof(double) * n);
        double chi2 = 0;
        for (int e = 0; e < g->ne; e++) {
            double err[6], Ji[36], Jj[36];
            int vi = g->from[e], vj = g->to[e];
            orc_se3_edge_error(g->pose + 7 * vi, g->pose + 7 * vj, g->meas + 7 * e, err, Ji, Jj);
            for (int k = 0; k < 6; k++)
                chi2 += err[k] * err[k];
            const double *J[2] = {Ji, Jj};
            int blk[2] = {vi - 1, vj - 1};
            for (int a = 0; a < 2; a++) {
                if (blk[a] < 0)
                    continue; /* fixed vertex 0 */
                for (int p = 0; p < 6; p++) {
                    double s = 0;
                    for (int k = 0; k < 6; k++)
                        s += J[a][6 * k + p] * err[k];
                    b[6 * blk[a] + p] += s;
                }
                for (int c = 0; c < 2; c++) {
                    if (blk[c] < 0)
                        continue;
                    for (int p = 0; p < 6; p++)
                        for (int q = 0; q < 6; q++) {
                            int r = 6 * blk[a] + p, cc = 6 * blk[c] + q;
                            if (cc > r)
                                continue; /* lower triangle only */
                            double s = 0;
                            for (int k = 0; k < 6; k++)
                                s += J[a][6 * k + p] * J[c][6 * k + q];
                            HS(r, cc) += s;
                        }
                }
            }
        }
        if (chi2_out)
            chi2_out[it] = chi2;
        if (it == iters)
            break;
        /* skyline Cholesky H = L L^T (row by row), in place */
        int ok = 1;
        for (int r = 0; r < n && ok; r++) {
            for (int c = env[r]; c <= r; c++) {
                int k0 = env[r] > env[c] ? env[r] : env[c];
                double s = HS(r, c);
                for (int k = k0; k < c; k++)
                    s -= HS(r, k) * HS(c, k);
                if (c == r) {
                    if (!(s > 0)) {
                        ok = 0;
                        break;
                    }
                    HS(r, r) = sqrt(s);
                } else
                    HS(r, c) = s / HS(c, c);
            }
        }
        if (!ok)
            break;
        /* L y = -b ; L^T dx = y */
        for (int r = 0; r < n; r++) {
            double s = -b[r];
            for (int k = env[r]; k < r; k++)
                s -= HS(r, k) * dx[k];
            dx[r] = s / HS(r, r);
        }
        for (int r = n - 1; r >= 0; r--) {
            dx[r] /= HS(r, r);
            for (int k = env[r]; k < r; k++)
                dx[k] -= HS(r, k) * dx[r];
        }
        for (int v = 1; v < g->nv; v++) {
            double out[7];
            orc_se3_oplus(g->pose + 7 * v, dx + 6 * (v - 1), out);
            memcpy(g->pose + 7 * v, out, sizeof(out));
        }
    }
#undef HS
    free(benv);
    free(rowoff);
    free(env);
    free(Hs);
    free(b);
    free(dx);
    return 0;
}

int orc_pg_write_g2o(const orc_posegraph *g, const char *path)
{
    FILE *f = fopen(path, "w");
    if (!f)
        return -1;
    for (int v = 0; v < g->nv; v++) {
        const double *p = g->pose + 7 * v;
        fprintf(f, "VERTEX_SE3:QUAT %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", v, p[0], p[1], p[2], p[3],
                p[4], p[5], p[6]);
    }
    for (int e = 0; e < g->ne; e++) {
        const double *z = g->meas + 7 * e;
        fprintf(f, "EDGE_SE3:QUAT %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g", g->from[e], g->to[e], z[0], z[1],
                z[2], z[3], z[4], z[5], z[6]);
        for (int i = 0; i < 6; i++)
            for (int j = i; j < 6; j++)
                fprintf(f, " %d", i == j ? 1 : 0);
        fprintf(f, "\n");
    }
    fclose(f);
    return 0;
}
