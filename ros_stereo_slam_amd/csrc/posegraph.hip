// posegraph.hip -- SE3 pose-graph Gauss-Newton for gfx950.
//
// Replaces the g2o wrapper globalPoseGraph of include/poseGraph.h:
//   initializeGraph :69-84, augmentNode :87-111, addLoopClosure :113-126,
//   globalOptimize :128-138 (Gauss-Newton, 10 iterations, BlockSolver<6,6> + sparse
//   Cholesky, identity information), saveStructure :140-179.
//
// One Gauss-Newton iteration is four launches on the context's stream (f64 throughout):
//   linearize  one thread per edge: error e = toVectorMQT(Z^-1 Xi^-1 Xj), analytic 6x6
//              Jacobians, and the edge's blocks Ji^T Ji, Ji^T Jj, Jj^T Jj, Ji^T e, Jj^T e
//              written to per-edge slots (the "per-edge 6x6 Jacobian blocks");
//   assemble   one thread per vertex: its diagonal block and gradient are summed from its
//              incident edges in edge order, its off-diagonal blocks are dropped into a
//              block-skyline store -- a gather, so the sum order is fixed and the result
//              reproducible (no float atomics);
//   solve      H dx = -b by a block-skyline Cholesky in vertex order.  The graph is a chain
//              (block tridiagonal) plus one long row per loop closure; eliminating in time
//              order keeps all fill inside those rows.  The recurrence along the chain is
//              inherently serial (latency-, not bandwidth-bound: ~3 MB per iteration), so one
//              wavefront walks it with the 36 lanes of a 6x6 block working in parallel;
//   update     one thread per vertex: X <- X * fromVectorMQT(dx).
#include <cmath>
#include <vector>

#include "svo_internal.h"

namespace {

// ---- SE3 with unit quaternions: pose7 = tx ty tz qx qy qz qw -----------------------------------
__host__ __device__ inline void q_mul(const double *a, const double *b, double *o)
{
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by - ax * bz + ay * bw + az * bx;
    o[2] = aw * bz + ax * by - ay * bx + az * bw;
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
}
__host__ __device__ inline void q_rot(const double *q, const double *v, double *o)
{
    const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    const double cx = uy * v[2] - uz * v[1], cy = uz * v[0] - ux * v[2], cz = ux * v[1] - uy * v[0];
    const double dx = uy * cz - uz * cy, dy = uz * cx - ux * cz, dz = ux * cy - uy * cx;
    o[0] = v[0] + 2 * (w * cx + dx);
    o[1] = v[1] + 2 * (w * cy + dy);
    o[2] = v[2] + 2 * (w * cz + dz);
}
__host__ __device__ inline void q_normalize(double *q)
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n > 0)
        for (int i = 0; i < 4; i++)
            q[i] /= n;
}
__host__ __device__ inline void se3_mul(const double *A, const double *B, double *O)
{
    double t[3], q[4];
    q_rot(A + 3, B, t);
    for (int i = 0; i < 3; i++)
        t[i] += A[i];
    q_mul(A + 3, B + 3, q);
    for (int i = 0; i < 3; i++)
        O[i] = t[i];
    for (int i = 0; i < 4; i++)
        O[3 + i] = q[i];
}
__host__ __device__ inline void se3_inv(const double *A, double *O)
{
    const double qi[4] = {-A[3], -A[4], -A[5], A[6]};
    double t[3];
    q_rot(qi, A, t);
    for (int i = 0; i < 3; i++)
        O[i] = -t[i];
    for (int i = 0; i < 4; i++)
        O[3 + i] = qi[i];
}
__device__ inline void q_to_R(const double *q, double *R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
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

// per-edge output slot: 3 blocks + 2 gradients + chi2
struct EdgeOut {
    double Hii[36], Hij[36], Hjj[36], bi[6], bj[6], chi2, pad;
};

__global__ __launch_bounds__(128) void pg_linearize_kernel(const double *__restrict__ pose, const int *__restrict__ efrom,
                                                           const int *__restrict__ eto, const double *__restrict__ meas,
                                                           int ne, EdgeOut *__restrict__ out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne)
        return;
    const double *Xi = pose + 7 * efrom[e], *Xj = pose + 7 * eto[e], *Z = meas + 7 * e;
    double A[7], Xi_inv[7], B[7], E[7];
    se3_inv(Z, A);
    se3_inv(Xi, Xi_inv);
    se3_mul(Xi_inv, Xj, B);
    se3_mul(A, B, E);
    const double s = E[6] < 0 ? -1. : 1.;
    const double err[6] = {E[0], E[1], E[2], s * E[3], s * E[4], s * E[5]};
    double Ji[36], Jj[36], Re[9], Ra[9];
#pragma unroll
    for (int k = 0; k < 36; k++) {
        Ji[k] = 0;
        Jj[k] = 0;
    }
    q_to_R(E + 3, Re);
    q_to_R(A + 3, Ra);
    {  // Xj <- Xj * D:  d te/du = Re,  d vec(qe)/dv = w I + [u]x
        const double ux = E[3], uy = E[4], uz = E[5], w = E[6];
        const double Q[9] = {w, -uz, uy, uz, w, -ux, -uy, ux, w};
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                Jj[6 * r + c] = Re[3 * r + c];
                Jj[6 * (3 + r) + 3 + c] = s * Q[3 * r + c];
            }
    }
    {  // Xi <- Xi * D:  d te/du = -Ra,  d te/dv = 2 Ra [tb]x,  d vec/dv = -[L(qa) R(qb)]_xyz
        const double tb[3] = {B[0], B[1], B[2]};
        const double Tx[9] = {0, -tb[2], tb[1], tb[2], 0, -tb[0], -tb[1], tb[0], 0};
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                Ji[6 * r + c] = -Ra[3 * r + c];
                double acc = 0;
#pragma unroll
                for (int k = 0; k < 3; k++)
                    acc += Ra[3 * r + k] * Tx[3 * k + c];
                Ji[6 * r + 3 + c] = 2. * acc;
            }
        const double aw = A[6], ax = A[3], ay = A[4], az = A[5];
        const double bw = B[6], bx = B[3], by = B[4], bz = B[5];
        const double L[16] = {aw, -ax, -ay, -az, ax, aw, -az, ay, ay, az, aw, -ax, az, -ay, ax, aw};
        const double Rm[16] = {bw, -bx, -by, -bz, bx, bw, bz, -by, by, -bz, bw, bx, bz, by, -bx, bw};
#pragma unroll
        for (int r = 1; r < 4; r++)
#pragma unroll
            for (int c = 1; c < 4; c++) {
                double acc = 0;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    acc += L[4 * r + k] * Rm[4 * k + c];
                Ji[6 * (3 + r - 1) + 3 + c - 1] = -s * acc;
            }
    }
    EdgeOut &o = out[e];
    double chi = 0;
#pragma unroll
    for (int k = 0; k < 6; k++)
        chi += err[k] * err[k];
    o.chi2 = chi;
#pragma unroll
    for (int p = 0; p < 6; p++) {
        double si = 0, sj = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            si += Ji[6 * k + p] * err[k];
            sj += Jj[6 * k + p] * err[k];
        }
        o.bi[p] = si;
        o.bj[p] = sj;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            double a = 0, b = 0, c = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                a += Ji[6 * k + p] * Ji[6 * k + q];
                b += Ji[6 * k + p] * Jj[6 * k + q];
                c += Jj[6 * k + p] * Jj[6 * k + q];
            }
            o.Hii[6 * p + q] = a;
            o.Hij[6 * p + q] = b;
            o.Hjj[6 * p + q] = c;
        }
    }
}

// Block-skyline store.  Unknown block b = vertex b+1 (vertex 0 is fixed).  Row b keeps its
// off-diagonal blocks for columns benv[b] .. b-1 at Ls[(rowptr[b] + c - benv[b]) * 36], the
// diagonal blocks live in Ld, the gradient in rhs.
//
// assemble: incident edges of vertex v are inc[incptr[v] .. incptr[v+1]) = edge*2 + role
// (role 0: v is the edge's `from`, 1: `to`), ascending in edge index.
__global__ __launch_bounds__(128) void pg_assemble_kernel(int nb, const int *__restrict__ incptr,
                                                          const int *__restrict__ inc, const int *__restrict__ efrom,
                                                          const int *__restrict__ eto, const EdgeOut *__restrict__ eo,
                                                          const int *__restrict__ benv, const int *__restrict__ rowptr,
                                                          double *__restrict__ Ld, double *__restrict__ Ls,
                                                          double *__restrict__ rhs)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb)
        return;
    const int v = b + 1;
    double D[36], g[6];
#pragma unroll
    for (int k = 0; k < 36; k++)
        D[k] = 0;
#pragma unroll
    for (int k = 0; k < 6; k++)
        g[k] = 0;
    for (int t = incptr[v]; t < incptr[v + 1]; t++) {
        const int e = inc[t] >> 1, role = inc[t] & 1;
        const EdgeOut &o = eo[e];
        const double *Hd = role ? o.Hjj : o.Hii, *gd = role ? o.bj : o.bi;
#pragma unroll
        for (int k = 0; k < 36; k++)
            D[k] += Hd[k];
#pragma unroll
        for (int k = 0; k < 6; k++)
            g[k] += gd[k];
        // off-diagonal block (v, other) goes to the row of the larger vertex
        const int other = role ? efrom[e] : eto[e];
        if (other >= 1 && other < v) {
            // block(v, other) = J_v^T J_other : role 1 (v = to): (Ji^T Jj)^T ; role 0 (v = from): Ji^T Jj
            double *dst = Ls + (size_t)(rowptr[b] + (other - 1) - benv[b]) * 36;
#pragma unroll
            for (int p = 0; p < 6; p++)
#pragma unroll
                for (int q = 0; q < 6; q++)
                    dst[6 * p + q] += role ? o.Hij[6 * q + p] : o.Hij[6 * p + q];
        }
    }
#pragma unroll
    for (int k = 0; k < 36; k++)
        Ld[(size_t)b * 36 + k] = D[k];
#pragma unroll
    for (int k = 0; k < 6; k++)
        rhs[(size_t)b * 6 + k] = g[k];
}

// loads / stores of data this kernel itself produces go around the vector L1 (agent scope)
__device__ __forceinline__ double ldc(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stc(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wave_sync_mem()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// status[0] = 0 ok, else 1 + the block row whose pivot was not positive
__global__ __launch_bounds__(64) void pg_solve_kernel(int nb, const int *__restrict__ benv,
                                                      const int *__restrict__ rowptr, double *Ld, double *Linv,
                                                      double *Ls, const double *__restrict__ rhs, double *y,
                                                      int *__restrict__ status)
{
    __shared__ double sS[36], sL[36], sLi[36], sv[8], sw[40];
    const int lane = threadIdx.x;
    const int i = lane / 6, j = lane - 6 * i;
    const bool act = lane < 36;
    bool failed = false;
    // ---- factorisation + forward substitution, row by row ----
    for (int b = 0; b < nb && !failed; b++) {
        const int env = benv[b], rp = rowptr[b];
        for (int c = env; c < b; c++) {
            double acc = act ? ldc(Ls + (size_t)(rp + c - env) * 36 + lane) : 0.;
            const int envc = benv[c], k0 = env > envc ? env : envc;
            for (int k = k0; k < c; k++) {
                if (act) {
                    const double *Ab = Ls + (size_t)(rp + k - env) * 36 + 6 * i;
                    const double *Bb = Ls + (size_t)(rowptr[c] + k - envc) * 36 + 6 * j;
#pragma unroll
                    for (int l = 0; l < 6; l++)
                        acc -= ldc(Ab + l) * ldc(Bb + l);
                }
            }
            if (act)
                sS[lane] = acc;
            wave_sync_mem();
            if (act) {  // L[b][c] = S * Linv[c]^T
                double x = 0;
                const double *Li = Linv + (size_t)c * 36 + 6 * j;
#pragma unroll
                for (int l = 0; l < 6; l++)
                    x += sS[6 * i + l] * ldc(Li + l);
                stc(Ls + (size_t)(rp + c - env) * 36 + lane, x);
            }
            wave_sync_mem();
        }
        // diagonal block: S = H[b][b] - sum_k L[b][k] L[b][k]^T
        double acc = act ? Ld[(size_t)b * 36 + lane] : 0.;
        for (int k = env; k < b; k++) {
            if (act) {
                const double *Ab = Ls + (size_t)(rp + k - env) * 36;
#pragma unroll
                for (int l = 0; l < 6; l++)
                    acc -= ldc(Ab + 6 * i + l) * ldc(Ab + 6 * j + l);
            }
        }
        if (act)
            sS[lane] = acc;
        wave_sync_mem();
        if (lane == 0) {  // 6x6 Cholesky and the inverse of its factor, serially (tiny)
            bool ok = true;
            for (int r = 0; r < 6; r++)
                for (int c = 0; c <= r; c++) {
                    double s = sS[6 * r + c];
                    for (int k = 0; k < c; k++)
                        s -= sL[6 * r + k] * sL[6 * c + k];
                    if (r == c) {
                        if (!(s > 0))
                            ok = false;
                        sL[6 * r + r] = sqrt(s);
                    } else
                        sL[6 * r + c] = s / sL[6 * c + c];
                }
            for (int r = 0; r < 6; r++)
                for (int c = r + 1; c < 6; c++)
                    sL[6 * r + c] = 0.;
            for (int c = 0; c < 6; c++) {  // column c of L^-1 by forward substitution
                for (int r = 0; r < 6; r++) {
                    double s = r == c ? 1. : 0.;
                    for (int k = c; k < r; k++)
                        s -= sL[6 * r + k] * sLi[6 * k + c];
                    sLi[6 * r + c] = r < c ? 0. : s / sL[6 * r + r];
                }
            }
            sv[6] = ok ? 0. : 1.;
        }
        wave_sync_mem();
        if (sv[6] != 0.) {
            if (lane == 0)
                status[0] = 1 + b;
            failed = true;
            break;
        }
        if (act) {
            stc(Ld + (size_t)b * 36 + lane, sL[lane]);
            stc(Linv + (size_t)b * 36 + lane, sLi[lane]);
        }
        // forward substitution: y_b = Linv_b * (-rhs_b - sum_c L[b][c] y_c)
        double part = 0;  // lane (i, l=j): sum over c of L[b][c][i][l] * y_c[l]
        for (int c = env; c < b; c++)
            if (act)
                part += ldc(Ls + (size_t)(rp + c - env) * 36 + lane) * ldc(y + (size_t)c * 6 + j);
        if (act)
            sw[lane] = part;
        wave_sync_mem();
        if (lane < 6) {
            double s = -rhs[(size_t)b * 6 + lane];
            for (int l = 0; l < 6; l++)
                s -= sw[6 * lane + l];
            sv[lane] = s;
        }
        wave_sync_mem();
        if (lane < 6) {
            double s = 0;
            for (int l = 0; l <= lane; l++)
                s += sLi[6 * lane + l] * sv[l];
            stc(y + (size_t)b * 6 + lane, s);
        }
        wave_sync_mem();
    }
    if (failed)
        return;
    if (lane == 0)
        status[0] = 0;
    // ---- back substitution: L^T x = y, rows in descending order, in place in y ----
    for (int b = nb - 1; b >= 0; b--) {
        const int env = benv[b], rp = rowptr[b];
        if (lane < 6) {  // x_b = Linv_b^T y_b
            double s = 0;
            for (int l = lane; l < 6; l++)
                s += ldc(Linv + (size_t)b * 36 + 6 * l + lane) * ldc(y + (size_t)b * 6 + l);
            sv[lane] = s;
        }
        wave_sync_mem();
        if (lane < 6)
            stc(y + (size_t)b * 6 + lane, sv[lane]);
        // y_c -= L[b][c]^T x_b for the columns of this row; lanes (c-stripe, component)
        for (int c0 = env; c0 < b; c0 += 10) {
            const int c = c0 + lane / 6, l = lane - 6 * (lane / 6);
            if (lane < 60 && c < b) {
                const double *Lb = Ls + (size_t)(rp + c - env) * 36;
                double s = 0;
                for (int r = 0; r < 6; r++)
                    s += ldc(Lb + 6 * r + l) * sv[r];
                stc(y + (size_t)c * 6 + l, ldc(y + (size_t)c * 6 + l) - s);
            }
        }
        wave_sync_mem();
    }
}

__global__ __launch_bounds__(128) void pg_update_kernel(int nv, double *__restrict__ pose, const double *__restrict__ dx)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (v >= nv)
        return;
    const double *d = dx + (size_t)(v - 1) * 6;
    double D[7] = {d[0], d[1], d[2], 0, 0, 0, 1}, O[7];
    const double w = 1. - (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    if (w >= 0) {  // else: identity rotation (g2o fromCompactQuaternion)
        D[3] = d[3];
        D[4] = d[4];
        D[5] = d[5];
        D[6] = sqrt(w);
    }
    se3_mul(pose + 7 * v, D, O);
    q_normalize(O + 3);
    for (int k = 0; k < 7; k++)
        pose[7 * v + k] = O[k];
}

__global__ __launch_bounds__(256) void pg_chi2_kernel(const EdgeOut *__restrict__ eo, int ne, double *__restrict__ out)
{
    __shared__ double s_p[256];
    double s = 0;
    for (int e = threadIdx.x; e < ne; e += 256)
        s += eo[e].chi2;
    s_p[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off)
            s_p[threadIdx.x] += s_p[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        *out = s_p[0];
}

}  // namespace

struct svo_posegraph {
    svo_ctx *ctx = nullptr;
    std::vector<double> pose;  // 7 per vertex
    std::vector<int> efrom, eto;
    std::vector<double> meas;  // 7 per edge
    int prev = -1;
    DevBuf d_pose, d_from, d_to, d_meas, d_eo, d_incptr, d_inc, d_benv, d_rowptr, d_Ld, d_Linv, d_Ls, d_rhs, d_y,
        d_misc;
    int nv() const { return (int)(pose.size() / 7); }
    int ne() const { return (int)efrom.size(); }
};

extern "C" {

int svo_pg_create(svo_ctx *ctx, svo_posegraph **out)
{
    SVO_CHECK_ARG(ctx && out);
    svo_posegraph *g = new svo_posegraph();
    g->ctx = ctx;
    *out = g;
    return svo_pg_initialize(g);
}

int svo_pg_destroy(svo_posegraph *g)
{
    if (!g)
        return SVO_OK;
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    DevBuf *bufs[] = {&g->d_pose, &g->d_from, &g->d_to,   &g->d_meas, &g->d_eo,  &g->d_incptr, &g->d_inc, &g->d_benv,
                      &g->d_rowptr, &g->d_Ld, &g->d_Linv, &g->d_Ls,   &g->d_rhs, &g->d_y,      &g->d_misc};
    for (DevBuf *b : bufs)
        b->release();
    delete g;
    return SVO_OK;
}

int svo_pg_initialize(svo_posegraph *g)
{
    SVO_CHECK_ARG(g);
    g->pose.assign({0, 0, 0, 0, 0, 0, 1});  // vertex 0: identity, fixed (poseGraph.h:69-84)
    g->efrom.clear();
    g->eto.clear();
    g->meas.clear();
    g->prev = 0;
    return SVO_OK;
}

int svo_pg_augment_node(svo_posegraph *g, const double *pose7)
{
    SVO_CHECK_ARG(g && pose7 && g->prev >= 0);
    double p[7];
    memcpy(p, pose7, sizeof(p));
    q_normalize(p + 3);
    g->pose.insert(g->pose.end(), p, p + 7);
    const int cur = g->nv() - 1;
    // measurement = prev^-1 * cur from the CURRENT estimates (poseGraph.h:97,102)
    double inv[7], z[7];
    se3_inv(&g->pose[7 * g->prev], inv);
    se3_mul(inv, &g->pose[7 * cur], z);
    q_normalize(z + 3);
    g->efrom.push_back(g->prev);
    g->eto.push_back(cur);
    g->meas.insert(g->meas.end(), z, z + 7);
    g->prev = cur;
    return SVO_OK;
}

int svo_pg_add_loop_closure(svo_posegraph *g, int from_id)
{
    SVO_CHECK_ARG(g && from_id >= 0 && from_id < g->nv() && g->prev >= 0);
    const double id[7] = {0, 0, 0, 0, 0, 0, 1};  // poseGraph.h:118-121: identity measurement
    g->efrom.push_back(g->prev);
    g->eto.push_back(from_id);
    g->meas.insert(g->meas.end(), id, id + 7);
    return SVO_OK;
}

int svo_pg_num_vertices(const svo_posegraph *g) { return g ? g->nv() : 0; }
int svo_pg_num_edges(const svo_posegraph *g) { return g ? g->ne() : 0; }

int svo_pg_get_estimates(const svo_posegraph *g, double *pose7_out)
{
    SVO_CHECK_ARG(g && pose7_out);
    memcpy(pose7_out, g->pose.data(), g->pose.size() * sizeof(double));
    return SVO_OK;
}

int svo_pg_get_edge(const svo_posegraph *g, int e, int *from, int *to, double *meas7)
{
    SVO_CHECK_ARG(g && e >= 0 && e < g->ne());
    if (from)
        *from = g->efrom[e];
    if (to)
        *to = g->eto[e];
    if (meas7)
        memcpy(meas7, &g->meas[7 * e], 7 * sizeof(double));
    return SVO_OK;
}

int svo_pg_optimize(svo_posegraph *g, int iters, double *chi2)
{
    SVO_CHECK_ARG(g && iters >= 0);
    svo_ctx *ctx = g->ctx;
    SVO_HIP(hipSetDevice(ctx->device));
    const int nv = g->nv(), ne = g->ne(), nb = nv - 1;
    if (nb <= 0 || ne == 0) {
        if (chi2)
            for (int i = 0; i <= iters; i++)
                chi2[i] = 0;
        return SVO_OK;
    }
    // ---- structure (host): incidence lists, block envelope, skyline offsets ----
    std::vector<int> incptr(nv + 1, 0), inc(2 * ne), benv(nb), rowptr(nb + 1);
    for (int e = 0; e < ne; e++) {
        incptr[g->efrom[e] + 1]++;
        incptr[g->eto[e] + 1]++;
    }
    for (int v = 0; v < nv; v++)
        incptr[v + 1] += incptr[v];
    {
        std::vector<int> fill(incptr.begin(), incptr.end() - 1);
        for (int e = 0; e < ne; e++) {  // ascending edge index within each vertex
            inc[fill[g->efrom[e]]++] = 2 * e;
            inc[fill[g->eto[e]]++] = 2 * e + 1;
        }
    }
    for (int b = 0; b < nb; b++)
        benv[b] = b;
    for (int e = 0; e < ne; e++) {
        const int i = g->efrom[e] - 1, j = g->eto[e] - 1;
        if (i < 0 || j < 0 || i == j)
            continue;
        const int hi = i > j ? i : j, lo = i > j ? j : i;
        if (lo < benv[hi])
            benv[hi] = lo;
    }
    rowptr[0] = 0;
    for (int b = 0; b < nb; b++)
        rowptr[b + 1] = rowptr[b] + (b - benv[b]);
    const size_t nblk = (size_t)rowptr[nb];
    int rc;
    if ((rc = g->d_pose.ensure((size_t)nv * 56)) || (rc = g->d_from.ensure((size_t)ne * 4)) ||
        (rc = g->d_to.ensure((size_t)ne * 4)) || (rc = g->d_meas.ensure((size_t)ne * 56)) ||
        (rc = g->d_eo.ensure((size_t)ne * sizeof(EdgeOut))) || (rc = g->d_incptr.ensure((size_t)(nv + 1) * 4)) ||
        (rc = g->d_inc.ensure((size_t)2 * ne * 4)) || (rc = g->d_benv.ensure((size_t)nb * 4)) ||
        (rc = g->d_rowptr.ensure((size_t)(nb + 1) * 4)) || (rc = g->d_Ld.ensure((size_t)nb * 288)) ||
        (rc = g->d_Linv.ensure((size_t)nb * 288)) || (rc = g->d_Ls.ensure((nblk + 1) * 288)) ||
        (rc = g->d_rhs.ensure((size_t)nb * 48)) || (rc = g->d_y.ensure((size_t)nb * 48)) ||
        (rc = g->d_misc.ensure(((size_t)iters + 4) * 8 + 64)))
        return rc;
    hipStream_t st = ctx->stream;
    SVO_HIP(hipMemcpyAsync(g->d_pose.p, g->pose.data(), (size_t)nv * 56, hipMemcpyHostToDevice, st));
    SVO_HIP(hipMemcpyAsync(g->d_from.p, g->efrom.data(), (size_t)ne * 4, hipMemcpyHostToDevice, st));
    SVO_HIP(hipMemcpyAsync(g->d_to.p, g->eto.data(), (size_t)ne * 4, hipMemcpyHostToDevice, st));
    SVO_HIP(hipMemcpyAsync(g->d_meas.p, g->meas.data(), (size_t)ne * 56, hipMemcpyHostToDevice, st));
    SVO_HIP(hipMemcpyAsync(g->d_incptr.p, incptr.data(), (size_t)(nv + 1) * 4, hipMemcpyHostToDevice, st));
    SVO_HIP(hipMemcpyAsync(g->d_inc.p, inc.data(), (size_t)2 * ne * 4, hipMemcpyHostToDevice, st));
    SVO_HIP(hipMemcpyAsync(g->d_benv.p, benv.data(), (size_t)nb * 4, hipMemcpyHostToDevice, st));
    SVO_HIP(hipMemcpyAsync(g->d_rowptr.p, rowptr.data(), (size_t)(nb + 1) * 4, hipMemcpyHostToDevice, st));
    // the host vectors above must outlive the async copies
    SVO_HIP(hipStreamSynchronize(st));
    double *d_chi = g->d_misc.as<double>();
    int *d_status = reinterpret_cast<int *>(d_chi + iters + 2);
    EdgeOut *eo = reinterpret_cast<EdgeOut *>(g->d_eo.p);
    ScopedKernelTime tm(ctx, SVO_K_POSEGRAPH);
    for (int it = 0; it <= iters; it++) {
        hipLaunchKernelGGL(pg_linearize_kernel, dim3((ne + 127) / 128), dim3(128), 0, st, g->d_pose.as<double>(),
                           g->d_from.as<int>(), g->d_to.as<int>(), g->d_meas.as<double>(), ne, eo);
        hipLaunchKernelGGL(pg_chi2_kernel, dim3(1), dim3(256), 0, st, eo, ne, d_chi + it);
        if (it == iters)
            break;
        SVO_HIP(hipMemsetAsync(g->d_Ls.p, 0, (nblk + 1) * 288, st));
        hipLaunchKernelGGL(pg_assemble_kernel, dim3((nb + 127) / 128), dim3(128), 0, st, nb, g->d_incptr.as<int>(),
                           g->d_inc.as<int>(), g->d_from.as<int>(), g->d_to.as<int>(), eo, g->d_benv.as<int>(),
                           g->d_rowptr.as<int>(), g->d_Ld.as<double>(), g->d_Ls.as<double>(), g->d_rhs.as<double>());
        hipLaunchKernelGGL(pg_solve_kernel, dim3(1), dim3(64), 0, st, nb, g->d_benv.as<int>(), g->d_rowptr.as<int>(),
                           g->d_Ld.as<double>(), g->d_Linv.as<double>(), g->d_Ls.as<double>(), g->d_rhs.as<double>(),
                           g->d_y.as<double>(), d_status);
        hipLaunchKernelGGL(pg_update_kernel, dim3((nv + 127) / 128), dim3(128), 0, st, nv, g->d_pose.as<double>(),
                           g->d_y.as<double>());
    }
    SVO_HIP(hipGetLastError());
    SVO_HIP(hipMemcpyAsync(g->pose.data(), g->d_pose.p, (size_t)nv * 56, hipMemcpyDeviceToHost, st));
    std::vector<double> hchi(iters + 1);
    SVO_HIP(hipMemcpyAsync(hchi.data(), d_chi, (size_t)(iters + 1) * 8, hipMemcpyDeviceToHost, st));
    int hstatus = 0;
    SVO_HIP(hipMemcpyAsync(&hstatus, d_status, 4, hipMemcpyDeviceToHost, st));
    SVO_HIP(hipStreamSynchronize(st));
    if (chi2)
        memcpy(chi2, hchi.data(), (size_t)(iters + 1) * 8);
    if (iters > 0 && hstatus != 0) {
        svo_set_error("pose graph: normal matrix not positive definite at block row %d", hstatus - 1);
        return SVO_ERR_STATE;
    }
    return SVO_OK;
}

int svo_pg_write_g2o(const svo_posegraph *g, const char *path)
{
    SVO_CHECK_ARG(g && path);
    FILE *f = fopen(path, "w");
    if (!f) {
        svo_set_error("cannot open %s", path);
        return SVO_ERR_ARG;
    }
    for (int v = 0; v < g->nv(); v++) {
        const double *p = &g->pose[7 * v];
        fprintf(f, "VERTEX_SE3:QUAT %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", v, p[0], p[1], p[2], p[3], p[4],
                p[5], p[6]);
    }
    for (int e = 0; e < g->ne(); e++) {
        const double *z = &g->meas[7 * e];
        fprintf(f, "EDGE_SE3:QUAT %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g", g->efrom[e], g->eto[e], z[0], z[1],
                z[2], z[3], z[4], z[5], z[6]);
        for (int i = 0; i < 6; i++)
            for (int j = i; j < 6; j++)
                fprintf(f, " %d", i == j ? 1 : 0);
        fprintf(f, "\n");
    }
    fclose(f);
    return SVO_OK;
}

}  // extern "C"
