// pnp.hip -- 3D-2D PnP-RANSAC for gfx950.
//
// Replaces cv::solvePnPRansac(obj, img, K, 0, rvec, tvec, false, 100, thr, conf, inliers)
// as the reference calls it at src/keyFrameManagement.cpp:84 (100, 1.0, 0.99) and :88
// (100, 8.0, 0.98): RANSAC over 5-point samples solved by EPnP, squared reprojection error
// (float) against thr^2, adaptive iteration bound, then an iterative refinement of the best
// hypothesis over its inliers.
//
//   solve   one WAVEFRONT per hypothesis.  The sample's 10x12 system, M^T M and the 12x12
//           symmetric eigenproblem live in LDS; the Jacobi sweeps run wave-parallel (each
//           round-robin round holds 6 disjoint rotations: 6 lanes compute them, then all
//           lanes apply the 72 column / row / eigenvector updates).  The three beta
//           linearisations (N = 1, 2, 3) + Gauss-Newton + rigid alignment run on lanes
//           0..2 side by side, entirely in registers.
//   score   one WAVEFRONT per hypothesis, N correspondences strided over the lanes.
//   finish  ONE workgroup, one launch: sequential-semantics replay (ransac_common.hip.h) with the
//           inlier count published to the host mailbox at once, mask + ordered inlier list of
//           the winner (ballot scan), then Levenberg-Marquardt on (R, t): 6x6 normal equations
//           reduced with a fixed-order tree (deterministic), solved by one lane.
#include <cfloat>

#include "ransac_common.hip.h"
#include "svo_internal.h"

using namespace svo;

namespace {

constexpr int MP = 5;  // model points

struct K4 {
    double fx, fy, cx, cy;
};
struct PnpResult;
// One PnP-RANSAC problem as the kernels see it.  A launch may carry several independent problems
// (blockIdx.y picks the job): these kernels are chains of one wave's f64 latency, so the jobs of
// chunks that run in lock step (svo_vo_run_chunks) cost the time of one.
struct PnpJob {
    const float *obj;
    const float *img;
    int n_host;
    const int *d_n;
    K4 K;
    uint64_t seed;
    int iterations;
    double confidence;
    float thr;
    int max_lm_iters;
    svo::RansacState *st;
    double *hyp;
    int *nmodels, *counts, *d_m;
    uint8_t *mask;
    int *inl;
    PnpResult *out;
    int direct;             // 1: no RANSAC -- hypothesis 0 (written by pnp_dlt_kernel) is refined over ALL points
    const int *dlt_status;  // direct mode: 0 = hypothesis 0 is valid
    VoChain *chain;         // chain mode (svo_pnp_job::chain): the frame's policy is decided here, on the device
    const int *cnt_trk;
    unsigned *ticket;  // "last wave of a phase" counter (self-resetting)
};
template <int NJ> struct PnpBatchN {  // NJ = 1: a chunk on its own (a sixteenth of the kernel arguments per launch)
    PnpJob j[NJ];
};
using PnpBatch = PnpBatchN<SVO_LK_MAX_JOBS>;
static_assert(sizeof(PnpBatch) + 16 <= 4096, "kernel arguments are limited to 4 KB");

// ---- small dense helpers (registers, static indexing) --------------------------------------
template <int N>
__device__ __forceinline__ void lstsq6(const double (&A)[6][N], const double (&b)[6], double (&x)[N])
{
    // normal equations, Tikhonov-damped by 1e-14 of the mean diagonal, Cholesky
    double Nm[N][N], rhs[N], L[N][N], y[N];
    double tr = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < N; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 6; k++)
                s += A[k][i] * A[k][j];
            Nm[i][j] = s;
        }
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++)
            s += A[k][i] * b[k];
        rhs[i] = s;
        tr += Nm[i][i];
    }
    const double damp = 1e-14 * tr / N;
#pragma unroll
    for (int i = 0; i < N; i++)
        x[i] = 0;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double s = Nm[i][j] + (i == j ? damp : 0.);
#pragma unroll
            for (int k = 0; k < j; k++)
                s -= L[i][k] * L[j][k];
            if (i == j) {
                if (!(s > 0))
                    ok = false;
                L[i][i] = sqrt(s);
            } else
                L[i][j] = s / L[j][j];
        }
    if (!ok)
        return;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double s = rhs[i];
#pragma unroll
        for (int k = 0; k < i; k++)
            s -= L[i][k] * y[k];
        y[i] = s / L[i][i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < N; k++)
            s -= L[k][i] * x[k];
        x[i] = s / L[i][i];
    }
}

// SVD of a 3x3 matrix by one-sided Jacobi; returns U and V (columns = singular vectors,
// decreasing singular value, U completed to a basis when the last one vanishes)
__device__ void svd3(const double (&Ain)[9], double (&U)[9], double (&Vout)[9])
{
    double A[3][3], V[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            A[i][j] = Ain[3 * i + j];
            V[i][j] = i == j ? 1. : 0.;
        }
    for (int sweep = 0; sweep < 12; sweep++) {
        bool rotated = false;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            constexpr int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
            const int p = P[e], q = Q[e];
            double al = 0, be = 0, ga = 0;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                al += A[i][p] * A[i][p];
                be += A[i][q] * A[i][q];
                ga += A[i][p] * A[i][q];
            }
            if (ga == 0 || fabs(ga) <= DBL_EPSILON * sqrt(al * be))
                continue;
            rotated = true;
            const double a2 = be - al, b2 = 2. * ga;
            const double h = fabs(a2) + sqrt(a2 * a2 + b2 * b2);
            const double inv = 1. / sqrt(h * h + b2 * b2);
            const double c = h * inv, s = (a2 >= 0 ? b2 : -b2) * inv;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double ap = A[i][p], aq = A[i][q];
                A[i][p] = c * ap - s * aq;
                A[i][q] = s * ap + c * aq;
                const double vp = V[i][p], vq = V[i][q];
                V[i][p] = c * vp - s * vq;
                V[i][q] = s * vp + c * vq;
            }
        }
        if (!rotated)
            break;
    }
    double nrm[3];
#pragma unroll
    for (int j = 0; j < 3; j++)
        nrm[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    // order by decreasing norm (same comparison network as the oracle's insertion loops)
    int o0 = 0, o1 = 1, o2 = 2;
    if (nrm[o1] > nrm[o0]) {
        int t = o0;
        o0 = o1;
        o1 = t;
    }
    if (nrm[o2] > nrm[o0]) {
        int t = o0;
        o0 = o2;
        o2 = t;
    }
    if (nrm[o2] > nrm[o1]) {
        int t = o1;
        o1 = o2;
        o2 = t;
    }
    const int ord[3] = {o0, o1, o2};
    double Uc[3][3], Vc[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int j = ord[k];
        const double nj = j == 0 ? nrm[0] : (j == 1 ? nrm[1] : nrm[2]);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double vij = j == 0 ? V[i][0] : (j == 1 ? V[i][1] : V[i][2]);
            const double aij = j == 0 ? A[i][0] : (j == 1 ? A[i][1] : A[i][2]);
            Vc[k][i] = vij;
            Uc[k][i] = nj > 0 ? aij / nj : 0.;
        }
    }
    const double n0 = o0 == 0 ? nrm[0] : (o0 == 1 ? nrm[1] : nrm[2]);
    const double n2 = o2 == 0 ? nrm[0] : (o2 == 1 ? nrm[1] : nrm[2]);
    if (!(n2 > 1e-12 * n0)) {
        Uc[2][0] = Uc[0][1] * Uc[1][2] - Uc[0][2] * Uc[1][1];
        Uc[2][1] = Uc[0][2] * Uc[1][0] - Uc[0][0] * Uc[1][2];
        Uc[2][2] = Uc[0][0] * Uc[1][1] - Uc[0][1] * Uc[1][0];
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            U[3 * i + k] = Uc[k][i];
            Vout[3 * i + k] = Vc[k][i];
        }
}

// ---- wave-parallel cyclic Jacobi for a symmetric n x n matrix in LDS -------------------------
// Round-robin pair schedule (n-1 rounds of n/2 disjoint pairs, odd n padded with an idle
// player).  Per round: lanes < m/2 compute their rotation, then all lanes apply the column
// updates, then the row updates, then the eigenvector updates -- the exact order the CPU
// oracle uses, so the two round identically.  A, V: row-major n x n; cs: 2*8 doubles; pq: 16 ints.
// n is a template parameter so the item -> (pair, index) maps divide by constants.
template <int n>
__device__ void wave_jacobi_eigen_sym(double *A, double *V, double *cs, int *pq, int sweeps, int lane)
{
    constexpr int m = n + (n & 1), half = m / 2, items = half * n;
    for (int i = lane; i < n * n; i += 64)
        V[i] = (i / n) == (i % n) ? 1. : 0.;
    wave_lds_fence();
    // the (pair, index) items of this lane: item = lane (+64); constant divisors
    const int e0 = lane / n, k0 = lane - e0 * n;
    const int e1 = (lane + 64) / n, k1 = (lane + 64) - e1 * n;
    const bool has0 = lane < items, has1 = lane + 64 < items;
    for (int s = 0; s < sweeps; s++)
        for (int r = 0; r < m - 1; r++) {
            if (lane < half) {
                const int k = lane;
                const int a = k == 0 ? m - 1 : (r + k) % (m - 1);
                const int b = k == 0 ? r % (m - 1) : (r - k + (m - 1)) % (m - 1);
                int p = a < b ? a : b, q = a < b ? b : a;
                double c = 1., sn = 0.;
                if (a >= n || b >= n) {
                    p = -1;
                    q = -1;
                } else {
                    const double apq = A[p * n + q];
                    const double app = A[p * n + p], aqq = A[q * n + q];
                    // a rotation below the resolution of a double is the identity (late sweeps skip the
                    // division and square roots); otherwise tan(phi) with ONE division, as the oracle
                    if (!(fabs(apq) <= 1e-19 * (fabs(app) + fabs(aqq)))) {
                        const double a2 = aqq - app, b2 = 2. * apq;
                        const double h = fabs(a2) + sqrt(a2 * a2 + b2 * b2);
                        const double inv = 1. / sqrt(h * h + b2 * b2);
                        c = h * inv;
                        sn = (a2 >= 0 ? b2 : -b2) * inv;
                    }
                }
                cs[2 * k] = c;
                cs[2 * k + 1] = sn;
                pq[2 * k] = p;
                pq[2 * k + 1] = q;
            }
            wave_lds_fence();
            const int p0 = has0 ? pq[2 * e0] : -1, q0 = has0 ? pq[2 * e0 + 1] : -1;
            const int p1 = has1 ? pq[2 * e1] : -1, q1 = has1 ? pq[2 * e1 + 1] : -1;
            const double c0 = has0 ? cs[2 * e0] : 1., s0 = has0 ? cs[2 * e0 + 1] : 0.;
            const double c1 = has1 ? cs[2 * e1] : 1., s1 = has1 ? cs[2 * e1 + 1] : 0.;
            // columns p,q:  A <- A J
            if (p0 >= 0) {
                const double akp = A[k0 * n + p0], akq = A[k0 * n + q0];
                A[k0 * n + p0] = c0 * akp - s0 * akq;
                A[k0 * n + q0] = s0 * akp + c0 * akq;
            }
            if (p1 >= 0) {
                const double akp = A[k1 * n + p1], akq = A[k1 * n + q1];
                A[k1 * n + p1] = c1 * akp - s1 * akq;
                A[k1 * n + q1] = s1 * akp + c1 * akq;
            }
            wave_lds_fence();
            // rows p,q:  A <- J^T A ;  V <- V J
            if (p0 >= 0) {
                const double apk = A[p0 * n + k0], aqk = A[q0 * n + k0];
                A[p0 * n + k0] = c0 * apk - s0 * aqk;
                A[q0 * n + k0] = s0 * apk + c0 * aqk;
                const double vkp = V[k0 * n + p0], vkq = V[k0 * n + q0];
                V[k0 * n + p0] = c0 * vkp - s0 * vkq;
                V[k0 * n + q0] = s0 * vkp + c0 * vkq;
            }
            if (p1 >= 0) {
                const double apk = A[p1 * n + k1], aqk = A[q1 * n + k1];
                A[p1 * n + k1] = c1 * apk - s1 * aqk;
                A[q1 * n + k1] = s1 * apk + c1 * aqk;
                const double vkp = V[k1 * n + p1], vkq = V[k1 * n + q1];
                V[k1 * n + p1] = c1 * vkp - s1 * vkq;
                V[k1 * n + q1] = s1 * vkp + c1 * vkq;
            }
            wave_lds_fence();
        }
}

__device__ __forceinline__ double dot3(const double *a, const double *b)
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

// per-wave LDS layout (doubles)
struct WaveLds {
    double A[144];
    double V[144];
    double M[120];
    double vsel[48];
    double L[60];
    double rho[8];
    double alphas[20];
    double pws[16];
    double us[10];
    double A3[9];
    double V3[9];
    double cs[16];
    int pq[16];
};

__device__ __forceinline__ float reproj_err_sq(const double (&P)[12], const K4 &K, float X, float Y, float Z, float u,
                                               float v);

// One RANSAC iteration by one wave: sample, EPnP, the hypothesis into hyp[it], its inlier count into
// counts[it], nmodels[it] = 1 / 0 (no model) / -1 (no sample: the sequential loop stops there).
__device__ __forceinline__ void pnp_hypothesis(const PnpJob &job, int it, int n, WaveLds &S, int lane)
{
    const float *__restrict__ obj = job.obj, *__restrict__ img = job.img;
    const K4 K = job.K;
    const uint64_t seed = job.seed;
    double *__restrict__ hyp = job.hyp;
    int *__restrict__ nmodels = job.nmodels;
    int *__restrict__ counts = job.counts;
    if (n < MP) {
        if (lane == 0) {
            nmodels[it] = -1;
            counts[it] = 0;
        }
        return;
    }
    // ---- sample (every lane draws the same indices) ----
    int idx[MP];
    bool filled = true;
    {
        uint32_t draw = 0;
        int guard = 0;
#pragma unroll
        for (int slot = 0; slot < MP; slot++) {
            int v = 0;
            bool got = false;
            while (!got && guard < kMaxDraws) {
                v = (int)(rng_u32(seed, (uint32_t)it, draw++) % (uint32_t)n);
                guard++;
                bool dup = false;
#pragma unroll
                for (int j = 0; j < MP; j++)
                    if (j < slot && idx[j] == v)
                        dup = true;
                got = !dup;
            }
            if (!got)
                filled = false;
            idx[slot] = v;
        }
    }
    if (!filled) {
        if (lane == 0)
            nmodels[it] = -1;
        if (lane == 0)
            counts[it] = 0;
        return;
    }
    if (lane < MP) {
        const int i = lane == 0 ? idx[0] : lane == 1 ? idx[1] : lane == 2 ? idx[2] : lane == 3 ? idx[3] : idx[4];
        S.pws[3 * lane] = obj[3 * i];
        S.pws[3 * lane + 1] = obj[3 * i + 1];
        S.pws[3 * lane + 2] = obj[3 * i + 2];
        S.us[2 * lane] = img[2 * i];
        S.us[2 * lane + 1] = img[2 * i + 1];
    }
    wave_lds_fence();
    // ---- control points: centroid + PCA (every lane, same values) ----
    double cws[4][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double s = 0;
#pragma unroll
        for (int i = 0; i < MP; i++)
            s += S.pws[3 * i + k];
        cws[0][k] = s / MP;
    }
    if (lane < 9) {
        const int a = lane / 3, b = lane % 3;
        double s = 0;
        for (int i = 0; i < MP; i++)
            s += (S.pws[3 * i + a] - cws[0][a == 0 ? 0 : a == 1 ? 1 : 2]) *
                 (S.pws[3 * i + b] - cws[0][b == 0 ? 0 : b == 1 ? 1 : 2]);
        S.A3[lane] = s;
    }
    wave_lds_fence();
    wave_jacobi_eigen_sym<3>(S.A3, S.V3, S.cs, S.pq, 5, lane);
    {
        const double wc[3] = {S.A3[0], S.A3[4], S.A3[8]};
        int o0 = 0, o1 = 1, o2 = 2;  // descending eigenvalue, same comparison order as the oracle
        if (wc[o1] > wc[o0]) {
            int t = o0;
            o0 = o1;
            o1 = t;
        }
        if (wc[o2] > wc[o0]) {
            int t = o0;
            o0 = o2;
            o2 = t;
        }
        if (wc[o2] > wc[o1]) {
            int t = o1;
            o1 = o2;
            o2 = t;
        }
        const int ord[3] = {o0, o1, o2};
#pragma unroll
        for (int i = 1; i < 4; i++) {
            const int e = ord[i - 1];
            const double lam = S.A3[4 * e];
            const double k = sqrt((lam > 0 ? lam : 0) / MP);
#pragma unroll
            for (int j = 0; j < 3; j++)
                cws[i][j] = cws[0][j] + k * S.V3[3 * j + e];
        }
    }
    // ---- barycentric coordinates ----
    double CC[9], CCi[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 1; j < 4; j++)
            CC[3 * i + j - 1] = cws[j][i] - cws[0][i];
    const double det = CC[0] * (CC[4] * CC[8] - CC[5] * CC[7]) - CC[1] * (CC[3] * CC[8] - CC[5] * CC[6]) +
                       CC[2] * (CC[3] * CC[7] - CC[4] * CC[6]);
    const double scale = fabs(CC[0]) + fabs(CC[4]) + fabs(CC[8]) + fabs(CC[1]) + fabs(CC[2]) + fabs(CC[3]) +
                         fabs(CC[5]) + fabs(CC[6]) + fabs(CC[7]);
    if (!(fabs(det) > 1e-18 * scale * scale * scale) || !isfinite(det)) {
        if (lane == 0)
            nmodels[it] = 0;  // coplanar / coincident sample: no model, the loop continues
        if (lane == 0)
            counts[it] = 0;
        return;
    }
    const double id = 1. / det;
    CCi[0] = (CC[4] * CC[8] - CC[5] * CC[7]) * id;
    CCi[1] = (CC[2] * CC[7] - CC[1] * CC[8]) * id;
    CCi[2] = (CC[1] * CC[5] - CC[2] * CC[4]) * id;
    CCi[3] = (CC[5] * CC[6] - CC[3] * CC[8]) * id;
    CCi[4] = (CC[0] * CC[8] - CC[2] * CC[6]) * id;
    CCi[5] = (CC[2] * CC[3] - CC[0] * CC[5]) * id;
    CCi[6] = (CC[3] * CC[7] - CC[4] * CC[6]) * id;
    CCi[7] = (CC[1] * CC[6] - CC[0] * CC[7]) * id;
    CCi[8] = (CC[0] * CC[4] - CC[1] * CC[3]) * id;
    if (lane < MP) {
        const double d0 = S.pws[3 * lane] - cws[0][0], d1 = S.pws[3 * lane + 1] - cws[0][1],
                     d2 = S.pws[3 * lane + 2] - cws[0][2];
        double a[4];
#pragma unroll
        for (int j = 0; j < 3; j++)
            a[1 + j] = CCi[3 * j] * d0 + CCi[3 * j + 1] * d1 + CCi[3 * j + 2] * d2;
        a[0] = 1.0 - a[1] - a[2] - a[3];
#pragma unroll
        for (int j = 0; j < 4; j++)
            S.alphas[4 * lane + j] = a[j];
    }
    wave_lds_fence();
    // ---- M (10 x 12) and M^T M ----
    for (int e = lane; e < 120; e += 64) {
        const int r = e / 12, c = e - r * 12;
        const int i = r >> 1, j = c / 3, comp = c - 3 * j;
        const double al = S.alphas[4 * i + j];
        double v;
        if ((r & 1) == 0)
            v = comp == 0 ? al * K.fx : (comp == 1 ? 0. : al * (K.cx - S.us[2 * i]));
        else
            v = comp == 0 ? 0. : (comp == 1 ? al * K.fy : al * (K.cy - S.us[2 * i + 1]));
        S.M[e] = v;
    }
    wave_lds_fence();
    for (int e = lane; e < 144; e += 64) {
        const int a = e / 12, b = e - a * 12;
        double s = 0;
        for (int r = 0; r < 2 * MP; r++)
            s += S.M[r * 12 + a] * S.M[r * 12 + b];
        S.A[e] = s;
    }
    wave_lds_fence();
    wave_jacobi_eigen_sym<12>(S.A, S.V, S.cs, S.pq, 6, lane);  // quadratic convergence: off-diagonals < 1e-12 after 6
    // ---- the four smallest eigenvalues, ascending (ties: lower index first) ----
    {
        int sel[4];
        unsigned used = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int best = -1;
            double wb = 0;
            for (int e = 0; e < 12; e++) {
                const double we = S.A[13 * e];
                if (!((used >> e) & 1u) && (best < 0 || we < wb)) {
                    best = e;
                    wb = we;
                }
            }
            used |= 1u << best;
            sel[k] = best;
        }
        if (lane < 48) {
            const int k = lane / 12, i = lane - 12 * k;
            const int sk = k == 0 ? sel[0] : k == 1 ? sel[1] : k == 2 ? sel[2] : sel[3];
            S.vsel[lane] = S.V[12 * i + sk];
        }
    }
    wave_lds_fence();
    // ---- L (6 x 10) and rho: lane i < 6 builds row i ----
    if (lane < 6) {
        constexpr int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {1, 2, 3, 2, 3, 3};
        int pa = 0, pb = 1;
#pragma unroll
        for (int e = 0; e < 6; e++)
            if (lane == e) {
                pa = PA[e];
                pb = PB[e];
            }
        double dv[4][3];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                dv[i][k] = S.vsel[12 * i + 3 * pa + k] - S.vsel[12 * i + 3 * pb + k];
        double *Lr = S.L + 10 * lane;
        Lr[0] = dot3(dv[0], dv[0]);
        Lr[1] = 2. * dot3(dv[0], dv[1]);
        Lr[2] = dot3(dv[1], dv[1]);
        Lr[3] = 2. * dot3(dv[0], dv[2]);
        Lr[4] = 2. * dot3(dv[1], dv[2]);
        Lr[5] = dot3(dv[2], dv[2]);
        Lr[6] = 2. * dot3(dv[0], dv[3]);
        Lr[7] = 2. * dot3(dv[1], dv[3]);
        Lr[8] = 2. * dot3(dv[2], dv[3]);
        Lr[9] = dot3(dv[3], dv[3]);
        double ca[3], cb[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            ca[k] = pa == 0 ? cws[0][k] : pa == 1 ? cws[1][k] : pa == 2 ? cws[2][k] : cws[3][k];
            cb[k] = pb == 0 ? cws[0][k] : pb == 1 ? cws[1][k] : pb == 2 ? cws[2][k] : cws[3][k];
        }
        const double d[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]};
        S.rho[lane] = dot3(d, d);
    }
    wave_lds_fence();
    // ---- lanes 0..2: beta linearisation N = lane+1, Gauss-Newton, rigid alignment ----
    double R[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, t[3] = {0, 0, 0};
    double err = 1e300;
    bool good = false;
    if (lane < 3) {
        double L[6][10], rho[6], b[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 6; i++) {
#pragma unroll
            for (int k = 0; k < 10; k++)
                L[i][k] = S.L[10 * i + k];
            rho[i] = S.rho[i];
        }
        if (lane == 0) {  // [B11 B12 B13 B14]
            double A[6][4], x[4];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                A[i][0] = L[i][0];
                A[i][1] = L[i][1];
                A[i][2] = L[i][3];
                A[i][3] = L[i][6];
            }
            lstsq6<4>(A, rho, x);
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
        } else if (lane == 1) {  // [B11 B12 B22]
            double A[6][3], x[3];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                A[i][0] = L[i][0];
                A[i][1] = L[i][1];
                A[i][2] = L[i][2];
            }
            lstsq6<3>(A, rho, x);
            if (x[0] < 0) {
                b[0] = sqrt(-x[0]);
                b[1] = x[2] < 0 ? sqrt(-x[2]) : 0.;
            } else {
                b[0] = sqrt(x[0]);
                b[1] = x[2] > 0 ? sqrt(x[2]) : 0.;
            }
            if (x[1] < 0)
                b[0] = -b[0];
        } else {  // [B11 B12 B22 B13 B23]
            double A[6][5], x[5];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int k = 0; k < 5; k++)
                    A[i][k] = L[i][k];
            lstsq6<5>(A, rho, x);
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
        }
        for (int gn = 0; gn < 5; gn++) {  // Gauss-Newton on the six control-point distances
            double A[6][4], r[6], x[4];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double *l = L[i];
                A[i][0] = 2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3];
                A[i][1] = l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3];
                A[i][2] = l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3];
                A[i][3] = l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3];
                r[i] = rho[i] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] +
                                 l[4] * b[1] * b[2] + l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] +
                                 l[8] * b[2] * b[3] + l[9] * b[3] * b[3]);
            }
            lstsq6<4>(A, r, x);
#pragma unroll
            for (int k = 0; k < 4; k++)
                b[k] += x[k];
        }
        // control points in the camera frame, sample points in the camera frame
        double ccs[4][3], pcs[MP][3];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                ccs[j][k] = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k = 0; k < 3; k++)
                    ccs[j][k] += b[i] * S.vsel[12 * i + 3 * j + k];
#pragma unroll
        for (int i = 0; i < MP; i++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                pcs[i][k] = S.alphas[4 * i] * ccs[0][k] + S.alphas[4 * i + 1] * ccs[1][k] +
                            S.alphas[4 * i + 2] * ccs[2][k] + S.alphas[4 * i + 3] * ccs[3][k];
        if (pcs[0][2] < 0.) {
#pragma unroll
            for (int i = 0; i < MP; i++)
#pragma unroll
                for (int k = 0; k < 3; k++)
                    pcs[i][k] = -pcs[i][k];
        }
        double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < MP; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                pc0[k] += pcs[i][k];
                pw0[k] += S.pws[3 * i + k];
            }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            pc0[k] /= MP;
            pw0[k] /= MP;
        }
        double ABt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, U[9], V[9];
#pragma unroll
        for (int i = 0; i < MP; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int k = 0; k < 3; k++)
                    ABt[3 * j + k] += (pcs[i][j] - pc0[j]) * (S.pws[3 * i + k] - pw0[k]);
        svd3(ABt, U, V);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
        const double dR = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                          R[2] * (R[3] * R[7] - R[4] * R[6]);
        if (dR < 0) {
            R[6] = -R[6];
            R[7] = -R[7];
            R[8] = -R[8];
        }
#pragma unroll
        for (int i = 0; i < 3; i++)
            t[i] = pc0[i] - (R[3 * i] * pw0[0] + R[3 * i + 1] * pw0[1] + R[3 * i + 2] * pw0[2]);
        double sum = 0;
#pragma unroll
        for (int i = 0; i < MP; i++) {
            const double *pw = S.pws + 3 * i;
            const double Xc = dot3(R, pw) + t[0], Yc = dot3(R + 3, pw) + t[1];
            const double inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
            const double ue = K.cx + K.fx * Xc * inv_Zc, ve = K.cy + K.fy * Yc * inv_Zc;
            const double du = S.us[2 * i] - ue, dv = S.us[2 * i + 1] - ve;
            sum += sqrt(du * du + dv * dv);
        }
        err = sum / MP;
        good = isfinite(err);
#pragma unroll
        for (int i = 0; i < 9; i++)
            good = good && isfinite(R[i]);
        if (!good)
            err = 1e300;
    }
    // ---- best of the three (lowest error, ties to the lower N like the oracle's scan) ----
    const double e0 = __shfl(err, 0), e1 = __shfl(err, 1), e2 = __shfl(err, 2);
    int best = -1;
    double eb = 1e300;
    if (e0 < 1e300) {
        best = 0;
        eb = e0;
    }
    if (e1 < 1e300 && (best < 0 || e1 < eb)) {
        best = 1;
        eb = e1;
    }
    if (e2 < 1e300 && (best < 0 || e2 < eb)) {
        best = 2;
        eb = e2;
    }
    if (best < 0) {
        if (lane == 0) {
            nmodels[it] = 0;
            counts[it] = 0;
        }
        return;
    }
    double P[12];
#pragma unroll
    for (int i = 0; i < 9; i++)
        P[i] = __shfl(R[i], best);
#pragma unroll
    for (int i = 0; i < 3; i++)
        P[9 + i] = __shfl(t[i], best);
    if (lane == 0) {
        double *dst = hyp + (size_t)it * 12;
#pragma unroll
        for (int i = 0; i < 12; i++)
            dst[i] = P[i];
        nmodels[it] = 1;
    }
    // ---- PnPRansacCallback::computeError over all points: the hypothesis's inlier count (a launch of its
    // own until round 2; same wave-per-hypothesis mapping, and one launch less in the chain) ----
    const float2 *__restrict__ img2 = reinterpret_cast<const float2 *>(img);
    const float thr = job.thr;
    int cnt = 0;
#pragma clang loop unroll(disable)  // unrolled, the f64 bodies take >128 VGPRs + scratch
    for (int i = lane; i < n; i += 64) {
        const float2 u = img2[i];
        cnt += reproj_err_sq(P, K, obj[3 * i], obj[3 * i + 1], obj[3 * i + 2], u.x, u.y) <= thr ? 1 : 0;
    }
    cnt = wave_sum_small(cnt);
    if (lane == 0)
        counts[it] = cnt;
}

// Iterations [it0, min(it1_cap, iterations)) of the RANSAC loop, one wave (= one workgroup) per iteration.
// Two phases as in fransac.hip: the LAST wave of a phase that has a successor replays the sequential loop over
// what has been scored so far and stores the state; the next phase's waves leave at once when the loop has
// ended (at VO inlier ratios the adaptive bound is reached after about 15 of the 100 iterations -- all 100
// hypotheses were 6 % of the vector instructions of a bench run).  pnp_finish_kernel replays the whole loop
// itself, reads only iterations below the bound, and so never sees a skipped one.
constexpr int PNP_PHASE_A = 32;
// LEAN (several jobs per launch, beside tracking launches): 128 VGPRs, four waves per SIMD, 172 spilled registers -- the
// launch is wide enough to hide them.  !LEAN (one job: a chunk on its own, where the launch is a chain of ONE wave's
// latency and every scratch access is on it): the whole register file, nothing spilled.  Same arithmetic either way.
template <bool LEAN> __global__ __launch_bounds__(64, LEAN ? 4 : 1) void pnp_solve_kernel(PnpBatchN<LEAN ? SVO_LK_MAX_JOBS : 1> batch, int it0, int it1_cap)
{
    svo_chain_priority();
    const PnpJob &job = batch.j[blockIdx.y];
    // the chain halted at an earlier frame: nothing of this chunk runs any more.  A wave that finds the chain halted does
    // no work but still takes its ticket below (as fr_ransac_kernel: the flag may change under a launch that another
    // stream dispatches, and a skipped ticket would leave the self-resetting counter at a partial count for good).
    const bool due = !(job.chain && __builtin_amdgcn_readfirstlane(__hip_atomic_load(&job.chain->run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0);
    const int it1 = it1_cap < job.iterations ? it1_cap : job.iterations;
    if (it0 > 0 && job.st->done)
        return;  // the same answer in every wave of the launch (written by the previous launch of this stream)
    __shared__ WaveLds s_lds[1];
    const int lane = threadIdx.x & 63;
    const int it = __builtin_amdgcn_readfirstlane(it0 + blockIdx.x);
    const int n = job.d_n ? min(*job.d_n, job.n_host) : job.n_host;  // as pnp_finish: never past the capacity
    if (due && it < it1)
        pnp_hypothesis(job, it, n, s_lds[0], lane);
    if (it1 >= job.iterations)
        return;  // no phase follows: pnp_finish_kernel replays
    int last = 0;  // bit 0: this wave holds the last ticket; bit 1: every wave of the launch found the chain live
    if (lane == 0) {
        __threadfence();
        const unsigned t = atomicAdd(job.ticket, due ? 1u : 0x10001u);  // low half: arrivals; high half: of them, halted views
        last = ((t & 0xffffu) == gridDim.x - 1 ? 1 : 0) | (due && (t >> 16) == 0 ? 2 : 0);
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!(last & 1))
        return;
    if (lane == 0)
        *job.ticket = 0;  // ready for the next launch
    if (!(last & 2)) {  // a halted view: slots of this phase were not written -- nothing to replay; "ended" for the next phase
        if (lane == 0) {
            RansacState r;
            r.niters = 0, r.next_iter = 0, r.best_iter = -1, r.best_model = 0, r.best_count = 0, r.done = 1, r.iters_run = 0, r.pad = 0;
            *job.st = r;
        }
        return;
    }
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // other waves' counts, not this CU's stale lines
        *job.st = ransac_replay<1>(job.st, it0 == 0 ? 1 : 0, it1, job.iterations, n, job.confidence, job.nmodels,
                                   job.counts, MP);
    }
}

// PnPRansacCallback::computeError for one correspondence
__device__ __forceinline__ float reproj_err_sq(const double (&P)[12], const K4 &K, float X, float Y, float Z,
                                               float u, float v)
{
    const double Xc = P[0] * X + P[1] * Y + P[2] * Z + P[9];
    const double Yc = P[3] * X + P[4] * Y + P[5] * Z + P[10];
    const double Zc = P[6] * X + P[7] * Y + P[8] * Z + P[11];
    const double z = Zc != 0 ? 1. / Zc : 1.;
    const float px = (float)(Xc * z * K.fx + K.cx), py = (float)(Yc * z * K.fy + K.cy);
    const float dx = u - px, dy = v - py;
    return (float)((double)dx * dx + (double)dy * dy);
}

// ---- Levenberg-Marquardt refinement over the inlier list -------------------------------------
__device__ void rodrigues_dev(const double *r, double *R)
{
    const double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < DBL_EPSILON) {
        for (int i = 0; i < 9; i++)
            R[i] = (i % 4) == 0 ? 1. : 0.;
        return;
    }
    const double c = svo_cos(th), s = svo_sin(th), c1 = 1. - c, it = 1. / th;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
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

__device__ void rodrigues_inv_dev(const double *R, double *r)
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
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
        const double vth = 1 / (2 * s) * theta;
        r[0] = rx * vth;
        r[1] = ry * vth;
        r[2] = rz * vth;
    }
}

__device__ bool chol6_solve(const double *Ain, const double *b, double *x)
{
    double L[36];
    for (int i = 0; i < 36; i++)
        L[i] = 0;
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

constexpr int NACC = 28;  // [0] squared error, [1..21] upper-triangular J^T J, [22..27] J^T r

// Fixed-order block reduction of the first NV of a thread's accumulators (256 threads):
// every thread parks its partials in LDS ([value][thread], rows padded against bank
// conflicts), thread (w*CH + k) adds the 64 partials of wave w for value k in lane order, and
// thread k adds the four wave totals.  The order never changes, so results are reproducible.
// The values go through LDS RED_CHUNK at a time: the workgroup then needs 16 KB of LDS instead of
// 59 KB and starts beside a tracking launch, whose waves leave about 24 KB of every CU's LDS free
// (with the big buffer the kernel waited ~1 ms for a CU to drain).
constexpr int RED_STRIDE = 257;
constexpr int RED_CHUNK = 7;
template <int NV> __device__ void block_reduce(const double *v, double *s_all /*RED_CHUNK*RED_STRIDE*/,
                                               double *s_part /*4*NACC*/, double *s_out /*NACC*/)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int c0 = 0; c0 < NV; c0 += RED_CHUNK) {
        constexpr int dummy = 0;
        (void)dummy;
        const int ch = NV - c0 < RED_CHUNK ? NV - c0 : RED_CHUNK;
#pragma unroll
        for (int k = 0; k < RED_CHUNK; k++)
            if (k < ch)
                s_all[k * RED_STRIDE + tid] = v[c0 + k];
        __syncthreads();
        if (tid < 4 * ch) {
            const int w = tid / ch, k = tid - w * ch;
            const double *src = s_all + k * RED_STRIDE + w * 64;
            double x = 0;
            for (int j = 0; j < 64; j++)
                x += src[j];
            s_part[w * NV + c0 + k] = x;
        }
        __syncthreads();
    }
    if (tid < NV)
        s_out[tid] = ((s_part[tid] + s_part[NV + tid]) + s_part[2 * NV + tid]) + s_part[3 * NV + tid];
    __syncthreads();
}

__device__ __forceinline__ void pnp_point_terms(const float *__restrict__ obj, const float2 *__restrict__ img, int i,
                                                const K4 &K, const double *R, const double *t, bool with_jac,
                                                double (&acc)[NACC])
{
    const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
    const double rx = R[0] * X + R[1] * Y + R[2] * Z;
    const double ry = R[3] * X + R[4] * Y + R[5] * Z;
    const double rz = R[6] * X + R[7] * Y + R[8] * Z;
    const double Xc = rx + t[0], Yc = ry + t[1], Zc = rz + t[2];
    const double iz = 1. / Zc;
    const double u = K.fx * Xc * iz + K.cx, v = K.fy * Yc * iz + K.cy;
    const float2 o = img[i];
    const double ru = u - o.x, rv = v - o.y;
    acc[0] += ru * ru + rv * rv;
    if (!with_jac)
        return;
    const double a0 = K.fx * iz, a2 = -K.fx * Xc * iz * iz;
    const double b1 = K.fy * iz, b2 = -K.fy * Yc * iz * iz;
    const double Ju[6] = {a2 * ry, a0 * rz - a2 * rx, -a0 * ry, a0, 0, a2};
    const double Jv[6] = {-b1 * rz + b2 * ry, -b2 * rx, b1 * rx, 0, b1, b2};
    int k = 1;
#pragma unroll
    for (int p = 0; p < 6; p++) {
#pragma unroll
        for (int q = p; q < 6; q++)
            acc[k++] += Ju[p] * Ju[q] + Jv[p] * Jv[q];
    }
#pragma unroll
    for (int p = 0; p < 6; p++)
        acc[22 + p] += Ju[p] * ru + Jv[p] * rv;
}

// ---- cv::solvePnP(obj, img, K, dist = 0, rvec, tvec), SOLVEPNP_ITERATIVE, no extrinsic guess -----------
// The reference's older ladder falls back to it when RANSAC finds too few inliers
// (src/bundleAdjust.cpp:470-477).  Upstream (cvFindExtrinsicCameraParams2, non-planar branch): DLT --
// rows [X Y Z 1 0 0 0 0 xX xY xZ x], [0 0 0 0 X Y Z 1 yX yY yZ y] with (x, y) = -(normalised image
// point), the right singular vector of the smallest singular value of L^T L as a 3x4 [RR | tt], sign
// by det(RR), R = U V^T of RR's SVD, t = tt * |R| / |RR| -- then Levenberg-Marquardt on all points.
// One workgroup: 40 sums (the 12x12 L^T L is [[S1, 0, Sx], [0, S1, Sy], [Sx, Sy, Sxy]] with S* = sum of
// w * P P^T, P = (X, Y, Z, 1), w = 1, x, y, x^2 + y^2: 10 unique entries each), fixed-order reduction,
// wave 0 runs the 12x12 Jacobi eigen-decomposition in LDS and writes hypothesis 0; pnp_finish_kernel in
// direct mode refines it over all points.  Coplanar object points take upstream's homography branch instead
// (see the kernel).  status: 0 ok, 1 = planar and degenerate (no homography), 2 = fewer than 6 points, 3 = degenerate.
struct DltArgs {
    const float *obj, *img;
    int n_host;
    const int *d_n;
    K4 K;
    double *hyp;
    int *status;
};

template <int KN> __device__ __forceinline__ void block_sum_fixed(double (&v)[KN], double *s_red /* [4][KN] */)
{
#pragma unroll
    for (int k = 0; k < KN; k++) {
        double x = v[k];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1)
            x = x + __shfl_xor(x, m, 64);  // balanced tree; commutative adds: every lane holds the same bits
        v[k] = x;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < KN; k++)
            s_red[wave * KN + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KN; k++)
        v[k] = ((s_red[k] + s_red[KN + k]) + s_red[2 * KN + k]) + s_red[3 * KN + k];
}

__global__ __launch_bounds__(256) void pnp_dlt_kernel(DltArgs a)
{
    __shared__ double s_red[4 * 40], s_A[144], s_V[144], s_cs[16];
    __shared__ int s_pq[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int n = a.d_n ? min(*a.d_n, a.n_host) : a.n_host;
    const float2 *__restrict__ img = reinterpret_cast<const float2 *>(a.img);
    double acc[40];
#pragma unroll
    for (int k = 0; k < 40; k++)
        acc[k] = 0;
    const double ifx = 1. / a.K.fx, ify = 1. / a.K.fy;
    for (int i = tid; i < n; i += 256) {
        const double P[4] = {(double)a.obj[3 * i], (double)a.obj[3 * i + 1], (double)a.obj[3 * i + 2], 1.};
        const float2 u = img[i];
        const double x = -(((double)u.x - a.K.cx) * ifx), y = -(((double)u.y - a.K.cy) * ify);
        const double w3 = x * x + y * y;
        int k = 0;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = r; c < 4; c++) {
                const double pp = P[r] * P[c];
                acc[k] += pp;
                acc[10 + k] += x * pp;
                acc[20 + k] += y * pp;
                acc[30 + k] += w3 * pp;
                k++;
            }
    }
    block_sum_fixed<40>(acc, s_red);
    if (tid >= 64)
        return;  // wave 0 goes on alone: no workgroup barrier below
    int status = 0;
    if (n < 6)
        status = 2;
    // planarity test of upstream: eigenvalues of the centred second moments, W[2] / W[1] < 1e-3
    if (status == 0) {
        const double inv_n = 1. / n;
        const double mx = acc[3] * inv_n, my = acc[6] * inv_n, mz = acc[8] * inv_n;  // sums of X, Y, Z (P P^T entries 03, 13, 23)
        double C[9] = {acc[0] - n * mx * mx, acc[1] - n * mx * my, acc[2] - n * mx * mz,
                       acc[1] - n * mx * my, acc[4] - n * my * my, acc[5] - n * my * mz,
                       acc[2] - n * mx * mz, acc[5] - n * my * mz, acc[7] - n * mz * mz};
        double U[9], V[9];
        svd3(C, U, V);
        // singular values of the symmetric PSD matrix = |column of C V| in decreasing order
        double w[3];
        for (int c = 0; c < 3; c++) {
            double q = 0;
            for (int r = 0; r < 3; r++) {
                const double e = C[3 * r] * V[c] + C[3 * r + 1] * V[3 + c] + C[3 * r + 2] * V[6 + c];
                q += e * e;
            }
            w[c] = sqrt(q);
        }
        if (!(w[1] > 0) || w[2] / w[1] < 1e-3)
            status = 1;
        if (status == 1 && w[1] > 0) {
            // ---- upstream's PLANAR branch (cvFindExtrinsicCameraParams2): all object points in one plane ----
            // R_transform = V^T of the second moments (rows: the plane's two axes, then its normal), identity when
            // the plane is z = const already; T_transform = -R_transform * mean; homography from the in-plane
            // coordinates to the normalised image points by the normalised DLT of findHomography(method 0: centroid +
            // mean absolute deviation scaling, 9x9 L^T L, eigenvector of the smallest eigenvalue); [h1 h2 h1 x h2]
            // orthonormalised (the Rodrigues round trip upstream = U V^T), t = 2 h3 / (|h1| + |h2|); back through
            // the plane transform.  (Upstream polishes the homography with ten LM steps of its own before the
            // decomposition; the pose refinement that follows here minimises the same reprojection error.)
            double Rp[9];
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
            double Tp[3];
            for (int r = 0; r < 3; r++)
                Tp[r] = -(Rp[3 * r] * mx + Rp[3 * r + 1] * my + Rp[3 * r + 2] * mz);
            auto wave_sum = [&](double x) {
#pragma unroll
                for (int m = 1; m < 64; m <<= 1)
                    x = x + __shfl_xor(x, m, 64);
                return x;
            };
            auto plane_xy = [&](int i, double &X, double &Y, double &u, double &v) {
                const double px = a.obj[3 * i], py = a.obj[3 * i + 1], pz = a.obj[3 * i + 2];
                X = Rp[0] * px + Rp[1] * py + Rp[2] * pz + Tp[0];
                Y = Rp[3] * px + Rp[4] * py + Rp[5] * pz + Tp[1];
                const float2 o = img[i];
                u = ((double)o.x - a.K.cx) * ifx;
                v = ((double)o.y - a.K.cy) * ify;
            };
            double c4[4] = {0, 0, 0, 0};  // centroids: X, Y, u, v
            for (int i = lane; i < n; i += 64) {
                double X, Y, u, v;
                plane_xy(i, X, Y, u, v);
                c4[0] += X;
                c4[1] += Y;
                c4[2] += u;
                c4[3] += v;
            }
            for (int k = 0; k < 4; k++)
                c4[k] = wave_sum(c4[k]) * inv_n;
            double d4[4] = {0, 0, 0, 0};  // mean absolute deviations
            for (int i = lane; i < n; i += 64) {
                double X, Y, u, v;
                plane_xy(i, X, Y, u, v);
                d4[0] += fabs(X - c4[0]);
                d4[1] += fabs(Y - c4[1]);
                d4[2] += fabs(u - c4[2]);
                d4[3] += fabs(v - c4[3]);
            }
            bool ok = true;
            for (int k = 0; k < 4; k++) {
                d4[k] = wave_sum(d4[k]);
                ok = ok && d4[k] > 2.220446049250313e-16;
            }
            if (ok) {
                double sc4[4];
                for (int k = 0; k < 4; k++)
                    sc4[k] = n / d4[k];
                double L45[45];
#pragma unroll
                for (int k = 0; k < 45; k++)
                    L45[k] = 0;
                for (int i = lane; i < n; i += 64) {
                    double X, Y, u, v;
                    plane_xy(i, X, Y, u, v);
                    X = (X - c4[0]) * sc4[0];
                    Y = (Y - c4[1]) * sc4[1];
                    u = (u - c4[2]) * sc4[2];
                    v = (v - c4[3]) * sc4[3];
                    const double Lx[9] = {X, Y, 1, 0, 0, 0, -u * X, -u * Y, -u};
                    const double Ly[9] = {0, 0, 0, X, Y, 1, -v * X, -v * Y, -v};
                    int k = 0;
#pragma unroll
                    for (int r = 0; r < 9; r++)
#pragma unroll
                        for (int c = r; c < 9; c++)
                            L45[k++] += Lx[r] * Lx[c] + Ly[r] * Ly[c];
                }
                {
                    int k = 0;
                    for (int r = 0; r < 9; r++)
                        for (int c = r; c < 9; c++) {
                            const double t = wave_sum(L45[k++]);
                            if (lane == 0) {
                                s_A[9 * r + c] = t;
                                s_A[9 * c + r] = t;
                            }
                        }
                }
                wave_lds_fence();
                wave_jacobi_eigen_sym<9>(s_A, s_V, s_cs, s_pq, 10, lane);
                int best = 0;
                for (int e = 1; e < 9; e++)
                    if (s_A[10 * e] < s_A[10 * best])
                        best = e;
                double H0[9], H1[9], H[9];
                for (int i = 0; i < 9; i++)
                    H0[i] = s_V[9 * i + best];
                // H = inv(T_image) * H0 * T_plane
                const double Ti[9] = {1. / sc4[2], 0, c4[2], 0, 1. / sc4[3], c4[3], 0, 0, 1};
                const double Tm[9] = {sc4[0], 0, -c4[0] * sc4[0], 0, sc4[1], -c4[1] * sc4[1], 0, 0, 1};
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++)
                        H1[3 * r + c] = H0[3 * r] * Tm[c] + H0[3 * r + 1] * Tm[3 + c] + H0[3 * r + 2] * Tm[6 + c];
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++)
                        H[3 * r + c] = Ti[3 * r] * H1[c] + Ti[3 * r + 1] * H1[3 + c] + Ti[3 * r + 2] * H1[6 + c];
                if (fabs(H[8]) > 2.220446049250313e-16) {
                    const double ih = 1. / H[8];
                    for (int i = 0; i < 9; i++)
                        H[i] *= ih;
                    const double n1 = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]);
                    const double n2 = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
                    const double i1 = 1. / fmax(n1, 2.220446049250313e-16), i2 = 1. / fmax(n2, 2.220446049250313e-16);
                    const double it = 2. / fmax(n1 + n2, 2.220446049250313e-16);
                    const double h1[3] = {H[0] * i1, H[3] * i1, H[6] * i1}, h2[3] = {H[1] * i2, H[4] * i2, H[7] * i2};
                    const double th[3] = {H[2] * it, H[5] * it, H[8] * it};
                    const double h3[3] = {h1[1] * h2[2] - h1[2] * h2[1], h1[2] * h2[0] - h1[0] * h2[2],
                                          h1[0] * h2[1] - h1[1] * h2[0]};
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
                    bool fin = true;
                    for (int i = 0; i < 9; i++)
                        fin = fin && isfinite(R[i]);
                    for (int i = 0; i < 3; i++)
                        fin = fin && isfinite(t[i]);
                    if (fin) {
                        if (lane == 0) {
                            for (int i = 0; i < 9; i++)
                                a.hyp[i] = R[i];
                            for (int i = 0; i < 3; i++)
                                a.hyp[9 + i] = t[i];
                        }
                        status = 4;  // planar start written; reported as a solution (status 0) below
                    }
                }
            }
        }
    }
    if (status == 4) {
        status = 0;
    } else if (status == 0) {
        // L^T L, row-major 12x12, from the four weighted second-moment blocks
        for (int e = lane; e < 144; e += 64) {
            const int r = e / 12, c = e - 12 * r;
            const int br = r >> 2, bc = c >> 2, i = r & 3, j = c & 3;
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            const int k = lo * 4 - lo * (lo - 1) / 2 + (hi - lo);  // index of (lo, hi) in the 10 unique entries
            int blk = -1;  // 0: S1, 1: Sx, 2: Sy, 3: Sxy
            if (br == bc)
                blk = br == 2 ? 3 : 0;
            else if (br + bc == 2 && (br == 2 || bc == 2))  // (0,2) / (2,0)
                blk = 1;
            else if (br + bc == 3)                          // (1,2) / (2,1)
                blk = 2;
            s_A[e] = blk < 0 ? 0. : acc[10 * blk + k];
        }
        wave_lds_fence();
        wave_jacobi_eigen_sym<12>(s_A, s_V, s_cs, s_pq, 10, lane);
        int best = 0;
        for (int e = 1; e < 12; e++)
            if (s_A[13 * e] < s_A[13 * best])
                best = e;
        double rt[12];  // 3x4 [RR | tt], row-major
        for (int i = 0; i < 12; i++)
            rt[i] = s_V[12 * i + best];
        double RR[9] = {rt[0], rt[1], rt[2], rt[4], rt[5], rt[6], rt[8], rt[9], rt[10]};
        double tt[3] = {rt[3], rt[7], rt[11]};
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
        if (!(sc > 2.220446049250313e-16)) {
            status = 3;
        } else {
            double U[9], V[9], R[9];
            svd3(RR, U, V);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++)
                    R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
            const double scale = sqrt(3.) / sc;  // cvNorm(R) / sc, |R|_F of a rotation = sqrt(3)
            if (lane == 0) {
                for (int i = 0; i < 9; i++)
                    a.hyp[i] = R[i];
                for (int i = 0; i < 3; i++)
                    a.hyp[9 + i] = tt[i] * scale;
            }
        }
    }
    if (lane == 0)
        *a.status = status;
}

struct PnpResult {      // what the host reads back after a localisation
    double rvec[3], tvec[3];
    double R[9];        // Rodrigues(rvec)
    double rms;
    int n_inliers, iters_run;
};

// One thread: the tracked-point count behind the record (PnpRecord::n_tracked sits right behind the result).
__device__ void publish_record(const PnpJob &job, const PnpResult &r, PnpResult *d_out)
{
    (void)r;
    const int n_tracked = job.cnt_trk ? *job.cnt_trk : 0;
    reinterpret_cast<int *>(d_out + 1)[0] = n_tracked;
}

// The tail of solvePnPRansac in ONE single-workgroup launch: (1) thread 0 replays the sequential
// RANSAC loop over the scored hypotheses and publishes the inlier count for the host policy;
// (2) the workgroup evaluates the winning hypothesis on every point, writes the mask and the
// order-preserving inlier index list (ballot scan); (3) Levenberg-Marquardt refinement over the
// inlier list.
// LEAN: capped at 96 VGPRs (with spills) so that the four-wave workgroup starts beside a full tracking launch
// of another context -- the lock-step groups; a lone chunk has the chip to itself between its tracking
// launches and takes the 128-VGPR build without spills (45 us faster alone).
// mode (chain runner, one chunk per GPU): PNP_FINISH_ALL does everything; PNP_FINISH_DECIDE stops after the frame's
// policy, the inlier list and the hand-over -- what the next frame's tracking pass and filters wait for -- and leaves
// the refinement to a later launch on another stream: PNP_FINISH_REFINE_KF runs only for a keyframe (the placement of
// its cloud waits for that pose, nothing else does), PNP_FINISH_REFINE_PLAIN only for any other frame (nothing in the
// next frame waits for it).
enum { PNP_FINISH_ALL = 0, PNP_FINISH_DECIDE = 1, PNP_FINISH_REFINE_KF = 2, PNP_FINISH_REFINE_PLAIN = 3 };
template <bool LEAN> __global__ __launch_bounds__(256, LEAN ? 5 : 4) void pnp_finish_kernel(PnpBatchN<LEAN ? SVO_LK_MAX_JOBS : 1> batch, int mode)
{
    svo_chain_priority();
    const PnpJob &job = batch.j[blockIdx.x];  // one workgroup per job
    const float *__restrict__ obj = job.obj;
    const float2 *__restrict__ img = reinterpret_cast<const float2 *>(job.img);
    const int n_host = job.n_host;
    const int *__restrict__ d_n = job.d_n;
    const K4 K = job.K;
    RansacState *__restrict__ st = job.st;
    const int iterations = job.iterations;
    const double confidence = job.confidence;
    const int *__restrict__ nmodels = job.nmodels;
    const int *__restrict__ counts = job.counts;
    const double *__restrict__ hyp = job.hyp;
    const float thr = job.thr;
    uint8_t *__restrict__ mask = job.mask;
    int *__restrict__ inl = job.inl;
    int *__restrict__ d_m = job.d_m;
    const int max_iters = job.max_lm_iters;
    PnpResult *__restrict__ out = job.out;
    VoChain *chain = job.chain;
    const bool refine_only = mode == PNP_FINISH_REFINE_KF || mode == PNP_FINISH_REFINE_PLAIN;
    if (refine_only) {
        if (!chain || chain->refine_due == 0 || (chain->kf != 0) != (mode == PNP_FINISH_REFINE_KF))
            return;  // the frame never got that far, or it is the other launch's
    } else if (chain && chain->run == 0) {
        return;  // the chain halted at an earlier frame
    }
    __shared__ double s_all[RED_CHUNK * RED_STRIDE], s_part[4 * NACC], s_sum[NACC], s_pose[12], s_trial[12];
    __shared__ double s_norm[NACC];  // the normal equations at the accepted pose (packed like the accumulators)
    __shared__ int s_flag, s_wave[4], s_base;
    __shared__ RansacState s_state;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = d_n ? min(*d_n, n_host) : n_host;
    if (tid == 0 && !refine_only) {
        RansacState r;
        if (job.direct) {  // cv::solvePnP: every point takes part, the start is the DLT pose
            r.niters = r.next_iter = r.iters_run = 0;
            r.best_iter = (*job.dlt_status == 0 && n >= 6) ? 0 : -1;
            r.best_model = 0;
            r.best_count = r.best_iter == 0 ? n : 0;
            r.done = 1;
            r.pad = 0;
        } else {
            r = ransac_replay<1>(nullptr, 1, iterations, iterations, n, confidence, nmodels, counts, MP);
        }
        *st = r;
        s_state = r;
        s_base = 0;
        s_flag = 0;
        if (chain) {
            // The reference's host policy, on the device.  < 10 inliers at 1 px: PerspectiveNpointEstimation tries
            // again at 8 px (src/keyFrameManagement.cpp:85-92) -- rare; the chain stops HERE with the tracked sets of
            // this frame in place and the host runs the retry.  Otherwise: keyframe iff fewer than 200 inliers
            // (src/VisualSLAM.cpp:120).
            const int ninl = r.best_iter >= 0 ? r.best_count : 0;
            if (ninl < chain->retry_below) {
                chain->run = 0;
                chain->kf = 0;
                chain->halt_code = SVO_HALT_RETRY;
                s_flag = 3;
            } else {
                chain->kf = ninl < chain->kf_min ? 1 : 0;
            }
        }
    }
    if (tid == 0 && refine_only) {
        s_state = *st;  // what the deciding launch left
        s_base = *d_m;
        s_flag = 0;
    }
    __syncthreads();
    if (s_flag == 3)
        return;  // halted: the host takes over at this frame
    const RansacState s = s_state;
    const bool have_model = s.best_iter >= 0 && s.best_count > 0;
    if (!refine_only) {
        double P[12];
#pragma unroll
        for (int k = 0; k < 12; k++)
            P[k] = have_model ? hyp[(size_t)s.best_iter * 12 + k] : 0.;
        const int n_pass = mask ? n_host : n;  // the mask covers the capacity; the index list only the live points
        for (int start = 0; start < n_pass; start += 256) {
            const int i = start + tid;
            bool keep = false;
            if (have_model && i < n) {
                const float2 u = img[i];
                keep = job.direct || reproj_err_sq(P, K, obj[3 * i], obj[3 * i + 1], obj[3 * i + 2], u.x, u.y) <= thr;
            }
            if (mask && i < n_host)
                mask[i] = keep ? 1 : 0;
            const unsigned long long bal = __ballot(keep);
            const int below = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0)
                s_wave[wave] = __popcll(bal);
            __syncthreads();
            int wbase = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int c = s_wave[w];
                wbase += w < wave ? c : 0;
                total += c;
            }
            const int base = s_base;
            if (keep)
                inl[base + wbase + below] = i;
            __syncthreads();
            if (tid == 0)
                s_base = base + total;
            __syncthreads();
        }
    }
    const int m = s_base;
    if (tid == 0 && !refine_only)
        *d_m = m;
    if (chain && !refine_only) {
        // The frame's policy is known: its counts go to the per-frame record; no keyframe -> the tracked sets become
        // the reference sets (src/VisualSLAM.cpp:143-146; a keyframe's sets are written by the stereo path that
        // follows).  The refinement below reads the tracked sets and the inlier list, not the reference sets.
        const int kf = chain->kf;  // written by thread 0 before the barriers of the pass above
        if (kf == 0) {
            const float2 *__restrict__ s2 = img;  // the job's point sets ARE the frame's tracked sets
            float2 *__restrict__ d2 = reinterpret_cast<float2 *>(chain->ref2d);
            const float *__restrict__ s3 = obj;
            float *__restrict__ d3 = chain->ref3d;
            for (int i = tid; i < n; i += 256)
                d2[i] = s2[i];
            for (int i = tid; i < 3 * n; i += 256)
                d3[i] = s3[i];
        }
        if (tid == 0) {
            VoOut *o = chain->out + chain->frame;
            o->inliers = s.best_count;
            o->tracked = n;
            o->keyframe = kf;
            o->pad = 0;
            if (kf == 0) {
                chain->nref = n;
                if (n < 5) {  // svo_vo_run_chunk: "tracking lost: n reference points" at the next frame
                    chain->run = 0;
                    chain->halt_code = SVO_HALT_FEW_REF;
                }
            }
            chain->refine_due = mode == PNP_FINISH_DECIDE ? 1 : 0;
        }
        if (mode == PNP_FINISH_DECIDE)
            return;  // a refining launch finishes the frame beside what follows
    }
    if (!have_model || m <= 0) {
        if (tid == 0) {
            PnpResult r;
            memset(&r, 0, sizeof(r));
            r.iters_run = s.iters_run;
            *out = r;
            publish_record(job, r, out);
        }
        return;
    }
    __syncthreads();  // the index list written above is read by other threads below
    if (tid < 12)
        s_pose[tid] = hyp[(size_t)s.best_iter * 12 + tid];
    __syncthreads();
    double acc[NACC];
    auto accumulate = [&](const double *pose, bool jac) {
#pragma unroll
        for (int k = 0; k < NACC; k++)
            acc[k] = 0;
        for (int e = tid; e < m; e += 256)
            pnp_point_terms(obj, img, inl[e], K, pose, pose + 9, jac, acc);
        if (jac)
            block_reduce<NACC>(acc, s_all, s_part, s_sum);
        else
            block_reduce<1>(acc, s_all, s_part, s_sum);  // the trial pass only needs the error
    };
    accumulate(s_pose, true);
    double err = s_sum[0], lambda = 1e-3;
    // The accepted normal equations stay in LDS (only thread 0 solves with them): as 42 doubles in every
    // thread's registers they put the kernel over the 96 VGPRs that let its workgroup start beside a full
    // tracking launch.
    auto keep_normal_equations = [&]() {
        if (tid < NACC)
            s_norm[tid] = s_sum[tid];
    };
    keep_normal_equations();
    __syncthreads();
    for (int it = 0; it < max_iters; it++) {
        // thread 0 proposes a step; flag: 0 = trial pose ready, 1 = solve failed (raise lambda), 2 = stop
        if (tid == 0) {
            double A[36], nb[6], d[6];
            {
                int k = 1;
                for (int p = 0; p < 6; p++)
                    for (int q = p; q < 6; q++) {
                        A[6 * p + q] = s_norm[k];
                        A[6 * q + p] = s_norm[k];
                        k++;
                    }
            }
            for (int k = 0; k < 6; k++) {
                const double dk = A[7 * k];
                A[7 * k] = dk + (lambda * dk + 1e-300);
                nb[k] = -s_norm[22 + k];
            }
            int flag = 0;
            if (!chol6_solve(A, nb, d)) {
                flag = lambda * 10 > 1e12 ? 2 : 1;
            } else {
                double dR[9];
                rodrigues_dev(d, dR);
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++)
                        s_trial[3 * i + j] =
                            dR[3 * i] * s_pose[j] + dR[3 * i + 1] * s_pose[3 + j] + dR[3 * i + 2] * s_pose[6 + j];
                for (int k = 0; k < 3; k++)
                    s_trial[9 + k] = s_pose[9 + k] + d[3 + k];
                s_part[0] = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
            }
            s_flag = flag;
        }
        __syncthreads();
        const int flag = s_flag;
        if (flag == 2)
            break;
        if (flag == 1) {
            lambda *= 10;
            __syncthreads();
            continue;
        }
        const double step = s_part[0];
        __syncthreads();
        // one pass gives the trial error AND, if the step is accepted, the normal equations there
        accumulate(s_trial, true);
        const double e2 = s_sum[0];
        if (e2 < err || !(err == err)) {
            if (tid < 12)
                s_pose[tid] = s_trial[tid];
            const double prev = err;
            err = e2;
            keep_normal_equations();
            lambda *= 0.1;
            if (lambda < 1e-12)
                lambda = 1e-12;
            __syncthreads();
            // CvLevMarq's criterion as cv::solvePnP sets it (20 iterations, FLT_EPSILON): the relative change of
            // the parameter vector, |step|^2 <= eps^2 * (1 + |t|^2) here (the increment is on the left, its norm is
            // the norm of the change; the 1 keeps a pose at the origin from iterating to the cap), or an error that
            // no longer decreases
            const double scale = 1. + s_pose[9] * s_pose[9] + s_pose[10] * s_pose[10] + s_pose[11] * s_pose[11];
            if (step <= 1.4210854715202004e-14 * scale || prev - err <= 1e-10 * prev)
                break;
        } else {
            lambda *= 10;
            if (lambda > 1e12)
                break;
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) {
        PnpResult r;
        rodrigues_inv_dev(s_pose, r.rvec);
        // R of the record is Rodrigues(rvec) -- what the reference forms from solvePnPRansac's output
        // (src/VisualSLAM.cpp:71) -- not the refinement's own matrix, which differs from it in the last places
        rodrigues_dev(r.rvec, r.R);
        r.tvec[0] = s_pose[9];
        r.tvec[1] = s_pose[10];
        r.tvec[2] = s_pose[11];
        r.rms = sqrt(err / m);
        r.n_inliers = m;
        r.iters_run = s.iters_run;
        *out = r;
        publish_record(job, r, out);
        if (chain) {
            // Rodrigues; R = R^T; t = -R * tvec (src/VisualSLAM.cpp:70-74): the same operations, in the same order, as
            // the host code of svo_vo_localize
            VoOut *o = chain->out + chain->frame;
            double R9[9];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++)
                    R9[3 * i + j] = r.R[3 * j + i];
            for (int i = 0; i < 3; i++) {
                const double ti = -(R9[3 * i] * r.tvec[0] + R9[3 * i + 1] * r.tvec[1] + R9[3 * i + 2] * r.tvec[2]);
                chain->t[i] = ti;
                o->t[i] = ti;
            }
            for (int k = 0; k < 9; k++) {
                chain->R[k] = R9[k];
                o->R[k] = R9[k];
            }
            chain->refine_due = 0;
            chain->frame = chain->frame + 1;  // the frame is finished
        }
    }
}

}  // namespace

static_assert(sizeof(PnpResult) == 17 * 8, "PnpResult layout");

// the kernels' view of the jobs (workspace: hypotheses in w_c, state + counters in w_d, per job); returns the number
// of jobs with points, < 0 on an error
static int pnp_fill_batch(svo_ctx *ctx, int n_jobs, const svo_pnp_job *jobs, PnpBatch &batch, int *it_max_out)
{
    if (n_jobs > SVO_LK_MAX_JOBS) {
        svo_set_error("pnp: at most %d jobs per launch", SVO_LK_MAX_JOBS);
        return SVO_ERR_ARG;
    }
    int it_max = 1;
    for (int k = 0; k < n_jobs; k++)
        it_max = jobs[k].iterations > it_max ? jobs[k].iterations : it_max;
    *it_max_out = it_max;
    const size_t h_stride = (size_t)it_max * 12, i_stride = ((size_t)it_max * 2 + 32 + 15) / 16 * 16;
    int rc;
    if ((rc = ctx->w_c.ensure(h_stride * sizeof(double) * n_jobs)) || (rc = ctx->w_d.ensure(i_stride * sizeof(int) * n_jobs)))
        return rc;
    int nb = 0;
    for (int k = 0; k < n_jobs; k++) {
        const svo_pnp_job &h = jobs[k];
        if (h.cap <= 0)
            continue;
        PnpJob &j = batch.j[nb];
        const int iterations = h.iterations < 1 ? 1 : h.iterations;
        int *ib = ctx->w_d.as<int>() + i_stride * nb;
        j.obj = h.obj;
        j.img = h.img;
        j.n_host = h.cap;
        j.d_n = h.d_n;
        j.K = {h.K4[0], h.K4[1], h.K4[2], h.K4[3]};
        j.seed = h.seed;
        j.iterations = iterations;
        j.confidence = h.confidence;
        j.thr = (float)(h.reproj_err * h.reproj_err);
        j.max_lm_iters = h.refine_iters > 0 ? h.refine_iters : 20;
        j.st = reinterpret_cast<RansacState *>(ib);
        j.d_m = ib + 8;  // inlier count of the winning hypothesis
        j.nmodels = ib + 16;
        j.counts = j.nmodels + iterations;
        j.hyp = ctx->w_c.as<double>() + h_stride * nb;
        j.mask = h.mask;
        j.inl = h.inliers;
        j.out = reinterpret_cast<PnpResult *>(h.d_result);
        j.direct = 0;
        j.dlt_status = nullptr;
        j.chain = h.chain;
        j.cnt_trk = h.cnt_trk;
        j.ticket = ctx->d_tickets + 16 + nb;  // slots 0..15: fransac.hip
        nb++;
    }
    for (int k = nb; k < SVO_LK_MAX_JOBS; k++)
        batch.j[k] = batch.j[0];
    return nb;
}

// Device-pointer form, several problems in one set of launches.  inliers: cap ints; d_result: one
// PnpResult (136 bytes): rvec[3] tvec[3] R[9] rms (doubles) n_inliers iters_run (ints).
int svo_launch_pnp_ransac_batch(svo_ctx *ctx, int n_jobs, const svo_pnp_job *jobs, bool split)
{
    if (n_jobs <= 0)
        return SVO_OK;
    PnpBatch batch;
    int it_max = 1;
    const int nb = pnp_fill_batch(ctx, n_jobs, jobs, batch, &it_max);
    if (nb <= 0)
        return nb;
    PnpBatchN<1> one;
    one.j[0] = batch.j[0];
    ScopedKernelTime tm(ctx, SVO_K_PNP);
    // single-wave workgroups: they get wave slots beside a tracking launch as soon as one frees
    const int bounds[3] = {0, it_max < PNP_PHASE_A ? it_max : PNP_PHASE_A, it_max};
    for (int ph = 0; ph < 2; ph++)
        if (bounds[ph + 1] > bounds[ph]) {  // solves AND scores; the second phase usually leaves at once
            if (nb > 1)
                hipLaunchKernelGGL(pnp_solve_kernel<true>, dim3(bounds[ph + 1] - bounds[ph], nb), dim3(64), 0, ctx->stream,
                                   batch, bounds[ph], bounds[ph + 1]);
            else
                hipLaunchKernelGGL(pnp_solve_kernel<false>, dim3(bounds[ph + 1] - bounds[ph], nb), dim3(64), 0, ctx->stream,
                                   one, bounds[ph], bounds[ph + 1]);
        }
    const int mode = split ? PNP_FINISH_DECIDE : PNP_FINISH_ALL;
    if (nb > 1)
        hipLaunchKernelGGL(pnp_finish_kernel<true>, dim3(nb), dim3(256), 0, ctx->stream, batch, mode);
    else
        hipLaunchKernelGGL(pnp_finish_kernel<false>, dim3(nb), dim3(256), 0, ctx->stream, one, mode);
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

int svo_launch_pnp_ransac_batch(svo_ctx *ctx, int n_jobs, const svo_pnp_job *jobs)
{
    return svo_launch_pnp_ransac_batch(ctx, n_jobs, jobs, false);
}

// the refinement a split launch left undone; same jobs, same workspace.  keyframes: refine only the frames that turned
// out to be keyframes / only those that did not (the two go to different places in the caller's streams)
int svo_launch_pnp_refine(svo_ctx *ctx, int n_jobs, const svo_pnp_job *jobs, bool keyframes)
{
    if (n_jobs <= 0)
        return SVO_OK;
    PnpBatch batch;
    int it_max = 1;
    const int nb = pnp_fill_batch(ctx, n_jobs, jobs, batch, &it_max);
    if (nb <= 0)
        return nb;
    ScopedKernelTime tm(ctx, SVO_K_PNP);
    const int mode = keyframes ? PNP_FINISH_REFINE_KF : PNP_FINISH_REFINE_PLAIN;
    if (nb > 1)
        hipLaunchKernelGGL(pnp_finish_kernel<true>, dim3(nb), dim3(256), 0, ctx->stream, batch, mode);
    else {
        PnpBatchN<1> one;
        one.j[0] = batch.j[0];
        hipLaunchKernelGGL(pnp_finish_kernel<false>, dim3(nb), dim3(256), 0, ctx->stream, one, mode);
    }
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

int svo_launch_pnp_ransac(svo_ctx *ctx, const float *obj, const float *img, int cap, const int *d_n,
                          const double *K4h, int iterations, double reproj_err, double confidence, uint64_t seed,
                          int refine_iters, int *inliers, uint8_t *mask, void *d_result)
{
    svo_pnp_job j;
    j.obj = obj;
    j.img = img;
    j.cap = cap;
    j.d_n = d_n;
    for (int k = 0; k < 4; k++)
        j.K4[k] = K4h[k];
    j.iterations = iterations;
    j.reproj_err = reproj_err;
    j.confidence = confidence;
    j.seed = seed;
    j.refine_iters = refine_iters;
    j.inliers = inliers;
    j.mask = mask;
    j.d_result = d_result;
    j.chain = nullptr;
    j.cnt_trk = nullptr;
    return svo_launch_pnp_ransac_batch(ctx, 1, &j);
}

// cv::solvePnP (ITERATIVE, no guess) on device arrays: DLT + LM over all points.  inliers: cap ints of
// scratch (receives 0..n-1); d_result: one PnpResult (n_inliers = n, or 0 when there is no model).
int svo_launch_solve_pnp(svo_ctx *ctx, const float *obj, const float *img, int cap, const int *d_n, const double *K4h,
                         int refine_iters, int *inliers, void *d_result)
{
    if (cap <= 0)
        return SVO_OK;
    int rc;
    if ((rc = ctx->w_c.ensure(12 * sizeof(double))) || (rc = ctx->w_d.ensure(64 * sizeof(int))))
        return rc;
    int *ib = ctx->w_d.as<int>();
    DltArgs d;
    d.obj = obj;
    d.img = img;
    d.n_host = cap;
    d.d_n = d_n;
    d.K = {K4h[0], K4h[1], K4h[2], K4h[3]};
    d.hyp = ctx->w_c.as<double>();
    d.status = ib + 12;
    PnpBatchN<1> batch;
    PnpJob &j = batch.j[0];
    j.obj = obj;
    j.img = img;
    j.n_host = cap;
    j.d_n = d_n;
    j.K = d.K;
    j.seed = 0;
    j.iterations = 1;
    j.confidence = 0.99;
    j.thr = 0.f;
    j.max_lm_iters = refine_iters > 0 ? refine_iters : 20;
    j.st = reinterpret_cast<RansacState *>(ib);
    j.d_m = ib + 8;
    j.nmodels = ib + 16;
    j.counts = ib + 17;
    j.hyp = d.hyp;
    j.mask = nullptr;
    j.inl = inliers;
    j.out = reinterpret_cast<PnpResult *>(d_result);
    j.direct = 1;
    j.dlt_status = d.status;
    j.chain = nullptr;
    j.cnt_trk = nullptr;
    ScopedKernelTime tm(ctx, SVO_K_PNP);
    hipLaunchKernelGGL(pnp_dlt_kernel, dim3(1), dim3(256), 0, ctx->stream, d);
    hipLaunchKernelGGL(pnp_finish_kernel<false>, dim3(1), dim3(256), 0, ctx->stream, batch, (int)PNP_FINISH_ALL);
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

extern "C" int svo_solve_pnp(svo_ctx *ctx, const float *obj, const float *img, int n, const double *K4h, double *rvec,
                             double *tvec, double *rms, int mem)
{
    SVO_CHECK_ARG(ctx && K4h && obj && img && rvec && tvec);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    if (n < 6) {
        svo_set_error("solvePnP (DLT) needs at least 6 points, got %d", n);
        return SVO_ERR_ARG;
    }
    int rc;
    if ((rc = ctx->s_d.ensure(sizeof(PnpResult) + 64)) || (rc = ctx->s_c.ensure((size_t)n * 4)))
        return rc;
    const float *dobj = obj, *dimg = img;
    if (mem == SVO_MEM_HOST) {
        if ((rc = ctx->s_a.ensure((size_t)n * 12)) || (rc = ctx->s_b.ensure((size_t)n * 8)))
            return rc;
        SVO_HIP(hipMemcpyAsync(ctx->s_a.p, obj, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
        SVO_HIP(hipMemcpyAsync(ctx->s_b.p, img, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        dobj = ctx->s_a.as<float>();
        dimg = ctx->s_b.as<float>();
    }
    if ((rc = svo_launch_solve_pnp(ctx, dobj, dimg, n, nullptr, K4h, 20, ctx->s_c.as<int>(), ctx->s_d.p)))
        return rc;
    SVO_HIP(hipMemcpyAsync(ctx->pinned, ctx->s_d.p, sizeof(PnpResult), hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    const PnpResult *r = reinterpret_cast<const PnpResult *>(ctx->pinned);
    if (r->n_inliers == 0) {
        svo_set_error("solvePnP: no initial pose (degenerate object points: collinear, coincident or behind the camera)");
        return SVO_ERR_STATE;
    }
    memcpy(rvec, r->rvec, sizeof(r->rvec));
    memcpy(tvec, r->tvec, sizeof(r->tvec));
    if (rms)
        *rms = r->rms;
    return SVO_OK;
}

extern "C" int svo_pnp_ransac(svo_ctx *ctx, const float *obj, const float *img, int n, const double *K4h,
                              int iterations, double reproj_err, double confidence, uint64_t seed, double *rvec,
                              double *tvec, int *inliers, int *n_inliers, int *iters_run, int mem)
{
    SVO_CHECK_ARG(ctx && K4h && n >= 0 && iterations > 0 && reproj_err > 0);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    if (n_inliers)
        *n_inliers = 0;
    if (iters_run)
        *iters_run = 0;
    if (n == 0)
        return SVO_OK;
    SVO_CHECK_ARG(obj && img && inliers && rvec && tvec);
    int rc;
    if ((rc = ctx->s_d.ensure(sizeof(PnpResult) + 64)))
        return rc;
    const float *dobj = obj, *dimg = img;
    int *dinl = inliers;
    if (mem == SVO_MEM_HOST) {
        if ((rc = ctx->s_a.ensure((size_t)n * 12)) || (rc = ctx->s_b.ensure((size_t)n * 8)) ||
            (rc = ctx->s_c.ensure((size_t)n * 4)))
            return rc;
        SVO_HIP(hipMemcpyAsync(ctx->s_a.p, obj, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
        SVO_HIP(hipMemcpyAsync(ctx->s_b.p, img, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        dobj = ctx->s_a.as<float>();
        dimg = ctx->s_b.as<float>();
        dinl = ctx->s_c.as<int>();
    }
    rc = svo_launch_pnp_ransac(ctx, dobj, dimg, n, nullptr, K4h, iterations, reproj_err, confidence, seed, 20, dinl,
                               nullptr, ctx->s_d.p);
    if (rc)
        return rc;
    // rvec / tvec / counts are host outputs in both modes (the caller decides on them)
    SVO_HIP(hipMemcpyAsync(ctx->pinned, ctx->s_d.p, sizeof(PnpResult), hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    const PnpResult *r = reinterpret_cast<const PnpResult *>(ctx->pinned);
    memcpy(rvec, r->rvec, sizeof(r->rvec));
    memcpy(tvec, r->tvec, sizeof(r->tvec));
    if (n_inliers)
        *n_inliers = r->n_inliers;
    if (iters_run)
        *iters_run = r->iters_run;
    if (mem == SVO_MEM_HOST && r->n_inliers > 0) {
        SVO_HIP(hipMemcpyAsync(inliers, dinl, (size_t)r->n_inliers * 4, hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SVO_OK;
}
