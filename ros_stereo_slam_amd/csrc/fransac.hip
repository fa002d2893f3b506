// fransac.hip -- fundamental-matrix RANSAC for gfx950.
//
// Replaces cv::findFundamentalMat(p1, p2, FM_RANSAC, thr, 0.99, mask) as the reference
// calls it at src/tracking.cpp:34 (FmatThresholding, 3 px) and src/tracking.cpp:75
// (PyrLKtrackFrame2Frame, method 8 == FM_RANSAC, 1 px).  Only the mask is consumed there.
//
// Structure (all on the context's stream, no host round trip): ONE kernel per phase,
//   one WAVEFRONT per RANSAC iteration (a wave of the second phase takes every 64th):
//     solve   lane 0: counter-based 7-sample (collinear samples re-drawn), 7x9 Gauss-Jordan in
//             LDS, cubic det(l*F1 + (1-l)*F2) = 0, up to three unit-norm models (LDS + HBM);
//     score   all lanes, model after model: the N correspondences strided over the 64 lanes,
//             symmetric epipolar distance in f64, inlier count reduced with DPP;
//     finish  the LAST wave to finish (ticket counter) replays the SEQUENTIAL loop over the
//             counts (first-best-wins, adaptive iteration bound), so the answer equals the serial
//             algorithm's, and -- once the loop has ended -- writes the mask of the winning model,
//             the model and the counts.
// Two phases (iterations [0,64) and [64,max)): the second launch (64 waves per problem) returns at
// once when the adaptive bound was reached in the first, which is the common case at VO inlier
// ratios (0.99 confidence, 85% inliers -> 12 iterations).  (Until round 2 this was five launches:
// solve, score, solve, score, mask -- each waiting its turn beside the tracking launches.)
#include "ransac_common.hip.h"
#include "svo_internal.h"

using namespace svo;

namespace {

constexpr int M = 7;
constexpr int PHASE_A = 64;
constexpr int PHASE_WAVES = 64;  // waves per problem and phase; a wave takes every PHASE_WAVES-th iteration

// One F-matrix RANSAC problem as the kernels see it; a launch may carry several (blockIdx.y picks
// the job) -- the chunks of a context that run in lock step (svo_vo_run_chunks).
struct FrJob {
    const float *p1, *p2;
    int n_host;
    const int *d_n;
    uint64_t seed;
    int max_iters;
    double confidence;
    float thr;
    RansacState *st;
    double *Fm;
    int *nmodels, *counts;
    unsigned *ticket;
    uint8_t *mask;
    double *Fbest;
    int *out_count, *out_iters;
    // order-preserving compaction by the mask, done by the finishing wave (c_in[0] == nullptr: none)
    const float *c_in[3];
    float *c_out[3];
    int c_stride[3];
    int *c_count;
    const int *gate;  // optional: every wave of the job leaves at once when *gate == 0 (chain runner)
    double *med;      // 3 doubles per iteration: the models' median errors (the least-median branch below 15 pairs)
    int cv_small;     // 1: cv::findFundamentalMat's small-sample behaviour (7 pairs: direct; 8..14: least median)
    int gate_stride;  // 0 in the product; svo_selftest_fransac_gate: workgroup b reads gate[b * gate_stride], so that ONE
                      // launch sees the gate both open and closed -- deterministically, not by a race between streams
};
template <int NJ> struct FrBatchN {  // NJ = 1: a chunk on its own (a sixteenth of the kernel arguments per launch)
    FrJob j[NJ];
};
using FrBatch = FrBatchN<SVO_LK_MAX_JOBS>;
static_assert(sizeof(FrBatch) + 16 <= 4096, "kernel arguments are limited to 4 KB");

__device__ __forceinline__ double det3(const double *m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
           m[2] * (m[3] * m[7] - m[4] * m[6]);
}

__device__ int solve_cubic(const double *c, double *r)
{
    int n = 0;
    const double a = c[0], b = c[1], cc = c[2], d = c[3];
    const double scale = fabs(a) + fabs(b) + fabs(cc) + fabs(d);
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
        const double a1 = b / a, a2 = cc / a, a3 = d / a;
        const double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        const double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        const double Qcubed = Q * Q * Q, dd = Qcubed - R * R;
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
    for (int k = 0; k < 3; k++)
        if (k < n)
            for (int it = 0; it < 2; it++) {  // Newton polish: irons out libm differences
                double x = r[k];
                double f = ((a * x + b) * x + cc) * x + d;
                double fp = (3 * a * x + 2 * b) * x + cc;
                if (fabs(fp) > 1e-300)
                    r[k] = x - f / fp;
            }
    return n;
}

// cv haveCollinearPoints: the LAST sample point against every pair of earlier ones
__device__ bool collinear_last(const float *__restrict__ p, const int (&idx)[M])
{
    const double xi = p[2 * idx[M - 1]], yi = p[2 * idx[M - 1] + 1];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < M - 1; j++) {
        const double dx1 = (double)p[2 * idx[j]] - xi, dy1 = (double)p[2 * idx[j] + 1] - yi;
#pragma unroll
        for (int k = 0; k < j; k++) {
            const double dx2 = (double)p[2 * idx[k]] - xi, dy2 = (double)p[2 * idx[k] + 1] - yi;
            if (fabs(dx2 * dy1 - dy2 * dx1) <=
                (double)FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                bad = true;
        }
    }
    return bad;
}

__device__ __forceinline__ double wave_max_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// One RANSAC iteration's models, by one WAVE: up to three unit-norm F (row-major) into Fk_out[27] (LDS, written
// by lane 0); returns their number (valid in lane 0), -1 when no sample could be drawn (the sequential loop
// stops there).  The sample is drawn by every lane alike; the 7x9 Gauss-Jordan elimination with full pivoting
// runs with ONE MATRIX ENTRY PER LANE (lane = 9 * row + column; as a loop of one lane over the matrix in LDS it
// was 1300 dependent LDS round trips, most of the kernel's time): every entry sees exactly the operations of
// the sequential elimination, in its order, so the result is the same bit for bit; the null-space tail (cubic,
// models) is lane 0's.  sA / sV / sPerm: LDS scratch (63 + 18 doubles, 9 ints).
// identity_sample: the sample is pairs 0..6 as they stand (findFundamentalMat on exactly seven pairs: no subset is drawn)
__device__ int fr_solve_wave(const float *__restrict__ p1, const float *__restrict__ p2, int n, uint64_t seed, int it,
                             double *sA, double *sV, int *sPerm, double *Fk_out, int lane, bool identity_sample = false)
{
    if (n < M)
        return -1;
    // ---- sample: M distinct indices, degenerate samples re-drawn ----
    int idx[M];
    bool ok = identity_sample;
    uint32_t draw = 0;
#pragma unroll
    for (int slot = 0; slot < M; slot++)
        idx[slot] = slot;
    for (int attempt = 0; attempt < kMaxAttempts && !ok; attempt++) {
        int guard = 0;
        bool filled = true;
#pragma unroll
        for (int slot = 0; slot < M; slot++) {
            int v = 0;
            bool got = false;
            while (!got && guard < kMaxDraws) {
                v = (int)(rng_u32(seed, (uint32_t)it, draw++) % (uint32_t)n);
                guard++;
                bool dup = false;
#pragma unroll
                for (int j = 0; j < M; j++)
                    if (j < slot && idx[j] == v)
                        dup = true;
                got = !dup;
            }
            if (!got)
                filled = false;
            idx[slot] = v;
        }
        if (!filled)
            break;
        ok = !collinear_last(p1, idx) && !collinear_last(p2, idx);
    }
    if (!ok)
        return -1;  // getSubset failed: the sequential loop stops here
    // ---- 7x9 epipolar system, one entry per lane; Gauss-Jordan with full pivoting ----
    const bool valid = lane < 63;
    const int li = valid ? lane / 9 : 7, lj = valid ? lane - 9 * li : 0;
    double a = 0;
    {
        int my = idx[0];
#pragma unroll
        for (int i = 1; i < M; i++)
            my = li == i ? idx[i] : my;
        const double u0 = p1[2 * my], v0 = p1[2 * my + 1];
        const double u1 = p2[2 * my], v1 = p2[2 * my + 1];
        const double row[9] = {u1 * u0, u1 * v0, u1, v1 * u0, v1 * v0, v1, u0, v0, 1.};
#pragma unroll
        for (int j = 0; j < 9; j++)
            a = lj == j ? row[j] : a;
        if (!valid)
            a = 0;
    }
    int permv = lane;  // lanes 0..8: the column permutation
    bool singular = false;
#pragma unroll 1
    for (int k = 0; k < M; k++) {
        // the first entry (row-major) of the largest magnitude in rows >= k, columns >= k
        const double v = (valid && li >= k && lj >= k) ? fabs(a) : -1.;
        const double best = wave_max_f64(v);
        if (best < 1e-12) {
            singular = true;
            break;
        }
        const int pl = __ffsll((unsigned long long)__ballot(v == best)) - 1;
        const int pr = pl / 9, pc = pl - 9 * pr;
        if (pr != k) {  // rows k <-> pr
            const int src = li == k ? pr * 9 + lj : (li == pr ? k * 9 + lj : lane);
            a = __shfl(a, src, 64);
        }
        if (pc != k) {  // columns k <-> pc
            const int src = lj == k ? li * 9 + pc : (lj == pc ? li * 9 + k : lane);
            a = __shfl(a, valid ? src : lane, 64);
            const int psrc = lane == k ? pc : (lane == pc ? k : lane);
            permv = __shfl(permv, psrc, 64);
        }
        const double inv = 1. / __shfl(a, k * 9 + k, 64);
        if (li == k)
            a *= inv;
        const double rowk = __shfl(a, k * 9 + lj, 64);                   // A[k][j], scaled
        const double f = __shfl(a, valid ? li * 9 + k : lane, 64);       // A[i][k]
        if (valid && li != k && f != 0)
            a -= f * rowk;
    }
    if (valid)
        sA[lane] = a;
    if (lane < 9)
        sPerm[lane] = permv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane != 0)
        return 0;  // the tail is lane 0's
    if (singular)
        return 0;
    for (int k = 0; k < M; k++) {
        const int c = sPerm[k];
        sV[c] = -sA[k * 9 + 7];
        sV[9 + c] = -sA[k * 9 + 8];
    }
    {
        const int c7 = sPerm[7], c8 = sPerm[8];
        sV[c7] = 1;
        sV[9 + c7] = 0;
        sV[c8] = 0;
        sV[9 + c8] = 1;
    }
    double G[9], H[9], Mx[9], c[4];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const double f1 = sV[i], f2 = sV[9 + i];
        G[i] = f1 - f2;
        H[i] = f2;
    }
    c[0] = det3(G);
    c[3] = det3(H);
    c[1] = 0;
    c[2] = 0;
#pragma unroll
    for (int row = 0; row < 3; row++) {
#pragma unroll
        for (int i = 0; i < 9; i++)
            Mx[i] = (i / 3 == row) ? H[i] : G[i];
        c[1] += det3(Mx);
#pragma unroll
        for (int i = 0; i < 9; i++)
            Mx[i] = (i / 3 == row) ? G[i] : H[i];
        c[2] += det3(Mx);
    }
    double roots[3] = {0, 0, 0};
    const int nr = solve_cubic(c, roots);
    int nm = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (k < nr) {
            double Fk[9], nrm = 0;
#pragma unroll
            for (int i = 0; i < 9; i++) {
                Fk[i] = H[i] + roots[k] * G[i];
                nrm += Fk[i] * Fk[i];
            }
            nrm = sqrt(nrm);
            if (nrm > 1e-300 && isfinite(nrm)) {
                double *dst = Fk_out + nm * 9;
#pragma unroll
                for (int i = 0; i < 9; i++)
                    dst[i] = Fk[i] / nrm;
                nm++;
            }
        }
    }
    return nm;
}

// a wave-uniform double as a scalar-register value
__device__ __forceinline__ double uniform_f64(double x)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// four consecutive correspondences of one lane
struct Quad {
    float x1[4], y1[4], x2[4], y2[4];
};
// pairs base .. base+3 (those below n; the rest read as 0): two 16-byte loads per array when the arrays are
// 16-byte aligned (`vec_ok`) and the four are inside, single pairs otherwise
__device__ __forceinline__ void load_quad(Quad &q, const float2 *__restrict__ p1, const float2 *__restrict__ p2, int base,
                                          int n, bool vec_ok)
{
    if (vec_ok && base + 3 < n) {
        const float4 a0 = *reinterpret_cast<const float4 *>(p1 + base), a1 = *reinterpret_cast<const float4 *>(p1 + base + 2);
        const float4 b0 = *reinterpret_cast<const float4 *>(p2 + base), b1 = *reinterpret_cast<const float4 *>(p2 + base + 2);
        q.x1[0] = a0.x, q.y1[0] = a0.y, q.x1[1] = a0.z, q.y1[1] = a0.w;
        q.x1[2] = a1.x, q.y1[2] = a1.y, q.x1[3] = a1.z, q.y1[3] = a1.w;
        q.x2[0] = b0.x, q.y2[0] = b0.y, q.x2[1] = b0.z, q.y2[1] = b0.w;
        q.x2[2] = b1.x, q.y2[2] = b1.y, q.x2[3] = b1.z, q.y2[3] = b1.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float2 a = {0.f, 0.f}, b = {0.f, 0.f};
            if (base + k < n) {
                a = p1[base + k];
                b = p2[base + k];
            }
            q.x1[k] = a.x, q.y1[k] = a.y, q.x2[k] = b.x, q.y2[k] = b.y;
        }
    }
}

// cv FMEstimatorCallback::computeError for one correspondence (float result)
__device__ __forceinline__ float f_error(const double (&F)[9], float x1, float y1, float x2, float y2)
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
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 > e2 ? e1 : e2);
}

// cv::findFundamentalMat below 15 pairs (oracle/geometry.c: orc_f_small has the statement and its source): seven pairs -- the
// 7-point solver once, the mask all ones; 8 .. 14 -- the least-median estimator: 300 iterations of 7-point samples
// (RANSACUpdateNumIters(confidence, 0.45, 7, 1000)), per model the median of the float errors, the first model with the
// smallest median wins, sigma = max(2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(median), 0.001), inliers err <= sigma^2, the
// result stands with at least seven.  Runs in the LAST launch of a call: every workgroup takes the iterations
// blockIdx.x, + gridDim.x, ... (wave 0 solves and takes the medians: n <= 14 lanes), the workgroup with the last ticket
// replays them in order and writes mask, model, counts and the compaction.  The threshold argument plays no part.
template <int NW>
__device__ void fr_small_case(const FrJob &job, int n, bool due, double *sA, double *sV, int *sPerm, double *sF, int *s_last,
                              int *s_all_due)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float2 *__restrict__ p1 = reinterpret_cast<const float2 *>(job.p1);
    const float2 *__restrict__ p2 = reinterpret_cast<const float2 *>(job.p2);
    const int niters = n == M ? 1 : update_num_iters(job.confidence, 0.45, M, 1000);
    float2 a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
    if (lane < n) {
        a1 = p1[lane];
        a2 = p2[lane];
    }
    if (due && wave == 0)
        for (int it = blockIdx.x; it < niters; it += gridDim.x) {
            const int nm0 = fr_solve_wave(job.p1, job.p2, n, job.seed, it, sA, sV, sPerm, sF, lane, n == M);
            const int nm = __builtin_amdgcn_readfirstlane(nm0);
            wave_lds_fence();  // lane 0's models, read by every lane below
            for (int k = 0; k < 3; k++) {
                if (k >= nm)
                    break;
                double F[9];
#pragma unroll
                for (int i = 0; i < 9; i++)
                    F[i] = sF[k * 9 + i];
                const float e = lane < n ? f_error(F, a1.x, a1.y, a2.x, a2.y) : 0.f;
                int rank = 0;  // place of this lane's error among the n (ties in lane order)
                for (int j = 0; j < n; j++) {
                    const float ej = __shfl(e, j, 64);
                    rank += (lane < n && (ej < e || (ej == e && j < lane))) ? 1 : 0;
                }
                const int hi = __ffsll((unsigned long long)__ballot(lane < n && rank == n / 2)) - 1;
                const int lo = __ffsll((unsigned long long)__ballot(lane < n && rank == n / 2 - 1)) - 1;
                const float eh = __shfl(e, hi, 64), el = __shfl(e, lo < 0 ? hi : lo, 64);
                const double median = (n & 1) ? (double)eh : (double)(el + eh) * 0.5;
                if (lane == 0) {
                    job.med[it * 3 + k] = median;
                    for (int i = 0; i < 9; i++)
                        job.Fm[((size_t)it * 3 + k) * 9 + i] = sF[k * 9 + i];
                }
            }
            if (lane == 0)
                job.nmodels[it] = nm;
            wave_lds_fence();  // sF is rewritten by the next iteration
        }
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned t = atomicAdd(job.ticket, due ? 1u : 0x10001u);
        *s_last = (t & 0xffffu) == gridDim.x - 1;
        *s_all_due = due && (t >> 16) == 0;
    }
    __syncthreads();
    if (!*s_last)
        return;
    if (threadIdx.x == 0)
        *job.ticket = 0;
    if (!*s_all_due || wave != 0)
        return;
    __shared__ double s_best[12];
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        double best = 1.7976931348623157e308;
        int bi = -1, bk = 0, it = 0;
        for (; it < niters; it++) {
            const int nm = job.nmodels[it];
            if (nm < 0)
                break;  // getSubset failed: the loop stops
            for (int k = 0; k < nm; k++)
                if (job.med[it * 3 + k] < best) {
                    best = job.med[it * 3 + k];
                    bi = it;
                    bk = k;
                }
        }
        s_best[9] = bi >= 0 ? 1. : 0.;
        s_best[10] = best;
        s_best[11] = (double)(n == M ? 0 : it);
        for (int i = 0; i < 9; i++)
            s_best[i] = bi >= 0 ? job.Fm[((size_t)bi * 3 + bk) * 9 + i] : 0.;
    }
    wave_lds_fence();
    const bool have = s_best[9] != 0.;
    double F[9];
#pragma unroll
    for (int i = 0; i < 9; i++)
        F[i] = s_best[i];
    bool keep = false;
    if (have && lane < n) {
        if (n == M)
            keep = true;  // seven pairs: the mask is all ones
        else {
            double sigma = 2.5 * 1.4826 * (1. + 5. / (n - M)) * sqrt(s_best[10]);
            sigma = sigma < 0.001 ? 0.001 : sigma;
            keep = f_error(F, a1.x, a1.y, a2.x, a2.y) <= (float)(sigma * sigma);
        }
    }
    unsigned long long bal = __ballot(keep);
    int count = __popcll(bal);
    if (count < M) {  // result = count >= modelPoints: no model
        keep = false;
        bal = 0;
        count = 0;
    }
    for (int i = lane; i < job.n_host; i += 64)  // the capacity of the mask
        job.mask[i] = (i == lane && keep) ? 1 : 0;
    if (job.c_in[0] && keep) {  // the compaction by the fresh mask, order kept
        const int pos = __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
        for (int r = 0; r < 3; r++)
            if (job.c_in[r]) {
                const int stn = job.c_stride[r];
                for (int k = 0; k < stn; k++)
                    job.c_out[r][(size_t)pos * stn + k] = job.c_in[r][(size_t)lane * stn + k];
            }
    }
    if (lane == 0) {
        if (job.c_in[0] && job.c_count)
            *job.c_count = count;
        if (job.out_count)
            *job.out_count = count;
        if (job.out_iters)
            *job.out_iters = (int)s_best[11];
        if (job.Fbest)
            for (int k = 0; k < 9; k++)
                job.Fbest[k] = count > 0 ? F[k] : 0.;
    }
}

// One phase of the RANSAC loop: iterations [it0, min(it1_cap, max_iters)), one wave per iteration (see the
// file header).  `final_phase`: no launch follows, so the last wave finishes the problem whatever the state.
// LEAN: the lock-step groups -- single-wave workgroups capped at 96 VGPRs (with spills), so that a wave starts beside
// four tracking waves of another context on its SIMD.  !LEAN: a lone problem has the chip to itself, its 64
// iterations would leave 15 of 16 SIMDs idle: FOUR waves per iteration (wave 0 solves, all four share the scoring
// pass, and the finishing workgroup's four waves share the mask / compaction pass), 128-VGPR build.
// TAIL: the instantiation the LAST launch of a call uses.  It also carries cv::findFundamentalMat's behaviour below 15
// pairs (fr_small_case), so that the kernel of the first 64 iterations -- the one on every frame's path -- stays as it was.
template <bool LEAN, bool TAIL>
__global__ __launch_bounds__(LEAN ? 64 : 256, LEAN ? 5 : 4) void fr_ransac_kernel(FrBatchN<LEAN ? SVO_LK_MAX_JOBS : 1> batch, int it0, int it1_cap,
                                                                                   int final_phase)
{
    constexpr int NW = LEAN ? 1 : 4;  // waves per workgroup
    svo_chain_priority();
    const FrJob &job = batch.j[blockIdx.y];
    // The gate (VoChain::kf / ::run) may be cleared on ANOTHER stream while this launch is being dispatched (the pipelined
    // chunk: the PnP stream halts the chain while the stereo stream's launch of a frame ahead starts), so the workgroups of
    // one launch need not agree on it.  A workgroup that finds it closed does no work but STILL takes its ticket: the
    // self-resetting counter then always reaches gridDim.x and returns to zero, whatever mixture of views the launch saw
    // (a skipped ticket left it at a partial count for every later launch on the context).  One read per
    // workgroup, shared through LDS: the scoring loop holds workgroup barriers.
    __shared__ int s_due;
    if (threadIdx.x == 0)
        s_due = !(job.gate && __hip_atomic_load(job.gate + (size_t)blockIdx.x * job.gate_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0);
    __syncthreads();
    const bool due = s_due != 0;
    const float2 *__restrict__ p1 = reinterpret_cast<const float2 *>(job.p1);
    const float2 *__restrict__ p2 = reinterpret_cast<const float2 *>(job.p2);
    const int n_host = job.n_host;
    const int *__restrict__ d_n = job.d_n;
    const int max_iters = job.max_iters;
    const int it1 = it1_cap < max_iters ? it1_cap : max_iters;
    RansacState *st = job.st;
    double *__restrict__ Fm = job.Fm;
    int *__restrict__ nmodels = job.nmodels;
    const float thr = job.thr;
    int *counts = job.counts;
    unsigned *ticket = job.ticket;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt0 = 4 * (int)threadIdx.x;  // this thread's first pair of a step; a step covers 256 * NW pairs
    constexpr int STEP = 256 * NW;
    __shared__ double sA[63], sV[18], sF[27];
    __shared__ int sPerm[9], s_nm, s_cnt[NW][3], s_last, s_all_due;
    const int n = d_n ? min(*d_n, n_host) : n_host;  // a live count never exceeds the capacity the buffers were sized for
    if (job.cv_small && n >= M && n < 15) {  // findFundamentalMat below 15 pairs is not a RANSAC (the same n in every workgroup)
        if (TAIL)
            fr_small_case<NW>(job, n, due, sA, sV, sPerm, sF, &s_last, &s_all_due);
        return;
    }
    if (it0 > 0 && st->done)  // the loop ended in an earlier phase (the same answer in every wave of the launch)
        return;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15) == 0;
    for (int it = it0 + (int)blockIdx.x; due && it < it1; it += gridDim.x) {
        if (wave == 0) {
            const int nm_l0 = fr_solve_wave(job.p1, job.p2, n, job.seed, it, sA, sV, sPerm, sF, lane);
            if (lane == 0) {
                const int nm = nm_l0;
                s_nm = nm;
                nmodels[it] = nm;
                for (int k = 0; k < 3; k++)
                    if (k < nm)
                        for (int i = 0; i < 9; i++)
                            Fm[((size_t)it * 3 + k) * 9 + i] = sF[k * 9 + i];  // the finishing wave reads the winner's
            }
        }
        __syncthreads();
        const int nm = s_nm;
        // Score the (up to three) models in ONE pass over the correspondences.  The loop is bound by the latency
        // of its loads (nothing else to switch to), so a lane takes FOUR consecutive pairs per step with 16-byte
        // loads and requests the next step's before it uses this step's; the model being evaluated sits in
        // scalar registers.
        int c0 = 0, c1 = 0, c2 = 0;
        if (nm > 0) {
            Quad cur, nxt;
            load_quad(cur, p1, p2, pt0, n, vec_ok);
            for (int base = pt0; base < n; base += STEP) {
                if (base + STEP < n)
                    load_quad(nxt, p1, p2, base + STEP, n, vec_ok);
                for (int k = 0; k < nm; k++) {
                    double F[9];
#pragma unroll
                    for (int i = 0; i < 9; i++)
                        F[i] = uniform_f64(sF[k * 9 + i]);
                    int c = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        c += (base + q < n && f_error(F, cur.x1[q], cur.y1[q], cur.x2[q], cur.y2[q]) <= thr) ? 1 : 0;
                    c0 += k == 0 ? c : 0;
                    c1 += k == 1 ? c : 0;
                    c2 += k == 2 ? c : 0;
                }
                cur = nxt;
            }
            c0 = wave_sum_small(c0);
            c1 = nm > 1 ? wave_sum_small(c1) : 0;
            c2 = nm > 2 ? wave_sum_small(c2) : 0;
        }
        if (NW > 1) {
            if (lane == 0) {
                s_cnt[wave][0] = c0;
                s_cnt[wave][1] = c1;
                s_cnt[wave][2] = c2;
            }
            __syncthreads();
            c0 = c1 = c2 = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                c0 += s_cnt[w][0];
                c1 += s_cnt[w][1];
                c2 += s_cnt[w][2];
            }
        }
        if (threadIdx.x == 0) {
            counts[it * 3 + 0] = c0;
            counts[it * 3 + 1] = c1;
            counts[it * 3 + 2] = c2;
        }
        __syncthreads();  // sF / s_nm / s_cnt are rewritten by the next iteration of this workgroup
    }
    // every workgroup takes a ticket once its counts are out; the holder of the last one sees them all
    // (low half: workgroups that have arrived; high half: how many of them found the gate closed)
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned t = atomicAdd(ticket, due ? 1u : 0x10001u);
        s_last = (t & 0xffffu) == gridDim.x - 1;
        s_all_due = due && (t >> 16) == 0;
    }
    __syncthreads();
    if (!s_last)
        return;
    __shared__ RansacState s_state;
    if (threadIdx.x == 0)
        *ticket = 0;  // ready for the next launch
    if (!s_all_due) {
        // some workgroup found the gate closed: its iteration slots hold whatever the workspace held before (it is
        // shared with other stages), so there is nothing to replay.  The state says "ended, no model": a phase that
        // follows leaves at once.  What the launch leaves behind belongs to a frame the halted chain discards and
        // the host queues again.
        if (threadIdx.x == 0) {
            RansacState r;
            r.niters = 0, r.next_iter = 0, r.best_iter = -1, r.best_model = 0, r.best_count = 0, r.done = 1, r.iters_run = 0, r.pad = 0;
            *st = r;
        }
        return;
    }
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // other waves' counts, not this CU's stale lines
        const RansacState r = ransac_replay<3>(st, it0 == 0 ? 1 : 0, it1, max_iters, n, job.confidence, nmodels, counts, M);
        *st = r;
        s_state = r;
    }
    __syncthreads();
    const RansacState s = s_state;
    if (!s.done && !final_phase)
        return;  // the next phase's last workgroup finishes
    // ---- the winning model: mask (cv: computeError <= thr on every correspondence), model, counts ----
    uint8_t *__restrict__ mask = job.mask;
    const bool have = s.best_iter >= 0 && s.best_count > 0;
    double F[9];
#pragma unroll
    for (int k = 0; k < 9; k++)
        F[k] = have ? Fm[((size_t)s.best_iter * 3 + s.best_model) * 9 + k] : 0.;
    if (threadIdx.x == 0) {
        if (job.out_count)
            *job.out_count = have ? s.best_count : 0;
        if (job.out_iters)
            *job.out_iters = s.iters_run;
        if (job.Fbest)
            for (int k = 0; k < 9; k++)
                job.Fbest[k] = F[k];
    }
    const bool compacting = job.c_in[0] != nullptr;
    __shared__ int s_kept[NW];
    int pos0 = 0;
    Quad cur, nxt;
    load_quad(cur, p1, p2, pt0, n, vec_ok);
    for (int start = 0; start < n_host; start += STEP) {  // n_host: the capacity of the mask
        const int base = start + pt0;
        if (base + STEP < n)  // the next four pairs are on their way while these are judged
            load_quad(nxt, p1, p2, base + STEP, n, vec_ok);
        bool keep[4];
        unsigned long long bal[4];
        int before = 0;  // kept entries of lower lanes in this step (a lane's four entries are consecutive)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            keep[q] = have && base + q < n && f_error(F, cur.x1[q], cur.y1[q], cur.x2[q], cur.y2[q]) <= thr;
            if (base + q < n_host)
                mask[base + q] = keep[q] ? 1 : 0;
            bal[q] = __ballot(keep[q]);
            before += __popcll(bal[q] & ((1ull << lane) - 1ull));
        }
        if (compacting) {  // what compact_kernel does with this mask, in the same pass (order kept)
            const int mine = __popcll(bal[0]) + __popcll(bal[1]) + __popcll(bal[2]) + __popcll(bal[3]);
            int lower = 0, all = mine;  // kept by the waves below this one / by the whole workgroup in this step
            if (NW > 1) {
                if (lane == 0)
                    s_kept[wave] = mine;
                __syncthreads();
                all = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    lower += w < wave ? s_kept[w] : 0;
                    all += s_kept[w];
                }
                __syncthreads();  // s_kept is rewritten by the next step
            }
            int pos = pos0 + lower + before;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (keep[q]) {
                    const int i = base + q;
#pragma unroll
                    for (int r = 0; r < 3; r++)
                        if (job.c_in[r]) {
                            const int stn = job.c_stride[r];
                            for (int k = 0; k < stn; k++)
                                job.c_out[r][(size_t)pos * stn + k] = job.c_in[r][(size_t)i * stn + k];
                        }
                    pos++;
                }
            pos0 += all;
        }
        cur = nxt;
    }
    if (compacting && job.c_count && threadIdx.x == 0)
        *job.c_count = pos0;
}

}  // namespace

// device-pointer form used by the C ABI and by the fused front-end, several problems at once.
// `cap` sizes the grids (the number of correspondences is *d_n when d_n != nullptr, else cap).
int svo_launch_fransac_batch(svo_ctx *ctx, int n_jobs, const svo_fransac_job *jobs)
{
    if (n_jobs <= 0)
        return SVO_OK;
    if (n_jobs > SVO_LK_MAX_JOBS) {
        svo_set_error("fransac: at most %d jobs per launch", SVO_LK_MAX_JOBS);
        return SVO_ERR_ARG;
    }
    int it_max = 1, cap_max = 0;
    for (int k = 0; k < n_jobs; k++) {
        const int mi = jobs[k].max_iters < 1 ? 1 : jobs[k].max_iters;
        it_max = mi > it_max ? mi : it_max;
        cap_max = jobs[k].cap > cap_max ? jobs[k].cap : cap_max;
    }
    const int it_ws = it_max < 304 ? 304 : it_max;  // the least-median branch runs 300 iterations whatever max_iters says
    const size_t f_stride = (size_t)it_ws * 30, i_stride = ((size_t)it_ws * 4 + 32 + 15) / 16 * 16;
    int rc;
    if ((rc = ctx->w_a.ensure(f_stride * sizeof(double) * n_jobs)) || (rc = ctx->w_b.ensure(i_stride * sizeof(int) * n_jobs)))
        return rc;
    FrBatch batch;
    int nb = 0;
    for (int k = 0; k < n_jobs; k++) {
        const svo_fransac_job &h = jobs[k];
        if (h.cap <= 0)
            continue;
        FrJob &j = batch.j[nb];
        const int mi = h.max_iters < 1 ? 1 : h.max_iters;
        int *ib = ctx->w_b.as<int>() + i_stride * nb;
        j.p1 = h.p1;
        j.p2 = h.p2;
        j.n_host = h.cap;
        j.d_n = h.d_n;
        j.seed = h.seed;
        j.max_iters = mi;
        j.confidence = h.confidence;
        j.thr = (float)(h.threshold * h.threshold);
        j.st = reinterpret_cast<RansacState *>(ib);
        j.nmodels = ib + 16;
        j.counts = j.nmodels + it_ws;
        j.Fm = ctx->w_a.as<double>() + f_stride * nb;
        j.med = j.Fm + (size_t)it_ws * 27;
        j.cv_small = h.cv_small ? 1 : 0;
        j.ticket = ctx->d_tickets + nb;  // slots 0..15 (16..31: pnp.hip)
        j.mask = h.mask;
        j.Fbest = h.d_F;
        j.out_count = h.d_count;
        j.out_iters = h.d_iters;
        for (int a = 0; a < 3; a++) {
            const svo_compact_job *c = h.then_compact;
            j.c_in[a] = c ? c->in[a] : nullptr;
            j.c_out[a] = c ? c->out[a] : nullptr;
            j.c_stride[a] = c ? c->stride[a] : 0;
        }
        j.c_count = h.then_compact ? h.then_compact->d_count : nullptr;
        j.gate = h.gate;
        j.gate_stride = h.gate_stride;
        nb++;
    }
    if (nb == 0)
        return SVO_OK;
    for (int k = nb; k < SVO_LK_MAX_JOBS; k++)
        batch.j[k] = batch.j[0];
    ScopedKernelTime tm(ctx, SVO_K_FRANSAC);
    const int bounds[3] = {0, it_max < PHASE_A ? it_max : PHASE_A, it_max};
    bool any_small = false;
    for (int k = 0; k < nb; k++)
        any_small |= batch.j[k].cv_small != 0;
    FrBatchN<1> one;
    one.j[0] = batch.j[0];
    auto launch = [&](int it0, int it1, int waves, bool tail) {
        const int fin = it1 >= it_max ? 1 : 0;
        if (nb > 1) {
            if (tail)
                hipLaunchKernelGGL((fr_ransac_kernel<true, true>), dim3(waves, nb), dim3(64), 0, ctx->stream, batch, it0, it1, fin);
            else
                hipLaunchKernelGGL((fr_ransac_kernel<true, false>), dim3(waves, nb), dim3(64), 0, ctx->stream, batch, it0, it1, fin);
        } else {
            if (tail)
                hipLaunchKernelGGL((fr_ransac_kernel<false, true>), dim3(waves, nb), dim3(256), 0, ctx->stream, one, it0, it1, fin);
            else
                hipLaunchKernelGGL((fr_ransac_kernel<false, false>), dim3(waves, nb), dim3(256), 0, ctx->stream, one, it0, it1, fin);
        }
    };
    // two phases (iterations 0..63, 64..max); the LAST launch of the call is the TAIL instantiation, which also carries the
    // small-sample branch.  A call of at most 64 iterations whose jobs want that branch gets an (empty) tail launch for it.
    const bool two = bounds[2] > bounds[1];
    launch(bounds[0], bounds[1], bounds[1] - bounds[0] < PHASE_WAVES ? bounds[1] - bounds[0] : PHASE_WAVES, !two && !any_small);
    if (two)
        launch(bounds[1], bounds[2], bounds[2] - bounds[1] < PHASE_WAVES ? bounds[2] - bounds[1] : PHASE_WAVES, true);
    else if (any_small)
        launch(bounds[2], bounds[2], PHASE_WAVES, true);
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

int svo_launch_fransac(svo_ctx *ctx, const float *p1, const float *p2, int cap, const int *d_n,
                       double threshold, double confidence, int max_iters, uint64_t seed, uint8_t *mask,
                       double *d_F, int *d_count, int *d_iters, const svo_compact_job *then_compact, bool cv_small)
{
    svo_fransac_job j;
    j.cv_small = cv_small;
    j.then_compact = then_compact;
    j.p1 = p1;
    j.p2 = p2;
    j.cap = cap;
    j.d_n = d_n;
    j.threshold = threshold;
    j.confidence = confidence;
    j.max_iters = max_iters;
    j.seed = seed;
    j.mask = mask;
    j.d_F = d_F;
    j.d_count = d_count;
    j.d_iters = d_iters;
    return svo_launch_fransac_batch(ctx, 1, &j);
}

// Diagnostics (fransac.hip gate / ticket): ONE launch of each phase in which the workgroups disagree about
// the gate -- even workgroups find it open, odd ones closed -- as the pipelined chunk can produce when the PnP stream
// halts the chain under a launch the stereo stream is dispatching.  Reports the value the self-resetting ticket counter
// is left at (must be 0) so that the caller can check that the NEXT ordinary call on the context is unaffected.
// Device pointers; `lean`: the single-wave build the lock-step groups use (two identical jobs) instead of the lone one.
extern "C" int svo_selftest_fransac_gate(svo_ctx *ctx, const float *d_p1, const float *d_p2, int n, double threshold,
                                         uint64_t seed, int lean, uint8_t *d_mask, unsigned *tickets_after)
{
    SVO_CHECK_ARG(ctx && d_p1 && d_p2 && n > 0 && d_mask && tickets_after);
    int rc;
    if ((rc = ctx->s_g.ensure(sizeof(int) * PHASE_WAVES)))
        return rc;
    int h_gate[PHASE_WAVES];
    for (int b = 0; b < PHASE_WAVES; b++)
        h_gate[b] = (b & 1) ? 0 : 1;
    SVO_HIP(hipMemcpyAsync(ctx->s_g.p, h_gate, sizeof(h_gate), hipMemcpyHostToDevice, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    svo_fransac_job j[2];
    for (int a = 0; a < 2; a++) {
        j[a].p1 = d_p1;
        j[a].p2 = d_p2;
        j[a].cap = n;
        j[a].d_n = nullptr;
        j[a].threshold = threshold;
        j[a].confidence = 0.99;
        j[a].max_iters = 1000;
        j[a].seed = seed;
        j[a].mask = d_mask;
        j[a].d_F = nullptr;
        j[a].d_count = nullptr;
        j[a].d_iters = nullptr;
        j[a].gate = ctx->s_g.as<int>();
        j[a].gate_stride = 1;
    }
    if ((rc = svo_launch_fransac_batch(ctx, lean ? 2 : 1, j)))
        return rc;
    SVO_HIP(hipMemcpyAsync(tickets_after, ctx->d_tickets, sizeof(unsigned) * 2, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    return SVO_OK;
}

extern "C" int svo_fransac(svo_ctx *ctx, const float *p1, const float *p2, int n, double threshold,
                           double confidence, int max_iters, uint64_t seed, uint8_t *mask, double *F9,
                           int *inlier_count, int *iters_run, int mem)
{
    return svo_fransac_ex(ctx, p1, p2, n, threshold, confidence, max_iters, seed, mask, F9, inlier_count, iters_run, mem, true);
}

// cv_small = false: a RANSAC at any count >= 7 (the loop detector's geometric check)
int svo_fransac_ex(svo_ctx *ctx, const float *p1, const float *p2, int n, double threshold, double confidence, int max_iters,
                   uint64_t seed, uint8_t *mask, double *F9, int *inlier_count, int *iters_run, int mem, bool cv_small)
{
    SVO_CHECK_ARG(ctx && n >= 0 && threshold > 0 && max_iters > 0);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    if (n == 0) {
        if (mem == SVO_MEM_HOST) {
            if (inlier_count)
                *inlier_count = 0;
            if (iters_run)
                *iters_run = 0;
        }
        return SVO_OK;
    }
    SVO_CHECK_ARG(p1 && p2 && mask);
    if (mem == SVO_MEM_DEVICE)
        return svo_launch_fransac(ctx, p1, p2, n, nullptr, threshold, confidence, max_iters, seed, mask, F9,
                                  inlier_count, iters_run, nullptr, cv_small);
    int rc;
    if ((rc = ctx->s_a.ensure((size_t)n * 8)) || (rc = ctx->s_b.ensure((size_t)n * 8)) ||
        (rc = ctx->s_c.ensure((size_t)n)) || (rc = ctx->s_d.ensure(256)))
        return rc;
    SVO_HIP(hipMemcpyAsync(ctx->s_a.p, p1, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    SVO_HIP(hipMemcpyAsync(ctx->s_b.p, p2, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    double *dF = ctx->s_d.as<double>();
    int *dcnt = reinterpret_cast<int *>(dF + 9), *dit = dcnt + 1;
    rc = svo_launch_fransac(ctx, ctx->s_a.as<float>(), ctx->s_b.as<float>(), n, nullptr, threshold, confidence,
                            max_iters, seed, ctx->s_c.as<uint8_t>(), dF, dcnt, dit, nullptr, cv_small);
    if (rc)
        return rc;
    SVO_HIP(hipMemcpyAsync(mask, ctx->s_c.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipMemcpyAsync(ctx->pinned, dF, 9 * sizeof(double) + 2 * sizeof(int), hipMemcpyDeviceToHost,
                           ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    const double *hF = reinterpret_cast<const double *>(ctx->pinned);
    const int *hI = reinterpret_cast<const int *>(hF + 9);
    if (F9)
        memcpy(F9, hF, 9 * sizeof(double));
    if (inlier_count)
        *inlier_count = hI[0];
    if (iters_run)
        *iters_run = hI[1];
    return SVO_OK;
}
