// anms.hip -- adaptive non-maximal suppression for gfx950.
//
// Replaces adaptiveNonMaximalSuppresion(keypoints, numToKeep), src/ANMS.cpp:18-67: sort by
// response (descending), suppression radius = distance to the nearest keypoint whose
// response exceeds 1.11f x own, keep every keypoint whose radius >= the (numToKeep+1)-th
// largest radius, in response order.  The reference's grid keypoints carry response 0
// (src/tracking.cpp:8), so the caller supplies one (the level-0 LK minimum eigenvalue).
//
// N is a few thousand: every step is an all-pairs pass (a rank-by-counting sort, the radius
// scan, a rank-by-counting selection), one WAVEFRONT per keypoint.  Ties in response keep input order (std::sort is unstable upstream);
// N <= numToKeep returns everything (the reference reads out of bounds at N == numToKeep).
#include <cfloat>

#include "svo_internal.h"

namespace {

// exact wave-wide integer sum (per-lane values < 2^24)
__device__ __forceinline__ int wave_sum_int(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// One WAVEFRONT per keypoint in every all-pairs pass: the 64 lanes stride over the other
// keypoints (coalesced reads of a 17-50 KB array that stays in L2) and the per-lane partials
// are combined with DPP / shuffles.  N waves instead of N/64 keep all 256 CUs busy.

// One ANMS problem; a launch may carry several of the same size n (blockIdx.y picks the job).
struct AnmsJob {
    const float2 *xy;
    const float *resp;
    int *order;
    float4 *sorted;
    double *radius, *decision;
    uint8_t *flags;
    int *out_idx, *d_count;
};
struct AnmsBatch {
    AnmsJob j[SVO_LK_MAX_JOBS];
};

// order[rank] = i, rank = #keypoints sorting before i (response descending, index ascending);
// also emits the sorted (x, y, response) triples the radius pass streams through.
__global__ __launch_bounds__(256) void anms_rank_kernel(AnmsBatch batch, int n)
{
    __builtin_amdgcn_s_setprio(3);  // short latency-bound kernel: win issue arbitration against co-resident LK waves
    const AnmsJob &job = batch.j[blockIdx.y];
    const float2 *__restrict__ xy = job.xy;
    const float *__restrict__ resp = job.resp;
    int *__restrict__ order = job.order;
    float4 *__restrict__ sorted = job.sorted;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x;  // one wave per workgroup (see svo_launch_anms_batch)
    if (i >= n)
        return;
    const float ri = resp[i];
    int cnt = 0;
    // "j sorts before i"  <=>  r_j > r_i, or r_j == r_i and j < i.  Whole 64-entry stripes before i's
    // own stripe need one compare (>=), the ones after it one compare (>); only i's stripe needs the
    // tie-break.  Scalar loop bounds: no per-lane range test except in the last, partial stripe.
    const float *__restrict__ rl = resp + lane;
    const int own = (i >> 6) << 6;
#pragma unroll 8
    for (int j0 = 0; j0 < own; j0 += 64)
        cnt += rl[j0] >= ri ? 1 : 0;
    {
        const int j = own + lane;
        if (j < n) {
            const float rj = rl[own];
            cnt += (rj > ri || (rj == ri && j < i)) ? 1 : 0;
        }
    }
    int j0 = own + 64;
#pragma unroll 8
    for (; j0 + 64 <= n; j0 += 64)
        cnt += rl[j0] > ri ? 1 : 0;
    if (j0 < n && j0 + lane < n)
        cnt += rl[j0] > ri ? 1 : 0;
    const int rank = wave_sum_int(cnt);
    if (lane == 0) {
        order[rank] = i;
        const float2 p = xy[i];
        sorted[rank] = make_float4(p.x, p.y, ri, 0.f);
    }
}

// squared suppression radius of the s-th sorted keypoint (DBL_MAX when nothing dominates it)
__global__ __launch_bounds__(256) void anms_radius_kernel(AnmsBatch batch, int n)
{
    __builtin_amdgcn_s_setprio(3);  // short latency-bound kernel: win issue arbitration against co-resident LK waves
    const float4 *__restrict__ sorted = batch.j[blockIdx.y].sorted;
    double *__restrict__ radius_sq = batch.j[blockIdx.y].radius;
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x;
    if (s >= n)
        return;
    const float4 me = sorted[s];
    const float thr = me.z * 1.11f;
    double best = DBL_MAX;
    for (int j0 = 0; j0 < s; j0 += 64) {
        const int j = j0 + lane;
        bool dom = false;
        if (j < s) {
            const float4 o = sorted[j];
            dom = o.z > thr;
            if (dom) {
                const float dx = me.x - o.x, dy = me.y - o.y;
                const double d = (double)dx * dx + (double)dy * dy;
                best = d < best ? d : best;
            }
        }
        // sorted by response: once a whole stripe fails, nothing later dominates either
        if (!__any(dom))
            break;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(best, off);
        best = o < best ? o : best;
    }
    if (lane == 0)
        radius_sq[s] = best;
}

// decision radius = the (keep+1)-th largest radius
__global__ __launch_bounds__(256) void anms_decide_kernel(AnmsBatch batch, int n, int keep)
{
    __builtin_amdgcn_s_setprio(3);  // short latency-bound kernel: win issue arbitration against co-resident LK waves
    const double *__restrict__ radius_sq = batch.j[blockIdx.y].radius;
    double *__restrict__ decision = batch.j[blockIdx.y].decision;
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x;
    if (s >= n)
        return;
    const double ri = radius_sq[s];
    int gt = 0, ge = 0;
    const double *__restrict__ rl = radius_sq + lane;
    int j0 = 0;
#pragma unroll 8
    for (; j0 + 64 <= n; j0 += 64) {  // scalar bounds: no per-lane range test in the body
        const double rj = rl[j0];
        gt += rj > ri ? 1 : 0;
        ge += rj >= ri ? 1 : 0;
    }
    if (j0 + lane < n) {
        const double rj = rl[j0];
        gt += rj > ri ? 1 : 0;
        ge += rj >= ri ? 1 : 0;
    }
    gt = wave_sum_int(gt);
    ge = wave_sum_int(ge);
    // radiiSorted[keep] (descending, 0-based) == ri  <=>  gt <= keep < ge
    if (lane == 0 && gt <= keep && keep < ge)
        *decision = ri;  // every wave that qualifies writes the same value
}

__global__ __launch_bounds__(256) void anms_flag_kernel(AnmsBatch batch, int n)
{
    __builtin_amdgcn_s_setprio(3);  // short latency-bound kernel: win issue arbitration against co-resident LK waves
    const double *__restrict__ radius_sq = batch.j[blockIdx.y].radius;
    const double *__restrict__ decision = batch.j[blockIdx.y].decision;
    uint8_t *__restrict__ flags = batch.j[blockIdx.y].flags;
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s < n)
        flags[s] = radius_sq[s] >= *decision ? 1 : 0;
}

// kept[] = order[s] for flagged s, in sorted order (single workgroup scan)
__global__ __launch_bounds__(1024) void anms_gather_kernel(AnmsBatch batch, int n)
{
    __builtin_amdgcn_s_setprio(3);  // short latency-bound kernel: win issue arbitration against co-resident LK waves
    const AnmsJob &job = batch.j[blockIdx.x];  // one workgroup per job
    const uint8_t *__restrict__ flags = job.flags;
    const int *__restrict__ order = job.order;
    int *__restrict__ out_idx = job.out_idx;
    int *__restrict__ d_count = job.d_count;
    __shared__ int s_sum[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b = t * per, e = min(b + per, n);
    int cnt = 0;
    for (int i = b; i < e; i++)
        cnt += flags[i];
    s_sum[t] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = t >= off ? s_sum[t - off] : 0;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    int pos = s_sum[t] - cnt;
    if (t == 1023)
        *d_count = s_sum[1023];
    for (int i = b; i < e; i++)
        if (flags[i])
            out_idx[pos++] = order[i];
}

__global__ void set_int_kernel(int *p, int v) { *p = v; }

}  // namespace

// Device form, several problems of the same size n.  out_idx: n ints (input indices of the kept
// keypoints, response order); d_count: device int.  Uses ctx->w_a..w_d as scratch.
int svo_launch_anms_batch(svo_ctx *ctx, int k, const float *const *xy, const float *const *resp, int n, int keep,
                          int *const *out_idx, int *const *d_count)
{
    if (n <= 0 || k <= 0)
        return SVO_OK;
    if (k > SVO_LK_MAX_JOBS) {
        svo_set_error("anms: at most %d jobs per launch", SVO_LK_MAX_JOBS);
        return SVO_ERR_ARG;
    }
    ScopedKernelTime tm(ctx, SVO_K_ANMS);
    int rc;
    const size_t na = ((size_t)n + 63) / 64 * 64;
    if ((rc = ctx->w_a.ensure(na * 4 * k)) || (rc = ctx->w_b.ensure((na * 8 + 64) * k)) ||
        (rc = ctx->w_c.ensure(na * k)) || (rc = ctx->w_d.ensure(na * 16 * k)))
        return rc;
    AnmsBatch batch;
    for (int a = 0; a < SVO_LK_MAX_JOBS; a++) {
        const int q = a < k ? a : 0;
        AnmsJob &j = batch.j[a];
        j.xy = reinterpret_cast<const float2 *>(xy[q]);
        j.resp = resp[q];
        j.order = ctx->w_a.as<int>() + na * q;
        j.radius = reinterpret_cast<double *>(ctx->w_b.as<uint8_t>() + (na * 8 + 64) * q);
        j.decision = j.radius + na;
        j.flags = ctx->w_c.as<uint8_t>() + na * q;
        j.sorted = ctx->w_d.as<float4>() + na * q;
        j.out_idx = out_idx[q];
        j.d_count = d_count[q];
    }
    // single-wave workgroups: beside a tracking launch (single-wave workgroups that take every freed wave
    // slot at once) a multi-wave workgroup waits until one CU has a slot free on several SIMDs together
    const dim3 wgrid(n, k), tgrid((n + 63) / 64, k), block(64);
    hipLaunchKernelGGL(anms_rank_kernel, wgrid, block, 0, ctx->stream, batch, n);
    if (n <= keep) {
        // everything is kept, in sorted order
        for (int a = 0; a < k; a++) {
            SVO_HIP(hipMemcpyAsync(out_idx[a], batch.j[a].order, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
            hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, ctx->stream, d_count[a], n);
        }
        SVO_HIP(hipGetLastError());
        return SVO_OK;
    }
    hipLaunchKernelGGL(anms_radius_kernel, wgrid, block, 0, ctx->stream, batch, n);
    hipLaunchKernelGGL(anms_decide_kernel, wgrid, block, 0, ctx->stream, batch, n, keep);
    hipLaunchKernelGGL(anms_flag_kernel, tgrid, block, 0, ctx->stream, batch, n);
    hipLaunchKernelGGL(anms_gather_kernel, dim3(k), dim3(1024), 0, ctx->stream, batch, n);
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

int svo_launch_anms(svo_ctx *ctx, const float *xy, const float *resp, int n, int keep, int *out_idx, int *d_count)
{
    return svo_launch_anms_batch(ctx, 1, &xy, &resp, n, keep, &out_idx, &d_count);
}

extern "C" int svo_anms(svo_ctx *ctx, const float *xy, const float *response, int n, int num_to_keep, int *out_idx,
                        int *count, int mem)
{
    SVO_CHECK_ARG(ctx && n >= 0 && num_to_keep >= 0);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    if (n == 0) {
        if (count && mem == SVO_MEM_HOST)
            *count = 0;
        return SVO_OK;
    }
    SVO_CHECK_ARG(xy && response && out_idx && count);
    if (mem == SVO_MEM_DEVICE)
        return svo_launch_anms(ctx, xy, response, n, num_to_keep, out_idx, count);
    int rc;
    if ((rc = ctx->s_a.ensure((size_t)n * 8)) || (rc = ctx->s_b.ensure((size_t)n * 4)) ||
        (rc = ctx->s_c.ensure((size_t)n * 4 + 64)))
        return rc;
    SVO_HIP(hipMemcpyAsync(ctx->s_a.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    SVO_HIP(hipMemcpyAsync(ctx->s_b.p, response, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    int *didx = ctx->s_c.as<int>(), *dcnt = didx + n;
    rc = svo_launch_anms(ctx, ctx->s_a.as<float>(), ctx->s_b.as<float>(), n, num_to_keep, didx, dcnt);
    if (rc)
        return rc;
    SVO_HIP(hipMemcpyAsync(ctx->pinned, dcnt, 4, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    const int k = *reinterpret_cast<int *>(ctx->pinned);
    SVO_HIP(hipMemcpyAsync(out_idx, didx, (size_t)k * 4, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    *count = k;
    return SVO_OK;
}
