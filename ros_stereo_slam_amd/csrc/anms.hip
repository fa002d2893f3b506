// anms.hip -- adaptive non-maximal suppression for gfx950.
//
// Replaces adaptiveNonMaximalSuppresion(keypoints, numToKeep), src/ANMS.cpp:18-67: sort by
// response (descending), suppression radius = distance to the nearest keypoint whose
// response exceeds 1.11f x own, keep every keypoint whose radius >= the (numToKeep+1)-th
// largest radius, in response order.  The reference's grid keypoints carry response 0
// (src/tracking.cpp:8), so the caller supplies one (the level-0 LK minimum eigenvalue).
//
// N is a few thousand: every step is an all-pairs pass with the compared array staged in
// LDS (a rank-by-counting sort, the radius scan, a rank-by-counting selection), one thread
// per keypoint.  Ties in response keep input order (std::sort is unstable upstream);
// N <= numToKeep returns everything (the reference reads out of bounds at N == numToKeep).
#include <cfloat>

#include "svo_internal.h"

namespace {

constexpr int CHUNK = 2048;  // keypoints staged in LDS per pass

// rank[i] = number of keypoints that sort before i (response descending, index ascending)
__global__ __launch_bounds__(256) void anms_rank_kernel(const float *__restrict__ resp, int n, int *__restrict__ order)
{
    __shared__ float s_r[CHUNK];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float ri = i < n ? resp[i] : 0.f;
    int rank = 0;
    for (int base = 0; base < n; base += CHUNK) {
        const int cnt = min(CHUNK, n - base);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += 256)
            s_r[k] = resp[base + k];
        __syncthreads();
        if (i < n)
            for (int k = 0; k < cnt; k++) {
                const float rj = s_r[k];
                const int j = base + k;
                rank += (rj > ri || (rj == ri && j < i)) ? 1 : 0;
            }
    }
    if (i < n)
        order[rank] = i;  // order[s] = input index of the s-th keypoint in sorted order
}

// squared suppression radius of the s-th sorted keypoint (DBL_MAX when nothing dominates it)
__global__ __launch_bounds__(256) void anms_radius_kernel(const float2 *__restrict__ xy, const float *__restrict__ resp,
                                                          const int *__restrict__ order, int n,
                                                          double *__restrict__ radius_sq)
{
    __shared__ float s_x[CHUNK], s_y[CHUNK], s_r[CHUNK];
    const int s = blockIdx.x * 256 + threadIdx.x;
    float xi = 0, yi = 0, thr = 0;
    if (s < n) {
        const int i = order[s];
        xi = xy[i].x;
        yi = xy[i].y;
        thr = resp[i] * 1.11f;
    }
    double best = DBL_MAX;
    // only sorted positions j < s can dominate; blocks never need chunks past their last s
    const int limit = min(n, (int)(blockIdx.x * 256 + 256));
    for (int base = 0; base < limit; base += CHUNK) {
        const int cnt = min(CHUNK, limit - base);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += 256) {
            const int j = order[base + k];
            s_x[k] = xy[j].x;
            s_y[k] = xy[j].y;
            s_r[k] = resp[j];
        }
        __syncthreads();
        if (s < n)
            for (int k = 0; k < cnt && base + k < s; k++) {
                if (!(s_r[k] > thr))
                    break;  // sorted by response: nothing later dominates either
                const float dx = xi - s_x[k], dy = yi - s_y[k];
                const double d = (double)dx * dx + (double)dy * dy;
                best = d < best ? d : best;
            }
    }
    if (s < n)
        radius_sq[s] = best;
}

// decision radius = the (keep+1)-th largest radius; flags[s] = radius[s] >= decision
__global__ __launch_bounds__(256) void anms_decide_kernel(const double *__restrict__ radius_sq, int n, int keep,
                                                          uint8_t *__restrict__ flags, double *__restrict__ decision)
{
    __shared__ double s_v[CHUNK];
    const int s = blockIdx.x * 256 + threadIdx.x;
    const double ri = s < n ? radius_sq[s] : 0.;
    int gt = 0, ge = 0;
    for (int base = 0; base < n; base += CHUNK) {
        const int cnt = min(CHUNK, n - base);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += 256)
            s_v[k] = radius_sq[base + k];
        __syncthreads();
        if (s < n)
            for (int k = 0; k < cnt; k++) {
                gt += s_v[k] > ri ? 1 : 0;
                ge += s_v[k] >= ri ? 1 : 0;
            }
    }
    // radiiSorted[keep] (descending, 0-based) == ri  <=>  gt <= keep < ge
    if (s < n && gt <= keep && keep < ge)
        *decision = ri;  // every thread that qualifies writes the same value
    (void)flags;
}

__global__ __launch_bounds__(256) void anms_flag_kernel(const double *__restrict__ radius_sq, int n,
                                                        const double *__restrict__ decision,
                                                        uint8_t *__restrict__ flags)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s < n)
        flags[s] = radius_sq[s] >= *decision ? 1 : 0;
}

// kept[] = order[s] for flagged s, in sorted order (single workgroup scan)
__global__ __launch_bounds__(1024) void anms_gather_kernel(const uint8_t *__restrict__ flags,
                                                           const int *__restrict__ order, int n,
                                                           int *__restrict__ out_idx, int *__restrict__ d_count)
{
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

// Device form.  out_idx: n ints (input indices of the kept keypoints, response order);
// d_count: device int.  Uses ctx->w_a..w_c as scratch.
int svo_launch_anms(svo_ctx *ctx, const float *xy, const float *resp, int n, int keep, int *out_idx, int *d_count)
{
    if (n <= 0)
        return SVO_OK;
    ScopedKernelTime tm(ctx, SVO_K_ANMS);
    int rc;
    if ((rc = ctx->w_a.ensure((size_t)n * 4)) || (rc = ctx->w_b.ensure((size_t)n * 8 + 64)) ||
        (rc = ctx->w_c.ensure((size_t)n)))
        return rc;
    int *order = ctx->w_a.as<int>();
    double *radius = ctx->w_b.as<double>();
    double *decision = radius + n;
    uint8_t *flags = ctx->w_c.as<uint8_t>();
    const dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(anms_rank_kernel, grid, block, 0, ctx->stream, resp, n, order);
    if (n <= keep) {
        // everything is kept, in sorted order
        SVO_HIP(hipMemcpyAsync(out_idx, order, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, ctx->stream, d_count, n);
        SVO_HIP(hipGetLastError());
        return SVO_OK;
    }
    hipLaunchKernelGGL(anms_radius_kernel, grid, block, 0, ctx->stream, reinterpret_cast<const float2 *>(xy), resp,
                       order, n, radius);
    hipLaunchKernelGGL(anms_decide_kernel, grid, block, 0, ctx->stream, radius, n, keep, flags, decision);
    hipLaunchKernelGGL(anms_flag_kernel, grid, block, 0, ctx->stream, radius, n, decision, flags);
    hipLaunchKernelGGL(anms_gather_kernel, dim3(1), dim3(1024), 0, ctx->stream, flags, order, n, out_idx, d_count);
    SVO_HIP(hipGetLastError());
    return SVO_OK;
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
