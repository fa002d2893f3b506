// sor.hip -- statistical outlier removal of the map points for gfx950.
//
// Replaces visualSLAM::SORcloud (src/rosFuncs.cpp:9-39): the far-point pre-filter (-z > 500,
// :12) and pcl::StatisticalOutlierRemoval with MeanK 200 / StddevMulThresh 0.01 (:19-23), whose
// published algorithm is: mean distance to the mean_k nearest OTHER points for every point,
// mean and standard deviation of those over the cloud, keep d_i <= mean + mul * stddev.
//
// Mapping: ONE WAVEFRONT PER POINT.  A lane computes the squared distances to the points
// lane, lane+64, ... once and keeps them in VGPRs (float bit patterns order like unsigned ints);
// the k-th smallest is found by a 31-step bitwise search with wave-wide counts -- no sort, no
// neighbour lists in memory -- and the mean is the sum of the square roots below it plus the
// ties at it.  The sums are doubles of float terms (exact for any realistic spread, so the lane
// order does not matter).  The cloud statistics are summed by one thread in point order, as
// PCL's loop does, so the threshold is bit-identical to the sequential algorithm.
// Brute force: N^2 distance evaluations out of L2 (N <= 9216: 110 kB), no HBM traffic to speak of.
#include "ransac_common.hip.h"
#include "svo_internal.h"

namespace {

__global__ __launch_bounds__(256) void sor_zmask_kernel(const float *__restrict__ xyz, int n, float z_limit,
                                                        uint8_t *__restrict__ mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        mask[i] = (z_limit > 0.f && -1.f * xyz[3 * i + 2] > z_limit) ? 0 : 1;
}

__device__ __forceinline__ double wave_sum_double(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

template <int T>
__global__ __launch_bounds__(256) void sor_knn_kernel(const float *__restrict__ xyz, const int *__restrict__ d_m,
                                                      int mean_k, float *__restrict__ dist)
{
    const int m = *d_m;
    const int lane = threadIdx.x & 63;
    const int a = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (a >= m)
        return;
    const int kk = mean_k < m - 1 ? mean_k : m - 1;
    if (kk <= 0) {
        if (lane == 0)
            dist[a] = 0.f;
        return;
    }
    const float px = xyz[3 * a], py = xyz[3 * a + 1], pz = xyz[3 * a + 2];
    unsigned key[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
        const int j = lane + 64 * t;
        float d = __builtin_inff();
        if (j < m && j != a) {  // the query point itself is PCL's skipped first neighbour
            const float dx = px - xyz[3 * j], dy = py - xyz[3 * j + 1], dz = pz - xyz[3 * j + 2];
            d = dx * dx + dy * dy + dz * dz;
        }
        key[t] = __float_as_uint(d);
    }
    // the kk-th smallest key: the largest v with count(key < v) < kk, built bit by bit
    unsigned kth = 0;
    for (int bit = 30; bit >= 0; bit--) {
        const unsigned test = kth | (1u << bit);
        int c = 0;
#pragma unroll
        for (int t = 0; t < T; t++)
            c += key[t] < test ? 1 : 0;
        if (svo::wave_sum_small(c) < kk)
            kth = test;
    }
    int c_less = 0;
    double s_less = 0.;
#pragma unroll
    for (int t = 0; t < T; t++)
        if (key[t] < kth) {
            c_less++;
            s_less += (double)sqrtf(__uint_as_float(key[t]));
        }
    c_less = svo::wave_sum_small(c_less);
    const double total = wave_sum_double(s_less) + (double)(kk - c_less) * (double)sqrtf(__uint_as_float(kth));
    if (lane == 0)
        dist[a] = (float)(total / kk);
}

// mean / stddev of the distances in point order by one thread (PCL's own loop), then the keep mask
__global__ __launch_bounds__(1024) void sor_threshold_kernel(const float *__restrict__ dist,
                                                             const int *__restrict__ d_m, double stddev_mul, int cap,
                                                             uint8_t *__restrict__ mask)
{
    __shared__ double s_thr;
    const int m = *d_m;
    if (threadIdx.x == 0) {
        double sum = 0, sq_sum = 0;
        for (int i = 0; i < m; i++) {
            const float d = dist[i];
            sum += d;
            sq_sum += d * d;  // the product in float, as upstream
        }
        double thr = 1.7976931348623157e308;
        if (m > 1) {
            const double mean = sum / m;
            double variance = (sq_sum - sum * sum / m) / (m - 1);
            if (variance < 0)
                variance = 0;
            thr = mean + stddev_mul * sqrt(variance);
        }
        s_thr = thr;
    }
    __syncthreads();
    const double thr = s_thr;
    for (int i = threadIdx.x; i < cap; i += 1024)
        mask[i] = (i < m && (double)dist[i] <= thr) ? 1 : 0;
}

}  // namespace

// Device-pointer form.  cap points in, compacted outputs (cap capacity), *d_count = points kept,
// d_mean_dist (cap floats, may be null): the mean neighbour distance of every point that passed
// the z filter, *d_pass (may be null) their number.
int svo_launch_sor(svo_ctx *ctx, const float *xyz, const float *color, int cap, int mean_k, double stddev_mul,
                   float z_limit, float *xyz_out, float *color_out, int *d_count, float *d_mean_dist, int *d_pass)
{
    if (cap <= 0)
        return SVO_OK;
    if (cap > 64 * 144) {
        svo_set_error("svo_sor_filter: at most %d points per call", 64 * 144);
        return SVO_ERR_ARG;
    }
    int rc;
    if ((rc = ctx->w_a.ensure((size_t)cap * 12)) || (rc = ctx->w_b.ensure((size_t)cap * 12)) ||
        (rc = ctx->w_c.ensure((size_t)cap * 4)) || (rc = ctx->w_d.ensure((size_t)cap + 64)) ||
        (rc = ctx->w_e.ensure(64)))
        return rc;
    float *xyz_c = ctx->w_a.as<float>(), *col_c = ctx->w_b.as<float>();
    float *dist = d_mean_dist ? d_mean_dist : ctx->w_c.as<float>();
    uint8_t *mask = ctx->w_d.as<uint8_t>();
    int *d_m = d_pass ? d_pass : ctx->w_e.as<int>();
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(sor_zmask_kernel, dim3((cap + 255) / 256), dim3(256), 0, st, xyz, cap, z_limit, mask);
    if ((rc = svo_launch_compact(ctx, mask, cap, nullptr, xyz, 3, xyz_c, color, color ? 3 : 0, color ? col_c : nullptr,
                                 nullptr, 0, nullptr, d_m)))
        return rc;
    const dim3 grid((cap + 3) / 4), block(256);
    if (cap <= 64 * 16)
        hipLaunchKernelGGL(sor_knn_kernel<16>, grid, block, 0, st, xyz_c, d_m, mean_k, dist);
    else if (cap <= 64 * 48)
        hipLaunchKernelGGL(sor_knn_kernel<48>, grid, block, 0, st, xyz_c, d_m, mean_k, dist);
    else if (cap <= 64 * 80)
        hipLaunchKernelGGL(sor_knn_kernel<80>, grid, block, 0, st, xyz_c, d_m, mean_k, dist);
    else
        hipLaunchKernelGGL(sor_knn_kernel<144>, grid, block, 0, st, xyz_c, d_m, mean_k, dist);
    hipLaunchKernelGGL(sor_threshold_kernel, dim3(1), dim3(1024), 0, st, dist, d_m, stddev_mul, cap, mask);
    if ((rc = svo_launch_compact(ctx, mask, cap, d_m, xyz_c, 3, xyz_out, color ? col_c : nullptr, color ? 3 : 0,
                                 color ? color_out : nullptr, nullptr, 0, nullptr, d_count)))
        return rc;
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

extern "C" int svo_sor_filter(svo_ctx *ctx, const float *xyz, const float *color, int n, int mean_k,
                              double stddev_mul, float z_limit, float *xyz_out, float *color_out, int *n_out,
                              float *mean_dist_out, int *n_pass_out, int mem)
{
    SVO_CHECK_ARG(ctx && n >= 0 && mean_k > 0 && n_out);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    *n_out = 0;
    if (n_pass_out)
        *n_pass_out = 0;
    if (n == 0)
        return SVO_OK;
    SVO_CHECK_ARG(xyz && xyz_out && (!color || color_out));
    int rc;
    if ((rc = ctx->s_g.ensure(64)))
        return rc;
    int *d_cnt = ctx->s_g.as<int>();  // [0] kept, [1] passed the z filter
    const float *dx = xyz, *dc = color;
    float *ox = xyz_out, *oc = color_out, *od = mean_dist_out;
    if (mem == SVO_MEM_HOST) {
        if ((rc = ctx->s_a.ensure((size_t)n * 12)) || (rc = ctx->s_b.ensure((size_t)n * 12)) ||
            (rc = ctx->s_c.ensure((size_t)n * 12)) || (rc = ctx->s_d.ensure((size_t)n * 12)) ||
            (rc = ctx->s_e.ensure((size_t)n * 4)))
            return rc;
        SVO_HIP(hipMemcpyAsync(ctx->s_a.p, xyz, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
        if (color)
            SVO_HIP(hipMemcpyAsync(ctx->s_b.p, color, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
        dx = ctx->s_a.as<float>();
        dc = color ? ctx->s_b.as<float>() : nullptr;
        ox = ctx->s_c.as<float>();
        oc = ctx->s_d.as<float>();
        od = ctx->s_e.as<float>();
    }
    if ((rc = svo_launch_sor(ctx, dx, dc, n, mean_k, stddev_mul, z_limit, ox, oc, d_cnt, od, d_cnt + 1)))
        return rc;
    SVO_HIP(hipMemcpyAsync(ctx->pinned, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    const int kept = reinterpret_cast<int *>(ctx->pinned)[0], passed = reinterpret_cast<int *>(ctx->pinned)[1];
    *n_out = kept;
    if (n_pass_out)
        *n_pass_out = passed;
    if (mem == SVO_MEM_HOST) {
        if (kept > 0) {
            SVO_HIP(hipMemcpyAsync(xyz_out, ox, (size_t)kept * 12, hipMemcpyDeviceToHost, ctx->stream));
            if (color)
                SVO_HIP(hipMemcpyAsync(color_out, oc, (size_t)kept * 12, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (mean_dist_out && passed > 0)
            SVO_HIP(hipMemcpyAsync(mean_dist_out, od, (size_t)passed * 4, hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SVO_OK;
}
