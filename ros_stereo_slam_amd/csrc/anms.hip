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
    int *out_idx, *d_count;
    // optional payload gathered with the index list: out_x[i] = in_x[out_idx[i]]
    const float2 *in_a, *in_b;
    float2 *out_a, *out_b;
    const uint8_t *in_s;
    uint8_t *out_s;
    const int *gate;  // optional: the job's workgroups leave at once when *gate == 0 (chain runner: no keyframe)
};
template <int NJ> struct AnmsBatchN {  // NJ = 1: a chunk on its own (a sixteenth of the kernel arguments per launch)
    AnmsJob j[NJ];
};
using AnmsBatch = AnmsBatchN<SVO_LK_MAX_JOBS>;

// KPW keypoints per wavefront in the all-pairs passes: a stripe of the other keypoints is loaded once
// and compared against all KPW of them (an eighth of the loads), and a launch has an eighth of the
// waves -- a wave per keypoint was 35 k waves per launch, every one of which queued for a wave slot
// beside the tracking launches of the other contexts.
constexpr int KPW_GROUP = 8;  // lock-step groups
constexpr int KPW_LONE = 2;   // a lone problem has the chip to itself: four times the waves, a quarter of the work each

// order[rank] = i, rank = #keypoints sorting before i (response descending, index ascending);
// also emits the sorted (x, y, response) triples the radius pass streams through.
template <int KPW> __global__ __launch_bounds__(64) void anms_rank_kernel(AnmsBatchN<KPW == KPW_LONE ? 1 : SVO_LK_MAX_JOBS> batch, int n)
{
    svo_chain_priority();
    const AnmsJob &job = batch.j[blockIdx.y];
    if (job.gate && *job.gate == 0)
        return;
    const float2 *__restrict__ xy = job.xy;
    const float *__restrict__ resp = job.resp;
    int *__restrict__ order = job.order;
    float4 *__restrict__ sorted = job.sorted;
    const int lane = threadIdx.x & 63;
    const int i0 = blockIdx.x * KPW;  // one wave per workgroup (see svo_launch_anms_batch); KPW | 64
    if (i0 >= n)
        return;
    float ri[KPW];
    int cnt[KPW];
#pragma unroll
    for (int q = 0; q < KPW; q++) {
        ri[q] = resp[min(i0 + q, n - 1)];
        cnt[q] = 0;
    }
    // "j sorts before i"  <=>  r_j > r_i, or r_j == r_i and j < i.  Whole 64-entry stripes before the
    // keypoints' own stripe need one compare (>=), the ones after it one compare (>); only their own
    // stripe (the same for all KPW: i0 is a multiple of KPW, KPW divides 64) needs the tie-break.
    const float *__restrict__ rl = resp + lane;
    const int own = (i0 >> 6) << 6;
#pragma unroll 2
    for (int j0 = 0; j0 < own; j0 += 64) {
        const float rj = rl[j0];
#pragma unroll
        for (int q = 0; q < KPW; q++)
            cnt[q] += rj >= ri[q] ? 1 : 0;
    }
    {
        const int j = own + lane;
        if (j < n) {
            const float rj = rl[own];
#pragma unroll
            for (int q = 0; q < KPW; q++)
                cnt[q] += (rj > ri[q] || (rj == ri[q] && j < i0 + q)) ? 1 : 0;
        }
    }
    int j0 = own + 64;
#pragma unroll 2
    for (; j0 + 64 <= n; j0 += 64) {
        const float rj = rl[j0];
#pragma unroll
        for (int q = 0; q < KPW; q++)
            cnt[q] += rj > ri[q] ? 1 : 0;
    }
    if (j0 < n && j0 + lane < n) {
        const float rj = rl[j0];
#pragma unroll
        for (int q = 0; q < KPW; q++)
            cnt[q] += rj > ri[q] ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < KPW; q++) {
        const int rank = wave_sum_int(cnt[q]);
        if (lane == 0 && i0 + q < n) {
            order[rank] = i0 + q;
            const float2 p = xy[i0 + q];
            sorted[rank] = make_float4(p.x, p.y, ri[q], 0.f);
        }
    }
}

// squared suppression radius of the s-th sorted keypoint (DBL_MAX when nothing dominates it)
template <int KPW> __global__ __launch_bounds__(64) void anms_radius_kernel(AnmsBatchN<KPW == KPW_LONE ? 1 : SVO_LK_MAX_JOBS> batch, int n)
{
    svo_chain_priority();
    if (batch.j[blockIdx.y].gate && *batch.j[blockIdx.y].gate == 0)
        return;
    const float4 *__restrict__ sorted = batch.j[blockIdx.y].sorted;
    double *__restrict__ radius_sq = batch.j[blockIdx.y].radius;
    const int lane = threadIdx.x & 63;
    const int s0 = blockIdx.x * KPW;
    if (s0 >= n)
        return;
    float mx[KPW], my[KPW], thr[KPW];
    double best[KPW];
    unsigned active = 0;  // wave-uniform: keypoints whose scan has not ended yet
#pragma unroll
    for (int q = 0; q < KPW; q++) {
        const float4 me = sorted[min(s0 + q, n - 1)];
        mx[q] = me.x;
        my[q] = me.y;
        thr[q] = me.z * 1.11f;
        best[q] = DBL_MAX;
        if (s0 + q < n && s0 + q > 0)
            active |= 1u << q;
    }
    for (int j0 = 0; j0 < s0 + KPW - 1 && active; j0 += 64) {
        const int j = j0 + lane;
        const float4 o = sorted[min(j, n - 1)];
#pragma unroll
        for (int q = 0; q < KPW; q++) {
            if (!((active >> q) & 1u))
                continue;  // wave-uniform
            bool dom = false;
            if (j < s0 + q) {
                dom = o.z > thr[q];
                if (dom) {
                    const float dx = mx[q] - o.x, dy = my[q] - o.y;
                    const double d = (double)dx * dx + (double)dy * dy;
                    best[q] = d < best[q] ? d : best[q];
                }
            }
            // sorted by response: once a whole stripe fails, nothing later dominates either; and the
            // scan of keypoint s ends with the stripe that holds s - 1
            if (!__any(dom) || j0 + 64 >= s0 + q)
                active &= ~(1u << q);
        }
    }
#pragma unroll
    for (int q = 0; q < KPW; q++) {
        double b = best[q];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double o = __shfl_xor(b, off);
            b = o < b ? o : b;
        }
        if (lane == 0 && s0 + q < n)
            radius_sq[s0 + q] = b;
    }
}

// decision radius = the (keep+1)-th largest radius
template <int KPW> __global__ __launch_bounds__(64) void anms_decide_kernel(AnmsBatchN<KPW == KPW_LONE ? 1 : SVO_LK_MAX_JOBS> batch, int n, int keep)
{
    svo_chain_priority();
    if (batch.j[blockIdx.y].gate && *batch.j[blockIdx.y].gate == 0)
        return;
    const double *__restrict__ radius_sq = batch.j[blockIdx.y].radius;
    double *__restrict__ decision = batch.j[blockIdx.y].decision;
    const int lane = threadIdx.x & 63;
    const int s0 = blockIdx.x * KPW;
    if (s0 >= n)
        return;
    double ri[KPW];
    int gt[KPW], ge[KPW];
#pragma unroll
    for (int q = 0; q < KPW; q++) {
        ri[q] = radius_sq[min(s0 + q, n - 1)];
        gt[q] = ge[q] = 0;
    }
    const double *__restrict__ rl = radius_sq + lane;
    int j0 = 0;
#pragma unroll 2
    for (; j0 + 64 <= n; j0 += 64) {  // scalar bounds: no per-lane range test in the body
        const double rj = rl[j0];
#pragma unroll
        for (int q = 0; q < KPW; q++) {
            gt[q] += rj > ri[q] ? 1 : 0;
            ge[q] += rj >= ri[q] ? 1 : 0;
        }
    }
    if (j0 + lane < n) {
        const double rj = rl[j0];
#pragma unroll
        for (int q = 0; q < KPW; q++) {
            gt[q] += rj > ri[q] ? 1 : 0;
            ge[q] += rj >= ri[q] ? 1 : 0;
        }
    }
#pragma unroll
    for (int q = 0; q < KPW; q++) {
        const int g = wave_sum_int(gt[q]), e = wave_sum_int(ge[q]);
        // radiiSorted[keep] (descending, 0-based) == ri  <=>  gt <= keep < ge
        if (lane == 0 && s0 + q < n && g <= keep && keep < e)
            *decision = ri[q];  // every keypoint that qualifies writes the same value
    }
}

// kept[] = order[s] for every s with radius[s] >= decision radius, in sorted order (single workgroup scan);
// the payload arrays of the kept keypoints are gathered in the same pass
// 256 threads: a workgroup of one wave per SIMD starts beside the tracking launches at once (four tracking waves leave 96
// registers of a SIMD free; the 1024-thread form of rounds 1-3 needed four waves on every SIMD of one compute unit and fitted
// only because it stayed under 24 registers: 76 us on average in the bench for 13 us of work).
constexpr int GATHER_T = 256;
template <int NJ> __global__ __launch_bounds__(GATHER_T) void anms_gather_kernel(AnmsBatchN<NJ> batch, int n)
{
    svo_chain_priority();
    const AnmsJob &job = batch.j[blockIdx.x];  // one workgroup per job
    if (job.gate && *job.gate == 0)
        return;
    const double *__restrict__ radius_sq = job.radius;
    const double decision = *job.decision;
    const int *__restrict__ order = job.order;
    int *__restrict__ out_idx = job.out_idx;
    int *__restrict__ d_count = job.d_count;
    __shared__ int s_wave[GATHER_T / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (n + GATHER_T - 1) / GATHER_T;
    const int b = t * per, e = min(b + per, n);
    int cnt = 0;
    for (int i = b; i < e; i++)
        cnt += radius_sq[i] >= decision ? 1 : 0;
    // exclusive prefix of the threads' counts: shuffles inside a wave, the wave totals through LDS -- one barrier
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        incl += lane >= off ? o : 0;
    }
    if (lane == 63)
        s_wave[wave] = incl;
    __syncthreads();
    int pos = incl - cnt, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < GATHER_T / 64; w2++) {
        const int wt = s_wave[w2];
        pos += w2 < wave ? wt : 0;
        total += wt;
    }
    if (t == 0)
        *d_count = total;
    for (int i = b; i < e; i++)
        if (radius_sq[i] >= decision) {
            const int j = order[i];
            out_idx[pos] = j;
            if (job.out_a) {
                job.out_a[pos] = job.in_a[j];
                job.out_b[pos] = job.in_b[j];
                job.out_s[pos] = job.in_s[j];
            }
            pos++;
        }
}

// everything is kept (n <= numToKeep): the index list, its count and the payload in sorted order
template <int NJ> __global__ __launch_bounds__(64) void anms_payload_kernel(AnmsBatchN<NJ> batch, int n)
{
    const AnmsJob &job = batch.j[blockIdx.y];
    if (job.gate && *job.gate == 0)
        return;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i == 0)
        *job.d_count = n;
    if (i >= n)
        return;
    const int j = job.order[i];
    job.out_idx[i] = j;
    if (!job.out_a)
        return;
    job.out_a[i] = job.in_a[j];
    job.out_b[i] = job.in_b[j];
    if (job.out_s)
        job.out_s[i] = job.in_s[j];
}

}  // namespace

// Device form, several problems of the same size n.  out_idx: n ints (input indices of the kept
// keypoints, response order); d_count: device int.  Uses ctx->w_a..w_d as scratch.
int svo_launch_anms_batch(svo_ctx *ctx, int k, const float *const *xy, const float *const *resp, int n, int keep,
                          int *const *out_idx, int *const *d_count, const svo_anms_gather *gather, const int *const *gates)
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
        (rc = ctx->w_d.ensure(na * 16 * k)))
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
        j.sorted = ctx->w_d.as<float4>() + na * q;
        j.out_idx = out_idx[q];
        j.d_count = d_count[q];
        j.in_a = gather ? reinterpret_cast<const float2 *>(gather[q].in_a) : nullptr;
        j.in_b = gather ? reinterpret_cast<const float2 *>(gather[q].in_b) : nullptr;
        j.out_a = gather ? reinterpret_cast<float2 *>(gather[q].out_a) : nullptr;
        j.out_b = gather ? reinterpret_cast<float2 *>(gather[q].out_b) : nullptr;
        j.in_s = gather ? gather[q].in_s : nullptr;
        j.out_s = gather ? gather[q].out_s : nullptr;
        j.gate = gates ? gates[q] : nullptr;
    }
    // single-wave workgroups: beside a tracking launch (single-wave workgroups that take every freed wave
    // slot at once) a multi-wave workgroup waits until one CU has a slot free on several SIMDs together
    const bool lone = k == 1;
    const int kpw = lone ? KPW_LONE : KPW_GROUP;
    const dim3 wgrid((n + kpw - 1) / kpw, k), block(64);
    AnmsBatchN<1> one;
    one.j[0] = batch.j[0];
    if (lone)
        hipLaunchKernelGGL(anms_rank_kernel<KPW_LONE>, wgrid, block, 0, ctx->stream, one, n);
    else
        hipLaunchKernelGGL(anms_rank_kernel<KPW_GROUP>, wgrid, block, 0, ctx->stream, batch, n);
    if (n <= keep) {
        // everything is kept, in sorted order (one launch: index list, count, payload)
        if (lone)
            hipLaunchKernelGGL(anms_payload_kernel<1>, dim3((n + 63) / 64, k), dim3(64), 0, ctx->stream, one, n);
        else
            hipLaunchKernelGGL(anms_payload_kernel<SVO_LK_MAX_JOBS>, dim3((n + 63) / 64, k), dim3(64), 0, ctx->stream, batch, n);
        SVO_HIP(hipGetLastError());
        return SVO_OK;
    }
    if (lone) {
        hipLaunchKernelGGL(anms_radius_kernel<KPW_LONE>, wgrid, block, 0, ctx->stream, one, n);
        hipLaunchKernelGGL(anms_decide_kernel<KPW_LONE>, wgrid, block, 0, ctx->stream, one, n, keep);
        hipLaunchKernelGGL(anms_gather_kernel<1>, dim3(k), dim3(GATHER_T), 0, ctx->stream, one, n);
    } else {
        hipLaunchKernelGGL(anms_radius_kernel<KPW_GROUP>, wgrid, block, 0, ctx->stream, batch, n);
        hipLaunchKernelGGL(anms_decide_kernel<KPW_GROUP>, wgrid, block, 0, ctx->stream, batch, n, keep);
        hipLaunchKernelGGL(anms_gather_kernel<SVO_LK_MAX_JOBS>, dim3(k), dim3(GATHER_T), 0, ctx->stream, batch, n);
    }
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
