// geometry.hip -- per-point geometry kernels of the keyframe path, plus the order-preserving
// mask compaction the reference does with push_back loops.
//
//   triangulate   cv::triangulatePoints(P1,P2,x1,x2) + float dehomogenisation,
//                 src/triangulation.cpp:142-160 (DLT: 4x4 system per point, right singular
//                 vector of the smallest singular value by one-sided Jacobi, all in
//                 registers; negative-depth and w~0 points are NOT filtered, as upstream)
//   transform     [R|t] (3x4 double) applied to float points, src/keyFrameManagement.cpp:20-30
//   colours       img.at<Vec3b>(int(y), int(x)) as 3 floats, include/monoUtils.h:180-193
//   compact       src/tracking.cpp:20-27, 35-42, 66-84 (status / mask filters)
// All are one thread per point, bandwidth-trivial (tens of bytes per point).
#include <cfloat>

#include "svo_internal.h"

namespace {

struct Mat34 {
    double m[12];
};

// right singular vector of the smallest singular value of a 4x4 matrix (Hestenes Jacobi)
__device__ void smallest_right_singular_vector4(double (&A)[4][4], double (&v)[4])
{
    double V[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            V[i][j] = i == j ? 1. : 0.;
    for (int sweep = 0; sweep < 30; sweep++) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                double al = 0, be = 0, ga = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    al += A[i][p] * A[i][p];
                    be += A[i][q] * A[i][q];
                    ga += A[i][p] * A[i][q];
                }
                if (fabs(ga) <= DBL_EPSILON * sqrt(al * be) || ga == 0)
                    continue;
                rotated = true;
                const double zeta = (be - al) / (2. * ga);
                const double t = (zeta >= 0 ? 1. : -1.) / (fabs(zeta) + sqrt(1. + zeta * zeta));
                const double c = 1. / sqrt(1. + t * t), s = c * t;
#pragma unroll
                for (int i = 0; i < 4; i++) {
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
    int best = 0;
    double bn = DBL_MAX;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double nn = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
            nn += A[i][j] * A[i][j];
        if (nn < bn) {
            bn = nn;
            best = j;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
        v[i] = best == 0 ? V[i][0] : best == 1 ? V[i][1] : best == 2 ? V[i][2] : V[i][3];
}

struct TriJob {
    const float2 *x1, *x2;
    int n_host;
    const int *d_n;
    float *out_xyz, *out_h;
    Mat34 Rt;
    int apply_rt;
    float *out_world;
    int *h_count;  // pinned host int that receives the live point count (the host waits on the stream), or null
    VoChain *chain;  // chain mode (svo_tri_job::chain) or null
    float2 *out_x1;  // optional copy of x1
    const uint8_t *cimg;  // optional colour gather at x1: level 0 of the left pyramid (pixel (0,0)), its pitch and size
    int cpitch, cw, chh, cc;
    float *cout;
};
template <int NJ> struct TriBatchN {  // blockIdx.y picks the job; NJ = 1: a lone problem (a sixteenth of the kernel arguments)
    TriJob j[NJ];
};
using TriBatch = TriBatchN<SVO_LK_MAX_JOBS>;

template <int NJ> __global__ __launch_bounds__(128) void triangulate_kernel(Mat34 P1, Mat34 P2, TriBatchN<NJ> batch)
{
    svo_chain_priority();
    const TriJob &job = batch.j[blockIdx.y];
    VoChain *chain = job.chain;
    if (chain && chain->kf == 0)
        return;  // chain runner: the frame is no keyframe (or the chain halted)
    const float2 *__restrict__ x1 = job.x1, *__restrict__ x2 = job.x2;
    const int n_host = job.n_host;
    const int *__restrict__ d_n = job.d_n;
    float *__restrict__ out_xyz = job.out_xyz, *__restrict__ out_h = job.out_h;
    Mat34 Rt = job.Rt;
    if (chain) {  // insertKeyFrames places the cloud with the pose the frame was localised at (keyFrameManagement.cpp:20-30)
#pragma unroll
        for (int r = 0; r < 3; r++) {
            Rt.m[4 * r] = chain->R[3 * r];
            Rt.m[4 * r + 1] = chain->R[3 * r + 1];
            Rt.m[4 * r + 2] = chain->R[3 * r + 2];
            Rt.m[4 * r + 3] = chain->t[r];
        }
    }
    const int apply_rt = job.apply_rt;
    float *__restrict__ out_world = job.out_world;
    const int n = d_n ? min(*d_n, n_host) : n_host;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && job.h_count)
        *job.h_count = n;  // what a one-thread launch of its own did before (store_counts_kernel)
    if (i == 0 && chain) {  // the new reference set's size; a chunk cannot go on from fewer than 5 points
        chain->nref = n;
        chain->kf_n = n;
        if (n < 5) {
            chain->run = 0;
            chain->halt_code = SVO_HALT_FEW_REF;
        }
    }
    if (i >= n)
        return;
    double A[4][4], v[4];
    const float2 a = x1[i], b = x2[i];
    if (job.out_x1)
        job.out_x1[i] = a;
    if (job.cout) {  // as colors_kernel: truncation, clamped where the reference reads out of bounds
        int cx = (int)a.x, cy = (int)a.y;
        cx = cx < 0 ? 0 : (cx >= job.cw ? job.cw - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= job.chh ? job.chh - 1 : cy);
        const uint8_t *px = job.cimg + (size_t)cy * job.cpitch + cx * job.cc;
#pragma unroll
        for (int k = 0; k < 3; k++)
            job.cout[3 * i + k] = (float)px[job.cc >= 3 ? k : 0];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        A[0][k] = (double)a.x * P1.m[8 + k] - P1.m[k];
        A[1][k] = (double)a.y * P1.m[8 + k] - P1.m[4 + k];
        A[2][k] = (double)b.x * P2.m[8 + k] - P2.m[k];
        A[3][k] = (double)b.y * P2.m[8 + k] - P2.m[4 + k];
    }
    smallest_right_singular_vector4(A, v);
    const float h0 = (float)v[0], h1 = (float)v[1], h2 = (float)v[2], h3 = (float)v[3];
    if (out_h) {
        out_h[4 * i] = h0;
        out_h[4 * i + 1] = h1;
        out_h[4 * i + 2] = h2;
        out_h[4 * i + 3] = h3;
    }
    const float x = h0 / h3, y = h1 / h3, z = h2 / h3;
    out_xyz[3 * i] = x;
    out_xyz[3 * i + 1] = y;
    out_xyz[3 * i + 2] = z;
    if (apply_rt) {
#pragma unroll
        for (int r = 0; r < 3; r++)
            out_world[3 * i + r] =
                (float)(Rt.m[4 * r] * x + Rt.m[4 * r + 1] * y + Rt.m[4 * r + 2] * z + Rt.m[4 * r + 3]);
    }
}

// The pose-dependent rest of a keyframe whose triangulation ran ahead of the decision (the pipelined chunk's stereo
// stream): exactly what triangulate_kernel does in chain mode after its DLT -- reference-set size and halt rule, the
// keyframe's 2-D set, `colors`, the camera-frame cloud, and the cloud placed with the pose the frame was localised at
// (src/keyFrameManagement.cpp:18-30) -- reading the camera-frame points instead of solving for them.
struct PlaceArgs {
    VoChain *chain;
    const float2 *x1;
    const float *xyz;
    int n_host;
    const int *d_n;
    float2 *out_x1;
    float *out_cam, *out_world;
    const uint8_t *cimg;
    int cpitch, cw, chh, cc;
    float *cout;
};
// what: 1 = the pose-free part (reference-set size and halt rule, the 2-D set, colours: all the next tracking pass
// needs), 2 = the clouds (camera frame and placed with the refined pose), 3 = both.
__global__ __launch_bounds__(64) void keyframe_place_kernel(PlaceArgs a, int what)
{
    svo_chain_priority();
    VoChain *chain = a.chain;
    if (chain->kf == 0)
        return;
    const int n = min(*a.d_n, a.n_host);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && (what & 1)) {
        chain->nref = n;
        chain->kf_n = n;
        if (n < 5) {
            chain->run = 0;
            chain->halt_code = SVO_HALT_FEW_REF;
        }
    }
    if (i >= n)
        return;
    if (what & 1) {
        const float2 p = a.x1[i];
        a.out_x1[i] = p;
        if (a.cout) {
            int cx = (int)p.x, cy = (int)p.y;
            cx = cx < 0 ? 0 : (cx >= a.cw ? a.cw - 1 : cx);
            cy = cy < 0 ? 0 : (cy >= a.chh ? a.chh - 1 : cy);
            const uint8_t *px = a.cimg + (size_t)cy * a.cpitch + cx * a.cc;
#pragma unroll
            for (int k = 0; k < 3; k++)
                a.cout[3 * i + k] = (float)px[a.cc >= 3 ? k : 0];
        }
    }
    if (what & 2) {
        const float x = a.xyz[3 * i], y = a.xyz[3 * i + 1], z = a.xyz[3 * i + 2];
        a.out_cam[3 * i] = x;
        a.out_cam[3 * i + 1] = y;
        a.out_cam[3 * i + 2] = z;
#pragma unroll
        for (int r = 0; r < 3; r++)
            a.out_world[3 * i + r] =
                (float)(chain->R[3 * r] * x + chain->R[3 * r + 1] * y + chain->R[3 * r + 2] * z + chain->t[r]);
    }
}

__global__ __launch_bounds__(256) void transform_kernel(Mat34 Rt, const float *__restrict__ in, int n_host,
                                                        const int *__restrict__ d_n, float *__restrict__ out)
{
    __builtin_amdgcn_s_setprio(3);  // short latency-bound kernel: win issue arbitration against co-resident LK waves
    const int n = d_n ? min(*d_n, n_host) : n_host;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
#pragma unroll
    for (int r = 0; r < 3; r++)
        out[3 * i + r] = (float)(Rt.m[4 * r] * x + Rt.m[4 * r + 1] * y + Rt.m[4 * r + 2] * z + Rt.m[4 * r + 3]);
}

__global__ __launch_bounds__(256) void colors_kernel(const uint8_t *__restrict__ lvl0, int pitch, int w, int h,
                                                     int c, const float2 *__restrict__ xy, int n_host,
                                                     const int *__restrict__ d_n, float *__restrict__ out)
{
    __builtin_amdgcn_s_setprio(3);  // short latency-bound kernel: win issue arbitration against co-resident LK waves
    const int n = d_n ? min(*d_n, n_host) : n_host;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    int x = (int)xy[i].x, y = (int)xy[i].y;
    x = x < 0 ? 0 : (x >= w ? w - 1 : x);  // the reference reads out of bounds here; clamped
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    const uint8_t *px = lvl0 + (size_t)y * pitch + x * c;
#pragma unroll
    for (int k = 0; k < 3; k++)
        out[3 * i + k] = (float)px[c >= 3 ? k : 0];
}

// Order-preserving compaction of up to three float arrays by a byte mask.  Single-wave workgroups
// with no communication between them: wave w owns the w-th segment of `seg` elements and finds its
// output base by counting the kept entries of the whole mask prefix itself (at most a few KB, 16
// bytes per lane and load) -- redundant work, but no second launch, no look-back and no multi-wave
// workgroup (which would wait for wave slots on several SIMDs of one CU beside a tracking launch).
struct CompactArgs {
    const float *in[3];
    float *out[3];
    int stride[3];
    const uint8_t *mask;
    int n_host;
    const int *d_n;
    int *d_count;
    const int *gate;
    const int *alt_sel;
    const uint8_t *alt_mask;
    const float *alt_in[3];
    const int *alt_d_n;
};
template <int NJ> struct CompactBatchN {  // blockIdx.y picks the job; NJ = 1: a lone problem
    CompactArgs j[NJ];
};
using CompactBatch = CompactBatchN<SVO_LK_MAX_JOBS>;

__device__ __forceinline__ int wave_sum_int(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o);
    return v;
}

// number of bytes equal to 1 in a dword
__device__ __forceinline__ int count_ones_bytes(uint32_t x)
{
    const uint32_t y = x ^ 0x01010101u;                                   // zero byte <=> the mask byte was 1
    const uint32_t t = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);  // 0x80 in every zero byte
    return __popc(t);
}

template <int NJ> __global__ __launch_bounds__(64) void compact_kernel(CompactBatchN<NJ> batch, int seg)
{
    svo_chain_priority();
    const CompactArgs &args = batch.j[blockIdx.y];
    if (args.gate && *args.gate == 0)
        return;
    const bool alt = args.alt_sel && *args.alt_sel != 0;  // wave-uniform
    const uint8_t *__restrict__ mask = alt && args.alt_mask ? args.alt_mask : args.mask;
    const float *src_of[3];
#pragma unroll
    for (int a = 0; a < 3; a++)
        src_of[a] = alt && args.alt_in[a] ? args.alt_in[a] : args.in[a];
    const int *d_n = alt && args.alt_d_n ? args.alt_d_n : args.d_n;
    const int n = d_n ? min(*d_n, args.n_host) : args.n_host;
    const int lane = threadIdx.x;
    const int first = blockIdx.x * seg;  // this wave's segment: [first, last)
    if (first >= n && !(n <= 0 && blockIdx.x == 0))
        return;
    const int last = min(first + seg, n);
    // kept entries before the segment
    int part = 0;
    if ((reinterpret_cast<uintptr_t>(mask) & 15) == 0) {
        for (int i = lane * 16; i < first; i += 64 * 16) {  // `first` is a multiple of 64: whole dwords, all inside the prefix
            const uint4 v = *reinterpret_cast<const uint4 *>(mask + i);
            part += i + 0 < first ? count_ones_bytes(v.x) : 0;
            part += i + 4 < first ? count_ones_bytes(v.y) : 0;
            part += i + 8 < first ? count_ones_bytes(v.z) : 0;
            part += i + 12 < first ? count_ones_bytes(v.w) : 0;
        }
    } else {
        for (int i = lane; i < first; i += 64)
            part += mask[i] == 1 ? 1 : 0;
    }
    int pos0 = __builtin_amdgcn_readfirstlane(wave_sum_int(part));
    for (int start = first; start < last; start += 64) {
        const int i = start + lane;
        const bool keep = i < last && mask[i] == 1;
        const unsigned long long bal = __ballot(keep);
        if (keep) {
            const int pos = pos0 + __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
            for (int a = 0; a < 3; a++)
                if (args.in[a]) {
                    const int st = args.stride[a];
                    const float *__restrict__ src = src_of[a];
                    for (int k = 0; k < st; k++)
                        args.out[a][(size_t)pos * st + k] = src[(size_t)i * st + k];
                }
        }
        pos0 += __popcll(bal);
    }
    if (args.d_count && last >= n && lane == 0)  // the wave that owns the end of the array
        *args.d_count = pos0;
}

Mat34 to_mat34(const double *p)
{
    Mat34 m;
    if (p)
        memcpy(m.m, p, sizeof(m.m));
    else
        memset(m.m, 0, sizeof(m.m));
    return m;
}

}  // namespace

int svo_launch_triangulate_batch(svo_ctx *ctx, const double *P1, const double *P2, int k, const svo_tri_job *jobs)
{
    if (k <= 0)
        return SVO_OK;
    if (k > SVO_LK_MAX_JOBS) {
        svo_set_error("triangulate: at most %d jobs per launch", SVO_LK_MAX_JOBS);
        return SVO_ERR_ARG;
    }
    TriBatch batch;
    int cap_max = 0;
    for (int a = 0; a < SVO_LK_MAX_JOBS; a++) {
        const svo_tri_job &h = jobs[a < k ? a : 0];
        TriJob &j = batch.j[a];
        j.x1 = reinterpret_cast<const float2 *>(h.x1);
        j.x2 = reinterpret_cast<const float2 *>(h.x2);
        j.n_host = h.cap;
        j.d_n = h.d_n;
        j.out_xyz = h.out_xyz;
        j.out_h = h.out_h;
        j.Rt = to_mat34(h.Rt);
        j.apply_rt = ((h.Rt || h.chain) && h.out_world) ? 1 : 0;
        j.out_world = h.out_world;
        j.h_count = a < k ? h.h_count : nullptr;
        j.chain = h.chain;
        j.out_x1 = reinterpret_cast<float2 *>(h.out_x1);
        j.cimg = h.color_src && h.color_out ? h.color_src->dev.lvl[0] : nullptr;
        j.cpitch = h.color_src ? h.color_src->dev.pitch[0] : 0;
        j.cw = h.color_src ? h.color_src->w : 0;
        j.chh = h.color_src ? h.color_src->h : 0;
        j.cc = h.color_src ? h.color_src->c : 0;
        j.cout = h.color_src ? h.color_out : nullptr;
        if (a < k)
            cap_max = h.cap > cap_max ? h.cap : cap_max;
    }
    if (cap_max <= 0)
        return SVO_OK;
    ScopedKernelTime tm(ctx, SVO_K_TRIANGULATE);
    if (k == 1) {
        TriBatchN<1> one;
        one.j[0] = batch.j[0];
        hipLaunchKernelGGL(triangulate_kernel<1>, dim3((cap_max + 63) / 64, k), dim3(64), 0, ctx->stream, to_mat34(P1),
                           to_mat34(P2), one);
    } else {
        hipLaunchKernelGGL(triangulate_kernel<SVO_LK_MAX_JOBS>, dim3((cap_max + 63) / 64, k), dim3(64), 0, ctx->stream,
                           to_mat34(P1), to_mat34(P2), batch);
    }
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

int svo_launch_keyframe_place(svo_ctx *ctx, VoChain *chain, const float *x1, const float *xyz, int cap, const int *d_n,
                              float *out_x1, float *out_cam, float *out_world, const svo_pyramid *color_src, float *color_out,
                              int what)
{
    if (cap <= 0)
        return SVO_OK;
    if (!chain || !x1 || !xyz || !d_n || !out_x1 || !out_cam || !out_world) {
        svo_set_error("keyframe_place: null argument");
        return SVO_ERR_ARG;
    }
    PlaceArgs a;
    a.chain = chain;
    a.x1 = reinterpret_cast<const float2 *>(x1);
    a.xyz = xyz;
    a.n_host = cap;
    a.d_n = d_n;
    a.out_x1 = reinterpret_cast<float2 *>(out_x1);
    a.out_cam = out_cam;
    a.out_world = out_world;
    a.cimg = color_src && color_out ? color_src->dev.lvl[0] : nullptr;
    a.cpitch = color_src ? color_src->dev.pitch[0] : 0;
    a.cw = color_src ? color_src->w : 0;
    a.chh = color_src ? color_src->h : 0;
    a.cc = color_src ? color_src->c : 0;
    a.cout = color_src ? color_out : nullptr;
    ScopedKernelTime tm(ctx, SVO_K_TRIANGULATE);
    hipLaunchKernelGGL(keyframe_place_kernel, dim3((cap + 63) / 64), dim3(64), 0, ctx->stream, a, what);
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

int svo_launch_triangulate(svo_ctx *ctx, const double *P1, const double *P2, const float *x1, const float *x2,
                           int cap, const int *d_n, float *out_xyz, float *out_h, const double *Rt,
                           float *out_world)
{
    svo_tri_job j = {x1, x2, cap, d_n, out_xyz, out_h, Rt, out_world, nullptr};
    return svo_launch_triangulate_batch(ctx, P1, P2, 1, &j);
}

int svo_launch_transform(svo_ctx *ctx, const double *Rt, const float *in, int cap, const int *d_n, float *out)
{
    if (cap <= 0)
        return SVO_OK;
    hipLaunchKernelGGL(transform_kernel, dim3((cap + 255) / 256), dim3(256), 0, ctx->stream, to_mat34(Rt), in, cap,
                       d_n, out);
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

int svo_launch_colors(svo_ctx *ctx, const svo_pyramid *pyr, const float *xy, int cap, const int *d_n, float *out)
{
    if (cap <= 0)
        return SVO_OK;
    hipLaunchKernelGGL(colors_kernel, dim3((cap + 255) / 256), dim3(256), 0, ctx->stream, pyr->dev.lvl[0],
                       pyr->dev.pitch[0], pyr->w, pyr->h, pyr->c, reinterpret_cast<const float2 *>(xy), cap, d_n,
                       out);
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

int svo_launch_compact_batch(svo_ctx *ctx, int n_jobs, const svo_compact_job *jobs)
{
    if (n_jobs <= 0)
        return SVO_OK;
    if (n_jobs > SVO_LK_MAX_JOBS) {
        svo_set_error("compact: at most %d jobs per launch", SVO_LK_MAX_JOBS);
        return SVO_ERR_ARG;
    }
    CompactBatch batch;
    for (int k = 0; k < SVO_LK_MAX_JOBS; k++) {
        const svo_compact_job &h = jobs[k < n_jobs ? k : 0];
        CompactArgs &a = batch.j[k];
        for (int q = 0; q < 3; q++) {
            a.in[q] = h.in[q];
            a.out[q] = h.out[q];
            a.stride[q] = h.stride[q];
        }
        a.mask = h.mask;
        a.n_host = h.cap;
        a.d_n = h.d_n;
        a.d_count = h.d_count;
        a.gate = h.gate;
        a.alt_sel = h.alt_sel;
        a.alt_mask = h.alt_mask;
        for (int q = 0; q < 3; q++)
            a.alt_in[q] = h.alt_in[q];
        a.alt_d_n = h.alt_d_n;
    }
    int cap_max = 0;
    for (int k = 0; k < n_jobs; k++)
        cap_max = jobs[k].cap > cap_max ? jobs[k].cap : cap_max;
    const int seg = ((cap_max + 15) / 16 + 63) / 64 * 64;  // 16 segments, whole 64-element strips
    if (n_jobs == 1) {
        CompactBatchN<1> one;
        one.j[0] = batch.j[0];
        hipLaunchKernelGGL(compact_kernel<1>, dim3(seg > 0 ? (cap_max + seg - 1) / seg : 1, n_jobs), dim3(64), 0, ctx->stream,
                           one, seg > 0 ? seg : 64);
    } else {
        hipLaunchKernelGGL(compact_kernel<SVO_LK_MAX_JOBS>, dim3(seg > 0 ? (cap_max + seg - 1) / seg : 1, n_jobs), dim3(64), 0,
                           ctx->stream, batch, seg > 0 ? seg : 64);
    }
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

int svo_launch_compact(svo_ctx *ctx, const uint8_t *mask, int cap, const int *d_n, const float *in_a, int stride_a,
                       float *out_a, const float *in_b, int stride_b, float *out_b, const float *in_c, int stride_c,
                       float *out_c, int *d_count)
{
    svo_compact_job j;
    j.mask = mask;
    j.cap = cap;
    j.d_n = d_n;
    j.in[0] = in_a;
    j.out[0] = out_a;
    j.stride[0] = stride_a;
    j.in[1] = in_b;
    j.out[1] = out_b;
    j.stride[1] = stride_b;
    j.in[2] = in_c;
    j.out[2] = out_c;
    j.stride[2] = stride_c;
    j.d_count = d_count;
    return svo_launch_compact_batch(ctx, 1, &j);
}

extern "C" {

int svo_stereo_projections(double fx, double fy, double cx, double cy, double baseline, double *P1, double *P2)
{
    SVO_CHECK_ARG(P1 && P2);
    const double K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            P1[4 * i + j] = K[3 * i + j];
            P2[4 * i + j] = K[3 * i + j];
        }
        P1[4 * i + 3] = 0;
        P2[4 * i + 3] = K[3 * i] * (-baseline);
    }
    return SVO_OK;
}

int svo_triangulate(svo_ctx *ctx, const double *P1, const double *P2, const float *x1, const float *x2, int n,
                    float *out_xyz, float *out_h4, int mem)
{
    SVO_CHECK_ARG(ctx && P1 && P2 && n >= 0);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    if (n == 0)
        return SVO_OK;
    SVO_CHECK_ARG(x1 && x2 && out_xyz);
    if (mem == SVO_MEM_DEVICE)
        return svo_launch_triangulate(ctx, P1, P2, x1, x2, n, nullptr, out_xyz, out_h4, nullptr, nullptr);
    int rc;
    if ((rc = ctx->s_a.ensure((size_t)n * 8)) || (rc = ctx->s_b.ensure((size_t)n * 8)) ||
        (rc = ctx->s_c.ensure((size_t)n * 12)) || (rc = ctx->s_d.ensure((size_t)n * 16)))
        return rc;
    SVO_HIP(hipMemcpyAsync(ctx->s_a.p, x1, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    SVO_HIP(hipMemcpyAsync(ctx->s_b.p, x2, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = svo_launch_triangulate(ctx, P1, P2, ctx->s_a.as<float>(), ctx->s_b.as<float>(), n, nullptr,
                                ctx->s_c.as<float>(), ctx->s_d.as<float>(), nullptr, nullptr);
    if (rc)
        return rc;
    SVO_HIP(hipMemcpyAsync(out_xyz, ctx->s_c.p, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
    if (out_h4)
        SVO_HIP(hipMemcpyAsync(out_h4, ctx->s_d.p, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    return SVO_OK;
}

int svo_transform_points(svo_ctx *ctx, const double *Rt, const float *in_xyz, int n, float *out_xyz, int mem)
{
    SVO_CHECK_ARG(ctx && Rt && n >= 0);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    if (n == 0)
        return SVO_OK;
    SVO_CHECK_ARG(in_xyz && out_xyz);
    if (mem == SVO_MEM_DEVICE)
        return svo_launch_transform(ctx, Rt, in_xyz, n, nullptr, out_xyz);
    int rc;
    if ((rc = ctx->s_a.ensure((size_t)n * 12)) || (rc = ctx->s_b.ensure((size_t)n * 12)))
        return rc;
    SVO_HIP(hipMemcpyAsync(ctx->s_a.p, in_xyz, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    rc = svo_launch_transform(ctx, Rt, ctx->s_a.as<float>(), n, nullptr, ctx->s_b.as<float>());
    if (rc)
        return rc;
    SVO_HIP(hipMemcpyAsync(out_xyz, ctx->s_b.p, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    return SVO_OK;
}

int svo_get_colors(svo_ctx *ctx, const svo_pyramid *pyr, const float *xy, int n, float *out_bgr, int mem)
{
    SVO_CHECK_ARG(ctx && pyr && n >= 0);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    if (n == 0)
        return SVO_OK;
    SVO_CHECK_ARG(xy && out_bgr);
    if (mem == SVO_MEM_DEVICE)
        return svo_launch_colors(ctx, pyr, xy, n, nullptr, out_bgr);
    int rc;
    if ((rc = ctx->s_a.ensure((size_t)n * 8)) || (rc = ctx->s_b.ensure((size_t)n * 12)))
        return rc;
    SVO_HIP(hipMemcpyAsync(ctx->s_a.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = svo_launch_colors(ctx, pyr, ctx->s_a.as<float>(), n, nullptr, ctx->s_b.as<float>());
    if (rc)
        return rc;
    SVO_HIP(hipMemcpyAsync(out_bgr, ctx->s_b.p, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    return SVO_OK;
}

int svo_compact(svo_ctx *ctx, const uint8_t *mask, int n, const float *in_a, int stride_a, float *out_a,
                const float *in_b, int stride_b, float *out_b, const float *in_c, int stride_c, float *out_c,
                int *count, int mem)
{
    SVO_CHECK_ARG(ctx && n >= 0);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    SVO_CHECK_ARG(stride_a >= 0 && stride_b >= 0 && stride_c >= 0 && stride_a <= 8 && stride_b <= 8 && stride_c <= 8);
    if (n == 0) {
        if (count && mem == SVO_MEM_HOST)
            *count = 0;
        return SVO_OK;
    }
    SVO_CHECK_ARG(mask);
    if (mem == SVO_MEM_DEVICE)
        return svo_launch_compact(ctx, mask, n, nullptr, in_a, stride_a, out_a, in_b, stride_b, out_b, in_c,
                                  stride_c, out_c, count);
    const float *in[3] = {in_a, in_b, in_c};
    float *out[3] = {out_a, out_b, out_c};
    const int st[3] = {stride_a, stride_b, stride_c};
    DevBuf *bi[3] = {&ctx->s_a, &ctx->s_b, &ctx->s_c};
    DevBuf *bo[3] = {&ctx->s_d, &ctx->s_e, &ctx->s_f};
    int rc;
    const size_t count_off = ((size_t)n + 63) & ~(size_t)63;  // the count lives behind the mask bytes
    if ((rc = ctx->s_g.ensure(count_off + 64)))
        return rc;
    SVO_HIP(hipMemcpyAsync(ctx->s_g.p, mask, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    const float *din[3] = {nullptr, nullptr, nullptr};
    float *dout[3] = {nullptr, nullptr, nullptr};
    for (int a = 0; a < 3; a++)
        if (in[a] && out[a] && st[a] > 0) {
            size_t bytes = (size_t)n * st[a] * 4;
            if ((rc = bi[a]->ensure(bytes)) || (rc = bo[a]->ensure(bytes)))
                return rc;
            SVO_HIP(hipMemcpyAsync(bi[a]->p, in[a], bytes, hipMemcpyHostToDevice, ctx->stream));
            din[a] = bi[a]->as<float>();
            dout[a] = bo[a]->as<float>();
        }
    int *dcount = reinterpret_cast<int *>(ctx->s_g.as<uint8_t>() + count_off);
    rc = svo_launch_compact(ctx, ctx->s_g.as<uint8_t>(), n, nullptr, din[0], st[0], dout[0], din[1], st[1], dout[1],
                            din[2], st[2], dout[2], dcount);
    if (rc)
        return rc;
    SVO_HIP(hipMemcpyAsync(ctx->pinned, dcount, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    const int k = *reinterpret_cast<int *>(ctx->pinned);
    for (int a = 0; a < 3; a++)
        if (dout[a])
            SVO_HIP(hipMemcpyAsync(out[a], dout[a], (size_t)k * st[a] * 4, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    if (count)
        *count = k;
    return SVO_OK;
}

// include/svo_math.h on the device, element by element (the parity tests compare it with the host build)
__global__ void math_eval_kernel(int fn, const double *__restrict__ x, int n, double *__restrict__ y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double v = x[i];
    double r;
    switch (fn) {
    case SVO_MATH_SIN:
        r = svo_sin(v);
        break;
    case SVO_MATH_COS:
        r = svo_cos(v);
        break;
    case SVO_MATH_ACOS:
        r = svo_acos(v);
        break;
    case SVO_MATH_CBRT:
        r = svo_cbrt(v);
        break;
    default:
        r = svo_log(v);
        break;
    }
    y[i] = r;
}

int svo_math_eval(svo_ctx *ctx, int fn, const double *x, int n, double *y, int mem)
{
    SVO_CHECK_ARG(ctx && n >= 0 && (n == 0 || (x && y)) && fn >= SVO_MATH_SIN && fn <= SVO_MATH_LOG);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    if (n == 0)
        return SVO_OK;
    SVO_HIP(hipSetDevice(ctx->device));
    const double *dx = x;
    double *dy = y;
    if (mem == SVO_MEM_HOST) {
        int rc;
        if ((rc = ctx->s_a.ensure((size_t)n * 8)) || (rc = ctx->s_b.ensure((size_t)n * 8)))
            return rc;
        SVO_HIP(hipMemcpyAsync(ctx->s_a.p, x, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        dx = ctx->s_a.as<double>();
        dy = ctx->s_b.as<double>();
    }
    hipLaunchKernelGGL(math_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, fn, dx, n, dy);
    SVO_HIP(hipGetLastError());
    if (mem == SVO_MEM_HOST) {
        SVO_HIP(hipMemcpyAsync(y, dy, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SVO_OK;
}

}  // extern "C"
