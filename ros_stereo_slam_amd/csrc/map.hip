// map.hip -- the keyframe map kept in HBM and its re-projection after a pose-graph solve.
//
// Replaces visualSLAM::updateOdometry(vector<Isometry3d>&), src/optimizationStuff.cpp:17-47: after
// globalOptimize every stored keyframe record's camera-frame cloud (kf.ref3dCoords, the
// `untransformed` cloud of src/VisualSLAM.cpp:156-160) is transformed again with
// [R_old | t_new] -- the record's UN-optimised rotation and the optimised translation of its
// trajectory entry (:29-41) -- and mapHistory is rebuilt from the records with `retrack` (:43-45).
// Upstream does that record by record on the host through update3dtransformation
// (src/keyFrameManagement.cpp:33-46) and repeats the work for every non-keyframe record, whose
// result it throws away.
//
// Here the camera-frame clouds stay resident (one float3 array, records back to back, a record id per
// point) and ONE launch re-transforms all of them: thread per point, the record's 12 doubles read
// through the scalar cache, p = (float)(R x + t) in double as upstream.  HBM-bound: 12 B in + 4 B id +
// 12 B out per point; a 4500-frame run with 2000-point keyframes every second frame is 126 MB, i.e.
// ~20 us at the 6.3 TB/s this chip streams, against ~50 ms for the host loop.
#include "svo_internal.h"

struct svo_map {
    struct Rec {
        int traj_index, retrack, n;
        size_t off;
        double Rt[12];  // [R_old | t]: rows of 4
    };
    svo_ctx *ctx = nullptr;
    std::vector<Rec> rec;
    size_t n_pts = 0, cap_pts = 0;
    float *d_cam = nullptr, *d_world = nullptr;
    int *d_id = nullptr;
    DevBuf d_rt;  // 12 doubles per record, uploaded per update
};

namespace {

__global__ __launch_bounds__(256) void map_transform_kernel(const float *__restrict__ cam, const int *__restrict__ id,
                                                            const double *__restrict__ rt, size_t first, size_t n,
                                                            float *__restrict__ world)
{
    const size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= first + n)
        return;
    const double *m = rt + 12 * (size_t)id[i];
    const float x = cam[3 * i], y = cam[3 * i + 1], z = cam[3 * i + 2];
#pragma unroll
    for (int r = 0; r < 3; r++)
        world[3 * i + r] = (float)(m[4 * r] * x + m[4 * r + 1] * y + m[4 * r + 2] * z + m[4 * r + 3]);
}

// one record, its 12 doubles in the kernel arguments: what svo_map_add_keyframe launches (no upload, no wait)
struct MapRt {
    double v[12];
};
__global__ __launch_bounds__(256) void map_transform_one_kernel(const float *__restrict__ cam, MapRt rt, size_t first,
                                                                size_t n, float *__restrict__ world, int *__restrict__ id,
                                                                int id_value)
{
    const size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= first + n)
        return;
    id[i] = id_value;
    const float x = cam[3 * i], y = cam[3 * i + 1], z = cam[3 * i + 2];
#pragma unroll
    for (int r = 0; r < 3; r++)
        world[3 * i + r] = (float)(rt.v[4 * r] * x + rt.v[4 * r + 1] * y + rt.v[4 * r + 2] * z + rt.v[4 * r + 3]);
}

int map_reserve(svo_map *m, size_t want)
{
    if (want <= m->cap_pts)
        return SVO_OK;
    size_t cap = m->cap_pts ? m->cap_pts : (size_t)1 << 16;
    while (cap < want)
        cap *= 2;
    float *cam = nullptr, *world = nullptr;
    int *id = nullptr;
    if (hipMalloc((void **)&cam, cap * 12) != hipSuccess || hipMalloc((void **)&world, cap * 12) != hipSuccess ||
        hipMalloc((void **)&id, cap * 4) != hipSuccess) {
        (void)hipFree(cam);
        (void)hipFree(world);
        (void)hipFree(id);
        svo_set_error("map: cannot grow to %zu points", cap);
        return SVO_ERR_HIP;
    }
    hipStream_t st = m->ctx->stream;
    if (m->n_pts) {
        SVO_HIP(hipMemcpyAsync(cam, m->d_cam, m->n_pts * 12, hipMemcpyDeviceToDevice, st));
        SVO_HIP(hipMemcpyAsync(world, m->d_world, m->n_pts * 12, hipMemcpyDeviceToDevice, st));
        SVO_HIP(hipMemcpyAsync(id, m->d_id, m->n_pts * 4, hipMemcpyDeviceToDevice, st));
    }
    SVO_HIP(hipStreamSynchronize(st));
    (void)hipFree(m->d_cam);
    (void)hipFree(m->d_world);
    (void)hipFree(m->d_id);
    m->d_cam = cam;
    m->d_world = world;
    m->d_id = id;
    m->cap_pts = cap;
    return SVO_OK;
}

int map_upload_rt(svo_map *m)
{
    const size_t k = m->rec.size();
    int rc = m->d_rt.ensure(k * 96 + 96);
    if (rc)
        return rc;
    std::vector<double> h(k * 12);
    for (size_t j = 0; j < k; j++)
        memcpy(&h[12 * j], m->rec[j].Rt, 96);
    SVO_HIP(hipMemcpyAsync(m->d_rt.p, h.data(), k * 96, hipMemcpyHostToDevice, m->ctx->stream));
    SVO_HIP(hipStreamSynchronize(m->ctx->stream));  // h leaves scope
    return SVO_OK;
}

}  // namespace

extern "C" {

int svo_map_create(svo_ctx *ctx, svo_map **out)
{
    SVO_CHECK_ARG(ctx && out);
    svo_map *m = new svo_map();
    m->ctx = ctx;
    *out = m;
    return SVO_OK;
}

int svo_map_destroy(svo_map *m)
{
    if (!m)
        return SVO_OK;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    (void)hipFree(m->d_cam);
    (void)hipFree(m->d_world);
    (void)hipFree(m->d_id);
    m->d_rt.release();
    delete m;
    return SVO_OK;
}

int svo_map_add_keyframe(svo_map *m, int traj_index, const double *R9, const double *t3, const float *xyz_cam, int n,
                         int retrack, int mem)
{
    SVO_CHECK_ARG(m && traj_index >= 0 && R9 && t3 && n >= 0 && (n == 0 || xyz_cam));
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    SVO_HIP(hipSetDevice(m->ctx->device));
    int rc = map_reserve(m, m->n_pts + (size_t)n);
    if (rc)
        return rc;
    svo_map::Rec r;
    r.traj_index = traj_index;
    r.retrack = retrack ? 1 : 0;
    r.n = n;
    r.off = m->n_pts;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            r.Rt[4 * i + j] = R9[3 * i + j];
        r.Rt[4 * i + 3] = t3[i];
    }
    const int id = (int)m->rec.size();
    m->rec.push_back(r);
    m->n_pts += (size_t)n;
    if (n == 0)
        return SVO_OK;
    hipStream_t st = m->ctx->stream;
    SVO_HIP(hipMemcpyAsync(m->d_cam + 3 * r.off, xyz_cam, (size_t)n * 12,
                           mem == SVO_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
    // the record's world cloud as insertKeyFrames places it (src/keyFrameManagement.cpp:20-30): ONE launch that
    // also writes the record id of its points; the record's [R|t] travels in the kernel arguments, so nothing
    // is uploaded and the stream is not waited for (svo_map_update uploads all records once per solve)
    MapRt rt;
    memcpy(rt.v, r.Rt, sizeof(rt.v));
    hipLaunchKernelGGL(map_transform_one_kernel, dim3((n + 255) / 256), dim3(256), 0, st, m->d_cam, rt, r.off,
                       (size_t)n, m->d_world, m->d_id, id);
    SVO_HIP(hipGetLastError());
    if (mem == SVO_MEM_HOST)
        SVO_HIP(hipStreamSynchronize(st));  // the caller's array may go away
    return SVO_OK;
}

int svo_map_num_keyframes(const svo_map *m) { return m ? (int)m->rec.size() : 0; }

int svo_map_update(svo_map *m, const double *t3s, int n_poses)
{
    SVO_CHECK_ARG(m && n_poses >= 0 && (n_poses == 0 || t3s));
    SVO_HIP(hipSetDevice(m->ctx->device));
    // trajectory[j] = translation(T_j); kf.t = trajectory[kf's index]; R keeps its un-optimised value
    // (src/optimizationStuff.cpp:19-32).  A record past the end of T keeps its translation (upstream
    // would read out of bounds there).
    for (svo_map::Rec &r : m->rec)
        if (r.traj_index < n_poses)
            for (int i = 0; i < 3; i++)
                r.Rt[4 * i + 3] = t3s[3 * (size_t)r.traj_index + i];
    if (m->n_pts == 0)
        return SVO_OK;
    int rc = map_upload_rt(m);
    if (rc)
        return rc;
    const size_t n = m->n_pts;
    hipLaunchKernelGGL(map_transform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->ctx->stream, m->d_cam,
                       m->d_id, m->d_rt.as<double>(), (size_t)0, n, m->d_world);
    SVO_HIP(hipGetLastError());
    return SVO_OK;
}

int svo_map_get_points(svo_map *m, float *xyz_world, size_t cap_points, int *counts, int cap_keyframes,
                       size_t *n_points, int *n_keyframes, int mem)
{
    SVO_CHECK_ARG(m && n_points && n_keyframes);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    size_t tot = 0;
    int nk = 0;
    for (const svo_map::Rec &r : m->rec)
        if (r.retrack) {  // mapHistory holds the records with retrack only (:43-45)
            tot += (size_t)r.n;
            nk++;
        }
    *n_points = tot;
    *n_keyframes = nk;
    if (!xyz_world && !counts)
        return SVO_OK;
    if ((xyz_world && cap_points < tot) || (counts && cap_keyframes < nk)) {
        svo_set_error("map: %zu points in %d keyframes exceed the capacity (%zu, %d)", tot, nk, cap_points, cap_keyframes);
        return SVO_ERR_CAPACITY;
    }
    SVO_HIP(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    size_t o = 0;
    int k = 0;
    // consecutive retrack records are copied as one run
    for (size_t j = 0; j < m->rec.size();) {
        if (!m->rec[j].retrack) {
            j++;
            continue;
        }
        size_t e = j, run = 0;
        while (e < m->rec.size() && m->rec[e].retrack) {
            if (counts)
                counts[k] = m->rec[e].n;
            k++;
            run += (size_t)m->rec[e].n;
            e++;
        }
        if (xyz_world && run)
            SVO_HIP(hipMemcpyAsync(xyz_world + 3 * o, m->d_world + 3 * m->rec[j].off, run * 12,
                                   mem == SVO_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
        o += run;
        j = e;
    }
    if (mem == SVO_MEM_HOST)
        SVO_HIP(hipStreamSynchronize(st));
    return SVO_OK;
}

}  // extern "C"
