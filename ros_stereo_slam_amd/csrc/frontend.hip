// frontend.hip -- the stereo-VO front-end frame loop on one GPU (svo_vo).
//
// Host-side mirror of visualSLAM::initSequence's per-frame body
// (src/VisualSLAM.cpp:11-169) with every data-parallel stage on the device:
//   svo_vo_init      stereoTriangulate of frame 0            VisualSLAM.cpp:22-41
//   svo_vo_localize  PerspectiveNpointEstimation + pose      VisualSLAM.cpp:64-74,
//                    composition                              keyFrameManagement.cpp:73-94
//   svo_vo_update    keyframe rule + reference hand-over     VisualSLAM.cpp:93-152,
//                                                             keyFrameManagement.cpp:9-31
// The split lets the caller run the pose graph between the two (VisualSLAM.cpp:76-89
// re-anchors t after a loop closure before the keyframe is inserted).
//
// Device residency: the reference image pyramid, the current pyramid, the reference point
// sets (2-D, 3-D world) and every intermediate live in HBM.  A localisation enqueues
// pyramid -> LK -> compaction -> F-RANSAC -> compaction -> PnP-RANSAC (+refine) on the
// context's stream with device-side counts chained between stages, and reads back ONE
// 144-byte record (pose, inlier count, tracked count) -- the values the reference's host
// policy branches on (inliers < 10: retry / shutdown; inliers < 200: keyframe).
#include <string>
#include <thread>
#include <vector>
#include <chrono>
#include <cmath>

#include "svo_internal.h"

struct PnpRecord {  // layout of pnp.hip's PnpResult + the tracked-point count behind it
    double rvec[3], tvec[3], R[9], rms;
    int n_inliers, iters_run;
    int n_tracked, pad;
};

// Pinned, fine-grained host memory the chunk runner's kernels publish into; the host spins on the
// tags instead of asking the runtime about events (no copy engine, no runtime lock on the way).
struct Mailbox {
    int early[4];     // tag, RANSAC inlier count, tracked points
    int pose_tag[2];  // per record slot
    int pad[2];
    PnpRecord rec[2];
};

struct svo_vo {
    svo_ctx *ctx = nullptr;
    Mailbox *mbox = nullptr;
    int tag = 0;
    svo_vo_params prm;
    int w = 0, h = 0, c = 0, cap = 0;
    svo_pyramid *pyr_ref = nullptr, *pyr_cur = nullptr, *pyr_right = nullptr, *pyr_next = nullptr;
    hipStream_t stream_b = nullptr;          // second stream: PnP of frame t beside pyramid + LK of frame t+1
    hipEvent_t ev_a = nullptr;               // "tracked sets of frame t are ready" (stream A -> B)
    float *sa2 = nullptr;                    // speculative LK output for frame t+1
    uint8_t *sstatus = nullptr;
    // point sets (device)
    float *ref2d = nullptr, *ref3d = nullptr, *trk2d = nullptr, *trk3d = nullptr;
    float *a2 = nullptr, *b2 = nullptr, *c2 = nullptr, *d2 = nullptr, *a3 = nullptr, *b3 = nullptr, *resp = nullptr;
    uint8_t *status = nullptr, *mask = nullptr, *st2 = nullptr;
    float *grid_xy = nullptr;  // the keypoint lattice of src/triangulation.cpp:89-96, written once at creation
    int *idx = nullptr, *d_cnt = nullptr;  // d_cnt[0..7]: stage counts
    PnpRecord *d_rec = nullptr;
    uint8_t *d_img = nullptr;  // staging for host images
    int nref = 0, ntrk = 0, frame = 0;
    int kf_n = 0;  // points of the last keyframe's camera-frame cloud in b3
    int ladder_ransac_inliers = 0;
    double R[9], t[3];
    bool has_cur = false;
};

namespace {

template <class T> int dev_alloc(T **p, size_t count)
{
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    if (e != hipSuccess) {
        svo_set_error("hipMalloc(%zu) -> %s", count * sizeof(T), hipGetErrorString(e));
        return SVO_ERR_HIP;
    }
    return SVO_OK;
}

// Spin on a tag a kernel releases into the pinned mailbox.  Now and then the stream is asked whether
// it is still alive (an error, or a stream that drained without publishing, ends the wait), and a
// wall-clock bound ends a wait on a stream that hangs without reporting an error.
constexpr double SVO_MAILBOX_TIMEOUT_S = 60.0;
int wait_mailbox_tag(const int *slot, int tag, hipStream_t stream)
{
    std::chrono::steady_clock::time_point t0;
    bool timing = false;
    for (unsigned spins = 1;; spins++) {
        if (__atomic_load_n(slot, __ATOMIC_ACQUIRE) == tag)
            return SVO_OK;
        if ((spins & 0x3FFFF) == 0) {
            hipError_t e = hipStreamQuery(stream);
            if (e == hipSuccess && __atomic_load_n(slot, __ATOMIC_ACQUIRE) != tag)
                e = hipErrorUnknown;  // drained without publishing
            if (e != hipSuccess && e != hipErrorNotReady) {
                svo_set_error("waiting for the PnP mailbox -> %s", hipGetErrorString(e));
                return SVO_ERR_HIP;
            }
            const auto now = std::chrono::steady_clock::now();
            if (!timing) {
                t0 = now;
                timing = true;
            } else if (std::chrono::duration<double>(now - t0).count() > SVO_MAILBOX_TIMEOUT_S) {
                svo_set_error("waiting for the PnP mailbox: no tag after %.0f s, the stream hangs", SVO_MAILBOX_TIMEOUT_S);
                return SVO_ERR_HIP;
            }
        }
    }
}

int grid_axis(int dim, int step)
{
    int k = 0;
    for (int v = step; v < dim - step; v += step)
        k++;
    return k;
}

uint64_t stage_seed(const svo_vo *v, int stage) { return v->prm.seed + 8ull * (uint64_t)v->frame + stage; }

__global__ void store_count_kernel(const int *__restrict__ src, int *__restrict__ dst) { *dst = *src; }

const uint8_t *stage_image(svo_vo *v, const uint8_t *img, int mem, int *rc)
{
    *rc = SVO_OK;
    if (mem == SVO_MEM_DEVICE)
        return img;
    hipError_t e =
        hipMemcpyAsync(v->d_img, img, (size_t)v->w * v->h * v->c, hipMemcpyHostToDevice, v->ctx->stream);
    if (e != hipSuccess) {
        svo_set_error("image upload -> %s", hipGetErrorString(e));
        *rc = SVO_ERR_HIP;
    }
    return v->d_img;
}

// visualSLAM::stereoTriangulate, dense branch (src/triangulation.cpp:87-103,137-165), with
// the optional ANMS stage, for k front-ends of one context at once (same image size, grid step
// and ANMS budget): every stage is one set of launches.  Per front-end: x1 -> out2d, camera-frame
// points -> v->b3 and, when Rt is given, world points -> out3d (else the camera-frame points).
// Count -> d_cnt[4] and host.
int stereo_triangulate_batch(int k, svo_vo *const *vs, svo_pyramid *const *lefts, svo_pyramid *const *rights,
                             const double *const *Rts, float *const *out2d, float *const *out3d, int *const *n_out)
{
    svo_vo *v0 = vs[0];
    svo_ctx *ctx = v0->ctx;
    int rc;
    const int n = grid_axis(v0->w, v0->prm.grid_step) * grid_axis(v0->h, v0->prm.grid_step);
    LkJob lk[SVO_LK_MAX_JOBS];
    for (int a = 0; a < k; a++) {
        svo_vo *v = vs[a];
        // denseLKtracking: LK left -> right (src/tracking.cpp:18); min-eig is the ANMS response
        LkJob &q = lk[a];
        q.prev = lefts[a]->dev;
        q.next = rights[a]->dev;
        q.dprev = lefts[a]->dbase;
        q.prev_pts = v->grid_xy;
        q.n_cap = n;
        q.d_n = nullptr;
        q.next_pts = v->b2;
        q.status = v->status;
        q.err = nullptr;
        q.min_eig = v->resp;
    }
    if ((rc = svo_launch_lk_batch(ctx, k, lk, lefts[0])))
        return rc;
    const float *pts[SVO_LK_MAX_JOBS], *trk[SVO_LK_MAX_JOBS];
    const uint8_t *stt[SVO_LK_MAX_JOBS];
    const int *d_n[SVO_LK_MAX_JOBS];
    for (int a = 0; a < k; a++) {
        pts[a] = vs[a]->grid_xy;
        trk[a] = vs[a]->b2;
        stt[a] = vs[a]->status;
        d_n[a] = nullptr;
    }
    if (v0->prm.anms_keep > 0) {
        const float *xy[SVO_LK_MAX_JOBS], *resp[SVO_LK_MAX_JOBS];
        int *oidx[SVO_LK_MAX_JOBS], *ocnt[SVO_LK_MAX_JOBS];
        svo_anms_gather ga[SVO_LK_MAX_JOBS];
        for (int a = 0; a < k; a++) {
            svo_vo *v = vs[a];
            xy[a] = v->grid_xy;
            resp[a] = v->resp;
            oidx[a] = v->idx;
            ocnt[a] = v->d_cnt + 2;
            ga[a] = {v->grid_xy, v->b2, v->c2, v->d2, v->status, v->st2};
        }
        // the kept keypoints' lattice points, tracked points and status bytes come out of the same launch
        // that lists them (a gather launch of its own before)
        if ((rc = svo_launch_anms_batch(ctx, k, xy, resp, n, v0->prm.anms_keep, oidx, ocnt, ga)))
            return rc;
        for (int a = 0; a < k; a++) {
            pts[a] = vs[a]->c2;
            trk[a] = vs[a]->d2;
            stt[a] = vs[a]->st2;
            d_n[a] = vs[a]->d_cnt + 2;
        }
    }
    // status compaction (src/tracking.cpp:20-27); ping-pong between the (a2,b2) and (c2,d2) pairs
    svo_compact_job c1[SVO_LK_MAX_JOBS], c2[SVO_LK_MAX_JOBS];
    svo_fransac_job fj[SVO_LK_MAX_JOBS];
    svo_tri_job tj[SVO_LK_MAX_JOBS];
    for (int a = 0; a < k; a++) {
        svo_vo *v = vs[a];
        float *o1 = pts[a] == v->grid_xy ? v->c2 : v->a2, *o2 = pts[a] == v->grid_xy ? v->d2 : v->b2;
        float *x1 = out2d[a], *x2 = o1 == v->a2 ? v->c2 : v->a2;
        c1[a] = {stt[a], n, d_n[a], {pts[a], trk[a], nullptr}, {o1, o2, nullptr}, {2, 2, 0}, v->d_cnt + 3};
        // FmatThresholding (src/tracking.cpp:30-43): 3 px, 0.99
        c2[a] = {v->mask, n, v->d_cnt + 3, {o1, o2, nullptr}, {x1, x2, nullptr}, {2, 2, 0}, v->d_cnt + 4};
        fj[a] = {o1, o2, n, v->d_cnt + 3, v->prm.f_thr_stereo, 0.99, 1000, stage_seed(v, 3), v->mask, nullptr, nullptr,
                 nullptr, &c2[a]};  // the mask compaction rides with the F-RANSAC
        tj[a] = {x1, x2, n, v->d_cnt + 4, Rts[a] ? v->b3 : out3d[a], nullptr, Rts[a], Rts[a] ? out3d[a] : nullptr,
                 reinterpret_cast<int *>(ctx->pinned) + a};  // the count the host reads after the wait below
    }
    double P1[12], P2[12];
    svo_stereo_projections(v0->prm.fx, v0->prm.fy, v0->prm.cx, v0->prm.cy, v0->prm.baseline, P1, P2);
    if ((rc = svo_launch_compact_batch(ctx, k, c1)) || (rc = svo_launch_fransac_batch(ctx, k, fj)) ||
        (rc = svo_launch_triangulate_batch(ctx, P1, P2, k, tj)))
        return rc;
    int *pin = reinterpret_cast<int *>(ctx->pinned);
    if ((rc = svo_wait(ctx)))
        return rc;
    for (int a = 0; a < k; a++) {
        *n_out[a] = pin[a];
        vs[a]->kf_n = Rts[a] ? pin[a] : 0;  // b3 holds the keyframe's camera-frame cloud
    }
    return SVO_OK;
}

// the PnP-RANSAC problem of a localisation as the chunk runners queue it: solvePnPRansac(100, thr, conf) with
// the inlier count published early (mailbox `early`) and the finished record into mailbox slot `slot`
svo_pnp_job pnp_job(svo_vo *v, int cap, const int *cnt_trk, double thr, double conf, uint64_t seed, PnpRecord *d_rec,
                    int early_tag, int slot, int pose_tag)
{
    svo_pnp_job q;
    q.obj = v->trk3d;
    q.img = v->trk2d;
    q.cap = cap;
    q.d_n = cnt_trk;
    q.K4[0] = v->prm.fx;
    q.K4[1] = v->prm.fy;
    q.K4[2] = v->prm.cx;
    q.K4[3] = v->prm.cy;
    q.iterations = 100;
    q.reproj_err = thr;
    q.confidence = conf;
    q.seed = seed;
    q.refine_iters = 20;
    q.inliers = v->idx;
    q.mask = nullptr;
    q.d_result = d_rec;
    q.early_mbox = v->mbox->early;
    q.early_tag = early_tag;
    q.h_rec = &v->mbox->rec[slot];
    q.h_tag = &v->mbox->pose_tag[slot];
    q.tag = pose_tag;
    q.cnt_trk = cnt_trk;
    return q;
}

// The pose ladder of the older visualOdometry::initSequence, src/bundleAdjust.cpp:462-480, on device point
// sets: (obj_f, img_f, cnt_f) = the tracked set after the F-matrix filter, (obj_s, img_s, cnt_s) = the
// status-filtered set ("retracking" without the filter gives exactly that: LK is deterministic).
//   rung 0  solvePnPRansac(100, 4.0, 0.99) on the filtered set
//   rung 1  < 20 inliers or tvec.x > 1000: the same on the status-filtered set
//   rung 2  < 10 inliers (or tvec.x > 1000 after rung 1): plain solvePnP on the set last used
// The record of the deciding solve is left in the context's pinned block (n_tracked = size of the set
// used; n_inliers = 0 when solvePnP had no solution); *ransac_inliers = the last RANSAC's count.
int ladder_pose(svo_ctx *ctx, const float *obj_f, const float *img_f, const int *cnt_f, const float *obj_s,
                const float *img_s, const int *cnt_s, int cap, const double *K4, uint64_t seed1, uint64_t seed2,
                int *idx_scratch, PnpRecord *d_rec, int *rung, int *ransac_inliers = nullptr)
{
    int rc;
    const PnpRecord *rec = reinterpret_cast<const PnpRecord *>(ctx->pinned);
    auto fetch = [&](const int *cnt) -> int {
        hipLaunchKernelGGL(store_count_kernel, dim3(1), dim3(1), 0, ctx->stream, cnt, &d_rec->n_tracked);
        SVO_HIP(hipMemcpyAsync(ctx->pinned, d_rec, sizeof(PnpRecord), hipMemcpyDeviceToHost, ctx->stream));
        return svo_wait(ctx);
    };
    const float *o3 = obj_f, *o2 = img_f;
    const int *cnt = cnt_f;
    *rung = 0;
    if ((rc = svo_launch_pnp_ransac(ctx, o3, o2, cap, cnt, K4, 100, 4.0, 0.99, seed1, 20, idx_scratch, nullptr, d_rec)) ||
        (rc = fetch(cnt)))
        return rc;
    bool plain = false;
    if (rec->n_inliers < 20 || rec->tvec[0] > 1000) {
        *rung = 1;
        o3 = obj_s;
        o2 = img_s;
        cnt = cnt_s;
        if ((rc = svo_launch_pnp_ransac(ctx, o3, o2, cap, cnt, K4, 100, 4.0, 0.99, seed2, 20, idx_scratch, nullptr,
                                        d_rec)) ||
            (rc = fetch(cnt)))
            return rc;
        plain = rec->n_inliers < 10 || rec->tvec[0] > 1000;
    }
    const int ninl = rec->n_inliers;
    if (ransac_inliers)
        *ransac_inliers = ninl;
    if (plain || ninl < 10) {  // "Skipping RANSAC all together": cv::solvePnP on everything that was tracked
        *rung = 2;
        if ((rc = svo_launch_solve_pnp(ctx, o3, o2, cap, cnt, K4, 20, idx_scratch, d_rec)) || (rc = fetch(cnt)))
            return rc;
    }
    return SVO_OK;
}

int stereo_triangulate(svo_vo *v, const svo_pyramid *left, const svo_pyramid *right, const double *Rt,
                       float *out2d, float *out3d, int *n_out)
{
    svo_pyramid *l = const_cast<svo_pyramid *>(left), *r = const_cast<svo_pyramid *>(right);
    return stereo_triangulate_batch(1, &v, &l, &r, &Rt, &out2d, &out3d, &n_out);
}

}  // namespace

extern "C" {

void svo_vo_default_params(svo_vo_params *p)
{
    if (!p)
        return;
    p->fx = 7.188560000000e+02;  // include/visualSLAM.h:82-87
    p->fy = 7.188560000000e+02;
    p->cx = 6.071928000000e+02;
    p->cy = 1.852157000000e+02;
    p->baseline = 0.54;  // include/visualSLAM.h:68
    p->grid_step = 30;   // src/triangulation.cpp:89
    p->anms_keep = 0;
    p->keyframe_min_inliers = 200;  // src/VisualSLAM.cpp:120
    p->f_thr_stereo = 3.0;          // src/tracking.cpp:34
    p->f_thr_temporal = 1.0;        // src/tracking.cpp:75
    p->seed = 0;
    p->policy = SVO_POLICY_SLAM;
}

int svo_vo_create(svo_ctx *ctx, const svo_vo_params *params, int width, int height, int channels, svo_vo **out)
{
    SVO_CHECK_ARG(ctx && params && out);
    SVO_CHECK_ARG(channels == 1 || channels == 3);
    SVO_CHECK_ARG(params->grid_step > 0 && params->keyframe_min_inliers >= 0);
    SVO_CHECK_ARG(params->policy == SVO_POLICY_SLAM || params->policy == SVO_POLICY_VO_LADDER);
    *out = nullptr;
    SVO_HIP(hipSetDevice(ctx->device));
    svo_vo *v = new svo_vo();
    v->ctx = ctx;
    v->prm = *params;
    v->w = width;
    v->h = height;
    v->c = channels;
    v->cap = grid_axis(width, params->grid_step) * grid_axis(height, params->grid_step);
    if (v->cap < 16)
        v->cap = 16;
    int rc = SVO_OK;
    const size_t n = (size_t)v->cap;
    if ((rc = svo_pyramid_create(ctx, width, height, channels, SVO_MAX_LEVELS, &v->pyr_ref)) ||
        (rc = svo_pyramid_create(ctx, width, height, channels, SVO_MAX_LEVELS, &v->pyr_cur)) ||
        // the right image is only ever the SECOND image of a tracking pass: no derivative levels
        (rc = svo_pyramid_create_ex(ctx, width, height, channels, SVO_MAX_LEVELS, false, &v->pyr_right)) ||
        (rc = svo_pyramid_create(ctx, width, height, channels, SVO_MAX_LEVELS, &v->pyr_next)) ||
        (rc = dev_alloc(&v->sa2, n * 2)) || (rc = dev_alloc(&v->sstatus, n)) ||
        (rc = dev_alloc(&v->ref2d, n * 2)) || (rc = dev_alloc(&v->ref3d, n * 3)) ||
        (rc = dev_alloc(&v->trk2d, n * 2)) || (rc = dev_alloc(&v->trk3d, n * 3)) ||
        (rc = dev_alloc(&v->a2, n * 2)) || (rc = dev_alloc(&v->b2, n * 2)) || (rc = dev_alloc(&v->c2, n * 2)) ||
        (rc = dev_alloc(&v->d2, n * 2)) || (rc = dev_alloc(&v->a3, n * 3)) || (rc = dev_alloc(&v->b3, n * 3)) ||
        (rc = dev_alloc(&v->resp, n)) || (rc = dev_alloc(&v->status, n)) || (rc = dev_alloc(&v->mask, n)) ||
        (rc = dev_alloc(&v->st2, n)) || (rc = dev_alloc(&v->idx, n)) || (rc = dev_alloc(&v->d_cnt, 16)) ||
        (rc = dev_alloc(&v->d_rec, 2)) || (rc = dev_alloc(&v->d_img, (size_t)width * height * channels)) ||
        (rc = dev_alloc(&v->grid_xy, n * 2)) ||
        (rc = svo_launch_grid(ctx, height, width, params->grid_step, v->grid_xy, (int)n))) {
        svo_vo_destroy(v);
        return rc;
    }
    if (hipEventCreateWithFlags(&v->ev_a, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&v->mbox), sizeof(Mailbox),
                      hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        svo_set_error("front-end: cannot create the second stream / event");
        svo_vo_destroy(v);
        return SVO_ERR_HIP;
    }
    memset(v->mbox, 0, sizeof(Mailbox));  // a recycled pinned block may hold a destroyed front-end's tags
    for (int i = 0; i < 9; i++)
        v->R[i] = (i % 4) == 0;
    v->t[0] = v->t[1] = v->t[2] = 0;
    *out = v;
    return SVO_OK;
}

int svo_vo_destroy(svo_vo *v)
{
    if (!v)
        return SVO_OK;
    (void)hipSetDevice(v->ctx->device);
    (void)hipStreamSynchronize(v->ctx->stream);
    svo_pyramid_destroy(v->ctx, v->pyr_ref);
    svo_pyramid_destroy(v->ctx, v->pyr_cur);
    svo_pyramid_destroy(v->ctx, v->pyr_right);
    svo_pyramid_destroy(v->ctx, v->pyr_next);
    if (v->stream_b) {
        (void)hipStreamSynchronize(v->stream_b);
        (void)hipStreamDestroy(v->stream_b);
    }
    if (v->ev_a)
        (void)hipEventDestroy(v->ev_a);
    if (v->mbox)
        (void)hipHostFree(v->mbox);
    if (v->sa2)
        (void)hipFree(v->sa2);
    if (v->sstatus)
        (void)hipFree(v->sstatus);
    void *bufs[] = {v->ref2d, v->ref3d, v->trk2d, v->trk3d, v->a2,   v->b2,    v->c2,    v->d2,   v->a3,
                    v->b3,    v->resp,  v->status, v->mask, v->st2, v->idx,   v->d_cnt, v->d_rec, v->d_img,
                    v->grid_xy};
    for (void *b : bufs)
        if (b)
            (void)hipFree(b);
    delete v;
    return SVO_OK;
}

int svo_vo_init(svo_vo *v, const uint8_t *left, const uint8_t *right, int mem, int *n_points)
{
    SVO_CHECK_ARG(v && left && right);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    int rc;
    v->frame = 0;
    for (int i = 0; i < 9; i++)
        v->R[i] = (i % 4) == 0;
    v->t[0] = v->t[1] = v->t[2] = 0;
    const uint8_t *d = stage_image(v, left, mem, &rc);
    if (rc || (rc = svo_build_pyramid_from_device(v->ctx, v->pyr_ref, d)))
        return rc;
    d = stage_image(v, right, mem, &rc);
    if (rc || (rc = svo_build_pyramid_from_device(v->ctx, v->pyr_right, d)))
        return rc;
    // identity [R|t]: the world cloud equals the camera-frame one bit for bit (x * 1 + y * 0 + z * 0 + 0 in
    // double), and the camera-frame cloud lands in b3 as at every later keyframe (svo_vo_get_keyframe_cloud)
    static const double I34[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    if ((rc = stereo_triangulate(v, v->pyr_ref, v->pyr_right, I34, v->ref2d, v->ref3d, &v->nref)))
        return rc;
    v->has_cur = false;
    if (n_points)
        *n_points = v->nref;
    return SVO_OK;
}

int svo_vo_localize(svo_vo *v, const uint8_t *left, int mem, double *R9, double *t3, int *n_inliers,
                    int *n_tracked)
{
    SVO_CHECK_ARG(v && left && R9 && t3);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    svo_ctx *ctx = v->ctx;
    int rc;
    v->frame++;
    const int n = v->nref;
    if (n_inliers)
        *n_inliers = 0;
    if (n_tracked)
        *n_tracked = 0;
    const uint8_t *d = stage_image(v, left, mem, &rc);
    if (rc || (rc = svo_build_pyramid_from_device(ctx, v->pyr_cur, d)))
        return rc;
    v->has_cur = true;
    if (n < 5) {
        svo_set_error("tracking lost: %d reference points", n);
        return SVO_ERR_TRACKING_LOST;
    }
    // PyrLKtrackFrame2Frame (src/tracking.cpp:46-91)
    if ((rc = svo_launch_lk(ctx, v->pyr_ref, v->pyr_cur, v->ref2d, n, v->a2, v->status, nullptr, nullptr)))
        return rc;
    if ((rc = svo_launch_compact(ctx, v->status, n, nullptr, v->ref2d, 2, v->b2, v->a2, 2, v->c2, v->ref3d, 3, v->a3,
                                 v->d_cnt)))
        return rc;
    {
        // the F-RANSAC's finishing wave also compacts by its mask (src/tracking.cpp:77-88)
        const svo_compact_job by_mask = {v->mask, n, v->d_cnt, {v->c2, v->a3, nullptr}, {v->trk2d, v->trk3d, nullptr},
                                         {2, 3, 0}, v->d_cnt + 1};
        if ((rc = svo_launch_fransac(ctx, v->b2, v->c2, n, v->d_cnt, v->prm.f_thr_temporal, 0.99, 1000, stage_seed(v, 0),
                                     v->mask, nullptr, nullptr, nullptr, &by_mask)))
            return rc;
    }
    const double K4[4] = {v->prm.fx, v->prm.fy, v->prm.cx, v->prm.cy};
    const PnpRecord *rec = reinterpret_cast<const PnpRecord *>(ctx->pinned);
    if (v->prm.policy == SVO_POLICY_VO_LADDER) {
        // visualOdometry::initSequence, src/bundleAdjust.cpp:452-480 (see svo_vo_params.policy)
        int rung = 0;
        if ((rc = ladder_pose(ctx, v->trk3d, v->trk2d, v->d_cnt + 1, v->a3, v->c2, v->d_cnt, n, K4, stage_seed(v, 1),
                              stage_seed(v, 2), v->idx, v->d_rec, &rung, &v->ladder_ransac_inliers)))
            return rc;
        if (rung >= 1) {  // the set the pose was computed from is the tracked set
            std::swap(v->trk2d, v->c2);
            std::swap(v->trk3d, v->a3);
        }
        v->ntrk = rec->n_tracked;
        if (n_inliers)
            *n_inliers = rung == 2 ? v->ladder_ransac_inliers : rec->n_inliers;
        if (n_tracked)
            *n_tracked = rec->n_tracked;
        if (rec->n_inliers == 0) {  // upstream: cv::Exception out of solvePnP (too few / planar points)
            svo_set_error("tracking lost at frame %d: solvePnP has no solution for %d points", v->frame, rec->n_tracked);
            return SVO_ERR_TRACKING_LOST;
        }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                R9[3 * i + j] = rec->R[3 * j + i];
        for (int i = 0; i < 3; i++)
            t3[i] = -(R9[3 * i] * rec->tvec[0] + R9[3 * i + 1] * rec->tvec[1] + R9[3 * i + 2] * rec->tvec[2]);
        return SVO_OK;
    }
    // solvePnPRansac (src/keyFrameManagement.cpp:84), retry (:85-92)
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((rc = svo_launch_pnp_ransac(ctx, v->trk3d, v->trk2d, n, v->d_cnt + 1, K4, 100, attempt ? 8.0 : 1.0,
                                        attempt ? 0.98 : 0.99, stage_seed(v, attempt ? 2 : 1), 20, v->idx, nullptr,
                                        v->d_rec)))
            return rc;
        hipLaunchKernelGGL(store_count_kernel, dim3(1), dim3(1), 0, ctx->stream, v->d_cnt + 1, &v->d_rec->n_tracked);
        SVO_HIP(hipMemcpyAsync(ctx->pinned, v->d_rec, sizeof(PnpRecord), hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = svo_wait(ctx)))
            return rc;
        if (rec->n_inliers >= 10)
            break;
    }
    v->ntrk = rec->n_tracked;
    if (n_inliers)
        *n_inliers = rec->n_inliers;
    if (n_tracked)
        *n_tracked = rec->n_tracked;
    if (rec->n_inliers < 10) {
        svo_set_error("tracking lost at frame %d: %d PnP inliers", v->frame, rec->n_inliers);
        return SVO_ERR_TRACKING_LOST;
    }
    // Rodrigues; R = R^T; t = -R * tvec (src/VisualSLAM.cpp:70-74)
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R9[3 * i + j] = rec->R[3 * j + i];
    for (int i = 0; i < 3; i++)
        t3[i] = -(R9[3 * i] * rec->tvec[0] + R9[3 * i + 1] * rec->tvec[1] + R9[3 * i + 2] * rec->tvec[2]);
    return SVO_OK;
}

int svo_vo_update(svo_vo *v, const uint8_t *right, int mem, const double *R9, const double *t3, int n_inliers,
                  int force_keyframe, int *was_keyframe)
{
    SVO_CHECK_ARG(v && R9 && t3);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    if (!v->has_cur) {
        svo_set_error("svo_vo_update without a preceding svo_vo_localize");
        return SVO_ERR_STATE;
    }
    memcpy(v->R, R9, sizeof(v->R));
    memcpy(v->t, t3, sizeof(v->t));
    // src/VisualSLAM.cpp:120; the older ladder re-triangulates on every frame (src/bundleAdjust.cpp:517-519)
    const bool kf = n_inliers < v->prm.keyframe_min_inliers || force_keyframe || v->prm.policy == SVO_POLICY_VO_LADDER;
    if (kf) {
        SVO_CHECK_ARG(right != nullptr);
        int rc;
        const uint8_t *d = stage_image(v, right, mem, &rc);
        if (rc || (rc = svo_build_pyramid_from_device(v->ctx, v->pyr_right, d)))
            return rc;
        double Rt[12];
        for (int i = 0; i < 3; i++) {
            Rt[4 * i] = R9[3 * i];
            Rt[4 * i + 1] = R9[3 * i + 1];
            Rt[4 * i + 2] = R9[3 * i + 2];
            Rt[4 * i + 3] = t3[i];
        }
        // insertKeyFrames (src/keyFrameManagement.cpp:9-31): re-triangulate at the current frame
        if ((rc = stereo_triangulate(v, v->pyr_cur, v->pyr_right, Rt, v->ref2d, v->ref3d, &v->nref)))
            return rc;
    } else {  // src/VisualSLAM.cpp:143-146: carry the tracked sets forward
        std::swap(v->ref2d, v->trk2d);
        std::swap(v->ref3d, v->trk3d);
        v->nref = v->ntrk;
    }
    std::swap(v->pyr_ref, v->pyr_cur);  // referenceImg = currentImage (src/VisualSLAM.cpp:151)
    v->has_cur = false;
    if (was_keyframe)
        *was_keyframe = kf ? 1 : 0;
    return SVO_OK;
}

int svo_vo_track(svo_vo *v, const uint8_t *left, const uint8_t *right, int mem, int force_keyframe, double *R9,
                 double *t3, int *n_inliers, int *was_keyframe, int *n_tracked)
{
    int ninl = 0;
    int rc = svo_vo_localize(v, left, mem, R9, t3, &ninl, n_tracked);
    if (n_inliers)
        *n_inliers = ninl;
    if (rc)
        return rc;
    return svo_vo_update(v, right, mem, R9, t3, ninl, force_keyframe, was_keyframe);
}

// A run of consecutive frames through the front-end without returning to the caller between
// frames (the "chunk runner" of SURVEY.md 8b/8e): exactly the result of n_frames calls of
// svo_vo_track(..., force_keyframe = 0), frame by frame.
//
// With pipeline != 0 the loop overlaps work on two HIP streams.  Stream A carries a frame's
// pyramid, LK, filters and the keyframe path; the PnP-RANSAC of frame t (a handful of
// wavefronts of f64 latency) runs on stream B while stream A already builds the pyramid of
// frame t+1 and tracks into it from the points frame t kept -- which is what frame t+1 will do
// unless frame t turns out to be a keyframe (fewer PnP inliers than the threshold).  In that
// case the speculative tracking is discarded and redone from the new keyframe's points; the
// pyramid is kept.  Scheduling only: every stage sees the same inputs as in the serial order,
// so results are identical (tests/test_gpu_frontend.py::test_run_chunk_*).
int svo_vo_run_chunk(svo_vo *v, const uint8_t *const *lefts, const uint8_t *const *rights, int n_frames, int mem,
                     int pipeline, double *R_out, double *t_out, int *inliers_out, int *tracked_out,
                     uint8_t *keyframe_out, int *n_done)
{
    SVO_CHECK_ARG(v && lefts && rights && n_frames >= 0 && R_out && t_out);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    if (v->prm.policy != SVO_POLICY_SLAM) {
        svo_set_error("the chunk runner drives the live policy (SVO_POLICY_SLAM) only");
        return SVO_ERR_ARG;
    }
    svo_ctx *ctx = v->ctx;
    if (pipeline && mem == SVO_MEM_DEVICE && !v->stream_b) {
        // The PnP stream exists only once pipelining is asked for, and in the high-priority class:
        // HIP keeps a separate pool of hardware queues per priority class, so the two streams of
        // a chunk never land on one queue (with both in the default class the runtime was seen to
        // put them on the same queue, which serialises the overlap away), and creating it does not
        // disturb the stream -> queue assignment of serial chunks running side by side.
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        SVO_HIP(hipStreamCreateWithPriority(&v->stream_b, hipStreamNonBlocking, prio_hi));
    }
    hipStream_t sA = ctx->stream, sB = v->stream_b;
    if (n_done)
        *n_done = 0;
    if (mem == SVO_MEM_HOST)
        pipeline = 0;  // host images go through one staging buffer; keep them strictly in order
    Mailbox *mb = v->mbox;
    int early_tag = 0, pose_tag[2] = {0, 0};
    bool spec = false, next_built = false;
    int pending = -1;  // frame whose refined pose has not been collected yet
    int rc;
    // spin on a tag a kernel releases into the mailbox; now and then make sure the stream is alive
    auto wait_tag = [&](const int *slot, int tag, hipStream_t stream) -> int { return wait_mailbox_tag(slot, tag, stream); };
    auto harvest = [&](int f) {  // record of frame f is on the host: pose composition (src/VisualSLAM.cpp:70-74)
        const PnpRecord *rec = &mb->rec[f & 1];
        double *R9 = R_out + 9 * (size_t)f, *t3 = t_out + 3 * (size_t)f;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                R9[3 * i + j] = rec->R[3 * j + i];
        for (int i = 0; i < 3; i++)
            t3[i] = -(R9[3 * i] * rec->tvec[0] + R9[3 * i + 1] * rec->tvec[1] + R9[3 * i + 2] * rec->tvec[2]);
        memcpy(v->R, R9, sizeof(v->R));
        memcpy(v->t, t3, sizeof(v->t));
    };
    for (int f = 0; f < n_frames; f++) {
        v->frame++;
        const int n = v->nref;
        if (n < 5) {
            if (pending >= 0 && wait_tag(&mb->pose_tag[pending & 1], pose_tag[pending & 1], pipeline ? sB : sA) == SVO_OK)
                harvest(pending);
            svo_set_error("tracking lost: %d reference points", n);
            return SVO_ERR_TRACKING_LOST;
        }
        if (spec) {  // frame f was tracked speculatively during frame f-1's PnP
            std::swap(v->pyr_cur, v->pyr_next);
            std::swap(v->a2, v->sa2);
            std::swap(v->status, v->sstatus);
        } else {
            if (next_built) {
                std::swap(v->pyr_cur, v->pyr_next);
            } else {
                const uint8_t *d = stage_image(v, lefts[f], mem, &rc);
                if (rc || (rc = svo_build_pyramid_from_device(ctx, v->pyr_cur, d)))
                    return rc;
            }
            if ((rc = svo_launch_lk(ctx, v->pyr_ref, v->pyr_cur, v->ref2d, n, v->a2, v->status, nullptr,
                                    nullptr)))
                return rc;
        }
        spec = next_built = false;
        // the tracked-point count alternates between two device slots: the refinement of frame
        // f-1 (stream B) may still read its slot while this frame's filters (stream A) write theirs
        int *cnt_trk = v->d_cnt + ((f & 1) ? 9 : 1);
        PnpRecord *d_rec = v->d_rec + (f & 1);
        const svo_compact_job by_mask = {v->mask, n, v->d_cnt, {v->c2, v->a3, nullptr}, {v->trk2d, v->trk3d, nullptr},
                                         {2, 3, 0}, cnt_trk};  // done by the F-RANSAC's finishing wave
        if ((rc = svo_launch_compact(ctx, v->status, n, nullptr, v->ref2d, 2, v->b2, v->a2, 2, v->c2, v->ref3d, 3,
                                     v->a3, v->d_cnt)) ||
            (rc = svo_launch_fransac(ctx, v->b2, v->c2, n, v->d_cnt, v->prm.f_thr_temporal, 0.99, 1000,
                                     stage_seed(v, 0), v->mask, nullptr, nullptr, nullptr, &by_mask)))
            return rc;
        // ---- PnP of this frame: stream B when pipelining ----
        hipStream_t sP = pipeline ? sB : sA;
        if (pipeline) {
            SVO_HIP(hipEventRecord(v->ev_a, sA));
            SVO_HIP(hipStreamWaitEvent(sB, v->ev_a, 0));
        }
        auto launch_pnp = [&](double thr, double conf, int stage) -> int {
            early_tag = ++v->tag;
            pose_tag[f & 1] = ++v->tag;
            ctx->stream = sP;
            svo_pnp_job q = pnp_job(v, n, cnt_trk, thr, conf, stage_seed(v, stage), d_rec, early_tag, f & 1, pose_tag[f & 1]);
            int r = svo_launch_pnp_ransac_batch(ctx, 1, &q);  // publishes the record itself
            ctx->stream = sA;
            return r;
        };
        if ((rc = launch_pnp(1.0, 0.99, 1)))
            return rc;
        // ---- speculation for the next frame on stream A ----
        bool speculated = false;
        if (pipeline && f + 1 < n_frames) {
            const uint8_t *d = stage_image(v, lefts[f + 1], mem, &rc);
            if (rc || (rc = svo_build_pyramid_from_device(ctx, v->pyr_next, d)))
                return rc;
            // next frame's reference = this frame's tracked set (its count lives in cnt_trk)
            if ((rc = svo_launch_lk(ctx, v->pyr_cur, v->pyr_next, v->trk2d, n, v->sa2, v->sstatus, nullptr,
                                    nullptr, cnt_trk)))
                return rc;
            speculated = true;
        }
        // ---- the policy needs only the RANSAC inlier count: known before mask / refinement end ----
        if ((rc = wait_tag(&mb->early[0], early_tag, sP)))
            return rc;
        if (pending >= 0) {  // stream B runs in order: frame f-1's record landed before this count
            if ((rc = wait_tag(&mb->pose_tag[pending & 1], pose_tag[pending & 1], sP)))
                return rc;
            harvest(pending);
            pending = -1;
        }
        int ninl = mb->early[1];
        if (ninl < 10) {  // retry at 8 px / 0.98 (src/keyFrameManagement.cpp:85-92)
            if ((rc = wait_tag(&mb->pose_tag[f & 1], pose_tag[f & 1], sP)) || (rc = launch_pnp(8.0, 0.98, 2)) ||
                (rc = wait_tag(&mb->early[0], early_tag, sP)))
                return rc;
            ninl = mb->early[1];
        }
        v->ntrk = mb->early[2];
        if (inliers_out)
            inliers_out[f] = ninl;
        if (tracked_out)
            tracked_out[f] = v->ntrk;
        if (ninl < 10) {
            (void)hipStreamSynchronize(sP);
            if (speculated)
                (void)hipStreamSynchronize(sA);
            svo_set_error("tracking lost at frame %d: %d PnP inliers", v->frame, ninl);
            return SVO_ERR_TRACKING_LOST;
        }
        const bool kf = ninl < v->prm.keyframe_min_inliers;  // src/VisualSLAM.cpp:120
        if (kf) {
            // the keyframe's points are placed with the refined pose: wait for it
            if ((rc = wait_tag(&mb->pose_tag[f & 1], pose_tag[f & 1], sP)))
                return rc;
            harvest(f);
            const uint8_t *d = stage_image(v, rights[f], mem, &rc);
            if (rc || (rc = svo_build_pyramid_from_device(ctx, v->pyr_right, d)))
                return rc;
            const double *R9 = R_out + 9 * (size_t)f, *t3 = t_out + 3 * (size_t)f;
            double Rt[12];
            for (int i = 0; i < 3; i++) {
                Rt[4 * i] = R9[3 * i];
                Rt[4 * i + 1] = R9[3 * i + 1];
                Rt[4 * i + 2] = R9[3 * i + 2];
                Rt[4 * i + 3] = t3[i];
            }
            if ((rc = stereo_triangulate(v, v->pyr_cur, v->pyr_right, Rt, v->ref2d, v->ref3d, &v->nref)))
                return rc;
            next_built = speculated;  // the speculative tracking is void, the next pyramid is not
        } else {
            pending = f;  // its refinement may still be running beside the next frame's filters
            std::swap(v->ref2d, v->trk2d);
            std::swap(v->ref3d, v->trk3d);
            v->nref = v->ntrk;
            spec = speculated;
        }
        std::swap(v->pyr_ref, v->pyr_cur);
        if (keyframe_out)
            keyframe_out[f] = kf ? 1 : 0;
        if (n_done)
            *n_done = f + 1;
    }
    if (pending >= 0) {
        if ((rc = wait_tag(&mb->pose_tag[pending & 1], pose_tag[pending & 1], pipeline ? sB : sA)))
            return rc;
        harvest(pending);
    }
    v->has_cur = false;
    return SVO_OK;
}

// Several chunks that share ONE context, advanced in lock step by one host thread on the context's
// stream: per frame the pyramids of all of them, ONE pyramidal-LK launch carrying all their
// tracking passes (the launch lasts as long as its slowest keypoint, so k jobs cost little more
// than one), then each chunk's filters + PnP, then each chunk's policy.  Every chunk gets
// exactly what svo_vo_run_chunk(pipeline = 0) gives it alone.
static int run_chunk_group(svo_chunk_job **jobs, int k)
{
    struct GS {
        svo_chunk_job *j;
        svo_vo *v;
        bool active = true;
        int pending = -1, early_tag = 0, pose_tag[2] = {0, 0}, n = 0;
        int *cnt_trk = nullptr;
        PnpRecord *d_rec = nullptr;
    };
    std::vector<GS> gs(k);
    svo_ctx *ctx = jobs[0]->vo->ctx;
    hipStream_t st = ctx->stream;
    int n_frames_max = 0;
    for (int a = 0; a < k; a++) {
        gs[a].j = jobs[a];
        gs[a].v = jobs[a]->vo;
        jobs[a]->n_done = 0;
        jobs[a]->rc = SVO_OK;
        if (jobs[a]->mem != SVO_MEM_DEVICE) {
            svo_set_error("chunks that share a context take device images");
            return SVO_ERR_ARG;
        }
        if (jobs[a]->vo->prm.policy != SVO_POLICY_SLAM) {
            svo_set_error("the chunk runner drives the live policy (SVO_POLICY_SLAM) only");
            return SVO_ERR_ARG;
        }
        n_frames_max = jobs[a]->n_frames > n_frames_max ? jobs[a]->n_frames : n_frames_max;
    }
    auto wait_tag = [&](const int *slot, int tag) -> int { return wait_mailbox_tag(slot, tag, st); };
    auto harvest = [&](GS &g, int f) {  // pose composition (src/VisualSLAM.cpp:70-74)
        const PnpRecord *rec = &g.v->mbox->rec[f & 1];
        double *R9 = g.j->R_out + 9 * (size_t)f, *t3 = g.j->t_out + 3 * (size_t)f;
        for (int i = 0; i < 3; i++)
            for (int c = 0; c < 3; c++)
                R9[3 * i + c] = rec->R[3 * c + i];
        for (int i = 0; i < 3; i++)
            t3[i] = -(R9[3 * i] * rec->tvec[0] + R9[3 * i + 1] * rec->tvec[1] + R9[3 * i + 2] * rec->tvec[2]);
        memcpy(g.v->R, R9, sizeof(g.v->R));
        memcpy(g.v->t, t3, sizeof(g.v->t));
    };
    auto launch_pnp = [&](GS &g, int f, double thr, double conf, int stage) -> int {
        svo_vo *v = g.v;
        g.early_tag = ++v->tag;
        g.pose_tag[f & 1] = ++v->tag;
        svo_pnp_job q = pnp_job(v, g.n, g.cnt_trk, thr, conf, stage_seed(v, stage), g.d_rec, g.early_tag, f & 1, g.pose_tag[f & 1]);
        return svo_launch_pnp_ransac_batch(ctx, 1, &q);
    };
    auto stop = [&](GS &g, int rc) {  // this chunk ends here (tracking lost or an error); the others go on
        g.active = false;
        g.j->rc = rc;
    };
    int rc;
    {   // ---- chunks that start here: svo_vo_init of all of them as one set of launches ----
        svo_vo *vs[SVO_LK_MAX_JOBS];
        svo_pyramid *pl[SVO_LK_MAX_JOBS], *pr[SVO_LK_MAX_JOBS];
        const uint8_t *il[SVO_LK_MAX_JOBS], *ir[SVO_LK_MAX_JOBS];
        const double *Rts[SVO_LK_MAX_JOBS];
        float *o2d[SVO_LK_MAX_JOBS], *o3d[SVO_LK_MAX_JOBS];
        int *nout[SVO_LK_MAX_JOBS];
        int ni = 0;
        for (GS &g : gs) {
            if (!g.j->init_left != !g.j->init_right) {
                svo_set_error("svo_vo_run_chunks: init_left and init_right go together");
                return SVO_ERR_ARG;
            }
            if (!g.j->init_left)
                continue;
            svo_vo *v = g.v;
            v->frame = 0;
            for (int i = 0; i < 9; i++)
                v->R[i] = (i % 4) == 0;
            v->t[0] = v->t[1] = v->t[2] = 0;
            v->has_cur = false;
            vs[ni] = v;
            pl[ni] = v->pyr_ref;
            pr[ni] = v->pyr_right;
            il[ni] = g.j->init_left;
            ir[ni] = g.j->init_right;
            static const double I34[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
            Rts[ni] = I34;  // as svo_vo_init
            o2d[ni] = v->ref2d;
            o3d[ni] = v->ref3d;
            nout[ni] = &v->nref;
            ni++;
        }
        if (ni > 0) {
            if ((rc = svo_build_pyramids_from_device(ctx, ni, pl, il)) ||
                (rc = svo_build_pyramids_from_device(ctx, ni, pr, ir)) ||
                (rc = stereo_triangulate_batch(ni, vs, pl, pr, Rts, o2d, o3d, nout)))
                return rc;
            for (GS &g : gs)
                if (g.j->init_left)
                    g.j->n_init_points = g.v->nref;
        }
    }
    for (int f = 0; f < n_frames_max; f++) {
        // ---- pyramids + ONE tracking launch for all the chunks still running ----
        LkJob lk[SVO_LK_MAX_JOBS];
        // the pyramids of this step: the left images first, then -- in the same set of launches -- the
        // right images of the same chunks.  A right pyramid is only read if its chunk turns out to keyframe
        // in this step (about half of them do), but built here it costs workgroups, not launches: beside the
        // other contexts' tracking launches every launch of this latency chain waits for its wave slots.
        svo_pyramid *pyrs[2 * SVO_LK_MAX_JOBS];
        const uint8_t *imgs[2 * SVO_LK_MAX_JOBS];
        svo_pyramid *rpyr[SVO_LK_MAX_JOBS];
        const uint8_t *rimg0[SVO_LK_MAX_JOBS];
        int nl = 0;
        for (GS &g : gs) {
            if (!g.active || f >= g.j->n_frames) {
                g.active = g.active && f < g.j->n_frames;
                continue;
            }
            svo_vo *v = g.v;
            v->frame++;
            g.n = v->nref;
            if (g.n < 5) {
                if (g.pending >= 0 && wait_tag(&v->mbox->pose_tag[g.pending & 1], g.pose_tag[g.pending & 1]) == SVO_OK)
                    harvest(g, g.pending);
                g.pending = -1;
                svo_set_error("tracking lost: %d reference points", g.n);
                stop(g, SVO_ERR_TRACKING_LOST);
                continue;
            }
            pyrs[nl] = v->pyr_cur;
            imgs[nl] = g.j->lefts[f];
            rpyr[nl] = v->pyr_right;
            rimg0[nl] = g.j->rights[f];
            LkJob &q = lk[nl++];
            q.prev = v->pyr_ref->dev;
            q.next = v->pyr_cur->dev;
            q.dprev = v->pyr_ref->dbase;
            q.prev_pts = v->ref2d;
            q.n_cap = g.n;
            q.d_n = nullptr;
            q.next_pts = v->a2;
            q.status = v->status;
            q.err = nullptr;
            q.min_eig = nullptr;
        }
        if (nl == 0)
            break;
        for (int a = 0; a < nl; a++) {
            pyrs[nl + a] = rpyr[a];
            imgs[nl + a] = rimg0[a];
        }
        if ((rc = svo_build_pyramids_from_device(ctx, 2 * nl, pyrs, imgs)) ||
            (rc = svo_launch_lk_batch(ctx, nl, lk, pyrs[0])))
            return rc;
        // ---- filters and PnP: every stage is ONE set of launches for all the chunks ----
        svo_pnp_job pj[SVO_LK_MAX_JOBS];
        svo_fransac_job fj[SVO_LK_MAX_JOBS];
        svo_compact_job c1[SVO_LK_MAX_JOBS], c2[SVO_LK_MAX_JOBS];
        GS *pg[SVO_LK_MAX_JOBS];
        int np = 0;
        for (GS &g : gs) {
            if (!g.active)
                continue;
            svo_vo *v = g.v;
            g.cnt_trk = v->d_cnt + ((f & 1) ? 9 : 1);
            g.d_rec = v->d_rec + (f & 1);
            // status filter (src/tracking.cpp:54-64) and, after the F-RANSAC, its mask filter (:77-88)
            c1[np] = {v->status, g.n, nullptr, {v->ref2d, v->a2, v->ref3d}, {v->b2, v->c2, v->a3}, {2, 2, 3}, v->d_cnt};
            c2[np] = {v->mask, g.n, v->d_cnt, {v->c2, v->a3, nullptr}, {v->trk2d, v->trk3d, nullptr}, {2, 3, 0}, g.cnt_trk};
            svo_fransac_job &q = fj[np];
            q.p1 = v->b2;
            q.p2 = v->c2;
            q.cap = g.n;
            q.d_n = v->d_cnt;
            q.threshold = v->prm.f_thr_temporal;
            q.confidence = 0.99;
            q.max_iters = 1000;
            q.seed = stage_seed(v, 0);
            q.mask = v->mask;
            q.d_F = nullptr;
            q.d_count = nullptr;
            q.d_iters = nullptr;
            q.then_compact = &c2[np];  // the mask compaction rides with the F-RANSAC
            pg[np++] = &g;
        }
        if ((rc = svo_launch_compact_batch(ctx, np, c1)) || (rc = svo_launch_fransac_batch(ctx, np, fj)))
            return rc;
        for (int a = 0; a < np; a++) {
            GS &g = *pg[a];
            svo_vo *v = g.v;
            g.early_tag = ++v->tag;
            g.pose_tag[f & 1] = ++v->tag;
            pj[a] = pnp_job(v, g.n, g.cnt_trk, 1.0, 0.99, stage_seed(v, 1), g.d_rec, g.early_tag, f & 1, g.pose_tag[f & 1]);
        }
        if ((rc = svo_launch_pnp_ransac_batch(ctx, np, pj)))  // every record is published by its own workgroup
            return rc;
        // ---- policy of every chunk; the chunks that keyframe are collected ----
        GS *kfs[SVO_LK_MAX_JOBS];
        int nk = 0;
        for (GS &g : gs) {
            if (!g.active)
                continue;
            svo_vo *v = g.v;
            Mailbox *mb = v->mbox;
            if ((rc = wait_tag(&mb->early[0], g.early_tag)))
                return rc;
            if (g.pending >= 0) {
                if ((rc = wait_tag(&mb->pose_tag[g.pending & 1], g.pose_tag[g.pending & 1])))
                    return rc;
                harvest(g, g.pending);
                g.pending = -1;
            }
            int ninl = mb->early[1];
            if (ninl < 10) {  // retry at 8 px / 0.98 (src/keyFrameManagement.cpp:85-92)
                if ((rc = wait_tag(&mb->pose_tag[f & 1], g.pose_tag[f & 1])) || (rc = launch_pnp(g, f, 8.0, 0.98, 2)) ||
                    (rc = wait_tag(&mb->early[0], g.early_tag)))
                    return rc;
                ninl = mb->early[1];
            }
            v->ntrk = mb->early[2];
            if (g.j->inliers_out)
                g.j->inliers_out[f] = ninl;
            if (g.j->tracked_out)
                g.j->tracked_out[f] = v->ntrk;
            if (ninl < 10) {
                (void)wait_tag(&mb->pose_tag[f & 1], g.pose_tag[f & 1]);
                svo_set_error("tracking lost at frame %d: %d PnP inliers", v->frame, ninl);
                stop(g, SVO_ERR_TRACKING_LOST);
                continue;
            }
            const bool kf = ninl < v->prm.keyframe_min_inliers;  // src/VisualSLAM.cpp:120
            if (kf) {
                kfs[nk++] = &g;
            } else {
                g.pending = f;
                std::swap(v->ref2d, v->trk2d);
                std::swap(v->ref3d, v->trk3d);
                v->nref = v->ntrk;
            }
            if (g.j->keyframe_out)
                g.j->keyframe_out[f] = kf ? 1 : 0;
        }
        // ---- the keyframe path of all the chunks that need it, every stage one set of launches ----
        if (nk > 0) {
            svo_vo *vs[SVO_LK_MAX_JOBS];
            svo_pyramid *lefts[SVO_LK_MAX_JOBS], *rights[SVO_LK_MAX_JOBS];
            const uint8_t *rimg[SVO_LK_MAX_JOBS];
            double Rt[SVO_LK_MAX_JOBS][12];
            const double *Rts[SVO_LK_MAX_JOBS];
            float *o2d[SVO_LK_MAX_JOBS], *o3d[SVO_LK_MAX_JOBS];
            int *nout[SVO_LK_MAX_JOBS];
            for (int a = 0; a < nk; a++) {
                GS &g = *kfs[a];
                svo_vo *v = g.v;
                // the keyframe's points are placed with the refined pose: wait for it
                if ((rc = wait_tag(&v->mbox->pose_tag[f & 1], g.pose_tag[f & 1])))
                    return rc;
                harvest(g, f);
                const double *R9 = g.j->R_out + 9 * (size_t)f, *t3 = g.j->t_out + 3 * (size_t)f;
                for (int i = 0; i < 3; i++) {
                    Rt[a][4 * i] = R9[3 * i];
                    Rt[a][4 * i + 1] = R9[3 * i + 1];
                    Rt[a][4 * i + 2] = R9[3 * i + 2];
                    Rt[a][4 * i + 3] = t3[i];
                }
                vs[a] = v;
                lefts[a] = v->pyr_cur;
                rights[a] = v->pyr_right;
                rimg[a] = g.j->rights[f];
                Rts[a] = Rt[a];
                o2d[a] = v->ref2d;
                o3d[a] = v->ref3d;
                nout[a] = &v->nref;
            }
            (void)rimg;  // the right pyramids of this step were built with the left ones
            if ((rc = stereo_triangulate_batch(nk, vs, lefts, rights, Rts, o2d, o3d, nout)))
                return rc;
        }
        for (GS &g : gs) {
            if (!g.active)
                continue;
            std::swap(g.v->pyr_ref, g.v->pyr_cur);
            g.j->n_done = f + 1;
        }
    }
    for (GS &g : gs) {
        if (g.pending >= 0) {
            if ((rc = wait_tag(&g.v->mbox->pose_tag[g.pending & 1], g.pose_tag[g.pending & 1])))
                return rc;
            harvest(g, g.pending);
        }
        g.v->has_cur = false;
    }
    return SVO_OK;
}

int svo_vo_run_chunks(svo_chunk_job *jobs, int n_jobs)
{
    SVO_CHECK_ARG(jobs && n_jobs >= 1);
    // jobs that share a context form a group (lock step, one tracking launch for all of them);
    // every group runs on its own host thread
    std::vector<std::vector<svo_chunk_job *>> groups;
    for (int a = 0; a < n_jobs; a++) {
        SVO_CHECK_ARG(jobs[a].vo != nullptr);
        bool placed = false;
        for (auto &g : groups)
            if (g[0]->vo->ctx == jobs[a].vo->ctx) {
                for (svo_chunk_job *o : g)
                    if (o->vo == jobs[a].vo) {
                        svo_set_error("svo_vo_run_chunks: a front-end appears in two jobs");
                        return SVO_ERR_ARG;
                    }
                {  // a group's stages go out as one set of launches sized from its first member
                    const svo_vo *x = g[0]->vo, *y = jobs[a].vo;
                    const svo_vo_params &p = x->prm, &q = y->prm;
                    if (x->w != y->w || x->h != y->h || x->c != y->c || x->cap != y->cap || p.grid_step != q.grid_step ||
                        p.anms_keep != q.anms_keep || p.fx != q.fx || p.fy != q.fy || p.cx != q.cx || p.cy != q.cy ||
                        p.baseline != q.baseline) {
                        svo_set_error("svo_vo_run_chunks: the front-ends of one context must share image size, "
                                      "grid step, ANMS budget, intrinsics and baseline");
                        return SVO_ERR_ARG;
                    }
                }
                if ((int)g.size() >= SVO_LK_MAX_JOBS) {
                    svo_set_error("svo_vo_run_chunks: at most %d chunks per context", SVO_LK_MAX_JOBS);
                    return SVO_ERR_ARG;
                }
                g.push_back(&jobs[a]);
                placed = true;
                break;
            }
        if (!placed)
            groups.push_back({&jobs[a]});
    }
    const int ng = (int)groups.size();
    std::vector<std::string> errs(ng);
    std::vector<int> grc(ng, SVO_OK);
    auto body = [&](int gi) {
        auto &g = groups[gi];
        if (hipSetDevice(g[0]->vo->ctx->device) != hipSuccess) {  // the current device is per host thread
            grc[gi] = SVO_ERR_HIP;
            errs[gi] = "hipSetDevice failed";
            return;
        }
        if (g.size() == 1) {
            svo_chunk_job &j = *g[0];
            j.n_done = 0;
            if (!j.init_left != !j.init_right) {
                grc[gi] = SVO_ERR_ARG;
                errs[gi] = "svo_vo_run_chunks: init_left and init_right go together";
                return;
            }
            if (j.init_left && (j.rc = svo_vo_init(j.vo, j.init_left, j.init_right, j.mem, &j.n_init_points))) {
                grc[gi] = j.rc;
                errs[gi] = svo_last_error();
                return;
            }
            j.rc = svo_vo_run_chunk(j.vo, j.lefts, j.rights, j.n_frames, j.mem, j.pipeline, j.R_out, j.t_out,
                                    j.inliers_out, j.tracked_out, j.keyframe_out, &j.n_done);
            if (j.rc && j.rc != SVO_ERR_TRACKING_LOST)
                grc[gi] = j.rc;
        } else {
            grc[gi] = run_chunk_group(g.data(), (int)g.size());
        }
        if (grc[gi])
            errs[gi] = svo_last_error();  // the error text is thread-local
    };
    std::vector<std::thread> th;
    for (int gi = 1; gi < ng; gi++)
        th.emplace_back(body, gi);
    body(0);
    for (auto &t : th)
        t.join();
    for (int gi = 0; gi < ng; gi++)
        if (grc[gi]) {
            svo_set_error("chunk group %d: %s", gi, errs[gi].c_str());
            return grc[gi];
        }
    return SVO_OK;
}

int svo_vo_get_reference(svo_vo *v, float *ref2d, float *ref3d, int cap, int *n, int mem)
{
    SVO_CHECK_ARG(v && n);
    *n = v->nref;
    if (cap < v->nref) {
        svo_set_error("reference set has %d points, capacity %d", v->nref, cap);
        return SVO_ERR_CAPACITY;
    }
    const hipMemcpyKind kind = mem == SVO_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (ref2d)
        SVO_HIP(hipMemcpyAsync(ref2d, v->ref2d, (size_t)v->nref * 8, kind, v->ctx->stream));
    if (ref3d)
        SVO_HIP(hipMemcpyAsync(ref3d, v->ref3d, (size_t)v->nref * 12, kind, v->ctx->stream));
    SVO_HIP(hipStreamSynchronize(v->ctx->stream));
    return SVO_OK;
}

int svo_pnp_ladder(svo_ctx *ctx, const float *obj_f, const float *img_f, int n_f, const float *obj_s, const float *img_s,
                   int n_s, const double *K4, uint64_t seed, double *rvec, double *tvec, int *n_inliers, int *rung, int mem)
{
    SVO_CHECK_ARG(ctx && obj_f && img_f && obj_s && img_s && n_f >= 0 && n_s >= 0 && K4 && rvec && tvec);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    SVO_HIP(hipSetDevice(ctx->device));
    int rc;
    const int cap = n_f > n_s ? n_f : n_s;
    if ((rc = ctx->s_e.ensure(sizeof(PnpRecord) * 2 + 64)) || (rc = ctx->s_f.ensure((size_t)(cap + 1) * 4)) ||
        (rc = ctx->s_g.ensure(64)))
        return rc;
    const float *of = obj_f, *uf = img_f, *os = obj_s, *us = img_s;
    if (mem == SVO_MEM_HOST) {
        if ((rc = ctx->s_a.ensure((size_t)(n_f + 1) * 12)) || (rc = ctx->s_b.ensure((size_t)(n_f + 1) * 8)) ||
            (rc = ctx->s_c.ensure((size_t)(n_s + 1) * 12)) || (rc = ctx->s_d.ensure((size_t)(n_s + 1) * 8)))
            return rc;
        SVO_HIP(hipMemcpyAsync(ctx->s_a.p, obj_f, (size_t)n_f * 12, hipMemcpyHostToDevice, ctx->stream));
        SVO_HIP(hipMemcpyAsync(ctx->s_b.p, img_f, (size_t)n_f * 8, hipMemcpyHostToDevice, ctx->stream));
        SVO_HIP(hipMemcpyAsync(ctx->s_c.p, obj_s, (size_t)n_s * 12, hipMemcpyHostToDevice, ctx->stream));
        SVO_HIP(hipMemcpyAsync(ctx->s_d.p, img_s, (size_t)n_s * 8, hipMemcpyHostToDevice, ctx->stream));
        of = ctx->s_a.as<float>();
        uf = ctx->s_b.as<float>();
        os = ctx->s_c.as<float>();
        us = ctx->s_d.as<float>();
    }
    int *d_cnt = ctx->s_g.as<int>();
    const int hc[2] = {n_f, n_s};
    SVO_HIP(hipMemcpyAsync(d_cnt, hc, sizeof(hc), hipMemcpyHostToDevice, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));  // hc is a stack array
    int r = 0, rin = 0;
    // the launchers skip empty sets: the capacity is at least 1, the live count sits on the device
    if ((rc = ladder_pose(ctx, of, uf, d_cnt, os, us, d_cnt + 1, cap > 0 ? cap : 1, K4, seed + 1, seed + 2,
                          ctx->s_f.as<int>(), reinterpret_cast<PnpRecord *>(ctx->s_e.p), &r, &rin)))
        return rc;
    const PnpRecord *rec = reinterpret_cast<const PnpRecord *>(ctx->pinned);
    if (rung)
        *rung = r;
    if (n_inliers)
        *n_inliers = rin;
    if (rec->n_inliers == 0) {
        svo_set_error("pose ladder: no solution (rung %d, %d points)", r, rec->n_tracked);
        return SVO_ERR_TRACKING_LOST;
    }
    memcpy(rvec, rec->rvec, sizeof(rec->rvec));
    memcpy(tvec, rec->tvec, sizeof(rec->tvec));
    return SVO_OK;
}

int svo_vo_get_keyframe_cloud(svo_vo *v, float *xyz_cam, int cap, int *n, int mem)
{
    SVO_CHECK_ARG(v && n);
    SVO_CHECK_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE);
    *n = v->kf_n;
    if (!xyz_cam)
        return SVO_OK;
    if (cap < v->kf_n) {
        svo_set_error("keyframe cloud has %d points, capacity %d", v->kf_n, cap);
        return SVO_ERR_CAPACITY;
    }
    if (v->kf_n == 0)
        return SVO_OK;
    SVO_HIP(hipMemcpyAsync(xyz_cam, v->b3, (size_t)v->kf_n * 12,
                           mem == SVO_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, v->ctx->stream));
    if (mem == SVO_MEM_HOST)
        SVO_HIP(hipStreamSynchronize(v->ctx->stream));
    return SVO_OK;
}

int svo_vo_capacity(const svo_vo *v) { return v ? v->cap : 0; }

}  // extern "C"
