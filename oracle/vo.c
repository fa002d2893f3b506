/*
 * oracle/vo.c -- CPU restatement of the reference's front-end frame loop
 * (TEST INFRASTRUCTURE; see svo_oracle.h.)
 *
 * Follows /root/reference/src/VisualSLAM.cpp:11-169 (initSequence), with
 *   stereoTriangulate          src/triangulation.cpp:73-166 (dense branch :87-103)
 *   denseLKtracking            src/tracking.cpp:14-28
 *   FmatThresholding           src/tracking.cpp:30-43      (3.0 px, 0.99)
 *   PyrLKtrackFrame2Frame      src/tracking.cpp:46-91      (F-RANSAC 1.0 px, 0.99)
 *   PerspectiveNpointEstimation src/keyFrameManagement.cpp:73-94 (100/1.0/0.99, retry 100/8.0/0.98)
 *   insertKeyFrames            src/keyFrameManagement.cpp:9-31
 * The glue is fully specified in the reference tree; the library calls go to the
 * restatements in lk.c / geometry.c / pnp.c.
 *
 * Deviations: the second compaction loop of PyrLKtrackFrame2Frame
 * runs over the mask length (the reference indexes past the compacted arrays); optional
 * ANMS (anms_keep > 0) with the level-0 LK minimum eigenvalue as response, because the
 * reference's grid keypoints all carry response 0; per-frame RANSAC seeds are
 * seed + 8*frame + stage.
 */
#include "svo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_vo_default_params(orc_vo_params *p)
{
    p->fx = 7.188560000000e+02; /* include/visualSLAM.h:82-87 */
    p->fy = 7.188560000000e+02;
    p->cx = 6.071928000000e+02;
    p->cy = 1.852157000000e+02;
    p->baseline = 0.54; /* include/visualSLAM.h:68 */
    p->grid_step = 30;  /* src/triangulation.cpp:89 */
    p->anms_keep = 0;
    p->keyframe_min_inliers = 200; /* src/VisualSLAM.cpp:120 */
    p->f_thr_stereo = 3.0;
    p->f_thr_temporal = 1.0;
    p->seed = 0;
    p->policy = 0;
    p->pnp_retry_below = 10; /* src/keyFrameManagement.cpp:85 */
    p->pnp_lost_below = 10;  /* src/keyFrameManagement.cpp:89 */
}

struct orc_vo {
    orc_vo_params prm;
    int w, h, c, cap;
    uint8_t *ref_img;
    float *ref2d, *ref3d; /* current reference set (2-D in ref_img, 3-D world) */
    int nref;
    float *trk2d, *trk3d; /* set produced by the last localisation */
    int ntrk;
    int frame;
    double R[9], t[3];
    /* scratch */
    float *a2, *b2, *c2, *a3, *resp;
    uint8_t *mask;
    int *idx;
};

orc_vo *orc_vo_create(const orc_vo_params *p, int w, int h, int c)
{
    orc_vo *v = (orc_vo *)calloc(1, sizeof(orc_vo));
    v->prm = *p;
    v->w = w;
    v->h = h;
    v->c = c;
    v->cap = orc_grid_keypoints(h, w, p->grid_step, 0, 0);
    if (v->cap < 16)
        v->cap = 16;
    size_t n = (size_t)v->cap;
    v->ref_img = (uint8_t *)malloc((size_t)w * h * c);
    v->ref2d = (float *)malloc(n * 8);
    v->ref3d = (float *)malloc(n * 12);
    v->trk2d = (float *)malloc(n * 8);
    v->trk3d = (float *)malloc(n * 12);
    v->a2 = (float *)malloc(n * 8);
    v->b2 = (float *)malloc(n * 8);
    v->c2 = (float *)malloc(n * 8);
    v->a3 = (float *)malloc(n * 12);
    v->resp = (float *)malloc(n * 4);
    v->mask = (uint8_t *)malloc(n);
    v->idx = (int *)malloc(n * 4);
    for (int i = 0; i < 9; i++)
        v->R[i] = (i % 4) == 0;
    return v;
}

void orc_vo_destroy(orc_vo *v)
{
    if (!v)
        return;
    free(v->ref_img);
    free(v->ref2d);
    free(v->ref3d);
    free(v->trk2d);
    free(v->trk3d);
    free(v->a2);
    free(v->b2);
    free(v->c2);
    free(v->a3);
    free(v->resp);
    free(v->mask);
    free(v->idx);
    free(v);
}

static uint64_t stage_seed(const orc_vo *v, int stage) { return v->prm.seed + 8ull * (uint64_t)v->frame + stage; }

/* src/triangulation.cpp:73-166, dense branch.  out2d/out3d (camera frame); returns count */
static int stereo_triangulate(orc_vo *v, const uint8_t *left, const uint8_t *right, float *out2d, float *out3d)
{
    int n = orc_grid_keypoints(v->h, v->w, v->prm.grid_step, v->a2, v->cap);
    float *pts = v->a2, *trk = v->b2;
    orc_lk_track(left, right, v->w, v->h, v->c, pts, n, trk, v->mask, 0, v->resp, 0);
    if (v->prm.anms_keep > 0) {
        /* ANMS on the grid with the level-0 minimum eigenvalue as response; the kept points
         * come out in response order (src/ANMS.cpp:26-30,60-64) */
        int k = orc_anms(pts, v->resp, n, v->prm.anms_keep, v->idx, 0);
        float *p2 = v->c2, *t2 = (float *)malloc((size_t)k * 8);
        uint8_t *m2 = (uint8_t *)malloc((size_t)k);
        for (int i = 0; i < k; i++) {
            p2[2 * i] = pts[2 * v->idx[i]];
            p2[2 * i + 1] = pts[2 * v->idx[i] + 1];
            t2[2 * i] = trk[2 * v->idx[i]];
            t2[2 * i + 1] = trk[2 * v->idx[i] + 1];
            m2[i] = v->mask[v->idx[i]];
        }
        memcpy(pts, p2, (size_t)k * 8);
        memcpy(trk, t2, (size_t)k * 8);
        memcpy(v->mask, m2, (size_t)k);
        free(t2);
        free(m2);
        n = k;
    }
    /* denseLKtracking: keep status == 1 (src/tracking.cpp:20-27) */
    int m = 0;
    for (int i = 0; i < n; i++)
        if (v->mask[i] == 1) {
            pts[2 * m] = pts[2 * i];
            pts[2 * m + 1] = pts[2 * i + 1];
            trk[2 * m] = trk[2 * i];
            trk[2 * m + 1] = trk[2 * i + 1];
            m++;
        }
    /* FmatThresholding (src/tracking.cpp:30-43) */
    orc_fransac_params fp = {v->prm.f_thr_stereo, 0.99, 1000, stage_seed(v, 3), 0};
    orc_fransac(pts, trk, m, &fp, v->mask, 0, 0);
    int k = 0;
    for (int i = 0; i < m; i++)
        if (v->mask[i] == 1) {
            pts[2 * k] = pts[2 * i];
            pts[2 * k + 1] = pts[2 * i + 1];
            trk[2 * k] = trk[2 * i];
            trk[2 * k + 1] = trk[2 * i + 1];
            k++;
        }
    double P1[12], P2[12];
    orc_stereo_projections(v->prm.fx, v->prm.fy, v->prm.cx, v->prm.cy, v->prm.baseline, P1, P2);
    orc_triangulate(P1, P2, pts, trk, k, out3d, 0);
    memcpy(out2d, pts, (size_t)k * 8);
    return k;
}

int orc_vo_init(orc_vo *v, const uint8_t *left, const uint8_t *right)
{
    v->frame = 0;
    for (int i = 0; i < 9; i++)
        v->R[i] = (i % 4) == 0;
    v->t[0] = v->t[1] = v->t[2] = 0;
    v->nref = stereo_triangulate(v, left, right, v->ref2d, v->ref3d);
    memcpy(v->ref_img, left, (size_t)v->w * v->h * v->c);
    return v->nref;
}

/* The pose ladder of the older visualOdometry::initSequence, src/bundleAdjust.cpp:462-480:
 * solvePnPRansac(100, 4.0, 0.99) on the F-filtered set; < 20 inliers or tvec.x > 1000: "skipping RANSAC
 * layer and retracking" = the same on the status-filtered set (PyrLKtrackFrame2Frame(..., false)); < 10 or
 * tvec.x > 1000: plain solvePnP; and plain solvePnP once more when the last RANSAC kept < 10 (:475-477).
 * rung: 0 / 1 / 2 = which of them decided; n_inliers: the last RANSAC's count.  Stage seeds seed + 1 / + 2.
 * Returns 0, -1 when solvePnP has no solution. */
int orc_pnp_ladder(const float *obj_f, const float *img_f, int n_f, const float *obj_s, const float *img_s, int n_s,
                   const double *K4, uint64_t seed, double *rvec, double *tvec, int *n_inliers, int *rung)
{
    const float *o3 = obj_f, *o2 = img_f;
    int cnt = n_f, plain = 0, r = 0;
    int *idx = (int *)malloc(sizeof(int) * (size_t)((n_f > n_s ? n_f : n_s) + 1));
    orc_pnp_params p4 = {100, 4.0, 0.99, seed + 1, 20};
    int ninl = orc_pnp_ransac(o3, o2, cnt, K4, &p4, rvec, tvec, idx, 0);
    if (ninl < 20 || tvec[0] > 1000) {
        r = 1;
        o3 = obj_s;
        o2 = img_s;
        cnt = n_s;
        orc_pnp_params p4b = {100, 4.0, 0.99, seed + 2, 20};
        ninl = orc_pnp_ransac(o3, o2, cnt, K4, &p4b, rvec, tvec, idx, 0);
        if (ninl < 10 || tvec[0] > 1000)
            plain = 1;
    }
    free(idx);
    if (ninl < 10)
        plain = 1;
    if (n_inliers)
        *n_inliers = ninl;
    int rc = 0;
    if (plain) {
        r = 2;
        rc = orc_solve_pnp(o3, o2, cnt, K4, rvec, tvec, 0) != 0 ? -1 : 0;
    }
    if (rung)
        *rung = r;
    return rc;
}

/* PerspectiveNpointEstimation + pose composition (VisualSLAM.cpp:64-74).  Leaves the tracked
 * set in v->trk2d/trk3d.  Returns 0 ok, -1 tracking lost (SHUTDOWN_FLAG). */
int orc_vo_localize(orc_vo *v, const uint8_t *left, double *R, double *t, int *n_inliers, int *n_tracked)
{
    v->frame++;
    const int n = v->nref;
    float *trk = v->a2;
    orc_lk_track(v->ref_img, left, v->w, v->h, v->c, v->ref2d, n, trk, v->mask, 0, 0, 0);
    /* status compaction of (ref 2-D, 3-D, tracked 2-D): src/tracking.cpp:66-72 */
    float *r2 = v->b2, *t2 = v->c2, *r3 = v->a3;
    int m = 0;
    for (int i = 0; i < n; i++)
        if (v->mask[i] == 1) {
            r2[2 * m] = v->ref2d[2 * i];
            r2[2 * m + 1] = v->ref2d[2 * i + 1];
            t2[2 * m] = trk[2 * i];
            t2[2 * m + 1] = trk[2 * i + 1];
            memcpy(r3 + 3 * m, v->ref3d + 3 * i, 12);
            m++;
        }
    orc_fransac_params fp = {v->prm.f_thr_temporal, 0.99, 1000, stage_seed(v, 0), 0};
    orc_fransac(r2, t2, m, &fp, v->mask, 0, 0);
    int k = 0;
    for (int i = 0; i < m; i++) /* over the mask length (reference bug: tracking.cpp:78) */
        if (v->mask[i] == 1) {
            v->trk2d[2 * k] = t2[2 * i];
            v->trk2d[2 * k + 1] = t2[2 * i + 1];
            memcpy(v->trk3d + 3 * k, r3 + 3 * i, 12);
            k++;
        }
    v->ntrk = k;
    if (n_tracked)
        *n_tracked = k;
    const double K4[4] = {v->prm.fx, v->prm.fy, v->prm.cx, v->prm.cy};
    double rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0};
    if (v->prm.policy == 1) { /* visualOdometry::initSequence, src/bundleAdjust.cpp:452-480 */
        int rung = 0, ninl = 0;
        const int rc = orc_pnp_ladder(v->trk3d, v->trk2d, k, r3, t2, m, K4, v->prm.seed + 8ull * (uint64_t)v->frame,
                                      rvec, tvec, &ninl, &rung);
        if (rung >= 1) { /* the set the pose was computed from is the tracked set */
            memcpy(v->trk2d, t2, (size_t)m * 8);
            memcpy(v->trk3d, r3, (size_t)m * 12);
            v->ntrk = m;
            if (n_tracked)
                *n_tracked = m;
        }
        if (n_inliers)
            *n_inliers = ninl;
        if (rc)
            return -1; /* upstream: cv::Exception out of solvePnP */
        orc_compose_camera_pose(rvec, tvec, R, t);
        return 0;
    }
    orc_pnp_params pp = {100, 1.0, 0.99, stage_seed(v, 1), 20};
    int ninl = orc_pnp_ransac(v->trk3d, v->trk2d, k, K4, &pp, rvec, tvec, v->idx, 0);
    if (ninl < v->prm.pnp_retry_below) { /* src/keyFrameManagement.cpp:85-92 */
        orc_pnp_params pr = {100, 8.0, 0.98, stage_seed(v, 2), 20};
        ninl = orc_pnp_ransac(v->trk3d, v->trk2d, k, K4, &pr, rvec, tvec, v->idx, 0);
        if (ninl < v->prm.pnp_lost_below) {
            if (n_inliers)
                *n_inliers = ninl;
            return -1;
        }
    }
    if (n_inliers)
        *n_inliers = ninl;
    orc_compose_camera_pose(rvec, tvec, R, t); /* VisualSLAM.cpp:70-74 */
    return 0;
}

/* keyframe decision + reference update (VisualSLAM.cpp:93-152) with the pose the caller
 * settled on (after an optional pose-graph re-anchoring of t, VisualSLAM.cpp:81-82). */
int orc_vo_update(orc_vo *v, const uint8_t *left, const uint8_t *right, const double *R, const double *t,
                  int n_inliers, int force_keyframe, int *was_keyframe)
{
    memcpy(v->R, R, sizeof(v->R));
    memcpy(v->t, t, sizeof(v->t));
    /* policy 1: relocalizeFrames on EVERY frame, src/bundleAdjust.cpp:517-519 */
    int kf = n_inliers < v->prm.keyframe_min_inliers || force_keyframe || v->prm.policy == 1;
    if (kf) {
        if (!right)
            return -2;
        float *new3d = (float *)malloc((size_t)v->cap * 12);
        int k = stereo_triangulate(v, left, right, v->ref2d, new3d);
        double Rt[12];
        for (int i = 0; i < 3; i++) {
            Rt[4 * i] = R[3 * i];
            Rt[4 * i + 1] = R[3 * i + 1];
            Rt[4 * i + 2] = R[3 * i + 2];
            Rt[4 * i + 3] = t[i];
        }
        orc_transform_points(Rt, new3d, k, v->ref3d); /* keyFrameManagement.cpp:20-30 */
        free(new3d);
        v->nref = k;
    } else { /* VisualSLAM.cpp:143-146 */
        memcpy(v->ref2d, v->trk2d, (size_t)v->ntrk * 8);
        memcpy(v->ref3d, v->trk3d, (size_t)v->ntrk * 12);
        v->nref = v->ntrk;
    }
    memcpy(v->ref_img, left, (size_t)v->w * v->h * v->c); /* VisualSLAM.cpp:151 */
    if (was_keyframe)
        *was_keyframe = kf;
    return 0;
}

int orc_vo_track(orc_vo *v, const uint8_t *left, const uint8_t *right, int force_keyframe, double *R, double *t,
                 int *n_inliers, int *was_keyframe, int *n_tracked)
{
    int ninl = 0;
    int rc = orc_vo_localize(v, left, R, t, &ninl, n_tracked);
    if (n_inliers)
        *n_inliers = ninl;
    if (rc)
        return rc;
    return orc_vo_update(v, left, right, R, t, ninl, force_keyframe, was_keyframe);
}

int orc_vo_num_ref(const orc_vo *v) { return v->nref; }
void orc_vo_get_ref(const orc_vo *v, float *ref2d, float *ref3d)
{
    if (ref2d)
        memcpy(ref2d, v->ref2d, (size_t)v->nref * 8);
    if (ref3d)
        memcpy(ref3d, v->ref3d, (size_t)v->nref * 12);
}
