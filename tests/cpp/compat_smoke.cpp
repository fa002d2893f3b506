// Smoke test of the C++ compat adaptors: drives the reference's member-function surface -- with the
// reference's own signatures (cv::Mat& for K / R / t / rvec / tvec / [R|t]) -- over a synthetic
// stereo sequence and checks every stage method against the fused svo_vo front-end, which the Python
// parity tests hold to the oracle.  Built twice by tests/test_compat_headers.py: with the POD
// stand-ins, and with -DSVO_WITH_OPENCV -DSVO_WITH_EIGEN against the minimal layout-compatible
// headers under tests/cpp/stubs/ (the branch a maintainer with the real libraries compiles).
// Build: g++ -std=c++17 -Iinclude compat_smoke.cpp -L... -lsvo_hip
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "svo_compat/bundleAdjust.hpp"
#include "svo_compat/stereoCV.hpp"
#include "svo_compat/visualSLAM.hpp"

using namespace svo_compat;

#define FAIL(...)                 \
    do {                          \
        std::printf("FAIL ");     \
        std::printf(__VA_ARGS__); \
        std::printf("\n");        \
        return 1;                 \
    } while (0)

// A fronto-parallel textured wall at depth z0 seen by a camera translated by (cx, cz): every pixel of the
// wall moves by fx * cx / z and scales with z0 / (z0 - cz).  dx = extra horizontal shift (stereo baseline).
static Mat make_view(int w, int h, double fx, double ppx, double ppy, double z0, double cam_x, double cam_z)
{
    Mat m(h, w, CV_8UC3);
    const double z = z0 - cam_z;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const double X = (x - ppx) / fx * z + cam_x, Y = (y - ppy) / fx * z;  // wall coordinates (m)
            // incommensurate waves from 200 px down to 7 px: nothing a 16 px disparity could alias onto
            double g = 128 + 34 * std::sin(0.94 * X + 0.3 * Y + 0.5) + 30 * std::cos(0.41 * X - 1.9 * Y) +
                       26 * std::sin(2.27 * X + 1.1 * Y) * std::cos(1.3 * Y - 0.7 * X) + 22 * std::sin(5.1 * X - 2.3 * Y + 1.0) +
                       16 * std::cos(3.7 * X + 6.1 * Y) + 12 * std::sin(11.1 * X + 4.0 * Y) + 9 * std::cos(9.0 * Y - 13.0 * X) +
                       6 * std::sin(27.0 * X) * std::sin(23.0 * Y);
            const uint8_t b = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
            for (int c = 0; c < 3; c++)
                m.data[((size_t)y * w + x) * 3 + c] = b;
        }
    return m;
}

static double max_abs_diff(const double *a, const double *b, int n)
{
    double d = 0;
    for (int i = 0; i < n; i++)
        d = std::fmax(d, std::fabs(a[i] - b[i]));
    return d;
}

int main()
{
    const int W = 640, H = 240;
    const double FX = 360.0, PX = 320.0, PY = 120.0, BASE = 0.54, Z0 = 12.0;
    // frames: the camera advances 0.25 m per frame towards the wall with a little side-slip
    const int NF = 8;
    Mat Ls[NF], Rs[NF];
    for (int i = 0; i < NF; i++) {
        Ls[i] = make_view(W, H, FX, PX, PY, Z0, 0.03 * i, 0.25 * i);
        Rs[i] = make_view(W, H, FX, PX, PY, Z0, 0.03 * i + BASE, 0.25 * i);
    }
    auto configure = [&](visualSLAM &v, uint64_t seed) {
        v.focal_x = v.focal_y = FX;
        v.cx = PX;
        v.cy = PY;
        v.baseline = BASE;
        v.gridStep = 20;
        v.keyframeMinInliers = 150;
        v.ransacSeed = seed;
    };
    svo_ctx *ctx = shared_context();

    // ---- stage methods ------------------------------------------------------------------------
    visualSLAM s;
    configure(s, 100);
    std::vector<KeyPoint> kps = s.denseKeypointExtractor(Ls[0], 30);
    if (kps.size() != 20u * 6u || kps[0].pt.x != 30.f || kps[0].size != 30.f || kps[0].response != 0.f)
        FAIL("grid: %zu keypoints", kps.size());  // x: 30..600 (20), y: 30..180 (6): `v < dim - step`
    std::vector<Point3f> ref3d;
    std::vector<Point2f> ref2d;
    s.stereoTriangulate(Ls[0], Rs[0], ref3d, ref2d);
    if (ref3d.size() < 150 || ref3d.size() != ref2d.size() || s.colors.size() != ref3d.size())
        FAIL("stereoTriangulate: %zu points", ref3d.size());
    size_t good = 0;
    for (const Point3f &p : ref3d)
        if (std::fabs(p.z - Z0) < 0.03 * Z0)
            good++;
    if (good < ref3d.size() * 9 / 10)
        FAIL("depth: %zu of %zu near %.2f", good, ref3d.size(), Z0);

    // the fused front-end on the same frames with the same seeds (stage seed = seed + 8 * frame + stage)
    svo_vo_params prm;
    svo_vo_default_params(&prm);
    prm.fx = prm.fy = FX;
    prm.cx = PX;
    prm.cy = PY;
    prm.baseline = BASE;
    prm.grid_step = 20;
    prm.keyframe_min_inliers = 150;
    prm.seed = 100;
    svo_vo *vo = nullptr;
    check(svo_vo_create(ctx, &prm, W, H, 3, &vo));
    int n0 = 0;
    check(svo_vo_init(vo, Ls[0].data, Rs[0].data, SVO_MEM_HOST, &n0));
    {
        std::vector<float> f2((size_t)n0 * 2), f3((size_t)n0 * 3);
        int n = 0;
        check(svo_vo_get_reference(vo, f2.data(), f3.data(), n0, &n, SVO_MEM_HOST));
        if ((size_t)n != ref3d.size())
            FAIL("stereoTriangulate kept %zu points, the fused init %d", ref3d.size(), n);
        for (int i = 0; i < n; i++)
            if (f2[2 * i] != ref2d[i].x || f2[2 * i + 1] != ref2d[i].y || f3[3 * i + 2] != ref3d[i].z)
                FAIL("stereoTriangulate differs from the fused init at point %d", i);
    }
    // PerspectiveNpointEstimation with the reference's signature: Mat& rvec / tvec (3x1 CV_64F)
    s.referenceImg = Ls[0];
    s.currentImage = Ls[1];
    s.ransacSeed = 100 + 8;  // frame 1
    std::vector<Point2f> trk2d;
    std::vector<Point3f> trk3d;
    std::vector<int> inliers;
    Mat rvec, tvec;
    s.PerspectiveNpointEstimation(s.referenceImg, s.currentImage, ref2d, ref3d, trk2d, trk3d, rvec, tvec, inliers);
    double R9[9], t3[3];
    int ninl = 0, ntrk = 0;
    check(svo_vo_localize(vo, Ls[1].data, SVO_MEM_HOST, R9, t3, &ninl, &ntrk));
    if (s.SHUTDOWN_FLAG || rvec.rows != 3 || rvec.cols != 1 || tvec.rows != 3 || (int)inliers.size() != ninl ||
        (int)trk2d.size() != ntrk || s.inlierReferencePyrLKPts.size() != trk2d.size())
        FAIL("PerspectiveNpointEstimation: %zu inliers of %zu tracked, fused %d of %d", inliers.size(), trk2d.size(), ninl,
             ntrk);
    // pose composition of src/VisualSLAM.cpp:70-74 from the adaptor's tvec must give the fused pose
    {
        double tv[3] = {tvec.at<double>(0, 0), tvec.at<double>(1, 0), tvec.at<double>(2, 0)}, tc[3];
        for (int i = 0; i < 3; i++)  // t = -R^T tvec with R9 = R^T already (camera in the world)
            tc[i] = -(R9[3 * i] * tv[0] + R9[3 * i + 1] * tv[1] + R9[3 * i + 2] * tv[2]);
        if (max_abs_diff(tc, t3, 3) > 1e-12)
            FAIL("PnP pose: adaptor (%g %g %g) vs fused (%g %g %g)", tc[0], tc[1], tc[2], t3[0], t3[1], t3[2]);
        if (std::fabs(t3[2] - 0.25) > 0.02 || std::fabs(t3[0] - 0.03) > 0.02)
            FAIL("PnP pose off the motion: %g %g %g", t3[0], t3[1], t3[2]);
    }
    // insertKeyFrames with the reference's signature (Mat& pose4dTransform) against the fused forced keyframe
    {
        Mat pose = Mat::zeros(3, 4, CV_64F);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++)
                pose.at<double>(i, j) = R9[3 * i + j];
            pose.at<double>(i, 3) = t3[i];
        }
        std::vector<Point2f> kf2d;
        std::vector<Point3f> kf3d;
        s.ransacSeed = 100 + 8;
        s.insertKeyFrames(0, Ls[1], Rs[1], pose, kf2d, kf3d);
        int was_kf = 0;
        check(svo_vo_update(vo, Rs[1].data, SVO_MEM_HOST, R9, t3, ninl, /*force_keyframe=*/1, &was_kf));
        int n = 0;
        std::vector<float> f2(8192), f3(12288);
        check(svo_vo_get_reference(vo, f2.data(), f3.data(), 4096, &n, SVO_MEM_HOST));
        if (!was_kf || (size_t)n != kf3d.size() || s.untransformed.size() != kf3d.size())
            FAIL("insertKeyFrames: %zu points, fused keyframe %d", kf3d.size(), n);
        for (int i = 0; i < n; i++)
            if (f3[3 * i] != kf3d[i].x || f3[3 * i + 1] != kf3d[i].y || f3[3 * i + 2] != kf3d[i].z || f2[2 * i] != kf2d[i].x)
                FAIL("insertKeyFrames differs from the fused keyframe at point %d", i);
        // update3dtransformation(vector&, Mat&) on the camera-frame cloud gives the same world points
        std::vector<Point3f> again = s.update3dtransformation(s.untransformed, pose);
        for (int i = 0; i < n; i++)
            if (again[i].x != kf3d[i].x || again[i].z != kf3d[i].z)
                FAIL("update3dtransformation differs at point %d", i);
    }
    svo_vo_destroy(vo);

    // SORcloud (src/rosFuncs.cpp:9-39): a far point (-z > 500) and an isolated point go, the wall stays
    {
        std::vector<Point3f> cloud, col;
        for (int i = 0; i < 30; i++)
            for (int j = 0; j < 30; j++) {
                cloud.emplace_back(0.1f * i, 0.1f * j, -5.f - 0.001f * ((i * 7 + j * 3) % 11));
                col.emplace_back((float)i, (float)j, 7.f);
            }
        cloud.emplace_back(0.f, 0.f, -600.f);
        col.emplace_back(0.f, 0.f, -600.f);
        cloud.emplace_back(40.f, -30.f, -80.f);
        col.emplace_back(40.f, -30.f, -80.f);
        const size_t before = cloud.size();
        s.SORcloud(cloud, col);
        bool bad = cloud.size() != col.size() || cloud.size() >= before - 1 || cloud.size() < 600;
        for (const Point3f &q : cloud)
            if (q.z < -50.f)
                bad = true;
        for (size_t i = 0; i < cloud.size() && !bad; i++)  // colours follow their points
            if (std::fabs(col[i].x - 10.f * cloud[i].x) > 1e-3f || col[i].z != 7.f)
                bad = true;
        if (bad)
            FAIL("SORcloud: %zu of %zu kept", cloud.size(), before);
    }
    // checkLoopDetectorStatus: 25 calls on the same image -- too young a database for a closure
    // (query - match > 100 can not hold), every frame is stored, flags stay down
    {
        for (int i = 0; i < 25; i++)
            s.checkLoopDetectorStatus(Ls[0], i);
        if (s.LC_FLAG || s.lastLoopResult.query != 24 || s.cooldownTimer != 0)
            FAIL("checkLoopDetectorStatus: query %d status %d", s.lastLoopResult.query, s.lastLoopResult.status);
        if (s.lastLoopResult.status == SVO_LC_CLOSE_MATCHES_ONLY)  // entries 0..3 are old enough to be queried
            FAIL("checkLoopDetectorStatus: no database query at frame 24");
    }
    // pose graph adaptor: a drifting chain closes; stageForPGO with the reference's four-Mat signature
    {
        globalPoseGraph pg;
        pg.writeResultFile = false;
        pg.initializeGraph();
        Isometry3d T = Isometry3d::Identity();
        for (int i = 1; i < 20; i++) {
            T(2, 3) = 0.9 * i;
            T(0, 3) = 0.01 * i * i;
            pg.augmentNode(T, T);
        }
        pg.addLoopClosure(T, 15);
        std::vector<Isometry3d> est = pg.globalOptimize();
        if (est.size() != 20 || pg.numEdges() != 20)
            FAIL("pose graph sizes");
        visualSLAM q;
        q.poseGraph.initializeGraph();
        Mat Rm = Mat::zeros(3, 3, CV_64F), tm = Mat::zeros(3, 1, CV_64F);
        Rm.at<double>(0, 0) = Rm.at<double>(1, 1) = Rm.at<double>(2, 2) = 1.0;
        tm.at<double>(2, 0) = 0.9;
        q.stageForPGO(Rm, tm, Rm, tm, false);
        q.LCidx = 0;
        q.stageForPGO(Rm, tm, Rm, tm, true);
        if (q.poseGraph.numVertices() != 2 || q.poseGraph.numEdges() != 2 || !q.LC_FLAG)
            FAIL("stageForPGO(Mat, Mat, Mat, Mat, bool): %d vertices %d edges", q.poseGraph.numVertices(),
                 q.poseGraph.numEdges());
    }

    // ---- the fused loop with an injected loop match: the closure branch of processFrame ------------
    // (stageForPGO x2, globalOptimize, t re-anchored, updateOdometry on the device map, forced keyframe)
    visualSLAM f;
    configure(f, 7);
    f.loopMinGap = 3;  // upstream: 100 (src/optimizationStuff.cpp:58); lowered so that 8 frames reach the branch
    f.loopCooldown = 2;
    f.keyframeMinInliers = 100000;  // every frame re-triangulates: several records with retrack for updateOdometry
    f.poseGraph.writeResultFile = false;
    Mat33d Rm;
    Vec3d t;
    double prev_z = 0, anchored_z = 0;
    size_t reprojected = 0;
    for (int i = 0; i < NF; i++) {
        const int match = (i == 5) ? 0 : -1;  // "frame 5 looks like frame 0"
        if (!f.processFrame(Ls[i], Rs[i], match, Rm, t))
            FAIL("processFrame lost tracking at frame %d", i);
        if (i >= 1 && i != 5 && std::fabs((t(2) - prev_z) - 0.25) > 0.05)
            FAIL("processFrame step at frame %d: z %g -> %g", i, prev_z, t(2));
        if (i == 5) {
            // the identity loop edge (vertex 4 -> vertex 0) pulls the chain back: four odometry edges of
            // 0.25 m against one edge that wants 0 m leaves 0.05 m each, so the re-anchored frame 5 sits
            // near 0.2 + 0.25 instead of 1.25
            anchored_z = t(2);
            if (!(t(2) < 0.7 && t(2) > 0.2))
                FAIL("closure branch did not re-anchor t: z = %g", t(2));
            if (f.poseGraph.numEdges() != 6 || f.poseGraph.numVertices() != 6)
                FAIL("closure branch: %d vertices, %d edges", f.poseGraph.numVertices(), f.poseGraph.numEdges());
            if (f.trajectory.size() != 7)  // 6 from updateOdometry + the keyframe's own push (:138-139)
                FAIL("updateOdometry: trajectory %zu", f.trajectory.size());
            // updateOdometry rebuilt mapHistory from the earlier records with retrack -- [R_old | t_new] each --
            // and the closure frame, a forced keyframe, appended its own cloud afterwards
            size_t m = 0;
            for (const keyFrame &k : f.keyFrameHistory) {
                if (!k.retrack || k.idx >= 5)
                    continue;
                if (k.t(2) != f.trajectory[(size_t)k.idx](2))
                    FAIL("updateOdometry: record %d kept its old translation", k.idx);
                Mat34d P = Mat34d::from(k.R, k.t);
                std::vector<Point3f> cam = k.ref3dCoords;
                std::vector<Point3f> want = f.update3dtransformation(cam, P);
                if (m >= f.mapHistory.size())
                    FAIL("updateOdometry: mapHistory has %zu clouds only", f.mapHistory.size());
                const std::vector<Point3f> &got = f.mapHistory[m++];
                if (got.size() != want.size() || got.empty())
                    FAIL("updateOdometry: cloud %zu has %zu points, expected %zu", m - 1, got.size(), want.size());
                for (size_t j = 0; j < got.size(); j++)
                    if (got[j].x != want[j].x || got[j].y != want[j].y || got[j].z != want[j].z)
                        FAIL("updateOdometry: cloud %zu differs at point %zu", m - 1, j);
            }
            if (m == 0 || f.mapHistory.size() != m + 1)
                FAIL("updateOdometry: %zu re-projected clouds, mapHistory %zu", m, f.mapHistory.size());
            reprojected = m;
        }
        prev_z = t(2);
    }
    if (f.keyFrameHistory.size() != (size_t)NF || !f.keyFrameHistory[5].retrack || f.keyFrameHistory[0].retrack)
        FAIL("processFrame: %zu records, closure frame keyframe %d", f.keyFrameHistory.size(),
             (int)f.keyFrameHistory[5].retrack);
    {   // every keyframe of the fused loop pushes its colours (src/VisualSLAM.cpp:125-136): B, G, R floats in [0, 255],
        // point for point with the cloud pushed in the same step
        if (f.colorHistory.empty() || f.mapHistory.empty())
            FAIL("processFrame: colorHistory %zu, mapHistory %zu", f.colorHistory.size(), f.mapHistory.size());
        const std::vector<Point3f> &c = f.colorHistory.back(), &m3 = f.mapHistory.back();
        if (c.size() != m3.size() || c.empty())
            FAIL("processFrame: last colour cloud has %zu entries, its map cloud %zu", c.size(), m3.size());
        double sum = 0;
        for (const Point3f &q : c) {
            if (q.x < 0 || q.x > 255 || q.y < 0 || q.y > 255 || q.z < 0 || q.z > 255 || q.x != (float)(int)q.x)
                FAIL("processFrame: colour (%g, %g, %g) is no pixel value", q.x, q.y, q.z);
            sum += q.x + q.y + q.z;
        }
        if (sum == 0)
            FAIL("processFrame: colours are all zero");
    }

    // ---- BundleAdjust3d2d with the reference's signature -----------------------------------------
    double ba_move = 0;
    {
        visualOdometry od;
        std::vector<Point2f> p2;
        std::vector<Point3f> p3;
        const double tw[3] = {0.1, -0.05, 0.3};
        unsigned lcg = 12345;
        auto rnd = [&]() {
            lcg = lcg * 1664525u + 1013904223u;
            return (double)(lcg >> 8) / (1 << 24);
        };
        for (int i = 0; i < 500; i++) {
            const double X = -8 + 16 * rnd(), Y = -2 + 4 * rnd(), Z = 6 + 30 * rnd();
            p3.emplace_back((float)X, (float)Y, (float)Z);
            const double xc = p3.back().x + tw[0], yc = p3.back().y + tw[1], zc = p3.back().z + tw[2];
            p2.emplace_back((float)(xc / zc * FX + PX + 0.4 * (rnd() - 0.5)), (float)(yc / zc * FX + PY + 0.4 * (rnd() - 0.5)));
        }
        Mat K = Mat::zeros(3, 3, CV_64F), Rw = Mat::zeros(3, 3, CV_64F), tt = Mat::zeros(3, 1, CV_64F);
        K.at<double>(0, 0) = FX;
        K.at<double>(1, 1) = FX;
        K.at<double>(0, 2) = PX;
        K.at<double>(1, 2) = PY;
        K.at<double>(2, 2) = 1;
        Rw.at<double>(0, 0) = Rw.at<double>(1, 1) = Rw.at<double>(2, 2) = 1;
        const double t0[3] = {0.16, -0.08, 0.36};
        for (int i = 0; i < 3; i++)
            tt.at<double>(i, 0) = t0[i];
        od.BundleAdjust3d2d(p2, p3, K, Rw, tt);
        double e0 = 0, e1 = 0;
        for (int i = 0; i < 3; i++) {
            e0 += (t0[i] - tw[i]) * (t0[i] - tw[i]);
            e1 += (tt.at<double>(i, 0) - tw[i]) * (tt.at<double>(i, 0) - tw[i]);
        }
        if (!(e1 < 0.25 * e0) || od.lastInfo[3] != 10 || !(od.lastInfo[1] < 1e-9 * od.lastInfo[0]))
            FAIL("BundleAdjust3d2d: |t - truth|^2 %g -> %g, chi2 %g -> %g after %g iterations", e0, e1, od.lastInfo[0],
                 od.lastInfo[1], od.lastInfo[3]);
        if (Rw.at<double>(0, 0) != 1.0)
            FAIL("BundleAdjust3d2d wrote R back (upstream writes t only)");
        ba_move = std::sqrt(e1);
    }

    StereoProcess sp;
    sp.focal_x = sp.focal_y = FX;
    sp.cx = PX;
    sp.cy = PY;
    sp.baseline = BASE;
    std::vector<Point3f> o3;
    sp.stereoTriangulate(Ls[0], Rs[0], o3);
    if (o3.size() < 50)
        FAIL("StereoProcess::stereoTriangulate: %zu points", o3.size());
    std::printf("compat smoke ok: %zu stereo points at %.1f m, PnP / keyframe adaptors equal the fused front-end, closure "
                "branch re-anchored t to z = %.3f, %zu map clouds re-projected, BA |t - truth| = %.4f, StereoProcess %zu "
                "points\n",
                ref3d.size(), Z0, anchored_z, reprojected, ba_move, o3.size());
    return 0;
}
