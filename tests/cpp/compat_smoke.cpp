// Smoke test of the C++ compat adaptors: drives the reference's member-function surface
// (stage methods) over a synthetic textured stereo pair and checks it against the fused
// svo_vo front-end.  Build: g++ -std=c++17 -Iinclude compat_smoke.cpp -L... -lsvo_hip
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "svo_compat/stereoCV.hpp"
#include "svo_compat/visualSLAM.hpp"

using namespace svo_compat;

static Mat make_image(int w, int h, float shift)
{
    Mat m(h, w, 3);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float u = (x - shift) * 0.11f, v = y * 0.13f;
            float g = 128 + 60 * std::sin(u) * std::cos(v) + 40 * std::sin(0.37f * u + 1.3f * v) + 20 * std::cos(2.1f * u - 0.7f * v);
            uint8_t b = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
            for (int c = 0; c < 3; c++)
                m.ptr()[((size_t)y * w + x) * 3 + c] = b;
        }
    return m;
}

int main()
{
    const int W = 640, H = 240;
    Mat L = make_image(W, H, 0.f), R = make_image(W, H, -8.f);  // 8 px disparity everywhere
    visualSLAM s;
    std::vector<KeyPoint> kps = s.denseKeypointExtractor(L, 30);
    if (kps.size() != 20u * 6u || kps[0].pt.x != 30.f || kps[0].size != 30.f || kps[0].response != 0.f) {
        std::printf("FAIL grid: %zu keypoints\n", kps.size());  // x: 30..600 (20), y: 30..180 (6): `v < dim - step`
        return 1;
    }
    std::vector<Point3f> p3;
    std::vector<Point2f> p2;
    s.stereoTriangulate(L, R, p3, p2);
    if (p3.size() < 50 || p3.size() != p2.size() || s.colors.size() != p3.size()) {
        std::printf("FAIL stereoTriangulate: %zu points\n", p3.size());
        return 1;
    }
    // disparity 8 px -> z = fx * b / 8
    const double z_expect = s.focal_x * s.baseline / 8.0;
    size_t good = 0;
    for (const Point3f &p : p3)
        if (std::fabs(p.z - z_expect) < 0.05 * z_expect)
            good++;
    if (good < p3.size() * 8 / 10) {
        std::printf("FAIL depth: %zu of %zu near %.2f\n", good, p3.size(), z_expect);
        return 1;
    }
    // SORcloud (src/rosFuncs.cpp:9-39): a far point (-z > 500) and an isolated point go, the wall stays
    {
        std::vector<Point3f> cloud, col;
        for (int i = 0; i < 30; i++)
            for (int j = 0; j < 30; j++) {
                Point3f q;
                q.x = 0.1f * i;
                q.y = 0.1f * j;
                q.z = -5.f - 0.001f * ((i * 7 + j * 3) % 11);
                cloud.push_back(q);
                Point3f c;
                c.x = (float)i;
                c.y = (float)j;
                c.z = 7.f;
                col.push_back(c);
            }
        Point3f far_pt, lone;
        far_pt.x = 0.f, far_pt.y = 0.f, far_pt.z = -600.f;
        lone.x = 40.f, lone.y = -30.f, lone.z = -80.f;
        cloud.push_back(far_pt);
        col.push_back(far_pt);
        cloud.push_back(lone);
        col.push_back(lone);
        const size_t before = cloud.size();
        s.SORcloud(cloud, col);
        bool bad = cloud.size() != col.size() || cloud.size() >= before - 1 || cloud.size() < 600;
        for (const Point3f &q : cloud)
            if (q.z < -50.f)
                bad = true;
        for (size_t i = 0; i < cloud.size() && !bad; i++)  // colours follow their points
            if (std::fabs(col[i].x - 10.f * cloud[i].x) > 1e-3f || col[i].z != 7.f)
                bad = true;
        if (bad) {
            std::printf("FAIL SORcloud: %zu of %zu kept\n", cloud.size(), before);
            return 1;
        }
    }
    // checkLoopDetectorStatus: 25 calls on the same image -- too young a database for a closure
    // (query - match > 100 can not hold), every frame is stored, flags stay down
    {
        for (int i = 0; i < 25; i++)
            s.checkLoopDetectorStatus(L, i);
        if (s.LC_FLAG || s.lastLoopResult.query != 24 || s.cooldownTimer != 0) {
            std::printf("FAIL checkLoopDetectorStatus: query %d status %d\n", s.lastLoopResult.query,
                        s.lastLoopResult.status);
            return 1;
        }
        if (s.lastLoopResult.status == SVO_LC_CLOSE_MATCHES_ONLY) {  // entries 0..3 are old enough to be queried
            std::printf("FAIL checkLoopDetectorStatus: no database query at frame 24\n");
            return 1;
        }
    }
    // pose graph adaptor: a square loop with drift closes
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
    if (est.size() != 20 || pg.numEdges() != 20) {
        std::printf("FAIL pose graph sizes\n");
        return 1;
    }
    // fused loop
    visualSLAM f;
    Mat33d Rm;
    Vec3d t;
    if (!f.processFrame(L, R, -1, Rm, t) || !f.processFrame(L, R, -1, Rm, t)) {
        std::printf("FAIL processFrame\n");
        return 1;
    }
    if (std::fabs(t(0)) + std::fabs(t(1)) + std::fabs(t(2)) > 0.02) {
        std::printf("FAIL static camera moved: %g %g %g\n", t(0), t(1), t(2));
        return 1;
    }
    StereoProcess sp;
    std::vector<Point3f> o3;
    sp.stereoTriangulate(L, R, o3);
    std::printf("compat smoke ok: %zu stereo points, depth %.2f m, pose graph %d vertices, static pose |t| = %.2e, "
                "StereoProcess %zu points\n",
                p3.size(), z_expect, pg.numVertices(), std::fabs(t(0)) + std::fabs(t(1)) + std::fabs(t(2)), o3.size());
    return 0;
}
