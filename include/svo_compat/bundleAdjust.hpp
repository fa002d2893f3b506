// svo_compat/bundleAdjust.hpp -- the hot-path members of the reference's older class
// `visualOdometry` (src/bundleAdjust.cpp:30-614; the file is not in upstream's build, CMakeLists.txt
// lists it only in a comment) on top of the C ABI, with the reference's names and argument meaning.
#pragma once

#include "types.hpp"

namespace svo_compat {

class visualOdometry {
  public:
    int baIterations = 10;  // optimizer.optimize(10), src/bundleAdjust.cpp:606
    // what the last BundleAdjust3d2d did: chi2 before / after, final lambda, iterations, trials
    double lastInfo[5] = {0, 0, 0, 0, 0};

    explicit visualOdometry(svo_ctx *ctx = nullptr) : ctx_(ctx ? ctx : shared_context()) {}

    // src/bundleAdjust.cpp:551-613.  points by value as upstream; K 3x3, R 3x3, t 3x1, all CV_64F;
    // (R, t) map world points into the camera (what cv::solvePnP returns after Rodrigues).  Only t is
    // written back (:609-611).  Upstream builds g2o's CameraParameters from K(0,0), K(0,2), K(1,2):
    // K(1,1) is not read.
    void BundleAdjust3d2d(std::vector<Point2f> points_2d, std::vector<Point3f> points_3d, Mat &K, Mat &R, Mat &t)
    {
        if (points_2d.size() != points_3d.size() || points_2d.empty())
            throw std::invalid_argument("BundleAdjust3d2d: one 2-D point per 3-D point, at least one");
        const Mat33d Km = mat33_of(K), Rm = mat33_of(R);
        Vec3d tv = vec3_of(t);
        const double K4[4] = {Km(0, 0), Km(1, 1), Km(0, 2), Km(1, 2)};
        check(svo_ba_3d2d(ctx_, reinterpret_cast<const float *>(points_2d.data()),
                          reinterpret_cast<const float *>(points_3d.data()), (int)points_2d.size(), K4, Rm.m, tv.v,
                          baIterations, nullptr, nullptr, lastInfo, SVO_MEM_HOST));
        for (int i = 0; i < 3; i++)  // eigen2cv(trans, t): t keeps its 3x1 shape
            (t.rows == 3 ? t.at<double>(i, 0) : t.at<double>(0, i)) = tv(i);
    }

  private:
    svo_ctx *ctx_;
};

}  // namespace svo_compat
