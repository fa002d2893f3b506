// svo_compat/stereoCV.hpp -- the one member of the reference's StereoProcess
// (include/stereoCV.h:59-71) that overlaps the hot path: stereoTriangulate(im1, im2, out3d).
// The rest of that class is a separate SGBM dense-stereo demo (src/StereoCV.cpp) publishing
// over ROS/PCL; it is outside the hot-path scope and is not provided.
#pragma once

#include "visualSLAM.hpp"

namespace svo_compat {

class StereoProcess {
  public:
    double baseline = 0.5707;  // include/stereoCV.h:40
    double focal_x = 7.188560000000e+02, cx = 6.071928000000e+02;
    double focal_y = 7.188560000000e+02, cy = 1.852157000000e+02;
    std::vector<Point3f> tri3dPoints, color3dMap;

    explicit StereoProcess(svo_ctx *ctx = nullptr) : slam_(ctx) {}

    // include/stereoCV.h:62.  The reference matches SIFT features here (src/StereoCV.cpp:64-121);
    // this adaptor uses the hot path's dense-grid LK + F-RANSAC + DLT triangulation instead and
    // says so: same output contract (camera-frame 3-D points, colours in color3dMap).
    void stereoTriangulate(const Mat &im1, const Mat &im2, std::vector<Point3f> &out3d)
    {
        slam_.baseline = baseline;
        slam_.focal_x = focal_x;
        slam_.focal_y = focal_y;
        slam_.cx = cx;
        slam_.cy = cy;
        std::vector<Point2f> pts2d;
        slam_.stereoTriangulate(im1, im2, out3d, pts2d);
        tri3dPoints = out3d;
        color3dMap = slam_.colors;
    }

  private:
    visualSLAM slam_;
};

}  // namespace svo_compat
