// svo_compat/types.hpp -- minimal stand-ins for the cv:: / Eigen:: types the reference's
// hot-path signatures use (include/visualSLAM.h:152-178, include/poseGraph.h:62-66), so the
// adaptors compile without OpenCV, Eigen, g2o, PCL or ROS.  Where the real libraries exist,
// define SVO_WITH_OPENCV / SVO_WITH_EIGEN before including and the aliases bind to them
// (same memory layout: cv::Point2f = {float x, y}, cv::Point3f = {float x, y, z},
// cv::Mat CV_8UC3 continuous, Eigen::Isometry3d = 4x4 double column-major).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../svo.h"

#if defined(SVO_WITH_OPENCV)
#include <opencv2/core.hpp>
#endif
#if defined(SVO_WITH_EIGEN)
#include <Eigen/Geometry>
#endif

namespace svo_compat {

#if defined(SVO_WITH_OPENCV)
using Point2f = cv::Point2f;
using Point3f = cv::Point3f;
using KeyPoint = cv::KeyPoint;
using Mat = cv::Mat;
#else
struct Point2f {
    float x = 0, y = 0;
    Point2f() = default;
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct Point3f {
    float x = 0, y = 0, z = 0;
    Point3f() = default;
    Point3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
// cv::KeyPoint(x, y, size): angle -1, response 0, octave 0 (what src/tracking.cpp:8 builds)
struct KeyPoint {
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
    KeyPoint() = default;
    KeyPoint(float x, float y, float s) : pt(x, y), size(s) {}
};
// cv::Mat's type codes for the two depths the hot path uses (depth + ((channels - 1) << 3))
constexpr int CV_8U = 0, CV_64F = 6, CV_8UC1 = 0, CV_8UC3 = 16;
// A dense continuous matrix with cv::Mat's shallow-copy semantics (by value = ref-counted header
// copy, no element copy): CV_8UC1 / CV_8UC3 images and the CV_64F matrices (K, R, t, rvec, tvec,
// [R|t]) the reference passes through its member functions as cv::Mat.
struct Mat {
    int rows = 0, cols = 0;
    uint8_t *data = nullptr;
    Mat() = default;
    Mat(int r, int c, int type) : rows(r), cols(c), type_(type)
    {
        store_ = std::make_shared<std::vector<uint8_t>>((size_t)r * c * elemSize(), (uint8_t)0);
        data = store_->data();
    }
    // non-owning view of caller memory (continuous: row stride = cols * elemSize())
    Mat(int r, int c, int type, void *ptr) : rows(r), cols(c), data(static_cast<uint8_t *>(ptr)), type_(type) {}
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elemSize() const { return (size_t)channels() * (depth() == CV_64F ? 8 : 1); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return true; }
    uint8_t *ptr() { return data; }
    template <class T> T &at(int r, int c) { return *reinterpret_cast<T *>(data + ((size_t)r * cols + c) * elemSize()); }
    template <class T> const T &at(int r, int c) const
    {
        return *reinterpret_cast<const T *>(data + ((size_t)r * cols + c) * elemSize());
    }
    Mat clone() const
    {
        Mat m(rows, cols, type_);
        if (data)
            std::memcpy(m.data, data, (size_t)rows * cols * elemSize());
        return m;
    }

  private:
    int type_ = CV_8UC1;
    std::shared_ptr<std::vector<uint8_t>> store_;
};
#endif
inline const uint8_t *mat_data(const Mat &m) { return m.data; }
inline int mat_rows(const Mat &m) { return m.rows; }
inline int mat_cols(const Mat &m) { return m.cols; }
inline int mat_channels(const Mat &m) { return m.channels(); }

// 3x3 / 3x1 / 3x4 double matrices of the reference (cv::Mat CV_64F there), row-major
struct Mat33d {
    double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double &operator()(int r, int c) { return m[3 * r + c]; }
    double operator()(int r, int c) const { return m[3 * r + c]; }
};
struct Vec3d {
    double v[3] = {0, 0, 0};
    double &operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
};
struct Mat34d {
    double m[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double &operator()(int r, int c) { return m[4 * r + c]; }
    double operator()(int r, int c) const { return m[4 * r + c]; }
    static Mat34d from(const Mat33d &R, const Vec3d &t)
    {
        Mat34d P;
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++)
                P(i, j) = R(i, j);
            P(i, 3) = t(i);
        }
        return P;
    }
};

#if defined(SVO_WITH_EIGEN)
using Isometry3d = Eigen::Isometry3d;
#else
// Eigen::Isometry3d stand-in: 4x4 double, column-major like Eigen, element access T(r, c)
struct Isometry3d {
    double m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    static Isometry3d Identity() { return Isometry3d(); }
    double &operator()(int r, int c) { return m[4 * c + r]; }
    double operator()(int r, int c) const { return m[4 * c + r]; }
};
#endif
// The conversions below touch an isometry through Identity() and T(r, c) only -- the part of
// Eigen::Transform's interface the stand-in shares -- so the SVO_WITH_EIGEN build runs the same code.
inline void iso_to_pose7(const Isometry3d &T, double *p)
{
    const double r00 = T(0, 0), r11 = T(1, 1), r22 = T(2, 2), tr = r00 + r11 + r22;
    double q[4];  // x y z w
    if (tr > 0) {
        const double s = std::sqrt(tr + 1.0) * 2;
        q[0] = (T(2, 1) - T(1, 2)) / s;
        q[1] = (T(0, 2) - T(2, 0)) / s;
        q[2] = (T(1, 0) - T(0, 1)) / s;
        q[3] = 0.25 * s;
    } else if (r00 > r11 && r00 > r22) {
        const double s = std::sqrt(1.0 + r00 - r11 - r22) * 2;
        q[0] = 0.25 * s;
        q[1] = (T(0, 1) + T(1, 0)) / s;
        q[2] = (T(0, 2) + T(2, 0)) / s;
        q[3] = (T(2, 1) - T(1, 2)) / s;
    } else if (r11 > r22) {
        const double s = std::sqrt(1.0 + r11 - r00 - r22) * 2;
        q[0] = (T(0, 1) + T(1, 0)) / s;
        q[1] = 0.25 * s;
        q[2] = (T(1, 2) + T(2, 1)) / s;
        q[3] = (T(0, 2) - T(2, 0)) / s;
    } else {
        const double s = std::sqrt(1.0 + r22 - r00 - r11) * 2;
        q[0] = (T(0, 2) + T(2, 0)) / s;
        q[1] = (T(1, 2) + T(2, 1)) / s;
        q[2] = 0.25 * s;
        q[3] = (T(1, 0) - T(0, 1)) / s;
    }
    const double sg = q[3] < 0 ? -1. : 1.;
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    p[0] = T(0, 3);
    p[1] = T(1, 3);
    p[2] = T(2, 3);
    for (int i = 0; i < 4; i++)
        p[3 + i] = sg * q[i] / n;
}
inline Isometry3d pose7_to_iso(const double *p)
{
    const double x = p[3], y = p[4], z = p[5], w = p[6];
    Isometry3d T = Isometry3d::Identity();
    T(0, 0) = 1 - 2 * (y * y + z * z);
    T(0, 1) = 2 * (x * y - z * w);
    T(0, 2) = 2 * (x * z + y * w);
    T(1, 0) = 2 * (x * y + z * w);
    T(1, 1) = 1 - 2 * (x * x + z * z);
    T(1, 2) = 2 * (y * z - x * w);
    T(2, 0) = 2 * (x * z - y * w);
    T(2, 1) = 2 * (y * z + x * w);
    T(2, 2) = 1 - 2 * (x * x + y * y);
    T(0, 3) = p[0];
    T(1, 3) = p[1];
    T(2, 3) = p[2];
    return T;
}
inline Isometry3d iso_from(const Mat33d &R, const Vec3d &t)
{
    Isometry3d T = Isometry3d::Identity();
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            T(i, j) = R(i, j);
        T(i, 3) = t(i);
    }
    return T;
}

// ---- cv::Mat (CV_64F) <-> the typed 3x3 / 3x1 / 3x4 holders (both Mat flavours offer at<double>) ----
inline Mat33d mat33_of(const Mat &m)
{
    if (m.rows != 3 || m.cols != 3 || m.type() != CV_64F)
        throw std::invalid_argument("expected a 3x3 CV_64F matrix");
    Mat33d R;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R(i, j) = m.at<double>(i, j);
    return R;
}
// 3x1 (tvec, rvec) or 1x3 (the reference's trajectory entries, include/monoUtils.h:109-127)
inline Vec3d vec3_of(const Mat &m)
{
    if (m.type() != CV_64F || !((m.rows == 3 && m.cols == 1) || (m.rows == 1 && m.cols == 3)))
        throw std::invalid_argument("expected a 3x1 or 1x3 CV_64F matrix");
    Vec3d v;
    for (int i = 0; i < 3; i++)
        v(i) = m.rows == 3 ? m.at<double>(i, 0) : m.at<double>(0, i);
    return v;
}
inline Mat34d mat34_of(const Mat &m)
{
    if (m.rows != 3 || m.cols != 4 || m.type() != CV_64F)
        throw std::invalid_argument("expected a 3x4 CV_64F matrix");
    Mat34d P;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++)
            P(i, j) = m.at<double>(i, j);
    return P;
}
inline Mat to_mat(const Mat33d &R)
{
    Mat m = Mat::zeros(3, 3, CV_64F);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            m.at<double>(i, j) = R(i, j);
    return m;
}
inline Mat to_mat(const Vec3d &v)  // 3x1, as solvePnPRansac returns rvec / tvec
{
    Mat m = Mat::zeros(3, 1, CV_64F);
    for (int i = 0; i < 3; i++)
        m.at<double>(i, 0) = v(i);
    return m;
}

// The reference never checks return codes (it has none); the adaptors turn a failing C-ABI
// call into an exception, the moral equivalent of the uncaught cv::Exception upstream.
struct SvoError : std::runtime_error {
    int code;
    SvoError(int c, const char *what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc)
{
    if (rc != SVO_OK)
        throw SvoError(rc, svo_last_error());
}

// one context per device, shared by the adaptors of a process
inline svo_ctx *shared_context(int device = 0)
{
    static svo_ctx *ctx = nullptr;
    static int dev = -1;
    if (!ctx) {
        check(svo_ctx_create(device, &ctx));
        dev = device;
    }
    (void)dev;
    return ctx;
}

}  // namespace svo_compat
