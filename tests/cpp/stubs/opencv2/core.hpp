// TEST STUB -- not OpenCV.  The smallest header that lets the SVO_WITH_OPENCV branch of
// include/svo_compat/ compile and run where OpenCV is not installed (this image): the types the
// reference's hot-path signatures use (include/visualSLAM.h:152-178), with OpenCV's public member
// names, type codes and -- for the three point types -- memory layout.  cv::Mat here owns or views a
// dense continuous buffer with cv::Mat's shallow-copy semantics; only what the adaptors touch exists.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

typedef unsigned char uchar;

#define CV_8U 0
#define CV_64F 6
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

namespace cv {

struct Point2f {
    float x, y;
    Point2f() : x(0), y(0) {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct Point3f {
    float x, y, z;
    Point3f() : x(0), y(0), z(0) {}
    Point3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_)
    {
    }
};

class Mat {
  public:
    int flags = 0, dims = 2, rows = 0, cols = 0;
    uchar *data = nullptr;

    Mat() = default;
    Mat(int r, int c, int type) : flags(type), rows(r), cols(c)
    {
        store_ = std::make_shared<std::vector<uchar>>((size_t)r * c * elemSize(), (uchar)0);
        data = store_->data();
    }
    Mat(int r, int c, int type, void *ptr) : flags(type), rows(r), cols(c), data(static_cast<uchar *>(ptr)) {}
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    int type() const { return flags & 0xFFF; }
    int depth() const { return flags & 7; }
    int channels() const { return ((flags & 0xFF8) >> CV_CN_SHIFT) + 1; }
    size_t elemSize() const { return (size_t)channels() * (depth() == CV_64F ? 8 : 1); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return true; }
    template <class T> T &at(int r, int c) { return *reinterpret_cast<T *>(data + ((size_t)r * cols + c) * elemSize()); }
    template <class T> const T &at(int r, int c) const
    {
        return *reinterpret_cast<const T *>(data + ((size_t)r * cols + c) * elemSize());
    }
    Mat clone() const
    {
        Mat m(rows, cols, type());
        if (data)
            std::memcpy(m.data, data, (size_t)rows * cols * elemSize());
        return m;
    }

  private:
    std::shared_ptr<std::vector<uchar>> store_;
};

}  // namespace cv
