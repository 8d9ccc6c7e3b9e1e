// NOT OpenCV.  A declaration-compatible subset of the OpenCV core API (cv::Mat, cv::KeyPoint,
// cv::InputArray, cv::OutputArray, CV_8UC1, CV_Assert) with just enough behaviour to push the
// ORBX_WITH_OPENCV branch of include/orbx_shim.hpp through a compiler and through one extraction on
// machines that have no OpenCV.  Member names, types and call signatures follow
// opencv2/core/mat.hpp and opencv2/core/types.hpp (3.x / 4.x); nothing here is used by the library
// or by the other tests.
#ifndef ORBX_TEST_OPENCV_API_SUBSET
#define ORBX_TEST_OPENCV_API_SUBSET
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#define CV_8U 0
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_Assert(expr) \
    do {                \
        if (!(expr)) throw std::runtime_error("CV_Assert failed: " #expr); \
    } while (0)

typedef unsigned char uchar;

namespace cv {

template <typename T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<float> Point2f;

class KeyPoint {
public:
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};

struct MatStep {
    size_t v;
    MatStep() : v(0) {}
    operator size_t() const { return v; }
    MatStep& operator=(size_t s) {
        v = s;
        return *this;
    }
};

class Mat {
public:
    Mat() : flags(0), rows(0), cols(0), data(nullptr) {}
    Mat(int r, int c, int type) : flags(0), rows(0), cols(0), data(nullptr) { create(r, c, type); }
    // borrowed memory, as cv::Mat(rows, cols, type, void* data, size_t step)
    Mat(int r, int c, int type, void* d, size_t s = 0) : flags(type), rows(r), cols(c), data((uchar*)d) {
        CV_Assert(type == CV_8UC1);
        step = s ? s : (size_t)c;
    }
    void create(int r, int c, int type) {
        CV_Assert(type == CV_8UC1);
        if (r == rows && c == cols && data && flags == type && owner_) return;
        owner_ = std::shared_ptr<uchar>((uchar*)std::malloc((size_t)r * c > 0 ? (size_t)r * c : 1), std::free);
        data = owner_.get();
        rows = r;
        cols = c;
        flags = type;
        step = (size_t)c;
    }
    void release() {
        owner_.reset();
        data = nullptr;
        rows = cols = 0;
    }
    int type() const { return flags; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    uchar* ptr(int r = 0) { return data + (size_t)r * step; }
    const uchar* ptr(int r = 0) const { return data + (size_t)r * step; }
    template <typename T>
    T& at(int r, int c) { return *(T*)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T>
    const T& at(int r, int c) const { return *(const T*)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    Mat row(int r) const {
        Mat m = *this;
        m.data = data + (size_t)r * step;
        m.rows = 1;
        return m;
    }

    int flags;
    int rows, cols;
    uchar* data;
    MatStep step;

private:
    std::shared_ptr<uchar> owner_;
};

class _InputArray {
public:
    _InputArray() : m_(nullptr) {}
    _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}
    bool empty() const { return !m_ || m_->empty(); }
    Mat getMat() const { return m_ ? *m_ : Mat(); }

protected:
    Mat* m_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray(Mat& m) { m_ = &m; }
    void create(int rows, int cols, int type) const { m_->create(rows, cols, type); }
    void release() const { m_->release(); }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv
#endif
