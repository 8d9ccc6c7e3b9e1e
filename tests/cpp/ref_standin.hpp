// Stand-ins for the types the reference's host code is written against, with the reference's
// member names (include/Frame.h, include/MapPoint.h) and the small part of cv:: they use, so that
// call sites of Tracking.cc compile and run in an image without OpenCV.  Test infrastructure.
#ifndef REF_STANDIN_HPP
#define REF_STANDIN_HPP

#include <cstring>
#include <vector>

namespace cv {

struct Point2f {
    float x, y;
};

// cv::KeyPoint's 28-byte layout
struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};

// A dense 2-D matrix of bytes or floats with the accessors the matcher adapters use.
class Mat {
public:
    int rows = 0, cols = 0;
    Mat() {}
    Mat(int r, int c, int elemSize) : rows(r), cols(c), esz_(elemSize), buf_((size_t)r * c * elemSize, 0) {}
    bool isContinuous() const { return true; }
    bool empty() const { return rows == 0 || cols == 0; }
    template <class T>
    T& at(int r, int c) {
        return *reinterpret_cast<T*>(&buf_[((size_t)r * cols + c) * esz_]);
    }
    template <class T>
    const T& at(int r, int c) const {
        return *reinterpret_cast<const T*>(&buf_[((size_t)r * cols + c) * esz_]);
    }
    template <class T>
    const T* ptr(int r = 0) const {
        return reinterpret_cast<const T*>(&buf_[(size_t)r * cols * esz_]);
    }
    template <class T>
    T* ptr(int r = 0) {
        return reinterpret_cast<T*>(&buf_[(size_t)r * cols * esz_]);
    }
    Mat row(int r) const {
        Mat m(1, cols, esz_);
        std::memcpy(m.buf_.data(), &buf_[(size_t)r * cols * esz_], (size_t)cols * esz_);
        return m;
    }
    Mat clone() const { return *this; }

private:
    int esz_ = 1;
    std::vector<unsigned char> buf_;
};

}  // namespace cv

namespace ORB_SLAM2 {

class MapPoint {
public:
    long unsigned int mnId = 0;
    // set by Frame::isInFrustum (reference src/Frame.cc:369-422)
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 0;
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    int Observations() { return nObs; }
    bool isBad() { return mbBad; }
    cv::Mat mWorldPos = cv::Mat(3, 1, 4);
    cv::Mat mDescriptor = cv::Mat(1, 32, 1);
    int nObs = 0;
    bool mbBad = false;
};

class Frame {
public:
    static float fx, fy, cx, cy;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    float mbf = 0, mb = 0;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    cv::Mat mTcw = cv::Mat(4, 4, 4);
    long unsigned int mnId = 0;
    std::vector<float> mvScaleFactors;
};
float Frame::fx, Frame::fy, Frame::cx, Frame::cy;
float Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY;

}  // namespace ORB_SLAM2

#endif
