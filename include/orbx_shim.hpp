// Header-only C++ drop-in for the two reference classes, on top of the orbx C ABI.
//
//   ORB_SLAM2::ORBextractor  (reference include/ORBextractor.h:32-105)
//   ORB_SLAM2::ORBmatcher    (reference include/ORBmatcher.h:25-93; the entry points
//                             that exist in the C ABI)
//
// With OpenCV available, define ORBX_WITH_OPENCV before including this header: the
// extractor then takes cv::InputArray / std::vector<cv::KeyPoint> / cv::OutputArray exactly
// like the reference.  (That branch needs OpenCV headers; it has not been through a compiler in
// an image without OpenCV, only the stand-in branch has.)  Without OpenCV the same classes are
// expressed on two small stand-in types (orbx::KeyPoint, orbx::Image).
//
// The matcher offers the reference's own signatures
//     SearchByProjection(Frame&, const Frame&, th, bMono)              src/ORBmatcher.cc:1533
//     SearchByProjection(Frame&, const std::vector<MapPoint*>&, th)    :62
//     SearchForInitialization(Frame&, Frame&, vbPrevMatched, vnMatches12, windowSize)   :501
//     DescriptorDistance(const Mat&, const Mat&)                       :1905
// as templates on the Frame / MapPoint / Mat types, so that the call sites of Tracking.cc
// (:944-950, :1333, :1339, :1943) compile as they stand, against the reference's classes or
// against any types with the same member names (tests/cpp/ref_standin.hpp).  What a type must
// offer is listed above each template.  mvImagePyramid is a member of the extractor that
// fetches a level to the host the first time it is indexed after an extraction.
//
// One extractor object owns one orbx handle (one HIP stream).  As in the reference, an
// object must not be used from two threads at once; two objects may run concurrently
// (stereo: reference src/Frame.cc:122-125).
#ifndef ORBX_SHIM_HPP
#define ORBX_SHIM_HPP

#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "orbx.h"

#ifdef ORBX_WITH_OPENCV
#include <opencv2/core/core.hpp>
#endif

namespace orbx {

// Layout-identical to cv::KeyPoint.
struct KeyPoint {
    float x, y, size, angle, response;
    int octave, class_id;
};
static_assert(sizeof(KeyPoint) == sizeof(orbx_keypoint), "KeyPoint must be 28 bytes");

// A borrowed 8-bit single-channel image.
struct Image {
    const unsigned char* data;
    int cols, rows;
    size_t step;
    bool empty() const { return !data || cols <= 0 || rows <= 0; }
};

inline void check(int rc, const char* what) {
    if (rc != ORBX_OK) throw std::runtime_error(std::string(what) + ": " + orbx_last_error());
}

}  // namespace orbx

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // max_width/max_height size the device scratch; larger images are rejected.
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                 int max_width = 1280, int max_height = 1024, int device = 0) {
        orbx_config cfg;
        std::memset(&cfg, 0, sizeof(cfg));
        cfg.nfeatures = nfeatures;
        cfg.scale_factor = scaleFactor;
        cfg.nlevels = nlevels;
        cfg.ini_th_fast = iniThFAST;
        cfg.min_th_fast = minThFAST;
        cfg.max_width = max_width;
        cfg.max_height = max_height;
        cfg.max_batch = 1;
        cfg.device = device;
        orbx::check(orbx_create(&cfg, &h_), "orbx_create");
        cap_ = orbx_max_keypoints(h_);
        kps_.resize(cap_);
        desc_.resize((size_t)cap_ * 32);
    }
    ~ORBextractor() { orbx_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // Mask is ignored, as in the reference (include/ORBextractor.h:45).
    void operator()(const orbx::Image& image, const orbx::Image& /*mask*/, std::vector<orbx::KeyPoint>& keypoints,
                    std::vector<unsigned char>& descriptors) {
        keypoints.clear();
        descriptors.clear();
        if (image.empty()) return;  // reference src/ORBextractor.cc:1055-1056
        int n = 0;
        orbx::check(orbx_extract(h_, image.data, image.cols, image.rows, image.step, kps_.data(), desc_.data(), cap_, &n),
                    "orbx_extract");
        keypoints.resize(n);
        if (n) std::memcpy(keypoints.data(), kps_.data(), (size_t)n * sizeof(orbx_keypoint));
        descriptors.assign(desc_.begin(), desc_.begin() + (size_t)n * 32);
        hostValid_.assign(hostValid_.size(), 0);
    }

#ifdef ORBX_WITH_OPENCV
    void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& _keypoints,
                    cv::OutputArray _descriptors) {
        if (_image.empty()) return;
        cv::Mat image = _image.getMat();
        CV_Assert(image.type() == CV_8UC1);
        int n = 0;
        orbx::check(orbx_extract(h_, image.data, image.cols, image.rows, image.step, kps_.data(), desc_.data(), cap_, &n),
                    "orbx_extract");
        hostValid_.assign(hostValid_.size(), 0);
        static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint layout");
        _keypoints.resize(n);
        if (n) std::memcpy((void*)_keypoints.data(), kps_.data(), (size_t)n * sizeof(orbx_keypoint));
        if (n == 0) {
            _descriptors.release();
        } else {
            _descriptors.create(n, 32, CV_8U);
            cv::Mat d = _descriptors.getMat();
            for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), desc_.data() + (size_t)i * 32, 32);
        }
    }
#endif

    // ---- mvImagePyramid (reference include/ORBextractor.h:73; read by Frame::ComputeStereoMatches,
    // src/Frame.cc:610, :700, :711, :716).  The levels live in device memory; indexing the member
    // copies that level to the host once per extraction and hands back a host image (cv::Mat with
    // OpenCV, orbx::HostImage without): `mvImagePyramid[l].rows`, `.rowRange(..).colRange(..)` and
    // `mvImagePyramid.size()` read as in the reference.
#ifdef ORBX_WITH_OPENCV
    typedef cv::Mat HostLevel;
#else
    struct HostLevel {
        std::vector<unsigned char> pixels;
        int rows = 0, cols = 0;
        size_t step = 0;
        unsigned char* data = nullptr;
        const unsigned char* ptr(int r) const { return data + (size_t)r * step; }
        unsigned char at(int r, int c) const { return data[(size_t)r * step + c]; }
        bool empty() const { return rows == 0 || cols == 0; }
    };
#endif
    class ImagePyramid {
    public:
        explicit ImagePyramid(ORBextractor* e) : e_(e) {}
        size_t size() const { return (size_t)orbx_levels(e_->h_); }
        const HostLevel& operator[](size_t level) const { return e_->host_level((int)level); }
    private:
        ORBextractor* e_;
    };
    ImagePyramid mvImagePyramid{this};

    int GetLevels() { return orbx_levels(h_); }
    float GetScaleFactor() { return orbx_scale_factor(h_); }
    std::vector<float> GetScaleFactors() { return vec(orbx_scale_factors(h_)); }
    std::vector<float> GetInverseScaleFactors() { return vec(orbx_inv_scale_factors(h_)); }
    std::vector<float> GetScaleSigmaSquares() { return vec(orbx_level_sigma2(h_)); }
    std::vector<float> GetInverseScaleSigmaSquares() { return vec(orbx_inv_level_sigma2(h_)); }

    // mvImagePyramid[level] (reference include/ORBextractor.h:73) as a device pointer.
    const unsigned char* PyramidLevelDevice(int level, int* w, int* h, size_t* stride) {
        const uint8_t* p = nullptr;
        orbx::check(orbx_get_level(h_, 0, level, &p, w, h, stride), "orbx_get_level");
        return p;
    }

    orbx_handle* handle() { return h_; }

private:
    std::vector<float> vec(const float* p) { return std::vector<float>(p, p + orbx_levels(h_)); }
    const HostLevel& host_level(int level) {
        const int nl = orbx_levels(h_);
        if (level < 0 || level >= nl) throw std::out_of_range("mvImagePyramid: level out of range");
        if ((int)hostLevels_.size() != nl) {
            hostLevels_.assign(nl, HostLevel());
            hostValid_.assign(nl, 0);
        }
        if (!hostValid_[level]) {
            const uint8_t* p = nullptr;
            int w = 0, h = 0;
            orbx::check(orbx_get_level(h_, 0, level, &p, &w, &h, nullptr), "orbx_get_level");
            HostLevel& L = hostLevels_[level];
#ifdef ORBX_WITH_OPENCV
            L.create(h, w, CV_8UC1);
            orbx::check(orbx_debug_level(h_, 0, level, L.data), "orbx_debug_level");  // tightly packed rows
#else
            L.pixels.resize((size_t)w * h);
            L.rows = h;
            L.cols = w;
            L.step = (size_t)w;
            L.data = L.pixels.data();
            orbx::check(orbx_debug_level(h_, 0, level, L.data), "orbx_debug_level");
#endif
            hostValid_[level] = 1;
        }
        return hostLevels_[level];
    }
    std::vector<HostLevel> hostLevels_;
    std::vector<char> hostValid_;
    orbx_handle* h_ = nullptr;
    int cap_ = 0;
    std::vector<orbx_keypoint> kps_;
    std::vector<unsigned char> desc_;
};

// What the matcher reads of a Frame (reference include/Frame.h): mvKeysUn, mDescriptors,
// mnMinX.., mvScaleFactors, mvuRight.
struct FrameView {
    const orbx::KeyPoint* keysUn;
    const unsigned char* descriptors;
    int N;
    float mnMinX, mnMinY, mnMaxX, mnMaxY;
    const float* scaleFactors;
    int nLevels;
    const float* uRight;  // N entries; negative = monocular point
    orbm_frame c() const {
        orbm_frame f;
        f.kps = reinterpret_cast<const orbx_keypoint*>(keysUn);
        f.desc = descriptors;
        f.n = N;
        f.min_x = mnMinX;
        f.min_y = mnMinY;
        f.max_x = mnMaxX;
        f.max_y = mnMaxY;
        return f;
    }
};

class ORBmatcher {
public:
    static const int TH_LOW = 50;        // reference src/ORBmatcher.cc:39
    static const int TH_HIGH = 100;      // :38
    static const int HISTO_LENGTH = 30;  // :40

    ORBmatcher(float nnratio = 0.6f, bool checkOri = true, int device = 0) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {
        orbx::check(orbm_create(device, &m_), "orbm_create");
    }
    ~ORBmatcher() { orbm_destroy(m_); }
    ORBmatcher(const ORBmatcher&) = delete;
    ORBmatcher& operator=(const ORBmatcher&) = delete;

    // reference src/ORBmatcher.cc:1905-1921 (descriptor rows of 32 bytes)
    static int DescriptorDistance(const unsigned char* a, const unsigned char* b) { return orbm_hamming256(a, b); }


    // ================= the reference's own signatures =================
    // Requirements on the types (all met by the reference's classes and by cv::):
    //   Mat:      rows, cols, isContinuous(), template at<T>(r, c), template ptr<T>(r)
    //   KeyPoint: the 28-byte layout of cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id)
    //   Frame:    N, mvKeys, mvKeysUn, mvuRight, mDescriptors, mvpMapPoints, mvbOutlier, mTcw, fx, fy, cx,
    //             cy, mb, mbf, mnMinX, mnMaxX, mnMinY, mnMaxY, mvScaleFactors   (include/Frame.h)
    //   MapPoint: GetWorldPos(), GetDescriptor(), Observations(), isBad(), mbTrackInView,
    //             mnTrackScaleLevel, mTrackViewCos, mTrackProjX, mTrackProjY, mTrackProjXR   (include/MapPoint.h)

    // reference src/ORBmatcher.cc:1905-1921: two descriptor rows of 32 bytes
    template <class MatT>
    static int DescriptorDistance(const MatT& a, const MatT& b) {
        return orbm_hamming256(a.template ptr<unsigned char>(0), b.template ptr<unsigned char>(0));
    }

    // reference src/ORBmatcher.cc:1533-1678.  The pose algebra of :1545-1556 and the projection
    // of :1570-1578 run here on the host in the reference's order of float operations (a 3x3 by
    // 3x1 cv::Mat product is the float sum a0*b0 + a1*b1 + a2*b2, left to right; the addition of
    // the translation rounds once more); the window search, the claims on
    // CurrentFrame.mvpMapPoints and the rotation histogram run on the GPU.
    template <class FrameT>
    int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono) {
        float Rcw[9], tcw[3], Rlw[9], tlw[3];
        pose(CurrentFrame.mTcw, Rcw, tcw);
        pose(LastFrame.mTcw, Rlw, tlw);
        // twc = -Rcw.t() * tcw ; tlc = Rlw * twc + tlw   (:1547, :1551)
        float twc[3], tlc[3];
        for (int i = 0; i < 3; i++) twc[i] = -(Rcw[0 + i] * tcw[0] + Rcw[3 + i] * tcw[1] + Rcw[6 + i] * tcw[2]);
        for (int i = 0; i < 3; i++) tlc[i] = (Rlw[3 * i] * twc[0] + Rlw[3 * i + 1] * twc[1] + Rlw[3 * i + 2] * twc[2]) + tlw[i];
        const bool bForward = tlc[2] > CurrentFrame.mb && !bMono;
        const bool bBackward = -tlc[2] > CurrentFrame.mb && !bMono;
        const int nLast = LastFrame.N, nCur = CurrentFrame.N;
        std::vector<unsigned char> valid(nLast, 0), hasObs(nLast, 0), mpDesc((size_t)nLast * 32, 0);
        std::vector<float> u(nLast, 0.f), v(nLast, 0.f), invz(nLast, 0.f);
        for (int i = 0; i < nLast; i++) {
            auto* pMP = LastFrame.mvpMapPoints[i];
            if (!pMP || LastFrame.mvbOutlier[i]) continue;
            const auto x3Dw = pMP->GetWorldPos();
            const float xw = x3Dw.template at<float>(0, 0), yw = x3Dw.template at<float>(1, 0), zw = x3Dw.template at<float>(2, 0);
            const float xc = (Rcw[0] * xw + Rcw[1] * yw + Rcw[2] * zw) + tcw[0];
            const float yc = (Rcw[3] * xw + Rcw[4] * yw + Rcw[5] * zw) + tcw[1];
            const float zc = (Rcw[6] * xw + Rcw[7] * yw + Rcw[8] * zw) + tcw[2];
            const float invzc = 1.0 / zc;  // a double division rounded to float, as written at :1575
            valid[i] = 1;
            invz[i] = invzc;
            u[i] = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
            v[i] = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
            hasObs[i] = pMP->Observations() > 0 ? 1 : 0;
            const auto d = pMP->GetDescriptor();
            std::memcpy(&mpDesc[(size_t)i * 32], d.template ptr<unsigned char>(0), 32);
        }
        std::vector<int> claimed;
        std::vector<unsigned char> claimedObs;
        claims_in(CurrentFrame, claimed, claimedObs);
        std::vector<unsigned char> curDesc, lastDesc;
        orbm_frame c = frame_of(CurrentFrame, curDesc), l = frame_of(LastFrame, lastDesc);
        int n = 0;
        orbx::check(orbm_search_by_projection(m_, &c, &l, mpDesc.data(), valid.data(), u.data(), v.data(), invz.data(),
                                              hasObs.data(), CurrentFrame.mvuRight.data(), CurrentFrame.mvScaleFactors.data(),
                                              (int)CurrentFrame.mvScaleFactors.size(), CurrentFrame.mbf, th,
                                              bForward ? 1 : (bBackward ? 2 : 0), mbCheckOrientation, claimed.data(),
                                              claimedObs.data(), nullptr, &n, ORBX_MEM_HOST),
                    "orbm_search_by_projection");
        for (int i2 = 0; i2 < nCur; i2++) {
            if (claimed[i2] >= kPrior) continue;  // untouched
            CurrentFrame.mvpMapPoints[i2] = claimed[i2] >= 0 ? LastFrame.mvpMapPoints[claimed[i2]] : nullptr;
        }
        return n;
    }

    // reference src/ORBmatcher.cc:62-155 (Tracking::SearchLocalPoints, src/Tracking.cc:1943)
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th = 3) {
        const int nq = (int)vpMapPoints.size();
        std::vector<unsigned char> valid(nq, 0), hasObs(nq, 0), desc((size_t)nq * 32, 0);
        std::vector<float> px(nq, 0.f), py(nq, 0.f), pxr(nq, 0.f), radius(nq, 0.f);
        std::vector<int> minLevel(nq, 0), maxLevel(nq, 0);
        const bool bFactor = th != 1.0;
        for (int i = 0; i < nq; i++) {
            MapPointT* pMP = vpMapPoints[i];
            if (!pMP->mbTrackInView) continue;
            if (pMP->isBad()) continue;
            const int nPredictedLevel = pMP->mnTrackScaleLevel;
            float r = pMP->mTrackViewCos > 0.998 ? 2.5 : 4.0;  // RadiusByViewingCos, :158-164
            if (bFactor) r *= th;
            valid[i] = 1;
            px[i] = pMP->mTrackProjX;
            py[i] = pMP->mTrackProjY;
            pxr[i] = pMP->mTrackProjXR;
            radius[i] = r * F.mvScaleFactors[nPredictedLevel];
            minLevel[i] = nPredictedLevel - 1;
            maxLevel[i] = nPredictedLevel;
            hasObs[i] = pMP->Observations() > 0 ? 1 : 0;
            const auto d = pMP->GetDescriptor();
            std::memcpy(&desc[(size_t)i * 32], d.template ptr<unsigned char>(0), 32);
        }
        orbm_proj_queries q;
        std::memset(&q, 0, sizeof(q));
        q.n = nq;
        q.valid = valid.data();
        q.u = px.data();
        q.v = py.data();
        q.radius = radius.data();
        q.min_level = minLevel.data();
        q.max_level = maxLevel.data();
        q.desc = desc.data();
        q.has_obs = hasObs.data();
        q.u_right = pxr.data();
        q.right_gate = radius.data();
        const orbm_proj_policy pol = {0, TH_HIGH, 1, mfNNratio, 0};
        std::vector<int> claimed;
        std::vector<unsigned char> claimedObs, fDesc;
        claims_in(F, claimed, claimedObs);
        orbm_frame f = frame_of(F, fDesc);
        int n = 0;
        orbx::check(orbm_search_projected(m_, &f, F.mvuRight.data(), &q, &pol, claimed.data(), claimedObs.data(), nullptr, &n,
                                          ORBX_MEM_HOST),
                    "orbm_search_projected");
        for (int i2 = 0; i2 < F.N; i2++)
            if (claimed[i2] >= 0 && claimed[i2] < kPrior) F.mvpMapPoints[i2] = vpMapPoints[claimed[i2]];
        return n;
    }

    // reference src/ORBmatcher.cc:501-628 (Tracking::MonocularInitialization, src/Tracking.cc:944-950)
    template <class FrameT, class Point2fT>
    int SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<Point2fT>& vbPrevMatched, std::vector<int>& vnMatches12,
                                int windowSize = 10) {
        static_assert(sizeof(Point2fT) == 8, "Point2f must be two floats");
        const int n1 = (int)F1.mvKeysUn.size();
        vnMatches12.assign(n1, -1);
        if (n1 == 0) return 0;
        std::vector<unsigned char> d1, d2;
        orbm_frame f1 = frame_of(F1, d1), f2 = frame_of(F2, d2);
        int n = 0;
        orbx::check(orbm_search_for_initialization(m_, &f1, &f2, reinterpret_cast<float*>(vbPrevMatched.data()),
                                                   vnMatches12.data(), windowSize, mfNNratio, mbCheckOrientation, &n,
                                                   ORBX_MEM_HOST),
                    "orbm_search_for_initialization");
        return n;
    }

    // ================= the same searches on plain arrays =================
    // reference src/ORBmatcher.cc:501-628
    int SearchForInitialization(const FrameView& F1, const FrameView& F2, std::vector<float>& vbPrevMatchedXY,
                                std::vector<int>& vnMatches12, int windowSize = 10) {
        vnMatches12.assign(F1.N, -1);
        if (F1.N == 0) return 0;
        int n = 0;
        orbm_frame f1 = F1.c(), f2 = F2.c();
        orbx::check(orbm_search_for_initialization(m_, &f1, &f2, vbPrevMatchedXY.data(), vnMatches12.data(), windowSize,
                                                   mfNNratio, mbCheckOrientation, &n, ORBX_MEM_HOST),
                    "orbm_search_for_initialization");
        return n;
    }

    // reference src/ORBmatcher.cc:1533-1678.  The pose algebra (projection of each last-frame
    // map point, :1545-1578) stays with the caller: valid/u/v/invz/hasObs are per last-frame
    // keypoint, mpDesc holds pMP->GetDescriptor() per last-frame keypoint (N x 32, or null to
    // use the last frame's own descriptors), direction is 0 (neither), 1 (forward) or
    // 2 (backward).  claimed[i2] plays the role of CurrentFrame.mvpMapPoints[i2] (index of the
    // last-frame point, or -1).  queryBest (optional) receives, per last-frame keypoint, the
    // current keypoint it was assigned before the orientation cull: the matched point lists of
    // perfect/src/ORBmatcher.cc:1835-1836 are its non-negative entries in order.
    int SearchByProjection(const FrameView& CurrentFrame, const FrameView& LastFrame, float th, const unsigned char* mpDesc,
                           const unsigned char* valid, const float* u, const float* v, const float* invz,
                           const unsigned char* hasObs, float bf, int direction, std::vector<int>& claimed,
                           std::vector<unsigned char>& claimedObs, std::vector<int>* queryBest = nullptr) {
        claimed.resize(CurrentFrame.N, -1);
        claimedObs.resize(CurrentFrame.N, 0);
        if (queryBest) queryBest->assign(LastFrame.N, -1);
        int n = 0;
        orbm_frame c = CurrentFrame.c(), l = LastFrame.c();
        orbx::check(orbm_search_by_projection(m_, &c, &l, mpDesc, valid, u, v, invz, hasObs, CurrentFrame.uRight,
                                              CurrentFrame.scaleFactors, CurrentFrame.nLevels, bf, th, direction,
                                              mbCheckOrientation, claimed.data(), claimedObs.data(),
                                              queryBest ? queryBest->data() : nullptr, &n, ORBX_MEM_HOST),
                    "orbm_search_by_projection");
        return n;
    }

    // The other projection searches (map points :62-155, keyframe :1694-1836, Sim3 :370-485)
    // are orbm_search_projected with the query/policy rows listed in orbx.h.
    int SearchProjected(const FrameView& train, const orbm_proj_queries& q, const orbm_proj_policy& p, std::vector<int>& claimed,
                        std::vector<unsigned char>& claimedObs, std::vector<int>* queryBest = nullptr) {
        claimed.resize(train.N, -1);
        claimedObs.resize(train.N, 0);
        if (queryBest) queryBest->assign(q.n, -1);
        int n = 0;
        orbm_frame t = train.c();
        orbx::check(orbm_search_projected(m_, &t, train.uRight, &q, &p, claimed.data(), claimedObs.data(),
                                          queryBest ? queryBest->data() : nullptr, &n, ORBX_MEM_HOST),
                    "orbm_search_projected");
        return n;
    }

    // Frame::ComputeStereoMatches (reference src/Frame.cc:603-774).  left/right are the two
    // extractors right after they produced keys/keysRight (their device pyramids are read in
    // place); mvuRight and mvDepth come back with N entries, -1 where there is no stereo point.
    int ComputeStereoMatches(ORBextractor& left, ORBextractor& right, const orbx::KeyPoint* keys,
                             const unsigned char* descriptors, int N, const orbx::KeyPoint* keysRight,
                             const unsigned char* descriptorsRight, int Nr, float mb, float mbf, std::vector<float>& mvuRight,
                             std::vector<float>& mvDepth) {
        mvuRight.assign(N, -1.0f);
        mvDepth.assign(N, -1.0f);
        if (N == 0) return 0;
        orbm_pyramid pl = pyramid(left), pr = pyramid(right);
        orbm_frame fl = {reinterpret_cast<const orbx_keypoint*>(keys), descriptors, N, 0, 0, 1, 1};
        orbm_frame fr = {reinterpret_cast<const orbx_keypoint*>(keysRight), descriptorsRight, Nr, 0, 0, 1, 1};
        int n = 0;
        orbx::check(orbm_compute_stereo_matches(m_, &fl, &fr, &pl, &pr, orbx_scale_factors(left.handle()),
                                                orbx_inv_scale_factors(left.handle()), mb, mbf, mvuRight.data(),
                                                mvDepth.data(), &n, ORBX_MEM_HOST),
                    "orbm_compute_stereo_matches");
        return n;
    }

    // A DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned int>>) flattened for the C ABI.
    struct FlatFeatureVector {
        std::vector<uint32_t> ids, items;
        std::vector<int32_t> start;
        orbm_feature_vector c() const {
            orbm_feature_vector f;
            f.n_nodes = (int32_t)ids.size();
            f.n_items = (int32_t)items.size();
            f.node_ids = ids.data();
            f.start = start.data();
            f.items = items.data();
            return f;
        }
    };
    template <typename MapT>
    static FlatFeatureVector Flatten(const MapT& fv) {
        FlatFeatureVector o;
        o.start.push_back(0);
        for (typename MapT::const_iterator it = fv.begin(); it != fv.end(); ++it) {
            o.ids.push_back((uint32_t)it->first);
            for (size_t k = 0; k < it->second.size(); k++) o.items.push_back((uint32_t)it->second[k]);
            o.start.push_back((int32_t)o.items.size());
        }
        return o;
    }

    // reference src/ORBmatcher.cc:209-355 (keyframePair = false; F2 is the frame) and :642-788
    // (keyframePair = true; valid2 required).  valid1[i]: keypoint i of the first keyframe holds a
    // good map point.  match12[i1] = keypoint of the second view or -1; match21 is its inverse
    // (vpMapPointMatches of the first overload, by index).
    template <typename MapT>
    int SearchByBoW(const FrameView& KF1, const MapT& featVec1, const unsigned char* valid1, const FrameView& F2,
                    const MapT& featVec2, const unsigned char* valid2, bool keyframePair, std::vector<int>& match12,
                    std::vector<int>& match21) {
        match12.assign(KF1.N, -1);
        match21.assign(F2.N, -1);
        const FlatFeatureVector a = Flatten(featVec1), b = Flatten(featVec2);
        const orbm_feature_vector ca = a.c(), cb = b.c();
        orbm_frame f1 = KF1.c(), f2 = F2.c();
        int n = 0;
        orbx::check(orbm_search_by_bow(m_, &f1, &ca, valid1, &f2, &cb, valid2, keyframePair ? TH_LOW - 1 : TH_LOW, mfNNratio,
                                       mbCheckOrientation, match12.data(), match21.data(), &n, ORBX_MEM_HOST),
                    "orbm_search_by_bow");
        return n;
    }

    // reference src/ORBmatcher.cc:803-992; ex, ey and F12 (row-major 3x3) from :809-818 and the caller.
    template <typename MapT>
    int SearchForTriangulation(const FrameView& KF1, const MapT& featVec1, const unsigned char* hasMapPoint1,
                               const FrameView& KF2, const MapT& featVec2, const unsigned char* hasMapPoint2, const float* F12,
                               float ex, float ey, const float* levelSigma2_2, bool bOnlyStereo,
                               std::vector<std::pair<size_t, size_t> >& vMatchedPairs) {
        std::vector<int> m12(KF1.N, -1);
        const FlatFeatureVector a = Flatten(featVec1), b = Flatten(featVec2);
        const orbm_feature_vector ca = a.c(), cb = b.c();
        orbm_frame f1 = KF1.c(), f2 = KF2.c();
        int n = 0;
        orbx::check(orbm_search_for_triangulation(m_, &f1, &ca, hasMapPoint1, KF1.uRight, &f2, &cb, hasMapPoint2, KF2.uRight, F12,
                                                  ex, ey, KF2.scaleFactors, levelSigma2_2, KF2.nLevels, bOnlyStereo,
                                                  mbCheckOrientation, m12.data(), &n, ORBX_MEM_HOST),
                    "orbm_search_for_triangulation");
        vMatchedPairs.clear();
        for (size_t i = 0; i < m12.size(); i++)
            if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
        return n;
    }

    // Candidate loop of Fuse and SearchBySim3 (reference src/ORBmatcher.cc:1066-1120, :1391-1416):
    // bestIdx / bestDist per projected map point; the caller applies TH_LOW / TH_HIGH and the
    // map point bookkeeping of :1123-1148.
    void WindowBest(const FrameView& KF, int n, const unsigned char* valid, const float* u, const float* v, const float* radius,
                    const int* level, const unsigned char* mpDesc, bool chi2Gate, const float* qURight,
                    const float* invLevelSigma2, std::vector<int>& bestIdx, std::vector<int>& bestDist) {
        bestIdx.assign(n, -1);
        bestDist.assign(n, 256);
        if (n == 0) return;
        orbm_frame f = KF.c();
        orbx::check(orbm_window_best(m_, &f, n, valid, u, v, radius, level, mpDesc, chi2Gate, KF.uRight, qURight, invLevelSigma2,
                                     KF.nLevels, bestIdx.data(), bestDist.data(), ORBX_MEM_HOST),
                    "orbm_window_best");
    }

    // ORBVocabulary::transform(vCurrentDesc, mBowVec, mFeatVec, levelsup) as Frame::ComputeBoW calls it
    // (reference src/Frame.cc:513-522), with the vocabulary tree given as arrays.  The descent runs
    // on the GPU; the containers (DBoW2::BowVector and DBoW2::FeatureVector are std::maps) are filled
    // here with DBoW2's arithmetic for the TF-IDF / L1 vocabulary: weights of a word are added in
    // feature order, stopped words (weight 0) are left out, the vector is divided by its L1 norm
    // summed in word order.
    template <class BowVectorT, class FeatureVectorT>
    void Transform(const orbm_vocabulary& voc, const unsigned char* descriptors, int n, BowVectorT& v, FeatureVectorT& fv,
                   int levelsup) {
        v.clear();
        fv.clear();
        if (n <= 0) return;
        std::vector<int> word(n), node(n);
        std::vector<double> weight(n);
        orbx::check(orbm_bow_transform(m_, &voc, descriptors, n, levelsup, word.data(), weight.data(), node.data(), ORBX_MEM_HOST),
                    "orbm_bow_transform");
        typedef typename BowVectorT::key_type WordId;
        typedef typename FeatureVectorT::key_type NodeId;
        for (int i = 0; i < n; i++) {
            if (!(weight[i] > 0)) continue;
            const WordId id = (WordId)word[i];
            typename BowVectorT::iterator vit = v.lower_bound(id);
            if (vit != v.end() && !(v.key_comp()(id, vit->first)))
                vit->second += weight[i];
            else
                v.insert(vit, typename BowVectorT::value_type(id, weight[i]));
            fv[(NodeId)node[i]].push_back((unsigned int)i);
        }
        double norm = 0.0;
        for (typename BowVectorT::iterator it = v.begin(); it != v.end(); ++it) norm += it->second < 0 ? -it->second : it->second;
        if (norm > 0.0)
            for (typename BowVectorT::iterator it = v.begin(); it != v.end(); ++it) it->second /= norm;
    }

    // MapPoint::ComputeDistinctiveDescriptors for many map points (reference src/MapPoint.cc:279-341).
    void ComputeDistinctiveDescriptors(const unsigned char* descriptors, const std::vector<int>& groupStart,
                                       std::vector<int>& bestIdx, std::vector<unsigned char>& chosen) {
        const int g = (int)groupStart.size() - 1;
        bestIdx.assign(g > 0 ? g : 0, -1);
        chosen.assign((size_t)(g > 0 ? g : 0) * 32, 0);
        if (g <= 0) return;
        orbx::check(orbm_distinctive_descriptors(m_, descriptors, groupStart.data(), g, bestIdx.data(), nullptr, chosen.data(),
                                                 groupStart.back(), ORBX_MEM_HOST),
                    "orbm_distinctive_descriptors");
    }

    orbm_context* context() { return m_; }

protected:
    // Prior claims (map points a frame's keypoints hold before the call) go in as indices from
    // kPrior on, so that they can be told from the indices the search assigns.
    static const int kPrior = 0x40000000;
    template <class MatT>
    static void pose(const MatT& Tcw, float* R, float* t) {
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) R[3 * r + c] = Tcw.template at<float>(r, c);
            t[r] = Tcw.template at<float>(r, 3);
        }
    }
    template <class FrameT>
    static void claims_in(const FrameT& F, std::vector<int>& claimed, std::vector<unsigned char>& claimedObs) {
        claimed.assign(F.N, -1);
        claimedObs.assign(F.N, 0);
        for (int i = 0; i < F.N; i++)
            if (F.mvpMapPoints[i]) {
                claimed[i] = kPrior + i;
                claimedObs[i] = F.mvpMapPoints[i]->Observations() > 0 ? 1 : 0;
            }
    }
    // mvKeysUn / mDescriptors / image bounds of a Frame as the C ABI's frame (descriptor rows are
    // copied only when the matrix is not continuous)
    template <class FrameT>
    static orbm_frame frame_of(const FrameT& F, std::vector<unsigned char>& scratch) {
        typedef typename std::remove_reference<decltype(F.mvKeysUn[0])>::type KP;
        static_assert(sizeof(KP) == sizeof(orbx_keypoint), "keypoints must have the 28-byte cv::KeyPoint layout");
        orbm_frame f;
        const int n = (int)F.mvKeysUn.size();
        f.kps = reinterpret_cast<const orbx_keypoint*>(F.mvKeysUn.data());
        if (n == 0 || F.mDescriptors.isContinuous()) {
            f.desc = n ? F.mDescriptors.template ptr<unsigned char>(0) : nullptr;
        } else {
            scratch.resize((size_t)n * 32);
            for (int i = 0; i < n; i++) std::memcpy(&scratch[(size_t)i * 32], F.mDescriptors.template ptr<unsigned char>(i), 32);
            f.desc = scratch.data();
        }
        f.n = n;
        f.min_x = F.mnMinX;
        f.min_y = F.mnMinY;
        f.max_x = F.mnMaxX;
        f.max_y = F.mnMaxY;
        return f;
    }
    static orbm_pyramid pyramid(ORBextractor& e) {
        orbm_pyramid p;
        std::memset(&p, 0, sizeof(p));
        p.nlevels = e.GetLevels();
        for (int l = 0; l < p.nlevels; l++) {
            int w = 0, h = 0;
            size_t st = 0;
            p.level[l] = e.PyramidLevelDevice(l, &w, &h, &st);
            p.width[l] = w;
            p.height[l] = h;
            p.pitch[l] = st;
        }
        return p;
    }
    float mfNNratio;
    bool mbCheckOrientation;
    orbm_context* m_ = nullptr;
};

}  // namespace ORB_SLAM2

#endif  // ORBX_SHIM_HPP
