// Header-only C++ drop-in for the two reference classes, on top of the orbx C ABI.
//
//   ORB_SLAM2::ORBextractor  (reference include/ORBextractor.h:32-105)
//   ORB_SLAM2::ORBmatcher    (reference include/ORBmatcher.h:25-93; the entry points
//                             that exist in the C ABI)
//
// With OpenCV available, define ORBX_WITH_OPENCV before including this header: the
// classes then take cv::InputArray / std::vector<cv::KeyPoint> / cv::OutputArray exactly
// like the reference, and Frame.cc / Tracking.cc compile against it unchanged.  Without
// OpenCV the same classes are expressed on two small stand-in types (orbx::KeyPoint,
// orbx::Image) so that host code and tests can be built anywhere.
//
// One extractor object owns one orbx handle (one HIP stream).  As in the reference, an
// object must not be used from two threads at once; two objects may run concurrently
// (stereo: reference src/Frame.cc:122-125).
#ifndef ORBX_SHIM_HPP
#define ORBX_SHIM_HPP

#include <cstring>
#include <stdexcept>
#include <string>
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
