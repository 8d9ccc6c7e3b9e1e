// The reference's call sites, as they stand in src/Tracking.cc, compiled against
// include/orbx_shim.hpp and run on the GPU: TrackWithMotionModel's frame-to-frame search
// (:1325-1340), SearchLocalPoints' map-point search (:1933-1943) and MonocularInitialization's
// search (:944-950).  Frames, poses and map points come from a file the Python test writes; the
// resulting mvpMapPoints / match arrays go back for comparison with the CPU oracle.
//   dropin_check <scenario.bin> <out.bin>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ref_standin.hpp"

#include "../../include/orbx_shim.hpp"

using namespace std;
using namespace ORB_SLAM2;

namespace {

struct Reader {
    FILE* f;
    template <class T>
    void get(T* p, size_t n) {
        if (n && fread(p, sizeof(T), n, f) != n) {
            fprintf(stderr, "scenario file too short\n");
            exit(3);
        }
    }
    int i32() {
        int v;
        get(&v, 1);
        return v;
    }
    float f32() {
        float v;
        get(&v, 1);
        return v;
    }
};

void read_frame(Reader& r, Frame& F) {
    F.N = r.i32();
    F.mvKeysUn.resize(F.N);
    r.get(F.mvKeysUn.data(), F.N);
    F.mvKeys = F.mvKeysUn;
    F.mDescriptors = cv::Mat(F.N, 32, 1);
    r.get(F.mDescriptors.ptr<unsigned char>(0), (size_t)F.N * 32);
    F.mvuRight.resize(F.N);
    r.get(F.mvuRight.data(), F.N);
    r.get(F.mTcw.ptr<float>(0), 16);
    F.mvScaleFactors.resize(8);
    r.get(F.mvScaleFactors.data(), 8);
    F.mbf = r.f32();
    F.mb = r.f32();
    F.mvpMapPoints.assign(F.N, static_cast<MapPoint*>(NULL));
    F.mvbOutlier.assign(F.N, false);
}

void read_points(Reader& r, vector<MapPoint>& pts, long unsigned idBase) {
    const int n = r.i32();
    pts.resize(n);
    for (int i = 0; i < n; i++) {
        MapPoint& p = pts[i];
        p.mnId = idBase + i;
        r.get(p.mWorldPos.ptr<float>(0), 3);
        r.get(p.mDescriptor.ptr<unsigned char>(0), 32);
        p.nObs = r.i32();
        p.mbBad = r.i32() != 0;
        p.mbTrackInView = r.i32() != 0;
        p.mnTrackScaleLevel = r.i32();
        p.mTrackViewCos = r.f32();
        p.mTrackProjX = r.f32();
        p.mTrackProjY = r.f32();
        p.mTrackProjXR = r.f32();
    }
}

struct System {
    enum { MONOCULAR = 0, STEREO = 1, RGBD = 2 };
};

}  // namespace

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    Reader r = {fopen(argv[1], "rb")};
    if (!r.f) return 3;
    Frame::fx = r.f32();
    Frame::fy = r.f32();
    Frame::cx = r.f32();
    Frame::cy = r.f32();
    Frame::mnMinX = r.f32();
    Frame::mnMinY = r.f32();
    Frame::mnMaxX = r.f32();
    Frame::mnMaxY = r.f32();
    const int mSensor = r.i32();
    Frame mLastFrame, mCurrentFrame, mInitialFrame;
    read_frame(r, mLastFrame);
    read_frame(r, mCurrentFrame);
    read_frame(r, mInitialFrame);
    vector<MapPoint> lastPoints, localPoints;
    read_points(r, lastPoints, 0);          // one per last-frame keypoint; mbBad marks "no map point here"
    read_points(r, localPoints, 1000000);   // the local map
    vector<int> outlier(mLastFrame.N);
    r.get(outlier.data(), outlier.size());
    for (int i = 0; i < mLastFrame.N; i++) {
        mLastFrame.mvpMapPoints[i] = lastPoints[i].mbBad ? static_cast<MapPoint*>(NULL) : &lastPoints[i];
        mLastFrame.mvbOutlier[i] = outlier[i] != 0;
    }
    vector<MapPoint*> mvpLocalMapPoints;
    for (size_t i = 0; i < localPoints.size(); i++) mvpLocalMapPoints.push_back(&localPoints[i]);
    vector<cv::Point2f> mvbPrevMatched(mInitialFrame.mvKeysUn.size());
    for (size_t i = 0; i < mInitialFrame.mvKeysUn.size(); i++) mvbPrevMatched[i] = mInitialFrame.mvKeysUn[i].pt;
    vector<int> mvIniMatches;
    fclose(r.f);
    int nmatchesMotion = 0, nmatchesLocal = 0, nmatchesIni = 0, nmatchesWide = -1;
    try {
        {
            // ---- src/Tracking.cc:1314, :1325-1340 (TrackWithMotionModel)
            ORBmatcher matcher(0.9,true);
            fill(mCurrentFrame.mvpMapPoints.begin(),mCurrentFrame.mvpMapPoints.end(),static_cast<MapPoint*>(NULL));

            // Project points seen in previous frame
            int th;
            if(mSensor  != System::STEREO)
                th=15;// search window
            else
                th=7;
            int nmatches = matcher.SearchByProjection(mCurrentFrame,mLastFrame,th,mSensor==System::MONOCULAR);
            nmatchesMotion = nmatches;

            // If few matches, uses a wider window search
            if(nmatches<20)
            {
                fill(mCurrentFrame.mvpMapPoints.begin(),mCurrentFrame.mvpMapPoints.end(),static_cast<MapPoint*>(NULL));
                nmatches = matcher.SearchByProjection(mCurrentFrame,mLastFrame,2*th,mSensor==System::MONOCULAR);
                nmatchesWide = nmatches;
            }
        }
        vector<long> afterMotion(mCurrentFrame.N);
        for (int i = 0; i < mCurrentFrame.N; i++)
            afterMotion[i] = mCurrentFrame.mvpMapPoints[i] ? (long)mCurrentFrame.mvpMapPoints[i]->mnId : -1;
        {
            // ---- src/Tracking.cc:1933-1943 (SearchLocalPoints)
            ORBmatcher matcher(0.8);// 0.8 Shortest distance and second shortest distance ratio difference
            int th = 1;
            if(mSensor==System::RGBD)
                th=3;
            nmatchesLocal = matcher.SearchByProjection(mCurrentFrame,mvpLocalMapPoints,th);
        }
        {
            // ---- src/Tracking.cc:944-950 (MonocularInitialization)
            ORBmatcher matcher(0.9,true);
            int nmatches = matcher.SearchForInitialization(mInitialFrame,mCurrentFrame,mvbPrevMatched,mvIniMatches,100);
            nmatchesIni = nmatches;
        }
        // ---- src/ORBmatcher.cc:1905 through its static signature
        const int dd = ORBmatcher::DescriptorDistance(mCurrentFrame.mDescriptors.row(0), mLastFrame.mDescriptors.row(1));
        FILE* o = fopen(argv[2], "wb");
        const int head[6] = {nmatchesMotion, nmatchesWide, nmatchesLocal, nmatchesIni, dd, mCurrentFrame.N};
        fwrite(head, 4, 6, o);
        fwrite(afterMotion.data(), sizeof(long), afterMotion.size(), o);
        vector<long> afterLocal(mCurrentFrame.N);
        for (int i = 0; i < mCurrentFrame.N; i++)
            afterLocal[i] = mCurrentFrame.mvpMapPoints[i] ? (long)mCurrentFrame.mvpMapPoints[i]->mnId : -1;
        fwrite(afterLocal.data(), sizeof(long), afterLocal.size(), o);
        const int ni = (int)mvIniMatches.size();
        fwrite(&ni, 4, 1, o);
        fwrite(mvIniMatches.data(), 4, mvIniMatches.size(), o);
        fwrite(mvbPrevMatched.data(), 8, mvbPrevMatched.size(), o);
        fclose(o);
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
