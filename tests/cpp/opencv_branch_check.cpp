// The ORBX_WITH_OPENCV branch of include/orbx_shim.hpp, compiled against a declaration-compatible
// subset of the OpenCV core API (tests/cpp/opencv_api_subset: this image has no OpenCV) and run the
// way the reference calls the extractor (src/Frame.cc:320-326):
//     (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
// Output: n, keypoints (28 bytes each), descriptors (32 bytes each), then rows/cols of
// mvImagePyramid[2] and its pixels, for the test to compare with the ctypes binding.
#define ORBX_WITH_OPENCV
#include "../../include/orbx_shim.hpp"

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const int w = atoi(argv[2]), h = atoi(argv[3]);
    std::vector<unsigned char> pixels((size_t)w * h);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(pixels.data(), 1, pixels.size(), f) != pixels.size()) return 3;
    fclose(f);
    try {
        ORB_SLAM2::ORBextractor* mpORBextractorLeft = new ORB_SLAM2::ORBextractor(1000, 1.2f, 8, 20, 7);
        cv::Mat im(h, w, CV_8UC1, pixels.data());
        std::vector<cv::KeyPoint> mvKeys;
        cv::Mat mDescriptors;
        (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
        const int n = (int)mvKeys.size();
        if (mDescriptors.rows != n || (n && mDescriptors.cols != 32)) return 4;
        FILE* o = fopen(argv[4], "wb");
        fwrite(&n, 4, 1, o);
        fwrite(mvKeys.data(), sizeof(cv::KeyPoint), mvKeys.size(), o);
        for (int i = 0; i < n; i++) fwrite(mDescriptors.ptr(i), 1, 32, o);
        const cv::Mat& lvl = mpORBextractorLeft->mvImagePyramid[2];
        const int dims[2] = {lvl.rows, lvl.cols};
        fwrite(dims, 4, 2, o);
        for (int r = 0; r < lvl.rows; r++) fwrite(lvl.ptr(r), 1, (size_t)lvl.cols, o);
        // an empty image leaves the outputs alone, as the reference's early return does (src/ORBextractor.cc:1055-1056)
        std::vector<cv::KeyPoint> none;
        cv::Mat noDesc;
        (*mpORBextractorLeft)(cv::Mat(), cv::Mat(), none, noDesc);
        const int ok = none.empty() && noDesc.empty();
        fwrite(&ok, 4, 1, o);
        fclose(o);
        delete mpORBextractorLeft;
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
