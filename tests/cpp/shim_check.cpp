// Host program built against include/orbx_shim.hpp, the way Frame.cc would call the
// extractor (reference src/Frame.cc:320-326).  Reads a raw 8-bit image, writes the
// keypoints and descriptors as raw bytes; the Python test compares them with the
// result obtained through ctypes.
//   shim_check <raw image> <width> <height> <out file> [<raw right image>]
// With a right image it also runs the stereo matcher the way Frame::ComputeStereoMatches would
// (reference src/Frame.cc:603) and appends mvuRight and mvDepth to the output.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "../../include/orbx_shim.hpp"

int main(int argc, char** argv) {
    if (argc != 5 && argc != 6) {
        std::fprintf(stderr, "usage: shim_check img.raw w h out.bin\n");
        return 2;
    }
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    std::vector<unsigned char> img((size_t)w * h);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(img.data(), 1, img.size(), f) != img.size()) return 3;
    std::fclose(f);
    try {
        ORB_SLAM2::ORBextractor extractor(1000, 1.2f, 8, 20, 7, w, h);
        std::vector<orbx::KeyPoint> keys;
        std::vector<unsigned char> desc;
        orbx::Image im = {img.data(), w, h, (size_t)w};
        orbx::Image none = {nullptr, 0, 0, 0};
        extractor(im, none, keys, desc);
        if (extractor.GetLevels() != 8 || extractor.GetScaleFactors().size() != 8) return 4;
        FILE* o = std::fopen(argv[4], "wb");
        int n = (int)keys.size();
        std::fwrite(&n, 4, 1, o);
        std::fwrite(keys.data(), sizeof(orbx::KeyPoint), keys.size(), o);
        std::fwrite(desc.data(), 1, desc.size(), o);
        {
            // mvImagePyramid as Frame::ComputeStereoMatches reads it (reference src/Frame.cc:610, :700)
            ORB_SLAM2::ORBextractor* mpORBextractorLeft = &extractor;
            const int nRows = mpORBextractorLeft->mvImagePyramid[0].rows;
            const int lvl = 3;
            const ORB_SLAM2::ORBextractor::HostLevel& L3 = mpORBextractorLeft->mvImagePyramid[lvl];
            const int dims[4] = {nRows, (int)mpORBextractorLeft->mvImagePyramid.size(), L3.rows, L3.cols};
            std::fwrite(dims, 4, 4, o);
            for (int r = 0; r < L3.rows; r++) std::fwrite(L3.ptr(r), 1, (size_t)L3.cols, o);
        }
        if (argc == 6) {
            std::vector<unsigned char> imgR((size_t)w * h);
            FILE* fr = std::fopen(argv[5], "rb");
            if (!fr || std::fread(imgR.data(), 1, imgR.size(), fr) != imgR.size()) return 3;
            std::fclose(fr);
            ORB_SLAM2::ORBextractor extractorRight(1000, 1.2f, 8, 20, 7, w, h);
            std::vector<orbx::KeyPoint> keysRight;
            std::vector<unsigned char> descRight;
            orbx::Image imR = {imgR.data(), w, h, (size_t)w};
            extractorRight(imR, none, keysRight, descRight);
            ORB_SLAM2::ORBmatcher stereo;
            std::vector<float> uRight, depth;
            int ns = stereo.ComputeStereoMatches(extractor, extractorRight, keys.data(), desc.data(), n, keysRight.data(),
                                                 descRight.data(), (int)keysRight.size(), 0.54f, 386.1448f, uRight, depth);
            std::fwrite(&ns, 4, 1, o);
            std::fwrite(uRight.data(), 4, uRight.size(), o);
            std::fwrite(depth.data(), 4, depth.size(), o);
            // SearchByBoW the way Tracking::TrackReferenceKeyFrame calls it (reference src/Tracking.cc:1166),
            // with a stand-in vocabulary: node = keypoint row band of 16 px
            std::map<unsigned, std::vector<unsigned> > fvL, fvR;
            for (size_t i = 0; i < keys.size(); i++) fvL[(unsigned)(keys[i].y / 16)].push_back((unsigned)i);
            for (size_t i = 0; i < keysRight.size(); i++) fvR[(unsigned)(keysRight[i].y / 16)].push_back((unsigned)i);
            const std::vector<float> sf = extractor.GetScaleFactors();
            ORB_SLAM2::FrameView vl = {keys.data(), desc.data(), n, 0, 0, (float)w, (float)h, sf.data(), 8, uRight.data()};
            ORB_SLAM2::FrameView vr = {keysRight.data(), descRight.data(), (int)keysRight.size(), 0, 0, (float)w, (float)h,
                                       sf.data(), 8, nullptr};
            std::vector<unsigned char> all(n, 1);
            std::vector<int> m12, m21;
            ORB_SLAM2::ORBmatcher bow(0.7f, true);
            int nb = bow.SearchByBoW(vl, fvL, all.data(), vr, fvR, nullptr, false, m12, m21);
            std::fwrite(&nb, 4, 1, o);
            std::fwrite(m12.data(), 4, m12.size(), o);
            // Frame::ComputeBoW (reference src/Frame.cc:513-522) on a one-level stand-in vocabulary whose 16
            // words are the first 16 left descriptors
            const int nw = 16;
            std::vector<unsigned char> nodeDesc((size_t)(nw + 1) * 32, 0);
            std::vector<int> childStart(nw + 2, nw), children(nw), wordId(nw + 1, -1);
            std::vector<double> weight(nw + 1, 0.0);
            childStart[0] = 0;
            for (int i = 0; i < nw; i++) {
                std::memcpy(&nodeDesc[(size_t)(i + 1) * 32], &desc[(size_t)i * 32], 32);
                children[i] = i + 1;
                wordId[i + 1] = i;
                weight[i + 1] = 1.0 + 0.125 * i;
            }
            orbm_vocabulary voc = {nw + 1, nw, 1, nodeDesc.data(), childStart.data(), children.data(), wordId.data(), weight.data()};
            std::map<unsigned, double> mBowVec;
            std::map<unsigned, std::vector<unsigned> > mFeatVec;
            bow.Transform(voc, desc.data(), n, mBowVec, mFeatVec, 0);
            const int nv = (int)mBowVec.size();
            std::fwrite(&nv, 4, 1, o);
            for (std::map<unsigned, double>::iterator it = mBowVec.begin(); it != mBowVec.end(); ++it) {
                std::fwrite(&it->first, 4, 1, o);
                std::fwrite(&it->second, 8, 1, o);
            }
            unsigned long nfeat = 0;
            for (std::map<unsigned, std::vector<unsigned> >::iterator it = mFeatVec.begin(); it != mFeatVec.end(); ++it)
                nfeat += it->first * 1000003ul + it->second.size() * 7ul + it->second.back();
            std::fwrite(&nfeat, 8, 1, o);
        }
        std::fclose(o);
        std::printf("%d keypoints\n", n);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
