// Host program built against include/orbx_shim.hpp, the way Frame.cc would call the
// extractor (reference src/Frame.cc:320-326).  Reads a raw 8-bit image, writes the
// keypoints and descriptors as raw bytes; the Python test compares them with the
// result obtained through ctypes.
//   shim_check <raw image> <width> <height> <out file>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/orbx_shim.hpp"

int main(int argc, char** argv) {
    if (argc != 5) {
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
        std::fclose(o);
        std::printf("%d keypoints\n", n);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
