// CPU restatement of the reference ORB extractor.  TEST INFRASTRUCTURE ONLY:
// the parity checker for the HIP path and the benchmark's CPU baseline.  The
// product library never links or loads this file.
//
// Parity status: "parity unpinned" at the OpenCV boundary.  The reference has
// no golden vectors for this path and OpenCV (version not pinned by the
// reference: CMake accepts 2.4.3+/3.x) is absent, so cv::resize, cv::FAST,
// cv::GaussianBlur, cv::fastAtan2 and cvRound are restated here from their
// published generic C++ algorithms.  Everything else follows the reference
// source line by line; each function cites the lines it restates.
//
// Build: g++ -O3 -ffp-contract=off (no FMA contraction: the reference's float
// expressions are evaluated one rounding per operation).

#include "orb_oracle.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <thread>
#include <utility>
#include <chrono>
#include <vector>

#include "../include/orbx_pattern_31.h"

namespace {

const int PATCH_SIZE = 31;       // reference src/ORBextractor.cc:51
const int HALF_PATCH_SIZE = 15;  // :52
const int EDGE_THRESHOLD = 19;   // :53

// cvRound: round half to even (SSE cvtsd2si under the default rounding mode).
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(double v) { return (int)std::floor(v); }
inline int cv_ceil(double v) { return (int)std::ceil(v); }

struct Tables {
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> quotas;
    int umax[HALF_PATCH_SIZE + 1];
};

// reference src/ORBextractor.cc:399-466.  scaleFactor is a double member that
// holds the float argument (include/ORBextractor.h:91).
Tables make_tables(const oracle_params& p) {
    Tables t;
    const int n = p.nlevels;
    const double sf = (double)p.scale_factor;
    t.scale.resize(n);
    t.sigma2.resize(n);
    t.inv_scale.resize(n);
    t.inv_sigma2.resize(n);
    t.quotas.resize(n);
    t.scale[0] = 1.0f;
    t.sigma2[0] = 1.0f;
    for (int i = 1; i < n; i++) {
        t.scale[i] = (float)(t.scale[i - 1] * sf);
        t.sigma2[i] = t.scale[i] * t.scale[i];
    }
    for (int i = 0; i < n; i++) {
        t.inv_scale[i] = 1.0f / t.scale[i];
        t.inv_sigma2[i] = 1.0f / t.sigma2[i];
    }
    float factor = (float)(1.0f / sf);
    float desired = p.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)n));
    int sum = 0;
    for (int level = 0; level < n - 1; level++) {
        t.quotas[level] = cv_round(desired);
        sum += t.quotas[level];
        desired *= factor;
    }
    t.quotas[n - 1] = std::max(p.nfeatures - sum, 0);

    int v, v0, vmax = cv_floor(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HALF_PATCH_SIZE * sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) t.umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
        t.umax[v] = v0;
        ++v0;
    }
    return t;
}

// Level size: reference src/ORBextractor.cc:1121-1122.
void level_size(const Tables& t, int w, int h, int level, int* lw, int* lh) {
    float s = t.inv_scale[level];
    *lw = cv_round((float)w * s);
    *lh = cv_round((float)h * s);
}

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    Image() {}
    Image(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
    const uint8_t* row(int y) const { return d.data() + (size_t)y * w; }
    uint8_t* row(int y) { return d.data() + (size_t)y * w; }
};

// cv::resize, 8UC1, INTER_LINEAR, generic fixed-point path (11-bit
// coefficients, INTER_RESIZE_COEF_BITS = 11).  Called at
// reference src/ORBextractor.cc:1134.
void resize_linear(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw, int dh,
                   size_t dstride) {
    const int SCALE = 2048;
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) {
            fx = 0;
            sx = 0;
        }
        if (sx + 1 >= sw) {
            xmax = std::min(xmax, dx);
            if (sx >= sw - 1) {
                fx = 0;
                sx = sw - 1;
            }
        }
        xofs[dx] = sx;
        ialpha[dx * 2] = (short)cv_round((1.f - fx) * SCALE);
        ialpha[dx * 2 + 1] = (short)cv_round(fx * SCALE);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = (short)cv_round((1.f - fy) * SCALE);
        ibeta[dy * 2 + 1] = (short)cv_round(fy * SCALE);
    }
    std::vector<int> r0(dw), r1(dw);
    auto hresize = [&](const uint8_t* S, int* D) {
        int dx = 0;
        for (; dx < xmax; dx++) {
            int sx = xofs[dx];
            D[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
        }
        for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * SCALE;
    };
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = std::min(std::max(yofs[dy], 0), sh - 1);
        int sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
        hresize(src + (size_t)sy0 * sstride, r0.data());
        hresize(src + (size_t)sy1 * sstride, r1.data());
        int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

// cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on a standalone 8-bit
// image, integer path: 8-bit kernel, 16-bit row sums, rounding shift by 16.
// Called at reference src/ORBextractor.cc:1094-1095 on a clone of the level.
void gaussian7(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride,
               int variant) {
    static const int K257[7] = {18, 34, 49, 55, 49, 34, 18};  // variants 0 and 2
    static const int K256[7] = {18, 34, 48, 56, 48, 34, 18};
    const int* k = variant == 1 ? K256 : K257;
    // row sums fit 16 bits (257 * 255 = 65535)
    std::vector<uint16_t> rows((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstride;
        uint16_t* R = rows.data() + (size_t)y * w;
        const int x0 = std::min(3, w), x1 = std::max(w - 3, x0);
        auto edge = [&](int x) {
            int s = 0;
            for (int i = -3; i <= 3; i++) s += k[i + 3] * S[reflect101(x + i, w)];
            R[x] = (uint16_t)s;
        };
        for (int x = 0; x < x0; x++) edge(x);
        for (int x = x0; x < x1; x++)
            R[x] = (uint16_t)(k[0] * (S[x - 3] + S[x + 3]) + k[1] * (S[x - 2] + S[x + 2]) + k[2] * (S[x - 1] + S[x + 1]) +
                              k[3] * S[x]);
        for (int x = x1; x < w; x++) edge(x);
    }
    for (int y = 0; y < h; y++) {
        uint8_t* D = dst + (size_t)y * dstride;
        const uint16_t* r[7];
        for (int i = -3; i <= 3; i++) r[i + 3] = rows.data() + (size_t)reflect101(y + i, h) * w;
        // Variant 2 (OpenCV up to 3.4.0, the versions the reference's README names): the column pass of
        // the 32-bit row sums runs in SSE2 single precision.  The column taps are the integers scaled by
        // 2^-16 as floats; per pixel s = (float)S[0]*ky[0], then s += (float)(S[k] + S[-k]) * ky[k] for
        // k = 1..3, and the result is converted with round-half-to-even (cvtps2dq) and saturated.  The
        // vector loop covers the columns in fours; the last w % 4 columns take the scalar fixed-point
        // cast, which rounds half up like variant 0.
        const int xvec = variant == 2 ? (w & ~3) : 0;
        const float ky[4] = {(float)k[3] * (1.f / 65536.f), (float)k[2] * (1.f / 65536.f), (float)k[1] * (1.f / 65536.f),
                             (float)k[0] * (1.f / 65536.f)};
        for (int x = 0; x < xvec; x++) {
            float s0 = (float)(int)r[3][x] * ky[0];
            s0 = s0 + (float)((int)r[4][x] + (int)r[2][x]) * ky[1];
            s0 = s0 + (float)((int)r[5][x] + (int)r[1][x]) * ky[2];
            s0 = s0 + (float)((int)r[6][x] + (int)r[0][x]) * ky[3];
            const int v = (int)lrintf(s0);
            D[x] = (uint8_t)std::min(std::max(v, 0), 255);
        }
        for (int x = xvec; x < w; x++) {
            const int s = k[0] * (r[0][x] + r[6][x]) + k[1] * (r[1][x] + r[5][x]) + k[2] * (r[2][x] + r[4][x]) + k[3] * r[3][x];
            const int v = (s + 32768) >> 16;
            D[x] = (uint8_t)std::min(v, 255);
        }
    }
}

// cv::fastAtan2 (degrees, [0,360)), scalar path.  Used at reference
// src/ORBextractor.cc:87.
float fast_atan2(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::abs(x), ay = std::abs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---- cv::FAST(roi, kps, threshold, nonmaxSuppression=true), FAST-9/16 ----
const int RING[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                         {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int corner_score16(const uint8_t* ptr, const int* pixel, int threshold) {
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[N];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

struct Cand {
    int x, y, resp;
};

void fast9_nms(const uint8_t* img, int w, int h, size_t stride, int threshold,
               std::vector<Cand>& out) {
    out.clear();
    if (w < 7 || h < 7) return;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = RING[k][0] + RING[k][1] * (int)stride;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    const int K = 8, N = 25;
    std::vector<uint8_t> score((size_t)w * h, 0);
    threshold = std::min(std::max(threshold, 0), 255);
    uint8_t threshold_tab[512];
    for (int i = -255; i <= 255; i++)
        threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    for (int i = 3; i < h - 3; i++) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        for (int j = 3; j < w - 3; j++, ptr++) {
            int v = ptr[0];
            const uint8_t* tab = &threshold_tab[0] - v + 255;
            int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
            if (d == 0) continue;
            d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
            d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
            d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
            if (d == 0) continue;
            d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
            d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
            d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
            d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
            bool corner = false;
            if (d & 1) {
                int vt = v - threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    int x = ptr[pixel[k]];
                    if (x < vt) {
                        if (++count > K) {
                            corner = true;
                            break;
                        }
                    } else
                        count = 0;
                }
            }
            if (!corner && (d & 2)) {
                int vt = v + threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    int x = ptr[pixel[k]];
                    if (x > vt) {
                        if (++count > K) {
                            corner = true;
                            break;
                        }
                    } else
                        count = 0;
                }
            }
            if (corner) score[(size_t)i * w + j] = (uint8_t)corner_score16(ptr, pixel, threshold);
        }
    }
    // Non-max suppression against the 8 neighbours' scores; non-corners and
    // everything outside rows/cols [3, dim-3) hold 0.
    for (int i = 3; i < h - 3; i++) {
        for (int j = 3; j < w - 3; j++) {
            int s = score[(size_t)i * w + j];
            if (s == 0) continue;
            const uint8_t* pp = &score[(size_t)(i - 1) * w + j];
            const uint8_t* pc = &score[(size_t)i * w + j];
            const uint8_t* pn = &score[(size_t)(i + 1) * w + j];
            if (s > pc[-1] && s > pc[1] && s > pp[-1] && s > pp[0] && s > pp[1] && s > pn[-1] &&
                s > pn[0] && s > pn[1])
                out.push_back({j, i, s});
        }
    }
}

// reference src/ORBextractor.cc:776-839: the per-cell detection loop of
// ComputeKeyPointsOctTree.  Output coordinates are relative to (minBorderX,
// minBorderY) exactly as vToDistributeKeys holds them.
void detect_cells(const uint8_t* lvl, int w, int h, size_t stride, int ini_th, int min_th,
                  std::vector<Cand>& all) {
    all.clear();
    const float W = 30;
    const int minBorderX = EDGE_THRESHOLD - 3;
    const int minBorderY = minBorderX;
    const int maxBorderX = w - EDGE_THRESHOLD + 3;
    const int maxBorderY = h - EDGE_THRESHOLD + 3;
    const float width = (maxBorderX - minBorderX);
    const float height = (maxBorderY - minBorderY);
    const int nCols = width / W;
    const int nRows = height / W;
    if (nCols <= 0 || nRows <= 0) return;
    const int wCell = ceil(width / nCols);
    const int hCell = ceil(height / nRows);
    std::vector<Cand> cell;
    for (int i = 0; i < nRows; i++) {
        const float iniY = minBorderY + i * hCell;
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = minBorderX + j * wCell;
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = maxBorderX;
            const int x0 = (int)iniX, y0 = (int)iniY, x1 = (int)maxX, y1 = (int)maxY;
            const uint8_t* roi = lvl + (size_t)y0 * stride + x0;
            fast9_nms(roi, x1 - x0, y1 - y0, stride, ini_th, cell);
            if (cell.empty()) fast9_nms(roi, x1 - x0, y1 - y0, stride, min_th, cell);
            for (const Cand& c : cell) all.push_back({c.x + j * wCell, c.y + i * hCell, c.resp});
        }
    }
}

// ---- DistributeOctTree: reference src/ORBextractor.cc:478-765 ----
struct Node {
    std::vector<int> keys;  // indices into the candidate array, insertion order
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::list<Node>::iterator lit;
    bool no_more = false;
    // Stand-in for the node's heap address in the reference's
    // sort(pair<int, ExtractorNode*>) at :686.  The reference order is
    // allocator-dependent; this restatement fixes "created later compares
    // greater".
    long serial = 0;
};

void divide_node(const Node& p, const std::vector<Cand>& c, Node& n1, Node& n2, Node& n3,
                 Node& n4) {
    const int halfX = ceil(static_cast<float>(p.URx - p.ULx) / 2);
    const int halfY = ceil(static_cast<float>(p.BRy - p.ULy) / 2);
    n1.ULx = p.ULx; n1.ULy = p.ULy;
    n1.URx = p.ULx + halfX; n1.URy = p.ULy;
    n1.BLx = p.ULx; n1.BLy = p.ULy + halfY;
    n1.BRx = p.ULx + halfX; n1.BRy = p.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = p.URx; n2.URy = p.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = p.URx; n2.BRy = p.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = p.BLx; n3.BLy = p.BLy;
    n3.BRx = n1.BRx; n3.BRy = p.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = p.BRx; n4.BRy = p.BRy;
    for (int idx : p.keys) {
        const float px = (float)c[idx].x, py = (float)c[idx].y;
        if (px < n1.URx) {
            if (py < n1.BRy) n1.keys.push_back(idx);
            else n3.keys.push_back(idx);
        } else if (py < n1.BRy)
            n2.keys.push_back(idx);
        else
            n4.keys.push_back(idx);
    }
    if (n1.keys.size() == 1) n1.no_more = true;
    if (n2.keys.size() == 1) n2.no_more = true;
    if (n3.keys.size() == 1) n3.no_more = true;
    if (n4.keys.size() == 1) n4.no_more = true;
}

void distribute(const std::vector<Cand>& c, int minX, int maxX, int minY, int maxY, int N,
                std::vector<int>& result) {
    result.clear();
    const int nIni = round(static_cast<float>(maxX - minX) / (maxY - minY));
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<Node> nodes;
    std::vector<Node*> ini(nIni);
    long serial = 0;
    for (int i = 0; i < nIni; i++) {
        Node ni;
        ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
        ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.serial = serial++;
        nodes.push_back(ni);
        ini[i] = &nodes.back();
    }
    for (size_t i = 0; i < c.size(); i++) ini[(size_t)((float)c[i].x / hX)]->keys.push_back((int)i);

    auto lit = nodes.begin();
    while (lit != nodes.end()) {
        if (lit->keys.size() == 1) {
            lit->no_more = true;
            lit++;
        } else if (lit->keys.empty())
            lit = nodes.erase(lit);
        else
            lit++;
    }

    bool finish = false;
    typedef std::pair<int, Node*> SizePtr;
    auto by_size_then_serial = [](const SizePtr& a, const SizePtr& b) {
        if (a.first != b.first) return a.first < b.first;
        return a.second->serial < b.second->serial;
    };
    std::vector<SizePtr> size_and_ptr;
    auto push_children = [&](Node* kids[4], int* to_expand) {
        for (int q = 0; q < 4; q++) {
            Node& n = *kids[q];
            if (n.keys.size() > 0) {
                n.serial = serial++;
                nodes.push_front(n);
                if (n.keys.size() > 1) {
                    if (to_expand) (*to_expand)++;
                    size_and_ptr.push_back(std::make_pair((int)n.keys.size(), &nodes.front()));
                    nodes.front().lit = nodes.begin();
                }
            }
        }
    };
    while (!finish) {
        int prevSize = nodes.size();
        lit = nodes.begin();
        int nToExpand = 0;
        size_and_ptr.clear();
        while (lit != nodes.end()) {
            if (lit->no_more) {
                lit++;
                continue;
            }
            Node n1, n2, n3, n4;
            divide_node(*lit, c, n1, n2, n3, n4);
            Node* kids[4] = {&n1, &n2, &n3, &n4};
            push_children(kids, &nToExpand);
            lit = nodes.erase(lit);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
            finish = true;
        } else if (((int)nodes.size() + nToExpand * 3) > N) {
            while (!finish) {
                prevSize = nodes.size();
                std::vector<SizePtr> prev = size_and_ptr;
                size_and_ptr.clear();
                std::sort(prev.begin(), prev.end(), by_size_then_serial);
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    divide_node(*prev[j].second, c, n1, n2, n3, n4);
                    Node* kids[4] = {&n1, &n2, &n3, &n4};
                    push_children(kids, nullptr);
                    nodes.erase(prev[j].second->lit);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
            }
        }
    }
    for (auto it = nodes.begin(); it != nodes.end(); it++) {
        int best = it->keys[0];
        float maxResponse = (float)c[best].resp;
        for (size_t k = 1; k < it->keys.size(); k++) {
            if ((float)c[it->keys[k]].resp > maxResponse) {
                best = it->keys[k];
                maxResponse = (float)c[best].resp;
            }
        }
        result.push_back(best);
    }
}

// reference src/ORBextractor.cc:59-88.
float ic_angle(const uint8_t* img, size_t stride, int x, int y, const int* umax) {
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)y * stride + x;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    int step = (int)stride;
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

// reference src/ORBextractor.cc:91-131.  cos/sin on a float argument under
// `using namespace std` select the float overloads, i.e. cosf/sinf.
const float factorPI = (float)(M_PI / 180.f);
// fma_variant 1: a build of the reference with -march=native on a host with fused multiply-add
// (perfect/CMakeLists.txt:20) lets GCC contract `x*b + y*a` and `x*a - y*b`: the first product is
// fused into the addition, the second product is rounded on its own.
void orb_descriptor(const uint8_t* img, size_t stride, int x, int y, float angle_deg,
                    uint8_t* desc, int fma_variant = 0) {
    float angle = angle_deg * factorPI;
    float a = cosf(angle), b = sinf(angle);
    const uint8_t* center = img + (size_t)y * stride + x;
    const int step = (int)stride;
    const signed char* pat = ORBX_PATTERN_31;
    auto get = [&](int idx) -> int {
        float px = (float)pat[2 * idx], py = (float)pat[2 * idx + 1];
        if (fma_variant == 1) {
            const float t1 = py * a, t2 = py * b;
            return center[cv_round(fmaf(px, b, t1)) * step + cv_round(fmaf(px, a, -t2))];
        }
        return center[cv_round(px * b + py * a) * step + cv_round(px * a - py * b)];
    };
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int t = 0; t < 8; t++) {
            int t0 = get(2 * t), t1 = get(2 * t + 1);
            val |= (t0 < t1) << t;
        }
        desc[i] = (uint8_t)val;
    }
}

void build_pyramid(const Tables& t, const uint8_t* img, int w, int h, size_t stride, int nlevels,
                   std::vector<Image>& pyr) {
    pyr.resize(nlevels);
    for (int level = 0; level < nlevels; ++level) {
        int lw, lh;
        level_size(t, w, h, level, &lw, &lh);
        pyr[level] = Image(lw, lh);
        if (level == 0) {
            for (int y = 0; y < h; y++) memcpy(pyr[0].row(y), img + (size_t)y * stride, w);
        } else {
            resize_linear(pyr[level - 1].d.data(), pyr[level - 1].w, pyr[level - 1].h,
                          pyr[level - 1].w, pyr[level].d.data(), lw, lh, lw);
        }
    }
}

// reference src/ORBextractor.cc:1052-1114 (operator()) with :772-863
// (ComputeKeyPointsOctTree) inlined.
// stage_s, when given, accumulates wall-clock seconds per stage: pyramid, FAST cells, quadtree,
// orientation, blur, descriptors (std::chrono::steady_clock, as the reference's examples time
// a frame, Examples/RGB-D/rgbd_tum.cc:92-108).
int extract(const oracle_params& p, const Tables& t, const uint8_t* img, int w, int h,
            size_t stride, oracle_keypoint* kps, uint8_t* desc, int cap, int* n_out, double* stage_s = nullptr) {
    typedef std::chrono::steady_clock Clock;
    Clock::time_point mark = Clock::now();
    auto lap = [&](int stage) {
        if (!stage_s) return;
        const Clock::time_point now = Clock::now();
        stage_s[stage] += std::chrono::duration<double>(now - mark).count();
        mark = now;
    };
    *n_out = 0;
    if (w <= 0 || h <= 0 || !img) return 0;
    std::vector<Image> pyr;
    build_pyramid(t, img, w, h, stride, p.nlevels, pyr);
    lap(0);
    std::vector<std::vector<oracle_keypoint>> all(p.nlevels);
    std::vector<Cand> cands;
    std::vector<int> sel;
    for (int level = 0; level < p.nlevels; ++level) {
        const Image& L = pyr[level];
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = L.w - EDGE_THRESHOLD + 3;
        const int maxBorderY = L.h - EDGE_THRESHOLD + 3;
        detect_cells(L.d.data(), L.w, L.h, L.w, p.ini_th_fast, p.min_th_fast, cands);
        lap(1);
        // A level more than twice as tall as wide has round(width / height) = 0 initial nodes:
        // the reference (src/ORBextractor.cc:553-555) then divides by zero and indexes an
        // empty vector.  Reported as an error instead of reproduced.
        if (!cands.empty() && maxBorderY > minBorderY &&
            (int)round(static_cast<float>(maxBorderX - minBorderX) / (maxBorderY - minBorderY)) < 1)
            return -3;
        distribute(cands, minBorderX, maxBorderX, minBorderY, maxBorderY, t.quotas[level], sel);
        const int scaledPatchSize = PATCH_SIZE * t.scale[level];
        for (int idx : sel) {
            oracle_keypoint k;
            k.x = (float)cands[idx].x;
            k.y = (float)cands[idx].y;
            k.x += minBorderX;
            k.y += minBorderY;
            k.size = scaledPatchSize;
            k.angle = -1;
            k.response = (float)cands[idx].resp;
            k.octave = level;
            k.class_id = -1;
            all[level].push_back(k);
        }
        lap(2);
    }
    for (int level = 0; level < p.nlevels; ++level)
        for (auto& k : all[level])
            k.angle = ic_angle(pyr[level].d.data(), pyr[level].w, cv_round(k.x), cv_round(k.y), t.umax);
    lap(3);

    int total = 0;
    for (int level = 0; level < p.nlevels; ++level) total += (int)all[level].size();
    if (total > cap) return -1;
    int offset = 0;
    for (int level = 0; level < p.nlevels; ++level) {
        std::vector<oracle_keypoint>& v = all[level];
        if (v.empty()) continue;
        const Image& L = pyr[level];
        Image blurred(L.w, L.h);
        gaussian7(L.d.data(), L.w, L.h, L.w, blurred.d.data(), L.w, p.blur_variant);
        lap(4);
        for (size_t i = 0; i < v.size(); i++)
            orb_descriptor(blurred.d.data(), L.w, cv_round(v[i].x), cv_round(v[i].y), v[i].angle,
                           desc + (size_t)(offset + i) * 32, p.fma_variant);
        if (level != 0) {
            float scale = t.scale[level];
            for (auto& k : v) {
                k.x *= scale;
                k.y *= scale;
            }
        }
        for (size_t i = 0; i < v.size(); i++) kps[offset + i] = v[i];
        offset += (int)v.size();
        lap(5);
    }
    *n_out = total;
    return 0;
}

}  // namespace

extern "C" {

void oracle_tables(const oracle_params* p, float* scale, float* inv_scale, float* sigma2,
                   float* inv_sigma2, int* quotas, int* umax16) {
    Tables t = make_tables(*p);
    for (int i = 0; i < p->nlevels; i++) {
        if (scale) scale[i] = t.scale[i];
        if (inv_scale) inv_scale[i] = t.inv_scale[i];
        if (sigma2) sigma2[i] = t.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = t.inv_sigma2[i];
        if (quotas) quotas[i] = t.quotas[i];
    }
    if (umax16)
        for (int i = 0; i <= HALF_PATCH_SIZE; i++) umax16[i] = t.umax[i];
}

void oracle_level_size(const oracle_params* p, int w, int h, int level, int* lw, int* lh) {
    Tables t = make_tables(*p);
    level_size(t, w, h, level, lw, lh);
}

int oracle_cv_round(double v) { return cv_round(v); }
void oracle_sincosf(const float* a, int n, float* s, float* c) {
    for (int i = 0; i < n; i++) {
        s[i] = sinf(a[i]);
        c[i] = cosf(a[i]);
    }
}
float oracle_fast_atan2(float y, float x) { return fast_atan2(y, x); }

void oracle_resize_linear(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw,
                          int dh, size_t dstride) {
    resize_linear(src, sw, sh, sstride, dst, dw, dh, dstride);
}

void oracle_gaussian7(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst,
                      size_t dstride, int variant) {
    gaussian7(src, w, h, sstride, dst, dstride, variant);
}

int oracle_fast9_nms(const uint8_t* roi, int w, int h, size_t stride, int threshold, int* xs,
                     int* ys, int* scores, int cap) {
    std::vector<Cand> out;
    fast9_nms(roi, w, h, stride, threshold, out);
    int n = std::min((int)out.size(), cap);
    for (int i = 0; i < n; i++) {
        xs[i] = out[i].x;
        ys[i] = out[i].y;
        scores[i] = out[i].resp;
    }
    return (int)out.size();
}

void oracle_pyramid_level(const oracle_params* p, const uint8_t* img, int w, int h, size_t stride,
                          int level, uint8_t* dst) {
    Tables t = make_tables(*p);
    std::vector<Image> pyr;
    build_pyramid(t, img, w, h, stride, level + 1, pyr);
    memcpy(dst, pyr[level].d.data(), pyr[level].d.size());
}

int oracle_detect_cells(const uint8_t* lvl, int w, int h, size_t stride, int ini_th, int min_th,
                        int* xs, int* ys, int* resp, int cap) {
    std::vector<Cand> all;
    detect_cells(lvl, w, h, stride, ini_th, min_th, all);
    int n = std::min((int)all.size(), cap);
    for (int i = 0; i < n; i++) {
        xs[i] = all[i].x;
        ys[i] = all[i].y;
        resp[i] = all[i].resp;
    }
    return (int)all.size();
}

int oracle_distribute(const int* xs, const int* ys, const int* resp, int n, int min_x, int max_x,
                      int min_y, int max_y, int n_target, int* out_idx, int cap) {
    std::vector<Cand> c(n);
    for (int i = 0; i < n; i++) c[i] = {xs[i], ys[i], resp[i]};
    std::vector<int> sel;
    if (n > 0 && (max_y <= min_y || (int)round(static_cast<float>(max_x - min_x) / (max_y - min_y)) < 1)) return -3;
    distribute(c, min_x, max_x, min_y, max_y, n_target, sel);
    int m = std::min((int)sel.size(), cap);
    for (int i = 0; i < m; i++) out_idx[i] = sel[i];
    return (int)sel.size();
}

float oracle_ic_angle(const uint8_t* lvl, size_t stride, int x, int y) {
    oracle_params p = {1000, 1.2f, 8, 20, 7, 0, 0};
    static const Tables t = make_tables(p);
    return ic_angle(lvl, stride, x, y, t.umax);
}

void oracle_descriptor(const uint8_t* blurred, size_t stride, int x, int y, float angle_deg,
                       uint8_t* desc32) {
    orb_descriptor(blurred, stride, x, y, angle_deg, desc32);
}

int oracle_extract(const oracle_params* p, const uint8_t* img, int w, int h, size_t stride,
                   oracle_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
    Tables t = make_tables(*p);
    return extract(*p, t, img, w, h, stride, kps, desc, cap, n_out);
}

int oracle_extract_stages(const oracle_params* p, const uint8_t* imgs, int nframes, int w, int h, size_t row_stride,
                          size_t img_stride, double* stage_s) {
    Tables t = make_tables(*p);
    const int cap = p->nfeatures + 4096;
    std::vector<oracle_keypoint> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    for (int i = 0; i < 6; i++) stage_s[i] = 0;
    for (int f = 0; f < nframes; f++) {
        int n = 0;
        const int rc = extract(*p, t, imgs + (size_t)f * img_stride, w, h, row_stride, kps.data(), desc.data(), cap, &n, stage_s);
        if (rc) return rc;
    }
    return 0;
}

int oracle_extract_batch(const oracle_params* p, const uint8_t* imgs, int nframes, int w, int h,
                         size_t row_stride, size_t img_stride, oracle_keypoint* kps, uint8_t* desc,
                         int cap_per_frame, int* n_out, int threads) {
    Tables t = make_tables(*p);
    if (threads < 1) threads = 1;
    std::vector<int> rc(nframes, 0);
    auto work = [&](int tid) {
        for (int f = tid; f < nframes; f += threads)
            rc[f] = extract(*p, t, imgs + (size_t)f * img_stride, w, h, row_stride,
                            kps + (size_t)f * cap_per_frame, desc + (size_t)f * cap_per_frame * 32,
                            cap_per_frame, &n_out[f]);
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; i++) pool.emplace_back(work, i);
        for (auto& th : pool) th.join();
    }
    for (int f = 0; f < nframes; f++)
        if (rc[f]) return rc[f];
    return 0;
}

// reference src/Frame.cc:603-774, Frame::ComputeStereoMatches.  mvImagePyramid of the two
// extractors is rebuilt here from the two images.  Returns the number of stereo points.
int oracle_compute_stereo_matches(const oracle_params* p, const uint8_t* img_left, const uint8_t* img_right, int w, int h,
                                  size_t stride, const oracle_keypoint* keys, const uint8_t* desc, int N,
                                  const oracle_keypoint* keys_right, const uint8_t* desc_right, int Nr, float mb,
                                  float mbf, float* mvuRight, float* mvDepth) {
    Tables t = make_tables(*p);
    std::vector<Image> pyrL, pyrR;
    build_pyramid(t, img_left, w, h, stride, p->nlevels, pyrL);
    build_pyramid(t, img_right, w, h, stride, p->nlevels, pyrR);
    for (int i = 0; i < N; i++) {
        mvuRight[i] = -1.0f;
        mvDepth[i] = -1.0f;
    }
    const int TH_HIGH = 100, TH_LOW = 50;
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = pyrL[0].h;
    std::vector<std::vector<size_t>> vRowIndices(nRows, std::vector<size_t>());
    for (int iR = 0; iR < Nr; iR++) {
        const oracle_keypoint& kp = keys_right[iR];
        const float& kpY = kp.y;
        if (kp.octave < 0 || kp.octave >= p->nlevels) continue;  // outside mvScaleFactors: undefined in the reference
        const float r = 2.0f * t.scale[keys_right[iR].octave];
        const int maxr = ceil(kpY + r);
        const int minr = floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);
    }
    const float minZ = mb;
    const float minD = 0;
    const float maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    auto popdist = [](const uint8_t* a, const uint8_t* b) {
        int d = 0;
        for (int i = 0; i < 32; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
        return d;
    };
    for (int iL = 0; iL < N; iL++) {
        const oracle_keypoint& kpL = keys[iL];
        const int& levelL = kpL.octave;
        const float& vL = kpL.y;
        const float& uL = kpL.x;
        if (levelL < 0 || levelL >= p->nlevels) continue;
        if (!(vL >= 0.0f) || !(vL < (float)nRows)) continue;  // outside the row table: undefined in the reference
        const std::vector<size_t>& vCandidates = vRowIndices[(size_t)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD;
        const float maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        const uint8_t* dL = desc + (size_t)iL * 32;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            const oracle_keypoint& kpR = keys_right[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float& uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = popdist(dL, desc_right + iR * 32);
                if (dist < bestDist) {
                    bestDist = dist;
                    bestIdxR = iR;
                }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = keys_right[bestIdxR].x;
            const float scaleFactor = t.inv_scale[kpL.octave];
            const float scaleduL = round(kpL.x * scaleFactor);
            const float scaledvL = round(kpL.y * scaleFactor);
            const float scaleduR0 = round(uR0 * scaleFactor);
            const int wnd = 5;
            const Image& IMG_L = pyrL[kpL.octave];
            const Image& IMG_R = pyrR[kpL.octave];
            {
                // windows leaving the level image: cv::Mat range assertion in the reference
                const int r0 = (int)(scaledvL - wnd), cl0 = (int)(scaleduL - wnd), cr0 = (int)(scaleduR0 - wnd);
                const float iniu0 = scaleduR0 + 5 - wnd, endu0 = scaleduR0 + 5 + wnd + 1;
                if (!(iniu0 < 0 || endu0 >= IMG_R.w) &&
                    (r0 < 0 || r0 + 10 >= IMG_L.h || cl0 < 0 || cl0 + 10 >= IMG_L.w || cr0 - 5 < 0 || cr0 + 15 >= IMG_R.w))
                    continue;
                if (iniu0 < 0 || endu0 >= IMG_R.w) continue;
            }
            float IL[11][11];
            {
                const int r0 = (int)(scaledvL - wnd), c0 = (int)(scaleduL - wnd);
                for (int y = 0; y < 11; y++)
                    for (int x = 0; x < 11; x++) IL[y][x] = (float)IMG_L.row(r0 + y)[c0 + x];
                const float c = IL[wnd][wnd];
                for (int y = 0; y < 11; y++)
                    for (int x = 0; x < 11; x++) IL[y][x] = IL[y][x] - c * 1.0f;
            }
            int bestDistSAD = INT_MAX;
            int bestincR = 0;
            const int L = 5;
            std::vector<float> vDists;
            vDists.resize(2 * L + 1);
            const float iniu = scaleduR0 + L - wnd;
            const float endu = scaleduR0 + L + wnd + 1;
            if (iniu < 0 || endu >= IMG_R.w) continue;
            for (int incR = -L; incR <= +L; incR++) {
                float IR[11][11];
                const int r0 = (int)(scaledvL - wnd), c0 = (int)(scaleduR0 + incR - wnd);
                for (int y = 0; y < 11; y++)
                    for (int x = 0; x < 11; x++) IR[y][x] = (float)IMG_R.row(r0 + y)[c0 + x];
                const float c = IR[wnd][wnd];
                float dist = 0;
                for (int y = 0; y < 11; y++)
                    for (int x = 0; x < 11; x++) dist += fabsf(IL[y][x] - (IR[y][x] - c * 1.0f));
                if (dist < bestDistSAD) {
                    bestDistSAD = dist;
                    bestincR = incR;
                }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1];
            const float dist2 = vDists[L + bestincR];
            const float dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = t.scale[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = 0.01;
                    bestuR = uL - 0.01;
                }
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDistSAD, iL));
            }
        }
    }
    if (vDistIdx.empty()) return 0;  // the reference indexes an empty vector here
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    int kept = (int)vDistIdx.size();
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist)
            break;
        else {
            mvuRight[vDistIdx[i].second] = -1;
            mvDepth[vDistIdx[i].second] = -1;
            kept--;
        }
    }
    return kept;
}

}  // extern "C"
