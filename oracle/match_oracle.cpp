// CPU restatement of the reference's keypoint grid and descriptor matchers.
// TEST INFRASTRUCTURE ONLY (parity checker + CPU baseline); see orb_oracle.h.
//
// Everything here follows reference source only (no OpenCV arithmetic is
// involved beyond float compares), but the reference ships no fixtures for
// it: "parity unpinned" in the sense that expected outputs are produced by
// this restatement, not by a run of the reference.

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <map>
#include <thread>
#include <vector>

#include "orb_oracle.h"

namespace {

const int TH_HIGH = 100;      // reference src/ORBmatcher.cc:38
const int TH_LOW = 50;        // :39
const int HISTO_LENGTH = 30;  // :40

// reference src/ORBmatcher.cc:1905-1921
int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

struct Grid {
    float inv_w, inv_h;
    std::vector<std::vector<int>> cells;  // [ix*48+iy]
};

// reference src/Frame.cc:211-212 (cell size), :494-502 (PosInGrid),
// :302-317 (AssignFeaturesToGrid)
void grid_params(const oracle_frame& f, float* inv_w, float* inv_h) {
    *inv_w = static_cast<float>(ORACLE_GRID_COLS) / static_cast<float>(f.max_x - f.min_x);
    *inv_h = static_cast<float>(ORACLE_GRID_ROWS) / static_cast<float>(f.max_y - f.min_y);
}

bool pos_in_grid(const oracle_frame& f, float inv_w, float inv_h, const oracle_keypoint& kp,
                 int& posX, int& posY) {
    posX = round((kp.x - f.min_x) * inv_w);
    posY = round((kp.y - f.min_y) * inv_h);
    if (posX < 0 || posX >= ORACLE_GRID_COLS || posY < 0 || posY >= ORACLE_GRID_ROWS) return false;
    return true;
}

Grid build_grid(const oracle_frame& f) {
    Grid g;
    grid_params(f, &g.inv_w, &g.inv_h);
    g.cells.resize(ORACLE_GRID_COLS * ORACLE_GRID_ROWS);
    for (int i = 0; i < f.n; i++) {
        int px, py;
        if (pos_in_grid(f, g.inv_w, g.inv_h, f.kps[i], px, py))
            g.cells[px * ORACLE_GRID_ROWS + py].push_back(i);
    }
    return g;
}

// reference src/Frame.cc:437-490
std::vector<int> features_in_area(const oracle_frame& f, const Grid& g, float x, float y, float r,
                                  int minLevel = -1, int maxLevel = -1) {
    std::vector<int> out;
    const int nMinCellX = std::max(0, (int)floor((x - f.min_x - r) * g.inv_w));
    if (nMinCellX >= ORACLE_GRID_COLS) return out;
    const int nMaxCellX = std::min((int)ORACLE_GRID_COLS - 1, (int)ceil((x - f.min_x + r) * g.inv_w));
    if (nMaxCellX < 0) return out;
    const int nMinCellY = std::max(0, (int)floor((y - f.min_y - r) * g.inv_h));
    if (nMinCellY >= ORACLE_GRID_ROWS) return out;
    const int nMaxCellY = std::min((int)ORACLE_GRID_ROWS - 1, (int)ceil((y - f.min_y + r) * g.inv_h));
    if (nMaxCellY < 0) return out;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const std::vector<int>& cell = g.cells[ix * ORACLE_GRID_ROWS + iy];
            for (size_t j = 0; j < cell.size(); j++) {
                const oracle_keypoint& kp = f.kps[cell[j]];
                if (bCheckLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0)
                        if (kp.octave > maxLevel) continue;
                }
                const float distx = kp.x - x;
                const float disty = kp.y - y;
                if (fabs(distx) < r && fabs(disty) < r) out.push_back(cell[j]);
            }
        }
    }
    return out;
}

// reference src/ORBmatcher.cc:1849-1894
void three_maxima(const int* sizes, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) {
            max3 = max2;
            max2 = max1;
            max1 = s;
            ind3 = ind2;
            ind2 = ind1;
            ind1 = i;
        } else if (s > max2) {
            max3 = max2;
            max2 = s;
            ind3 = ind2;
            ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

int rot_bin(float a1, float a2) {
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

}  // namespace

extern "C" {

int oracle_hamming256(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

void oracle_grid_build(const oracle_frame* f, int* cell_start, int* cell_items) {
    Grid g = build_grid(*f);
    int pos = 0;
    for (int c = 0; c < ORACLE_GRID_COLS * ORACLE_GRID_ROWS; c++) {
        cell_start[c] = pos;
        for (int idx : g.cells[c]) cell_items[pos++] = idx;
    }
    cell_start[ORACLE_GRID_COLS * ORACLE_GRID_ROWS] = pos;
}

int oracle_features_in_area(const oracle_frame* f, const int* cell_start, const int* cell_items,
                            float x, float y, float r, int min_level, int max_level, int* out,
                            int cap) {
    Grid g;
    grid_params(*f, &g.inv_w, &g.inv_h);
    g.cells.resize(ORACLE_GRID_COLS * ORACLE_GRID_ROWS);
    for (int c = 0; c < ORACLE_GRID_COLS * ORACLE_GRID_ROWS; c++)
        g.cells[c].assign(cell_items + cell_start[c], cell_items + cell_start[c + 1]);
    std::vector<int> v = features_in_area(*f, g, x, y, r, min_level, max_level);
    int n = std::min((int)v.size(), cap);
    for (int i = 0; i < n; i++) out[i] = v[i];
    return (int)v.size();
}

void oracle_three_maxima(const int* hist_sizes, int len, int* ind1, int* ind2, int* ind3) {
    int a = -1, b = -1, c = -1;
    three_maxima(hist_sizes, len, a, b, c);
    *ind1 = a;
    *ind2 = b;
    *ind3 = c;
}

// reference src/ORBmatcher.cc:501-628
int oracle_search_for_initialization(const oracle_frame* f1, const oracle_frame* f2,
                                     float* prev_matched_xy, int* matches12, int window_size,
                                     float nn_ratio, int check_orientation) {
    const oracle_frame& F1 = *f1;
    const oracle_frame& F2 = *f2;
    Grid g2 = build_grid(F2);
    int nmatches = 0;
    for (int i = 0; i < F1.n; i++) matches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    std::vector<int> vMatchedDistance(F2.n, INT_MAX);
    std::vector<int> vnMatches21(F2.n, -1);
    for (int i1 = 0; i1 < F1.n; i1++) {
        const oracle_keypoint& kp1 = F1.kps[i1];
        int level1 = kp1.octave;
        if (level1 > 0) continue;
        std::vector<int> vIndices2 =
            features_in_area(F2, g2, prev_matched_xy[2 * i1], prev_matched_xy[2 * i1 + 1],
                             (float)window_size, level1, level1);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = F1.desc + (size_t)i1 * 32;
        int bestDist = INT_MAX;
        int bestDist2 = INT_MAX;
        int bestIdx2 = -1;
        for (int i2 : vIndices2) {
            const uint8_t* d2 = F2.desc + (size_t)i2 * 32;
            int dist = descriptor_distance(d1, d2);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nn_ratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    matches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                matches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) {
                    int bin = rot_bin(F1.kps[i1].angle, F2.kps[bestIdx2].angle);
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                int idx1 = rotHist[i][j];
                if (matches12[idx1] >= 0) {
                    matches12[idx1] = -1;
                    nmatches--;
                }
            }
        }
    }
    for (int i1 = 0; i1 < F1.n; i1++)
        if (matches12[i1] >= 0) {
            prev_matched_xy[2 * i1] = F2.kps[matches12[i1]].x;
            prev_matched_xy[2 * i1 + 1] = F2.kps[matches12[i1]].y;
        }
    return nmatches;
}

// reference src/ORBmatcher.cc:1533-1678 from the point where the projection
// (u, v, invzc) of each last-frame map point is known (:1577-1578).
int oracle_search_by_projection_frames(const oracle_frame* cur, const oracle_frame* last, const uint8_t* mp_desc,
                                       const uint8_t* valid, const float* u_, const float* v_,
                                       const float* invz, const uint8_t* has_obs,
                                       const float* cur_u_right, float bf, float th,
                                       int direction, float nn_ratio, int check_orientation,
                                       int* cur_claimed, uint8_t* cur_claimed_obs, int* query_best) {
    (void)nn_ratio;
    if (query_best)
        for (int i = 0; i < last->n; i++) query_best[i] = -1;
    const oracle_frame& C = *cur;
    const oracle_frame& L = *last;
    Grid g = build_grid(C);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const bool bForward = direction == 1;
    const bool bBackward = direction == 2;
    for (int i = 0; i < L.n; i++) {
        if (!valid[i]) continue;
        const float invzc = invz[i];
        if (invzc < 0) continue;
        float u = u_[i];
        float v = v_[i];
        if (u < C.min_x || u > C.max_x) continue;
        if (v < C.min_y || v > C.max_y) continue;
        int nLastOctave = L.kps[i].octave;
        float radius = th * C.scale_factors[nLastOctave];
        std::vector<int> vIndices2;
        if (bForward)
            vIndices2 = features_in_area(C, g, u, v, radius, nLastOctave);
        else if (bBackward)
            vIndices2 = features_in_area(C, g, u, v, radius, 0, nLastOctave);
        else
            vIndices2 = features_in_area(C, g, u, v, radius, nLastOctave - 1, nLastOctave + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = (mp_desc ? mp_desc : L.desc) + (size_t)i * 32;
        int bestDist = 256;
        int bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (cur_claimed[i2] >= 0)
                if (cur_claimed_obs[i2]) continue;
            if (cur_u_right[i2] > 0) {
                const float ur = u - bf * invzc;
                const float er = fabs(ur - cur_u_right[i2]);
                if (er > radius) continue;
            }
            const int dist = descriptor_distance(dMP, C.desc + (size_t)i2 * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= TH_HIGH) {
            cur_claimed[bestIdx2] = i;
            cur_claimed_obs[bestIdx2] = has_obs[i];
            nmatches++;
            if (query_best) query_best[i] = bestIdx2;  // perfect/src/ORBmatcher.cc:1835-1836 pushes the pair here
            if (check_orientation) {
                int bin = rot_bin(L.kps[i].angle, C.kps[bestIdx2].angle);
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    cur_claimed[rotHist[i][j]] = -1;
                    cur_claimed_obs[rotHist[i][j]] = 0;
                    nmatches--;
                }
            }
        }
    }
    return nmatches;
}

// reference src/ORBmatcher.cc:62-164: SearchByProjection(Frame&, const vector<MapPoint*>&, th).
// Per map point: valid = mbTrackInView && !isBad(), the isInFrustum outputs, its descriptor
// and whether it has observations.  claimed[] plays F.mvpMapPoints.
// Pose algebra in front of the frame-to-frame search (reference src/ORBmatcher.cc:1545-1556 and
// :1570-1578): camera motion between the frames decides forward / backward, and every last-frame
// map point is projected into the current frame.  A cv::Mat product of a 3x3 by a 3x1 float matrix
// is the float sum a0*b0 + a1*b1 + a2*b2 taken left to right, the added translation rounds once
// more (OpenCV's small-matrix product; this file is built without contraction).  Tcw / Tlw: 4x4
// row-major; x3Dw: n x 3.  direction: 0 neither, 1 forward, 2 backward.
void oracle_project_last_frame(const float* Tcw, const float* Tlw, const float* x3Dw, int n, float fx, float fy, float cx,
                               float cy, float mb, int bMono, float* u, float* v, float* invz, int* direction) {
    float Rcw[3][3], tcw[3], Rlw[3][3], tlw[3];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) {
            Rcw[r][c] = Tcw[4 * r + c];
            Rlw[r][c] = Tlw[4 * r + c];
        }
        tcw[r] = Tcw[4 * r + 3];
        tlw[r] = Tlw[4 * r + 3];
    }
    float twc[3], tlc[3];
    for (int i = 0; i < 3; i++) {  // twc = -Rcw.t()*tcw
        const float p = Rcw[0][i] * tcw[0] + Rcw[1][i] * tcw[1] + Rcw[2][i] * tcw[2];
        twc[i] = -p;
    }
    for (int i = 0; i < 3; i++) {  // tlc = Rlw*twc + tlw
        const float p = Rlw[i][0] * twc[0] + Rlw[i][1] * twc[1] + Rlw[i][2] * twc[2];
        tlc[i] = p + tlw[i];
    }
    const bool bForward = tlc[2] > mb && !bMono;
    const bool bBackward = -tlc[2] > mb && !bMono;
    *direction = bForward ? 1 : (bBackward ? 2 : 0);
    for (int k = 0; k < n; k++) {
        const float* X = x3Dw + 3 * k;
        float x3Dc[3];
        for (int i = 0; i < 3; i++) {  // x3Dc = Rcw*x3Dw + tcw
            const float p = Rcw[i][0] * X[0] + Rcw[i][1] * X[1] + Rcw[i][2] * X[2];
            x3Dc[i] = p + tcw[i];
        }
        const float xc = x3Dc[0];
        const float yc = x3Dc[1];
        const float invzc = 1.0 / x3Dc[2];
        invz[k] = invzc;
        u[k] = fx * xc * invzc + cx;
        v[k] = fy * yc * invzc + cy;
    }
}

int oracle_search_by_projection_mappoints(const oracle_frame* f, const float* f_u_right, int nmp,
                                          const uint8_t* valid, const float* proj_x, const float* proj_y,
                                          const float* proj_xr, const int* level, const float* view_cos,
                                          const uint8_t* desc, const uint8_t* has_obs, float th, float nn_ratio,
                                          int* claimed, uint8_t* claimed_obs) {
    const oracle_frame& F = *f;
    Grid g = build_grid(F);
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (int iMP = 0; iMP < nmp; iMP++) {
        if (!valid[iMP]) continue;
        const int nPredictedLevel = level[iMP];
        float r = view_cos[iMP] > 0.998 ? 2.5 : 4.0;
        if (bFactor) r *= th;
        const std::vector<int> vIndices = features_in_area(F, g, proj_x[iMP], proj_y[iMP],
                                                           r * F.scale_factors[nPredictedLevel],
                                                           nPredictedLevel - 1, nPredictedLevel);
        if (vIndices.empty()) continue;
        const uint8_t* MPdescriptor = desc + (size_t)iMP * 32;
        int bestDist = 256;
        int bestLevel = -1;
        int bestDist2 = 256;
        int bestLevel2 = -1;
        int bestIdx = -1;
        for (int idx : vIndices) {
            if (claimed[idx] >= 0)
                if (claimed_obs[idx]) continue;
            if (f_u_right[idx] > 0) {
                const float er = fabs(proj_xr[iMP] - f_u_right[idx]);
                if (er > r * F.scale_factors[nPredictedLevel]) continue;
            }
            const int dist = descriptor_distance(MPdescriptor, F.desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = F.kps[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = F.kps[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nn_ratio * bestDist2) continue;
            claimed[bestIdx] = iMP;
            claimed_obs[bestIdx] = has_obs[iMP];
            nmatches++;
        }
    }
    return nmatches;
}

// reference src/ORBmatcher.cc:1694-1836 from the point where the keyframe's map point i is
// projected to (u, v) and its level predicted (:1730-1754).  valid folds :1715-1719 and the
// distance check :1748.  claimed[] plays CurrentFrame.mvpMapPoints.
int oracle_search_by_projection_keyframe(const oracle_frame* cur, int n, const uint8_t* valid, const float* u_,
                                         const float* v_, const int* level, const uint8_t* desc,
                                         const float* kf_angle, float th, int orb_dist, int check_orientation,
                                         int* claimed) {
    const oracle_frame& C = *cur;
    Grid g = build_grid(C);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < n; i++) {
        if (!valid[i]) continue;
        const float u = u_[i], v = v_[i];
        if (u < C.min_x || u > C.max_x) continue;
        if (v < C.min_y || v > C.max_y) continue;
        int nPredictedLevel = level[i];
        const float radius = th * C.scale_factors[nPredictedLevel];
        const std::vector<int> vIndices2 = features_in_area(C, g, u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = desc + (size_t)i * 32;
        int bestDist = 256;
        int bestIdx2 = -1;
        for (int i2 : vIndices2) {
            if (claimed[i2] >= 0) continue;
            const int dist = descriptor_distance(dMP, C.desc + (size_t)i2 * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= orb_dist) {
            claimed[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                int bin = rot_bin(kf_angle[i], C.kps[bestIdx2].angle);
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    claimed[rotHist[i][j]] = -1;
                    nmatches--;
                }
            }
        }
    }
    return nmatches;
}

// reference src/ORBmatcher.cc:370-485 from the point where candidate map point i is projected
// with the similarity transform to (u, v) and its level predicted (:411-442).  valid folds
// :403, :413, :434, :438.  matched[] plays vpMatched (index of the map point or -1).
int oracle_search_by_projection_sim3(const oracle_frame* kf, int n, const uint8_t* valid, const float* u_,
                                     const float* v_, const int* level, const uint8_t* desc, int th, int* matched) {
    const oracle_frame& K = *kf;
    Grid g = build_grid(K);
    int nmatches = 0;
    for (int iMP = 0; iMP < n; iMP++) {
        if (!valid[iMP]) continue;
        const float u = u_[iMP], v = v_[iMP];
        if (!(u >= K.min_x && u < K.max_x && v >= K.min_y && v < K.max_y)) continue;  // KeyFrame::IsInImage
        int nPredictedLevel = level[iMP];
        const float radius = th * K.scale_factors[nPredictedLevel];
        const std::vector<int> vIndices = features_in_area(K, g, u, v, radius);  // KeyFrame::GetFeaturesInArea: no levels
        if (vIndices.empty()) continue;
        const uint8_t* dMP = desc + (size_t)iMP * 32;
        int bestDist = 256;
        int bestIdx = -1;
        for (int idx : vIndices) {
            if (matched[idx] >= 0) continue;
            const int kpLevel = K.kps[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int dist = descriptor_distance(dMP, K.desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = idx;
            }
        }
        if (bestDist <= TH_LOW) {
            matched[bestIdx] = iMP;
            nmatches++;
        }
    }
    return nmatches;
}

// Serial scan of src/ORBmatcher.cc:255-282 over a whole train set.
void oracle_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* best_idx, int* best_d,
                 int* second_d) {
    for (int i = 0; i < nq; i++) {
        int bestDist1 = 256, bestIdx = -1, bestDist2 = 256;
        for (int j = 0; j < nt; j++) {
            const int dist = descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (dist < bestDist1) {
                bestDist2 = bestDist1;
                bestDist1 = dist;
                bestIdx = j;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        best_idx[i] = bestIdx;
        best_d[i] = bestDist1;
        second_d[i] = bestDist2;
    }
}

// DBoW2 TemplatedVocabulary<FORB::TDescriptor, FORB>::transform, restated from the published
// library (DBoW2/TemplatedVocabulary.h: transform(feature, word_id, weight, nid, levelsup) and
// transform(features, BowVector, FeatureVector, levelsup)); the reference calls it at
// src/Frame.cc:513-522 but does not vendor it.  The tree comes as arrays: children of node i are
// children[child_start[i] .. child_start[i+1]) in the order of m_nodes[i].children.
void oracle_bow_transform(int n_nodes, int levels, const uint8_t* node_desc, const int* child_start, const int* children,
                          const int* word_id, const double* weight, const uint8_t* desc, int n, int levelsup, int* word_out,
                          double* weight_out, int* node_out) {
    (void)n_nodes;
    for (int i_feature = 0; i_feature < n; i_feature++) {
        const uint8_t* feature = desc + (size_t)i_feature * 32;
        const int nid_level = levels - levelsup;
        int nid = 0;  // root when nid_level <= 0
        int final_id = 0;
        int current_level = 0;
        do {
            ++current_level;
            const int* nodes = children + child_start[final_id];
            const int nnodes = child_start[final_id + 1] - child_start[final_id];
            final_id = nodes[0];
            double best_d = descriptor_distance(feature, node_desc + (size_t)final_id * 32);
            for (int k = 1; k < nnodes; k++) {
                const int id = nodes[k];
                const double d = descriptor_distance(feature, node_desc + (size_t)id * 32);
                if (d < best_d) {
                    best_d = d;
                    final_id = id;
                }
            }
            if (current_level == nid_level) nid = final_id;
        } while (child_start[final_id + 1] > child_start[final_id]);  // !isLeaf()
        word_out[i_feature] = word_id[final_id];
        weight_out[i_feature] = weight[final_id];
        node_out[i_feature] = nid;
    }
}

// BowVector of the TF / TF_IDF weightings with L1 scoring (the ORB vocabulary): addWeight per
// feature in feature order for weights > 0, then normalize(L1) in ascending word order.  ids / vals:
// room for n entries; returns the number of words.
int oracle_bow_vector(const int* word, const double* weight, int n, int* ids, double* vals) {
    std::map<int, double> v;
    for (int i = 0; i < n; i++) {
        if (weight[i] > 0) {
            std::map<int, double>::iterator vit = v.lower_bound(word[i]);
            if (vit != v.end() && !(word[i] < vit->first))
                vit->second += weight[i];
            else
                v.insert(vit, std::make_pair(word[i], weight[i]));
        }
    }
    double norm = 0.0;
    for (std::map<int, double>::iterator it = v.begin(); it != v.end(); ++it) norm += fabs(it->second);
    if (norm > 0.0)
        for (std::map<int, double>::iterator it = v.begin(); it != v.end(); ++it) it->second /= norm;
    int k = 0;
    for (std::map<int, double>::iterator it = v.begin(); it != v.end(); ++it, ++k) {
        ids[k] = it->first;
        vals[k] = it->second;
    }
    return k;
}

// The same scan with the queries split over host threads (every query is still one serial scan).
void oracle_knn2_mt(const uint8_t* q, int nq, const uint8_t* t, int nt, int* best_idx, int* best_d, int* second_d,
                    int threads) {
    threads = std::max(1, std::min(threads, nq));
    std::vector<std::thread> pool;
    for (int k = 0; k < threads; k++) {
        const int i0 = (int)((long long)nq * k / threads), i1 = (int)((long long)nq * (k + 1) / threads);
        pool.emplace_back([=]() {
            oracle_knn2(q + (size_t)i0 * 32, i1 - i0, t, nt, best_idx + i0, best_d + i0, second_d + i0);
        });
    }
    for (auto& th : pool) th.join();
}

// Shared body of the two SearchByBoW overloads (reference src/ORBmatcher.cc:209-355 for
// keyframe -> frame, :642-788 for keyframe -> keyframe).  valid1[idx1] folds "has a map point
// that is not bad" (:248-251, :684-688); valid2[idx2] (NULL = all) folds the same test on the
// second keyframe (:701-704); claimed second-side features are vpMapPointMatches / vbMatched2.
// max_dist is inclusive: TH_LOW for :284, TH_LOW - 1 for :721.
int oracle_search_by_bow(const oracle_frame* f1, const oracle_feature_vector* fv1, const uint8_t* valid1,
                         const oracle_frame* f2, const oracle_feature_vector* fv2, const uint8_t* valid2, int max_dist,
                         float mfNNratio, int mbCheckOrientation, int* match12, int* match21) {
    for (int i = 0; i < f1->n; i++) match12[i] = -1;
    for (int i = 0; i < f2->n; i++) match21[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    int it1 = 0, it2 = 0;
    while (it1 < fv1->n_nodes && it2 < fv2->n_nodes) {
        if (fv1->node_ids[it1] == fv2->node_ids[it2]) {
            for (int a = fv1->start[it1]; a < fv1->start[it1 + 1]; a++) {
                const unsigned int idx1 = fv1->items[a];
                if (!valid1[idx1]) continue;
                const uint8_t* d1 = f1->desc + (size_t)idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int b = fv2->start[it2]; b < fv2->start[it2 + 1]; b++) {
                    const unsigned int idx2 = fv2->items[b];
                    if (match21[idx2] >= 0) continue;
                    if (valid2 && !valid2[idx2]) continue;
                    const int dist = descriptor_distance(d1, f2->desc + (size_t)idx2 * 32);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdx2 = idx2;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 <= max_dist) {
                    if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
                        match12[idx1] = bestIdx2;
                        match21[bestIdx2] = idx1;
                        if (mbCheckOrientation) rotHist[rot_bin(f1->kps[idx1].angle, f2->kps[bestIdx2].angle)].push_back(idx1);
                        nmatches++;
                    }
                }
            }
            it1++;
            it2++;
        } else if (fv1->node_ids[it1] < fv2->node_ids[it2]) {
            it1 = std::lower_bound(fv1->node_ids, fv1->node_ids + fv1->n_nodes, fv2->node_ids[it2]) - fv1->node_ids;
        } else {
            it2 = std::lower_bound(fv2->node_ids, fv2->node_ids + fv2->n_nodes, fv1->node_ids[it1]) - fv2->node_ids;
        }
    }
    if (mbCheckOrientation) {
        int sizes[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i]) {
                match21[match12[idx1]] = -1;
                match12[idx1] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// reference src/ORBmatcher.cc:170-190 (CheckDistEpipolarLine)
static bool check_dist_epipolar_line(const oracle_keypoint& kp1, const oracle_keypoint& kp2, const float* F12,
                                     const float* sigma2) {
    const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
    const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
    const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
    const float num = a * kp2.x + b * kp2.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2[kp2.octave];
}

// reference src/ORBmatcher.cc:803-992 from the point where the epipole (ex, ey) and F12 are
// known (:809-818 is pose algebra).  has_mp*: the keypoint already holds a map point
// (:858, :882); u_right*: mvuRight.  F12 row-major 3x3.  match12 = vMatches12.
int oracle_search_for_triangulation(const oracle_frame* f1, const oracle_feature_vector* fv1, const uint8_t* has_mp1,
                                    const float* u_right1, const oracle_frame* f2, const oracle_feature_vector* fv2,
                                    const uint8_t* has_mp2, const float* u_right2, const float* F12, float ex, float ey,
                                    const float* level_sigma2, int bOnlyStereo, int mbCheckOrientation, int* vMatches12) {
    int nmatches = 0;
    std::vector<bool> vbMatched2(f2->n, false);
    for (int i = 0; i < f1->n; i++) vMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    int it1 = 0, it2 = 0;
    while (it1 < fv1->n_nodes && it2 < fv2->n_nodes) {
        if (fv1->node_ids[it1] == fv2->node_ids[it2]) {
            for (int a = fv1->start[it1]; a < fv1->start[it1 + 1]; a++) {
                const size_t idx1 = fv1->items[a];
                if (has_mp1[idx1]) continue;
                const bool bStereo1 = u_right1[idx1] >= 0;
                if (bOnlyStereo)
                    if (!bStereo1) continue;
                const oracle_keypoint& kp1 = f1->kps[idx1];
                const uint8_t* d1 = f1->desc + idx1 * 32;
                int bestDist = TH_LOW;
                int bestIdx2 = -1;
                for (int b = fv2->start[it2]; b < fv2->start[it2 + 1]; b++) {
                    size_t idx2 = fv2->items[b];
                    if (vbMatched2[idx2] || has_mp2[idx2]) continue;
                    const bool bStereo2 = u_right2[idx2] >= 0;
                    if (bOnlyStereo)
                        if (!bStereo2) continue;
                    const int dist = descriptor_distance(d1, f2->desc + idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const oracle_keypoint& kp2 = f2->kps[idx2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - kp2.x;
                        const float distey = ey - kp2.y;
                        if (distex * distex + distey * distey < 100 * f2->scale_factors[kp2.octave]) continue;
                    }
                    if (check_dist_epipolar_line(kp1, kp2, F12, level_sigma2)) {
                        bestIdx2 = idx2;
                        bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {
                    const oracle_keypoint& kp2 = f2->kps[bestIdx2];
                    vMatches12[idx1] = bestIdx2;  // the reference never sets vbMatched2 here
                    nmatches++;
                    if (mbCheckOrientation) rotHist[rot_bin(kp1.angle, kp2.angle)].push_back((int)idx1);
                }
            }
            it1++;
            it2++;
        } else if (fv1->node_ids[it1] < fv2->node_ids[it2]) {
            it1 = std::lower_bound(fv1->node_ids, fv1->node_ids + fv1->n_nodes, fv2->node_ids[it2]) - fv1->node_ids;
        } else {
            it2 = std::lower_bound(fv2->node_ids, fv2->node_ids + fv2->n_nodes, fv1->node_ids[it1]) - fv2->node_ids;
        }
    }
    if (mbCheckOrientation) {
        int sizes[HISTO_LENGTH], ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        three_maxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i]) {
                vMatches12[idx1] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// The candidate loop shared by Fuse (reference src/ORBmatcher.cc:1066-1120), the Sim3 form of
// Fuse (:1233-1262) and both directions of SearchBySim3 (:1391-1416, :1462-1489): query i is a
// map point already projected to (u, v) with predicted level `level`; the closest descriptor
// among the keypoints in the window whose octave is level-1 or level wins (first one on ties).
// chi2_gate = 1 adds Fuse's reprojection test (:1085-1111): kf_u_right = mvuRight of the
// keyframe, q_u_right = u - bf*invz of the query, inv_level_sigma2 = mvInvLevelSigma2.
// best_dist is 256 when nothing qualified.
void oracle_window_best(const oracle_frame* kf, int n, const uint8_t* valid, const float* u_, const float* v_,
                        const float* radius_, const int* level, const uint8_t* desc, int chi2_gate,
                        const float* kf_u_right, const float* q_u_right, const float* inv_level_sigma2, int* best_idx,
                        int* best_dist) {
    const oracle_frame& K = *kf;
    Grid g = build_grid(K);
    for (int i = 0; i < n; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
        if (!valid[i]) continue;
        const float u = u_[i], v = v_[i];
        const int nPredictedLevel = level[i];
        const std::vector<int> vIndices = features_in_area(K, g, u, v, radius_[i]);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = desc + (size_t)i * 32;
        int bestDist = 256;
        int bestIdx = -1;
        for (int idx : vIndices) {
            const oracle_keypoint& kp = K.kps[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (chi2_gate) {
                if (kf_u_right[idx] >= 0) {
                    const float ur = q_u_right[i];
                    const float ex = u - kp.x;
                    const float ey = v - kp.y;
                    const float er = ur - kf_u_right[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = u - kp.x;
                    const float ey = v - kp.y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = descriptor_distance(dMP, K.desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = idx;
            }
        }
        best_idx[i] = bestIdx;
        best_dist[i] = bestDist;
    }
}

// reference src/ORBmatcher.cc:1298-1517 from the projections on: q12 are the map points of
// keyframe 1 projected into keyframe 2 (one per keypoint of keyframe 1; valid folds :1355-1385),
// q21 the reverse.  match12[i1] = keypoint of keyframe 2 when both directions agree (:1499-1514).
int oracle_search_by_sim3(const oracle_frame* kf1, const oracle_frame* kf2, const uint8_t* valid12, const float* u12,
                          const float* v12, const float* r12, const int* level12, const uint8_t* desc1,
                          const uint8_t* valid21, const float* u21, const float* v21, const float* r21,
                          const int* level21, const uint8_t* desc2, int* match12) {
    std::vector<int> vnMatch1(kf1->n, -1), vnMatch2(kf2->n, -1), d1(kf1->n), d2(kf2->n);
    oracle_window_best(kf2, kf1->n, valid12, u12, v12, r12, level12, desc1, 0, nullptr, nullptr, nullptr, vnMatch1.data(),
                       d1.data());
    oracle_window_best(kf1, kf2->n, valid21, u21, v21, r21, level21, desc2, 0, nullptr, nullptr, nullptr, vnMatch2.data(),
                       d2.data());
    for (int i = 0; i < kf1->n; i++)
        if (d1[i] > TH_HIGH) vnMatch1[i] = -1;
    for (int i = 0; i < kf2->n; i++)
        if (d2[i] > TH_HIGH) vnMatch2[i] = -1;
    int nFound = 0;
    for (int i1 = 0; i1 < kf1->n; i1++) {
        match12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            const int idx1 = vnMatch2[idx2];
            if (idx1 == i1) {
                match12[i1] = idx2;
                nFound++;
            }
        }
    }
    return nFound;
}

// MapPoint::ComputeDistinctiveDescriptors for a batch of map points (reference
// src/MapPoint.cc:307-340): group g owns the descriptors group_start[g] .. group_start[g+1]
// (vDescriptors of one map point, bad keyframes already skipped); best[g] = BestIdx within the
// group, -1 for an empty group (:303-304 returns without touching the descriptor).
void oracle_distinctive_descriptors(const uint8_t* desc, const int* group_start, int ngroups, int* best, int* best_median) {
    for (int g = 0; g < ngroups; g++) {
        const uint8_t* D = desc + (size_t)group_start[g] * 32;
        const size_t N = (size_t)(group_start[g + 1] - group_start[g]);
        best[g] = -1;
        if (best_median) best_median[g] = INT_MAX;
        if (N == 0) continue;
        std::vector<float> Distances(N * N);
        for (size_t i = 0; i < N; i++) {
            Distances[i * N + i] = 0;
            for (size_t j = i + 1; j < N; j++) {
                int distij = descriptor_distance(D + i * 32, D + j * 32);
                Distances[i * N + j] = distij;
                Distances[j * N + i] = distij;
            }
        }
        int BestMedian = INT_MAX;
        int BestIdx = 0;
        for (size_t i = 0; i < N; i++) {
            std::vector<int> vDists(Distances.begin() + i * N, Distances.begin() + i * N + N);
            std::sort(vDists.begin(), vDists.end());
            int median = vDists[0.5 * (N - 1)];
            if (median < BestMedian) {
                BestMedian = median;
                BestIdx = i;
            }
        }
        best[g] = BestIdx;
        if (best_median) best_median[g] = BestMedian;
    }
}

// Dynamic-object keypoint filter of the RGB-D constructor (reference perfect/src/Frame.cc:357-374):
// when the mask covers more than 65 % of the image (sum of its bytes), keep only keypoints
// whose mask byte is 1, in order; otherwise everything stays.  Returns the new count.
int oracle_filter_by_mask(const oracle_keypoint* kps, const uint8_t* desc, int n, const uint8_t* mask, int w, int h,
                          size_t stride, oracle_keypoint* out_kps, uint8_t* out_desc) {
    double s = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) s += mask[(size_t)y * stride + x];
    int m = 0;
    if (s > h * w * 0.65) {
        for (int i = 0; i < n; i++) {
            int val = (int)mask[(size_t)(int)kps[i].y * stride + (int)kps[i].x];
            if (val == 1) {
                out_kps[m] = kps[i];
                memcpy(out_desc + (size_t)m * 32, desc + (size_t)i * 32, 32);
                m++;
            }
        }
    } else {
        for (int i = 0; i < n; i++) {
            out_kps[i] = kps[i];
            memcpy(out_desc + (size_t)i * 32, desc + (size_t)i * 32, 32);
        }
        m = n;
    }
    return m;
}

// Keypoint records of the binary map file (reference perfect/src/Map.cc:361-380 write,
// :169-187 read): per keypoint pt.x, pt.y, size, angle, response (float), octave (int), the 32
// descriptor bytes and the map point index (unsigned long, ULONG_MAX = none): 64 bytes.
void oracle_pack_keyframe_records(const oracle_keypoint* kps, const uint8_t* desc, const uint64_t* mp_index, int n,
                                  uint8_t* records) {
    uint8_t* p = records;
    for (int i = 0; i < n; i++) {
        memcpy(p, &kps[i].x, 4); p += 4;
        memcpy(p, &kps[i].y, 4); p += 4;
        memcpy(p, &kps[i].size, 4); p += 4;
        memcpy(p, &kps[i].angle, 4); p += 4;
        memcpy(p, &kps[i].response, 4); p += 4;
        memcpy(p, &kps[i].octave, 4); p += 4;
        for (int j = 0; j < 32; j++) *p++ = desc[(size_t)i * 32 + j];
        const uint64_t idx = mp_index ? mp_index[i] : ~0ull;
        memcpy(p, &idx, 8); p += 8;
    }
}

void oracle_unpack_keyframe_records(const uint8_t* records, int n, oracle_keypoint* kps, uint8_t* desc, uint64_t* mp_index) {
    const uint8_t* p = records;
    for (int i = 0; i < n; i++) {
        oracle_keypoint kp;
        kp.class_id = -1;  // cv::KeyPoint's default
        memcpy(&kp.x, p, 4); p += 4;
        memcpy(&kp.y, p, 4); p += 4;
        memcpy(&kp.size, p, 4); p += 4;
        memcpy(&kp.angle, p, 4); p += 4;
        memcpy(&kp.response, p, 4); p += 4;
        memcpy(&kp.octave, p, 4); p += 4;
        kps[i] = kp;
        for (int j = 0; j < 32; j++) desc[(size_t)i * 32 + j] = *p++;
        memcpy(&mp_index[i], p, 8); p += 8;
    }
}

}  // extern "C"
