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

}  // extern "C"
