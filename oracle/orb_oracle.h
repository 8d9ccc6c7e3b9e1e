/* CPU restatement of the reference ORB front-end (extraction + matching).
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker for the HIP
 * path and the CPU baseline of the benchmark.  Nothing in the product
 * (slam_amd/, include/) may call, link or load it.
 *
 * Parity status: the reference ships no golden vectors, tests or fixtures for
 * this path and its OpenCV dependency is absent, so the OpenCV-side semantics
 * (resize, FAST, GaussianBlur, fastAtan2, cvRound) are restated from their
 * published algorithms: "parity unpinned" at that boundary.  Everything that
 * is reference source (ORBextractor.cc, ORBmatcher.cc, Frame.cc) is followed
 * line by line and cited at each function.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Layout-identical to cv::KeyPoint (28 bytes). */
typedef struct oracle_keypoint {
    float x, y;
    float size;
    float angle;
    float response;
    int octave;
    int class_id;
} oracle_keypoint;

typedef struct oracle_params {
    int nfeatures;
    float scale_factor;
    int nlevels;
    int ini_th_fast;
    int min_th_fast;
    int blur_variant; /* 0: integer kernel summing to 257; 1: kernel summing to 256; 2: the 257 kernel with the
                         single-precision column pass of OpenCV <= 3.4.0 (round half to even) */
    int fma_variant;  /* 0: x*b + y*a with every product rounded; 1: first product fused into the sum */
} oracle_params;

/* ---- constructor tables: reference src/ORBextractor.cc:399-466 ---- */
void oracle_tables(const oracle_params* p, float* scale, float* inv_scale, float* sigma2,
                   float* inv_sigma2, int* quotas, int* umax16);
void oracle_level_size(const oracle_params* p, int w, int h, int level, int* lw, int* lh);

/* ---- restated OpenCV primitives ---- */
int oracle_cv_round(double v);
/* host libm sinf/cosf, as the reference's descriptor code calls them */
void oracle_sincosf(const float* a, int n, float* s, float* c);
float oracle_fast_atan2(float y, float x);
void oracle_resize_linear(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw,
                          int dh, size_t dstride);
void oracle_gaussian7(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst,
                      size_t dstride, int variant);
/* FAST-9/16 with non-max suppression on a standalone ROI; returns count (raster order). */
int oracle_fast9_nms(const uint8_t* roi, int w, int h, size_t stride, int threshold, int* xs,
                     int* ys, int* scores, int cap);

/* ---- extractor stages ---- */
/* Pyramid level `level` of the image, tightly packed into dst (lw*lh bytes). */
void oracle_pyramid_level(const oracle_params* p, const uint8_t* img, int w, int h, size_t stride,
                          int level, uint8_t* dst);
/* Per-cell FAST candidates of one level image, in the reference's append order.
 * Coordinates are relative to the detection origin (16,16), as in the reference. */
int oracle_detect_cells(const uint8_t* lvl, int w, int h, size_t stride, int ini_th, int min_th,
                        int* xs, int* ys, int* resp, int cap);
/* DistributeOctTree.  Returns number of selected keys; out_idx = indices into the input. */
int oracle_distribute(const int* xs, const int* ys, const int* resp, int n, int min_x, int max_x,
                      int min_y, int max_y, int n_target, int* out_idx, int cap);
float oracle_ic_angle(const uint8_t* lvl, size_t stride, int x, int y);
void oracle_descriptor(const uint8_t* blurred, size_t stride, int x, int y, float angle_deg,
                       uint8_t* desc32);

/* Whole operator(): returns 0, number of keypoints in *n_out. */
int oracle_extract(const oracle_params* p, const uint8_t* img, int w, int h, size_t stride,
                   oracle_keypoint* kps, uint8_t* desc, int cap, int* n_out);
/* Frame-parallel batch over `threads` std::threads; same per-frame output layout. */
/* Single-threaded extraction of nframes images with wall-clock seconds accumulated per stage:
 * stage_s[6] = pyramid, FAST cells, quadtree, orientation, blur, descriptors. */
int oracle_extract_stages(const oracle_params* p, const uint8_t* imgs, int nframes, int w, int h, size_t row_stride,
                          size_t img_stride, double* stage_s);

int oracle_extract_batch(const oracle_params* p, const uint8_t* imgs, int nframes, int w, int h,
                         size_t row_stride, size_t img_stride, oracle_keypoint* kps, uint8_t* desc,
                         int cap_per_frame, int* n_out, int threads);

/* Frame::ComputeStereoMatches (reference src/Frame.cc:603-774) on a rectified pair. */
int oracle_compute_stereo_matches(const oracle_params* p, const uint8_t* img_left, const uint8_t* img_right, int w, int h,
                                  size_t stride, const oracle_keypoint* keys, const uint8_t* desc, int n,
                                  const oracle_keypoint* keys_right, const uint8_t* desc_right, int n_right, float mb,
                                  float mbf, float* u_right, float* depth);

/* ---- grid + matcher: reference src/Frame.cc, src/ORBmatcher.cc ---- */
#define ORACLE_GRID_COLS 64
#define ORACLE_GRID_ROWS 48

typedef struct oracle_frame {
    const oracle_keypoint* kps; /* undistorted keypoints (mvKeysUn) */
    const uint8_t* desc;        /* n x 32 */
    int n;
    float min_x, min_y, max_x, max_y; /* image bounds mnMinX.. */
    const float* scale_factors;       /* mvScaleFactors */
    int nlevels;
} oracle_frame;

int oracle_hamming256(const uint8_t* a, const uint8_t* b);
/* CSR grid: cell_start[64*48+1] (column-major cell = ix*48+iy), cell_items[n]. */
void oracle_grid_build(const oracle_frame* f, int* cell_start, int* cell_items);
int oracle_features_in_area(const oracle_frame* f, const int* cell_start, const int* cell_items,
                            float x, float y, float r, int min_level, int max_level, int* out,
                            int cap);
void oracle_three_maxima(const int* hist_sizes, int len, int* ind1, int* ind2, int* ind3);

int oracle_search_for_initialization(const oracle_frame* f1, const oracle_frame* f2,
                                     float* prev_matched_xy, int* matches12, int window_size,
                                     float nn_ratio, int check_orientation);

/* SearchByProjection(Frame& cur, const Frame& last, th, bMono) with the pose
 * algebra already applied by the caller: for last-frame keypoint i the caller
 * passes valid[i] (map point present and not an outlier), the projection
 * (u[i], v[i]), inverse depth invz[i] and has_obs[i] (Observations() > 0).
 * cur_claimed[] in/out: -1 free, otherwise index of the last-frame point held;
 * cur_claimed_obs[] in/out: whether that holder has observations.
 * u_right: mvuRight of the current frame.  direction: 0 neither, 1 forward, 2 backward. */
int oracle_search_by_projection_frames(const oracle_frame* cur, const oracle_frame* last, const uint8_t* mp_desc,
                                       const uint8_t* valid, const float* u, const float* v,
                                       const float* invz, const uint8_t* has_obs,
                                       const float* cur_u_right, float bf, float th,
                                       int direction, float nn_ratio, int check_orientation,
                                       int* cur_claimed, uint8_t* cur_claimed_obs, int* query_best);
void oracle_project_last_frame(const float* Tcw, const float* Tlw, const float* x3Dw, int n, float fx, float fy, float cx,
                               float cy, float mb, int bMono, float* u, float* v, float* invz, int* direction);
int oracle_search_by_projection_mappoints(const oracle_frame* f, const float* f_u_right, int nmp,
                                          const uint8_t* valid, const float* proj_x, const float* proj_y,
                                          const float* proj_xr, const int* level, const float* view_cos,
                                          const uint8_t* desc, const uint8_t* has_obs, float th, float nn_ratio,
                                          int* claimed, uint8_t* claimed_obs);
int oracle_search_by_projection_keyframe(const oracle_frame* cur, int n, const uint8_t* valid, const float* u,
                                         const float* v, const int* level, const uint8_t* desc,
                                         const float* kf_angle, float th, int orb_dist, int check_orientation,
                                         int* claimed);
int oracle_search_by_projection_sim3(const oracle_frame* kf, int n, const uint8_t* valid, const float* u,
                                     const float* v, const int* level, const uint8_t* desc, int th, int* matched);

/* DBoW2::FeatureVector as CSR: node ids ascending, items = keypoint indices in insertion order. */
typedef struct oracle_feature_vector {
    int n_nodes;
    const uint32_t* node_ids;
    const int* start; /* n_nodes + 1 */
    const uint32_t* items;
} oracle_feature_vector;

int oracle_search_by_bow(const oracle_frame* f1, const oracle_feature_vector* fv1, const uint8_t* valid1,
                         const oracle_frame* f2, const oracle_feature_vector* fv2, const uint8_t* valid2, int max_dist,
                         float nn_ratio, int check_orientation, int* match12, int* match21);
int oracle_search_for_triangulation(const oracle_frame* f1, const oracle_feature_vector* fv1, const uint8_t* has_mp1,
                                    const float* u_right1, const oracle_frame* f2, const oracle_feature_vector* fv2,
                                    const uint8_t* has_mp2, const float* u_right2, const float* F12, float ex, float ey,
                                    const float* level_sigma2, int only_stereo, int check_orientation, int* match12);
void oracle_window_best(const oracle_frame* kf, int n, const uint8_t* valid, const float* u, const float* v,
                        const float* radius, const int* level, const uint8_t* desc, int chi2_gate,
                        const float* kf_u_right, const float* q_u_right, const float* inv_level_sigma2, int* best_idx,
                        int* best_dist);
/* perfect/src/Frame.cc:357-374 (mask filter) and perfect/src/Map.cc:361-380, :169-187 (map file records). */
int oracle_filter_by_mask(const oracle_keypoint* kps, const uint8_t* desc, int n, const uint8_t* mask, int w, int h,
                          size_t stride, oracle_keypoint* out_kps, uint8_t* out_desc);
void oracle_pack_keyframe_records(const oracle_keypoint* kps, const uint8_t* desc, const uint64_t* mp_index, int n,
                                  uint8_t* records);
void oracle_unpack_keyframe_records(const uint8_t* records, int n, oracle_keypoint* kps, uint8_t* desc, uint64_t* mp_index);
/* MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:279-341), batched. */
void oracle_distinctive_descriptors(const uint8_t* desc, const int* group_start, int ngroups, int* best, int* best_median);
int oracle_search_by_sim3(const oracle_frame* kf1, const oracle_frame* kf2, const uint8_t* valid12, const float* u12,
                          const float* v12, const float* r12, const int* level12, const uint8_t* desc1,
                          const uint8_t* valid21, const float* u21, const float* v21, const float* r21,
                          const int* level21, const uint8_t* desc2, int* match12);

/* Brute-force 2-nearest search with the serial-scan semantics of the
 * reference's descriptor loops (src/ORBmatcher.cc:260-288): strict '<'
 * updates, second-best may equal best, ratio and TH_LOW tests. */
void oracle_bow_transform(int n_nodes, int levels, const uint8_t* node_desc, const int* child_start, const int* children,
                          const int* word_id, const double* weight, const uint8_t* desc, int n, int levelsup, int* word_out,
                          double* weight_out, int* node_out);
int oracle_bow_vector(const int* word, const double* weight, int n, int* ids, double* vals);
void oracle_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* best_idx, int* best_d,
                 int* second_d);
void oracle_knn2_mt(const uint8_t* q, int nq, const uint8_t* t, int nt, int* best_idx, int* best_d,
                    int* second_d, int threads);

#ifdef __cplusplus
}
#endif
#endif
