/* orbx — C ABI of the MI355X ORB front-end (extraction + binary matching).
 *
 * This is the boundary a host program binds instead of the reference's C++
 * classes ORB_SLAM2::ORBextractor and ORB_SLAM2::ORBmatcher.  Plain pointers
 * and sizes only; no exceptions cross it.  Every function that computes runs
 * HIP kernels on one MI355X; there is no CPU fallback: without a usable HIP
 * device orbx_create / the orbm_* calls return ORBX_ERR_NO_DEVICE.
 *
 * Conventions
 *   - return value: 0 (ORBX_OK) or a negative orbx_status;
 *     orbx_last_error() gives a thread-local message for the last failure.
 *   - outputs are caller-allocated with explicit capacities.
 *   - mem_space says where the caller's image / output pointers live.
 *   - one handle = one HIP stream + its scratch memory.  A handle is not
 *     thread-safe; independent handles may be used concurrently (the reference
 *     uses one extractor object per camera, src/Tracking.cc:213-221, and two
 *     objects from two threads for stereo, src/Frame.cc:122-125).
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum orbx_status {
    ORBX_OK = 0,
    ORBX_ERR_INVALID = -1,   /* bad argument */
    ORBX_ERR_NO_DEVICE = -2, /* no HIP device / HIP runtime failure at start-up */
    ORBX_ERR_HIP = -3,       /* a HIP call failed */
    ORBX_ERR_CAPACITY = -4,  /* caller capacity or configured limit too small */
    ORBX_ERR_UNSUPPORTED = -5
} orbx_status;

enum { ORBX_MEM_HOST = 0, ORBX_MEM_DEVICE = 1 };

/* Layout-identical to cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle,
 * response, octave, class_id.  Replaces the std::vector<cv::KeyPoint> argument
 * of ORBextractor::operator() (reference include/ORBextractor.h:46-48). */
typedef struct orbx_keypoint {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} orbx_keypoint;

/* Constructor arguments of ORBextractor (reference include/ORBextractor.h:38-39,
 * read from YAML at src/Tracking.cc:201-209) plus device-side sizing. */
typedef struct orbx_config {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t max_width;    /* largest image width the handle will see */
    int32_t max_height;   /* largest image height */
    int32_t max_batch;    /* frames processed per internal pass (scratch is sized for this) */
    int32_t device;       /* HIP device ordinal */
    /* Which build of the reference is reproduced where its results depend on the OpenCV version or
     * on compiler flags.  blur_variant: 0 = Gaussian taps {18,34,49,55,49,34,18} with an integer
     * column pass (OpenCV 3.4.1 .. 4.4); 1 = taps {18,34,48,56,48,34,18} (OpenCV >= 4.5); 2 = the taps
     * of 0 with the single-precision column pass of OpenCV <= 3.4.0, the versions the reference's
     * README names (ties round to even; the last width % 4 columns as in 0).  fma_variant: 0 = the
     * descriptor's steering sums x*b + y*a, x*a - y*b with every product rounded (a build without
     * fused multiply-add); 1 = the first product fused into the sum (GCC with -march=native on a
     * host that has FMA, perfect/CMakeLists.txt:20). */
    int32_t blur_variant;
    int32_t fma_variant;
} orbx_config;

typedef struct orbx_handle orbx_handle;

const char* orbx_last_error(void);
/* "orbx <major>.<minor> (gfx950)".  0.2: orbx_set_stream / orbm_set_stream take NULL as the legacy default
 * stream (0.1 took it as "back to the handle's own stream"; that is orbx_reset_stream / orbm_reset_stream now). */
const char* orbx_version(void);

/* ORBextractor::ORBextractor (reference src/ORBextractor.cc:399-466). */
int orbx_create(const orbx_config* cfg, orbx_handle** out);
void orbx_destroy(orbx_handle* h);

/* Run the handle's kernels on an existing hipStream_t.  Every value selects that stream: NULL is
 * the legacy default stream, like any other handle.  orbx_reset_stream returns to the handle's own
 * (non-blocking) stream.  Work the caller enqueues on the selected stream before and after a
 * device-space call is ordered with the call's kernels. */
int orbx_set_stream(orbx_handle* h, void* hip_stream);
int orbx_reset_stream(orbx_handle* h);
/* Waits for the handle's stream.  Errors that kernels of device-space calls detected since the last
 * wait surface here: ORBX_ERR_CAPACITY when cap_per_frame was smaller than a frame's keypoint count
 * (that frame's output was truncated to cap_per_frame entries) or a quadtree node table overflowed.
 * The error state is cleared by the call that reports it. */
int orbx_synchronize(orbx_handle* h);
/* The blur of a pass depends only on the pyramid; by default it runs on a second
 * stream beside FAST + quadtree and joins before the descriptor kernel.  on = 0
 * serialises all kernels on the handle's stream (clean per-kernel timings). */
int orbx_set_overlap(orbx_handle* h, int on);

/* Upper bound on keypoints per frame (>= nfeatures; the quadtree may return a
 * few more than the quota per level, reference src/ORBextractor.cc:671-733). */
int orbx_max_keypoints(const orbx_handle* h);

/* ORBextractor::operator()(image, mask, keypoints, descriptors)
 * (reference src/ORBextractor.cc:1052-1114).  The mask is ignored by the
 * reference (include/ORBextractor.h:45) and has no parameter here.
 * img: 8-bit single channel, `stride` bytes per row, host memory.
 * kps[cap], desc[cap*32]: host memory.  An empty image (w or h == 0) returns
 * ORBX_OK with *n_out = 0, as the reference returns silently. */
int orbx_extract(orbx_handle* h, const uint8_t* img, int w, int h_, size_t stride,
                 orbx_keypoint* kps, uint8_t* desc, int cap, int* n_out);

/* Batched form: B frames of equal size.  Frame f starts at imgs + f*img_stride.
 * Outputs for frame f start at kps + f*cap_per_frame and desc + f*cap_per_frame*32;
 * n_out[f] receives its keypoint count; what entries n_out[f] .. cap_per_frame - 1 of a frame hold
 * afterwards is unspecified.  mem_space applies to imgs, kps, desc and n_out alike.  Host-space calls
 * return when every output has arrived: the frames go through the GPU in passes whose uploads, kernels
 * and downloads overlap (rows of any stride; buffers that are already page-locked are copied from and
 * to directly).  With ORBX_MEM_DEVICE the call only enqueues work on the
 * handle's stream; capacity errors found by its kernels are reported by the next
 * orbx_synchronize (or the next host-space call). */
int orbx_extract_batch(orbx_handle* h, const uint8_t* imgs, int B, int w, int h_,
                       size_t row_stride, size_t img_stride, orbx_keypoint* kps, uint8_t* desc,
                       int cap_per_frame, int* n_out, int mem_space);

/* Page-locks a host buffer the caller keeps using across calls (image batches, output arrays), so that
 * host-space calls copy from / to it directly instead of through the handle's own pinned staging; a
 * wrapper over hipHostRegister / hipHostUnregister for callers that do not link the HIP runtime.  It
 * has no counterpart in the reference (whose images live in cv::Mat buffers, rgbd_tum.cc:77-119). */
int orbx_register_host(void* ptr, size_t bytes);
int orbx_unregister_host(void* ptr);

/* mvImagePyramid[level] of frame `frame` of the last extract call (public
 * member read by Frame::ComputeStereoMatches, reference src/Frame.cc:610,700).
 * *ptr is DEVICE memory owned by the handle (level 0 aliases the caller's
 * image when it was passed as device memory with base, row stride and image stride multiples
 * of 4; any other device layout is copied once to aligned rows on the device and level 0 is
 * that copy), valid until the next call.  A batch longer
 * than max_batch is processed in passes; only the frames of the last pass stay resident. */
int orbx_get_level(orbx_handle* h, int frame, int level, const uint8_t** ptr, int* w, int* h_,
                   size_t* stride);

/* Getters (reference include/ORBextractor.h:51-71).  Arrays have nlevels entries. */
int orbx_levels(const orbx_handle* h);
float orbx_scale_factor(const orbx_handle* h);
const float* orbx_scale_factors(const orbx_handle* h);
const float* orbx_inv_scale_factors(const orbx_handle* h);
const float* orbx_level_sigma2(const orbx_handle* h);
const float* orbx_inv_level_sigma2(const orbx_handle* h);
const int32_t* orbx_features_per_level(const orbx_handle* h);

/* ---- stage timing (HIP events on the handle's stream) ---- */
enum {
    ORBX_STAGE_PYRAMID = 0,
    ORBX_STAGE_FAST = 1,
    ORBX_STAGE_QUADTREE = 2,
    ORBX_STAGE_BLUR = 3,
    ORBX_STAGE_DESCRIBE = 4,
    ORBX_STAGE_COUNT = 5
};
int orbx_profile_enable(orbx_handle* h, int on);
/* Synchronises, then returns accumulated milliseconds and launch counts per
 * stage since the last reset. */
int orbx_profile_read(orbx_handle* h, float* ms, int* launches, int n);
int orbx_profile_reset(orbx_handle* h);

/* ---- debug taps used by the parity tests (stage outputs of the last pass) ---- */
/* Device evaluation of the two float primitives of the descriptor path: the steering
 * sine/cosine (radians) and fastAtan2 (degrees), n values each, host arrays. */
int orbx_debug_sincos(orbx_handle* h, const float* angles, int n, float* sin_out, float* cos_out);
int orbx_debug_atan2(orbx_handle* h, const float* y, const float* x, int n, float* out);
/* The constant tables of the descriptor kernel as they sit in device memory, 752 words: 248
 * weight words and 248 membership words of the radius-15 orientation patch (31 rows x 8 words
 * covering u = -16..15; weight byte = u + 15), then the 256 rBRIEF tests as (x0, y0, x1, y1)
 * signed bytes (reference src/ORBextractor.cc:135-393, :450-465). */
int orbx_debug_tables(orbx_handle* h, uint32_t* out, int n);
/* The taps below number frames as orbx_get_level does: `frame` is the index in the batch of the
 * last call, and only the frames of its last pass are resident (ORBX_ERR_INVALID otherwise).
 * Pyramid level image of frame `frame`: copies w*h bytes, tightly packed, to host dst. */
int orbx_debug_level(orbx_handle* h, int frame, int level, uint8_t* dst);
/* Blurred level image of frame `frame`: copies w*h bytes, tightly packed, to host dst. */
int orbx_debug_blurred(orbx_handle* h, int frame, int level, uint8_t* dst, int* w, int* h_);
/* FAST candidates of (frame, level) in the reference's append order; x,y relative
 * to the detection origin (16,16).  Returns count via *n (may exceed cap). */
int orbx_debug_candidates(orbx_handle* h, int frame, int level, int* xs, int* ys, int* resp,
                          int cap, int* n);
/* Quadtree selection of (frame, level): level-image pixel coordinates. */
int orbx_debug_selected(orbx_handle* h, int frame, int level, int* xs, int* ys, int* resp, int cap,
                        int* n);

/* =====================  matching  ===================== */

/* ORBmatcher::DescriptorDistance (reference src/ORBmatcher.cc:1905-1921). */
int orbm_hamming256(const uint8_t* a, const uint8_t* b);

#define ORBM_GRID_COLS 64 /* FRAME_GRID_COLS, reference include/Frame.h:26 */
#define ORBM_GRID_ROWS 48 /* FRAME_GRID_ROWS, reference include/Frame.h:25 */
#define ORBM_MAX_LEVELS 16

typedef struct orbm_context orbm_context;

/* A matcher context owns a stream and scratch on one device. */
int orbm_create(int device, orbm_context** out);
void orbm_destroy(orbm_context* m);
/* Stream selection as for orbx_set_stream / orbx_reset_stream (NULL = legacy default stream). */
int orbm_set_stream(orbm_context* m, void* hip_stream);
int orbm_reset_stream(orbm_context* m);
int orbm_synchronize(orbm_context* m);

/* One frame's matching inputs, all arrays in the memory space given to the call. */
typedef struct orbm_frame {
    const orbx_keypoint* kps; /* mvKeysUn */
    const uint8_t* desc;      /* n x 32 */
    int32_t n;
    float min_x, min_y, max_x, max_y; /* mnMinX, mnMinY, mnMaxX, mnMaxY */
} orbm_frame;

/* Frame::AssignFeaturesToGrid + PosInGrid (reference src/Frame.cc:302-317,
 * 494-502).  CSR over 64x48 cells, cell id = ix*48 + iy; cell_start has
 * 64*48+1 entries, cell_items up to n entries in ascending keypoint index. */
int orbm_grid_build(orbm_context* m, const orbm_frame* f, int32_t* cell_start, int32_t* cell_items,
                    int mem_space);

/* Frame::GetFeaturesInArea for many queries (reference src/Frame.cc:437-490):
 * query q = (x, y, r, min_level, max_level).  Emits, per query, the candidate
 * indices in the reference's order (ix outer, iy inner, bucket order) and
 * their Hamming distance to qdesc[q] when qdesc != NULL.  Fixed row capacity
 * max_cand; counts[q] holds the true count (may exceed max_cand). */
int orbm_window_candidates(orbm_context* m, const orbm_frame* train, const float* qx,
                           const float* qy, const float* qr, const int32_t* qmin_level,
                           const int32_t* qmax_level, const uint8_t* qdesc, int nq, int max_cand,
                           int32_t* cand_idx, int32_t* cand_dist, int32_t* counts, int mem_space);

/* ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:501-628),
 * including its sequential best-distance bookkeeping and match stealing.
 * prev_matched_xy: n1 x 2 floats in/out (vbPrevMatched); matches12: n1 ints out.
 * *nmatches receives the return value of the reference function. */
int orbm_search_for_initialization(orbm_context* m, const orbm_frame* f1, const orbm_frame* f2,
                                   float* prev_matched_xy, int32_t* matches12, int window_size,
                                   float nn_ratio, int check_orientation, int* nmatches,
                                   int mem_space);

/* ---- projected searches ----
 * Every SearchByProjection variant of the reference has the same shape: the host projects
 * map points into a frame (pose algebra stays in host C++), then per projected point the
 * keypoints inside a window and level range are scanned in GetFeaturesInArea order for the
 * closest descriptor, skipping keypoints that already hold a map point, and the winner is
 * claimed immediately (so later queries see it).  orbm_search_projected is that shape;
 * the variants differ only in how queries and policy are filled:
 *
 *   variant (reference src/ORBmatcher.cc)        radius                 levels            right gate  policy {claim, max_dist, ratio, ori}
 *   (Frame&, vector<MapPoint*>&, th)   :62-155   r(viewCos)*th*s[lvl]   [lvl-1, lvl]      projXR      {0, TH_HIGH, 1, 0}
 *   (Frame&, const Frame&, th, bMono)  :1533-1678 th*s[octave]          by motion         u-bf*invz   {0, TH_HIGH, 0, checkOri}
 *   (Frame&, KeyFrame*, sFound, th, d) :1694-1836 th*s[lvl]             [lvl-1, lvl+1]    none        {1, d,       0, checkOri}
 *   (KeyFrame*, Scw, pts, matched, th) :370-485   th*s[lvl]             [lvl-1, lvl]      none        {1, TH_LOW,  0, 0}
 */
typedef struct orbm_proj_queries {
    int32_t n;
    const uint8_t* valid;     /* n: 0 = skip this map point (bad, out of view, ...) */
    const float* u;           /* n: projection */
    const float* v;
    const float* radius;      /* n: window half-size in pixels */
    const int32_t* min_level; /* n: GetFeaturesInArea minLevel / maxLevel (-1 = open) */
    const int32_t* max_level;
    const uint8_t* desc;      /* n x 32: pMP->GetDescriptor() */
    const float* angle;       /* n: keypoint angle for the rotation histogram (NULL if unused) */
    const uint8_t* has_obs;   /* n: pMP->Observations() > 0 (NULL = all 1) */
    const float* u_right;     /* n: projected right-image coordinate, or NULL for no stereo gate */
    const float* right_gate;  /* n: candidates with mvuRight > 0 and |u_right - mvuRight| > gate are skipped */
} orbm_proj_queries;

typedef struct orbm_proj_policy {
    int32_t claim_rule;        /* 0: skip keypoints whose map point has observations; 1: skip any keypoint holding a point */
    int32_t max_dist;          /* accept best distance <= max_dist */
    int32_t ratio_test;        /* 1: reject when best and second best share a level and best > nn_ratio * second */
    float nn_ratio;
    int32_t check_orientation; /* rotation histogram + ComputeThreeMaxima cull */
} orbm_proj_policy;

/* claimed (in/out, train->n): index of the query assigned to each train keypoint or -1
 * (mvpMapPoints / vpMatched); claimed_obs (in/out): whether that holder has observations.
 * query_best (out, n, may be NULL): train keypoint each query was assigned when it was
 * processed, before the orientation cull, or -1.  *nmatches: the reference's return value. */
int orbm_search_projected(orbm_context* m, const orbm_frame* train, const float* train_u_right,
                          const orbm_proj_queries* q, const orbm_proj_policy* p, int32_t* claimed,
                          uint8_t* claimed_obs, int32_t* query_best, int* nmatches, int mem_space);

/* ORBmatcher::SearchByProjection(Frame& cur, const Frame& last, th, bMono)
 * (reference src/ORBmatcher.cc:1533-1678; perfect/src/ORBmatcher.cc:1726 adds the lists of
 * matched points, which are query_best in query order) from the point where each
 * last-frame map point has been projected into the current frame (:1577-1578): per
 * last-frame keypoint i the caller passes valid[i] (map point present and not an
 * outlier, :1564-1566), u[i], v[i], invz[i], has_obs[i] and mp_desc (pMP->GetDescriptor(),
 * n_last x 32; NULL = use last->desc).  direction: 0 neither, 1 forward, 2 backward
 * (:1554-1556).  The checks of :1575-1582, the radius (:1591), the level window
 * (:1593-1598) and the right-image gate (:1615-1621) are applied inside. */
int orbm_search_by_projection(orbm_context* m, const orbm_frame* cur, const orbm_frame* last,
                              const uint8_t* mp_desc, const uint8_t* valid, const float* u, const float* v,
                              const float* invz, const uint8_t* has_obs, const float* cur_u_right,
                              const float* scale_factors, int nlevels, float bf, float th, int direction,
                              int check_orientation, int32_t* cur_claimed, uint8_t* cur_claimed_obs,
                              int32_t* query_best, int* nmatches, int mem_space);

/* ---- vocabulary-guided searches ----
 * DBoW2::FeatureVector (a std::map<NodeId, std::vector<unsigned int>>, reference
 * Thirdparty/DBoW2/DBoW2/FeatureVector.h) flattened to CSR: node_ids strictly ascending (map
 * order), items[start[i] .. start[i+1]) = the keypoint indices of node i in vector order.
 * Every keypoint occurs in at most one node (DBoW2 files a feature under exactly one node); the
 * kernels process nodes concurrently and rely on it.  Host-memory vectors are validated
 * (ORBX_ERR_INVALID); with ORBX_MEM_DEVICE nothing is read back, so ascending ids, items < n and
 * disjoint nodes are preconditions the caller guarantees. */
typedef struct orbm_feature_vector {
    int32_t n_nodes;
    int32_t n_items;
    const uint32_t* node_ids; /* n_nodes */
    const int32_t* start;     /* n_nodes + 1 */
    const uint32_t* items;    /* n_items */
} orbm_feature_vector;

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, matches) (reference src/ORBmatcher.cc:209-355)
 * and SearchByBoW(KeyFrame*, KeyFrame*, matches12) (:642-788).  f1/fv1 = the (first)
 * keyframe, f2/fv2 = the frame or second keyframe.  valid1[i] = keypoint i of f1 holds a map
 * point that is not bad (:248-251, :684-688); valid2 (NULL = all) = the same for f2 (:700-704).
 * max_dist is inclusive: TH_LOW (50) for the first overload (:284), TH_LOW - 1 for the second
 * (:721).  match12[i1] = matched keypoint of f2 or -1 (vpMatches12 by index); match21[i2] =
 * keypoint of f1 whose map point keypoint i2 of f2 received, or -1 (vpMapPointMatches by
 * index).  *nmatches = the reference's return value. */
int orbm_search_by_bow(orbm_context* m, const orbm_frame* f1, const orbm_feature_vector* fv1,
                       const uint8_t* valid1, const orbm_frame* f2, const orbm_feature_vector* fv2,
                       const uint8_t* valid2, int max_dist, float nn_ratio, int check_orientation,
                       int32_t* match12, int32_t* match21, int* nmatches, int mem_space);

/* ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:803-992) from the point
 * where the epipole (ex, ey) of keyframe 1 in keyframe 2 (:809-818) and the fundamental
 * matrix are known.  has_mp*[i] = keypoint i already holds a map point (:858, :882);
 * u_right* = mvuRight; f12: 3x3 row-major, host memory; scale_factors2 / level_sigma2_2:
 * mvScaleFactors / mvLevelSigma2 of keyframe 2, host memory.  match12 = vMatches12; the pair
 * list of the reference is its non-negative entries in index order (:984-989). */
int orbm_search_for_triangulation(orbm_context* m, const orbm_frame* f1, const orbm_feature_vector* fv1,
                                  const uint8_t* has_mp1, const float* u_right1, const orbm_frame* f2,
                                  const orbm_feature_vector* fv2, const uint8_t* has_mp2,
                                  const float* u_right2, const float* f12, float ex, float ey,
                                  const float* scale_factors2, const float* level_sigma2_2, int nlevels,
                                  int only_stereo, int check_orientation, int32_t* match12,
                                  int* nmatches, int mem_space);

/* The candidate loop shared by ORBmatcher::Fuse (reference src/ORBmatcher.cc:1066-1120), its
 * Sim3 form (:1233-1262) and both directions of SearchBySim3 (:1391-1416, :1462-1489): query i
 * is a map point already projected to (u, v) with search radius and predicted level (pose
 * algebra and the culls before the window stay with the caller: valid[i]).  Keypoints of
 * level-1 or level inside the window compete; the closest descriptor wins, the first in
 * GetFeaturesInArea order on ties.  chi2_gate = 1 applies Fuse's reprojection test first
 * (:1085-1111): kf_u_right = mvuRight (NULL = monocular), q_u_right[i] = u - bf*invz,
 * inv_level_sigma2 = mvInvLevelSigma2 (host memory, nlevels entries).  best_dist[i] = 256 and
 * best_idx[i] = -1 when nothing qualified; the caller applies TH_LOW / TH_HIGH and the map
 * point bookkeeping (:1123-1148). */
int orbm_window_best(orbm_context* m, const orbm_frame* kf, int n, const uint8_t* valid, const float* u,
                     const float* v, const float* radius, const int32_t* level, const uint8_t* desc,
                     int chi2_gate, const float* kf_u_right, const float* q_u_right,
                     const float* inv_level_sigma2, int nlevels, int32_t* best_idx, int32_t* best_dist,
                     int mem_space);

/* ORBmatcher::SearchBySim3 (reference src/ORBmatcher.cc:1298-1517) from the projections on:
 * *12 arrays have kf1->n entries (map point of keypoint i of keyframe 1 projected into
 * keyframe 2; valid12 folds :1355-1385 including vbAlreadyMatched1), *21 arrays kf2->n entries
 * for the reverse direction.  match12[i1] = keypoint of keyframe 2 when both directions agree
 * (:1499-1514), else -1; *nfound = the return value. */
int orbm_search_by_sim3(orbm_context* m, const orbm_frame* kf1, const orbm_frame* kf2,
                        const uint8_t* valid12, const float* u12, const float* v12, const float* r12,
                        const int32_t* level12, const uint8_t* mp_desc1, const uint8_t* valid21,
                        const float* u21, const float* v21, const float* r21, const int32_t* level21,
                        const uint8_t* mp_desc2, int32_t* match12, int* nfound, int mem_space);

/* Dynamic-object keypoint filter of the RGB-D frame constructor (reference
 * perfect/src/Frame.cc:357-374): when the sum of the mask bytes exceeds 0.65 * w * h, only the
 * keypoints whose mask byte at (int(pt.y), int(pt.x)) equals 1 are kept, in order; otherwise
 * all are kept.  mask: 8-bit image in the call's memory space.  out arrays have room for n
 * keypoints; *n_out (host memory) = the number kept.  Keypoints outside the mask image are
 * dropped (the reference reads out of bounds there). */
int orbm_filter_by_mask(orbm_context* m, const orbx_keypoint* kps, const uint8_t* desc, int n,
                        const uint8_t* mask, int w, int h, size_t stride, orbx_keypoint* out_kps,
                        uint8_t* out_desc, int* n_out, int mem_space);

/* Keypoint records of the binary map file (reference perfect/src/Map.cc:361-380 write,
 * :169-187 read): per keypoint pt.x, pt.y, size, angle, response (float), octave (int), the 32
 * descriptor bytes, the map point index (unsigned long; ULONG_MAX = none): 64 bytes, in
 * keypoint order.  mp_index may be NULL (pack: every index ULONG_MAX; unpack: not returned).
 * Unpacked keypoints get class_id = -1, cv::KeyPoint's default. */
int orbm_pack_keyframe_records(orbm_context* m, const orbx_keypoint* kps, const uint8_t* desc,
                               const uint64_t* mp_index, int n, uint8_t* records, int mem_space);
int orbm_unpack_keyframe_records(orbm_context* m, const uint8_t* records, int n, orbx_keypoint* kps,
                                 uint8_t* desc, uint64_t* mp_index, int mem_space);

/* MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:279-341) for a batch of
 * map points: group g owns descriptors group_start[g] .. group_start[g+1] of desc (the rows
 * vDescriptors collected at :296-302; total_desc rows in all).  best[g] = BestIdx inside the
 * group (-1 for an empty group, which the reference leaves untouched, :303-304);
 * best_median[g] (may be NULL) = BestMedian; out_desc (may be NULL, ngroups x 32) receives
 * vDescriptors[BestIdx].  A group may hold at most 4096 descriptors. */
int orbm_distinctive_descriptors(orbm_context* m, const uint8_t* desc, const int32_t* group_start,
                                 int ngroups, int32_t* best, int32_t* best_median, uint8_t* out_desc,
                                 int total_desc, int mem_space);

/* Frame::ComputeStereoMatches (reference src/Frame.cc:603-774) for a rectified pair.
 * left/right: mvKeys / mDescriptors and mvKeysRight / mDescriptorsRight (the grid bounds of
 * orbm_frame are not read).  pyr_left / pyr_right: mvImagePyramid of the two extractors
 * (reference include/ORBextractor.h:73) as DEVICE images, e.g. from orbx_get_level; the
 * extractor work that produced them must have completed (orbx_extract returns after it has;
 * after a device-space orbx_extract_batch call orbx_synchronize first).  scale_factors /
 * inv_scale_factors: mvScaleFactors / mvInvScaleFactors, nlevels entries each.  mb, mbf:
 * baseline in metres and baseline times fx.  u_right / depth (left->n floats): mvuRight and
 * mvDepth, -1 where no stereo point was found.  *n_stereo (may be NULL; host memory): number of
 * stereo points kept after the median cut (:763-774); asking for it makes the call synchronous.
 * Where the reference would index outside its row table or cut an image window that leaves the
 * pyramid level (cv::Mat range assertion), the keypoint gets no match instead. */
typedef struct orbm_pyramid {
    int32_t nlevels;
    const uint8_t* level[ORBM_MAX_LEVELS];
    int32_t width[ORBM_MAX_LEVELS], height[ORBM_MAX_LEVELS];
    size_t pitch[ORBM_MAX_LEVELS];
} orbm_pyramid;
int orbm_compute_stereo_matches(orbm_context* m, const orbm_frame* left, const orbm_frame* right,
                                const orbm_pyramid* pyr_left, const orbm_pyramid* pyr_right,
                                const float* scale_factors, const float* inv_scale_factors, float mb,
                                float mbf, float* u_right, float* depth, int* n_stereo, int mem_space);

/* Frame::ComputeStereoMatches (reference src/Frame.cc:603-774) for a table of independent rectified
 * pairs in one set of launches: the stereo step of a batch of frames whose two images were extracted
 * with orbx_extract_batch (the Frame constructor runs it once per stereo frame, src/Frame.cc:118-152).
 * Every array a job points to, the pyramid levels included, is DEVICE memory (the table itself is host
 * memory); arguments as in orbm_compute_stereo_matches; all jobs share the scale tables, mb and mbf and
 * have pyramids of `nlevels` levels.  n_stereo is written on return (the call is synchronous). */
typedef struct orbm_stereo_job {
    orbm_frame left, right;
    orbm_pyramid pyr_left, pyr_right;
    float* u_right; /* left.n, out */
    float* depth;   /* left.n, out */
    int32_t n_stereo; /* out */
} orbm_stereo_job;
int orbm_compute_stereo_matches_batch(orbm_context* m, orbm_stereo_job* jobs, int njobs, const float* scale_factors,
                                      const float* inv_scale_factors, int nlevels, float mb, float mbf);

/* Many independent frame pairs in one call: orbm_search_by_projection for every entry of
 * `pairs`, each with the meaning of the arguments of the same name above.  Every array a
 * pair points to is DEVICE memory (the pair table itself is host memory); cur_claimed /
 * cur_claimed_obs are in/out per pair, query_best may be NULL, nmatches is written on return.
 * scale_factors (host memory), bf, th and check_orientation are shared by all pairs. */
typedef struct orbm_projection_pair {
    orbm_frame cur, last;
    const uint8_t* mp_desc; /* last.n x 32, or NULL = last.desc */
    const uint8_t* valid;   /* last.n */
    const float *u, *v, *invz;
    const uint8_t* has_obs;
    const float* cur_u_right; /* cur.n */
    int32_t direction;
    int32_t* cur_claimed;     /* cur.n, in/out */
    uint8_t* cur_claimed_obs; /* cur.n, in/out */
    int32_t* query_best;      /* last.n or NULL */
    int32_t nmatches;         /* out */
} orbm_projection_pair;
int orbm_search_by_projection_batch(orbm_context* m, orbm_projection_pair* pairs, int npairs,
                                    const float* scale_factors, int nlevels, float bf, float th,
                                    int check_orientation);

/* Many independent SearchByBoW calls in one set of launches (relocalisation and loop detection match
 * one frame or keyframe against many candidate keyframes, reference src/Tracking.cc:1609-1625,
 * src/LoopClosing.cc:239-258): orbm_search_by_bow for every entry of `jobs` with shared max_dist,
 * nn_ratio and check_orientation.  Every array a job points to is DEVICE memory (the job table itself is
 * host memory) and meets the preconditions stated for device feature vectors above; valid2 may be
 * NULL (the (KeyFrame*, Frame&) overload); match12 (f1.n) and match21 (f2.n) are outputs, nmatches is
 * written on return. */
typedef struct orbm_bow_job {
    orbm_frame f1, f2;
    orbm_feature_vector fv1, fv2;
    const uint8_t* valid1; /* f1.n */
    const uint8_t* valid2; /* f2.n or NULL */
    int32_t* match12;      /* f1.n, out */
    int32_t* match21;      /* f2.n, out */
    int32_t nmatches;      /* out */
} orbm_bow_job;
int orbm_search_by_bow_batch(orbm_context* m, orbm_bow_job* jobs, int njobs, int max_dist, float nn_ratio,
                             int check_orientation);

/* Many independent SearchForInitialization calls in one set of launches (reference
 * src/ORBmatcher.cc:497-617, one call per initialisation attempt at src/Tracking.cc:744; many sequences
 * or many candidate second frames at once here): orbm_search_for_initialization for every entry of
 * `jobs` with shared window_size, nn_ratio and check_orientation.  Every array a job points to is
 * DEVICE memory (the job table itself is host memory); prev_matched_xy (2 * f1.n floats) is in/out,
 * matches12 (f1.n) is an output, nmatches is written on return. */
typedef struct orbm_init_job {
    orbm_frame f1, f2;
    float* prev_matched_xy; /* 2 * f1.n, in/out */
    int32_t* matches12;     /* f1.n, out */
    int32_t nmatches;       /* out */
} orbm_init_job;
int orbm_search_for_initialization_batch(orbm_context* m, orbm_init_job* jobs, int njobs, int window_size,
                                         float nn_ratio, int check_orientation);

/* Many independent projected searches in one call: orbm_search_projected for every entry of `jobs`
 * under one policy, i.e. the batched form of the three SearchByProjection overloads that project map
 * points into a frame or keyframe (reference src/ORBmatcher.cc:62-164, :370-485, :1694-1836; the
 * frame-to-frame overload has orbm_search_by_projection_batch).  Every array a job points to is
 * DEVICE memory (the job table itself is host memory); claimed / claimed_obs are in/out per job,
 * query_best may be NULL, nmatches is written on return.  window_hint: an upper bound of the window
 * half-sizes (q.radius) if the caller knows one, 0 otherwise; it only selects how many queries share a
 * wave in the window walk (<= 8 and <= 15 pixels have forms of their own) and never changes a result. */
typedef struct orbm_projected_job {
    orbm_frame train;
    const float* train_u_right; /* train.n, or NULL */
    orbm_proj_queries q;
    int32_t* claimed;           /* train.n, in/out */
    uint8_t* claimed_obs;       /* train.n, in/out */
    int32_t* query_best;        /* q.n or NULL */
    int32_t nmatches;           /* out */
} orbm_projected_job;
int orbm_search_projected_batch(orbm_context* m, orbm_projected_job* jobs, int njobs,
                                const orbm_proj_policy* p, float window_hint);

/* Brute-force 2-nearest-neighbour search over 256-bit descriptors with the
 * serial-scan semantics of the reference's descriptor loops
 * (src/ORBmatcher.cc:255-282): best = smallest distance, lowest train index on
 * ties; second = second smallest counting duplicates of the minimum; both start
 * at 256.  train_index_base is added to best_idx (for sharded train sets). */
int orbm_knn2(orbm_context* m, const uint8_t* q, int nq, const uint8_t* t, int nt,
              int train_index_base, int32_t* best_idx, int32_t* best_d, int32_t* second_d,
              int mem_space);

/* Debug tap: which form of the scan orbm_knn2 runs.  0 = chosen by problem size (the default),
 * 1 = vector ALU (xor + popcount), 2 = matrix cores (block-scaled fp4 / e2m1 contraction,
 * v_mfma_scale_f32_32x32x64_f8f6f4).  Both give the same results; the tap exists so that tests and
 * measurements can run either on any size. */
int orbm_debug_knn_path(orbm_context* m, int path);

/* ---- bag-of-words transform ----
 * DBoW2's vocabulary tree as plain arrays (DBoW2 itself and its vocabulary file are dependencies the
 * reference does not vendor, Thirdparty/DBoW2/readme.md): node 0 is the root; children of node i
 * are children[child_start[i] .. child_start[i+1]) in the order of m_nodes[i].children; a node
 * without children is a word with word_id and weight (TemplatedVocabulary::Node); levels = m_L. */
typedef struct orbm_vocabulary {
    int32_t n_nodes;
    int32_t n_children;          /* entries of `children` (= n_nodes - 1 for a tree) */
    int32_t levels;              /* m_L */
    const uint8_t* node_desc;    /* n_nodes x 32 (the root's row is not read) */
    const int32_t* child_start;  /* n_nodes + 1 */
    const int32_t* children;     /* n_children */
    const int32_t* word_id;      /* n_nodes; read for nodes without children */
    const double* weight;        /* n_nodes; read for nodes without children */
} orbm_vocabulary;

/* TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(feature, word_id, weight, &nid, levelsup)
 * for every row of desc: the call Frame::ComputeBoW makes per descriptor (reference
 * src/Frame.cc:513-522, levelsup = 4).  From the root the feature goes to the closest child in
 * Hamming distance, the first child on ties, down to a node without children; node_id[i] is the
 * node passed at level levels - levelsup (0, the root, when that is <= 0).  The BowVector
 * (sum of weights per word, L1-normalised) and the FeatureVector (features per node, ascending) are
 * std::map containers the host fills from these three arrays. */
int orbm_bow_transform(orbm_context* m, const orbm_vocabulary* voc, const uint8_t* desc, int n,
                       int levelsup, int32_t* word_id, double* weight, int32_t* node_id, int mem_space);

/* The acceptance rule that follows the best / second-best scan in the reference's descriptor
 * loops (src/ORBmatcher.cc:284-288, :561-565): accept[i] = best_d[i] <= max_dist (TH_LOW or
 * TH_HIGH at the call site) && (float)best_d[i] < nn_ratio * (float)second_d[i]. */
int orbm_ratio_test(orbm_context* m, int nq, const int32_t* best_d, const int32_t* second_d,
                    float nn_ratio, int max_dist, uint8_t* accept, int mem_space);

/* orbm_knn2 for callers that cannot allocate: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  It has no counterpart in the reference.  Same results as
 * orbm_knn2 with ORBX_MEM_DEVICE (q, t and the three outputs are device arrays, q and t 16-byte
 * aligned); the call only enqueues kernels on the context's stream and calls no hipMalloc,
 * hipFree or other allocating routine.  Its partial results live in a 12 MiB array that is part
 * of the library's code object on every GPU it runs on.  Limits: nq <= 1048576 and nt <= 4194304
 * (ORBX_ERR_CAPACITY beyond).  All calls on one GPU share that array.  Calls made outside graph
 * capture are ordered by the library: each waits, on the device, for the previous call on that GPU
 * (an event made in orbm_create), so two contexts on two streams or two host threads do not race.
 * Calls made under graph capture cannot take part in that ordering: graphs that contain this entry
 * point must not run at the same time as another use of it on the same GPU. */
int orbm_knn2_noalloc(orbm_context* m, const uint8_t* q, int nq, const uint8_t* t, int nt,
                      int train_index_base, int32_t* best_idx, int32_t* best_d, int32_t* second_d);

/* Merge two partial kNN-2 results for the same queries (shard A precedes shard
 * B in global train order): reproduces the serial scan over A then B. */
int orbm_knn2_merge(orbm_context* m, int nq, int32_t* best_idx_a, int32_t* best_d_a,
                    int32_t* second_d_a, const int32_t* best_idx_b, const int32_t* best_d_b,
                    const int32_t* second_d_b, int mem_space);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
