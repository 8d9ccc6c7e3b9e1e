// 256-bit binary descriptor matching on MI355X (gfx950): keypoint grid,
// windowed candidate search, the sequential-state matchers of the reference
// (SearchForInitialization, SearchByProjection frame-to-frame) and a
// brute-force 2-nearest-neighbour search.
//
// Replaces the grid functions of ORB_SLAM2::Frame (reference src/Frame.cc:
// 302-317, 437-502) and ORB_SLAM2::ORBmatcher (reference src/ORBmatcher.cc).
// Distances are XOR + popcount on 32/64-bit words (v_bcnt); nothing here is a
// dense contraction, so the matrix cores are not used.

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "orbx_internal.h"

namespace {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef unsigned long long u64;
typedef long long i64;

const int GRID_COLS = ORBM_GRID_COLS, GRID_ROWS = ORBM_GRID_ROWS, GRID_CELLS = GRID_COLS * GRID_ROWS;
const int TH_HIGH = 100;      // reference src/ORBmatcher.cc:38
const int TH_LOW = 50;        // :39
const int HISTO_LENGTH = 30;  // :40

struct DevFrame {
    const orbx_keypoint* kps;
    const u8* desc;
    int n;
    float minX, minY, maxX, maxY;
    float invW, invH;  // mfGridElementWidthInv / HeightInv (reference src/Frame.cc:211-212)
};

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__device__ __forceinline__ int wave_excl_count(bool flag, int* total) {
    const u64 m = __ballot(flag);
    *total = __popcll(m);
    return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0));
}

// ------------------------------------------------------------------ grid
// Frame::PosInGrid + AssignFeaturesToGrid: one workgroup per frame.  Cell id
// = ix*48 + iy so that a window's cells of one column are contiguous in the
// CSR; items inside a cell keep ascending keypoint order.
__device__ __forceinline__ void grid_build_body(DevFrame f, int* cellStart, int* cellItems, u16* cellOf) {
    __shared__ int hist[GRID_CELLS + 1];
    __shared__ int wsum[16];
    const int tid = threadIdx.x;
    for (int i = tid; i <= GRID_CELLS; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < f.n; i += 1024) {
        const orbx_keypoint kp = f.kps[i];
        const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, f.minX), f.invW));
        const int py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, f.minY), f.invH));
        u16 c = 0xFFFF;
        if (px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS) {
            c = (u16)(px * GRID_ROWS + py);
            atomicAdd(&hist[c], 1);
        }
        cellOf[i] = c;
    }
    __syncthreads();
    // exclusive scan of 3072 counts: 3 per thread
    int v[3], s = 0;
    for (int k = 0; k < 3; k++) {
        v[k] = hist[tid * 3 + k];
        s += v[k];
    }
    int inc = s;
    const int lane = tid & 63, wv = tid >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wv; k++) base += wsum[k];
    int run = base + inc - s;
    __syncthreads();
    for (int k = 0; k < 3; k++) {
        hist[tid * 3 + k] = run;
        cellStart[tid * 3 + k] = run;
        run += v[k];
    }
    if (tid == 1023) {
        hist[GRID_CELLS] = run;
        cellStart[GRID_CELLS] = run;
    }
    __syncthreads();
    // Items of a cell keep ascending keypoint order.  In chunks of 64 keypoints a keypoint is ranked among
    // the lower lanes that fall in the same cell (all 16 waves, a chunk each); then one wave walks the
    // chunks in order and advances each cell's write cursor (hist[] holds it) by the number the chunk
    // put there.  Linear in the number of keypoints; 8192 keypoints per round of the two steps.
    __shared__ u16 sCell[8192];
    __shared__ u16 sRank[8192];  // rank in the chunk | (count in the chunk << 8)
    for (int sb = 0; sb < f.n; sb += 8192) {
        const int m = min(8192, f.n - sb);
        for (int cb = wv * 64; cb < m; cb += 1024) {
            const int k = cb + lane;
            const int c = k < m ? (int)cellOf[sb + k] : 0xFFFF;
            int r = 0, cnt = 0;
#pragma unroll 8
            for (int l = 0; l < 64; l++) {
                const int cl = __builtin_amdgcn_readlane(c, l);
                const bool same = cl == c;
                r += (same && l < lane) ? 1 : 0;
                cnt += same ? 1 : 0;
            }
            if (k < m) {
                sCell[k] = (u16)c;
                sRank[k] = (u16)(r | (cnt << 8));
            }
        }
        __syncthreads();
        if (tid < 64) {
            for (int cb = 0; cb < m; cb += 64) {
                const int k = cb + lane;
                if (k < m) {
                    const int c = sCell[k];
                    const int rc = sRank[k];
                    const int r = rc & 0xFF, cnt = rc >> 8;
                    if (c != 0xFFFF) {
                        const int start = hist[c];
                        cellItems[start + r] = sb + k;
                        if (r == cnt - 1) hist[c] = start + cnt;  // the last of the cell in this chunk moves the cursor
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ window walk
struct Window {
    int x0, x1, y0, y1;
    bool empty;
};

// Cell range of Frame::GetFeaturesInArea (reference src/Frame.cc:442-456).
__device__ __forceinline__ Window window_cells(const DevFrame& f, float x, float y, float r) {
    Window w;
    w.empty = true;
    w.x0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, f.minX), r), f.invW)));
    if (w.x0 >= GRID_COLS) return w;
    w.x1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, f.minX), r), f.invW)));
    if (w.x1 < 0) return w;
    w.y0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, f.minY), r), f.invH)));
    if (w.y0 >= GRID_ROWS) return w;
    w.y1 = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, f.minY), r), f.invH)));
    if (w.y1 < 0) return w;
    w.empty = w.x1 < w.x0 || w.y1 < w.y0;
    return w;
}

struct Query {
    float x, y, r;
    int minLevel, maxLevel;
    bool useRight;     // apply the right-image gate (reference src/ORBmatcher.cc:1615-1621)
    float ur, rGate;
};

// One wave walks a query's window in the reference's order (ix outer, iy
// inner, bucket order) and calls emit(pos, idx) for each accepted candidate
// with its ordinal `pos`; returns the number accepted.  For one ix the
// cells iy0..iy1 are a contiguous CSR run.
template <typename Emit>
__device__ __forceinline__ int walk_window(const DevFrame& f, const int* __restrict__ cellStart,
                                           const int* __restrict__ cellItems, const float* __restrict__ uRight,
                                           const Query& q, Emit emit) {
    const Window w = window_cells(f, q.x, q.y, q.r);
    if (w.empty) return 0;
    const int lane = threadIdx.x & 63;
    const bool checkLevels = (q.minLevel > 0) || (q.maxLevel >= 0);
    int count = 0;
    for (int ix = w.x0; ix <= w.x1; ix++) {
        const int b = cellStart[ix * GRID_ROWS + w.y0], e = cellStart[ix * GRID_ROWS + w.y1 + 1];
        for (int base = b; base < e; base += 64) {
            const int k = base + lane;
            bool ok = false;
            int idx = -1;
            if (k < e) {
                idx = cellItems[k];
                const orbx_keypoint kp = f.kps[idx];
                ok = true;
                if (checkLevels) {
                    if (kp.octave < q.minLevel) ok = false;
                    if (q.maxLevel >= 0 && kp.octave > q.maxLevel) ok = false;
                }
                const float dx = __fsub_rn(kp.x, q.x), dy = __fsub_rn(kp.y, q.y);
                if (!(fabsf(dx) < q.r && fabsf(dy) < q.r)) ok = false;
                if (ok && q.useRight) {
                    const float urk = uRight[idx];
                    if (urk > 0) {
                        const float er = fabsf(__fsub_rn(q.ur, urk));
                        if (er > q.rGate) ok = false;
                    }
                }
            }
            int tot;
            const int off = wave_excl_count(ok, &tot);
            if (ok) emit(count + off, idx);
            count += tot;
        }
    }
    return count;
}

// The same walk with G (16 or 8) lanes per query, four or eight queries to a wave, for windows of a few cells (a
// frame-to-frame search at th 7..15 sees a dozen keypoints in its cell rectangle: one query per wave
// leaves 60 lanes idle behind four dependent loads per grid column).  The columns' CSR runs are
// concatenated (prefix sum over the columns, G at a time), so one trip takes G items across all
// columns in window order.  All 64 lanes stay in the loops (the shuffles and ballots need them);
// `active` masks a group without a query.
template <int G, typename Emit>
__device__ __forceinline__ int walk_window_sub(const DevFrame& f, const int* __restrict__ cellStart,
                                               const int* __restrict__ cellItems, const float* __restrict__ uRight,
                                               const Query& q, bool active, Emit emit) {
    const int lane = threadIdx.x & 63, sub = lane & (G - 1), shift = lane & (64 - G);
    Window w = window_cells(f, q.x, q.y, q.r);
    const bool live = active && !w.empty;
    const int ncols = live ? w.x1 - w.x0 + 1 : 0;
    const bool checkLevels = (q.minLevel > 0) || (q.maxLevel >= 0);
    int count = 0;
    for (int c0 = 0; __any(c0 < ncols); c0 += G) {
        const int c = c0 + sub;
        int b = 0, len = 0;
        if (c < ncols) {
            b = cellStart[(w.x0 + c) * GRID_ROWS + w.y0];
            len = cellStart[(w.x0 + c) * GRID_ROWS + w.y1 + 1] - b;
        }
        int incl = len;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            const int t = __shfl_up(incl, o, G);
            if (sub >= o) incl += t;
        }
        const int total = __shfl(incl, G - 1, G);
        for (int t0 = 0; __any(t0 < total); t0 += G) {
            const int t = t0 + sub;
            // column of item t = number of columns whose run ends at or before t
            int col = 0;
            for (int j = 0; j < G; j++) {
                if (!__any(c0 + j < ncols)) break;
                col += __shfl(incl, j, G) <= t ? 1 : 0;
            }
            col = min(col, G - 1);
            const int bcol = __shfl(b, col, G);
            const int ecol = __shfl(incl - len, col, G);
            bool ok = false;
            int idx = -1;
            if (t < total) {
                idx = cellItems[bcol + (t - ecol)];
                const orbx_keypoint kp = f.kps[idx];
                ok = true;
                if (checkLevels) {
                    if (kp.octave < q.minLevel) ok = false;
                    if (q.maxLevel >= 0 && kp.octave > q.maxLevel) ok = false;
                }
                const float dx = __fsub_rn(kp.x, q.x), dy = __fsub_rn(kp.y, q.y);
                if (!(fabsf(dx) < q.r && fabsf(dy) < q.r)) ok = false;
                if (ok && q.useRight) {
                    const float urk = uRight[idx];
                    if (urk > 0) {
                        const float er = fabsf(__fsub_rn(q.ur, urk));
                        if (er > q.rGate) ok = false;
                    }
                }
            }
            const u32 gm = (u32)(__ballot(ok) >> shift) & ((1u << G) - 1u);
            if (ok) emit(count + __popc(gm & ((1u << sub) - 1u)), idx);
            count += __popc(gm);
        }
    }
    return count;
}

__device__ __forceinline__ int desc_distance(const u8* a, const u8* b) {
    const uint4* pa = (const uint4*)a;
    const uint4* pb = (const uint4*)b;
    return hamming256(pa[0], pa[1], pb[0], pb[1]);
}

struct QueryArrays {
    const float *x, *y, *r;
    const int *minLevel, *maxLevel;
    const u8* desc;  // may be null
};

// mode 0: counts only.  mode 1: fill rows of fixed capacity maxCand.  mode 2: fill CSR rows at offs[q].
__global__ __launch_bounds__(256) void k_window(DevFrame train, const int* __restrict__ cellStart,
                                                 const int* __restrict__ cellItems, QueryArrays qa, int nq,
                                                 int mode, int maxCand, const int* __restrict__ offs,
                                                 int* candIdx, int* candDist, int* counts) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    Query qq;
    qq.x = qa.x[q];
    qq.y = qa.y[q];
    qq.r = qa.r[q];
    qq.minLevel = qa.minLevel ? qa.minLevel[q] : -1;
    qq.maxLevel = qa.maxLevel ? qa.maxLevel[q] : -1;
    qq.useRight = false;
    qq.ur = qq.rGate = 0.f;
    const u8* qd = qa.desc ? qa.desc + (i64)q * 32 : nullptr;
    i64 rowBase = 0;
    int cap = 0;
    if (mode == 1) {
        rowBase = (i64)q * maxCand;
        cap = maxCand;
    } else if (mode == 2) {
        rowBase = offs[q];
        cap = offs[q + 1] - offs[q];
    }
    const int n = walk_window(train, cellStart, cellItems, nullptr, qq, [&](int pos, int idx) {
        if (mode != 0 && pos < cap) {
            candIdx[rowBase + pos] = idx;
            if (candDist) candDist[rowBase + pos] = qd ? desc_distance(qd, train.desc + (i64)idx * 32) : 0;
        }
    });
    if ((threadIdx.x & 63) == 0 && counts) counts[q] = n;
}

// Exclusive scan of n ints by one workgroup (n is a few thousand).
__device__ __forceinline__ void scan_body(const int* in, int* out, int n) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v = i < n ? in[i] : 0;
        int inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int b = carry;
        for (int k = 0; k < wv; k++) b += wsum[k];
        if (i < n) out[i] = b + inc - v;
        __syncthreads();
        if (tid == 1023) carry = b + inc;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
}

// ------------------------------------------------------------------ projected searches
// All SearchByProjection variants of the reference share one shape: per query (a map
// point already projected into the train frame by the host) a window, a level range, an
// optional right-image gate and a descriptor; then a scan of the candidates in
// GetFeaturesInArea order under a claim rule, and (some variants) a rotation histogram.
struct ProjQueries {
    int n;
    const u8* valid;
    const float *u, *v, *radius;
    const int *minLevel, *maxLevel;
    const u8* desc;         // n x 32, pMP->GetDescriptor()
    const float* angle;     // for the rotation histogram; may be null
    const u8* hasObs;       // pMP->Observations() > 0
    const float* uRight;    // projected right coordinate for the stereo gate; may be null
    const float* rightGate; // largest accepted |uRight - mvuRight[i2]|
};

struct ProjPolicy {
    int claimRule;   // 0: skip train keypoints whose holder has observations; 1: skip any held keypoint
    int maxDist;
    int ratioTest;   // level-aware ratio test of SearchByProjection(Frame&, vector<MapPoint*>&)
    float nnRatio;
    int checkOri;
};

__device__ __forceinline__ bool projq_query(const ProjQueries& q, int i, Query* out) {
    if (!q.valid[i]) return false;
    out->x = q.u[i];
    out->y = q.v[i];
    out->r = q.radius[i];
    out->minLevel = q.minLevel[i];
    out->maxLevel = q.maxLevel[i];
    out->useRight = q.uRight != nullptr;
    out->ur = q.uRight ? q.uRight[i] : 0.f;
    out->rGate = q.rightGate ? q.rightGate[i] : 0.f;
    return true;
}

// Candidate lists of every query (mode 0: counts; mode 2: CSR fill with distances).
// The right-image gate is state-free and applied here.
__device__ __forceinline__ void projq_candidates_body(DevFrame train, const int* __restrict__ cellStart,
                                                           const int* __restrict__ cellItems,
                                                           const float* __restrict__ trainURight, ProjQueries q, int mode,
                                                           const int* __restrict__ offs, int* candIdx, int* candDist,
                                                           int* counts, u32* slab = nullptr, int slabCap = 0,
                                                           int* overflow = nullptr) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= q.n) return;
    Query qq;
    int n = 0;
    if (projq_query(q, i, &qq)) {
        const u8* qd = q.desc + (i64)i * 32;
        const i64 rowBase = mode == 2 ? offs[i] : 0;
        n = walk_window(train, cellStart, cellItems, trainURight, qq, [&](int pos, int idx) {
            if (mode == 2) {
                candIdx[rowBase + pos] = idx;
                candDist[rowBase + pos] = desc_distance(qd, train.desc + (i64)idx * 32);
            } else if (mode == 3 && pos < slabCap) {
                // one walk: the first slabCap candidates of the query go to its fixed slot, distance included
                slab[(i64)i * slabCap + pos] = ((u32)desc_distance(qd, train.desc + (i64)idx * 32) << 16) | (u32)idx;
            }
        });
    }
    if ((threadIdx.x & 63) == 0 && (mode == 0 || mode == 3)) {
        counts[i] = n;
        if (mode == 3 && n > slabCap) atomicOr(overflow, 1);
    }
}

// G lanes per query; same outputs as projq_candidates_body.
template <int G>
__device__ __forceinline__ void projq_candidates_sub_body(DevFrame train, const int* __restrict__ cellStart,
                                                          const int* __restrict__ cellItems,
                                                          const float* __restrict__ trainURight, ProjQueries q, int mode,
                                                          const int* __restrict__ offs, int* candIdx, int* candDist,
                                                          int* counts, u32* slab, int slabCap, int* overflow) {
    const int i = blockIdx.x * (256 / G) + threadIdx.x / G;
    Query qq;
    qq.x = qq.y = 0.f;
    qq.r = -1.f;
    qq.minLevel = qq.maxLevel = -1;
    qq.useRight = false;
    qq.ur = qq.rGate = 0.f;
    const bool inRange = i < q.n;
    const bool active = inRange && projq_query(q, i, &qq);
    const u8* qd = q.desc + (i64)(inRange ? i : 0) * 32;
    const i64 rowBase = (mode == 2 && inRange) ? offs[i] : 0;
    const int n = walk_window_sub<G>(train, cellStart, cellItems, trainURight, qq, active, [&](int pos, int idx) {
        if (mode == 2) {
            candIdx[rowBase + pos] = idx;
            candDist[rowBase + pos] = desc_distance(qd, train.desc + (i64)idx * 32);
        } else if (mode == 3 && pos < slabCap) {
            slab[(i64)i * slabCap + pos] = ((u32)desc_distance(qd, train.desc + (i64)idx * 32) << 16) | (u32)idx;
        }
    });
    if (inRange && (threadIdx.x & (G - 1)) == 0 && (mode == 0 || mode == 3)) {
        counts[i] = n;
        if (mode == 3 && n > slabCap) atomicOr(overflow, 1);
    }
}

// Per-query inputs of SearchByProjection(cur, last): reference src/ORBmatcher.cc:1564-1598,
// 1615-1621.  One thread per last-frame keypoint.
__device__ __forceinline__ void m4_queries_body(DevFrame cur, DevFrame last, const u8* valid, const float* u, const float* v,
                             const float* invz, const float* scaleFactors, float bf, float th, int direction,
                             u8* qValid, float* qRadius, int* qMin, int* qMax, float* qAngle, float* qUR) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= last.n) return;
    bool ok = valid[i] != 0;
    const float invzc = invz[i];
    if (invzc < 0) ok = false;
    const float uu = u[i], vv = v[i];
    if (uu < cur.minX || uu > cur.maxX) ok = false;
    if (vv < cur.minY || vv > cur.maxY) ok = false;
    const orbx_keypoint kp = last.kps[i];
    const int oct = kp.octave;
    const float radius = __fmul_rn(th, scaleFactors[oct]);
    qValid[i] = ok ? 1 : 0;
    qRadius[i] = radius;
    if (direction == 1) {
        qMin[i] = oct;
        qMax[i] = -1;
    } else if (direction == 2) {
        qMin[i] = 0;
        qMax[i] = oct;
    } else {
        qMin[i] = oct - 1;
        qMax[i] = oct + 1;
    }
    qAngle[i] = kp.angle;
    qUR[i] = __fsub_rn(uu, __fmul_rn(bf, invzc));
}

// reference src/ORBmatcher.cc:1849-1894
__device__ void three_maxima(const int* histo, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
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
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
        ind2 = -1;
        ind3 = -1;
    } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
        ind3 = -1;
    }
}

__device__ __forceinline__ int rot_bin(float a1, float a2) {
    // reference src/ORBmatcher.cc:1639-1644
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = __fsub_rn(a1, a2);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, factor));
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// Wave-wide minimum of a 64-bit key.
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const u64 t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}

// Sequential part of the projected searches: queries in index order, candidates of one
// query spread over the 64 lanes.  One wave.  Best = first minimum among admissible
// candidates (strict '<' scan starting at 256).  With ratioTest the second-best distance
// and the pyramid level of the keypoint holding it are reproduced as the sequential scan
// leaves them: when the final best arrives the previous best becomes second; afterwards
// only a strictly smaller distance replaces it.
__device__ __forceinline__ void projq_resolve_body(DevFrame train, ProjQueries q, ProjPolicy pol,
                                                        const int* __restrict__ offs, const int* __restrict__ candIdx,
                                                        const int* __restrict__ candDist, int* claimed, u8* claimedObs,
                                                        int* qBest, int* qBin, int* nmatchesOut) {
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int keep[3];
    const int lane = threadIdx.x;
    if (lane < HISTO_LENGTH) hist[lane] = 0;
    __syncthreads();
    int nmatches = 0;
    for (int i = 0; i < q.n; i++) {
        const int b = offs[i], e = offs[i + 1];
        int best = -1;
        if (e > b) {
            u64 key = ~0ull;
            for (int k = b + lane; k < e; k += 64) {
                const int i2 = candIdx[k];
                const bool held = claimed[i2] >= 0 && (pol.claimRule == 1 || claimedObs[i2]);
                if (held) continue;
                const u64 kk = ((u64)(u32)candDist[k] << 32) | (u32)(k - b);
                key = kk < key ? kk : key;
            }
            key = wave_min_u64(key);
            const int bestDist = (int)(key >> 32);
            if (key != ~0ull && bestDist < 256 && bestDist <= pol.maxDist) {
                const int bestPos = (int)(key & 0xFFFFFFFFu);
                best = candIdx[b + bestPos];
                if (pol.ratioTest) {
                    u64 kp = ~0ull, kq = ~0ull;  // first minimum before / after the best
                    for (int k = b + lane; k < e; k += 64) {
                        const int pos = k - b;
                        if (pos == bestPos) continue;
                        const int i2 = candIdx[k];
                        const bool held = claimed[i2] >= 0 && (pol.claimRule == 1 || claimedObs[i2]);
                        if (held) continue;
                        const u64 kk = ((u64)(u32)candDist[k] << 32) | (u32)pos;
                        if (pos < bestPos) kp = kk < kp ? kk : kp;
                        else kq = kk < kq ? kk : kq;
                    }
                    kp = wave_min_u64(kp);
                    kq = wave_min_u64(kq);
                    int second = 256, level2 = -1;
                    if (kp != ~0ull && (int)(kp >> 32) < 256) {
                        second = (int)(kp >> 32);
                        level2 = train.kps[candIdx[b + (int)(kp & 0xFFFFFFFFu)]].octave;
                    }
                    if (kq != ~0ull && (int)(kq >> 32) < second) {
                        second = (int)(kq >> 32);
                        level2 = train.kps[candIdx[b + (int)(kq & 0xFFFFFFFFu)]].octave;
                    }
                    const int level = train.kps[best].octave;
                    if (level == level2 && (float)bestDist > __fmul_rn(pol.nnRatio, (float)second)) best = -1;
                }
            }
        }
        int bin = -1;
        if (best >= 0) {
            nmatches++;
            if (pol.checkOri) bin = rot_bin(q.angle[i], train.kps[best].angle);
            if (lane == 0) {
                claimed[best] = i;
                claimedObs[best] = q.hasObs ? q.hasObs[i] : 1;
                if (bin >= 0) hist[bin]++;
            }
            __threadfence_block();
        }
        if (lane == 0) {
            qBest[i] = best;
            qBin[i] = bin;
        }
    }
    __syncthreads();
    if (pol.checkOri) {
        if (lane == 0) {
            int i1 = -1, i2 = -1, i3 = -1;
            three_maxima(hist, HISTO_LENGTH, i1, i2, i3);
            keep[0] = i1;
            keep[1] = i2;
            keep[2] = i3;
        }
        __syncthreads();
        int removed = 0;
        for (int i = lane; i < q.n; i += 64) {
            const int bin = qBin[i];
            if (bin >= 0 && bin != keep[0] && bin != keep[1] && bin != keep[2]) {
                claimed[qBest[i]] = -1;
                claimedObs[qBest[i]] = 0;
                removed++;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o);
        nmatches -= removed;
    }
    if (lane == 0) *nmatchesOut = nmatches;
}

// ---- kernels over the bodies above: one problem per launch, or one problem ("job") per
// blockIdx.y / blockIdx.x for batches of independent frame pairs
struct ProjJob {
    DevFrame train, last;            // train = the frame that is searched; last = source of the queries (M4 only)
    const float* trainURight;
    ProjQueries q;
    int *cellStart, *cellItems;
    u16* cellOf;
    int *counts, *offs, *candIdx, *candDist;
    int* claimed;
    u8* claimedObs;
    int *qBest, *qBin, *nmatches;
    u32* slab;                       // slabCap packed candidates per query, written by the single walk
    int slabCap;
    int* overflow;                   // set when some query has more candidates than a slab slot holds
    // inputs of the frame-to-frame form (reference src/ORBmatcher.cc:1533-1678)
    const u8* m4Valid;
    const float *m4U, *m4V, *m4Invz;
    int direction;
    u8* qValid;
    float *qRadius, *qAngle, *qUR;
    int *qMin, *qMax;
    // SearchForInitialization over a table of pairs (reference src/ORBmatcher.cc:497-617)
    float* initPrev;                 // vbPrevMatched as x, y pairs, read and updated
    float *initU, *initV;            // the same, split, as the candidate walk wants its queries
    int* initM21;                    // vnMatches21 of the wave-per-pair form
};

__global__ __launch_bounds__(1024) void k_grid_build(DevFrame f, int* cellStart, int* cellItems, u16* cellOf) {
    grid_build_body(f, cellStart, cellItems, cellOf);
}
__global__ __launch_bounds__(1024) void k_grid_build_jobs(const ProjJob* __restrict__ jobs) {
    const ProjJob& J = jobs[blockIdx.x];
    grid_build_body(J.train, J.cellStart, J.cellItems, J.cellOf);
}
__global__ __launch_bounds__(1024) void k_scan(const int* in, int* out, int n) { scan_body(in, out, n); }
__global__ __launch_bounds__(1024) void k_scan_jobs(const ProjJob* __restrict__ jobs, int* totals) {
    const ProjJob& J = jobs[blockIdx.x];
    scan_body(J.counts, J.offs, J.q.n);
    __syncthreads();
    if (threadIdx.x == 0) totals[blockIdx.x] = J.offs[J.q.n];
}
__global__ __launch_bounds__(256) void k_projq_candidates(DevFrame train, const int* __restrict__ cellStart,
                                                           const int* __restrict__ cellItems,
                                                           const float* __restrict__ trainURight, ProjQueries q, int mode,
                                                           const int* __restrict__ offs, int* candIdx, int* candDist,
                                                           int* counts, u32* slab, int slabCap, int* overflow) {
    projq_candidates_body(train, cellStart, cellItems, trainURight, q, mode, offs, candIdx, candDist, counts, slab, slabCap,
                          overflow);
}
__global__ __launch_bounds__(256) void k_projq_candidates_jobs(const ProjJob* __restrict__ jobs, int mode) {
    const ProjJob& J = jobs[blockIdx.y];
    projq_candidates_body(J.train, J.cellStart, J.cellItems, J.trainURight, J.q, mode, J.offs, J.candIdx, J.candDist, J.counts,
                          J.slab, J.slabCap, J.overflow);
}
template <int G>
__global__ __launch_bounds__(256) void k_projq_candidates_jobs_sub(const ProjJob* __restrict__ jobs, int mode) {
    const ProjJob& J = jobs[blockIdx.y];
    if ((int)blockIdx.x * (256 / G) >= J.q.n) return;
    projq_candidates_sub_body<G>(J.train, J.cellStart, J.cellItems, J.trainURight, J.q, mode, J.offs, J.candIdx, J.candDist,
                              J.counts, J.slab, J.slabCap, J.overflow);
}
__global__ void k_m4_queries(DevFrame cur, DevFrame last, const u8* valid, const float* u, const float* v,
                             const float* invz, const float* scaleFactors, float bf, float th, int direction,
                             u8* qValid, float* qRadius, int* qMin, int* qMax, float* qAngle, float* qUR) {
    m4_queries_body(cur, last, valid, u, v, invz, scaleFactors, bf, th, direction, qValid, qRadius, qMin, qMax, qAngle, qUR);
}
__global__ void k_m4_queries_jobs(const ProjJob* __restrict__ jobs, const float* scaleFactors, float bf, float th) {
    const ProjJob& J = jobs[blockIdx.y];
    m4_queries_body(J.train, J.last, J.m4Valid, J.m4U, J.m4V, J.m4Invz, scaleFactors, bf, th, J.direction, J.qValid, J.qRadius,
                    J.qMin, J.qMax, J.qAngle, J.qUR);
}
__global__ __launch_bounds__(64) void k_projq_resolve(DevFrame train, ProjQueries q, ProjPolicy pol,
                                                        const int* __restrict__ offs, const int* __restrict__ candIdx,
                                                        const int* __restrict__ candDist, int* claimed, u8* claimedObs,
                                                        int* qBest, int* qBin, int* nmatchesOut) {
    projq_resolve_body(train, q, pol, offs, candIdx, candDist, claimed, claimedObs, qBest, qBin, nmatchesOut);
}
// Wave-wide minimum of a 32-bit key, in every lane's return value: DPP row shifts, then the two row broadcasts.
__device__ __forceinline__ u32 wave_min_u32_dpp(u32 v) {
    v = min(v, (u32)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = min(v, (u32)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = min(v, (u32)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = min(v, (u32)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = min(v, (u32)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
    v = min(v, (u32)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2, 3
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

// The same resolution with the whole problem in LDS (candidates as distance << 16 | train index, the
// per-query offsets, the claim state per train keypoint, the pyramid levels for the ratio test), used
// when it fits (the host checks the byte count).  Queries are taken 64 at a time, one per lane:
//   1. every pending lane scans its candidates exactly as the reference's loop does, against the state
//      left by everything committed so far;
//   2. lanes publish the keypoint they would claim (LDS atomic min of the lane number), and a lane
//      whose best or second-best candidate was published by an EARLIER pending lane is "dirty";
//   3. all pending lanes below the lowest dirty one commit: neither candidate their outcome rests on is
//      touched by an earlier uncommitted query, so what they computed is what the sequential order
//      gives, and no two of them claim the same keypoint; the rest repeat from 1 against the new state.
// The lowest pending lane is never dirty, so every round commits at least one query; two queries among
// 64 wanting the same keypoint is rare, and a chunk takes one or two rounds.  Claims live in LDS and are written back once at the end.
template <bool RATIO>
__device__ __forceinline__ void projq_resolve_lds_body(u8* rsm, DevFrame train, ProjQueries q, ProjPolicy pol,
                                                        const int* __restrict__ offs, const int* __restrict__ candIdx,
                                                        const int* __restrict__ candDist, int* claimed, u8* claimedObs,
                                                        int* qBest, int* qBin, int* nmatchesOut, int total,
                                                        const u32* __restrict__ slab = nullptr, int slabCap = 0) {
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int keep[3];
    __shared__ int counters[2];  // matches, removed
    const int tid = threadIdx.x, lane = tid & 63;
    const int n = q.n, nt = train.n;
    u32* cand = (u32*)rsm;
    int* off = (int*)(cand + total);          // off[i] is overwritten with query i's result once it is known
    u32* first = (u32*)(off + n + 1);          // lowest pending lane that wants this keypoint in the current round
    int* owner = (int*)(first + nt);           // claimed[]
    int* level = owner + nt;                   // nt entries with the ratio test, none without
    u8* held = (u8*)(level + (RATIO ? nt : 0));
    u8* ownerObs = held + nt;                  // claimedObs[]
    u8* obs = ownerObs + nt;
    if (slab) {
        for (int i = tid; i < n; i += 256) {
            const int b0 = offs[i], c = offs[i + 1] - b0;
            for (int pp = 0; pp < c; pp++) cand[b0 + pp] = slab[(i64)i * slabCap + pp];
        }
    } else {
        for (int k = tid; k < total; k += 256) cand[k] = ((u32)candDist[k] << 16) | (u32)candIdx[k];
    }
    for (int i = tid; i <= n; i += 256) off[i] = offs[i];
    for (int i = tid; i < nt; i += 256) {
        const int c = claimed[i];
        const u8 o = claimedObs[i];
        owner[i] = c;
        ownerObs[i] = o;
        held[i] = (c >= 0 && (pol.claimRule == 1 || o)) ? 1 : 0;
        first[i] = 0xFFFFFFFFu;
        if (RATIO) level[i] = train.kps[i].octave;
    }
    for (int i = tid; i < n; i += 256) obs[i] = q.hasObs ? q.hasObs[i] : 1;
    if (tid < HISTO_LENGTH) hist[tid] = 0;
    if (tid < 2) counters[tid] = 0;
    __syncthreads();
    if (tid < 64) {
        int nmatches = 0;
        for (int q0 = 0; q0 < n; q0 += 64) {
            const int i = q0 + lane;
            const bool have = i < n;
            const int b = have ? off[i] : 0, e = have ? off[i + 1] : 0;
            // the first RCACHE candidates of the lane stay in registers for all rounds of the chunk, so a
            // round is a few batches of independent LDS reads rather than a chain of dependent ones
            constexpr int RCACHE = 8;
            u32 cw[RCACHE];
#pragma unroll
            for (int j = 0; j < RCACHE; j++) cw[j] = b + j < e ? cand[b + j] : 0xFFFFFFFFu;
            u64 pending = __ballot(have);
            while (pending) {
                const bool mine = (pending >> lane) & 1ull;
                int result = -1;
                int depBest = -1, depSecond = -1;  // the only two candidates the outcome depends on
                {
                    // Keys (distance << 16 | position) of the candidates that are not held; the reference's
                    // strict '<' scan from 256 keeps the first minimum, i.e. the smallest key.  Its second
                    // best (src/ORBmatcher.cc:121-134) is what the scan leaves: the first minimum before
                    // the best, replaced by the first minimum after it only if strictly smaller.
                    u32 key[RCACHE];
#pragma unroll
                    for (int j = 0; j < RCACHE; j++) {
                        const u32 c = cw[j];  // 0xFFFFFFFF past the end of the list: distance field >= 256
                        const bool ok = mine && b + j < e && held[c & 0xFFFFu] == 0;
                        key[j] = ok ? ((c & 0xFFFF0000u) | (u32)j) : 0xFFFFFFFFu;
                    }
                    u32 bestKey = 0xFFFFFFFFu;
#pragma unroll
                    for (int j = 0; j < RCACHE; j++) bestKey = min(bestKey, key[j]);
                    if (mine) {
                        for (int k = b + RCACHE; k < e; k++) {
                            const u32 c = cand[k];
                            if (held[c & 0xFFFFu] == 0) bestKey = min(bestKey, (c & 0xFFFF0000u) | (u32)(k - b));
                        }
                    }
                    const int bestDist = (int)(bestKey >> 16);
                    if (bestKey != 0xFFFFFFFFu && bestDist < 256 && bestDist <= pol.maxDist) {
                        const int bestPos = (int)(bestKey & 0xFFFFu);
                        u32 bw = cw[0];
#pragma unroll
                        for (int j = 1; j < RCACHE; j++) bw = bestPos == j ? cw[j] : bw;
                        if (bestPos >= RCACHE) bw = cand[b + bestPos];
                        result = (int)(bw & 0xFFFFu);
                        depBest = result;
                        if (RATIO) {
                            u32 kp = 0xFFFFFFFFu, kq = 0xFFFFFFFFu;
#pragma unroll
                            for (int j = 0; j < RCACHE; j++) {
                                kp = j < bestPos ? min(kp, key[j]) : kp;
                                kq = j > bestPos ? min(kq, key[j]) : kq;
                            }
                            for (int k = b + RCACHE; k < e; k++) {
                                const int pos = k - b;
                                if (pos == bestPos) continue;
                                const u32 c = cand[k];
                                if (held[c & 0xFFFFu]) continue;
                                const u32 kk = (c & 0xFFFF0000u) | (u32)pos;
                                if (pos < bestPos) kp = min(kp, kk);
                                else kq = min(kq, kk);
                            }
                            int second = 256, level2 = -1;
                            if (kp != 0xFFFFFFFFu && (int)(kp >> 16) < 256) {
                                second = (int)(kp >> 16);
                                depSecond = (int)(cand[b + (int)(kp & 0xFFFFu)] & 0xFFFFu);
                                level2 = level[depSecond];
                            }
                            if (kq != 0xFFFFFFFFu && (int)(kq >> 16) < second) {
                                second = (int)(kq >> 16);
                                depSecond = (int)(cand[b + (int)(kq & 0xFFFFu)] & 0xFFFFu);
                                level2 = level[depSecond];
                            }
                            if (level[result] == level2 && (float)bestDist > __fmul_rn(pol.nnRatio, (float)second)) result = -1;
                        }
                    }
                }
                const bool wants = mine && result >= 0;
                if (wants) atomicMin(&first[result], (u32)lane);
                // A claim by an earlier query only removes a candidate from this one's list.  Removing a
                // candidate that is neither the best nor the second best the scan settled on leaves both as
                // they are (each is a minimum over its side of the list), so those two are the whole dependency.
                const u32 fa = depBest >= 0 ? first[depBest] : 0xFFFFFFFFu;
                const u32 fb = depSecond >= 0 ? first[depSecond] : 0xFFFFFFFFu;
                const bool dirty = mine && min(fa, fb) < (u32)lane;
                if (wants) first[result] = 0xFFFFFFFFu;  // leave the table clean for the next round
                const u64 dm = __ballot(mine && dirty);
                const int d = dm ? __ffsll((unsigned long long)dm) - 1 : 64;
                const bool commit = mine && lane < d;
                if (commit) {
                    off[i] = result;
                    if (result >= 0) {
                        const u8 o = obs[i];
                        held[result] = (pol.claimRule == 1 || o) ? 1 : 0;
                        owner[result] = i;
                        ownerObs[result] = o;
                    }
                }
                nmatches += __popcll(__ballot(commit && result >= 0));
                pending &= ~__ballot(commit);
            }
        }
        if (lane == 0) counters[0] = nmatches;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int best = off[i];
        int bin = -1;
        if (best >= 0 && pol.checkOri) {
            bin = rot_bin(q.angle[i], train.kps[best].angle);
            atomicAdd(&hist[bin], 1);
        }
        qBest[i] = best;
        qBin[i] = bin;
    }
    __syncthreads();
    if (pol.checkOri) {
        if (tid == 0) {
            int i1 = -1, i2 = -1, i3 = -1;
            three_maxima(hist, HISTO_LENGTH, i1, i2, i3);
            keep[0] = i1;
            keep[1] = i2;
            keep[2] = i3;
        }
        __syncthreads();
        int removed = 0;
        for (int i = tid; i < n; i += 256) {
            const int best = off[i];
            if (best < 0) continue;
            const int bin = rot_bin(q.angle[i], train.kps[best].angle);
            if (bin != keep[0] && bin != keep[1] && bin != keep[2]) {
                owner[best] = -1;
                ownerObs[best] = 0;
                removed++;
            }
        }
        if (removed) atomicAdd(&counters[1], removed);
        __syncthreads();
    }
    for (int i = tid; i < nt; i += 256) {
        claimed[i] = owner[i];
        claimedObs[i] = ownerObs[i];
    }
    if (tid == 0) *nmatchesOut = counters[0] - counters[1];
}
template <bool RATIO>
__global__ __launch_bounds__(256) void k_projq_resolve_lds(DevFrame train, ProjQueries q, ProjPolicy pol,
                                                            const int* __restrict__ offs, const int* __restrict__ candIdx,
                                                            const int* __restrict__ candDist, int* claimed, u8* claimedObs,
                                                            int* qBest, int* qBin, int* nmatchesOut, int total,
                                                            const u32* __restrict__ slab, int slabCap) {
    extern __shared__ __align__(16) u8 rsm[];
    projq_resolve_lds_body<RATIO>(rsm, train, q, pol, offs, candIdx, candDist, claimed, claimedObs, qBest, qBin, nmatchesOut, total,
                                  slab, slabCap);
}
// The same launched before the host knows the candidate total: the kernel reads it, and declines (flag
// set, nothing else written) when a query overflowed its slot or the problem does not fit the LDS it
// was given; the host then takes the two-walk route.  Saves the host round trip in the middle of a call.
template <bool RATIO>
__global__ __launch_bounds__(256) void k_projq_resolve_lds_auto(DevFrame train, ProjQueries q, ProjPolicy pol,
                                                                 const int* __restrict__ offs, int* claimed, u8* claimedObs,
                                                                 int* qBest, int* qBin, int* nmatchesOut,
                                                                 const u32* __restrict__ slab, int slabCap,
                                                                 const int* __restrict__ overflow, int fixedBytes, int ldsBytes,
                                                                 int* declined) {
    extern __shared__ __align__(16) u8 rsm[];
    const int total = offs[q.n];
    if (*overflow || total >= 65536 || (i64)total * 4 + fixedBytes > (i64)ldsBytes) {
        if (threadIdx.x == 0) *declined = 1;
        return;
    }
    projq_resolve_lds_body<RATIO>(rsm, train, q, pol, offs, nullptr, nullptr, claimed, claimedObs, qBest, qBin, nmatchesOut, total,
                                  slab, slabCap);
}
// One job (frame pair) per workgroup; the dynamic LDS is sized by the host for the largest job.
template <bool RATIO>
__global__ __launch_bounds__(256) void k_projq_resolve_lds_jobs(const ProjJob* __restrict__ jobs, ProjPolicy pol, int useSlab) {
    extern __shared__ __align__(16) u8 rsm[];
    const ProjJob& J = jobs[blockIdx.x];
    projq_resolve_lds_body<RATIO>(rsm, J.train, J.q, pol, J.offs, J.candIdx, J.candDist, J.claimed, J.claimedObs, J.qBest, J.qBin,
                                  J.nmatches, J.offs[J.q.n], useSlab ? J.slab : nullptr, J.slabCap);
}
__global__ __launch_bounds__(64) void k_projq_resolve_jobs(const ProjJob* __restrict__ jobs, ProjPolicy pol) {
    const ProjJob& J = jobs[blockIdx.x];
    projq_resolve_body(J.train, J.q, pol, J.offs, J.candIdx, J.candDist, J.claimed, J.claimedObs, J.qBest, J.qBin, J.nmatches);
}

// ------------------------------------------------------------------ SearchForInitialization
// Sequential part (reference src/ORBmatcher.cc:515-591): one wave, queries in
// index order; vMatchedDistance / vnMatches21 / vnMatches12 live in global
// scratch written and read by this wave only.
__device__ __forceinline__ void init_resolve_body(DevFrame f1, DevFrame f2, const int* __restrict__ offs,
                                                  const int* __restrict__ candIdx, const int* __restrict__ candDist,
                                                  float nnRatio, int checkOri, int* matchedDist, int* matches21,
                                                  int* matches12, int* qBin, float* prevMatched, int* nmatchesOut) {
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int keep[3];
    const int lane = threadIdx.x;
    if (lane < HISTO_LENGTH) hist[lane] = 0;
    for (int i = lane; i < f1.n; i += 64) {
        matches12[i] = -1;
        qBin[i] = -1;
    }
    for (int i = lane; i < f2.n; i += 64) {
        matchedDist[i] = 0x7FFFFFFF;
        matches21[i] = -1;
    }
    __threadfence_block();
    __syncthreads();
    int nmatches = 0;
    for (int i1 = 0; i1 < f1.n; i1++) {
        const int b = offs[i1], e = offs[i1 + 1];
        if (e <= b) continue;
        // best = first minimum among admissible candidates; second = minimum of the rest
        u64 key = ~0ull;
        for (int k = b + lane; k < e; k += 64) {
            const int i2 = candIdx[k];
            const int d = candDist[k];
            if (matchedDist[i2] <= d) continue;
            const u64 kk = ((u64)(u32)d << 32) | (u32)(k - b);
            key = kk < key ? kk : key;
        }
        key = wave_min_u64(key);
        if (key == ~0ull) continue;
        const int bestDist = (int)(key >> 32), bestPos = (int)(key & 0xFFFFFFFFu);
        u64 key2 = ~0ull;
        for (int k = b + lane; k < e; k += 64) {
            if (k - b == bestPos) continue;
            const int i2 = candIdx[k];
            const int d = candDist[k];
            if (matchedDist[i2] <= d) continue;
            const u64 kk = (u64)(u32)d;
            key2 = kk < key2 ? kk : key2;
        }
        key2 = wave_min_u64(key2);
        const float second = key2 == ~0ull ? (float)0x7FFFFFFF : (float)(int)key2;
        if (bestDist <= TH_LOW && (float)bestDist < __fmul_rn(second, nnRatio)) {
            const int bestIdx2 = candIdx[b + bestPos];
            const int prev = matches21[bestIdx2];
            if (prev >= 0) nmatches--;
            nmatches++;
            int bin = -1;
            if (checkOri) bin = rot_bin(f1.kps[i1].angle, f2.kps[bestIdx2].angle);
            if (lane == 0) {
                if (prev >= 0) matches12[prev] = -1;
                matches12[i1] = bestIdx2;
                matches21[bestIdx2] = i1;
                matchedDist[bestIdx2] = bestDist;
                qBin[i1] = bin;
                if (bin >= 0) hist[bin]++;
            }
            __threadfence_block();
        }
    }
    __syncthreads();
    if (checkOri) {
        if (lane == 0) {
            int a = -1, b2 = -1, c = -1;
            three_maxima(hist, HISTO_LENGTH, a, b2, c);
            keep[0] = a;
            keep[1] = b2;
            keep[2] = c;
        }
        __syncthreads();
        int removed = 0;
        for (int i = lane; i < f1.n; i += 64) {
            const int bin = qBin[i];
            if (bin >= 0 && bin != keep[0] && bin != keep[1] && bin != keep[2] && matches12[i] >= 0) {
                matches12[i] = -1;
                removed++;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o);
        nmatches -= removed;
    }
    __threadfence_block();
    __syncthreads();
    for (int i = lane; i < f1.n; i += 64) {
        const int m = matches12[i];
        if (m >= 0) {
            prevMatched[2 * i] = f2.kps[m].x;
            prevMatched[2 * i + 1] = f2.kps[m].y;
        }
    }
    if (lane == 0) *nmatchesOut = nmatches;
}
__global__ __launch_bounds__(64) void k_init_resolve(DevFrame f1, DevFrame f2, const int* __restrict__ offs,
                                                       const int* __restrict__ candIdx, const int* __restrict__ candDist,
                                                       float nnRatio, int checkOri, int* matchedDist, int* matches21,
                                                       int* matches12, int* qBin, float* prevMatched, int* nmatchesOut) {
    init_resolve_body(f1, f2, offs, candIdx, candDist, nnRatio, checkOri, matchedDist, matches21, matches12, qBin, prevMatched,
                      nmatchesOut);
}

// Candidate rows sorted by (distance, position in the row): sorted[b + rank] = distance << 16 | index.
// One wave per query; the rank of an entry is the number of smaller keys in its row (keys are unique).
// With a row in this order the reference's scan (src/ORBmatcher.cc:529-548) reads off its result from
// the front: the best is the first admissible entry, the second-best distance that of the next one.
__device__ __forceinline__ void sort_rows_body(const int* __restrict__ offs, const int* __restrict__ candIdx,
                                               const int* __restrict__ candDist, u32* __restrict__ sorted, int nq) {
    __shared__ u32 rows[4][1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wv;
    if (q >= nq) return;
    u32* row = rows[wv];
    const int b = offs[q], n = offs[q + 1] - b;
    if (n <= 0) return;
    auto keyAt = [&](int k) { return ((u32)candDist[b + k] << 16) | (u32)(k & 0xFFFF); };
    const bool small = n <= 1024;
    if (small)
        for (int k = lane; k < n; k += 64) row[k] = keyAt(k);
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int j = t0 + lane;
        const u32 myKey = j < n ? keyAt(j) : 0xFFFFFFFFu;
        int rank = 0;
        if (small) {
#pragma unroll 4
            for (int k = 0; k < n; k++) rank += row[k] < myKey ? 1 : 0;
        } else {
            // rows longer than the staging buffer go through it tile by tile; positions wrap at 65536
            // only for rows no frame of 65000 keypoints can produce
            for (int tile0 = 0; tile0 < n; tile0 += 1024) {
                const int m = min(1024, n - tile0);
                for (int k = lane; k < m; k += 64) row[k] = keyAt(tile0 + k);
                for (int k = 0; k < m; k++) rank += row[k] < myKey ? 1 : 0;
            }
        }
        if (j < n) sorted[b + rank] = (myKey & 0xFFFF0000u) | (u32)candIdx[b + j];
    }
}
__global__ __launch_bounds__(256) void k_sort_rows(const int* __restrict__ offs, const int* __restrict__ candIdx,
                                                    const int* __restrict__ candDist, u32* __restrict__ sorted, int nq) {
    sort_rows_body(offs, candIdx, candDist, sorted, nq);
}

// SearchForInitialization's sequential part over sorted rows, all state in LDS, 64 queries at a time
// (one per lane).  A match only lowers vMatchedDistance of its target, i.e. removes that target from
// later queries' lists, so a query depends on earlier uncommitted ones only through its best and
// second-best entries: a lane whose best or second is wanted by an earlier pending lane waits a round,
// all pending lanes below the lowest such lane commit together, and the result is that of the serial loop.
__device__ __forceinline__ void init_resolve_sorted_body(u8* ism, DevFrame f1, DevFrame f2, const int* __restrict__ offs,
                                                         const u32* __restrict__ sorted, float nnRatio, int checkOri,
                                                         int* matches12, float* prevMatched, int* nmatchesOut) {
    __shared__ int hist[HISTO_LENGTH];
    __shared__ int keep[3];
    __shared__ int counters[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int n1 = f1.n, n2 = f2.n;
    int* mdist = (int*)ism;          // vMatchedDistance
    int* m21 = mdist + n2;           // vnMatches21
    u32* first = (u32*)(m21 + n2);   // lowest pending lane that wants this keypoint in the current round
    int* m12 = (int*)(first + n2);   // vnMatches12
    int* qbin = m12 + n1;
    for (int i = tid; i < n2; i += 256) {
        mdist[i] = 0x7FFFFFFF;
        m21[i] = -1;
        first[i] = 0xFFFFFFFFu;
    }
    for (int i = tid; i < n1; i += 256) {
        m12[i] = -1;
        qbin[i] = -1;
    }
    if (tid < HISTO_LENGTH) hist[tid] = 0;
    if (tid < 2) counters[tid] = 0;
    __syncthreads();
    if (tid < 64) {
        int nmatches = 0;
        for (int q0 = 0; q0 < n1; q0 += 64) {
            const int i = q0 + lane;
            const bool have = i < n1;
            const int b = have ? offs[i] : 0, e = have ? offs[i + 1] : 0;
            constexpr int RC = 4;
            u32 cw[RC];
#pragma unroll
            for (int j = 0; j < RC; j++) cw[j] = b + j < e ? sorted[b + j] : 0xFFFFFFFFu;
            u64 pending = __ballot(have && e > b);
            while (pending) {
                const bool mine = (pending >> lane) & 1ull;
                u32 bestW = 0xFFFFFFFFu, secondW = 0xFFFFFFFFu;
                int found = 0;
#pragma unroll
                for (int j = 0; j < RC; j++) {
                    const u32 c = cw[j];
                    const bool adm = mine && c != 0xFFFFFFFFu && mdist[c & 0xFFFFu] > (int)(c >> 16);
                    if (adm && found == 1) {
                        secondW = c;
                        found = 2;
                    }
                    if (adm && found == 0) {
                        bestW = c;
                        found = 1;
                    }
                }
                if (mine) {
                    for (int k = b + RC; k < e && found < 2; k++) {
                        const u32 c = sorted[k];
                        if (mdist[c & 0xFFFFu] > (int)(c >> 16)) {
                            if (found == 0) bestW = c;
                            else secondW = c;
                            found++;
                        }
                    }
                }
                const int bestDist = (int)(bestW >> 16);
                const float second = found == 2 ? (float)(int)(secondW >> 16) : (float)0x7FFFFFFF;
                const bool inReach = found >= 1 && bestDist <= TH_LOW;  // otherwise no later state can produce a match
                const bool wants = inReach && (float)bestDist < __fmul_rn(second, nnRatio);
                const int target = (int)(bestW & 0xFFFFu);
                if (wants) atomicMin(&first[target], (u32)lane);
                const u32 fa = inReach ? first[target] : 0xFFFFFFFFu;
                const u32 fb = (inReach && found == 2) ? first[secondW & 0xFFFFu] : 0xFFFFFFFFu;
                const bool dirty = mine && min(fa, fb) < (u32)lane;
                if (wants) first[target] = 0xFFFFFFFFu;
                const u64 dm = __ballot(dirty);
                const int d = dm ? __ffsll((unsigned long long)dm) - 1 : 64;
                const bool commit = mine && lane < d;
                bool replaced = false;
                if (commit && wants) {
                    const int prev = m21[target];
                    replaced = prev >= 0;
                    if (replaced) m12[prev] = -1;
                    m12[i] = target;
                    m21[target] = i;
                    mdist[target] = bestDist;
                    if (checkOri) {
                        const int bin = rot_bin(f1.kps[i].angle, f2.kps[target].angle);
                        qbin[i] = bin;
                        atomicAdd(&hist[bin], 1);
                    }
                }
                nmatches += __popcll(__ballot(commit && wants)) - __popcll(__ballot(replaced));
                pending &= ~__ballot(commit);
            }
        }
        if (lane == 0) counters[0] = nmatches;
    }
    __syncthreads();
    if (checkOri) {
        if (tid == 0) {
            int a = -1, b2 = -1, c = -1;
            three_maxima(hist, HISTO_LENGTH, a, b2, c);
            keep[0] = a;
            keep[1] = b2;
            keep[2] = c;
        }
        __syncthreads();
        int removed = 0;
        for (int i = tid; i < n1; i += 256) {
            const int bin = qbin[i];
            if (bin >= 0 && bin != keep[0] && bin != keep[1] && bin != keep[2] && m12[i] >= 0) {
                m12[i] = -1;
                removed++;
            }
        }
        if (removed) atomicAdd(&counters[1], removed);
        __syncthreads();
    }
    for (int i = tid; i < n1; i += 256) {
        const int mt = m12[i];
        matches12[i] = mt;
        if (mt >= 0) {
            prevMatched[2 * i] = f2.kps[mt].x;
            prevMatched[2 * i + 1] = f2.kps[mt].y;
        }
    }
    if (tid == 0) *nmatchesOut = counters[0] - counters[1];
}
__global__ __launch_bounds__(256) void k_init_resolve_sorted(DevFrame f1, DevFrame f2, const int* __restrict__ offs,
                                                              const u32* __restrict__ sorted, float nnRatio, int checkOri,
                                                              int* matches12, float* prevMatched, int* nmatchesOut) {
    extern __shared__ __align__(16) u8 ism[];
    init_resolve_sorted_body(ism, f1, f2, offs, sorted, nnRatio, checkOri, matches12, prevMatched, nmatchesOut);
}

// The same over a table of independent frame pairs: `last` is F1 (the queries), `train` is F2.
__global__ void k_init_queries_jobs(const ProjJob* __restrict__ jobs, float window) {
    const ProjJob& J = jobs[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= J.last.n) return;
    J.qValid[i] = J.last.kps[i].octave <= 0 ? 1 : 0;
    J.initU[i] = J.initPrev[2 * i];
    J.initV[i] = J.initPrev[2 * i + 1];
    J.qRadius[i] = window;
    J.qMin[i] = 0;
    J.qMax[i] = 0;
}
__global__ __launch_bounds__(256) void k_sort_rows_jobs(const ProjJob* __restrict__ jobs) {
    const ProjJob& J = jobs[blockIdx.y];
    sort_rows_body(J.offs, J.candIdx, J.candDist, J.slab, J.q.n);
}
__global__ __launch_bounds__(256) void k_init_resolve_sorted_jobs(const ProjJob* __restrict__ jobs, float nnRatio, int checkOri) {
    extern __shared__ __align__(16) u8 ism[];
    const ProjJob& J = jobs[blockIdx.x];
    init_resolve_sorted_body(ism, J.last, J.train, J.offs, J.slab, nnRatio, checkOri, J.qBest, J.initPrev, J.nmatches);
}
__global__ __launch_bounds__(64) void k_init_resolve_jobs(const ProjJob* __restrict__ jobs, float nnRatio, int checkOri) {
    const ProjJob& J = jobs[blockIdx.x];
    init_resolve_body(J.last, J.train, J.offs, J.candIdx, J.candDist, nnRatio, checkOri, J.claimed, J.initM21, J.qBest, J.qBin,
                      J.initPrev, J.nmatches);
}

// Queries of SearchForInitialization: level-0 keypoints of F1 search a window
// around their previous match in F2 at level 0 (reference :515-523).
__global__ void k_init_queries(DevFrame f1, const float* prevMatched, float window, float* qx, float* qy, float* qr,
                               int* qmin, int* qmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= f1.n) return;
    const bool use = f1.kps[i].octave <= 0;
    qx[i] = prevMatched[2 * i];
    qy[i] = prevMatched[2 * i + 1];
    // a non-positive radius accepts nothing (|d| < r fails), which skips the query
    qr[i] = use ? window : -1.f;
    qmin[i] = 0;
    qmax[i] = 0;
}

// ------------------------------------------------------------------ brute-force kNN-2
__device__ __forceinline__ u32 umed3(u32 a, u32 b, u32 c) {
    u32 r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

#define KNN_QPT 4          // queries per thread
#define KNN_IDX_BITS 22    // train rows per launch chunk: (distance << 22) | row fits 32 bits

// ------------------------------------------------------------------ vocabulary-guided searches
// DBoW2::FeatureVector as CSR (node ids ascending; items = keypoint indices in list order).
struct DevFeatVec {
    int nNodes;
    const u32* nodeIds;
    const int* start;
    const u32* items;
};

// Index of `node` in v.nodeIds or -1 (the std::map lookups of reference
// src/ORBmatcher.cc:234-327 reduce to an intersection of the two sorted id lists).
__device__ __forceinline__ int find_node(const DevFeatVec& v, u32 node) {
    int lo = 0, hi = v.nNodes - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const u32 id = v.nodeIds[mid];
        if (id == node) return mid;
        if (id < node) lo = mid + 1;
        else hi = mid - 1;
    }
    return -1;
}

__device__ __forceinline__ u32 wave_min_u32(u32 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (u32)__shfl_xor((int)v, o));
    return v;
}

// ORBmatcher::SearchByBoW, both overloads (reference src/ORBmatcher.cc:209-355, :642-788):
// one wave per vocabulary node present in both feature vectors.  Second-side keypoints
// belong to one node only, so the claims of different nodes never meet; inside a node the
// first-side keypoints are taken in list order and the second-side list is scanned by the
// 64 lanes (best = smallest distance, first in list order; second best counts duplicates).
__device__ __forceinline__ void bow_match_body(const int n1, DevFrame f1, DevFeatVec v1, const u8* __restrict__ valid1,
                                               DevFrame f2, DevFeatVec v2, const u8* __restrict__ valid2, int maxDist,
                                               float nnRatio, int checkOri, int* match12, int* match21, int* bin12, int* hist) {
    const int lane = threadIdx.x;
    const int n2 = find_node(v2, v1.nodeIds[n1]);
    if (n2 < 0) return;
    const int b2 = v2.start[n2], e2 = v2.start[n2 + 1];
    {
        // A node of ordinary size (at most BOW_CAP pairs): all distances of the node first, in parallel,
        // into LDS; the serial claims of the reference then run from LDS alone, and the matches are
        // written (and their rotation bins counted) in parallel at the end.
        constexpr int BOW_CAP = 4096;
        __shared__ u16 sD[BOW_CAP];      // distance of pair (a, k), 0xFFFF when either side is not eligible
        __shared__ u16 sIdx1[BOW_CAP], sIdx2[BOW_CAP], sMatch[BOW_CAP];
        __shared__ u8 sTaken[BOW_CAP];
        const int b1 = v1.start[n1], n1c = v1.start[n1 + 1] - b1, n2c = e2 - b2;
        if (n1c <= 0 || n2c <= 0) return;
        if ((i64)n1c * n2c <= BOW_CAP && f1.n <= 65535 && f2.n <= 65535) {
            for (int k = lane; k < n2c; k += 64) {
                sTaken[k] = 0;
                sIdx2[k] = (u16)v2.items[b2 + k];
            }
            for (int a = lane; a < n1c; a += 64) {
                sIdx1[a] = (u16)v1.items[b1 + a];
                sMatch[a] = 0xFFFF;
            }
            __syncthreads();
            const int pairs = n1c * n2c;
            for (int p = lane; p < pairs; p += 64) {
                const int a = p / n2c, k = p - a * n2c;
                const int idx1 = sIdx1[a], idx2 = sIdx2[k];
                u32 dist = 0xFFFFu;
                if (valid1[idx1] && (!valid2 || valid2[idx2])) {
                    const uint4* d1 = (const uint4*)(f1.desc + (i64)idx1 * 32);
                    const uint4* d2 = (const uint4*)(f2.desc + (i64)idx2 * 32);
                    dist = (u32)hamming256(d1[0], d1[1], d2[0], d2[1]);
                }
                sD[p] = (u16)dist;
            }
            __syncthreads();
            for (int a = 0; a < n1c; a++) {
                u32 bkey = 0xFFFFFFFFu, second = 256;
                for (int k = lane; k < n2c; k += 64) {
                    const u32 dist = sD[a * n2c + k];
                    if (sTaken[k] || dist == 0xFFFFu) continue;
                    const u32 key = (dist << 16) | (u32)k;
                    if (key < bkey) {
                        if (bkey != 0xFFFFFFFFu) second = min(second, bkey >> 16);
                        bkey = key;
                    } else {
                        second = min(second, dist);
                    }
                }
                const u32 win = wave_min_u32_dpp(bkey);
                const u32 mine = bkey == win ? second : (bkey == 0xFFFFFFFFu ? 256u : (bkey >> 16));
                const u32 bestDist2 = wave_min_u32_dpp(mine);
                if (win == 0xFFFFFFFFu) continue;
                const int bestDist1 = (int)(win >> 16);
                if (bestDist1 <= maxDist && (float)bestDist1 < __fmul_rn(nnRatio, (float)bestDist2)) {
                    const int bestK = (int)(win & 0xFFFFu);
                    const u16 taken2 = sIdx2[bestK];
                    for (int k = lane; k < n2c; k += 64)
                        if (sIdx2[k] == taken2) sTaken[k] = 1;
                    if (lane == 0) sMatch[a] = (u16)bestK;
                }
            }
            __syncthreads();
            for (int a = lane; a < n1c; a += 64) {
                const int mk = sMatch[a];
                if (mk == 0xFFFF) continue;
                const int idx1 = sIdx1[a], idx2 = sIdx2[mk];
                match12[idx1] = idx2;
                match21[idx2] = idx1;
                if (checkOri) {
                    const int bin = rot_bin(f1.kps[idx1].angle, f2.kps[idx2].angle);
                    bin12[idx1] = bin;
                    atomicAdd(&hist[bin], 1);
                }
            }
            return;
        }
    }
    for (int a = v1.start[n1]; a < v1.start[n1 + 1]; a++) {
        const int idx1 = (int)v1.items[a];
        if (!valid1[idx1]) continue;
        const uint4* d1 = (const uint4*)(f1.desc + (i64)idx1 * 32);
        const uint4 a0 = d1[0], a1 = d1[1];
        u64 bkey = ~0ull;
        u32 second = 256;
        for (int k = b2 + lane; k < e2; k += 64) {
            const int idx2 = (int)v2.items[k];
            if (match21[idx2] >= 0) continue;
            if (valid2 && !valid2[idx2]) continue;
            const uint4* d2 = (const uint4*)(f2.desc + (i64)idx2 * 32);
            const u32 dist = (u32)hamming256(a0, a1, d2[0], d2[1]);
            const u64 key = ((u64)dist << 32) | (u32)(k - b2);
            if (key < bkey) {
                if (bkey != ~0ull) second = min(second, (u32)(bkey >> 32));
                bkey = key;
            } else {
                second = min(second, dist);
            }
        }
        const u64 win = wave_min_u64(bkey);
        const u32 mine = bkey == win ? second : (bkey == ~0ull ? 256u : (u32)(bkey >> 32));
        const u32 bestDist2 = wave_min_u32(mine);
        if (win == ~0ull) continue;
        const int bestDist1 = (int)(win >> 32);
        if (bestDist1 <= maxDist && (float)bestDist1 < __fmul_rn(nnRatio, (float)bestDist2)) {
            const int bestIdx2 = (int)v2.items[b2 + (int)(u32)win];
            if (lane == 0) {
                match12[idx1] = bestIdx2;
                match21[bestIdx2] = idx1;
                if (checkOri) {
                    const int bin = rot_bin(f1.kps[idx1].angle, f2.kps[bestIdx2].angle);
                    bin12[idx1] = bin;
                    atomicAdd(&hist[bin], 1);
                }
            }
            __threadfence_block();
        }
    }
}

struct TriParams {
    float F12[9];
    float ex, ey;
    float scale2[ORBM_MAX_LEVELS], sigma2[ORBM_MAX_LEVELS];
    int onlyStereo, checkOri;
};

// ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:803-992), same node
// pairing.  The reference never marks second-side keypoints as taken, so every first-side
// keypoint is independent: among the second-side keypoints of the node that pass the
// distance, epipole and epipolar-line gates the smallest distance wins, the LAST one in list
// order on ties (:896 lets an equal distance replace the best).
__global__ __launch_bounds__(64) void k_tri_match(DevFrame f1, DevFeatVec v1, const u8* __restrict__ hasMp1,
                                                   const float* __restrict__ uRight1, DevFrame f2, DevFeatVec v2,
                                                   const u8* __restrict__ hasMp2, const float* __restrict__ uRight2,
                                                   TriParams P, int* match12, int* bin12, int* hist) {
    const int n1 = blockIdx.x, lane = threadIdx.x;
    const int n2 = find_node(v2, v1.nodeIds[n1]);
    if (n2 < 0) return;
    const int b2 = v2.start[n2], e2 = v2.start[n2 + 1];
    for (int a = v1.start[n1]; a < v1.start[n1 + 1]; a++) {
        const int idx1 = (int)v1.items[a];
        if (hasMp1[idx1]) continue;
        const bool bStereo1 = uRight1[idx1] >= 0;
        if (P.onlyStereo && !bStereo1) continue;
        const orbx_keypoint kp1 = f1.kps[idx1];
        const uint4* d1 = (const uint4*)(f1.desc + (i64)idx1 * 32);
        const uint4 a0 = d1[0], a1 = d1[1];
        // epipolar line of kp1 in the second image (reference src/ORBmatcher.cc:178-180)
        const float la = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, P.F12[0]), __fmul_rn(kp1.y, P.F12[3])), P.F12[6]);
        const float lb = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, P.F12[1]), __fmul_rn(kp1.y, P.F12[4])), P.F12[7]);
        const float lc = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, P.F12[2]), __fmul_rn(kp1.y, P.F12[5])), P.F12[8]);
        const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
        u64 bkey = ~0ull;
        for (int k = b2 + lane; k < e2; k += 64) {
            const int idx2 = (int)v2.items[k];
            if (hasMp2[idx2]) continue;
            const bool bStereo2 = uRight2[idx2] >= 0;
            if (P.onlyStereo && !bStereo2) continue;
            const uint4* d2 = (const uint4*)(f2.desc + (i64)idx2 * 32);
            const int dist = hamming256(a0, a1, d2[0], d2[1]);
            if (dist > TH_LOW) continue;
            const orbx_keypoint kp2 = f2.kps[idx2];
            const int oct = min(max(kp2.octave, 0), ORBM_MAX_LEVELS - 1);
            if (!bStereo1 && !bStereo2) {
                const float distex = __fsub_rn(P.ex, kp2.x), distey = __fsub_rn(P.ey, kp2.y);
                if (__fadd_rn(__fmul_rn(distex, distex), __fmul_rn(distey, distey)) < __fmul_rn(100.0f, P.scale2[oct])) continue;
            }
            const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, kp2.x), __fmul_rn(lb, kp2.y)), lc);
            if (den == 0) continue;
            const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
            if (!((double)dsqr < 3.84 * (double)P.sigma2[oct])) continue;
            const u64 key = ((u64)(u32)dist << 32) | (0xFFFFFFFFu - (u32)(k - b2));
            bkey = key < bkey ? key : bkey;
        }
        const u64 win = wave_min_u64(bkey);
        if (win == ~0ull) continue;
        const int bestIdx2 = (int)v2.items[b2 + (int)(0xFFFFFFFFu - (u32)win)];
        if (lane == 0) {
            match12[idx1] = bestIdx2;
            if (P.checkOri) {
                const int bin = rot_bin(kp1.angle, f2.kps[bestIdx2].angle);
                bin12[idx1] = bin;
                atomicAdd(&hist[bin], 1);
            }
        }
    }
}

// One vocabulary node of the first side per workgroup; a single pair, or one pair ("job") per blockIdx.y.
struct BowJob {
    DevFrame f1, f2;
    DevFeatVec v1, v2;
    const u8 *valid1, *valid2;
    int *match12, *match21, *bin12, *hist;  // hist: HISTO_LENGTH bins + the match count + one spare
};
__global__ __launch_bounds__(64) void k_bow_match(DevFrame f1, DevFeatVec v1, const u8* __restrict__ valid1, DevFrame f2,
                                                   DevFeatVec v2, const u8* __restrict__ valid2, int maxDist, float nnRatio,
                                                   int checkOri, int* match12, int* match21, int* bin12, int* hist) {
    bow_match_body((int)blockIdx.x, f1, v1, valid1, f2, v2, valid2, maxDist, nnRatio, checkOri, match12, match21, bin12, hist);
}
__global__ __launch_bounds__(64) void k_bow_match_jobs(const BowJob* __restrict__ jobs, int maxDist, float nnRatio, int checkOri) {
    const BowJob& J = jobs[blockIdx.y];
    if ((int)blockIdx.x >= J.v1.nNodes || J.v2.nNodes == 0 || J.f1.n == 0 || J.f2.n == 0) return;
    bow_match_body((int)blockIdx.x, J.f1, J.v1, J.valid1, J.f2, J.v2, J.valid2, maxDist, nnRatio, checkOri, J.match12, J.match21,
                   J.bin12, J.hist);
}
__global__ void k_bow_init_jobs(const BowJob* __restrict__ jobs) {
    const BowJob& J = jobs[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < J.f1.n) {
        J.match12[i] = -1;
        J.bin12[i] = -1;
    }
    if (i < J.f2.n) J.match21[i] = -1;
    if (i < HISTO_LENGTH + 2) J.hist[i] = 0;
}

// Rotation-consistency cull shared by the two kernels above (reference
// src/ORBmatcher.cc:330-352): keep the three fullest bins, count what is left.
__device__ __forceinline__ void bow_finish_body(int n1, int checkOri, int* match12, int* match21, const int* bin12,
                                                const int* hist, int* nmatchesOut) {
    __shared__ int keep[3];
    __shared__ int total;
    const int tid = threadIdx.x;
    if (tid == 0) {
        int i1 = -1, i2 = -1, i3 = -1;
        if (checkOri) three_maxima(hist, HISTO_LENGTH, i1, i2, i3);
        keep[0] = i1;
        keep[1] = i2;
        keep[2] = i3;
        total = 0;
    }
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < n1; i += 256) {
        const int m = match12[i];
        if (m < 0) continue;
        if (checkOri) {
            const int bin = bin12[i];
            if (bin != keep[0] && bin != keep[1] && bin != keep[2]) {
                if (match21) match21[m] = -1;
                match12[i] = -1;
                continue;
            }
        }
        mine++;
    }
    atomicAdd(&total, mine);
    __syncthreads();
    if (tid == 0) *nmatchesOut = total;
}
__global__ __launch_bounds__(256) void k_bow_finish(int n1, int checkOri, int* match12, int* match21, const int* bin12,
                                                     const int* hist, int* nmatchesOut) {
    bow_finish_body(n1, checkOri, match12, match21, bin12, hist, nmatchesOut);
}
__global__ __launch_bounds__(256) void k_bow_finish_jobs(const BowJob* __restrict__ jobs, int checkOri, int* counts) {
    const BowJob& J = jobs[blockIdx.x];
    bow_finish_body(J.f1.n, checkOri, J.match12, J.match21, J.bin12, J.hist, counts + blockIdx.x);
}

struct WindowBestParams {
    float invSigma2[ORBM_MAX_LEVELS];
    int chi2Gate;
};

// The candidate loop of Fuse and SearchBySim3 (reference src/ORBmatcher.cc:1066-1120,
// :1233-1262, :1391-1416, :1462-1489): queries are independent, one wave each.  The window is
// walked in GetFeaturesInArea order; keypoints one level below or at the predicted level
// qualify; with chi2Gate the reprojection test of Fuse (:1085-1111) is applied first.  The
// closest descriptor wins, the first one met on ties; 256 / -1 when nothing qualified.
__global__ __launch_bounds__(256) void k_window_best(DevFrame kf, const int* __restrict__ cellStart,
                                                      const int* __restrict__ cellItems, int n, const u8* __restrict__ valid,
                                                      const float* __restrict__ qu, const float* __restrict__ qv,
                                                      const float* __restrict__ qr, const int* __restrict__ qlevel,
                                                      const u8* __restrict__ qdesc, const float* __restrict__ kfURight,
                                                      const float* __restrict__ qURight, WindowBestParams P, int* bestIdx,
                                                      int* bestDist) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= n) return;
    const int lane = threadIdx.x & 63;
    int outIdx = -1, outDist = 256;
    if (valid[q]) {
        Query qq;
        qq.x = qu[q];
        qq.y = qv[q];
        qq.r = qr[q];
        const int level = qlevel[q];
        qq.minLevel = level - 1;
        qq.maxLevel = level;
        qq.useRight = false;
        qq.ur = qq.rGate = 0.f;
        const uint4* dq = (const uint4*)(qdesc + (i64)q * 32);
        const uint4 a0 = dq[0], a1 = dq[1];
        const float ur = (P.chi2Gate && qURight) ? qURight[q] : 0.f;
        u64 bkey = ~0ull;
        walk_window(kf, cellStart, cellItems, nullptr, qq, [&](int pos, int idx) {
            const orbx_keypoint kp = kf.kps[idx];
            if (P.chi2Gate) {
                const int oct = min(max(kp.octave, 0), ORBM_MAX_LEVELS - 1);
                const float ex = __fsub_rn(qq.x, kp.x), ey = __fsub_rn(qq.y, kp.y);
                const float kur = kfURight ? kfURight[idx] : -1.f;
                if (kur >= 0) {
                    const float er = __fsub_rn(ur, kur);
                    const float e2 = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(er, er));
                    if ((double)__fmul_rn(e2, P.invSigma2[oct]) > 7.8) return;
                } else {
                    const float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                    if ((double)__fmul_rn(e2, P.invSigma2[oct]) > 5.99) return;
                }
            }
            const uint4* dk = (const uint4*)(kf.desc + (i64)idx * 32);
            const u32 dist = (u32)hamming256(a0, a1, dk[0], dk[1]);
            // distance, then ordinal in walk order (first met wins ties), then the keypoint index
            // itself (frames hold fewer than 2^20 keypoints)
            const u64 key = ((u64)dist << 40) | ((u64)(u32)pos << 20) | (u32)idx;
            bkey = key < bkey ? key : bkey;
        });
        const u64 win = wave_min_u64(bkey);
        if (win != ~0ull) {
            outDist = (int)(win >> 40);
            outIdx = (int)(win & 0xFFFFFu);
        }
    }
    if (lane == 0) {
        bestIdx[q] = outIdx;
        bestDist[q] = outDist;
    }
}

// Agreement step of SearchBySim3 (reference src/ORBmatcher.cc:1499-1514).
__global__ __launch_bounds__(256) void k_sim3_agree(int n1, int n2, const int* __restrict__ idx12, const int* __restrict__ d12,
                                                     const int* __restrict__ idx21, const int* __restrict__ d21, int* match12,
                                                     int* nFound) {
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    int mine = 0;
    for (int i1 = threadIdx.x; i1 < n1; i1 += 256) {
        int m = -1;
        const int idx2 = d12[i1] <= TH_HIGH ? idx12[i1] : -1;
        if (idx2 >= 0 && idx2 < n2) {
            const int back = d21[idx2] <= TH_HIGH ? idx21[idx2] : -1;
            if (back == i1) m = idx2;
        }
        match12[i1] = m;
        mine += m >= 0;
    }
    atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) *nFound = total;
}

// ------------------------------------------------------------------ distinctive descriptors
// MapPoint::ComputeDistinctiveDescriptors for many map points (reference
// src/MapPoint.cc:307-340): one wave per map point.  Row i of the distance matrix is formed
// by the 64 lanes (kept in LDS as 16-bit values); its median, element (N-1)/2 of the sorted
// row, is the smallest value v with at least (N-1)/2 + 1 entries <= v, found by bisection
// over [0, 256] with wave ballots.  The first row with the smallest median wins.
#define DD_MAX_GROUP 4096

__global__ __launch_bounds__(256) void k_distinctive(const u8* __restrict__ desc, const int* __restrict__ groupStart,
                                                      int ngroups, int* best, int* bestMedian, u8* outDesc) {
    extern __shared__ u16 ddRows[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + wv;
    if (g >= ngroups) return;
    u16* row = ddRows + wv * DD_MAX_GROUP;
    const int b = groupStart[g], N = groupStart[g + 1] - b;
    int bestIdx = -1, bestMed = 0x7fffffff;
    if (N > 0 && N <= DD_MAX_GROUP) {
        const int k = (N - 1) >> 1;
        for (int i = 0; i < N; i++) {
            const uint4* di = (const uint4*)(desc + (i64)(b + i) * 32);
            const uint4 a0 = di[0], a1 = di[1];
            for (int j = lane; j < N; j += 64) {
                const uint4* dj = (const uint4*)(desc + (i64)(b + j) * 32);
                row[j] = (u16)hamming256(a0, a1, dj[0], dj[1]);
            }
            __builtin_amdgcn_wave_barrier();
            int lo = 0, hi = 256;  // smallest v with count(row <= v) >= k + 1
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                int cnt = 0;
                for (int j0 = 0; j0 < N; j0 += 64) {
                    const int j = j0 + lane;
                    cnt += __popcll(__ballot(j < N && (int)row[j] <= mid));
                }
                if (cnt >= k + 1) hi = mid;
                else lo = mid + 1;
            }
            if (lo < bestMed) {
                bestMed = lo;
                bestIdx = i;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane == 0) {
        best[g] = bestIdx;
        if (bestMedian) bestMedian[g] = bestMed;
    }
    if (outDesc && bestIdx >= 0 && lane < 8) ((u32*)(outDesc + (i64)g * 32))[lane] = ((const u32*)(desc + (i64)(b + bestIdx) * 32))[lane];
}

// ------------------------------------------------------------------ mask filter, map records
// Sum of the mask bytes (reference perfect/src/Frame.cc:357): the whole job of one launch.
__global__ __launch_bounds__(256) void k_mask_sum(const u8* __restrict__ mask, int w, int h, i64 stride, u64* sum) {
    u64 acc = 0;
    for (int y = blockIdx.x; y < h; y += gridDim.x) {
        const u8* row = mask + (i64)y * stride;
        for (int x = threadIdx.x; x < w; x += 256) acc += row[x];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(sum, acc);
}

// Stable compaction of the keypoints whose mask byte is 1 (reference
// perfect/src/Frame.cc:358-373); one workgroup, keypoints in index order.
__global__ __launch_bounds__(1024) void k_mask_filter(const orbx_keypoint* __restrict__ kps, const u8* __restrict__ desc, int n,
                                                       const u8* __restrict__ mask, int w, int h, i64 stride,
                                                       const u64* __restrict__ sum, orbx_keypoint* outKps, u8* outDesc,
                                                       int* nOut) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool active = (double)*sum > (double)h * (double)w * 0.65;  // h * w * 0.65 as the reference forms it
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        bool keep = false;
        orbx_keypoint kp;
        if (i < n) {
            kp = kps[i];
            if (active) {
                const int px = (int)kp.x, py = (int)kp.y;
                keep = px >= 0 && px < w && py >= 0 && py < h && mask[(i64)py * stride + px] == 1;
            } else {
                keep = true;
            }
        }
        int tot;
        const int off = wave_excl_count(keep, &tot);
        if (lane == 0) wsum[wv] = tot;
        __syncthreads();
        int pre = carry;
        for (int k = 0; k < wv; k++) pre += wsum[k];
        if (keep) {
            const int o = pre + off;
            outKps[o] = kp;
            const uint4* s = (const uint4*)(desc + (i64)i * 32);
            uint4* d = (uint4*)(outDesc + (i64)o * 32);
            d[0] = s[0];
            d[1] = s[1];
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int k = 0; k < 16; k++) t += wsum[k];
            carry += t;
        }
        __syncthreads();
    }
    if (tid == 0) *nOut = carry;
}

// 64-byte keypoint records of the binary map file (reference perfect/src/Map.cc:361-380,
// :169-187): 16 lanes move one record, 4 bytes each.
__global__ __launch_bounds__(256) void k_records(int pack, int n, orbx_keypoint* kps, u8* desc, u64* mpIndex, u8* records) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 i = t >> 4;
    const int wd = (int)(t & 15);
    if (i >= n) return;
    u32* rec = (u32*)(records + i * 64) + wd;
    u32* src;
    if (wd < 6) src = (u32*)(kps + i) + wd;                       // pt.x, pt.y, size, angle, response, octave
    else if (wd < 14) src = (u32*)(desc + i * 32) + (wd - 6);      // descriptor
    else src = mpIndex ? (u32*)(mpIndex + i) + (wd - 14) : nullptr;  // map point index
    if (pack) {
        *rec = src ? *src : 0xFFFFFFFFu;  // no index array: ULONG_MAX
    } else {
        if (src) *src = *rec;
        if (wd == 0) kps[i].class_id = -1;  // cv::KeyPoint's default
    }
}

// ------------------------------------------------------------------ stereo
// Frame::ComputeStereoMatches (reference src/Frame.cc:603-774) for rectified pairs.  As in the
// reference, every image row has a list of the right keypoints that may match a left keypoint on
// that row: right keypoint i is in the lists of rows floor(y - r) .. ceil(y + r), r =
// 2*scale[octave] (:618-630).  The lists are built per pair as a CSR table (counts, prefix sum,
// fill: the order inside a list is whatever the atomics give, which does not matter because the
// best candidate is the minimum of (distance, index) keys: the first minimum in index order, as in
// the reference's scan of a list that holds ascending indices).
const int MAX_LEVELS = ORBM_MAX_LEVELS;
const int STEREO_MAX_ROWS = 4096;

struct DevPyramid {
    const u8* ptr[MAX_LEVELS];
    int pitch[MAX_LEVELS], w[MAX_LEVELS], h[MAX_LEVELS];
};

struct StereoParams {
    float scale[MAX_LEVELS], invScale[MAX_LEVELS];
    int nlevels;
    float mbf, maxD;
};

// One rectified pair.  rows[i] = (first row, last row, octave, x) of right keypoint i (an empty range
// for an octave outside the pyramid); rowStart[nRows + 1] / rowItems: the per-row lists.
struct StereoJob {
    DevFrame left, right;
    DevPyramid pl, pr;
    int4* rows;
    int *rowStart, *rowItems;
    int* sad;
    float *uRight, *depth;
    int* kept;
    int nRows;
    int useLists;  // 0: few right keypoints, every left keypoint scans all of them (no lists are built)
};

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup of 1024 threads per pair: row ranges, counts per row, prefix sum, fill.
__device__ __forceinline__ void stereo_rows_body(const StereoJob& J, const StereoParams& sp) {
    __shared__ int cnt[STEREO_MAX_ROWS];
    __shared__ int part[32];
    const int tid = threadIdx.x, nRows = J.nRows, n = J.right.n;
    for (int r = tid; r < STEREO_MAX_ROWS; r += 1024) cnt[r] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const orbx_keypoint kp = J.right.kps[i];
        int4 o = make_int4(1, 0, kp.octave, __float_as_int(kp.x));  // empty range for an octave outside the pyramid
        if (kp.octave >= 0 && kp.octave < sp.nlevels && kp.y == kp.y) {
            const float r = 2.0f * sp.scale[kp.octave];
            // clamped like the reference's "yi >= 0 && yi < nRows"; the float clamp keeps the conversion defined
            o.y = (int)fminf(ceilf(kp.y + r), (float)(nRows - 1));
            o.x = (int)fmaxf(floorf(kp.y - r), 0.0f);
            if (J.useLists)
                for (int y = o.x; y <= o.y; y++) atomicAdd(&cnt[y], 1);
        }
        J.rows[i] = o;
    }
    if (!J.useLists) return;  // uniform over the workgroup
    __syncthreads();
    // exclusive prefix sum over the rows: four rows per thread, then the threads' totals
    const int r0 = 4 * tid;
    const int c0 = cnt[r0], c1 = cnt[r0 + 1], c2 = cnt[r0 + 2], c3 = cnt[r0 + 3];
    // inclusive scan of the threads' totals: inside each wave by shuffles, then over the 16 wave totals
    const int mine = c0 + c1 + c2 + c3;
    int inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(inc, d);
        if ((tid & 63) >= d) inc += v;
    }
    if ((tid & 63) == 63) part[tid >> 6] = inc;
    __syncthreads();
    if (tid < 16) {
        int w = part[tid];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const int v = __shfl_up(w, d, 16);
            if (tid >= d) w += v;
        }
        part[16 + tid] = w;  // inclusive totals of waves 0 .. tid
    }
    __syncthreads();
    const int waveBase = (tid >> 6) ? part[16 + (tid >> 6) - 1] : 0;
    const int base = waveBase + inc - mine;
    cnt[r0] = base;
    cnt[r0 + 1] = base + c0;
    cnt[r0 + 2] = base + c0 + c1;
    cnt[r0 + 3] = base + c0 + c1 + c2;
    __syncthreads();
    for (int r = tid; r < nRows; r += 1024) J.rowStart[r] = cnt[r];
    if (tid == 0) J.rowStart[nRows] = part[16 + 15];
    __syncthreads();
    // fill: cnt[row] is now the row's write cursor
    for (int i = tid; i < n; i += 1024) {
        const int4 o = J.rows[i];
        for (int y = o.x; y <= o.y; y++) J.rowItems[atomicAdd(&cnt[y], 1)] = i;
    }
}

__global__ __launch_bounds__(1024) void k_stereo_rows_jobs(const StereoJob* __restrict__ jobs, StereoParams sp) {
    stereo_rows_body(jobs[blockIdx.x], sp);
}
// (one pair, the record passed as a kernel argument: no upload in front of a call that is all latency)
__global__ __launch_bounds__(1024) void k_stereo_rows_one(StereoJob J, StereoParams sp) { stereo_rows_body(J, sp); }

// One wave per left keypoint: descriptor argmin over its row's list, then the 11x11 SAD
// search over 11 column shifts on the keypoint's pyramid level and the parabola fit.
// sad[iL] receives the best SAD (the key of the later median cut) or -1.
template <bool LISTS>
__device__ __forceinline__ void stereo_match_body(const StereoJob& J, const StereoParams& sp) {
    const DevFrame& left = J.left;
    const DevFrame& right = J.right;
    const DevPyramid& pl = J.pl;
    const DevPyramid& pr = J.pr;
    const int lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iL >= left.n) return;
    const orbx_keypoint kpL = left.kps[iL];
    const float uL = kpL.x, vL = kpL.y;
    const int levelL = kpL.octave;
    float outU = -1.0f, outD = -1.0f;
    int outS = -1;
    do {
        if (levelL < 0 || levelL >= sp.nlevels) break;
        if (!(vL >= 0.0f) || !(vL < (float)J.nRows)) break;
        const int row = (int)vL;
        const float minD = 0;
        const float minU = uL - sp.maxD;
        const float maxU = uL - minD;
        if (maxU < 0) break;
        const uint4* dl = (const uint4*)(left.desc + (size_t)iL * 32);
        const uint4 a0 = dl[0], a1 = dl[1];
        u64 best = ~0ull;
        // candidates: the row's list, or (a pair with few right keypoints, for which building the lists would
        // take longer than it saves) every right keypoint, tested against its row range
        const int4* __restrict__ rows = J.rows;
        const int* __restrict__ items = J.rowItems;
        const u8* __restrict__ rdesc = right.desc;
        const int b0 = LISTS ? J.rowStart[row] : 0, b1 = LISTS ? J.rowStart[row + 1] : right.n;
        for (int p = b0 + lane; p < b1; p += 64) {
            const int i = LISTS ? items[p] : p;
            const int4 rw = rows[i];
            const float uR = __int_as_float(rw.w);
            if ((LISTS || (rw.x <= row && row <= rw.y)) && rw.z >= levelL - 1 && rw.z <= levelL + 1 && uR >= minU && uR <= maxU) {
                const uint4* dr = (const uint4*)(rdesc + (size_t)i * 32);
                const int d = hamming256(a0, a1, dr[0], dr[1]);
                if (d < TH_HIGH) {  // bestDist starts at TH_HIGH, strict < (:650, :672)
                    const u64 key = ((u64)d << 32) | (u32)i;
                    best = key < best ? key : best;
                }
            }
        }
        best = wave_min_u64(best);
        if (best == ~0ull) break;
        const int bestDist = (int)(best >> 32);
        const int bestIdxR = (int)(u32)best;
        if (!(bestDist < (TH_HIGH + TH_LOW) / 2)) break;  // thOrbDist, :608

        // :684-690
        const float uR0 = right.kps[bestIdxR].x;
        const float scaleFactor = sp.invScale[levelL];
        const float scaleduL = roundf(uL * scaleFactor);
        const float scaledvL = roundf(vL * scaleFactor);
        const float scaleduR0 = roundf(uR0 * scaleFactor);
        const int w = 5, L = 5;
        const float iniu = scaleduR0 + L - w;
        const float endu = scaleduR0 + L + w + 1;
        if (iniu < 0 || endu >= pr.w[levelL]) break;  // :706-709
        // Every pixel the windows touch must exist (the reference's cv::Mat ranges throw otherwise).
        const int r0 = (int)(scaledvL - w), cl0 = (int)(scaleduL - w), cr0 = (int)(scaleduR0 - w);
        if (r0 < 0 || r0 + 2 * w >= pl.h[levelL] || r0 + 2 * w >= pr.h[levelL] || cl0 < 0 || cl0 + 2 * w >= pl.w[levelL] ||
            cr0 - L < 0 || cr0 + L + 2 * w >= pr.w[levelL])
            break;

        const int p0 = lane, p1 = lane + 64;
        const bool has1 = p1 < 121;
        const int y0 = p0 / 11, x0 = p0 - y0 * 11;
        const int y1 = has1 ? p1 / 11 : 0, x1 = has1 ? p1 - y1 * 11 : 0;
        const int pitchL = pl.pitch[levelL], pitchR = pr.pitch[levelL];
        const u8* Lp = pl.ptr[levelL] + (size_t)r0 * pitchL + cl0;
        const int cL = Lp[w * pitchL + w];
        const int il0 = (int)Lp[y0 * pitchL + x0] - cL;
        const int il1 = (int)Lp[y1 * pitchL + x1] - cL;
        int bestDistSAD = 0x7fffffff, bestincR = 0;
        int d1 = 0, d2 = 0, d3 = 0, prev = 0;
        bool wantNext = false;
        for (int incR = -L; incR <= L; incR++) {
            const u8* Rp = pr.ptr[levelL] + (size_t)r0 * pitchR + (cr0 + incR);
            const int cR = Rp[w * pitchR + w];
            int s = abs(il0 - ((int)Rp[y0 * pitchR + x0] - cR));
            if (has1) s += abs(il1 - ((int)Rp[y1 * pitchR + x1] - cR));
            s = wave_sum_i32(s);
            if (wantNext) {
                d3 = s;
                wantNext = false;
            }
            if (s < bestDistSAD) {  // :719-723
                bestDistSAD = s;
                bestincR = incR;
                d1 = prev;
                d2 = s;
                wantNext = true;
            }
            prev = s;
        }
        if (bestincR == -L || bestincR == L) break;  // :728-729
        // :732-737
        const float dist1 = (float)d1, dist2 = (float)d2, dist3 = (float)d3;
        const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
        if (deltaR < -1 || deltaR > 1) break;
        // :743-758
        float bestuR = sp.scale[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
        float disparity = (uL - bestuR);
        if (disparity >= minD && disparity < sp.maxD) {
            if (disparity <= 0) {
                disparity = 0.01;
                bestuR = uL - 0.01;
            }
            outD = sp.mbf / disparity;
            outU = bestuR;
            outS = bestDistSAD;
        }
    } while (0);
    if (lane == 0) {
        J.uRight[iL] = outU;
        J.depth[iL] = outD;
        J.sad[iL] = outS;
    }
}

__global__ __launch_bounds__(256) void k_stereo_match_jobs(const StereoJob* __restrict__ jobs, StereoParams sp) {
    stereo_match_body<true>(jobs[blockIdx.y], sp);
}
template <bool LISTS>
__global__ __launch_bounds__(256) void k_stereo_match_one(StereoJob J, StereoParams sp) {
    stereo_match_body<LISTS>(J, sp);
}

// :763-774: median of the accepted SAD distances (element size/2 of the sorted list),
// then every match with distance >= 1.5*1.4*median is dropped.  SAD <= 121*510, so the
// median comes from two LDS histogram passes (high bits, then low bits).
__device__ __forceinline__ void stereo_median_body(const StereoJob& J) {
    const int n = J.left.n;
    const int* __restrict__ sad = J.sad;
    float* uRight = J.uRight;
    float* depth = J.depth;
    int* kept = J.kept;
    __shared__ int hist[1024];
    __shared__ int sh[4];
    const int tid = threadIdx.x;
    hist[tid] = 0;
    if (tid < 4) sh[tid] = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = tid; i < n; i += 1024) {
        const int d = sad[i];
        if (d >= 0) {
            atomicAdd(&hist[min(d >> 6, 1023)], 1);
            cnt++;
        }
    }
    atomicAdd(&sh[0], cnt);
    __syncthreads();
    const int total = sh[0];
    if (total == 0) {
        if (tid == 0 && kept) *kept = 0;
        return;
    }
    const int k = total / 2;
    if (tid == 0) {
        int acc = 0, b = 0;
        for (; b < 1024; b++) {
            if (acc + hist[b] > k) break;
            acc += hist[b];
        }
        sh[1] = b;
        sh[2] = k - acc;
    }
    __syncthreads();
    const int hb = sh[1], k2 = sh[2];
    __syncthreads();
    hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int d = sad[i];
        if (d >= 0 && min(d >> 6, 1023) == hb) atomicAdd(&hist[hb == 1023 ? min(d - (1023 << 6), 1023) : (d & 63)], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0, b = 0;
        for (; b < 1024; b++) {
            if (acc + hist[b] > k2) break;
            acc += hist[b];
        }
        sh[1] = (hb << 6) + b;
        sh[3] = 0;
    }
    __syncthreads();
    const float median = sh[1];
    const float thDist = 1.5f * 1.4f * median;
    int mine = 0;
    for (int i = tid; i < n; i += 1024) {
        const int d = sad[i];
        if (d < 0) continue;
        if (d < thDist)
            mine++;
        else {
            uRight[i] = -1;
            depth[i] = -1;
        }
    }
    if (kept) {
        atomicAdd(&sh[3], mine);
        __syncthreads();
        if (tid == 0) *kept = sh[3];
    }
}
__global__ __launch_bounds__(1024) void k_stereo_median_jobs(const StereoJob* __restrict__ jobs) { stereo_median_body(jobs[blockIdx.x]); }
__global__ __launch_bounds__(1024) void k_stereo_median_one(StereoJob J) { stereo_median_body(J); }

// Each thread keeps KNN_QPT query descriptors in registers.  Every thread of the grid walks
// the same train rows in the same order, so a train descriptor is a wave-uniform value: it is
// fetched with scalar loads and used as the scalar operand of XOR (no LDS, no barriers).  The
// two smallest (distance, row) keys are kept with  second = med3(best, second, key);
// best = min(best, key): unique keys make "first row wins ties" and "second counts duplicates
// of the minimum" fall out of the ordering.
__global__ __launch_bounds__(256) void k_knn2(const uint4* __restrict__ q, int nq, const uint4* __restrict__ t, int nt,
                                               int indexBase, int* bestIdx, int* bestD, int* secondD, int mergeWithOutput,
                                               int sliceRows) {
    const int tid = threadIdx.x;
    const int q0 = (blockIdx.x * 256 + tid) * KNN_QPT;
    if (sliceRows > 0) {
        // few queries: blockIdx.y takes a slice of the train rows and writes its own partial results
        const int r0 = (int)blockIdx.y * sliceRows;
        t += (i64)r0 * 2;
        nt = min(sliceRows, nt - r0);
        indexBase += r0;
        bestIdx += (i64)blockIdx.y * nq;
        bestD += (i64)blockIdx.y * nq;
        secondD += (i64)blockIdx.y * nq;
    }
    uint4 qa[KNN_QPT][2];
    u32 best[KNN_QPT], second[KNN_QPT];
#pragma unroll
    for (int k = 0; k < KNN_QPT; k++) {
        const int qi = min(q0 + k, nq - 1);
        qa[k][0] = q[(i64)qi * 2];
        qa[k][1] = q[(i64)qi * 2 + 1];
        best[k] = 0xFFFFFFFFu;
        second[k] = 0xFFFFFFFFu;
    }
    auto visit = [&](const uint4 b0, const uint4 b1, int j) {
#pragma unroll
        for (int k = 0; k < KNN_QPT; k++) {
            const u32 d = (u32)hamming256(qa[k][0], qa[k][1], b0, b1);
            const u32 key = (d << KNN_IDX_BITS) | (u32)j;
            second[k] = umed3(best[k], second[k], key);
            best[k] = min(best[k], key);
        }
    };
    int j = 0;
    // Four train rows per step.  The top-2 update is skipped for a step unless some lane of the wave met
    // a distance that can enter its top 2 (distance <= current second-best distance), which soon
    // becomes rare.  Two sets of scalar registers alternate: the rows of the next step are requested
    // before the current step's math, so a wave does not sit out the scalar-load latency every step.
    auto step = [&](const uint4* b, int j0) {
        u32 d[KNN_QPT][4];
        bool any = false;
#pragma unroll
        for (int k = 0; k < KNN_QPT; k++) {
#pragma unroll
            for (int r = 0; r < 4; r++) d[k][r] = (u32)hamming256(qa[k][0], qa[k][1], b[2 * r], b[2 * r + 1]);
            const u32 m = min(min(d[k][0], d[k][1]), min(d[k][2], d[k][3]));
            any |= m <= (second[k] >> KNN_IDX_BITS);
        }
        if (__any(any)) {
#pragma unroll
            for (int k = 0; k < KNN_QPT; k++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const u32 key = (d[k][r] << KNN_IDX_BITS) | (u32)(j0 + r);
                    second[k] = umed3(best[k], second[k], key);
                    best[k] = min(best[k], key);
                }
            }
        }
    };
    auto load4 = [&](uint4* b, int j0) {
        // rows past the end are never used; the index is clamped so that the request stays in range
        const int jc = max(min(j0, nt - 4), 0);
#pragma unroll
        for (int r = 0; r < 8; r++) b[r] = t[(i64)jc * 2 + r];
    };
    if (nt >= 4) {
        uint4 ba[8], bb[8];
        load4(ba, 0);
        for (; j + 8 <= nt; j += 8) {
            // Scalar loads return out of order, so they can only be waited for all at once: the set about
            // to be used is touched (which places that wait) BEFORE the other set is requested, and the
            // fences keep the request ahead of the math.
            asm volatile("" ::"s"(ba[0].x));
            __builtin_amdgcn_sched_barrier(0);
            load4(bb, j + 4);
            __builtin_amdgcn_sched_barrier(0);
            step(ba, j);
            asm volatile("" ::"s"(bb[0].x));
            __builtin_amdgcn_sched_barrier(0);
            load4(ba, j + 8);
            __builtin_amdgcn_sched_barrier(0);
            step(bb, j + 4);
        }
        if (j + 4 <= nt) {
            step(ba, j);
            j += 4;
        }
    }
    for (; j < nt; j++) visit(t[(i64)j * 2], t[(i64)j * 2 + 1], j);
#pragma unroll
    for (int k = 0; k < KNN_QPT; k++) {
        if (q0 + k < nq) {
            // a distance of 256 never replaces the initial 256 (strict '<' from 256)
            int bd = best[k] == 0xFFFFFFFFu ? 256 : (int)(best[k] >> KNN_IDX_BITS);
            int bi = (bd >= 256) ? -1 : (int)(best[k] & ((1u << KNN_IDX_BITS) - 1)) + indexBase;
            int sd = second[k] == 0xFFFFFFFFu ? 256 : (int)(second[k] >> KNN_IDX_BITS);
            bd = min(bd, 256);
            sd = min(sd, 256);
            if (mergeWithOutput) {
                // earlier chunk of the same train set is already in the output arrays
                const int pb = bestD[q0 + k], ps = secondD[q0 + k], pi = bestIdx[q0 + k];
                const int s2 = min(max(pb, bd), min(ps, sd));
                if (!(bd < pb)) {
                    bd = pb;
                    bi = pi;
                }
                sd = s2;
            }
            bestIdx[q0 + k] = bi;
            bestD[q0 + k] = bd;
            secondD[q0 + k] = sd;
        }
    }
}


// ---- brute-force 2-NN on the matrix cores
// With a train descriptor a as 256 values 0 / 1 and a query q as 256 values +1 (bit clear) / -1
// (bit set), hamming(a, q) = popcount(q) + <a, q>: a contraction of small integers, exact in the
// block-scaled fp4 form of the matrix instruction (v_mfma_scale_f32_32x32x64_f8f6f4 with e2m1
// operands 0, 1.0, -1.0, unit scales and f32 accumulation; four steps of K = 64 per pair of
// descriptors, twice the rate of the int8 form and half its operand bytes).  A wave keeps KNN_MQ
// tiles of 32 queries expanded in registers (B operand: the query is the column, i.e. the lane).
// The four waves of a block share the train rows: stages of KNN_MROWS rows are expanded once per
// block through a 256-entry table (a byte -> eight nibbles) into a double-buffered LDS tile, from
// which every wave reads its A operands.  A 32 x 32 result tile leaves 16 train rows of one query
// in a lane, so the two smallest (distance, row) keys are kept per lane and query tile as in
// k_knn2 and the two lanes that share a query are folded at the end.  The position of a bit inside
// the contraction is arbitrary: both operands use the same one (descriptor byte 8 s + 4 h + i / 8,
// bit i % 8 at step s, lane half h, element i).
#define KNN_MQ 4
#define KNN_MROWS 64
#define KNN_MPITCH 144  // bytes per expanded row: 128 + 16, so that 16-byte reads down 16 rows touch every bank once
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// the eight nibbles (e2m1 1.0 = 0x2 where the bit is set) of each of the four bytes of w
__device__ __forceinline__ uint4 knn_nibbles(const u32* lut, const u32 w) {
    return make_uint4(lut[w & 0xFFu], lut[(w >> 8) & 0xFFu], lut[(w >> 16) & 0xFFu], lut[w >> 24]);
}

// The accumulators start at 2^23 + 256, so the f32 result 2^23 + 256 + <a, q> has the bit pattern
// KNN_ACC0 + <a, q>: an unsigned integer that orders like the distance and whose low bits, shifted,
// give the key without a conversion.
#define KNN_ACC0 0x4B000100u
__device__ __forceinline__ u32 f32_bits(float x) { return __float_as_uint(x); }
__device__ __forceinline__ u32 umin3(u32 a, u32 b, u32 c) { return min(min(a, b), c); }

template <bool TAIL>
__device__ __forceinline__ void knn_fold16(const v16f acc, const int pq, const int row0, const int nt, u32& best, u32& second,
                                           u32& thr) {
    // key = distance << KNN_IDX_BITS | row, distance = popcount(query) + <a, q>; (bits << KNN_IDX_BITS) keeps
    // only 256 + <a, q> of the pattern, so the constant carries popcount - 256
    const u32 k0 = ((u32)(pq - 256) << KNN_IDX_BITS) + (u32)row0;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int off = (r & 3) + 8 * (r >> 2);
        u32 key = (f32_bits(acc[r]) << KNN_IDX_BITS) + (k0 + (u32)off);
        if (TAIL) key = row0 + off < nt ? key : 0xFFFFFFFFu;
        second = umed3(best, second, key);
        best = min(best, key);
    }
    thr = KNN_ACC0 + (second >> KNN_IDX_BITS) - (u32)pq;
}

__global__ __launch_bounds__(256) void k_knn2_mfma(const uint4* __restrict__ q, int nq, const uint4* __restrict__ t, int nt,
                                                    int indexBase, int* bestIdx, int* bestD, int* secondD, int mergeWithOutput,
                                                    int sliceRows) {
    __shared__ u32 lut[256];
    __shared__ __align__(16) u8 tile[2][KNN_MROWS * KNN_MPITCH];
    if (sliceRows > 0) {
        // few queries: blockIdx.y takes a slice of the train rows and writes its own partial results
        const int r0 = (int)blockIdx.y * sliceRows;
        t += (i64)r0 * 2;
        nt = min(sliceRows, nt - r0);
        indexBase += r0;
        bestIdx += (i64)blockIdx.y * nq;
        bestD += (i64)blockIdx.y * nq;
        secondD += (i64)blockIdx.y * nq;
    }
    {
        const u32 i = threadIdx.x;
        u32 e = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) e |= ((i >> j) & 1u) << (4 * j + 1);
        lut[i] = e;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int q0 = (blockIdx.x * 4 + wv) * (32 * KNN_MQ);
    // waves past the last query keep staging train rows and meeting the barriers; they hold a copy of the last query
    v4i Bq[KNN_MQ][4];
    int pq[KNN_MQ];
#pragma unroll
    for (int m = 0; m < KNN_MQ; m++) {
        const int qi = min(q0 + 32 * m + c, nq - 1);
        const uint4 a = q[(i64)qi * 2], b = q[(i64)qi * 2 + 1];
        const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        pq[m] = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) pq[m] += __popc(w[k]);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            // +1.0 (0x2) where the bit is clear, -1.0 (0xA) where it is set
            const uint4 e = knn_nibbles(lut, w[2 * s] ^ ((w[2 * s] ^ w[2 * s + 1]) & (0u - (u32)h)));
            Bq[m][s] = (v4i){(int)(0x22222222u | (e.x << 2)), (int)(0x22222222u | (e.y << 2)), (int)(0x22222222u | (e.z << 2)),
                             (int)(0x22222222u | (e.w << 2))};
        }
    }
    u32 best[KNN_MQ], second[KNN_MQ], thr[KNN_MQ];
#pragma unroll
    for (int m = 0; m < KNN_MQ; m++) {
        best[m] = second[m] = 0xFFFFFFFFu;
        thr[m] = 0xFFFFFFFFu;  // a row can change a lane's two smallest only if its pattern is below this
    }
    // staging role of this thread: row sr of the stage, descriptor bytes 8 sp .. 8 sp + 7
    const int sr = threadIdx.x >> 2, sp = threadIdx.x & 3;
    const uint2* t2 = (const uint2*)t;
    const int nStages = (nt + KNN_MROWS - 1) / KNN_MROWS;
    u8* const myDst = &tile[0][0] + sr * KNN_MPITCH + sp * 32;
    const u8* const mySrc = &tile[0][0] + c * KNN_MPITCH + h * 16;
    auto stage = [&](u8* dst, const uint2 raw) {
        *(uint4*)dst = knn_nibbles(lut, raw.x);
        *(uint4*)(dst + 16) = knn_nibbles(lut, raw.y);
    };
    stage(myDst, t2[(i64)min(sr, nt - 1) * 4 + sp]);
    uint2 pre = t2[(i64)min(KNN_MROWS + sr, nt - 1) * 4 + sp];
    __syncthreads();
    const int unit = 0x7F7F7F7F;  // block scales 2^0
    const float c0 = __uint_as_float(KNN_ACC0);
    // sixteen registers that hold the constant for the whole scan (opaque to the compiler, which otherwise
    // rebuilds the tuple with sixteen moves per tile)
    v16f init = {c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0, c0};
    asm volatile("" : "+v"(init));
    constexpr int NT = (KNN_MROWS / 32) * KNN_MQ;  // result tiles per stage and wave
    for (int st = 0; st < nStages; st++) {
        const int buf = (st & 1) * (KNN_MROWS * KNN_MPITCH);
        // the next stage goes into the other buffer (last read before the barrier that ended the previous stage)
        if (st + 1 < nStages) stage(myDst + ((KNN_MROWS * KNN_MPITCH) - buf), pre);
        pre = t2[(i64)min((st + 2) * KNN_MROWS + sr, nt - 1) * 4 + sp];
        const bool tail = (st + 1) * KNN_MROWS > nt;
        v4i At[KNN_MROWS / 32][4];
#pragma unroll
        for (int sub = 0; sub < KNN_MROWS / 32; sub++)
#pragma unroll
            for (int s = 0; s < 4; s++) At[sub][s] = *(const v4i*)(mySrc + buf + sub * 32 * KNN_MPITCH + s * 32);
        auto step = [&](const int j, const int s, const v16f cin) {
            const v4i a = At[j / KNN_MQ][s], b = Bq[j % KNN_MQ][s];
            const v8i a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
            const v8i b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
            return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, cin, 4, 4, 0, unit, 0, unit);
        };
        // The four matrix instructions of tile j + 1 are issued between the pieces of the minimum over
        // tile j's sixteen results, so the vector ALU works in the shadow of the matrix pipe.
        v16f cur = step(0, 0, init);
#pragma unroll
        for (int s = 1; s < 4; s++) cur = step(0, s, cur);
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int m = j % KNN_MQ;
            v16f nxt = init;
            u32 mn;
            if (j + 1 < NT) nxt = step(j + 1, 0, init);
            mn = umin3(f32_bits(cur[0]), f32_bits(cur[1]), f32_bits(cur[2]));
            mn = umin3(mn, f32_bits(cur[3]), f32_bits(cur[4]));
            if (j + 1 < NT) nxt = step(j + 1, 1, nxt);
            mn = umin3(mn, f32_bits(cur[5]), f32_bits(cur[6]));
            mn = umin3(mn, f32_bits(cur[7]), f32_bits(cur[8]));
            if (j + 1 < NT) nxt = step(j + 1, 2, nxt);
            mn = umin3(mn, f32_bits(cur[9]), f32_bits(cur[10]));
            mn = umin3(mn, f32_bits(cur[11]), f32_bits(cur[12]));
            if (j + 1 < NT) nxt = step(j + 1, 3, nxt);
            mn = umin3(mn, f32_bits(cur[13]), f32_bits(cur[14]));
            mn = min(mn, f32_bits(cur[15]));
            if (j + 1 < NT) {
                // keeps tile j + 1's instructions on this side of the branch, one per two minimum steps
                asm volatile("" : "+v"(nxt));
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
            }
            if (__any(mn < thr[m])) {
                // result register r of a lane is train row (r & 3) + 8 (r >> 2) + 4 h of the sub-tile
                const int row0 = st * KNN_MROWS + (j / KNN_MQ) * 32 + 4 * h;
                if (tail)
                    knn_fold16<true>(cur, pq[m], row0, nt, best[m], second[m], thr[m]);
                else
                    knn_fold16<false>(cur, pq[m], row0, nt, best[m], second[m], thr[m]);
            }
            cur = nxt;
        }
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < KNN_MQ; m++) {
        // fold the two lanes of a query: smallest and second smallest key of the union
        const u32 ob = (u32)__shfl_xor((int)best[m], 32), os = (u32)__shfl_xor((int)second[m], 32);
        const u32 b = min(best[m], ob);
        const u32 s2 = min(max(best[m], ob), min(second[m], os));
        const int qi = q0 + 32 * m + c;
        if (h == 0 && qi < nq) {
            int bd = b == 0xFFFFFFFFu ? 256 : (int)(b >> KNN_IDX_BITS);
            int bi = (bd >= 256) ? -1 : (int)(b & ((1u << KNN_IDX_BITS) - 1)) + indexBase;
            int sd = s2 == 0xFFFFFFFFu ? 256 : (int)(s2 >> KNN_IDX_BITS);
            bd = min(bd, 256);
            sd = min(sd, 256);
            if (mergeWithOutput) {
                const int pb = bestD[qi], ps = secondD[qi], pi = bestIdx[qi];
                const int s3 = min(max(pb, bd), min(ps, sd));
                if (!(bd < pb)) {
                    bd = pb;
                    bi = pi;
                }
                sd = s3;
            }
            bestIdx[qi] = bi;
            bestD[qi] = bd;
            secondD[qi] = sd;
        }
    }
}

// Serial scan over shard A then shard B, reproduced from the two partials:
// best = min (A wins ties: lower global index); second = second smallest of
// the union multiset = min(max(bestA, bestB), secondA, secondB).
__global__ void k_knn2_merge(int nq, int* ia, int* da, int* sa, const int* ib, const int* db, const int* sb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int bA = da[i], bB = db[i];
    const int s = min(max(bA, bB), min(sa[i], sb[i]));
    if (bB < bA) {
        da[i] = bB;
        ia[i] = ib[i];
    }
    sa[i] = s;
}

// DBoW2 TemplatedVocabulary::transform for one feature (the published algorithm behind the call at
// reference src/Frame.cc:513-522): from the root, go to the child whose descriptor is closest in
// Hamming distance (the first child on ties, strict '<'), until a node without children; report its
// word id, its weight and the node passed at level nidLevel (the root, 0, when nidLevel <= 0).
// Sixteen lanes per feature: each takes children c, c + 16, ... of the current node.
struct DevVocabulary {
    int nNodes;
    const u8* nodeDesc;
    const int* childStart;
    const int* children;
    const int* wordId;
    const double* weight;
};
__global__ __launch_bounds__(256) void k_bow_transform(DevVocabulary V, const u8* __restrict__ desc, int n, int nidLevel,
                                                        int* __restrict__ wordOut, double* __restrict__ weightOut,
                                                        int* __restrict__ nodeOut) {
    const int lane16 = threadIdx.x & 15;
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool live = i < n;
    const uint4* fd = (const uint4*)(desc + (i64)(live ? i : 0) * 32);
    const uint4 f0 = fd[0], f1 = fd[1];
    int node = 0, nid = 0, level = 0;
    // every lane of a group follows the same path; groups of a wave may differ in depth, so the loop
    // runs while any group is still descending and finished groups idle (bounded by the tree depth)
    bool going = live;
    for (int guard = 0; guard < 64; guard++) {
        const int b = going ? V.childStart[node] : 0, e = going ? V.childStart[node + 1] : 0;
        if (e <= b) going = false;  // a leaf
        if (!__any(going)) break;
        u32 best = 0xFFFFFFFFu;     // distance << 16 | position in the children list
        if (going) {
            for (int c = b + lane16; c < e; c += 16) {
                const uint4* nd = (const uint4*)(V.nodeDesc + (i64)V.children[c] * 32);
                const u32 d = (u32)hamming256(f0, f1, nd[0], nd[1]);
                best = min(best, (d << 16) | (u32)(c - b));
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) best = min(best, (u32)__shfl_xor((int)best, o, 16));
        if (going) {
            node = V.children[b + (int)(best & 0xFFFFu)];
            level++;
            if (level == nidLevel) nid = node;
        }
    }
    if (live && lane16 == 0) {
        wordOut[i] = V.wordId[node];
        weightOut[i] = V.weight[node];
        nodeOut[i] = nid;
    }
}

// Acceptance rule of the reference's descriptor loops (src/ORBmatcher.cc:284-288): best <= th and
// (float)best < ratio * (float)second.
__global__ void k_ratio_test(int nq, const int* __restrict__ bd, const int* __restrict__ sd, float ratio, int th, u8* accept) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int b = bd[i];
    accept[i] = (b <= th && (float)b < __fmul_rn(ratio, (float)sd[i])) ? 1 : 0;
}

// The partials of consecutive train slices (slice-major arrays of nq entries) folded in slice order by the same rule.
__global__ void k_knn2_merge_slices(int nq, int nSlices, const int* pi, const int* pd, const int* ps, int* oi, int* od, int* os) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int bi = pi[i], bd = pd[i], sd = ps[i];
    for (int s = 1; s < nSlices; s++) {
        const int ib = pi[(i64)s * nq + i], db = pd[(i64)s * nq + i], sb = ps[(i64)s * nq + i];
        sd = min(max(bd, db), min(sd, sb));
        if (db < bd) {
            bd = db;
            bi = ib;
        }
    }
    oi[i] = bi;
    od[i] = bd;
    os[i] = sd;
}

}  // namespace

// ------------------------------------------------------------------ no-allocation 2-NN scratch
// Partial results of orbm_knn2_noalloc: three arrays (best index, best distance, second
// distance) of KNN_NA_MAX_PART entries each, one entry per (train slice, query).  12 MiB that
// every GPU the library runs on holds until the process ends.  The array's name is only a host-side
// stand-in; host code uses the per-device address fetched with hipGetSymbolAddress.
#define KNN_NA_MAX_PART (1 << 20)
static __device__ __align__(256) int scratch[3 * KNN_NA_MAX_PART];

namespace {
const int MAX_DEVICES = 64;
std::mutex g_scratchMu;
int* g_scratchPtr[MAX_DEVICES] = {nullptr};  // device address of `scratch` on each GPU
// Calls that share a GPU's array are ordered on the device: each waits for the event the previous one
// recorded behind its merge kernel (created with the address, so that the entry point itself creates nothing).
hipEvent_t g_scratchEv[MAX_DEVICES] = {nullptr};
bool g_scratchEvRecorded[MAX_DEVICES] = {false};

// With `device` current: the address of that GPU's copy of `scratch`, looked up once.
int scratch_address(int device, int** out) {
    if (device < 0 || device >= MAX_DEVICES) {
        orbx_set_error("device ordinal %d out of range", device);
        return ORBX_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lock(g_scratchMu);
    if (!g_scratchPtr[device]) {
        void* p = nullptr;
        ORBX_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) {
            orbx_set_error("hipGetSymbolAddress returned a null address");
            return ORBX_ERR_HIP;
        }
        ORBX_HIP_CHECK(hipEventCreateWithFlags(&g_scratchEv[device], hipEventDisableTiming));
        g_scratchPtr[device] = (int*)p;
    }
    *out = g_scratchPtr[device];
    return ORBX_OK;
}
}  // namespace

// ------------------------------------------------------------------ host
struct orbm_context {
    int device = 0;
    hipStream_t ownStream = nullptr, stream = nullptr;
    // grow-only scratch, one block per purpose
    std::vector<std::pair<void*, size_t>> pool;
    std::vector<int> totalsHost;
    bool resolveLdsReady = false, resolveJobsLdsReady = false, initSortedReady = false, initSortedJobsReady = false, distinctiveReady = false;
    int knnPath = 0;  // 0 = by size, 1 = vector ALU, 2 = matrix cores (orbm_debug_knn_path)  // the LDS form of the sequential resolve has had its dynamic LDS limit raised
    // Host arrays travel through one pinned arena per call: copies from and to pageable memory are staged
    // and serialised by the runtime (10-15 us each), copies from pinned memory are plain queue entries.
    // Results land in the arena and are handed to the caller's arrays after the stream has been waited for.
    unsigned char* arena = nullptr;
    size_t arenaCap = 0, arenaUsed = 0;
    std::vector<void*> arenaRetired;
    // Device mirror of the arena: an input array staged at arena offset o is read by the kernels at
    // dArena + o, and everything staged since the last launch goes up in ONE copy just before the next
    // launch (a call with a dozen input arrays pays one queue entry instead of a dozen).
    unsigned char* dArena = nullptr;
    std::vector<void*> dArenaRetired;
    size_t upLo = 0, upHi = 0;          // staged and not uploaded yet: [upLo, upHi) of the current arena
    // Result arrays live in the mirror too: the pieces asked back are fetched at the next wait, neighbours in one copy.
    std::vector<std::pair<size_t, size_t>> downloads;  // (offset, bytes) in the current arena
    hipError_t asyncError = hipSuccess;  // first failure of a deferred upload, reported by the next wait
    int* naScratch = nullptr;  // this device's copy of the static scratch array (orbm_knn2_noalloc)
    struct Delivery {
        void* dst;
        const void* src;
        size_t bytes;
    };
    std::vector<Delivery> pending;
};

namespace {

hipError_t sync_deliver(orbm_context* m);
void flush_uploads(orbm_context* m);
void flush_downloads(orbm_context* m);
// every kernel launch of the host side goes through this: staged inputs are uploaded first
#define ORBM_LAUNCH(...)                 \
    do {                                 \
        flush_uploads(m);                \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

int pool_get(orbm_context* m, int slot, size_t bytes, void** out) {
    if ((int)m->pool.size() <= slot) m->pool.resize(slot + 1, std::make_pair((void*)nullptr, (size_t)0));
    std::pair<void*, size_t>& e = m->pool[slot];
    if (e.second < bytes) {
        ORBX_HIP_CHECK(sync_deliver(m));
        if (e.first) (void)hipFree(e.first);
        e.first = nullptr;
        e.second = 0;
        const size_t want = std::max(bytes, (size_t)256) * 3 / 2;
        ORBX_HIP_CHECK(hipMalloc(&e.first, want));
        e.second = want;
    }
    *out = e.first;
    return ORBX_OK;
}

enum {
    SL_F1_KPS = 0, SL_F1_DESC, SL_F2_KPS, SL_F2_DESC, SL_CELL_START, SL_CELL_ITEMS, SL_CELL_OF, SL_QX, SL_QY,
    SL_QR, SL_QMIN, SL_QMAX, SL_QDESC, SL_COUNTS, SL_OFFS, SL_CAND_IDX, SL_CAND_DIST, SL_STATE_A, SL_STATE_B,
    SL_STATE_C, SL_STATE_D, SL_PREV, SL_SCALAR, SL_VALID, SL_U, SL_V, SL_INVZ, SL_HASOBS, SL_URIGHT, SL_SF,
    SL_KNN_Q, SL_KNN_T, SL_KNN_I, SL_KNN_D, SL_KNN_S, SL_KNN_I2, SL_KNN_D2, SL_KNN_S2, SL_ST_ROWS, SL_ST_SAD, SL_ST_START, SL_ST_ITEMS,
    SL_ST_U, SL_ST_D, SL_FV1_NODES, SL_FV1_START, SL_FV1_ITEMS, SL_FV2_NODES, SL_FV2_START, SL_FV2_ITEMS, SL_BW_VALID1,
    SL_BW_VALID2, SL_BW_M12, SL_BW_M21, SL_BW_BIN, SL_BW_HIST, SL_BW_UR1, SL_BW_UR2, SL_WB_LEVEL, SL_WB_IDX, SL_WB_DIST,
    SL_WB_QUR, SL_WB_IDX2, SL_WB_DIST2, SL_DD_DESC, SL_DD_START, SL_DD_BEST, SL_DD_MED, SL_DD_OUT,
    SL_MK_MASK, SL_MK_KPS, SL_MK_DESC, SL_RC_REC, SL_RC_IDX, SL_JB_JOBS, SL_JB_I32, SL_JB_U8, SL_JB_TOT, SL_JB_SF, SL_JB_SLAB, SL_SORTED,
    SL_BT_DESC, SL_BT_START, SL_BT_CHILD, SL_BT_WORD, SL_BT_WEIGHT, SL_BT_FEAT, SL_BT_OUTW, SL_BT_OUTV, SL_BT_OUTN
};

// Device view of a caller array: the pointer itself for device memory, else a staged copy.
// ---- pinned staging (host memory mode)
void* arena_alloc(orbm_context* m, size_t bytes) {
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (m->arenaUsed + need > m->arenaCap) {
        // blocks handed out earlier in this call stay valid: the old arena is kept until the next call
        const size_t cap = std::max(m->arenaCap * 2, std::max(need * 2, (size_t)1 << 20));
        flush_downloads(m);
        flush_uploads(m);  // what is staged so far belongs to the old pair of buffers
        void *p = nullptr, *dp = nullptr;
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
            orbx_set_error("hipHostMalloc of %zu bytes failed", cap);
            return nullptr;
        }
        if (hipMalloc(&dp, cap) != hipSuccess) {
            (void)hipHostFree(p);
            orbx_set_error("hipMalloc of %zu bytes failed", cap);
            return nullptr;
        }
        if (m->arena) m->arenaRetired.push_back(m->arena);
        if (m->dArena) m->dArenaRetired.push_back(m->dArena);
        m->arena = (unsigned char*)p;
        m->dArena = (unsigned char*)dp;
        m->upLo = m->upHi = 0;
        m->arenaCap = cap;
        m->arenaUsed = 0;
    }
    void* r = m->arena + m->arenaUsed;
    m->arenaUsed += need;
    return r;
}

// Start of an entry point: nothing of the previous call is in flight any more (host-mode calls end with
// sync_deliver), so the arena starts over.
void arena_reset(orbm_context* m) {
    m->arenaUsed = 0;
    m->upLo = m->upHi = 0;
    m->downloads.clear();
    for (void* p : m->arenaRetired) (void)hipHostFree(p);
    m->arenaRetired.clear();
    for (void* p : m->dArenaRetired) (void)hipFree(p);
    m->dArenaRetired.clear();
    m->pending.clear();
}

// Issues the device-to-host copies of the result pieces recorded so far; pieces that touch go as one copy.
void flush_downloads(orbm_context* m) {
    if (m->downloads.empty()) return;
    std::sort(m->downloads.begin(), m->downloads.end());
    size_t lo = m->downloads[0].first, hi = lo + m->downloads[0].second;
    auto issue = [&]() {
        const hipError_t e = hipMemcpyAsync(m->arena + lo, m->dArena + lo, hi - lo, hipMemcpyDeviceToHost, m->stream);
        if (e != hipSuccess && m->asyncError == hipSuccess) m->asyncError = e;
    };
    for (size_t k = 1; k < m->downloads.size(); k++) {
        const size_t o = m->downloads[k].first, b = m->downloads[k].second;
        if (o <= hi) {
            hi = std::max(hi, o + b);
        } else {
            issue();
            lo = o;
            hi = o + b;
        }
    }
    issue();
    m->downloads.clear();
}

void flush_uploads(orbm_context* m) {
    if (m->upHi <= m->upLo) return;
    const hipError_t e = hipMemcpyAsync(m->dArena + m->upLo, m->arena + m->upLo, m->upHi - m->upLo, hipMemcpyHostToDevice, m->stream);
    if (e != hipSuccess && m->asyncError == hipSuccess) m->asyncError = e;
    m->upLo = m->upHi = 0;
}

int h2d(orbm_context* m, void* dev, const void* host, size_t bytes) {
    if (!bytes) return ORBX_OK;
    void* p = arena_alloc(m, bytes);
    if (!p) return ORBX_ERR_HIP;
    memcpy(p, host, bytes);
    ORBX_HIP_CHECK(hipMemcpyAsync(dev, p, bytes, hipMemcpyHostToDevice, m->stream));
    return ORBX_OK;
}

int d2h(orbm_context* m, void* host, const void* dev, size_t bytes) {
    if (!bytes) return ORBX_OK;
    void* p = arena_alloc(m, bytes);
    if (!p) return ORBX_ERR_HIP;
    ORBX_HIP_CHECK(hipMemcpyAsync(p, dev, bytes, hipMemcpyDeviceToHost, m->stream));
    m->pending.push_back({host, p, bytes});
    return ORBX_OK;
}

hipError_t sync_deliver(orbm_context* m) {
    flush_uploads(m);
    flush_downloads(m);
    const hipError_t e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) return e;
    if (m->asyncError != hipSuccess) {
        const hipError_t a = m->asyncError;
        m->asyncError = hipSuccess;
        return a;
    }
    for (const auto& d : m->pending) memcpy(d.dst, d.src, d.bytes);
    m->pending.clear();
    return hipSuccess;
}

template <typename T>
int stage_in(orbm_context* m, int slot, const T* src, size_t count, int memSpace, const T** out) {
    if (memSpace == ORBX_MEM_DEVICE || src == nullptr) {
        *out = src;
        return ORBX_OK;
    }
    (void)slot;
    const size_t bytes = count * sizeof(T);
    unsigned char* p = (unsigned char*)arena_alloc(m, std::max(bytes, (size_t)64));
    if (!p) return ORBX_ERR_HIP;
    memcpy(p, src, bytes);
    const size_t off = (size_t)(p - m->arena);
    if (m->upHi <= m->upLo) m->upLo = off;
    m->upHi = off + ((bytes + 63) & ~(size_t)63);
    *out = (const T*)(m->dArena + off);
    return ORBX_OK;
}

template <typename T>
int stage_out(orbm_context* m, int slot, T* dst, size_t count, int memSpace, T** out) {
    if (memSpace == ORBX_MEM_DEVICE) {
        *out = dst;
        return ORBX_OK;
    }
    (void)slot;
    unsigned char* p = (unsigned char*)arena_alloc(m, std::max(count * sizeof(T), (size_t)64));
    if (!p) return ORBX_ERR_HIP;
    *out = (T*)(m->dArena + (size_t)(p - m->arena));
    return ORBX_OK;
}

template <typename T>
int copy_back(orbm_context* m, T* host, const T* dev, size_t count, int memSpace) {
    if (memSpace == ORBX_MEM_DEVICE || count == 0) return ORBX_OK;
    const unsigned char* d = (const unsigned char*)dev;
    const size_t bytes = count * sizeof(T);
    if (m->dArena && d >= m->dArena && d + bytes <= m->dArena + m->arenaCap) {
        // a block of the mirror: it lands at the same offset of the pinned arena, fetched at the next wait
        const size_t off = (size_t)(d - m->dArena);
        m->downloads.push_back(std::make_pair(off, std::min((bytes + 63) & ~(size_t)63, m->arenaCap - off)));  // padded: neighbours touch
        m->pending.push_back({host, m->arena + off, bytes});
        return ORBX_OK;
    }
    return d2h(m, host, dev, bytes);
}

int make_frame(orbm_context* m, const orbm_frame* f, int slotKps, int slotDesc, int memSpace, DevFrame* out) {
    if (!f || f->n < 0 || (f->n > 0 && (!f->kps || !f->desc))) {
        orbx_set_error("invalid frame");
        return ORBX_ERR_INVALID;
    }
    if (f->n > 65000) {
        orbx_set_error("frame has %d keypoints; limit is 65000", f->n);
        return ORBX_ERR_UNSUPPORTED;
    }
    int rc = stage_in(m, slotKps, f->kps, (size_t)f->n, memSpace, &out->kps);
    if (rc) return rc;
    rc = stage_in(m, slotDesc, f->desc, (size_t)f->n * 32, memSpace, &out->desc);
    if (rc) return rc;
    out->n = f->n;
    out->minX = f->min_x;
    out->minY = f->min_y;
    out->maxX = f->max_x;
    out->maxY = f->max_y;
    // reference src/Frame.cc:211-212
    out->invW = static_cast<float>(GRID_COLS) / static_cast<float>(f->max_x - f->min_x);
    out->invH = static_cast<float>(GRID_ROWS) / static_cast<float>(f->max_y - f->min_y);
    return ORBX_OK;
}

int build_grid(orbm_context* m, const DevFrame& f, int** cellStart, int** cellItems) {
    void *cs, *ci, *co;
    int rc = pool_get(m, SL_CELL_START, (GRID_CELLS + 1) * sizeof(int), &cs);
    if (rc) return rc;
    rc = pool_get(m, SL_CELL_ITEMS, (size_t)std::max(f.n, 1) * sizeof(int), &ci);
    if (rc) return rc;
    rc = pool_get(m, SL_CELL_OF, (size_t)std::max(f.n, 1) * sizeof(u16), &co);
    if (rc) return rc;
    ORBM_LAUNCH(k_grid_build, dim3(1), dim3(1024), 0, m->stream, f, (int*)cs, (int*)ci, (u16*)co);
    ORBX_HIP_CHECK(hipGetLastError());
    *cellStart = (int*)cs;
    *cellItems = (int*)ci;
    return ORBX_OK;
}

int use_device(orbm_context* m) {
    if (!m) {
        orbx_set_error("null context");
        return ORBX_ERR_INVALID;
    }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    arena_reset(m);
    return ORBX_OK;
}

}  // namespace

extern "C" {

int orbm_hamming256(const uint8_t* a, const uint8_t* b) {
    int d = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * i, 8);
        memcpy(&y, b + 8 * i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

int orbm_create(int device, orbm_context** out) {
    if (!out) return ORBX_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        orbx_set_error("no usable HIP device (count=%d, requested %d)", ndev, device);
        return ORBX_ERR_NO_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) {
        orbx_set_error("hipSetDevice(%d) failed", device);
        return ORBX_ERR_NO_DEVICE;
    }
    orbm_context* m = new orbm_context();
    m->device = device;
    if (hipStreamCreateWithFlags(&m->ownStream, hipStreamNonBlocking) != hipSuccess) {
        delete m;
        orbx_set_error("hipStreamCreate failed");
        return ORBX_ERR_HIP;
    }
    m->stream = m->ownStream;
    // set-up of the no-allocation entry point: this device is current here
    const int rcs = scratch_address(device, &m->naScratch);
    if (rcs != ORBX_OK) {
        (void)hipStreamDestroy(m->ownStream);
        delete m;
        return rcs;
    }
    *out = m;
    return ORBX_OK;
}

void orbm_destroy(orbm_context* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    for (auto& e : m->pool)
        if (e.first) (void)hipFree(e.first);
    for (void* p : m->arenaRetired) (void)hipHostFree(p);
    if (m->arena) (void)hipHostFree(m->arena);
    for (void* p : m->dArenaRetired) (void)hipFree(p);
    if (m->dArena) (void)hipFree(m->dArena);
    if (m->ownStream) (void)hipStreamDestroy(m->ownStream);
    delete m;
}

int orbm_set_stream(orbm_context* m, void* hip_stream) {
    if (!m) return ORBX_ERR_INVALID;
    ORBX_HIP_CHECK(sync_deliver(m));
    m->stream = (hipStream_t)hip_stream;  // NULL is the legacy default stream, like any other handle
    return ORBX_OK;
}

int orbm_reset_stream(orbm_context* m) {
    if (!m) return ORBX_ERR_INVALID;
    ORBX_HIP_CHECK(sync_deliver(m));
    m->stream = m->ownStream;
    return ORBX_OK;
}

int orbm_synchronize(orbm_context* m) {
    if (!m) return ORBX_ERR_INVALID;
    ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

int orbm_grid_build(orbm_context* m, const orbm_frame* f, int32_t* cell_start, int32_t* cell_items, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (!cell_start || !cell_items) return ORBX_ERR_INVALID;
    DevFrame df;
    rc = make_frame(m, f, SL_F2_KPS, SL_F2_DESC, mem_space, &df);
    if (rc) return rc;
    int *cs, *ci;
    rc = build_grid(m, df, &cs, &ci);
    if (rc) return rc;
    if (mem_space == ORBX_MEM_DEVICE) {
        ORBX_HIP_CHECK(hipMemcpyAsync(cell_start, cs, (GRID_CELLS + 1) * sizeof(int), hipMemcpyDeviceToDevice, m->stream));
        if (df.n) ORBX_HIP_CHECK(hipMemcpyAsync(cell_items, ci, (size_t)df.n * sizeof(int), hipMemcpyDeviceToDevice, m->stream));
    } else {
        ORBX_HIP_CHECK(hipMemcpyAsync(cell_start, cs, (GRID_CELLS + 1) * sizeof(int), hipMemcpyDeviceToHost, m->stream));
        ORBX_HIP_CHECK(sync_deliver(m));
        const int total = cell_start[GRID_CELLS];
        if (total) ORBX_HIP_CHECK(hipMemcpy(cell_items, ci, (size_t)total * sizeof(int), hipMemcpyDeviceToHost));
    }
    return ORBX_OK;
}

int orbm_window_candidates(orbm_context* m, const orbm_frame* train, const float* qx, const float* qy, const float* qr,
                           const int32_t* qmin_level, const int32_t* qmax_level, const uint8_t* qdesc, int nq,
                           int max_cand, int32_t* cand_idx, int32_t* cand_dist, int32_t* counts, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (nq < 0 || max_cand < 1 || !qx || !qy || !qr || !cand_idx || !counts) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (nq == 0) return ORBX_OK;
    DevFrame df;
    rc = make_frame(m, train, SL_F2_KPS, SL_F2_DESC, mem_space, &df);
    if (rc) return rc;
    int *cs, *ci;
    rc = build_grid(m, df, &cs, &ci);
    if (rc) return rc;
    QueryArrays qa;
    if ((rc = stage_in(m, SL_QX, qx, (size_t)nq, mem_space, &qa.x))) return rc;
    if ((rc = stage_in(m, SL_QY, qy, (size_t)nq, mem_space, &qa.y))) return rc;
    if ((rc = stage_in(m, SL_QR, qr, (size_t)nq, mem_space, &qa.r))) return rc;
    if ((rc = stage_in(m, SL_QMIN, (const int*)qmin_level, (size_t)nq, mem_space, &qa.minLevel))) return rc;
    if ((rc = stage_in(m, SL_QMAX, (const int*)qmax_level, (size_t)nq, mem_space, &qa.maxLevel))) return rc;
    if ((rc = stage_in(m, SL_QDESC, qdesc, (size_t)nq * 32, mem_space, &qa.desc))) return rc;
    int *dIdx, *dDist = nullptr, *dCnt;
    const size_t rows = (size_t)nq * max_cand;
    if ((rc = stage_out(m, SL_CAND_IDX, (int*)cand_idx, rows, mem_space, &dIdx))) return rc;
    if (cand_dist && (rc = stage_out(m, SL_CAND_DIST, (int*)cand_dist, rows, mem_space, &dDist))) return rc;
    if ((rc = stage_out(m, SL_COUNTS, (int*)counts, (size_t)nq, mem_space, &dCnt))) return rc;
    ORBM_LAUNCH(k_window, dim3((nq + 3) / 4), dim3(256), 0, m->stream, df, (const int*)cs, (const int*)ci, qa,
                       nq, 1, max_cand, (const int*)nullptr, dIdx, dDist, dCnt);
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, (int*)cand_idx, dIdx, rows, mem_space))) return rc;
    if (cand_dist && (rc = copy_back(m, (int*)cand_dist, dDist, rows, mem_space))) return rc;
    if ((rc = copy_back(m, (int*)counts, dCnt, (size_t)nq, mem_space))) return rc;
    if (mem_space == ORBX_MEM_HOST) ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

int orbm_search_for_initialization(orbm_context* m, const orbm_frame* f1, const orbm_frame* f2, float* prev_matched_xy,
                                   int32_t* matches12, int window_size, float nn_ratio, int check_orientation,
                                   int* nmatches, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (!prev_matched_xy || !matches12 || !nmatches) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    DevFrame d1, d2;
    if ((rc = make_frame(m, f1, SL_F1_KPS, SL_F1_DESC, mem_space, &d1))) return rc;
    if ((rc = make_frame(m, f2, SL_F2_KPS, SL_F2_DESC, mem_space, &d2))) return rc;
    *nmatches = 0;
    if (d1.n == 0) return ORBX_OK;
    int *cs, *ci;
    if ((rc = build_grid(m, d2, &cs, &ci))) return rc;
    const int n1 = d1.n, n2 = std::max(d2.n, 1);
    float* dPrev;
    void *qx, *qy, *qr, *qmin, *qmax, *cnt, *offs, *sA, *sB, *sC, *sD, *scal;
    if (mem_space == ORBX_MEM_DEVICE) {
        dPrev = prev_matched_xy;
    } else {
        void* p;
        if ((rc = pool_get(m, SL_PREV, (size_t)n1 * 2 * sizeof(float), &p))) return rc;
        dPrev = (float*)p;
        if ((rc = h2d(m, dPrev, prev_matched_xy, (size_t)n1 * 2 * sizeof(float)))) return rc;
    }
    if ((rc = pool_get(m, SL_QX, (size_t)n1 * 4, &qx)) || (rc = pool_get(m, SL_QY, (size_t)n1 * 4, &qy)) ||
        (rc = pool_get(m, SL_QR, (size_t)n1 * 4, &qr)) || (rc = pool_get(m, SL_QMIN, (size_t)n1 * 4, &qmin)) ||
        (rc = pool_get(m, SL_QMAX, (size_t)n1 * 4, &qmax)) || (rc = pool_get(m, SL_COUNTS, (size_t)n1 * 4, &cnt)) ||
        (rc = pool_get(m, SL_OFFS, (size_t)(n1 + 1) * 4, &offs)) || (rc = pool_get(m, SL_STATE_A, (size_t)n2 * 4, &sA)) ||
        (rc = pool_get(m, SL_STATE_B, (size_t)n2 * 4, &sB)) || (rc = pool_get(m, SL_STATE_C, (size_t)n1 * 4, &sC)) ||
        (rc = pool_get(m, SL_STATE_D, (size_t)n1 * 4, &sD)) || (rc = pool_get(m, SL_SCALAR, 64, &scal)))
        return rc;
    ORBM_LAUNCH(k_init_queries, dim3((n1 + 255) / 256), dim3(256), 0, m->stream, d1, (const float*)dPrev,
                       (float)window_size, (float*)qx, (float*)qy, (float*)qr, (int*)qmin, (int*)qmax);
    QueryArrays qa = {(const float*)qx, (const float*)qy, (const float*)qr, (const int*)qmin, (const int*)qmax, d1.desc};
    ORBM_LAUNCH(k_window, dim3((n1 + 3) / 4), dim3(256), 0, m->stream, d2, (const int*)cs, (const int*)ci, qa, n1,
                       0, 0, (const int*)nullptr, (int*)nullptr, (int*)nullptr, (int*)cnt);
    ORBM_LAUNCH(k_scan, dim3(1), dim3(1024), 0, m->stream, (const int*)cnt, (int*)offs, n1);
    int total = 0;
    if ((rc = d2h(m, &total, (int*)offs + n1, sizeof(int)))) return rc;
    ORBX_HIP_CHECK(sync_deliver(m));
    void *cIdx, *cDist;
    if ((rc = pool_get(m, SL_CAND_IDX, (size_t)std::max(total, 1) * 4, &cIdx)) ||
        (rc = pool_get(m, SL_CAND_DIST, (size_t)std::max(total, 1) * 4, &cDist)))
        return rc;
    ORBM_LAUNCH(k_window, dim3((n1 + 3) / 4), dim3(256), 0, m->stream, d2, (const int*)cs, (const int*)ci, qa, n1,
                       2, 0, (const int*)offs, (int*)cIdx, (int*)cDist, (int*)nullptr);
    int* dM12;
    if ((rc = stage_out(m, SL_STATE_C, (int*)matches12, (size_t)n1, mem_space, &dM12))) return rc;
    const size_t ldsBytes = (size_t)n2 * 12 + (size_t)n1 * 8 + 16;
    if (ldsBytes <= 150 * 1024 && d2.n <= 65536) {
        // rows sorted by distance in parallel, then the serial claims 64 queries at a time from LDS
        void* srt;
        if ((rc = pool_get(m, SL_SORTED, (size_t)std::max(total, 1) * 4, &srt))) return rc;
        if (!m->initSortedReady) {
            ORBX_HIP_CHECK(hipFuncSetAttribute((const void*)k_init_resolve_sorted, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               150 * 1024));
            m->initSortedReady = true;
        }
        ORBM_LAUNCH(k_sort_rows, dim3((n1 + 3) / 4), dim3(256), 0, m->stream, (const int*)offs, (const int*)cIdx,
                           (const int*)cDist, (u32*)srt, n1);
        ORBM_LAUNCH(k_init_resolve_sorted, dim3(1), dim3(256), ldsBytes, m->stream, d1, d2, (const int*)offs,
                           (const u32*)srt, nn_ratio, check_orientation, dM12, dPrev, (int*)scal);
    } else {
        ORBM_LAUNCH(k_init_resolve, dim3(1), dim3(64), 0, m->stream, d1, d2, (const int*)offs, (const int*)cIdx,
                           (const int*)cDist, nn_ratio, check_orientation, (int*)sA, (int*)sB, dM12, (int*)sD, dPrev,
                           (int*)scal);
    }
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, (int*)matches12, (const int*)dM12, (size_t)n1, mem_space))) return rc;
    if ((rc = copy_back(m, prev_matched_xy, (const float*)dPrev, (size_t)n1 * 2, mem_space))) return rc;
    if ((rc = d2h(m, nmatches, scal, sizeof(int)))) return rc;
    ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

// Shared host part: grid of the train frame, candidate CSR, sequential resolve.
static int run_projected(orbm_context* m, const DevFrame& train, const float* dTrainURight, const ProjQueries& q,
                         const ProjPolicy& pol, int32_t* claimed, uint8_t* claimed_obs, int32_t* query_best, int* nmatches,
                         int mem_space) {
    int rc;
    const int nq = q.n, nt = train.n;
    *nmatches = 0;
    if (nq == 0 || nt == 0) {
        if (query_best && nq > 0) {
            if (mem_space == ORBX_MEM_DEVICE) ORBX_HIP_CHECK(hipMemsetAsync(query_best, 0xFF, (size_t)nq * 4, m->stream));
            else for (int i = 0; i < nq; i++) query_best[i] = -1;
        }
        return ORBX_OK;
    }
    // claim state first, so that it travels with the other staged inputs
    const int* dClaimIn;
    const u8* dClaimObsIn;
    if ((rc = stage_in(m, SL_STATE_A, (const int*)claimed, (size_t)nt, mem_space, &dClaimIn)) ||
        (rc = stage_in(m, SL_STATE_B, (const u8*)claimed_obs, (size_t)nt, mem_space, &dClaimObsIn)))
        return rc;
    int* dClaim = const_cast<int*>(dClaimIn);
    u8* dClaimObs = const_cast<u8*>(dClaimObsIn);
    int *cs, *ci;
    if ((rc = build_grid(m, train, &cs, &ci))) return rc;
    void *cnt, *offs, *qn, *scal, *slab;
    // One walk per query when no query has more than SLAB_CAP candidates: they land, distance included, in a
    // fixed slot per query, and the claim resolution packs them into LDS.  Otherwise a second walk fills a CSR.
    const int SLAB_CAP = 32;
    if ((rc = pool_get(m, SL_COUNTS, (size_t)nq * 4, &cnt)) || (rc = pool_get(m, SL_OFFS, (size_t)(nq + 1) * 4, &offs)) ||
        (rc = pool_get(m, SL_STATE_D, (size_t)nq * 4, &qn)) || (rc = pool_get(m, SL_SCALAR, 64, &scal)) ||
        (rc = pool_get(m, SL_JB_SLAB, (size_t)nq * SLAB_CAP * 4, &slab)))
        return rc;
    int* dOverflow = (int*)scal + 8;   // followed by the "declined" flag of the resolution
    ORBX_HIP_CHECK(hipMemsetAsync(dOverflow, 0, 2 * sizeof(int), m->stream));
    ORBM_LAUNCH(k_projq_candidates, dim3((nq + 3) / 4), dim3(256), 0, m->stream, train, (const int*)cs, (const int*)ci,
                dTrainURight, q, 3, (const int*)nullptr, (int*)nullptr, (int*)nullptr, (int*)cnt, (u32*)slab, SLAB_CAP, dOverflow);
    ORBM_LAUNCH(k_scan, dim3(1), dim3(1024), 0, m->stream, (const int*)cnt, (int*)offs, nq);
    int* dQBest;
    if (mem_space == ORBX_MEM_DEVICE && query_best) {
        dQBest = (int*)query_best;
    } else {
        void* p3;
        if ((rc = pool_get(m, SL_STATE_C, (size_t)nq * 4, &p3))) return rc;
        dQBest = (int*)p3;
    }
    if (!m->resolveLdsReady) {
        ORBX_HIP_CHECK(hipFuncSetAttribute((const void*)k_projq_resolve_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           150 * 1024));
        ORBX_HIP_CHECK(hipFuncSetAttribute((const void*)k_projq_resolve_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           150 * 1024));
        ORBX_HIP_CHECK(hipFuncSetAttribute((const void*)k_projq_resolve_lds_auto<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        ORBX_HIP_CHECK(hipFuncSetAttribute((const void*)k_projq_resolve_lds_auto<true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        m->resolveLdsReady = true;
    }
    // The resolution is launched at once, on the full LDS allowance (a single workgroup runs, so the
    // size costs nothing); it declines when the problem is too large for that, which the host learns with the results.
    const size_t fixedBytes = ((size_t)nq + 1) * 4 + (size_t)nt * (8 + (pol.ratioTest ? 4 : 0) + 2) + (size_t)nq + 16;
    static const bool noAuto = getenv("ORBM_NO_AUTO") != nullptr;  // measurement knob: the route with the host round trip
    if (!noAuto && nt <= 65536 && fixedBytes + (size_t)nq * 4 <= 150 * 1024) {
        int* dDeclined = dOverflow + 1;
        auto ka = pol.ratioTest ? k_projq_resolve_lds_auto<true> : k_projq_resolve_lds_auto<false>;
        ORBM_LAUNCH(ka, dim3(1), dim3(256), 150 * 1024, m->stream, train, q, pol, (const int*)offs, dClaim, dClaimObs, dQBest,
                    (int*)qn, (int*)scal, (const u32*)slab, SLAB_CAP, (const int*)dOverflow, (int)fixedBytes, 150 * 1024, dDeclined);
        ORBX_HIP_CHECK(hipGetLastError());
        int declined = 0;
        if ((rc = copy_back(m, (int*)claimed, (const int*)dClaim, (size_t)nt, mem_space))) return rc;
        if ((rc = copy_back(m, claimed_obs, (const u8*)dClaimObs, (size_t)nt, mem_space))) return rc;
        if (query_best && (rc = copy_back(m, (int*)query_best, (const int*)dQBest, (size_t)nq, mem_space))) return rc;
        if ((rc = d2h(m, nmatches, scal, sizeof(int))) || (rc = d2h(m, &declined, dDeclined, sizeof(int)))) return rc;
        ORBX_HIP_CHECK(sync_deliver(m));
        if (!declined) return ORBX_OK;
        *nmatches = 0;
    }
    int total = 0, overflow = 0;
    if ((rc = d2h(m, &total, (int*)offs + nq, sizeof(int))) || (rc = d2h(m, &overflow, dOverflow, sizeof(int)))) return rc;
    ORBX_HIP_CHECK(sync_deliver(m));
    // the whole problem in LDS when it fits (any ordinary frame pair does): see k_projq_resolve_lds
    const size_t ldsBytes = (size_t)total * 4 + ((size_t)nq + 1) * 4 + (size_t)nt * (8 + (pol.ratioTest ? 4 : 0) + 2) + (size_t)nq + 16;
    const bool fits = ldsBytes <= 150 * 1024 && total < 65536 && nt <= 65536;
    const bool oneWalk = fits && !overflow;
    void *cIdx = nullptr, *cDist = nullptr;
    if (!oneWalk) {
        if ((rc = pool_get(m, SL_CAND_IDX, (size_t)std::max(total, 1) * 4, &cIdx)) ||
            (rc = pool_get(m, SL_CAND_DIST, (size_t)std::max(total, 1) * 4, &cDist)))
            return rc;
        ORBM_LAUNCH(k_projq_candidates, dim3((nq + 3) / 4), dim3(256), 0, m->stream, train, (const int*)cs, (const int*)ci,
                    dTrainURight, q, 2, (const int*)offs, (int*)cIdx, (int*)cDist, (int*)nullptr, (u32*)nullptr, 0, (int*)nullptr);
    }
    if (fits) {
        auto kr = pol.ratioTest ? k_projq_resolve_lds<true> : k_projq_resolve_lds<false>;
        ORBM_LAUNCH(kr, dim3(1), dim3(256), ldsBytes, m->stream, train, q, pol, (const int*)offs, (const int*)cIdx,
                    (const int*)cDist, dClaim, dClaimObs, dQBest, (int*)qn, (int*)scal, total,
                    (const u32*)(oneWalk ? slab : nullptr), oneWalk ? SLAB_CAP : 0);
    } else {
        ORBM_LAUNCH(k_projq_resolve, dim3(1), dim3(64), 0, m->stream, train, q, pol, (const int*)offs, (const int*)cIdx,
                           (const int*)cDist, dClaim, dClaimObs, dQBest, (int*)qn, (int*)scal);
    }
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, (int*)claimed, (const int*)dClaim, (size_t)nt, mem_space))) return rc;
    if ((rc = copy_back(m, claimed_obs, (const u8*)dClaimObs, (size_t)nt, mem_space))) return rc;
    if (query_best && (rc = copy_back(m, (int*)query_best, (const int*)dQBest, (size_t)nq, mem_space))) return rc;
    if ((rc = d2h(m, nmatches, scal, sizeof(int)))) return rc;
    ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

int orbm_search_projected(orbm_context* m, const orbm_frame* train, const float* train_u_right,
                          const orbm_proj_queries* q, const orbm_proj_policy* p, int32_t* claimed, uint8_t* claimed_obs,
                          int32_t* query_best, int* nmatches, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (!q || !p || !claimed || !claimed_obs || !nmatches || q->n < 0 ||
        (q->n > 0 && (!q->valid || !q->u || !q->v || !q->radius || !q->min_level || !q->max_level || !q->desc)) ||
        (p->check_orientation && q->n > 0 && !q->angle) || (q->u_right && !q->right_gate) || (q->u_right && !train_u_right)) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    DevFrame dt;
    if ((rc = make_frame(m, train, SL_F2_KPS, SL_F2_DESC, mem_space, &dt))) return rc;
    const size_t n = (size_t)q->n;
    ProjQueries dq;
    dq.n = q->n;
    const float* dUR = nullptr;
    if ((rc = stage_in(m, SL_VALID, q->valid, n, mem_space, &dq.valid)) || (rc = stage_in(m, SL_U, q->u, n, mem_space, &dq.u)) ||
        (rc = stage_in(m, SL_V, q->v, n, mem_space, &dq.v)) || (rc = stage_in(m, SL_QR, q->radius, n, mem_space, &dq.radius)) ||
        (rc = stage_in(m, SL_QMIN, (const int*)q->min_level, n, mem_space, &dq.minLevel)) ||
        (rc = stage_in(m, SL_QMAX, (const int*)q->max_level, n, mem_space, &dq.maxLevel)) ||
        (rc = stage_in(m, SL_QDESC, q->desc, n * 32, mem_space, &dq.desc)) ||
        (rc = stage_in(m, SL_QX, q->angle, n, mem_space, &dq.angle)) ||
        (rc = stage_in(m, SL_HASOBS, q->has_obs, n, mem_space, &dq.hasObs)) ||
        (rc = stage_in(m, SL_INVZ, q->u_right, n, mem_space, &dq.uRight)) ||
        (rc = stage_in(m, SL_QY, q->right_gate, n, mem_space, &dq.rightGate)) ||
        (rc = stage_in(m, SL_URIGHT, train_u_right, (size_t)dt.n, mem_space, &dUR)))
        return rc;
    ProjPolicy pol = {p->claim_rule, p->max_dist, p->ratio_test, p->nn_ratio, p->check_orientation};
    return run_projected(m, dt, dUR, dq, pol, claimed, claimed_obs, query_best, nmatches, mem_space);
}

int orbm_search_by_projection(orbm_context* m, const orbm_frame* cur, const orbm_frame* last, const uint8_t* mp_desc,
                              const uint8_t* valid, const float* u, const float* v, const float* invz,
                              const uint8_t* has_obs, const float* cur_u_right, const float* scale_factors, int nlevels,
                              float bf, float th, int direction, int check_orientation, int32_t* cur_claimed,
                              uint8_t* cur_claimed_obs, int32_t* query_best, int* nmatches, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (!valid || !u || !v || !invz || !has_obs || !cur_u_right || !scale_factors || nlevels < 1 || !cur_claimed ||
        !cur_claimed_obs || !nmatches) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    DevFrame dc, dl;
    if ((rc = make_frame(m, cur, SL_F2_KPS, SL_F2_DESC, mem_space, &dc))) return rc;
    if ((rc = make_frame(m, last, SL_F1_KPS, SL_F1_DESC, mem_space, &dl))) return rc;
    const int nl = dl.n;
    const u8 *dValid, *dHasObs, *dMpDesc;
    const float *dU, *dV, *dInvz, *dUR;
    if ((rc = stage_in(m, SL_VALID, valid, (size_t)nl, mem_space, &dValid)) ||
        (rc = stage_in(m, SL_U, u, (size_t)nl, mem_space, &dU)) || (rc = stage_in(m, SL_V, v, (size_t)nl, mem_space, &dV)) ||
        (rc = stage_in(m, SL_INVZ, invz, (size_t)nl, mem_space, &dInvz)) ||
        (rc = stage_in(m, SL_HASOBS, has_obs, (size_t)nl, mem_space, &dHasObs)) ||
        (rc = stage_in(m, SL_QDESC, mp_desc, (size_t)nl * 32, mem_space, &dMpDesc)) ||
        (rc = stage_in(m, SL_URIGHT, cur_u_right, (size_t)dc.n, mem_space, &dUR)))
        return rc;
    void *sf, *qv, *qr, *qmin, *qmax, *qa, *qur;
    const size_t n1 = (size_t)std::max(nl, 1);
    if ((rc = pool_get(m, SL_SF, (size_t)nlevels * sizeof(float), &sf)) || (rc = pool_get(m, SL_KNN_Q, n1, &qv)) ||
        (rc = pool_get(m, SL_QR, n1 * 4, &qr)) || (rc = pool_get(m, SL_QMIN, n1 * 4, &qmin)) ||
        (rc = pool_get(m, SL_QMAX, n1 * 4, &qmax)) || (rc = pool_get(m, SL_QX, n1 * 4, &qa)) ||
        (rc = pool_get(m, SL_QY, n1 * 4, &qur)))
        return rc;
    ORBX_HIP_CHECK(hipMemcpyAsync(sf, scale_factors, (size_t)nlevels * sizeof(float), hipMemcpyDefault, m->stream));
    if (nl > 0)
        ORBM_LAUNCH(k_m4_queries, dim3((nl + 255) / 256), dim3(256), 0, m->stream, dc, dl, dValid, dU, dV, dInvz,
                           (const float*)sf, bf, th, direction, (u8*)qv, (float*)qr, (int*)qmin, (int*)qmax, (float*)qa,
                           (float*)qur);
    ProjQueries dq;
    dq.n = nl;
    dq.valid = (const u8*)qv;
    dq.u = dU;
    dq.v = dV;
    dq.radius = (const float*)qr;
    dq.minLevel = (const int*)qmin;
    dq.maxLevel = (const int*)qmax;
    dq.desc = mp_desc ? dMpDesc : dl.desc;
    dq.angle = (const float*)qa;
    dq.hasObs = dHasObs;
    dq.uRight = (const float*)qur;
    dq.rightGate = (const float*)qr;  // er > radius rejects (reference :1619)
    ProjPolicy pol = {0, TH_HIGH, 0, 0.f, check_orientation};
    return run_projected(m, dc, dUR, dq, pol, cur_claimed, cur_claimed_obs, query_best, nmatches, mem_space);
}

// The launches of a table of projected-search jobs (one job per frame pair or per (frame, point set)):
// grids, the frame-to-frame query set-up when m4, one walk of every window into the per-query slots, and
// the claim resolution from LDS; the two-walk CSR route when a slot overflowed or a job exceeds the LDS.
// `th` only selects the form of the window walk (queries per wave).  On return m->totalsHost[i] is job i's
// number of matches.
static int run_projection_jobs(orbm_context* m, ProjJob* jobs, void* dJobs, int npairs, int maxNl, ProjPolicy pol, bool m4,
                               const void* dSf, float bf, float th, void* dTot, int* dOverflow, int* nmArr) {
    int rc;
    const size_t jobBytes = (size_t)npairs * sizeof(ProjJob);
    ORBX_HIP_CHECK(hipMemcpyAsync(dJobs, jobs, jobBytes, hipMemcpyHostToDevice, m->stream));
    const ProjJob* dJ = (const ProjJob*)dJobs;
    ORBM_LAUNCH(k_grid_build_jobs, dim3(npairs), dim3(1024), 0, m->stream, dJ);
    // windows of a few grid cells (radius th * scale factor of the level) go four queries to a wave
    const bool smallWindows = th <= 15.f;
    auto launch_candidates = [&](int mode) {
        if (th <= 8.f)
            ORBM_LAUNCH(k_projq_candidates_jobs_sub<8>, dim3((maxNl + 31) / 32, npairs), dim3(256), 0, m->stream, dJ, mode);
        else if (smallWindows)
            ORBM_LAUNCH(k_projq_candidates_jobs_sub<16>, dim3((maxNl + 15) / 16, npairs), dim3(256), 0, m->stream, dJ, mode);
        else
            ORBM_LAUNCH(k_projq_candidates_jobs, dim3((maxNl + 3) / 4, npairs), dim3(256), 0, m->stream, dJ, mode);
    };
    if (maxNl > 0) {
        if (m4)
            ORBM_LAUNCH(k_m4_queries_jobs, dim3((maxNl + 255) / 256, npairs), dim3(256), 0, m->stream, dJ, (const float*)dSf, bf,
                        th);
        launch_candidates(3);
    }
    ORBM_LAUNCH(k_scan_jobs, dim3(npairs), dim3(1024), 0, m->stream, dJ, (int*)dTot);
    ORBX_HIP_CHECK(hipGetLastError());
    m->totalsHost.resize((size_t)npairs);
    int overflow = 0;
    if ((rc = d2h(m, m->totalsHost.data(), dTot, (size_t)npairs * sizeof(int))) || (rc = d2h(m, &overflow, dOverflow, sizeof(int))))
        return rc;
    ORBX_HIP_CHECK(sync_deliver(m));
    size_t maxLds = 0;
    bool fits = true;
    for (int i = 0; i < npairs; i++) {
        const size_t tot = (size_t)m->totalsHost[i], nq = (size_t)jobs[i].q.n, nt = (size_t)jobs[i].train.n;
        maxLds = std::max(maxLds, tot * 4 + (nq + 1) * 4 + nt * (8 + (pol.ratioTest ? 4 : 0) + 2) + nq + 16);
        fits = fits && tot < 65536 && nt <= 65536;
    }
    fits = fits && maxLds <= 150 * 1024;
    if (fits && !m->resolveJobsLdsReady) {
        ORBX_HIP_CHECK(hipFuncSetAttribute((const void*)k_projq_resolve_lds_jobs<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        ORBX_HIP_CHECK(hipFuncSetAttribute((const void*)k_projq_resolve_lds_jobs<true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        m->resolveJobsLdsReady = true;
    }
    auto resolve_lds = [&](int fromSlab) {
        if (pol.ratioTest)
            ORBM_LAUNCH(k_projq_resolve_lds_jobs<true>, dim3(npairs), dim3(256), maxLds, m->stream, dJ, pol, fromSlab);
        else
            ORBM_LAUNCH(k_projq_resolve_lds_jobs<false>, dim3(npairs), dim3(256), maxLds, m->stream, dJ, pol, fromSlab);
    };
    const bool oneWalk = fits && !overflow;
    if (oneWalk) {
        resolve_lds(1);
    } else {
    size_t total = 0;
    for (int i = 0; i < npairs; i++) total += (size_t)m->totalsHost[i];
    void *cIdx, *cDist;
    if ((rc = pool_get(m, SL_CAND_IDX, std::max(total, (size_t)1) * 4, &cIdx)) ||
        (rc = pool_get(m, SL_CAND_DIST, std::max(total, (size_t)1) * 4, &cDist)))
        return rc;
    size_t off = 0;
    for (int i = 0; i < npairs; i++) {
        jobs[i].candIdx = (int*)cIdx + off;
        jobs[i].candDist = (int*)cDist + off;
        off += (size_t)m->totalsHost[i];
    }
    ORBX_HIP_CHECK(hipMemcpyAsync(dJobs, jobs, jobBytes, hipMemcpyHostToDevice, m->stream));
    if (maxNl > 0) launch_candidates(2);
    if (fits)
        resolve_lds(0);
    else
        ORBM_LAUNCH(k_projq_resolve_jobs, dim3(npairs), dim3(64), 0, m->stream, dJ, pol);
    }
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = d2h(m, m->totalsHost.data(), nmArr, (size_t)npairs * sizeof(int)))) return rc;
    ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

int orbm_search_by_projection_batch(orbm_context* m, orbm_projection_pair* pairs, int npairs, const float* scale_factors,
                                    int nlevels, float bf, float th, int check_orientation) {
    int rc = use_device(m);
    if (rc) return rc;
    if (npairs < 0 || (npairs > 0 && !pairs) || !scale_factors || nlevels < 1 || nlevels > ORBM_MAX_LEVELS) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (npairs == 0) return ORBX_OK;
    // scratch layout: 32-bit words and bytes, one slice per pair
    size_t words = 0, bytes = 0;
    int maxNl = 0;
    for (int i = 0; i < npairs; i++) {
        const orbm_projection_pair& P = pairs[i];
        if (P.cur.n < 0 || P.last.n < 0 || P.cur.n > 65000 || P.last.n > 65000 ||
            (P.last.n > 0 && (!P.valid || !P.u || !P.v || !P.invz || !P.has_obs || !P.last.kps || !P.last.desc)) ||
            (P.cur.n > 0 && (!P.cur_u_right || !P.cur_claimed || !P.cur_claimed_obs || !P.cur.kps || !P.cur.desc))) {
            orbx_set_error("pair %d: invalid argument", i);
            return ORBX_ERR_INVALID;
        }
        const size_t nl = (size_t)P.last.n, nc = (size_t)P.cur.n;
        words += (GRID_CELLS + 1) + nc + nl + (nl + 1) + 7 * nl + 1 + 8;  // cellStart, cellItems, counts, offs, qBin/qMin/qMax/qBest/qRadius/qAngle/qUR, nmatches
        bytes += 2 * nc + nl + 16;                                       // cellOf, qValid
        maxNl = std::max(maxNl, P.last.n);
    }
    void *dJobs, *dWords, *dBytes, *dTot, *dSf;
    if ((rc = pool_get(m, SL_JB_JOBS, (size_t)npairs * sizeof(ProjJob), &dJobs)) ||
        (rc = pool_get(m, SL_JB_I32, words * 4, &dWords)) || (rc = pool_get(m, SL_JB_U8, bytes, &dBytes)) ||
        (rc = pool_get(m, SL_JB_TOT, ((size_t)npairs * 2 + 8) * sizeof(int), &dTot)) ||
        (rc = pool_get(m, SL_JB_SF, ORBM_MAX_LEVELS * sizeof(float), &dSf)))
        return rc;
    if ((rc = h2d(m, dSf, scale_factors, (size_t)nlevels * sizeof(float)))) return rc;
    // One walk per query when no query has more than SLAB_CAP candidates (the rule for projected
    // searches): candidates land in a fixed slot per query and the claim resolution packs them into LDS.
    const int SLAB_CAP = 32;
    size_t slabWords = 0;
    for (int i = 0; i < npairs; i++) slabWords += (size_t)pairs[i].last.n * SLAB_CAP;
    void* dSlab;
    if ((rc = pool_get(m, SL_JB_SLAB, std::max(slabWords, (size_t)1) * 4, &dSlab))) return rc;
    int* dOverflow = (int*)dTot + 2 * (size_t)npairs;
    ORBX_HIP_CHECK(hipMemsetAsync(dOverflow, 0, sizeof(int), m->stream));
    size_t slabOff = 0;
    // the job table is built in pinned memory, so its upload is one asynchronous copy
    ProjJob* jobs = (ProjJob*)arena_alloc(m, (size_t)npairs * sizeof(ProjJob));
    if (!jobs) return ORBX_ERR_HIP;
    int* wp = (int*)dWords;
    u8* bp = (u8*)dBytes;
    int* nmArr = (int*)dTot + npairs;
    for (int i = 0; i < npairs; i++) {
        const orbm_projection_pair& P = pairs[i];
        ProjJob& J = jobs[i];
        memset(&J, 0, sizeof(J));
        if ((rc = make_frame(m, &P.cur, 0, 0, ORBX_MEM_DEVICE, &J.train))) return rc;
        if ((rc = make_frame(m, &P.last, 0, 0, ORBX_MEM_DEVICE, &J.last))) return rc;
        const int nl = P.last.n, nc = P.cur.n;
        auto takeW = [&](size_t n) {
            int* r = wp;
            wp += n;
            return r;
        };
        J.trainURight = P.cur_u_right;
        J.cellStart = takeW(GRID_CELLS + 1);
        J.cellItems = takeW((size_t)nc);
        J.counts = takeW((size_t)nl);
        J.offs = takeW((size_t)nl + 1);
        J.qBin = takeW((size_t)nl);
        J.qMin = takeW((size_t)nl);
        J.qMax = takeW((size_t)nl);
        int* ownBest = takeW((size_t)nl);
        J.qRadius = (float*)takeW((size_t)nl);
        J.qAngle = (float*)takeW((size_t)nl);
        J.qUR = (float*)takeW((size_t)nl);
        wp += 8;
        J.cellOf = (u16*)bp;
        bp += 2 * (size_t)nc;
        J.qValid = bp;
        bp += (size_t)nl;
        bp = (u8*)(((uintptr_t)bp + 15) & ~(uintptr_t)15);
        J.claimed = (int*)P.cur_claimed;
        J.claimedObs = P.cur_claimed_obs;
        J.qBest = P.query_best ? (int*)P.query_best : ownBest;
        J.nmatches = nmArr + i;
        J.slab = (u32*)dSlab + slabOff;
        J.slabCap = SLAB_CAP;
        J.overflow = dOverflow;
        slabOff += (size_t)nl * SLAB_CAP;
        J.m4Valid = P.valid;
        J.m4U = P.u;
        J.m4V = P.v;
        J.m4Invz = P.invz;
        J.direction = P.direction;
        J.q.n = nl;
        J.q.valid = J.qValid;
        J.q.u = P.u;
        J.q.v = P.v;
        J.q.radius = J.qRadius;
        J.q.minLevel = J.qMin;
        J.q.maxLevel = J.qMax;
        J.q.desc = P.mp_desc ? P.mp_desc : P.last.desc;
        J.q.angle = J.qAngle;
        J.q.hasObs = P.has_obs;
        J.q.uRight = J.qUR;
        J.q.rightGate = J.qRadius;  // er > radius rejects (reference src/ORBmatcher.cc:1619)
    }
    ProjPolicy pol = {0, TH_HIGH, 0, 0.f, check_orientation};
    if ((rc = run_projection_jobs(m, jobs, dJobs, npairs, maxNl, pol, true, dSf, bf, th, dTot, dOverflow, nmArr))) return rc;
    for (int i = 0; i < npairs; i++) pairs[i].nmatches = m->totalsHost[i];
    return ORBX_OK;
}

int orbm_search_projected_batch(orbm_context* m, orbm_projected_job* jobsIn, int njobs, const orbm_proj_policy* p,
                                float window_hint) {
    int rc = use_device(m);
    if (rc) return rc;
    if (njobs < 0 || (njobs > 0 && !jobsIn) || !p) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (njobs == 0) return ORBX_OK;
    size_t words = 0, bytes = 0;
    int maxNq = 0;
    for (int i = 0; i < njobs; i++) {
        const orbm_projected_job& P = jobsIn[i];
        const orbm_proj_queries& Q = P.q;
        if (P.train.n < 0 || Q.n < 0 || P.train.n > 65000 || Q.n > 65000 ||
            (Q.n > 0 && (!Q.valid || !Q.u || !Q.v || !Q.radius || !Q.min_level || !Q.max_level || !Q.desc)) ||
            (p->check_orientation && Q.n > 0 && !Q.angle) || (Q.u_right && !Q.right_gate) ||
            (P.train.n > 0 && (!P.claimed || !P.claimed_obs || !P.train.kps || !P.train.desc))) {
            orbx_set_error("job %d: invalid argument", i);
            return ORBX_ERR_INVALID;
        }
        const size_t nq = (size_t)Q.n, nt = (size_t)P.train.n;
        words += (GRID_CELLS + 1) + nt + nq + (nq + 1) + 2 * nq + 1 + 8;  // cellStart, cellItems, counts, offs, qBin, qBest, nmatches
        bytes += 2 * nt + 16;                                             // cellOf
        maxNq = std::max(maxNq, Q.n);
    }
    void *dJobs, *dWords, *dBytes, *dTot;
    if ((rc = pool_get(m, SL_JB_JOBS, (size_t)njobs * sizeof(ProjJob), &dJobs)) ||
        (rc = pool_get(m, SL_JB_I32, words * 4, &dWords)) || (rc = pool_get(m, SL_JB_U8, bytes, &dBytes)) ||
        (rc = pool_get(m, SL_JB_TOT, ((size_t)njobs * 2 + 8) * sizeof(int), &dTot)))
        return rc;
    const int SLAB_CAP = 32;
    size_t slabWords = 0;
    for (int i = 0; i < njobs; i++) slabWords += (size_t)jobsIn[i].q.n * SLAB_CAP;
    void* dSlab;
    if ((rc = pool_get(m, SL_JB_SLAB, std::max(slabWords, (size_t)1) * 4, &dSlab))) return rc;
    int* dOverflow = (int*)dTot + 2 * (size_t)njobs;
    ORBX_HIP_CHECK(hipMemsetAsync(dOverflow, 0, sizeof(int), m->stream));
    ProjJob* jobs = (ProjJob*)arena_alloc(m, (size_t)njobs * sizeof(ProjJob));
    if (!jobs) return ORBX_ERR_HIP;
    int* wp = (int*)dWords;
    u8* bp = (u8*)dBytes;
    int* nmArr = (int*)dTot + njobs;
    size_t slabOff = 0;
    for (int i = 0; i < njobs; i++) {
        const orbm_projected_job& P = jobsIn[i];
        ProjJob& J = jobs[i];
        memset(&J, 0, sizeof(J));
        if ((rc = make_frame(m, &P.train, 0, 0, ORBX_MEM_DEVICE, &J.train))) return rc;
        const int nq = P.q.n, nt = P.train.n;
        auto takeW = [&](size_t n) {
            int* r = wp;
            wp += n;
            return r;
        };
        J.trainURight = P.train_u_right;
        J.cellStart = takeW(GRID_CELLS + 1);
        J.cellItems = takeW((size_t)nt);
        J.counts = takeW((size_t)nq);
        J.offs = takeW((size_t)nq + 1);
        J.qBin = takeW((size_t)nq);
        int* ownBest = takeW((size_t)nq);
        wp += 8;
        J.cellOf = (u16*)bp;
        bp += 2 * (size_t)nt;
        bp = (u8*)(((uintptr_t)bp + 15) & ~(uintptr_t)15);
        J.claimed = (int*)P.claimed;
        J.claimedObs = P.claimed_obs;
        J.qBest = P.query_best ? (int*)P.query_best : ownBest;
        J.nmatches = nmArr + i;
        J.slab = (u32*)dSlab + slabOff;
        J.slabCap = SLAB_CAP;
        J.overflow = dOverflow;
        slabOff += (size_t)nq * SLAB_CAP;
        J.q.n = nq;
        J.q.valid = P.q.valid;
        J.q.u = P.q.u;
        J.q.v = P.q.v;
        J.q.radius = P.q.radius;
        J.q.minLevel = P.q.min_level;
        J.q.maxLevel = P.q.max_level;
        J.q.desc = P.q.desc;
        J.q.angle = P.q.angle;
        J.q.hasObs = P.q.has_obs;  // NULL = every point has observations
        J.q.uRight = P.q.u_right;
        J.q.rightGate = P.q.right_gate;
    }
    ProjPolicy pol = {p->claim_rule, p->max_dist, p->ratio_test, p->nn_ratio, p->check_orientation};
    if ((rc = run_projection_jobs(m, jobs, dJobs, njobs, maxNq, pol, false, nullptr, 0.f, window_hint > 0.f ? window_hint : 1e9f,
                                  dTot, dOverflow, nmArr)))
        return rc;
    for (int i = 0; i < njobs; i++) jobsIn[i].nmatches = m->totalsHost[i];
    return ORBX_OK;
}

int orbm_search_for_initialization_batch(orbm_context* m, orbm_init_job* jobsIn, int njobs, int window_size, float nn_ratio,
                                         int check_orientation) {
    int rc = use_device(m);
    if (rc) return rc;
    if (njobs < 0 || (njobs > 0 && !jobsIn)) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (njobs == 0) return ORBX_OK;
    size_t words = 0, bytes = 0;
    int maxN1 = 0;
    for (int i = 0; i < njobs; i++) {
        const orbm_init_job& P = jobsIn[i];
        if (P.f1.n < 0 || P.f2.n < 0 || P.f1.n > 65000 || P.f2.n > 65000 ||
            (P.f1.n > 0 && (!P.f1.kps || !P.f1.desc || !P.prev_matched_xy || !P.matches12)) ||
            (P.f2.n > 0 && (!P.f2.kps || !P.f2.desc))) {
            orbx_set_error("job %d: invalid argument", i);
            return ORBX_ERR_INVALID;
        }
        const size_t n1 = (size_t)P.f1.n, n2 = (size_t)P.f2.n;
        // cellStart, cellItems, counts, offs, qMin, qMax, qBin, u, v, radius, vMatchedDistance, vnMatches21, nmatches
        words += (GRID_CELLS + 1) + n2 + n1 + (n1 + 1) + 6 * n1 + 2 * n2 + 1 + 8;
        bytes += 2 * n2 + n1 + 32;  // cellOf, qValid
        maxN1 = std::max(maxN1, P.f1.n);
    }
    void *dJobs, *dWords, *dBytes, *dTot;
    if ((rc = pool_get(m, SL_JB_JOBS, (size_t)njobs * sizeof(ProjJob), &dJobs)) ||
        (rc = pool_get(m, SL_JB_I32, words * 4, &dWords)) || (rc = pool_get(m, SL_JB_U8, bytes, &dBytes)) ||
        (rc = pool_get(m, SL_JB_TOT, ((size_t)njobs * 2 + 8) * sizeof(int), &dTot)))
        return rc;
    ProjJob* jobs = (ProjJob*)arena_alloc(m, (size_t)njobs * sizeof(ProjJob));
    if (!jobs) return ORBX_ERR_HIP;
    int* wp = (int*)dWords;
    u8* bp = (u8*)dBytes;
    int* nmArr = (int*)dTot + njobs;
    for (int i = 0; i < njobs; i++) {
        const orbm_init_job& P = jobsIn[i];
        ProjJob& J = jobs[i];
        memset(&J, 0, sizeof(J));
        if ((rc = make_frame(m, &P.f2, 0, 0, ORBX_MEM_DEVICE, &J.train))) return rc;
        if ((rc = make_frame(m, &P.f1, 0, 0, ORBX_MEM_DEVICE, &J.last))) return rc;
        const size_t n1 = (size_t)P.f1.n, n2 = (size_t)P.f2.n;
        auto takeW = [&](size_t n) {
            int* r = wp;
            wp += n;
            return r;
        };
        J.cellStart = takeW(GRID_CELLS + 1);
        J.cellItems = takeW(n2);
        J.counts = takeW(n1);
        J.offs = takeW(n1 + 1);
        J.qMin = takeW(n1);
        J.qMax = takeW(n1);
        J.qBin = takeW(n1);
        J.initU = (float*)takeW(n1);
        J.initV = (float*)takeW(n1);
        J.qRadius = (float*)takeW(n1);
        J.claimed = takeW(n2);  // vMatchedDistance of the wave-per-pair form
        J.initM21 = takeW(n2);
        wp += 8;
        J.cellOf = (u16*)bp;
        bp += 2 * n2;
        J.qValid = bp;
        bp += n1;
        bp = (u8*)(((uintptr_t)bp + 15) & ~(uintptr_t)15);
        J.initPrev = P.prev_matched_xy;
        J.qBest = (int*)P.matches12;
        J.nmatches = nmArr + i;
        J.q.n = P.f1.n;
        J.q.valid = J.qValid;
        J.q.u = J.initU;
        J.q.v = J.initV;
        J.q.radius = J.qRadius;
        J.q.minLevel = J.qMin;
        J.q.maxLevel = J.qMax;
        J.q.desc = P.f1.desc;
    }
    const size_t jobBytes = (size_t)njobs * sizeof(ProjJob);
    ORBX_HIP_CHECK(hipMemcpyAsync(dJobs, jobs, jobBytes, hipMemcpyHostToDevice, m->stream));
    const ProjJob* dJ = (const ProjJob*)dJobs;
    ORBM_LAUNCH(k_grid_build_jobs, dim3(njobs), dim3(1024), 0, m->stream, dJ);
    if (maxN1 > 0) {
        ORBM_LAUNCH(k_init_queries_jobs, dim3((maxN1 + 255) / 256, njobs), dim3(256), 0, m->stream, dJ, (float)window_size);
        ORBM_LAUNCH(k_projq_candidates_jobs, dim3((maxN1 + 3) / 4, njobs), dim3(256), 0, m->stream, dJ, 0);
    }
    ORBM_LAUNCH(k_scan_jobs, dim3(njobs), dim3(1024), 0, m->stream, dJ, (int*)dTot);
    ORBX_HIP_CHECK(hipGetLastError());
    m->totalsHost.resize((size_t)njobs);
    if ((rc = d2h(m, m->totalsHost.data(), dTot, (size_t)njobs * sizeof(int)))) return rc;
    ORBX_HIP_CHECK(sync_deliver(m));
    size_t total = 0, maxLds = 0;
    bool fits = true;
    for (int i = 0; i < njobs; i++) {
        total += (size_t)m->totalsHost[i];
        const size_t n1 = (size_t)jobsIn[i].f1.n, n2 = (size_t)std::max(jobsIn[i].f2.n, 1);
        maxLds = std::max(maxLds, n2 * 12 + n1 * 8 + 16);
    }
    fits = maxLds <= 150 * 1024;
    void *cIdx, *cDist, *srt;
    if ((rc = pool_get(m, SL_CAND_IDX, std::max(total, (size_t)1) * 4, &cIdx)) ||
        (rc = pool_get(m, SL_CAND_DIST, std::max(total, (size_t)1) * 4, &cDist)) ||
        (rc = pool_get(m, SL_SORTED, std::max(total, (size_t)1) * 4, &srt)))
        return rc;
    size_t off = 0;
    for (int i = 0; i < njobs; i++) {
        jobs[i].candIdx = (int*)cIdx + off;
        jobs[i].candDist = (int*)cDist + off;
        jobs[i].slab = (u32*)srt + off;  // the rows again, sorted by distance
        off += (size_t)m->totalsHost[i];
    }
    ORBX_HIP_CHECK(hipMemcpyAsync(dJobs, jobs, jobBytes, hipMemcpyHostToDevice, m->stream));
    if (maxN1 > 0) ORBM_LAUNCH(k_projq_candidates_jobs, dim3((maxN1 + 3) / 4, njobs), dim3(256), 0, m->stream, dJ, 2);
    if (fits) {
        if (!m->initSortedJobsReady) {
            ORBX_HIP_CHECK(hipFuncSetAttribute((const void*)k_init_resolve_sorted_jobs, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               150 * 1024));
            m->initSortedJobsReady = true;
        }
        if (maxN1 > 0) ORBM_LAUNCH(k_sort_rows_jobs, dim3((maxN1 + 3) / 4, njobs), dim3(256), 0, m->stream, dJ);
        ORBM_LAUNCH(k_init_resolve_sorted_jobs, dim3(njobs), dim3(256), maxLds, m->stream, dJ, nn_ratio, check_orientation);
    } else {
        ORBM_LAUNCH(k_init_resolve_jobs, dim3(njobs), dim3(64), 0, m->stream, dJ, nn_ratio, check_orientation);
    }
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = d2h(m, m->totalsHost.data(), nmArr, (size_t)njobs * sizeof(int)))) return rc;
    ORBX_HIP_CHECK(sync_deliver(m));
    for (int i = 0; i < njobs; i++) jobsIn[i].nmatches = m->totalsHost[i];
    return ORBX_OK;
}

static int fill_pyramid(const orbm_pyramid* p, DevPyramid* d, const char* which) {
    if (!p || p->nlevels < 1 || p->nlevels > MAX_LEVELS) {
        orbx_set_error("%s pyramid: nlevels must be in [1, %d]", which, MAX_LEVELS);
        return ORBX_ERR_INVALID;
    }
    memset(d, 0, sizeof(*d));
    for (int l = 0; l < p->nlevels; l++) {
        if (!p->level[l] || p->width[l] <= 0 || p->height[l] <= 0 || p->pitch[l] < (size_t)p->width[l] ||
            p->pitch[l] > 0x7fffffff) {
            orbx_set_error("%s pyramid: level %d is not a valid image", which, l);
            return ORBX_ERR_INVALID;
        }
        d->ptr[l] = p->level[l];
        d->pitch[l] = (int)p->pitch[l];
        d->w[l] = p->width[l];
        d->h[l] = p->height[l];
    }
    return ORBX_OK;
}

// The three launches of the stereo step for a table of pairs already in device memory.
static int run_stereo_jobs(orbm_context* m, const StereoJob* dJobs, int njobs, int maxL, const StereoParams& sp) {
    ORBM_LAUNCH(k_stereo_rows_jobs, dim3(njobs), dim3(1024), 0, m->stream, dJobs, sp);
    ORBX_HIP_CHECK(hipGetLastError());
    if (maxL) {
        ORBM_LAUNCH(k_stereo_match_jobs, dim3((maxL + 3) / 4, njobs), dim3(256), 0, m->stream, dJobs, sp);
        ORBX_HIP_CHECK(hipGetLastError());
    }
    ORBM_LAUNCH(k_stereo_median_jobs, dim3(njobs), dim3(1024), 0, m->stream, dJobs);
    ORBX_HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

static int stereo_params(const float* scale_factors, const float* inv_scale_factors, int nlevels, float mb, float mbf,
                         StereoParams* sp, int* bandMax) {
    memset(sp, 0, sizeof(*sp));
    sp->nlevels = nlevels;
    float smax = 1.0f;
    for (int l = 0; l < nlevels; l++) {
        sp->scale[l] = scale_factors[l];
        sp->invScale[l] = inv_scale_factors[l];
        if (scale_factors[l] > smax) smax = scale_factors[l];
    }
    if (!(smax < 1000.0f)) {
        orbx_set_error("scale factors out of range");
        return ORBX_ERR_INVALID;
    }
    sp->mbf = mbf;
    sp->maxD = mbf / mb;  // minZ = mb, reference src/Frame.cc:633-636
    // rows a right keypoint can be listed on: ceil(y + r) - floor(y - r) + 1 <= 2 r + 3 with r = 2 * scale
    *bandMax = (int)(4.0f * smax) + 4;
    return ORBX_OK;
}

int orbm_compute_stereo_matches(orbm_context* m, const orbm_frame* left, const orbm_frame* right,
                                const orbm_pyramid* pyr_left, const orbm_pyramid* pyr_right, const float* scale_factors,
                                const float* inv_scale_factors, float mb, float mbf, float* u_right, float* depth,
                                int* n_stereo, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (!scale_factors || !inv_scale_factors || !u_right || !depth || !pyr_left || !pyr_right) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    StereoJob J;
    memset(&J, 0, sizeof(J));
    if ((rc = fill_pyramid(pyr_left, &J.pl, "left"))) return rc;
    if ((rc = fill_pyramid(pyr_right, &J.pr, "right"))) return rc;
    if (pyr_left->nlevels != pyr_right->nlevels) {
        orbx_set_error("left and right pyramids differ in depth");
        return ORBX_ERR_INVALID;
    }
    if ((rc = make_frame(m, left, SL_F1_KPS, SL_F1_DESC, mem_space, &J.left))) return rc;
    if ((rc = make_frame(m, right, SL_F2_KPS, SL_F2_DESC, mem_space, &J.right))) return rc;
    if (n_stereo) *n_stereo = 0;
    if (J.left.n == 0) return ORBX_OK;
    StereoParams sp;
    int bandMax;
    if ((rc = stereo_params(scale_factors, inv_scale_factors, pyr_left->nlevels, mb, mbf, &sp, &bandMax))) return rc;
    J.nRows = J.pl.h[0];  // reference src/Frame.cc:610
    if (J.nRows > STEREO_MAX_ROWS) {
        orbx_set_error("stereo: images of more than %d rows are not supported", STEREO_MAX_ROWS);
        return ORBX_ERR_UNSUPPORTED;
    }
    void *dRows, *dSad, *dKept, *dStart, *dItems;
    if ((rc = pool_get(m, SL_ST_ROWS, (size_t)std::max(J.right.n, 1) * sizeof(int4), &dRows)) ||
        (rc = pool_get(m, SL_ST_SAD, (size_t)J.left.n * sizeof(int), &dSad)) || (rc = pool_get(m, SL_SCALAR, 64, &dKept)) ||
        (rc = pool_get(m, SL_ST_START, (size_t)(J.nRows + 1) * sizeof(int), &dStart)) ||
        (rc = pool_get(m, SL_ST_ITEMS, (size_t)std::max(J.right.n, 1) * bandMax * sizeof(int), &dItems)))
        return rc;
    if ((rc = stage_out(m, SL_ST_U, u_right, (size_t)J.left.n, mem_space, &J.uRight))) return rc;
    if ((rc = stage_out(m, SL_ST_D, depth, (size_t)J.left.n, mem_space, &J.depth))) return rc;
    J.rows = (int4*)dRows;
    J.rowStart = (int*)dStart;
    J.rowItems = (int*)dItems;
    J.sad = (int*)dSad;
    J.kept = (int*)dKept;
    J.useLists = J.right.n > 1536;  // one pair per call is a latency path: below that the direct scan is faster than the set-up of the lists
    ORBM_LAUNCH(k_stereo_rows_one, dim3(1), dim3(1024), 0, m->stream, J, sp);
    if (J.useLists)
        ORBM_LAUNCH(k_stereo_match_one<true>, dim3((J.left.n + 3) / 4, 1), dim3(256), 0, m->stream, J, sp);
    else
        ORBM_LAUNCH(k_stereo_match_one<false>, dim3((J.left.n + 3) / 4, 1), dim3(256), 0, m->stream, J, sp);
    ORBM_LAUNCH(k_stereo_median_one, dim3(1), dim3(1024), 0, m->stream, J);
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, u_right, (const float*)J.uRight, (size_t)J.left.n, mem_space))) return rc;
    if ((rc = copy_back(m, depth, (const float*)J.depth, (size_t)J.left.n, mem_space))) return rc;
    if (n_stereo && (rc = d2h(m, n_stereo, dKept, sizeof(int)))) return rc;
    if (mem_space == ORBX_MEM_HOST || n_stereo) ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

int orbm_compute_stereo_matches_batch(orbm_context* m, orbm_stereo_job* jobsIn, int njobs, const float* scale_factors,
                                      const float* inv_scale_factors, int nlevels, float mb, float mbf) {
    int rc = use_device(m);
    if (rc) return rc;
    if (njobs < 0 || (njobs > 0 && !jobsIn) || !scale_factors || !inv_scale_factors || nlevels < 1 || nlevels > MAX_LEVELS) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (njobs == 0) return ORBX_OK;
    StereoParams sp;
    int bandMax;
    if ((rc = stereo_params(scale_factors, inv_scale_factors, nlevels, mb, mbf, &sp, &bandMax))) return rc;
    size_t rowsTotal = 0, sadTotal = 0, startTotal = 0;
    int maxL = 0;
    for (int i = 0; i < njobs; i++) {
        const orbm_stereo_job& P = jobsIn[i];
        if (P.left.n < 0 || P.right.n < 0 || P.left.n > 65000 || P.right.n > 65000 ||
            (P.left.n > 0 && (!P.u_right || !P.depth)) || P.pyr_left.nlevels != nlevels || P.pyr_right.nlevels != nlevels ||
            P.pyr_left.height[0] < 1 || P.pyr_left.height[0] > STEREO_MAX_ROWS) {
            orbx_set_error("stereo job %d: invalid argument", i);
            return ORBX_ERR_INVALID;
        }
        rowsTotal += (size_t)P.right.n;
        sadTotal += (size_t)P.left.n;
        startTotal += (size_t)P.pyr_left.height[0] + 1;
        maxL = std::max(maxL, P.left.n);
    }
    void *dJobs, *dRows, *dSad, *dKept, *dStart, *dItems;
    if ((rc = pool_get(m, SL_JB_JOBS, (size_t)njobs * sizeof(StereoJob), &dJobs)) ||
        (rc = pool_get(m, SL_ST_ROWS, std::max(rowsTotal, (size_t)1) * sizeof(int4), &dRows)) ||
        (rc = pool_get(m, SL_ST_SAD, std::max(sadTotal, (size_t)1) * sizeof(int), &dSad)) ||
        (rc = pool_get(m, SL_JB_TOT, (size_t)njobs * sizeof(int), &dKept)) ||
        (rc = pool_get(m, SL_ST_START, startTotal * sizeof(int), &dStart)) ||
        (rc = pool_get(m, SL_ST_ITEMS, std::max(rowsTotal, (size_t)1) * bandMax * sizeof(int), &dItems)))
        return rc;
    StereoJob* jobs = (StereoJob*)arena_alloc(m, (size_t)njobs * sizeof(StereoJob));
    if (!jobs) return ORBX_ERR_HIP;
    size_t ro = 0, so = 0, sto = 0;
    for (int i = 0; i < njobs; i++) {
        const orbm_stereo_job& P = jobsIn[i];
        StereoJob& J = jobs[i];
        memset(&J, 0, sizeof(J));
        if ((rc = fill_pyramid(&P.pyr_left, &J.pl, "left"))) return rc;
        if ((rc = fill_pyramid(&P.pyr_right, &J.pr, "right"))) return rc;
        if ((rc = make_frame(m, &P.left, 0, 0, ORBX_MEM_DEVICE, &J.left))) return rc;
        if ((rc = make_frame(m, &P.right, 0, 0, ORBX_MEM_DEVICE, &J.right))) return rc;
        J.nRows = J.pl.h[0];  // reference src/Frame.cc:610
        J.rows = (int4*)dRows + ro;
        J.rowStart = (int*)dStart + sto;
        J.rowItems = (int*)dItems + ro * (size_t)bandMax;
        J.sad = (int*)dSad + so;
        J.uRight = P.u_right;
        J.depth = P.depth;
        J.kept = (int*)dKept + i;
        J.useLists = 1;
        ro += (size_t)P.right.n;
        so += (size_t)P.left.n;
        sto += (size_t)J.nRows + 1;
    }
    ORBX_HIP_CHECK(hipMemcpyAsync(dJobs, jobs, (size_t)njobs * sizeof(StereoJob), hipMemcpyHostToDevice, m->stream));
    if ((rc = run_stereo_jobs(m, (const StereoJob*)dJobs, njobs, maxL, sp))) return rc;
    if ((int)m->totalsHost.size() < njobs) m->totalsHost.resize(njobs);
    if ((rc = d2h(m, m->totalsHost.data(), dKept, (size_t)njobs * sizeof(int)))) return rc;
    ORBX_HIP_CHECK(sync_deliver(m));
    for (int i = 0; i < njobs; i++) jobsIn[i].n_stereo = m->totalsHost[i];
    return ORBX_OK;
}

// Start state of the vocabulary-guided searches in one launch: match tables and bins at -1, histogram at 0.
__global__ void k_bow_init(int* a, int na, int* b, int nb, int* c, int nc, int* zero, int nz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (a && i < na) a[i] = -1;
    if (b && i < nb) b[i] = -1;
    if (c && i < nc) c[i] = -1;
    if (i < nz) zero[i] = 0;
}

static int make_featvec(orbm_context* m, const orbm_feature_vector* fv, int slotNodes, int slotStart, int slotItems,
                        int memSpace, DevFeatVec* out) {
    if (!fv || fv->n_nodes < 0 || fv->n_items < 0 || (fv->n_nodes > 0 && (!fv->node_ids || !fv->start)) ||
        (fv->n_items > 0 && !fv->items)) {
        orbx_set_error("invalid feature vector");
        return ORBX_ERR_INVALID;
    }
    int rc;
    if ((rc = stage_in(m, slotNodes, fv->node_ids, (size_t)fv->n_nodes, memSpace, &out->nodeIds))) return rc;
    if ((rc = stage_in(m, slotStart, (const int*)fv->start, (size_t)fv->n_nodes + 1, memSpace, &out->start))) return rc;
    if ((rc = stage_in(m, slotItems, fv->items, (size_t)fv->n_items, memSpace, &out->items))) return rc;
    out->nNodes = fv->n_nodes;
    return ORBX_OK;
}

// host-side checks on a host-memory feature vector: ascending ids, monotone offsets, items in range,
// and every keypoint in at most one node (DBoW2 files each feature under exactly one node; the
// kernels run the nodes concurrently and rely on their keypoint sets being disjoint)
static int check_featvec_host(const orbm_feature_vector* fv, int nKeypoints) {
    for (int i = 0; i < fv->n_nodes; i++) {
        if (i > 0 && fv->node_ids[i] <= fv->node_ids[i - 1]) {
            orbx_set_error("feature vector: node ids must be strictly ascending");
            return ORBX_ERR_INVALID;
        }
        if (fv->start[i] < 0 || fv->start[i + 1] < fv->start[i] || fv->start[i + 1] > fv->n_items) {
            orbx_set_error("feature vector: bad offsets at node %d", i);
            return ORBX_ERR_INVALID;
        }
    }
    std::vector<bool> seen((size_t)std::max(nKeypoints, 1), false);
    for (int i = 0; i < fv->n_items; i++) {
        if (fv->items[i] >= (uint32_t)nKeypoints) {
            orbx_set_error("feature vector: item %d names keypoint %u of %d", i, fv->items[i], nKeypoints);
            return ORBX_ERR_INVALID;
        }
        if (seen[fv->items[i]]) {
            orbx_set_error("feature vector: keypoint %u occurs more than once", fv->items[i]);
            return ORBX_ERR_INVALID;
        }
        seen[fv->items[i]] = true;
    }
    return ORBX_OK;
}

int orbm_search_by_bow(orbm_context* m, const orbm_frame* f1, const orbm_feature_vector* fv1, const uint8_t* valid1,
                       const orbm_frame* f2, const orbm_feature_vector* fv2, const uint8_t* valid2, int max_dist,
                       float nn_ratio, int check_orientation, int32_t* match12, int32_t* match21, int* nmatches,
                       int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (!valid1 || !match12 || !match21 || !nmatches) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    DevFrame d1, d2;
    if ((rc = make_frame(m, f1, SL_F1_KPS, SL_F1_DESC, mem_space, &d1))) return rc;
    if ((rc = make_frame(m, f2, SL_F2_KPS, SL_F2_DESC, mem_space, &d2))) return rc;
    if (mem_space == ORBX_MEM_HOST) {
        if ((rc = check_featvec_host(fv1, d1.n)) || (rc = check_featvec_host(fv2, d2.n))) return rc;
    }
    DevFeatVec v1, v2;
    if ((rc = make_featvec(m, fv1, SL_FV1_NODES, SL_FV1_START, SL_FV1_ITEMS, mem_space, &v1))) return rc;
    if ((rc = make_featvec(m, fv2, SL_FV2_NODES, SL_FV2_START, SL_FV2_ITEMS, mem_space, &v2))) return rc;
    const u8 *dValid1, *dValid2;
    if ((rc = stage_in(m, SL_BW_VALID1, valid1, (size_t)d1.n, mem_space, &dValid1))) return rc;
    if ((rc = stage_in(m, SL_BW_VALID2, valid2, (size_t)d2.n, mem_space, &dValid2))) return rc;
    int *dM12, *dM21;
    void *dBin, *dHist;
    if ((rc = stage_out(m, SL_BW_M12, (int*)match12, (size_t)d1.n, mem_space, &dM12))) return rc;
    if ((rc = stage_out(m, SL_BW_M21, (int*)match21, (size_t)d2.n, mem_space, &dM21))) return rc;
    if ((rc = pool_get(m, SL_BW_BIN, (size_t)std::max(d1.n, 1) * sizeof(int), &dBin))) return rc;
    if ((rc = pool_get(m, SL_BW_HIST, (HISTO_LENGTH + 2) * sizeof(int), &dHist))) return rc;
    {
        const int most = std::max(std::max(d1.n, d2.n), HISTO_LENGTH + 2);
        ORBM_LAUNCH(k_bow_init, dim3((most + 255) / 256), dim3(256), 0, m->stream, dM12, d1.n, dM21, d2.n, (int*)dBin, d1.n,
                    (int*)dHist, HISTO_LENGTH + 2);
    }
    if (v1.nNodes > 0 && v2.nNodes > 0 && d1.n > 0 && d2.n > 0) {
        ORBM_LAUNCH(k_bow_match, dim3(v1.nNodes), dim3(64), 0, m->stream, d1, v1, dValid1, d2, v2, dValid2, max_dist,
                           nn_ratio, check_orientation, dM12, dM21, (int*)dBin, (int*)dHist);
        ORBX_HIP_CHECK(hipGetLastError());
    }
    int* dCount = (int*)dHist + HISTO_LENGTH;
    ORBM_LAUNCH(k_bow_finish, dim3(1), dim3(256), 0, m->stream, d1.n, check_orientation, dM12, dM21, (const int*)dBin,
                       (const int*)dHist, dCount);
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, (int*)match12, (const int*)dM12, (size_t)d1.n, mem_space))) return rc;
    if ((rc = copy_back(m, (int*)match21, (const int*)dM21, (size_t)d2.n, mem_space))) return rc;
    if ((rc = d2h(m, nmatches, dCount, sizeof(int)))) return rc;
    ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

int orbm_search_by_bow_batch(orbm_context* m, orbm_bow_job* jobsIn, int njobs, int max_dist, float nn_ratio,
                             int check_orientation) {
    int rc = use_device(m);
    if (rc) return rc;
    if (njobs < 0 || (njobs > 0 && !jobsIn)) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (njobs == 0) return ORBX_OK;
    size_t words = 0;
    int maxNodes = 0, maxN = HISTO_LENGTH + 2;
    for (int i = 0; i < njobs; i++) {
        const orbm_bow_job& P = jobsIn[i];
        const bool fvBad = P.fv1.n_nodes < 0 || P.fv2.n_nodes < 0 || P.fv1.n_items < 0 || P.fv2.n_items < 0 ||
                           (P.fv1.n_nodes > 0 && (!P.fv1.node_ids || !P.fv1.start)) ||
                           (P.fv2.n_nodes > 0 && (!P.fv2.node_ids || !P.fv2.start)) || (P.fv1.n_items > 0 && !P.fv1.items) ||
                           (P.fv2.n_items > 0 && !P.fv2.items);
        if (P.f1.n < 0 || P.f2.n < 0 || fvBad || (P.f1.n > 0 && (!P.valid1 || !P.match12 || !P.f1.kps || !P.f1.desc)) ||
            (P.f2.n > 0 && (!P.match21 || !P.f2.kps || !P.f2.desc))) {
            orbx_set_error("job %d: invalid argument", i);
            return ORBX_ERR_INVALID;
        }
        words += (size_t)std::max(P.f1.n, 1) + HISTO_LENGTH + 2;
        maxNodes = std::max(maxNodes, P.fv1.n_nodes);
        maxN = std::max(maxN, std::max(P.f1.n, P.f2.n));
    }
    void *dJobs, *dWords, *dCounts;
    if ((rc = pool_get(m, SL_JB_JOBS, (size_t)njobs * sizeof(BowJob), &dJobs)) || (rc = pool_get(m, SL_JB_I32, words * 4, &dWords)) ||
        (rc = pool_get(m, SL_JB_TOT, (size_t)njobs * sizeof(int), &dCounts)))
        return rc;
    BowJob* jobs = (BowJob*)arena_alloc(m, (size_t)njobs * sizeof(BowJob));
    if (!jobs) return ORBX_ERR_HIP;
    int* wp = (int*)dWords;
    for (int i = 0; i < njobs; i++) {
        const orbm_bow_job& P = jobsIn[i];
        BowJob& J = jobs[i];
        memset(&J, 0, sizeof(J));
        if ((rc = make_frame(m, &P.f1, 0, 0, ORBX_MEM_DEVICE, &J.f1)) || (rc = make_frame(m, &P.f2, 0, 0, ORBX_MEM_DEVICE, &J.f2)))
            return rc;
        J.v1.nNodes = P.fv1.n_nodes;
        J.v1.nodeIds = P.fv1.node_ids;
        J.v1.start = P.fv1.start;
        J.v1.items = P.fv1.items;
        J.v2.nNodes = P.fv2.n_nodes;
        J.v2.nodeIds = P.fv2.node_ids;
        J.v2.start = P.fv2.start;
        J.v2.items = P.fv2.items;
        J.valid1 = P.valid1;
        J.valid2 = P.valid2;
        J.match12 = P.match12;
        J.match21 = P.match21;
        J.bin12 = wp;
        wp += std::max(P.f1.n, 1);
        J.hist = wp;
        wp += HISTO_LENGTH + 2;
    }
    ORBX_HIP_CHECK(hipMemcpyAsync(dJobs, jobs, (size_t)njobs * sizeof(BowJob), hipMemcpyHostToDevice, m->stream));
    const BowJob* dJ = (const BowJob*)dJobs;
    ORBM_LAUNCH(k_bow_init_jobs, dim3((maxN + 255) / 256, njobs), dim3(256), 0, m->stream, dJ);
    if (maxNodes > 0)
        ORBM_LAUNCH(k_bow_match_jobs, dim3(maxNodes, njobs), dim3(64), 0, m->stream, dJ, max_dist, nn_ratio, check_orientation);
    ORBM_LAUNCH(k_bow_finish_jobs, dim3(njobs), dim3(256), 0, m->stream, dJ, check_orientation, (int*)dCounts);
    ORBX_HIP_CHECK(hipGetLastError());
    m->totalsHost.resize((size_t)njobs);
    if ((rc = d2h(m, m->totalsHost.data(), dCounts, (size_t)njobs * sizeof(int)))) return rc;
    ORBX_HIP_CHECK(sync_deliver(m));
    for (int i = 0; i < njobs; i++) jobsIn[i].nmatches = m->totalsHost[i];
    return ORBX_OK;
}

int orbm_search_for_triangulation(orbm_context* m, const orbm_frame* f1, const orbm_feature_vector* fv1,
                                  const uint8_t* has_mp1, const float* u_right1, const orbm_frame* f2,
                                  const orbm_feature_vector* fv2, const uint8_t* has_mp2, const float* u_right2,
                                  const float* f12, float ex, float ey, const float* scale_factors2,
                                  const float* level_sigma2_2, int nlevels, int only_stereo, int check_orientation,
                                  int32_t* match12, int* nmatches, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (!has_mp1 || !has_mp2 || !u_right1 || !u_right2 || !f12 || !scale_factors2 || !level_sigma2_2 || !match12 ||
        !nmatches || nlevels < 1 || nlevels > ORBM_MAX_LEVELS) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    DevFrame d1, d2;
    if ((rc = make_frame(m, f1, SL_F1_KPS, SL_F1_DESC, mem_space, &d1))) return rc;
    if ((rc = make_frame(m, f2, SL_F2_KPS, SL_F2_DESC, mem_space, &d2))) return rc;
    if (mem_space == ORBX_MEM_HOST) {
        if ((rc = check_featvec_host(fv1, d1.n)) || (rc = check_featvec_host(fv2, d2.n))) return rc;
    }
    DevFeatVec v1, v2;
    if ((rc = make_featvec(m, fv1, SL_FV1_NODES, SL_FV1_START, SL_FV1_ITEMS, mem_space, &v1))) return rc;
    if ((rc = make_featvec(m, fv2, SL_FV2_NODES, SL_FV2_START, SL_FV2_ITEMS, mem_space, &v2))) return rc;
    const u8 *dMp1, *dMp2;
    const float *dUr1, *dUr2;
    if ((rc = stage_in(m, SL_BW_VALID1, has_mp1, (size_t)d1.n, mem_space, &dMp1))) return rc;
    if ((rc = stage_in(m, SL_BW_VALID2, has_mp2, (size_t)d2.n, mem_space, &dMp2))) return rc;
    if ((rc = stage_in(m, SL_BW_UR1, u_right1, (size_t)d1.n, mem_space, &dUr1))) return rc;
    if ((rc = stage_in(m, SL_BW_UR2, u_right2, (size_t)d2.n, mem_space, &dUr2))) return rc;
    TriParams P;
    memset(&P, 0, sizeof(P));
    for (int i = 0; i < 9; i++) P.F12[i] = f12[i];
    P.ex = ex;
    P.ey = ey;
    for (int l = 0; l < nlevels; l++) {
        P.scale2[l] = scale_factors2[l];
        P.sigma2[l] = level_sigma2_2[l];
    }
    P.onlyStereo = only_stereo;
    P.checkOri = check_orientation;
    int* dM12;
    void *dBin, *dHist;
    if ((rc = stage_out(m, SL_BW_M12, (int*)match12, (size_t)d1.n, mem_space, &dM12))) return rc;
    if ((rc = pool_get(m, SL_BW_BIN, (size_t)std::max(d1.n, 1) * sizeof(int), &dBin))) return rc;
    if ((rc = pool_get(m, SL_BW_HIST, (HISTO_LENGTH + 2) * sizeof(int), &dHist))) return rc;
    {
        const int most = std::max(d1.n, HISTO_LENGTH + 2);
        ORBM_LAUNCH(k_bow_init, dim3((most + 255) / 256), dim3(256), 0, m->stream, dM12, d1.n, (int*)nullptr, 0, (int*)dBin, d1.n,
                    (int*)dHist, HISTO_LENGTH + 2);
    }
    if (v1.nNodes > 0 && v2.nNodes > 0 && d1.n > 0 && d2.n > 0) {
        ORBM_LAUNCH(k_tri_match, dim3(v1.nNodes), dim3(64), 0, m->stream, d1, v1, dMp1, dUr1, d2, v2, dMp2, dUr2, P,
                           dM12, (int*)dBin, (int*)dHist);
        ORBX_HIP_CHECK(hipGetLastError());
    }
    int* dCount = (int*)dHist + HISTO_LENGTH;
    ORBM_LAUNCH(k_bow_finish, dim3(1), dim3(256), 0, m->stream, d1.n, check_orientation, dM12, (int*)nullptr,
                       (const int*)dBin, (const int*)dHist, dCount);
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, (int*)match12, (const int*)dM12, (size_t)d1.n, mem_space))) return rc;
    if ((rc = d2h(m, nmatches, dCount, sizeof(int)))) return rc;
    ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

// Enqueues one window-best search; outputs stay on the device.
static int run_window_best(orbm_context* m, const DevFrame& kf, int n, const u8* dValid, const float* dU, const float* dV,
                           const float* dR, const int* dLevel, const u8* dDesc, int chi2Gate, const float* dKfUr,
                           const float* dQUr, const float* invSigma2, int nlevels, int* dIdx, int* dDist) {
    if (n == 0) return ORBX_OK;
    int *cs, *ci;
    int rc = build_grid(m, kf, &cs, &ci);
    if (rc) return rc;
    WindowBestParams P;
    memset(&P, 0, sizeof(P));
    P.chi2Gate = chi2Gate;
    if (chi2Gate)
        for (int l = 0; l < nlevels; l++) P.invSigma2[l] = invSigma2[l];
    ORBM_LAUNCH(k_window_best, dim3((n + 3) / 4), dim3(256), 0, m->stream, kf, (const int*)cs, (const int*)ci, n, dValid,
                       dU, dV, dR, dLevel, dDesc, dKfUr, dQUr, P, dIdx, dDist);
    ORBX_HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

int orbm_window_best(orbm_context* m, const orbm_frame* kf, int n, const uint8_t* valid, const float* u, const float* v,
                     const float* radius, const int32_t* level, const uint8_t* desc, int chi2_gate,
                     const float* kf_u_right, const float* q_u_right, const float* inv_level_sigma2, int nlevels,
                     int32_t* best_idx, int32_t* best_dist, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (n < 0 || !valid || !u || !v || !radius || !level || !desc || !best_idx || !best_dist ||
        (chi2_gate && (!inv_level_sigma2 || nlevels < 1 || nlevels > ORBM_MAX_LEVELS))) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (n == 0) return ORBX_OK;
    DevFrame dk;
    if ((rc = make_frame(m, kf, SL_F2_KPS, SL_F2_DESC, mem_space, &dk))) return rc;
    const u8 *dValid, *dDesc;
    const float *dU, *dV, *dR, *dKfUr, *dQUr;
    const int* dLevel;
    if ((rc = stage_in(m, SL_VALID, valid, (size_t)n, mem_space, &dValid))) return rc;
    if ((rc = stage_in(m, SL_U, u, (size_t)n, mem_space, &dU))) return rc;
    if ((rc = stage_in(m, SL_V, v, (size_t)n, mem_space, &dV))) return rc;
    if ((rc = stage_in(m, SL_QR, radius, (size_t)n, mem_space, &dR))) return rc;
    if ((rc = stage_in(m, SL_WB_LEVEL, (const int*)level, (size_t)n, mem_space, &dLevel))) return rc;
    if ((rc = stage_in(m, SL_QDESC, desc, (size_t)n * 32, mem_space, &dDesc))) return rc;
    if ((rc = stage_in(m, SL_URIGHT, kf_u_right, (size_t)dk.n, mem_space, &dKfUr))) return rc;
    if ((rc = stage_in(m, SL_WB_QUR, q_u_right, (size_t)n, mem_space, &dQUr))) return rc;
    int *dIdx, *dDist;
    if ((rc = stage_out(m, SL_WB_IDX, (int*)best_idx, (size_t)n, mem_space, &dIdx))) return rc;
    if ((rc = stage_out(m, SL_WB_DIST, (int*)best_dist, (size_t)n, mem_space, &dDist))) return rc;
    if ((rc = run_window_best(m, dk, n, dValid, dU, dV, dR, dLevel, dDesc, chi2_gate, dKfUr, dQUr, inv_level_sigma2, nlevels,
                              dIdx, dDist)))
        return rc;
    if ((rc = copy_back(m, (int*)best_idx, (const int*)dIdx, (size_t)n, mem_space))) return rc;
    if ((rc = copy_back(m, (int*)best_dist, (const int*)dDist, (size_t)n, mem_space))) return rc;
    if (mem_space == ORBX_MEM_HOST) ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

int orbm_search_by_sim3(orbm_context* m, const orbm_frame* kf1, const orbm_frame* kf2, const uint8_t* valid12,
                        const float* u12, const float* v12, const float* r12, const int32_t* level12,
                        const uint8_t* mp_desc1, const uint8_t* valid21, const float* u21, const float* v21,
                        const float* r21, const int32_t* level21, const uint8_t* mp_desc2, int32_t* match12, int* nfound,
                        int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (!valid12 || !u12 || !v12 || !r12 || !level12 || !mp_desc1 || !valid21 || !u21 || !v21 || !r21 || !level21 ||
        !mp_desc2 || !match12 || !nfound) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    DevFrame d1, d2;
    if ((rc = make_frame(m, kf1, SL_F1_KPS, SL_F1_DESC, mem_space, &d1))) return rc;
    if ((rc = make_frame(m, kf2, SL_F2_KPS, SL_F2_DESC, mem_space, &d2))) return rc;
    *nfound = 0;
    if (d1.n == 0) return ORBX_OK;
    void *i12, *dd12, *i21, *dd21, *dCount;
    if ((rc = pool_get(m, SL_WB_IDX, (size_t)d1.n * sizeof(int), &i12))) return rc;
    if ((rc = pool_get(m, SL_WB_DIST, (size_t)d1.n * sizeof(int), &dd12))) return rc;
    if ((rc = pool_get(m, SL_WB_IDX2, (size_t)std::max(d2.n, 1) * sizeof(int), &i21))) return rc;
    if ((rc = pool_get(m, SL_WB_DIST2, (size_t)std::max(d2.n, 1) * sizeof(int), &dd21))) return rc;
    if ((rc = pool_get(m, SL_SCALAR, 64, &dCount))) return rc;
    // direction 1 -> 2: the map points of keyframe 1 searched among the keypoints of keyframe 2
    {
        const u8 *dValid, *dDesc;
        const float *dU, *dV, *dR;
        const int* dLevel;
        if ((rc = stage_in(m, SL_VALID, valid12, (size_t)d1.n, mem_space, &dValid))) return rc;
        if ((rc = stage_in(m, SL_U, u12, (size_t)d1.n, mem_space, &dU))) return rc;
        if ((rc = stage_in(m, SL_V, v12, (size_t)d1.n, mem_space, &dV))) return rc;
        if ((rc = stage_in(m, SL_QR, r12, (size_t)d1.n, mem_space, &dR))) return rc;
        if ((rc = stage_in(m, SL_WB_LEVEL, (const int*)level12, (size_t)d1.n, mem_space, &dLevel))) return rc;
        if ((rc = stage_in(m, SL_QDESC, mp_desc1, (size_t)d1.n * 32, mem_space, &dDesc))) return rc;
        if ((rc = run_window_best(m, d2, d1.n, dValid, dU, dV, dR, dLevel, dDesc, 0, nullptr, nullptr, nullptr, 0, (int*)i12,
                                  (int*)dd12)))
            return rc;
    }
    // direction 2 -> 1 (the staging blocks are reused: stream order keeps the first search ahead of the copies)
    if (d2.n > 0) {
        const u8 *dValid, *dDesc;
        const float *dU, *dV, *dR;
        const int* dLevel;
        if ((rc = stage_in(m, SL_BW_VALID2, valid21, (size_t)d2.n, mem_space, &dValid))) return rc;
        if ((rc = stage_in(m, SL_BW_UR1, u21, (size_t)d2.n, mem_space, &dU))) return rc;
        if ((rc = stage_in(m, SL_BW_UR2, v21, (size_t)d2.n, mem_space, &dV))) return rc;
        if ((rc = stage_in(m, SL_WB_QUR, r21, (size_t)d2.n, mem_space, &dR))) return rc;
        if ((rc = stage_in(m, SL_BW_BIN, (const int*)level21, (size_t)d2.n, mem_space, &dLevel))) return rc;
        if ((rc = stage_in(m, SL_KNN_Q, mp_desc2, (size_t)d2.n * 32, mem_space, &dDesc))) return rc;
        if ((rc = run_window_best(m, d1, d2.n, dValid, dU, dV, dR, dLevel, dDesc, 0, nullptr, nullptr, nullptr, 0, (int*)i21,
                                  (int*)dd21)))
            return rc;
    }
    int* dM12;
    if ((rc = stage_out(m, SL_BW_M12, (int*)match12, (size_t)d1.n, mem_space, &dM12))) return rc;
    ORBM_LAUNCH(k_sim3_agree, dim3(1), dim3(256), 0, m->stream, d1.n, d2.n, (const int*)i12, (const int*)dd12,
                       (const int*)i21, (const int*)dd21, dM12, (int*)dCount);
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, (int*)match12, (const int*)dM12, (size_t)d1.n, mem_space))) return rc;
    if ((rc = d2h(m, nfound, dCount, sizeof(int)))) return rc;
    ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

int orbm_distinctive_descriptors(orbm_context* m, const uint8_t* desc, const int32_t* group_start, int ngroups,
                                 int32_t* best, int32_t* best_median, uint8_t* out_desc, int total_desc, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (ngroups < 0 || total_desc < 0 || !group_start || !best || (total_desc > 0 && !desc)) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (ngroups == 0) return ORBX_OK;
    if (mem_space == ORBX_MEM_HOST) {
        for (int g = 0; g < ngroups; g++) {
            const int n = group_start[g + 1] - group_start[g];
            if (group_start[g] < 0 || n < 0 || group_start[g + 1] > total_desc) {
                orbx_set_error("group %d has bad bounds", g);
                return ORBX_ERR_INVALID;
            }
            if (n > DD_MAX_GROUP) {
                orbx_set_error("group %d has %d descriptors; limit is %d", g, n, DD_MAX_GROUP);
                return ORBX_ERR_UNSUPPORTED;
            }
        }
    }
    const u8* dDesc;
    const int* dStart;
    if ((rc = stage_in(m, SL_DD_DESC, desc, (size_t)total_desc * 32, mem_space, &dDesc))) return rc;
    if ((rc = stage_in(m, SL_DD_START, (const int*)group_start, (size_t)ngroups + 1, mem_space, &dStart))) return rc;
    int *dBest, *dMed = nullptr;
    u8* dOut = nullptr;
    if ((rc = stage_out(m, SL_DD_BEST, (int*)best, (size_t)ngroups, mem_space, &dBest))) return rc;
    if (best_median && (rc = stage_out(m, SL_DD_MED, (int*)best_median, (size_t)ngroups, mem_space, &dMed))) return rc;
    if (out_desc && (rc = stage_out(m, SL_DD_OUT, out_desc, (size_t)ngroups * 32, mem_space, &dOut))) return rc;
    const int ldsBytes = 4 * DD_MAX_GROUP * (int)sizeof(u16);
    if (!m->distinctiveReady) {
        // per context (function attributes are per device; every context sets the same constant)
        ORBX_HIP_CHECK(hipFuncSetAttribute((const void*)k_distinctive, hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes));
        m->distinctiveReady = true;
    }
    ORBM_LAUNCH(k_distinctive, dim3((ngroups + 3) / 4), dim3(256), ldsBytes, m->stream, dDesc, dStart, ngroups, dBest,
                       dMed, dOut);
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, (int*)best, (const int*)dBest, (size_t)ngroups, mem_space))) return rc;
    if (best_median && (rc = copy_back(m, (int*)best_median, (const int*)dMed, (size_t)ngroups, mem_space))) return rc;
    if (out_desc && (rc = copy_back(m, out_desc, (const u8*)dOut, (size_t)ngroups * 32, mem_space))) return rc;
    if (mem_space == ORBX_MEM_HOST) ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

int orbm_filter_by_mask(orbm_context* m, const orbx_keypoint* kps, const uint8_t* desc, int n, const uint8_t* mask, int w,
                        int h, size_t stride, orbx_keypoint* out_kps, uint8_t* out_desc, int* n_out, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (n < 0 || w < 1 || h < 1 || stride < (size_t)w || !mask || !n_out || (n > 0 && (!kps || !desc || !out_kps || !out_desc))) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    const orbx_keypoint* dKps;
    const u8 *dDesc, *dMask;
    if ((rc = stage_in(m, SL_F1_KPS, kps, (size_t)n, mem_space, &dKps))) return rc;
    if ((rc = stage_in(m, SL_F1_DESC, desc, (size_t)n * 32, mem_space, &dDesc))) return rc;
    if ((rc = stage_in(m, SL_MK_MASK, mask, stride * (size_t)h, mem_space, &dMask))) return rc;
    orbx_keypoint* dOutK;
    u8* dOutD;
    if ((rc = stage_out(m, SL_MK_KPS, out_kps, (size_t)n, mem_space, &dOutK))) return rc;
    if ((rc = stage_out(m, SL_MK_DESC, out_desc, (size_t)n * 32, mem_space, &dOutD))) return rc;
    void* dScal;
    if ((rc = pool_get(m, SL_SCALAR, 64, &dScal))) return rc;
    ORBX_HIP_CHECK(hipMemsetAsync(dScal, 0, 64, m->stream));
    ORBM_LAUNCH(k_mask_sum, dim3(std::min(h, 1024)), dim3(256), 0, m->stream, dMask, w, h, (i64)stride, (u64*)dScal);
    ORBX_HIP_CHECK(hipGetLastError());
    ORBM_LAUNCH(k_mask_filter, dim3(1), dim3(1024), 0, m->stream, dKps, dDesc, n, dMask, w, h, (i64)stride,
                       (const u64*)dScal, dOutK, dOutD, (int*)dScal + 4);
    ORBX_HIP_CHECK(hipGetLastError());
    ORBX_HIP_CHECK(hipMemcpyAsync(n_out, (int*)dScal + 4, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    ORBX_HIP_CHECK(sync_deliver(m));
    if (mem_space == ORBX_MEM_HOST && *n_out > 0) {
        ORBX_HIP_CHECK(hipMemcpy(out_kps, dOutK, (size_t)*n_out * sizeof(orbx_keypoint), hipMemcpyDeviceToHost));
        ORBX_HIP_CHECK(hipMemcpy(out_desc, dOutD, (size_t)*n_out * 32, hipMemcpyDeviceToHost));
    }
    return ORBX_OK;
}

int orbm_pack_keyframe_records(orbm_context* m, const orbx_keypoint* kps, const uint8_t* desc, const uint64_t* mp_index, int n,
                               uint8_t* records, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!kps || !desc || !records))) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (n == 0) return ORBX_OK;
    const orbx_keypoint* dKps;
    const u8* dDesc;
    const uint64_t* dIdx;
    u8* dRec;
    if ((rc = stage_in(m, SL_F1_KPS, kps, (size_t)n, mem_space, &dKps))) return rc;
    if ((rc = stage_in(m, SL_F1_DESC, desc, (size_t)n * 32, mem_space, &dDesc))) return rc;
    if ((rc = stage_in(m, SL_RC_IDX, mp_index, (size_t)n, mem_space, &dIdx))) return rc;
    if ((rc = stage_out(m, SL_RC_REC, records, (size_t)n * 64, mem_space, &dRec))) return rc;
    ORBM_LAUNCH(k_records, dim3(((size_t)n * 16 + 255) / 256), dim3(256), 0, m->stream, 1, n, (orbx_keypoint*)dKps,
                       (u8*)dDesc, (u64*)dIdx, dRec);
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, records, (const u8*)dRec, (size_t)n * 64, mem_space))) return rc;
    if (mem_space == ORBX_MEM_HOST) ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

int orbm_unpack_keyframe_records(orbm_context* m, const uint8_t* records, int n, orbx_keypoint* kps, uint8_t* desc,
                                 uint64_t* mp_index, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!kps || !desc || !records))) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (n == 0) return ORBX_OK;
    const u8* dRec;
    orbx_keypoint* dKps;
    u8* dDesc;
    uint64_t* dIdx = nullptr;
    if ((rc = stage_in(m, SL_RC_REC, records, (size_t)n * 64, mem_space, &dRec))) return rc;
    if ((rc = stage_out(m, SL_F1_KPS, kps, (size_t)n, mem_space, &dKps))) return rc;
    if ((rc = stage_out(m, SL_F1_DESC, desc, (size_t)n * 32, mem_space, &dDesc))) return rc;
    if (mp_index && (rc = stage_out(m, SL_RC_IDX, mp_index, (size_t)n, mem_space, &dIdx))) return rc;
    ORBM_LAUNCH(k_records, dim3(((size_t)n * 16 + 255) / 256), dim3(256), 0, m->stream, 0, n, dKps, dDesc, (u64*)dIdx,
                       (u8*)dRec);
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, kps, (const orbx_keypoint*)dKps, (size_t)n, mem_space))) return rc;
    if ((rc = copy_back(m, desc, (const u8*)dDesc, (size_t)n * 32, mem_space))) return rc;
    if (mp_index && (rc = copy_back(m, mp_index, (const uint64_t*)dIdx, (size_t)n, mem_space))) return rc;
    if (mem_space == ORBX_MEM_HOST) ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

// How one 2-NN scan is laid out on the GPU.  Two forms of the same scan: on the matrix cores (an
// exact fp4 contraction) when the problem is large enough to pay for the expansion of bits to
// nibbles, on the vector ALU (xor + popcount) otherwise.  With too few queries to fill the GPU the
// train rows are split over blockIdx.y (slices > 1) and the partials are folded in order;
// maxPartEntries bounds slices * nq (the partial arrays' capacity), 0 = no bound.
struct KnnPlan {
    bool mfma;
    int blocks, slices, sliceRows;
};
static KnnPlan knn_plan(int path, int nq, int nt, size_t maxPartEntries, bool alwaysSliced) {
    KnnPlan p;
    p.mfma = path == 2 || (path == 0 && nq >= 512 && nt >= 512 && (i64)nq * nt >= ((i64)1 << 24));
    p.blocks = p.mfma ? (nq + 4 * 32 * KNN_MQ - 1) / (4 * 32 * KNN_MQ) : (nq + 256 * KNN_QPT - 1) / (256 * KNN_QPT);
    const int rowQuantum = p.mfma ? KNN_MROWS : 8;
    p.slices = 1;
    if (p.blocks < (p.mfma ? 512 : 128) && nt >= 128 && nt <= (1 << KNN_IDX_BITS)) {
        p.slices = std::min(nt / (p.mfma ? 2 * KNN_MROWS : 32), std::max((p.mfma ? 2048 : 1024) / p.blocks, 1));
        if (maxPartEntries) p.slices = (int)std::min<size_t>((size_t)p.slices, maxPartEntries / (size_t)nq);
        p.slices = std::max(p.slices, 1);
    }
    p.sliceRows = 0;
    if (p.slices > 1 || alwaysSliced) {
        p.sliceRows = std::max(((nt + p.slices - 1) / p.slices + rowQuantum - 1) / rowQuantum * rowQuantum, rowQuantum);
        p.slices = std::max((nt + p.sliceRows - 1) / p.sliceRows, 1);
    }
    return p;
}

int orbm_knn2(orbm_context* m, const uint8_t* q, int nq, const uint8_t* t, int nt, int train_index_base,
              int32_t* best_idx, int32_t* best_d, int32_t* second_d, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (nq < 0 || nt < 0 || !best_idx || !best_d || !second_d || (nq > 0 && !q) || (nt > 0 && !t)) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (nq == 0) return ORBX_OK;
    if (((uintptr_t)q | (uintptr_t)t) & 15u) {
        if (mem_space == ORBX_MEM_DEVICE) {
            orbx_set_error("device descriptor arrays must be 16-byte aligned");
            return ORBX_ERR_INVALID;
        }
    }
    const u8 *dq, *dt;
    if ((rc = stage_in(m, SL_KNN_Q, q, (size_t)nq * 32, mem_space, &dq))) return rc;
    if ((rc = stage_in(m, SL_KNN_T, t, (size_t)nt * 32, mem_space, &dt))) return rc;
    int *di, *dd, *ds;
    if ((rc = stage_out(m, SL_KNN_I, (int*)best_idx, (size_t)nq, mem_space, &di))) return rc;
    if ((rc = stage_out(m, SL_KNN_D, (int*)best_d, (size_t)nq, mem_space, &dd))) return rc;
    if ((rc = stage_out(m, SL_KNN_S, (int*)second_d, (size_t)nq, mem_space, &ds))) return rc;
    const int chunkRows = 1 << KNN_IDX_BITS;
    const KnnPlan p = knn_plan(m->knnPath, nq, nt, 0, false);
    if (p.slices > 1) {
        void *pi, *pd, *ps;
        const size_t partBytes = (size_t)p.slices * nq * sizeof(int);
        if ((rc = pool_get(m, SL_KNN_I2, partBytes, &pi)) || (rc = pool_get(m, SL_KNN_D2, partBytes, &pd)) ||
            (rc = pool_get(m, SL_KNN_S2, partBytes, &ps)))
            return rc;
        if (p.mfma)
            ORBM_LAUNCH(k_knn2_mfma, dim3(p.blocks, p.slices), dim3(256), 0, m->stream, (const uint4*)dq, nq, (const uint4*)dt,
                        nt, train_index_base, (int*)pi, (int*)pd, (int*)ps, 0, p.sliceRows);
        else
            ORBM_LAUNCH(k_knn2, dim3(p.blocks, p.slices), dim3(256), 0, m->stream, (const uint4*)dq, nq, (const uint4*)dt, nt,
                        train_index_base, (int*)pi, (int*)pd, (int*)ps, 0, p.sliceRows);
        ORBM_LAUNCH(k_knn2_merge_slices, dim3((nq + 255) / 256), dim3(256), 0, m->stream, nq, p.slices, (const int*)pi,
                    (const int*)pd, (const int*)ps, di, dd, ds);
    } else {
        // whole train set per block; sets beyond the key's index bits go chunk by chunk, merged into the output
        int c0 = 0;
        do {
            const int rows = std::min(chunkRows, nt - c0);
            if (p.mfma && rows > 0)
                ORBM_LAUNCH(k_knn2_mfma, dim3(p.blocks), dim3(256), 0, m->stream, (const uint4*)dq, nq,
                            (const uint4*)(dt + (size_t)c0 * 32), rows, train_index_base + c0, di, dd, ds, c0 > 0 ? 1 : 0, 0);
            else
                ORBM_LAUNCH(k_knn2, dim3(p.blocks), dim3(256), 0, m->stream, (const uint4*)dq, nq,
                            (const uint4*)(dt + (size_t)c0 * 32), rows, train_index_base + c0, di, dd, ds, c0 > 0 ? 1 : 0, 0);
            c0 += rows;
        } while (c0 < nt);
    }
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, (int*)best_idx, (const int*)di, (size_t)nq, mem_space))) return rc;
    if ((rc = copy_back(m, (int*)best_d, (const int*)dd, (size_t)nq, mem_space))) return rc;
    if ((rc = copy_back(m, (int*)second_d, (const int*)ds, (size_t)nq, mem_space))) return rc;
    if (mem_space == ORBX_MEM_HOST) ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

int orbm_debug_knn_path(orbm_context* m, int path) {
    if (!m || path < 0 || path > 2) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    m->knnPath = path;
    return ORBX_OK;
}

int orbm_knn2_noalloc(orbm_context* m, const uint8_t* q, int nq, const uint8_t* t, int nt, int train_index_base,
                      int32_t* best_idx, int32_t* best_d, int32_t* second_d) {
    // Nothing here allocates or frees: no arena, no pool, no staging.  The train rows are always
    // split over blockIdx.y; every slice writes its partials for all queries into the static
    // scratch array (each entry is written before it is read, so nothing needs clearing), and the
    // merge kernel folds them in slice order into the caller's arrays.
    if (!m) {
        orbx_set_error("null context");
        return ORBX_ERR_INVALID;
    }
    if (nq < 0 || nt < 0 || !best_idx || !best_d || !second_d || (nq > 0 && !q) || (nt > 0 && !t)) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (nq == 0) return ORBX_OK;
    if (nq > KNN_NA_MAX_PART) {
        orbx_set_error("orbm_knn2_noalloc: %d queries exceed the limit of %d", nq, KNN_NA_MAX_PART);
        return ORBX_ERR_CAPACITY;
    }
    if (nt > (1 << KNN_IDX_BITS)) {
        orbx_set_error("orbm_knn2_noalloc: %d train rows exceed the limit of %d", nt, 1 << KNN_IDX_BITS);
        return ORBX_ERR_CAPACITY;
    }
    if (((uintptr_t)q | (uintptr_t)t) & 15u) {
        orbx_set_error("device descriptor arrays must be 16-byte aligned");
        return ORBX_ERR_INVALID;
    }
    ORBX_HIP_CHECK(hipSetDevice(m->device));
    int* const base = m->naScratch;  // the device address, never the array's name
    if (!base) {
        orbx_set_error("scratch address missing");
        return ORBX_ERR_HIP;
    }
    int *pi = base, *pd = base + KNN_NA_MAX_PART, *ps = base + 2 * KNN_NA_MAX_PART;
    const KnnPlan p = knn_plan(m->knnPath, nq, nt, KNN_NA_MAX_PART, true);
    // Every context of this GPU shares the one array, and contexts default to streams of their own: the
    // calls are put in one order (the mutex) and each waits on the device for the one before it (the
    // event), so two matchers on two host threads cannot overwrite each other's partials.  A stream under
    // graph capture cannot wait for or record an event of the outside world: there the caller orders
    // the graph against other users of this entry point.
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool ordered = hipStreamIsCapturing(m->stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
    if (!ordered) (void)hipGetLastError();
    std::lock_guard<std::mutex> lock(g_scratchMu);
    if (ordered && g_scratchEvRecorded[m->device]) ORBX_HIP_CHECK(hipStreamWaitEvent(m->stream, g_scratchEv[m->device], 0));
    if (p.mfma && nt > 0)
        hipLaunchKernelGGL(k_knn2_mfma, dim3(p.blocks, p.slices), dim3(256), 0, m->stream, (const uint4*)q, nq, (const uint4*)t,
                           nt, train_index_base, pi, pd, ps, 0, p.sliceRows);
    else
        hipLaunchKernelGGL(k_knn2, dim3(p.blocks, p.slices), dim3(256), 0, m->stream, (const uint4*)q, nq, (const uint4*)t, nt,
                           train_index_base, pi, pd, ps, 0, p.sliceRows);
    hipLaunchKernelGGL(k_knn2_merge_slices, dim3((nq + 255) / 256), dim3(256), 0, m->stream, nq, p.slices, (const int*)pi,
                       (const int*)pd, (const int*)ps, (int*)best_idx, (int*)best_d, (int*)second_d);
    ORBX_HIP_CHECK(hipGetLastError());
    if (ordered) {
        ORBX_HIP_CHECK(hipEventRecord(g_scratchEv[m->device], m->stream));
        g_scratchEvRecorded[m->device] = true;
    }
    return ORBX_OK;
}

int orbm_bow_transform(orbm_context* m, const orbm_vocabulary* voc, const uint8_t* desc, int n, int levelsup, int32_t* word_id,
                       double* weight, int32_t* node_id, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (!voc || n < 0 || !word_id || !weight || !node_id || (n > 0 && !desc) || voc->n_nodes < 1 || !voc->node_desc ||
        !voc->child_start || !voc->word_id || !voc->weight || (voc->n_children > 0 && !voc->children)) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (n == 0) return ORBX_OK;
    if (mem_space == ORBX_MEM_HOST) {
        // host tables can be checked: offsets monotone, children in range, a child list of at most 65535 entries
        for (int i = 0; i < voc->n_nodes; i++) {
            const int b = voc->child_start[i], e = voc->child_start[i + 1];
            if (b < 0 || e < b || e > voc->n_children || e - b > 65535) {
                orbx_set_error("vocabulary: bad child list of node %d", i);
                return ORBX_ERR_INVALID;
            }
        }
        for (int c = 0; c < voc->n_children; c++)
            if (voc->children[c] <= 0 || voc->children[c] >= voc->n_nodes) {
                orbx_set_error("vocabulary: child entry %d names node %d of %d", c, voc->children[c], voc->n_nodes);
                return ORBX_ERR_INVALID;
            }
    }
    DevVocabulary V;
    V.nNodes = voc->n_nodes;
    if ((rc = stage_in(m, SL_BT_DESC, voc->node_desc, (size_t)voc->n_nodes * 32, mem_space, &V.nodeDesc))) return rc;
    if ((rc = stage_in(m, SL_BT_START, (const int*)voc->child_start, (size_t)voc->n_nodes + 1, mem_space, &V.childStart))) return rc;
    if ((rc = stage_in(m, SL_BT_CHILD, (const int*)voc->children, (size_t)std::max(voc->n_children, 1), mem_space, &V.children))) return rc;
    if ((rc = stage_in(m, SL_BT_WORD, (const int*)voc->word_id, (size_t)voc->n_nodes, mem_space, &V.wordId))) return rc;
    if ((rc = stage_in(m, SL_BT_WEIGHT, voc->weight, (size_t)voc->n_nodes, mem_space, &V.weight))) return rc;
    const u8* dDesc;
    if ((rc = stage_in(m, SL_BT_FEAT, desc, (size_t)n * 32, mem_space, &dDesc))) return rc;
    int *dWord, *dNode;
    double* dWeight;
    if ((rc = stage_out(m, SL_BT_OUTW, (int*)word_id, (size_t)n, mem_space, &dWord))) return rc;
    if ((rc = stage_out(m, SL_BT_OUTV, weight, (size_t)n, mem_space, &dWeight))) return rc;
    if ((rc = stage_out(m, SL_BT_OUTN, (int*)node_id, (size_t)n, mem_space, &dNode))) return rc;
    ORBM_LAUNCH(k_bow_transform, dim3((n * 16 + 255) / 256), dim3(256), 0, m->stream, V, dDesc, n, voc->levels - levelsup, dWord,
                dWeight, dNode);
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, (int*)word_id, (const int*)dWord, (size_t)n, mem_space))) return rc;
    if ((rc = copy_back(m, weight, (const double*)dWeight, (size_t)n, mem_space))) return rc;
    if ((rc = copy_back(m, (int*)node_id, (const int*)dNode, (size_t)n, mem_space))) return rc;
    if (mem_space == ORBX_MEM_HOST) ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

int orbm_ratio_test(orbm_context* m, int nq, const int32_t* best_d, const int32_t* second_d, float nn_ratio, int max_dist,
                    uint8_t* accept, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (nq < 0 || !best_d || !second_d || !accept) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (nq == 0) return ORBX_OK;
    const int *db, *ds;
    u8* da;
    if ((rc = stage_in(m, SL_KNN_D, (const int*)best_d, (size_t)nq, mem_space, &db))) return rc;
    if ((rc = stage_in(m, SL_KNN_S, (const int*)second_d, (size_t)nq, mem_space, &ds))) return rc;
    if ((rc = stage_out(m, SL_KNN_I, (u8*)accept, (size_t)nq, mem_space, &da))) return rc;
    ORBM_LAUNCH(k_ratio_test, dim3((nq + 255) / 256), dim3(256), 0, m->stream, nq, db, ds, nn_ratio, max_dist, da);
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, (u8*)accept, (const u8*)da, (size_t)nq, mem_space))) return rc;
    if (mem_space == ORBX_MEM_HOST) ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

int orbm_knn2_merge(orbm_context* m, int nq, int32_t* best_idx_a, int32_t* best_d_a, int32_t* second_d_a,
                    const int32_t* best_idx_b, const int32_t* best_d_b, const int32_t* second_d_b, int mem_space) {
    int rc = use_device(m);
    if (rc) return rc;
    if (nq < 0 || !best_idx_a || !best_d_a || !second_d_a || !best_idx_b || !best_d_b || !second_d_b) return ORBX_ERR_INVALID;
    if (nq == 0) return ORBX_OK;
    const int *ib, *db, *sb;
    int *ia, *da, *sa;
    if (mem_space == ORBX_MEM_DEVICE) {
        ia = (int*)best_idx_a; da = (int*)best_d_a; sa = (int*)second_d_a;
        ib = (const int*)best_idx_b; db = (const int*)best_d_b; sb = (const int*)second_d_b;
    } else {
        const int *t0, *t1, *t2;
        if ((rc = stage_in(m, SL_KNN_I, (const int*)best_idx_a, (size_t)nq, mem_space, &t0)) ||
            (rc = stage_in(m, SL_KNN_D, (const int*)best_d_a, (size_t)nq, mem_space, &t1)) ||
            (rc = stage_in(m, SL_KNN_S, (const int*)second_d_a, (size_t)nq, mem_space, &t2)) ||
            (rc = stage_in(m, SL_KNN_I2, (const int*)best_idx_b, (size_t)nq, mem_space, &ib)) ||
            (rc = stage_in(m, SL_KNN_D2, (const int*)best_d_b, (size_t)nq, mem_space, &db)) ||
            (rc = stage_in(m, SL_KNN_S2, (const int*)second_d_b, (size_t)nq, mem_space, &sb)))
            return rc;
        ia = (int*)t0; da = (int*)t1; sa = (int*)t2;
    }
    ORBM_LAUNCH(k_knn2_merge, dim3((nq + 255) / 256), dim3(256), 0, m->stream, nq, ia, da, sa, ib, db, sb);
    ORBX_HIP_CHECK(hipGetLastError());
    if ((rc = copy_back(m, (int*)best_idx_a, (const int*)ia, (size_t)nq, mem_space))) return rc;
    if ((rc = copy_back(m, (int*)best_d_a, (const int*)da, (size_t)nq, mem_space))) return rc;
    if ((rc = copy_back(m, (int*)second_d_a, (const int*)sa, (size_t)nq, mem_space))) return rc;
    if (mem_space == ORBX_MEM_HOST) ORBX_HIP_CHECK(sync_deliver(m));
    return ORBX_OK;
}

}  // extern "C"
