// ORB extraction on MI355X (gfx950): pyramid, per-cell FAST-9/16 with threshold
// fallback, quadtree distribution, intensity-centroid orientation, 7x7
// Gaussian blur and steered BRIEF, batched over frames.
//
// Replaces ORB_SLAM2::ORBextractor (reference src/ORBextractor.cc).  All
// integer stages are exact; float stages keep the reference's operation order
// (this file must be compiled with -ffp-contract=off).
//
// Kernel map (one stream, no host round trip between stages):
//   k_resize    level l from level l-1      (ComputePyramid, :1117-1145)
//   k_fast      one wave per 30-px cell     (ComputeKeyPointsOctTree, :772-839)
//   k_quadtree  one workgroup per (frame,level) (DistributeOctTree, :540-765)
//   k_blur      all levels, LDS-tiled separable integer filter (:1094-1095)
//   k_describe  one wave per keypoint: IC_Angle (:59-88) + rBRIEF (:92-131)

#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/orbx_pattern_31.h"
#include "orbx_internal.h"

namespace {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef unsigned long long u64;
typedef long long i64;

const int ORBX_PIPES = 4;
const int EDGE_THRESHOLD = 19;  // reference src/ORBextractor.cc:53
const int MIN_BORDER = EDGE_THRESHOLD - 3;
const int FAST_LDS_SMALL = 5120;  // LDS comes in 1280-byte granules: four of them let 32 one-wave workgroups share a compute unit's 160 KB
const int MAX_CELL = 59;  // largest cell side: ceil(width / floor(width/30)) < 60
const int MAX_ROOTS = 64;

struct LevelGeom {
    int w, h;
    i64 pyrOff;   // bytes into the per-frame pyramid block (levels >= 1)
    int pyrPitch;
    i64 blurOff;  // bytes into the per-frame blurred block
    int blurPitch;
    int blurTileBase, blurTilesX, blurTiles;
    int blurGroups, blurBands, blurItemBase;  // streaming blur: 4-px column groups x row bands
    // FAST cells
    int nCols, nRows, wCell, hCell;
    int cellBase, nCells, cellCap, slotBase;
    // quadtree
    int quota, nIni, lcap, selBase, keyBase;
    float hX;
    // output
    float scale, kpSize;
};

struct Geom {
    int nlevels, w, h;
    int totalCells, totalSlots, totalSel, totalBlurTiles, totalBlurItems;
    int fastRoiPitch;  // k_fast: widest row pitch of its cell buffer over all levels
    i64 pyrFrameBytes, blurFrameBytes;
    LevelGeom lv[ORBX_MAX_LEVELS];
};

struct ImgRef {
    const u8* base;
    i64 frameStride;
    int pitch;
};

__device__ __forceinline__ ImgRef level_image(const Geom& g, int l, ImgRef img0, const u8* pyr) {
    if (l == 0) return img0;
    ImgRef r;
    r.base = pyr + g.lv[l].pyrOff;
    r.frameStride = g.pyrFrameBytes;
    r.pitch = g.lv[l].pyrPitch;
    return r;
}

// 24-bit multiplies run at the full vector rate; a 32-bit v_mul_lo_u32 takes several times
// longer.  Every product below has operands under 2^24.
__device__ __forceinline__ u32 mul24(u32 a, u32 b) { return __umul24(a, b); }
__device__ __forceinline__ u32 mad24(u32 a, u32 b, u32 c) {
    u32 r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 as_u16x2(u32 v) {
    u16x2 r;
    __builtin_memcpy(&r, &v, 4);
    return r;
}
__device__ __forceinline__ u32 as_u32(u16x2 v) {
    u32 r;
    __builtin_memcpy(&r, &v, 4);
    return r;
}

// A load from a table that no kernel writes (the host fills it before the first launch), through the
// constant address space: the compiler may then fetch it through the scalar cache whenever the address
// is wave-uniform, without having to prove that none of the kernel's stores reaches it (a proof it
// gives up on as a kernel grows).
template <typename T>
__device__ __forceinline__ T load_table(const void* p) {
#if __HIP_DEVICE_COMPILE__
    return *(const __attribute__((address_space(4))) T*)(uintptr_t)p;
#else
    return *(const T*)p;  // the host pass of the compiler only parses this
#endif
}

// ------------------------------------------------------------------ resize
struct ResizeTap {
    u16 i0, i1;  // source indices (already clamped)
    short c0, c1;  // 11-bit fixed-point weights
};

// Bilinear 8-bit downscale, fixed point: horizontal sums with 11-bit weights,
// vertical combine ((b*(t>>4))>>16 twice, +2, >>2).  4 destination pixels per
// thread, one 32-bit store.
__global__ __launch_bounds__(256) void k_resize(ImgRef src, u8* dst, int dPitch, i64 dFrame, int dw,
                                                 int dh, const ResizeTap* __restrict__ xt,
                                                 const ResizeTap* __restrict__ yt) {
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= dw || y >= dh) return;
    const int f = blockIdx.z;
    const ResizeTap ty = yt[y];
    const u8* r0 = src.base + (i64)f * src.frameStride + (i64)ty.i0 * src.pitch;
    const u8* r1 = src.base + (i64)f * src.frameStride + (i64)ty.i1 * src.pitch;
    u32 out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = x0 + k;
        if (x < dw) {
            const ResizeTap tx = xt[x];
            const int t0 = r0[tx.i0] * tx.c0 + r0[tx.i1] * tx.c1;
            const int t1 = r1[tx.i0] * tx.c0 + r1[tx.i1] * tx.c1;
            const int v = ((((int)ty.c0 * (t0 >> 4)) >> 16) + (((int)ty.c1 * (t1 >> 4)) >> 16) + 2) >> 2;
            out |= (u32)(v & 0xFF) << (8 * k);
        }
    }
    *(u32*)(dst + (i64)f * dFrame + (i64)y * dPitch + x0) = out;
}

#define RESIZE_BH 32  // destination rows per band of the streaming kernel

#define RESIZE_PF 4  // source rows fetched ahead

// Streaming form of the same arithmetic.  A lane owns 4 destination columns and walks
// down the SOURCE rows of a band of destination rows.  Per source row it loads three
// aligned words, shifts them into an 8-byte window that starts at its first tap
// (v_alignbyte_b32), picks each pixel's two bytes with a per-lane byte permutation fixed at
// start-up and forms the horizontal sum with one v_dot2_u32_u16; the sums of the previous
// source row stay in registers, so every source row is read and summed once.  A
// destination row is produced when the walk reaches its second source row (a table indexed
// by source row says which, so the walk has no data-dependent loads).  Source rows and
// table entries are fetched RESIZE_PF rows ahead of their use.  Requires 4-byte aligned source rows, taps
// within 8 bytes of the group's first tap, second tap = first or first + 1, and row taps
// that never step backwards (checked on the host; true for scale factors 1 < s <= 1.6).
__global__ __launch_bounds__(256) void k_resize_stream(ImgRef src, int maxWord, int sh, u8* dst, int dPitch, i64 dFrame,
                                                        int dw, int dh, const ResizeTap* __restrict__ xt,
                                                        const ResizeTap* __restrict__ yt,
                                                        const ResizeTap* __restrict__ st, int bandRows) {
    const int G = (dw + 3) >> 2;
    const int nb = (dh + bandRows - 1) / bandRows;
    // blockIdx.x is the frame: blocks are dealt round-robin over the 8 XCDs, so with a batch
    // that is a multiple of 8 all blocks of one frame share an XCD (and its L2) and the rows
    // two neighbouring bands both read are fetched from HBM once.
    const int item = blockIdx.y * 256 + threadIdx.x;
    if (item >= G * nb) return;
    const int band = item / G, gx = item - band * G;
    const int f = blockIdx.x;
    const ResizeTap t0 = xt[min(4 * gx, dw - 1)];
    const int wb = t0.i0 >> 2;
    const u32 shift = t0.i0 & 3;
    u32 sel[4], coef[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const ResizeTap t = xt[min(4 * gx + j, dw - 1)];
        const u32 d0 = t.i0 - t0.i0, d1 = t.i1 - t0.i0;
        // bytes: [tap0, 0, tap1, 0]; selector 12 yields a zero byte
        sel[j] = d0 | (12u << 8) | (d1 << 16) | (12u << 24);
        coef[j] = (u32)(u16)t.c0 | ((u32)(u16)t.c1 << 16);
    }
    const int wA = min(wb, maxWord), wB = min(wb + 1, maxWord), wC = min(wb + 2, maxWord);
    const u8* sbase = src.base + (i64)f * src.frameStride;
    const u32 sPitch = (u32)src.pitch, dPitchU = (u32)dPitch;
    u8* dbase = dst + (i64)f * dFrame + 4 * gx;
    const int y0 = band * bandRows, y1 = min(y0 + bandRows, dh);

    struct Raw {
        u32 a, b, c;
    };
    auto fetch = [&](int s) {
        const u32* row = (const u32*)(sbase + mul24((u32)min(s, sh - 1), sPitch));
        Raw r;
        r.a = row[wA];
        r.b = row[wB];
        r.c = row[wC];
        return r;
    };
    // horizontal sums of one source row, already shifted right by 4 as the vertical step wants
    auto hsum = [&](const Raw& r, u32* t) {
        const u32 W0 = __builtin_amdgcn_alignbyte(r.b, r.a, shift), W1 = __builtin_amdgcn_alignbyte(r.c, r.b, shift);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 pr = __builtin_amdgcn_perm(W1, W0, sel[j]);
            t[j] = __builtin_amdgcn_udot2(as_u16x2(pr), as_u16x2(coef[j]), 0u, false) >> 4;
        }
    };

    // st[2 * s + e], e = 0, 1: the destination rows whose second tap is source row s, as
    // {i0 = destination row or 0xFFFF, i1 = 1 when both taps are row s, c0, c1}
    int s = yt[y0].i0;                // first source row of the band
    const int sLast = yt[y1 - 1].i1;  // last one
    u32 prev[4] = {0, 0, 0, 0}, cur[4];
    struct Group {
        Raw raw[RESIZE_PF];
        uint4 tab[RESIZE_PF];
    };
    auto load_group = [&](Group& g, int s0) {
#pragma unroll
        for (int k = 0; k < RESIZE_PF; k++) {
            g.raw[k] = fetch(s0 + k);
            g.tab[k] = *(const uint4*)(st + 2 * min(s0 + k, sh - 1));
        }
    };
    auto run_group = [&](const Group& g, int s0) {
#pragma unroll
        for (int k = 0; k < RESIZE_PF; k++) {
            hsum(g.raw[k], cur);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const u32 w0 = e == 0 ? g.tab[k].x : g.tab[k].z, w1 = e == 0 ? g.tab[k].y : g.tab[k].w;
                const int y = (int)(w0 & 0xffffu);
                if (y >= y0 && y < y1 && s0 + k <= sLast) {
                    const bool same = (w0 >> 16) != 0;
                    // weights are 0..2048 and the shifted sums < 2^15: (b * t) >> 16 = mulhi(b << 16, t)
                    const u32 b0 = w1 << 16, b1 = w1 & 0xffff0000u;
                    u32 v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const u32 ta = same ? cur[j] : prev[j];
                        v[j] = (__umulhi(b0, ta) + __umulhi(b1, cur[j]) + 2u) >> 2;
                    }
                    const u32 out = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
                    *(u32*)(dbase + mul24((u32)y, dPitchU)) = out;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) prev[j] = cur[j];
        }
    };
    // Two register sets used alternately: a set is refilled right after its rows have been consumed and
    // is not touched again until the other set has been worked through, so its loads have a whole
    // group of rows to land (copying "next" into "current" would wait for them at the end of every group).
    Group ga, gb;
    load_group(ga, s);
    while (true) {
        load_group(gb, s + RESIZE_PF);
        run_group(ga, s);
        s += RESIZE_PF;
        if (s > sLast) break;
        load_group(ga, s + RESIZE_PF);
        run_group(gb, s);
        s += RESIZE_PF;
        if (s > sLast) break;
    }
}

// The same walk with wave-uniform control: a wave owns a band of destination rows and 64 lanes of
// column groups (4 destination pixels each) that share the band's geometry.  The source row, the
// destination row, the row weights and every branch of the walk are then the same for all lanes:
// they live in scalar registers, the row tables come through the scalar cache, and a row's address
// is a buffer descriptor (scalar) + a scalar row offset + a per-lane 32-bit offset fixed at
// start-up, so neither the loads nor the stores need address arithmetic.  The vector ALU is left
// with the arithmetic: per source row one 12-byte load, 2 v_alignbyte, 4 v_perm, 4 v_dot2, 4 shifts; per
// destination row 8 v_mul_hi, the adds, shifts and the byte pack.
//
// The lane's taps lie within 8 bytes of its first tap.  A source row of its window is one 12-byte load
// that starts on a word of the row, and a byte shift (v_alignbyte, 2 per row) moves the 8 bytes the
// selectors address to the front: the memory pipeline, not the ALU, bounds this kernel, and it takes a
// wave's unaligned 8-byte loads (a lane's window starts every 4.8 bytes) at about half the rate of
// aligned ones (640x480: 0.47 -> 0.40 us per frame; three aligned word loads per row cost a quarter more
// time than the unaligned load).  The first tap is pulled back from the row's end so that the 8 bytes
// never reach past the last pixel of the image.
struct ResizeLane {
    u32 sel[4], coef[4];
    u32 offS, offD;
    u32 shift;  // bytes between the aligned 12-byte window that is loaded and the 8 bytes the selectors address
};
__device__ __forceinline__ ResizeLane resize_lane(const ResizeTap* __restrict__ xt, int gx, int dw, int sw, u32 srcLaneBase,
                                                  u32 dstLaneBase) {
    ResizeLane L;
    const ResizeTap t0 = xt[min(4 * gx, dw - 1)];
    const int start = min((int)t0.i0, sw - 8);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const ResizeTap t = xt[min(4 * gx + j, dw - 1)];
        const u32 d0 = t.i0 - start, d1 = t.i1 - start;
        // bytes: [tap0, 0, tap1, 0]; selector 12 yields a zero byte
        L.sel[j] = d0 | (12u << 8) | (d1 << 16) | (12u << 24);
        L.coef[j] = (u32)(u16)t.c0 | ((u32)(u16)t.c1 << 16);
    }
    // the window that is loaded starts on a word of the row when that keeps its 12 bytes inside the row,
    // and at the first tap otherwise (the last lanes of a row; their third word is never selected)
    const int first = (start & ~3) + 12 <= sw ? (start & ~3) : start;
    L.offS = srcLaneBase + (u32)first;
    L.shift = (u32)(start - first);
    L.offD = dstLaneBase + 4u * (u32)gx;
    return L;
}

// One band: destination rows y0 .. y1 - 1.  Everything but L is wave-uniform.
__device__ __forceinline__ void resize_walk(const ResizeLane& L, const __amdgpu_buffer_rsrc_t srcRs,
                                            const __amdgpu_buffer_rsrc_t dstRs, int srcPitch, int dPitch, int sh,
                                            const ResizeTap* __restrict__ yt, const ResizeTap* __restrict__ st, int y0, int y1) {
    typedef u32 u32x3 __attribute__((ext_vector_type(3)));
    struct Group {
        u32x3 raw[RESIZE_PF];
        uint4 tab[RESIZE_PF];  // scalar: st[2 s], st[2 s + 1]
    };
    auto load_group = [&](Group& g, int s0) {
#pragma unroll
        for (int k = 0; k < RESIZE_PF; k++) {
            const int sr = min(s0 + k, sh - 1);
            g.raw[k] = __builtin_amdgcn_raw_buffer_load_b96(srcRs, (int)L.offS, sr * srcPitch, 0);
            g.tab[k] = load_table<uint4>(st + 2 * sr);
        }
    };
    // st[2 * s + e], e = 0, 1: the destination rows whose second tap is source row s, as
    // {i0 = destination row or 0xFFFF, i1 = 1 when both taps are row s, c0, c1}
    // the band is the same for every lane: saying so keeps the row walk in scalar registers whatever
    // divergent code surrounds the call
    y0 = __builtin_amdgcn_readfirstlane(y0);
    y1 = __builtin_amdgcn_readfirstlane(y1);
    int s = __builtin_amdgcn_readfirstlane((int)load_table<ResizeTap>(yt + y0).i0);                // first source row of the band
    const int sLast = __builtin_amdgcn_readfirstlane((int)load_table<ResizeTap>(yt + y1 - 1).i1);  // last one
    u32 prev[4] = {0, 0, 0, 0}, cur[4];
    u32 two = 2;
    asm volatile("" : "+v"(two));  // a vector register for the byte-inserting shifts below
    auto run_group = [&](const Group& g, int s0) {
#pragma unroll
        for (int k = 0; k < RESIZE_PF; k++) {
            const u32 W0 = __builtin_amdgcn_alignbyte(g.raw[k][1], g.raw[k][0], L.shift);
            const u32 W1 = __builtin_amdgcn_alignbyte(g.raw[k][2], g.raw[k][1], L.shift);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 pr = __builtin_amdgcn_perm(W1, W0, L.sel[j]);
                // horizontal sum, already shifted right by 4 as the vertical step wants
                cur[j] = __builtin_amdgcn_udot2(as_u16x2(pr), as_u16x2(L.coef[j]), 0u, false) >> 4;
            }
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const u32 w0 = e == 0 ? g.tab[k].x : g.tab[k].z, w1 = e == 0 ? g.tab[k].y : g.tab[k].w;
                const int y = (int)(w0 & 0xffffu);
                if (y >= y0 && y < y1 && s0 + k <= sLast) {  // scalar branch
                    if ((w0 >> 16) != 0) {                   // both taps are this row (clamped border)
#pragma unroll
                        for (int j = 0; j < 4; j++) prev[j] = cur[j];
                        // keeps this a branch: as a select it would cost four instructions on every row
                        asm volatile("" : "+v"(prev[0]), "+v"(prev[1]), "+v"(prev[2]), "+v"(prev[3]));
                    }
                    // weights are 0..2048 and the shifted sums < 2^15: (b * t) >> 16 = mulhi(b << 16, t)
                    const u32 b0 = w1 << 16, b1 = w1 & 0xffff0000u;
                    u32 v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = __umulhi(b0, prev[j]) + __umulhi(b1, cur[j]) + 2u;
                    // (v >> 2) <= 255: pixels 1..3 are shifted straight into their byte of the output word
                    u32 out = v[0] >> 2;
                    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                        : "+v"(out)
                        : "v"(two), "v"(v[1]));
                    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                        : "+v"(out)
                        : "v"(two), "v"(v[2]));
                    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                        : "+v"(out)
                        : "v"(two), "v"(v[3]));
                    __builtin_amdgcn_raw_buffer_store_b32(out, dstRs, (int)L.offD, y * dPitch, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) prev[j] = cur[j];
        }
    };
    // Two register sets used alternately: a set is refilled right after its rows have been consumed and
    // is not touched again until the other set has been worked through.
    Group ga, gb;
    load_group(ga, s);
    while (true) {
        load_group(gb, s + RESIZE_PF);
        run_group(ga, s);
        s += RESIZE_PF;
        if (s > sLast) break;
        load_group(ga, s + RESIZE_PF);
        run_group(gb, s);
        s += RESIZE_PF;
        if (s > sLast) break;
    }
}

// One level for a batch: the four waves of a workgroup take four consecutive bands; a wave's lanes are
// W column groups (blockIdx.z) x K = 64 / W consecutive frames (lane = frame slot * W + group), W chosen
// per level so that few lanes hang over the row's end.
__global__ __launch_bounds__(256) void k_resize_rows(ImgRef src, int sw, int sh, u8* dst, int dPitch, i64 dFrame, int dw,
                                                     int dh, const ResizeTap* __restrict__ xt,
                                                     const ResizeTap* __restrict__ yt,
                                                     const ResizeTap* __restrict__ st, int bandRows, int logW, int nFrames) {
    const int G = (dw + 3) >> 2;
    const int lane = threadIdx.x & 63;
    // the wave number, read from lane 0, is a scalar
    const int band = (int)blockIdx.y * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (band * bandRows >= dh) return;
    const int gx = ((int)blockIdx.z << logW) + (lane & ((1 << logW) - 1));
    const int fk = lane >> logW;
    const int f0 = (int)blockIdx.x * (64 >> logW);
    if (gx >= G || f0 + fk >= nFrames) return;
    const ResizeLane L = resize_lane(xt, gx, dw, sw, (u32)fk * (u32)src.frameStride, (u32)fk * (u32)dFrame);
    const int K = 64 >> logW;
    const __amdgpu_buffer_rsrc_t srcRs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(src.base + (i64)f0 * src.frameStride), 0, (int)((u32)K * (u32)src.frameStride), 0x00020000);
    const __amdgpu_buffer_rsrc_t dstRs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(dst + (i64)f0 * dFrame), 0, (int)((u32)K * (u32)dFrame), 0x00020000);
    resize_walk(L, srcRs, dstRs, src.pitch, dPitch, sh, yt, st, band * bandRows, min(band * bandRows + bandRows, dh));
}

// The whole pyramid of one frame in one workgroup: frames are independent, so the levels of a frame
// need no grid-wide ordering, only a workgroup barrier between a level and the next (the rows a wave
// reads were stored by waves of its own workgroup, on its own compute unit, and were never loaded
// before).  One launch instead of one per level: no launch gaps, no idle tail at the end of every
// level (workgroups drift apart, so small levels of one frame overlap large levels of another),
// and a level is read back while it is still in the caches.
struct PyramidLevel {
    int sw, sh, dw, dh, srcPitch, dstPitch, chunks;
    int stripGroups;     // column groups past the last full chunk, done as a block of rows per wave step (0: none)
    int stripLogW;       // log2 of the lanes a row of that block takes
    u32 srcOff, dstOff;  // byte offsets of the two levels inside a frame of the pyramid buffer
    const ResizeTap *xt, *yt, *st;
};
struct PyramidArgs {
    PyramidLevel lv[ORBX_MAX_LEVELS];
};
#ifndef PYR_WAVES
#define PYR_WAVES 16
#endif
#ifdef ORBX_PYR_DIAG
// Diagnostic build only (tools/probe/pyr_phases.py): per frame, wave and level three s_memtime stamps
// (level start, walk done, barrier passed).
__device__ unsigned long long* d_pyrDiag;
#endif
#define PYR_STRIP_MAX 16  // widest remainder (column groups) that is done as a block of rows instead of a chunk of walks
__global__ __launch_bounds__(64 * PYR_WAVES, 8) void k_pyramid_frame(ImgRef img0, u8* pyr, i64 pyrFrameBytes, int imgBytes,
                                                                     PyramidArgs args, int nlevels) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int f = blockIdx.x;
    u8* const frame = pyr + (i64)f * pyrFrameBytes;
    // a wave's place in level l: chunk of 64 column groups (wave w -> chunk w % chunks) and, among the waves of
    // that chunk, one contiguous band of rows: one start-up and one exposed load latency per wave and
    // level, and bands that differ by at most a row
    auto lane_setup = [&](int l) {
        const PyramidLevel& A = args.lv[l];
        const int gx = (wave % A.chunks) * 64 + lane;
        return resize_lane(A.xt, min(gx, ((A.dw + 3) >> 2) - 1), A.dw, A.sw, 0u, 0u);
    };
    ResizeLane L = lane_setup(1);
    for (int l = 1; l < nlevels; l++) {
        const PyramidLevel& A = args.lv[l];
#ifdef ORBX_PYR_DIAG
        unsigned long long* drec = d_pyrDiag ? d_pyrDiag + (((size_t)f * PYR_WAVES + wave) * ORBX_MAX_LEVELS + l) * 4 : nullptr;
        if (drec && lane == 0) drec[0] = __builtin_amdgcn_s_memtime();
#endif
        const __amdgpu_buffer_rsrc_t srcRs =
            l == 1 ? __builtin_amdgcn_make_buffer_rsrc((void*)(img0.base + (i64)f * img0.frameStride), 0, imgBytes, 0x00020000)
                   : __builtin_amdgcn_make_buffer_rsrc((void*)(frame + A.srcOff), 0, (int)(A.sh * A.srcPitch), 0x00020000);
        const __amdgpu_buffer_rsrc_t dstRs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(frame + A.dstOff), 0, (int)(A.dh * A.dstPitch), 0x00020000);
        const int G = (A.dw + 3) >> 2;
        const int Gwalk = A.stripGroups > 0 ? A.chunks * 64 : G;  // groups covered by the row walks
        const int chunk = wave % A.chunks, idx = wave / A.chunks;
        const int nw = (PYR_WAVES - chunk + A.chunks - 1) / A.chunks;  // waves on this chunk
        const int y0 = (int)((i64)idx * A.dh / nw), y1 = (int)((i64)(idx + 1) * A.dh / nw);
        if (chunk * 64 + lane < Gwalk && y1 > y0) resize_walk(L, srcRs, dstRs, A.srcPitch, A.dstPitch, A.sh, A.yt, A.st, y0, y1);
        // A few column groups past the last full chunk would cost a chunk's worth of row walks for a handful
        // of lanes (a walk takes the same time whatever the number of active lanes).  They are done as a
        // block instead: a wave's lanes are Wp column groups (the remainder rounded up to a power of two)
        // x 64 / Wp consecutive destination rows, each lane fetching the windows of its own two source
        // rows; same arithmetic, no carried row sums, and a step still touches few cache lines.
        if (A.stripGroups > 0) {
            const int logWp = A.stripLogW, R = 64 >> logWp;
            const int k = lane & ((1 << logWp) - 1), r = lane >> logWp;
            const int steps = (A.dh + R - 1) / R;
            if (PYR_WAVES - 1 - wave < steps && k < A.stripGroups) {
                const ResizeLane S = resize_lane(A.xt, A.chunks * 64 + k, A.dw, A.sw, 0u, 0u);
                for (int stp = PYR_WAVES - 1 - wave; stp < steps; stp += PYR_WAVES) {
                    const int y = stp * R + r;
                    if (y < A.dh) {
                        const ResizeTap ty = A.yt[y];
                        typedef u32 u32x3 __attribute__((ext_vector_type(3)));
                        const u32x3 a0 = __builtin_amdgcn_raw_buffer_load_b96(srcRs, (int)(S.offS + (u32)ty.i0 * (u32)A.srcPitch), 0, 0);
                        const u32x3 a1 = __builtin_amdgcn_raw_buffer_load_b96(srcRs, (int)(S.offS + (u32)ty.i1 * (u32)A.srcPitch), 0, 0);
                        const u32 r0[2] = {__builtin_amdgcn_alignbyte(a0[1], a0[0], S.shift), __builtin_amdgcn_alignbyte(a0[2], a0[1], S.shift)};
                        const u32 r1[2] = {__builtin_amdgcn_alignbyte(a1[1], a1[0], S.shift), __builtin_amdgcn_alignbyte(a1[2], a1[1], S.shift)};
                        const u32 b0 = (u32)(u16)ty.c0 << 16, b1 = (u32)(u16)ty.c1 << 16;
                        u32 out = 0;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const u32 t0 = __builtin_amdgcn_udot2(as_u16x2(__builtin_amdgcn_perm(r0[1], r0[0], S.sel[j])), as_u16x2(S.coef[j]), 0u, false) >> 4;
                            const u32 t1 = __builtin_amdgcn_udot2(as_u16x2(__builtin_amdgcn_perm(r1[1], r1[0], S.sel[j])), as_u16x2(S.coef[j]), 0u, false) >> 4;
                            out |= ((__umulhi(b0, t0) + __umulhi(b1, t1) + 2u) >> 2) << (8 * j);
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(out, dstRs, (int)(S.offD + (u32)y * (u32)A.dstPitch), 0, 0);
                    }
                }
            }
        }
        // the next level's column taps do not depend on this level's pixels: fetched while the frame's other waves finish
        if (l + 1 < nlevels) L = lane_setup(l + 1);
#ifdef ORBX_PYR_DIAG
        if (drec && lane == 0) drec[1] = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();  // waits for this wave's stores, then for every wave of the frame
#ifdef ORBX_PYR_DIAG
        if (drec && lane == 0) drec[2] = __builtin_amdgcn_s_memtime();
#endif
    }
}

// ------------------------------------------------------------------ blur
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

#define BLUR_TW 128
#define BLUR_TH 16

// Separable 7-tap integer Gaussian with reflect-101 borders.  Row sums fit 16
// bits (257*255 = 65535); output = sat_u8((sum + 32768) >> 16).
__global__ __launch_bounds__(256) void k_blur(Geom g, ImgRef img0, const u8* __restrict__ pyr,
                                               u8* __restrict__ blur, int k0, int k1, int k2, int k3,
                                               int tileOffset, int halfEven) {
    __shared__ u8 tin[BLUR_TH + 6][BLUR_TW + 8];
    __shared__ u16 hs[BLUR_TH + 6][BLUR_TW];
    const int f = blockIdx.y;
    const int tileId = blockIdx.x + tileOffset;
    int l = 0;
    while (l + 1 < g.nlevels && tileId >= g.lv[l + 1].blurTileBase) l++;
    const LevelGeom& L = g.lv[l];
    const int t = tileId - L.blurTileBase;
    const int tx0 = (t % L.blurTilesX) * BLUR_TW, ty0 = (t / L.blurTilesX) * BLUR_TH;
    const ImgRef im = level_image(g, l, img0, pyr);
    const u8* src = im.base + (i64)f * im.frameStride;
    const int tid = threadIdx.x;
    for (int i = tid; i < (BLUR_TH + 6) * (BLUR_TW + 6); i += 256) {
        const int ry = i / (BLUR_TW + 6), rx = i - ry * (BLUR_TW + 6);
        const int sy = reflect101(min(ty0 + ry - 3, L.h + 2), L.h);
        const int sx = reflect101(min(tx0 + rx - 3, L.w + 2), L.w);
        tin[ry][rx] = src[(i64)sy * im.pitch + sx];
    }
    __syncthreads();
    for (int i = tid; i < (BLUR_TH + 6) * BLUR_TW; i += 256) {
        const int ry = i / BLUR_TW, rx = i - ry * BLUR_TW;
        const u8* p = &tin[ry][rx];
        const int s = k0 * (p[0] + p[6]) + k1 * (p[1] + p[5]) + k2 * (p[2] + p[4]) + k3 * p[3];
        hs[ry][rx] = (u16)s;
    }
    __syncthreads();
    u8* dst = blur + (i64)f * g.blurFrameBytes + L.blurOff;
    for (int i = tid; i < BLUR_TH * (BLUR_TW / 4); i += 256) {
        const int ry = i / (BLUR_TW / 4), rx = (i - ry * (BLUR_TW / 4)) * 4;
        const int oy = ty0 + ry, ox = tx0 + rx;
        if (oy >= L.h || ox >= L.w) continue;
        u32 out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = rx + k;
            const int s = k0 * ((int)hs[ry][x] + hs[ry + 6][x]) + k1 * ((int)hs[ry + 1][x] + hs[ry + 5][x]) +
                          k2 * ((int)hs[ry + 2][x] + hs[ry + 4][x]) + k3 * (int)hs[ry + 3][x];
            // halfEven: ties go to the even value in the columns below w & ~3 (single-precision column
            // pass of OpenCV <= 3.4.0; its last w % 4 columns round half up like the integer path)
            const int bias = (halfEven && ox + k < (L.w & ~3)) ? 0x7FFF + ((s >> 16) & 1) : 32768;
            int v = (s + bias) >> 16;
            v = v > 255 ? 255 : v;
            out |= (u32)v << (8 * k);
        }
        *(u32*)(dst + (i64)oy * L.blurPitch + ox) = out;
    }
}

#ifndef BLUR_BH
#define BLUR_BH 84  // output rows per band (a multiple of the 7-row window; 28..168 measured, 84 fastest)
#endif

// Streaming form of the same filter: no LDS, no barriers.  A lane owns a column
// of 4 pixels and walks down a band of rows keeping the last 7 horizontal sums
// in registers.  Per row it loads three aligned 32-bit words (left, own, right
// group); reflect-101 at the image's left/right edges is a per-lane byte
// permutation fixed at start-up, at the top/bottom a reflected row index.
// The 7 taps of a row are two 4-byte dot products (v_dot4_u32_u8).
// HALF_EVEN: rounding of the column pass as in k_blur.
template <bool HALF_EVEN>
__global__ __launch_bounds__(256) void k_blur_stream(Geom g, ImgRef img0, const u8* __restrict__ pyr,
                                                      u8* __restrict__ blur, int k0, int k1, int k2, int k3,
                                                      int firstLevel) {
    const int item = blockIdx.y * 256 + threadIdx.x;  // blockIdx.x = frame (one XCD per frame)
    const int f = blockIdx.x;
    int l = firstLevel;
    while (l + 1 < g.nlevels && item >= g.lv[l + 1].blurItemBase) l++;
    const LevelGeom& L = g.lv[l];
    const int it = item - L.blurItemBase;
    const int G = L.blurGroups;
    if (it < 0 || it >= G * L.blurBands) return;
    const int band = it / G, gx = it - band * G;
    const int w = L.w, h = L.h;
    if (h < 4 || w < 4) return;  // such levels go through k_blur (the reflections below are single-step)
    const ImgRef im = level_image(g, l, img0, pyr);
    const u8* src = im.base + (i64)f * im.frameStride;
    const u32 srcPitch = (u32)im.pitch, dstPitch = (u32)L.blurPitch;  // locals: nothing below re-reads the arguments
    u8* dst = blur + (i64)f * g.blurFrameBytes + L.blurOff + 4 * gx;
    const int gl = max(gx - 1, 0), gr = min(gx + 1, G - 1);
    const bool lastGroup = gx == G - 1;
    const u32 tailUp = (lastGroup && (w & 3)) ? 1u : 0u;

    // byte selectors for the 12 positions x = 4*gx-4 .. 4*gx+7
    u32 selL = 0, selC = 0, selR = 0;
#pragma unroll
    for (int p = 0; p < 12; p++) {
        const int x = min(max(4 * gx - 4 + p, -3), w + 2);
        int xr = x < 0 ? -x : x;  // single-step reflect-101 (w >= 4 here)
        xr = xr >= w ? 2 * (w - 1) - xr : xr;
        const int gs = xr >> 2, bs = xr & 3;
        u32 code;
        if (p < 8) {
            // sources: S1 = left word (0..3), S0 = own word (4..7)
            code = gs == gx ? 4 + bs : gs == gl ? bs : 4;
        } else {
            // sources: S1 = right word (left word for the last group), S0 = own word
            const int ga = lastGroup ? gl : gr;
            code = gs == gx ? 4 + bs : gs == ga ? bs : 4;
        }
        if (p < 4) selL |= code << (8 * p);
        else if (p < 8) selC |= code << (8 * (p - 4));
        else selR |= code << (8 * (p - 8));
    }
    const u32 W1 = (u32)k0 | ((u32)k1 << 8) | ((u32)k2 << 16) | ((u32)k3 << 24);
    const u32 W2 = (u32)k2 | ((u32)k1 << 8) | ((u32)k0 << 16);
    const int y0 = band * BLUR_BH;
    const int yEnd = min(y0 + BLUR_BH, h);

    // Rows are fetched one 7-row group ahead of their use: a wave keeps 7 rows (about 1.8 KB) in
    // flight while it filters the previous group, which is what hides the HBM latency here.
    struct Raw {
        u32 a, c, b;
    };
    auto fetch = [&](int y) {
        // one reflection covers y in [-3, h+2] (h >= 4 here)
        y = min(y, h + 2);
        int r = y < 0 ? -y : y;
        r = r >= h ? 2 * (h - 1) - r : r;
        const u32* row = (const u32*)(src + mul24((u32)r, srcPitch));
        Raw o;
        o.a = row[gl];
        o.c = row[gx];
        o.b = row[gr];
        return o;
    };
    auto hrow = [&](const Raw& rw, u32* hs) {
        const u32 Lw = __builtin_amdgcn_perm(rw.c, rw.a, selL);
        const u32 Cw = __builtin_amdgcn_perm(rw.c, rw.a, selC);
        const u32 Rw = __builtin_amdgcn_perm(rw.c, lastGroup ? rw.a : rw.b, selR);
        // output j uses bytes j+1..j+4 (taps k0..k3) and j+5..j+7 (taps k2,k1,k0)
        hs[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rw, Cw, 1), W2,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Cw, Lw, 1), W1, 0u, false), false);
        hs[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rw, Cw, 2), W2,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Cw, Lw, 2), W1, 0u, false), false);
        hs[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rw, Cw, 3), W2,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Cw, Lw, 3), W1, 0u, false), false);
        hs[3] = __builtin_amdgcn_udot4(Rw, W2, __builtin_amdgcn_udot4(Cw, W1, 0u, false), false);
    };
    // Vertical pass on pairs of consecutive rows: P_r = (hs[r-1], hs[r]) as two 16-bit halves
    // (row sums are < 2^16), so that output row y is four v_dot2_u32_u16:
    //   P[y-2].(k0,k1) + P[y].(k2,k3) + P[y+2].(k2,k1) + P[y+3].(0,k0) + 32768.
    // win[] is a ring of the last seven P (slot (ph + 6) % 7 receives P[y+3] at phase ph).
    const u32 W01 = (u32)k0 | ((u32)k1 << 16), W23 = (u32)k2 | ((u32)k3 << 16), W21 = (u32)k2 | ((u32)k1 << 16),
              W0h = (u32)k0 << 16;
    u32 win[7][4], lastHs[4];
    Raw cur[7];
    {
        Raw ini[6];
#pragma unroll
        for (int i = 0; i < 6; i++) ini[i] = fetch(y0 - 3 + i);
#pragma unroll
        for (int i = 0; i < 7; i++) cur[i] = fetch(y0 + 3 + i);
        u32 h0[4], h1[4];
        hrow(ini[0], h0);
#pragma unroll
        for (int i = 1; i < 6; i++) {
            hrow(ini[i], h1);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                win[i][k] = h0[k] | (h1[k] << 16);
                h0[k] = h1[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) lastHs[k] = h0[k];
    }
    auto load_group = [&](Raw* grp, int yFirst) {
#pragma unroll
        for (int i = 0; i < 7; i++) grp[i] = fetch(yFirst + i);
    };
    auto run_group = [&](const Raw* grp, int yb) {
#pragma unroll
        for (int ph = 0; ph < 7; ph++) {
            const int y = yb + ph;
            u32 hs[4];
            hrow(grp[ph], hs);
            // The rounded sum is < 2^25; clamped to 0xFFFFFF its byte 2 is the saturated output pixel.
            u32 sv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 P3 = lastHs[k] | (hs[k] << 16);
                win[(ph + 6) % 7][k] = P3;
                lastHs[k] = hs[k];
                u32 s = __builtin_amdgcn_udot2(as_u16x2(win[(ph + 1) % 7][k]), as_u16x2(W01), HALF_EVEN ? 0u : 32768u, false);
                s = __builtin_amdgcn_udot2(as_u16x2(win[(ph + 3) % 7][k]), as_u16x2(W23), s, false);
                s = __builtin_amdgcn_udot2(as_u16x2(win[(ph + 5) % 7][k]), as_u16x2(W21), s, false);
                s = __builtin_amdgcn_udot2(as_u16x2(P3), as_u16x2(W0h), s, false);
                if (HALF_EVEN) s += 0x7FFFu + (((s >> 16) & 1u) | tailUp);  // ties to even; the image's last w % 4 columns half up
                sv[k] = min(s, 0xFFFFFFu);
            }
            const u32 lo = __builtin_amdgcn_perm(sv[1], sv[0], 0x0c0c0602u);  // bytes: sv0[2], sv1[2], 0, 0
            const u32 hi = __builtin_amdgcn_perm(sv[3], sv[2], 0x06020c0cu);  // bytes: 0, 0, sv2[2], sv3[2]
            const u32 out = lo | hi;
            if (y < yEnd) *(u32*)(dst + mul24((u32)y, dstPitch)) = out;
        }
    };
    // Two register sets used alternately (no copy from "next" to "current", which would wait for the
    // loads at the end of every group): a set is refilled right after its rows have been filtered.
    Raw other[7];
    int yb = y0;
    while (true) {
        if (yb + 7 < yEnd) load_group(other, yb + 10);
        run_group(cur, yb);
        yb += 7;
        if (yb >= yEnd) break;
        if (yb + 7 < yEnd) load_group(cur, yb + 10);
        run_group(other, yb);
        yb += 7;
        if (yb >= yEnd) break;
    }
}

// The same filter with wave-uniform control, as in the resize: a wave owns one band of one level for
// W column groups x K = 64 / W frames (lane = frame slot * W + group), so the row walk, the reflected
// row indices, the phase of the 7-row ring and every branch are scalar, and a row of a lane's window
// is ONE 12-byte buffer load (descriptor of the wave's first frame + scalar row offset + a per-lane
// offset fixed at start-up) instead of three word loads with per-lane addresses.  The word before
// the first group's window starts at its own word (the selectors of that lane account for it), and the
// word after the last group is loaded but never selected.  Arithmetic and selectors are those of
// k_blur_stream.  task = (level, band, chunk of W groups); tasks of a level are contiguous from its
// blurTaskBase.
struct BlurRowsArgs {
    int taskBase[ORBX_MAX_LEVELS + 1];  // first task of each level; [nlevels] = total
    int chunks[ORBX_MAX_LEVELS];        // chunks of W groups per row of the level
    int logW;
    int firstLevel;
};
template <bool HALF_EVEN>
__global__ __launch_bounds__(256) void k_blur_rows(Geom g, ImgRef img0, const u8* __restrict__ pyr, u8* __restrict__ blur,
                                                    int k0, int k1, int k2, int k3, BlurRowsArgs A, int nFrames) {
    const int lane = threadIdx.x & 63;
    const int task = (int)blockIdx.y * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (task >= A.taskBase[g.nlevels]) return;
    int l = A.firstLevel;
    while (l + 1 < g.nlevels && task >= A.taskBase[l + 1]) l++;
    const LevelGeom& L = g.lv[l];
    const int t = task - A.taskBase[l];
    const int chunks = A.chunks[l];
    const int band = t / chunks, chunk = t - band * chunks;
    const int logW = A.logW, K = 64 >> logW;
    const int gx = (chunk << logW) + (lane & ((1 << logW) - 1));
    const int fk = lane >> logW;
    const int f0 = (int)blockIdx.x * K;
    const int G = L.blurGroups;
    if (gx >= G || f0 + fk >= nFrames) return;
    const int w = L.w, h = L.h;
    const ImgRef im = level_image(g, l, img0, pyr);
    const int srcPitch = im.pitch, dstPitch = L.blurPitch;
    const __amdgpu_buffer_rsrc_t srcRs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(im.base + (i64)f0 * im.frameStride), 0, (int)((u32)K * (u32)im.frameStride), 0x00020000);
    const __amdgpu_buffer_rsrc_t dstRs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(blur + (i64)f0 * g.blurFrameBytes + L.blurOff), 0, (int)((u32)K * (u32)g.blurFrameBytes), 0x00020000);
    // the lane's 12-byte window holds words wa, wa + 1, wa + 2 of a row (registers A, C, B): wa = gx - 1,
    // except for the first group, whose window starts at its own word
    const int wa = max(gx - 1, 0);
    const int offS = (int)((u32)fk * (u32)im.frameStride) + 4 * wa;
    const int offD = (int)((u32)fk * (u32)g.blurFrameBytes) + 4 * gx;
    const bool lastGroup = gx == G - 1 && gx > 0;
    const u32 tailUp = (gx == G - 1 && (w & 3)) ? 1u : 0u;

    // byte selectors for the 12 positions x = 4*gx-4 .. 4*gx+7
    u32 selL = 0, selC = 0, selR = 0;
#pragma unroll
    for (int p = 0; p < 12; p++) {
        const int x = min(max(4 * gx - 4 + p, -3), w + 2);
        int xr = x < 0 ? -x : x;  // single-step reflect-101 (w >= 4 here)
        xr = xr >= w ? 2 * (w - 1) - xr : xr;
        const int gs = xr >> 2, bs = xr & 3;
        u32 code;
        if (p < 8) {
            // sources: S1 = register A (0..3), S0 = register C (4..7)
            code = gs == wa + 1 ? 4 + bs : gs == wa ? bs : 4;
        } else {
            // sources: S1 = register B (register A for the last group), S0 = register C
            const int ga = lastGroup ? wa : wa + 2;
            code = gs == wa + 1 ? 4 + bs : gs == ga ? bs : 4;
        }
        if (p < 4) selL |= code << (8 * p);
        else if (p < 8) selC |= code << (8 * (p - 4));
        else selR |= code << (8 * (p - 8));
    }
    const u32 W1 = (u32)k0 | ((u32)k1 << 8) | ((u32)k2 << 16) | ((u32)k3 << 24);
    const u32 W2 = (u32)k2 | ((u32)k1 << 8) | ((u32)k0 << 16);
    const int y0 = band * BLUR_BH;
    const int yEnd = min(y0 + BLUR_BH, h);

    typedef u32 u32x3 __attribute__((ext_vector_type(3)));
    auto fetch = [&](int y) {
        // one reflection covers y in [-3, h+2] (h >= 4 here); all of it scalar
        y = min(y, h + 2);
        int r = y < 0 ? -y : y;
        r = r >= h ? 2 * (h - 1) - r : r;
        return __builtin_amdgcn_raw_buffer_load_b96(srcRs, offS, r * srcPitch, 0);
    };
    auto hrow = [&](const u32x3& rw, u32* hs) {
        const u32 a = rw[0], c = rw[1], b = rw[2];
        const u32 Lw = __builtin_amdgcn_perm(c, a, selL);
        const u32 Cw = __builtin_amdgcn_perm(c, a, selC);
        const u32 Rw = __builtin_amdgcn_perm(c, lastGroup ? a : b, selR);
        // output j uses bytes j+1..j+4 (taps k0..k3) and j+5..j+7 (taps k2,k1,k0)
        hs[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rw, Cw, 1), W2,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Cw, Lw, 1), W1, 0u, false), false);
        hs[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rw, Cw, 2), W2,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Cw, Lw, 2), W1, 0u, false), false);
        hs[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rw, Cw, 3), W2,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Cw, Lw, 3), W1, 0u, false), false);
        hs[3] = __builtin_amdgcn_udot4(Rw, W2, __builtin_amdgcn_udot4(Cw, W1, 0u, false), false);
    };
    // Vertical pass on pairs of consecutive rows: P_r = (hs[r-1], hs[r]) as two 16-bit halves
    // (row sums are < 2^16), so that output row y is four v_dot2_u32_u16:
    //   P[y-2].(k0,k1) + P[y].(k2,k3) + P[y+2].(k2,k1) + P[y+3].(0,k0) + 32768.
    // win[] is a ring of the last seven P (slot (ph + 6) % 7 receives P[y+3] at phase ph).
    const u32 W01 = (u32)k0 | ((u32)k1 << 16), W23 = (u32)k2 | ((u32)k3 << 16), W21 = (u32)k2 | ((u32)k1 << 16),
              W0h = (u32)k0 << 16;
    u32 win[7][4], lastHs[4];
    u32x3 cur[7];
    {
        u32x3 ini[6];
#pragma unroll
        for (int i = 0; i < 6; i++) ini[i] = fetch(y0 - 3 + i);
#pragma unroll
        for (int i = 0; i < 7; i++) cur[i] = fetch(y0 + 3 + i);
        u32 h0[4], h1[4];
        hrow(ini[0], h0);
#pragma unroll
        for (int i = 1; i < 6; i++) {
            hrow(ini[i], h1);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                win[i][k] = h0[k] | (h1[k] << 16);
                h0[k] = h1[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) lastHs[k] = h0[k];
    }
    auto load_group = [&](u32x3* grp, int yFirst) {
#pragma unroll
        for (int i = 0; i < 7; i++) grp[i] = fetch(yFirst + i);
    };
    auto run_group = [&](const u32x3* grp, int yb) {
#pragma unroll
        for (int ph = 0; ph < 7; ph++) {
            const int y = yb + ph;
            u32 hs[4];
            hrow(grp[ph], hs);
            // The rounded sum is < 2^25; clamped to 0xFFFFFF its byte 2 is the saturated output pixel.
            u32 sv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 P3 = lastHs[k] | (hs[k] << 16);
                win[(ph + 6) % 7][k] = P3;
                lastHs[k] = hs[k];
                u32 s = __builtin_amdgcn_udot2(as_u16x2(win[(ph + 1) % 7][k]), as_u16x2(W01), HALF_EVEN ? 0u : 32768u, false);
                s = __builtin_amdgcn_udot2(as_u16x2(win[(ph + 3) % 7][k]), as_u16x2(W23), s, false);
                s = __builtin_amdgcn_udot2(as_u16x2(win[(ph + 5) % 7][k]), as_u16x2(W21), s, false);
                s = __builtin_amdgcn_udot2(as_u16x2(P3), as_u16x2(W0h), s, false);
                if (HALF_EVEN) s += 0x7FFFu + (((s >> 16) & 1u) | tailUp);  // ties to even; the image's last w % 4 columns half up
                sv[k] = min(s, 0xFFFFFFu);
            }
            const u32 lo = __builtin_amdgcn_perm(sv[1], sv[0], 0x0c0c0602u);  // bytes: sv0[2], sv1[2], 0, 0
            const u32 hi = __builtin_amdgcn_perm(sv[3], sv[2], 0x06020c0cu);  // bytes: 0, 0, sv2[2], sv3[2]
            const u32 out = lo | hi;
            if (y < yEnd) __builtin_amdgcn_raw_buffer_store_b32(out, dstRs, offD, y * dstPitch, 0);  // scalar branch
        }
    };
    // Two register sets used alternately: a set is refilled right after its rows have been filtered.
    u32x3 other[7];
    int yb = y0;
    while (true) {
        if (yb + 7 < yEnd) load_group(other, yb + 10);
        run_group(cur, yb);
        yb += 7;
        if (yb >= yEnd) break;
        if (yb + 7 < yEnd) load_group(cur, yb + 10);
        run_group(other, yb);
        yb += 7;
        if (yb >= yEnd) break;
    }
}

// ------------------------------------------------------------------ one walk per level: blur + next level
// A level is read ONCE: the row walk of k_blur_rows (a lane = 4 columns of the level, one 12-byte window
// per row) also produces the next pyramid level (cv::resize INTER_LINEAR, arithmetic of resize_walk).  The
// window of a lane holds both taps of every destination pixel whose first tap lies in the lane's own 4
// columns: at most 4 consecutive pixels (scale factor > 1), whose horizontal sums, vertical combine and
// rounding are those of the resize kernels.  Destination rows are not aligned with the lanes' columns, so
// a finished row is regrouped across lanes: every lane packs its (at most 4) bytes into a word, and the
// lane that stores destination word k fetches the words of the one or two lanes that hold its 4 pixels
// (two ds_bpermute, one byte permutation) and stores 4 bytes.  A chunk of a row is a run of destination
// words together with the W column groups that hold their taps (host table: first group, first word,
// words); neighbouring chunks overlap by a group or two, whose blurred pixels are then stored twice with
// the same value.  Lanes past the row's end or the batch's last frame stay active (the regrouping reads
// every lane): their loads and stores fall outside the buffer descriptors, which return zero / drop them.
struct WalkChunk {
    int g0, k0, nk, pad;  // first column group, first destination word, destination words
};
struct WalkArgs {
    ImgRef src;              // the level that is walked
    int w, h;
    u32 blurOff;             // inside a frame of the blurred block
    int blurPitch;
    int dw, dh, dstPitch;    // next level (dw = 0: none)
    u32 dstOff;              // inside a frame of the pyramid block
    const ResizeTap *xt, *st;  // column taps of the next level; per source row, the destination rows that finish on it
    const u16* firstDest;    // per column group g: first destination pixel whose first tap lies in group >= g
    const WalkChunk* chunks;
    int nChunks, logW, bands, bandRows;  // bandRows: a multiple of the 7-row window
};
template <bool HALF_EVEN, bool RESIZE>
__global__ __launch_bounds__(256) void k_level_walk(WalkArgs A, u8* __restrict__ pyr, i64 pyrFrameBytes, u8* __restrict__ blur,
                                                     i64 blurFrameBytes, int k0, int k1, int k2, int k3, int nFrames) {
    const int lane = threadIdx.x & 63;
    const int task = (int)blockIdx.y * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (task >= A.bands * A.nChunks) return;
    const int band = task / A.nChunks, chunk = task - band * A.nChunks;
    const int4 CH = load_table<int4>(A.chunks + chunk);
    const int logW = A.logW, K = 64 >> logW, Wm = (1 << logW) - 1;
    const int li = lane & Wm, fk = lane >> logW;
    const int f0 = (int)blockIdx.x * K;
    const int Kv = min(K, nFrames - f0);  // frames of this wave that exist
    const int w = A.w, h = A.h, G = (w + 3) >> 2;
    const int gxr = CH.x + li;
    const bool laneOn = gxr < G;
    const int gx = min(gxr, G - 1);
    const int srcPitch = A.src.pitch, dstPitch = A.blurPitch;
    const __amdgpu_buffer_rsrc_t srcRs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(A.src.base + (i64)f0 * A.src.frameStride), 0, (int)((u32)Kv * (u32)A.src.frameStride), 0x00020000);
    const __amdgpu_buffer_rsrc_t dstRs = __builtin_amdgcn_make_buffer_rsrc((void*)(blur + (i64)f0 * blurFrameBytes + A.blurOff), 0,
                                                                            (int)((u32)Kv * (u32)blurFrameBytes), 0x00020000);
    const u32 OOB = 0x7FFFFFF0u;  // beyond every descriptor's size: such a store is dropped
    const int wa = max(gx - 1, 0);
    const int offS = (int)((u32)fk * (u32)A.src.frameStride) + 4 * wa;
    const int offD = laneOn ? (int)((u32)fk * (u32)blurFrameBytes) + 4 * gx : (int)OOB;
    const bool lastGroup = gx == G - 1 && gx > 0;
    const u32 tailUp = (gx == G - 1 && (w & 3)) ? 1u : 0u;

    // byte selectors for the 12 positions x = 4*gx-4 .. 4*gx+7 (as in k_blur_rows)
    u32 selL = 0, selC = 0, selR = 0;
#pragma unroll
    for (int p = 0; p < 12; p++) {
        const int x = min(max(4 * gx - 4 + p, -3), w + 2);
        int xr = x < 0 ? -x : x;
        xr = xr >= w ? 2 * (w - 1) - xr : xr;
        const int gs = xr >> 2, bs = xr & 3;
        u32 code;
        if (p < 8) {
            code = gs == wa + 1 ? 4 + bs : gs == wa ? bs : 4;
        } else {
            const int ga = lastGroup ? wa : wa + 2;
            code = gs == wa + 1 ? 4 + bs : gs == ga ? bs : 4;
        }
        if (p < 4) selL |= code << (8 * p);
        else if (p < 8) selC |= code << (8 * (p - 4));
        else selR |= code << (8 * (p - 8));
    }
    const u32 W1 = (u32)k0 | ((u32)k1 << 8) | ((u32)k2 << 16) | ((u32)k3 << 24);
    const u32 W2 = (u32)k2 | ((u32)k1 << 8) | ((u32)k0 << 16);
    const int y0 = band * A.bandRows;
    const int yEnd = min(y0 + A.bandRows, h);

    // ---- the next level: this lane's destination pixels, and the destination word it stores
    u32 rsel[4] = {0x0c0c0c0cu, 0x0c0c0c0cu, 0x0c0c0c0cu, 0x0c0c0c0cu}, rcoef[4] = {0, 0, 0, 0};
    u32 selG = 0x0c0c0c0cu;
    int addrA = 0, addrB = 0, offP = (int)OOB;
    __amdgpu_buffer_rsrc_t pyrRs = dstRs;
    const bool firstGroup = gx == 0;  // its own word is the first register of the window
    if (RESIZE) {
        pyrRs = __builtin_amdgcn_make_buffer_rsrc((void*)(pyr + (i64)f0 * pyrFrameBytes + A.dstOff), 0, (int)((u32)Kv * (u32)pyrFrameBytes),
                                                  0x00020000);
        const int dLo = 4 * CH.y, dHi = min(4 * (CH.y + CH.z), A.dw);  // destination pixels of this chunk
        // pixels of this lane: first tap in its own 4 columns, inside the chunk's range
        auto lane_range = [&](int g, int* b, int* e) {
            *b = max((int)A.firstDest[g], dLo);
            *e = min((int)A.firstDest[g + 1], dHi);
        };
        if (laneOn) {
            int db, de;
            lane_range(gx, &db, &de);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (db + j < de) {
                    const ResizeTap t = A.xt[db + j];
                    rsel[j] = (u32)(t.i0 - 4 * gx) | (12u << 8) | ((u32)(t.i1 - 4 * gx) << 16) | (12u << 24);
                    rcoef[j] = (u32)(u16)t.c0 | ((u32)(u16)t.c1 << 16);
                }
            }
        }
        if (li < CH.z) {
            const int kd = CH.y + li;
            int sA = -1;
            selG = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int d = 4 * kd + t;
                u32 code = 12u;
                if (d < A.dw) {
                    const int gs = (int)(A.xt[d].i0 >> 2);
                    if (sA < 0) sA = gs;
                    int db, de;
                    lane_range(gs, &db, &de);
                    code = (u32)(d - db) + (gs == sA ? 0u : 4u);
                }
                selG |= code << (8 * t);
            }
            sA = max(sA, CH.x) - CH.x;
            addrA = 4 * ((fk << logW) + sA);
            addrB = 4 * ((fk << logW) + min(sA + 1, Wm));
            offP = (int)((u32)fk * (u32)pyrFrameBytes) + 4 * kd;
        }
    }
    u32 rprev[4] = {0, 0, 0, 0};
    u32 two = 2;
    asm volatile("" : "+v"(two));  // a vector register for the byte-inserting shifts below

    typedef u32 u32x3 __attribute__((ext_vector_type(3)));
    auto fetch = [&](int y) {
        y = min(y, h + 2);
        int r = y < 0 ? -y : y;
        r = r >= h ? 2 * (h - 1) - r : r;
        return __builtin_amdgcn_raw_buffer_load_b96(srcRs, offS, r * srcPitch, 0);
    };
    auto hrow = [&](const u32x3& rw, u32* hs) {
        const u32 a = rw[0], c = rw[1], b = rw[2];
        const u32 Lw = __builtin_amdgcn_perm(c, a, selL);
        const u32 Cw = __builtin_amdgcn_perm(c, a, selC);
        const u32 Rw = __builtin_amdgcn_perm(c, lastGroup ? a : b, selR);
        hs[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rw, Cw, 1), W2,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Cw, Lw, 1), W1, 0u, false), false);
        hs[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rw, Cw, 2), W2,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Cw, Lw, 2), W1, 0u, false), false);
        hs[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rw, Cw, 3), W2,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Cw, Lw, 3), W1, 0u, false), false);
        hs[3] = __builtin_amdgcn_udot4(Rw, W2, __builtin_amdgcn_udot4(Cw, W1, 0u, false), false);
    };
    // One source row of the next level's resize: horizontal sums of the lane's pixels; `tab` names the (at
    // most two) destination rows whose second tap is this row (all of it scalar).
    auto rz_row = [&](const u32x3& rw, bool emit, const uint4& tab) {
        const u32 lo = firstGroup ? rw[0] : rw[1], hi = firstGroup ? rw[1] : rw[2];
        u32 rcur[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            rcur[j] = __builtin_amdgcn_udot2(as_u16x2(__builtin_amdgcn_perm(hi, lo, rsel[j])), as_u16x2(rcoef[j]), 0u, false) >> 4;
        if (emit) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const u32 w0 = e == 0 ? tab.x : tab.z, w1 = e == 0 ? tab.y : tab.w;
                const int y = (int)(w0 & 0xffffu);
                if (y != 0xffff) {          // scalar branch
                    if ((w0 >> 16) != 0) {  // both taps are this row (clamped border)
#pragma unroll
                        for (int j = 0; j < 4; j++) rprev[j] = rcur[j];
                        asm volatile("" : "+v"(rprev[0]), "+v"(rprev[1]), "+v"(rprev[2]), "+v"(rprev[3]));
                    }
                    const u32 b0 = w1 << 16, b1 = w1 & 0xffff0000u;
                    u32 v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = __umulhi(b0, rprev[j]) + __umulhi(b1, rcur[j]) + 2u;
                    u32 out = v[0] >> 2;
                    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                        : "+v"(out)
                        : "v"(two), "v"(v[1]));
                    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                        : "+v"(out)
                        : "v"(two), "v"(v[2]));
                    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                        : "+v"(out)
                        : "v"(two), "v"(v[3]));
                    const u32 wA = (u32)__builtin_amdgcn_ds_bpermute(addrA, (int)out);
                    const u32 wB = (u32)__builtin_amdgcn_ds_bpermute(addrB, (int)out);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(wB, wA, selG), pyrRs, offP, y * A.dstPitch, 0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) rprev[j] = rcur[j];
    };
    auto row_tab = [&](int s) { return load_table<uint4>(A.st + 2 * min(s, h - 1)); };

    const u32 W01 = (u32)k0 | ((u32)k1 << 16), W23 = (u32)k2 | ((u32)k3 << 16), W21 = (u32)k2 | ((u32)k1 << 16),
              W0h = (u32)k0 << 16;
    u32 win[7][4], lastHs[4];
    u32x3 cur[7];
    {
        u32x3 ini[6];
#pragma unroll
        for (int i = 0; i < 6; i++) ini[i] = fetch(y0 - 3 + i);
#pragma unroll
        for (int i = 0; i < 7; i++) cur[i] = fetch(y0 + 3 + i);
        u32 h0[4], h1[4];
        hrow(ini[0], h0);
#pragma unroll
        for (int i = 1; i < 6; i++) {
            hrow(ini[i], h1);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                win[i][k] = h0[k] | (h1[k] << 16);
                h0[k] = h1[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) lastHs[k] = h0[k];
        if (RESIZE) {
            // the walk proper starts at row y0 + 3: the first band also owns rows 0..2 (the window's rows 3..5),
            // every other band needs the sums of row y0 + 2 as the first "previous row"
            if (y0 == 0) {
                rz_row(ini[3], true, row_tab(0));
                rz_row(ini[4], true, row_tab(1));
                rz_row(ini[5], true, row_tab(2));
            } else {
                rz_row(ini[5], false, make_uint4(0, 0, 0, 0));
            }
        }
    }
    auto load_group = [&](u32x3* grp, int yFirst) {
#pragma unroll
        for (int i = 0; i < 7; i++) grp[i] = fetch(yFirst + i);
    };
    auto run_group = [&](const u32x3* grp, int yb) {
        uint4 tabs[7];
        if (RESIZE) {
#pragma unroll
            for (int ph = 0; ph < 7; ph++) tabs[ph] = row_tab(yb + ph + 3);
        }
#pragma unroll
        for (int ph = 0; ph < 7; ph++) {
            const int y = yb + ph;
            u32 hs[4];
            hrow(grp[ph], hs);
            u32 sv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 P3 = lastHs[k] | (hs[k] << 16);
                win[(ph + 6) % 7][k] = P3;
                lastHs[k] = hs[k];
                u32 s = __builtin_amdgcn_udot2(as_u16x2(win[(ph + 1) % 7][k]), as_u16x2(W01), HALF_EVEN ? 0u : 32768u, false);
                s = __builtin_amdgcn_udot2(as_u16x2(win[(ph + 3) % 7][k]), as_u16x2(W23), s, false);
                s = __builtin_amdgcn_udot2(as_u16x2(win[(ph + 5) % 7][k]), as_u16x2(W21), s, false);
                s = __builtin_amdgcn_udot2(as_u16x2(P3), as_u16x2(W0h), s, false);
                if (HALF_EVEN) s += 0x7FFFu + (((s >> 16) & 1u) | tailUp);
                sv[k] = min(s, 0xFFFFFFu);
            }
            const u32 lo = __builtin_amdgcn_perm(sv[1], sv[0], 0x0c0c0602u);
            const u32 hi = __builtin_amdgcn_perm(sv[3], sv[2], 0x06020c0cu);
            const u32 out = lo | hi;
            if (y < yEnd) {  // scalar branch
                __builtin_amdgcn_raw_buffer_store_b32(out, dstRs, offD, y * dstPitch, 0);
                // row y + 3 is a row of this band's share of the next level unless it is a reflection
                if (RESIZE && y + 3 < h) rz_row(grp[ph], true, tabs[ph]);
            }
        }
    };
    u32x3 other[7];
    int yb = y0;
    while (true) {
        if (yb + 7 < yEnd) load_group(other, yb + 10);
        run_group(cur, yb);
        yb += 7;
        if (yb >= yEnd) break;
        if (yb + 7 < yEnd) load_group(cur, yb + 10);
        run_group(other, yb);
        yb += 7;
        if (yb >= yEnd) break;
    }
}

// ------------------------------------------------------------------ level-0 repack
// Rows of a device-resident image whose base, row stride or frame stride is not a multiple of 4 (a
// 1241-pixel-wide image with contiguous rows, say) are copied once to rows of a 64-byte-aligned pitch,
// so that every kernel of the pass takes its word-aligned path.  A thread moves one 32-bit word: an
// unaligned load from the source row, an aligned store; the last word of a row is assembled from
// bytes so that nothing beyond the row is read.
__global__ __launch_bounds__(256) void k_repack_rows(ImgRef src, int w, u8* __restrict__ dst, int dPitch, i64 dFrame) {
    const int xw = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (4 * xw >= w) return;
    const u8* sp = src.base + (i64)f * src.frameStride + (i64)y * src.pitch + 4 * xw;
    u32 v;
    if (4 * xw + 4 <= w) {
        __builtin_memcpy(&v, sp, 4);
    } else {
        v = 0;
        for (int k = 0; 4 * xw + k < w; k++) v |= (u32)sp[k] << (8 * k);
    }
    *(u32*)(dst + (i64)f * dFrame + (i64)y * dPitch + 4 * xw) = v;
}

// ------------------------------------------------------------------ FAST
__device__ __forceinline__ int wave_excl_count(bool flag, int* total) {
    const u64 m = __ballot(flag);
    *total = __popcll(m);
    const u32 lo = (u32)m, hi = (u32)(m >> 32);
    return __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0));
}

// floor(i / d) for i, d < 65536 with magic = floor(2^32 / d) + 1 (d >= 2).
__device__ __forceinline__ u32 fastdiv(u32 i, u32 d, u32 magic) { return d == 1 ? i : __umulhi(i, magic); }
__host__ __device__ __forceinline__ u32 div_magic(u32 d) { return d <= 1 ? 0u : (u32)(0x100000000ull / d) + 1u; }

// Per-cell constants, computed on the host when the geometry is set up so that a wave
// starts without divisions, level searches or pyramid address arithmetic.
//   w0 = level | roi width << 8 | roi height << 16 | (iniX & 3) << 24
//   w1 = iniX | iniY << 16
//   w2 = slot offset inside the frame's slot block
//   w3 = slot capacity | nk << 16 | (64 / nk) << 24
//   w4 = gx0 | gy0 << 16 (detection-origin coordinates of the first interior pixel)
//   w5 = magic(nWc)   w6 = magic(nk)   w7 = nWc | (64 / nWc) << 8
//   w8 = byte offset of (iniY, iniX & ~3) inside the per-frame pyramid block (levels >= 1)
//   w9 = row pitch of the level (levels >= 1)   w10 = magic(roi width)
//   w11 = nFull | rem << 8: of the first 8 row groups of the ROI copy, groups below nFull hold a row
//         for every copying lane and group nFull holds rem rows
// nk = 32-bit words per row that hold interior pixels when interior pixel 0 sits on a word
// boundary, nWc = nk + 2 = words copied per ROI row (one before, one after).
//   w12 = the cell's index (cell counts and the quadtree number cells level by level; the records
//         are ordered group by group, see FastGroup)   w13..15 = 0
struct FastCell {
    u32 w[16];
};

// The cells of a pyramid are launched in up to two groups of levels, each with the LDS carve-up that
// its own largest cell needs: LDS is handed out in granules, and the levels whose buffers fit 5120
// bytes run 32 waves per compute unit where the largest cells of the pyramid would hold every level
// to 25 (640x480: levels 0..3, 85 % of the cells).
struct FastGroup {
    int recBegin, nCells;                       // this group's slice of the cell records
    int roiPitch, planeOff, queueOff, ldsBytes;  // ROI rows | strength plane (also the word queue) | pixel queue
    u32 roiMagic;                               // div_magic(roiPitch)
    int copyRows;                               // ROI rows per load instruction of the copy: 64 / (roiPitch / 4)
    u32 copyMagic;                              // div_magic(roiPitch / 4)
};

// v_pk_minimum3_f16 / v_pk_maximum3_f16: three-input minimum / maximum on both 16-bit halves.
// Used on bit patterns 0x0401..0x05FF only: positive normal half floats of one exponent, whose
// order is the order of the patterns, so the result is the integer minimum / maximum of each half.
__device__ __forceinline__ u32 pk_min3h(u32 a, u32 b, u32 c) {
    u32 r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ u32 pk_max3h(u32 a, u32 b, u32 c) {
    u32 r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Inclusive prefix sum over the 64 lanes with DPP adds: Hillis-Steele inside each row of 16
// lanes, then the row totals handed on with the two row broadcasts.
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

#ifdef ORBX_FAST_DIAG
// Diagnostic build only (tools/probe/fast_phases.py): every wave writes one record of 8 words:
// cycles of the five phases (copy, pair test, expansion + strength, NMS, tail), then
// n1 | n2 << 32, survivors | second-threshold << 32, flagged words.  Plain stores: atomics on
// shared counters would dominate the kernel.
__device__ unsigned long long* d_fastDiag;
#define FAST_STAMP(i)                                                        \
    do {                                                                     \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
        diagT[i] += now_ - diagLast;                                         \
        diagLast = now_;                                                     \
    } while (0)
#else
#define FAST_STAMP(i)
#endif

// LDS accesses of k_fast by LDS address (the address of its dynamic array + a byte offset): written
// this way a read is "base register + immediate offset", and the array's address joins the other
// constants of the base register instead of costing an addition per access.
#define LDS_U8(a) (*(const __attribute__((address_space(3))) u8*)(size_t)(u32)(a))
#define LDS_U32(a) (*(const __attribute__((address_space(3))) u32*)(size_t)(u32)(a))
#define LDS_U32W(a) (*(__attribute__((address_space(3))) u32*)(size_t)(u32)(a))

// One wave per cell.  The ROI goes to LDS with interior pixel 0 on a word boundary (byte 4 of
// its row), then a cascade, all of it in raster order:
//   1. a necessary condition for a FAST-9 corner on 4 pixels per lane: with u/d/l/r the ring
//      pixels 3 above, below, left and right of v, a 9-arc holds one pixel of each opposite
//      pair, so (u or d darker than v - th) and (l or r darker), or the same with brighter,
//      must hold.  Two pixels per 32-bit operation in 16-bit fields, comparisons as sign bits
//      of biased sums (adds and bitwise operations only, which issue at twice the rate of the
//      packed minimum / maximum forms).  Words with a surviving pixel are queued with a wave
//      ballot (one store per word);
//   2. the word queue is expanded to a pixel queue (byte offset of the pixel in the ROI);
//   3. full 16-arc strength for queued pixels, two pixels per lane in the two halves of
//      packed registers (three-input packed minimum / maximum); pixels above the threshold are
//      written to the strength plane (same geometry as the ROI) and re-queued;
//   4. non-max suppression over the queue; survivors written to the cell's slot range.
// The cascade runs at iniTh and is repeated at minTh only when nothing survived (the
// reference's per-cell fallback).  A single wave executes LDS operations in program order,
// so the phases need no workgroup barrier beyond the compiler-level ordering of
// __syncthreads().
#define FAST_CPW 1  // cells per wave

template <int RPC>
__global__ __launch_bounds__(64) void k_fast(Geom g, ImgRef img0, int img0Aligned, const u8* __restrict__ pyr,
                                              const FastCell* __restrict__ cells, u32* __restrict__ cellCount,
                                              u32* __restrict__ slots, int iniTh, int minTh, FastGroup grp) {
    extern __shared__ __align__(16) u8 lds[];
    // LDS_U8 / LDS_U32 take LDS addresses: the array's own address (a link-time constant, 0 while it is
    // the kernel's only LDS object) is folded into every base register below, at no cost
    const u32 L0 = (u32)(size_t)(const __attribute__((address_space(3))) u8*)lds;
    u8* plane = lds + grp.planeOff;
    u32* wq = (u32*)plane;  // the word queue is dead before the plane is cleared
    u16* queue = (u16*)(lds + grp.queueOff);
    // blockIdx.x = frame: with a batch that is a multiple of 8 all cells of a frame run on one
    // XCD, so the halo rows and cache lines that neighbouring cells share come from its L2.
    const int f = blockIdx.x;
    const int rec0 = grp.recBegin + blockIdx.y * FAST_CPW, recEnd = grp.recBegin + grp.nCells;
    const int lane = threadIdx.x;
    // RPC = the ROI row pitch as a compile-time constant (0: taken from the geometry): with a constant
    // pitch every ring, neighbour and row access below is one base register plus an immediate offset
    const int RP = RPC ? RPC : grp.roiPitch;
    (void)img0Aligned;  // the copy below takes any source alignment
#ifdef ORBX_FAST_DIAG
    unsigned long long diagT[5] = {0, 0, 0, 0, 0}, diagLast = __builtin_amdgcn_s_memtime();
    int diagN1 = 0, diagN2 = 0, diagPass2 = 0, diagNW = 0, diagTotal = 0;
#endif
    // ROI -> LDS: ROI column c lands on byte c + 1 of its LDS row, i.e. word wd of LDS row y holds the level's
    // bytes (iniX - 1 + 4 * wd ...) of row iniY + y.  The words go from global memory straight into the LDS
    // (direct-to-LDS loads take any byte address, so the shift that puts interior pixel 0 on a word boundary
    // costs nothing and source rows need no alignment): lane = (row lw, word wd) of a group of copyRows rows,
    // a group per load instruction, its 64 words consecutive in the LDS.  No registers, no vector ALU work
    // beyond the lane's offset.
    const int cpW = RP >> 2, cpRows = RPC ? 64 / (RPC ? RPC >> 2 : 1) : grp.copyRows;
    const int cpLw = (int)fastdiv((u32)lane, (u32)cpW, RPC ? div_magic((u32)(RPC ? RPC >> 2 : 1)) : grp.copyMagic), cpWd = lane - cpLw * cpW;
    auto copy_roi = [&](const FastCell& C) {
        const int rh = (C.w[0] >> 16) & 255, nWc = (int)(C.w[7] & 255);
        const u8* sb;
        int pitch;
        if ((C.w[0] & 255u) > 0u) {
            pitch = (int)C.w[9];
            sb = pyr + (i64)f * g.pyrFrameBytes + C.w[8] + (i64)(C.w[0] >> 24) - 1;
        } else {
            pitch = img0.pitch;
            sb = img0.base + (i64)f * img0.frameStride + (i64)(C.w[1] >> 16) * img0.pitch + (i64)(C.w[1] & 0xffff) - 1;
        }
        if (cpLw < cpRows && cpWd < nWc) {
            u32 voff = mul24((u32)cpLw, (u32)pitch) + 4u * (u32)cpWd;
            const u32 vstep = (u32)(cpRows * pitch);
            for (int y0 = 0; y0 < rh; y0 += cpRows) {  // uniform
                // a whole group of rows needs no per-lane test
                if (y0 + cpRows <= rh || cpLw < rh - y0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + (size_t)voff),
                                                     (__attribute__((address_space(3))) void*)(size_t)(L0 + (u32)(y0 * RP)), 4, 0, 0);
                voff += vstep;
            }
        }
        // direct-to-LDS loads are not LDS writes to the compiler: wait for them here
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    FastCell C = cells[min(rec0, recEnd - 1)];
    for (int ci = 0; ci < FAST_CPW; ci++) {
        const int rec = rec0 + ci;
        if (rec >= recEnd) break;
        const int cell = (int)C.w[12];
        const int rw = (C.w[0] >> 8) & 255, rh = (C.w[0] >> 16) & 255;
        const int cellCap = C.w[3] & 0xffff;
        u32* cnt = cellCount + (i64)f * g.totalCells + cell;
        const bool empty = rw < 7 || rh < 7;
        const int iw = rw - 6, ih = rh - 6;
        const int nk = (int)((C.w[3] >> 16) & 255), rowsPer = (int)(C.w[3] >> 24);
        __syncthreads();  // the previous cell's phases are through with the LDS
        if (!empty) copy_roi(C);
        __syncthreads();
        const FastCell Cn = cells[min(rec + 1, recEnd - 1)];
        FAST_STAMP(0);
        if (empty) {
            if (lane == 0) *cnt = 0;
            C = Cn;
            continue;
        }

        u32* out = slots + (i64)f * g.totalSlots + C.w[2];
        const int gx0 = C.w[4] & 0xffff, gy0 = C.w[4] >> 16;  // relative to the detection origin
        // Pair test: a lane owns interior word kk of row lr inside a group of rowsPer rows and moves
        // down the interior one group per step, so its addresses advance by a constant and its mask
        // of pixels that lie inside the interior is fixed.
        const int lr = (int)fastdiv((u32)lane, (u32)nk, C.w[6]), kk = lane - lr * nk;
        const bool laneOn = lr < rowsPer;
        // surviving pixels of a word are flagged in bits 14, 15, 30, 31 (pixels 0, 1, 2, 3)
        const int nv = min(max(iw - 4 * kk, 0), 4);  // pixels of the word that lie inside the interior
        const u32 vmAll = (((1u << min(nv, 2)) - 1u) << 14) | (((1u << max(nv - 2, 0)) - 1u) << 30);
        const u32 vmE = vmAll & 0x40004000u, vmO = vmAll & 0x80008000u;
        const u32 entBase = mul24((u32)(lr + 3), (u32)RP) + 4u * (u32)(kk + 1);  // byte offset of the lane's word in the ROI
        const int stepBytes = rowsPer * RP;
        // 16-byte units that cover plane rows 2 .. rh-3 (rows 0, 1 and the rows past rh-3 are never read;
        // the plane block holds fastRoiPitch * (tallest ROI) bytes, rounded up to 16)
        const int planeLo = (2 * RP) >> 4, planeHi = ((rh - 2) * RP + 15) >> 4;
        int total = 0;
        for (int pass = 0; pass < 2; pass++) {
            const int th = pass == 0 ? iniTh : minTh;  // clamped to [0, 255] by the host
            // notdark(x) = bit 15 of x + (0x8000 + th - v); bright(x) = bit 15 of x + (0x8000 - th - 1 - v)
            const u32 k1 = 0x8000u + (u32)th, k2 = 0x8000u - (u32)th - 1u;
            const u32 K1 = k1 | (k1 << 16), K2 = k2 | (k2 << 16);
            int nW = 0;
            {
                u32 r0 = entBase + (L0 - (u32)(3 * RP + 4));
                u32 ent = entBase;
                for (int yb = 0; yb < ih; yb += rowsPer) {
                    u32 m = 0;
                    if (laneOn && yb + lr < ih) {
                        // r0 = lowest address read: the offsets are non-negative immediates
                        const u32 Uw = LDS_U32(r0 + 4u), Lw = LDS_U32(r0 + (u32)(3 * RP)), Cw = LDS_U32(r0 + (u32)(3 * RP + 4));
                        const u32 Rw = LDS_U32(r0 + (u32)(3 * RP + 8)), Dw = LDS_U32(r0 + (u32)(6 * RP + 4));
                        const u32 Rt = __builtin_amdgcn_alignbyte(Rw, Cw, 3);  // pixels x + 3
                        const u32 Lf = __builtin_amdgcn_alignbyte(Cw, Lw, 1);  // pixels x - 3
                        const u32 M = 0x00FF00FFu;
                        u32 pe, po;
                        {
                            const u32 c = Cw & M, wd = K1 - c, wb = K2 - c;
                            const u32 u = Uw & M, d = Dw & M, lf = Lf & M, rt = Rt & M;
                            const u32 nd = ((u + wd) & (d + wd)) | ((lf + wd) & (rt + wd));
                            const u32 br = ((u + wb) | (d + wb)) & ((lf + wb) | (rt + wb));
                            pe = ~nd | br;
                        }
                        {
                            const u32 c = (Cw >> 8) & M, wd = K1 - c, wb = K2 - c;
                            const u32 u = (Uw >> 8) & M, d = (Dw >> 8) & M, lf = (Lf >> 8) & M, rt = (Rt >> 8) & M;
                            const u32 nd = ((u + wd) & (d + wd)) | ((lf + wd) & (rt + wd));
                            const u32 br = ((u + wb) | (d + wb)) & ((lf + wb) | (rt + wb));
                            po = ~nd | br;
                        }
                        m = ((pe >> 1) & vmE) | (po & vmO);
                    }
                    int tot;
                    const int off = wave_excl_count(m != 0, &tot);
                    if (m != 0) wq[nW + off] = m | ent;
                    nW += tot;
                    r0 += (u32)stepBytes;
                    ent += (u32)stepBytes;
                }
            }
            __syncthreads();
            FAST_STAMP(1);
            // word queue -> pixel queue, still raster order
            int n1 = 0;
            for (int base = 0; base < nW; base += 64) {
                const int wi = base + lane;
                const u32 w = wi < nW ? wq[wi] : 0u;
                const int c = __popc(w & 0xC000C000u);
                const int inc = wave_incl_scan_dpp(c);
                const int tot = __builtin_amdgcn_readlane(inc, 63);
                const u32 o = w & 0x3FFFu;
                u16* qp = queue + n1 + inc - c;
                if (w & 0x4000u) *qp++ = (u16)o;
                if (w & 0x8000u) *qp++ = (u16)(o + 1);
                if (w & 0x40000000u) *qp++ = (u16)(o + 2);
                if (w & 0x80000000u) *qp = (u16)(o + 3);
                n1 += tot;
            }
            __syncthreads();
            for (int i = planeLo + lane; i < planeHi; i += 64) ((uint4*)plane)[i] = make_uint4(0, 0, 0, 0);
            __syncthreads();
            // strength: lanes take queue entries base + lane (low halves) and base + 64 + lane (high halves)
            int n2 = 0;
            const u32 safe = (u32)(3 * RP + 4);  // interior pixel (0, 0): its ring stays inside the ROI
            for (int base = 0; base < n1; base += 128) {
                const int ia = base + lane, ib = base + 64 + lane;
                const bool okA = ia < n1, okB = ib < n1;
                const u32 oa = okA ? (u32)queue[ia] : safe, ob = okB ? (u32)queue[ib] : safe;
                // lowest ring address of each candidate, hidden from the optimiser so that the 16 ring reads
                // stay "this register + a non-negative immediate" (it would fold the constant back and
                // spend an addition on every offset that comes out negative)
                u32 pa = oa + (L0 - (u32)(3 * RP + 3)), pb = ob + (L0 - (u32)(3 * RP + 3));
                asm("" : "+v"(pa), "+v"(pb));
#define RING2(dx, dy) ((u32)LDS_U8(pa + (u32)(((dy) + 3) * RP + (dx) + 3)) | ((u32)LDS_U8(pb + (u32)(((dy) + 3) * RP + (dx) + 3)) << 16))
                // D[k] = v - ring[k] + 0x500 in each half (0x401 .. 0x5FF)
                const u32 V = RING2(0, 0) + 0x05000500u;
                u32 D[16];
                D[0] = V - RING2(0, 3);
                D[1] = V - RING2(1, 3);
                D[2] = V - RING2(2, 2);
                D[3] = V - RING2(3, 1);
                D[4] = V - RING2(3, 0);
                D[5] = V - RING2(3, -1);
                D[6] = V - RING2(2, -2);
                D[7] = V - RING2(1, -3);
                D[8] = V - RING2(0, -3);
                D[9] = V - RING2(-1, -3);
                D[10] = V - RING2(-2, -2);
                D[11] = V - RING2(-3, -1);
                D[12] = V - RING2(-3, 0);
                D[13] = V - RING2(-3, 1);
                D[14] = V - RING2(-2, 2);
                D[15] = V - RING2(-1, 3);
#undef RING2
                u32 mn3[16], mx3[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    mn3[k] = pk_min3h(D[k], D[(k + 1) & 15], D[(k + 2) & 15]);
                    mx3[k] = pk_max3h(D[k], D[(k + 1) & 15], D[(k + 2) & 15]);
                }
                u32 mn9[16], mx9[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    mn9[k] = pk_min3h(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]);
                    mx9[k] = pk_max3h(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]);
                }
                // dark = max over arcs of the arc minimum; bright_neg = min over arcs of the arc maximum
                u32 dk = pk_max3h(mn9[0], mn9[1], mn9[2]), bn = pk_min3h(mx9[0], mx9[1], mx9[2]);
#pragma unroll
                for (int k = 3; k < 15; k += 2) {
                    dk = pk_max3h(dk, mn9[k], mn9[k + 1]);
                    bn = pk_min3h(bn, mx9[k], mx9[k + 1]);
                }
                dk = pk_max3h(dk, mn9[15], mn9[15]);
                bn = pk_min3h(bn, mx9[15], mx9[15]);
                // strength = max(dark - 0x500, 0x500 - bright_neg) in each half
                const i16x2 K = (i16x2)(0x500);
                i16x2 dkv, bnv;
                __builtin_memcpy(&dkv, &dk, 4);
                __builtin_memcpy(&bnv, &bn, 4);
                const i16x2 st = __builtin_elementwise_max((i16x2)(dkv - K), (i16x2)(K - bnv));
                const int ma = (int)st.x, mb = (int)st.y;
                const bool ca = okA && ma > th, cb = okB && mb > th;
                if (ca) plane[oa] = (u8)ma;
                if (cb) plane[ob] = (u8)mb;
                int ta, tb;
                const int offA = wave_excl_count(ca, &ta);
                const int offB = wave_excl_count(cb, &tb);
                __syncthreads();  // every lane has read its queue entries before the in-place compaction writes
                if (ca) queue[n2 + offA] = (u16)oa;
                if (cb) queue[n2 + ta + offB] = (u16)ob;
                n2 += ta + tb;
            }
            __syncthreads();
            FAST_STAMP(2);
#ifdef ORBX_FAST_DIAG
            diagN1 += n1;
            diagN2 += n2;
            diagPass2 += pass;
            diagNW += nW;
#endif
            // Non-max suppression.  Every plane entry is either 0 or a strength above th, so a pixel of
            // strength m (response m - 1) beats a neighbour of strength q iff m - 1 > max(q - 1, 0),
            // i.e. m > max(q, 1): one maximum over the 8 neighbours.
            for (int base = 0; base < n2; base += 64) {
                const int wi = base + lane;
                bool keep = false;
                u32 o = safe;
                int s = 0;
                if (wi < n2) {
                    o = queue[wi];
                    u32 q0 = o + (L0 + (u32)grp.planeOff - (u32)(RP + 1));  // top-left neighbour in the plane
                    asm("" : "+v"(q0));
#define PL(k) ((int)LDS_U8(q0 + (u32)(k)))
                    const int m = PL(RP + 1);
                    const int nb = max(max(max(PL(RP), PL(RP + 2)), max(PL(0), PL(1))),
                                       max(max(PL(2), PL(2 * RP)), max(PL(2 * RP + 1), PL(2 * RP + 2))));
#undef PL
                    keep = m > max(nb, 1);
                    s = m - 1;
                }
                int tot;
                const int off = wave_excl_count(keep, &tot);
                if (keep && total + off < cellCap) {
                    const int yy = RPC ? (int)(o / (u32)(RPC ? RPC : 1)) : (int)__umulhi(o, grp.roiMagic);
                    const int xx = (int)o - yy * RP;  // ROI row, byte in row
                    out[total + off] = ORBX_PACK(gx0 + xx - 4, gy0 + yy - 3, s);
                }
                total += tot;
            }
            FAST_STAMP(3);
            if (total > 0) break;
        }
        if (lane == 0) *cnt = (u32)min(total, cellCap);
#ifdef ORBX_FAST_DIAG
        diagTotal += total;
#endif
        C = Cn;
    }
#ifdef ORBX_FAST_DIAG
    FAST_STAMP(4);
    if (lane == 0 && d_fastDiag) {
        unsigned long long* r = d_fastDiag + ((size_t)f * g.totalCells + rec0) * 8;
        for (int i = 0; i < 5; i++) r[i] = diagT[i];
        r[5] = (unsigned long long)(u32)diagN1 | ((unsigned long long)(u32)diagN2 << 32);
        r[6] = (unsigned long long)(u32)diagTotal | ((unsigned long long)(u32)diagPass2 << 32);
        r[7] = (unsigned long long)(u32)diagNW;
    }
#endif
}

// ------------------------------------------------------------------ quadtree
// Inclusive prefix sum over the wave (all lanes active): the DPP form, six dependent additions, where
// the shuffle form paid a round trip through the LDS crossbar per step.
__device__ __forceinline__ int wave_incl_scan(int v) { return wave_incl_scan_dpp(v); }

// Threads per quadtree workgroup.  Measured on MI355X at 640x480 / 1000 features: 256 is best (128 and 512
// slower, a single wave much slower); levels with thousands of nodes and 10^5 candidates want more.

// Exclusive scan over the QT_THREADS threads of the block; *total = block sum.
// wsum: 2 x 16 partial sums, the two halves used by alternate calls (every thread of the block makes the
// same sequence of calls): the readers of a half have all passed the barrier of the next call before
// the call after that writes it again, so one barrier per scan is enough.
template <int QT_THREADS>
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* wsumBoth, int& phase) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int inc = wave_incl_scan(v);
    if (QT_THREADS == 64) {
        *total = __builtin_amdgcn_readlane(inc, 63);
        return inc - v;
    }
    int* wsum = wsumBoth + 16 * phase;
    phase ^= 1;
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < QT_THREADS / 64; k++) {
        const int s = wsum[k];
        if (k < wv) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

#define QT_KPT 8  // keys per thread that a level may hold in registers

struct QtShared {
    int n, L, prevL, tnew, ncut, nproc, toExpand, done, careful, C;
    int wsum[32];
};

__device__ __forceinline__ int child_quadrant(u32 key, short4 b) {
    // b = (x0, x1, y0, y1); split at x0 + ceil((x1-x0)/2), y0 + ceil((y1-y0)/2)
    const int xm = b.x + ((b.y - b.x + 1) >> 1);
    const int ym = b.z + ((b.w - b.z + 1) >> 1);
    const int x = ORBX_PX(key), y = ORBX_PY(key);
    return (x < xm ? 0 : 1) + (y < ym ? 0 : 2);
}

__device__ __forceinline__ short4 child_box(short4 b, int q) {
    const short xm = (short)(b.x + ((b.y - b.x + 1) >> 1));
    const short ym = (short)(b.z + ((b.w - b.z + 1) >> 1));
    short4 r;
    r.x = (q & 1) ? xm : b.x;
    r.y = (q & 1) ? b.y : xm;
    r.z = (q & 2) ? ym : b.z;
    r.w = (q & 2) ? b.w : ym;
    return r;
}

#ifdef ORBX_QT_DIAG
// Diagnostic build: thread 0 of the level-0 workgroup of frame 0 prints the cycles between phase marks.
#define QT_MARK(name)                                                                         \
    do {                                                                                      \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                         \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                     \
            printf("qt %s %u L %d nkeys %d\n", name, (unsigned)(now_ - qtLast), S.L, nkeys);  \
            qtLast = __builtin_amdgcn_s_memtime();                                            \
        }                                                                                     \
    } while (0)
#else
#define QT_MARK(name)
#endif

// DistributeOctTree for one (frame, level).  The list of nodes is kept as
// arrays indexed by list position; every pass rebuilds the list with prefix
// sums so that positions, creation order and the stop rule equal those of the
// sequential algorithm.  Node "addresses" in the size sort are modelled by
// creation order (later created = greater).
// GTAB = false: the node tables live in LDS (72 bytes per list entry); true: in a slice of global
// memory per (frame, level), for feature counts whose tables do not fit the LDS.
template <int QT_THREADS, bool GTAB>
__global__ __launch_bounds__(QT_THREADS) void k_quadtree(Geom g, const u32* __restrict__ cellCount,
                                                   const u32* __restrict__ slots, u32* keyPacked,
                                                   u16* keyNode, u32* sel, int* selCount,
                                                   int* errFlag, unsigned char* nodeTables, size_t nodeTableStride) {
    extern __shared__ __align__(16) unsigned char smemLds[];
    __shared__ QtShared S;
    // Workgroups are dispatched in linear order: with the level as the slow index all workgroups of the
    // large levels start first and the small levels fill the end of the launch, instead of every round
    // of resident workgroups lasting as long as its level-0 members.
    const int l = blockIdx.y, f = blockIdx.x;
    const LevelGeom& LG = g.lv[l];
    const int tid = threadIdx.x;
    int scanPhase = 0;  // which half of S.wsum the next block scan uses
#ifdef ORBX_QT_DIAG
    unsigned long long qtLast = __builtin_amdgcn_s_memtime();
#endif
    const int lcap = LG.lcap;
    const int N = LG.quota;
    int* selCnt = selCount + (i64)f * ORBX_MAX_LEVELS + l;
    u32* selOut = sel + (i64)f * g.totalSel + LG.selBase;

    // LDS carve-up (all arrays have lcap entries unless noted)
    // (the two copies of a table are named, not an array of pointers indexed by the pass: the compiler
    // then sees that they point into the LDS and reads them with LDS instructions instead of flat ones)
    unsigned char* p = smemLds;
    if (GTAB) p = nodeTables + ((size_t)f * g.nlevels + l) * nodeTableStride;
    short4* const box0 = (short4*)p; p += (size_t)lcap * 8;
    short4* const box1 = (short4*)p; p += (size_t)lcap * 8;
    u32* childCnt = (u32*)p; p += (size_t)lcap * 16;  // 4 per node; reused as u64 best[]
    u32* const cnt0 = (u32*)p; p += (size_t)lcap * 4;
    u32* const cnt1 = (u32*)p; p += (size_t)lcap * 4;
    u32* const seq0 = (u32*)p; p += (size_t)lcap * 4;
    u32* const seq1 = (u32*)p; p += (size_t)lcap * 4;
    u32* gpre = (u32*)p; p += (size_t)lcap * 4;     // by rank: exclusive prefix of children
    u16* newId = (u16*)p; p += (size_t)lcap * 8;    // 4 per node: list position of each child
    u16* moved = (u16*)p; p += (size_t)lcap * 2;    // new position of a node that stays
    u16* rankOf = (u16*)p; p += (size_t)lcap * 2;   // processing rank of a node (0xFFFF: none)
    u16* nodeOfRank = (u16*)p; p += (size_t)lcap * 2;
    u8* mOf = (u8*)p; p += (size_t)lcap;            // non-empty children of a node

    // ---- gather the level's candidates into a dense array (reference order:
    // cell row-major, raster inside a cell)
    const u32* cc = cellCount + (i64)f * g.totalCells + LG.cellBase;
    const u32* sl = slots + (i64)f * g.totalSlots + LG.slotBase;
    u32* kp = keyPacked + (i64)f * g.totalSlots + LG.slotBase;
    u16* kn = keyNode + (i64)f * g.totalSlots + LG.slotBase;
    int nkeys = 0;
    for (int base = 0; base < LG.nCells; base += QT_THREADS) {
        const int c = base + tid;
        const int cN = c < LG.nCells ? (int)cc[c] : 0;
        int tot;
        const int off = block_excl_scan<QT_THREADS>(cN, &tot, S.wsum, scanPhase);
        // eight entries of the cell per trip, their loads issued together (a cell holds 6 on average: one
        // wait for memory instead of one per entry)
        const u32* cs = sl + (i64)c * LG.cellCap;
        u32* cd = kp + nkeys + off;
        for (int k0 = 0; k0 < cN; k0 += 8) {
            u32 v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = cs[min(k0 + i, cN - 1)];
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (k0 + i < cN) cd[k0 + i] = v[i];
        }
        nkeys += tot;
    }
    __syncthreads();
    QT_MARK("gather");
    if (nkeys == 0) {
        if (tid == 0) *selCnt = 0;
        return;
    }
    // Levels with at most QT_KPT keys per thread (every level of a 640x480 frame) keep a thread's keys and
    // their nodes in registers for the whole subdivision: the passes below then touch no global memory,
    // where each of them paid a round trip for the keys and another for the node numbers.  Key j of thread
    // tid is key tid + j * QT_THREADS of the level.  Larger levels keep both arrays in global memory.
    // Both forms are instances of the code below (REG: a compile-time constant inside it).
    auto subdivide = [&](auto regTag) {
    constexpr bool REG = decltype(regTag)::value;
    u32 rkey[REG ? QT_KPT : 1], rnode[REG ? QT_KPT : 1];
    if constexpr (REG) {
#pragma unroll
        for (int j = 0; j < QT_KPT; j++) {
            rkey[j] = kp[min(tid + j * QT_THREADS, nkeys - 1)];
            rnode[j] = 0;
        }
    }

    // ---- roots (reference :545-587)
    const int nIni = LG.nIni;
    const float hX = LG.hX;
    const int ySpan = LG.h - 2 * MIN_BORDER;
    for (int i = tid; i < lcap; i += QT_THREADS) cnt0[i] = 0;
    __syncthreads();
    // With one or two initial nodes (the usual aspect ratios) every key of a wave would hit the same one
    // or two LDS counters: the wave counts its keys per node with ballots and adds once per node.
    const bool fewRoots = nIni <= 2;
    auto count_root = [&](int r) {  // r = -1: no key; called by every thread of a wave together
        if (fewRoots) {
            const u64 m0 = __ballot(r == 0), m1 = __ballot(r == 1);
            if ((tid & 63) == 0) {
                if (m0) atomicAdd(&cnt0[0], (u32)__popcll(m0));
                if (m1) atomicAdd(&cnt0[1], (u32)__popcll(m1));
            }
        } else if (r >= 0) {
            atomicAdd(&cnt0[r], 1u);
        }
    };
    if constexpr (REG) {
#pragma unroll
        for (int j = 0; j < QT_KPT; j++) {
            if (j * QT_THREADS < nkeys) {  // uniform
                const int k = tid + j * QT_THREADS;
                const int r = k < nkeys ? (int)((float)ORBX_PX(rkey[j]) / hX) : -1;
                rnode[j] = (u32)r;
                count_root(r);
            }
        }
    } else {
        for (int k0 = 0; k0 < nkeys; k0 += QT_THREADS) {
            const int k = k0 + tid;
            const int r = k < nkeys ? (int)((float)ORBX_PX(kp[k]) / hX) : -1;
            if (k < nkeys) kn[k] = (u16)r;
            count_root(r);
        }
    }
    __syncthreads();
    if (tid == 0) {
        // compact non-empty roots, keeping order; moved[r] = list position
        int L = 0;
        for (int r = 0; r < nIni; r++) {
            const u32 c = cnt0[r];
            if (c > 0) {
                moved[r] = (u16)L;
                short4 b;
                b.x = (short)(int)(hX * (float)r);
                b.y = (short)(int)(hX * (float)(r + 1));
                b.z = 0;
                b.w = (short)ySpan;
                box1[L] = b;
                cnt1[L] = c;
                seq1[L] = (u32)r;
                L++;
            }
        }
        S.L = L;
        S.careful = 0;
        S.done = 0;
    }
    __syncthreads();
    // Every pass needs, per node that can be split, the number of its keys in each quadrant.  The counts
    // of a pass are gathered while the keys are moved to the nodes of that pass (here for the roots,
    // at the end of a pass for the next one), so a pass sweeps the keys once, not twice.
    for (int i = tid; i < S.L * 4; i += QT_THREADS) childCnt[i] = 0;
    __syncthreads();
    auto count_code = [&](int code) {  // code = node * 4 + quadrant of a key that is counted, or -1
        if (fewRoots) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const u64 m = __ballot(code == c);
                if (m && (tid & 63) == 0) atomicAdd(&childCnt[c], (u32)__popcll(m));
            }
        } else if (code >= 0) {
            atomicAdd(&childCnt[code], 1u);
        }
    };
    if constexpr (REG) {
#pragma unroll
        for (int j = 0; j < QT_KPT; j++) {
            if (j * QT_THREADS < nkeys) {
                const int k = tid + j * QT_THREADS;
                int code = -1;
                if (k < nkeys) {
                    const int nd = moved[rnode[j]];
                    rnode[j] = (u32)nd;
                    if (cnt1[nd] > 1) code = nd * 4 + child_quadrant(rkey[j], box1[nd]);
                }
                count_code(code);
            }
        }
    } else {
        for (int k0 = 0; k0 < nkeys; k0 += QT_THREADS) {
            const int k = k0 + tid;
            int code = -1;
            if (k < nkeys) {
                const int nd = moved[kn[k]];
                kn[k] = (u16)nd;
                if (cnt1[nd] > 1) code = nd * 4 + child_quadrant(kp[k], box1[nd]);
            }
            count_code(code);
        }
    }
    int cur = 1;  // buffer holding the current list
    u32 nextSeq = (u32)nIni;
    __syncthreads();
    QT_MARK("roots");

    // ---- subdivision passes (reference :596-741)
    for (int iter = 0; iter < 64; iter++) {
        const int L = S.L;
        const int careful = S.careful;
        const short4* cb = (cur ? box1 : box0);
        const u32* ccnt = (cur ? cnt1 : cnt0);
        const u32* cseq = (cur ? seq1 : seq0);
        const int nxt = cur ^ 1;
        short4* const xb = nxt ? box1 : box0;
        u32* const xcnt = nxt ? cnt1 : cnt0;
        u32* const xseq = nxt ? seq1 : seq0;
        // (childCnt holds the quadrant counts of this pass's nodes: gathered when the keys were moved here)
        // processing order of expandable nodes
        int C = 0;
        if (!careful) {
            for (int base = 0; base < L; base += QT_THREADS) {
                const int i = base + tid;
                const int e = (i < L && ccnt[i] > 1) ? 1 : 0;
                int tot;
                const int off = block_excl_scan<QT_THREADS>(e, &tot, S.wsum, scanPhase);
                if (i < L) rankOf[i] = e ? (u16)(C + off) : (u16)0xFFFF;
                C += tot;
            }
        } else {
            // Rank by (count desc, creation order desc).  Every node that can still be split was pushed
            // to the front of the list when it was made (only the initial nodes were appended, and those
            // with more than one key are all split in the first pass), so among the ranked nodes a later
            // creation means an earlier list position: the order is (count desc, position asc).  A
            // bitonic sort of count << 32 | ~position over the next power of two; the sort buffer lies
            // over gpre / newId / moved / rankOf, which are dead here (16 bytes per list entry = room for
            // twice the list), and only its zero padding overlaps rankOf.
            u64* srt = (u64*)gpre;
            int P = 2;
            while (P < L) P <<= 1;
            if (P <= QT_THREADS) {
                // One element per thread, kept in a register: a compare-exchange whose partner sits in the
                // same wave (distance below 64) is a shuffle, only the longer distances go through LDS
                // and a barrier (3 of the 36 steps for 256 elements).  Element i of a pair (i, i ^ j) keeps
                // the larger value when it is the lower index of a descending block or the upper index of
                // an ascending one.
                u64 val = (tid < L && ccnt[tid] > 1) ? (((u64)ccnt[tid] << 32) | (u64)(0xFFFFFFFFu - (u32)tid)) : 0ull;
                for (int k = 2; k <= P; k <<= 1) {
                    const bool desc = (tid & k) == 0;   // descending blocks first: the whole ends descending
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        u64 other;
                        if (j >= 64) {                   // uniform over the block
                            __syncthreads();             // the previous exchange has been read
                            if (tid < P) srt[tid] = val;
                            __syncthreads();
                            other = tid < P ? srt[tid ^ j] : 0ull;
                        } else {
                            other = __shfl_xor(val, j);
                        }
                        const bool lower = (tid & j) == 0;
                        const u64 hi = val > other ? val : other, lo = val > other ? other : val;
                        val = (desc == lower) ? hi : lo;
                    }
                }
                __syncthreads();  // the last exchange buffer has been read: rankOf may overlap its padding
                for (int i = tid; i < L; i += QT_THREADS) rankOf[i] = 0xFFFF;
                __syncthreads();
                if (tid < P && val != 0ull) rankOf[0xFFFFFFFFu - (u32)val] = (u16)tid;
            } else {
            for (int i = tid; i < P; i += QT_THREADS)
                srt[i] = (i < L && ccnt[i] > 1) ? (((u64)ccnt[i] << 32) | (u64)(0xFFFFFFFFu - (u32)i)) : 0ull;
            __syncthreads();
            for (int k = 2; k <= P; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < P; i += QT_THREADS) {
                        const int x = i ^ j;
                        if (x > i) {
                            const u64 a = srt[i], b = srt[x];
                            const bool desc = (i & k) == 0;   // descending blocks first: the whole ends descending
                            if (desc ? a < b : a > b) {
                                srt[i] = b;
                                srt[x] = a;
                            }
                        }
                    }
                    __syncthreads();
                }
            }
            for (int i = tid; i < L; i += QT_THREADS) rankOf[i] = 0xFFFF;
            __syncthreads();
            for (int r = tid; r < L; r += QT_THREADS) {
                const u64 key = srt[r];
                if (key != 0ull) rankOf[0xFFFFFFFFu - (u32)key] = (u16)r;
            }
            }
            for (int base = 0; base < L; base += QT_THREADS) {
                const int i = base + tid;
                const int e = (i < L && ccnt[i] > 1) ? 1 : 0;
                int tot;
                block_excl_scan<QT_THREADS>(e, &tot, S.wsum, scanPhase);
                C += tot;
            }
        }
        __syncthreads();
        QT_MARK(careful ? "sort+rank" : "rank");
        for (int i = tid; i < L; i += QT_THREADS) {
            const u16 r = rankOf[i];
            int m = 0;
            if (r != 0xFFFF) {
                nodeOfRank[r] = (u16)i;
                m = (childCnt[i * 4] > 0) + (childCnt[i * 4 + 1] > 0) + (childCnt[i * 4 + 2] > 0) +
                    (childCnt[i * 4 + 3] > 0);
            }
            mOf[i] = (u8)m;
        }
        __syncthreads();
        // scan over ranks: children created before rank j, and the cut
        int run = 0, below = 0;
        for (int base = 0; base < C; base += QT_THREADS) {
            const int j = base + tid;
            const int m = j < C ? (int)mOf[nodeOfRank[j]] : 0;
            int tot;
            const int off = block_excl_scan<QT_THREADS>(m, &tot, S.wsum, scanPhase);
            int lt = 0;
            if (j < C) {
                gpre[j] = (u32)(run + off);
                // list size after processing ranks 0..j (each split removes the parent)
                lt = (L + (run + off + m) - (j + 1)) < N ? 1 : 0;
            }
            int tl;
            block_excl_scan<QT_THREADS>(lt, &tl, S.wsum, scanPhase);
            below += tl;
            run += tot;
        }
        int nproc = C;
        if (careful) nproc = min(C, below + 1);
        __syncthreads();
        QT_MARK("cutscan");
        int tnew = 0;
        if (nproc > 0) {
            const int last = nodeOfRank[nproc - 1];
            tnew = (int)gpre[nproc - 1] + (int)mOf[last];
        }
        // nodes that stay: new position = tnew + rank among stayers in list order
        int stay = 0;
        for (int base = 0; base < L; base += QT_THREADS) {
            const int i = base + tid;
            const int s = (i < L && (rankOf[i] == 0xFFFF || rankOf[i] >= nproc)) ? 1 : 0;
            int tot;
            const int off = block_excl_scan<QT_THREADS>(s, &tot, S.wsum, scanPhase);
            if (s) {
                const int np = tnew + stay + off;
                moved[i] = (u16)np;
                if (np < lcap) {
                    xb[np] = cb[i];
                    xcnt[np] = ccnt[i];
                    xseq[np] = cseq[i];
                }
            }
            stay += tot;
        }
        // children of processed nodes, pushed to the front in processing order
        int expand = 0;
        for (int i = tid; i < L; i += QT_THREADS) {
            const u16 r = rankOf[i];
            if (r != 0xFFFF && r < nproc) {
                int gidx = (int)gpre[r];
                for (int q = 0; q < 4; q++) {
                    const u32 c = childCnt[i * 4 + q];
                    if (c > 0) {
                        const int np = tnew - 1 - gidx;
                        newId[i * 4 + q] = (u16)np;
                        if (np < lcap) {
                            xb[np] = child_box(cb[i], q);
                            xcnt[np] = c;
                            xseq[np] = nextSeq + (u32)gidx;
                        }
                        if (c > 1) expand++;
                        gidx++;
                    }
                }
            }
        }
        {
            int tot;
            block_excl_scan<QT_THREADS>(expand, &tot, S.wsum, scanPhase);
            expand = tot;
        }
        QT_MARK("stay+child");
        // (the barrier inside the scan: every thread is through with this pass's quadrant counts)
        const int newL = tnew + stay;
        const int nextL = min(newL, lcap);
        for (int i = tid; i < nextL * 4; i += QT_THREADS) childCnt[i] = 0;
        __syncthreads();
        // keys move to their nodes of the next list and are counted into the quadrants of those nodes;
        // four keys per trip: the loads of a trip are issued together, so a thread waits for memory once
        // per four keys
        const short4* nb = xb;
        const u32* ncnt = xcnt;
        if constexpr (REG) {
#pragma unroll
            for (int j = 0; j < QT_KPT; j++) {
                if (tid + j * QT_THREADS < nkeys) {
                    const int nd = (int)rnode[j];
                    const u16 r = rankOf[nd];
                    int to;
                    if (r != 0xFFFF && r < nproc)
                        to = newId[nd * 4 + child_quadrant(rkey[j], cb[nd])];
                    else
                        to = moved[nd];
                    rnode[j] = (u32)to;
                    if (to < nextL && ncnt[to] > 1) atomicAdd(&childCnt[to * 4 + child_quadrant(rkey[j], nb[to])], 1u);
                }
            }
        } else
        for (int k0 = tid; k0 < nkeys; k0 += 4 * QT_THREADS) {
            int nd[4];
            u32 key[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = min(k0 + j * QT_THREADS, nkeys - 1);
                nd[j] = kn[k];
                key[j] = kp[k];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = k0 + j * QT_THREADS;
                if (k < nkeys) {
                    const u16 r = rankOf[nd[j]];
                    int to;
                    if (r != 0xFFFF && r < nproc)
                        to = newId[nd[j] * 4 + child_quadrant(key[j], cb[nd[j]])];
                    else
                        to = moved[nd[j]];
                    kn[k] = (u16)to;
                    if (to < nextL && ncnt[to] > 1) atomicAdd(&childCnt[to * 4 + child_quadrant(key[j], nb[to])], 1u);
                }
            }
        }
        nextSeq += (u32)tnew;
        __syncthreads();
        QT_MARK("sweep");
        if (tid == 0) {
            if (newL > lcap) {
                *errFlag = 1;
                S.done = 1;
            }
            S.L = min(newL, lcap);
            if (newL >= N || newL == L)
                S.done = 1;
            else if (!careful && newL + 3 * expand > N)
                S.careful = 1;
        }
        cur = nxt;
        __syncthreads();
        if (S.done) break;
    }

    // ---- best key per node: max response, first in candidate order on ties
    const int L = S.L;
    u64* best = (u64*)childCnt;
    for (int i = tid; i < L; i += QT_THREADS) best[i] = 0;
    __syncthreads();
    if constexpr (REG) {
#pragma unroll
        for (int j = 0; j < QT_KPT; j++) {
            const int k = tid + j * QT_THREADS;
            if (k < nkeys) atomicMax(&best[rnode[j]], ((u64)ORBX_PR(rkey[j]) << 32) | (u64)(0xFFFFFFFFu - (u32)k));
        }
    } else
    for (int k0 = tid; k0 < nkeys; k0 += 4 * QT_THREADS) {
        int nd[4];
        u32 key[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = min(k0 + j * QT_THREADS, nkeys - 1);
            nd[j] = kn[k];
            key[j] = kp[k];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = k0 + j * QT_THREADS;
            if (k < nkeys) atomicMax(&best[nd[j]], ((u64)ORBX_PR(key[j]) << 32) | (u64)(0xFFFFFFFFu - (u32)k));
        }
    }
    __syncthreads();
    for (int i = tid; i < L; i += QT_THREADS) {
        const u32 k = 0xFFFFFFFFu - (u32)(best[i] & 0xFFFFFFFFu);
        const u32 key = kp[k];
        selOut[i] = ORBX_PACK(ORBX_PX(key) + MIN_BORDER, ORBX_PY(key) + MIN_BORDER, ORBX_PR(key));
    }
    QT_MARK("best");
    if (tid == 0) *selCnt = L;
    };
    if (nkeys <= QT_THREADS * QT_KPT)  // the same for every thread
        subdivide(std::true_type{});
    else
        subdivide(std::false_type{});
}

// ------------------------------------------------------------------ describe
// Tables staged into LDS by every describe block (uploaded once at start-up):
//   patch: 248 slots = 31 rows x 8 four-pixel words covering u in [-16, 15];
//          per slot the byte weights (u + 15, or 0 outside the radius-15 disc)
//          and the 0/1 membership bytes,
//   pattern: the 256 rBRIEF tests, (x0, y0, x1, y1) as four signed bytes.
struct DescribeTables {
    u32 patchU[248];
    u32 patchM[248];
    u32 pattern[256];
};
__device__ DescribeTables d_tables;

// cv::fastAtan2 restated: degrees in [0, 360), one rounding per operation.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float eps = 2.2204460492503131e-16f;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// sinf/cosf of the steering angle, bit-identical to glibc 2.28+ for |x| < 120:
// the same algorithm (double-precision range reduction by pi/2 and degree-7/8
// minimax polynomials, one final rounding to float) with the same constants.
// An exhaustive host check over every float in [0, 6.5] shows this evaluation
// equals libm's result whether or not the products are fused.
__device__ __forceinline__ float sincosf_poly(double x, double x2, double sgn, int n) {
    // sgn = -1 selects the negated cosine table of quadrants 2 and 3
    if ((n & 1) == 0) {
        const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
        const double x3 = x * x2;
        const double t1 = s2 + x2 * s3;
        const double x7 = x3 * x2;
        const double s = x + x3 * s1;
        return (float)(s + x7 * t1);
    } else {
        const double c0 = 0x1p0 * sgn, c1 = -0x1.ffffffd0c621cp-2 * sgn, c2 = 0x1.55553e1068f19p-5 * sgn,
                     c3 = -0x1.6c087e89a359dp-10 * sgn, c4 = 0x1.99343027bf8c3p-16 * sgn;
        const double x4 = x2 * x2;
        const double t2 = c3 + x2 * c4;
        const double t1 = c1 + x2 * c2;
        const double x6 = x4 * x2;
        const double c = c0 + x2 * t1;
        return (float)(c + x6 * t2);
    }
}

__device__ __forceinline__ void steer_sincos(float y, float* sn, float* cs) {
    const u32 top = (__float_as_uint(y) >> 20) & 0x7ff;
    const double x = (double)y;
    if (top < ((__float_as_uint(0x1.921FB6p-1f) >> 20) & 0x7ff)) {
        const double x2 = x * x;
        if (top < ((__float_as_uint(0x1p-12f) >> 20) & 0x7ff)) {
            *sn = y;
            *cs = 1.0f;
            return;
        }
        *sn = sincosf_poly(x, x2, 1.0, 0);
        *cs = sincosf_poly(x, x2, 1.0, 1);
        return;
    }
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int)r + 0x800000) >> 24;
    const double xr = x - (double)n * 0x1.921FB54442D18p0;
    const double x2 = xr * xr;
    // sine: sign[n & 3], table 1 when n & 2
    {
        const double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        *sn = sincosf_poly(xr * sg, x2, (n & 2) ? -1.0 : 1.0, n);
    }
    // cosine: the same with n + 1 for the sign/table and n ^ 1 for the polynomial
    {
        const int m = n + 1;
        const double sg = ((m & 3) == 1 || (m & 3) == 2) ? -1.0 : 1.0;
        *cs = sincosf_poly(xr * sg, x2, (m & 2) ? -1.0 : 1.0, n ^ 1);
    }
}

__global__ void k_debug_sincos(const float* in, float* sn, float* cs, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) steer_sincos(in[i], &sn[i], &cs[i]);
}

__global__ void k_debug_atan2(const float* y, const float* x, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fast_atan2_deg(y[i], x[i]);
}

#ifndef DESC_KPW
#define DESC_KPW 8  // keypoints per wave (8 or 4)
#endif
#define DESC_LPV (32 / DESC_KPW)  // lanes that end up holding one moment of one keypoint after the exchange
#define DESC_R 18            // window radius of the steered tests
#define DESC_ROW_DW 10       // 32-bit words per window row (37 bytes + up to 3 of alignment)
#define DESC_WIN_DW 384      // window slots in LDS: 37 * 10 = 370 used, six loads of 64 lanes
#ifndef DESC_BATCH
#define DESC_BATCH DESC_KPW  // windows staged at a time per wave
#endif

__device__ __forceinline__ int readlane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Orientation + descriptor, DESC_KPW keypoints per wave.
//   A. moments: the radius-15 disc of every keypoint is read as 248 unaligned 32-bit words
//      (4 per lane, all keypoints' loads in flight together) and reduced with two
//      v_dot4_u32_u8 per word (weights u+15 and membership);
//   B. the 16 partial sums of a lane (m10, m01 of 8 keypoints) are summed over the wave by a
//      halving exchange (17 cross-lane moves instead of 96) that leaves keypoint j's sums in
//      lanes 8j..8j+7, where fastAtan2 and the steering sine/cosine are evaluated once for
//      all keypoints;
//   C. per keypoint: 256 steered tests on the blurred level, one bit per lane and round,
//      gathered with wave ballots.  The rotated test points stay within 18 pixels of the keypoint
//      (the pattern's largest radius is 18.38), so the 37 x 40-byte window around each keypoint is
//      copied to LDS with six row-coalesced direct-to-LDS loads and the 512 byte gathers of a
//      keypoint hit LDS: gathering from global memory costs a wave up to 64 cache-line
//      lookups per load instruction, which bounded this kernel.
// FMA = true: the steering sums x*b + y*a and x*a - y*b with the first product fused into the sum
// (what GCC emits for the reference when it is built for a host with fused multiply-add,
// perfect/CMakeLists.txt:20); false: every product rounded, as written (src/ORBextractor.cc:103-104).
template <bool FMA>
__global__ __launch_bounds__(256) void k_describe(Geom g, ImgRef img0, const u8* __restrict__ pyr,
                                                   const u8* __restrict__ blur,
                                                   const u32* __restrict__ sel,
                                                   const int* __restrict__ selCount,
                                                   orbx_keypoint* kps, u8* desc, int cap, int* nOut,
                                                   int* errFlag, int nFrames, int nChunks, u32 chunkMagic) {
    // Workgroups go to the 8 XCDs round-robin by linear id.  XCD x takes frames x, x + 8, ... and
    // walks through all keypoint chunks of one frame before the next, so the patches and windows of
    // a frame, which overlap heavily, are fetched into that XCD's L2 once instead of once per chunk.
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int fq = (int)fastdiv((u32)seq, (u32)nChunks, chunkMagic);
    const int f = fq * 8 + xcd;
    const int chunk = seq - fq * nChunks;
    if (f >= nFrames) return;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int* sc = selCount + (i64)f * ORBX_MAX_LEVELS;
    int cum[ORBX_MAX_LEVELS + 1];
    cum[0] = 0;
#pragma unroll
    for (int k = 0; k < ORBX_MAX_LEVELS; k++) cum[k + 1] = cum[k] + (k < g.nlevels ? sc[k] : 0);
    const int total = min(cum[ORBX_MAX_LEVELS], cap);
    const int sBase = (chunk * 4 + wv) * DESC_KPW;
    if (sBase == 0 && lane == 0) {
        nOut[f] = total;
        if (cum[ORBX_MAX_LEVELS] > cap) *errFlag = 2;
    }
    if (sBase >= total) return;

    // tables: this lane's 4 disc words and its 4 tests (one per round), kept in registers
    u32 wU[4], wM[4], wV[4];
    float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int slot = min(lane + 64 * i, 247);
        wU[i] = lane + 64 * i < 248 ? d_tables.patchU[slot] : 0u;
        wM[i] = lane + 64 * i < 248 ? d_tables.patchM[slot] : 0u;
        wV[i] = wM[i] * (u32)(slot >> 3);  // membership bytes (0 / 1) times the word's row (v + 15): no carries
        const int pw = (int)d_tables.pattern[i * 64 + lane];
        px0[i] = (float)((pw << 24) >> 24);
        py0[i] = (float)((pw << 16) >> 24);
        px1[i] = (float)((pw << 8) >> 24);
        py1[i] = (float)(pw >> 24);
    }

    // lane j < DESC_KPW looks up keypoint sBase + j (slots past the end repeat the first one)
    int myL = 0, myX = 0, myY = 0, myR = 0;
    {
        const int s = (lane < DESC_KPW && sBase + lane < total) ? sBase + lane : sBase;
        int l = 0;
#pragma unroll
        for (int k = 1; k < ORBX_MAX_LEVELS; k++) l += (s >= cum[k]) ? 1 : 0;
        l = min(l, g.nlevels - 1);
        int c = 0, selBase = 0;
#pragma unroll
        for (int k = 0; k < ORBX_MAX_LEVELS; k++) {
            if (k < g.nlevels && l == k) {
                c = cum[k];
                selBase = g.lv[k].selBase;
            }
        }
        const u32 key = sel[(i64)f * g.totalSel + selBase + (s - c)];
        myL = l;
        myX = ORBX_PX(key);
        myY = ORBX_PY(key);
        myR = ORBX_PR(key);
    }

    __shared__ u32 window[4][DESC_BATCH][DESC_WIN_DW];
    u32 winOff[6];  // this lane's six window words: byte offset from the window's corner = row * pitch + 4 * word
    u32 winRow[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int i = min(k * 64 + lane, DESC_ROW_DW * (2 * DESC_R + 1) - 1);
        const int row = (int)__umulhi((u32)i, div_magic(DESC_ROW_DW));
        winRow[k] = (u32)row;
        winOff[k] = (u32)(4 * (i - row * DESC_ROW_DW));
    }
    auto issue_windows = [&](int j0) {
#pragma unroll
        for (int jj = 0; jj < DESC_BATCH; jj++) {
            const int j = j0 + jj;
            const int l = readlane_i(myL, j), x = readlane_i(myX, j), y = readlane_i(myY, j);
            const u32 blurPitch = (u32)g.lv[l].blurPitch;
            // top-left corner of the window: a wave-uniform base plus a non-negative 32-bit lane offset
            const u8* wb = blur + (i64)f * g.blurFrameBytes + g.lv[l].blurOff + (i64)(y - DESC_R) * blurPitch + ((x - DESC_R) & ~3);
#pragma unroll
            for (int k = 0; k < 6; k++)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(wb + (size_t)mad24(winRow[k], blurPitch, winOff[k])),
                    (__attribute__((address_space(3))) void*)&window[wv][jj][k * 64], 4, 0, 0);
        }
    };
    // A: moments
    int V[2 * DESC_KPW];
    {
        u32 px[DESC_KPW][4];
#pragma unroll
        for (int j = 0; j < DESC_KPW; j++) {
            const int l = readlane_i(myL, j), x = readlane_i(myX, j), y = readlane_i(myY, j);
            const ImgRef im = level_image(g, l, img0, pyr);
            // corner of the 31 x 32 patch: a wave-uniform base plus a non-negative 32-bit lane offset
            const u8* corner = im.base + (i64)f * im.frameStride + (i64)(y - 15) * im.pitch + (x - 16);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const u32 slot = (u32)min(lane + 64 * i, 247);
                __builtin_memcpy(&px[j][i], corner + (size_t)mad24(slot >> 3, (u32)im.pitch, (slot & 7) * 4), 4);
            }
        }
        // The first windows travel while the moments are computed.  (Scheduling fences: the first use of
        // a patch word makes the compiler wait for every load issued so far, so no such use may drift
        // in between the window loads.)
        __builtin_amdgcn_sched_barrier(0);
        issue_windows(0);
        __builtin_amdgcn_sched_barrier(0);
        // Per lane: D = sum px * (u + 15), E = sum px * (v + 15), S = sum px over the disc bytes of its four
        // words, each a chain of four dot products (the accumulator input does the additions); then
        // m10 = D - 15 S and m01 = E - 15 S.  All three sums stay below 2^17, so the product is a
        // 24-bit multiply.
#pragma unroll
        for (int j = 0; j < DESC_KPW; j++) {
            u32 D = 0, E = 0, S = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                D = __builtin_amdgcn_udot4(px[j][i], wU[i], D, false);
                E = __builtin_amdgcn_udot4(px[j][i], wV[i], E, false);
                S = __builtin_amdgcn_udot4(px[j][i], wM[i], S, false);
            }
            const int S15 = (int)mul24(S, 15u);
            V[2 * j] = (int)D - S15;
            V[2 * j + 1] = (int)E - S15;
        }
    }
    // B: halving exchange.  After the steps with lane distance 32, 16, 8, 4 a lane holds the
    // sum over 4 lanes... of value index (lane >> 2) & 15; distances 2 and 1 finish the sum.
    {
#pragma unroll
        for (int half = DESC_KPW, dist = 32; half >= 1; half >>= 1, dist >>= 1) {
            const bool up = (lane & dist) != 0;
#pragma unroll
            for (int t = 0; t < half; t++) {
                const int keep = up ? V[t + half] : V[t];
                const int send = up ? V[t] : V[t + half];
                V[t] = keep + __shfl_xor(send, dist);
            }
        }
#pragma unroll
        for (int d = DESC_LPV / 2; d >= 1; d >>= 1) V[0] += __shfl_xor(V[0], d);
    }
    // with L = DESC_LPV: lanes 2Lj .. 2Lj+L-1 hold m10 of keypoint j, lanes 2Lj+L .. 2Lj+2L-1 its m01
    const int other = __shfl_xor(V[0], DESC_LPV);
    const int m10 = (lane & DESC_LPV) ? other : V[0], m01 = (lane & DESC_LPV) ? V[0] : other;
    const float factorPI = (float)(M_PI / 180.f);
    const float angleV = fast_atan2_deg((float)m01, (float)m10);
    float aV, bV;
    steer_sincos(__fmul_rn(angleV, factorPI), &bV, &aV);

    // C: descriptors and keypoint records
#pragma unroll 1
    for (int j0 = 0; j0 < DESC_KPW; j0 += DESC_BATCH) {
        if (sBase + j0 >= total) break;
        if (j0 > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous batch has been read
            issue_windows(j0);
        }
        // direct-to-LDS loads are not tracked as LDS writes by the compiler: wait for them here
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 2
        for (int jj = 0; jj < DESC_BATCH; jj++) {
            const int j = j0 + jj;
            const int s = sBase + j;
            if (s >= total) break;
            const int l = readlane_i(myL, j), x = readlane_i(myX, j), y = readlane_i(myY, j), resp = readlane_i(myR, j);
            const float a = readlane_f(aV, 2 * DESC_LPV * j), b = readlane_f(bV, 2 * DESC_LPV * j), angle = readlane_f(angleV, 2 * DESC_LPV * j);
            // Byte address of the keypoint in its LDS window.  Rounding to nearest-even is done by adding
            // 1.5 * 2^23: the sum's bit pattern is 0x4B400000 + round(v) for |v| < 2^22, and a 24-bit
            // multiply-add on the patterns gives 40 * row + column plus a constant folded into the base.
            const u32 MAGIC = 0x4B400000u;
            const u32 bc = (u32)__builtin_amdgcn_readfirstlane(
                (int)((u32)(size_t)(const __attribute__((address_space(3))) u8*)&window[wv][jj][0] +
                      DESC_R * (4 * DESC_ROW_DW) + DESC_R + ((x - DESC_R) & 3) -
                      ((MAGIC & 0xFFFFFFu) * (4 * DESC_ROW_DW) + MAGIC)));
            u32 t0[4], t1[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float fr0, fc0, fr1, fc1;
                if (FMA) {
                    fr0 = __fadd_rn(__fmaf_rn(px0[r], b, __fmul_rn(py0[r], a)), 12582912.f);
                    fc0 = __fadd_rn(__fmaf_rn(px0[r], a, -__fmul_rn(py0[r], b)), 12582912.f);
                    fr1 = __fadd_rn(__fmaf_rn(px1[r], b, __fmul_rn(py1[r], a)), 12582912.f);
                    fc1 = __fadd_rn(__fmaf_rn(px1[r], a, -__fmul_rn(py1[r], b)), 12582912.f);
                } else {
                    fr0 = __fadd_rn(__fadd_rn(__fmul_rn(px0[r], b), __fmul_rn(py0[r], a)), 12582912.f);
                    fc0 = __fadd_rn(__fsub_rn(__fmul_rn(px0[r], a), __fmul_rn(py0[r], b)), 12582912.f);
                    fr1 = __fadd_rn(__fadd_rn(__fmul_rn(px1[r], b), __fmul_rn(py1[r], a)), 12582912.f);
                    fc1 = __fadd_rn(__fsub_rn(__fmul_rn(px1[r], a), __fmul_rn(py1[r], b)), 12582912.f);
                }
                const u32 a0 = mad24(__float_as_uint(fr0), 4 * DESC_ROW_DW, __float_as_uint(fc0)) + bc;
                const u32 a1 = mad24(__float_as_uint(fr1), 4 * DESC_ROW_DW, __float_as_uint(fc1)) + bc;
                t0[r] = *(const __attribute__((address_space(3))) u8*)(size_t)a0;
                t1[r] = *(const __attribute__((address_space(3))) u8*)(size_t)a1;
            }
            u32 word = 0;  // lane k < 8 ends up holding descriptor bytes 4k..4k+3
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const u64 bits = __ballot(t0[r] < t1[r]);
                // (a VALU compare result read by v_writelane right away returns stale data, so the
                // selection stays on conditional moves)
                word = lane == 2 * r ? (u32)bits : word;
                word = lane == 2 * r + 1 ? (u32)(bits >> 32) : word;
            }
            const i64 o = (i64)f * cap + s;
            if (lane < 8) ((u32*)(desc + o * 32))[lane] = word;
            // the 7 dwords of the 28-byte keypoint, one per lane
            float kx = (float)x, ky = (float)y;
            const float scale = g.lv[l].scale, kpSize = g.lv[l].kpSize;
            if (l != 0) {
                kx = __fmul_rn(kx, scale);
                ky = __fmul_rn(ky, scale);
            }
            u32 kv = __float_as_uint(kx);
            kv = lane == 1 ? __float_as_uint(ky) : kv;
            kv = lane == 2 ? __float_as_uint(kpSize) : kv;
            kv = lane == 3 ? __float_as_uint(angle) : kv;
            kv = lane == 4 ? __float_as_uint((float)resp) : kv;
            kv = lane == 5 ? (u32)l : kv;
            kv = lane == 6 ? 0xFFFFFFFFu : kv;
            if (lane < 7) ((u32*)(kps + o))[lane] = kv;
        }
    }
}

// ------------------------------------------------------------------ host
int cv_round_host(double v) { return (int)lrint(v); }

struct StageProfile {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[ORBX_STAGE_COUNT];
    size_t used[ORBX_STAGE_COUNT] = {0, 0, 0, 0, 0};
    double ms[ORBX_STAGE_COUNT] = {0, 0, 0, 0, 0};
    int launches[ORBX_STAGE_COUNT] = {0, 0, 0, 0, 0};
};

}  // namespace

// Worker threads for the host-side copies of the host-memory path (caller's pageable images -> pinned
// staging, pinned results -> caller's arrays): one thread moves about 10 GB/s, a pass of 640x480 frames
// wants several times that.  run(n, fn) calls fn(0) .. fn(n - 1) on the workers and the calling thread.
struct CopyPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cvDone;
    std::function<void(int)> job;
    int nTasks = 0, next = 0, pending = 0;
    unsigned long long gen = 0;
    bool stop = false;
    explicit CopyPool(int workers) {
        for (int i = 0; i < workers; i++) th.emplace_back([this] { loop(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> l(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    void loop() {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> l(mu);
        while (true) {
            cv.wait(l, [&] { return stop || (gen != seen && next < nTasks); });
            if (stop) return;
            while (next < nTasks) {
                const int i = next++;
                l.unlock();
                job(i);
                l.lock();
                if (--pending == 0) cvDone.notify_all();
            }
            seen = gen;
        }
    }
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (th.empty() || n == 1) {
            for (int i = 0; i < n; i++) fn(i);
            return;
        }
        std::unique_lock<std::mutex> l(mu);
        job = fn;
        nTasks = n;
        next = 0;
        pending = n;
        gen++;
        cv.notify_all();
        while (next < nTasks) {  // the caller works too
            const int i = next++;
            l.unlock();
            fn(i);
            l.lock();
            --pending;
        }
        cvDone.wait(l, [&] { return pending == 0; });
    }
};

struct orbx_handle {
    orbx_config cfg;
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    // the blur of a pass only depends on the pyramid: it runs on a second stream beside
    // FAST + quadtree and joins before the descriptor kernel
    hipStream_t auxStream = nullptr;
    hipEvent_t evPyramid = nullptr, evBlur = nullptr;
    bool overlapBlur = true;
    // constructor tables
    float scale[ORBX_MAX_LEVELS], invScale[ORBX_MAX_LEVELS], sigma2[ORBX_MAX_LEVELS],
        invSigma2[ORBX_MAX_LEVELS];
    int32_t quota[ORBX_MAX_LEVELS];
    int gaussK[4];
    // geometry of the current image size
    Geom g;
    bool geomValid = false;
    size_t qtLds = 0;        // bytes of node tables per (frame, level)
    bool qtGlobal = false;   // tables too large for the LDS: they live in dNodeTables
    int qtThreads = 256;
    ResizeTap* dTaps = nullptr;  // x taps then y taps for each level >= 1
    FastCell* dCells = nullptr;  // one record per FAST cell, all levels, ordered group by group
    std::vector<FastGroup> fastGroups;   // one launch of k_fast each
    FastGroup fastWhole;                 // the same cells as one launch with the buffers of the largest cell
    std::vector<int> fastLevelGroup;     // level -> group
    bool fastOneGroup = false;           // ORBX_FAST_GROUPS=1: one launch with the buffers of the largest cell (measurement / test knob)
    bool fastAlwaysSplit = false;        // ORBX_FAST_GROUPS=2: the level groups also for passes too small to fill the GPU (test knob)
    size_t cellsCap = 0;
    size_t tapsCap = 0;
    size_t tapXOff[ORBX_MAX_LEVELS], tapYOff[ORBX_MAX_LEVELS], tapSOff[ORBX_MAX_LEVELS];
    bool resizeStreamOk[ORBX_MAX_LEVELS];  // taps of a 4-pixel group fit the three-word window
    PyramidArgs pyrArgs;                   // per-level arguments of the one-launch pyramid
    bool pyrFrameOk = false;               // every level fits the one-launch pyramid's assumptions
    int pyrFused = -1;                     // ORBX_PYR_FUSED: 1 / 0 forces the one-launch pyramid on / off (-1: by batch size)
    int blurLogW = 0;                      // ORBX_BLUR_LOGW: forces the lane split of the wave-uniform blur (measurement knob)
    bool blurRows = true;                  // ORBX_BLUR_ROWS=0: per-lane form of the streaming blur (measurement knob)
    int maxPipes = 3;                      // ORBX_NPIPES: passes in flight (measurement knob)
    int forkBlur = 1;                      // ORBX_FORK_BLUR: 0 the blur stays on the pass's own stream, 1 forked after the pyramid, 2 forked after FAST
    bool noRepack = false;                 // ORBX_NO_REPACK=1: device-resident frames with an unaligned layout are read in place (test knob)
    bool fastGenericPitch = false;         // ORBX_FAST_GENERIC=1: the FAST kernel with the ROI pitch as a run-time value (test knob)
    int blurLds = 0;                       // ORBX_BLUR_LDS: unused LDS bytes per blur workgroup, caps its share of a compute unit (measurement knob)
    int resizeLogW = 0;                    // ORBX_RESIZE_LOGW: forces the lane split of the wave-uniform form (measurement knob)
    bool resizeRows = true;                // wave-uniform form of the streaming resize (ORBX_RESIZE_ROWS=0: per-lane form)
    // one walk per level (k_level_walk: blurred level + next level from one pass over the rows)
    struct WalkLevel {
        size_t fdOff = 0, chOff[2] = {0, 0};  // byte offsets into dWalk: first-destination table, chunk tables for W = 64 / 32
        int nCh[2] = {0, 0};
    } walk[ORBX_MAX_LEVELS];
    bool walkOk = false;                   // every level meets the walk's assumptions
    int walkMode = 0;                      // ORBX_WALK=1: the walk where it applies (measured slower than the separate pyramid and blur kernels: both forms are bound by vector-instruction issue, not by bytes, and the walk issues more)
    unsigned char* dWalk = nullptr;
    size_t walkCap = 0;
    // scratch sized for scratchBatch frames of the current geometry
    int scratchBatch = 0;
    size_t scratchSig = 0;
    // d* are the blocks of the pass being enqueued (a view into the owned base* allocations:
    // with several passes in flight each one works in its own slice)
    u8 *dPyr = nullptr, *dBlur = nullptr;
    u32 *dCellCount = nullptr, *dSlots = nullptr, *dKeyPacked = nullptr, *dSel = nullptr;
    u16* dKeyNode = nullptr;
    unsigned char *dNodeTables = nullptr, *baseNodeTables = nullptr;  // only with qtGlobal
    int* dSelCount = nullptr;
    // Error word of the kernels (1: quadtree node table full, 2: keypoint capacity per frame too small).
    // It lives in pinned host memory that the device writes directly, so that reading it after a
    // stream wait costs no copy; dErr is the device view of hErr.
    int *hErr = nullptr, *dErr = nullptr;
    u8 *basePyr = nullptr, *baseBlur = nullptr;
    u32 *baseCellCount = nullptr, *baseSlots = nullptr, *baseKeyPacked = nullptr, *baseSel = nullptr;
    u16* baseKeyNode = nullptr;
    int* baseSelCount = nullptr;
    // Device-memory batches longer than one pass are spread over ORBX_PIPES passes in flight,
    // each on its own pair of streams (pipe 0 = the handle's own), so that the tail of one pass
    // and its latency-bound kernels run beside the next pass.
    struct Pipe {
        hipStream_t st = nullptr, aux = nullptr;
        hipEvent_t evPyramid = nullptr, evBlur = nullptr, evDone = nullptr;
    } pipes[ORBX_PIPES];
    hipEvent_t evStart = nullptr;
    // staging for host-memory calls: a ring of slots, each a pass of frames in pinned host memory and in
    // device memory plus that pass's outputs in both, so that the host copies of pass k + 1, the upload of
    // pass k, the kernels of pass k - 1 and the download of pass k - 2 overlap
    struct HostSlot {
        u8 *pinIn = nullptr, *dIn = nullptr;
        size_t inCap = 0;
        u8 *pinOut = nullptr, *dOut = nullptr;  // keypoints, descriptors, counts of one pass
        size_t outCap = 0;
        hipEvent_t evIn = nullptr, evComp = nullptr, evOut = nullptr;
        int f0 = 0, n = 0;
        bool busy = false;
    } hostSlots[3];
    hipStream_t copyIn = nullptr, copyOut = nullptr;
    struct CopyPool* pool = nullptr;
    u8* dAlignStage = nullptr;  // aligned copies of device-resident frames whose layout is not word-aligned (one slice per pass in flight)
    size_t alignStageCap = 0;
    // last call (for orbx_get_level and the debug taps)
    ImgRef lastImg0 = {nullptr, 0, 0};
    int lastBatch = 0, lastFirstFrame = 0;
    bool profiling = false;
    StageProfile prof;
};

namespace {

void free_dev(void* p) {
    if (p) (void)hipFree(p);
}

void make_tables(orbx_handle* h) {
    // reference src/ORBextractor.cc:404-440; scaleFactor is stored as a double
    const int n = h->cfg.nlevels;
    const double sf = (double)h->cfg.scale_factor;
    h->scale[0] = 1.0f;
    h->sigma2[0] = 1.0f;
    for (int i = 1; i < n; i++) {
        h->scale[i] = (float)(h->scale[i - 1] * sf);
        h->sigma2[i] = h->scale[i] * h->scale[i];
    }
    for (int i = 0; i < n; i++) {
        h->invScale[i] = 1.0f / h->scale[i];
        h->invSigma2[i] = 1.0f / h->sigma2[i];
    }
    float factor = (float)(1.0f / sf);
    float desired = h->cfg.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)n));
    int sum = 0;
    for (int level = 0; level < n - 1; level++) {
        h->quota[level] = cv_round_host(desired);
        sum += h->quota[level];
        desired *= factor;
    }
    h->quota[n - 1] = std::max(h->cfg.nfeatures - sum, 0);
    if (h->cfg.blur_variant == 1) {
        h->gaussK[0] = 18; h->gaussK[1] = 34; h->gaussK[2] = 48; h->gaussK[3] = 56;
    } else {
        h->gaussK[0] = 18; h->gaussK[1] = 34; h->gaussK[2] = 49; h->gaussK[3] = 55;
    }
}

const int QT_MAX_LDS = 160 * 1024 - 1024;  // dynamic LDS the quadtree kernels may ask for

int upload_constants_locked();

// Process-wide set-up of one device, done once under a lock however many handles and threads there
// are: the descriptor tables go to that device's copy of d_tables and the quadtree kernels get
// their dynamic-LDS ceiling (function attributes are per device and shared by all handles).
int device_init_once(int device) {
    static std::mutex mu;
    static bool done[64] = {false};
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0 || device >= 64) {
        orbx_set_error("device ordinal %d out of range", device);
        return ORBX_ERR_INVALID;
    }
    if (done[device]) return ORBX_OK;
    int rc = upload_constants_locked();
    if (rc) return rc;
    ORBX_HIP_CHECK(hipFuncSetAttribute((const void*)k_quadtree<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, QT_MAX_LDS));
    ORBX_HIP_CHECK(hipFuncSetAttribute((const void*)k_quadtree<512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, QT_MAX_LDS));
    ORBX_HIP_CHECK(hipFuncSetAttribute((const void*)k_quadtree<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, QT_MAX_LDS));
    done[device] = true;
    return ORBX_OK;
}

int upload_constants_locked() {
    // circular patch of radius 15 (reference src/ORBextractor.cc:450-465: umax)
    static const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    DescribeTables t;
    int inside = 0;
    for (int slot = 0; slot < 248; slot++) {
        const int v = (slot >> 3) - 15, u0 = (slot & 7) * 4 - 16;
        u32 wu = 0, wm = 0;
        for (int j = 0; j < 4; j++) {
            const int u = u0 + j;
            const int d = umax[v < 0 ? -v : v];
            if (u >= -d && u <= d) {
                wu |= (u32)(u + 15) << (8 * j);
                wm |= 1u << (8 * j);
                inside++;
            }
        }
        t.patchU[slot] = wu;
        t.patchM[slot] = wm;
    }
    if (inside != 749) {
        orbx_set_error("internal: patch has %d pixels", inside);
        return ORBX_ERR_INVALID;
    }
    for (int i = 0; i < 256; i++) {
        u32 w = 0;
        for (int j = 0; j < 4; j++) w |= (u32)(u8)ORBX_PATTERN_31[4 * i + j] << (8 * j);
        t.pattern[i] = w;
    }
    ORBX_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_tables), &t, sizeof(t)));
    return ORBX_OK;
}

int align_up(int v, int a) { return (v + a - 1) / a * a; }

// Level sizes, cell grids, quadtree parameters and scratch offsets for a w x h image.
int compute_geometry(orbx_handle* h, int w, int hh) {
    Geom& g = h->g;
    if (h->geomValid && g.w == w && g.h == hh) return ORBX_OK;
    memset(&g, 0, sizeof(g));
    g.nlevels = h->cfg.nlevels;
    g.w = w;
    g.h = hh;
    i64 pyrOff = 0, blurOff = 0;
    int cellBase = 0, slotBase = 0, selBase = 0, tileBase = 0, itemBase = 0;
    size_t maxLds = 0;
    for (int l = 0; l < g.nlevels; l++) {
        LevelGeom& L = g.lv[l];
        // reference src/ORBextractor.cc:1121-1122
        L.w = cv_round_host((float)w * h->invScale[l]);
        L.h = cv_round_host((float)hh * h->invScale[l]);
        if (L.w < 1 || L.h < 1 || L.w > 4096 || L.h > 4096) {
            orbx_set_error("level %d size %dx%d outside [1,4096]", l, L.w, L.h);
            return ORBX_ERR_UNSUPPORTED;
        }
        L.pyrPitch = align_up(L.w, 64);
        L.blurPitch = align_up(L.w, 64);
        if (l > 0) {
            // a level starts on a cache line of its own: the one-launch pyramid reads a level while the next
            // one is being written, and a line holding bytes of both could be kept stale in the vector cache
            pyrOff = (pyrOff + 255) & ~(i64)255;
            L.pyrOff = pyrOff;
            pyrOff += (i64)L.pyrPitch * L.h;
        }
        L.blurOff = blurOff;
        blurOff += (i64)L.blurPitch * L.h;
        L.blurTilesX = (L.w + BLUR_TW - 1) / BLUR_TW;
        L.blurTiles = L.blurTilesX * ((L.h + BLUR_TH - 1) / BLUR_TH);
        L.blurTileBase = tileBase;
        tileBase += L.blurTiles;
        L.blurGroups = (L.w + 3) / 4;
        L.blurBands = (L.h + BLUR_BH - 1) / BLUR_BH;
        L.blurItemBase = itemBase;
        itemBase += align_up(L.blurGroups * L.blurBands, 64);
        // reference src/ORBextractor.cc:781-797
        const float width = (float)((L.w - MIN_BORDER) - MIN_BORDER);
        const float height = (float)((L.h - MIN_BORDER) - MIN_BORDER);
        L.nCols = (int)(width / 30.f);
        L.nRows = (int)(height / 30.f);
        if (L.nCols <= 0 || L.nRows <= 0) {
            L.nCols = L.nRows = 0;
            L.wCell = L.hCell = 1;
        } else {
            L.wCell = (int)ceilf(width / L.nCols);
            L.hCell = (int)ceilf(height / L.nRows);
        }
        if (L.wCell > MAX_CELL || L.hCell > MAX_CELL) {
            orbx_set_error("internal: cell %dx%d too large", L.wCell, L.hCell);
            return ORBX_ERR_UNSUPPORTED;
        }
        L.nCells = L.nCols * L.nRows;
        L.cellBase = cellBase;
        cellBase += L.nCells;
        // no two survivors of 8-neighbour non-max suppression are adjacent
        L.cellCap = ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);
        L.slotBase = slotBase;
        L.keyBase = slotBase;
        slotBase += L.nCells * L.cellCap;
        // reference src/ORBextractor.cc:545-547
        L.quota = h->quota[l];
        L.nIni = 0;
        L.hX = 1.f;
        if (L.nCells > 0) {
            const int dx = L.w - 2 * MIN_BORDER, dy = L.h - 2 * MIN_BORDER;
            L.nIni = (int)roundf((float)dx / (float)dy);
            if (L.nIni < 1 || L.nIni > MAX_ROOTS) {
                orbx_set_error("level %d aspect ratio %d:%d unsupported (initial nodes %d)", l, dx, dy, L.nIni);
                return ORBX_ERR_UNSUPPORTED;
            }
            L.hX = (float)dx / (float)L.nIni;
        }
        L.lcap = align_up(std::max(std::max(L.quota, 4 * L.nIni), MAX_ROOTS) + 8, 8);
        if (L.lcap > 60000) {
            orbx_set_error("nfeatures too large (%d at level %d)", L.quota, l);
            return ORBX_ERR_UNSUPPORTED;
        }
        L.selBase = selBase;
        selBase += L.lcap;
        maxLds = std::max(maxLds, (size_t)L.lcap * 72 + 128);
        L.scale = h->scale[l];
        L.kpSize = (float)(int)(31 * h->scale[l]);  // reference :847, :857
    }
    g.totalCells = std::max(cellBase, 1);
    g.totalSlots = std::max(slotBase, 1);
    g.totalSel = selBase;
    g.totalBlurTiles = tileBase;
    g.totalBlurItems = itemBase;
    {
        int mw = 1, mh = 1;
        for (int l = 0; l < g.nlevels; l++)
            if (g.lv[l].nCells > 0) {
                mw = std::max(mw, g.lv[l].wCell);
                mh = std::max(mh, g.lv[l].hCell);
            }
        // ROI rows: one word before the interior, ceil(mw / 4) interior words, one after; the
        // strength plane has the ROI's geometry (and holds the word queue before it is cleared);
        // the pixel queue has one 16-bit entry per interior pixel
        g.fastRoiPitch = 4 * ((mw + 3) / 4 + 2);
        if (const char* e = getenv("ORBX_FAST_PITCH")) {  // measurement knob: a wider pitch (multiple of 4) changes the LDS bank pattern
            const int want = atoi(e) & ~3;
            if (want > g.fastRoiPitch && want <= 128) g.fastRoiPitch = want;
        }
        // LDS carve-up for a set of levels (largest cell mw x mh, row pitch at least minPitch)
        if (const char* e = getenv("ORBX_FAST_GROUPS")) {
            h->fastOneGroup = atoi(e) == 1;
            h->fastAlwaysSplit = atoi(e) == 2;
        }
        auto layout = [&](int mw_, int mh_, int minPitch) {
            FastGroup G;
            memset(&G, 0, sizeof(G));
            G.roiPitch = std::max(4 * ((mw_ + 3) / 4 + 2), minPitch);
            G.roiMagic = div_magic((u32)G.roiPitch);
            G.copyRows = std::max(64 / (G.roiPitch / 4), 1);  // (cells are at most 59 + 6 pixels wide: 19 words)
            G.copyMagic = div_magic((u32)(G.roiPitch / 4));
            G.planeOff = align_up(G.roiPitch * (mh_ + 6), 16);
            G.queueOff = G.planeOff + align_up(G.roiPitch * (mh_ + 6), 16);
            G.ldsBytes = G.queueOff + align_up(2 * mw_ * mh_ + 8, 16);
            return G;
        };
        const bool forcedPitch = g.fastRoiPitch != 4 * ((mw + 3) / 4 + 2);
        // Levels whose own buffers fit FAST_LDS_SMALL bytes form the first group when that buys anything:
        // some other level needs more, and the group as a whole (its widest and its tallest cell) still fits.
        h->fastLevelGroup.assign(g.nlevels, 0);
        int smw = 0, smh = 0, nSmall = 0, nLarge = 0;
        for (int l = 0; l < g.nlevels; l++) {
            if (g.lv[l].nCells <= 0) continue;
            if (layout(g.lv[l].wCell, g.lv[l].hCell, 0).ldsBytes <= FAST_LDS_SMALL) {
                smw = std::max(smw, g.lv[l].wCell);
                smh = std::max(smh, g.lv[l].hCell);
                nSmall++;
            } else {
                h->fastLevelGroup[l] = 1;
                nLarge++;
            }
        }
        const bool split = !forcedPitch && !h->fastOneGroup && nSmall > 0 && nLarge > 0 &&
                           layout(smw, smh, 0).ldsBytes <= FAST_LDS_SMALL;
        h->fastGroups.clear();
        h->fastWhole = layout(mw, mh, g.fastRoiPitch);  // all cells in one launch: passes too small to fill the GPU
        if (split) {
            int lmw = 1, lmh = 1;
            for (int l = 0; l < g.nlevels; l++)
                if (g.lv[l].nCells > 0 && h->fastLevelGroup[l] == 1) {
                    lmw = std::max(lmw, g.lv[l].wCell);
                    lmh = std::max(lmh, g.lv[l].hCell);
                }
            h->fastGroups.push_back(layout(smw, smh, 0));
            h->fastGroups.push_back(layout(lmw, lmh, 0));
        } else {
            h->fastLevelGroup.assign(g.nlevels, 0);
            h->fastGroups.push_back(layout(mw, mh, g.fastRoiPitch));
        }
    }
    g.pyrFrameBytes = align_up((int)std::max<i64>(pyrOff, 64), 256);
    g.blurFrameBytes = align_up((int)blurOff, 256);
    h->qtGlobal = maxLds > (size_t)QT_MAX_LDS;
    h->qtLds = (maxLds + 255) & ~(size_t)255;
    {
        // one workgroup walks all candidates of a level several times and ranks its nodes: levels with
        // thousands of nodes (4K frames with 10^4 features) want more threads than VGA levels do
        int maxQuota = 0;
        for (int l = 0; l < g.nlevels; l++) maxQuota = std::max(maxQuota, g.lv[l].quota);
        h->qtThreads = maxQuota >= 1400 ? 1024 : maxQuota >= 700 ? 512 : 256;
        // a handle for a few frames at a time is used for latency: the widest workgroup finishes a level soonest
        if (h->cfg.max_batch <= 4) h->qtThreads = 1024;
    }
    if (const char* e = getenv("ORBX_QT_THREADS")) h->qtThreads = atoi(e);
    if (const char* e = getenv("ORBX_RESIZE_ROWS")) h->resizeRows = atoi(e) != 0;  // measurement knob
    if (const char* e = getenv("ORBX_RESIZE_LOGW")) h->resizeLogW = atoi(e);
    if (const char* e = getenv("ORBX_PYR_FUSED")) h->pyrFused = atoi(e);
    if (const char* e = getenv("ORBX_BLUR_ROWS")) h->blurRows = atoi(e) != 0;
    if (const char* e = getenv("ORBX_BLUR_LOGW")) h->blurLogW = atoi(e);
    if (const char* e = getenv("ORBX_NPIPES")) h->maxPipes = std::min(std::max(atoi(e), 1), ORBX_PIPES);
    if (const char* e = getenv("ORBX_FORK_BLUR")) h->forkBlur = std::min(std::max(atoi(e), 0), 2);
    if (const char* e = getenv("ORBX_NO_REPACK")) h->noRepack = atoi(e) != 0;
    if (const char* e = getenv("ORBX_FAST_GENERIC")) h->fastGenericPitch = atoi(e) != 0;
    if (const char* e = getenv("ORBX_BLUR_LDS")) h->blurLds = std::min(std::max(atoi(e), 0), 65536);
    if (h->qtThreads != 256 && h->qtThreads != 512 && h->qtThreads != 1024) h->qtThreads = 256;

    // resize taps (cv::resize INTER_LINEAR, 8-bit, 11-bit fixed-point weights)
    std::vector<ResizeTap> taps;
    for (int l = 1; l < g.nlevels; l++) {
        const int sw = g.lv[l - 1].w, sh = g.lv[l - 1].h, dw = g.lv[l].w, dh = g.lv[l].h;
        const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
        h->tapXOff[l] = taps.size();
        for (int dx = 0; dx < dw; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = (int)floorf(fx);
            fx -= sx;
            if (sx < 0) {
                fx = 0;
                sx = 0;
            }
            if (sx >= sw - 1) {
                fx = 0;
                sx = sw - 1;
            }
            ResizeTap t;
            t.i0 = (u16)sx;
            t.i1 = (u16)std::min(sx + 1, sw - 1);
            t.c0 = (short)cv_round_host((1.f - fx) * 2048);
            t.c1 = (short)cv_round_host(fx * 2048);
            taps.push_back(t);
        }
        {
            bool ok = true;
            const ResizeTap* xt = taps.data() + h->tapXOff[l];
            for (int gx = 0; gx * 4 < dw; gx++) {
                const int wb = xt[gx * 4].i0 >> 2;
                for (int j = 0; j < 4; j++) {
                    const ResizeTap& t = xt[std::min(gx * 4 + j, dw - 1)];
                    if (t.i0 < xt[gx * 4].i0 || t.i1 - xt[gx * 4].i0 > 7 || t.i1 - 4 * wb > 11 || t.i1 < t.i0 || t.i1 - t.i0 > 1)
                        ok = false;
                }
            }
            h->resizeStreamOk[l] = ok;
        }
        h->tapYOff[l] = taps.size();
        for (int dy = 0; dy < dh; dy++) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = (int)floorf(fy);
            fy -= sy;
            ResizeTap t;
            t.i0 = (u16)std::min(std::max(sy, 0), sh - 1);
            t.i1 = (u16)std::min(std::max(sy + 1, 0), sh - 1);
            t.c0 = (short)cv_round_host((1.f - fy) * 2048);
            t.c1 = (short)cv_round_host(fy * 2048);
            taps.push_back(t);
        }
        {
            // the streaming kernel walks source rows forwards: second taps must not decrease,
            // a first tap is the second tap or the row before it, and weights stay in [0, 2048]
            const ResizeTap* yt = taps.data() + h->tapYOff[l];
            for (int dy = 0; dy < dh; dy++) {
                const ResizeTap& t = yt[dy];
                if (t.i1 < t.i0 || t.i1 - t.i0 > 1 || (dy > 0 && (t.i1 < yt[dy - 1].i1 || t.i0 < yt[dy - 1].i0)) || t.c0 < 0 ||
                    t.c1 < 0 || t.c0 > 2048 || t.c1 > 2048)
                    h->resizeStreamOk[l] = false;
            }
            // per source row: the (at most two) destination rows that finish on it; the table
            // starts on an even entry so that a row's pair is one 16-byte load
            if (taps.size() & 1) taps.push_back(ResizeTap());
            h->tapSOff[l] = taps.size();
            std::vector<ResizeTap> bySrc((size_t)2 * sh);
            for (size_t i = 0; i < bySrc.size(); i++) {
                bySrc[i].i0 = 0xFFFF;
                bySrc[i].i1 = 0;
                bySrc[i].c0 = bySrc[i].c1 = 0;
            }
            yt = taps.data() + h->tapYOff[l];
            for (int dy = 0; dy < dh; dy++) {
                const ResizeTap& t = yt[dy];
                ResizeTap* e = &bySrc[(size_t)2 * t.i1];
                if (e[0].i0 != 0xFFFF) e++;
                if (e[0].i0 != 0xFFFF) {
                    h->resizeStreamOk[l] = false;
                    break;
                }
                e->i0 = (u16)dy;
                e->i1 = t.i0 == t.i1 ? 1 : 0;
                e->c0 = t.c0;
                e->c1 = t.c1;
            }
            taps.insert(taps.end(), bySrc.begin(), bySrc.end());
        }
    }
    if (taps.size() > h->tapsCap) {
        free_dev(h->dTaps);
        h->dTaps = nullptr;
        h->tapsCap = 0;
        ORBX_HIP_CHECK(hipMalloc((void**)&h->dTaps, taps.size() * sizeof(ResizeTap)));
        h->tapsCap = taps.size();
    }
    if (!taps.empty()) {
        ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
        ORBX_HIP_CHECK(hipMemcpy(h->dTaps, taps.data(), taps.size() * sizeof(ResizeTap), hipMemcpyHostToDevice));
    }
    // arguments of the one-launch pyramid
    h->pyrFrameOk = g.nlevels > 1 && g.pyrFrameBytes < 0x7FFFFFFFll;
    memset(&h->pyrArgs, 0, sizeof(h->pyrArgs));
    for (int l = 1; l < g.nlevels; l++) {
        PyramidLevel& A = h->pyrArgs.lv[l];
        A.sw = g.lv[l - 1].w;
        A.sh = g.lv[l - 1].h;
        A.dw = g.lv[l].w;
        A.dh = g.lv[l].h;
        A.srcPitch = l > 1 ? g.lv[l - 1].pyrPitch : 0;  // level 0 is the caller's image: filled in at launch
        A.dstPitch = g.lv[l].pyrPitch;
        A.srcOff = l > 1 ? (u32)g.lv[l - 1].pyrOff : 0u;
        A.dstOff = (u32)g.lv[l].pyrOff;
        {
            const int G = (A.dw + 3) / 4, full = G / 64, rem = G % 64;
            A.chunks = (G + 63) / 64;
            A.stripGroups = 0;
            A.stripLogW = 0;
            static const int stripMax = getenv("ORBX_PYR_STRIP") ? atoi(getenv("ORBX_PYR_STRIP")) : PYR_STRIP_MAX;
            if (full >= 1 && rem > 0 && rem <= min(stripMax, 64)) {  // a thin remainder: blocks of rows instead of a chunk of walks
                A.chunks = full;
                A.stripGroups = rem;
                A.stripLogW = 0;
                while ((1 << A.stripLogW) < rem) A.stripLogW++;
            }
        }
        A.xt = h->dTaps + h->tapXOff[l];
        A.yt = h->dTaps + h->tapYOff[l];
        A.st = h->dTaps + h->tapSOff[l];
        if (!h->resizeStreamOk[l] || A.sw < 8 || A.chunks > PYR_WAVES || (i64)A.sh * std::max(A.srcPitch, 1) > 0x7FFFFFFFll) h->pyrFrameOk = false;
    }
    // tables of the one-walk-per-level kernel
    {
        if (const char* e = getenv("ORBX_WALK")) h->walkMode = atoi(e);
        std::vector<unsigned char> wt;
        auto pad16 = [&]() { wt.resize((wt.size() + 15) & ~(size_t)15, 0); };
        bool ok = true;
        for (int l = 0; l < g.nlevels && ok; l++) {
            const LevelGeom& L = g.lv[l];
            const int G = (L.w + 3) / 4;
            const bool next = l + 1 < g.nlevels;
            if (L.w < 8 || L.h < 4) ok = false;
            std::vector<u16> fd((size_t)G + 1, 0);
            const ResizeTap* xt = next ? taps.data() + h->tapXOff[l + 1] : nullptr;
            const int dw = next ? g.lv[l + 1].w : 0, Gd = (dw + 3) / 4;
            if (next && ok) {
                if (!h->resizeStreamOk[l + 1]) ok = false;
                for (int d = 0; d < dw && ok; d++) {
                    if (xt[d].i1 < xt[d].i0 || xt[d].i1 - xt[d].i0 > 1 || (d > 0 && xt[d].i0 < xt[d - 1].i0)) ok = false;
                }
                int d = 0;
                for (int gI = 0; gI <= G; gI++) {
                    while (d < dw && (xt[d].i0 >> 2) < gI) d++;
                    fd[(size_t)gI] = (u16)d;
                }
                for (int gI = 0; gI < G && ok; gI++)
                    if (fd[(size_t)gI + 1] - fd[(size_t)gI] > 4) ok = false;  // a lane packs at most four pixels
                for (int kd = 0; kd < Gd && ok; kd++)                          // a destination word gathers from at most two lanes
                    for (int t = 1; t < 4; t++)
                        if (4 * kd + t < dw && (xt[4 * kd + t].i0 >> 2) - (xt[4 * kd].i0 >> 2) > 1) ok = false;
            }
            pad16();
            h->walk[l].fdOff = wt.size();
            wt.insert(wt.end(), (unsigned char*)fd.data(), (unsigned char*)(fd.data() + fd.size()));
            for (int wi = 0; wi < 2 && ok; wi++) {
                const int Wd = wi == 0 ? 64 : 32;
                std::vector<WalkChunk> ch;
                int covered = 0;
                if (next) {
                    int k0 = 0;
                    while (k0 < Gd && ok) {
                        const int g0 = xt[4 * k0].i0 >> 2;
                        while (g0 > covered) {
                            ch.push_back(WalkChunk{covered, Gd, 0, 0});
                            covered += Wd;
                        }
                        int nk = 0;
                        while (k0 + nk < Gd) {
                            bool fits = true;
                            for (int t = 0; t < 4; t++) {
                                const int d = 4 * (k0 + nk) + t;
                                if (d < dw && (xt[d].i0 >> 2) - g0 > Wd - 1) fits = false;
                            }
                            if (!fits) break;
                            nk++;
                        }
                        if (nk == 0) {
                            ok = false;
                            break;
                        }
                        ch.push_back(WalkChunk{g0, k0, nk, 0});
                        covered = std::max(covered, g0 + Wd);
                        k0 += nk;
                    }
                }
                while (covered < G) {
                    ch.push_back(WalkChunk{covered, Gd, 0, 0});
                    covered += Wd;
                }
                pad16();
                h->walk[l].chOff[wi] = wt.size();
                h->walk[l].nCh[wi] = (int)ch.size();
                wt.insert(wt.end(), (unsigned char*)ch.data(), (unsigned char*)(ch.data() + ch.size()));
            }
        }
        h->walkOk = ok && g.nlevels >= 1;
        if (h->walkOk) {
            if (wt.size() > h->walkCap) {
                free_dev(h->dWalk);
                h->dWalk = nullptr;
                h->walkCap = 0;
                ORBX_HIP_CHECK(hipMalloc((void**)&h->dWalk, wt.size()));
                h->walkCap = wt.size();
            }
            ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
            ORBX_HIP_CHECK(hipMemcpy(h->dWalk, wt.data(), wt.size(), hipMemcpyHostToDevice));
        }
    }
    // FAST cell records (reference src/ORBextractor.cc:799-817 for the ROI bounds)
    std::vector<FastCell> cells((size_t)g.totalCells);
    memset(cells.data(), 0, cells.size() * sizeof(FastCell));
    int recNext = 0;
    for (size_t gi = 0; gi < h->fastGroups.size(); gi++) {
        h->fastGroups[gi].recBegin = recNext;
        for (int l = 0; l < g.nlevels; l++) {
            if (h->fastLevelGroup[l] != (int)gi) continue;
            const LevelGeom& L = g.lv[l];
            const int maxBX = L.w - MIN_BORDER, maxBY = L.h - MIN_BORDER;
            for (int c = 0; c < L.nCells; c++) {
                const int ci = c / L.nCols, cj = c - ci * L.nCols;
                const int iniX = MIN_BORDER + cj * L.wCell, iniY = MIN_BORDER + ci * L.hCell;
                const int rw = std::max(std::min(iniX + L.wCell + 6, maxBX) - iniX, 0);
                const int rh = std::max(std::min(iniY + L.hCell + 6, maxBY) - iniY, 0);
                const int iw = std::max(rw - 6, 1);
                const int nk = (iw + 3) / 4, nWc = nk + 2;
                auto magic = [](int d) { return d <= 1 ? 0u : (u32)(0x100000000ull / (u32)d) + 1u; };
                FastCell& F = cells[(size_t)recNext++];
                F.w[12] = (u32)(L.cellBase + c);
                F.w[0] = (u32)l | ((u32)rw << 8) | ((u32)rh << 16) | ((u32)(iniX & 3) << 24);
                F.w[1] = (u32)iniX | ((u32)iniY << 16);
                F.w[2] = (u32)(L.slotBase + c * L.cellCap);
                F.w[3] = (u32)L.cellCap | ((u32)nk << 16) | ((u32)(64 / nk) << 24);
                F.w[4] = (u32)(cj * L.wCell + 3) | ((u32)(ci * L.hCell + 3) << 16);
                F.w[5] = magic(nWc);
                F.w[6] = magic(nk);
                F.w[7] = (u32)nWc | ((u32)(64 / nWc) << 8);
                F.w[8] = l > 0 ? (u32)(L.pyrOff + (i64)iniY * L.pyrPitch + (iniX & ~3)) : 0u;
                F.w[9] = (u32)L.pyrPitch;
                F.w[10] = magic(rw);
                {  // ROI rows are copied in groups of rowsW rows, the first 8 groups through registers
                    const int rowsW = 64 / nWc;
                    const int nFull = std::min(rh / rowsW, 8);
                    const int rem = nFull < 8 ? rh - nFull * rowsW : 0;
                    F.w[11] = (u32)nFull | ((u32)rem << 8);
                }
                if (l > 0 && L.pyrOff + (i64)L.pyrPitch * L.h > 0xFFFFFFFFll) {
                    orbx_set_error("internal: pyramid block exceeds 4 GiB");
                    return ORBX_ERR_UNSUPPORTED;
                }
                if (L.cellCap > 0xffff || rw > 255 || rh > 255) {
                    orbx_set_error("internal: cell record overflow");
                    return ORBX_ERR_UNSUPPORTED;
                }
            }
        }
        h->fastGroups[gi].nCells = recNext - h->fastGroups[gi].recBegin;
        h->fastWhole.recBegin = 0;
        h->fastWhole.nCells = recNext;
        if (getenv("ORBX_VERBOSE"))
            fprintf(stderr, "orbx: FAST group %d: %d cells, row pitch %d, %d bytes of LDS\n", (int)gi, h->fastGroups[gi].nCells,
                    h->fastGroups[gi].roiPitch, h->fastGroups[gi].ldsBytes);
    }
    if (cells.size() > h->cellsCap) {
        free_dev(h->dCells);
        h->dCells = nullptr;
        h->cellsCap = 0;
        ORBX_HIP_CHECK(hipMalloc((void**)&h->dCells, cells.size() * sizeof(FastCell)));
        h->cellsCap = cells.size();
    }
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    ORBX_HIP_CHECK(hipMemcpy(h->dCells, cells.data(), cells.size() * sizeof(FastCell), hipMemcpyHostToDevice));
    h->geomValid = true;
    h->scratchSig = 0;  // force a scratch size check
    return ORBX_OK;
}

int ensure_scratch(orbx_handle* h, int batch) {
    const Geom& g = h->g;
    const size_t sig = (size_t)g.w * 73856093u ^ (size_t)g.h * 19349663u;
    if (h->scratchSig == sig && batch <= h->scratchBatch) return ORBX_OK;
    ORBX_HIP_CHECK(hipDeviceSynchronize());
    free_dev(h->basePyr); free_dev(h->baseBlur); free_dev(h->baseCellCount); free_dev(h->baseSlots);
    free_dev(h->baseKeyPacked); free_dev(h->baseKeyNode); free_dev(h->baseSel); free_dev(h->baseSelCount);
    free_dev(h->baseNodeTables);
    h->baseNodeTables = nullptr;
    h->basePyr = h->baseBlur = nullptr;
    h->baseCellCount = h->baseSlots = h->baseKeyPacked = h->baseSel = nullptr;
    h->baseKeyNode = nullptr;
    h->baseSelCount = nullptr;
    h->scratchBatch = 0;
    const size_t B = (size_t)batch;
    ORBX_HIP_CHECK(hipMalloc((void**)&h->basePyr, B * g.pyrFrameBytes + 256));
    ORBX_HIP_CHECK(hipMalloc((void**)&h->baseBlur, B * g.blurFrameBytes + 256));
    ORBX_HIP_CHECK(hipMalloc((void**)&h->baseCellCount, B * g.totalCells * sizeof(u32)));
    ORBX_HIP_CHECK(hipMalloc((void**)&h->baseSlots, B * g.totalSlots * sizeof(u32)));
    ORBX_HIP_CHECK(hipMalloc((void**)&h->baseKeyPacked, B * g.totalSlots * sizeof(u32)));
    ORBX_HIP_CHECK(hipMalloc((void**)&h->baseKeyNode, B * g.totalSlots * sizeof(u16)));
    ORBX_HIP_CHECK(hipMalloc((void**)&h->baseSel, B * g.totalSel * sizeof(u32)));
    ORBX_HIP_CHECK(hipMalloc((void**)&h->baseSelCount, B * ORBX_MAX_LEVELS * sizeof(int)));
    if (h->qtGlobal) ORBX_HIP_CHECK(hipMalloc((void**)&h->baseNodeTables, B * g.nlevels * h->qtLds));
    h->scratchBatch = batch;
    h->scratchSig = sig;
    return ORBX_OK;
}

// Points the d* views at the slice that starts `frame0` frames into the scratch.
void select_scratch(orbx_handle* h, size_t frame0) {
    const Geom& g = h->g;
    h->dPyr = h->basePyr + frame0 * g.pyrFrameBytes;
    h->dBlur = h->baseBlur + frame0 * g.blurFrameBytes;
    h->dCellCount = h->baseCellCount + frame0 * g.totalCells;
    h->dSlots = h->baseSlots + frame0 * g.totalSlots;
    h->dKeyPacked = h->baseKeyPacked + frame0 * g.totalSlots;
    h->dKeyNode = h->baseKeyNode + frame0 * g.totalSlots;
    h->dSel = h->baseSel + frame0 * g.totalSel;
    h->dSelCount = h->baseSelCount + frame0 * ORBX_MAX_LEVELS;
    h->dNodeTables = h->baseNodeTables ? h->baseNodeTables + frame0 * g.nlevels * h->qtLds : nullptr;
}

// Streams and events of pipe p >= 1, created on first use.
int ensure_pipe(orbx_handle* h, int p) {
    orbx_handle::Pipe& P = h->pipes[p];
    if (P.st) return ORBX_OK;
    ORBX_HIP_CHECK(hipStreamCreateWithFlags(&P.st, hipStreamNonBlocking));
    ORBX_HIP_CHECK(hipStreamCreateWithFlags(&P.aux, hipStreamNonBlocking));
    ORBX_HIP_CHECK(hipEventCreateWithFlags(&P.evPyramid, hipEventDisableTiming));
    ORBX_HIP_CHECK(hipEventCreateWithFlags(&P.evBlur, hipEventDisableTiming));
    ORBX_HIP_CHECK(hipEventCreateWithFlags(&P.evDone, hipEventDisableTiming));
    return ORBX_OK;
}

struct StageTimer {
    orbx_handle* h;
    int stage;
    hipStream_t st;
    hipEvent_t stop = nullptr;
    StageTimer(orbx_handle* h_, int stage_, hipStream_t st_ = nullptr) : h(h_), stage(stage_), st(st_ ? st_ : h_->stream) {
        if (!h->profiling) return;
        StageProfile& p = h->prof;
        if (p.used[stage] == p.ev[stage].size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            p.ev[stage].push_back(std::make_pair(a, b));
        }
        std::pair<hipEvent_t, hipEvent_t>& e = p.ev[stage][p.used[stage]++];
        (void)hipEventRecord(e.first, st);
        stop = e.second;
    }
    ~StageTimer() {
        if (stop) (void)hipEventRecord(stop, st);
    }
};

// ORBX_RESIZE_BH=a,b,c,... forces the band height of the streaming resize for levels 1, 2, 3, ...
// (a measurement knob).  Parsed once; the table is read-only afterwards.
struct ForcedBands {
    int rows[ORBX_MAX_LEVELS];
    ForcedBands() {
        for (int i = 0; i < ORBX_MAX_LEVELS; i++) rows[i] = 0;
        if (const char* e = getenv("ORBX_RESIZE_BH")) {
            int i = 1;
            for (const char* q = e; *q && i < ORBX_MAX_LEVELS; i++) {
                rows[i] = atoi(q);
                while (*q && *q != ',') q++;
                if (*q == ',') q++;
            }
        }
    }
};
int forced_band_rows(int level) {
    static const ForcedBands fb;  // initialised once, thread-safe
    return fb.rows[level];
}

// Enqueue the whole pipeline for `n` frames starting at device image img0.
int run_pass(orbx_handle* h, ImgRef img0, int n, orbx_keypoint* dKps, u8* dDesc, int cap, int* dN) {
    const Geom& g = h->g;
    hipStream_t st = h->stream;
    // the one-launch pyramid wants a workgroup per frame on most compute units; small batches go level by level
    // (640x480, serialised, us per frame, one launch against seven: 128 frames 0.88 / 0.75, 192 frames 0.63 / 0.69,
    // 256 frames 0.48 / 0.66)
    const i64 imgBytes0 = (i64)(g.h - 1) * img0.pitch + g.w;
    const bool fusedPyramid = h->pyrFrameOk && h->resizeRows && imgBytes0 <= 0x7FFFFFFFll &&
                              (h->pyrFused >= 0 ? h->pyrFused != 0 : n >= 160);
    // One walk per level (blurred level + next level) where the geometry allows it: level 0 word-aligned,
    // 32-bit lane offsets for 64 frames of any of the three blocks.
    const bool walk = h->walkOk && h->walkMode != 0 && h->blurRows && h->resizeRows &&
                      ((((uintptr_t)img0.base) | (uintptr_t)img0.pitch | (uintptr_t)img0.frameStride) & 3u) == 0 &&
                      img0.pitch >= align_up(g.w, 4) && (u64)img0.frameStride * 64 < 0x7FFFFFFFull &&
                      (u64)g.pyrFrameBytes * 64 < 0x7FFFFFFFull && (u64)g.blurFrameBytes * 64 < 0x7FFFFFFFull;
    if (walk) {
        StageTimer t(h, ORBX_STAGE_PYRAMID);
        for (int l = 0; l < g.nlevels; l++) {
            const LevelGeom& L = g.lv[l];
            const bool next = l + 1 < g.nlevels;
            WalkArgs A;
            memset(&A, 0, sizeof(A));
            A.src = img0;
            if (l > 0) {
                A.src.base = h->dPyr + L.pyrOff;
                A.src.frameStride = g.pyrFrameBytes;
                A.src.pitch = L.pyrPitch;
            }
            A.w = L.w;
            A.h = L.h;
            A.blurOff = (u32)L.blurOff;
            A.blurPitch = L.blurPitch;
            if (next) {
                A.dw = g.lv[l + 1].w;
                A.dh = g.lv[l + 1].h;
                A.dstPitch = g.lv[l + 1].pyrPitch;
                A.dstOff = (u32)g.lv[l + 1].pyrOff;
                A.xt = h->dTaps + h->tapXOff[l + 1];
                A.st = h->dTaps + h->tapSOff[l + 1];
            }
            A.firstDest = (const u16*)(h->dWalk + h->walk[l].fdOff);
            // lanes = W column groups x 64 / W frames: the split with the fewest idle lanes
            const int G = (L.w + 3) / 4;
            int wi = 0;
            {
                const double e64 = (double)G * n / ((double)h->walk[l].nCh[0] * 64 * n);
                const double e32 = (double)G * n / ((double)h->walk[l].nCh[1] * 32 * ((n + 1) / 2) * 2);
                if (e32 > e64 + 1e-9) wi = 1;
                if (h->blurLogW == 6) wi = 0;
                if (h->blurLogW == 5) wi = 1;
            }
            A.logW = wi == 0 ? 6 : 5;
            A.chunks = (const WalkChunk*)(h->dWalk + h->walk[l].chOff[wi]);
            A.nChunks = h->walk[l].nCh[wi];
            // rows per band: BLUR_BH where that gives the launch enough waves to fill the GPU, fewer on the small
            // levels (a band re-reads six rows of its neighbours)
            const int K = 64 >> A.logW;
            static const int wantWaves = getenv("ORBX_WALK_WAVES") ? atoi(getenv("ORBX_WALK_WAVES")) : 8192;
            A.bandRows = BLUR_BH;
            while (A.bandRows > 14 && (i64)((n + K - 1) / K) * A.nChunks * ((L.h + A.bandRows - 1) / A.bandRows) < wantWaves) A.bandRows -= 14;
            A.bands = (L.h + A.bandRows - 1) / A.bandRows;
            const int tasks = A.bands * A.nChunks;
            dim3 grid((n + K - 1) / K, (tasks + 3) / 4, 1), block(256, 1, 1);
            const bool he = h->cfg.blur_variant == 2;
            auto kw = next ? (he ? k_level_walk<true, true> : k_level_walk<false, true>)
                           : (he ? k_level_walk<true, false> : k_level_walk<false, false>);
            hipLaunchKernelGGL(kw, grid, block, 0, st, A, h->dPyr, g.pyrFrameBytes, h->dBlur, g.blurFrameBytes, h->gaussK[0], h->gaussK[1],
                               h->gaussK[2], h->gaussK[3], n);
        }
    } else if (fusedPyramid) {
        StageTimer t(h, ORBX_STAGE_PYRAMID);
        PyramidArgs args = h->pyrArgs;
        args.lv[1].srcPitch = img0.pitch;
        hipLaunchKernelGGL(k_pyramid_frame, dim3(n), dim3(64 * PYR_WAVES), 0, st, img0, h->dPyr, g.pyrFrameBytes, (int)imgBytes0, args,
                           g.nlevels);
    } else {
        StageTimer t(h, ORBX_STAGE_PYRAMID);
        for (int l = 1; l < g.nlevels; l++) {
            ImgRef src = img0;
            if (l > 1) {
                src.base = h->dPyr + g.lv[l - 1].pyrOff;
                src.frameStride = g.pyrFrameBytes;
                src.pitch = g.lv[l - 1].pyrPitch;
            }
            const LevelGeom& L = g.lv[l];
            const bool srcAligned = ((((uintptr_t)src.base) | (uintptr_t)src.pitch | (uintptr_t)src.frameStride) & 3u) == 0;
            if (srcAligned && h->resizeStreamOk[l]) {
                // destination rows per band: 32 where that still gives the launch enough waves, fewer
                // on the small levels (a band re-reads about two source rows of its neighbour)
                int bandRows = RESIZE_BH;
                // small batches cannot fill the GPU with 32-row bands: shorter bands give more, shorter walks
                while (bandRows > 4 && (i64)n * ((((L.w + 3) / 4) * ((L.h + bandRows - 1) / bandRows) + 255) / 256) < 1024) bandRows >>= 1;
                {
                    const int forced = forced_band_rows(l);  // measurement knob
                    if (forced > 0) bandRows = forced;
                }
                const int G = (L.w + 3) / 4, nb = (L.h + bandRows - 1) / bandRows;
                // frames of 3 x the pyramid or image stride must fit the kernel's 32-bit lane offsets
                const bool rowsOk = h->resizeRows && (u64)src.frameStride * 64 < 0xFFFFFFFFull && (u64)g.pyrFrameBytes * 64 < 0xFFFFFFFFull &&
                                    nb <= 65535 && g.lv[l - 1].w >= 8;
                if (rowsOk) {
                    // lanes = W column groups x 64 / W frames: the W (>= 8) that wastes the fewest lanes, wider on ties
                    int logW = 6;
                    double bestEff = 0;
                    for (int lw = 6; lw >= 3; lw--) {
                        const int Wd = 1 << lw, K = 64 >> lw;
                        const double eff = (double)G * n / ((double)((G + Wd - 1) / Wd) * Wd * ((n + K - 1) / K) * K);
                        if (eff > bestEff + 1e-9) {
                            bestEff = eff;
                            logW = lw;
                        }
                    }
                    if (h->resizeLogW >= 3 && h->resizeLogW <= 6) logW = h->resizeLogW;
                    const int Wd = 1 << logW, K = 64 >> logW;
                    dim3 grid((n + K - 1) / K, (nb + 3) / 4, (G + Wd - 1) / Wd), block(256, 1, 1);
                    hipLaunchKernelGGL(k_resize_rows, grid, block, 0, st, src, g.lv[l - 1].w, g.lv[l - 1].h, h->dPyr + L.pyrOff,
                                       L.pyrPitch, g.pyrFrameBytes, L.w, L.h, h->dTaps + h->tapXOff[l],
                                       h->dTaps + h->tapYOff[l], h->dTaps + h->tapSOff[l], bandRows, logW, n);
                } else {
                    const int items = G * nb;
                    dim3 grid(n, (items + 255) / 256, 1), block(256, 1, 1);
                    hipLaunchKernelGGL(k_resize_stream, grid, block, 0, st, src, src.pitch / 4 - 1, g.lv[l - 1].h,
                                       h->dPyr + L.pyrOff, L.pyrPitch, g.pyrFrameBytes, L.w, L.h, h->dTaps + h->tapXOff[l],
                                       h->dTaps + h->tapYOff[l], h->dTaps + h->tapSOff[l], bandRows);
                }
            } else {
                dim3 grid((L.w + 255) / 256, (L.h + 3) / 4, n), block(64, 4, 1);
                hipLaunchKernelGGL(k_resize, grid, block, 0, st, src, h->dPyr + L.pyrOff, L.pyrPitch,
                                   g.pyrFrameBytes, L.w, L.h, h->dTaps + h->tapXOff[l], h->dTaps + h->tapYOff[l]);
            }
        }
    }
    const bool fork = !walk && h->overlapBlur && h->forkBlur && h->auxStream != nullptr;
    // forkBlur 1: the blur starts when the pyramid is done and runs beside FAST and the quadtree;
    // forkBlur 2: it starts when FAST is done and runs beside the quadtree only (FAST and the blur
    // both live on vector-instruction issue and slow each other down; the quadtree is a chain of
    // barriers and scans that leaves most of a compute unit idle)
    const bool lateFork = fork && h->forkBlur == 2;
    hipStream_t bst = fork ? h->auxStream : st;
    auto launch_blur = [&]() -> int {
        if (fork) {
            ORBX_HIP_CHECK(hipEventRecord(h->evPyramid, st));
            ORBX_HIP_CHECK(hipStreamWaitEvent(h->auxStream, h->evPyramid, 0));
        }
        {
            StageTimer t(h, ORBX_STAGE_BLUR, bst);
            // Level 0 is the caller's buffer: the streaming kernel reads it as aligned 32-bit
            // words, so it needs 4-byte aligned rows that hold the last word of the row in full.
            const bool aligned0 = ((((uintptr_t)img0.base) | (uintptr_t)img0.pitch | (uintptr_t)img0.frameStride) & 3u) == 0 &&
                                  img0.pitch >= align_up(g.w, 4) && g.h >= 4 && g.w >= 4;
            const int first = aligned0 ? 0 : 1;
            if (!aligned0) {
                dim3 grid(g.lv[0].blurTiles, n, 1), block(256, 1, 1);
                hipLaunchKernelGGL(k_blur, grid, block, 0, bst, g, img0, (const u8*)h->dPyr, h->dBlur, h->gaussK[0],
                                   h->gaussK[1], h->gaussK[2], h->gaussK[3], 0, h->cfg.blur_variant == 2 ? 1 : 0);
            }
            for (int l = 1; l < g.nlevels; l++) {
                if (g.lv[l].h >= 4 && g.lv[l].w >= 4) continue;
                dim3 grid(g.lv[l].blurTiles, n, 1), block(256, 1, 1);
                hipLaunchKernelGGL(k_blur, grid, block, 0, bst, g, img0, (const u8*)h->dPyr, h->dBlur, h->gaussK[0],
                                   h->gaussK[1], h->gaussK[2], h->gaussK[3], g.lv[l].blurTileBase, h->cfg.blur_variant == 2 ? 1 : 0);
            }
            const int itemBegin = g.lv[first].blurItemBase;
            if (first < g.nlevels && g.totalBlurItems > itemBegin) {
                // blockIdx.x counts from the first level's item base
                Geom gs = g;
                for (int l = first; l < g.nlevels; l++) gs.lv[l].blurItemBase -= itemBegin;
                // lane offsets are 32-bit: K frames of either image must stay below 2^31 bytes
                const bool rowsOk = h->blurRows && (u64)img0.frameStride * 64 < 0x7FFFFFFFull && (u64)g.pyrFrameBytes * 64 < 0x7FFFFFFFull &&
                                    (u64)g.blurFrameBytes * 64 < 0x7FFFFFFFull;
                if (rowsOk) {
                    // one lane split (W column groups x 64 / W frames) for all levels: the one with the fewest idle lanes
                    BlurRowsArgs A;
                    memset(&A, 0, sizeof(A));
                    // (64 x 1 or 32 x 2: narrower rows per frame cost more in the memory pipeline than idle lanes do,
                    // 0.58 and 0.81 us per frame at 16 x 4 and 8 x 8 against 0.51 at 32 x 2 for 640x480)
                    double bestEff = -1;
                    for (int lw = 6; lw >= 5; lw--) {
                        const int Wd = 1 << lw, K = 64 >> lw;
                        double used = 0, slots = 0;
                        for (int l = first; l < g.nlevels; l++) {
                            if (g.lv[l].h < 4 || g.lv[l].w < 4) continue;
                            used += (double)g.lv[l].blurGroups * g.lv[l].h * n;
                            slots += (double)((g.lv[l].blurGroups + Wd - 1) / Wd) * Wd * g.lv[l].h * ((n + K - 1) / K) * K;
                        }
                        const double eff = slots > 0 ? used / slots : 0;
                        if (eff > bestEff + 1e-9) {
                            bestEff = eff;
                            A.logW = lw;
                        }
                    }
                    if (h->blurLogW >= 3 && h->blurLogW <= 6) A.logW = h->blurLogW;
                    const int Wd = 1 << A.logW, K = 64 >> A.logW;
                    int tasks = 0;
                    for (int l = 0; l < g.nlevels; l++) {
                        A.taskBase[l] = tasks;
                        A.chunks[l] = std::max((g.lv[l].blurGroups + Wd - 1) / Wd, 1);
                        if (l >= first && g.lv[l].h >= 4 && g.lv[l].w >= 4) tasks += A.chunks[l] * g.lv[l].blurBands;
                    }
                    A.taskBase[g.nlevels] = tasks;
                    A.firstLevel = first;
                    dim3 grid((n + K - 1) / K, (tasks + 3) / 4, 1), block(256, 1, 1);
                    auto kb = h->cfg.blur_variant == 2 ? k_blur_rows<true> : k_blur_rows<false>;
                    hipLaunchKernelGGL(kb, grid, block, h->blurLds, bst, g, img0, (const u8*)h->dPyr, h->dBlur, h->gaussK[0], h->gaussK[1],
                                       h->gaussK[2], h->gaussK[3], A, n);
                } else {
                    dim3 grid(n, (g.totalBlurItems - itemBegin + 255) / 256, 1), block(256, 1, 1);
                    auto kb = h->cfg.blur_variant == 2 ? k_blur_stream<true> : k_blur_stream<false>;
                    hipLaunchKernelGGL(kb, grid, block, h->blurLds, bst, gs, img0, (const u8*)h->dPyr, h->dBlur, h->gaussK[0],
                                       h->gaussK[1], h->gaussK[2], h->gaussK[3], first);
                }
            }
        }
        if (fork) {
            ORBX_HIP_CHECK(hipEventRecord(h->evBlur, h->auxStream));
        }
        return ORBX_OK;
    };
    if (!lateFork && !walk) {
        const int rc = launch_blur();
        if (rc != ORBX_OK) return rc;
    }
    {
        StageTimer t(h, ORBX_STAGE_FAST);
        const int aligned0 = ((((uintptr_t)img0.base) | (uintptr_t)img0.pitch | (uintptr_t)img0.frameStride) & 3u) == 0;
        static const int ldsPad = getenv("ORBX_FAST_LDS_PAD") ? atoi(getenv("ORBX_FAST_LDS_PAD")) : 0;  // measurement knob: fewer waves per compute unit
        // Grouping buys waves per compute unit, which only matters when there are waves to fill the GPU with
        // (256 compute units x 32); below that one launch is cheaper than two (one 640x480 frame: 190 against 199 us)
        const bool whole = h->fastGroups.size() > 1 && !h->fastAlwaysSplit && (i64)n * g.totalCells < 4 * 8192;
        const FastGroup* groups = whole ? &h->fastWhole : h->fastGroups.data();
        const size_t ngroups = whole ? 1 : h->fastGroups.size();
        for (size_t gi = 0; gi < ngroups; gi++) {
            const FastGroup& G = groups[gi];
            if (G.nCells <= 0) continue;
            dim3 grid(n, (G.nCells + FAST_CPW - 1) / FAST_CPW, 1), block(64, 1, 1);
            // the kernel is instantiated for every ROI pitch a cell of 30 .. 59 pixels can have
            auto kf = k_fast<0>;
            switch (G.roiPitch) {
                case 40: kf = k_fast<40>; break;
                case 44: kf = k_fast<44>; break;
                case 48: kf = k_fast<48>; break;
                case 52: kf = k_fast<52>; break;
                case 56: kf = k_fast<56>; break;
                case 60: kf = k_fast<60>; break;
                case 64: kf = k_fast<64>; break;
                case 68: kf = k_fast<68>; break;
                default: break;
            }
            if (h->fastGenericPitch) kf = k_fast<0>;
            hipLaunchKernelGGL(kf, grid, block, G.ldsBytes + ldsPad, st, g, img0, aligned0, (const u8*)h->dPyr,
                               (const FastCell*)h->dCells, h->dCellCount, h->dSlots, std::min(std::max(h->cfg.ini_th_fast, 0), 255),
                               std::min(std::max(h->cfg.min_th_fast, 0), 255), G);
        }
    }
    if (lateFork) {
        const int rc = launch_blur();
        if (rc != ORBX_OK) return rc;
    }
    {
        StageTimer t(h, ORBX_STAGE_QUADTREE);
        dim3 grid(n, g.nlevels, 1), block(h->qtThreads, 1, 1);
        if (h->qtGlobal) {
            // node tables in global memory: one slice per (frame, level) of this pass
            hipLaunchKernelGGL((k_quadtree<1024, true>), grid, dim3(1024, 1, 1), 0, st, g, (const u32*)h->dCellCount,
                               (const u32*)h->dSlots, h->dKeyPacked, h->dKeyNode, h->dSel, h->dSelCount, h->dErr,
                               h->dNodeTables, h->qtLds);
        } else {
            auto kq = h->qtThreads == 1024 ? k_quadtree<1024, false> : h->qtThreads == 512 ? k_quadtree<512, false> : k_quadtree<256, false>;
            hipLaunchKernelGGL(kq, grid, block, h->qtLds, st, g, (const u32*)h->dCellCount,
                               (const u32*)h->dSlots, h->dKeyPacked, h->dKeyNode, h->dSel, h->dSelCount, h->dErr,
                               (unsigned char*)nullptr, (size_t)0);
        }
    }
    if (fork) ORBX_HIP_CHECK(hipStreamWaitEvent(st, h->evBlur, 0));
    {
        StageTimer t(h, ORBX_STAGE_DESCRIBE);
        const int nChunks = (std::min(cap, g.totalSel) + 4 * DESC_KPW - 1) / (4 * DESC_KPW);
        dim3 grid(8 * ((n + 7) / 8) * nChunks, 1, 1), block(256, 1, 1);
        auto kd = h->cfg.fma_variant == 1 ? k_describe<true> : k_describe<false>;
        hipLaunchKernelGGL(kd, grid, block, 0, st, g, img0, (const u8*)h->dPyr, (const u8*)h->dBlur,
                           (const u32*)h->dSel, (const int*)h->dSelCount, dKps, dDesc, cap, dN, h->dErr, n, nChunks,
                           div_magic((u32)nChunks));
    }
    ORBX_HIP_CHECK(hipGetLastError());
    return ORBX_OK;
}

// Waits for the handle's stream, then reads and clears the error word of the kernels.
int check_device_error(orbx_handle* h) {
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    const int e = *(volatile int*)h->hErr;
    if (e) {
        *(volatile int*)h->hErr = 0;
        orbx_set_error(e == 2 ? "keypoint capacity per frame too small: output truncated"
                              : "quadtree node capacity exceeded");
        return ORBX_ERR_CAPACITY;
    }
    return ORBX_OK;
}

// True when `p` is page-locked host memory the GPU can copy from directly (hipHostMalloc, hipHostRegister).
bool is_pinned_host(const void* p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary malloc'd pointer is "invalid value" to the runtime
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// Host images in, host keypoints out (the shape of the reference's call: the Frame constructor owns a
// cv::Mat).  Passes of `chunk` frames go through a ring of three slots: worker threads copy the caller's
// rows into pinned memory at the staged pitch (pageable memory cannot be copied from asynchronously, and
// one thread moves far less than the link does), an upload stream brings the slot to the device, the
// handle's stream runs the kernels, a download stream returns the slot's results to pinned memory, and
// the workers copy them to the caller's arrays: the host copy of pass k + 1, the upload of pass k, the
// kernels of pass k - 1 and the download of pass k - 2 run side by side.  Images and output arrays that are
// already page-locked (hipHostMalloc / hipHostRegister, orbx_register_host) skip the host copies.
int extract_host_passes(orbx_handle* h, const uint8_t* imgs, int B, int w, int hh, size_t row_stride, size_t img_stride,
                        orbx_keypoint* kps, uint8_t* desc, int cap, int* n_out, int chunk) {
    const size_t sp = (size_t)align_up(w, 64);  // staged rows are padded to 64 bytes: aligned word loads everywhere
    const size_t inBytes = (size_t)chunk * sp * hh;
    const size_t kpBytes = (size_t)chunk * cap * sizeof(orbx_keypoint), dsBytes = (size_t)chunk * cap * 32;
    const size_t outBytes = kpBytes + dsBytes + (((size_t)chunk * sizeof(int) + 255) & ~(size_t)255);
    const int npass = (B + chunk - 1) / chunk;
    const int S = std::min(3, npass);
    if (!h->copyIn) {
        ORBX_HIP_CHECK(hipStreamCreateWithFlags(&h->copyIn, hipStreamNonBlocking));
        ORBX_HIP_CHECK(hipStreamCreateWithFlags(&h->copyOut, hipStreamNonBlocking));
    }
    if (!h->pool) {
        int workers = 7;  // + the calling thread
        if (const char* e = getenv("ORBX_HOST_THREADS")) workers = std::min(std::max(atoi(e) - 1, 0), 63);
        workers = std::min(workers, std::max((int)std::thread::hardware_concurrency() - 1, 0));
        h->pool = new CopyPool(workers);
    }
    for (int s = 0; s < S; s++) {
        orbx_handle::HostSlot& H = h->hostSlots[s];
        if (!H.evIn) {
            ORBX_HIP_CHECK(hipEventCreateWithFlags(&H.evIn, hipEventDisableTiming));
            ORBX_HIP_CHECK(hipEventCreateWithFlags(&H.evComp, hipEventDisableTiming));
            ORBX_HIP_CHECK(hipEventCreateWithFlags(&H.evOut, hipEventDisableTiming));
        }
        if (H.inCap < inBytes) {
            ORBX_HIP_CHECK(hipDeviceSynchronize());
            if (H.pinIn) (void)hipHostFree(H.pinIn);
            free_dev(H.dIn);
            H.pinIn = H.dIn = nullptr;
            H.inCap = 0;
            ORBX_HIP_CHECK(hipHostMalloc((void**)&H.pinIn, inBytes + 256, hipHostMallocDefault));
            ORBX_HIP_CHECK(hipMalloc((void**)&H.dIn, inBytes + 256));
            H.inCap = inBytes;
        }
        if (H.outCap < outBytes) {
            ORBX_HIP_CHECK(hipDeviceSynchronize());
            if (H.pinOut) (void)hipHostFree(H.pinOut);
            free_dev(H.dOut);
            H.pinOut = H.dOut = nullptr;
            H.outCap = 0;
            ORBX_HIP_CHECK(hipHostMalloc((void**)&H.pinOut, outBytes, hipHostMallocDefault));
            ORBX_HIP_CHECK(hipMalloc((void**)&H.dOut, outBytes));
            H.outCap = outBytes;
        }
        H.busy = false;
    }
    const bool pinnedIn = is_pinned_host(imgs) && row_stride == sp && img_stride == sp * (size_t)hh;
    const bool pinnedOut = is_pinned_host(kps) && is_pinned_host(desc) && is_pinned_host(n_out);
    // results of the pass a slot holds: pinned staging -> the caller's arrays
    auto deliver = [&](orbx_handle::HostSlot& H) -> int {
        if (!H.busy) return ORBX_OK;
        ORBX_HIP_CHECK(hipEventSynchronize(H.evOut));
        H.busy = false;
        if (pinnedOut) return ORBX_OK;
        const int n = H.n, f0 = H.f0;
        const int* cnt = (const int*)(H.pinOut + kpBytes + dsBytes);
        memcpy(n_out + f0, cnt, (size_t)n * sizeof(int));
        const u8* pk = H.pinOut;
        const u8* pd = H.pinOut + kpBytes;
        h->pool->run(n, [&](int f) {
            // only the entries a frame produced travel on (the rest of its capacity holds nothing)
            const size_t m = (size_t)std::min(std::max(cnt[f], 0), cap);
            memcpy(kps + (size_t)(f0 + f) * cap, pk + (size_t)f * cap * sizeof(orbx_keypoint), m * sizeof(orbx_keypoint));
            memcpy(desc + (size_t)(f0 + f) * cap * 32, pd + (size_t)f * cap * 32, m * 32);
        });
        return ORBX_OK;
    };
    int rc = ORBX_OK;
    hipStream_t st = h->stream;
    for (int k = 0; k < npass; k++) {
        orbx_handle::HostSlot& H = h->hostSlots[k % S];
        if ((rc = deliver(H))) return rc;  // also: the slot's previous kernels and download are done
        const int f0 = k * chunk, n = std::min(chunk, B - f0);
        H.f0 = f0;
        H.n = n;
        const u8* up = H.pinIn;
        if (pinnedIn) {
            up = imgs + (size_t)f0 * img_stride;
        } else {
            h->pool->run(n, [&](int f) {
                const u8* src = imgs + (size_t)(f0 + f) * img_stride;
                u8* dst = H.pinIn + (size_t)f * sp * hh;
                if (row_stride == sp) {
                    memcpy(dst, src, sp * (size_t)(hh - 1) + (size_t)w);
                } else {
                    for (int y = 0; y < hh; y++) memcpy(dst + (size_t)y * sp, src + (size_t)y * row_stride, (size_t)w);
                }
            });
        }
        ORBX_HIP_CHECK(hipMemcpyAsync(H.dIn, up, (size_t)n * sp * hh - (sp - (size_t)w), hipMemcpyHostToDevice, h->copyIn));
        ORBX_HIP_CHECK(hipEventRecord(H.evIn, h->copyIn));
        ORBX_HIP_CHECK(hipStreamWaitEvent(st, H.evIn, 0));
        ImgRef img0;
        img0.base = H.dIn;
        img0.pitch = (int)sp;
        img0.frameStride = (i64)sp * hh;
        orbx_keypoint* dK = (orbx_keypoint*)H.dOut;
        u8* dD = H.dOut + kpBytes;
        int* dN = (int*)(H.dOut + kpBytes + dsBytes);
        if ((rc = run_pass(h, img0, n, dK, dD, cap, dN))) return rc;
        ORBX_HIP_CHECK(hipEventRecord(H.evComp, st));
        ORBX_HIP_CHECK(hipStreamWaitEvent(h->copyOut, H.evComp, 0));
        if (pinnedOut) {
            ORBX_HIP_CHECK(hipMemcpyAsync(kps + (size_t)f0 * cap, dK, (size_t)n * cap * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, h->copyOut));
            ORBX_HIP_CHECK(hipMemcpyAsync(desc + (size_t)f0 * cap * 32, dD, (size_t)n * cap * 32, hipMemcpyDeviceToHost, h->copyOut));
            ORBX_HIP_CHECK(hipMemcpyAsync(n_out + f0, dN, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, h->copyOut));
        } else if (n == chunk) {
            ORBX_HIP_CHECK(hipMemcpyAsync(H.pinOut, H.dOut, outBytes, hipMemcpyDeviceToHost, h->copyOut));
        } else {
            ORBX_HIP_CHECK(hipMemcpyAsync(H.pinOut, dK, (size_t)n * cap * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, h->copyOut));
            ORBX_HIP_CHECK(hipMemcpyAsync(H.pinOut + kpBytes, dD, (size_t)n * cap * 32, hipMemcpyDeviceToHost, h->copyOut));
            ORBX_HIP_CHECK(hipMemcpyAsync(H.pinOut + kpBytes + dsBytes, dN, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, h->copyOut));
        }
        ORBX_HIP_CHECK(hipEventRecord(H.evOut, h->copyOut));
        // (the next upload into this slot follows deliver(), which has waited for this download and so for these kernels)
        H.busy = true;
        h->lastImg0 = img0;
        h->lastBatch = n;
        h->lastFirstFrame = f0;
    }
    for (int k = npass; k < npass + S; k++)
        if ((rc = deliver(h->hostSlots[k % S]))) return rc;
    return check_device_error(h);
}

int extract_host_pipelined(orbx_handle* h, const uint8_t* imgs, int B, int w, int hh, size_t row_stride, size_t img_stride,
                           orbx_keypoint* kps, uint8_t* desc, int cap, int* n_out, int chunk) {
    const int rc = extract_host_passes(h, imgs, B, w, hh, row_stride, img_stride, kps, desc, cap, n_out, chunk);
    if (rc != ORBX_OK) {
        // a pass failed with copies of its neighbours still in flight: nothing of this call may touch the
        // slots or the caller's arrays once it has returned
        (void)hipDeviceSynchronize();
        for (auto& H : h->hostSlots) H.busy = false;
    }
    return rc;
}

}  // namespace

extern "C" {

const char* orbx_version(void) { return "orbx 0.2 (gfx950)"; }

int orbx_create(const orbx_config* cfg, orbx_handle** out) {
    if (!cfg || !out) {
        orbx_set_error("null argument");
        return ORBX_ERR_INVALID;
    }
    *out = nullptr;
    if (cfg->nlevels < 1 || cfg->nlevels > ORBX_MAX_LEVELS || cfg->nfeatures < 1 ||
        !(cfg->scale_factor > 1.0f) || cfg->max_batch < 1 || cfg->max_width < 1 || cfg->max_height < 1 ||
        cfg->blur_variant < 0 || cfg->blur_variant > 2 || cfg->fma_variant < 0 || cfg->fma_variant > 1) {
        orbx_set_error("invalid configuration");
        return ORBX_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
        orbx_set_error("no usable HIP device (count=%d, requested %d)", ndev, cfg->device);
        return ORBX_ERR_NO_DEVICE;
    }
    if (hipSetDevice(cfg->device) != hipSuccess) {
        orbx_set_error("hipSetDevice(%d) failed", cfg->device);
        return ORBX_ERR_NO_DEVICE;
    }
    orbx_handle* h = new orbx_handle();
    h->cfg = *cfg;
    h->device = cfg->device;
    make_tables(h);
    int rc = device_init_once(cfg->device);
    if (rc == ORBX_OK && hipStreamCreateWithFlags(&h->ownStream, hipStreamNonBlocking) != hipSuccess) {
        orbx_set_error("hipStreamCreate failed");
        rc = ORBX_ERR_HIP;
    }
    h->stream = h->ownStream;
    if (rc == ORBX_OK && (hipStreamCreateWithFlags(&h->auxStream, hipStreamNonBlocking) != hipSuccess ||
                          hipEventCreateWithFlags(&h->evPyramid, hipEventDisableTiming) != hipSuccess ||
                          hipEventCreateWithFlags(&h->evBlur, hipEventDisableTiming) != hipSuccess)) {
        orbx_set_error("hipStreamCreate/hipEventCreate failed");
        rc = ORBX_ERR_HIP;
    }
    if (rc == ORBX_OK) {
        if (hipHostMalloc((void**)&h->hErr, 64, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&h->dErr, h->hErr, 0) != hipSuccess) {
            orbx_set_error("hipHostMalloc failed");
            rc = ORBX_ERR_HIP;
        } else {
            *h->hErr = 0;
        }
    }
    if (rc == ORBX_OK) rc = compute_geometry(h, cfg->max_width, cfg->max_height);
    if (rc == ORBX_OK) rc = ensure_scratch(h, cfg->max_batch);
    if (rc != ORBX_OK) {
        orbx_destroy(h);
        return rc;
    }
    *out = h;
    return ORBX_OK;
}

void orbx_destroy(orbx_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    free_dev(h->basePyr); free_dev(h->baseBlur); free_dev(h->baseCellCount); free_dev(h->baseSlots);
    free_dev(h->baseKeyPacked); free_dev(h->baseKeyNode); free_dev(h->baseSel); free_dev(h->baseSelCount);
    free_dev(h->baseNodeTables);
    for (int p = 1; p < ORBX_PIPES; p++) {
        orbx_handle::Pipe& P = h->pipes[p];
        if (P.st) (void)hipStreamDestroy(P.st);
        if (P.aux) (void)hipStreamDestroy(P.aux);
        if (P.evPyramid) (void)hipEventDestroy(P.evPyramid);
        if (P.evBlur) (void)hipEventDestroy(P.evBlur);
        if (P.evDone) (void)hipEventDestroy(P.evDone);
    }
    if (h->evStart) (void)hipEventDestroy(h->evStart);
    for (auto& H : h->hostSlots) {
        if (H.pinIn) (void)hipHostFree(H.pinIn);
        if (H.pinOut) (void)hipHostFree(H.pinOut);
        free_dev(H.dIn);
        free_dev(H.dOut);
        if (H.evIn) (void)hipEventDestroy(H.evIn);
        if (H.evComp) (void)hipEventDestroy(H.evComp);
        if (H.evOut) (void)hipEventDestroy(H.evOut);
    }
    if (h->copyIn) (void)hipStreamDestroy(h->copyIn);
    if (h->copyOut) (void)hipStreamDestroy(h->copyOut);
    delete h->pool;
    if (h->hErr) (void)hipHostFree(h->hErr);
    free_dev(h->dTaps); free_dev(h->dWalk); free_dev(h->dCells); free_dev(h->dAlignStage);
    for (int s = 0; s < ORBX_STAGE_COUNT; s++)
        for (auto& e : h->prof.ev[s]) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    if (h->auxStream) {
        (void)hipStreamSynchronize(h->auxStream);
        (void)hipStreamDestroy(h->auxStream);
    }
    if (h->evPyramid) (void)hipEventDestroy(h->evPyramid);
    if (h->evBlur) (void)hipEventDestroy(h->evBlur);
    if (h->ownStream) (void)hipStreamDestroy(h->ownStream);
    delete h;
}

int orbx_register_host(void* ptr, size_t bytes) {
    if (!ptr || !bytes) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // the failure must not surface in a later, unrelated call
        e = hipHostRegister(ptr, bytes, hipHostRegisterPortable | hipHostRegisterMapped);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        orbx_set_error("hipHostRegister of %zu bytes failed: %s (the locked-memory limit of the process may be too small)", bytes,
                       hipGetErrorString(e));
        return ORBX_ERR_HIP;
    }
    return ORBX_OK;
}

int orbx_unregister_host(void* ptr) {
    if (!ptr) return ORBX_OK;
    const hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        orbx_set_error("hipHostUnregister failed: %s", hipGetErrorString(e));
        return ORBX_ERR_HIP;
    }
    return ORBX_OK;
}

int orbx_set_stream(orbx_handle* h, void* hip_stream) {
    if (!h) return ORBX_ERR_INVALID;
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->stream = (hipStream_t)hip_stream;  // NULL is the legacy default stream, like any other handle
    return ORBX_OK;
}

int orbx_reset_stream(orbx_handle* h) {
    if (!h) return ORBX_ERR_INVALID;
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->stream = h->ownStream;
    return ORBX_OK;
}

int orbx_set_overlap(orbx_handle* h, int on) {
    if (!h) return ORBX_ERR_INVALID;
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->overlapBlur = on != 0;
    return ORBX_OK;
}

int orbx_synchronize(orbx_handle* h) {
    if (!h) return ORBX_ERR_INVALID;
    return check_device_error(h);
}

int orbx_max_keypoints(const orbx_handle* h) { return h ? h->g.totalSel : 0; }
int orbx_levels(const orbx_handle* h) { return h->cfg.nlevels; }
float orbx_scale_factor(const orbx_handle* h) { return (float)(double)h->cfg.scale_factor; }
const float* orbx_scale_factors(const orbx_handle* h) { return h->scale; }
const float* orbx_inv_scale_factors(const orbx_handle* h) { return h->invScale; }
const float* orbx_level_sigma2(const orbx_handle* h) { return h->sigma2; }
const float* orbx_inv_level_sigma2(const orbx_handle* h) { return h->invSigma2; }
const int32_t* orbx_features_per_level(const orbx_handle* h) { return h->quota; }

int orbx_extract_batch(orbx_handle* h, const uint8_t* imgs, int B, int w, int hh, size_t row_stride,
                       size_t img_stride, orbx_keypoint* kps, uint8_t* desc, int cap_per_frame,
                       int* n_out, int mem_space) {
    if (!h || B < 0 || !n_out) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (B == 0) return ORBX_OK;
    if (w <= 0 || hh <= 0 || !imgs) {
        // an empty image returns silently (reference src/ORBextractor.cc:1055-1056)
        if (mem_space == ORBX_MEM_HOST)
            for (int f = 0; f < B; f++) n_out[f] = 0;
        else
            ORBX_HIP_CHECK(hipMemsetAsync(n_out, 0, sizeof(int) * B, h->stream));
        return ORBX_OK;
    }
    if (!kps || !desc || cap_per_frame < 1 || row_stride < (size_t)w) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    if (w > h->cfg.max_width || hh > h->cfg.max_height) {
        orbx_set_error("image %dx%d exceeds configured maximum %dx%d", w, hh, h->cfg.max_width, h->cfg.max_height);
        return ORBX_ERR_CAPACITY;
    }
    ORBX_HIP_CHECK(hipSetDevice(h->device));
    int rc = compute_geometry(h, w, hh);
    if (rc) return rc;
    const int chunk = std::min(B, h->cfg.max_batch);
    const int npass = (B + chunk - 1) / chunk;
    const int npipes = (mem_space == ORBX_MEM_DEVICE && h->overlapBlur) ? std::min(std::min(ORBX_PIPES, h->maxPipes), npass) : 1;
    rc = ensure_scratch(h, chunk * npipes);
    if (rc) return rc;
    select_scratch(h, 0);
    const int cap = cap_per_frame;
    hipStream_t mainStream = h->stream, mainAux = h->auxStream;
    hipEvent_t mainEvPyr = h->evPyramid, mainEvBlur = h->evBlur;
    if (npipes > 1) {
        if (!h->evStart) ORBX_HIP_CHECK(hipEventCreateWithFlags(&h->evStart, hipEventDisableTiming));
        ORBX_HIP_CHECK(hipEventRecord(h->evStart, mainStream));
        for (int p = 1; p < npipes; p++) {
            if ((rc = ensure_pipe(h, p))) return rc;
            ORBX_HIP_CHECK(hipStreamWaitEvent(h->pipes[p].st, h->evStart, 0));
        }
    }
    if (mem_space == ORBX_MEM_HOST) {
        // the transfers, not the kernels, bound this path: shorter passes (128 frames of 640x480 still keep the
        // kernels far ahead of the link) fill and drain the ring of slots sooner
        static const int hostPass = getenv("ORBX_HOST_PASS") ? std::max(atoi(getenv("ORBX_HOST_PASS")), 1) : 128;
        return extract_host_pipelined(h, imgs, B, w, hh, row_stride, img_stride, kps, desc, cap, n_out, std::min(chunk, hostPass));
    }
    // Device-resident frames whose layout is not word-aligned are repacked pass by pass (ORBX_NO_REPACK=1
    // keeps the unaligned forms of the kernels, which the tests compare with the repacked path).
    const bool repackDevice = mem_space == ORBX_MEM_DEVICE && !h->noRepack && hh <= 65535 && chunk <= 65535 &&
                              (((((uintptr_t)imgs) | (uintptr_t)row_stride | (uintptr_t)img_stride) & 3u) != 0 ||
                               row_stride < (size_t)align_up(w, 4));
    if (repackDevice) {
        const size_t need = (size_t)chunk * npipes * align_up(w, 64) * hh;
        if (need > h->alignStageCap) {
            ORBX_HIP_CHECK(hipDeviceSynchronize());
            free_dev(h->dAlignStage);
            h->dAlignStage = nullptr;
            h->alignStageCap = 0;
            ORBX_HIP_CHECK(hipMalloc((void**)&h->dAlignStage, need + 256));
            h->alignStageCap = need;
        }
    }
    for (int f0 = 0; f0 < B; f0 += chunk) {
        const int n = std::min(chunk, B - f0);
        ImgRef img0;
        {
            img0.base = imgs + (size_t)f0 * img_stride;
            img0.pitch = (int)row_stride;
            img0.frameStride = (i64)img_stride;
            const int p = (f0 / chunk) % npipes;
            if (p > 0) {
                h->stream = h->pipes[p].st;
                h->auxStream = h->pipes[p].aux;
                h->evPyramid = h->pipes[p].evPyramid;
                h->evBlur = h->pipes[p].evBlur;
            }
            select_scratch(h, (size_t)p * chunk);
            if (repackDevice) {
                // one read and one write of the frame buys the word-aligned forms of the resize, the
                // FAST cell copy and the blur for level 0 (1241 x 376: 146 k -> 2xx k frames/s)
                const int sp = align_up(w, 64);
                u8* slice = h->dAlignStage + (size_t)p * chunk * sp * hh;
                dim3 grid((((w + 3) / 4) + 255) / 256, hh, n), block(256, 1, 1);
                hipLaunchKernelGGL(k_repack_rows, grid, block, 0, h->stream, img0, w, slice, sp, (i64)sp * hh);
                img0.base = slice;
                img0.pitch = sp;
                img0.frameStride = (i64)sp * hh;
            }
            rc = run_pass(h, img0, n, kps + (size_t)f0 * cap, desc + (size_t)f0 * cap * 32, cap, n_out + f0);
            h->stream = mainStream;
            h->auxStream = mainAux;
            h->evPyramid = mainEvPyr;
            h->evBlur = mainEvBlur;
            if (rc) return rc;
        }
        h->lastImg0 = img0;
        h->lastBatch = n;
        h->lastFirstFrame = f0;
    }
    // the handle's stream continues when every pipe has finished its passes
    for (int p = 1; p < npipes; p++) {
        ORBX_HIP_CHECK(hipEventRecord(h->pipes[p].evDone, h->pipes[p].st));
        ORBX_HIP_CHECK(hipStreamWaitEvent(mainStream, h->pipes[p].evDone, 0));
    }
    return ORBX_OK;
}

int orbx_extract(orbx_handle* h, const uint8_t* img, int w, int hh, size_t stride, orbx_keypoint* kps,
                 uint8_t* desc, int cap, int* n_out) {
    return orbx_extract_batch(h, img, 1, w, hh, stride, stride * (size_t)(hh > 0 ? hh : 0), kps, desc, cap, n_out,
                              ORBX_MEM_HOST);
}

int orbx_get_level(orbx_handle* h, int frame, int level, const uint8_t** ptr, int* w, int* hh, size_t* stride) {
    if (!h || !ptr || !h->geomValid || level < 0 || level >= h->g.nlevels) {
        orbx_set_error("invalid argument");
        return ORBX_ERR_INVALID;
    }
    const int fl = frame - h->lastFirstFrame;
    if (fl < 0 || fl >= h->lastBatch) {
        orbx_set_error("frame %d is not resident (last pass held frames %d..%d)", frame, h->lastFirstFrame,
                       h->lastFirstFrame + h->lastBatch - 1);
        return ORBX_ERR_INVALID;
    }
    const LevelGeom& L = h->g.lv[level];
    if (level == 0) {
        *ptr = h->lastImg0.base + (i64)fl * h->lastImg0.frameStride;
        if (stride) *stride = (size_t)h->lastImg0.pitch;
    } else {
        *ptr = h->dPyr + (i64)fl * h->g.pyrFrameBytes + L.pyrOff;
        if (stride) *stride = (size_t)L.pyrPitch;
    }
    if (w) *w = L.w;
    if (hh) *hh = L.h;
    return ORBX_OK;
}

int orbx_profile_enable(orbx_handle* h, int on) {
    if (!h) return ORBX_ERR_INVALID;
    h->profiling = on != 0;
    return ORBX_OK;
}

static int profile_collect(orbx_handle* h) {
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    StageProfile& p = h->prof;
    for (int s = 0; s < ORBX_STAGE_COUNT; s++) {
        for (size_t i = 0; i < p.used[s]; i++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, p.ev[s][i].first, p.ev[s][i].second) == hipSuccess) {
                p.ms[s] += ms;
                p.launches[s]++;
            }
        }
        p.used[s] = 0;
    }
    return ORBX_OK;
}

int orbx_profile_read(orbx_handle* h, float* ms, int* launches, int n) {
    if (!h || !ms) return ORBX_ERR_INVALID;
    int rc = profile_collect(h);
    if (rc) return rc;
    for (int s = 0; s < n && s < ORBX_STAGE_COUNT; s++) {
        ms[s] = (float)h->prof.ms[s];
        if (launches) launches[s] = h->prof.launches[s];
    }
    return ORBX_OK;
}

int orbx_profile_reset(orbx_handle* h) {
    if (!h) return ORBX_ERR_INVALID;
    int rc = profile_collect(h);
    if (rc) return rc;
    for (int s = 0; s < ORBX_STAGE_COUNT; s++) {
        h->prof.ms[s] = 0;
        h->prof.launches[s] = 0;
    }
    return ORBX_OK;
}

int orbx_debug_sincos(orbx_handle* h, const float* angles, int n, float* sin_out, float* cos_out) {
    if (!h || n < 0 || !angles || !sin_out || !cos_out) return ORBX_ERR_INVALID;
    if (n == 0) return ORBX_OK;
    float *d = nullptr;
    ORBX_HIP_CHECK(hipMalloc((void**)&d, (size_t)n * 3 * sizeof(float)));
    hipError_t e = hipMemcpy(d, angles, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_sincos, dim3((n + 255) / 256), dim3(256), 0, h->stream, (const float*)d, d + n, d + 2 * (size_t)n, n);
        e = hipStreamSynchronize(h->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(sin_out, d + n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(cos_out, d + 2 * (size_t)n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    ORBX_HIP_CHECK(e);
    return ORBX_OK;
}

int orbx_debug_tables(orbx_handle* h, uint32_t* out, int n) {
    if (!h || !out || n < 0) return ORBX_ERR_INVALID;
    DescribeTables t;
    ORBX_HIP_CHECK(hipSetDevice(h->device));
    ORBX_HIP_CHECK(hipMemcpyFromSymbol(&t, HIP_SYMBOL(d_tables), sizeof(t)));
    const u32* w = (const u32*)&t;
    for (int i = 0; i < n && i < (int)(sizeof(t) / 4); i++) out[i] = w[i];
    return ORBX_OK;
}

int orbx_debug_atan2(orbx_handle* h, const float* y, const float* x, int n, float* out) {
    if (!h || n < 0 || !y || !x || !out) return ORBX_ERR_INVALID;
    if (n == 0) return ORBX_OK;
    float* d = nullptr;
    ORBX_HIP_CHECK(hipMalloc((void**)&d, (size_t)n * 3 * sizeof(float)));
    hipError_t e = hipMemcpy(d, y, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_atan2, dim3((n + 255) / 256), dim3(256), 0, h->stream, (const float*)d, (const float*)(d + n), d + 2 * (size_t)n, n);
        e = hipStreamSynchronize(h->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, d + 2 * (size_t)n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    ORBX_HIP_CHECK(e);
    return ORBX_OK;
}

// Index of `frame` (numbered within the last call, as in orbx_get_level) inside the last pass,
// whose stage outputs are the ones still resident; -1 when it is not.
static int resident_frame(orbx_handle* h, int frame) {
    const int fl = frame - h->lastFirstFrame;
    if (fl < 0 || fl >= h->lastBatch) {
        orbx_set_error("frame %d is not resident (last pass held frames %d..%d)", frame, h->lastFirstFrame,
                       h->lastFirstFrame + h->lastBatch - 1);
        return -1;
    }
    return fl;
}

int orbx_debug_level(orbx_handle* h, int frame, int level, uint8_t* dst) {
    const uint8_t* ptr = nullptr;
    int w = 0, hh = 0;
    size_t stride = 0;
    int rc = orbx_get_level(h, frame, level, &ptr, &w, &hh, &stride);
    if (rc) return rc;
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    ORBX_HIP_CHECK(hipMemcpy2D(dst, w, ptr, stride, w, hh, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

int orbx_debug_blurred(orbx_handle* h, int frame, int level, uint8_t* dst, int* w, int* hh) {
    if (!h || !dst || !h->geomValid || level < 0 || level >= h->g.nlevels) return ORBX_ERR_INVALID;
    if ((frame = resident_frame(h, frame)) < 0) return ORBX_ERR_INVALID;
    const LevelGeom& L = h->g.lv[level];
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    ORBX_HIP_CHECK(hipMemcpy2D(dst, L.w, h->dBlur + (i64)frame * h->g.blurFrameBytes + L.blurOff, L.blurPitch, L.w,
                               L.h, hipMemcpyDeviceToHost));
    if (w) *w = L.w;
    if (hh) *hh = L.h;
    return ORBX_OK;
}

int orbx_debug_candidates(orbx_handle* h, int frame, int level, int* xs, int* ys, int* resp, int cap, int* n) {
    if (!h || !n || !h->geomValid || level < 0 || level >= h->g.nlevels) return ORBX_ERR_INVALID;
    if ((frame = resident_frame(h, frame)) < 0) return ORBX_ERR_INVALID;
    const Geom& g = h->g;
    const LevelGeom& L = g.lv[level];
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    std::vector<u32> cc(std::max(L.nCells, 1)), sl((size_t)std::max(L.nCells, 1) * L.cellCap);
    if (L.nCells > 0) {
        ORBX_HIP_CHECK(hipMemcpy(cc.data(), h->dCellCount + (size_t)frame * g.totalCells + L.cellBase,
                                 (size_t)L.nCells * sizeof(u32), hipMemcpyDeviceToHost));
        ORBX_HIP_CHECK(hipMemcpy(sl.data(), h->dSlots + (size_t)frame * g.totalSlots + L.slotBase,
                                 (size_t)L.nCells * L.cellCap * sizeof(u32), hipMemcpyDeviceToHost));
    }
    int k = 0;
    for (int c = 0; c < L.nCells; c++)
        for (u32 i = 0; i < cc[c]; i++, k++) {
            const u32 p = sl[(size_t)c * L.cellCap + i];
            if (k < cap) {
                xs[k] = ORBX_PX(p);
                ys[k] = ORBX_PY(p);
                resp[k] = ORBX_PR(p);
            }
        }
    *n = k;
    return ORBX_OK;
}

int orbx_debug_selected(orbx_handle* h, int frame, int level, int* xs, int* ys, int* resp, int cap, int* n) {
    if (!h || !n || !h->geomValid || level < 0 || level >= h->g.nlevels) return ORBX_ERR_INVALID;
    if ((frame = resident_frame(h, frame)) < 0) return ORBX_ERR_INVALID;
    const Geom& g = h->g;
    const LevelGeom& L = g.lv[level];
    ORBX_HIP_CHECK(hipStreamSynchronize(h->stream));
    int cnt = 0;
    ORBX_HIP_CHECK(hipMemcpy(&cnt, h->dSelCount + (size_t)frame * ORBX_MAX_LEVELS + level, sizeof(int),
                             hipMemcpyDeviceToHost));
    std::vector<u32> s(std::max(cnt, 1));
    if (cnt > 0)
        ORBX_HIP_CHECK(hipMemcpy(s.data(), h->dSel + (size_t)frame * g.totalSel + L.selBase, (size_t)cnt * sizeof(u32),
                                 hipMemcpyDeviceToHost));
    for (int i = 0; i < cnt && i < cap; i++) {
        xs[i] = ORBX_PX(s[i]);
        ys[i] = ORBX_PY(s[i]);
        resp[i] = ORBX_PR(s[i]);
    }
    *n = cnt;
    return ORBX_OK;
}

#ifdef ORBX_PYR_DIAG
// Diagnostic build only: points the k_pyramid_frame stamps at a caller's device buffer of
// frames * 16 * ORBX_MAX_LEVELS * 4 64-bit words (NULL turns them off).
int orbx_debug_pyr_diag(unsigned long long* dev_buf) {
    ORBX_HIP_CHECK(hipDeviceSynchronize());
    ORBX_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_pyrDiag), &dev_buf, sizeof(dev_buf)));
    return ORBX_OK;
}
#endif

#ifdef ORBX_FAST_DIAG
// Diagnostic build only: points the k_fast phase records at a caller's device buffer of
// 8 * frames * cells 64-bit words (NULL turns them off).
int orbx_debug_fast_diag(unsigned long long* dev_buf) {
    ORBX_HIP_CHECK(hipDeviceSynchronize());
    ORBX_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_fastDiag), &dev_buf, sizeof(dev_buf)));
    return ORBX_OK;
}
#endif

}  // extern "C"
