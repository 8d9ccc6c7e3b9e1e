// Attainable HBM rates on this GPU for the access shapes the streaming kernels use:
// one lane = V bytes of a row, a wave walks down R rows; read + write (copy) and read-only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline unsigned xor_t(unsigned a, unsigned b) { return a ^ b; }
__device__ inline uint2 xor_t(uint2 a, uint2 b) { return make_uint2(a.x ^ b.x, a.y ^ b.y); }
__device__ inline uint4 xor_t(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }
__device__ inline bool is_magic(unsigned a) { return a == 12345u; }
__device__ inline bool is_magic(uint2 a) { return a.x == 12345u; }
__device__ inline bool is_magic(uint4 a) { return a.x == 12345u; }

template <typename T, int PF>
__global__ __launch_bounds__(256) void k_copy(const T* __restrict__ src, T* __restrict__ dst, int wordsPerRow, int rows,
                                              int band, size_t frameWords, int doWrite) {
    const int f = blockIdx.x;
    const int nb = (rows + band - 1) / band;
    const int item = blockIdx.y * 256 + threadIdx.x;
    if (item >= wordsPerRow * nb) return;
    const int b = item / wordsPerRow, gx = item - b * wordsPerRow;
    const T* s = src + (size_t)f * frameWords + gx;
    T* d = dst + (size_t)f * frameWords + gx;
    const int y0 = b * band, y1 = min(y0 + band, rows);
    T acc = T();
    for (int y = y0; y < y1; y += PF) {
        T v[PF];
#pragma unroll
        for (int i = 0; i < PF; i++) v[i] = s[(size_t)min(y + i, rows - 1) * wordsPerRow];
#pragma unroll
        for (int i = 0; i < PF; i++) {
            if (doWrite) {
                if (y + i < y1) d[(size_t)(y + i) * wordsPerRow] = v[i];
            } else {
                acc = xor_t(acc, v[i]);
            }
        }
    }
    if (!doWrite && is_magic(acc)) d[0] = acc;
}

template <typename T, int PF>
void run(const char* name, int W, int H, int B, int band, int doWrite) {
    const int wordsPerRow = W / sizeof(T);
    const size_t frameWords = (size_t)wordsPerRow * H;
    T *src, *dst;
    CK(hipMalloc(&src, frameWords * B * sizeof(T)));
    CK(hipMalloc(&dst, frameWords * B * sizeof(T)));
    CK(hipMemset(src, 1, frameWords * B * sizeof(T)));
    CK(hipMemset(dst, 0, frameWords * B * sizeof(T)));
    const int nb = (H + band - 1) / band;
    dim3 grid(B, (wordsPerRow * nb + 255) / 256), block(256);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k_copy<T, PF>), grid, block, 0, 0, src, dst, wordsPerRow, H, band, frameWords, doWrite);
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int it = 0; it < reps; it++) hipLaunchKernelGGL((k_copy<T, PF>), grid, block, 0, 0, src, dst, wordsPerRow, H, band, frameWords, doWrite);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)frameWords * sizeof(T) * B * (doWrite ? 2 : 1) * reps;
    printf("%-28s band %3d pf %d %s: %7.1f GB/s  (%.3f ms/launch)\n", name, band, PF, doWrite ? "copy" : "read", bytes / ms / 1e6, ms / reps);
    CK(hipFree(src));
    CK(hipFree(dst));
}

int main() {
    const int W = 640, H = 480, B = 512;
    run<unsigned, 1>("dword/lane", W, H, B, 28, 1);
    run<unsigned, 4>("dword/lane", W, H, B, 28, 1);
    run<unsigned, 8>("dword/lane", W, H, B, 28, 1);
    run<unsigned, 8>("dword/lane", W, H, B, 56, 1);
    run<unsigned, 8>("dword/lane", W, H, B, 120, 1);
    run<uint2, 4>("dwordx2/lane", W, H, B, 28, 1);
    run<uint2, 8>("dwordx2/lane", W, H, B, 56, 1);
    run<uint4, 4>("dwordx4/lane", W, H, B, 28, 1);
    run<uint4, 8>("dwordx4/lane", W, H, B, 56, 1);
    run<unsigned, 8>("dword/lane", W, H, B, 56, 0);
    run<uint4, 8>("dwordx4/lane", W, H, B, 56, 0);
    return 0;
}
