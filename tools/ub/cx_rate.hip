// Micro-benchmark: the cost of one compare-exchange stage of the 16-lane sorting network (merge.h nnd_q16_cx) per wave --
// 64-bit keys (two DPP moves + v_cmp_lt_u64 + two selects) against 32-bit keys with a 32-bit payload (compare and selects can take
// the partner through DPP operands).   hipcc -O3 --offload-arch=gfx950 cx_rate.hip -o cx_rate && ./cx_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define QX1 0xB1
#define QX2 0x4E
#define QMIR 0x1B
#define HMIR 0x141
#define RMIR 0x140
template <int CTRL>
__device__ __forceinline__ uint32_t partner(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ void cx64(uint32_t &lo, uint32_t &hi, bool keep_max) {
    const uint32_t plo = partner<CTRL>(lo), phi = partner<CTRL>(hi);
    const bool less = (((uint64_t)phi << 32) | plo) < (((uint64_t)hi << 32) | lo);
    const bool take = less != keep_max;
    lo = take ? plo : lo;
    hi = take ? phi : hi;
}
template <int CTRL>
__device__ __forceinline__ void cx32(uint32_t &pay, uint32_t &key, bool keep_max) {
    const uint32_t pk = partner<CTRL>(key), pp = partner<CTRL>(pay);
    const bool less = pk < key;
    const bool take = less != keep_max;
    pay = take ? pp : pay;
    key = take ? pk : key;
}
template <int CTRL>
__device__ __forceinline__ void cx32x(uint32_t &pay, uint32_t &key, bool keep_max) {  // exact (key, pay) order with 32-bit compares
    const uint32_t pk = partner<CTRL>(key), pp = partner<CTRL>(pay);
    const bool less = (pk < key) | ((pk == key) & (pp < pay));
    const bool take = less != keep_max;
    pay = take ? pp : pay;
    key = take ? pk : key;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters) {
    const int lane = threadIdx.x & 63, j = lane & 15;
    uint32_t lo = threadIdx.x * 2654435761u + blockIdx.x, hi = (lo >> 7) * 40503u;
    const bool b0 = j & 1, b1 = j & 2, b2 = j & 4, b3 = j & 8;
    for (int it = 0; it < iters; it++) {
#define ST(C, B) if (MODE == 0) cx64<C>(lo, hi, B); else if (MODE == 1) cx32<C>(lo, hi, B); else cx32x<C>(lo, hi, B);
        ST(QX1, b0) ST(QMIR, b1) ST(QX1, b0) ST(HMIR, b2) ST(QX2, b1) ST(QX1, b0) ST(RMIR, b3) ST(QX2, b1) ST(QX1, b0) ST(HMIR, b2)
        lo += it; hi ^= lo;
    }
    out[blockIdx.x * 256 + threadIdx.x] = lo ^ hi;
}
int main() {
    uint32_t *d; hipMalloc(&d, 4 * 256 * 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, grid = 4096;
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9;
        for (int r = 0; r < 4; r++) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, iters);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, iters);
            else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, iters);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        // stages per SIMD: grid * 4 waves * iters * 10 / (256 CUs * 4 SIMDs)
        const double stages = (double)grid * 4 * iters * 10 / (256.0 * 4);
        printf("mode %d (%s): %.3f ms  -> %.1f ns per stage per SIMD (%.1f cycles at 2.4 GHz)\n", mode,
               mode == 0 ? "64-bit key" : mode == 1 ? "32-bit key + payload" : "32-bit compares, exact (key, payload) order", best,
               best * 1e6 / stages, best * 1e6 / stages * 2.4);
    }
    return 0;
}
