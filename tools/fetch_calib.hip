// fetch_calib.hip -- calibrate rocprofv3's FETCH_SIZE for the access patterns of this library's kernels
// (/opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE counts 128-byte requests of wide
// coalesced streaming reads at 64 B -- other patterns are uncalibrated, so calibrate on a known byte count).
//
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/calib -- /tmp/fetch_calib
//
// Each kernel reads every byte of a 2 GiB table (8x the 256 MiB Infinity Cache) exactly once:
//   k_calib_stream : 16 B per lane, consecutive lanes consecutive addresses (prep, finalize streams, k_route's row loads)
//   k_calib_gather : the local join's operand gather -- a wave takes 32 RANDOM 512-byte rows; lane (r16, g) loads the
//                    16-byte chunks 4t + g of row r16 (new tile) and of row 16 + r16 (old tile), t = 0..7
//   k_calib_quad   : the forest's margin gathers -- a quad per random 256-byte (bf16) row, lane sub loads chunks sub, sub + 4, ...
// factor = bytes read / (FETCH_SIZE * 1024).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

__global__ void k_calib_stream(const uint4 *__restrict__ t, int64_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 v = t[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void k_calib_gather(const float *__restrict__ x, const int32_t *__restrict__ order, int64_t n_rows, uint32_t *sink) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (int64_t b = wave * 32; b + 32 <= n_rows; b += n_waves * 32) {
        const float *pa = x + (int64_t)order[b + r16] * 128 + 4 * g;
        const float *pb = x + (int64_t)order[b + 16 + r16] * 128 + 4 * g;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const uint4 a = *(const uint4 *)(pa + 16 * t), c = *(const uint4 *)(pb + 16 * t);
            acc ^= a.x ^ a.w ^ c.y ^ c.z;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void k_calib_quad(const uint16_t *__restrict__ xh, const int32_t *__restrict__ order, int64_t n_rows, uint32_t *sink) {
    const int sub = threadIdx.x & 3;
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int64_t n_quads = ((int64_t)gridDim.x * blockDim.x) >> 2;
    uint32_t acc = 0;
    for (int64_t i = q; i < n_rows; i += n_quads) {
        const uint4 *r = (const uint4 *)(xh + (int64_t)order[i] * 128);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint4 v = r[sub + 4 * j];
            acc ^= v.x ^ v.w;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const int64_t n_rows = 4 * 1024 * 1024;  // x 512 B = 2 GiB (f32 rows); the bf16 table is the first GiB
    float *x = nullptr;
    int32_t *order = nullptr;
    uint32_t *sink = nullptr;
    CHECK(hipMalloc((void **)&x, (size_t)n_rows * 512));
    CHECK(hipMalloc((void **)&order, sizeof(int32_t) * (size_t)n_rows));
    CHECK(hipMalloc((void **)&sink, 4));
    CHECK(hipMemset(x, 1, (size_t)n_rows * 512));
    int32_t *h = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_rows);
    for (int64_t i = 0; i < n_rows; i++) h[i] = (int32_t)i;
    uint64_t s = 88172645463325252ull;
    for (int64_t i = n_rows - 1; i > 0; i--) {  // Fisher-Yates with xorshift64
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int64_t j = (int64_t)(s % (uint64_t)(i + 1));
        const int32_t t = h[i]; h[i] = h[j]; h[j] = t;
    }
    CHECK(hipMemcpy(order, h, sizeof(int32_t) * (size_t)n_rows, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_calib_stream, dim3(4096), dim3(256), 0, 0, (const uint4 *)x, n_rows * 32, sink);
        hipLaunchKernelGGL(k_calib_gather, dim3(2048), dim3(256), 0, 0, x, order, n_rows, sink);
        hipLaunchKernelGGL(k_calib_quad, dim3(4096), dim3(256), 0, 0, (const uint16_t *)x, order, n_rows, sink);
    }
    CHECK(hipDeviceSynchronize());
    printf("bytes_per_launch stream %lld gather %lld quad %lld\n", (long long)(n_rows * 512), (long long)(n_rows * 512), (long long)(n_rows * 256));
    return 0;
}
