// Probe: __builtin_amdgcn_global_load_lds with per-lane source addresses, wave-uniform LDS base (+ lane * 16), EXEC-masked lanes.
//   hipcc -O3 --offload-arch=gfx950 glds_probe.hip -o glds_probe && ./glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void *src, void *lds_dst_wave_uniform) {
    const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_dst_wave_uniform);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)src,
                                     (__attribute__((address_space(3))) void *)off, 16, 0, 0);
}
__global__ __launch_bounds__(256) void k(const uint32_t *src, const int *rows, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[4][8 * 256];  // per wave: 8 instructions x 1 KB
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = 0; i < 8 * 256; i += 64) buf[w][i + lane] = 0xDEADBEEFu;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int row = rows[(blockIdx.x * 4 + w) * 64 + 8 * i + (lane >> 3)], slot = lane & 7;
        const bool ok = row >= 0;
        if (ok) glds16(src + (size_t)row * 32 + 4 * slot, &buf[w][i * 256]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = 0; i < 8 * 256; i += 64) out[(size_t)(blockIdx.x * 4 + w) * 2048 + i + lane] = buf[w][i + lane];
}
int main() {
    const int nrows = 100000, nblk = 64;
    std::vector<uint32_t> h(nrows * 32);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)i * 2654435761u;
    std::vector<int> rows(nblk * 4 * 64);
    for (size_t i = 0; i < rows.size(); i++) rows[i] = (i % 7 == 3) ? -1 : (int)((i * 7919u) % nrows);
    uint32_t *ds, *dout; int *dr;
    hipMalloc(&ds, h.size() * 4); hipMalloc(&dr, rows.size() * 4); hipMalloc(&dout, (size_t)nblk * 4 * 2048 * 4);
    hipMemcpy(ds, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dr, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(nblk), dim3(256), 0, 0, ds, dr, dout);
    std::vector<uint32_t> o((size_t)nblk * 4 * 2048);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int wv = 0; wv < nblk * 4; wv++)
        for (int r = 0; r < 64; r++)
            for (int c = 0; c < 32; c++) {
                const int row = rows[wv * 64 + r];
                const uint32_t want = row >= 0 ? h[(size_t)row * 32 + c] : 0xDEADBEEFu, got = o[(size_t)wv * 2048 + r * 32 + c];
                if (want != got) { if (bad < 5) printf("wave %d row %d word %d: want %08x got %08x\n", wv, r, c, want, got); bad++; }
            }
    printf("mismatches: %ld (masked lanes must leave LDS untouched)\n", bad);
    return bad != 0;
}
