// prep.hip -- one-time preparation of the point set in HBM, k-list reset, counters.
//
// Layout decisions (DESIGN.md "Data layout"):
//   xp  (n, dp) float32, dp = dim rounded up to 32 floats so every row is a whole number of
//       128-byte lines; euclidean rows are centred on the column mean (distances are translation
//       invariant; centring shrinks |x|^2 and with it the cancellation error of the Gram form),
//       cosine rows are L2-normalised once (the reference recomputes both norms in every
//       distance call, distances.py:617-620); zero rows stay zero and are flagged in nrm.
//   nrm (n) float32: |xp_i|^2 for euclidean; 1 (non-zero row) / 0 (zero row) for cosine.
#include "common.h"
#include "state.h"

// ---- column means, deterministic two-stage reduction (no float atomics) ----
// The centre only has to be NEAR the data (any translation leaves the distances alone): it is the mean of at most
// NND_MEAN_ROWS rows taken at a fixed stride -- a full pass over x for the exact mean was a quarter of the prep time.
#define NND_MEAN_ROWS 65536
__global__ void k_colsum_partial(const float *__restrict__ x, int64_t n_s, int64_t stride, int d, int rows_per_block,
                                 double *__restrict__ partial) {
    int j = threadIdx.x;  // column
    int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block < n_s ? r0 + rows_per_block : n_s;
    for (; j < d; j += blockDim.x) {
        double s = 0.0;
        for (int64_t r = r0; r < r1; r++) s += (double)x[r * stride * d + j];
        partial[(int64_t)blockIdx.x * d + j] = s;
    }
}
// one workgroup per column; fixed summation order (tree over a fixed partition) keeps the mean deterministic
__global__ __launch_bounds__(256) void k_colsum_final(const double *__restrict__ partial, int nblocks, int d, int dp, int64_t n,
                                                      float *__restrict__ mean) {
    __shared__ double red[256];
    const int j = blockIdx.x;
    double s = 0.0;
    if (j < d)
        for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(int64_t)b * d + j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean[j] = j < d ? (float)(red[0] / (double)n) : 0.0f;
}

// ---- one wave per row: pad + centre / normalise + norm ----
// nr2[row] = (nrm[row], |xp_row - bf16(xp_row)| rounded up): the norm and how far the bf16 copy the forest screens with
// is from the row, side by side (one 8-byte load per point in the margin kernels)
__global__ __launch_bounds__(256) void k_prep_rows(const float *__restrict__ x, int64_t n, int d, int dp, int metric,
                                                   const float *__restrict__ mean, float *__restrict__ xp,
                                                   float *__restrict__ nrm, uint16_t *__restrict__ xh,
                                                   float2 *__restrict__ nr2, long long *__restrict__ nonfinite) {
    int lane = nnd_lane();
    int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *src = x + row * d;
    float *dst = xp + row * dp;
    bool bad = false;  // a NaN / inf in the input: the host raises what check_array raises in the reference (pynndescent_.py:1054)
    if (metric == 0) {
        float s = 0.0f, r2 = 0.0f;
        for (int j = lane; j < dp; j += 64) {
            const float raw = j < d ? src[j] : 0.0f;
            bad |= !isfinite(raw);
            float v = j < d ? raw - mean[j] : 0.0f;
            dst[j] = v;
            if (xh) {
                const uint16_t b = nnd_f32_to_bf16(v);
                xh[row * dp + j] = b;
                const float e = v - __uint_as_float((uint32_t)b << 16);
                r2 += e * e;
            }
            s += v * v;
        }
        s = nnd_wave_sum_f32(s);
        if (xh) r2 = nnd_wave_sum_f32(r2);
        if (lane == 0) {
            nrm[row] = s;
            if (xh) nr2[row] = make_float2(s, sqrtf(r2) * 1.000001f);
        }
    } else {
        float s = 0.0f;
        for (int j = lane; j < d; j += 64) {
            float v = src[j];
            bad |= !isfinite(v);
            s += v * v;
        }
        s = nnd_wave_sum_f32(s);
        float inv = s > 0.0f ? 1.0f / sqrtf(s) : 0.0f;
        float r2 = 0.0f;
        for (int j = lane; j < dp; j += 64) {
            const float v = j < d ? src[j] * inv : 0.0f;
            dst[j] = v;
            if (xh) {
                const uint16_t b = nnd_f32_to_bf16(v);
                xh[row * dp + j] = b;
                const float e = v - __uint_as_float((uint32_t)b << 16);
                r2 += e * e;
            }
        }
        if (xh) r2 = nnd_wave_sum_f32(r2);
        if (lane == 0) {
            nrm[row] = s > 0.0f ? 1.0f : 0.0f;
            if (xh) nr2[row] = make_float2(s > 0.0f ? 1.0f : 0.0f, sqrtf(r2) * 1.000001f);
        }
    }
    if (__ballot(bad) && lane == 0) atomicOr((unsigned long long *)nonfinite, 1ull);
}

// ---- the same, 16 bytes per lane (d a multiple of 4, rows 16-byte aligned): LPR = dp/4 rounded up to a power of two
// lanes per row (32 at d = 128: two rows per wave), whole rows in flight per load instruction ----
__global__ __launch_bounds__(256) void k_prep_rows_v4(const float *__restrict__ x, int64_t n, int d, int dp, int metric, int lpr,
                                                      const float *__restrict__ mean, float *__restrict__ xp,
                                                      float *__restrict__ nrm, uint16_t *__restrict__ xh,
                                                      float2 *__restrict__ nr2, long long *__restrict__ nonfinite) {
    const int lane = nnd_lane();
    const int rpw = 64 / lpr, sub = lane / lpr, jl = lane - sub * lpr;
    const int64_t row = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * rpw + sub;
    const bool on = row < n;
    const int nc = dp >> 2, ncd = d >> 2;  // 16-byte chunks of the padded row / holding data
    const float4 *src = (const float4 *)(x + (on ? row : 0) * d);
    float4 *dst = (float4 *)(xp + (on ? row : 0) * dp);
    uint2 *dsth = xh ? (uint2 *)(xh + (on ? row : 0) * dp) : nullptr;
    bool bad = false;
    float s = 0.0f, r2 = 0.0f, inv = 1.0f;
    if (metric != 0) {  // cosine: the norm first (the row stays in L1 / L2 for the second pass)
        for (int c = jl; c < ncd; c += lpr) {
            const float4 v = on ? src[c] : make_float4(0, 0, 0, 0);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int o = lpr >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        inv = s > 0.0f ? 1.0f / sqrtf(s) : 0.0f;
    }
    float acc = 0.0f;
    for (int c = jl; c < nc; c += lpr) {
        float4 v = (on && c < ncd) ? src[c] : make_float4(0, 0, 0, 0);
        bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
        if (metric == 0) {
            if (c < ncd) {
                const float4 m = ((const float4 *)mean)[c];
                v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w;
            }
        } else {
            v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        }
        if (on) dst[c] = v;
        if (xh) {
            const uint16_t b0 = nnd_f32_to_bf16(v.x), b1 = nnd_f32_to_bf16(v.y), b2 = nnd_f32_to_bf16(v.z), b3 = nnd_f32_to_bf16(v.w);
            if (on) dsth[c] = make_uint2((uint32_t)b0 | ((uint32_t)b1 << 16), (uint32_t)b2 | ((uint32_t)b3 << 16));
            const float e0 = v.x - __uint_as_float((uint32_t)b0 << 16), e1 = v.y - __uint_as_float((uint32_t)b1 << 16);
            const float e2 = v.z - __uint_as_float((uint32_t)b2 << 16), e3 = v.w - __uint_as_float((uint32_t)b3 << 16);
            r2 += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
        }
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) {
        acc += __shfl_xor(acc, o, 64);
        r2 += __shfl_xor(r2, o, 64);
    }
    if (on && jl == 0) {
        const float nv = metric == 0 ? acc : (s > 0.0f ? 1.0f : 0.0f);
        nrm[row] = nv;
        if (xh) nr2[row] = make_float2(nv, sqrtf(r2) * 1.000001f);
    }
    if (__ballot(bad) && lane == 0) atomicOr((unsigned long long *)nonfinite, 1ull);
}

int nnd_launch_prep(nnd_ctx *ctx) {
    int64_t n = ctx->n;
    int d = ctx->d, dp = ctx->dp;
    if (ctx->p.metric == 0) {
        int rows_per_block = 128;
        const int64_t n_s = n < NND_MEAN_ROWS ? n : NND_MEAN_ROWS, stride = n_s > 0 ? n / n_s : 1;
        int nblocks = (int)((n_s + rows_per_block - 1) / rows_per_block);
        const size_t need = (size_t)nblocks * d;
        if (need > ctx->colsum_cap) {  // grow-only scratch: no hipMalloc / hipFree (both synchronise) per build
            if (ctx->colsum_partial) { NND_HIP_CHECK(hipFree(ctx->colsum_partial)); ctx->colsum_partial = nullptr; }
            NND_HIP_CHECK(hipMalloc((void **)&ctx->colsum_partial, sizeof(double) * need));
            ctx->colsum_cap = need;
        }
        double *partial = ctx->colsum_partial;
        int bt = ((d + 63) / 64) * 64;  // one thread per column: no idle half-blocks at d = 128
        if (bt > 256) bt = 256;
        hipLaunchKernelGGL(k_colsum_partial, dim3(nblocks), dim3(bt), 0, ctx->stream, ctx->x_orig, n_s, stride, d,
                           rows_per_block, partial);
        hipLaunchKernelGGL(k_colsum_final, dim3(dp), dim3(256), 0, ctx->stream, partial, nblocks, d, dp,
                           n_s, ctx->mean);
    } else {
        NND_HIP_CHECK(hipMemsetAsync(ctx->mean, 0, sizeof(float) * dp, ctx->stream));
    }
    long long *flag = ctx->counters_sum + CNT_SCRATCH;  // a spare word of the reduced-counter block
    NND_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(long long), ctx->stream));
    if ((d & 3) == 0 && ((uintptr_t)ctx->x_orig & 15) == 0) {
        int lpr = 1;
        while (lpr < (dp >> 2) && lpr < 64) lpr <<= 1;
        const int64_t rows_per_wg = 4 * (64 / lpr);
        hipLaunchKernelGGL(k_prep_rows_v4, dim3((unsigned)((n + rows_per_wg - 1) / rows_per_wg)), dim3(256), 0, ctx->stream, ctx->x_orig, n, d,
                           dp, ctx->p.metric, lpr, ctx->mean, ctx->xp, ctx->nrm, ctx->xh, ctx->nr2, flag);
    } else {
        hipLaunchKernelGGL(k_prep_rows, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, ctx->x_orig, n, d, dp,
                           ctx->p.metric, ctx->mean, ctx->xp, ctx->nrm, ctx->xh, ctx->nr2, flag);
    }
    NND_HIP_CHECK(hipGetLastError());
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 63, flag, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));  // read by nnd_data_nonfinite
    return 0;
}

// ---- make_heap (reference utils.py:130-158): all slots (-1, +inf, flag 0) ----
__global__ void k_reset_graph(uint32_t *__restrict__ e, float *__restrict__ dd, int64_t total, float *__restrict__ th,
                              int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        e[i] = NND_EMPTY_E;
        dd[i] = INFINITY;
    }
    if (i < n) th[i] = INFINITY;
}

int nnd_launch_reset_graph(nnd_ctx *ctx) {
    ctx->all_new = true;
    int64_t total = ctx->n * ctx->ks;
    hipLaunchKernelGGL(k_reset_graph, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, ctx->knn_e,
                       ctx->knn_d, total, ctx->th, ctx->n);
    // k_merge and k_sample_select re-arm every slot they consume, so after a complete single-GPU iteration both slot
    // tables are EMPTY again; the flags are cleared by every kernel launch that writes slots
    if (!ctx->pbuf_clean) {
        NND_HIP_CHECK(hipMemsetAsync(ctx->pbuf + (size_t)ctx->slim_row0() * ctx->pcap, 0xFF, sizeof(uint64_t) * (size_t)ctx->slim_rows() * ctx->pcap, ctx->stream));
        if (ctx->pbuf_r) NND_HIP_CHECK(hipMemsetAsync(ctx->pbuf_r, 0xFF, sizeof(uint64_t) * (size_t)ctx->n * ctx->pcap_r, ctx->stream));
    }
    if (!ctx->rbuf_clean)
        NND_HIP_CHECK(hipMemsetAsync(ctx->rbuf + (size_t)ctx->slim_row0() * 2 * ctx->rcap, 0xFF, sizeof(uint32_t) * (size_t)ctx->slim_rows() * 2 * ctx->rcap, ctx->stream));
    ctx->pbuf_clean = ctx->rbuf_clean = true;
    NND_HIP_CHECK(hipGetLastError());
    ctx->iter = 0;
    return 0;
}

int nnd_zero_counters(nnd_ctx *ctx) {
    NND_HIP_CHECK(hipMemsetAsync(ctx->counters, 0, sizeof(long long) * CNT_COUNT * NND_CNT_STRIPES, ctx->stream));
    return 0;
}
// stripes -> one value per counter, on the device (the host then reads CNT_COUNT words, not CNT_COUNT * 512)
__global__ __launch_bounds__(256) void k_counters_reduce(const long long *__restrict__ counters, long long *__restrict__ out) {
    __shared__ long long red[256];
    for (int c = 0; c < CNT_COUNT; c++) {
        long long s = 0;
        for (int i = threadIdx.x; i < NND_CNT_STRIPES; i += 256) s += counters[(size_t)i * CNT_COUNT + c];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[c] = red[0];
        __syncthreads();
    }
}
int nnd_read_counters(nnd_ctx *ctx) {
    hipLaunchKernelGGL(k_counters_reduce, dim3(1), dim3(256), 0, ctx->stream, ctx->counters, ctx->counters_sum);
    NND_HIP_CHECK(hipGetLastError());
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin, ctx->counters_sum, sizeof(long long) * CNT_COUNT, hipMemcpyDeviceToHost, ctx->stream));
    NND_HIP_CHECK(nnd_sync_spin(ctx));
    for (int c = 0; c < CNT_COUNT; c++) ctx->h_counters[c] = ctx->h_pin[c];
    return 0;
}
