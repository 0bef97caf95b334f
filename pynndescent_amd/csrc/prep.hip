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
__global__ void k_colsum_partial(const float *__restrict__ x, int64_t n, int d, int rows_per_block,
                                 double *__restrict__ partial) {
    int j = threadIdx.x;  // column
    int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    for (; j < d; j += blockDim.x) {
        double s = 0.0;
        for (int64_t r = r0; r < r1; r++) s += (double)x[r * d + j];
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

int nnd_launch_prep(nnd_ctx *ctx) {
    int64_t n = ctx->n;
    int d = ctx->d, dp = ctx->dp;
    if (ctx->p.metric == 0) {
        int rows_per_block = 128;
        int nblocks = (int)((n + rows_per_block - 1) / rows_per_block);
        const size_t need = (size_t)nblocks * d;
        if (need > ctx->colsum_cap) {  // grow-only scratch: no hipMalloc / hipFree (both synchronise) per build
            if (ctx->colsum_partial) { NND_HIP_CHECK(hipFree(ctx->colsum_partial)); ctx->colsum_partial = nullptr; }
            NND_HIP_CHECK(hipMalloc((void **)&ctx->colsum_partial, sizeof(double) * need));
            ctx->colsum_cap = need;
        }
        double *partial = ctx->colsum_partial;
        int bt = ((d + 63) / 64) * 64;  // one thread per column: no idle half-blocks at d = 128
        if (bt > 256) bt = 256;
        hipLaunchKernelGGL(k_colsum_partial, dim3(nblocks), dim3(bt), 0, ctx->stream, ctx->x_orig, n, d,
                           rows_per_block, partial);
        hipLaunchKernelGGL(k_colsum_final, dim3(dp), dim3(256), 0, ctx->stream, partial, nblocks, d, dp,
                           n, ctx->mean);
    } else {
        NND_HIP_CHECK(hipMemsetAsync(ctx->mean, 0, sizeof(float) * dp, ctx->stream));
    }
    int64_t blocks = (n + 3) / 4;
    long long *flag = ctx->counters_sum + CNT_SCRATCH;  // a spare word of the reduced-counter block
    NND_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(long long), ctx->stream));
    hipLaunchKernelGGL(k_prep_rows, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->x_orig, n, d, dp,
                       ctx->p.metric, ctx->mean, ctx->xp, ctx->nrm, ctx->xh, ctx->nr2, flag);
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
