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
// partial[block][0 .. d) = column sums of the block's sample rows, partial[block][d] = largest |component| among them.
// Sample member r is row r * stride of the whole set; x points at row `row0` (the sharded build hands every rank the
// members among its own rows [row0, row0 + ...): r in [r_lo, r_hi)).
__global__ void k_colsum_partial(const float *__restrict__ x, int64_t row0, int64_t r_lo, int64_t r_hi, int64_t stride, int d,
                                 int rows_per_block, double *__restrict__ partial) {
    __shared__ float wmax[4];
    int64_t r0 = r_lo + (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block < r_hi ? r0 + rows_per_block : r_hi;
    float mx = 0.0f;
    for (int j = threadIdx.x; j < d; j += blockDim.x) {
        double s = 0.0;
        for (int64_t r = r0; r < r1; r++) {
            const float v = x[(r * stride - row0) * d + j];
            s += (double)v;
            mx = fmaxf(mx, fabsf(v));  // (a NaN is dropped here; the prep kernel raises the non-finite flag)
        }
        partial[(int64_t)blockIdx.x * (d + 1) + j] = s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (nnd_lane() == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) mx = fmaxf(mx, wmax[w]);
        partial[(int64_t)blockIdx.x * (d + 1) + d] = (double)mx;
    }
}
// one workgroup per column; fixed summation order (tree over a fixed partition) keeps the mean deterministic
// Workgroup j < dp: mean of column j.  Workgroup dp: the largest |component| of the sample -> mean[dp + 2] (the scale of the
// half-precision screening copies is derived from it by k_screen_scale once the means are known).
__global__ __launch_bounds__(256) void k_colsum_final(const double *__restrict__ partial, int nblocks, int d, int dp, int64_t n,
                                                      float *__restrict__ mean) {
    __shared__ double red[256];
    const int j = blockIdx.x;
    double s = 0.0;
    if (j < d)
        for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(int64_t)b * (d + 1) + j];
    if (j == dp)
        for (int b = threadIdx.x; b < nblocks; b += 256) s = fmax(s, partial[(int64_t)b * (d + 1) + d]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = j == dp ? fmax(red[threadIdx.x], red[threadIdx.x + o]) : red[threadIdx.x] + red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (j == dp) mean[dp + 2] = (float)red[0];
        else mean[j] = j < d ? (float)(red[0] / (double)n) : 0.0f;
    }
}
// mean[dp] = scale of the screening copies (common.h nnd_f32_to_h16), mean[dp + 1] = 1 / scale^2 (a screened margin is
// sum (s x_i)(s h_i)).  A power of two that takes the largest prepared component to <= 2^11: a hyperplane is a difference of
// two rows (<= 2^12 after scaling), far from the half-precision overflow at 65504 (a row beyond the sampled maximum by
// more than 16x overflows to inf and is always rechecked exactly: correct, slower).  Cosine rows are unit vectors.
__global__ void k_screen_scale(float *__restrict__ mean, int d, int dp, int metric) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float bound = 1.0f;
    if (metric == 0) {
        float mm = 0.0f;
        for (int j = 0; j < d; j++) mm = fmaxf(mm, fabsf(mean[j]));
        bound = mean[dp + 2] + mm;  // |x_ij - mean_j| <= max |x| + max |mean|
    }
    int e = 0;
    if (bound > 0.0f && bound < 3.0e38f) e = 11 - (int)ceilf(log2f(bound));
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    mean[dp] = exp2f((float)e);
    mean[dp + 1] = exp2f((float)(-2 * e));
}

// ---- one wave per row: pad + centre / normalise + norm ----
// nr2[row] = (nrm[row], |xp_row - bf16(xp_row)| rounded up): the norm and how far the bf16 copy the forest screens with
// is from the row, side by side (one 8-byte load per point in the margin kernels)
__global__ __launch_bounds__(256) void k_prep_rows(const float *__restrict__ x, int64_t row_lo, int64_t row_hi, int d, int dp, int metric,
                                                   const float *__restrict__ mean, float *__restrict__ xp,
                                                   float *__restrict__ nrm, uint16_t *__restrict__ xh,
                                                   float2 *__restrict__ nr2, long long *__restrict__ nonfinite) {
    int lane = nnd_lane();
    int64_t row = row_lo + (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= row_hi) return;
    const float hsc = mean[dp], hinv = 1.0f / hsc;
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
                const uint16_t b = nnd_f32_to_h16(v, hsc);
                xh[row * dp + j] = b;
                const float e = v - nnd_h16_to_f32(b, hinv);
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
                const uint16_t b = nnd_f32_to_h16(v, hsc);
                xh[row * dp + j] = b;
                const float e = v - nnd_h16_to_f32(b, hinv);
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
__global__ __launch_bounds__(256) void k_prep_rows_v4(const float *__restrict__ x, int64_t row_lo, int64_t row_hi, int d, int dp, int metric, int lpr,
                                                      const float *__restrict__ mean, float *__restrict__ xp,
                                                      float *__restrict__ nrm, uint16_t *__restrict__ xh,
                                                      float2 *__restrict__ nr2, long long *__restrict__ nonfinite) {
    const int lane = nnd_lane();
    const int rpw = 64 / lpr, sub = lane / lpr, jl = lane - sub * lpr;
    const int64_t row = row_lo + ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * rpw + sub;
    const bool on = row < row_hi;
    const float hsc = mean[dp], hinv = 1.0f / hsc;
    const int nc = dp >> 2, ncd = d >> 2;  // 16-byte chunks of the padded row / holding data
    const float4 *src = (const float4 *)(x + (on ? row : row_lo) * d);
    float4 *dst = (float4 *)(xp + (on ? row : row_lo) * dp);
    uint2 *dsth = xh ? (uint2 *)(xh + (on ? row : row_lo) * dp) : nullptr;
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
            const uint16_t b0 = nnd_f32_to_h16(v.x, hsc), b1 = nnd_f32_to_h16(v.y, hsc), b2 = nnd_f32_to_h16(v.z, hsc), b3 = nnd_f32_to_h16(v.w, hsc);
            if (on) dsth[c] = make_uint2((uint32_t)b0 | ((uint32_t)b1 << 16), (uint32_t)b2 | ((uint32_t)b3 << 16));
            const float e0 = v.x - nnd_h16_to_f32(b0, hinv), e1 = v.y - nnd_h16_to_f32(b1, hinv);
            const float e2 = v.z - nnd_h16_to_f32(b2, hinv), e3 = v.w - nnd_h16_to_f32(b3, hinv);
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

// The sample rows of the column means are members r = 0 .. n_s - 1, row r * stride of the whole set.  The sharded build
// computes the partial sums of the members among a rank's OWN rows (x_rows points at row `row0`), exchanges them, and
// finishes the means from all of them in rank order (nnd_prep_mean_finish): the double-precision sums make the rounded
// float means the same whatever the split.
void nnd_prep_mean_geometry(int64_t n, int64_t *n_s, int64_t *stride) {
    *n_s = n < NND_MEAN_ROWS ? n : NND_MEAN_ROWS;
    *stride = *n_s > 0 ? n / *n_s : 1;
}
// blocks of sample members [r_lo, r_hi): partial (nblocks, d + 1) doubles at `partial`; returns the number of blocks
int nnd_prep_mean_partial(nnd_ctx *ctx, const float *x_rows, int64_t row0, int64_t r_lo, int64_t r_hi, int64_t stride, double *partial) {
    const int rows_per_block = 128, d = ctx->d;
    const int nblocks = (int)((r_hi - r_lo + rows_per_block - 1) / rows_per_block);
    if (nblocks <= 0) return 0;
    int bt = ((d + 63) / 64) * 64;  // one thread per column: no idle half-blocks at d = 128
    if (bt > 256) bt = 256;
    hipLaunchKernelGGL(k_colsum_partial, dim3(nblocks), dim3(bt), 0, ctx->stream, x_rows, row0, r_lo, r_hi, stride, d, rows_per_block, partial);
    return nblocks;
}
int nnd_prep_mean_finish(nnd_ctx *ctx, const double *partial, int nblocks, int64_t n_s) {
    const int d = ctx->d, dp = ctx->dp;
    if (ctx->p.metric == 0) {
        hipLaunchKernelGGL(k_colsum_final, dim3(dp + 1), dim3(256), 0, ctx->stream, partial, nblocks, d, dp, n_s, ctx->mean);
    } else {
        NND_HIP_CHECK(hipMemsetAsync(ctx->mean, 0, sizeof(float) * (dp + 4), ctx->stream));
    }
    hipLaunchKernelGGL(k_screen_scale, dim3(1), dim3(64), 0, ctx->stream, ctx->mean, d, dp, ctx->p.metric);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}
int nnd_prep_partial_blocks(int64_t members) { return (int)((members + 127) / 128); }
double *nnd_prep_partial_buffer(nnd_ctx *ctx, size_t doubles) {
    if (doubles > ctx->colsum_cap) {  // grow-only scratch: no hipMalloc / hipFree (both synchronise) per build
        if (ctx->colsum_partial) { (void)hipFree(ctx->colsum_partial); ctx->colsum_partial = nullptr; }
        ctx->colsum_cap = 0;
        if (hipMalloc((void **)&ctx->colsum_partial, sizeof(double) * doubles) != hipSuccess) { ctx->set_error("hipMalloc of the column-sum scratch failed"); return nullptr; }
        ctx->colsum_cap = doubles;
    }
    return ctx->colsum_partial;
}
// pad + centre / normalise + norms + screening copy of rows [row_lo, row_hi); x_rows points at row 0 of the WHOLE set (a
// rank that only holds its own rows passes a pointer biased by -row_lo rows)
int nnd_prep_rows(nnd_ctx *ctx, const float *x_all, int64_t row_lo, int64_t row_hi, bool first) {
    const int d = ctx->d, dp = ctx->dp;
    const int64_t rows = row_hi - row_lo;
    long long *flag = ctx->counters_sum + CNT_SCRATCH;  // a spare word of the reduced-counter block
    if (first) NND_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(long long), ctx->stream));
    if (rows <= 0) return 0;
    if ((d & 3) == 0 && ((uintptr_t)x_all & 15) == 0) {
        int lpr = 1;
        while (lpr < (dp >> 2) && lpr < 64) lpr <<= 1;
        const int64_t rows_per_wg = 4 * (64 / lpr);
        hipLaunchKernelGGL(k_prep_rows_v4, dim3((unsigned)((rows + rows_per_wg - 1) / rows_per_wg)), dim3(256), 0, ctx->stream, x_all, row_lo, row_hi, d,
                           dp, ctx->p.metric, lpr, ctx->mean, ctx->xp, ctx->nrm, ctx->xh, ctx->nr2, flag);
    } else {
        hipLaunchKernelGGL(k_prep_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, x_all, row_lo, row_hi, d, dp,
                           ctx->p.metric, ctx->mean, ctx->xp, ctx->nrm, ctx->xh, ctx->nr2, flag);
    }
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

int nnd_launch_prep(nnd_ctx *ctx) {
    const int64_t n = ctx->n;
    int64_t n_s, stride;
    nnd_prep_mean_geometry(n, &n_s, &stride);
    int nblocks = 0;
    double *partial = nullptr;
    if (ctx->p.metric == 0) {
        partial = nnd_prep_partial_buffer(ctx, (size_t)nnd_prep_partial_blocks(n_s) * (ctx->d + 1));
        if (!partial) return 1;
        nblocks = nnd_prep_mean_partial(ctx, ctx->x_orig, 0, 0, n_s, stride, partial);
    }
    if (nnd_prep_mean_finish(ctx, partial, nblocks, n_s)) return 1;
    if (nnd_prep_rows(ctx, ctx->x_orig, 0, n, true)) return 1;
    long long *flag = ctx->counters_sum + CNT_SCRATCH;
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
    ctx->last_updates = -1;
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
