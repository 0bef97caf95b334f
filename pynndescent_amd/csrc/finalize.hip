// finalize.hip -- final row ordering and exact distances; debug Gram tile.
//
// k_finalize replaces deheap_sort (reference utils.py:189-218): rows ascending by distance, empty
// slots (-1, +inf) last.  The k-lists carry Gram-form f32 distances of centred / normalised rows
// (good enough to RANK); the distances handed back are recomputed from the ORIGINAL rows in the
// reference's own formulas (distances.py:63-91 squared difference sum; distances.py:583-630
// log2(sqrt(|x|^2|y|^2)/<x,y>)) with float64 accumulation, then rows are re-sorted by that value.
#include "common.h"
#include "state.h"

// sum over the 16 lanes of an aligned 16-lane group
__device__ __forceinline__ double fin_group16_sum_f64(double v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wave per row.  The 64 lanes work as 4 groups of 16: group g takes neighbours g, 4+g, 8+g, ... so four distances
// are accumulated side by side and up to 16 neighbour rows are in flight per wave (the gather is latency bound when
// the rows are fetched one after another).
// METRIC is a template parameter: the euclidean instance does not carry the cosine accumulators (146 -> far fewer
// registers, i.e. more waves per SIMD for what is a gather-latency-bound kernel).
template <int METRIC>
__global__ __launch_bounds__(256, METRIC == 0 ? 6 : 4) void k_finalize(const float *__restrict__ x, int d, int64_t lo, int64_t n, int k, int ks,
                                                  const uint32_t *__restrict__ knn_e, const int32_t *__restrict__ order,
                                                  int32_t *__restrict__ out_idx, float *__restrict__ out_dist) {
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    // rows are visited in `order` (spatially coherent, see nnd_vertex_order), one contiguous eighth per XCD, so the
    // neighbour rows of concurrently running waves overlap in L2
    int64_t b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int64_t g = lo + b * 4 + w;  // owned rows [lo, n); output row index is v - lo
    if (g >= n) return;
    const int64_t v = order ? (int64_t)order[g] : g;
    uint32_t e = lane < k ? knn_e[v * ks + lane] : NND_EMPTY_E;
    const float *xv = x + v * d;
    const int grp = lane >> 4, l16 = lane & 15;
    const bool vec = (d & 3) == 0;  // rows 16-byte aligned
    constexpr int metric = METRIC;
    float mine = INFINITY;
    const int nsteps = (k + 3) >> 2;
    for (int s0 = 0; s0 < nsteps; s0 += 4) {
        const float *xu[4];
        bool on[4];
        double s[4], dot[4], nx[4], ny[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = 4 * (s0 + u) + grp;
            const uint32_t ej = __shfl(e, j & 63, 64);
            on[u] = j < k && ej != NND_EMPTY_E;
            xu[u] = x + (int64_t)(on[u] ? (ej & NND_IDX_MASK) : 0) * d;
            s[u] = dot[u] = nx[u] = ny[u] = 0.0;
        }
        if (vec) {
            for (int t = 4 * l16; t < d; t += 64) {
                const float4 a = *(const float4 *)(xv + t);
                float4 q[4];
#pragma unroll
                for (int u = 0; u < 4; u++) q[u] = *(const float4 *)(xu[u] + t);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const double a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w;
                    const double b0 = q[u].x, b1 = q[u].y, b2 = q[u].z, b3 = q[u].w;
                    if (metric == 0) {
                        s[u] += (a0 - b0) * (a0 - b0) + (a1 - b1) * (a1 - b1) + (a2 - b2) * (a2 - b2) + (a3 - b3) * (a3 - b3);
                    } else {
                        dot[u] += a0 * b0 + a1 * b1 + a2 * b2 + a3 * b3;
                        nx[u] += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
                        ny[u] += b0 * b0 + b1 * b1 + b2 * b2 + b3 * b3;
                    }
                }
            }
        } else {
            for (int t = l16; t < d; t += 16) {
                const double a0 = xv[t];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const double b0 = xu[u][t];
                    if (metric == 0) s[u] += (a0 - b0) * (a0 - b0);
                    else {
                        dot[u] += a0 * b0;
                        nx[u] += a0 * a0;
                        ny[u] += b0 * b0;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float val;
            if (metric == 0) {
                val = (float)fin_group16_sum_f64(s[u]);
            } else {
                const double dt = fin_group16_sum_f64(dot[u]), ax = fin_group16_sum_f64(nx[u]), ay = fin_group16_sum_f64(ny[u]);
                if (ax == 0.0 && ay == 0.0) val = 0.0f;
                else if (ax == 0.0 || ay == 0.0 || dt <= 0.0) val = NND_FLT_MAX;
                else {
                    const double r = log2(sqrt(ax * ay) / dt);
                    val = r > 0.0 ? (float)r : 0.0f;
                }
            }
            if (!on[u]) val = INFINITY;
            // neighbour j = 4 * (s0 + u) + group: lane j takes it from (any lane of) group j & 3
            const float got = __shfl(val, 16 * (lane & 3), 64);
            if ((lane >> 2) == s0 + u) mine = got;
        }
    }
    // rank by (distance, index); empty entries are (+inf, 0x7FFFFFFF) and land at the tail
    const uint32_t myidx = e == NND_EMPTY_E ? NND_IDX_MASK : (e & NND_IDX_MASK);
    const uint64_t mykey = ((uint64_t)__float_as_uint(mine) << 32) | myidx;
    int r = 0;
    for (int j = 0; j < k; j++) {
        uint64_t kj = ((uint64_t)__float_as_uint(__shfl(mine, j, 64)) << 32) | __shfl(myidx, j, 64);
        r += (kj < mykey || (kj == mykey && j < lane)) ? 1 : 0;
    }
    if (lane < k) {
        out_idx[(v - lo) * k + r] = e == NND_EMPTY_E ? -1 : (int32_t)(e & NND_IDX_MASK);
        out_dist[(v - lo) * k + r] = mine;
    }
}

// 64 < k <= NND_WIDE_K: one wave per row, ids / exact distances through LDS, four neighbours at a time (16 lanes each), ranks by
// counting over the LDS copy.  Same arithmetic as k_finalize (float64 accumulation of the reference's formulas).
__global__ __launch_bounds__(256) void k_finalize_wide(const float *__restrict__ x, int d, int64_t lo, int64_t n, int k, int ks, int metric,
                                                       const uint32_t *__restrict__ knn_e, int32_t *__restrict__ out_idx,
                                                       float *__restrict__ out_dist) {
    __shared__ uint32_t sid[4][NND_WIDE_K];
    __shared__ float sdist[4][NND_WIDE_K];
    const int lane = nnd_lane(), w = threadIdx.x >> 6, grp = lane >> 4, l16 = lane & 15;
    const int64_t v = lo + (int64_t)blockIdx.x * 4 + w;
    if (v >= n) return;
    for (int j = lane; j < k; j += 64) sid[w][j] = knn_e[v * ks + j];
    nnd_wave_lds_sync();
    const float *xv = x + v * d;
    for (int j0 = 0; j0 < k; j0 += 4) {
        const int j = j0 + grp;
        const uint32_t ej = j < k ? sid[w][j] : NND_EMPTY_E;
        const bool on = ej != NND_EMPTY_E;
        const float *xu = x + (int64_t)(on ? (ej & NND_IDX_MASK) : 0) * d;
        double s = 0.0, dot = 0.0, nx = 0.0, ny = 0.0;
        for (int t = l16; t < d; t += 16) {
            const double a0 = xv[t], b0 = xu[t];
            if (metric == 0) s += (a0 - b0) * (a0 - b0);
            else {
                dot += a0 * b0;
                nx += a0 * a0;
                ny += b0 * b0;
            }
        }
        float val;
        if (metric == 0) {
            val = (float)fin_group16_sum_f64(s);
        } else {
            const double dt = fin_group16_sum_f64(dot), ax = fin_group16_sum_f64(nx), ay = fin_group16_sum_f64(ny);
            if (ax == 0.0 && ay == 0.0) val = 0.0f;
            else if (ax == 0.0 || ay == 0.0 || dt <= 0.0) val = NND_FLT_MAX;
            else {
                const double r = log2(sqrt(ax * ay) / dt);
                val = r > 0.0 ? (float)r : 0.0f;
            }
        }
        if (!on) val = INFINITY;
        if (l16 == 0 && j < k) sdist[w][j] = val;
    }
    nnd_wave_lds_sync();
    for (int j = lane; j < k; j += 64) {
        const uint32_t e = sid[w][j];
        const uint32_t myidx = e == NND_EMPTY_E ? NND_IDX_MASK : (e & NND_IDX_MASK);
        const uint64_t mykey = ((uint64_t)__float_as_uint(sdist[w][j]) << 32) | myidx;
        int r = 0;
        for (int q = 0; q < k; q++) {
            const uint32_t eq = sid[w][q];
            const uint64_t kq = ((uint64_t)__float_as_uint(sdist[w][q]) << 32) | (eq == NND_EMPTY_E ? NND_IDX_MASK : (eq & NND_IDX_MASK));
            r += (kq < mykey || (kq == mykey && q < j)) ? 1 : 0;
        }
        out_idx[(v - lo) * k + r] = e == NND_EMPTY_E ? -1 : (int32_t)(e & NND_IDX_MASK);
        out_dist[(v - lo) * k + r] = sdist[w][j];
    }
}

int nnd_launch_finalize(nnd_ctx *ctx, int32_t *out_idx_dev, float *out_dist_dev) {
    unsigned grid = (unsigned)((ctx->own_hi - ctx->own_lo + 3) / 4);
    if (ctx->k > 64) {
        hipLaunchKernelGGL(k_finalize_wide, dim3(grid), dim3(256), 0, ctx->stream, ctx->x_orig, ctx->d, ctx->own_lo, ctx->own_hi, ctx->k, ctx->ks,
                           ctx->p.metric, ctx->knn_e, out_idx_dev, out_dist_dev);
        NND_HIP_CHECK(hipGetLastError());
        return 0;
    }
    grid = (grid + 7u) & ~7u;  // whole multiples of the XCD count
    if (ctx->p.metric == 0)
        hipLaunchKernelGGL(k_finalize<0>, dim3(grid), dim3(256), 0, ctx->stream, ctx->x_orig, ctx->d, ctx->own_lo, ctx->own_hi,
                           ctx->k, ctx->ks, ctx->knn_e, nnd_vertex_order(ctx), out_idx_dev, out_dist_dev);
    else
        hipLaunchKernelGGL(k_finalize<1>, dim3(grid), dim3(256), 0, ctx->stream, ctx->x_orig, ctx->d, ctx->own_lo, ctx->own_hi,
                           ctx->k, ctx->ks, ctx->knn_e, nnd_vertex_order(ctx), out_idx_dev, out_dist_dev);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- debug / parity: the MFMA Gram tile on arbitrary row lists (tests/test_gpu_kernels.py) ----
// one wave per 16x16 output tile; operands come straight from global memory with the same K mapping
// as gram.h (chunk c of a row feeds lane group c & 3).
__global__ __launch_bounds__(64) void k_pairwise(const float *__restrict__ xp, int dp, const float *__restrict__ nrm,
                                                 int metric, const int32_t *__restrict__ rows_a, int na,
                                                 const int32_t *__restrict__ rows_b, int nb, float *__restrict__ out) {
    const int lane = nnd_lane(), r16 = lane & 15, g = lane >> 4;
    const int ta = blockIdx.y, tb = blockIdx.x;
    int ia = ta * 16 + r16, ib = tb * 16 + r16;
    int ida = ia < na ? rows_a[ia] : -1, idb = ib < nb ? rows_b[ib] : -1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < (dp >> 4); t++) {
        int c = 4 * t + g;
        float4 a = ida >= 0 ? *(const float4 *)(xp + (int64_t)ida * dp + 4 * c) : make_float4(0, 0, 0, 0);
        float4 b = idb >= 0 ? *(const float4 *)(xp + (int64_t)idb * dp + 4 * c) : make_float4(0, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    }
    int col = tb * 16 + r16;
    float nbv = idb >= 0 ? nrm[idb] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int row = ta * 16 + 4 * g + r;
        int idr = row < na ? rows_a[row] : -1;
        if (row < na && col < nb) out[(int64_t)row * nb + col] = nnd_gram_to_dist(metric, acc[r], idr >= 0 ? nrm[idr] : 0.0f, nbv);
    }
}

int nnd_launch_pairwise(nnd_ctx *ctx, const int32_t *rows_a_dev, int na, const int32_t *rows_b_dev, int nb, float *out_dev) {
    dim3 grid((nb + 15) / 16, (na + 15) / 16);
    hipLaunchKernelGGL(k_pairwise, grid, dim3(64), 0, ctx->stream, ctx->xp, ctx->dp, ctx->nrm, ctx->p.metric, rows_a_dev, na,
                       rows_b_dev, nb, out_dev);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}
