// finalize.hip -- final row ordering and exact distances; debug Gram tile.
//
// k_finalize replaces deheap_sort (reference utils.py:189-218): rows ascending by distance, empty
// slots (-1, +inf) last.  The k-lists carry Gram-form f32 distances of centred / normalised rows
// (good enough to RANK); the distances handed back are recomputed from the ORIGINAL rows in the
// reference's own formulas (distances.py:63-91 squared difference sum; distances.py:583-630
// log2(sqrt(|x|^2|y|^2)/<x,y>)) with float64 accumulation, then rows are re-sorted by that value.
#include "common.h"
#include "state.h"

__global__ __launch_bounds__(256) void k_finalize(const float *__restrict__ x, int d, int metric, int64_t lo, int64_t n, int k, int ks,
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
    float mine = INFINITY;
    for (int j = 0; j < k; j++) {
        uint32_t ej = __shfl(e, j, 64);
        if (ej == NND_EMPTY_E) continue;  // wave-uniform
        const float *xu = x + (int64_t)(ej & NND_IDX_MASK) * d;
        float val;
        if (metric == 0) {
            double s = 0.0;
            for (int t = lane; t < d; t += 64) {
                double df = (double)xv[t] - (double)xu[t];
                s += df * df;
            }
            val = (float)nnd_wave_sum_f64(s);
        } else {
            double dot = 0.0, nx = 0.0, ny = 0.0;
            for (int t = lane; t < d; t += 64) {
                double a = xv[t], b = xu[t];
                dot += a * b;
                nx += a * a;
                ny += b * b;
            }
            dot = nnd_wave_sum_f64(dot);
            nx = nnd_wave_sum_f64(nx);
            ny = nnd_wave_sum_f64(ny);
            if (nx == 0.0 && ny == 0.0) val = 0.0f;
            else if (nx == 0.0 || ny == 0.0 || dot <= 0.0) val = NND_FLT_MAX;
            else {
                double r = log2(sqrt(nx * ny) / dot);
                val = r > 0.0 ? (float)r : 0.0f;
            }
        }
        if (lane == j) mine = val;
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

int nnd_launch_finalize(nnd_ctx *ctx, int32_t *out_idx_dev, float *out_dist_dev) {
    unsigned grid = (unsigned)((ctx->own_hi - ctx->own_lo + 3) / 4);
    grid = (grid + 7u) & ~7u;  // whole multiples of the XCD count
    hipLaunchKernelGGL(k_finalize, dim3(grid), dim3(256), 0, ctx->stream, ctx->x_orig, ctx->d, ctx->p.metric, ctx->own_lo,
                       ctx->own_hi, ctx->k, ctx->ks, ctx->knn_e, nnd_vertex_order(ctx), out_idx_dev, out_dist_dev);
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
