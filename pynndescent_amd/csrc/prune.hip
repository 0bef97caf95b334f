// prune.hip -- search-graph pruning pass (BASELINE config 5: "+ graph diversification/prune pass").
//
// Replaces, for the standard diversify method at diversify_prob = 1 (the defaults):
//   diversify             reference pynndescent_.py:369-403   -> k_diversify_rows
//   diversify_csr         reference pynndescent_.py:549-588   -> k_diversify_csr
//   degree_prune_internal reference pynndescent_.py:728-738   -> k_degree_prune
// The COO/CSR conversions, the transpose, the element-wise maximum and the binarisation between them are
// scipy calls in the reference (pynndescent_.py:1509-1611) and stay host glue (pynndescent_amd/search_graph.py).
//
// All three are row-parallel: one wave per row, the row's entries in lanes, pair distances by wave-cooperative
// dot products over the prepared rows (xp: centred for euclidean -- differences are unchanged -- or
// L2-normalised for cosine), decisions broadcast with readlane.  The pruning rule is order dependent inside a
// row (an entry is tested against the entries KEPT so far), so a row is walked sequentially exactly like the
// reference does; rows are independent.
#include "common.h"
#include "state.h"

#define PRUNE_EPS 1.1920929e-07f  // np.finfo(np.float32).eps (pynndescent_.py:65)

// alt-space distance between prepared rows a and b (all 64 lanes participate)
__device__ __forceinline__ float prune_pair_dist(const float *__restrict__ xp, int dp, const float *__restrict__ nrm, int metric,
                                                 int64_t a, int64_t b) {
    const float *xa = xp + a * dp, *xb = xp + b * dp;
    float s = 0.0f;
    for (int j = nnd_lane(); j < dp; j += 64) {
        const float p = xa[j], q = xb[j];
        s += metric == 0 ? (p - q) * (p - q) : p * q;
    }
    s = nnd_wave_sum_f32(s);
    if (metric == 0) return nnd_clamp_dist(s);
    return nnd_gram_to_dist(1, s, nrm[a], nrm[b]);
}

// pynndescent_.py:369-403.  rows: (n,k) ascending; pruned slots become (-1, +inf); kept entries stay in order.
__global__ __launch_bounds__(256) void k_diversify_rows(const float *__restrict__ xp, int dp, const float *__restrict__ nrm,
                                                        int metric, int64_t n, int k, int32_t *__restrict__ idx,
                                                        float *__restrict__ dist) {
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + w;
    if (i >= n) return;
    const int32_t my_idx = lane < k ? idx[i * k + lane] : -1;
    const float my_d = lane < k ? dist[i * k + lane] : INFINITY;
    unsigned long long kept = 1ull;  // position 0 is always kept (pynndescent_.py:374-375)
    for (int j = 1; j < k; j++) {
        const int32_t idj = __builtin_amdgcn_readlane(my_idx, j);
        if (idj < 0) break;  // pynndescent_.py:377-378
        const float dj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_d), j));
        bool flag = true;
        unsigned long long m = kept;
        while (m) {
            const int c = __builtin_ctzll(m);
            m &= m - 1;
            const float dc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_d), c));
            if (dc > PRUNE_EPS) {
                const int32_t idc = __builtin_amdgcn_readlane(my_idx, c);
                const float d = prune_pair_dist(xp, dp, nrm, metric, idj, idc);
                if (d < dj) {  // pynndescent_.py:386-389 (prune_probability = 1)
                    flag = false;
                    break;
                }
            }
        }
        if (flag) kept |= 1ull << j;
    }
    // slot s of the output takes the s-th kept entry (kept entries stay in order); the tail is (-1, +inf)
    const int nk = __popcll(kept);
    unsigned long long m = kept;
    for (int t = 0; t < lane && m; t++) m &= m - 1;  // drop the `lane` lowest set bits
    const int src = (lane < nk) ? __builtin_ctzll(m) : 0;
    const int32_t out_i = __shfl(my_idx, src, 64);
    const float out_d = __shfl(my_d, src, 64);
    if (lane < k) {
        idx[i * k + lane] = lane < nk ? out_i : -1;
        dist[i * k + lane] = lane < nk ? out_d : INFINITY;
    }
}

// pynndescent_.py:549-588 on CSR rows of length <= 64 (rows of a diversified k-NN graph have <= k entries).
// Entries are walked in ascending weight order (ties by position); NOTE the reference takes the comparison POINT
// from storage position `k` (`current_indices[k]`, line 577) while it takes weight and retained-flag from the
// weight order (`l = order[k]`); that is reproduced here.  Pruned entries get weight 0.
__global__ __launch_bounds__(256) void k_diversify_csr(const float *__restrict__ xp, int dp, const float *__restrict__ nrm,
                                                       int metric, int64_t n_rows, const int32_t *__restrict__ indptr,
                                                       const int32_t *__restrict__ indices, float *__restrict__ data,
                                                       int *__restrict__ too_long) {
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + w;
    if (i >= n_rows) return;
    const int a = indptr[i], len = indptr[i + 1] - a;
    if (len <= 1) return;
    if (len > 64) {
        if (lane == 0) atomicAdd(too_long, 1);
        return;
    }
    const int32_t my_idx = lane < len ? indices[a + lane] : -1;
    const float my_w = lane < len ? data[a + lane] : INFINITY;
    // rank of this entry in ascending weight order, ties by position (np.argsort order up to ties)
    int rank = 0;
    for (int t = 0; t < len; t++) {
        const float wt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), t));
        rank += (wt < my_w || (wt == my_w && t < lane)) ? 1 : 0;
    }
    unsigned long long retained = len == 64 ? ~0ull : ((1ull << len) - 1ull);  // bit per storage position
    for (int idx = 1; idx < len; idx++) {
        const int j = __builtin_ctzll(__ballot(lane < len && rank == idx));  // order[idx]
        const float wj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), j));
        const int32_t idj = __builtin_amdgcn_readlane(my_idx, j);
        for (int kk = 0; kk < idx; kk++) {
            const int l = __builtin_ctzll(__ballot(lane < len && rank == kk));  // order[kk]
            if ((retained >> l) & 1ull) {
                const float wl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), l));
                if (wl > PRUNE_EPS) {
                    const int32_t idk = __builtin_amdgcn_readlane(my_idx, kk);  // storage position kk (reference quirk)
                    const float d = prune_pair_dist(xp, dp, nrm, metric, idj, idk);
                    if (d < wj) {
                        retained &= ~(1ull << j);
                        break;
                    }
                }
            }
        }
    }
    if (lane < len && !((retained >> lane) & 1ull)) data[a + lane] = 0.0f;
}

// pynndescent_.py:728-738: rows longer than max_degree keep the entries <= sorted(row)[max_degree]
__global__ __launch_bounds__(256) void k_degree_prune(int64_t n_rows, const int32_t *__restrict__ indptr,
                                                      float *__restrict__ data, int max_degree) {
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + w;
    if (i >= n_rows) return;
    const int a = indptr[i], len = indptr[i + 1] - a;
    if (len <= max_degree) return;
    // cut = the value v with  #(row < v) <= max_degree < #(row <= v)
    float cut = INFINITY;
    for (int e0 = 0; e0 < len; e0 += 64) {
        const int e = e0 + lane;
        const float v = e < len ? data[a + e] : INFINITY;
        int lt = 0, le = 0;
        for (int t = 0; t < len; t++) {
            const float u = data[a + t];
            lt += u < v ? 1 : 0;
            le += u <= v ? 1 : 0;
        }
        const bool is_cut = e < len && lt <= max_degree && max_degree < le;
        const unsigned long long m = __ballot(is_cut);
        if (m) {
            cut = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_ctzll(m)));
            break;
        }
    }
    for (int e = lane; e < len; e += 64)
        if (data[a + e] > cut) data[a + e] = 0.0f;
}

int nnd_launch_diversify_rows(nnd_ctx *ctx, int32_t *idx_dev, float *dist_dev) {
    hipLaunchKernelGGL(k_diversify_rows, dim3((unsigned)((ctx->n + 3) / 4)), dim3(256), 0, ctx->stream, ctx->xp, ctx->dp, ctx->nrm,
                       ctx->p.metric, ctx->n, ctx->k, idx_dev, dist_dev);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}
int nnd_launch_diversify_csr(nnd_ctx *ctx, const int32_t *indptr_dev, const int32_t *indices_dev, float *data_dev,
                             int *too_long_dev) {
    hipLaunchKernelGGL(k_diversify_csr, dim3((unsigned)((ctx->n + 3) / 4)), dim3(256), 0, ctx->stream, ctx->xp, ctx->dp, ctx->nrm,
                       ctx->p.metric, ctx->n, indptr_dev, indices_dev, data_dev, too_long_dev);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}
int nnd_launch_degree_prune(nnd_ctx *ctx, const int32_t *indptr_dev, float *data_dev, int max_degree) {
    hipLaunchKernelGGL(k_degree_prune, dim3((unsigned)((ctx->n + 3) / 4)), dim3(256), 0, ctx->stream, ctx->n, indptr_dev, data_dev,
                       max_degree);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}
