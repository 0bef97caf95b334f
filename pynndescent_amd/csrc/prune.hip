// prune.hip -- search-graph pruning pass (BASELINE config 5: "+ graph diversification/prune pass").
//
// Replaces
//   diversify                   reference pynndescent_.py:369-403   -> k_diversify_rows<false>
//   diversify_degree_aware      reference pynndescent_.py:433-546   -> k_diversify_rows<true>
//   diversify_csr               reference pynndescent_.py:549-588   -> k_diversify_csr<false>
//   diversify_csr_degree_aware  reference pynndescent_.py:625-726   -> k_diversify_csr<true>
//   degree_prune_internal       reference pynndescent_.py:728-738   -> k_degree_prune
// prune_probability < 1 (diversify_prob): the reference draws tau_rand(rng_state) from ONE state shared by all
// prange threads (a data race: its coin sequence depends on the thread schedule); here the coin of a test is the
// counter hash of (seed, row, entry, compared entry) -- statistically the same Bernoulli(prune_probability).
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

// u in [0,1): the coin of one pruning test (reference: tau_rand(rng_state) < prune_probability)
__device__ __forceinline__ bool prune_coin(uint32_t seed, uint32_t row, uint32_t a, uint32_t b, float prob) {
    if (prob >= 1.0f) return true;
    const uint32_t h = nnd_hash3(seed, row, a * 64u + b);
    return (float)(h >> 8) * (1.0f / 16777216.0f) < prob;
}

// pynndescent_.py:369-403 (AWARE = false) / 433-546 (AWARE = true).  rows: (n,k) ascending; pruned slots become
// (-1, +inf); kept entries stay in order.  AWARE: an entry u whose undirected degree exceeds max_degree is pruned
// against d(i,u) * threshold_factor * alpha (pynndescent_.py:506-531); there is no coin in that variant (the
// reference call site hands diversify_prob to `alpha`, pynndescent_.py:1486-1497 -- the host passes it the same way).
template <bool AWARE>
__global__ __launch_bounds__(256) void k_diversify_rows(const float *__restrict__ xp, int dp, const float *__restrict__ nrm,
                                                        int metric, int64_t n, int k, int32_t *__restrict__ idx,
                                                        float *__restrict__ dist, float prob, uint32_t seed,
                                                        const int32_t *__restrict__ degree, int max_degree,
                                                        float base_rate, float alpha) {
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + w;
    if (i >= n) return;
    const int32_t my_idx = lane < k ? idx[i * k + lane] : -1;
    const float my_d = lane < k ? dist[i * k + lane] : INFINITY;
    float my_fac = 1.0f;
    if (AWARE && my_idx >= 0) {  // pynndescent_.py:506-521
        const float ratio = (float)degree[my_idx] / (float)max_degree;
        if (ratio > 1.0f) {
            const float excess = fminf(ratio - 1.0f, 2.0f);
            my_fac = fmaxf(0.8f, fminf(1.2f, 1.0f + base_rate * excess));
        }
    }
    unsigned long long kept = 1ull;  // position 0 is always kept (pynndescent_.py:374-375)
    for (int j = 1; j < k; j++) {
        const int32_t idj = __builtin_amdgcn_readlane(my_idx, j);
        if (idj < 0) break;  // pynndescent_.py:377-378
        const float dj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_d), j));
        const float lim = AWARE ? dj * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_fac), j)) * alpha : dj;
        bool flag = true;
        unsigned long long m = kept;
        while (m) {
            const int c = __builtin_ctzll(m);
            m &= m - 1;
            const float dc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_d), c));
            if (dc > PRUNE_EPS) {
                const int32_t idc = __builtin_amdgcn_readlane(my_idx, c);
                const float d = prune_pair_dist(xp, dp, nrm, metric, idj, idc);
                if (d < lim && (AWARE || prune_coin(seed, (uint32_t)i, (uint32_t)j, (uint32_t)c, prob))) {  // pynndescent_.py:386-389
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

// pynndescent_.py:549-588 (AWARE = false) / 625-726 (AWARE = true) on CSR rows of length <= 64 (rows of a diversified
// k-NN graph have <= k entries).  Entries are walked in ascending weight order (ties by position).
// AWARE = false: NOTE the reference takes the comparison POINT from storage position `k` (`current_indices[k]`,
// line 577) while it takes weight and retained-flag from the weight order (`l = order[k]`); that is reproduced here.
// AWARE = true (pynndescent_.py:682-719): the walk starts at the first entry, skips weight-0 entries, compares with
// the point at order[k], has no FLOAT32_EPS guard, and prunes when d * threshold_factor(degree of entry j) < w_j.
// Pruned entries get weight 0.
template <bool AWARE>
__global__ __launch_bounds__(256) void k_diversify_csr(const float *__restrict__ xp, int dp, const float *__restrict__ nrm,
                                                       int metric, int64_t n_rows, const int32_t *__restrict__ indptr,
                                                       const int32_t *__restrict__ indices, float *__restrict__ data,
                                                       int *__restrict__ too_long, float prob, uint32_t seed,
                                                       const int32_t *__restrict__ degree, int max_degree,
                                                       float aggressiveness) {
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
    float my_fac = 1.0f;
    if (AWARE && my_idx >= 0) {  // pynndescent_.py:700-709
        const int tgt = (int64_t)my_idx < n_rows ? degree[my_idx] : 0;
        const float ratio = (float)tgt / (float)(max_degree > 1 ? max_degree : 1);
        my_fac = fmaxf(1.0f + 0.04f * aggressiveness * fminf(ratio - 1.0f, 2.0f), 1.0f);
    }
    // rank of this entry in ascending weight order, ties by position (np.argsort order up to ties)
    int rank = 0;
    for (int t = 0; t < len; t++) {
        const float wt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), t));
        rank += (wt < my_w || (wt == my_w && t < lane)) ? 1 : 0;
    }
    unsigned long long retained = len == 64 ? ~0ull : ((1ull << len) - 1ull);  // bit per storage position
    for (int idx = AWARE ? 0 : 1; idx < len; idx++) {
        const int j = __builtin_ctzll(__ballot(lane < len && rank == idx));  // order[idx]
        const float wj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), j));
        if (AWARE && wj == 0.0f) continue;  // pynndescent_.py:685-686
        const int32_t idj = __builtin_amdgcn_readlane(my_idx, j);
        const float fj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_fac), j));
        for (int kk = 0; kk < idx; kk++) {
            const int l = __builtin_ctzll(__ballot(lane < len && rank == kk));  // order[kk]
            if ((retained >> l) & 1ull) {
                const float wl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), l));
                if (AWARE || wl > PRUNE_EPS) {
                    // AWARE: the point at order[kk]; standard: storage position kk (reference quirk)
                    const int32_t idk = __builtin_amdgcn_readlane(my_idx, AWARE ? l : kk);
                    // AWARE has no EPS guard: the row's own vertex (weight EPS) is a comparison point and d(x_i, x_j) is
                    // compared with the stored d(i, j) = wj, the same quantity -- bitwise equal in the reference (same
                    // function, same operands), so never pruned at factor >= 1.  Use the stored value, not a
                    // recomputation that may differ by an ulp.
                    const float d = (AWARE && wl <= PRUNE_EPS) ? wj : prune_pair_dist(xp, dp, nrm, metric, idj, idk);
                    if ((AWARE ? d * fj : d) < wj && prune_coin(seed, (uint32_t)i, (uint32_t)j, (uint32_t)kk, prob)) {
                        retained &= ~(1ull << j);
                        break;
                    }
                }
            }
        }
    }
    if (lane < len && !((retained >> lane) & 1ull)) data[a + lane] = 0.0f;
}


// ---- rows of 65 .. 256 entries (n_neighbors up to NND_WIDE_K: the reference has no bound, utils.py:130-158).  The same walks with
// the row in LDS (indices, distances / weights, factors, kept flags per wave) instead of one entry per lane: LDS
// broadcasts take the place of v_readlane.  Same decisions as the kernels above on rows that fit both (tests).
#define PRUNE_WIDE NND_WIDE_K
struct prune_wide_row {
    int32_t idx[PRUNE_WIDE];
    float w[PRUNE_WIDE];
    float fac[PRUNE_WIDE];
    int32_t rank[PRUNE_WIDE];  // csr: storage position of the entry with this rank in ascending weight order (order[])
    uint8_t kept[PRUNE_WIDE];
};

template <bool AWARE>
__global__ __launch_bounds__(256) void k_diversify_rows_wide(const float *__restrict__ xp, int dp, const float *__restrict__ nrm,
                                                             int metric, int64_t n, int k, int32_t *__restrict__ idx,
                                                             float *__restrict__ dist, float prob, uint32_t seed,
                                                             const int32_t *__restrict__ degree, int max_degree,
                                                             float base_rate, float alpha) {
    __shared__ prune_wide_row rows[4];
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + w;
    if (i >= n) return;
    prune_wide_row &r = rows[w];
    for (int j = lane; j < PRUNE_WIDE; j += 64) {
        const int32_t id = j < k ? idx[i * k + j] : -1;
        r.idx[j] = id;
        r.w[j] = j < k ? dist[i * k + j] : INFINITY;
        float fac = 1.0f;
        if (AWARE && id >= 0) {  // pynndescent_.py:506-521
            const float ratio = (float)degree[id] / (float)max_degree;
            if (ratio > 1.0f) fac = fmaxf(0.8f, fminf(1.2f, 1.0f + base_rate * fminf(ratio - 1.0f, 2.0f)));
        }
        r.fac[j] = fac;
        r.kept[j] = j == 0 ? 1 : 0;  // position 0 is always kept (pynndescent_.py:374-375)
    }
    nnd_wave_lds_sync();
    for (int j = 1; j < k; j++) {
        const int32_t idj = r.idx[j];
        if (idj < 0) break;  // pynndescent_.py:377-378
        const float dj = r.w[j];
        const float lim = AWARE ? dj * r.fac[j] * alpha : dj;
        bool flag = true;
        for (int c = 0; c < j; c++) {
            if (!r.kept[c]) continue;
            if (r.w[c] > PRUNE_EPS) {
                const float d = prune_pair_dist(xp, dp, nrm, metric, idj, r.idx[c]);
                if (d < lim && (AWARE || prune_coin(seed, (uint32_t)i, (uint32_t)j, (uint32_t)c, prob))) {  // pynndescent_.py:386-389
                    flag = false;
                    break;
                }
            }
        }
        if (flag && lane == 0) r.kept[j] = 1;
        nnd_wave_lds_sync();
    }
    // the kept entries, in order, then (-1, +inf)
    int nk = 0;
    for (int j = 0; j < k; j++) {
        if (!r.kept[j]) continue;
        if (lane == 0) {
            idx[i * k + nk] = r.idx[j];
            dist[i * k + nk] = r.w[j];
        }
        nk++;
    }
    for (int j = nk + lane; j < k; j += 64) {
        idx[i * k + j] = -1;
        dist[i * k + j] = INFINITY;
    }
}

template <bool AWARE>
__global__ __launch_bounds__(256) void k_diversify_csr_wide(const float *__restrict__ xp, int dp, const float *__restrict__ nrm,
                                                            int metric, int64_t n_rows, const int32_t *__restrict__ indptr,
                                                            const int32_t *__restrict__ indices, float *__restrict__ data,
                                                            int *__restrict__ too_long, float prob, uint32_t seed,
                                                            const int32_t *__restrict__ degree, int max_degree,
                                                            float aggressiveness) {
    __shared__ prune_wide_row rows[4];
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + w;
    if (i >= n_rows) return;
    const int a = indptr[i], len = indptr[i + 1] - a;
    if (len <= 1) return;
    if (len > PRUNE_WIDE) {
        if (lane == 0) atomicAdd(too_long, 1);
        return;
    }
    prune_wide_row &r = rows[w];
    for (int j = lane; j < len; j += 64) {
        const int32_t id = indices[a + j];
        r.idx[j] = id;
        r.w[j] = data[a + j];
        float fac = 1.0f;
        if (AWARE) {  // pynndescent_.py:700-709
            const int tgt = (int64_t)id < n_rows ? degree[id] : 0;
            const float ratio = (float)tgt / (float)(max_degree > 1 ? max_degree : 1);
            fac = fmaxf(1.0f + 0.04f * aggressiveness * fminf(ratio - 1.0f, 2.0f), 1.0f);
        }
        r.fac[j] = fac;
        r.kept[j] = 1;
    }
    nnd_wave_lds_sync();
    // order[]: storage position by ascending weight, ties by position (np.argsort order up to ties)
    for (int j = lane; j < len; j += 64) {
        const float wj = r.w[j];
        int rk = 0;
        for (int t = 0; t < len; t++) rk += (r.w[t] < wj || (r.w[t] == wj && t < j)) ? 1 : 0;
        r.rank[rk] = j;
    }
    nnd_wave_lds_sync();
    for (int idx = AWARE ? 0 : 1; idx < len; idx++) {
        const int j = r.rank[idx];  // order[idx]
        const float wj = r.w[j];
        if (AWARE && wj == 0.0f) continue;  // pynndescent_.py:685-686
        const int32_t idj = r.idx[j];
        const float fj = r.fac[j];
        for (int kk = 0; kk < idx; kk++) {
            const int l = r.rank[kk];  // order[kk]
            if (!r.kept[l]) continue;
            const float wl = r.w[l];
            if (AWARE || wl > PRUNE_EPS) {
                const int32_t idk = r.idx[AWARE ? l : kk];  // AWARE: the point at order[kk]; standard: storage position kk (reference quirk)
                const float d = (AWARE && wl <= PRUNE_EPS) ? wj : prune_pair_dist(xp, dp, nrm, metric, idj, idk);
                if ((AWARE ? d * fj : d) < wj && prune_coin(seed, (uint32_t)i, (uint32_t)j, (uint32_t)kk, prob)) {
                    if (lane == 0) r.kept[j] = 0;
                    break;
                }
            }
        }
        nnd_wave_lds_sync();
    }
    for (int j = lane; j < len; j += 64)
        if (!r.kept[j]) data[a + j] = 0.0f;
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

int nnd_launch_diversify_rows(nnd_ctx *ctx, int32_t *idx_dev, float *dist_dev, const nnd_prune_opts *o, const int32_t *degree_dev) {
    const dim3 grid((unsigned)((ctx->n + 3) / 4));
    if (ctx->k > PRUNE_WIDE) { ctx->set_error("the pruning pass handles rows of at most %d neighbours", PRUNE_WIDE); return 1; }
    if (ctx->k > 64) {  // rows that do not fit one entry per lane: the LDS variants
        const float base_rate = 0.04f * fmaxf(0.0f, o->aggressiveness);
        if (o->degree_aware)
            hipLaunchKernelGGL(k_diversify_rows_wide<true>, grid, dim3(256), 0, ctx->stream, ctx->xp, ctx->dp, ctx->nrm, ctx->p.metric, ctx->n, ctx->k,
                               idx_dev, dist_dev, 1.0f, o->seed, degree_dev, o->max_degree, base_rate, o->alpha);
        else
            hipLaunchKernelGGL(k_diversify_rows_wide<false>, grid, dim3(256), 0, ctx->stream, ctx->xp, ctx->dp, ctx->nrm, ctx->p.metric, ctx->n, ctx->k,
                               idx_dev, dist_dev, o->prune_probability, o->seed, (const int32_t *)nullptr, 1, 0.0f, 1.0f);
        NND_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (o->degree_aware) {
        const float base_rate = 0.04f * fmaxf(0.0f, o->aggressiveness);  // pynndescent_.py:487-488
        hipLaunchKernelGGL(k_diversify_rows<true>, grid, dim3(256), 0, ctx->stream, ctx->xp, ctx->dp, ctx->nrm, ctx->p.metric,
                           ctx->n, ctx->k, idx_dev, dist_dev, 1.0f, o->seed, degree_dev, o->max_degree, base_rate, o->alpha);
    } else {
        hipLaunchKernelGGL(k_diversify_rows<false>, grid, dim3(256), 0, ctx->stream, ctx->xp, ctx->dp, ctx->nrm, ctx->p.metric,
                           ctx->n, ctx->k, idx_dev, dist_dev, o->prune_probability, o->seed, (const int32_t *)nullptr, 1, 0.0f, 1.0f);
    }
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}
int nnd_launch_diversify_csr(nnd_ctx *ctx, const int32_t *indptr_dev, const int32_t *indices_dev, float *data_dev,
                             int *too_long_dev, const nnd_prune_opts *o, const int32_t *degree_dev) {
    const dim3 grid((unsigned)((ctx->n + 3) / 4));
    if (ctx->k > 64) {  // rows of up to PRUNE_WIDE entries: the LDS variants
        if (o->degree_aware)
            hipLaunchKernelGGL(k_diversify_csr_wide<true>, grid, dim3(256), 0, ctx->stream, ctx->xp, ctx->dp, ctx->nrm, ctx->p.metric, ctx->n, indptr_dev,
                               indices_dev, data_dev, too_long_dev, o->prune_probability, o->seed ^ 0x51ED270Bu, degree_dev, o->max_degree, o->aggressiveness);
        else
            hipLaunchKernelGGL(k_diversify_csr_wide<false>, grid, dim3(256), 0, ctx->stream, ctx->xp, ctx->dp, ctx->nrm, ctx->p.metric, ctx->n, indptr_dev,
                               indices_dev, data_dev, too_long_dev, o->prune_probability, o->seed ^ 0x51ED270Bu, (const int32_t *)nullptr, 1, 0.0f);
        NND_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (o->degree_aware)
        hipLaunchKernelGGL(k_diversify_csr<true>, grid, dim3(256), 0, ctx->stream, ctx->xp, ctx->dp, ctx->nrm, ctx->p.metric,
                           ctx->n, indptr_dev, indices_dev, data_dev, too_long_dev, o->prune_probability, o->seed ^ 0x51ED270Bu,
                           degree_dev, o->max_degree, o->aggressiveness);
    else
        hipLaunchKernelGGL(k_diversify_csr<false>, grid, dim3(256), 0, ctx->stream, ctx->xp, ctx->dp, ctx->nrm, ctx->p.metric,
                           ctx->n, indptr_dev, indices_dev, data_dev, too_long_dev, o->prune_probability, o->seed ^ 0x51ED270Bu,
                           (const int32_t *)nullptr, 1, 0.0f);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}
int nnd_launch_degree_prune(nnd_ctx *ctx, const int32_t *indptr_dev, float *data_dev, int max_degree) {
    hipLaunchKernelGGL(k_degree_prune, dim3((unsigned)((ctx->n + 3) / 4)), dim3(256), 0, ctx->stream, ctx->n, indptr_dev, data_dev,
                       max_degree);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}
