// searchgraph.hip -- the pruning pass of NNDescent._init_search_graph (reference pynndescent_.py:1451-1611) END TO END on the
// device: BASELINE configs[4] "+ graph diversification / prune pass".
//
// The reference glues its numba kernels with scipy on the host: COO -> CSR (1527-1537), `transpose()` (1549), the element-wise
// `maximum` (1599), `setdiag(0)` + `eliminate_zeros()` (1602-1604), `degree_prune` (1606-1609), `(graph != 0)` (1611).  Rounds
// 2-4 of this library ran the three kernels of prune.hip on the device and kept those scipy calls: 128 ms at 290 k points, 92 %
// of it host glue and transfers.  Here the k-NN graph never leaves HBM between the kernels:
//   forward diversify (prune.hip)            rows (n, k), pruned slots -1
//   k_sg_row_counts -> scan -> k_sg_compact  COO -> CSR: forward matrix F (distance 0 -> FLOAT32_EPS, pynndescent_.py:1525)
//   reverse diversify_csr (prune.hip) on F   -- the reference's transpose() SHARES the forward arrays (scipy returns a CSC view
//                                            over the same indptr / indices / data), so its "reverse" pass walks the forward
//                                            rows again and its eliminate_zeros() prunes the forward matrix too: F'
//   k_sg_edge_keys                           every surviving edge (i, j, w) of F', i != j, twice: key (i, j) and key (j, i)
//   rocprim::radix_sort_pairs                by (row, column): the union max(F', F'^T) in CSR order, duplicates adjacent
//   k_sg_heads -> scan -> k_sg_unique        one entry per key with the larger weight (a key occurs at most twice); row counts
//   degree_prune (prune.hip)                 rows longer than round(multiplier * k): entries above the cut get weight 0
//   k_sg_flags -> scan -> k_sg_final         zeros dropped: indptr / indices of the search graph, sorted by column (binary: the
//                                            weights are not part of the result, pynndescent_.py:1611)
// ONE device-to-host copy of indptr / indices at the end (nnd_search_graph_fetch).  The radix sort is rocPRIM's (a library
// sort for glue the reference does with scipy); everything else is a few streaming kernels over <= 2 n k edges.
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "common.h"
#include "state.h"

#define SG_EPS 1.1920929e-07f  // np.finfo(np.float32).eps (pynndescent_.py:65)

// ---------------------------------------------------------------------------------------------- exclusive scan (any n)
#define SG_TILE 2048
__global__ __launch_bounds__(256) void k_sg_block_sums(const int32_t *__restrict__ in, int64_t n, int32_t *__restrict__ bsum) {
    __shared__ int red[4];
    const int64_t base = (int64_t)blockIdx.x * SG_TILE;
    int s = 0;
    for (int q = 0; q < SG_TILE / 256; q++) {
        const int64_t i = base + q * 256 + threadIdx.x;
        s += i < n ? in[i] : 0;
    }
    s = nnd_wave_sum_i32(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// in place, one workgroup: bsum[i] <- sum of bsum[0 .. i); bsum[nb] <- total
__global__ __launch_bounds__(1024) void k_sg_scan_single(int32_t *__restrict__ bsum, int64_t nb) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t i = base + tid;
        const int x = i < nb ? bsum[i] : 0;
        const int inc = nnd_wave_incl_scan_i32(x);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int before = carry_s;
        for (int q = 0; q < w; q++) before += wsum[q];
        if (i < nb) bsum[i] = before + inc - x;
        __syncthreads();
        if (tid == 1023) carry_s = before + inc;
        __syncthreads();
    }
    if (tid == 0) bsum[nb] = carry_s;
}
// out[i] = sum of in[0 .. i), i <= n
__global__ __launch_bounds__(256) void k_sg_scan_apply(const int32_t *__restrict__ in, int64_t n, const int32_t *__restrict__ bsum,
                                                       int32_t *__restrict__ out) {
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * SG_TILE + (int64_t)tid * (SG_TILE / 256);
    int v[SG_TILE / 256];
    int s = 0;
#pragma unroll
    for (int q = 0; q < SG_TILE / 256; q++) {
        v[q] = base + q < n ? in[base + q] : 0;
        s += v[q];
    }
    const int inc = nnd_wave_incl_scan_i32(s);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int run = bsum[blockIdx.x] + inc - s;
    for (int q = 0; q < w; q++) run += wsum[q];
#pragma unroll
    for (int q = 0; q < SG_TILE / 256; q++) {
        if (base + q <= n) out[base + q] = run;  // (position n = the total: the tile that holds it exists, the grid covers n + 1)
        run += v[q];
    }
}
// exclusive scan of in[0 .. n) into out[0 .. n]; `bsum`: scratch of (n + 1) / SG_TILE + 2 words.  in may alias out.
static int sg_scan(nnd_ctx *ctx, const int32_t *in, int32_t *out, int64_t n, int32_t *bsum) {
    const int64_t nb = n / SG_TILE + 1;  // tiles over positions 0 .. n
    hipLaunchKernelGGL(k_sg_block_sums, dim3((unsigned)nb), dim3(256), 0, ctx->stream, in, n, bsum);
    hipLaunchKernelGGL(k_sg_scan_single, dim3(1), dim3(1024), 0, ctx->stream, bsum, nb);
    hipLaunchKernelGGL(k_sg_scan_apply, dim3((unsigned)nb), dim3(256), 0, ctx->stream, in, n, bsum, out);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------- the glue kernels
// compute_degrees (pynndescent_.py:406-418): entries of the row + occurrences as a neighbour
__global__ void k_sg_degrees(const int32_t *__restrict__ idx, int64_t total, int k, int32_t *__restrict__ deg) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int32_t u = idx[e];
    if (u < 0) return;
    atomicAdd(&deg[e / k], 1);
    atomicAdd(&deg[u], 1);
}
// compute_degrees_csr (pynndescent_.py:591-622): row length + occurrences as a column
__global__ void k_sg_degrees_csr(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t n, int32_t *__restrict__ deg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = indptr[i], b = indptr[i + 1];
    atomicAdd(&deg[i], b - a);
    for (int e = a; e < b; e++) {
        const int32_t j = indices[e];
        if (j >= 0 && j < n) atomicAdd(&deg[j], 1);
    }
}
// kept entries of every row of the forward pass
__global__ void k_sg_row_counts(const int32_t *__restrict__ idx, int64_t n, int k, int32_t *__restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = 0;
    for (int j = 0; j < k; j++) c += idx[i * k + j] >= 0 ? 1 : 0;
    cnt[i] = c;
}
// COO -> CSR (pynndescent_.py:1527-1537): the kept entries in row order; distance 0 -> FLOAT32_EPS (1525); min weight (1539)
__global__ void k_sg_compact(const int32_t *__restrict__ idx, float *__restrict__ dist, int64_t n, int k, const int32_t *__restrict__ indptr,
                             int32_t *__restrict__ f_ind, float *__restrict__ f_dat, uint32_t *__restrict__ min_bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int at = indptr[i];
    float mn = INFINITY;
    for (int j = 0; j < k; j++) {
        const int32_t u = idx[i * k + j];
        float w = dist[i * k + j];
        if (w == 0.0f) {
            w = SG_EPS;
            dist[i * k + j] = w;  // (the forward arrays of the `stages` view carry the substitution too)
        }
        if (u < 0) continue;
        f_ind[at] = u;
        f_dat[at] = w;
        mn = fminf(mn, w);
        at++;
    }
    if (mn < INFINITY) atomicMin(min_bits, __float_as_uint(mn));  // weights are positive: the bit patterns order like the values
}
// the two keyed copies of every surviving off-diagonal edge of F'; dead entries get the largest key and sort to the end
__global__ void k_sg_edge_keys(const int32_t *__restrict__ indptr, const int32_t *__restrict__ f_ind, const float *__restrict__ f_dat, int64_t n,
                               uint64_t *__restrict__ keys, float *__restrict__ vals, unsigned long long *__restrict__ live) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int nl = 0, nz = 0;
    for (int e = indptr[i]; e < indptr[i + 1]; e++) {
        const int32_t j = f_ind[e];
        const float w = f_dat[e];
        nz += w != 0.0f ? 1 : 0;
        const bool ok = w != 0.0f && j != (int32_t)i;  // eliminate_zeros (1588); setdiag(0) + eliminate_zeros (1602-1604)
        keys[2 * (int64_t)e] = ok ? ((uint64_t)(uint32_t)i << 32) | (uint32_t)j : ~0ull;
        keys[2 * (int64_t)e + 1] = ok ? ((uint64_t)(uint32_t)j << 32) | (uint32_t)i : ~0ull;
        vals[2 * (int64_t)e] = w;
        vals[2 * (int64_t)e + 1] = w;
        nl += ok ? 2 : 0;
    }
    if (nl) atomicAdd(&live[0], (unsigned long long)nl);
    if (nz) atomicAdd(&live[1], (unsigned long long)nz);  // nnz of F' (the reference's reverse_graph.nnz after eliminate_zeros)
}
// head of every run of equal keys among the first m sorted entries
__global__ void k_sg_heads(const uint64_t *__restrict__ keys, int64_t m, int32_t *__restrict__ head) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    head[e] = (e == 0 || keys[e] != keys[e - 1]) ? 1 : 0;
}
// one entry per key: column, max of the (at most two) weights (fwd.maximum(rev), 1599); entries per row
__global__ void k_sg_unique(const uint64_t *__restrict__ keys, const float *__restrict__ vals, int64_t m, const int32_t *__restrict__ pos,
                            int32_t *__restrict__ u_ind, float *__restrict__ u_dat, int32_t *__restrict__ row_cnt) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    const uint64_t key = keys[e];
    if (e > 0 && keys[e - 1] == key) return;
    float w = vals[e];
    if (e + 1 < m && keys[e + 1] == key) w = fmaxf(w, vals[e + 1]);
    const int32_t at = pos[e];
    u_ind[at] = (int32_t)(uint32_t)key;
    u_dat[at] = w;
    atomicAdd(&row_cnt[key >> 32], 1);
}
__global__ void k_sg_flags(const float *__restrict__ dat, int64_t m, int32_t *__restrict__ flag) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < m) flag[e] = dat[e] != 0.0f ? 1 : 0;
}
// the entries that survived degree_prune, in order (sorted by column inside a row); the rows' new boundaries
__global__ void k_sg_final(const int32_t *__restrict__ u_ind, const float *__restrict__ u_dat, int64_t m, const int32_t *__restrict__ pos,
                           int32_t *__restrict__ out_ind) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < m && u_dat[e] != 0.0f) out_ind[pos[e]] = u_ind[e];
}
__global__ void k_sg_final_ptr(const int32_t *__restrict__ uptr, const int32_t *__restrict__ pos, int64_t n, int32_t *__restrict__ out_ptr,
                               int32_t *__restrict__ max_deg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    out_ptr[i] = pos[uptr[i]];
    if (i < n) atomicMax(max_deg, pos[uptr[i + 1]] - pos[uptr[i]]);
}

// ---------------------------------------------------------------------------------------------- orchestration
struct nnd_sg_state {
    unsigned char *buf = nullptr;  // one allocation, grow-only, carved below
    size_t cap = 0;
    int32_t *out_ptr = nullptr, *out_ind = nullptr;  // the finished graph (device)
    int64_t final_nnz = 0;
};

void nnd_search_graph_free(nnd_ctx *ctx) {
    if (!ctx->sg) return;
    if (ctx->sg->buf) (void)hipFree(ctx->sg->buf);
    delete ctx->sg;
    ctx->sg = nullptr;
}

int nnd_search_graph_impl(nnd_ctx *ctx, const int32_t *idx_src, const float *dist_src, bool src_on_device, int n_neighbors, float multiplier,
                          float diversify_prob, bool aware, float aggressiveness, uint32_t seed, int32_t *fwd_rows_host, float *fwd_dist_host,
                          nnd_search_graph_stats *st) {
    const int64_t n = ctx->n;
    const int k = ctx->k;
    const int64_t nk = n * k;
    if (2 * nk >= (int64_t)0x7FFFFFF0) { ctx->set_error("nnd_search_graph: 2 n k = %lld edges exceed the int32 edge positions of the pass", (long long)(2 * nk)); return 1; }
    hipStream_t s = ctx->stream;
    if (!ctx->sg) ctx->sg = new nnd_sg_state();
    nnd_sg_state *g = ctx->sg;
    // ---- carve the workspace
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t o_di = take(4 * nk), o_dd = take(4 * nk), o_cnt = take(4 * (n + 2)), o_ptrF = take(4 * (n + 2)), o_deg = take(4 * (n + 2)),
                 o_find = take(4 * nk), o_fdat = take(4 * nk), o_keys = take(8 * 2 * nk), o_keys2 = take(8 * 2 * nk), o_vals = take(4 * 2 * nk),
                 o_vals2 = take(4 * 2 * nk), o_flag = take(4 * (2 * nk + 2)), o_pos = take(4 * (2 * nk + 2)), o_uind = take(4 * 2 * nk),
                 o_udat = take(4 * 2 * nk), o_uptr = take(4 * (n + 2)), o_bsum = take(4 * ((2 * nk + 1) / SG_TILE + 4)), o_misc = take(256),
                 o_outp = take(4 * (n + 2)), o_outi = take(4 * 2 * nk);
    size_t sort_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (float *)nullptr, (float *)nullptr, (size_t)(2 * nk), 0, 64, s) !=
        hipSuccess) { ctx->set_error("nnd_search_graph: rocprim::radix_sort_pairs (size query) failed"); return 1; }
    const size_t o_sort = take(sort_bytes + 256);
    if (off > g->cap) {
        if (g->buf) NND_HIP_CHECK(hipFree(g->buf));
        g->buf = nullptr;
        g->cap = 0;
        NND_HIP_CHECK(hipMalloc((void **)&g->buf, off));
        g->cap = off;
    }
    unsigned char *B = g->buf;
    int32_t *di = (int32_t *)(B + o_di), *cnt = (int32_t *)(B + o_cnt), *ptrF = (int32_t *)(B + o_ptrF), *deg = (int32_t *)(B + o_deg),
            *f_ind = (int32_t *)(B + o_find), *flag = (int32_t *)(B + o_flag), *pos = (int32_t *)(B + o_pos), *u_ind = (int32_t *)(B + o_uind),
            *uptr = (int32_t *)(B + o_uptr), *bsum = (int32_t *)(B + o_bsum), *out_ptr = (int32_t *)(B + o_outp), *out_ind = (int32_t *)(B + o_outi);
    float *dd = (float *)(B + o_dd), *f_dat = (float *)(B + o_fdat), *vals = (float *)(B + o_vals), *vals2 = (float *)(B + o_vals2), *u_dat = (float *)(B + o_udat);
    uint64_t *keys = (uint64_t *)(B + o_keys), *keys2 = (uint64_t *)(B + o_keys2);
    unsigned long long *misc = (unsigned long long *)(B + o_misc);  // [0] live keyed edges, [1] nnz of F', [2] min weight bits, [3] max degree, [4] csr too long
    hipEvent_t e0 = ctx->ev0, e1 = ctx->ev1;
    NND_HIP_CHECK(hipEventRecord(e0, s));
    // ---- the caller's graph is not modified: the pass works on a copy
    const hipMemcpyKind kind = src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    NND_HIP_CHECK(hipMemcpyAsync(di, idx_src, 4 * (size_t)nk, kind, s));
    NND_HIP_CHECK(hipMemcpyAsync(dd, dist_src, 4 * (size_t)nk, kind, s));
    NND_HIP_CHECK(hipMemsetAsync(misc, 0, 256, s));
    NND_HIP_CHECK(hipMemsetAsync((unsigned char *)misc + 16, 0xFF, 4, s));
    const unsigned gn = (unsigned)((n + 255) / 256), gn1 = (unsigned)((n + 256) / 256);
    // ---- forward pass (1476-1518)
    nnd_prune_opts of{};
    of.prune_probability = aware ? 1.0f : diversify_prob;
    of.degree_aware = aware ? 1 : 0;
    of.max_degree = aware ? (int)(multiplier * n_neighbors > 1.0f ? multiplier * n_neighbors : 1.0f) : 1;  // int(multiplier * k), 1478
    of.aggressiveness = aggressiveness;
    of.alpha = aware ? diversify_prob : 1.0f;  // the reference hands diversify_prob to `alpha` there (1486-1497)
    of.seed = seed;
    if (aware) {
        NND_HIP_CHECK(hipMemsetAsync(deg, 0, 4 * (size_t)n, s));
        hipLaunchKernelGGL(k_sg_degrees, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, di, nk, k, deg);
    }
    if (nnd_launch_diversify_rows(ctx, di, dd, &of, aware ? deg : nullptr)) return 1;
    // ---- COO -> CSR
    hipLaunchKernelGGL(k_sg_row_counts, dim3(gn), dim3(256), 0, s, di, n, k, cnt);
    if (sg_scan(ctx, cnt, ptrF, n, bsum)) return 1;
    hipLaunchKernelGGL(k_sg_compact, dim3(gn), dim3(256), 0, s, di, dd, n, k, ptrF, f_ind, f_dat, (uint32_t *)misc + 4);
    if (fwd_rows_host) NND_HIP_CHECK(hipMemcpyAsync(fwd_rows_host, di, 4 * (size_t)nk, hipMemcpyDeviceToHost, s));
    if (fwd_dist_host) NND_HIP_CHECK(hipMemcpyAsync(fwd_dist_host, dd, 4 * (size_t)nk, hipMemcpyDeviceToHost, s));
    // ---- "reverse" pass on the forward rows (1549-1587; shared arrays)
    nnd_prune_opts orv = of;
    orv.prune_probability = diversify_prob;
    orv.max_degree = n_neighbors;  // 1567
    orv.alpha = 1.0f;
    if (aware) {
        NND_HIP_CHECK(hipMemsetAsync(deg, 0, 4 * (size_t)n, s));
        hipLaunchKernelGGL(k_sg_degrees_csr, dim3(gn), dim3(256), 0, s, ptrF, f_ind, n, deg);
    }
    if (nnd_launch_diversify_csr(ctx, ptrF, f_ind, f_dat, (int *)(misc + 4), &orv, aware ? deg : nullptr)) return 1;
    // ---- union max(F', F'^T) without its diagonal: keyed edges, sorted, duplicates folded
    hipLaunchKernelGGL(k_sg_edge_keys, dim3(gn), dim3(256), 0, s, ptrF, f_ind, f_dat, n, keys, vals, misc);
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 52, misc, 40, hipMemcpyDeviceToHost, s));
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 57, ptrF + n, 4, hipMemcpyDeviceToHost, s));
    NND_HIP_CHECK(nnd_sync_spin(ctx));  // the host needs the edge counts to size the sort and the launches behind it
    const int64_t m_live = (int64_t)ctx->h_pin[52], nnz_fp = (int64_t)ctx->h_pin[53];
    const int64_t forward_nnz = (int64_t) * (const int32_t *)(ctx->h_pin + 57);
    if ((int)ctx->h_pin[56] != 0) { ctx->set_error("nnd_search_graph: %d rows are longer than the pass handles", (int)ctx->h_pin[56]); return 1; }
    const int64_t n_keyed = 2 * forward_nnz;
    // key bits that matter: the 32 of the column + those of a row number < n, one to spare: the dead keys (all ones) then have
    // a "row" of all ones in the sorted bits, larger than any live row, and still sort last
    unsigned key_bits = 34;
    while (key_bits < 64 && ((uint64_t)n >> (key_bits - 33)) != 0) key_bits++;
    if (n_keyed > 0 &&
        rocprim::radix_sort_pairs(B + o_sort, sort_bytes, keys, keys2, vals, vals2, (size_t)n_keyed, 0u, key_bits, s) != hipSuccess) {
        ctx->set_error("nnd_search_graph: rocprim::radix_sort_pairs failed");
        return 1;
    }
    int64_t union_nnz = 0;
    NND_HIP_CHECK(hipMemsetAsync(cnt, 0, 4 * (size_t)(n + 1), s));
    if (m_live > 0) {
        hipLaunchKernelGGL(k_sg_heads, dim3((unsigned)((m_live + 255) / 256)), dim3(256), 0, s, keys2, m_live, flag);
        if (sg_scan(ctx, flag, pos, m_live, bsum)) return 1;
        hipLaunchKernelGGL(k_sg_unique, dim3((unsigned)((m_live + 255) / 256)), dim3(256), 0, s, keys2, vals2, m_live, pos, u_ind, u_dat, cnt);
        NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 58, pos + m_live, 4, hipMemcpyDeviceToHost, s));
    }
    if (sg_scan(ctx, cnt, uptr, n, bsum)) return 1;
    NND_HIP_CHECK(nnd_sync_spin(ctx));
    if (m_live > 0) union_nnz = (int64_t) * (const int32_t *)(ctx->h_pin + 58);
    // ---- degree_prune (1606-1609) and the final compaction; (graph != 0) keeps the pattern only (1611)
    const int max_degree = (int)nearbyintf(multiplier * (float)n_neighbors);  // np.round: half to even
    if (union_nnz > 0) {
        if (nnd_launch_degree_prune(ctx, uptr, u_dat, max_degree)) return 1;
        hipLaunchKernelGGL(k_sg_flags, dim3((unsigned)((union_nnz + 255) / 256)), dim3(256), 0, s, u_dat, union_nnz, flag);
        if (sg_scan(ctx, flag, pos, union_nnz, bsum)) return 1;
        hipLaunchKernelGGL(k_sg_final, dim3((unsigned)((union_nnz + 255) / 256)), dim3(256), 0, s, u_ind, u_dat, union_nnz, pos, out_ind);
        hipLaunchKernelGGL(k_sg_final_ptr, dim3(gn1), dim3(256), 0, s, uptr, pos, n, out_ptr, (int32_t *)(misc + 3));
        NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 59, pos + union_nnz, 4, hipMemcpyDeviceToHost, s));
    } else {
        NND_HIP_CHECK(hipMemsetAsync(out_ptr, 0, 4 * (size_t)(n + 1), s));
    }
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 60, misc + 2, 16, hipMemcpyDeviceToHost, s));
    NND_HIP_CHECK(hipEventRecord(e1, s));
    NND_HIP_CHECK(nnd_sync_spin(ctx));
    NND_HIP_CHECK(hipGetLastError());
    g->out_ptr = out_ptr;
    g->out_ind = out_ind;
    g->final_nnz = union_nnz > 0 ? (int64_t) * (const int32_t *)(ctx->h_pin + 59) : 0;
    if (st) {
        memset(st, 0, sizeof(*st));
        st->forward_nnz = forward_nnz;
        st->reverse_nnz = nnz_fp;
        st->union_nnz = union_nnz;
        st->final_nnz = g->final_nnz;
        const uint32_t mb = *(const uint32_t *)(ctx->h_pin + 60);
        st->min_distance = forward_nnz > 0 && mb != 0xFFFFFFFFu ? __builtin_bit_cast(float, mb) : 0.0f;
        st->max_degree_out = *(const int32_t *)(ctx->h_pin + 61);
        (void)hipEventElapsedTime(&st->ms_device, e0, e1);
    }
    return 0;
}

int nnd_search_graph_fetch_impl(nnd_ctx *ctx, int32_t *indptr_host, int32_t *indices_host) {
    if (!ctx->sg || !ctx->sg->out_ptr) { ctx->set_error("nnd_search_graph_fetch: no search graph has been built on this handle"); return 1; }
    NND_HIP_CHECK(hipMemcpyAsync(indptr_host, ctx->sg->out_ptr, 4 * (size_t)(ctx->n + 1), hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->sg->final_nnz > 0)
        NND_HIP_CHECK(hipMemcpyAsync(indices_host, ctx->sg->out_ind, 4 * (size_t)ctx->sg->final_nnz, hipMemcpyDeviceToHost, ctx->stream));
    NND_HIP_CHECK(nnd_sync_spin(ctx));
    return 0;
}
