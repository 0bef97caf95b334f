// hubtree.hip -- the graph-informed ("hub") search tree of NNDescent.prepare(), built on the GPU.
//
// Replaces make_hub_tree / make_hub_euclidean_tree / make_hub_angular_tree / euclidean_hub_split /
// angular_hub_split / get_top_k_hub_indices (reference rp_trees.py:714-1312) and convert_tree_format /
// recursive_convert (rp_trees.py:2926-3049): the output is the reference's FlatTree -- hyperplanes (n_nodes, dim),
// offsets (n_nodes), children (n_nodes, 2), indices (n) in pre-order numbering -- so the reference's (or this
// package's) search closure can descend it unchanged.
//
// The reference recurses node by node; every node: the three members of highest global in-degree ("hubs", ties to the
// member that comes first) give three candidate hyperplanes (the bisectors of the hub pairs); every member is
// projected on all three; the most balanced valid split wins; a best balance below 0.1 makes the node a (large) leaf.
// The tree is DETERMINISTIC given the graph, so it is rebuilt here decision for decision:
//   * members of a node are always in ascending id order there (root = arange(n), stable partitions), so "the hub that
//     comes first" is the smaller id and the three hubs of a node are its three members of smallest GLOBAL hub rank
//     (rank = position in the order by (-in-degree, id)).  Every node's members are therefore kept in TWO orders, by
//     id and by hub rank, both partitioned stably level by level: the hubs of a node are simply the first three
//     entries of its rank-ordered segment -- no per-node search at all;
//   * level-synchronous: one pass over all positions per level computes the three margins of every member of every
//     splittable node (thread per member, float32 multiply-then-add in dimension order -- the arithmetic of the
//     reference's loops, rp_trees.py:865-869, 990-993; no FMA contraction -- so the sides and with them the tree
//     match the un-jitted reference run bit for bit), one pass picks the winner per node, the stable partition reuses
//     the scan / scatter kernels of the RP forest (rpforest.hip);
//   * the host assembles the per-level tables into the pre-order FlatTree (nnd_hub_tree_fetch).
// Only the all-candidates-invalid fallback differs: the reference re-draws every member's side from the sequential
// Tausworthe stream (rp_trees.py:902-910, in recursion order); here it is the counter hash of (seed, id, depth).
#include <algorithm>
#include <vector>

#include "common.h"
#include "state.h"

#define HUB_EPS 1e-8f             // rp_trees.py:23
#define HUB_MIN_BALANCE 0.1f      // rp_trees.py:803 MIN_SPLIT_BALANCE

// ------------------------------------------------------------------ kernels --
__global__ void k_hub_init(int32_t *__restrict__ ord_id, int32_t *__restrict__ pos_id, int32_t *__restrict__ pos_rk, int64_t n,
                           int splittable) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    ord_id[g] = (int32_t)g;
    pos_id[g] = splittable ? 0 : -1;
    pos_rk[g] = splittable ? 0 : -1;
}

// One thread per (splittable segment, candidate): the candidate's hyperplane and offset, float32, in dimension order.
// euclidean (rp_trees.py:853-863): h = x_l - x_r; off -= h[d] * (x_l[d] + x_r[d]) / 2
// angular   (rp_trees.py:967-986): h = x_l / |x_l| - x_r / |x_r|, normalised; offset 0 (norms < EPS -> 1)
__global__ void k_hub_planes(const float *__restrict__ x, int d, const int32_t *__restrict__ ord_rk,
                             const int32_t *__restrict__ seg_start, const int32_t *__restrict__ seg_len,
                             const uint8_t *__restrict__ seg_split, int n_segs, int angular, float *__restrict__ hv,
                             float *__restrict__ off) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = t / 3, c = t - 3 * s;
    if (s >= n_segs || !seg_split[s]) return;
    const int len = seg_len[s];
    const int nh = len < 3 ? len : 3;
    const int hi = c == 2 ? 1 : 0, hj = c == 0 ? 1 : 2;  // pairs (0,1), (0,2), (1,2) in the reference's loop order
    float *h = hv + ((int64_t)s * 3 + c) * d;
    if (hj >= nh) {
        off[s * 4 + c] = 0.0f;
        return;
    }
    const float *xl = x + (int64_t)ord_rk[seg_start[s] + hi] * d;
    const float *xr = x + (int64_t)ord_rk[seg_start[s] + hj] * d;
    if (!angular) {
        float o = 0.0f;
        for (int j = 0; j < d; j++) {
            const float v = __fsub_rn(xl[j], xr[j]);
            h[j] = v;
            o = __fsub_rn(o, __fmul_rn(__fmul_rn(v, __fadd_rn(xl[j], xr[j])), 0.5f));
        }
        off[s * 4 + c] = o;
    } else {
        float ln = 0.0f, rn = 0.0f;
        for (int j = 0; j < d; j++) {
            ln = __fadd_rn(ln, __fmul_rn(xl[j], xl[j]));  // utils.py:86-91 norm()
            rn = __fadd_rn(rn, __fmul_rn(xr[j], xr[j]));
        }
        ln = __fsqrt_rn(ln);
        rn = __fsqrt_rn(rn);
        if (fabsf(ln) < HUB_EPS) ln = 1.0f;
        if (fabsf(rn) < HUB_EPS) rn = 1.0f;
        float hn = 0.0f;
        for (int j = 0; j < d; j++) {
            const float v = __fsub_rn(__fdiv_rn(xl[j], ln), __fdiv_rn(xr[j], rn));
            h[j] = v;
            hn = __fadd_rn(hn, __fmul_rn(v, v));
        }
        hn = __fsqrt_rn(hn);
        if (fabsf(hn) < HUB_EPS) hn = 1.0f;
        for (int j = 0; j < d; j++) h[j] = __fdiv_rn(h[j], hn);
        off[s * 4 + c] = 0.0f;
    }
}

// One thread per position of the id order: the member's side under each candidate (bits 0..2) and under the fallback
// coin (bit 3); left counts per (segment, candidate) -- one atomic per wave when the wave lies inside one segment.
__global__ __launch_bounds__(256) void k_hub_margins(const float *__restrict__ x, int d, const int32_t *__restrict__ ord_id,
                                                     const int32_t *__restrict__ pos_id, const int32_t *__restrict__ seg_start,
                                                     const int32_t *__restrict__ seg_len, int64_t n, const float *__restrict__ hv,
                                                     const float *__restrict__ off, uint32_t seed, int depth,
                                                     uint8_t *__restrict__ sidebits, int32_t *__restrict__ nleft4) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = g < n ? pos_id[g] : -1;
    int bits = 0;
    if (s >= 0) {
        const int p = ord_id[g];
        const int i = (int)(g - seg_start[s]);
        const int len = seg_len[s];
        const int ncand = len >= 3 ? 3 : (len == 2 ? 1 : 0);
        const float *xr = x + (int64_t)p * d;
        const float *h0 = hv + (int64_t)s * 3 * d, *h1 = h0 + d, *h2 = h1 + d;
        float m0 = off[s * 4 + 0], m1 = off[s * 4 + 1], m2 = off[s * 4 + 2];
        for (int j = 0; j < d; j++) {  // rp_trees.py:866-869 / 991-993: margin += hyperplane_vector[d] * data[indices[i], d]
            const float xv = xr[j];
            m0 = __fadd_rn(m0, __fmul_rn(h0[j], xv));
            if (ncand == 3) {
                m1 = __fadd_rn(m1, __fmul_rn(h1[j], xv));
                m2 = __fadd_rn(m2, __fmul_rn(h2[j], xv));
            }
        }
        const float mm[3] = {m0, m1, m2};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            int side = 1;
            if (c < ncand) side = mm[c] > HUB_EPS ? 0 : (mm[c] < -HUB_EPS ? 1 : (i & 1));  // rp_trees.py:871-882
            bits |= side << c;
        }
        bits |= (int)(nnd_hash3(seed ^ 0x6A09E667u, (uint32_t)p, (uint32_t)depth) & 1u) << 3;
        sidebits[p] = (uint8_t)bits;
    }
    // left counts
    const int s0 = __builtin_amdgcn_readfirstlane(s);
    const bool uniform = __ballot(s != s0) == 0ull;
    if (uniform) {
        if (s0 < 0) return;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int cnt = __popcll(__ballot(((bits >> c) & 1) == 0));
            if (nnd_lane() == 0 && cnt) atomicAdd(&nleft4[s0 * 4 + c], cnt);
        }
    } else if (s >= 0) {
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (((bits >> c) & 1) == 0) atomicAdd(&nleft4[s * 4 + c], 1);
    }
}

// One thread per segment: the winner (rp_trees.py:884-899), the balance rule (rp_trees.py:1079-1084), the fallback.
// choice: 0..2 candidate, 3 fallback coin, -1 the node is a leaf.
__global__ void k_hub_choose(const int32_t *__restrict__ seg_len, const uint8_t *__restrict__ seg_split, int n_segs,
                             const int32_t *__restrict__ nleft4, int32_t *__restrict__ choice, int32_t *__restrict__ nl_out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    if (!seg_split[s]) {
        choice[s] = -1;
        nl_out[s] = 0;
        return;
    }
    const int len = seg_len[s];
    const int ncand = len >= 3 ? 3 : (len == 2 ? 1 : 0);
    float best = 0.0f;
    int bc = -1, bnl = 0;
    for (int c = 0; c < ncand; c++) {
        const int nl = nleft4[s * 4 + c], nr = len - nl;
        if (nl == 0 || nr == 0) continue;
        const float bal = __fdiv_rn((float)(nl < nr ? nl : nr), (float)len);
        if (bal > best) {
            best = bal;
            bc = c;
            bnl = nl;
        }
    }
    // No valid candidate: the reference assigns sides at random (rp_trees.py:902-910) but returns best_balance = 0, so
    // make_hub_*_tree always turns that node into a leaf (balance < MIN_SPLIT_BALANCE, rp_trees.py:1079-1084): the coin
    // split never survives.  Same here: bc stays -1.
    if (best < HUB_MIN_BALANCE) bc = -1;  // too unbalanced: a leaf instead (rp_trees.py:1079-1084)
    choice[s] = bc;
    nl_out[s] = bc >= 0 ? bnl : 0;
}

// single workgroup: children of the splitting segments -> next level's segment list; seg_child for the scatter
__global__ __launch_bounds__(256) void k_hub_children(const int32_t *__restrict__ seg_start, const int32_t *__restrict__ seg_len,
                                                      const int32_t *__restrict__ choice, const int32_t *__restrict__ nl, int n_segs,
                                                      int leaf_size, int child_can_split, int32_t *__restrict__ next_start,
                                                      int32_t *__restrict__ next_len, uint8_t *__restrict__ next_split,
                                                      int32_t *__restrict__ seg_child, int32_t *__restrict__ split_rank,
                                                      int32_t *__restrict__ out_counts) {
    __shared__ int part[256], parts[256];
    const int chunk = (n_segs + 255) / 256;
    const int s0 = threadIdx.x * chunk, s1 = s0 + chunk < n_segs ? s0 + chunk : n_segs;
    int cnt = 0;
    for (int s = s0; s < s1; s++) cnt += choice[s] >= 0 ? 1 : 0;
    part[threadIdx.x] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; i++) {
            const int v = part[i];
            part[i] = run;
            run += v;
        }
        out_counts[0] = 2 * run;  // segments of the next level
    }
    __syncthreads();
    int run = part[threadIdx.x], nsplit = 0;
    for (int s = s0; s < s1; s++) {
        if (choice[s] < 0) {
            seg_child[2 * s] = seg_child[2 * s + 1] = -1;
            split_rank[s] = -1;
            continue;
        }
        split_rank[s] = run;
        const int a = seg_start[s], len = seg_len[s], l = nl[s];
        const int lens[2] = {l, len - l}, starts[2] = {a, a + l};
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int idx = 2 * run + c;
            const int sp = (child_can_split && lens[c] > leaf_size) ? 1 : 0;  // rp_trees.py:1066
            next_start[idx] = starts[c];
            next_len[idx] = lens[c];
            next_split[idx] = (uint8_t)sp;
            seg_child[2 * s + c] = sp ? idx : -1;  // members of a child that is a leaf are final
            nsplit += sp;
        }
        run++;
    }
    parts[threadIdx.x] = nsplit;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int i = 0; i < 256; i++) tot += parts[i];
        out_counts[1] = tot;  // of which splittable
    }
}

// the winning hyperplane of the r-th splitting segment -> win[r] (what the host needs of this level's 3 * S candidates)
__global__ void k_hub_winners(const float *__restrict__ hv, const float *__restrict__ off, const int32_t *__restrict__ choice,
                              const int32_t *__restrict__ split_rank, int n_segs, int d, float *__restrict__ win,
                              float *__restrict__ win_off) {
    const int s = blockIdx.x;
    if (s >= n_segs) return;
    const int r = split_rank[s], c = choice[s];
    if (r < 0) return;
    for (int j = threadIdx.x; j < d; j += blockDim.x) win[(int64_t)r * d + j] = c < 3 ? hv[((int64_t)s * 3 + c) * d + j] : 0.0f;
    if (threadIdx.x == 0) win_off[r] = c < 3 ? off[s * 4 + c] : 0.0f;  // fallback split: zero hyperplane, offset 0 (rp_trees.py:843-845)
}

// per position of one ordering: the side under the segment's winning candidate; members of nodes that do not split
// leave the passes (pos = -1)
__global__ void k_hub_sides(const int32_t *__restrict__ ord, int32_t *__restrict__ pos, const int32_t *__restrict__ choice,
                            const uint8_t *__restrict__ sidebits, int64_t n, uint8_t *__restrict__ side) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const int s = pos[g];
    if (s < 0) return;
    const int c = choice[s];
    if (c < 0) {
        pos[g] = -1;
        return;
    }
    side[g] = (sidebits[ord[g]] >> c) & 1;
}

// ------------------------------------------------------------------ host --
struct hub_level {
    int n_segs = 0;
    std::vector<int32_t> start, len, choice, nl;
    std::vector<float> hv;   // (n_splitting, d) winning hyperplanes, in segment order
    std::vector<float> off;  // (n_splitting)
};

struct nnd_hub_result {
    std::vector<float> hyperplanes, offsets;
    std::vector<int32_t> children, indices;
    int64_t n_nodes = 0;
    int32_t max_leaf = 0;
};

#define HUB_HIP(expr)                                                                                \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            ctx->set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            rc = 1;                                                                                  \
            goto done;                                                                               \
        }                                                                                            \
    } while (0)

int nnd_hub_tree_build_impl(nnd_ctx *ctx, const int32_t *rank_order_host, int leaf_size, int max_depth, int angular) {
    const int64_t n = ctx->n;
    const int d = ctx->d;
    int rc = 0;
    if (!ctx->perm[0] || ctx->P < n) { ctx->set_error("nnd_hub_tree_build: create the handle with n_trees >= 1 (it borrows the forest's scan / scatter buffers)"); return 1; }
    if (leaf_size < 1) leaf_size = 1;
    const int64_t S_max = 2 * (n / (leaf_size + 1) + 1) + 2;  // children of one level's splitting nodes
    std::vector<void *> tmp;
    auto dev = [&](size_t bytes) -> void * {
        void *p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
        tmp.push_back(p);
        return p;
    };
    int32_t *ord_rk[2] = {(int32_t *)dev(4 * n), (int32_t *)dev(4 * n)};
    int32_t *pos_rk[2] = {(int32_t *)dev(4 * n), (int32_t *)dev(4 * n)};
    uint8_t *side_rk = (uint8_t *)dev(n), *sidebits = (uint8_t *)dev(n);
    int32_t *sst[2] = {(int32_t *)dev(4 * S_max), (int32_t *)dev(4 * S_max)};
    int32_t *sln[2] = {(int32_t *)dev(4 * S_max), (int32_t *)dev(4 * S_max)};
    uint8_t *ssp[2] = {(uint8_t *)dev(S_max), (uint8_t *)dev(S_max)};
    int32_t *choice = (int32_t *)dev(4 * S_max), *nl = (int32_t *)dev(4 * S_max), *nleft4 = (int32_t *)dev(16 * S_max);
    int32_t *nleft_a = (int32_t *)dev(4 * S_max), *nleft_b = (int32_t *)dev(4 * S_max), *seg_child = (int32_t *)dev(8 * S_max);
    float *hv = (float *)dev(sizeof(float) * 3 * (size_t)d * (size_t)S_max), *off = (float *)dev(16 * S_max);
    int32_t *counts = (int32_t *)dev(8), *split_rank = (int32_t *)dev(4 * S_max);
    float *win = (float *)dev(sizeof(float) * (size_t)d * (size_t)(S_max / 2 + 2)), *win_off = (float *)dev(sizeof(float) * (size_t)(S_max / 2 + 2));
    std::vector<hub_level> levels;
    nnd_hub_result *res = new nnd_hub_result();
    int32_t *ord_id[2] = {ctx->perm[0], ctx->perm[1]}, *pos_id[2] = {ctx->pos_seg[0], ctx->pos_seg[1]};
    int cur = 0, depth = 0, S = 1, n_split = 0;
    for (void *p : tmp)
        if (!p) { ctx->set_error("nnd_hub_tree_build: out of device memory"); rc = 1; goto done; }
    {
        const int splittable = (n > leaf_size && max_depth > 0) ? 1 : 0;
        const unsigned gridN = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_hub_init, dim3(gridN), dim3(256), 0, ctx->stream, ord_id[0], pos_id[0], pos_rk[0], n, splittable);
        HUB_HIP(hipMemcpyAsync(ord_rk[0], rank_order_host, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        const int32_t h0[2] = {0, (int32_t)n};
        const uint8_t hs = (uint8_t)splittable;
        HUB_HIP(hipMemcpyAsync(sst[0], &h0[0], 4, hipMemcpyHostToDevice, ctx->stream));
        HUB_HIP(hipMemcpyAsync(sln[0], &h0[1], 4, hipMemcpyHostToDevice, ctx->stream));
        HUB_HIP(hipMemcpyAsync(ssp[0], &hs, 1, hipMemcpyHostToDevice, ctx->stream));
        HUB_HIP(hipStreamSynchronize(ctx->stream));
        n_split = splittable;
        while (true) {
            hub_level lv;
            lv.n_segs = S;
            if (n_split > 0) {
                HUB_HIP(hipMemsetAsync(nleft4, 0, sizeof(int32_t) * 4 * (size_t)S, ctx->stream));
                hipLaunchKernelGGL(k_hub_planes, dim3((unsigned)((3 * S + 255) / 256)), dim3(256), 0, ctx->stream, ctx->x_orig, d, ord_rk[cur],
                                   sst[cur], sln[cur], ssp[cur], S, angular, hv, off);
                hipLaunchKernelGGL(k_hub_margins, dim3(gridN), dim3(256), 0, ctx->stream, ctx->x_orig, d, ord_id[cur], pos_id[cur], sst[cur],
                                   sln[cur], n, hv, off, ctx->seed, depth, sidebits, nleft4);
            }
            hipLaunchKernelGGL(k_hub_choose, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, ctx->stream, sln[cur], ssp[cur], S, nleft4, choice, nl);
            const int child_can_split = (max_depth - (depth + 1)) > 0 ? 1 : 0;
            hipLaunchKernelGGL(k_hub_children, dim3(1), dim3(256), 0, ctx->stream, sst[cur], sln[cur], choice, nl, S, leaf_size, child_can_split,
                               sst[1 - cur], sln[1 - cur], ssp[1 - cur], seg_child, split_rank, counts);
            if (n_split > 0)
                hipLaunchKernelGGL(k_hub_winners, dim3((unsigned)S), dim3(64), 0, ctx->stream, hv, off, choice, split_rank, S, d, win, win_off);
            HUB_HIP(hipGetLastError());
            // this level's tables -> host (the tree is assembled there)
            lv.start.resize(S); lv.len.resize(S); lv.choice.resize(S); lv.nl.resize(S);
            HUB_HIP(hipMemcpyAsync(lv.start.data(), sst[cur], 4 * (size_t)S, hipMemcpyDeviceToHost, ctx->stream));
            HUB_HIP(hipMemcpyAsync(lv.len.data(), sln[cur], 4 * (size_t)S, hipMemcpyDeviceToHost, ctx->stream));
            HUB_HIP(hipMemcpyAsync(lv.choice.data(), choice, 4 * (size_t)S, hipMemcpyDeviceToHost, ctx->stream));
            HUB_HIP(hipMemcpyAsync(lv.nl.data(), nl, 4 * (size_t)S, hipMemcpyDeviceToHost, ctx->stream));
            int32_t hc[2] = {0, 0};
            HUB_HIP(hipMemcpyAsync(hc, counts, 8, hipMemcpyDeviceToHost, ctx->stream));
            HUB_HIP(hipStreamSynchronize(ctx->stream));
            if (hc[0] > 0) {  // the winners of this level's hc[0] / 2 splitting segments
                const size_t ns = (size_t)hc[0] / 2;
                lv.hv.resize(ns * d);
                lv.off.resize(ns);
                HUB_HIP(hipMemcpyAsync(lv.hv.data(), win, sizeof(float) * lv.hv.size(), hipMemcpyDeviceToHost, ctx->stream));
                HUB_HIP(hipMemcpyAsync(lv.off.data(), win_off, sizeof(float) * lv.off.size(), hipMemcpyDeviceToHost, ctx->stream));
            }
            HUB_HIP(hipStreamSynchronize(ctx->stream));
            levels.push_back(std::move(lv));
            if (hc[0] == 0) break;  // nothing split on this level
            if (hc[0] > S_max) { ctx->set_error("nnd_hub_tree_build: %d segments exceed the allocation", hc[0]); rc = 1; goto done; }
            // stable partition of both orderings by the winning sides
            hipLaunchKernelGGL(k_hub_sides, dim3(gridN), dim3(256), 0, ctx->stream, ord_id[cur], pos_id[cur], choice, sidebits, n, ctx->side);
            hipLaunchKernelGGL(k_hub_sides, dim3(gridN), dim3(256), 0, ctx->stream, ord_rk[cur], pos_rk[cur], choice, sidebits, n, side_rk);
            nnd_forest_stable_partition(ctx, n, ord_id[cur], pos_id[cur], ctx->side, sst[cur], sln[cur], S, nleft_a, seg_child, ord_id[1 - cur], pos_id[1 - cur]);
            nnd_forest_stable_partition(ctx, n, ord_rk[cur], pos_rk[cur], side_rk, sst[cur], sln[cur], S, nleft_b, seg_child, ord_rk[1 - cur], pos_rk[1 - cur]);
            HUB_HIP(hipGetLastError());
            cur = 1 - cur;
            depth++;
            S = hc[0];
            n_split = hc[1];
        }
    }
    {  // ---- host: pre-order FlatTree (rp_trees.py:2926-3049) ----
        std::vector<int32_t> final_ord((size_t)n);
        HUB_HIP(hipMemcpyAsync(final_ord.data(), ord_id[cur], 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        HUB_HIP(hipStreamSynchronize(ctx->stream));
        // rank of every splitting segment inside its level (children of the r-th splitter are segments 2r, 2r+1 of the next level)
        std::vector<std::vector<int32_t>> srank(levels.size());
        int64_t n_nodes = 0;
        for (size_t l = 0; l < levels.size(); l++) {
            srank[l].assign(levels[l].n_segs, -1);
            int r = 0;
            for (int s = 0; s < levels[l].n_segs; s++)
                if (levels[l].choice[s] >= 0) srank[l][s] = r++;
            n_nodes += levels[l].n_segs;
        }
        res->n_nodes = n_nodes;
        res->hyperplanes.assign((size_t)n_nodes * d, 0.0f);
        res->offsets.assign((size_t)n_nodes, 0.0f);
        res->children.assign((size_t)n_nodes * 2, -1);
        res->indices.assign((size_t)n, -1);
        res->max_leaf = leaf_size;
        // iterative pre-order walk; the stack holds (level, segment, slot to patch with this node's number)
        struct item { int l, s; int64_t patch; };
        std::vector<item> stack;
        stack.push_back({0, 0, -1});
        int64_t node_num = 0, leaf_start = 0;
        while (!stack.empty()) {
            const item it = stack.back();
            stack.pop_back();
            const hub_level &lv = levels[it.l];
            const int64_t me = node_num++;
            if (it.patch >= 0) res->children[it.patch] = (int32_t)me;
            const int c = lv.choice[it.s];
            if (c < 0) {  // leaf: children = (-leaf_start, -leaf_end), its members in id order
                const int len = lv.len[it.s];
                res->children[2 * me] = (int32_t)(-leaf_start);
                res->children[2 * me + 1] = (int32_t)(-(leaf_start + len));
                std::copy(final_ord.begin() + lv.start[it.s], final_ord.begin() + lv.start[it.s] + len, res->indices.begin() + leaf_start);
                leaf_start += len;
                if (len > res->max_leaf) res->max_leaf = len;
            } else {
                const int r = srank[it.l][it.s];
                std::copy(lv.hv.begin() + (size_t)r * d, lv.hv.begin() + (size_t)(r + 1) * d, res->hyperplanes.begin() + (size_t)me * d);
                res->offsets[me] = lv.off[(size_t)r];
                // left child is numbered next (me + 1): push right first, then left
                stack.push_back({it.l + 1, 2 * r + 1, 2 * me + 1});
                stack.push_back({it.l + 1, 2 * r, 2 * me});
            }
        }
    }
    delete ctx->hub;
    ctx->hub = res;
    res = nullptr;
done:
    (void)hipStreamSynchronize(ctx->stream);
    for (void *p : tmp)
        if (p) (void)hipFree(p);
    delete res;
    return rc;
}

int nnd_hub_tree_fetch_impl(nnd_ctx *ctx, float *hyperplanes, float *offsets, int32_t *children, int32_t *indices, int32_t *max_leaf) {
    if (!ctx->hub) { ctx->set_error("nnd_hub_tree_fetch: no tree was built"); return 1; }
    const nnd_hub_result *r = ctx->hub;
    std::copy(r->hyperplanes.begin(), r->hyperplanes.end(), hyperplanes);
    std::copy(r->offsets.begin(), r->offsets.end(), offsets);
    std::copy(r->children.begin(), r->children.end(), children);
    std::copy(r->indices.begin(), r->indices.end(), indices);
    if (max_leaf) *max_leaf = r->max_leaf;
    return 0;
}

int64_t nnd_hub_tree_nodes(const nnd_ctx *ctx) { return ctx->hub ? ctx->hub->n_nodes : 0; }
void nnd_hub_tree_free(nnd_ctx *ctx) {
    delete ctx->hub;
    ctx->hub = nullptr;
}
