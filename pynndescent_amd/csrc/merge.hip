// merge.hip -- apply buffered proposals to the k-lists; random and user-graph initialisation.
//
// k_merge replaces apply_graph_update_array (reference utils.py:661-733): the reference lets every
// thread scan ALL (p,q,d) updates and apply those whose endpoint it owns.  Here proposals were
// already routed to their target by the join (join.hip), so one wave per target row that has
// pending proposals merges them (merge.h) and re-arms the slots.  c (utils.py:725,731) is the
// number of proposals that end up in a list.
//
// k_random_init replaces init_random (pynndescent_.py:188-203); k_graph_init replaces
// initalize_heap_from_graph_indices[_and_distances] (utils.py:836-860).  Both write their
// candidates into the proposal slots and reuse k_merge.
#include "common.h"
#include "merge.h"
#include "state.h"

// One wave per 64 consecutive rows: their dirty flags are read with one load, and only the rows that HAVE pending
// proposals are visited, two at a time (slots and k-lists of both rows are fetched before the first merge starts, and
// the next pair's while the current pair is merged: the kernel is latency bound).  Past the first iterations a few
// percent of the rows are dirty; a wave per two rows spent its time finding that out.
__global__ __launch_bounds__(256) void k_merge(uint64_t *__restrict__ pbuf, uint8_t *__restrict__ pdirty, int pcap,
                                               int64_t lo, int64_t n, int k, int ks, uint32_t *__restrict__ knn_e,
                                               float *__restrict__ knn_d, float *__restrict__ th,
                                               long long *__restrict__ counters) {
    __shared__ int wacc[4];
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t base = lo + ((int64_t)blockIdx.x * 4 + w) * 64;  // [lo, n): the rows this handle owns
    int acc = 0;
    unsigned long long m = 0;
    if (base < n) m = __ballot(base + lane < n && pdirty[base + lane < n ? base + lane : lo] != 0);
    struct pair_t {
        int64_t v[2];
        uint64_t key[2];
        uint32_t e[2];
        float d[2];
    };
    auto fetch = [&](pair_t &p) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            p.v[u] = -1;
            if (m) {  // wave-uniform
                p.v[u] = base + __builtin_ctzll(m);
                m &= m - 1;
            }
            const int64_t vv = p.v[u] >= 0 ? p.v[u] : lo;
            p.key[u] = lane < pcap ? pbuf[vv * pcap + lane] : NND_EMPTY_KEY;  // pcap <= 64: one slot per lane
            p.e[u] = lane < k ? knn_e[vv * ks + lane] : NND_EMPTY_E;
            p.d[u] = lane < k ? knn_d[vv * ks + lane] : INFINITY;
        }
    };
    pair_t cur, nxt;
    if (m) fetch(cur);
    else cur.v[0] = cur.v[1] = -1;
    while (cur.v[0] >= 0) {
        if (m) fetch(nxt);
        else nxt.v[0] = nxt.v[1] = -1;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (cur.v[u] < 0) continue;  // wave-uniform
            const int64_t v = cur.v[u];
            const uint64_t mykey = cur.key[u];
            acc += nnd_merge_row_regs<1>(knn_e + v * ks, knn_d + v * ks, th + v, cur.e[u], cur.d[u], k, pcap,
                                         [&](int c, uint32_t &id, float &dc) {
                                             id = nnd_key_idx(mykey);
                                             dc = nnd_key_dist(mykey);
                                             return mykey != NND_EMPTY_KEY;
                                         });
            if (lane < pcap && mykey != NND_EMPTY_KEY) pbuf[v * pcap + lane] = NND_EMPTY_KEY;
            if (lane == 0) pdirty[v] = 0;
        }
        cur = nxt;
    }
    if (lane == 0) wacc[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long a = (long long)wacc[0] + wacc[1] + wacc[2] + wacc[3];
        if (a) nnd_count(counters, CNT_ACCEPT, a);
    }
}

// k <= 16: the same walk, FOUR dirty rows per step -- quarter-wave merges (merge.h nnd_merge_rows_q16: 16 lanes hold a
// row's sorted list, its 64 proposal slots are taken in four blocks of 16).
__global__ __launch_bounds__(256) void k_merge_q(uint64_t *__restrict__ pbuf, uint8_t *__restrict__ pdirty, int64_t lo,
                                                 int64_t n, int k, int ks, uint32_t *__restrict__ knn_e,
                                                 float *__restrict__ knn_d, float *__restrict__ th,
                                                 long long *__restrict__ counters) {
    constexpr int PCAP = 64, NB = PCAP / 16;
    __shared__ int wacc[4];
    const int lane = nnd_lane(), w = threadIdx.x >> 6, j = lane & 15, grp = lane >> 4;
    const int64_t base = lo + ((int64_t)blockIdx.x * 4 + w) * 64;  // [lo, n): the rows this handle owns
    int acc = 0;
    unsigned long long m = 0;
    if (base < n) m = __ballot(base + lane < n && pdirty[base + lane < n ? base + lane : lo] != 0);
    struct quad_t {
        int64_t v;         // this 16-lane group's row (-1: none)
        uint64_t key[NB];  // slot blk * 16 + j of that row
        uint32_t e;
        float d;
    };
    auto fetch = [&](quad_t &q) __attribute__((always_inline)) {
        // the next (up to) four dirty rows: group g takes the g-th of them
        int64_t vg = -1;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (m) {  // wave-uniform
                const int64_t v = base + __builtin_ctzll(m);
                m &= m - 1;
                if (grp == g) vg = v;
            }
        }
        q.v = vg;
        const int64_t vv = vg >= 0 ? vg : lo;
#pragma unroll
        for (int b = 0; b < NB; b++) q.key[b] = pbuf[vv * PCAP + b * 16 + j];
        q.e = j < k ? knn_e[vv * ks + j] : NND_EMPTY_E;
        q.d = j < k ? knn_d[vv * ks + j] : INFINITY;
    };
    quad_t cur, nxt;
    cur.v = nxt.v = -1;
    bool have = m != 0;
    if (have) fetch(cur);
    while (have) {
        have = m != 0;
        if (have) fetch(nxt);
        const bool on = cur.v >= 0;
        const int64_t v = on ? cur.v : lo;
        const uint32_t e0 = on ? cur.e : NND_EMPTY_E;
        const float d0 = on ? cur.d : INFINITY;
        acc += nnd_merge_rows_q16<NB>(on, knn_e + v * ks, knn_d + v * ks, th + v, e0, d0, k, PCAP,
                                      [&](int c, uint32_t &id, float &dc) {
                                          const int b = c >> 4;
                                          const uint64_t kk = b == 0 ? cur.key[0] : (b == 1 ? cur.key[1] : (b == 2 ? cur.key[2] : cur.key[3]));
                                          id = nnd_key_idx(kk);
                                          dc = nnd_key_dist(kk);
                                          return kk != NND_EMPTY_KEY;
                                      });
        if (on) {
#pragma unroll
            for (int b = 0; b < NB; b++)
                if (cur.key[b] != NND_EMPTY_KEY) pbuf[v * PCAP + b * 16 + j] = NND_EMPTY_KEY;
            if (j == 0) pdirty[v] = 0;
        }
        cur = nxt;
    }
    if (lane == 0) wacc[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long a = (long long)wacc[0] + wacc[1] + wacc[2] + wacc[3];
        if (a) nnd_count(counters, CNT_ACCEPT, a);
    }
}

// 64 < k <= NND_WIDE_K: the same walk over the dirty rows, one at a time, merged through LDS (merge.h nnd_merge_row_lds)
__global__ __launch_bounds__(256) void k_merge_wide(uint64_t *__restrict__ pbuf, uint8_t *__restrict__ pdirty, int pcap, int64_t lo, int64_t n,
                                                    int k, int ks, uint32_t *__restrict__ knn_e, float *__restrict__ knn_d,
                                                    float *__restrict__ th, long long *__restrict__ counters) {
    __shared__ uint64_t scr[4][NND_WIDE_SCRATCH_WORDS];
    __shared__ int wacc[4];
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t base = lo + ((int64_t)blockIdx.x * 4 + w) * 64;
    int acc = 0;
    unsigned long long m = 0;
    if (base < n) m = __ballot(base + lane < n && pdirty[base + lane < n ? base + lane : lo] != 0);
    while (m) {
        const int64_t v = base + __builtin_ctzll(m);
        m &= m - 1;
        const uint64_t mykey = lane < pcap ? pbuf[v * pcap + lane] : NND_EMPTY_KEY;
        acc += nnd_merge_row_lds<1>(scr[w], knn_e + v * ks, knn_d + v * ks, th + v, k, pcap, [&](int c, uint32_t &id, float &dc) {
            id = nnd_key_idx(mykey);
            dc = nnd_key_dist(mykey);
            return mykey != NND_EMPTY_KEY;
        });
        if (lane < pcap && mykey != NND_EMPTY_KEY) pbuf[v * pcap + lane] = NND_EMPTY_KEY;
        if (lane == 0) pdirty[v] = 0;
        nnd_wave_lds_sync();
    }
    if (lane == 0) wacc[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long a = (long long)wacc[0] + wacc[1] + wacc[2] + wacc[3];
        if (a) nnd_count(counters, CNT_ACCEPT, a);
    }
}

int nnd_launch_merge(nnd_ctx *ctx) {
    if (ctx->pcap > 64) { ctx->set_error("k_merge expects at most 64 proposal slots per row"); return 1; }
    const dim3 grid((unsigned)((ctx->own_hi - ctx->own_lo + 255) / 256));
    if (ctx->k > NND_MAX_K)
        hipLaunchKernelGGL(k_merge_wide, grid, dim3(256), 0, ctx->stream, ctx->pbuf, ctx->pdirty, ctx->pcap, ctx->own_lo, ctx->own_hi, ctx->k,
                           ctx->ks, ctx->knn_e, ctx->knn_d, ctx->th, ctx->counters);
    else if (ctx->k <= 16 && ctx->pcap == 64)
        hipLaunchKernelGGL(k_merge_q, grid, dim3(256), 0, ctx->stream, ctx->pbuf, ctx->pdirty, ctx->own_lo, ctx->own_hi, ctx->k, ctx->ks,
                           ctx->knn_e, ctx->knn_d, ctx->th, ctx->counters);
    else
    hipLaunchKernelGGL(k_merge, grid, dim3(256), 0, ctx->stream, ctx->pbuf,
                       ctx->pdirty, ctx->pcap, ctx->own_lo, ctx->own_hi, ctx->k, ctx->ks, ctx->knn_e, ctx->knn_d, ctx->th, ctx->counters);
    NND_HIP_CHECK(hipGetLastError());
    if (ctx->n_ranks <= 1) ctx->pbuf_clean = true;  // every row with proposals was merged and its slots re-armed
    return 0;
}

// wave-cooperative alt-space distance between prepared rows a and b (difference form for euclid)
__device__ __forceinline__ float row_dist(const float *__restrict__ xp, int dp, const float *__restrict__ nrm, int metric,
                                          int64_t a, int64_t b) {
    const float *xa = xp + a * dp, *xb = xp + b * dp;
    float s = 0.0f;
    for (int j = nnd_lane(); j < dp; j += 64) {
        float p = xa[j], q = xb[j];
        s += metric == 0 ? (p - q) * (p - q) : p * q;
    }
    s = nnd_wave_sum_f32(s);
    if (metric == 0) return nnd_clamp_dist(s);
    return nnd_gram_to_dist(1, s, nrm[a], nrm[b]);
}

// init_random (pynndescent_.py:188-203): rows that are not full get (k - filled) random candidates
__global__ __launch_bounds__(256) void k_random_init(const float *__restrict__ xp, int dp, const float *__restrict__ nrm,
                                                     int metric, int64_t n, int64_t lo, int64_t hi, int k, int ks,
                                                     const uint32_t *__restrict__ knn_e, uint32_t seed,
                                                     uint64_t *__restrict__ pbuf, uint8_t *__restrict__ pdirty, int pcap) {
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t v = lo + (int64_t)blockIdx.x * 4 + w;
    if (v >= hi) return;
    int filled = 0;
    for (int j0 = 0; j0 < k; j0 += 64) {  // (wide rows: several entries per lane)
        const uint32_t e = j0 + lane < k ? knn_e[v * ks + j0 + lane] : NND_EMPTY_E;
        filled += __popcll(__ballot(e != NND_EMPTY_E));
    }
    int todo = k - filled;  // pynndescent_.py:196
    if (todo <= 0) return;
    if (todo > 64) todo = 64;  // one pass offers at most 64 picks (the proposal slots); wide rows take a second pass
    uint32_t pick = (uint32_t)(nnd_hash3(seed, (uint32_t)v, (uint32_t)lane) % (uint64_t)n);  // pynndescent_.py:197
    bool mine = lane < todo;
    for (int j = 0; j < todo; j++) {  // the same id drawn twice is pushed once (utils.py:489-492)
        uint32_t other = __shfl(pick, j, 64);
        if (j < lane && other == pick) mine = false;
    }
    for (int j = 0; j < todo; j++) {
        uint32_t id = __shfl(pick, j, 64);
        bool on = __shfl((int)mine, j, 64);
        if (!on) continue;
        float d = row_dist(xp, dp, nrm, metric, v, (int64_t)id);
        if (lane == 0) pbuf[v * pcap + j] = nnd_make_key(d, id);
    }
    if (lane == 0) pdirty[v] = 1;
}

int nnd_launch_random_init(nnd_ctx *ctx) {
    for (int pass = 0; pass < (ctx->k + 63) / 64; pass++) {
        ctx->pbuf_clean = false;
        hipLaunchKernelGGL(k_random_init, dim3((unsigned)((ctx->own_hi - ctx->own_lo + 3) / 4)), dim3(256), 0, ctx->stream, ctx->xp,
                           ctx->dp, ctx->nrm, ctx->p.metric, ctx->n, ctx->own_lo, ctx->own_hi, ctx->k, ctx->ks, ctx->knn_e,
                           ctx->seed ^ 0x3C6EF372u ^ (uint32_t)(pass * 0x9E3779B9u), ctx->pbuf, ctx->pdirty, ctx->pcap);
        NND_HIP_CHECK(hipGetLastError());
        if (nnd_launch_merge(ctx)) return 1;
    }
    return 0;
}

// utils.py:836-860: every valid (i, j=graph[i][c]) is pushed with distance metric(x_i, x_j) or the given one
__global__ __launch_bounds__(256) void k_graph_init(const float *__restrict__ xp, int dp, const float *__restrict__ nrm,
                                                    int metric, int64_t n, int64_t lo, int64_t hi, const int32_t *__restrict__ gidx,
                                                    const float *__restrict__ gdist, int width, int col0, int wchunk,
                                                    uint64_t *__restrict__ pbuf, uint8_t *__restrict__ pdirty, int pcap) {
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t v = lo + (int64_t)blockIdx.x * 4 + w;  // the rows this handle owns; gidx / gdist hold THOSE rows (row v - lo)
    if (v >= hi) return;
    gidx += -lo * width;
    if (gdist) gdist += -lo * width;
    // columns [col0, col0 + wchunk) of the row (wchunk <= 64 = the proposal slots; wider graphs come in several launches,
    // an id repeated across launches is dropped by the merge, utils.py:489-492)
    int32_t id = lane < wchunk ? gidx[v * width + col0 + lane] : -1;
    bool mine = id >= 0 && (int64_t)id < n;
    for (int j = 0; j < wchunk; j++) {
        int32_t other = __shfl(id, j, 64);
        if (j < lane && other == id) mine = false;
    }
    for (int j = 0; j < wchunk; j++) {
        int32_t q = __shfl(id, j, 64);
        bool on = __shfl((int)mine, j, 64);
        if (!on) continue;
        float d = gdist ? nnd_clamp_dist(gdist[v * width + col0 + j]) : row_dist(xp, dp, nrm, metric, v, (int64_t)q);
        if (lane == 0) pbuf[v * pcap + j] = nnd_make_key(d, (uint32_t)q);
    }
    if (lane == 0) pdirty[v] = 1;
}

int nnd_launch_init_from_graph(nnd_ctx *ctx, const int32_t *idx_dev, const float *dist_dev, int width) {
    if (ctx->pcap < 64) { ctx->set_error("init graphs need 64 proposal slots per row"); return 1; }
    // (a shard: idx_dev / dist_dev hold the OWNED rows of the init graph, ids are global; every row of the point set is
    // prepared on every rank, so the distances to neighbours owned elsewhere are computed here like any other)
    const int64_t rows = ctx->own_hi - ctx->own_lo;
    if (rows <= 0) return 0;
    for (int col0 = 0; col0 < width; col0 += 64) {
        const int wchunk = width - col0 < 64 ? width - col0 : 64;
        ctx->pbuf_clean = false;
        hipLaunchKernelGGL(k_graph_init, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, ctx->xp, ctx->dp,
                           ctx->nrm, ctx->p.metric, ctx->n, ctx->own_lo, ctx->own_hi, idx_dev, dist_dev, width, col0, wchunk, ctx->pbuf, ctx->pdirty,
                           ctx->pcap);
        NND_HIP_CHECK(hipGetLastError());
        if (nnd_launch_merge(ctx)) return 1;
    }
    return 0;
}

// every entry of the owned rows becomes "old" (init_from_neighbor_graph pushes with flag 0, pynndescent_.py:213)
__global__ void k_clear_new_flags(uint32_t *__restrict__ knn_e, int64_t lo, int64_t hi, int ks) {
    int64_t t = lo * ks + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= hi * ks) return;
    const uint32_t e = knn_e[t];
    if (e != NND_EMPTY_E) knn_e[t] = e & NND_IDX_MASK;
}
int nnd_launch_clear_new_flags(nnd_ctx *ctx) {
    const int64_t total = (ctx->own_hi - ctx->own_lo) * ctx->ks;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_clear_new_flags, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, ctx->knn_e,
                       ctx->own_lo, ctx->own_hi, ctx->ks);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Row-sharded multi-GPU build: every handle keeps global-sized arrays indexed by
// global vertex id but OWNS the rows [own_lo, own_hi).  Proposals whose target is owned elsewhere are
// compacted into (key, target) records, shipped to the owner (host side: all-to-all-v over RCCL),
// and folded into the owner's slots.  This is the cross-process form of the reference's ownership
// test in apply_graph_update_array (utils.py:721-731).

// fold received records into this handle's slots (same hashed-slot atomicMin as the join)
__global__ void k_import_proposals(const uint64_t *__restrict__ keys, const int32_t *__restrict__ targets, int64_t count,
                                   uint64_t *__restrict__ pbuf, uint8_t *__restrict__ pdirty, int pcap, uint32_t slot_seed,
                                   const uint32_t *__restrict__ knn_e, const float *__restrict__ th, int ks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t key = keys[i];
    const int64_t t = targets[i];
    if (t < 0) return;  // hole left by a deferred vertex (k_proposal_export_regions)
    // The sender could not test the proposal against the target's neighbour list (it lives here): do it now, BEFORE the
    // record competes for a slot -- most records are "already present" (utils.py:489-492), and those are by construction
    // NEAR the target: left in, they win the slots (atomicMin keeps the nearer key) and push the genuine candidates out.
    // The threshold may have tightened since the sender read it, too (utils.py:484).
    const uint32_t src = nnd_key_idx(key);
    if (!(nnd_key_dist(key) < th[t])) return;
    const u32x4 *row = (const u32x4 *)(knn_e + t * ks);
    bool present = false;
    for (int c = 0; c < (ks >> 2); c++) {
        const u32x4 wv = row[c] & NND_IDX_MASK;
        present |= (wv.x == src) | (wv.y == src) | (wv.z == src) | (wv.w == src);
    }
    if (present) return;
    const uint32_t s = nnd_hash2(slot_seed, src) & (uint32_t)(pcap - 1);
    atomicMin((unsigned long long *)&pbuf[t * pcap + s], (unsigned long long)key);
    pdirty[t] = 1;
}

// merge rows [lo, hi) of another handle's partial k-lists (e_src/d_src: (hi-lo, ks) row blocks) into ours:
// used to combine the per-rank forests' leaf seeding at the owner
__global__ __launch_bounds__(256) void k_merge_graph_rows(int64_t lo, int64_t hi, int k, int ks, const uint32_t *__restrict__ e_src,
                                                          const float *__restrict__ d_src, uint32_t *__restrict__ knn_e,
                                                          float *__restrict__ knn_d, float *__restrict__ th) {
    const int w = threadIdx.x >> 6;
    const int64_t v = lo + (int64_t)blockIdx.x * 4 + w;
    if (v >= hi) return;
    const uint32_t *es = e_src + (v - lo) * ks;
    const float *ds = d_src + (v - lo) * ks;
    auto cf = [&](int c, uint32_t &id, float &dc) {
        const uint32_t e = es[c];
        id = e & NND_IDX_MASK;
        dc = ds[c];
        return e != NND_EMPTY_E;
    };
    if (k > NND_MAX_K) {
        __shared__ uint64_t scr[4][NND_WIDE_SCRATCH_WORDS];
        nnd_merge_row_lds<NND_WIDE_U>(scr[w], knn_e + v * ks, knn_d + v * ks, th + v, k, k, cf);
    } else {
        nnd_merge_row<1>(v, k, ks, knn_e, knn_d, th, k, cf);
    }
}

// Proposals for vertices owned elsewhere -> records in their owners' regions (regions of `cap` records per destination).
// They sit in the shard's NARROW table pbuf_r (PR = 16 slots per row: a rank sends a remote row a few proposals per
// iteration; the wide 64-slot rows of the first version made this export stream 1 KB per dirty row -- 1.8 ms per launch
// at 10 M points, most of what the exchange cost a rank).  A workgroup takes 128 consecutive rows, 32 per wave; a wave
// reads the dirty flags of its rows with one load and then its rows four at a time (lane = row-in-step * 16 + slot),
// ONCE, into registers, counting with ballots.  Space is reserved hierarchically: the wave adds the total of each run of
// rows with one destination to the workgroup's LDS counter of that destination (rows ascend, owners are contiguous
// ranges: one run, rarely two), ONE global atomic per workgroup and destination follows -- the cursors are <= 64 hot
// addresses; a global atomic per wave serialised the launch (measured: 8 ms) -- and the records are written straight
// from the registers.  A row whose records would cross the end of the region keeps them (slots and dirty flag
// untouched; the part of the reservation inside the region becomes holes, target -1): they travel with the next
// iteration's -- proposals are suggestions with exact distances, a late one is as valid as a fresh one.
template <int PR>
__global__ __launch_bounds__(256) void k_proposal_export_regions(uint64_t *__restrict__ pbuf_r, uint8_t *__restrict__ pdirty,
                                                                 int64_t n, int64_t own_lo, int64_t own_hi,
                                                                 const int64_t *__restrict__ bounds, int n_ranks, int64_t cap,
                                                                 long long *__restrict__ cursors, int32_t *__restrict__ targets,
                                                                 uint64_t *__restrict__ keys, long long *__restrict__ deferred) {
    constexpr int R = 32, RPS = 64 / PR, STEPS = R / RPS, MAXRUN = 3;  // RPS rows per step: lane = row-in-step * PR + slot
    constexpr unsigned GM = PR == 32 ? 0xFFFFFFFFu : 0xFFFFu, SM = (1u << RPS) - 1u;
    __shared__ int wg_cnt[64];
    __shared__ long long wg_base[64];
    const int lane = nnd_lane(), w = threadIdx.x >> 6, grp = lane / PR, sl = lane % PR;
    if (threadIdx.x < 64) wg_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = ((int64_t)blockIdx.x * 4 + w) * R;
    const int64_t v_l = base + lane;
    const unsigned m = base < n ? (unsigned)__ballot(lane < R && v_l < n && (v_l < own_lo || v_l >= own_hi) && pdirty[v_l < n ? v_l : 0] != 0) : 0u;
    uint64_t key[STEPS];
    unsigned long long bal[STEPS];
#pragma unroll
    for (int st = 0; st < STEPS; st++) {
        key[st] = NND_EMPTY_KEY;
        if ((m >> (RPS * st)) & SM) {  // wave-uniform: some row of this step is dirty
            const int i = RPS * st + grp;
            if ((m >> i) & 1u) key[st] = pbuf_r[(base + i) * PR + sl];
        }
        bal[st] = __ballot(key[st] != NND_EMPTY_KEY);
    }
    // runs of dirty rows with one destination (wave-uniform bookkeeping); c = live slots of a row
    int run_d[MAXRUN], run_first[MAXRUN], run_tot[MAXRUN], run_off[MAXRUN];
    int n_run = 0;
    int off[R];
    {
        int d = -1;
#pragma unroll
        for (int i = 0; i < R; i++) {
            off[i] = 0;
            if (!((m >> i) & 1u)) continue;
            const int c = __popc((unsigned)(bal[i / RPS] >> (PR * (i % RPS))) & GM);
            const int64_t v = base + i;
            if (d < 0 || v >= bounds[d + 1]) {  // a new run starts at this row
                d = d < 0 ? nnd_owner_of(bounds, n_ranks, v) : d + 1;
                while (v >= bounds[d + 1]) d++;
                if (n_run < MAXRUN) {
                    run_d[n_run] = d;
                    run_first[n_run] = i;
                    run_tot[n_run] = 0;
                }
                n_run++;
            }
            const int r = n_run - 1 < MAXRUN ? n_run - 1 : MAXRUN - 1;  // (more runs than slots: ranks of < 11 rows; per-row reservations below)
            off[i] = run_tot[r];
            run_tot[r] += c;
        }
    }
    const bool simple = n_run <= MAXRUN;
    if (simple) {
#pragma unroll
        for (int r = 0; r < MAXRUN; r++) {
            run_off[r] = 0;
            if (r < n_run && run_tot[r] > 0) {
                int o = 0;
                if (lane == 0) o = atomicAdd(&wg_cnt[run_d[r]], run_tot[r]);
                run_off[r] = __builtin_amdgcn_readfirstlane(o);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < n_ranks && wg_cnt[threadIdx.x] > 0)
        wg_base[threadIdx.x] = (long long)atomicAdd((unsigned long long *)&cursors[threadIdx.x], (unsigned long long)wg_cnt[threadIdx.x]);
    __syncthreads();
    if (!m) return;
    long long n_deferred = 0;
    if (!simple) {  // degenerate geometry (ranks of < 11 rows: more than MAXRUN destinations in one wave): every row
                    // reserves its records with a global atomic of its own -- slow, and only ever a handful of rows
#pragma unroll
        for (int st = 0; st < STEPS; st++) {
            if (!((m >> (RPS * st)) & SM)) continue;  // wave-uniform
            const int i = RPS * st + grp;
            const bool dirty = (m >> i) & 1u;
            const int64_t v = base + i;
            const unsigned gb = (unsigned)(bal[st] >> (PR * grp)) & GM;  // live slots of my row
            const int c = __popc(gb);
            const bool on = key[st] != NND_EMPTY_KEY;
            const int pre = __popc(gb & (sl ? (0xFFFFFFFFu >> (32 - sl)) : 0u));
            int d = 0, at_lo = 0, at_hi = 0;
            if (dirty && c > 0) {  // uniform inside a row's lanes
                d = nnd_owner_of(bounds, n_ranks, v);
                if (sl == 0) {
                    const unsigned long long a = atomicAdd((unsigned long long *)&cursors[d], (unsigned long long)c);
                    at_lo = (int)(uint32_t)a;
                    at_hi = (int)(uint32_t)(a >> 32);
                }
            }
            at_lo = __shfl(at_lo, grp * PR, 64);
            at_hi = __shfl(at_hi, grp * PR, 64);
            const long long at = (long long)(((unsigned long long)(uint32_t)at_hi << 32) | (uint32_t)at_lo);
            if (!dirty) continue;
            if (c > 0 && at + c > cap) {
                if (sl == 0) n_deferred += c;
                if (on && at + pre < cap) targets[(int64_t)d * cap + at + pre] = -1;
            } else {
                if (on) {
                    const int64_t idx = (int64_t)d * cap + at + pre;
                    keys[idx] = key[st];
                    targets[idx] = (int32_t)v;
                }
                pbuf_r[v * PR + sl] = NND_EMPTY_KEY;
                if (sl == 0) pdirty[v] = 0;
            }
        }
        n_deferred = nnd_wave_sum_i32((int)n_deferred);
        if (lane == 0 && n_deferred) atomicAdd((unsigned long long *)deferred, (unsigned long long)n_deferred);
        return;
    }
#pragma unroll
    for (int st = 0; st < STEPS; st++) {
        if (!((m >> (RPS * st)) & SM)) continue;  // wave-uniform
        // this lane's row of the step: i = RPS * st + grp (its bookkeeping values are picked from the uniform arrays)
        int my_off = 0, my_run = 0;
#pragma unroll
        for (int g = 0; g < RPS; g++) {
            const int i = RPS * st + g;
            int r = 0;
#pragma unroll
            for (int q = 1; q < MAXRUN; q++)
                if (q < n_run && i >= run_first[q]) r = q;
            if (grp == g) {
                my_off = off[i];
                my_run = r;
            }
        }
        const int i = RPS * st + grp;
        const bool dirty = (m >> i) & 1u;
        const int64_t v = base + i;
        const unsigned gb = (unsigned)(bal[st] >> (PR * grp)) & GM;  // live slots of my row
        const int c = __popc(gb);
        int d = run_d[0];
        int roff = run_off[0];
#pragma unroll
        for (int q = 1; q < MAXRUN; q++)
            if (my_run == q) { d = run_d[q]; roff = run_off[q]; }
        const bool on = key[st] != NND_EMPTY_KEY;
        const int pre = __popc(gb & (sl ? (0xFFFFFFFFu >> (32 - sl)) : 0u));
        if (dirty) {
            const long long at = wg_base[d] + roff + my_off;
            if (c > 0 && at + c > cap) {  // does not fit: deferred; what lies inside the region becomes holes
                if (sl == 0) n_deferred += c;
                if (on && at + pre < cap) targets[(int64_t)d * cap + at + pre] = -1;
            } else {
                if (on) {
                    const int64_t idx = (int64_t)d * cap + at + pre;
                    keys[idx] = key[st];
                    targets[idx] = (int32_t)v;
                }
                pbuf_r[v * PR + sl] = NND_EMPTY_KEY;  // the whole row: a full 128-byte line
                if (sl == 0) pdirty[v] = 0;
            }
        }
    }
    n_deferred = nnd_wave_sum_i32((int)n_deferred);
    if (lane == 0 && n_deferred) atomicAdd((unsigned long long *)deferred, (unsigned long long)n_deferred);
}

int nnd_launch_proposal_export_regions(nnd_ctx *ctx, int64_t cap, int32_t *targets_dev, uint64_t *keys_dev, long long *counts_dev) {
    if (ctx->n_ranks < 1 || !ctx->shard_bounds) { ctx->set_error("nnd_proposal_export: the handle is not a shard"); return 1; }
    NND_HIP_CHECK(hipMemsetAsync(ctx->shard_cursors, 0, sizeof(long long) * 66, ctx->stream));
    if (ctx->pbuf_r) {
        if (ctx->pcap_r != 16 && ctx->pcap_r != 32) { ctx->set_error("nnd_proposal_export expects 16 or 32 remote proposal slots per row"); return 1; }
        if (ctx->pcap_r == 16)
            hipLaunchKernelGGL(k_proposal_export_regions<16>, dim3((unsigned)((ctx->n + 127) / 128)), dim3(256), 0, ctx->stream, ctx->pbuf_r, ctx->pdirty,
                               ctx->n, ctx->own_lo, ctx->own_hi, ctx->shard_bounds, ctx->n_ranks, cap, ctx->shard_cursors, targets_dev,
                               keys_dev, ctx->shard_cursors + 65);
        else
            hipLaunchKernelGGL(k_proposal_export_regions<32>, dim3((unsigned)((ctx->n + 127) / 128)), dim3(256), 0, ctx->stream, ctx->pbuf_r, ctx->pdirty,
                               ctx->n, ctx->own_lo, ctx->own_hi, ctx->shard_bounds, ctx->n_ranks, cap, ctx->shard_cursors, targets_dev,
                               keys_dev, ctx->shard_cursors + 65);
        NND_HIP_CHECK(hipGetLastError());
    }
    // a region holds the rows that fit entirely: the host ships min(cursor, cap) records per region (deferred rows' slots
    // inside the region are holes)
    NND_HIP_CHECK(hipMemcpyAsync(counts_dev, ctx->shard_cursors, sizeof(long long) * (size_t)ctx->n_ranks, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

int nnd_launch_import_proposals(nnd_ctx *ctx, const uint64_t *keys, const int32_t *targets, int64_t count) {
    if (count <= 0) return 0;
    // same slot hash as the join of the iteration that produced the records (iter was not advanced yet)
    uint32_t slot_seed = nnd_hash2(ctx->seed ^ 0x2545F491u, (uint32_t)ctx->iter);
    ctx->pbuf_clean = false;
    hipLaunchKernelGGL(k_import_proposals, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, keys, targets, count,
                       ctx->pbuf, ctx->pdirty, ctx->pcap, slot_seed, ctx->knn_e, ctx->th, ctx->ks);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}
// k <= 16: all sources in ONE launch, four rows per wave (quarter-wave merges chained in registers): a row is loaded and
// stored once instead of once per source.  e_src / d_src: n_src blocks of (hi - lo, ks) rows, `stride` words apart.
__global__ __launch_bounds__(256) void k_merge_graph_rows_q(int64_t lo, int64_t hi, int k, int ks, int n_src, int64_t stride,
                                                            const uint32_t *__restrict__ e_src, const float *__restrict__ d_src,
                                                            uint32_t *__restrict__ knn_e, float *__restrict__ knn_d, float *__restrict__ th) {
    const int lane = nnd_lane(), w = threadIdx.x >> 6, j = lane & 15, grp = lane >> 4;
    const int64_t v = lo + ((int64_t)blockIdx.x * 4 + w) * 4 + grp;
    const bool on = v < hi;
    const int64_t vv = on ? v : lo;
    uint32_t e = (on && j < k) ? knn_e[vv * ks + j] : NND_EMPTY_E;
    float d = (on && j < k) ? knn_d[vv * ks + j] : INFINITY;
    const uint32_t e_in = e;
    const float d_in = d;
    uint32_t se = e_src[(vv - lo) * ks + j];
    float sd = d_src[(vv - lo) * ks + j];
    for (int s = 0; s < n_src; s++) {
        const uint32_t ce = se;
        const float cd = sd;
        if (s + 1 < n_src) {  // the next source's row is in flight during this merge
            se = e_src[(int64_t)(s + 1) * stride + (vv - lo) * ks + j];
            sd = d_src[(int64_t)(s + 1) * stride + (vv - lo) * ks + j];
        }
        nnd_merge_rows_q16_regs<1>(on, e, d, k, k, [&](int c, uint32_t &id, float &dc) {
            id = ce & NND_IDX_MASK;
            dc = cd;
            return ce != NND_EMPTY_E;
        });
    }
    if (on && j < k && (e != e_in || d != d_in)) {
        knn_e[v * ks + j] = e;
        knn_d[v * ks + j] = d;
        if (j == k - 1) th[v] = d;
    }
}

// n_src source blocks of (hi - lo, ks) rows each, `stride` words apart (stride = 0 with n_src = 1: one block)
int nnd_launch_merge_graph_rows(nnd_ctx *ctx, int64_t lo, int64_t hi, const uint32_t *e_src, const float *d_src, int n_src, int64_t stride) {
    if (hi <= lo || n_src <= 0) return 0;
    if (ctx->k <= 16) {
        hipLaunchKernelGGL(k_merge_graph_rows_q, dim3((unsigned)((hi - lo + 15) / 16)), dim3(256), 0, ctx->stream, lo, hi, ctx->k, ctx->ks,
                           n_src, stride, e_src, d_src, ctx->knn_e, ctx->knn_d, ctx->th);
    } else {
        for (int s = 0; s < n_src; s++)
            hipLaunchKernelGGL(k_merge_graph_rows, dim3((unsigned)((hi - lo + 3) / 4)), dim3(256), 0, ctx->stream, lo, hi, ctx->k, ctx->ks,
                               e_src + (int64_t)s * stride, d_src + (int64_t)s * stride, ctx->knn_e, ctx->knn_d, ctx->th);
    }
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

