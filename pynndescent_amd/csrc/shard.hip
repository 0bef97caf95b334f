// shard.hip -- one rank of the row-sharded multi-GPU build, and nnd_build_multi (one host thread per GPU).
//
// The reference is single-process; what it offers as a sharding rule is owner-computes over contiguous vertex ranges
// (apply_graph_update_array utils.py:709-731, new_build_candidates utils.py:259-306, init_rp_tree
// pynndescent_.py:154-185).  Here that rule crosses GPUs (include/pynnd_amd.h for the scheme).
// Everything a rank does is queued on ONE HIP stream -- kernels of the single-GPU build (state.h launchers) and the
// exchanges of comm.h -- and the host waits only for the record counts that size the next exchange (twice per
// iteration).  The update counts of all ranks ride on the first of those waits, so the stop rule costs no collective of
// its own: the decision for iteration i is taken after the (speculative, harmless) first half of iteration i + 1's
// sampling has been queued.
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "comm.h"
#include "common.h"
#include "state.h"

struct nnd_shard_s {
    nnd_comm_s *comm = nullptr;
    nnd_ctx *h = nullptr;
    nnd_params gp{};  // global parameters
    int world = 1, rank = 0, k = 0, ks = 0, t0 = 0, t1 = 0;
    int64_t n_total = 0, lo = 0, hi = 0, max_range = 0;
    int64_t bounds[NND_MAX_RANKS + 1] = {0};
    // device buffers
    float *x_full = nullptr;                     // (n_total, d) replicated point set (world > 1)
    uint32_t *recv_e = nullptr;                  // (world - 1 sources) x (n_own, ks) partial k-list rows
    float *recv_d = nullptr;
    int32_t *off_t = nullptr, *prop_t = nullptr;  // record regions, one per destination rank
    uint32_t *off_k = nullptr;   // reverse-offer records: the source vertex (the target and class are in off_t)
    uint64_t *prop_k = nullptr;  // proposal records: distance bits << 32 | candidate
    int64_t cap_o = 0, cap_p = 0;
    int32_t *in_t = nullptr;                     // received records (grow-only)
    uint64_t *in_k = nullptr;
    int64_t in_cap = 0;
    long long *cvec = nullptr;                   // (world + 4) count vector handed to comm_gather_counts
    bool gather_lists = true;                    // all-gather the neighbour ids before every join (see shard_build)
    int32_t *own_order = nullptr;                // (n_own) owned vertices in the first local tree's leaf order
    int *order_cursor = nullptr;
    // forest sharded by cell (forest_by_cell below)
    bool by_cell = false;
    int t_alloc = 0;                             // trees the handle's forest tables are sized for
    unsigned char *pack_all = nullptr;           // packed node records of ALL trees (rank-major), grow-only
    float *hf_all = nullptr;                     // their f32 hyperplanes
    int64_t nodes_cap = 0;
    int32_t *cells_i32 = nullptr;                // cell_count_all | cell_depth_all (2 x cells_cap)
    int64_t cells_cap = 0;
    int32_t *maps = nullptr;                     // small host-built tables (cell renumbering, roots, ...), grow-only
    int64_t maps_cap = 0;
    hipEvent_t ev_x = nullptr;                   // the point-set all-gather on the second channel has finished
    hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr; // per iteration: the owned rows are final / the id + threshold gather on the second channel is done
    nnd_shard_info info{};
    char err[512] = {0};
    void set_error(const char *fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, sizeof(err), fmt, ap);
        va_end(ap);
    }
};

static thread_local char g_serr2[512] = {0};
extern "C" const char *nnd_shard_last_error(nnd_shard_t s) { return s ? s->err : g_serr2; }

#define S_HIP(expr)                                                                                 \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            s->set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)
#define S_CTX(expr)                                          \
    do {                                                     \
        if ((expr) != 0) {                                   \
            s->set_error("%s: %s", #expr, s->h->err);        \
            return 1;                                        \
        }                                                    \
    } while (0)
#define S_COMM(expr)                                         \
    do {                                                     \
        if ((expr) != 0) {                                   \
            s->set_error("%s: %s", #expr, s->comm->err);     \
            return 1;                                        \
        }                                                    \
    } while (0)

// [0, G) record counts per destination | G: this rank's update count of the finished iteration | G+1: dropped | G+2: deferred
__global__ void k_pack_counts(const long long *__restrict__ cursors, const long long *__restrict__ counters_sum, int G, int with_c,
                              long long *__restrict__ out) {
    const int i = threadIdx.x;
    if (i < G) out[i] = cursors[i];
    if (i == G) out[G] = with_c ? counters_sum[CNT_ACCEPT] : 0;
    if (i == G + 1) out[G + 1] = cursors[64];
    if (i == G + 2) out[G + 2] = cursors[65];
}
__global__ void k_counters_reduce_async(const long long *__restrict__ counters, long long *__restrict__ out) {
    // one value per counter (same as prep.hip k_counters_reduce, kept device-side: no read-back)
    __shared__ long long red[256];
    for (int c = 0; c < CNT_COUNT; c++) {
        long long v = 0;
        for (int i = threadIdx.x; i < NND_CNT_STRIPES; i += 256) v += counters[(size_t)i * CNT_COUNT + c];
        red[threadIdx.x] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[c] = red[0];
        __syncthreads();
    }
}

// The owned vertices in the order they have in the first local tree's leaf sequence: vertices that are close in space are
// joined at the same time (their candidate rows, neighbour lists and proposal slots share L2 lines).  A workgroup
// compacts a run of 4096 positions; runs land in the order their workgroups reserve space -- coherence inside a run is
// what matters, and the join's results do not depend on the visiting order (proposals are atomicMin'ed).
__global__ __launch_bounds__(256) void k_compact_owned(const int32_t *__restrict__ perm, int64_t n, int64_t lo, int64_t hi,
                                                       int32_t *__restrict__ out, int *__restrict__ cursor) {
    __shared__ int wcnt[4], wbase[4];
    __shared__ int blk_base;
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * 4096;
    int cnt = 0;
    for (int it = 0; it < 16; it++) {
        const int64_t p = p0 + it * 256 + threadIdx.x;
        const int v = p < n ? perm[p] : -1;
        cnt += __popcll(__ballot(v >= lo && v < hi));
    }
    if (lane == 0) wcnt[w] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        blk_base = tot ? atomicAdd(cursor, tot) : 0;
    }
    __syncthreads();
    // second pass in the same order: iteration-major, wave-minor
    int at = blk_base;
    for (int it = 0; it < 16; it++) {
        const int64_t p = p0 + it * 256 + threadIdx.x;
        const int v = p < n ? perm[p] : -1;
        const bool on = v >= lo && v < hi;
        const unsigned long long m = __ballot(on);
        if (lane == 0) wbase[w] = __popcll(m);
        __syncthreads();
        int before = 0;
        for (int q = 0; q < w; q++) before += wbase[q];
        const int tot = wbase[0] + wbase[1] + wbase[2] + wbase[3];
        if (on) out[at + before + nnd_prefix_popc(m)] = v;
        at += tot;
        __syncthreads();
    }
}

static void shard_free(nnd_shard_s *s) {
    std::lock_guard<std::recursive_mutex> lifecycle(nnd_lifecycle_mutex());
    if (s->h) (void)hipSetDevice(s->h->p.device);
    void *ptrs[] = {s->x_full, s->recv_e, s->recv_d, s->off_t, s->off_k, s->prop_t, s->prop_k, s->in_t, s->in_k, s->cvec, s->own_order, s->order_cursor,
                    s->pack_all, s->hf_all, s->cells_i32, s->maps};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (s->ev_x) (void)hipEventDestroy(s->ev_x);
    if (s->ev_g0) (void)hipEventDestroy(s->ev_g0);
    if (s->ev_g1) (void)hipEventDestroy(s->ev_g1);
    if (s->h) (void)nnd_destroy(s->h);
    delete s;
}

// A shard whose forest was to be sharded by cell becomes one that builds its own trees (the round-3 scheme), IN PLACE: the
// handle was created for ceil(1.25 T / G) + 1 trees' worth of positions, which holds the (t1 - t0) whole trees of the split by
// tree; only the tree count, the position space and the tree seed change (every rank draws its own trees from a seed derived
// from the global one and its first tree's number, as nnd_shard_create does for a by-tree shard).  Called on EVERY rank or on
// none: at creation (the routing forest is not available on this geometry) or when the ranks have agreed that the by-cell
// forest cannot be built on this data (forest_by_cell returns 2: tops / cell tables outgrown on some rank).
static void shard_switch_to_by_tree(nnd_shard_s *s) {
    nnd_ctx *h = s->h;
    s->by_cell = false;
    s->info.forest_by_cell = 0;
    h->p.n_trees = s->t1 - s->t0;
    h->P = (int64_t)h->p.n_trees * h->n;
    h->p.tree_rng[1] = (int64_t)((uint64_t)s->gp.tree_rng[1] + 0x9E3779B97F4A7C15ull * (uint64_t)(s->t0 + 1));
    h->tree_seed = nnd_mix32((uint32_t)h->p.tree_rng[0] ^ nnd_mix32((uint32_t)h->p.tree_rng[1] + 0x9E3779B9u) ^ nnd_mix32((uint32_t)h->p.tree_rng[2] + 0x7F4A7C15u));
    h->own_order = nullptr;
    h->forest_built = false;
}

extern "C" int32_t nnd_shard_create(nnd_shard_t *out, const nnd_params *params, nnd_comm_t comm, const int64_t *shard_sizes) {
    auto fail = [&](const char *msg) {
        snprintf(g_serr2, sizeof(g_serr2), "nnd_shard_create: %s", msg);
        return 1;
    };
    if (!out || !params || !comm || !shard_sizes) return fail("null argument");
    *out = nullptr;
    (void)nnd_release_pending();  // handles destroyed earlier may still hold HBM this shard needs
    const int G = comm->world, rank = comm->rank;
    nnd_shard_s *s = new nnd_shard_s();
    s->comm = comm;
    s->gp = *params;
    s->world = G;
    s->rank = rank;
    int64_t acc = 0;
    for (int r = 0; r < G; r++) {
        if (shard_sizes[r] < 1) { delete s; return fail("every rank must own at least one row (shard sizes >= 1)"); }
        s->bounds[r] = acc;
        acc += shard_sizes[r];
        if (shard_sizes[r] > s->max_range) s->max_range = shard_sizes[r];
    }
    s->bounds[G] = acc;
    if (acc != params->n) { delete s; return fail("the shard sizes must add up to params->n (the GLOBAL point count)"); }
    s->n_total = acc;
    s->lo = s->bounds[rank];
    s->hi = s->bounds[rank + 1];
    // forest split by tree: contiguous runs (ranks beyond n_trees get none; every tree is built exactly once)
    s->t0 = (int)((int64_t)params->n_trees * rank / G);
    s->t1 = (int)((int64_t)params->n_trees * (rank + 1) / G);
    nnd_params p = *params;
    p.n_trees = s->t1 - s->t0;
    p.device = comm->device;
    // Forest sharded BY CELL (forest_by_cell): when the routing forest applies (as in nnd_create: n >= 131072, rows of
    // <= 256 floats) and the tree counts fit the exchange vectors.  The handle's forest tables are then sized for this
    // rank's share of ALL trees' cells (T n / G point-trees + 25 %) and the forest is the single-GPU forest, whatever G is
    // (global tree seeds).  Otherwise: split by tree, every rank drawing its own trees.
    const int t_loc_max = (params->n_trees + G - 1) / G;
    s->by_cell = G > 1 && params->n_trees > 0 && params->n_trees <= 1024 && t_loc_max <= 64 && params->n >= 131072 && ((params->dim + 31) & ~31) <= 256 &&
                 !(params->flags & NND_FLAG_TEST_FOREST_BY_TREE);
    if (s->by_cell) {
        const int share = (int)(((int64_t)params->n_trees * 5 + 4 * G - 1) / (4 * G)) + 1;  // ceil(1.25 T / G) + 1 trees' worth of positions
        s->t_alloc = share > t_loc_max ? share : t_loc_max;
        p.n_trees = s->t_alloc;
    } else {
        // every rank draws its own trees: the per-tree state is derived from the global one and the first tree's number
        p.tree_rng[1] = (int64_t)((uint64_t)p.tree_rng[1] + 0x9E3779B97F4A7C15ull * (uint64_t)(s->t0 + 1));
    }
    if (nnd_create_impl(&s->h, &p, s->bounds, G, rank)) {
        snprintf(g_serr2, sizeof(g_serr2), "nnd_shard_create: %s", nnd_last_global_error());
        delete s;
        return 1;
    }
    s->k = s->h->k;
    s->ks = s->h->ks;
    // (the same on every rank: a function of the parameters) the handle has no routing forest after all -- a library built with
    // the experiment knobs and NND_FOREST_WHOLE=1, or a future change of the condition in nnd_create_impl
    if (s->by_cell && !(s->h->s_m > 0 && s->h->s_stride > 0)) shard_switch_to_by_tree(s);
    const int64_t n_own = s->hi - s->lo;
    std::lock_guard<std::recursive_mutex> lifecycle(nnd_lifecycle_mutex());
    bool ok = hipSetDevice(p.device) == hipSuccess;
    ok = ok && hipMalloc((void **)&s->cvec, sizeof(long long) * (size_t)(NND_MAX_RANKS + 8)) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&s->ev_x, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&s->ev_g0, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&s->ev_g1, hipEventDisableTiming) == hipSuccess;
    if (ok && G > 1 && p.n_trees > 0) {
        ok = ok && hipMalloc((void **)&s->own_order, sizeof(int32_t) * (size_t)(n_own > 0 ? n_own : 1)) == hipSuccess;
        ok = ok && hipMalloc((void **)&s->order_cursor, sizeof(int)) == hipSuccess;
    }
    if (ok && G > 1) {
        // offers: at most every owned edge goes to ONE other rank; proposals: 32 of a row's 64 slots may travel per
        // iteration (what does not fit stays and travels next time: counted in nnd_shard_info.deferred)
        s->cap_o = n_own * s->k > 0 ? n_own * s->k : 1;
        s->cap_p = s->max_range * 32 > 64 ? s->max_range * 32 : 64;
        if (params->flags & NND_FLAG_TEST_SMALL_REGIONS) s->cap_p = s->max_range > 64 ? s->max_range : 64;  // test hook: forces deferrals
        ok = ok && hipMalloc((void **)&s->x_full, sizeof(float) * (size_t)s->n_total * params->dim) == hipSuccess;
        if (params->n_trees > 0) {
            const size_t rows = (size_t)(G - 1) * (size_t)(n_own > 0 ? n_own : 1) * s->ks;
            ok = ok && hipMalloc((void **)&s->recv_e, sizeof(uint32_t) * rows) == hipSuccess;
            ok = ok && hipMalloc((void **)&s->recv_d, sizeof(float) * rows) == hipSuccess;
        }
        ok = ok && hipMalloc((void **)&s->off_t, sizeof(int32_t) * (size_t)G * s->cap_o) == hipSuccess;
        ok = ok && hipMalloc((void **)&s->off_k, sizeof(uint32_t) * (size_t)G * s->cap_o) == hipSuccess;
        ok = ok && hipMalloc((void **)&s->prop_t, sizeof(int32_t) * (size_t)G * s->cap_p) == hipSuccess;
        ok = ok && hipMalloc((void **)&s->prop_k, sizeof(uint64_t) * (size_t)G * s->cap_p) == hipSuccess;
        s->in_cap = n_own * s->k + 1024;  // a typical iteration receives about as many offers as it sends; grows on demand
        ok = ok && hipMalloc((void **)&s->in_t, sizeof(int32_t) * (size_t)s->in_cap) == hipSuccess;
        ok = ok && hipMalloc((void **)&s->in_k, sizeof(uint64_t) * (size_t)s->in_cap) == hipSuccess;
    }
    if (!ok) {
        snprintf(g_serr2, sizeof(g_serr2), "nnd_shard_create: out of device memory (rank %d of %d, %lld owned rows of %lld)", rank, G,
                 (long long)n_own, (long long)s->n_total);
        shard_free(s);
        return 1;
    }
    s->info.n_total = s->n_total;
    s->info.own_lo = s->lo;
    s->info.own_hi = s->hi;
    s->info.world = G;
    s->info.rank = rank;
    s->info.local_trees = s->t1 - s->t0;
    s->info.forest_by_cell = s->by_cell ? 1 : 0;
    *out = s;
    return 0;
}

extern "C" int32_t nnd_shard_destroy(nnd_shard_t s) {
    if (s) shard_free(s);
    return 0;
}
extern "C" int32_t nnd_shard_get_info(nnd_shard_t s, nnd_shard_info *out) {
    if (!s || !out) return 1;
    *out = s->info;
    return 0;
}
extern "C" nnd_handle_t nnd_shard_handle(nnd_shard_t s) { return s ? s->h : nullptr; }
extern "C" int32_t nnd_shard_get_stats(nnd_shard_t s, nnd_stats *out) {
    if (!s || !out) return 1;
    *out = s->h->stats;
    return 0;
}

// the neighbour ids are all-gathered before a join only while the previous iteration changed at least this fraction of the
// n * k list entries (see shard_build, step 2)
#define NND_GATHER_MIN 0.05

// compute sections: in LOCAL serial mode (tools/rank_critical_path.py) each is timed with the GPU to itself
struct section_timer {
    nnd_shard_s *s;
    std::chrono::steady_clock::time_point t;
    bool serial, open = true;
    ~section_timer() {  // an error return inside the section must not keep the serial-mode token
        if (open) comm_compute_end(s->comm, s->h->stream);
    }
    explicit section_timer(nnd_shard_s *s_) : s(s_) {
        serial = s->comm->kind == NND_COMM_LOCAL && s->comm->grp->serial;
        comm_compute_begin(s->comm);
        if (serial) (void)hipStreamSynchronize(s->h->stream);  // the copies of the exchange before it are not this section's
        t = std::chrono::steady_clock::now();
    }
    void end() {
        open = false;
        comm_compute_end(s->comm, s->h->stream);  // serial mode: drains the stream, then releases the GPU
        if (serial && s->info.n_sections < 256) {
            const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t).count();
            s->info.section_ms[s->info.n_sections] = ms;
            s->info.section_bytes[s->info.n_sections] = 0;
            s->info.n_sections++;
        }
    }
};
static void note_bytes(nnd_shard_s *s, int64_t before) {  // payload of the exchange that followed the last section
    if (s->info.n_sections > 0 && s->info.n_sections <= 256) s->info.section_bytes[s->info.n_sections - 1] += s->comm->bytes_sent - before;
}

static int grow_inbox(nnd_shard_s *s, int64_t need) {
    if (need <= s->in_cap) return 0;
    S_COMM(comm_wait(s->comm, s->h->stream, "inbox growth"));
    if (s->in_t) S_HIP(hipFree(s->in_t));
    if (s->in_k) S_HIP(hipFree(s->in_k));
    s->in_t = nullptr;
    s->in_k = nullptr;
    s->in_cap = need + need / 4;
    S_HIP(hipMalloc((void **)&s->in_t, sizeof(int32_t) * (size_t)s->in_cap));
    S_HIP(hipMalloc((void **)&s->in_k, sizeof(uint64_t) * (size_t)s->in_cap));
    return 0;
}

// all-to-all-v of the per-destination record regions [d * cap, d * cap + cnt[d]) of (targets, keys); the records of the
// OTHER ranks land back to back in (in_t, in_k).  matrix: the gathered count vectors (row = sender), nv words per row;
// caps[r]: region capacity of sender r (a cursor beyond it means records that were not written: dropped / deferred).
static int exchange_records(nnd_shard_s *s, int32_t *reg_t, void *reg_k, int key_bytes, const int64_t *caps, const long long *matrix, int nv,
                            int64_t *n_in_out, int64_t *n_sent_out) {
    const int G = s->world, me = s->rank;
    size_t soff[NND_MAX_RANKS], scnt[NND_MAX_RANKS], roff[NND_MAX_RANKS], rcnt[NND_MAX_RANKS];
    int64_t n_in = 0, n_sent = 0;
    for (int r = 0; r < G; r++) {
        long long sc = matrix[(size_t)me * nv + r], rc = matrix[(size_t)r * nv + me];
        if (sc > caps[me]) sc = caps[me];
        if (rc > caps[r]) rc = caps[r];
        if (r == me) sc = rc = 0;  // own records were applied locally
        soff[r] = (size_t)r * (size_t)caps[me];
        scnt[r] = (size_t)sc;
        roff[r] = (size_t)n_in;
        rcnt[r] = (size_t)rc;
        n_in += rc;
        n_sent += sc;
    }
    if (grow_inbox(s, n_in)) return 1;
    void *sb[2] = {reg_t, reg_k}, *rb[2] = {s->in_t, s->in_k};
    const int eb[2] = {4, key_bytes};  // the inbox's second array is sized for 8-byte keys; 4-byte ones lie packed at its front
    S_COMM(comm_alltoallv(s->comm, s->h->stream, 2, sb, rb, eb, soff, scnt, roff, rcnt));
    *n_in_out = n_in;
    *n_sent_out = n_sent;
    return 0;
}


static int shard_wait_hook(void *user, hipStream_t stream) {
    nnd_shard_s *s = (nnd_shard_s *)user;
    return comm_wait(s->comm, stream, "build");
}

template <typename T>
static int grow_dev(nnd_shard_s *s, T **buf, int64_t *cap, int64_t need) {
    if (need <= *cap) return 0;
    S_COMM(comm_wait(s->comm, s->h->stream, "buffer growth"));
    if (*buf) S_HIP(hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    S_HIP(hipMalloc((void **)buf, sizeof(T) * (size_t)(need + need / 8 + 64)));
    *cap = need + need / 8 + 64;
    return 0;
}

// owned vertices in the order of their cells in tree 0: the records of tree 0 lie cell-sorted in G segments (one per owner)
__global__ void k_order_from_records(const int32_t *__restrict__ rec_row, const int32_t *__restrict__ cell_scan, const int32_t *__restrict__ seg_cells /* (G, 2): first / end cell */,
                                     int G, int32_t n_cells_all, const int32_t *__restrict__ total, int32_t *__restrict__ out) {
    __shared__ int seg_lo[NND_MAX_RANKS], seg_hi[NND_MAX_RANKS], seg_out[NND_MAX_RANKS + 1];
    if (threadIdx.x == 0) {
        int run = 0;
        for (int q = 0; q < G; q++) {
            const int c0 = seg_cells[2 * q], c1 = seg_cells[2 * q + 1];
            seg_lo[q] = c0 < n_cells_all ? cell_scan[c0] : total[0];
            seg_hi[q] = c1 < n_cells_all ? cell_scan[c1] : total[0];
            seg_out[q] = run;
            run += seg_hi[q] - seg_lo[q];
        }
        seg_out[G] = run;
    }
    __syncthreads();
    for (int q = 0; q < G; q++) {
        const int len = seg_hi[q] - seg_lo[q];
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) out[seg_out[q] + i] = rec_row[seg_lo[q] + i];
    }
}

// ---- section 0 of a sharded build with the forest sharded by cell.  On return the handle holds: x_orig = the replicated
// point set, every row prepared, this rank's leaves (cells of ALL trees, 1 / G of each) and their leaf tables.
static int forest_by_cell(nnd_shard_s *s, const float *x_local_dev, hipEvent_t ev_x /* x_full complete (recorded on the second channel), or nullptr */) {
    nnd_ctx *h = s->h;
    nnd_comm_s *c = s->comm;
    const int G = s->world, me = s->rank, T = s->gp.n_trees, d = h->d, dp = h->dp;
    const int64_t n = s->n_total, lo = s->lo, hi = s->hi, n_own = hi - lo;
    hipStream_t st = h->stream;
    const int T_loc = s->t1 - s->t0;
    auto t0_of = [&](int r) { return (int)((int64_t)T * r / G); };
    // ---- (a) column means from every rank's own sample rows, own rows prepared, own sample members gathered ----
    int64_t n_s, mstride;
    nnd_prep_mean_geometry(n, &n_s, &mstride);
    std::vector<int64_t> rlo(G + 1);  // sample member r = row r * mstride: rank q holds the members [rlo[q], rlo[q + 1])
    for (int q = 0; q <= G; q++) {
        int64_t r = (s->bounds[q] + mstride - 1) / mstride;
        rlo[q] = r < n_s ? r : n_s;
    }
    rlo[G] = n_s;
    std::vector<int> pblocks(G + 1, 0);  // partial blocks of every rank (rank-major in the gathered table)
    for (int q = 0; q < G; q++) pblocks[q + 1] = pblocks[q] + nnd_prep_partial_blocks(rlo[q + 1] - rlo[q]);
    const float *x_biased = x_local_dev - (size_t)lo * d;  // row i of the whole set at x_biased + i * d (only own rows are touched)
    {
        section_timer sec(s);
        const int tp = t_begin(h);
        double *partial = nullptr;
        if (h->p.metric == 0) {
            partial = nnd_prep_partial_buffer(h, (size_t)pblocks[G] * (d + 1));
            if (!partial) { s->set_error("%s", h->err); return 1; }
            (void)nnd_prep_mean_partial(h, x_local_dev, lo, rlo[me], rlo[me + 1], mstride, partial + (size_t)pblocks[me] * (d + 1));
        }
        t_end(h, tp, &h->stats.ms_prep, false);
        sec.end();
        if (h->p.metric == 0) {  // all-gather of the partial sums, in place
            size_t soff[NND_MAX_RANKS], scnt[NND_MAX_RANKS], roff[NND_MAX_RANKS], rcnt[NND_MAX_RANKS];
            for (int r = 0; r < G; r++) {
                soff[r] = (size_t)pblocks[me] * (d + 1);
                scnt[r] = (size_t)(pblocks[me + 1] - pblocks[me]) * (d + 1);
                roff[r] = (size_t)pblocks[r] * (d + 1);
                rcnt[r] = (size_t)(pblocks[r + 1] - pblocks[r]) * (d + 1);
            }
            void *sb[1] = {partial}, *rb[1] = {partial};
            const int eb[1] = {8};
            const int64_t b0 = c->bytes_sent;
            S_COMM(comm_alltoallv(c, st, 1, sb, rb, eb, soff, scnt, roff, rcnt));
            note_bytes(s, b0);
        }
        section_timer sec2(s);
        const int tp2 = t_begin(h);
        S_CTX(nnd_prep_mean_finish(h, partial, pblocks[G], n_s));
        h->x_orig = x_biased;
        h->x_owned = false;
        h->x_valid = true;
        S_CTX(nnd_prep_rows(h, x_biased, lo, hi, true));
        S_CTX(nnd_launch_reset_graph(h));
        t_end(h, tp2, &h->stats.ms_prep, true);
        const int tf = t_begin(h);
        // the GLOBAL sample: member j = row j * s_stride + jitter(j); rank q holds the members [jlo[q], jlo[q + 1])
        sec2.end();
        (void)tf;
    }
    const int64_t M = h->s_m, sstride = h->s_stride;
    std::vector<int64_t> jlo(G + 1);
    auto sample_row = [&](int64_t j) { return j * sstride + (int64_t)(nnd_hash2(h->tree_seed ^ 0x7F4A7C15u, (uint32_t)j) % (uint32_t)sstride); };
    for (int q = 0; q <= G; q++) {
        int64_t j = s->bounds[q] / sstride;
        if (j > M) j = M;
        while (j > 0 && sample_row(j - 1) >= s->bounds[q]) j--;
        while (j < M && sample_row(j) < s->bounds[q]) j++;
        jlo[q] = j;
    }
    jlo[G] = M;
    {
        section_timer sec(s);
        const int tf = t_begin(h);
        S_CTX(nnd_forest_sample_gather(h, jlo[me], jlo[me + 1]));
        t_end(h, tf, &h->stats.ms_forest, false);
        sec.end();
    }
    {  // all-gather of the sample rows (f32, half precision, norms), in place
        size_t soff[NND_MAX_RANKS], scnt[NND_MAX_RANKS], roff[NND_MAX_RANKS], rcnt[NND_MAX_RANKS];
        for (int r = 0; r < G; r++) {
            soff[r] = (size_t)jlo[me];
            scnt[r] = (size_t)(jlo[me + 1] - jlo[me]);
            roff[r] = (size_t)jlo[r];
            rcnt[r] = (size_t)(jlo[r + 1] - jlo[r]);
        }
        void *sb[3] = {h->xs, h->xsh, h->nr2s}, *rb[3] = {h->xs, h->xsh, h->nr2s};
        const int eb[3] = {4 * dp, 2 * dp, 8};
        const int64_t b0 = c->bytes_sent;
        S_COMM(comm_alltoallv(c, st, 3, sb, rb, eb, soff, scnt, roff, rcnt));
        note_bytes(s, b0);
    }
    // ---- (b) tops of this rank's trees, counts to everybody ----
    nnd_tops_info ti;
    bool tops_over = false;
    {
        section_timer sec(s);
        const int tf = t_begin(h);
        const int rc_t = nnd_forest_tops(h, T_loc, s->t0, &ti);
        if (rc_t == 1) { s->set_error("%s", h->err); return 1; }
        // rc 2: this rank's recorded tops outgrew their tables (duplicate-heavy / degenerate data).  One GPU falls back to the
        // whole-set passes; here the word rides on the count vector every rank reads anyway and ALL ranks fall back together
        tops_over = rc_t == 2 || (NND_TEST_FALLBACK_AT(s->gp.flags) == 1 && me == G - 1);
        if (tops_over) ti = nnd_tops_info();
        t_end(h, tf, &h->stats.ms_forest, true);
        sec.end();
    }
    h->stats.tree_levels = ti.levels;
    const int nv = 3 + 64;
    std::vector<long long> cv(nv, 0), matrix((size_t)G * nv);
    cv[0] = ti.n_packed;
    cv[1] = ti.n_cells;
    cv[2] = tops_over ? 1 : 0;
    for (int t = 0; t < T_loc; t++) cv[3 + t] = ti.tree_cells[t];
    S_HIP(hipMemcpyAsync(s->cvec, cv.data(), sizeof(long long) * nv, hipMemcpyHostToDevice, st));
    S_COMM(comm_gather_counts(c, st, s->cvec, nv, matrix.data()));
    t_flush(h);
    for (int r = 0; r < G; r++)
        if (matrix[(size_t)r * nv + 2]) return 2;  // (the same matrix on every rank: everybody leaves here)
    // ---- (c) the numbering of nodes and cells over the whole build (identical on every rank) ----
    std::vector<int64_t> node_base(G + 1, 0);
    std::vector<int32_t> C(T), lbase(G + 1, 0);  // cells per tree; first cell of every rank in tops-major order
    for (int r = 0; r < G; r++) {
        node_base[r + 1] = node_base[r] + matrix[(size_t)r * nv + 0];
        lbase[r + 1] = lbase[r] + (int32_t)matrix[(size_t)r * nv + 1];
        for (int t = t0_of(r); t < t0_of(r + 1); t++) C[t] = (int32_t)matrix[(size_t)r * nv + 3 + (t - t0_of(r))];
    }
    const int64_t nodes_all = node_base[G];
    const int32_t cells_all = lbase[G];
    if (nodes_all >= (int64_t)0x3FFFFFF0) { s->set_error("sharded forest: %lld recorded nodes exceed the 2^30 node ids of the routing passes", (long long)nodes_all); return 1; }
    auto fj = [&](int t, int q) { return (int32_t)((int64_t)C[t] * q / G); };  // rank q finishes the cells [fj(t, q), fj(t, q + 1)) of tree t
    std::vector<int32_t> own_base(G + 1, 0);
    std::vector<std::vector<int32_t>> tree_off(G, std::vector<int32_t>(T + 1, 0));  // owner-major numbering: rank, then tree, then cell
    for (int q = 0; q < G; q++) {
        for (int t = 0; t < T; t++) tree_off[q][t + 1] = tree_off[q][t] + (fj(t, q + 1) - fj(t, q));
        own_base[q + 1] = own_base[q] + tree_off[q][T];
    }
    const int32_t cells_own = own_base[me + 1] - own_base[me];
    // host-built tables, one upload: [cell_gid (own tops' cells) | depth_map (own cells -> tops-major) | roots (T) | dest_cell (G + 1) |
    //                               tree_first_cell (T + 1) | order segments (2 G)]
    std::vector<int32_t> tab;
    const size_t o_gid = 0;
    tab.resize((size_t)ti.n_cells);
    {
        int32_t lc = 0;
        for (int t = s->t0; t < s->t1; t++) {
            int q = 0;
            for (int32_t j = 0; j < C[t]; j++, lc++) {
                while (q + 1 < G && j >= fj(t, q + 1)) q++;
                tab[o_gid + lc] = own_base[q] + tree_off[q][t] + (j - fj(t, q));
            }
        }
    }
    const size_t o_dmap = tab.size();
    tab.resize(o_dmap + (size_t)cells_own);
    {
        int32_t oc = 0;
        for (int t = 0; t < T; t++) {
            int r = 0;
            while (t >= t0_of(r + 1)) r++;  // the rank that built tree t's top
            int32_t cb = lbase[r];
            for (int u = t0_of(r); u < t; u++) cb += C[u];
            for (int32_t j = fj(t, me); j < fj(t, me + 1); j++) tab[o_dmap + oc++] = cb + j;
        }
    }
    const size_t o_roots = tab.size();
    for (int t = 0; t < T; t++) {
        int r = 0;
        while (t >= t0_of(r + 1)) r++;
        tab.push_back((int32_t)(node_base[r] + (t - t0_of(r))));
    }
    const size_t o_dest = tab.size();
    for (int q = 0; q <= G; q++) tab.push_back(own_base[q]);
    const size_t o_tfc = tab.size();
    for (int t = 0; t <= T; t++) tab.push_back(tree_off[me][t]);
    const size_t o_oseg = tab.size();
    for (int q = 0; q < G; q++) {
        tab.push_back(own_base[q] + tree_off[q][0]);
        tab.push_back(own_base[q] + tree_off[q][1]);
    }
    if (grow_dev(s, &s->maps, &s->maps_cap, (int64_t)tab.size())) return 1;
    int32_t cells_own_max = 0;
    for (int q = 0; q < G; q++) cells_own_max = own_base[q + 1] - own_base[q] > cells_own_max ? own_base[q + 1] - own_base[q] : cells_own_max;
    const int64_t cpad = (cells_all + 3) & ~3;
    // [counts of this rank's rows per cell (scanned in place) | depths, tops-major | the counts again (sent to the owners) | G count vectors received]
    if (grow_dev(s, &s->cells_i32, &s->cells_cap, 3 * cpad + (int64_t)G * cells_own_max + 16)) return 1;
    if (nodes_all > s->nodes_cap) {
        S_COMM(comm_wait(c, st, "node tables"));
        if (s->pack_all) S_HIP(hipFree(s->pack_all));
        if (s->hf_all) S_HIP(hipFree(s->hf_all));
        s->pack_all = nullptr;
        s->hf_all = nullptr;
        s->nodes_cap = 0;
        const int64_t cap = nodes_all + nodes_all / 8 + 64;
        S_HIP(hipMalloc((void **)&s->pack_all, (size_t)cap * (2 * dp + 16)));
        S_HIP(hipMalloc((void **)&s->hf_all, sizeof(float) * (size_t)cap * (dp + 4)));
        s->nodes_cap = cap;
    }
    S_HIP(hipMemcpyAsync(s->maps, tab.data(), sizeof(int32_t) * tab.size(), hipMemcpyHostToDevice, st));
    int32_t *cell_count_all = s->cells_i32, *cell_depth_all = s->cells_i32 + cpad, *count_copy = s->cells_i32 + 2 * cpad, *cnt_recv = s->cells_i32 + 3 * cpad;
    const int rec = 2 * dp + 16, hs = dp + 4;
    {
        section_timer sec(s);
        const int tf = t_begin(h);
        S_CTX(nnd_forest_tops_pack(h, &ti, node_base[me], s->maps + o_gid, s->pack_all + (size_t)node_base[me] * rec, s->hf_all + (size_t)node_base[me] * hs));
        if (ti.n_cells > 0)
            S_HIP(hipMemcpyAsync(cell_depth_all + lbase[me], h->cell_depth, sizeof(int32_t) * (size_t)ti.n_cells, hipMemcpyDeviceToDevice, st));
        t_end(h, tf, &h->stats.ms_forest, true);
        sec.end();
    }
    {  // all-gather of the packed records, the f32 hyperplanes and the cell depths, in place
        size_t soff[NND_MAX_RANKS], scnt[NND_MAX_RANKS], roff[NND_MAX_RANKS], rcnt[NND_MAX_RANKS];
        for (int r = 0; r < G; r++) {
            soff[r] = (size_t)node_base[me];
            scnt[r] = (size_t)(node_base[me + 1] - node_base[me]);
            roff[r] = (size_t)node_base[r];
            rcnt[r] = (size_t)(node_base[r + 1] - node_base[r]);
        }
        void *sb[2] = {s->pack_all, s->hf_all}, *rb[2] = {s->pack_all, s->hf_all};
        const int eb[2] = {rec, 4 * hs};
        const int64_t b0 = c->bytes_sent;
        S_COMM(comm_alltoallv(c, st, 2, sb, rb, eb, soff, scnt, roff, rcnt));
        for (int r = 0; r < G; r++) {
            soff[r] = (size_t)lbase[me];
            scnt[r] = (size_t)(lbase[me + 1] - lbase[me]);
            roff[r] = (size_t)lbase[r];
            rcnt[r] = (size_t)(lbase[r + 1] - lbase[r]);
        }
        void *sb2[1] = {cell_depth_all}, *rb2[1] = {cell_depth_all};
        const int eb2[1] = {4};
        S_COMM(comm_alltoallv(c, st, 1, sb2, rb2, eb2, soff, scnt, roff, rcnt));
        note_bytes(s, b0);
    }
    // ---- (d) this rank's rows through ALL trees; (cell, row) records sorted by cell = by owner ----
    int32_t *rec_cell = h->inv, *rec_row = h->scan_out;  // free once the routing passes are over (they were its scratch)
    {
        section_timer sec(s);
        const int tf = t_begin(h);
        S_HIP(hipMemsetAsync(cell_count_all, 0, sizeof(int32_t) * (size_t)cells_all, st));
        S_CTX(nnd_forest_route_rows(h, s->pack_all, s->hf_all, s->maps + o_roots, T, lo, n_own, cells_all, cell_count_all));
        // the records overwrite the routing scratch: cell_of / rank_of (pos_seg) are read, inv / scan_out written
        S_CTX(nnd_forest_route_records(h, T, lo, n_own, cells_all, cell_count_all, count_copy, s->maps + o_dest, G, rec_cell, rec_row, s->cvec));
        if (s->own_order && n_own > 0) {  // the owned vertices in the order of their tree-0 cells (spatially coherent visiting order)
            hipLaunchKernelGGL(k_order_from_records, dim3(256), dim3(256), 0, st, rec_row, cell_count_all, s->maps + o_oseg, G, cells_all,
                               (const int32_t *)(h->counters + CNT_SCRATCH), s->own_order);
            h->own_order = s->own_order;
            h->forest_gen++;  // the buffer is rewritten in place: the sampler's inverse of the visiting order (rv_pos) is stale
        }
        t_end(h, tf, &h->stats.ms_forest, true);
        sec.end();
    }
    std::vector<long long> m2((size_t)G * (G + 1));
    S_COMM(comm_gather_counts(c, st, s->cvec, G + 1, m2.data()));  // host wait: record offsets per destination, of every rank
    t_flush(h);
    // Does every owner's share fit its forest tables (T n / G point-trees + 25 % + one tree)?  A skewed share (most rows in the
    // cells of one owner) is not an error of the build either: every rank evaluates every rank's share from the same matrix, and
    // all of them fall back to the split by tree together.  (The tables have the same size on every rank.)
    for (int q = 0; q < G; q++) {
        int64_t in_q = 0;
        for (int r = 0; r < G; r++) in_q += (int64_t)(m2[(size_t)r * (G + 1) + q + 1] - m2[(size_t)r * (G + 1) + q]);
        const int64_t cells_q = own_base[q + 1] - own_base[q];
        if (in_q > h->P || cells_q > h->cell_cap || cells_q + in_q / (h->p.leaf_size + 1) > h->max_segs) return 2;
        if (NND_TEST_FALLBACK_AT(s->gp.flags) == 2) return 2;
    }
    int64_t n_in = 0;
    {
        size_t soff[NND_MAX_RANKS], scnt[NND_MAX_RANKS], roff[NND_MAX_RANKS], rcnt[NND_MAX_RANKS];
        for (int r = 0; r < G; r++) {
            soff[r] = (size_t)m2[(size_t)me * (G + 1) + r];
            scnt[r] = (size_t)(m2[(size_t)me * (G + 1) + r + 1] - m2[(size_t)me * (G + 1) + r]);
            roff[r] = (size_t)n_in;
            rcnt[r] = (size_t)(m2[(size_t)r * (G + 1) + me + 1] - m2[(size_t)r * (G + 1) + me]);
            n_in += (int64_t)rcnt[r];
        }
        if (grow_inbox(s, n_in)) return 1;
        void *sb[2] = {rec_cell, rec_row}, *rb[2] = {s->in_t, s->in_k};
        const int eb[2] = {4, 4};
        const int64_t b0 = c->bytes_sent;
        S_COMM(comm_alltoallv(c, st, 2, sb, rb, eb, soff, scnt, roff, rcnt));
        // ... and every rank's counts of the destination's cells (the owner sums them: no counting pass over the records)
        for (int r = 0; r < G; r++) {
            soff[r] = (size_t)own_base[r];
            scnt[r] = (size_t)(own_base[r + 1] - own_base[r]);
            roff[r] = (size_t)r * cells_own;
            rcnt[r] = (size_t)cells_own;
        }
        void *sb2[1] = {count_copy}, *rb2[1] = {cnt_recv};
        const int eb2[1] = {4};
        S_COMM(comm_alltoallv(c, st, 1, sb2, rb2, eb2, soff, scnt, roff, rcnt));
        note_bytes(s, b0);
    }
    s->info.n_sections_overlap = s->info.n_sections;  // everything above needed only this rank's rows and the sample
    // ---- (e) the replicated point set is needed from here on: the other ranks' rows are prepared ----
    if (ev_x) S_HIP(hipStreamWaitEvent(st, ev_x, 0));
    {
        section_timer sec(s);
        const int tp = t_begin(h);
        h->x_orig = s->x_full;
        S_CTX(nnd_prep_rows(h, s->x_full, 0, lo, false));
        S_CTX(nnd_prep_rows(h, s->x_full, hi, n, false));
        S_HIP(hipMemcpyAsync(h->h_pin + 63, h->counters_sum + CNT_SCRATCH, sizeof(long long), hipMemcpyDeviceToHost, st));  // non-finite flag (nnd_data_nonfinite)
        t_end(h, tp, &h->stats.ms_prep, true);
        // ---- (f) the cells this rank owns: rows placed, cells finished down to leaves, leaf tables ----
        const int tf = t_begin(h);
        const int rc_f = nnd_forest_finish_owned(h, s->in_t, (const int32_t *)s->in_k, n_in, own_base[me], cells_own, cnt_recv, G, cell_depth_all,
                                                 s->maps + o_dmap, s->maps + o_tfc, T);
        if (rc_f == 1) {
            s->set_error("nnd_forest_finish_owned: %s", h->err);
            return 1;
        }
        t_end(h, tf, &h->stats.ms_forest, true);
        // too many over-long cells on SOME rank (rc 2; one GPU takes the whole-set passes then): one word per rank goes round --
        // the leaf tables' read-back has just synchronised, the wait costs an exchange of G words
        sec.end();
        std::vector<long long> agree((size_t)G, 0);
        const long long mine = (rc_f == 2 || (NND_TEST_FALLBACK_AT(s->gp.flags) == 3 && me == 0)) ? 1 : 0;
        S_HIP(hipMemcpyAsync(s->cvec, &mine, sizeof(long long), hipMemcpyHostToDevice, st));
        S_COMM(comm_gather_counts(c, st, s->cvec, 1, agree.data()));
        t_flush(h);
        for (int r = 0; r < G; r++)
            if (agree[r]) return 2;
    }
    s->info.forest_positions = n_in;
    return 0;
}

// init_idx_dev (nullable): the owned rows of an init graph (n_own, init_width), global ids -- the warm start of
// NNDescent(init_graph=...) (pynndescent_.py:1225-1242, utils.py:836-860): no forest, no random fill
static int shard_build(nnd_shard_s *s, const float *x_local_dev, void *x_stream, int32_t *out_idx_dev, float *out_dist_dev,
                       const int32_t *init_idx_dev = nullptr, const float *init_dist_dev = nullptr, int init_width = 0, int init_mode = 0) {
    // init_mode 1: NNDescent(init_graph=...) -- entries enter as NEW, no forest (n_trees = 0), no random fill;
    // init_mode 2: NNDescent.update() (pynndescent_.py:2498-2535) -- the previous graph's entries enter as OLD (init_from_neighbor_graph,
    //              pynndescent_.py:206-214) BEFORE the fresh forest's leaves are joined, no random fill
    auto seed_old = [&]() -> int {
        if (init_mode != 2 || !init_idx_dev) return 0;
        if (nnd_launch_init_from_graph(s->h, init_idx_dev, init_dist_dev, init_width) || nnd_launch_clear_new_flags(s->h)) { s->set_error("%s", s->h->err); return 1; }
        s->h->all_new = false;
        return 0;
    };
    nnd_ctx *h = s->h;
    nnd_comm_s *c = s->comm;
    const int G = s->world, me = s->rank, nv = G + 3;
    const int64_t n_own = s->hi - s->lo;
    S_HIP(hipSetDevice(h->p.device));
    hipStream_t st = h->stream;
    const auto wall0 = std::chrono::steady_clock::now();
    c->bytes_sent = 0;
    s->info.iters = 0;
    s->info.n_sections = 0;
    s->info.dropped_offers = 0;
    memset(s->info.gather_bytes, 0, sizeof(s->info.gather_bytes));
    for (int i = 0; i < 64; i++) s->info.gather_section[i] = -1;
    memset(s->info.c, 0, sizeof(s->info.c));
    memset(s->info.offer_records, 0, sizeof(s->info.offer_records));
    memset(s->info.proposal_records, 0, sizeof(s->info.proposal_records));
    memset(s->info.deferred, 0, sizeof(s->info.deferred));
    memset(&h->stats, 0, sizeof(h->stats));
    if (x_stream && (hipStream_t)x_stream != st) {  // the caller's rows were produced on another stream
        S_HIP(hipEventRecord(h->ev0, (hipStream_t)x_stream));
        S_HIP(hipStreamWaitEvent(st, h->ev0, 0));
    }
    hipEvent_t e0 = h->ev0, e1 = h->ev1;

    // ---- replicate the point set once (all-gather over xGMI): candidate vectors never travel again.  With the forest
    //      sharded by cell the transfer runs on the SECOND channel (its own communicator and stream) while this rank
    //      prepares and routes its own rows; otherwise it is waited for here ----
    h->wait_hook = shard_wait_hook;
    h->wait_user = s;
    const float *x_use = x_local_dev;
    const bool serial = c->kind == NND_COMM_LOCAL && c->grp->serial;
    bool x_async = false;
    if (G > 1) {
        size_t soff[NND_MAX_RANKS], scnt[NND_MAX_RANKS], roff[NND_MAX_RANKS], rcnt[NND_MAX_RANKS];
        for (int r = 0; r < G; r++) {
            soff[r] = 0;
            scnt[r] = (size_t)n_own * h->d;
            roff[r] = (size_t)s->bounds[r] * h->d;
            rcnt[r] = (size_t)(s->bounds[r + 1] - s->bounds[r]) * h->d;
        }
        void *sb[1] = {(void *)x_local_dev}, *rb[1] = {s->x_full};
        const int eb[1] = {4};
        // (serial mode measures compute sections with the GPU to itself: the copy is not left running beside them; the
        // critical-path tool prices the overlap from n_sections_overlap)
        x_async = s->by_cell && c->aux && c->aux_stream && !serial;
        nnd_comm_s *xc = x_async ? c->aux : c;
        hipStream_t xs = x_async ? c->aux_stream : st;
        if (x_async) {  // the second stream starts behind whatever produced the rows
            S_HIP(hipEventRecord(e0, st));
            S_HIP(hipStreamWaitEvent(xs, e0, 0));
        }
        S_HIP(hipEventRecord(e0, xs));
        const int64_t b0 = xc->bytes_sent;
        if (comm_alltoallv(xc, xs, 1, sb, rb, eb, soff, scnt, roff, rcnt)) { s->set_error("point-set all-gather: %s", xc->err); return 1; }
        if (xc != c) c->bytes_sent += xc->bytes_sent - b0;
        S_HIP(hipEventRecord(e1, xs));
        if (x_async) {
            S_HIP(hipEventRecord(s->ev_x, xs));
        } else {
            S_COMM(comm_wait(c, st, "point-set all-gather"));
            (void)hipEventElapsedTime(&s->info.ms_allgather, e0, e1);
        }
        x_use = s->x_full;
    }
    bool seeded = false;
    if (s->by_cell) {
        const int rc_c = forest_by_cell(s, x_local_dev, x_async ? s->ev_x : nullptr);
        if (rc_c == 1) return 1;
        if (rc_c == 2) {
            // The ranks have AGREED (forest_by_cell: the word travels with counts they exchange anyway) that this forest cannot be
            // sharded by cell on this data -- recorded tops or an owner's share outgrew their tables, too many over-long cells:
            // conditions one GPU recovers from by its whole-set passes.  All of them switch to the split by tree, for this
            // build and the later ones of this shard, and start over from the replicated point set.
            shard_switch_to_by_tree(s);
            if (x_async) {
                S_HIP(hipStreamWaitEvent(st, s->ev_x, 0));
                S_COMM(comm_wait(c, st, "point-set all-gather"));
                (void)hipEventElapsedTime(&s->info.ms_allgather, e0, e1);
            }
        } else {
            if (x_async) (void)hipEventElapsedTime(&s->info.ms_allgather, e0, e1);  // (the stream has drained behind the leaf tables' read-back)
            section_timer sec(s);
            if (seed_old()) return 1;
            const int tl = t_begin(h);
            S_CTX(nnd_launch_leaf_init(h));
            t_end(h, tl, &h->stats.ms_leaf_init, false);
            sec.end();
            seeded = true;
        }
    }
    if (!seeded) {
    // ---- prep (all rows: 5 ms at 10 M points, cheaper than shipping the prepared copies over xGMI), reset ----
        section_timer sec(s);
        h->x_orig = x_use;
        h->x_owned = false;
        h->x_valid = true;
        const int tp = t_begin(h);
        S_CTX(nnd_launch_prep(h));
        S_CTX(nnd_launch_reset_graph(h));
        t_end(h, tp, &h->stats.ms_prep, false);
        if (seed_old()) return 1;
        // ---- forest split by tree: this rank seeds ALL rows from its own trees ----
        if (h->p.n_trees > 0) {
            const int tf = t_begin(h);
            S_CTX(nnd_launch_forest(h));
            t_end(h, tf, &h->stats.ms_forest, false);
            const int tl = t_begin(h);
            S_CTX(nnd_launch_leaf_init(h));
            t_end(h, tl, &h->stats.ms_leaf_init, false);
            if (s->own_order && n_own > 0) {  // the first local tree's leaf order, restricted to the owned vertices
                S_HIP(hipMemsetAsync(s->order_cursor, 0, sizeof(int), st));
                hipLaunchKernelGGL(k_compact_owned, dim3((unsigned)((h->n + 4095) / 4096)), dim3(256), 0, st, h->perm[h->cur], h->n, s->lo, s->hi,
                                   s->own_order, s->order_cursor);
                h->own_order = s->own_order;
                h->forest_gen++;  // (as above: own_order is one buffer rewritten by every build)
            }
        }
        sec.end();
    }
    // ---- partial k-list rows -> their owners (all-to-all-v of row blocks), merged there ----
    if (G > 1 && s->gp.n_trees > 0) {
        const int64_t b0 = c->bytes_sent;
        auto trees_of = [&](int r) { return s->by_cell ? 1 : (int)((int64_t)s->gp.n_trees * (r + 1) / G) - (int)((int64_t)s->gp.n_trees * r / G); };
        size_t soff[NND_MAX_RANKS], scnt[NND_MAX_RANKS], roff[NND_MAX_RANKS], rcnt[NND_MAX_RANKS];
        size_t at = 0;
        for (int r = 0; r < G; r++) {
            const size_t rows_r = (size_t)(s->bounds[r + 1] - s->bounds[r]);
            soff[r] = (size_t)s->bounds[r] * s->ks;
            scnt[r] = (r != me && h->p.n_trees > 0) ? rows_r * s->ks : 0;  // a rank without trees has nothing to offer
            roff[r] = at;
            rcnt[r] = (r != me && trees_of(r) > 0) ? (size_t)n_own * s->ks : 0;
            at += rcnt[r];
        }
        void *sb[2] = {h->knn_e, h->knn_d}, *rb[2] = {s->recv_e, s->recv_d};
        const int eb[2] = {4, 4};
        S_HIP(hipEventRecord(e0, st));
        S_COMM(comm_alltoallv(c, st, 2, sb, rb, eb, soff, scnt, roff, rcnt));
        S_HIP(hipEventRecord(e1, st));
        note_bytes(s, b0);
        section_timer sec(s);
        // (merged entries carry the "new" flag like everything else before the first sampling pass: all_new stays set)
        int n_src = 0;  // the sources' blocks lie back to back: (n_own, ks) rows each
        for (int r = 0; r < G; r++) n_src += rcnt[r] ? 1 : 0;
        S_CTX(nnd_launch_merge_graph_rows(h, s->lo, s->hi, s->recv_e, s->recv_d, n_src, (int64_t)n_own * s->ks));
        const int tr = t_begin(h);
        if (init_mode != 2) S_CTX(nnd_launch_random_init(h));  // owned rows that are still not full (pynndescent_.py:188-203)
        t_end(h, tr, &h->stats.ms_random_init, false);
        sec.end();
    } else if (init_idx_dev && init_mode == 2) {
        // (update() without a forest: the old entries are in, nothing else to seed from)
    } else if (init_idx_dev) {
        section_timer sec(s);
        const int tr = t_begin(h);
        S_CTX(nnd_launch_init_from_graph(h, init_idx_dev, init_dist_dev, init_width));
        t_end(h, tr, &h->stats.ms_random_init, false);
        sec.end();
    } else {
        section_timer sec(s);
        const int tr = t_begin(h);
        S_CTX(nnd_launch_random_init(h));
        t_end(h, tr, &h->stats.ms_random_init, false);
        sec.end();
    }

    // ---- NN-descent (pynndescent_.py:296-320), every rank scanning only ITS rows ----
    std::vector<long long> matrix((size_t)G * nv);
    int64_t caps_o[NND_MAX_RANKS], caps_p[NND_MAX_RANKS];
    for (int r = 0; r < G; r++) {
        const int64_t rows_r = s->bounds[r + 1] - s->bounds[r];
        caps_o[r] = rows_r * s->k > 0 ? rows_r * s->k : 1;  // = nnd_shard_create's cap_o of rank r
        caps_p[r] = s->cap_p;
    }
    // per-iteration kernel counters: reduced on the device after the merge, copied to a pinned block, read after the
    // next host wait (no read-back of their own)
    auto harvest = [&](int it_done) {
        if (it_done < 0 || it_done >= 64) return;
        h->stats.join_pairs[it_done] = h->h_pin[CNT_PAIRS];
        h->stats.join_rows[it_done] = h->h_pin[CNT_ROWS];
        h->stats.join_active[it_done] = h->h_pin[CNT_ACTIVE];
        h->stats.proposals[it_done] = h->h_pin[CNT_PROPOSALS];
        h->stats.join_mfma[it_done] = h->h_pin[CNT_MFMA];
    };
    static float sink;
    const double stop_at = (double)s->gp.delta * s->k * (double)s->n_total;
    bool stopped = false;
    long long c_prev = 0;  // the global update count of the previous iteration
    for (int it = 0; it < s->gp.n_iters; it++) {
        float *ms_s = it < 64 ? &h->stats.ms_sample[it] : &sink, *ms_j = it < 64 ? &h->stats.ms_join[it] : &sink,
              *ms_m = it < 64 ? &h->stats.ms_merge[it] : &sink;
        if (it == 1 && (s->gp.flags & (NND_FLAG_TEST_FAIL | NND_FLAG_TEST_VANISH))) {  // test hooks: see include/pynnd_amd.h
            s->set_error("test hook: rank %d %s at iteration 1", me, (s->gp.flags & NND_FLAG_TEST_VANISH) ? "vanishes" : "fails");
            return (s->gp.flags & NND_FLAG_TEST_VANISH) ? 2 : 1;
        }
        // (1) sampling, first half: own new edges; offers to targets owned elsewhere become records
        {
            section_timer sec(s);
            const int ts = t_begin(h);
            S_CTX(nnd_launch_sample_begin(h, s->cap_o > 0 ? s->cap_o : 1, s->off_t, s->off_k, s->cvec));
            t_end(h, ts, ms_s, false);
            hipLaunchKernelGGL(k_pack_counts, dim3(1), dim3(128), 0, st, h->shard_cursors, h->counters_sum, G, it > 0 ? 1 : 0, s->cvec);
            sec.end();
        }
        S_COMM(comm_gather_counts(c, st, s->cvec, nv, matrix.data()));  // host wait: record counts (+ last iteration's c)
        t_flush(h);
        if (it > 0) {  // the stop rule of the iteration that just finished (pynndescent_.py:317), on the GLOBAL count
            harvest(it - 1);
            long long ctot = 0;
            for (int r = 0; r < G; r++) ctot += matrix[(size_t)r * nv + G];
            c_prev = ctot;
            if (it - 1 < 64) s->info.c[it - 1] = ctot;
            if (it - 1 < 64) h->stats.updates[it - 1] = matrix[(size_t)me * nv + G];
            if ((double)ctot <= stop_at) {
                stopped = true;
                break;  // what sample_begin queued is harmless: reverse-offer slots are re-armed by the next build's reset
            }
        }
        // (2) what the join needs of the rows owned elsewhere, all-gathered in place (this rank's slices are where the
        //     others read from): the thresholds, 4 bytes per row (a stale threshold only admits extra proposals), and the
        //     neighbour ids, 4 * ks bytes per row -- with them a proposal for a remote vertex passes the same membership
        //     test as a local one (utils.py:489-492) BEFORE it competes for a proposal slot.  Without the ids most records
        //     shipped were "already present"; being near their target by construction they also won the hashed slots from
        //     the genuine candidates (recall at 8 ranks x 10 M points: 0.969 against 0.980 on one GPU).
        //     STALE ids are a safe filter too: an id that has left the row since was evicted by k closer ones (it cannot
        //     come back), an id that has entered since only lets a junk record through to the owner's own test.  So once
        //     an iteration changes under NND_GATHER_MIN of the entries (known here: the count wait above) the ids are not
        //     gathered again -- 4 * ks of the 4 + 4 * ks bytes per row, in the iterations where few rows changed.
        //     Round 6: only the JOIN needs what this gather brings, so with a second channel it runs THERE, beside the offer
        //     exchange and the second half of the sampling (which read and write this rank's own rows only; the peers read the
        //     slices being sent while the sampling clears "new" flags in them -- the membership tests mask that bit), and the
        //     build's stream waits for it in front of the join.  LOCAL serial mode (the critical-path tool) keeps the gather on
        //     the build's channel -- nothing may run beside a timed section -- and reports it for the tool to price the overlap.
        bool gather_async = false;
        if (G > 1) {
            const bool ids_now = s->gather_lists && (it == 0 || (double)c_prev >= NND_GATHER_MIN * (double)s->k * (double)s->n_total);
            size_t soff[NND_MAX_RANKS], scnt[NND_MAX_RANKS], roff[NND_MAX_RANKS], rcnt[NND_MAX_RANKS];
            for (int r = 0; r < G; r++) {
                soff[r] = (size_t)s->lo;
                scnt[r] = (size_t)n_own;
                roff[r] = (size_t)s->bounds[r];
                rcnt[r] = (size_t)(s->bounds[r + 1] - s->bounds[r]);
            }
            void *sb[2] = {h->th, h->knn_e}, *rb[2] = {h->th, h->knn_e};
            const int eb[2] = {4, 4 * s->ks};
            gather_async = c->aux && c->aux_stream && !serial && !(s->gp.flags & NND_FLAG_TEST_GATHER_INLINE);
            nnd_comm_s *gc = gather_async ? c->aux : c;
            hipStream_t gs = gather_async ? c->aux_stream : st;
            if (gather_async) {  // the second stream starts behind the merge that made this rank's rows final
                S_HIP(hipEventRecord(s->ev_g0, st));
                S_HIP(hipStreamWaitEvent(gs, s->ev_g0, 0));
            }
            const int64_t b0 = gc->bytes_sent;
            if (comm_alltoallv(gc, gs, ids_now ? 2 : 1, sb, rb, eb, soff, scnt, roff, rcnt)) { s->set_error("id / threshold gather: %s", gc->err); return 1; }
            const int64_t moved = gc->bytes_sent - b0;
            if (gc != c) c->bytes_sent += moved;
            if (gather_async) S_HIP(hipEventRecord(s->ev_g1, gs));
            if (it < 64) {
                s->info.gather_bytes[it] = moved;
                s->info.gather_section[it] = -1;
            }
            if (!gather_async && !serial) note_bytes(s, b0);  // (serial mode: priced apart by the tool, see gather_section)
            h->lists_replicated = s->gather_lists;
        }
        s->info.dropped_offers += matrix[(size_t)me * nv + G + 1];
        int64_t n_in = 0, n_sent = 0;
        if (G > 1) {
            const int64_t b0 = c->bytes_sent;
            if (exchange_records(s, s->off_t, s->off_k, 4, caps_o, matrix.data(), nv, &n_in, &n_sent)) return 1;
            note_bytes(s, b0);
        }
        if (it < 64) s->info.offer_records[it] = n_sent;
        {   // the second half of the sampling: a section of its own since round 6 -- it is what the gather above runs beside
            section_timer sec(s);
            const int ts = t_begin(h);
            S_CTX(nnd_launch_sample_finish(h, s->in_t, (const uint32_t *)s->in_k, n_in));
            t_end(h, ts, ms_s, true);
            if (G > 1 && it < 64 && (gather_async || serial)) s->info.gather_section[it] = s->info.n_sections;  // (the index this section gets)
            sec.end();
        }
        if (gather_async) S_HIP(hipStreamWaitEvent(st, s->ev_g1, 0));  // thresholds and ids of the rows owned elsewhere are in place
        {
            section_timer sec(s);
            // (3) local join of the owned vertices
            S_CTX(nnd_zero_counters(h));
            // (rows of more than 64 neighbours: join_blocks sub-steps as in the one-GPU build, capi.hip auto_join_blocks -- the
            // owned rows are merged between the sub-steps, the proposals for rows owned elsewhere keep collecting in their narrow
            // table and travel once, below)
            const int tj = t_begin(h);
            const int nsub = h->p.join_blocks > 1 ? h->p.join_blocks : 1;
            if (h->iter < 64) h->stats.join_substeps[h->iter] = nsub;
            for (int b = 0; b < nsub; b++) {
                const int64_t v0 = s->lo + (int64_t)n_own * b / nsub, v1 = s->lo + (int64_t)n_own * (b + 1) / nsub;
                S_CTX(nnd_launch_join(h, v0, v1));
                if (b + 1 < nsub) S_CTX(nnd_launch_merge(h));
            }
            t_end(h, tj, ms_j, false);
            // (4) proposals for vertices owned elsewhere -> their owners' regions
            if (G > 1) {
                S_CTX(nnd_launch_proposal_export_regions(h, s->cap_p, s->prop_t, s->prop_k, s->cvec));
                hipLaunchKernelGGL(k_pack_counts, dim3(1), dim3(128), 0, st, h->shard_cursors, h->counters_sum, G, 0, s->cvec);
            }
            sec.end();
        }
        n_in = 0;
        if (G > 1) {
            S_COMM(comm_gather_counts(c, st, s->cvec, nv, matrix.data()));  // host wait: proposal record counts
            if (it < 64) s->info.deferred[it] = matrix[(size_t)me * nv + G + 2];
            // a region holds the rows that fit entirely: cursor minus what was deferred behind it may exceed cap only by
            // deferred rows, whose slots inside the region are marked invalid (target -1) -- ship min(cursor, cap)
            const int64_t b0 = c->bytes_sent;
            if (exchange_records(s, s->prop_t, s->prop_k, 8, caps_p, matrix.data(), nv, &n_in, &n_sent)) return 1;
            note_bytes(s, b0);
            if (it < 64) s->info.proposal_records[it] = n_sent;
        }
        {
            section_timer sec(s);
            if (n_in) S_CTX(nnd_launch_import_proposals(h, s->in_k, s->in_t, n_in));
            // (5) owner-side merge; its update count stays on the device and rides on the next count exchange
            const int tm = t_begin(h);
            S_CTX(nnd_launch_merge(h));
            t_end(h, tm, ms_m, false);
            hipLaunchKernelGGL(k_counters_reduce_async, dim3(1), dim3(256), 0, st, h->counters, h->counters_sum);
            S_HIP(hipMemcpyAsync(h->h_pin, h->counters_sum, sizeof(long long) * CNT_COUNT, hipMemcpyDeviceToHost, st));
            sec.end();
        }
        h->iter++;
        h->stats.n_iters_run = h->iter;
        s->info.iters = it + 1;
    }
    if (!stopped) {  // n_iters reached: the last iteration's count is still on the device (statistics only)
        hipLaunchKernelGGL(k_pack_counts, dim3(1), dim3(128), 0, st, h->shard_cursors, h->counters_sum, G, 1, s->cvec);
        S_COMM(comm_gather_counts(c, st, s->cvec, nv, matrix.data()));
        long long ctot = 0;
        for (int r = 0; r < G; r++) ctot += matrix[(size_t)r * nv + G];
        const int it = s->info.iters - 1;
        harvest(it);
        if (it >= 0 && it < 64) {
            s->info.c[it] = ctot;
            h->stats.updates[it] = matrix[(size_t)me * nv + G];
        }
    }
    {
        section_timer sec(s);
        const int tf = t_begin(h);
        S_CTX(nnd_launch_finalize(h, out_idx_dev, out_dist_dev));
        t_end(h, tf, &h->stats.ms_finalize, false);
        t_flush(h);
        sec.end();
    }
    S_COMM(comm_wait(c, st, "end of the build"));
    // LOCAL: the ranks' streams wait on each other's events; nobody returns (and tears its stream / events down) while a
    // peer's stream may still hold such a wait.  After this barrier every rank's stream has drained.
    if (c->kind == NND_COMM_LOCAL && G > 1) S_COMM(comm_barrier(c));
    s->info.ms_klist_exchange = 0.0f;
    if (G > 1 && s->gp.n_trees > 0) (void)hipEventElapsedTime(&s->info.ms_klist_exchange, e0, e1);
    s->info.bytes_sent = c->bytes_sent;
    s->info.ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (s->info.dropped_offers) {
        s->set_error("sharded build: %lld reverse-offer records did not fit their region (cap %lld): results would be degraded", (long long)s->info.dropped_offers, (long long)s->cap_o);
        return 1;
    }
    return 0;
}

extern "C" int32_t nnd_shard_build(nnd_shard_t s, const float *x_local_dev, void *x_stream, int32_t *out_idx_dev, float *out_dist_dev) {
    return nnd_shard_build_from_graph(s, x_local_dev, x_stream, nullptr, nullptr, 0, out_idx_dev, out_dist_dev);
}
static int32_t shard_build_entry(nnd_shard_t s, const float *x_local_dev, void *x_stream, const int32_t *init_idx_dev, const float *init_dist_dev,
                                 int32_t init_width, int init_mode, int32_t *out_idx_dev, float *out_dist_dev);
extern "C" int32_t nnd_shard_build_from_graph(nnd_shard_t s, const float *x_local_dev, void *x_stream, const int32_t *init_idx_dev,
                                              const float *init_dist_dev, int32_t init_width, int32_t *out_idx_dev, float *out_dist_dev) {
    return shard_build_entry(s, x_local_dev, x_stream, init_idx_dev, init_dist_dev, init_width, init_idx_dev ? 1 : 0, out_idx_dev, out_dist_dev);
}
extern "C" int32_t nnd_shard_build_update(nnd_shard_t s, const float *x_local_dev, void *x_stream, const int32_t *old_idx_dev,
                                          const float *old_dist_dev, int32_t width, int32_t *out_idx_dev, float *out_dist_dev) {
    if (s && (!old_idx_dev || !old_dist_dev)) { s->set_error("nnd_shard_build_update: the previous graph (ids and alt-space distances) is required"); return 1; }
    return shard_build_entry(s, x_local_dev, x_stream, old_idx_dev, old_dist_dev, width, 2, out_idx_dev, out_dist_dev);
}
static int32_t shard_build_entry(nnd_shard_t s, const float *x_local_dev, void *x_stream, const int32_t *init_idx_dev, const float *init_dist_dev,
                                 int32_t init_width, int init_mode, int32_t *out_idx_dev, float *out_dist_dev) {
    if (!s) { snprintf(g_serr2, sizeof(g_serr2), "nnd_shard_build: null shard"); return 1; }
    if (!x_local_dev || !out_idx_dev || !out_dist_dev) { s->set_error("nnd_shard_build: null buffer"); return 1; }
    if (init_idx_dev && ((init_mode == 1 && s->gp.n_trees != 0) || init_width < 1 || init_width > NND_WIDE_K)) {
        s->set_error("nnd_shard_build_from_graph: an init graph needs a shard created with n_trees = 0 (pynndescent_.py:1059-1062) and a width in 1..%d", NND_WIDE_K);
        return 1;
    }
    const int rc = shard_build(s, x_local_dev, x_stream, out_idx_dev, out_dist_dev, init_idx_dev, init_dist_dev, init_width, init_mode);
    s->h->wait_hook = nullptr;
    // a rank that fails tells the ranks of its process (shared flag, LOCAL barriers) and cancels its own collectives
    // (ncclCommAbort): nobody is left waiting in a collective.  (rc 2: the test hook of a rank that dies silently.)
    if (rc == 1) (void)nnd_comm_abort(s->comm);
    return rc ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// nnd_build over several GPUs of this node: host buffers in, host buffers out, one host thread per GPU.
extern "C" int32_t nnd_build_multi(const nnd_params *params, const float *x, int32_t n_devices, const int32_t *devices, int32_t *out_idx,
                                   float *out_dist, nnd_stats *stats, nnd_shard_info *info_rank0, char *err, int32_t errlen) {
    return nnd_build_multi_from_graph(params, x, n_devices, devices, nullptr, nullptr, 0, out_idx, out_dist, stats, info_rank0, err, errlen);
}
// ... with the warm start of NNDescent(init_graph=..., init_dist=...) (pynndescent_.py:1225-1242): init_idx host (n, init_width) global
// ids, init_dist nullable (distances are computed); params->n_trees is taken as 0 (an init graph disables the forest, 1059-1062)
static int32_t build_multi_impl(const nnd_params *params_in, const float *x, int32_t n_devices, const int32_t *devices, const int32_t *init_idx,
                                const float *init_dist, int32_t init_width, int init_mode, int32_t *out_idx, float *out_dist, nnd_stats *stats,
                                nnd_shard_info *info_rank0, char *err, int32_t errlen);
extern "C" int32_t nnd_build_multi_from_graph(const nnd_params *params_in, const float *x, int32_t n_devices, const int32_t *devices,
                                              const int32_t *init_idx, const float *init_dist, int32_t init_width, int32_t *out_idx,
                                              float *out_dist, nnd_stats *stats, nnd_shard_info *info_rank0, char *err, int32_t errlen) {
    return build_multi_impl(params_in, x, n_devices, devices, init_idx, init_dist, init_width, init_idx ? 1 : 0, out_idx, out_dist, stats, info_rank0, err, errlen);
}
// NNDescent.update() (pynndescent_.py:2498-2535) over n_devices GPUs: a fresh forest of params->n_trees trees over the (changed) point
// set, the previous graph's surviving entries as OLD entries (old_idx (n, width) global ids, -1 where an entry was invalidated;
// old_dist: their alt-space distances), no random fill
extern "C" int32_t nnd_build_multi_update(const nnd_params *params_in, const float *x, int32_t n_devices, const int32_t *devices,
                                          const int32_t *old_idx, const float *old_dist, int32_t width, int32_t *out_idx, float *out_dist,
                                          nnd_stats *stats, nnd_shard_info *info_rank0, char *err, int32_t errlen) {
    if (!old_idx || !old_dist) {
        if (err && errlen > 0) snprintf(err, (size_t)errlen, "nnd_build_multi_update: the previous graph (ids and alt-space distances) is required");
        return 1;
    }
    return build_multi_impl(params_in, x, n_devices, devices, old_idx, old_dist, width, 2, out_idx, out_dist, stats, info_rank0, err, errlen);
}
static int32_t build_multi_impl(const nnd_params *params_in, const float *x, int32_t n_devices, const int32_t *devices, const int32_t *init_idx,
                                const float *init_dist, int32_t init_width, int init_mode, int32_t *out_idx, float *out_dist, nnd_stats *stats,
                                nnd_shard_info *info_rank0, char *err, int32_t errlen) {
    nnd_params params_v{};
    if (params_in) params_v = *params_in;
    if (init_idx && init_mode == 1) params_v.n_trees = 0;
    const nnd_params *params = params_in ? &params_v : nullptr;
    auto fail = [&](const std::string &msg) {
        if (err && errlen > 0) { strncpy(err, msg.c_str(), (size_t)errlen - 1); err[errlen - 1] = 0; }
        return 1;
    };
    if (!params || !x || !out_idx || !out_dist) return fail("nnd_build_multi: null argument");
    if (init_idx && (init_width < 1 || init_width > NND_WIDE_K)) return fail("nnd_build_multi_from_graph: init_width must be in 1..256");
    if (n_devices < 1 || n_devices > NND_MAX_RANKS) return fail("nnd_build_multi: n_devices must be in 1..64");
    if (params->n < 1) return fail("nnd_build_multi: need n >= 1");
    const int G = (int64_t)n_devices <= params->n ? n_devices : (int)params->n;  // every rank owns at least one row: the first n devices build a set of n < n_devices points
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("nnd_build_multi: no HIP device visible (this library has no CPU path)");
    std::vector<int32_t> dev(G);
    bool distinct = true;
    for (int r = 0; r < G; r++) {
        dev[r] = devices ? devices[r] : r;
        if (dev[r] < 0 || dev[r] >= ndev) return fail("nnd_build_multi: device " + std::to_string(dev[r]) + " out of range (" + std::to_string(ndev) + " visible)");
        for (int q = 0; q < r; q++) distinct = distinct && dev[q] != dev[r];
    }
    std::vector<int64_t> sizes(G);
    std::vector<int64_t> lo(G + 1);
    for (int r = 0; r <= G; r++) lo[r] = params->n * r / G;
    for (int r = 0; r < G; r++) sizes[r] = lo[r + 1] - lo[r];
    // communicators: RCCL between distinct GPUs, the LOCAL transport when ranks share one
    std::vector<nnd_comm_t> comms(G, nullptr);
    unsigned char id[NND_COMM_ID_BYTES], id2[NND_COMM_ID_BYTES];
    const bool use_rccl = distinct && G > 1;
    if (use_rccl) {
        if (nnd_comm_unique_id(id) || nnd_comm_unique_id(id2)) return fail(std::string("nnd_build_multi: ") + nnd_comm_last_error(nullptr));
    } else {
        if (nnd_comm_create_local(comms.data(), G, dev.data())) return fail(std::string("nnd_build_multi: ") + nnd_comm_last_error(nullptr));
    }
    std::vector<std::string> errs(G);
    std::vector<int> rcs(G, 0);
    std::vector<nnd_stats> st(G);
    std::vector<nnd_shard_info> inf(G);
    // The rank threads agree before anything collective starts: a rank that cannot set its device up, create its shard or
    // stage its rows raises `failed`; after the barrier every rank sees it and nobody enters the build (under RCCL the
    // others would sit in the first ncclSend / ncclRecv for ever).  Failures DURING the build travel through the
    // communicators' shared abort flag (comm.h): the ranks that notice call ncclCommAbort and return.
    std::atomic<int> failed{0}, abort_flag{0};
    struct thread_barrier {
        std::mutex mu;
        std::condition_variable cv;
        int n, arrived = 0;
        uint64_t gen = 0;
        explicit thread_barrier(int n_) : n(n_) {}
        void wait() {
            std::unique_lock<std::mutex> lk(mu);
            const uint64_t g = gen;
            if (++arrived == n) {
                arrived = 0;
                gen++;
                cv.notify_all();
            } else {
                cv.wait(lk, [&] { return gen != g; });
            }
        }
    } bar(G);
    auto run = [&](int r) {
        auto bail = [&](const std::string &m) {
            errs[r] = m;
            rcs[r] = 1;
            failed.store(1);
        };
        if (hipSetDevice(dev[r]) != hipSuccess) bail("hipSetDevice failed");
        bar.wait();  // (1) every rank has its device: ncclCommInitRank returns only when all ranks have called it
        if (failed.load()) return;
        if (use_rccl) {
            if (nnd_comm_create_rccl(&comms[r], id, G, r, dev[r])) bail(nnd_comm_last_error(nullptr));
            bar.wait();  // (1b) every rank has its first channel, or nobody asks for the second (its creation is collective too)
            const bool go2 = !failed.load();
            bar.wait();  // (1c) ... and every rank has READ that decision before anybody can change `failed` again: a rank whose
                         // add_channel fails at once must not make a slower rank skip the collective the others already sit in
            if (go2 && nnd_comm_add_channel_rccl(comms[r], id2)) bail(nnd_comm_last_error(nullptr));
            if (comms[r]) {
                comms[r]->abort_flag = &abort_flag;
                if (comms[r]->aux) comms[r]->aux->abort_flag = &abort_flag;
            }
        }
        nnd_params p = *params;
        p.device = dev[r];
        nnd_shard_t sh = nullptr;
        const size_t nl = (size_t)sizes[r];
        float *dx = nullptr;
        int32_t *di = nullptr;
        float *dd = nullptr;
        int32_t *gi = nullptr;  // the owned rows of the init graph
        float *gd = nullptr;
        if (!rcs[r] && nnd_shard_create(&sh, &p, comms[r], sizes.data())) bail(nnd_shard_last_error(nullptr));
        if (!rcs[r]) {
            bool ok = hipMalloc((void **)&dx, sizeof(float) * (nl ? nl : 1) * p.dim) == hipSuccess &&
                      hipMalloc((void **)&di, sizeof(int32_t) * (nl ? nl : 1) * p.n_neighbors) == hipSuccess &&
                      hipMalloc((void **)&dd, sizeof(float) * (nl ? nl : 1) * p.n_neighbors) == hipSuccess;
            if (ok && nl) ok = hipMemcpy(dx, x + (size_t)lo[r] * p.dim, sizeof(float) * nl * p.dim, hipMemcpyHostToDevice) == hipSuccess;
            if (ok && init_idx) {
                ok = hipMalloc((void **)&gi, sizeof(int32_t) * (nl ? nl : 1) * init_width) == hipSuccess &&
                     (!init_dist || hipMalloc((void **)&gd, sizeof(float) * (nl ? nl : 1) * init_width) == hipSuccess);
                if (ok && nl) ok = hipMemcpy(gi, init_idx + (size_t)lo[r] * init_width, sizeof(int32_t) * nl * init_width, hipMemcpyHostToDevice) == hipSuccess;
                if (ok && nl && init_dist) ok = hipMemcpy(gd, init_dist + (size_t)lo[r] * init_width, sizeof(float) * nl * init_width, hipMemcpyHostToDevice) == hipSuccess;
            }
            if (!ok) bail("allocation / H2D of the shard failed");
        }
        bar.wait();  // (2) every rank is ready, or nobody builds
        if (!failed.load()) {
            if (shard_build_entry(sh, dx, nullptr, gi, gd, init_width, init_mode, di, dd)) {
                errs[r] = nnd_shard_last_error(sh);
                rcs[r] = 1;
            } else if (nl && (hipMemcpy(out_idx + (size_t)lo[r] * p.n_neighbors, di, sizeof(int32_t) * nl * p.n_neighbors, hipMemcpyDeviceToHost) != hipSuccess ||
                              hipMemcpy(out_dist + (size_t)lo[r] * p.n_neighbors, dd, sizeof(float) * nl * p.n_neighbors, hipMemcpyDeviceToHost) != hipSuccess)) {
                errs[r] = "D2H of the result failed";
                rcs[r] = 1;
            }
        } else if (!rcs[r]) {
            errs[r] = "another rank failed before the build started";
            rcs[r] = 2;
        }
        if (sh) {
            (void)nnd_shard_get_stats(sh, &st[r]);
            (void)nnd_shard_get_info(sh, &inf[r]);
        }
        // (3) nobody frees while a peer's stream may still hold a copy that reads this rank's buffers (LOCAL transport: peer
        // copies queued on the OTHER rank's stream; a failed build returns with work still queued): every rank is out of its
        // build -- the failing ones have aborted their communicators -- then every rank drains its own streams, bounded
        bar.wait();
        if (sh) {
            nnd_handle_t hh = nnd_shard_handle(sh);
            const auto t_dr = std::chrono::steady_clock::now();
            while (hh && hh->stream && hipStreamQuery(hh->stream) == hipErrorNotReady &&
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t_dr).count() < 5.0) {
            }
            (void)hipGetLastError();
        }
        bar.wait();  // (4) ... and everybody has drained before the first hipFree
        if (dx) (void)hipFree(dx);
        if (di) (void)hipFree(di);
        if (dd) (void)hipFree(dd);
        if (gi) (void)hipFree(gi);
        if (gd) (void)hipFree(gd);
        if (sh) (void)nnd_shard_destroy(sh);
    };
    std::vector<std::thread> th;
    for (int r = 0; r < G; r++) th.emplace_back(run, r);
    for (auto &t : th) t.join();
    for (int r = 0; r < G; r++)
        if (comms[r]) (void)nnd_comm_destroy(comms[r]);
    for (int r = 0; r < G; r++)  // the rank that failed first-hand, if any
        if (rcs[r] == 1 && errs[r].find("another rank failed") == std::string::npos) return fail("nnd_build_multi: rank " + std::to_string(r) + ": " + errs[r]);
    for (int r = 0; r < G; r++)
        if (rcs[r]) return fail("nnd_build_multi: rank " + std::to_string(r) + ": " + errs[r]);
    if (stats) *stats = st[0];
    if (info_rank0) *info_rank0 = inf[0];
    return 0;
}
