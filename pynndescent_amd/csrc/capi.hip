// capi.hip -- extern "C" boundary (include/pynnd_amd.h): handle lifetime, HBM allocation, and
// the orchestration that mirrors nn_descent / nn_descent_internal (reference pynndescent_.py:266-366).
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "state.h"

static thread_local char g_err[512] = {0};

static void gerr(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int32_t nnd_abi_version(void) { return NND_ABI_VERSION; }
extern "C" const char *nnd_last_global_error(void) { return g_err; }
extern "C" const char *nnd_last_error(nnd_handle_t h) { return h ? h->err : g_err; }

#define API_HIP(expr)                                                                                \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            ctx->set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

void nnd_release_parked();
template <typename T>
static int dalloc(nnd_ctx *ctx, T **p, size_t count) {
    if (hipMalloc((void **)p, sizeof(T) * (count ? count : 1)) != hipSuccess) {
        (void)hipGetLastError();
        nnd_release_parked();  // a parked handle (nnd_destroy) may hold what is missing
        API_HIP(hipMalloc((void **)p, sizeof(T) * (count ? count : 1)));
    }
    // debugging aid: fresh hipMalloc pages are usually zero, recycled ones are not -- NND_POISON=<byte> fills every buffer with
    // that byte (try 165: negative ints / tiny floats, and 1 or 127: positive ints) before the build initialises it
    static const int poison = [] { const char *e = nnd_knob("NND_POISON"); return e ? atoi(e) : 0; }();  // the fill byte
    // (on the handle's own stream, like the memsets of nnd_create_impl below: a hipMemset on the NULL stream queues behind
    // whatever the caller's framework still has in flight there and would land in the middle of the build)
    if (poison) {
        API_HIP(hipMemsetAsync(*p, poison & 0xFF, sizeof(T) * (count ? count : 1), ctx->stream));
        API_HIP(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

// Temporary device buffers of one API call: released on every return path.
struct nnd_scratch {
    std::vector<void *> ptrs;
    ~nnd_scratch() {
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
    }
    template <typename T>
    T *get(nnd_ctx *ctx, size_t count) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, sizeof(T) * (count ? count : 1));
        if (e != hipSuccess) {
            ctx->set_error("hipMalloc of %zu scratch bytes failed: %s", sizeof(T) * count, hipGetErrorString(e));
            return nullptr;
        }
        ptrs.push_back(p);
        return (T *)p;
    }
};

std::recursive_mutex &nnd_lifecycle_mutex() {
#ifdef NND_TEST_NO_LIFECYCLE_LOCK  // heap-check builds only (tools/gpu_asan.sh nolock): every thread gets its own mutex
    static thread_local std::recursive_mutex m;
#else
    static std::recursive_mutex m;
#endif
    return m;
}

static void free_all(nnd_ctx *ctx) {
    auto F = [](void *p) {
        if (p) (void)hipFree(p);
    };
    if (ctx->x_owned) F((void *)ctx->x_orig);
    for (void *&a : ctx->slim_alloc) { F(a); a = nullptr; }  // cand / rbuf / active (the working pointers may be biased)
    F(ctx->xp); F(ctx->nrm); F(ctx->nr2); F(ctx->xh); F(ctx->mean); F(ctx->knn_e); F(ctx->knn_d); F(ctx->th); F(ctx->pbuf_r);
    F(ctx->pdirty); F(ctx->out_idx); F(ctx->out_dist);
    F(ctx->rv_pos); F(ctx->rv_in_cursor); F(ctx->rv_in_rec); F(ctx->rv_ov);
    for (int i = 0; i < 2; i++) { F(ctx->perm[i]); F(ctx->pos_seg[i]); F(ctx->seg_start[i]); F(ctx->seg_len[i]); }
    F(ctx->inv); F(ctx->side); F(ctx->side_pt); F(ctx->leaf_flag); F(ctx->scan_out); F(ctx->scan_blk); F(ctx->seg_nleft); F(ctx->seg_child);
    F(ctx->xs); F(ctx->xsh); F(ctx->nr2s); F(ctx->node_hf); F(ctx->node_hh); F(ctx->node_child); F(ctx->node_pack); F(ctx->node_hfc); F(ctx->route_roots); F(ctx->route_ws); F(ctx->s_leaf_depth);
    F(ctx->cell_count); F(ctx->cell_start); F(ctx->cell_depth); F(ctx->small_list);
    F(ctx->hyper); F(ctx->hyper_h); F(ctx->leaf_start); F(ctx->leaf_len); F(ctx->wl_start); F(ctx->wl_len); F(ctx->colsum_partial); F(ctx->counters_sum);
    if (ctx->h_pin) { (void)hipHostFree(ctx->h_pin); ctx->h_pin = nullptr; }
    if (ctx->h_tree_begin) { (void)hipHostFree(ctx->h_tree_begin); ctx->h_tree_begin = nullptr; }
    F(ctx->tree_begin_dev); F(ctx->counters);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_spin) (void)hipEventDestroy(ctx->ev_spin);
    for (hipEvent_t e : ctx->tev) if (e) (void)hipEventDestroy(e);
    ctx->tev.clear();
    F(ctx->shard_bounds); F(ctx->shard_cursors);
    nnd_hub_tree_free(ctx);
    nnd_search_graph_free(ctx);
    if (ctx->stream && ctx->stream_owned) (void)hipStreamDestroy(ctx->stream);
}

static nnd_ctx *take_parked(const nnd_params *p);
extern "C" int32_t nnd_create(nnd_handle_t *out, const nnd_params *p) { return nnd_create_impl(out, p, nullptr, 0, 0); }

// join_blocks = 0: chosen here.  A row takes at most 64 updates per merge (its proposal slots); rows of more than 64 neighbours
// change by more than that per iteration while the graph is poor (the reference's heaps have no such bound, utils.py:459-500), so
// their iterations are cut into sub-steps (join a part of the vertices, merge, ...: pynndescent_.py:239-261 does the same in
// blocks of 16384 vertices) until the slots of an iteration add up to 2 k.
// More than 64 candidates per class run as five passes of the 64-slot join (join.hip launch_join_blocked) that all deposit into
// the same 64 proposal slots of a row: twice the sub-steps, so that a merge empties the slots between them (round-5 advisor item).
static void jb_knobs(nnd_ctx *ctx) {  // experiments (KNOBS builds only): the schedule of nnd_join_substeps
    if (const char *e = nnd_knob("NND_JB_MAX")) ctx->jb_max = atoi(e) < 1 ? 1 : atoi(e);
    if (const char *e = nnd_knob("NND_JB_DIV")) ctx->jb_div = atoi(e) < 1 ? 1 : atoi(e);
    if (const char *e = nnd_knob("NND_JB_FIRST")) ctx->jb_first = atoi(e) < 0 ? 0 : atoi(e);
}
static int auto_join_blocks(int k, int mc) { return (k <= 64 ? 1 : (k + 31) / 32) * (mc > 64 ? 2 : 1); }

int nnd_create_impl(nnd_handle_t *out, const nnd_params *p, const int64_t *bounds_host, int n_ranks, int rank) {
    if (!out || !p) { gerr("nnd_create: null argument"); return 1; }
    *out = nullptr;
    if (p->n < 1 || p->dim < 1) { gerr("nnd_create: need n >= 1 and dim >= 1 (got n=%lld dim=%d)", (long long)p->n, p->dim); return 1; }
    if (p->metric != NND_METRIC_SQEUCLIDEAN && p->metric != NND_METRIC_ALT_COSINE) { gerr("nnd_create: unknown metric %d", p->metric); return 1; }
    if (p->n_neighbors < 1 || p->n_neighbors > NND_WIDE_K) { gerr("nnd_create: n_neighbors must be in 1..%d (got %d)", NND_WIDE_K, p->n_neighbors); return 1; }
    if (p->max_candidates < 1 || p->max_candidates > 128) { gerr("nnd_create: max_candidates must be in 1..128 (got %d)", p->max_candidates); return 1; }
    if (p->n_trees < 0 || p->n_trees > 4096 || p->leaf_size < 1) { gerr("nnd_create: bad n_trees (0..4096) / leaf_size"); return 1; }
    if (p->n >= (int64_t)0x7FFFFFF0) { gerr("nnd_create: n too large for int32 ids"); return 1; }
    if (p->n_trees > 0 && (int64_t)p->n_trees * p->n >= (int64_t)0x7FFFFFF0) {
        gerr("nnd_create: n_trees * n = %lld exceeds the forest's int32 position space (2^31)", (long long)((int64_t)p->n_trees * p->n));
        return 1;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { gerr("nnd_create: no HIP device visible (this library has no CPU path)"); return 1; }
    if (p->device < 0 || p->device >= ndev) { gerr("nnd_create: device %d out of range (%d visible)", p->device, ndev); return 1; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) != hipSuccess) { gerr("nnd_create: hipGetDeviceProperties failed"); return 1; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { gerr("nnd_create: device %d is %s; this build targets gfx950 (MI355X) only", p->device, prop.gcnArchName); return 1; }
    if (hipSetDevice(p->device) != hipSuccess) { gerr("nnd_create: hipSetDevice failed"); return 1; }
    if (!bounds_host) {
        if (nnd_ctx *parked = take_parked(p)) {
            *out = parked;
            return 0;
        }
        nnd_release_parked();  // a parked handle of another geometry: its memory is wanted now
    }

    std::lock_guard<std::recursive_mutex> lifecycle(nnd_lifecycle_mutex());
    nnd_ctx *ctx = new nnd_ctx();
    ctx->p = *p;
    ctx->n = p->n;
    ctx->own_lo = 0;
    ctx->own_hi = p->n;
    if (bounds_host) {  // one shard of a row-sharded build: the geometry is known before anything is allocated
        if (n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks || bounds_host[0] != 0 || bounds_host[n_ranks] != p->n) {
            gerr("nnd_create: bad shard bounds (need 1 <= n_ranks <= 64, bounds from 0 to n)");
            delete ctx;
            return 1;
        }
        for (int r = 0; r < n_ranks; r++)
            if (bounds_host[r] > bounds_host[r + 1]) { gerr("nnd_create: shard bounds must not decrease"); delete ctx; return 1; }
        ctx->n_ranks = n_ranks;
        ctx->own_lo = bounds_host[rank];
        ctx->own_hi = bounds_host[rank + 1];
        ctx->slim = n_ranks > 1;
    }
    ctx->d = p->dim;
    ctx->dp = (p->dim + 31) & ~31;
    ctx->k = p->n_neighbors;
    ctx->ks = (p->n_neighbors + 15) & ~15;
    ctx->mc = p->max_candidates;
    ctx->mcp = p->max_candidates <= 16 ? 16 : (p->max_candidates <= 32 ? 32 : (p->max_candidates <= 64 ? 64 : 128));  // (128: the blocked passes of join.hip)
    if (ctx->ks > 64 && ctx->mcp < 32) ctx->mcp = 32;  // wide rows: the join that reads neighbour lists from global memory (join.hip k_local_join_w)
    // reverse-offer slots per (vertex, class): at least max_candidates rounded up to a power of two, so that a vertex
    // can fill its list from reverse offers alone, as the reference's max_candidates-deep heaps can (utils.py:277-306)
    ctx->rcap = p->max_candidates <= 32 ? 32 : (p->max_candidates <= 64 ? 64 : 128);  // (128: hashed slots, the bucketed pass stops at 64)
    if (const char *rc_env = nnd_knob("NND_RCAP")) {  // experiments: reverse-offer slots per (vertex, class), a power of two
        const int r = atoi(rc_env);
        if (r == 16 || r == 32 || r == 64) ctx->rcap = r;
    }
    ctx->pcap = 64;  // one candidate per lane in k_merge (merge.h NCHUNK = 1)
    if (const char *pc_env = nnd_knob("NND_PCAP")) {  // experiments: proposal slots per vertex, a power of two <= 64
        const int r = atoi(pc_env);
        if (r == 16 || r == 32 || r == 64) ctx->pcap = r;
    }
    ctx->jb_auto = ctx->p.join_blocks < 1;
    if (ctx->p.join_blocks < 1) ctx->p.join_blocks = auto_join_blocks(ctx->p.n_neighbors, ctx->p.max_candidates);
    jb_knobs(ctx);
    ctx->seed = nnd_mix32((uint32_t)p->rng_state[0] ^ nnd_mix32((uint32_t)p->rng_state[1] + 0x9E3779B9u) ^
                          nnd_mix32((uint32_t)p->rng_state[2] + 0x7F4A7C15u));
    ctx->tree_seed = nnd_mix32((uint32_t)p->tree_rng[0] ^ nnd_mix32((uint32_t)p->tree_rng[1] + 0x9E3779B9u) ^
                               nnd_mix32((uint32_t)p->tree_rng[2] + 0x7F4A7C15u));
    int rc = 0;
    do {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { ctx->set_error("hipStreamCreate failed"); rc = 1; break; }
        if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_spin, hipEventDisableTiming) != hipSuccess) { ctx->set_error("hipEventCreate failed"); rc = 1; break; }
        const size_t n = (size_t)ctx->n;
        const bool graph = !(p->flags & NND_FLAG_NO_GRAPH), prepared = !(p->flags & NND_FLAG_NO_PREP);
        if (!prepared && graph) { ctx->set_error("NND_FLAG_NO_PREP needs NND_FLAG_NO_GRAPH (the build reads the prepared rows)"); rc = 1; break; }
        if (prepared) {
            if ((rc = dalloc(ctx, &ctx->xp, n * ctx->dp))) break;
            if ((rc = dalloc(ctx, &ctx->nrm, n))) break;
            if (p->n_trees > 0 && (rc = dalloc(ctx, &ctx->xh, n * ctx->dp))) break;
            if (p->n_trees > 0 && (rc = dalloc(ctx, &ctx->nr2, n))) break;
            if ((rc = dalloc(ctx, &ctx->mean, (size_t)ctx->dp + 4))) break;  // + scale of the screening copies, 1 / scale^2, sampled max
        }
        if (graph) {
            if ((rc = dalloc(ctx, &ctx->knn_e, n * ctx->ks))) break;
            if ((rc = dalloc(ctx, &ctx->knn_d, n * ctx->ks))) break;
            if ((rc = dalloc(ctx, &ctx->th, n))) break;
            const size_t rows = ctx->slim ? (size_t)(ctx->own_hi - ctx->own_lo) : n;  // per-OWNED-row tables
            int32_t *a_cand = nullptr;
            uint32_t *a_rbuf = nullptr;
            uint8_t *a_active = nullptr;
            if ((rc = dalloc(ctx, &a_cand, rows * 2 * ctx->mcp))) break;
            ctx->slim_alloc[0] = a_cand;
            if ((rc = dalloc(ctx, &a_rbuf, rows * 2 * ctx->rcap))) break;
            ctx->slim_alloc[1] = a_rbuf;
            if ((rc = dalloc(ctx, &a_active, rows))) break;
            ctx->slim_alloc[2] = a_active;
            // the working pointers are biased by -own_lo rows (0 on a plain handle): kernels index by global vertex id
            ctx->cand = a_cand - (size_t)ctx->own_lo * (ctx->slim ? 1 : 0) * 2 * ctx->mcp;
            ctx->rbuf = a_rbuf - (size_t)ctx->own_lo * (ctx->slim ? 1 : 0) * 2 * ctx->rcap;
            ctx->active = a_active - (size_t)ctx->own_lo * (ctx->slim ? 1 : 0);
            uint64_t *a_pbuf = nullptr;
            if ((rc = dalloc(ctx, &a_pbuf, rows * ctx->pcap))) break;
            ctx->slim_alloc[3] = a_pbuf;
            ctx->pbuf = a_pbuf - (size_t)ctx->own_lo * (ctx->slim ? 1 : 0) * ctx->pcap;
            if (ctx->slim && (rc = dalloc(ctx, &ctx->pbuf_r, n * ctx->pcap_r))) break;
            if ((rc = dalloc(ctx, &ctx->pdirty, n))) break;
            // on the handle's stream: the NULL-stream form is ordered behind the caller's pending NULL-stream work (torch's
            // default stream) and not with this handle's non-blocking stream -- it could clear the flags of a build in progress
            if (hipMemsetAsync(ctx->pdirty, 0, n, ctx->stream) != hipSuccess) { ctx->set_error("hipMemset failed"); rc = 1; break; }
            if (ctx->n_ranks > 0) {
                if (hipMalloc((void **)&ctx->shard_bounds, sizeof(int64_t) * 65) != hipSuccess || hipMalloc((void **)&ctx->shard_cursors, sizeof(long long) * 66) != hipSuccess ||
                    hipMemcpy(ctx->shard_bounds, bounds_host, sizeof(int64_t) * (size_t)(n_ranks + 1), hipMemcpyHostToDevice) != hipSuccess) {
                    ctx->set_error("allocation of the shard tables failed"); rc = 1; break;
                }
            }
        }
        if ((rc = dalloc(ctx, &ctx->counters, (size_t)CNT_COUNT * NND_CNT_STRIPES))) break;
        if ((rc = dalloc(ctx, &ctx->counters_sum, (size_t)CNT_COUNT))) break;
        if (hipHostMalloc((void **)&ctx->h_pin, sizeof(long long) * 64, hipHostMallocDefault) != hipSuccess) { ctx->set_error("hipHostMalloc failed"); rc = 1; break; }
        memset(ctx->h_pin, 0, sizeof(long long) * 64);
        if (hipHostGetDevicePointer((void **)&ctx->h_pin_dev, ctx->h_pin, 0) != hipSuccess) { (void)hipGetLastError(); ctx->h_pin_dev = nullptr; }
        if (p->n_trees > 0) {
            ctx->P = (int64_t)p->n_trees * ctx->n;
            const size_t P = (size_t)ctx->P;
            ctx->max_segs = ctx->P / (p->leaf_size + 1) + p->n_trees + 8;
            // Routing pass (rpforest.hip): the top of the trees is built from every 16th point when the set is large
            // enough for that sample to resolve cells of a few hundred points, and rows fit the route kernel's registers.
            // (Round 4: stride 8 / cells of <= 48 sample members -> 16 / 24: the same cells on average, half the sample
            // passes -- 4.66 -> 4.28 ms per forest at 1 M x 8 trees, 14.5 -> 12.2 ms at 10 M x 2; in a sharded build the
            // sample tops are the part of the forest that is not divided by the number of ranks.)
            // NND_FOREST_WHOLE=1 forces the whole-set level-synchronous build (A/B measurements).
            const char *whole = nnd_knob("NND_FOREST_WHOLE");
            if (graph && p->n >= 131072 && ctx->dp <= 256 && !(whole && whole[0] == '1')) {
                const char *ss = nnd_knob("NND_SAMPLE_STRIDE");
                ctx->s_stride = ss ? atoi(ss) : 16;
                if (ctx->s_stride < 2) ctx->s_stride = 2;
                ctx->s_m = p->n / ctx->s_stride;
                const char *cl = nnd_knob("NND_CELL_LEAF");
                const char *es = nnd_knob("NND_EARLY_STOP");
                ctx->early_stop = es ? atoi(es) : 0;
                ctx->cell_leaf = cl ? atoi(cl) : 24;  // x stride: cells of <= ~450 points, ~215 on average (one wave per cell)
                if (ctx->cell_leaf < 8) ctx->cell_leaf = 8;
                const int64_t Ps = (int64_t)p->n_trees * ctx->s_m;
                ctx->node_cap = Ps / (ctx->cell_leaf / 4 > 1 ? ctx->cell_leaf / 4 : 1) + 4 * p->n_trees + 64;
                ctx->cell_cap = ctx->node_cap + p->n_trees;
                ctx->max_segs += ctx->cell_cap;
                if ((rc = dalloc(ctx, &ctx->xs, (size_t)ctx->s_m * ctx->dp))) break;
                if ((rc = dalloc(ctx, &ctx->xsh, (size_t)ctx->s_m * ctx->dp))) break;
                if ((rc = dalloc(ctx, &ctx->nr2s, (size_t)ctx->s_m))) break;
                if ((rc = dalloc(ctx, &ctx->node_hf, (size_t)ctx->node_cap * (ctx->dp + 4)))) break;
                if ((rc = dalloc(ctx, &ctx->node_hh, (size_t)ctx->node_cap * ctx->dp))) break;
                if ((rc = dalloc(ctx, &ctx->node_child, (size_t)ctx->node_cap * 2))) break;
                if ((rc = dalloc(ctx, &ctx->node_pack, (size_t)ctx->node_cap * (2 * ctx->dp + 16)))) break;
                if ((rc = dalloc(ctx, &ctx->node_hfc, (size_t)ctx->node_cap * (ctx->dp + 4)))) break;
                if ((rc = dalloc(ctx, &ctx->route_roots, (size_t)4096))) break;
                if ((rc = dalloc(ctx, &ctx->s_leaf_depth, (size_t)Ps))) break;
                if ((rc = dalloc(ctx, &ctx->cell_count, (size_t)ctx->cell_cap))) break;
                if ((rc = dalloc(ctx, &ctx->cell_start, (size_t)ctx->cell_cap))) break;
                if ((rc = dalloc(ctx, &ctx->cell_depth, (size_t)ctx->cell_cap))) break;
                if ((rc = dalloc(ctx, &ctx->small_list, (size_t)ctx->cell_cap * 3))) break;
            }
            const size_t S = (size_t)ctx->max_segs;
            for (int i = 0; i < 2 && !rc; i++) {
                if ((rc = dalloc(ctx, &ctx->perm[i], P))) break;
                if ((rc = dalloc(ctx, &ctx->pos_seg[i], P))) break;
                if ((rc = dalloc(ctx, &ctx->seg_start[i], S))) break;
                if ((rc = dalloc(ctx, &ctx->seg_len[i], S))) break;
            }
            if (rc) break;
            if ((rc = dalloc(ctx, &ctx->inv, P))) break;
            if ((rc = dalloc(ctx, &ctx->side, P))) break;
            if ((rc = dalloc(ctx, &ctx->side_pt, P))) break;
            if ((rc = dalloc(ctx, &ctx->leaf_flag, P))) break;
            if ((rc = dalloc(ctx, &ctx->scan_out, P + 1))) break;
            if ((rc = dalloc(ctx, &ctx->scan_blk, P / 2048 + 2))) break;
            if ((rc = dalloc(ctx, &ctx->seg_nleft, S))) break;
            if ((rc = dalloc(ctx, &ctx->seg_child, 5 * S))) break;  // child ids (2S) + finisher work list (3S)
            if ((rc = dalloc(ctx, &ctx->hyper, S * (size_t)(ctx->dp + 4)))) break;
            if ((rc = dalloc(ctx, &ctx->hyper_h, S * (size_t)ctx->dp))) break;
            // (sized for any tree count: a shard finishes cells of ALL the build's trees, whatever its own allocation)
            if ((rc = dalloc(ctx, &ctx->tree_begin_dev, (size_t)4097))) break;
            if (hipHostMalloc((void **)&ctx->h_tree_begin, sizeof(long long) * (size_t)4097, hipHostMallocDefault) != hipSuccess) { ctx->set_error("hipHostMalloc failed"); rc = 1; break; }
        }
    } while (0);
    if (rc) {
        gerr("nnd_create: %s", ctx->err);
        free_all(ctx);
        delete ctx;
        return 1;
    }
    *out = ctx;
    return 0;
}

// Creating and releasing a handle's HBM costs ~10 ms at 1 M points (every hipFree synchronises the device and unmaps;
// doing it on a background thread only moved the cost into the next call's hipMalloc).  nnd_destroy therefore PARKS one
// plain handle instead of freeing it, and nnd_create re-arms the parked handle when the geometry matches (same device,
// n, dim, metric, k, trees, leaf size, candidates, flags): repeated builds -- NNDescent(...) in a loop, nnd_build -- pay
// neither.  The parked handle holds its memory until a different geometry arrives, nnd_release_pending() is called, or
// the process ends; an allocation that fails while a handle is parked releases it and tries once more.
static std::mutex g_park_mu;
static nnd_ctx *g_parked = nullptr;

static void destroy_now(nnd_ctx *ctx) {
    std::lock_guard<std::recursive_mutex> lifecycle(nnd_lifecycle_mutex());
    (void)hipSetDevice(ctx->p.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_all(ctx);
    delete ctx;
}
void nnd_release_parked() {
    nnd_ctx *old = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_park_mu);
        old = g_parked;
        g_parked = nullptr;
    }
    if (old) destroy_now(old);
}
static bool same_geometry(const nnd_params &a, const nnd_params &b) {
    return a.n == b.n && a.dim == b.dim && a.metric == b.metric && a.n_neighbors == b.n_neighbors && a.n_trees == b.n_trees &&
           a.leaf_size == b.leaf_size && a.max_candidates == b.max_candidates && a.device == b.device && a.flags == b.flags;
}
// a parked handle of this geometry, re-armed for a new build (seeds, counters, per-build flags), or nullptr
static nnd_ctx *take_parked(const nnd_params *p) {
    nnd_ctx *ctx = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_park_mu);
        if (g_parked && same_geometry(g_parked->p, *p)) {
            ctx = g_parked;
            g_parked = nullptr;
        }
    }
    if (!ctx) return nullptr;
    (void)hipSetDevice(p->device);
    ctx->p = *p;
    jb_knobs(ctx);
    ctx->jb_auto = ctx->p.join_blocks < 1;
    if (ctx->p.join_blocks < 1) ctx->p.join_blocks = auto_join_blocks(ctx->p.n_neighbors, ctx->p.max_candidates);
    ctx->seed = nnd_mix32((uint32_t)p->rng_state[0] ^ nnd_mix32((uint32_t)p->rng_state[1] + 0x9E3779B9u) ^ nnd_mix32((uint32_t)p->rng_state[2] + 0x7F4A7C15u));
    ctx->tree_seed = nnd_mix32((uint32_t)p->tree_rng[0] ^ nnd_mix32((uint32_t)p->tree_rng[1] + 0x9E3779B9u) ^ nnd_mix32((uint32_t)p->tree_rng[2] + 0x7F4A7C15u));
    ctx->iter = 0;
    ctx->stats = nnd_stats{};
    ctx->err[0] = 0;
    ctx->forest_built = false;
    ctx->h_leaf_valid = false;
    ctx->n_leaves = 0;
    ctx->max_leaf = 0;
    ctx->own_order = nullptr;
    ctx->lists_replicated = false;
    nnd_hub_tree_free(ctx);
    if (!ctx->stream_owned) {  // a borrowed stream must not outlive its lender
        ctx->stream = nullptr;
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { destroy_now(ctx); return nullptr; }
        ctx->stream_owned = true;
    }
    if (!ctx->x_owned) ctx->x_orig = nullptr;  // a borrowed point set is gone; an owned copy's buffer is reused by nnd_set_data_host
    ctx->x_valid = false;
    ctx->tlog.clear();
    ctx->tev_used = 0;
    return ctx;
}

extern "C" int32_t nnd_destroy(nnd_handle_t ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->p.device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    nnd_ctx *old = nullptr;
    if (ctx->n_ranks == 0) {  // plain handles only: a shard's tables are sized by its slice
        std::lock_guard<std::mutex> lk(g_park_mu);
        old = g_parked;
        g_parked = ctx;
    } else {
        old = ctx;
    }
    if (old) destroy_now(old);
    return 0;
}
// release the parked handle's device memory now
extern "C" int32_t nnd_release_pending(void) {
    nnd_release_parked();
    return 0;
}

#define ENTER(ctx)                                           \
    if (!ctx) { gerr("null handle"); return 1; }             \
    API_HIP(hipSetDevice(ctx->p.device));

static int need_data(nnd_ctx *ctx) {
    if (!ctx->x_orig || !ctx->x_valid) { ctx->set_error("no data set (call nnd_set_data_host/device first)"); return 1; }
    return 0;
}
// build entry points: the handle must hold the graph state (not an auxiliary NND_FLAG_NO_GRAPH handle)
static int need_graph(nnd_ctx *ctx) {
    if (ctx->p.flags & NND_FLAG_NO_GRAPH) { ctx->set_error("this handle was created with NND_FLAG_NO_GRAPH: it has no k-lists / candidate tables (pruning pass and hub tree only)"); return 1; }
    return 0;
}

static int after_data(nnd_ctx *ctx) {
    if (ctx->p.flags & NND_FLAG_NO_PREP) return 0;  // hub-tree handle: the original rows are all it reads
    const int t_ = t_begin(ctx);
    if (nnd_launch_prep(ctx)) return 1;
    if (!(ctx->p.flags & NND_FLAG_NO_GRAPH) && nnd_launch_reset_graph(ctx)) return 1;
    t_end(ctx, t_, &ctx->stats.ms_prep, false);
    t_flush(ctx);
    return 0;
}

// Pageable host memory -> device.  The runtime stages such a copy through its own pinned buffers; how fast depends on the box
// (one staging thread, the NUMA node of the caller's pages): the same 488 MB took 9.7 ms on one MI355X host and visibly more on
// another (round-5 review: 32.9 vs 42.2 ms for the whole nnd_build call).  Here: eight pinned 8 MB buffers per device (allocated
// once per process), eight host threads -- thread t copies chunks t, t + 8, ... into ITS buffer and queues the DMA of each on the
// handle's stream itself (the chunks are independent; what follows on the stream is ordered behind all of them).  A pinned
// source is copied directly.
static std::mutex g_up_mu[64];  // per device: the ranks of nnd_build_multi upload side by side
static char *g_up_stage[64][16] = {};
static hipEvent_t g_up_ev[64][16] = {};
static int h2d_parallel(nnd_ctx *ctx, void *dst_dev, const void *src, size_t bytes) {
    constexpr size_t STAGE = (size_t)8 << 20;
    int P = 8;  // staging threads (= buffers); 4 .. 16 measured the same 10.8 ms for 488 MB: the link, not the host copies, sets the rate
    if (const char *e = nnd_knob("NND_H2D_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 16) P = v; }
    bool direct = bytes < (size_t)(16u << 20) || ctx->p.device < 0 || ctx->p.device >= 64;
    if (!direct) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost) direct = true;
        else (void)hipGetLastError();
    }
    if (direct) {
        API_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return 0;
    }
    const int dev = ctx->p.device;
    std::lock_guard<std::mutex> lk(g_up_mu[dev]);
    API_HIP(hipSetDevice(dev));
    for (int b = 0; b < P; b++) {
        if (!g_up_stage[dev][b] && hipHostMalloc((void **)&g_up_stage[dev][b], STAGE, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();  // no pinned memory for the staging buffers: the runtime's own pageable path
            g_up_stage[dev][b] = nullptr;
            API_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, ctx->stream));
            return 0;
        }
        if (!g_up_ev[dev][b]) {
            API_HIP(hipEventCreateWithFlags(&g_up_ev[dev][b], hipEventDisableTiming));
        } else {
            API_HIP(hipEventSynchronize(g_up_ev[dev][b]));  // (a previous call's last DMA out of this buffer)
        }
    }
    const size_t nchunks = (bytes + STAGE - 1) / STAGE;
    std::atomic<int> failed{0};
    hipStream_t st = ctx->stream;
    auto work = [&](int t) {
        if (hipSetDevice(dev) != hipSuccess) { failed = 1; return; }
        bool first = true;
        for (size_t c = (size_t)t; c < nchunks && !failed; c += P) {
            const size_t o = c * STAGE, len = bytes - o < STAGE ? bytes - o : STAGE;
            if (!first && hipEventSynchronize(g_up_ev[dev][t]) != hipSuccess) { failed = 1; return; }
            first = false;
            memcpy(g_up_stage[dev][t], (const char *)src + o, len);
            if (hipMemcpyAsync((char *)dst_dev + o, g_up_stage[dev][t], len, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipEventRecord(g_up_ev[dev][t], st) != hipSuccess) { failed = 1; return; }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < P; t++) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    if (failed) { (void)hipGetLastError(); ctx->set_error("nnd_set_data_host: staged host-to-device copy failed"); return 1; }
    return 0;
}

extern "C" int32_t nnd_set_data_host(nnd_handle_t ctx, const float *x) {
    ENTER(ctx);
    if (!x) { ctx->set_error("nnd_set_data_host: null data"); return 1; }
    if (!(ctx->x_owned && ctx->x_orig)) {  // the handle's own copy of the rows: allocated once, reused by later calls
        float *dx = nullptr;
        API_HIP(hipMalloc((void **)&dx, sizeof(float) * (size_t)ctx->n * ctx->d));
        ctx->x_orig = dx;
        ctx->x_owned = true;
    }
    if (h2d_parallel(ctx, (void *)ctx->x_orig, x, sizeof(float) * (size_t)ctx->n * ctx->d)) return 1;
    ctx->x_valid = true;
    return after_data(ctx);
}

extern "C" int32_t nnd_set_data_device(nnd_handle_t ctx, const float *x_dev) {
    ENTER(ctx);
    if (!x_dev) { ctx->set_error("nnd_set_data_device: null data"); return 1; }
    if (ctx->x_owned && ctx->x_orig) { API_HIP(hipFree((void *)ctx->x_orig)); }
    ctx->x_orig = x_dev;
    ctx->x_owned = false;
    ctx->x_valid = true;
    return after_data(ctx);
}

// 1 when the point set handed to nnd_set_data_* held a NaN or an infinity (seen by the prep kernel): the reference rejects
// such input in check_array (pynndescent_.py:1054); the host mirror raises the same error from this flag
extern "C" int32_t nnd_data_nonfinite(nnd_handle_t ctx, int32_t *out) {
    ENTER(ctx);
    if (need_data(ctx)) return 1;
    if (ctx->p.flags & NND_FLAG_NO_PREP) { *out = 0; return 0; }
    API_HIP(nnd_sync_spin(ctx));
    *out = ctx->h_pin[63] != 0 ? 1 : 0;
    return 0;
}

extern "C" int32_t nnd_make_forest(nnd_handle_t ctx) {
    ENTER(ctx);
    if (ctx->p.flags & NND_FLAG_NO_PREP) { ctx->set_error("nnd_make_forest: this handle holds no prepared rows (NND_FLAG_NO_PREP)"); return 1; }
    if (need_data(ctx)) return 1;
    const int t_ = t_begin(ctx);
    if (nnd_launch_forest(ctx)) return 1;
    t_end(ctx, t_, &ctx->stats.ms_forest, false);
    t_flush(ctx);
    return 0;
}

extern "C" int32_t nnd_leaf_array_shape(nnd_handle_t ctx, int64_t *n_leaves, int32_t *max_leaf_size) {
    ENTER(ctx);
    if (!ctx->forest_built) {  // rp_trees.py:2921-2922: np.array([[-1]])
        *n_leaves = 1;
        *max_leaf_size = 1;
        return 0;
    }
    *n_leaves = ctx->n_leaves;
    *max_leaf_size = ctx->max_leaf;
    return 0;
}

extern "C" int32_t nnd_get_leaf_array(nnd_handle_t ctx, int32_t *out_host) {
    ENTER(ctx);
    if (!ctx->forest_built) {
        out_host[0] = -1;
        return 0;
    }
    size_t total = (size_t)ctx->n_leaves * ctx->max_leaf;
    nnd_scratch tmp;
    int32_t *d = tmp.get<int32_t>(ctx, total);
    if (!d) return 1;
    if (nnd_launch_leaf_array(ctx, d)) return 1;
    API_HIP(hipMemcpyAsync(out_host, d, sizeof(int32_t) * total, hipMemcpyDeviceToHost, ctx->stream));
    API_HIP(nnd_sync_spin(ctx));
    return 0;
}

extern "C" int32_t nnd_reset_graph(nnd_handle_t ctx) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    return nnd_launch_reset_graph(ctx);
}

extern "C" int32_t nnd_init_from_leaves(nnd_handle_t ctx) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    if (need_data(ctx)) return 1;
    const int t_ = t_begin(ctx);
    if (nnd_launch_leaf_init(ctx)) return 1;
    t_end(ctx, t_, &ctx->stats.ms_leaf_init, false);
    t_flush(ctx);
    return 0;
}

// init_rp_tree with the caller's leaf_array (the `leaf_array` argument of nn_descent, pynndescent_.py:324-337)
extern "C" int32_t nnd_init_from_leaf_array(nnd_handle_t ctx, const int32_t *leaf_array, int64_t n_leaves, int32_t max_leaf_size) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    if (need_data(ctx)) return 1;
    if (!leaf_array || n_leaves < 0 || max_leaf_size < 1) { ctx->set_error("nnd_init_from_leaf_array: bad arguments"); return 1; }
    const int t_ = t_begin(ctx);
    if (nnd_launch_leaf_init_array(ctx, leaf_array, n_leaves, max_leaf_size)) return 1;
    t_end(ctx, t_, &ctx->stats.ms_leaf_init, false);
    t_flush(ctx);
    return 0;
}

extern "C" int32_t nnd_init_random(nnd_handle_t ctx) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    if (need_data(ctx)) return 1;
    const int t_ = t_begin(ctx);
    if (nnd_launch_random_init(ctx)) return 1;
    t_end(ctx, t_, &ctx->stats.ms_random_init, false);
    t_flush(ctx);
    return 0;
}

extern "C" int32_t nnd_init_from_graph(nnd_handle_t ctx, const int32_t *init_idx, const float *init_dist, int32_t width) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    if (need_data(ctx)) return 1;
    if (!init_idx || width < 1 || width > NND_WIDE_K) { ctx->set_error("nnd_init_from_graph: width must be in 1..%d", NND_WIDE_K); return 1; }
    // the caller's arrays are the FULL (n, width) graph; the launcher takes the rows this handle OWNS (all of them, or a
    // shard's slice -- nnd_shard_handle exposes such handles): upload exactly those
    const int64_t rows = ctx->own_hi - ctx->own_lo;
    if (rows <= 0) return 0;
    const size_t cnt = (size_t)rows * width, off = (size_t)ctx->own_lo * width;
    nnd_scratch tmp;
    int32_t *di = tmp.get<int32_t>(ctx, cnt);
    float *dd = init_dist ? tmp.get<float>(ctx, cnt) : nullptr;
    if (!di || (init_dist && !dd)) return 1;
    API_HIP(hipMemcpyAsync(di, init_idx + off, sizeof(int32_t) * cnt, hipMemcpyHostToDevice, ctx->stream));
    if (init_dist) API_HIP(hipMemcpyAsync(dd, init_dist + off, sizeof(float) * cnt, hipMemcpyHostToDevice, ctx->stream));
    int rc = nnd_launch_init_from_graph(ctx, di, dd, width);
    (void)hipStreamSynchronize(ctx->stream);  // the scratch buffers are released on return
    return rc;
}

// init_from_neighbor_graph (pynndescent_.py:206-214), the warm start of NNDescent.update: the entries of an existing
// graph (alt-space distances given) are inserted with flag 0 ("old").  Call on a freshly reset graph.
extern "C" int32_t nnd_init_from_neighbor_graph(nnd_handle_t ctx, const int32_t *init_idx, const float *init_dist, int32_t width) {
    if (!ctx) return 1;
    if (!init_dist) { ENTER(ctx); ctx->set_error("nnd_init_from_neighbor_graph: distances are required"); return 1; }
    if (nnd_init_from_graph(ctx, init_idx, init_dist, width)) return 1;
    if (nnd_launch_clear_new_flags(ctx)) return 1;
    ctx->all_new = false;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->set_error("nnd_init_from_neighbor_graph: synchronize failed"); return 1; }
    return 0;
}

extern "C" int32_t nnd_sample_candidates(nnd_handle_t ctx) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    return nnd_launch_sample(ctx);
}

// Sub-steps of an iteration (join a part of the vertices, merge, ...).  The reference applies the updates of every block of 16384
// vertices before it generates the next block's (pynndescent_.py:239-261): thresholds tighten INSIDE an iteration, and a row
// never has more than a block's worth of pushes pending.  One launch per iteration (rounds 1-5) leaves a row's 64 hashed
// proposal slots to take a whole iteration's proposals -- 23 per row in the first iteration of the 1 M bench set, more where
// convergence is slow -- and what collides is lost: measured on 200 000 iid Gaussian points x 32 (the reference algorithm
// reaches recall@10 0.613 there, tools/mid_regime_study.py): 0.6094 with one launch, 0.6108 / 0.6112 / 0.6125 with 2 / 4 / 12
// sub-steps (12 = the reference's blocking at that size).  When join_blocks is left to the library the sub-steps of an iteration
// follow the insertions per row the previous iteration made (halving is the slowest decay seen), so that late iterations --
// few updates, nothing to collide -- stay single launches.  An explicit join_blocks is taken as given.
int nnd_join_substeps(const nnd_ctx *ctx) {
    int nb = ctx->p.join_blocks;
    if (!ctx->jb_auto || ctx->k > 64) return nb;  // (wide rows: auto_join_blocks has cut their iterations already)
    const double rows = (double)(ctx->own_hi - ctx->own_lo);
    const double per_row = (ctx->iter == 0 || ctx->last_updates < 0 || rows <= 0) ? (double)ctx->jb_first : (double)ctx->last_updates / rows;
    int m = 1;
    while (m < ctx->jb_max && (double)(m * ctx->jb_div) < per_row) m <<= 1;
    return nb * m;
}

// one iteration of nn_descent_internal (pynndescent_.py:296-320)
static int descent_iter(nnd_ctx *ctx, int64_t *c_out, bool timed) {
    const int it = ctx->iter;
    static float sink;  // timer target for iterations beyond the 64 that the stats block records
    {
        const int t_ = timed ? t_begin(ctx) : -1;
        if (nnd_launch_sample(ctx)) return 1;
        if (timed) t_end(ctx, t_, it < 64 ? &ctx->stats.ms_sample[it] : &sink, false);
    }
    if (nnd_zero_counters(ctx)) return 1;
    // The reference joins vertices in blocks of 16384 and applies updates between blocks
    // (pynndescent_.py:239-261) so thresholds tighten inside an iteration; join_blocks sub-steps do the same.
    const int nb = nnd_join_substeps(ctx);
    if (it < 64) {
        ctx->stats.ms_join[it] = ctx->stats.ms_merge[it] = 0.f;
        ctx->stats.join_substeps[it] = nb;
    }
    for (int b = 0; b < nb; b++) {
        const int64_t span = ctx->own_hi - ctx->own_lo;
        int64_t v0 = ctx->own_lo + span * b / nb, v1 = ctx->own_lo + span * (b + 1) / nb;
        const int tj = timed ? t_begin(ctx) : -1;
        if (nnd_launch_join(ctx, v0, v1)) return 1;
        if (timed) t_end(ctx, tj, it < 64 ? &ctx->stats.ms_join[it] : &sink, true);
        const int tm = timed ? t_begin(ctx) : -1;
        if (nnd_launch_merge(ctx)) return 1;
        if (timed) t_end(ctx, tm, it < 64 ? &ctx->stats.ms_merge[it] : &sink, true);
    }
    if (nnd_read_counters(ctx)) return 1;  // the host needs c here anyway: the timers are read at no extra wait
    t_flush(ctx);
    if (it < 64) {
        ctx->stats.join_pairs[it] = ctx->h_counters[CNT_PAIRS];
        ctx->stats.join_rows[it] = ctx->h_counters[CNT_ROWS];
        ctx->stats.join_active[it] = ctx->h_counters[CNT_ACTIVE];
        ctx->stats.proposals[it] = ctx->h_counters[CNT_PROPOSALS];
        ctx->stats.updates[it] = ctx->h_counters[CNT_ACCEPT];
        ctx->stats.join_mfma[it] = ctx->h_counters[CNT_MFMA];
    }
    *c_out = ctx->h_counters[CNT_ACCEPT];
    ctx->last_updates = ctx->h_counters[CNT_ACCEPT];
    ctx->iter++;
    ctx->stats.n_iters_run = ctx->iter;
    return 0;
}

extern "C" int32_t nnd_descent_iter(nnd_handle_t ctx, int64_t *c_out) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    if (need_data(ctx)) return 1;
    int64_t c = 0;
    if (descent_iter(ctx, &c, true)) return 1;
    if (c_out) *c_out = c;
    return 0;
}

static int descent_loop(nnd_ctx *ctx, bool timed) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, ctx->stream);
    int rc = 0;
    for (int it = 0; it < ctx->p.n_iters; it++) {
        int64_t c = 0;
        if ((rc = descent_iter(ctx, &c, timed))) break;
        if ((double)c <= (double)ctx->p.delta * ctx->k * (double)ctx->n) break;  // pynndescent_.py:317
    }
    (void)hipEventRecord(e1, ctx->stream);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ctx->stats.ms_descent, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

extern "C" int32_t nnd_descent(nnd_handle_t ctx) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    if (need_data(ctx)) return 1;
    return descent_loop(ctx, true);
}

extern "C" int32_t nnd_finalize_device(nnd_handle_t ctx, int32_t *out_idx_dev, float *out_dist_dev) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    if (need_data(ctx)) return 1;
    const int t_ = t_begin(ctx);
    if (nnd_launch_finalize(ctx, out_idx_dev, out_dist_dev)) return 1;
    t_end(ctx, t_, &ctx->stats.ms_finalize, false);
    t_flush(ctx);
    return 0;
}

// Device -> pageable host memory.  The runtime stages such a copy through pinned buffers with a single-threaded memcpy
// (~9 GB/s: 13 ms for the 114 MB graph of a 1 M-point index).  Here: two pinned 32 MB buffers (allocated once per
// process), the DMA of chunk c + 1 in flight while chunk c is copied out of its buffer by four host threads.
static std::mutex g_stage_mu;
// per DEVICE: an event can only be recorded on a stream of the device it was created on (a build on device 1 after one on
// device 0 in the same process), and the pinned buffers are registered with the device that was current at allocation
static char *g_stage_dev[64][2] = {{nullptr, nullptr}};
static hipEvent_t g_stage_ev_dev[64][2] = {{nullptr, nullptr}};
static int d2h_parallel(nnd_ctx *ctx, void *dst, const void *src, size_t bytes, int parts) {
    constexpr size_t STAGE = (size_t)32 << 20;
    if (bytes < (size_t)(4u << 20) || ctx->p.device < 0 || ctx->p.device >= 64) {
        API_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
        return 0;
    }
    {   // a pinned destination (nnd_host_alloc: the result arrays of the drop-in class) takes the DMA directly
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeHost) {
            API_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
            API_HIP(hipStreamSynchronize(ctx->stream));
            return 0;
        }
        (void)hipGetLastError();  // (an ordinary host pointer is "invalid value" to the query)
    }
    std::lock_guard<std::mutex> lk(g_stage_mu);
    char **g_stage = g_stage_dev[ctx->p.device];
    hipEvent_t *g_stage_ev = g_stage_ev_dev[ctx->p.device];
    API_HIP(hipSetDevice(ctx->p.device));
    for (int b = 0; b < 2; b++) {
        if (!g_stage[b]) API_HIP(hipHostMalloc((void **)&g_stage[b], STAGE, hipHostMallocDefault));
        if (!g_stage_ev[b]) API_HIP(hipEventCreateWithFlags(&g_stage_ev[b], hipEventDisableTiming));
    }
    const size_t nchunks = (bytes + STAGE - 1) / STAGE;
    for (size_t c = 0; c <= nchunks; c++) {
        if (c < nchunks) {
            const size_t o = c * STAGE, len = bytes - o < STAGE ? bytes - o : STAGE;
            API_HIP(hipMemcpyAsync(g_stage[c & 1], (const char *)src + o, len, hipMemcpyDeviceToHost, ctx->stream));
            API_HIP(hipEventRecord(g_stage_ev[c & 1], ctx->stream));
        }
        if (c >= 1) {
            const size_t o = (c - 1) * STAGE, len = bytes - o < STAGE ? bytes - o : STAGE;
            API_HIP(hipEventSynchronize(g_stage_ev[(c - 1) & 1]));
            const char *from = g_stage[(c - 1) & 1];
            char *to = (char *)dst + o;
            std::vector<std::thread> th;
            const size_t piece = ((len + parts - 1) / parts + 4095) & ~(size_t)4095;
            for (int t = 1; t < parts; t++) {
                const size_t po = (size_t)t * piece;
                if (po >= len) break;
                const size_t pl = len - po < piece ? len - po : piece;
                th.emplace_back([=] { memcpy(to + po, from + po, pl); });
            }
            memcpy(to, from, len < piece ? len : piece);
            for (auto &t : th) t.join();
        }
    }
    return 0;
}

// ---- host-side helpers of the drop-in class (include/pynnd_amd.h): first-touch-bound array operations over a few threads
template <typename F>
static void host_parallel(size_t bytes, size_t unit, F fn) {  // fn(offset_units, count_units); pieces are multiples of a page
    const size_t total = bytes / unit;
    int parts = bytes >= ((size_t)32 << 20) ? 16 : (bytes >= ((size_t)4 << 20) ? 8 : 1);
    const unsigned hc = std::thread::hardware_concurrency();
    if (hc && (unsigned)parts > hc) parts = (int)hc;
    const size_t per = ((total + parts - 1) / parts + (4096 / unit) - 1) / (4096 / unit) * (4096 / unit);
    std::vector<std::thread> th;
    for (int t = 1; t < parts; t++) {
        const size_t o = (size_t)t * per;
        if (o >= total) break;
        const size_t c = total - o < per ? total - o : per;
        th.emplace_back([=] { fn(o, c); });
    }
    fn(0, total < per ? total : per);
    for (auto &t : th) t.join();
}
// Pinned (page-locked, resident) host memory for result arrays: no first-touch page faults when the graph lands in it, and the
// device-to-host copy is one DMA at the link rate instead of a staged copy.  NULL when there is no device or no memory: the
// caller then uses ordinary memory.
extern "C" void *nnd_host_alloc(int64_t bytes) {
    void *p = nullptr;
    if (bytes <= 0) return nullptr;
    if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
extern "C" int32_t nnd_host_free(void *p) {
    if (p && hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); gerr("nnd_host_free: not a pointer of nnd_host_alloc"); return 1; }
    return 0;
}
extern "C" int32_t nnd_host_copy(void *dst, const void *src, int64_t bytes) {
    if (bytes < 0 || (bytes > 0 && (!dst || !src))) { gerr("nnd_host_copy: bad arguments"); return 1; }
    host_parallel((size_t)bytes, 1, [=](size_t o, size_t c) { memcpy((char *)dst + o, (const char *)src + o, c); });
    return 0;
}
// IEEE square roots, eight per instruction where the host has AVX2 (vsqrtps is correctly rounded: the same bits as sqrtf / numpy.sqrt;
// the scalar loop does not vectorise under the default -fmath-errno)
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(_M_X64))
#include <immintrin.h>
__attribute__((target("avx2"))) static void host_sqrt_avx2(float *dst, const float *src, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) _mm256_storeu_ps(dst + i, _mm256_sqrt_ps(_mm256_loadu_ps(src + i)));
    for (; i < n; i++) dst[i] = sqrtf(src[i]);
}
static bool host_has_avx2() { return __builtin_cpu_supports("avx2"); }
#else
static void host_sqrt_avx2(float *dst, const float *src, size_t n) { for (size_t i = 0; i < n; i++) dst[i] = sqrtf(src[i]); }
static bool host_has_avx2() { return false; }
#endif
extern "C" int32_t nnd_host_sqrt_f32(float *dst, const float *src, int64_t count) {
    if (count < 0 || (count > 0 && (!dst || !src))) { gerr("nnd_host_sqrt_f32: bad arguments"); return 1; }
    const bool avx2 = host_has_avx2();
    host_parallel((size_t)count * sizeof(float), sizeof(float), [=](size_t o, size_t c) {
        if (avx2) host_sqrt_avx2(dst + o, src + o, c);
        else
            for (size_t i = o; i < o + c; i++) dst[i] = sqrtf(src[i]);
    });
    return 0;
}

// grow-only device buffers for the finished graph of the host-buffer entry points (no hipMalloc / hipFree per call)
static int out_buffers(nnd_ctx *ctx, size_t cnt) {
    if (cnt <= ctx->out_cap) return 0;
    if (ctx->out_idx) { API_HIP(hipFree(ctx->out_idx)); ctx->out_idx = nullptr; }
    if (ctx->out_dist) { API_HIP(hipFree(ctx->out_dist)); ctx->out_dist = nullptr; }
    ctx->out_cap = 0;
    API_HIP(hipMalloc((void **)&ctx->out_idx, sizeof(int32_t) * cnt));
    API_HIP(hipMalloc((void **)&ctx->out_dist, sizeof(float) * cnt));
    ctx->out_cap = cnt;
    return 0;
}

extern "C" int32_t nnd_finalize_host(nnd_handle_t ctx, int32_t *out_idx, float *out_dist) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    if (need_data(ctx)) return 1;
    size_t cnt = (size_t)(ctx->own_hi - ctx->own_lo) * ctx->k;  // owned rows only
    if (out_buffers(ctx, cnt)) return 1;
    if (nnd_finalize_device(ctx, ctx->out_idx, ctx->out_dist)) return 1;  // (ends with a flush: the stream has drained)
    if (d2h_parallel(ctx, out_idx, ctx->out_idx, sizeof(int32_t) * cnt, 4)) return 1;
    if (d2h_parallel(ctx, out_dist, ctx->out_dist, sizeof(float) * cnt, 4)) return 1;
    return 0;
}

// nn_descent (pynndescent_.py:323-366) on a resident point set: EMPTY_GRAPH branch
extern "C" int32_t nnd_build_device(nnd_handle_t ctx, int32_t *out_idx_dev, float *out_dist_dev) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    if (need_data(ctx)) return 1;
    if (nnd_launch_reset_graph(ctx)) return 1;
    if (ctx->p.n_trees > 0) {
        const int tf = t_begin(ctx);
        if (nnd_launch_forest(ctx)) return 1;
        t_end(ctx, tf, &ctx->stats.ms_forest, false);
        const int tl = t_begin(ctx);
        if (nnd_launch_leaf_init(ctx)) return 1;
        t_end(ctx, tl, &ctx->stats.ms_leaf_init, false);
    }
    const int tr = t_begin(ctx);
    if (nnd_launch_random_init(ctx)) return 1;
    t_end(ctx, tr, &ctx->stats.ms_random_init, false);
    if (descent_loop(ctx, true)) return 1;
    const int t_ = t_begin(ctx);
    if (nnd_launch_finalize(ctx, out_idx_dev, out_dist_dev)) return 1;
    t_end(ctx, t_, &ctx->stats.ms_finalize, false);
    t_flush(ctx);
    return 0;
}

extern "C" int32_t nnd_build(const nnd_params *params, const float *x, const int32_t *init_idx, const float *init_dist,
                             int32_t init_width, int32_t *out_idx, float *out_dist, nnd_stats *stats, char *err,
                             int32_t errlen) {
    auto fail = [&](const char *msg) {
        if (err && errlen > 0) { strncpy(err, msg, (size_t)errlen - 1); err[errlen - 1] = 0; }
        return 1;
    };
    nnd_handle_t h = nullptr;
    nnd_params p = *params;
    if (init_idx) p.n_trees = 0;  // pynndescent_.py:1059-1062: an init graph disables the forest
    if (nnd_create(&h, &p)) return fail(g_err);
    int rc = nnd_set_data_host(h, x);
    if (!rc) {
        if (init_idx) {
            rc = nnd_init_from_graph(h, init_idx, init_dist, init_width);
            if (!rc) rc = nnd_descent(h);
            if (!rc) rc = nnd_finalize_host(h, out_idx, out_dist);
        } else {
            if (!rc) rc = out_buffers(h, (size_t)h->n * h->k);
            if (!rc) rc = nnd_build_device(h, h->out_idx, h->out_dist);
            if (!rc) rc = d2h_parallel(h, out_idx, h->out_idx, sizeof(int32_t) * (size_t)h->n * h->k, 4);
            if (!rc) rc = d2h_parallel(h, out_dist, h->out_dist, sizeof(float) * (size_t)h->n * h->k, 4);
        }
    }
    if (stats) *stats = h->stats;
    std::string msg = h->err;
    nnd_destroy(h);
    if (rc) return fail(msg.c_str());
    return 0;
}

extern "C" int32_t nnd_get_stats(nnd_handle_t ctx, nnd_stats *out) {
    if (!ctx || !out) { gerr("null argument"); return 1; }
    *out = ctx->stats;
    return 0;
}

extern "C" int32_t nnd_synchronize(nnd_handle_t ctx) {
    ENTER(ctx);
    API_HIP(nnd_sync_spin(ctx));
    return 0;
}

// ---- introspection for the parity tests ----
extern "C" int32_t nnd_get_graph(nnd_handle_t ctx, int32_t *idx, float *dist, uint8_t *flags) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    size_t cnt = (size_t)ctx->n * ctx->ks;
    std::vector<uint32_t> he(cnt);
    std::vector<float> hd(cnt);
    API_HIP(hipMemcpyAsync(he.data(), ctx->knn_e, sizeof(uint32_t) * cnt, hipMemcpyDeviceToHost, ctx->stream));
    API_HIP(hipMemcpyAsync(hd.data(), ctx->knn_d, sizeof(float) * cnt, hipMemcpyDeviceToHost, ctx->stream));
    API_HIP(nnd_sync_spin(ctx));
    for (int64_t v = 0; v < ctx->n; v++)
        for (int j = 0; j < ctx->k; j++) {
            uint32_t e = he[v * ctx->ks + j];
            size_t o = (size_t)v * ctx->k + j;
            if (idx) idx[o] = e == NND_EMPTY_E ? -1 : (int32_t)(e & NND_IDX_MASK);
            if (dist) dist[o] = hd[v * ctx->ks + j];
            if (flags) flags[o] = e == NND_EMPTY_E ? 0 : (uint8_t)(e >> 31);
        }
    return 0;
}

extern "C" int32_t nnd_get_candidates(nnd_handle_t ctx, int32_t *new_idx, int32_t *old_idx) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    size_t cnt = (size_t)ctx->n * 2 * ctx->mcp;
    std::vector<int32_t> hc(cnt);
    API_HIP(hipMemcpyAsync(hc.data(), ctx->cand, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, ctx->stream));
    API_HIP(nnd_sync_spin(ctx));
    for (int64_t v = 0; v < ctx->n; v++)
        for (int j = 0; j < ctx->mc; j++) {
            if (new_idx) new_idx[v * ctx->mc + j] = hc[v * 2 * ctx->mcp + j];
            if (old_idx) old_idx[v * ctx->mc + j] = hc[v * 2 * ctx->mcp + ctx->mcp + j];
        }
    return 0;
}

extern "C" int32_t nnd_pairwise_gram(nnd_handle_t ctx, const int32_t *rows_a, int32_t na, const int32_t *rows_b,
                                     int32_t nb, float *out) {
    ENTER(ctx);
    if (ctx->p.flags & NND_FLAG_NO_PREP) { ctx->set_error("nnd_pairwise_gram: this handle holds no prepared rows (NND_FLAG_NO_PREP)"); return 1; }
    if (need_data(ctx)) return 1;
    nnd_scratch tmp;
    int32_t *da = tmp.get<int32_t>(ctx, (size_t)na), *db = tmp.get<int32_t>(ctx, (size_t)nb);
    float *dout = tmp.get<float>(ctx, (size_t)na * nb);
    if (!da || !db || !dout) return 1;
    API_HIP(hipMemcpyAsync(da, rows_a, sizeof(int32_t) * na, hipMemcpyHostToDevice, ctx->stream));
    API_HIP(hipMemcpyAsync(db, rows_b, sizeof(int32_t) * nb, hipMemcpyHostToDevice, ctx->stream));
    if (nnd_launch_pairwise(ctx, da, na, db, nb, dout)) return 1;
    API_HIP(hipMemcpyAsync(out, dout, sizeof(float) * (size_t)na * nb, hipMemcpyDeviceToHost, ctx->stream));
    API_HIP(nnd_sync_spin(ctx));
    return 0;
}

// Run on the caller's HIP stream (e.g. torch's current stream) instead of the handle's own: the library's kernels and the
// caller's work are then ordered by the stream itself, no host synchronisation between them.  NULL: back to own.
extern "C" int32_t nnd_set_stream(nnd_handle_t ctx, void *hip_stream) {
    ENTER(ctx);
    API_HIP(hipStreamSynchronize(ctx->stream));
    if (hip_stream) {
        if (ctx->stream_owned && ctx->stream) { API_HIP(hipStreamDestroy(ctx->stream)); }
        ctx->stream = (hipStream_t)hip_stream;
        ctx->stream_owned = false;
    } else if (!ctx->stream_owned) {
        API_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->stream_owned = true;
    }
    return 0;
}

extern "C" int32_t nnd_descent_sample(nnd_handle_t ctx) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    static float sink;
    const int t_ = t_begin(ctx);
    if (nnd_launch_sample(ctx)) return 1;
    t_end(ctx, t_, ctx->iter < 64 ? &ctx->stats.ms_sample[ctx->iter] : &sink, false);
    return 0;
}
extern "C" int32_t nnd_descent_join(nnd_handle_t ctx) {
    ENTER(ctx);
    if (need_graph(ctx)) return 1;
    if (nnd_zero_counters(ctx)) return 1;
    static float sink;
    const int t_ = t_begin(ctx);
    if (nnd_launch_join(ctx, ctx->own_lo, ctx->own_hi)) return 1;
    t_end(ctx, t_, ctx->iter < 64 ? &ctx->stats.ms_join[ctx->iter] : &sink, false);
    if (nnd_read_counters(ctx)) return 1;  // (a test / profiling entry point: the join's counters are in the stats when it returns)
    t_flush(ctx);
    if (ctx->iter < 64) {
        ctx->stats.join_pairs[ctx->iter] = ctx->h_counters[CNT_PAIRS];
        ctx->stats.join_rows[ctx->iter] = ctx->h_counters[CNT_ROWS];
        ctx->stats.join_active[ctx->iter] = ctx->h_counters[CNT_ACTIVE];
        ctx->stats.proposals[ctx->iter] = ctx->h_counters[CNT_PROPOSALS];
    }
    return 0;
}

// ---- search-graph pruning pass (BASELINE config 5); host glue: pynndescent_amd/search_graph.py ----
// Host-buffer entry points: the graph of this stage is handed over and taken back as numpy arrays by the
// reference as well (its glue between the numba kernels is scipy on the host, pynndescent_.py:1509-1611).
static nnd_prune_opts prune_defaults(const nnd_prune_opts *o) {
    nnd_prune_opts d{};
    d.prune_probability = 1.0f;
    d.alpha = 1.0f;
    d.max_degree = 1;
    return o ? *o : d;
}

extern "C" int32_t nnd_diversify_host(nnd_handle_t ctx, int32_t *idx /* (n,k) in/out */, float *dist /* (n,k) in/out */,
                                      const nnd_prune_opts *opts, const int32_t *degree /* (n), degree-aware only */) {
    ENTER(ctx);
    if (ctx->p.flags & NND_FLAG_NO_PREP) { ctx->set_error("nnd_diversify_host: this handle holds no prepared rows (NND_FLAG_NO_PREP)"); return 1; }
    if (need_data(ctx)) return 1;
    const nnd_prune_opts o = prune_defaults(opts);
    if (o.degree_aware && (!degree || o.max_degree < 1)) { ctx->set_error("nnd_diversify_host: the degree-aware method needs degrees and max_degree >= 1"); return 1; }
    size_t cnt = (size_t)ctx->n * ctx->k;
    nnd_scratch tmp;
    int32_t *di = tmp.get<int32_t>(ctx, cnt);
    float *dd = tmp.get<float>(ctx, cnt);
    int32_t *dg = o.degree_aware ? tmp.get<int32_t>(ctx, (size_t)ctx->n) : nullptr;
    if (!di || !dd || (o.degree_aware && !dg)) return 1;
    API_HIP(hipMemcpyAsync(di, idx, sizeof(int32_t) * cnt, hipMemcpyHostToDevice, ctx->stream));
    API_HIP(hipMemcpyAsync(dd, dist, sizeof(float) * cnt, hipMemcpyHostToDevice, ctx->stream));
    if (dg) API_HIP(hipMemcpyAsync(dg, degree, sizeof(int32_t) * (size_t)ctx->n, hipMemcpyHostToDevice, ctx->stream));
    if (nnd_launch_diversify_rows(ctx, di, dd, &o, dg)) return 1;
    API_HIP(hipMemcpyAsync(idx, di, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, ctx->stream));
    API_HIP(hipMemcpyAsync(dist, dd, sizeof(float) * cnt, hipMemcpyDeviceToHost, ctx->stream));
    API_HIP(nnd_sync_spin(ctx));
    return 0;
}

extern "C" int32_t nnd_diversify_csr_host(nnd_handle_t ctx, const int32_t *indptr /* n+1 */, const int32_t *indices,
                                          float *data /* nnz in/out */, int64_t nnz, const nnd_prune_opts *opts,
                                          const int32_t *degree /* (n), degree-aware only */) {
    ENTER(ctx);
    if (ctx->p.flags & NND_FLAG_NO_PREP) { ctx->set_error("nnd_diversify_csr_host: this handle holds no prepared rows (NND_FLAG_NO_PREP)"); return 1; }
    if (need_data(ctx)) return 1;
    const nnd_prune_opts o = prune_defaults(opts);
    if (o.degree_aware && !degree) { ctx->set_error("nnd_diversify_csr_host: the degree-aware method needs degrees"); return 1; }
    nnd_scratch tmp;
    int32_t *dp = tmp.get<int32_t>(ctx, (size_t)(ctx->n + 1)), *di = tmp.get<int32_t>(ctx, (size_t)nnz);
    float *dd = tmp.get<float>(ctx, (size_t)nnz);
    int *flag = tmp.get<int>(ctx, 1);
    int32_t *dg = o.degree_aware ? tmp.get<int32_t>(ctx, (size_t)ctx->n) : nullptr;
    if (!dp || !di || !dd || !flag || (o.degree_aware && !dg)) return 1;
    API_HIP(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
    API_HIP(hipMemcpyAsync(dp, indptr, sizeof(int32_t) * (size_t)(ctx->n + 1), hipMemcpyHostToDevice, ctx->stream));
    API_HIP(hipMemcpyAsync(di, indices, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
    API_HIP(hipMemcpyAsync(dd, data, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
    if (dg) API_HIP(hipMemcpyAsync(dg, degree, sizeof(int32_t) * (size_t)ctx->n, hipMemcpyHostToDevice, ctx->stream));
    if (nnd_launch_diversify_csr(ctx, dp, di, dd, flag, &o, dg)) return 1;
    int too_long = 0;
    API_HIP(hipMemcpyAsync(data, dd, sizeof(float) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
    API_HIP(hipMemcpyAsync(&too_long, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    API_HIP(nnd_sync_spin(ctx));
    if (too_long) {
        ctx->set_error("nnd_diversify_csr_host: %d rows are longer than 64 entries (rows of a diversified k-NN graph have <= k <= 64)", too_long);
        return 1;
    }
    return 0;
}

extern "C" int32_t nnd_degree_prune_host(nnd_handle_t ctx, const int32_t *indptr /* n+1 */, float *data /* nnz in/out */,
                                         int64_t nnz, int32_t max_degree) {
    ENTER(ctx);
    nnd_scratch tmp;
    int32_t *dp = tmp.get<int32_t>(ctx, (size_t)(ctx->n + 1));
    float *dd = tmp.get<float>(ctx, (size_t)nnz);
    if (!dp || !dd) return 1;
    API_HIP(hipMemcpyAsync(dp, indptr, sizeof(int32_t) * (size_t)(ctx->n + 1), hipMemcpyHostToDevice, ctx->stream));
    API_HIP(hipMemcpyAsync(dd, data, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
    if (nnd_launch_degree_prune(ctx, dp, dd, max_degree)) return 1;
    API_HIP(hipMemcpyAsync(data, dd, sizeof(float) * (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
    API_HIP(nnd_sync_spin(ctx));
    return 0;
}

// the whole pass on the device (searchgraph.hip)
extern "C" int32_t nnd_search_graph(nnd_handle_t ctx, const int32_t *idx, const float *dist, int32_t on_device, int32_t n_neighbors,
                                    float pruning_degree_multiplier, float diversify_prob, int32_t degree_aware, float degree_prune_aggressiveness,
                                    uint32_t seed, int32_t *fwd_rows_host, float *fwd_dist_host, nnd_search_graph_stats *stats) {
    ENTER(ctx);
    if (ctx->p.flags & NND_FLAG_NO_PREP) { ctx->set_error("nnd_search_graph: this handle holds no prepared rows (NND_FLAG_NO_PREP)"); return 1; }
    if (need_data(ctx)) return 1;
    if (!idx || !dist || n_neighbors < 1) { ctx->set_error("nnd_search_graph: bad arguments"); return 1; }
    return nnd_search_graph_impl(ctx, idx, dist, on_device != 0, n_neighbors, pruning_degree_multiplier, diversify_prob, degree_aware != 0,
                                 degree_prune_aggressiveness, seed, fwd_rows_host, fwd_dist_host, stats);
}
extern "C" int32_t nnd_search_graph_fetch(nnd_handle_t ctx, int32_t *indptr_host, int32_t *indices_host) {
    ENTER(ctx);
    if (!indptr_host) { ctx->set_error("nnd_search_graph_fetch: null argument"); return 1; }
    return nnd_search_graph_fetch_impl(ctx, indptr_host, indices_host);
}

// ---- hub search tree of NNDescent.prepare() (reference rp_trees.py:714-1312, 2926-3049; host glue: search_tree.py) ----
extern "C" int32_t nnd_hub_tree_build(nnd_handle_t ctx, const int32_t *rank_order /* host (n): ids by (-in-degree, id) */,
                                      int32_t leaf_size, int32_t max_depth, int64_t *n_nodes_out) {
    ENTER(ctx);
    if (need_data(ctx)) return 1;
    if (!rank_order) { ctx->set_error("nnd_hub_tree_build: null rank order"); return 1; }
    if (nnd_hub_tree_build_impl(ctx, rank_order, leaf_size, max_depth, ctx->p.metric == NND_METRIC_ALT_COSINE)) return 1;
    if (n_nodes_out) *n_nodes_out = nnd_hub_tree_nodes(ctx);
    return 0;
}
extern "C" int32_t nnd_hub_tree_fetch(nnd_handle_t ctx, float *hyperplanes, float *offsets, int32_t *children, int32_t *indices,
                                      int32_t *max_leaf_size) {
    ENTER(ctx);
    return nnd_hub_tree_fetch_impl(ctx, hyperplanes, offsets, children, indices, max_leaf_size);
}
